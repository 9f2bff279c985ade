// rp_coverage.hpp -- dense P-site coverage built on the device from columnar alignments
// (first part).
//
// The reference keeps the 5'-end histogram of the BAM as nested dicts / Counters keyed by
// (chrom, pos) (bam.py:105-135), shifts and merges the read lengths with one dict operation
// per key (merge_read_lengths, detect_orfs.py:54-83) and finally looks every nucleotide of
// every ORF up in the merged Counter (detect_orfs.py:176-187).  Here the histogram is a set of
// columns (group = (strand, chrom) of the candidate-ORF index, position already shifted by
// the read length's P-site offset, count), and ONE launch accumulates them into the dense
// coverage array the profile gather reads: coverage[start[g] + pos - lo[g]] += count.
// Entries of several read lengths that land on one position add up (atomics, integer: exact
// and order independent); entries outside every ORF's extent are dropped, which is what the
// reference's missing-key lookups amount to.
#pragma once

#include "rp_device.hpp"

namespace rp {

// ---------------------------------------------------------------------------
// Compact coverage (round 4).  The dense layout gives every position of every (strand, chromosome) extent a slot --
// 25 GB for a human index, 85 GB for the 11 M-ORF synthetic one -- although only positions under an exon are ever
// read (the reference looks nothing else up: detect_orfs.py:176-187).  A BLOCK MAP keeps the blocks of 2^shift
// positions that an exon interval touches and packs them in order: one bit per block, and per 64-bit word the number
// of kept blocks in front of it.  position -> slot is a shift, a popcount and an add; an interval stays contiguous
// (the blocks it covers are all kept, and consecutive kept blocks are consecutive slots).  With 64-position blocks
// the coverage shrinks to the exonic part + at most 126 positions per interval (85 GB -> 9 GB), and with it the
// allocation, the memset and the address range the scoring kernels wander over.  With ONE-position blocks (shift 0)
// nothing but exonic positions is left: exons that were neighbours across an intron become neighbours in memory, the
// pieces of a spliced ORF merge into one run of the gather plan, and the cache lines at both ends of a piece are
// shared with the next piece instead of being fetched for a few counts each (what the fused kernel paid 1.28x the
// algorithmic traffic for on gapped layouts); the map then takes dense_len / 4 bytes instead of dense_len / 256.
// ---------------------------------------------------------------------------
struct BlockMap {
    const unsigned long long *bits;  // [n_words]      bit b of word w: block 64 w + b (positions (64 w + b) << shift ...) is kept
    const long long *rank;           // [n_words + 1]  kept blocks in words < w; rank[n_words] = all of them
    long long n_words;
    int shift;                       // log2 of the block size in positions: 0 ... 6
};
constexpr int kMapChunk = 1024;  // words per scan chunk

inline bool map_block_ok(int block_positions) { return block_positions >= 1 && block_positions <= 64 && (block_positions & (block_positions - 1)) == 0; }
inline int map_shift(int block_positions) { return __builtin_ctz((unsigned)block_positions); }
inline long long map_words(long long dense_len, int shift) { return (dense_len + (64ll << shift) - 1) >> (6 + shift); }
inline long long map_chunks(long long n_words) { return (n_words + kMapChunk - 1) / kMapChunk; }
inline size_t map_bytes(long long dense_len, int shift)
{
    const long long w = map_words(dense_len, shift);
    return (size_t)w * 8 + (size_t)(w + 1) * 8 + (size_t)(map_chunks(w) + 1) * 8 + 256;
}

// slot of dense position idx, or -1 when its block is not kept (no exon interval touches it)
__device__ __forceinline__ long long map_position(const BlockMap &m, long long idx)
{
    const long long block = idx >> m.shift, w = block >> 6;
    if (idx < 0 || w >= m.n_words) return -1;
    const unsigned long long word = m.bits[w];
    const int b = (int)(block & 63);
    if (!((word >> b) & 1ull)) return -1;
    const long long kept = m.rank[w] + __builtin_popcountll(word & ((1ull << b) - 1ull));
    return (kept << m.shift) | (idx & ((1ll << m.shift) - 1));
}

__global__ void k_map_mark(const int64_t *__restrict__ iv_start, const int32_t *__restrict__ iv_len, long long n_iv,
                           long long dense_len, int shift, unsigned long long *__restrict__ bits, int *__restrict__ err)
{
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_iv) return;
    const long long s = iv_start[k], n = iv_len[k];
    if (n <= 0 || s < 0 || s + n > dense_len) {  // (not plannable: rp_gather_plan_create_dev reports it)
        atomicOr(err, 1);
        return;
    }
    const long long b0 = s >> shift, b1 = (s + n - 1) >> shift;  // blocks b0 .. b1, a word's worth at a time
    for (long long w = b0 >> 6; w <= b1 >> 6; ++w) {
        const int lo = w == (b0 >> 6) ? (int)(b0 & 63) : 0;
        const int hi = w == (b1 >> 6) ? (int)(b1 & 63) : 63;
        const unsigned long long mask = (hi == 63 ? ~0ull : (1ull << (hi + 1)) - 1ull) & ~((1ull << lo) - 1ull);
        if ((bits[w] & mask) != mask) atomicOr(&bits[w], mask);  // (nested candidate ORFs mark the same exons over and over)
    }
}

// exclusive scan of popcount(bits[w]) in three steps: per-chunk sums, the chunks' scan (one workgroup), per-chunk scans
__global__ __launch_bounds__(256) void k_map_chunk_sums(const unsigned long long *__restrict__ bits, long long n_words,
                                                        long long *__restrict__ partial)
{
    __shared__ int s_sum[4];
    const long long w0 = (long long)blockIdx.x * kMapChunk;
    int mine = 0;
    for (int i = threadIdx.x; i < kMapChunk; i += 256)
        if (w0 + i < n_words) mine += __builtin_popcountll(bits[w0 + i]);
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off, 64);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
}

__global__ __launch_bounds__(1024) void k_map_scan_partials(long long *__restrict__ partial, long long n_chunks)
{
    __shared__ long long s_carry;
    __shared__ long long s_wave[16];
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (long long c0 = 0; c0 < n_chunks + 1; c0 += 1024) {  // (entry n_chunks receives the total)
        const long long c = c0 + threadIdx.x;
        const long long v = c < n_chunks ? partial[c] : 0;
        long long incl = v;
        for (int off = 1; off < 64; off <<= 1) {
            const long long up = __shfl_up(incl, off, 64);
            if ((int)(threadIdx.x & 63) >= off) incl += up;
        }
        if ((threadIdx.x & 63) == 63) s_wave[threadIdx.x >> 6] = incl;
        __syncthreads();
        long long before = s_carry;
        for (int wv = 0; wv < (int)(threadIdx.x >> 6); ++wv) before += s_wave[wv];
        if (c <= n_chunks) partial[c] = before + incl - v;  // exclusive
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = before + incl;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_map_rank(const unsigned long long *__restrict__ bits, long long n_words,
                                                  const long long *__restrict__ partial, long long *__restrict__ rank)
{
    __shared__ int s_wave[4];
    const long long w0 = (long long)blockIdx.x * kMapChunk;
    long long carry = partial[blockIdx.x];
    for (int pass = 0; pass < kMapChunk / 256; ++pass) {
        const long long w = w0 + pass * 256 + threadIdx.x;
        const int v = w < n_words ? __builtin_popcountll(bits[w]) : 0;
        int incl = v;
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(incl, off, 64);
            if ((int)(threadIdx.x & 63) >= off) incl += up;
        }
        if ((threadIdx.x & 63) == 63) s_wave[threadIdx.x >> 6] = incl;
        __syncthreads();
        int before = 0;
        for (int wv = 0; wv < (int)(threadIdx.x >> 6); ++wv) before += s_wave[wv];
        if (w < n_words) rank[w] = carry + before + incl - v;
        const int total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        __syncthreads();
        carry += total;
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) rank[n_words] = partial[gridDim.x];
}

__global__ void k_map_remap(int64_t *__restrict__ iv_start, long long n_iv, BlockMap m)
{
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_iv) return;
    iv_start[k] = map_position(m, iv_start[k]);  // (every interval's blocks were marked: never -1)
}

// err[0] |= 1 when an accumulated count passes RP_MAX_COUNT (what the fp32 codon arithmetic of the scorers takes
// exactly: the caller finishes the ORFs that hold such a position in float64, rp_coverage_big_positions_dev),
// err[0] |= 2 when a count is negative or a sum passes INT32_MAX (not representable in the coverage array)
__global__ void k_coverage_build(const int32_t *__restrict__ group, const int64_t *__restrict__ pos,
                                 const int32_t *__restrict__ count, long long n,
                                 const int64_t *__restrict__ group_start, const int64_t *__restrict__ group_lo,
                                 const int64_t *__restrict__ group_hi, int n_groups,
                                 int32_t *__restrict__ coverage, long long coverage_len, int *__restrict__ err)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) {
        const int g = group[k];
        if (g < 0 || g >= n_groups) continue;
        const long long p = pos[k];
        if (p < group_lo[g] || p > group_hi[g]) continue;
        const long long idx = group_start[g] + (p - group_lo[g]);
        if (idx < 0 || idx >= coverage_len) continue;
        const int c = count[k];
        if (c < 0) {
            atomicOr(err, 2);
            continue;
        }
        const int before = atomicAdd(&coverage[idx], c);
        const long long after = (long long)before + c;
        if (before < 0 || after > 2147483647ll) atomicOr(err, 2);  // (wrapped)
        else if (after > RP_MAX_COUNT) atomicOr(err, 1);
    }
}

// The same straight from the columns merge_read_lengths hands over -- strand uint8, chromosome code
// int32, position int64, count int64 -- with the (strand, chromosome) -> group lookup done here
// (lut[strand * n_chroms + chrom], -1: no ORF lives there): no per-row work is left on the host.
// err bits as above; only rows that land inside a group's extent take part -- the reference never looks the
// others up (detect_orfs.py:176-187).
__global__ void k_coverage_build_rows(const uint8_t *__restrict__ strand, const int32_t *__restrict__ chrom,
                                      const int64_t *__restrict__ pos, const int64_t *__restrict__ count, long long n,
                                      const int32_t *__restrict__ lut, int n_chroms, const int64_t *__restrict__ group_start,
                                      const int64_t *__restrict__ group_lo, const int64_t *__restrict__ group_hi, int n_groups,
                                      int32_t *__restrict__ coverage, long long coverage_len, int *__restrict__ err,
                                      BlockMap map)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) {
        const int s = strand[k], c = chrom[k];
        if (s > 1 || c < 0 || c >= n_chroms) continue;
        const int g = lut[s * n_chroms + c];
        if (g < 0 || g >= n_groups) continue;
        const long long p = pos[k];
        if (p < group_lo[g] || p > group_hi[g]) continue;
        long long idx = group_start[g] + (p - group_lo[g]);
        if (map.bits != nullptr) idx = map_position(map, idx);  // compact coverage: positions under no exon have no slot (never looked up)
        if (idx < 0 || idx >= coverage_len) continue;
        const long long cnt = count[k];
        if (cnt < 0 || cnt > 2147483647ll) {
            atomicOr(err, 2);
            continue;
        }
        const int before = atomicAdd(&coverage[idx], (int)cnt);
        const long long after = (long long)before + cnt;
        if (before < 0 || after > 2147483647ll) atomicOr(err, 2);  // (wrapped)
        else if (after > RP_MAX_COUNT) atomicOr(err, 1);
    }
}

// The positions of a dense coverage whose count passes RP_MAX_COUNT, appended (any order) to `positions`
// while they fit; *found counts them all.  Runs only when a coverage build reported such a count: a pass
// over the whole array (a saturated position is a handful-per-sample event: rRNA / tRNA pile-ups).
__global__ void k_big_positions(const int32_t *__restrict__ coverage, long long coverage_len, long long *__restrict__ positions,
                                long long capacity, unsigned long long *__restrict__ found)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < coverage_len; k += stride) {
        if (coverage[k] > RP_MAX_COUNT) {
            const unsigned long long at = atomicAdd(found, 1ull);
            if ((long long)at < capacity) positions[at] = k;
        }
    }
}

// ---------------------------------------------------------------------------
// Metagene profiles (metagene_coverage, ribotricer/metagene.py:160-265), one read length.
// Input: the leader + ORF + trailer profiles of the annotated ORFs, already truncated to
// max_positions and in transcript orientation, CSR-packed (the profile gather made them).
//   k_metagene_means   mean coverage of every profile (from_start.mean(), metagene.py:213)
//   k_metagene_sums    one thread per metagene position and side: the ORFs are added IN INDEX
//                      ORDER, each value divided by its profile's mean first -- the very
//                      float64 operations pandas performs (Series / mean, then
//                      Series.add(fill_value=0) ORF after ORF, metagene.py:214-228), so the
//                      sums carry the reference's bits; profiles with mean <= 0 are skipped.
// side 0 aligns the profiles at their first position (start codon side), side 1 at their last
// (stop codon side: slot m = m-th position from the end).
// ---------------------------------------------------------------------------
__global__ void k_metagene_means(const int32_t *__restrict__ counts, const int64_t *__restrict__ offsets,
                                 long long n_orfs, double *__restrict__ mean)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_orfs) return;
    const long long beg = offsets[i], end = offsets[i + 1];
    long long s = 0;
    for (long long k = beg; k < end; ++k) s += counts[k];
    mean[i] = end > beg ? (double)s / (double)(end - beg) : 0.0;
}

__global__ void k_metagene_sums(const int32_t *__restrict__ counts, const int64_t *__restrict__ offsets,
                                const double *__restrict__ mean, long long n_orfs, int max_positions,
                                double *__restrict__ sum, int32_t *__restrict__ seen)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2 * max_positions) return;
    const int side = t / max_positions, slot = t % max_positions;
    double acc = 0.0;
    int n = 0;
    for (long long i = 0; i < n_orfs; ++i) {
        const double m = mean[i];
        if (!(m > 0.0)) continue;
        const long long beg = offsets[i];
        const long long len = offsets[i + 1] - beg;
        if (slot >= len) continue;
        const long long k = side == 0 ? beg + slot : beg + len - 1 - slot;
        acc = acc + (double)counts[k] / m;
        ++n;
    }
    sum[t] = acc;
    seen[t] = n;
}

}  // namespace rp
