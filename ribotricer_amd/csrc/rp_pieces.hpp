// rp_pieces.hpp -- the profile space of an index as PIECES of the dense coverage.
//
// The reference fills every ORF's profile with one dict lookup per nucleotide, exon by exon,
// reversed for '-' strand ORFs (orf_coverage, detect_orfs.py:134-203).  Here the concatenated
// profile space [0, total_nt) of a whole index is described once, from the interval table
// alone, as a sorted list of pieces -- piece j covers profile positions [start_j, start_j+1)
// and position p reads coverage[base_j + p] (forward) or coverage[base_j - p] ('-' strand) --
// and every flat tile of the scorer (rp_tile.hpp) gets a fixed-stride row of CHUNKS: its
// pieces clipped to the tile and cut into runs of <= 64 positions.  With that a workgroup
// stages its tile of counts straight from the coverage arrays with LDS-DMA, one chunk per
// instruction, issued from straight-line code:
//   * k_tile_gather writes the staged tile out: the CSR `counts` array (report_all mode);
//   * k_tile_score<true, TILE> scores it in place: the profiles are never written to HBM.
// The plan depends on the index (and the coverage layout derived from it) only, so it is
// built once per index and reused for every sample.
#pragma once

#include <cstddef>
#include <cstdint>

#include <hip/hip_runtime.h>

namespace rp {

typedef unsigned long long chunk_desc_t;

constexpr unsigned long long kPieceNeg = 1ull << 63;  // start word: the piece runs down the coverage array
constexpr int kFastSlots = 256;                    // the slots the fast path stages from: four waves x 64 lanes
constexpr int kMaxChunks = kFastSlots;             // slots of a tile's fixed-stride row (2 KiB per 31 KiB tile; until round 5 the stride was
                                                   // 352 slots of which 96 were never read: 0.75 KiB of plan memory per tile for nothing)
constexpr long long kTileSlow = INT64_MIN;         // tile_lo of a tile whose chunks do not fit its row

// A chunk (8 bytes): <= 64 consecutive positions of one run, inside one tile -- every field one scalar instruction
// away from where the issuing code needs it.
//   bits  0-31  byte offset of the chunk's lowest-ADDRESS element from cov + tile_lo, low half
//   bits 32-37  64 - positions (the low six bits of the high word: a 64-bit shift takes its count from there)
//   bit  38     the source index falls as the position rises ('-' strand piece)
//   bits 39-45  the byte offset's high half (units of 4 GiB; round 4): the pieces of one tile may lie anywhere
//               in a coverage of up to 512 GiB -- an index whose consecutive transcripts sit on different
//               chromosomes (gigabytes apart in the dense coverage) sent 93 % of its tiles down the scalar slow
//               path while the offset had 32 bits (profiles/archive/r04_fused_nested_before.json)
//   bits 48-63  LDS BYTE offset of its first position (the high word >> 16)
// The first kFastSlots slots of a row are what the fast path stages from, thread (wave w, lane i) holding slot 4 i + w:
// the FORWARD chunks fill them from slot 0 upwards, the '-' strand chunks from slot kFastSlots - 1 downwards, so that
// every wave issues its forward chunks from lane 0 up and its reverse chunks from lane 63 down, each with its own
// loop-invariant lane offsets.  Both regions are padded to whole blocks of four lanes with copies of a chunk of their
// direction (staging a chunk twice is harmless): the issuing code tests for the end every four steps.

struct PiecePlan {
    const unsigned long long *start;  // [n_pieces + 1] first profile position | kPieceNeg; sentinel total_nt
    const long long *base;            // [n_pieces + 1] coverage index = base + p, or base - p
    const long long *orf_piece;       // [n_orfs + 1]   pieces of ORF i: orf_piece[i] .. orf_piece[i+1]
    const long long *tile_piece0;     // [n_tiles]      piece holding the tile's first position
    const long long *tile_lo;         // [2 * n_tiles]  {lowest coverage index the tile reads - 64 (or kTileSlow), chunks}
    const chunk_desc_t *rows;         // [n_tiles][kMaxChunks]
    long long n_pieces;
    long long coverage_len;
};

inline size_t piece_plan_bytes(long long n_orfs, long long n_pieces, long long n_tiles)
{
    auto up = [](size_t b) { return (b + 127) & ~(size_t)127; };
    return up((size_t)(n_pieces + 1) * 8) * 2 + up((size_t)(n_orfs + 1) * 8) + up((size_t)n_tiles * 8) + up((size_t)n_tiles * 16) +
           (size_t)n_tiles * kMaxChunks * sizeof(chunk_desc_t);
}

struct PiecePlanMem {
    unsigned long long *start;
    long long *base;
    long long *orf_piece;
    long long *tile_piece0;
    long long *tile_lo;
    chunk_desc_t *rows;
};

inline PiecePlanMem carve_piece_plan(void *mem, long long n_orfs, long long n_pieces, long long n_tiles)
{
    auto up = [](size_t b) { return (b + 127) & ~(size_t)127; };
    PiecePlanMem m;
    char *p = reinterpret_cast<char *>(mem);
    m.start = reinterpret_cast<unsigned long long *>(p);
    p += up((size_t)(n_pieces + 1) * 8);
    m.base = reinterpret_cast<long long *>(p);
    p += up((size_t)(n_pieces + 1) * 8);
    m.orf_piece = reinterpret_cast<long long *>(p);
    p += up((size_t)(n_orfs + 1) * 8);
    m.tile_piece0 = reinterpret_cast<long long *>(p);
    p += up((size_t)n_tiles * 8);
    m.tile_lo = reinterpret_cast<long long *>(p);
    p += up((size_t)n_tiles * 16);
    m.rows = reinterpret_cast<chunk_desc_t *>(p);
    return m;
}

// One thread per ORF: its exon intervals (ascending, orf.py:100) -> pieces in profile order
// (the intervals backwards for a '-' strand ORF).  err bit 0: the intervals do not add up to
// the profile length; bit 1: an interval is empty or reaches outside the coverage array.
__global__ void k_piece_build(const int64_t *__restrict__ iv_start, const int32_t *__restrict__ iv_len,
                              const int64_t *__restrict__ orf_iv, const uint8_t *__restrict__ reverse,
                              const int64_t *__restrict__ offsets, long long n_orfs, long long n_pieces,
                              long long total_nt, long long coverage_len, PiecePlanMem out, int *__restrict__ err)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n_orfs) return;
    if (i == n_orfs) {  // sentinel
        out.start[n_pieces] = (unsigned long long)total_nt;
        out.base[n_pieces] = 0;
        out.orf_piece[n_orfs] = orf_iv[n_orfs];
        if (orf_iv[n_orfs] != n_pieces || orf_iv[0] != 0) atomicOr(err, 1);
        return;
    }
    const long long k0 = orf_iv[i];
    const long long nk = (long long)orf_iv[i + 1] - k0;
    out.orf_piece[i] = k0;
    if (nk < 0 || k0 < 0 || k0 + nk > n_pieces) {
        atomicOr(err, 1);
        return;
    }
    const long long beg = offsets[i];
    const long long len = (long long)offsets[i + 1] - beg;
    const bool rev = reverse[i] != 0;
    long long asc = 0;
    int bad = 0;
    for (long long t = 0; t < nk; ++t) {
        const long long s = iv_start[k0 + t];
        const long long n = iv_len[k0 + t];
        if (n <= 0 || s < 0 || s + n > coverage_len) bad |= 2;
        if (!rev) {
            const long long p0 = beg + asc;
            out.start[k0 + t] = (unsigned long long)p0;
            out.base[k0 + t] = s - p0;
        } else {
            const long long p0 = beg + (len - asc - n);
            out.start[k0 + nk - 1 - t] = (unsigned long long)p0 | kPieceNeg;
            out.base[k0 + nk - 1 - t] = s + n - 1 + p0;
        }
        asc += n;
    }
    if (asc != len) bad |= 1;
    if (bad) atomicOr(err, bad);
}

// A piece clipped to [t0, t_end): first position (LDS index), positions, source index of the first.
struct Clipped {
    long long src;
    int off, n;
    bool neg;
};

__device__ __forceinline__ Clipped clip_piece(unsigned long long start_word, unsigned long long next_word,
                                              long long base, long long t0, long long t_end)
{
    const long long s = (long long)(start_word & ~kPieceNeg);
    const long long e = (long long)(next_word & ~kPieceNeg);
    const bool neg = (start_word & kPieceNeg) != 0;
    const long long a = s > t0 ? s : t0;
    const long long b = e < t_end ? e : t_end;
    Clipped c;
    c.neg = neg;
    c.n = b > a ? (int)(b - a) : 0;
    c.off = (int)(a - t0);
    c.src = neg ? base - a : base + a;
    return c;
}

constexpr long long kMaxTileSpan = (1ll << 37) - 256;  // positions between a tile's lowest and highest source (7 + 32 bits of byte offset)

// Where a run is cut into chunks: at multiples of 64 ELEMENTS OF THE COVERAGE (256-byte lines of the source: the
// array is allocated on such a boundary), not at multiples of 64 positions of the run -- a chunk then asks the L1 for
// two whole cache lines instead of three partial ones, and no line is asked for by two chunks of one run (the fused
// kernel sent 1.8x the plain kernel's read requests to the L2: profiles/archive/r04_fused_ta_tcp_counters.txt).  The price is
// the run's first chunk: `head` positions up to the first boundary (0: the run starts on one).
#ifndef RP_CHUNK_ALIGN
#define RP_CHUNK_ALIGN 1
#endif
__device__ __forceinline__ int run_head(const Clipped &c)
{
#if RP_CHUNK_ALIGN
    if (c.n < 64) return 0;  // (one chunk either way; cut in two it would only fill the rows of short-exon layouts sooner)
    return c.neg ? (int)((c.src + 1) & 63) : (int)((64 - (c.src & 63)) & 63);
#else
    return 0;
#endif
}
__device__ __forceinline__ int run_chunks(const Clipped &c, int head) { return (head ? 1 : 0) + ((c.n - head + 63) >> 6); }

__device__ __forceinline__ chunk_desc_t make_chunk(const Clipped &c, long long tile_lo, int k, int head)
{
    const unsigned long long rel = (unsigned long long)(c.src - tile_lo);  // >= 64, < 2^37 (checked by the caller)
    int p, cnt;  // positions [p, p + cnt) of the run
    if (head && k == 0) {
        p = 0;
        cnt = head;
    } else {
        p = head + 64 * (k - (head ? 1 : 0));
        cnt = c.n - p < 64 ? c.n - p : 64;
    }
    // (the issuing code adds the offset to the tile's base in 64-bit scalar arithmetic and the lanes' own 0 ... 252 bytes
    // in the address unit: no carry is lost, whatever the low half holds)
    const unsigned long long soff = (c.neg ? rel - (unsigned long long)p - 63ull : rel + (unsigned long long)p) * 4ull;
    const unsigned hi = (64u - (unsigned)cnt) | ((c.neg ? 1u : 0u) << 6) | ((unsigned)(soff >> 32) << 7) | (((unsigned)(c.off + p) * 4u) << 16);
    return (chunk_desc_t)(unsigned)soff | ((chunk_desc_t)hi << 32);
}
__device__ __forceinline__ bool chunk_is_wide(chunk_desc_t cd) { return ((cd >> 39) & 0x7full) != 0; }

// One workgroup per tile: find the piece that holds the tile's first position, clip the
// tile's pieces to [t0, t0 + TILE + HALO), number their chunks and write the row.  A tile
// whose chunks do not fit the row's two lane regions (or with pieces > 512 GiB apart) is marked kTileSlow.
constexpr int kRowBlock = 256;

// Pieces that CONTINUE their predecessor -- the same base and direction: consecutive profile positions read consecutive
// coverage elements, which is what a compact coverage with one-position blocks makes of exons that face each other
// across an intron, and of consecutive ORFs on gapless layouts -- are staged as one RUN: the thread of the run's first
// piece leaves with the whole run, the threads of the others with nothing.  Fewer, fuller chunks (a chunk never spans
// two runs), and no partial cache line where two pieces meet.  Workgroup-uniform call; a run ends at the end of a batch
// of kRowBlock pieces.
// (Round 4 also built 16-byte-per-lane chunks for forward runs of >= 64 positions -- global_load_lds_dwordx4 does take a
// source and an LDS address that are only 4-byte aligned, scripts/probes/dma16_probe.hip -- with a second row region and
// a second issue loop: a fifth fewer requests, bit-identical results, and the fused kernel 10 % SLOWER on both bench
// layouts, whether the quad part started with the run, at a 16-byte aligned source or at a 16-byte aligned LDS address
// (profiles/archive/r04_ab_quad_chunks.txt).  Source and destination alignments differ piece by piece, so one of the two is
// always off; the dword chunks stay.)
__device__ __forceinline__ Clipped merge_runs(Clipped c, long long base, long long *s_base, int *s_n, int *s_cont, int t)
{
    __syncthreads();  // (the arrays of the previous batch are consumed)
    s_base[t] = base;
    s_n[t] = c.neg ? -c.n : c.n;
    __syncthreads();
    const bool cont = t > 0 && c.n > 0 && s_n[t - 1] != 0 && (s_n[t - 1] < 0) == c.neg && s_base[t - 1] == base;
    s_cont[t] = cont ? 1 : 0;
    __syncthreads();
    if (cont) {
        c.n = 0;
        return c;
    }
    if (c.n > 0)
        for (int k = t + 1; k < kRowBlock && s_cont[k]; ++k) c.n += s_n[k] < 0 ? -s_n[k] : s_n[k];
    return c;
}

template <int TILE, int HALO>
__global__ __launch_bounds__(kRowBlock) void k_chunk_rows(PiecePlanMem plan, long long n_pieces, long long total_nt)
{
    __shared__ long long s_j0, s_lo, s_hi;
    __shared__ int s_total, s_base, s_wide;
    __shared__ int s_wave[kRowBlock / 64];
    __shared__ long long s_run_base[kRowBlock];
    __shared__ int s_run_n[kRowBlock], s_run_cont[kRowBlock];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const long long b = blockIdx.x;
    const long long t0 = b * (long long)TILE;
    long long t_end = t0 + TILE + HALO;
    if (t_end > total_nt) t_end = total_nt;
    if (wave == 0) {  // last j with start_j <= t0 (start_0 == 0 <= t0)
        // 64-ary search by the wave: every step probes 64 evenly spaced pieces of [lo, hi) at once (one
        // memory latency per step, 4-5 steps for 23 M pieces; a binary search by one thread took 25
        // dependent loads and was most of this kernel's 3.6 ms at 500 000 tiles)
        long long lo = 0, hi = n_pieces;  // invariant: start[lo] <= t0 < start[hi] (sentinel: total_nt) unless t0 >= total_nt
        while (hi - lo > 1) {
            const long long span = hi - lo;
            const long long step = (span + 63) / 64;
            const long long probe = lo + step * (long long)lane;  // lane 0 probes lo itself: always <= t0
            const bool le = probe < hi && (long long)(plan.start[probe] & ~kPieceNeg) <= t0;
            const unsigned long long m = __ballot(le);
            const int top = 63 - __builtin_clzll(m | 1ull);  // the highest lane whose probe is <= t0
            const long long nlo = lo + step * (long long)top;
            long long nhi = nlo + step;
            if (nhi > hi) nhi = hi;
            lo = nlo;
            hi = nhi;
        }
        if (lane == 0) {
            s_j0 = lo;
            plan.tile_piece0[b] = lo;
            s_lo = INT64_MAX;
            s_hi = INT64_MIN;
            s_total = 0;
        }
    }
    __syncthreads();
    // pass 1: extent and chunk count of ALL the tile's pieces (kRowBlock at a time)
    for (long long j0 = s_j0;; j0 += kRowBlock) {  // workgroup-uniform
        const long long j = j0 + t;
        Clipped c{0, 0, 0, false};
        const long long base = j < n_pieces ? plan.base[j] : 0;
        if (j < n_pieces) c = clip_piece(plan.start[j], plan.start[j + 1], base, t0, t_end);
        c = merge_runs(c, base, s_run_base, s_run_n, s_run_cont, t);
        if (c.n > 0) {
            atomicMin(&s_lo, c.neg ? c.src - (c.n - 1) : c.src);
            atomicMax(&s_hi, c.neg ? c.src : c.src + c.n - 1);
            atomicAdd(&s_total, run_chunks(c, run_head(c)) << (c.neg ? 16 : 0));  // forward | reverse << 16
        }
        const long long last = j0 + kRowBlock;  // does the tile go on past this batch?
        if (!(last < n_pieces && (long long)(plan.start[last] & ~kPieceNeg) < t_end)) break;
    }
    __syncthreads();
    const int n_fwd = s_total & 0xffff, n_rev = s_total >> 16;  // (a tile owns < 8 000 positions: both fit 16 bits)
    const int fwd_lanes = ((n_fwd + 3) / 4 + 3) / 4 * 4;  // lanes 0 ... of every wave that hold forward chunks or their padding
    const int rev_lanes = ((n_rev + 3) / 4 + 3) / 4 * 4;  // lanes 63 ... downwards: the '-' strand ones
    const bool fast = s_total > 0 && fwd_lanes + rev_lanes <= 64 && s_hi - s_lo < kMaxTileSpan;
    const long long tile_lo = fast ? s_lo - 64 : kTileSlow;
    if (t == 0) {
        plan.tile_lo[2 * b] = tile_lo;
        plan.tile_lo[2 * b + 1] = (long long)(unsigned)s_total;
        s_base = 0;
        s_wide = 0;
    }
    if (!fast) return;  // (the row stays unwritten: never read)
    // pass 2: the chunks, numbered by a prefix sum over the pieces
    chunk_desc_t *row = plan.rows + b * kMaxChunks;
    for (long long j0 = s_j0;; j0 += kRowBlock) {
        const long long j = j0 + t;
        Clipped c{0, 0, 0, false};
        const long long base = j < n_pieces ? plan.base[j] : 0;
        if (j < n_pieces) c = clip_piece(plan.start[j], plan.start[j + 1], base, t0, t_end);
        c = merge_runs(c, base, s_run_base, s_run_n, s_run_cont, t);
        const int head = c.n > 0 ? run_head(c) : 0;
        const int mine = c.n > 0 ? run_chunks(c, head) : 0;
        const int nch = mine << (c.neg ? 16 : 0);  // (both prefix sums in one scan)
        int incl = nch;
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, true);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, true);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xf, 0xf, true);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xf, 0xf, true);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xa, 0xf, false);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xc, 0xf, false);
        __syncthreads();  // (s_wave / s_base of the previous batch are consumed)
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        int before = s_base + incl - nch;
        for (int w = 0; w < wave; ++w) before += s_wave[w];
        for (int k = 0; k < mine; ++k) {
            const chunk_desc_t cd = make_chunk(c, tile_lo, k, head);
            row[c.neg ? kFastSlots - 1 - ((before >> 16) + k) : (before & 0xffff) + k] = cd;
            if (chunk_is_wide(cd)) s_wide = 1;  // (benign race: every writer stores 1)
        }
        __syncthreads();
        if (t == 0) s_base += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        const long long last = j0 + kRowBlock;
        if (!(last < n_pieces && (long long)(plan.start[last] & ~kPieceNeg) < t_end)) break;
    }
    __syncthreads();
    if (s_wide && t == 0) plan.tile_lo[2 * b + 1] = (long long)(unsigned)s_total | (1ll << 32);  // (kTileWide, defined below with its reader)
    // pad: the slots up to the end of a direction's last block of four lanes repeat that direction's first chunk
    __threadfence_block();
    if (n_fwd > 0) {
        const chunk_desc_t first = row[0];
        for (int c = n_fwd + t; c < 4 * fwd_lanes; c += kRowBlock) row[c] = first;
    }
    if (n_rev > 0) {
        const chunk_desc_t first = row[kFastSlots - 1];
        for (int c = n_rev + t; c < 4 * rev_lanes; c += kRowBlock) row[kFastSlots - 1 - c] = first;
    }
}

// ---------------------------------------------------------------------------------------
// Staging a tile, the fast path.  Thread (wave w, lane i) holds slot 4 i + w of the row in two registers (loaded with
// the head row: loads that depend on blockIdx only); the wave then issues its FORWARD chunks from lane 0 upwards and
// its '-' strand chunks from lane 63 downwards, straight-line code with an exit test every four steps.  A step is two
// lane reads, EXEC from the high word's low six bits (the ragged end), M0 from its high half (the LDS address), a
// 64-bit scalar add of the chunk's offset to the tile's base, and one global_load_lds whose vector operand is
// loop-invariant: lane * 4 going up, (63 - lane) * 4 for a '-' strand chunk.  Two vector instructions per chunk, where
// rounds 2-4 spent four (the lane's offset computed per step): the scorer runs at 80 % VALU occupancy and the fused
// kernel at 87 %, and four MORE per step cost it 9 % (profiles/archive/r04_ab_scalar_issue.txt; issuing through the scalar unit
// alone -- descriptors by s_load_dwordx16 -- lost what it gained to the scalar loads' latency, which cannot be overlapped
// beyond one group: scalar loads return out of order, the only wait is for all of them).  (A scalar loop over pieces,
// with its decode and branch chain, was measured at ~1 000 cycles per piece next to three other workgroups' lane runs.)
// Nothing is waited for here.
// ---------------------------------------------------------------------------------------
#ifndef RP_CHUNK_POLICY
#define RP_CHUNK_POLICY " nt"  // the coverage is read once per launch: tile gather -5 % on gapped / 60-nt layouts, else unchanged
#endif
// NARROW step (all chunks of the tile within 4 GiB of its lowest source: every chromosome-sorted index); WIDE step
// (pieces of one tile anywhere in 512 GiB): the high offset half joins the carry.  The base of a step is s[20:21]
// (named registers: an asm operand cannot be addressed by halves; both are on the clobber list).
#define RP_DMA_STEP_NARROW(I)                                   \
    "v_readlane_b32 %[so], %[w0], " #I "\n\t"                   \
    "v_readlane_b32 %[s1], %[w1], " #I "\n\t"                   \
    "s_lshr_b64 exec, -1, %[s1]\n\t"                            \
    "s_lshr_b32 m0, %[s1], 16\n\t"                              \
    "s_add_u32 s20, %[blo], %[so]\n\t"                          \
    "s_addc_u32 s21, %[bhi], 0\n\t"                             \
    "global_load_lds_dword %[voff], s[20:21]" RP_CHUNK_POLICY "\n\t"
#define RP_DMA_STEP_WIDE(I)                                     \
    "v_readlane_b32 %[so], %[w0], " #I "\n\t"                   \
    "v_readlane_b32 %[s1], %[w1], " #I "\n\t"                   \
    "s_lshr_b64 exec, -1, %[s1]\n\t"                            \
    "s_lshr_b32 m0, %[s1], 16\n\t"                              \
    "s_bfe_u32 %[st], %[s1], 0x70007\n\t"                       \
    "s_add_u32 s20, %[blo], %[so]\n\t"                          \
    "s_addc_u32 s21, %[bhi], %[st]\n\t"                         \
    "global_load_lds_dword %[voff], s[20:21]" RP_CHUNK_POLICY "\n\t"
#define RP_DMA_STEP4(S, A, B, C, D, LIM)                        \
    S(A) S(B) S(C) S(D)                                         \
    "s_cmp_le_u32 %[steps], " #LIM "\n\t"                       \
    "s_cbranch_scc1 1f\n\t"
#define RP_DMA_STEPS_UP(S)                                                                                      \
    RP_DMA_STEP4(S, 0, 1, 2, 3, 4) RP_DMA_STEP4(S, 4, 5, 6, 7, 8) RP_DMA_STEP4(S, 8, 9, 10, 11, 12)             \
    RP_DMA_STEP4(S, 12, 13, 14, 15, 16) RP_DMA_STEP4(S, 16, 17, 18, 19, 20) RP_DMA_STEP4(S, 20, 21, 22, 23, 24) \
    RP_DMA_STEP4(S, 24, 25, 26, 27, 28) RP_DMA_STEP4(S, 28, 29, 30, 31, 32) RP_DMA_STEP4(S, 32, 33, 34, 35, 36) \
    RP_DMA_STEP4(S, 36, 37, 38, 39, 40) RP_DMA_STEP4(S, 40, 41, 42, 43, 44) RP_DMA_STEP4(S, 44, 45, 46, 47, 48) \
    RP_DMA_STEP4(S, 48, 49, 50, 51, 52) RP_DMA_STEP4(S, 52, 53, 54, 55, 56) RP_DMA_STEP4(S, 56, 57, 58, 59, 60) \
    RP_DMA_STEP4(S, 60, 61, 62, 63, 64)
#define RP_DMA_STEPS_DOWN(S)                                                                                    \
    RP_DMA_STEP4(S, 63, 62, 61, 60, 4) RP_DMA_STEP4(S, 59, 58, 57, 56, 8) RP_DMA_STEP4(S, 55, 54, 53, 52, 12)   \
    RP_DMA_STEP4(S, 51, 50, 49, 48, 16) RP_DMA_STEP4(S, 47, 46, 45, 44, 20) RP_DMA_STEP4(S, 43, 42, 41, 40, 24) \
    RP_DMA_STEP4(S, 39, 38, 37, 36, 28) RP_DMA_STEP4(S, 35, 34, 33, 32, 32) RP_DMA_STEP4(S, 31, 30, 29, 28, 36) \
    RP_DMA_STEP4(S, 27, 26, 25, 24, 40) RP_DMA_STEP4(S, 23, 22, 21, 20, 44) RP_DMA_STEP4(S, 19, 18, 17, 16, 48) \
    RP_DMA_STEP4(S, 15, 14, 13, 12, 52) RP_DMA_STEP4(S, 11, 10, 9, 8, 56) RP_DMA_STEP4(S, 7, 6, 5, 4, 60)       \
    RP_DMA_STEP4(S, 3, 2, 1, 0, 64)

// Call sites must be wave-uniform with all 64 lanes active: the block overwrites EXEC and leaves
// it all-ones (stage_tile_chunks is reached through workgroup-uniform branches only).  M0 and EXEC
// are on the clobber list so that no M0 value (the compiler's own LDS-DMA set-up, merged M0
// initialisations) and no EXEC-dependent state is carried across the block; clang notes that both
// are reserved registers, which is the point.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
// UP: the wave's first `steps` lanes, with lane offsets voff = lane * 4; !UP: its last `steps` lanes, voff = (63 - lane) * 4
template <bool WIDE, bool UP>
__device__ __forceinline__ void issue_chunks(unsigned blo, unsigned bhi, unsigned w0, unsigned w1, int steps, int voff)
{
    unsigned so, s1, st;
#define RP_ISSUE(STEPS, STEP)                                                                                                  \
    asm volatile(STEPS(STEP) "1:\n\ts_mov_b64 exec, -1"                                                                       \
                 : [so] "=&s"(so), [s1] "=&s"(s1), [st] "=&s"(st)                                                              \
                 : [w0] "v"(w0), [w1] "v"(w1), [voff] "v"(voff), [blo] "s"(blo), [bhi] "s"(bhi), [steps] "s"(steps)            \
                 : "memory", "scc", "m0", "exec", "s20", "s21")  // every step rewrites M0 and EXEC (EXEC is left all-ones, as it came in)
    if constexpr (WIDE && UP)
        RP_ISSUE(RP_DMA_STEPS_UP, RP_DMA_STEP_WIDE);
    else if constexpr (WIDE)
        RP_ISSUE(RP_DMA_STEPS_DOWN, RP_DMA_STEP_WIDE);
    else if constexpr (UP)
        RP_ISSUE(RP_DMA_STEPS_UP, RP_DMA_STEP_NARROW);
    else
        RP_ISSUE(RP_DMA_STEPS_DOWN, RP_DMA_STEP_NARROW);
#undef RP_ISSUE
}
#pragma clang diagnostic pop
#undef RP_DMA_STEPS_DOWN
#undef RP_DMA_STEPS_UP
#undef RP_DMA_STEP4
#undef RP_DMA_STEP_NARROW
#undef RP_DMA_STEP_WIDE

// pin a workgroup-uniform pointer to scalar registers (an "s" asm operand alone does not)
__device__ __forceinline__ const int32_t *scalar_ptr(const int32_t *p)
{
    const unsigned long long u = (unsigned long long)p;
    return reinterpret_cast<const int32_t *>(
        ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32)) << 32) |
        (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u));
}

__device__ __forceinline__ unsigned lds_address(const int *p)
{
    return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const int *)p;
}

// the row slot that this thread keeps: chunk 4 * lane + wave
__device__ __forceinline__ const chunk_desc_t *chunk_slot(const PiecePlan &pp, long long b, int lane, int wave)
{
    return pp.rows + b * kMaxChunks + 4 * lane + wave;
}

// `e`: the thread's slot (already loaded); total_word = tile_lo[2 b + 1]: forward chunks | '-' strand chunks << 16,
// with kTileWide set when a chunk's offset does not fit 32 bits
constexpr long long kTileWide = 1ll << 32;

__device__ __forceinline__ void stage_tile_chunks(const int32_t *__restrict__ cov, const PiecePlan &pp, long long b,
                                                  long long tile_lo, long long total_word, chunk_desc_t e, int *s_counts, int lane, int wave)
{
    const int n_fwd = (int)(total_word & 0xffff), n_rev = (int)((total_word >> 16) & 0xffff);
    const bool wide = (total_word & kTileWide) != 0;  // workgroup-uniform
    const unsigned long long base0 = (unsigned long long)scalar_ptr(cov + tile_lo);
    const unsigned blo = (unsigned)base0, bhi = (unsigned)(base0 >> 32);
    const unsigned w0 = (unsigned)e;
    const unsigned w1 = (unsigned)(e >> 32) + (lds_address(s_counts) << 16);  // (the LDS byte offset becomes the LDS address)
    // forward chunks wave, wave + 4, ... < n_fwd in lanes 0 ...; reverse chunk j in slot 255 - j: wave 3 - j % 4, lane 63 - j / 4
    const int fsteps = __builtin_amdgcn_readfirstlane((n_fwd - wave + 3) >> 2);
    const int rsteps = __builtin_amdgcn_readfirstlane((n_rev + wave) >> 2);
    if (fsteps > 0) {
        if (wide)
            issue_chunks<true, true>(blo, bhi, w0, w1, fsteps, lane * 4);
        else
            issue_chunks<false, true>(blo, bhi, w0, w1, fsteps, lane * 4);
    }
    if (rsteps > 0) {
        if (wide)
            issue_chunks<true, false>(blo, bhi, w0, w1, rsteps, (63 - lane) * 4);
        else
            issue_chunks<false, false>(blo, bhi, w0, w1, rsteps, (63 - lane) * 4);
    }
}

// ---------------------------------------------------------------------------------------
// The slow path (a tile with more chunks than a row holds: runs of very short exons, or
// pieces > 4 GiB apart): the pieces straight from the global list, 64 at a time, one lane
// each; the wave then issues every piece's chunks from a scalar loop.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void stage_piece_slow(const int32_t *__restrict__ cov, long long src, int off, int n, bool neg,
                                                 int *s_counts, int lane)
{
    typedef const __attribute__((address_space(1))) void *gptr_t;
    typedef __attribute__((address_space(3))) void *lptr_t;
    for (int c0 = 0; c0 < n; c0 += 64) {  // wave-uniform
        if (lane < n - c0) {
            const int32_t *g = neg ? cov + (src - c0 - lane) : cov + (src + c0 + lane);
            __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(s_counts + off + c0), 4, 0, 0);
        }
    }
}

template <int TILE, int HALO>
__device__ __forceinline__ void stage_tile_slow(const int32_t *__restrict__ cov, const PiecePlan &pp, long long b,
                                                long long total_nt, int *s_counts, int lane, int wave)
{
    const long long t0 = b * (long long)TILE;
    long long t_end = t0 + TILE + HALO;
    if (t_end > total_nt) t_end = total_nt;
    long long j = pp.tile_piece0[b];
    for (;;) {  // wave-uniform: every wave walks the whole list and takes the pieces of lanes wave, wave + 4, ...
        Clipped c{0, 0, 0, false};
        if (j + lane < pp.n_pieces) c = clip_piece(pp.start[j + lane], pp.start[j + lane + 1], pp.base[j + lane], t0, t_end);
        for (int l = wave; l < 64; l += 4) {
            const int n = __builtin_amdgcn_readlane(c.n, l);
            if (n == 0) continue;
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(unsigned long long)c.src, l);
            const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)c.src >> 32), l);
            stage_piece_slow(cov, (long long)(((unsigned long long)hi << 32) | lo), __builtin_amdgcn_readlane(c.off, l), n,
                             __builtin_amdgcn_readlane((int)c.neg, l) != 0, s_counts, lane);
        }
        j += 64;
        if (!(j < pp.n_pieces && (long long)(pp.start[j] & ~kPieceNeg) < t_end)) break;  // the list ran past the tile
    }
}

// An ORF's profile read through its pieces (float64 re-walks, tie replays: ~0.4 % of the ORFs).
struct PieceView {
    const int32_t *cov;
    const unsigned long long *start;
    const long long *base;
    long long j0, j1, beg;
    __device__ __forceinline__ int operator[](long long a) const
    {
        const long long p = beg + a;
        long long j = j0;
        while (j + 1 < j1 && (long long)(start[j + 1] & ~kPieceNeg) <= p) ++j;
        const long long idx = (start[j] & kPieceNeg) ? base[j] - p : base[j] + p;
        return cov[idx];
    }
};

// What the finish kernels read an ORF's counts from: the CSR array, or the coverage through the plan.
struct CsrSource {
    const int32_t *counts;
    __device__ __forceinline__ const int32_t *orf(long long i, long long beg) const { return counts + beg; }
};

struct CoverageSource {
    const int32_t *cov;
    PiecePlan pp;
    __device__ __forceinline__ PieceView orf(long long i, long long beg) const
    {
        return PieceView{cov, pp.start, pp.base, pp.orf_piece[i], pp.orf_piece[i + 1], beg};
    }
};

// ---------------------------------------------------------------------------------------
// The profiles of a SUBSET of the ORFs (default mode prints the translating ORFs only, detect_orfs.py:301-303:
// ~14-23 % of an index) straight through the plan's pieces: one wave per chosen ORF copies its pieces, coalesced,
// '-' strand pieces backwards, to out[out_off[w] ...].  Nothing of the interval table is needed on the host (round 3
// built a sub-table of the chosen ORFs with numpy and uploaded it: 0.1 s of a 0.8 s export).
// ---------------------------------------------------------------------------------------
constexpr int kSelectedBlock = 256;

__global__ __launch_bounds__(kSelectedBlock) void k_gather_selected(const int32_t *__restrict__ cov, PiecePlan pp,
                                                                    const long long *__restrict__ chosen, long long n_chosen,
                                                                    const long long *__restrict__ out_off, int32_t *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const long long w = (long long)blockIdx.x * (kSelectedBlock / 64) + (threadIdx.x >> 6);
    if (w >= n_chosen) return;  // (wave-uniform)
    const long long orf = chosen[w];
    const long long j0 = pp.orf_piece[orf], j1 = pp.orf_piece[orf + 1];
    if (j1 <= j0) return;
    const long long beg = (long long)(pp.start[j0] & ~kPieceNeg);  // the ORF's first profile position
    int32_t *dst = out + out_off[w] - beg;
    for (long long j = j0; j < j1; ++j) {  // wave-uniform
        const unsigned long long sw = pp.start[j];
        const long long s = (long long)(sw & ~kPieceNeg);
        const long long e = (long long)(pp.start[j + 1] & ~kPieceNeg);  // (the next piece of the profile space, or the sentinel)
        const long long base = pp.base[j];
        const bool neg = (sw & kPieceNeg) != 0;
        for (long long pos = s + lane; pos < e; pos += 64) dst[pos] = cov[neg ? base - pos : base + pos];
    }
}

// ---------------------------------------------------------------------------------------
// The CSR `counts` array of a whole index (report_all mode, detect_orfs.py:301-324 prints
// every profile): a workgroup stages its tile exactly as the fused scorer does and writes
// it out, 16 bytes per lane.  8 bytes of traffic per nucleotide, the coverage read once.
// ---------------------------------------------------------------------------------------
constexpr int kGatherTileBlock = 256;

template <int TILE, int HALO>
__global__ __launch_bounds__(kGatherTileBlock) void k_tile_gather(const int32_t *__restrict__ cov, PiecePlan pp,
                                                                  long long total_nt, int32_t *__restrict__ counts)
{
    typedef int i32x4_t __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) int s_counts[TILE + HALO];  // the rows are clipped for the scorer: halo included
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long b = blockIdx.x;
    const long long tile_lo = pp.tile_lo[2 * b];
    if (tile_lo != kTileSlow)
        stage_tile_chunks(cov, pp, b, tile_lo, pp.tile_lo[2 * b + 1], *chunk_slot(pp, b, lane, wave), s_counts, lane, wave);
    else
        stage_tile_slow<TILE, HALO>(cov, pp, b, total_nt, s_counts, lane, wave);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const long long t0 = b * (long long)TILE;
    const long long left = total_nt - t0;
    if (left >= TILE) {
#pragma unroll
        for (int c = tid; c < TILE / 4; c += kGatherTileBlock)
            stream_store(reinterpret_cast<i32x4_t *>(counts + t0) + c, reinterpret_cast<const i32x4_t *>(s_counts)[c]);
    } else {
        for (int c = tid; c < (int)left; c += kGatherTileBlock) stream_store(counts + t0 + c, (int32_t)s_counts[c]);
    }
}

}  // namespace rp
