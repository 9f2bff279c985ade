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
                                      int32_t *__restrict__ coverage, long long coverage_len, int *__restrict__ err)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) {
        const int s = strand[k], c = chrom[k];
        if (s > 1 || c < 0 || c >= n_chroms) continue;
        const int g = lut[s * n_chroms + c];
        if (g < 0 || g >= n_groups) continue;
        const long long p = pos[k];
        if (p < group_lo[g] || p > group_hi[g]) continue;
        const long long idx = group_start[g] + (p - group_lo[g]);
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
