// rp_wave.hpp -- "one wavefront per ORF" kernels, streaming straight from HBM.
//
// Lane t owns triplets t, t+64, ... of the ORF: positions 3j..3j+2 plus a two
// value halo, i.e. exactly one codon of each reading frame per step
// (statistics.py:67-91 walks the three frames one after the other; here they
// share one read of the profile).  Used as
//   * the simple scoring path (RP_ALGO_WAVE),
//   * the float64 re-walk of too-close-to-call ORFs in k_orf_finish (rp_tile.hpp),
//   * the per-frame diagnostics and float64-profile entry points.
#pragma once

#include "rp_device.hpp"

namespace rp {

// Per-lane partial results of one ORF walk.
template <typename Real>
struct WalkResult {
    FrameAcc<Real> acc[3];
    long long count;  // sum of the codon sums this lane saw
    int min_codon;    // min of those codon sums (frame-0 codons incl. the partial last one)
};

// Walk ORF [v, v+len): this thread takes triplets first, first + stride, ...  (a wave
// passes lane / 64, a whole workgroup tid / 256); int32 counts.
template <typename Real>
__device__ __forceinline__ void wave_walk(const int32_t *v, long long len, int first,
                                          WalkResult<Real> &w, int stride = kWave)
{
    acc_clear(w.acc);
    w.count = 0;
    w.min_codon = RP_MIN_CODON_COV_EMPTY;
    const long long n_trip = (len + 2) / 3;  // ceil(len/3): common.py:164-180 codon count
    for (long long j = first; j < n_trip; j += stride) {
        const long long p = 3 * j;
        const long long rem = len - p;  // >= 1
        // five back-to-back loads at clamped (always in-range) indices, masked afterwards:
        // a select per value instead of a branch + wait per load
        const long long last = len - 1;
        const int l0 = v[p];
        const int l1 = v[p + 1 < last ? p + 1 : last];
        const int l2 = v[p + 2 < last ? p + 2 : last];
        const int l3 = v[p + 3 < last ? p + 3 : last];
        const int l4 = v[p + 4 < last ? p + 4 : last];
        const int v0 = l0;
        const int v1 = rem > 1 ? l1 : 0;
        const int v2 = rem > 2 ? l2 : 0;
        const int v3 = rem > 3 ? l3 : 0;
        const int v4 = rem > 4 ? l4 : 0;
        const int codon = v0 + v1 + v2;
        w.count += codon;
        w.min_codon = min(w.min_codon, codon);
        codon_add(w.acc[0], v0, v1, v2, rem > 2);
        codon_add(w.acc[1], v1, v2, v3, rem > 3);
        codon_add(w.acc[2], v2, v3, v4, rem > 4);
    }
}

// Reduce a walk across the wave and score the three frames (all lanes get the result).
template <typename Real>
__device__ __forceinline__ void wave_reduce_frames(const WalkResult<Real> &w, FrameScore (&fr)[3],
                                                   long long &count, int &min_codon)
{
#pragma unroll
    for (int f = 0; f < 3; ++f) {
        const double p = wave_sum((double)w.acc[f].p);
        const double q = wave_sum((double)w.acc[f].q);
        const int n = wave_sum(w.acc[f].n);
        const int m = wave_sum(w.acc[f].m);
        fr[f] = frame_score(p, q, n, m);
    }
    count = wave_sum(w.count);
    min_codon = wave_min(w.min_codon);
}

// Full scoring of one ORF by one wave.  fp32 lane partials for short profiles with a
// float64 re-walk when the frame decision is too close to call; float64 throughout
// for long profiles (per-lane fp32 sums would grow past the recheck margin).
constexpr long long kWaveFp32MaxLen = 16384;

__device__ __forceinline__ void wave_score_orf(const int32_t *__restrict__ v, long long len, int lane,
                                               double &phase, int &valid, long long &count,
                                               int &min_codon, unsigned &flags)
{
    FrameScore fr[3];
    unsigned extra = 0;
    bool need64 = len > kWaveFp32MaxLen;
    if (!need64) {
        WalkResult<float> w;
        wave_walk<float>(v, len, lane, w);
        wave_reduce_frames(w, fr, count, min_codon);
        need64 = fp32_decision_unsafe(fr);  // wave-uniform
    }
    if (need64) {
        WalkResult<double> w;
        wave_walk<double>(v, len, lane, w);
        wave_reduce_frames(w, fr, count, min_codon);
        extra = RP_FLAG_RECHECK64;
    }
    combine_frames(fr, phase, valid, flags);
    flags |= extra;
}

// ---------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------
constexpr int kWaveBlock = 256;  // 4 waves per workgroup

__global__ __launch_bounds__(kWaveBlock) void k_wave_score(const int32_t *__restrict__ counts,
                                                           const int64_t *__restrict__ offsets,
                                                           long long n_orfs, OrfOutputs out,
                                                           FilterParams fp)
{
    const int lane = threadIdx.x & (kWave - 1);
    const long long waves_total = (long long)gridDim.x * (kWaveBlock / kWave);
    long long orf = (long long)blockIdx.x * (kWaveBlock / kWave) + (threadIdx.x >> 6);
    for (; orf < n_orfs; orf += waves_total) {
        const long long beg = offsets[orf];
        const long long len = offsets[orf + 1] - beg;
        double phase;
        int valid, min_codon;
        long long count;
        unsigned flags;
        wave_score_orf(counts + beg, len, lane, phase, valid, count, min_codon, flags);
        if (lane == 0) store_orf(out, fp, orf, phase, valid, count, min_codon, flags, len);
    }
}

// per-frame float64 diagnostics
__global__ __launch_bounds__(kWaveBlock) void k_wave_frames(const int32_t *__restrict__ counts,
                                                            const int64_t *__restrict__ offsets,
                                                            long long n_orfs,
                                                            double *__restrict__ frame_score_out,
                                                            int32_t *__restrict__ frame_n,
                                                            int32_t *__restrict__ frame_m)
{
    const int lane = threadIdx.x & (kWave - 1);
    const long long waves_total = (long long)gridDim.x * (kWaveBlock / kWave);
    long long orf = (long long)blockIdx.x * (kWaveBlock / kWave) + (threadIdx.x >> 6);
    for (; orf < n_orfs; orf += waves_total) {
        const long long beg = offsets[orf];
        const long long len = offsets[orf + 1] - beg;
        WalkResult<double> w;
        wave_walk<double>(counts + beg, len, lane, w);
        FrameScore fr[3];
        long long count;
        int min_codon;
        wave_reduce_frames(w, fr, count, min_codon);
        if (lane < 3) {
            const FrameScore r = lane == 0 ? fr[0] : (lane == 1 ? fr[1] : fr[2]);
            frame_score_out[3 * orf + lane] = r.score;
            frame_n[3 * orf + lane] = r.n;
            frame_m[3 * orf + lane] = r.m;
        }
    }
}

// float64-valued profiles (metagene.py:243-244 -> statistics.py:48)
__global__ __launch_bounds__(kWaveBlock) void k_wave_score_f64in(const double *__restrict__ values,
                                                                 const int64_t *__restrict__ offsets,
                                                                 long long n_profiles,
                                                                 double *__restrict__ phase_out,
                                                                 int32_t *__restrict__ valid_out,
                                                                 uint8_t *__restrict__ flags_out)
{
    const int lane = threadIdx.x & (kWave - 1);
    const long long waves_total = (long long)gridDim.x * (kWaveBlock / kWave);
    long long prof = (long long)blockIdx.x * (kWaveBlock / kWave) + (threadIdx.x >> 6);
    for (; prof < n_profiles; prof += waves_total) {
        const long long beg = offsets[prof];
        const long long len = offsets[prof + 1] - beg;
        const double *v = values + beg;
        FrameAcc<double> acc[3];
        acc_clear(acc);
        for (long long p = 3LL * lane; p + 2 < len; p += 3LL * kWave) {
            const long long rem = len - p;
            const double v0 = v[p], v1 = v[p + 1], v2 = v[p + 2];
            const double v3 = rem > 3 ? v[p + 3] : 0.0;
            const double v4 = rem > 4 ? v[p + 4] : 0.0;
            codon_add_f64in(acc[0], v0, v1, v2, true);
            codon_add_f64in(acc[1], v1, v2, v3, rem > 3);
            codon_add_f64in(acc[2], v2, v3, v4, rem > 4);
        }
        FrameScore fr[3];
#pragma unroll
        for (int f = 0; f < 3; ++f)
            fr[f] = frame_score(wave_sum(acc[f].p), wave_sum(acc[f].q), wave_sum(acc[f].n),
                                wave_sum(acc[f].m));
        double phase;
        int valid;
        unsigned flags;
        combine_frames(fr, phase, valid, flags);
        if (lane == 0) {
            phase_out[prof] = phase;
            valid_out[prof] = valid;
            flags_out[prof] = (uint8_t)flags;
        }
    }
}

// Profile gather (detect_orfs.py:134-203 for all ORFs): one wave per ORF copies its exon
// intervals out of the dense coverage array, coalesced on both sides; '-' strand ORFs are
// written back to front (detect_orfs.py:201-202).
__global__ __launch_bounds__(kWaveBlock) void k_gather_profiles(
    const int32_t *__restrict__ coverage, long long coverage_len, const int64_t *__restrict__ iv_start,
    const int32_t *__restrict__ iv_len, const int64_t *__restrict__ orf_iv,
    const uint8_t *__restrict__ reverse, const int64_t *__restrict__ offsets, long long n_orfs,
    int32_t *__restrict__ counts)
{
    const int lane = threadIdx.x & (kWave - 1);
    const long long waves_total = (long long)gridDim.x * (kWaveBlock / kWave);
    long long orf = (long long)blockIdx.x * (kWaveBlock / kWave) + (threadIdx.x >> 6);
    for (; orf < n_orfs; orf += waves_total) {
        const long long out0 = offsets[orf];
        const long long len = (long long)offsets[orf + 1] - out0;
        const bool rev = reverse[orf] != 0;
        long long asc = 0;  // ascending position of the interval's first nucleotide in the ORF
        for (long long k = orf_iv[orf]; k < (long long)orf_iv[orf + 1]; ++k) {
            const long long src0 = iv_start[k];
            const int n = iv_len[k];
            for (int j = lane; j < n; j += kWave) {
                const long long src = src0 + j;
                const int v = (src >= 0 && src < coverage_len) ? coverage[src] : 0;
                const long long a = asc + j;
                if (a < len) counts[out0 + (rev ? len - 1 - a : a)] = v;
            }
            asc += n;
        }
    }
}

// Input validation: offsets[0] == 0, monotone, offsets[n] == total; 0 <= count <= RP_MAX_COUNT.
// err[0] |= 1 for offsets, |= 2 for counts.
__global__ void k_validate(const int32_t *__restrict__ counts, const int64_t *__restrict__ offsets,
                           long long n_orfs, long long total_nt, int *__restrict__ err)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int bad = 0;
    for (long long i = tid; i <= n_orfs; i += stride) {
        const long long o = offsets[i];
        if (i == 0 && o != 0) bad |= 1;
        if (i == n_orfs && o != total_nt) bad |= 1;
        if (i < n_orfs && offsets[i + 1] < o) bad |= 1;
    }
    for (long long k = tid; k < total_nt; k += stride) {
        const int c = counts[k];
        if (c < 0 || c > RP_MAX_COUNT) bad |= 2;
    }
    if (bad) atomicOr(err, bad);
}

}  // namespace rp
