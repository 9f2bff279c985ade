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
// (`v` is anything indexable by ORF position: the CSR pointer, or a view of the coverage.)
template <typename Real, typename Counts>
__device__ __forceinline__ void wave_walk(Counts v, long long len, int first,
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
                                               const FilterParams &fp, double &phase, int &valid,
                                               long long &count, int &min_codon, unsigned &flags, ReplayLds *replay_lds)
{
    FrameScore fr[3];
    unsigned extra = 0;
    bool need64 = len > kWaveFp32MaxLen;
    if (!need64) {
        WalkResult<float> w;
        wave_walk<float>(v, len, lane, w);
        wave_reduce_frames(w, fr, count, min_codon);
        need64 = fp32_decision_unsafe(fr);  // wave-uniform
        if (!need64 && fp.enabled) {  // status must not hinge on fp32 rounding next to the cutoff
            combine_frames(fr, phase, valid, flags);
            need64 = near_cutoff(fp, phase);
        }
    }
    if (need64) {
        WalkResult<double> w;
        wave_walk<double>(v, len, lane, w);
        wave_reduce_frames(w, fr, count, min_codon);
        extra = RP_FLAG_RECHECK64;
    }
    combine_frames(fr, phase, valid, flags);
    flags |= extra;
    if (flags & RP_FLAG_TIE) {  // an exact frame tie: the reference's own bits decide (wave-uniform)
        const bool big = replay_tie_wave(v, len, lane, phase, valid, replay_lds);
        flags |= RP_FLAG_REPLAY | (big ? RP_FLAG_BIGTIE : 0u);
    }
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
    __shared__ ReplayLds s_replay[kWaveBlock / kWave];  // one per wave
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
        wave_score_orf(counts + beg, len, lane, fp, phase, valid, count, min_codon, flags, &s_replay[threadIdx.x >> 6]);
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

// Profile gather (detect_orfs.py:134-203 for all ORFs): the exon intervals of every ORF are
// copied out of the dense coverage array into the CSR counts array, coalesced on both
// sides; '-' strand ORFs are written back to front (detect_orfs.py:201-202).
//
// A wave takes a batch of 64 consecutive ORFs.  The per-ORF metadata (offsets, interval
// range, strand) is loaded lane-parallel, one coalesced load per array for the whole batch,
// and handed out with v_readlane; the interval descriptors of ORF j + 1 are fetched (lane t
// = interval t) while ORF j is being copied.  An ORF is copied kGatherUnroll * 64 nucleotides
// per round: a wave-uniform walk over its (few) descriptors tells every lane where each of
// its positions comes from, all loads are issued, then all stores.  That leaves no dependent
// global load in front of the copies except the descriptor load, which the prefetch hides.
// Measured on 1 M ORFs of ~300 nt (2.4 GB moved): 0.73 ms = 3.3 TB/s; the first version (one
// wave per ORF, metadata re-loaded at every step, one element per lane in flight) 1.32 ms.
// Two flatter variants (intervals of several ORFs copied together) were slower.
constexpr int kGatherUnroll = 8;  // independent loads in flight per lane: 512 nt per round

// inclusive add-scan over the wave on the DPP network
__device__ __forceinline__ int wave_add_scan_i32(int x)
{
    x += __builtin_amdgcn_update_dpp(0, x, 0x111 /* row_shr:1 */, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112 /* row_shr:2 */, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114 /* row_shr:4 */, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118 /* row_shr:8 */, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142 /* row_bcast:15 */, 0xa, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x143 /* row_bcast:31 */, 0xc, 0xf, false);
    return x;
}

__device__ __forceinline__ long long readlane64(long long v, int l)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(unsigned long long)v, l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)v >> 32), l);
    return (long long)(((unsigned long long)hi << 32) | lo);
}

// One interval, one element per lane and step (used for the rare ORFs with > 64 exons).
__device__ __forceinline__ void gather_copy(const int32_t *__restrict__ coverage, long long coverage_len,
                                            long long src0, int n, long long asc, long long out0, long long len,
                                            bool rev, int32_t *__restrict__ counts, int lane)
{
    for (int i = lane; i < n; i += kWave) {
        const long long src = src0 + i;
        const int v = (src >= 0 && src < coverage_len) ? coverage[src] : 0;
        const long long a = asc + i;
        if (a < len) counts[out0 + (rev ? len - 1 - a : a)] = v;
    }
}

__global__ __launch_bounds__(kWaveBlock) void k_gather_profiles(
    const int32_t *__restrict__ coverage, long long coverage_len, const int64_t *__restrict__ iv_start,
    const int32_t *__restrict__ iv_len, const int64_t *__restrict__ orf_iv,
    const uint8_t *__restrict__ reverse, const int64_t *__restrict__ offsets, long long n_orfs,
    int32_t *__restrict__ counts)
{
    const int lane = threadIdx.x & (kWave - 1);
    const long long waves_total = (long long)gridDim.x * (kWaveBlock / kWave);
    const long long n_batches = (n_orfs + kWave - 1) / kWave;
    long long batch = (long long)blockIdx.x * (kWaveBlock / kWave) + (threadIdx.x >> 6);
    for (; batch < n_batches; batch += waves_total) {
        const long long base = batch * kWave;
        const long long orf = base + lane;
        long long out0 = 0, len = 0, k0 = 0;
        int nk = 0, rev = 0;
        if (orf < n_orfs) {
            out0 = offsets[orf];
            len = (long long)offsets[orf + 1] - out0;
            k0 = orf_iv[orf];
            nk = (int)((long long)orf_iv[orf + 1] - k0);
            rev = reverse[orf];
        }
        const int n_valid = (int)(n_orfs - base < kWave ? n_orfs - base : kWave);
        // descriptors of the batch's first ORF, lane t = its interval t
        long long s_cur = 0;
        int n_cur = 0;
        {
            const long long k0_0 = readlane64(k0, 0);
            const int nk_0 = __builtin_amdgcn_readlane(nk, 0);
            if (lane < nk_0) {
                s_cur = iv_start[k0_0 + lane];
                n_cur = iv_len[k0_0 + lane];
            }
        }
        for (int j = 0; j < n_valid; ++j) {  // wave-uniform
            const long long out0_j = readlane64(out0, j);
            const long long len_j = readlane64(len, j);
            const long long k0_j = readlane64(k0, j);
            const int nk_j = __builtin_amdgcn_readlane(nk, j);
            const bool rev_j = __builtin_amdgcn_readlane(rev, j) != 0;
            // prefetch the next ORF's descriptors
            long long s_nxt = 0;
            int n_nxt = 0;
            if (j + 1 < n_valid) {
                const long long k0_n = readlane64(k0, j + 1);
                const int nk_n = __builtin_amdgcn_readlane(nk, j + 1);
                if (lane < nk_n) {
                    s_nxt = iv_start[k0_n + lane];
                    n_nxt = iv_len[k0_n + lane];
                }
            }
            // ascending position of each interval's first nucleotide inside the ORF
            const int incl = wave_add_scan_i32(n_cur);
            const int asc_lane = incl - n_cur;
            const int first = nk_j < kWave ? nk_j : kWave;
            const int covered = __builtin_amdgcn_readlane(incl, kWave - 1);  // nt in the first 64 intervals
            const int todo = (int)(covered < len_j ? covered : len_j);
            int32_t *const dst = counts + out0_j;
            for (int a0 = 0; a0 < todo; a0 += kGatherUnroll * kWave) {
                int val[kGatherUnroll];
#pragma unroll
                for (int u = 0; u < kGatherUnroll; ++u) val[u] = 0;
                for (int t = 0; t < first; ++t) {  // wave-uniform walk over the descriptors
                    const int st = __builtin_amdgcn_readlane(asc_lane, t);
                    const int nn = __builtin_amdgcn_readlane(n_cur, t);
                    if (st + nn <= a0 || st >= a0 + kGatherUnroll * kWave) continue;
                    const long long s0 = readlane64(s_cur, t);
                    const int32_t *const src = coverage + (s0 - st);  // src[a] is the count at ORF position a
                    if (s0 >= 0 && s0 + nn <= coverage_len) {          // whole interval inside the array
#pragma unroll
                        for (int u = 0; u < kGatherUnroll; ++u) {
                            const int a = a0 + u * kWave + lane;
                            if (a >= st && a < st + nn) val[u] = src[a];
                        }
                    } else {  // hangs off an end: those positions read as 0
#pragma unroll
                        for (int u = 0; u < kGatherUnroll; ++u) {
                            const int a = a0 + u * kWave + lane;
                            const long long g = s0 + (a - st);
                            if (a >= st && a < st + nn && g >= 0 && g < coverage_len) val[u] = coverage[g];
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < kGatherUnroll; ++u) {  // every load above is in flight by now
                    const int a = a0 + u * kWave + lane;
                    if (a < todo) stream_store(dst + (rev_j ? (int)len_j - 1 - a : a), (int32_t)val[u]);
                }
            }
            if (nk_j > kWave) {  // more than 64 exons: the rest straight from memory
                long long asc = covered;
                for (long long k = k0_j + kWave; k < k0_j + nk_j; ++k) {
                    const int n = iv_len[k];
                    gather_copy(coverage, coverage_len, iv_start[k], n, asc, out0_j, len_j, rev_j, counts, lane);
                    asc += n;
                }
            }
            s_cur = s_nxt;
            n_cur = n_nxt;
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
