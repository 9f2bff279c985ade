// rp_device.hpp -- device-side building blocks shared by every phase-score kernel.
//
// gfx950 only (wave64).  The arithmetic follows the closed form of
// ribotricer/statistics.py:67-108 in an oblique integer
// basis that keeps the per-codon work small:
//
//   codon (a,b,c):  d0 = a-b, d1 = b-c                    (exact int32)
//                   q  = d0^2 + d0*d1 + d1^2  = |a + b w + c w^2|^2,  w = e^{2 pi i/3}
//                   unit vector u = ((d0 + d1/2), (sqrt3/2) d1) / sqrt(q)
//   frame sums:     P = sum d0/sqrt(q),  Q = sum d1/sqrt(q)
//                   |sum u|^2 = P^2 + P*Q + Q^2
//   score_f = |sum u|^2 / (N_f * M_f),   N_f = #codons not all-zero,
//                                        M_f = #codons not a==b==c  (q != 0)
//
// q is an integer >= 1 whenever it is non-zero, so the all-equal test is exact
// and no epsilon is involved anywhere before the final frame comparison.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ribophase.h"

namespace rp {

constexpr int kWave = 64;

// Margins under which an fp32 frame comparison is not trusted and the ORF is re-derived in
// float64.
//  * kRecheckMargin, relative to max(1, score): the wave-per-ORF path, whose per-lane fp32
//    sums grow with the ORF (error <= ~3e-6 for the longest we allow in fp32).
//  * the tile path keeps every fp32 sum short (<= 45 terms per lane, <= 16 lanes per row,
//    float64 above): with gamma ~ 61 * 2^-24 the error of P, Q is <= gamma * M, hence the
//    error of score = |sum u|^2 / (N M) is <= 2 gamma |sum u| / N = 2 gamma sqrt(score M / N)
//    <= 7.4e-6 * sqrt(score) in the worst case (typically 30x less).  Margin:
//    kTileMargin * sqrt(score) + kTileMarginAbs; the absolute part covers the second-order
//    term gamma^2 M / N and stays above the tie tolerance (RP_TIE_RTOL).
constexpr double kRecheckMargin = 1e-5;
constexpr double kTileMargin = 1e-5;
constexpr double kTileMarginAbs = 1e-8;
// A phase score built from fp32 sums is off by <= kTileMargin * sqrt(score) / (2 sqrt(score)) =
// 5e-6 (tile path; the wave path's bound is of the same size): when it lands closer than
// this to phase_score_cutoff the `>=` of detect_orfs.py:290 is decided in float64 instead.
constexpr double kCutoffMargin = 2e-5;

// ---------------------------------------------------------------------------
// per-codon accumulation
// ---------------------------------------------------------------------------
template <typename Real>
struct FrameAcc {
    Real p;  // sum d0 / sqrt(q)
    Real q;  // sum d1 / sqrt(q)
    int n;   // codons that are not all-zero
    int m;   // codons that are not a == b == c
};

template <typename Real>
__device__ __forceinline__ void acc_clear(FrameAcc<Real> (&acc)[3])
{
#pragma unroll
    for (int f = 0; f < 3; ++f) {
        acc[f].p = Real(0);
        acc[f].q = Real(0);
        acc[f].n = 0;
        acc[f].m = 0;
    }
}

// fp32: one v_rsq_f32 (1 ulp) per codon; q >= 1 so no denormal handling is needed.
__device__ __forceinline__ void codon_add(FrameAcc<float> &acc, int a, int b, int c, bool in_range)
{
    const int d0i = a - b;
    const int d1i = b - c;
    const bool nz = in_range && ((a | b | c) != 0);
    const bool use = in_range && ((d0i | d1i) != 0);
    const float d0 = (float)d0i;
    const float d1 = (float)d1i;
    const float qq = __builtin_fmaf(d0, d0 + d1, d1 * d1);
    const float r = use ? __builtin_amdgcn_rsqf(qq) : 0.0f;
    acc.p = __builtin_fmaf(d0, r, acc.p);
    acc.q = __builtin_fmaf(d1, r, acc.q);
    acc.n += nz ? 1 : 0;
    acc.m += use ? 1 : 0;
}

// fp64: q is an exact integer-valued double, r = rsqrt(q) after one Newton step is good
// to ~1 ulp.  Frames that tie in exact arithmetic may then differ by ~1e-16 relative,
// which the RP_TIE_RTOL rule of combine_frames absorbs (a correctly rounded sqrt + two
// divisions cost 4x more and made the float64 re-walks the slowest part of the finish kernel).
__device__ __forceinline__ double rsqrt_f64(double q)
{
    double r = __builtin_amdgcn_rsq(q);          // v_rsq_f64, ~2^-26 relative
    const double e = __builtin_fma(-q * r, r, 1.0);  // 1 - q r^2
    r = __builtin_fma(r * e, __builtin_fma(e, 0.375, 0.5), r);  // r (1 + e/2 + 3 e^2/8)
    return r;
}

__device__ __forceinline__ void codon_add(FrameAcc<double> &acc, int a, int b, int c, bool in_range)
{
    const int d0i = a - b;
    const int d1i = b - c;
    const bool nz = in_range && ((a | b | c) != 0);
    const bool use = in_range && ((d0i | d1i) != 0);
    const double d0 = (double)d0i;
    const double d1 = (double)d1i;
    const double q = __builtin_fma(d0, d0 + d1, d1 * d1);
    const double r = use ? rsqrt_f64(q) : 0.0;
    acc.p = __builtin_fma(d0, r, acc.p);
    acc.q = __builtin_fma(d1, r, acc.q);
    acc.n += nz ? 1 : 0;
    acc.m += use ? 1 : 0;
}

// float64 INPUT values (metagene-style profiles): same closed form, tests are
// exact comparisons on the given doubles (statistics.py:72).
__device__ __forceinline__ void codon_add_f64in(FrameAcc<double> &acc, double a, double b, double c,
                                                bool in_range)
{
    const bool nz = in_range && !(a == 0.0 && b == 0.0 && c == 0.0);
    const bool use = nz && !(a == b && b == c);
    if (use) {
        const double d0 = a - b;
        const double d1 = b - c;
        const double s = sqrt(__builtin_fma(d0, d0 + d1, d1 * d1));
        acc.p += d0 / s;
        acc.q += d1 / s;
    }
    acc.n += nz ? 1 : 0;
    acc.m += use ? 1 : 0;
}

// ---------------------------------------------------------------------------
// wave64 reductions on the DPP network (row_shr 1/2/4/8, row_bcast 15/31 -- gfx9 forms):
// lane 63 ends with the total, which is then broadcast from an SGPR.  About 4x cheaper
// than the ds_bpermute butterflies __shfl_xor compiles to (24 vs ~4.5 cycles per step).
// ---------------------------------------------------------------------------
template <int CTRL, int ROW_MASK, bool ZERO_FILL>
__device__ __forceinline__ int dpp_mov(int old, int src)
{
    return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, 0xf, ZERO_FILL);
}

#define RP_DPP_REDUCE_STEPS(STEP)  \
    STEP(0x111, 0xf, true)         \
    STEP(0x112, 0xf, true)         \
    STEP(0x114, 0xf, true)         \
    STEP(0x118, 0xf, true)         \
    STEP(0x142, 0xa, false)        \
    STEP(0x143, 0xc, false)

__device__ __forceinline__ int wave_sum(int v)
{
#define RP_STEP(C, M, Z) v += dpp_mov<C, M, Z>(0, v);
    RP_DPP_REDUCE_STEPS(RP_STEP)
#undef RP_STEP
    return __builtin_amdgcn_readlane(v, 63);
}

__device__ __forceinline__ int wave_min(int v)
{
#define RP_STEP(C, M, Z) v = min(v, dpp_mov<C, M, false>(v, v));
    RP_DPP_REDUCE_STEPS(RP_STEP)
#undef RP_STEP
    return __builtin_amdgcn_readlane(v, 63);
}

__device__ __forceinline__ double wave_sum(double v)
{
#define RP_STEP(C, M, Z)                                                              \
    {                                                                                 \
        const int lo = dpp_mov<C, M, Z>(0, __double2loint(v));                        \
        const int hi = dpp_mov<C, M, Z>(0, __double2hiint(v));                        \
        v += __hiloint2double(hi, lo);                                                \
    }
    RP_DPP_REDUCE_STEPS(RP_STEP)
#undef RP_STEP
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ long long wave_sum(long long v)
{
#define RP_STEP(C, M, Z)                                                              \
    {                                                                                 \
        const unsigned lo = (unsigned)dpp_mov<C, M, Z>(0, (int)(unsigned)v);          \
        const unsigned hi = (unsigned)dpp_mov<C, M, Z>(0, (int)(unsigned)((unsigned long long)v >> 32)); \
        v += (long long)(((unsigned long long)hi << 32) | lo);                        \
    }
    RP_DPP_REDUCE_STEPS(RP_STEP)
#undef RP_STEP
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 63);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)v >> 32), 63);
    return (long long)(((unsigned long long)hi << 32) | lo);
}

// ---------------------------------------------------------------------------
// frame scores and the frame state machine
// ---------------------------------------------------------------------------
struct FrameScore {
    double score;  // 0 when n == 0, NaN when m == 0 < n
    int n;
    int m;
};

__device__ __forceinline__ FrameScore frame_score(double p, double q, int n, int m)
{
    FrameScore r;
    r.n = n;
    r.m = m;
    if (n == 0) {
        r.score = 0.0;
    } else if (m == 0) {
        r.score = __builtin_nan("");
    } else {
        // x / (n m) through v_rcp_f64 + two Newton steps + one residual correction (~1 ulp;
        // frames that tie exactly may now differ by 1e-16, which RP_TIE_RTOL absorbs) -- a
        // third of the instructions of the IEEE division sequence, and this runs on one wave
        const double x = __builtin_fma(p, p + q, q * q);
        const double y = (double)n * (double)m;  // exact: n, m < 2^26
        double inv = __builtin_amdgcn_rcp(y);
        inv = __builtin_fma(__builtin_fma(-y, inv, 1.0), inv, inv);
        inv = __builtin_fma(__builtin_fma(-y, inv, 1.0), inv, inv);
        const double s0 = x * inv;
        r.score = __builtin_fma(__builtin_fma(-s0, y, x), inv, s0);
    }
    return r;
}

__device__ __forceinline__ double tie_tol(double best)
{
    return __builtin_fma(RP_TIE_RTOL, best, RP_TIE_ATOL);
}

// statistics.py:64-66,94-95,109-115.  An empty frame resets; a later frame wins only
// when it beats the running best by more than the tie tolerance (exact-arithmetic
// rule: a tie keeps the earlier frame); RP_FLAG_TIE marks ORFs where another live
// frame with a different N sits within the tolerance of the final best.
__device__ __forceinline__ void combine_frames(const FrameScore (&fr)[3], double &phase, int &valid,
                                               unsigned &flags)
{
    double coh = 0.0;
    int v = -1;
    int first_live = 0;
#pragma unroll
    for (int f = 0; f < 3; ++f) {
        if (fr[f].n == 0) {
            coh = 0.0;
            v = 0;
            first_live = f + 1;
        } else {
            if (fr[f].score > coh + tie_tol(coh)) {
                coh = fr[f].score;
                v = fr[f].n;
            }
            if (v == -1) v = fr[f].n;
        }
    }
    unsigned fl = 0;
#pragma unroll
    for (int f = 0; f < 3; ++f) {
        if (f >= first_live && fr[f].n != v && fabs(fr[f].score - coh) <= tie_tol(coh)) fl |= RP_FLAG_TIE;
    }
    phase = coh > 0.0 ? coh * rsqrt_f64(coh) : 0.0;  // sqrt to ~1 ulp (np.sqrt: statistics.py:115)
    valid = v;
    flags = fl;
}

// Can the frame decision made on fp32-accumulated scores be trusted?  The outcome of the
// state machine depends only on the largest live score and on which frame reaches it
// first (the initial best is 0 with the fallback N), so only the CONTENDERS matter: the
// frames -- and the initial 0 -- within the fp32 error margin of that maximum.  If they
// all carry the same N the choice between them cannot change valid_codons; if two of them
// differ in N the ORF is re-walked in float64.  (Close pairs further down the ranking are
// common in sparse profiles and irrelevant.)
__device__ __forceinline__ bool fp32_decision_unsafe(const FrameScore (&fr)[3], bool tile_sums = false)
{
    int first_live = 0;
#pragma unroll
    for (int f = 0; f < 3; ++f)
        if (fr[f].n == 0) first_live = f + 1;
    if (first_live >= 3) return false;
    const int n_fallback = first_live == 0 ? fr[0].n : (first_live == 1 ? fr[1].n : fr[2].n);
    double smax = 0.0;
#pragma unroll
    for (int f = 0; f < 3; ++f)
        if (f >= first_live) smax = fmax(smax, fr[f].score);  // fmax ignores a NaN score
    const double margin = tile_sums ? kTileMargin * (smax > 0.0 ? smax * rsqrt_f64(smax) : 0.0) + kTileMarginAbs
                                    : kRecheckMargin * fmax(1.0, smax);
    const double thr = smax - margin;
    int n_ref = 0.0 >= thr ? n_fallback : -1;
    bool unsafe = false;
#pragma unroll
    for (int f = 0; f < 3; ++f) {
        if (f < first_live || !(fr[f].score >= thr)) continue;  // NaN frames are decided by integers
        if (n_ref < 0)
            n_ref = fr[f].n;
        else if (fr[f].n != n_ref)
            unsafe = true;
    }
    return unsafe;
}

// ---------------------------------------------------------------------------
// Exact frame ties: replay of the reference's own float64 arithmetic.
//
// When two reading frames score the same in exact arithmetic, statistics.py:109's strict `>`
// is decided by the last bits of what numpy/scipy computed.  For those ORFs
// (RP_FLAG_TIE, ~0.3 %) a wave repeats that computation operation for operation in IEEE
// float64 -- per codon in parallel, the sums as the same left folds numpy performs -- and its
// (phase, valid_codons) replace the closed form's.  The sequence (established against scipy
// 1.15.3 / numpy 2.2.6 on x86-64 with FMA; oracle/scipy_replay.c is its CPU twin and is
// bit-identical to the reference on every golden vector):
//   codon (a,b,c): real = (a + b cos(2pi/3)) + c cos(4pi/3), image = b sin(2pi/3) + c sin(4pi/3)
//                  norm = sqrt(pow(real,2) + pow(image,2))   statistics.py:75-85
//                  v = (a,b,c) / norm                        statistics.py:86-90
//   coherence():   m = ((v0+v1)+v2)/3, d = v - m             detrend 'constant'
//                  X = (d0 - (d1+d2)/2, tw (d2-d1))          pocketfft radix-3, bin 1
//                  pxx = (fma(Xr,Xr,Xi Xi)/3) 2 ; pxy = ((Xr/3) 2, (-Xi/3) 2) ; pyy = 2/3
//                  Pxx = fold(pxx)/N, Pyy = fold(pyy)/N, Pxy = fold(pxy) * (1/N)   (N == 1: no mean)
//                  |Pxy| = max sqrt(fma(q,q,1)), q = min/max ; Cxy = |Pxy|^2 / Pxx / Pyy
// pow() is glibc's, which is NOT x*x (it differs in the last bit for 0.8 % of the arguments),
// and everything up to (pxx, pxy) depends on the codon (a,b,c) alone: the three numbers of
// every codon with counts < 16 come from a table the HOST fills once per device with its own
// libm (ribophase.hip: the libm the reference would run on) -- tie-flagged ORFs are sparse, so
// that is nearly all of them; rarer, larger codons are evaluated here with x*x for pow.  No fp
// contraction anywhere in here: every fused operation is explicit.
// ---------------------------------------------------------------------------
constexpr int kCodonTabBits = 4;
struct alignas(32) CodonTerms {
    double pxx, pxr, pxi, pad;
};
__device__ CodonTerms rp_codon_tab[1 << (3 * kCodonTabBits)];

__device__ __forceinline__ double readlane_f64(double x, int l)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
    return __hiloint2double(hi, lo);
}

#pragma clang fp contract(off)
// (pxx, pxr, pxi) of one non-zero codon whose counts are past the host-filled table (pow as x*x)
__device__ __forceinline__ void replay_codon_compute(int a, int b, int c, double &pxx, double &pxr, double &pxi)
{
    constexpr double kC23 = -0x1.ffffffffffffcp-2, kC43 = -0x1.0000000000004p-1;
    constexpr double kS23 = 0x1.bb67ae8584cabp-1, kS43 = -0x1.bb67ae8584ca8p-1;
    constexpr double kTwI = 0x1.bb67ae8584caap-1, kScale = 0x1.5555555555555p-2;
    const double real = ((double)a + (double)b * kC23) + (double)c * kC43;
    const double image = (double)b * kS23 + (double)c * kS43;
    double norm = __builtin_sqrt(real * real + image * image);
    if (norm == 0.0) norm = 1.0;
    const double v0 = (double)a / norm, v1 = (double)b / norm, v2 = (double)c / norm;
    const double m = ((v0 + v1) + v2) / 3.0;
    const double d0 = v0 - m, d1 = v1 - m, d2 = v2 - m;
    const double xr = d0 + (-0.5) * (d1 + d2);
    const double xi = kTwI * (d2 - d1);
    pxx = (__builtin_fma(xr, xr, xi * xi) * kScale) * 2.0;
    pxr = (xr * kScale) * 2.0;
    pxi = (-xi * kScale) * 2.0;
}

#ifdef RP_REWALK_STAMPS
__device__ unsigned long long g_replay_stamps[8];
#endif
// One wave, one ORF.  Lane t takes triplets t, t + 64, ...: five counts give it one codon of
// each reading frame (one pass over the profile for all three frames).  The per-frame sums are
// plain left folds in codon order, exactly as numpy performs them -- inherently sequential, so
// they run in ONE lane per frame (lanes 0, 1, 2, side by side): every chunk's non-zero codons
// are compacted into LDS in codon order and the three lanes add theirs up, then score their
// frame.  (The fold used to be wave-uniform, 15 instructions and six lane reads per codon and
// frame after frame; it was most of the ~18 us a replay took.)
struct ReplayLds {
    double t[3][65][3];  // [frame][slot][pxx, pxr, pxi]; 65: the frames' rows on different banks
};

// Returns true (wave-uniform) when a codon past the table took part: the result then stands on
// x*x where the reference has the host libm's pow() -- RP_FLAG_BIGTIE, resolved on the host.
template <typename Counts>
__device__ __forceinline__ bool replay_tie_wave(Counts v, long long len, int lane,
                                             double &phase, int &valid, ReplayLds *lds)
{
#ifdef RP_REWALK_STAMPS  // (timing experiment: phases of the replay, summed into g_replay_stamps)
    unsigned long long rs_mem = 0, rs_fold = 0, rs_tail = 0;
    unsigned long long rs_t = __builtin_amdgcn_s_memtime();
#define RP_RS_LAP(acc) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc += now_ - rs_t; rs_t = now_; } while (0)
#else
#define RP_RS_LAP(acc)
#endif
    bool any_big = false;
    constexpr double kPyySeg = 0x1.5555555555555p-1;
    double sxx = 0.0, sxr = 0.0, sxi = 0.0;  // lane f < 3: the running sums of frame f
    int n = 0;
    const long long n_trip = len / 3;  // frame f has a codon at triplet j iff 3j + f + 2 < len
    const long long last = len > 0 ? len - 1 : 0;
    // the five counts of triplet j: five back-to-back loads at clamped (always in-range) indices, masked afterwards -- a
    // select per value, no per-lane branch.  (Until round 6 every load sat behind `if (j < n_trip)`: five branches, a wait
    // behind each, FIVE dependent memory round trips where this is one -- 70 % of a tie's re-walk, profiles/r06_ab_finish_tail.txt.)
    auto load5 = [&](long long j, int (&w)[5]) {
        const bool live = j < n_trip;
        const long long p = live ? 3 * j : 0;
        int x[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) x[k] = len > 0 ? v[p + k < last ? p + k : last] : 0;  // (len > 0: wave-uniform)
#pragma unroll
        for (int k = 0; k < 5; ++k) w[k] = (live && p + k < len) ? x[k] : 0;
    };
    int w[5], wn[5];
    load5(lane, w);
    const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (long long j0 = 0; j0 < n_trip; j0 += kWave) {
        const long long j = j0 + lane;
        const long long p = 3 * j;
        load5(j + kWave, wn);  // the next chunk is in flight while this one is folded
        int mine = 0;  // lane f < 3: how many codons of frame f this chunk holds
        bool nz[3], big[3];
        CodonTerms t[3];
#pragma unroll
        for (int f = 0; f < 3; ++f) {  // the three frames' table rows travel together: one memory latency, not three
            const int a = w[f], b = w[f + 1], c = w[f + 2];
            nz[f] = j < n_trip && p + f + 2 < len && (a | b | c) != 0;
            big[f] = (unsigned)(a | b | c) >= (1u << kCodonTabBits);
#ifdef RP_EXPERIMENT_NO_TABLE  // timing experiment only (x*x where the reference has pow(): last bits differ on 0.8 % of the codons)
            big[f] = true;
#endif
            t[f] = rp_codon_tab[(nz[f] && !big[f]) ? ((a << (2 * kCodonTabBits)) | (b << kCodonTabBits) | c) : 0];
        }
#pragma unroll
        for (int f = 0; f < 3; ++f) {
            const unsigned long long mask = __ballot(nz[f]);
            if (nz[f]) {
                double pxx = t[f].pxx, pxr = t[f].pxr, pxi = t[f].pxi;
                if (big[f]) {
                    replay_codon_compute(w[f], w[f + 1], w[f + 2], pxx, pxr, pxi);
                    any_big = true;
                }
                double *slot = lds->t[f][__builtin_popcountll(mask & below)];
                slot[0] = pxx;
                slot[1] = pxr;
                slot[2] = pxi;
            }
            if (lane == f) mine = __builtin_popcountll(mask);
        }
        __builtin_amdgcn_wave_barrier();  // (one wave: its LDS operations complete in order)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        RP_RS_LAP(rs_mem);  // the chunk's counts and table rows have arrived, its terms are in LDS
        if (lane < 3) {
            const double(*row)[3] = lds->t[lane];
            for (int i = 0; i < mine; ++i) {  // numpy's reductions here are plain left folds in segment order
                const double t0 = row[i][0], t1 = row[i][1], t2 = row[i][2];
                if (n == 0) {
                    sxx = t0;
                    sxr = t1;
                    sxi = t2;
                } else {
                    sxx = sxx + t0;
                    sxr = sxr + t1;
                    sxi = sxi + t2;
                }
                ++n;
            }
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the rows are rewritten by the next chunk
        RP_RS_LAP(rs_fold);
#pragma unroll
        for (int k = 0; k < 5; ++k) w[k] = wn[k];
    }
    // lanes 0..2: the frame's coherence at f = 1/3
    double score = 0.0;
    if (lane < 3 && n > 0) {
        double pxx_m = sxx, pyy_m = kPyySeg, re = sxr, im = sxi;
        if (n > 1) {
            const double dn = (double)n;
            pxx_m = sxx / dn;
            double syy = kPyySeg;
            for (int k = 1; k < n; ++k) syy = syy + kPyySeg;
            pyy_m = syy / dn;
            const double scl = 1.0 / dn;
            re = sxr * scl;
            im = sxi * scl;
        }
        const double ar = __builtin_fabs(re), ai = __builtin_fabs(im);
        const double mx = ar > ai ? ar : ai, mn = ar > ai ? ai : ar;
        double ab = 0.0;
        if (mx != 0.0) {
            const double q = mn / mx;
            ab = mx * __builtin_sqrt(__builtin_fma(q, q, 1.0));
        }
        score = ((ab * ab) / pxx_m) / pyy_m;
    }
    // the frame state machine, wave-uniform, on the three lanes' results
    double coh = 0.0;
    int val = -1;
#pragma unroll
    for (int f = 0; f < 3; ++f) {
        const int nf = __builtin_amdgcn_readlane(n, f);
        const double sf = readlane_f64(score, f);
        if (nf == 0) {  // empty frame: reset (statistics.py:94-95)
            coh = 0.0;
            val = 0;
            continue;
        }
        if (sf > coh) {  // the reference's own strict '>' (statistics.py:109); NaN never wins
            coh = sf;
            val = nf;
        }
        if (val == -1) val = nf;
    }
    phase = __builtin_sqrt(coh);
    valid = val;
#ifdef RP_REWALK_STAMPS
    RP_RS_LAP(rs_tail);
    if (lane == 0) {
        atomicAdd(&g_replay_stamps[0], 1ull);
        atomicAdd(&g_replay_stamps[1], rs_mem);
        atomicAdd(&g_replay_stamps[2], rs_fold);
        atomicAdd(&g_replay_stamps[3], rs_tail);
        atomicAdd(&g_replay_stamps[4], (unsigned long long)__builtin_amdgcn_readlane(n, 0));
    }
#endif
    return __ballot(any_big) != 0;
}
#pragma clang fp contract(fast)

// detect_orfs.py:281,285-299
struct FilterParams {
    double phase_score_cutoff;
    double min_valid_codons_ratio;
    double min_density_over_orf;
    double min_reads_per_codon;
    int min_valid_codons;
    int enabled;       // thresholds given and a status array to fill
    int printed_only;  // rp_filter_params.flags & RP_FILTER_PRINTED_ONLY (and enabled)
    int pad_;
};

__device__ __forceinline__ unsigned char orf_status(const FilterParams &fp, double phase, int valid,
                                                    long long read_count, int min_codon_cov,
                                                    long long length)
{
    bool ok = phase >= fp.phase_score_cutoff && valid >= fp.min_valid_codons &&
              (double)min_codon_cov >= fp.min_reads_per_codon;
    // ratio and density are >= 0, so the two IEEE divisions only matter for positive
    // thresholds (the reference defaults are 0, const.py:35,39); wave-uniform branch
    if (fp.min_valid_codons_ratio > 0.0 || fp.min_density_over_orf > 0.0) {
        const long long n_codons = (length / 3) > 1 ? (length / 3) : 1;  // max(1, length // 3)
        const double ratio = (double)valid / (double)n_codons;
        const double density = (double)read_count / (double)n_codons;
        ok = ok && ratio >= fp.min_valid_codons_ratio && density >= fp.min_density_over_orf;
    }
    return ok ? 1 : 0;
}

// Is an fp32-based phase score too close to the cutoff for the status comparison
// (detect_orfs.py:290, `coh >= phase_score_cutoff`) to be trusted?
__device__ __forceinline__ bool near_cutoff(const FilterParams &fp, double phase)
{
    return fp.enabled && fabs(phase - fp.phase_score_cutoff) <= kCutoffMargin;
}

// RP_FILTER_PRINTED_ONLY: can NO resolution of this ORF's frame decision make it "translating"?  Whatever the float64
// re-walk or the tie replay would return, valid_codons is the N of one of the three frames (or 0 after a reset,
// statistics.py:94-95,109-113) and the phase score is the root of one frame's score (or 0); read_count, min_codon_cov
// and the length are exact integers already.  So the ORF is nontranslating for sure (detect_orfs.py:289-299) when
//   max_f N_f < min_valid_codons,  or  max_f N_f / n_codons < min_valid_codons_ratio,  or an integer condition fails,
//   or  max_f sqrt(score_f) lies below the cutoff by more than the fp32 error margin (kCutoffMargin).
// In default mode (detect_orfs.py:301-302: only translating rows are printed) such an ORF prints nothing either way.
__device__ __forceinline__ bool cannot_be_translating(const FilterParams &fp, const FrameScore (&fr)[3], long long read_count,
                                                      int min_codon_cov, long long length)
{
    const int n_max = max(fr[0].n, max(fr[1].n, fr[2].n));
    if (n_max < fp.min_valid_codons || (double)min_codon_cov < fp.min_reads_per_codon) return true;
    if (fp.min_valid_codons_ratio > 0.0 || fp.min_density_over_orf > 0.0) {
        const long long n_codons = (length / 3) > 1 ? (length / 3) : 1;
        if (!((double)n_max / (double)n_codons >= fp.min_valid_codons_ratio) ||
            !((double)read_count / (double)n_codons >= fp.min_density_over_orf))
            return true;
    }
    double s_max = 0.0;
#pragma unroll
    for (int f = 0; f < 3; ++f) s_max = fmax(s_max, fr[f].score);  // (fmax ignores the NaN of an all-flat frame: it never wins)
    const double phase_max = s_max > 0.0 ? s_max * rsqrt_f64(s_max) : 0.0;
    return phase_max + kCutoffMargin < fp.phase_score_cutoff;
}

// Stores of data that this kernel will not touch again.  Measured on gfx950 (scripts/probe_rw.py,
// profiles/archive/r03_probe_rw.txt): a read stream at 7.15 TB/s that also writes 1 152 bytes per 32 KiB read
// -- 3.5 % more bytes, the segment records of k_tile_score -- loses 17 % with ordinary stores and 9 %
// with `nt` stores (sc1: 10 %; sc0, dword instead of dwordx4, scalar stores, atomics, one plane or
// three, earlier in the workgroup: all 16-19 %).  Writes cost five times their share of the bytes
// when they ride a read stream, so every byte that need not be written is worth five read.
#ifndef RP_NT_STORES
#define RP_NT_STORES 1
#endif
template <typename T>
__device__ __forceinline__ void stream_store(T *p, T v)
{
#if RP_NT_STORES
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

struct OrfOutputs {
    double *phase;
    int32_t *valid;
    int64_t *read_count;
    int32_t *min_codon_cov;
    uint8_t *flags;
    uint8_t *status;  // may be null
};

__device__ __forceinline__ void store_orf(const OrfOutputs &out, const FilterParams &fp, int64_t i,
                                          double phase, int valid, long long read_count,
                                          int min_codon_cov, unsigned flags, long long length)
{
    // results are written once and read by nobody on the device: streaming stores (see RP_NT_STORES)
    stream_store(out.phase + i, phase);
    stream_store(out.valid + i, (int32_t)valid);
    stream_store(out.read_count + i, (int64_t)read_count);
    stream_store(out.min_codon_cov + i, (int32_t)min_codon_cov);
    stream_store(out.flags + i, (uint8_t)flags);
    if (out.status != nullptr && fp.enabled)
        stream_store(out.status + i, (uint8_t)orf_status(fp, phase, valid, read_count, min_codon_cov, length));
}

}  // namespace rp
