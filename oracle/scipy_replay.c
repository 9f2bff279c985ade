/*
 * scipy_replay.c -- operation-for-operation float64 replay of what
 * ribotricer/statistics.py:67-115 makes numpy/scipy compute, WITHOUT numpy or scipy.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  The closed form of phase_oracle.c is
 * exact in exact arithmetic; the reference's own floating-point result differs from it in
 * the last bits, and when two reading frames tie those last bits decide `valid_codons`
 * (SURVEY.md Appendix A.4).  This file reproduces the reference's bits.  The sequence was
 * established in this container by comparing every intermediate array with scipy 1.15.3 /
 * numpy 2.2.6 on an x86-64 CPU with FMA (numpy's AVX2/AVX512 loops) -- 13 701 reading
 * frames, 0 mismatches (tests/golden/check_replay_vs_reference.py keeps that check):
 *
 *   per non-zero codon (a,b,c)                                       statistics.py:72-90
 *     real  = (a + b*cos(2pi/3)) + c*cos(4pi/3)      python floats, left to right
 *     image = b*sin(2pi/3) + c*sin(4pi/3)
 *     norm  = sqrt(pow(real,2) + pow(image,2))       libm pow (NOT real*real: glibc's pow
 *                                                    is not correctly rounded, 0.8 % differ)
 *     v     = (a/norm, b/norm, c/norm)
 *   scipy.signal.coherence(x, [1,0,0]*N, window=[1,1,1], nperseg=3, noverlap=0)
 *     segment k: m = ((v0+v1)+v2)/3 ; d = v - m      detrend 'constant' (np.mean, 3 terms)
 *       X = (d0 - 0.5*(d1+d2),  tw*(d2-d1))          pocketfft radf3, bin 1
 *       pxx_k = (fma(Xr,Xr, Xi*Xi) * (1/3)) * 2      conj(X)*X in numpy's FMA complex multiply
 *       pxy_k = ((Xr * (1/3)) * 2, (-Xi * (1/3)) * 2)   conj(X)*Y with Y = (1,0) exactly
 *       pyy_k = 0x1.5555555555555p-1
 *     Pxx = (left fold of pxx_k) / N ; Pyy likewise  np.mean over a strided view: the nditer
 *                                                    puts the reduced axis OUTSIDE -> plain
 *                                                    sequential sums, true division
 *     Pxy = (fold re * (1/N), fold im * (1/N))       complex / real: numpy's Smith division
 *     (N == 1: no mean at all)
 *     |Pxy| = max * sqrt(fma(q,q,1)), q = min/max    numpy's SIMD complex absolute
 *     Cxy  = ((|Pxy|*|Pxy|) / Pxx) / Pyy
 *   frame state machine, strict '>'                                  statistics.py:94-115
 *
 * Compiled with -ffp-contract=off: every fused operation above is an explicit fma().
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#define RP_OK 0
#define RP_ERR_NULL (-1)
#define RP_ERR_SIZE (-2)

static const double kC23 = -0x1.ffffffffffffcp-2; /* cos(2*pi/3) as python computes it */
static const double kC43 = -0x1.0000000000004p-1; /* cos(4*pi/3) */
static const double kS23 = 0x1.bb67ae8584cabp-1;  /* sin(2*pi/3) */
static const double kS43 = -0x1.bb67ae8584ca8p-1; /* sin(4*pi/3) */
static const double kTwI = 0x1.bb67ae8584caap-1;  /* pocketfft radix-3 twiddle (sqrt(3)/2) */
static const double kScale = 0x1.5555555555555p-2; /* 1.0 / (fs * (win*win).sum()) = fl(1/3) */
static const double kPyySeg = 0x1.5555555555555p-1; /* (fma(1,1,-0) * kScale) * 2 */

/* the reference squares through libm's pow(); called through a volatile pointer so that the
 * compiler cannot turn pow(x, 2.0) into x * x (the two differ in the last bit 0.8 % of the time) */
static double (*volatile libm_pow)(double, double) = pow;

typedef struct {
    double score; /* Cxy[f = 1/3] of this frame; meaningless when n == 0 */
    int32_t n;    /* segments = codons that are not all-zero */
} replay_frame;

static replay_frame replay_one_frame(const int32_t *v, int64_t len, int frame)
{
    replay_frame r = {0.0, 0};
    double sxx = 0.0, sxr = 0.0, sxi = 0.0;
    double first_xx = 0.0, first_xr = 0.0, first_xi = 0.0;
    for (int64_t i = frame; i + 2 < len; i += 3) {
        const int32_t a = v[i], b = v[i + 1], c = v[i + 2];
        if (a == 0 && b == 0 && c == 0) continue;
        const double real = ((double)a + (double)b * kC23) + (double)c * kC43;
        const double image = (double)b * kS23 + (double)c * kS43;
        double norm = sqrt(libm_pow(real, 2.0) + libm_pow(image, 2.0));
        if (norm == 0.0) norm = 1.0;
        const double v0 = (double)a / norm, v1 = (double)b / norm, v2 = (double)c / norm;
        const double m = ((v0 + v1) + v2) / 3.0;
        const double d0 = v0 - m, d1 = v1 - m, d2 = v2 - m;
        const double xr = d0 + (-0.5) * (d1 + d2);
        const double xi = kTwI * (d2 - d1);
        const double pxx = (fma(xr, xr, xi * xi) * kScale) * 2.0;
        const double pxr = (xr * kScale) * 2.0;
        const double pxi = (-xi * kScale) * 2.0;
        if (r.n == 0) {
            sxx = first_xx = pxx;
            sxr = first_xr = pxr;
            sxi = first_xi = pxi;
        } else {
            sxx = sxx + pxx;
            sxr = sxr + pxr;
            sxi = sxi + pxi;
        }
        r.n += 1;
    }
    if (r.n == 0) return r;
    double pxx_m, pyy_m, re, im;
    if (r.n == 1) {
        pxx_m = first_xx;
        pyy_m = kPyySeg;
        re = first_xr;
        im = first_xi;
    } else {
        const double n = (double)r.n;
        pxx_m = sxx / n;
        double syy = kPyySeg;
        for (int32_t k = 1; k < r.n; ++k) syy = syy + kPyySeg;
        pyy_m = syy / n;
        const double scl = 1.0 / n;
        re = sxr * scl;
        im = sxi * scl;
    }
    const double ar = fabs(re), ai = fabs(im);
    const double mx = ar > ai ? ar : ai, mn = ar > ai ? ai : ar;
    double ab = 0.0;
    if (mx != 0.0) {
        const double q = mn / mx;
        ab = mx * sqrt(fma(q, q, 1.0));
    }
    r.score = ((ab * ab) / pxx_m) / pyy_m;
    return r;
}

/* statistics.py:64-66,94-95,109-115 with the reference's own strict '>' */
static void replay_profile(const int32_t *v, int64_t len, double *phase, int32_t *valid,
                           double *frame_score, int32_t *frame_n)
{
    double coh = 0.0;
    int32_t val = -1;
    for (int f = 0; f < 3; ++f) {
        const replay_frame r = replay_one_frame(v, len, f);
        if (frame_score) frame_score[f] = r.n ? r.score : NAN;
        if (frame_n) frame_n[f] = r.n;
        if (r.n == 0) {
            coh = 0.0;
            val = 0;
            continue;
        }
        if (r.score > coh) {
            coh = r.score;
            val = r.n;
        }
        if (val == -1) val = r.n;
    }
    *phase = sqrt(coh);
    *valid = val;
}

int rp_oracle_replay_csr(const int32_t *counts, const int64_t *offsets, int64_t n_orfs, double *phase,
                         int32_t *valid, double *frame_score, int32_t *frame_n)
{
    if (n_orfs < 0) return RP_ERR_SIZE;
    if (!offsets || !phase || !valid) return RP_ERR_NULL;
    for (int64_t i = 0; i < n_orfs; ++i)
        replay_profile(counts + offsets[i], offsets[i + 1] - offsets[i], &phase[i], &valid[i],
                       frame_score ? frame_score + 3 * i : NULL, frame_n ? frame_n + 3 * i : NULL);
    return RP_OK;
}
