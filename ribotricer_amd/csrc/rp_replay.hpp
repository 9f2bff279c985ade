// rp_replay.hpp -- HOST side of the exact-tie replay (see rp_device.hpp, replay_tie_wave).
//
// When two reading frames of a profile score the same in exact arithmetic, the strict `>` of
// ribotricer/statistics.py:109 is decided by the last bits of what Python, numpy and scipy
// computed.  One step of that computation cannot be restated in device code: `real**2 +
// image**2` (statistics.py:83) goes through the C library's pow(), whose result is not always
// the correctly rounded square (glibc: 0.8 % of the arguments differ from x*x in the last bit)
// and belongs to the machine the reference runs on.  So that step is taken HERE, with this
// host's own libm, in two roles:
//   * codon_terms(): what a codon (a,b,c) contributes to the segment spectra -- tabulated once
//     per device for all codons with counts < 16 (ribophase.hip: fill_codon_table), which is what
//     the device replay reads for nearly every tie-flagged ORF (they are sparse);
//   * replay_profile(): the whole per-profile sequence for the profiles the device cannot finish
//     with the reference's bits -- integer profiles whose tie involves a count >= 16
//     (RP_FLAG_BIGTIE) and float-valued profiles (metagene.py:243-244) -- behind
//     rp_tie_replay_host / rp_tie_replay_f64_host.  A handful of profiles per sample.
//
// The sequence (established against scipy 1.15.3 / numpy 2.2.6 on x86-64 with FMA; checked
// against the reference itself by tests/golden/check_replay_vs_reference.py and the G8 fixtures):
//   codon:  real  = (a + b cos(2pi/3)) + c cos(4pi/3),  image = b sin(2pi/3) + c sin(4pi/3)
//           norm  = sqrt(pow(real,2) + pow(image,2))  (0 -> 1);  v = (a,b,c) / norm
//   coherence(v, [1,0,0]*N, window=[1,1,1], nperseg=3, noverlap=0), per segment:
//           m = ((v0+v1)+v2)/3, d = v - m                        detrend 'constant'
//           X = (d0 - (d1+d2)/2, tw (d2-d1))                     pocketfft radix-3, bin 1
//           pxx = (fma(Xr,Xr,Xi Xi)/3) 2 ; pxy = ((Xr/3) 2, (-Xi/3) 2) ; pyy = 2/3
//   Pxx = fold(pxx)/N, Pyy = fold(pyy)/N, Pxy = fold(pxy) * (1/N)   plain left folds (N == 1: no mean)
//   |Pxy| = max sqrt(fma(q,q,1)), q = min/max ;  Cxy = |Pxy|^2 / Pxx / Pyy
//   frame state machine with the reference's own strict '>'       statistics.py:94-115
// No fp contraction in this file: every fused operation is an explicit fma().
#pragma once

#include <cmath>
#include <cstdint>

namespace rpreplay {

#pragma clang fp contract(off)

struct Terms {
    double pxx, pxr, pxi;
};

// pow through a volatile pointer: the compiler must not fold pow(x, 2.0) into x * x
inline double libm_pow2(double x)
{
    static double (*volatile fn)(double, double) = pow;
    return fn(x, 2.0);
}

inline Terms codon_terms(double a, double b, double c)
{
    static const double c23 = cos(2 * M_PI / 3), c43 = cos(4 * M_PI / 3), s23 = sin(2 * M_PI / 3), s43 = sin(4 * M_PI / 3);
    constexpr double tw = 0x1.bb67ae8584caap-1, scale = 0x1.5555555555555p-2;
    const double real = (a + b * c23) + c * c43;
    const double image = b * s23 + c * s43;
    double norm = sqrt(libm_pow2(real) + libm_pow2(image));
    if (norm == 0.0) norm = 1.0;
    const double v0 = a / norm, v1 = b / norm, v2 = c / norm;
    const double m = ((v0 + v1) + v2) / 3.0;
    const double d0 = v0 - m, d1 = v1 - m, d2 = v2 - m;
    const double xr = d0 + (-0.5) * (d1 + d2);
    const double xi = tw * (d2 - d1);
    Terms t;
    t.pxx = (fma(xr, xr, xi * xi) * scale) * 2.0;
    t.pxr = (xr * scale) * 2.0;
    t.pxi = (-xi * scale) * 2.0;
    return t;
}

// One reading frame: Cxy at f = 1/3 and the number of segments (codons that are not all zero).
template <typename T>
inline void replay_frame(const T *v, int64_t len, int frame, double *score, int32_t *n_out)
{
    constexpr double kPyySeg = 0x1.5555555555555p-1;
    double sxx = 0.0, sxr = 0.0, sxi = 0.0;
    int32_t n = 0;
    for (int64_t i = frame; i + 2 < len; i += 3) {
        const double a = (double)v[i], b = (double)v[i + 1], c = (double)v[i + 2];
        if (a == 0.0 && b == 0.0 && c == 0.0) continue;
        const Terms t = codon_terms(a, b, c);
        if (n == 0) {
            sxx = t.pxx;
            sxr = t.pxr;
            sxi = t.pxi;
        } else {
            sxx = sxx + t.pxx;
            sxr = sxr + t.pxr;
            sxi = sxi + t.pxi;
        }
        ++n;
    }
    *n_out = n;
    *score = 0.0;
    if (n == 0) return;
    double pxx_m = sxx, pyy_m = kPyySeg, re = sxr, im = sxi;
    if (n > 1) {
        const double dn = (double)n;
        pxx_m = sxx / dn;
        double syy = kPyySeg;
        for (int32_t k = 1; k < n; ++k) syy = syy + kPyySeg;
        pyy_m = syy / dn;
        const double scl = 1.0 / dn;
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
    *score = ((ab * ab) / pxx_m) / pyy_m;
}

template <typename T>
inline void replay_profile(const T *v, int64_t len, double *phase, int32_t *valid)
{
    double coh = 0.0;
    int32_t val = -1;
    for (int f = 0; f < 3; ++f) {
        double s;
        int32_t n;
        replay_frame(v, len, f, &s, &n);
        if (n == 0) {  // empty frame: reset (statistics.py:94-95)
            coh = 0.0;
            val = 0;
            continue;
        }
        if (s > coh) {  // NaN never wins
            coh = s;
            val = n;
        }
        if (val == -1) val = n;
    }
    *phase = sqrt(coh);
    *valid = val;
}

#pragma clang fp contract(on)

}  // namespace rpreplay
