// rp_format.hpp -- host-side TSV row renderer.
//
// Replaces the per-ORF `formatter.format(...)` of detect_orfs.py:301-324: for human
// `--report_all` runs the `profile` column alone is ~3 bytes of text per nucleotide
// (~10 GB), and str(list) / str(float) in a Python loop is what is left on the critical
// path once scoring and gathering run on the GPU.  Plain C++17, no HIP; byte-identical to
// CPython's renderings:
//   float  -> repr(float): shortest round-trip digits (std::to_chars), fixed notation
//             when -4 <= exp10 < 16 (with ".0" for integral values), else d.ddde+XX
//   int    -> decimal
//   list   -> "[a, b, c]" ("[]" when empty)
#pragma once

#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstring>

namespace rpfmt {

// repr(float) of CPython 3 (float_repr_style == 'short').  buf must hold 32 bytes.
inline int double_repr(double v, char *buf)
{
    if (std::isnan(v)) {
        std::memcpy(buf, "nan", 3);
        return 3;
    }
    if (std::isinf(v)) {
        if (v < 0) {
            std::memcpy(buf, "-inf", 4);
            return 4;
        }
        std::memcpy(buf, "inf", 3);
        return 3;
    }
    char sci[40];
    const auto r = std::to_chars(sci, sci + sizeof(sci), v, std::chars_format::scientific);
    const char *p = sci;
    char *o = buf;
    if (*p == '-') *o++ = *p++;
    // digits d[.ddd] then e[+-]XX
    char digits[24];
    int nd = 0;
    digits[nd++] = *p++;
    if (*p == '.') {
        ++p;
        while (*p != 'e') digits[nd++] = *p++;
    }
    ++p;  // 'e'
    const bool eneg = *p == '-';
    ++p;
    int e10 = 0;
    while (p < r.ptr) e10 = e10 * 10 + (*p++ - '0');
    if (eneg) e10 = -e10;
    if (nd == 1 && digits[0] == '0') {  // +-0.0
        std::memcpy(o, "0.0", 3);
        return (int)(o - buf) + 3;
    }
    if (e10 < -4 || e10 >= 16) {  // float_repr: decpt <= -4 || decpt > 16, decpt = e10 + 1
        *o++ = digits[0];
        if (nd > 1) {
            *o++ = '.';
            std::memcpy(o, digits + 1, (size_t)nd - 1);
            o += nd - 1;
        }
        *o++ = 'e';
        *o++ = e10 < 0 ? '-' : '+';
        int a = e10 < 0 ? -e10 : e10;
        if (a >= 100) {
            *o++ = (char)('0' + a / 100);
            a %= 100;
            *o++ = (char)('0' + a / 10);
            *o++ = (char)('0' + a % 10);
        } else {
            *o++ = (char)('0' + a / 10);
            *o++ = (char)('0' + a % 10);
        }
    } else if (e10 < 0) {  // 0.000ddd
        *o++ = '0';
        *o++ = '.';
        for (int k = 0; k < -e10 - 1; ++k) *o++ = '0';
        std::memcpy(o, digits, (size_t)nd);
        o += nd;
    } else {  // ddd.ddd or ddd000.0
        const int ip = e10 + 1;  // digits before the point
        if (nd <= ip) {
            std::memcpy(o, digits, (size_t)nd);
            o += nd;
            for (int k = nd; k < ip; ++k) *o++ = '0';
            *o++ = '.';
            *o++ = '0';
        } else {
            std::memcpy(o, digits, (size_t)ip);
            o += ip;
            *o++ = '.';
            std::memcpy(o, digits + ip, (size_t)(nd - ip));
            o += nd - ip;
        }
    }
    return (int)(o - buf);
}

// decimal rendering of a signed 64-bit integer; buf must hold 21 bytes
inline int int_str(long long v, char *buf)
{
    char tmp[24];
    int n = 0;
    unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
    do {
        tmp[n++] = (char)('0' + u % 10);
        u /= 10;
    } while (u);
    char *o = buf;
    if (v < 0) *o++ = '-';
    while (n) *o++ = tmp[--n];
    return (int)(o - buf);
}

// the body lines of a variableStep WIG block, detect_orfs.py:346-351: "{pos}\t{count}\n" per position.
// Returns bytes written; out must hold 42 * n bytes.
inline size_t wig_rows_str(const int64_t *pos, const int64_t *count, long long n, char *out)
{
    char *o = out;
    for (long long k = 0; k < n; ++k) {
        o += int_str(pos[k], o);
        *o++ = '\t';
        o += int_str(count[k], o);
        *o++ = '\n';
    }
    return (size_t)(o - out);
}

// str(list_of_int): "[a, b, c]".  Returns bytes written; out must hold list_bound(n).
inline size_t list_bound(long long n) { return 2 + (size_t)(n > 0 ? n : 0) * 13; }  // "-2147483648, "

inline size_t int_list_str(const int32_t *v, long long n, char *out)
{
    char *o = out;
    *o++ = '[';
    for (long long k = 0; k < n; ++k) {
        if (k) {
            *o++ = ',';
            *o++ = ' ';
        }
        const int32_t x = v[k];
        if (x >= 0 && x < 10) {
            *o++ = (char)('0' + x);  // the overwhelmingly common case for P-site counts
        } else {
            o += int_str(x, o);
        }
    }
    *o++ = ']';
    return (size_t)(o - out);
}

struct RowInputs {
    const int32_t *counts;
    const int64_t *offsets;
    const double *phase;
    const int32_t *valid;
    const int64_t *read_count;
    const uint8_t *status;
    const char *head;         // "ORF_ID\tORF_type" of every ORF, concatenated
    const int64_t *head_off;  // [n + 1]
    const char *tail;         // "transcript_id\t...\tstart_codon" of every ORF, concatenated
    const int64_t *tail_off;  // [n + 1]
};

// Rows of ORFs first, first+1, ... while they fit into out[0..cap): column order and
// renderings of detect_orfs.py:301-324.  Returns the index of the first ORF NOT written
// (== n_orfs when done); *len = bytes written.  A row that alone exceeds cap is reported
// by returning `first` with *len == 0 and *need = its bound.
inline long long format_rows(const RowInputs &in, long long n_orfs, bool report_all, long long first, char *out,
                             size_t cap, size_t *len, size_t *need)
{
    char *o = out;
    char *const end = out + cap;
    long long i = first;
    *need = 0;
    for (; i < n_orfs; ++i) {
        const bool translating = in.status[i] != 0;
        if (!report_all && !translating) continue;
        const long long beg = in.offsets[i];
        const long long L = in.offsets[i + 1] - beg;
        const size_t hl = (size_t)(in.head_off[i + 1] - in.head_off[i]);
        const size_t tl = (size_t)(in.tail_off[i + 1] - in.tail_off[i]);
        const size_t bound = hl + tl + 160 + list_bound(L);
        if ((size_t)(end - o) < bound) {
            if (o == out) *need = bound;
            break;
        }
        const long long nc = L / 3 > 1 ? L / 3 : 1;  // detect_orfs.py:281
        std::memcpy(o, in.head + in.head_off[i], hl);
        o += hl;
        *o++ = '\t';
        if (translating) {
            std::memcpy(o, "translating", 11);
            o += 11;
        } else {
            std::memcpy(o, "nontranslating", 14);
            o += 14;
        }
        *o++ = '\t';
        o += double_repr(in.phase[i], o);
        *o++ = '\t';
        o += int_str(in.read_count[i], o);
        *o++ = '\t';
        o += int_str(L, o);
        *o++ = '\t';
        o += int_str(in.valid[i], o);
        *o++ = '\t';
        o += double_repr((double)in.valid[i] / (double)nc, o);  // detect_orfs.py:285
        *o++ = '\t';
        o += double_repr((double)in.read_count[i] / (double)nc, o);  // detect_orfs.py:287
        *o++ = '\t';
        std::memcpy(o, in.tail + in.tail_off[i], tl);
        o += tl;
        *o++ = '\t';
        o += int_list_str(in.counts + beg, L, o);
        *o++ = '\n';
    }
    *len = (size_t)(o - out);
    return i;
}

}  // namespace rpfmt
