// rp_index.hpp -- host-side ribotricer index parser.
//
// One pass over the bytes of `{prefix}_candidate_orfs.tsv` (written by prepare-orfs,
// prepare_orfs.py:370-404) that yields everything the GPU path needs, with no per-line
// Python: the exon-interval table for rp_gather_profiles_dev, the (strand, chrom) groups
// and their extents for the dense coverage layout, and the two string tables that
// rp_format_rows_host prints around the numeric columns.
//
// Line semantics follow ORF.from_string / ORF.__init__ (orf.py:88-182):
//   * exactly 11 tab-separated fields, else the reference exits (orf.py:143-152);
//   * field 10 is "s-e,s-e,...": each group splits on '-' into exactly two integers
//     (surrounding whitespace ignored, as int() does: the last one carries the newline);
//   * intervals are sorted by start, stably (orf.py:100);
//   * ORF_ID is recomputed as tid_start_end_length (orf.py:103), index column 0 is ignored;
//   * start_codon is the first three characters of field 9, or None when it is shorter
//     (orf.py:106-118).
// Plain C++17, no HIP.
#pragma once

#include <emmintrin.h>  // SSE2: part of every x86-64
#include <sys/mman.h>

#include "rp_host.hpp"

#include <algorithm>
#include <cstdlib>
#include <new>
#include <cstdint>
#include <cstring>
#include <string>
#include <string_view>
#include <thread>
#include <type_traits>
#include <unordered_map>
#include <utility>
#include <vector>

namespace rpidx {

enum ParseError : int { kOk = 0, kColumns = 1, kCoordinate = 2 };

// std::vector / byte buffer whose resize() leaves new elements uninitialised: the stitched arrays of
// an 11 M-line index are 2 GB that would otherwise be zero-filled by one thread before the parser
// threads overwrite every byte of them.
// Large blocks (>= 8 MiB) are 2 MiB aligned and marked MADV_HUGEPAGE: the parser's output arrays of an
// 11 M-line index are ~2 GB touched for the first time by the parser threads, i.e. half a million
// 4 KiB page faults, or a thousand 2 MiB ones where transparent huge pages are in `madvise` mode
// (the GPU boxes).  RIBOPHASE_HUGEPAGES=0 switches it off (A/B).
inline bool use_hugepages()
{
    static const bool on = [] {
        const char *e = std::getenv("RIBOPHASE_HUGEPAGES");
        return !(e && e[0] == '0');
    }();
    return on;
}

template <typename T>
struct DefaultInitAlloc : std::allocator<T> {
    template <typename U>
    struct rebind {
        using other = DefaultInitAlloc<U>;
    };
    using std::allocator<T>::allocator;
    static constexpr size_t kHuge = size_t(2) << 20, kHugeMin = size_t(8) << 20;
    T *allocate(size_t n)
    {
        const size_t bytes = n * sizeof(T);
        if (bytes >= kHugeMin && use_hugepages()) {
            void *p = nullptr;
            const size_t rounded = (bytes + kHuge - 1) & ~(kHuge - 1);
            if (posix_memalign(&p, kHuge, rounded) != 0) throw std::bad_alloc();
            madvise(p, rounded, MADV_HUGEPAGE);
            return static_cast<T *>(p);
        }
        void *p = std::malloc(bytes ? bytes : 1);
        if (!p) throw std::bad_alloc();
        return static_cast<T *>(p);
    }
    void deallocate(T *p, size_t) noexcept { std::free(p); }
    template <typename U>
    void construct(U *p) noexcept(std::is_nothrow_default_constructible<U>::value)
    {
        ::new (static_cast<void *>(p)) U;
    }
    template <typename U, typename... A>
    void construct(U *p, A &&...a)
    {
        ::new (static_cast<void *>(p)) U(std::forward<A>(a)...);
    }
};
template <typename T>
using Vec = std::vector<T, DefaultInitAlloc<T>>;

inline void append(Vec<char> &dst, std::string_view s) { dst.insert(dst.end(), s.begin(), s.end()); }

struct Index {
    // per ORF
    Vec<int64_t> orf_iv;    // [n + 1] first interval of each ORF
    Vec<int64_t> length;    // [n] sum of interval lengths
    Vec<int32_t> group;     // [n] index into the (strand, chrom) groups
    Vec<uint8_t> reverse;   // [n] 1 for '-' strand
    // per interval, ascending by start inside each ORF, 1-based closed
    Vec<int64_t> iv_start, iv_end;
    // (strand, chrom) groups in order of first appearance
    std::string group_names;             // "strand\tchrom" concatenated
    std::vector<int64_t> group_off;      // [g + 1]
    std::vector<int64_t> group_lo, group_hi;  // extent of the ORFs of each group
    // string tables for the TSV writer
    Vec<char> head;  // "ORF_ID\tORF_type"
    Vec<int64_t> head_off;
    Vec<char> tail;  // "transcript_id\ttranscript_type\tgene_id\tgene_name\tgene_type\tchrom\tstrand\tstart_codon"
    Vec<int64_t> tail_off;
    // diagnostics
    int64_t error_line = 0;  // 1-based line number of the first malformed line
};

inline bool is_space(char c) { return c == ' ' || c == '\n' || c == '\r' || c == '\t' || c == '\v' || c == '\f'; }

// int(text) for the plain cases an index holds: optional whitespace, optional '+', ASCII digits
inline bool parse_int(std::string_view t, int64_t &out)
{
    size_t a = 0, b = t.size();
    while (a < b && is_space(t[a])) ++a;
    while (b > a && is_space(t[b - 1])) --b;
    if (a < b && t[a] == '+') ++a;
    if (a == b) return false;
    int64_t v = 0;
    size_t digits = 0;
    for (size_t k = a; k < b; ++k) {
        const char c = t[k];
        if (c < '0' || c > '9') return false;
        if (v != 0 || c != '0') ++digits;  // (leading zeros do not count)
        if (digits > 18) return false;     // >= 1e18: no genome position; Python's int would go on, this parser reports the line
        v = v * 10 + (c - '0');
    }
    out = v;
    return true;
}

inline void append_int(Vec<char> &s, int64_t v)
{
    char tmp[24];
    int n = 0;
    uint64_t u = v < 0 ? 0ull - (uint64_t)v : (uint64_t)v;
    do {
        tmp[n++] = (char)('0' + u % 10);
        u /= 10;
    } while (u);
    if (v < 0) s.push_back('-');
    while (n) s.push_back(tmp[--n]);
}

// first three characters (UTF-8 code points) of s, or "None" when it has fewer
inline std::string_view start_codon(std::string_view s)
{
    size_t chars = 0, cut = std::string_view::npos;
    for (size_t k = 0; k < s.size(); ++k) {
        if (((unsigned char)s[k] & 0xC0) != 0x80) {
            if (chars == 3) {
                cut = k;
                break;
            }
            ++chars;
        }
    }
    if (chars < 3) return "None";
    return cut == std::string_view::npos ? s : s.substr(0, cut);
}

// Positions of the first <= 11 tabs of [p, e) into tab[], how many were found (12 = more than 11 fields' worth: stop).
// 16 bytes a step where 16 bytes are left; the scalar tail never reads past e.
inline int find_tabs(const char *p, const char *e, const char **tab)
{
    int n = 0;
    const __m128i t = _mm_set1_epi8('\t');
    const char *q = p;
    for (; q + 16 <= e; q += 16) {
        unsigned m = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i *>(q)), t));
        while (m) {
            if (n == 11) return 12;
            tab[n++] = q + __builtin_ctz(m);
            m &= m - 1;
        }
    }
    for (; q < e; ++q)
        if (*q == '\t') {
            if (n == 11) return 12;
            tab[n++] = q;
        }
    return n;
}

// The coordinate column in its canonical form -- "123-456,789-1011" with optional whitespace around a number and an
// optional '+' in front of it, exactly what parse_int accepts -- in one pass.  false: not canonical (the caller takes
// the general path, which also produces the error codes).
inline bool scan_coordinates(const char *p, const char *e, std::vector<std::pair<int64_t, int64_t>> &blocks)
{
    auto number = [&](int64_t &out) {
        while (p < e && is_space(*p)) ++p;
        if (p < e && *p == '+') ++p;
        const char *d0 = p;
        int64_t v = 0;
        while (p < e && (unsigned)(*p - '0') <= 9u) v = v * 10 + (*p++ - '0');
        if (p == d0 || p - d0 > 18) return false;  // (no digit; or long enough to overflow: the general path decides)
        while (p < e && is_space(*p)) ++p;
        out = v;
        return true;
    };
    for (;;) {
        int64_t s = 0, t = 0;
        if (!number(s) || p >= e || *p != '-') return false;
        ++p;
        if (!number(t)) return false;
        blocks.emplace_back(s, t);
        if (p == e) return true;
        if (*p != ',') return false;
        ++p;
    }
}

inline char *put_int(char *dst, int64_t v)  // v >= 0 here (sums and ends of parsed, non-negative numbers)
{
    char tmp[24];
    int n = 0;
    uint64_t u = (uint64_t)v;
    do {
        tmp[n++] = (char)('0' + u % 10);
        u /= 10;
    } while (u);
    while (n) *dst++ = tmp[--n];
    return dst;
}

// Parse `text` (the whole file).  Lines end at '\n' (kept on the line, as Python's file
// iteration does); `skip_header` drops the first line (detect_orfs.py:273).
// One contiguous run of whole lines (group ids are local to the run: order of first appearance).
inline int parse_run(const char *text, size_t len, bool skip_header, Index &ix)
{
    ix = Index();
    ix.orf_iv.push_back(0);
    ix.head_off.push_back(0);
    ix.tail_off.push_back(0);
    ix.group_off.push_back(0);
    {   // size the arrays once: growing them by doubling costs more than the parse itself
        size_t n_lines = 0;
        for (const char *p = text, *e = text + len; p < e;) {
            const char *nl = (const char *)std::memchr(p, '\n', (size_t)(e - p));
            ++n_lines;
            if (!nl) break;
            p = nl + 1;
        }
        ix.orf_iv.reserve(n_lines + 1);
        ix.length.reserve(n_lines);
        ix.group.reserve(n_lines);
        ix.reverse.reserve(n_lines);
        ix.head_off.reserve(n_lines + 1);
        ix.tail_off.reserve(n_lines + 1);
        ix.iv_start.reserve(3 * n_lines);
        ix.iv_end.reserve(3 * n_lines);
        // (head and tail: sized for the worst case below, written through cursors)
    }
    std::unordered_map<std::string, int32_t> groups;
    std::vector<std::pair<int64_t, int64_t>> blocks;
    std::string key;
    std::string_view last_strand, last_chrom;
    int32_t last_gid = -1;
    size_t pos = 0;
    int64_t line_no = 0;
    // The two string tables are written through cursors into buffers sized for the worst case (a line's head is its
    // transcript id and ORF type plus three numbers, its tail a substring of the line plus "None"): no capacity check
    // per field, one memcpy for the tail's seven columns (they are contiguous in the line).  Untouched pages of the
    // over-allocation cost nothing; the vectors are cut to what was written at the end.
    const size_t n_reserved = ix.length.capacity();
    ix.head.resize(len + 72 * n_reserved + 64);
    ix.tail.resize(len + 8 * n_reserved + 64);
    char *head_at = ix.head.data(), *tail_at = ix.tail.data();
    auto finish_tables = [&] {
        ix.head.resize((size_t)(head_at - ix.head.data()));
        ix.tail.resize((size_t)(tail_at - ix.tail.data()));
    };
    while (pos < len) {
        const char *nl = (const char *)std::memchr(text + pos, '\n', len - pos);
        const size_t end = nl ? (size_t)(nl - text) + 1 : len;  // one past the line, newline included
        const char *lp = text + pos, *le = text + end;
        pos = end;
        ++line_no;
        if (skip_header && line_no == 1) continue;
        // split on tabs: 11 fields = exactly 10 tabs
        const char *tab[11];
        const int n_tabs = find_tabs(lp, le, tab);
        if (n_tabs != 10) {
            ix.error_line = line_no;
            finish_tables();
            return kColumns;
        }
        std::string_view f[11];
        f[0] = std::string_view(lp, (size_t)(tab[0] - lp));
        for (int k = 1; k < 10; ++k) f[k] = std::string_view(tab[k - 1] + 1, (size_t)(tab[k] - tab[k - 1] - 1));
        f[10] = std::string_view(tab[9] + 1, (size_t)(le - tab[9] - 1));
        // coordinates: the canonical form in one pass, anything else through the general path (same results, and the errors)
        blocks.clear();
        const std::string_view coord = f[10];
        if (!scan_coordinates(coord.data(), coord.data() + coord.size(), blocks)) {
            blocks.clear();
            size_t g0 = 0;
            for (size_t k = 0; k <= coord.size(); ++k) {
                if (k == coord.size() || coord[k] == ',') {
                    const std::string_view grp = coord.substr(g0, k - g0);
                    g0 = k + 1;
                    const size_t dash = grp.find('-');
                    int64_t s = 0, e = 0;
                    if (dash == std::string_view::npos || grp.find('-', dash + 1) != std::string_view::npos ||
                        !parse_int(grp.substr(0, dash), s) || !parse_int(grp.substr(dash + 1), e)) {
                        ix.error_line = line_no;
                        finish_tables();
                        return kCoordinate;
                    }
                    blocks.emplace_back(s, e);
                }
            }
        }
        bool ascending = true;
        for (size_t k = 1; k < blocks.size(); ++k) ascending = ascending && blocks[k - 1].first <= blocks[k].first;
        if (!ascending)
            std::stable_sort(blocks.begin(), blocks.end(), [](const auto &x, const auto &y) { return x.first < y.first; });
        // A block whose end lies below its start is parsable (orf.py:165-170 does not look): it counts with its raw
        // -- negative -- size in the ORF id (orf.py:103) and contributes no position to the profile (detect_orfs.py:177:
        // range(start, end + 1) is empty).  Such a block gets no interval here; an ORF made of nothing else has none
        // and the profile length 0.  id_length = the sum the id prints, length = the positions the profile holds.
        int64_t length = 0, id_length = 0, lo = INT64_MAX, hi = INT64_MIN;
        for (const auto &b : blocks) {
            id_length += b.second - b.first + 1;
            if (b.second < b.first) continue;
            ix.iv_start.push_back(b.first);
            ix.iv_end.push_back(b.second);
            length += b.second - b.first + 1;
            lo = std::min(lo, b.first);
            hi = std::max(hi, b.second);
        }
        const int64_t first = blocks.front().first, last = blocks.back().second;
        ix.orf_iv.push_back((int64_t)ix.iv_start.size());
        ix.length.push_back(length);
        const std::string_view chrom = f[7], strand = f[8];
        ix.reverse.push_back(strand == "-" ? 1 : 0);
        // (strand, chrom) group; an index lists the ORFs of a transcript -- of a chromosome -- together,
        // so the previous line's group is nearly always this line's: no hashing then
        int32_t gid;
        if (last_gid >= 0 && strand == last_strand && chrom == last_chrom) {
            gid = last_gid;
            ix.group_lo[gid] = std::min(ix.group_lo[gid], lo);
            ix.group_hi[gid] = std::max(ix.group_hi[gid], hi);
        } else {
            key.assign(strand);
            key.push_back('\t');
            key.append(chrom);
            auto it = groups.find(key);
            if (it == groups.end()) {
                gid = (int32_t)groups.size();
                groups.emplace(key, gid);
                ix.group_names.append(key);
                ix.group_off.push_back((int64_t)ix.group_names.size());
                ix.group_lo.push_back(lo);  // (INT64_MAX / INT64_MIN while the group has no interval: close_extents)
                ix.group_hi.push_back(hi);
            } else {
                gid = it->second;
                ix.group_lo[gid] = std::min(ix.group_lo[gid], lo);
                ix.group_hi[gid] = std::max(ix.group_hi[gid], hi);
            }
            last_gid = gid;
            last_strand = strand;  // (views into `text`, which outlives the run)
            last_chrom = chrom;
        }
        ix.group.push_back(gid);
        // head: ORF_ID \t ORF_type   (ORF_ID = transcript_first_last_length, orf.py:103)
        std::memcpy(head_at, f[2].data(), f[2].size());
        head_at += f[2].size();
        *head_at++ = '_';
        if (first < 0 || last < 0 || id_length < 0) {  // (a negative sum: blocks with end < start, e.g. tx_10_5_-4)
            Vec<char> tmp;
            append_int(tmp, first);
            tmp.push_back('_');
            append_int(tmp, last);
            tmp.push_back('_');
            append_int(tmp, id_length);
            std::memcpy(head_at, tmp.data(), tmp.size());
            head_at += tmp.size();
        } else {
            head_at = put_int(head_at, first);
            *head_at++ = '_';
            head_at = put_int(head_at, last);
            *head_at++ = '_';
            head_at = put_int(head_at, id_length);
        }
        *head_at++ = '\t';
        std::memcpy(head_at, f[1].data(), f[1].size());
        head_at += f[1].size();
        ix.head_off.push_back((int64_t)(head_at - ix.head.data()));
        // tail: fields 2..8 -- contiguous in the line, their tabs included -- a tab, then the start codon
        const size_t seven = (size_t)(tab[8] + 1 - f[2].data());  // through the tab behind field 8
        std::memcpy(tail_at, f[2].data(), seven);
        tail_at += seven;
        const std::string_view codon = start_codon(f[9]);
        std::memcpy(tail_at, codon.data(), codon.size());
        tail_at += codon.size();
        ix.tail_off.push_back((int64_t)(tail_at - ix.tail.data()));
    }
    finish_tables();
    return kOk;
}

inline int64_t count_lines(const char *text, size_t len)
{
    int64_t n = 0;
    for (const char *p = text, *e = text + len; p < e;) {
        const char *nl = (const char *)std::memchr(p, '\n', (size_t)(e - p));
        ++n;
        if (!nl) break;
        p = nl + 1;
    }
    return n;
}

// A (strand, chromosome) whose ORFs hold no position at all (every block with end < start) leaves parse_run with the
// extent (INT64_MAX, INT64_MIN): one position, never looked up, stands in for it.
inline void close_extents(Index &ix)
{
    for (size_t g = 0; g < ix.group_lo.size(); ++g)
        if (ix.group_lo[g] > ix.group_hi[g]) ix.group_lo[g] = ix.group_hi[g] = 1;
}

// The whole index text: cut into runs of whole lines, one per thread, parsed independently and
// stitched together in file order -- same arrays, same group numbering (first appearance in the
// file) and the same first malformed line as one sequential pass.  The stitch runs on the same
// threads: every run copies itself to its place in the final arrays.
inline int parse(const char *text, size_t len, bool skip_header, Index &ix, int threads = 0)
{
    if (threads <= 0) {
        // twice the usable cores: half of the parse is page faults on the output arrays, which overlap
        // (16 usable cores: 16 threads 0.57 s, 24-32 threads 0.46-0.50 s, profiles/archive/r03_index_parse_threads.txt)
        threads = 2 * rphost::usable_threads();
        threads = threads > 32 ? 32 : threads;
    }
    const size_t min_run = (size_t)4 << 20;
    if ((size_t)threads > len / min_run) threads = (int)(len / min_run);
    if (threads <= 1) {
        const int rc = parse_run(text, len, skip_header, ix);
        close_extents(ix);
        return rc;
    }
    std::vector<size_t> cut(threads + 1, len);
    cut[0] = 0;
    for (int t = 1; t < threads; ++t) {  // the first line start at or after t/threads of the text
        size_t p = len / threads * t;
        if (p < cut[t - 1]) p = cut[t - 1];
        const char *nl = p == 0 ? nullptr : (const char *)std::memchr(text + p - 1, '\n', len - (p - 1));
        cut[t] = p == 0 ? 0 : (nl ? (size_t)(nl - text) + 1 : len);
    }
    std::vector<Index> part(threads);
    std::vector<int> rc(threads, kOk);
    {
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; ++t)
            pool.emplace_back([&, t] { rc[t] = parse_run(text + cut[t], cut[t + 1] - cut[t], skip_header && t == 0, part[t]); });
        for (auto &th : pool) th.join();
    }
    ix = Index();
    for (int t = 0; t < threads; ++t) {  // the first malformed line of the file
        if (rc[t] != kOk) {
            ix.error_line = count_lines(text, cut[t]) + part[t].error_line;
            return rc[t];
        }
    }
    // where every run goes
    std::vector<size_t> at_orf(threads + 1, 0), at_iv(threads + 1, 0), at_head(threads + 1, 0), at_tail(threads + 1, 0);
    for (int t = 0; t < threads; ++t) {
        at_orf[t + 1] = at_orf[t] + part[t].length.size();
        at_iv[t + 1] = at_iv[t] + part[t].iv_start.size();
        at_head[t + 1] = at_head[t] + part[t].head.size();
        at_tail[t + 1] = at_tail[t] + part[t].tail.size();
    }
    const size_t n = at_orf[threads];
    ix.orf_iv.resize(n + 1);
    ix.length.resize(n);
    ix.group.resize(n);
    ix.reverse.resize(n);
    ix.iv_start.resize(at_iv[threads]);
    ix.iv_end.resize(at_iv[threads]);
    ix.head.resize(at_head[threads]);
    ix.tail.resize(at_tail[threads]);
    ix.head_off.resize(n + 1);
    ix.tail_off.resize(n + 1);
    ix.orf_iv[0] = 0;
    ix.head_off[0] = 0;
    ix.tail_off[0] = 0;
    ix.group_off.push_back(0);
    // the runs' groups, in file order of first appearance, against the file-wide numbering (a few dozen keys)
    std::unordered_map<std::string, int32_t> groups;
    std::vector<std::vector<int32_t>> remap(threads);
    for (int t = 0; t < threads; ++t) {
        const Index &p = part[t];
        remap[t].resize(p.group_lo.size());
        for (size_t g = 0; g < remap[t].size(); ++g) {
            const std::string key = p.group_names.substr((size_t)p.group_off[g], (size_t)(p.group_off[g + 1] - p.group_off[g]));
            auto it = groups.find(key);
            if (it == groups.end()) {
                remap[t][g] = (int32_t)groups.size();
                groups.emplace(key, remap[t][g]);
                ix.group_names.append(key);
                ix.group_off.push_back((int64_t)ix.group_names.size());
                ix.group_lo.push_back(p.group_lo[g]);
                ix.group_hi.push_back(p.group_hi[g]);
            } else {
                remap[t][g] = it->second;
                ix.group_lo[remap[t][g]] = std::min(ix.group_lo[remap[t][g]], p.group_lo[g]);
                ix.group_hi[remap[t][g]] = std::max(ix.group_hi[remap[t][g]], p.group_hi[g]);
            }
        }
    }
    {
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; ++t)
            pool.emplace_back([&, t] {
                Index &p = part[t];
                const size_t o = at_orf[t], m = p.length.size();
                for (size_t k = 0; k < m; ++k) {
                    ix.orf_iv[o + 1 + k] = p.orf_iv[k + 1] + (int64_t)at_iv[t];
                    ix.head_off[o + 1 + k] = p.head_off[k + 1] + (int64_t)at_head[t];
                    ix.tail_off[o + 1 + k] = p.tail_off[k + 1] + (int64_t)at_tail[t];
                    ix.group[o + k] = remap[t][(size_t)p.group[k]];
                }
                if (m) {
                    std::memcpy(ix.length.data() + o, p.length.data(), m * sizeof(int64_t));
                    std::memcpy(ix.reverse.data() + o, p.reverse.data(), m);
                }
                if (!p.iv_start.empty()) {
                    std::memcpy(ix.iv_start.data() + at_iv[t], p.iv_start.data(), p.iv_start.size() * sizeof(int64_t));
                    std::memcpy(ix.iv_end.data() + at_iv[t], p.iv_end.data(), p.iv_end.size() * sizeof(int64_t));
                }
                if (!p.head.empty()) std::memcpy(ix.head.data() + at_head[t], p.head.data(), p.head.size());
                if (!p.tail.empty()) std::memcpy(ix.tail.data() + at_tail[t], p.tail.data(), p.tail.size());
                p = Index();  // release the run's copy as soon as it is stitched in
            });
        for (auto &th : pool) th.join();
    }
    close_extents(ix);
    return kOk;
}

}  // namespace rpidx
