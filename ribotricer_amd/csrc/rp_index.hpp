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

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <vector>

namespace rpidx {

enum ParseError : int { kOk = 0, kColumns = 1, kCoordinate = 2 };

struct Index {
    // per ORF
    std::vector<int64_t> orf_iv;    // [n + 1] first interval of each ORF
    std::vector<int64_t> length;    // [n] sum of interval lengths
    std::vector<int32_t> group;     // [n] index into the (strand, chrom) groups
    std::vector<uint8_t> reverse;   // [n] 1 for '-' strand
    // per interval, ascending by start inside each ORF, 1-based closed
    std::vector<int64_t> iv_start, iv_end;
    // (strand, chrom) groups in order of first appearance
    std::string group_names;             // "strand\tchrom" concatenated
    std::vector<int64_t> group_off;      // [g + 1]
    std::vector<int64_t> group_lo, group_hi;  // extent of the ORFs of each group
    // string tables for the TSV writer
    std::string head;  // "ORF_ID\tORF_type"
    std::vector<int64_t> head_off;
    std::string tail;  // "transcript_id\ttranscript_type\tgene_id\tgene_name\tgene_type\tchrom\tstrand\tstart_codon"
    std::vector<int64_t> tail_off;
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
    for (size_t k = a; k < b; ++k) {
        const char c = t[k];
        if (c < '0' || c > '9') return false;
        v = v * 10 + (c - '0');
    }
    out = v;
    return true;
}

inline void append_int(std::string &s, int64_t v)
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
        ix.head.reserve(len / 2);
        ix.tail.reserve(len);
    }
    std::unordered_map<std::string, int32_t> groups;
    std::vector<std::pair<int64_t, int64_t>> blocks;
    std::string key;
    size_t pos = 0;
    int64_t line_no = 0;
    while (pos < len) {
        const char *nl = (const char *)std::memchr(text + pos, '\n', len - pos);
        const size_t end = nl ? (size_t)(nl - text) + 1 : len;  // one past the line, newline included
        const std::string_view line(text + pos, end - pos);
        pos = end;
        ++line_no;
        if (skip_header && line_no == 1) continue;
        // split on tabs
        std::string_view f[11];
        int nf = 0;
        size_t a = 0;
        bool too_many = false;
        for (size_t k = 0; k <= line.size(); ++k) {
            if (k == line.size() || line[k] == '\t') {
                if (nf == 11) {
                    too_many = true;
                    break;
                }
                f[nf++] = line.substr(a, k - a);
                a = k + 1;
            }
        }
        if (too_many || nf != 11) {
            ix.error_line = line_no;
            return kColumns;
        }
        // coordinates
        blocks.clear();
        const std::string_view coord = f[10];
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
                    return kCoordinate;
                }
                blocks.emplace_back(s, e);
            }
        }
        std::stable_sort(blocks.begin(), blocks.end(), [](const auto &x, const auto &y) { return x.first < y.first; });
        int64_t length = 0;
        for (const auto &b : blocks) {
            ix.iv_start.push_back(b.first);
            ix.iv_end.push_back(b.second);
            length += b.second - b.first + 1;
        }
        const int64_t first = blocks.front().first, last = blocks.back().second;
        ix.orf_iv.push_back((int64_t)ix.iv_start.size());
        ix.length.push_back(length);
        const std::string_view chrom = f[7], strand = f[8];
        ix.reverse.push_back(strand == "-" ? 1 : 0);
        // (strand, chrom) group
        key.assign(strand);
        key.push_back('\t');
        key.append(chrom);
        auto it = groups.find(key);
        int32_t gid;
        if (it == groups.end()) {
            gid = (int32_t)groups.size();
            groups.emplace(key, gid);
            ix.group_names.append(key);
            ix.group_off.push_back((int64_t)ix.group_names.size());
            ix.group_lo.push_back(first);
            ix.group_hi.push_back(last);
        } else {
            gid = it->second;
            ix.group_lo[gid] = std::min(ix.group_lo[gid], first);
            ix.group_hi[gid] = std::max(ix.group_hi[gid], last);
        }
        ix.group.push_back(gid);
        // head: ORF_ID \t ORF_type
        ix.head.append(f[2]);
        ix.head.push_back('_');
        append_int(ix.head, first);
        ix.head.push_back('_');
        append_int(ix.head, last);
        ix.head.push_back('_');
        append_int(ix.head, length);
        ix.head.push_back('\t');
        ix.head.append(f[1]);
        ix.head_off.push_back((int64_t)ix.head.size());
        // tail: fields 2..8 then the start codon
        for (int k = 2; k <= 8; ++k) {
            ix.tail.append(f[k]);
            ix.tail.push_back('\t');
        }
        ix.tail.append(start_codon(f[9]));
        ix.tail_off.push_back((int64_t)ix.tail.size());
    }
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

template <typename T>
inline void append_shifted(std::vector<T> &dst, const std::vector<T> &src, size_t skip, T shift)
{
    const size_t at = dst.size();
    dst.resize(at + src.size() - skip);
    for (size_t k = skip; k < src.size(); ++k) dst[at + k - skip] = src[k] + shift;
}

// The whole index text: cut into runs of whole lines, one per thread, parsed independently and
// stitched together in file order -- same arrays, same group numbering (first appearance in the
// file) and the same first malformed line as one sequential pass.
inline int parse(const char *text, size_t len, bool skip_header, Index &ix, int threads = 0)
{
    if (threads <= 0) {
        threads = (int)std::thread::hardware_concurrency();
        threads = threads > 8 ? 8 : (threads < 1 ? 1 : threads);
    }
    const size_t min_run = (size_t)4 << 20;
    if ((size_t)threads > len / min_run) threads = (int)(len / min_run);
    if (threads <= 1) return parse_run(text, len, skip_header, ix);
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
    size_t n = 0, n_iv = 0, n_head = 0, n_tail = 0;
    for (const Index &p : part) {
        n += p.length.size();
        n_iv += p.iv_start.size();
        n_head += p.head.size();
        n_tail += p.tail.size();
    }
    ix.orf_iv.reserve(n + 1);
    ix.length.reserve(n);
    ix.group.reserve(n);
    ix.reverse.reserve(n);
    ix.iv_start.reserve(n_iv);
    ix.iv_end.reserve(n_iv);
    ix.head.reserve(n_head);
    ix.tail.reserve(n_tail);
    ix.head_off.reserve(n + 1);
    ix.tail_off.reserve(n + 1);
    ix.orf_iv.push_back(0);
    ix.head_off.push_back(0);
    ix.tail_off.push_back(0);
    ix.group_off.push_back(0);
    std::unordered_map<std::string, int32_t> groups;
    for (Index &p : part) {
        // this run's groups, in its order of first appearance, against the file-wide numbering
        std::vector<int32_t> remap(p.group_lo.size());
        for (size_t g = 0; g < remap.size(); ++g) {
            const std::string key = p.group_names.substr((size_t)p.group_off[g], (size_t)(p.group_off[g + 1] - p.group_off[g]));
            auto it = groups.find(key);
            if (it == groups.end()) {
                remap[g] = (int32_t)groups.size();
                groups.emplace(key, remap[g]);
                ix.group_names.append(key);
                ix.group_off.push_back((int64_t)ix.group_names.size());
                ix.group_lo.push_back(p.group_lo[g]);
                ix.group_hi.push_back(p.group_hi[g]);
            } else {
                remap[g] = it->second;
                ix.group_lo[remap[g]] = std::min(ix.group_lo[remap[g]], p.group_lo[g]);
                ix.group_hi[remap[g]] = std::max(ix.group_hi[remap[g]], p.group_hi[g]);
            }
        }
        append_shifted(ix.orf_iv, p.orf_iv, 1, (int64_t)ix.iv_start.size());
        append_shifted(ix.head_off, p.head_off, 1, (int64_t)ix.head.size());
        append_shifted(ix.tail_off, p.tail_off, 1, (int64_t)ix.tail.size());
        ix.length.insert(ix.length.end(), p.length.begin(), p.length.end());
        ix.reverse.insert(ix.reverse.end(), p.reverse.begin(), p.reverse.end());
        ix.iv_start.insert(ix.iv_start.end(), p.iv_start.begin(), p.iv_start.end());
        ix.iv_end.insert(ix.iv_end.end(), p.iv_end.begin(), p.iv_end.end());
        ix.head.append(p.head);
        ix.tail.append(p.tail);
        const size_t at = ix.group.size();
        ix.group.resize(at + p.group.size());
        for (size_t k = 0; k < p.group.size(); ++k) ix.group[at + k] = remap[(size_t)p.group[k]];
        p = Index();  // release the run's copy as soon as it is stitched in
    }
    return kOk;
}

}  // namespace rpidx
