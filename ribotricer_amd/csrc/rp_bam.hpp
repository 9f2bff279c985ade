// rp_bam.hpp -- pysam-free BAM front end: the 5'-end histogram of
// split_bam (ribotricer/bam.py:33-153) straight from the BGZF file, as columns.
//
// Host code (no GPU involved): BGZF blocks are inflated with zlib, the BAM records are walked
// in file order and every read goes through the reference's decision list:
//   qcfail (0x200) -> duplicate (0x400) -> secondary (0x100) -> unmapped (0x4) -> not uniquely
//   mapped (is_read_uniq_mapping, common.py:33-70: NH tag == 1, or -- without an NH tag --
//   MAPQ == 255; everything else counts as "multi", the undecidable case included because the
//   caller tests `not is_read_uniq_mapping(read)`)                                bam.py:77-95
//   aligned length = number of reference positions under M/=/X operations
//   (len(read.get_reference_positions()), bam.py:99-102), filtered by --read_lengths
//   strand and 5' end by protocol (bam.py:108-128): forward keeps the mapping strand, reverse
//   flips it; the 5' end is the first aligned reference position on '+', the last on '-'
//   key (length, strand, chrom, pos + 1) += 1                                      bam.py:130-135
// The keys are packed into 64-bit words, sorted and run-length counted: the result is the
// reference's nested Counter as five columns, sorted by (length, strand, chrom, pos).
#pragma once

#include <zlib.h>

#include "rp_host.hpp"

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <thread>
#include <vector>

namespace rpbam {

enum { kOk = 0, kOpen = 1, kFormat = 2, kInflate = 3, kTooManyRefs = 4 };

struct Split {
    // columns (one row per distinct key)
    std::vector<int32_t> length;
    std::vector<uint8_t> strand;  // 0 '+', 1 '-'
    std::vector<int32_t> chrom;   // reference id of the BAM header
    std::vector<int64_t> pos;     // 1-based
    std::vector<int64_t> count;
    // reference names of the header, '\n'-joined offsets
    std::string ref_names;
    std::vector<int64_t> ref_off;
    // aligned lengths in the order the first counted read of each was met: the insertion order of
    // the reference's read_length_counts dict, which downstream code iterates (metagene.py:296-312)
    std::vector<int32_t> length_order;
    // the counters of the summary file (bam.py:139-145)
    int64_t total = 0, valid = 0, qcfail = 0, duplicate = 0, secondary = 0, unmapped = 0, multi = 0;
    std::string error;
};

// ---- BGZF -> contiguous byte stream ------------------------------------------------------------
// Blocks are read from the file in order, kBatch at a time, inflated on up to 8 threads (a BGZF
// block is an independent raw-deflate stream of <= 64 KiB) and handed out in file order; an error
// surfaces when the reader reaches the block it belongs to, exactly as a sequential reader's would.
class BgzfReader {
public:
    explicit BgzfReader(FILE *fh) : fh_(fh)
    {
        threads_ = rphost::usable_threads();
        threads_ = threads_ > 8 ? 8 : threads_;
    }
    // read exactly n bytes of the uncompressed stream; false at a clean EOF before the first byte
    int read(void *dst, size_t n, bool *eof)
    {
        unsigned char *out = static_cast<unsigned char *>(dst);
        size_t got = 0;
        *eof = false;
        while (got < n) {
            if (cur_ == nullptr || at_ == cur_->size()) {
                const int rc = next_block();
                if (rc != kOk) return rc;
                if (cur_ == nullptr) {  // end of file
                    if (got == 0) {
                        *eof = true;
                        return kOk;
                    }
                    return kFormat;  // truncated in the middle of an item
                }
                continue;
            }
            const size_t take = std::min(n - got, cur_->size() - at_);
            memcpy(out + got, cur_->data() + at_, take);
            at_ += take;
            got += take;
        }
        return kOk;
    }

    // n bytes of the stream WITHOUT a copy when they lie inside the current block (the common case
    // for alignment records); nullptr otherwise -- then read() assembles them
    const unsigned char *take(size_t n)
    {
        if (cur_ == nullptr || cur_->size() - at_ < n) return nullptr;
        const unsigned char *p = cur_->data() + at_;
        at_ += n;
        return p;
    }

private:
    static constexpr size_t kBatch = 128;  // blocks per refill (<= 8 MiB of text)
    struct Block {
        std::vector<unsigned char> comp, text;
        unsigned isize = 0;
        int rc = kOk;
    };

    // one block's header + payload from the file: kOk, or kFormat; *end = clean end of file
    int fetch(Block &blk, bool *end)
    {
        *end = false;
        unsigned char hdr[18];
        const size_t h = fread(hdr, 1, 18, fh_);
        if (h == 0) {
            *end = true;
            return kOk;
        }
        if (h != 18 || hdr[0] != 31 || hdr[1] != 139 || hdr[2] != 8 || !(hdr[3] & 4)) return kFormat;
        const unsigned xlen = hdr[10] | (hdr[11] << 8);
        // the BC subfield is the first one in every BGZF writer in use; find it anyway
        std::vector<unsigned char> extra(xlen);
        memcpy(extra.data(), hdr + 12, std::min<size_t>(6, xlen));
        if (xlen > 6 && fread(extra.data() + 6, 1, xlen - 6, fh_) != xlen - 6) return kFormat;
        int bsize = -1;
        for (unsigned p = 0; p + 4 <= xlen;) {
            const unsigned slen = extra[p + 2] | (extra[p + 3] << 8);
            if (extra[p] == 'B' && extra[p + 1] == 'C' && slen == 2 && p + 6 <= xlen) bsize = extra[p + 4] | (extra[p + 5] << 8);
            p += 4 + slen;
        }
        if (bsize < 0) return kFormat;
        const long payload = (long)bsize + 1 - 12 - (long)xlen - 8;  // compressed bytes of this block
        if (payload < 0 || payload > 65536) return kFormat;  // (BSIZE is 16 bits: a block is <= 64 KiB)
        blk.comp.resize((size_t)payload + 8);
        if (fread(blk.comp.data(), 1, blk.comp.size(), fh_) != blk.comp.size()) return kFormat;
        const unsigned char *t = blk.comp.data() + payload;
        blk.isize = t[4] | (t[5] << 8) | (t[6] << 16) | ((unsigned)t[7] << 24);
        // ISIZE comes from the file: the BGZF format caps the uncompressed size of a block at 64 KiB,
        // anything larger is a damaged trailer (and would otherwise size the inflate buffer)
        if (blk.isize > 65536) return kFormat;
        return kOk;
    }

    static void inflate_block(Block &blk)
    {
        blk.text.resize(blk.isize);
        if (blk.isize == 0) return;
        z_stream zs;
        memset(&zs, 0, sizeof(zs));
        if (inflateInit2(&zs, -15) != Z_OK) {
            blk.rc = kInflate;
            return;
        }
        zs.next_in = blk.comp.data();
        zs.avail_in = (unsigned)(blk.comp.size() - 8);
        zs.next_out = blk.text.data();
        zs.avail_out = blk.isize;
        const int rc = inflate(&zs, Z_FINISH);
        inflateEnd(&zs);
        if (rc != Z_STREAM_END || zs.total_out != blk.isize) blk.rc = kInflate;
    }

    // fill the batch: read until kBatch blocks, the end of the file, or a malformed block (kept as the batch's last)
    void refill()
    {
        batch_.clear();
        next_ = 0;
        while (batch_.size() < kBatch && !done_) {
            batch_.emplace_back();
            bool end = false;
            const int rc = fetch(batch_.back(), &end);
            if (end) {
                batch_.pop_back();
                done_ = true;
            } else if (rc != kOk) {
                batch_.back().rc = rc;
                done_ = true;
            }
        }
        const size_t n = batch_.size();
        const int workers = (int)std::min<size_t>((size_t)threads_, n / 4);  // a few blocks: not worth a thread
        auto work = [&](size_t first, size_t step) {
            for (size_t k = first; k < n; k += step)
                if (batch_[k].rc == kOk) inflate_block(batch_[k]);
        };
        if (workers <= 1) {
            work(0, 1);
        } else {
            std::vector<std::thread> pool;
            for (int t = 1; t < workers; ++t) pool.emplace_back(work, (size_t)t, (size_t)workers);
            work(0, (size_t)workers);
            for (auto &th : pool) th.join();
        }
    }

    // cur_ = the next non-empty block (the EOF marker is an empty one), or nullptr at the end of the file
    int next_block()
    {
        cur_ = nullptr;
        at_ = 0;
        for (;;) {
            if (next_ == batch_.size()) {
                if (done_) return kOk;
                refill();
                if (batch_.empty()) return kOk;
            }
            Block &blk = batch_[next_++];
            if (blk.rc != kOk) return blk.rc;
            if (blk.text.empty()) continue;
            cur_ = &blk.text;
            return kOk;
        }
    }
    FILE *fh_;
    std::vector<Block> batch_;
    size_t next_ = 0;
    const std::vector<unsigned char> *cur_ = nullptr;
    size_t at_ = 0;
    bool done_ = false;
    int threads_ = 1;
};

inline uint32_t le32(const unsigned char *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint16_t le16(const unsigned char *p) { return (uint16_t)(p[0] | (p[1] << 8)); }

// The aux block of a record, as far as the decision list needs it.  The reference asks
// `dict(read.get_tags())["NH"] == 1` (common.py:54-58): the LAST NH tag of the record decides, and a
// tag of any type takes part in the comparison -- an integer or a float equal to 1 is unique, a
// character or string value never is.  Also found here: the real CIGAR of a record with more than
// 65 535 operations, which the BAM format moves into a CG:B,I tag (SAM spec 4.2.2; htslib swaps
// it back in before pysam sees the read).
struct AuxScan {
    bool has_nh = false;
    bool nh_is_one = false;
    const unsigned char *cg = nullptr;  // CG:B,I payload (little-endian uint32 operations)
    uint32_t cg_ops = 0;
};

inline void scan_aux(const unsigned char *aux, const unsigned char *end, AuxScan *out)
{
    while (aux + 3 <= end) {
        const unsigned char t0 = aux[0], t1 = aux[1], ty = aux[2];
        aux += 3;
        const bool is_nh = t0 == 'N' && t1 == 'H';
        auto integer = [&](size_t n, int64_t v) {
            if (is_nh) {
                out->has_nh = true;
                out->nh_is_one = v == 1;
            }
            aux += n;
        };
        auto other = [&]() {  // a value Python would not find equal to 1
            if (is_nh) {
                out->has_nh = true;
                out->nh_is_one = false;
            }
        };
        switch (ty) {
            case 'A': if (aux + 1 > end) return; other(); aux += 1; break;
            case 'c': if (aux + 1 > end) return; integer(1, (int8_t)aux[0]); break;
            case 'C': if (aux + 1 > end) return; integer(1, aux[0]); break;
            case 's': if (aux + 2 > end) return; integer(2, (int16_t)le16(aux)); break;
            case 'S': if (aux + 2 > end) return; integer(2, le16(aux)); break;
            case 'i': if (aux + 4 > end) return; integer(4, (int32_t)le32(aux)); break;
            case 'I': if (aux + 4 > end) return; integer(4, le32(aux)); break;
            case 'f': {
                if (aux + 4 > end) return;
                float f;
                const uint32_t bits = le32(aux);
                memcpy(&f, &bits, 4);
                if (is_nh) {
                    out->has_nh = true;
                    out->nh_is_one = f == 1.0f;
                }
                aux += 4;
                break;
            }
            case 'Z':
            case 'H':
                other();
                while (aux < end && *aux) ++aux;
                ++aux;
                break;
            case 'B': {
                if (aux + 5 > end) return;
                const unsigned char sub = aux[0];
                const uint32_t cnt = le32(aux + 1);
                const size_t w = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
                if ((size_t)(end - (aux + 5)) < (size_t)cnt * w) return;
                other();
                if (t0 == 'C' && t1 == 'G' && sub == 'I') {
                    out->cg = aux + 5;
                    out->cg_ops = cnt;
                }
                aux += 5 + (size_t)cnt * w;
                break;
            }
            default: return;  // unknown type: stop scanning
        }
    }
}

// protocol: 0 = forward, 1 = reverse (bam.py:108-128).  read_lengths == nullptr: every length.
// ascending sort of the 64-bit keys: LSD radix, one pass per byte that differs somewhere
inline void sort_keys(std::vector<uint64_t> &keys)
{
    if (keys.size() < 4096) {
        std::sort(keys.begin(), keys.end());
        return;
    }
    uint64_t all_or = 0, all_and = ~0ull;
    for (uint64_t k : keys) {
        all_or |= k;
        all_and &= k;
    }
    const uint64_t varies = all_or ^ all_and;
    std::vector<uint64_t> other(keys.size());
    uint64_t *src = keys.data(), *dst = other.data();
    for (int byte = 0; byte < 8; ++byte) {
        if (((varies >> (8 * byte)) & 0xff) == 0) continue;
        size_t count[256] = {0};
        for (size_t i = 0; i < keys.size(); ++i) ++count[(src[i] >> (8 * byte)) & 0xff];
        size_t at = 0;
        for (int b = 0; b < 256; ++b) {
            const size_t c = count[b];
            count[b] = at;
            at += c;
        }
        for (size_t i = 0; i < keys.size(); ++i) dst[count[(src[i] >> (8 * byte)) & 0xff]++] = src[i];
        std::swap(src, dst);
    }
    if (src != keys.data()) memcpy(keys.data(), src, keys.size() * sizeof(uint64_t));
}

inline int split_bam(const char *path, int protocol, const int32_t *read_lengths, int n_lengths, Split &out)
{
    FILE *fh = fopen(path, "rb");
    if (!fh) {
        out.error = std::string("cannot open ") + path;
        return kOpen;
    }
    BgzfReader rd(fh);
    bool eof = false;
    auto fail = [&](int rc, const char *what) {
        out.error = what;
        fclose(fh);
        return rc;
    };
    unsigned char w[4];
    int rc = rd.read(w, 4, &eof);
    if (rc != kOk || eof || memcmp(w, "BAM\1", 4) != 0) return fail(rc != kOk ? rc : kFormat, "not a BAM file (bad magic)");
    if ((rc = rd.read(w, 4, &eof)) != kOk || eof) return fail(kFormat, "truncated header");
    if (le32(w) > (1u << 30)) return fail(kFormat, "implausible header text length");
    std::vector<unsigned char> tmp(le32(w));
    if (!tmp.empty() && ((rc = rd.read(tmp.data(), tmp.size(), &eof)) != kOk || eof)) return fail(kFormat, "truncated header text");
    if ((rc = rd.read(w, 4, &eof)) != kOk || eof) return fail(kFormat, "truncated header");
    const uint32_t n_ref = le32(w);
    if (n_ref >= (1u << 21)) return fail(kTooManyRefs, "more than 2^21 reference sequences");
    out.ref_off.assign(1, 0);
    for (uint32_t r = 0; r < n_ref; ++r) {
        if ((rc = rd.read(w, 4, &eof)) != kOk || eof) return fail(kFormat, "truncated reference list");
        if (le32(w) == 0 || le32(w) > (1u << 16)) return fail(kFormat, "implausible reference name length");
        tmp.resize(le32(w));
        if ((rc = rd.read(tmp.data(), tmp.size(), &eof)) != kOk || eof) return fail(kFormat, "truncated reference list");
        size_t len = tmp.size();
        while (len > 0 && tmp[len - 1] == 0) --len;
        out.ref_names.append(reinterpret_cast<const char *>(tmp.data()), len);
        out.ref_off.push_back((int64_t)out.ref_names.size());
        if ((rc = rd.read(w, 4, &eof)) != kOk || eof) return fail(kFormat, "truncated reference list");
    }
    // key: length (10 bits) | strand (1) | chrom (21) | pos (32)  -> sorts by (length, strand, chrom, pos)
    std::vector<uint64_t> keys;
    std::vector<unsigned char> rec;
    std::vector<char> length_seen(1024, 0);
    std::map<int64_t, std::map<uint64_t, int64_t>> long_reads;  // aligned length >= 1024 -> key (strand | chrom | pos) -> count
    std::set<int64_t> long_seen;
    for (;;) {
        rc = rd.read(w, 4, &eof);
        if (rc != kOk) return fail(rc, "corrupt BGZF block");
        if (eof) break;
        const uint32_t block = le32(w);
        if (block < 32) return fail(kFormat, "alignment record shorter than its fixed part");
        if (block > (1u << 28)) return fail(kFormat, "implausible alignment record length");
        const unsigned char *r = rd.take(block);
        if (r == nullptr) {  // the record straddles BGZF blocks
            rec.resize(block);
            if ((rc = rd.read(rec.data(), block, &eof)) != kOk || eof) return fail(kFormat, "truncated alignment record");
            r = rec.data();
        }
        out.total += 1;
        const int32_t ref_id = (int32_t)le32(r);
        const int32_t pos0 = (int32_t)le32(r + 4);
        const unsigned l_name = r[8], mapq = r[9];
        const unsigned n_cigar = le16(r + 12), flag = le16(r + 14);
        const uint32_t l_seq = le32(r + 16);
        if (flag & 0x200) { out.qcfail += 1; continue; }
        if (flag & 0x400) { out.duplicate += 1; continue; }
        if (flag & 0x100) { out.secondary += 1; continue; }
        if (flag & 0x4) { out.unmapped += 1; continue; }
        const size_t cigar_at = 32 + (size_t)l_name;
        const size_t aux_at = cigar_at + 4 * (size_t)n_cigar + ((size_t)l_seq + 1) / 2 + l_seq;
        if (aux_at > block) return fail(kFormat, "alignment record fields overrun the record");
        AuxScan aux;
        scan_aux(r + aux_at, r + block, &aux);
        const bool uniq = aux.has_nh ? aux.nh_is_one : mapq == 255;
        if (!uniq) { out.multi += 1; continue; }
        // the CIGAR: in place, or -- behind the <l_seq>S<n>N placeholder of a record with more than
        // 65 535 operations -- in the CG tag
        const unsigned char *cigar = r + cigar_at;
        uint32_t n_ops = n_cigar;
        if (n_cigar == 2 && aux.cg != nullptr && le32(cigar) == ((l_seq << 4) | 4u) && (le32(cigar + 4) & 15u) == 3u) {
            cigar = aux.cg;
            n_ops = aux.cg_ops;
        }
        // reference positions under M / = / X
        int64_t refpos = pos0, first = -1, last = -1, aligned = 0;
        for (uint32_t k = 0; k < n_ops; ++k) {
            const uint32_t c = le32(cigar + 4 * (size_t)k);
            const uint32_t op = c & 15, len = c >> 4;
            if (op == 0 || op == 7 || op == 8) {  // M = X
                if (len > 0) {
                    if (first < 0) first = refpos;
                    last = refpos + len - 1;
                }
                aligned += len;
                refpos += len;
            } else if (op == 2 || op == 3) {  // D N
                refpos += len;
            }
        }
        // no aligned base, or no reference name (pysam's reference_name is None for an id outside the
        // header's list, and bam.py:130 then skips the read), or a negative position: nothing to key
        if (aligned == 0 || ref_id < 0 || (uint32_t)ref_id >= n_ref || pos0 < 0) continue;
        if (read_lengths) {
            bool wanted = false;
            for (int k = 0; k < n_lengths; ++k) wanted = wanted || read_lengths[k] == aligned;
            if (!wanted) continue;
        }
        const bool reverse_map = (flag & 0x10) != 0;
        // forward: strand = mapping strand; reverse: flipped.  5' end: first position on '+', last on '-'
        // of the ASSIGNED strand under forward, and the opposite end under reverse (bam.py:108-128)
        bool minus;
        int64_t five;
        if (protocol == 0) {
            minus = reverse_map;
            five = reverse_map ? last : first;
        } else {
            minus = !reverse_map;
            five = reverse_map ? first : last;
        }
        if (five < 0 || five + 1 >= (1LL << 32) || aligned > INT32_MAX) continue;  // (pos0 is an int32: cannot happen on a well-formed record)
        if (aligned >= 1024) {
            // longer than the packed key's 10-bit length field (never a Ribo-seq footprint, but the reference counts
            // every read: bam.py:99-131): an ordered map, merged behind the packed keys below
            if (long_seen.insert(aligned).second) out.length_order.push_back((int32_t)aligned);
            long_reads[aligned][((uint64_t)(minus ? 1 : 0) << 53) | ((uint64_t)ref_id << 32) | (uint64_t)(five + 1)] += 1;
            out.valid += 1;
            continue;
        }
        if (!length_seen[aligned]) {
            length_seen[aligned] = 1;
            out.length_order.push_back((int32_t)aligned);
        }
        keys.push_back(((uint64_t)aligned << 54) | ((uint64_t)(minus ? 1 : 0) << 53) | ((uint64_t)ref_id << 32) | (uint64_t)(five + 1));
        out.valid += 1;
    }
    fclose(fh);
    sort_keys(keys);
    for (size_t i = 0; i < keys.size();) {
        size_t j = i;
        while (j < keys.size() && keys[j] == keys[i]) ++j;
        const uint64_t k = keys[i];
        out.length.push_back((int32_t)(k >> 54));
        out.strand.push_back((uint8_t)((k >> 53) & 1));
        out.chrom.push_back((int32_t)((k >> 32) & ((1u << 21) - 1)));
        out.pos.push_back((int64_t)(k & 0xffffffffu));
        out.count.push_back((int64_t)(j - i));
        i = j;
    }
    for (const auto &by_length : long_reads)  // ascending length, then (strand, chrom, pos): the packed keys' order continued
        for (const auto &kv : by_length.second) {
            out.length.push_back((int32_t)by_length.first);
            out.strand.push_back((uint8_t)((kv.first >> 53) & 1));
            out.chrom.push_back((int32_t)((kv.first >> 32) & ((1u << 21) - 1)));
            out.pos.push_back((int64_t)(kv.first & 0xffffffffu));
            out.count.push_back(kv.second);
        }
    return kOk;
}

}  // namespace rpbam
