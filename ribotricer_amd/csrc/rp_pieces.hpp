// rp_pieces.hpp -- the profile space of an index as PIECES of the dense coverage.
//
// The reference fills every ORF's profile with one dict lookup per nucleotide, exon by exon,
// reversed for '-' strand ORFs (orf_coverage, detect_orfs.py:134-203).  Here the concatenated
// profile space [0, total_nt) of a whole index is described once, from the interval table
// alone, as a sorted list of pieces -- piece j covers profile positions [start_j, start_j+1)
// and position p reads coverage[base_j + p] (forward) or coverage[base_j - p] ('-' strand) --
// and every flat tile of the scorer (rp_tile.hpp) gets a fixed-stride row of its pieces,
// clipped to the tile.  With that a workgroup stages its tile of counts straight from the
// coverage arrays with LDS-DMA, 64 consecutive positions of one piece per instruction:
//   * k_tile_gather writes the staged tile out: the CSR `counts` array (report_all mode);
//   * k_tile_score<true> scores it in place: the profiles are never written to HBM.
// The plan depends on the index (and the coverage layout derived from it) only, so it is
// built once per index and reused for every sample.
#pragma once

#include <cstddef>
#include <cstdint>

#include <hip/hip_runtime.h>

namespace rp {

typedef unsigned long long piece_desc_t;

constexpr int kRowPieces = 256;                    // clipped pieces in a tile's fixed-stride row (2 KiB per 31 KiB tile)
constexpr unsigned long long kPieceNeg = 1ull << 63;  // start word: the piece runs down the coverage array
constexpr long long kMaxCoverage = 1ll << 34;      // a clipped piece keeps its source index in 34 bits

// A clipped piece (8 bytes; 0 = empty slot):
//   bits  0-33  src    coverage index of the piece's first position inside the tile
//   bits 34-46  off    that position's index in the tile's LDS image
//   bits 47-59  n      positions (1 .. kTile + kHalo)
//   bit  60     neg    the source index falls as the position rises
//   bit  61     more   (last slot of the row only) the tile has further pieces: tile_next + kRowPieces ...
constexpr int kPieceOffAt = 34, kPieceNAt = 47, kPieceNegAt = 60, kPieceMoreAt = 61;

// The fast staging path (stage_tile_chunks) cuts a tile's pieces into chunks of <= 64
// positions, at most kMaxChunks of them, addressed with 32-bit byte offsets from one base.
constexpr int kMaxChunks = 352;
constexpr long long kTileSlow = INT64_MIN;  // tile_lo: the row is not eligible (too many pieces / chunks, > 4 GiB apart)

struct PiecePlan {
    const unsigned long long *start;  // [n_pieces + 1] first profile position | kPieceNeg; sentinel total_nt
    const long long *base;            // [n_pieces + 1] coverage index = base + p, or base - p
    const long long *orf_piece;       // [n_orfs + 1]   pieces of ORF i: orf_piece[i] .. orf_piece[i+1]
    const long long *tile_next;       // [n_tiles]      first piece of the tile that is not in its row
    const piece_desc_t *rows;         // [n_tiles][kRowPieces]
    const long long *tile_lo;         // [n_tiles]      lowest coverage index the row reads, minus 64 (kTileSlow: see stage_tile)
    long long n_pieces;
    long long coverage_len;
};

inline size_t piece_plan_bytes(long long n_orfs, long long n_pieces, long long n_tiles)
{
    auto up = [](size_t b) { return (b + 127) & ~(size_t)127; };
    return up((size_t)(n_pieces + 1) * 8) * 2 + up((size_t)(n_orfs + 1) * 8) + up((size_t)n_tiles * 8) * 2 +
           (size_t)n_tiles * kRowPieces * sizeof(piece_desc_t);
}

struct PiecePlanMem {
    unsigned long long *start;
    long long *base;
    long long *orf_piece;
    long long *tile_next;
    long long *tile_lo;
    piece_desc_t *rows;
};

inline PiecePlanMem carve_piece_plan(void *mem, long long n_orfs, long long n_pieces, long long n_tiles)
{
    auto up = [](size_t b) { return (b + 127) & ~(size_t)127; };
    PiecePlanMem m;
    char *p = reinterpret_cast<char *>(mem);
    m.start = reinterpret_cast<unsigned long long *>(p);
    p += up((size_t)(n_pieces + 1) * 8);
    m.base = reinterpret_cast<long long *>(p);
    p += up((size_t)(n_pieces + 1) * 8);
    m.orf_piece = reinterpret_cast<long long *>(p);
    p += up((size_t)(n_orfs + 1) * 8);
    m.tile_next = reinterpret_cast<long long *>(p);
    p += up((size_t)n_tiles * 8);
    m.tile_lo = reinterpret_cast<long long *>(p);
    p += up((size_t)n_tiles * 8);
    m.rows = reinterpret_cast<piece_desc_t *>(p);
    return m;
}

// One thread per ORF: its exon intervals (ascending, orf.py:100) -> pieces in profile order
// (the intervals backwards for a '-' strand ORF).  err bit 0: the intervals do not add up to
// the profile length; bit 1: an interval is empty or reaches outside the coverage array.
__global__ void k_piece_build(const int64_t *__restrict__ iv_start, const int32_t *__restrict__ iv_len,
                              const int64_t *__restrict__ orf_iv, const uint8_t *__restrict__ reverse,
                              const int64_t *__restrict__ offsets, long long n_orfs, long long n_pieces,
                              long long total_nt, long long coverage_len, PiecePlanMem out, int *__restrict__ err)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n_orfs) return;
    if (i == n_orfs) {  // sentinel
        out.start[n_pieces] = (unsigned long long)total_nt;
        out.base[n_pieces] = 0;
        out.orf_piece[n_orfs] = orf_iv[n_orfs];
        if (orf_iv[n_orfs] != n_pieces || orf_iv[0] != 0) atomicOr(err, 1);
        return;
    }
    const long long k0 = orf_iv[i];
    const long long nk = (long long)orf_iv[i + 1] - k0;
    out.orf_piece[i] = k0;
    if (nk < 0 || k0 < 0 || k0 + nk > n_pieces) {
        atomicOr(err, 1);
        return;
    }
    const long long beg = offsets[i];
    const long long len = (long long)offsets[i + 1] - beg;
    const bool rev = reverse[i] != 0;
    long long asc = 0;
    int bad = 0;
    for (long long t = 0; t < nk; ++t) {
        const long long s = iv_start[k0 + t];
        const long long n = iv_len[k0 + t];
        if (n <= 0 || s < 0 || s + n > coverage_len) bad |= 2;
        if (!rev) {
            const long long p0 = beg + asc;
            out.start[k0 + t] = (unsigned long long)p0;
            out.base[k0 + t] = s - p0;
        } else {
            const long long p0 = beg + (len - asc - n);
            out.start[k0 + nk - 1 - t] = (unsigned long long)p0 | kPieceNeg;
            out.base[k0 + nk - 1 - t] = s + n - 1 + p0;
        }
        asc += n;
    }
    if (asc != len) bad |= 1;
    if (bad) atomicOr(err, bad);
}

__device__ __forceinline__ piece_desc_t clip_piece(unsigned long long start_word, unsigned long long next_word,
                                                   long long base, long long t0, long long t_end)
{
    const long long s = (long long)(start_word & ~kPieceNeg);
    const long long e = (long long)(next_word & ~kPieceNeg);
    const bool neg = (start_word & kPieceNeg) != 0;
    const long long a = s > t0 ? s : t0;
    const long long b = e < t_end ? e : t_end;
    if (b <= a) return 0;
    const long long src = neg ? base - a : base + a;
    return (piece_desc_t)src | ((piece_desc_t)(a - t0) << kPieceOffAt) | ((piece_desc_t)(b - a) << kPieceNAt) |
           ((piece_desc_t)(neg ? 1 : 0) << kPieceNegAt);
}

// One workgroup of kRowPieces threads per tile: find the piece that holds the tile's first
// position, clip it and its successors to [t0, t0 + TILE + HALO) and lay them into the row --
// long ones as several slots of <= kSlotPositions positions, so that no thread of the staging
// code has more than 8 chunks to cut.  tile_next[b] = the first piece that did not fit.
constexpr int kSlotPositions = 512;

template <int TILE, int HALO>
__global__ __launch_bounds__(kRowPieces) void k_piece_rows(PiecePlanMem plan, long long n_pieces, long long total_nt)
{
    __shared__ long long s_j0, s_lo, s_hi;
    __shared__ int s_chunks, s_fit, s_more;
    __shared__ int s_slot[kRowPieces], s_nsub[kRowPieces];
    const int t = threadIdx.x;
    const long long b = blockIdx.x;
    const long long t0 = b * (long long)TILE;
    long long t_end = t0 + TILE + HALO;
    if (t_end > total_nt) t_end = total_nt;
    if (t == 0) {  // last j with start_j <= t0 (start_0 == 0 <= t0)
        long long lo = 0, hi = n_pieces;  // start[hi] (sentinel or later piece) > t0 unless t0 >= total_nt
        while (hi - lo > 1) {
            const long long mid = lo + (hi - lo) / 2;
            if ((long long)(plan.start[mid] & ~kPieceNeg) <= t0) lo = mid; else hi = mid;
        }
        s_j0 = lo;
        s_lo = INT64_MAX;
        s_hi = INT64_MIN;
        s_chunks = 0;
    }
    __syncthreads();
    const long long j = s_j0 + t;
    piece_desc_t d = 0;
    if (j < n_pieces) d = clip_piece(plan.start[j], plan.start[j + 1], plan.base[j], t0, t_end);
    const long long src = (long long)(d & ((1ull << kPieceOffAt) - 1));
    const int off = (int)(d >> kPieceOffAt) & 0x1fff;
    const int n = (int)(d >> kPieceNAt) & 0x1fff;
    const bool neg = ((d >> kPieceNegAt) & 1) != 0;
    const int nsub = (n + kSlotPositions - 1) / kSlotPositions;
    s_nsub[t] = nsub;
    plan.rows[b * kRowPieces + t] = 0;
    __syncthreads();
    if (t == 0) {  // slots of the pieces, in order; the first piece whose slots pass the row's end and all after it stay out
        int acc = 0, fit = kRowPieces;
        for (int k = 0; k < kRowPieces; ++k) {
            s_slot[k] = acc;
            acc += s_nsub[k];
            if (acc > kRowPieces && fit == kRowPieces) fit = k;
        }
        int more = 0;
        for (int k = fit; k < kRowPieces; ++k) more |= s_nsub[k];
        const long long after = s_j0 + kRowPieces;  // the candidate pieces end here: does the tile go on?
        if (after < n_pieces && (long long)(plan.start[after] & ~kPieceNeg) < t_end) more = 1;
        s_fit = fit;
        s_more = more != 0;
        plan.tile_next[b] = s_j0 + fit;
    }
    __syncthreads();
    if (t < s_fit && n > 0) {
        for (int k = 0; k < nsub; ++k) {
            const int nk = n - k * kSlotPositions < kSlotPositions ? n - k * kSlotPositions : kSlotPositions;
            const long long sk = neg ? src - (long long)k * kSlotPositions : src + (long long)k * kSlotPositions;
            plan.rows[b * kRowPieces + s_slot[t] + k] = (piece_desc_t)sk | ((piece_desc_t)(off + k * kSlotPositions) << kPieceOffAt) |
                                                        ((piece_desc_t)nk << kPieceNAt) | ((piece_desc_t)(neg ? 1 : 0) << kPieceNegAt);
        }
        // eligibility for the chunk-table staging: the lowest / highest coverage index and the chunk count
        atomicMin(&s_lo, neg ? src - (n - 1) : src);
        atomicMax(&s_hi, neg ? src : src + n - 1);
        atomicAdd(&s_chunks, (n + 63) >> 6);
    }
    __syncthreads();
    if (t == 0) {
        if (s_more) plan.rows[b * kRowPieces + kRowPieces - 1] |= (piece_desc_t)1 << kPieceMoreAt;
        // (a slot boundary may add a chunk to a piece whose length is not a multiple of 64: none -- 512 is)
        const bool fast = !s_more && s_chunks > 0 && s_chunks <= kMaxChunks && s_hi - s_lo < (1ll << 30) - 256;
        plan.tile_lo[b] = fast ? s_lo - 64 : kTileSlow;
    }
}

// ---------------------------------------------------------------------------------------
// Staging a tile: every wave takes the pieces held by the lanes l == wave (mod 4), one after
// the other, 64 positions per LDS-DMA instruction.  The global address is a scalar base plus
// a per-lane constant (lane * 4 going up the coverage array, (63 - lane) * 4 going down), so
// a chunk costs one v_cmp for the ragged end and scalar arithmetic.  Not waited for here.
// ---------------------------------------------------------------------------------------
// One LDS-DMA instruction: lanes [0, live) load 4 bytes each from g + lane offset + OFF into
// lds + OFF + 4 * lane (the instruction offset applies to both sides).
template <int OFF>
__device__ __forceinline__ void dma_chunk(const char *g, unsigned lane_off, int *lds)
{
    typedef const __attribute__((address_space(1))) void *gptr_t;
    typedef __attribute__((address_space(3))) void *lptr_t;
    __builtin_amdgcn_global_load_lds((gptr_t)(g + lane_off), (lptr_t)lds, 4, OFF, 0);
}

// up to 16 consecutive full chunks going UP the coverage array: one address, one M0, immediate offsets
__device__ __forceinline__ void dma_run_up(const char *g, unsigned lane_off, int *lds, int chunks)
{
#define RP_UP(C) if (chunks > C) dma_chunk<256 * C>(g, lane_off, lds)
    RP_UP(0); RP_UP(1); RP_UP(2); RP_UP(3); RP_UP(4); RP_UP(5); RP_UP(6); RP_UP(7);
    RP_UP(8); RP_UP(9); RP_UP(10); RP_UP(11); RP_UP(12); RP_UP(13); RP_UP(14); RP_UP(15);
#undef RP_UP
}

// going DOWN the coverage array the two sides move in opposite directions: the instruction
// offset walks the global side (-256 per chunk), the LDS pointer makes up for it (+512)
__device__ __forceinline__ void dma_run_down(const char *g, unsigned lane_off, int *lds, int chunks)
{
#define RP_DOWN(C) if (chunks > C) dma_chunk<-256 * C>(g, lane_off, lds + 128 * C)
    RP_DOWN(0); RP_DOWN(1); RP_DOWN(2); RP_DOWN(3); RP_DOWN(4); RP_DOWN(5); RP_DOWN(6); RP_DOWN(7);
    RP_DOWN(8); RP_DOWN(9); RP_DOWN(10); RP_DOWN(11); RP_DOWN(12); RP_DOWN(13); RP_DOWN(14); RP_DOWN(15);
#undef RP_DOWN
}

__device__ __forceinline__ void stage_piece(const int32_t *__restrict__ cov, piece_desc_t d, int *s_counts, int lane)
{
    const long long src = (long long)(d & ((1ull << kPieceOffAt) - 1));
    const int off = (int)(d >> kPieceOffAt) & 0x1fff;
    const int n = (int)(d >> kPieceNAt) & 0x1fff;
    const bool neg = ((d >> kPieceNegAt) & 1) != 0;
    int full = n >> 6;
    const int rest = n & 63;
    int *lds = s_counts + off;
    if (!neg) {
        const unsigned up4 = (unsigned)lane * 4u;
        const char *g = reinterpret_cast<const char *>(cov + src);
        while (full > 16) {  // (pieces of more than 1 024 positions)
            dma_run_up(g, up4, lds, 16);
            g += 4096, lds += 1024, full -= 16;
        }
        dma_run_up(g, up4, lds, full);
        if (lane < rest) dma_chunk<0>(g + 256 * full, up4, lds + 64 * full);
    } else {
        const unsigned down4 = (unsigned)(63 - lane) * 4u;
        const char *g = reinterpret_cast<const char *>(cov + src - 63);
        while (full > 16) {
            dma_run_down(g, down4, lds, 16);
            g -= 4096, lds += 1024, full -= 16;
        }
        dma_run_down(g, down4, lds, full);
        if (lane < rest) dma_chunk<0>(g - 256 * full, down4, lds + 64 * full);
    }
}

__device__ __forceinline__ unsigned long long readlane_u64(unsigned long long v, int l)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
    return ((unsigned long long)hi << 32) | lo;
}

// the descriptors one per lane -> this wave stages those of lanes wave, wave + 4, ...
__device__ __forceinline__ void stage_round(const int32_t *__restrict__ cov, piece_desc_t mine, int *s_counts, int lane, int wave)
{
#pragma unroll 1
    for (int l = wave; l < 64; l += 4) {
        const piece_desc_t d = readlane_u64(mine, l);
        if (d != 0) stage_piece(cov, d, s_counts, lane);
    }
}

// The row as two halves of 128 slots; lane l holds slots 2l, 2l + 1 of each half.
__device__ __forceinline__ const uint4 *piece_row_ptr(const PiecePlan &pp, long long b, int lane, int half)
{
    return reinterpret_cast<const uint4 *>(pp.rows + b * kRowPieces + half * (kRowPieces / 2)) + lane;
}

__device__ __forceinline__ void stage_half(const int32_t *__restrict__ cov, uint4 h, int *s_counts, int lane, int wave)
{
    stage_round(cov, ((piece_desc_t)h.y << 32) | h.x, s_counts, lane, wave);
    stage_round(cov, ((piece_desc_t)h.w << 32) | h.z, s_counts, lane, wave);
}

// Stage tile b (positions [t0, min(t0 + TILE + HALO, total_nt))) given this lane's row words.
template <int TILE, int HALO>
__device__ __forceinline__ void stage_tile(const int32_t *__restrict__ cov, const PiecePlan &pp, long long b,
                                           long long total_nt, uint4 lo, uint4 hi, int *s_counts, int lane, int wave)
{
    stage_half(cov, lo, s_counts, lane, wave);
    const unsigned more = (unsigned)__builtin_amdgcn_readlane((int)(hi.w >> (kPieceMoreAt - 32)) & 1, 63);
    if (!more && __builtin_amdgcn_readlane((int)(hi.x | hi.y), 0) == 0) return;  // slots fill in order: <= 128 of them
    stage_half(cov, hi, s_counts, lane, wave);
    if (more) {  // rare: more than kRowPieces pieces in the tile (runs of very short exons / ORFs)
        const long long t0 = b * (long long)TILE;
        long long t_end = t0 + TILE + HALO;
        if (t_end > total_nt) t_end = total_nt;
        long long j = pp.tile_next[b];
        for (;;) {  // wave-uniform
            piece_desc_t d = 0;
            if (j + lane < pp.n_pieces) d = clip_piece(pp.start[j + lane], pp.start[j + lane + 1], pp.base[j + lane], t0, t_end);
            stage_round(cov, d, s_counts, lane, wave);
            j += 64;
            if (__builtin_amdgcn_readlane((int)(d != 0), 63) == 0) break;  // the list ran past the tile
        }
    }
}

// ---------------------------------------------------------------------------------------
// The fast staging path.  The scalar loop above spends ~1 000 cycles per piece on a busy CU
// (two lane reads, a decode and a branch chain per piece, at one instruction per ~10 cycles
// next to three other workgroups' lane runs); here the per-piece work is done by the vector
// unit, one piece per THREAD, and the DMA is issued from straight-line code:
//   1. thread t cuts the row's piece t into chunks of <= 64 positions; a workgroup-wide prefix
//      sum numbers them; each thread writes its chunks into a table in LDS:
//         bits 0-31 byte offset of the chunk's lowest-lane element from cov + tile_lo
//         bits 32-44 LDS index   45-50 64 - positions   51 the source index falls
//   2. wave w, lane i takes chunk 4 i + w and keeps (offset, LDS address | shift << 16 | dir << 24)
//      in two registers;
//   3. 64 unrolled steps -- two lane reads, M0, the EXEC mask of the ragged end, two VALU for
//      the lane's byte offset ((lane or 63 - lane) * 4 + chunk offset), one global_load_lds with
//      a scalar base -- no branches but an exit test every 8 steps.  Lanes past the last chunk
//      hold chunk 0 again (staging a chunk twice is harmless).
// `s_tab`: kMaxChunks 8-byte words, `s_part`: 4 ints of LDS scratch, both dead afterwards.
// ---------------------------------------------------------------------------------------
#define RP_DMA_STEP(I)                                          \
    "v_readlane_b32 %[so], %[w0], " #I "\n\t"                   \
    "v_readlane_b32 %[s1], %[w1], " #I "\n\t"                   \
    "s_and_b32 m0, %[s1], 0xffff\n\t"                           \
    "s_lshr_b32 %[st], %[s1], 16\n\t"                           \
    "s_lshr_b64 exec, -1, %[st]\n\t"                            \
    "s_lshr_b32 %[sd], %[st], 8\n\t"                            \
    "v_mad_i32_i24 %[vt], %[sd], %[vdelta], %[vup]\n\t"         \
    "v_add_u32 %[vt], %[so], %[vt]\n\t"                         \
    "global_load_lds_dword %[vt], %[base]\n\t"
#define RP_DMA_STEP8(A, B, C, D, E, F, G, H, LIM)               \
    RP_DMA_STEP(A) RP_DMA_STEP(B) RP_DMA_STEP(C) RP_DMA_STEP(D) \
    RP_DMA_STEP(E) RP_DMA_STEP(F) RP_DMA_STEP(G) RP_DMA_STEP(H) \
    "s_cmp_le_u32 %[steps], " #LIM "\n\t"                       \
    "s_cbranch_scc1 1f\n\t"

__device__ __forceinline__ void issue_chunks(const int32_t *base, unsigned w0, unsigned w1, int steps, int lane)
{
    const int vup = lane * 4, vdelta = (63 - 2 * lane) * 4;  // vup + vdelta = (63 - lane) * 4
    unsigned so, s1, st, sd;
    int vt;
    asm volatile(
        RP_DMA_STEP8(0, 1, 2, 3, 4, 5, 6, 7, 8)
        RP_DMA_STEP8(8, 9, 10, 11, 12, 13, 14, 15, 16)
        RP_DMA_STEP8(16, 17, 18, 19, 20, 21, 22, 23, 24)
        RP_DMA_STEP8(24, 25, 26, 27, 28, 29, 30, 31, 32)
        RP_DMA_STEP8(32, 33, 34, 35, 36, 37, 38, 39, 40)
        RP_DMA_STEP8(40, 41, 42, 43, 44, 45, 46, 47, 48)
        RP_DMA_STEP8(48, 49, 50, 51, 52, 53, 54, 55, 56)
        RP_DMA_STEP8(56, 57, 58, 59, 60, 61, 62, 63, 64)
        "1:\n\t"
        "s_mov_b64 exec, -1"
        : [so] "=&s"(so), [s1] "=&s"(s1), [st] "=&s"(st), [sd] "=&s"(sd), [vt] "=&v"(vt)
        : [w0] "v"(w0), [w1] "v"(w1), [vdelta] "v"(vdelta), [vup] "v"(vup), [base] "s"(base), [steps] "s"(steps)
        : "memory", "scc");
}
#undef RP_DMA_STEP8
#undef RP_DMA_STEP

// all kRowPieces threads of the workgroup; `mine` = the row's piece threadIdx.x
__device__ __forceinline__ void stage_tile_chunks(const int32_t *__restrict__ cov, long long tile_lo, piece_desc_t mine,
                                                  unsigned long long *s_tab, int *s_part, int *s_counts, int tid)
{
    const int lane = tid & 63, wave = tid >> 6;
    const long long src = (long long)(mine & ((1ull << kPieceOffAt) - 1));
    const int off = (int)(mine >> kPieceOffAt) & 0x1fff;
    const int n = (int)(mine >> kPieceNAt) & 0x1fff;
    const unsigned neg = (unsigned)(mine >> kPieceNegAt) & 1u;
    const int nch = (n + 63) >> 6;
    // exclusive prefix of the chunk counts over the workgroup
    int incl = nch;
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, true);
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, true);
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xf, 0xf, true);
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xf, 0xf, true);
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xa, 0xf, false);
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xc, 0xf, false);
    if (lane == 63) s_part[wave] = incl;
    __syncthreads();
    const int p0 = s_part[0], p1 = s_part[1], p2 = s_part[2], p3 = s_part[3];
    const int total = p0 + p1 + p2 + p3;
    const int c0 = (wave > 0 ? p0 : 0) + (wave > 1 ? p1 : 0) + (wave > 2 ? p2 : 0) + incl - nch;
    const unsigned rel = (unsigned)(src - tile_lo);  // < 2^30 (k_piece_rows checked)
    for (int k = 0; k < nch; ++k) {
        const int left = n - 64 * k;
        const unsigned cnt = left < 64 ? (unsigned)left : 64u;
        const unsigned soff = (neg ? rel - 64u * k - 63u : rel + 64u * k) * 4u;
        s_tab[c0 + k] = (unsigned long long)soff | ((unsigned long long)(unsigned)(off + 64 * k) << 32) |
                        ((unsigned long long)(64u - cnt) << 45) | ((unsigned long long)neg << 51);
    }
    __syncthreads();
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) int *)s_counts;
    const unsigned long long base_u = (unsigned long long)(cov + tile_lo);  // workgroup-uniform: pin it to scalar registers
    const int32_t *base = reinterpret_cast<const int32_t *>(
        ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(base_u >> 32)) << 32) |
        (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)base_u));
    for (int r0 = 0; r0 < total; r0 += 4 * 64) {  // workgroup-uniform
        const int c = r0 + 4 * lane + wave;
        const unsigned long long e = s_tab[c < total ? c : 0];
        const unsigned w0 = (unsigned)e;
        const unsigned hi = (unsigned)(e >> 32);
        const unsigned w1 = (lds0 + (hi & 0x1fffu) * 4u) | (((hi >> 13) & 0x3fu) << 16) | (((hi >> 19) & 1u) << 24);
        int steps = (total - r0 - wave + 3) >> 2;  // chunks r0 + wave, r0 + wave + 4, ... < total
        steps = __builtin_amdgcn_readfirstlane(steps > 64 ? 64 : steps);
        if (steps > 0) issue_chunks(base, w0, w1, steps, lane);
    }
}

// An ORF's profile read through its pieces (float64 re-walks, tie replays: ~0.4 % of the ORFs).
struct PieceView {
    const int32_t *cov;
    const unsigned long long *start;
    const long long *base;
    long long j0, j1, beg;
    __device__ __forceinline__ int operator[](long long a) const
    {
        const long long p = beg + a;
        long long j = j0;
        while (j + 1 < j1 && (long long)(start[j + 1] & ~kPieceNeg) <= p) ++j;
        const long long idx = (start[j] & kPieceNeg) ? base[j] - p : base[j] + p;
        return cov[idx];
    }
};

// What the finish kernels read an ORF's counts from: the CSR array, or the coverage through the plan.
struct CsrSource {
    const int32_t *counts;
    __device__ __forceinline__ const int32_t *orf(long long i, long long beg) const { return counts + beg; }
};

struct CoverageSource {
    const int32_t *cov;
    PiecePlan pp;
    __device__ __forceinline__ PieceView orf(long long i, long long beg) const
    {
        return PieceView{cov, pp.start, pp.base, pp.orf_piece[i], pp.orf_piece[i + 1], beg};
    }
};

// ---------------------------------------------------------------------------------------
// The CSR `counts` array of a whole index (report_all mode, detect_orfs.py:301-324 prints
// every profile): a workgroup stages its tile exactly as the fused scorer does and writes
// it out, 16 bytes per lane.  8 bytes of traffic per nucleotide, the coverage read once.
// ---------------------------------------------------------------------------------------
constexpr int kGatherTileBlock = 256;

template <int TILE, int HALO>
__global__ __launch_bounds__(kGatherTileBlock) void k_tile_gather(const int32_t *__restrict__ cov, PiecePlan pp,
                                                                  long long total_nt, int32_t *__restrict__ counts)
{
    static_assert(kGatherTileBlock == kRowPieces, "one thread per row slot");
    __shared__ __attribute__((aligned(16))) int s_counts[TILE + HALO];  // the rows are clipped for the scorer: halo included
    __shared__ unsigned long long s_tab[kMaxChunks];
    __shared__ int s_part[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long b = blockIdx.x;
    const long long tile_lo = pp.tile_lo[b];
    if (tile_lo != kTileSlow) {
        stage_tile_chunks(cov, tile_lo, pp.rows[b * kRowPieces + tid], s_tab, s_part, s_counts, tid);
    } else {
        const uint4 lo = *piece_row_ptr(pp, b, lane, 0), hi = *piece_row_ptr(pp, b, lane, 1);
        stage_tile<TILE, HALO>(cov, pp, b, total_nt, lo, hi, s_counts, lane, wave);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const long long t0 = b * (long long)TILE;
    const long long left = total_nt - t0;
    if (left >= TILE) {
#pragma unroll
        for (int c = tid; c < TILE / 4; c += kGatherTileBlock)
            reinterpret_cast<int4 *>(counts + t0)[c] = reinterpret_cast<const int4 *>(s_counts)[c];
    } else {
        for (int c = tid; c < (int)left; c += kGatherTileBlock) counts[t0 + c] = s_counts[c];
    }
}

}  // namespace rp
