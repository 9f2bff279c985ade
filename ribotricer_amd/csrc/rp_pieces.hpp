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
//   bit  61     more   (last slot of the row only) the tile has further pieces: tile_piece0 + kRowPieces ...
constexpr int kPieceOffAt = 34, kPieceNAt = 47, kPieceNegAt = 60, kPieceMoreAt = 61;

struct PiecePlan {
    const unsigned long long *start;  // [n_pieces + 1] first profile position | kPieceNeg; sentinel total_nt
    const long long *base;            // [n_pieces + 1] coverage index = base + p, or base - p
    const long long *orf_piece;       // [n_orfs + 1]   pieces of ORF i: orf_piece[i] .. orf_piece[i+1]
    const long long *tile_piece0;     // [n_tiles]      piece holding the tile's first position
    const piece_desc_t *rows;         // [n_tiles][kRowPieces]
    long long n_pieces;
    long long coverage_len;
};

inline size_t piece_plan_bytes(long long n_orfs, long long n_pieces, long long n_tiles)
{
    auto up = [](size_t b) { return (b + 127) & ~(size_t)127; };
    return up((size_t)(n_pieces + 1) * 8) * 2 + up((size_t)(n_orfs + 1) * 8) + up((size_t)n_tiles * 8) +
           (size_t)n_tiles * kRowPieces * sizeof(piece_desc_t);
}

struct PiecePlanMem {
    unsigned long long *start;
    long long *base;
    long long *orf_piece;
    long long *tile_piece0;
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
    m.tile_piece0 = reinterpret_cast<long long *>(p);
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
// position, clip it and the next kRowPieces - 1 to [t0, t0 + TILE + HALO).
template <int TILE, int HALO>
__global__ __launch_bounds__(kRowPieces) void k_piece_rows(PiecePlanMem plan, long long n_pieces, long long total_nt)
{
    __shared__ long long s_j0;
    const long long b = blockIdx.x;
    const long long t0 = b * (long long)TILE;
    long long t_end = t0 + TILE + HALO;
    if (t_end > total_nt) t_end = total_nt;
    if (threadIdx.x == 0) {  // last j with start_j <= t0 (start_0 == 0 <= t0)
        long long lo = 0, hi = n_pieces;  // start[hi] (sentinel or later piece) > t0 unless t0 >= total_nt
        while (hi - lo > 1) {
            const long long mid = lo + (hi - lo) / 2;
            if ((long long)(plan.start[mid] & ~kPieceNeg) <= t0) lo = mid; else hi = mid;
        }
        s_j0 = lo;
        plan.tile_piece0[b] = lo;
    }
    __syncthreads();
    const long long j = s_j0 + threadIdx.x;
    piece_desc_t d = 0;
    if (j < n_pieces) d = clip_piece(plan.start[j], plan.start[j + 1], plan.base[j], t0, t_end);
    if (threadIdx.x == kRowPieces - 1 && j + 1 < n_pieces && (long long)(plan.start[j + 1] & ~kPieceNeg) < t_end)
        d |= (piece_desc_t)1 << kPieceMoreAt;
    plan.rows[b * kRowPieces + threadIdx.x] = d;
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
    if (__builtin_amdgcn_readlane((int)(hi.x | hi.y), 0) == 0) return;  // slots fill in order: <= 128 pieces
    stage_half(cov, hi, s_counts, lane, wave);
    const unsigned more = (unsigned)__builtin_amdgcn_readlane((int)(hi.w >> (kPieceMoreAt - 32)) & 1, 63);
    if (more) {  // rare: more than kRowPieces pieces in the tile (runs of very short exons / ORFs)
        const long long t0 = b * (long long)TILE;
        long long t_end = t0 + TILE + HALO;
        if (t_end > total_nt) t_end = total_nt;
        long long j = pp.tile_piece0[b] + kRowPieces;
        for (;;) {  // wave-uniform
            piece_desc_t d = 0;
            if (j + lane < pp.n_pieces) d = clip_piece(pp.start[j + lane], pp.start[j + lane + 1], pp.base[j + lane], t0, t_end);
            stage_round(cov, d, s_counts, lane, wave);
            j += 64;
            if (__builtin_amdgcn_readlane((int)(d != 0), 63) == 0) break;  // the list ran past the tile
        }
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
    __shared__ __attribute__((aligned(16))) int s_counts[TILE + HALO];  // the rows are clipped for the scorer: halo included
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long b = blockIdx.x;
    const uint4 lo = *piece_row_ptr(pp, b, lane, 0), hi = *piece_row_ptr(pp, b, lane, 1);
    stage_tile<TILE, HALO>(cov, pp, b, total_nt, lo, hi, s_counts, lane, wave);
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
