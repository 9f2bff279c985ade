// rp_tile.hpp -- flat-tile phase scoring: the throughput path (RP_ALGO_TILE).
//
// The CSR counts array is cut into fixed tiles of kTile positions on the 16-byte
// aligned address grid, one workgroup per tile, so every workgroup streams the same
// number of bytes regardless of how ragged the ORFs are (25 short ORFs or a slice of
// one 100 k-nt ORF cost the same).  Inside a tile:
//
//   1. the last wave stages the tile (+ a halo) in LDS with the LDS-DMA form of global_load
//      (each count leaves HBM once); meanwhile every wave reads its part of the tile's head row
//      of the plan (segment descriptors, lane map) and works out, in registers, which triplets
//      its lanes walk: slot 0 is the ORF that straddles in from the left ("head"), the others
//      the ORFs that start inside; each segment gets ceil(T/kRun) lanes, T = codon triplets
//      whose first position lies in the tile;
//   2. every lane walks a contiguous run of <= kRun triplets of ONE segment out of LDS
//      (odd dword stride between lanes -> bank-conflict free), one codon of each reading
//      frame per step, predicate-free fp32 arithmetic (one v_rsq_f32 per codon);
//   3. integer sums go to per-segment LDS atomics (exact, order independent); the six
//      float sums are folded by a segmented DPP scan inside each 16-lane row into one
//      record per (segment, row) -- deterministic, no float atomics;
//   4. wave f < 3 sums reading frame f of every segment's row records in float64 and writes
//      plane f of ONE 48-byte segment record (P_f, Q_f, N_f | M_f, a third of read count /
//      minimum codon coverage) to the workspace at index  orf + tile  (unique and increasing);
//   5. k_orf_finish, one thread per ORF, adds the records of the tiles the ORF spans (one, for
//      most), scores the frames, runs the state machine and the filters and stores the
//      outputs; ORFs whose fp32 frame decision is too close to call are re-walked in float64
//      from global memory by a whole wave (wave_walk, rp_wave.hpp).
//
// The scoring kernel does NOT finish ORFs itself: the float64 chain at the end of every workgroup costs it three times
// what the 48 bytes of write + read per segment cost (measured twice: DESIGN_HISTORY.md, "Notes moved out of rp_tile.hpp").
//
// Ownership rule: a triplet (3 positions from an ORF-relative multiple of 3) belongs
// to the tile that holds its FIRST position; its frame-1/2 codons may reach 4
// positions past the tile end, hence the halo.  A count is therefore consumed by
// exactly one lane (plus halo re-reads out of LDS, never out of HBM).
#pragma once

#include "rp_device.hpp"
#include "rp_wave.hpp"

namespace rp {

constexpr int kTileBlock = 256;
#ifndef RP_ROW_PREFETCH
#define RP_ROW_PREFETCH 0  // fused kernel: prefetch the plan rows of tile b + this many into L2 (a multiple of 8).  A/B knob, OFF: measured
                          // 2-3 % SLOWER at 512 / 1024 / 2048 on both layouts (profiles/archive/r04_ab_fused_row_prefetch.txt)
#endif
#ifndef RP_LOADERS
#define RP_LOADERS 1
#endif
constexpr int kLoaders = RP_LOADERS;  // waves (the last ones of the workgroup) that issue the tile DMA
#ifndef RP_TILE
#define RP_TILE 7936
#endif
constexpr int kTile = RP_TILE;  // positions per tile (31 KiB of int32): 3 lane-run passes of 64 x 15 triplets
#ifndef RP_KRUN
#define RP_KRUN 15
#endif
constexpr int kRun = RP_KRUN;   // triplets per lane run; odd => lane stride 45 dwords, conflict free
constexpr int kSegChunk = 64;   // segments set up per round (one per lane of wave 0)
constexpr int kHalo = 8;        // dwords staged past the tile end (4 needed, 2 chunks loaded)
// Indexes of short ORFs (mean length below kSmallTileMeanNt) are cut into smaller tiles: a tile of
// 60-nt ORFs then holds 102 segments = 2 rounds of the short-ORF path instead of 132 = 3.
#ifndef RP_TILE_SMALL
#define RP_TILE_SMALL 6144
#endif
constexpr int kTileSmall = RP_TILE_SMALL;
#ifndef RP_SMALL_MEAN
#define RP_SMALL_MEAN 180
#endif
constexpr long long kSmallTileMeanNt = RP_SMALL_MEAN;
inline int pick_tile(long long n_orfs, long long total_nt)
{
    return (n_orfs > 0 && total_nt < kSmallTileMeanNt * n_orfs) ? kTileSmall : kTile;
}
template <int TILE>
constexpr int lds_counts() { return TILE + kHalo + 3 * kRun + 16; }  // runs may read (masked) past the halo
static_assert(kTile / 3 < 65536, "N_f / M_f of a segment are kept in 16-bit fields");
constexpr int kMaxRecs = kSegChunk + (kTile / (3 * kRun) + kSegChunk + 2 * 64) / 16 + 1;  // one per (segment, 16-lane row)

struct TilePlan {
    long long n_tiles;
    long long total_nt;
    long long n_orfs;
    int mis;  // (counts address / 4) % 4: tiles live on the 16-byte aligned grid
};

// One record per (ORF, tile) segment: the sums over the triplets of the ORF that the tile
// owns.  Indexed by  orf + tile : an ORF spanning tiles s..e owns the ids orf+s .. orf+e, and
// the next ORF starts in a tile >= e, so ids never collide.  48 bytes as three 16-byte words
// in three planes (rec[f * n_rec + id]), one per reading frame: writers (wave f of the scoring
// kernel, a thread per segment) and readers (a thread per ORF) touch consecutive words with
// consecutive threads.
//   plane f  p[f]  q[f]  n_f | m_f << 16  extra_f     p, q fp32: the float64 sum of <= 11 fp32 row
//            records, rounded once; the read count is (extra_2 << 16) + extra_0 (sums of the row
//            records' high and low halves), extra_1 = min_codon (a tile owns < 2^16 triplets)
constexpr size_t kRecordBytes = 48;  // (other formats -- 32-byte records, one contiguous run per tile -- were built to parity and lost: DESIGN.md section 4)
__device__ __forceinline__ long long rec_index(long long n_rec, long long id, int f)
{
    return f * n_rec + id;
}

// position -> tile for x >= 0 without a 64-bit division (TILE = 2^k * m, m odd and small): the
// shifted value fits 32 bits for every set that fits this GPU's memory (x < 2^(32 + k) >= 2^40
// positions), and a 32-bit division by a constant is a multiply-high.  k_orf_finish is bound by
// its vector ALUs (70 % busy), and the two 64-bit divisions were ~50 of its ~350 main-path instructions.
template <int TILE>
__device__ __forceinline__ long long tile_of(long long x)
{
    constexpr int k = __builtin_ctz((unsigned)TILE);
    constexpr unsigned m = (unsigned)TILE >> k;
    return (long long)((unsigned)((unsigned long long)x >> k) / m);
}

// One descriptor per segment id, derived from the offsets alone (k_tile_desc): where the
// segment's triplets lie inside its tile's LDS image and how many lanes walk them.
//   bits  0-12  qfirst   LDS index of the first owned triplet
//   bits 13-25  endq     ORF end in LDS coordinates, clamped to kTile + kHalo
//   bits 26-37  ntrip    owned triplets
//   bits 38-50  tail     LDS index of an owned partial last codon (L % 3 != 0) ...
//   bits 51-52  part     ... and its length (0 = none)
//   bits 53-60  lanes    ceil(ntrip / kRun)
//   bit  63     live     0 = this id is a gap (no segment: empty ORF, or an unused id)
typedef unsigned long long seg_desc_t;
static_assert(kTile + kHalo < 8192 && kTile / 3 < 4096 && (kTile / 3 + kRun - 1) / kRun < 256, "descriptor field widths");
static_assert(kTileSmall <= kTile && kTileSmall % 256 == 0, "the small tile reuses the big tile's field widths and table sizes");

// The first kHeadSlots segment slots of every tile are ALSO kept tile-major, behind the tile's
// own [a0, a1) ORF range, and followed by the tile's lane map:
//   row b = { a0, a1, desc(slot 0) ... desc(slot kHeadSlots-1), vlmap[256 bytes] }
// vlmap[v] = the slot whose segment virtual lane v walks (0xff: none).  A workgroup reads its
// row with loads that depend on nothing but blockIdx, so every wave knows which triplets its
// lanes walk before the tile data has landed; only tiles with more segments than kHeadSlots
// (short-ORF batches) take the dependent path through the per-segment array and a table in LDS.
#ifndef RP_HEAD_SLOTS
#define RP_HEAD_SLOTS 64
#endif
constexpr int kHeadSlots = RP_HEAD_SLOTS;
static_assert(kHeadSlots <= kWave, "one head-row descriptor per lane");
constexpr int kHeadMapAt = 2 + kHeadSlots;    // 8-byte index where the lane map starts
constexpr int kHeadRow = kHeadMapAt + 256 / 8;  // 8-byte entries per row (640 B per 31 KiB tile)
static_assert(kTile / (3 * kRun) + kHeadSlots + 1 <= 256, "a head-row tile must fit one pass per wave");

// Too-close-to-call ORFs longer than this are not re-walked by the one wave that finds them
// (k_orf_finish) but queued for k_rewalk_long, where a 1024-thread workgroup takes each: a
// single wave streams a 100 k-nt profile for a millisecond.
constexpr long long kLongWalk = 4096;
constexpr int kLongBlock = 1024;
constexpr int kLongCountCap = 4096;  // k_tile_desc counts the index's ORFs beyond kLongWalk up to (about) here

struct TileWorkspace {
    long long *tile_first;  // [n_tiles + 1] first ORF starting at/after each tile start
    seg_desc_t *head;       // [max_tiles][kHeadRow]
    seg_desc_t *desc;       // [n_rec]
    uint4 *rec;             // [3 * n_rec]
    long long n_rec;
    int *long_count;        // number of queued long ORFs (zeroed by k_tile_score)
    long long *long_list;   // [long_capacity]: at most one entry per ORF longer than kLongWalk
};

inline long long long_capacity(long long total_nt) { return total_nt / kLongWalk + 1; }

inline size_t long_bytes(long long total_nt) { return 128 + (size_t)long_capacity(total_nt) * sizeof(long long); }

inline long long max_tiles(long long total_nt, int tile) { return (total_nt + 3 + tile - 1) / tile + 1; }

inline long long max_records(long long n_orfs, long long total_nt, int tile)
{
    return (n_orfs + max_tiles(total_nt, tile) + 1 + 15) & ~15LL;  // keeps every array 16-byte aligned
}

inline TilePlan make_tile_plan(long long n_orfs, long long total_nt, int mis, int tile)
{
    TilePlan p;
    p.n_orfs = n_orfs;
    p.total_nt = total_nt;
    p.mis = mis;
    p.n_tiles = (total_nt + p.mis + tile - 1) / tile;
    if (p.n_tiles < 1) p.n_tiles = 1;
    return p;
}

inline int counts_phase(const void *counts) { return (int)((reinterpret_cast<uintptr_t>(counts) >> 2) & 3u); }

inline size_t tile_index_bytes(long long total_nt, int tile)
{
    const size_t b = ((size_t)max_tiles(total_nt, tile) + 1) * sizeof(long long);
    return (b + 127) & ~(size_t)127;
}

inline size_t head_bytes(long long total_nt, int tile)
{
    return (size_t)max_tiles(total_nt, tile) * kHeadRow * sizeof(seg_desc_t);
}

// plan part (depends on the offsets only): tile index + segment descriptors
inline size_t plan_bytes(long long n_orfs, long long total_nt, int tile)
{
    return tile_index_bytes(total_nt, tile) + head_bytes(total_nt, tile) +
           (size_t)max_records(n_orfs, total_nt, tile) * sizeof(seg_desc_t);
}

// per-call part: the segment records
inline size_t record_bytes(long long n_orfs, long long total_nt, int tile)
{
    return (size_t)max_records(n_orfs, total_nt, tile) * kRecordBytes + long_bytes(total_nt);
}

inline size_t workspace_bytes(long long n_orfs, long long total_nt, int tile)
{
    return record_bytes(n_orfs, total_nt, tile) + plan_bytes(n_orfs, total_nt, tile);
}

// records first (always in the workspace), then -- unless the caller brings a plan -- the plan part
inline TileWorkspace carve_workspace(void *base, void *plan_base, long long n_orfs, long long total_nt, int tile)
{
    TileWorkspace ws;
    char *p = reinterpret_cast<char *>(base);
    ws.n_rec = max_records(n_orfs, total_nt, tile);
    ws.rec = reinterpret_cast<uint4 *>(p);
    p += (size_t)ws.n_rec * kRecordBytes;
    ws.long_count = reinterpret_cast<int *>(p);
    ws.long_list = reinterpret_cast<long long *>(p + 128);
    p += long_bytes(total_nt);
    char *q = plan_base ? reinterpret_cast<char *>(plan_base) : p;
    ws.tile_first = reinterpret_cast<long long *>(q);
    q += tile_index_bytes(total_nt, tile);
    ws.head = reinterpret_cast<seg_desc_t *>(q);
    q += head_bytes(total_nt, tile);
    ws.desc = reinterpret_cast<seg_desc_t *>(q);
    return ws;
}

// ---------------------------------------------------------------------------
// pass 1: tile_first[b] = lower_bound(offsets[0..n], start position of tile b)
// ---------------------------------------------------------------------------
// `err` (may be null) receives bit 0 when the offsets are not a valid CSR index for total_nt
// nucleotides (offsets[0] != 0, a decreasing step, offsets[n] != total_nt): the plan builder
// (rp_plan_create_dev) checks it once, so the per-sample scoring calls need not.
template <int TILE>
__global__ void k_tile_index(const int64_t *__restrict__ offsets, long long n_orfs, TilePlan plan,
                             long long *__restrict__ tile_first, int *__restrict__ err)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n_orfs) return;
    if (i == 0) tile_first[0] = 0;
    if (i == n_orfs) tile_first[plan.n_tiles] = n_orfs;
    const long long o = offsets[i];
    const long long o_prev = i > 0 ? (long long)offsets[i - 1] : -1 - (long long)plan.mis;
    if (err != nullptr) {
        const bool bad = (i == 0 && o != 0) || (i > 0 && o < o_prev) || (i == n_orfs && o != plan.total_nt);
        if (bad) atomicOr(err, 1);
    }
    // tiles b >= 1 whose start position b*TILE - mis lies in (o_prev, o]
    long long b_lo = (o_prev + plan.mis) / TILE + 1;
    long long b_hi = (o + plan.mis) / TILE;
    if (b_lo < 1) b_lo = 1;
    if (b_hi > plan.n_tiles - 1) b_hi = plan.n_tiles - 1;
    for (long long b = b_lo; b <= b_hi; ++b) tile_first[b] = i;
}

// Segment descriptors: one thread per ORF walks the tiles it spans.  Everything here depends
// on the offsets only, so with a plan it runs once per index, not once per sample.
template <int TILE>
__global__ void k_tile_desc(const int64_t *__restrict__ offsets, long long n_orfs, TilePlan plan,
                            const long long *__restrict__ tile_first, seg_desc_t *__restrict__ head,
                            seg_desc_t *__restrict__ desc, int *__restrict__ n_long)
{
    const long long orf = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (orf >= n_orfs) return;
    const long long beg = offsets[orf];
    const long long end = offsets[orf + 1];
    if (end <= beg) return;  // empty profile: its slot stays a gap, k_orf_finish reads no record for it
    if (beg < 0 || end > plan.total_nt) return;  // not a CSR index (rp_plan_create_dev reports it): stay in bounds
    // how many ORFs could ever reach k_rewalk_long (capped: all that the launcher wants to know is 0, a few, or >= its
    // grid) -- a plan's scoring calls size that launch by it and skip it for an index without such ORFs
    if (n_long != nullptr && end - beg > kLongWalk && *n_long < kLongCountCap) atomicAdd(n_long, 1);
    const long long b_first = (beg + plan.mis) / TILE;
    const long long b_last = (end - 1 + plan.mis) / TILE;
    for (long long b = b_first; b <= b_last; ++b) {
        const long long t0 = b * (long long)TILE - plan.mis;
        long long t1 = t0 + TILE;
        if (t1 > plan.total_nt) t1 = plan.total_nt;
        const int kt = (int)(t1 - t0);  // tile length in positions (<= TILE)
        int qfirst;                     // LDS index of the first triplet start >= tile start
        if (b > b_first) {              // the ORF straddles in from the left
            const unsigned m3 = (unsigned)((unsigned long long)(t0 - beg) % 3ull);
            qfirst = m3 == 0 ? 0 : 3 - (int)m3;
        } else {
            qfirst = (int)(beg - t0);
        }
        const long long rem = end - t0;  // > 0
        const int endq = rem <= TILE + kHalo ? (int)rem : TILE + kHalo;
        const int lim_q = kt < endq ? kt : endq;  // owned triplets start below this
        const int ntrip = lim_q > qfirst ? (lim_q - qfirst + 2) / 3 : 0;
        // partial last codon (L % 3 != 0): common.py:164-180 still sums it; it belongs to the
        // tile that holds its first position and is added by the record stage
        int tail = 0, part = 0;
        if (rem <= TILE + kHalo && endq > qfirst) {
            const int pt = (endq - qfirst) % 3;
            if (pt != 0 && endq - pt < kt) {
                tail = endq - pt;
                part = pt;
            }
        }
        const int lanes = (ntrip + kRun - 1) / kRun;
        const seg_desc_t d = (seg_desc_t)qfirst | ((seg_desc_t)endq << 13) | ((seg_desc_t)ntrip << 26) |
                             ((seg_desc_t)tail << 38) | ((seg_desc_t)part << 51) | ((seg_desc_t)lanes << 53) |
                             (1ull << 63);
        desc[orf + b] = d;
        const long long slot = orf - (tile_first[b] - 1);  // slot 0 = the ORF straddling in from the left
        if (slot >= 0 && slot < kHeadSlots) head[b * kHeadRow + 2 + slot] = d;
    }
}

// The [a0, a1) header and the lane map of every head row (after k_tile_desc): one wave per tile.
// Lane = slot reads its descriptor, a wave-wide scan gives every segment's first virtual lane, the
// segment marks it in a 256-entry LDS strip (entry v = virtual lane v), and a max-scan over the
// strip -- four entries per lane, then across lanes -- names the segment of every virtual lane.
// (The first version walked the 46 slots once per thread with dependent loads: 0.93 ms for the
// 500 000 tiles of an 11 M-ORF index, most of a plan build; this one is bandwidth-bound.)
constexpr int kHeadBlock = 256;

__device__ __forceinline__ int wave_max_scan_head(int x)  // inclusive max-scan over the wave, values >= 0
{
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true));
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true));
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true));
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true));
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false));
    return x;
}

__global__ __launch_bounds__(kHeadBlock) void k_tile_head(long long n_tiles, const long long *__restrict__ tile_first,
                                                          seg_desc_t *__restrict__ head)
{
    __shared__ __attribute__((aligned(16))) int s_mark[kHeadBlock / 64][256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long b = (long long)blockIdx.x * (kHeadBlock / 64) + w;
    if (b >= n_tiles) return;  // (wave-uniform; no workgroup barrier below)
    seg_desc_t *row = head + b * kHeadRow;
    const seg_desc_t d = lane < kHeadSlots ? row[2 + lane] : 0;
    if (lane == 0) {
        row[0] = (seg_desc_t)tile_first[b];
        row[1] = (seg_desc_t)tile_first[b + 1];
    }
    const int lanes_i = (int)(d >> 53) & 0xff;
    int incl = lanes_i;
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, true);
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, true);
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xf, 0xf, true);
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xf, 0xf, true);
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xa, 0xf, false);
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xc, 0xf, false);
    const int vs = incl - lanes_i;
    const int total = __builtin_amdgcn_readlane(incl, 63);  // <= 256 (static_assert at kHeadSlots)
    int *mk = s_mark[w];
    *reinterpret_cast<int4 *>(mk + 4 * lane) = make_int4(0, 0, 0, 0);
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (one wave: its LDS operations complete in order)
    if (lanes_i > 0 && vs < 256) mk[vs] = lane + 1;
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    int4 m = *reinterpret_cast<const int4 *>(mk + 4 * lane);
    m.y = max(m.x, m.y);
    m.z = max(m.y, m.z);
    m.w = max(m.z, m.w);
    const int scan = wave_max_scan_head(m.w);
    int before = __builtin_amdgcn_ds_bpermute(((lane + 63) & 63) << 2, scan);  // the previous lane's inclusive value
    if (lane == 0) before = 0;
    const int v0 = 4 * lane;
    const unsigned b0 = v0 + 0 < total ? (unsigned)(max(m.x, before) - 1) & 0xffu : 0xffu;
    const unsigned b1 = v0 + 1 < total ? (unsigned)(max(m.y, before) - 1) & 0xffu : 0xffu;
    const unsigned b2 = v0 + 2 < total ? (unsigned)(max(m.z, before) - 1) & 0xffu : 0xffu;
    const unsigned b3 = v0 + 3 < total ? (unsigned)(max(m.w, before) - 1) & 0xffu : 0xffu;
    reinterpret_cast<unsigned *>(row + kHeadMapAt)[lane] = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
}

// ---------------------------------------------------------------------------
// pass 2: the scoring kernel
// ---------------------------------------------------------------------------
// What one 16-lane row contributes to one segment (48 bytes): the six float sums, and -- since
// round 3 -- the integer results as well, carried through the same segmented DPP scan instead of
// per-segment LDS atomics (8-16 lanes on one address: a kernel without them ran 6.7 % faster,
// profiles/archive/r03_ab_no_int_atomics_upper_bound.txt):
//   S[f]      the census sum of reading frame f, 2^13 (E + 256 Z) + eps (see kDust*): E, Z <= 240 per row
//   clo, chi  the read count as two floats (low 16 bits, high bits): row sums < 2^24, exact
//   mn        the minimum codon coverage
struct alignas(16) RunRec {
    float p[3];
    float q[3];
    float S[3];
    float clo, chi;
    unsigned mn;
};

}  // namespace rp
#include "rp_pieces.hpp"
namespace rp {

// Stage positions [t0, t0 + kTile + kHalo) into LDS.  (counts + t0) is 16-byte aligned.
// Interior tiles use the LDS-DMA form of global_load (no VGPR round trip, one instruction
// per KiB row, rows dealt round-robin to the four waves); the first / last tile take the
// guarded path with zero fill.  The DMA is NOT waited for here.
#ifndef RP_DMA_AUX
#define RP_DMA_AUX 2  // nt: every count is read exactly once (cfg2-sized batches -3.5 %, 16 GB batches unchanged)
#endif
template <int TILE>
__device__ __forceinline__ int load_tile_to_lds(const int32_t *__restrict__ counts, long long t0,
                                                long long total_nt, int *s_counts, int tid)
{
    constexpr int n_chunks = (TILE + kHalo) / 4;  // 16-byte chunks
    constexpr int kRowPos = 256;                    // positions per DMA row (64 lanes x 16 B)
    static_assert(TILE % kRowPos == 0, "tile must be a whole number of 1 KiB rows");
    static_assert(kHalo == 8, "halo is loaded as two extra chunks");
    const bool interior = (t0 >= 0) && (t0 + TILE + kHalo <= total_nt);  // workgroup-uniform
    if (interior) {
        typedef const __attribute__((address_space(1))) void *gptr_t;
        typedef __attribute__((address_space(3))) void *lptr_t;
        const int lane = tid & (kWave - 1);
        const int32_t *src = counts + t0 + 4 * lane;
        // rows dealt round-robin to the LAST kLoaders waves.  A wave with LDS-DMA in flight cannot
        // wait for anything else it has loaded short of vmcnt(0) (measured on gfx950: LDS-DMA and
        // ordinary loads retire out of order with respect to each other, so a counted vmcnt(N)
        // does not cover an older ordinary load), so the waves that walk the first passes issue
        // none and get their plan rows early.  The wave index is made scalar so that the row loop
        // is straight-line code (M0 and the global offset are SALU arithmetic).
        constexpr int kIssuers = kLoaders;
        constexpr int kRows = TILE / kRowPos;
        constexpr int kEven = kRows / kIssuers;  // rows every issuing wave takes
        const int w = __builtin_amdgcn_readfirstlane(tid >> 6) - (kTileBlock / kWave - kIssuers);
        if (w >= 0) {
#pragma unroll
            for (int k = 0; k < kEven; ++k) {
                const int row = w + kIssuers * k;  // < kRows for every issuing wave
                __builtin_amdgcn_global_load_lds((gptr_t)(src + row * kRowPos), (lptr_t)(s_counts + row * kRowPos), 16, 0, RP_DMA_AUX);
            }
#pragma unroll
            for (int row = kEven * kIssuers; row < kRows; ++row)  // the left-over rows, one per wave
                if (w == row - kEven * kIssuers)
                    __builtin_amdgcn_global_load_lds((gptr_t)(src + row * kRowPos), (lptr_t)(s_counts + row * kRowPos), 16, 0, RP_DMA_AUX);
            if (w == kIssuers - 1 && lane < kHalo / 4)  // halo: 2 chunks past the tile
                __builtin_amdgcn_global_load_lds((gptr_t)(src + TILE), (lptr_t)(s_counts + TILE), 16, 0, RP_DMA_AUX);
        }
        // NOTE: no wait here -- the caller waits (vmcnt only tracks the DMA) after it has
        // issued its own independent loads.  Returned: how many loads this wave put in flight.
        if (w < 0) return 0;
        return kEven + (w < kRows - kEven * kIssuers ? 1 : 0) + (w == kIssuers - 1 ? 1 : 0);
    } else {
#pragma unroll 1
        for (int c = tid; c < n_chunks; c += kTileBlock) {
            const long long pos = t0 + 4LL * c;
            int4 v = make_int4(0, 0, 0, 0);
            if (pos + 0 >= 0 && pos + 0 < total_nt) v.x = counts[pos + 0];
            if (pos + 1 >= 0 && pos + 1 < total_nt) v.y = counts[pos + 1];
            if (pos + 2 >= 0 && pos + 2 < total_nt) v.z = counts[pos + 2];
            if (pos + 3 >= 0 && pos + 3 < total_nt) v.w = counts[pos + 3];
            *reinterpret_cast<int4 *>(s_counts + 4 * c) = v;
        }
        return 0;  // (the caller then waits for everything)
    }
}

// --- DPP building blocks (gfx9 row_shr / row_bcast controls) ------------------------------
constexpr int kDppRowShr1 = 0x111, kDppRowShr2 = 0x112, kDppRowShr4 = 0x114, kDppRowShr8 = 0x118;
constexpr int kDppRowBcast15 = 0x142, kDppRowBcast31 = 0x143;

template <int CTRL>
__device__ __forceinline__ int dpp_row(int src)
{
    // row_shr within the 16-lane row; lanes without a source read 0 (bound_ctrl)
    return __builtin_amdgcn_update_dpp(0, src, CTRL, 0xf, 0xf, true);
}

struct LaneSums {
    float p[3];
    float q[3];
    float S[3];      // census sums (flat / all-zero codons), decoded per row in the record stage
    float clo, chi;  // read count: (cnt & 0xffff), (cnt >> 16) -- <= 65 535 and <= 11 520 per lane
    unsigned mn;
};

// One step of the segmented scan: x += m * x[lane - K] inside the 16-lane row for the six
// float sums (lanes without a source read 0), as six v_fmac_f32 with a DPP source.  hipcc
// keeps a separate v_mov_b32_dpp in front of every fma, hence the asm; it is ONE block so
// that the six stay together: a DPP read needs 2 wait states after the VALU write of its
// source, which the leading s_nop gives the first and the five others give each following
// step (the compiler inserts no wait states around inline asm).
#define RP_SCAN_FMAC(n, ctrl) "v_fmac_f32_dpp %" #n ", %" #n ", %11 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define RP_SCAN_STEP(ctrl)                                                                              \
    asm volatile("s_nop 1\n\t"                                                                          \
                 RP_SCAN_FMAC(0, ctrl) RP_SCAN_FMAC(1, ctrl) RP_SCAN_FMAC(2, ctrl) RP_SCAN_FMAC(3, ctrl)  \
                 RP_SCAN_FMAC(4, ctrl) RP_SCAN_FMAC(5, ctrl) RP_SCAN_FMAC(6, ctrl) RP_SCAN_FMAC(7, ctrl)  \
                 RP_SCAN_FMAC(8, ctrl) RP_SCAN_FMAC(9, ctrl)                                              \
                 "v_fmac_f32_dpp %10, %10, %11 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1"          \
                 : "+v"(v.p[0]), "+v"(v.p[1]), "+v"(v.p[2]), "+v"(v.q[0]), "+v"(v.q[1]), "+v"(v.q[2]),    \
                   "+v"(v.S[0]), "+v"(v.S[1]), "+v"(v.S[2]), "+v"(v.clo), "+v"(v.chi)                     \
                 : "v"(m))

template <int K>
__device__ __forceinline__ void seg_scan_step(LaneSums &v, int key)
{
    static_assert(K == 1 || K == 2 || K == 4 || K == 8, "row_shr distance");
    constexpr int ctrl = K == 1 ? kDppRowShr1 : K == 2 ? kDppRowShr2 : K == 4 ? kDppRowShr4 : kDppRowShr8;
    const bool same = dpp_row<ctrl>(key) == key;  // inside the segment (false across its start and for lanes without a source)
    const float m = same ? 1.0f : 0.0f;
    // the minimum: the source lane's value where it belongs to the same segment (lanes without a source keep their own)
    const unsigned from = (unsigned)__builtin_amdgcn_update_dpp((int)v.mn, (int)v.mn, ctrl, 0xf, 0xf, false);
    v.mn = same ? min(v.mn, from) : v.mn;
#ifdef RP_SCAN_PLAIN  // the same step in plain C++ (A/B reference)
#pragma unroll
    for (int f = 0; f < 3; ++f) {
        v.p[f] = __builtin_fmaf(__int_as_float(dpp_row<ctrl>(__float_as_int(v.p[f]))), m, v.p[f]);
        v.q[f] = __builtin_fmaf(__int_as_float(dpp_row<ctrl>(__float_as_int(v.q[f]))), m, v.q[f]);
        v.S[f] = __builtin_fmaf(__int_as_float(dpp_row<ctrl>(__float_as_int(v.S[f]))), m, v.S[f]);
    }
    v.clo = __builtin_fmaf(__int_as_float(dpp_row<ctrl>(__float_as_int(v.clo))), m, v.clo);
    v.chi = __builtin_fmaf(__int_as_float(dpp_row<ctrl>(__float_as_int(v.chi))), m, v.chi);
    return;
#endif
    if constexpr (K == 1) RP_SCAN_STEP("row_shr:1");
    if constexpr (K == 2) RP_SCAN_STEP("row_shr:2");
    if constexpr (K == 4) RP_SCAN_STEP("row_shr:4");
    if constexpr (K == 8) RP_SCAN_STEP("row_shr:8");
}

__device__ __forceinline__ void seg_scan_rows(LaneSums &v, int key)
{
    seg_scan_step<1>(v, key);
    seg_scan_step<2>(v, key);
    seg_scan_step<4>(v, key);
    seg_scan_step<8>(v, key);
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_fetch(int old, int src)
{
    // lanes without a valid source (or outside ROW_MASK) keep `old`
    return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, 0xf, false);
}

// inclusive max-scan over the wave (unsegmented), values >= 0
__device__ __forceinline__ int wave_max_scan(int x)
{
    x = max(x, dpp_row<kDppRowShr1>(x));
    x = max(x, dpp_row<kDppRowShr2>(x));
    x = max(x, dpp_row<kDppRowShr4>(x));
    x = max(x, dpp_row<kDppRowShr8>(x));
    x = max(x, dpp_fetch<kDppRowBcast15, 0xa>(0, x));
    x = max(x, dpp_fetch<kDppRowBcast31, 0xc>(0, x));
    return x;
}

// inclusive add-scan over the wave (unsegmented)
__device__ __forceinline__ int wave_add_scan(int x)
{
    x += dpp_row<kDppRowShr1>(x);
    x += dpp_row<kDppRowShr2>(x);
    x += dpp_row<kDppRowShr4>(x);
    x += dpp_row<kDppRowShr8>(x);
    x += dpp_fetch<kDppRowBcast15, 0xa>(0, x);
    x += dpp_fetch<kDppRowBcast31, 0xc>(0, x);
    return x;
}

#ifndef RP_KRUNBLOCK
#define RP_KRUNBLOCK 3
#endif
constexpr int kRunBlock = RP_KRUNBLOCK;  // triplets per fully unrolled block of the lane run
// Tiles of nothing but very short segments (> kHeadSlots of them, none longer than kLaneTrip triplets: an index of
// 60-66-nt ORFs puts 103 on a 6 144-position tile): ONE LANE walks a whole segment (lane_segments below).
#ifndef RP_LANE_SEG
#define RP_LANE_SEG 1
#endif
#ifndef RP_LANE_TRIP
#define RP_LANE_TRIP 22
#endif
constexpr int kLaneTrip = RP_LANE_TRIP;     // longest segment a single lane takes (22 triplets = 66 nt; even: kLaneRun is odd)
constexpr int kLaneRun = kLaneTrip + 1;     // its run length as lane_run sees it (odd)
static_assert(kLaneRun >= kRun && kLaneRun % 2 == 1, "lane_run is instantiated for kRun, 9, 5 and kLaneRun");

// min(x, hi) for x, hi >= 0 on the BIT PATTERNS: non-negative IEEE floats order like
// unsigned integers (+inf = 0x7f800000 included), so this is one 32-bit-encoded v_min_u32
// (2.8 cycles measured) instead of a VOP3 v_med3_f32 (4.4), it needs no NaN-canonicalising
// v_max(x, x) the way fminf() does, and it is a compiler-visible instruction (hipcc inserts
// no hazard wait states around inline asm, and v_rsq_f32 results need them).
__device__ __forceinline__ float min_nonneg(float x, float hi)
{
    const unsigned a = __float_as_uint(x), b = __float_as_uint(hi);
    return __uint_as_float(a < b ? a : b);
}

// clamp(x, 0, 1): folds into the producing instruction's clamp modifier
__device__ __forceinline__ float clamp01(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, 1.0f); }

// ---------------------------------------------------------------------------------------
// The lane run: <= kRun triplets = 3*kRun codon starts of ONE segment out of consecutive LDS
// dwords (lane stride 3*kRun dwords, odd -> bank-conflict free), one codon of each reading
// frame per position, fp32, no v_cmp / v_cndmask in the float part (statistics.py:67-91):
//   d0 = a - b, d1 = b - c   exact (counts < 2^24);   q = d0^2 + d0 d1 + d1^2 (+ dust, below)
//   r  = rsq(q);   P += d0 r,  Q += d1 r          the unit vector in the oblique basis
//   S += r                                        the census of flat codons (a == b == c)
//   frame 0 also sums the integer codon (read count, minimum codon coverage)
// A run is walked in blocks of kRunBlock triplets, "ragged block first".  Only ONE block of a
// lane can hold codon starts that are not the lane's to count -- the last one that holds any
// (the run is short, or it ends the ORF and the last triplet's frame-1/2 codons reach past
// it) -- so that block is processed FIRST, by every lane at once, with the roots of the
// starts past `lim` multiplied by 0.  The blocks below it are full by construction and are
// walked downwards with NO validity arithmetic at all (lanes with fewer blocks drop out
// through EXEC): 10 VALU instructions per position + the integer codon sum (12 in the masked
// block).  The two look-ahead values of a block are the first two of the block above it, kept
// in registers, so every count is read from LDS and converted exactly once.
// ---------------------------------------------------------------------------------------
struct RunAcc {
    float P[3], Q[3];
    float S[3];  // sum of the reciprocal roots: the census of flat codons (see kDust*)
    unsigned cnt, mn;
};

// q is kept off zero by a "dust" folded into d0^2 that depends on the codon's first count a:
// 2^-42 for a == 0, 2^-26 for a >= 1 (one v_med3 of the converted count).  Any q >= 1 absorbs
// either exactly, and a flat codon (a == b == c, so d0 = d1 = 0 and its products with the root
// vanish) gets a root that names its kind: 2^21 for an all-zero codon, 2^13 for a flat one with
// reads.  Every other root is <= 1, so ONE running sum per frame S = 2^13 (E + 256 Z) + eps
// carries both counts -- of a lane (E, Z <= kRun) and, summed by the segmented scan, of a 16-lane
// row (E, Z <= 240 < 256; S < 2^29, eps and its fp32 rounding stay far below half a unit):
// M = codons - Z - E, N = codons - Z -- no per-position indicator arithmetic and no integer
// reduction at all; the record stage decodes the row sums.
constexpr float kDustZero = 0x1p-42f, kDustFlat = 0x1p-26f, kFlatUnit = 0x1p-13f;
static_assert(16 * kRun < 256, "the flat-codon census packs E into 8 bits per 16-lane row and frame");

// One block of 3 * kRunBlock codon starts at s[0..]; (nf0, nf1) = the two values after the
// block as floats.  MASKED: only the first `lim` starts count (their roots are multiplied by
// 0 / 1).  On return (nf0, nf1) are the block's own first two values, for the block below.
template <bool MASKED>
__device__ __forceinline__ void run_block(const int *__restrict__ s, int lim, float &nf0, float &nf1, RunAcc &a)
{
    constexpr int B = 3 * kRunBlock;
    int v[B];
    float f[B + 2];
#pragma unroll
    for (int c = 0; c < B; ++c) v[c] = s[c];
#pragma unroll
    for (int c = 0; c < B; ++c) f[c] = (float)v[c];
    f[B] = nf0;
    f[B + 1] = nf1;
    float d[B + 1];
#pragma unroll
    for (int c = 0; c <= B; ++c) d[c] = f[c] - f[c + 1];
    float sq[B];
#pragma unroll
    for (int c = 0; c < B; ++c) sq[c] = __builtin_fmaf(d[c], d[c], __builtin_amdgcn_fmed3f(f[c], kDustZero, kDustFlat));
    const float limf = (float)lim;
#pragma unroll
    for (int c = 0; c < B; ++c) {
        const int fr = c % 3;
        const float qq = __builtin_fmaf(d[c + 1], f[c] - f[c + 2], sq[c]);  // d0^2 + d1 (d0 + d1)
        float r = __builtin_amdgcn_rsqf(qq);  // <= 1 for q >= 1; 2^13 / 2^18 for a flat codon
        if constexpr (MASKED) r *= clamp01(limf - (float)c);
        a.P[fr] = __builtin_fmaf(d[c], r, a.P[fr]);
        a.Q[fr] = __builtin_fmaf(d[c + 1], r, a.Q[fr]);
        a.S[fr] += r;
        if (fr == 0) {
            const unsigned codon = (unsigned)(v[c] + v[c + 1] + v[c + 2]);
            if constexpr (MASKED) {
                const bool valid = c < lim;
                a.cnt += valid ? codon : 0u;
                a.mn = min(a.mn, valid ? codon : (unsigned)RP_MIN_CODON_COV_EMPTY);
            } else {
                a.cnt += codon;
                a.mn = min(a.mn, codon);
            }
        }
    }
    nf0 = f[0];
    nf1 = f[1];
}

template <int KRUN>
__device__ __forceinline__ void lane_run(const int *__restrict__ s, int lim, LaneSums &o)
{
    static_assert(KRUN % 2 == 1 && KRUN <= kLaneRun, "odd run lengths keep the LDS lane stride conflict free");
    constexpr int B = 3 * kRunBlock;
    constexpr int kBlocks = (KRUN + kRunBlock - 1) / kRunBlock;
    lim = lim > 0 ? lim : 0;  // a run that starts in the last two positions of an ORF owns no codon start
    RunAcc a;
#pragma unroll
    for (int f = 0; f < 3; ++f) a.P[f] = a.Q[f] = a.S[f] = 0.f;
    a.cnt = 0;
    a.mn = (unsigned)RP_MIN_CODON_COV_EMPTY;
    // last block holding a valid codon start (block 0 for an idle lane: lim == 0 masks it all)
    const int lb = lim > 0 ? (lim - 1) / B : 0;
    // all block addresses as constant offsets from ONE per-lane base (the lowest block a lane
    // could reach, which may lie below its run -- never dereferenced there)
    const int *lo = s + (lb - (kBlocks - 1)) * B;
    float nf0 = (float)lo[kBlocks * B], nf1 = (float)lo[kBlocks * B + 1];
    run_block<true>(lo + (kBlocks - 1) * B, lim - lb * B, nf0, nf1, a);
#pragma unroll
    for (int it = 1; it < kBlocks; ++it) {
        if (lb >= it)  // lanes whose run has fewer blocks sit this one out (EXEC)
            run_block<false>(lo + (kBlocks - 1 - it) * B, B, nf0, nf1, a);
    }
#pragma unroll
    for (int f = 0; f < 3; ++f) {
        o.p[f] = a.P[f];
        o.q[f] = a.Q[f];
        o.S[f] = a.S[f];
    }
    o.clo = (float)(a.cnt & 0xffffu);  // <= 3 * kRun * RP_MAX_COUNT < 2^30: two exact floats
    o.chi = (float)(a.cnt >> 16);
    o.mn = a.mn;
}

constexpr int kMaxVl = kTile / (3 * kRun) + kSegChunk + 2 * kWave;  // virtual lanes per chunk (upper bound)

#ifndef RP_MIN_WAVES
#define RP_MIN_WAVES 4
#endif

// Phase stamps (-DRP_STAMPS, measurement builds only): per wave, the shader-clock time spent
// up to each point of the kernel, summed over all workgroups into rp_dbg_stamps[wave][k].
#ifdef RP_STAMPS
constexpr int kStampEvery = 64, kStampSlots = 4096;
__device__ unsigned long long rp_dbg_stamps[kStampSlots][4][8];  // plain stores by every 64th workgroup
#define RP_STAMP_DECL unsigned long long t_stamp_[8]; int n_stamp_ = 0;
#define RP_STAMP() do { t_stamp_[n_stamp_++] = __builtin_amdgcn_s_memtime(); } while (0)
#define RP_STAMP_FLUSH()                                                                        \
    do {                                                                                        \
        if (lane == 0 && blockIdx.x % kStampEvery == 0 && blockIdx.x / kStampEvery < kStampSlots) \
            for (int k_ = 0; k_ < n_stamp_; ++k_) rp_dbg_stamps[blockIdx.x / kStampEvery][wave][k_] = t_stamp_[k_]; \
    } while (0)
#else
#define RP_STAMP_DECL
#define RP_STAMP() do {} while (0)
#define RP_STAMP_FLUSH() do {} while (0)
#endif
// One pass of a wave: every lane walks its run, then the per-segment reduction -- a segmented
// inclusive scan inside each 16-lane DPP row over the six float sums, the three census sums, the
// two halves of the read count (all as v_fmac_f32_dpp) and the minimum: deterministic, no atomics;
// the last lane of a segment in a row stores one 48-byte row record.
template <int KRUN>
__device__ __forceinline__ void tile_pass(const int *__restrict__ s_counts, RunRec *__restrict__ s_rec, int q0, int lim,
                                          bool active, int seg, int vl)
{
    LaneSums sv;
    lane_run<KRUN>(s_counts + q0, lim, sv);
    const int key = active ? seg + 1 : kSegChunk + 1;
    seg_scan_rows(sv, key);
    const int key_next = dpp_fetch<0x101 /* row_shl:1 */, 0xf>(0, key);  // 0 at the row's last lane
    const bool run_end = active && (key_next != key);
    if (run_end) {
        RunRec &rec = s_rec[seg + (vl >> 4)];
#pragma unroll
        for (int f = 0; f < 3; ++f) {
            rec.p[f] = sv.p[f];
            rec.q[f] = sv.q[f];
            rec.S[f] = sv.S[f];
        }
        rec.clo = sv.clo;
        rec.chi = sv.chi;
        rec.mn = sv.mn;
    }
}

// Record stage: the row records of a segment -> ONE record.  The words of a record live in planes (rec[f * n_rec + id]);
// wave f < 3 sums reading frame f for all 64 slots (thread = slot), so the three short dependency chains run side by
// side on three SIMDs and every store instruction covers consecutive 16-byte words.  Per row the census sum of the frame
// is decoded -- k = round(S / 2^13) = E + 256 Z, the row's flat and all-zero codons -- and the frame's codon starts come
// from the segment's geometry (s_geom: how many of its start positions are valid), so N = codons - Z and M = N - E need
// no counting anywhere.  The segment leaves plane f = {P_f, Q_f, N_f | M_f << 16, extra_f} (48 bytes: k_orf_finish adds
// the sums of the tiles an ORF spans).
__device__ __forceinline__ void record_stage(const int *__restrict__ s_counts, const RunRec *__restrict__ s_rec,
                                             const int *__restrict__ s_vlstart, const int *__restrict__ s_tail,
                                             const int *__restrict__ s_live, const int *__restrict__ s_geom,
                                             uint4 *__restrict__ rec, long long n_rec, long long id0, int wave, int seg)
{
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    const bool live = wave < 3 && s_live[seg] != 0;  // the slot holds a segment
    double a0 = 0.0, a1 = 0.0;
    unsigned n = 0, m = 0;
    unsigned extra = wave == 1 ? (unsigned)RP_MIN_CODON_COV_EMPTY : 0u;  // wave 0: sum of clo, wave 1: minimum, wave 2: sum of chi
    if (live) {
        const int vs = s_vlstart[seg];
        const int ve = s_vlstart[seg + 1];
        const int w_first = vs >> 4;
        const int w_last = ve > vs ? (ve - 1) >> 4 : w_first - 1;
        const int tail = s_tail[seg];
        int flat = 0, zero = 0;
        for (int w = w_first; w <= w_last; ++w) {
            const RunRec &r = s_rec[seg + w];
            a0 += (double)(wave == 0 ? r.p[0] : wave == 1 ? r.p[1] : r.p[2]);
            a1 += (double)(wave == 0 ? r.q[0] : wave == 1 ? r.q[1] : r.q[2]);
            const int k = (int)__builtin_fmaf(wave == 0 ? r.S[0] : wave == 1 ? r.S[1] : r.S[2], kFlatUnit, 0.5f);  // E + 256 Z of this row
            flat += k & 255;
            zero += k >> 8;
            if (wave == 0)
                extra += (unsigned)r.clo;  // <= 11 rows x 2^20
            else if (wave == 1)
                extra = min(extra, r.mn);
            else
                extra += (unsigned)r.chi;  // <= 11 rows x 2^18
        }
        const int lim_seg = s_geom[seg];  // the segment's valid codon starts are its first lim_seg start positions
        const int codons = ((lim_seg + 2 - wave) * 21846) >> 16;  // those of frame `wave`: floor((lim + 2 - f) / 3), lim < 8 192
        n = (unsigned)(codons - zero);
        m = n - (unsigned)flat;
        if (tail >= 0 && wave < 2) {  // an owned partial last codon (L % 3 != 0): common.py:164-180 still sums it
            unsigned codon = (unsigned)s_counts[tail & 0xffff];
            if ((tail >> 16) == 2) codon += (unsigned)s_counts[(tail & 0xffff) + 1];
            extra = wave == 0 ? extra + codon : min(extra, codon);  // (<= 2 x 2^24 on top of the clo sum: fits)
        }
    }
    if (!live) return;
    // written once, read once by the next kernel: a streaming store (rp_device.hpp, stream_store)
    stream_store(reinterpret_cast<u32x4_t *>(rec + rec_index(n_rec, id0 + seg, wave)),
                 u32x4_t{__float_as_uint((float)a0), __float_as_uint((float)a1), n | (m << 16), extra});
}

// the segment's number of valid codon-start positions (every frame counted), from its descriptor:
// its owned triplets offer 3 * ntrip starts, of which those whose codon would reach past the ORF's
// end (rem = endq - qfirst positions from the first start on) are not valid
__device__ __forceinline__ int seg_geom(seg_desc_t d)
{
    const int rem = (((int)(d >> 13)) & 0x1fff) - ((int)d & 0x1fff);
    const int starts = 3 * ((int)(d >> 26) & 0xfff);
    const int lim = rem - 2 < starts ? rem - 2 : starts;
    return lim > 0 ? lim : 0;
}

// One round of the short-ORF path: the 64 segments whose descriptors the lanes hold (`dc`), walked
// with runs of KRUN triplets.  Every wave finds its lanes' segments through 64 private words of
// LDS (marks at the segments' first lanes, then a max-scan); tables, pass, reduction as in the
// common path.  Ends after barrier 2 (the caller runs the record stage).
template <int KRUN>
__device__ __forceinline__ void short_round(seg_desc_t dc, const int *s_counts, RunRec *s_rec, int *s_vlstart, int *s_tail,
                                            int *s_live, int *s_geom, int *s_owner, int wave, int lane)
{
    const int ntrip_i = (int)(dc >> 26) & 0xfff;
    const int lanes_i = (dc >> 63) ? (ntrip_i + KRUN - 1) / KRUN : 0;
    const int incl = wave_add_scan(lanes_i);
    const int vs_i = incl - lanes_i;
    const int total_vl = __builtin_amdgcn_readlane(incl, kWave - 1);
    const int vbase = wave * kWave;
    const int vl = vbase + lane;
    const bool pass = vbase < total_vl;  // wave-uniform
    const bool active = vl < total_vl;
    int q0 = 0, lim = 0, seg = 0;
    if (pass) {
        int *mk = s_owner + vbase;  // this wave's 64 words
        mk[lane] = 0;
        if (lanes_i > 0) {
            const int tgt = vs_i - vbase;
            if (tgt >= 0 && tgt < kWave)
                mk[tgt] = lane + 1;
            else if (tgt < 0 && vs_i + lanes_i > vbase)
                mk[0] = lane + 1;  // the one segment that straddles into this pass
        }
        asm volatile("" ::: "memory");  // LDS operations of one wave complete in order
        seg = wave_max_scan(mk[lane]) - 1;
        if (seg < 0) seg = 0;  // (only on a malformed index: keep the lane fetches in range)
        const unsigned dlo = (unsigned)__builtin_amdgcn_ds_bpermute(seg << 2, (int)(unsigned)dc);
        const unsigned dhi = (unsigned)__builtin_amdgcn_ds_bpermute(seg << 2, (int)(unsigned)(dc >> 32));
        const int vs_s = __builtin_amdgcn_ds_bpermute(seg << 2, vs_i);
        const seg_desc_t ds = ((seg_desc_t)dhi << 32) | dlo;
        const int r = vl - vs_s;
        int n_run = ((int)(ds >> 26) & 0xfff) - r * KRUN;
        n_run = n_run > KRUN ? KRUN : n_run;
        q0 = active ? ((int)ds & 0x1fff) + 3 * KRUN * r : 0;
        const int rem0 = ((int)(ds >> 13) & 0x1fff) - q0;
        lim = rem0 - 2 < 3 * n_run ? rem0 - 2 : 3 * n_run;
        if (!active) lim = 0;
    }
    if (wave == 0) {  // what the record stage needs, per slot
        const int part = (int)(dc >> 51) & 3;
        s_vlstart[lane] = vs_i;
        if (lane == kWave - 1) s_vlstart[kSegChunk] = incl;
        s_tail[lane] = part ? (((int)(dc >> 38) & 0x1fff) | (part << 16)) : -1;
        s_live[lane] = (int)(dc >> 63);
        s_geom[lane] = seg_geom(dc);
    }
    __syncthreads();  // the previous round's record stage is done with the row records
    if (pass) tile_pass<KRUN>(s_counts, s_rec, q0, lim, active, seg, vl);
    __syncthreads();
}

// Lane-per-segment rounds (round 4).  For a tile whose segments are all <= kLaneTrip triplets the 64-slot rounds above
// spend most of their instructions on bookkeeping: per round two barriers, the lane mapping through LDS marks, a
// segmented scan of twelve sums for runs of 5 triplets, row records, a record stage -- twice for the 103 segments of an
// all-60-nt tile.  Here thread t takes slot 256 r + t whole: one lane run of <= kLaneRun triplets straight to the
// segment's 48-byte record -- no scan, no row records, no barrier after the tile has landed.  A lane's census sums
// decode in the lane (E, Z <= 23), the frame's codon starts follow from the geometry as in record_stage.
// (LDS: lanes read at a stride of their segments' lengths; 60-nt ORFs put 64 lanes on 8 banks -- ~0.5 k cycles of
// conflicts per wave against ~2.6 k cycles of arithmetic: paid, and still half the instructions of the rounds.)
// Record words as record_stage writes them: plane f = {P_f, Q_f, N_f | M_f << 16, extra_f}, extra_0 / extra_2 = low /
// high half of the read count (+ an owned partial last codon in the low half), extra_1 = the minimum codon.
template <int TILE>
__device__ __forceinline__ void lane_segments(const int *__restrict__ s_counts, const seg_desc_t d, uint4 *__restrict__ rec,
                                              long long n_rec, long long id)
{
    if (!(d >> 63)) return;
    const int q0 = (int)d & 0x1fff;
    const int endq = (int)(d >> 13) & 0x1fff;
    const int ntrip = (int)(d >> 26) & 0xfff;
    const int part = (int)(d >> 51) & 3;
    const int rem0 = endq - q0;
    int lim = rem0 - 2 < 3 * ntrip ? rem0 - 2 : 3 * ntrip;
    lim = lim > 0 ? lim : 0;
    LaneSums sv;
    lane_run<kLaneRun>(s_counts + q0, lim, sv);
    unsigned lo = (unsigned)sv.clo, hi = (unsigned)sv.chi, mn = sv.mn;
    if (part) {  // an owned partial last codon (L % 3 != 0): common.py:164-180 still sums it
        const int tail = (int)(d >> 38) & 0x1fff;
        unsigned codon = (unsigned)s_counts[tail];
        if (part == 2) codon += (unsigned)s_counts[tail + 1];
        lo += codon;
        mn = min(mn, codon);
    }
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    unsigned n[3], m[3];
#pragma unroll
    for (int f = 0; f < 3; ++f) {
        const int k = (int)__builtin_fmaf(sv.S[f], kFlatUnit, 0.5f);  // E + 256 Z of this lane's run
        const int codons = ((lim + 2 - f) * 21846) >> 16;             // floor((lim + 2 - f) / 3), lim < 8 192
        n[f] = (unsigned)(codons - (k >> 8));
        m[f] = n[f] - (unsigned)(k & 255);
    }
#pragma unroll
    for (int f = 0; f < 3; ++f) {
        const unsigned extra = f == 0 ? lo : f == 1 ? mn : hi;
        stream_store(reinterpret_cast<u32x4_t *>(rec + rec_index(n_rec, id, f)),
                     u32x4_t{__float_as_uint(sv.p[f]), __float_as_uint(sv.q[f]), n[f] | (m[f] << 16), extra});
    }
}

// FUSED: `counts` is the dense coverage and the tile is staged through the piece plan
// (rp_pieces.hpp) -- the profiles never exist in HBM (plan.mis == 0 there).
// The work of one workgroup on one tile (k_tile_score below calls it once, or RP_TILES_PER_WG times).
template <bool FUSED, int TILE>
__device__ __forceinline__ void tile_body(const int32_t *__restrict__ counts, const TilePlan &plan, const TileWorkspace &ws,
                                          const PiecePlan &pp, const long long b, int *__restrict__ s_counts, int *__restrict__ s_live,
                                          int *__restrict__ s_tail, int *__restrict__ s_vlstart, int *__restrict__ s_owner,
                                          RunRec *__restrict__ s_rec, int *__restrict__ s_geom)
{
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = tid >> 6;
    const long long t0 = b * (long long)TILE - plan.mis;  // position of LDS index 0 (may be < 0 for b == 0)
    if (b == 0 && tid == 0) *ws.long_count = 0;  // (k_orf_finish, next in the stream, appends)
    unsigned prefetched = 0;  // (fused: destination of the row prefetch; kept live up to the wait that covers it)
    RP_STAMP_DECL
    RP_STAMP();  // 0: entry
    // The tile's head row first -- its [a0, a1) ORF range through the scalar unit, its segment
    // descriptors one per lane, in EVERY wave -- then the tile DMA behind it.  Slot L of the
    // tile holds ORF a0 - 1 + L: slot 0 is the ORF that straddles in from the left, if any.
    const long long a0 = (long long)ws.head[b * kHeadRow + 0];
    const long long a1 = (long long)ws.head[b * kHeadRow + 1];
    seg_desc_t d;
    unsigned vmap;
    {   // both depend on blockIdx only; lanes past the row's descriptors re-read its first word
        const seg_desc_t *row = ws.head + b * kHeadRow;
        const seg_desc_t *src = row + (lane < kHeadSlots ? 2 + lane : 0);
        const unsigned char *msrc = reinterpret_cast<const unsigned char *>(row + kHeadMapAt) + tid;
        asm volatile("global_load_dwordx2 %0, %2, off\n\tglobal_load_ubyte %1, %3, off"
                     : "=&v"(d), "=&v"(vmap)
                     : "v"(src), "v"(msrc)
                     : "memory");
    }
    if constexpr (FUSED) {
        // The thread's chunk of the tile's row (rp_pieces.hpp) arrives with the head row: one wait
        // for both, here -- the DMA issued next must not sit in front of them.
        const long long tile_lo = pp.tile_lo[2 * b];
        const long long n_chunks = pp.tile_lo[2 * b + 1];  // (count | kTileWide)
        chunk_desc_t mine;
        asm volatile("global_load_dwordx2 %0, %1, off" : "=&v"(mine) : "v"(chunk_slot(pp, b, lane, wave)) : "memory");
#if RP_ROW_PREFETCH > 0 && defined(RP_ROW_PREFETCH_EARLY)
        {   // (A/B: the same prefetch issued at ENTRY, beside the tile's own row loads -- it returns with them)
            const long long bn = b + RP_ROW_PREFETCH;
            if (wave == 0 && bn < plan.n_tiles && lane < 8 + kMaxChunks / 16) {
                const char *line = lane < 7    ? reinterpret_cast<const char *>(ws.head + bn * kHeadRow) + 128 * lane
                                   : lane < 7 + kMaxChunks / 16 ? reinterpret_cast<const char *>(pp.rows + bn * kMaxChunks) + 128 * (lane - 7)
                                               : reinterpret_cast<const char *>(pp.tile_lo + 2 * bn);
                asm volatile("global_load_dword %0, %1, off" : "=v"(prefetched) : "v"(line) : "memory");
            }
        }
#endif
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(d), "+v"(vmap), "+v"(mine), "+v"(prefetched) : : "memory");
        __builtin_amdgcn_s_setprio(3);  // scalar-heavy and on the critical path: ahead of the other workgroups' lane runs
        if (tile_lo != kTileSlow)
            stage_tile_chunks(counts, pp, b, tile_lo, n_chunks, mine, s_counts, lane, wave);
        else
            stage_tile_slow<TILE, kHalo>(counts, pp, b, plan.total_nt, s_counts, lane, wave);
        __builtin_amdgcn_s_setprio(0);
#if RP_ROW_PREFETCH > 0 && !defined(RP_ROW_PREFETCH_EARLY)
        // The plan rows of the tile this CU slot will most likely run NEXT (b + the number of workgroups in flight; a
        // multiple of 8, so it is dispatched to the same XCD and finds the lines in ITS L2): 7 lines of head row, 22 of
        // chunk row, the tile_lo entry -- one load per line by wave 0, into a register nobody reads.  In the fused kernel
        // the rows are on the critical path (no DMA can be issued before the chunks are known): an L2 hit instead of an
        // HBM miss takes ~1 k cycles off a workgroup's ~13 k.  Issued BEHIND this tile's DMA, so it lands with it and
        // adds nothing to the wait at barrier 1.
        {
            const long long bn = b + RP_ROW_PREFETCH;
            if (wave == 0 && bn < plan.n_tiles && lane < 8 + kMaxChunks / 16) {
                const char *line = lane < 7    ? reinterpret_cast<const char *>(ws.head + bn * kHeadRow) + 128 * lane
                                   : lane < 7 + kMaxChunks / 16 ? reinterpret_cast<const char *>(pp.rows + bn * kMaxChunks) + 128 * (lane - 7)
                                               : reinterpret_cast<const char *>(pp.tile_lo + 2 * bn);
                asm volatile("global_load_dword %0, %1, off" : "=v"(prefetched) : "v"(line) : "memory");
            }
        }
#endif
    } else {
        load_tile_to_lds<TILE>(counts, t0, plan.total_nt, s_counts, tid);
    }
    RP_STAMP();  // 1: loads issued

    if (a1 - a0 + 1 <= kHeadSlots) {
        // ---- the common case: all segments of the tile sit in the head row ---------------------
        // Every wave works out what ITS 64 virtual lanes walk by itself, in registers (the plan's
        // lane map names the segment, a wave-wide scan + three lane fetches give the run), while
        // the tile is still streaming in: no table in LDS to wait for, nothing serial in front of
        // barrier 1 but the DMA.  (<= kHeadSlots segments need < 256 virtual lanes: one pass per wave.)
        // the two row loads came through inline asm (hipcc's own counter bookkeeping would put a
        // vmcnt(0) in front of every LDS access of the DMA-issuing waves' code path, i.e. of all
        // waves); tying their registers to the wait keeps every use behind it.  Waves without DMA
        // in flight get their rows as soon as they arrive; the loader waves wait for the tile too.
        if constexpr (!FUSED) asm volatile("s_waitcnt vmcnt(0)" : "+v"(d), "+v"(vmap) : : "memory");
        if (lane >= kHeadSlots) d = 0;
        RP_STAMP();  // 2: descriptors here
        const int lanes_i = (int)(d >> 53) & 0xff;
        const int incl = wave_add_scan(lanes_i);
        const int vs_i = incl - lanes_i;
        const int total_vl = __builtin_amdgcn_readlane(incl, kWave - 1);
        const int vbase = wave * kWave;
        const int vl = vbase + lane;
        const bool pass = vbase < total_vl;  // wave-uniform
        const bool active = vmap != 0xffu;    // == vl < total_vl
        const int seg = active ? (int)vmap : 0;
        int q0 = 0, lim = 0;
        if (pass) {
            // the segment's fields, from the lane that holds its descriptor
            const unsigned dlo = (unsigned)__builtin_amdgcn_ds_bpermute(seg << 2, (int)(unsigned)d);
            const unsigned dhi = (unsigned)__builtin_amdgcn_ds_bpermute(seg << 2, (int)(unsigned)(d >> 32));
            const int vs_s = __builtin_amdgcn_ds_bpermute(seg << 2, vs_i);
            const seg_desc_t ds = ((seg_desc_t)dhi << 32) | dlo;
            const int r = vl - vs_s;
            int n_run = ((int)(ds >> 26) & 0xfff) - r * kRun;
            n_run = n_run > kRun ? kRun : n_run;
            q0 = active ? ((int)ds & 0x1fff) + 3 * kRun * r : 0;
            const int rem0 = ((int)(ds >> 13) & 0x1fff) - q0;  // positions of the ORF from q0 on (clamped far end)
            lim = rem0 - 2 < 3 * n_run ? rem0 - 2 : 3 * n_run;
            if (!active) lim = 0;
        }
        if (wave == 0) {  // what the record stage needs, per slot
            const int part = (int)(d >> 51) & 3;
            s_vlstart[lane] = vs_i;
            if (lane == kWave - 1) s_vlstart[kSegChunk] = incl;
            s_tail[lane] = part ? (((int)(d >> 38) & 0x1fff) | (part << 16)) : -1;
            s_live[lane] = (int)(d >> 63);
            s_geom[lane] = seg_geom(d);
        }
        RP_STAMP();  // 3: mapped, arrived at barrier 1
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(prefetched) : : "memory");  // LDS-DMA completion is tracked by vmcnt only
        __syncthreads();
        RP_STAMP();  // 4: tile landed (barrier 1)
        if (pass) tile_pass<kRun>(s_counts, s_rec, q0, lim, active, seg, vl);
        RP_STAMP();  // 5: this wave's lane runs done
        __syncthreads();
        RP_STAMP();  // 6: all lane runs done (barrier 2)
        record_stage(s_counts, s_rec, s_vlstart, s_tail, s_live, s_geom, ws.rec, ws.n_rec, a0 - 1 + b, wave, lane);
        RP_STAMP();  // 7: records stored
        RP_STAMP_FLUSH();
        return;
    }

    // ---- many short ORFs: > kHeadSlots segments, 64 slots at a time ------------------------------
    // (128 slots a round -- two table sets, waves {0,1} and {2,3} each walking 64 slots side by side --
    // was built and measured in round 3: no gain on all-60-nt tiles, the longer runs eat the saved
    // round, and the extra table indirection cost the common path 6 %: profiles/archive/r03_ab_dual_rounds.txt)
    // Same scheme, minus the head row and the lane map: every wave reads the chunk's descriptors
    // from the per-segment array (slot L of chunk c = ORF a0 - 1 + 64 c + L) and finds its
    // lanes' segments through 64 private words of LDS (marks at the segments' first lanes, then
    // a max-scan); 64 segments need < 256 virtual lanes, so still one pass per wave.
    const long long n_slots = a1 - a0 + 1;
    seg_desc_t dc = 0;
    {
        const long long orf = a0 - 1 + lane;
        if (orf >= 0 && orf < a1) dc = ws.desc[orf + b];  // chunk 0, in flight with the tile
    }
#if RP_LANE_SEG
    // all segments short?  thread t looks at slots t, t + 256, ... (its own for the lane-per-segment rounds)
    seg_desc_t mine0 = 0;
    int too_long = 0;
    for (long long c0 = 0; c0 < n_slots; c0 += kTileBlock) {  // workgroup-uniform
        const long long orf = a0 - 1 + c0 + tid;
        seg_desc_t dx = 0;
        if (orf >= 0 && orf < a1) dx = ws.desc[orf + b];
        if (c0 == 0) mine0 = dx;
        too_long |= (int)((dx >> 63) != 0 && (((int)(dx >> 26) & 0xfff) > kLaneTrip));
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(prefetched), "+v"(dc), "+v"(mine0), "+v"(too_long) : : "memory");
    if (!__syncthreads_or(too_long)) {  // (also barrier 1: the tile has landed)
        for (long long c0 = 0; c0 < n_slots; c0 += kTileBlock) {
            seg_desc_t dx = mine0;
            if (c0 > 0) {
                const long long orf = a0 - 1 + c0 + tid;
                dx = orf < a1 ? ws.desc[orf + b] : 0;
            }
            lane_segments<TILE>(s_counts, dx, ws.rec, ws.n_rec, a0 - 1 + c0 + tid + b);
        }
        RP_STAMP_FLUSH();
        return;
    }
#else
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(prefetched) : : "memory");
#endif
    for (long long c0 = 0; c0 < n_slots; c0 += kSegChunk) {
        if (c0 > 0) {
            __syncthreads();  // the previous chunk's record stage is done with the tables
            const long long orf = a0 - 1 + c0 + lane;
            dc = orf < a1 ? ws.desc[orf + b] : 0;
        }
        // Run length of this round: the shortest of 5 / 9 / 15 triplets per lane that still puts the
        // 64 segments' lanes into one pass of the four waves.  Short ORFs then spread over all four
        // waves and every lane walks 2 or 3 blocks instead of 5 (a 60-nt ORF: 4 lanes of 5 triplets
        // instead of 15 + 5).
        const int ntrip_i = (int)(dc >> 26) & 0xfff;
        const int live_i = (int)(dc >> 63);
        const int total5 = __builtin_amdgcn_readlane(wave_add_scan(live_i ? (ntrip_i + 4) / 5 : 0), kWave - 1);
        const int total9 = __builtin_amdgcn_readlane(wave_add_scan(live_i ? (ntrip_i + 8) / 9 : 0), kWave - 1);
        if (total5 <= kTileBlock)
            short_round<5>(dc, s_counts, s_rec, s_vlstart, s_tail, s_live, s_geom, s_owner, wave, lane);
        else if (total9 <= kTileBlock)
            short_round<9>(dc, s_counts, s_rec, s_vlstart, s_tail, s_live, s_geom, s_owner, wave, lane);
        else
            short_round<kRun>(dc, s_counts, s_rec, s_vlstart, s_tail, s_live, s_geom, s_owner, wave, lane);
#ifdef RP_STAMPS
        if (n_stamp_ < 7) RP_STAMP();  // short path: 2, 4, 6 = after barrier 2 of rounds 0, 1, 2
#endif
        record_stage(s_counts, s_rec, s_vlstart, s_tail, s_live, s_geom, ws.rec, ws.n_rec, a0 - 1 + c0 + b, wave, lane);
#ifdef RP_STAMPS
        if (n_stamp_ < 7) RP_STAMP();  // short path: 3, 5, 7 = records of rounds 0, 1, 2 stored
#endif
    }
    RP_STAMP_FLUSH();
}

#ifndef RP_TILES_PER_WG
#define RP_TILES_PER_WG 1  // > 1: a workgroup takes that many consecutive tiles one after the other (A/B knob)
#endif
template <bool FUSED, int TILE>
__device__ __forceinline__ void tile_score_workgroup(const int32_t *__restrict__ counts, const TilePlan &plan, const TileWorkspace &ws,
                                                     const PiecePlan &pp)
{
    __shared__ __attribute__((aligned(16))) int s_counts[lds_counts<TILE>()];
    __shared__ int s_live[kSegChunk];     // the slot holds a segment of this tile
    __shared__ int s_tail[kSegChunk];     // LDS index of an owned partial last codon | its length << 16, or -1
    __shared__ int s_vlstart[kSegChunk + 1];
    __shared__ int s_owner[kTileBlock];   // short-ORF path: 64 private words per wave for the segment marks
    __shared__ RunRec s_rec[kMaxRecs];
    __shared__ int s_geom[kSegChunk];     // valid codon-start positions of the slot's segment (record stage)
#if RP_TILES_PER_WG == 1
    tile_body<FUSED, TILE>(counts, plan, ws, pp, (long long)blockIdx.x, s_counts, s_live, s_tail, s_vlstart, s_owner, s_rec, s_geom);
#else
    for (int k = 0; k < RP_TILES_PER_WG; ++k) {
#ifdef RP_TILES_STRIDED
        const long long b = (long long)k * gridDim.x + blockIdx.x;
#else
        const long long b = (long long)blockIdx.x * RP_TILES_PER_WG + k;
#endif
        if (b >= plan.n_tiles) break;  // (workgroup-uniform)
        if (k) __syncthreads();        // the previous tile's record stage is done with LDS
        tile_body<FUSED, TILE>(counts, plan, ws, pp, b, s_counts, s_live, s_tail, s_vlstart, s_owner, s_rec, s_geom);
    }
#endif
}

template <bool FUSED, int TILE>
__global__ __launch_bounds__(kTileBlock, RP_MIN_WAVES) void k_tile_score(const int32_t *__restrict__ counts,
                                                           long long n_orfs, TilePlan plan,
                                                           TileWorkspace ws, PiecePlan pp)
{
    tile_score_workgroup<FUSED, TILE>(counts, plan, ws, pp);
}

// The same kernel under a second name: launches issued while rp_measurement_tag(1) is in force (a search
// like engine.tune_workspace, which times the scorer on candidate placements) use this instantiation, so
// that a profiler's per-kernel statistics of k_tile_score hold the production launches only.
template <bool FUSED, int TILE>
__global__ __launch_bounds__(kTileBlock, RP_MIN_WAVES) void k_tile_score_probe(const int32_t *__restrict__ counts,
                                                                 long long n_orfs, TilePlan plan,
                                                                 TileWorkspace ws, PiecePlan pp)
{
    tile_score_workgroup<FUSED, TILE>(counts, plan, ws, pp);
}

// ---------------------------------------------------------------------------
// pass 3: one thread per ORF -- add the records of the tiles it spans, score, filter,
// store.  Too-close-to-call ORFs (~0.4 %) are re-walked in float64 by the wave that found
// them, one after the other: short ones on the spot, long ones queued for pass 4.
// The re-walks stay INSIDE this pass: spread over its 172 000 waves they overlap the other waves' record reads; every split
// into a queue + drain kernel (rounds 3-5) was slower.  What round 6 changed is WHICH ORFs are
// re-walked: with RP_FILTER_PRINTED_ONLY a too-close-to-call ORF that cannot be translating is left at its fp32 result.
// The fused path (CoverageSource) first copies the ORF's profile out of the coverage into LDS,
// piece by piece and coalesced ('-' strand pieces backwards): walk and replay then read plain
// LDS instead of finding the piece of every position they touch.
// ---------------------------------------------------------------------------
#ifndef RP_FINISH_BLOCK
#define RP_FINISH_BLOCK 64
#endif
constexpr int kFinishBlock = RP_FINISH_BLOCK;  // threads per workgroup; the waves of a workgroup never synchronise (a re-walk holds up nobody else)
static_assert(kFinishBlock % kWave == 0 && kFinishBlock <= 1024, "whole waves");
#ifndef RP_STAGE_NT
#define RP_STAGE_NT 632
#endif
// fused path: profiles up to this long are copied to LDS first.  632 nt = 2.5 KB per wave: with the replay's 4.7 KB that
// leaves room for five one-wave workgroups per SIMD (20 per CU), which the fused flavour is then asked to fit its registers
// into (95 VGPRs instead of 103, 12 bytes of scratch per lane): every wave slot counts in this pass.
// Measured against 1 016 nt / four waves (profiles/archive/r05_ab_finish_occupancy.txt): fused finish 0.413 -> 0.388 ms; 760 nt:
// 0.405; 504 nt: 0.393; 440 nt / six waves: 0.457 (spills).  Longer profiles are read through their pieces.
constexpr int kStageNt = RP_STAGE_NT;
#ifndef RP_FINISH_WAVES_FUSED
#define RP_FINISH_WAVES_FUSED 5
#endif

// copy the profile of ORF `orf` out of the coverage into `stage`, piece by piece, coalesced; the
// piece descriptors of up to 63 pieces are fetched lane-parallel first (one round trip, not one
// per piece)
__device__ __forceinline__ void stage_profile(const CoverageSource &source, long long orf, long long beg, int *stage, int lane)
{
    const PiecePlan &pp = source.pp;
    const long long j0 = pp.orf_piece[orf], j1 = pp.orf_piece[orf + 1];
    for (long long jb = j0; jb < j1; jb += kWave - 1) {  // wave-uniform
        const int cnt = (int)(j1 - jb < kWave - 1 ? j1 - jb : kWave - 1);
        const long long sw_l = lane <= cnt ? (long long)pp.start[jb + lane] : 0;  // (start[j1] exists: the next piece or the sentinel)
        const long long bs_l = lane < cnt ? pp.base[jb + lane] : 0;
        for (int t = 0; t < cnt; ++t) {
            const unsigned long long sw = (unsigned long long)readlane64(sw_l, t);
            const long long s = (long long)(sw & ~kPieceNeg);
            const long long e = (long long)((unsigned long long)readlane64(sw_l, t + 1) & ~kPieceNeg);
            const long long base = readlane64(bs_l, t);
            const bool neg = (sw & kPieceNeg) != 0;
            for (long long pos = s + lane; pos < e; pos += kWave) stage[pos - beg] = source.cov[neg ? base - pos : base + pos];
        }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // (one wave: its LDS operations complete in order)
}

// float64 walk of one too-close-to-call ORF by one wave, the tie replay if need be, and the stores
// (the integer results of the fp32 pass stand: they are exact)
#ifdef RP_REWALK_STAMPS  // timing experiment: where does a re-walk's time go?  (s_memtime deltas summed per phase)
__device__ unsigned long long g_rewalk_stamps[16];
#define RP_RW_STAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#else
#define RP_RW_STAMP(var)
#endif
template <typename Counts>
__device__ __forceinline__ void finish_unsafe(Counts v, long long orf, long long len, long long count, int min_codon, unsigned split,
                                              int lane, ReplayLds *replay, const OrfOutputs &out, const FilterParams &fp)
{
    RP_RW_STAMP(ts0);
    WalkResult<double> w;
    wave_walk<double>(v, len, lane, w);
    RP_RW_STAMP(ts1);
    FrameScore fr2[3];
#pragma unroll
    for (int f = 0; f < 3; ++f)
        fr2[f] = frame_score(wave_sum(w.acc[f].p), wave_sum(w.acc[f].q), wave_sum(w.acc[f].n), wave_sum(w.acc[f].m));
    double phase;
    int valid;
    unsigned flags;
    combine_frames(fr2, phase, valid, flags);
    RP_RW_STAMP(ts2);
    if (flags & RP_FLAG_TIE) {
        const bool big = replay_tie_wave(v, len, lane, phase, valid, replay);
        flags |= RP_FLAG_REPLAY | (big ? RP_FLAG_BIGTIE : 0u);
    }
    RP_RW_STAMP(ts3);
    if (lane == 0) store_orf(out, fp, orf, phase, valid, count, min_codon, flags | split | RP_FLAG_RECHECK64, len);
#ifdef RP_REWALK_STAMPS
    if (lane == 0) {
        const int t = (flags & RP_FLAG_TIE) ? 8 : 0;
        atomicAdd(&g_rewalk_stamps[t + 0], 1ull);
        atomicAdd(&g_rewalk_stamps[t + 1], ts1 - ts0);
        atomicAdd(&g_rewalk_stamps[t + 2], ts2 - ts1);
        atomicAdd(&g_rewalk_stamps[t + 3], ts3 - ts2);
        atomicAdd(&g_rewalk_stamps[t + 4], (unsigned long long)len);
    }
#endif
}

#ifndef RP_FINISH_WAVES
#define RP_FINISH_WAVES 1
#endif
template <int TILE, typename Source>
__global__ __launch_bounds__(kFinishBlock, sizeof(Source) != sizeof(CsrSource) ? RP_FINISH_WAVES_FUSED : RP_FINISH_WAVES) void k_orf_finish(Source source,
                                                             const int64_t *__restrict__ offsets,
                                                             long long n_orfs, TilePlan plan,
                                                             TileWorkspace ws, OrfOutputs out,
                                                             FilterParams fp)
{
    constexpr bool kStaged = sizeof(Source) != sizeof(CsrSource);  // the fused path reads through the gather plan
    constexpr int kWaves = kFinishBlock / kWave;
    __shared__ ReplayLds s_replay_w[kWaves];  // (per wave: the waves of a workgroup work independently)
    __shared__ int s_stage_w[kWaves][kStaged ? kStageNt + 8 : 1];
    const int lane = threadIdx.x & (kWave - 1);
    ReplayLds &s_replay = s_replay_w[threadIdx.x / kWave];
    int *const s_stage = s_stage_w[threadIdx.x / kWave];
    // One batch of 64 ORFs per one-wave workgroup: a one-shot grid hands a finished wave's slot to the next batch, which
    // balances the tail of re-walking waves; persistent looping waves were slower.
    const long long orf = (long long)blockIdx.x * kFinishBlock + threadIdx.x;
    long long beg = 0, len = 0, count = 0;
    int min_codon = RP_MIN_CODON_COV_EMPTY;
    unsigned split = 0;
    bool unsafe = false;
    if (orf < n_orfs) {
        beg = offsets[orf];
        len = (long long)offsets[orf + 1] - beg;
        double p[3] = {0, 0, 0}, q[3] = {0, 0, 0};
        int n[3] = {0, 0, 0}, m[3] = {0, 0, 0};
        FrameScore fr[3];
        if (len > 0) {
            const long long b_first = tile_of<TILE>(beg + plan.mis);
            const long long b_last = tile_of<TILE>(beg + len - 1 + plan.mis);
            for (long long b = b_first; b <= b_last; ++b) {  // tile order: deterministic sums
                const uint4 w0 = ws.rec[rec_index(ws.n_rec, orf + b, 0)], w1 = ws.rec[rec_index(ws.n_rec, orf + b, 1)],
                            w2 = ws.rec[rec_index(ws.n_rec, orf + b, 2)];
                p[0] += (double)__uint_as_float(w0.x);
                q[0] += (double)__uint_as_float(w0.y);
                p[1] += (double)__uint_as_float(w1.x);
                q[1] += (double)__uint_as_float(w1.y);
                p[2] += (double)__uint_as_float(w2.x);
                q[2] += (double)__uint_as_float(w2.y);
                n[0] += (int)(w0.z & 0xffffu);
                m[0] += (int)(w0.z >> 16);
                n[1] += (int)(w1.z & 0xffffu);
                m[1] += (int)(w1.z >> 16);
                n[2] += (int)(w2.z & 0xffffu);
                m[2] += (int)(w2.z >> 16);
                count += (long long)(((unsigned long long)w2.w << 16) + w0.w);
                min_codon = min(min_codon, (int)w1.w);
            }
            if (b_last > b_first) split = RP_FLAG_SPLIT;
        }
#pragma unroll
        for (int f = 0; f < 3; ++f) fr[f] = frame_score(p[f], q[f], n[f], m[f]);
        double phase;
        int valid;
        unsigned flags;
        combine_frames(fr, phase, valid, flags);
        // re-walk in float64 when the frame decision OR the cutoff comparison is too close to call
        unsafe = fp32_decision_unsafe(fr, /*tile_sums=*/true) || near_cutoff(fp, phase);
        unsigned left_open = 0;
        if (unsafe && fp.printed_only && cannot_be_translating(fp, fr, count, min_codon, len)) {
            // too close to call, and no way of calling it makes the ORF translating: in default mode it prints nothing
            // (detect_orfs.py:301-302) -- the fp32 results stand, marked, and no wave spends ~20 us on its re-walk
            unsafe = false;
            left_open = RP_FLAG_UNRESOLVED;
        }
        if (!unsafe) store_orf(out, fp, orf, phase, valid, count, min_codon, flags | split | left_open, len);
    }

    // The too-close-to-call ORFs of this wave (~0.4 %), one after the other, by the whole wave:
    // float64 walk from global memory (the integer results stand: they are exact), then -- on an
    // exact frame tie -- the replay of the reference's own arithmetic (rp_device.hpp).
    unsigned long long todo = __ballot(unsafe);
#ifdef RP_EXPERIMENT_NO_REWALK  // timing experiment only (results wrong): what the in-wave re-walks cost the finish pass
    todo = 0;
#endif
#ifdef RP_EXPERIMENT_MAX_REWALKS  // timing experiment only (results wrong): a wave re-walks at most this many of its ORFs
    for (int keep = 0; keep < RP_EXPERIMENT_MAX_REWALKS; ++keep) {}
    {
        unsigned long long kept = 0, rest = todo;
        for (int k = 0; k < RP_EXPERIMENT_MAX_REWALKS && rest != 0; ++k) {
            kept |= rest & (0ull - rest);
            rest &= rest - 1;
        }
        todo = kept;
    }
#endif
    while (todo != 0) {
        const int l = __builtin_ctzll(todo);
        todo &= todo - 1;
        const long long orf_s = readlane64(orf, l);
        const long long beg_s = readlane64(beg, l);
        const long long len_s = readlane64(len, l);
        const long long count_s = readlane64(count, l);
        const int min_s = __builtin_amdgcn_readlane(min_codon, l);
        const unsigned split_s = (unsigned)__builtin_amdgcn_readlane((int)split, l);
        if (len_s > kLongWalk) {  // a whole workgroup takes it (k_rewalk_long)
            if (lane == 0) ws.long_list[atomicAdd(ws.long_count, 1)] = orf_s;
            continue;
        }
        if constexpr (kStaged) {
            if (len_s <= kStageNt) {
                stage_profile(source, orf_s, beg_s, s_stage, lane);
                finish_unsafe(static_cast<const int *>(s_stage), orf_s, len_s, count_s, min_s, split_s, lane, &s_replay, out, fp);
                __builtin_amdgcn_wave_barrier();  // (the stage is rewritten by the next item)
                continue;
            }
        }
        finish_unsafe(source.orf(orf_s, beg_s), orf_s, len_s, count_s, min_s, split_s, lane, &s_replay, out, fp);
    }
}

// ---------------------------------------------------------------------------
// pass 4: the queued long too-close-to-call ORFs (rare: a handful per million), one
// 1024-thread workgroup each -- float64 walk by all 16 waves, then, on an exact frame tie,
// the replay of the reference's arithmetic by wave 0.  Grid-stride over the queue.
// ---------------------------------------------------------------------------
template <int TILE, typename Source>
__global__ __launch_bounds__(kLongBlock) void k_rewalk_long(Source source,
                                                            const int64_t *__restrict__ offsets, TilePlan plan,
                                                            TileWorkspace ws, OrfOutputs out, FilterParams fp)
{
    constexpr int kWaves = kLongBlock / kWave;
    __shared__ double s_part[kWaves][6];
    __shared__ int s_parti[kWaves][7];
    __shared__ long long s_partc[kWaves];
    __shared__ ReplayLds s_replay;  // (wave 0 replays)
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int n_long = *ws.long_count;
    for (int e = blockIdx.x; e < n_long; e += gridDim.x) {  // workgroup-uniform
        const long long orf = ws.long_list[e];
        const long long beg = offsets[orf];
        const long long len = (long long)offsets[orf + 1] - beg;
        WalkResult<double> w;
        wave_walk<double>(source.orf(orf, beg), len, (int)threadIdx.x, w, kLongBlock);
        // per-wave sums -> LDS -> every thread adds the partials in the same order
#pragma unroll
        for (int f = 0; f < 3; ++f) {
            const double ps = wave_sum(w.acc[f].p), qs = wave_sum(w.acc[f].q);
            const int ns = wave_sum(w.acc[f].n), ms = wave_sum(w.acc[f].m);
            if (lane == 0) {
                s_part[wave][2 * f] = ps;
                s_part[wave][2 * f + 1] = qs;
                s_parti[wave][2 * f] = ns;
                s_parti[wave][2 * f + 1] = ms;
            }
        }
        const long long cs = wave_sum(w.count);
        const int mins = wave_min(w.min_codon);
        if (lane == 0) {
            s_partc[wave] = cs;
            s_parti[wave][6] = mins;
        }
        __syncthreads();
        FrameScore fr[3];
        long long count = 0;
        int min_codon = RP_MIN_CODON_COV_EMPTY;
#pragma unroll
        for (int f = 0; f < 3; ++f) {
            double ps = 0.0, qs = 0.0;
            int ns = 0, ms = 0;
            for (int wv = 0; wv < kWaves; ++wv) {
                ps += s_part[wv][2 * f];
                qs += s_part[wv][2 * f + 1];
                ns += s_parti[wv][2 * f];
                ms += s_parti[wv][2 * f + 1];
            }
            fr[f] = frame_score(ps, qs, ns, ms);
        }
        for (int wv = 0; wv < kWaves; ++wv) {
            count += s_partc[wv];
            min_codon = min(min_codon, s_parti[wv][6]);
        }
        __syncthreads();  // the partial slots are reused by the next item
        if (wave != 0) continue;
        double phase;
        int valid;
        unsigned flags;
        combine_frames(fr, phase, valid, flags);
        if (flags & RP_FLAG_TIE) {
            const bool big = replay_tie_wave(source.orf(orf, beg), len, lane, phase, valid, &s_replay);
            flags |= RP_FLAG_REPLAY | (big ? RP_FLAG_BIGTIE : 0u);
        }
        const unsigned split = (beg + plan.mis) / TILE != (beg + len - 1 + plan.mis) / TILE ? RP_FLAG_SPLIT : 0u;
        if (lane == 0) store_orf(out, fp, orf, phase, valid, count, min_codon, flags | split | RP_FLAG_RECHECK64, len);
    }
}

}  // namespace rp
