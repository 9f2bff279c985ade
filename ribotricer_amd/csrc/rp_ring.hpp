// rp_ring.hpp -- the flat-tile scorer as PERSISTENT workgroups with a two-slot LDS ring.
//
// k_tile_score (rp_tile.hpp) launches one workgroup per tile: its life is load (36 %), lane runs,
// records -- and while it computes it has nothing in flight; with four of them per CU about 1.5
// tiles (46 KB) are being loaded at any time, short of what 8 TB/s times the loaded latency asks
// for, and every tile pays a workgroup launch plus a dependent plan-row load before its first
// DMA.  Here a workgroup owns TWO tile images in LDS and walks tiles b, b + G, b + 2G, ...:
//
//   barrier 1 (tile k and its head row have landed; everybody is done with tile k-1)
//   -> the loader wave issues the 31 LDS-DMA rows of tile k+1 into the other image, and its 640-byte
//      head row into the other row slot -- source addresses depend on the tile number only, nothing
//      is waited for
//   -> every wave maps its virtual lanes from the head row of tile k (LDS -> registers), then its
//      lane runs of tile k out of LDS (ds_read only: the DMA of tile k+1 stays in flight behind them)
//   -> barrier 2 -> the loader's s_waitcnt vmcnt(0) for tile k+1, BEFORE the record stores of
//      tile k (stores count on vmcnt: waited for behind them, every tile would sit out their
//      round trip) -> record stage of tile k
//
// so each of the two workgroups of a CU keeps a whole tile (31 KB) in flight ALL the time, plan
// rows are a tile ahead, and there is one launch per 1 000 tiles.  The per-tile work is the code
// of rp_tile.hpp (tile_pass, record_stage, short_round), unchanged.  No load of this kernel's
// steady state has a register destination: nothing in flight can be copied or spilled by the
// compiler (an asm load's destination counts as written when the statement ends).
//
// The DMA goes through inline asm: hipcc puts a vmcnt(0) in front of every LDS access once it
// has seen a global_load_lds builtin in flight, which would serialise exactly the overlap this
// kernel exists for.  What the compiler cannot see it cannot order either, hence the explicit
// waits: vmcnt(0) before barrier 1 (the only wait that covers a DMA: LDS-DMA and ordinary loads
// retire out of order with respect to each other on gfx950).
//
// LDS hazards across tiles: everything written AFTER barrier 1 of tile k (the slot tables, the row
// records, the scores) is safe single-buffered -- a wave reaches barrier 1 of tile k only after
// its record stage of tile k-1; what is written BEFORE it (the zeroed integer accumulators) is
// double-buffered.
#pragma once

#include "rp_tile.hpp"

namespace rp {

#ifndef RP_RING_WGS
#define RP_RING_WGS 2  // workgroups per CU (2 x (2 x 31.3 KB + 9 KB of tables) = 143 KB of LDS)
#endif
constexpr int kRingWgsPerCu = RP_RING_WGS;

// Rows of one interior tile as LDS-DMA, issued by ONE wave from straight-line code: M0 (the LDS
// destination of the row), the lane's byte offset, one global_load_lds_dwordx4 with a scalar base.
// `src` = counts + t0 (16-byte aligned, wave-uniform).  Nothing is waited for.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
template <int TILE>
__device__ __forceinline__ void ring_issue_rows(const int32_t *src, int *s_dst, int lane)
{
    constexpr int kRows = TILE / 256;  // 1 KiB rows
    const unsigned lds0 = lds_address(s_dst);
    const int32_t *base = scalar_ptr(src);
    const unsigned voff = (unsigned)lane * 16u;
#pragma unroll
    for (int row = 0; row < kRows; ++row) {
        asm volatile("s_mov_b32 m0, %1\n\t"
                     "s_nop 0\n\t"
                     "global_load_lds_dwordx4 %0, %2 nt"
                     :
                     : "v"(voff + (unsigned)row * 1024u), "s"(lds0 + (unsigned)row * 1024u), "s"(base)
                     : "memory", "m0");
    }
    // the halo: two 16-byte chunks past the tile (lanes 0 and 1)
    static_assert(kHalo == 8, "halo is loaded as two extra chunks");
    asm volatile("s_mov_b32 m0, %1\n\t"
                 "s_mov_b64 exec, 3\n\t"
                 "global_load_lds_dwordx4 %0, %2 nt\n\t"
                 "s_mov_b64 exec, -1"
                 :
                 : "v"(voff + (unsigned)kRows * 1024u), "s"(lds0 + (unsigned)kRows * 1024u), "s"(base)
                 : "memory", "m0", "exec");
}
#pragma clang diagnostic pop

// Stage tile b into `s_dst`: interior tiles by the loader wave's DMA (not waited for), the first /
// last tile by everybody through registers with zero fill.  Workgroup-uniform control flow.
template <int TILE>
__device__ __forceinline__ void ring_issue_tile(const int32_t *__restrict__ counts, long long b, const TilePlan &plan,
                                                int *s_dst, int tid)
{
    const long long t0 = b * (long long)TILE - plan.mis;
    const bool interior = (t0 >= 0) && (t0 + TILE + kHalo <= plan.total_nt);
    if (interior) {
        if (__builtin_amdgcn_readfirstlane(tid >> 6) == kTileBlock / kWave - 1) ring_issue_rows<TILE>(counts + t0, s_dst, tid & (kWave - 1));
    } else {
        constexpr int n_chunks = (TILE + kHalo) / 4;
#pragma unroll 1
        for (int c = tid; c < n_chunks; c += kTileBlock) {
            const long long pos = t0 + 4LL * c;
            int4 v = make_int4(0, 0, 0, 0);
            if (pos + 0 >= 0 && pos + 0 < plan.total_nt) v.x = counts[pos + 0];
            if (pos + 1 >= 0 && pos + 1 < plan.total_nt) v.y = counts[pos + 1];
            if (pos + 2 >= 0 && pos + 2 < plan.total_nt) v.z = counts[pos + 2];
            if (pos + 3 >= 0 && pos + 3 < plan.total_nt) v.w = counts[pos + 3];
            *reinterpret_cast<int4 *>(s_dst + 4 * c) = v;
        }
    }
}

__device__ __forceinline__ long long first_lane_i64(unsigned lo, unsigned hi)
{
    return (long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)hi) << 32) |
                       (unsigned)__builtin_amdgcn_readfirstlane((int)lo));
}

// The 640-byte head row of tile b into LDS, by the loader wave: one dwordx4 DMA, lanes 0..39.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void ring_issue_head(const seg_desc_t *head, long long b, seg_desc_t *s_dst, int lane)
{
    static_assert(kHeadRow * sizeof(seg_desc_t) <= 1024 && (kHeadRow * sizeof(seg_desc_t)) % 16 == 0, "one masked dwordx4 DMA per head row");
    constexpr unsigned long long kMask = (1ull << (kHeadRow * sizeof(seg_desc_t) / 16)) - 1ull;
    const unsigned lds0 = lds_address(reinterpret_cast<const int *>(s_dst));
    const int32_t *base = scalar_ptr(reinterpret_cast<const int32_t *>(head + b * kHeadRow));
    asm volatile("s_mov_b32 m0, %1\n\t"
                 "s_mov_b64 exec, %3\n\t"
                 "global_load_lds_dwordx4 %0, %2\n\t"
                 "s_mov_b64 exec, -1"
                 :
                 : "v"((unsigned)lane * 16u), "s"(lds0), "s"(base), "s"(kMask)
                 : "memory", "m0", "exec");
}
#pragma clang diagnostic pop

// Phase stamps (-DRP_STAMPS): round kRingStampIt of every workgroup stores its shader-clock times
#ifdef RP_STAMPS
constexpr int kRingStampIt = 8;
#define RP_RSTAMP(k)                                                                                         \
    do {                                                                                                     \
        if (it == kRingStampIt && lane == 0 && blockIdx.x < kStampSlots) rp_dbg_stamps[blockIdx.x][wave][k] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define RP_RSTAMP(k) do {} while (0)
#endif

template <int TILE>
__global__ __launch_bounds__(kTileBlock, kRingWgsPerCu) void k_tile_score_ring(const int32_t *__restrict__ counts,
                                                                              long long n_orfs, TilePlan plan, TileWorkspace ws,
                                                                              OrfOutputs out, FilterParams fp)
{
    __shared__ __attribute__((aligned(16))) int s_img[2][lds_counts<TILE>()];  // the ring: two tile images ...
    __shared__ __attribute__((aligned(16))) seg_desc_t s_head[2][kHeadRow];     // ... and their head rows
    __shared__ int s_live[kSegChunk];
    __shared__ int s_tail[kSegChunk];
    __shared__ int s_vlstart[kSegChunk + 1];
    __shared__ int s_owner[kTileBlock];
    __shared__ RunRec s_rec[kMaxRecs];
    __shared__ SegInts s_ints2[2][kSegChunk];  // zeroed before barrier 1: double-buffered (see the file header)
    __shared__ double s_score[3][kSegChunk];

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = tid >> 6;
    const bool loader = __builtin_amdgcn_readfirstlane(wave) == kTileBlock / kWave - 1;
    const long long stride = gridDim.x;
    long long b = blockIdx.x;
    if (blockIdx.x == 0 && tid == 0) *ws.long_count = 0;  // (k_orf_finish, next in the stream, appends)
    if (b >= plan.n_tiles) return;

    // prologue: the first tile and its head row
    ring_issue_tile<TILE>(counts, b, plan, s_img[0], tid);
    if (loader) ring_issue_head(ws.head, b, s_head[0], lane);
    if (wave == 0) {
        s_ints2[0][lane].nn = 0;
        s_ints2[0][lane].mm = 0;
        s_ints2[0][lane].count = 0;
        s_ints2[0][lane].min_codon = (unsigned)RP_MIN_CODON_COV_EMPTY;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    for (int it = 0;; ++it) {  // workgroup-uniform
        RP_RSTAMP(7);  // arrival at barrier 1 (of the stamped round: how long the previous record stage + wait took)
        __syncthreads();  // barrier 1: tile `b` and its head row have landed; every wave is past the previous tile
        RP_RSTAMP(0);
        const int *s_counts = s_img[it & 1];
        const seg_desc_t *row = s_head[it & 1];
        SegInts *s_ints = s_ints2[it & 1];
        const long long nb = b + stride;
        const bool more = nb < plan.n_tiles;
        const long long a0 = first_lane_i64((unsigned)row[0], (unsigned)(row[0] >> 32));
        const long long a1 = first_lane_i64((unsigned)row[1], (unsigned)(row[1] >> 32));
        const bool common = a1 - a0 + 1 <= kHeadSlots;
        // the next tile goes into the other image now (its source address depends on the tile number only)
        if (more) {
            ring_issue_tile<TILE>(counts, nb, plan, s_img[(it + 1) & 1], tid);
            if (loader) ring_issue_head(ws.head, nb, s_head[(it + 1) & 1], lane);
        }
        RP_RSTAMP(1);  // next tile issued
        if (common) {
            // every wave maps its 64 virtual lanes in registers (as k_tile_score does), from the row in LDS
            const seg_desc_t d = lane < kHeadSlots ? row[2 + lane] : 0;
            const unsigned vmap = reinterpret_cast<const unsigned char *>(row + kHeadMapAt)[tid];
            const int lanes_i = (int)(d >> 53) & 0xff;
            const int incl = wave_add_scan(lanes_i);
            const int vs_i = incl - lanes_i;
            const int total_vl = __builtin_amdgcn_readlane(incl, kWave - 1);
            const int vbase = wave * kWave;
            const int vl = vbase + lane;
            const bool pass = vbase < total_vl;
            const bool active = vmap != 0xffu;
            const int seg = active ? (int)vmap : 0;
            int q0 = 0, lim = 0;
            if (pass) {
                const unsigned dlo = (unsigned)__builtin_amdgcn_ds_bpermute(seg << 2, (int)(unsigned)d);
                const unsigned dhi = (unsigned)__builtin_amdgcn_ds_bpermute(seg << 2, (int)(unsigned)(d >> 32));
                const int vs_s = __builtin_amdgcn_ds_bpermute(seg << 2, vs_i);
                const seg_desc_t ds = ((seg_desc_t)dhi << 32) | dlo;
                const int r = vl - vs_s;
                int n_run = ((int)(ds >> 26) & 0xfff) - r * kRun;
                n_run = n_run > kRun ? kRun : n_run;
                q0 = active ? ((int)ds & 0x1fff) + 3 * kRun * r : 0;
                const int rem0 = ((int)(ds >> 13) & 0x1fff) - q0;
                lim = rem0 - 2 < 3 * n_run ? rem0 - 2 : 3 * n_run;
                if (!active) lim = 0;
            }
            if (wave == 0) {  // what the record stage needs, per slot (written after barrier 1, read after barrier 2)
                const int part = (int)(d >> 51) & 3;
                s_vlstart[lane] = vs_i;
                if (lane == kWave - 1) s_vlstart[kSegChunk] = incl;
                s_tail[lane] = part ? (((int)(d >> 38) & 0x1fff) | (part << 16)) : -1;
                s_live[lane] = live_word(d);
            }
            RP_RSTAMP(2);  // mapped
            if (pass) tile_pass<kRun>(s_counts, s_ints, s_rec, q0, lim, active, seg, vl);
            RP_RSTAMP(3);  // this wave's lane runs done
            __syncthreads();  // barrier 2
            RP_RSTAMP(4);
            // The loader waits for the next tile HERE, in front of this tile's record stores: stores
            // count on vmcnt too, and a wait behind them would sit out their round trip to L2 every
            // tile; like this they drain while the next tile is mapped and walked.
            if (loader) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            RP_RSTAMP(5);  // (loader) the next tile has landed
            record_stage(s_counts, s_ints, s_rec, s_vlstart, s_tail, s_live, s_score, ws.rec, ws.n_rec, a0 - 1 + b, a0 - 1, wave, lane, out, fp);
            RP_RSTAMP(6);  // records stored
        } else {
            // many short ORFs: 64 slots at a time through the per-segment descriptor array (rp_tile.hpp)
            const long long n_slots = a1 - a0 + 1;
            for (long long c0 = 0; c0 < n_slots; c0 += kSegChunk) {
                if (c0 > 0) {
                    __syncthreads();  // the previous chunk's record stage is done with the tables
                    if (wave == 0) {
                        s_ints[lane].nn = 0;
                        s_ints[lane].mm = 0;
                        s_ints[lane].count = 0;
                        s_ints[lane].min_codon = (unsigned)RP_MIN_CODON_COV_EMPTY;
                    }
                }
                const long long orf = a0 - 1 + c0 + lane;
                const seg_desc_t dc = (orf >= 0 && orf < a1) ? ws.desc[orf + b] : 0;
                const int ntrip_i = (int)(dc >> 26) & 0xfff;
                const int live_i = (int)(dc >> 63);
                const int total5 = __builtin_amdgcn_readlane(wave_add_scan(live_i ? (ntrip_i + 4) / 5 : 0), kWave - 1);
                const int total9 = __builtin_amdgcn_readlane(wave_add_scan(live_i ? (ntrip_i + 8) / 9 : 0), kWave - 1);
                if (total5 <= kTileBlock)
                    short_round<5>(dc, s_counts, s_ints, s_rec, s_vlstart, s_tail, s_live, s_owner, wave, lane);
                else if (total9 <= kTileBlock)
                    short_round<9>(dc, s_counts, s_ints, s_rec, s_vlstart, s_tail, s_live, s_owner, wave, lane);
                else
                    short_round<kRun>(dc, s_counts, s_ints, s_rec, s_vlstart, s_tail, s_live, s_owner, wave, lane);
                if (loader && c0 + kSegChunk >= n_slots) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (as above, before the last stores)
                record_stage(s_counts, s_ints, s_rec, s_vlstart, s_tail, s_live, s_score, ws.rec, ws.n_rec, a0 - 1 + c0 + b, a0 - 1 + c0, wave, lane, out, fp);
            }
        }
        if (!more) break;
        if (wave == 0) {  // the next tile's integer accumulators (the other buffer: nobody is using it)
            SegInts *nx = s_ints2[(it + 1) & 1];
            nx[lane].nn = 0;
            nx[lane].mm = 0;
            nx[lane].count = 0;
            nx[lane].min_codon = (unsigned)RP_MIN_CODON_COV_EMPTY;
        }
        b = nb;
    }
}

}  // namespace rp
