// rp_tile_pipe.hpp -- persistent, register-prefetching form of the flat-tile scorer.
//
// Same ownership rules, arithmetic and reduction as rp_tile.hpp (whose device helpers
// it reuses), different schedule.  k_tile_score is bound by the bytes it can keep in
// flight: a workgroup holds its 32 KiB LDS tile while it waits for HBM *and* while it
// computes, the 160 KiB of a CU take four such tiles, and during the ~35 % of its life a
// workgroup spends computing it has no load outstanding (measured: 4.0 TB/s against
// 6.2 TB/s for the same kernel with the arithmetic removed).  The LDS cannot hold more, but
// the register file (512 KiB per CU) can: here a workgroup stays resident, walks tiles
// b, b+G, b+2G, ... and keeps the NEXT tile in flight in VGPRs (global_load_dwordx4, ten
// per thread) for the whole time it computes the current one out of LDS:
//
//     commit: prefetched registers -> LDS (ds_write_b128), issue the prefetch of tile i+1
//     ------------------------------ barrier
//     lane runs of tile i            (waves 0..3)
//     ------------------------------ barrier
//     finish of tile i   (wave 0)  | segment table of tile i+1 (wave 1, from prefetched offsets)
//     ------------------------------ barrier
//
// Barriers are LDS-only (s_waitcnt lgkmcnt(0); s_barrier): a vmcnt wait would drain the
// prefetch.  Nothing in the steady state issues a global load other than the prefetch
// itself (tile_first comes through the scalar cache), so the in-order vmcnt never makes a
// wave wait for more than it needs.
#pragma once

#include "rp_tile.hpp"

namespace rp {

#ifndef RP_PIPE_TILE
#define RP_PIPE_TILE 10240
#endif
constexpr int kPipeTile = RP_PIPE_TILE;  // positions per tile: whole rounds of one int4 per thread
constexpr int kPipeRows = kPipeTile / (4 * kTileBlock);
static_assert(kPipeTile % (4 * kTileBlock) == 0, "tile must be whole int4 rounds");
static_assert(kPipeTile / 3 < 65536, "N_f / M_f of a segment are kept in 16-bit fields");
#ifndef RP_PIPE_BPC
#define RP_PIPE_BPC 3
#endif
constexpr int kPipeBlocksPerCu = RP_PIPE_BPC;  // LDS: 3 x ~52 KiB; VGPRs: 168 per lane
constexpr int kPipeLdsCounts = kPipeTile + kHalo + 3 * kRun + 12;
constexpr int kPipeMaxVl = kPipeTile / (3 * kRun) + kSegChunk + 2 * kWave;
constexpr int kPipeMaxRecs = kSegChunk + kPipeMaxVl / 16 + 1;

struct SegSet {  // the segment table of one tile chunk
    int qfirst[kSegChunk];     // LDS index of the first owned triplet
    int endq[kSegChunk];       // ORF end in LDS coordinates (clamped)
    int ntrip[kSegChunk];      // owned triplets
    int live[kSegChunk];       // the slot holds a segment of this tile
    int vlstart[kSegChunk + 1];
    int owner[kPipeMaxVl];  // slot+1 at the first lane of a segment / wave pass, else 0
    long long first_orf;    // ORF index of slot 0
    int more;               // ORFs that start in this tile beyond this chunk
    int pad;
};

// LDS-only workgroup barrier: every wave's LDS traffic is complete and visible, global
// loads stay in flight.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int kPipeSetupWave = 1;  // builds the next tile's segment table while wave 0 finishes

// One tile in flight in registers: kPipeRows int4 per thread (+ the 8-position halo in
// threads 0-1).
typedef int v4i __attribute__((ext_vector_type(4)));

#ifdef RP_PIPE_NT
#define RP_PIPE_LOAD(p) __builtin_nontemporal_load(p)
#else
#define RP_PIPE_LOAD(p) (*(p))
#endif

struct Prefetch {
    v4i row[kPipeRows];
    v4i halo;
};

// Issue (do not wait for) the loads of tile [t0, t0 + kPipeTile + halo) as whole 16-byte
// chunks of the aligned extent of the batch, [counts - mis, counts + total_nt) rounded out
// to 16 bytes: chunk indices are clamped into it, so the first tile (t0 = -mis) and the
// last one take the same path as every other.  Positions outside [0, total_nt) then hold
// neighbouring data instead of zeros; no owned codon reads them (runs past a segment's
// end are masked, and the values are finite).
__device__ __forceinline__ void pipe_prefetch(Prefetch &pf, const v4i *__restrict__ base, long long chunk0,
                                              long long last_chunk, int tid)
{
#pragma unroll
    for (int r = 0; r < kPipeRows; ++r) {
        long long c = chunk0 + r * kTileBlock + tid;
        c = c < last_chunk ? c : last_chunk;
        pf.row[r] = RP_PIPE_LOAD(base + c);
    }
    if (tid < kHalo / 4) {
        long long c = chunk0 + kPipeTile / 4 + tid;
        c = c < last_chunk ? c : last_chunk;
        pf.halo = RP_PIPE_LOAD(base + c);
    }
}

// Registers -> LDS tile (waits for the prefetch to land).
__device__ __forceinline__ void pipe_commit(const Prefetch &pf, int *s_counts, int tid)
{
    v4i *dst = reinterpret_cast<v4i *>(s_counts) + tid;
#pragma unroll
    for (int r = 0; r < kPipeRows; ++r) dst[r * kTileBlock] = pf.row[r];
    if (tid < kHalo / 4) reinterpret_cast<v4i *>(s_counts + kPipeTile)[tid] = pf.halo;
}

// The setup wave's loads of offsets[a0 - 1 + lane], [.. + 1] for a tile's first 64 slots.
// They are issued a whole stage before the prefetch they precede is committed, so the
// commit's own wait covers them.
__device__ __forceinline__ void pipe_load_bounds(const int64_t *__restrict__ offsets, long long a0, long long a1,
                                                 int lane, long long &beg, long long &end)
{
    const long long orf = a0 - 1 + lane;
    beg = 0;
    end = 0;
    if (orf >= 0 && orf < a1) {
        beg = offsets[orf];
        end = offsets[orf + 1];
    }
}

// Segment table of one chunk of tile b, by ONE wave.  Slot L holds ORF c0 + L; for the
// first chunk c0 = a0 - 1, i.e. slot 0 is the ORF that straddles in from the left (or is
// unused).  Slots past the last ORF that starts in the tile are unused.  `beg`/`end` are
// this lane's ORF bounds (ignored when the slot is out of range).
__device__ __forceinline__ void pipe_setup(SegSet &S, long long beg, long long end, long long a0,
                                           long long a1, long long c0, long long t0, long long t1, int lane)
{
    for (int k = lane; k < kPipeMaxVl; k += kWave) S.owner[k] = 0;
    const long long orf = c0 + lane;
    int lanes = 0;
    int live = 0;
    if (orf >= 0 && orf < a1) {
        const bool head = orf < a0;  // only ORF a0 - 1 can be
        if (!head || end > t0) {
            const int kt = (int)(t1 - t0);  // own range of the tile in positions
            int qfirst;
            if (head) {
                const unsigned long long rel0 = (unsigned long long)(t0 - beg);  // > 0
                const unsigned m3 = ((unsigned)(rel0 >> 32) % 3u + (unsigned)(rel0 & 0xffffffffu) % 3u) % 3u;  // 2^32 == 1 (mod 3)
                qfirst = m3 == 0 ? 0 : 3 - (int)m3;
            } else {
                qfirst = (int)(beg - t0);
            }
            const long long rem = end - t0;  // >= 0
            const int endq = rem <= kPipeTile + kHalo ? (int)rem : kPipeTile + kHalo;
            const int lim_q = kt < endq ? kt : endq;  // owned triplets start below this
            const int ntrip = lim_q > qfirst ? (lim_q - qfirst + 2) / 3 : 0;
            S.qfirst[lane] = qfirst;
            S.endq[lane] = endq;
            S.ntrip[lane] = ntrip;
            live = 1;
            lanes = (ntrip + kRun - 1) / kRun;
        }
    }
    S.live[lane] = live;
    const int incl = wave_add_scan(lanes);
    const int vs = incl - lanes;
    S.vlstart[lane] = vs;
    if (lane == kWave - 1) S.vlstart[kSegChunk] = incl;
    if (lanes > 0) {
        S.owner[vs] = lane + 1;
        for (int w = (vs >> 6) + 1; (w << 6) < incl; ++w) S.owner[w << 6] = lane + 1;  // each wave pass starts marked
    }
    if (lane == 0) {
        S.first_orf = c0;
        const long long rest = a1 - (c0 + kSegChunk);
        S.more = rest > 0 ? 1 : 0;
    }
}

// Lane runs of one chunk: every wave takes passes of 64 virtual lanes.
__device__ __forceinline__ void pipe_run(const SegSet &S, const int *s_counts, RunRec *s_rec,
                                         SegInts *s_ints, int wave, int lane)
{
    const int total_vl = S.vlstart[kSegChunk];
    for (int vbase = wave * kWave; vbase < total_vl; vbase += kTileBlock) {
        const int vl = vbase + lane;
        const bool active = vl < total_vl;
        const int seg = wave_max_scan(S.owner[vl]) - 1;  // >= 0: lane 0 of the pass is marked
        const int r = vl - S.vlstart[seg];
        int n_run = S.ntrip[seg] - r * kRun;
        n_run = n_run > kRun ? kRun : n_run;
        const int q0 = active ? S.qfirst[seg] + 3 * kRun * r : 0;
        const int rem0 = S.endq[seg] - q0;  // positions of the ORF from q0 on (clamped far end)
        int lim = rem0 - 2 < 3 * n_run ? rem0 - 2 : 3 * n_run;
        if (!active) lim = 0;

        LaneSums sv;
        lane_run(s_counts + q0, lim, sv);

        // partial last codon (L % 3 != 0): common.py:164-180 still sums it
        const int ip = (rem0 % 3 != 0) ? rem0 / 3 : -1;
        const bool has_partial = active && ip >= 0 && ip < n_run;
        if (__any(has_partial)) {
            if (has_partial) {
                unsigned codon = (unsigned)s_counts[q0 + 3 * ip];
                if (rem0 - 3 * ip == 2) codon += (unsigned)s_counts[q0 + 3 * ip + 1];
                sv.count += codon;
                sv.mn = min(sv.mn, codon);
            }
        }
        if (active) {  // integer sums: exact, order independent -> LDS atomics per segment
            SegInts &acc = s_ints[seg];
            atomicAdd(&acc.nn, sv.nn);
            atomicAdd(&acc.mm, sv.mm);
            atomicAdd(&acc.count, (unsigned long long)sv.count);
            atomicMin(&acc.min_codon, sv.mn);
        }
        // float sums: deterministic segmented scan inside each 16-lane row
        const int key = active ? seg + 1 : kSegChunk + 1;
        seg_scan_rows(sv, key);
        const int key_next = dpp_fetch<0x101 /* row_shl:1 */, 0xf>(0, key);  // 0 at the row's last lane
        if (active && key_next != key) {
            RunRec &rec = s_rec[seg + (vl >> 4)];
#pragma unroll
            for (int f = 0; f < 3; ++f) {
                rec.p[f] = sv.p[f];
                rec.q[f] = sv.q[f];
            }
        }
    }
}

// Finish one chunk by ONE wave: one lane per slot sums its row records in float64 and
// writes the segment record (k_orf_finish scores the ORFs afterwards).
__device__ __forceinline__ void pipe_finish(const SegSet &S, RunRec *s_rec, SegInts *s_ints, long long b,
                                            const TileWorkspace &ws, int lane)
{
    if (S.live[lane]) {
        const int vs = S.vlstart[lane];
        const int ve = S.vlstart[lane + 1];
        double p[3] = {0.0, 0.0, 0.0}, q[3] = {0.0, 0.0, 0.0};
        unsigned long long nn = 0, mm = 0, count = 0;
        unsigned min_codon = (unsigned)RP_MIN_CODON_COV_EMPTY;
        if (ve > vs) {
            const int w_first = vs >> 4;
            const int w_last = (ve - 1) >> 4;
            for (int w = w_first; w <= w_last; ++w) {
                const RunRec &rec = s_rec[lane + w];
#pragma unroll
                for (int f = 0; f < 3; ++f) {
                    p[f] += (double)rec.p[f];
                    q[f] += (double)rec.q[f];
                }
            }
            const SegInts &acc = s_ints[lane];
            nn = acc.nn;
            mm = acc.mm;
            count = acc.count;
            min_codon = acc.min_codon;
        }
        store_record(ws.rec, S.first_orf + lane + b, p, q, nn, mm, count, min_codon);
    }
    // the integer accumulators belong to the next chunk from here on
    s_ints[lane].nn = 0;
    s_ints[lane].mm = 0;
    s_ints[lane].count = 0;
    s_ints[lane].min_codon = (unsigned)RP_MIN_CODON_COV_EMPTY;
}

__global__ __launch_bounds__(kTileBlock, kPipeBlocksPerCu) void k_tile_score_pipe(
    const int32_t *__restrict__ counts, const int64_t *__restrict__ offsets, long long n_orfs, TilePlan plan,
    TileWorkspace ws)
{
    __shared__ __attribute__((aligned(16))) int s_counts[kPipeLdsCounts];
    __shared__ SegSet s_seg[2];
    __shared__ RunRec s_rec[kPipeMaxRecs];
    __shared__ SegInts s_ints[kSegChunk];

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = tid >> 6;
    const long long stride = gridDim.x;
    long long b = blockIdx.x;
    if (b >= plan.n_tiles) return;

    // ---- prologue: tile b in flight, its segment table; tile_first two tiles ahead ----------
    long long t0 = b * (long long)kPipeTile - plan.mis;
    long long t1 = t0 + kPipeTile < plan.total_nt ? t0 + kPipeTile : plan.total_nt;
    long long a0 = ws.tile_first[b];
    long long a1 = ws.tile_first[b + 1];
    const v4i *base = reinterpret_cast<const v4i *>(counts - plan.mis);  // 16-byte aligned (make_pipe_plan)
    const long long last_chunk = (plan.total_nt + plan.mis - 1) / 4;
    Prefetch pf;
    pipe_prefetch(pf, base, b * (long long)(kPipeTile / 4), last_chunk, tid);
    long long bn = b + stride;
    long long na0 = bn < plan.n_tiles ? ws.tile_first[bn] : 0;
    long long na1 = bn < plan.n_tiles ? ws.tile_first[bn + 1] : 0;
    long long bnn = bn + stride;
    long long nna0 = bnn < plan.n_tiles ? ws.tile_first[bnn] : 0;
    long long nna1 = bnn < plan.n_tiles ? ws.tile_first[bnn + 1] : 0;
    if (tid < kSegChunk) {
        s_ints[tid].nn = 0;
        s_ints[tid].mm = 0;
        s_ints[tid].count = 0;
        s_ints[tid].min_codon = (unsigned)RP_MIN_CODON_COV_EMPTY;
    }
    long long nbeg = 0, nend = 0;  // setup wave: bounds of the NEXT tile's slots, in flight
    if (wave == kPipeSetupWave) {
        pipe_load_bounds(offsets, a0, a1, lane, nbeg, nend);
        pipe_setup(s_seg[0], nbeg, nend, a0, a1, a0 - 1, t0, t1, lane);
        if (bn < plan.n_tiles) pipe_load_bounds(offsets, na0, na1, lane, nbeg, nend);
    }

    int cur = 0;
    for (;;) {
        // ---- commit tile b to LDS, put tile b + G in flight ---------------------------------
        pipe_commit(pf, s_counts, tid);
        const bool has_next = bn < plan.n_tiles;
        long long nt0 = 0, nt1 = 0;
        if (has_next) {
            nt0 = bn * (long long)kPipeTile - plan.mis;
            nt1 = nt0 + kPipeTile < plan.total_nt ? nt0 + kPipeTile : plan.total_nt;
            pipe_prefetch(pf, base, bn * (long long)(kPipeTile / 4), last_chunk, tid);
        }
        lds_barrier();

        // ---- lane runs of tile b (first chunk) ----------------------------------------------
#if !defined(RP_PIPE_EXP) || RP_PIPE_EXP >= 2
        pipe_run(s_seg[cur], s_counts, s_rec, s_ints, wave, lane);
#endif
        lds_barrier();  // records + integer sums complete

        // ---- rare: more than kSegChunk segments start in this tile -> sequential chunks.  Their
        //      offsets are plain global loads, which wait for the prefetch ahead of them.
        if (s_seg[cur].more) {
            long long c0 = s_seg[cur].first_orf + kSegChunk;
            if (wave == 0) pipe_finish(s_seg[cur], s_rec, s_ints, b, ws, lane);
            lds_barrier();
            for (;;) {
                if (wave == 0) {
                    const long long orf = c0 + lane;
                    long long beg = 0, end = 0;
                    if (orf < a1) {
                        beg = offsets[orf];
                        end = offsets[orf + 1];
                    }
                    pipe_setup(s_seg[cur], beg, end, a0, a1, c0, t0, t1, lane);
                }
                lds_barrier();
                pipe_run(s_seg[cur], s_counts, s_rec, s_ints, wave, lane);
                lds_barrier();
                const int more = s_seg[cur].more;
                if (!more) break;  // the last chunk is finished by the common code below
                if (wave == 0) pipe_finish(s_seg[cur], s_rec, s_ints, b, ws, lane);
                c0 += kSegChunk;
                lds_barrier();
            }
        }

        // ---- finish(b) | segment table of tile b + G, bounds of tile b + 2G ---------------------
#if !defined(RP_PIPE_EXP) || RP_PIPE_EXP >= 3
        if (wave == 0) pipe_finish(s_seg[cur], s_rec, s_ints, b, ws, lane);
#else
        if (tid == 0 && s_counts[tid] == 0x7fffffff) ws.rec.nn[b] = 1;  // keep the tile live
#endif
        if (wave == kPipeSetupWave && has_next) {
            pipe_setup(s_seg[cur ^ 1], nbeg, nend, na0, na1, na0 - 1, nt0, nt1, lane);
            if (bnn < plan.n_tiles) pipe_load_bounds(offsets, nna0, nna1, lane, nbeg, nend);
        }
        if (!has_next) break;
        // roll the pipeline registers
        b = bn;
        t0 = nt0;
        t1 = nt1;
        a0 = na0;
        a1 = na1;
        bn = bnn;
        na0 = nna0;
        na1 = nna1;
        bnn = bn + stride;
        nna0 = bnn < plan.n_tiles ? ws.tile_first[bnn] : 0;
        nna1 = bnn < plan.n_tiles ? ws.tile_first[bnn + 1] : 0;
        cur ^= 1;
        lds_barrier();  // the tile buffer is free, the next table is built
    }
}

}  // namespace rp
