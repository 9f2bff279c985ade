// rp_tile_pipe.hpp -- persistent, software-pipelined form of the flat-tile scorer.
//
// Same ownership rules, arithmetic and reduction as rp_tile.hpp (whose device helpers
// it reuses), different schedule.  In k_tile_score a workgroup walks
//     DMA tile -> wait -> setup (wave 0) -> lane runs -> finish (wave 0)
// strictly in sequence, and the LDS tile buffers of the 4 resident workgroups already
// fill the CU's 160 KiB, so the only way to hide that chain behind other work is to
// shorten it.  Here a workgroup stays resident, walks tiles b, b+G, b+2G, ... and
// overlaps the three latency-bound stages of neighbouring tiles:
//
//     lane runs of tile i            (waves 0..3, VALU bound)
//     ------------------------------ barrier
//     finish of tile i   (wave 0)  | setup of tile i+1 (wave 1) | DMA of tile i+1
//     ------------------------------ wait DMA, barrier
//
// The tile buffer is single (it is free once the lane runs are done), segment tables are
// double buffered, tile_first is prefetched two tiles ahead, so the critical path per
// tile is  lane runs + max(finish, setup, DMA)  instead of their sum.
#pragma once

#include "rp_tile.hpp"

namespace rp {

constexpr int kPipeTile = 7168;  // positions per tile: 28 DMA rows, 3 lane-run passes of 64 x 15 triplets
constexpr int kPipeLdsCounts = kPipeTile + kHalo + 3 * kRun + 8;
constexpr int kPipeMaxVl = kPipeTile / (3 * kRun) + kSegChunk + 2 * kWave;
constexpr int kPipeMaxRecs = kSegChunk + kPipeMaxVl / 16 + 1;

struct SegSet {  // the segment table of one tile chunk
    long long len[kSegChunk];  // ORF length
    int qfirst[kSegChunk];     // LDS index of the first owned triplet
    int endq[kSegChunk];       // ORF end in LDS coordinates (clamped)
    int ntrip[kSegChunk];      // owned triplets
    int kind[kSegChunk];       // SegKind or kSegNone
    int vlstart[kSegChunk + 1];
    int owner[kPipeMaxVl];  // slot+1 at the first lane of a segment / wave pass, else 0
    long long first_orf;    // ORF index of slot 0
    int more;               // ORFs that start in this tile beyond this chunk
    int pad;
};

inline long long pipe_max_tiles(long long total_nt) { return (total_nt + 3 + kPipeTile - 1) / kPipeTile + 1; }

inline TilePlan make_pipe_plan(long long n_orfs, long long total_nt, const void *counts)
{
    TilePlan p;
    p.n_orfs = n_orfs;
    p.total_nt = total_nt;
    p.mis = (int)((reinterpret_cast<uintptr_t>(counts) >> 2) & 3u);
    p.n_tiles = (total_nt + p.mis + kPipeTile - 1) / kPipeTile;
    if (p.n_tiles < 1) p.n_tiles = 1;
    return p;
}

inline size_t pipe_workspace_bytes(long long total_nt)
{
    const size_t nt = (size_t)pipe_max_tiles(total_nt);
    size_t b = (nt + 1) * sizeof(long long);
    b = (b + 127) & ~(size_t)127;
    b += nt * 2 * sizeof(TilePartial);
    return b;
}

inline TileWorkspace pipe_carve_workspace(void *base, long long total_nt)
{
    const size_t nt = (size_t)pipe_max_tiles(total_nt);
    size_t b = (nt + 1) * sizeof(long long);
    b = (b + 127) & ~(size_t)127;
    TileWorkspace ws;
    ws.tile_first = reinterpret_cast<long long *>(base);
    ws.partials = reinterpret_cast<TilePartial *>(reinterpret_cast<char *>(base) + b);
    return ws;
}

// tile_first for the pipelined tile size (same rule as k_tile_index)
__global__ void k_pipe_index(const int64_t *__restrict__ offsets, long long n_orfs, TilePlan plan,
                             TileWorkspace ws)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n_orfs) return;
    if (i == 0) ws.tile_first[0] = 0;
    if (i == n_orfs) ws.tile_first[plan.n_tiles] = n_orfs;
    const long long o = offsets[i];
    const long long o_prev = i > 0 ? (long long)offsets[i - 1] : -1 - (long long)plan.mis;
    long long b_lo = (o_prev + plan.mis) / kPipeTile + 1;
    long long b_hi = (o + plan.mis) / kPipeTile;
    if (b_lo < 1) b_lo = 1;
    if (b_hi > plan.n_tiles - 1) b_hi = plan.n_tiles - 1;
    for (long long b = b_lo; b <= b_hi; ++b) ws.tile_first[b] = i;
}

// Issue (do not wait for) the load of tile [t0, t0 + kPipeTile + halo) into LDS.
// `first_wave`/`n_waves`: which waves of the workgroup issue the DMA rows (issuing stalls
// the wave until the memory pipe has taken the requests, so busy waves are left out).
__device__ __forceinline__ void pipe_issue_tile(const int32_t *__restrict__ counts, long long t0,
                                                long long total_nt, int *s_counts, int tid,
                                                int first_wave, int n_waves)
{
    constexpr int n_chunks = (kPipeTile + kHalo) / 4;
    constexpr int kRowPos = 256;
    static_assert(kPipeTile % kRowPos == 0, "tile must be a whole number of 1 KiB rows");
    const bool interior = (t0 >= 0) && (t0 + kPipeTile + kHalo <= total_nt);  // workgroup-uniform
    if (interior) {
        typedef const __attribute__((address_space(1))) void *gptr_t;
        typedef __attribute__((address_space(3))) void *lptr_t;
        const int lane = tid & (kWave - 1);
        const int32_t *src = counts + t0 + 4 * lane;
        const int my = (tid >> 6) - first_wave;  // 0 .. n_waves-1 for issuing waves
        if (my >= 0 && my < n_waves) {
#pragma unroll
            for (int row = 0; row < kPipeTile / kRowPos; ++row) {
                if (row % n_waves == my)
                    __builtin_amdgcn_global_load_lds((gptr_t)(src + row * kRowPos), (lptr_t)(s_counts + row * kRowPos), 16, 0, 0);
            }
        }
        if (tid >= first_wave * kWave && tid < first_wave * kWave + 2) {
            const int h = tid - first_wave * kWave;
            const int4 v = *reinterpret_cast<const int4 *>(counts + t0 + kPipeTile + 4 * h);
            *reinterpret_cast<int4 *>(s_counts + kPipeTile + 4 * h) = v;
        }
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
    }
}

// Segment table of one chunk of tile b, by ONE wave.  Slot L holds ORF c0 + L; for the
// first chunk c0 = a0 - 1, i.e. slot 0 is the ORF that straddles in from the left (or is
// unused).  Slots past the last ORF that starts in the tile are unused.
__device__ __forceinline__ void pipe_setup(SegSet &S, const int64_t *__restrict__ offsets,
                                           long long n_orfs, long long a0, long long a1, long long c0,
                                           long long t0, long long t1, int lane)
{
    for (int k = lane; k < kPipeMaxVl; k += kWave) S.owner[k] = 0;
    const long long orf = c0 + lane;
    int lanes = 0;
    int kind = kSegNone;
    if (orf >= 0 && orf < a1) {
        const long long beg = offsets[orf];
        const long long end = offsets[orf + 1];
        const bool head = orf < a0;  // only ORF a0 - 1 can be
        const bool live = !head || end > t0;
        if (live) {
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
            const bool end_in_lds = rem <= kPipeTile + kHalo;
            const int endq = end_in_lds ? (int)rem : kPipeTile + kHalo;
            const int lim_q = kt < endq ? kt : endq;  // owned triplets start below this
            const int ntrip = lim_q > qfirst ? (lim_q - qfirst + 2) / 3 : 0;
            const bool complete = !head && end_in_lds && ntrip == (endq - qfirst + 2) / 3;
            S.qfirst[lane] = qfirst;
            S.endq[lane] = endq;
            S.ntrip[lane] = ntrip;
            S.len[lane] = end - beg;
            kind = head ? kSegHead : (complete ? kSegComplete : kSegTail);
            lanes = (ntrip + kRun - 1) / kRun;
        }
    }
    S.kind[lane] = kind;
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
            const unsigned long long nn64 = (unsigned long long)sv.n[0] | ((unsigned long long)sv.n[1] << 21) |
                                            ((unsigned long long)sv.n[2] << 42);
            const unsigned long long mm64 = (unsigned long long)sv.m[0] | ((unsigned long long)sv.m[1] << 21) |
                                            ((unsigned long long)sv.m[2] << 42);
            atomicAdd(&acc.nn, nn64);
            atomicAdd(&acc.mm, mm64);
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

// Finish one chunk by ONE wave: one lane per slot sums its records in float64, scores or
// writes a partial; too-close-to-call ORFs are then re-walked in float64 by the same wave
// straight from global memory (the LDS tile may already be receiving the next tile).
__device__ __forceinline__ void pipe_finish(const SegSet &S, RunRec *s_rec, SegInts *s_ints,
                                            const int32_t *__restrict__ counts,
                                            const int64_t *__restrict__ offsets, long long b,
                                            TileWorkspace ws, const OrfOutputs &out,
                                            const FilterParams &fp, int lane)
{
    const int kind = S.kind[lane];
    const long long orf = S.first_orf + lane;
    bool unsafe = false;
    if (kind != kSegNone) {
        const int vs = S.vlstart[lane];
        const int ve = S.vlstart[lane + 1];
        TilePartial t;
#pragma unroll
        for (int f = 0; f < 3; ++f) {
            t.p[f] = 0.0;
            t.q[f] = 0.0;
            t.n[f] = 0;
            t.m[f] = 0;
        }
        t.count = 0;
        t.min_codon = RP_MIN_CODON_COV_EMPTY;
        t.pad = 0;
        if (ve > vs) {
            const int w_first = vs >> 4;
            const int w_last = (ve - 1) >> 4;
            for (int w = w_first; w <= w_last; ++w) {
                const RunRec &rec = s_rec[lane + w];
#pragma unroll
                for (int f = 0; f < 3; ++f) {
                    t.p[f] += (double)rec.p[f];
                    t.q[f] += (double)rec.q[f];
                }
            }
            const SegInts &acc = s_ints[lane];
#pragma unroll
            for (int f = 0; f < 3; ++f) {
                t.n[f] = (int)((acc.nn >> (21 * f)) & 0x1fffffu);
                t.m[f] = (int)((acc.mm >> (21 * f)) & 0x1fffffu);
            }
            t.count = (long long)acc.count;
            t.min_codon = (int)acc.min_codon;
        }
        if (kind == kSegComplete) {
            FrameScore fr[3];
#pragma unroll
            for (int f = 0; f < 3; ++f) fr[f] = frame_score(t.p[f], t.q[f], t.n[f], t.m[f]);
            unsafe = fp32_decision_unsafe(fr);
            if (!unsafe) {
                double phase;
                int valid;
                unsigned flags;
                combine_frames(fr, phase, valid, flags);
                store_orf(out, fp, orf, phase, valid, t.count, t.min_codon, flags, S.len[lane]);
            }
        } else {
            ws.partials[2 * b + (kind == kSegHead ? 0 : 1)] = t;
        }
    }
    // the integer accumulators belong to the next chunk from here on
    s_ints[lane].nn = 0;
    s_ints[lane].mm = 0;
    s_ints[lane].count = 0;
    s_ints[lane].min_codon = (unsigned)RP_MIN_CODON_COV_EMPTY;

    unsigned long long todo = __ballot(unsafe);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const long long orf_s = S.first_orf + src;
        const long long len_s = S.len[src];
        const long long beg_s = offsets[orf_s];
        WalkResult<double> w;
        wave_walk<double>(counts + beg_s, len_s, lane, w);
        FrameScore fr[3];
        long long count;
        int min_codon;
        wave_reduce_frames(w, fr, count, min_codon);
        double phase;
        int valid;
        unsigned flags;
        combine_frames(fr, phase, valid, flags);
        if (lane == 0)
            store_orf(out, fp, orf_s, phase, valid, count, min_codon, flags | RP_FLAG_RECHECK64, len_s);
    }
}

__global__ __launch_bounds__(kTileBlock, 4) void k_tile_score_pipe(const int32_t *__restrict__ counts,
                                                                const int64_t *__restrict__ offsets,
                                                                long long n_orfs, TilePlan plan,
                                                                TileWorkspace ws, OrfOutputs out,
                                                                FilterParams fp)
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

    // ---- prologue: tile b ------------------------------------------------------------------
    long long t0 = b * (long long)kPipeTile - plan.mis;
    long long t1 = t0 + kPipeTile < plan.total_nt ? t0 + kPipeTile : plan.total_nt;
    pipe_issue_tile(counts, t0, plan.total_nt, s_counts, tid, 0, 4);
    long long a0 = ws.tile_first[b];
    long long a1 = ws.tile_first[b + 1];
    // tile_first of the next tile, one iteration ahead of its use
    long long bn = b + stride;
    long long na0 = bn < plan.n_tiles ? ws.tile_first[bn] : 0;
    long long na1 = bn < plan.n_tiles ? ws.tile_first[bn + 1] : 0;
    if (tid < kSegChunk) {
        s_ints[tid].nn = 0;
        s_ints[tid].mm = 0;
        s_ints[tid].count = 0;
        s_ints[tid].min_codon = (unsigned)RP_MIN_CODON_COV_EMPTY;
    }
    if (wave == 1) pipe_setup(s_seg[0], offsets, n_orfs, a0, a1, a0 - 1, t0, t1, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // LDS-DMA completion is tracked by vmcnt only
    __syncthreads();

    int cur = 0;
    for (;;) {
        // ---- lane runs of tile b (first chunk) ----------------------------------------------
        pipe_run(s_seg[cur], s_counts, s_rec, s_ints, wave, lane);
        __syncthreads();  // records + integer sums complete; the tile is still needed if `more`

        // ---- rare: more than kSegChunk segments in this tile -> plain sequential chunks -------
        if (s_seg[cur].more) {
            long long c0 = s_seg[cur].first_orf + kSegChunk;
            if (wave == 0) pipe_finish(s_seg[cur], s_rec, s_ints, counts, offsets, b, ws, out, fp, lane);
            __syncthreads();
            for (;;) {
                if (wave == 0) pipe_setup(s_seg[cur], offsets, n_orfs, a0, a1, c0, t0, t1, lane);
                __syncthreads();
                pipe_run(s_seg[cur], s_counts, s_rec, s_ints, wave, lane);
                __syncthreads();
                const int more = s_seg[cur].more;
                if (!more) break;  // the last chunk is finished by the common code below
                if (wave == 0) pipe_finish(s_seg[cur], s_rec, s_ints, counts, offsets, b, ws, out, fp, lane);
                c0 += kSegChunk;
                __syncthreads();
            }
        }

        // ---- overlapped stage: finish(b) | setup(b + G) | DMA(b + G) --------------------------
        const bool has_next = bn < plan.n_tiles;
        long long nt0 = 0, nt1 = 0;
        if (has_next) {
            nt0 = bn * (long long)kPipeTile - plan.mis;
            nt1 = nt0 + kPipeTile < plan.total_nt ? nt0 + kPipeTile : plan.total_nt;
            pipe_issue_tile(counts, nt0, plan.total_nt, s_counts, tid, 2, 2);  // waves 2-3: 0 and 1 are busy
        }
        if (wave == 0) pipe_finish(s_seg[cur], s_rec, s_ints, counts, offsets, b, ws, out, fp, lane);
        if (wave == 1 && has_next) pipe_setup(s_seg[cur ^ 1], offsets, n_orfs, na0, na1, na0 - 1, nt0, nt1, lane);
        if (!has_next) break;
        // roll the pipeline registers
        b = bn;
        t0 = nt0;
        t1 = nt1;
        a0 = na0;
        a1 = na1;
        bn = b + stride;
        na0 = bn < plan.n_tiles ? ws.tile_first[bn] : 0;
        na1 = bn < plan.n_tiles ? ws.tile_first[bn + 1] : 0;
        cur ^= 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
}

}  // namespace rp
