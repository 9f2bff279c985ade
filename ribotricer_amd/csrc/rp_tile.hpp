// rp_tile.hpp -- flat-tile phase scoring: the throughput path (RP_ALGO_TILE).
//
// The CSR counts array is cut into fixed tiles of kTile positions on the 16-byte
// aligned address grid, one workgroup per tile, so every workgroup streams the same
// number of bytes with aligned dwordx4 loads regardless of how ragged the ORFs are
// (20 short ORFs or a slice of one 100 k-nt ORF cost the same).  Inside a tile:
//
//   1. the tile (+ a 4-dword halo) is staged in LDS, each count read from HBM once;
//   2. wave 0 lists the tile's segments -- the ORF that straddles in from the left
//      ("head") and the ORFs that start inside -- and gives each segment
//      ceil(T/kRun) lanes, T = codon triplets whose first position lies in the tile;
//   3. every lane walks a contiguous run of <= kRun triplets of ONE segment out of
//      LDS (odd dword stride between lanes -> bank-conflict free), one codon of each
//      reading frame per step, fp32 unit vectors (one v_rsq_f32 per codon);
//   4. a segmented wave scan (lanes of a segment are consecutive) folds lane
//      partials into one record per (segment, wave);
//   5. one thread per segment sums its records in float64 and either finishes the
//      ORF (frame scores -> state machine -> status -> store) or, when the ORF
//      straddles a tile boundary, writes a partial record for k_tile_finalize;
//   6. ORFs whose fp32 frame decision is too close to call are re-walked in float64
//      by a whole wave (rp_wave.hpp).
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
constexpr int kTile = 6144;     // positions per tile (24 KiB of int32)
constexpr int kRun = 9;         // triplets per lane run; odd => lane stride 27 dwords, conflict free
constexpr int kSegChunk = 64;   // segments set up per round (one per lane of wave 0)
constexpr int kHalo = 8;        // dwords staged past the tile end (4 needed, 2 chunks loaded)
constexpr int kLdsCounts = kTile + kHalo + 3 * kRun + 8;
constexpr int kLoadRounds = (kTile / 4 + kTileBlock - 1) / kTileBlock;  // dwordx4 chunks per thread
constexpr int kMaxRecs = kSegChunk + 8;

enum SegKind : int { kSegComplete = 0, kSegHead = 1, kSegTail = 2 };

struct TilePlan {
    long long n_tiles;
    long long total_nt;
    long long n_orfs;
    int mis;  // (counts address / 4) % 4: tiles live on the 16-byte aligned grid
};

// Partial sums of the part of an ORF that one tile owns.
struct alignas(16) TilePartial {
    double p[3];
    double q[3];
    int n[3];
    int m[3];
    long long count;
    int min_codon;
    int pad;
};

struct TileWorkspace {
    long long *tile_first;  // [n_tiles + 1] first ORF starting at/after each tile start
    TilePartial *partials;  // [n_tiles][2]: slot 0 head segment, slot 1 tail segment
};

inline long long max_tiles(long long total_nt) { return (total_nt + 3 + kTile - 1) / kTile + 1; }

inline TilePlan make_tile_plan(long long n_orfs, long long total_nt, const void *counts = nullptr)
{
    TilePlan p;
    p.n_orfs = n_orfs;
    p.total_nt = total_nt;
    p.mis = (int)((reinterpret_cast<uintptr_t>(counts) >> 2) & 3u);
    p.n_tiles = (total_nt + p.mis + kTile - 1) / kTile;
    if (p.n_tiles < 1) p.n_tiles = 1;
    return p;
}

inline size_t workspace_bytes(long long total_nt)
{
    const size_t nt = (size_t)max_tiles(total_nt);
    size_t b = (nt + 1) * sizeof(long long);
    b = (b + 127) & ~(size_t)127;
    b += nt * 2 * sizeof(TilePartial);
    return b;
}

inline TileWorkspace carve_workspace(void *base, long long total_nt)
{
    const size_t nt = (size_t)max_tiles(total_nt);
    size_t b = (nt + 1) * sizeof(long long);
    b = (b + 127) & ~(size_t)127;
    TileWorkspace ws;
    ws.tile_first = reinterpret_cast<long long *>(base);
    ws.partials = reinterpret_cast<TilePartial *>(reinterpret_cast<char *>(base) + b);
    return ws;
}

// ---------------------------------------------------------------------------
// pass 1: tile_first[b] = lower_bound(offsets[0..n], start position of tile b)
// ---------------------------------------------------------------------------
__global__ void k_tile_index(const int64_t *__restrict__ offsets, long long n_orfs, TilePlan plan,
                             TileWorkspace ws)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n_orfs) return;
    if (i == 0) ws.tile_first[0] = 0;
    if (i == n_orfs) ws.tile_first[plan.n_tiles] = n_orfs;
    const long long o = offsets[i];
    const long long o_prev = i > 0 ? (long long)offsets[i - 1] : -1 - (long long)plan.mis;
    // tiles b >= 1 whose start position b*kTile - mis lies in (o_prev, o]
    long long b_lo = (o_prev + plan.mis) / kTile + 1;
    long long b_hi = (o + plan.mis) / kTile;
    if (b_lo < 1) b_lo = 1;
    if (b_hi > plan.n_tiles - 1) b_hi = plan.n_tiles - 1;
    for (long long b = b_lo; b <= b_hi; ++b) ws.tile_first[b] = i;
}

// ---------------------------------------------------------------------------
// pass 2: the scoring kernel
// ---------------------------------------------------------------------------
struct RunRec {  // what one wave contributes to one segment
    float p[3];
    float q[3];
    unsigned nn;  // n[0] | n[1] << 10 | n[2] << 20
    unsigned mm;
    unsigned long long count;
    int min_codon;
    int pad;
};

__device__ __forceinline__ void load_tile_to_lds(const int32_t *__restrict__ counts, long long t0,
                                                 long long total_nt, int *s_counts, int tid)
{
    // chunk c covers LDS dwords [4c, 4c+4) = positions t0 + 4c ..; (counts + t0) is 16-byte aligned
    constexpr int n_chunks = (kTile + kHalo) / 4;
    int4 regs[kLoadRounds + 1];
#pragma unroll
    for (int k = 0; k <= kLoadRounds; ++k) {
        const int c = tid + k * kTileBlock;
        const long long pos = t0 + 4LL * c;
        int4 v = make_int4(0, 0, 0, 0);
        if (c < n_chunks) {
            if (pos >= 0 && pos + 4 <= total_nt) {
                v = *reinterpret_cast<const int4 *>(counts + pos);
            } else {
                if (pos + 0 >= 0 && pos + 0 < total_nt) v.x = counts[pos + 0];
                if (pos + 1 >= 0 && pos + 1 < total_nt) v.y = counts[pos + 1];
                if (pos + 2 >= 0 && pos + 2 < total_nt) v.z = counts[pos + 2];
                if (pos + 3 >= 0 && pos + 3 < total_nt) v.w = counts[pos + 3];
            }
        }
        regs[k] = v;
    }
#pragma unroll
    for (int k = 0; k <= kLoadRounds; ++k) {
        const int c = tid + k * kTileBlock;
        if (c < n_chunks) *reinterpret_cast<int4 *>(s_counts + 4 * c) = regs[k];
    }
}

// Hillis-Steele segmented inclusive scan over the wave; keys are non-decreasing in lane.
template <typename T, typename Op>
__device__ __forceinline__ T seg_scan_step(T x, int d, bool take, Op op)
{
    const T up = __shfl_up(x, d, kWave);
    return take ? op(x, up) : x;
}

__global__ __launch_bounds__(kTileBlock) void k_tile_score(const int32_t *__restrict__ counts,
                                                           const int64_t *__restrict__ offsets,
                                                           long long n_orfs, TilePlan plan,
                                                           TileWorkspace ws, OrfOutputs out,
                                                           FilterParams fp)
{
    __shared__ __attribute__((aligned(16))) int s_counts[kLdsCounts];
    __shared__ int s_qfirst[kSegChunk];   // LDS index of the first owned triplet
    __shared__ int s_endq[kSegChunk];     // ORF end in LDS coordinates (clamped)
    __shared__ int s_ntrip[kSegChunk];    // owned triplets
    __shared__ int s_kind[kSegChunk];
    __shared__ int s_vlstart[kSegChunk + 1];
    __shared__ RunRec s_rec[kMaxRecs];
    __shared__ int s_recheck[kSegChunk];
    __shared__ int s_n_recheck;

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = tid >> 6;
    const long long b = blockIdx.x;
    const long long t0 = b * (long long)kTile - plan.mis;  // position of LDS index 0 (may be < 0 for b == 0)
    long long t1 = t0 + kTile;
    if (t1 > plan.total_nt) t1 = plan.total_nt;

    load_tile_to_lds(counts, t0, plan.total_nt, s_counts, tid);

    const long long a0 = ws.tile_first[b];
    const long long a1 = ws.tile_first[b + 1];
    // the ORF that straddles in from the left, if any (offsets[a0] is the first start >= t0)
    const int has_head = (a0 > 0 && (long long)offsets[a0] > t0) ? 1 : 0;
    const long long n_seg_total = has_head + (a1 - a0);
    if (tid == 0) s_n_recheck = 0;
    __syncthreads();

    for (long long chunk = 0; chunk < n_seg_total; chunk += kSegChunk) {
        const int nseg = (int)((n_seg_total - chunk) < kSegChunk ? (n_seg_total - chunk) : kSegChunk);

        // ---- segment setup + lane allocation (wave 0) ---------------------------------
        if (wave == 0) {
            int lanes = 0;
            if (lane < nseg) {
                const long long s = chunk + lane;
                const long long orf = a0 - has_head + s;
                const long long beg = offsets[orf];
                const long long end = offsets[orf + 1];
                const long long len = end - beg;
                const long long rel0 = t0 - beg;  // > 0 only for the head segment
                const long long jlo = rel0 > 0 ? (rel0 + 2) / 3 : 0;
                const long long ntrip_all = (len + 2) / 3;
                long long jhi = (t1 - beg + 2) / 3;  // triplets whose first position is < t1
                if (jhi > ntrip_all) jhi = ntrip_all;
                const long long ntrip = jhi > jlo ? jhi - jlo : 0;
                long long endq = end - t0;
                if (endq > kTile + kHalo) endq = kTile + kHalo;
                s_qfirst[lane] = (int)(beg + 3 * jlo - t0);
                s_endq[lane] = (int)endq;
                s_ntrip[lane] = (int)ntrip;
                s_kind[lane] = (has_head && s == 0) ? kSegHead : (jhi == ntrip_all ? kSegComplete : kSegTail);
                lanes = (int)((ntrip + kRun - 1) / kRun);
            }
            // exclusive prefix sum of lanes over the wave
            int incl = lanes;
#pragma unroll
            for (int d = 1; d < kWave; d <<= 1) {
                const int up = __shfl_up(incl, d, kWave);
                if (lane >= d) incl += up;
            }
            s_vlstart[lane] = incl - lanes;
            if (lane == kWave - 1) s_vlstart[kSegChunk] = incl;
        }
        __syncthreads();
        const int total_vl = s_vlstart[kSegChunk];

        // ---- lane runs + segmented wave reduction ---------------------------------------
        for (int vbase = wave * kWave; vbase < total_vl; vbase += kTileBlock) {
            const int vl = vbase + lane;
            const bool active = vl < total_vl;
            // largest s with vlstart[s] <= vl  (zero-lane segments are skipped automatically)
            int lo = 0, hi = nseg;
#pragma unroll
            for (int it = 0; it < 7; ++it) {  // range <= kSegChunk = 2^6: at most 7 halvings
                if (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (s_vlstart[mid] <= vl)
                        lo = mid + 1;
                    else
                        hi = mid;
                }
            }
            int seg = lo - 1;
            if (!active || seg < 0) seg = 0;
            const int r = vl - s_vlstart[seg];
            int n_run = s_ntrip[seg] - r * kRun;
            n_run = n_run > kRun ? kRun : n_run;
            if (!active) n_run = 0;
            const int endq = s_endq[seg];
            int q = s_qfirst[seg] + 3 * kRun * r;
            if (!active) q = 0;

            FrameAcc<float> acc[3];
            acc_clear(acc);
            unsigned long long cnt = 0;
            int mn = RP_MIN_CODON_COV_EMPTY;
            int v0 = s_counts[q];
            int v1 = s_counts[q + 1];
#pragma unroll
            for (int i = 0; i < kRun; ++i) {
                const int v2 = s_counts[q + 2];
                const int v3 = s_counts[q + 3];
                const int v4 = s_counts[q + 4];
                const int rem = (i < n_run) ? endq - q : 0;  // positions left in the ORF from q
                const int codon = (rem > 0 ? v0 : 0) + (rem > 1 ? v1 : 0) + (rem > 2 ? v2 : 0);
                cnt += (unsigned)codon;
                mn = rem > 0 ? min(mn, codon) : mn;
                codon_add(acc[0], v0, v1, v2, rem > 2);
                codon_add(acc[1], v1, v2, v3, rem > 3);
                codon_add(acc[2], v2, v3, v4, rem > 4);
                v0 = v3;
                v1 = v4;
                q += 3;
            }

            // segmented inclusive scan keyed by segment (inactive lanes share a key past the end)
            const int key = active ? seg : kSegChunk;
            unsigned nn = (unsigned)acc[0].n | ((unsigned)acc[1].n << 10) | ((unsigned)acc[2].n << 20);
            unsigned mm = (unsigned)acc[0].m | ((unsigned)acc[1].m << 10) | ((unsigned)acc[2].m << 20);
            float p0 = acc[0].p, p1 = acc[1].p, p2 = acc[2].p;
            float q0 = acc[0].q, q1 = acc[1].q, q2 = acc[2].q;
#pragma unroll
            for (int d = 1; d < kWave; d <<= 1) {
                const int kup = __shfl_up(key, d, kWave);
                const bool take = (lane >= d) && (kup == key);
                auto addf = [](float a, float c) { return a + c; };
                auto addu = [](unsigned a, unsigned c) { return a + c; };
                auto addl = [](unsigned long long a, unsigned long long c) { return a + c; };
                auto mini = [](int a, int c) { return a < c ? a : c; };
                p0 = seg_scan_step(p0, d, take, addf);
                p1 = seg_scan_step(p1, d, take, addf);
                p2 = seg_scan_step(p2, d, take, addf);
                q0 = seg_scan_step(q0, d, take, addf);
                q1 = seg_scan_step(q1, d, take, addf);
                q2 = seg_scan_step(q2, d, take, addf);
                nn = seg_scan_step(nn, d, take, addu);
                mm = seg_scan_step(mm, d, take, addu);
                cnt = seg_scan_step(cnt, d, take, addl);
                mn = seg_scan_step(mn, d, take, mini);
            }
            const int key_next = __shfl_down(key, 1, kWave);
            const bool run_end = active && (lane == kWave - 1 || key_next != key);
            if (run_end) {
                RunRec &rec = s_rec[seg + (vbase >> 6)];
                rec.p[0] = p0;
                rec.p[1] = p1;
                rec.p[2] = p2;
                rec.q[0] = q0;
                rec.q[1] = q1;
                rec.q[2] = q2;
                rec.nn = nn;
                rec.mm = mm;
                rec.count = cnt;
                rec.min_codon = mn;
            }
        }
        __syncthreads();

        // ---- one thread per segment: float64 combine, finish or emit a partial ------------
        if (tid < nseg) {
            const int seg = tid;
            const long long orf = a0 - has_head + chunk + seg;
            const int vs = s_vlstart[seg];
            const int ve = s_vlstart[seg + 1];
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
                const int w_first = vs >> 6;
                const int w_last = (ve - 1) >> 6;
                for (int w = w_first; w <= w_last; ++w) {
                    const RunRec &rec = s_rec[seg + w];
#pragma unroll
                    for (int f = 0; f < 3; ++f) {
                        t.p[f] += (double)rec.p[f];
                        t.q[f] += (double)rec.q[f];
                        t.n[f] += (int)((rec.nn >> (10 * f)) & 1023u);
                        t.m[f] += (int)((rec.mm >> (10 * f)) & 1023u);
                    }
                    t.count += (long long)rec.count;
                    t.min_codon = min(t.min_codon, rec.min_codon);
                }
            }
            const int kind = s_kind[seg];
            if (kind == kSegComplete) {
                FrameScore fr[3];
#pragma unroll
                for (int f = 0; f < 3; ++f) fr[f] = frame_score(t.p[f], t.q[f], t.n[f], t.m[f]);
                if (fp32_decision_unsafe(fr)) {
                    const int slot = atomicAdd(&s_n_recheck, 1);
                    s_recheck[slot] = seg;
                } else {
                    double phase;
                    int valid;
                    unsigned flags;
                    combine_frames(fr, phase, valid, flags);
                    const long long len = (long long)offsets[orf + 1] - (long long)offsets[orf];
                    store_orf(out, fp, orf, phase, valid, t.count, t.min_codon, flags, len);
                }
            } else {
                ws.partials[2 * b + (kind == kSegHead ? 0 : 1)] = t;
            }
        }
        __syncthreads();

        // ---- float64 re-walk of the too-close-to-call ORFs, one wave each -----------------
        const int n_re = s_n_recheck;
        for (int k = wave; k < n_re; k += kTileBlock / kWave) {
            const long long orf = a0 - has_head + chunk + s_recheck[k];
            const long long beg = offsets[orf];
            const long long len = (long long)offsets[orf + 1] - beg;
            WalkResult<double> w;
            wave_walk<double>(counts + beg, len, lane, w);
            FrameScore fr[3];
            long long count;
            int min_codon;
            wave_reduce_frames(w, fr, count, min_codon);
            double phase;
            int valid;
            unsigned flags;
            combine_frames(fr, phase, valid, flags);
            if (lane == 0)
                store_orf(out, fp, orf, phase, valid, count, min_codon, flags | RP_FLAG_RECHECK64, len);
        }
        __syncthreads();
        if (tid == 0) s_n_recheck = 0;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// pass 3: ORFs that straddle a tile boundary -- one wave per tile whose last ORF
// does not end inside it; sums the tail partial of that tile and the head partials
// of the tiles the ORF runs through.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kTileBlock) void k_tile_finalize(const int32_t *__restrict__ counts,
                                                              const int64_t *__restrict__ offsets,
                                                              long long n_orfs, TilePlan plan,
                                                              TileWorkspace ws, OrfOutputs out,
                                                              FilterParams fp)
{
    const int lane = threadIdx.x & (kWave - 1);
    const long long b = (long long)blockIdx.x * (kTileBlock / kWave) + (threadIdx.x >> 6);
    if (b >= plan.n_tiles) return;
    const long long a0 = ws.tile_first[b];
    const long long a1 = ws.tile_first[b + 1];
    if (a1 <= a0) return;  // no ORF starts in this tile
    const long long orf = a1 - 1;
    const long long beg = offsets[orf];
    const long long len = (long long)offsets[orf + 1] - beg;
    const long long ntrip_all = (len + 2) / 3;
    if (ntrip_all == 0) return;
    long long t1 = (b + 1) * (long long)kTile - plan.mis;
    if (t1 > plan.total_nt) t1 = plan.total_nt;
    const long long last_first = beg + 3 * (ntrip_all - 1);  // first position of the last triplet
    if (last_first < t1) return;                             // the ORF was finished inside its tile
    const long long b_end = (last_first + plan.mis) / kTile;

    double p[3] = {0, 0, 0}, q[3] = {0, 0, 0};
    int n[3] = {0, 0, 0}, m[3] = {0, 0, 0};
    long long count = 0;
    int min_codon = RP_MIN_CODON_COV_EMPTY;
    for (long long k = lane; k <= b_end - b; k += kWave) {
        const TilePartial &t = (k == 0) ? ws.partials[2 * b + 1] : ws.partials[2 * (b + k)];
#pragma unroll
        for (int f = 0; f < 3; ++f) {
            p[f] += t.p[f];
            q[f] += t.q[f];
            n[f] += t.n[f];
            m[f] += t.m[f];
        }
        count += t.count;
        min_codon = min(min_codon, t.min_codon);
    }
    FrameScore fr[3];
#pragma unroll
    for (int f = 0; f < 3; ++f)
        fr[f] = frame_score(wave_sum(p[f]), wave_sum(q[f]), wave_sum(n[f]), wave_sum(m[f]));
    count = wave_sum(count);
    min_codon = wave_min(min_codon);
    unsigned extra = RP_FLAG_SPLIT;
    if (fp32_decision_unsafe(fr)) {  // wave-uniform
        WalkResult<double> w;
        wave_walk<double>(counts + beg, len, lane, w);
        wave_reduce_frames(w, fr, count, min_codon);
        extra |= RP_FLAG_RECHECK64;
    }
    double phase;
    int valid;
    unsigned flags;
    combine_frames(fr, phase, valid, flags);
    if (lane == 0) store_orf(out, fp, orf, phase, valid, count, min_codon, flags | extra, len);
}

}  // namespace rp
