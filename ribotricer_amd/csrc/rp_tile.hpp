// rp_tile.hpp -- flat-tile phase scoring: the throughput path (RP_ALGO_TILE).
//
// The CSR counts array is cut into fixed tiles of kTile positions on the 16-byte
// aligned address grid, one workgroup per tile, so every workgroup streams the same
// number of bytes regardless of how ragged the ORFs are (25 short ORFs or a slice of
// one 100 k-nt ORF cost the same).  Inside a tile:
//
//   1. waves 1-3 stage the tile (+ a 4-dword halo) in LDS with the LDS-DMA form of
//      global_load (each count leaves HBM once); meanwhile wave 0 reads the tile index
//      and the offsets of its 64 segment slots and builds the segment table: slot 0 is
//      the ORF that straddles in from the left ("head"), the others the ORFs that start
//      inside; each segment gets ceil(T/kRun) lanes, T = codon triplets whose first
//      position lies in the tile;
//   2. every lane walks a contiguous run of <= kRun triplets of ONE segment out of LDS
//      (odd dword stride between lanes -> bank-conflict free), one codon of each reading
//      frame per step, predicate-free fp32 arithmetic (one v_rsq_f32 per codon);
//   3. integer sums go to per-segment LDS atomics (exact, order independent); the six
//      float sums are folded by a segmented DPP scan inside each 16-lane row into one
//      record per (segment, row) -- deterministic, no float atomics;
//   4. one thread per segment sums its records in float64 and either finishes the ORF
//      (frame scores -> state machine -> status -> store) or, when the ORF crosses a
//      tile boundary, writes a partial record for k_tile_finalize;
//   5. ORFs whose fp32 frame decision is too close to call are re-walked in float64 out
//      of LDS by a whole wave (wave_walk, rp_wave.hpp).
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
constexpr int kTile = 7936;    // positions per tile (31 KiB of int32): 3 lane-run passes of 64 x 15 triplets
constexpr int kRun = 15;        // triplets per lane run; odd => lane stride 45 dwords, conflict free
constexpr int kSegChunk = 64;   // segments set up per round (one per lane of wave 0)
constexpr int kHalo = 8;        // dwords staged past the tile end (4 needed, 2 chunks loaded)
constexpr int kLdsCounts = kTile + kHalo + 3 * kRun + 8;  // runs may read (masked) past the halo
constexpr int kMaxRecs = kSegChunk + (kTile / (3 * kRun) + kSegChunk + 2 * 64) / 16 + 1;  // one per (segment, 16-lane row)

enum SegKind : int { kSegComplete = 0, kSegHead = 1, kSegTail = 2, kSegNone = 3 };

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
struct alignas(16) RunRec {  // float sums one 16-lane row contributes to one segment
    float p[3];
    float q[3];
    float pad[2];
};

// integer sums of one segment, accumulated with LDS atomics (order independent, exact)
struct SegInts {
    unsigned long long nn;     // n[0] | n[1] << 21 | n[2] << 42
    unsigned long long mm;     // same packing
    unsigned long long count;  // read count
    unsigned min_codon;
    unsigned pad;
};

// Stage positions [t0, t0 + kTile + kHalo) into LDS.  (counts + t0) is 16-byte aligned.
// Interior tiles use the LDS-DMA form of global_load (no VGPR round trip, one instruction
// per KiB row, rows dealt round-robin to the four waves); the first / last tile take the
// guarded path with zero fill.  The DMA is NOT waited for here.
__device__ __forceinline__ void load_tile_to_lds(const int32_t *__restrict__ counts, long long t0,
                                                 long long total_nt, int *s_counts, int tid)
{
    constexpr int n_chunks = (kTile + kHalo) / 4;  // 16-byte chunks
    constexpr int kRowPos = 256;                    // positions per DMA row (64 lanes x 16 B)
    static_assert(kTile % kRowPos == 0, "tile must be a whole number of 1 KiB rows");
    static_assert(kHalo == 8, "halo is loaded as two extra chunks");
    const bool interior = (t0 >= 0) && (t0 + kTile + kHalo <= total_nt);  // workgroup-uniform
    if (interior) {
        typedef const __attribute__((address_space(1))) void *gptr_t;
        typedef __attribute__((address_space(3))) void *lptr_t;
        const int lane = tid & (kWave - 1);
        const int32_t *src = counts + t0 + 4 * lane;
        // rows dealt round-robin to waves 1-3: issuing stalls a wave until the memory pipe has
        // taken its requests, and wave 0 builds the segment table meanwhile
        const int w = (tid >> 6) - 1;
        if (w >= 0) {
#pragma unroll
            for (int row = 0; row < kTile / kRowPos; ++row) {
                if (row % (kTileBlock / kWave - 1) == w)
                    __builtin_amdgcn_global_load_lds((gptr_t)(src + row * kRowPos), (lptr_t)(s_counts + row * kRowPos), 16, 0, 0);
            }
        }
        if (tid >= kWave && tid < kWave + 2) {  // halo: 2 chunks past the tile
            const int h = tid - kWave;
            const int4 v = *reinterpret_cast<const int4 *>(counts + t0 + kTile + 4 * h);
            *reinterpret_cast<int4 *>(s_counts + kTile + 4 * h) = v;
        }
        // NOTE: no wait here -- the caller waits (vmcnt only tracks the DMA) after it has
        // issued its own independent loads
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
    unsigned n[3], m[3];  // per lane: <= kRun each
    unsigned count;   // <= 3 * kRun * RP_MAX_COUNT < 2^32
    unsigned mn;
};

// One step of the segmented inclusive scan inside each 16-lane row: lanes whose DPP
// source carries the same segment key (keys are >= 1) fold the source's running value
// into their own.  Masking with all-ones / zero bits instead of a select keeps it at
// (fetch, and, add) per value.
template <int CTRL>
__device__ __forceinline__ void seg_scan_step(LaneSums &v, int key)
{
    const int bits = (dpp_row<CTRL>(key) == key) ? -1 : 0;
#pragma unroll
    for (int f = 0; f < 3; ++f) {
        v.p[f] += __int_as_float(dpp_row<CTRL>(__float_as_int(v.p[f])) & bits);
        v.q[f] += __int_as_float(dpp_row<CTRL>(__float_as_int(v.q[f])) & bits);
    }
}

__device__ __forceinline__ void seg_scan_rows(LaneSums &v, int key)
{
    seg_scan_step<kDppRowShr1>(v, key);
    seg_scan_step<kDppRowShr2>(v, key);
    seg_scan_step<kDppRowShr4>(v, key);
    seg_scan_step<kDppRowShr8>(v, key);
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

// The lane run: kRun triplets = 3*kRun codon starts read from 3*kRun + 2 consecutive LDS
// dwords.  `lim` = number of leading codon starts that are real codons of this lane's ORF
// and owned by this run.  Predicates are evaluated once per POSITION (is the count zero?
// does it equal its successor?) and combined per codon on the scalar unit.
constexpr int kRunBlock = 3;  // triplets per fully unrolled block of the lane run
static_assert(kRun % kRunBlock == 0, "kRun must be a multiple of kRunBlock");

// min(x, hi) for x >= 0 as v_med3_f32(x, 0, hi): a compiler-visible instruction (hipcc
// inserts no hazard wait states around inline asm, and v_rsq_f32 results need them) that
// does not drag in the NaN-canonicalising v_max(x, x) pair of fminf().
__device__ __forceinline__ float min_nonneg(float x, float hi) { return __builtin_amdgcn_fmed3f(x, 0.0f, hi); }

// clamp(x, 0, 1): folds into the producing instruction's clamp modifier
__device__ __forceinline__ float clamp01(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, 1.0f); }

// The lane run, predicate-free: every test is folded into fp32 arithmetic so the loop is
// 16 VALU instructions per codon with no v_cmp / v_cndmask / scalar mask traffic:
//   vF  = clamp(lim - c, 0, 1)          1 while codon c is a real codon of this run
//   r   = min(rsq(q), vF)               1/sqrt(q);  a == b == c gives rsq(0) = inf -> 1, and
//                                       its d0 = d1 = 0 make the products below exactly 0
//   P  += d0 r,  Q += d1 r              unit-vector sums
//   u   = min(q, vF)                    q is an integer: 0 or >= 1, so u is exactly 1 for a
//                                       counted codon, 0 otherwise                  -> M
//   E  += (vF - u) * min(a, 1)          flat non-zero codons; N = M + E
// M and E are sums of <= kRun exact 0/1 values per frame.
// Counts are converted to fp32 first, exact below 2^24 (RP_MAX_COUNT).
__device__ __forceinline__ void lane_run(const int *__restrict__ s, int lim, LaneSums &o)
{
    int v0 = s[0], v1 = s[1];
    float f0 = (float)v0, f1 = (float)v1;
    float df0 = f0 - f1;
    float limf = (float)lim;
    float P[3] = {0.f, 0.f, 0.f}, Q[3] = {0.f, 0.f, 0.f};
    float M[3] = {0.f, 0.f, 0.f}, E[3] = {0.f, 0.f, 0.f};
    unsigned cnt = 0, mn = (unsigned)RP_MIN_CODON_COV_EMPTY;
#pragma unroll 1
    for (int blk = 0; blk < kRun / kRunBlock; ++blk) {
#pragma unroll
        for (int c = 0; c < 3 * kRunBlock; ++c) {
            const int f = c % 3;
            const int v2 = s[c + 2];
            const float f2 = (float)v2;
            const float df1 = f1 - f2;
            const float qq = __builtin_fmaf(df0, df0 + df1, df1 * df1);
            const float vF = clamp01(limf - (float)c);
            const float r = min_nonneg(__builtin_amdgcn_rsqf(qq), vF);
            P[f] = __builtin_fmaf(df0, r, P[f]);
            Q[f] = __builtin_fmaf(df1, r, Q[f]);
            const float u = min_nonneg(qq, vF);  // q is 0 or >= 1: exactly 1 for a counted codon
            M[f] += u;
            E[f] = __builtin_fmaf(vF - u, clamp01(f0), E[f]);
            if (f == 0) {
                const bool valid = c < lim;
                const unsigned codon = (unsigned)(v0 + v1 + v2);
                cnt += valid ? codon : 0u;
                mn = min(mn, valid ? codon : (unsigned)RP_MIN_CODON_COV_EMPTY);
            }
            v0 = v1;
            v1 = v2;
            f0 = f1;
            f1 = f2;
            df0 = df1;
        }
        s += 3 * kRunBlock;
        lim -= 3 * kRunBlock;
        limf -= (float)(3 * kRunBlock);
    }
#pragma unroll
    for (int f = 0; f < 3; ++f) {
        o.p[f] = P[f];
        o.q[f] = Q[f];
        o.m[f] = (unsigned)(int)M[f];            // exact small integers
        o.n[f] = o.m[f] + (unsigned)(int)E[f];
    }
    o.count = cnt;
    o.mn = mn;
}

constexpr int kMaxVl = kTile / (3 * kRun) + kSegChunk + 2 * kWave;  // virtual lanes per chunk (upper bound)

__global__ __launch_bounds__(kTileBlock, 4) void k_tile_score(const int32_t *__restrict__ counts,
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
    __shared__ long long s_len[kSegChunk];  // ORF length (for n_codons and the float64 re-walk)
    __shared__ int s_vlstart[kSegChunk + 1];
    __shared__ int s_owner[kMaxVl];       // segment+1 at the first lane of a segment / wave, else 0
    __shared__ RunRec s_rec[kMaxRecs];
    __shared__ SegInts s_ints[kSegChunk];
    __shared__ int s_recheck[kSegChunk];
    __shared__ int s_n_recheck;

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = tid >> 6;
    const long long b = blockIdx.x;
    const long long t0 = b * (long long)kTile - plan.mis;  // position of LDS index 0 (may be < 0 for b == 0)
    long long t1 = t0 + kTile;
    if (t1 > plan.total_nt) t1 = plan.total_nt;

    // Issue the tile DMA, then everything that does not depend on it -- the tile index, the
    // scratch clear and (wave 0) the offsets of the first 64 segment slots -- then wait once.
    // Slot L of a chunk holds ORF c0 + L; the first chunk starts at c0 = a0 - 1, so slot 0 is
    // the ORF that straddles in from the left, when there is one.
    load_tile_to_lds(counts, t0, plan.total_nt, s_counts, tid);
    const long long a0 = ws.tile_first[b];
    const long long a1 = ws.tile_first[b + 1];
    long long beg0 = 0, end0 = 0;
    if (wave == 0) {
        // wave 0 owns the segment scratch until the first barrier: it alone clears it (the
        // other waves are still busy issuing DMA rows and must not clobber its marks later)
        for (int k = lane; k < kMaxVl; k += kWave) s_owner[k] = 0;
        s_ints[lane].nn = 0;
        s_ints[lane].mm = 0;
        s_ints[lane].count = 0;
        s_ints[lane].min_codon = (unsigned)RP_MIN_CODON_COV_EMPTY;
        if (lane == 0) s_n_recheck = 0;
        const long long orf = a0 - 1 + lane;
        if (orf >= 0 && orf < a1) {
            beg0 = offsets[orf];
            end0 = offsets[orf + 1];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // LDS-DMA completion is tracked by vmcnt only

    for (long long c0 = a0 - 1; c0 < a1; c0 += kSegChunk) {
        const bool first_chunk = c0 == a0 - 1;
        if (!first_chunk) {  // (the first chunk's scratch was cleared while the tile streamed in)
            for (int k = tid; k < kMaxVl; k += kTileBlock) s_owner[k] = 0;
            if (tid < kSegChunk) {
                s_ints[tid].nn = 0;
                s_ints[tid].mm = 0;
                s_ints[tid].count = 0;
                s_ints[tid].min_codon = (unsigned)RP_MIN_CODON_COV_EMPTY;
            }
            __syncthreads();  // previous chunk's readers are done, scratch is clear
        }

        // ---- segment setup + lane allocation (wave 0; needs only offsets, not the tile) ----
        if (wave == 0) {
            int lanes = 0;
            int kind = kSegNone;
            const long long orf = c0 + lane;
            if (orf >= 0 && orf < a1) {
                const long long beg = first_chunk ? beg0 : (long long)offsets[orf];
                const long long end = first_chunk ? end0 : (long long)offsets[orf + 1];
                const bool head = orf < a0;  // only ORF a0 - 1 can be; it counts if it reaches the tile
                if (!head || end > t0) {
                    const int kt = (int)(t1 - t0);  // tile length in positions (<= kTile)
                    int qfirst;                     // LDS index of the first triplet start >= tile start
                    if (head) {
                        const unsigned long long rel0 = (unsigned long long)(t0 - beg);  // > 0
                        // 2^32 == 1 (mod 3)
                        const unsigned m3 = ((unsigned)(rel0 >> 32) % 3u + (unsigned)(rel0 & 0xffffffffu) % 3u) % 3u;
                        qfirst = m3 == 0 ? 0 : 3 - (int)m3;
                    } else {
                        qfirst = (int)(beg - t0);
                    }
                    const long long rem = end - t0;  // >= 0
                    const bool end_in_lds = rem <= kTile + kHalo;
                    const int endq = end_in_lds ? (int)rem : kTile + kHalo;
                    const int lim_q = kt < endq ? kt : endq;  // owned triplets start below this
                    const int ntrip = lim_q > qfirst ? (lim_q - qfirst + 2) / 3 : 0;
                    const bool complete = !head && end_in_lds && ntrip == (endq - qfirst + 2) / 3;
                    s_qfirst[lane] = qfirst;
                    s_endq[lane] = endq;
                    s_ntrip[lane] = ntrip;
                    s_len[lane] = end - beg;
                    kind = head ? kSegHead : (complete ? kSegComplete : kSegTail);
                    lanes = (ntrip + kRun - 1) / kRun;
                }
            }
            s_kind[lane] = kind;
            const int incl = wave_add_scan(lanes);
            const int vs = incl - lanes;
            s_vlstart[lane] = vs;
            if (lane == kWave - 1) s_vlstart[kSegChunk] = incl;
            if (lanes > 0) {
                s_owner[vs] = lane + 1;
                // every wave-pass must find its segment at its first lane
                for (int w = (vs >> 6) + 1; (w << 6) < incl; ++w) s_owner[w << 6] = lane + 1;
            }
        }
        __syncthreads();
        const int total_vl = s_vlstart[kSegChunk];

        // ---- lane runs + segmented wave reduction ---------------------------------------
        for (int vbase = wave * kWave; vbase < total_vl; vbase += kTileBlock) {
            const int vl = vbase + lane;
            const bool active = vl < total_vl;
            const int seg = wave_max_scan(s_owner[vl]) - 1;  // >= 0: lane 0 of the pass is marked
            const int r = vl - s_vlstart[seg];
            int n_run = s_ntrip[seg] - r * kRun;
            n_run = n_run > kRun ? kRun : n_run;
            const int q0 = active ? s_qfirst[seg] + 3 * kRun * r : 0;
            const int rem0 = s_endq[seg] - q0;  // positions of the ORF from q0 on (clamped far end)
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
            // integer sums: exact and order independent -> LDS atomics straight per segment
            if (active) {
                SegInts &acc = s_ints[seg];
                // three 21-bit fields per 64-bit word: n[0] | n[1] << 21 | n[2] << 42
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
            const bool run_end = active && (key_next != key);
            if (run_end) {
                RunRec &rec = s_rec[seg + (vl >> 4)];
#pragma unroll
                for (int f = 0; f < 3; ++f) {
                    rec.p[f] = sv.p[f];
                    rec.q[f] = sv.q[f];
                }
            }
        }
        __syncthreads();

        // ---- one thread per segment: float64 combine, finish or emit a partial ------------
        if (tid < kSegChunk && s_kind[tid] != kSegNone) {
            const int seg = tid;
            const long long orf = c0 + seg;
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
                const int w_first = vs >> 4;
                const int w_last = (ve - 1) >> 4;
                for (int w = w_first; w <= w_last; ++w) {
                    const RunRec &rec = s_rec[seg + w];
#pragma unroll
                    for (int f = 0; f < 3; ++f) {
                        t.p[f] += (double)rec.p[f];
                        t.q[f] += (double)rec.q[f];
                    }
                }
                const SegInts &acc = s_ints[seg];
#pragma unroll
                for (int f = 0; f < 3; ++f) {
                    t.n[f] = (int)((acc.nn >> (21 * f)) & 0x1fffffu);
                    t.m[f] = (int)((acc.mm >> (21 * f)) & 0x1fffffu);
                }
                t.count = (long long)acc.count;
                t.min_codon = (int)acc.min_codon;
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
                    store_orf(out, fp, orf, phase, valid, t.count, t.min_codon, flags, s_len[seg]);
                }
            } else {
                ws.partials[2 * b + (kind == kSegHead ? 0 : 1)] = t;
            }
        }
        __syncthreads();

        // ---- float64 re-walk of the too-close-to-call ORFs, one wave each -----------------
        const int n_re = s_n_recheck;
        if (n_re > 0) {
            for (int k = wave; k < n_re; k += kTileBlock / kWave) {
                const int seg = s_recheck[k];
                const long long orf = c0 + seg;
                const long long len = s_len[seg];
                WalkResult<double> w;
                // a complete segment starts in this tile and ends inside tile + halo: walk LDS
                wave_walk<double>(s_counts + s_qfirst[seg], len, lane, w);
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
        }
    }
}

// ---------------------------------------------------------------------------
// pass 3: ORFs that straddle a tile boundary -- one LANE per tile whose last ORF does
// not end inside it; sums the tail partial of that tile and the head partials of the
// tiles the ORF runs through.  Too-close-to-call ORFs are re-walked by the whole wave.
// ---------------------------------------------------------------------------
template <int TILE>
__global__ __launch_bounds__(kTileBlock) void k_tile_finalize(const int32_t *__restrict__ counts,
                                                              const int64_t *__restrict__ offsets,
                                                              long long n_orfs, TilePlan plan,
                                                              TileWorkspace ws, OrfOutputs out,
                                                              FilterParams fp)
{
    const int lane = threadIdx.x & (kWave - 1);
    const long long b = (long long)blockIdx.x * kTileBlock + threadIdx.x;
    bool work = false;
    long long orf = 0, beg = 0, len = 0, b_end = 0;
    if (b < plan.n_tiles) {
        const long long a0 = ws.tile_first[b];
        const long long a1 = ws.tile_first[b + 1];
        if (a1 > a0) {  // some ORF starts in this tile; only the last one can leave it
            orf = a1 - 1;
            beg = offsets[orf];
            len = (long long)offsets[orf + 1] - beg;
            const long long ntrip_all = (len + 2) / 3;
            long long t1 = (b + 1) * (long long)TILE - plan.mis;
            if (t1 > plan.total_nt) t1 = plan.total_nt;
            const long long last_first = beg + 3 * (ntrip_all - 1);  // first position of the last triplet
            if (ntrip_all > 0 && last_first >= t1) {
                work = true;
                b_end = (last_first + plan.mis) / TILE;
            }
        }
    }
    FrameScore fr[3];
    long long count = 0;
    int min_codon = RP_MIN_CODON_COV_EMPTY;
    bool unsafe = false;
    if (work) {
        double p[3] = {0, 0, 0}, q[3] = {0, 0, 0};
        int n[3] = {0, 0, 0}, m[3] = {0, 0, 0};
        for (long long k = 0; k <= b_end - b; ++k) {
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
#pragma unroll
        for (int f = 0; f < 3; ++f) fr[f] = frame_score(p[f], q[f], n[f], m[f]);
        unsafe = fp32_decision_unsafe(fr);
        if (!unsafe) {
            double phase;
            int valid;
            unsigned flags;
            combine_frames(fr, phase, valid, flags);
            store_orf(out, fp, orf, phase, valid, count, min_codon, flags | RP_FLAG_SPLIT, len);
        }
    }
    // too-close-to-call ORFs: queue them per workgroup; short ones are re-walked in float64
    // by one wave each, long ones (a single wave needs ~3 us per 1000 nt) by all four waves
    constexpr long long kBlockWalkLen = 2048;
    __shared__ long long s_list_orf[kTileBlock], s_list_beg[kTileBlock], s_list_len[kTileBlock];
    __shared__ int s_n;
    __shared__ double s_part[kTileBlock / kWave][6];
    __shared__ int s_parti[kTileBlock / kWave][7];
    __shared__ long long s_partc[kTileBlock / kWave];
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    if (unsafe) {  // hand orf / start / length over: the re-walk then needs no dependent index loads
        const int slot = atomicAdd(&s_n, 1);
        s_list_orf[slot] = orf;
        s_list_beg[slot] = beg;
        s_list_len[slot] = len;
    }
    __syncthreads();
    const int n_list = s_n;
    const int wave = threadIdx.x >> 6;
    for (int k = 0; k < n_list; ++k) {  // workgroup-uniform loop
        const long long orf_s = s_list_orf[k];
        const long long beg_s = s_list_beg[k];
        const long long len_s = s_list_len[k];
        const bool block_walk = len_s > kBlockWalkLen;
        if (!block_walk && (k & (kTileBlock / kWave - 1)) != wave) continue;
        WalkResult<double> w;
        if (block_walk)
            wave_walk<double>(counts + beg_s, len_s, (int)threadIdx.x, w, kTileBlock);
        else
            wave_walk<double>(counts + beg_s, len_s, lane, w);
        FrameScore fr2[3];
        long long count2;
        int min2;
        if (block_walk) {
            // per-wave sums -> LDS -> every thread adds the four partials in the same order
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
            count2 = 0;
            min2 = RP_MIN_CODON_COV_EMPTY;
#pragma unroll
            for (int f = 0; f < 3; ++f) {
                double ps = 0.0, qs = 0.0;
                int ns = 0, ms = 0;
                for (int wv = 0; wv < kTileBlock / kWave; ++wv) {
                    ps += s_part[wv][2 * f];
                    qs += s_part[wv][2 * f + 1];
                    ns += s_parti[wv][2 * f];
                    ms += s_parti[wv][2 * f + 1];
                }
                fr2[f] = frame_score(ps, qs, ns, ms);
            }
            for (int wv = 0; wv < kTileBlock / kWave; ++wv) {
                count2 += s_partc[wv];
                min2 = min(min2, s_parti[wv][6]);
            }
            __syncthreads();  // partial slots are reused by the next long item
        } else {
            wave_reduce_frames(w, fr2, count2, min2);
        }
        double phase;
        int valid;
        unsigned flags;
        combine_frames(fr2, phase, valid, flags);
        if (lane == 0 && (!block_walk || wave == 0))
            store_orf(out, fp, orf_s, phase, valid, count2, min2, flags | RP_FLAG_SPLIT | RP_FLAG_RECHECK64, len_s);
    }
}

}  // namespace rp
