// shared_probe.hip -- a SPEED PROBE, not a product path (built into libstreamprobe.so): the LDS stage of a position-major
// scorer for nested candidate indexes (profiles/r06_prefix_prototype.txt).  One workgroup per tile of the COMPACT coverage:
//   1. the tile (+ 2 positions of halo) is staged with the contiguous 16-byte LDS-DMA of the CSR kernel;
//   2. every thread walks 27 consecutive positions, ONE codon term per position (x2 = 2a - b - c, y = b - c, one v_rsq_f32),
//      summed per class (position mod 3) in registers and flushed at every boundary of the tile's ELEMENTARY INTERVALS
//      (ORF piece starts and ends - 2) into the interval's LDS accumulators with INTEGER atomics: the unit-vector sums as
//      2^30 fixed point in 64 bits (order-independent, so the result does not depend on which thread adds first), the
//      census and the codon-sum total by add, the codon minimum by min;
//   3. every (ORF piece x tile) segment folds the intervals it covers -- sums by addition, the minimum by min (which is
//      why it is intervals, not prefix differences) -- for the three frames (class of frame f = c0 +- f) and writes one
//      48-byte record in the product's format.
// What it leaves out (all per ORF, none per position): the codons that straddle two pieces of a spliced ORF, partial last
// codons, tiles that straddle a strand change, tiles with more than kMaxIntervals intervals (skipped and counted).  The
// plan comes from scripts/shared_probe.py (numpy, host).  Sanity: single-piece ORFs' N_f / M_f / read counts against numpy.
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace {

constexpr int kT = 6912;          // positions per tile: 256 threads x 27 -- a multiple of 3 (class = position mod 3 = k mod 3), an odd
constexpr int kPer = 27;          // dword stride between the threads' runs (no LDS bank conflicts), 27 DMA rows of 1 KiB
constexpr int kThreadsS = 256;
constexpr int kMaxIntervals = 192;
constexpr int kMaxBounds = kMaxIntervals + 1;

struct Seg {            // one ORF piece inside one tile (12 bytes)
    uint16_t ia, ib;    // the tile's intervals [ia, ib)
    uint32_t slot;      // record index (segments are numbered in ORF order)
    uint32_t c0_dir;    // bits 0-1: class of frame 0; bit 2: 1 = classes go DOWN with the frame ('-' strand)
};

struct Acc {            // per interval and class (32 bytes)
    long long p, q;     // 2^30 fixed point
    unsigned long long s;  // codon-sum total
    unsigned nm;        // N | M << 16
    unsigned mn;        // minimum codon sum
};

__global__ __launch_bounds__(kThreadsS, 3) void k_shared_probe(const int* __restrict__ cov, long long n_pos, const unsigned* __restrict__ tile_b_off,
                                                               const uint16_t* __restrict__ bounds, const unsigned* __restrict__ tile_s_off,
                                                               const Seg* __restrict__ segs, const uint8_t* __restrict__ tile_rev,
                                                               uint4* __restrict__ rec, long long n_rec, unsigned* __restrict__ skipped) {
    __shared__ __attribute__((aligned(16))) int s_cov[kT + 16];
    __shared__ uint16_t s_b[kMaxBounds + 3];
    __shared__ Acc s_acc[kMaxIntervals][3];
    const long long b = blockIdx.x;
    const long long t0 = b * (long long)kT;
    const int tid = threadIdx.x;
    const unsigned b0 = tile_b_off[b], nb = tile_b_off[b + 1] - b0;  // boundaries inside the tile, ascending, bounds[b0] == 0
    if (nb > (unsigned)kMaxIntervals) {
        if (tid == 0) atomicAdd(skipped, 1u);
        return;
    }
    // 1. stage: rows of 1 KiB, four loader waves (kT * 4 = 27 KiB = 27 rows) + the halo by plain loads
    {
        typedef const __attribute__((address_space(1))) void* gptr_t;
        typedef __attribute__((address_space(3))) void* lptr_t;
        const char* src = reinterpret_cast<const char*>(cov + t0);
        const int wave = tid >> 6, lane = tid & 63;
        const bool whole = t0 + kT + 16 <= n_pos;
        if (whole) {
            for (int r = wave; r < kT * 4 / 1024; r += 4)
                __builtin_amdgcn_global_load_lds((gptr_t)(src + r * 1024 + lane * 16), (lptr_t)(s_cov + r * 256), 16, 0, 2);
            if (tid < 16) s_cov[kT + tid] = cov[t0 + kT + tid];
        } else {
            for (int i = tid; i < kT + 16; i += kThreadsS) s_cov[i] = t0 + i < n_pos ? cov[t0 + i] : 0;
        }
    }
    for (int i = tid; i < (int)nb; i += kThreadsS) s_b[i] = bounds[b0 + i];
    if (tid == 0) s_b[nb] = (uint16_t)65535;
    for (int i = tid; i < (int)nb * 3; i += kThreadsS) {
        Acc z;
        z.p = 0; z.q = 0; z.s = 0; z.nm = 0; z.mn = 0xffffffffu;
        s_acc[i / 3][i % 3] = z;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // 2. terms: 27 positions per thread, class = k % 3
    const bool rev = tile_rev[b] != 0;
    const int p0 = tid * kPer;
    int lo = 0, hi = (int)nb;  // interval holding p0: the last boundary <= p0
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if ((int)s_b[mid] <= p0) lo = mid; else hi = mid;
    }
    int iv = lo;
    int next_b = s_b[iv + 1];
    float ap[3] = {0.f, 0.f, 0.f}, aq[3] = {0.f, 0.f, 0.f};
    unsigned anm[3] = {0, 0, 0}, amn[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu};
    unsigned as[3] = {0, 0, 0};
    auto flush = [&](int into) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (anm[c] == 0 && as[c] == 0) continue;  // (nothing but all-zero codons: the accumulators are untouched)
            Acc* a = &s_acc[into][c];
            atomicAdd(reinterpret_cast<unsigned long long*>(&a->p), (unsigned long long)(long long)__float2ll_rn(ap[c] * 1073741824.f));
            atomicAdd(reinterpret_cast<unsigned long long*>(&a->q), (unsigned long long)(long long)__float2ll_rn(aq[c] * 1073741824.f));
            atomicAdd(&a->s, (unsigned long long)as[c]);
            atomicAdd(&a->nm, anm[c]);
            ap[c] = 0.f; aq[c] = 0.f; as[c] = 0; anm[c] = 0;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (amn[c] != 0xffffffffu) atomicMin(&s_acc[into][c].mn, amn[c]);
            amn[c] = 0xffffffffu;
        }
    };
    int v0 = s_cov[p0], v1 = s_cov[p0 + 1];
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        const int p = p0 + k;
        if (p == next_b) {  // (boundaries are unique: one flush per boundary)
            flush(iv);
            ++iv;
            next_b = s_b[iv + 1];
        }
        const int v2 = s_cov[p + 2];
        const int a = rev ? v2 : v0, bq = v1, c = rev ? v0 : v2;
        const float x2 = (float)(2 * a - bq - c), y = (float)(bq - c);
        const float q = x2 * x2 + 3.f * y * y;
        const float r = q > 0.f ? __builtin_amdgcn_rsqf(q) : 0.f;
        const unsigned sum = (unsigned)(a + bq + c);
        ap[k % 3] += x2 * r;
        aq[k % 3] += 1.7320508f * y * r;
        anm[k % 3] += (sum != 0 ? 1u : 0u) + (q > 0.f ? 65536u : 0u);
        as[k % 3] += sum;
        amn[k % 3] = min(amn[k % 3], sum);
        v0 = v1;
        v1 = v2;
    }
    flush(iv);
    __syncthreads();
    // 3. segments: fold the intervals [ia, ib) per frame
    const unsigned s0 = tile_s_off[b], ns = tile_s_off[b + 1] - s0;
    for (unsigned j = tid; j < ns; j += kThreadsS) {
        const Seg sg = segs[s0 + j];
        const int c0 = (int)(sg.c0_dir & 3u);
        const bool down = (sg.c0_dir & 4u) != 0;
        long long P[3] = {0, 0, 0}, Q[3] = {0, 0, 0};
        unsigned NM[3] = {0, 0, 0};
        unsigned long long S0 = 0;
        unsigned MN0 = 0xffffffffu;
        for (int i = sg.ia; i < sg.ib; ++i) {
#pragma unroll
            for (int f = 0; f < 3; ++f) {
                const int cls = down ? (c0 + 3 - f) % 3 : (c0 + f) % 3;
                const Acc a = s_acc[i][cls];
                P[f] += a.p;
                Q[f] += a.q;
                NM[f] += a.nm;
                if (f == 0) {
                    S0 += a.s;
                    MN0 = min(MN0, a.mn);
                }
            }
        }
        const float k = 1.f / 1073741824.f;
        rec[0 * n_rec + sg.slot] = make_uint4(__float_as_uint((float)P[0] * k), __float_as_uint((float)Q[0] * k), NM[0], (unsigned)(S0 & 0xffffu));
        rec[1 * n_rec + sg.slot] = make_uint4(__float_as_uint((float)P[1] * k), __float_as_uint((float)Q[1] * k), NM[1], MN0);
        rec[2 * n_rec + sg.slot] = make_uint4(__float_as_uint((float)P[2] * k), __float_as_uint((float)Q[2] * k), NM[2], (unsigned)(S0 >> 16));
    }
}

}  // namespace

extern "C" int sp_shared_probe(const void* cov, long long n_pos, long long n_tiles, const void* tile_b_off, const void* bounds,
                               const void* tile_s_off, const void* segs, const void* tile_rev, void* rec, long long n_rec, void* skipped,
                               void* stream) {
    if (n_tiles <= 0 || ((uintptr_t)cov & 15) || ((uintptr_t)rec & 15)) return 1;
    hipLaunchKernelGGL(k_shared_probe, dim3((unsigned)n_tiles), dim3(kThreadsS), 0, (hipStream_t)stream, (const int*)cov, n_pos,
                       (const unsigned*)tile_b_off, (const uint16_t*)bounds, (const unsigned*)tile_s_off, (const Seg*)segs,
                       (const uint8_t*)tile_rev, (uint4*)rec, n_rec, (unsigned*)skipped);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
