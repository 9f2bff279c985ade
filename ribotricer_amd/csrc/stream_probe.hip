// libstreamprobe.so -- a measuring stick, not part of the product: how fast does THIS GPU, in THIS
// process, at THIS moment stream a buffer out of HBM?  bench.py and scripts/clock_trace.py run it on
// the very counts buffer the scoring kernel reads, so that a roofline fraction measured on a slow
// box (or a slow minute of a fast one: the same kernel on the same bytes was seen at 2.70, 2.81,
// 2.90 and 3.08 ms on four boxes of one pool, shader clock pinned at 2.39 GHz every time) can be
// read next to what a plain read reached there and then.
//
// Two flavours, both reading every byte once in 32 KiB pieces, one workgroup of 256 threads per piece
// (the scoring kernel's shape: 31 KiB tiles, 4 workgroups per CU):
//   sp_stream_read      8 x global_load_dwordx4 (nt) per thread in flight, summed in registers
//   sp_stream_read_lds  the piece DMA'd into LDS (global_load_lds_dwordx4, nt) by ONE wave, as
//                       k_tile_score's loader wave does, then one LDS word per thread summed
// C ABI: (pointer, bytes (multiple of 32 768), 8-byte device scratch, stream).
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace {

constexpr int kThreads = 256;
constexpr int kPiece = 32768;               // bytes per workgroup
constexpr int kVec = kPiece / kThreads / 16; // 8 dwordx4 per thread

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(kThreads, 4) void k_stream_read(const u32x4* __restrict__ p, unsigned long long* sink) {
    const u32x4* base = p + (size_t)blockIdx.x * (kPiece / 16) + threadIdx.x;
    u32x4 v[kVec];
#pragma unroll
    for (int i = 0; i < kVec; ++i) v[i] = __builtin_nontemporal_load(base + i * kThreads);
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < kVec; ++i) s += v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
    if (s == 0x9e3779b9u) atomicAdd(sink, 1ull);  // (keeps the loads alive; practically never taken)
}

__global__ __launch_bounds__(kThreads, 4) void k_stream_read_lds(const char* __restrict__ p, unsigned long long* sink) {
    __shared__ __attribute__((aligned(16))) unsigned s_tile[kPiece / 4];
    const char* src = p + (size_t)blockIdx.x * kPiece;
    if (threadIdx.x < 64) {  // one loader wave: 32 rows of 64 lanes x 16 bytes
        typedef const __attribute__((address_space(1))) void* gptr_t;
        typedef __attribute__((address_space(3))) void* lptr_t;
#pragma unroll
        for (int r = 0; r < kPiece / 1024; ++r)  // (per-lane global address; the LDS side adds lane * 16 itself)
            __builtin_amdgcn_global_load_lds((gptr_t)(src + r * 1024 + threadIdx.x * 16), (lptr_t)(s_tile + r * 256), 16, 0, 2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < kPiece / 4 / kThreads; i += 8) s += s_tile[threadIdx.x + i * kThreads];
    if (s == 0x9e3779b9u) atomicAdd(sink, 1ull);
}

// ---- FETCH_SIZE calibration for the fused kernel's access pattern (round 6) ----
// k_tile_score<true> stages its tile with global_load_lds_dword: 4 bytes per lane, one chunk of <= 64 consecutive
// coverage elements per instruction, chunks cut at the 256-byte lines of the source (rp_pieces.hpp: run_head / run_chunks).
// MI355X_MICROARCH.md calibrates FETCH_SIZE (x2) for 16-byte-per-lane streaming reads only; these kernels read a KNOWN
// byte count -- every byte of the buffer exactly once -- in the dword pattern, so that one `rocprofv3 --pmc FETCH_SIZE`
// pass over scripts/fetch_calibration.py gives the factor for it (profiles/r06_fetch_calibration.txt):
//   MODE 0  whole lines: 64 lanes x 4 bytes, ascending            (a forward chunk in the middle of a run)
//   MODE 1  a line in two instructions: lanes 0-23, then 24-63    (the head / tail chunks either side of a cut)
//   MODE 2  whole lines, lane l reads element 63 - l              (a '-' strand chunk: descending addresses)
template <int MODE>
__global__ __launch_bounds__(kThreads, 4) void k_stream_read_lds_dword(const char* __restrict__ p, unsigned long long* sink) {
    __shared__ __attribute__((aligned(16))) unsigned s_tile[kPiece / 4];
    const char* src = p + (size_t)blockIdx.x * kPiece;
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // (four loader waves, as the fused kernel has: wave w takes the lines w, w + 4, ...)
    for (int r = wave; r < kPiece / 256; r += 4) {
        const char* line = src + r * 256;
        if (MODE == 0) {
            __builtin_amdgcn_global_load_lds((gptr_t)(line + lane * 4), (lptr_t)(s_tile + r * 64), 4, 0, 2);
        } else if (MODE == 1) {
            if (lane < 24) __builtin_amdgcn_global_load_lds((gptr_t)(line + lane * 4), (lptr_t)(s_tile + r * 64), 4, 0, 2);
            if (lane >= 24) __builtin_amdgcn_global_load_lds((gptr_t)(line + lane * 4), (lptr_t)(s_tile + r * 64), 4, 0, 2);
        } else {
            __builtin_amdgcn_global_load_lds((gptr_t)(line + (63 - lane) * 4), (lptr_t)(s_tile + r * 64), 4, 0, 2);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < kPiece / 4 / kThreads; i += 8) s += s_tile[threadIdx.x + i * kThreads];
    if (s == 0x9e3779b9u) atomicAdd(sink, 1ull);
}

// ---- a read stream with a sprinkle of writes: what do ~1.2 KB of records per 32 KiB tile cost, and in which form? ----
// The piece is DMA'd into LDS as above (one loader wave); then the workgroup writes `w_bytes` to out + block * w_bytes:
//   mode 0  nothing                     mode 1  one wave, dwordx4 per lane, contiguous
//   mode 2  three waves, a third each, three planes (the product's record layout)
//   mode 3  as 1, nt                    mode 4  as 1, sc0 sc1 (write-through to memory)
//   mode 5  as 1, dword per lane        mode 6  only every 8th workgroup writes, 8 x w_bytes
//   mode 7  scalar stores (s_store_dwordx4 through the scalar cache; one wave, uniform data)
//   mode 8  as 1 but at the START of the workgroup, before its DMA
//   mode 9  return-less 64-bit atomic swaps (execute at the L2), two per 16 bytes
//   mode 10 dword nt   11 sc1 nt   12 sc0 sc1 nt   13 sc0   14 sc1   15 three planes, nt      (+ 0x100: no reads at all)
//   mode 16 / 17  one wave, nt / ordinary: the workgroup's 64-byte pieces `plane_bytes` apart (strided; see the case)
//   mode 18 / 19  one wave, nt / ordinary: contiguous chunks round a ring of `plane_bytes` (does a small write footprint stay on the die?)
__global__ __launch_bounds__(kThreads, 4) void k_stream_rw(const char* __restrict__ p, char* __restrict__ out, long long plane_bytes,
                                                           int w_bytes, int mode, unsigned long long* sink) {
    __shared__ __attribute__((aligned(16))) unsigned s_tile[kPiece / 4];
    const char* src = p + (size_t)blockIdx.x * kPiece;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32x4 val = {threadIdx.x, blockIdx.x, 3u, 4u};
    if (mode == 8 && wave == 0 && lane * 16 < w_bytes) *reinterpret_cast<u32x4*>(out + (size_t)blockIdx.x * w_bytes + lane * 16) = val;
    const bool no_read = (mode & 0x100) != 0;  // + 0x100: the writes alone
    mode &= 0xff;
    if (wave == 3 && !no_read) {
        typedef const __attribute__((address_space(1))) void* gptr_t;
        typedef __attribute__((address_space(3))) void* lptr_t;
#pragma unroll
        for (int r = 0; r < kPiece / 1024; ++r)
            __builtin_amdgcn_global_load_lds((gptr_t)(src + r * 1024 + lane * 16), (lptr_t)(s_tile + r * 256), 16, 0, 2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < kPiece / 4 / kThreads; i += 8) s += s_tile[threadIdx.x + i * kThreads];
    if (s == 0x9e3779b9u) atomicAdd(sink, 1ull);
    char* dst = out + (size_t)blockIdx.x * w_bytes;
    const u32x4 v = {s, val.y, val.x, 7u};
    switch (mode) {
    case 1: if (wave == 0) for (int o = lane * 16; o < w_bytes; o += 1024) *reinterpret_cast<u32x4*>(dst + o) = v; break;
    case 2: if (wave < 3 && lane * 16 < w_bytes / 3) *reinterpret_cast<u32x4*>(out + wave * plane_bytes + (size_t)blockIdx.x * (w_bytes / 3) + lane * 16) = v; break;
    case 3: if (wave == 0) for (int o = lane * 16; o < w_bytes; o += 1024) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(dst + o)); break;
    case 4: if (wave == 0) for (int o = lane * 16; o < w_bytes; o += 1024) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(dst + o), "v"(v) : "memory"); break;
    case 5: if (wave == 0) for (int o = lane * 4; o < w_bytes; o += 256) *reinterpret_cast<unsigned*>(dst + o) = s; break;
    case 6: if (wave == 0 && (blockIdx.x & 7) == 7) for (int o = lane * 16; o < 8 * w_bytes; o += 1024) *reinterpret_cast<u32x4*>(dst - 7 * (size_t)w_bytes + o) = v; break;
    case 7: if (wave == 0) {
            const unsigned su = __builtin_amdgcn_readfirstlane(s);
            for (int o = 0; o < w_bytes; o += 16)  // (uniform loop: one scalar store per 16 bytes)
                asm volatile("s_store_dwordx4 %0, %1, %2" : : "s"(u32x4{su, su, su, su}), "s"(dst), "s"(o) : "memory");
            asm volatile("s_dcache_wb" ::: "memory");
        } break;
    case 9: if (wave == 0) {
            for (int o = lane * 16; o < w_bytes; o += 1024)
                asm volatile("global_atomic_swap_x2 %0, %1, off\n\tglobal_atomic_swap_x2 %0, %2, off offset:8"
                             : : "v"(dst + o), "v"(((unsigned long long)v.y << 32) | v.x), "v"(((unsigned long long)v.w << 32) | v.z) : "memory");
        } break;
    case 10: if (wave == 0) for (int o = lane * 4; o < w_bytes; o += 256) __builtin_nontemporal_store(s, reinterpret_cast<unsigned*>(dst + o)); break;
    case 11: if (wave == 0) for (int o = lane * 16; o < w_bytes; o += 1024) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" : : "v"(dst + o), "v"(v) : "memory"); break;
    case 12: if (wave == 0) for (int o = lane * 16; o < w_bytes; o += 1024) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" : : "v"(dst + o), "v"(v) : "memory"); break;
    case 13: if (wave == 0) for (int o = lane * 16; o < w_bytes; o += 1024) asm volatile("global_store_dwordx4 %0, %1, off sc0" : : "v"(dst + o), "v"(v) : "memory"); break;
    case 14: if (wave == 0) for (int o = lane * 16; o < w_bytes; o += 1024) asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(dst + o), "v"(v) : "memory"); break;
    case 15: if (wave < 3 && lane * 16 < w_bytes / 3)  // three planes, nt
            __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(out + wave * plane_bytes + (size_t)blockIdx.x * (w_bytes / 3) + lane * 16));
        break;
    case 16: case 17: if (wave == 0) {  // 64-byte pieces of one workgroup `plane_bytes` (= stride S, a multiple of 64) apart, memory filled densely:
            // S / 64 consecutive workgroups share a region of pieces x S bytes, workgroup j owning the 64-byte column j of every row.
            // (If a write costs the read stream a bus turn-around per channel visit, pieces that meet in one channel should be cheaper.)
            const long long S = plane_bytes, per = S / 64, pieces = (w_bytes + 63) / 64;
            char* region = out + ((long long)blockIdx.x / per) * (pieces * S) + ((long long)blockIdx.x % per) * 64;
            for (int o = lane * 16; o < w_bytes; o += 1024) {
                u32x4* q = reinterpret_cast<u32x4*>(region + (long long)(o / 64) * S + (o % 64));
                if (mode == 16) __builtin_nontemporal_store(v, q); else *q = v;
            }
        } break;
    case 18: case 19: if (wave == 0) {  // nt / ordinary: the chunks go round a RING of plane_bytes (a footprint the Infinity Cache could hold)
            char* ring = out + ((long long)blockIdx.x % (plane_bytes / w_bytes)) * w_bytes;
            for (int o = lane * 16; o < w_bytes; o += 1024) {
                if (mode == 18) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(ring + o)); else *reinterpret_cast<u32x4*>(ring + o) = v;
            }
        } break;
    default: break;
    }
}

}  // namespace

extern "C" {

// read `bytes` of p in 32 KiB pieces; every workgroup writes w_bytes (multiple of 48) to `out` (>= 3 * plane_bytes, plane_bytes >= blocks * w_bytes)
int sp_stream_rw(const void* p, size_t bytes, void* out, long long plane_bytes, int w_bytes, int mode, void* scratch8, void* stream) {
    if (bytes == 0 || bytes % kPiece || ((uintptr_t)p & 15) || ((uintptr_t)out & 15) || w_bytes % 48 || w_bytes > 3072) return 1;
    hipLaunchKernelGGL(k_stream_rw, dim3((unsigned)(bytes / kPiece)), dim3(kThreads), 0, (hipStream_t)stream, (const char*)p, (char*)out,
                       plane_bytes, w_bytes, mode, (unsigned long long*)scratch8);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

int sp_stream_read(const void* p, size_t bytes, void* scratch8, void* stream) {
    if (bytes == 0 || bytes % kPiece || ((uintptr_t)p & 15)) return 1;
    hipLaunchKernelGGL(k_stream_read, dim3((unsigned)(bytes / kPiece)), dim3(kThreads), 0, (hipStream_t)stream,
                       (const u32x4*)p, (unsigned long long*)scratch8);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

int sp_stream_read_lds(const void* p, size_t bytes, void* scratch8, void* stream) {
    if (bytes == 0 || bytes % kPiece || ((uintptr_t)p & 15)) return 1;
    hipLaunchKernelGGL(k_stream_read_lds, dim3((unsigned)(bytes / kPiece)), dim3(kThreads), 0, (hipStream_t)stream,
                       (const char*)p, (unsigned long long*)scratch8);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

// the fused kernel's dword LDS-DMA pattern over a known byte count (mode 0 / 1 / 2: see k_stream_read_lds_dword)
int sp_stream_read_lds_dword(const void* p, size_t bytes, void* scratch8, void* stream, int mode) {
    if (bytes == 0 || bytes % kPiece || ((uintptr_t)p & 255) || mode < 0 || mode > 2) return 1;
    const dim3 grid((unsigned)(bytes / kPiece)), block(kThreads);
    if (mode == 0)
        hipLaunchKernelGGL(k_stream_read_lds_dword<0>, grid, block, 0, (hipStream_t)stream, (const char*)p, (unsigned long long*)scratch8);
    else if (mode == 1)
        hipLaunchKernelGGL(k_stream_read_lds_dword<1>, grid, block, 0, (hipStream_t)stream, (const char*)p, (unsigned long long*)scratch8);
    else
        hipLaunchKernelGGL(k_stream_read_lds_dword<2>, grid, block, 0, (hipStream_t)stream, (const char*)p, (unsigned long long*)scratch8);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // extern "C"
