// Does global_load_lds_dwordx4 (LDS-DMA, 16 bytes per lane) take a global address and an LDS base that are only
// 4-byte aligned?  (The fused staging copies pieces whose source and destination alignments differ: if both may be any
// multiple of 4, a forward run could go 256 positions per instruction instead of 64.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_probe(const int *src, int *out, int g_off, int l_off)
{
    __shared__ __attribute__((aligned(16))) int lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = -1;
    __syncthreads();
    const int *g = src + g_off + 4 * threadIdx.x;  // lane k: 16 bytes at element g_off + 4 k
    __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void *)(lds + l_off), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) out[i] = lds[i];
}

int main()
{
    std::vector<int> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = i;
    int *d_src, *d_out;
    hipMalloc(&d_src, 4096 * 4);
    hipMalloc(&d_out, 1024 * 4);
    hipMemcpy(d_src, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    std::vector<int> o(1024);
    for (int g_off = 0; g_off < 4; ++g_off)
        for (int l_off = 0; l_off < 4; ++l_off) {
            hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, d_src, d_out, 64 + g_off, 16 + l_off);
            if (hipDeviceSynchronize() != hipSuccess) { printf("g+%d l+%d: launch failed\n", g_off, l_off); return 1; }
            hipMemcpy(o.data(), d_out, 1024 * 4, hipMemcpyDeviceToHost);
            int good = 0, touched = 0;
            for (int i = 0; i < 1024; ++i) {
                if (o[i] != -1) ++touched;
                const int k = i - (16 + l_off);
                if (k >= 0 && k < 256 && o[i] == 64 + g_off + k) ++good;
            }
            printf("global +%d dwords, lds +%d dwords: %d of 256 right, %d touched; lds[%d..] = %d %d %d %d %d\n", g_off, l_off, good, touched,
                   16 + l_off - 1, o[16 + l_off - 1], o[16 + l_off], o[16 + l_off + 1], o[16 + l_off + 2], o[16 + l_off + 3]);
        }
    return 0;
}
