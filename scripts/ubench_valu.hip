// ubench_valu.hip -- per-instruction VALU issue cost on gfx950 (wave64), to ground the
// instruction-diet decisions of the tile kernel.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/ubench scripts/ubench_valu.hip && /tmp/ubench
// Each kernel issues REP x 64 copies of one instruction on 8 independent register
// chains; every SIMD holds 4 waves, so the figure is throughput, not latency.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define REP 512

#define BODY8(INS)                                                             \
    asm volatile(INS(0) "\n" INS(1) "\n" INS(2) "\n" INS(3) "\n" INS(4) "\n" INS(5) "\n" INS(6) "\n" INS(7) \
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]),        \
                   "+v"(r[6]), "+v"(r[7])                                       \
                 : "v"(a), "v"(b), "s"(m)                                       \
                 : "vcc");

#define KERNEL(NAME, INS)                                                      \
    __global__ __launch_bounds__(256) void NAME(float *out, float a, float b)  \
    {                                                                          \
        float r[8];                                                            \
        for (int i = 0; i < 8; ++i) r[i] = a + i + threadIdx.x;                \
        unsigned long long m = 0x5555555555555555ull;                          \
        for (int it = 0; it < REP; ++it) {                                     \
            BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS) \
        }                                                                      \
        float s = 0;                                                           \
        for (int i = 0; i < 8; ++i) s += r[i];                                 \
        if (s == 12345.678f) out[threadIdx.x] = s;                             \
    }

#define I_FMA(k) "v_fma_f32 %" #k ", %8, %9, %" #k
#define I_MUL(k) "v_mul_f32 %" #k ", %8, %" #k
#define I_ADDF(k) "v_add_f32 %" #k ", %8, %" #k
#define I_MAXF(k) "v_max_f32 %" #k ", %8, %" #k
#define I_ADDU(k) "v_add_u32 %" #k ", %8, %" #k
#define I_SUBU(k) "v_sub_u32 %" #k ", %8, %" #k
#define I_OR(k) "v_or_b32 %" #k ", %8, %" #k
#define I_OR3(k) "v_or3_b32 %" #k ", %8, %9, %" #k
#define I_ADD3(k) "v_add3_u32 %" #k ", %8, %9, %" #k
#define I_MINI(k) "v_min_i32 %" #k ", %8, %" #k
#define I_CVT(k) "v_cvt_f32_i32 %" #k ", %" #k
#define I_RSQ(k) "v_rsq_f32 %" #k ", %" #k
#define I_RCP(k) "v_rcp_f32 %" #k ", %" #k
#define I_SQRT(k) "v_sqrt_f32 %" #k ", %" #k
#define I_CNDS(k) "v_cndmask_b32 %" #k ", %8, %" #k ", %10"
#define I_CNDV(k) "v_cndmask_b32 %" #k ", %8, %" #k ", vcc"
#define I_CMPS(k) "v_cmp_lt_i32 s[20:21], %8, %" #k
#define I_CMPV(k) "v_cmp_lt_i32 vcc, %8, %" #k
#define I_CMPNE(k) "v_cmp_ne_u32 s[20:21], 0, %" #k
#define I_ADDC(k) "v_addc_co_u32 %" #k ", vcc, 0, %" #k ", %10"
#define I_PKFMA(k) "v_pk_fma_f32 v[40:41], v[42:43], v[44:45], v[40:41]"
#define I_PKMUL(k) "v_pk_mul_f32 v[40:41], v[42:43], v[40:41]"
#define I_PKADD(k) "v_pk_add_f32 v[40:41], v[42:43], v[40:41]"
#define I_LSHLADD64(k) "v_lshl_add_u64 v[40:41], v[42:43], 0, v[40:41]"
#define I_DPPMOV(k) "v_mov_b32_dpp %" #k ", %" #k " row_shr:1 row_mask:0xf bank_mask:0xf"
#define I_DPPADD(k) "v_add_f32_dpp %" #k ", %" #k ", %" #k " row_shr:1 row_mask:0xf bank_mask:0xf"
#define I_FMA64(k) "v_fma_f64 v[40:41], v[42:43], v[44:45], v[40:41]"
#define I_ADD64(k) "v_add_f64 v[40:41], v[42:43], v[40:41]"
#define I_MADU24(k) "v_mad_u32_u24 %" #k ", %8, %9, %" #k
#define I_LSHLADD(k) "v_lshl_add_u32 %" #k ", %8, 1, %" #k
#define I_BFE(k) "v_bfe_u32 %" #k ", %" #k ", 3, 5"
#define I_AND(k) "v_and_b32 %" #k ", %8, %" #k
#define I_BPERM(k) "ds_bpermute_b32 %" #k ", %8, %" #k
#define I_MED3(k) "v_med3_f32 %" #k ", %8, %9, %" #k
#define I_SNOP(k) "s_nop 0"

KERNEL(k_fma, I_FMA)
KERNEL(k_mul, I_MUL)
KERNEL(k_addf, I_ADDF)
KERNEL(k_maxf, I_MAXF)
KERNEL(k_addu, I_ADDU)
KERNEL(k_subu, I_SUBU)
KERNEL(k_or, I_OR)
KERNEL(k_or3, I_OR3)
KERNEL(k_add3, I_ADD3)
KERNEL(k_mini, I_MINI)
KERNEL(k_cvt, I_CVT)
KERNEL(k_rsq, I_RSQ)
KERNEL(k_rcp, I_RCP)
KERNEL(k_sqrt, I_SQRT)
KERNEL(k_cnds, I_CNDS)
KERNEL(k_cndv, I_CNDV)
KERNEL(k_cmps, I_CMPS)
KERNEL(k_cmpv, I_CMPV)
KERNEL(k_cmpne, I_CMPNE)
KERNEL(k_addc, I_ADDC)
KERNEL(k_pkfma, I_PKFMA)
KERNEL(k_pkmul, I_PKMUL)
KERNEL(k_pkadd, I_PKADD)
KERNEL(k_lshladd64, I_LSHLADD64)
KERNEL(k_dppmov, I_DPPMOV)
KERNEL(k_dppadd, I_DPPADD)
KERNEL(k_fma64, I_FMA64)
KERNEL(k_add64, I_ADD64)
KERNEL(k_madu24, I_MADU24)
KERNEL(k_lshladd, I_LSHLADD)
KERNEL(k_bfe, I_BFE)
KERNEL(k_and, I_AND)
KERNEL(k_bperm, I_BPERM)
KERNEL(k_med3, I_MED3)
KERNEL(k_snop, I_SNOP)

typedef void (*kern_t)(float *, float, float);
struct Entry {
    const char *name;
    kern_t fn;
};

int main()
{
    float *d;
    hipMalloc(&d, 4096);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double clk = prop.clockRate * 1e3;  // Hz
    printf("device %s, %d CUs, clock %.0f MHz\n", prop.name, cus, clk / 1e6);
    std::vector<Entry> ks = {
        {"v_fma_f32", k_fma},       {"v_mul_f32", k_mul},         {"v_add_f32", k_addf},
        {"v_max_f32", k_maxf},      {"v_add_u32", k_addu},        {"v_sub_u32", k_subu},
        {"v_or_b32", k_or},         {"v_or3_b32", k_or3},         {"v_add3_u32", k_add3},
        {"v_min_i32", k_mini},      {"v_cvt_f32_i32", k_cvt},     {"v_rsq_f32", k_rsq},
        {"v_rcp_f32", k_rcp},       {"v_sqrt_f32", k_sqrt},       {"v_cndmask(sgpr)", k_cnds},
        {"v_cndmask(vcc)", k_cndv}, {"v_cmp_lt->sgpr", k_cmps},   {"v_cmp_lt->vcc", k_cmpv},
        {"v_cmp_ne->sgpr", k_cmpne}, {"v_addc_co(sgpr)", k_addc}, {"v_pk_fma_f32", k_pkfma},
        {"v_pk_mul_f32", k_pkmul},  {"v_pk_add_f32", k_pkadd},    {"v_lshl_add_u64", k_lshladd64},
        {"v_mov_dpp", k_dppmov},    {"v_add_f32_dpp", k_dppadd},  {"v_fma_f64", k_fma64},
        {"v_add_f64", k_add64},     {"v_mad_u32_u24", k_madu24},  {"v_lshl_add_u32", k_lshladd},
        {"v_bfe_u32", k_bfe},       {"v_and_b32", k_and},         {"ds_bpermute_b32", k_bperm},
        {"v_med3_f32", k_med3},     {"s_nop", k_snop},
    };
    const int blocks = cus * 4;  // 4 blocks x 4 waves per CU = 4 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (auto &k : ks) {
        hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, d, 1.0f, 2.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, d, 1.0f, 2.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_wave = (double)REP * 64;
        const double waves_per_simd = 4.0;
        const double cyc = ms * 1e-3 * clk / (instr_per_wave * waves_per_simd);
        printf("%-18s %8.3f ms  %6.2f cycles per wave-instruction per SIMD\n", k.name, ms, cyc);
    }
    return 0;
}
