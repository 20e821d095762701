// Micro-benchmark: issue cost of the VALU instructions the trace loop is made of, on gfx950.
// For each instruction: 8 independent dependency chains per wave, every CU filled with
// W waves per SIMD; reports SIMD cycles per wave-instruction (assuming the measured clock).
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int kIters = 4096;

#define DEF_KERNEL_F32(NAME, ASM)                                                            \
__global__ void NAME(float *out, float x, float y) {                                          \
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    for (int i = 0; i < kIters; ++i) {                                                        \
        asm volatile(ASM : "+v"(a0) : "v"(x), "v"(y));                                        \
        asm volatile(ASM : "+v"(a1) : "v"(x), "v"(y));                                        \
        asm volatile(ASM : "+v"(a2) : "v"(x), "v"(y));                                        \
        asm volatile(ASM : "+v"(a3) : "v"(x), "v"(y));                                        \
        asm volatile(ASM : "+v"(a4) : "v"(x), "v"(y));                                        \
        asm volatile(ASM : "+v"(a5) : "v"(x), "v"(y));                                        \
        asm volatile(ASM : "+v"(a6) : "v"(x), "v"(y));                                        \
        asm volatile(ASM : "+v"(a7) : "v"(x), "v"(y));                                        \
    }                                                                                         \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;       \
}

typedef float float2v __attribute__((ext_vector_type(2)));
#define DEF_KERNEL_PK(NAME, ASM)                                                             \
__global__ void NAME(float *out, float x, float y) {                                          \
    float2v xx = {x, y}, yy = {y, x};                                                         \
    float2v a0 = {(float)threadIdx.x, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f; \
    for (int i = 0; i < kIters; ++i) {                                                        \
        asm volatile(ASM : "+v"(a0) : "v"(xx), "v"(yy));                                      \
        asm volatile(ASM : "+v"(a1) : "v"(xx), "v"(yy));                                      \
        asm volatile(ASM : "+v"(a2) : "v"(xx), "v"(yy));                                      \
        asm volatile(ASM : "+v"(a3) : "v"(xx), "v"(yy));                                      \
        asm volatile(ASM : "+v"(a4) : "v"(xx), "v"(yy));                                      \
        asm volatile(ASM : "+v"(a5) : "v"(xx), "v"(yy));                                      \
        asm volatile(ASM : "+v"(a6) : "v"(xx), "v"(yy));                                      \
        asm volatile(ASM : "+v"(a7) : "v"(xx), "v"(yy));                                      \
    }                                                                                         \
    float2v s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                        \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;                                   \
}

#define DEF_KERNEL_U64(NAME, ASM)                                                            \
__global__ void NAME(float *out, float x, float y) {                                          \
    uint32_t m = __float_as_uint(x) | 1u;                                                     \
    uint64_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    for (int i = 0; i < kIters; ++i) {                                                        \
        asm volatile(ASM : "+v"(a0) : "v"(m) : "vcc");                                        \
        asm volatile(ASM : "+v"(a1) : "v"(m) : "vcc");                                        \
        asm volatile(ASM : "+v"(a2) : "v"(m) : "vcc");                                        \
        asm volatile(ASM : "+v"(a3) : "v"(m) : "vcc");                                        \
        asm volatile(ASM : "+v"(a4) : "v"(m) : "vcc");                                        \
        asm volatile(ASM : "+v"(a5) : "v"(m) : "vcc");                                        \
        asm volatile(ASM : "+v"(a6) : "v"(m) : "vcc");                                        \
        asm volatile(ASM : "+v"(a7) : "v"(m) : "vcc");                                        \
    }                                                                                         \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7); \
}

DEF_KERNEL_F32(k_fma, "v_fma_f32 %0, %1, %2, %0")
DEF_KERNEL_F32(k_mul, "v_mul_f32 %0, %1, %0")
DEF_KERNEL_F32(k_add, "v_add_f32 %0, %1, %0")
DEF_KERNEL_F32(k_sqrt, "v_sqrt_f32 %0, %0")
DEF_KERNEL_F32(k_rcp, "v_rcp_f32 %0, %0")
DEF_KERNEL_F32(k_cndmask, "v_cndmask_b32 %0, %1, %0, vcc")
DEF_KERNEL_F32(k_mullo, "v_mul_lo_u32 %0, %1, %0")
DEF_KERNEL_F32(k_mulhi, "v_mul_hi_u32 %0, %1, %0")
DEF_KERNEL_F32(k_mul24, "v_mul_u32_u24 %0, %1, %0")
DEF_KERNEL_F32(k_xor, "v_xor_b32 %0, %1, %0")
DEF_KERNEL_F32(k_cmp, "v_cmp_lt_f32 vcc, %1, %0")
DEF_KERNEL_F32(k_rndne, "v_rndne_f32 %0, %0")
DEF_KERNEL_F32(k_cvt, "v_cvt_f32_u32 %0, %0")
DEF_KERNEL_F32(k_cnd64, "v_cndmask_b32_e64 %0, %1, %0, s[10:11]")
DEF_KERNEL_F32(k_cndvcc2, "v_cmp_lt_f32 vcc, %1, %2\n s_nop 1\n v_cndmask_b32 %0, %1, %0, vcc")
DEF_KERNEL_F32(k_max, "v_max_f32 %0, %1, %0")
DEF_KERNEL_F32(k_sub_sgpr, "v_sub_f32 %0, s10, %0")
DEF_KERNEL_F32(k_mov, "v_mov_b32 %0, %1")
DEF_KERNEL_F32(k_and, "v_and_b32 %0, %1, %0")
DEF_KERNEL_F32(k_lshl_add, "v_lshl_add_u32 %0, %0, 1, %1")
DEF_KERNEL_F32(k_add3, "v_add3_u32 %0, %0, %1, %2")
DEF_KERNEL_F32(k_rsq, "v_rsq_f32 %0, %0")
DEF_KERNEL_F32(k_divfixup, "v_div_fixup_f32 %0, %0, %1, %2")
DEF_KERNEL_F32(k_divscale, "v_div_scale_f32 %0, vcc, %0, %1, %2")
DEF_KERNEL_F32(k_divfmas, "v_div_fmas_f32 %0, %0, %1, %2")
DEF_KERNEL_F32(k_cmpx, "v_cmp_lt_f32_e64 s[12:13], %1, %0")
DEF_KERNEL_F32(k_sub_v, "v_sub_f32 %0, %1, %0")
DEF_KERNEL_F32(k_add_s, "v_add_f32 %0, s10, %0")
DEF_KERNEL_F32(k_mul_s, "v_mul_f32 %0, s10, %0")
DEF_KERNEL_F32(k_fma_s, "v_fma_f32 %0, s10, %1, %0")
DEF_KERNEL_F32(k_mul_lit, "v_mul_f32 %0, 0x3e2aaaab, %0")
DEF_KERNEL_F32(k_add_inl, "v_add_f32 %0, 1.0, %0")
DEF_KERNEL_F32(k_addu, "v_add_u32 %0, %1, %0")
DEF_KERNEL_F32(k_sub_neg, "v_add_f32 %0, %1, -%0")
DEF_KERNEL_F32(k_add_e64, "v_add_f32_e64 %0, %1, %0")
DEF_KERNEL_F32(k_min, "v_min_f32 %0, %1, %0")
DEF_KERNEL_F32(k_lshr, "v_lshrrev_b32 %0, 1, %0")
DEF_KERNEL_F32(k_bfe, "v_bfe_u32 %0, %0, 1, 31")
DEF_KERNEL_F32(k_or, "v_or_b32 %0, %1, %0")
DEF_KERNEL_F32(k_readlane, "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
DEF_KERNEL_F32(k_mul_lit2, "v_mul_f32 %0, 0x3e2aaaab, %0")
DEF_KERNEL_F32(k_add_lit2, "v_add_f32 %0, 0x3e2aaaab, %0")
DEF_KERNEL_F32(k_fmac_v, "v_fmac_f32 %0, %1, %2")
DEF_KERNEL_F32(k_fmaak, "v_fmaak_f32 %0, %1, %0, 0x3e2aaaab")
DEF_KERNEL_F32(k_fmamk, "v_fmamk_f32 %0, %1, 0x3e2aaaab, %0")
DEF_KERNEL_F32(k_xor_s, "v_xor_b32 %0, s10, %0")
DEF_KERNEL_F32(k_mov_s, "v_mov_b32 %0, s10")
DEF_KERNEL_F32(k_fma_s2, "v_fma_f32 %0, %1, s10, %0")
DEF_KERNEL_PK(k_pk_fma, "v_pk_fma_f32 %0, %1, %2, %0")
DEF_KERNEL_PK(k_pk_mul, "v_pk_mul_f32 %0, %1, %0")
DEF_KERNEL_PK(k_pk_add, "v_pk_add_f32 %0, %1, %0")
DEF_KERNEL_U64(k_mad64, "v_mad_u64_u32 %0, vcc, %1, %1, %0")

typedef void (*kern_t)(float *, float, float);

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
    float *out;
    CHECK(hipMalloc(&out, (size_t)cus * 64 * 2048 * sizeof(float)));
    struct K { const char *name; kern_t fn; int lanes_ops; };
    std::vector<K> ks = {
        {"v_fma_f32", k_fma, 1}, {"v_mul_f32", k_mul, 1}, {"v_add_f32", k_add, 1},
        {"v_pk_fma_f32", k_pk_fma, 2}, {"v_pk_mul_f32", k_pk_mul, 2}, {"v_pk_add_f32", k_pk_add, 2},
        {"v_sqrt_f32", k_sqrt, 1}, {"v_rcp_f32", k_rcp, 1}, {"v_cndmask_b32", k_cndmask, 1},
        {"v_cmp_lt_f32", k_cmp, 1}, {"v_xor_b32", k_xor, 1}, {"v_mul_u32_u24", k_mul24, 1},
        {"v_mul_lo_u32", k_mullo, 1}, {"v_mul_hi_u32", k_mulhi, 1}, {"v_mad_u64_u32", k_mad64, 1},
        {"v_rndne_f32", k_rndne, 1}, {"v_cvt_f32_u32", k_cvt, 1},
        {"cndmask_e64_sgpr", k_cnd64, 1}, {"cmp+nop+cndmask", k_cndvcc2, 1}, {"v_max_f32", k_max, 1},
        {"v_sub_f32 sgpr", k_sub_sgpr, 1}, {"v_mov_b32", k_mov, 1}, {"v_and_b32", k_and, 1},
        {"v_rsq_f32", k_rsq, 1}, {"v_div_fixup_f32", k_divfixup, 1}, {"v_div_scale_f32", k_divscale, 1},
        {"v_div_fmas_f32", k_divfmas, 1}, {"v_cmp_e64 sgpr", k_cmpx, 1},
        {"v_sub_f32 vgpr", k_sub_v, 1}, {"v_add_f32 sgpr", k_add_s, 1}, {"v_mul_f32 sgpr", k_mul_s, 1},
        {"v_add_f32 inline", k_add_inl, 1}, {"v_add_u32", k_addu, 1}, {"v_add_f32 neg", k_sub_neg, 1},
        {"v_add_f32_e64", k_add_e64, 1}, {"v_min_f32", k_min, 1}, {"v_lshrrev_b32", k_lshr, 1},
        {"v_bfe_u32", k_bfe, 1}, {"v_or_b32", k_or, 1}, {"v_mov_dpp", k_readlane, 1},
        {"v_mul_f32 literal", k_mul_lit2, 1}, {"v_add_f32 literal", k_add_lit2, 1}, {"v_fmac_f32 vgpr", k_fmac_v, 1},
        {"v_fmaak_f32 lit", k_fmaak, 1}, {"v_fmamk_f32 lit", k_fmamk, 1}, {"v_xor_b32 sgpr", k_xor_s, 1},
        {"v_mov_b32 sgpr", k_mov_s, 1}, {"v_fma_f32 sgpr", k_fma_s2, 1},
    };
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("%-16s %28s\n", "instruction", "ns per wave-instr per SIMD at waves/SIMD = 1, 2, 4, 8 (x2.4 = cycles at 2.4 GHz)");
    for (auto &k : ks) {
        printf("%-16s", k.name);
        for (int wps : {1, 2, 4, 8}) {
            // one block of 256 threads = 4 waves = 1 wave per SIMD of a CU; wps blocks per CU
            const int blocks = cus * wps;
            hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double instr_per_simd = (double)kIters * 8 * wps;
            printf("  %7.3f", ms * 1e6 / instr_per_simd);
        }
        printf("\n");
    }
    CHECK(hipFree(out));
    return 0;
}
