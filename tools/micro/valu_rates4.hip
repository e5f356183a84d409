// Issue rates of the instructions the confidence band kernel leans on (64-bit float conversions and arithmetic,
// carry adds, 24-bit multiplies, sub-dword selects, whole-wave DPP shifts).  8 waves per SIMD, 16 independent chains.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define OPS(X) \
    X(0, "v_add_u32 (reference)", "v_add_u32 %0, %2, %0") \
    X(1, "v_cvt_f64_i32", "v_cvt_f64_i32 %1, %0") \
    X(2, "v_cvt_f64_u32", "v_cvt_f64_u32 %1, %0") \
    X(3, "v_cvt_f32_f64", "v_cvt_f32_f64 %0, %1") \
    X(4, "v_mul_f64", "v_mul_f64 %1, %1, %3") \
    X(5, "v_add_f64", "v_add_f64 %1, %1, %3") \
    X(6, "v_fma_f64", "v_fma_f64 %1, %1, %3, %3") \
    X(7, "v_ldexp_f64", "v_ldexp_f64 %1, %1, 16") \
    X(8, "v_add_co_u32 + v_addc_co_u32", "v_add_co_u32 %0, vcc, %2, %0\n v_addc_co_u32 %0, vcc, %2, %0, vcc") \
    X(9, "v_lshl_add_u64", "v_lshl_add_u64 %1, %1, 0, %3") \
    X(10, "v_mul_i32_i24", "v_mul_i32_i24 %0, %2, %0") \
    X(11, "v_mul_i32_i24 sdwa", "v_mul_i32_i24_sdwa %0, sext(%0), sext(%0) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_0") \
    X(12, "v_add_u32 sdwa w1", "v_add_u32_sdwa %0, %2, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD") \
    X(13, "v_sub_u32", "v_sub_u32 %0, %0, %2") \
    X(14, "v_ashrrev_i32", "v_ashrrev_i32 %0, 16, %0") \
    X(15, "v_bfe_i32", "v_bfe_i32 %0, %0, 0, 16") \
    X(16, "v_mov_b32 dpp wave_shr:1", "v_mov_b32_dpp %0, %2 wave_shr:1 row_mask:0xf bank_mask:0xf") \
    X(17, "v_add_u32 dpp wave_shr:1", "v_add_u32_dpp %0, %2, %0 wave_shr:1 row_mask:0xf bank_mask:0xf") \
    X(18, "v_mul_f32", "v_mul_f32 %0, %2, %0") \
    X(19, "v_cvt_f32_i32", "v_cvt_f32_i32 %0, %0") \
    X(20, "v_add3_u32", "v_add3_u32 %0, %2, %2, %0") \
    X(21, "v_perm_b32", "v_perm_b32 %0, %0, %2, %2") \
    X(22, "v_mad_u64_u32", "v_mad_u64_u32 %1, vcc, %0, %2, %1") \
    X(23, "v_cndmask_b32 e64 (sgpr mask)", "v_cndmask_b32_e64 %0, %0, %2, s[10:11]")

template <int OP>
__global__ void __launch_bounds__(256) probe(uint32_t* out, int iters, uint32_t seed)
{
    uint32_t r[16]; double d[16];
#pragma unroll
    for (int i = 0; i < 16; i++) { r[i] = seed * (i + 1) + threadIdx.x; d[i] = 1.0 + 1e-9 * (double)(i + threadIdx.x); }
    uint32_t b = seed ^ 0x01020304u; double c = 1.0000000001;
    asm volatile("s_mov_b64 s[10:11], exec" ::: "s10", "s11");
    asm volatile("" : "+v"(b), "+v"(c));
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
#define X(N, NAME, ASM) if (OP == N) asm volatile(ASM : "+v"(r[i]), "+v"(d[i]) : "v"(b), "v"(c) : "vcc");
            OPS(X)
#undef X
        }
    }
    uint32_t s = 0; double t = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) { s ^= r[i]; t += d[i]; }
    out[blockIdx.x * 256 + threadIdx.x] = s ^ (uint32_t)t;
}

template <int OP>
void run(const char* name, uint32_t* out)
{
    const int iters = 10000, wps = 8, blocks = 256 * wps;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(256), 0, 0, out, 1000, 12345u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 12345u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-32s %.3f ms -> %.2f cycles per wave-instruction(s) per SIMD at 2.4 GHz\n", name, ms, ms * 1e-3 * 2.4e9 / ((double)iters * 16 * wps));
}

int main()
{
    uint32_t* out; hipMalloc(&out, 256 * 8 * 256 * 4);
#define X(N, NAME, ASM) run<N>(NAME, out);
    OPS(X)
#undef X
    return 0;
}
