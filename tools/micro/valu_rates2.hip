// Which VALU instructions issue at the SIMD-32 rate (2 cycles per wave64) and which at 4?  8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define OPS(X) \
    X(0, "v_add_u32", "v_add_u32 %0, %1, %0") \
    X(1, "v_add_f32", "v_add_f32 %0, %1, %0") \
    X(2, "v_min_f32", "v_min_f32 %0, %1, %0") \
    X(3, "v_min_u32", "v_min_u32 %0, %1, %0") \
    X(4, "v_cndmask_b32 (vcc)", "v_cndmask_b32 %0, %1, %0, vcc") \
    X(5, "v_cmp_gt_u32 -> vcc", "v_cmp_gt_u32 vcc, %1, %0") \
    X(6, "v_cmp_gt_f32 -> vcc", "v_cmp_gt_f32 vcc, %1, %0") \
    X(7, "v_and_b32", "v_and_b32 %0, %1, %0") \
    X(8, "v_cvt_f32_u32", "v_cvt_f32_u32 %0, %0") \
    X(9, "v_mov_b32", "v_mov_b32 %0, %1") \
    X(10, "v_pk_add_u16", "v_pk_add_u16 %0, %1, %0") \
    X(11, "v_lshl_add_u32", "v_lshl_add_u32 %0, %1, 16, %0") \
    X(12, "v_sad_u8", "v_sad_u8 %0, %1, %1, %0") \
    X(13, "v_fma_f32", "v_fma_f32 %0, %1, %1, %0") \
    X(14, "v_pk_min_u16", "v_pk_min_u16 %0, %1, %0") \
    X(15, "v_max3_u32", "v_max3_u32 %0, %1, %1, %0") \
    X(16, "v_cvt_f32_u32 sdwa w1", "v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1") \
    X(17, "v_min_u32 sdwa w1", "v_min_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD") \
    X(18, "v_lshrrev_b32", "v_lshrrev_b32 %0, 16, %0") \
    X(19, "v_min_i32", "v_min_i32 %0, %1, %0")

template <int OP>
__global__ void __launch_bounds__(256) probe(uint32_t* out, int iters, uint32_t seed)
{
    uint32_t r[16];
#pragma unroll
    for (int i = 0; i < 16; i++) r[i] = seed * (i + 1) + threadIdx.x;
    uint32_t b = seed ^ 0x01020304u;
    asm volatile("" : "+v"(b));
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
#define X(N, NAME, ASM) if (OP == N) asm volatile(ASM : "+v"(r[i]) : "v"(b) : "vcc");
            OPS(X)
#undef X
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
void run(const char* name, uint32_t* out)
{
    const int iters = 20000, wps = 8, blocks = 256 * wps;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(256), 0, 0, out, 2000, 12345u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 12345u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-24s %.3f ms -> %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", name, ms, ms * 1e-3 * 2.4e9 / ((double)iters * 16 * wps));
}

int main()
{
    uint32_t* out; hipMalloc(&out, 256 * 8 * 256 * 4);
#define X(N, NAME, ASM) run<N>(NAME, out);
    OPS(X)
#undef X
    return 0;
}
