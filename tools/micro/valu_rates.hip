// Issue-rate probe for the integer instructions of the block matcher: v_add_u32, v_sad_u8, v_msad_u8,
// v_add_u32 with DPP, v_cndmask, ds_bpermute.  hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int OP>
__global__ void __launch_bounds__(256) probe(uint32_t* out, int iters, uint32_t seed)
{
    uint32_t r[16];
#pragma unroll
    for (int i = 0; i < 16; i++) r[i] = seed * (i + 1) + threadIdx.x;
    const uint32_t b = seed ^ 0x01020304u;
    const int addr = ((threadIdx.x + 7) & 63) * 4;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (OP == 0) r[i] = r[i] + b;
            if (OP == 1) r[i] = __builtin_amdgcn_sad_u8(b, seed, r[i]);
            if (OP == 2) r[i] = __builtin_amdgcn_msad_u8(b, seed, r[i]);
            if (OP == 3) r[i] = r[i] + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r[i], 0x111, 0xF, 0xF, true);
            if (OP == 4) r[i] = (r[i] > b) ? seed : r[i] + 1;                 // cmp + cndmask (+add)
            if (OP == 5) r[i] = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)r[i]);
            if (OP == 6) r[i] = min(r[i], b + i);
            asm volatile("" : "+v"(r[i]));
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
void run(const char* name, uint32_t* out, int waves_per_simd)
{
    const int iters = 60000, blocks = 256 * waves_per_simd;     // 256 CUs x (4 waves per block = 1 per SIMD)
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(256), 0, 0, out, 10, 12345u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 12345u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)iters * 16 * waves_per_simd;
    printf("%-22s waves/SIMD %d: %.3f ms -> %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", name, waves_per_simd, ms,
           ms * 1e-3 * 2.4e9 / instr_per_simd);
}

int main()
{
    uint32_t* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    for (int w : {1, 2, 4}) {
        run<0>("v_add_u32", out, w); run<1>("v_sad_u8", out, w); run<2>("v_msad_u8", out, w);
        run<3>("v_add_u32_dpp row_shr", out, w); run<4>("cmp+cndmask+add", out, w); run<5>("ds_bpermute_b32", out, w);
        run<6>("v_min_u32", out, w);
    }
    return 0;
}
