// VALU issue model of gfx950: time per wave-instruction for dependent and independent chains of
// non-fused v_mul / v_add, at 1..8 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize (with SLP the
// compiler packs two chains into v_pk_* instructions)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CHAINS>
__global__ __launch_bounds__(64) void k(float *out, float a, float b, int iters)
{
    float x[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) x[c] = threadIdx.x * 0.001f + c;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++)
#pragma unroll
            for (int c = 0; c < CHAINS; c++) x[c] = x[c] * a + b;  // two instructions, the add depends on the mul
    }
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) s += x[c];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int CHAINS>
void run(float *d, int waves_per_simd)
{
    const int blocks = 256 * 4 * waves_per_simd, iters = 2048;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    // warm the clocks up: short kernels otherwise run at idle frequency
    for (int r = 0; r < 40; r++) hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(64), 0, 0, d, 1.0001f, 0.5f, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(64), 0, 0, d, 1.0001f, 0.5f, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_wave = (double)iters * 16 * CHAINS * 2;
    const double total = instr_per_wave * blocks;
    // cycles per wave-instruction per SIMD at 2.4 GHz
    printf("chains %d waves/SIMD %d: %.3f ms, %.2f cycles per instruction per SIMD, %.2f cycles per instruction of one wave\n", CHAINS,
           waves_per_simd, ms, ms * 1e-3 * 2.4e9 / (total / 1024.0), ms * 1e-3 * 2.4e9 / instr_per_wave);
}

int main()
{
    float *d;
    hipMalloc(&d, 256 * 4 * 8 * 64 * sizeof(float));
    for (int w : {1, 2, 3, 4, 5, 6, 8}) run<1>(d, w);
    for (int w : {1, 2, 4, 8}) run<2>(d, w);
    for (int w : {1, 2, 4, 8}) run<4>(d, w);
    return 0;
}
