// Packed (v_pk_mul_f32 / v_pk_add_f32) against scalar FP32 on gfx950: cycles per instruction per SIMD for
// C independent dependent-chains per wave, at several waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int CHAINS, typename T>
__global__ __launch_bounds__(64) void k(float *out, float a, float b, int iters, int active_lanes, int vector_operands)
{
    if ((int)threadIdx.x >= active_lanes) return;  // the rest of the kernel runs with a partial EXEC mask
    // per-lane multiplier and addend, so that they live in vector registers (kernel arguments would be
    // scalar-register operands)
    if (vector_operands) {
        a += threadIdx.x * 1e-9f;
        b += threadIdx.x * 1e-9f;
    }
    T x[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) x[c] = (T)(threadIdx.x * 0.001f + c);
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++)
#pragma unroll
            for (int c = 0; c < CHAINS; c++) x[c] = x[c] * a + b;
    }
    T s = (T)0.0f;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) s += x[c];
    if constexpr (sizeof(T) == 8) out[blockIdx.x * 64 + threadIdx.x] = s.x + s.y;
    else out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int CHAINS, typename T>
void run(float *d, int waves_per_simd, const char *name, int active_lanes = 64, int vector_operands = 0)
{
    const int blocks = 256 * 4 * waves_per_simd, iters = 1024;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int r = 0; r < 30; r++) hipLaunchKernelGGL((k<CHAINS, T>), dim3(blocks), dim3(64), 0, 0, d, 1.0001f, 0.5f, iters, active_lanes, vector_operands);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<CHAINS, T>), dim3(blocks), dim3(64), 0, 0, d, 1.0001f, 0.5f, iters, active_lanes, vector_operands);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_wave = (double)iters * 16 * CHAINS * 2;
    printf("%s chains %d waves/SIMD %d active lanes %d vector operands %d: %.2f cycles per instruction per SIMD (2.4 GHz)\n", name, CHAINS, waves_per_simd, active_lanes, vector_operands,
           ms * 1e-3 * 2.4e9 / (instr_per_wave * blocks / 1024.0));
}

int main()
{
    float *d;
    (void)hipMalloc(&d, 256 * 4 * 8 * 64 * sizeof(float));
    for (int vo : {0, 1}) {
        run<4, float>(d, 4, "scalar", 64, vo);
        run<4, v2f>(d, 4, "packed", 64, vo);
        run<4, float>(d, 4, "scalar", 16, vo);
        run<4, float>(d, 4, "scalar", 32, vo);
    }
    return 0;
}
