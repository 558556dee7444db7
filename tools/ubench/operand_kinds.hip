// Cost of a wave64 FP32 VALU instruction on gfx950 by the kind of its second operand: vector register,
// scalar register, 32-bit literal, inline constant.  4 waves per SIMD, 4 independent chains.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize
#include <hip/hip_runtime.h>
#include <cstdio>

template <int KIND>
__global__ __launch_bounds__(64) void k(float *out, float sa, float sb, int iters)
{
    const float va = sa + threadIdx.x * 1e-9f, vb = sb + threadIdx.x * 1e-9f;  // per lane: vector registers
    float x[4];
#pragma unroll
    for (int c = 0; c < 4; c++) x[c] = threadIdx.x * 0.001f + c;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                if (KIND == 0) x[c] = x[c] * va + vb;
                if (KIND == 1) x[c] = x[c] * sa + sb;
                if (KIND == 2) x[c] = x[c] * 1.00010001f + 0.50010001f;   // 32-bit literals
                if (KIND == 3) x[c] = x[c] * 0.5f + 2.0f;                 // inline constants
                if (KIND == 4) x[c] = x[c] * va + 0.50010001f;            // one literal in two instructions
                if (KIND == 5) x[c] = (x[c] > vb) ? va : x[c] + vb;       // compare, select (vector operands) + add
                if (KIND == 6) x[c] = __builtin_fminf(x[c] + va, vb);     // add + v_min_f32
                if (KIND == 7) x[c] = __builtin_fmaf(x[c], va, vb);       // v_fma_f32, vector operands
                if (KIND == 8) x[c] = __builtin_amdgcn_sqrtf(x[c] + va);  // add + v_sqrt_f32
            }
    }
    out[blockIdx.x * 64 + threadIdx.x] = x[0] + x[1] + x[2] + x[3];
}

template <int KIND>
void run(float *d, const char *name)
{
    const int blocks = 256 * 4 * 4, iters = 1024;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int r = 0; r < 20; r++) hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d, 1.0001f, 0.5f, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d, 1.0001f, 0.5f, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)iters * 16 * 4 * 2 * blocks / 1024.0;
    printf("%-28s %.2f cycles per instruction per SIMD at 2.4 GHz\n", name, ms * 1e-3 * 2.4e9 / instr);
}

int main()
{
    float *d;
    (void)hipMalloc(&d, 256 * 4 * 4 * 64 * 4);
    run<0>(d, "vector registers");
    run<1>(d, "scalar register operand");
    run<2>(d, "32-bit literal operand");
    run<3>(d, "inline constant operand");
    run<4>(d, "literal in every other");
    run<5>(d, "add, compare, select (x1.5)");
    run<6>(d, "add, min");
    run<7>(d, "fma (per 2 counted)");
    run<8>(d, "add, sqrt");
    return 0;
}
