// Do different VALU instruction classes overlap on gfx950?  Each kernel runs NF float mul/add steps and NI
// integer (or other) steps per trip on independent chains; if classes overlapped, time would follow
// the larger group instead of the sum.  4 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int NF, int NI, int KIND>
__global__ __launch_bounds__(64) void k(uint32_t *out, float a, float b, uint32_t m, int iters)
{
    float f[4];
    uint32_t q[4];
    v2f p[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        f[c] = threadIdx.x * 0.01f + c;
        q[c] = threadIdx.x * 2654435761u + c;
        p[c] = (v2f)(threadIdx.x * 0.02f + c);
    }
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int c = 0; c < NF; c++) f[c & 3] = f[c & 3] * a + b;            // 2 FP32 instructions
#pragma unroll
            for (int c = 0; c < NI; c++) {
                if (KIND == 0) q[c & 3] = (q[c & 3] >> 3) ^ (q[c & 3] + m);      // shift, add, xor: 3 integer instructions
                if (KIND == 1) q[c & 3] = q[c & 3] * (q[c & 3] | 1u) + m;        // v_mul_lo_u32 + or + add
                if (KIND == 2) p[c & 3] = p[c & 3] * a + b;                      // 2 packed FP32 instructions
                if (KIND == 3) f[c & 3] = (f[c & 3] > b) ? f[c & 3] * a : f[c & 3];  // cmp + mul + cndmask
            }
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) s += __float_as_uint(f[c]) + q[c] + __float_as_uint(p[c].x + p[c].y);
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int NF, int NI, int KIND>
void run(uint32_t *d, const char *name)
{
    const int blocks = 256 * 4 * 4, iters = 1024;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int r = 0; r < 20; r++) hipLaunchKernelGGL((k<NF, NI, KIND>), dim3(blocks), dim3(64), 0, 0, d, 1.0001f, 0.5f, 0x9E3779B9u, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NF, NI, KIND>), dim3(blocks), dim3(64), 0, 0, d, 1.0001f, 0.5f, 0x9E3779B9u, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double trips = (double)iters * 8 * blocks / 1024.0;
    printf("%-44s %.1f cycles per trip per SIMD at 2.4 GHz\n", name, ms * 1e-3 * 2.4e9 / trips);
}

int main()
{
    uint32_t *d;
    (void)hipMalloc(&d, 256 * 4 * 4 * 64 * 4);
    run<4, 0, 0>(d, "8 fp32");
    run<0, 4, 0>(d, "12 int (shift add xor)");
    run<4, 4, 0>(d, "8 fp32 + 12 int");
    run<0, 4, 1>(d, "4 mul_lo + 8 int");
    run<4, 4, 1>(d, "8 fp32 + 4 mul_lo + 8 int");
    run<0, 4, 2>(d, "8 packed fp32");
    run<4, 4, 2>(d, "8 fp32 + 8 packed fp32");
    run<0, 4, 3>(d, "4 x (cmp, mul, cndmask)");
    return 0;
}
