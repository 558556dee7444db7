// Issue cost of VALU instruction classes on gfx950, 4 waves per SIMD, 4 independent chains per wave.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int OP>
__global__ __launch_bounds__(64) void k(uint32_t *out, uint32_t a, uint32_t b, int iters)
{
    uint32_t x[4];
#pragma unroll
    for (int c = 0; c < 4; c++) x[c] = threadIdx.x * 7u + c;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 32; u++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                if (OP == 0) x[c] = __float_as_uint(__uint_as_float(x[c]) * __uint_as_float(a));                 // v_mul_f32
                if (OP == 1) x[c] = __float_as_uint(__builtin_fmaf(__uint_as_float(x[c]), __uint_as_float(a), __uint_as_float(b)));  // v_fma_f32
                if (OP == 2) x[c] = x[c] + a;                                                                      // v_add_u32
                if (OP == 3) x[c] = (x[c] ^ a) ;                                                                   // v_xor_b32
                if (OP == 4) x[c] = x[c] * a;                                                                      // v_mul_lo_u32
                if (OP == 5) x[c] = (x[c] > b) ? a : x[c];                                                         // v_cmp + v_cndmask
                if (OP == 6) x[c] = __float_as_uint(__builtin_amdgcn_sqrtf(__uint_as_float(x[c])));               // v_sqrt_f32
                if (OP == 7) x[c] = __float_as_uint(__builtin_amdgcn_rcpf(__uint_as_float(x[c])));                // v_rcp_f32
                if (OP == 8) x[c] = (x[c] >> 3) ^ x[c];                                                            // shift + xor
                if (OP == 9) x[c] = __float_as_uint((float)x[c]);                                                  // v_cvt_f32_u32
            }
    }
    out[blockIdx.x * 64 + threadIdx.x] = x[0] + x[1] + x[2] + x[3];
}

template <int OP>
void run(uint32_t *d, const char *name, int instr_per_step)
{
    const int blocks = 256 * 4 * 4, iters = 512;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int r = 0; r < 20; r++) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, d, 0x3f800001u, 12345u, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, d, 0x3f800001u, 12345u, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double steps = (double)iters * 32 * 4 * blocks / 1024.0;
    printf("%-24s %.2f cycles per step per SIMD (%d instruction(s) per step) at 2.4 GHz\n", name, ms * 1e-3 * 2.4e9 / steps, instr_per_step);
}

int main()
{
    uint32_t *d;
    (void)hipMalloc(&d, 256 * 4 * 4 * 64 * 4);
    run<0>(d, "v_mul_f32", 1);
    run<1>(d, "v_fma_f32", 1);
    run<2>(d, "v_add_u32", 1);
    run<3>(d, "v_xor_b32", 1);
    run<4>(d, "v_mul_lo_u32", 1);
    run<5>(d, "v_cmp + v_cndmask", 2);
    run<6>(d, "v_sqrt_f32", 1);
    run<7>(d, "v_rcp_f32", 1);
    run<8>(d, "v_lshrrev + v_xor", 2);
    run<9>(d, "v_cvt_f32_u32", 1);
    return 0;
}
