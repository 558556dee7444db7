// Stand-in for <hip/hip_runtime.h> when the device headers of the library are compiled FOR THE HOST by the
// tests (tests/emu): one lane, no GPU.  It gives the device code the few names it uses: vector types,
// qualifiers that mean nothing on the host, and one-lane versions of the cross-lane operations (a ballot is
// the lane's own bit, a shuffle returns the lane's own value).  The hardware square-root and reciprocal
// estimates become the correctly rounded operations, which are inside the 1 ulp the device code allows for.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>

#define RT_HOST_EMULATION 1

#define __device__
#define __host__
#define __global__
#define __shared__ static
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)

struct float4 {
    float x, y, z, w;
};
struct uint4 {
    uint32_t x, y, z, w;
};
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }

struct EmuDim3 {
    uint32_t x = 0, y = 0, z = 0;
};
static EmuDim3 threadIdx, blockIdx, blockDim, gridDim;

static inline unsigned long long __ballot(bool p) { return p ? 1ull : 0ull; }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
template <class T>
static inline T __shfl(T v, int, int)
{
    return v;
}
static inline float __uint_as_float(uint32_t u)
{
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
static inline uint32_t __float_as_uint(float f)
{
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}
static inline float rtmi_emu_sqrt(float x) { return std::sqrt(x); }
static inline float rtmi_emu_rcp(float x) { return 1.0f / x; }
#define __builtin_amdgcn_sqrtf(x) rtmi_emu_sqrt(x)
#define __builtin_amdgcn_rcpf(x) rtmi_emu_rcp(x)
#define __builtin_amdgcn_fence(...) ((void)0)
#define __builtin_amdgcn_wave_barrier() ((void)0)
#define __builtin_amdgcn_ds_bpermute(addr, v) (v)
#define __builtin_amdgcn_ds_permute(addr, v) (v)
#define __builtin_amdgcn_s_memtime() 0ull
