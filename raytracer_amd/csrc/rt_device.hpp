// rt_device.hpp — device-side arithmetic and per-segment routines for gfx950 (wave64).
//
// Arithmetic contract ("strict"): every float operation below is one individually rounded IEEE
// binary32 operation, in a fixed order; the translation unit is compiled with -ffp-contract=off so
// hipcc never fuses a multiply into an add; '/' and sqrtf are correctly rounded (hipcc's default
// -fhip-fp32-correctly-rounded-divide-sqrt); min/max are compare+select with GLSL's operand
// rule; pow is a fixed polynomial exp2(y*log2(x)).  Under this contract the kernels reproduce a
// scalar CPU evaluation of the shader bit for bit.
//
// Second profile ("fused", -DRT_PROFILE_FMA): the same sequences with a fused multiply-add, one rounding,
// at a fixed list of sites: the sums of products of dot, cross, mix, reflect and the matrix-vector products,
// the Horner steps of the pow polynomials, b*b - 4ac of the sphere test, origin + direction * t, the blend of
// the shading normals, and light + emission * throughput / light + sky * throughput.  fmaf is correctly
// rounded on the CPU and on gfx950 alike, so kernels and CPU evaluation agree bit for bit within this profile
// too; results differ from the strict profile's by rounding only.  A vendor's shader compiler is free to
// contract the shader this way (Raytracing.comp carries no `precise`).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace rtd {

constexpr uint32_t kMissId = 0xFFFFFFFFu;
constexpr float kFltMax = 3.402823466e+38f;

#ifdef RT_PROFILE_FMA
constexpr bool kFused = true;
#else
constexpr bool kFused = false;
#endif
// a * b + c at a site the fused profile fuses
__device__ __forceinline__ float mul_add(float a, float b, float c)
{
    if constexpr (kFused) return __builtin_fmaf(a, b, c);
    else return a * b + c;
}

struct V3 {
    float x, y, z;
};

__device__ __forceinline__ V3 mk(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float dot(V3 a, V3 b)
{
    if constexpr (kFused) return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x));
    else return a.x * b.x + a.y * b.y + a.z * b.z;
}
__device__ __forceinline__ V3 cross(V3 a, V3 b)
{
    if constexpr (kFused)
        return mk(__builtin_fmaf(a.y, b.z, -(b.y * a.z)), __builtin_fmaf(a.z, b.x, -(b.z * a.x)), __builtin_fmaf(a.x, b.y, -(b.x * a.y)));
    else
        return mk(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
// o + d * t
__device__ __forceinline__ V3 along(V3 o, V3 d, float t)
{
    if constexpr (kFused) return mk(__builtin_fmaf(d.x, t, o.x), __builtin_fmaf(d.y, t, o.y), __builtin_fmaf(d.z, t, o.z));
    else return o + d * t;
}
// l + a * b, componentwise
__device__ __forceinline__ V3 add_mul(V3 l, V3 a, V3 b)
{
    if constexpr (kFused) return mk(__builtin_fmaf(a.x, b.x, l.x), __builtin_fmaf(a.y, b.y, l.y), __builtin_fmaf(a.z, b.z, l.z));
    else return l + a * b;
}
// GLSL min/max operand rule: min(x,y) = y < x ? y : x ; max(x,y) = x < y ? y : x
__device__ __forceinline__ float gmin(float x, float y) { return (y < x) ? y : x; }
__device__ __forceinline__ float gmax(float x, float y) { return (x < y) ? y : x; }
__device__ __forceinline__ float gclamp(float x, float lo, float hi) { return gmin(gmax(x, lo), hi); }
__device__ __forceinline__ float gmix(float a, float b, float t)
{
    if constexpr (kFused) return __builtin_fmaf(b, t, a * (1.0f - t));
    else return a * (1.0f - t) + b * t;
}
__device__ __forceinline__ V3 gmix(V3 a, V3 b, float t)
{
    return mk(gmix(a.x, b.x, t), gmix(a.y, b.y, t), gmix(a.z, b.z, t));
}

// ---- Correctly rounded sqrt, reciprocal and division by a constant for arguments in the normal
// range, in fewer instructions than the compiler's general expansions (which also scale denormals and
// patch special values).  Each returns exactly what sqrtf(x), 1.0f / x and x / c return for every
// argument its range predicate admits; that is checked on the GPU for ALL admitted bit patterns
// (k_selftest_exhaustive).  Callers send everything else through the general operation.
constexpr float kTwoPowMinus96 = 1.2621774483536189e-29f, kTwoPow96 = 7.9228162514264338e28f;
constexpr float kTwoPowMinus60 = 8.6736173798840355e-19f, kTwoPow60 = 1.152921504606846976e18f;

// hardware estimate (within 1 ulp), then the neighbour whose residual changes sign
__device__ __forceinline__ bool sqrt_rn_admits(float x) { return x >= kTwoPowMinus96; }  // up to and including +inf
__device__ __forceinline__ float sqrt_rn(float x)
{
    float s = __builtin_amdgcn_sqrtf(x);
    const float sd = __uint_as_float(__float_as_uint(s) - 1u), su = __uint_as_float(__float_as_uint(s) + 1u);
    const float rd = __builtin_fmaf(-sd, s, x), ru = __builtin_fmaf(-su, s, x);
    s = (rd <= 0.0f) ? sd : s;
    s = (ru > 0.0f) ? su : s;
    return s;
}

// hardware estimate (within 1 ulp), one Newton step, one correction by the exact residual
__device__ __forceinline__ bool rcp_rn_admits(float x) { return x >= kTwoPowMinus60 && x <= kTwoPow60; }
__device__ __forceinline__ float rcp_rn(float x)
{
    float r = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    e = __builtin_fmaf(-x, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}

// x / c for a constant c with rc = 1.0f / c: product, then two corrections by the exact remainder
__device__ __forceinline__ bool div_const_admits(float x)
{
    const float ax = __builtin_fabsf(x);
    return ax >= kTwoPowMinus60 && ax <= kTwoPow60;  // zeros excluded: the corrections would turn -0 into +0
}
__device__ __forceinline__ float div_const_rn(float x, float c, float rc)
{
    float q = x * rc;
    float rem = __builtin_fmaf(-c, q, x);
    q = __builtin_fmaf(rem, rc, q);
    rem = __builtin_fmaf(-c, q, x);
    return __builtin_fmaf(rem, rc, q);
}

__device__ __forceinline__ V3 normalize(V3 a)
{
    const float dd = dot(a, a);
    float inv;
    // 2^-96 <= dd <= 2^96 keeps the root within rcp_rn's range
    if (dd >= kTwoPowMinus96 && dd <= kTwoPow96) inv = rcp_rn(sqrt_rn(dd));
    else inv = 1.0f / sqrtf(dd);
    return a * inv;
}
__device__ __forceinline__ V3 reflect(V3 i, V3 n)
{
    float k = 2.0f * dot(n, i);
    if constexpr (kFused) return mk(__builtin_fmaf(-n.x, k, i.x), __builtin_fmaf(-n.y, k, i.y), __builtin_fmaf(-n.z, k, i.z));
    else return i - n * k;
}
// 3 - 2 t and 2 r - 1 in one instruction: doubling is exact, so the fused form rounds the same exact value
// as the two-operation form does (in both profiles)
__device__ __forceinline__ float three_minus_twice(float t) { return __builtin_fmaf(-2.0f, t, 3.0f); }
__device__ __forceinline__ float twice_minus_one(float r) { return __builtin_fmaf(r, 2.0f, -1.0f); }
__device__ __forceinline__ float smoothstep(float e0, float e1, float x)
{
    float t = gclamp((x - e0) / (e1 - e0), 0.0f, 1.0f);
    return t * t * three_minus_twice(t);
}
// smoothstep with compile-time edges: the division is by a constant
__device__ __forceinline__ float smoothstep_const(float e0, float e1, float x)
{
    const float c = e1 - e0, rc = 1.0f / c;  // folded at compile time for literal edges
    const float num = x - e0;
    float q;
    if (div_const_admits(num)) q = div_const_rn(num, c, rc);
    else q = num / c;
    const float t = gclamp(q, 0.0f, 1.0f);
    return t * t * three_minus_twice(t);
}

// ---- pow(x, y) = exp2(y * log2(x)) with fixed polynomials (coefficients: the public-domain Cephes
// single-precision log / exp2 fits).  Step for step the same sequence as the CPU checker uses.
// Both are written without branches: the polynomial runs on every input (special inputs are first
// replaced by a harmless in-range value) and the defined results for NaN, negative, zero, infinite
// and out-of-range arguments are selected at the end.
__device__ __forceinline__ float log2_poly(float x0)
{
    const float inf = __uint_as_float(0x7F800000u);
    const bool is_den = x0 < __uint_as_float(0x00800000u);
    float x = is_den ? x0 * 16777216.0f : x0;
    int e = is_den ? -24 : 0;
    uint32_t b = __float_as_uint(x);
    e += (int)((b >> 23) & 0xFFu) - 127;
    float m = __uint_as_float((b & 0x007FFFFFu) | 0x3F800000u);
    const bool big = m > 1.41421356237f;
    m = big ? m * 0.5f : m;
    e += big ? 1 : 0;
    float f = m - 1.0f;
    float z = f * f;
    float p = 7.0376836292e-2f;
    p = mul_add(p, f, -1.1514610310e-1f);
    p = mul_add(p, f, 1.1676998740e-1f);
    p = mul_add(p, f, -1.2420140846e-1f);
    p = mul_add(p, f, 1.4249322787e-1f);
    p = mul_add(p, f, -1.6668057665e-1f);
    p = mul_add(p, f, 2.0000714765e-1f);
    p = mul_add(p, f, -2.4999993993e-1f);
    p = mul_add(p, f, 3.3333331174e-1f);
    float y = (p * f) * z;
    y = __builtin_fmaf(-0.5f, z, y);   // y - 0.5 z: halving z is exact (z is 0 or at least 2^-48), so one rounding either way
    float r = y * 0.44269504088896340736f;
    if constexpr (kFused) r = __builtin_fmaf(f, 0.44269504088896340736f, r);
    else r = r + f * 0.44269504088896340736f;
    r = r + y;
    r = r + f;
    r = r + (float)e;
    r = (x0 == inf) ? x0 : r;
    r = (x0 == 0.0f) ? -inf : r;
    r = (x0 < 0.0f) ? __uint_as_float(0x7FC00000u) : r;
    r = (x0 == x0) ? r : x0;
    return r;
}

__device__ __forceinline__ float exp2_poly(float x0)
{
    const bool is_nan = !(x0 == x0), too_big = x0 > 128.0f, too_small = x0 < -150.0f;
    const float x = (is_nan || too_big || too_small) ? 0.0f : x0;
    float fl = floorf(x);
    int n = (int)fl;
    float r = x - fl;
    const bool up = r > 0.5f;
    n += up ? 1 : 0;
    r = up ? r - 1.0f : r;
    float p = 1.535336188319500e-4f;
    p = mul_add(p, r, 1.339887440266574e-3f);
    p = mul_add(p, r, 9.618437357674640e-3f);
    p = mul_add(p, r, 5.550332471162809e-2f);
    p = mul_add(p, r, 2.402264791363012e-1f);
    p = mul_add(p, r, 6.931472028550421e-1f);
    p = mul_add(p, r, 1.0f);
    const bool hi = n > 127;
    p = hi ? p * __uint_as_float(0x7F000000u) : p;
    n = hi ? n - 127 : n;
    const bool lo = n < -126;
    p = lo ? p * __uint_as_float(0x00800000u) : p;
    n = lo ? n + 126 : n;
    n = n < -126 ? -126 : n;
    float res = p * __uint_as_float((uint32_t)(n + 127) << 23);
    res = too_small ? 0.0f : res;
    res = too_big ? __uint_as_float(0x7F800000u) : res;
    res = is_nan ? x0 : res;
    return res;
}

__device__ __forceinline__ float pow_poly(float x, float y)
{
    const float r = exp2_poly(y * log2_poly(x));
    return (x == 0.0f && y > 0.0f) ? 0.0f : r;
}

// ---- RNG (Raytracing.comp:106-112).  The shader divides by the literal 4294967295.0, which is
// 2^32 in binary32, so the scaling is exact.
__device__ __forceinline__ float rng_next(uint32_t &s)
{
    s = s * 747796405u + 2891336453u;
    uint32_t r = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
    r = (r >> 22) ^ r;
    return (float)r * 2.3283064365386963e-10f;  // exact: 2^-32
}

// float -> UNORM8 of the rgba8 image: NaN -> 0, clamp, *255, round to nearest even.
__device__ __forceinline__ uint32_t unorm8(float v)
{
    if (!(v == v)) return 0u;
    float c = gclamp(v, 0.0f, 1.0f);
    return (uint32_t)rintf(c * 255.0f);
}

// ---- device scene: uniform-indexed records are 16-byte aligned so the compiler can fetch them with
// scalar loads (one s_load_dwordx4..x16 per record) and feed them to VALU as SGPR operands.
struct DeviceScene {
    const float4 *sphere_geom;  // [S]      {cx, cy, cz, r*r}
    const float4 *tri_isect;    // [T][4]   {p1,-} {AB,-} {AC,-} {N = AB x AC,-}
    const float4 *tri_shade;    // [T][3]   {n1,-} {n2,-} {n3,-}
    const float4 *mesh_box;     // [M][2]   {lo.xyz, bits(start)} {hi.xyz, bits(count)}
    const float4 *materials;    // [S + M][2] {r, g, b, light} {smoothness, 0, 0, 0}
    uint32_t n_spheres, n_triangles, n_meshes;
    // bounding volume hierarchy (only for scenes too large for the uniform loops; null otherwise)
    const uint4 *bvh_nodes;         // [N][2]   {left, right, l.x, l.y} {l.z, r.x, r.y, r.z}: child references and, per axis,
                                    //          lower | upper << 16 step counts of the child boxes in the frame below
    float bvh_base[3];              // frame origin per axis
    float bvh_step[3];
    float bvh_slack;                // bound on the rounding of a dequantised plane relative to the ray origin
    float bvh_sphere_floor;         // smallest radius in the sphere hierarchy (shrinks the margin of its boxes)
    const float4 *bvh_sphere_geom;  // [S]      sphere_geom in leaf order
    const uint32_t *bvh_sphere_id;  // [S]      sphere index of each leaf-order entry
    const float4 *bvh_tri_isect;    // [L][4]   tri_isect in leaf order (L = sum of the meshes' counts)
    const uint32_t *bvh_tri_id;     // [L]      triangle index of each leaf-order entry
    const uint4 *bvh_mesh_info;     // [M]      per mesh {root reference of its hierarchy, cone word, reach, 0}: the last two
                                    //          for rays that run almost inside a triangle's plane (rt_bvh_device.hpp)
    uint32_t bvh_sphere_root;       // root reference of the sphere hierarchy
    uint32_t bvh_node_count;        // nodes in bvh_nodes (all hierarchies)
    // rays almost inside a triangle's plane: per node of the meshes' hierarchies {left cone, right cone,
    // left reach, right reach} (rt_bvh_device.hpp); entry 0 belongs to node bvh_graze_first
    const uint4 *bvh_graze;
    uint32_t bvh_graze_first;
    float bvh_graze_cap;            // largest margin, as a fraction of the distance, that such a ray is given
    uint32_t has_glass;             // some material transmits (the extension in bounce<GLASS>)
};

// Fetch of a record whose index is the same in every lane.  The scene arrays are never written by a
// kernel that reads them, so they may be read through the constant address space, which lets the
// compiler use one scalar load (s_load_dwordx4) into SGPRs whatever other memory traffic the kernel has.
__device__ __forceinline__ float4 ld_uniform(const float4 *base, uint32_t index)
{
#ifdef RT_HOST_EMULATION   // the tests' host build of this header: an ordinary load
    return base[index];
#else
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const f32x4 v = ((const __attribute__((address_space(4))) f32x4 *)base)[index];
#pragma clang diagnostic pop
    return make_float4(v.x, v.y, v.z, v.w);
#endif
}

struct Hit {
    float t;
    uint32_t prim;  // kMissId when nothing was hit
    uint32_t mat;   // row of materials[]
    float u, v;     // barycentrics of the winning triangle
};

struct Counters {
#ifdef RT_PATHS_PROFILE
    uint32_t c[16];   // profiling build of k_paths: lane-occupancy tallies instead of event counters
#else
    uint32_t c[11];
#endif
};

// ClosestHit of Raytracing.comp:338-388 with the ray-invariant terms hoisted and everything that is
// only needed for the winner (hit point, normal, material) left to the caller.  Primitive loops are
// wave-uniform: record fetches are scalar, lanes differ only in the ray.
// DEFER: candidates are stashed and evaluated in batches instead of where they are found (below); same result.
template <bool COUNT, bool DEFER = false>
__device__ __forceinline__ Hit closest_hit_spheres(const DeviceScene &sc, uint32_t n_spheres, V3 o, V3 d, Counters &cn)
{
    Hit best;
    best.t = kFltMax;
    best.prim = kMissId;
    best.mat = 0;
    best.u = 0.0f;
    best.v = 0.0f;

    // RaySphere, Raytracing.comp:130-161.  Every sphere's root is numer / (2a) with the same
    // positive denominator, and a correctly rounded divide is monotonic in its numerator, so the
    // closest sphere is found by comparing numerators and only the winner is divided.  The three
    // places where a numerator comparison could disagree with the comparison of the rounded
    // quotients (root within 1e-5 relative of the 0.0001 threshold, two candidates within 1e-5
    // relative of each other, denominator or quotient outside the normal range) are detected and
    // sent through the literal per-sphere evaluation below, so the outcome is always the one the
    // shader's own sequence gives.
    if (n_spheres > 0) {
        const float a = dot(d, d);
        const float a4 = 4.0f * a;
        const float a2 = 2.0f * a;
        const float thr = 0.0001f * a2;
        const float thr_lo = thr * 0.99999f, thr_hi = thr * 1.00001f;
        float best_n = __uint_as_float(0x7F800000u);
        uint32_t best_i = kMissId;
        bool ambiguous = !(a2 > 9.094947e-13f && a2 < 1.0995116e12f);  // 2^-40 .. 2^40
        // b and the discriminant of sphere g (Raytracing.comp:137-143)
        auto quadratic = [&](const float4 g, float &b, float &disc) {
            V3 oc = o - mk(g.x, g.y, g.z);
            b = 2.0f * dot(oc, d);
            float c = dot(oc, oc) - g.w;
            if constexpr (kFused) disc = __builtin_fmaf(b, b, -(a4 * c));
            else disc = b * b - a4 * c;
            if (COUNT) {
                cn.c[0]++;
                if (disc >= 0.0f) cn.c[1]++;
            }
        };
        // b >= 0 makes the numerator -b - sqrt(disc) <= 0, which can never pass t > 0.0001
        // (a discriminant below 2^-96 is outside the short square root's range: such a ray goes through the
        // literal evaluation below)
        auto candidate = [&](float b, float disc, uint32_t i) {
#if defined(RT_PATHS_PROFILE) && RT_PATHS_PROFILE != 2
            {
                const unsigned long long m = __ballot(disc >= 0.0f && b < 0.0f);
                if (m != 0ull && (threadIdx.x & 63u) == (uint32_t)__ffsll((long long)m) - 1u) cn.c[12]++;  // block runs
            }
#endif
            if (disc >= 0.0f && b < 0.0f) {
#if defined(RT_PATHS_PROFILE) && RT_PATHS_PROFILE != 2
                cn.c[11]++;
#endif
                ambiguous = ambiguous || !sqrt_rn_admits(disc);
                const float n = -b - sqrt_rn(disc);
                if (n > thr_lo && n < best_n) {
                    ambiguous = ambiguous || (n < thr_hi) || (n > best_n * 0.99999f);
                    best_n = n;
                    best_i = i;
                }
            }
        };
        // DEFER: a candidate is not evaluated where it is found.  Few lanes of a wave are candidates of any one
        // sphere (3.5 of 64 on the 11-sphere benchmark scene), so evaluating on the spot runs the block above once
        // per sphere that any lane touches, 6.9 times per trace there.  Instead the lane stashes the candidate
        // (b, discriminant, index), and the wave evaluates everything stashed in one run when a lane that already
        // holds one finds another, and once at the end: 2.9 runs per trace at 8.3 lanes each.  Each lane still
        // sees its own candidates in sphere order, so the outcome is the same bit for bit.  Worth 3 % of the
        // kernel's time on that scene; in front of the reference scene's meshes, with its one sphere, there is
        // nothing to merge and it costs 0.5-0.8 % (one more comparison and branch per sphere), hence the switch.
        float st_b = 0.0f, st_d = 0.0f;
        uint32_t st_i = kMissId;
        auto flush = [&]() {
            if (st_i != kMissId) candidate(st_b, st_d, st_i);
            st_i = kMissId;
        };
        auto found = [&](float b, float disc, uint32_t i) {
            if constexpr (DEFER) {
                const bool cand = disc >= 0.0f && b < 0.0f;
                if (__ballot(cand && st_i != kMissId) != 0ull) flush();
                if (cand) {
                    st_b = b;
                    st_d = disc;
                    st_i = i;
                }
            } else {
                candidate(b, disc, i);
            }
        };
        // Four records per trip while four are left (one 64-byte scalar fetch and one wait for all of them,
        // then four independent quadratics), then two, then one.  Waiting for the scalar cache is what a wave
        // of the path kernels spends most of its idle time on: fetching pairs only cost 3 % more time.
        uint32_t i = 0;
        for (; i + 3 < n_spheres; i += 4) {
            const float4 g0 = ld_uniform(sc.sphere_geom, i);
            const float4 g1 = ld_uniform(sc.sphere_geom, i + 1);
            const float4 g2 = ld_uniform(sc.sphere_geom, i + 2);
            const float4 g3 = ld_uniform(sc.sphere_geom, i + 3);
            float b0, d0, b1, d1, b2, d2, b3, d3;
            quadratic(g0, b0, d0);
            quadratic(g1, b1, d1);
            quadratic(g2, b2, d2);
            quadratic(g3, b3, d3);
            found(b0, d0, i);
            found(b1, d1, i + 1);
            found(b2, d2, i + 2);
            found(b3, d3, i + 3);
        }
        for (; i + 1 < n_spheres; i += 2) {
            const float4 g0 = ld_uniform(sc.sphere_geom, i);
            const float4 g1 = ld_uniform(sc.sphere_geom, i + 1);
            float b0, d0, b1, d1;
            quadratic(g0, b0, d0);
            quadratic(g1, b1, d1);
            found(b0, d0, i);
            found(b1, d1, i + 1);
        }
        if (i < n_spheres) {
            const float4 g0 = ld_uniform(sc.sphere_geom, i);
            float b0, d0;
            quadratic(g0, b0, d0);
            found(b0, d0, i);
        }
        if constexpr (DEFER) {
            if (__ballot(st_i != kMissId) != 0ull) flush();
        }
        if (best_i != kMissId) {
            best.t = best_n / a2;
            best.prim = best_i;
            best.mat = best_i;
            ambiguous = ambiguous || !(best.t < kFltMax);
        }
#if defined(RT_PATHS_PROFILE) && RT_PATHS_PROFILE != 2
        {
            const unsigned long long m = __ballot(ambiguous);
            if (m != 0ull && (threadIdx.x & 63u) == (uint32_t)__ffsll((long long)m) - 1u) cn.c[14]++;
            if (ambiguous) cn.c[13]++;
        }
#endif
        if (ambiguous) {
            // literal evaluation order of the shader, one divide per sphere
            best.t = kFltMax;
            best.prim = kMissId;
            for (uint32_t i = 0; i < n_spheres; i++) {
                const float4 g = ld_uniform(sc.sphere_geom, i);
                V3 oc = o - mk(g.x, g.y, g.z);
                float b = 2.0f * dot(oc, d);
                float c = dot(oc, oc) - g.w;
                float disc;
                if constexpr (kFused) disc = __builtin_fmaf(b, b, -(a4 * c));
                else disc = b * b - a4 * c;
                if (disc >= 0.0f) {
                    float t = (-b - sqrtf(disc)) / a2;
                    if (t > 0.0001f && t < best.t) {
                        best.t = t;
                        best.prim = i;
                        best.mat = i;
                        }
                }
            }
        }
    }
    return best;
}

// One triangle of RayTriangle (Raytracing.comp:286-309) against the best hit so far: best is replaced
// when the shader's own sequence accepts the triangle with a smaller t.  P, AB, AC, N are the 64-byte
// intersection record {p1} {AB} {AC} {AB x AC}; prim and mat are what a win records.
__device__ __forceinline__ void triangle_candidate(const float4 P, const float4 AB, const float4 AC, const float4 N, V3 o, V3 d,
                                                   uint32_t prim, uint32_t mat, Hit &best)
{
    const float det = -dot(d, mk(N.x, N.y, N.z));
    if (det >= 1e-6f) {
        // numerators of t, u, v (Raytracing.comp:291-299 before the multiply by 1/det)
        const V3 ao = o - mk(P.x, P.y, P.z);
        const V3 dao = cross(ao, d);
        const float tn = dot(ao, mk(N.x, N.y, N.z));
        const float un = dot(mk(AC.x, AC.y, AC.z), dao);
        const float vn = -dot(mk(AB.x, AB.y, AB.z), dao);
        // 1/det is positive and at least 2^-128, so a numerator below -2^-20 gives a
        // product below zero (no underflow to -0); u + v exceeds 1 by 2^-11 or more when
        // un + vn > det (1 + 2^-10); t is beyond the best when tn > best.t det (1 + 2^-10).
        // Each of these makes the shader's test fail, so the divide is only made for
        // lanes that none of them rules out.
        const float lim = det * 1.0009765625f;
        const bool ruled_out = (tn < -9.5367431640625e-7f) | (un < -9.5367431640625e-7f) |
                               (vn < -9.5367431640625e-7f) | (un + vn > lim) | (tn > best.t * lim);
        if (!ruled_out) {
            const float inv = 1.0f / det;
            const float t = tn * inv;
            const float u = un * inv;
            const float v = vn * inv;
            const float w = 1.0f - u - v;
            if (t >= 0.0f && u >= 0.0f && v >= 0.0f && w >= 0.0f && t < best.t) {
                best.t = t;
                best.prim = prim;
                best.mat = mat;
                best.u = u;
                best.v = v;
            }
        }
    }
}

// RayBox (Raytracing.comp:311-336) with the reciprocal direction f = 1 / d
__device__ __forceinline__ bool mesh_gate(const float4 lo, const float4 hi, V3 o, V3 f)
{
    float t1 = (lo.x - o.x) * f.x;
    float t2 = (hi.x - o.x) * f.x;
    float t3 = (lo.y - o.y) * f.y;
    float t4 = (hi.y - o.y) * f.y;
    float t5 = (lo.z - o.z) * f.z;
    float t6 = (hi.z - o.z) * f.z;
    float tmin = gmax(gmax(gmin(t1, t2), gmin(t3, t4)), gmin(t5, t6));
    float tmax = gmin(gmin(gmax(t1, t2), gmax(t3, t4)), gmax(t5, t6));
    return !(tmax < 0.0f) && !(tmin > tmax);
}

// The mesh part of ClosestHit: RayBox gate + RayTriangle per mesh, Raytracing.comp:352-368.
template <bool COUNT>
__device__ __forceinline__ void closest_hit_meshes(Hit &best, const DeviceScene &sc, uint32_t n_spheres, uint32_t n_meshes, V3 o,
                                                   V3 d, Counters &cn)
{
    if (n_meshes > 0) {
        const V3 f = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
        for (uint32_t m = 0; m < n_meshes; m++) {
            const float4 lo = ld_uniform(sc.mesh_box, 2 * m + 0);
            const float4 hi = ld_uniform(sc.mesh_box, 2 * m + 1);
            if (COUNT) cn.c[3]++;
            if (mesh_gate(lo, hi, o, f)) {
                const uint32_t start = __float_as_uint(lo.w);
                const uint32_t count = __float_as_uint(hi.w);
                // Two records per trip while two are left: one wait for the scalar cache instead of two.  A launch of
                // one frame lasts as long as its longest lane (the device is almost empty: 0.3 waves per SIMD at
                // 1280 x 720), and a wave by itself spends a third of a segment waiting for these fetches: 121 -> 113 us
                // per frame at the reference's defaults.  (Four per trip: 119 us, the scalar registers run out.  The
                // wave-cooperative kernel has none to spare even for two.)
                uint32_t j = start;
                for (; j + 1u < start + count; j += 2u) {
                    const float4 P0 = ld_uniform(sc.tri_isect, 4 * j + 0), AB0 = ld_uniform(sc.tri_isect, 4 * j + 1);
                    const float4 AC0 = ld_uniform(sc.tri_isect, 4 * j + 2), N0 = ld_uniform(sc.tri_isect, 4 * j + 3);
                    const float4 P1 = ld_uniform(sc.tri_isect, 4 * j + 4), AB1 = ld_uniform(sc.tri_isect, 4 * j + 5);
                    const float4 AC1 = ld_uniform(sc.tri_isect, 4 * j + 6), N1 = ld_uniform(sc.tri_isect, 4 * j + 7);
                    if (COUNT) cn.c[4] += 2u;
                    triangle_candidate(P0, AB0, AC0, N0, o, d, n_spheres + j, sc.n_spheres + m, best);
                    triangle_candidate(P1, AB1, AC1, N1, o, d, n_spheres + j + 1u, sc.n_spheres + m, best);
                }
                for (; j < start + count; j++) {
                    // the whole 64-byte record in one scalar fetch
                    const float4 P = ld_uniform(sc.tri_isect, 4 * j + 0);
                    const float4 AB = ld_uniform(sc.tri_isect, 4 * j + 1);
                    const float4 AC = ld_uniform(sc.tri_isect, 4 * j + 2);
                    const float4 N = ld_uniform(sc.tri_isect, 4 * j + 3);
                    if (COUNT) cn.c[4]++;
                    // material rows follow the uploaded sphere count
                    triangle_candidate(P, AB, AC, N, o, d, n_spheres + j, sc.n_spheres + m, best);
                }
            }
        }
    }
}

template <bool COUNT>
__device__ __forceinline__ Hit closest_hit(const DeviceScene &sc, uint32_t n_spheres, uint32_t n_meshes,
                                           V3 o, V3 d, Counters &cn)
{
    Hit best = closest_hit_spheres<COUNT>(sc, n_spheres, o, d, cn);
    closest_hit_meshes<COUNT>(best, sc, n_spheres, n_meshes, o, d, cn);
    return best;
}

// Hit point and shading normal of the winner (Raytracing.comp:150-151, 304-305), evaluated once.
__device__ __forceinline__ void resolve_hit(const DeviceScene &sc, uint32_t n_spheres, V3 o, V3 d,
                                            const Hit &h, V3 &pos, V3 &nrm)
{
    pos = along(o, d, h.t);
    if (h.prim < n_spheres) {
        const float4 g = sc.sphere_geom[h.prim];
        nrm = normalize(pos - mk(g.x, g.y, g.z));
    } else {
        const uint32_t j = h.prim - n_spheres;
        const float4 n1 = sc.tri_shade[3 * j + 0];
        const float4 n2 = sc.tri_shade[3 * j + 1];
        const float4 n3 = sc.tri_shade[3 * j + 2];
        const float w = 1.0f - h.u - h.v;
        V3 bl;
        if constexpr (kFused) {
            bl = mk(__builtin_fmaf(n3.x, h.v, __builtin_fmaf(n2.x, h.u, n1.x * w)), __builtin_fmaf(n3.y, h.v, __builtin_fmaf(n2.y, h.u, n1.y * w)),
                    __builtin_fmaf(n3.z, h.v, __builtin_fmaf(n2.z, h.u, n1.z * w)));
        } else {
            bl = mk(n1.x, n1.y, n1.z) * w + mk(n2.x, n2.y, n2.z) * h.u + mk(n3.x, n3.y, n3.z) * h.v;
        }
        nrm = normalize(bl);
    }
}

struct Sky {
    V3 horizon, zenith, ground;
    V3 neg_sun;  // -(sunLightDirection.xzy)
    float sun_focus, sun_intensity;
};

// GetEnvironmentLight, Raytracing.comp:390-406
__device__ __forceinline__ V3 env_light(const Sky &sky, V3 d)
{
    float grad_t = pow_poly(smoothstep_const(0.0f, 0.4f, d.y), 0.35f);
    V3 grad = gmix(sky.horizon, sky.zenith, grad_t);
    float sun = pow_poly(gmax(0.0f, dot(d, sky.neg_sun)), sky.sun_focus) * sky.sun_intensity;
    float g2s = smoothstep_const(-0.01f, 0.0f, d.y);
    float mask = (g2s >= 1.0f) ? 1.0f : 0.0f;
    V3 base = gmix(sky.ground, grad, g2s);
    float add = sun * mask;
    return mk(base.x + add, base.y + add, base.z + add);
}

// log2_poly for x in [2^-126, FLT_MAX] and exp2_poly for x in [-125, 127]: the same operations in the same
// order as the general routines, without the scaling of denormals, the range reduction of huge or tiny
// results and the final selections for special inputs, none of which these arguments can trigger.  The bits
// returned are those of log2_poly / exp2_poly.
__device__ __forceinline__ bool log2_poly_normal_admits(float x) { return x >= __uint_as_float(0x00800000u) && x <= kFltMax; }
__device__ __forceinline__ float log2_poly_normal(float x)
{
    const uint32_t b = __float_as_uint(x);
    int e = (int)(b >> 23) - 127;
    float m = __uint_as_float((b & 0x007FFFFFu) | 0x3F800000u);
    const bool big = m > 1.41421356237f;
    m = big ? m * 0.5f : m;
    e += big ? 1 : 0;
    float f = m - 1.0f;
    float z = f * f;
    float p = 7.0376836292e-2f;
    p = mul_add(p, f, -1.1514610310e-1f);
    p = mul_add(p, f, 1.1676998740e-1f);
    p = mul_add(p, f, -1.2420140846e-1f);
    p = mul_add(p, f, 1.4249322787e-1f);
    p = mul_add(p, f, -1.6668057665e-1f);
    p = mul_add(p, f, 2.0000714765e-1f);
    p = mul_add(p, f, -2.4999993993e-1f);
    p = mul_add(p, f, 3.3333331174e-1f);
    float y = (p * f) * z;
    y = __builtin_fmaf(-0.5f, z, y);   // y - 0.5 z: halving z is exact (z is 0 or at least 2^-48), so one rounding either way
    float r = y * 0.44269504088896340736f;
    if constexpr (kFused) r = __builtin_fmaf(f, 0.44269504088896340736f, r);
    else r = r + f * 0.44269504088896340736f;
    r = r + y;
    r = r + f;
    r = r + (float)e;
    return r;
}
__device__ __forceinline__ bool exp2_poly_normal_admits(float x) { return x >= -125.0f && x <= 127.0f; }
__device__ __forceinline__ float exp2_poly_normal(float x)
{
    float fl = floorf(x);
    int n = (int)fl;
    float r = x - fl;
    const bool up = r > 0.5f;
    n += up ? 1 : 0;
    r = up ? r - 1.0f : r;
    float p = 1.535336188319500e-4f;
    p = mul_add(p, r, 1.339887440266574e-3f);
    p = mul_add(p, r, 9.618437357674640e-3f);
    p = mul_add(p, r, 5.550332471162809e-2f);
    p = mul_add(p, r, 2.402264791363012e-1f);
    p = mul_add(p, r, 6.931472028550421e-1f);
    p = mul_add(p, r, 1.0f);
    return p * __uint_as_float((uint32_t)(n + 127) << 23);   // -125 <= n <= 127
}

// GetEnvironmentLight (Raytracing.comp:390-406) for a whole wave: every lane calls it, `want` lanes hold
// a direction whose sky radiance is needed; the value returned to the other lanes is meaningless.
//
// Two thirds of the instructions of env_light are its two pow() evaluations, and a lane needs them only
// for part of its directions: pow(smoothstep(0, 0.4, d.y), 0.35) is exactly 1 from d.y = 0.4 upwards and
// exactly 0 from d.y = 0 downwards (the smoothstep is exactly 1 resp. +0 there, the polynomial log2 of 1 is
// exactly 0 and exp2 of 0 exactly 1, and pow(0, y > 0) is 0 by definition), and the sun term is pow(0, focus)
// = 0 for every direction that points away from the sun.  The pow evaluations that remain (both kinds
// together are usually fewer than 64 per call) are handed out to the lanes of the wave, idle ones included,
// through an LDS mailbox `mail` of kSkyMail floats owned by the wave (56, not 64: with the frame ring of the
// path kernel that is exactly 5 of the 1280-byte units in which a CU hands out its LDS; one float more
// would cost a sixth unit and with it three of the 24 waves a CU holds), evaluated by ONE instance of the
// pow code, and read back with cross-lane reads.  Each evaluation is the arithmetic env_light performs, so the bits are
// the same.  Likewise smoothstep(-0.01, 0, d.y) is exactly 1 for d.y >= 0 and exactly +0 for d.y <= -0.01
// and is only evaluated when some lane's direction lies in between.
constexpr uint32_t kSkyMail = 56u;
__device__ __forceinline__ V3 env_light_wave(const Sky &sky, V3 d, bool want, uint32_t lane, float *mail)
{
    const float grad_s = smoothstep_const(0.0f, 0.4f, d.y);
    const float sun_x = gmax(0.0f, dot(d, sky.neg_sun));
    const bool need_a = want && !(grad_s == 0.0f || grad_s == 1.0f || grad_s != grad_s);   // pow(s, 0.35) = s for 0, 1, NaN
    const bool need_b = want && !(sun_x == 0.0f && sky.sun_focus > 0.0f);                     // pow(0, y > 0) = 0
    float pow_a = grad_s, pow_b = 0.0f;
    const unsigned long long mask_a = __ballot(need_a), mask_b = __ballot(need_b);
    const uint32_t n_a = (uint32_t)__popcll(mask_a), total = n_a + (uint32_t)__popcll(mask_b);
    const unsigned long long below = (1ull << lane) - 1ull;
    const uint32_t rank_a = (uint32_t)__popcll(mask_a & below), rank_b = n_a + (uint32_t)__popcll(mask_b & below);
    for (uint32_t base = 0u; base < total; base += kSkyMail) {   // wave-uniform; a second trip only beyond kSkyMail requests
        const uint32_t slot_a = rank_a - base, slot_b = rank_b - base;   // a request is served in this trip if its slot is < kSkyMail
        const bool now_a = need_a && slot_a < kSkyMail, now_b = need_b && slot_b < kSkyMail;
        if (now_a) mail[slot_a] = grad_s;
        if (now_b) mail[slot_b] = sun_x;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // lane w evaluates request base + w (lanes beyond the last request evaluate whatever the mailbox holds)
        const bool serving = lane < kSkyMail && base + lane < total;
        const float x = mail[lane < kSkyMail ? lane : 0u];
        const float y = (base + lane < n_a) ? 0.35f : sky.sun_focus;
        float lg;
        if (__ballot(serving && !log2_poly_normal_admits(x)) == 0ull) lg = log2_poly_normal(x);
        else lg = log2_poly(x);
        const float arg = y * lg;
        float r;
        if (__ballot(serving && !exp2_poly_normal_admits(arg)) == 0ull) r = exp2_poly_normal(arg);
        else r = exp2_poly(arg);
        r = (x == 0.0f && y > 0.0f) ? 0.0f : r;
        const float back_a = __shfl(r, (int)(slot_a & 63u), 64), back_b = __shfl(r, (int)(slot_b & 63u), 64);
        pow_a = now_a ? back_a : pow_a;
        pow_b = now_b ? back_b : pow_b;
        __builtin_amdgcn_wave_barrier();   // the mailbox is rewritten by the next trip
    }
    const V3 grad = gmix(sky.horizon, sky.zenith, pow_a);
    const float sun = pow_b * sky.sun_intensity;
    // smoothstep(-0.01, 0, d.y): exactly 1 for d.y >= 0, exactly +0 for d.y <= -0.01
    const bool above = d.y >= 0.0f, between = !above && !(d.y <= -0.01f);   // NaN counts as between
    float g2s = above ? 1.0f : 0.0f;
    if (__ballot(want && between) != 0ull) g2s = smoothstep_const(-0.01f, 0.0f, d.y);
    const float mask = (g2s >= 1.0f) ? 1.0f : 0.0f;
    const V3 base_c = gmix(sky.ground, grad, g2s);
    const float add = sun * mask;
    return mk(base_c.x + add, base_c.y + add, base_c.z + add);
}

// The hit branch of Trace, Raytracing.comp:417-431.
//
// GLASS (an extension without counterpart in the shader, whose Material, Raytracing.comp:26-34, has colour,
// emission and smoothness only): a material whose first padding word is above 0.5 transmits, with the index of
// refraction in its second padding word.  Every hit the shader's tests report is from outside (near sphere root
// only, :146-148; front-facing triangles only, :293-294), so the ray always enters: mirrored with Schlick's
// reflectance as probability (the coin is the first of the three random numbers the branch draws anyway),
// refracted otherwise.  Scenes without such a material run the instantiations without this code.
template <bool GLASS = false>
__device__ __forceinline__ void bounce(const DeviceScene &sc, uint32_t n_spheres, const Hit &h,
                                       V3 &o, V3 &d, V3 &light, V3 &thr, uint32_t &rng)
{
    V3 pos, nrm;
    resolve_hit(sc, n_spheres, o, d, h, pos, nrm);
    const float4 m0 = sc.materials[2 * h.mat + 0];
    const float4 m1 = sc.materials[2 * h.mat + 1];   // {smoothness, transmission flag, index of refraction, 0}
    const float smooth = m1.x;
    float rx = twice_minus_one(rng_next(rng));
    float ry = twice_minus_one(rng_next(rng));
    float rz = twice_minus_one(rng_next(rng));
    bool transmits = false;
    if constexpr (GLASS) transmits = m1.y > 0.5f;
    if (transmits) {
        const float ior = m1.z;
        const V3 dn = normalize(d);
        const float cosi = gclamp(-dot(dn, nrm), 0.0f, 1.0f);
        const float eta = 1.0f / ior;
        const float k = 1.0f - (eta * eta) * (1.0f - cosi * cosi);
        float r0 = (1.0f - ior) / (1.0f + ior);
        r0 = r0 * r0;
        const float x = 1.0f - cosi;
        const float x2 = x * x;
        const float refl = r0 + (1.0f - r0) * ((x2 * x2) * x);
        o = pos;
        if (k < 0.0f || rx < twice_minus_one(refl)) d = reflect(dn, nrm);
        else d = along(dn * eta, nrm, eta * cosi - sqrtf(k));
    } else {
        V3 diffuse = normalize(nrm + mk(rx, ry, rz));
        V3 spec = reflect(d, nrm);
        o = pos;
        d = gmix(diffuse, spec, smooth);
    }
    float e = 1.0f * m0.w;
    light = add_mul(light, mk(e, e, e), thr);
    thr = thr * mk(m0.x, m0.y, m0.z);
}

}  // namespace rtd
