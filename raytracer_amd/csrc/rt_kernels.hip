// rt_kernels.hip — gfx950 kernels for the path-tracing dispatch (Raytracing.comp:441-481) and
// their launchers.  One lane owns one pixel for the whole launch; a wave covers an 8x8 pixel tile
// and a 256-thread workgroup a 16x16 tile, so the float4 accumulation rows a wave touches are
// whole 128-byte lines.  The running sum of a pixel lives in registers across all frames of a
// launch and goes to HBM once.
#include "rt_kernels.hpp"

#include "rt_device.hpp"

#pragma clang fp contract(off)

using namespace rtd;

namespace {

constexpr int kBlockThreads = 256;
constexpr int kTile = 16;

struct PixelSetup {
    uint32_t x, y;
    bool in_range;
    V3 origin, dir;
    float lin;  // float(x) + width * float(y), the float part of the seed (Raytracing.comp:462)
};

__device__ __forceinline__ PixelSetup pixel_setup(const RenderParams &p)
{
    PixelSetup s;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    s.x = blockIdx.x * kTile + (wave & 1u) * 8u + (lane & 7u);
    s.y = blockIdx.y * kTile + (wave >> 1) * 8u + (lane >> 3);
    s.in_range = s.x < p.width && s.y < p.height;
    const float fw = p.window_w, fh = p.window_h;
    // Raytracing.comp:449-453
    const float cx = ((float)s.x / fw) * 2.0f - 1.0f;
    const float cy = ((float)s.y / fh) * 2.0f - 1.0f;
    const float *ip = p.inv_proj;
    float tx = ip[0] * cx + ip[4] * cy + ip[8] * 1.0f + ip[12] * 1.0f;
    float ty = ip[1] * cx + ip[5] * cy + ip[9] * 1.0f + ip[13] * 1.0f;
    float tz = ip[2] * cx + ip[6] * cy + ip[10] * 1.0f + ip[14] * 1.0f;
    float tw = ip[3] * cx + ip[7] * cy + ip[11] * 1.0f + ip[15] * 1.0f;
    V3 n = normalize(mk(tx / tw, ty / tw, tz / tw));
    const float *iv = p.inv_view;
    s.dir.x = iv[0] * n.x + iv[4] * n.y + iv[8] * n.z + iv[12] * 0.0f;
    s.dir.y = iv[1] * n.x + iv[5] * n.y + iv[9] * n.z + iv[13] * 0.0f;
    s.dir.z = iv[2] * n.x + iv[6] * n.y + iv[10] * n.z + iv[14] * 0.0f;
    s.origin = mk(p.cam_pos[0], p.cam_pos[1], p.cam_pos[2]);
    s.lin = (float)s.x + fw * (float)s.y;
    return s;
}

__device__ __forceinline__ Sky make_sky(const RenderParams &p)
{
    Sky k;
    k.horizon = mk(p.sky_horizon[0], p.sky_horizon[1], p.sky_horizon[2]);
    k.zenith = mk(p.sky_zenith[0], p.sky_zenith[1], p.sky_zenith[2]);
    k.ground = mk(p.ground[0], p.ground[1], p.ground[2]);
    k.neg_sun = mk(-p.sun_dir[0], -p.sun_dir[2], -p.sun_dir[1]);  // -(v.xzy)
    k.sun_focus = p.sun_focus;
    k.sun_intensity = p.sun_intensity;
    return k;
}

// incomingLight / raysPerPixel (Raytracing.comp:470).  For a power-of-two count the exact
// reciprocal multiply gives the same correctly rounded quotient as the divide.
__device__ __forceinline__ V3 average_samples(V3 l, const RenderParams &p)
{
    if (p.rpp_is_pow2) return l * p.rpp_recip;
    const float r = (float)p.rays_per_pixel;
    return mk(l.x / r, l.y / r, l.z / r);
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <bool VALIDATE>
__device__ __forceinline__ void finish_pixel(const RenderParams &p, const PixelSetup &s, V3 sum, uint32_t last_frame,
                                             uint32_t first_prim, uint32_t hash, uint32_t rng, Counters &cn)
{
    if (s.in_range) {
        const size_t px = (size_t)s.y * p.width + s.x;
        p.accum[px] = make_float4(sum.x, sum.y, sum.z, 1.0f);
        // Raytracing.comp:480: display = sum / frameIndex, stored to the rgba8 image
        const float div = (float)last_frame;
        const uint32_t r = unorm8(sum.x / div), g = unorm8(sum.y / div), b = unorm8(sum.z / div);
        p.out_rgba8[px] = r | (g << 8) | (b << 16) | 0xFF000000u;
        if (VALIDATE) {
            p.hit_ids[px] = first_prim;
            p.chain_hash[px] = hash;
            p.rng_state[px] = rng;
        }
    }
    if (VALIDATE) {
        // lanes outside the image hold zero counters
#pragma unroll
        for (int i = 0; i < 11; i++) {
            const uint32_t tot = wave_sum(s.in_range ? cn.c[i] : 0u);
            if ((threadIdx.x & 63u) == 0u && tot != 0u) atomicAdd(&p.counters[i], (unsigned long long)tot);
        }
    }
}

// --------------------------------------------------------------------------------------------
// Variant 0: loops nested exactly as the shader nests them (frames of the launch, samples of a
// frame, bounces of a path).
// --------------------------------------------------------------------------------------------
template <bool VALIDATE>
__global__ __launch_bounds__(kBlockThreads) void k_megakernel(const RenderParams p)
{
    const PixelSetup s = pixel_setup(p);
    const Sky sky = make_sky(p);
    const DeviceScene &sc = p.scene;
    const uint32_t nS = p.sphere_number, nM = p.mesh_number;
    Counters cn;
    if (VALIDATE) {
#pragma unroll
        for (int i = 0; i < 11; i++) cn.c[i] = 0u;
    }
    uint32_t hash = 2166136261u, first_prim = kMissId, rng = 0u;

    V3 sum = mk(0.0f, 0.0f, 0.0f);
    if (s.in_range && !(p.first_frame == 1u)) {
        const float4 a = p.accum[(size_t)s.y * p.width + s.x];
        sum = mk(a.x, a.y, a.z);
    }
    if (s.in_range) {
        for (uint32_t f = 0; f < p.n_frames; f++) {
            const uint32_t fi = p.first_frame + f;
            rng = (uint32_t)s.lin + fi * 719393u;
            V3 frame_light = mk(0.0f, 0.0f, 0.0f);
            for (uint32_t k = 0; k < p.rays_per_pixel; k++) {
                rng += k;
                V3 o = s.origin, d = s.dir;
                V3 light = mk(0.0f, 0.0f, 0.0f), thr = mk(1.0f, 1.0f, 1.0f);
                if (VALIDATE) cn.c[9]++;
                for (uint32_t i = 0; i < p.max_bounce; i++) {
                    const Hit h = closest_hit<VALIDATE>(sc, nS, nM, o, d, cn);
                    if (VALIDATE) {
                        cn.c[10]++;
                        hash = (hash ^ h.prim) * 16777619u;
                        if (i == 0 && k == 0) first_prim = h.prim;
                    }
                    if (h.prim != kMissId) {
                        if (VALIDATE) {
                            cn.c[6]++;
                            if (h.prim < nS) cn.c[2]++;
                            else cn.c[5]++;
                        }
                        bounce(sc, nS, h, o, d, light, thr, rng);
                    } else {
                        if (VALIDATE) cn.c[7]++;
                        light = light + env_light(sky, d) * thr;
                        break;
                    }
                }
                frame_light = frame_light + light;
            }
            frame_light = average_samples(frame_light, p);
            if (VALIDATE) cn.c[8]++;
            // Raytracing.comp:471-477
            sum = (fi == 1u) ? frame_light : sum + frame_light;
        }
    }
    finish_pixel<VALIDATE>(p, s, sum, p.first_frame + p.n_frames - 1u, first_prim, hash, rng, cn);
}

// --------------------------------------------------------------------------------------------
// Variant 1: the three loops flattened into one loop over segments.  A lane whose path ends
// (miss or bounce limit) starts its own next path in the same iteration slot, so lanes of a wave
// stay busy until their pixel has no frames left; the order of every pixel's operations, and so
// every bit of its result, is unchanged.
// --------------------------------------------------------------------------------------------
template <bool VALIDATE>
__global__ __launch_bounds__(kBlockThreads) void k_regenerate(const RenderParams p)
{
    const PixelSetup s = pixel_setup(p);
    const Sky sky = make_sky(p);
    const DeviceScene &sc = p.scene;
    const uint32_t nS = p.sphere_number, nM = p.mesh_number;
    Counters cn;
    if (VALIDATE) {
#pragma unroll
        for (int i = 0; i < 11; i++) cn.c[i] = 0u;
    }
    uint32_t hash = 2166136261u, first_prim = kMissId;

    V3 sum = mk(0.0f, 0.0f, 0.0f);
    if (s.in_range && !(p.first_frame == 1u)) {
        const float4 a = p.accum[(size_t)s.y * p.width + s.x];
        sum = mk(a.x, a.y, a.z);
    }

    // per-lane cursor: frame f (0-based within the launch), sample k, bounce i
    uint32_t f = 0, k = 0, i = 0;
    uint32_t rng = (uint32_t)s.lin + p.first_frame * 719393u;  // + k (= 0)
    V3 o = s.origin, d = s.dir;
    V3 light = mk(0.0f, 0.0f, 0.0f), thr = mk(1.0f, 1.0f, 1.0f);
    V3 frame_light = mk(0.0f, 0.0f, 0.0f);
    bool active = s.in_range && p.n_frames > 0u;
    if (VALIDATE && active) cn.c[9]++;

    while (active) {
        const Hit h = closest_hit<VALIDATE>(sc, nS, nM, o, d, cn);
        if (VALIDATE) {
            cn.c[10]++;
            hash = (hash ^ h.prim) * 16777619u;
            if (i == 0 && k == 0) first_prim = h.prim;
        }
        bool path_done;
        if (h.prim != kMissId) {
            if (VALIDATE) {
                cn.c[6]++;
                if (h.prim < nS) cn.c[2]++;
                else cn.c[5]++;
            }
            bounce(sc, nS, h, o, d, light, thr, rng);
            i++;
            path_done = i >= p.max_bounce;
        } else {
            if (VALIDATE) cn.c[7]++;
            light = light + env_light(sky, d) * thr;
            path_done = true;
        }
        if (path_done) {
            frame_light = frame_light + light;
            k++;
            if (k >= p.rays_per_pixel) {
                frame_light = average_samples(frame_light, p);
                if (VALIDATE) cn.c[8]++;
                const uint32_t fi = p.first_frame + f;
                sum = (fi == 1u) ? frame_light : sum + frame_light;
                frame_light = mk(0.0f, 0.0f, 0.0f);
                k = 0;
                f++;
                active = f < p.n_frames;
                if (active) rng = (uint32_t)s.lin + (p.first_frame + f) * 719393u;
            } else {
                rng += k;
            }
            if (active) {
                if (VALIDATE) cn.c[9]++;
                o = s.origin;
                d = s.dir;
                light = mk(0.0f, 0.0f, 0.0f);
                thr = mk(1.0f, 1.0f, 1.0f);
                i = 0;
            }
        }
    }
    finish_pixel<VALIDATE>(p, s, sum, p.first_frame + p.n_frames - 1u, first_prim, hash, rng, cn);
}

// display image from an already summed accumulation image: rgba8 = unorm8(sum / divisor).  Also puts
// the alpha of the sum back to 1 (a sum of k partial images carries alpha k).
__global__ __launch_bounds__(kBlockThreads) void k_resolve(float4 *__restrict__ accum, uint32_t *__restrict__ out,
                                                           uint32_t n_pixels, float divisor)
{
    const uint32_t i = blockIdx.x * kBlockThreads + threadIdx.x;
    if (i >= n_pixels) return;
    float4 a = accum[i];
    if (a.w != 1.0f) {
        a.w = 1.0f;
        accum[i] = a;
    }
    const uint32_t r = unorm8(a.x / divisor), g = unorm8(a.y / divisor), b = unorm8(a.z / divisor);
    out[i] = r | (g << 8) | (b << 16) | 0xFF000000u;
}

__global__ __launch_bounds__(kBlockThreads) void k_selftest(int op, const float *__restrict__ x, const float *__restrict__ y,
                                                            float *__restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * kBlockThreads + threadIdx.x;
    if (i >= n) return;
    const float a = x[i];
    const float b = y ? y[i] : 0.0f;
    float r;
    switch (op) {
    case 0: r = 1.0f / a; break;
    case 1: r = sqrtf(a); break;
    case 2: r = log2_poly(a); break;
    case 3: r = exp2_poly(a); break;
    case 4: r = (float)__float_as_uint(a) * 2.3283064365386963e-10f; break;
    case 5: r = pow_poly(a, b); break;
    default: r = a / b; break;
    }
    out[i] = r;
}

}  // namespace

const char *rt_kernel_name(int variant)
{
    switch (variant) {
    case 0: return "k_megakernel";
    case 1: return "k_regenerate";
    default: return "unknown";
    }
}

hipError_t rt_launch_render(int variant, bool validate, const RenderParams &p, hipStream_t stream)
{
    if (p.width == 0 || p.height == 0 || p.n_frames == 0) return hipErrorInvalidValue;
    const dim3 grid((p.width + kTile - 1) / kTile, (p.height + kTile - 1) / kTile, 1);
    const dim3 block(kBlockThreads, 1, 1);
    // the flattened loop assumes every path has at least one segment
    if (variant == 1 && (p.max_bounce == 0 || p.rays_per_pixel == 0)) variant = 0;
    if (variant == 0) {
        if (validate) hipLaunchKernelGGL(k_megakernel<true>, grid, block, 0, stream, p);
        else hipLaunchKernelGGL(k_megakernel<false>, grid, block, 0, stream, p);
    } else if (variant == 1) {
        if (validate) hipLaunchKernelGGL(k_regenerate<true>, grid, block, 0, stream, p);
        else hipLaunchKernelGGL(k_regenerate<false>, grid, block, 0, stream, p);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t rt_launch_resolve(float4 *accum, uint32_t *out, uint32_t n_pixels, float divisor, hipStream_t stream)
{
    const dim3 grid((n_pixels + kBlockThreads - 1) / kBlockThreads, 1, 1);
    hipLaunchKernelGGL(k_resolve, grid, dim3(kBlockThreads, 1, 1), 0, stream, accum, out, n_pixels, divisor);
    return hipGetLastError();
}

hipError_t rt_launch_selftest(int op, const float *x, const float *y, float *out, size_t n, hipStream_t stream)
{
    const dim3 grid((unsigned)((n + kBlockThreads - 1) / kBlockThreads), 1, 1);
    hipLaunchKernelGGL(k_selftest, grid, dim3(kBlockThreads, 1, 1), 0, stream, op, x, y, out, n);
    return hipGetLastError();
}
