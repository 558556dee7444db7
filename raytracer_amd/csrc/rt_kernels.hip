// rt_kernels.hip — gfx950 kernels for the path-tracing dispatch (Raytracing.comp:441-481) and
// their launchers.  One lane owns one pixel for a whole launch and keeps the pixel's running sum in
// registers across all frames of the launch, so the float4 accumulation image is read and written
// once per launch instead of once per frame.
//
// Four variants, all bit-identical in every output:
//   k_megakernel   loops nested as the shader nests them (frames, samples, bounces);
//   k_regenerate   the three loops flattened into one loop over segments (a lane that ends a path
//                  starts its next one at once);
//   k_classify + k_paths
//                  the shader casts the same primary ray for every sample of a pixel (no jitter,
//                  Raytracing.comp:449-453), so a pixel's first closest-hit is evaluated once per
//                  launch instead of once per path.  k_classify finishes pixels whose primary ray
//                  leaves the scene (their whole path is the same every frame) and stream-compacts
//                  the others with a wave ballot + prefix count; k_paths traces them: each wave owns 64
//                  compacted pixels and its lanes share all (pixel, frame) work items of those pixels,
//                  a lane that finishes an item taking the wave's next one, while per-frame results
//                  pass through a small LDS ring so that every pixel's sum is still formed in frame
//                  order, and the triangle tests of a mesh are spread over the lanes of the wave;
//   k_stream       (scenes with a hierarchy) persistent waves whose lanes take pixel after pixel from a
//                  launch-wide cursor and resume their walks through the hierarchy.
#include "rt_kernels.hpp"

#include <type_traits>

#include "rt_device.hpp"
#include "rt_bvh_device.hpp"

#pragma clang fp contract(off)

using namespace rtd;

namespace {

constexpr int kBlockThreads = 256;
constexpr int kTile = 16;
constexpr uint32_t kHashInit = 2166136261u, kHashPrime = 16777619u, kHashFrameKey = 0x9E3779B9u;

// Validation hash of a pixel: sum over frames of each frame's own fold of its segment ids, started
// from a value keyed by the frame index.  A sum so that frames may be finished in any order.
__device__ __forceinline__ uint32_t frame_hash_init(uint32_t frame_index) { return kHashInit ^ (frame_index * kHashFrameKey); }

struct PixelSetup {
    uint32_t x, y;
    bool in_range;
    V3 origin, dir;
    float lin;  // float(x) + width * float(y), the float part of the seed (Raytracing.comp:462)
};

// Raytracing.comp:446-453 and the float part of :462 for pixel (x, y)
__device__ __forceinline__ PixelSetup pixel_setup_xy(const RenderParams &p, uint32_t x, uint32_t y, bool in_range)
{
    PixelSetup s;
    s.x = x;
    s.y = y;
    s.in_range = in_range;
    const float fw = p.window_w, fh = p.window_h;
    const float cx = twice_minus_one((float)x / fw);
    const float cy = twice_minus_one((float)y / fh);
    const float *ip = p.inv_proj;
    // matrix times (cx, cy, 1, 1) and times (n, 0), columns summed left to right
    auto row4 = [](float m0, float m1, float m2, float m3, float v0, float v1, float v2, float v3) {
        if constexpr (kFused) return __builtin_fmaf(m3, v3, __builtin_fmaf(m2, v2, __builtin_fmaf(m1, v1, m0 * v0)));
        else return m0 * v0 + m1 * v1 + m2 * v2 + m3 * v3;
    };
    float tx = row4(ip[0], ip[4], ip[8], ip[12], cx, cy, 1.0f, 1.0f);
    float ty = row4(ip[1], ip[5], ip[9], ip[13], cx, cy, 1.0f, 1.0f);
    float tz = row4(ip[2], ip[6], ip[10], ip[14], cx, cy, 1.0f, 1.0f);
    float tw = row4(ip[3], ip[7], ip[11], ip[15], cx, cy, 1.0f, 1.0f);
    V3 n = normalize(mk(tx / tw, ty / tw, tz / tw));
    const float *iv = p.inv_view;
    s.dir.x = row4(iv[0], iv[4], iv[8], iv[12], n.x, n.y, n.z, 0.0f);
    s.dir.y = row4(iv[1], iv[5], iv[9], iv[13], n.x, n.y, n.z, 0.0f);
    s.dir.z = row4(iv[2], iv[6], iv[10], iv[14], n.x, n.y, n.z, 0.0f);
    s.origin = mk(p.cam_pos[0], p.cam_pos[1], p.cam_pos[2]);
    s.lin = (float)x + fw * (float)y;
    return s;
}

// Square tile of 8x8-pixel waves per workgroup of THREADS lanes (256: 16x16 pixels, 1024: 32x32)
template <int THREADS>
__device__ __forceinline__ PixelSetup pixel_setup_tile(const RenderParams &p)
{
    constexpr uint32_t kWavesX = THREADS == 1024 ? 4u : 2u;
    static_assert(THREADS == 256 || THREADS == 1024, "tiles of 2x2 or 4x4 waves");
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t x = blockIdx.x * (kWavesX * 8u) + (wave % kWavesX) * 8u + (lane & 7u);
    const uint32_t y = blockIdx.y * (kWavesX * 8u) + (wave / kWavesX) * 8u + (lane >> 3);
    return pixel_setup_xy(p, x, y, x < p.width && y < p.height);
}

// 16x16 pixel tile per workgroup, 8x8 per wave: the float4 rows a wave touches are whole 128-byte lines
__device__ __forceinline__ PixelSetup pixel_setup(const RenderParams &p)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t x = blockIdx.x * kTile + (wave & 1u) * 8u + (lane & 7u);
    const uint32_t y = blockIdx.y * kTile + (wave >> 1) * 8u + (lane >> 3);
    return pixel_setup_xy(p, x, y, x < p.width && y < p.height);
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

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ void zero(Counters &c)
{
#pragma unroll
    for (int i = 0; i < (int)(sizeof(c.c) / sizeof(c.c[0])); i++) c.c[i] = 0u;
}

// Closest hit by the uniform loops or through the hierarchy; the choice is made per launch.
template <bool COUNT, bool BVH>
__device__ __forceinline__ Hit scene_hit(const DeviceScene &sc, uint32_t n_spheres, uint32_t n_meshes, V3 o, V3 d,
                                         Counters &cn, uint32_t *stack, uint32_t stride)
{
    if constexpr (BVH) return closest_hit_bvh<COUNT>(sc, n_spheres, n_meshes, o, d, cn, stack, stride);
    else return closest_hit<COUNT>(sc, n_spheres, n_meshes, o, d, cn);
}

// per-lane traversal stacks of a workgroup of THREADS lanes: level l of lane t is word l * THREADS + t
#define RT_BVH_STACK(THREADS)                                                     \
    __shared__ uint32_t bvh_stack_mem[BVH ? kBvhStack * (THREADS) : 1];           \
    uint32_t *const bvh_stack = &bvh_stack_mem[BVH ? threadIdx.x : 0u];           \
    constexpr uint32_t bvh_stride = (THREADS)

// Store one finished pixel (Raytracing.comp:479-480) and, in validating builds, its check values;
// then fold the wave's event counters into the global totals.  Every lane of the wave must call it.
// cn holds events counted one by one; per_path holds the events of one cached primary segment, which
// the shader would have repeated for each of the pixel's n_paths paths.
template <bool VALIDATE>
__device__ __forceinline__ void store_pixel(const RenderParams &p, uint32_t x, uint32_t y, V3 sum, uint32_t first_prim, uint32_t hash,
                                            uint32_t rng)
{
    const size_t px = (size_t)y * p.width + x;
    p.accum[px] = make_float4(sum.x, sum.y, sum.z, 1.0f);
    const float div = (float)(p.first_frame + p.n_frames - 1u);
    const uint32_t r = unorm8(sum.x / div), g = unorm8(sum.y / div), b = unorm8(sum.z / div);
    p.out_rgba8[px] = r | (g << 8) | (b << 16) | 0xFF000000u;
    if (VALIDATE) {
        p.hit_ids[px] = first_prim;
        p.chain_hash[px] = hash;
        p.rng_state[px] = rng;
    }
}

template <bool VALIDATE>
__device__ __forceinline__ void finish_pixel(const RenderParams &p, uint32_t x, uint32_t y, bool write, V3 sum,
                                             uint32_t first_prim, uint32_t hash, uint32_t rng, const Counters &cn,
                                             const Counters &per_path, uint32_t n_paths)
{
    if (write) store_pixel<VALIDATE>(p, x, y, sum, first_prim, hash, rng);
    if (VALIDATE) {
#pragma unroll
        for (int i = 0; i < 11; i++) {
            // cn counts what this lane did (for any pixel); per_path belongs to the pixel the lane writes
            const unsigned long long mine = (unsigned long long)cn.c[i] + (write ? (unsigned long long)per_path.c[i] * n_paths : 0ull);
            const unsigned long long tot = wave_sum(mine);
            if ((threadIdx.x & 63u) == 0u && tot != 0ull) atomicAdd(&p.counters[i], tot);
        }
    }
}

__device__ __forceinline__ V3 load_sum(const RenderParams &p, uint32_t x, uint32_t y, bool in_range)
{
    // frame_index == 1 overwrites, so the old contents are only needed otherwise (Raytracing.comp:471-477)
    if (in_range && p.first_frame != 1u) {
        const float4 a = p.accum[(size_t)y * p.width + x];
        return mk(a.x, a.y, a.z);
    }
    return mk(0.0f, 0.0f, 0.0f);
}

// --------------------------------------------------------------------------------------------
// Variant 0: loops nested exactly as the shader nests them.
// --------------------------------------------------------------------------------------------
template <bool VALIDATE, bool BVH, bool GLASS = false>
__global__ __launch_bounds__(kBlockThreads) void k_megakernel(const RenderParams p)
{
    RT_BVH_STACK(kBlockThreads);
    const PixelSetup s = pixel_setup(p);
    const Sky sky = make_sky(p);
    const DeviceScene &sc = p.scene;
    const uint32_t nS = p.sphere_number, nM = p.mesh_number;
    Counters cn, none;
    zero(cn);
    zero(none);
    uint32_t hash = 0u, first_prim = kMissId, rng = 0u;

    V3 sum = load_sum(p, s.x, s.y, s.in_range);
    if (s.in_range) {
        for (uint32_t f = 0; f < p.n_frames; f++) {
            const uint32_t fi = p.first_frame + f;
            rng = (uint32_t)s.lin + fi * 719393u;
            uint32_t fh = frame_hash_init(fi);
            V3 frame_light = mk(0.0f, 0.0f, 0.0f);
            for (uint32_t k = 0; k < p.rays_per_pixel; k++) {
                rng += k;
                V3 o = s.origin, d = s.dir;
                V3 light = mk(0.0f, 0.0f, 0.0f), thr = mk(1.0f, 1.0f, 1.0f);
                if (VALIDATE) cn.c[9]++;
                for (uint32_t i = 0; i < p.max_bounce; i++) {
                    const Hit h = scene_hit<VALIDATE, BVH>(sc, nS, nM, o, d, cn, bvh_stack, bvh_stride);
                    if (VALIDATE) {
                        cn.c[10]++;
                        fh = (fh ^ h.prim) * kHashPrime;
                        if (i == 0 && k == 0) first_prim = h.prim;
                    }
                    if (h.prim != kMissId) {
                        if (VALIDATE) {
                            cn.c[6]++;
                            if (h.prim < nS) cn.c[2]++;
                            else cn.c[5]++;
                        }
                        bounce<GLASS>(sc, nS, h, o, d, light, thr, rng);
                    } else {
                        if (VALIDATE) cn.c[7]++;
                        light = add_mul(light, env_light(sky, d), thr);
                        break;
                    }
                }
                frame_light = frame_light + light;
            }
            frame_light = average_samples(frame_light, p);
            if (VALIDATE) {
                cn.c[8]++;
                hash += fh;
            }
            sum = (fi == 1u) ? frame_light : sum + frame_light;
        }
    }
    finish_pixel<VALIDATE>(p, s.x, s.y, s.in_range, sum, first_prim, hash, rng, cn, none, 0u);
}

// --------------------------------------------------------------------------------------------
// Variant 1: one loop over segments; lanes regenerate their own next path.
// --------------------------------------------------------------------------------------------
template <bool VALIDATE, bool BVH, bool STORE = false, bool GLASS = false>
__global__ __launch_bounds__(kBlockThreads) void k_regenerate(const RenderParams p)
{
    RT_BVH_STACK(kBlockThreads);
    const PixelSetup s = pixel_setup(p);
    const Sky sky = make_sky(p);
    const DeviceScene &sc = p.scene;
    const uint32_t nS = p.sphere_number, nM = p.mesh_number;
    Counters cn, none;
    zero(cn);
    zero(none);
    uint32_t hash = 0u, first_prim = kMissId;
    uint32_t fh = frame_hash_init(p.first_frame);

    V3 sum = load_sum(p, s.x, s.y, s.in_range);

    // per-lane cursor: frame f (0-based within the launch), sample k, bounce i
    uint32_t f = 0, k = 0, i = 0;
    uint32_t rng = (uint32_t)s.lin + p.first_frame * 719393u;  // + k (= 0)
    V3 o = s.origin, d = s.dir;
    V3 light = mk(0.0f, 0.0f, 0.0f), thr = mk(1.0f, 1.0f, 1.0f);
    V3 frame_light = mk(0.0f, 0.0f, 0.0f);
    bool active = s.in_range && p.n_frames > 0u;
    if (VALIDATE && active) cn.c[9]++;

    while (active) {
        const Hit h = scene_hit<VALIDATE, BVH>(sc, nS, nM, o, d, cn, bvh_stack, bvh_stride);
        if (VALIDATE) {
            cn.c[10]++;
            fh = (fh ^ h.prim) * kHashPrime;
            if (i == 0 && k == 0) first_prim = h.prim;
        }
        bool path_done;
        const bool very_first = STORE && i == 0u && k == 0u && f == 0u;  // the pixel's primary segment
        if (h.prim != kMissId) {
            if (VALIDATE) {
                cn.c[6]++;
                if (h.prim < nS) cn.c[2]++;
                else cn.c[5]++;
            }
            if (very_first) {
                const size_t px = (size_t)s.y * p.width + s.x;
                p.primary[px] = make_float4(h.t, h.u, h.v, __uint_as_float(h.prim));
                p.primary_mat[px] = h.mat;
            }
            bounce<GLASS>(sc, nS, h, o, d, light, thr, rng);
            i++;
            path_done = i >= p.max_bounce;
        } else {
            if (VALIDATE) cn.c[7]++;
            light = add_mul(light, env_light(sky, d), thr);
            // a path whose primary ray leaves the scene: 0 + sky * 1, the same for every sample and frame
            if (very_first) p.primary[(size_t)s.y * p.width + s.x] = make_float4(light.x, light.y, light.z, __uint_as_float(kMissId));
            path_done = true;
        }
        if (path_done) {
            frame_light = frame_light + light;
            k++;
            if (k >= p.rays_per_pixel) {
                frame_light = average_samples(frame_light, p);
                if (VALIDATE) {
                    cn.c[8]++;
                    hash += fh;
                    fh = frame_hash_init(p.first_frame + f + 1u);
                }
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
    finish_pixel<VALIDATE>(p, s.x, s.y, s.in_range, sum, first_prim, hash, rng, cn, none, 0u);
}

// --------------------------------------------------------------------------------------------
// Variant 4, for scenes with a hierarchy: persistent waves that stream pixels.
//
// With a hierarchy, walks differ in length by orders of magnitude, and a kernel whose lanes are tied
// to one pixel each (k_regenerate) waits, segment after segment, for the longest walk of the wave;
// k_paths avoids that but needs several frames per launch to have spare work for its lanes.  Here a
// lane takes the next pixel from a launch-wide cursor whenever it has finished one (all frames and
// samples of a pixel stay in one lane, in the shader's order), walks are resumable as in k_paths
// (all walking lanes step until half of them are done, the finished ones shade and start their next
// ray), and the only idle lanes are those at the very end of the launch.  Pixels are handed out in
// 8 x 8 tiles.  p.stream_cursor is the cursor and must be zero at launch.
// --------------------------------------------------------------------------------------------
constexpr int kStreamThreads = 64;
// Waves per SIMD the kernel is compiled for.  5 is what the compiler picks by itself (96 registers).  6 (80
// registers, eight values in scratch) is 3 % faster on the 10^6 + 10^6 scene, whose walks wait for memory most
// of the time (one more wave per SIMD hides more of that than the spills cost), and 3 % slower on a scene of
// 10^4 primitives, whose hierarchy stays in L2 and whose walks are bound by instruction issue.  The launcher
// takes the second build when the nodes do not fit one XCD's 4 MB L2.
#ifndef RT_STREAM_MIN_WAVES
#define RT_STREAM_MIN_WAVES 5
#endif
#ifndef RT_STREAM_MIN_WAVES_LARGE
#define RT_STREAM_MIN_WAVES_LARGE 6
#endif
constexpr uint32_t kStreamLargeSceneNodes = (4u << 20) / 32u;

// RPP1: the launch has one sample per pixel and frame (the launcher checks): a finished path is a finished frame,
// and no sample counter or partial frame sum is carried.
template <bool VALIDATE, bool RPP1 = false, int WAVES = RT_STREAM_MIN_WAVES, bool GLASS = false>
__global__ __launch_bounds__(kStreamThreads, WAVES) void k_stream(const RenderParams p)
{
    constexpr bool BVH = true;
    RT_BVH_STACK(kStreamThreads);
    const uint32_t lane = threadIdx.x;
    const Sky sky = make_sky(p);
    const DeviceScene &sc = p.scene;
    const uint32_t nS = p.sphere_number, nM = p.mesh_number;
    const uint32_t tiles_x = (p.width + 7u) / 8u, tiles_y = (p.height + 7u) / 8u;
    const uint32_t n_items = tiles_x * tiles_y * 64u;
    Counters cn, none;
    zero(cn);
    zero(none);

    // the pixel this lane is working on
    bool has_pixel = false;
    uint32_t px_x = 0u, px_y = 0u;
    const V3 origin = mk(p.cam_pos[0], p.cam_pos[1], p.cam_pos[2]);  // the same for every pixel
    V3 sum = mk(0.0f, 0.0f, 0.0f);
    uint32_t hash = 0u, first_prim = kMissId, fh = 0u;
    // cursor within the pixel: frame f (0-based within the launch), sample k, bounce i
    uint32_t f = 0u, k = 0u, i = 0u, rng = 0u;
    // the pixel's primary direction is kept while the register budget allows; the 6-wave build recomputes it at
    // every path start instead (about a hundred instructions against the thousands of a walk)
    constexpr bool kKeepDir = WAVES <= 5;
    V3 px_dir = mk(0.0f, 0.0f, 0.0f);
    V3 o = origin, d = mk(0.0f, 0.0f, 0.0f);
    V3 light = mk(0.0f, 0.0f, 0.0f), thr = mk(1.0f, 1.0f, 1.0f), frame_light = mk(0.0f, 0.0f, 0.0f);
    BvhWalk walk;
    bvh_idle(walk);
    bool in_walk = false;
    bool exhausted = false;  // wave-uniform: the cursor has passed the last item
    // The shader casts the same primary ray for every sample of a pixel (Raytracing.comp:449-453): the
    // closest hit of the pixel's first path is kept and its later paths start from it without a walk.
    Hit primary;
    primary.t = 0.0f;
    primary.prim = kMissId;
    primary.mat = 0u;
    primary.u = 0.0f;
    primary.v = 0.0f;
    bool have_primary = false;

    for (;;) {
        // ---- lanes without a pixel take the next items
        const unsigned long long need = __ballot(!has_pixel);
        if (need != 0ull && !exhausted) {
            const uint32_t want = (uint32_t)__popcll(need);
            uint32_t first = 0u;
            if (lane == 0u) first = atomicAdd(p.stream_cursor, want);
            first = (uint32_t)__shfl((int)first, 0, 64);
            exhausted = first + want >= n_items || first + want < first;
            const uint32_t item = first + (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
            if (!has_pixel && item < n_items && item >= first) {
                const uint32_t tile = item >> 6, j = item & 63u;
                const uint32_t x = (tile % tiles_x) * 8u + (j & 7u), y = (tile / tiles_x) * 8u + (j >> 3);
                if (x < p.width && y < p.height) {
                    const PixelSetup s = pixel_setup_xy(p, x, y, true);
                    has_pixel = true;
                    px_x = x;
                    px_y = y;
                    if constexpr (kKeepDir) px_dir = s.dir;
                    sum = load_sum(p, x, y, true);
                    hash = 0u;
                    first_prim = kMissId;
                    fh = frame_hash_init(p.first_frame);
                    f = 0u;
                    k = 0u;
                    i = 0u;
                    rng = (uint32_t)s.lin + p.first_frame * 719393u;
                    o = s.origin;
                    d = s.dir;
                    light = mk(0.0f, 0.0f, 0.0f);
                    thr = mk(1.0f, 1.0f, 1.0f);
                    frame_light = mk(0.0f, 0.0f, 0.0f);
                    have_primary = false;
                    if (VALIDATE) cn.c[9]++;
                }
            }
        }
        if (__ballot(has_pixel) == 0ull) {
            if (exhausted) break;
            continue;  // only pixels outside the image so far
        }

        // ---- trace: every lane with a pixel holds a ray; walks are resumed where they stopped
        const bool from_primary = has_pixel && !in_walk && i == 0u && have_primary;
        if (has_pixel && !in_walk && !from_primary) {
            bvh_begin<VALIDATE>(walk, sc, nS, nM, o, d, cn);
            in_walk = true;
        }
        const uint32_t target = ((uint32_t)__popcll(__ballot(in_walk)) * RT_BVH_KEEP_NUM) >> RT_BVH_KEEP_SHIFT;
        for (;;) {
            const bool stepping = in_walk && walk.active();
            if ((uint32_t)__popcll(__ballot(stepping)) <= target) break;
            bvh_pass<VALIDATE>(stepping, walk, sc, nS, nM, o, d, cn, bvh_stack, bvh_stride);
        }

        // ---- lanes whose walk is over: the rest of Trace (Raytracing.comp:417-437) and of main (:464-480)
        const bool walked = in_walk && !walk.active();
        if (walked || from_primary) {
            in_walk = in_walk && !walked;
            const Hit h = walked ? walk.best : primary;
            if (walked && i == 0u) {
                primary = h;
                have_primary = true;
            }
            if (VALIDATE) {
                cn.c[10]++;
                fh = (fh ^ h.prim) * kHashPrime;
                if (i == 0u && k == 0u) first_prim = h.prim;
            }
            bool path_done;
            if (h.prim != kMissId) {
                if (VALIDATE) {
                    cn.c[6]++;
                    if (h.prim < nS) cn.c[2]++;
                    else cn.c[5]++;
                }
                bounce<GLASS>(sc, nS, h, o, d, light, thr, rng);
                i++;
                path_done = i >= p.max_bounce;
            } else {
                if (VALIDATE) cn.c[7]++;
                light = add_mul(light, env_light(sky, d), thr);
                path_done = true;
            }
            if (path_done) {
                bool more = true, frame_done = true;
                if constexpr (RPP1) {
                    frame_light = mk(0.0f, 0.0f, 0.0f) + light;   // incomingLight += Trace(), once (Raytracing.comp:464-468)
                } else {
                    frame_light = frame_light + light;
                    k++;
                    frame_done = k >= p.rays_per_pixel;
                }
                if (frame_done) {
                    frame_light = average_samples(frame_light, p);
                    if (VALIDATE) {
                        cn.c[8]++;
                        hash += fh;
                        fh = frame_hash_init(p.first_frame + f + 1u);
                    }
                    const uint32_t fi = p.first_frame + f;
                    sum = (fi == 1u) ? frame_light : sum + frame_light;
                    frame_light = mk(0.0f, 0.0f, 0.0f);
                    k = 0u;
                    f++;
                    more = f < p.n_frames;
                    // the float part of the seed, Raytracing.comp:462
                    if (more) rng = (uint32_t)((float)px_x + p.window_w * (float)px_y) + (p.first_frame + f) * 719393u;
                } else {
                    rng += k;
                }
                if (more) {
                    if (VALIDATE) cn.c[9]++;
                    o = origin;
                    if constexpr (kKeepDir) d = px_dir;
                    else d = pixel_setup_xy(p, px_x, px_y, true).dir;
                    light = mk(0.0f, 0.0f, 0.0f);
                    thr = mk(1.0f, 1.0f, 1.0f);
                    i = 0u;
                } else {
                    store_pixel<VALIDATE>(p, px_x, px_y, sum, first_prim, hash, rng);
                    has_pixel = false;
                }
            }
        }
    }
    // the wave's event counters
    finish_pixel<VALIDATE>(p, 0u, 0u, false, sum, 0u, 0u, 0u, cn, none, 0u);
}

// --------------------------------------------------------------------------------------------
// Variant 1 with a stored primary segment (static camera): one lane per pixel as in k_regenerate, but a
// path starts at its first bounce off the stored primary hit, and a pixel whose primary ray leaves the
// scene is finished in closed form.  Requires max_bounce >= 2 and rays_per_pixel >= 1.
// --------------------------------------------------------------------------------------------
template <bool BVH>
__global__ __launch_bounds__(kBlockThreads) void k_from_primary(const RenderParams p)
{
    RT_BVH_STACK(kBlockThreads);
    const PixelSetup s = pixel_setup(p);
    const Sky sky = make_sky(p);
    const DeviceScene &sc = p.scene;
    const uint32_t nS = p.sphere_number, nM = p.mesh_number;
    Counters cn, none;
    zero(cn);
    zero(none);

    const size_t px = s.in_range ? (size_t)s.y * p.width + s.x : 0;
    const float4 c = p.primary[px];
    const bool primary_hit = s.in_range && __float_as_uint(c.w) != kMissId;
    V3 sum = load_sum(p, s.x, s.y, s.in_range);

    if (s.in_range && !primary_hit) {
        const V3 light = mk(c.x, c.y, c.z);
        V3 frame_light = mk(0.0f, 0.0f, 0.0f);
        for (uint32_t k = 0; k < p.rays_per_pixel; k++) frame_light = frame_light + light;
        frame_light = average_samples(frame_light, p);
        for (uint32_t f = 0; f < p.n_frames; f++) sum = (p.first_frame + f == 1u) ? frame_light : sum + frame_light;
    }

    // what every path of this pixel starts from
    V3 own_pos = s.origin, own_nrm = mk(0.0f, 1.0f, 0.0f), own_color = mk(0.0f, 0.0f, 0.0f);
    float own_emit = 0.0f, own_smooth = 0.0f;
    if (primary_hit) {
        Hit h0;
        h0.t = c.x;
        h0.u = c.y;
        h0.v = c.z;
        h0.prim = __float_as_uint(c.w);
        h0.mat = p.primary_mat[px];
        resolve_hit(sc, nS, s.origin, s.dir, h0, own_pos, own_nrm);
        const float4 m0 = sc.materials[2 * h0.mat + 0];
        own_color = mk(m0.x, m0.y, m0.z);
        own_emit = m0.w;
        own_smooth = sc.materials[2 * h0.mat + 1].x;
    }
    const V3 own_spec = reflect(s.dir, own_nrm);

    uint32_t f = 0, k = 0, i = 0;
    uint32_t rng = (uint32_t)s.lin + p.first_frame * 719393u;
    V3 o = s.origin, d = s.dir;
    V3 light = mk(0.0f, 0.0f, 0.0f), thr = mk(1.0f, 1.0f, 1.0f), frame_light = mk(0.0f, 0.0f, 0.0f);
    bool active = primary_hit && p.n_frames > 0u;

    // the hit branch of Trace (Raytracing.comp:417-431) for the stored primary hit
    auto start_path = [&]() {
        const float rx = rng_next(rng) * 2.0f - 1.0f;
        const float ry = rng_next(rng) * 2.0f - 1.0f;
        const float rz = rng_next(rng) * 2.0f - 1.0f;
        const V3 diffuse = normalize(own_nrm + mk(rx, ry, rz));
        o = own_pos;
        d = gmix(diffuse, own_spec, own_smooth);
        const float e = 1.0f * own_emit;
        light = mk(0.0f, 0.0f, 0.0f) + mk(e * 1.0f, e * 1.0f, e * 1.0f);
        thr = mk(1.0f, 1.0f, 1.0f) * own_color;
        i = 1u;
    };
    if (active) start_path();

    while (active) {
        const Hit h = scene_hit<false, BVH>(sc, nS, nM, o, d, cn, bvh_stack, bvh_stride);
        bool path_done;
        if (h.prim != kMissId) {
            bounce(sc, nS, h, o, d, light, thr, rng);
            i++;
            path_done = i >= p.max_bounce;
        } else {
            light = add_mul(light, env_light(sky, d), thr);
            path_done = true;
        }
        if (path_done) {
            frame_light = frame_light + light;
            k++;
            if (k >= p.rays_per_pixel) {
                frame_light = average_samples(frame_light, p);
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
            if (active) start_path();
        }
    }
    finish_pixel<false>(p, s.x, s.y, s.in_range, sum, kMissId, 0u, rng, cn, none, 0u);
}

// --------------------------------------------------------------------------------------------
// Variant 2, first kernel: evaluate every pixel's primary segment once.  A pixel whose primary ray
// hits nothing gets the same radiance every frame (one sky evaluation, no random numbers), so it is
// finished here with the frame-by-frame additions the shader would have made.  Pixels that hit are
// appended to hit_list: each wave ballots its hit lanes, the workgroup reserves the slots of all its
// waves with a single atomic, and each hit lane takes the slot given by the count of hit lanes below it.
// --------------------------------------------------------------------------------------------
// The reservations are atomics on one address and limit the kernel: without a hierarchy (whose stacks would not
// fit the LDS otherwise) a workgroup has 1024 lanes, a quarter of the reservations.
constexpr int kClassifyThreadsLarge = 1024;
template <bool VALIDATE, bool BVH, bool STORE = false, int THREADS = kBlockThreads>
__global__ __launch_bounds__(THREADS) void k_classify(const RenderParams p)
{
    static_assert(!BVH || THREADS == kBlockThreads, "the traversal stacks are sized for 256 lanes");
    RT_BVH_STACK(THREADS);
    const PixelSetup s = pixel_setup_tile<THREADS>(p);
    const DeviceScene &sc = p.scene;
    const uint32_t nS = p.sphere_number, nM = p.mesh_number;
    Counters seg, none;
    zero(seg);
    zero(none);

    bool hit = false;
    if (s.in_range) {
        const Hit h0 = scene_hit<VALIDATE, BVH>(sc, nS, nM, s.origin, s.dir, seg, bvh_stack, bvh_stride);
        hit = h0.prim != kMissId;
        if (STORE && hit) {
            const size_t px = (size_t)s.y * p.width + s.x;
            p.primary[px] = make_float4(h0.t, h0.u, h0.v, __uint_as_float(h0.prim));
            p.primary_mat[px] = h0.mat;
        }
    }

    // one reservation per workgroup: the waves leave their counts in LDS, the first thread adds
    // the total to the list's counter (the atomics on that one address are what limits this kernel)
    // and every wave starts where the waves before it end
    __shared__ uint32_t wave_hits[THREADS / 64];
    __shared__ uint32_t block_base;
    const unsigned long long mask = __ballot(hit);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (lane == 0u) wave_hits[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0u) {
        uint32_t total = 0u;
        for (uint32_t w = 0; w < THREADS / 64; w++) total += wave_hits[w];
        block_base = total != 0u ? atomicAdd(p.hit_count, total) : 0u;
    }
    __syncthreads();
    if (hit) {
        uint32_t base = block_base;
        for (uint32_t w = 0; w < wave; w++) base += wave_hits[w];
        p.hit_list[base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = s.y * p.width + s.x;
    }

    const bool sky_pixel = s.in_range && !hit;
    V3 sum = load_sum(p, s.x, s.y, sky_pixel);
    uint32_t hash = 0u, rng = 0u;
    const uint32_t n_paths = p.n_frames * p.rays_per_pixel;
    if (sky_pixel) {
        const Sky sky = make_sky(p);
        // Trace(): incomingLight = 0 + env * rayColor(1,1,1), Raytracing.comp:410-411, 434
        const V3 light = add_mul(mk(0.0f, 0.0f, 0.0f), env_light(sky, s.dir), mk(1.0f, 1.0f, 1.0f));
        if (STORE) p.primary[(size_t)s.y * p.width + s.x] = make_float4(light.x, light.y, light.z, __uint_as_float(kMissId));
        V3 frame_light = mk(0.0f, 0.0f, 0.0f);
        for (uint32_t k = 0; k < p.rays_per_pixel; k++) frame_light = frame_light + light;
        frame_light = average_samples(frame_light, p);
        for (uint32_t f = 0; f < p.n_frames; f++) sum = (p.first_frame + f == 1u) ? frame_light : sum + frame_light;
        if (VALIDATE) {
            for (uint32_t f = 0; f < p.n_frames; f++) {
                uint32_t fh = frame_hash_init(p.first_frame + f);
                for (uint32_t k = 0; k < p.rays_per_pixel; k++) fh = (fh ^ kMissId) * kHashPrime;
                hash += fh;
            }
            rng = (uint32_t)s.lin + (p.first_frame + p.n_frames - 1u) * 719393u;
            for (uint32_t k = 0; k < p.rays_per_pixel; k++) rng += k;
            seg.c[7] = 1u;   // one sky evaluation,
            seg.c[9] = 1u;   // one path,
            seg.c[10] = 1u;  // one segment per path
            none.c[8] = p.n_frames;
        }
    }
    finish_pixel<VALIDATE>(p, s.x, s.y, sky_pixel, sum, kMissId, hash, rng, none, seg, n_paths);
}

// Variant 2, second kernel: wave-cooperative tracing of the compacted pixels.
#include "rt_paths.inc"

// Variant 2 with a stored primary segment: the pixels whose primary ray leaves the scene add the stored
// radiance, sample by sample and frame by frame as the shader would (Raytracing.comp:464-479).
__global__ __launch_bounds__(kBlockThreads) void k_sky_add(const RenderParams p)
{
    const uint32_t i = blockIdx.x * kBlockThreads + threadIdx.x;
    if (i >= p.width * p.height) return;
    const float4 c = p.primary[i];
    if (__float_as_uint(c.w) != kMissId) return;
    const V3 light = mk(c.x, c.y, c.z);   // one path's radiance
    V3 frame_light = mk(0.0f, 0.0f, 0.0f);
    for (uint32_t k = 0; k < p.rays_per_pixel; k++) frame_light = frame_light + light;
    frame_light = average_samples(frame_light, p);
    V3 sum = mk(0.0f, 0.0f, 0.0f);
    if (p.first_frame != 1u) {
        const float4 a = p.accum[i];
        sum = mk(a.x, a.y, a.z);
    }
    for (uint32_t f = 0; f < p.n_frames; f++) sum = (p.first_frame + f == 1u) ? frame_light : sum + frame_light;
    p.accum[i] = make_float4(sum.x, sum.y, sum.z, 1.0f);
    const float div = (float)(p.first_frame + p.n_frames - 1u);
    const uint32_t r = unorm8(sum.x / div), g = unorm8(sum.y / div), b = unorm8(sum.z / div);
    p.out_rgba8[i] = r | (g << 8) | (b << 16) | 0xFF000000u;
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

// Compares the short exact routines of rt_device.hpp with the general operations for EVERY bit pattern
// their range predicates admit (2^24 threads x 256 patterns).  out: {patterns tested, mismatches,
// smallest mismatching pattern}.
__global__ __launch_bounds__(kBlockThreads) void k_selftest_exhaustive(int op, float c, unsigned long long *__restrict__ out)
{
    const uint32_t tid = blockIdx.x * kBlockThreads + threadIdx.x;
    const float rc = 1.0f / c;
    unsigned long long tested = 0ull, bad = 0ull, first = ~0ull;
    for (uint32_t k = 0; k < 256u; k++) {
        const uint32_t bits = (k << 24) + tid;
        const float x = __uint_as_float(bits);
        bool in;
        float a, b;
        switch (op) {
        case 0: in = sqrt_rn_admits(x); a = sqrt_rn(x); b = sqrtf(x); break;
        case 1: in = rcp_rn_admits(x); a = rcp_rn(x); b = 1.0f / x; break;
        default: in = div_const_admits(x); a = div_const_rn(x, c, rc); b = x / c; break;
        }
        if (in) {
            tested++;
            if (__float_as_uint(a) != __float_as_uint(b)) {
                bad++;
                if ((unsigned long long)bits < first) first = bits;
            }
        }
    }
    tested = wave_sum(tested);
    bad = wave_sum(bad);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(first, off, 64);
        first = o < first ? o : first;
    }
    if ((threadIdx.x & 63u) == 0u) {
        atomicAdd(&out[0], tested);
        if (bad) {
            atomicAdd(&out[1], bad);
            atomicMin(&out[2], first);
        }
    }
}

// GetEnvironmentLight of n arbitrary directions (test hook): the per-lane routine, or the routine that
// shares the pow evaluations out over the wave (one wave per workgroup, as in k_paths; every third lane
// holds no direction, as lanes without an escaping path do there)
__global__ __launch_bounds__(64) void k_selftest_env(const RenderParams p, const float *__restrict__ dirs, size_t n, int wave_form,
                                                     float *__restrict__ rgb_out)
{
    __shared__ float mail[kSkyMail];
    const Sky sky = make_sky(p);
    const uint32_t lane = threadIdx.x;
    // 43 directions per workgroup in the wave form (lanes 2, 5, 8, ... 62 idle), 64 otherwise
    const bool idle = wave_form && lane % 3u == 2u;
    const size_t per_group = wave_form ? 43u : 64u;
    const size_t i = (size_t)blockIdx.x * per_group + (wave_form ? lane - lane / 3u : lane);
    const bool want = !idle && i < n;
    const V3 d = want ? mk(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]) : mk(0.3f, -0.005f, __uint_as_float(0x7FC00000u));
    V3 e;
    if (wave_form) e = env_light_wave(sky, d, want, lane, mail);
    else e = env_light(sky, d);
    if (want) {
        rgb_out[3 * i + 0] = e.x;
        rgb_out[3 * i + 1] = e.y;
        rgb_out[3 * i + 2] = e.z;
    }
}

// closest hit of n arbitrary rays against the uploaded scene (test hook)
template <bool BVH>
__global__ __launch_bounds__(kBlockThreads) void k_selftest_hit(DeviceScene sc, uint32_t n_spheres, uint32_t n_meshes,
                                                                const float *__restrict__ origins, const float *__restrict__ dirs,
                                                                size_t n, float *__restrict__ t_out, uint32_t *__restrict__ prim_out)
{
    RT_BVH_STACK(kBlockThreads);
    const size_t i = (size_t)blockIdx.x * kBlockThreads + threadIdx.x;
    if (i >= n) return;
    Counters cn;
    const Hit h = scene_hit<false, BVH>(sc, n_spheres, n_meshes, mk(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]),
                                        mk(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]), cn, bvh_stack, bvh_stride);
    t_out[i] = h.t;
    prim_out[i] = h.prim;
}

// ClosestHit of n arbitrary rays through the hierarchy the way the path kernels walk it: persistent waves, a
// lane takes ray after ray from a launch-wide cursor and resumes its walk (bvh_pass), the wave leaves the
// stepping loop when half of its walks are over.  Nothing but traversal: what the walks of a given set of
// rays cost in a given ORDER (measurement hook; same results as k_selftest_hit).  *cursor must be zero at launch.
__global__ __launch_bounds__(kStreamThreads, RT_STREAM_MIN_WAVES_LARGE) void k_selftest_walk(DeviceScene sc, uint32_t n_spheres, uint32_t n_meshes,
                                                                                           const float *__restrict__ origins,
                                                                                           const float *__restrict__ dirs, uint32_t n,
                                                                                           float *__restrict__ t_out, uint32_t *__restrict__ prim_out,
                                                                                           uint32_t *cursor)
{
    constexpr bool BVH = true;
    RT_BVH_STACK(kStreamThreads);
    const uint32_t lane = threadIdx.x;
    Counters cn;
    zero(cn);
    BvhWalk walk;
    bvh_idle(walk);
    bool has_ray = false, exhausted = false;
    uint32_t ray = 0u;
    V3 o = mk(0.0f, 0.0f, 0.0f), d = mk(0.0f, 0.0f, 1.0f);
    for (;;) {
        const unsigned long long need = __ballot(!has_ray);
        if (need != 0ull && !exhausted) {
            const uint32_t want = (uint32_t)__popcll(need);
            uint32_t first = 0u;
            if (lane == 0u) first = atomicAdd(cursor, want);
            first = (uint32_t)__shfl((int)first, 0, 64);
            exhausted = first + want >= n || first + want < first;
            const uint32_t item = first + (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
            if (!has_ray && item < n && item >= first) {
                ray = item;
                has_ray = true;
                o = mk(origins[3 * (size_t)item], origins[3 * (size_t)item + 1], origins[3 * (size_t)item + 2]);
                d = mk(dirs[3 * (size_t)item], dirs[3 * (size_t)item + 1], dirs[3 * (size_t)item + 2]);
                bvh_begin<false>(walk, sc, n_spheres, n_meshes, o, d, cn);
            }
        }
        const uint32_t holding = (uint32_t)__popcll(__ballot(has_ray));
        if (holding == 0u) {
            if (exhausted) break;
            continue;
        }
        const uint32_t target = (holding * RT_BVH_KEEP_NUM) >> RT_BVH_KEEP_SHIFT;
        for (;;) {
            const bool stepping = has_ray && walk.active();
            if ((uint32_t)__popcll(__ballot(stepping)) <= target) break;
            bvh_pass<false>(stepping, walk, sc, n_spheres, n_meshes, o, d, cn, bvh_stack, bvh_stride);
        }
        if (has_ray && !walk.active()) {
            t_out[ray] = walk.best.t;
            prim_out[ray] = walk.best.prim;
            has_ray = false;
        }
    }
}

}  // namespace

hipError_t RT_SYM(rt_launch_selftest_walk)(const DeviceScene &sc, uint32_t n_spheres, uint32_t n_meshes, const float *origins, const float *dirs,
                                           uint32_t n, float *t_out, uint32_t *prim_out, uint32_t *cursor, uint32_t n_cus, hipStream_t stream)
{
    const uint32_t cus = n_cus ? n_cus : 256u;
    const dim3 grid(cus * 4u * RT_STREAM_MIN_WAVES_LARGE, 1, 1);
    hipLaunchKernelGGL(k_selftest_walk, grid, dim3(kStreamThreads, 1, 1), 0, stream, sc, n_spheres, n_meshes, origins, dirs, n, t_out, prim_out,
                       cursor);
    return hipGetLastError();
}

hipError_t RT_SYM(rt_launch_selftest_hit)(const DeviceScene &sc, bool use_bvh, uint32_t n_spheres, uint32_t n_meshes, const float *origins,
                                  const float *dirs, size_t n, float *t_out, uint32_t *prim_out, hipStream_t stream)
{
    const dim3 grid((unsigned)((n + kBlockThreads - 1) / kBlockThreads), 1, 1);
    if (use_bvh)
        hipLaunchKernelGGL(k_selftest_hit<true>, grid, dim3(kBlockThreads, 1, 1), 0, stream, sc, n_spheres, n_meshes, origins,
                           dirs, n, t_out, prim_out);
    else
        hipLaunchKernelGGL(k_selftest_hit<false>, grid, dim3(kBlockThreads, 1, 1), 0, stream, sc, n_spheres, n_meshes, origins,
                           dirs, n, t_out, prim_out);
    return hipGetLastError();
}

hipError_t RT_SYM(rt_launch_selftest_env)(const RenderParams &p, const float *dirs, size_t n, int wave_form, float *rgb_out, hipStream_t stream)
{
    const size_t per_group = wave_form ? 43u : 64u;
    const dim3 grid((unsigned)((n + per_group - 1) / per_group), 1, 1);
    hipLaunchKernelGGL(k_selftest_env, grid, dim3(64, 1, 1), 0, stream, p, dirs, n, wave_form, rgb_out);
    return hipGetLastError();
}

const char *RT_SYM(rt_kernel_name)(int variant)
{
    switch (variant) {
    case 0: return "k_megakernel";
    case 1: return "k_regenerate";
    case 2: return "k_classify+k_paths";
    case 4: return "k_stream";
    default: return "unknown";
    }
}

// Workgroups of a k_paths launch: one single-wave workgroup per 64 list entries at least; the list's length is
// only known on the device, so the grid covers the worst case and surplus workgroups leave at once.  k_paths may
// deal a short list out 32, 16, 8 or 4 pixels to a wave: the grid allows for twice as many waves as 64 to a wave
// need, sixteen times as many for images up to 2^20 pixels, where surplus workgroups cost next to nothing.
static uint32_t paths_grid(uint32_t n_pixels, bool bvh)
{
    const uint32_t per_group = bvh ? 64u : (n_pixels <= (1u << 20) ? 4u : 32u);
    return (n_pixels + per_group - 1u) / per_group;
}

uint32_t RT_SYM(rt_paths_deal)(uint32_t n_pixels, uint32_t n_hit, uint32_t n_frames, uint32_t rays_per_pixel, uint32_t n_cus, bool meshes)
{
    return paths_deal(n_hit, n_frames, rays_per_pixel, n_cus, paths_grid(n_pixels, false), meshes);
}

hipError_t RT_SYM(rt_launch_render)(int variant, bool validate, const RenderParams &p, hipStream_t stream, RtCacheUse cache)
{
    if (p.width == 0 || p.height == 0 || p.n_frames == 0) return hipErrorInvalidValue;
    const dim3 grid((p.width + kTile - 1) / kTile, (p.height + kTile - 1) / kTile, 1);
    const dim3 block(kBlockThreads, 1, 1);
    // the flattened loops assume every path has at least one segment; the cached first bounce
    // assumes at least one more after it
    const bool bvh = p.use_bvh != 0u;
    // the streaming kernel exists for hierarchy scenes only, and its 32-bit cursor counts whole 8 x 8 tiles
    const uint64_t stream_items = (uint64_t)((p.width + 7u) / 8u) * ((p.height + 7u) / 8u) * 64u;
    if (variant == 4 && (!bvh || !p.stream_cursor || stream_items > 0xF0000000ull)) variant = 1;
    if (variant == 2 && (p.max_bounce < 2 || p.rays_per_pixel == 0)) variant = 1;
    if ((variant == 1 || variant == 4) && (p.max_bounce == 0 || p.rays_per_pixel == 0)) variant = 0;
#define RT_LAUNCH(KERNEL, GRID, BLOCK)                                                                    \
    do {                                                                                                  \
        if (validate && bvh) hipLaunchKernelGGL((KERNEL<true, true>), GRID, BLOCK, 0, stream, p);         \
        else if (validate) hipLaunchKernelGGL((KERNEL<true, false>), GRID, BLOCK, 0, stream, p);          \
        else if (bvh) hipLaunchKernelGGL((KERNEL<false, true>), GRID, BLOCK, 0, stream, p);               \
        else hipLaunchKernelGGL((KERNEL<false, false>), GRID, BLOCK, 0, stream, p);                       \
    } while (0)
    // Scenes with a transmitting material (an extension, rt_device.hpp bounce<GLASS>): the nested, the regenerating
    // and the streaming kernel have an instantiation for it; the compacting pair and the launches that start from
    // stored primary hits keep the shader's two-lobe material in their start records and are not used.
    const bool glass = p.scene.has_glass != 0u;
    if (glass) {
        if (cache != RT_CACHE_NONE) return hipErrorInvalidValue;
        if (variant == 2) variant = 1;
        if ((variant == 1 || variant == 4) && (p.max_bounce == 0 || p.rays_per_pixel == 0)) variant = 0;
        if (variant == 0) {
            if (validate && bvh) hipLaunchKernelGGL((k_megakernel<true, true, true>), grid, block, 0, stream, p);
            else if (validate) hipLaunchKernelGGL((k_megakernel<true, false, true>), grid, block, 0, stream, p);
            else if (bvh) hipLaunchKernelGGL((k_megakernel<false, true, true>), grid, block, 0, stream, p);
            else hipLaunchKernelGGL((k_megakernel<false, false, true>), grid, block, 0, stream, p);
            return hipGetLastError();
        }
        if (variant == 1) {
            if (validate && bvh) hipLaunchKernelGGL((k_regenerate<true, true, false, true>), grid, block, 0, stream, p);
            else if (validate) hipLaunchKernelGGL((k_regenerate<true, false, false, true>), grid, block, 0, stream, p);
            else if (bvh) hipLaunchKernelGGL((k_regenerate<false, true, false, true>), grid, block, 0, stream, p);
            else hipLaunchKernelGGL((k_regenerate<false, false, false, true>), grid, block, 0, stream, p);
            return hipGetLastError();
        }
    }
    if (variant == 0) {
        RT_LAUNCH(k_megakernel, grid, block);
    } else if (variant == 1) {
        if (cache != RT_CACHE_NONE && (validate || !p.primary || !p.primary_mat || p.max_bounce < 2)) return hipErrorInvalidValue;
        if (cache == RT_CACHE_LOAD) {
            if (bvh) hipLaunchKernelGGL((k_from_primary<true>), grid, block, 0, stream, p);
            else hipLaunchKernelGGL((k_from_primary<false>), grid, block, 0, stream, p);
        } else if (cache == RT_CACHE_STORE) {
            if (bvh) hipLaunchKernelGGL((k_regenerate<false, true, true>), grid, block, 0, stream, p);
            else hipLaunchKernelGGL((k_regenerate<false, false, true>), grid, block, 0, stream, p);
        } else {
            RT_LAUNCH(k_regenerate, grid, block);
        }
    } else if (variant == 2) {
        if (!p.hit_list || !p.hit_count) return hipErrorInvalidValue;
        if (cache != RT_CACHE_NONE && (validate || !p.primary || !p.primary_mat)) return hipErrorInvalidValue;
        const uint32_t n_pixels = p.width * p.height;
        const dim3 grid_paths(paths_grid(n_pixels, bvh), 1, 1);
        const dim3 block_paths(kPathsThreads, 1, 1);
        // k_paths for this scene: with a hierarchy, without meshes (no mesh code at all), or with meshes
        // (triangle tests shared by the lanes of a wave)
        const bool meshes = p.mesh_number != 0u;
        const bool rpp1 = p.rays_per_pixel == 1u;
        auto launch_paths = [&](auto validate_c, auto cached_c) {
            constexpr bool V = decltype(validate_c)::value, C = decltype(cached_c)::value;
            if (bvh) hipLaunchKernelGGL((k_paths<V, true, C, true>), grid_paths, block_paths, 0, stream, p);
            else if (meshes && rpp1) hipLaunchKernelGGL((k_paths<V, false, C, true, true>), grid_paths, block_paths, 0, stream, p);
            else if (meshes) hipLaunchKernelGGL((k_paths<V, false, C, true, false>), grid_paths, block_paths, 0, stream, p);
            else if (rpp1) hipLaunchKernelGGL((k_paths<V, false, C, false, true>), grid_paths, block_paths, 0, stream, p);
            else hipLaunchKernelGGL((k_paths<V, false, C, false, false>), grid_paths, block_paths, 0, stream, p);
        };
        if (cache == RT_CACHE_LOAD) {
            hipLaunchKernelGGL(k_sky_add, dim3((n_pixels + kBlockThreads - 1) / kBlockThreads, 1, 1), block, 0, stream, p);
            launch_paths(std::false_type{}, std::true_type{});
        } else {
            hipError_t e = hipMemsetAsync(p.hit_count, 0, sizeof(uint32_t), stream);
            if (e != hipSuccess) return e;
            const dim3 grid_large((p.width + 31u) / 32u, (p.height + 31u) / 32u, 1), block_large(kClassifyThreadsLarge, 1, 1);
            if (cache == RT_CACHE_STORE) {
                if (bvh) hipLaunchKernelGGL((k_classify<false, true, true>), grid, block, 0, stream, p);
                else hipLaunchKernelGGL((k_classify<false, false, true, kClassifyThreadsLarge>), grid_large, block_large, 0, stream, p);
            } else if (validate && bvh) {
                hipLaunchKernelGGL((k_classify<true, true>), grid, block, 0, stream, p);
            } else if (validate) {
                hipLaunchKernelGGL((k_classify<true, false, false, kClassifyThreadsLarge>), grid_large, block_large, 0, stream, p);
            } else if (bvh) {
                hipLaunchKernelGGL((k_classify<false, true>), grid, block, 0, stream, p);
            } else {
                hipLaunchKernelGGL((k_classify<false, false, false, kClassifyThreadsLarge>), grid_large, block_large, 0, stream, p);
            }
            if (validate) launch_paths(std::true_type{}, std::false_type{});
            else launch_paths(std::false_type{}, std::false_type{});
        }
#undef RT_LAUNCH
    } else if (variant == 4) {
        if (cache != RT_CACHE_NONE) return hipErrorInvalidValue;
        hipError_t e = hipMemsetAsync(p.stream_cursor, 0, sizeof(uint32_t), stream);
        if (e != hipSuccess) return e;
        // as many single-wave workgroups as the device keeps resident (more would only queue up)
        int device = 0, cus = (int)p.n_cus;
        if (cus == 0) {
            if ((e = hipGetDevice(&device)) != hipSuccess) return e;
            if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device)) != hipSuccess) return e;
        }
        const uint32_t n_tiles = ((p.width + 7u) / 8u) * ((p.height + 7u) / 8u);
        const bool large = !validate && p.scene.bvh_node_count > kStreamLargeSceneNodes;
        const uint32_t waves_per_simd = large ? RT_STREAM_MIN_WAVES_LARGE : RT_STREAM_MIN_WAVES;
        const uint32_t resident = (uint32_t)(cus > 0 ? cus : 1) * 4u * waves_per_simd;
        const dim3 grid_stream(n_tiles < resident ? n_tiles : resident, 1, 1);
#ifdef RT_STREAM_NO_RPP1
        const bool one_sample = false;
#else
        const bool one_sample = p.rays_per_pixel == 1u;
#endif
        if (glass && validate) hipLaunchKernelGGL((k_stream<true, false, RT_STREAM_MIN_WAVES, true>), grid_stream, dim3(kStreamThreads, 1, 1), 0, stream, p);
        else if (glass && large) hipLaunchKernelGGL((k_stream<false, false, RT_STREAM_MIN_WAVES_LARGE, true>), grid_stream, dim3(kStreamThreads, 1, 1), 0, stream, p);
        else if (glass) hipLaunchKernelGGL((k_stream<false, false, RT_STREAM_MIN_WAVES, true>), grid_stream, dim3(kStreamThreads, 1, 1), 0, stream, p);
        else if (validate && one_sample) hipLaunchKernelGGL((k_stream<true, true>), grid_stream, dim3(kStreamThreads, 1, 1), 0, stream, p);
        else if (validate) hipLaunchKernelGGL((k_stream<true, false>), grid_stream, dim3(kStreamThreads, 1, 1), 0, stream, p);
        else if (large && one_sample)
            hipLaunchKernelGGL((k_stream<false, true, RT_STREAM_MIN_WAVES_LARGE>), grid_stream, dim3(kStreamThreads, 1, 1), 0, stream, p);
        else if (large)
            hipLaunchKernelGGL((k_stream<false, false, RT_STREAM_MIN_WAVES_LARGE>), grid_stream, dim3(kStreamThreads, 1, 1), 0, stream, p);
        else if (one_sample) hipLaunchKernelGGL((k_stream<false, true>), grid_stream, dim3(kStreamThreads, 1, 1), 0, stream, p);
        else hipLaunchKernelGGL((k_stream<false, false>), grid_stream, dim3(kStreamThreads, 1, 1), 0, stream, p);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t RT_SYM(rt_launch_resolve)(float4 *accum, uint32_t *out, uint32_t n_pixels, float divisor, hipStream_t stream)
{
    const dim3 grid((n_pixels + kBlockThreads - 1) / kBlockThreads, 1, 1);
    hipLaunchKernelGGL(k_resolve, grid, dim3(kBlockThreads, 1, 1), 0, stream, accum, out, n_pixels, divisor);
    return hipGetLastError();
}

hipError_t RT_SYM(rt_launch_selftest_exhaustive)(int op, float c, unsigned long long *out, hipStream_t stream)
{
    hipLaunchKernelGGL(k_selftest_exhaustive, dim3((1u << 24) / kBlockThreads, 1, 1), dim3(kBlockThreads, 1, 1), 0, stream, op, c, out);
    return hipGetLastError();
}

hipError_t RT_SYM(rt_launch_selftest)(int op, const float *x, const float *y, float *out, size_t n, hipStream_t stream)
{
    const dim3 grid((unsigned)((n + kBlockThreads - 1) / kBlockThreads), 1, 1);
    hipLaunchKernelGGL(k_selftest, grid, dim3(kBlockThreads, 1, 1), 0, stream, op, x, y, out, n);
    return hipGetLastError();
}
