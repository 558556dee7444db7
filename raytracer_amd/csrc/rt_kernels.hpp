// rt_kernels.hpp — launch interface between the C ABI layer and the gfx950 kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_device.hpp"

// Everything one dispatch needs, passed by value in the kernel argument segment (scalar loads).
// The first block mirrors the uniform block of Raytracing.comp:3-24.
struct RenderParams {
    float inv_proj[16];
    float inv_view[16];
    float cam_pos[3];
    float window_w, window_h;  // frameData.window as floats
    uint32_t width, height;    // image extent guarded by the kernels
    uint32_t rays_per_pixel, max_bounce;
    float sky_horizon[3], sky_zenith[3], sun_dir[3], ground[3];
    float sun_focus, sun_intensity;
    uint32_t sphere_number, mesh_number;
    uint32_t first_frame, n_frames;  // frame_index runs first_frame .. first_frame + n_frames - 1
    uint32_t rpp_is_pow2;
    float rpp_recip;

    rtd::DeviceScene scene;
    uint32_t use_bvh;       // closest hit through scene.bvh_* instead of the uniform loops

    float4 *accum;          // [height][width] running sum, w = 1
    uint32_t *out_rgba8;    // [height][width] packed r | g<<8 | b<<16 | a<<24
    uint32_t *hit_ids;      // validation only
    uint32_t *chain_hash;   // validation only
    uint32_t *rng_state;    // validation only
    unsigned long long *counters;  // validation only, 16 entries
    uint32_t *hit_list;     // variant 2: pixels whose primary ray hits something, compacted
    uint32_t *hit_count;    // variant 2: length of hit_list, zeroed before every launch that builds the list
    uint32_t *stream_cursor;  // variant 4: next pixel item to hand out, zeroed before every launch (a word of its own:
                              // a stored work list keeps hit_count alive between launches)
    uint32_t n_cus;         // compute units of the device (variant 4 sizes its grid by it); 0 = ask the runtime
    // Primary-segment cache (variant 2, static camera): per pixel {t, u, v, bits(prim)} of the primary
    // hit, or {r, g, b, bits(miss id)} = the radiance a sky pixel adds per frame, and the material row.
    float4 *primary;
    uint32_t *primary_mat;
};

// What a variant-2 launch does with the primary-segment cache
enum RtCacheUse {
    RT_CACHE_NONE = 0,   // evaluate every pixel's primary segment, keep nothing
    RT_CACHE_STORE = 1,  // evaluate it and store it (also leaves hit_list / hit_count for later launches)
    RT_CACHE_LOAD = 2    // no classification: sky pixels add their stored radiance, paths start from the stored hits
};

// The launchers exist once per arithmetic profile: rt_kernels.hip is compiled twice, plainly (strict profile,
// names as written) and with -DRT_PROFILE_FMA (fused profile, names with the suffix _fma).
#define RT_DECLARE_LAUNCHERS(S)                                                                                                       \
    const char *rt_kernel_name##S(int variant);                                                                                       \
    hipError_t rt_launch_render##S(int variant, bool validate, const RenderParams &p, hipStream_t stream, RtCacheUse cache);          \
    uint32_t rt_paths_deal##S(uint32_t n_pixels, uint32_t n_hit, uint32_t n_frames, uint32_t rays_per_pixel, uint32_t n_cus,          \
                              bool meshes);                                                                                          \
    hipError_t rt_launch_resolve##S(float4 *accum, uint32_t *out, uint32_t n_pixels, float divisor, hipStream_t stream);              \
    hipError_t rt_launch_selftest_hit##S(const rtd::DeviceScene &sc, bool use_bvh, uint32_t n_spheres, uint32_t n_meshes,             \
                                         const float *origins, const float *dirs, size_t n, float *t_out, uint32_t *prim_out,         \
                                         hipStream_t stream);                                                                         \
    hipError_t rt_launch_selftest_walk##S(const rtd::DeviceScene &sc, uint32_t n_spheres, uint32_t n_meshes, const float *origins,          \
                                          const float *dirs, uint32_t n, float *t_out, uint32_t *prim_out, uint32_t *cursor,            \
                                          uint32_t n_cus, hipStream_t stream);                                                         \
    hipError_t rt_launch_selftest_env##S(const RenderParams &p, const float *dirs, size_t n, int wave_form, float *rgb_out,           \
                                         hipStream_t stream);                                                                         \
    hipError_t rt_launch_selftest_exhaustive##S(int op, float c, unsigned long long *out, hipStream_t stream);                        \
    hipError_t rt_launch_selftest##S(int op, const float *x, const float *y, float *out, size_t n, hipStream_t stream);
RT_DECLARE_LAUNCHERS()
RT_DECLARE_LAUNCHERS(_fma)
#undef RT_DECLARE_LAUNCHERS

#ifdef RT_PROFILE_FMA
#define RT_SYM(name) name##_fma
#else
#define RT_SYM(name) name
#endif
