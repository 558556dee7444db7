/*
 * rtmi.h — C ABI of the MI355X path-tracing library (librtmi.so).
 *
 * The library replaces exactly one thing in VoidRune/Raytracer: the compute dispatch of
 * VulkanRaytracer/res/Shaders/Raytracing.comp together with the buffers bound to it
 * (descriptor set bindings 0..5, main.cpp:89-96, 201-208, dispatch at main.cpp:374-376).
 * Scene records and the per-frame uniform block keep the reference's byte layouts so a caller can
 * hand over the very arrays it would have memcpy'd into Vulkan buffers.
 *
 * Conventions: every function returns 0 on success or a negative rtmi_status; nothing throws
 * across this boundary; a context is driven from one thread at a time (the reference has a single
 * render thread and fully serialised frames, Renderer.cpp:20-21, 66-71); separate contexts, e.g.
 * one per GPU, are independent.  Inputs are copied during the call, outputs are copied into
 * caller-provided buffers.
 */
#ifndef RTMI_H
#define RTMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define RTMI_API __attribute__((visibility("default")))
#else
#define RTMI_API
#endif

/* ------------------------------------------------------------------------------------------
 * Records.  Layouts are those of RayTracingStructs.h:4-60 / the std140 blocks of
 * Raytracing.comp:3-72 (sizes 260 / 32 / 48 / 80 / 96 bytes).
 * ---------------------------------------------------------------------------------------- */
typedef struct rtmi_frame_data {
    float    camera_inverse_projection[16]; /* column-major                    (RT.comp:5)  */
    float    camera_inverse_view[16];       /*                                 (RT.comp:6)  */
    float    camera_pos[4];                 /*                                 (RT.comp:7)  */
    float    camera_direction[4];           /* carried, never read by the path (RT.comp:8)  */
    float    window[2];                     /* width, height as floats         (RT.comp:9)  */
    uint32_t rays_per_pixel;                /*                                 (RT.comp:11) */
    uint32_t max_bounce_limit;              /* "maxBouceLimit"                 (RT.comp:12) */
    float    sky_color_horizon[4];
    float    sky_color_zenith[4];
    float    sun_light_direction[4];        /* used swizzled .xzy              (RT.comp:394) */
    float    ground_color[4];
    float    sun_focus;
    float    sun_intensity;
    uint32_t frame_index;                   /* 1 = overwrite the accumulation  (RT.comp:473) */
    uint32_t sphere_number;
    uint32_t mesh_number;
} rtmi_frame_data;

typedef struct rtmi_material {
    float color[3];
    float light;
    float smoothness;
    float pad1, pad2, pad3;
} rtmi_material;

typedef struct rtmi_sphere {
    float center[3];
    float radius;
    rtmi_material material;
} rtmi_sphere;

typedef struct rtmi_mesh {
    int32_t start_triangle;
    int32_t num_triangles;
    float   pad1, pad2;
    float   bounding_point1[4];   /* box minimum */
    float   bounding_point2[4];   /* box maximum */
    rtmi_material material;
} rtmi_mesh;

typedef struct rtmi_triangle {
    float p1[4], p2[4], p3[4];
    float n1[4], n2[4], n3[4];
} rtmi_triangle;

typedef struct rtmi_ctx rtmi_ctx;

typedef enum rtmi_status {
    RTMI_OK = 0,
    RTMI_ERR_INVALID_ARGUMENT = -1,
    RTMI_ERR_NO_DEVICE = -2,       /* no usable HIP device / kernels not loadable */
    RTMI_ERR_HIP = -3,             /* a HIP runtime call failed, see rtmi_last_error */
    RTMI_ERR_OUT_OF_MEMORY = -4,
    RTMI_ERR_NO_SCENE = -5,        /* render before rtmi_upload_scene */
    RTMI_ERR_SCENE_MISMATCH = -6,  /* frame data counts exceed the uploaded scene */
    RTMI_ERR_IO = -7,
    RTMI_ERR_UNSUPPORTED = -8
} rtmi_status;

/* Kernel variants.  All produce bit-identical results. */
typedef enum rtmi_variant {
    RTMI_VARIANT_MEGAKERNEL = 0,   /* one lane per pixel, frame / sample / bounce loops nested as in RT.comp:464-468, 413 */
    RTMI_VARIANT_REGENERATE = 1,   /* one lane per pixel, single segment loop; a lane that ends a path starts its next one */
    RTMI_VARIANT_COMPACT    = 2,   /* the primary segment (identical for every sample of a pixel, RT.comp:449-453)
                                      is evaluated once per launch; pixels that see only sky are finished at once, the
                                      others are stream-compacted (wave ballot + prefix count) and traced by dense waves */
    RTMI_VARIANT_AUTO       = 3,   /* default: per launch, COMPACT when the launch holds at least 8 frames (4 for images
                                      up to 2^21 pixels), REGENERATE otherwise: COMPACT hands out work by the frame (the
                                      samples of a frame share a random number stream and stay with one lane), so a launch
                                      of few frames, the reference's own calling pattern, leaves its lanes idle however
                                      many rays per pixel a frame has; with a hierarchy, STREAM.
                                      While the uniform block (apart from frame_index) and the scene stay the same
                                      from launch to launch, AUTO also keeps every pixel's primary segment and lets
                                      later launches start from it; results are the same bits either way */
    RTMI_VARIANT_STREAM     = 4    /* scenes with a hierarchy: persistent waves whose lanes take pixel after pixel from
                                      a launch-wide cursor and resume their walks, so that no lane waits for the
                                      longest walk of its wave (COMPACT needs several frames per launch to keep
                                      its lanes fed).  Without a hierarchy it runs as REGENERATE */
} rtmi_variant;

/* How ClosestHit (RT.comp:338-388) finds its candidates.  The reference loops over every sphere and
 * every triangle of every mesh whose box the ray meets; that is also what RTMI_ACCEL_NONE does.  With
 * a bounding volume hierarchy (built on the host at upload: one over the spheres, one per mesh) only
 * the primitives near the ray get the shader's tests.  The hit returned is the same: the tests are
 * the shader's own, the box tests carry margins for their rounding, and ties keep the shader's loop
 * order.  The margins follow the error bounds of the two tests.  For a sphere, and for a triangle seen by a ray
 * that is not almost parallel to it, a box grown by 2^-12 of its distance from the ray origin is enough.  A ray
 * that runs almost inside a triangle's plane is different: RayTriangle (RT.comp:286-309) divides by
 * det = -dot(d, n), and near det = 1e-6 the rounding of the numerators can make it accept a ray that passes the
 * triangle at a distance, the further the larger the triangle and the flatter the ray.  Every mesh therefore
 * carries the cone of its normals and the size of its largest triangle (and every node of its hierarchy the
 * same for each child).  A ray outside the cone's band keeps the usual margin.  Inside the band: below a
 * large triangle (edges of about half a unit and more) the node is visited whatever its box says, which is
 * what the error bound demands; boxes of small triangles are grown by the bound itself, at most by the
 * fraction of the distance set with rtmi_set_grazing_cap (default 2^-9).  With the cap at 1 the hierarchy
 * provably returns the hit of the exhaustive loops for every ray; with a smaller cap that holds for every ray
 * except those within about 1.2e-6 G / (|n| cap) radians of the plane of a small triangle (G / |n|, a measure of
 * its aspect ratio, is 3.5 for an equilateral triangle: 0.12 degrees at the default) that they pass at more than
 * cap times their distance from it.  Probes of several million rays aimed along
 * triangle planes found no difference at any cap, 0 included (tools/grazing_rate.py).
 * RTMI_ACCEL_AUTO builds hierarchies when the scene holds more than 256 primitives.  The
 * reference has no counterpart (its scene holds 1 sphere and 14 triangles in 2 meshes, main.cpp:161-185). */
typedef enum rtmi_accel {
    RTMI_ACCEL_AUTO = 0,
    RTMI_ACCEL_NONE = 1,
    RTMI_ACCEL_BVH  = 2
} rtmi_accel;

/* Arithmetic profile.  STRICT (default): every multiply and add of the shader is one rounded binary32
 * operation, in the order the shader writes them: what a SPIR-V consumer that does not contract computes.
 * FMA: the same sequences with a fused multiply-add (one rounding) at a fixed list of sites: the sums of
 * products of dot, cross, mix, reflect and the matrix-vector products of RT.comp:448-453, the Horner steps
 * of pow, b*b - 4ac of RaySphere, origin + direction * t, the blend of the shading normals, and the
 * light / throughput updates of Trace.  RT.comp carries no `precise` qualifier, so a driver's compiler may
 * contract it this way.  Each profile is deterministic and has its own CPU oracle build; images of the two
 * differ by rounding only (a hit decided by the last bit may flip).  FMA is 7-8 % faster.
 * The reference has no counterpart. */
typedef enum rtmi_profile {
    RTMI_PROFILE_STRICT = 0,
    RTMI_PROFILE_FMA    = 1
} rtmi_profile;

/* Event counters (uint64 each), index = rtmi_counter.  Same events the CPU oracle counts.  With a
 * hierarchy in use, the three test counters count the tests actually made (box tests include the
 * hierarchy's node boxes), which are fewer than the oracle's exhaustive loops make. */
typedef enum rtmi_counter {
    RTMI_N_SPHERE_TESTS = 0,
    RTMI_N_SPHERE_DISC_GE0,
    RTMI_N_SPHERE_WINS,
    RTMI_N_BOX_TESTS,
    RTMI_N_TRI_TESTS,
    RTMI_N_TRI_WINS,
    RTMI_N_HIT_BOUNCES,
    RTMI_N_ENV_EVALS,
    RTMI_N_PIXEL_FRAMES,
    RTMI_N_PATHS,
    RTMI_N_SEGMENTS,
    RTMI_N_COUNTERS = 16
} rtmi_counter;

#define RTMI_MISS_ID 0xFFFFFFFFu

/* ------------------------------------------------------------------------------------------
 * Context.  Replaces Core/Renderer device selection + the two storage images created at
 * main.cpp:74-82 (rgba8 output, rgba32f accumulation), both zero-initialised.
 * ---------------------------------------------------------------------------------------- */
RTMI_API int rtmi_create(int device, uint32_t width, uint32_t height, rtmi_ctx **out);
RTMI_API int rtmi_destroy(rtmi_ctx *ctx);
/* Message of the last failure on ctx (or of the last failed rtmi_create when ctx is NULL). */
RTMI_API const char *rtmi_last_error(const rtmi_ctx *ctx);
RTMI_API const char *rtmi_version(void);

/* Replaces the three storage-buffer uploads at main.cpp:171-200 (Buffer::Map + memcpy of
 * Sphere[], Triangle[], Mesh[]).  Takes the reference's AoS records and repacks them on the way to
 * the device.  May be called again to change the scene (main.cpp:296-298). */
RTMI_API int rtmi_upload_scene(rtmi_ctx *ctx,
                               const rtmi_sphere *spheres, uint32_t n_spheres,
                               const rtmi_triangle *triangles, uint32_t n_triangles,
                               const rtmi_mesh *meshes, uint32_t n_meshes);

/* Replaces "memcpy the uniform block + vkCmdDispatch" (main.cpp:370-376): one frame with
 * fd->frame_index, including the in-shader overwrite when frame_index == 1 (RT.comp:473-477). */
RTMI_API int rtmi_render(rtmi_ctx *ctx, const rtmi_frame_data *fd);

/* Same result as calling rtmi_render n_frames times with frame_index = first_frame,
 * first_frame+1, ... and otherwise unchanged *fd (fd->frame_index itself is ignored), but in one
 * launch with the running sum kept in registers.  No counterpart call in the reference; it is the
 * reference frame loop (main.cpp:236-444) with a static camera.  At most 2^24 frames per call, and
 * first_frame + n_frames - 1 must fit the shader's 32-bit frame index (RTMI_ERR_INVALID_ARGUMENT otherwise). */
RTMI_API int rtmi_render_frames(rtmi_ctx *ctx, const rtmi_frame_data *fd,
                                uint32_t first_frame, uint32_t n_frames);

/* rtmi_render_frames for callers that cannot allocate device memory: code under HIP graph capture,
 * tight serving loops, harnesses that forbid it.  No counterpart in the reference.  Same images, bit
 * for bit, as rtmi_render_frames; it calls no hipMalloc / hipFree / hipEventCreate and records no
 * timing events (rtmi_last_render_ms then has nothing to report).  The kernels' work list and counters
 * live in a static device array inside the library (4 bytes per pixel + 256 bytes, held by every GPU
 * the library runs on until the process ends) instead of buffers of the context, so the image may
 * hold at most RTMI_NOALLOC_MAX_PIXELS pixels (RTMI_ERR_UNSUPPORTED beyond that, and with validation
 * outputs enabled).  All contexts on one GPU share that array.  The library orders its users: every
 * no-allocation launch records an event of its context (created with the context), and the next one on that
 * GPU, if it runs on another stream, waits for it on the device first; so contexts on their own streams may
 * call in any order without synchronising, and simply take turns.  (A caller that puts several contexts on
 * the SAME stream handle it passed in gets the stream's own order.  A call on a stream in capture mode is
 * recorded into the graph without these events: when the graph runs relative to other users of the work area is
 * for its launcher to arrange.)  The kernel choice is that of
 * rtmi_set_variant, except that primary segments are never kept between launches. */
#define RTMI_NOALLOC_MAX_PIXELS 2097152u   /* 2^21: covers 1920 x 1080; 8 MiB of device memory per GPU */
RTMI_API int rtmi_render_frames_noalloc(rtmi_ctx *ctx, const rtmi_frame_data *fd,
                                        uint32_t first_frame, uint32_t n_frames);

RTMI_API int rtmi_set_variant(rtmi_ctx *ctx, int variant);
RTMI_API int rtmi_get_variant(const rtmi_ctx *ctx);
/* Mode for the NEXT rtmi_upload_scene (default RTMI_ACCEL_AUTO); rtmi_get_acceleration tells what
 * the uploaded scene uses: RTMI_ACCEL_NONE or RTMI_ACCEL_BVH. */
/* Bytes of one hierarchy node as the kernels fetch it (both child boxes and references): what a measurement
 * harness prices a node visit at. */
RTMI_API int rtmi_bvh_node_bytes(void);
RTMI_API int rtmi_set_acceleration(rtmi_ctx *ctx, int mode);
RTMI_API int rtmi_get_acceleration(const rtmi_ctx *ctx);
/* Largest extra margin, as a fraction of the distance from the ray origin, that the boxes of a mesh of small
 * triangles are given for a ray that runs almost inside a triangle's plane (see rtmi_accel above).  0 = the
 * usual margin only; 1 = the full error bound, whatever it costs (on a mesh of a million small triangles:
 * every node).  A kernel argument: it applies from the next launch on, no new upload needed.  Cost on the
 * 10^6 + 10^6 primitive stress scene, ms per 16 frames at 1080p: 0 -> 148, 2^-9 -> 155, 2^-8 -> 160,
 * 2^-7 -> 187.  No counterpart in the reference. */
#define RTMI_GRAZING_CAP_DEFAULT 0.001953125f   /* 2^-9 */
RTMI_API int rtmi_set_grazing_cap(rtmi_ctx *ctx, float cap);
RTMI_API float rtmi_get_grazing_cap(const rtmi_ctx *ctx);
/* EXTENSION without a counterpart in the reference, off by default; applies to the NEXT rtmi_upload_scene.
 * The reference's Material (RT.comp:26-34, RayTracingStructs.h:24-32) has colour, emission and smoothness; its
 * three padding words are never read.  With this switched on, a material whose pad1 is above 0.5 transmits, with
 * pad2 as its index of refraction: a ray that hits it is mirrored with Schlick's reflectance as probability and
 * refracted into the object otherwise (every hit that RaySphere / RayTriangle report is from outside, so the ray
 * always enters; the tests see no inside, so it then leaves unseen).  The branch draws the same three random
 * numbers as the diffuse / specular one, so all other paths of an image keep their random streams.  There is no
 * parity target for it: the kernels agree bit for bit with the CPU oracle's statement of the same rule, nothing
 * more.  Scenes with such a material are rendered by the REGENERATE and STREAM kernels (COMPACT falls back).
 * Switched off, the padding is ignored whatever it holds. */
RTMI_API int rtmi_enable_transmission(rtmi_ctx *ctx, int on);
/* Profile for the NEXT rtmi_upload_scene (default RTMI_PROFILE_STRICT): the upload precomputes each
 * triangle's AB x AC in the profile's arithmetic, and launches use the kernels of the profile the scene
 * was uploaded under.  rtmi_get_profile tells which that is (before any upload: the mode set).  The
 * rtmi_selftest_math / _env_light hooks follow the mode set, rtmi_selftest_closest_hit the uploaded scene. */
RTMI_API int rtmi_set_profile(rtmi_ctx *ctx, int profile);
RTMI_API int rtmi_get_profile(const rtmi_ctx *ctx);

/* Accumulation image (binding 5, rgba32f, running SUM not mean; RT.comp:471-479). */
RTMI_API int rtmi_clear_accum(rtmi_ctx *ctx);
RTMI_API int rtmi_read_accum(rtmi_ctx *ctx, float *rgba, size_t bytes);
RTMI_API int rtmi_write_accum(rtmi_ctx *ctx, const float *rgba, size_t bytes);
/* Display image (binding 4, rgba8: sum / frame_index of the last rendered frame; RT.comp:480). */
RTMI_API int rtmi_read_output_rgba8(rtmi_ctx *ctx, uint8_t *rgba, size_t bytes);
/* Recompute the display image as sum / divisor and put the sum's alpha back to 1 (used after summing
 * the partial images of several GPUs, whose alphas add up). */
RTMI_API int rtmi_resolve_output(rtmi_ctx *ctx, uint32_t divisor);

/* Snapshot of the accumulation image plus the number of frames summed into it, and its reload into a
 * context of the same extent (which also rebuilds the display image as sum / frames_done).  Going on
 * with frame_index = frames_done + 1 gives the bits an uninterrupted run gives.  The reference keeps
 * its accumulation image in device memory only (main.cpp:78-82) and cannot resume. */
RTMI_API int rtmi_checkpoint_save(rtmi_ctx *ctx, const char *path, uint32_t frames_done);
RTMI_API int rtmi_checkpoint_load(rtmi_ctx *ctx, const char *path, uint32_t *frames_done);

/* Use a caller-owned device buffer of width*height float4 as the accumulation image (for example
 * memory that a collective library reduces in place).  NULL returns to the internal buffer.  The
 * shader has no counterpart; in Vulkan terms this is rebinding descriptor 5. */
RTMI_API int rtmi_bind_accum(rtmi_ctx *ctx, void *device_float4);
RTMI_API void *rtmi_accum_device_ptr(rtmi_ctx *ctx);
/* Launch on a caller-owned hipStream_t.  NULL returns to the context's own (non-blocking) stream; it
 * does NOT mean HIP's null stream, which is named by the runtime's hipStreamLegacy handle. */
RTMI_API int rtmi_set_stream(rtmi_ctx *ctx, void *hip_stream);
RTMI_API int rtmi_synchronize(rtmi_ctx *ctx);

/* Several GPUs in one process (no counterpart in the reference, which drives one GPU): the number of HIP
 * devices, and the sum of the accumulation images of n contexts, one per GPU and all of one extent, onto
 * contexts[root] by one RCCL reduce over xGMI, enqueued on every context's stream without a host
 * synchronisation.  Each context renders a disjoint frame range from a cleared image (rtmi_clear_accum,
 * rtmi_render_frames; a frame's contribution depends only on pixel, frame index, scene and camera,
 * RT.comp:462); after the sum, rtmi_resolve_output(root, total_frames) restores alpha and rebuilds the
 * display image.  RCCL is opened at run time, and only when n > 1. */
RTMI_API int rtmi_device_count(void);
RTMI_API int rtmi_reduce_accum(rtmi_ctx *const *contexts, uint32_t n, uint32_t root);
/* The collective library is librccl.so unless the environment variable RTMI_RCCL_LIBRARY names another
 * one with the same entry points (ncclCommInitAll, ncclCommDestroy, ncclReduce, ncclGroupStart, ncclGroupEnd,
 * ncclGetErrorString).  rtmi_reduce_same_device_allowed opens it and returns 1 when it is the tests'
 * stand-in (tests/rccl_stub), which sums images that share one GPU; RCCL proper returns 0, and
 * rtmi_reduce_accum then wants every context on a GPU of its own.  rtmi_shutdown waits for the GPUs of the
 * last sum, destroys the communicators rtmi_reduce_accum keeps between calls and closes the library; the
 * next rtmi_reduce_accum opens it again.  Contexts are not touched.  No counterpart in the reference. */
RTMI_API int rtmi_reduce_same_device_allowed(void);
RTMI_API int rtmi_shutdown(void);
/* Device time of the most recent render call, from HIP events recorded around its kernels on the
 * stream they ran on; waits for them to complete. */
RTMI_API int rtmi_last_render_ms(rtmi_ctx *ctx, float *ms);
/* Launch timing log for measurement harnesses: after rtmi_timing_reset(ctx, 1) every render call
 * records its own HIP event pair on its stream without synchronising; rtmi_timing_read waits for
 * them and returns the summed device time and the number of render calls.  rtmi_timing_reset(ctx, 0)
 * turns the log off. */
RTMI_API int rtmi_timing_reset(rtmi_ctx *ctx, int enable);
RTMI_API int rtmi_timing_read(rtmi_ctx *ctx, double *total_ms, uint32_t *launches);

/* Validation outputs (the shader exports none; ids are defined as: sphere i -> i, triangle j ->
 * sphere_number + j, miss -> RTMI_MISS_ID).  Enabling them selects instrumented kernels that also
 * fill the event counters; results of the images are unaffected. */
RTMI_API int rtmi_enable_validation(rtmi_ctx *ctx, int on);
RTMI_API int rtmi_read_hit_ids(rtmi_ctx *ctx, uint32_t *ids, size_t bytes);      /* first segment of the last frame */
RTMI_API int rtmi_read_chain_hash(rtmi_ctx *ctx, uint32_t *hash, size_t bytes);  /* per pixel: sum over the call's frames of a hash of that frame's segment ids */
RTMI_API int rtmi_read_rng_state(rtmi_ctx *ctx, uint32_t *state, size_t bytes);  /* RNG state after the last frame */
RTMI_API int rtmi_get_counters(rtmi_ctx *ctx, uint64_t *counters, size_t n);
RTMI_API int rtmi_reset_counters(rtmi_ctx *ctx);

/* Device arithmetic self-test hooks: evaluate one primitive of the arithmetic contract on the GPU
 * for n inputs.  op: 0 1/x, 1 sqrt, 2 log2 (library polynomial), 3 exp2 (library polynomial),
 * 4 uint32 -> float scaled by 2^-32 (input reinterpreted as uint32), 5 pow(x, y), 6 x / y. */
RTMI_API int rtmi_selftest_math(rtmi_ctx *ctx, int op, const float *x, const float *y, float *out, size_t n);
/* GetEnvironmentLight (RT.comp:390-406) of n arbitrary directions (3 floats each) under fd's sky constants:
 * rgb per direction.  wave_form = 0 evaluates the per-lane routine, 1 the routine the path kernels use,
 * which shares the pow evaluations of a wave's directions out over its lanes. */
RTMI_API int rtmi_selftest_env_light(rtmi_ctx *ctx, const rtmi_frame_data *fd, const float *directions, size_t n,
                                     int wave_form, float *rgb_out);
/* The kernels replace sqrt, 1/x and x / constant by shorter instruction sequences wherever the
 * argument is in the normal range.  This hook compares such a sequence with the general operation for
 * EVERY binary32 bit pattern the sequence is used for.  op: 0 sqrt, 1 reciprocal, 2 division by
 * `divisor`.  out = {patterns tested, mismatches, smallest mismatching pattern}. */
RTMI_API int rtmi_selftest_exhaustive(rtmi_ctx *ctx, int op, float divisor, uint64_t out[3]);
/* ClosestHit (RT.comp:338-388) of n arbitrary rays (3 floats each) against the uploaded scene, using
 * the first sphere_number spheres and mesh_number meshes: ray parameter t (FLT_MAX on a miss) and
 * primitive id per ray.  A test hook for analytic and corner-case rays. */
RTMI_API int rtmi_selftest_closest_hit(rtmi_ctx *ctx, uint32_t sphere_number, uint32_t mesh_number,
                                       const float *origins, const float *directions, size_t n,
                                       float *t_out, uint32_t *prim_out);
/* What the library's scheduling decisions take for the number of compute units of the device from now on (0 = what
 * the device reports, the default): how RTMI_VARIANT_COMPACT deals its work list to waves, how many persistent waves
 * RTMI_VARIANT_STREAM starts.  Never changes a result.  A test hook: with a small number, small images go through the
 * forms that only large ones take on a real device.  No counterpart in the reference. */
RTMI_API int rtmi_assume_compute_units(rtmi_ctx *ctx, int n);
/* How RTMI_VARIANT_COMPACT deals its work list to waves on this context's device and image size when the list
 * holds n_listed pixels and the launch renders n_frames frames of rays_per_pixel samples of a scene with mesh_number
 * meshes (without a hierarchy): pixels per wave (64, 32, 16, 8 or 4), and whether a wave's pixels are neighbours in the
 * list (1) or spread evenly over it (0).  The same function the kernel evaluates.  The choice changes how the work
 * is spread, never a result; tests use this to know which form a configuration exercises.  No counterpart in the
 * reference (the shader's work groups are fixed, Raytracing.comp:89). */
RTMI_API int rtmi_paths_pixels_per_wave(rtmi_ctx *ctx, uint32_t n_listed, uint32_t n_frames, uint32_t rays_per_pixel,
                                        uint32_t mesh_number, uint32_t *pixels_out, int *neighbours_out);
/* The same function for a given image size and number of compute units, without a context or a device (host logic:
 * what the CPU-only tests check the dealing rules with). */
RTMI_API int rtmi_paths_pixels_per_wave_for(uint32_t width, uint32_t height, uint32_t compute_units, uint32_t n_listed,
                                            uint32_t n_frames, uint32_t rays_per_pixel, uint32_t mesh_number,
                                            uint32_t *pixels_out, int *neighbours_out);
/* The same question, for a scene with a hierarchy, answered by a kernel that does nothing but walk: persistent
 * waves whose lanes take ray after ray (in the order given) and resume their walks as the path kernels do.  Same
 * results; *ms_out receives the shortest of `repeats` launches.  A measurement hook: what the walks of a set of
 * rays cost in a given order, without shading, queues or sorting.  No counterpart in the reference. */
RTMI_API int rtmi_selftest_walk(rtmi_ctx *ctx, uint32_t sphere_number, uint32_t mesh_number, const float *origins,
                                const float *directions, size_t n, uint32_t repeats, float *t_out, uint32_t *prim_out,
                                float *ms_out);

/* ------------------------------------------------------------------------------------------
 * Host side kept from the reference (C wrappers over the C++ classes of the same names).
 * ---------------------------------------------------------------------------------------- */

/* LoadModel of ModelLoader.h:6-11 / ModelLoader.cpp:31-142: appends the file's faces to
 * triangles[] and one record to meshes[].  quad_rule: 0 = split a quad along its shorter diagonal,
 * ties as (0,1,3),(1,2,3); 1 = fan (0,1,2),(0,2,3).  Returns the number of triangles appended.
 * A file that cannot be read returns RTMI_ERR_IO and, like the reference (ModelLoader.cpp:43-45),
 * appends nothing. */
RTMI_API int rtmi_load_model(const char *path,
                             rtmi_triangle *triangles, uint32_t *n_triangles, uint32_t triangle_capacity,
                             rtmi_mesh *meshes, uint32_t *n_meshes, uint32_t mesh_capacity,
                             const rtmi_material *material, const float translate[3], const float scale[3],
                             int quad_rule);

/* Text scene file holding the reference's records field by field (RayTracingStructs.h:24-60; the
 * reference itself builds its scene from literals, main.cpp:125-185).  Lines: "rtmi-scene 1", then
 * "sphere cx cy cz radius r g b light smoothness", "tri p1 p2 p3 n1 n2 n3" (xyz each) and
 * "mesh start count lo hi r g b light smoothness"; '#' starts a comment line.  Floats are written
 * with nine significant digits, so a saved scene loads back bit for bit.  rtmi_scene_load with all
 * three arrays NULL only reports the counts; otherwise the arrays must be large enough
 * (RTMI_ERR_OUT_OF_MEMORY if not; the counts are still reported). */
RTMI_API int rtmi_scene_save(const char *path, const rtmi_sphere *spheres, uint32_t n_spheres,
                             const rtmi_triangle *triangles, uint32_t n_triangles,
                             const rtmi_mesh *meshes, uint32_t n_meshes);
RTMI_API int rtmi_scene_load(const char *path, rtmi_sphere *spheres, uint32_t *n_spheres, uint32_t sphere_capacity,
                             rtmi_triangle *triangles, uint32_t *n_triangles, uint32_t triangle_capacity,
                             rtmi_mesh *meshes, uint32_t *n_meshes, uint32_t mesh_capacity);

/* The hierarchy builder that rtmi_upload_scene uses, callable on its own (no device needed): boxes
 * holds n records {lo.x, lo.y, lo.z, hi.x, hi.y, hi.z}.  Writes 64-byte inner nodes
 * {left box lo, left ref} {left box hi, right ref} {right box lo, 0} {right box hi, 0} to nodes_out,
 * the leaf order (leaf position -> box index) to order_out and the root reference to root_out, and
 * returns the number of nodes.  A reference is an inner node index, or a leaf: bit 31 set, bits 28-30
 * the count (1..2), bits 0-27 the first leaf position counted from leaf_base; 0xFFFFFFFF for n = 0.
 * No counterpart in the reference. */
RTMI_API int rtmi_bvh_build(const float *boxes, uint32_t n, uint32_t leaf_base, void *nodes_out, size_t nodes_bytes,
                            uint32_t *order_out, uint32_t *root_out);
/* The second step of the upload, also callable on its own: n_nodes 64-byte nodes as rtmi_bvh_build writes them
 * become the 32-byte nodes the kernels walk, {left ref, right ref, left box x y z, right box x y z}, every box
 * axis a pair of 16-bit step counts lower | upper << 16 in the frame written to frame_out (origin x y z, step
 * x y z): plane = origin + count * step, lower planes rounded down and upper planes up.  A child whose box has
 * an infinite or out-of-frame bound gets bit 27 of its reference set (always visited).  Returns n_nodes.
 * No counterpart in the reference. */
RTMI_API int rtmi_bvh_quantize(const void *nodes, uint32_t n_nodes, void *qnodes_out, size_t qnodes_bytes, float *frame_out);

/* Constants of main.cpp:116-123, 151-153 (rays_per_pixel 4, max bounces 6, sky / sun / ground). */
RTMI_API void rtmi_default_frame_data(rtmi_frame_data *fd);

/* CameraFPS (CameraFPS.h:5-39, CameraFPS.cpp:5-88) without the window: input is passed in. */
typedef struct rtmi_camera {
    float pitch, yaw;               /* degrees; defaults 0, -90 */
    float fov, near_plane, far_plane; /* 70 (handed to the projection as radians), 0.1, 1000 */
    float sensitivity;              /* 0.2 */
    int   moved;
    float projection[16], view[16], inverse_projection[16], inverse_view[16];
    float position[3], forward[3], up[3], right[3];
    double last_mouse_x, last_mouse_y;
    uint32_t width, height;
} rtmi_camera;

typedef struct rtmi_camera_input {
    double mouse_x, mouse_y;
    int mouse_button_2;
    int key_w, key_s, key_a, key_d, key_space, key_left_shift, key_f, key_c;
} rtmi_camera_input;

RTMI_API void rtmi_camera_init(rtmi_camera *cam, uint32_t width, uint32_t height,
                               double mouse_x, double mouse_y);
/* Place the camera explicitly (no counterpart key in the reference; equals setting the public
 * members position / yaw / pitch and recomputing as CameraFPS.cpp:48-53, 85-86 do). */
RTMI_API void rtmi_camera_set_pose(rtmi_camera *cam, const float position[3], float yaw, float pitch);
RTMI_API void rtmi_camera_update(rtmi_camera *cam, double delta_time, const rtmi_camera_input *in);
/* The per-frame policy of main.cpp:355-363 and 332-335: copy the camera into fd, frame_index++,
 * frame_index = 1 if the camera moved; reset_key (the Q key) zeroes frame_index beforehand. */
RTMI_API void rtmi_frame_advance(rtmi_frame_data *fd, const rtmi_camera *cam, int reset_key);

/* Headless replacements for presentation.  rows are written top-down for an upright picture
 * (buffer row 0 is the bottom of the view: main.cpp:340-347).  srgb != 0 applies the
 * linear -> sRGB encode the B8G8R8A8_SRGB swapchain performs (Swapchain.cpp:260-265). */
RTMI_API int rtmi_write_ppm(const char *path, const uint8_t *rgba8, uint32_t width, uint32_t height, int srgb);
/* Mean radiance sum / frames as a little-endian PFM (3 x float32). */
RTMI_API int rtmi_write_pfm(const char *path, const float *accum_rgba, uint32_t width, uint32_t height, uint32_t frames);
/* The same mean radiance as an OpenEXR file: one part, scan lines top-down, no compression, 32-bit float
 * channels B, G, R.  No counterpart in the reference (its only output is the swapchain). */
RTMI_API int rtmi_write_exr(const char *path, const float *accum_rgba, uint32_t width, uint32_t height, uint32_t frames);

#ifdef __cplusplus
}
#endif
#endif /* RTMI_H */
