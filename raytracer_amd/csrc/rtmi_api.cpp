// rtmi_api.cpp — the C ABI of include/rtmi.h over the gfx950 kernels: context, scene repack and
// upload, render calls, read-backs.  There is no CPU fallback: without a HIP device every entry
// point that needs one fails with RTMI_ERR_NO_DEVICE / RTMI_ERR_HIP.
#include "rtmi.h"

#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>   // types and prototypes only: the library is opened at run time, when several GPUs are summed

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <exception>
#include <mutex>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "rt_bvh.hpp"
#include "rt_pack.hpp"
#include "rt_host.hpp"
#include "rt_bvh_device.hpp"
#include "rt_kernels.hpp"

#pragma clang fp contract(off)

static_assert(rtmi::kMaxDepth <= rtd::kBvhStack, "the traversal stack holds one entry per level the builder may create");
static_assert(sizeof(rtmi_frame_data) == 260, "uniform block is 260 bytes");
static_assert(sizeof(rtmi_material) == 32, "material is 32 bytes");
static_assert(sizeof(rtmi_sphere) == 48, "sphere is 48 bytes");
static_assert(sizeof(rtmi_mesh) == 80, "mesh is 80 bytes");
static_assert(sizeof(rtmi_triangle) == 96, "triangle is 96 bytes");

struct rtmi_ctx {
    int device = 0;
    uint32_t width = 0, height = 0;
    int variant = RTMI_VARIANT_AUTO;
    bool validate = false;
    std::string err;

    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    hipEvent_t ev_noalloc = nullptr;   // recorded after every no-allocation launch (orders the users of the static work area)
    bool timed = false;
    // launch timing log: one event pair per render call since the last rtmi_timing_reset
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    size_t ev_used = 0;
    bool ev_logging = false;

    float4 *own_accum = nullptr;
    float4 *accum = nullptr;
    uint32_t *out_rgba8 = nullptr;
    uint32_t *hit_ids = nullptr, *chain_hash = nullptr, *rng_state = nullptr;
    uint32_t *hit_list = nullptr, *hit_count = nullptr;  // variant 2 work list; hit_count[16] is variant 4's cursor
    uint32_t *scratch = nullptr;   // this device's copy of the library's static work area (rtmi_render_frames_noalloc)
    int n_cus = 0;
    int assumed_cus = 0;   // rtmi_assume_compute_units: what the scheduling decisions take for the device's size (0 = n_cus)
    // Primary-segment cache (RTMI_VARIANT_AUTO only).  With an unchanged camera, scene and frame
    // constants every launch would classify the same pixels and find the same primary hits; they are
    // stored once (with the work list) and later launches start from them.  The key is the whole
    // uniform block except frame_index, plus the scene upload it was built for.
    float4 *primary = nullptr;
    uint32_t *primary_mat = nullptr;
    bool cache_valid = false, cache_has_list = false;   // primary hits stored / work list stored with them
    rtmi_frame_data cache_key{};
    uint64_t scene_generation = 0, cache_generation = 0;
    unsigned long long *counters = nullptr;

    // scene
    bool has_scene = false;
    uint32_t n_spheres = 0, n_triangles = 0, n_meshes = 0;
    void *scene_blob = nullptr;
    size_t scene_blob_bytes = 0;
    rtd::DeviceScene dscene{};
    int accel_mode = RTMI_ACCEL_AUTO;  // applied by the next rtmi_upload_scene
    float graze_cap = RTMI_GRAZING_CAP_DEFAULT;  // rtmi_set_grazing_cap
    bool transmission = false;         // rtmi_enable_transmission: applied by the next rtmi_upload_scene
    bool bvh_active = false;           // the uploaded scene carries hierarchies and the kernels use them
    int profile_mode = RTMI_PROFILE_STRICT;  // applied by the next rtmi_upload_scene
    bool fused = false;                // the uploaded scene was prepared for, and is rendered by, the fused profile
    std::vector<rtmi_mesh> host_meshes;  // kept for range checks against frame data
};

// Work area of rtmi_render_frames_noalloc, which may not allocate: a launch-wide counter block (64
// words: [0] length of the work list, [16] the pixel-streaming kernel's cursor) followed by the work
// list of the classify + paths kernels, one word per pixel.  Every device the library runs on holds its
// own copy at its own address (fetched once per context with hipGetSymbolAddress, never used by name in
// host code); all no-allocation calls on one device share it.
constexpr uint32_t kNoAllocMaxPixels = RTMI_NOALLOC_MAX_PIXELS;
constexpr uint32_t kScratchHeadWords = 64u;
static __device__ __align__(256) uint32_t scratch[kScratchHeadWords + kNoAllocMaxPixels];

namespace {

// The last no-allocation launch on each device: the next one, if it runs on another stream, first waits for
// that launch's event (created with its context, so nothing is allocated here), which orders the users of
// the device's one work area whatever streams their contexts use.
struct ScratchUser {
    rtmi_ctx *ctx = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
};
constexpr int kMaxDevices = 64;
ScratchUser g_scratch_user[kMaxDevices];
std::mutex g_scratch_lock;

thread_local std::string g_create_error;

int fail(rtmi_ctx *ctx, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    else g_create_error = buf;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                        \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(ctx, e_ == hipErrorOutOfMemory ? RTMI_ERR_OUT_OF_MEMORY : RTMI_ERR_HIP,   \
                        "%s failed: %s", #expr, hipGetErrorString(e_));                           \
    } while (0)

size_t n_pixels(const rtmi_ctx *c) { return (size_t)c->width * c->height; }

int use_device(rtmi_ctx *ctx)
{
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return RTMI_OK;
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

inline bool is_pow2(uint32_t v) { return v && !(v & (v - 1)); }

// frames per launch from which RTMI_VARIANT_AUTO takes the compacting pair (measurements where it is used); half as
// many for images up to 2^21 pixels
#ifndef RTMI_AUTO_COMPACT_FRAMES
#define RTMI_AUTO_COMPACT_FRAMES 8
#endif
constexpr uint32_t kAutoCompactFrames = RTMI_AUTO_COMPACT_FRAMES;
// primitive count above which RTMI_ACCEL_AUTO builds hierarchies
// (measured on random sphere scenes at 1080p: the hierarchy overtakes the exhaustive loops between 256 and
// 512 primitives)
constexpr uint32_t kAutoBvhThreshold = 256;

// largest float <= v and smallest float >= v
inline float round_down(double v)
{
    float f = (float)v;
    return (double)f > v ? std::nextafterf(f, -INFINITY) : f;
}
inline float round_up(double v)
{
    float f = (float)v;
    return (double)f < v ? std::nextafterf(f, INFINITY) : f;
}

// The kernels' view of a packed scene whose bytes start at `base` (device memory for the library; the
// tests' host emulation of the traversal binds the host copy the same way).
void bind_scene(const rtmi::PackedScene &pk, const unsigned char *base, rtd::DeviceScene &ds)
{
    RTMI_BIND_PACKED_SCENE(pk, base, ds);
}

}  // namespace

extern "C" {

const char *rtmi_version(void) { return "rtmi 0.1 (gfx950, strict-fp32)"; }

const char *rtmi_last_error(const rtmi_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int rtmi_create(int device, uint32_t width, uint32_t height, rtmi_ctx **out)
{
    if (!out) return fail(nullptr, RTMI_ERR_INVALID_ARGUMENT, "rtmi_create: out is NULL");
    *out = nullptr;
    if (width == 0 || height == 0 || (uint64_t)width * height > 0x7FFFFFFFull)
        return fail(nullptr, RTMI_ERR_INVALID_ARGUMENT, "rtmi_create: bad extent %ux%u", width, height);
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0)
        return fail(nullptr, RTMI_ERR_NO_DEVICE, "rtmi_create: no HIP device (%s)", hipGetErrorString(e));
    if (device < 0 || device >= count)
        return fail(nullptr, RTMI_ERR_INVALID_ARGUMENT, "rtmi_create: device %d out of range (0..%d)", device, count - 1);
    rtmi_ctx *c = new (std::nothrow) rtmi_ctx();
    if (!c) return fail(nullptr, RTMI_ERR_OUT_OF_MEMORY, "rtmi_create: host allocation failed");
    c->device = device;
    c->width = width;
    c->height = height;
    auto bail = [&](hipError_t err, const char *what) {
        int code = fail(nullptr, err == hipErrorOutOfMemory ? RTMI_ERR_OUT_OF_MEMORY : RTMI_ERR_HIP, "rtmi_create: %s: %s",
                        what, hipGetErrorString(err));
        rtmi_destroy(c);
        return code;
    };
    if ((e = hipSetDevice(device)) != hipSuccess) return bail(e, "hipSetDevice");
    if ((e = hipDeviceGetAttribute(&c->n_cus, hipDeviceAttributeMultiprocessorCount, device)) != hipSuccess) return bail(e, "hipDeviceGetAttribute");
    // this device's copy of the static work area (the symbol's name is not a device address in host code)
    if ((e = hipGetSymbolAddress(reinterpret_cast<void **>(&c->scratch), HIP_SYMBOL(scratch))) != hipSuccess) return bail(e, "hipGetSymbolAddress(scratch)");
    if ((e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
    c->stream = c->own_stream;
    if ((e = hipEventCreate(&c->ev_start)) != hipSuccess) return bail(e, "hipEventCreate");
    if ((e = hipEventCreate(&c->ev_stop)) != hipSuccess) return bail(e, "hipEventCreate");
    if ((e = hipEventCreateWithFlags(&c->ev_noalloc, hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
    const size_t np = n_pixels(c);
    if ((e = hipMalloc(&c->own_accum, np * sizeof(float4))) != hipSuccess) return bail(e, "hipMalloc(accum)");
    c->accum = c->own_accum;
    if ((e = hipMalloc(&c->out_rgba8, np * 4)) != hipSuccess) return bail(e, "hipMalloc(rgba8)");
    if ((e = hipMalloc(&c->counters, RTMI_N_COUNTERS * sizeof(unsigned long long))) != hipSuccess) return bail(e, "hipMalloc(counters)");
    if ((e = hipMalloc(&c->hit_list, np * sizeof(uint32_t))) != hipSuccess) return bail(e, "hipMalloc(hit_list)");
    if ((e = hipMalloc(&c->hit_count, 256)) != hipSuccess) return bail(e, "hipMalloc(hit_count)");
    // main.cpp:78-82: both images start zeroed
    if ((e = hipMemsetAsync(c->own_accum, 0, np * sizeof(float4), c->stream)) != hipSuccess) return bail(e, "hipMemset");
    if ((e = hipMemsetAsync(c->out_rgba8, 0, np * 4, c->stream)) != hipSuccess) return bail(e, "hipMemset");
    if ((e = hipMemsetAsync(c->counters, 0, RTMI_N_COUNTERS * sizeof(unsigned long long), c->stream)) != hipSuccess) return bail(e, "hipMemset");
    if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) return bail(e, "hipStreamSynchronize");
    *out = c;
    return RTMI_OK;
}

int rtmi_destroy(rtmi_ctx *c)
{
    if (!c) return RTMI_OK;
    (void)hipSetDevice(c->device);
    if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
    if (c->device >= 0 && c->device < kMaxDevices) {
        // the work area's last user, if it is this context, has finished once its stream has drained
        std::lock_guard<std::mutex> guard(g_scratch_lock);
        ScratchUser &u = g_scratch_user[c->device];
        if (u.ctx == c) {
            if (u.stream != c->own_stream && u.done) (void)hipEventSynchronize(u.done);
            u = ScratchUser();
        }
    }
    if (c->ev_noalloc) (void)hipEventDestroy(c->ev_noalloc);
    (void)hipFree(c->own_accum);
    (void)hipFree(c->out_rgba8);
    (void)hipFree(c->hit_ids);
    (void)hipFree(c->hit_list);
    (void)hipFree(c->hit_count);
    (void)hipFree(c->primary);
    (void)hipFree(c->primary_mat);
    (void)hipFree(c->chain_hash);
    (void)hipFree(c->rng_state);
    (void)hipFree(c->counters);
    (void)hipFree(c->scene_blob);
    if (c->ev_start) (void)hipEventDestroy(c->ev_start);
    if (c->ev_stop) (void)hipEventDestroy(c->ev_stop);
    for (auto &pr : c->ev_pool) {
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return RTMI_OK;
}

// Repack the reference's AoS records into the device layout of rtd::DeviceScene.  Everything
// precomputed here (r*r, the two edges, their cross product) is ray-independent and evaluated with
// the same single operations the shader would repeat per ray (Raytracing.comp:141, 288-290), so the
// bits the kernels consume are the bits the shader would have produced.
int rtmi_upload_scene(rtmi_ctx *c, const rtmi_sphere *spheres, uint32_t nS, const rtmi_triangle *tris, uint32_t nT,
                      const rtmi_mesh *meshes, uint32_t nM)
{
    if (!c) return RTMI_ERR_INVALID_ARGUMENT;
    if ((nS && !spheres) || (nT && !tris) || (nM && !meshes))
        return fail(c, RTMI_ERR_INVALID_ARGUMENT, "rtmi_upload_scene: NULL array with non-zero count");
    for (uint32_t m = 0; m < nM; m++) {
        const int64_t s = meshes[m].start_triangle, n = meshes[m].num_triangles;
        if (s < 0 || n < 0 || s + n > (int64_t)nT)
            return fail(c, RTMI_ERR_INVALID_ARGUMENT, "rtmi_upload_scene: mesh %u triangle range [%lld,+%lld) exceeds %u triangles",
                        m, (long long)s, (long long)n, nT);
    }
    if (int rc = use_device(c)) return rc;

    const bool fused = c->profile_mode == RTMI_PROFILE_FMA;
    const bool want_bvh = c->accel_mode == RTMI_ACCEL_BVH ||
                          (c->accel_mode == RTMI_ACCEL_AUTO && (uint64_t)nS + nT > kAutoBvhThreshold);
    // everything up to the bytes that go to the device is host work (host/rt_pack.cpp)
    rtmi::PackedScene pk;
    std::string why;
    if (const int rc = rtmi::PackScene(spheres, nS, tris, nT, meshes, nM, want_bvh, fused, pk, why, c->transmission)) return fail(c, rc, "%s", why.c_str());
    const size_t total = pk.blob.size();

    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->scene_blob_bytes < total) {
        (void)hipFree(c->scene_blob);
        c->scene_blob = nullptr;
        c->scene_blob_bytes = 0;
        HIP_TRY(c, hipMalloc(&c->scene_blob, total));
        c->scene_blob_bytes = total;
    }
    HIP_TRY(c, hipMemcpy(c->scene_blob, pk.blob.data(), total, hipMemcpyHostToDevice));
    unsigned char *base = static_cast<unsigned char *>(c->scene_blob);
    bind_scene(pk, base, c->dscene);
    c->dscene.bvh_graze_cap = c->graze_cap;
    c->bvh_active = want_bvh;
    c->fused = fused;
    c->dscene.n_spheres = nS;
    c->dscene.n_triangles = nT;
    c->dscene.n_meshes = nM;
    c->n_spheres = nS;
    c->n_triangles = nT;
    c->n_meshes = nM;
    c->has_scene = true;
    c->scene_generation++;  // whatever was stored for the previous scene is stale
    return RTMI_OK;
}

static int ensure_validation_buffers(rtmi_ctx *c)
{
    const size_t bytes = n_pixels(c) * sizeof(uint32_t);
    if (!c->hit_ids) HIP_TRY(c, hipMalloc(&c->hit_ids, bytes));
    if (!c->chain_hash) HIP_TRY(c, hipMalloc(&c->chain_hash, bytes));
    if (!c->rng_state) HIP_TRY(c, hipMalloc(&c->rng_state, bytes));
    return RTMI_OK;
}

// the uniform block's camera, image and sky members (Raytracing.comp:3-24) as kernel arguments
static void fill_frame_params(const rtmi_ctx *c, const rtmi_frame_data *fd, RenderParams &p)
{
    memset(&p, 0, sizeof p);
    memcpy(p.inv_proj, fd->camera_inverse_projection, sizeof p.inv_proj);
    memcpy(p.inv_view, fd->camera_inverse_view, sizeof p.inv_view);
    memcpy(p.cam_pos, fd->camera_pos, sizeof p.cam_pos);
    p.window_w = fd->window[0];
    p.window_h = fd->window[1];
    p.width = c->width;
    p.height = c->height;
    p.rays_per_pixel = fd->rays_per_pixel;
    p.max_bounce = fd->max_bounce_limit;
    memcpy(p.sky_horizon, fd->sky_color_horizon, sizeof p.sky_horizon);
    memcpy(p.sky_zenith, fd->sky_color_zenith, sizeof p.sky_zenith);
    memcpy(p.sun_dir, fd->sun_light_direction, sizeof p.sun_dir);
    memcpy(p.ground, fd->ground_color, sizeof p.ground);
    p.sun_focus = fd->sun_focus;
    p.sun_intensity = fd->sun_intensity;
}

// Both render entry points.  no_alloc: use the library's static work area instead of the context's
// buffers, never store primary segments, record no events: nothing on this path allocates.
static int render_impl(rtmi_ctx *c, const rtmi_frame_data *fd, uint32_t first_frame, uint32_t n_frames, bool no_alloc)
{
    if (!c) return RTMI_ERR_INVALID_ARGUMENT;
    if (!fd) return fail(c, RTMI_ERR_INVALID_ARGUMENT, "rtmi_render: frame data is NULL");
    if (!c->has_scene) return fail(c, RTMI_ERR_NO_SCENE, "rtmi_render: no scene uploaded");
    if (n_frames == 0) return RTMI_OK;
    // the kernels count items (frames x pixel slots of a wave) in 32 bits, and the shader's frame index has 32
    if (n_frames > (1u << 24) || first_frame > 0xFFFFFFFFu - (n_frames - 1u))
        return fail(c, RTMI_ERR_INVALID_ARGUMENT, "rtmi_render_frames: %u frames from frame %u on (at most 2^24 per call, frame index below 2^32)",
                    n_frames, first_frame);
    if (fd->sphere_number > c->n_spheres || fd->mesh_number > c->n_meshes)
        return fail(c, RTMI_ERR_SCENE_MISMATCH, "rtmi_render: frame data names %u spheres / %u meshes, scene holds %u / %u",
                    fd->sphere_number, fd->mesh_number, c->n_spheres, c->n_meshes);
    // The shader trusts window == image extent (no bounds check, main.cpp:376); here the image
    // extent is fixed at creation and a mismatching window is rejected instead of read out of range.
    if ((uint32_t)fd->window[0] != c->width || (uint32_t)fd->window[1] != c->height)
        return fail(c, RTMI_ERR_INVALID_ARGUMENT, "rtmi_render: window %gx%g does not match context %ux%u",
                    (double)fd->window[0], (double)fd->window[1], c->width, c->height);
    if (no_alloc) {
        if (n_pixels(c) > kNoAllocMaxPixels)
            return fail(c, RTMI_ERR_UNSUPPORTED, "rtmi_render_frames_noalloc: %ux%u = %zu pixels exceed the static work area of %u pixels",
                        c->width, c->height, n_pixels(c), kNoAllocMaxPixels);
        if (c->validate) return fail(c, RTMI_ERR_UNSUPPORTED, "rtmi_render_frames_noalloc: not available with validation outputs enabled");
    }
    if (int rc = use_device(c)) return rc;
    if (c->validate)
        if (int rc = ensure_validation_buffers(c)) return rc;

    RenderParams p;
    fill_frame_params(c, fd, p);
    p.sphere_number = fd->sphere_number;
    p.mesh_number = fd->mesh_number;
    p.first_frame = first_frame;
    p.n_frames = n_frames;
    p.rpp_is_pow2 = is_pow2(fd->rays_per_pixel) ? 1u : 0u;
    p.rpp_recip = p.rpp_is_pow2 ? 1.0f / (float)fd->rays_per_pixel : 0.0f;
    p.scene = c->dscene;
    p.use_bvh = c->bvh_active ? 1u : 0u;
    p.accum = c->accum;
    p.out_rgba8 = c->out_rgba8;
    p.hit_ids = c->hit_ids;
    p.hit_list = no_alloc ? c->scratch + kScratchHeadWords : c->hit_list;
    p.hit_count = no_alloc ? c->scratch : c->hit_count;
    p.stream_cursor = p.hit_count + 16;
    p.n_cus = (uint32_t)(c->assumed_cus > 0 ? c->assumed_cus : c->n_cus);
    p.chain_hash = c->chain_hash;
    p.rng_state = c->rng_state;
    p.counters = c->counters;

    int variant = c->variant;
    RtCacheUse cache = RT_CACHE_NONE;
    bool store_has_list = false;
    rtmi_frame_data key = *fd;
    key.frame_index = 0;
    if (variant == RTMI_VARIANT_AUTO) {
        // The compacting pair hands out work by the frame: the samples of one frame of a pixel share a random number
        // stream (Raytracing.comp:464-468) and stay with one lane.  With few frames in a launch its lanes run dry
        // while the longest pixel finishes, however many samples a frame has, and the one-lane-per-pixel kernel,
        // whose waves hold neighbouring pixels of similar cost, is ahead: measured on both benchmark scenes at
        // 1280x720, 1080p and 3840x2160 from stored hits, 1 frame of 8 / 16 samples 250-2240 us against the pair's
        // 370-4540, 2 frames of 4 about 20 % ahead; 4 frames a tie up to 1080p (+-4 %) and 13-17 % ahead at
        // 3840x2160; from 8 frames on the pair wins everywhere (8 frames of 1 sample: 136-1202 against 216-1285 us).
        const bool compact_runs = fd->max_bounce_limit >= 2 && fd->rays_per_pixel >= 1;  // else the launcher falls back
        const bool same_as_stored = c->cache_valid && c->cache_generation == c->scene_generation &&
                                    memcmp(&key, &c->cache_key, sizeof key) == 0;
        // With a hierarchy the pixel-streaming kernel is ahead at every launch size (10^6 + 10^6 primitive
        // scene at 1080p: 1 frame 11.4 ms against COMPACT's 27, 8 frames 80 against 93, 16 frames 158 against
        // 172, 64 frames 630 against 643); it starts from nothing stored.
        const bool stream = c->bvh_active;
        // a scene with transmitting materials (the extension) is rendered by the one-lane-per-pixel and the
        // streaming kernels only, and starts from nothing stored
        const bool glass = c->dscene.has_glass != 0u;
        const bool use_store = compact_runs && !c->validate && !stream && !no_alloc && !glass;
        const bool compact = !stream && !glass &&
                             (n_frames >= kAutoCompactFrames || (2u * n_frames >= kAutoCompactFrames && n_pixels(c) <= (size_t(1) << 21)));
        if (use_store) {
            const size_t np = n_pixels(c);
            if (!c->primary) HIP_TRY(c, hipMalloc(&c->primary, np * sizeof(float4)));
            if (!c->primary_mat) HIP_TRY(c, hipMalloc(&c->primary_mat, np * sizeof(uint32_t)));
            // the work list is only needed (and only built) by COMPACT
            cache = (same_as_stored && (!compact || c->cache_has_list)) ? RT_CACHE_LOAD : RT_CACHE_STORE;
            store_has_list = compact;
        }
        variant = stream ? RTMI_VARIANT_STREAM : compact ? RTMI_VARIANT_COMPACT : RTMI_VARIANT_REGENERATE;
    }
    // A launch that (re)writes the stored primary segments or the context's work list makes what was
    // stored unusable until it has been enqueued successfully; so does a failure of any kind below.
    const bool rewrites_store = cache == RT_CACHE_STORE || (variant == RTMI_VARIANT_COMPACT && cache == RT_CACHE_NONE && !no_alloc);
    if (cache == RT_CACHE_STORE) c->cache_valid = false;
    if (rewrites_store) c->cache_has_list = false;
    p.primary = c->primary;
    p.primary_mat = c->primary_mat;
    hipEvent_t ev0 = c->ev_start, ev1 = c->ev_stop;
    const bool events = !no_alloc;   // hipEventCreate may allocate
    if (events && c->ev_logging) {
        if (c->ev_used == c->ev_pool.size()) {
            hipEvent_t a = nullptr, b = nullptr;
            HIP_TRY(c, hipEventCreate(&a));
            HIP_TRY(c, hipEventCreate(&b));
            c->ev_pool.emplace_back(a, b);
        }
        ev0 = c->ev_pool[c->ev_used].first;
        ev1 = c->ev_pool[c->ev_used].second;
        c->ev_used++;
    }
    auto drop_store = [&]() {
        c->cache_valid = false;
        c->cache_has_list = false;
    };
    if (events) {
        hipError_t er = hipEventRecord(ev0, c->stream);
        if (er != hipSuccess) {
            drop_store();
            return fail(c, RTMI_ERR_HIP, "hipEventRecord failed: %s", hipGetErrorString(er));
        }
    }
    hipError_t e = hipSuccess;
    if (no_alloc) {
        // one work area per device: wait for its previous user unless that was this very stream
        if (c->device >= kMaxDevices) return fail(c, RTMI_ERR_UNSUPPORTED, "rtmi_render_frames_noalloc: device %d beyond %d", c->device, kMaxDevices);
        std::lock_guard<std::mutex> guard(g_scratch_lock);
        ScratchUser &u = g_scratch_user[c->device];
        // A stream in capture mode runs nothing now: the launch becomes part of a graph, and when that graph
        // runs relative to other users of the work area is for whoever launches it to arrange.
        hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
        e = hipStreamIsCapturing(c->stream, &capture);
        const bool live = e == hipSuccess && capture == hipStreamCaptureStatusNone;
        if (e == hipSuccess && live && u.ctx && u.stream != c->stream) e = hipStreamWaitEvent(c->stream, u.done, 0);
        if (e == hipSuccess)
            e = c->fused ? rt_launch_render_fma(variant, c->validate, p, c->stream, cache) : rt_launch_render(variant, c->validate, p, c->stream, cache);
        if (e == hipSuccess && live) e = hipEventRecord(c->ev_noalloc, c->stream);
        if (e == hipSuccess && live) {
            u.ctx = c;
            u.stream = c->stream;
            u.done = c->ev_noalloc;
        }
    } else {
        e = c->fused ? rt_launch_render_fma(variant, c->validate, p, c->stream, cache) : rt_launch_render(variant, c->validate, p, c->stream, cache);
    }
    if (e != hipSuccess) {
        drop_store();
        return fail(c, RTMI_ERR_HIP, "kernel launch (%s) failed: %s", rt_kernel_name(variant), hipGetErrorString(e));
    }
    if (events) {
        hipError_t er = hipEventRecord(ev1, c->stream);
        if (er != hipSuccess) {
            drop_store();
            return fail(c, RTMI_ERR_HIP, "hipEventRecord failed: %s", hipGetErrorString(er));
        }
    }
    if (cache == RT_CACHE_STORE) {
        // stored by the launch just enqueued, on c->stream (rtmi_set_stream drops the store: a launch on
        // another stream would not be ordered after this one)
        c->cache_key = key;
        c->cache_generation = c->scene_generation;
        c->cache_valid = true;
        c->cache_has_list = store_has_list;
    }
    c->timed = events && !c->ev_logging;
    return RTMI_OK;
}

int rtmi_render_frames(rtmi_ctx *c, const rtmi_frame_data *fd, uint32_t first_frame, uint32_t n_frames)
{
    return render_impl(c, fd, first_frame, n_frames, false);
}

// No counterpart in the reference.  For callers that cannot allocate device memory (code under HIP
// graph capture, tight serving loops, harnesses that forbid it).
int rtmi_render_frames_noalloc(rtmi_ctx *c, const rtmi_frame_data *fd, uint32_t first_frame, uint32_t n_frames)
{
    return render_impl(c, fd, first_frame, n_frames, true);
}

int rtmi_render(rtmi_ctx *c, const rtmi_frame_data *fd)
{
    if (!c) return RTMI_ERR_INVALID_ARGUMENT;
    if (!fd) return fail(c, RTMI_ERR_INVALID_ARGUMENT, "rtmi_render: frame data is NULL");
    return rtmi_render_frames(c, fd, fd->frame_index, 1);
}

int rtmi_set_variant(rtmi_ctx *c, int variant)
{
    if (!c) return RTMI_ERR_INVALID_ARGUMENT;
    if (variant != RTMI_VARIANT_MEGAKERNEL && variant != RTMI_VARIANT_REGENERATE && variant != RTMI_VARIANT_COMPACT &&
        variant != RTMI_VARIANT_AUTO && variant != RTMI_VARIANT_STREAM)
        return fail(c, RTMI_ERR_UNSUPPORTED, "rtmi_set_variant: variant %d not available", variant);
    c->variant = variant;
    return RTMI_OK;
}

int rtmi_get_variant(const rtmi_ctx *c) { return c ? c->variant : RTMI_ERR_INVALID_ARGUMENT; }

int rtmi_set_acceleration(rtmi_ctx *c, int mode)
{
    if (!c) return RTMI_ERR_INVALID_ARGUMENT;
    if (mode != RTMI_ACCEL_AUTO && mode != RTMI_ACCEL_NONE && mode != RTMI_ACCEL_BVH)
        return fail(c, RTMI_ERR_UNSUPPORTED, "rtmi_set_acceleration: mode %d not available", mode);
    c->accel_mode = mode;
    return RTMI_OK;
}

int rtmi_set_grazing_cap(rtmi_ctx *c, float cap)
{
    if (!c) return RTMI_ERR_INVALID_ARGUMENT;
    if (!(cap >= 0.0f)) return fail(c, RTMI_ERR_INVALID_ARGUMENT, "rtmi_set_grazing_cap: %g is not a fraction of the distance", (double)cap);
    c->graze_cap = cap;
    c->dscene.bvh_graze_cap = cap;   // a kernel argument: the next launch reads it
    return RTMI_OK;
}

float rtmi_get_grazing_cap(const rtmi_ctx *c) { return c ? c->graze_cap : -1.0f; }

int rtmi_enable_transmission(rtmi_ctx *c, int on)
{
    if (!c) return RTMI_ERR_INVALID_ARGUMENT;
    c->transmission = on != 0;
    return RTMI_OK;
}

int rtmi_get_acceleration(const rtmi_ctx *c) { return c ? (c->bvh_active ? RTMI_ACCEL_BVH : RTMI_ACCEL_NONE) : RTMI_ERR_INVALID_ARGUMENT; }

int rtmi_set_profile(rtmi_ctx *c, int profile)
{
    if (!c) return RTMI_ERR_INVALID_ARGUMENT;
    if (profile != RTMI_PROFILE_STRICT && profile != RTMI_PROFILE_FMA)
        return fail(c, RTMI_ERR_UNSUPPORTED, "rtmi_set_profile: profile %d not available", profile);
    c->profile_mode = profile;
    return RTMI_OK;
}

int rtmi_get_profile(const rtmi_ctx *c)
{
    if (!c) return RTMI_ERR_INVALID_ARGUMENT;
    return c->has_scene ? (c->fused ? RTMI_PROFILE_FMA : RTMI_PROFILE_STRICT) : c->profile_mode;
}

int rtmi_clear_accum(rtmi_ctx *c)
{
    if (!c) return RTMI_ERR_INVALID_ARGUMENT;
    if (int rc = use_device(c)) return rc;
    HIP_TRY(c, hipMemsetAsync(c->accum, 0, n_pixels(c) * sizeof(float4), c->stream));
    HIP_TRY(c, hipMemsetAsync(c->out_rgba8, 0, n_pixels(c) * 4, c->stream));
    return RTMI_OK;
}

static int read_back(rtmi_ctx *c, void *dst, const void *src, size_t bytes, size_t need, const char *what)
{
    if (!c) return RTMI_ERR_INVALID_ARGUMENT;
    if (!dst || bytes < need) return fail(c, RTMI_ERR_INVALID_ARGUMENT, "%s: need a %zu-byte buffer, got %zu", what, need, bytes);
    if (!src) return fail(c, RTMI_ERR_INVALID_ARGUMENT, "%s: nothing to read (validation outputs not enabled?)", what);
    if (int rc = use_device(c)) return rc;
    HIP_TRY(c, hipMemcpyAsync(dst, src, need, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RTMI_OK;
}

int rtmi_read_accum(rtmi_ctx *c, float *rgba, size_t bytes)
{
    return read_back(c, rgba, c ? c->accum : nullptr, bytes, c ? n_pixels(c) * 16 : 0, "rtmi_read_accum");
}

int rtmi_write_accum(rtmi_ctx *c, const float *rgba, size_t bytes)
{
    if (!c) return RTMI_ERR_INVALID_ARGUMENT;
    const size_t need = n_pixels(c) * 16;
    if (!rgba || bytes < need) return fail(c, RTMI_ERR_INVALID_ARGUMENT, "rtmi_write_accum: need %zu bytes, got %zu", need, bytes);
    if (int rc = use_device(c)) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->accum, rgba, need, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RTMI_OK;
}

int rtmi_read_output_rgba8(rtmi_ctx *c, uint8_t *rgba, size_t bytes)
{
    return read_back(c, rgba, c ? c->out_rgba8 : nullptr, bytes, c ? n_pixels(c) * 4 : 0, "rtmi_read_output_rgba8");
}

int rtmi_resolve_output(rtmi_ctx *c, uint32_t divisor)
{
    if (!c) return RTMI_ERR_INVALID_ARGUMENT;
    if (int rc = use_device(c)) return rc;
    if (divisor == 0) return fail(c, RTMI_ERR_INVALID_ARGUMENT, "rtmi_resolve_output: divisor is 0");
    hipError_t e = rt_launch_resolve(c->accum, c->out_rgba8, (uint32_t)n_pixels(c), (float)divisor, c->stream);
    if (e != hipSuccess) return fail(c, RTMI_ERR_HIP, "resolve launch failed: %s", hipGetErrorString(e));
    return RTMI_OK;
}

// Snapshot of the accumulation image and the frame count it holds.  A context that loads it and goes
// on with frame_index = frames_done + 1 produces the bits an uninterrupted run produces: a frame's
// contribution depends only on pixel, frame index, scene and camera (Raytracing.comp:462), and the sum
// is carried in this image alone (Raytracing.comp:471-477).
int rtmi_checkpoint_save(rtmi_ctx *c, const char *path, uint32_t frames_done)
{
    if (!c) return RTMI_ERR_INVALID_ARGUMENT;
    if (!path) return fail(c, RTMI_ERR_INVALID_ARGUMENT, "rtmi_checkpoint_save: path is NULL");
    if (int rc = use_device(c)) return rc;
    std::vector<float> host;
    try {
        host.resize(n_pixels(c) * 4);
    } catch (const std::bad_alloc &) {
        return fail(c, RTMI_ERR_OUT_OF_MEMORY, "rtmi_checkpoint_save: host allocation failed");
    }
    HIP_TRY(c, hipMemcpyAsync(host.data(), c->accum, host.size() * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (!rtmi::SaveSnapshot(path, c->width, c->height, frames_done, host.data()))
        return fail(c, RTMI_ERR_IO, "rtmi_checkpoint_save: cannot write %s", path);
    return RTMI_OK;
}

int rtmi_checkpoint_load(rtmi_ctx *c, const char *path, uint32_t *frames_done)
{
    if (!c) return RTMI_ERR_INVALID_ARGUMENT;
    if (!path || !frames_done) return fail(c, RTMI_ERR_INVALID_ARGUMENT, "rtmi_checkpoint_load: NULL argument");
    rtmi::Snapshot snap;
    try {
        if (!rtmi::LoadSnapshot(path, snap)) return fail(c, RTMI_ERR_IO, "rtmi_checkpoint_load: %s is not a readable snapshot", path);
    } catch (const std::bad_alloc &) {
        return fail(c, RTMI_ERR_OUT_OF_MEMORY, "rtmi_checkpoint_load: host allocation failed");
    }
    if (snap.width != c->width || snap.height != c->height)
        return fail(c, RTMI_ERR_INVALID_ARGUMENT, "rtmi_checkpoint_load: snapshot is %ux%u, context is %ux%u", snap.width,
                    snap.height, c->width, c->height);
    if (int rc = use_device(c)) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->accum, snap.accum.data(), snap.accum.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
    if (snap.frames > 0) {
        // the display image the interrupted run had after its last frame (Raytracing.comp:480)
        hipError_t e = rt_launch_resolve(c->accum, c->out_rgba8, (uint32_t)n_pixels(c), (float)snap.frames, c->stream);
        if (e != hipSuccess) return fail(c, RTMI_ERR_HIP, "resolve launch failed: %s", hipGetErrorString(e));
    } else {
        HIP_TRY(c, hipMemsetAsync(c->out_rgba8, 0, n_pixels(c) * 4, c->stream));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));  // snap.accum must outlive the copy
    *frames_done = snap.frames;
    return RTMI_OK;
}

int rtmi_bind_accum(rtmi_ctx *c, void *device_float4)
{
    if (!c) return RTMI_ERR_INVALID_ARGUMENT;
    if (device_float4 && (reinterpret_cast<uintptr_t>(device_float4) & 15u))
        return fail(c, RTMI_ERR_INVALID_ARGUMENT, "rtmi_bind_accum: pointer must be 16-byte aligned");
    c->accum = device_float4 ? static_cast<float4 *>(device_float4) : c->own_accum;
    c->cache_valid = c->cache_has_list = false;
    return RTMI_OK;
}

void *rtmi_accum_device_ptr(rtmi_ctx *c) { return c ? c->accum : nullptr; }

int rtmi_set_stream(rtmi_ctx *c, void *hip_stream)
{
    if (!c) return RTMI_ERR_INVALID_ARGUMENT;
    c->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->own_stream;
    c->timed = false;
    // what an earlier launch stored was written on the previous stream; nothing orders this one after it
    c->cache_valid = c->cache_has_list = false;
    return RTMI_OK;
}

int rtmi_synchronize(rtmi_ctx *c)
{
    if (!c) return RTMI_ERR_INVALID_ARGUMENT;
    if (int rc = use_device(c)) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RTMI_OK;
}

int rtmi_last_render_ms(rtmi_ctx *c, float *ms)
{
    if (!c || !ms) return RTMI_ERR_INVALID_ARGUMENT;
    if (!c->timed) return fail(c, RTMI_ERR_INVALID_ARGUMENT, "rtmi_last_render_ms: no render call has been made");
    if (int rc = use_device(c)) return rc;
    HIP_TRY(c, hipEventSynchronize(c->ev_stop));
    HIP_TRY(c, hipEventElapsedTime(ms, c->ev_start, c->ev_stop));
    return RTMI_OK;
}

int rtmi_timing_reset(rtmi_ctx *c, int enable)
{
    if (!c) return RTMI_ERR_INVALID_ARGUMENT;
    c->ev_used = 0;
    c->ev_logging = enable != 0;
    c->timed = false;
    return RTMI_OK;
}

int rtmi_timing_read(rtmi_ctx *c, double *total_ms, uint32_t *launches)
{
    if (!c || !total_ms || !launches) return RTMI_ERR_INVALID_ARGUMENT;
    if (int rc = use_device(c)) return rc;
    double sum = 0.0;
    for (size_t i = 0; i < c->ev_used; i++) {
        float ms = 0.0f;
        HIP_TRY(c, hipEventSynchronize(c->ev_pool[i].second));
        HIP_TRY(c, hipEventElapsedTime(&ms, c->ev_pool[i].first, c->ev_pool[i].second));
        sum += ms;
    }
    *total_ms = sum;
    *launches = (uint32_t)c->ev_used;
    return RTMI_OK;
}

int rtmi_enable_validation(rtmi_ctx *c, int on)
{
    if (!c) return RTMI_ERR_INVALID_ARGUMENT;
    c->validate = on != 0;
    c->cache_valid = c->cache_has_list = false;
    if (c->validate) {
        if (int rc = use_device(c)) return rc;
        return ensure_validation_buffers(c);
    }
    return RTMI_OK;
}

int rtmi_read_hit_ids(rtmi_ctx *c, uint32_t *ids, size_t bytes)
{
    return read_back(c, ids, c ? c->hit_ids : nullptr, bytes, c ? n_pixels(c) * 4 : 0, "rtmi_read_hit_ids");
}

int rtmi_read_chain_hash(rtmi_ctx *c, uint32_t *hash, size_t bytes)
{
    return read_back(c, hash, c ? c->chain_hash : nullptr, bytes, c ? n_pixels(c) * 4 : 0, "rtmi_read_chain_hash");
}

int rtmi_read_rng_state(rtmi_ctx *c, uint32_t *state, size_t bytes)
{
    return read_back(c, state, c ? c->rng_state : nullptr, bytes, c ? n_pixels(c) * 4 : 0, "rtmi_read_rng_state");
}

int rtmi_get_counters(rtmi_ctx *c, uint64_t *counters, size_t n)
{
    if (!c) return RTMI_ERR_INVALID_ARGUMENT;
    if (!counters || n < RTMI_N_COUNTERS) return fail(c, RTMI_ERR_INVALID_ARGUMENT, "rtmi_get_counters: need %d entries", RTMI_N_COUNTERS);
    return read_back(c, counters, c->counters, n * 8, RTMI_N_COUNTERS * 8, "rtmi_get_counters");
}

int rtmi_reset_counters(rtmi_ctx *c)
{
    if (!c) return RTMI_ERR_INVALID_ARGUMENT;
    if (int rc = use_device(c)) return rc;
    HIP_TRY(c, hipMemsetAsync(c->counters, 0, RTMI_N_COUNTERS * sizeof(unsigned long long), c->stream));
    return RTMI_OK;
}

int rtmi_bvh_node_bytes(void) { return (int)sizeof(rtmi::QNode); }

int rtmi_assume_compute_units(rtmi_ctx *c, int n)
{
    if (!c) return fail(nullptr, RTMI_ERR_INVALID_ARGUMENT, "rtmi_assume_compute_units: NULL context");
    if (n < 0 || n > 65536) return fail(c, RTMI_ERR_INVALID_ARGUMENT, "rtmi_assume_compute_units: %d is not a number of compute units", n);
    c->assumed_cus = n;
    return RTMI_OK;
}

int rtmi_paths_pixels_per_wave_for(uint32_t width, uint32_t height, uint32_t compute_units, uint32_t n_listed, uint32_t n_frames,
                                   uint32_t rays_per_pixel, uint32_t mesh_number, uint32_t *pixels_out, int *neighbours_out)
{
    if (!pixels_out || !neighbours_out) return fail(nullptr, RTMI_ERR_INVALID_ARGUMENT, "rtmi_paths_pixels_per_wave_for: NULL argument");
    if (width == 0 || height == 0 || (uint64_t)width * height > 0xFFFFFFFFull || n_listed > (uint64_t)width * height)
        return fail(nullptr, RTMI_ERR_INVALID_ARGUMENT, "rtmi_paths_pixels_per_wave_for: %u listed pixels of a %u x %u image", n_listed, width, height);
    const uint32_t deal = rt_paths_deal(width * height, n_listed, n_frames, rays_per_pixel, compute_units, mesh_number != 0u);
    *pixels_out = 1u << (deal & 255u);
    *neighbours_out = (deal & 256u) != 0u;
    return RTMI_OK;
}

int rtmi_paths_pixels_per_wave(rtmi_ctx *c, uint32_t n_listed, uint32_t n_frames, uint32_t rays_per_pixel, uint32_t mesh_number,
                               uint32_t *pixels_out, int *neighbours_out)
{
    if (!c) return fail(nullptr, RTMI_ERR_INVALID_ARGUMENT, "rtmi_paths_pixels_per_wave: NULL context");
    const int rc = rtmi_paths_pixels_per_wave_for(c->width, c->height, (uint32_t)(c->assumed_cus > 0 ? c->assumed_cus : c->n_cus), n_listed,
                                                  n_frames, rays_per_pixel, mesh_number, pixels_out, neighbours_out);
    return rc == RTMI_OK ? rc : fail(c, rc, "%s", rtmi_last_error(nullptr));
}

int rtmi_device_count(void)
{
    int count = 0;
    const hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess) return fail(nullptr, RTMI_ERR_NO_DEVICE, "rtmi_device_count: %s", hipGetErrorString(e));
    return count;
}

// RCCL, opened when first needed (a process that renders on one GPU never loads it).  The environment
// variable RTMI_RCCL_LIBRARY names another library with the same six entry points; one that also exports
// the marker rtmi_rccl_stub_same_device (the sum-by-copy stand-in that the tests build, which needs no second
// GPU) lifts the rule that every context of a sum sits on a GPU of its own.
namespace {
struct Rccl {
    void *handle = nullptr;
    bool same_device_ok = false;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclReduce) Reduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    // communicators are kept for the device list they were made for (making them costs about a second)
    std::vector<int> devices;
    std::vector<ncclComm_t> comms;
    std::mutex lock;
    bool Load(std::string &why)
    {
        if (handle) return true;
        const char *named = getenv("RTMI_RCCL_LIBRARY");
        if (named && *named) {
            handle = dlopen(named, RTLD_NOW | RTLD_LOCAL);
        } else {
            for (const char *name : {"librccl.so.1", "librccl.so"}) {
                handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
                if (handle) break;
            }
        }
        if (!handle) {
            const char *msg = dlerror();
            why = msg ? msg : "dlopen failed";
            return false;
        }
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(dlsym(handle, "ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(handle, "ncclCommDestroy"));
        Reduce = reinterpret_cast<decltype(Reduce)>(dlsym(handle, "ncclReduce"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(dlsym(handle, "ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(dlsym(handle, "ncclGroupEnd"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(handle, "ncclGetErrorString"));
        if (!CommInitAll || !CommDestroy || !Reduce || !GroupStart || !GroupEnd || !GetErrorString) {
            why = "the collective library lacks an expected symbol";
            dlclose(handle);
            handle = nullptr;
            return false;
        }
        same_device_ok = dlsym(handle, "rtmi_rccl_stub_same_device") != nullptr;
        return true;
    }
    // Communicators may still have collectives in flight on the streams they were used with: every device
    // of the list is drained first.
    void DropComms()
    {
        for (int dev : devices)
            if (hipSetDevice(dev) == hipSuccess) (void)hipDeviceSynchronize();
        for (ncclComm_t cm : comms)
            if (cm) (void)CommDestroy(cm);
        comms.clear();
        devices.clear();
    }
} g_rccl;

// puts the calling thread's current device back when the scope ends
struct DeviceRestore {
    int saved = -1;
    DeviceRestore() { if (hipGetDevice(&saved) != hipSuccess) saved = -1; }
    ~DeviceRestore() { if (saved >= 0) (void)hipSetDevice(saved); }
};
}  // namespace

int rtmi_reduce_same_device_allowed(void)
{
    std::lock_guard<std::mutex> guard(g_rccl.lock);
    std::string why;
    if (!g_rccl.Load(why)) return 0;
    return g_rccl.same_device_ok ? 1 : 0;
}

int rtmi_shutdown(void)
{
    std::lock_guard<std::mutex> guard(g_rccl.lock);
    if (!g_rccl.handle) return RTMI_OK;
    DeviceRestore restore;
    g_rccl.DropComms();
    dlclose(g_rccl.handle);
    g_rccl.handle = nullptr;
    g_rccl.same_device_ok = false;
    return RTMI_OK;
}

// Sum of the accumulation images of n contexts, one per GPU, onto contexts[root]: one RCCL reduce over xGMI,
// enqueued on every context's stream (no host synchronisation; the root's stream holds the sum once its
// part of the collective has run).  The partial images must have been rendered from cleared images over
// disjoint frame ranges (rtmi_clear_accum, then rtmi_render_frames); afterwards rtmi_resolve_output on
// the root restores alpha and rebuilds the display image.
int rtmi_reduce_accum(rtmi_ctx *const *contexts, uint32_t n, uint32_t root)
{
    if (!contexts || n == 0 || root >= n) return RTMI_ERR_INVALID_ARGUMENT;
    rtmi_ctx *r = contexts[root];
    if (!r) return RTMI_ERR_INVALID_ARGUMENT;
    std::vector<int> devices(n);
    bool shared_gpu = false;
    for (uint32_t i = 0; i < n; i++) {
        if (!contexts[i]) return fail(r, RTMI_ERR_INVALID_ARGUMENT, "rtmi_reduce_accum: context %u is NULL", i);
        if (contexts[i]->width != r->width || contexts[i]->height != r->height)
            return fail(r, RTMI_ERR_INVALID_ARGUMENT, "rtmi_reduce_accum: context %u is %ux%u, the root %ux%u", i, contexts[i]->width,
                        contexts[i]->height, r->width, r->height);
        devices[i] = contexts[i]->device;
        for (uint32_t j = 0; j < i; j++) {
            if (contexts[j] == contexts[i])
                return fail(r, RTMI_ERR_INVALID_ARGUMENT, "rtmi_reduce_accum: contexts %u and %u are the same context", j, i);
            if (devices[j] == devices[i]) shared_gpu = true;
        }
    }
    if (n == 1) return RTMI_OK;   // a sum over one image
    std::lock_guard<std::mutex> guard(g_rccl.lock);
    std::string why;
    if (!shared_gpu || (getenv("RTMI_RCCL_LIBRARY") && *getenv("RTMI_RCCL_LIBRARY"))) {
        if (!g_rccl.Load(why)) return fail(r, RTMI_ERR_UNSUPPORTED, "rtmi_reduce_accum: cannot open RCCL: %s", why.c_str());
    }
    // RCCL proper wants one GPU per rank; only the tests' stand-in sums images that share one
    if (shared_gpu && !(g_rccl.handle && g_rccl.same_device_ok))
        return fail(r, RTMI_ERR_INVALID_ARGUMENT, "rtmi_reduce_accum: two of the contexts are on the same GPU");
    DeviceRestore restore;
    if (g_rccl.devices != devices) {
        g_rccl.DropComms();
        g_rccl.comms.assign(n, nullptr);
        const ncclResult_t rc = g_rccl.CommInitAll(g_rccl.comms.data(), (int)n, devices.data());
        if (rc != ncclSuccess) {
            g_rccl.comms.clear();
            return fail(r, RTMI_ERR_HIP, "rtmi_reduce_accum: ncclCommInitAll: %s", g_rccl.GetErrorString(rc));
        }
        g_rccl.devices = devices;
    }
    const size_t count = n_pixels(r) * 4;
    ncclResult_t rc = g_rccl.GroupStart();
    if (rc != ncclSuccess) return fail(r, RTMI_ERR_HIP, "rtmi_reduce_accum: ncclGroupStart: %s", g_rccl.GetErrorString(rc));
    for (uint32_t i = 0; i < n && rc == ncclSuccess; i++) {
        if (hipSetDevice(devices[i]) != hipSuccess) rc = ncclUnhandledCudaError;
        else rc = g_rccl.Reduce(contexts[i]->accum, contexts[i]->accum, count, ncclFloat32, ncclSum, (int)root, g_rccl.comms[i], contexts[i]->stream);
    }
    // the group is closed even after a failed call, or the library stays in group mode
    const ncclResult_t rc_end = g_rccl.GroupEnd();
    if (rc == ncclSuccess) rc = rc_end;
    if (rc != ncclSuccess) return fail(r, RTMI_ERR_HIP, "rtmi_reduce_accum: %s", g_rccl.GetErrorString(rc));
    return RTMI_OK;
}

int rtmi_selftest_math(rtmi_ctx *c, int op, const float *x, const float *y, float *out, size_t n)
{
    if (!c) return RTMI_ERR_INVALID_ARGUMENT;
    if (!x || !out || op < 0 || op > 6 || ((op == 5 || op == 6) && !y))
        return fail(c, RTMI_ERR_INVALID_ARGUMENT, "rtmi_selftest_math: bad arguments");
    if (n == 0) return RTMI_OK;
    if (int rc = use_device(c)) return rc;
    float *dx = nullptr, *dy = nullptr, *dout = nullptr;
    int rc = RTMI_OK;
    hipError_t e = hipMalloc(&dx, n * 4);
    if (e == hipSuccess) e = hipMalloc(&dout, n * 4);
    if (e == hipSuccess && y) e = hipMalloc(&dy, n * 4);
    if (e == hipSuccess) e = hipMemcpyAsync(dx, x, n * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && y) e = hipMemcpyAsync(dy, y, n * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = c->profile_mode == RTMI_PROFILE_FMA ? rt_launch_selftest_fma(op, dx, dy, dout, n, c->stream) : rt_launch_selftest(op, dx, dy, dout, n, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out, dout, n * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) rc = fail(c, RTMI_ERR_HIP, "rtmi_selftest_math: %s", hipGetErrorString(e));
    (void)hipFree(dx);
    (void)hipFree(dy);
    (void)hipFree(dout);
    return rc;
}

int rtmi_selftest_env_light(rtmi_ctx *c, const rtmi_frame_data *fd, const float *directions, size_t n, int wave_form, float *rgb_out)
{
    if (!c) return RTMI_ERR_INVALID_ARGUMENT;
    if (!fd || !directions || !rgb_out) return fail(c, RTMI_ERR_INVALID_ARGUMENT, "rtmi_selftest_env_light: NULL argument");
    if (n == 0) return RTMI_OK;
    if (int rc = use_device(c)) return rc;
    RenderParams p;
    fill_frame_params(c, fd, p);
    float *dbuf = nullptr;
    hipError_t e = hipMalloc(&dbuf, n * 6 * sizeof(float));
    if (e != hipSuccess || !dbuf)
        return fail(c, e == hipErrorOutOfMemory ? RTMI_ERR_OUT_OF_MEMORY : RTMI_ERR_HIP, "rtmi_selftest_env_light: hipMalloc: %s",
                    hipGetErrorString(e));
    int rc = RTMI_OK;
    e = hipMemcpyAsync(dbuf, directions, n * 12, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = c->profile_mode == RTMI_PROFILE_FMA ? rt_launch_selftest_env_fma(p, dbuf, n, wave_form != 0, dbuf + 3 * n, c->stream)
                                                             : rt_launch_selftest_env(p, dbuf, n, wave_form != 0, dbuf + 3 * n, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(rgb_out, dbuf + 3 * n, n * 12, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) rc = fail(c, RTMI_ERR_HIP, "rtmi_selftest_env_light: %s", hipGetErrorString(e));
    (void)hipFree(dbuf);
    return rc;
}

int rtmi_selftest_exhaustive(rtmi_ctx *c, int op, float divisor, uint64_t out[3])
{
    if (!c) return RTMI_ERR_INVALID_ARGUMENT;
    if (!out || op < 0 || op > 2 || (op == 2 && !(std::fabs(divisor) >= 1e-6f && std::fabs(divisor) <= 1e6f)))
        return fail(c, RTMI_ERR_INVALID_ARGUMENT, "rtmi_selftest_exhaustive: bad arguments");
    if (int rc = use_device(c)) return rc;
    unsigned long long *d = nullptr;
    const unsigned long long init[3] = {0ull, 0ull, ~0ull};
    int rc = RTMI_OK;
    hipError_t e = hipMalloc(&d, sizeof init);
    if (e == hipSuccess) e = hipMemcpyAsync(d, init, sizeof init, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = rt_launch_selftest_exhaustive(op, divisor, d, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out, d, sizeof init, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) rc = fail(c, RTMI_ERR_HIP, "rtmi_selftest_exhaustive: %s", hipGetErrorString(e));
    (void)hipFree(d);
    return rc;
}

int rtmi_selftest_closest_hit(rtmi_ctx *c, uint32_t sphere_number, uint32_t mesh_number, const float *origins,
                              const float *directions, size_t n, float *t_out, uint32_t *prim_out)
{
    if (!c) return RTMI_ERR_INVALID_ARGUMENT;
    if (!origins || !directions || !t_out || !prim_out) return fail(c, RTMI_ERR_INVALID_ARGUMENT, "rtmi_selftest_closest_hit: NULL argument");
    if (!c->has_scene) return fail(c, RTMI_ERR_NO_SCENE, "rtmi_selftest_closest_hit: no scene uploaded");
    if (sphere_number > c->n_spheres || mesh_number > c->n_meshes)
        return fail(c, RTMI_ERR_SCENE_MISMATCH, "rtmi_selftest_closest_hit: counts exceed the uploaded scene");
    if (n == 0) return RTMI_OK;
    if (int rc = use_device(c)) return rc;
    float *dbuf = nullptr;
    int rc = RTMI_OK;
    // one allocation: origins | directions | t | prim
    hipError_t e = hipMalloc(&dbuf, n * 8 * sizeof(float));
    if (e != hipSuccess || !dbuf)
        return fail(c, e == hipErrorOutOfMemory ? RTMI_ERR_OUT_OF_MEMORY : RTMI_ERR_HIP, "rtmi_selftest_closest_hit: hipMalloc: %s",
                    hipGetErrorString(e));
    float *d_o = dbuf, *d_d = dbuf + 3 * n, *d_t = dbuf + 6 * n;
    uint32_t *d_p = reinterpret_cast<uint32_t *>(dbuf + 7 * n);
    if (e == hipSuccess) e = hipMemcpyAsync(d_o, origins, n * 12, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_d, directions, n * 12, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = c->fused ? rt_launch_selftest_hit_fma(c->dscene, c->bvh_active, sphere_number, mesh_number, d_o, d_d, n, d_t, d_p, c->stream)
                                        : rt_launch_selftest_hit(c->dscene, c->bvh_active, sphere_number, mesh_number, d_o, d_d, n, d_t, d_p, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(t_out, d_t, n * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(prim_out, d_p, n * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) rc = fail(c, RTMI_ERR_HIP, "rtmi_selftest_closest_hit: %s", hipGetErrorString(e));
    (void)hipFree(dbuf);
    return rc;
}

// The same rays through the walk-only streaming kernel, timed: `repeats` launches, the shortest in *ms_out.
int rtmi_selftest_walk(rtmi_ctx *c, uint32_t sphere_number, uint32_t mesh_number, const float *origins, const float *directions,
                       size_t n, uint32_t repeats, float *t_out, uint32_t *prim_out, float *ms_out)
{
    if (!c) return RTMI_ERR_INVALID_ARGUMENT;
    if (!origins || !directions || !t_out || !prim_out) return fail(c, RTMI_ERR_INVALID_ARGUMENT, "rtmi_selftest_walk: NULL argument");
    if (!c->has_scene || !c->bvh_active) return fail(c, RTMI_ERR_NO_SCENE, "rtmi_selftest_walk: no scene with a hierarchy uploaded");
    if (sphere_number > c->n_spheres || mesh_number > c->n_meshes)
        return fail(c, RTMI_ERR_SCENE_MISMATCH, "rtmi_selftest_walk: counts exceed the uploaded scene");
    if (n == 0) return RTMI_OK;
    if (n > 0x7FFFFFFFull) return fail(c, RTMI_ERR_UNSUPPORTED, "rtmi_selftest_walk: at most 2^31 - 1 rays");
    if (repeats == 0) repeats = 1;
    if (int rc = use_device(c)) return rc;
    float *dbuf = nullptr;
    int rc = RTMI_OK;
    // one allocation: origins | directions | t | prim | cursor
    hipError_t e = hipMalloc(&dbuf, n * 8 * sizeof(float) + 256);
    if (e != hipSuccess || !dbuf)
        return fail(c, e == hipErrorOutOfMemory ? RTMI_ERR_OUT_OF_MEMORY : RTMI_ERR_HIP, "rtmi_selftest_walk: hipMalloc: %s", hipGetErrorString(e));
    float *d_o = dbuf, *d_d = dbuf + 3 * n, *d_t = dbuf + 6 * n;
    uint32_t *d_p = reinterpret_cast<uint32_t *>(dbuf + 7 * n);
    uint32_t *d_cursor = reinterpret_cast<uint32_t *>(dbuf + 8 * n);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) e = hipMemcpyAsync(d_o, origins, n * 12, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_d, directions, n * 12, hipMemcpyHostToDevice, c->stream);
    float best = 0.0f;
    for (uint32_t r = 0; r < repeats && e == hipSuccess; r++) {
        e = hipMemsetAsync(d_cursor, 0, 4, c->stream);
        if (e == hipSuccess) e = hipEventRecord(e0, c->stream);
        if (e == hipSuccess)
            e = c->fused ? rt_launch_selftest_walk_fma(c->dscene, sphere_number, mesh_number, d_o, d_d, (uint32_t)n, d_t, d_p, d_cursor, (uint32_t)c->n_cus, c->stream)
                         : rt_launch_selftest_walk(c->dscene, sphere_number, mesh_number, d_o, d_d, (uint32_t)n, d_t, d_p, d_cursor, (uint32_t)c->n_cus, c->stream);
        if (e == hipSuccess) e = hipEventRecord(e1, c->stream);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        float ms = 0.0f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e == hipSuccess && (r == 0 || ms < best)) best = ms;
    }
    if (e == hipSuccess) e = hipMemcpyAsync(t_out, d_t, n * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(prim_out, d_p, n * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) rc = fail(c, RTMI_ERR_HIP, "rtmi_selftest_walk: %s", hipGetErrorString(e));
    if (ms_out) *ms_out = best;
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(dbuf);
    return rc;
}

}  // extern "C"
