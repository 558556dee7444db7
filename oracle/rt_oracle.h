/*
 * rt_oracle.h — CPU oracle for the path-tracing hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference compute shader
 * (VulkanRaytracer/res/Shaders/Raytracing.comp) and of the small amount of host code that
 * feeds it (ModelLoader.cpp, CameraFPS.cpp, main.cpp).  It is the checker the GPU kernels are
 * compared against.  Nothing in the shipped library links, loads or calls it; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * PARITY PIN STATUS: "parity unpinned" by the reference — the reference ships no tests, golden
 * vectors or reproducible images for this path and its shader cannot be executed in the build
 * container (no Vulkan loader / ICD / GLSL compiler).  The oracle is therefore pinned only by
 * known-answer vectors derived independently of it (integer RNG stream, std140 struct offsets,
 * dispatch geometry, camera closed forms, analytic intersections); see tests/test_oracle_kat.py.
 *
 * Arithmetic contract ("strict" profile): every float operation is an individually rounded
 * IEEE-754 binary32 operation in the order the shader text writes it (no contraction into FMA),
 * division and sqrt are correctly rounded, and pow() is a fixed polynomial exp2(y*log2(x)) so
 * that a GPU implementation of the same contract is bit-identical.  Compile with
 * -ffp-contract=off and without -ffast-math.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- records with the reference's buffer layouts (RayTracingStructs.h:4-60, Raytracing.comp:3-72) */

typedef struct {
    float    inv_proj[16];      /*   0  column-major mat4                       */
    float    inv_view[16];      /*  64                                          */
    float    cam_pos[4];        /* 128                                          */
    float    cam_dir[4];        /* 144  (never read by the shader)              */
    float    window[2];         /* 160  width, height as floats                 */
    uint32_t rays_per_pixel;    /* 168                                          */
    uint32_t max_bounce;        /* 172                                          */
    float    sky_horizon[4];    /* 176                                          */
    float    sky_zenith[4];     /* 192                                          */
    float    sun_dir[4];        /* 208                                          */
    float    ground[4];         /* 224                                          */
    float    sun_focus;         /* 240                                          */
    float    sun_intensity;     /* 244                                          */
    uint32_t frame_index;       /* 248                                          */
    uint32_t sphere_count;      /* 252                                          */
    uint32_t mesh_count;        /* 256                                          */
} orc_frame_data;               /* 260 bytes */

typedef struct { float color[3]; float light; float smoothness; float pad[3]; } orc_material; /* 32 */
typedef struct { float center[3]; float radius; orc_material material; } orc_sphere;          /* 48 */
typedef struct { float p1[4], p2[4], p3[4], n1[4], n2[4], n3[4]; } orc_triangle;              /* 96 */
typedef struct {
    uint32_t start_triangle, num_triangles; float pad[2];
    float bound_lo[4], bound_hi[4];
    orc_material material;
} orc_mesh;                                                                                   /* 80 */

/* event counters (all uint64), index names */
enum {
    ORC_N_SPHERE_TESTS = 0,
    ORC_N_SPHERE_DISC_GE0,
    ORC_N_SPHERE_WINS,
    ORC_N_BOX_TESTS,
    ORC_N_TRI_TESTS,
    ORC_N_TRI_WINS,
    ORC_N_HIT_BOUNCES,
    ORC_N_ENV_EVALS,
    ORC_N_PIXEL_FRAMES,
    ORC_N_PATHS,
    ORC_N_SEGMENTS,
    ORC_N_COUNTERS = 16
};

#define ORC_MISS_ID 0xFFFFFFFFu

size_t   orc_sizeof(int which); /* 0 frame_data, 1 material, 2 sphere, 3 mesh, 4 triangle */
int      orc_profile(void);      /* 0 strict (liboracle.so), 1 fused multiply-adds at fixed sites (liboracle_fma.so, -DORC_PROFILE_FMA) */
size_t   orc_frame_data_offset(int field); /* field index in declaration order, 0..16 */

/* RNG: Raytracing.comp:106-112 */
float    orc_rng_next(uint32_t *state, uint32_t *raw_out);
uint32_t orc_pixel_seed(uint32_t x, uint32_t y, float width, uint32_t frame_index);

/* Intersections: Raytracing.comp:130-161, 286-309, 311-336.  Return 1 on hit. */
int orc_ray_sphere(const float o[3], const float d[3], const orc_sphere *s,
                   float *t, float pos[3], float nrm[3]);
int orc_ray_triangle(const float o[3], const float d[3], const orc_triangle *tri,
                     float *t, float pos[3], float nrm[3]);
int orc_ray_box(const float o[3], const float d[3], const float lo[3], const float hi[3]);

/* ClosestHit (Raytracing.comp:338-388) of n arbitrary rays against the first sphere_count spheres and
 * mesh_count meshes: ray parameter (FLT_MAX on a miss) and primitive id per ray. */
void orc_closest_hit(uint32_t sphere_count, uint32_t mesh_count,
                     const orc_sphere *spheres, const orc_triangle *tris, const orc_mesh *meshes,
                     const float *origins, const float *dirs, size_t n, float *t_out, uint32_t *prim_out);

/* Sky: Raytracing.comp:390-406 */
void  orc_env_light(const orc_frame_data *fd, const float d[3], float rgb[3]);
/* Extension without counterpart in the shader (off by default): materials with pad[0] > 0.5 transmit, pad[1] is
 * the index of refraction.  Process-wide switch. */
void  orc_set_transmission(int on);
float orc_pow(float x, float y);
float orc_log2(float x);
float orc_exp2(float x);

/* Primary ray of pixel (x,y): Raytracing.comp:446-453 */
void orc_primary_ray(const orc_frame_data *fd, uint32_t x, uint32_t y, float o[3], float d[3]);

/* Host math restated from glm's published algorithms (CameraFPS.cpp:13-25, 85-86). */
void orc_perspective_fov_lh_zo(float fov, float w, float h, float zn, float zf, float out[16]);
void orc_look_at_lh(const float eye[3], const float center[3], const float up[3], float out[16]);
void orc_mat4_inverse(const float m[16], float out[16]);
void orc_camera_forward(float yaw_deg, float pitch_deg, float fwd[3]);
/* Fills inv_proj, inv_view, cam_pos, cam_dir, window of fd for a pose. */
void orc_camera_pose(orc_frame_data *fd, const float pos[3], float yaw_deg, float pitch_deg,
                     uint32_t width, uint32_t height);

/* OBJ loader + per-file mesh record: ModelLoader.cpp:31-142.
 * quad_rule: 0 = shorter diagonal, tie -> (0,1,3),(1,2,3); 1 = fan (0,1,2),(0,2,3).
 * Appends to tris/meshes; returns number of triangles appended or -1 on error. */
int orc_load_model(const char *path, orc_triangle *tris, uint32_t *n_tris, uint32_t tri_cap,
                   orc_mesh *meshes, uint32_t *n_meshes, uint32_t mesh_cap,
                   const orc_material *mat, const float translate[3], const float scale[3],
                   int quad_rule);

/* Scene constants of main.cpp:116-123,151-153 (sky, sun, rpp=4, bounces=6). */
void orc_default_frame_constants(orc_frame_data *fd);

/*
 * Render frames first_frame .. first_frame+n_frames-1 exactly as n_frames successive dispatches of
 * the shader would (Raytracing.comp:441-481), accumulating into accum (float4 per pixel, row-major,
 * in/out).  All optional outputs may be NULL.
 *   out_rgba8  : display image of the LAST frame (sum / frameIndex -> UNORM8, alpha 255)
 *   prim_id    : closest-hit primitive id of the FIRST segment of the first path of the LAST frame
 *                (sphere i -> i; triangle j -> sphere_count + j; miss -> ORC_MISS_ID)
 *   chain_hash : per pixel, the sum (mod 2^32) over the frames of the call of each frame's hash: an
 *                FNV-style fold (h = (h ^ id) * 16777619) over the primitive ids of all the frame's
 *                segments in order, started at 2166136261 ^ (frame_index * 0x9E3779B9)
 *   rng_state  : per pixel RNG state after the LAST frame
 *   counters   : ORC_N_COUNTERS uint64 event totals (added to)
 */
int orc_render(const orc_frame_data *fd,
               const orc_sphere *spheres, const orc_triangle *tris, uint32_t n_tris,
               const orc_mesh *meshes,
               uint32_t first_frame, uint32_t n_frames,
               float *accum, uint8_t *out_rgba8, uint32_t *prim_id, uint32_t *chain_hash,
               uint32_t *rng_state, uint64_t *counters, int n_threads);

/* Same for image rows y0 <= y < y1 only (buffers are full size; other rows are left untouched). */
int orc_render_rows(const orc_frame_data *fd,
                    const orc_sphere *spheres, const orc_triangle *tris, uint32_t n_tris,
                    const orc_mesh *meshes,
                    uint32_t first_frame, uint32_t n_frames,
                    float *accum, uint8_t *out_rgba8, uint32_t *prim_id, uint32_t *chain_hash,
                    uint32_t *rng_state, uint64_t *counters, int n_threads, uint32_t y0, uint32_t y1);

#ifdef __cplusplus
}
#endif
#endif
