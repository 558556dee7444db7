/*
 * rt_oracle.c — CPU oracle (TEST INFRASTRUCTURE ONLY; see rt_oracle.h for the contract and the
 * "parity unpinned" statement).  Each function cites the reference lines it restates.
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fno-fast-math -fopenmp -fPIC -shared
 */
#include "rt_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

_Static_assert(sizeof(orc_frame_data) == 260, "FrameData is 260 bytes (RayTracingStructs.h:4-22)");
_Static_assert(sizeof(orc_material) == 32, "Material is 32 bytes");
_Static_assert(sizeof(orc_sphere) == 48, "Sphere is 48 bytes");
_Static_assert(sizeof(orc_mesh) == 80, "Mesh is 80 bytes");
_Static_assert(sizeof(orc_triangle) == 96, "Triangle is 96 bytes");

typedef struct { float x, y, z; } v3;

size_t orc_sizeof(int which)
{
    switch (which) {
    case 0: return sizeof(orc_frame_data);
    case 1: return sizeof(orc_material);
    case 2: return sizeof(orc_sphere);
    case 3: return sizeof(orc_mesh);
    case 4: return sizeof(orc_triangle);
    default: return 0;
    }
}

size_t orc_frame_data_offset(int field)
{
    switch (field) {
    case 0:  return offsetof(orc_frame_data, inv_proj);
    case 1:  return offsetof(orc_frame_data, inv_view);
    case 2:  return offsetof(orc_frame_data, cam_pos);
    case 3:  return offsetof(orc_frame_data, cam_dir);
    case 4:  return offsetof(orc_frame_data, window);
    case 5:  return offsetof(orc_frame_data, rays_per_pixel);
    case 6:  return offsetof(orc_frame_data, max_bounce);
    case 7:  return offsetof(orc_frame_data, sky_horizon);
    case 8:  return offsetof(orc_frame_data, sky_zenith);
    case 9:  return offsetof(orc_frame_data, sun_dir);
    case 10: return offsetof(orc_frame_data, ground);
    case 11: return offsetof(orc_frame_data, sun_focus);
    case 12: return offsetof(orc_frame_data, sun_intensity);
    case 13: return offsetof(orc_frame_data, frame_index);
    case 14: return offsetof(orc_frame_data, sphere_count);
    case 15: return offsetof(orc_frame_data, mesh_count);
    default: return (size_t)-1;
    }
}

/* ---------------------------------------------------------------------------------------------
 * GLSL built-ins, restated with one rounding per operation and a fixed association order.
 * min/max follow the GLSL definition text ("y if y < x, otherwise x" / "y if x < y, otherwise x")
 * so NaN operands behave deterministically.
 * ------------------------------------------------------------------------------------------ */
static inline float g_min(float x, float y) { return (y < x) ? y : x; }
static inline float g_max(float x, float y) { return (x < y) ? y : x; }
static inline float g_clamp(float x, float lo, float hi) { return g_min(g_max(x, lo), hi); }
static inline float g_mix(float a, float b, float t) { return a * (1.0f - t) + b * t; }

static inline v3 v3_make(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_mul(v3 a, v3 b) { return v3_make(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 v3_scale(v3 a, float s) { return v3_make(a.x * s, a.y * s, a.z * s); }
static inline v3 v3_neg(v3 a) { return v3_make(-a.x, -a.y, -a.z); }
static inline float v3_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 v3_cross(v3 a, v3 b)
{
    return v3_make(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
/* normalize(v) = v * (1 / sqrt(dot(v, v))) */
static inline v3 v3_normalize(v3 a)
{
    float inv = 1.0f / sqrtf(v3_dot(a, a));
    return v3_scale(a, inv);
}
static inline v3 v3_mix(v3 a, v3 b, float t)
{
    return v3_make(g_mix(a.x, b.x, t), g_mix(a.y, b.y, t), g_mix(a.z, b.z, t));
}
/* reflect(I, N) = I - 2 * dot(N, I) * N */
static inline v3 v3_reflect(v3 i, v3 n)
{
    float k = 2.0f * v3_dot(n, i);
    return v3_sub(i, v3_scale(n, k));
}
static inline float g_smoothstep(float e0, float e1, float x)
{
    float t = g_clamp((x - e0) / (e1 - e0), 0.0f, 1.0f);
    return t * t * (3.0f - 2.0f * t);
}
static inline v3 v3_from(const float *p) { return v3_make(p[0], p[1], p[2]); }

/* ---------------------------------------------------------------------------------------------
 * Arithmetic profile of the SHADER's operations (the host-side camera and loader arithmetic above and
 * below is not affected).  Default ("strict"): every multiply and add is one rounded operation.  With
 * -DORC_PROFILE_FMA ("fused", liboracle_fma.so) a fixed list of sites uses fmaf, one rounding: the sums
 * of products of dot, cross, mix, reflect and the matrix-vector products, the Horner steps of pow,
 * b*b - 4ac, origin + direction * t, the blend of the shading normals, and light + emission * throughput /
 * light + sky * throughput.  Raytracing.comp carries no `precise`, so a shader compiler may contract it
 * like this; fmaf is correctly rounded everywhere, so the fused profile is as reproducible as the strict one.
 * ------------------------------------------------------------------------------------------ */
#ifdef ORC_PROFILE_FMA
#define ORC_FUSED 1
#else
#define ORC_FUSED 0
#endif
int orc_profile(void) { return ORC_FUSED; }

/* a * b + c at a site the fused profile fuses */
static inline float sh_mul_add(float a, float b, float c)
{
#if ORC_FUSED
    return fmaf(a, b, c);
#else
    return a * b + c;
#endif
}
static inline float sh_dot(v3 a, v3 b)
{
#if ORC_FUSED
    return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x));
#else
    return a.x * b.x + a.y * b.y + a.z * b.z;
#endif
}
static inline v3 sh_cross(v3 a, v3 b)
{
#if ORC_FUSED
    return v3_make(fmaf(a.y, b.z, -(b.y * a.z)), fmaf(a.z, b.x, -(b.z * a.x)), fmaf(a.x, b.y, -(b.x * a.y)));
#else
    return v3_make(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
#endif
}
static inline v3 sh_normalize(v3 a)
{
    float inv = 1.0f / sqrtf(sh_dot(a, a));
    return v3_scale(a, inv);
}
static inline float sh_mix1(float a, float b, float t)
{
#if ORC_FUSED
    return fmaf(b, t, a * (1.0f - t));
#else
    return a * (1.0f - t) + b * t;
#endif
}
static inline v3 sh_mix(v3 a, v3 b, float t)
{
    return v3_make(sh_mix1(a.x, b.x, t), sh_mix1(a.y, b.y, t), sh_mix1(a.z, b.z, t));
}
/* reflect(I, N) = I - 2 * dot(N, I) * N */
static inline v3 sh_reflect(v3 i, v3 n)
{
    float k = 2.0f * sh_dot(n, i);
#if ORC_FUSED
    return v3_make(fmaf(-n.x, k, i.x), fmaf(-n.y, k, i.y), fmaf(-n.z, k, i.z));
#else
    return v3_sub(i, v3_scale(n, k));
#endif
}
/* o + d * t */
static inline v3 sh_along(v3 o, v3 d, float t)
{
#if ORC_FUSED
    return v3_make(fmaf(d.x, t, o.x), fmaf(d.y, t, o.y), fmaf(d.z, t, o.z));
#else
    return v3_add(o, v3_scale(d, t));
#endif
}
/* l + a * b, componentwise */
static inline v3 sh_add_mul(v3 l, v3 a, v3 b)
{
#if ORC_FUSED
    return v3_make(fmaf(a.x, b.x, l.x), fmaf(a.y, b.y, l.y), fmaf(a.z, b.z, l.z));
#else
    return v3_add(l, v3_mul(a, b));
#endif
}
/* column-major mat4 times vec4, summed left to right over columns */
static inline void sh_mat4_mul_vec4(const float m[16], const float v[4], float out[4])
{
    for (int r = 0; r < 4; r++) {
#if ORC_FUSED
        out[r] = fmaf(m[12 + r], v[3], fmaf(m[8 + r], v[2], fmaf(m[4 + r], v[1], m[0 + r] * v[0])));
#else
        out[r] = m[0 + r] * v[0] + m[4 + r] * v[1] + m[8 + r] * v[2] + m[12 + r] * v[3];
#endif
    }
}

/* column-major mat4 times vec4, summed left to right over columns */
static inline void mat4_mul_vec4(const float m[16], const float v[4], float out[4])
{
    for (int r = 0; r < 4; r++)
        out[r] = m[0 + r] * v[0] + m[4 + r] * v[1] + m[8 + r] * v[2] + m[12 + r] * v[3];
}

/* ---------------------------------------------------------------------------------------------
 * pow(x, y) := exp2(y * log2(x)) with fixed single-precision polynomials (the coefficient sets
 * are the classic public-domain Cephes logf / exp2f minimax fits).  Every step is a plain
 * binary32 operation so the same sequence on any IEEE machine gives the same bits.
 * ------------------------------------------------------------------------------------------ */
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

float orc_log2(float x)
{
    if (!(x == x)) return x;                      /* NaN */
    if (x < 0.0f) return u2f(0x7FC00000u);        /* NaN */
    if (x == 0.0f) return u2f(0xFF800000u);       /* -inf */
    if (x == u2f(0x7F800000u)) return x;          /* +inf */
    int e = 0;
    if (x < u2f(0x00800000u)) { x = x * 16777216.0f; e = -24; }   /* denormal -> normal */
    uint32_t b = f2u(x);
    e += (int)((b >> 23) & 0xFFu) - 127;
    float m = u2f((b & 0x007FFFFFu) | 0x3F800000u);               /* [1, 2) */
    if (m > 1.41421356237f) { m = m * 0.5f; e += 1; }
    float f = m - 1.0f;
    float z = f * f;
    float p = 7.0376836292e-2f;
    p = sh_mul_add(p, f, -1.1514610310e-1f);
    p = sh_mul_add(p, f, 1.1676998740e-1f);
    p = sh_mul_add(p, f, -1.2420140846e-1f);
    p = sh_mul_add(p, f, 1.4249322787e-1f);
    p = sh_mul_add(p, f, -1.6668057665e-1f);
    p = sh_mul_add(p, f, 2.0000714765e-1f);
    p = sh_mul_add(p, f, -2.4999993993e-1f);
    p = sh_mul_add(p, f, 3.3333331174e-1f);
    float y = (p * f) * z;
    y = y - 0.5f * z;
    /* log2(m) = (f + y) * log2(e), split as log2(e) = 1 + 0.44269504088896340736 */
    float r = y * 0.44269504088896340736f;
#if ORC_FUSED
    r = fmaf(f, 0.44269504088896340736f, r);
#else
    r = r + f * 0.44269504088896340736f;
#endif
    r = r + y;
    r = r + f;
    r = r + (float)e;
    return r;
}

float orc_exp2(float x)
{
    if (!(x == x)) return x;
    if (x > 128.0f) return u2f(0x7F800000u);
    if (x < -150.0f) return 0.0f;
    float fl = floorf(x);
    int n = (int)fl;
    float r = x - fl;
    if (r > 0.5f) { n += 1; r = r - 1.0f; }
    float p = 1.535336188319500e-4f;
    p = sh_mul_add(p, r, 1.339887440266574e-3f);
    p = sh_mul_add(p, r, 9.618437357674640e-3f);
    p = sh_mul_add(p, r, 5.550332471162809e-2f);
    p = sh_mul_add(p, r, 2.402264791363012e-1f);
    p = sh_mul_add(p, r, 6.931472028550421e-1f);
    p = sh_mul_add(p, r, 1.0f);
    /* scale by 2^n in at most two exact-constant multiplies */
    if (n > 127) { p = p * u2f(0x7F000000u); n -= 127; }          /* 2^127 */
    if (n < -126) { p = p * u2f(0x00800000u); n += 126; }         /* 2^-126 */
    if (n < -126) n = -126;                                       /* x >= -150 keeps n >= -25 here */
    return p * u2f((uint32_t)(n + 127) << 23);
}

float orc_pow(float x, float y)
{
    if (x == 0.0f && y > 0.0f) return 0.0f;
    return orc_exp2(y * orc_log2(x));
}

/* ---------------------------------------------------------------------------------------------
 * RNG — Raytracing.comp:106-112.  The literal 4294967295.0 rounds to 2^32 in binary32, so the
 * divide is an exact scaling by 2^-32; results lie in [0, 1] inclusive.
 * ------------------------------------------------------------------------------------------ */
float orc_rng_next(uint32_t *state, uint32_t *raw_out)
{
    uint32_t s = *state * 747796405u + 2891336453u;
    *state = s;
    uint32_t r = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
    r = (r >> 22) ^ r;
    if (raw_out) *raw_out = r;
    return (float)r / 4294967296.0f;
}

/* Raytracing.comp:462 — float arithmetic for x + width*y, truncated to uint, plus frame stride */
uint32_t orc_pixel_seed(uint32_t x, uint32_t y, float width, uint32_t frame_index)
{
    float lin = (float)x + width * (float)y;
    return (uint32_t)lin + frame_index * 719393u;
}

/* Raytracing.comp:114-121 */
static inline v3 rng_direction(uint32_t *state)
{
    float x = orc_rng_next(state, NULL) * 2.0f - 1.0f;
    float y = orc_rng_next(state, NULL) * 2.0f - 1.0f;
    float z = orc_rng_next(state, NULL) * 2.0f - 1.0f;
    return v3_make(x, y, z);
}

/* ---------------------------------------------------------------------------------------------
 * Intersections
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int hit;
    v3 pos, nrm;
    float dist;
    orc_material mat;
    uint32_t prim;
} hit_info;

typedef struct { uint64_t c[ORC_N_COUNTERS]; } counters_t;

/* Raytracing.comp:130-161 */
static hit_info ray_sphere(v3 o, v3 d, const orc_sphere *s, counters_t *cn)
{
    hit_info h;
    memset(&h, 0, sizeof h);
    v3 c = v3_from(s->center);
    float radius = s->radius;
    v3 oc = v3_sub(o, c);
    float a = sh_dot(d, d);
    float b = 2.0f * sh_dot(oc, d);
    float cc = sh_dot(oc, oc) - radius * radius;
#if ORC_FUSED
    float disc = fmaf(b, b, -(4.0f * a * cc));
#else
    float disc = b * b - 4.0f * a * cc;
#endif
    if (cn) cn->c[ORC_N_SPHERE_TESTS]++;
    if (disc >= 0.0f) {
        if (cn) cn->c[ORC_N_SPHERE_DISC_GE0]++;
        float t = (-b - sqrtf(disc)) / (2.0f * a);
        if (t > 0.0001f) {
            v3 p = sh_along(o, d, t);
            h.hit = 1;
            h.pos = p;
            h.nrm = sh_normalize(v3_sub(p, c));
            h.dist = t;
            h.mat = s->material;
        }
    }
    return h;
}

/* Raytracing.comp:286-309 */
static hit_info ray_triangle(v3 o, v3 d, const orc_triangle *tr, counters_t *cn)
{
    hit_info h;
    memset(&h, 0, sizeof h);
    v3 p1 = v3_from(tr->p1), p2 = v3_from(tr->p2), p3 = v3_from(tr->p3);
    v3 ab = v3_sub(p2, p1);
    v3 ac = v3_sub(p3, p1);
    v3 n = sh_cross(ab, ac);
    v3 ao = v3_sub(o, p1);
    v3 dao = sh_cross(ao, d);
    float det = -sh_dot(d, n);
    float inv = 1.0f / det;
    float t = sh_dot(ao, n) * inv;
    float u = sh_dot(ac, dao) * inv;
    float v = -sh_dot(ab, dao) * inv;
    float w = 1.0f - u - v;
    if (cn) cn->c[ORC_N_TRI_TESTS]++;
    h.hit = det >= 1e-6f && t >= 0.0f && u >= 0.0f && v >= 0.0f && w >= 0.0f;
    h.pos = sh_along(o, d, t);
#if ORC_FUSED
    v3 n1 = v3_from(tr->n1), n2 = v3_from(tr->n2), n3 = v3_from(tr->n3);
    v3 bl = v3_make(fmaf(n3.x, v, fmaf(n2.x, u, n1.x * w)), fmaf(n3.y, v, fmaf(n2.y, u, n1.y * w)),
                    fmaf(n3.z, v, fmaf(n2.z, u, n1.z * w)));
#else
    v3 bl = v3_add(v3_add(v3_scale(v3_from(tr->n1), w), v3_scale(v3_from(tr->n2), u)),
                   v3_scale(v3_from(tr->n3), v));
#endif
    h.nrm = sh_normalize(bl);
    h.dist = t;
    return h;
}

/* Raytracing.comp:311-336 */
static int ray_box(v3 o, v3 d, v3 lo, v3 hi)
{
    float fx = 1.0f / d.x, fy = 1.0f / d.y, fz = 1.0f / d.z;
    float t1 = (lo.x - o.x) * fx;
    float t2 = (hi.x - o.x) * fx;
    float t3 = (lo.y - o.y) * fy;
    float t4 = (hi.y - o.y) * fy;
    float t5 = (lo.z - o.z) * fz;
    float t6 = (hi.z - o.z) * fz;
    float tmin = g_max(g_max(g_min(t1, t2), g_min(t3, t4)), g_min(t5, t6));
    float tmax = g_min(g_min(g_max(t1, t2), g_max(t3, t4)), g_max(t5, t6));
    if (tmax < 0.0f) return 0;
    if (tmin > tmax) return 0;
    return 1;
}

int orc_ray_sphere(const float o[3], const float d[3], const orc_sphere *s,
                   float *t, float pos[3], float nrm[3])
{
    hit_info h = ray_sphere(v3_from(o), v3_from(d), s, NULL);
    if (h.hit) {
        if (t) *t = h.dist;
        if (pos) { pos[0] = h.pos.x; pos[1] = h.pos.y; pos[2] = h.pos.z; }
        if (nrm) { nrm[0] = h.nrm.x; nrm[1] = h.nrm.y; nrm[2] = h.nrm.z; }
    }
    return h.hit;
}

int orc_ray_triangle(const float o[3], const float d[3], const orc_triangle *tri,
                     float *t, float pos[3], float nrm[3])
{
    hit_info h = ray_triangle(v3_from(o), v3_from(d), tri, NULL);
    if (t) *t = h.dist;
    if (pos) { pos[0] = h.pos.x; pos[1] = h.pos.y; pos[2] = h.pos.z; }
    if (nrm) { nrm[0] = h.nrm.x; nrm[1] = h.nrm.y; nrm[2] = h.nrm.z; }
    return h.hit;
}

int orc_ray_box(const float o[3], const float d[3], const float lo[3], const float hi[3])
{
    return ray_box(v3_from(o), v3_from(d), v3_from(lo), v3_from(hi));
}

/* Raytracing.comp:338-388 (the commented torus block 370-385 is dead and not restated) */
static hit_info closest_hit(const orc_frame_data *fd, v3 o, v3 d,
                            const orc_sphere *spheres, const orc_triangle *tris,
                            const orc_mesh *meshes, counters_t *cn)
{
    hit_info best;
    memset(&best, 0, sizeof best);
    best.hit = 0;
    best.dist = 3.402823466e+38f;
    best.prim = ORC_MISS_ID;
    for (uint32_t i = 0; i < fd->sphere_count; i++) {
        hit_info h = ray_sphere(o, d, &spheres[i], cn);
        if (h.hit && h.dist < best.dist) {
            best = h;
            best.prim = i;
        }
    }
    for (uint32_t i = 0; i < fd->mesh_count; i++) {
        const orc_mesh *m = &meshes[i];
        cn->c[ORC_N_BOX_TESTS]++;
        if (ray_box(o, d, v3_from(m->bound_lo), v3_from(m->bound_hi))) {
            for (uint32_t j = m->start_triangle; j < m->start_triangle + m->num_triangles; j++) {
                hit_info h = ray_triangle(o, d, &tris[j], cn);
                if (h.hit && h.dist < best.dist) {
                    best = h;
                    best.mat = m->material;
                    best.prim = fd->sphere_count + j;
                }
            }
        }
    }
    return best;
}

/* ClosestHit of n arbitrary rays: t (FLT_MAX on a miss) and primitive id per ray */
void orc_closest_hit(uint32_t sphere_count, uint32_t mesh_count,
                     const orc_sphere *spheres, const orc_triangle *tris, const orc_mesh *meshes,
                     const float *origins, const float *dirs, size_t n, float *t_out, uint32_t *prim_out)
{
    orc_frame_data fd;
    memset(&fd, 0, sizeof fd);
    fd.sphere_count = sphere_count;
    fd.mesh_count = mesh_count;
    /* rays are independent: threads split them (large scenes make one ray cost milliseconds) */
#pragma omp parallel for schedule(dynamic, 16)
    for (size_t i = 0; i < n; i++) {
        counters_t cn;
        memset(&cn, 0, sizeof cn);
        hit_info h = closest_hit(&fd, v3_from(origins + 3 * i), v3_from(dirs + 3 * i), spheres, tris, meshes, &cn);
        t_out[i] = h.dist;
        prim_out[i] = h.prim;
    }
}

/* Raytracing.comp:390-406.  Note the .xzy swizzle of the sun direction and the raw (not
 * normalised) ray direction. */
static v3 env_light(const orc_frame_data *fd, v3 d)
{
    v3 horizon = v3_from(fd->sky_horizon);
    v3 zenith = v3_from(fd->sky_zenith);
    v3 sun_dir = v3_make(fd->sun_dir[0], fd->sun_dir[2], fd->sun_dir[1]);
    v3 ground = v3_from(fd->ground);
    float grad_t = orc_pow(g_smoothstep(0.0f, 0.4f, d.y), 0.35f);
    v3 grad = sh_mix(horizon, zenith, grad_t);
    float sun = orc_pow(g_max(0.0f, sh_dot(d, v3_neg(sun_dir))), fd->sun_focus) * fd->sun_intensity;
    float g2s = g_smoothstep(-0.01f, 0.0f, d.y);
    float mask = (g2s >= 1.0f) ? 1.0f : 0.0f;
    v3 base = sh_mix(ground, grad, g2s);
    float add = sun * mask;
    return v3_make(base.x + add, base.y + add, base.z + add);
}

void orc_env_light(const orc_frame_data *fd, const float d[3], float rgb[3])
{
    v3 c = env_light(fd, v3_from(d));
    rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
}

/* The transmitting-material extension (no counterpart in the shader) is applied only when switched on; off, the
 * padding words of a material are ignored as the shader ignores them. */
static int g_transmission = 0;
void orc_set_transmission(int on) { g_transmission = on != 0; }

/* Raytracing.comp:408-439 */
static v3 trace(const orc_frame_data *fd, v3 o, v3 d, uint32_t *rng,
                const orc_sphere *spheres, const orc_triangle *tris, const orc_mesh *meshes,
                counters_t *cn, uint32_t *hash, uint32_t *first_prim)
{
    v3 light = v3_make(0.0f, 0.0f, 0.0f);
    v3 thr = v3_make(1.0f, 1.0f, 1.0f);
    cn->c[ORC_N_PATHS]++;
    for (uint32_t i = 0; i < fd->max_bounce; i++) {
        hit_info h = closest_hit(fd, o, d, spheres, tris, meshes, cn);
        cn->c[ORC_N_SEGMENTS]++;
        *hash = (*hash ^ h.prim) * 16777619u;
        if (i == 0 && first_prim) *first_prim = h.prim;
        if (h.hit) {
            cn->c[ORC_N_HIT_BOUNCES]++;
            if (h.prim < fd->sphere_count) cn->c[ORC_N_SPHERE_WINS]++;
            else cn->c[ORC_N_TRI_WINS]++;
            v3 rd = rng_direction(rng);
            if (g_transmission && h.mat.pad[0] > 0.5f) {
                /* EXTENSION, no counterpart in Raytracing.comp (its Material, :26-34, has colour, emission and
                 * smoothness only): a transmitting material, flagged by the first padding word, index of refraction
                 * in the second.  Every hit the shader's tests report is from outside (near sphere root only,
                 * :146-148; triangles front-facing only, :293-294), so the ray always ENTERS: it is mirrored with
                 * Schlick's reflectance as probability (the coin is the first of the three random numbers the
                 * branch draws anyway, so the stream stays in step with the other materials) and refracted
                 * otherwise; the refracted ray then leaves the object unseen, as the tests see no inside. */
                float ior = h.mat.pad[1];
                v3 dn = sh_normalize(d);
                float cosi = g_clamp(-sh_dot(dn, h.nrm), 0.0f, 1.0f);
                float eta = 1.0f / ior;
                float k = 1.0f - (eta * eta) * (1.0f - cosi * cosi);
                float r0 = (1.0f - ior) / (1.0f + ior);
                r0 = r0 * r0;
                float x = 1.0f - cosi;
                float x2 = x * x;
                float refl = r0 + (1.0f - r0) * ((x2 * x2) * x);
                o = h.pos;
                if (k < 0.0f || rd.x < refl * 2.0f - 1.0f) d = sh_reflect(dn, h.nrm);
                else d = sh_along(v3_scale(dn, eta), h.nrm, eta * cosi - sqrtf(k));
            } else {
                v3 diffuse = sh_normalize(v3_add(h.nrm, rd));
                v3 spec = sh_reflect(d, h.nrm);
                o = h.pos;
                d = sh_mix(diffuse, spec, h.mat.smoothness);
            }
            float e = 1.0f * h.mat.light;
            light = sh_add_mul(light, v3_make(e, e, e), thr);
            thr = v3_mul(thr, v3_from(h.mat.color));
        } else {
            cn->c[ORC_N_ENV_EVALS]++;
            light = sh_add_mul(light, env_light(fd, d), thr);
            break;
        }
    }
    return light;
}

/* Raytracing.comp:446-453 */
void orc_primary_ray(const orc_frame_data *fd, uint32_t x, uint32_t y, float o[3], float d[3])
{
    float width = fd->window[0], height = fd->window[1];
    float cx = ((float)x / width) * 2.0f - 1.0f;
    float cy = ((float)y / height) * 2.0f - 1.0f;
    float cv[4] = { cx, cy, 1.0f, 1.0f }, tgt[4];
    sh_mat4_mul_vec4(fd->inv_proj, cv, tgt);
    v3 n = sh_normalize(v3_make(tgt[0] / tgt[3], tgt[1] / tgt[3], tgt[2] / tgt[3]));
    float nv[4] = { n.x, n.y, n.z, 0.0f }, dv[4];
    sh_mat4_mul_vec4(fd->inv_view, nv, dv);
    o[0] = fd->cam_pos[0]; o[1] = fd->cam_pos[1]; o[2] = fd->cam_pos[2];
    d[0] = dv[0]; d[1] = dv[1]; d[2] = dv[2];
}

/* float -> UNORM8 of the rgba8 storage image: clamp to [0,1], scale by 255, round to nearest even;
 * NaN converts to 0. */
static inline uint8_t to_unorm8(float v)
{
    if (!(v == v)) return 0;
    float c = g_clamp(v, 0.0f, 1.0f);
    return (uint8_t)rintf(c * 255.0f);
}

/* Raytracing.comp:441-481 for every pixel of every requested frame */
int orc_render(const orc_frame_data *fd_in,
               const orc_sphere *spheres, const orc_triangle *tris, uint32_t n_tris,
               const orc_mesh *meshes,
               uint32_t first_frame, uint32_t n_frames,
               float *accum, uint8_t *out_rgba8, uint32_t *prim_id, uint32_t *chain_hash,
               uint32_t *rng_state, uint64_t *counters, int n_threads)
{
    if (!fd_in) return -1;
    return orc_render_rows(fd_in, spheres, tris, n_tris, meshes, first_frame, n_frames, accum, out_rgba8,
                           prim_id, chain_hash, rng_state, counters, n_threads, 0, (uint32_t)fd_in->window[1]);
}

/* Same, restricted to image rows y0 <= y < y1 (full-size buffers; other rows are left untouched).
 * A pixel's result does not depend on any other pixel, so this equals the same rows of a full render. */
int orc_render_rows(const orc_frame_data *fd_in,
                    const orc_sphere *spheres, const orc_triangle *tris, uint32_t n_tris,
                    const orc_mesh *meshes,
                    uint32_t first_frame, uint32_t n_frames,
                    float *accum, uint8_t *out_rgba8, uint32_t *prim_id, uint32_t *chain_hash,
                    uint32_t *rng_state, uint64_t *counters, int n_threads, uint32_t y0, uint32_t y1)
{
    (void)n_tris;
    if (!fd_in || !accum) return -1;
    const uint32_t W = (uint32_t)fd_in->window[0], H = (uint32_t)fd_in->window[1];
    if (W == 0 || H == 0 || y0 > y1 || y1 > H) return -1;
    counters_t total;
    memset(&total, 0, sizeof total);
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#else
    (void)n_threads;
#endif

#pragma omp parallel
    {
        counters_t cn;
        memset(&cn, 0, sizeof cn);
        orc_frame_data fd = *fd_in;
#pragma omp for schedule(dynamic, 4)
        for (uint32_t y = y0; y < y1; y++) {
            for (uint32_t x = 0; x < W; x++) {
                size_t px = (size_t)y * W + x;
                float of[3], df[3];
                orc_primary_ray(&fd, x, y, of, df);
                v3 o = v3_from(of), d = v3_from(df);
                uint32_t hash = 0u, first = ORC_MISS_ID, rng = 0;
                float sum[3] = { accum[4 * px + 0], accum[4 * px + 1], accum[4 * px + 2] };
                uint32_t fi = first_frame;
                for (uint32_t f = 0; f < n_frames; f++) {
                    fi = first_frame + f;
                    v3 light = v3_make(0.0f, 0.0f, 0.0f);
                    rng = orc_pixel_seed(x, y, fd.window[0], fi);
                    /* the frame's own hash: ids of all its segments in order, keyed by the frame index */
                    uint32_t frame_hash = 2166136261u ^ (fi * 0x9E3779B9u);
                    for (uint32_t k = 0; k < fd.rays_per_pixel; k++) {
                        rng += k;
                        uint32_t fp = ORC_MISS_ID;
                        v3 l = trace(&fd, o, d, &rng, spheres, tris, meshes, &cn, &frame_hash, &fp);
                        if (k == 0) first = fp;
                        light = v3_add(light, l);
                    }
                    hash += frame_hash;
                    float rpp = (float)fd.rays_per_pixel;
                    light = v3_make(light.x / rpp, light.y / rpp, light.z / rpp);
                    cn.c[ORC_N_PIXEL_FRAMES]++;
                    if (fi == 1u) {
                        sum[0] = light.x; sum[1] = light.y; sum[2] = light.z;
                    } else {
                        sum[0] = sum[0] + light.x; sum[1] = sum[1] + light.y; sum[2] = sum[2] + light.z;
                    }
                }
                if (n_frames > 0) {
                    accum[4 * px + 0] = sum[0];
                    accum[4 * px + 1] = sum[1];
                    accum[4 * px + 2] = sum[2];
                    accum[4 * px + 3] = 1.0f;
                    if (out_rgba8) {
                        float div = (float)fi;
                        out_rgba8[4 * px + 0] = to_unorm8(sum[0] / div);
                        out_rgba8[4 * px + 1] = to_unorm8(sum[1] / div);
                        out_rgba8[4 * px + 2] = to_unorm8(sum[2] / div);
                        out_rgba8[4 * px + 3] = 255;
                    }
                    if (prim_id) prim_id[px] = first;
                    if (chain_hash) chain_hash[px] = hash;
                    if (rng_state) rng_state[px] = rng;
                }
            }
        }
#pragma omp critical
        {
            for (int i = 0; i < ORC_N_COUNTERS; i++) total.c[i] += cn.c[i];
        }
    }
    if (counters)
        for (int i = 0; i < ORC_N_COUNTERS; i++) counters[i] += total.c[i];
    return 0;
}

/* ---------------------------------------------------------------------------------------------
 * Host math: glm call sites CameraFPS.cpp:15-25, 85-86 under GLM_FORCE_LEFT_HANDED and
 * GLM_FORCE_DEPTH_ZERO_TO_ONE (VKHeaders.h:10-11).  glm is not vendored in the reference; these
 * restate its published algorithms (perspectiveFovLH_ZO, lookAtLH, cofactor mat4 inverse).
 * ------------------------------------------------------------------------------------------ */
void orc_perspective_fov_lh_zo(float fov, float w, float h, float zn, float zf, float out[16])
{
    float hh = cosf(0.5f * fov) / sinf(0.5f * fov);
    float ww = hh * h / w;
    memset(out, 0, 16 * sizeof(float));
    out[0] = ww;
    out[5] = hh;
    out[10] = zf / (zf - zn);
    out[11] = 1.0f;
    out[14] = -(zf * zn) / (zf - zn);
}

void orc_look_at_lh(const float eye[3], const float center[3], const float up[3], float out[16])
{
    v3 e = v3_from(eye);
    v3 f = v3_normalize(v3_sub(v3_from(center), e));
    v3 s = v3_normalize(v3_cross(v3_from(up), f));
    v3 u = v3_cross(f, s);
    memset(out, 0, 16 * sizeof(float));
    out[0] = s.x; out[4] = s.y; out[8] = s.z;
    out[1] = u.x; out[5] = u.y; out[9] = u.z;
    out[2] = f.x; out[6] = f.y; out[10] = f.z;
    out[12] = -v3_dot(s, e);
    out[13] = -v3_dot(u, e);
    out[14] = -v3_dot(f, e);
    out[15] = 1.0f;
}

#define M(c, r) m[(c) * 4 + (r)]
void orc_mat4_inverse(const float m[16], float out[16])
{
    float c00 = M(2, 2) * M(3, 3) - M(3, 2) * M(2, 3);
    float c02 = M(1, 2) * M(3, 3) - M(3, 2) * M(1, 3);
    float c03 = M(1, 2) * M(2, 3) - M(2, 2) * M(1, 3);
    float c04 = M(2, 1) * M(3, 3) - M(3, 1) * M(2, 3);
    float c06 = M(1, 1) * M(3, 3) - M(3, 1) * M(1, 3);
    float c07 = M(1, 1) * M(2, 3) - M(2, 1) * M(1, 3);
    float c08 = M(2, 1) * M(3, 2) - M(3, 1) * M(2, 2);
    float c10 = M(1, 1) * M(3, 2) - M(3, 1) * M(1, 2);
    float c11 = M(1, 1) * M(2, 2) - M(2, 1) * M(1, 2);
    float c12 = M(2, 0) * M(3, 3) - M(3, 0) * M(2, 3);
    float c14 = M(1, 0) * M(3, 3) - M(3, 0) * M(1, 3);
    float c15 = M(1, 0) * M(2, 3) - M(2, 0) * M(1, 3);
    float c16 = M(2, 0) * M(3, 2) - M(3, 0) * M(2, 2);
    float c18 = M(1, 0) * M(3, 2) - M(3, 0) * M(1, 2);
    float c19 = M(1, 0) * M(2, 2) - M(2, 0) * M(1, 2);
    float c20 = M(2, 0) * M(3, 1) - M(3, 0) * M(2, 1);
    float c22 = M(1, 0) * M(3, 1) - M(3, 0) * M(1, 1);
    float c23 = M(1, 0) * M(2, 1) - M(2, 0) * M(1, 1);

    float fac[6][4] = {
        { c00, c00, c02, c03 }, { c04, c04, c06, c07 }, { c08, c08, c10, c11 },
        { c12, c12, c14, c15 }, { c16, c16, c18, c19 }, { c20, c20, c22, c23 },
    };
    float vec[4][4] = {
        { M(1, 0), M(0, 0), M(0, 0), M(0, 0) }, { M(1, 1), M(0, 1), M(0, 1), M(0, 1) },
        { M(1, 2), M(0, 2), M(0, 2), M(0, 2) }, { M(1, 3), M(0, 3), M(0, 3), M(0, 3) },
    };
    static const float sign_a[4] = { +1.0f, -1.0f, +1.0f, -1.0f };
    static const float sign_b[4] = { -1.0f, +1.0f, -1.0f, +1.0f };
    float inv[16];
    for (int i = 0; i < 4; i++) {
        float i0 = vec[1][i] * fac[0][i] - vec[2][i] * fac[1][i] + vec[3][i] * fac[2][i];
        float i1 = vec[0][i] * fac[0][i] - vec[2][i] * fac[3][i] + vec[3][i] * fac[4][i];
        float i2 = vec[0][i] * fac[1][i] - vec[1][i] * fac[3][i] + vec[3][i] * fac[5][i];
        float i3 = vec[0][i] * fac[2][i] - vec[1][i] * fac[4][i] + vec[2][i] * fac[5][i];
        inv[0 * 4 + i] = i0 * sign_a[i];
        inv[1 * 4 + i] = i1 * sign_b[i];
        inv[2 * 4 + i] = i2 * sign_a[i];
        inv[3 * 4 + i] = i3 * sign_b[i];
    }
    float d0 = M(0, 0) * inv[0 * 4 + 0], d1 = M(0, 1) * inv[1 * 4 + 0];
    float d2 = M(0, 2) * inv[2 * 4 + 0], d3 = M(0, 3) * inv[3 * 4 + 0];
    float det = (d0 + d1) + (d2 + d3);
    float ood = 1.0f / det;
    for (int i = 0; i < 16; i++) out[i] = inv[i] * ood;
}
#undef M

/* CameraFPS.cpp:15-19: forward from yaw/pitch in degrees, glm::radians then normalize */
void orc_camera_forward(float yaw_deg, float pitch_deg, float fwd[3])
{
    const float k = 0.01745329251994329576923690768489f;
    float yaw = yaw_deg * k, pitch = pitch_deg * k;
    v3 f = v3_normalize(v3_make(cosf(yaw) * cosf(pitch), sinf(pitch), sinf(yaw) * cosf(pitch)));
    fwd[0] = f.x; fwd[1] = f.y; fwd[2] = f.z;
}

/* CameraFPS.cpp:13-25 + main.cpp:355-360.  fov = 70.0f is handed to perspectiveFov as radians. */
void orc_camera_pose(orc_frame_data *fd, const float pos[3], float yaw_deg, float pitch_deg,
                     uint32_t width, uint32_t height)
{
    float fwd[3], center[3], view[16], proj[16];
    const float up[3] = { 0.0f, 1.0f, 0.0f };
    orc_camera_forward(yaw_deg, pitch_deg, fwd);
    for (int i = 0; i < 3; i++) center[i] = pos[i] + fwd[i];
    orc_look_at_lh(pos, center, up, view);
    orc_perspective_fov_lh_zo(70.0f, (float)width, (float)height, 0.1f, 1000.0f, proj);
    orc_mat4_inverse(proj, fd->inv_proj);
    orc_mat4_inverse(view, fd->inv_view);
    for (int i = 0; i < 3; i++) { fd->cam_pos[i] = pos[i]; fd->cam_dir[i] = fwd[i]; }
    fd->cam_pos[3] = 0.0f;
    fd->cam_dir[3] = 0.0f;
    fd->window[0] = (float)width;
    fd->window[1] = (float)height;
}

/* main.cpp:116-123, 151-153 */
void orc_default_frame_constants(orc_frame_data *fd)
{
    fd->rays_per_pixel = 4;
    fd->max_bounce = 6;
    fd->frame_index = 0;
    const float sun[4] = { -0.4f, -0.4f, -0.4f, 0.0f };
    const float hz[4] = { 0.7f, 0.3f, 0.1f, 0.0f };
    const float zn[4] = { 0.2f, 0.56f, 0.95f, 0.0f };
    const float gr[4] = { 0.9f, 0.9f, 0.9f, 0.0f };
    memcpy(fd->sun_dir, sun, sizeof sun);
    memcpy(fd->sky_horizon, hz, sizeof hz);
    memcpy(fd->sky_zenith, zn, sizeof zn);
    memcpy(fd->ground, gr, sizeof gr);
    fd->sun_focus = 1.0f;
    fd->sun_intensity = 1.0f;
}

/* ---------------------------------------------------------------------------------------------
 * OBJ loader — ModelLoader.cpp:31-142.  The reference parses with tinyobjloader (not vendored,
 * version unknown) with triangulation on; this minimal reader handles v / vn / f records and
 * triangulates quads by the rule selected.  Positions are scaled then translated per component
 * (one multiply, one add), normals are copied untouched, the box is seeded at +-1024.
 * ------------------------------------------------------------------------------------------ */
typedef struct { float *v; size_t n, cap; } fvec;
static int fvec_push3(fvec *a, float x, float y, float z)
{
    if (a->n + 3 > a->cap) {
        size_t nc = a->cap ? a->cap * 2 : 96;
        float *nv = (float *)realloc(a->v, nc * sizeof(float));
        if (!nv) return -1;
        a->v = nv; a->cap = nc;
    }
    a->v[a->n++] = x; a->v[a->n++] = y; a->v[a->n++] = z;
    return 0;
}

static int parse_face_vertex(const char **pp, long nv, long nn, long *vi, long *ni)
{
    char *end;
    const char *p = *pp;
    long a = strtol(p, &end, 10), c = 0;
    if (end == p) return -1;
    p = end;
    if (*p == '/') {
        p++;
        if (*p != '/') { (void)strtol(p, &end, 10); p = end; }
        if (*p == '/') { p++; c = strtol(p, &end, 10); p = end; }
    }
    *vi = a > 0 ? a - 1 : nv + a;
    *ni = c > 0 ? c - 1 : (c < 0 ? nn + c : -1);
    *pp = p;
    return 0;
}

static float sqr_dist(const float *a, const float *b)
{
    float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    return dx * dx + dy * dy + dz * dz;
}

int orc_load_model(const char *path, orc_triangle *tris, uint32_t *n_tris, uint32_t tri_cap,
                   orc_mesh *meshes, uint32_t *n_meshes, uint32_t mesh_cap,
                   const orc_material *mat, const float translate[3], const float scale[3],
                   int quad_rule)
{
    FILE *fp = fopen(path, "r");
    if (!fp) return -1;
    fvec pos = { 0 }, nrm = { 0 };
    float lo[3] = { 1024.0f, 1024.0f, 1024.0f }, hi[3] = { -1024.0f, -1024.0f, -1024.0f };
    uint32_t start = *n_tris, count = 0;
    char line[1024];
    int rc = 0;
    while (fgets(line, sizeof line, fp)) {
        const char *p = line;
        while (*p == ' ' || *p == '\t') p++;
        if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) {
            char *e;
            double x = strtod(p + 2, &e), y = strtod(e, &e), z = strtod(e, &e);
            if (fvec_push3(&pos, (float)x, (float)y, (float)z)) { rc = -1; break; }
        } else if (p[0] == 'v' && p[1] == 'n') {
            char *e;
            double x = strtod(p + 3, &e), y = strtod(e, &e), z = strtod(e, &e);
            if (fvec_push3(&nrm, (float)x, (float)y, (float)z)) { rc = -1; break; }
        } else if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
            long vi[64], ni[64];
            int k = 0;
            p += 2;
            for (;;) {
                while (*p == ' ' || *p == '\t') p++;
                if (*p == '\0' || *p == '\n' || *p == '\r' || k == 64) break;
                if (parse_face_vertex(&p, (long)(pos.n / 3), (long)(nrm.n / 3), &vi[k], &ni[k])) break;
                k++;
            }
            if (k < 3) continue;
            int order[62][3], nt = 0;
            if (k == 4 && quad_rule == 0) {
                float d02 = sqr_dist(&pos.v[3 * vi[0]], &pos.v[3 * vi[2]]);
                float d13 = sqr_dist(&pos.v[3 * vi[1]], &pos.v[3 * vi[3]]);
                if (d02 < d13) {
                    int t0[3] = { 0, 1, 2 }, t1[3] = { 0, 2, 3 };
                    memcpy(order[0], t0, sizeof t0); memcpy(order[1], t1, sizeof t1);
                } else {
                    int t0[3] = { 0, 1, 3 }, t1[3] = { 1, 2, 3 };
                    memcpy(order[0], t0, sizeof t0); memcpy(order[1], t1, sizeof t1);
                }
                nt = 2;
            } else {
                for (int i = 1; i + 1 < k; i++) {
                    order[nt][0] = 0; order[nt][1] = i; order[nt][2] = i + 1;
                    nt++;
                }
            }
            for (int t = 0; t < nt; t++) {
                if (*n_tris >= tri_cap) { rc = -1; break; }
                orc_triangle tr;
                memset(&tr, 0, sizeof tr);
                float *pp[3] = { tr.p1, tr.p2, tr.p3 }, *nn[3] = { tr.n1, tr.n2, tr.n3 };
                for (int c = 0; c < 3; c++) {
                    long v = vi[order[t][c]], n = ni[order[t][c]];
                    if (v < 0 || (size_t)(3 * v + 2) >= pos.n) { rc = -1; break; }
                    for (int a = 0; a < 3; a++) {
                        pp[c][a] = pos.v[3 * v + a] * scale[a] + translate[a];
                        nn[c][a] = (n >= 0 && (size_t)(3 * n + 2) < nrm.n) ? nrm.v[3 * n + a] : 0.0f;
                        if (pp[c][a] < lo[a]) lo[a] = pp[c][a];
                        if (pp[c][a] > hi[a]) hi[a] = pp[c][a];
                    }
                }
                if (rc) break;
                tris[(*n_tris)++] = tr;
                count++;
            }
            if (rc) break;
        }
    }
    fclose(fp);
    free(pos.v);
    free(nrm.v);
    if (rc) return rc;
    if (*n_meshes >= mesh_cap) return -1;
    orc_mesh m;
    memset(&m, 0, sizeof m);
    m.start_triangle = start;
    m.num_triangles = count;
    for (int a = 0; a < 3; a++) { m.bound_lo[a] = lo[a]; m.bound_hi[a] = hi[a]; }
    m.material = *mat;
    meshes[(*n_meshes)++] = m;
    return (int)count;
}
