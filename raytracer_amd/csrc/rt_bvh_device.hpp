// rt_bvh_device.hpp — closest hit through a bounding volume hierarchy, for scenes whose primitive
// counts make the uniform loops of ClosestHit (Raytracing.comp:338-388) impractical.
//
// The hierarchy only selects which primitives receive the shader's tests; the tests themselves are
// the literal per-primitive sequences, and the winner is chosen by the rule the shader's loop order
// implies: smallest t, and among equal t the first primitive in the shader's visiting order (spheres
// by index, then meshes in order, triangles by index within a mesh).
//
// A box may only be skipped when no primitive inside can pass its test with a t that would win.  The
// tests are evaluated in rounded arithmetic, so a ray can "hit" a primitive it geometrically misses,
// by a margin that grows with the distance between ray origin and primitive:
//   sphere    following the roundings of b, c and the discriminant one by one, the discriminant's
//             error stays below 76 ulp of |oc|^2 |d|^2, i.e. below 19 * 2^-24 |oc|^2 in units of
//             r^2 - p^2 (p the distance of the centre from the ray).  The test therefore behaves like an
//             exact test against a sphere of radius sqrt(r^2 + kappa |oc|^2), kappa = 1.4e-6, and the t
//             it returns is not below the entry into that larger sphere.
//   triangle  the numerators of u, v carry errors of about 8 ulp of |ao| |edge| |d|; in the triangle's
//             plane that is a strip of width 8 * 2^-24 |ao| / cos(phi) around the triangle (phi the
//             angle between ray and normal).  A margin of 2^-12 |ao| covers every ray that is more
//             than 0.12 degrees off the triangle's plane; flatter rays are culled by the det >= 1e-6
//             test for all but tiny triangles.
// Every box test grows the box by margin = sqrt(f^2 + kappa D^2) - f, where D^2 bounds |oc|^2 (|ao|^2)
// for everything in the box (sum over the axes of the squared largest distance between ray origin
// and box) and f is the smallest radius below the node (0 for triangles), plus 2^-19 of the largest
// distance component for the rounding of the slab arithmetic itself.  The comparison of a box's
// entry distance with the best t so far allows 2^-7 relative slack for the error of t.
#pragma once

#include "rt_device.hpp"

namespace rtd {

constexpr uint32_t kBvhLeafFlag = 0x80000000u;
constexpr uint32_t kBvhEmpty = 0xFFFFFFFFu;
constexpr int kBvhStack = 24;  // one entry per level of inner nodes; the builder stops at 24 levels

constexpr float kBvhSphereKappa = 1.4e-6f;             // > 19 * 2^-24
constexpr float kBvhTriangleKappa = 5.9604644775e-8f;  // 2^-24: margin 2^-12 |ao|
constexpr float kBvhTSlack = 1.0078125f;               // 1 + 2^-7

// Entry distance of the ray into the grown box; false when the ray cannot reach anything inside that
// could win.  NaN (0 * inf, inf - inf) makes a comparison false, which keeps the box.
__device__ __forceinline__ bool bvh_box(const float4 lo, const float4 hi, float floor_r, V3 o, V3 inv, float kappa,
                                        float t_limit, float &t_near)
{
    const float lx = lo.x - o.x, hx = hi.x - o.x;
    const float ly = lo.y - o.y, hy = hi.y - o.y;
    const float lz = lo.z - o.z, hz = hi.z - o.z;
    const float mx = fmaxf(fabsf(lx), fabsf(hx)), my = fmaxf(fabsf(ly), fabsf(hy)), mz = fmaxf(fabsf(lz), fabsf(hz));
    const float d2 = mx * mx + my * my + mz * mz;
    // the hardware square root is good to 1 ulp; the factor 1 + 2^-20 keeps the result an upper bound
    const float grown = __builtin_amdgcn_sqrtf(floor_r * floor_r + kappa * d2) * 1.00000095367431640625f;
    const float e = (grown - floor_r) + 1.9073486328125e-6f * fmaxf(fmaxf(mx, my), mz);
    const float t1 = (lx - e) * inv.x, t2 = (hx + e) * inv.x;
    const float t3 = (ly - e) * inv.y, t4 = (hy + e) * inv.y;
    const float t5 = (lz - e) * inv.z, t6 = (hz + e) * inv.z;
    const float tmin = fmaxf(fmaxf(fminf(t1, t2), fminf(t3, t4)), fminf(t5, t6));
    const float tmax = fminf(fminf(fmaxf(t1, t2), fmaxf(t3, t4)), fmaxf(t5, t6));
    t_near = tmin;
    return !(tmax < 0.0f) && !(tmin > tmax) && !(tmin > t_limit);
}

// Walks one hierarchy; leaf(first, count) tests the leaf-order entries [first, first + count) and
// updates best.  stack points at this lane's first slot; consecutive levels are stride words apart.
template <bool COUNT, class Leaf>
__device__ __forceinline__ void bvh_walk(const float4 *__restrict__ nodes, uint32_t root, V3 o, V3 inv, float kappa,
                                         const Hit &best, uint32_t *stack, uint32_t stride, Counters &cn, Leaf &&leaf)
{
    if (root == kBvhEmpty) return;
    uint32_t ref = root;
    int sp = 0;
    for (;;) {
        if (ref & kBvhLeafFlag) {
            leaf(ref & 0x0FFFFFFFu, (ref >> 28) & 7u);
            if (sp == 0) break;
            ref = stack[(uint32_t)(--sp) * stride];
        } else {
            const float4 q0 = nodes[4 * (size_t)ref + 0];
            const float4 q1 = nodes[4 * (size_t)ref + 1];
            const float4 q2 = nodes[4 * (size_t)ref + 2];
            const float4 q3 = nodes[4 * (size_t)ref + 3];
            if (COUNT) cn.c[3] += 2u;
            const float t_limit = best.t * kBvhTSlack;
            float tl, tr;
            const bool hl = bvh_box(q0, q1, q2.w, o, inv, kappa, t_limit, tl);
            const bool hr = bvh_box(q2, q3, q3.w, o, inv, kappa, t_limit, tr);
            const uint32_t L = __float_as_uint(q0.w), R = __float_as_uint(q1.w);
            if (hl && hr) {
                const bool right_first = tr < tl;
                if (sp < kBvhStack) stack[(uint32_t)(sp++) * stride] = right_first ? L : R;
                ref = right_first ? R : L;
            } else if (hl) {
                ref = L;
            } else if (hr) {
                ref = R;
            } else {
                if (sp == 0) break;
                ref = stack[(uint32_t)(--sp) * stride];
            }
        }
    }
}

// ClosestHit (Raytracing.comp:338-388) over the hierarchies of the uploaded scene.
template <bool COUNT>
__device__ __forceinline__ Hit closest_hit_bvh(const DeviceScene &sc, uint32_t n_spheres, uint32_t n_meshes, V3 o, V3 d,
                                               Counters &cn, uint32_t *stack, uint32_t stride)
{
    Hit best;
    best.t = kFltMax;
    best.prim = kMissId;
    best.mat = 0;
    best.u = 0.0f;
    best.v = 0.0f;
    const V3 inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);

    // RaySphere, Raytracing.comp:130-161, for the spheres of the leaves the ray reaches
    if (n_spheres > 0) {
        const float a = dot(d, d);
        const float a4 = 4.0f * a;
        const float a2 = 2.0f * a;
        bvh_walk<COUNT>(sc.bvh_nodes, sc.bvh_sphere_root, o, inv, kBvhSphereKappa, best, stack, stride, cn,
                        [&](uint32_t first, uint32_t count) {
                            for (uint32_t k = 0; k < count; k++) {
                                const uint32_t id = sc.bvh_sphere_id[first + k];
                                if (id >= n_spheres) continue;  // not among the spheres this frame names
                                const float4 g = sc.bvh_sphere_geom[first + k];
                                const V3 oc = o - mk(g.x, g.y, g.z);
                                const float b = 2.0f * dot(oc, d);
                                const float c = dot(oc, oc) - g.w;
                                const float disc = b * b - a4 * c;
                                if (COUNT) {
                                    cn.c[0]++;
                                    if (disc >= 0.0f) cn.c[1]++;
                                }
                                if (disc >= 0.0f) {
                                    const float t = (-b - sqrtf(disc)) / a2;
                                    // the shader's loop keeps the lowest index among equal t
                                    if (t > 0.0001f && (t < best.t || (t == best.t && id < best.prim && best.prim != kMissId))) {
                                        best.t = t;
                                        best.prim = id;
                                        best.mat = id;
                                    }
                                }
                            }
                        });
    }

    // RayBox gate (Raytracing.comp:311-336, 352-356) exactly as the shader evaluates it, then
    // RayTriangle (Raytracing.comp:286-309) for the triangles of the leaves the ray reaches
    for (uint32_t m = 0; m < n_meshes; m++) {
        const float4 lo = ld_uniform(sc.mesh_box, 2 * m + 0);
        const float4 hi = ld_uniform(sc.mesh_box, 2 * m + 1);
        const float t1 = (lo.x - o.x) * inv.x;
        const float t2 = (hi.x - o.x) * inv.x;
        const float t3 = (lo.y - o.y) * inv.y;
        const float t4 = (hi.y - o.y) * inv.y;
        const float t5 = (lo.z - o.z) * inv.z;
        const float t6 = (hi.z - o.z) * inv.z;
        const float tmin = gmax(gmax(gmin(t1, t2), gmin(t3, t4)), gmin(t5, t6));
        const float tmax = gmin(gmin(gmax(t1, t2), gmax(t3, t4)), gmax(t5, t6));
        if (COUNT) cn.c[3]++;
        const bool inside = !(tmax < 0.0f) && !(tmin > tmax);
        if (!inside) continue;
        // a triangle of this mesh replaces an equal-t winner only if that winner is a higher-indexed
        // triangle of this same mesh: earlier spheres and meshes keep their ties
        bool own_best = false;
        bvh_walk<COUNT>(sc.bvh_nodes, sc.bvh_mesh_root[m], o, inv, kBvhTriangleKappa, best, stack, stride, cn,
                        [&](uint32_t first, uint32_t count) {
                            for (uint32_t k = 0; k < count; k++) {
                                const size_t e = (size_t)first + k;
                                const float4 P = sc.bvh_tri_isect[4 * e + 0];
                                const float4 AB = sc.bvh_tri_isect[4 * e + 1];
                                const float4 AC = sc.bvh_tri_isect[4 * e + 2];
                                const float4 N = sc.bvh_tri_isect[4 * e + 3];
                                if (COUNT) cn.c[4]++;
                                const V3 nrm = mk(N.x, N.y, N.z);
                                const float det = -dot(d, nrm);
                                if (det >= 1e-6f) {
                                    const V3 ao = o - mk(P.x, P.y, P.z);
                                    const V3 dao = cross(ao, d);
                                    const float inv_det = 1.0f / det;
                                    const float t = dot(ao, nrm) * inv_det;
                                    const float u = dot(mk(AC.x, AC.y, AC.z), dao) * inv_det;
                                    const float v = -dot(mk(AB.x, AB.y, AB.z), dao) * inv_det;
                                    const float w = 1.0f - u - v;
                                    if (t >= 0.0f && u >= 0.0f && v >= 0.0f && w >= 0.0f) {
                                        const uint32_t prim = n_spheres + sc.bvh_tri_id[e];
                                        if (t < best.t || (own_best && t == best.t && prim < best.prim)) {
                                            best.t = t;
                                            best.prim = prim;
                                            best.mat = sc.n_spheres + m;
                                            best.u = u;
                                            best.v = v;
                                            own_best = true;
                                        }
                                    }
                                }
                            }
                        });
    }
    return best;
}

}  // namespace rtd
