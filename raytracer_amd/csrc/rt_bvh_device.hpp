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
//             than a fraction of a degree off the triangle's plane; flatter rays get a larger margin
//             (bvh_graze_margin below).
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

// ClosestHit (Raytracing.comp:338-388) over the hierarchies of the uploaded scene, as a resumable
// per-lane walk.
//
// A lane walks, in the shader's group order, the sphere hierarchy and then the hierarchy of every mesh
// whose box gate its ray passes.  Each step does ONE thing for the lane: test the two child boxes of
// an inner node, or test one primitive of a leaf, or move on to the lane's next group.  All state is
// in the struct (and the lane's stack slots), so a caller may stop after any step, do other work with
// the lanes that are done, start new rays in them, and come back: lanes of a wave are then at
// different places of different hierarchies, for different rays, at the same time.
struct BvhWalk {
    Hit best;
    V3 inv;                  // 1 / d per component (Raytracing.comp:313-315)
    float a;                 // dot(d, d)
    float gm;                // extra margin of the current mesh for this ray, per 2^-12 of the distance (0 in the sphere hierarchy)
    unsigned long long gates;  // bit b: the ray enters mesh gate_base + b and has not walked it yet
    uint32_t gate_base;      // first mesh of the current chunk of 64 gates
    uint32_t cur_mesh, ref, leaf_pos, leaf_left;
    int sp;
    // per-lane flags in one register (as separate bools they would each occupy a lane-mask register pair)
    uint32_t flags;
    static constexpr uint32_t kActive = 1u;          // the walk has work left
    static constexpr uint32_t kWalking = 2u;         // ref / leaf_left describe work inside the current group
    static constexpr uint32_t kInSpheres = 8u;       // the current group is the sphere hierarchy
    static constexpr uint32_t kOwnBest = 16u;        // best is a triangle of the current mesh
    static constexpr uint32_t kGrazeNodes = 32u;     // the current mesh holds large triangles and the ray may run along
                                                     // one: the children's own cones and reaches decide, node by node
    __device__ __forceinline__ bool is(uint32_t f) const { return (flags & f) != 0u; }
    __device__ __forceinline__ void set(uint32_t f) { flags |= f; }
    __device__ __forceinline__ void clear(uint32_t f) { flags &= ~f; }
    __device__ __forceinline__ bool active() const { return is(kActive); }
};

// RayBox gates (Raytracing.comp:311-336, 352-356) of meshes base .. base + 63, exactly as the shader
// evaluates them.  base may differ between lanes unless UNIFORM says it does not (the boxes then come
// through scalar fetches).
template <bool COUNT, bool UNIFORM = false>
__device__ __forceinline__ unsigned long long bvh_gates(const DeviceScene &sc, uint32_t base, uint32_t n_meshes, V3 o, V3 inv,
                                                        Counters &cn)
{
    unsigned long long gates = 0ull;
    const uint32_t chunk_end = (n_meshes - base < 64u) ? n_meshes : base + 64u;
    for (uint32_t m = base; m < chunk_end; m++) {
        const float4 lo = UNIFORM ? ld_uniform(sc.mesh_box, 2 * m + 0) : sc.mesh_box[2 * m + 0];
        const float4 hi = UNIFORM ? ld_uniform(sc.mesh_box, 2 * m + 1) : sc.mesh_box[2 * m + 1];
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
        gates |= inside ? (1ull << (m - base)) : 0ull;
    }
    return gates;
}

// a walk with nothing to do
__device__ __forceinline__ void bvh_idle(BvhWalk &w)
{
    w.best.t = kFltMax;
    w.best.prim = kMissId;
    w.best.mat = 0;
    w.best.u = 0.0f;
    w.best.v = 0.0f;
    w.inv = mk(0.0f, 0.0f, 0.0f);
    w.a = 0.0f;
    w.gm = 0.0f;
    w.gates = 0ull;
    w.gate_base = 0u;
    w.cur_mesh = 0u;
    w.ref = kBvhEmpty;
    w.leaf_pos = 0u;
    w.leaf_left = 0u;
    w.sp = 0;
    w.flags = 0u;
}

template <bool COUNT>
__device__ __forceinline__ void bvh_begin(BvhWalk &w, const DeviceScene &sc, uint32_t n_spheres, uint32_t n_meshes, V3 o, V3 d,
                                          Counters &cn)
{
    w.best.t = kFltMax;
    w.best.prim = kMissId;
    w.best.mat = 0;
    w.best.u = 0.0f;
    w.best.v = 0.0f;
    w.inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    w.a = dot(d, d);
    w.gm = 0.0f;
    w.gate_base = 0u;
    w.gates = bvh_gates<COUNT, true>(sc, 0u, n_meshes, o, w.inv, cn);
    w.cur_mesh = 0u;
    w.leaf_pos = 0u;
    w.leaf_left = 0u;
    w.sp = 0;
    if (n_spheres > 0u && sc.bvh_sphere_root != kBvhEmpty) {
        // the first group: the sphere hierarchy
        w.flags = BvhWalk::kActive | BvhWalk::kWalking | BvhWalk::kInSpheres;
        w.ref = sc.bvh_sphere_root;
    } else {
        w.flags = (w.gates != 0ull || n_meshes > 64u) ? BvhWalk::kActive : 0u;
        w.ref = kBvhEmpty;
    }
}

// The lane has finished its current group; the walk is over when no gated mesh is left either.
__device__ __forceinline__ void bvh_group_done(BvhWalk &w, uint32_t n_meshes)
{
    w.clear(BvhWalk::kWalking);
    if (w.gates == 0ull && n_meshes - w.gate_base <= 64u) w.clear(BvhWalk::kActive);
}

// true when the next thing the lane's walk needs is an inner node (both child boxes)
__device__ __forceinline__ bool bvh_at_inner(const BvhWalk &w)
{
    return w.is(BvhWalk::kWalking) && w.leaf_left == 0u && (w.ref & kBvhLeafFlag) == 0u;
}

// One inner node for a lane with bvh_at_inner(): both child boxes, then on to the nearer child that
// the ray can reach (the other one, if reachable too, goes on the stack).  stack points at this lane's
// first slot; consecutive levels are stride words apart.
//
// A node is 32 bytes: two child references and twelve 16-bit plane positions q in the scene's frame,
// plane = origin + q * step.  A count becomes a float with one conversion that also picks its half of the
// word (v_cvt_f32_u32 with a word select), and one fused multiply-add gives the plane relative to the ray
// origin: q * step + (origin - o).
// The upload chose q so that the quantised box encloses the child's box in exact arithmetic; the two
// roundings here (of origin - o, of the fused result) move a plane by less than
// 2^-24 (|origin - o| + |plane - o|) <= 2^-24 (2 |plane - o| + 65535 step), which 2^-22 of the largest
// distance component and sc.bvh_slack cover.
//
// Both boxes are grown by ONE margin, computed for their union: D^2 over the union bounds |oc|^2 for
// everything below either child, and the smallest radius in the hierarchy gives the largest margin.
// The margin arithmetic is not part of the shader's arithmetic (it only has to be an upper bound), so
// it uses fused multiply-adds.  A child flagged unbounded (an infinite box) is visited whatever the ray.
constexpr uint32_t kBvhUnbounded = 0x08000000u;
constexpr uint32_t kBvhIndexMask = 0x07FFFFFFu;

// Rays that run almost inside the plane of a triangle (host/rt_bvh.hpp has the derivation).  A set of
// triangles (a mesh, or what lies below one child of a node) is described by a cone word (the axis of a cone
// around their normals as three int8, and a code byte for its width) and a reach (eps 1e6 G of the largest
// triangle of the set).  A ray outside the cone's band cannot be that flat to any triangle of the set: the
// usual margin covers it.  Inside the band the rounding of RayTriangle (Raytracing.comp:286-309) can accept the
// ray up to 26 g / (1 - 2.25 g) of the distance away from a triangle, g = reach |d|; from g = 0.05 on
// (triangles whose edges are about half a unit and longer) that is the whole distance and more.
__device__ __forceinline__ bool bvh_in_band(uint32_t word, V3 d, float dd)
{
    const float ax = (float)((int)(word << 24) >> 24), ay = (float)((int)(word << 16) >> 24), az = (float)((int)(word << 8) >> 24);
    const uint32_t code = word >> 24;
    const float dq = __builtin_fmaf(d.z, az, __builtin_fmaf(d.y, ay, d.x * ax));
    const float T = __uint_as_float((code + 992u) << 20);
    // NaN compares false: such a ray counts as inside the band
    return code == 255u || !(dq * dq > T * dd);
}

// The ray enters a mesh's hierarchy: what it needs there.  Every triangle small (g < 0.05): ONE extra margin
// for every box of the hierarchy, min(30 g, cap) of the distance, returned per 2^-12 of the distance (the
// unit in which bvh_inner has the distance at hand); node_by_node stays false.  Otherwise the children's own
// cones and reaches are consulted at every node (bvh_graze_child), so that a few large triangles do not open
// the whole hierarchy to every ray.
__device__ __forceinline__ float bvh_graze_enter(uint32_t word, float reach, V3 d, float dd, float cap, bool &node_by_node)
{
    node_by_node = false;
    if (!bvh_in_band(word, d, dd)) return 0.0f;
    // upper bound of |d| (hardware square root: 1 ulp)
    const float g = reach * (__builtin_amdgcn_sqrtf(dd) * 1.000001f);
    if (!(g < 0.05f)) {   // also when g is not a number
        node_by_node = true;
        return 0.0f;
    }
    return fminf(30.0f * g, cap) * 4096.01f;
}

// One child of a node of a mesh that is walked node by node: its margin e_child (the usual e outside the band),
// or `always`: visited whatever its box says.
__device__ __forceinline__ void bvh_graze_child(uint32_t word, float reach, V3 d, float dd, float dlen, float dist, float cap, float e,
                                                float &e_child, bool &always)
{
    e_child = e;
    always = false;
    if (bvh_in_band(word, d, dd)) {
        const float g = reach * dlen;
        if (!(g < 0.05f)) always = true;
        else e_child = __builtin_fmaf(fminf(30.0f * g, cap), dist, e);
    }
}

template <bool COUNT>
__device__ __forceinline__ void bvh_inner(BvhWalk &w, const DeviceScene &sc, uint32_t n_meshes, V3 o, V3 d, Counters &cn, uint32_t *stack,
                                          uint32_t stride)
{
    const uint4 *__restrict__ rec = sc.bvh_nodes + 2 * (size_t)(w.ref & kBvhIndexMask);
    const uint4 n0 = rec[0], n1 = rec[1];
    const bool TRI = !w.is(BvhWalk::kInSpheres);
    if (COUNT) cn.c[3] += 2u;
    const float kappa = TRI ? kBvhTriangleKappa : kBvhSphereKappa;
    const float floor_r = TRI ? 0.0f : sc.bvh_sphere_floor;
    const float t_limit = w.best.t * kBvhTSlack;
    const float cx = sc.bvh_base[0] - o.x, cy = sc.bvh_base[1] - o.y, cz = sc.bvh_base[2] - o.z;
    const float sx = sc.bvh_step[0], sy = sc.bvh_step[1], sz = sc.bvh_step[2];
    auto lower = [](uint32_t packed) { return (float)(packed & 0xFFFFu); };
    auto upper = [](uint32_t packed) { return (float)(packed >> 16); };
    const float l0x = __builtin_fmaf(lower(n0.z), sx, cx), l1x = __builtin_fmaf(upper(n0.z), sx, cx);
    const float l0y = __builtin_fmaf(lower(n0.w), sy, cy), l1y = __builtin_fmaf(upper(n0.w), sy, cy);
    const float l0z = __builtin_fmaf(lower(n1.x), sz, cz), l1z = __builtin_fmaf(upper(n1.x), sz, cz);
    const float r0x = __builtin_fmaf(lower(n1.y), sx, cx), r1x = __builtin_fmaf(upper(n1.y), sx, cx);
    const float r0y = __builtin_fmaf(lower(n1.z), sy, cy), r1y = __builtin_fmaf(upper(n1.z), sy, cy);
    const float r0z = __builtin_fmaf(lower(n1.w), sz, cz), r1z = __builtin_fmaf(upper(n1.w), sz, cz);
    const float mx = fmaxf(fmaxf(fmaxf(fabsf(l0x), fabsf(l1x)), fabsf(r0x)), fabsf(r1x));
    const float my = fmaxf(fmaxf(fmaxf(fabsf(l0y), fabsf(l1y)), fabsf(r0y)), fabsf(r1y));
    const float mz = fmaxf(fmaxf(fmaxf(fabsf(l0z), fabsf(l1z)), fabsf(r0z)), fabsf(r1z));
    const float d2 = __builtin_fmaf(mz, mz, __builtin_fmaf(my, my, mx * mx));
    // the hardware square root is good to 1 ulp; the factor 1 + 2^-20 keeps the result an upper bound
    const float grown = __builtin_amdgcn_sqrtf(__builtin_fmaf(kappa, d2, floor_r * floor_r)) * 1.00000095367431640625f;
    // 2^-19 of the largest distance component for the slab arithmetic, 2^-22 of it and the frame's slack for
    // the planes themselves.  In a mesh's hierarchy `grown` is 2^-12 of (an upper bound of) the distance to
    // anything below the node: the ray's margin for running along a triangle's plane, w.gm of those, comes on
    // top (w.gm is 0 in the sphere hierarchy).
    const float e = __builtin_fmaf(grown, w.gm, __builtin_fmaf(2.1457672119140625e-6f, fmaxf(fmaxf(mx, my), mz), grown - floor_r) + sc.bvh_slack);
    // meshes with large triangles, rays that may run along one: the children's own margins.  No lane of a wave
    // is in this state on scenes of small triangles, and the wave then skips the block.
    float el = e, er = e;
    bool always_l = false, always_r = false;
    if (w.is(BvhWalk::kGrazeNodes)) {
        const uint4 gz = sc.bvh_graze[(w.ref & kBvhIndexMask) - sc.bvh_graze_first];
        // upper bounds of |d| and of the distance to anything below the node (hardware square root: 1 ulp)
        const float dlen = __builtin_amdgcn_sqrtf(w.a) * 1.000001f;
        const float dist = __builtin_amdgcn_sqrtf(d2) * 1.000001f;
        bvh_graze_child(gz.x, __uint_as_float(gz.z), d, w.a, dlen, dist, sc.bvh_graze_cap, e, el, always_l);
        bvh_graze_child(gz.y, __uint_as_float(gz.w), d, w.a, dlen, dist, sc.bvh_graze_cap, e, er, always_r);
    }
    // Entry and exit distances of the grown boxes.  NaN (0 * inf, inf - inf) makes a comparison false,
    // which keeps the box.  (The margin is applied to the plane, not to the quotient: folded into one fused
    // multiply-add per plane, plane / d -+ margin / d, a direction component of exactly 0 turns a ray origin
    // inside the margin into inf - inf, and the minimum of that NaN and the other plane's infinity culls the box.)
    const V3 inv = w.inv;
    const float a1 = (l0x - el) * inv.x, a2 = (l1x + el) * inv.x;
    const float a3 = (l0y - el) * inv.y, a4 = (l1y + el) * inv.y;
    const float a5 = (l0z - el) * inv.z, a6 = (l1z + el) * inv.z;
    const float b1 = (r0x - er) * inv.x, b2 = (r1x + er) * inv.x;
    const float b3 = (r0y - er) * inv.y, b4 = (r1y + er) * inv.y;
    const float b5 = (r0z - er) * inv.z, b6 = (r1z + er) * inv.z;
    const float tl = fmaxf(fmaxf(fminf(a1, a2), fminf(a3, a4)), fminf(a5, a6));
    const float tl_far = fminf(fminf(fmaxf(a1, a2), fmaxf(a3, a4)), fmaxf(a5, a6));
    const float tr = fmaxf(fmaxf(fminf(b1, b2), fminf(b3, b4)), fminf(b5, b6));
    const float tr_far = fminf(fminf(fmaxf(b1, b2), fmaxf(b3, b4)), fmaxf(b5, b6));
    const uint32_t L = n0.x, R = n0.y;
    const bool hl = ((L & kBvhUnbounded) != 0u) || always_l || (!(tl_far < 0.0f) && !(tl > tl_far) && !(tl > t_limit));
    const bool hr = ((R & kBvhUnbounded) != 0u) || always_r || (!(tr_far < 0.0f) && !(tr > tr_far) && !(tr > t_limit));
    // Where the walk goes next.  Written with selects on purpose: as a chain of if / else if / else (which
    // is how this read before) hipcc 7.2 has been seen to drop the copy that carries the popped or the right
    // child's reference into the register of w.ref in one instantiation, leaving lanes on a stale reference.
    const bool both = hl && hr, any = hl || hr;
    const bool right_first = tr < tl;
    if (both) {
        stack[(uint32_t)w.sp * stride] = right_first ? L : R;  // sp < kBvhStack: one entry per level
        w.sp++;
    }
    uint32_t next = (hr && (!hl || right_first)) ? R : L;
    if (!any) {
        next = w.ref;
        if (w.sp == 0) bvh_group_done(w, n_meshes);
        else next = stack[(uint32_t)(--w.sp) * stride];
    }
    w.ref = next;
}

// RaySphere (Raytracing.comp:130-161) for leaf entry e, whose record is g = {centre, r^2}
template <bool COUNT>
__device__ __forceinline__ void bvh_sphere(BvhWalk &w, const DeviceScene &sc, uint32_t n_spheres, V3 o, V3 d, float4 g, size_t e,
                                           Counters &cn)
{
    const V3 oc = o - mk(g.x, g.y, g.z);
    const float b = 2.0f * dot(oc, d);
    const float c = dot(oc, oc) - g.w;
    float disc;
    if constexpr (kFused) disc = __builtin_fmaf(b, b, -((4.0f * w.a) * c));
    else disc = b * b - (4.0f * w.a) * c;
    // with b >= 0 the root (-b - sqrt(disc)) / (2a) is not above zero (or not a number), which the
    // shader's t > 0.0001 rejects: no need to evaluate it
    if (disc >= 0.0f && b < 0.0f) {
        const float root = sqrt_rn_admits(disc) ? sqrt_rn(disc) : sqrtf(disc);
        const float t = (-b - root) / (2.0f * w.a);
        if (t > 0.0001f && t <= w.best.t) {
            // the shader's loop keeps the lowest index among equal t
            const uint32_t id = sc.bvh_sphere_id[e];
            if (id < n_spheres && (t < w.best.t || (id < w.best.prim && w.best.prim != kMissId))) {
                w.best.t = t;
                w.best.prim = id;
                w.best.mat = id;
            }
        }
    }
    if (COUNT) {
        // spheres beyond the count this frame names are not tests the shader makes
        if (sc.bvh_sphere_id[e] < n_spheres) {
            cn.c[0]++;
            if (disc >= 0.0f) cn.c[1]++;
        }
    }
}

// One step for a lane that is not at an inner node: the lane's next group (the spheres first, then the
// gated meshes in index order), or the next primitive(s) of the leaf it is in.
template <bool COUNT>
__device__ __forceinline__ void bvh_other(BvhWalk &w, const DeviceScene &sc, uint32_t n_spheres, uint32_t n_meshes, V3 o, V3 d,
                                          Counters &cn, uint32_t *stack, uint32_t stride)
{
    if (!w.is(BvhWalk::kWalking)) {
        if (w.gates != 0ull) {
            const uint32_t b = (uint32_t)__ffsll((long long)w.gates) - 1u;
            w.gates &= w.gates - 1ull;
            w.cur_mesh = w.gate_base + b;
            w.clear(BvhWalk::kInSpheres | BvhWalk::kOwnBest | BvhWalk::kGrazeNodes);
            const uint4 info = sc.bvh_mesh_info[w.cur_mesh];
            w.ref = info.x;
            bool node_by_node;
            w.gm = bvh_graze_enter(info.y, __uint_as_float(info.z), d, w.a, sc.bvh_graze_cap, node_by_node);
            if (node_by_node) w.set(BvhWalk::kGrazeNodes);
            if (w.ref == kBvhEmpty) {
                bvh_group_done(w, n_meshes);
                return;
            }
            w.set(BvhWalk::kWalking);
            w.leaf_left = 0u;
            w.sp = 0;
        } else if (n_meshes - w.gate_base > 64u) {
            w.gate_base += 64u;
            w.gates = bvh_gates<COUNT>(sc, w.gate_base, n_meshes, o, w.inv, cn);
            if (w.gates == 0ull && n_meshes - w.gate_base <= 64u) w.clear(BvhWalk::kActive);
        } else {
            w.clear(BvhWalk::kActive);
        }
        return;
    }

    if (w.leaf_left == 0u) {
        // bvh_at_inner() is false, so ref names a leaf
        w.leaf_pos = w.ref & kBvhIndexMask;
        w.leaf_left = (w.ref >> 28) & 7u;
    }
    const size_t e = w.leaf_pos;
    if (w.is(BvhWalk::kInSpheres)) {
        // up to two spheres of the leaf per step, fetched together
        const bool two = w.leaf_left >= 2u;
        const size_t e1 = two ? e + 1u : e;
        const float4 g0 = sc.bvh_sphere_geom[e];
        const float4 g1 = sc.bvh_sphere_geom[e1];
        bvh_sphere<COUNT>(w, sc, n_spheres, o, d, g0, e, cn);
        if (two) bvh_sphere<COUNT>(w, sc, n_spheres, o, d, g1, e1, cn);
        const uint32_t took = two ? 2u : 1u;
        w.leaf_pos += took;
        w.leaf_left -= took;
    } else {
        // RayTriangle, Raytracing.comp:286-309
        w.leaf_pos++;
        w.leaf_left--;
        const float4 *__restrict__ rec = sc.bvh_tri_isect + 4 * e;
        const float4 P = rec[0], AB = rec[1], AC = rec[2], N = rec[3];
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
            const float wgt = 1.0f - u - v;
            if (t >= 0.0f && u >= 0.0f && v >= 0.0f && wgt >= 0.0f && t <= w.best.t) {
                // a triangle of this mesh replaces an equal-t winner only if that winner is a
                // higher-indexed triangle of this same mesh: earlier groups keep their ties
                const uint32_t prim = n_spheres + sc.bvh_tri_id[e];
                if (t < w.best.t || (w.is(BvhWalk::kOwnBest) && prim < w.best.prim)) {
                    w.best.t = t;
                    w.best.prim = prim;
                    w.best.mat = sc.n_spheres + w.cur_mesh;
                    w.best.u = u;
                    w.best.v = v;
                    w.set(BvhWalk::kOwnBest);
                }
            }
        }
    }
    if (w.leaf_left == 0u) {
        if (w.sp == 0) bvh_group_done(w, n_meshes);
        else w.ref = stack[(uint32_t)(--w.sp) * stride];
    }
}

// kernels that resume walks (k_paths, k_stream): the walking lanes of a wave step together until at
// most NUM / 2^SHIFT of them are still walking
#ifndef RT_BVH_KEEP_NUM
#define RT_BVH_KEEP_NUM 1
#define RT_BVH_KEEP_SHIFT 1
#endif

// the lanes that are not at an inner node wait while they are fewer than NUM / DEN of those that are
#ifndef RT_BVH_WAIT_NUM
#define RT_BVH_WAIT_NUM 3
#define RT_BVH_WAIT_DEN 4
#endif

// One pass of the wave over its walks; every lane of the wave that is in the enclosing control flow
// must call it, stepping or not.  The three kinds of step (inner node, sphere of a leaf, triangle of a
// leaf or change of group) are different code, and a wave whose lanes want different kinds would run
// all of it for every pass, most lanes idle in each part.  So a pass runs ONE kind: inner nodes while
// enough lanes are at one, otherwise everything else for the lanes that have been waiting.  (Inner nodes of
// the meshes' hierarchies, which carry extra work for rays along triangle planes, were tried as part of
// "everything else": 146 -> 174 ms per 16 frames of the stress scene, the lanes at such nodes wait too long.)
template <bool COUNT>
__device__ __forceinline__ void bvh_pass(bool stepping, BvhWalk &w, const DeviceScene &sc, uint32_t n_spheres, uint32_t n_meshes, V3 o,
                                         V3 d, Counters &cn, uint32_t *stack, uint32_t stride)
{
    const bool inner = stepping && bvh_at_inner(w);
    const bool other = stepping && !inner;
    const uint32_t n_inner = (uint32_t)__popcll(__ballot(inner));
    const uint32_t n_other = (uint32_t)__popcll(__ballot(other));
    if (n_inner != 0u && n_other * RT_BVH_WAIT_DEN < n_inner * RT_BVH_WAIT_NUM) {
        if (inner) bvh_inner<COUNT>(w, sc, n_meshes, o, d, cn, stack, stride);
    } else {
        if (other) bvh_other<COUNT>(w, sc, n_spheres, n_meshes, o, d, cn, stack, stride);
    }
}

// The whole walk of one ray.
template <bool COUNT>
__device__ __forceinline__ Hit closest_hit_bvh(const DeviceScene &sc, uint32_t n_spheres, uint32_t n_meshes, V3 o, V3 d,
                                               Counters &cn, uint32_t *stack, uint32_t stride)
{
    BvhWalk w;
    bvh_begin<COUNT>(w, sc, n_spheres, n_meshes, o, d, cn);
    while (__ballot(w.active()) != 0ull) bvh_pass<COUNT>(w.active(), w, sc, n_spheres, n_meshes, o, d, cn, stack, stride);
    return w.best;
}

}  // namespace rtd
