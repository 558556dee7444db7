// rt_bvh.hpp — host-side bounding volume hierarchy builder for scenes too large for the
// brute-force loops of Raytracing.comp:343-368.  The reference has no acceleration structure; the
// hierarchy only decides WHICH primitives get the shader's exact tests, never how they are tested.
// Its boxes are the primitives' own bounds; the traversal grows them, per ray, by a bound on the
// rounding error of those tests, so the closest hit found is the one the brute-force loops find.
#pragma once

#include <cmath>
#include <cstdint>
#include <vector>

namespace rtmi {

// One inner node holds the boxes of both children: a visit fetches 64 bytes and tests two boxes.
// margin_floor is the smallest "radius" of the primitives below the child (sphere radius; 0 for
// triangles): the traversal's error margin shrinks with it (see rt_bvh_device.hpp).
struct BvhNode {
    float l_lo[3]; uint32_t left;    // child reference
    float l_hi[3]; uint32_t right;   // child reference
    float r_lo[3]; float l_margin_floor;
    float r_hi[3]; float r_margin_floor;
};
static_assert(sizeof(BvhNode) == 64, "node is 64 bytes");

// Child reference: inner node index, or a leaf: kLeafFlag | count << 28 | first (range of the
// leaf-ordered primitive arrays).  kEmptyRef marks "no primitives".  Indices and leaf positions stay
// below 2^27: bit 27 is left to the quantised form of the nodes (kUnboundedFlag).
constexpr uint32_t kLeafFlag = 0x80000000u;
constexpr uint32_t kEmptyRef = 0xFFFFFFFFu;
constexpr uint32_t kUnboundedFlag = 0x08000000u;
constexpr uint32_t kRefIndexMask = 0x07FFFFFFu;
constexpr uint32_t kLeafSize = 2;  // measured on the 10^6 + 10^6 scene: 1 -> 239 ms, 2 -> 241, 3 -> 246, 4 -> 252, 6 -> 271
                                   // (with the later kernels: 1 -> 151, 2 -> 155, 3 -> 167, 4 -> 177; 1 doubles the node memory)
constexpr int kMaxDepth = 24;  // inner-node levels = entries of the traversal stack
constexpr uint64_t kMaxPrimitives = (uint64_t)kLeafSize << kMaxDepth;  // what a balanced tree of that depth holds

struct PrimBox {
    float lo[3], hi[3];
    float margin_floor = 0.0f;
};

// Builds a hierarchy over boxes[0..n) top-down: each range is split where 16-bin sweeps along the
// three axes of its centroids find the smallest (surface area x count) sum, as long as both sides
// can still be finished within kMaxDepth levels; otherwise (and when all centroids coincide) the
// range is halved.  Leaves hold at most kLeafSize primitives.  Appends inner nodes to `nodes`,
// returns the root reference, and fills order[k] = index of the primitive at leaf position k
// (order is local to this call: 0..n-1).  Leaf references count from leaf_base, so that several
// hierarchies can share one leaf-ordered primitive array (n at most kMaxPrimitives, leaf_base + n below 2^27).
// Large ranges are split over several threads; the nodes come out in pre-order with the same bytes as
// a single-threaded build would give.
uint32_t BuildBvh(const std::vector<PrimBox> &boxes, std::vector<BvhNode> &nodes, std::vector<uint32_t> &order,
                  uint32_t leaf_base);

// The form of the nodes the kernels walk: 32 bytes instead of 64, so that a visit is two 16-byte
// fetches per lane instead of four (the traversal is bound by the number of cache-line accesses its
// gathers make).  Every box plane is a 16-bit step count in one frame shared by all hierarchies of a
// scene, plane = origin + q * step per axis, lower planes rounded down and upper planes rounded up, so a
// quantised box encloses the box it stands for.  A child whose box has an infinite bound carries
// kUnboundedFlag in its reference and is always visited; a child without any finite primitive keeps an
// inverted box (lower 0xFFFF, upper 0) that no ray enters.
struct QNode {
    uint32_t left, right;  // child references (kUnboundedFlag possibly set)
    uint32_t l[3];         // left box per axis: lower count | upper count << 16
    uint32_t r[3];         // right box
};
static_assert(sizeof(QNode) == 32, "quantised node is 32 bytes");

struct QFrame {
    float origin[3] = {0.0f, 0.0f, 0.0f};
    float step[3] = {1.0f, 1.0f, 1.0f};
    float extent_max = 0.0f;   // largest 65535 * step over the axes
    float origin_max = 0.0f;   // largest |origin| over the axes
};

// Chooses the frame from the finite box planes of `nodes` and converts the nodes.
void QuantizeNodes(const std::vector<BvhNode> &nodes, QFrame &frame, std::vector<QNode> &out);

// ---- Rays that run almost inside a triangle's plane (triangle hierarchies only).
//
// RayTriangle (Raytracing.comp:286-309) decides by the signs of three triple products (the moments of the ray
// about the triangle's edges), each computed with an absolute error of up to c eps |edge| |ao| |d| (eps = 2^-24,
// c <= 5.5: sqrt(2) for the cross product ao x d, 4 sin(ao, d) for the dot product and the subtraction).  Seen
// along the ray, the triangle is a sliver of height |n| x / |edge| (n = AB x AC, x = |cos| of the angle between
// ray and normal) whose edge lines the errors push outwards, and at a tip of the sliver two such lines meet at
// an angle proportional to x: a ray can be accepted although it passes the triangle at a distance of up to
//     reach(x) = c eps D G / (|n| x),   G = max(2 |AB||AC|, (|AB|+|AC|+|BC|) max(|AB|,|AC|)),  D = |ao|,
// and the t it reports is off by less than the same amount.  det >= 1e-6 bounds x from below by
// 1e-6 / (|d| |n|) (less the error of det itself), so reach <= about 26 g D with g = eps 1e6 G |d|, a number
// that depends on the SIZE of the triangle: 0.0596 G |d|.
//
// The traversal grows boxes by 2^-12 D, which covers reach(x) for x >= graze_x = 4.91e-3 G / |n|.  For the
// flatter rays each child of a triangle-hierarchy node carries
//   * a cone around the normals below it, widened by asin(graze_x): a ray outside the band that the cone
//     leaves around 90 degrees cannot be that flat to any triangle below, and takes the usual margin;
//   * the largest g / |d| below it, from which the traversal takes the margin for rays inside the band (or
//     visits the child whatever the box says, when g is so large that the margin would be the whole scene).
struct TriGraze {
    double unit_normal[3] = {0.0, 0.0, 0.0};
    double graze_x = 0.0;    // std margin suffices for |cos(ray, normal)| >= graze_x
    double reach = 0.0;      // g / |d|
    bool contributes = false;  // false: the triangle can never be hit (AB x AC rounds to the zero vector)
    bool wild = false;         // coordinates that are not finite: the child is visited whatever the ray
};
// ab, ac, n: the edges and their cross product as the kernels use them (rounded binary32 values).
TriGraze TriangleGraze(const float ab[3], const float ac[3], const float n[3]);

// What the kernels read per inner node of a triangle hierarchy (16 bytes, same index as the node).
// cone word: bytes 0-2 the axis as three int8 (not normalised), byte 3 a code: 255 = every ray is in the band,
// otherwise a ray is in the band iff (d . axis)^2 <= T dot(d, d) with T = bits((code + 992) << 20) as binary32,
// i.e. 2^(code / 8 - 3) with a 3-bit mantissa.
struct GrazeNode {
    uint32_t l_cone, r_cone;
    float l_reach, r_reach;
};
static_assert(sizeof(GrazeNode) == 16, "graze record is 16 bytes");

// One record per node of `nodes` (a hierarchy built by BuildBvh with this leaf_base and order; tris[i] belongs
// to box i of that call).  root is what BuildBvh returned.
void BuildGrazeNodes(const std::vector<BvhNode> &nodes, uint32_t root, uint32_t leaf_base, const std::vector<uint32_t> &order,
                     const std::vector<TriGraze> &tris, std::vector<GrazeNode> &out);
// The same two numbers for a whole mesh: what a ray is checked against when it enters the mesh's hierarchy.
// A ray outside the mesh's band needs none of the above.  A ray inside it, with every triangle of the mesh
// small (g < 0.05), is given ONE margin for all of the mesh's nodes, from the mesh's reach, which costs a
// multiply-add per node visit; only a mesh with larger triangles makes the traversal look at the children's
// own cones and reaches, node by node (evaluating those for every ray cost 16 % of a launch on the stress
// scene, whose triangles are all small and whose white-noise heights leave no narrow cone anywhere).
struct MeshGraze {
    uint32_t cone = 0xFF000000u;
    float reach = INFINITY;
};
MeshGraze SummariseGraze(const std::vector<TriGraze> &tris);
// The threshold T of a cone word's code (for tests and for the builder's own rounding).
float GrazeConeThreshold(uint32_t code);

}  // namespace rtmi
