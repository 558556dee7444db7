// rt_bvh.hpp — host-side bounding volume hierarchy builder for scenes too large for the
// brute-force loops of Raytracing.comp:343-368.  The reference has no acceleration structure; the
// hierarchy only decides WHICH primitives get the shader's exact tests, never how they are tested.
// Its boxes are the primitives' own bounds; the traversal grows them, per ray, by a bound on the
// rounding error of those tests, so the closest hit found is the one the brute-force loops find.
#pragma once

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

}  // namespace rtmi
