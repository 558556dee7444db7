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
// leaf-ordered primitive arrays).  kEmptyRef marks "no primitives".
constexpr uint32_t kLeafFlag = 0x80000000u;
constexpr uint32_t kEmptyRef = 0xFFFFFFFFu;
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
// hierarchies can share one leaf-ordered primitive array (n at most kMaxPrimitives, leaf_base + n below 2^28).
// Large ranges are split over several threads; the nodes come out in pre-order with the same bytes as
// a single-threaded build would give.
uint32_t BuildBvh(const std::vector<PrimBox> &boxes, std::vector<BvhNode> &nodes, std::vector<uint32_t> &order,
                  uint32_t leaf_base);

}  // namespace rtmi
