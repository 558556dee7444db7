// rt_bvh.cpp — top-down hierarchy builder: binned surface-area splits with a hard depth limit
// (see rt_bvh.hpp).
#include "rt_bvh.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <numeric>

namespace rtmi {
namespace {

constexpr int kBins = 16;

struct Box {
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
    float hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    // NaN bounds never pass the comparisons, so a primitive with NaN coordinates (which no ray can
    // hit) leaves the box unchanged
    void Grow(const float *l, const float *h)
    {
        for (int a = 0; a < 3; a++) {
            if (l[a] < lo[a]) lo[a] = l[a];
            if (h[a] > hi[a]) hi[a] = h[a];
        }
    }
    void Grow(const Box &b) { Grow(b.lo, b.hi); }
    // half the surface area; 0 for an empty box
    double HalfArea() const
    {
        const double x = (double)hi[0] - lo[0], y = (double)hi[1] - lo[1], z = (double)hi[2] - lo[2];
        if (!(x >= 0.0) || !(y >= 0.0) || !(z >= 0.0)) return 0.0;
        const double s = x * y + y * z + z * x;
        return std::isfinite(s) ? s : DBL_MAX;
    }
};

inline int BinOf(float c, float lo, float scale)
{
    const int q = (int)((c - lo) * scale);
    return q < 0 ? 0 : (q >= kBins ? kBins - 1 : q);
}

// levels a balanced tree needs for n primitives
int BalancedDepth(uint32_t n)
{
    int d = 0;
    while (n > kLeafSize) {
        n = (n + 1) / 2;
        d++;
    }
    return d;
}

struct Builder {
    const std::vector<PrimBox> &boxes;
    std::vector<uint32_t> &order;
    std::vector<BvhNode> &nodes;
    uint32_t leaf_base;
    std::vector<float> cen;       // centroids, 3 per primitive (NaN replaced by 0)
    std::vector<uint32_t> tmp;    // partition scratch

    struct Result {
        uint32_t ref;
        Box box;
        float margin_floor;
    };

    Result Leaf(uint32_t b, uint32_t e)
    {
        Result r;
        r.margin_floor = FLT_MAX;
        for (uint32_t k = b; k < e; k++) {
            const PrimBox &p = boxes[order[k]];
            r.box.Grow(p.lo, p.hi);
            const float f = p.margin_floor >= 0.0f ? p.margin_floor : 0.0f;  // NaN -> 0
            if (f < r.margin_floor) r.margin_floor = f;
        }
        r.ref = kLeafFlag | ((e - b) << 28) | (leaf_base + b);
        return r;
    }

    // Stable split of order[b, e) by pred; returns the first index of the second part.
    template <class Pred>
    uint32_t Partition(uint32_t b, uint32_t e, Pred pred)
    {
        uint32_t w = b, t = 0;
        for (uint32_t k = b; k < e; k++) {
            const uint32_t id = order[k];
            if (pred(id)) order[w++] = id;
            else tmp[t++] = id;
        }
        std::copy(tmp.begin(), tmp.begin() + t, order.begin() + w);
        return w;
    }

    // Builds order[b, e) with at most `levels` further levels of inner nodes.
    Result Range(uint32_t b, uint32_t e, int levels)
    {
        const uint32_t n = e - b;
        if (n <= kLeafSize) return Leaf(b, e);

        uint32_t mid = b + n / 2;
        bool split_found = false;
        // a free split is allowed while a balanced finish of the larger side still fits afterwards
        if (levels > BalancedDepth(n)) {
            Box cb;
            for (uint32_t k = b; k < e; k++) {
                const float *c = &cen[3 * (size_t)order[k]];
                cb.Grow(c, c);
            }
            // the cheapest of the 3 x (kBins - 1) candidate planes: (surface area x count) summed over both sides
            double best = DBL_MAX;
            int best_axis = -1, best_q = -1;
            float best_lo = 0.0f, best_scale = 0.0f;
            for (int axis = 0; axis < 3; axis++) {
                const float ext = cb.hi[axis] - cb.lo[axis];
                if (!(std::isfinite(ext) && ext > 0.0f)) continue;
                const float scale = (float)kBins / ext, lo = cb.lo[axis];
                Box bin_box[kBins];
                uint32_t bin_n[kBins] = {};
                for (uint32_t k = b; k < e; k++) {
                    const uint32_t id = order[k];
                    const int q = BinOf(cen[3 * (size_t)id + axis], lo, scale);
                    bin_n[q]++;
                    bin_box[q].Grow(boxes[id].lo, boxes[id].hi);
                }
                double right_area[kBins];
                Box acc;
                for (int q = kBins - 1; q > 0; q--) {
                    acc.Grow(bin_box[q]);
                    right_area[q] = acc.HalfArea();
                }
                Box left;
                uint32_t left_n = 0;
                for (int q = 0; q + 1 < kBins; q++) {
                    left.Grow(bin_box[q]);
                    left_n += bin_n[q];
                    const uint32_t right_n = n - left_n;
                    if (left_n == 0 || right_n == 0) continue;
                    // the larger side must still fit into the remaining levels
                    if (BalancedDepth(std::max(left_n, right_n)) > levels - 1) continue;
                    const double cost = left.HalfArea() * left_n + right_area[q + 1] * right_n;
                    if (cost < best) {
                        best = cost;
                        best_axis = axis;
                        best_q = q;
                        best_lo = lo;
                        best_scale = scale;
                    }
                }
            }
            if (best_axis >= 0) {
                mid = Partition(b, e, [&](uint32_t id) { return BinOf(cen[3 * (size_t)id + best_axis], best_lo, best_scale) <= best_q; });
                split_found = mid > b && mid < e;
                if (!split_found) mid = b + n / 2;
            }
        }
        (void)split_found;  // without one the range is halved in its current order

        const uint32_t self = (uint32_t)nodes.size();
        nodes.emplace_back();
        const Result l = Range(b, mid, levels - 1);
        const Result r = Range(mid, e, levels - 1);
        BvhNode &nd = nodes[self];
        for (int a = 0; a < 3; a++) {
            nd.l_lo[a] = l.box.lo[a];
            nd.l_hi[a] = l.box.hi[a];
            nd.r_lo[a] = r.box.lo[a];
            nd.r_hi[a] = r.box.hi[a];
        }
        nd.left = l.ref;
        nd.right = r.ref;
        nd.l_margin_floor = l.margin_floor;
        nd.r_margin_floor = r.margin_floor;
        Result out;
        out.ref = self;
        out.box = l.box;
        out.box.Grow(r.box);
        out.margin_floor = std::min(l.margin_floor, r.margin_floor);
        return out;
    }
};

// spreads the low 10 bits of v so that two zero bits separate consecutive bits
uint32_t Spread10(uint32_t v)
{
    v &= 0x3FFu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

}  // namespace

uint32_t BuildBvh(const std::vector<PrimBox> &boxes, std::vector<BvhNode> &nodes, std::vector<uint32_t> &order,
                  uint32_t leaf_base)
{
    const uint32_t n = (uint32_t)boxes.size();
    order.resize(n);
    if (n == 0) return kEmptyRef;
    std::iota(order.begin(), order.end(), 0u);

    Builder bld{boxes, order, nodes, leaf_base, {}, {}};
    bld.cen.resize(3 * (size_t)n);
    bld.tmp.resize(n);
    Box cb;
    for (uint32_t i = 0; i < n; i++)
        for (int a = 0; a < 3; a++) {
            float m = 0.5f * boxes[i].lo[a] + 0.5f * boxes[i].hi[a];
            if (!std::isfinite(m)) m = 0.0f;
            bld.cen[3 * (size_t)i + a] = m;
        }
    for (uint32_t i = 0; i < n; i++) cb.Grow(&bld.cen[3 * (size_t)i], &bld.cen[3 * (size_t)i]);

    // Start from a Morton order (ties by index): ranges that end up halved without a surface-area
    // split (all centroids equal, or the depth limit reached) are then still spatially coherent.
    std::vector<uint32_t> code(n);
    float scale[3];
    for (int a = 0; a < 3; a++) {
        const float ext = cb.hi[a] - cb.lo[a];
        scale[a] = ext > 0.0f && std::isfinite(ext) ? 1023.0f / ext : 0.0f;
    }
    for (uint32_t i = 0; i < n; i++) {
        uint32_t q[3];
        for (int a = 0; a < 3; a++) {
            float f = (bld.cen[3 * (size_t)i + a] - cb.lo[a]) * scale[a];
            if (!(f >= 0.0f)) f = 0.0f;
            if (f > 1023.0f) f = 1023.0f;
            q[a] = (uint32_t)f;
        }
        code[i] = (Spread10(q[0]) << 2) | (Spread10(q[1]) << 1) | Spread10(q[2]);
    }
    std::sort(order.begin(), order.end(),
              [&](uint32_t x, uint32_t y) { return code[x] != code[y] ? code[x] < code[y] : x < y; });

    return bld.Range(0, n, kMaxDepth).ref;
}

}  // namespace rtmi
