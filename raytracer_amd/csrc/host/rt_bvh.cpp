// rt_bvh.cpp — top-down hierarchy builder: binned surface-area splits with a hard depth limit
// (see rt_bvh.hpp).
#include "rt_bvh.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <exception>
#include <numeric>
#include <system_error>
#include <thread>

namespace rtmi {
namespace {

constexpr int kBins = 16;
// The two halves of a range are built side by side on two threads for the first kForkLevels levels,
// as long as a half holds at least kForkMin primitives.
constexpr int kForkLevels = 4;
constexpr uint32_t kForkMin = 16384;

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

// A primitive as the builder moves it around: the passes over a range then read memory in order.
struct Item {
    float lo[3], hi[3];
    float cen[3];        // centroid (NaN replaced by 0)
    float margin_floor;  // NaN and negative values replaced by 0
    uint32_t id;
};

struct Builder {
    std::vector<Item> items;  // in leaf order when the build is done
    std::vector<Item> tmp;    // partition scratch: a range [b, e) of items uses tmp[b, e)
    uint32_t leaf_base;

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
            const Item &p = items[k];
            r.box.Grow(p.lo, p.hi);
            if (p.margin_floor < r.margin_floor) r.margin_floor = p.margin_floor;
        }
        r.ref = kLeafFlag | ((e - b) << 28) | (leaf_base + b);   // leaf_base + b < 2^27 (callers check)
        return r;
    }

    // Stable split of items[b, e) by pred; returns the first index of the second part.
    template <class Pred>
    uint32_t Partition(uint32_t b, uint32_t e, Pred pred)
    {
        uint32_t w = b, t = b;
        for (uint32_t k = b; k < e; k++) {
            if (pred(items[k])) items[w++] = items[k];
            else tmp[t++] = items[k];
        }
        std::copy(tmp.begin() + b, tmp.begin() + t, items.begin() + w);
        return w;
    }

    // Builds items[b, e) with at most `levels` further levels of inner nodes, appending its nodes to
    // `nodes` in pre-order; references are indices into `nodes`.  `fork` levels of the recursion may
    // still split their work over two threads; the result does not depend on it.
    Result Range(uint32_t b, uint32_t e, int levels, std::vector<BvhNode> &nodes, int fork)
    {
        const uint32_t n = e - b;
        if (n <= kLeafSize) return Leaf(b, e);

        uint32_t mid = b + n / 2;
        // a free split is allowed while a balanced finish of the larger side still fits afterwards
        if (levels > BalancedDepth(n)) {
            Box cb;
            for (uint32_t k = b; k < e; k++) cb.Grow(items[k].cen, items[k].cen);
            // the cheapest of the 3 x (kBins - 1) candidate planes: (surface area x count) summed over both sides
            double best = DBL_MAX;
            int best_axis = -1, best_q = -1;
            float best_lo = 0.0f, best_scale = 0.0f;
            Box bin_box[3][kBins];
            uint32_t bin_n[3][kBins] = {};
            float lo[3], scale[3];
            bool usable[3];
            for (int axis = 0; axis < 3; axis++) {
                const float ext = cb.hi[axis] - cb.lo[axis];
                usable[axis] = std::isfinite(ext) && ext > 0.0f;
                scale[axis] = usable[axis] ? (float)kBins / ext : 0.0f;
                lo[axis] = cb.lo[axis];
            }
            for (uint32_t k = b; k < e; k++) {
                const Item &p = items[k];
                for (int axis = 0; axis < 3; axis++) {
                    if (!usable[axis]) continue;
                    const int q = BinOf(p.cen[axis], lo[axis], scale[axis]);
                    bin_n[axis][q]++;
                    bin_box[axis][q].Grow(p.lo, p.hi);
                }
            }
            for (int axis = 0; axis < 3; axis++) {
                if (!usable[axis]) continue;
                double right_area[kBins];
                Box acc;
                for (int q = kBins - 1; q > 0; q--) {
                    acc.Grow(bin_box[axis][q]);
                    right_area[q] = acc.HalfArea();
                }
                Box left;
                uint32_t left_n = 0;
                for (int q = 0; q + 1 < kBins; q++) {
                    left.Grow(bin_box[axis][q]);
                    left_n += bin_n[axis][q];
                    const uint32_t right_n = n - left_n;
                    if (left_n == 0 || right_n == 0) continue;
                    // the larger side must still fit into the remaining levels
                    if (BalancedDepth(std::max(left_n, right_n)) > levels - 1) continue;
                    const double cost = left.HalfArea() * left_n + right_area[q + 1] * right_n;
                    if (cost < best) {
                        best = cost;
                        best_axis = axis;
                        best_q = q;
                        best_lo = lo[axis];
                        best_scale = scale[axis];
                    }
                }
            }
            if (best_axis >= 0) {
                mid = Partition(b, e, [&](const Item &p) { return BinOf(p.cen[best_axis], best_lo, best_scale) <= best_q; });
                if (mid == b || mid == e) mid = b + n / 2;  // cannot happen for a split with both counts > 0
            }
        }
        // without a split the range is halved in its current order

        const uint32_t self = (uint32_t)nodes.size();
        nodes.emplace_back();
        Result l, r;
        if (fork > 0 && mid - b >= kForkMin && e - mid >= kForkMin) {
            // the right half on a thread of its own, into a vector of its own that is appended (with
            // its references moved) behind the left half's nodes
            std::vector<BvhNode> right_nodes;
            right_nodes.reserve(e - mid);  // more than it can need; reserved here so that the worker never allocates
            std::exception_ptr failure;
            std::thread worker;
            bool forked = true;
            try {
                worker = std::thread([&] {
                    try {
                        r = Range(mid, e, levels - 1, right_nodes, fork - 1);
                    } catch (...) {
                        failure = std::current_exception();
                    }
                });
            } catch (const std::system_error &) {
                forked = false;  // no thread to be had: this one builds both halves, into the same two vectors
            }
            try {
                l = Range(b, mid, levels - 1, nodes, fork - 1);
                if (!forked) r = Range(mid, e, levels - 1, right_nodes, 0);
            } catch (...) {
                if (forked) worker.join();
                throw;
            }
            if (forked) worker.join();
            if (failure) std::rethrow_exception(failure);
            const uint32_t shift = (uint32_t)nodes.size();
            for (BvhNode &nd : right_nodes) {
                if (!(nd.left & kLeafFlag)) nd.left += shift;
                if (!(nd.right & kLeafFlag)) nd.right += shift;
            }
            if (!(r.ref & kLeafFlag)) r.ref += shift;
            nodes.insert(nodes.end(), right_nodes.begin(), right_nodes.end());
        } else {
            l = Range(b, mid, levels - 1, nodes, 0);
            r = Range(mid, e, levels - 1, nodes, 0);
        }
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

    Builder bld;
    bld.leaf_base = leaf_base;
    std::vector<Item> unsorted(n);
    Box cb;
    for (uint32_t i = 0; i < n; i++) {
        Item &it = unsorted[i];
        for (int a = 0; a < 3; a++) {
            it.lo[a] = boxes[i].lo[a];
            it.hi[a] = boxes[i].hi[a];
            float m = 0.5f * boxes[i].lo[a] + 0.5f * boxes[i].hi[a];
            if (!std::isfinite(m)) m = 0.0f;
            it.cen[a] = m;
        }
        it.margin_floor = boxes[i].margin_floor >= 0.0f ? boxes[i].margin_floor : 0.0f;  // NaN -> 0
        it.id = i;
        cb.Grow(it.cen, it.cen);
    }

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
            float f = (unsorted[i].cen[a] - cb.lo[a]) * scale[a];
            if (!(f >= 0.0f)) f = 0.0f;
            if (f > 1023.0f) f = 1023.0f;
            q[a] = (uint32_t)f;
        }
        code[i] = (Spread10(q[0]) << 2) | (Spread10(q[1]) << 1) | Spread10(q[2]);
    }
    std::iota(order.begin(), order.end(), 0u);
    std::sort(order.begin(), order.end(),
              [&](uint32_t x, uint32_t y) { return code[x] != code[y] ? code[x] < code[y] : x < y; });
    bld.items.resize(n);
    for (uint32_t k = 0; k < n; k++) bld.items[k] = unsorted[order[k]];
    std::vector<Item>().swap(unsorted);
    bld.tmp.resize(n);

    nodes.reserve(nodes.size() + n);
    const uint32_t root = bld.Range(0, n, kMaxDepth, nodes, kForkLevels).ref;
    for (uint32_t k = 0; k < n; k++) order[k] = bld.items[k].id;
    return root;
}

void QuantizeNodes(const std::vector<BvhNode> &nodes, QFrame &frame, std::vector<QNode> &out)
{
    // the frame: the finite planes' range per axis
    double lo[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, hi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
    auto take = [&](const float *l, const float *h) {
        // an inverted box (a child without any primitive that has coordinates) is not part of the scene
        if (!(l[0] <= h[0]) || !(l[1] <= h[1]) || !(l[2] <= h[2])) return;
        for (int a = 0; a < 3; a++) {
            if (std::isfinite(l[a]) && l[a] < lo[a]) lo[a] = l[a];
            if (std::isfinite(h[a]) && h[a] > hi[a]) hi[a] = h[a];
            // a box that is infinite on one side still has a finite plane on the other
            if (std::isfinite(h[a]) && h[a] < lo[a]) lo[a] = h[a];
            if (std::isfinite(l[a]) && l[a] > hi[a]) hi[a] = l[a];
        }
    };
    for (const BvhNode &nd : nodes) {
        take(nd.l_lo, nd.l_hi);
        take(nd.r_lo, nd.r_hi);
    }
    frame = QFrame();
    for (int a = 0; a < 3; a++) {
        if (!(lo[a] <= hi[a])) {
            lo[a] = 0.0;
            hi[a] = 0.0;
        }
        frame.origin[a] = (float)lo[a];
        if ((double)frame.origin[a] > lo[a]) frame.origin[a] = std::nextafterf(frame.origin[a], -INFINITY);
        const double ext = hi[a] - (double)frame.origin[a];
        float st = (float)(ext / 65534.0);   // one step of slack at the top for the rounding below
        if ((double)st * 65534.0 < ext) st = std::nextafterf(st, INFINITY);
        if (!(st > 0.0f) || !std::isfinite(st)) st = 1.0f;
        frame.step[a] = st;
        frame.extent_max = std::max(frame.extent_max, 65535.0f * st);
        frame.origin_max = std::max(frame.origin_max, std::fabs(frame.origin[a]));
    }
    // one child box -> three packed axes; returns false if a bound cannot be represented (infinite)
    auto pack = [&](const float *l, const float *h, uint32_t *q) {
        bool bounded = true;
        const bool empty = !(l[0] <= h[0]) || !(l[1] <= h[1]) || !(l[2] <= h[2]);   // inverted (or NaN, which Grow never stores)
        for (int a = 0; a < 3; a++) {
            if (empty) {
                q[a] = 0x0000FFFFu;   // lower 0xFFFF, upper 0: no ray enters
                continue;
            }
            const double o = frame.origin[a], st = frame.step[a];
            if (!std::isfinite(l[a]) || !std::isfinite(h[a])) {
                // an infinite bound has no step count (and must not reach the integer conversion below)
                bounded = false;
                q[a] = 0xFFFF0000u;   // the whole frame; the child is flagged and visited whatever its box says
                continue;
            }
            // one step of slack each way: the divisions are rounded.  The quotients are clamped before the
            // conversion: a plane far outside the frame must not overflow the integer
            const double fl = std::min(std::max(std::floor(((double)l[a] - o) / st) - 1.0, -2.0), 65537.0);
            const double fh = std::min(std::max(std::ceil(((double)h[a] - o) / st) + 1.0, -2.0), 65537.0);
            long long ql = (long long)fl, qh = (long long)fh;
            if (ql < -1 || qh > 65536) bounded = false;
            ql = std::min(std::max(ql, 0ll), 65535ll);
            qh = std::min(std::max(qh, 0ll), 65535ll);
            // what was clamped must still enclose
            if (o + (double)ql * st > (double)l[a] || o + (double)qh * st < (double)h[a]) bounded = false;
            q[a] = (uint32_t)ql | ((uint32_t)qh << 16);
        }
        return bounded || empty;
    };
    out.resize(nodes.size());
    for (size_t k = 0; k < nodes.size(); k++) {
        const BvhNode &nd = nodes[k];
        QNode &q = out[k];
        q.left = nd.left;
        q.right = nd.right;
        if (!pack(nd.l_lo, nd.l_hi, q.l)) q.left |= kUnboundedFlag;
        if (!pack(nd.r_lo, nd.r_hi, q.r)) q.right |= kUnboundedFlag;
    }
}

// ---------------------------------------------------------------------------------------------------------
// rays almost inside a triangle's plane (see rt_bvh.hpp)

namespace {

constexpr double kPi2 = 1.57079632679489661923;
// 2 (reach and the error of t) x 5.5 (rounding constant) x 1.82 (D + 0.41 longest edge <= 1.82 x the node's
// distance bound) x 2^-24, over the usual margin of 2^-12 of the distance
constexpr double kGrazeX = 2.0 * 5.5 * 1.82 * 5.9604644775390625e-8 * 4096.0;
constexpr double kReachPerG = 5.9604644775390625e-8 * 1e6;   // eps * 1e6

struct ConeAcc {
    double sum[3] = {0.0, 0.0, 0.0};
    double graze_x = 0.0, reach = 0.0;
    uint32_t n = 0;
    bool wild = false;
};

// Cone and reach of a set of triangles: tri(k), k in [b, e).
template <class TriAt>
void GrazeOfSet(uint32_t b, uint32_t e, TriAt tri, uint32_t &cone, float &reach)
{
    ConeAcc acc;
    for (uint32_t k = b; k < e; k++) {
        const TriGraze &t = tri(k);
        if (t.wild) acc.wild = true;
        if (!t.contributes) continue;
        for (int a = 0; a < 3; a++) acc.sum[a] += t.unit_normal[a];
        acc.graze_x = std::max(acc.graze_x, t.graze_x);
        acc.reach = std::max(acc.reach, t.reach);
        acc.n++;
    }
    if (acc.wild) {
        cone = 0xFF000000u;
        reach = INFINITY;
        return;
    }
    if (acc.n == 0) {
        // nothing below can be hit: no ray is ever in the band (T = smallest code, axis 0 would make every
        // ray pass, so the axis is a unit vector and reach 0 keeps the usual margin even then)
        cone = 0x0000007Fu;
        reach = 0.0f;
        return;
    }
    float r = (float)acc.reach;
    if ((double)r < acc.reach) r = std::nextafterf(r, INFINITY);
    reach = r;
    const double len = std::sqrt(acc.sum[0] * acc.sum[0] + acc.sum[1] * acc.sum[1] + acc.sum[2] * acc.sum[2]);
    if (!(len > 1e-9 * acc.n) || !(acc.graze_x < 1.0)) {
        cone = 0xFF000000u;
        return;
    }
    int q[3];
    double ql = 0.0;
    for (int a = 0; a < 3; a++) {
        q[a] = (int)std::lrint(127.0 * acc.sum[a] / len);
        ql += (double)q[a] * q[a];
    }
    if (ql == 0.0) {
        cone = 0xFF000000u;
        return;
    }
    // the half-angle of the cone about the axis AS STORED
    double cos_min = 1.0;
    for (uint32_t k = b; k < e; k++) {
        const TriGraze &t = tri(k);
        if (!t.contributes) continue;
        const double c = (t.unit_normal[0] * q[0] + t.unit_normal[1] * q[1] + t.unit_normal[2] * q[2]) / std::sqrt(ql);
        cos_min = std::min(cos_min, c);
    }
    const double theta = std::acos(std::min(1.0, std::max(-1.0, cos_min)));
    const double band = theta + std::asin(acc.graze_x) + 1e-6;
    if (!(band < kPi2)) {
        cone = 0xFF000000u;
        return;
    }
    const double s = std::sin(band);
    // the kernel compares (d . q)^2 with T dot(d, d), both in binary32: 2^-18 relative covers their rounding
    const double need = s * s * ql * (1.0 + 1e-4) + 4e-6 * ql;
    uint32_t code = 0;
    while (code < 255u && (double)GrazeConeThreshold(code) < need) code++;
    cone = (code << 24) | ((uint32_t)(q[2] & 0xFF) << 16) | ((uint32_t)(q[1] & 0xFF) << 8) | (uint32_t)(q[0] & 0xFF);
}

struct GrazeBuilder {
    const std::vector<BvhNode> &nodes;
    const std::vector<uint32_t> &order;
    const std::vector<TriGraze> &tris;
    uint32_t leaf_base;
    std::vector<GrazeNode> &out;

    // One child: the triangles at leaf positions [b, e).
    void Child(uint32_t b, uint32_t e, uint32_t &cone, float &reach) const
    {
        GrazeOfSet(b, e, [this](uint32_t k) -> const TriGraze & { return tris[order[k]]; }, cone, reach);
    }

    // Returns the range of leaf positions below ref.
    void Visit(uint32_t ref, uint32_t &b, uint32_t &e)
    {
        if (ref & kLeafFlag) {
            b = (ref & kRefIndexMask) - leaf_base;
            e = b + ((ref >> 28) & 7u);
            return;
        }
        const BvhNode &nd = nodes[ref];
        uint32_t bl, el, br, er;
        Visit(nd.left, bl, el);
        Visit(nd.right, br, er);
        GrazeNode &g = out[ref];
        Child(bl, el, g.l_cone, g.l_reach);
        Child(br, er, g.r_cone, g.r_reach);
        b = std::min(bl, br);
        e = std::max(el, er);
    }
};

}  // namespace

float GrazeConeThreshold(uint32_t code)
{
    const uint32_t bits = (code + 992u) << 20;
    float f;
    static_assert(sizeof f == sizeof bits, "binary32");
    std::memcpy(&f, &bits, sizeof f);
    return f;
}

TriGraze TriangleGraze(const float ab[3], const float ac[3], const float n[3])
{
    TriGraze t;
    for (int a = 0; a < 3; a++)
        if (!std::isfinite(ab[a]) || !std::isfinite(ac[a]) || !std::isfinite(n[a])) t.wild = true;
    if (t.wild) return t;
    const double nn = std::sqrt((double)n[0] * n[0] + (double)n[1] * n[1] + (double)n[2] * n[2]);
    if (!(nn > 0.0)) return t;   // det = -dot(d, n) is 0 for every ray
    const double la = std::sqrt((double)ab[0] * ab[0] + (double)ab[1] * ab[1] + (double)ab[2] * ab[2]);
    const double lb = std::sqrt((double)ac[0] * ac[0] + (double)ac[1] * ac[1] + (double)ac[2] * ac[2]);
    const double bc[3] = {(double)ac[0] - ab[0], (double)ac[1] - ab[1], (double)ac[2] - ab[2]};
    const double lc = std::sqrt(bc[0] * bc[0] + bc[1] * bc[1] + bc[2] * bc[2]);
    const double G = std::max(2.0 * la * lb, (la + lb + lc) * std::max(la, lb));
    for (int a = 0; a < 3; a++) t.unit_normal[a] = n[a] / nn;
    // the rounded cross product may point up to 1.5 eps |AB||AC| / |n| away from the exact one
    t.graze_x = std::min(1.0, kGrazeX * G / nn + 2e-7 * la * lb / nn);
    t.reach = kReachPerG * G;
    t.contributes = true;
    return t;
}

MeshGraze SummariseGraze(const std::vector<TriGraze> &tris)
{
    MeshGraze out;
    GrazeOfSet(0u, (uint32_t)tris.size(), [&tris](uint32_t k) -> const TriGraze & { return tris[k]; }, out.cone, out.reach);
    return out;
}

void BuildGrazeNodes(const std::vector<BvhNode> &nodes, uint32_t root, uint32_t leaf_base, const std::vector<uint32_t> &order,
                     const std::vector<TriGraze> &tris, std::vector<GrazeNode> &out)
{
    out.assign(nodes.size(), GrazeNode{0xFF000000u, 0xFF000000u, INFINITY, INFINITY});
    if (root == kEmptyRef || (root & kLeafFlag)) return;
    GrazeBuilder gb{nodes, order, tris, leaf_base, out};
    uint32_t b, e;
    gb.Visit(root, b, e);
}

}  // namespace rtmi
