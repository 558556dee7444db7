// rt_pack.cpp — see rt_pack.hpp.
#include "rt_pack.hpp"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <exception>
#include <system_error>
#include <thread>

namespace rtmi {
namespace {

int Fail(std::string &err, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    err = buf;
    return code;
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

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

// AB, AC and AB x AC of a triangle with the single operations the shader repeats per ray
// (Raytracing.comp:288-290), in the arithmetic of the profile.
void triangle_edges(const rtmi_triangle &t, bool fused, float ab[3], float ac[3], float n[3])
{
    for (int a = 0; a < 3; a++) {
        ab[a] = t.p2[a] - t.p1[a];
        ac[a] = t.p3[a] - t.p1[a];
    }
    if (fused) {
        // the fused profile's cross product: one product rounded, the other fused into the difference
        n[0] = std::fmaf(ab[1], ac[2], -(ac[1] * ab[2]));
        n[1] = std::fmaf(ab[2], ac[0], -(ac[2] * ab[0]));
        n[2] = std::fmaf(ab[0], ac[1], -(ac[0] * ab[1]));
    } else {
        n[0] = ab[1] * ac[2] - ac[1] * ab[2];
        n[1] = ab[2] * ac[0] - ac[2] * ab[0];
        n[2] = ab[0] * ac[1] - ac[0] * ab[1];
    }
}


}  // namespace

// Everything precomputed here (r*r, the two edges, their cross product) is ray-independent and evaluated with
// the same single operations the shader would repeat per ray (Raytracing.comp:141, 288-290), so the bits the
// kernels consume are the bits the shader would have produced.
int PackScene(const rtmi_sphere *spheres, uint32_t nS, const rtmi_triangle *tris, uint32_t nT, const rtmi_mesh *meshes,
              uint32_t nM, bool want_bvh, bool fused, PackedScene &out, std::string &err, bool transmission)
{
    out = PackedScene();
    const size_t off_sph = 0;
    const size_t off_tri = align_up(off_sph + (size_t)nS * 16, 256);
    const size_t off_shd = align_up(off_tri + (size_t)nT * 64, 256);
    const size_t off_box = align_up(off_shd + (size_t)nT * 48, 256);
    const size_t off_mat = align_up(off_box + (size_t)nM * 32, 256);
    size_t total = align_up(off_mat + (size_t)(nS + nM) * 32, 256) + 256;

    // Hierarchies: one over all spheres, one per mesh over that mesh's triangle range.  The boxes are
    // the primitives' exact bounds (rounded outward); the margins that make skipping a box safe are
    // applied by the traversal, relative to each ray.
    std::vector<rtmi::BvhNode> nodes;
    std::vector<rtmi::QNode> qnodes;   // what is uploaded: the nodes in their 32-byte form
    std::vector<rtmi::MeshGraze> mesh_summary(nM);  // per mesh: cone and reach for rays along triangle planes
    std::vector<rtmi::GrazeNode> graze;  // the same per child of every node of the meshes' hierarchies
    uint32_t graze_first = 0u;           // index of the first such node
    rtmi::QFrame qframe;
    float sphere_floor = 0.0f;
    std::vector<uint32_t> sphere_order, tri_order;  // leaf position -> primitive index
    std::vector<uint32_t> mesh_root(nM, rtmi::kEmptyRef);
    uint32_t sphere_root = rtmi::kEmptyRef;
    size_t off_nodes = 0, off_bsph = 0, off_bsid = 0, off_btri = 0, off_btid = 0, off_mroot = 0, off_graze = 0;
    if (want_bvh) {
        uint64_t leaf_total = 0;
        for (uint32_t m = 0; m < nM; m++) leaf_total += (uint32_t)meshes[m].num_triangles;
        if (nS > rtmi::kMaxPrimitives || leaf_total > rtmi::kMaxPrimitives || nS >= rtmi::kUnboundedFlag || leaf_total >= rtmi::kUnboundedFlag)
            return Fail(err, RTMI_ERR_UNSUPPORTED, "rtmi_upload_scene: more than %llu primitives in a hierarchy",
                        (unsigned long long)rtmi::kMaxPrimitives);
        try {
            std::vector<rtmi::PrimBox> boxes(nS);
            for (uint32_t i = 0; i < nS; i++) {
                // the test uses the rounded square of the radius: 1e-6 relative covers that rounding
                const double r = std::fabs((double)spheres[i].radius) * (1.0 + 1e-6);
                for (int a = 0; a < 3; a++) {
                    boxes[i].lo[a] = round_down((double)spheres[i].center[a] - r);
                    boxes[i].hi[a] = round_up((double)spheres[i].center[a] + r);
                }
                boxes[i].margin_floor = (float)(std::fabs((double)spheres[i].radius) * 0.999999);
                // the smallest radius of all (not a number counts as 0): one margin floor for the whole hierarchy
                const float f = boxes[i].margin_floor >= 0.0f ? boxes[i].margin_floor : 0.0f;
                sphere_floor = (i == 0u || f < sphere_floor) ? f : sphere_floor;
            }
            // the sphere hierarchy on a thread of its own while the meshes are built below; its nodes
            // are the first in the node array either way
            std::exception_ptr sphere_failed;
            std::thread sphere_builder;
            try {
                sphere_builder = std::thread([&] {
                    try {
                        sphere_root = rtmi::BuildBvh(boxes, nodes, sphere_order, 0u);
                    } catch (...) {
                        sphere_failed = std::current_exception();
                    }
                });
            } catch (const std::system_error &) {
                sphere_root = rtmi::BuildBvh(boxes, nodes, sphere_order, 0u);  // no thread to be had: build it here
            }
            struct Joiner {
                std::thread &t;
                ~Joiner() { if (t.joinable()) t.join(); }
            } join_on_exit{sphere_builder};
            // The meshes' hierarchies do not depend on each other: a few threads build them side by
            // side, each mesh into node and order vectors of its own, which are then appended in mesh
            // order (node references moved), so the result does not depend on the thread count.
            std::vector<uint32_t> leaf_base(nM + 1u, 0u);
            for (uint32_t m = 0; m < nM; m++) leaf_base[m + 1] = leaf_base[m] + (uint32_t)meshes[m].num_triangles;
            std::vector<std::vector<rtmi::BvhNode>> mesh_nodes(nM);
            std::vector<std::vector<uint32_t>> mesh_order(nM);
            std::vector<std::vector<rtmi::GrazeNode>> mesh_graze(nM);
            auto build_meshes = [&](uint32_t first, uint32_t stride) {
                std::vector<rtmi::PrimBox> tb;
                std::vector<rtmi::TriGraze> tg;
                for (uint32_t m = first; m < nM; m += stride) {
                    const uint32_t start = (uint32_t)meshes[m].start_triangle, count = (uint32_t)meshes[m].num_triangles;
                    tb.resize(count);
                    for (uint32_t k = 0; k < count; k++) {
                        const rtmi_triangle &t = tris[start + k];
                        for (int a = 0; a < 3; a++) {
                            tb[k].lo[a] = std::min(t.p1[a], std::min(t.p2[a], t.p3[a]));
                            tb[k].hi[a] = std::max(t.p1[a], std::max(t.p2[a], t.p3[a]));
                        }
                        tb[k].margin_floor = 0.0f;
                    }
                    mesh_root[m] = rtmi::BuildBvh(tb, mesh_nodes[m], mesh_order[m], leaf_base[m]);
                    // cones and sizes for rays almost inside a triangle's plane, from the edges and the cross
                    // product exactly as the kernels will read them
                    tg.resize(count);
                    for (uint32_t k = 0; k < count; k++) {
                        float ab[3], ac[3], n[3];
                        triangle_edges(tris[start + k], fused, ab, ac, n);
                        tg[k] = rtmi::TriangleGraze(ab, ac, n);
                    }
                    rtmi::BuildGrazeNodes(mesh_nodes[m], mesh_root[m], leaf_base[m], mesh_order[m], tg, mesh_graze[m]);
                    mesh_summary[m] = rtmi::SummariseGraze(tg);
                    for (uint32_t k = 0; k < count; k++) mesh_order[m][k] += start;
                }
            };
            const uint32_t hw = std::max(1u, std::thread::hardware_concurrency());
            const uint32_t n_threads = (leaf_total < 32768u || nM < 2u) ? 1u : std::min({nM, hw, 8u});
            if (n_threads == 1u) {
                build_meshes(0u, 1u);
            } else {
                std::vector<std::thread> pool;
                std::vector<std::exception_ptr> failed(n_threads);
                std::vector<uint32_t> not_started;
                for (uint32_t t = 0; t < n_threads; t++) {
                    try {
                        pool.emplace_back([&, t] {
                            try {
                                build_meshes(t, n_threads);
                            } catch (...) {
                                failed[t] = std::current_exception();
                            }
                        });
                    } catch (const std::system_error &) {
                        not_started.push_back(t);
                    }
                }
                // shares whose thread could not be created are built here; the result is the same
                std::exception_ptr here;
                try {
                    for (uint32_t t : not_started) build_meshes(t, n_threads);
                } catch (...) {
                    here = std::current_exception();
                }
                for (std::thread &t : pool) t.join();
                if (here) std::rethrow_exception(here);
                for (const std::exception_ptr &f : failed)
                    if (f) std::rethrow_exception(f);
            }
            if (sphere_builder.joinable()) sphere_builder.join();
            if (sphere_failed) std::rethrow_exception(sphere_failed);
            tri_order.reserve(leaf_total);
            graze_first = (uint32_t)nodes.size();
            for (uint32_t m = 0; m < nM; m++) {
                const uint32_t shift = (uint32_t)nodes.size();
                graze.insert(graze.end(), mesh_graze[m].begin(), mesh_graze[m].end());
                std::vector<rtmi::GrazeNode>().swap(mesh_graze[m]);
                for (rtmi::BvhNode &nd : mesh_nodes[m]) {
                    if (!(nd.left & rtmi::kLeafFlag)) nd.left += shift;
                    if (!(nd.right & rtmi::kLeafFlag)) nd.right += shift;
                }
                if (!(mesh_root[m] & rtmi::kLeafFlag)) mesh_root[m] += shift;  // kEmptyRef has the flag set
                nodes.insert(nodes.end(), mesh_nodes[m].begin(), mesh_nodes[m].end());
                tri_order.insert(tri_order.end(), mesh_order[m].begin(), mesh_order[m].end());
                std::vector<rtmi::BvhNode>().swap(mesh_nodes[m]);
                std::vector<uint32_t>().swap(mesh_order[m]);
            }
            if (nodes.size() >= rtmi::kUnboundedFlag)
                return Fail(err, RTMI_ERR_UNSUPPORTED, "rtmi_upload_scene: %zu hierarchy nodes exceed the reference format", nodes.size());
            rtmi::QuantizeNodes(nodes, qframe, qnodes);
            std::vector<rtmi::BvhNode>().swap(nodes);
        } catch (const std::bad_alloc &) {
            return Fail(err, RTMI_ERR_OUT_OF_MEMORY, "rtmi_upload_scene: host allocation failed while building hierarchies");
        } catch (const std::exception &ex) {
            // e.g. std::system_error from thread creation
            return Fail(err, RTMI_ERR_UNSUPPORTED, "rtmi_upload_scene: building hierarchies failed: %s", ex.what());
        } catch (...) {
            return Fail(err, RTMI_ERR_UNSUPPORTED, "rtmi_upload_scene: building hierarchies failed");
        }
        off_nodes = align_up(total, 256);
        off_bsph = align_up(off_nodes + qnodes.size() * sizeof(rtmi::QNode), 256);
        off_bsid = align_up(off_bsph + (size_t)nS * 16, 256);
        off_btri = align_up(off_bsid + (size_t)nS * 4, 256);
        off_btid = align_up(off_btri + tri_order.size() * 64, 256);
        off_mroot = align_up(off_btid + tri_order.size() * 4, 256);
        off_graze = align_up(off_mroot + (size_t)nM * 16, 256);
        total = align_up(off_graze + graze.size() * sizeof(rtmi::GrazeNode), 256) + 256;
    }
    std::vector<unsigned char> &blob = out.blob;
    try {
        blob.assign(total, 0);
    } catch (const std::bad_alloc &) {
        return Fail(err, RTMI_ERR_OUT_OF_MEMORY, "rtmi_upload_scene: host allocation of %zu bytes failed", total);
    }
    float *sph = reinterpret_cast<float *>(blob.data() + off_sph);
    float *tri = reinterpret_cast<float *>(blob.data() + off_tri);
    float *shd = reinterpret_cast<float *>(blob.data() + off_shd);
    float *box = reinterpret_cast<float *>(blob.data() + off_box);
    float *mat = reinterpret_cast<float *>(blob.data() + off_mat);

    auto put_material = [&](uint32_t row, const rtmi_material &m) {
        float *d = mat + 8 * (size_t)row;
        d[0] = m.color[0]; d[1] = m.color[1]; d[2] = m.color[2]; d[3] = m.light;
        d[4] = m.smoothness;
        // the transmitting-material extension, when the caller has switched it on: flag and index of
        // refraction travel in the record's padding, which the shader never reads (and a host may leave unset)
        if (transmission && m.pad1 > 0.5f) {
            d[5] = m.pad1;
            d[6] = m.pad2;
            out.has_glass = true;
        }
    };
    for (uint32_t i = 0; i < nS; i++) {
        const rtmi_sphere &s = spheres[i];
        float *d = sph + 4 * (size_t)i;
        d[0] = s.center[0]; d[1] = s.center[1]; d[2] = s.center[2];
        d[3] = s.radius * s.radius;
        put_material(i, s.material);
    }
    for (uint32_t j = 0; j < nT; j++) {
        const rtmi_triangle &t = tris[j];
        float ab[3], ac[3], n[3];
        triangle_edges(t, fused, ab, ac, n);
        float *d = tri + 16 * (size_t)j;
        for (int a = 0; a < 3; a++) {
            d[0 + a] = t.p1[a];
            d[4 + a] = ab[a];
            d[8 + a] = ac[a];
            d[12 + a] = n[a];
        }
        float *s = shd + 12 * (size_t)j;
        for (int a = 0; a < 3; a++) {
            s[0 + a] = t.n1[a];
            s[4 + a] = t.n2[a];
            s[8 + a] = t.n3[a];
        }
    }
    for (uint32_t m = 0; m < nM; m++) {
        const rtmi_mesh &ms = meshes[m];
        float *d = box + 8 * (size_t)m;
        const uint32_t start = (uint32_t)ms.start_triangle, count = (uint32_t)ms.num_triangles;
        for (int a = 0; a < 3; a++) {
            d[0 + a] = ms.bounding_point1[a];
            d[4 + a] = ms.bounding_point2[a];
        }
        memcpy(&d[3], &start, 4);
        memcpy(&d[7], &count, 4);
        put_material(nS + m, ms.material);
    }

    if (want_bvh) {
        if (!qnodes.empty()) memcpy(blob.data() + off_nodes, qnodes.data(), qnodes.size() * sizeof(rtmi::QNode));
        float *bs = reinterpret_cast<float *>(blob.data() + off_bsph);
        uint32_t *bi = reinterpret_cast<uint32_t *>(blob.data() + off_bsid);
        for (uint32_t k = 0; k < nS; k++) {
            memcpy(bs + 4 * (size_t)k, sph + 4 * (size_t)sphere_order[k], 16);
            bi[k] = sphere_order[k];
        }
        float *bt = reinterpret_cast<float *>(blob.data() + off_btri);
        uint32_t *bj = reinterpret_cast<uint32_t *>(blob.data() + off_btid);
        for (size_t k = 0; k < tri_order.size(); k++) {
            memcpy(bt + 16 * k, tri + 16 * (size_t)tri_order[k], 64);
            bj[k] = tri_order[k];
        }
        // per mesh {root reference, cone word, reach, 0}
        uint32_t *mi = reinterpret_cast<uint32_t *>(blob.data() + off_mroot);
        for (uint32_t m = 0; m < nM; m++) {
            mi[4 * (size_t)m + 0] = mesh_root[m];
            mi[4 * (size_t)m + 1] = mesh_summary[m].cone;
            memcpy(&mi[4 * (size_t)m + 2], &mesh_summary[m].reach, 4);
            mi[4 * (size_t)m + 3] = 0u;
        }
        if (!graze.empty()) memcpy(blob.data() + off_graze, graze.data(), graze.size() * sizeof(rtmi::GrazeNode));
    }


    out.off_sph = off_sph;
    out.off_tri = off_tri;
    out.off_shd = off_shd;
    out.off_box = off_box;
    out.off_mat = off_mat;
    out.off_nodes = off_nodes;
    out.off_bsph = off_bsph;
    out.off_bsid = off_bsid;
    out.off_btri = off_btri;
    out.off_btid = off_btid;
    out.off_mroot = off_mroot;
    out.off_graze = off_graze;
    out.bvh = want_bvh;
    out.qframe = qframe;
    // 2^-24 of the frame's extent: the part of a dequantised plane's rounding that does not scale with its
    // distance from the ray origin (rt_bvh_device.hpp)
    out.slack = round_up((double)qframe.extent_max * 6.1e-8);
    out.sphere_floor = sphere_floor;
    out.sphere_root = sphere_root;
    out.node_count = (uint32_t)qnodes.size();
    out.graze_first = graze_first;
    out.n_spheres = nS;
    out.n_triangles = nT;
    out.n_meshes = nM;
    return 0;
}

}  // namespace rtmi
