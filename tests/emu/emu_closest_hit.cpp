// emu_closest_hit.cpp — TEST infrastructure: the library's own traversal code (rt_bvh_device.hpp) and its own
// scene packing (host/rt_pack.cpp), compiled for the host and run one ray at a time, so that the hierarchy path
// can be checked on a machine without a GPU, under AddressSanitizer, and stepped through with gdb.  Every array
// access of the traversal is checked against the packed block's extent.
#include <hip/hip_runtime.h>   // tests/emu/hip/hip_runtime.h

#include "rt_bvh_device.hpp"
#include "rt_pack.hpp"

#include <string>
#include <vector>

extern "C" {

// ClosestHit (Raytracing.comp:338-388) of n rays (3 floats each) against the scene, through the hierarchy
// (use_bvh = 1) or the exhaustive loops (0), exactly as rtmi_selftest_closest_hit does on the device.
// Returns 0, or a negative rtmi_status from packing; msg (256 bytes) receives the reason.
__attribute__((visibility("default"))) int emu_closest_hit(const rtmi_sphere *spheres, uint32_t nS, const rtmi_triangle *tris,
                                                           uint32_t nT, const rtmi_mesh *meshes, uint32_t nM, int use_bvh,
                                                           float graze_cap, const float *origins, const float *dirs, size_t n,
                                                           float *t_out, uint32_t *prim_out, char *msg)
{
    rtmi::PackedScene pk;
    std::string why;
    const int rc = rtmi::PackScene(spheres, nS, tris, nT, meshes, nM, use_bvh != 0, rtd::kFused, pk, why);
    if (rc != 0) {
        if (msg) snprintf(msg, 256, "%s", why.c_str());
        return rc;
    }
    // an exact-size copy: AddressSanitizer then sees every read beyond the block
    std::vector<unsigned char> exact(pk.blob.begin(), pk.blob.end());
    rtd::DeviceScene sc{};
    RTMI_BIND_PACKED_SCENE(pk, exact.data(), sc);
    sc.bvh_graze_cap = graze_cap;
    std::vector<uint32_t> stack(rtd::kBvhStack, 0u);
    for (size_t i = 0; i < n; i++) {
        const rtd::V3 o = rtd::mk(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]);
        const rtd::V3 d = rtd::mk(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]);
        rtd::Counters cn;
        rtd::Hit h;
        if (use_bvh) h = rtd::closest_hit_bvh<false>(sc, nS, nM, o, d, cn, stack.data(), 1u);
        else h = rtd::closest_hit<false>(sc, nS, nM, o, d, cn);
        t_out[i] = h.t;
        prim_out[i] = h.prim;
    }
    return 0;
}

}  // extern "C"
