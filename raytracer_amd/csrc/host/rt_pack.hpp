// rt_pack.hpp — a scene in the reference's records (RayTracingStructs.h:24-60) packed into the one block of
// bytes the kernels read: repacked primitive arrays and, for large scenes, the hierarchies.  Pure host code:
// rtmi_upload_scene copies the block to the device; the tests' host emulation of the traversal reads it
// where it is.
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "rt_bvh.hpp"
#include "rtmi.h"

namespace rtmi {

struct PackedScene {
    std::vector<unsigned char> blob;
    // byte offsets of the arrays inside blob (all multiples of 256)
    size_t off_sph = 0, off_tri = 0, off_shd = 0, off_box = 0, off_mat = 0;
    size_t off_nodes = 0, off_bsph = 0, off_bsid = 0, off_btri = 0, off_btid = 0, off_mroot = 0, off_graze = 0;
    bool bvh = false;        // the hierarchy arrays are present
    QFrame qframe;           // frame of the quantised node boxes
    float slack = 0.0f;      // bound on the rounding of a dequantised plane that does not scale with distance
    float sphere_floor = 0.0f;
    uint32_t sphere_root = kEmptyRef;
    uint32_t node_count = 0;
    uint32_t graze_first = 0;  // node index of entry 0 of the graze records
    uint32_t n_spheres = 0, n_triangles = 0, n_meshes = 0;
    bool has_glass = false;  // some material has its transmission flag (first padding word) set
};

// Repack the records; with want_bvh also build one hierarchy over the spheres and one per mesh.  fused picks
// the arithmetic of AB x AC (the fused profile rounds it once per component).  transmission: read the first two
// padding words of every material as transmission flag and index of refraction (rtmi_enable_transmission);
// otherwise the padding is ignored, as the shader ignores it.  Returns 0 or a negative rtmi_status with the
// reason in err.
int PackScene(const rtmi_sphere *spheres, uint32_t nS, const rtmi_triangle *tris, uint32_t nT, const rtmi_mesh *meshes,
              uint32_t nM, bool want_bvh, bool fused, PackedScene &out, std::string &err, bool transmission = false);

}  // namespace rtmi

// Points a DeviceScene (rt_device.hpp; not named here so that this header needs no device types) at a packed
// scene whose bytes start at BASE.  A macro because the float4 / uint4 pointer types belong to whoever includes
// the device header.
#define RTMI_BIND_PACKED_SCENE(PK, BASE, DS)                                                                              \
    do {                                                                                                                  \
        const unsigned char *b_ = (BASE);                                                                                 \
        (DS).sphere_geom = reinterpret_cast<const float4 *>(b_ + (PK).off_sph);                                            \
        (DS).tri_isect = reinterpret_cast<const float4 *>(b_ + (PK).off_tri);                                              \
        (DS).tri_shade = reinterpret_cast<const float4 *>(b_ + (PK).off_shd);                                              \
        (DS).mesh_box = reinterpret_cast<const float4 *>(b_ + (PK).off_box);                                               \
        (DS).materials = reinterpret_cast<const float4 *>(b_ + (PK).off_mat);                                              \
        (DS).n_spheres = (PK).n_spheres;                                                                                   \
        (DS).n_triangles = (PK).n_triangles;                                                                               \
        (DS).n_meshes = (PK).n_meshes;                                                                                     \
        (DS).bvh_nodes = (PK).bvh ? reinterpret_cast<const uint4 *>(b_ + (PK).off_nodes) : nullptr;                        \
        for (int a_ = 0; a_ < 3; a_++) {                                                                                   \
            (DS).bvh_base[a_] = (PK).qframe.origin[a_];                                                                    \
            (DS).bvh_step[a_] = (PK).qframe.step[a_];                                                                      \
        }                                                                                                                 \
        (DS).bvh_slack = (PK).slack;                                                                                       \
        (DS).bvh_sphere_floor = (PK).sphere_floor;                                                                         \
        (DS).bvh_sphere_geom = (PK).bvh ? reinterpret_cast<const float4 *>(b_ + (PK).off_bsph) : nullptr;                  \
        (DS).bvh_sphere_id = (PK).bvh ? reinterpret_cast<const uint32_t *>(b_ + (PK).off_bsid) : nullptr;                  \
        (DS).bvh_tri_isect = (PK).bvh ? reinterpret_cast<const float4 *>(b_ + (PK).off_btri) : nullptr;                    \
        (DS).bvh_tri_id = (PK).bvh ? reinterpret_cast<const uint32_t *>(b_ + (PK).off_btid) : nullptr;                     \
        (DS).bvh_mesh_info = (PK).bvh ? reinterpret_cast<const uint4 *>(b_ + (PK).off_mroot) : nullptr;                    \
        (DS).bvh_sphere_root = (PK).sphere_root;                                                                           \
        (DS).bvh_node_count = (PK).bvh ? (PK).node_count : 0u;                                                             \
        (DS).has_glass = (PK).has_glass ? 1u : 0u;                                                                         \
        (DS).bvh_graze = (PK).bvh ? reinterpret_cast<const uint4 *>(b_ + (PK).off_graze) : nullptr;                        \
        (DS).bvh_graze_first = (PK).graze_first;                                                                           \
    } while (0)
