// rt_host_c.cpp — C wrappers (rtmi.h) over the C++ host classes of rt_host.hpp.
#include <cstring>
#include <new>
#include <vector>

#include "rt_bvh.hpp"
#include "rt_host.hpp"

namespace {

void CameraToC(const rtmi::CameraFPS &c, rtmi_camera *o)
{
    o->pitch = c.pitch;
    o->yaw = c.yaw;
    o->fov = c.fov;
    o->near_plane = c.nearPlane;
    o->far_plane = c.farPlane;
    o->sensitivity = c.sensitivity;
    o->moved = c.moved ? 1 : 0;
    std::memcpy(o->projection, c.projection.data(), 64);
    std::memcpy(o->view, c.view.data(), 64);
    std::memcpy(o->inverse_projection, c.inverseProjection.data(), 64);
    std::memcpy(o->inverse_view, c.inverseView.data(), 64);
    const rtmi::Vec3 *src[4] = {&c.position, &c.forward, &c.up, &c.right};
    float *dst[4] = {o->position, o->forward, o->up, o->right};
    for (int i = 0; i < 4; i++) {
        dst[i][0] = src[i]->x;
        dst[i][1] = src[i]->y;
        dst[i][2] = src[i]->z;
    }
    o->last_mouse_x = c.lastMouseX;
    o->last_mouse_y = c.lastMouseY;
    o->width = c.width;
    o->height = c.height;
}

rtmi::CameraFPS CameraFromC(const rtmi_camera *i)
{
    rtmi::CameraFPS c(i->width, i->height, i->last_mouse_x, i->last_mouse_y);
    c.pitch = i->pitch;
    c.yaw = i->yaw;
    c.fov = i->fov;
    c.nearPlane = i->near_plane;
    c.farPlane = i->far_plane;
    c.sensitivity = i->sensitivity;
    c.moved = i->moved != 0;
    std::memcpy(c.projection.data(), i->projection, 64);
    std::memcpy(c.view.data(), i->view, 64);
    std::memcpy(c.inverseProjection.data(), i->inverse_projection, 64);
    std::memcpy(c.inverseView.data(), i->inverse_view, 64);
    c.position = {i->position[0], i->position[1], i->position[2]};
    c.forward = {i->forward[0], i->forward[1], i->forward[2]};
    c.up = {i->up[0], i->up[1], i->up[2]};
    c.right = {i->right[0], i->right[1], i->right[2]};
    return c;
}

}  // namespace

extern "C" {

int rtmi_load_model(const char *path, rtmi_triangle *triangles, uint32_t *n_triangles, uint32_t triangle_capacity,
                    rtmi_mesh *meshes, uint32_t *n_meshes, uint32_t mesh_capacity, const rtmi_material *material,
                    const float translate[3], const float scale[3], int quad_rule)
{
    if (!path || !triangles || !n_triangles || !meshes || !n_meshes || !material || !translate || !scale)
        return RTMI_ERR_INVALID_ARGUMENT;
    if (quad_rule != 0 && quad_rule != 1) return RTMI_ERR_INVALID_ARGUMENT;
    try {
        // LoadModel records start_triangle = triangles.size(); keep the caller's count as the base
        std::vector<rtmi::Triangle> tris(*n_triangles);
        std::vector<rtmi::Mesh> ms;
        if (!rtmi::LoadModel(path, tris, ms, *material, rtmi::Vec3{translate[0], translate[1], translate[2]},
                             rtmi::Vec3{scale[0], scale[1], scale[2]}, static_cast<rtmi::QuadRule>(quad_rule)))
            return RTMI_ERR_IO;
        const size_t added = tris.size() - *n_triangles;
        if (tris.size() > triangle_capacity || *n_meshes + ms.size() > mesh_capacity) return RTMI_ERR_OUT_OF_MEMORY;
        std::memcpy(triangles + *n_triangles, tris.data() + *n_triangles, added * sizeof(rtmi_triangle));
        for (const auto &m : ms) meshes[(*n_meshes)++] = m;
        *n_triangles = (uint32_t)tris.size();
        return (int)added;
    } catch (const std::bad_alloc &) {
        return RTMI_ERR_OUT_OF_MEMORY;
    } catch (...) {
        return RTMI_ERR_IO;
    }
}

void rtmi_default_frame_data(rtmi_frame_data *fd)
{
    if (fd) rtmi::DefaultFrameData(*fd);
}

void rtmi_camera_init(rtmi_camera *cam, uint32_t width, uint32_t height, double mouse_x, double mouse_y)
{
    if (!cam) return;
    rtmi::CameraFPS c(width, height, mouse_x, mouse_y);
    CameraToC(c, cam);
}

void rtmi_camera_set_pose(rtmi_camera *cam, const float position[3], float yaw, float pitch)
{
    if (!cam || !position) return;
    rtmi::CameraFPS c = CameraFromC(cam);
    c.SetPose(rtmi::Vec3{position[0], position[1], position[2]}, yaw, pitch);
    CameraToC(c, cam);
}

void rtmi_camera_update(rtmi_camera *cam, double delta_time, const rtmi_camera_input *in)
{
    if (!cam || !in) return;
    rtmi::CameraFPS c = CameraFromC(cam);
    rtmi::CameraInput ci;
    ci.mouseX = in->mouse_x;
    ci.mouseY = in->mouse_y;
    ci.mouseButton2 = in->mouse_button_2 != 0;
    ci.keyW = in->key_w != 0;
    ci.keyS = in->key_s != 0;
    ci.keyA = in->key_a != 0;
    ci.keyD = in->key_d != 0;
    ci.keySpace = in->key_space != 0;
    ci.keyLeftShift = in->key_left_shift != 0;
    ci.keyF = in->key_f != 0;
    ci.keyC = in->key_c != 0;
    c.Update(delta_time, ci);
    CameraToC(c, cam);
}

void rtmi_frame_advance(rtmi_frame_data *fd, const rtmi_camera *cam, int reset_key)
{
    if (!fd || !cam) return;
    rtmi::CameraFPS c = CameraFromC(cam);
    rtmi::FrameAdvance(*fd, c, reset_key != 0);
}

int rtmi_bvh_build(const float *boxes, uint32_t n, uint32_t leaf_base, void *nodes_out, size_t nodes_bytes,
                   uint32_t *order_out, uint32_t *root_out)
{
    if ((n && !boxes) || (n && !order_out) || !root_out) return RTMI_ERR_INVALID_ARGUMENT;
    if (n > rtmi::kMaxPrimitives || (uint64_t)leaf_base + n >= (1u << 27)) return RTMI_ERR_UNSUPPORTED;
    try {
        std::vector<rtmi::PrimBox> pb(n);
        for (uint32_t i = 0; i < n; i++)
            for (int a = 0; a < 3; a++) {
                pb[i].lo[a] = boxes[6 * (size_t)i + a];
                pb[i].hi[a] = boxes[6 * (size_t)i + 3 + a];
            }
        std::vector<rtmi::BvhNode> nodes;
        std::vector<uint32_t> order;
        *root_out = rtmi::BuildBvh(pb, nodes, order, leaf_base);
        if (nodes.size() * sizeof(rtmi::BvhNode) > nodes_bytes || (!nodes.empty() && !nodes_out)) return RTMI_ERR_OUT_OF_MEMORY;
        if (!nodes.empty()) std::memcpy(nodes_out, nodes.data(), nodes.size() * sizeof(rtmi::BvhNode));
        if (n) std::memcpy(order_out, order.data(), (size_t)n * sizeof(uint32_t));
        return (int)nodes.size();
    } catch (const std::bad_alloc &) {
        return RTMI_ERR_OUT_OF_MEMORY;
    } catch (...) {
        return RTMI_ERR_UNSUPPORTED;
    }
}

int rtmi_bvh_quantize(const void *nodes, uint32_t n_nodes, void *qnodes_out, size_t qnodes_bytes, float *frame_out)
{
    if ((n_nodes && !nodes) || (n_nodes && !qnodes_out) || !frame_out) return RTMI_ERR_INVALID_ARGUMENT;
    if ((size_t)n_nodes * sizeof(rtmi::QNode) > qnodes_bytes) return RTMI_ERR_OUT_OF_MEMORY;
    try {
        std::vector<rtmi::BvhNode> in(n_nodes);
        if (n_nodes) std::memcpy(in.data(), nodes, (size_t)n_nodes * sizeof(rtmi::BvhNode));
        std::vector<rtmi::QNode> q;
        rtmi::QFrame fr;
        rtmi::QuantizeNodes(in, fr, q);
        if (q.size() != n_nodes) return RTMI_ERR_UNSUPPORTED;
        if (n_nodes) std::memcpy(qnodes_out, q.data(), (size_t)n_nodes * sizeof(rtmi::QNode));
        for (int a = 0; a < 3; a++) {
            frame_out[a] = fr.origin[a];
            frame_out[3 + a] = fr.step[a];
        }
        return (int)n_nodes;
    } catch (const std::bad_alloc &) {
        return RTMI_ERR_OUT_OF_MEMORY;
    } catch (...) {
        return RTMI_ERR_UNSUPPORTED;
    }
}

int rtmi_scene_save(const char *path, const rtmi_sphere *spheres, uint32_t n_spheres, const rtmi_triangle *triangles,
                    uint32_t n_triangles, const rtmi_mesh *meshes, uint32_t n_meshes)
{
    if (!path || (n_spheres && !spheres) || (n_triangles && !triangles) || (n_meshes && !meshes)) return RTMI_ERR_INVALID_ARGUMENT;
    try {
        rtmi::Scene sc;
        sc.spheres.assign(spheres, spheres + n_spheres);
        sc.triangles.assign(triangles, triangles + n_triangles);
        sc.meshes.assign(meshes, meshes + n_meshes);
        return rtmi::SaveScene(path, sc) ? RTMI_OK : RTMI_ERR_IO;
    } catch (const std::bad_alloc &) {
        return RTMI_ERR_OUT_OF_MEMORY;
    } catch (...) {
        return RTMI_ERR_IO;
    }
}

int rtmi_scene_load(const char *path, rtmi_sphere *spheres, uint32_t *n_spheres, uint32_t sphere_capacity,
                    rtmi_triangle *triangles, uint32_t *n_triangles, uint32_t triangle_capacity, rtmi_mesh *meshes,
                    uint32_t *n_meshes, uint32_t mesh_capacity)
{
    if (!path || !n_spheres || !n_triangles || !n_meshes) return RTMI_ERR_INVALID_ARGUMENT;
    try {
        rtmi::Scene sc;
        if (!rtmi::LoadScene(path, sc)) return RTMI_ERR_IO;
        *n_spheres = (uint32_t)sc.spheres.size();
        *n_triangles = (uint32_t)sc.triangles.size();
        *n_meshes = (uint32_t)sc.meshes.size();
        if (!spheres && !triangles && !meshes) return RTMI_OK;  // size query
        if (sc.spheres.size() > sphere_capacity || sc.triangles.size() > triangle_capacity || sc.meshes.size() > mesh_capacity)
            return RTMI_ERR_OUT_OF_MEMORY;
        if ((!spheres && !sc.spheres.empty()) || (!triangles && !sc.triangles.empty()) || (!meshes && !sc.meshes.empty()))
            return RTMI_ERR_INVALID_ARGUMENT;
        if (!sc.spheres.empty()) std::memcpy(spheres, sc.spheres.data(), sc.spheres.size() * sizeof(rtmi_sphere));
        if (!sc.triangles.empty()) std::memcpy(triangles, sc.triangles.data(), sc.triangles.size() * sizeof(rtmi_triangle));
        if (!sc.meshes.empty()) std::memcpy(meshes, sc.meshes.data(), sc.meshes.size() * sizeof(rtmi_mesh));
        return RTMI_OK;
    } catch (const std::bad_alloc &) {
        return RTMI_ERR_OUT_OF_MEMORY;
    } catch (...) {
        return RTMI_ERR_IO;
    }
}

int rtmi_write_ppm(const char *path, const uint8_t *rgba8, uint32_t width, uint32_t height, int srgb)
{
    if (!path || !rgba8 || !width || !height) return RTMI_ERR_INVALID_ARGUMENT;
    return rtmi::WritePPM(path, rgba8, width, height, srgb != 0) ? RTMI_OK : RTMI_ERR_IO;
}

int rtmi_write_pfm(const char *path, const float *accum_rgba, uint32_t width, uint32_t height, uint32_t frames)
{
    if (!path || !accum_rgba || !width || !height) return RTMI_ERR_INVALID_ARGUMENT;
    return rtmi::WritePFM(path, accum_rgba, width, height, frames) ? RTMI_OK : RTMI_ERR_IO;
}

int rtmi_write_exr(const char *path, const float *accum_rgba, uint32_t width, uint32_t height, uint32_t frames)
{
    if (!path || !accum_rgba || !width || !height) return RTMI_ERR_INVALID_ARGUMENT;
    try {
        return rtmi::WriteEXR(path, accum_rgba, width, height, frames) ? RTMI_OK : RTMI_ERR_IO;
    } catch (const std::bad_alloc &) {
        return RTMI_ERR_OUT_OF_MEMORY;
    } catch (...) {
        return RTMI_ERR_IO;
    }
}

}  // extern "C"
