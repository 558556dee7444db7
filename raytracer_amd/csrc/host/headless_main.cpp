// headless_main.cpp — the reference's main() (main.cpp:11-452) without a window: build the scene of
// main.cpp:161-185, run the frame loop of main.cpp:236-444 through the C ABI, write the picture.
// Uses only include/rtmi.h.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rtmi.h"

namespace {

struct Options {
    uint32_t width = 1280, height = 720;  // main.cpp:15
    uint32_t frames = 64;
    int rpp = -1, bounces = -1;  // -1: keep main.cpp:118-119 (4, 6)
    bool have_pose = false;
    float pose[5] = {0.0f, 1.0f, -1.0f, -90.0f, 0.0f};
    std::string meshes = "assets/meshes";
    std::string ppm, pfm;
    int variant = RTMI_VARIANT_REGENERATE;
    bool batch = false;
    int move_at = -1;  // frame at which the W key is held for one frame (exercises reset-on-move)
    int device = 0;
    int quad_rule = 0;
};

void Usage()
{
    std::puts("rtmi_headless [--size W H] [--frames N] [--rpp N] [--bounces N] [--pose x y z yaw pitch]\n"
              "              [--meshes DIR] [--ppm FILE] [--pfm FILE] [--variant 0|1] [--batch]\n"
              "              [--move-at FRAME] [--device N] [--fan]");
}

bool Check(rtmi_ctx *ctx, int rc, const char *what)
{
    if (rc >= 0) return true;
    std::fprintf(stderr, "%s failed (%d): %s\n", what, rc, rtmi_last_error(ctx));
    return false;
}

}  // namespace

int main(int argc, char **argv)
{
    Options o;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto need = [&](int n) { return i + n < argc; };
        if (a == "--size" && need(2)) { o.width = std::atoi(argv[++i]); o.height = std::atoi(argv[++i]); }
        else if (a == "--frames" && need(1)) o.frames = std::atoi(argv[++i]);
        else if (a == "--rpp" && need(1)) o.rpp = std::atoi(argv[++i]);
        else if (a == "--bounces" && need(1)) o.bounces = std::atoi(argv[++i]);
        else if (a == "--pose" && need(5)) { for (int k = 0; k < 5; k++) o.pose[k] = (float)std::atof(argv[++i]); o.have_pose = true; }
        else if (a == "--meshes" && need(1)) o.meshes = argv[++i];
        else if (a == "--ppm" && need(1)) o.ppm = argv[++i];
        else if (a == "--pfm" && need(1)) o.pfm = argv[++i];
        else if (a == "--variant" && need(1)) o.variant = std::atoi(argv[++i]);
        else if (a == "--batch") o.batch = true;
        else if (a == "--move-at" && need(1)) o.move_at = std::atoi(argv[++i]);
        else if (a == "--device" && need(1)) o.device = std::atoi(argv[++i]);
        else if (a == "--fan") o.quad_rule = 1;
        else { Usage(); return a == "--help" ? 0 : 2; }
    }

    // scene: main.cpp:161-185
    std::vector<rtmi_sphere> spheres(1);
    std::memset(spheres.data(), 0, sizeof(rtmi_sphere));
    spheres[0].center[0] = 1.0f; spheres[0].center[1] = 1.0f; spheres[0].center[2] = 0.0f;
    spheres[0].radius = 0.5f;
    spheres[0].material = rtmi_material{{1.0f, 1.0f, 1.0f}, 0.0f, 0.1f, 0.0f, 0.0f, 0.0f};
    std::vector<rtmi_triangle> tris(64);
    std::vector<rtmi_mesh> meshes(4);
    uint32_t nT = 0, nM = 0;
    const rtmi_material planeMat{{1.0f, 1.0f, 1.0f}, 0.0f, 0.8f, 0.0f, 0.0f, 0.0f};
    const rtmi_material cubeMat{{0.9f, 0.9f, 0.9f}, 0.0f, 0.1f, 0.0f, 0.0f, 0.0f};
    const float zero[3] = {0.0f, 0.0f, 0.0f}, one[3] = {1.0f, 1.0f, 1.0f};
    const float planeScale[3] = {2.0f, 1.0f, 2.0f}, cubeMove[3] = {-1.0f, 1.0f, 0.0f};
    int rc = rtmi_load_model((o.meshes + "/plane.obj").c_str(), tris.data(), &nT, (uint32_t)tris.size(), meshes.data(), &nM,
                             (uint32_t)meshes.size(), &planeMat, zero, planeScale, o.quad_rule);
    if (rc < 0) std::fprintf(stderr, "warning: plane.obj not loaded from %s (%d)\n", o.meshes.c_str(), rc);
    rc = rtmi_load_model((o.meshes + "/cube.obj").c_str(), tris.data(), &nT, (uint32_t)tris.size(), meshes.data(), &nM,
                         (uint32_t)meshes.size(), &cubeMat, cubeMove, one, o.quad_rule);
    if (rc < 0) std::fprintf(stderr, "warning: cube.obj not loaded from %s (%d)\n", o.meshes.c_str(), rc);

    rtmi_ctx *ctx = nullptr;
    if (!Check(nullptr, rtmi_create(o.device, o.width, o.height, &ctx), "rtmi_create")) return 1;
    if (!Check(ctx, rtmi_set_variant(ctx, o.variant), "rtmi_set_variant")) return 1;
    if (!Check(ctx, rtmi_upload_scene(ctx, spheres.data(), (uint32_t)spheres.size(), tris.data(), nT, meshes.data(), nM),
               "rtmi_upload_scene"))
        return 1;

    rtmi_frame_data fd;
    rtmi_default_frame_data(&fd);
    if (o.rpp >= 0) fd.rays_per_pixel = (uint32_t)o.rpp;
    if (o.bounces >= 0) fd.max_bounce_limit = (uint32_t)o.bounces;
    fd.sphere_number = (uint32_t)spheres.size();
    fd.mesh_number = nM;

    rtmi_camera cam;
    rtmi_camera_init(&cam, o.width, o.height, 0.0, 0.0);
    if (o.have_pose) {
        rtmi_camera_set_pose(&cam, o.pose, o.pose[3], o.pose[4]);
        cam.moved = 0;
    }

    const auto t0 = std::chrono::steady_clock::now();
    uint32_t shown_frame_index = 0;
    if (o.batch) {
        rtmi_camera_input in{};
        rtmi_camera_update(&cam, 0.0, &in);
        rtmi_frame_advance(&fd, &cam, 0);  // frame_index becomes 1
        if (!Check(ctx, rtmi_render_frames(ctx, &fd, 1, o.frames), "rtmi_render_frames")) return 1;
        shown_frame_index = o.frames;
    } else {
        for (uint32_t f = 0; f < o.frames; f++) {
            rtmi_camera_input in{};
            in.key_w = ((int)f == o.move_at) ? 1 : 0;
            rtmi_camera_update(&cam, 1.0 / 60.0, &in);
            rtmi_frame_advance(&fd, &cam, 0);
            if (!Check(ctx, rtmi_render(ctx, &fd), "rtmi_render")) return 1;
            shown_frame_index = fd.frame_index;
        }
    }
    if (!Check(ctx, rtmi_synchronize(ctx), "rtmi_synchronize")) return 1;
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const double paths = (double)o.width * o.height * o.frames * fd.rays_per_pixel;
    std::printf("{\"width\": %u, \"height\": %u, \"frames\": %u, \"rays_per_pixel\": %u, \"max_bounce\": %u, "
                "\"seconds\": %.6f, \"mpaths_per_s\": %.3f, \"frame_index\": %u}\n",
                o.width, o.height, o.frames, fd.rays_per_pixel, fd.max_bounce_limit, secs, paths / secs / 1e6,
                shown_frame_index);

    const size_t np = (size_t)o.width * o.height;
    if (!o.ppm.empty()) {
        std::vector<uint8_t> img(np * 4);
        if (!Check(ctx, rtmi_read_output_rgba8(ctx, img.data(), img.size()), "rtmi_read_output_rgba8")) return 1;
        if (rtmi_write_ppm(o.ppm.c_str(), img.data(), o.width, o.height, 1) != RTMI_OK) {
            std::fprintf(stderr, "cannot write %s\n", o.ppm.c_str());
            return 1;
        }
    }
    if (!o.pfm.empty()) {
        std::vector<float> acc(np * 4);
        if (!Check(ctx, rtmi_read_accum(ctx, acc.data(), acc.size() * sizeof(float)), "rtmi_read_accum")) return 1;
        if (rtmi_write_pfm(o.pfm.c_str(), acc.data(), o.width, o.height, shown_frame_index) != RTMI_OK) {
            std::fprintf(stderr, "cannot write %s\n", o.pfm.c_str());
            return 1;
        }
    }
    rtmi_destroy(ctx);
    return 0;
}
