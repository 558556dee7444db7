// headless_main.cpp — the reference's main() (main.cpp:11-452) without a window: build the scene of
// main.cpp:161-185 (or read a scene file), run the frame loop of main.cpp:236-444 through the C ABI
// with the camera input a script supplies instead of GLFW, write the picture.  Uses only include/rtmi.h.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <thread>
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
    std::string ppm, pfm, exr;
    int variant = RTMI_VARIANT_AUTO;
    bool batch = false;
    int move_at = -1;  // frame at which the W key is held for one frame (exercises reset-on-move)
    std::string scene, script, resume, checkpoint;
    int device = 0;
    int gpus = 0;  // > 0: frames split over GPUs 0 .. gpus-1 of this node, images summed with RCCL (implies --batch)
    int quad_rule = 0;
    bool transmission = false;   // the transmitting-material extension (rtmi_enable_transmission)
    int profile = RTMI_PROFILE_STRICT;  // arithmetic profile (--profile, default from the environment variable RTMI_PROFILE)
};

void Usage()
{
    std::puts("rtmi_headless [--size W H] [--frames N] [--rpp N] [--bounces N] [--pose x y z yaw pitch]\n"
              "              [--meshes DIR] [--ppm FILE] [--pfm FILE] [--exr FILE] [--variant 0|1|2|3|4] [--batch]\n"
              "              [--move-at FRAME] [--device N] [--fan] [--profile strict|fma] [--transmission]\n"
              "              [--gpus N]           static camera: frames 1..N split over GPUs 0..N-1, one RCCL sum at the end\n"
              "              [--scene FILE]       text scene file instead of the built-in scene\n"
              "              [--script FILE]      camera input, one line per run of frames:\n"
              "                                   FRAMES DT KEYS MOUSE_X MOUSE_Y BUTTON2 RESET\n"
              "                                   KEYS from W S A D F C and ^ (space) v (left shift), or -\n"
              "              [--resume FILE] [--checkpoint FILE]   accumulation snapshots (static camera)");
}

// One line of a camera script: the input held for `frames` consecutive frames.
struct ScriptStep {
    uint32_t frames = 0;
    double dt = 0.0;
    rtmi_camera_input in{};
    int reset = 0;
};

bool ReadScript(const std::string &path, std::vector<ScriptStep> &steps)
{
    std::ifstream f(path);
    if (!f) return false;
    std::string line;
    while (std::getline(f, line)) {
        if (line.empty() || line[0] == '#') continue;
        std::istringstream ls(line);
        ScriptStep st;
        std::string keys;
        int button = 0;
        if (!(ls >> st.frames >> st.dt >> keys >> st.in.mouse_x >> st.in.mouse_y >> button >> st.reset)) return false;
        st.in.mouse_button_2 = button;
        for (char k : keys) {
            switch (k) {
            case 'W': st.in.key_w = 1; break;
            case 'S': st.in.key_s = 1; break;
            case 'A': st.in.key_a = 1; break;
            case 'D': st.in.key_d = 1; break;
            case 'F': st.in.key_f = 1; break;
            case 'C': st.in.key_c = 1; break;
            case '^': st.in.key_space = 1; break;
            case 'v': st.in.key_left_shift = 1; break;
            case '-': break;
            default: return false;
            }
        }
        steps.push_back(st);
    }
    return true;
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
    auto profile_by_name = [](const std::string &name, int &out) {
        if (name == "strict") out = RTMI_PROFILE_STRICT;
        else if (name == "fma") out = RTMI_PROFILE_FMA;
        else return false;
        return true;
    };
    if (const char *env = std::getenv("RTMI_PROFILE")) {
        if (*env && !profile_by_name(env, o.profile)) {
            std::fprintf(stderr, "rtmi_headless: RTMI_PROFILE=%s is neither strict nor fma\n", env);
            return 2;
        }
    }
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
        else if (a == "--exr" && need(1)) o.exr = argv[++i];
        else if (a == "--variant" && need(1)) o.variant = std::atoi(argv[++i]);
        else if (a == "--batch") o.batch = true;
        else if (a == "--move-at" && need(1)) o.move_at = std::atoi(argv[++i]);
        else if (a == "--device" && need(1)) o.device = std::atoi(argv[++i]);
        else if (a == "--gpus" && need(1)) { o.gpus = std::atoi(argv[++i]); o.batch = true; }
        else if (a == "--fan") o.quad_rule = 1;
        else if (a == "--transmission") o.transmission = true;
        else if (a == "--profile" && need(1)) { if (!profile_by_name(argv[++i], o.profile)) { Usage(); return 2; } }
        else if (a == "--scene" && need(1)) o.scene = argv[++i];
        else if (a == "--script" && need(1)) o.script = argv[++i];
        else if (a == "--resume" && need(1)) o.resume = argv[++i];
        else if (a == "--checkpoint" && need(1)) o.checkpoint = argv[++i];
        else { Usage(); return a == "--help" ? 0 : 2; }
    }

    std::vector<rtmi_sphere> spheres;
    std::vector<rtmi_triangle> tris;
    std::vector<rtmi_mesh> meshes;
    uint32_t nT = 0, nM = 0;
    if (!o.scene.empty()) {
        uint32_t nS = 0;
        int rc = rtmi_scene_load(o.scene.c_str(), nullptr, &nS, 0, nullptr, &nT, 0, nullptr, &nM, 0);
        if (rc == RTMI_OK) {
            spheres.resize(nS + 1);
            tris.resize(nT + 1);
            meshes.resize(nM + 1);
            rc = rtmi_scene_load(o.scene.c_str(), spheres.data(), &nS, nS, tris.data(), &nT, nT, meshes.data(), &nM, nM);
        }
        if (rc != RTMI_OK) {
            std::fprintf(stderr, "cannot read scene file %s (%d)\n", o.scene.c_str(), rc);
            return 1;
        }
        spheres.resize(nS);
    } else {
        // scene: main.cpp:161-185
        spheres.resize(1);
        std::memset(spheres.data(), 0, sizeof(rtmi_sphere));
        spheres[0].center[0] = 1.0f; spheres[0].center[1] = 1.0f; spheres[0].center[2] = 0.0f;
        spheres[0].radius = 0.5f;
        spheres[0].material = rtmi_material{{1.0f, 1.0f, 1.0f}, 0.0f, 0.1f, 0.0f, 0.0f, 0.0f};
        tris.resize(64);
        meshes.resize(4);
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
    }
    std::vector<ScriptStep> script;
    if (!o.script.empty() && !ReadScript(o.script, script)) {
        std::fprintf(stderr, "cannot read camera script %s\n", o.script.c_str());
        return 1;
    }

    // --gpus N: one context per GPU, each set up by a thread of its own (an upload may build hierarchies)
    std::vector<rtmi_ctx *> group;
    if (o.gpus > 0) {
        const int have = rtmi_device_count();
        // with the tests' stand-in for RCCL (RTMI_RCCL_LIBRARY) the contexts may share the GPUs there are
        const bool may_share = have >= 1 && have < o.gpus && rtmi_reduce_same_device_allowed() == 1;
        if (have < o.gpus && !may_share) {
            std::fprintf(stderr, "rtmi_headless: --gpus %d asked for, but only %d device%s visible on this machine\n", o.gpus,
                         have < 0 ? 0 : have, have == 1 ? " is" : "s are");
            return 1;
        }
        if (!o.resume.empty() || !o.script.empty()) {
            std::fprintf(stderr, "rtmi_headless: --gpus renders a static camera from frame 1; --resume and --script do not apply\n");
            return 2;
        }
        group.assign((size_t)o.gpus, nullptr);
        std::vector<int> ok((size_t)o.gpus, 0);
        std::vector<std::thread> setup;
        for (int g = 0; g < o.gpus; g++)
            setup.emplace_back([&, g] {
                rtmi_ctx *c = nullptr;
                if (!Check(nullptr, rtmi_create(g % have, o.width, o.height, &c), "rtmi_create")) return;
                group[(size_t)g] = c;
                if (!Check(c, rtmi_set_variant(c, o.variant), "rtmi_set_variant")) return;
                if (!Check(c, rtmi_set_profile(c, o.profile), "rtmi_set_profile")) return;
                if (!Check(c, rtmi_enable_transmission(c, o.transmission ? 1 : 0), "rtmi_enable_transmission")) return;
                if (!Check(c, rtmi_upload_scene(c, spheres.data(), (uint32_t)spheres.size(), tris.data(), nT, meshes.data(), nM),
                           "rtmi_upload_scene"))
                    return;
                ok[(size_t)g] = 1;
            });
        for (std::thread &t : setup) t.join();
        for (int g = 0; g < o.gpus; g++)
            if (!ok[(size_t)g]) return 1;
    }

    rtmi_ctx *ctx = nullptr;
    if (!group.empty()) {
        ctx = group[0];   // the root: holds the sum, is read back and written out
    } else {
        if (!Check(nullptr, rtmi_create(o.device, o.width, o.height, &ctx), "rtmi_create")) return 1;
        if (!Check(ctx, rtmi_set_variant(ctx, o.variant), "rtmi_set_variant")) return 1;
        if (!Check(ctx, rtmi_set_profile(ctx, o.profile), "rtmi_set_profile")) return 1;
        if (!Check(ctx, rtmi_enable_transmission(ctx, o.transmission ? 1 : 0), "rtmi_enable_transmission")) return 1;
        if (!Check(ctx, rtmi_upload_scene(ctx, spheres.data(), (uint32_t)spheres.size(), tris.data(), nT, meshes.data(), nM),
                   "rtmi_upload_scene"))
            return 1;
    }

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

    uint32_t frames_done = 0;
    if (!o.resume.empty() && !Check(ctx, rtmi_checkpoint_load(ctx, o.resume.c_str(), &frames_done), "rtmi_checkpoint_load")) return 1;

    const auto t0 = std::chrono::steady_clock::now();
    uint32_t shown_frame_index = 0, rendered = 0;
    if (!group.empty()) {
        // Frames 1 .. N in contiguous, balanced ranges, one per GPU, each summed into a cleared image (a frame's
        // contribution depends only on pixel, frame index, scene and camera: Raytracing.comp:462); then one RCCL
        // reduce onto GPU 0, which puts alpha back to 1 and rebuilds the display image as sum / N.
        rtmi_camera_input in{};
        rtmi_camera_update(&cam, 0.0, &in);
        rtmi_frame_advance(&fd, &cam, 0);
        const uint32_t world = (uint32_t)group.size(), q = o.frames / world, rem = o.frames % world;
        for (uint32_t g = 0; g < world; g++) {
            const uint32_t n = q + (g < rem ? 1u : 0u), first = 1u + g * q + (g < rem ? g : rem);
            if (!Check(group[g], rtmi_clear_accum(group[g]), "rtmi_clear_accum")) return 1;
            if (n && !Check(group[g], rtmi_render_frames(group[g], &fd, first, n), "rtmi_render_frames")) return 1;
        }
        if (!Check(ctx, rtmi_reduce_accum(group.data(), world, 0), "rtmi_reduce_accum")) return 1;
        if (!Check(ctx, rtmi_resolve_output(ctx, o.frames), "rtmi_resolve_output")) return 1;
        for (uint32_t g = 1; g < world; g++)
            if (!Check(group[g], rtmi_synchronize(group[g]), "rtmi_synchronize")) return 1;
        shown_frame_index = o.frames;
        rendered = o.frames;
    } else if (o.batch) {
        rtmi_camera_input in{};
        rtmi_camera_update(&cam, 0.0, &in);
        rtmi_frame_advance(&fd, &cam, 0);  // frame_index becomes 1
        if (!Check(ctx, rtmi_render_frames(ctx, &fd, frames_done + 1, o.frames), "rtmi_render_frames")) return 1;
        shown_frame_index = frames_done + o.frames;
        rendered = o.frames;
    } else if (!script.empty()) {
        // main.cpp:236-444 with the script standing in for the window's input
        fd.frame_index = frames_done;
        for (const ScriptStep &st : script)
            for (uint32_t f = 0; f < st.frames; f++) {
                rtmi_camera_update(&cam, st.dt, &st.in);
                rtmi_frame_advance(&fd, &cam, st.reset);
                if (!Check(ctx, rtmi_render(ctx, &fd), "rtmi_render")) return 1;
                shown_frame_index = fd.frame_index;
                rendered++;
            }
    } else {
        fd.frame_index = frames_done;
        for (uint32_t f = 0; f < o.frames; f++) {
            rtmi_camera_input in{};
            in.key_w = ((int)f == o.move_at) ? 1 : 0;
            rtmi_camera_update(&cam, 1.0 / 60.0, &in);
            rtmi_frame_advance(&fd, &cam, 0);
            if (!Check(ctx, rtmi_render(ctx, &fd), "rtmi_render")) return 1;
            shown_frame_index = fd.frame_index;
            rendered++;
        }
    }
    if (!Check(ctx, rtmi_synchronize(ctx), "rtmi_synchronize")) return 1;
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const double paths = (double)o.width * o.height * rendered * fd.rays_per_pixel;
    std::printf("{\"width\": %u, \"height\": %u, \"frames\": %u, \"rays_per_pixel\": %u, \"max_bounce\": %u, "
                "\"seconds\": %.6f, \"mpaths_per_s\": %.3f, \"frame_index\": %u, \"gpus\": %d, \"profile\": \"%s\"}\n",
                o.width, o.height, rendered, fd.rays_per_pixel, fd.max_bounce_limit, secs, paths / secs / 1e6,
                shown_frame_index, group.empty() ? 1 : (int)group.size(), rtmi_get_profile(ctx) == RTMI_PROFILE_FMA ? "fma" : "strict");

    if (!o.checkpoint.empty() && !Check(ctx, rtmi_checkpoint_save(ctx, o.checkpoint.c_str(), shown_frame_index), "rtmi_checkpoint_save"))
        return 1;

    const size_t np = (size_t)o.width * o.height;
    if (!o.ppm.empty()) {
        std::vector<uint8_t> img(np * 4);
        if (!Check(ctx, rtmi_read_output_rgba8(ctx, img.data(), img.size()), "rtmi_read_output_rgba8")) return 1;
        if (rtmi_write_ppm(o.ppm.c_str(), img.data(), o.width, o.height, 1) != RTMI_OK) {
            std::fprintf(stderr, "cannot write %s\n", o.ppm.c_str());
            return 1;
        }
    }
    if (!o.pfm.empty() || !o.exr.empty()) {
        std::vector<float> acc(np * 4);
        if (!Check(ctx, rtmi_read_accum(ctx, acc.data(), acc.size() * sizeof(float)), "rtmi_read_accum")) return 1;
        if (!o.pfm.empty() && rtmi_write_pfm(o.pfm.c_str(), acc.data(), o.width, o.height, shown_frame_index) != RTMI_OK) {
            std::fprintf(stderr, "cannot write %s\n", o.pfm.c_str());
            return 1;
        }
        if (!o.exr.empty() && rtmi_write_exr(o.exr.c_str(), acc.data(), o.width, o.height, shown_frame_index) != RTMI_OK) {
            std::fprintf(stderr, "cannot write %s\n", o.exr.c_str());
            return 1;
        }
    }
    if (group.empty()) rtmi_destroy(ctx);
    if (!group.empty()) rtmi_shutdown();   // communicators of the sum, before the contexts' streams go
    for (rtmi_ctx *c : group) rtmi_destroy(c);
    return 0;
}
