// rt_host.hpp — the host side that the path tracer keeps from the reference, as C++ with the
// reference's names: scene records, LoadModel, CameraFPS, the per-frame policy and headless image
// output.  No windowing, no Vulkan.  The C wrappers in rt_host_c.cpp export it through rtmi.h.
#pragma once

#include <array>
#include <cstdint>
#include <string>
#include <vector>

#include "rtmi.h"

namespace rtmi {

using FrameData = rtmi_frame_data;  // RayTracingStructs.h:4-22
using Material = rtmi_material;     // RayTracingStructs.h:24-32
using Sphere = rtmi_sphere;         // RayTracingStructs.h:34-39
using Mesh = rtmi_mesh;             // RayTracingStructs.h:41-50
using Triangle = rtmi_triangle;     // RayTracingStructs.h:52-60

struct Vec3 {
    float x = 0.0f, y = 0.0f, z = 0.0f;
};
using Mat4 = std::array<float, 16>;  // column-major

enum class QuadRule : int {
    ShorterDiagonal = 0,  // tie -> (0,1,3),(1,2,3)
    Fan = 1               // (0,1,2),(0,2,3)
};

// ModelLoader.h:6-11.  Returns false (and appends nothing) when the file cannot be read.
bool LoadModel(const char *filePath, std::vector<Triangle> &triangles, std::vector<Mesh> &meshes, const Material &material,
               Vec3 translate, Vec3 scale = Vec3{1.0f, 1.0f, 1.0f}, QuadRule rule = QuadRule::ShorterDiagonal);

// glm call sites of CameraFPS.cpp restated (left-handed, depth 0..1; VKHeaders.h:10-11)
Mat4 PerspectiveFovLH_ZO(float fov, float width, float height, float zNear, float zFar);
Mat4 LookAtLH(Vec3 eye, Vec3 center, Vec3 up);
Mat4 Inverse(const Mat4 &m);

struct CameraInput {
    double mouseX = 0.0, mouseY = 0.0;
    bool mouseButton2 = false;
    bool keyW = false, keyS = false, keyA = false, keyD = false;
    bool keySpace = false, keyLeftShift = false, keyF = false, keyC = false;
};

// CameraFPS.h:5-39 with the GLFW window replaced by explicit extent and input.
class CameraFPS {
public:
    CameraFPS(uint32_t windowWidth, uint32_t windowHeight, double mouseX = 0.0, double mouseY = 0.0);
    void Update(double deltaTime, const CameraInput &in);
    void SetPose(Vec3 position, float yaw, float pitch);

    float pitch = 0.0f;
    float yaw = -90.0f;
    float fov = 70.0f;  // handed to the projection unconverted, i.e. as radians (CameraFPS.cpp:23)
    float nearPlane = 0.1f;
    float farPlane = 1000.0f;
    float sensitivity = 0.2f;
    double lastMouseX = 0.0, lastMouseY = 0.0;
    bool moved = false;

    Mat4 projection{}, view{}, inverseProjection{}, inverseView{};
    Vec3 position, forward, up, right;
    uint32_t width, height;

private:
    void RecomputeForward();
    void RecomputeView();
};

// main.cpp:116-123, 151-153
void DefaultFrameData(FrameData &fd);
// main.cpp:332-335, 355-363
void FrameAdvance(FrameData &fd, const CameraFPS &camera, bool resetKey);

// main.cpp:161-185: the scene the reference builds, from mesh files in meshDir.
struct Scene {
    std::vector<Sphere> spheres;
    std::vector<Triangle> triangles;
    std::vector<Mesh> meshes;
};
bool BuildReferenceScene(const std::string &meshDir, Scene &scene, QuadRule rule = QuadRule::ShorterDiagonal);

// Text scene file: one record per line, floats with nine significant digits (exact for binary32).
// LoadScene replaces `scene` only when the whole file parses and every mesh range is in bounds.
bool SaveScene(const char *path, const Scene &scene);
bool LoadScene(const char *path, Scene &scene);

// Accumulation snapshot: the float4 running sum and the number of frames summed into it.
struct Snapshot {
    uint32_t width = 0, height = 0, frames = 0;
    std::vector<float> accum;  // width * height * 4
};
bool SaveSnapshot(const char *path, uint32_t width, uint32_t height, uint32_t frames, const float *accumRgba);
bool LoadSnapshot(const char *path, Snapshot &snap);

bool WritePPM(const char *path, const uint8_t *rgba8, uint32_t width, uint32_t height, bool srgb);
bool WritePFM(const char *path, const float *accumRgba, uint32_t width, uint32_t height, uint32_t frames);
bool WriteEXR(const char *path, const float *accumRgba, uint32_t width, uint32_t height, uint32_t frames);

}  // namespace rtmi
