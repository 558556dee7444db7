// rt_host.cpp — scene loading, camera and frame policy kept from the reference (see rt_host.hpp).
#include "rt_host.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>

#pragma clang fp contract(off)

namespace rtmi {

namespace {

inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator*(Vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float Dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vec3 Cross(Vec3 a, Vec3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline Vec3 Normalize(Vec3 a) { return a * (1.0f / std::sqrt(Dot(a, a))); }
inline float Radians(float deg) { return deg * 0.01745329251994329576923690768489f; }

struct FaceVertex {
    long v = -1, n = -1;
};

// "a", "a/b", "a//c", "a/b/c" with 1-based or negative (relative) indices
bool ParseFaceVertex(const std::string &tok, long nPositions, long nNormals, FaceVertex &out)
{
    long idx[3] = {0, 0, 0};
    int field = 0;
    size_t pos = 0;
    while (field < 3 && pos <= tok.size()) {
        size_t slash = tok.find('/', pos);
        std::string part = tok.substr(pos, slash == std::string::npos ? std::string::npos : slash - pos);
        if (!part.empty()) idx[field] = std::strtol(part.c_str(), nullptr, 10);
        field++;
        if (slash == std::string::npos) break;
        pos = slash + 1;
    }
    if (idx[0] == 0) return false;
    out.v = idx[0] > 0 ? idx[0] - 1 : nPositions + idx[0];
    out.n = idx[2] > 0 ? idx[2] - 1 : (idx[2] < 0 ? nNormals + idx[2] : -1);
    return true;
}

}  // namespace

// ModelLoader.cpp:31-142.  The reference lets tinyobjloader parse and triangulate; positions are
// scaled then translated component-wise (one multiply, one add), normals are copied as stored,
// the bounding box starts at (1024, -1024) and one Mesh record is appended per file.
bool LoadModel(const char *filePath, std::vector<Triangle> &triangles, std::vector<Mesh> &meshes, const Material &material,
               Vec3 translate, Vec3 scale, QuadRule rule)
{
    std::ifstream in(filePath);
    if (!in) return false;

    std::vector<Vec3> positions, normals;
    std::vector<Triangle> out;
    float lo[3] = {1024.0f, 1024.0f, 1024.0f};
    float hi[3] = {-1024.0f, -1024.0f, -1024.0f};
    const float sc[3] = {scale.x, scale.y, scale.z}, tr[3] = {translate.x, translate.y, translate.z};

    std::string line;
    while (std::getline(in, line)) {
        std::istringstream ls(line);
        std::string tag;
        if (!(ls >> tag)) continue;
        if (tag == "v" || tag == "vn") {
            double x = 0.0, y = 0.0, z = 0.0;
            ls >> x >> y >> z;
            (tag == "v" ? positions : normals).push_back(Vec3{(float)x, (float)y, (float)z});
        } else if (tag == "f") {
            std::vector<FaceVertex> fv;
            std::string tok;
            while (ls >> tok) {
                FaceVertex f;
                if (!ParseFaceVertex(tok, (long)positions.size(), (long)normals.size(), f)) break;
                if (f.v < 0 || f.v >= (long)positions.size()) return false;
                fv.push_back(f);
            }
            if (fv.size() < 3) continue;
            std::vector<std::array<int, 3>> order;
            if (fv.size() == 4 && rule == QuadRule::ShorterDiagonal) {
                Vec3 d02 = positions[fv[0].v] - positions[fv[2].v];
                Vec3 d13 = positions[fv[1].v] - positions[fv[3].v];
                if (Dot(d02, d02) < Dot(d13, d13)) order = {{0, 1, 2}, {0, 2, 3}};
                else order = {{0, 1, 3}, {1, 2, 3}};
            } else {
                for (int i = 1; i + 1 < (int)fv.size(); i++) order.push_back({0, i, i + 1});
            }
            for (const auto &o : order) {
                Triangle t;
                std::memset(&t, 0, sizeof t);
                float *pp[3] = {t.p1, t.p2, t.p3}, *nn[3] = {t.n1, t.n2, t.n3};
                for (int c = 0; c < 3; c++) {
                    const FaceVertex &f = fv[o[c]];
                    const Vec3 &p = positions[f.v];
                    const float src[3] = {p.x, p.y, p.z};
                    for (int a = 0; a < 3; a++) {
                        pp[c][a] = src[a] * sc[a] + tr[a];
                        lo[a] = std::min(lo[a], pp[c][a]);
                        hi[a] = std::max(hi[a], pp[c][a]);
                    }
                    if (f.n >= 0 && f.n < (long)normals.size()) {
                        nn[c][0] = normals[f.n].x;
                        nn[c][1] = normals[f.n].y;
                        nn[c][2] = normals[f.n].z;
                    }
                }
                out.push_back(t);
            }
        }
    }

    Mesh mesh;
    std::memset(&mesh, 0, sizeof mesh);
    mesh.start_triangle = (int32_t)triangles.size();
    mesh.num_triangles = (int32_t)out.size();
    for (int a = 0; a < 3; a++) {
        mesh.bounding_point1[a] = lo[a];
        mesh.bounding_point2[a] = hi[a];
    }
    mesh.material = material;
    triangles.insert(triangles.end(), out.begin(), out.end());
    meshes.push_back(mesh);
    return true;
}

// glm::perspectiveFovLH_ZO as published: h = cos(fov/2)/sin(fov/2), w = h*height/width
Mat4 PerspectiveFovLH_ZO(float fov, float width, float height, float zNear, float zFar)
{
    const float h = std::cos(0.5f * fov) / std::sin(0.5f * fov);
    const float w = h * height / width;
    Mat4 m{};
    m[0] = w;
    m[5] = h;
    m[10] = zFar / (zFar - zNear);
    m[11] = 1.0f;
    m[14] = -(zFar * zNear) / (zFar - zNear);
    return m;
}

// glm::lookAtLH as published
Mat4 LookAtLH(Vec3 eye, Vec3 center, Vec3 upDir)
{
    const Vec3 f = Normalize(center - eye);
    const Vec3 s = Normalize(Cross(upDir, f));
    const Vec3 u = Cross(f, s);
    Mat4 m{};
    m[0] = s.x; m[4] = s.y; m[8] = s.z;
    m[1] = u.x; m[5] = u.y; m[9] = u.z;
    m[2] = f.x; m[6] = f.y; m[10] = f.z;
    m[12] = -Dot(s, eye);
    m[13] = -Dot(u, eye);
    m[14] = -Dot(f, eye);
    m[15] = 1.0f;
    return m;
}

// glm's 4x4 inverse as published: 2x2 sub-determinants -> cofactor columns -> scale by 1/det
Mat4 Inverse(const Mat4 &m)
{
    auto e = [&](int c, int r) { return m[c * 4 + r]; };
    auto sub = [&](int c0, int r0, int c1, int r1, int c2, int r2, int c3, int r3) {
        return e(c0, r0) * e(c1, r1) - e(c2, r2) * e(c3, r3);
    };
    const float c00 = sub(2, 2, 3, 3, 3, 2, 2, 3), c02 = sub(1, 2, 3, 3, 3, 2, 1, 3), c03 = sub(1, 2, 2, 3, 2, 2, 1, 3);
    const float c04 = sub(2, 1, 3, 3, 3, 1, 2, 3), c06 = sub(1, 1, 3, 3, 3, 1, 1, 3), c07 = sub(1, 1, 2, 3, 2, 1, 1, 3);
    const float c08 = sub(2, 1, 3, 2, 3, 1, 2, 2), c10 = sub(1, 1, 3, 2, 3, 1, 1, 2), c11 = sub(1, 1, 2, 2, 2, 1, 1, 2);
    const float c12 = sub(2, 0, 3, 3, 3, 0, 2, 3), c14 = sub(1, 0, 3, 3, 3, 0, 1, 3), c15 = sub(1, 0, 2, 3, 2, 0, 1, 3);
    const float c16 = sub(2, 0, 3, 2, 3, 0, 2, 2), c18 = sub(1, 0, 3, 2, 3, 0, 1, 2), c19 = sub(1, 0, 2, 2, 2, 0, 1, 2);
    const float c20 = sub(2, 0, 3, 1, 3, 0, 2, 1), c22 = sub(1, 0, 3, 1, 3, 0, 1, 1), c23 = sub(1, 0, 2, 1, 2, 0, 1, 1);
    const float fac[6][4] = {{c00, c00, c02, c03}, {c04, c04, c06, c07}, {c08, c08, c10, c11},
                             {c12, c12, c14, c15}, {c16, c16, c18, c19}, {c20, c20, c22, c23}};
    const float vec[4][4] = {{e(1, 0), e(0, 0), e(0, 0), e(0, 0)}, {e(1, 1), e(0, 1), e(0, 1), e(0, 1)},
                             {e(1, 2), e(0, 2), e(0, 2), e(0, 2)}, {e(1, 3), e(0, 3), e(0, 3), e(0, 3)}};
    Mat4 inv{};
    for (int i = 0; i < 4; i++) {
        const float sa = (i & 1) ? -1.0f : 1.0f, sb = -sa;
        inv[0 * 4 + i] = (vec[1][i] * fac[0][i] - vec[2][i] * fac[1][i] + vec[3][i] * fac[2][i]) * sa;
        inv[1 * 4 + i] = (vec[0][i] * fac[0][i] - vec[2][i] * fac[3][i] + vec[3][i] * fac[4][i]) * sb;
        inv[2 * 4 + i] = (vec[0][i] * fac[1][i] - vec[1][i] * fac[3][i] + vec[3][i] * fac[5][i]) * sa;
        inv[3 * 4 + i] = (vec[0][i] * fac[2][i] - vec[1][i] * fac[4][i] + vec[2][i] * fac[5][i]) * sb;
    }
    const float d0 = e(0, 0) * inv[0], d1 = e(0, 1) * inv[4], d2 = e(0, 2) * inv[8], d3 = e(0, 3) * inv[12];
    const float oneOverDet = 1.0f / ((d0 + d1) + (d2 + d3));
    for (float &v : inv) v = v * oneOverDet;
    return inv;
}

// CameraFPS.cpp:5-26
CameraFPS::CameraFPS(uint32_t windowWidth, uint32_t windowHeight, double mouseX, double mouseY)
    : lastMouseX(mouseX), lastMouseY(mouseY), width(windowWidth), height(windowHeight)
{
    position = Vec3{0.0f, 1.0f, -1.0f};
    up = Vec3{0.0f, 1.0f, 0.0f};
    RecomputeForward();
    RecomputeView();
    projection = PerspectiveFovLH_ZO(fov, (float)width, (float)height, nearPlane, farPlane);
    inverseProjection = Inverse(projection);
}

// CameraFPS.cpp:15-20 / 48-53
void CameraFPS::RecomputeForward()
{
    forward = Normalize(Vec3{std::cos(Radians(yaw)) * std::cos(Radians(pitch)), std::sin(Radians(pitch)),
                             std::sin(Radians(yaw)) * std::cos(Radians(pitch))});
    right = Cross(forward, up);
}

// CameraFPS.cpp:22, 25 / 85-86
void CameraFPS::RecomputeView()
{
    view = LookAtLH(position, position + forward, up);
    inverseView = Inverse(view);
}

void CameraFPS::SetPose(Vec3 pos, float newYaw, float newPitch)
{
    position = pos;
    yaw = newYaw;
    pitch = newPitch;
    RecomputeForward();
    RecomputeView();
    moved = true;
}

// CameraFPS.cpp:33-88
void CameraFPS::Update(double deltaTime, const CameraInput &in)
{
    moved = false;
    if (in.mouseButton2 && (in.mouseX != lastMouseX || in.mouseY != lastMouseY)) {
        yaw = (float)(yaw + (in.mouseX - lastMouseX) * sensitivity);
        pitch = (float)(pitch + (in.mouseY - lastMouseY) * sensitivity);
        yaw = std::fmod(yaw, 360.0f);
        pitch = std::clamp(pitch, -89.0f, 89.0f);
        RecomputeForward();
        moved = true;
    }
    lastMouseX = in.mouseX;
    lastMouseY = in.mouseY;

    Vec3 velocity{};
    if (in.keyW) { velocity = velocity - forward; moved = true; }
    if (in.keyS) { velocity = velocity + forward; moved = true; }
    if (in.keyD) { velocity = velocity - right; moved = true; }
    if (in.keyA) { velocity = velocity + right; moved = true; }
    velocity.y = 0.0f;
    if (velocity.x != 0.0f || velocity.y != 0.0f || velocity.z != 0.0f) velocity = Normalize(velocity);
    if (in.keySpace) { velocity.y += 1.0f; moved = true; }
    if (in.keyLeftShift) { velocity.y -= 1.0f; moved = true; }

    float speed = 2.0f;
    if (in.keyF) speed = 10.0f;
    if (in.keyC) speed = 0.5f;
    position = position + velocity * speed * (float)deltaTime;

    if (moved) RecomputeView();
}

// main.cpp:116-123, 151-153
void DefaultFrameData(FrameData &fd)
{
    std::memset(&fd, 0, sizeof fd);
    fd.rays_per_pixel = 4;
    fd.max_bounce_limit = 6;
    fd.frame_index = 0;
    const float sun[4] = {-0.4f, -0.4f, -0.4f, 0.0f}, hz[4] = {0.7f, 0.3f, 0.1f, 0.0f};
    const float zn[4] = {0.2f, 0.56f, 0.95f, 0.0f}, gr[4] = {0.9f, 0.9f, 0.9f, 0.0f};
    std::memcpy(fd.sun_light_direction, sun, sizeof sun);
    std::memcpy(fd.sky_color_horizon, hz, sizeof hz);
    std::memcpy(fd.sky_color_zenith, zn, sizeof zn);
    std::memcpy(fd.ground_color, gr, sizeof gr);
    fd.sun_focus = 1.0f;
    fd.sun_intensity = 1.0f;
}

// main.cpp:332-335 (Q key) then 355-363
void FrameAdvance(FrameData &fd, const CameraFPS &camera, bool resetKey)
{
    if (resetKey) fd.frame_index = 0;
    std::memcpy(fd.camera_inverse_projection, camera.inverseProjection.data(), sizeof fd.camera_inverse_projection);
    std::memcpy(fd.camera_inverse_view, camera.inverseView.data(), sizeof fd.camera_inverse_view);
    const float pos[4] = {camera.position.x, camera.position.y, camera.position.z, 0.0f};
    const float dir[4] = {camera.forward.x, camera.forward.y, camera.forward.z, 0.0f};
    std::memcpy(fd.camera_pos, pos, sizeof pos);
    std::memcpy(fd.camera_direction, dir, sizeof dir);
    fd.window[0] = (float)camera.width;
    fd.window[1] = (float)camera.height;
    fd.frame_index++;
    if (camera.moved) fd.frame_index = 1;
}

// main.cpp:161-185
bool BuildReferenceScene(const std::string &meshDir, Scene &scene, QuadRule rule)
{
    scene = Scene{};
    Sphere s;
    std::memset(&s, 0, sizeof s);
    s.center[0] = 1.0f; s.center[1] = 1.0f; s.center[2] = 0.0f;
    s.radius = 0.5f;
    s.material = Material{{1.0f, 1.0f, 1.0f}, 0.0f, 0.1f, 0.0f, 0.0f, 0.0f};
    scene.spheres.push_back(s);
    const Material planeMat{{1.0f, 1.0f, 1.0f}, 0.0f, 0.8f, 0.0f, 0.0f, 0.0f};
    const Material cubeMat{{0.9f, 0.9f, 0.9f}, 0.0f, 0.1f, 0.0f, 0.0f, 0.0f};
    bool ok = LoadModel((meshDir + "/plane.obj").c_str(), scene.triangles, scene.meshes, planeMat, Vec3{0.0f, 0.0f, 0.0f},
                        Vec3{2.0f, 1.0f, 2.0f}, rule);
    ok = LoadModel((meshDir + "/cube.obj").c_str(), scene.triangles, scene.meshes, cubeMat, Vec3{-1.0f, 1.0f, 0.0f},
                   Vec3{1.0f, 1.0f, 1.0f}, rule) && ok;
    return ok;
}

namespace {
// linear -> sRGB of an 8-bit UNORM value, as a B8G8R8A8_SRGB attachment encodes on write
uint8_t EncodeSrgb8(uint8_t lin)
{
    const double c = lin / 255.0;
    const double s = c <= 0.0031308 ? 12.92 * c : 1.055 * std::pow(c, 1.0 / 2.4) - 0.055;
    return (uint8_t)std::lround(std::min(1.0, std::max(0.0, s)) * 255.0);
}
}  // namespace

bool WritePPM(const char *path, const uint8_t *rgba8, uint32_t width, uint32_t height, bool srgb)
{
    FILE *fp = std::fopen(path, "wb");
    if (!fp) return false;
    uint8_t lut[256];
    for (int i = 0; i < 256; i++) lut[i] = srgb ? EncodeSrgb8((uint8_t)i) : (uint8_t)i;
    std::fprintf(fp, "P6\n%u %u\n255\n", width, height);
    std::vector<uint8_t> row(3 * (size_t)width);
    // buffer row 0 is the bottom of the view (flipped viewport, main.cpp:340-347)
    for (uint32_t y = height; y-- > 0;) {
        const uint8_t *src = rgba8 + 4 * (size_t)y * width;
        for (uint32_t x = 0; x < width; x++)
            for (int c = 0; c < 3; c++) row[3 * x + c] = lut[src[4 * x + c]];
        if (std::fwrite(row.data(), 1, row.size(), fp) != row.size()) {
            std::fclose(fp);
            return false;
        }
    }
    return std::fclose(fp) == 0;
}

bool WritePFM(const char *path, const float *accumRgba, uint32_t width, uint32_t height, uint32_t frames)
{
    FILE *fp = std::fopen(path, "wb");
    if (!fp) return false;
    // PFM stores rows bottom-up, which is the accumulation buffer's own order
    std::fprintf(fp, "PF\n%u %u\n-1.0\n", width, height);
    const float div = frames ? (float)frames : 1.0f;
    std::vector<float> row(3 * (size_t)width);
    for (uint32_t y = 0; y < height; y++) {
        const float *src = accumRgba + 4 * (size_t)y * width;
        for (uint32_t x = 0; x < width; x++)
            for (int c = 0; c < 3; c++) row[3 * x + c] = src[4 * x + c] / div;
        if (std::fwrite(row.data(), sizeof(float), row.size(), fp) != row.size()) {
            std::fclose(fp);
            return false;
        }
    }
    return std::fclose(fp) == 0;
}

// OpenEXR, the simplest form the format allows: one part, scan lines, no compression, three 32-bit float
// channels.  Header attributes, a table with the file offset of every scan line, then per scan line {y, byte
// count, the line's B values, its G values, its R values} (channels in alphabetical order).  All little endian.
bool WriteEXR(const char *path, const float *accumRgba, uint32_t width, uint32_t height, uint32_t frames)
{
    if (width == 0 || height == 0 || width > 0x7FFFFFFFu / 12u || height > 0x7FFFFFFFu) return false;
    std::vector<unsigned char> head;
    auto put = [&](const void *p, size_t n) { head.insert(head.end(), (const unsigned char *)p, (const unsigned char *)p + n); };
    auto put_i32 = [&](int32_t v) { put(&v, 4); };
    auto put_f32 = [&](float v) { put(&v, 4); };
    auto put_str = [&](const char *t) { put(t, std::strlen(t) + 1); };
    auto attribute = [&](const char *name, const char *type, int32_t size) {
        put_str(name);
        put_str(type);
        put_i32(size);
    };
    const unsigned char magic[8] = {0x76, 0x2f, 0x31, 0x01, 2, 0, 0, 0};   // 20000630, version 2, no flags
    put(magic, 8);
    attribute("channels", "chlist", 3 * 18 + 1);
    for (const char *ch : {"B", "G", "R"}) {
        put_str(ch);
        put_i32(2);                      // FLOAT
        const unsigned char linear_and_reserved[4] = {0, 0, 0, 0};
        put(linear_and_reserved, 4);
        put_i32(1);                      // x sampling
        put_i32(1);                      // y sampling
    }
    head.push_back(0);
    attribute("compression", "compression", 1);
    head.push_back(0);                   // none
    for (const char *name : {"dataWindow", "displayWindow"}) {
        attribute(name, "box2i", 16);
        put_i32(0);
        put_i32(0);
        put_i32((int32_t)width - 1);
        put_i32((int32_t)height - 1);
    }
    attribute("lineOrder", "lineOrder", 1);
    head.push_back(0);                   // increasing y
    attribute("pixelAspectRatio", "float", 4);
    put_f32(1.0f);
    attribute("screenWindowCenter", "v2f", 8);
    put_f32(0.0f);
    put_f32(0.0f);
    attribute("screenWindowWidth", "float", 4);
    put_f32(1.0f);
    head.push_back(0);                   // end of the header

    FILE *fp = std::fopen(path, "wb");
    if (!fp) return false;
    bool ok = std::fwrite(head.data(), 1, head.size(), fp) == head.size();
    const uint64_t line_bytes = 8u + 12u * (uint64_t)width;
    uint64_t offset = head.size() + 8u * (uint64_t)height;
    for (uint32_t y = 0; ok && y < height; y++, offset += line_bytes) ok = std::fwrite(&offset, 8, 1, fp) == 1;
    const float div = frames ? (float)frames : 1.0f;
    std::vector<float> line(3 * (size_t)width);
    for (uint32_t y = 0; ok && y < height; y++) {
        // EXR's y grows downwards; the accumulation buffer's row 0 is the bottom of the view
        const float *src = accumRgba + 4 * (size_t)(height - 1u - y) * width;
        for (uint32_t x = 0; x < width; x++) {
            line[x] = src[4 * x + 2] / div;
            line[(size_t)width + x] = src[4 * x + 1] / div;
            line[2 * (size_t)width + x] = src[4 * x + 0] / div;
        }
        const int32_t yy = (int32_t)y, n = (int32_t)(12u * width);
        ok = std::fwrite(&yy, 4, 1, fp) == 1 && std::fwrite(&n, 4, 1, fp) == 1 &&
             std::fwrite(line.data(), sizeof(float), line.size(), fp) == line.size();
    }
    return (std::fclose(fp) == 0) && ok;
}

}  // namespace rtmi
