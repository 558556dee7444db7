// rt_files.cpp — scene files and accumulation snapshots.  The reference has neither: its scene is a
// set of literals in main (main.cpp:125-185) and its accumulation image lives only in device memory
// (main.cpp:78-82).  A scene file carries the reference's own records (RayTracingStructs.h:24-60)
// field by field; a snapshot is the accumulation image plus the number of frames summed into it, which
// is all that a later launch needs to continue the sum (Raytracing.comp:462, 471-477).
#include <cerrno>
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "rt_host.hpp"

namespace rtmi {
namespace {

struct File {
    FILE *f = nullptr;
    File(const char *path, const char *mode) : f(std::fopen(path, mode)) {}
    ~File()
    {
        if (f) std::fclose(f);
    }
    bool Close()
    {
        const bool ok = f && std::fclose(f) == 0;
        f = nullptr;
        return ok;
    }
};

// nine significant digits identify a binary32 value uniquely
void PutFloats(FILE *f, const float *v, int n)
{
    for (int i = 0; i < n; i++) std::fprintf(f, " %.9g", (double)v[i]);
}

void PutMaterial(FILE *f, const Material &m)
{
    PutFloats(f, m.color, 3);
    PutFloats(f, &m.light, 1);
    PutFloats(f, &m.smoothness, 1);
    // the transmitting-material extension: flag and index of refraction, written only when set
    if (m.pad1 != 0.0f || m.pad2 != 0.0f) {
        PutFloats(f, &m.pad1, 1);
        PutFloats(f, &m.pad2, 1);
    }
}

// reads n floats from *p, advancing it; false if fewer are there
bool GetFloats(char *&p, float *v, int n)
{
    for (int i = 0; i < n; i++) {
        char *end = nullptr;
        errno = 0;
        const float x = std::strtof(p, &end);
        if (end == p) return false;
        v[i] = x;
        p = end;
    }
    return true;
}

bool GetMaterial(char *&p, Material &m)
{
    std::memset(&m, 0, sizeof m);
    if (!GetFloats(p, m.color, 3) || !GetFloats(p, &m.light, 1) || !GetFloats(p, &m.smoothness, 1)) return false;
    // optionally the two words of the transmitting-material extension (both or neither)
    const char *q = p;
    while (*q == ' ' || *q == '\t' || *q == '\r' || *q == '\n') q++;
    if (*q == '\0') return true;
    return GetFloats(p, &m.pad1, 1) && GetFloats(p, &m.pad2, 1);
}

bool OnlySpaceLeft(const char *p)
{
    while (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n') p++;
    return *p == '\0';
}

constexpr char kSnapshotMagic[8] = {'R', 'T', 'M', 'I', 'A', 'C', 'C', '1'};

}  // namespace

bool SaveScene(const char *path, const Scene &scene)
{
    File out(path, "w");
    if (!out.f) return false;
    std::fprintf(out.f, "rtmi-scene 1\n# sphere cx cy cz radius  r g b light smoothness [transmission ior]\n");
    for (const Sphere &s : scene.spheres) {
        std::fputs("sphere", out.f);
        PutFloats(out.f, s.center, 3);
        PutFloats(out.f, &s.radius, 1);
        PutMaterial(out.f, s.material);
        std::fputc('\n', out.f);
    }
    std::fprintf(out.f, "# tri p1 p2 p3 n1 n2 n3 (xyz each)\n");
    for (const Triangle &t : scene.triangles) {
        std::fputs("tri", out.f);
        const float *v[6] = {t.p1, t.p2, t.p3, t.n1, t.n2, t.n3};
        for (const float *q : v) PutFloats(out.f, q, 3);
        std::fputc('\n', out.f);
    }
    std::fprintf(out.f, "# mesh start_triangle num_triangles  box_lo box_hi  r g b light smoothness [transmission ior]\n");
    for (const Mesh &m : scene.meshes) {
        std::fprintf(out.f, "mesh %" PRId32 " %" PRId32, m.start_triangle, m.num_triangles);
        PutFloats(out.f, m.bounding_point1, 3);
        PutFloats(out.f, m.bounding_point2, 3);
        PutMaterial(out.f, m.material);
        std::fputc('\n', out.f);
    }
    return !std::ferror(out.f) && out.Close();
}

bool LoadScene(const char *path, Scene &scene)
{
    File in(path, "r");
    if (!in.f) return false;
    Scene s;
    char line[1024];
    bool header = false;
    while (std::fgets(line, sizeof line, in.f)) {
        char *p = line;
        while (*p == ' ' || *p == '\t') p++;
        if (*p == '#' || OnlySpaceLeft(p)) continue;
        if (!header) {
            int version = 0;
            if (std::sscanf(p, "rtmi-scene %d", &version) != 1 || version != 1) return false;
            header = true;
            continue;
        }
        if (std::strncmp(p, "sphere", 6) == 0) {
            p += 6;
            Sphere sp;
            std::memset(&sp, 0, sizeof sp);
            if (!GetFloats(p, sp.center, 3) || !GetFloats(p, &sp.radius, 1) || !GetMaterial(p, sp.material) || !OnlySpaceLeft(p))
                return false;
            s.spheres.push_back(sp);
        } else if (std::strncmp(p, "tri", 3) == 0) {
            p += 3;
            Triangle t;
            std::memset(&t, 0, sizeof t);
            float *v[6] = {t.p1, t.p2, t.p3, t.n1, t.n2, t.n3};
            for (float *q : v)
                if (!GetFloats(p, q, 3)) return false;
            if (!OnlySpaceLeft(p)) return false;
            s.triangles.push_back(t);
        } else if (std::strncmp(p, "mesh", 4) == 0) {
            p += 4;
            Mesh m;
            std::memset(&m, 0, sizeof m);
            char *end = nullptr;
            const long start = std::strtol(p, &end, 10);
            if (end == p) return false;
            p = end;
            const long count = std::strtol(p, &end, 10);
            if (end == p) return false;
            p = end;
            if (start < 0 || count < 0 || start > INT32_MAX || count > INT32_MAX) return false;
            m.start_triangle = (int32_t)start;
            m.num_triangles = (int32_t)count;
            if (!GetFloats(p, m.bounding_point1, 3) || !GetFloats(p, m.bounding_point2, 3) || !GetMaterial(p, m.material) ||
                !OnlySpaceLeft(p))
                return false;
            s.meshes.push_back(m);
        } else {
            return false;
        }
    }
    if (!header || std::ferror(in.f)) return false;
    for (const Mesh &m : s.meshes)
        if ((int64_t)m.start_triangle + m.num_triangles > (int64_t)s.triangles.size()) return false;
    scene = std::move(s);
    return true;
}

bool SaveSnapshot(const char *path, uint32_t width, uint32_t height, uint32_t frames, const float *accumRgba)
{
    File out(path, "wb");
    if (!out.f) return false;
    const uint32_t head[4] = {width, height, frames, 0u};
    const size_t n = (size_t)width * height * 4;
    if (std::fwrite(kSnapshotMagic, 1, 8, out.f) != 8 || std::fwrite(head, 4, 4, out.f) != 4 ||
        std::fwrite(accumRgba, sizeof(float), n, out.f) != n)
        return false;
    return out.Close();
}

bool LoadSnapshot(const char *path, Snapshot &snap)
{
    File in(path, "rb");
    if (!in.f) return false;
    char magic[8];
    uint32_t head[4];
    if (std::fread(magic, 1, 8, in.f) != 8 || std::memcmp(magic, kSnapshotMagic, 8) != 0) return false;
    if (std::fread(head, 4, 4, in.f) != 4) return false;
    if (head[0] == 0 || head[1] == 0 || (uint64_t)head[0] * head[1] > 0x7FFFFFFFull) return false;
    const size_t n = (size_t)head[0] * head[1] * 4;
    // the header is not trusted: the file must hold exactly the payload it announces before anything
    // of that size is allocated
    if (std::fseek(in.f, 0, SEEK_END) != 0) return false;
    const long long file_bytes = (long long)std::ftell(in.f);
    if (file_bytes < 0 || (unsigned long long)file_bytes != 24ull + (unsigned long long)n * sizeof(float)) return false;
    if (std::fseek(in.f, 24, SEEK_SET) != 0) return false;
    std::vector<float> data(n);
    if (std::fread(data.data(), sizeof(float), n, in.f) != n) return false;
    if (std::fgetc(in.f) != EOF) return false;  // trailing bytes: not a snapshot of this extent
    snap.width = head[0];
    snap.height = head[1];
    snap.frames = head[2];
    snap.accum = std::move(data);
    return true;
}

}  // namespace rtmi
