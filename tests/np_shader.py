"""A second, independent restatement of Raytracing.comp, vectorised over pixels in numpy float32.

TEST INFRASTRUCTURE ONLY.  It exists to cross-check the C oracle: it was written from the shader text
(line references below), shares no code or structure with oracle/rt_oracle.c (arrays and masks instead
of per-pixel scalar code), and follows the same arithmetic contract: every operation is one binary32
operation in the order the shader writes it (numpy float32 arithmetic rounds each operation
separately), normalize = v * (1 / sqrt(dot)), dot and matrix products summed left to right, GLSL's
min / max operand rule.  pow() is the contract's fixed polynomial and is taken from the oracle library
(orc_pow): it is not part of the shader's logic.

Shader(..., fused=True) restates the fused arithmetic profile instead: the same sequences with a fused
multiply-add at the profile's list of sites.  numpy has no fma; fma() below computes one exactly: the
product of two binary32 values is exact in binary64, the binary64 sum is corrected to round-to-odd
with the exact error of the addition (TwoSum), and rounding a round-to-odd binary64 value to binary32 gives
the correctly rounded result of the exact a * b + c."""
import numpy as np

F = np.float32
FLT_MAX = F(3.402823466e+38)
MISS = np.uint32(0xFFFFFFFF)


def _v(a):
    return np.asarray(a, dtype=F)


def fma(a, b, c):
    """Correctly rounded binary32 a * b + c, elementwise."""
    a, b, c = np.broadcast_arrays(np.asarray(a, F), np.asarray(b, F), np.asarray(c, F))
    with np.errstate(all="ignore"):
        p = a.astype(np.float64) * b.astype(np.float64)        # exact
        c64 = c.astype(np.float64)
        s = p + c64
        bb = s - p
        err = (p - (s - bb)) + (c64 - bb)                       # exact error of the sum (TwoSum)
        inexact = np.isfinite(s) & (err != 0.0)
        even = (s.view(np.int64) & 1) == 0
        toward = np.where(err > 0.0, np.inf, -np.inf)
        s = np.where(inexact & even, np.nextafter(s, toward), s)   # round to odd
        return s.astype(F)


class Ops:
    """The operations whose rounding differs between the two arithmetic profiles."""

    def __init__(self, fused):
        self.fused = bool(fused)

    def mul_add(self, a, b, c):
        return fma(a, b, c) if self.fused else a * b + c

    def dot3(self, a, b):
        if self.fused:
            return fma(a[2], b[2], fma(a[1], b[1], a[0] * b[0]))
        return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]

    def cross3(self, a, b):
        if self.fused:
            return (fma(a[1], b[2], -(b[1] * a[2])), fma(a[2], b[0], -(b[2] * a[0])), fma(a[0], b[1], -(b[0] * a[1])))
        return (a[1] * b[2] - b[1] * a[2], a[2] * b[0] - b[2] * a[0], a[0] * b[1] - b[0] * a[1])

    def normalize3(self, a):
        inv = F(1.0) / np.sqrt(self.dot3(a, a))
        return (a[0] * inv, a[1] * inv, a[2] * inv)

    def mix(self, a, b, t):
        if self.fused:
            return fma(b, t, a * (F(1.0) - t))
        return a * (F(1.0) - t) + b * t

    def row4(self, m, v):
        """m[0] v[0] + m[1] v[1] + m[2] v[2] + m[3] v[3], summed left to right"""
        if self.fused:
            return fma(m[3], v[3], fma(m[2], v[2], fma(m[1], v[1], m[0] * v[0])))
        return m[0] * v[0] + m[1] * v[1] + m[2] * v[2] + m[3] * v[3]


def gmin(x, y):
    return np.where(y < x, y, x)


def gmax(x, y):
    return np.where(x < y, y, x)


def smoothstep(e0, e1, x):
    t = gmin(gmax((x - F(e0)) / (F(e1) - F(e0)), F(0.0)), F(1.0))
    return t * t * (F(3.0) - F(2.0) * t)


def random_value(state):
    """Raytracing.comp:106-112; state is a uint32 array, updated in place."""
    with np.errstate(over="ignore"):
        state *= np.uint32(747796405)
        state += np.uint32(2891336453)
        result = ((state >> ((state >> np.uint32(28)) + np.uint32(4))) ^ state) * np.uint32(277803737)
    result = (result >> np.uint32(22)) ^ result
    return result.astype(F) * F(2.0 ** -32)      # the literal 4294967295.0 is 2^32 in binary32


class Shader:
    def __init__(self, fd, spheres, triangles, meshes, pow_fn, fused=False):
        """fd: the oracle's FrameData ctypes struct; spheres / triangles / meshes: numpy record arrays
        with the reference's layouts; pow_fn(x, y) -> float32 array (of the same profile)."""
        self.fd, self.sph, self.tri, self.msh, self.pow = fd, spheres, triangles, meshes, pow_fn
        self.ops = Ops(fused)
        self.inv_proj = _v(list(fd.inv_proj)).reshape(4, 4)   # [column][row]
        self.inv_view = _v(list(fd.inv_view)).reshape(4, 4)

    # Raytracing.comp:338-388 (with :130-161, :286-309, :311-336)
    def closest_hit(self, o, d):
        n = o[0].shape[0]
        best_t = np.full(n, FLT_MAX, dtype=F)
        did = np.zeros(n, dtype=bool)
        pos = [np.zeros(n, F) for _ in range(3)]
        nrm = [np.zeros(n, F) for _ in range(3)]
        mat = np.zeros(n, dtype=np.int64)          # row of the material table below
        prim = np.full(n, MISS, dtype=np.uint32)
        ns, nm = int(self.fd.sphere_count), int(self.fd.mesh_count)
        ops = self.ops
        dot3, cross3, normalize3 = ops.dot3, ops.cross3, ops.normalize3
        for i in range(ns):
            c, r = self.sph["center"][i], self.sph["radius"][i]
            oc = (o[0] - c[0], o[1] - c[1], o[2] - c[2])
            a = dot3(d, d)
            b = F(2.0) * dot3(oc, d)
            cc = dot3(oc, oc) - r * r
            disc = fma(b, b, -(F(4.0) * a * cc)) if ops.fused else b * b - F(4.0) * a * cc
            dst = (-b - np.sqrt(disc)) / (F(2.0) * a)
            hit = (disc >= F(0.0)) & (dst > F(0.0001))
            hp = tuple(ops.mul_add(d[k], dst, o[k]) if ops.fused else o[k] + d[k] * dst for k in range(3))
            hn = normalize3((hp[0] - c[0], hp[1] - c[1], hp[2] - c[2]))
            take = hit & (dst < best_t)
            best_t = np.where(take, dst, best_t)
            for k in range(3):
                pos[k] = np.where(take, hp[k], pos[k])
                nrm[k] = np.where(take, hn[k], nrm[k])
            mat = np.where(take, i, mat)
            prim = np.where(take, np.uint32(i), prim)
            did |= take
        for m in range(nm):
            p1, p2 = self.msh["bounding_point1"][m], self.msh["bounding_point2"][m]
            frac = (F(1.0) / d[0], F(1.0) / d[1], F(1.0) / d[2])
            t1 = (p1[0] - o[0]) * frac[0]
            t2 = (p2[0] - o[0]) * frac[0]
            t3 = (p1[1] - o[1]) * frac[1]
            t4 = (p2[1] - o[1]) * frac[1]
            t5 = (p1[2] - o[2]) * frac[2]
            t6 = (p2[2] - o[2]) * frac[2]
            tmin = gmax(gmax(gmin(t1, t2), gmin(t3, t4)), gmin(t5, t6))
            tmax = gmin(gmin(gmax(t1, t2), gmax(t3, t4)), gmax(t5, t6))
            inside = ~(tmax < F(0.0)) & ~(tmin > tmax)
            start, count = int(self.msh["start_triangle"][m]), int(self.msh["num_triangles"][m])
            for j in range(start, start + count):
                t = self.tri[j]
                A, B, C = t["p1"][:3], t["p2"][:3], t["p3"][:3]
                ab = (B[0] - A[0], B[1] - A[1], B[2] - A[2])
                ac = (C[0] - A[0], C[1] - A[1], C[2] - A[2])
                nn = cross3(ab, ac)
                ao = (o[0] - A[0], o[1] - A[1], o[2] - A[2])
                dao = cross3(ao, d)
                det = -dot3(d, nn)
                inv_det = F(1.0) / det
                dst = dot3(ao, nn) * inv_det
                u = dot3(ac, dao) * inv_det
                v = -dot3(ab, dao) * inv_det
                w = F(1.0) - u - v
                hit = (det >= F(1e-6)) & (dst >= F(0.0)) & (u >= F(0.0)) & (v >= F(0.0)) & (w >= F(0.0))
                hp = tuple(ops.mul_add(d[k], dst, o[k]) if ops.fused else o[k] + d[k] * dst for k in range(3))
                n1, n2, n3 = t["n1"][:3], t["n2"][:3], t["n3"][:3]
                if ops.fused:
                    hn = normalize3(tuple(fma(n3[k], v, fma(n2[k], u, n1[k] * w)) for k in range(3)))
                else:
                    hn = normalize3(tuple(n1[k] * w + n2[k] * u + n3[k] * v for k in range(3)))
                take = inside & hit & (dst < best_t)
                best_t = np.where(take, dst, best_t)
                for k in range(3):
                    pos[k] = np.where(take, hp[k], pos[k])
                    nrm[k] = np.where(take, hn[k], nrm[k])
                mat = np.where(take, len(self.sph) + m, mat)
                prim = np.where(take, np.uint32(ns + j), prim)
                did |= take
        return did, pos, nrm, mat, prim

    # Raytracing.comp:390-406
    def environment(self, d):
        fd = self.fd
        horizon, zenith, ground = _v(list(fd.sky_horizon)[:3]), _v(list(fd.sky_zenith)[:3]), _v(list(fd.ground)[:3])
        s = _v(list(fd.sun_dir))
        sun_dir = (s[0], s[2], s[1])                       # .xzy
        grad_t = self.pow(smoothstep(0.0, 0.4, d[1]), np.full_like(d[1], F(0.35)))
        mix, dot3 = self.ops.mix, self.ops.dot3
        grad = [mix(horizon[k], zenith[k], grad_t) for k in range(3)]
        sun = self.pow(gmax(F(0.0), dot3(d, (-sun_dir[0], -sun_dir[1], -sun_dir[2]))),
                       np.full_like(d[1], F(fd.sun_focus))) * F(fd.sun_intensity)
        g2s = smoothstep(-0.01, 0.0, d[1])
        mask = (g2s >= F(1.0)).astype(F)
        return [mix(ground[k], grad[k], g2s) + sun * mask for k in range(3)]

    # Raytracing.comp:408-439
    def trace(self, o, d, rng, first_prim=None):
        n = o[0].shape[0]
        light = [np.zeros(n, F) for _ in range(3)]
        color = [np.ones(n, F) for _ in range(3)]
        o, d = [x.copy() for x in o], [x.copy() for x in d]
        alive = np.ones(n, dtype=bool)
        mats = np.concatenate([self.sph["material"], self.msh["material"]]) if len(self.msh) else self.sph["material"]
        ops = self.ops
        dot3, normalize3, mix = ops.dot3, ops.normalize3, ops.mix
        for i in range(int(self.fd.max_bounce)):
            did, pos, nrm, mat, prim = self.closest_hit(o, d)
            if i == 0 and first_prim is not None:
                first_prim[:] = prim
            hit = alive & did
            miss = alive & ~did
            # miss branch (:432-436)
            env = self.environment(d)
            for k in range(3):
                light[k] = np.where(miss, ops.mul_add(env[k], color[k], light[k]) if ops.fused else light[k] + env[k] * color[k], light[k])
            alive = hit
            if not alive.any():
                break
            # hit branch (:417-431); the random numbers are drawn only by lanes that hit
            sub = rng[hit]
            rx = random_value(sub) * F(2.0) - F(1.0)
            ry = random_value(sub) * F(2.0) - F(1.0)
            rz = random_value(sub) * F(2.0) - F(1.0)
            rng[hit] = sub
            r = [np.zeros(n, F) for _ in range(3)]
            r[0][hit], r[1][hit], r[2][hit] = rx, ry, rz
            diffuse = normalize3((nrm[0] + r[0], nrm[1] + r[1], nrm[2] + r[2]))
            kk = F(2.0) * dot3(nrm, d)
            if ops.fused:
                spec = tuple(fma(-nrm[k], kk, d[k]) for k in range(3))
            else:
                spec = (d[0] - nrm[0] * kk, d[1] - nrm[1] * kk, d[2] - nrm[2] * kk)
            m = mats[mat]
            smooth, emit = m["smoothness"].astype(F), m["light"].astype(F)
            for k in range(3):
                nd = mix(diffuse[k], spec[k], smooth)
                o[k] = np.where(hit, pos[k], o[k])
                d[k] = np.where(hit, nd, d[k])
                emitted = F(1.0) * emit
                light[k] = np.where(hit, ops.mul_add(emitted, color[k], light[k]) if ops.fused else light[k] + emitted * color[k], light[k])
                color[k] = np.where(hit, color[k] * m["color"][:, k].astype(F), color[k])
        return light

    # Raytracing.comp:441-480 for one frame; accum is [H, W, 4] float32, returns the new one
    def frame(self, accum, frame_index):
        fd = self.fd
        W, H = int(fd.window[0]), int(fd.window[1])
        ys, xs = np.meshgrid(np.arange(H, dtype=np.uint32), np.arange(W, dtype=np.uint32), indexing="ij")
        x, y = xs.reshape(-1), ys.reshape(-1)
        width, height = F(fd.window[0]), F(fd.window[1])
        cx = x.astype(F) / width * F(2.0) - F(1.0)
        cy = y.astype(F) / height * F(2.0) - F(1.0)
        ip, iv = self.inv_proj, self.inv_view
        one, zero = np.ones_like(cx), np.zeros_like(cx)
        row4 = self.ops.row4
        tgt = [row4([ip[0][r], ip[1][r], ip[2][r], ip[3][r]], [cx, cy, one, one]) for r in range(4)]
        nd = self.ops.normalize3((tgt[0] / tgt[3], tgt[1] / tgt[3], tgt[2] / tgt[3]))
        d = [row4([iv[0][r], iv[1][r], iv[2][r], iv[3][r]], [nd[0], nd[1], nd[2], zero]) for r in range(3)]
        cam = _v(list(fd.cam_pos))
        o = [np.full(x.shape, cam[k], dtype=F) for k in range(3)]
        with np.errstate(over="ignore"):
            rng = (x.astype(F) + width * y.astype(F)).astype(np.uint32) + np.uint32(frame_index) * np.uint32(719393)
        total = [np.zeros(x.shape, F) for _ in range(3)]
        first_prim = np.full(x.shape, MISS, dtype=np.uint32)
        for k in range(int(fd.rays_per_pixel)):
            with np.errstate(over="ignore"):
                rng += np.uint32(k)
            light = self.trace(o, d, rng, first_prim if k == 0 else None)
            for c in range(3):
                total[c] = total[c] + light[c]
        rpp = F(fd.rays_per_pixel)
        total = [t / rpp for t in total]
        out = accum.reshape(-1, 4).copy()
        for c in range(3):
            out[:, c] = total[c] if frame_index == 1 else out[:, c] + total[c]
        out[:, 3] = F(1.0)
        return out.reshape(H, W, 4), first_prim.reshape(H, W), rng.reshape(H, W)
