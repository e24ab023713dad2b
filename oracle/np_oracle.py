"""np_oracle.py -- second, independent CPU restatement of the radiance hot path of
addrianyy/path-tracer in numpy float32 (TEST INFRASTRUCTURE, like pt_oracle.c).

Written from the Rust text (reference tree at /root/reference; every function cites
file:line), NOT from the C oracle: it exists so that two separately written
restatements can be required to agree bit for bit (tests/test_oracle_twin.py).  The
reference ships no tests or golden vectors and cannot be built here (no rustc), so
parity stays "unpinned"; this agreement plus the KATs is the strongest pin available.

Every numpy float32 ufunc call is one correctly rounded IEEE operation (numpy never
fuses multiply-add), so expression order below IS the reference's operation order.
All rays of one sample index are processed together (arrays over pixels); samples
are accumulated in order, which reproduces the sequential sum of raytracer.rs:88.

Portable definitions shared with pt_oracle.c and the HIP library:
P1 RSQ(x) = 1/sqrt(x); P2 fp64 atan2/asin rounded once to fp32; P3 one xorshift64
stream per (pixel, sample).
"""
import numpy as np

F = np.float32
U64 = np.uint64
T_MIN = F(0.001)                 # scene/mod.rs:25
F32_MAX = np.finfo(np.float32).max   # scene/mod.rs:27
LAMBERTIAN, METAL, DIELECTRIC = 0, 1, 2
_err = dict(over="ignore", invalid="ignore", divide="ignore")


# ---- Vec3 as (..., 3) float32 arrays (math/vec.rs) -----------------------------------

def dot(a, b):                                   # vec.rs:32-36,99-101: (x + y) + z
    p = a * b
    return (p[..., 0] + p[..., 1]) + p[..., 2]


def rsq(x):                                      # P1 (vec.rs:38-44 uses _mm_rsqrt_ps)
    with np.errstate(**_err):
        return F(1.0) / np.sqrt(x)


def normalized(v):                               # vec.rs:95-97
    with np.errstate(**_err):
        return v * rsq(dot(v, v))[..., None]


def sse_min(a, b):                               # _mm_min_ps: a < b ? a : b  (vec.rs:50)
    return np.where(a < b, a, b)


def sse_max(a, b):                               # _mm_max_ps: a > b ? a : b  (vec.rs:51)
    return np.where(a > b, a, b)


def rust_max(a, b):                              # f32::max ignores NaN (aabb.rs:32)
    return np.fmax(a, b)


def rust_min(a, b):                              # f32::min ignores NaN (aabb.rs:33)
    return np.fmin(a, b)


# ---- RNG (rng.rs) ---------------------------------------------------------------------

def rng_raw(state):                              # rng.rs:50-60 (state: uint64 array, updated in place)
    x = state
    x ^= x << U64(13)
    x ^= x >> U64(7)
    x ^= x << U64(17)
    return x


def rng_with_seed(seed):                         # rng.rs:27-35
    s = np.array(seed, dtype=U64, copy=True)
    for _ in range(10):
        rng_raw(s)
    return s


def rng_f32(state):                              # rng.rs:76-80
    v = (rng_raw(state) & U64(0xFFFFFFFF)).astype(np.uint32)
    return (v >> np.uint32(8)).astype(F) / F(16777216.0)


def rng_range(state, lo, hi):                    # rng.rs:83-85
    return rng_f32(state) * (F(hi) - F(lo)) + F(lo)


def path_stream(seed, pixel, sample):            # P3: stream state after with_seed; splitmix64 finaliser, never 0
    with np.errstate(over="ignore"):
        k = (pixel.astype(U64) << U64(32)) | U64(sample)
        z = U64(seed) + (k + U64(1)) * U64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> U64(30))) * U64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> U64(27))) * U64(0x94D049BB133111EB)
        z = z ^ (z >> U64(31))
    return np.where(z == 0, U64(0x9E3779B97F4A7C15), z)


# ---- P2: portable atan2f / asinf (sphere.rs:9-10) --------------------------------------

def _atan_unit(q):                               # q in [0, 1], float64
    big = q > 0.41421356237309504880
    with np.errstate(**_err):
        t = np.where(big, (q - 1.0) / (q + 1.0), q)
    base = np.where(big, 0.78539816339744830962, 0.0)
    z = t * t
    p = np.full_like(q, 1.0 / 45.0)
    for k in range(21, -1, -1):
        p = (1.0 / float(2 * k + 1)) - z * p
    return base + t * p


def _atan2_core(y, x):
    ay, ax = np.abs(y), np.abs(x)
    with np.errstate(**_err):
        a = np.where(ay > ax, 1.57079632679489661923 - _atan_unit(ax / ay), _atan_unit(ay / ax))
    a = np.where((ax == 0.0) & (ay == 0.0), 0.0, a)
    a = np.where(np.signbit(x), 3.14159265358979323846 - a, a)
    a = np.where(np.signbit(y), -a, a)
    return np.where(np.isnan(y) | np.isnan(x), np.nan, a)


def atan2f(y, x):
    return _atan2_core(y.astype(np.float64), x.astype(np.float64)).astype(F)


def asinf(y):
    yd = y.astype(np.float64)
    with np.errstate(**_err):
        c = np.sqrt((1.0 - yd) * (1.0 + yd))
        r = _atan2_core(yd, c)
    return np.where(np.abs(yd) <= 1.0, r, np.nan).astype(F)


# ---- textures (sphere.rs:6-18, picture.rs:17-38) -----------------------------------------

def _trunc_sat_i32(f):                           # Rust `as i32`
    with np.errstate(**_err):
        out = np.trunc(np.nan_to_num(f.astype(np.float64), nan=0.0, posinf=2.0 ** 40, neginf=-2.0 ** 40))
    return np.clip(out, -2147483648.0, 2147483647.0).astype(np.int64)


def sphere_uv(n):
    phi = atan2f(n[..., 2], n[..., 0])
    theta = asinf(n[..., 1])
    pi = F(np.pi)
    with np.errstate(**_err):
        u = F(1.0) - (phi + pi) / (F(2.0) * pi)
        v = (theta + pi / F(2.0)) / pi
    return u, v


def texture_color(rgb8, u, v):
    h, w = rgb8.shape[0], rgb8.shape[1]
    with np.errstate(**_err):
        i = _trunc_sat_i32(u * F(w))
        j = _trunc_sat_i32(((F(1.0) - v) * F(h)) - F(0.001))
    i = np.maximum(np.minimum(i, w - 1), 0)      # value.min(max).max(min), picture.rs:24-26
    j = np.maximum(np.minimum(j, h - 1), 0)
    return rgb8[j, i].astype(F) / F(255.0)


# ---- BVH build (scene/bvh.rs:13-91) ------------------------------------------------------

class Node:
    __slots__ = ("bmin", "bmax", "prim", "left", "right")


def _volume(bmin, bmax):                         # aabb.rs:45-49
    e = bmax - bmin
    return np.abs(e[..., 0] * e[..., 1] * e[..., 2])


def _construct(prims, bmins, bmaxs):
    node = Node()
    n = len(prims)
    if n == 1:                                   # bvh.rs:20-24
        node.prim, node.bmin, node.bmax, node.left, node.right = int(prims[0]), bmins[0], bmaxs[0], None, None
        return node
    wmin, wmax = bmins.min(axis=0), bmaxs.max(axis=0)          # bvh.rs:35
    ext = wmax - wmin
    axis = 0                                     # bvh.rs:37-49
    for k in (1, 2):
        if ext[k] > ext[axis]:
            axis = k
    centers = (bmins[:, axis] + bmaxs[:, axis]) * F(0.5)       # aabb.rs:21-23
    order = np.argsort(centers, kind="stable")   # bvh.rs:51-56 (ties: stable; see pt_oracle.c)
    prims, bmins, bmaxs = prims[order], bmins[order], bmaxs[order]
    lmin, lmax = np.minimum.accumulate(bmins, axis=0), np.maximum.accumulate(bmaxs, axis=0)
    rmin = np.minimum.accumulate(bmins[::-1], axis=0)[::-1]
    rmax = np.maximum.accumulate(bmaxs[::-1], axis=0)[::-1]
    i = np.arange(1, n)
    wvol = _volume(wmin, wmax)
    cost = (_volume(lmin[i - 1], lmax[i - 1]) / wvol) * i.astype(F) + (_volume(rmin[i], rmax[i]) / wvol) * (n - i).astype(F)
    best, lowest = 0, F32_MAX                    # bvh.rs:58-80: strict '<' keeps the first minimum
    for k in range(n - 1):
        if cost[k] < lowest:
            best, lowest = k + 1, cost[k]
    assert best > 0
    node.left = _construct(prims[:best], bmins[:best], bmaxs[:best])
    node.right = _construct(prims[best:], bmins[best:], bmaxs[best:])
    node.bmin = sse_min(node.left.bmin, node.right.bmin)       # bvh.rs:86
    node.bmax = sse_max(node.left.bmax, node.right.bmax)
    node.prim = -1
    return node


def build_bvh(centers, radii):
    """-> pre-order arrays (bmin, bmax, skip, prim) of the BvhNode tree"""
    centers, radii = np.asarray(centers, F), np.asarray(radii, F)
    bmins, bmaxs = centers - radii[:, None], centers + radii[:, None]   # sphere.rs:71-76
    root = _construct(np.arange(len(radii)), bmins, bmaxs)
    out = []

    def flat(nd):
        me = len(out)
        out.append([nd.bmin, nd.bmax, 0, nd.prim])
        if nd.prim < 0:
            flat(nd.left)
            flat(nd.right)
        out[me][2] = len(out)
    import sys
    sys.setrecursionlimit(max(10000, sys.getrecursionlimit()))
    flat(root)
    return (np.array([o[0] for o in out], F), np.array([o[1] for o in out], F),
            np.array([o[2] for o in out], np.int64), np.array([o[3] for o in out], np.int64))


# ---- scene, traversal, materials ----------------------------------------------------------

class Scene:
    def __init__(self, spheres, materials, textures):
        """spheres: structured array (cx, cy, cz, radius, material); materials: list of
        (kind, (r, g, b), param, texture); textures: list of (H, W, 3) uint8 arrays"""
        self.c = np.stack([spheres["cx"], spheres["cy"], spheres["cz"]], axis=1).astype(F)
        self.r = spheres["radius"].astype(F)
        self.mat = spheres["material"].astype(np.int64)
        self.kind = np.array([m[0] for m in materials], np.int64)
        self.albedo = np.array([m[1] for m in materials], F)
        self.param = np.array([m[2] for m in materials], F)
        self.tex = np.array([m[3] for m in materials], np.int64)
        self.textures = textures
        self.bmin, self.bmax, self.skip, self.prim = build_bvh(self.c, self.r)

    def trace(self, o, d):
        """Scene::trace (scene/mod.rs:24-34) + BvhNode::trace (bvh.rs:106-127) for N rays:
        pre-order walk, left before right, max_t = closest so far.  -> (t, prim)"""
        n = o.shape[0]
        with np.errstate(**_err):
            inv = F(1.0) / d                                    # scene/mod.rs:31
        closest = np.full(n, F32_MAX, F)
        best = np.full(n, -1, np.int64)
        idx = np.zeros(n, np.int64)
        n_nodes = len(self.skip)
        a = dot(d, d)                                           # sphere.rs:49
        alive = idx < n_nodes
        while alive.any():
            r = np.nonzero(alive)[0]
            i = idx[r]
            leaf = self.prim[i] >= 0
            # Split: AABB::intersect (aabb.rs:25-36)
            rs, ns = r[~leaf], i[~leaf]
            if len(rs):
                with np.errstate(**_err):
                    v0 = (self.bmin[ns] - o[rs]) * inv[rs]
                    v1 = (self.bmax[ns] - o[rs]) * inv[rs]
                mn, mx = sse_min(v0, v1), sse_max(v0, v1)
                tmin = rust_max(T_MIN, rust_max(mn[:, 0], rust_max(mn[:, 1], mn[:, 2])))
                tmax = rust_min(closest[rs], rust_min(mx[:, 0], rust_min(mx[:, 1], mx[:, 2])))
                idx[rs] = np.where(tmin < tmax, ns + 1, self.skip[ns])
            # Leaf: Sphere::trace (sphere.rs:46-69)
            rl, nl = r[leaf], i[leaf]
            if len(rl):
                p = self.prim[nl]
                with np.errstate(**_err):
                    oc = o[rl] - self.c[p]
                    b = dot(oc, d[rl])
                    c = dot(oc, oc) - self.r[p] * self.r[p]
                    disc = b * b - a[rl] * c
                    sq = np.sqrt(np.where(disc > 0, disc, F(0.0)))
                    near = (-b - sq) / a[rl]
                    far = (-b + sq) / a[rl]
                pos = disc > 0
                ok_near = pos & (near < closest[rl]) & (near > T_MIN)
                ok_far = pos & ~ok_near & (far < closest[rl]) & (far > T_MIN)
                t = np.where(ok_near, near, far)
                hit = ok_near | ok_far
                closest[rl] = np.where(hit, t, closest[rl])
                best[rl] = np.where(hit, p, best[rl])
                idx[rl] = nl + 1
            alive = idx < n_nodes
        return closest, best

    def scatter(self, o, d, t, prim, rng):
        """Sphere::record (sphere.rs:37-42) + Material::scatter for the rays that hit.
        -> (scattered mask, attenuation multiplier, new origin, new direction)"""
        with np.errstate(**_err):
            p = o + d * t[:, None]                              # ray.rs:26-28
            nrm = normalized(p - self.c[prim])
        return self.scatter_at(d, p, nrm, self.mat[prim], rng)

    def scatter_at(self, d, p, nrm, m, rng):
        """Material::scatter (material/*.rs) for hits at p with normal nrm and material indices m."""
        t = m
        kind = self.kind[m]
        n = len(t)
        ok = np.ones(n, bool)
        att = np.ones((n, 3), F)
        nd = np.zeros((n, 3), F)

        def unit(state):                                        # math/mod.rs:13-19
            x = rng_range(state, -1.0, 1.0)
            y = rng_range(state, -1.0, 1.0)
            z = rng_range(state, -1.0, 1.0)
            return normalized(np.stack([x, y, z], axis=1))

        def reflect(dd, nn):                                    # math/mod.rs:21-23
            return dd - ((nn * dot(dd, nn)[:, None]) * F(2.0))

        L = np.nonzero(kind == LAMBERTIAN)[0]                   # lambertian.rs:32-45
        if len(L):
            st = rng[L]
            with np.errstate(**_err):
                target = (p[L] + nrm[L]) + unit(st)
            rng[L] = st
            col = self.albedo[m[L]].copy()
            tex = self.tex[m[L]]
            for ti in np.unique(tex[tex >= 0]):
                sel = tex == ti
                u, v = sphere_uv(nrm[L][sel])
                col[sel] = texture_color(self.textures[ti], u, v)
            att[L] = col
            with np.errstate(**_err):
                nd[L] = normalized(target - p[L])               # ray.rs:19-24
        M = np.nonzero(kind == METAL)[0]                        # metal.rs:22-32
        if len(M):
            with np.errstate(**_err):
                refl = reflect(d[M], nrm[M])
                good = dot(refl, nrm[M]) > 0
            G = M[good]
            st = rng[G]
            with np.errstate(**_err):
                fuzz = unit(st) * self.param[m[G]][:, None]
                nd[G] = normalized(refl[good] + fuzz)
            rng[G] = st
            att[G] = self.albedo[m[G]]
            ok[M[~good]] = False
        D = np.nonzero(kind == DIELECTRIC)[0]                   # dielectric.rs:20-50
        if len(D):
            ri = self.param[m[D]]
            dd, nn = d[D], nrm[D]
            with np.errstate(**_err):
                dt = dot(dd, nn)
                front = dt > 0
                outward = np.where(front[:, None], nn * F(-1.0), nn)
                k = np.where(front, ri, F(1.0) / ri)
                cosine = np.where(front, ri * dt, -dt)
                refl = reflect(dd, nn)
                dto = dot(dd, outward)                          # math/mod.rs:25-34
                disc = F(1.0) - (k * k) * (F(1.0) - dto * dto)
                can = disc > 0
                refr = ((dd - outward * dto[:, None]) * k[:, None]) - (outward * np.sqrt(np.where(can, disc, F(0.0)))[:, None])
                r0 = (F(1.0) - ri) / (F(1.0) + ri)              # math/mod.rs:36-45
                r0 = r0 * r0
                v = r0 + (F(1.0) - r0) * (F(1.0) - cosine)
                v2 = v * v
                prob = (v2 * v2) * v
            C = D[can]
            st = rng[C]
            rnd = rng_f32(st)                                   # one draw only when refraction is possible
            rng[C] = st
            choose_refl = np.ones(len(D), bool)
            choose_refl[can] = rnd < prob[can]
            with np.errstate(**_err):
                nd[D] = normalized(np.where(choose_refl[:, None], refl, refr))
        return ok, att, p, nd


def camera_new(eye, target, up, fov, width, height):            # camera.rs:14-34
    eye, target, up = np.asarray(eye, F), np.asarray(target, F), np.asarray(up, F)
    aspect = F(width) / F(height)
    half_h = np.tan(F(fov) * (F(np.pi) / F(180.0)) / F(2.0)).astype(F)
    half_w = half_h * aspect

    def cross(a, b):                                            # vec.rs:103-112
        return np.array([a[1] * b[2] - a[2] * b[1], -(a[0] * b[2] - a[2] * b[0]), a[0] * b[1] - a[1] * b[0]], F)
    w = normalized(eye - target)
    u = normalized(cross(up, w))
    v = cross(w, u)
    return dict(llc=((eye - u * half_w) - v * half_h) - w, origin=eye, horizontal=(u * F(2.0)) * half_w,
                vertical=(v * F(2.0)) * half_h, width=int(width), height=int(height))


def sample_rays(cam, S, s):                                     # raytracer.rs:77-85 + camera.rs:37-44
    W, H = cam["width"], cam["height"]
    py, px = np.divmod(np.arange(W * H), W)                     # raytracer.rs:104-106
    sx, sy = divmod(s, S)
    x = px.astype(F) + (F(sx) / F(S - 1))
    u = x / F(W)
    y = py.astype(F) + (F(sy) / F(S - 1))
    v = F(1.0) - (y / F(H))
    d = ((cam["llc"] + cam["horizontal"] * u[:, None]) + cam["vertical"] * v[:, None]) - cam["origin"]
    o = np.broadcast_to(cam["origin"], d.shape).astype(F).copy()
    return o, d.astype(F)


def render(scene, cam, S, seed, sample_range=None):
    """-> (radiance sums (H, W, 3) float32, primary hit ids of sample 0 (H, W) int32)"""
    W, H = cam["width"], cam["height"]
    npx = W * H
    s0, s1 = sample_range if sample_range is not None else (0, S * S)
    total = np.zeros((npx, 3), F)
    pixel = np.arange(npx, dtype=np.uint32)
    for s in range(s0, s1):
        o, d = sample_rays(cam, S, s)
        rng = path_stream(seed, pixel, s).copy()
        att = np.ones((npx, 3), F)                              # raytracer.rs:50
        dead = np.zeros(npx, bool)
        active = np.arange(npx)
        for _ in range(5):                                      # MAX_TRACES, raytracer.rs:48-65
            if not len(active):
                break
            t, prim = scene.trace(o[active], d[active])
            hit = prim >= 0
            hi = active[hit]
            st = rng[hi]
            ok, a, p, nd = scene.scatter(o[hi], d[hi], t[hit], prim[hit], st)
            rng[hi] = st
            att[hi[ok]] = att[hi[ok]] * a[ok]                   # raytracer.rs:57
            o[hi[ok]], d[hi[ok]] = p[ok], nd[ok]
            dead[hi[~ok]] = True                                # absorbed: trace_ray returns 0 (raytracer.rs:60)
            active = hi[ok]
        with np.errstate(**_err):
            tt = F(0.5) * (d[:, 1] + F(1.0))                    # raytracer.rs:67-70
            sky = (np.ones(3, F) * (F(1.0) - tt)[:, None]) + (np.array([0.5, 0.7, 1.0], F) * tt[:, None])
            color = np.where(dead[:, None], F(0.0), sky * att)
        total = total + color                                   # raytracer.rs:88
    ids = scene.trace(*sample_rays(cam, S, 0))[1].astype(np.int32).reshape(H, W)
    return total.reshape(H, W, 3), ids
