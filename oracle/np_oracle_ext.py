"""np_oracle_ext.py -- numpy-float32 twin of the EXTENSION (rects, Translate(RotateY(..)) instances, a diffuse
light, trace_ray's constants as parameters; include/pt_amd.h).  TEST INFRASTRUCTURE.

The extension is not in the reference, so there is nothing to restate from Rust here: this file is written from
the specification in DESIGN.md section 10, separately from oracle/pt_oracle.c's version, so that the two can be
required to agree bit for bit (tests/test_oracle_twin.py).  It builds on np_oracle.py for everything the
reference does define (Vec3 arithmetic, RNG, materials, the BVH algorithm of bvh.rs:13-91).
"""
import numpy as np

import np_oracle as npo
from np_oracle import F, F32_MAX, T_MIN, _err, dot, normalized

DIFFUSE_LIGHT = 3
SPHERE, RECT = 0, 1


def _rot_to_object(s, c, p):                     # rotation about y by -theta
    return np.stack([c * p[..., 0] - s * p[..., 2], p[..., 1], s * p[..., 0] + c * p[..., 2]], axis=-1)


def _rot_to_world(s, c, p):                      # rotation about y by +theta
    return np.stack([c * p[..., 0] + s * p[..., 2], p[..., 1], c * p[..., 2] - s * p[..., 0]], axis=-1)


class ExtScene(npo.Scene):
    def __init__(self, prims, materials, textures, transforms):
        """prims: list of dicts (kind, material, transform, axis, flip, v[5]); materials / textures as for
        np_oracle.Scene; transforms: list of (offset(3), sin, cos)."""
        n = len(prims)
        self.pkind = np.array([p["kind"] for p in prims], np.int64)
        self.mat = np.array([p["material"] for p in prims], np.int64)
        self.xf = np.array([p["transform"] for p in prims], np.int64)
        self.axis = np.array([p["axis"] for p in prims], np.int64)
        self.flip = np.array([p["flip"] for p in prims], np.int64)
        self.v = np.array([p["v"] for p in prims], F).reshape(n, 5)
        self.c = self.v[:, :3].copy()                                      # a sphere's centre
        self.r = self.v[:, 3].copy()
        self.kind = np.array([m[0] for m in materials], np.int64)
        self.albedo = np.array([m[1] for m in materials], F)
        self.param = np.array([m[2] for m in materials], F)
        self.tex = np.array([m[3] for m in materials], np.int64)
        self.textures = textures
        self.off = np.array([t[0] for t in transforms], F).reshape(len(transforms), 3)
        self.sin = np.array([t[1] for t in transforms], F)
        self.cos = np.array([t[2] for t in transforms], F)
        self.ia = np.where(self.axis == 0, 1, 0)                           # the rect's two in-plane axes
        self.ib = np.where(self.axis == 2, 1, 2)
        bmins, bmaxs = np.zeros((n, 3), F), np.zeros((n, 3), F)
        for i in range(n):
            bmins[i], bmaxs[i] = self._bbox(i)
        root = npo._construct(np.arange(n), bmins, bmaxs)
        out = []

        def flat(nd):
            me = len(out)
            out.append([nd.bmin, nd.bmax, 0, nd.prim])
            if nd.prim < 0:
                flat(nd.left)
                flat(nd.right)
            out[me][2] = len(out)
        flat(root)
        self.bmin, self.bmax = np.array([o[0] for o in out], F), np.array([o[1] for o in out], F)
        self.skip, self.prim = np.array([o[2] for o in out], np.int64), np.array([o[3] for o in out], np.int64)
        self.integrator = dict(max_bounces=5, background=None, rr_start=-1)

    def _bbox(self, i):
        if self.pkind[i] == SPHERE:
            return self.c[i] - self.r[i], self.c[i] + self.r[i]            # sphere.rs:71-76
        lo, hi = np.zeros(3, F), np.zeros(3, F)
        ax, ia, ib = self.axis[i], self.ia[i], self.ib[i]
        lo[ax], hi[ax] = self.v[i, 4] - F(0.0001), self.v[i, 4] + F(0.0001)
        lo[ia], hi[ia] = self.v[i, 0], self.v[i, 1]
        lo[ib], hi[ib] = self.v[i, 2], self.v[i, 3]
        if self.xf[i] < 0:
            return lo, hi
        s, c, off = self.sin[self.xf[i]], self.cos[self.xf[i]], self.off[self.xf[i]]
        corners = np.array([[hi[0] if k & 1 else lo[0], hi[1] if k & 2 else lo[1], hi[2] if k & 4 else lo[2]] for k in range(8)], F)
        w = _rot_to_world(s, c, corners)
        mn, mx = w[0].copy(), w[0].copy()
        for k in range(1, 8):
            mn, mx = npo.sse_min(w[k], mn), npo.sse_max(w[k], mx)
        return mn + off, mx + off

    def _object_rays(self, p, o, d):
        """rays (o, d) in the object space of rect primitives p"""
        x = self.xf[p]
        has = x >= 0
        if has.any():
            xs = np.where(has, x, 0)
            s, c = self.sin[xs], self.cos[xs]
            with np.errstate(**_err):
                oo = _rot_to_object(s, c, o - self.off[xs])
                dd = _rot_to_object(s, c, d)
            o = np.where(has[:, None], oo, o)
            d = np.where(has[:, None], dd, d)
        return o, d

    def trace(self, o, d):
        """Scene::trace + BvhNode::trace as in np_oracle, with a rect test for rect leaves"""
        n = o.shape[0]
        with np.errstate(**_err):
            inv = F(1.0) / d
        closest = np.full(n, F32_MAX, F)
        best = np.full(n, -1, np.int64)
        idx = np.zeros(n, np.int64)
        n_nodes = len(self.skip)
        a = dot(d, d)
        alive = idx < n_nodes
        rows = np.arange(n)
        while alive.any():
            r = np.nonzero(alive)[0]
            i = idx[r]
            leaf = self.prim[i] >= 0
            rs, ns = r[~leaf], i[~leaf]
            if len(rs):                                                      # aabb.rs:25-36
                with np.errstate(**_err):
                    v0 = (self.bmin[ns] - o[rs]) * inv[rs]
                    v1 = (self.bmax[ns] - o[rs]) * inv[rs]
                mn, mx = npo.sse_min(v0, v1), npo.sse_max(v0, v1)
                tmin = npo.rust_max(T_MIN, npo.rust_max(mn[:, 0], npo.rust_max(mn[:, 1], mn[:, 2])))
                tmax = npo.rust_min(closest[rs], npo.rust_min(mx[:, 0], npo.rust_min(mx[:, 1], mx[:, 2])))
                idx[rs] = np.where(tmin < tmax, ns + 1, self.skip[ns])
            rl, nl = r[leaf], i[leaf]
            if len(rl):
                p = self.prim[nl]
                sph = self.pkind[p] == SPHERE
                q, ps = rl[sph], p[sph]
                if len(q):                                                   # sphere.rs:46-69
                    with np.errstate(**_err):
                        oc = o[q] - self.c[ps]
                        b = dot(oc, d[q])
                        c = dot(oc, oc) - self.r[ps] * self.r[ps]
                        disc = b * b - a[q] * c
                        sq = np.sqrt(np.where(disc > 0, disc, F(0.0)))
                        near = (-b - sq) / a[q]
                        far = (-b + sq) / a[q]
                    pos = disc > 0
                    ok_near = pos & (near < closest[q]) & (near > T_MIN)
                    ok_far = pos & ~ok_near & (far < closest[q]) & (far > T_MIN)
                    hit = ok_near | ok_far
                    closest[q] = np.where(hit, np.where(ok_near, near, far), closest[q])
                    best[q] = np.where(hit, ps, best[q])
                q, pr = rl[~sph], p[~sph]
                if len(q):                                                   # the extension's rect
                    oo, dd = self._object_rays(pr, o[q], d[q])
                    k = np.arange(len(q))
                    ax, ia, ib = self.axis[pr], self.ia[pr], self.ib[pr]
                    with np.errstate(**_err):
                        t = (self.v[pr, 4] - oo[k, ax]) / dd[k, ax]
                        ca = oo[k, ia] + dd[k, ia] * t
                        cb = oo[k, ib] + dd[k, ib] * t
                        hit = (t < closest[q]) & (t > T_MIN) & (ca >= self.v[pr, 0]) & (ca <= self.v[pr, 1]) \
                            & (cb >= self.v[pr, 2]) & (cb <= self.v[pr, 3])
                    closest[q] = np.where(hit, t, closest[q])
                    best[q] = np.where(hit, pr, best[q])
                idx[rl] = nl + 1
            alive = idx < n_nodes
        del rows
        return closest, best

    def record(self, o, d, t, prim):
        """HitRecord (point, normal) of the hits"""
        n = len(t)
        p, nrm = np.zeros((n, 3), F), np.zeros((n, 3), F)
        sph = self.pkind[prim] == SPHERE
        if sph.any():
            with np.errstate(**_err):
                ps = o[sph] + d[sph] * t[sph][:, None]                       # ray.rs:26-28
                p[sph], nrm[sph] = ps, normalized(ps - self.c[prim[sph]])    # sphere.rs:37-42
        rc = ~sph
        if rc.any():
            pr = prim[rc]
            oo, dd = self._object_rays(pr, o[rc], d[rc])
            with np.errstate(**_err):
                pt = oo + dd * t[rc][:, None]
            nn = np.zeros((len(pr), 3), F)
            nn[np.arange(len(pr)), self.axis[pr]] = np.where(self.flip[pr] != 0, F(-1.0), F(1.0))
            x = self.xf[pr]
            has = x >= 0
            if has.any():
                xs = np.where(has, x, 0)
                s, c = self.sin[xs], self.cos[xs]
                with np.errstate(**_err):
                    pw = _rot_to_world(s, c, pt) + self.off[xs]
                    nw = _rot_to_world(s, c, nn)
                pt = np.where(has[:, None], pw, pt)
                nn = np.where(has[:, None], nw, nn)
            p[rc], nrm[rc] = pt, nn
        return p, nrm


def render_ext(scene, cam, S, seed, sample_range=None):
    """trace_ray with an emitted term and the integrator's parameters -> (radiance sums (H, W, 3), hit ids of sample 0)"""
    W, H = cam["width"], cam["height"]
    npx = W * H
    I = scene.integrator
    s0, s1 = sample_range if sample_range is not None else (0, S * S)
    total = np.zeros((npx, 3), F)
    pixel = np.arange(npx, dtype=np.uint32)
    for s in range(s0, s1):
        o, d = npo.sample_rays(cam, S, s)
        rng = npo.path_stream(seed, pixel, s).copy()
        att = np.ones((npx, 3), F)
        rad = np.zeros((npx, 3), F)
        background = np.ones(npx, bool)                  # the path ends on the background term (miss or bounce cap)
        active = np.arange(npx)
        for bounce in range(1, I["max_bounces"] + 1):
            if not len(active):
                break
            t, prim = scene.trace(o[active], d[active])
            hit = prim >= 0
            hi, pr, th = active[hit], prim[hit], t[hit]
            m = scene.mat[pr]
            light = scene.kind[m] == DIFFUSE_LIGHT
            with np.errstate(**_err):
                rad[hi[light]] = rad[hi[light]] + scene.albedo[m[light]] * att[hi[light]]     # emitted x attenuation
            background[hi[light]] = False                # a light does not scatter: the path returns its radiance
            hs, ps, ts, ms = hi[~light], pr[~light], th[~light], m[~light]
            p, nrm = scene.record(o[hs], d[hs], ts, ps)
            st = rng[hs]
            ok, a, _, nd = scene.scatter_at(d[hs], p, nrm, ms, st)
            rng[hs] = st
            background[hs[~ok]] = False                  # absorbed (metal.rs:29-31)
            g = hs[ok]
            with np.errstate(**_err):
                att[g] = att[g] * a[ok]
            o[g], d[g] = p[ok], nd[ok]
            if I["rr_start"] >= 0 and bounce >= I["rr_start"]:
                q = npo.rust_max(att[g, 0], npo.rust_max(att[g, 1], att[g, 2]))
                play = q < F(1.0)
                gp = g[play]
                st = rng[gp]
                r = npo.rng_f32(st)
                rng[gp] = st
                live = r < q[play]
                background[gp[~live]] = False
                with np.errstate(**_err):
                    att[gp[live]] = att[gp[live]] / q[play][live][:, None]
                g = np.concatenate([g[~play], gp[live]])
            active = g
        with np.errstate(**_err):
            if I["background"] is None:                  # raytracer.rs:67-68
                tt = F(0.5) * (d[:, 1] + F(1.0))
                col = (np.ones(3, F) * (F(1.0) - tt)[:, None]) + (np.array([0.5, 0.7, 1.0], F) * tt[:, None])
            else:
                col = np.broadcast_to(np.asarray(I["background"], F), (npx, 3))
            color = np.where(background[:, None], rad + col * att, rad)
        total = total + color
    ids = scene.trace(*npo.sample_rays(cam, S, 0))[1].astype(np.int32).reshape(H, W)
    return total.reshape(H, W, 3), ids
