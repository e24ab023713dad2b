"""The two independently written CPU restatements -- oracle/pt_oracle.c (C, scalar,
recursive BVH) and oracle/np_oracle.py (numpy float32, vectorised, skip-link walk) --
must agree BIT FOR BIT: BVH, primary hit ids and fp32 radiance sums.  With no reference
fixtures and no Rust toolchain this is the strongest pin the oracle can have."""
import os
import sys

import numpy as np
import pytest

from conftest import MAIN_RS_CAMERA, RENDER_SEED, make_raytracer

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import np_oracle  # noqa: E402


def twin_scene(rt):
    p = rt.packed
    mats = [(m.kind, tuple(m.albedo), m.param, m.texture) for m in p.materials]
    return np_oracle.Scene(p.spheres_numpy(), mats, p.textures)


def same_bits(a, b):
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    return bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))))


@pytest.mark.parametrize("kind,w,h,s", [("simple", 48, 27, 3), ("rtow", 40, 22, 3), ("rtow", 17, 9, 2)])
def test_twins_agree_bit_for_bit(oracle, kind, w, h, s):
    rt = make_raytracer(kind, w, h, s)
    c_scene = oracle.OracleScene(rt.packed)
    ref = c_scene.render(rt.camera.pod, s, seed=RENDER_SEED, hit_ids=True)
    tw = twin_scene(rt)
    cam = np_oracle.camera_new(MAIN_RS_CAMERA["eyes"], MAIN_RS_CAMERA["target"], MAIN_RS_CAMERA["up"],
                               MAIN_RS_CAMERA["fov"], w, h)
    # the twin derives its own camera: must equal the product's pt_camera_look_at bit for bit
    pod = rt.camera.pod
    for name, field in (("llc", pod.lower_left_corner), ("origin", pod.origin), ("horizontal", pod.horizontal),
                        ("vertical", pod.vertical)):
        assert same_bits(cam[name], np.array(field[:], np.float32)), name
    rad, ids = np_oracle.render(tw, cam, s, RENDER_SEED)
    assert np.array_equal(ids, ref["hit_id"])
    assert same_bits(rad, ref["radiance"]), "%d radiance words differ" % np.sum(
        rad.view(np.uint32) != ref["radiance"].view(np.uint32))


def test_twin_bvh_equals_c_oracle_and_product(oracle):
    rt = make_raytracer("rtow", 16, 9, 2)
    tw = twin_scene(rt)
    nodes = oracle.OracleScene(rt.packed).flatten()
    prod = rt.nodes
    assert len(nodes) == len(tw.skip) == len(prod)
    for i in range(len(nodes)):
        for nd in (nodes[i], prod[i]):
            assert nd.skip == tw.skip[i] and nd.prim == tw.prim[i]
            assert same_bits(np.array(nd.bmin[:], np.float32), tw.bmin[i])
            assert same_bits(np.array(nd.bmax[:], np.float32), tw.bmax[i])


def test_twin_sample_shard(oracle):
    rt = make_raytracer("simple", 24, 14, 3)
    c_scene = oracle.OracleScene(rt.packed)
    tw = twin_scene(rt)
    cam = np_oracle.camera_new(MAIN_RS_CAMERA["eyes"], MAIN_RS_CAMERA["target"], MAIN_RS_CAMERA["up"],
                               MAIN_RS_CAMERA["fov"], 24, 14)
    ref = c_scene.render(rt.camera.pod, 3, seed=5, sample_range=(4, 8))["radiance"]
    rad, _ = np_oracle.render(tw, cam, 3, 5, sample_range=(4, 8))
    assert same_bits(rad, ref)


# ---- the extension (rects, instances, a diffuse light, integrator options) ----
# Not in the reference: the "oracle" for it is this repository's own specification, restated twice
# (oracle/pt_oracle.c in C, oracle/np_oracle_ext.py in numpy) -- the two must agree bit for bit.

def ext_twin(rt):
    import np_oracle_ext
    p = rt.packed
    prims = [dict(kind=q.kind, material=q.material, transform=q.transform, axis=q.axis, flip=q.flip, v=list(q.v))
             for q in p.prims]
    mats = [(m.kind, tuple(m.albedo), m.param, m.texture) for m in p.materials]
    xf = [(tuple(t.offset), t.sin_theta, t.cos_theta) for t in p.transforms]
    tw = np_oracle_ext.ExtScene(prims, mats, p.textures, xf)
    if rt.integrator is not None:
        pod = rt.integrator.pod
        tw.integrator = dict(max_bounces=pod.max_bounces, background=tuple(pod.background) if pod.background_kind == 1 else None,
                             rr_start=pod.rr_start)
    return tw


def ext_mixed_scene():
    """spheres of every material + plain, flipped and rotated rects + a light, under the sky"""
    import path_tracer_amd as pt
    sc = pt.Scene.new()
    sc.add(pt.Sphere.new((0.0, -100.5, -1.0), 100.0, pt.Lambertian.new_solid((0.5, 0.6, 0.4))))
    sc.add(pt.Sphere.new((0.0, 0.0, -1.0), 0.5, pt.Dielectric.new(1.5)))
    sc.add(pt.Sphere.new((1.2, 0.0, -1.3), 0.5, pt.Metal.new((0.8, 0.6, 0.2), 0.2)))
    sc.add(pt.Rect(2, -2.0, 2.0, -0.5, 1.5, -2.5, pt.Lambertian.new_solid((0.7, 0.2, 0.2))))
    sc.add(pt.Rect(1, -0.6, 0.6, -1.6, -0.4, 1.4, pt.DiffuseLight.new((4.0, 4.0, 3.0)), flip=True))
    tr = pt.Transform.new((-1.6, -0.5, -1.6), 30.0)
    for r in pt.box_rects((0.0, 0.0, 0.0), (0.6, 0.9, 0.6), pt.Metal.new((0.9, 0.9, 0.9), 0.0), tr):
        sc.add(r)
    return sc


EXT_CAMERA = dict(eyes=(0.5, 0.8, 2.5), target=(0.0, 0.0, -1.0), up=(0.0, 1.0, 0.0), fov=50.0)


@pytest.mark.parametrize("case", ["cornell", "mixed_sky_rr", "mixed_defaults"])
def test_extension_twins_agree_bit_for_bit(oracle, case):
    import path_tracer_amd as pt
    import np_oracle_ext
    if case == "cornell":
        sc = pt.Scene.new()
        pt.cornell_box(sc)
        w, h, s = 24, 24, 3
        cam_args = pt.CORNELL_CAMERA
        integ = pt.Integrator(max_bounces=12, background=(0.0, 0.0, 0.0))
    else:
        sc = ext_mixed_scene()
        w, h, s = 32, 20, 3
        cam_args = EXT_CAMERA
        integ = pt.Integrator(max_bounces=8, rr_start=2) if case == "mixed_sky_rr" else None
    cam_pod = pt.Camera.new(width=w, height=h, **cam_args)
    rt = pt.Raytracer.new(cam_pod, sc, s, integrator=integ)
    c_scene = oracle.OracleScene(rt.packed)
    if integ is not None:
        c_scene.set_integrator(integ.pod)
    ref = c_scene.render(cam_pod.pod, s, seed=RENDER_SEED, hit_ids=True)
    tw = ext_twin(rt)
    # the tree: C oracle == numpy twin == product (pt_bvh_build_ext)
    nodes, prod = c_scene.flatten(), rt.nodes
    assert len(nodes) == len(tw.skip) == len(prod)
    for i in range(len(nodes)):
        for nd in (nodes[i], prod[i]):
            assert nd.skip == tw.skip[i] and nd.prim == tw.prim[i]
            assert same_bits(np.array(nd.bmin[:], np.float32), tw.bmin[i])
            assert same_bits(np.array(nd.bmax[:], np.float32), tw.bmax[i])
    cam = np_oracle.camera_new(cam_args["eyes"], cam_args["target"], cam_args["up"], cam_args["fov"], w, h)
    rad, ids = np_oracle_ext.render_ext(tw, cam, s, RENDER_SEED)
    assert np.array_equal(ids, ref["hit_id"])
    assert same_bits(rad, ref["radiance"]), "%d radiance words differ" % np.sum(
        rad.view(np.uint32) != ref["radiance"].view(np.uint32))
    assert np.isfinite(ref["radiance"]).all() and ref["radiance"].max() > 0.0


def test_extension_integrator_defaults_reproduce_the_reference_path(oracle):
    """A sphere-only scene pushed through the extension's scene type and integrator (defaults: 5 bounces, sky,
    no roulette) gives the reference path's radiance bit for bit: radiance stays +0 and 0 + x == x."""
    rt = make_raytracer("rtow", 40, 22, 3)
    ref = oracle.OracleScene(rt.packed).render(rt.camera.pod, 3, seed=RENDER_SEED, hit_ids=True)
    import path_tracer_amd as pt
    sc = pt.Scene.new()
    pt.random_scene(sc, conftest_scene_seed(), pt.RTOW_BOUNDS)
    sc.force_ext = True
    rt2 = pt.Raytracer.new(rt.camera, sc, 3)
    got = oracle.OracleScene(rt2.packed).render(rt.camera.pod, 3, seed=RENDER_SEED, hit_ids=True)
    assert np.array_equal(got["hit_id"], ref["hit_id"])
    assert same_bits(got["radiance"], ref["radiance"])


def conftest_scene_seed():
    from conftest import SCENE_SEED
    return SCENE_SEED
