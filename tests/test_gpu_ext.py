"""GPU parity tests of the EXTENSION (rects, Translate(RotateY(..)) instances, a diffuse light, integrator
options: include/pt_amd.h) -- BASELINE config 4, the Cornell box.

The reference contains none of this (one primitive, three materials, fixed 5 bounces, sky), so no reference
parity exists: the oracle is this repository's own specification, restated twice (oracle/pt_oracle.c and
oracle/np_oracle_ext.py, which tests/test_oracle_twin.py requires to agree bit for bit).  The HIP library goes
through its C ABI (pt_upload_scene_ext, pt_set_integrator, pt_render) and is compared with the C oracle: primary
hit ids bit-exact, fp32 radiance sums bit for bit (and so within the north-star tolerance)."""
import numpy as np
import pytest

from conftest import RENDER_SEED, SCENE_SEED, make_raytracer
from test_gpu_parity import same_bits, tonemapped_agreement
from test_oracle_twin import EXT_CAMERA, ext_mixed_scene

pytestmark = pytest.mark.gpu


def cornell(width, height, samples, **integ):
    import path_tracer_amd as pt
    sc = pt.Scene.new()
    pt.cornell_box(sc)
    cam = pt.Camera.new(width=width, height=height, **pt.CORNELL_CAMERA)
    kw = dict(max_bounces=50, background=(0.0, 0.0, 0.0))
    kw.update(integ)
    return pt.Raytracer.new(cam, sc, samples, integrator=pt.Integrator(**kw))


def check(renderer, oracle, rt, samples, lanes=1, seed=RENDER_SEED):
    osc = oracle.OracleScene(rt.packed)
    if rt.integrator is not None:
        osc.set_integrator(rt.integrator.pod)
    renderer.set_lanes_per_pixel(lanes)
    try:
        used = renderer.lanes_per_pixel(samples, 0, samples * samples) if lanes != 1 else 1
        img, rad, ids = renderer.render(rt, seed=seed, radiance=True, hit_ids=True)
    finally:
        renderer.set_lanes_per_pixel(0)
    chunk = 0 if lanes == 1 else -(-(samples * samples) // used)
    ref = osc.render(rt.camera.pod, samples, seed=seed, hit_ids=True, chunk=chunk)
    assert np.array_equal(ids, ref["hit_id"]), "%d primary hit ids differ" % np.sum(ids != ref["hit_id"])
    assert tonemapped_agreement(osc, rad, ref["radiance"], samples) >= 0.999
    assert same_bits(rad, ref["radiance"]), "%d of %d radiance words differ" % (
        np.sum(rad.view(np.uint32) != ref["radiance"].view(np.uint32)), rad.size)
    assert np.array_equal(img, osc.resolve(ref["radiance"], samples))
    return rad, ids


@pytest.mark.parametrize("w,h,s,lanes", [(128, 128, 4, 1), (75, 53, 3, 1), (96, 96, 4, 0)])
def test_cornell_box_bit_exact(renderer, oracle, w, h, s, lanes):
    rad, ids = check(renderer, oracle, cornell(w, h, s), s, lanes)
    assert rad.max() > 0.0 and set(np.unique(ids)) >= {0, 1, 2, 3, 4, 5}      # walls and the light are seen directly


@pytest.mark.parametrize("extra_boxes", [0, 1], ids=["two_transforms_cached", "three_transforms_general"])
def test_object_ray_cache_and_general_instantiation(renderer, oracle, extra_boxes):
    """The Cornell box as it is runs the kernel instantiation that keeps a walk's object-space rays in LDS (at most two
    transforms: pt_device.hpp obj_cache_fill); with a third instanced box it runs the general one.  Both equal the
    oracle bit for bit, in both lane modes, with a glass box and a sphere among the instances (rect and sphere
    Leaves in one LEAF run)."""
    import path_tracer_amd as pt
    sc = pt.Scene.new()
    pt.cornell_box(sc)
    sc.add(pt.Sphere.new((190.0, 90.0, 190.0), 90.0, pt.Dielectric.new(1.5)))
    for i in range(extra_boxes):
        tr = pt.Transform.new((60.0 + 40.0 * i, 330.0, 120.0), 33.0)
        for r in pt.box_rects((0.0, 0.0, 0.0), (80.0, 60.0, 80.0), pt.Metal.new((0.8, 0.8, 0.9), 0.1), tr):
            sc.add(r)
    cam = pt.Camera.new(width=96, height=96, **pt.CORNELL_CAMERA)
    rt = pt.Raytracer.new(cam, sc, 4, integrator=pt.Integrator(max_bounces=50, background=(0.0, 0.0, 0.0)))
    check(renderer, oracle, rt, 4, 1)
    check(renderer, oracle, rt, 4, 0)


def test_cornell_box_full_frame_size_reduced_spp(renderer, oracle):
    """BASELINE config 4's frame (1024 x 1024) at 9 spp, bounce cap 50"""
    check(renderer, oracle, cornell(1024, 1024, 3), 3)


@pytest.mark.parametrize("integ", [dict(max_bounces=8, rr_start=2), dict(max_bounces=3), None,
                                   dict(max_bounces=20, background=(0.1, 0.2, 0.3), rr_start=0)],
                         ids=["sky_rr", "cap3", "defaults", "constant_rr0"])
def test_mixed_scene_bit_exact(renderer, oracle, integ):
    """spheres of all three materials, plain / flipped / rotated rects and a light, with Russian roulette,
    a short bounce cap, the defaults, and a constant background"""
    import path_tracer_amd as pt
    cam = pt.Camera.new(width=160, height=100, **EXT_CAMERA)
    rt = pt.Raytracer.new(cam, ext_mixed_scene(), 4, integrator=pt.Integrator(**integ) if integ else None)
    check(renderer, oracle, rt, 4)


def test_extension_kernel_reproduces_the_reference_path(renderer, oracle):
    """The RTOW scene through pt_upload_scene_ext (extension kernel, default integrator) equals the same scene
    through pt_upload_scene (the reference path's kernel) bit for bit -- and so the oracle of the reference path."""
    import path_tracer_amd as pt
    rt = make_raytracer("rtow", 160, 90, 4)
    renderer.set_lanes_per_pixel(1)
    try:
        _, base, base_ids = renderer.render(rt, seed=RENDER_SEED, radiance=True, hit_ids=True)
        sc = pt.Scene.new()
        pt.random_scene(sc, SCENE_SEED, pt.RTOW_BOUNDS)
        sc.force_ext = True
        _, ext, ext_ids = renderer.render(pt.Raytracer.new(rt.camera, sc, 4), seed=RENDER_SEED, radiance=True, hit_ids=True)
    finally:
        renderer.set_lanes_per_pixel(0)
    assert np.array_equal(base_ids, ext_ids)
    assert same_bits(base, ext)


def test_extension_error_codes(renderer):
    import ctypes as C
    import path_tracer_amd as pt
    from path_tracer_amd import _abi
    lib, ctx = pt.lib(), renderer._ctx
    mats = (_abi.PtMaterial * 1)(_abi.PtMaterial(_abi.PT_LAMBERTIAN, (C.c_float * 3)(0.5, 0.5, 0.5), 0.0, -1))
    bad_axis = (_abi.PtPrimitive * 1)(_abi.PtPrimitive(_abi.PT_PRIM_RECT, 0, -1, 3, 0, (C.c_float * 5)(0, 1, 0, 1, 0)))
    assert lib.pt_upload_scene_ext(ctx, bad_axis, 1, mats, 1, None, 0, None, 0, None, 0) == -1
    sphere_xf = (_abi.PtPrimitive * 1)(_abi.PtPrimitive(_abi.PT_PRIM_SPHERE, 0, 0, 0, 0, (C.c_float * 5)(0, 0, 0, 1, 0)))
    xf = (_abi.PtTransform * 1)(_abi.PtTransform((C.c_float * 3)(0, 0, 0), 0.0, 1.0))
    assert lib.pt_upload_scene_ext(ctx, sphere_xf, 1, mats, 1, None, 0, xf, 1, None, 0) == -1
    bad_integ = _abi.PtIntegrator(0, 0, (C.c_float * 3)(0, 0, 0), -1)
    assert lib.pt_set_integrator(ctx, C.byref(bad_integ)) == -1
    light = (_abi.PtMaterial * 1)(_abi.PtMaterial(_abi.PT_DIFFUSE_LIGHT, (C.c_float * 3)(1, 1, 1), 0.0, -1))
    sph = (_abi.PtSphere * 1)(_abi.PtSphere(0, 0, 0, 1, 0))
    assert lib.pt_upload_scene(ctx, sph, 1, light, 1, None, 0, None, 0) == -1     # lights need pt_upload_scene_ext
    with pytest.raises(ValueError):
        pt.Raytracer.new(pt.Camera.new(width=8, height=8, **EXT_CAMERA), _sphere_scene(pt), 2, integrator=pt.Integrator(max_bounces=9))
    assert renderer.render(make_raytracer("simple", 32, 16, 2)).shape == (16, 32, 3)   # the context survives


def _sphere_scene(pt):
    sc = pt.Scene.new()
    sc.add(pt.Sphere.new((0, 0, 0), 1.0, pt.Lambertian.new_solid((0.5, 0.5, 0.5))))
    return sc


def _ext_fuzz_material(rnd):
    import path_tracer_amd as pt
    m = int(rnd.integers(0, 5))
    if m <= 1:
        return pt.Lambertian.new_solid(tuple(float(x) for x in rnd.uniform(0.05, 0.95, 3)))
    if m == 2:
        return pt.Metal.new(tuple(float(x) for x in rnd.uniform(0.3, 1.0, 3)), float(rnd.choice([0.0, 0.3, 1.0])))
    if m == 3:
        return pt.Dielectric.new(float(rnd.choice([1.5, 0.7, 2.4])))
    return pt.DiffuseLight.new(tuple(float(x) for x in rnd.uniform(0.5, 8.0, 3)))


def _ext_fuzz_case(k):
    """Seeded random extension scene + camera + integrator (tools/ext_fuzz_campaign.py runs 300,000 of them)."""
    import path_tracer_amd as pt
    rnd = np.random.default_rng(500000 + k)
    sc = pt.Scene.new()
    n_rects = int(rnd.choice([1, 2, 4, 8, 14]))
    for _ in range(n_rects):
        axis = int(rnd.integers(0, 3))
        a0, b0 = (float(x) for x in rnd.uniform(-3.0, 2.0, 2))
        a1, b1 = a0 + float(rnd.uniform(0.2, 4.0)), b0 + float(rnd.uniform(0.2, 4.0))
        sc.add(pt.Rect(axis, a0, a1, b0, b1, float(rnd.uniform(-3.0, 3.0)), _ext_fuzz_material(rnd), flip=bool(rnd.integers(0, 2))))
    for _ in range(int(rnd.choice([0, 1, 2, 3]))):
        tr = pt.Transform.new(tuple(float(x) for x in rnd.uniform(-2.5, 2.5, 3)), float(rnd.choice([0.0, 15.0, -18.0, 45.0, 90.0, 133.0, -270.0])))
        size = tuple(float(x) for x in rnd.uniform(0.3, 2.0, 3))
        for r in pt.box_rects((0.0, 0.0, 0.0), size, _ext_fuzz_material(rnd), tr if rnd.integers(0, 4) else None):
            sc.add(r)
    for _ in range(int(rnd.choice([0, 1, 3, 9]))):
        sc.add(pt.Sphere.new(tuple(float(x) for x in rnd.uniform(-3.0, 3.0, 3)), float(rnd.choice([0.1, 0.5, 1.5, 30.0], p=[0.3, 0.4, 0.25, 0.05])),
                             _ext_fuzz_material(rnd)))
    eye = rnd.uniform(-5.0, 5.0, 3)
    target = rnd.uniform(-1.0, 1.0, 3)
    if k % 7 == 0:                                          # axis-aligned view: zero direction components
        eye, target = np.array([0.0, 0.0, 6.0]), np.array([0.0, 0.0, 0.0])
    fov = float(rnd.choice([10.0, 40.0, 90.0, 150.0]))
    w, h = [(33, 17), (64, 36), (48, 48), (24, 64)][k % 4]
    cam = pt.Camera.new(tuple(map(float, eye)), tuple(map(float, target)), (0.0, 1.0, 0.0), fov, w, h)
    kind = k % 4
    if kind == 0:
        integ = None
    elif kind == 1:
        integ = pt.Integrator(max_bounces=int(rnd.choice([1, 3, 12, 50])), background=tuple(float(x) for x in rnd.uniform(0.0, 0.5, 3)))
    elif kind == 2:
        integ = pt.Integrator(max_bounces=int(rnd.choice([4, 20])), rr_start=int(rnd.choice([0, 2, 5])))
    else:
        integ = pt.Integrator(max_bounces=50, background=(0.0, 0.0, 0.0), rr_start=int(rnd.choice([-1, 3])))
    samples = int(rnd.choice([2, 3, 5]))
    return pt.Raytracer.new(cam, sc, samples, integrator=integ), samples


@pytest.mark.parametrize("k", range(24))
def test_extension_fuzz_scenes_bit_exact(renderer, oracle, k):
    """random rects / instanced boxes / spheres / lights / integrator options / cameras, both lane modes"""
    rt, s = _ext_fuzz_case(k)
    check(renderer, oracle, rt, s, 1, seed=k)
    check(renderer, oracle, rt, s, 0, seed=k)
