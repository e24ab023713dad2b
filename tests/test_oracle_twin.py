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
