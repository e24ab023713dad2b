"""CPU-only tests of the product's host side: the C-ABI library loads and exports every
symbol include/pt_amd.h declares, its pure-CPU helpers (BVH build, Camera::new, seeded
generators) agree bit for bit with the oracle's independent restatements, and errors
come back as status codes.  No GPU compute calls here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import path_tracer_amd as pt
from path_tracer_amd import _abi

from conftest import MAIN_RS_CAMERA, SCENE_SEED, make_raytracer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "pt_amd.h")).read()
    declared = set(re.findall(r"\b(pt_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_abi.SIGNATURES), declared ^ set(_abi.SIGNATURES)
    lib = pt.lib()                          # binds every signature; AttributeError if one is missing
    for name in declared:
        assert hasattr(lib, name)


def test_struct_layouts_match_header():
    assert C.sizeof(_abi.PtSphere) == 20 and C.sizeof(_abi.PtMaterial) == 24
    assert C.sizeof(_abi.PtBvhNode) == 32 and C.sizeof(_abi.PtCamera) == 56
    assert C.sizeof(_abi.PtTexture) == 16


def test_no_gpu_is_an_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pt.PtError) as e:
        pt.ParallelRenderer.new(0)
    assert e.value.code == -4               # PT_ERR_NO_DEVICE


def test_camera_matches_oracle(oracle):
    L = oracle.load()
    for (w, h, fov) in ((400, 225, 20.0), (1920, 1080, 20.0), (7, 3, 65.5)):
        cam = pt.Camera.new(width=w, height=h, **dict(MAIN_RS_CAMERA, fov=fov))
        ref = oracle.OraCamera()
        L.ora_camera_new(oracle.f3(MAIN_RS_CAMERA["eyes"]), oracle.f3(MAIN_RS_CAMERA["target"]),
                         oracle.f3(MAIN_RS_CAMERA["up"]), fov, w, h, C.byref(ref))
        assert bytes(cam.pod) == bytes(ref)
    bad = _abi.PtCamera()
    z = (C.c_float * 3)(0, 0, 0)
    assert pt.lib().pt_camera_look_at(z, z, z, 20.0, 0, 10, C.byref(bad)) == -1


@pytest.mark.parametrize("bounds", [pt.RTOW_BOUNDS, (-3, 2, -1, 4), pt.REFERENCE_BOUNDS])
def test_random_scene_matches_oracle(oracle, bounds):
    L = oracle.load()
    need = -L.ora_gen_random_scene(SCENE_SEED, *bounds, None, None, 0)
    s_ref, m_ref = (oracle.OraSphere * need)(), (oracle.OraMaterial * need)()
    assert L.ora_gen_random_scene(SCENE_SEED, *bounds, s_ref, m_ref, need) == need
    s_got, m_got = (_abi.PtSphere * need)(), (_abi.PtMaterial * need)()
    assert pt.lib().pt_gen_random_scene(SCENE_SEED, *bounds, s_got, m_got, need) == need
    assert bytes(s_got) == bytes(s_ref) and bytes(m_got) == bytes(m_ref)
    # too small a buffer reports the size needed
    assert pt.lib().pt_gen_random_scene(SCENE_SEED, *bounds, s_got, m_got, need - 1) == -need
    if bounds == pt.REFERENCE_BOUNDS:
        assert 10000 < need <= 44 * 243 + 4          # generators.rs:42-43: 10,692 candidates + 4
    if bounds == pt.RTOW_BOUNDS:
        assert 400 < need <= 22 * 22 + 4


def test_random_scene_material_mix(oracle):
    scene = pt.Scene.new()
    pt.random_scene(scene, SCENE_SEED, pt.RTOW_BOUNDS)
    kinds = [type(o.material).__name__ for o in scene.objects]
    n = len(kinds)
    assert 0.7 < kinds.count("Lambertian") / n < 0.9      # choose_mat < 0.8 (generators.rs:53)
    assert 0.08 < kinds.count("Metal") / n < 0.25
    assert kinds[-3:] == ["Dielectric", "Lambertian", "Metal"]   # generators.rs:72-88
    assert scene.objects[0].radius == 1000.0


def test_simple_scene_matches_oracle(oracle):
    L = oracle.load()
    s_ref, m_ref = (oracle.OraSphere * 6)(), (oracle.OraMaterial * 6)()
    assert L.ora_gen_simple_scene(s_ref, m_ref) == 6
    s_got, m_got = (_abi.PtSphere * 6)(), (_abi.PtMaterial * 6)()
    assert pt.lib().pt_gen_simple_scene(s_got, m_got) == 6
    assert bytes(s_got) == bytes(s_ref) and bytes(m_got) == bytes(m_ref)


def test_earthmap_fixture_is_the_decoded_reference_texture():
    import hashlib
    tex = pt.PictureTexture.earthmap()
    assert (tex.width, tex.height) == (1200, 600)
    assert hashlib.sha256(tex.rgb8.tobytes()).hexdigest().startswith("5bfdb5d86a920e66")


@pytest.mark.parametrize("kind", ["simple", "rtow", "reference"])
def test_bvh_build_matches_oracle(oracle, kind):
    """pt_bvh_build (prefix/suffix sweep, C++) vs the oracle's literal O(n^2) restatement of
    bvh.rs:13-91: identical pre-order trees, bit for bit."""
    rt = make_raytracer(kind, 16, 9, 2)
    ref = oracle.OracleScene(rt.packed).flatten()
    assert len(ref) == len(rt.nodes) == 2 * rt.packed.n_spheres - 1
    assert bytes(ref) == bytes(rt.nodes)
    prims = sorted(n.prim for n in rt.nodes if n.prim >= 0)
    assert prims == list(range(rt.packed.n_spheres))        # every sphere is exactly one leaf


def test_bvh_build_rejects_bad_arguments():
    s = (_abi.PtSphere * 2)(_abi.PtSphere(0, 0, 0, 1, 0), _abi.PtSphere(3, 0, 0, 1, 0))
    out = (_abi.PtBvhNode * 3)()
    assert pt.lib().pt_bvh_build(s, 2, out, 3) == 3
    assert pt.lib().pt_bvh_build(s, 2, out, 2) == -5        # PT_ERR_LIMIT
    assert pt.lib().pt_bvh_build(s, 0, out, 3) == -1        # bvh.rs:19: no objects
    assert out[0].prim == -1 and out[0].skip == 3 and out[1].prim == 0 and out[2].prim == 1


def test_scene_builder_api_mirrors_the_reference():
    scene = pt.Scene.new()
    with pytest.raises(ValueError):
        scene.pack()                                         # "Cannot build BVH without any objects."
    shared = pt.Metal.new((0.8, 0.6, 0.2), 0.0)
    scene.add(pt.Sphere.new((0, 0, -1), 0.5, shared))
    scene.add(pt.Sphere.new((1, 0, -1), 0.5, shared))        # shared Arc<dyn Material>: one entry
    scene.add(pt.Sphere.new((2, 0, -1), 0.5, pt.Lambertian.new(pt.SolidTexture.new((0.1, 0.2, 0.3)))))
    tex = pt.PictureTexture.from_rgb8(np.zeros((2, 2, 3), np.uint8))
    scene.add(pt.Sphere.new((3, 0, -1), 0.5, pt.Lambertian.new(tex)))
    p = scene.pack()
    assert (p.n_spheres, p.n_materials, len(p.textures)) == (4, 3, 1)
    assert [s.material for s in p.spheres] == [0, 0, 1, 2]
    assert p.materials[1].kind == 0 and p.materials[1].texture == -1 and tuple(p.materials[1].albedo) == pytest.approx((0.1, 0.2, 0.3))
    assert p.materials[2].texture == 0
    with pytest.raises(TypeError):
        scene.add("not a sphere")
    with pytest.raises(ValueError):
        pt.Raytracer.new(pt.Camera.new((0, 0, 0), (0, 0, -1), (0, 1, 0), 20.0, 4, 4), scene, 1)   # 0/0 grid


def test_shard_range_partitions_the_grid():
    for samples in (2, 3, 10, 64):
        for world in (1, 2, 3, 4, 8):
            ranges = [pt.shard_range(samples, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == samples * samples
            assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
            sizes = [b - a for a, b in ranges]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        pt.shard_range(4, 2, 2)


def test_med3_slab_identity():
    """The REGULAR slab test of pt_device.hpp uses two med3 chains.  For EVERY weak ordering of
    the eight values (T_MIN, closest, 3 x (v0, v1)) the decision equals the reference's
    max(T_MIN, lo_x, lo_y, lo_z) < min(closest, hi_x, hi_y, hi_z).  The operations are
    order-only, so seven levels already cover most orders; the full 8-level run is in DESIGN.md."""
    import itertools
    V = 6
    g = np.array(list(itertools.product(range(V), repeat=8)), dtype=np.int8)
    E, Fc, x0, x1, y0, y1, z0, z1 = [g[:, i] for i in range(8)]

    def med3(a, b, c):
        return np.maximum(np.minimum(a, b), np.minimum(np.maximum(a, b), c))
    true = np.maximum.reduce([E, np.minimum(x0, x1), np.minimum(y0, y1), np.minimum(z0, z1)]) < \
        np.minimum.reduce([Fc, np.maximum(x0, x1), np.maximum(y0, y1), np.maximum(z0, z1)])
    e = med3(med3(med3(E, x0, x1), y0, y1), z0, z1)
    f = med3(med3(med3(Fc, x0, x1), y0, y1), z0, z1)
    assert np.array_equal(true, e < f)
    # and on random floats, including infinities and signed zeros (no NaN: REGULAR rays)
    rng = np.random.default_rng(0)
    vals = rng.standard_normal((200000, 8)).astype(np.float32)
    vals[rng.random(vals.shape) < 0.05] = np.inf
    vals[rng.random(vals.shape) < 0.05] = -np.inf
    vals[rng.random(vals.shape) < 0.05] = 0.0
    vals[rng.random(vals.shape) < 0.05] = -0.0
    E, Fc, x0, x1, y0, y1, z0, z1 = vals.T
    true = np.maximum.reduce([E, np.minimum(x0, x1), np.minimum(y0, y1), np.minimum(z0, z1)]) < \
        np.minimum.reduce([Fc, np.maximum(x0, x1), np.maximum(y0, y1), np.maximum(z0, z1)])
    e = med3(med3(med3(E, x0, x1), y0, y1), z0, z1)
    f = med3(med3(med3(Fc, x0, x1), y0, y1), z0, z1)
    assert np.array_equal(true, e < f)


def test_cpp_host_png_writer_roundtrip(tmp_path):
    """host/pt_host.hpp write_png (the output.png of main.rs:118-121): a decoder reads back the
    exact pixels, including a frame wider than one 64 KiB stored block."""
    import subprocess
    from PIL import Image
    src = tmp_path / "png.cpp"
    src.write_text('#include "%s/host/pt_host.hpp"\n'
                   'int main(int c, char **v) { int w = 301, h = 97; std::vector<uint8_t> px((size_t)w * h * 3);\n'
                   '  for (size_t i = 0; i < px.size(); i++) px[i] = (uint8_t)((i * 2654435761u) >> 13);\n'
                   '  return c > 1 && pt::write_png(v[1], px.data(), w, h) ? 0 : 1; }\n' % ROOT)
    exe, out = tmp_path / "png", tmp_path / "o.png"
    subprocess.run(["g++", "-O1", "-std=c++17", "-o", str(exe), str(src)], check=True)
    subprocess.run([str(exe), str(out)], check=True)
    img = np.asarray(Image.open(out).convert("RGB"))
    i = np.arange(301 * 97 * 3, dtype=np.uint64)
    want = (((i * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)) >> np.uint64(13)).astype(np.uint8).reshape(97, 301, 3)
    assert img.shape == (97, 301, 3) and np.array_equal(img, want)


def test_bvh_build_refuses_degenerate_bounds():
    """bvh.rs:58-80 on zero-volume bounds: every candidate cost is NaN, the reference recurses into an
    empty range and panics (bvh.rs:19).  pt_bvh_build returns PT_ERR_INVALID -- it does not invent a split."""
    spheres = (_abi.PtSphere * 3)()
    for i in range(3):
        spheres[i] = _abi.PtSphere(1.0, 2.0, 3.0, 0.0, 0)          # three points: whole volume 0
    nodes = (_abi.PtBvhNode * 5)()
    assert pt.lib().pt_bvh_build(spheres, 3, nodes, 5) == -1       # PT_ERR_INVALID
    # non-finite bounds likewise
    spheres[1] = _abi.PtSphere(float("inf"), 0.0, 0.0, 1.0, 0)
    assert pt.lib().pt_bvh_build(spheres, 3, nodes, 5) == -1
    # and a regular input still builds
    for i in range(3):
        spheres[i] = _abi.PtSphere(float(i), 0.0, 0.0, 0.25, 0)
    assert pt.lib().pt_bvh_build(spheres, 3, nodes, 5) == 5
    assert pt.lib().pt_bvh_build(spheres, 3, nodes, 4) == -5       # PT_ERR_LIMIT: capacity


def test_group_shard_partitions_the_grid():
    """pt_group_shard (library) == distributed.shard_range (torch.distributed path): contiguous shares that
    tile [0, S*S), sizes differing by at most one."""
    for samples in (2, 3, 10, 16, 64):
        for n in (1, 2, 3, 4, 7, 8):
            if samples * samples < n:
                continue
            shares = [pt.group_shard(samples, r, n) for r in range(n)]
            assert shares == [pt.shard_range(samples, r, n) for r in range(n)]
            assert shares[0][0] == 0 and shares[-1][1] == samples * samples
            assert all(shares[r][1] == shares[r + 1][0] for r in range(n - 1))
            sizes = [e - b for b, e in shares]
            assert max(sizes) - min(sizes) <= 1


def test_group_shard_bad_arguments_give_the_empty_range():
    """n < 1 (a division by zero before round 3) or a rank outside [0, n): [0, 0), never a signal"""
    for samples, rank, n in ((8, 0, 0), (8, 0, -3), (8, -1, 4), (8, 4, 4), (8, 9, 4), (-2, 0, 1)):
        assert pt.group_shard(samples, rank, n) == (0, 0)
    b = C.c_int32(7)
    pt.lib().pt_group_shard(8, 0, 0, C.byref(b), None)             # NULL outputs are allowed
    assert b.value == 0


def test_group_create_argument_errors():
    lib = pt.lib()
    out = C.c_void_p()
    assert lib.pt_group_create(None, 1, C.byref(out)) == -1
    two = (C.c_int32 * 2)(0, 0)
    assert lib.pt_group_create(two, 2, C.byref(out)) == -1          # a device listed twice
    assert b"twice" in lib.pt_group_last_error(None)
    assert lib.pt_group_size(None) == 0 and not lib.pt_group_context(None, 0)
    assert lib.pt_group_set_timeout(None, 5) == -1 and lib.pt_group_inject_fault(None, 0, 1) == -1
    assert lib.pt_group_rccl_version(None, None, None) == -1 and lib.pt_clock_probe(None, None) == -1


def _c_structs(header):
    """{name: [(field, rust type)]} for every `typedef struct { ... } name;` of the header"""
    ctype = {"float": "f32", "int32_t": "i32", "int64_t": "i64", "uint64_t": "u64", "char": "c_char"}
    out = {}
    for body, name in re.findall(r"typedef struct \{(.*?)\}\s*(pt_[a-z_]+);", header, re.S):
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        fields = []
        for decl in filter(None, (d.strip() for d in body.split(";"))):
            m = re.match(r"(const\s+)?(\w+)\s+(.*)", decl, re.S)
            const, base, rest = m.group(1), m.group(2), m.group(3)
            for item in (x.strip() for x in rest.split(",")):
                ptr = item.startswith("*")
                ident, dim = re.match(r"\*?\s*(\w+)(?:\[(\d+)\])?", item).groups()
                rust = {"uint8_t": "u8"}.get(base, ctype.get(base))
                assert rust, decl
                if ptr:
                    rust = ("*const " if const else "*mut ") + rust
                if dim:
                    rust = "[%s; %s]" % (rust, dim)
                fields.append((ident, rust))
        out[name] = fields
    return out


def _rust_structs(source):
    out = {}
    for name, body in re.findall(r"#\[repr\(C\)\][^\n]*\npub struct (pt_\w+)\s*\{(.*?)\}", source, re.S):
        out[name] = [(f, " ".join(t.split())) for f, t in re.findall(r"pub (\w+):\s*([^,}]+?)\s*(?:,|$)", body.strip(), re.S)]
    return out


def _params(arglist):
    depth, n, seen = 0, 0, False
    for ch in arglist:
        if ch in "(<[":
            depth += 1
        elif ch in ")>]":
            depth -= 1
        elif ch == "," and depth == 0:
            n += 1
        if not ch.isspace():
            seen = True
    return n + 1 if seen and arglist.strip() != "void" else 0


def test_rust_binding_matches_header():
    """integration/rust/src/gpu.rs (uncompiled: no rustc here) against include/pt_amd.h: every struct with the
    same fields, in order, with the corresponding Rust types; every export declared, with the same number of
    parameters; the constants of the header's enums with the same values."""
    header = open(os.path.join(ROOT, "include", "pt_amd.h")).read()
    rust = open(os.path.join(ROOT, "integration", "rust", "src", "gpu.rs")).read()
    c_structs, r_structs = _c_structs(header), _rust_structs(rust)
    assert set(c_structs) == set(r_structs), set(c_structs) ^ set(r_structs)
    for name, fields in c_structs.items():
        assert fields == r_structs[name], (name, fields, r_structs[name])
    nocomment = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    c_fns = {n: _params(a) for n, a in re.findall(r"\b(pt_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", nocomment, re.S)}
    extern = rust[rust.index('extern "C" {'):]
    extern = re.sub(r"//[^\n]*", "", extern[:extern.index("\n}\n")])
    r_fns = {n: _params(a) for n, a in re.findall(r"pub fn (pt_\w+)\((.*?)\)\s*(?:->[^;]+)?;", extern, re.S)}
    assert set(c_fns) == set(_abi.SIGNATURES)
    assert set(r_fns) == set(c_fns), set(r_fns) ^ set(c_fns)
    assert r_fns == c_fns, {n: (c_fns[n], r_fns[n]) for n in c_fns if c_fns[n] != r_fns[n]}
    c_consts = {}
    for body in re.findall(r"enum\s*\{(.*?)\}", nocomment, re.S) + re.findall(r"typedef enum\s*\{(.*?)\}", nocomment, re.S):
        for n, v in re.findall(r"(PT_[A-Z_]+)\s*=\s*(-?\d+)", body):
            c_consts[n] = int(v)
    r_consts = {n: int(v) for n, v in re.findall(r"pub const (PT_[A-Z_]+): \w+ = (-?\d+);", rust)}
    assert r_consts == c_consts, set(r_consts.items()) ^ set(c_consts.items())


def test_item_call_saves_no_callee_saved_registers(tmp_path):
    """The persistent kernel calls render_item_call once per (piece, tile) item -- 524,288 times per headline frame.
    The function uses all 64 vector registers, 16 of them callee-saved by the ABI; saved and restored per call they
    were 4.8 GB of HBM traffic per frame for values nobody held.  `not_tail_called` lets the compiler drop the saves
    (DESIGN.md section 6); this test compiles the kernels' ISA and pins that: the reference scenes' item function keeps
    two whole-wave stores (the carriers of its saved scalars) and no other access to its private segment outside the
    texture path, so that a compiler or source change that brings the traffic back does not pass unnoticed."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "path_tracer_amd", "csrc")
    out = tmp_path / "pt_kernels.s"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize",
                    "-x", "hip", "--cuda-device-only", "-S", "-o", str(out), "pt_kernels.hip"], cwd=csrc, check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    lines = out.read_text().split("\n")
    for name in ("render_item_callILb1ELb0E", "render_item_callILb0ELb0E"):
        start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + name + r"\S*:", l))
        end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
        body = lines[start:end]
        stores = [l for l in body if "scratch_store" in l]
        assert len(stores) <= 2, stores
        # the ABI's callee-saved vector registers: v40-v47, v56-v63 (of which v62, v63 carry saved scalars here)
        saved = [l for l in stores if re.search(r"\bv(4[0-7]|5[6-9]|6[01])\b", l)]
        assert not saved, saved
