"""GPU parity tests: the HIP library (through its C ABI) against the CPU oracle on the
same scene, camera and seed.  Bars (BASELINE.json north_star / SURVEY.md section 4):
  * primary-ray hit primitive ids: bit-exact;
  * per-pixel radiance at equal spp: >= 99.9 % of pixels within 1/255 after tonemapping
    -- and, because oracle and library share the portable definitions P1-P3 and the
    fp32 summation order, the fp32 radiance sums are in fact compared BIT FOR BIT;
  * size-independent properties at BASELINE's full frame sizes.
"""
import numpy as np
import pytest

from conftest import RENDER_SEED, make_raytracer

pytestmark = pytest.mark.gpu


def same_bits(a, b):
    """bit-for-bit equality of two float32 arrays, NaN payloads ignored"""
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    nan = np.isnan(a) & np.isnan(b)
    return bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | nan))


def tonemapped_agreement(oracle, rad_a, rad_b, samples):
    """fraction of pixels whose 8-bit tonemapped colours differ by at most 1 (= 1/255) in every channel"""
    sc = oracle  # any OracleScene: resolve is scene independent
    a = sc.resolve(rad_a, samples).astype(np.int32)
    b = sc.resolve(rad_b, samples).astype(np.int32)
    return float(np.mean(np.all(np.abs(a - b) <= 1, axis=-1)))


CASES = [
    # kind, width, height, samples
    ("simple", 160, 90, 4),
    ("rtow", 160, 90, 4),
    ("rtow", 67, 41, 3),      # ragged: neither dimension a multiple of the 8x8 tile
    ("reference", 96, 54, 2),  # the ~10.7k-sphere built-in scene
]


@pytest.mark.parametrize("kind,w,h,s", CASES)
def test_primary_hit_ids_bit_exact(renderer, oracle, kind, w, h, s):
    rt = make_raytracer(kind, w, h, s)
    osc = oracle.OracleScene(rt.packed)
    ref = osc.render(rt.camera.pod, s, radiance=False, hit_ids=True)["hit_id"]
    _, _, ids = renderer.render(rt, seed=RENDER_SEED, hit_ids=True)
    assert ids.shape == ref.shape
    assert np.array_equal(ids, ref), "%d of %d primary hit ids differ" % (np.sum(ids != ref), ids.size)
    assert (ids >= 0).any() and (ids == -1).any()   # the case exercises both hits and misses


@pytest.mark.parametrize("kind,w,h,s", CASES)
def test_radiance_bit_exact_one_lane_per_pixel(renderer, oracle, kind, w, h, s):
    """lanes_per_pixel = 1 reproduces the reference's sequential sum order (raytracer.rs:88)."""
    rt = make_raytracer(kind, w, h, s)
    osc = oracle.OracleScene(rt.packed)
    ref = osc.render(rt.camera.pod, s, seed=RENDER_SEED)["radiance"]
    renderer.set_lanes_per_pixel(1)
    try:
        img, rad, _ = renderer.render(rt, seed=RENDER_SEED, radiance=True)
    finally:
        renderer.set_lanes_per_pixel(0)
    assert tonemapped_agreement(osc, rad, ref, s) >= 0.999       # the stated tolerance
    assert same_bits(rad, ref), "%d of %d radiance words differ" % (
        np.sum(rad.view(np.uint32) != ref.view(np.uint32)), rad.size)
    assert np.array_equal(img, osc.resolve(ref, s))              # the u8 buffer of main.rs:98


@pytest.mark.parametrize("lanes", [2, 4, 8, 16, 64])
def test_radiance_bit_exact_split_pixels(renderer, oracle, lanes):
    """Small frames split a pixel's samples over several lanes: contiguous chunks summed in
    order.  The oracle's `chunk` argument restates that order."""
    kind, w, h, s = "rtow", 61, 35, 5
    rt = make_raytracer(kind, w, h, s)
    osc = oracle.OracleScene(rt.packed)
    renderer.set_lanes_per_pixel(lanes)
    try:
        used = renderer.lanes_per_pixel(s, 0, s * s)
        _, rad, _ = renderer.render(rt, seed=RENDER_SEED, radiance=True)
    finally:
        renderer.set_lanes_per_pixel(0)
    chunk = -(-(s * s) // used)
    ref = osc.render(rt.camera.pod, s, seed=RENDER_SEED, chunk=chunk)["radiance"]
    assert same_bits(rad, ref)
    seq = osc.render(rt.camera.pod, s, seed=RENDER_SEED)["radiance"]
    assert tonemapped_agreement(osc, rad, seq, s) >= 0.999


def test_auto_lanes_and_default_render(renderer, oracle):
    rt = make_raytracer("simple", 100, 60, 4)
    osc = oracle.OracleScene(rt.packed)
    renderer.bind(rt)
    used = renderer.lanes_per_pixel(4, 0, 16)
    assert used == 16            # tiny frame: as many lanes per pixel as there are samples (capped)
    img, rad, _ = renderer.render(rt, seed=3, radiance=True)
    ref = osc.render(rt.camera.pod, 4, seed=3, chunk=1)["radiance"]
    assert same_bits(rad, ref)
    assert np.array_equal(img, osc.resolve(ref, 4))


def test_sample_shards_sum_to_full_frame(renderer, oracle):
    """spp-sharding: each shard is bit-exact against the oracle's shard; the shard sum agrees
    with the unsharded frame to rounding."""
    import path_tracer_amd as pt
    rt = make_raytracer("rtow", 96, 54, 4)
    osc = oracle.OracleScene(rt.packed)
    renderer.set_lanes_per_pixel(1)
    try:
        full = renderer.render(rt, seed=RENDER_SEED, radiance=True)[1]
        total = np.zeros_like(full)
        for rank in range(4):
            rng = pt.shard_range(4, rank, 4)
            part = renderer.render(rt, seed=RENDER_SEED, sample_range=rng, radiance=True)[1]
            ref = osc.render(rt.camera.pod, 4, seed=RENDER_SEED, sample_range=rng)["radiance"]
            assert same_bits(part, ref)
            total += part
    finally:
        renderer.set_lanes_per_pixel(0)
    assert np.allclose(total, full, rtol=1e-5, atol=1e-6)
    assert tonemapped_agreement(osc, total, full, 4) >= 0.999


def test_error_codes(renderer):
    import path_tracer_amd as pt
    rt = make_raytracer("simple", 32, 16, 2)
    renderer.bind(rt)
    with pytest.raises(pt.PtError):
        renderer.render(rt, sample_range=(0, 5))          # beyond samples^2
    with pytest.raises(pt.PtError):
        renderer.set_lanes_per_pixel(3)                    # not a power of two
    with pytest.raises(ValueError):
        renderer.render(rt, buffer=np.zeros((3, 3, 3), np.uint8))
    assert renderer.render(rt).shape == (16, 32, 3)        # the context survives the errors
