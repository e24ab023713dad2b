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

from conftest import MAIN_RS_CAMERA, RENDER_SEED, make_raytracer

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


def test_group_render_matches_oracle_shards(oracle):
    """pt_group_render (ncclCommInitAll + ncclReduce inside the library): with one GPU the frame equals the
    oracle's bit for bit; with two (when the box has them) it equals the sum of the oracle's two shard
    renders bit for bit (a + b is commutative, so RCCL's order cannot matter)."""
    import path_tracer_amd as pt
    torch = pytest.importorskip("torch")
    rt = make_raytracer("rtow", 160, 90, 6)
    osc = oracle.OracleScene(rt.packed)
    for n in (1, 2):
        if torch.cuda.device_count() < n:
            continue
        group = pt.RendererGroup.new(range(n))
        try:
            group.set_lanes_per_pixel(1)
            img, rad = group.render(rt, seed=RENDER_SEED, radiance=True)
            total = None
            for r in range(n):
                part = osc.render(rt.camera.pod, 6, seed=RENDER_SEED, sample_range=pt.group_shard(6, r, n))["radiance"]
                total = part if total is None else total + part
            assert same_bits(rad, total)
            assert np.array_equal(img, osc.resolve(total, 6))
            k, red = group.timings()
            assert len(k) == n and all(x > 0.0 for x in k) and red >= 0.0
            img2 = group.render(rt, seed=RENDER_SEED)              # a second frame through the same communicators
            assert np.array_equal(img2, img)
        finally:
            group.close()


def test_group_wait_reports_a_failed_or_late_gpu_instead_of_hanging(oracle):
    """The bounded wait of pt_group_render (hipStreamQuery + ncclCommGetAsyncError per GPU): a GPU that reports an
    asynchronous RCCL error, or does not finish within the group's timeout, comes back as PT_ERR_HIP naming it; the
    communicators are aborted, the group refuses further renders, and destroying it returns.  The faults are
    injected on the host side of the poll (pt_group_inject_fault): nothing is provoked on the GPU."""
    import path_tracer_amd as pt
    rt = make_raytracer("rtow", 160, 90, 4)
    osc = oracle.OracleScene(rt.packed)
    for kind, word in ((1, "asynchronous error"), (2, "had not finished")):
        group = pt.RendererGroup.new([0])
        try:
            compiled, loaded = group.rccl_version()
            assert compiled // 10000 == loaded // 10000 == 2            # the same major version answers
            group.set_lanes_per_pixel(1)
            img = group.render(rt, seed=RENDER_SEED)                    # a good frame first
            assert np.array_equal(img, osc.resolve(osc.render(rt.camera.pod, 4, seed=RENDER_SEED)["radiance"], 4))
            with pytest.raises(pt.PtError):
                group.set_timeout(-1)
            with pytest.raises(pt.PtError):
                group.inject_fault(1, 1)                                # no such GPU in a group of one
            group.set_timeout(300)
            group.inject_fault(0, kind)
            with pytest.raises(pt.PtError) as err:
                group.render(rt, seed=RENDER_SEED)
            assert err.value.code == -2 and "GPU 0" in str(err.value) and word in str(err.value)
            with pytest.raises(pt.PtError) as err:                      # the group is gone, and says so
                group.render(rt, seed=RENDER_SEED)
            assert err.value.code == -2 and "aborted" in str(err.value)
        finally:
            group.close()                                               # returns: the stream did drain
    # a fresh group on the same GPU works after the aborts
    group = pt.RendererGroup.new([0])
    try:
        group.set_lanes_per_pixel(1)
        assert np.array_equal(group.render(rt, seed=RENDER_SEED), img)
    finally:
        group.close()


def test_clock_probe_reads_a_plausible_shader_clock(renderer):
    ghz = renderer.clock_probe()
    assert 0.8 < ghz < 2.6, ghz            # MI355X: up to 2.4 GHz, lower under load
    rt = make_raytracer("rtow", 640, 360, 8)
    renderer.render(rt, seed=1)
    ghz = renderer.last_path_kernel_clock_ghz()      # the persistent launch's own two stamps
    assert 0.8 < ghz < 2.6, ghz
    renderer.render(make_raytracer("simple", 64, 36, 2), seed=1)
    assert renderer.last_path_kernel_clock_ghz() == -1.0   # the small-tree launch does not stamp


def test_stream_ordered_render_reduce_resolve_equals_render(renderer, oracle):
    """The one-process-per-GPU step of bench.py with a single rank: render_device -> (reduce) -> resolve_device
    -> D2H, all on ONE non-default torch stream, equals pt_render's image."""
    torch = pytest.importorskip("torch")
    rt = make_raytracer("simple", 320, 180, 4)
    dev = torch.device("cuda", 0)
    n = rt.width() * rt.height() * 3
    accum = torch.zeros(n, dtype=torch.float32, device=dev)
    rgb8 = torch.zeros(n, dtype=torch.uint8, device=dev)
    host = torch.zeros(n, dtype=torch.uint8).pin_memory()
    side = torch.cuda.Stream(dev)
    with torch.cuda.stream(side):
        renderer.render_device(rt, accum.data_ptr(), seed=9, stream=side.cuda_stream)
        renderer.resolve_device(rt, accum.data_ptr(), rgb8.data_ptr(), stream=side.cuda_stream)
        host.copy_(rgb8, non_blocking=True)
    side.synchronize()
    assert np.array_equal(host.numpy().reshape(rt.height(), rt.width(), 3), renderer.render(rt, seed=9))


def test_second_stream_while_a_launch_is_running_is_refused(renderer, oracle):
    """A context's launches share one work-distribution scratch (pt_amd.h): a launch on a second stream
    while the previous one is still running is refused (PT_ERR_INVALID), never raced; once the first has
    finished the second stream is fine, and the first launch's result is untouched."""
    import path_tracer_amd as pt
    torch = pytest.importorskip("torch")
    rt = make_raytracer("rtow", 1280, 720, 16)                 # tens of milliseconds of kernel time
    dev = torch.device("cuda", 0)
    n = rt.width() * rt.height() * 3
    a, b = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    buf_a = torch.zeros(n, dtype=torch.float32, device=dev)
    buf_b = torch.zeros(n, dtype=torch.float32, device=dev)
    renderer.bind(rt)
    torch.cuda.synchronize()
    renderer.render_device(rt, buf_a.data_ptr(), seed=5, stream=a.cuda_stream)
    with pytest.raises(pt.PtError) as err:
        renderer.render_device(rt, buf_b.data_ptr(), seed=5, stream=b.cuda_stream)
    assert err.value.code == -1
    a.synchronize()
    renderer.render_device(rt, buf_b.data_ptr(), seed=5, stream=b.cuda_stream)
    b.synchronize()
    assert renderer.last_path_kernel_ms() > 0.0
    assert same_bits(buf_a.cpu().numpy(), buf_b.cpu().numpy())


def test_two_contexts_render_on_one_gpu_at_once(renderer, oracle):
    """Two contexts, two streams, launches in flight together: each launch is sized to fill the GPU by itself
    (persistent) and waits only for items its own running waves hold, so the second one takes the slots the
    first one frees -- both frames come out as when rendered alone."""
    import path_tracer_amd as pt
    torch = pytest.importorskip("torch")
    rt_a = make_raytracer("rtow", 960, 540, 12)
    rt_b = make_raytracer("reference", 640, 360, 6)
    other = pt.ParallelRenderer.new(0)
    try:
        dev = torch.device("cuda", 0)
        sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        buf_a = torch.zeros(rt_a.width() * rt_a.height() * 3, dtype=torch.float32, device=dev)
        buf_b = torch.zeros(rt_b.width() * rt_b.height() * 3, dtype=torch.float32, device=dev)
        renderer.bind(rt_a)
        other.bind(rt_b)
        torch.cuda.synchronize()
        renderer.render_device(rt_a, buf_a.data_ptr(), seed=5, stream=sa.cuda_stream)
        other.render_device(rt_b, buf_b.data_ptr(), seed=6, stream=sb.cuda_stream)
        sa.synchronize()
        sb.synchronize()
        alone_a = renderer.render(rt_a, seed=5, radiance=True)[1]
        alone_b = other.render(rt_b, seed=6, radiance=True)[1]
        assert same_bits(buf_a.cpu().numpy().reshape(alone_a.shape), alone_a)
        assert same_bits(buf_b.cpu().numpy().reshape(alone_b.shape), alone_b)
    finally:
        other.close()


def test_axis_aligned_rays_nan_and_inf_slabs(renderer, oracle):
    """Camera looking straight down -z with eye.x = eye.y = 0: the centre column / row of
    pixels get direction components that are EXACTLY zero (1/0 = inf), and spheres are placed
    so that box planes coincide with the ray origin (0 * inf = NaN).  These rays take the
    general slab path (SSE 'second operand on NaN' rule, aabb.rs:25-36 + vec.rs:50-51)."""
    import path_tracer_amd as pt
    scene = pt.Scene.new()
    grey = pt.Lambertian.new_solid((0.6, 0.6, 0.6))
    metal = pt.Metal.new((0.9, 0.9, 0.9), 0.0)
    glass = pt.Dielectric.new(1.5)
    # bmin.x = 0 / bmax.x = 0 / bmin.y = 0 / bmax.y = 0 planes through the camera axis
    scene.add(pt.Sphere.new((0.5, 0.0, -3.0), 0.5, grey))
    scene.add(pt.Sphere.new((-0.5, 0.0, -4.0), 0.5, metal))
    scene.add(pt.Sphere.new((0.0, 0.5, -5.0), 0.5, glass))
    scene.add(pt.Sphere.new((0.0, -0.5, -6.0), 0.5, grey))
    scene.add(pt.Sphere.new((0.0, 0.0, -9.0), 1.0, metal))
    scene.add(pt.Sphere.new((0.0, -101.0, -5.0), 100.0, grey))
    w, h, s = 32, 16, 3
    cam = pt.Camera.new((0.0, 0.0, 0.0), (0.0, 0.0, -1.0), (0.0, 1.0, 0.0), 40.0, w, h)
    rt = pt.Raytracer.new(cam, scene, s)
    pod = cam.pod
    # sample 0 of column w/2 has u = 0.5 exactly -> dir.x == 0; row h/2 has v = 0.5 -> dir.y == 0
    dx = (np.float32(pod.lower_left_corner[0]) + np.float32(pod.horizontal[0]) * np.float32(0.5)) - np.float32(pod.origin[0])
    dy = (np.float32(pod.lower_left_corner[1]) + np.float32(pod.vertical[1]) * np.float32(0.5)) - np.float32(pod.origin[1])
    assert dx == 0.0 and dy == 0.0
    osc = oracle.OracleScene(rt.packed)
    ref = osc.render(pod, s, seed=11, hit_ids=True)
    renderer.set_lanes_per_pixel(1)
    try:
        img, rad, ids = renderer.render(rt, seed=11, radiance=True, hit_ids=True)
    finally:
        renderer.set_lanes_per_pixel(0)
    assert np.array_equal(ids, ref["hit_id"])
    assert same_bits(rad, ref["radiance"])
    assert np.array_equal(img, osc.resolve(ref["radiance"], s))


def test_textured_sphere_poles_and_seam(renderer, oracle):
    """Earth sphere seen from above its pole and along the u seam: exercises atan2(0,0),
    |n.y| >= 1 -> NaN -> row 0, and the clamp of picture.rs:24-26."""
    import path_tracer_amd as pt
    scene = pt.Scene.new()
    scene.add(pt.Sphere.new((0.0, 0.0, 0.0), 1.0, pt.Lambertian.new(pt.PictureTexture.earthmap())))
    for eye in ((0.0, 4.0, 0.0001), (-4.0, 0.0, 0.0), (0.0, -4.0, 0.0001)):
        cam = pt.Camera.new(eye, (0.0, 0.0, 0.0), (0.0, 0.0, 1.0) if eye[1] else (0.0, 1.0, 0.0), 35.0, 48, 48)
        rt = pt.Raytracer.new(cam, scene, 2)
        osc = oracle.OracleScene(rt.packed)
        ref = osc.render(cam.pod, 2, seed=3)["radiance"]
        renderer.set_lanes_per_pixel(1)
        try:
            _, rad, _ = renderer.render(rt, seed=3, radiance=True)
        finally:
            renderer.set_lanes_per_pixel(0)
        assert same_bits(rad, ref)


def test_single_sphere_scene_root_is_a_leaf(renderer, oracle):
    import path_tracer_amd as pt
    scene = pt.Scene.new()
    scene.add(pt.Sphere.new((0.0, 0.0, -2.0), 0.7, pt.Dielectric.new(1.5)))
    cam = pt.Camera.new((0.0, 0.0, 0.0), (0.0, 0.0, -1.0), (0.0, 1.0, 0.0), 50.0, 40, 24)
    rt = pt.Raytracer.new(cam, scene, 3)
    osc = oracle.OracleScene(rt.packed)
    ref = osc.render(cam.pod, 3, seed=2, hit_ids=True)
    renderer.set_lanes_per_pixel(1)
    try:
        _, rad, ids = renderer.render(rt, seed=2, radiance=True, hit_ids=True)
    finally:
        renderer.set_lanes_per_pixel(0)
    assert np.array_equal(ids, ref["hit_id"]) and same_bits(rad, ref["radiance"])


def test_library_built_bvh_equals_host_built(renderer):
    """nodes = NULL: pt_upload_scene builds the tree itself (bvh.rs:13-91)."""
    import path_tracer_amd as pt
    rt = make_raytracer("rtow", 80, 45, 2)
    a = renderer.render(rt, seed=1, radiance=True)[1]
    rt2 = make_raytracer("rtow", 80, 45, 2)
    rt2.nodes = None
    b = renderer.render(rt2, seed=1, radiance=True)[1]
    assert same_bits(a, b)


def test_big_scene_beyond_lds_prefix(renderer, oracle):
    """~10.7k spheres: 2048 of the 21k nodes sit in LDS, the rest come through the L1."""
    rt = make_raytracer("reference", 120, 68, 2)
    assert len(rt.nodes) > 2048
    osc = oracle.OracleScene(rt.packed)
    ref = osc.render(rt.camera.pod, 2, seed=RENDER_SEED, hit_ids=True)
    renderer.set_lanes_per_pixel(1)
    try:
        _, rad, ids = renderer.render(rt, seed=RENDER_SEED, radiance=True, hit_ids=True)
    finally:
        renderer.set_lanes_per_pixel(0)
    assert np.array_equal(ids, ref["hit_id"]) and same_bits(rad, ref["radiance"])


def test_tree_that_fills_the_staged_set(renderer, oracle):
    """~800 spheres, 1024 < nodes <= 2048: the whole tree is staged, both halves -- more than the 48 KiB of dynamic
    LDS a kernel gets without asking (launch_path raises the cap) -- in both walk orders."""
    import path_tracer_amd as pt
    scene = pt.Scene.new()
    pt.random_scene(scene, 7, (-11, 11, -11, 26))
    cam = pt.Camera.new(width=120, height=68, **MAIN_RS_CAMERA)
    rt = pt.Raytracer.new(cam, scene, 3)
    assert 1024 < len(rt.nodes) <= 2048
    osc = oracle.OracleScene(rt.packed)
    ref = osc.render(rt.camera.pod, 3, seed=RENDER_SEED, hit_ids=True)
    renderer.set_lanes_per_pixel(1)
    try:
        for view in (True, False):
            renderer.set_walk_order(view)
            _, rad, ids = renderer.render(rt, seed=RENDER_SEED, radiance=True, hit_ids=True)
            assert np.array_equal(ids, ref["hit_id"]) and same_bits(rad, ref["radiance"])
    finally:
        renderer.set_walk_order(True)
        renderer.set_lanes_per_pixel(0)


def test_accumulate_continues_the_sequential_sum(renderer, oracle):
    """pt_render_device(accumulate=1) continues the per-pixel sum where the previous launch
    stopped: two launches over [0,k) and [k,S^2) equal one launch, bit for bit."""
    import torch
    rt = make_raytracer("rtow", 72, 40, 3)
    osc = oracle.OracleScene(rt.packed)
    ref = osc.render(rt.camera.pod, 3, seed=9)["radiance"]
    acc = torch.zeros(72 * 40 * 3, dtype=torch.float32, device="cuda")
    renderer.set_lanes_per_pixel(1)
    try:
        st = torch.cuda.current_stream().cuda_stream
        renderer.render_device(rt, acc.data_ptr(), seed=9, sample_range=(0, 4), stream=st)
        renderer.render_device(rt, acc.data_ptr(), seed=9, sample_range=(4, 9), accumulate=True, stream=st)
        torch.cuda.synchronize()
    finally:
        renderer.set_lanes_per_pixel(0)
    assert same_bits(acc.cpu().numpy().reshape(40, 72, 3), ref)


def test_full_size_properties_1080p(renderer, oracle):
    """BASELINE frame size (1920x1080, the earth scene) at reduced spp -- properties that need no
    full-size oracle run: (1) determinism, (2) primary hit ids = the oracle's (cheap: no paths),
    (3) sample shards sum to the frame, (4) oracle agreement on a 64-row band."""
    import path_tracer_amd as pt
    rt = make_raytracer("simple", 1920, 1080, 4)
    a = renderer.render(rt, seed=5, radiance=True, hit_ids=True)
    b = renderer.render(rt, seed=5, radiance=True)
    assert same_bits(a[1], b[1]) and np.array_equal(a[0], b[0])
    osc = oracle.OracleScene(rt.packed)
    assert np.array_equal(a[2], osc.render(rt.camera.pod, 4, radiance=False, hit_ids=True)["hit_id"])
    parts = sum(renderer.render(rt, seed=5, sample_range=pt.shard_range(4, r, 2), radiance=True)[1] for r in range(2))
    assert np.allclose(parts, a[1], rtol=2e-6, atol=1e-6)
    other = renderer.render(rt, seed=6, radiance=True)[1]
    assert not same_bits(other, a[1])                       # the seed matters
    assert np.isfinite(a[1]).all() and (a[1] >= 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("kind,w,h,s", [("rtow", 1920, 1080, 6), ("simple", 1920, 1080, 16), ("reference", 3840, 2160, 3)],
                         ids=["cfg3-frame", "cfg2-frame", "cfg5-frame"])
def test_full_frames_bit_exact(renderer, oracle, kind, w, h, s):
    """The BASELINE frames at their real sizes and scenes, at a sample count the oracle finishes in
    seconds on the box's host cores (cfg 2 at its full 256 spp, cfg 3 at 36, cfg 5 at 9; the full cfg 3 and
    cfg 5 frames: tools/full_frame_check.py): pieces, one lane per pixel: every radiance sum and
    every primary hit id equals the oracle's bit for bit."""
    rt = make_raytracer(kind, w, h, s)
    osc = oracle.OracleScene(rt.packed)
    ref = osc.render(rt.camera.pod, s, seed=RENDER_SEED, hit_ids=True)
    renderer.bind(rt)
    assert renderer.lanes_per_pixel(s, 0, s * s) == 1
    img, rad, ids = renderer.render(rt, seed=RENDER_SEED, radiance=True, hit_ids=True)
    assert np.array_equal(ids, ref["hit_id"])
    assert same_bits(rad, ref["radiance"])
    assert np.array_equal(img, osc.resolve(ref["radiance"], s))


def test_cfg1_frame_against_oracle(renderer, oracle):
    """BASELINE config 1 at its real size: RTOW scene 400x225 @ 100 spp (9 M paths; the oracle
    needs a few seconds).  Auto lanes-per-pixel (8): bit-exact against the oracle's chunked sum,
    >= 99.9 % of pixels within 1/255 of the sequential-order frame, RMSE far below 1e-3."""
    rt = make_raytracer("rtow", 400, 225, 10)
    osc = oracle.OracleScene(rt.packed)
    renderer.bind(rt)
    lanes = renderer.lanes_per_pixel(10, 0, 100)
    img, rad, ids = renderer.render(rt, seed=RENDER_SEED, radiance=True, hit_ids=True)
    chunk = -(-100 // lanes)
    ref_chunked = osc.render(rt.camera.pod, 10, seed=RENDER_SEED, chunk=chunk, hit_ids=True)
    assert np.array_equal(ids, ref_chunked["hit_id"])
    assert same_bits(rad, ref_chunked["radiance"])
    seq = osc.render(rt.camera.pod, 10, seed=RENDER_SEED)["radiance"]
    assert tonemapped_agreement(osc, rad, seq, 10) >= 0.999
    tm = lambda r: np.sqrt(r / np.float32(100.0))
    assert float(np.sqrt(np.mean((tm(rad) - tm(seq)) ** 2))) < 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("fn", [0, 1, 2], ids=["rcp", "sqrt", "rsq"])
def test_unary_short_forms(renderer, fn):
    """The library's 6-10 instruction forms of 1/x, sqrt(x), 1/sqrt(x) (pt_device.hpp) return the
    correctly rounded IEEE result for every one of the 2^32 fp32 inputs (bit-exact bar)."""
    assert renderer.selftest_unary(fn) == 0


@pytest.mark.gpu
def test_reciprocal_division(renderer):
    """Sphere::trace's x / a through RN(1/a) and two fused corrections (pt_device.hpp div_by) equals
    the IEEE division on 2^34 operand pairs over the REGULAR class, half of them next to rounding
    midpoints (bit-exact bar; the class's edges in scenes: test_scaled_scenes_bit_exact)."""
    assert renderer.selftest_division(34) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("size,samples", [((1920, 1080), 64), ((1920, 1080), 16), ((3840, 2160), 32), ((400, 225), 10),
                                           ((1023, 767), 7), ((16, 9), 2), ((8191, 4099), 33)])
def test_sample_grid_division(renderer, size, samples):
    """Every sample-grid coordinate (raytracer.rs:77-85) of these frame sizes, computed through the
    reciprocal + fused corrections of the path kernel, equals the IEEE divisions bit for bit."""
    rt = make_raytracer("simple", size[0], size[1], samples)
    renderer.bind(rt)
    assert renderer.selftest_grid(samples) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("kind,size,samples", [("rtow", (128, 72), 12), ("simple", (200, 120), 9)])
def test_pieces_continue_the_sum_in_order(renderer, oracle, kind, size, samples):
    """One lane per pixel and > 32 samples: the launch is cut into (piece, tile) items that
    different waves render and chain through the radiance buffer (ticket order + per-tile
    release/acquire).  The per-pixel sum must still be the oracle's sequential fp32 sum."""
    rt = make_raytracer(kind, size[0], size[1], samples)
    osc = oracle.OracleScene(rt.packed)
    ref = osc.render(rt.camera.pod, samples, seed=77)["radiance"]
    renderer.set_lanes_per_pixel(1)
    try:
        img, rad, _ = renderer.render(rt, seed=77, radiance=True)
    finally:
        renderer.set_lanes_per_pixel(0)
    assert same_bits(rad, ref)
    assert np.array_equal(img, osc.resolve(ref, samples))


@pytest.mark.gpu
def test_render_progress_reports_and_matches_render(renderer, oracle):
    """pt_render_progress: the callback sees every pass (the reporter of main.rs:41-71), the image
    is the one-launch image bit for bit, and a true return value cancels."""
    import path_tracer_amd as pt
    rt = make_raytracer("rtow", 96, 54, 6)
    seen = []
    renderer.set_lanes_per_pixel(1)
    try:
        ref_img, ref_rad, _ = renderer.render(rt, seed=11, radiance=True)
        img, rad = renderer.render_progress(rt, 5, lambda done, total: seen.append((done, total)), seed=11, radiance=True)
    finally:
        renderer.set_lanes_per_pixel(0)
    assert [d for d, _ in seen] == [7, 14, 21, 28, 36] and all(t == 36 for _, t in seen)
    assert same_bits(rad, ref_rad) and np.array_equal(img, ref_img)
    with pytest.raises(pt.PtError) as err:
        renderer.render_progress(rt, 4, lambda done, total: done >= 18, seed=11)
    assert err.value.code == -6                                                   # PT_ERR_CANCELLED
    assert np.array_equal(renderer.render(rt, seed=11), renderer.render(rt, seed=11))   # context still usable


def _fuzz_case(k):
    """Seeded random scene + camera: sphere counts 1..80, overlapping / nested / huge / tiny spheres,
    all three materials (fuzz 0 and 1, ref_idx below and above 1, textured Lambertian), cameras
    inside spheres, axis-aligned views, extreme fields of view."""
    import path_tracer_amd as pt
    rnd = np.random.default_rng(1000 + k)
    scene = pt.Scene.new()
    n = int(rnd.choice([1, 2, 3, 7, 20, 45, 80]))
    tex = pt.PictureTexture.earthmap() if k % 3 == 0 else None
    for i in range(n):
        c = rnd.uniform(-4, 4, 3)
        r = float(rnd.choice([0.05, 0.3, 0.8, 2.5, 40.0], p=[0.2, 0.4, 0.25, 0.1, 0.05]))
        if i == 0 and k % 4 == 0:
            c, r = np.array([0.0, -500.0, 0.0]), 500.0
        m = rnd.integers(0, 4)
        if m == 0:
            mat = pt.Lambertian.new_solid(tuple(rnd.uniform(0, 1, 3)))
        elif m == 1 and tex is not None:
            mat = pt.Lambertian.new(tex)
        elif m <= 2:
            mat = pt.Metal.new(tuple(rnd.uniform(0.3, 1, 3)), float(rnd.choice([0.0, 0.2, 1.0])))
        else:
            mat = pt.Dielectric.new(float(rnd.choice([1.5, 0.7, 1.0, 2.4])))
        scene.add(pt.Sphere.new(tuple(float(x) for x in c), r, mat))
    eye = rnd.uniform(-6, 6, 3)
    target = rnd.uniform(-1, 1, 3)
    if k % 5 == 0:                                        # axis-aligned view: zero direction components
        eye, target = np.array([0.0, 0.0, 6.0]), np.array([0.0, 0.0, 0.0])
    fov = float(rnd.choice([1.0, 20.0, 60.0, 120.0, 170.0]))
    w, h = [(33, 17), (64, 36), (80, 45), (24, 64)][k % 4]
    cam = pt.Camera.new(tuple(map(float, eye)), tuple(map(float, target)), (0.0, 1.0, 0.0), fov, w, h)
    samples = int(rnd.choice([2, 3, 5, 7]))
    return pt.Raytracer.new(cam, scene, samples), samples


@pytest.mark.gpu
@pytest.mark.parametrize("k", range(24))
def test_fuzz_scenes_bit_exact(renderer, oracle, k):
    """Random scenes and cameras: radiance sums and primary hit ids equal the oracle's bit for bit,
    with one lane per pixel (sequential sums, pieces when samples^2 > 32) and with the automatic
    lanes-per-pixel split (chunked sums)."""
    rt, samples = _fuzz_case(k)
    osc = oracle.OracleScene(rt.packed)
    ref = osc.render(rt.camera.pod, samples, seed=k, hit_ids=True)
    renderer.set_lanes_per_pixel(1)
    try:
        img, rad, ids = renderer.render(rt, seed=k, radiance=True, hit_ids=True)
    finally:
        renderer.set_lanes_per_pixel(0)
    assert np.array_equal(ids, ref["hit_id"])
    assert same_bits(rad, ref["radiance"])
    assert np.array_equal(img, osc.resolve(ref["radiance"], samples))
    renderer.bind(rt)
    lanes = renderer.lanes_per_pixel(samples, 0, samples * samples)
    chunk = -(-samples * samples // lanes)
    _, rad2, _ = renderer.render(rt, seed=k, radiance=True)
    assert same_bits(rad2, osc.render(rt.camera.pod, samples, seed=k, chunk=chunk)["radiance"])


def _scaled_case(scale, k=3):
    """The same small scene at very different sizes: sphere centres and radii multiplied by `scale`,
    seen from the origin (the reference's camera keeps its image plane at distance 1 from the eye,
    camera.rs, so an eye far from the origin would quantise every direction away)."""
    import path_tracer_amd as pt
    rnd = np.random.default_rng(77 + k)
    scene = pt.Scene.new()
    scene.add(pt.Sphere.new((0.0, -101.0 * scale, -4.0 * scale), 100.0 * scale, pt.Lambertian.new_solid((0.8, 0.8, 0.0))))
    for i in range(12):
        c = (rnd.uniform(-1, 1, 3) * np.array([2.5, 1.0, 1.5]) + np.array([0.0, 0.0, -4.5])) * scale
        r = float(rnd.choice([0.2, 0.5, -0.45, 0.9])) * scale            # negative radius: the hollow-glass idiom
        mat = [pt.Lambertian.new_solid((0.7, 0.3, 0.3)), pt.Metal.new((0.8, 0.6, 0.2), 0.3), pt.Dielectric.new(1.5)][i % 3]
        scene.add(pt.Sphere.new(tuple(float(x) for x in c), r, mat))
    cam = pt.Camera.new((0.0, 0.0, 0.0), (0.0, 0.0, -1.0), (0.0, 1.0, 0.0), 60.0, 48, 27)
    return pt.Raytracer.new(cam, scene, 3), 3


@pytest.mark.gpu
@pytest.mark.parametrize("log2_scale", [-48, -44, -40, -14, -10, -8, 0, 12, 20, 24, 25, 26, 29, 31, 34])
def test_scaled_scenes_bit_exact(renderer, oracle, log2_scale):
    """Edges of the REGULAR class (pt_device.hpp regular_ray): the reciprocal-based division of
    Sphere::trace is used for coordinates below 2^32 and the IEEE division beyond (the ground
    sphere's radius passes 2^32 at scale 2^26, hit points pass it from 2^24 on), roots straddle
    T_MIN = 0.001 at the small scales; both must equal the oracle's plain division bit for bit.
    Beyond 2^35 and below 2^-48 the reference's BVH build panics (box volumes overflow / vanish, bvh.rs:19)."""
    rt, samples = _scaled_case(2.0 ** log2_scale)
    osc = oracle.OracleScene(rt.packed)
    ref = osc.render(rt.camera.pod, samples, seed=5, hit_ids=True)
    img, rad, ids = renderer.render(rt, seed=5, radiance=True, hit_ids=True)
    assert np.array_equal(ids, ref["hit_id"])
    lanes = renderer.lanes_per_pixel(samples, 0, samples * samples)
    chunk = -(-samples * samples // lanes)
    assert same_bits(rad, osc.render(rt.camera.pod, samples, seed=5, chunk=chunk)["radiance"])


@pytest.mark.gpu
@pytest.mark.parametrize("kind,w,h,s", [("rtow", 160, 90, 4), ("simple", 160, 90, 4), ("reference", 96, 54, 2)])
def test_walk_orders_agree(renderer, oracle, kind, w, h, s):
    """pt_set_walk_order: PT_WALK_VIEW (the default every other test runs under: the child nearer to the camera
    first) and PT_WALK_REFERENCE (left child first, bvh.rs:111-118) return the oracle's hit ids and radiance,
    bit for bit -- a ray's closest hit does not depend on the order in which its candidates are found."""
    rt = make_raytracer(kind, w, h, s)
    osc = oracle.OracleScene(rt.packed)
    ref = osc.render(rt.camera.pod, s, seed=RENDER_SEED, hit_ids=True)
    renderer.set_lanes_per_pixel(1)
    try:
        for view in (True, False, True):
            renderer.set_walk_order(view)
            img, rad, ids = renderer.render(rt, seed=RENDER_SEED, radiance=True, hit_ids=True)
            assert renderer.walk_order() == int(view)
            assert np.array_equal(ids, ref["hit_id"])
            assert same_bits(rad, ref["radiance"])
            assert np.array_equal(img, osc.resolve(ref["radiance"], s))
    finally:
        renderer.set_walk_order(True)
        renderer.set_lanes_per_pixel(0)


@pytest.mark.gpu
def test_view_order_follows_the_camera(renderer, oracle):
    """pt_set_camera alone re-threads the node arrays for the new eye (no re-upload of the scene): the same
    scene seen from the opposite side, then from inside the sphere field, still equals the oracle bit for bit."""
    import ctypes as C
    import path_tracer_amd as pt
    scene = pt.Scene.new()
    pt.random_scene(scene, 0x5EED5CE7E, pt.RTOW_BOUNDS)
    first = None
    renderer.set_lanes_per_pixel(1)
    try:
        for eye in ((12.0, 2.0, 3.0), (-12.0, 3.0, -4.0), (0.5, 0.3, 0.5), (0.0, 30.0, 0.1)):
            cam = pt.Camera.new(eye, (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 30.0, 96, 54)
            rt = pt.Raytracer.new(cam, scene, 3)
            if first is None:
                first = rt
                renderer.bind(rt)
            else:                                   # camera only: the scene stays as uploaded for `first`
                pt._abi.check(renderer._ctx, pt.lib().pt_set_camera(renderer._ctx, C.byref(cam.pod)))
                rt.packed, rt.nodes = first.packed, first.nodes
                renderer._raytracer = rt
            osc = oracle.OracleScene(rt.packed)
            ref = osc.render(cam.pod, 3, seed=3, hit_ids=True)
            img, rad, ids = renderer.render(rt, seed=3, radiance=True, hit_ids=True)
            assert renderer.walk_order() == 1
            assert np.array_equal(ids, ref["hit_id"])
            assert same_bits(rad, ref["radiance"])
    finally:
        renderer._raytracer = None
        renderer.set_lanes_per_pixel(0)


@pytest.mark.gpu
def test_ties_go_to_the_reference_walks_first_candidate(renderer, oracle):
    """Two spheres with the same centre and radius return the same t for every ray; the reference keeps the one
    its walk meets first (strict '<', sphere.rs:57).  Under PT_WALK_VIEW the walk may meet them the other way
    round: the lower rank in the reference's order wins the tie (pt_device.hpp closer), from either side."""
    import path_tracer_amd as pt
    scene = pt.Scene.new()
    scene.add(pt.Sphere.new((0.0, -100.5, -1.0), 100.0, pt.Lambertian.new_solid((0.8, 0.8, 0.0))))
    scene.add(pt.Sphere.new((0.0, 0.0, -1.0), 0.5, pt.Lambertian.new_solid((0.9, 0.1, 0.1))))
    scene.add(pt.Sphere.new((1.0, 0.0, -1.0), 0.5, pt.Metal.new((0.8, 0.6, 0.2), 0.0)))
    scene.add(pt.Sphere.new((0.0, 0.0, -1.0), 0.5, pt.Dielectric.new(1.5)))
    scene.add(pt.Sphere.new((-1.0, 0.0, -1.0), 0.5, pt.Dielectric.new(1.5)))
    scene.add(pt.Sphere.new((1.0, 0.0, -1.0), 0.5, pt.Lambertian.new_solid((0.1, 0.1, 0.9))))
    try:
        for eye in ((0.0, 0.5, 2.0), (0.0, 0.5, -4.0), (5.0, 0.2, -1.0), (-5.0, 0.2, -1.0)):
            cam = pt.Camera.new(eye, (0.0, 0.0, -1.0), (0.0, 1.0, 0.0), 60.0, 64, 36)
            rt = pt.Raytracer.new(cam, scene, 3)
            osc = oracle.OracleScene(rt.packed)
            ref = osc.render(cam.pod, 3, seed=9, hit_ids=True)
            lanes = None
            for view in (True, False):
                renderer.set_walk_order(view)
                img, rad, ids = renderer.render(rt, seed=9, radiance=True, hit_ids=True)
                assert renderer.walk_order() == int(view)
                assert np.array_equal(ids, ref["hit_id"])
                lanes = renderer.lanes_per_pixel(3, 0, 9)
                assert same_bits(rad, osc.render(cam.pod, 3, seed=9, chunk=-(-9 // lanes))["radiance"])
    finally:
        renderer.set_walk_order(True)


@pytest.mark.gpu
def test_boxes_that_do_not_contain_their_spheres_keep_the_reference_order(renderer, oracle):
    """A sphere of negative radius (the hollow-glass idiom) gets a box with min and max swapped
    (Sphere::bounding_box is centre -+ radius as written, sphere.rs:71-76), and the boxes above it no longer
    contain it: a box test can then hide a closer hit behind a farther one found earlier, so the result depends
    on the order of the walk.  Such a tree is walked in the reference's order whatever pt_set_walk_order says."""
    import path_tracer_amd as pt
    scene = pt.Scene.new()
    scene.add(pt.Sphere.new((0.0, -100.5, -1.0), 100.0, pt.Lambertian.new_solid((0.8, 0.8, 0.0))))
    scene.add(pt.Sphere.new((0.0, 0.0, -1.0), 0.5, pt.Lambertian.new_solid((0.1, 0.2, 0.5))))
    scene.add(pt.Sphere.new((-1.0, 0.0, -1.0), 0.5, pt.Dielectric.new(1.5)))
    scene.add(pt.Sphere.new((2.6, 0.0, -1.0), -0.45, pt.Dielectric.new(1.5)))     # on its own: nothing else covers it
    scene.add(pt.Sphere.new((1.0, 0.0, -1.0), 0.5, pt.Metal.new((0.8, 0.6, 0.2), 0.0)))
    for eye in ((3.0, 2.0, 2.0), (-3.0, 0.5, -4.0)):
        cam = pt.Camera.new(eye, (0.0, 0.0, -1.0), (0.0, 1.0, 0.0), 40.0, 64, 36)
        rt = pt.Raytracer.new(cam, scene, 4)
        osc = oracle.OracleScene(rt.packed)
        ref = osc.render(cam.pod, 4, seed=2, hit_ids=True)
        renderer.set_walk_order(True)
        img, rad, ids = renderer.render(rt, seed=2, radiance=True, hit_ids=True)
        assert renderer.walk_order() == 0
        assert np.array_equal(ids, ref["hit_id"])
        lanes = renderer.lanes_per_pixel(4, 0, 16)
        assert same_bits(rad, osc.render(cam.pod, 4, seed=2, chunk=-(-16 // lanes))["radiance"])


@pytest.mark.gpu
@pytest.mark.parametrize("kind,w,h,s,lanes", [("rtow", 320, 180, 4, 1), ("rtow", 200, 120, 4, 0), ("reference", 256, 144, 3, 1)])
def test_primary_lists_on_and_off_agree(renderer, oracle, kind, w, h, s, lanes):
    """pt_set_primary_lists: camera rays that test their pixel block's candidate spheres instead of walking the tree
    return the oracle's radiance bit for bit, as the plain walk does; the hit-id AOV (always a walk) agrees too."""
    rt = make_raytracer(kind, w, h, s)
    osc = oracle.OracleScene(rt.packed)
    renderer.set_lanes_per_pixel(lanes)
    try:
        for on in (True, False, True):
            renderer.set_primary_lists(on)
            img, rad, ids = renderer.render(rt, seed=RENDER_SEED, radiance=True, hit_ids=True)
            assert renderer.primary_lists() == on
            used = renderer.lanes_per_pixel(s, 0, s * s)
            ref = osc.render(rt.camera.pod, s, seed=RENDER_SEED, chunk=-(-(s * s) // used), hit_ids=True)
            assert same_bits(rad, ref["radiance"]), "lists %s: %d radiance words differ" % (on, np.sum(rad.view(np.uint32) != ref["radiance"].view(np.uint32)))
            assert np.array_equal(ids, ref["hit_id"])
    finally:
        renderer.set_primary_lists(True)
        renderer.set_lanes_per_pixel(0)


@pytest.mark.gpu
def test_primary_lists_crowded_blocks_eye_inside_a_sphere_and_duplicates(renderer, oracle):
    """Blocks with more candidates than a list holds (a row of spheres seen end on) walk the tree; an eye inside a
    glass sphere, spheres behind the eye, duplicated spheres (exact ties) and a huge ground sphere all come out as
    the oracle's, from several eyes, with lists on."""
    import path_tracer_amd as pt
    scene = pt.Scene.new()
    scene.add(pt.Sphere.new((0.0, -1000.3, 0.0), 1000.0, pt.Lambertian.new_solid((0.5, 0.5, 0.5))))
    for k in range(2, 42):                                              # 40 spheres in a row along -z
        mat = pt.Metal.new((0.8, 0.7, 0.6), 0.1) if k % 3 == 0 else (pt.Dielectric.new(1.5) if k % 3 == 1 else pt.Lambertian.new_solid((0.2, 0.4, 0.8)))
        scene.add(pt.Sphere.new((0.05 * (k % 5), 0.0, -float(k)), 0.3, mat))
    scene.add(pt.Sphere.new((0.0, 0.0, 0.5), 1.5, pt.Dielectric.new(1.5)))        # around the first eye
    scene.add(pt.Sphere.new((0.0, 0.0, 6.0), 1.0, pt.Lambertian.new_solid((0.9, 0.1, 0.1))))   # behind it
    for _ in range(2):                                                  # exact duplicates: ties in t
        scene.add(pt.Sphere.new((1.5, 0.0, -4.0), 0.5, pt.Lambertian.new_solid((0.1, 0.9, 0.1))))
    renderer.set_lanes_per_pixel(1)
    try:
        for eye, target in (((0.0, 0.0, 0.5), (0.0, 0.0, -10.0)), ((0.02, 0.1, 1.0), (0.1, 0.0, -20.0)),
                            ((6.0, 1.0, -20.0), (0.0, 0.0, -20.0)), ((0.0, 30.0, -20.0), (0.0, 0.0, -20.5))):
            cam = pt.Camera.new(eye, target, (0.0, 1.0, 0.0), 40.0, 96, 56)
            rt = pt.Raytracer.new(cam, scene, 3)
            osc = oracle.OracleScene(rt.packed)
            ref = osc.render(cam.pod, 3, seed=4, hit_ids=True)
            img, rad, ids = renderer.render(rt, seed=4, radiance=True, hit_ids=True)
            assert renderer.primary_lists()
            assert np.array_equal(ids, ref["hit_id"])
            assert same_bits(rad, ref["radiance"]), "eye %s: %d words differ" % (eye, np.sum(rad.view(np.uint32) != ref["radiance"].view(np.uint32)))
    finally:
        renderer.set_lanes_per_pixel(0)


@pytest.mark.gpu
def test_primary_lists_are_off_where_boxes_do_not_contain_their_spheres(renderer, oracle):
    """A tree with a box that does not contain its sphere (negative radius: sphere.rs:71-76) keeps the plain walk for
    camera rays too: there a box test can change the closest hit, so none may be skipped."""
    import path_tracer_amd as pt
    scene = pt.Scene.new()
    pt.random_scene(scene, 77, (-3, 3, -3, 3))
    scene.add(pt.Sphere.new((0.0, 1.0, 0.0), -0.9, pt.Dielectric.new(1.5)))
    cam = pt.Camera.new(width=96, height=54, **MAIN_RS_CAMERA)
    rt = pt.Raytracer.new(cam, scene, 3)
    osc = oracle.OracleScene(rt.packed)
    renderer.set_lanes_per_pixel(1)
    try:
        img, rad, _ = renderer.render(rt, seed=2, radiance=True)
        assert not renderer.primary_lists() and renderer.walk_order() == 0
        assert same_bits(rad, osc.render(cam.pod, 3, seed=2)["radiance"])
    finally:
        renderer.set_lanes_per_pixel(0)


@pytest.mark.gpu
@pytest.mark.parametrize("target", [(0.0, 2.0, 0.0), (-4.0, 2.0, 0.0), (4.0, 2.0, 0.0)])
def test_primary_lists_rays_grazing_a_sphere_where_it_touches_its_boxes(renderer, oracle, target):
    """The defect the first version of the lists had, found by the full cfg 5 frame (11 pixels of 8.3 M) and by nothing
    smaller: main.rs's eye sits at y = 2, exactly the height of the tops of the three big spheres (centre y = 1, r = 1),
    so a horizontal camera ray passes a few 1e-6 above the very point where such a sphere touches the top face of every
    box around it.  The slab test says "above the box" exactly; Sphere::trace's discriminant (a difference of two numbers
    of size 150) cannot tell 1e-6 above from 1e-6 below and reports a hit for some of those rays -- which the reference
    never tests, because its walk stops at the box.  A chain of Leaves alone accepted them; the chain now carries each
    candidate's parent box.  Here the camera of main.rs looks horizontally at one of those poles through a 0.01-degree
    field of view, so that most of the frame is within rounding of it; lists on, lists off and the oracle must agree bit
    for bit.  (tools/lists_diag.py shows the same on the full cfg 5 frame.)"""
    import path_tracer_amd as pt
    scene = pt.Scene.new()
    pt.random_scene(scene, 0x5EED5CE7E, pt.RTOW_BOUNDS)
    cam = pt.Camera.new((12.0, 2.0, 3.0), target, (0.0, 1.0, 0.0), 0.01, 64, 64)
    rt = pt.Raytracer.new(cam, scene, 8)
    osc = oracle.OracleScene(rt.packed)
    ref = osc.render(cam.pod, 8, seed=8, hit_ids=True)
    renderer.set_lanes_per_pixel(1)
    try:
        for on in (True, False):
            renderer.set_primary_lists(on)
            img, rad, ids = renderer.render(rt, seed=8, radiance=True, hit_ids=True)
            assert renderer.primary_lists() == on
            assert np.array_equal(ids, ref["hit_id"])
            assert same_bits(rad, ref["radiance"]), "lists %s: %d radiance words differ" % (on, np.sum(rad.view(np.uint32) != ref["radiance"].view(np.uint32)))
    finally:
        renderer.set_primary_lists(True)
        renderer.set_lanes_per_pixel(0)
