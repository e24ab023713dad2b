"""``Raytracer`` and ``ParallelRenderer``: the reference's L3/L4 interface
(src/raytracer.rs:29-44, src/parallel_renderer/mod.rs:61-66,73,155) with the CPU
tile loop replaced by the gfx950 library.  Everything numeric happens in
libpt_amd.so; this file is plumbing (ctypes + optional torch device buffers).
"""
import ctypes as C

import numpy as np

from . import _abi
from .distributed import shard_range  # noqa: F401  (re-exported)
from .scene import Camera, Scene


class ParallelRenderer:
    """Replaces ParallelRenderer::new (mod.rs:73-153): binds one MI355X instead of
    spawning one worker per logical CPU.  One instance per GPU / per process."""

    def __init__(self, device=0):
        self._ctx = C.c_void_p()
        rc = _abi.lib().pt_create(int(device), C.byref(self._ctx))
        if rc != _abi.PT_OK:
            msg = _abi.lib().pt_last_error(None)
            raise _abi.PtError(rc, msg.decode() if msg else "pt_create failed")
        self.device = int(device)
        self._raytracer = None

    @classmethod
    def new(cls, device=0):
        return cls(device)

    def close(self):
        if self._ctx:
            _abi.lib().pt_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def device_info(self):
        info = _abi.PtDeviceInfo()
        _abi.check(self._ctx, _abi.lib().pt_query(self._ctx, C.byref(info)))
        return {"name": info.name.decode(), "arch": info.arch.decode(), "compute_units": info.compute_units,
                "wavefront_size": info.wavefront_size, "clock_khz": info.clock_khz,
                "memory_clock_khz": info.memory_clock_khz, "l2_bytes": info.l2_bytes, "hbm_bytes": info.hbm_bytes}

    # -- scene binding --------------------------------------------------------------
    def bind(self, raytracer):
        """Upload the raytracer's scene + camera (the work Raytracer::new does once,
        raytracer.rs:36-44, plus the H2D copy)."""
        if self._raytracer is raytracer:
            return
        p = raytracer.packed
        tex = p.texture_structs()
        nodes = raytracer.nodes
        if hasattr(p, "prims"):          # extension scene (rects, lights, instances, integrator options)
            _abi.check(self._ctx, _abi.lib().pt_upload_scene_ext(
                self._ctx, p.prims, p.n_prims, p.materials, p.n_materials, tex, len(p.textures),
                p.transforms if len(p.transforms) else None, len(p.transforms),
                nodes, len(nodes) if nodes is not None else 0))
            _abi.check(self._ctx, _abi.lib().pt_set_integrator(
                self._ctx, C.byref(raytracer.integrator.pod) if raytracer.integrator is not None else None))
        else:
            _abi.check(self._ctx, _abi.lib().pt_upload_scene(
                self._ctx, p.spheres, p.n_spheres, p.materials, p.n_materials, tex, len(p.textures),
                nodes, len(nodes) if nodes is not None else 0))
        _abi.check(self._ctx, _abi.lib().pt_set_camera(self._ctx, C.byref(raytracer.camera.pod)))
        self._raytracer = raytracer

    def set_lanes_per_pixel(self, lanes):
        _abi.check(self._ctx, _abi.lib().pt_set_lanes_per_pixel(self._ctx, int(lanes)))

    def set_mode(self, fast):
        """pt_set_mode: False = PT_EXACT (default, bit parity with the oracle), True = PT_FAST (fused arithmetic,
        within the north-star tolerance)."""
        _abi.check(self._ctx, _abi.lib().pt_set_mode(self._ctx, _abi.PT_FAST if fast else _abi.PT_EXACT))

    def set_walk_order(self, view):
        """pt_set_walk_order: True = PT_WALK_VIEW (default: the child nearer to the camera first), False =
        PT_WALK_REFERENCE (left child first, bvh.rs:111-118).  Same results either way."""
        _abi.check(self._ctx, _abi.lib().pt_set_walk_order(self._ctx, _abi.PT_WALK_VIEW if view else _abi.PT_WALK_REFERENCE))

    def set_primary_lists(self, enabled):
        """pt_set_primary_lists: per-block candidate lists for camera rays (default on); off = every ray walks the tree"""
        _abi.check(self._ctx, _abi.lib().pt_set_primary_lists(self._ctx, 1 if enabled else 0))

    def primary_lists(self):
        """True if the bound scene + camera render with primary-ray candidate lists"""
        return _abi.lib().pt_get_primary_lists(self._ctx) == 1

    def walk_order(self):
        """the order in effect for the uploaded scene (scenes that can tie keep the reference's)"""
        return _abi.lib().pt_get_walk_order(self._ctx)

    def lanes_per_pixel(self, samples, sample_begin, sample_end):
        return _abi.lib().pt_get_lanes_per_pixel(self._ctx, samples, sample_begin, sample_end)

    # -- the render call (main.rs:110-114) ------------------------------------------
    def render(self, raytracer, buffer=None, seed=0, sample_range=None, radiance=False, hit_ids=False):
        """Fill ``buffer`` ((H*W, 3) or (H, W, 3) uint8, row-major from the top row:
        the ``buffer`` of main.rs:98) and return it.  ``radiance`` / ``hit_ids`` also
        return the fp32 radiance-sum and primary-hit-id AOVs:
        ``(buffer, radiance_or_None, hit_ids_or_None)``."""
        self.bind(raytracer)
        w, h, s = raytracer.width(), raytracer.height(), raytracer.samples
        if buffer is None:
            buffer = np.zeros((h, w, 3), dtype=np.uint8)
        if buffer.dtype != np.uint8 or buffer.size != w * h * 3 or not buffer.flags["C_CONTIGUOUS"]:
            raise ValueError("buffer must be a C-contiguous uint8 array of %d x %d pixels" % (w, h))
        s0, s1 = sample_range if sample_range is not None else (0, s * s)
        rad = np.zeros((h, w, 3), dtype=np.float32) if radiance else None
        ids = np.zeros((h, w), dtype=np.int32) if hit_ids else None
        _abi.check(self._ctx, _abi.lib().pt_render(
            self._ctx, s, int(seed) & (2 ** 64 - 1), int(s0), int(s1), buffer.ctypes.data,
            rad.ctypes.data if rad is not None else None, ids.ctypes.data if ids is not None else None))
        if radiance or hit_ids:
            return buffer, rad, ids
        return buffer

    def render_progress(self, raytracer, passes, progress=None, seed=0, radiance=False):
        """pt_render_progress: the frame in `passes` launches; ``progress(done, total)`` is called
        after each (return a true value to cancel -> PtError with code PT_ERR_CANCELLED)."""
        self.bind(raytracer)
        w, h, s = raytracer.width(), raytracer.height(), raytracer.samples
        buffer = np.zeros((h, w, 3), dtype=np.uint8)
        rad = np.zeros((h, w, 3), dtype=np.float32) if radiance else None
        cb = _abi.PROGRESS_FN(lambda _user, done, total: 1 if (progress and progress(done, total)) else 0)
        _abi.check(self._ctx, _abi.lib().pt_render_progress(
            self._ctx, s, int(seed) & (2 ** 64 - 1), int(passes), C.cast(cb, C.c_void_p), None, buffer.ctypes.data,
            rad.ctypes.data if rad is not None else None))
        return (buffer, rad) if radiance else buffer

    # -- device-resident halves (spp-sharded multi-GPU) -----------------------------
    def render_device(self, raytracer, d_radiance_ptr, seed=0, sample_range=None, accumulate=False, stream=None,
                      static_scratch=False):
        """pt_render_device; ``static_scratch`` = pt_render_device_static, the variant that allocates
        nothing (bind the raytracer beforehand: binding uploads, which allocates)."""
        self.bind(raytracer)
        s = raytracer.samples
        s0, s1 = sample_range if sample_range is not None else (0, s * s)
        fn = _abi.lib().pt_render_device_static if static_scratch else _abi.lib().pt_render_device
        _abi.check(self._ctx, fn(
            self._ctx, s, int(seed) & (2 ** 64 - 1), int(s0), int(s1), 1 if accumulate else 0,
            C.c_void_p(d_radiance_ptr), C.c_void_p(stream) if stream else None))

    def resolve_device(self, raytracer, d_radiance_ptr, d_rgb8_ptr, stream=None):
        self.bind(raytracer)
        _abi.check(self._ctx, _abi.lib().pt_resolve_device(
            self._ctx, raytracer.samples, C.c_void_p(d_radiance_ptr), C.c_void_p(d_rgb8_ptr),
            C.c_void_p(stream) if stream else None))

    def primary_hits_device(self, raytracer, d_hit_id_ptr, d_hit_t_ptr=None, sample_index=0, stream=None):
        self.bind(raytracer)
        _abi.check(self._ctx, _abi.lib().pt_primary_hits_device(
            self._ctx, raytracer.samples, int(sample_index), C.c_void_p(d_hit_id_ptr),
            C.c_void_p(d_hit_t_ptr) if d_hit_t_ptr else None, C.c_void_p(stream) if stream else None))

    def last_path_kernel_ms(self):
        ms = float(_abi.lib().pt_last_path_kernel_ms(self._ctx))
        if ms == -2.0:
            _abi.check(self._ctx, -2)        # PT_ERR_HIP with the kernel's message
        return ms

    def last_path_kernel_clock_ghz(self):
        """pt_last_path_kernel_clock_ghz: the clock the last persistent path-kernel launch ran at (-1: none / small-tree launch)"""
        return float(_abi.lib().pt_last_path_kernel_clock_ghz(self._ctx))

    def clock_probe(self):
        """pt_clock_probe: the GPU's shader clock right now, in GHz (a 2 ms chip-filling multiply chain)."""
        ghz = C.c_float(0.0)
        _abi.check(self._ctx, _abi.lib().pt_clock_probe(self._ctx, C.byref(ghz)))
        return float(ghz.value)

    def selftest_grid(self, samples):
        """pt_selftest_grid for the bound camera: sample-grid coordinates that differ from the
        reference's IEEE divisions (must be 0)."""
        n = C.c_uint64(0)
        _abi.check(self._ctx, _abi.lib().pt_selftest_grid(self._ctx, int(samples), C.byref(n)))
        return int(n.value)

    def selftest_division(self, log2_pairs=34):
        """pt_selftest_division: operand pairs (of 2^log2_pairs) on which the reciprocal-based
        division of Sphere::trace differs from the IEEE division."""
        n = C.c_uint64(0)
        _abi.check(self._ctx, _abi.lib().pt_selftest_division(self._ctx, int(log2_pairs), C.byref(n)))
        return int(n.value)

    def selftest_unary(self, fn):
        """pt_selftest_unary: inputs (of all 2^32) on which short form `fn` (0: 1/x, 1: sqrt,
        2: 1/sqrt) differs from the IEEE operation."""
        n = C.c_uint64(0)
        _abi.check(self._ctx, _abi.lib().pt_selftest_unary(self._ctx, int(fn), C.byref(n)))
        return int(n.value)


class RendererGroup:
    """Several GPUs of one node in ONE process (include/pt_amd.h, pt_group_*): the samples x samples grid
    is shared out over the GPUs, one RCCL reduce merges the fp32 radiance buffers on the first GPU,
    which resolves.  No counterpart in the reference (shared-memory threads only)."""

    def __init__(self, devices):
        devices = [int(d) for d in devices]
        self._grp = C.c_void_p()
        arr = (C.c_int32 * len(devices))(*devices)
        rc = _abi.lib().pt_group_create(arr, len(devices), C.byref(self._grp))
        if rc != _abi.PT_OK:
            msg = _abi.lib().pt_group_last_error(None)
            raise _abi.PtError(rc, msg.decode() if msg else "pt_group_create failed")
        self.devices = devices
        self._raytracer = None

    @classmethod
    def new(cls, devices):
        return cls(devices)

    def _check(self, rc):
        if rc != _abi.PT_OK:
            msg = _abi.lib().pt_group_last_error(self._grp)
            raise _abi.PtError(rc, msg.decode() if msg else "unknown error")

    def close(self):
        if self._grp:
            _abi.lib().pt_group_destroy(self._grp)
            self._grp = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def size(self):
        return _abi.lib().pt_group_size(self._grp)

    def set_lanes_per_pixel(self, lanes):
        for r in range(self.size()):
            ctx = C.c_void_p(_abi.lib().pt_group_context(self._grp, r))
            _abi.check(ctx, _abi.lib().pt_set_lanes_per_pixel(ctx, int(lanes)))

    def bind(self, raytracer):
        if self._raytracer is raytracer:
            return
        p = raytracer.packed
        tex = p.texture_structs()
        nodes = raytracer.nodes
        if hasattr(p, "prims"):
            self._check(_abi.lib().pt_group_upload_scene_ext(
                self._grp, p.prims, p.n_prims, p.materials, p.n_materials, tex, len(p.textures),
                p.transforms if len(p.transforms) else None, len(p.transforms),
                nodes, len(nodes) if nodes is not None else 0))
            self._check(_abi.lib().pt_group_set_integrator(
                self._grp, C.byref(raytracer.integrator.pod) if raytracer.integrator is not None else None))
        else:
            self._check(_abi.lib().pt_group_upload_scene(
                self._grp, p.spheres, p.n_spheres, p.materials, p.n_materials, tex, len(p.textures),
                nodes, len(nodes) if nodes is not None else 0))
        self._check(_abi.lib().pt_group_set_camera(self._grp, C.byref(raytracer.camera.pod)))
        self._raytracer = raytracer

    def render(self, raytracer, buffer=None, seed=0, radiance=False):
        """The full frame over the group; returns ``buffer`` or ``(buffer, radiance)``."""
        self.bind(raytracer)
        w, h, s = raytracer.width(), raytracer.height(), raytracer.samples
        if buffer is None:
            buffer = np.zeros((h, w, 3), dtype=np.uint8)
        if buffer.dtype != np.uint8 or buffer.size != w * h * 3 or not buffer.flags["C_CONTIGUOUS"]:
            raise ValueError("buffer must be a C-contiguous uint8 array of %d x %d pixels" % (w, h))
        rad = np.zeros((h, w, 3), dtype=np.float32) if radiance else None
        self._check(_abi.lib().pt_group_render(self._grp, s, int(seed) & (2 ** 64 - 1), buffer.ctypes.data,
                                               rad.ctypes.data if rad is not None else None))
        return (buffer, rad) if radiance else buffer

    def context(self, rank):
        """the borrowed pt_context of GPU `rank` (pt_group_context)"""
        return C.c_void_p(_abi.lib().pt_group_context(self._grp, int(rank)))

    def set_timeout(self, milliseconds):
        """pt_group_set_timeout: longest wait of a render for its GPUs (0 = no limit)"""
        self._check(_abi.lib().pt_group_set_timeout(self._grp, int(milliseconds)))

    def rccl_version(self):
        """(compiled against, loaded) NCCL version codes (pt_group_rccl_version)"""
        a, b = C.c_int32(0), C.c_int32(0)
        self._check(_abi.lib().pt_group_rccl_version(self._grp, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def inject_fault(self, rank, kind):
        """pt_group_inject_fault (test hook, host side only)"""
        self._check(_abi.lib().pt_group_inject_fault(self._grp, int(rank), int(kind)))

    def last_path_kernel_clock_ghz(self, rank=0):
        return float(_abi.lib().pt_last_path_kernel_clock_ghz(self.context(rank)))

    def clock_probe(self, rank=0):
        ghz = C.c_float(0.0)
        ctx = self.context(rank)
        _abi.check(ctx, _abi.lib().pt_clock_probe(ctx, C.byref(ghz)))
        return float(ghz.value)

    def timings(self):
        """({rank: path-kernel ms}, reduce ms) of the last render."""
        n = self.size()
        k = (C.c_float * n)()
        red = C.c_float(-1.0)
        _abi.lib().pt_group_timings(self._grp, k, n, C.byref(red))
        return [float(x) for x in k], float(red.value)


def group_shard(samples, rank, n):
    """pt_group_shard: the share of GPU `rank` of `n` (equals distributed.shard_range)."""
    b, e = C.c_int32(0), C.c_int32(0)
    _abi.lib().pt_group_shard(int(samples), int(rank), int(n), C.byref(b), C.byref(e))
    return int(b.value), int(e.value)


class Raytracer:
    """raytracer.rs:29-44: camera + scene + samples; ``new`` builds the BVH
    (Scene::construct_bvh, scene/mod.rs:50-60) -- here with pt_bvh_build on the host,
    flattened for the GPU."""

    def __init__(self, camera, scene, samples, build_bvh=True, integrator=None):
        if not isinstance(camera, Camera) or not isinstance(scene, Scene):
            raise TypeError("Raytracer.new(camera: Camera, scene: Scene, samples)")
        if integrator is not None and not scene.is_ext():
            raise ValueError("integrator options apply to extension scenes only (Rect / DiffuseLight / Scene.force_ext)")
        self.integrator = integrator       # extension: scene.Integrator or None (= the reference's constants)
        if int(samples) < 2:
            # samples == 1 divides 0/0 in the sample grid (raytracer.rs:78); the library refuses it too
            raise ValueError("samples must be >= 2")
        self.camera, self.samples = camera, int(samples)
        self.packed = scene.pack()
        self.nodes = self.packed.build_bvh() if build_bvh else None   # None: the library builds it at upload

    @classmethod
    def new(cls, camera, scene, samples, integrator=None):
        return cls(camera, scene, samples, integrator=integrator)

    def width(self):
        return self.camera.width

    def height(self):
        return self.camera.height

    def pixel_count(self):
        return self.width() * self.height()
