"""ctypes binding of libpt_amd.so (include/pt_amd.h).

Loading fails loudly: there is no CPU fallback anywhere in this package.  The
library is built in-tree by ``__graft_entry__.build()`` (``make -C
path_tracer_amd/csrc``) and travels to the GPU box next to this file.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PT_AMD_LIB: alternative build of the same library (kernel tuning experiments only)
LIB_PATH = os.environ.get("PT_AMD_LIB") or os.path.join(_HERE, "libpt_amd.so")

PT_OK = 0
PT_ERR_CANCELLED = -6
PROGRESS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_int32)   # pt_progress_fn
PT_LAMBERTIAN, PT_METAL, PT_DIELECTRIC = 0, 1, 2
PT_DIFFUSE_LIGHT = 3                       # extension (include/pt_amd.h)
PT_PRIM_SPHERE, PT_PRIM_RECT = 0, 1
PT_EXACT, PT_FAST = 0, 1
PT_WALK_REFERENCE, PT_WALK_VIEW = 0, 1


class PtSphere(C.Structure):
    _fields_ = [("cx", C.c_float), ("cy", C.c_float), ("cz", C.c_float), ("radius", C.c_float),
                ("material", C.c_int32)]


class PtMaterial(C.Structure):
    _fields_ = [("kind", C.c_int32), ("albedo", C.c_float * 3), ("param", C.c_float), ("texture", C.c_int32)]


class PtTexture(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("rgb8", C.c_void_p)]


class PtCamera(C.Structure):
    _fields_ = [("lower_left_corner", C.c_float * 3), ("origin", C.c_float * 3), ("horizontal", C.c_float * 3),
                ("vertical", C.c_float * 3), ("width", C.c_int32), ("height", C.c_int32)]


class PtBvhNode(C.Structure):
    _fields_ = [("bmin", C.c_float * 3), ("skip", C.c_int32), ("bmax", C.c_float * 3), ("prim", C.c_int32)]


class PtPrimitive(C.Structure):
    _fields_ = [("kind", C.c_int32), ("material", C.c_int32), ("transform", C.c_int32), ("axis", C.c_int32),
                ("flip", C.c_int32), ("v", C.c_float * 5)]


class PtTransform(C.Structure):
    _fields_ = [("offset", C.c_float * 3), ("sin_theta", C.c_float), ("cos_theta", C.c_float)]


class PtIntegrator(C.Structure):
    _fields_ = [("max_bounces", C.c_int32), ("background_kind", C.c_int32), ("background", C.c_float * 3),
                ("rr_start", C.c_int32)]


class PtDeviceInfo(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("arch", C.c_char * 32), ("compute_units", C.c_int32),
                ("wavefront_size", C.c_int32), ("clock_khz", C.c_int32), ("memory_clock_khz", C.c_int32),
                ("l2_bytes", C.c_int32), ("hbm_bytes", C.c_int64)]


# name -> (restype, argtypes): every symbol include/pt_amd.h declares
SIGNATURES = {
    "pt_create": (C.c_int, [C.c_int32, C.POINTER(C.c_void_p)]),
    "pt_destroy": (None, [C.c_void_p]),
    "pt_last_error": (C.c_char_p, [C.c_void_p]),
    "pt_query": (C.c_int, [C.c_void_p, C.POINTER(PtDeviceInfo)]),
    "pt_upload_scene": (C.c_int, [C.c_void_p, C.POINTER(PtSphere), C.c_int32, C.POINTER(PtMaterial), C.c_int32,
                                  C.POINTER(PtTexture), C.c_int32, C.POINTER(PtBvhNode), C.c_int32]),
    "pt_set_camera": (C.c_int, [C.c_void_p, C.POINTER(PtCamera)]),
    "pt_render": (C.c_int, [C.c_void_p, C.c_int32, C.c_uint64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                            C.c_void_p]),
    "pt_render_device": (C.c_int, [C.c_void_p, C.c_int32, C.c_uint64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                   C.c_void_p]),
    "pt_render_device_static": (C.c_int, [C.c_void_p, C.c_int32, C.c_uint64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                          C.c_void_p]),
    "pt_render_progress": (C.c_int, [C.c_void_p, C.c_int32, C.c_uint64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p]),
    "pt_resolve_device": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pt_primary_hits_device": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pt_last_path_kernel_ms": (C.c_float, [C.c_void_p]),
    "pt_set_mode": (C.c_int, [C.c_void_p, C.c_int32]),
    "pt_set_walk_order": (C.c_int, [C.c_void_p, C.c_int32]),
    "pt_get_walk_order": (C.c_int, [C.c_void_p]),
    "pt_set_primary_lists": (C.c_int, [C.c_void_p, C.c_int32]),
    "pt_get_primary_lists": (C.c_int, [C.c_void_p]),
    "pt_last_path_kernel_clock_ghz": (C.c_float, [C.c_void_p]),
    "pt_clock_probe": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "pt_set_lanes_per_pixel": (C.c_int, [C.c_void_p, C.c_int32]),
    "pt_get_lanes_per_pixel": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "pt_selftest_unary": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_uint64)]),
    "pt_selftest_grid": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_uint64)]),
    "pt_selftest_division": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_uint64)]),
    "pt_upload_scene_ext": (C.c_int, [C.c_void_p, C.POINTER(PtPrimitive), C.c_int32, C.POINTER(PtMaterial), C.c_int32,
                                      C.POINTER(PtTexture), C.c_int32, C.POINTER(PtTransform), C.c_int32,
                                      C.POINTER(PtBvhNode), C.c_int32]),
    "pt_set_integrator": (C.c_int, [C.c_void_p, C.POINTER(PtIntegrator)]),
    "pt_bvh_build_ext": (C.c_int32, [C.POINTER(PtPrimitive), C.c_int32, C.POINTER(PtTransform), C.c_int32,
                                     C.POINTER(PtBvhNode), C.c_int32]),
    "pt_gen_cornell_box": (C.c_int32, [C.POINTER(PtPrimitive), C.c_int32, C.POINTER(PtMaterial), C.c_int32,
                                       C.POINTER(PtTransform), C.c_int32]),
    "pt_group_upload_scene_ext": (C.c_int, [C.c_void_p, C.POINTER(PtPrimitive), C.c_int32, C.POINTER(PtMaterial), C.c_int32,
                                            C.POINTER(PtTexture), C.c_int32, C.POINTER(PtTransform), C.c_int32,
                                            C.POINTER(PtBvhNode), C.c_int32]),
    "pt_group_set_integrator": (C.c_int, [C.c_void_p, C.POINTER(PtIntegrator)]),
    "pt_group_create": (C.c_int, [C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_void_p)]),
    "pt_group_destroy": (None, [C.c_void_p]),
    "pt_group_last_error": (C.c_char_p, [C.c_void_p]),
    "pt_group_size": (C.c_int32, [C.c_void_p]),
    "pt_group_context": (C.c_void_p, [C.c_void_p, C.c_int32]),
    "pt_group_upload_scene": (C.c_int, [C.c_void_p, C.POINTER(PtSphere), C.c_int32, C.POINTER(PtMaterial), C.c_int32,
                                        C.POINTER(PtTexture), C.c_int32, C.POINTER(PtBvhNode), C.c_int32]),
    "pt_group_set_camera": (C.c_int, [C.c_void_p, C.POINTER(PtCamera)]),
    "pt_group_render": (C.c_int, [C.c_void_p, C.c_int32, C.c_uint64, C.c_void_p, C.c_void_p]),
    "pt_group_set_timeout": (C.c_int, [C.c_void_p, C.c_int32]),
    "pt_group_rccl_version": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "pt_group_inject_fault": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "pt_group_shard": (None, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "pt_group_timings": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_int32, C.POINTER(C.c_float)]),
    "pt_bvh_build": (C.c_int32, [C.POINTER(PtSphere), C.c_int32, C.POINTER(PtBvhNode), C.c_int32]),
    "pt_camera_look_at": (C.c_int, [C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float,
                                    C.c_int32, C.c_int32, C.POINTER(PtCamera)]),
    "pt_gen_simple_scene": (C.c_int32, [C.POINTER(PtSphere), C.POINTER(PtMaterial)]),
    "pt_gen_random_scene": (C.c_int32, [C.c_uint64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(PtSphere),
                                        C.POINTER(PtMaterial), C.c_int32]),
}

_lib = None


class PtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("libpt_amd: %s (status %d)" % (message, code))
        self.code = code


def lib():
    """The loaded library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
        try:
            # One HIP runtime per process: when PyTorch is installed, let it load its own
            # libamdhip64 first so that torch tensors / streams and this library share it
            # (loading /opt/rocm's copy first leaves torch with "No HIP GPUs are available").
            import torch  # noqa: F401
        except ImportError:
            pass
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)   # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(ctx, rc):
    if rc != PT_OK:
        msg = lib().pt_last_error(ctx)
        raise PtError(rc, msg.decode() if msg else "unknown error")
