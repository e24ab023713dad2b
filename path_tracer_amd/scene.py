"""Host-side mirror of the reference's scene builder API, on top of the C ABI.

Same names and argument meaning as the Rust host keeps:
``Scene::{new,add}`` (src/scene/mod.rs:17,46), ``Sphere::new``
(src/traceable/sphere.rs:27), ``Lambertian::{new,new_solid}``
(src/material/lambertian.rs:18,24), ``Metal::new`` (metal.rs:13),
``Dielectric::new`` (dielectric.rs:12), ``SolidTexture::new`` (texture/solid.rs:9),
``PictureTexture::new`` (texture/picture.rs:9), ``Camera::new``
(src/math/camera.rs:14) and ``scene::generators`` (src/scene/generators.rs).
The objects only record what the builders were given; ``Scene.pack()`` turns them
into the plain arrays that cross the ABI (what the Rust shim of INTEGRATION.md
does by walking ``Arc<dyn Material>`` / ``BvhNode``).
"""
import ctypes as C
import os

import numpy as np

from . import _abi

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
EARTHMAP_PNG = os.path.join(_DATA, "earthmap_rgb8.png")


def _f3(v):
    a = np.asarray(v, dtype=np.float32).reshape(3)
    return (float(a[0]), float(a[1]), float(a[2]))


class SolidTexture:
    """texture/solid.rs: a constant colour."""

    def __init__(self, color):
        self.color = _f3(color)

    @classmethod
    def new(cls, color):
        return cls(color)


class PictureTexture:
    """texture/picture.rs: decoded RGB8 image, row-major from the top row.

    ``new(path)`` decodes with PIL (the reference uses image::open(..).into_rgb(),
    picture.rs:11); ``earthmap()`` loads the repository's pre-decoded texels of the
    reference's earthmap.jpg; ``from_rgb8`` wraps an (H, W, 3) uint8 array.
    """

    def __init__(self, rgb8):
        a = np.ascontiguousarray(rgb8, dtype=np.uint8)
        if a.ndim != 3 or a.shape[2] != 3 or a.shape[0] < 1 or a.shape[1] < 1:
            raise ValueError("texture must be (H, W, 3) uint8")
        self.rgb8 = a

    @classmethod
    def new(cls, path):
        from PIL import Image
        return cls(np.asarray(Image.open(path).convert("RGB"), dtype=np.uint8))

    @classmethod
    def from_rgb8(cls, rgb8):
        return cls(rgb8)

    @classmethod
    def earthmap(cls):
        return cls.new(EARTHMAP_PNG)

    @property
    def width(self):
        return self.rgb8.shape[1]

    @property
    def height(self):
        return self.rgb8.shape[0]


class Lambertian:
    """material/lambertian.rs: albedo is a solid colour or a texture."""

    def __init__(self, albedo=None, texture=None):
        self.albedo, self.texture = albedo, texture

    @classmethod
    def new(cls, texture):
        if isinstance(texture, SolidTexture):   # Texture::color ignores (u, v): same as new_solid
            return cls(albedo=texture.color)
        if not isinstance(texture, PictureTexture):
            raise TypeError("Lambertian.new takes a SolidTexture or PictureTexture")
        return cls(texture=texture)

    @classmethod
    def new_solid(cls, albedo):
        return cls(albedo=_f3(albedo))


class Metal:
    """material/metal.rs"""

    def __init__(self, albedo, fuziness):
        self.albedo, self.fuziness = _f3(albedo), float(np.float32(fuziness))

    @classmethod
    def new(cls, albedo, fuziness):
        return cls(albedo, fuziness)


class Dielectric:
    """material/dielectric.rs"""

    def __init__(self, ref_idx):
        self.ref_idx = float(np.float32(ref_idx))

    @classmethod
    def new(cls, ref_idx):
        return cls(ref_idx)


class Sphere:
    """traceable/sphere.rs:20-34"""

    def __init__(self, center, radius, material):
        if not isinstance(material, (Lambertian, Metal, Dielectric)):
            raise TypeError("material must be Lambertian, Metal or Dielectric")
        self.center, self.radius, self.material = _f3(center), float(np.float32(radius)), material

    @classmethod
    def new(cls, center, radius, material):
        return cls(center, radius, material)


class PackedScene:
    """The plain arrays that cross the ABI (pt_sphere[], pt_material[], pt_texture[])."""

    def __init__(self, spheres, materials, textures):
        self.spheres, self.materials, self.textures = spheres, materials, textures   # ctypes arrays, list of ndarrays
        self.n_spheres, self.n_materials = len(spheres), len(materials)

    def texture_structs(self):
        arr = (_abi.PtTexture * max(1, len(self.textures)))()
        for i, t in enumerate(self.textures):
            arr[i].width, arr[i].height = t.shape[1], t.shape[0]
            arr[i].rgb8 = t.ctypes.data
        return arr

    def spheres_numpy(self):
        return np.array([(s.cx, s.cy, s.cz, s.radius, s.material) for s in self.spheres],
                        dtype=[("cx", "f4"), ("cy", "f4"), ("cz", "f4"), ("radius", "f4"), ("material", "i4")])

    def build_bvh(self):
        """pt_bvh_build: BvhNode::construct + pre-order flatten (bvh.rs:13-91)."""
        cap = 2 * self.n_spheres - 1
        nodes = (_abi.PtBvhNode * cap)()
        got = _abi.lib().pt_bvh_build(self.spheres, self.n_spheres, nodes, cap)
        if got != cap:
            raise _abi.PtError(got, "pt_bvh_build failed")
        return nodes


class Scene:
    """scene/mod.rs: an ordered object list; primitive id = insertion order (mod.rs:46-48)."""

    def __init__(self):
        self.objects = []

    @classmethod
    def new(cls):
        return cls()

    def add(self, obj):
        if not isinstance(obj, Sphere):
            raise TypeError("only Sphere implements Traceable (traceable/mod.rs)")
        self.objects.append(obj)

    def pack(self):
        if not self.objects:
            raise ValueError("Cannot build BVH without any objects.")   # bvh.rs:19
        mats, mat_index, texs, tex_index = [], {}, [], {}
        spheres = (_abi.PtSphere * len(self.objects))()
        for i, o in enumerate(self.objects):
            m = o.material
            if id(m) not in mat_index:                       # shared Arc<dyn Material>: one entry
                mat_index[id(m)] = len(mats)
                mats.append(m)
            spheres[i] = _abi.PtSphere(o.center[0], o.center[1], o.center[2], o.radius, mat_index[id(m)])
        materials = (_abi.PtMaterial * len(mats))()
        for i, m in enumerate(mats):
            if isinstance(m, Lambertian):
                tex = -1
                if m.texture is not None:
                    if id(m.texture) not in tex_index:
                        tex_index[id(m.texture)] = len(texs)
                        texs.append(m.texture.rgb8)
                    tex = tex_index[id(m.texture)]
                alb = m.albedo if m.albedo is not None else (0.0, 0.0, 0.0)
                materials[i] = _abi.PtMaterial(_abi.PT_LAMBERTIAN, (C.c_float * 3)(*alb), 0.0, tex)
            elif isinstance(m, Metal):
                materials[i] = _abi.PtMaterial(_abi.PT_METAL, (C.c_float * 3)(*m.albedo), m.fuziness, -1)
            else:
                materials[i] = _abi.PtMaterial(_abi.PT_DIELECTRIC, (C.c_float * 3)(0.0, 0.0, 0.0), m.ref_idx, -1)
        return PackedScene(spheres, materials, texs)


class Camera:
    """math/camera.rs: only the derived fields exist after ``new`` (camera.rs:4-11)."""

    def __init__(self, pod):
        self.pod = pod

    @classmethod
    def new(cls, eyes, target, up, fov, width, height):
        pod = _abi.PtCamera()
        f3 = lambda v: (C.c_float * 3)(*_f3(v))
        rc = _abi.lib().pt_camera_look_at(f3(eyes), f3(target), f3(up), float(fov), int(width), int(height),
                                          C.byref(pod))
        if rc != _abi.PT_OK:
            raise _abi.PtError(rc, "pt_camera_look_at: bad arguments")
        return cls(pod)

    @property
    def width(self):
        return self.pod.width

    @property
    def height(self):
        return self.pod.height


# ---- scene::generators (src/scene/generators.rs) ------------------------------------

def _from_arrays(scene, spheres, materials, n, textures=()):
    mats = []
    for i in range(len(materials)):
        m = materials[i]
        if m.kind == _abi.PT_LAMBERTIAN:
            mats.append(Lambertian.new(textures[m.texture]) if m.texture >= 0 else Lambertian.new_solid(tuple(m.albedo)))
        elif m.kind == _abi.PT_METAL:
            mats.append(Metal.new(tuple(m.albedo), m.param))
        else:
            mats.append(Dielectric.new(m.param))
    for i in range(n):
        s = spheres[i]
        scene.add(Sphere.new((s.cx, s.cy, s.cz), s.radius, mats[s.material]))


def simple_scene(scene, earthmap=None):
    """generators.rs:8-23.  ``earthmap``: PictureTexture (default: the repository's decoded earthmap)."""
    spheres, materials = (_abi.PtSphere * 6)(), (_abi.PtMaterial * 6)()
    n = _abi.lib().pt_gen_simple_scene(spheres, materials)
    _from_arrays(scene, spheres, materials, n, textures=[earthmap or PictureTexture.earthmap()])


# the reference's loop bounds (generators.rs:42-43) and the RTOW-classic bounds used for
# BASELINE configs 1 and 3 (a stated deviation, ~480 spheres)
REFERENCE_BOUNDS = (-22, 22, -22, 221)
RTOW_BOUNDS = (-11, 11, -11, 11)


def random_scene(scene, seed, bounds=REFERENCE_BOUNDS):
    """generators.rs:25-89 with Rng::with_seed(seed) in place of Rng::new() (generators.rs:32)."""
    a0, a1, b0, b1 = bounds
    need = _abi.lib().pt_gen_random_scene(int(seed), a0, a1, b0, b1, None, None, 0)
    cap = -need
    spheres, materials = (_abi.PtSphere * cap)(), (_abi.PtMaterial * cap)()
    n = _abi.lib().pt_gen_random_scene(int(seed), a0, a1, b0, b1, spheres, materials, cap)
    if n != cap:
        raise _abi.PtError(n, "pt_gen_random_scene failed")
    _from_arrays(scene, spheres, materials, n)
