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


class DiffuseLight:
    """EXTENSION (not in the reference; include/pt_amd.h): a material whose
    ``scatter`` returns None and whose ``emitted`` term is ``color`` (e.g. (15, 15, 15))."""

    def __init__(self, color):
        self.color = _f3(color)

    @classmethod
    def new(cls, color):
        return cls(color)


class Transform:
    """EXTENSION: Translate(RotateY(object)) -- rotate about y by ``degrees``, then move by ``offset``.
    sin / cos are computed here (host libm) and cross the ABI as values."""

    def __init__(self, offset, degrees):
        self.offset = _f3(offset)
        rad = np.float32(degrees) * (np.float32(np.pi) / np.float32(180.0))
        self.sin_theta, self.cos_theta = float(np.sin(rad, dtype=np.float32)), float(np.cos(rad, dtype=np.float32))

    @classmethod
    def new(cls, offset, degrees):
        return cls(offset, degrees)

    @classmethod
    def from_sin_cos(cls, offset, sin_theta, cos_theta):
        t = cls.__new__(cls)
        t.offset, t.sin_theta, t.cos_theta = _f3(offset), float(np.float32(sin_theta)), float(np.float32(cos_theta))
        return t


class Rect:
    """EXTENSION: an axis-aligned rectangle, the plane ``axis`` = k (0: yz-rect, 1: xz-rect, 2: xy-rect)
    over [a0, a1] x [b0, b1] on the two other axes in increasing axis order; normal = the axis' unit
    vector, negated by ``flip``; optional ``transform``."""

    def __init__(self, axis, a0, a1, b0, b1, k, material, flip=False, transform=None):
        if axis not in (0, 1, 2):
            raise ValueError("axis must be 0, 1 or 2")
        if not isinstance(material, (Lambertian, Metal, Dielectric, DiffuseLight)):
            raise TypeError("material must be Lambertian, Metal, Dielectric or DiffuseLight")
        if transform is not None and not isinstance(transform, Transform):
            raise TypeError("transform must be a Transform")
        self.axis, self.flip, self.material, self.transform = int(axis), bool(flip), material, transform
        self.v = tuple(float(np.float32(x)) for x in (a0, a1, b0, b1, k))

    @classmethod
    def new(cls, *args, **kw):
        return cls(*args, **kw)


def box_rects(p0, p1, material, transform=None):
    """EXTENSION: the six rects of an axis-aligned box from p0 to p1 (outward normals)."""
    (x0, y0, z0), (x1, y1, z1) = _f3(p0), _f3(p1)
    return [Rect(2, x0, x1, y0, y1, z1, material, False, transform), Rect(2, x0, x1, y0, y1, z0, material, True, transform),
            Rect(1, x0, x1, z0, z1, y1, material, False, transform), Rect(1, x0, x1, z0, z1, y0, material, True, transform),
            Rect(0, y0, y1, z0, z1, x1, material, False, transform), Rect(0, y0, y1, z0, z1, x0, material, True, transform)]


class Integrator:
    """EXTENSION: trace_ray's constants as parameters (pt_integrator).  Defaults = the reference:
    5 bounces (raytracer.rs:48), the sky gradient (raytracer.rs:67-68), no Russian roulette."""

    def __init__(self, max_bounces=5, background=None, rr_start=-1):
        bg = _f3(background) if background is not None else (0.0, 0.0, 0.0)
        self.pod = _abi.PtIntegrator(int(max_bounces), 0 if background is None else 1, (C.c_float * 3)(*bg), int(rr_start))


class Sphere:
    """traceable/sphere.rs:20-34"""

    def __init__(self, center, radius, material):
        if not isinstance(material, (Lambertian, Metal, Dielectric, DiffuseLight)):
            raise TypeError("material must be Lambertian, Metal or Dielectric (or the extension's DiffuseLight)")
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


class PackedSceneExt:
    """EXTENSION: the plain arrays of pt_upload_scene_ext (pt_primitive[], pt_material[], pt_texture[], pt_transform[])."""

    def __init__(self, prims, materials, textures, transforms):
        self.prims, self.materials, self.textures, self.transforms = prims, materials, textures, transforms
        self.n_prims, self.n_materials = len(prims), len(materials)
        self.n_spheres = self.n_prims          # primitive count, under the name the benchmark reports

    texture_structs = PackedScene.texture_structs

    def build_bvh(self):
        cap = 2 * self.n_prims - 1
        nodes = (_abi.PtBvhNode * cap)()
        got = _abi.lib().pt_bvh_build_ext(self.prims, self.n_prims, self.transforms if len(self.transforms) else None,
                                          len(self.transforms), nodes, cap)
        if got != cap:
            raise _abi.PtError(got, "pt_bvh_build_ext failed")
        return nodes


class Scene:
    """scene/mod.rs: an ordered object list; primitive id = insertion order (mod.rs:46-48)."""

    def __init__(self):
        self.objects = []
        self.force_ext = False      # render a sphere-only scene through the extension's entry points (tests)

    @classmethod
    def new(cls):
        return cls()

    def add(self, obj):
        if not isinstance(obj, (Sphere, Rect)):
            raise TypeError("only Sphere implements Traceable (traceable/mod.rs); the extension adds Rect")
        self.objects.append(obj)

    def is_ext(self):
        return self.force_ext or any(isinstance(o, Rect) or isinstance(o.material, DiffuseLight) for o in self.objects)

    def pack(self):
        if not self.objects:
            raise ValueError("Cannot build BVH without any objects.")   # bvh.rs:19
        mats, mat_index, texs, tex_index = [], {}, [], {}
        ext = self.is_ext()
        spheres = None if ext else (_abi.PtSphere * len(self.objects))()
        prims = (_abi.PtPrimitive * len(self.objects))() if ext else None
        xforms, xform_index = [], {}
        for i, o in enumerate(self.objects):
            m = o.material
            if id(m) not in mat_index:                       # shared Arc<dyn Material>: one entry
                mat_index[id(m)] = len(mats)
                mats.append(m)
            if not ext:
                spheres[i] = _abi.PtSphere(o.center[0], o.center[1], o.center[2], o.radius, mat_index[id(m)])
            elif isinstance(o, Sphere):
                prims[i] = _abi.PtPrimitive(_abi.PT_PRIM_SPHERE, mat_index[id(m)], -1, 0, 0,
                                            (C.c_float * 5)(o.center[0], o.center[1], o.center[2], o.radius, 0.0))
            else:
                t = -1
                if o.transform is not None:
                    if id(o.transform) not in xform_index:
                        xform_index[id(o.transform)] = len(xforms)
                        xforms.append(o.transform)
                    t = xform_index[id(o.transform)]
                prims[i] = _abi.PtPrimitive(_abi.PT_PRIM_RECT, mat_index[id(m)], t, o.axis, 1 if o.flip else 0,
                                            (C.c_float * 5)(*o.v))
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
            elif isinstance(m, DiffuseLight):
                materials[i] = _abi.PtMaterial(_abi.PT_DIFFUSE_LIGHT, (C.c_float * 3)(*m.color), 0.0, -1)
            else:
                materials[i] = _abi.PtMaterial(_abi.PT_DIELECTRIC, (C.c_float * 3)(0.0, 0.0, 0.0), m.ref_idx, -1)
        if not ext:
            return PackedScene(spheres, materials, texs)
        transforms = (_abi.PtTransform * len(xforms))()
        for i, t in enumerate(xforms):
            transforms[i] = _abi.PtTransform((C.c_float * 3)(*t.offset), t.sin_theta, t.cos_theta)
        return PackedSceneExt(prims, materials, texs, transforms)


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


# The room of "Ray Tracing: The Next Week" (cornell_box()), camera of its main(): BASELINE config 4
CORNELL_CAMERA = dict(eyes=(278.0, 278.0, -800.0), target=(278.0, 278.0, 0.0), up=(0.0, 1.0, 0.0), fov=40.0)


def cornell_box(scene):
    """EXTENSION: pt_gen_cornell_box -- five walls, the light, two rotated boxes (18 rects)."""
    prims, materials, transforms = (_abi.PtPrimitive * 18)(), (_abi.PtMaterial * 4)(), (_abi.PtTransform * 2)()
    n = _abi.lib().pt_gen_cornell_box(prims, 18, materials, 4, transforms, 2)
    if n != 18:
        raise _abi.PtError(n, "pt_gen_cornell_box failed")
    mats = []
    for m in materials:
        mats.append(DiffuseLight.new(tuple(m.albedo)) if m.kind == _abi.PT_DIFFUSE_LIGHT else Lambertian.new_solid(tuple(m.albedo)))
    xf = [Transform.from_sin_cos(tuple(t.offset), t.sin_theta, t.cos_theta) for t in transforms]
    for p in prims:
        scene.add(Rect(p.axis, p.v[0], p.v[1], p.v[2], p.v[3], p.v[4], mats[p.material], bool(p.flip),
                       xf[p.transform] if p.transform >= 0 else None))
