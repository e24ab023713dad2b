// pt_host.cpp -- host-side (CPU) halves of the boundary: what the Rust host does
// before it calls render, restated in C++ so the C++/Python harnesses can feed the
// library the same inputs: BVH construction + flattening (src/scene/bvh.rs:13-91),
// Camera::new (src/math/camera.rs:14-34) and seeded versions of the two built-in
// scene generators (src/scene/generators.rs).  Pure CPU code; compiled with
// -ffp-contract=off so that every fp32 op rounds once, like the reference.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/pt_amd.h"

namespace {

struct Box { float mn[3], mx[3]; };

// _mm_min_ps / _mm_max_ps lane rule (vec.rs:50-51)
inline float lane_min(float a, float b) { return a < b ? a : b; }
inline float lane_max(float a, float b) { return a > b ? a : b; }

inline Box sphere_box(const pt_sphere &s)                                        // sphere.rs:71-76
{
    Box b;
    const float c[3] = {s.cx, s.cy, s.cz};
    for (int k = 0; k < 3; k++) { b.mn[k] = c[k] - s.radius; b.mx[k] = c[k] + s.radius; }
    return b;
}
inline Box join(const Box &a, const Box &b)                                      // aabb.rs:38-43
{
    Box r;
    for (int k = 0; k < 3; k++) { r.mn[k] = lane_min(a.mn[k], b.mn[k]); r.mx[k] = lane_max(a.mx[k], b.mx[k]); }
    return r;
}
inline float volume(const Box &b)                                                // aabb.rs:45-49
{
    const float x = b.mx[0] - b.mn[0], y = b.mx[1] - b.mn[1], z = b.mx[2] - b.mn[2];
    return std::fabs(x * y * z);
}

struct Item { int32_t prim; Box box; };

struct Builder {
    std::vector<Item> items;
    std::vector<Box> suffix;     // scratch: suffix[i] = join(items[i..end)) of the current range
    pt_bvh_node *out;
    int32_t next = 0;
    bool degenerate = false;     // a range whose every candidate split costs NaN: the reference panics (bvh.rs:19)

    // BvhNode::construct (bvh.rs:13-91) emitting DFS pre-order nodes directly.
    // The reference recomputes both side boxes for every candidate split
    // (O(n^2) per node); min/max are exact, so a prefix/suffix sweep gives the
    // same boxes, costs and split index in O(n).
    Box construct(int lo, int hi)
    {
        const int32_t me = next++;
        const int n = hi - lo;
        if (n == 1) {                                                            // bvh.rs:20-24
            const Item &it = items[lo];
            emit(me, it.box, next, it.prim);
            return it.box;
        }
        Box whole = items[lo].box;                                               // bvh.rs:35
        for (int i = lo + 1; i < hi; i++) whole = join(whole, items[i].box);
        int axis = 0;                                                            // bvh.rs:37-49
        float ext[3];
        for (int k = 0; k < 3; k++) ext[k] = whole.mx[k] - whole.mn[k];
        for (int k = 1; k < 3; k++) if (ext[k] > ext[axis]) axis = k;
        // bvh.rs:51-56.  sort_unstable_by leaves the order of equal keys
        // unspecified; ties here keep their current order (stable).
        std::stable_sort(items.begin() + lo, items.begin() + hi, [axis](const Item &a, const Item &b) {
            const float ca = (a.box.mn[axis] + a.box.mx[axis]) * 0.5f;           // aabb.rs:21-23
            const float cb = (b.box.mn[axis] + b.box.mx[axis]) * 0.5f;
            return ca < cb;
        });
        suffix[hi - 1] = items[hi - 1].box;
        for (int i = hi - 2; i > lo; i--) suffix[i] = join(items[i].box, suffix[i + 1]);
        const float wvol = volume(whole);
        int best = 0;                                                            // bvh.rs:58-80
        float lowest = 3.402823466e+38f;
        Box prefix = items[lo].box;
        for (int i = 1; i < n; i++) {
            if (i > 1) prefix = join(prefix, items[lo + i - 1].box);
            const float cl = (volume(prefix) / wvol) * (float)i;
            const float cr = (volume(suffix[lo + i]) / wvol) * (float)(n - i);
            const float cost = cl + cr;
            if (cost < lowest) { best = i; lowest = cost; }
        }
        // No candidate beat f32::MAX: every cost is NaN (zero-volume or non-finite bounds).  The
        // reference recurses into an empty left range and panics (bvh.rs:19); no tree is invented
        // here either -- the build is abandoned and pt_bvh_build returns PT_ERR_INVALID.
        if (best == 0) { degenerate = true; return whole; }
        const Box l = construct(lo, lo + best);                                  // bvh.rs:82-85
        if (degenerate) return whole;
        const Box r = construct(lo + best, hi);
        if (degenerate) return whole;
        const Box bounds = join(l, r);                                           // bvh.rs:86
        emit(me, bounds, next, -1);
        return bounds;
    }
    void emit(int32_t i, const Box &b, int32_t skip, int32_t prim)
    {
        for (int k = 0; k < 3; k++) { out[i].bmin[k] = b.mn[k]; out[i].bmax[k] = b.mx[k]; }
        out[i].skip = skip;
        out[i].prim = prim;
    }
};

// ---- extension (include/pt_amd.h): boxes of rects and instances ----
inline void rect_axes(int axis, int &ia, int &ib) { ia = axis == 0 ? 1 : 0; ib = axis == 2 ? 1 : 2; }
// object space -> world: rotation about y by +theta
inline void rot_to_world(const pt_transform &x, const float p[3], float w[3])
{
    w[0] = x.cos_theta * p[0] + x.sin_theta * p[2];
    w[1] = p[1];
    w[2] = x.cos_theta * p[2] - x.sin_theta * p[0];
}
// bounding_box() of a rect: padded by 0.0001 on its thin axis (a flat box has volume 0, which
// turns every split cost of bvh.rs:65 into NaN); an instance: the box of the 8 rotated corners, + offset
inline Box rect_box(const pt_primitive &p, const pt_transform *xforms)
{
    float lo[3], hi[3];
    int ia, ib;
    rect_axes(p.axis, ia, ib);
    lo[p.axis] = p.v[4] - 0.0001f; hi[p.axis] = p.v[4] + 0.0001f;
    lo[ia] = p.v[0]; hi[ia] = p.v[1];
    lo[ib] = p.v[2]; hi[ib] = p.v[3];
    Box b;
    if (p.transform < 0) {
        for (int k = 0; k < 3; k++) { b.mn[k] = lo[k]; b.mx[k] = hi[k]; }
        return b;
    }
    const pt_transform &x = xforms[p.transform];
    for (int c = 0; c < 8; c++) {
        const float corner[3] = {(c & 1) ? hi[0] : lo[0], (c & 2) ? hi[1] : lo[1], (c & 4) ? hi[2] : lo[2]};
        float w[3];
        rot_to_world(x, corner, w);
        for (int k = 0; k < 3; k++) {
            if (c == 0) { b.mn[k] = w[k]; b.mx[k] = w[k]; }
            else { b.mn[k] = lane_min(w[k], b.mn[k]); b.mx[k] = lane_max(w[k], b.mx[k]); }
        }
    }
    for (int k = 0; k < 3; k++) { b.mn[k] = b.mn[k] + x.offset[k]; b.mx[k] = b.mx[k] + x.offset[k]; }
    return b;
}

// rng.rs restated for the seeded generators
struct HostRng {
    uint64_t s;
    explicit HostRng(uint64_t seed) : s(seed) { for (int i = 0; i < 10; i++) raw(); }   // rng.rs:27-35
    uint64_t raw() { uint64_t x = s; x ^= x << 13; x ^= x >> 7; x ^= x << 17; s = x; return x; }  // rng.rs:50-60
    float f32() { const uint32_t v = (uint32_t)raw(); return (float)(v >> 8) / 16777216.0f; }    // rng.rs:76-80
    float range(float mn, float mx) { return f32() * (mx - mn) + mn; }                            // rng.rs:83-85
};

inline float host_rsq(float x) { return 1.0f / std::sqrt(x); }   // portable RSQ

}  // namespace

extern "C" {

int32_t pt_bvh_build(const pt_sphere *spheres, int32_t n_spheres, pt_bvh_node *out, int32_t capacity)
{
    if (!spheres || !out || n_spheres < 1) return PT_ERR_INVALID;
    const int64_t need = 2ll * n_spheres - 1;
    if (capacity < need) return PT_ERR_LIMIT;
    // nothing may unwind across the C ABI: resize and std::stable_sort (which takes a buffer) can throw
    try {
        Builder b;
        b.items.resize((size_t)n_spheres);
        b.suffix.resize((size_t)n_spheres);
        b.out = out;
        for (int32_t i = 0; i < n_spheres; i++) { b.items[i].prim = i; b.items[i].box = sphere_box(spheres[i]); }
        b.construct(0, n_spheres);
        if (b.degenerate) return PT_ERR_INVALID;
        return b.next;
    } catch (...) {
        return PT_ERR_LIMIT;   // out of host memory
    }
}

int32_t pt_bvh_build_ext(const pt_primitive *prims, int32_t n_prims, const pt_transform *transforms, int32_t n_transforms,
                         pt_bvh_node *out, int32_t capacity)
{
    if (!prims || !out || n_prims < 1 || n_transforms < 0 || (n_transforms > 0 && !transforms)) return PT_ERR_INVALID;
    const int64_t need = 2ll * n_prims - 1;
    if (capacity < need) return PT_ERR_LIMIT;
    for (int32_t i = 0; i < n_prims; i++) {
        const pt_primitive &p = prims[i];
        if (p.kind == PT_PRIM_RECT) {
            if (p.axis < 0 || p.axis > 2 || p.transform < -1 || p.transform >= n_transforms) return PT_ERR_INVALID;
        } else if (p.kind != PT_PRIM_SPHERE || p.transform != -1) {
            return PT_ERR_INVALID;
        }
    }
    try {
        Builder b;
        b.items.resize((size_t)n_prims);
        b.suffix.resize((size_t)n_prims);
        b.out = out;
        for (int32_t i = 0; i < n_prims; i++) {
            const pt_primitive &p = prims[i];
            b.items[i].prim = i;
            if (p.kind == PT_PRIM_RECT) b.items[i].box = rect_box(p, transforms);
            else b.items[i].box = sphere_box(pt_sphere{p.v[0], p.v[1], p.v[2], p.v[3], p.material});
        }
        b.construct(0, n_prims);
        if (b.degenerate) return PT_ERR_INVALID;
        return b.next;
    } catch (...) {
        return PT_ERR_LIMIT;
    }
}

// "Ray Tracing: The Next Week", cornell_box(): five walls of the 555-unit room, the light under the
// ceiling, two boxes (six rects each) as Translate(RotateY(box)) instances.
int32_t pt_gen_cornell_box(pt_primitive *prims, int32_t prim_capacity, pt_material *materials, int32_t material_capacity,
                           pt_transform *transforms, int32_t transform_capacity)
{
    if (!prims || !materials || !transforms) return PT_ERR_INVALID;
    if (prim_capacity < 18 || material_capacity < 4 || transform_capacity < 2) return PT_ERR_LIMIT;
    enum { RED = 0, WHITE = 1, GREEN = 2, LIGHT = 3 };
    materials[RED] = pt_material{PT_LAMBERTIAN, {0.65f, 0.05f, 0.05f}, 0.0f, -1};
    materials[WHITE] = pt_material{PT_LAMBERTIAN, {0.73f, 0.73f, 0.73f}, 0.0f, -1};
    materials[GREEN] = pt_material{PT_LAMBERTIAN, {0.12f, 0.45f, 0.15f}, 0.0f, -1};
    materials[LIGHT] = pt_material{PT_DIFFUSE_LIGHT, {15.0f, 15.0f, 15.0f}, 0.0f, -1};
    int32_t n = 0;
    auto rect = [&](int axis, float a0, float a1, float b0, float b1, float k, int material, int flip, int transform) {
        prims[n++] = pt_primitive{PT_PRIM_RECT, material, transform, axis, flip, {a0, a1, b0, b1, k}};
    };
    rect(0, 0.0f, 555.0f, 0.0f, 555.0f, 555.0f, GREEN, 1, -1);          // flip_normals(yz_rect(0,555,0,555,555, green))
    rect(0, 0.0f, 555.0f, 0.0f, 555.0f, 0.0f, RED, 0, -1);              // yz_rect(0,555,0,555,0, red)
    rect(1, 213.0f, 343.0f, 227.0f, 332.0f, 554.0f, LIGHT, 0, -1);      // xz_rect(213,343,227,332,554, light)
    rect(1, 0.0f, 555.0f, 0.0f, 555.0f, 555.0f, WHITE, 1, -1);          // flip_normals(xz_rect(.., 555, white))
    rect(1, 0.0f, 555.0f, 0.0f, 555.0f, 0.0f, WHITE, 0, -1);            // xz_rect(.., 0, white)
    rect(2, 0.0f, 555.0f, 0.0f, 555.0f, 555.0f, WHITE, 1, -1);          // flip_normals(xy_rect(.., 555, white))
    // box(p0, p1): xy at z1, flipped xy at z0, xz at y1, flipped xz at y0, yz at x1, flipped yz at x0
    auto box = [&](float x1, float y1, float z1, int transform) {
        rect(2, 0.0f, x1, 0.0f, y1, z1, WHITE, 0, transform);
        rect(2, 0.0f, x1, 0.0f, y1, 0.0f, WHITE, 1, transform);
        rect(1, 0.0f, x1, 0.0f, z1, y1, WHITE, 0, transform);
        rect(1, 0.0f, x1, 0.0f, z1, 0.0f, WHITE, 1, transform);
        rect(0, 0.0f, y1, 0.0f, z1, x1, WHITE, 0, transform);
        rect(0, 0.0f, y1, 0.0f, z1, 0.0f, WHITE, 1, transform);
    };
    const float deg = 3.14159265358979323846f / 180.0f;
    transforms[0] = pt_transform{{130.0f, 0.0f, 65.0f}, std::sin(-18.0f * deg), std::cos(-18.0f * deg)};
    transforms[1] = pt_transform{{265.0f, 0.0f, 295.0f}, std::sin(15.0f * deg), std::cos(15.0f * deg)};
    box(165.0f, 165.0f, 165.0f, 0);
    box(165.0f, 330.0f, 165.0f, 1);
    return n;
}

int pt_camera_look_at(const float eye[3], const float target[3], const float up[3], float fov_degrees, int32_t width,
                      int32_t height, pt_camera *out)
{
    if (!eye || !target || !up || !out || width < 1 || height < 1) return PT_ERR_INVALID;
    const float aspect = (float)width / (float)height;                           // camera.rs:15
    const float radians = fov_degrees * (3.14159265358979323846f / 180.0f);      // f32::to_radians
    const float half_height = std::tan(radians / 2.0f);                          // camera.rs:17
    const float half_width = half_height * aspect;
    float w[3], u[3], v[3];
    for (int k = 0; k < 3; k++) w[k] = eye[k] - target[k];                       // camera.rs:20
    float r = host_rsq((w[0] * w[0] + w[1] * w[1]) + w[2] * w[2]);
    for (int k = 0; k < 3; k++) w[k] = w[k] * r;
    u[0] = up[1] * w[2] - up[2] * w[1];                                          // vec.rs:103-112
    u[1] = -(up[0] * w[2] - up[2] * w[0]);
    u[2] = up[0] * w[1] - up[1] * w[0];
    r = host_rsq((u[0] * u[0] + u[1] * u[1]) + u[2] * u[2]);                     // camera.rs:21
    for (int k = 0; k < 3; k++) u[k] = u[k] * r;
    v[0] = w[1] * u[2] - w[2] * u[1];                                            // camera.rs:22
    v[1] = -(w[0] * u[2] - w[2] * u[0]);
    v[2] = w[0] * u[1] - w[1] * u[0];
    for (int k = 0; k < 3; k++) {
        out->lower_left_corner[k] = ((eye[k] - u[k] * half_width) - v[k] * half_height) - w[k];   // camera.rs:24
        out->origin[k] = eye[k];
        out->horizontal[k] = (u[k] * 2.0f) * half_width;                         // camera.rs:31
        out->vertical[k] = (v[k] * 2.0f) * half_height;                          // camera.rs:32
    }
    out->width = width;
    out->height = height;
    return PT_OK;
}

int32_t pt_gen_simple_scene(pt_sphere *spheres, pt_material *materials)
{
    if (!spheres || !materials) return PT_ERR_INVALID;
    // generators.rs:9-10,14-16,21 (materials) and :11-12,17-19,22 (spheres)
    const pt_material m[6] = {
        {PT_LAMBERTIAN, {0.0f, 0.0f, 0.0f}, 0.0f, 0},   {PT_LAMBERTIAN, {0.3f, 0.0f, 0.0f}, 0.0f, -1},
        {PT_METAL, {0.8f, 0.6f, 0.2f}, 0.0f, -1},       {PT_DIELECTRIC, {0.0f, 0.0f, 0.0f}, 1.8f, -1},
        {PT_DIELECTRIC, {0.0f, 0.0f, 0.0f}, 0.4f, -1},  {PT_METAL, {0.1f, 1.0f, 0.7f}, 0.1f, -1},
    };
    const pt_sphere s[6] = {
        {0.0f, 0.0f, -1.0f, 0.5f, 0},  {0.0f, -100.5f, -1.0f, 100.0f, 1}, {1.5f, 0.0f, -2.0f, 0.5f, 2},
        {-1.5f, 0.0f, -2.0f, 0.5f, 3}, {3.5f, 0.0f, -2.0f, 0.8f, 4},      {10.0f, 0.0f, -10.0f, 3.0f, 5},
    };
    std::memcpy(materials, m, sizeof m);
    std::memcpy(spheres, s, sizeof s);
    return 6;
}

int32_t pt_gen_random_scene(uint64_t scene_seed, int32_t a0, int32_t a1, int32_t b0, int32_t b1, pt_sphere *spheres,
                            pt_material *materials, int32_t capacity)
{
    if (capacity > 0 && (!spheres || !materials)) return PT_ERR_INVALID;
    int32_t n = 0;
    auto push = [&](float cx, float cy, float cz, float radius, int32_t kind, float r, float g, float b, float param) {
        if (n < capacity) {
            spheres[n] = pt_sphere{cx, cy, cz, radius, n};
            materials[n] = pt_material{kind, {r, g, b}, param, -1};
        }
        n++;
    };
    push(0.0f, -1000.0f, 0.0f, 1000.0f, PT_LAMBERTIAN, 0.5f, 0.5f, 0.5f, 0.0f);  // generators.rs:26-30
    HostRng rng(scene_seed);                                                     // generators.rs:32 (seeded)
    for (int32_t a = a0; a < a1; a++) {                                          // generators.rs:42-43
        for (int32_t b = b0; b < b1; b++) {
            const float cx = (float)a + 0.9f * rng.f32();                        // generators.rs:44-48
            const float cz = (float)b + 0.9f * rng.f32();
            const float dx = cx - 3.0f, dy = 0.2f - 0.2f, dz = cz - 0.0f;        // generators.rs:50
            if (std::sqrt((dx * dx + dy * dy) + dz * dz) > 0.9f) {
                const float choose = rng.f32();
                if (choose < 0.8f) {                                             // generators.rs:53-57
                    float c0[3], c1[3];
                    for (float &c : c0) c = rng.range(0.0f, 1.0f);
                    for (float &c : c1) c = rng.range(0.0f, 1.0f);
                    push(cx, 0.2f, cz, 0.2f, PT_LAMBERTIAN, c0[0] * c1[0], c0[1] * c1[1], c0[2] * c1[2], 0.0f);
                } else if (choose < 0.95f) {                                     // generators.rs:58-62
                    float c0[3];
                    for (float &c : c0) c = rng.range(0.5f, 1.0f);
                    const float fuzz = rng.range(0.0f, 0.5f);
                    push(cx, 0.2f, cz, 0.2f, PT_METAL, c0[0], c0[1], c0[2], fuzz);
                } else {                                                         // generators.rs:63-65
                    push(cx, 0.2f, cz, 0.2f, PT_DIELECTRIC, 0.0f, 0.0f, 0.0f, 1.5f);
                }
            }
        }
    }
    push(0.0f, 1.0f, 0.0f, 1.0f, PT_DIELECTRIC, 0.0f, 0.0f, 0.0f, 1.5f);         // generators.rs:72-76
    push(-4.0f, 1.0f, 0.0f, 1.0f, PT_LAMBERTIAN, 0.4f, 0.2f, 0.1f, 0.0f);        // generators.rs:78-82
    push(4.0f, 1.0f, 0.0f, 1.0f, PT_METAL, 0.7f, 0.6f, 0.5f, 0.0f);              // generators.rs:84-88
    return n <= capacity ? n : -n;
}

}  // extern "C"
