/*
 * pt_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the radiance hot path of addrianyy/path-tracer
 * (reference tree mounted at /root/reference; every function below cites the
 * reference file:line it follows).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library, and only as the checker /
 * reported baseline -- the product (path_tracer_amd/) never links or calls it.
 *
 * PARITY STATUS: "parity unpinned".  The reference ships no tests, no golden
 * vectors and no fixtures for this path, and it cannot be
 * built here (Rust; no rustc/cargo in the image).  The oracle is therefore
 * pinned only by (i) hand-derived known-answer tests in tests/test_oracle_kat.py
 * and (ii) agreement with a second, independently written numpy-float32
 * restatement (oracle/np_oracle.py).
 *
 * Numeric contract: every + - * / sqrt is ONE correctly
 * rounded fp32 operation, never fused.  Build with -ffp-contract=off and
 * without -ffast-math (see oracle/Makefile).
 *
 * Three choices the reference leaves open are fixed here and mirrored by the
 * HIP library (DESIGN.md "Portable definitions"):
 *   P1  RSQ(x): the reference uses _mm_rsqrt_ps (src/math/vec.rs:38-44), whose
 *       bits differ between CPU models.  Portable definition: 1.0f / sqrtf(x).
 *       -DORA_RSQRT_SSE switches to the real rsqrtps (cost-faithful timing
 *       build only; never used for parity).
 *   P2  atan2f / asinf (src/traceable/sphere.rs:9-10) come from the platform
 *       libm in the reference.  Portable definition: a fixed fp64 algorithm
 *       (ora_atan2f / ora_asinf below) rounded once to fp32.
 *   P3  RNG streams: the reference seeds one stream per worker thread from
 *       rdseed (src/rng.rs:4-25, src/parallel_renderer/mod.rs:124), which is
 *       not reproducible.  Here every (pixel, sample) pair owns one stream whose
 *       state after construction is mix(seed, pixel, sample) (ora_path_stream
 *       below), i.e. the stream Rng::with_seed(ora_path_seed(...)): with_seed's
 *       ten rounds are a bijection on the nonzero states, so naming the state
 *       names the seed, and the rounds need not be run once per path.
 */
#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef ORA_RSQRT_SSE
#include <xmmintrin.h>
#endif

#define ORA_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ types */

typedef struct { float x, y, z; } v3;

/* Same POD layouts as include/pt_amd.h (kept textually separate on purpose). */
typedef struct { float cx, cy, cz, radius; int32_t material; } ora_sphere;
typedef struct { int32_t kind; float albedo[3]; float param; int32_t texture; } ora_material;
typedef struct { int32_t width, height; const uint8_t *rgb8; } ora_texture;
typedef struct {
    float llc[3], origin[3], horizontal[3], vertical[3];
    int32_t width, height;
} ora_camera;
/* pre-order dump of the reference's BvhNode tree (for cross-checking builders) */
typedef struct { float bmin[3]; int32_t skip; float bmax[3]; int32_t prim; } ora_node;

enum { ORA_LAMBERTIAN = 0, ORA_METAL = 1, ORA_DIELECTRIC = 2, ORA_DIFFUSE_LIGHT = 3 /* extension */ };

/* EXTENSION (not in the reference): rects, Translate(RotateY(..)) instances,
 * a diffuse light, trace_ray's constants as parameters.  Same POD layouts as include/pt_amd.h. */
enum { ORA_PRIM_SPHERE = 0, ORA_PRIM_RECT = 1 };
typedef struct { int32_t kind, material, transform, axis, flip; float v[5]; } ora_primitive;
typedef struct { float offset[3]; float sin_theta, cos_theta; } ora_transform;
typedef struct { int32_t max_bounces, background_kind; float background[3]; int32_t rr_start; } ora_integrator;

/* event counters for the algorithmic-flop model of SURVEY.md section 8(d) */
typedef struct {
    uint64_t paths, segments, aabb_tests, sphere_tests, sphere_near_hits,
        sphere_far_hits, lambert, lambert_textured, metal_scatter, metal_absorb,
        dielectric_refract, dielectric_tir, rng_draws, sky_evals,
        /* extension */ rect_tests, rect_hits, rect_transformed, light_hits, rr_draws, rect_in_range, rect_hits_transformed;
} ora_counters;

typedef struct bvh_node {
    int is_leaf;
    v3 bmin, bmax;
    int prim;                 /* leaf: insertion index (Scene::add order) */
    struct bvh_node *left, *right;
} bvh_node;

typedef struct {
    ora_sphere *spheres; int n_spheres;
    ora_material *materials; int n_materials;
    ora_texture *textures; int n_textures;
    bvh_node *root;
    int n_nodes;
    /* extension: set by ora_scene_create_ext only (ext == 0: the reference's path, untouched) */
    int ext;
    ora_primitive *prims; int n_prims;      /* spheres[i].material is filled for every primitive i */
    ora_transform *xforms; int n_xforms;
    ora_integrator integ;
} ora_scene;

/* ------------------------------------------------------- Vec3 (vec.rs) */

static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 vfill(float f) { return V(f, f, f); }                      /* vec.rs:79 */
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); } /* vec.rs:145 */
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); } /* vec.rs:153 */
static inline v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); } /* vec.rs:177 */
static inline v3 vdiv(v3 a, v3 b) { return V(a.x / b.x, a.y / b.y, a.z / b.z); } /* vec.rs:185 */
static inline v3 vmulf(v3 a, float f) { return vmul(a, vfill(f)); }         /* vec.rs:161-166 */
static inline v3 vdivf(v3 a, float f) { return vdiv(a, vfill(f)); }         /* vec.rs:169-174 */
static inline v3 vneg(v3 a) { return vmul(a, vfill(-1.0f)); }               /* vec.rs:137-142 */
/* vec.rs:32-36: sum = x + y + z, left to right */
static inline float vsum(v3 a) { return (a.x + a.y) + a.z; }
static inline float vdot(v3 a, v3 b) { return vsum(vmul(a, b)); }           /* vec.rs:99-101 */
static inline float vlength_sqr(v3 a) { return vsum(vmul(a, a)); }          /* vec.rs:87-89 */
static inline float vlength(v3 a) { return sqrtf(vlength_sqr(a)); }         /* vec.rs:91-93 */
/* _mm_min_ps / _mm_max_ps lane semantics (vec.rs:50-51): second operand on NaN */
static inline float sse_min(float a, float b) { return a < b ? a : b; }
static inline float sse_max(float a, float b) { return a > b ? a : b; }
static inline v3 vmin(v3 a, v3 b) { return V(sse_min(a.x, b.x), sse_min(a.y, b.y), sse_min(a.z, b.z)); }
static inline v3 vmax(v3 a, v3 b) { return V(sse_max(a.x, b.x), sse_max(a.y, b.y), sse_max(a.z, b.z)); }
static inline v3 vsqrt(v3 a) { return V(sqrtf(a.x), sqrtf(a.y), sqrtf(a.z)); } /* vec.rs:122 */
/* vec.rs:103-112 */
static inline v3 vcross(v3 a, v3 b)
{
    float x = a.y * b.z - a.z * b.y;
    float y = -(a.x * b.z - a.z * b.x);
    float z = a.x * b.y - a.y * b.x;
    return V(x, y, z);
}

/* P1: portable RSQ (see header).  vec.rs:38-44 */
ORA_API float ora_rsqrt(float x)
{
#ifdef ORA_RSQRT_SSE
    return _mm_cvtss_f32(_mm_rsqrt_ss(_mm_set_ss(x)));
#else
    return 1.0f / sqrtf(x);
#endif
}
static inline v3 vnormalized(v3 a) { return vmulf(a, ora_rsqrt(vlength_sqr(a))); } /* vec.rs:95-97 */

/* Rust `as i32` / `as u8`: truncate, saturate, NaN -> 0 (picture.rs:21-22, raytracer.rs:121) */
static inline int32_t trunc_sat_i32(float f)
{
    if (f != f) return 0;
    if (f >= 2147483648.0f) return INT32_MAX;
    if (f <= -2147483648.0f) return INT32_MIN;
    return (int32_t)f;
}
static inline uint8_t trunc_sat_u8(float f)
{
    if (f != f) return 0;
    if (f >= 255.0f) return 255;
    if (f <= 0.0f) return 0;
    return (uint8_t)f;
}
/* Rust f32::max / f32::min: IEEE maxNum / minNum, NaN-ignoring (aabb.rs:32-33) */
static inline float rust_max(float a, float b) { return fmaxf(a, b); }
static inline float rust_min(float a, float b) { return fminf(a, b); }

/* ------------------------------------------------------- RNG (rng.rs) */

/* rng.rs:50-60 */
ORA_API uint64_t ora_rng_raw(uint64_t *state)
{
    uint64_t x = *state;
    x ^= x << 13;
    x ^= x >> 7;
    x ^= x << 17;
    *state = x;
    return x;
}
/* rng.rs:27-35 */
ORA_API uint64_t ora_rng_with_seed(uint64_t seed)
{
    uint64_t s = seed;
    for (int i = 0; i < 10; i++) (void)ora_rng_raw(&s);
    return s;
}
/* rng.rs:76-80 */
ORA_API float ora_rng_f32(uint64_t *state)
{
    uint32_t v = (uint32_t)ora_rng_raw(state);
    return (float)(v >> 8) / (float)(1u << 24);
}
/* rng.rs:83-85 */
static inline float rng_range(uint64_t *state, float min, float max)
{
    return ora_rng_f32(state) * (max - min) + min;
}
/* P3: state of the (pixel, sample) stream after Rng::with_seed; splitmix64 finaliser,
 * never 0 (state 0 is the xorshift fixed point, rng.rs:50-60). */
ORA_API uint64_t ora_path_stream(uint64_t seed, uint32_t pixel, uint32_t sample)
{
    uint64_t k = ((uint64_t)pixel << 32) | (uint64_t)sample;
    uint64_t z = seed + (k + 1ull) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    if (z == 0) z = 0x9E3779B97F4A7C15ull;
    return z;
}
/* the seed that stream is Rng::with_seed of: ten inverse rounds of rng.rs:50-60
 * (x ^= x << n inverts as x ^= x << n, << 2n, << 4n ...; likewise >>) */
ORA_API uint64_t ora_path_seed(uint64_t seed, uint32_t pixel, uint32_t sample)
{
    uint64_t x = ora_path_stream(seed, pixel, sample);
    for (int i = 0; i < 10; i++) {
        x ^= x << 17; x ^= x << 34;
        x ^= x >> 7;  x ^= x >> 14; x ^= x >> 28; x ^= x >> 56;
        x ^= x << 13; x ^= x << 26; x ^= x << 52;
    }
    return x;
}

typedef struct { uint64_t state; ora_counters *cnt; } rng_t;
static inline float rnd(rng_t *r) { if (r->cnt) r->cnt->rng_draws++; return ora_rng_f32(&r->state); }
static inline float rnd_range(rng_t *r, float min, float max) { return rnd(r) * (max - min) + min; }

/* ------------------------------------------ P2: portable atan2f / asinf */

static const double ORA_PI = 3.14159265358979323846;
static const double ORA_PI_2 = 1.57079632679489661923;
static const double ORA_PI_4 = 0.78539816339744830962;
static const double ORA_TAN_PI_8 = 0.41421356237309504880;

/* atan(q), q in [0,1], fp64: one reduction step + 23-term odd Taylor series */
static double atan_unit(double q)
{
    double base = 0.0, t = q;
    if (q > ORA_TAN_PI_8) { t = (q - 1.0) / (q + 1.0); base = ORA_PI_4; }
    double z = t * t;
    double p = 1.0 / 45.0;
    for (int k = 21; k >= 0; k--) {
        double c = 1.0 / (double)(2 * k + 1);
        p = c - z * p;
    }
    return base + t * p;
}
/* finite inputs (NaN in -> NaN out); zero handling follows IEEE atan2 */
static double atan2_core(double y, double x)
{
    if (y != y || x != x) return NAN;
    double ay = fabs(y), ax = fabs(x), a;
    if (ax == 0.0 && ay == 0.0) a = 0.0;
    else if (ay > ax) a = ORA_PI_2 - atan_unit(ax / ay);
    else a = atan_unit(ay / ax);
    if (signbit(x)) a = ORA_PI - a;
    return signbit(y) ? -a : a;
}
ORA_API float ora_atan2f(float y, float x) { return (float)atan2_core((double)y, (double)x); }
ORA_API float ora_asinf(float y)
{
    double yd = (double)y;
    if (!(fabs(yd) <= 1.0)) return NAN;
    double c = sqrt((1.0 - yd) * (1.0 + yd));
    return (float)atan2_core(yd, c);
}

/* ------------------------------------------------- math/mod.rs helpers */

/* math/mod.rs:13-19 */
static v3 random_in_unit_sphere(rng_t *r)
{
    float x = rnd_range(r, -1.0f, 1.0f);
    float y = rnd_range(r, -1.0f, 1.0f);
    float z = rnd_range(r, -1.0f, 1.0f);
    return vnormalized(V(x, y, z));
}
/* math/mod.rs:21-23 */
static inline v3 reflect(v3 d, v3 n) { return vsub(d, vmulf(vmulf(n, vdot(d, n)), 2.0f)); }
/* math/mod.rs:25-34 */
static inline int refract(v3 d, v3 n, float ni_over_nt, v3 *out)
{
    float dt = vdot(d, n);
    float D = 1.0f - ni_over_nt * ni_over_nt * (1.0f - dt * dt);
    if (D > 0.0f) {
        *out = vsub(vmulf(vsub(d, vmulf(n, dt)), ni_over_nt), vmulf(n, sqrtf(D)));
        return 1;
    }
    return 0;
}
/* math/mod.rs:36-45 (note: v^5, not r0+(1-r0)(1-cos)^5 -- reproduced as written) */
ORA_API float ora_schlick(float cosine, float ref_idx)
{
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    r0 = r0 * r0;
    float v = r0 + (1.0f - r0) * (1.0f - cosine);
    float v2 = v * v;
    float v4 = v2 * v2;
    return v4 * v;
}
ORA_API void ora_reflect(const float d[3], const float n[3], float out[3])
{
    v3 r = reflect(V(d[0], d[1], d[2]), V(n[0], n[1], n[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
ORA_API int ora_refract(const float d[3], const float n[3], float k, float out[3])
{
    v3 r = V(0, 0, 0);
    int ok = refract(V(d[0], d[1], d[2]), V(n[0], n[1], n[2]), k, &r);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
    return ok;
}

/* ------------------------------------------------------- Ray / AABB */

typedef struct { v3 o, d; } ray_t;
/* ray.rs:19-24 */
static inline ray_t ray_new(v3 o, v3 d) { ray_t r = {o, vnormalized(d)}; return r; }
/* ray.rs:26-28 */
static inline v3 ray_point(const ray_t *r, float t) { return vadd(r->o, vmulf(r->d, t)); }

/* aabb.rs:25-36 */
static inline int aabb_intersect(v3 bmin, v3 bmax, const ray_t *ray, v3 inv, float min_t, float max_t)
{
    v3 v0 = vmul(vsub(bmin, ray->o), inv);
    v3 v1 = vmul(vsub(bmax, ray->o), inv);
    v3 mn = vmin(v0, v1);
    v3 mx = vmax(v0, v1);
    float tmin = rust_max(min_t, rust_max(mn.x, rust_max(mn.y, mn.z)));
    float tmax = rust_min(max_t, rust_min(mx.x, rust_min(mx.y, mx.z)));
    return tmin < tmax;
}
ORA_API int ora_aabb_intersect(const float bmin[3], const float bmax[3], const float o[3],
                               const float d[3], float min_t, float max_t)
{
    ray_t r = {V(o[0], o[1], o[2]), V(d[0], d[1], d[2])};
    v3 inv = vdiv(vfill(1.0f), r.d);                       /* scene/mod.rs:31 */
    return aabb_intersect(V(bmin[0], bmin[1], bmin[2]), V(bmax[0], bmax[1], bmax[2]), &r, inv, min_t, max_t);
}
/* aabb.rs:38-43 */
static inline void aabb_enclose(v3 *bmin, v3 *bmax, v3 omin, v3 omax)
{
    *bmin = vmin(*bmin, omin);
    *bmax = vmax(*bmax, omax);
}
/* aabb.rs:45-49 */
static inline float aabb_volume(v3 bmin, v3 bmax)
{
    v3 e = vsub(bmax, bmin);
    return fabsf(e.x * e.y * e.z);
}

/* ------------------------------------------------------- Sphere (sphere.rs) */

/* sphere.rs:71-76 */
static inline void sphere_bbox(const ora_sphere *s, v3 *bmin, v3 *bmax)
{
    v3 c = V(s->cx, s->cy, s->cz);
    *bmin = vsub(c, vfill(s->radius));
    *bmax = vadd(c, vfill(s->radius));
}

/* hit_t: the HitRecord of traceable/mod.rs:11-17 + the primitive id */
typedef struct { float t; v3 point, normal; int prim; } hit_t;

/* sphere.rs:37-42 */
static inline void sphere_record(const ora_sphere *s, int prim, float t, const ray_t *ray, hit_t *h)
{
    v3 p = ray_point(ray, t);
    h->t = t;
    h->point = p;
    h->normal = vnormalized(vsub(p, V(s->cx, s->cy, s->cz)));
    h->prim = prim;
}
/* sphere.rs:46-69 */
static inline void ev(char c);
static inline int sphere_trace(const ora_sphere *s, int prim, const ray_t *ray, float min_t, float max_t,
                               hit_t *h, ora_counters *cnt)
{
    v3 oc = vsub(ray->o, V(s->cx, s->cy, s->cz));
    float a = vdot(ray->d, ray->d);
    float b = vdot(oc, ray->d);
    float c = vdot(oc, oc) - s->radius * s->radius;
    float d = b * b - a * c;
    if (cnt) cnt->sphere_tests++;
    if (d > 0.0f) {
        ev('P');                                        /* positive discriminant: roots evaluated */
        float sol = (-b - sqrtf(d)) / a;
        if (sol < max_t && sol > min_t) {
            if (cnt) cnt->sphere_near_hits++;
            sphere_record(s, prim, sol, ray, h);
            return 1;
        }
        sol = (-b + sqrtf(d)) / a;
        if (sol < max_t && sol > min_t) {
            if (cnt) cnt->sphere_far_hits++;
            sphere_record(s, prim, sol, ray, h);
            return 1;
        }
    }
    return 0;
}
ORA_API int ora_sphere_trace(const float center[3], float radius, const float o[3], const float d[3],
                             float min_t, float max_t, float out_t_p_n[7])
{
    ora_sphere s = {center[0], center[1], center[2], radius, 0};
    ray_t r = {V(o[0], o[1], o[2]), V(d[0], d[1], d[2])};
    hit_t h;
    memset(&h, 0, sizeof h);
    int ok = sphere_trace(&s, 0, &r, min_t, max_t, &h, NULL);
    out_t_p_n[0] = h.t;
    out_t_p_n[1] = h.point.x; out_t_p_n[2] = h.point.y; out_t_p_n[3] = h.point.z;
    out_t_p_n[4] = h.normal.x; out_t_p_n[5] = h.normal.y; out_t_p_n[6] = h.normal.z;
    return ok;
}

/* sphere.rs:6-18 */
ORA_API void ora_sphere_uv(const float n[3], float uv[2])
{
    float phi = ora_atan2f(n[2], n[0]);
    float theta = ora_asinf(n[1]);
    const float pi = 3.14159265358979323846f;              /* std::f32::consts::PI */
    uv[0] = 1.0f - (phi + pi) / (2.0f * pi);
    uv[1] = (theta + pi / 2.0f) / pi;
}
/* picture.rs:17-38 */
ORA_API void ora_texture_color(const uint8_t *rgb8, int32_t width, int32_t height, float u, float v, float out[3])
{
    int32_t i = trunc_sat_i32(u * (float)width);
    int32_t j = trunc_sat_i32(((1.0f - v) * (float)height) - 0.001f);
    /* clamp = value.min(max).max(min)  (picture.rs:24-26) */
    i = i < width - 1 ? i : width - 1;   i = i > 0 ? i : 0;
    j = j < height - 1 ? j : height - 1; j = j > 0 ? j : 0;
    const uint8_t *px = rgb8 + 3 * ((size_t)j * (size_t)width + (size_t)i);
    out[0] = (float)px[0] / 255.0f;
    out[1] = (float)px[1] / 255.0f;
    out[2] = (float)px[2] / 255.0f;
}

/* ------------------------------------------------------- EXTENSION: rects and instances
 * Not in the reference.  Shapes follow Traceable (traceable/mod.rs:36-39): trace(ray, min_t, max_t)
 * -> Option<HitRecord>, bounding_box().  Semantics:
 *   rect: plane `axis` = k over [a0,a1] x [b0,b1] on the other two axes (increasing axis order);
 *         t = (k - o[axis]) / d[axis]; hit iff t < max_t && t > min_t (the comparison of
 *         sphere.rs:57) and a0 <= a <= a1 and b0 <= b <= b1 with a = o[a] + d[a]*t; a NaN fails;
 *         normal = the axis' unit vector, negated by `flip`; point = o + d*t;
 *   Translate(RotateY(rect)): the ray goes to object space un-normalised (o - offset, then the
 *         rotation by -theta on x/z), t is the same in both spaces; point and normal come back
 *         by the rotation by +theta, the point then + offset. */
static inline void rect_axes(int axis, int *ia, int *ib) { *ia = axis == 0 ? 1 : 0; *ib = axis == 2 ? 1 : 2; }
static inline v3 rot_to_object(const ora_transform *x, v3 p)
{
    return V(x->cos_theta * p.x - x->sin_theta * p.z, p.y, x->sin_theta * p.x + x->cos_theta * p.z);
}
static inline v3 rot_to_world(const ora_transform *x, v3 p)
{
    return V(x->cos_theta * p.x + x->sin_theta * p.z, p.y, x->cos_theta * p.z - x->sin_theta * p.x);
}
static inline int rect_trace(const ora_scene *sc, const ora_primitive *p, int prim, const ray_t *ray, float min_t,
                             float max_t, hit_t *h, ora_counters *cnt)
{
    v3 o = ray->o, d = ray->d;
    const ora_transform *x = p->transform >= 0 ? &sc->xforms[p->transform] : NULL;
    if (cnt) { cnt->rect_tests++; if (x) cnt->rect_transformed++; }
    if (x) {
        o = rot_to_object(x, vsub(o, V(x->offset[0], x->offset[1], x->offset[2])));
        d = rot_to_object(x, d);
    }
    const float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
    int ia, ib;
    rect_axes(p->axis, &ia, &ib);
    float t = (p->v[4] - oo[p->axis]) / dd[p->axis];
    if (!(t < max_t && t > min_t)) return 0;
    if (cnt) cnt->rect_in_range++;
    float a = oo[ia] + dd[ia] * t, b = oo[ib] + dd[ib] * t;
    if (!(a >= p->v[0] && a <= p->v[1] && b >= p->v[2] && b <= p->v[3])) return 0;
    v3 pt = vadd(o, vmulf(d, t));
    float nn[3] = {0.0f, 0.0f, 0.0f};
    nn[p->axis] = p->flip ? -1.0f : 1.0f;
    v3 n = V(nn[0], nn[1], nn[2]);
    if (x) {
        pt = vadd(rot_to_world(x, pt), V(x->offset[0], x->offset[1], x->offset[2]));
        n = rot_to_world(x, n);
    }
    if (cnt) { cnt->rect_hits++; if (x) cnt->rect_hits_transformed++; }
    h->t = t; h->point = pt; h->normal = n; h->prim = prim;
    return 1;
}
/* bounding_box(): the rect padded by 0.0001 on its thin axis; a transformed rect: the box of its
 * 8 rotated corners, then + offset */
static inline void rect_bbox(const ora_scene *sc, const ora_primitive *p, v3 *bmin, v3 *bmax)
{
    float lo[3], hi[3];
    int ia, ib;
    rect_axes(p->axis, &ia, &ib);
    lo[p->axis] = p->v[4] - 0.0001f; hi[p->axis] = p->v[4] + 0.0001f;
    lo[ia] = p->v[0]; hi[ia] = p->v[1];
    lo[ib] = p->v[2]; hi[ib] = p->v[3];
    if (p->transform < 0) { *bmin = V(lo[0], lo[1], lo[2]); *bmax = V(hi[0], hi[1], hi[2]); return; }
    const ora_transform *x = &sc->xforms[p->transform];
    for (int c = 0; c < 8; c++) {
        v3 w = rot_to_world(x, V((c & 1) ? hi[0] : lo[0], (c & 2) ? hi[1] : lo[1], (c & 4) ? hi[2] : lo[2]));
        if (c == 0) { *bmin = w; *bmax = w; }
        else { *bmin = vmin(w, *bmin); *bmax = vmax(w, *bmax); }
    }
    *bmin = vadd(*bmin, V(x->offset[0], x->offset[1], x->offset[2]));
    *bmax = vadd(*bmax, V(x->offset[0], x->offset[1], x->offset[2]));
}

/* ------------------------------------------------------- BVH (scene/bvh.rs) */

typedef struct { int prim; v3 bmin, bmax; } build_item;

/* stable merge sort on the bbox centre along `axis`.  The reference uses
 * sort_unstable_by (bvh.rs:51-56, pdqsort), whose order for EQUAL keys is an
 * implementation detail of the Rust std version: unpinned.  Ties here keep
 * their previous order (stable).  Tie order changes only the tree shape, never
 * which sphere is closest. */
static float item_key(const build_item *it, int axis)
{
    /* aabb.rs:21-23 center = (min + max) * 0.5 */
    v3 c = vmulf(vadd(it->bmin, it->bmax), 0.5f);
    return axis == 0 ? c.x : (axis == 1 ? c.y : c.z);
}
static void stable_sort_items(build_item *a, build_item *tmp, int n, int axis)
{
    if (n < 2) return;
    int h = n / 2;
    stable_sort_items(a, tmp, h, axis);
    stable_sort_items(a + h, tmp, n - h, axis);
    int i = 0, j = h, k = 0;
    while (i < h && j < n) {
        /* partial_cmp(..).unwrap_or(Equal): take right only if strictly less */
        if (item_key(&a[j], axis) < item_key(&a[i], axis)) tmp[k++] = a[j++];
        else tmp[k++] = a[i++];
    }
    while (i < h) tmp[k++] = a[i++];
    while (j < n) tmp[k++] = a[j++];
    memcpy(a, tmp, (size_t)n * sizeof *a);
}
/* bvh.rs:26-33 */
static void enclosing_bbox(const build_item *it, int n, v3 *bmin, v3 *bmax)
{
    *bmin = it[0].bmin; *bmax = it[0].bmax;
    for (int i = 1; i < n; i++) aabb_enclose(bmin, bmax, it[i].bmin, it[i].bmax);
}
/* bvh.rs:13-91, literal (O(n^2) per node like the reference) */
static bvh_node *bvh_construct(build_item *items, build_item *tmp, int n, int *n_nodes)
{
    bvh_node *node = (bvh_node *)calloc(1, sizeof *node);
    (*n_nodes)++;
    if (n == 1) {                                           /* bvh.rs:20-24 */
        node->is_leaf = 1;
        node->prim = items[0].prim;
        node->bmin = items[0].bmin; node->bmax = items[0].bmax;
        return node;
    }
    v3 wmin, wmax;
    enclosing_bbox(items, n, &wmin, &wmax);                 /* bvh.rs:35 */
    v3 e = vsub(wmax, wmin);                                /* aabb.rs:17-19 */
    float ext[3] = {e.x, e.y, e.z};
    int axis = 0;                                           /* bvh.rs:37-49 */
    for (int i = 1; i < 3; i++) if (ext[i] > ext[axis]) axis = i;
    stable_sort_items(items, tmp, n, axis);                 /* bvh.rs:51-56 */
    int best = 0; float lowest = 3.402823466e+38f;          /* bvh.rs:58-80 */
    float wvol = aabb_volume(wmin, wmax);
    for (int i = 1; i < n; i++) {
        v3 lmin, lmax, rmin, rmax;
        enclosing_bbox(items, i, &lmin, &lmax);
        enclosing_bbox(items + i, n - i, &rmin, &rmax);
        float cl = (aabb_volume(lmin, lmax) / wvol) * (float)i;
        float cr = (aabb_volume(rmin, rmax) / wvol) * (float)(n - i);
        float cost = cl + cr;
        if (cost < lowest) { best = i; lowest = cost; }
    }
    if (best == 0) abort();                                 /* bvh.rs:19 panic */
    node->left = bvh_construct(items, tmp, best, n_nodes);  /* bvh.rs:82-85 */
    node->right = bvh_construct(items + best, tmp, n - best, n_nodes);
    node->bmin = node->left->bmin; node->bmax = node->left->bmax; /* bvh.rs:86 */
    aabb_enclose(&node->bmin, &node->bmax, node->right->bmin, node->right->bmax);
    return node;
}
static void bvh_free(bvh_node *n)
{
    if (!n) return;
    bvh_free(n->left); bvh_free(n->right); free(n);
}
/* optional per-thread event log (tests/experiments/sim_vote.py): 'S' box test, 'P' (before L/X) the sphere test has a positive discriminant,
 * 'L' sphere test that misses, 'X' one that hits,
 * 'H' scatter, 'E' end of path */
static __thread char *ev_buf; static __thread int ev_n, ev_cap;
static inline void ev(char c) { if (ev_buf) { if (ev_n < ev_cap) ev_buf[ev_n] = c; ev_n++; } }
/* bvh.rs:106-127 */
static int bvh_trace(const ora_scene *sc, const bvh_node *n, const ray_t *ray, v3 inv, float min_t,
                     float max_t, hit_t *h, ora_counters *cnt)
{
    if (n->is_leaf) {
        int r = (sc->ext && sc->prims[n->prim].kind == ORA_PRIM_RECT)
                    ? rect_trace(sc, &sc->prims[n->prim], n->prim, ray, min_t, max_t, h, cnt)   /* extension */
                    : sphere_trace(&sc->spheres[n->prim], n->prim, ray, min_t, max_t, h, cnt);
        ev(r ? 'X' : 'L');
        return r;
    }
    ev('S');
    if (cnt) cnt->aabb_tests++;
    if (!aabb_intersect(n->bmin, n->bmax, ray, inv, min_t, max_t)) return 0;
    hit_t hl;
    if (bvh_trace(sc, n->left, ray, inv, min_t, max_t, &hl, cnt)) {
        hit_t hr;
        if (bvh_trace(sc, n->right, ray, inv, min_t, hl.t, &hr, cnt)) *h = hr;
        else *h = hl;
        return 1;
    }
    return bvh_trace(sc, n->right, ray, inv, min_t, max_t, h, cnt);
}
/* scene/mod.rs:24-34 */
static inline int scene_trace(const ora_scene *sc, const ray_t *ray, hit_t *h, ora_counters *cnt)
{
    const float T_MIN = 0.001f;
    v3 inv = vdiv(vfill(1.0f), ray->d);
    return bvh_trace(sc, sc->root, ray, inv, T_MIN, 3.402823466e+38f, h, cnt);
}

/* ------------------------------------------------------- materials */

static inline v3 sphere_uv_color(const ora_scene *sc, const ora_material *m, const hit_t *h)
{
    float n[3] = {h->normal.x, h->normal.y, h->normal.z}, uv[2], c[3];
    ora_sphere_uv(n, uv);
    const ora_texture *t = &sc->textures[m->texture];
    ora_texture_color(t->rgb8, t->width, t->height, uv[0], uv[1], c);
    return V(c[0], c[1], c[2]);
}
/* returns 1 = scattered (att, out), 0 = absorbed */
static int scatter(const ora_scene *sc, const ray_t *ray, const hit_t *h, rng_t *rng, v3 *att, ray_t *out,
                   ora_counters *cnt)
{
    const ora_material *m = &sc->materials[sc->spheres[h->prim].material];
    switch (m->kind) {
    case ORA_DIFFUSE_LIGHT:                                 /* extension: a light does not scatter */
        return 0;
    case ORA_LAMBERTIAN: {                                  /* lambertian.rs:32-45 */
        v3 target = vadd(vadd(h->point, h->normal), random_in_unit_sphere(rng));
        if (m->texture >= 0) { *att = sphere_uv_color(sc, m, h); if (cnt) cnt->lambert_textured++; }
        else *att = V(m->albedo[0], m->albedo[1], m->albedo[2]);
        if (cnt) cnt->lambert++;
        *out = ray_new(h->point, vsub(target, h->point));
        return 1;
    }
    case ORA_METAL: {                                       /* metal.rs:22-32 */
        v3 reflected = reflect(ray->d, h->normal);
        if (vdot(reflected, h->normal) > 0.0f) {
            v3 fuzz = vmulf(random_in_unit_sphere(rng), m->param);
            *att = V(m->albedo[0], m->albedo[1], m->albedo[2]);
            *out = ray_new(h->point, vadd(reflected, fuzz));
            if (cnt) cnt->metal_scatter++;
            return 1;
        }
        if (cnt) cnt->metal_absorb++;
        return 0;
    }
    default: {                                              /* dielectric.rs:20-50 */
        float ref_idx = m->param;
        v3 dir = ray->d;
        float dot = vdot(dir, h->normal);
        v3 outward; float ni_over_nt, cosine;
        if (dot > 0.0f) { cosine = ref_idx * dot; outward = vneg(h->normal); ni_over_nt = ref_idx; }
        else { cosine = -dot; outward = h->normal; ni_over_nt = 1.0f / ref_idx; }
        v3 reflected = reflect(dir, h->normal);
        v3 refracted;
        *att = vfill(1.0f);
        if (refract(dir, outward, ni_over_nt, &refracted)) {
            float p = ora_schlick(cosine, ref_idx);
            float r = rnd(rng);
            *out = ray_new(h->point, r < p ? reflected : refracted);
            if (cnt) cnt->dielectric_refract++;
        } else {
            *out = ray_new(h->point, reflected);
            if (cnt) cnt->dielectric_tir++;
        }
        return 1;
    }
    }
}

/* ------------------------------------------------------- camera (camera.rs) */

/* camera.rs:14-34; fov in degrees.  tanf is the host libm's (unpinned, host side
 * only: the derived fields are what cross the product's C ABI). */
ORA_API void ora_camera_new(const float eye[3], const float target[3], const float up[3], float fov,
                            int32_t width, int32_t height, ora_camera *out)
{
    float aspect = (float)width / (float)height;
    float radians = fov * (3.14159265358979323846f / 180.0f); /* f32::to_radians */
    float half_height = tanf(radians / 2.0f);
    float half_width = half_height * aspect;
    v3 e = V(eye[0], eye[1], eye[2]), t = V(target[0], target[1], target[2]), u0 = V(up[0], up[1], up[2]);
    v3 w = vnormalized(vsub(e, t));
    v3 u = vnormalized(vcross(u0, w));
    v3 v = vcross(w, u);
    v3 llc = vsub(vsub(vsub(e, vmulf(u, half_width)), vmulf(v, half_height)), w);
    v3 hor = vmulf(vmulf(u, 2.0f), half_width);
    v3 ver = vmulf(vmulf(v, 2.0f), half_height);
    out->llc[0] = llc.x; out->llc[1] = llc.y; out->llc[2] = llc.z;
    out->origin[0] = e.x; out->origin[1] = e.y; out->origin[2] = e.z;
    out->horizontal[0] = hor.x; out->horizontal[1] = hor.y; out->horizontal[2] = hor.z;
    out->vertical[0] = ver.x; out->vertical[1] = ver.y; out->vertical[2] = ver.z;
    out->width = width; out->height = height;
}
/* camera.rs:37-44 */
static inline ray_t camera_ray(const ora_camera *c, float u, float v)
{
    v3 llc = V(c->llc[0], c->llc[1], c->llc[2]), org = V(c->origin[0], c->origin[1], c->origin[2]);
    v3 hor = V(c->horizontal[0], c->horizontal[1], c->horizontal[2]);
    v3 ver = V(c->vertical[0], c->vertical[1], c->vertical[2]);
    ray_t r;
    r.o = org;
    r.d = vsub(vadd(vadd(llc, vmulf(hor, u)), vmulf(ver, v)), org);
    return r;
}

/* ------------------------------------------------------- integrator (raytracer.rs) */

/* raytracer.rs:46-71 */
static v3 trace_ray(const ora_scene *sc, ray_t ray, rng_t *rng, ora_counters *cnt)
{
    const int MAX_TRACES = 5;
    v3 attenuation = vfill(1.0f);
    if (cnt) cnt->paths++;
    for (int i = 0; i < MAX_TRACES; i++) {
        hit_t h;
        if (cnt) cnt->segments++;
        if (scene_trace(sc, &ray, &h, cnt)) {
            v3 att; ray_t nr;
            ev('H');
            if (scatter(sc, &ray, &h, rng, &att, &nr, cnt)) {
                attenuation = vmul(attenuation, att);
                ray = nr;
            } else {
                ev('E');
                return vfill(0.0f);
            }
        } else {
            break;
        }
    }
    ev('E');
    if (cnt) cnt->sky_evals++;
    float t = 0.5f * (ray.d.y + 1.0f);
    v3 color = vadd(vmulf(V(1.0f, 1.0f, 1.0f), 1.0f - t), vmulf(V(0.5f, 0.7f, 1.0f), t));
    return vmul(color, attenuation);
}

/* EXTENSION: trace_ray with an emitted term and its constants as parameters (ora_integrator).  With the
 * defaults (5 bounces, sky, no roulette) and no light in the scene it returns, bit for bit, what
 * trace_ray returns: radiance stays +0 and 0 + x == x. */
static v3 trace_ray_ext(const ora_scene *sc, ray_t ray, rng_t *rng, ora_counters *cnt)
{
    const ora_integrator *I = &sc->integ;
    v3 attenuation = vfill(1.0f), radiance = vfill(0.0f);
    if (cnt) cnt->paths++;
    for (int i = 0; i < I->max_bounces; i++) {
        hit_t h;
        if (cnt) cnt->segments++;
        if (!scene_trace(sc, &ray, &h, cnt)) break;
        ev('H');
        const ora_material *m = &sc->materials[sc->spheres[h.prim].material];
        if (m->kind == ORA_DIFFUSE_LIGHT) {                 /* emitted(): the light's colour, whatever the side */
            radiance = vadd(radiance, vmul(V(m->albedo[0], m->albedo[1], m->albedo[2]), attenuation));
            if (cnt) cnt->light_hits++;
        }
        v3 att; ray_t nr;
        if (!scatter(sc, &ray, &h, rng, &att, &nr, cnt)) { ev('E'); return radiance; }
        attenuation = vmul(attenuation, att);
        ray = nr;
        if (I->rr_start >= 0 && i + 1 >= I->rr_start) {     /* Russian roulette on the largest component */
            float q = rust_max(attenuation.x, rust_max(attenuation.y, attenuation.z));
            if (q < 1.0f) {
                if (cnt) cnt->rr_draws++;
                float r = rnd(rng);
                if (!(r < q)) { ev('E'); return radiance; }
                attenuation = vdivf(attenuation, q);
            }
        }
    }
    ev('E');
    if (cnt) cnt->sky_evals++;
    v3 color;
    if (I->background_kind == 0) {                          /* raytracer.rs:67-68 */
        float t = 0.5f * (ray.d.y + 1.0f);
        color = vadd(vmulf(V(1.0f, 1.0f, 1.0f), 1.0f - t), vmulf(V(0.5f, 0.7f, 1.0f), t));
    } else {
        color = V(I->background[0], I->background[1], I->background[2]);
    }
    return vadd(radiance, vmul(color, attenuation));
}

/* the sample grid of raytracer.rs:77-86; s = sx * S + sy (sx outer, sy inner) */
static inline ray_t sample_ray(const ora_camera *cam, int S, int px, int py, int s)
{
    int sx = s / S, sy = s % S;
    float x = (float)px + ((float)sx / (float)(S - 1));
    float u = x / (float)cam->width;
    float y = (float)py + ((float)sy / (float)(S - 1));
    float v = 1.0f - (y / (float)cam->height);
    return camera_ray(cam, u, v);
}

/* raytracer.rs:73-96 restricted to sample indices [s0, s1): the un-normalised
 * radiance SUM.  chunk == 0: one sequential fp32 sum in sample order (the
 * reference's order, raytracer.rs:88).  chunk > 0: sequential sums over
 * consecutive groups of `chunk` samples (counted from s0),
 * then the group sums added in order -- the order the HIP library uses when it
 * splits a pixel's samples over several lanes. */
static v3 trace_pixel_sum(const ora_scene *sc, const ora_camera *cam, int S, uint64_t seed, int px, int py,
                          int s0, int s1, int chunk, ora_counters *cnt)
{
    uint32_t pixel = (uint32_t)(py * cam->width + px);
    v3 total = vfill(0.0f);
    v3 part = vfill(0.0f);
    for (int s = s0; s < s1; s++) {
        rng_t rng = {ora_path_stream(seed, pixel, (uint32_t)s), cnt};
        v3 c = sc->ext ? trace_ray_ext(sc, sample_ray(cam, S, px, py, s), &rng, cnt)
                       : trace_ray(sc, sample_ray(cam, S, px, py, s), &rng, cnt);
        if (chunk > 0) {
            part = vadd(part, c);
            if ((s - s0 + 1) % chunk == 0 || s + 1 == s1) { total = vadd(total, part); part = vfill(0.0f); }
        } else {
            total = vadd(total, c);
        }
    }
    return total;
}

/* raytracer.rs:92-93 + 119-121 */
ORA_API void ora_resolve(const float *radiance_sum, int64_t n_pixels, int32_t samples, uint8_t *rgb8)
{
    float n = (float)(samples * samples);
    for (int64_t i = 0; i < n_pixels; i++) {
        v3 c = vsqrt(vdivf(V(radiance_sum[3 * i], radiance_sum[3 * i + 1], radiance_sum[3 * i + 2]), n));
        v3 b = vmulf(c, 255.0f);
        rgb8[3 * i + 0] = trunc_sat_u8(b.x);
        rgb8[3 * i + 1] = trunc_sat_u8(b.y);
        rgb8[3 * i + 2] = trunc_sat_u8(b.z);
    }
}

/* ------------------------------------------------------- scene objects */

/* copies of the caller's arrays; no tree yet */
static ora_scene *scene_alloc(const ora_sphere *spheres, int32_t n_spheres, const ora_material *materials,
                              int32_t n_materials, const ora_texture *textures, int32_t n_textures)
{
    ora_scene *sc = (ora_scene *)calloc(1, sizeof *sc);
    sc->n_spheres = n_spheres; sc->n_materials = n_materials; sc->n_textures = n_textures;
    sc->spheres = (ora_sphere *)malloc((size_t)n_spheres * sizeof *spheres);
    memcpy(sc->spheres, spheres, (size_t)n_spheres * sizeof *spheres);
    sc->materials = (ora_material *)malloc((size_t)(n_materials > 0 ? n_materials : 1) * sizeof *materials);
    memcpy(sc->materials, materials, (size_t)n_materials * sizeof *materials);
    sc->textures = (ora_texture *)calloc((size_t)(n_textures > 0 ? n_textures : 1), sizeof *textures);
    for (int i = 0; i < n_textures; i++) {
        size_t bytes = (size_t)textures[i].width * (size_t)textures[i].height * 3;
        uint8_t *copy = (uint8_t *)malloc(bytes);
        memcpy(copy, textures[i].rgb8, bytes);
        sc->textures[i].width = textures[i].width;
        sc->textures[i].height = textures[i].height;
        sc->textures[i].rgb8 = copy;
    }
    return sc;
}
ORA_API ora_scene *ora_scene_create(const ora_sphere *spheres, int32_t n_spheres, const ora_material *materials,
                                    int32_t n_materials, const ora_texture *textures, int32_t n_textures)
{
    if (n_spheres < 1) return NULL;
    ora_scene *sc = scene_alloc(spheres, n_spheres, materials, n_materials, textures, n_textures);
    build_item *items = (build_item *)malloc((size_t)n_spheres * sizeof *items);
    build_item *tmp = (build_item *)malloc((size_t)n_spheres * sizeof *tmp);
    for (int i = 0; i < n_spheres; i++) {                   /* scene/mod.rs:46-48: insertion order */
        items[i].prim = i;
        sphere_bbox(&spheres[i], &items[i].bmin, &items[i].bmax);
    }
    sc->root = bvh_construct(items, tmp, n_spheres, &sc->n_nodes); /* scene/mod.rs:50-60 */
    free(items); free(tmp);
    return sc;
}
/* EXTENSION: a scene over a primitive list (spheres, rects with optional instances); trace_ray_ext is its
 * integrator (defaults = the reference's constants until ora_set_integrator). */
ORA_API ora_scene *ora_scene_create_ext(const ora_primitive *prims, int32_t n_prims, const ora_material *materials,
                                        int32_t n_materials, const ora_texture *textures, int32_t n_textures,
                                        const ora_transform *xforms, int32_t n_xforms)
{
    if (n_prims < 1) return NULL;
    ora_sphere *spheres = (ora_sphere *)calloc((size_t)n_prims, sizeof *spheres);
    for (int i = 0; i < n_prims; i++) {
        if (prims[i].kind == ORA_PRIM_SPHERE) {
            ora_sphere s = {prims[i].v[0], prims[i].v[1], prims[i].v[2], prims[i].v[3], prims[i].material};
            spheres[i] = s;
        } else {
            spheres[i].material = prims[i].material;
        }
    }
    ora_scene *sc = scene_alloc(spheres, n_prims, materials, n_materials, textures, n_textures);
    free(spheres);
    sc->ext = 1;
    sc->n_prims = n_prims; sc->n_xforms = n_xforms;
    sc->prims = (ora_primitive *)malloc((size_t)n_prims * sizeof *prims);
    memcpy(sc->prims, prims, (size_t)n_prims * sizeof *prims);
    sc->xforms = (ora_transform *)calloc((size_t)(n_xforms > 0 ? n_xforms : 1), sizeof *xforms);
    if (n_xforms > 0) memcpy(sc->xforms, xforms, (size_t)n_xforms * sizeof *xforms);
    ora_integrator dflt = {5, 0, {0.0f, 0.0f, 0.0f}, -1};    /* raytracer.rs:48,67-68 */
    sc->integ = dflt;
    build_item *items = (build_item *)malloc((size_t)n_prims * sizeof *items);
    build_item *tmp = (build_item *)malloc((size_t)n_prims * sizeof *tmp);
    for (int i = 0; i < n_prims; i++) {
        items[i].prim = i;
        if (prims[i].kind == ORA_PRIM_RECT) rect_bbox(sc, &prims[i], &items[i].bmin, &items[i].bmax);
        else sphere_bbox(&sc->spheres[i], &items[i].bmin, &items[i].bmax);
    }
    sc->root = bvh_construct(items, tmp, n_prims, &sc->n_nodes);
    free(items); free(tmp);
    return sc;
}
ORA_API int ora_set_integrator(ora_scene *sc, const ora_integrator *integ)
{
    if (!sc || !sc->ext || !integ || integ->max_bounces < 1) return -1;
    sc->integ = *integ;
    return 0;
}
ORA_API void ora_scene_destroy(ora_scene *sc)
{
    if (!sc) return;
    free(sc->prims); free(sc->xforms);
    bvh_free(sc->root);
    for (int i = 0; i < sc->n_textures; i++) free((void *)sc->textures[i].rgb8);
    free(sc->textures); free(sc->materials); free(sc->spheres); free(sc);
}
ORA_API int32_t ora_scene_node_count(const ora_scene *sc) { return sc->n_nodes; }

static int flatten_rec(const bvh_node *n, ora_node *out, int idx)
{
    int me = idx++;
    out[me].bmin[0] = n->bmin.x; out[me].bmin[1] = n->bmin.y; out[me].bmin[2] = n->bmin.z;
    out[me].bmax[0] = n->bmax.x; out[me].bmax[1] = n->bmax.y; out[me].bmax[2] = n->bmax.z;
    out[me].prim = n->is_leaf ? n->prim : -1;
    if (!n->is_leaf) { idx = flatten_rec(n->left, out, idx); idx = flatten_rec(n->right, out, idx); }
    out[me].skip = idx;
    return idx;
}
/* DFS pre-order dump (left before right); skip = index of the first node after the subtree */
ORA_API int32_t ora_scene_flatten(const ora_scene *sc, ora_node *out, int32_t capacity)
{
    if (capacity < sc->n_nodes) return -1;
    return flatten_rec(sc->root, out, 0);
}

/* ------------------------------------------------------- parallel render */

typedef struct {
    const ora_scene *sc; const ora_camera *cam;
    int S, s0, s1, chunk; uint64_t seed;
    float *radiance; int32_t *hit_id; float *hit_t; int aov_sample;
    int64_t pixel_count, pixels_per_range; int range_count;
    atomic_int next_range;
    int want_counters;
    ora_counters *per_thread;
} render_job;

static void render_range(render_job *job, int64_t start, int64_t size, ora_counters *cnt)
{
    const int W = job->cam->width;
    for (int64_t i = 0; i < size; i++) {                    /* raytracer.rs:104-106 */
        int64_t idx = i + start;
        int x = (int)(idx % W), y = (int)(idx / W);
        if (job->radiance) {
            v3 sum = trace_pixel_sum(job->sc, job->cam, job->S, job->seed, x, y, job->s0, job->s1, job->chunk, cnt);
            job->radiance[3 * idx] = sum.x; job->radiance[3 * idx + 1] = sum.y; job->radiance[3 * idx + 2] = sum.z;
        }
        if (job->hit_id || job->hit_t) {
            ray_t r = sample_ray(job->cam, job->S, x, y, job->aov_sample);
            hit_t h;
            int ok = scene_trace(job->sc, &r, &h, NULL);
            if (job->hit_id) job->hit_id[idx] = ok ? h.prim : -1;
            if (job->hit_t) job->hit_t[idx] = ok ? h.t : 3.402823466e+38f;
        }
    }
}
/* worker loop of parallel_renderer/mod.rs:126-138 (atomic pop of contiguous ranges) */
static void *worker(void *arg)
{
    render_job *job = (render_job *)((void **)arg)[0];
    ora_counters *cnt = (ora_counters *)((void **)arg)[1];
    for (;;) {
        int r = atomic_fetch_add(&job->next_range, 1);      /* mod.rs:31-39 */
        if (r >= job->range_count) break;
        int64_t start = job->pixels_per_range * r;
        if (start >= job->pixel_count) continue;
        int64_t size = job->pixels_per_range;
        if (start + size > job->pixel_count) size = job->pixel_count - start; /* F13 fixed */
        render_range(job, start, size, cnt);
    }
    return NULL;
}

/* ParallelRenderer::render (parallel_renderer/mod.rs:155-203) driving
 * Raytracer::render_fragment (raytracer.rs:98-123) over sample indices
 * [sample_begin, sample_end).  Outputs (any may be NULL):
 *   radiance : W*H*3 fp32 un-normalised sums;  hit_id / hit_t : primary-hit AOV of
 *   sample index aov_sample;  counters : summed event counts. */
ORA_API int ora_render(const ora_scene *sc, const ora_camera *cam, int32_t samples, uint64_t seed,
                       int32_t sample_begin, int32_t sample_end, int32_t chunk, int32_t threads,
                       float *radiance, int32_t *hit_id, float *hit_t, int32_t aov_sample,
                       ora_counters *counters)
{
    if (!sc || !cam || samples < 2 || threads < 1) return -1;
    if (sample_begin < 0 || sample_end > samples * samples || sample_begin > sample_end) return -1;
    render_job job;
    memset(&job, 0, sizeof job);
    job.sc = sc; job.cam = cam; job.S = samples; job.s0 = sample_begin; job.s1 = sample_end;
    job.chunk = chunk; job.seed = seed; job.radiance = radiance; job.hit_id = hit_id; job.hit_t = hit_t;
    job.aov_sample = aov_sample;
    job.pixel_count = (int64_t)cam->width * cam->height;
    job.range_count = threads * 64;                         /* mod.rs:162 */
    job.pixels_per_range = (job.pixel_count + job.range_count - 1) / job.range_count; /* mod.rs:163 */
    atomic_init(&job.next_range, 0);
    ora_counters *per = (ora_counters *)calloc((size_t)threads, sizeof *per);
    pthread_t *th = (pthread_t *)malloc((size_t)threads * sizeof *th);
    void **args = (void **)malloc((size_t)threads * 2 * sizeof(void *));
    for (int t = 0; t < threads; t++) {
        args[2 * t] = &job; args[2 * t + 1] = counters ? &per[t] : NULL;
        pthread_create(&th[t], NULL, worker, &args[2 * t]);
    }
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    if (counters) {
        memset(counters, 0, sizeof *counters);
        uint64_t *dst = (uint64_t *)counters;
        for (int t = 0; t < threads; t++) {
            const uint64_t *src = (const uint64_t *)&per[t];
            for (size_t k = 0; k < sizeof(ora_counters) / sizeof(uint64_t); k++) dst[k] += src[k];
        }
    }
    free(args); free(th); free(per);
    return 0;
}

/* one full path, for step-by-step debugging and KATs */
ORA_API void ora_trace_path(const ora_scene *sc, const ora_camera *cam, int32_t samples, uint64_t seed,
                            int32_t px, int32_t py, int32_t s, float out_rgb[3])
{
    uint32_t pixel = (uint32_t)(py * cam->width + px);
    rng_t rng = {ora_path_stream(seed, pixel, (uint32_t)s), NULL};
    v3 c = sc->ext ? trace_ray_ext(sc, sample_ray(cam, samples, px, py, s), &rng, NULL)
                   : trace_ray(sc, sample_ray(cam, samples, px, py, s), &rng, NULL);
    out_rgb[0] = c.x; out_rgb[1] = c.y; out_rgb[2] = c.z;
}

/* the same path, returning its event string (see ev()); result = events produced */
ORA_API int32_t ora_trace_path_events(const ora_scene *sc, const ora_camera *cam, int32_t samples, uint64_t seed,
                                      int32_t px, int32_t py, int32_t s, char *buf, int32_t cap)
{
    float rgb[3];
    ev_buf = buf; ev_n = 0; ev_cap = cap;
    ora_trace_path(sc, cam, samples, seed, px, py, s, rgb);
    ev_buf = NULL;
    return ev_n;
}

/* ------------------------------------------------------- scene generators */

/* scene/generators.rs:8-23.  Texture 0 must be the decoded earthmap (RGB8). */
ORA_API int32_t ora_gen_simple_scene(ora_sphere *spheres, ora_material *materials)
{
    const ora_material m[6] = {
        {ORA_LAMBERTIAN, {0, 0, 0}, 0.0f, 0},               /* matte1: PictureTexture */
        {ORA_LAMBERTIAN, {0.3f, 0.0f, 0.0f}, 0.0f, -1},     /* matte2 */
        {ORA_METAL, {0.8f, 0.6f, 0.2f}, 0.0f, -1},          /* metal1 */
        {ORA_DIELECTRIC, {0, 0, 0}, 1.8f, -1},              /* glass1 */
        {ORA_DIELECTRIC, {0, 0, 0}, 0.4f, -1},              /* glass2 */
        {ORA_METAL, {0.1f, 1.0f, 0.7f}, 0.1f, -1},          /* metal2 */
    };
    const ora_sphere s[6] = {
        {0.0f, 0.0f, -1.0f, 0.5f, 0},  {0.0f, -100.5f, -1.0f, 100.0f, 1}, {1.5f, 0.0f, -2.0f, 0.5f, 2},
        {-1.5f, 0.0f, -2.0f, 0.5f, 3}, {3.5f, 0.0f, -2.0f, 0.8f, 4},      {10.0f, 0.0f, -10.0f, 3.0f, 5},
    };
    memcpy(materials, m, sizeof m);
    memcpy(spheres, s, sizeof s);
    return 6;
}

/* scene/generators.rs:25-89 with Rng::with_seed(scene_seed) instead of Rng::new()
 * (generators.rs:32) and the grid bounds as parameters: the reference loops
 * a in -22..22, b in -22..221 (generators.rs:42-43).  One material per sphere.
 * Returns the sphere count, or -(needed) if capacity is too small. */
ORA_API int32_t ora_gen_random_scene(uint64_t scene_seed, int32_t a0, int32_t a1, int32_t b0, int32_t b1,
                                     ora_sphere *spheres, ora_material *materials, int32_t capacity)
{
    int n = 0;
    rng_t rng = {ora_rng_with_seed(scene_seed), NULL};
#define PUSH(CX, CY, CZ, R, KIND, AR, AG, AB, PARAM)                                             \
    do {                                                                                         \
        if (n < capacity) {                                                                      \
            ora_sphere s_ = {CX, CY, CZ, R, n};                                                  \
            ora_material m_ = {KIND, {AR, AG, AB}, PARAM, -1};                                   \
            spheres[n] = s_; materials[n] = m_;                                                  \
        }                                                                                        \
        n++;                                                                                     \
    } while (0)
    PUSH(0.0f, -1000.0f, 0.0f, 1000.0f, ORA_LAMBERTIAN, 0.5f, 0.5f, 0.5f, 0.0f);
    for (int a = a0; a < a1; a++) {
        for (int b = b0; b < b1; b++) {
            float cx = (float)a + 0.9f * rnd(&rng);          /* generators.rs:44-48 */
            float cz = (float)b + 0.9f * rnd(&rng);
            v3 center = V(cx, 0.2f, cz);
            if (vlength(vsub(center, V(3.0f, 0.2f, 0.0f))) > 0.9f) { /* generators.rs:50 */
                float choose = rnd(&rng);
                if (choose < 0.8f) {                         /* generators.rs:53-57 */
                    float x0 = rnd_range(&rng, 0.0f, 1.0f), y0 = rnd_range(&rng, 0.0f, 1.0f), z0 = rnd_range(&rng, 0.0f, 1.0f);
                    float x1 = rnd_range(&rng, 0.0f, 1.0f), y1 = rnd_range(&rng, 0.0f, 1.0f), z1 = rnd_range(&rng, 0.0f, 1.0f);
                    PUSH(cx, 0.2f, cz, 0.2f, ORA_LAMBERTIAN, x0 * x1, y0 * y1, z0 * z1, 0.0f);
                } else if (choose < 0.95f) {                 /* generators.rs:58-62 */
                    float x0 = rnd_range(&rng, 0.5f, 1.0f), y0 = rnd_range(&rng, 0.5f, 1.0f), z0 = rnd_range(&rng, 0.5f, 1.0f);
                    float fuzz = rnd_range(&rng, 0.0f, 0.5f);
                    PUSH(cx, 0.2f, cz, 0.2f, ORA_METAL, x0, y0, z0, fuzz);
                } else {                                     /* generators.rs:63-65 */
                    PUSH(cx, 0.2f, cz, 0.2f, ORA_DIELECTRIC, 0.0f, 0.0f, 0.0f, 1.5f);
                }
            }
        }
    }
    PUSH(0.0f, 1.0f, 0.0f, 1.0f, ORA_DIELECTRIC, 0.0f, 0.0f, 0.0f, 1.5f);      /* generators.rs:72-76 */
    PUSH(-4.0f, 1.0f, 0.0f, 1.0f, ORA_LAMBERTIAN, 0.4f, 0.2f, 0.1f, 0.0f);     /* generators.rs:78-82 */
    PUSH(4.0f, 1.0f, 0.0f, 1.0f, ORA_METAL, 0.7f, 0.6f, 0.5f, 0.0f);           /* generators.rs:84-88 */
#undef PUSH
    return n <= capacity ? n : -n;
}
