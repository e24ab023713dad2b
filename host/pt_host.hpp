// pt_host.hpp -- C++ mirror of the host side the Rust crate keeps (SURVEY.md section 1,
// layers L5/L2 builder API), on top of the C ABI of include/pt_amd.h.  Same names and
// argument meaning as the reference: Vec3 (src/math/vec.rs:75), Camera::new
// (src/math/camera.rs:14), Scene::{new,add} (src/scene/mod.rs:17,46), Sphere::new
// (src/traceable/sphere.rs:27), Lambertian::{new,new_solid} / Metal::new / Dielectric::new
// (src/material/*.rs), SolidTexture / PictureTexture (src/texture/*.rs), Raytracer::new
// (src/raytracer.rs:36) and ParallelRenderer::{new,render} (src/parallel_renderer/mod.rs:73,155).
// The objects only record what the builders were given; Raytracer packs them into the
// plain arrays that cross the ABI -- the job the Rust shim of INTEGRATION.md does by
// walking Arc<dyn Material> / BvhNode.  Errors become exceptions HERE (host side); the
// ABI itself only returns status codes.
#pragma once
#include <cstdint>
#include <cstdio>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../include/pt_amd.h"

namespace pt {

struct Vec3 { float x, y, z; static Vec3 make(float x, float y, float z) { return Vec3{x, y, z}; } };

struct Texture { virtual ~Texture() = default; };
using SharedTexture = std::shared_ptr<Texture>;
struct SolidTexture : Texture {
    Vec3 color;
    static SharedTexture make(Vec3 c) { auto t = std::make_shared<SolidTexture>(); t->color = c; return t; }
};
struct PictureTexture : Texture {   // decoded RGB8, row-major from the top row (picture.rs:9-13)
    int32_t width = 0, height = 0;
    std::vector<uint8_t> rgb8;
    static SharedTexture from_rgb8(int32_t w, int32_t h, std::vector<uint8_t> texels)
    {
        if (w < 1 || h < 1 || texels.size() != (size_t)w * h * 3) throw std::invalid_argument("bad texture");
        auto t = std::make_shared<PictureTexture>();
        t->width = w; t->height = h; t->rgb8 = std::move(texels);
        return t;
    }
};

struct Material { virtual ~Material() = default; virtual pt_material describe(std::map<const Texture *, int32_t> &, std::vector<pt_texture> &) const = 0; };
using SharedMaterial = std::shared_ptr<Material>;

struct Lambertian : Material {
    Vec3 albedo{0, 0, 0};
    SharedTexture texture;   // PictureTexture or null
    static SharedMaterial make(const SharedTexture &tex)     // Lambertian::new (lambertian.rs:18)
    {
        auto m = std::make_shared<Lambertian>();
        if (auto solid = std::dynamic_pointer_cast<SolidTexture>(tex)) m->albedo = solid->color;   // colour ignores (u, v)
        else m->texture = tex;
        return m;
    }
    static SharedMaterial make_solid(Vec3 c) { auto m = std::make_shared<Lambertian>(); m->albedo = c; return m; }
    pt_material describe(std::map<const Texture *, int32_t> &index, std::vector<pt_texture> &out) const override
    {
        int32_t tex = -1;
        if (texture) {
            auto it = index.find(texture.get());
            if (it == index.end()) {
                const auto *p = static_cast<const PictureTexture *>(texture.get());
                it = index.emplace(texture.get(), (int32_t)out.size()).first;
                out.push_back(pt_texture{p->width, p->height, p->rgb8.data()});
            }
            tex = it->second;
        }
        return pt_material{PT_LAMBERTIAN, {albedo.x, albedo.y, albedo.z}, 0.0f, tex};
    }
};
struct Metal : Material {
    Vec3 albedo; float fuziness;
    static SharedMaterial make(Vec3 a, float f) { auto m = std::make_shared<Metal>(); m->albedo = a; m->fuziness = f; return m; }
    pt_material describe(std::map<const Texture *, int32_t> &, std::vector<pt_texture> &) const override
    { return pt_material{PT_METAL, {albedo.x, albedo.y, albedo.z}, fuziness, -1}; }
};
struct Dielectric : Material {
    float ref_idx;
    static SharedMaterial make(float ri) { auto m = std::make_shared<Dielectric>(); m->ref_idx = ri; return m; }
    pt_material describe(std::map<const Texture *, int32_t> &, std::vector<pt_texture> &) const override
    { return pt_material{PT_DIELECTRIC, {0, 0, 0}, ref_idx, -1}; }
};

struct Sphere {   // the only Traceable (traceable/mod.rs)
    Vec3 center; float radius; SharedMaterial material;
    static Sphere make(Vec3 c, float r, const SharedMaterial &m) { return Sphere{c, r, m}; }
};

struct Scene {
    std::vector<Sphere> objects;               // primitive id = insertion order (scene/mod.rs:46-48)
    void add(Sphere s) { objects.push_back(std::move(s)); }
};

struct Camera {
    pt_camera pod{};
    static Camera make(Vec3 eyes, Vec3 target, Vec3 up, float fov, int32_t width, int32_t height)
    {
        Camera c;
        const float e[3] = {eyes.x, eyes.y, eyes.z}, t[3] = {target.x, target.y, target.z}, u[3] = {up.x, up.y, up.z};
        if (pt_camera_look_at(e, t, u, fov, width, height, &c.pod) != PT_OK) throw std::invalid_argument("Camera::new");
        return c;
    }
};

// scene::generators (src/scene/generators.rs), seeded
namespace generators {
inline void from_arrays(Scene &scene, const std::vector<pt_sphere> &s, const std::vector<pt_material> &m, const SharedTexture &tex0)
{
    std::vector<SharedMaterial> mats;
    for (const pt_material &x : m) {
        const Vec3 a{x.albedo[0], x.albedo[1], x.albedo[2]};
        if (x.kind == PT_LAMBERTIAN) mats.push_back(x.texture >= 0 ? Lambertian::make(tex0) : Lambertian::make_solid(a));
        else if (x.kind == PT_METAL) mats.push_back(Metal::make(a, x.param));
        else mats.push_back(Dielectric::make(x.param));
    }
    for (const pt_sphere &x : s) scene.add(Sphere::make(Vec3{x.cx, x.cy, x.cz}, x.radius, mats[(size_t)x.material]));
}
inline void simple_scene(Scene &scene, const SharedTexture &earthmap)     // generators.rs:8-23
{
    std::vector<pt_sphere> s(6); std::vector<pt_material> m(6);
    pt_gen_simple_scene(s.data(), m.data());
    from_arrays(scene, s, m, earthmap);
}
inline void random_scene(Scene &scene, uint64_t seed, int a0 = -22, int a1 = 22, int b0 = -22, int b1 = 221)   // generators.rs:25-89
{
    const int32_t n = -pt_gen_random_scene(seed, a0, a1, b0, b1, nullptr, nullptr, 0);
    std::vector<pt_sphere> s((size_t)n); std::vector<pt_material> m((size_t)n);
    if (pt_gen_random_scene(seed, a0, a1, b0, b1, s.data(), m.data(), n) != n) throw std::runtime_error("random_scene");
    from_arrays(scene, s, m, nullptr);
}
}  // namespace generators

// Raytracer::new (raytracer.rs:36-44): owns camera + packed scene; BVH built on the host
struct Raytracer {
    Camera camera; int32_t samples;
    std::vector<pt_sphere> spheres; std::vector<pt_material> materials; std::vector<pt_texture> textures;
    std::vector<pt_bvh_node> nodes;
    std::vector<SharedMaterial> keep;          // keeps texel storage alive
    static Raytracer make(Camera camera, const Scene &scene, int32_t samples)
    {
        if (scene.objects.empty()) throw std::invalid_argument("Cannot build BVH without any objects.");   // bvh.rs:19
        Raytracer r; r.camera = camera; r.samples = samples;
        std::map<const Material *, int32_t> mat_index; std::map<const Texture *, int32_t> tex_index;
        for (const Sphere &s : scene.objects) {
            auto it = mat_index.find(s.material.get());
            if (it == mat_index.end()) {       // shared Arc<dyn Material>: one entry
                it = mat_index.emplace(s.material.get(), (int32_t)r.materials.size()).first;
                r.materials.push_back(s.material->describe(tex_index, r.textures));
                r.keep.push_back(s.material);
            }
            r.spheres.push_back(pt_sphere{s.center.x, s.center.y, s.center.z, s.radius, it->second});
        }
        r.nodes.resize(2 * r.spheres.size() - 1);
        if (pt_bvh_build(r.spheres.data(), (int32_t)r.spheres.size(), r.nodes.data(), (int32_t)r.nodes.size()) != (int32_t)r.nodes.size())
            throw std::runtime_error("pt_bvh_build");
        return r;
    }
    int32_t width() const { return camera.pod.width; }
    int32_t height() const { return camera.pod.height; }
    size_t pixel_count() const { return (size_t)width() * height(); }
};

using Pixel = uint8_t[3];   // raytracer.rs:9

// ParallelRenderer (parallel_renderer/mod.rs): the worker pool is one MI355X
class ParallelRenderer {
    pt_context *ctx_ = nullptr;
    void check(int rc) const { if (rc != PT_OK) throw std::runtime_error(pt_last_error(ctx_)); }
public:
    explicit ParallelRenderer(int device = 0) { if (pt_create(device, &ctx_) != PT_OK) throw std::runtime_error(pt_last_error(nullptr)); }
    ~ParallelRenderer() { pt_destroy(ctx_); }
    ParallelRenderer(const ParallelRenderer &) = delete;
    ParallelRenderer &operator=(const ParallelRenderer &) = delete;
    // renderer.render(&context, &mut buffer, ..) of main.rs:110-114: fills W*H x [u8;3], row-major, top row first
    void render(const Raytracer &rt, uint8_t *buffer, uint64_t seed, float *radiance = nullptr, int32_t *hit_id = nullptr)
    {
        check(pt_upload_scene(ctx_, rt.spheres.data(), (int32_t)rt.spheres.size(), rt.materials.data(), (int32_t)rt.materials.size(),
                              rt.textures.data(), (int32_t)rt.textures.size(), rt.nodes.data(), (int32_t)rt.nodes.size()));
        check(pt_set_camera(ctx_, &rt.camera.pod));
        check(pt_render(ctx_, rt.samples, seed, 0, rt.samples * rt.samples, buffer, radiance, hit_id));
    }
    // the same frame in `passes` launches with a progress report after each: Statistics::pixels_done
    // + the reporter thread of main.rs:41-71 (progress(done, total) returns true to cancel)
    template <typename F>
    void render_progress(const Raytracer &rt, uint8_t *buffer, uint64_t seed, int passes, F progress)
    {
        check(pt_upload_scene(ctx_, rt.spheres.data(), (int32_t)rt.spheres.size(), rt.materials.data(), (int32_t)rt.materials.size(),
                              rt.textures.data(), (int32_t)rt.textures.size(), rt.nodes.data(), (int32_t)rt.nodes.size()));
        check(pt_set_camera(ctx_, &rt.camera.pod));
        auto thunk = [](void *user, int32_t done, int32_t total) -> int { return (*(F *)user)(done, total) ? 1 : 0; };
        check(pt_render_progress(ctx_, rt.samples, seed, passes, thunk, &progress, buffer, nullptr));
    }
    float last_path_kernel_ms() { return pt_last_path_kernel_ms(ctx_); }
};

// Minimal PNG writer (8-bit RGB, zlib "stored" blocks: no compression) for the output.png of
// main.rs:118-121; the reference encodes through the `png` crate, any decoder reads both.
inline bool write_png(const char *path, const uint8_t *rgb8, int width, int height)
{
    auto crc32 = [](const std::vector<uint8_t> &d, size_t from) {
        uint32_t c = 0xffffffffu;
        for (size_t i = from; i < d.size(); i++) {
            c ^= d[i];
            for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0xedb88320u & (0u - (c & 1u)));
        }
        return c ^ 0xffffffffu;
    };
    auto be32 = [](std::vector<uint8_t> &d, uint32_t v) { for (int s = 24; s >= 0; s -= 8) d.push_back((uint8_t)(v >> s)); };
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    auto chunk = [&](const char *type, const std::vector<uint8_t> &body) {
        be32(out, (uint32_t)body.size());
        const size_t from = out.size();
        out.insert(out.end(), type, type + 4);
        out.insert(out.end(), body.begin(), body.end());
        be32(out, crc32(out, from));
    };
    std::vector<uint8_t> ihdr;
    be32(ihdr, (uint32_t)width); be32(ihdr, (uint32_t)height);
    ihdr.insert(ihdr.end(), {8, 2, 0, 0, 0});                       // 8 bit, colour type 2 (RGB), no interlace
    chunk("IHDR", ihdr);
    std::vector<uint8_t> raw;                                        // filter byte 0 + row
    raw.reserve((size_t)height * ((size_t)width * 3 + 1));
    for (int y = 0; y < height; y++) {
        raw.push_back(0);
        raw.insert(raw.end(), rgb8 + (size_t)y * width * 3, rgb8 + (size_t)(y + 1) * width * 3);
    }
    std::vector<uint8_t> z = {0x78, 0x01};                           // zlib header, then stored deflate blocks
    uint32_t a = 1, b = 0;                                           // Adler-32 of raw
    for (size_t pos = 0; pos < raw.size() || pos == 0;) {
        const size_t n = raw.size() - pos < 65535 ? raw.size() - pos : 65535;
        z.push_back(pos + n >= raw.size() ? 1 : 0);
        z.push_back((uint8_t)(n & 255)); z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)(~n & 255)); z.push_back((uint8_t)((~n >> 8) & 255));
        for (size_t i = pos; i < pos + n; i++) { a = (a + raw[i]) % 65521u; b = (b + a) % 65521u; }
        z.insert(z.end(), raw.begin() + (std::ptrdiff_t)pos, raw.begin() + (std::ptrdiff_t)(pos + n));
        pos += n;
        if (n == 0) break;
    }
    be32(z, (b << 16) | a);
    chunk("IDAT", z);
    chunk("IEND", {});
    std::FILE *f = std::fopen(path, "wb");
    if (!f) return false;
    const bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
    return std::fclose(f) == 0 && ok;
}

}  // namespace pt
