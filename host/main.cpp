// main.cpp -- plays the role of the reference's src/main.rs (:73-122) on top of the C ABI:
// hard-coded camera (main.rs:80-87), scene choice (main.rs:91-92), render, write the image.
// Progress lines like the reporter thread's (main.rs:41-71); output.png like main.rs:118-121
// (uncompressed PNG, written by pt::write_png).
//   pt_main [random|rtow|simple <earthmap.ppm>] [width height samples]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>

#include "pt_host.hpp"

static pt::SharedTexture load_ppm(const char *path)
{
    std::ifstream f(path, std::ios::binary);
    std::string magic; int w = 0, h = 0, maxv = 0;
    f >> magic >> w >> h >> maxv;
    f.get();
    if (magic != "P6" || maxv != 255) throw std::runtime_error("texture must be a binary PPM (P6, 8 bit)");
    std::vector<uint8_t> texels((size_t)w * h * 3);
    f.read((char *)texels.data(), (std::streamsize)texels.size());
    return pt::PictureTexture::from_rgb8(w, h, std::move(texels));
}

int main(int argc, char **argv)
{
    try {
        const char *which = argc > 1 ? argv[1] : "rtow";
        int arg = 2;
        pt::Scene scene;
        if (!std::strcmp(which, "simple")) {
            if (argc < 3) { std::fprintf(stderr, "simple needs the decoded earthmap as a PPM\n"); return 2; }
            pt::generators::simple_scene(scene, load_ppm(argv[arg++]));
        } else if (!std::strcmp(which, "random")) pt::generators::random_scene(scene, 0x5EED5CE7Eull);
        else pt::generators::random_scene(scene, 0x5EED5CE7Eull, -11, 11, -11, 11);
        const int width = argc > arg ? std::atoi(argv[arg]) : 3840 - 200;          // main.rs:76-78
        const int height = argc > arg + 1 ? std::atoi(argv[arg + 1]) : 2160 - 200;
        const int samples = argc > arg + 2 ? std::atoi(argv[arg + 2]) : 16;
        const pt::Camera camera = pt::Camera::make({12.0f, 2.0f, 3.0f}, {0.0f, 0.0f, 0.0f}, {0.0f, 1.0f, 0.0f}, 20.0f, width, height);
        const auto t0 = std::chrono::steady_clock::now();
        pt::Raytracer raytracer = pt::Raytracer::make(camera, scene, samples);      // builds the BVH (raytracer.rs:37)
        const auto t1 = std::chrono::steady_clock::now();
        std::printf("Constructing BVH for %zu objects... done in %.3fs.\n", scene.objects.size(),
                    std::chrono::duration<double>(t1 - t0).count());               // scene/mod.rs:51-59
        pt::ParallelRenderer renderer(0);
        std::vector<uint8_t> buffer(raytracer.pixel_count() * 3);                   // main.rs:98
        const int passes = samples * samples >= 64 ? 8 : 1;
        renderer.render_progress(raytracer, buffer.data(), 20260711ull, passes, [&](int done, int total) {
            const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
            std::printf("Progress: %.1f%%, %.1fs elapsed, ~%.1fs left\n", 100.0 * done / total, el, el * (total - done) / done);
            return false;
        });
        const auto t2 = std::chrono::steady_clock::now();
        const double paths = (double)raytracer.pixel_count() * samples * samples;
        const double render_s = std::chrono::duration<double>(t2 - t1).count();
        std::printf("%d x %d @ %d spp: %.3fs elapsed incl. upload (%.0f Mpaths/s)\n", width, height, samples * samples, render_s,
                    paths / render_s / 1e6);
        if (!pt::write_png("output.png", buffer.data(), width, height)) throw std::runtime_error("cannot write output.png");
        return 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}
