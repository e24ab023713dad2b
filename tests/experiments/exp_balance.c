/* Design experiment (not product code, not test code): how much of a wave's lane-time is lost
 * because a lane is bound to one pixel?  Per-pixel cost of a piece of samples (in SPLIT-step
 * units, from the oracle's counters) over a 1080p frame of the bench scene; a group of 64 pixels
 * costs 64 x its dearest pixel.  Compares 8x8 tiles with groups of 64 pixels of similar cost.
 *   gcc -O2 -ffp-contract=off -o /tmp/exp_balance tests/experiments/exp_balance.c -lm -lpthread */
#include "../../oracle/pt_oracle.c"   /* test-side experiment */
#include <stdio.h>

static int cmp_f(const void *a, const void *b) { float x = *(const float *)a, y = *(const float *)b; return x < y ? -1 : x > y; }

int main(int argc, char **argv)
{
    const int W = 1920, H = 1080, S = 64, piece = argc > 1 ? atoi(argv[1]) : 48, step = 2;
    const int cap = 1 << 15;
    ora_sphere *sp = malloc(sizeof *sp * cap); ora_material *mt = malloc(sizeof *mt * cap);
    int n = ora_gen_random_scene(0x5EED5CE7Eull, -11, 11, -11, 11, sp, mt, cap);
    ora_scene *sc = ora_scene_create(sp, n, mt, n, NULL, 0);
    float eye[3] = {12, 2, 3}, tgt[3] = {0, 0, 0}, up[3] = {0, 1, 0};
    ora_camera cam; ora_camera_new(eye, tgt, up, 20.0f, W, H, &cam);
    const int tx = W / 8, ty = H / 8;
    int ntiles = 0;
    float *cost = malloc(sizeof(float) * (size_t)(tx / step + 1) * (ty / step + 1) * 64);
    double total = 0, tile_bound = 0;
    for (int j = 0; j < ty; j += step) for (int i = 0; i < tx; i += step) {
        float mx = 0;
        for (int l = 0; l < 64; l++) {
            const int px = i * 8 + (l & 7), py = j * 8 + (l >> 3);
            ora_counters c; memset(&c, 0, sizeof c);
            for (int s = 0; s < piece; s++) {
                rng_t rng = {ora_path_stream(20260711, (uint32_t)(py * W + px), (uint32_t)s), &c};
                (void)trace_ray(sc, sample_ray(&cam, S, px, py, s), &rng, &c);
            }
            /* vector instructions, roughly: 21 per box test, 48 per sphere test, 200 per scatter, 140 per path (tails, sky) */
            const float k = 21.0f * c.aabb_tests + 48.0f * c.sphere_tests + 200.0f * (c.segments - piece) + 140.0f * piece;
            cost[(size_t)ntiles * 64 + l] = k; total += k; if (k > mx) mx = k;
        }
        tile_bound += 64.0 * mx; ntiles++;
    }
    qsort(cost, (size_t)ntiles * 64, sizeof(float), cmp_f);
    double sorted_bound = 0;
    for (int g = 0; g < ntiles; g++) sorted_bound += 64.0 * cost[(size_t)g * 64 + 63];
    printf("%d tiles, %d samples per piece: lane-time / work = %.3f for 8x8 tiles, %.3f for cost-sorted groups of 64\n", ntiles, piece,
           tile_bound / total, sorted_bound / total);
    return 0;
}
