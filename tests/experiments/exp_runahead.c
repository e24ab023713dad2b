/* Design experiment (not product code, not test code): what a lane bound to one pixel costs when a tile's sample
 * range is cut into pieces that all 64 lanes must finish before the next piece starts (today: a piece costs
 * 64 x its dearest pixel), against lanes that RUN AHEAD into their own pixel's next samples while the slowest
 * lane finishes the piece (the tile then costs 64 x its dearest pixel over ALL samples).
 *   gcc -O2 -ffp-contract=off -o /tmp/exp_runahead tests/experiments/exp_runahead.c -lm -lpthread
 *   /tmp/exp_runahead [pieces=17] [piece_len=241] [tile step=12] */
#include "../../oracle/pt_oracle.c"   /* test-side experiment */
#include <stdio.h>

int main(int argc, char **argv)
{
    const int W = 1920, H = 1080, S = 64;
    const int pieces = argc > 1 ? atoi(argv[1]) : 17, piece = argc > 2 ? atoi(argv[2]) : 241, step = argc > 3 ? atoi(argv[3]) : 12;
    const int cap = 1 << 15;
    ora_sphere *sp = malloc(sizeof *sp * cap); ora_material *mt = malloc(sizeof *mt * cap);
    int n = ora_gen_random_scene(0x5EED5CE7Eull, -11, 11, -11, 11, sp, mt, cap);
    ora_scene *sc = ora_scene_create(sp, n, mt, n, NULL, 0);
    float eye[3] = {12, 2, 3}, tgt[3] = {0, 0, 0}, up[3] = {0, 1, 0};
    ora_camera cam; ora_camera_new(eye, tgt, up, 20.0f, W, H, &cam);
    const int tx = W / 8, ty = H / 8;
    double total = 0, piece_bound = 0, tile_bound = 0;
    int ntiles = 0;
    for (int j = 0; j < ty; j += step) for (int i = (j / step) % step; i < tx; i += step) {
        double sum_p[64]; memset(sum_p, 0, sizeof sum_p);
        for (int k = 0; k < pieces; k++) {
            double mx = 0;
            for (int l = 0; l < 64; l++) {
                const int px = i * 8 + (l & 7), py = j * 8 + (l >> 3);
                ora_counters c; memset(&c, 0, sizeof c);
                for (int s = k * piece; s < (k + 1) * piece; s++) {
                    rng_t rng = {ora_path_stream(20260711, (uint32_t)(py * W + px), (uint32_t)s), &c};
                    (void)trace_ray(sc, sample_ray(&cam, S, px, py, s), &rng, &c);
                }
                const double w = 21.0 * c.aabb_tests + 48.0 * c.sphere_tests + 200.0 * (c.segments - piece) + 140.0 * piece;
                sum_p[l] += w; total += w; if (w > mx) mx = w;
            }
            piece_bound += 64.0 * mx;
        }
        double mx = 0;
        for (int l = 0; l < 64; l++) if (sum_p[l] > mx) mx = sum_p[l];
        tile_bound += 64.0 * mx; ntiles++;
    }
    printf("%d tiles, %d pieces of %d samples: lane-time / work = %.4f piece by piece (today), %.4f with lanes running ahead\n",
           ntiles, pieces, piece, piece_bound / total, tile_bound / total);
    return 0;
}
