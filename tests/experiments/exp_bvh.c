/* Design experiment (not product code, not test code): how many box / sphere tests per path
 * would other BVH builds and child orders need, and how often would the closest hit differ
 * from the reference order's?  Reuses the oracle's static functions by inclusion.
 *   gcc -O2 -ffp-contract=off -o /tmp/exp_bvh tests/experiments/exp_bvh.c -lm -lpthread */
#include "../../oracle/pt_oracle.c"   /* test-side experiment: the only place besides tests/ proper that may */
#include <stdio.h>

typedef struct xnode { int leaf, prim, axis; v3 bmin, bmax; struct xnode *l, *r; } xnode;
static float sa(v3 mn, v3 mx) { v3 e = vsub(mx, mn); return 2.0f * (e.x * e.y + e.y * e.z + e.z * e.x); }
static int g_axis;
static int cmp_items(const void *a, const void *b)
{
    float ka = item_key((const build_item *)a, g_axis), kb = item_key((const build_item *)b, g_axis);
    return ka < kb ? -1 : (ka > kb ? 1 : 0);
}
/* mode 0: reference rule (longest axis, volume cost); 1: all axes, surface-area cost */
static xnode *xbuild(build_item *it, int n, int mode)
{
    xnode *nd = calloc(1, sizeof *nd);
    if (n == 1) { nd->leaf = 1; nd->prim = it[0].prim; nd->bmin = it[0].bmin; nd->bmax = it[0].bmax; return nd; }
    v3 wmin, wmax; enclosing_bbox(it, n, &wmin, &wmax);
    v3 e = vsub(wmax, wmin); float ext[3] = {e.x, e.y, e.z};
    int longest = 0; for (int i = 1; i < 3; i++) if (ext[i] > ext[longest]) longest = i;
    int best_axis = longest, best = 0; float lowest = 3.4e38f;
    float *suf = malloc(sizeof(float) * (n + 1));
    for (int ax = 0; ax < 3; ax++) {
        if (mode == 0 && ax != longest) continue;
        g_axis = ax; qsort(it, n, sizeof *it, cmp_items);
        v3 mn = it[n - 1].bmin, mx = it[n - 1].bmax;
        for (int i = n - 1; i >= 1; i--) { aabb_enclose(&mn, &mx, it[i].bmin, it[i].bmax); suf[i] = mode ? sa(mn, mx) : aabb_volume(mn, mx); }
        mn = it[0].bmin; mx = it[0].bmax;
        for (int i = 1; i < n; i++) {
            aabb_enclose(&mn, &mx, it[i - 1].bmin, it[i - 1].bmax);
            float c = (mode ? sa(mn, mx) : aabb_volume(mn, mx)) * i + suf[i] * (n - i);
            if (c < lowest) { lowest = c; best = i; best_axis = ax; }
        }
    }
    free(suf);
    g_axis = best_axis; qsort(it, n, sizeof *it, cmp_items);
    nd->axis = best_axis;
    nd->l = xbuild(it, best, mode); nd->r = xbuild(it + best, n - best, mode);
    nd->bmin = nd->l->bmin; nd->bmax = nd->l->bmax; aabb_enclose(&nd->bmin, &nd->bmax, nd->r->bmin, nd->r->bmax);
    return nd;
}
static long nS, nL;
static int g_order;  /* 0 left first, 1 near first by sign of d[axis] */
static int xtrace(const ora_scene *sc, const xnode *n, const ray_t *ray, v3 inv, float min_t, float max_t, hit_t *h)
{
    if (n->leaf) { nL++; return sphere_trace(&sc->spheres[n->prim], n->prim, ray, min_t, max_t, h, NULL); }
    nS++;
    if (!aabb_intersect(n->bmin, n->bmax, ray, inv, min_t, max_t)) return 0;
    const xnode *a = n->l, *b = n->r;
    if (g_order) { float d = n->axis == 0 ? ray->d.x : (n->axis == 1 ? ray->d.y : ray->d.z); if (d < 0) { a = n->r; b = n->l; } }
    hit_t h1;
    if (xtrace(sc, a, ray, inv, min_t, max_t, &h1)) {
        hit_t h2;
        if (xtrace(sc, b, ray, inv, min_t, h1.t, &h2)) *h = h2; else *h = h1;
        return 1;
    }
    return xtrace(sc, b, ray, inv, min_t, max_t, h);
}
static const xnode *g_root;
static long mism, segs;
static v3 xtrace_ray(const ora_scene *sc, ray_t ray, rng_t *rng)
{
    v3 att = vfill(1.0f);
    for (int i = 0; i < 5; i++) {
        hit_t h, hr; segs++;
        v3 inv = vdiv(vfill(1.0f), ray.d);
        int got = xtrace(sc, g_root, &ray, inv, 0.001f, 3.402823466e+38f, &h);
        int ref = scene_trace(sc, &ray, &hr, NULL);
        if (got != ref || (got && (h.prim != hr.prim || h.t != hr.t))) mism++;
        if (ref) {
            v3 a; ray_t nr;
            if (scatter(sc, &ray, &hr, rng, &a, &nr, NULL)) { att = vmul(att, a); ray = nr; } else return vfill(0.0f);
        } else break;
    }
    return att;
}
int main(int argc, char **argv)
{
    int big = argc > 1 && atoi(argv[1]);
    int cap = 20000;
    ora_sphere *sp = malloc(sizeof *sp * cap); ora_material *mt = malloc(sizeof *mt * cap);
    int n = big ? ora_gen_random_scene(0x5EED5CE7Eull, -22, 22, -22, 221, sp, mt, cap)
                : ora_gen_random_scene(0x5EED5CE7Eull, -11, 11, -11, 11, sp, mt, cap);
    ora_scene *sc = ora_scene_create(sp, n, mt, n, NULL, 0);
    float eye[3] = {12, 2, 3}, tgt[3] = {0, 0, 0}, up[3] = {0, 1, 0};
    ora_camera cam; ora_camera_new(eye, tgt, up, 20.0f, 1920, 1080, &cam);
    printf("%d spheres\n", n);
    for (int mode = 0; mode < 2; mode++) {
        build_item *it = malloc(sizeof *it * n);
        for (int i = 0; i < n; i++) { it[i].prim = i; sphere_bbox(&sc->spheres[i], &it[i].bmin, &it[i].bmax); }
        g_root = xbuild(it, n, mode);
        for (g_order = 0; g_order < 2; g_order++) {
            nS = nL = mism = segs = 0; long paths = 0;
            for (int py = 0; py < 1080; py += 7) for (int px = 0; px < 1920; px += 7) for (int s = 0; s < 256; s += 37) {
                uint32_t pixel = (uint32_t)(py * 1920 + px);
                rng_t rng = {ora_path_stream(20260711, pixel, (uint32_t)s), NULL};
                xtrace_ray(sc, sample_ray(&cam, 16, px, py, s), &rng); paths++;
            }
            printf("build %s order %s: S/path %.1f L/path %.2f  segs/path %.2f  mismatching segments %ld of %ld\n",
                   mode ? "SAH-3axis" : "reference", g_order ? "near-first" : "left-first", (double)nS / paths,
                   (double)nL / paths, (double)segs / paths, mism, segs);
        }
    }
    return 0;
}
