/* Design experiment (not product code, not test code): SURVEY 8(f)-3, the node shapes that were never counted.
 * For the reference's tree (bvh.rs:13-91) walked nearer-child-first from the eye (the shipped PT_WALK_VIEW order),
 * per PATH on the bench camera: node visits ("lane-steps": what a lane of the kernel spends a step of its wave on),
 * box tests and sphere tests for
 *   binary     the shipped walk: a Split visit = 1 box test, a Leaf visit = 1 sphere test
 *   leafbox    + a Leaf's own box is tested (as one more Split-like visit) before its sphere
 *   leafpair   a Split whose two children are Leaves is ONE visit: its box test + both sphere tests
 *   wide4      Splits collapsed into nodes of up to 4 children (a child that is a Split with the larger box is opened
 *              first); a visit tests all its children's boxes (Leaf boxes included), then enters the hit ones in order
 * and the same split into camera rays (which the kernel no longer walks at all: pt_set_primary_lists) and the rest.
 * Cost model (VALU wave-instructions per visit, from the kernel's ISA): box test 21 + 3 loop control, sphere test 48,
 * a wide visit 18 per child box + 8 to order and enter them.
 *   gcc -O2 -ffp-contract=off -o /tmp/exp_wide tests/experiments/exp_wide.c -lm -lpthread && /tmp/exp_wide [1 = cfg 5 scene] */
#include "../../oracle/pt_oracle.c"
#include <stdio.h>

typedef struct xnode { int leaf, prim; v3 bmin, bmax; struct xnode *c[2]; int nw; struct xnode *w[4]; } xnode;
static int g_axis;
static int cmp_items(const void *a, const void *b)
{
    float ka = item_key((const build_item *)a, g_axis), kb = item_key((const build_item *)b, g_axis);
    return ka < kb ? -1 : (ka > kb ? 1 : 0);
}
static xnode *xbuild(build_item *it, int n)   /* the reference's rule: longest axis, volume cost, first minimum */
{
    xnode *nd = calloc(1, sizeof *nd);
    if (n == 1) { nd->leaf = 1; nd->prim = it[0].prim; nd->bmin = it[0].bmin; nd->bmax = it[0].bmax; return nd; }
    v3 wmin, wmax; enclosing_bbox(it, n, &wmin, &wmax);
    v3 e = vsub(wmax, wmin); float ext[3] = {e.x, e.y, e.z};
    int ax = 0; for (int i = 1; i < 3; i++) if (ext[i] > ext[ax]) ax = i;
    g_axis = ax; qsort(it, n, sizeof *it, cmp_items);
    float *suf = malloc(sizeof(float) * (n + 1));
    v3 mn = it[n - 1].bmin, mx = it[n - 1].bmax;
    for (int i = n - 1; i >= 1; i--) { aabb_enclose(&mn, &mx, it[i].bmin, it[i].bmax); suf[i] = aabb_volume(mn, mx); }
    mn = it[0].bmin; mx = it[0].bmax;
    int best = 0; float lowest = 3.4e38f;
    for (int i = 1; i < n; i++) {
        aabb_enclose(&mn, &mx, it[i - 1].bmin, it[i - 1].bmax);
        float c = aabb_volume(mn, mx) * i + suf[i] * (n - i);
        if (c < lowest) { lowest = c; best = i; }
    }
    free(suf);
    nd->c[0] = xbuild(it, best); nd->c[1] = xbuild(it + best, n - best);
    nd->bmin = nd->c[0]->bmin; nd->bmax = nd->c[0]->bmax; aabb_enclose(&nd->bmin, &nd->bmax, nd->c[1]->bmin, nd->c[1]->bmax);
    return nd;
}
static v3 g_eye;
static float d2eye(const xnode *n) { v3 c = vmul(vadd(n->bmin, n->bmax), vfill(0.5f)); v3 d = vsub(c, g_eye); return d.x * d.x + d.y * d.y + d.z * d.z; }
static float area(const xnode *n) { v3 e = vsub(n->bmax, n->bmin); return e.x * e.y + e.y * e.z + e.z * e.x; }
static void order_and_widen(xnode *n)
{
    if (n->leaf) return;
    if (d2eye(n->c[1]) < d2eye(n->c[0])) { xnode *t = n->c[0]; n->c[0] = n->c[1]; n->c[1] = t; }   /* nearer child first */
    order_and_widen(n->c[0]); order_and_widen(n->c[1]);
    /* up to 4 children: open the Split child with the larger box, twice */
    n->nw = 2; n->w[0] = n->c[0]; n->w[1] = n->c[1];
    while (n->nw < 4) {
        int pick = -1;
        for (int i = 0; i < n->nw; i++) if (!n->w[i]->leaf && (pick < 0 || area(n->w[i]) > area(n->w[pick]))) pick = i;
        if (pick < 0) break;
        xnode *p = n->w[pick];
        for (int i = n->nw; i > pick + 1; i--) n->w[i] = n->w[i - 1];   /* keep the walk's order */
        n->w[pick] = p->c[0]; n->w[pick + 1] = p->c[1]; n->nw++;
    }
}
typedef struct { long visits, box, sph; } cnt_t;
static cnt_t C[4];
static const ora_scene *g_sc;
static int box_hit(const xnode *n, const ray_t *ray, v3 inv, float max_t) { return aabb_intersect(n->bmin, n->bmax, ray, inv, 0.001f, max_t); }
static void leaf_test(const xnode *n, const ray_t *ray, hit_t *h, int *got)
{
    hit_t t;
    if (sphere_trace(&g_sc->spheres[n->prim], n->prim, ray, 0.001f, *got ? h->t : 3.402823466e+38f, &t, NULL)) { *h = t; *got = 1; }
}
static void walk_binary(const xnode *n, const ray_t *ray, v3 inv, hit_t *h, int *got, int leafbox, cnt_t *c)
{
    const float mx = *got ? h->t : 3.402823466e+38f;
    if (n->leaf) {
        if (leafbox) { c->visits++; c->box++; if (!box_hit(n, ray, inv, mx)) return; }
        c->visits++; c->sph++; leaf_test(n, ray, h, got); return;
    }
    c->visits++; c->box++;
    if (!box_hit(n, ray, inv, mx)) return;
    walk_binary(n->c[0], ray, inv, h, got, leafbox, c); walk_binary(n->c[1], ray, inv, h, got, leafbox, c);
}
static void walk_pair(const xnode *n, const ray_t *ray, v3 inv, hit_t *h, int *got, cnt_t *c)
{
    const float mx = *got ? h->t : 3.402823466e+38f;
    if (n->leaf) { c->visits++; c->sph++; leaf_test(n, ray, h, got); return; }
    c->visits++; c->box++;
    if (!box_hit(n, ray, inv, mx)) return;
    if (n->c[0]->leaf && n->c[1]->leaf) { c->sph += 2; leaf_test(n->c[0], ray, h, got); leaf_test(n->c[1], ray, h, got); return; }   /* same visit */
    walk_pair(n->c[0], ray, inv, h, got, c); walk_pair(n->c[1], ray, inv, h, got, c);
}
static void walk_wide(const xnode *n, const ray_t *ray, v3 inv, hit_t *h, int *got, cnt_t *c)
{   /* n is a Split whose own box has been passed */
    int hit[4];
    c->visits++;
    for (int i = 0; i < n->nw; i++) { c->box++; hit[i] = box_hit(n->w[i], ray, inv, *got ? h->t : 3.402823466e+38f); }
    for (int i = 0; i < n->nw; i++) {
        if (!hit[i]) continue;
        if (n->w[i]->leaf) { c->visits++; c->sph++; leaf_test(n->w[i], ray, h, got); }
        else walk_wide(n->w[i], ray, inv, h, got, c);   /* (its box was tested against the closest hit of a moment ago) */
    }
}
int main(int argc, char **argv)
{
    int big = argc > 1 && atoi(argv[1]);
    int cap = 20000;
    ora_sphere *sp = malloc(sizeof *sp * cap); ora_material *mt = malloc(sizeof *mt * cap);
    int n = big ? ora_gen_random_scene(0x5EED5CE7Eull, -22, 22, -22, 221, sp, mt, cap) : ora_gen_random_scene(0x5EED5CE7Eull, -11, 11, -11, 11, sp, mt, cap);
    ora_scene *sc = ora_scene_create(sp, n, mt, n, NULL, 0); g_sc = sc;
    const int W = big ? 3840 : 1920, H = big ? 2160 : 1080, S = big ? 32 : 64;
    float eye[3] = {12, 2, 3}, tgt[3] = {0, 0, 0}, up[3] = {0, 1, 0};
    ora_camera cam; ora_camera_new(eye, tgt, up, 20.0f, W, H, &cam);
    g_eye = (v3){12, 2, 3};
    build_item *it = malloc(sizeof *it * n);
    for (int i = 0; i < n; i++) { it[i].prim = i; sphere_bbox(&sc->spheres[i], &it[i].bmin, &it[i].bmax); }
    xnode *root = xbuild(it, n);
    order_and_widen(root);
    cnt_t P[4] = {{0}}, R[4] = {{0}};   /* camera rays, the rest */
    long paths = 0, mism = 0;
    for (int py = 0; py < H; py += 7) for (int px = 0; px < W; px += 7) for (int s = 0; s < S * S; s += S * S / 7 + 1) {
        uint32_t pixel = (uint32_t)(py * W + px);
        rng_t rng = {ora_path_stream(20260711, pixel, (uint32_t)s), NULL};
        ray_t ray = sample_ray(&cam, S, px, py, s);
        paths++;
        for (int seg = 0; seg < 5; seg++) {
            v3 inv = vdiv(vfill(1.0f), ray.d);
            hit_t hr; int ref = scene_trace(sc, &ray, &hr, NULL);
            cnt_t *acc = seg == 0 ? P : R;
            for (int v = 0; v < 4; v++) {
                hit_t h; int got = 0;
                if (v == 0) walk_binary(root, &ray, inv, &h, &got, 0, &acc[0]);
                else if (v == 1) walk_binary(root, &ray, inv, &h, &got, 1, &acc[1]);
                else if (v == 2) walk_pair(root, &ray, inv, &h, &got, &acc[2]);
                else { acc[3].visits++; acc[3].box++; if (box_hit(root, &ray, inv, 3.402823466e+38f)) { if (root->leaf) leaf_test(root, &ray, &h, &got); else walk_wide(root, &ray, inv, &h, &got, &acc[3]); } }
                if (got != ref || (got && h.t != hr.t)) mism++;
            }
            if (!ref) break;
            v3 a; ray_t nr;
            if (!scatter(sc, &ray, &hr, &rng, &a, &nr, NULL)) break;
            ray = nr;
        }
    }
    const char *nm[4] = {"binary", "leafbox", "leafpair", "wide4"};
    printf("%d spheres, %ld paths, closest hit differs from the reference walk's in %ld walks\n", n, paths, mism);
    printf("%-9s | camera rays: visits box sph  instr | other rays: visits box sph  instr | other rays vs binary: visits instr\n", "");
    double base_v = 0, base_i = 0;
    for (int v = 0; v < 4; v++) {
        double pv = (double)P[v].visits / paths, pb = (double)P[v].box / paths, ps = (double)P[v].sph / paths;
        double rv = (double)R[v].visits / paths, rb = (double)R[v].box / paths, rs = (double)R[v].sph / paths;
        double wide_visits = v == 3 ? rv - rs : 0;
        double pi = v == 3 ? 18 * pb + 8 * (pv - ps) + 48 * ps : 24 * pb + 48 * ps;
        double ri = v == 3 ? 18 * rb + 8 * wide_visits + 48 * rs : 24 * rb + 48 * rs;
        if (v == 0) { base_v = rv; base_i = ri; }
        printf("%-9s | %6.2f %6.2f %5.2f %6.0f | %6.2f %6.2f %5.2f %6.0f | %+5.1f %% %+5.1f %%\n", nm[v], pv, pb, ps, pi, rv, rb, rs, ri,
               100 * (rv / base_v - 1), 100 * (ri / base_i - 1));
    }
    return 0;
}
