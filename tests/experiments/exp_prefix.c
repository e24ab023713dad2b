/* Design experiment (not product code, not test code): of the node visits of a walk over a tree larger than the
 * LDS prefix, how many land in the first P nodes breadth-first?  (pt_api.cpp stages that prefix.)
 *   gcc -O2 -ffp-contract=off -o /tmp/exp_prefix tests/experiments/exp_prefix.c -lm -lpthread && /tmp/exp_prefix */
#include "../../oracle/pt_oracle.c"   /* test-side experiment */
#include <stdio.h>

#define MAXN 65536
static const bvh_node *g_nodes[MAXN]; static int g_n;
static long g_hits[MAXN];
static int rank_of(const bvh_node *n) { for (int i = 0; i < g_n; i++) if (g_nodes[i] == n) return i; return -1; }
/* pointer -> BFS rank through a small open hash */
#define HS (1 << 18)
static const bvh_node *hk[HS]; static int hv[HS];
static void hput(const bvh_node *k, int v) { size_t h = ((size_t)k >> 4) & (HS - 1); while (hk[h]) h = (h + 1) & (HS - 1); hk[h] = k; hv[h] = v; }
static int hget(const bvh_node *k) { size_t h = ((size_t)k >> 4) & (HS - 1); while (hk[h] != k) h = (h + 1) & (HS - 1); return hv[h]; }
static int xtrace(const ora_scene *sc, const bvh_node *n, const ray_t *ray, v3 inv, float min_t, float max_t, hit_t *h)
{
    g_hits[hget(n)]++;
    if (n->is_leaf) return sphere_trace(&sc->spheres[n->prim], n->prim, ray, min_t, max_t, h, NULL);
    if (!aabb_intersect(n->bmin, n->bmax, ray, inv, min_t, max_t)) return 0;
    hit_t h1;
    if (xtrace(sc, n->left, ray, inv, min_t, max_t, &h1)) {
        hit_t h2;
        if (xtrace(sc, n->right, ray, inv, min_t, h1.t, &h2)) *h = h2; else *h = h1;
        return 1;
    }
    return xtrace(sc, n->right, ray, inv, min_t, max_t, h);
}
int main(void)
{
    int cap = 20000;
    ora_sphere *sp = malloc(sizeof *sp * cap); ora_material *mt = malloc(sizeof *mt * cap);
    int n = ora_gen_random_scene(0x5EED5CE7Eull, -22, 22, -22, 221, sp, mt, cap);
    ora_scene *sc = ora_scene_create(sp, n, mt, n, NULL, 0);
    float eye[3] = {12, 2, 3}, tgt[3] = {0, 0, 0}, up[3] = {0, 1, 0};
    ora_camera cam; ora_camera_new(eye, tgt, up, 20.0f, 3840, 2160, &cam);
    g_nodes[0] = sc->root; g_n = 1;
    for (int i = 0; i < g_n; i++) { hput(g_nodes[i], i); if (!g_nodes[i]->is_leaf) { g_nodes[g_n++] = g_nodes[i]->left; g_nodes[g_n++] = g_nodes[i]->right; } }
    printf("%d spheres, %d nodes\n", n, g_n);
    (void)rank_of;
    for (int py = 0; py < 2160; py += 11) for (int px = 0; px < 3840; px += 11) for (int s = 0; s < 1024; s += 171) {
        rng_t rng = {ora_path_stream(20260711, (uint32_t)(py * 3840 + px), (uint32_t)s), NULL};
        ray_t ray = sample_ray(&cam, 32, px, py, s);
        for (int i = 0; i < 5; i++) {
            hit_t h; v3 inv = vdiv(vfill(1.0f), ray.d);
            if (!xtrace(sc, sc->root, &ray, inv, 0.001f, 3.402823466e+38f, &h)) break;
            v3 a; ray_t nr;
            if (!scatter(sc, &ray, &h, &rng, &a, &nr, NULL)) break;
            ray = nr;
        }
    }
    long total = 0, acc = 0; for (int i = 0; i < g_n; i++) total += g_hits[i];
    int marks[] = {256, 512, 1024, 2048, 3072, 4096, 8192};
    int m = 0;
    for (int i = 0; i < g_n && m < 7; i++) { acc += g_hits[i]; if (i + 1 == marks[m]) { printf("first %5d nodes breadth-first: %.1f %% of the visits\n", marks[m], 100.0 * acc / total); m++; } }
    /* the same number of nodes chosen by the surface area of the box whose test admits the walk to the node
     * (its parent's; the root's own for the root), weighted or not by the inverse distance to the eye */
    static double key[MAXN]; static int parent[MAXN]; static int order[MAXN];
    parent[0] = 0;
    for (int i = 0, q = 1; i < g_n; i++) if (!g_nodes[i]->is_leaf) { parent[q++] = i; parent[q++] = i; }
    for (int mode = 0; mode < 3; mode++) {
        for (int i = 0; i < g_n; i++) {
            const bvh_node *pn = g_nodes[parent[i]];
            v3 e = vsub(pn->bmax, pn->bmin);
            double sa = 2.0 * ((double)e.x * e.y + (double)e.y * e.z + (double)e.z * e.x);
            v3 c = vmul(vadd(pn->bmin, pn->bmax), vfill(0.5f));
            double dx = c.x - eye[0], dy = c.y - eye[1], dz = c.z - eye[2], d2 = dx * dx + dy * dy + dz * dz + 1.0;
            key[i] = mode == 0 ? sa : (mode == 1 ? sa / sqrt(d2) : sa / d2);
            order[i] = i;
        }
        /* selection by repeated partial sort is fine at this size */
        for (int i = 0; i < g_n; i++) for (int j = i + 1; j < g_n && i < 4096; j++) if (key[order[j]] > key[order[i]]) { int t = order[i]; order[i] = order[j]; order[j] = t; }
        acc = 0; m = 0;
        for (int i = 0; i < 4096 && m < 6; i++) { acc += g_hits[order[i]]; if (i + 1 == marks[m]) { printf("mode %d, the %5d nodes of largest key: %.1f %% of the visits\n", mode, marks[m], 100.0 * acc / total); m++; } }
    }
    return 0;
}
