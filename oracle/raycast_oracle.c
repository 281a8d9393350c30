/*
 * CPU oracle for the ray cast (TEST INFRASTRUCTURE, see oracle/__init__.py).
 *
 * Reference boundary: LAB/utils/warp/kernels.py:13-75 -> wp.mesh_query_ray of warp-lang 1.8.0
 * (environment.yml:285), which is NOT under /root/reference and not installed here; the reference holds no
 * golden vector for it: ray-hit parity is UNPINNED by the reference.  This file restates
 * the published watertight ray/triangle test (Woop, Benthin, Wald, JCGT 2013) that mesh_query_ray uses
 * -- the arithmetic lives in isaaclab_amd/csrc/raycast_device.hpp and is compiled here for the host with
 * gcc -ffp-contract=off, so CPU and GPU evaluate identical IEEE expression trees -- and defines the
 * answer as: closest hit over ALL triangles (brute force), ties on t broken by the smallest face id.
 * A uniform-grid variant (own builder, bounding-box registration) is provided for the CPU baseline at
 * full mesh size; tests check it against the brute-force answer.
 */
#define MDP_RC_HOST 1
#include "../isaaclab_amd/csrc/raycast_device.hpp"
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static void write_hit(const mdp_mesh_view_t* m, int64_t i, const float* s, const float* d, mdp_hit_t h, float* hits,
                      float* dist, float* normal, int32_t* face) {
    if (h.face >= 0) {
        hits[3 * i] = s[0] + h.t * d[0];
        hits[3 * i + 1] = s[1] + h.t * d[1];
        hits[3 * i + 2] = s[2] + h.t * d[2];
        if (dist) dist[i] = h.t;
        if (normal) mdp_rc_normal(m->tri_verts + 9 * (int64_t)h.face, normal + 3 * i, normal + 3 * i + 1, normal + 3 * i + 2);
        if (face) face[i] = h.face;
    } else {
        hits[3 * i] = hits[3 * i + 1] = hits[3 * i + 2] = INFINITY;
        if (dist) dist[i] = INFINITY;
        if (normal) normal[3 * i] = normal[3 * i + 1] = normal[3 * i + 2] = INFINITY;
        if (face) face[i] = -1;
    }
}

/* tri_verts: (T,9) pre-gathered triangle vertices */
void oracle_raycast_brute(const float* tri_verts, int32_t num_tris, int64_t num_rays, const float* starts,
                          const float* dirs, float max_dist, float* hits, float* dist, float* normal, int32_t* face) {
    mdp_mesh_view_t m;
    memset(&m, 0, sizeof(m));
    m.tri_verts = tri_verts;
    m.num_tris = num_tris;
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < num_rays; ++i) {
        mdp_ray_t r = mdp_rc_make_ray(starts[3 * i], starts[3 * i + 1], starts[3 * i + 2], dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]);
        mdp_hit_t h = mdp_rc_brute(&m, &r, max_dist);
        write_hit(&m, i, starts + 3 * i, dirs + 3 * i, h, hits, dist, normal, face);
    }
}

/* ---- grid variant (CPU baseline at full size) ---- */
typedef struct {
    mdp_mesh_view_t view;
    int32_t* cell_start;
    int32_t* cell_tris;
} oracle_grid_t;

oracle_grid_t* oracle_grid_build(const float* tri_verts, int32_t num_tris, float cell) {
    oracle_grid_t* g = (oracle_grid_t*)calloc(1, sizeof(oracle_grid_t));
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = 0; i < (int64_t)num_tris * 3; ++i)
        for (int a = 0; a < 3; ++a) {
            float v = tri_verts[3 * i + a];
            if (v < mn[a]) mn[a] = v;
            if (v > mx[a]) mx[a] = v;
        }
    float gx0 = mn[0] - 0.5f * cell, gy0 = mn[1] - 0.5f * cell;
    int ncx = (int)floorf((mx[0] - gx0) / cell) + 1, ncy = (int)floorf((mx[1] - gy0) / cell) + 1;
    if (ncx < 1) ncx = 1;
    if (ncy < 1) ncy = 1;
    float pad = 1e-3f * cell + 1e-6f;
    int64_t ncell = (int64_t)ncx * ncy;
    int32_t* start = (int32_t*)calloc(ncell + 1, sizeof(int32_t));
    for (int pass = 0; pass < 2; ++pass) {
        int32_t* cursor = NULL;
        if (pass == 1) {
            int32_t run = 0;
            for (int64_t c = 0; c <= ncell; ++c) { int32_t v = start[c]; start[c] = run; run += v; }
            g->cell_tris = (int32_t*)malloc(sizeof(int32_t) * (size_t)(start[ncell] > 0 ? start[ncell] : 1));
            cursor = (int32_t*)malloc(sizeof(int32_t) * (size_t)ncell);
            memcpy(cursor, start, sizeof(int32_t) * (size_t)ncell);
        }
        for (int32_t f = 0; f < num_tris; ++f) {
            const float* t = tri_verts + 9 * (int64_t)f;
            float x0 = fminf(t[0], fminf(t[3], t[6])) - pad, x1 = fmaxf(t[0], fmaxf(t[3], t[6])) + pad;
            float y0 = fminf(t[1], fminf(t[4], t[7])) - pad, y1 = fmaxf(t[1], fmaxf(t[4], t[7])) + pad;
            int cx0 = (int)floorf((x0 - gx0) / cell), cx1 = (int)floorf((x1 - gx0) / cell);
            int cy0 = (int)floorf((y0 - gy0) / cell), cy1 = (int)floorf((y1 - gy0) / cell);
            if (cx0 < 0) cx0 = 0;
            if (cy0 < 0) cy0 = 0;
            if (cx1 > ncx - 1) cx1 = ncx - 1;
            if (cy1 > ncy - 1) cy1 = ncy - 1;
            for (int cy = cy0; cy <= cy1; ++cy)
                for (int cx = cx0; cx <= cx1; ++cx) {
                    int64_t c = (int64_t)cy * ncx + cx;
                    if (pass == 0) start[c]++;
                    else g->cell_tris[cursor[c]++] = f;
                }
        }
        free(cursor);
    }
    g->cell_start = start;
    g->view.tri_verts = tri_verts;
    g->view.num_tris = num_tris;
    g->view.cell_start = start;
    g->view.cell_tris = g->cell_tris;
    g->view.gx0 = gx0; g->view.gy0 = gy0; g->view.cell = cell; g->view.inv_cell = 1.0f / cell;
    g->view.ncx = ncx; g->view.ncy = ncy;
    g->view.zmin = mn[2]; g->view.zmax = mx[2];
    g->view.accel = 1;
    return g;
}

void oracle_grid_free(oracle_grid_t* g) {
    if (!g) return;
    free(g->cell_start);
    free(g->cell_tris);
    free(g);
}

void oracle_raycast_grid(const oracle_grid_t* g, int64_t num_rays, const float* starts, const float* dirs,
                         float max_dist, float* hits, float* dist, float* normal, int32_t* face, int num_threads) {
#ifdef _OPENMP
    if (num_threads > 0) omp_set_num_threads(num_threads);
#endif
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < num_rays; ++i) {
        mdp_ray_t r = mdp_rc_make_ray(starts[3 * i], starts[3 * i + 1], starts[3 * i + 2], dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]);
        mdp_hit_t h = mdp_rc_grid(&g->view, &r, max_dist);
        write_hit(&g->view, i, starts + 3 * i, dirs + 3 * i, h, hits, dist, normal, face);
    }
}
