/*
 * CPU oracle for the ray cast -- TEST INFRASTRUCTURE (see oracle/__init__.py); never linked into the product.
 *
 * Reference boundary: LAB/utils/warp/ops.py:24-127 (raycast_mesh) -> LAB/utils/warp/kernels.py:13-75 ->
 * wp.mesh_query_ray of warp-lang (pinned ==1.8.0 in environment.yml:285).  warp-lang is a third-party dependency that
 * is NOT under /root/reference and not installed in this image, and the reference's tests hold no golden vector at this
 * boundary: PARITY UNPINNED by the reference.  What is restated here is the published algorithm that
 * mesh_query_ray documents: S. Woop, C. Benthin, I. Wald, "Watertight Ray/Triangle Intersection", JCGT 2(1), 2013
 * (shear the triangle into ray space, three scaled barycentrics with a double-precision retry when one is exactly 0,
 * two-sided, hit when 0 <= t < max_t), and the answer of a query is DEFINED as the closest such hit over all triangles
 * of the mesh, exact ties on t going to the smallest face id.
 *
 * This file is self-contained on purpose: it shares no source text with isaaclab_amd/csrc (round-1 review: an oracle
 * that compiles the product's header compares the product with itself).  Compile with -ffp-contract=off so every
 * operation rounds once, which is what makes hit distances comparable bit for bit.
 *
 *   woop_hit()      one ray against one triangle
 *   rcref_brute()   every ray against every triangle
 *   rcref_bins_*()  the same answer through xy bins: a ray is tested against every triangle whose xy bounding box
 *                   shares a bin with the xy bounding box of the ray's clipped segment (no traversal order, no early
 *                   exit -- a superset of the triangles the ray can touch, reduced with the same closest/tie rule)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    int kx, ky, kz;
    float sx, sy, sz;
    float org[3];
} woop_ray;

static woop_ray woop_prepare(const float* org, const float* dir) {
    woop_ray w;
    float ax = fabsf(dir[0]), ay = fabsf(dir[1]), az = fabsf(dir[2]);
    /* dimension where the direction is largest (the first one on ties) */
    w.kz = 0;
    if (ay > ax) w.kz = 1;
    if (az > (w.kz == 0 ? ax : ay)) w.kz = 2;
    w.kx = (w.kz + 1) % 3;
    w.ky = (w.kx + 1) % 3;
    if (dir[w.kz] < 0.0f) { int s = w.kx; w.kx = w.ky; w.ky = s; }   /* keep the winding */
    w.sx = dir[w.kx] / dir[w.kz];
    w.sy = dir[w.ky] / dir[w.kz];
    w.sz = 1.0f / dir[w.kz];
    w.org[0] = org[0]; w.org[1] = org[1]; w.org[2] = org[2];
    return w;
}

/* 1 and *t on a hit (any t sign-consistent with the determinant; the caller applies 0 <= t < best) */
static int woop_hit(const woop_ray* w, const float* tri, float* t) {
    float p[3][3];
    for (int v = 0; v < 3; ++v)
        for (int a = 0; a < 3; ++a) p[v][a] = tri[3 * v + a] - w->org[a];
    float x[3], y[3];
    for (int v = 0; v < 3; ++v) {
        x[v] = p[v][w->kx] - w->sx * p[v][w->kz];
        y[v] = p[v][w->ky] - w->sy * p[v][w->kz];
    }
    float u = x[2] * y[1] - y[2] * x[1];
    float v = x[0] * y[2] - y[0] * x[2];
    float ww = x[1] * y[0] - y[1] * x[0];
    if (u == 0.0f || v == 0.0f || ww == 0.0f) {
        u = (float)((double)x[2] * (double)y[1] - (double)y[2] * (double)x[1]);
        v = (float)((double)x[0] * (double)y[2] - (double)y[0] * (double)x[2]);
        ww = (float)((double)x[1] * (double)y[0] - (double)y[1] * (double)x[0]);
    }
    int any_neg = u < 0.0f || v < 0.0f || ww < 0.0f;
    int any_pos = u > 0.0f || v > 0.0f || ww > 0.0f;
    if (any_neg && any_pos) return 0;
    float det = (u + v) + ww;
    if (det == 0.0f) return 0;
    float z0 = w->sz * p[0][w->kz], z1 = w->sz * p[1][w->kz], z2 = w->sz * p[2][w->kz];
    float ts = (u * z0 + v * z1) + ww * z2;
    if (det < 0.0f ? ts > 0.0f : ts < 0.0f) return 0;
    *t = ts * (1.0f / det);
    return 1;
}

typedef struct { float t; int32_t face; } closest;

static void keep_closest(closest* c, float t, int32_t face) {
    if (!(t >= 0.0f)) return;
    if (t < c->t || (t == c->t && c->face >= 0 && face < c->face)) { c->t = t; c->face = face; }
}

static void face_normal(const float* tri, float* n) {
    float e1[3], e2[3];
    for (int a = 0; a < 3; ++a) { e1[a] = tri[3 + a] - tri[a]; e2[a] = tri[6 + a] - tri[a]; }
    float x = e1[1] * e2[2] - e1[2] * e2[1], y = e1[2] * e2[0] - e1[0] * e2[2], z = e1[0] * e2[1] - e1[1] * e2[0];
    float len = sqrtf((x * x + y * y) + z * z);
    if (len > 0.0f) { x /= len; y /= len; z /= len; }
    n[0] = x; n[1] = y; n[2] = z;
}

static void store(const float* soup, int64_t i, const float* org, const float* dir, closest c, float* hits, float* dist,
                  float* normal, int32_t* face) {
    if (c.face >= 0) {
        for (int a = 0; a < 3; ++a) hits[3 * i + a] = org[a] + c.t * dir[a];
        if (dist) dist[i] = c.t;
        if (normal) face_normal(soup + 9 * (int64_t)c.face, normal + 3 * i);
        if (face) face[i] = c.face;
    } else {
        for (int a = 0; a < 3; ++a) hits[3 * i + a] = INFINITY;
        if (dist) dist[i] = INFINITY;
        if (normal) for (int a = 0; a < 3; ++a) normal[3 * i + a] = INFINITY;
        if (face) face[i] = -1;
    }
}

/* soup: (T, 9) pre-gathered triangle vertices a, b, c */
void rcref_brute(const float* soup, int32_t num_tris, int64_t num_rays, const float* starts, const float* dirs,
                 float max_dist, float* hits, float* dist, float* normal, int32_t* face) {
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t i = 0; i < num_rays; ++i) {
        woop_ray w = woop_prepare(starts + 3 * i, dirs + 3 * i);
        closest c = {max_dist, -1};
        for (int32_t f = 0; f < num_tris; ++f) {
            float t;
            if (woop_hit(&w, soup + 9 * (int64_t)f, &t)) keep_closest(&c, t, f);
        }
        store(soup, i, starts + 3 * i, dirs + 3 * i, c, hits, dist, normal, face);
    }
}

/* ---- xy bins ------------------------------------------------------------------------------------------------- */
typedef struct {
    const float* soup;
    int32_t num_tris;
    float x0, y0, bin, lo[3], hi[3];
    int nx, ny;
    int32_t* first;   /* nx * ny + 1 */
    int32_t* ids;
} rcref_bins;

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

rcref_bins* rcref_bins_build(const float* soup, int32_t num_tris, float bin) {
    rcref_bins* b = (rcref_bins*)calloc(1, sizeof(rcref_bins));
    b->soup = soup; b->num_tris = num_tris; b->bin = bin;
    for (int a = 0; a < 3; ++a) { b->lo[a] = INFINITY; b->hi[a] = -INFINITY; }
    for (int64_t k = 0; k < (int64_t)num_tris * 3; ++k)
        for (int a = 0; a < 3; ++a) {
            float v = soup[3 * k + a];
            if (v < b->lo[a]) b->lo[a] = v;
            if (v > b->hi[a]) b->hi[a] = v;
        }
    b->x0 = b->lo[0] - bin; b->y0 = b->lo[1] - bin;          /* one spare bin all round */
    b->nx = (int)ceilf((b->hi[0] - b->x0) / bin) + 2;
    b->ny = (int)ceilf((b->hi[1] - b->y0) / bin) + 2;
    int64_t nb = (int64_t)b->nx * b->ny;
    b->first = (int32_t*)calloc((size_t)nb + 1, sizeof(int32_t));
    const float slack = 1e-3f * bin;                          /* boxes grow a little: membership only has to be a superset */
    for (int pass = 0; pass < 2; ++pass) {
        int32_t* cur = NULL;
        if (pass == 1) {
            int32_t run = 0;
            for (int64_t k = 0; k <= nb; ++k) { int32_t n = b->first[k]; b->first[k] = run; run += n; }
            b->ids = (int32_t*)malloc(sizeof(int32_t) * (size_t)(run > 0 ? run : 1));
            cur = (int32_t*)malloc(sizeof(int32_t) * (size_t)nb);
            memcpy(cur, b->first, sizeof(int32_t) * (size_t)nb);
        }
        for (int32_t f = 0; f < num_tris; ++f) {
            const float* t = soup + 9 * (int64_t)f;
            float xl = fminf(t[0], fminf(t[3], t[6])) - slack, xh = fmaxf(t[0], fmaxf(t[3], t[6])) + slack;
            float yl = fminf(t[1], fminf(t[4], t[7])) - slack, yh = fmaxf(t[1], fmaxf(t[4], t[7])) + slack;
            int i0 = clampi((int)floorf((xl - b->x0) / bin), 0, b->nx - 1), i1 = clampi((int)floorf((xh - b->x0) / bin), 0, b->nx - 1);
            int j0 = clampi((int)floorf((yl - b->y0) / bin), 0, b->ny - 1), j1 = clampi((int)floorf((yh - b->y0) / bin), 0, b->ny - 1);
            for (int j = j0; j <= j1; ++j)
                for (int i = i0; i <= i1; ++i) {
                    int64_t k = (int64_t)j * b->nx + i;
                    if (pass == 0) b->first[k]++;
                    else b->ids[cur[k]++] = f;
                }
        }
        free(cur);
    }
    return b;
}

void rcref_bins_free(rcref_bins* b) {
    if (!b) return;
    free(b->first);
    free(b->ids);
    free(b);
}

void rcref_bins_cast(const rcref_bins* b, int64_t num_rays, const float* starts, const float* dirs, float max_dist,
                     float* hits, float* dist, float* normal, int32_t* face, int num_threads) {
#ifdef _OPENMP
    if (num_threads > 0) omp_set_num_threads(num_threads);
#endif
#pragma omp parallel for schedule(dynamic, 128)
    for (int64_t r = 0; r < num_rays; ++r) {
        const float* o = starts + 3 * r;
        const float* d = dirs + 3 * r;
        closest c = {max_dist, -1};
        /* parameter range in which the ray can be inside the mesh's (padded) bounding box, in double */
        double t0 = 0.0, t1 = (double)max_dist;
        int outside = 0;
        for (int a = 0; a < 3; ++a) {
            double lo = (double)b->lo[a] - 1e-3 - 1e-6 * fabs((double)b->lo[a]);
            double hi = (double)b->hi[a] + 1e-3 + 1e-6 * fabs((double)b->hi[a]);
            if (d[a] == 0.0f) {
                if ((double)o[a] < lo || (double)o[a] > hi) outside = 1;
            } else {
                double ta = (lo - (double)o[a]) / (double)d[a], tb = (hi - (double)o[a]) / (double)d[a];
                if (ta > tb) { double s = ta; ta = tb; tb = s; }
                ta -= 1e-6 * fabs(ta) + 1e-9; tb += 1e-6 * fabs(tb) + 1e-9;
                if (ta > t0) t0 = ta;
                if (tb < t1) t1 = tb;
            }
        }
        if (!outside && t0 <= t1) {
            double xa = (double)o[0] + t0 * (double)d[0], xb = (double)o[0] + t1 * (double)d[0];
            double ya = (double)o[1] + t0 * (double)d[1], yb = (double)o[1] + t1 * (double)d[1];
            double xl = fmin(xa, xb) - 1e-3 * b->bin, xh = fmax(xa, xb) + 1e-3 * b->bin;
            double yl = fmin(ya, yb) - 1e-3 * b->bin, yh = fmax(ya, yb) + 1e-3 * b->bin;
            int i0 = clampi((int)floor((xl - b->x0) / b->bin), 0, b->nx - 1), i1 = clampi((int)floor((xh - b->x0) / b->bin), 0, b->nx - 1);
            int j0 = clampi((int)floor((yl - b->y0) / b->bin), 0, b->ny - 1), j1 = clampi((int)floor((yh - b->y0) / b->bin), 0, b->ny - 1);
            woop_ray w = woop_prepare(o, d);
            for (int j = j0; j <= j1; ++j)
                for (int i = i0; i <= i1; ++i) {
                    int64_t k = (int64_t)j * b->nx + i;
                    for (int32_t q = b->first[k]; q < b->first[k + 1]; ++q) {
                        int32_t f = b->ids[q];
                        float t;
                        if (woop_hit(&w, b->soup + 9 * (int64_t)f, &t)) keep_closest(&c, t, f);
                    }
                }
        }
        store(b->soup, r, o, d, c, hits, dist, normal, face);
    }
}
