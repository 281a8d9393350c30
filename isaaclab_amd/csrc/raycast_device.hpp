// Device-side closest-hit ray / triangle-mesh query shared by raycast.hip and programs.hip.
//
// Reference boundary: warp's `mesh_query_ray` as called by LAB/utils/warp/kernels.py:66.  warp-lang
// (pinned 1.8.0 in environment.yml:285) is not part of /root/reference; its ray/triangle test is the
// published watertight algorithm of Woop, Benthin & Wald, "Watertight Ray/Triangle Intersection",
// JCGT 2(1), 2013, restated here (double-precision fallback on zero edge functions, two-sided,
// hits with t >= 0 and t < max_t).  Closest hit; exact ties on t are broken by the SMALLEST face id so the
// result is independent of traversal order (grid, BVH and the brute-force oracle agree bit for bit).
//
// The CPU oracle (oracle/raycast_ref.c) is a separate restatement of the same published algorithm and does NOT
// include this file; both are compiled with -ffp-contract=off, so hit distances can be compared bit for bit.
// (MDP_RC_HOST lets host tools compile this header; nothing under oracle/ or tests/ does.)
#pragma once
#include <stdint.h>
#include <math.h>

#ifdef MDP_RC_HOST
#define MDP_RC_FN static inline
typedef struct { int32_t x, y; } int2;
#else
#include <hip/hip_runtime.h>
#define MDP_RC_FN __device__ __forceinline__
#endif

typedef struct {
    // triangle soup (pre-gathered): tri_verts[9*f .. 9*f+8] = a, b, c
    const float* tri_verts;
    int32_t num_tris;
    // uniform xy cell grid: cell_start[2 * cell] = {first entry of the cell's triangle list (top bit: MDP_RC_CELL_DISJOINT),
    // number of entries}; cell_tris[...] the lists.  Cells with identical lists SHARE one list (and one run of cell
    // records): on large faces -- stair treads, box tops, the border -- thousands of neighbouring cells name the same
    // few records, so the rays of a scan that fall on one face read the same lines instead of private copies.
    const int32_t* cell_start;
    const int32_t* cell_tris;
    float gx0, gy0, cell, inv_cell;
    int32_t ncx, ncy;
    float zmin, zmax;  // mesh z range (slab test for DDA)
    // BVH: nodes[i] = {lo.x, lo.y, lo.z, left_or_first, hi.x, hi.y, hi.z, count}; count > 0 => leaf
    const float* bvh_nodes;
    const int32_t* bvh_tris;  // leaf triangle ids
    int32_t num_nodes;
    int32_t accel;  // 1 grid, 2 bvh
    // optional cell-ordered triangle records (device builder): one 48-byte record per cell reference, stored
    // in cell order, within a cell by descending zmax: {a.xyz, b.xyz, c.xyz, face id (int bits), zmax, zmin}.
    // A ray reads ONE contiguous run per visited cell instead of chasing cell_tris -> tri_verts, and a
    // downward ray stops at the first record that lies entirely below its current best hit.
    const float* cell_recs;
    // the same two arrays for EXACTLY vertical rays (mdp_rc_grid_down_exact): lists without the triangles that project
    // to an axis-parallel segment, which such a ray can never hit (raycast.hip: is_wall); equal to cell_start /
    // cell_recs when the mesh has none
    const int32_t* cell_start_down;
    const float* cell_recs_down;
    int32_t finite;  // every vertex coordinate is finite (device builder; lets vertical rays drop their zero products)
    // optional lattice patches (device builder; vertical rays only): runs of faces that form a regular height-field
    // lattice (two triangles per lattice square, LAB/terrains/height_field/utils.py:79-173) are also stored as their
    // vertex lattice -- (x, y, z, flags) per vertex, 16 B, shared by the six triangles around it -- and found through a
    // coarse xy table of 32-byte entries.  See mdp_rc_patch_down.
    const float* patch_verts;        // 4 floats per lattice vertex
    const int32_t* patch_cells;      // 8 words per coarse cell: {o_i, o_j, inv_h (float bits), voff, ncols, ni, nj, swap}
    float pgx0, pgy0, pg_inv;
    int32_t pgnx, pgny;
} mdp_mesh_view_t;

typedef struct {
    float t;
    int32_t face;
} mdp_hit_t;

// Top bit of a cell_start entry's first word (the low 31 bits are the list offset): the triangles of the
// cell have pairwise disjoint xy interiors; a triangle that projects to a segment (vertical wall) stays out of every
// open interior.  A VERTICAL ray strictly inside one
// footprint of such a cell cannot touch any other triangle of the cell, so its walk ends there whatever the record
// order; the records of such a cell are ordered by coverage of the cell, NOT by zmax, so the zmax early exit must not
// be used in it (a per-record skip is still valid).
#define MDP_RC_CELL_DISJOINT 0x80000000u
#define MDP_RC_CELL_OFFSET(x) ((int32_t)((uint32_t)(x) & 0x7fffffffu))

MDP_RC_FN int mdp_rc_max_dim(float ax, float ay, float az) {
    // index of the largest component (first wins on ties, like a chain of > comparisons)
    int k = 0;
    float m = ax;
    if (ay > m) { m = ay; k = 1; }
    if (az > m) { k = 2; }
    return k;
}

// Ray prepared once per ray (the "precompute" block of the published algorithm).
typedef struct {
    float ox, oy, oz;
    float dx, dy, dz;
    int kx, ky, kz;
    float Sx, Sy, Sz;
} mdp_ray_t;

MDP_RC_FN float mdp_rc_comp(float x, float y, float z, int k) { return k == 0 ? x : (k == 1 ? y : z); }

MDP_RC_FN mdp_ray_t mdp_rc_make_ray(float ox, float oy, float oz, float dx, float dy, float dz) {
    mdp_ray_t r;
    r.ox = ox; r.oy = oy; r.oz = oz; r.dx = dx; r.dy = dy; r.dz = dz;
    int kz = mdp_rc_max_dim(fabsf(dx), fabsf(dy), fabsf(dz));
    int kx = kz + 1; if (kx == 3) kx = 0;
    int ky = kx + 1; if (ky == 3) ky = 0;
    float dkz = mdp_rc_comp(dx, dy, dz, kz);
    if (dkz < 0.0f) { int tmp = kx; kx = ky; ky = tmp; }
    r.kx = kx; r.ky = ky; r.kz = kz;
    r.Sx = mdp_rc_comp(dx, dy, dz, kx) / dkz;
    r.Sy = mdp_rc_comp(dx, dy, dz, ky) / dkz;
    r.Sz = 1.0f / dkz;
    return r;
}

// returns 1 on hit and writes t (distance along the ray in units of |dir|), u, v barycentrics, det sign
MDP_RC_FN int mdp_rc_intersect(const mdp_ray_t* r, const float* tv, float* t_out, float* u_out, float* v_out) {
    float Ax3 = tv[0] - r->ox, Ay3 = tv[1] - r->oy, Az3 = tv[2] - r->oz;
    float Bx3 = tv[3] - r->ox, By3 = tv[4] - r->oy, Bz3 = tv[5] - r->oz;
    float Cx3 = tv[6] - r->ox, Cy3 = tv[7] - r->oy, Cz3 = tv[8] - r->oz;
    float Akz = mdp_rc_comp(Ax3, Ay3, Az3, r->kz), Bkz = mdp_rc_comp(Bx3, By3, Bz3, r->kz),
          Ckz = mdp_rc_comp(Cx3, Cy3, Cz3, r->kz);
    float Ax = mdp_rc_comp(Ax3, Ay3, Az3, r->kx) - r->Sx * Akz;
    float Ay = mdp_rc_comp(Ax3, Ay3, Az3, r->ky) - r->Sy * Akz;
    float Bx = mdp_rc_comp(Bx3, By3, Bz3, r->kx) - r->Sx * Bkz;
    float By = mdp_rc_comp(Bx3, By3, Bz3, r->ky) - r->Sy * Bkz;
    float Cx = mdp_rc_comp(Cx3, Cy3, Cz3, r->kx) - r->Sx * Ckz;
    float Cy = mdp_rc_comp(Cx3, Cy3, Cz3, r->ky) - r->Sy * Ckz;
    float U = Cx * By - Cy * Bx;
    float V = Ax * Cy - Ay * Cx;
    float W = Bx * Ay - By * Ax;
    if (U == 0.0f || V == 0.0f || W == 0.0f) {
        double CxBy = (double)Cx * (double)By, CyBx = (double)Cy * (double)Bx;
        U = (float)(CxBy - CyBx);
        double AxCy = (double)Ax * (double)Cy, AyCx = (double)Ay * (double)Cx;
        V = (float)(AxCy - AyCx);
        double BxAy = (double)Bx * (double)Ay, ByAx = (double)By * (double)Ax;
        W = (float)(BxAy - ByAx);
    }
    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return 0;
    float det = (U + V) + W;
    if (det == 0.0f) return 0;
    float Az = r->Sz * Akz, Bz = r->Sz * Bkz, Cz = r->Sz * Ckz;
    float T = (U * Az + V * Bz) + W * Cz;
    // sign(T) must equal sign(det) (t >= 0)
    if ((det < 0.0f && T > 0.0f) || (det > 0.0f && T < 0.0f)) return 0;
    float rcp = 1.0f / det;
    *u_out = U * rcp;
    *v_out = V * rcp;
    *t_out = T * rcp;
    return 1;
}

MDP_RC_FN void mdp_rc_consider(const mdp_ray_t* r, const float* tri_verts, int32_t f, mdp_hit_t* best) {
    float t, u, v;
    if (mdp_rc_intersect(r, tri_verts + 9 * (int64_t)f, &t, &u, &v)) {
        if (t >= 0.0f && (t < best->t || (t == best->t && f < best->face && best->face >= 0))) {
            best->t = t;
            best->face = f;
        }
    }
}

// Cell record: one 64-byte line {a.xyz, b.xyz, c.xyz, face id, zmax, zmin, zmax of the NEXT record of the cell (-inf
// after the last one), 3 spare}.  A record never straddles a 64-byte sector, and a downward walk through a zmax-sorted
// cell knows from the record it holds whether the next one can still matter: the common vertical ray -- first record
// hit, second one entirely below the hit -- costs exactly one sector.
#define MDP_RC_REC_STRIDE 16
typedef struct { float v[13]; } mdp_rec_t;

MDP_RC_FN mdp_rec_t mdp_rc_load_rec(const float* rec) {
    mdp_rec_t o;
#ifdef MDP_RC_HOST
    for (int i = 0; i < 13; ++i) o.v[i] = rec[i];
#else
    const float4* q = (const float4*)rec;  // records are 64-byte aligned: four 128-bit loads of one line
    float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
    o.v[0] = q0.x; o.v[1] = q0.y; o.v[2] = q0.z; o.v[3] = q0.w;
    o.v[4] = q1.x; o.v[5] = q1.y; o.v[6] = q1.z; o.v[7] = q1.w;
    o.v[8] = q2.x; o.v[9] = q2.y; o.v[10] = q2.z; o.v[11] = q2.w;
    o.v[12] = q3.x;
#endif
    return o;
}

MDP_RC_FN int32_t mdp_rc_rec_face(const mdp_rec_t* rec) {
    union { float f; int32_t i; } u;
    u.f = rec->v[9];
    return u.i;
}

// same as mdp_rc_consider for a cell record
MDP_RC_FN void mdp_rc_consider_rec(const mdp_ray_t* r, const mdp_rec_t* rec, mdp_hit_t* best) {
    int32_t f = mdp_rc_rec_face(rec);
    float t, u, v;
    if (mdp_rc_intersect(r, rec->v, &t, &u, &v)) {
        if (t >= 0.0f && (t < best->t || (t == best->t && f < best->face && best->face >= 0))) {
            best->t = t;
            best->face = f;
        }
    }
}

#ifndef MDP_RC_BATCH
#define MDP_RC_BATCH 4
#endif
// All records of one cell.  Loads are issued MDP_RC_BATCH records at a time (independent of the running best hit)
// so a lane has 12 x 128-bit loads in flight instead of one dependent round trip per triangle.
MDP_RC_FN void mdp_rc_cell_recs(const mdp_mesh_view_t* m, const mdp_ray_t* r, int32_t b, int32_t e, int zsorted,
                                mdp_hit_t* best) {
    // records are sorted by descending zmax (zsorted): on a downward ray a triangle lying entirely below the current
    // best hit needs a strictly larger t, and so does every record after it; without the order it is only skipped
    const int down = r->dz < 0.0f;
    const float inv_dz = down ? 1.0f / r->dz : 0.0f;
    int stop = 0;
    for (int32_t k = b; k < e && !stop; k += MDP_RC_BATCH) {
        mdp_rec_t R[MDP_RC_BATCH];
        for (int i = 0; i < MDP_RC_BATCH; ++i) {
            int32_t kk = k + i < e ? k + i : e - 1;
            R[i] = mdp_rc_load_rec(m->cell_recs + MDP_RC_REC_STRIDE * (int64_t)kk);
        }
        for (int i = 0; i < MDP_RC_BATCH; ++i) {
            if (k + i >= e || stop) continue;
            if (down && best->face >= 0) {
                float t_reach = (R[i].v[10] - r->oz) * inv_dz;  // parameter at which the ray reaches zmax
                if (t_reach > best->t + (fabsf(best->t) * 1e-5f + 1e-6f)) { stop = zsorted; continue; }
            }
            mdp_rc_consider_rec(r, &R[i], best);
        }
    }
}

// ---- brute force (oracle + tiny meshes) ------------------------------------------------------
MDP_RC_FN mdp_hit_t mdp_rc_brute(const mdp_mesh_view_t* m, const mdp_ray_t* r, float max_t) {
    mdp_hit_t best; best.t = max_t; best.face = -1;
    for (int32_t f = 0; f < m->num_tris; ++f) mdp_rc_consider(r, m->tri_verts, f, &best);
    return best;
}

// ---- uniform xy grid with conservative 2-D DDA -----------------------------------------------
// Every triangle is registered in all cells its (slightly padded) xy bounding box touches, so a ray
// only has to visit the cells its xy projection crosses.  Vertical rays (the height scan) touch one
// cell.  The walk stops as soon as the best hit lies before the entry point of the next cell.
MDP_RC_FN mdp_hit_t mdp_rc_grid(const mdp_mesh_view_t* m, const mdp_ray_t* r, float max_t) {
    mdp_hit_t best; best.t = max_t; best.face = -1;
    float dx = r->dx, dy = r->dy;
    // clip the parametric range [t0,t1] against the grid's xy extent and the mesh z slab
    float t0 = 0.0f, t1 = max_t;
    float lo[3] = {m->gx0, m->gy0, m->zmin};
    float hi[3] = {m->gx0 + m->cell * (float)m->ncx, m->gy0 + m->cell * (float)m->ncy, m->zmax};
    float o[3] = {r->ox, r->oy, r->oz};
    float d[3] = {r->dx, r->dy, r->dz};
    float pad = 1e-3f * m->cell + 1e-6f;
    for (int a = 0; a < 3; ++a) {
        float l = lo[a] - pad, h = hi[a] + pad;
        if (d[a] == 0.0f) {
            if (o[a] < l || o[a] > h) return best;
        } else {
            float inv = 1.0f / d[a];
            float ta = (l - o[a]) * inv, tb = (h - o[a]) * inv;
            if (ta > tb) { float tmp = ta; ta = tb; tb = tmp; }
            // widen by a relative epsilon: the clip only prunes, exactness comes from the triangle test
            ta -= fabsf(ta) * 1e-5f + 1e-6f;
            tb += fabsf(tb) * 1e-5f + 1e-6f;
            if (ta > t0) t0 = ta;
            if (tb < t1) t1 = tb;
        }
    }
    if (t0 > t1) return best;
    // entry cell
    float px = r->ox + t0 * dx, py = r->oy + t0 * dy;
    int cx = (int)floorf((px - m->gx0) * m->inv_cell);
    int cy = (int)floorf((py - m->gy0) * m->inv_cell);
    if (cx < 0) cx = 0; if (cx >= m->ncx) cx = m->ncx - 1;
    if (cy < 0) cy = 0; if (cy >= m->ncy) cy = m->ncy - 1;
    int sx = dx > 0.0f ? 1 : (dx < 0.0f ? -1 : 0);
    int sy = dy > 0.0f ? 1 : (dy < 0.0f ? -1 : 0);
    const float BIG = 3.0e38f;
    // Boundary parameters are recomputed from the integer cell index every step (no accumulation
    // drift).  Triangles are registered with a padded footprint (pad = 1e-3 cell), so walking the
    // unpadded lattice never misses one as long as the current cell is tested before stepping.
    float tstep = BIG;
    if (sx != 0) tstep = fabsf(m->cell / dx);
    if (sy != 0) { float ty = fabsf(m->cell / dy); if (ty < tstep) tstep = ty; }
    for (;;) {
        int32_t c = cy * m->ncx + cx;
        const int2 cref = reinterpret_cast<const int2*>(m->cell_start)[c];   // one aligned 8-byte load
        int32_t braw = cref.x;
        int32_t b = MDP_RC_CELL_OFFSET(braw), e = b + cref.y;
        float tmx = sx != 0 ? ((m->gx0 + m->cell * (float)(cx + (sx > 0 ? 1 : 0))) - r->ox) / dx : BIG;
        float tmy = sy != 0 ? ((m->gy0 + m->cell * (float)(cy + (sy > 0 ? 1 : 0))) - r->oy) / dy : BIG;
        float t_exit = tmx < tmy ? tmx : tmy;
        if (m->cell_recs) {
            // a slanted ray flies over most of the cells it crosses: in a cell whose records are sorted by descending
            // zmax the first record bounds the cell from above, and a downward ray that leaves the cell still above
            // that bound cannot hit anything registered here (a triangle reaching further is registered in the next
            // cells too) -- one record read instead of all of them
            int skip = 0;
            if (braw >= 0 && r->dz < 0.0f && b < e) {
                float t_low = t_exit < t1 ? t_exit : t1;
                float z_low = r->oz + t_low * r->dz;
                float z_top = m->cell_recs[MDP_RC_REC_STRIDE * (int64_t)b + 10];
                skip = z_low > z_top + (fabsf(z_top) + fabsf(z_low)) * 1e-5f + 1e-5f;
            }
            if (!skip) mdp_rc_cell_recs(m, r, b, e, braw >= 0, &best);
        } else {
            for (int32_t k = b; k < e; ++k) mdp_rc_consider(r, m->tri_verts, m->cell_tris[k], &best);
        }
        // a hit at t is final once every cell still ahead only holds points with parameter > t
        float slack = fabsf(t_exit) * 1e-4f + 1e-2f * tstep;
        if (best.face >= 0 && best.t < t_exit - slack) break;
        if (t_exit > t1) break;
        if (tmx < tmy) cx += sx; else cy += sy;
        if (cx < 0 || cx >= m->ncx || cy < 0 || cy >= m->ncy) break;
    }
    return best;
}

#ifndef MDP_RC_HOST
// ---- vertical downward rays (the height scan: pattern direction (0,0,-c), yaw-only attachment) -------------------
// Same expression trees as mdp_rc_make_ray / mdp_rc_intersect / mdp_rc_grid for d = (0, 0, dz < 0): there kz = 2 and,
// because d[kz] < 0, (kx, ky) = (1, 0); Sx = dy / dz and Sy = dx / dz are kept as computed values (-0.0f) so every
// intermediate is bit-identical to the generic path (which the CPU oracle runs).  With the permutation a compile-time
// constant the component selects vanish, and a vertical ray crosses exactly one xy cell: no DDA state.  The kernel
// built on this needs about half the registers of the generic one (twice the waves per SIMD on a latency-bound walk).
typedef struct { float ox, oy, oz, Sx, Sy, Sz, dz; } mdp_vray_t;

__device__ __forceinline__ mdp_vray_t mdp_rc_make_vray(float ox, float oy, float oz, float dx, float dy, float dz) {
    mdp_vray_t r;
    r.ox = ox; r.oy = oy; r.oz = oz; r.dz = dz;
    r.Sx = dy / dz;   // comp(d, kx = 1) / d[kz]
    r.Sy = dx / dz;   // comp(d, ky = 0) / d[kz]
    r.Sz = 1.0f / dz;
    return r;
}

__device__ __forceinline__ int mdp_rc_intersect_down(const mdp_vray_t* r, const float* tv, float* t_out) {
    float Ax3 = tv[0] - r->ox, Ay3 = tv[1] - r->oy, Az3 = tv[2] - r->oz;
    float Bx3 = tv[3] - r->ox, By3 = tv[4] - r->oy, Bz3 = tv[5] - r->oz;
    float Cx3 = tv[6] - r->ox, Cy3 = tv[7] - r->oy, Cz3 = tv[8] - r->oz;
    float Ax = Ay3 - r->Sx * Az3, Ay = Ax3 - r->Sy * Az3;
    float Bx = By3 - r->Sx * Bz3, By = Bx3 - r->Sy * Bz3;
    float Cx = Cy3 - r->Sx * Cz3, Cy = Cx3 - r->Sy * Cz3;
    float U = Cx * By - Cy * Bx;
    float V = Ax * Cy - Ay * Cx;
    float W = Bx * Ay - By * Ax;
    if (U == 0.0f || V == 0.0f || W == 0.0f) {
        double CxBy = (double)Cx * (double)By, CyBx = (double)Cy * (double)Bx;
        U = (float)(CxBy - CyBx);
        double AxCy = (double)Ax * (double)Cy, AyCx = (double)Ay * (double)Cx;
        V = (float)(AxCy - AyCx);
        double BxAy = (double)Bx * (double)Ay, ByAx = (double)By * (double)Ax;
        W = (float)(BxAy - ByAx);
    }
    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return 0;
    float det = (U + V) + W;
    if (det == 0.0f) return 0;
    float Az = r->Sz * Az3, Bz = r->Sz * Bz3, Cz = r->Sz * Cz3;
    float T = (U * Az + V * Bz) + W * Cz;
    if ((det < 0.0f && T > 0.0f) || (det > 0.0f && T < 0.0f)) return 0;
    float rcp = 1.0f / det;
    *t_out = T * rcp;
    return 1;
}

template <int BATCH>
__device__ __forceinline__ mdp_hit_t mdp_rc_grid_down(const mdp_mesh_view_t* m, const mdp_vray_t* r, float max_t) {
    mdp_hit_t best; best.t = max_t; best.face = -1;
    // clip of mdp_rc_grid with d.x == d.y == 0: the xy axes only test the origin, z gives [t0, t1]
    float pad = 1e-3f * m->cell + 1e-6f;
    float hix = m->gx0 + m->cell * (float)m->ncx, hiy = m->gy0 + m->cell * (float)m->ncy;
    if (r->ox < m->gx0 - pad || r->ox > hix + pad) return best;
    if (r->oy < m->gy0 - pad || r->oy > hiy + pad) return best;
    float t0 = 0.0f, t1 = max_t;
    {
        float l = m->zmin - pad, h = m->zmax + pad;
        float inv = 1.0f / r->dz;
        float ta = (l - r->oz) * inv, tb = (h - r->oz) * inv;
        if (ta > tb) { float tmp = ta; ta = tb; tb = tmp; }
        ta -= fabsf(ta) * 1e-5f + 1e-6f;
        tb += fabsf(tb) * 1e-5f + 1e-6f;
        if (ta > t0) t0 = ta;
        if (tb < t1) t1 = tb;
    }
    if (t0 > t1) return best;
    int cx = (int)floorf((r->ox - m->gx0) * m->inv_cell);
    int cy = (int)floorf((r->oy - m->gy0) * m->inv_cell);
    if (cx < 0) cx = 0; if (cx >= m->ncx) cx = m->ncx - 1;
    if (cy < 0) cy = 0; if (cy >= m->ncy) cy = m->ncy - 1;
    int32_t c = cy * m->ncx + cx;
    const int2 cref = reinterpret_cast<const int2*>(m->cell_start)[c];   // one aligned 8-byte load
    const int32_t braw = cref.x;
    const int disjoint = braw < 0;
    int32_t b = MDP_RC_CELL_OFFSET(braw), e = b + cref.y;
    const float inv_dz = 1.0f / r->dz;
    int stop = 0;
    int n = 1;   // the head record alone first, then BATCH at a time
    for (int32_t k = b; k < e && !stop; k += n, n = BATCH) {
        mdp_rec_t R[BATCH];
#pragma unroll
        for (int i = 0; i < BATCH; ++i) {
            if (i >= n) continue;
            int32_t kk = k + i < e ? k + i : e - 1;
            R[i] = mdp_rc_load_rec(m->cell_recs + MDP_RC_REC_STRIDE * (int64_t)kk);
        }
#pragma unroll
        for (int i = 0; i < BATCH; ++i) {
            if (i >= n || k + i >= e || stop) continue;
            if (best.face >= 0) {
                float t_reach = (R[i].v[10] - r->oz) * inv_dz;
                if (t_reach > best.t + (fabsf(best.t) * 1e-5f + 1e-6f)) { stop = !disjoint; continue; }
            }
            int32_t f = mdp_rc_rec_face(&R[i]);
            float t;
            if (mdp_rc_intersect_down(r, R[i].v, &t)) {
                if (t >= 0.0f && (t < best.t || (t == best.t && f < best.face && best.face >= 0))) {
                    best.t = t;
                    best.face = f;
                }
            }
            // the test the next record would fail first, made from this record's copy of its zmax: no load for it
            if (!disjoint && best.face >= 0) {
                float t_next = (R[i].v[12] - r->oz) * inv_dz;
                if (t_next > best.t + (fabsf(best.t) * 1e-5f + 1e-6f)) stop = 1;
            }
        }
    }
    return best;
}
// ---- lattice patches: the height-field shortcut of an EXACTLY vertical ray (d.x == d.y == 0, d.z < 0, finite mesh) ----
// A ray over a lattice patch finds its lattice square by two multiplications, reads the square's four vertices (64 B in
// two runs that the neighbouring rays of the scan share: ~1 vertex of new data per ray instead of a private line of cell
// records behind a cell_start lookup) and tests the square's two triangles with the arithmetic of
// mdp_rc_grid_down_exact, vertices in face order.  It only ACCEPTS when the ray lies strictly inside one of the two
// footprints (all three edge functions non-zero with one sign) of a square the builder marked safe: every grid cell the
// square touches has pairwise disjoint, non-degenerate footprints, so no other triangle of the mesh can contain the
// point, and by the watertight edge rule (an edge shared by two triangles gives them the SAME edge-function value with
// opposite signs) the neighbour rejects it too: the closest hit over all triangles is this one, same t bit for bit.
// Everything else -- unsafe square, a zero edge function, a point outside both footprints (vertices moved by the
// slope correction), a ray outside every patch -- returns 0 and the caller walks the grid cell as before.
// The three steps of mdp_rc_patch_down as separate routines (the wave-per-env cast issues the loads of several rays
// between them): coarse cell of the ray -> lattice square from the cell's entry -> test of the square's two triangles.
__device__ __forceinline__ bool mdp_rc_patch_cell(const mdp_mesh_view_t* m, float ox, float oy, int* pcell) {
    const int pcx = (int)floorf((ox - m->pgx0) * m->pg_inv);
    const int pcy = (int)floorf((oy - m->pgy0) * m->pg_inv);
    if (pcx < 0 || pcx >= m->pgnx || pcy < 0 || pcy >= m->pgny) return false;
    *pcell = pcy * m->pgnx + pcx;
    return true;
}

// c0, c1 = the coarse cell's two words-of-four; false: no patch here / the ray is outside the patch's lattice
__device__ __forceinline__ bool mdp_rc_patch_square(int4 c0, int4 c1, float ox, float oy, int* vidx, int* ncols_out) {
    const int voff = c0.w;
    if (voff < 0) return false;
    const int ncols = c1.x, ni = c1.y, nj = c1.z, swap = c1.w;
    const float o_i = __int_as_float(c0.x), o_j = __int_as_float(c0.y), inv_h = __int_as_float(c0.z);
    const int i = (int)floorf(((swap ? oy : ox) - o_i) * inv_h);
    const int j = (int)floorf(((swap ? ox : oy) - o_j) * inv_h);
    if (i < 0 || i >= ni || j < 0 || j >= nj) return false;
    *vidx = voff + i * ncols + j;
    *ncols_out = ncols;
    return true;
}

__device__ __forceinline__ int mdp_rc_patch_eval(float4 v00, float4 v01, float4 v10, float4 v11, float ox, float oy,
                                                 float oz, float dz, float max_t, mdp_hit_t* out) {
    if (!(__float_as_int(v00.w) & 1)) return 0;
    const float Sz = 1.0f / dz;
    // face 2k = (v00, v11, v01), face 2k + 1 = (v00, v10, v11); edge functions as in mdp_rc_grid_down_exact
    const float Ax = v00.y - oy, Ay = v00.x - ox;
    const float Dx = v11.y - oy, Dy = v11.x - ox;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float4 b = k == 0 ? v11 : v10, c = k == 0 ? v01 : v11;
        const float Bx = k == 0 ? Dx : v10.y - oy, By = k == 0 ? Dy : v10.x - ox;
        const float Cx = k == 0 ? v01.y - oy : Dx, Cy = k == 0 ? v01.x - ox : Dy;
        const float U = Cx * By - Cy * Bx;
        const float V_ = Ax * Cy - Ay * Cx;
        const float W = Bx * Ay - By * Ax;
        const bool pos = U > 0.0f && V_ > 0.0f && W > 0.0f, neg = U < 0.0f && V_ < 0.0f && W < 0.0f;
        if (!(pos || neg)) continue;
        out->t = max_t; out->face = -1;
        const float det = (U + V_) + W;
        const float Az = Sz * (v00.z - oz), Bz = Sz * (b.z - oz), Cz = Sz * (c.z - oz);
        const float T = (U * Az + V_ * Bz) + W * Cz;
        if ((det < 0.0f && T > 0.0f) || (det > 0.0f && T < 0.0f)) return 1;   // behind the origin: the only candidate
        const float rcp = 1.0f / det;
        const float t = T * rcp;
        if (t >= 0.0f && t < max_t) { out->t = t; out->face = 0; }
        return 1;
    }
    return 0;
}

__device__ __forceinline__ int mdp_rc_patch_down(const mdp_mesh_view_t* m, float ox, float oy, float oz, float dz,
                                                 float max_t, mdp_hit_t* out) {
    int pcell;
    if (!mdp_rc_patch_cell(m, ox, oy, &pcell)) return 0;
    const int4* pc = reinterpret_cast<const int4*>(m->patch_cells) + 2 * pcell;
    const int4 c0 = pc[0], c1 = pc[1];
    int vidx, ncols;
    if (!mdp_rc_patch_square(c0, c1, ox, oy, &vidx, &ncols)) return 0;
    const float4* V = reinterpret_cast<const float4*>(m->patch_verts) + vidx;
    return mdp_rc_patch_eval(V[0], V[1], V[ncols], V[ncols + 1], ox, oy, oz, dz, max_t, out);
}

// mdp_rc_grid_down for an EXACTLY vertical ray (d.x == d.y == 0, d.z < 0) on a mesh with finite vertices, with the
// arithmetic that cannot change a result removed (the cast kernel is bound by vector instructions, not by memory):
//   * Sx = d.y / d.z and Sy = d.x / d.z are zeros, so Sx * Az3 is a zero and Ax = Ay3 - Sx * Az3 == Ay3 bit for bit for
//     every non-zero Ay3 (a zero Ay3 may come out with the other sign, which no later product, sum with a non-zero
//     term or comparison against 0 can see);
//   * the z-slab clip only prunes: a triangle hit needs 0 <= t < max_t on its own.
// Same records, same order, same early exit, same t as mdp_rc_grid_down / mdp_rc_grid (the CPU oracle's path).
//
// One record of the walk: tv = the record's 16 words {a, b, c, face, zmax, zmin, zmax of the next record, ...}.  Updates
// the best hit and returns whether the walk ends here (`disjoint`: the cell's footprints are pairwise disjoint).
__device__ __forceinline__ bool mdp_rc_down_exact_record(const float* tv, float ox, float oy, float oz, float Sz,
                                                         bool disjoint, mdp_hit_t* best) {
    // (zmax-sorted cells) the test the NEXT record would fail first, made from this record's copy of that record's zmax
    // once this record has been tried: the next line is not loaded at all
    #define MDP_RC_NEXT_BELOW()                                                                        \
        if (!disjoint && best->face >= 0) {                                                            \
            float t_next = (tv[12] - oz) * Sz;                                                         \
            if (t_next > best->t + (fabsf(best->t) * 1e-5f + 1e-6f)) return true;                      \
        }
    if (best->face >= 0) {
        float t_reach = (tv[10] - oz) * Sz;
        if (t_reach > best->t + (fabsf(best->t) * 1e-5f + 1e-6f)) return !disjoint;
    }
    float Ax = tv[1] - oy, Ay = tv[0] - ox;
    float Bx = tv[4] - oy, By = tv[3] - ox;
    float Cx = tv[7] - oy, Cy = tv[6] - ox;
    float U = Cx * By - Cy * Bx;
    float V = Ax * Cy - Ay * Cx;
    float W = Bx * Ay - By * Ax;
    if (U == 0.0f || V == 0.0f || W == 0.0f) {
        double CxBy = (double)Cx * (double)By, CyBx = (double)Cy * (double)Bx;
        U = (float)(CxBy - CyBx);
        double AxCy = (double)Ax * (double)Cy, AyCx = (double)Ay * (double)Cx;
        V = (float)(AxCy - AyCx);
        double BxAy = (double)Bx * (double)Ay, ByAx = (double)By * (double)Ax;
        W = (float)(BxAy - ByAx);
    }
    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) { MDP_RC_NEXT_BELOW(); return false; }
    // strictly inside this footprint, in a cell of disjoint footprints: no other record can be touched
    bool stop = disjoint && ((U > 0.0f && V > 0.0f && W > 0.0f) || (U < 0.0f && V < 0.0f && W < 0.0f));
    float det = (U + V) + W;
    if (det == 0.0f) { MDP_RC_NEXT_BELOW(); return stop; }
    float Az = Sz * (tv[2] - oz), Bz = Sz * (tv[5] - oz), Cz = Sz * (tv[8] - oz);
    float T = (U * Az + V * Bz) + W * Cz;
    if ((det < 0.0f && T > 0.0f) || (det > 0.0f && T < 0.0f)) { MDP_RC_NEXT_BELOW(); return stop; }
    float rcp = 1.0f / det;
    float t = T * rcp;
    if (t >= 0.0f && t < best->t) { best->t = t; best->face = 0; }   // ties on t do not change t
    MDP_RC_NEXT_BELOW();
    return stop;
    #undef MDP_RC_NEXT_BELOW
}

// the ray's grid cell: false when the ray origin lies outside the (padded) grid, i.e. a miss
__device__ __forceinline__ bool mdp_rc_down_cell(const mdp_mesh_view_t* m, float ox, float oy, int32_t* cell) {
    float pad = 1e-3f * m->cell + 1e-6f;
    float hix = m->gx0 + m->cell * (float)m->ncx, hiy = m->gy0 + m->cell * (float)m->ncy;
    if (ox < m->gx0 - pad || ox > hix + pad) return false;
    if (oy < m->gy0 - pad || oy > hiy + pad) return false;
    int cx = (int)floorf((ox - m->gx0) * m->inv_cell);
    int cy = (int)floorf((oy - m->gy0) * m->inv_cell);
    if (cx < 0) cx = 0; if (cx >= m->ncx) cx = m->ncx - 1;
    if (cy < 0) cy = 0; if (cy >= m->ncy) cy = m->ncy - 1;
    *cell = cy * m->ncx + cx;
    return true;
}

template <int BATCH>
__device__ __forceinline__ mdp_hit_t mdp_rc_grid_down_exact(const mdp_mesh_view_t* m, float ox, float oy, float oz, float dz,
                                                            float max_t, int* dbg_iters = nullptr) {
    mdp_hit_t best; best.t = max_t; best.face = -1;
    int32_t c;
    if (!mdp_rc_down_cell(m, ox, oy, &c)) return best;
    const int2 cref = reinterpret_cast<const int2*>(m->cell_start_down)[c];   // one aligned 8-byte load
    const int32_t braw = cref.x;
    const bool disjoint = braw < 0;
    int32_t b = MDP_RC_CELL_OFFSET(braw), e = b + cref.y;
    const float Sz = 1.0f / dz;
    int stop = 0;
    // diagnostic (stamp build only): 1000 * disjoint + 10 * records in the cell + batches read
    if (dbg_iters) *dbg_iters = (disjoint ? 1000 : 0) + (e - b) * 10;
    int n = 1;   // the head record alone first (one 64-byte sector), then BATCH at a time
    for (int32_t k = b; k < e && !stop; k += n, n = BATCH) {
        if (dbg_iters) *dbg_iters += 1;
        mdp_rec_t R[BATCH];
#pragma unroll
        for (int i = 0; i < BATCH; ++i) {
            if (i >= n) continue;
            int32_t kk = k + i < e ? k + i : e - 1;
            R[i] = mdp_rc_load_rec(m->cell_recs_down + MDP_RC_REC_STRIDE * (int64_t)kk);
        }
#pragma unroll
        for (int i = 0; i < BATCH; ++i) {
            if (i >= n || k + i >= e || stop) continue;
            if (mdp_rc_down_exact_record(R[i].v, ox, oy, oz, Sz, disjoint, &best)) stop = 1;
        }
    }
    return best;
}
#endif  // !MDP_RC_HOST

// ---- BVH (binary, built on the host; stack traversal) -----------------------------------------
// one axis of the slab test.  A direction component that is exactly 0 constrains nothing as long as the
// origin lies inside the (closed) slab -- (hi - o) * inv would give 0 or NaN there and cut real hits.
MDP_RC_FN int mdp_rc_axis(float lo, float hi, float o, float d, float inv, float* tmin, float* tmax) {
    if (d == 0.0f) {
        *tmin = -3.0e38f;
        *tmax = 3.0e38f;
        return !(o < lo || o > hi);
    }
    float a = (lo - o) * inv, b = (hi - o) * inv;
    *tmin = fminf(a, b);
    *tmax = fmaxf(a, b);
    return 1;
}

MDP_RC_FN int mdp_rc_slab(const float* n, const mdp_ray_t* r, float inv_x, float inv_y, float inv_z, float tmax,
                          float* t_near) {
    // n = {lo.x, lo.y, lo.z, *, hi.x, hi.y, hi.z, *}; conservative (padded by relative eps)
    float tminx, tmaxx, tminy, tmaxy, tminz, tmaxz;
    int ok = mdp_rc_axis(n[0], n[4], r->ox, r->dx, inv_x, &tminx, &tmaxx);
    ok &= mdp_rc_axis(n[1], n[5], r->oy, r->dy, inv_y, &tminy, &tmaxy);
    ok &= mdp_rc_axis(n[2], n[6], r->oz, r->dz, inv_z, &tminz, &tmaxz);
    float tn = fmaxf(fmaxf(tminx, tminy), fmaxf(tminz, 0.0f));
    float tf = fminf(fminf(tmaxx, tmaxy), fminf(tmaxz, tmax));
    tn -= fabsf(tn) * 1e-5f + 1e-6f;
    tf += fabsf(tf) * 1e-5f + 1e-6f;
    *t_near = tn;
    return ok && tn <= tf;
}

MDP_RC_FN float mdp_rc_safe_inv(float d) {
    if (d == 0.0f) return 0.0f;  // unused for that axis (mdp_rc_axis handles d == 0)
    return 1.0f / d;
}

MDP_RC_FN mdp_hit_t mdp_rc_bvh(const mdp_mesh_view_t* m, const mdp_ray_t* r, float max_t) {
    mdp_hit_t best; best.t = max_t; best.face = -1;
    if (m->num_nodes <= 0) return best;
    float ix = mdp_rc_safe_inv(r->dx), iy = mdp_rc_safe_inv(r->dy), iz = mdp_rc_safe_inv(r->dz);
    int32_t stack[48];
    int sp = 0;
    stack[sp++] = 0;
    while (sp > 0) {
        int32_t ni = stack[--sp];
        const float* n = m->bvh_nodes + 8 * (int64_t)ni;
        float tn;
        // allow equal-t hits with a smaller face id: prune only strictly beyond best.t
        if (!mdp_rc_slab(n, r, ix, iy, iz, best.t, &tn)) continue;
        int32_t first = ((const int32_t*)n)[3];
        int32_t count = ((const int32_t*)n)[7];
        if (count > 0) {
            for (int32_t k = 0; k < count; ++k) mdp_rc_consider(r, m->tri_verts, m->bvh_tris[first + k], &best);
        } else {
            // children are first, first+1; push the farther one first
            const float* n0 = m->bvh_nodes + 8 * (int64_t)first;
            const float* n1 = n0 + 8;
            float t0n, t1n;
            int h0 = mdp_rc_slab(n0, r, ix, iy, iz, best.t, &t0n);
            int h1 = mdp_rc_slab(n1, r, ix, iy, iz, best.t, &t1n);
            if (h0 && h1) {
                if (t0n <= t1n) { stack[sp++] = first + 1; stack[sp++] = first; }
                else { stack[sp++] = first; stack[sp++] = first + 1; }
            } else if (h0) {
                stack[sp++] = first;
            } else if (h1) {
                stack[sp++] = first + 1;
            }
            if (sp > 46) sp = 46;  // depth guard (builder limits depth to 40)
        }
    }
    return best;
}

MDP_RC_FN mdp_hit_t mdp_rc_query(const mdp_mesh_view_t* m, const mdp_ray_t* r, float max_t, int accel) {
    if (accel == 2) return mdp_rc_bvh(m, r, max_t);
    if (accel == 1) return mdp_rc_grid(m, r, max_t);
    return mdp_rc_brute(m, r, max_t);
}

// face normal: normalize(cross(b - a, c - a))
MDP_RC_FN void mdp_rc_normal(const float* tv, float* nx, float* ny, float* nz) {
    float abx = tv[3] - tv[0], aby = tv[4] - tv[1], abz = tv[5] - tv[2];
    float acx = tv[6] - tv[0], acy = tv[7] - tv[1], acz = tv[8] - tv[2];
    float x = aby * acz - abz * acy, y = abz * acx - abx * acz, z = abx * acy - aby * acx;
    float l = sqrtf((x * x + y * y) + z * z);
    if (l > 0.0f) { x /= l; y /= l; z /= l; }
    *nx = x; *ny = y; *nz = z;
}
