// Mesh acceleration-structure builder (host) and ray-casting kernels.
// Reference: LAB/utils/warp/ops.py:24-145, LAB/utils/warp/kernels.py:13-75,
//            LAB/sensors/ray_caster/ray_caster.py:220-260, LAB/sensors/sensor_base.py:284-293
#include "mdp_device.hpp"
#include "mdp_host.hpp"
#include "raycast_device.hpp"
#include "mesh_handle.hpp"
#include "obs_post_device.hpp"
#include "raycaster_update_device.hpp"
#include "raycaster_update_host.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <limits>
#include <numeric>
#include <unordered_map>
#include <vector>

using namespace mdp;

// =============================================================================================
// Host builders
// =============================================================================================
namespace {

struct Tri2 {
    float ax, ay, bx, by, cx, cy;
};

// conservative 2-D triangle / axis-aligned rectangle overlap (separating axis test, padded)
bool tri_rect_overlap(const Tri2& t, float x0, float y0, float x1, float y1, float pad) {
    x0 -= pad; y0 -= pad; x1 += pad; y1 += pad;
    float tminx = std::min(t.ax, std::min(t.bx, t.cx)), tmaxx = std::max(t.ax, std::max(t.bx, t.cx));
    float tminy = std::min(t.ay, std::min(t.by, t.cy)), tmaxy = std::max(t.ay, std::max(t.by, t.cy));
    if (tmaxx < x0 || tminx > x1 || tmaxy < y0 || tminy > y1) return false;
    // triangle edge normals
    const float px[3] = {t.ax, t.bx, t.cx}, py[3] = {t.ay, t.by, t.cy};
    const float rx[4] = {x0, x1, x1, x0}, ry[4] = {y0, y0, y1, y1};
    for (int i = 0; i < 3; ++i) {
        int j = (i + 1) % 3, k = (i + 2) % 3;
        double nx = -(double)(py[j] - py[i]), ny = (double)(px[j] - px[i]);
        double len = std::sqrt(nx * nx + ny * ny);
        if (len == 0.0) continue;  // degenerate edge: bbox test above already covers it
        // orient the normal away from the opposite vertex
        double side = nx * ((double)px[k] - px[i]) + ny * ((double)py[k] - py[i]);
        if (side > 0) { nx = -nx; ny = -ny; }
        // rectangle entirely on the outside of this edge?
        double dmin = 1e300;
        for (int c = 0; c < 4; ++c) {
            double d = nx * ((double)rx[c] - px[i]) + ny * ((double)ry[c] - py[i]);
            dmin = std::min(dmin, d);
        }
        if (dmin > 1e-7 * len * (std::fabs((double)x0) + std::fabs((double)y0) + 1.0)) return false;
    }
    return true;
}

inline uint32_t expand_bits(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

struct BvhBuilder {
    const float* tv;  // 9 floats per tri
    std::vector<uint32_t> codes;
    std::vector<int32_t> order;  // sorted triangle ids
    std::vector<float> nodes;    // 8 floats per node
    int leaf_size = 4;

    void bounds(int lo, int hi, float* mn, float* mx) const {
        for (int a = 0; a < 3; ++a) { mn[a] = std::numeric_limits<float>::infinity(); mx[a] = -mn[a]; }
        for (int i = lo; i < hi; ++i) {
            const float* t = tv + 9 * (int64_t)order[i];
            for (int v = 0; v < 3; ++v)
                for (int a = 0; a < 3; ++a) {
                    mn[a] = std::min(mn[a], t[3 * v + a]);
                    mx[a] = std::max(mx[a], t[3 * v + a]);
                }
        }
    }
    int find_split(int lo, int hi) const {
        uint32_t first = codes[lo], last = codes[hi - 1];
        if (first == last) return (lo + hi) / 2;
        int common = __builtin_clz(first ^ last);
        int split = lo;
        int step = hi - 1 - lo;
        do {
            step = (step + 1) >> 1;
            int ns = split + step;
            if (ns < hi - 1) {
                if (__builtin_clz(first ^ codes[ns]) > common) split = ns;
            }
        } while (step > 1);
        return split + 1;
    }
    // iterative build with an explicit work list; children of a node are allocated adjacently
    void build(int n) {
        nodes.clear();
        nodes.resize(8);
        struct Item { int node, lo, hi, depth; };
        std::vector<Item> work;
        work.push_back({0, 0, n, 0});
        while (!work.empty()) {
            Item it = work.back();
            work.pop_back();
            float mn[3], mx[3];
            bounds(it.lo, it.hi, mn, mx);
            float* nd = nodes.data() + 8 * (int64_t)it.node;
            nd[0] = mn[0]; nd[1] = mn[1]; nd[2] = mn[2];
            nd[4] = mx[0]; nd[5] = mx[1]; nd[6] = mx[2];
            int count = it.hi - it.lo;
            if (count <= leaf_size || it.depth >= 56) {
                reinterpret_cast<int32_t*>(nd)[3] = it.lo;
                reinterpret_cast<int32_t*>(nd)[7] = count;
                continue;
            }
            int mid = find_split(it.lo, it.hi);
            int child = (int)(nodes.size() / 8);
            nodes.resize(nodes.size() + 16);
            nd = nodes.data() + 8 * (int64_t)it.node;  // resize may move
            reinterpret_cast<int32_t*>(nd)[3] = child;
            reinterpret_cast<int32_t*>(nd)[7] = 0;
            work.push_back({child, it.lo, mid, it.depth + 1});
            work.push_back({child + 1, mid, it.hi, it.depth + 1});
        }
    }
};

template <class T>
int upload(const std::vector<T>& h, T** d, int64_t* bytes) {
    size_t n = h.size() * sizeof(T);
    if (n == 0) n = sizeof(T);
    MDP_CHECK_HIP(hipMalloc((void**)d, n));
    if (!h.empty()) MDP_CHECK_HIP(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    *bytes += (int64_t)n;
    return MDP_OK;
}

}  // namespace

extern "C" int mdp_mesh_create(const float* points, int num_points, const int32_t* indices, int num_tris,
                               int device, int accel, mdp_mesh_t** out) {
    MDP_REQUIRE(points && indices && out, "null pointer");
    MDP_REQUIRE(num_points > 0 && num_tris > 0, "empty mesh (%d points, %d triangles)", num_points, num_tris);
    MDP_REQUIRE(accel >= 0 && accel <= 2, "accel must be 0 (auto), 1 (grid) or 2 (bvh)");
    for (int64_t i = 0; i < (int64_t)num_tris * 3; ++i)
        MDP_REQUIRE(indices[i] >= 0 && indices[i] < num_points, "triangle index %d out of range at %lld",
                    indices[i], (long long)i);
    MDP_CHECK_HIP(hipSetDevice(device));
    mdp_mesh* m = new mdp_mesh();
    m->device = device;
    m->num_tris = num_tris;
    // ---- triangle soup
    std::vector<float> tv((size_t)num_tris * 9);
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int f = 0; f < num_tris; ++f)
        for (int v = 0; v < 3; ++v) {
            const float* p = points + 3 * (int64_t)indices[3 * (int64_t)f + v];
            for (int a = 0; a < 3; ++a) {
                tv[9 * (size_t)f + 3 * v + a] = p[a];
                mn[a] = std::min(mn[a], p[a]);
                mx[a] = std::max(mx[a], p[a]);
            }
        }
    // ---- uniform xy grid
    // cell edge = median triangle footprint size sqrt(2 * area_xy): on a height field (the common case) that is the
    // lattice pitch, so with the half-cell offset below every cell overlaps ~6 triangles.  A size derived from
    // the mesh extent would be skewed by empty borders and by a few huge triangles (stairs, slabs).
    double ex = (double)mx[0] - mn[0], ey = (double)mx[1] - mn[1];
    double cell;
    {
        std::vector<float> sizes;
        sizes.reserve((size_t)num_tris);
        for (int f = 0; f < num_tris; ++f) {
            const float* t = tv.data() + 9 * (size_t)f;
            double a2 = std::fabs(((double)t[3] - t[0]) * ((double)t[7] - t[1]) - ((double)t[6] - t[0]) * ((double)t[4] - t[1]));
            if (a2 > 0.0) sizes.push_back((float)std::sqrt(a2));   // a2 = 2 * area_xy
        }
        if (sizes.empty()) {
            cell = std::sqrt(std::max(ex, 1e-6) * std::max(ey, 1e-6));
        } else {
            std::nth_element(sizes.begin(), sizes.begin() + sizes.size() / 2, sizes.end());
            cell = sizes[sizes.size() / 2];
        }
        // half the lattice pitch: ~5 records per cell instead of ~6.4 and, more importantly, most vertical rays
        // finish within their first two records (measured on the 1.16 M-triangle terrain: 41 -> 33 us per
        // 766 k-ray height scan).  MDP_GRID_CELL_SCALE overrides the factor (tools/bench_stages.py).
        double f = 0.5;
        if (const char* sc = std::getenv("MDP_GRID_CELL_SCALE")) {
            double g = std::atof(sc);
            if (g > 0.0) f = g;
        }
        cell *= f;
        cell = std::max(cell, 1e-4);
    }
    // memory guards: at most 8 M cells and (bounding-box estimate) 48 M cell references (64 B each)
    const double max_cells = 8.0e6, max_refs = 48.0e6;
    for (int guard = 0; guard < 64; ++guard) {
        double cells = (ex / cell + 2) * (ey / cell + 2);
        double refs = 0.0;
        if (cells <= max_cells) {
            for (int f = 0; f < num_tris; ++f) {
                const float* t = tv.data() + 9 * (size_t)f;
                double wx = std::max(t[0], std::max(t[3], t[6])) - std::min(t[0], std::min(t[3], t[6]));
                double wy = std::max(t[1], std::max(t[4], t[7])) - std::min(t[1], std::min(t[4], t[7]));
                refs += 0.5 * (wx / cell + 1.0) * (wy / cell + 1.0) + 1.0;
            }
        }
        if (cells <= max_cells && refs <= max_refs) break;
        cell *= 1.25;
    }
    if (!(cell > 0)) cell = 1.0;
    // half-cell offset keeps lattice-aligned height-field edges away from cell borders
    float gx0 = (float)(mn[0] - 0.5 * cell), gy0 = (float)(mn[1] - 0.5 * cell);
    int ncx = (int)std::floor((mx[0] - gx0) / cell) + 1, ncy = (int)std::floor((mx[1] - gy0) / cell) + 1;
    ncx = std::max(ncx, 1);
    ncy = std::max(ncy, 1);
    float cellf = (float)cell;
    float pad = 1e-3f * cellf + 1e-6f;
    auto cell_range = [&](const float* t, int& x0, int& x1, int& y0, int& y1) {
        float tminx = std::min(t[0], std::min(t[3], t[6])) - pad, tmaxx = std::max(t[0], std::max(t[3], t[6])) + pad;
        float tminy = std::min(t[1], std::min(t[4], t[7])) - pad, tmaxy = std::max(t[1], std::max(t[4], t[7])) + pad;
        x0 = std::max(0, (int)std::floor((tminx - gx0) / cellf));
        x1 = std::min(ncx - 1, (int)std::floor((tmaxx - gx0) / cellf));
        y0 = std::max(0, (int)std::floor((tminy - gy0) / cellf));
        y1 = std::min(ncy - 1, (int)std::floor((tmaxy - gy0) / cellf));
    };
    auto overlaps = [&](const float* t, int cx, int cy) {
        Tri2 t2{t[0], t[1], t[3], t[4], t[6], t[7]};
        float x0 = gx0 + cellf * (float)cx, y0 = gy0 + cellf * (float)cy;
        return tri_rect_overlap(t2, x0, y0, x0 + cellf, y0 + cellf, pad);
    };
    // A triangle whose three vertices share their x (or their y) projects to an axis-parallel segment: for an EXACTLY
    // vertical ray its three edge functions are fl(k p) - fl(k q), fl(k q) - fl(k r), fl(k r) - fl(k p) with ONE common
    // factor k (the shared coordinate minus the ray's) -- they cannot all be >= 0 (or all <= 0) unless all three vanish,
    // and then det == 0: mdp_rc_down_exact_record rejects it whatever the ray (the double-precision retry has the same
    // cyclic form).  Such walls (stair risers, box sides, the vertical faces the height-field slope correction makes)
    // never produce a hit for the height scan, but they sit in every cell along an edge and lengthen its walk: the
    // vertical-ray routines get their own lists without them ("down" lists; slanted rays keep the full ones).  Measured
    // on the benchmark terrain: stand-alone 766 k-ray scan 36.3 -> 30.9 us, post-physics launch 39.2 -> 33.0 us.
    auto is_wall = [](const float* t) { return (t[0] == t[3] && t[3] == t[6]) || (t[1] == t[4] && t[4] == t[7]); };
    auto make_lists = [&](bool down, std::vector<int32_t>& ct, std::vector<int32_t>& cell_ref, std::vector<float>& recs,
                          std::vector<uint8_t>& disjoint, int64_t& total_cell_refs) {
    std::vector<int32_t> counts((size_t)ncx * ncy + 1, 0);
    for (int pass = 0; pass < 2; ++pass) {
        std::vector<int32_t> cursor;
        if (pass == 1) {
            // exclusive scan
            int32_t run = 0;
            for (size_t c = 0; c < counts.size(); ++c) { int32_t v = counts[c]; counts[c] = run; run += v; }
            cursor.assign(counts.begin(), counts.end() - 1);
            ct.resize((size_t)counts.back());
        }
        for (int f = 0; f < num_tris; ++f) {
            const float* t = tv.data() + 9 * (size_t)f;
            if (down && is_wall(t)) continue;
            int x0, x1, y0, y1;
            cell_range(t, x0, x1, y0, y1);
            for (int cy = y0; cy <= y1; ++cy)
                for (int cx = x0; cx <= x1; ++cx) {
                    if (!overlaps(t, cx, cy)) continue;
                    size_t c = (size_t)cy * ncx + cx;
                    if (pass == 0) counts[c]++;
                    else ct[(size_t)cursor[c]++] = f;  // ascending face id within a cell
                }
        }
    }
    {
        const size_t ncell = (size_t)ncx * ncy;
        std::vector<float> zmax_of((size_t)num_tris), zmin_of((size_t)num_tris);
        for (int f = 0; f < num_tris; ++f) {
            const float* t = tv.data() + 9 * (size_t)f;
            zmax_of[f] = std::max(t[2], std::max(t[5], t[8]));
            zmin_of[f] = std::min(t[2], std::min(t[5], t[8]));
        }
        for (size_t c = 0; c < ncell; ++c) {
            auto b = ct.begin() + counts[c], e = ct.begin() + counts[c + 1];
            if (e - b > 1)
                std::sort(b, e, [&](int32_t x, int32_t y) {
                    return zmax_of[x] > zmax_of[y] || (zmax_of[x] == zmax_of[y] && x < y);
                });
        }
        // ---- cells whose triangles have pairwise disjoint xy interiors (every cell of a height field away from its
        // vertical walls): a vertical ray strictly inside one triangle's footprint cannot touch another one, so the
        // walk may stop there (raycast_device.hpp: MDP_RC_CELL_DISJOINT).  Such a cell is ordered by how much of the
        // cell each triangle covers (most likely hit first); the zmax order is only needed by the zmax early exit,
        // which these cells do not use.
        disjoint.assign(ncell, 0);
        auto orient = [&](const float* t, int i, int j, double px, double py) {
            double ax = t[3 * i], ay = t[3 * i + 1], bx = t[3 * j], by = t[3 * j + 1];
            return (bx - ax) * (py - ay) - (by - ay) * (px - ax);   // exact sign for float inputs
        };
        auto separated_by_edge_of = [&](const float* A, const float* B) {
            double area = orient(A, 0, 1, A[6], A[7]);
            if (area == 0.0) return false;
            for (int i = 0; i < 3; ++i) {
                int j = (i + 1) % 3;
                bool all_out = true;
                for (int k = 0; k < 3 && all_out; ++k) {
                    double o = orient(A, i, j, B[3 * k], B[3 * k + 1]);
                    all_out = area > 0.0 ? o <= 0.0 : o >= 0.0;
                }
                if (all_out) return true;
            }
            return false;
        };
        // vertical walls may sit in a "disjoint" cell when their segment stays out of every open footprint of the cell
        // (stair risers on the shared edge of two treads): -1.4 us per 766 k-ray scan.  MDP_DISJOINT_WALLS=0: A/B runs
        const bool walls_ok = [] { const char* v = std::getenv("MDP_DISJOINT_WALLS"); return !v || std::atoi(v) != 0; }();
        std::vector<std::pair<int, int32_t>> scored;
        for (size_t c = 0; c < ncell; ++c) {
            int32_t* b = ct.data() + counts[c];
            const int n = counts[c + 1] - counts[c];
            if (n < 1 || n > 16) continue;
            bool ok = true;
            for (int i = 0; i < n && ok; ++i) {
                const float* A = tv.data() + 9 * (size_t)b[i];
                const bool seg_a = orient(A, 0, 1, A[6], A[7]) == 0.0;   // projects to a segment (vertical wall)
                if (seg_a && !walls_ok) ok = false;
                for (int j = i + 1; j < n && ok; ++j) {
                    const float* B = tv.data() + 9 * (size_t)b[j];
                    const bool seg_b = orient(B, 0, 1, B[6], B[7]) == 0.0;
                    // a wall's segment must stay out of the open interior of every footprint of the cell (all its
                    // vertices on the closed outer side of one edge): a point strictly inside a footprint is then not on
                    // the wall, and the wall itself never satisfies "strictly inside" (its edge functions sum to zero)
                    if (seg_a && seg_b) ok = true;
                    else if (seg_b) ok = separated_by_edge_of(A, B);
                    else if (seg_a) ok = separated_by_edge_of(B, A);
                    else ok = separated_by_edge_of(A, B) || separated_by_edge_of(B, A);
                }
            }
            if (!ok) continue;
            disjoint[c] = 1;
            if (n < 2) continue;
            // coverage score: how many of 4 x 4 sample points of the cell fall inside the triangle
            const int cx = (int)(c % (size_t)ncx), cy = (int)(c / (size_t)ncx);
            scored.clear();
            for (int i = 0; i < n; ++i) {
                const float* A = tv.data() + 9 * (size_t)b[i];
                double area = orient(A, 0, 1, A[6], A[7]);
                int score = 0;
                for (int sy = 0; sy < 4; ++sy)
                    for (int sx = 0; sx < 4; ++sx) {
                        double px = (double)gx0 + (double)cellf * (cx + (sx + 0.5) * 0.25);
                        double py = (double)gy0 + (double)cellf * (cy + (sy + 0.5) * 0.25);
                        double o0 = orient(A, 0, 1, px, py), o1 = orient(A, 1, 2, px, py), o2 = orient(A, 2, 0, px, py);
                        bool in = area > 0.0 ? (o0 >= 0 && o1 >= 0 && o2 >= 0) : (o0 <= 0 && o1 <= 0 && o2 <= 0);
                        score += in ? 1 : 0;
                    }
                scored.emplace_back(-score, b[i]);
            }
            // stable: equal coverage keeps the zmax / face-id order
            std::stable_sort(scored.begin(), scored.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
            for (int i = 0; i < n; ++i) b[i] = scored[i].second;
        }
        // ---- interning: cells whose (ordered) lists are identical share one list and one run of records
        {
            std::vector<int32_t> uniq;                       // concatenated unique lists
            uniq.reserve(ct.size() / 4 + 16);
            cell_ref.assign(2 * ncell, 0);
            std::unordered_map<uint64_t, std::vector<std::pair<int32_t, int32_t>>> seen;   // hash -> (offset, count) candidates
            seen.reserve(ncell / 4 + 16);
            for (size_t c = 0; c < ncell; ++c) {
                const int32_t* b = ct.data() + counts[c];
                const int32_t n = counts[c + 1] - counts[c];
                int32_t off = 0;
                if (n > 0) {
                    uint64_t h = 1469598103934665603ull ^ (uint64_t)n;
                    for (int32_t k = 0; k < n; ++k) { h ^= (uint64_t)(uint32_t)b[k]; h *= 1099511628211ull; h ^= h >> 29; }
                    auto& cands = seen[h];
                    bool found = false;
                    for (const auto& cd : cands)
                        if (cd.second == n && std::memcmp(uniq.data() + cd.first, b, sizeof(int32_t) * (size_t)n) == 0) { off = cd.first; found = true; break; }
                    if (!found) {
                        off = (int32_t)uniq.size();
                        uniq.insert(uniq.end(), b, b + n);
                        cands.emplace_back(off, n);
                    }
                }
                cell_ref[2 * c] = off;
                cell_ref[2 * c + 1] = n;
            }
            // the per-list "next zmax" field needs the list boundaries: walk the unique lists through their first owner
            recs.assign(uniq.size() * MDP_RC_REC_STRIDE, 0.0f);
            std::vector<uint8_t> done(uniq.size(), 0);
            for (size_t c = 0; c < ncell; ++c) {
                const int32_t off = cell_ref[2 * c], n = cell_ref[2 * c + 1];
                if (n == 0 || done[(size_t)off]) continue;
                done[(size_t)off] = 1;
                for (int32_t k = 0; k < n; ++k) {
                    const int32_t f = uniq[(size_t)off + k];
                    float* r = recs.data() + (size_t)MDP_RC_REC_STRIDE * ((size_t)off + k);
                    std::memcpy(r, tv.data() + 9 * (size_t)f, 9 * sizeof(float));
                    std::memcpy(r + 9, &f, sizeof(int32_t));
                    r[10] = zmax_of[f];
                    r[11] = zmin_of[f];
                    // zmax of the record behind this one in the same list (raycast_device.hpp: mdp_rec_t)
                    r[12] = k + 1 < n ? zmax_of[uniq[(size_t)off + k + 1]] : -INFINITY;
                }
            }
            total_cell_refs = (int64_t)ct.size();   // before sharing: what the grid / BVH choice looks at
            ct.swap(uniq);
        }
    }
    };
    // ---- cell records: per cell sorted by descending zmax (ties: ascending face id), 16 floats each
    std::vector<float> recs, recs_down;
    std::vector<int32_t> cell_ref, cell_ref_down;   // {list offset, count} per cell
    std::vector<int32_t> ct_down;
    int64_t total_cell_refs = 0, total_down_refs = 0;
    std::vector<uint8_t> disjoint, disjoint_down;
    make_lists(false, m->h_cell_tris, cell_ref, recs, disjoint, total_cell_refs);
    bool has_walls = false;
    for (int f = 0; f < num_tris && !has_walls; ++f) has_walls = is_wall(tv.data() + 9 * (size_t)f);
    const bool down_lists = has_walls && !std::getenv("MDP_NO_DOWN_LISTS");
    if (down_lists) make_lists(true, ct_down, cell_ref_down, recs_down, disjoint_down, total_down_refs);
    const std::vector<uint8_t>& disjoint_v = down_lists ? disjoint_down : disjoint;   // what a vertical ray walks
    // ---- lattice patches (raycast_device.hpp: mdp_rc_patch_down).  A patch is a run of faces with the index pattern
    // of a height-field mesh (LAB/terrains/height_field/utils.py:158-171: per lattice square (a, a+C+1, a+1) and
    // (a, a+C, a+C+1), squares row by row, rows C vertices apart) whose vertices sit on an axis-aligned lattice.
    std::vector<float> pverts;            // 4 floats per lattice vertex: x, y, z, flag bits
    std::vector<int32_t> pcells;          // 8 words per coarse cell
    float pgx0 = 0.0f, pgy0 = 0.0f, pg_inv = 0.0f;
    int pgnx = 0, pgny = 0;
    bool all_finite = true;
    for (size_t i = 0; i < tv.size() && all_finite; ++i) all_finite = std::isfinite(tv[i]);
    if (all_finite && !std::getenv("MDP_NO_LATTICE_PATCHES")) {
        struct Patch { int a0, C, ni, nj, swap; float o_i, o_j, h; size_t voff; };
        std::vector<Patch> patches;
        auto idx = [&](int f, int k) { return indices[3 * (int64_t)f + k]; };
        auto square = [&](int f, int a, int C) {
            return f + 1 < num_tris && idx(f, 0) == a && idx(f, 1) == a + C + 1 && idx(f, 2) == a + 1 && idx(f + 1, 0) == a &&
                   idx(f + 1, 1) == a + C && idx(f + 1, 2) == a + C + 1;
        };
        double h_min = 1e300;
        for (int f = 0; f + 1 < num_tris;) {
            const int a = idx(f, 0), C = idx(f + 1, 1) - a;
            if (C < 3 || !square(f, a, C)) { ++f; continue; }
            int rows = 0, ff = f;
            for (;;) {
                bool ok = true;
                for (int j = 0; j < C - 1 && ok; ++j) ok = square(ff + 2 * j, a + rows * C + j, C);
                if (!ok) break;
                ++rows;
                ff += 2 * (C - 1);
            }
            if (rows < 2) { ++f; continue; }
            const int f0 = f;
            f = ff;
            auto P = [&](int i, int j) { return points + 3 * (int64_t)(a + i * C + j); };
            const float* p00 = P(0, 0);
            const double dix = (double)P(rows, 0)[0] - p00[0], diy = (double)P(rows, 0)[1] - p00[1];
            const double djx = (double)P(0, C - 1)[0] - p00[0], djy = (double)P(0, C - 1)[1] - p00[1];
            Patch pt{a, C, rows, C - 1, 0, 0.0f, 0.0f, 0.0f, 0};
            double hi, hj, off_i, off_j;
            if (std::fabs(dix) >= std::fabs(diy)) { pt.swap = 0; hi = dix / rows; hj = djy / (C - 1); off_i = diy; off_j = djx; pt.o_i = p00[0]; pt.o_j = p00[1]; }
            else { pt.swap = 1; hi = diy / rows; hj = djx / (C - 1); off_i = dix; off_j = djy; pt.o_i = p00[1]; pt.o_j = p00[0]; }
            if (!(hi > 0.0 && hj > 0.0) || std::fabs(hi - hj) > 1e-3 * hi || std::fabs(off_i) > 1e-3 * hi * rows ||
                std::fabs(off_j) > 1e-3 * hj * (C - 1))
                continue;                       // not an axis-aligned square lattice with increasing coordinates
            pt.h = (float)hi;
            pt.voff = pverts.size() / 4;
            if (pt.voff + (size_t)(rows + 1) * C > (size_t)0x7fffffff) break;
            h_min = std::min(h_min, hi);
            // vertices + per-square flag: regular (the square's vertices sit on their lattice points, so its two
            // triangles tile the nominal square), non-degenerate, and every grid cell it touches is a disjoint one
            for (int i = 0; i <= rows; ++i)
                for (int j = 0; j < C; ++j) {
                    const float* v = P(i, j);
                    int32_t flag = 0;
                    if (i < rows && j < C - 1) {
                        bool ok = true;
                        float bx0 = INFINITY, by0 = INFINITY, bx1 = -INFINITY, by1 = -INFINITY;
                        for (int di = 0; di < 2 && ok; ++di)
                            for (int dj = 0; dj < 2 && ok; ++dj) {
                                const float* q = P(i + di, j + dj);
                                const double ci = pt.swap ? q[1] : q[0], cj = pt.swap ? q[0] : q[1];
                                ok = std::fabs(ci - (pt.o_i + (double)(i + di) * hi)) <= 0.01 * hi &&
                                     std::fabs(cj - (pt.o_j + (double)(j + dj) * hi)) <= 0.01 * hi;
                                bx0 = std::min(bx0, q[0]); bx1 = std::max(bx1, q[0]);
                                by0 = std::min(by0, q[1]); by1 = std::max(by1, q[1]);
                            }
                        if (ok) {
                            const int cx0 = std::max(0, (int)std::floor((bx0 - pad - gx0) / cellf)), cx1 = std::min(ncx - 1, (int)std::floor((bx1 + pad - gx0) / cellf));
                            const int cy0 = std::max(0, (int)std::floor((by0 - pad - gy0) / cellf)), cy1 = std::min(ncy - 1, (int)std::floor((by1 + pad - gy0) / cellf));
                            for (int cy = cy0; cy <= cy1 && ok; ++cy)
                                for (int cx = cx0; cx <= cx1 && ok; ++cx) ok = disjoint_v[(size_t)cy * ncx + cx] != 0;
                        }
                        flag = ok ? 1 : 0;
                    }
                    float w;
                    std::memcpy(&w, &flag, sizeof(w));
                    pverts.insert(pverts.end(), {v[0], v[1], v[2], w});
                }
            patches.push_back(pt);
            (void)f0;
        }
        if (!patches.empty()) {
            // coarse lookup table: a coarse cell names the patch that covers most of it (rays in the rest of the cell
            // fail the patch's bounds test and take the ordinary path)
            double G = 8.0 * h_min;
            while ((ex / G + 2) * (ey / G + 2) > 1.0e6) G *= 2.0;
            pgx0 = mn[0]; pgy0 = mn[1];
            pgnx = (int)std::floor(ex / G) + 1; pgny = (int)std::floor(ey / G) + 1;
            pg_inv = (float)(1.0 / G);
            pcells.assign((size_t)pgnx * pgny * 8, 0);
            for (size_t c = 0; c < (size_t)pgnx * pgny; ++c) pcells[8 * c + 3] = -1;
            std::vector<float> best_area((size_t)pgnx * pgny, 0.0f);
            for (const Patch& pt : patches) {
                const double li = (double)pt.h * pt.ni, lj = (double)pt.h * pt.nj;
                const double x0 = pt.swap ? pt.o_j : pt.o_i, y0 = pt.swap ? pt.o_i : pt.o_j;
                const double x1 = x0 + (pt.swap ? lj : li), y1 = y0 + (pt.swap ? li : lj);
                const int cx0 = std::max(0, (int)std::floor((x0 - pgx0) / G)), cx1 = std::min(pgnx - 1, (int)std::floor((x1 - pgx0) / G));
                const int cy0 = std::max(0, (int)std::floor((y0 - pgy0) / G)), cy1 = std::min(pgny - 1, (int)std::floor((y1 - pgy0) / G));
                for (int cy = cy0; cy <= cy1; ++cy)
                    for (int cx = cx0; cx <= cx1; ++cx) {
                        const double ax0 = std::max(x0, pgx0 + cx * G), ax1 = std::min(x1, pgx0 + (cx + 1) * G);
                        const double ay0 = std::max(y0, pgy0 + cy * G), ay1 = std::min(y1, pgy0 + (cy + 1) * G);
                        const float area = (float)(std::max(0.0, ax1 - ax0) * std::max(0.0, ay1 - ay0));
                        const size_t c = (size_t)cy * pgnx + cx;
                        if (area <= best_area[c]) continue;
                        best_area[c] = area;
                        int32_t* e = pcells.data() + 8 * c;
                        const float inv_h = 1.0f / pt.h;
                        std::memcpy(e + 0, &pt.o_i, 4); std::memcpy(e + 1, &pt.o_j, 4); std::memcpy(e + 2, &inv_h, 4);
                        e[3] = (int32_t)pt.voff; e[4] = pt.C; e[5] = pt.ni; e[6] = pt.nj; e[7] = pt.swap;
                    }
            }
        }
    }
    // ---- BVH
    BvhBuilder bb;
    bb.tv = tv.data();
    bb.codes.resize(num_tris);
    bb.order.resize(num_tris);
    std::iota(bb.order.begin(), bb.order.end(), 0);
    {
        double sx = 1.0 / std::max((double)mx[0] - mn[0], 1e-12), sy = 1.0 / std::max((double)mx[1] - mn[1], 1e-12),
               sz = 1.0 / std::max((double)mx[2] - mn[2], 1e-12);
        std::vector<uint32_t> code_by_tri(num_tris);
        for (int f = 0; f < num_tris; ++f) {
            const float* t = tv.data() + 9 * (size_t)f;
            double cxv = ((double)t[0] + t[3] + t[6]) / 3.0, cyv = ((double)t[1] + t[4] + t[7]) / 3.0,
                   czv = ((double)t[2] + t[5] + t[8]) / 3.0;
            auto q = [](double v) { return (uint32_t)std::min(std::max(v * 1024.0, 0.0), 1023.0); };
            code_by_tri[f] = (expand_bits(q((cxv - mn[0]) * sx)) << 2) | (expand_bits(q((cyv - mn[1]) * sy)) << 1) |
                             expand_bits(q((czv - mn[2]) * sz));
        }
        std::stable_sort(bb.order.begin(), bb.order.end(),
                         [&](int32_t a, int32_t b) { return code_by_tri[a] < code_by_tri[b]; });
        for (int i = 0; i < num_tris; ++i) bb.codes[i] = code_by_tri[bb.order[i]];
    }
    bb.build(num_tris);

    // ---- upload
    int64_t bytes = 0;
    int rc;
    float* d_tv = nullptr; int32_t* d_cs = nullptr; int32_t* d_ct = nullptr; float* d_nodes = nullptr; int32_t* d_bt = nullptr;
    float* d_recs = nullptr; float* d_pverts = nullptr; int32_t* d_pcells = nullptr;
    for (size_t c = 0; c < disjoint.size(); ++c)
        if (disjoint[c]) cell_ref[2 * c] |= (int32_t)MDP_RC_CELL_DISJOINT;
    for (size_t c = 0; c < disjoint_down.size(); ++c)
        if (disjoint_down[c]) cell_ref_down[2 * c] |= (int32_t)MDP_RC_CELL_DISJOINT;
    int32_t* d_cs_down = nullptr; float* d_recs_down = nullptr;
    if (down_lists && ((rc = upload(cell_ref_down, &d_cs_down, &bytes)) || (rc = upload(recs_down, &d_recs_down, &bytes)))) {
        delete m;
        return rc;
    }
    if ((rc = upload(tv, &d_tv, &bytes)) || (rc = upload(cell_ref, &d_cs, &bytes)) || (rc = upload(recs, &d_recs, &bytes)) ||
        (rc = upload(m->h_cell_tris, &d_ct, &bytes)) || (rc = upload(bb.nodes, &d_nodes, &bytes)) ||
        (rc = upload(bb.order, &d_bt, &bytes)) || (rc = upload(pverts, &d_pverts, &bytes)) ||
        (rc = upload(pcells, &d_pcells, &bytes))) {
        delete m;
        return rc;
    }
    m->num_cell_refs = (int64_t)m->h_cell_tris.size();
    m->h_cell_tris.clear();
    m->h_cell_tris.shrink_to_fit();
    m->device_bytes = bytes;
    mdp_mesh_view_t& v = m->view;
    v.tri_verts = d_tv;
    v.num_tris = num_tris;
    v.cell_start = d_cs;
    v.cell_tris = d_ct;
    v.cell_recs = d_recs;
    // exactly vertical rays: the lists without axis-parallel walls (the full lists when the mesh has no such wall)
    v.cell_start_down = down_lists ? d_cs_down : d_cs;
    v.cell_recs_down = down_lists ? d_recs_down : d_recs;
    v.finite = 1;
    for (size_t i = 0; i < tv.size(); ++i)
        if (!std::isfinite(tv[i])) { v.finite = 0; break; }
    if (!pcells.empty()) {
        v.patch_verts = d_pverts; v.patch_cells = d_pcells;
        v.pgx0 = pgx0; v.pgy0 = pgy0; v.pg_inv = pg_inv; v.pgnx = pgnx; v.pgny = pgny;
    } else {
        (void)hipFree(d_pverts); (void)hipFree(d_pcells);
    }
    m->num_patch_verts = (int64_t)(pverts.size() / 4);
    v.gx0 = gx0; v.gy0 = gy0; v.cell = cellf; v.inv_cell = 1.0f / cellf;
    v.ncx = ncx; v.ncy = ncy;
    v.zmin = mn[2]; v.zmax = mx[2];
    v.bvh_nodes = d_nodes;
    v.bvh_tris = d_bt;
    v.num_nodes = (int32_t)(bb.nodes.size() / 8);
    // auto: terrain-like meshes (few refs per cell on average, bounded z-extent per footprint) use the
    // grid; anything whose cell lists blow up (many overlapping layers / huge triangles) uses the BVH
    double refs_per_cell = (double)total_cell_refs / ((double)ncx * ncy);
    double refs_per_tri = (double)total_cell_refs / (double)num_tris;
    int chosen = accel;
    if (chosen == 0) chosen = (refs_per_cell <= 64.0 && refs_per_tri <= 64.0) ? 1 : 2;
    v.accel = chosen;
    *out = m;
    return MDP_OK;
}

extern "C" int mdp_mesh_destroy(mdp_mesh_t* mesh) {
    if (!mesh) return MDP_OK;
    (void)hipSetDevice(mesh->device);
    (void)hipFree((void*)mesh->view.tri_verts);
    (void)hipFree((void*)mesh->view.cell_start);
    (void)hipFree((void*)mesh->view.cell_tris);
    if (mesh->view.cell_start_down != mesh->view.cell_start) (void)hipFree((void*)mesh->view.cell_start_down);
    if (mesh->view.cell_recs_down != mesh->view.cell_recs) (void)hipFree((void*)mesh->view.cell_recs_down);
    (void)hipFree((void*)mesh->view.cell_recs);
    (void)hipFree((void*)mesh->view.patch_verts);
    (void)hipFree((void*)mesh->view.patch_cells);
    (void)hipFree((void*)mesh->view.bvh_nodes);
    (void)hipFree((void*)mesh->view.bvh_tris);
    delete mesh;
    return MDP_OK;
}

extern "C" int mdp_mesh_info(const mdp_mesh_t* mesh, int64_t* num_tris, int64_t* num_cells, int64_t* num_cell_refs,
                             int64_t* num_bvh_nodes, int64_t* device_bytes, int* accel) {
    MDP_REQUIRE(mesh, "null mesh");
    if (num_tris) *num_tris = mesh->num_tris;
    if (num_cells) *num_cells = (int64_t)mesh->view.ncx * mesh->view.ncy;
    if (num_cell_refs) *num_cell_refs = mesh->num_cell_refs;
    if (num_bvh_nodes) *num_bvh_nodes = mesh->view.num_nodes;
    if (device_bytes) *device_bytes = mesh->device_bytes;
    if (accel) *accel = mesh->view.accel;
    return MDP_OK;
}

// =============================================================================================
// Kernels
// =============================================================================================
namespace {

// compile-time traversal choice: the grid instantiation carries neither the BVH stack (scratch) nor its code
template <int ACCEL>
__device__ __forceinline__ mdp_hit_t rc_query(const mdp_mesh_view_t* m, const mdp_ray_t* r, float max_t) {
    if (ACCEL == 2) return mdp_rc_bvh(m, r, max_t);
    if (ACCEL == 1) return mdp_rc_grid(m, r, max_t);
    return mdp_rc_brute(m, r, max_t);
}

// ops.py:61-127 + kernels.py:55-75: one thread per ray
template <int ACCEL>
__global__ void k_raycast(mdp_mesh_view_t mesh, int64_t num_rays, const float* __restrict__ starts,
                          const float* __restrict__ dirs, float max_dist, float* __restrict__ hits,
                          float* __restrict__ dist, float* __restrict__ normal, int32_t* __restrict__ face) {
    const float inf = __builtin_inff();
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < num_rays;
         i += (int64_t)gridDim.x * blockDim.x) {
        float ox = starts[3 * i], oy = starts[3 * i + 1], oz = starts[3 * i + 2];
        float dx = dirs[3 * i], dy = dirs[3 * i + 1], dz = dirs[3 * i + 2];
        mdp_ray_t r = mdp_rc_make_ray(ox, oy, oz, dx, dy, dz);
        mdp_hit_t h = rc_query<ACCEL>(&mesh, &r, max_dist);
        if (h.face >= 0) {
            // kernels.py:69: ray_hits = start + t * dir
            hits[3 * i] = ox + h.t * dx;
            hits[3 * i + 1] = oy + h.t * dy;
            hits[3 * i + 2] = oz + h.t * dz;
            if (dist) dist[i] = h.t;
            if (normal) {
                float nx, ny, nz;
                mdp_rc_normal(mesh.tri_verts + 9 * (int64_t)h.face, &nx, &ny, &nz);
                normal[3 * i] = nx; normal[3 * i + 1] = ny; normal[3 * i + 2] = nz;
            }
            if (face) face[i] = h.face;
        } else {
            hits[3 * i] = inf; hits[3 * i + 1] = inf; hits[3 * i + 2] = inf;
            if (dist) dist[i] = inf;
            if (normal) { normal[3 * i] = inf; normal[3 * i + 1] = inf; normal[3 * i + 2] = inf; }
            if (face) face[i] = -1;
        }
    }
}

// ray_caster.py:220-260 with the lazy-update bookkeeping of sensor_base.py:284-293.
// A block owns whole environments (slot_w threads each) so the outdated flag can be read by every
// ray thread and cleared by the env's first thread after a block barrier.
template <int ACCEL>
__global__ void k_raycaster_update(mdp_mesh_view_t mesh, int N, int R, int epb, int slot_w,
                                   const float* __restrict__ pose, int ld_pose, const float* __restrict__ drift,
                                   const float* __restrict__ starts_l, const float* __restrict__ dirs_l,
                                   int yaw_only, float max_dist, float* __restrict__ pos_w,
                                   float* __restrict__ quat_w, float* __restrict__ hits,
                                   float* __restrict__ ts, float* __restrict__ ts_last,
                                   uint8_t* __restrict__ outdated, mdp_height_scan_obs_t scan, int pose_xyzw,
                                   float tick_dt, float period) {
    const float inf = __builtin_inff();
    int le = threadIdx.x / slot_w;
    int lr = threadIdx.x - le * slot_w;
    uint64_t seed = 0, epoch = 0;
    if (scan.obs && scan.rng) { seed = scan.rng[0]; epoch = scan.rng[1 + blockIdx.x]; }
    for (int base = blockIdx.x * epb; base < N; base += gridDim.x * epb) {
        int e = base + le;
        bool active = le < epb && e < N;
        if (tick_dt > 0.0f) {
            // SensorBase.update of this physics step (sensor_base.py:197-205) done here, so the cast does not have
            // to wait for the launch that updates the rest of the scene
            if (active && lr == 0) {
                float t = ts[e] + tick_dt;
                ts[e] = t;
                if ((t - ts_last[e]) + 1e-6f >= period) outdated[e] = 1;
            }
            __syncthreads();
        }
        bool od = active && (outdated ? outdated[e] != 0 : true);
        __syncthreads();
        if (active && !od && scan.obs) {
            float sz = pos_w[3 * (int64_t)e + 2];
            for (int r = lr; r < R; r += slot_w) scan_store(scan, e, r, R, sz, hits[((int64_t)e * R + r) * 3 + 2], seed, epoch);
        }
        if (od) {
            const float* p = pose + (int64_t)e * ld_pose;
            V3 pos = ld3(p);
            if (drift) { pos.x += drift[3 * e]; pos.y += drift[3 * e + 1]; pos.z += drift[3 * e + 2]; }
            // physics transforms are (x, y, z, w) (ray_caster.py:226-232 converts them); root_state rows are (w, x, y, z)
            Q4 q = pose_xyzw ? Q4{p[6], p[3], p[4], p[5]} : ld4(p + 3);
            Q4 qr = yaw_only ? yaw_quat_halfangle(q) : q;   // (one form for every path of the sensor)
            for (int r = lr; r < R; r += slot_w) {
                V3 s = quat_apply(qr, ld3(starts_l + 3 * r));
                s = V3{s.x + pos.x, s.y + pos.y, s.z + pos.z};
                V3 d = ld3(dirs_l + 3 * r);
                if (!yaw_only) d = quat_apply(q, d);
                mdp_ray_t ray = mdp_rc_make_ray(s.x, s.y, s.z, d.x, d.y, d.z);
                mdp_hit_t h = rc_query<ACCEL>(&mesh, &ray, max_dist);
                float* o = hits + ((int64_t)e * R + r) * 3;
                float hz = inf;
                if (h.face >= 0) {
                    hz = s.z + h.t * d.z;
                    o[0] = s.x + h.t * d.x; o[1] = s.y + h.t * d.y; o[2] = hz;
                } else {
                    o[0] = inf; o[1] = inf; o[2] = inf;
                }
                if (scan.obs) scan_store(scan, e, r, R, pos.z, hz, seed, epoch);
            }
            if (lr == 0) {
                st3(pos_w + 3 * (int64_t)e, pos);
                st4(quat_w + 4 * (int64_t)e, q);
                if (ts_last) ts_last[e] = ts[e];
                if (outdated) outdated[e] = 0;
            }
        }
    }
    if (scan.obs && scan.rng && threadIdx.x == 0) scan.rng[1 + blockIdx.x] = epoch + 1;
}

// The height-scan case of k_raycaster_update (body in raycaster_update_device.hpp, shared with the post-physics kernel)
#ifndef MDP_RCD_MIN_WAVES
#define MDP_RCD_MIN_WAVES 1
#endif
// ONE: a block per env, a lane per ray (raycaster_update_down_block_one: the loop-free routine)
template <int BATCH, bool STAGED, bool ONE = false>
__global__ void __launch_bounds__(256, MDP_RCD_MIN_WAVES)
k_raycaster_update_down(RaycasterUpdateArgs a, int epb, int slot_w) {
    extern __shared__ __align__(16) unsigned char s_stage[];
    if constexpr (STAGED) raycaster_update_down_blocks_staged(a, epb, slot_w, blockIdx.x, gridDim.x, s_stage);
    else if constexpr (ONE) raycaster_update_down_block_one<BATCH>(a, blockIdx.x);
    else raycaster_update_down_blocks<BATCH>(a, epb, slot_w, blockIdx.x, gridDim.x);
}

// RayCasterCamera._update_buffers_impl (ray_caster_camera.py:257-318): one thread per (env, ray) of the outdated envs.
// The camera pose is the view pose combined with the offset (:415-430); rays are rotated by the full camera
// orientation; depth outputs are post-processed like :285-312.  Flags are only read here: k_camera_bookkeeping clears
// them afterwards (threads of one env sit in different blocks).
template <int ACCEL>
__global__ void k_raycaster_camera(mdp_mesh_view_t mesh, int N, int R, mdp_raycaster_camera_args_t a) {
    const float inf = __builtin_inff();
    const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (idx >= (int64_t)N * R) return;
    const int e = (int)(idx / R), r = (int)(idx - (int64_t)e * R);
    if (a.is_outdated && !a.is_outdated[e]) return;
    const float* t = a.view_pose + (int64_t)e * a.ld_pose;
    V3 p{t[0], t[1], t[2]};
    Q4 q = a.pose_xyzw ? Q4{t[6], t[3], t[4], t[5]} : Q4{t[3], t[4], t[5], t[6]};
    // combine_frame_transforms(pos, quat, offset_pos, offset_quat), math.py:749-781
    const V3 o = quat_apply(q, ld3(a.offset_pos + 3 * (int64_t)e));
    p = V3{p.x + o.x, p.y + o.y, p.z + o.z};
    q = quat_mul(q, ld4(a.offset_quat + 4 * (int64_t)e));
    if (r == 0) {
        st3(a.pos_w + 3 * (int64_t)e, p);
        st4(a.quat_w_world + 4 * (int64_t)e, q);
    }
    V3 s = quat_apply(q, ld3(a.ray_starts + 3 * idx));
    s = V3{s.x + p.x, s.y + p.y, s.z + p.z};
    const V3 d = quat_apply(q, ld3(a.ray_directions + 3 * idx));
    mdp_ray_t ray = mdp_rc_make_ray(s.x, s.y, s.z, d.x, d.y, d.z);
    mdp_hit_t h = rc_query<ACCEL>(&mesh, &ray, 1e6f);
    float depth = inf;
    float* hit = a.ray_hits_w + 3 * idx;
    if (h.face >= 0) {
        depth = h.t;
        hit[0] = s.x + h.t * d.x; hit[1] = s.y + h.t * d.y; hit[2] = s.z + h.t * d.z;
    } else {
        hit[0] = inf; hit[1] = inf; hit[2] = inf;
    }
    if (a.normals) {
        float nx = inf, ny = inf, nz = inf;
        if (h.face >= 0) mdp_rc_normal(mesh.tri_verts + 9 * (int64_t)h.face, &nx, &ny, &nz);
        a.normals[3 * idx] = nx; a.normals[3 * idx + 1] = ny; a.normals[3 * idx + 2] = nz;
    }
    if (a.distance_to_image_plane) {
        const Q4 inv = normalize4(quat_conjugate(q), 1e-9f);
        float v = quat_apply(inv, V3{depth * d.x, depth * d.y, depth * d.z}).x;
        if (a.depth_clipping == 1) {                 // "max": torch.clip(max=...) keeps NaN, then NaN -> max
            v = tclamp_max(v, a.max_distance);
            if (v != v) v = a.max_distance;
        } else if (a.depth_clipping == 2) {          // "zero"
            if (v > a.max_distance) v = 0.0f;
            if (v != v) v = 0.0f;
        }
        a.distance_to_image_plane[idx] = v;
    }
    if (a.distance_to_camera) {
        float v = depth;
        if (a.depth_clipping == 1) v = tclamp_max(v, a.max_distance);
        else if (a.depth_clipping == 2 && v > a.max_distance) v = 0.0f;
        a.distance_to_camera[idx] = v;
    }
}

__global__ void k_camera_bookkeeping(int N, mdp_raycaster_camera_args_t a) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N || (a.is_outdated && !a.is_outdated[e])) return;
    if (a.frame) a.frame[e] += 1;                    // ray_caster_camera.py:260
    if (a.is_outdated) {
        a.is_outdated[e] = 0;
        if (a.timestamp && a.timestamp_last_update) a.timestamp_last_update[e] = a.timestamp[e];
    }
}

template <int BATCH>
__global__ void __launch_bounds__(256)
k_raycaster_update_flagged(RaycasterUpdateArgs a) {
    raycaster_update_flagged_blocks<BATCH>(a, blockIdx.x);
}

}  // namespace

extern "C" int mdp_raycast(const mdp_mesh_t* mesh, int64_t num_rays, const float* ray_starts, const float* ray_dirs,
                           float max_dist, float* ray_hits, float* ray_dist, float* ray_normal,
                           int32_t* ray_face_id, int accel, void* stream) {
    MDP_REQUIRE(mesh && ray_starts && ray_dirs && ray_hits, "null pointer");
    MDP_REQUIRE(accel >= 0 && accel <= 3, "bad accel");
    if (num_rays == 0) return MDP_OK;
    int a = accel == 0 ? mesh->view.accel : (accel == 3 ? 0 : accel);  // 3 = brute force (tests)
#define RC_LAUNCH(A)                                                                                              \
    hipLaunchKernelGGL(k_raycast<A>, dim3(grid_for(num_rays, 128, 1 << 20)), dim3(128), 0, as_stream(stream),     \
                       mesh->view, num_rays, ray_starts, ray_dirs, max_dist, ray_hits, ray_dist, ray_normal,       \
                       ray_face_id)
    if (a == 2) RC_LAUNCH(2); else if (a == 1) RC_LAUNCH(1); else RC_LAUNCH(0);
#undef RC_LAUNCH
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_raycaster_update(const mdp_mesh_t* mesh, int num_envs, int num_rays, const float* root_pose,
                                    int ld_pose, const float* drift, const float* ray_starts_local,
                                    const float* ray_dirs_local, int attach_yaw_only, int pattern_vertical_down,
                                    float max_dist, float* pos_w, float* quat_w, float* ray_hits_w,
                                    float* timestamp, float* timestamp_last_update, uint8_t* is_outdated,
                                    const mdp_height_scan_obs_t* scan, int pose_xyzw, float tick_dt,
                                    float update_period, void* stream) {
    MDP_REQUIRE(mesh && root_pose && ray_starts_local && ray_dirs_local && pos_w && quat_w && ray_hits_w,
                "null pointer");
    MDP_REQUIRE(ld_pose >= 7, "root_pose must be (N,>=7)");
    MDP_REQUIRE((timestamp_last_update == nullptr) || timestamp, "timestamp required with timestamp_last_update");
    MDP_REQUIRE(!(tick_dt > 0.0f) || (timestamp && timestamp_last_update && is_outdated),
                "tick_dt needs timestamp, timestamp_last_update and is_outdated");
    if (num_envs == 0 || num_rays == 0) return MDP_OK;
    int epb = num_rays >= 256 ? 1 : 256 / num_rays;   // <= RCU_MAX_EPB (256) env slots of the vertical-ray routine
    int slot_w = num_rays >= 256 ? 256 : num_rays;
    int threads = ((epb * slot_w + 63) / 64) * 64;
    int64_t blocks = ((int64_t)num_envs + epb - 1) / epb;
    mdp_height_scan_obs_t sc;
    std::memset(&sc, 0, sizeof(sc));
    if (scan && scan->obs) {
        MDP_REQUIRE(scan->ld_obs >= scan->col_offset + num_rays && scan->col_offset >= 0,
                    "height-scan columns [%d, %d) do not fit the observation row (%d)", scan->col_offset,
                    scan->col_offset + num_rays, scan->ld_obs);
        const bool generated = (scan->post.noise_kind == 1 || scan->post.noise_kind == 2) && !scan->post.noise_u;
        MDP_REQUIRE(!generated || scan->rng, "generated noise needs the rng words");
        MDP_REQUIRE(!scan->normalize.out || (scan->normalize.mean32 && scan->normalize.denom &&
                                              scan->normalize.ld_out >= scan->col_offset + num_rays),
                    "height-scan normalize: bad tables / ld_out");
        sc = *scan;
        if (!generated) sc.rng = nullptr;
    }
    if (pattern_vertical_down && attach_yaw_only && mesh->view.accel == 1 && mesh->view.cell_recs) {
        static const int batch = [] { const char* b = std::getenv("MDP_RC_BATCH"); return b ? std::atoi(b) : 1; }();
        static const int64_t block_cap = [] {   // > 0: at most this many blocks, each strides over its envs
            const char* b = std::getenv("MDP_RC_BLOCKS");
            int64_t c = b ? std::atoll(b) : 0;
            return c > 0 ? c : (int64_t)(1 << 20);
        }();
        const char* one_v = std::getenv("MDP_RC_ONE");        // (0: the general routine for every launch shape)
        const bool one_env = !one_v || std::atoi(one_v) != 0;
        const char* stage_v = std::getenv("MDP_RC_STAGE");    // read at every call: tests flip it inside one process
        const int stage_env = stage_v ? std::atoi(stage_v) : 0;
        RaycasterUpdateArgs ra;
        std::memset(&ra, 0, sizeof(ra));
        ra.stage = (stage_env != 0 && mesh->view.finite) ? 1 : 0;
        const size_t stage_lds = ra.stage ? rc_stage_lds_bytes() : 0;
        if (ra.stage && epb > (int)RCU_MAX_EPB_STAGED) {
            epb = RCU_MAX_EPB_STAGED;
            threads = ((epb * slot_w + 63) / 64) * 64;
            blocks = ((int64_t)num_envs + epb - 1) / epb;
        }
        ra.mesh = mesh->view; ra.N = num_envs; ra.R = num_rays; ra.pose = root_pose; ra.ld_pose = ld_pose;
        ra.pose_xyzw = pose_xyzw; ra.drift = drift; ra.starts_l = ray_starts_local; ra.dirs_l = ray_dirs_local;
        ra.max_dist = max_dist; ra.pos_w = pos_w; ra.quat_w = quat_w; ra.hits = ray_hits_w; ra.ts = timestamp;
        ra.ts_last = timestamp_last_update; ra.outdated = is_outdated; ra.scan = sc; ra.tick_dt = tick_dt;
        ra.period = update_period; ra.reset_mask = nullptr; ra.pose_ws = nullptr; ra.pose_ready = 0;
#define RCD_LAUNCH(B)                                                                                             \
    hipLaunchKernelGGL((k_raycaster_update_down<B, false>), dim3(grid_for(blocks, 1, block_cap)), dim3(threads), 0, \
                       as_stream(stream), ra, epb, slot_w)
        if (ra.stage)
            hipLaunchKernelGGL((k_raycaster_update_down<1, true>), dim3(grid_for(blocks, 1, block_cap)), dim3(threads),
                               stage_lds, as_stream(stream), ra, epb, slot_w);
        else if (batch == 1 && one_env && epb == 1 && num_rays <= threads && blocks <= block_cap)
            hipLaunchKernelGGL((k_raycaster_update_down<1, false, true>), dim3((unsigned)blocks), dim3(threads), 0, as_stream(stream),
                               ra, epb, slot_w);
        else if (batch == 1) RCD_LAUNCH(1); else if (batch == 2) RCD_LAUNCH(2); else if (batch == 3) RCD_LAUNCH(3); else RCD_LAUNCH(4);
#undef RCD_LAUNCH
        MDP_LAUNCH_CHECK();
        return MDP_OK;
    }
#define RCU_LAUNCH(A)                                                                                             \
    hipLaunchKernelGGL(k_raycaster_update<A>, dim3(grid_for(blocks, 1, 1 << 20)), dim3(threads), 0,               \
                       as_stream(stream), mesh->view, num_envs, num_rays, epb, slot_w, root_pose, ld_pose, drift,  \
                       ray_starts_local, ray_dirs_local, attach_yaw_only, max_dist, pos_w, quat_w, ray_hits_w,     \
                       timestamp, timestamp_last_update, is_outdated, sc, pose_xyzw, tick_dt, update_period)
    if (mesh->view.accel == 2) RCU_LAUNCH(2); else if (mesh->view.accel == 1) RCU_LAUNCH(1); else RCU_LAUNCH(0);
#undef RCU_LAUNCH
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_raycaster_update_flagged(int num_envs, const mdp_raycaster_update_args_t* cast,
                                            const uint8_t* reset_mask, void* stream) {
    RaycasterUpdateArgs ra;
    int rc = raycaster_flagged_args(num_envs, cast, reset_mask, &ra);
    if (rc != MDP_OK) return rc;
    if (num_envs == 0) return MDP_OK;
    const int num_rays = cast->num_rays;
    const int threads = num_rays >= 256 ? 256 : ((num_rays + 63) / 64) * 64;
    const int64_t blocks = ((int64_t)num_envs + RCF_ENVS - 1) / RCF_ENVS;
    hipLaunchKernelGGL(k_raycaster_update_flagged<2>, dim3((unsigned)blocks), dim3(threads), 0, as_stream(stream), ra);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_raycaster_camera_update(int num_envs, const mdp_raycaster_camera_args_t* a, void* stream) {
    MDP_REQUIRE(a, "null arguments");
    MDP_REQUIRE(a->mesh && a->view_pose && a->offset_pos && a->offset_quat && a->ray_starts && a->ray_directions &&
                    a->pos_w && a->quat_w_world && a->ray_hits_w, "null pointer");
    MDP_REQUIRE(a->num_rays > 0 && a->ld_pose >= 7, "bad shape");
    MDP_REQUIRE(a->depth_clipping >= 0 && a->depth_clipping <= 2, "depth_clipping must be 0 (none), 1 (max) or 2 (zero)");
    if (num_envs <= 0) return MDP_OK;
    const int64_t total = (int64_t)num_envs * a->num_rays;
    const unsigned grid = (unsigned)((total + 255) / 256);
    hipStream_t s = as_stream(stream);
    const mdp_mesh_view_t& v = a->mesh->view;
    if (v.accel == 2) hipLaunchKernelGGL(k_raycaster_camera<2>, dim3(grid), dim3(256), 0, s, v, num_envs, a->num_rays, *a);
    else if (v.accel == 1) hipLaunchKernelGGL(k_raycaster_camera<1>, dim3(grid), dim3(256), 0, s, v, num_envs, a->num_rays, *a);
    else hipLaunchKernelGGL(k_raycaster_camera<0>, dim3(grid), dim3(256), 0, s, v, num_envs, a->num_rays, *a);
    hipLaunchKernelGGL(k_camera_bookkeeping, dim3(grid_for(num_envs, 256)), dim3(256), 0, s, num_envs, *a);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

