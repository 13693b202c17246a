// tn_mesh.cpp -- host-side mesh preprocessing for libtetranerf_hip.
//
//  * build_face_table: tetrahedra -> unique faces in FIRST-SEEN order with the unsorted
//    first-seen vertex triple, plus face -> (tet, tet|EMPTY).  The ordering is observable
//    through trace_rays' vertex_indices / barycentric ordering, so it reproduces the
//    enumeration of the reference (src/tetrahedra_tracer.cpp:45-71: tets ascending, local
//    face j = vertices (j+1)%4,(j+2)%4,(j+3)%4; third sighting is an error).
//  * build_wide_bvh: 64-ary complete tree over Morton-ordered faces (replaces the OptiX GAS
//    build, src/tetrahedra_tracer.cpp:285-332).
//  * build_tet_records: 128-byte per-tet adjacency records for the walk kernel.
#include <algorithm>
#include <functional>
#include <cmath>
#include <cstring>
#include <limits>
#include <numeric>

#include "tn_build_core.h"
#include "tn_common.h"

namespace tn {

namespace {

inline uint64_t mix64(uint64_t x) {
    x ^= x >> 31; x *= 0x7fb5d329728ea185ULL;
    x ^= x >> 27; x *= 0x81dadef4bc2dd44dULL;
    x ^= x >> 33;
    return x;
}

struct Key3 {
    uint32_t a, b, c;  // ascending
};

inline Key3 sorted_key(uint32_t a, uint32_t b, uint32_t c) {
    if (a > b) std::swap(a, b);
    if (b > c) std::swap(b, c);
    if (a > b) std::swap(a, b);
    return {a, b, c};
}

// spread the low 21 bits of v so that there are two zero bits between each
inline uint64_t spread21(uint64_t v) {
    v &= 0x1fffffULL;
    v = (v | v << 32) & 0x1f00000000ffffULL;
    v = (v | v << 16) & 0x1f0000ff0000ffULL;
    v = (v | v << 8) & 0x100f00f00f00f00fULL;
    v = (v | v << 4) & 0x10c30c30c30c30c3ULL;
    v = (v | v << 2) & 0x1249249249249249ULL;
    return v;
}

}  // namespace

void build_face_table(size_t T, const uint32_t *cells, HostMesh &out) {
    size_t cap = 16;
    while (cap < 8 * T + 16) cap <<= 1;
    std::vector<uint32_t> slot(cap, TN_EMPTY);  // -> face index
    std::vector<Key3> keys;
    keys.reserve(2 * T + 16);
    out.faces.clear();
    out.face_tets.clear();
    out.faces.reserve(3 * (2 * T + 16));
    out.face_tets.reserve(2 * (2 * T + 16));
    for (size_t i = 0; i < T; ++i) {
        const uint32_t *c = cells + 4 * i;
        for (int j = 0; j < 4; ++j) {
            const uint32_t v0 = c[(j + 1) & 3], v1 = c[(j + 2) & 3], v2 = c[(j + 3) & 3];
            const Key3 k = sorted_key(v0, v1, v2);
            size_t h = mix64((uint64_t(k.a) * 0x9E3779B97F4A7C15ULL) ^ (uint64_t(k.b) << 32 | k.c)) & (cap - 1);
            for (;;) {
                const uint32_t f = slot[h];
                if (f == TN_EMPTY) {
                    slot[h] = (uint32_t)keys.size();
                    keys.push_back(k);
                    out.faces.push_back(v0); out.faces.push_back(v1); out.faces.push_back(v2);
                    out.face_tets.push_back((uint32_t)i); out.face_tets.push_back(TN_EMPTY);
                    break;
                }
                if (keys[f].a == k.a && keys[f].b == k.b && keys[f].c == k.c) {
                    if (out.face_tets[2 * (size_t)f + 1] != TN_EMPTY)
                        throw Error("A triangle is shared by more than two tetrahedra!");
                    out.face_tets[2 * (size_t)f + 1] = (uint32_t)i;
                    break;
                }
                h = (h + 1) & (cap - 1);
            }
        }
    }
}

void build_wide_bvh(const float *xyz, const uint32_t *faces, const std::vector<uint32_t> &ids,
                    HostWideBvh &out, uint32_t leaf_w) {
    const size_t LW = leaf_w;   // faces per leaf block
    out.leaf_w = leaf_w;
    const size_t n = ids.size();
    out.leaf_tri.clear(); out.leaf_id.clear(); out.boxes.clear(); out.child.clear();
    // per-face boxes and centroids
    std::vector<float> fb(6 * n), cen(3 * n);
    for (size_t i = 0; i < n; ++i) {
        const uint32_t *f = faces + 3 * (size_t)ids[i];
        for (int k = 0; k < 3; ++k) {
            const float a = xyz[3 * (size_t)f[0] + k], b = xyz[3 * (size_t)f[1] + k], c = xyz[3 * (size_t)f[2] + k];
            fb[6 * i + k] = std::min(a, std::min(b, c));
            fb[6 * i + 3 + k] = std::max(a, std::max(b, c));
            cen[3 * i + k] = (a + b + c) * (1.0f / 3.0f);
        }
    }
    // 1. binary tree by median splits along the widest centroid axis, leaves of <= 64 faces
    struct BNode { float lo[3], hi[3]; uint32_t first, count; int left, right; };
    std::vector<BNode> bn;
    bn.reserve(n / 16 + 16);
    std::vector<uint32_t> order(n);
    std::iota(order.begin(), order.end(), 0u);
    std::function<int(uint32_t, uint32_t)> build = [&](uint32_t first, uint32_t count) -> int {
        BNode nd;
        float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int k = 0; k < 3; ++k) { nd.lo[k] = INFINITY; nd.hi[k] = -INFINITY; }
        for (uint32_t i = first; i < first + count; ++i) {
            const uint32_t f = order[i];
            for (int k = 0; k < 3; ++k) {
                nd.lo[k] = std::min(nd.lo[k], fb[6 * (size_t)f + k]); nd.hi[k] = std::max(nd.hi[k], fb[6 * (size_t)f + 3 + k]);
                clo[k] = std::min(clo[k], cen[3 * (size_t)f + k]); chi[k] = std::max(chi[k], cen[3 * (size_t)f + k]);
            }
        }
        nd.first = first; nd.count = count; nd.left = nd.right = -1;
        const int me = (int)bn.size();
        bn.push_back(nd);
        if (count > leaf_w) {
            int ax = 0;
            if (chi[1] - clo[1] > chi[ax] - clo[ax]) ax = 1;
            if (chi[2] - clo[2] > chi[ax] - clo[ax]) ax = 2;
            const uint32_t half = count / 2;
            std::nth_element(order.begin() + first, order.begin() + first + half, order.begin() + first + count,
                             [&](uint32_t a, uint32_t b) { return cen[3 * (size_t)a + ax] < cen[3 * (size_t)b + ax]; });
            const int l = build(first, half);
            const int r = build(first + half, count - half);
            bn[me].left = l; bn[me].right = r;
        }
        return me;
    };
    const int root = n ? build(0, (uint32_t)n) : -1;

    // 2. leaves -> SoA triangle blocks
    std::vector<int> leaf_of(bn.size(), -1);
    for (size_t b = 0; b < bn.size(); ++b) {
        if (bn[b].left >= 0) continue;
        const size_t l = out.leaf_id.size() / LW;
        leaf_of[b] = (int)l;
        out.leaf_tri.resize((l + 1) * 9 * LW, 0.0f);
        out.leaf_id.resize((l + 1) * LW, TN_EMPTY);
        for (uint32_t i = 0; i < bn[b].count; ++i) {
            const uint32_t fid = ids[order[bn[b].first + i]];
            const uint32_t *f = faces + 3 * (size_t)fid;
            out.leaf_id[l * LW + i] = fid;
            for (int v = 0; v < 3; ++v)
                for (int k = 0; k < 3; ++k) out.leaf_tri[(l * 9 + v * 3 + k) * LW + i] = xyz[3 * (size_t)f[v] + k];
        }
    }
    if (out.leaf_id.empty()) { out.leaf_tri.assign(9 * LW, 0.0f); out.leaf_id.assign(LW, TN_EMPTY); }

    // 3. collapse to 64-wide nodes: open the child with the largest box until 64 children (or only leaves)
    std::vector<std::pair<int, uint32_t>> todo;  // (binary subtree root, wide node index)
    auto new_node = [&]() -> uint32_t {
        const uint32_t id = (uint32_t)(out.child.size() / WIDE);
        out.child.resize((size_t)(id + 1) * WIDE, TN_EMPTY);
        out.boxes.resize((size_t)(id + 1) * 6 * WIDE, 0.0f);
        for (int k = 0; k < 3; ++k)
            for (int i = 0; i < WIDE; ++i) { out.boxes[((size_t)id * 6 + k) * WIDE + i] = INFINITY; out.boxes[((size_t)id * 6 + 3 + k) * WIDE + i] = -INFINITY; }
        return id;
    };
    const uint32_t root_wide = new_node();
    (void)root_wide;
    if (root >= 0) todo.push_back({root, 0u});
    while (!todo.empty()) {
        const auto [sub, wid] = todo.back();
        todo.pop_back();
        struct HostTree {
            const std::vector<BNode> &bn;
            int left(int k) const { return bn[k].left; }
            int right(int k) const { return bn[k].right; }
            float area(int k) const { return core::box_area(bn[k].lo, bn[k].hi); }
        };
        int kid_buf[WIDE];
        const int nk = core::collapse_node(sub, HostTree{bn}, kid_buf);
        const std::vector<int> kids(kid_buf, kid_buf + nk);
        for (size_t i = 0; i < kids.size(); ++i) {
            const int k = kids[i];
            for (int a = 0; a < 3; ++a) {
                out.boxes[((size_t)wid * 6 + a) * WIDE + i] = bn[k].lo[a];
                out.boxes[((size_t)wid * 6 + 3 + a) * WIDE + i] = bn[k].hi[a];
            }
            if (bn[k].left < 0) out.child[(size_t)wid * WIDE + i] = 0x80000000u | (uint32_t)leaf_of[k];
            else {
                const uint32_t cw = new_node();
                out.child[(size_t)wid * WIDE + i] = cw;
                todo.push_back({k, cw});
            }
        }
    }
    // Worst-case occupancy of the traversal stack of collect_hits (tn_trace_general.hip): a popped node pushes all
    // its internal children (in child order) and the last pushed is popped first, so below the j-th pushed child j
    // entries are waiting.  need(node) = max(#internal children, max_j (j + need(child_j))).  Children have larger
    // node indices than their parent, so one reverse sweep evaluates it.
    out.max_stack = wide_bvh_max_stack(out.child.data(), out.child.size() / WIDE);
}

void build_tet_records(size_t T, const uint32_t *cells, const float *xyz, const HostMesh &hm,
                       std::vector<TetRec> &out, std::vector<uint32_t> &rec_of_tet) {
    const size_t F = hm.face_tets.size() / 2;
    // face id of (tet, local face): re-derive by replaying the first-seen enumeration.
    // A face's first sighting is (face_tets.x, some j); its second (face_tets.y, some j').
    // Replaying with a per-face cursor avoids a second hash table.
    out.assign(T, TetRec{});
    std::vector<uint32_t> tet_face(4 * T, TN_EMPTY);
    {
        // bucket faces by tet
        std::vector<uint32_t> deg(T + 1, 0);
        for (size_t f = 0; f < F; ++f) {
            deg[hm.face_tets[2 * f]]++;
            if (hm.face_tets[2 * f + 1] != TN_EMPTY) deg[hm.face_tets[2 * f + 1]]++;
        }
        std::vector<uint32_t> start(T + 1, 0);
        for (size_t i = 0; i < T; ++i) start[i + 1] = start[i] + deg[i];
        std::vector<uint32_t> fill(start.begin(), start.end() - 1);
        std::vector<uint32_t> bucket(start[T]);
        for (size_t f = 0; f < F; ++f) {
            bucket[fill[hm.face_tets[2 * f]]++] = (uint32_t)f;
            if (hm.face_tets[2 * f + 1] != TN_EMPTY) bucket[fill[hm.face_tets[2 * f + 1]]++] = (uint32_t)f;
        }
        for (size_t i = 0; i < T; ++i) {
            const uint32_t *c = cells + 4 * i;
            for (int j = 0; j < 4; ++j) {
                const Key3 k = sorted_key(c[(j + 1) & 3], c[(j + 2) & 3], c[(j + 3) & 3]);
                for (uint32_t s = start[i]; s < start[i + 1]; ++s) {
                    const uint32_t f = bucket[s];
                    const Key3 kf = sorted_key(hm.faces[3 * (size_t)f], hm.faces[3 * (size_t)f + 1], hm.faces[3 * (size_t)f + 2]);
                    if (kf.a == k.a && kf.b == k.b && kf.c == k.c) { tet_face[4 * i + j] = f; break; }
                }
                if (tet_face[4 * i + j] == TN_EMPTY) throw Error("internal: face of a tetrahedron not found");
            }
        }
    }
    for (size_t i = 0; i < T; ++i) {
        TetRec &r = out[i];
        const uint32_t *c = cells + 4 * i;
        r.perm = 0; r.back = 0;
        for (int k = 0; k < 4; ++k) {
            r.vert[k] = c[k];
            for (int a = 0; a < 3; ++a) r.pos[k][a] = xyz[3 * (size_t)c[k] + a];
        }
        for (int k = 0; k < 4; ++k) {
            const uint32_t f = tet_face[4 * i + k];
            r.face[k] = f;
            const uint32_t t0 = hm.face_tets[2 * (size_t)f], t1 = hm.face_tets[2 * (size_t)f + 1];
            const uint32_t nb = (t0 == (uint32_t)i) ? t1 : t0;
            r.nbr[k] = nb;
            if (nb != TN_EMPTY) {
                uint32_t bk = 0;
                for (; bk < 4; ++bk) if (tet_face[4 * (size_t)nb + bk] == f) break;
                if (bk == 4) throw Error("internal: neighbour back-pointer not found");
                r.back |= bk << (2 * k);
            }
            for (int m = 0; m < 3; ++m) {
                const uint32_t sv = hm.faces[3 * (size_t)f + m];
                uint32_t li = 0;
                for (; li < 4; ++li) if (c[li] == sv) break;
                if (li == 4) throw Error("internal: stored face vertex not in tetrahedron");
                r.perm |= li << (6 * k + 2 * m);
            }
        }
        // derived selection codes (see TetRec): edge-function reuse and combine_indices by local ids
        auto pair_code = [](uint32_t a, uint32_t b) -> uint32_t {
            static const int idx[4][4] = {{-1, 0, 1, 2}, {0, -1, 3, 4}, {1, 3, -1, 5}, {2, 4, 5, -1}};
            if (a == b) return 0;  // degenerate tet (repeated vertex): any code, the walk flags zero edge functions
            return a < b ? (uint32_t)idx[a][b] : ((uint32_t)idx[b][a] | 8u);
        };
        r.euv[0] = r.euv[1] = 0;
        r.cmb[0] = r.cmb[1] = r.cmb[2] = 0;
        uint32_t loc[4][3];
        for (int k = 0; k < 4; ++k)
            for (int m = 0; m < 3; ++m) loc[k][m] = (r.perm >> (6 * k + 2 * m)) & 3u;
        for (int k = 0; k < 4; ++k) {
            const uint32_t a = loc[k][0], b = loc[k][1], c2 = loc[k][2];
            const uint32_t code = pair_code(b, c2) | (pair_code(c2, a) << 4) | (pair_code(a, b) << 8);  // U, V, W
            r.euv[k >> 1] |= code << (12 * (k & 1));
        }
        for (int e = 0; e < 4; ++e)
            for (int x = 0; x < 4; ++x) {
                if (x == e) continue;
                uint32_t code = 0;
                for (int j = 0; j < 3; ++j) {
                    uint32_t pos = 3;
                    for (int i2 = 0; i2 < 3; ++i2) if (loc[x][i2] == loc[e][j]) { pos = (uint32_t)i2; break; }
                    code |= pos << (2 * j);
                }
                const int pid = 3 * e + x - (x > e ? 1 : 0);
                const int bit = 6 * pid;
                r.cmb[bit >> 5] |= code << (bit & 31);
                if ((bit & 31) > 26) r.cmb[(bit >> 5) + 1] |= code >> (32 - (bit & 31));
            }
    }
    // Morton order of the tet centroids: consecutive walk steps and neighbouring rays then touch
    // neighbouring 128-B lines (L2 / TLB locality); nbr[] become record indices.
    {
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        std::vector<float> cen(3 * T);
        for (size_t i = 0; i < T; ++i)
            for (int a = 0; a < 3; ++a) {
                const float c = 0.25f * (out[i].pos[0][a] + out[i].pos[1][a] + out[i].pos[2][a] + out[i].pos[3][a]);
                cen[3 * i + a] = c;
                lo[a] = std::min(lo[a], c); hi[a] = std::max(hi[a], c);
            }
        std::vector<std::pair<uint64_t, uint32_t>> order(T);
        for (size_t i = 0; i < T; ++i) {
            uint64_t code = 0;
            for (int a = 0; a < 3; ++a) {
                const double ext = (double)hi[a] - (double)lo[a];
                const double u = ext > 0 ? ((double)cen[3 * i + a] - lo[a]) / ext : 0.0;
                code |= spread21((uint64_t)std::min(2097151.0, std::max(0.0, u * 2097152.0))) << a;
            }
            order[i] = {code, (uint32_t)i};
        }
        std::sort(order.begin(), order.end());
        rec_of_tet.assign(T, 0);
        for (size_t r = 0; r < T; ++r) rec_of_tet[order[r].second] = (uint32_t)r;
        std::vector<TetRec> sorted(T);
        for (size_t r = 0; r < T; ++r) {
            TetRec rec = out[order[r].second];
            rec.orig = order[r].second;
            for (int k = 0; k < 4; ++k)
                if (rec.nbr[k] != TN_EMPTY) rec.nbr[k] = rec_of_tet[rec.nbr[k]];
            sorted[r] = rec;
        }
        out.swap(sorted);
    }
}

void build_walk_variants(const std::vector<TetRec> &recs, std::vector<WalkVar> &out) {
    const size_t T = recs.size();
    out.assign(4 * T, WalkVar{});
    for (size_t r = 0; r < T; ++r) {
        const TetRec &t = recs[r];
        core::TetAdj adj;
        adj.ok = true;
        for (int k = 0; k < 4; ++k) {
            adj.vert[k] = t.vert[k];
            adj.fid[k] = t.face[k];
            adj.nbr[k] = t.nbr[k];
            adj.back[k] = (t.back >> (2 * k)) & 3u;
            for (int m = 0; m < 3; ++m) adj.loc[k][m] = (t.perm >> (6 * k + 2 * m)) & 3u;  // stored order of face k in local ids
        }
        for (uint32_t e = 0; e < 4; ++e) out[4 * r + e] = core::make_walk_var(adj, t.nbr, t.pos[e], t.orig, e);
    }
    // thin-neighbourhood exponent (tn_build_core.h): star minima of the smallest tet height, second smallest per tet
    uint32_t V = 0;
    for (const TetRec &t : recs) for (int k = 0; k < 4; ++k) V = std::max(V, t.vert[k] + 1);
    std::vector<uint32_t> vmin(V, 0x7F800000u);
    for (const TetRec &t : recs) {
        const uint32_t bits = core::tet_min_height_bits(t.pos);
        for (int k = 0; k < 4; ++k) core::atomic_min_u32(&vmin[t.vert[k]], bits);
    }
    for (size_t r = 0; r < T; ++r) {
        const TetRec &t = recs[r];
        const uint32_t e = core::thin_exponent(vmin[t.vert[0]], vmin[t.vert[1]], vmin[t.vert[2]], vmin[t.vert[3]]);
        for (uint32_t k = 0; k < 4; ++k) out[4 * r + k].code_hi |= e << core::THIN_SHIFT;
    }
}

void build_hull_threaded(const float *xyz, const uint32_t *faces, const uint32_t *face_tets,
                         const std::vector<uint32_t> &ids, const std::vector<TetRec> &recs,
                         const std::vector<uint32_t> &rec_of_tet, HostHullBvh &out) {
    // per hull face, in `ids` order: v0.xyz, face id | v1.xyz, tet record | v2.xyz, local face
    const size_t n = ids.size();
    std::vector<float> info(n * 12);
    for (size_t i = 0; i < n; ++i) {
        const uint32_t fid = ids[i];
        const uint32_t *f = faces + 3 * (size_t)fid;
        for (int v = 0; v < 3; ++v)
            for (int k = 0; k < 3; ++k) info[i * 12 + v * 4 + k] = xyz[3 * (size_t)f[v] + k];
        // the hull face's only tet (as a record index) and the face's local index in it
        const uint32_t rec = rec_of_tet[face_tets[2 * (size_t)fid]];
        uint32_t loc = 0;
        for (; loc < 4; ++loc) if (recs[rec].face[loc] == fid) break;
        if (loc == 4) throw Error("internal: hull face not found in its tetrahedron");
        std::memcpy(&info[i * 12 + 3], &fid, 4);
        std::memcpy(&info[i * 12 + 7], &rec, 4);
        std::memcpy(&info[i * 12 + 11], &loc, 4);
    }
    build_hull_from_info(info, out);
}

// Threaded binary BVH over the hull faces, nodes in DFS pre-order: the "hit" successor of a node
// is the next node, the "miss" successor is `skip`.  A lane can traverse it without a stack and in
// a ray-independent order (all crossings are wanted, not the nearest).  Leaves hold <= 4 faces.
// `info` = 12 floats per hull face in ascending face-id order (see build_hull_threaded / core::hull_face_info).
void build_hull_from_info(const std::vector<float> &info, HostHullBvh &out) {
    const size_t n = info.size() / 12;
    out.nodes.clear();
    out.tris.clear();
    out.flat.clear();
    if (n == 0) return;
    // Morton order of the face centroids
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    std::vector<float> cent(3 * n);
    for (size_t i = 0; i < n; ++i) {
        for (int k = 0; k < 3; ++k) {
            const float c = (info[i * 12 + k] + info[i * 12 + 4 + k] + info[i * 12 + 8 + k]) * (1.0f / 3.0f);
            cent[3 * i + k] = c;
            lo[k] = std::min(lo[k], c); hi[k] = std::max(hi[k], c);
        }
    }
    std::vector<std::pair<uint64_t, uint32_t>> order(n);
    for (size_t i = 0; i < n; ++i) order[i] = {core::morton63(&cent[3 * i], lo, hi), (uint32_t)i};
    std::sort(order.begin(), order.end());
    // triangle slots in Morton order: 3 x float4 (v.xyz, w)
    out.tris.resize(n * 12);
    std::vector<float> fb(6 * n);
    for (size_t s = 0; s < n; ++s) {
        const float *src = &info[(size_t)order[s].second * 12];
        std::memcpy(&out.tris[s * 12], src, 12 * sizeof(float));
        for (int k = 0; k < 3; ++k) { fb[6 * s + k] = INFINITY; fb[6 * s + 3 + k] = -INFINITY; }
        for (int v = 0; v < 3; ++v)
            for (int k = 0; k < 3; ++k) {
                const float x = src[v * 4 + k];
                fb[6 * s + k] = std::min(fb[6 * s + k], x); fb[6 * s + 3 + k] = std::max(fb[6 * s + 3 + k], x);
            }
    }
    // recursive emit in pre-order
    std::function<void(size_t, size_t)> emit = [&](size_t a, size_t b) {
        const size_t me = out.nodes.size() / 8;
        out.nodes.resize(out.nodes.size() + 8);
        float blo[3] = {INFINITY, INFINITY, INFINITY}, bhi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (size_t s = a; s < b; ++s)
            for (int k = 0; k < 3; ++k) { blo[k] = std::min(blo[k], fb[6 * s + k]); bhi[k] = std::max(bhi[k], fb[6 * s + 3 + k]); }
        uint32_t leaf = 0xFFFFFFFFu;
        if (b - a <= 4) leaf = (uint32_t)(a << 3) | (uint32_t)(b - a);
        else {
            const size_t m = a + (b - a) / 2;
            emit(a, m);
            emit(m, b);
        }
        const uint32_t skip = (uint32_t)(out.nodes.size() / 8);
        float *nd = out.nodes.data() + me * 8;
        nd[0] = blo[0]; nd[1] = blo[1]; nd[2] = blo[2]; std::memcpy(&nd[3], &skip, 4);
        nd[4] = bhi[0]; nd[5] = bhi[1]; nd[6] = bhi[2]; std::memcpy(&nd[7], &leaf, 4);
    };
    emit(0, n);
    // flat two-level table over the same face slots (tn_common.h: HostHullBvh::flat)
    const uint32_t L = hull_flat_leaves((uint32_t)n), G = hull_flat_groups((uint32_t)n);
    if (L) {
        out.flat.assign((size_t)(G + L) * 8, 0.f);
        auto put = [&](size_t slot, const float *blo, const float *bhi, uint32_t first, uint32_t count) {
            float *nd = out.flat.data() + slot * 8;
            nd[0] = blo[0]; nd[1] = blo[1]; nd[2] = blo[2]; std::memcpy(&nd[3], &first, 4);
            nd[4] = bhi[0]; nd[5] = bhi[1]; nd[6] = bhi[2]; std::memcpy(&nd[7], &count, 4);
        };
        auto box_of = [&](size_t a, size_t b, float *blo, float *bhi) {
            for (int k = 0; k < 3; ++k) { blo[k] = INFINITY; bhi[k] = -INFINITY; }
            for (size_t s = a; s < b; ++s)
                for (int k = 0; k < 3; ++k) { blo[k] = std::min(blo[k], fb[6 * s + k]); bhi[k] = std::max(bhi[k], fb[6 * s + 3 + k]); }
        };
        float blo[3], bhi[3];
        for (uint32_t l = 0; l < L; ++l) {
            const size_t a = 2 * (size_t)l, b = std::min(a + 2, n);
            box_of(a, b, blo, bhi);
            put(G + l, blo, bhi, (uint32_t)a, (uint32_t)(b - a));
        }
        for (uint32_t g = 0; g < G; ++g) {
            const uint32_t l0 = 8 * g, l1 = std::min(l0 + 8u, L);
            box_of(2 * (size_t)l0, std::min(2 * (size_t)l1, n), blo, bhi);
            put(g, blo, bhi, l0, l1 - l0);
        }
    }
}

// Shape of the median-split binary tree over n faces (core::BinNode): BFS order, level by level; a node of `count`
// faces splits into count / 2 and count - count / 2 while count > WIDE.  frontier[l] = the partition of [0, n) after
// l splitting rounds, as node indices in position order (nodes that stopped splitting stay in the later frontiers).
void build_bin_topology(size_t n, std::vector<core::BinNode> &bn, std::vector<std::vector<uint32_t>> &frontier,
                        std::vector<uint32_t> &level_start, std::vector<uint32_t> &leaf_nodes, uint32_t leaf_w) {
    bn.clear(); frontier.clear(); level_start.clear(); leaf_nodes.clear();
    if (n == 0) return;
    bn.push_back(core::BinNode{0u, (uint32_t)n, -1, -1, -1, 0u});
    level_start.push_back(0u);
    std::vector<uint32_t> cur{0u};
    for (uint32_t level = 0;; ++level) {
        frontier.push_back(cur);
        bool any = false;
        for (uint32_t k : cur) any = any || bn[k].count > leaf_w;
        if (!any) break;
        level_start.push_back((uint32_t)bn.size());
        std::vector<uint32_t> next;
        next.reserve(2 * cur.size());
        for (uint32_t k : cur) {
            if (bn[k].count > leaf_w && bn[k].left < 0) {
                const uint32_t half = bn[k].count / 2;
                const int l = (int)bn.size();
                bn.push_back(core::BinNode{bn[k].first, half, -1, -1, -1, level + 1});
                bn.push_back(core::BinNode{bn[k].first + half, bn[k].count - half, -1, -1, -1, level + 1});
                bn[k].left = l; bn[k].right = l + 1;
                next.push_back((uint32_t)l); next.push_back((uint32_t)l + 1);
            } else {
                next.push_back(k);
            }
        }
        cur.swap(next);
    }
    level_start.push_back((uint32_t)bn.size());
    // leaves in position order = the last frontier
    for (uint32_t k : frontier.back()) { bn[k].leaf = (int32_t)leaf_nodes.size(); leaf_nodes.push_back(k); }
}

// Worst-case occupancy of the traversal stack of collect_hits for a wide tree whose children have larger node
// indices than their parent (see build_wide_bvh).
uint32_t wide_bvh_max_stack(const uint32_t *child, size_t n_nodes) {
    std::vector<uint32_t> need(n_nodes, 0);
    for (size_t w = n_nodes; w-- > 0;) {
        uint32_t j = 0, m = 0;
        for (int i = 0; i < WIDE; ++i) {
            const uint32_t ch = child[w * WIDE + i];
            if (ch == TN_EMPTY || (ch >> 31)) continue;
            m = std::max(m, j + need[ch]);
            ++j;
        }
        need[w] = std::max(m, j);
    }
    return n_nodes ? std::max<uint32_t>(need[0], 1u) : 1u;
}

}  // namespace tn
