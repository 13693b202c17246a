// tn_build_core.h -- the per-element arithmetic of the structure build (load_tetrahedra), written once for the host
// build (tn_mesh.cpp), the device build (tn_build.hip) and the CPU emulation of the device build
// (tests/host/gpu_build_emul.cpp): sorted face keys and their hash, Morton codes, the entry-face-specialised walk
// record of a (tetrahedron, entry face), the greedy collapse of a binary subtree into one 64-wide BVH node.
// Everything here is a pure function of its arguments, so the three users agree bit for bit by construction.
#pragma once
#include <cstdint>
#include <cstring>
#include <cmath>

#include "tn_common.h"

#if defined(__HIPCC__)
#define TN_HD __host__ __device__ __forceinline__
#else
#define TN_HD inline
#endif

namespace tn {
namespace core {

struct Key3 { uint32_t a, b, c; };  // ascending

TN_HD Key3 sorted_key(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t t;
    if (a > b) { t = a; a = b; b = t; }
    if (b > c) { t = b; b = c; c = t; }
    if (a > b) { t = a; a = b; b = t; }
    return Key3{a, b, c};
}
TN_HD bool same_key(const Key3 &x, const Key3 &y) { return x.a == y.a && x.b == y.b && x.c == y.c; }

TN_HD uint64_t mix64(uint64_t x) {
    x ^= x >> 31; x *= 0x7fb5d329728ea185ULL;
    x ^= x >> 27; x *= 0x81dadef4bc2dd44dULL;
    x ^= x >> 33;
    return x;
}
TN_HD uint64_t key_hash(const Key3 &k) {
    return mix64((uint64_t(k.a) * 0x9E3779B97F4A7C15ULL) ^ (uint64_t(k.b) << 32 | k.c));
}

// the three vertices of local face j of a tetrahedron, in the reference's enumeration order
// (src/tetrahedra_tracer.cpp:45-71: local face j = vertices (j+1)%4, (j+2)%4, (j+3)%4)
TN_HD void face_of_tet(const uint32_t *c, int j, uint32_t &v0, uint32_t &v1, uint32_t &v2) {
    v0 = c[(j + 1) & 3]; v1 = c[(j + 2) & 3]; v2 = c[(j + 3) & 3];
}

// spread the low 21 bits of v so that there are two zero bits between each
TN_HD uint64_t spread21(uint64_t v) {
    v &= 0x1fffffULL;
    v = (v | v << 32) & 0x1f00000000ffffULL;
    v = (v | v << 16) & 0x1f0000ff0000ffULL;
    v = (v | v << 8) & 0x100f00f00f00f00fULL;
    v = (v | v << 4) & 0x10c30c30c30c30c3ULL;
    v = (v | v << 2) & 0x1249249249249249ULL;
    return v;
}
// 63-bit Morton code of a point inside the box [lo, hi] (21 bits per axis, double arithmetic: IEEE on both sides)
TN_HD uint64_t morton63(const float c[3], const float lo[3], const float hi[3]) {
    uint64_t code = 0;
    for (int a = 0; a < 3; ++a) {
        const double ext = (double)hi[a] - (double)lo[a];
        const double u = ext > 0 ? ((double)c[a] - (double)lo[a]) / ext : 0.0;
        double q = u * 2097152.0;
        q = q < 0.0 ? 0.0 : q;
        q = q > 2097151.0 ? 2097151.0 : q;
        code |= spread21((uint64_t)q) << a;
    }
    return code;
}

// float <-> unsigned with the same order (for atomicMin / atomicMax and radix-sort keys)
TN_HD uint32_t float_ordered(float f) {
    uint32_t u;
#if defined(__HIP_DEVICE_COMPILE__)
    u = __float_as_uint(f);
#else
    std::memcpy(&u, &f, 4);
#endif
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
TN_HD float ordered_float(uint32_t u) {
    u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
    float f;
#if defined(__HIP_DEVICE_COMPILE__)
    f = __uint_as_float(u);
#else
    std::memcpy(&f, &u, 4);
#endif
    return f;
}

// ---------------------------------------------------------------------------------------------------------
// Adjacency of one tetrahedron, as the walk records need it.
//   fid[k]   face id of local face k (opposite local vertex k)
//   nbr[k]   tet behind it (caller's tet index, TN_EMPTY on the hull), back[k] its local face index there
//   loc[k][m] local vertex index (0..3) of the m-th STORED vertex of face k (first match, as the host build)
struct TetAdj {
    uint32_t vert[4];
    uint32_t fid[4];
    uint32_t nbr[4];
    uint32_t back[4];
    uint32_t loc[4][3];
    bool ok;   // false: internal inconsistency (a stored face vertex that is not a vertex of the tet)
};

// tet_face: [T][4] face id per (tet, local face); faces: [F][3] stored triples; face_tets: [F][2]
TN_HD TetAdj tet_adjacency(uint32_t i, const uint32_t *cells, const uint32_t *tet_face, const uint32_t *faces,
                           const uint32_t *face_tets) {
    TetAdj r;
    r.ok = true;
    const uint32_t *c = cells + 4 * (size_t)i;
    for (int k = 0; k < 4; ++k) r.vert[k] = c[k];
    for (int k = 0; k < 4; ++k) {
        const uint32_t f = tet_face[4 * (size_t)i + k];
        r.fid[k] = f;
        const uint32_t t0 = face_tets[2 * (size_t)f], t1 = face_tets[2 * (size_t)f + 1];
        const uint32_t nb = (t0 == i) ? t1 : t0;
        r.nbr[k] = nb;
        r.back[k] = 0;
        if (nb != TN_EMPTY) {
            uint32_t bk = 0;
            for (; bk < 4; ++bk)
                if (tet_face[4 * (size_t)nb + bk] == f) break;
            if (bk == 4) { r.ok = false; bk = 0; }
            r.back[k] = bk;
        }
        for (int m = 0; m < 3; ++m) {
            const uint32_t sv = faces[3 * (size_t)f + m];
            uint32_t li = 0;
            for (; li < 4; ++li)
                if (c[li] == sv) break;
            if (li == 4) { r.ok = false; li = 0; }
            r.loc[k][m] = li;
        }
    }
    return r;
}

constexpr uint32_t TIE_SHIFT = 16;   // bits 16..18 of WalkVar::code_hi (bits 0..3: code word, 8..15: thin exponent)

// The walk record of (tet, entry face e).  `nbr_rec[k]` = record index of the neighbour behind face k (or TN_EMPTY),
// pn = position of the vertex opposite the entry face, orig = the caller's tet id.  See WalkVar in tn_common.h.
TN_HD WalkVar make_walk_var(const TetAdj &t, const uint32_t nbr_rec[4], const float pn[3], uint32_t orig, uint32_t e) {
    WalkVar v;
    uint32_t canon[4] = {0, 0, 0, 0};  // tet-local vertex index -> {0: n, 1: a, 2: b, 3: c}
    canon[e] = 0;
    for (int m = 2; m >= 0; --m) canon[t.loc[e][m]] = (uint32_t)m + 1;  // first match wins on degenerate tets
    for (int a = 0; a < 3; ++a) v.pn[a] = pn[a];
    v.orig = orig;
    v.vid[0] = t.vert[e];
    for (int m = 0; m < 3; ++m) v.vid[m + 1] = t.vert[t.loc[e][m]];
    uint64_t codes = 0;
    for (uint32_t x = 0; x < 3; ++x) {
        const uint32_t k = t.loc[e][x];  // the exit face is the one opposite a / b / c
        v.set_fid(x, t.fid[k]);
        v.nb[x] = nbr_rec[k] == TN_EMPTY ? TN_EMPTY : 4u * nbr_rec[k] + t.back[k];
        uint32_t p[3];
        for (int m = 0; m < 3; ++m) p[m] = canon[t.loc[k][m]];
        uint32_t code = p[0] | (p[1] << 2) | (p[2] << 4);
        for (uint32_t j = 0; j < 3; ++j) {
            uint32_t pos = 3;
            for (uint32_t m = 0; m < 3; ++m)
                if (p[m] == j + 1) { pos = m; break; }
            code |= pos << (6 + 2 * j);
        }
        codes |= (uint64_t)code << (12 * x);
    }
    v.code_lo = (uint32_t)codes;
    v.code_hi = (uint32_t)(codes >> 32);
    // bits 16..18: face id of exit x > face id of the entry face -- the tie-break of the total order (t, face id) between
    // the hit on the entry face and the hit on exit x, which is all the walk ever needs of the face ids
    for (uint32_t x = 0; x < 3; ++x)
        if (t.fid[t.loc[e][x]] > t.fid[e]) v.code_hi |= 1u << (TIE_SHIFT + x);
    return v;
}

// ---------------------------------------------------------------------------------------------------------
// Binary median-split tree over n faces: its SHAPE depends on n only (a node of `count` faces splits into count / 2
// and count - count / 2 while count > WIDE), so the host lays it out and only the contents are computed on the device.
struct BinNode {
    uint32_t first, count;  // range of the (sorted) face order
    int32_t left, right;    // children (-1: leaf)
    int32_t leaf;           // leaf index in position order (-1: internal)
    uint32_t level;
};

TN_HD float box_area(const float *lo, const float *hi) {
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return dx * dy + dy * dz + dz * dx;
}

// Greedy collapse of the binary subtree rooted at `sub` into one wide node: open the internal child with the
// largest box until there are WIDE children (or only leaves).  kids[] receives the binary node indices in the order
// the children are stored; returns their number.  `Tree` provides left(k) (< 0: leaf), right(k), area(k).
template <class Tree>
TN_HD int collapse_node(int sub, const Tree &tree, int *kids) {
    float area[WIDE];   // of the internal kids; -2: leaf (never opened)
    int nk = 0;
    auto push = [&](int k) {
        kids[nk] = k;
        area[nk] = tree.left(k) >= 0 ? tree.area(k) : -2.f;
        ++nk;
    };
    if (tree.left(sub) < 0) push(sub);
    else { push(tree.left(sub)); push(tree.right(sub)); }
    for (;;) {
        int best = -1;
        float best_a = -1.f;
        for (int i = 0; i < nk; ++i)
            if (area[i] > best_a) { best_a = area[i]; best = i; }   // first of the largest; a NaN area is never opened
        if (best < 0 || nk >= WIDE) break;
        const int k = kids[best];
        const int l = tree.left(k), r = tree.right(k);
        kids[best] = l;
        area[best] = tree.left(l) >= 0 ? tree.area(l) : -2.f;
        push(r);
    }
    return nk;
}

// The tree as the device build stores it: BinNode array + boxes [n_nodes][3] lo / hi.
struct BinTreeView {
    const BinNode *bn;
    const float *lo, *hi;
    TN_HD int left(int k) const { return bn[k].left; }
    TN_HD int right(int k) const { return bn[k].right; }
    TN_HD float area(int k) const { return box_area(lo + 3 * (size_t)k, hi + 3 * (size_t)k); }
};

// ---------------------------------------------------------------------------------------------------------
// Element functions of the device build (one call = one GPU thread; the CPU emulation calls them in a loop).
TN_HD uint32_t atomic_cas_u32(uint32_t *p, uint32_t cmp, uint32_t val) {
#if defined(__HIP_DEVICE_COMPILE__)
    return atomicCAS(p, cmp, val);
#else
    const uint32_t old = *p;
    if (old == cmp) *p = val;
    return old;
#endif
}
TN_HD void atomic_or_u32(uint32_t *p, uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicOr(p, v);
#else
    *p |= v;
#endif
}
TN_HD uint32_t atomic_add_u32(uint32_t *p, uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return atomicAdd(p, v);
#else
    const uint32_t old = *p; *p += v; return old;
#endif
}

enum BuildFlag : uint32_t {
    FLAG_CELL_OOB = 1u,      // cells contains a vertex index that is out of bounds
    FLAG_TRIPLE_FACE = 2u,   // a triangle is shared by more than two tetrahedra
    FLAG_INTERNAL = 4u,      // internal inconsistency
};

// Sighting i = 4 * tet + local face.  Open-addressing table of sighting indices keyed by the sorted vertex triple: the
// first sighting to claim a slot owns it, a later one with the same key pairs up with the owner through `partner`
// (a third one is the reference's "shared by more than two tetrahedra" error).  Which of the two claims the slot
// depends on the schedule; the PAIR does not, and the face's first sighting is the smaller index of the pair.
TN_HD void face_hash_insert(uint32_t i, const uint32_t *cells, uint32_t *slot, uint64_t cap_mask, uint32_t *partner,
                            uint32_t *flags) {
    const uint32_t *c = cells + 4 * (size_t)(i >> 2);
    uint32_t v0, v1, v2;
    face_of_tet(c, (int)(i & 3u), v0, v1, v2);
    const Key3 k = sorted_key(v0, v1, v2);
    uint64_t h = key_hash(k) & cap_mask;
    for (;;) {
        const uint32_t old = atomic_cas_u32(slot + h, TN_EMPTY, i);
        if (old == TN_EMPTY) return;   // owner
        const uint32_t *co = cells + 4 * (size_t)(old >> 2);
        uint32_t w0, w1, w2;
        face_of_tet(co, (int)(old & 3u), w0, w1, w2);
        if (same_key(k, sorted_key(w0, w1, w2))) {
            const uint32_t prev = atomic_cas_u32(partner + old, TN_EMPTY, i);
            if (prev != TN_EMPTY) atomic_or_u32(flags, FLAG_TRIPLE_FACE);
            else partner[i] = old;
            return;
        }
        h = (h + 1) & cap_mask;
    }
}
// first sighting of its face?
TN_HD bool face_is_first(uint32_t i, const uint32_t *partner) { return partner[i] == TN_EMPTY || i < partner[i]; }
// face table entry of a first sighting i with face id f
TN_HD void face_emit(uint32_t i, uint32_t f, const uint32_t *cells, const uint32_t *partner, uint32_t *faces,
                     uint32_t *face_tets, uint32_t *tet_face) {
    const uint32_t *c = cells + 4 * (size_t)(i >> 2);
    uint32_t v0, v1, v2;
    face_of_tet(c, (int)(i & 3u), v0, v1, v2);
    faces[3 * (size_t)f] = v0; faces[3 * (size_t)f + 1] = v1; faces[3 * (size_t)f + 2] = v2;
    const uint32_t p = partner[i];
    face_tets[2 * (size_t)f] = i >> 2;
    face_tets[2 * (size_t)f + 1] = p == TN_EMPTY ? TN_EMPTY : (p >> 2);
    tet_face[i] = f;
    if (p != TN_EMPTY) tet_face[p] = f;
}
// centroid of tet i exactly as the host build computes it
TN_HD void tet_centroid(uint32_t i, const uint32_t *cells, const float *xyz, float c[3]) {
    const uint32_t *v = cells + 4 * (size_t)i;
    for (int a = 0; a < 3; ++a)
        c[a] = 0.25f * (xyz[3 * (size_t)v[0] + a] + xyz[3 * (size_t)v[1] + a] + xyz[3 * (size_t)v[2] + a] + xyz[3 * (size_t)v[3] + a]);
}
// walk record of (record r, entry e); order[r] = tet of record r, rec_of_tet its inverse
TN_HD WalkVar walk_var_of(uint32_t r, uint32_t e, const uint32_t *order, const uint32_t *rec_of_tet, const uint32_t *cells,
                          const float *xyz, const uint32_t *tet_face, const uint32_t *faces, const uint32_t *face_tets,
                          uint32_t *flags) {
    const uint32_t i = order[r];
    const TetAdj adj = tet_adjacency(i, cells, tet_face, faces, face_tets);
    if (!adj.ok) atomic_or_u32(flags, FLAG_INTERNAL);
    uint32_t nbr_rec[4];
    for (int k = 0; k < 4; ++k) nbr_rec[k] = adj.nbr[k] == TN_EMPTY ? TN_EMPTY : rec_of_tet[adj.nbr[k]];
    float pn[3];
    for (int a = 0; a < 3; ++a) pn[a] = xyz[3 * (size_t)adj.vert[e] + a];
    return make_walk_var(adj, nbr_rec, pn, i, e);
}
// ---------------------------------------------------------------------------------------------------------
// "Thin neighbourhood" of a tetrahedron (the walk's certification rule 8, tn_trace_walk.hip): the rounded projection
// of the mesh along a ray can FOLD only where a tetrahedron is thin enough for the rounding of its projected vertices to
// invert it (its smallest height comparable to the rounding distance).  Per tet: h = its smallest height; per vertex:
// the minimum of h over the vertex's star; per tet again: the SECOND smallest of its four vertex minima = a lower
// bound of min h over the tets around its most suspicious edge (the ring of edge (a,b) lies in star(a) and star(b)).
// Stored as the float's exponent byte (floor(log2) + 127) in bits 8..15 of WalkVar::code_hi; 0 = degenerate.
TN_HD uint32_t tet_min_height_bits(const float p[4][3]) {
    // The exponent byte of this value is compared across the host build, the device build and the CPU emulation (byte-equal
    // walk records), so every product below must round once on all of them: the library is built with -ffp-contract=off;
    // the pragma pins it for any other translation unit that includes this header (clang honours it per block, gcc's
    // default for ISO C++ is already "off")
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    double q[4][3];
    for (int i = 0; i < 4; ++i) for (int a = 0; a < 3; ++a) q[i][a] = (double)p[i][a];
    auto cross = [](const double *u, const double *v, double *w) {
        w[0] = u[1] * v[2] - u[2] * v[1]; w[1] = u[2] * v[0] - u[0] * v[2]; w[2] = u[0] * v[1] - u[1] * v[0];
    };
    double e1[3], e2[3], e3[3], n[3];
    for (int a = 0; a < 3; ++a) { e1[a] = q[1][a] - q[0][a]; e2[a] = q[2][a] - q[0][a]; e3[a] = q[3][a] - q[0][a]; }
    cross(e1, e2, n);
    const double vol6 = fabs(n[0] * e3[0] + n[1] * e3[1] + n[2] * e3[2]);
    double hmin = 1e300;
    const int f[4][3] = {{1, 2, 3}, {0, 2, 3}, {0, 1, 3}, {0, 1, 2}};
    for (int k = 0; k < 4; ++k) {
        double u[3], v[3], w[3];
        for (int a = 0; a < 3; ++a) { u[a] = q[f[k][1]][a] - q[f[k][0]][a]; v[a] = q[f[k][2]][a] - q[f[k][0]][a]; }
        cross(u, v, w);
        const double area2 = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
        const double h = area2 > 0.0 ? vol6 / area2 : 0.0;
        hmin = h < hmin ? h : hmin;
    }
    const float hf = (float)hmin;
    uint32_t bits;
#if defined(__HIP_DEVICE_COMPILE__)
    bits = __float_as_uint(hf);
#else
    std::memcpy(&bits, &hf, 4);
#endif
    return bits;   // non-negative floats order like their bit patterns
}
TN_HD void atomic_min_u32(uint32_t *p, uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicMin(p, v);
#else
    if (v < *p) *p = v;
#endif
}
// exponent byte of the second smallest of the four star minima
TN_HD uint32_t thin_exponent(uint32_t m0, uint32_t m1, uint32_t m2, uint32_t m3) {
    uint32_t lo = m0 < m1 ? m0 : m1, hi = m0 < m1 ? m1 : m0;   // smallest, second smallest so far
    if (m2 < lo) { hi = lo; lo = m2; } else if (m2 < hi) hi = m2;
    if (m3 < lo) { hi = lo; lo = m3; } else if (m3 < hi) hi = m3;
    return (hi >> 23) & 0xFFu;
}
constexpr uint32_t THIN_SHIFT = 8;   // bits 8..15 of WalkVar::code_hi (bits 0..3: the code word's top bits)

// the 12 floats the hull tree build wants per hull face (tn_mesh.cpp: build_hull_from_info): v0.xyz, face id |
// v1.xyz, tet record | v2.xyz, local face
TN_HD void hull_face_info(uint32_t fid, const uint32_t *faces, const uint32_t *face_tets, const uint32_t *tet_face,
                          const uint32_t *rec_of_tet, const float *xyz, uint32_t *out12, uint32_t *flags) {
    const uint32_t *f = faces + 3 * (size_t)fid;
    for (int v = 0; v < 3; ++v) {
        for (int k = 0; k < 3; ++k) {
            const float x = xyz[3 * (size_t)f[v] + k];
            uint32_t u;
#if defined(__HIP_DEVICE_COMPILE__)
            u = __float_as_uint(x);
#else
            std::memcpy(&u, &x, 4);
#endif
            out12[v * 4 + k] = u;
        }
    }
    const uint32_t tet = face_tets[2 * (size_t)fid];
    uint32_t loc = 0;
    for (; loc < 4; ++loc)
        if (tet_face[4 * (size_t)tet + loc] == fid) break;
    if (loc == 4) { atomic_or_u32(flags, FLAG_INTERNAL); loc = 0; }
    out12[3] = fid; out12[7] = rec_of_tet[tet]; out12[11] = loc;
}
// box and centroid of face f (host: build_wide_bvh)
TN_HD void face_box(uint32_t f, const uint32_t *faces, const float *xyz, float *fb6, float *cen3) {
    const uint32_t *t = faces + 3 * (size_t)f;
    for (int k = 0; k < 3; ++k) {
        const float a = xyz[3 * (size_t)t[0] + k], b = xyz[3 * (size_t)t[1] + k], c = xyz[3 * (size_t)t[2] + k];
        const float lo_bc = c < b ? c : b, hi_bc = b < c ? c : b;   // std::min(b, c), std::max(b, c)
        fb6[k] = lo_bc < a ? lo_bc : a;                             // std::min(a, .)
        fb6[3 + k] = a < hi_bc ? hi_bc : a;                         // std::max(a, .)
        cen3[k] = (a + b + c) * (1.0f / 3.0f);
    }
}
// split axis of a segment from its centroid bounds (host: widest extent, ties to the lower axis)
TN_HD int split_axis(const float clo[3], const float chi[3]) {
    int ax = 0;
    if (chi[1] - clo[1] > chi[ax] - clo[ax]) ax = 1;
    if (chi[2] - clo[2] > chi[ax] - clo[ax]) ax = 2;
    return ax;
}

}  // namespace core
}  // namespace tn
