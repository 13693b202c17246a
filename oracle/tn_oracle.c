/*
 * tn_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * A plain-C restatement of the reference's ray -> tetrahedra hot path, used only
 * by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the
 * checker.  Nothing under tetra-nerf_amd/ may import, link or call this file.
 *
 * What it restates (all citations are into /root/reference):
 *   - face table in first-seen order ........ src/tetrahedra_tracer.cpp:21-71
 *   - all-hits collection (OptiX any-hit) ... src/optix/optix_trace_rays.cu:268-331
 *   - sort by t .............................. src/optix/optix_trace_rays.cu:78-108
 *   - dedupe / pairing / tail fill .......... src/optix/optix_trace_rays.cu:22-75,110-266
 *   - sample -> segment merge + lerp ........ src/tetrahedra_tracer.cu:115-160
 *   - barycentric gather + adjoint .......... src/tetrahedra_tracer.cu:195-248
 *   - output defaults / dtypes .............. src/py_binding.cpp:53-57,188-191
 *
 * PARITY PINNING.  The ray/triangle arithmetic of the reference lives in NVIDIA
 * OptiX 7.2-7.6 (closed source, absent from /root/reference; call sites
 * optix_trace_rays.cu:280-292,311-326).  The reference's tests hold no golden hit
 * lists for this path (only a geometric on-ray property,
 * tests/test_tetrahedra_tracer.py:204-207, and the einsum definition of the gather,
 * :410-416,444-453).  Hit lists are therefore "parity unpinned" against the real
 * reference; gather fwd/bwd IS pinned (einsum restatement in tests/).  The decisions
 * that define "bit-exact" here:
 *   - triangle test = watertight edge-function test (Woop/Benthin/Wald 2013), fp32,
 *     fixed operation order, no FMA contraction, IEEE division, double fallback
 *     when an edge function is exactly 0; accept 0 < t < 1e16, either orientation;
 *     (u,v) = weights of the face's 2nd and 3rd stored vertices (OptiX convention).
 *   - no duplicate any-hit calls; hits sorted by the total order (t, face id);
 *   - overflow: keep the M-1 nearest hits (reference: traversal-order dependent);
 *   - the out-of-bounds read at optix_trace_rays.cu:131-134 is guarded;
 *   - slots >= num_visited: visited/vertex ids = -1 (reference), barycentrics and
 *     distances = 0 (reference leaves sort scratch there);
 *   - IEEE division in the lerp (reference build uses --use_fast_math).
 *
 * Build: see oracle/Makefile  (-O2 -ffp-contract=off, OpenMP).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define TNO_EMPTY 0xFFFFFFFFu
#define TNO_EPS 1e-6f /* optix_trace_rays.cu:8 */

/* ------------------------------------------------------------------------- */
/* Face table: src/tetrahedra_tracer.cpp:21-71                                */
/* ------------------------------------------------------------------------- */

static inline void sort3(uint32_t *a, uint32_t *b, uint32_t *c) {
    /* order_faces, tetrahedra_tracer.cpp:21-33 */
    uint32_t t;
    if (*a > *b) { t = *a; *a = *b; *b = t; }
    if (*b > *c) { t = *b; *b = *c; *c = t; }
    if (*a > *b) { t = *a; *a = *b; *b = t; }
}

static inline uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x;
}

/* returns 0 ok, 1 = "A triangle is shared by more than two tetrahedra!" */
int tno_build_faces(uint64_t T, const uint32_t *cells, uint32_t *faces /*[4T,3]*/,
                    uint32_t *face_tets /*[4T,2]*/, uint64_t *F_out) {
    uint64_t cap = 16;
    while (cap < 8 * T + 16) cap <<= 1;
    uint32_t *slot = (uint32_t *)malloc(cap * sizeof(uint32_t)); /* face index or EMPTY */
    uint32_t *keys = (uint32_t *)malloc((4 * T + 1) * 3 * sizeof(uint32_t)); /* sorted triples */
    if (!slot || !keys) { free(slot); free(keys); return 2; }
    memset(slot, 0xFF, cap * sizeof(uint32_t));
    uint64_t F = 0;
    int rc = 0;
    for (uint64_t i = 0; i < T && !rc; ++i) {
        const uint32_t *c = cells + 4 * i;
        for (int j = 0; j < 4; ++j) {
            uint32_t a = c[(j + 1) % 4], b = c[(j + 2) % 4], d = c[(j + 3) % 4];
            uint32_t sa = a, sb = b, sd = d;
            sort3(&sa, &sb, &sd);
            uint64_t h = mix64(((uint64_t)sa << 42) ^ ((uint64_t)sb << 21) ^ (uint64_t)sd) & (cap - 1);
            for (;;) {
                uint32_t f = slot[h];
                if (f == TNO_EMPTY) {
                    slot[h] = (uint32_t)F;
                    keys[3 * F] = sa; keys[3 * F + 1] = sb; keys[3 * F + 2] = sd;
                    faces[3 * F] = a; faces[3 * F + 1] = b; faces[3 * F + 2] = d; /* unsorted first-seen triple */
                    face_tets[2 * F] = (uint32_t)i; face_tets[2 * F + 1] = TNO_EMPTY;
                    ++F;
                    break;
                }
                if (keys[3 * f] == sa && keys[3 * f + 1] == sb && keys[3 * f + 2] == sd) {
                    if (face_tets[2 * f + 1] != TNO_EMPTY) { rc = 1; break; }
                    face_tets[2 * f + 1] = (uint32_t)i;
                    break;
                }
                h = (h + 1) & (cap - 1);
            }
            if (rc) break;
        }
    }
    free(slot); free(keys);
    *F_out = F;
    return rc;
}

/* ------------------------------------------------------------------------- */
/* Ray / triangle: the routine that DEFINES a hit for this project            */
/* (stands in for OptiX's built-in triangle test, optix_trace_rays.cu:280-326) */
/* ------------------------------------------------------------------------- */

typedef struct {
    int kx, ky, kz;
    float Sx, Sy, Sz;
    float o[3];
} RayPre;

static inline void ray_pre(const float *o, const float *d, RayPre *r) {
    int kz = 0;
    float m = fabsf(d[0]);
    if (fabsf(d[1]) > m) { kz = 1; m = fabsf(d[1]); }
    if (fabsf(d[2]) > m) { kz = 2; }
    int kx = kz + 1; if (kx == 3) kx = 0;
    int ky = kx + 1; if (ky == 3) ky = 0;
    if (d[kz] < 0.0f) { int t = kx; kx = ky; ky = t; }
    r->kx = kx; r->ky = ky; r->kz = kz;
    r->Sx = d[kx] / d[kz];
    r->Sy = d[ky] / d[kz];
    r->Sz = 1.0f / d[kz];
    r->o[0] = o[0]; r->o[1] = o[1]; r->o[2] = o[2];
}

static inline int tri_hit(const RayPre *r, const float *p0, const float *p1, const float *p2,
                          float *t_out, float *u_out, float *v_out) {
    const int kx = r->kx, ky = r->ky, kz = r->kz;
    float A[3] = {p0[0] - r->o[0], p0[1] - r->o[1], p0[2] - r->o[2]};
    float B[3] = {p1[0] - r->o[0], p1[1] - r->o[1], p1[2] - r->o[2]};
    float C[3] = {p2[0] - r->o[0], p2[1] - r->o[1], p2[2] - r->o[2]};
    const float Ax = A[kx] - r->Sx * A[kz], Ay = A[ky] - r->Sy * A[kz];
    const float Bx = B[kx] - r->Sx * B[kz], By = B[ky] - r->Sy * B[kz];
    const float Cx = C[kx] - r->Sx * C[kz], Cy = C[ky] - r->Sy * C[kz];
    float U = Cx * By - Cy * Bx;
    float V = Ax * Cy - Ay * Cx;
    float W = Bx * Ay - By * Ax;
    if (U == 0.0f || V == 0.0f || W == 0.0f) {
        U = (float)((double)Cx * (double)By - (double)Cy * (double)Bx);
        V = (float)((double)Ax * (double)Cy - (double)Ay * (double)Cx);
        W = (float)((double)Bx * (double)Ay - (double)By * (double)Ax);
    }
    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return 0;
    const float det = (U + V) + W;
    if (det == 0.0f) return 0;
    const float Az = r->Sz * A[kz], Bz = r->Sz * B[kz], Cz = r->Sz * C[kz];
    const float T = (U * Az + V * Bz) + W * Cz;
    const float t = T / det;
    if (!(t > 0.0f && t < 1e16f)) return 0; /* tmin 0, tmax 1e16: optix_trace_rays.cu:284-285 */
    *t_out = t;
    *u_out = V / det; /* weight of the 2nd stored vertex */
    *v_out = W / det; /* weight of the 3rd stored vertex */
    return 1;
}

typedef struct {
    float t;
    uint32_t id;
    float u, v;
} Hit;

static int hit_cmp(const void *a, const void *b) {
    const Hit *x = (const Hit *)a, *y = (const Hit *)b;
    if (x->t < y->t) return -1;
    if (x->t > y->t) return 1;
    if (x->id < y->id) return -1;
    if (x->id > y->id) return 1;
    return 0;
}

typedef struct {
    Hit *h;
    size_t n, cap;
} HitVec;

static inline void hv_push(HitVec *v, Hit x) {
    if (v->n == v->cap) {
        v->cap = v->cap ? 2 * v->cap : 256;
        v->h = (Hit *)realloc(v->h, v->cap * sizeof(Hit));
    }
    v->h[v->n++] = x;
}

/* ------------------------------------------------------------------------- */
/* A plain binary BVH over faces (used for the CPU baseline; validated        */
/* against the brute-force loop in tests).  Node test = FULL-LINE slab test   */
/* on boxes padded so that it can never cull a face tri_hit accepts.          */
/* ------------------------------------------------------------------------- */

typedef struct {
    float lo[3], hi[3];
    uint32_t left, right; /* internal: children; leaf: left = first, right = count | 0x80000000 */
} BNode;

typedef struct {
    uint64_t F;
    BNode *nodes;
    uint32_t n_nodes;
    uint32_t *order; /* face ids in leaf order */
    float scene_max; /* max |coordinate| over referenced vertices */
} Bvh;

typedef struct {
    const float *xyz;
    const uint32_t *faces;
    float *cent; /* [F,3] */
    Bvh *b;
} BuildCtx;

static void face_box(const float *xyz, const uint32_t *f, float *lo, float *hi) {
    for (int k = 0; k < 3; ++k) {
        float a = xyz[3 * f[0] + k], b = xyz[3 * f[1] + k], c = xyz[3 * f[2] + k];
        lo[k] = fminf(a, fminf(b, c));
        hi[k] = fmaxf(a, fmaxf(b, c));
    }
}

static uint32_t bvh_rec(BuildCtx *cx, uint32_t first, uint32_t count) {
    Bvh *b = cx->b;
    uint32_t me = b->n_nodes++;
    BNode *n = &b->nodes[me];
    float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int k = 0; k < 3; ++k) { n->lo[k] = INFINITY; n->hi[k] = -INFINITY; }
    for (uint32_t i = first; i < first + count; ++i) {
        uint32_t f = b->order[i];
        float lo[3], hi[3];
        face_box(cx->xyz, cx->faces + 3 * (size_t)f, lo, hi);
        for (int k = 0; k < 3; ++k) {
            n->lo[k] = fminf(n->lo[k], lo[k]); n->hi[k] = fmaxf(n->hi[k], hi[k]);
            clo[k] = fminf(clo[k], cx->cent[3 * (size_t)f + k]); chi[k] = fmaxf(chi[k], cx->cent[3 * (size_t)f + k]);
        }
    }
    if (count <= 4) {
        n->left = first; n->right = count | 0x80000000u;
        return me;
    }
    int ax = 0;
    if (chi[1] - clo[1] > chi[ax] - clo[ax]) ax = 1;
    if (chi[2] - clo[2] > chi[ax] - clo[ax]) ax = 2;
    float mid = 0.5f * (clo[ax] + chi[ax]);
    uint32_t i = first, j = first + count;
    while (i < j) {
        if (cx->cent[3 * (size_t)b->order[i] + ax] < mid) ++i;
        else { --j; uint32_t t = b->order[i]; b->order[i] = b->order[j]; b->order[j] = t; }
    }
    uint32_t nl = i - first;
    if (nl == 0 || nl == count) nl = count / 2;
    uint32_t l = bvh_rec(cx, first, nl);
    uint32_t r = bvh_rec(cx, first + nl, count - nl);
    b->nodes[me].left = l; b->nodes[me].right = r;
    return me;
}

void *tno_bvh_build(uint64_t V, const float *xyz, uint64_t F, const uint32_t *faces) {
    (void)V;
    Bvh *b = (Bvh *)calloc(1, sizeof(Bvh));
    b->F = F;
    b->nodes = (BNode *)malloc((2 * F + 2) * sizeof(BNode));
    b->order = (uint32_t *)malloc((F + 1) * sizeof(uint32_t));
    float *cent = (float *)malloc((3 * F + 3) * sizeof(float));
    float smax = 0.0f;
    for (uint64_t f = 0; f < F; ++f) {
        b->order[f] = (uint32_t)f;
        for (int k = 0; k < 3; ++k) {
            float a = xyz[3 * faces[3 * f] + k], bb = xyz[3 * faces[3 * f + 1] + k], c = xyz[3 * faces[3 * f + 2] + k];
            cent[3 * f + k] = (a + bb + c) * (1.0f / 3.0f);
            smax = fmaxf(smax, fmaxf(fabsf(a), fmaxf(fabsf(bb), fabsf(c))));
        }
    }
    b->scene_max = smax;
    BuildCtx cx = {xyz, faces, cent, b};
    if (F > 0) bvh_rec(&cx, 0, (uint32_t)F);
    free(cent);
    return b;
}

void tno_bvh_free(void *h) {
    Bvh *b = (Bvh *)h;
    if (!b) return;
    free(b->nodes); free(b->order); free(b);
}

/* line-vs-padded-box; conservative w.r.t. tri_hit (see DESIGN.md "conservative culling") */
static inline int line_box(const float *o, const float *inv, const float *lo, const float *hi, float pad) {
    float tn = -INFINITY, tf = INFINITY;
    for (int k = 0; k < 3; ++k) {
        float a = ((lo[k] - o[k]) - pad) * inv[k];
        float b = ((hi[k] - o[k]) + pad) * inv[k];
        float mn = a < b ? a : b, mx = a < b ? b : a;
        /* NaN (0*inf) compares false: leaves tn/tf untouched = conservative */
        if (mn > tn) tn = mn;
        if (mx < tf) tf = mx;
    }
    /* widen by a few ulp */
    float slack = 4.0f * 1.1920929e-7f * (fabsf(tn) + fabsf(tf));
    return tn <= tf + slack || !(tn == tn) || !(tf == tf);
}

static void collect_hits_bvh(const Bvh *b, const float *xyz, const uint32_t *faces, const float *o,
                             const float *d, HitVec *out) {
    RayPre rp;
    ray_pre(o, d, &rp);
    float inv[3];
    for (int k = 0; k < 3; ++k) {
        float dk = d[k];
        if (fabsf(dk) < 1e-30f) dk = (dk < 0.0f || (dk == 0.0f && signbit(dk))) ? -1e-30f : 1e-30f;
        inv[k] = 1.0f / dk;
    }
    float omax = fmaxf(fabsf(o[0]), fmaxf(fabsf(o[1]), fabsf(o[2])));
    float pad = 16.0f * 1.1920929e-7f * (omax + b->scene_max);
    uint32_t stack[128];
    int sp = 0;
    if (b->F == 0) return;
    stack[sp++] = 0;
    while (sp) {
        const BNode *n = &b->nodes[stack[--sp]];
        if (!line_box(o, inv, n->lo, n->hi, pad)) continue;
        if (n->right & 0x80000000u) {
            uint32_t cnt = n->right & 0x7FFFFFFFu;
            for (uint32_t i = 0; i < cnt; ++i) {
                uint32_t f = b->order[n->left + i];
                const uint32_t *fv = faces + 3 * (size_t)f;
                Hit h;
                if (tri_hit(&rp, xyz + 3 * (size_t)fv[0], xyz + 3 * (size_t)fv[1], xyz + 3 * (size_t)fv[2], &h.t, &h.u, &h.v)) {
                    h.id = f;
                    hv_push(out, h);
                }
            }
        } else {
            stack[sp++] = n->left;
            stack[sp++] = n->right;
        }
    }
}

static void collect_hits_brute(const float *xyz, uint64_t F, const uint32_t *faces, const float *o,
                               const float *d, HitVec *out) {
    RayPre rp;
    ray_pre(o, d, &rp);
    for (uint64_t f = 0; f < F; ++f) {
        const uint32_t *fv = faces + 3 * f;
        Hit h;
        if (tri_hit(&rp, xyz + 3 * (size_t)fv[0], xyz + 3 * (size_t)fv[1], xyz + 3 * (size_t)fv[2], &h.t, &h.u, &h.v)) {
            h.id = (uint32_t)f;
            hv_push(out, h);
        }
    }
}

/* ------------------------------------------------------------------------- */
/* Post-process: optix_trace_rays.cu:22-75,110-266, restated literally        */
/* ------------------------------------------------------------------------- */

static inline int get_common_tetrahedra(const uint32_t *a, const uint32_t *b, uint32_t *cell) {
    /* optix_trace_rays.cu:22-37 (note: two hull faces "match" through EMPTY) */
    if (a[0] == b[0]) { *cell = a[0]; return 1; }
    if (a[0] == b[1]) { *cell = a[0]; return 1; }
    if (a[1] == b[0]) { *cell = a[1]; return 1; }
    if (a[1] == b[1]) { *cell = a[1]; return 1; }
    return 0;
}

static inline void combine_indices(const uint32_t *id1, const uint32_t *id2, float u1, float v1,
                                   float u2, float v2, uint32_t *out4, float *bc1, float *bc2) {
    /* optix_trace_rays.cu:39-75 */
    out4[0] = 0; out4[1] = id1[0]; out4[2] = id1[1]; out4[3] = id1[2];
    bc1[0] = 1.0f - u1 - v1; bc1[1] = u1; bc1[2] = v1;
    float ref2[3] = {1.0f - u2 - v2, u2, v2};
    bc2[0] = 0.0f; bc2[1] = 0.0f; bc2[2] = 0.0f;
    for (int i = 0; i < 3; ++i) {
        int was_break = 0;
        for (int j = 0; j < 3; ++j) {
            if (id1[j] == id2[i]) { bc2[j] = ref2[i]; was_break = 1; break; }
        }
        if (!was_break) out4[0] = id2[i];
    }
}

/*
 * Rows (length M) hold the sorted hits on entry:  t[j] = face id, dl[2j] = t, dl[2j+1] = 0,
 * bcs[6j..6j+2] = (u, v, 0).  Exactly the in-place scratch layout of the reference
 * (optix_trace_rays.cu:316-326).  On exit they hold the outputs.
 */
static uint32_t post_process_row(uint32_t ray_len, uint32_t M, const uint32_t *faces,
                                 const uint32_t *face_tets, uint32_t *t, float *dl, float *bcs,
                                 uint32_t *verts) {
    size_t jc = 0;
    /* phase 1: optix_trace_rays.cu:124-159 */
    for (size_t j = 0; j + 1 < ray_len; ++j) {
        if (t[j] == TNO_EMPTY) continue;
        const float dn = dl[2 * j];
        int clear_self = 0;
        for (size_t off = 1; j + off < ray_len && (t[j + off] == TNO_EMPTY || fabsf(dl[2 * (j + off)] - dn) < TNO_EPS); ++off) {
            uint32_t cell;
            if (t[j + off] != TNO_EMPTY &&
                get_common_tetrahedra(face_tets + 2 * (size_t)t[j], face_tets + 2 * (size_t)t[j + off], &cell)) {
                if (t[j] != t[j + off]) clear_self = 1;
                if (dl[2 * (j + off) + 1] > 0.0f) t[j + off] = TNO_EMPTY;
                else dl[2 * (j + off) + 1] = 1.0f;
            }
        }
        if (clear_self) {
            if (dl[2 * j + 1] > 0.0f) t[j] = TNO_EMPTY;
        }
        dl[2 * j + 1] = 0.0f;
    }
    /* phase 2: optix_trace_rays.cu:188-257 */
    for (size_t j = 0; j < ray_len; ++j) {
        if (t[j] == TNO_EMPTY) continue;
        const uint32_t orig_tj[2] = {face_tets[2 * (size_t)t[j]], face_tets[2 * (size_t)t[j] + 1]};
        float dn = dl[2 * j];
        size_t real_off = 1;
        for (size_t off = 1; j + off < ray_len && (real_off < 3 || t[j + off] == TNO_EMPTY || fabsf(dl[2 * (j + off)] - dn) < TNO_EPS); ++off) {
            if (t[j + off] == TNO_EMPTY) continue;
            uint32_t cell;
            if (get_common_tetrahedra(orig_tj, face_tets + 2 * (size_t)t[j + off], &cell)) {
                if (fabsf(dl[2 * j] - dl[2 * (j + off)]) >= TNO_EPS) {
                    const float u1 = bcs[6 * j], v1 = bcs[6 * j + 1];
                    const float u2 = bcs[6 * (j + off)], v2 = bcs[6 * (j + off) + 1];
                    uint32_t vi[4];
                    float b1[3], b2[3];
                    combine_indices(faces + 3 * (size_t)t[j], faces + 3 * (size_t)t[j + off], u1, v1, u2, v2, vi, b1, b2);
                    const float tin = dl[2 * j], tout = dl[2 * (j + off)];
                    memcpy(bcs + 6 * jc, b1, sizeof b1);
                    memcpy(bcs + 6 * jc + 3, b2, sizeof b2);
                    memcpy(verts + 4 * jc, vi, sizeof vi);
                    dl[2 * jc] = tin; dl[2 * jc + 1] = tout;
                    t[jc] = cell;
                    jc++;
                }
                if (off > 1) {
                    float f; uint32_t q;
                    for (int k = 0; k < 2; ++k) { f = dl[2 * (j + off) + k]; dl[2 * (j + off) + k] = dl[2 * (j + 1) + k]; dl[2 * (j + 1) + k] = f; }
                    for (int k = 0; k < 3; ++k) { f = bcs[6 * (j + off) + k]; bcs[6 * (j + off) + k] = bcs[6 * (j + 1) + k]; bcs[6 * (j + 1) + k] = f; }
                    q = t[j + off]; t[j + off] = t[j + 1]; t[j + 1] = q;
                }
                break;
            }
            dn = dl[2 * (j + off)];
            real_off++;
        }
    }
    /* tail: optix_trace_rays.cu:260-265 (+ our definition: zero the float scratch) */
    for (size_t j = jc; j < M; ++j) {
        t[j] = TNO_EMPTY;
        verts[4 * j] = verts[4 * j + 1] = verts[4 * j + 2] = verts[4 * j + 3] = TNO_EMPTY;
        dl[2 * j] = dl[2 * j + 1] = 0.0f;
        for (int k = 0; k < 6; ++k) bcs[6 * j + k] = 0.0f;
    }
    return (uint32_t)jc;
}

static void load_row(const Hit *h, uint32_t n, uint32_t *t, float *dl, float *bcs) {
    for (uint32_t j = 0; j < n; ++j) {
        t[j] = h[j].id;
        dl[2 * j] = h[j].t; dl[2 * j + 1] = 0.0f;
        bcs[6 * j] = h[j].u; bcs[6 * j + 1] = h[j].v; bcs[6 * j + 2] = 0.0f;
        bcs[6 * j + 3] = bcs[6 * j + 4] = bcs[6 * j + 5] = 0.0f;
    }
}

/*
 * trace_rays: PyTetrahedraTracer::trace_rays (py_binding.cpp:41-76) +
 * __raygen__rg (optix_trace_rays.cu:268-302).
 * bvh == NULL -> brute force over all faces (the definition);
 * otherwise the BVH from tno_bvh_build (CPU-baseline variant).
 * raw_* (nullable): the sorted all-hits list before post-processing, rows of M
 *   (what trace_rays_triangles would see).
 */
int tno_trace_rays(uint64_t V, const float *xyz, uint64_t F, const uint32_t *faces,
                   const uint32_t *face_tets, const void *bvh, uint64_t R, uint32_t M,
                   const float *origins, const float *dirs, uint32_t *num_visited,
                   uint32_t *visited, float *bary, float *dist, uint32_t *verts,
                   uint32_t *raw_count, uint32_t *raw_ids, float *raw_t, float *raw_uv,
                   int nthreads) {
    (void)V;
    if (M == 0 || (M & (M - 1)) != 0) return 3; /* "max_ray_triangles must be a power of 2." */
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
    {
        HitVec hv = {0, 0, 0};
#pragma omp for schedule(dynamic, 16)
        for (int64_t r = 0; r < (int64_t)R; ++r) {
            hv.n = 0;
            if (bvh) collect_hits_bvh((const Bvh *)bvh, xyz, faces, origins + 3 * r, dirs + 3 * r, &hv);
            else collect_hits_brute(xyz, F, faces, origins + 3 * r, dirs + 3 * r, &hv);
            qsort(hv.h, hv.n, sizeof(Hit), hit_cmp);
            uint32_t n = (uint32_t)(hv.n < (size_t)(M - 1) ? hv.n : (size_t)(M - 1));
            if (raw_count) {
                raw_count[r] = n;
                for (uint32_t j = 0; j < M; ++j) {
                    raw_ids[r * M + j] = j < n ? hv.h[j].id : TNO_EMPTY;
                    raw_t[r * M + j] = j < n ? hv.h[j].t : 0.0f;
                    raw_uv[2 * (r * M + j)] = j < n ? hv.h[j].u : 0.0f;
                    raw_uv[2 * (r * M + j) + 1] = j < n ? hv.h[j].v : 0.0f;
                }
            }
            uint32_t *t = visited + (size_t)r * M;
            float *dl = dist + (size_t)r * M * 2;
            float *bcs = bary + (size_t)r * M * 6;
            uint32_t *vi = verts + (size_t)r * M * 4;
            load_row(hv.h, n, t, dl, bcs);
            num_visited[r] = post_process_row(n, M, faces, face_tets, t, dl, bcs, vi);
        }
        free(hv.h);
    }
    return 0;
}

/* find_tetrahedra: optix_find_tetrahedra.cu:84-212 (two closest-hit rays +x / -x, common
 * tetrahedron, blended barycentrics).  Closest = smallest (t, face id).  Defaults (not found):
 * tetrahedron 0xFFFFFFFF, barycentrics and vertex ids 0 (torch::zeros, py_binding.cpp:121-129). */
int tno_find_tetrahedra(const float *xyz, uint64_t F, const uint32_t *faces, const uint32_t *face_tets,
                        uint64_t N, const float *points, uint32_t *tets, float *bary, uint32_t *verts) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)N; ++i) {
        uint32_t fid[2] = {TNO_EMPTY, TNO_EMPTY};
        float ht[2] = {0, 0}, hu[2] = {0, 0}, hv[2] = {0, 0};
        for (int side = 0; side < 2; ++side) {
            const float d[3] = {side == 0 ? 1.0f : -1.0f, 0.0f, 0.0f};
            RayPre rp;
            ray_pre(points + 3 * i, d, &rp);
            for (uint64_t f = 0; f < F; ++f) {
                const uint32_t *fv = faces + 3 * f;
                float t, u, v;
                if (tri_hit(&rp, xyz + 3 * (size_t)fv[0], xyz + 3 * (size_t)fv[1], xyz + 3 * (size_t)fv[2], &t, &u, &v)) {
                    if (fid[side] == TNO_EMPTY || t < ht[side]) { fid[side] = (uint32_t)f; ht[side] = t; hu[side] = u; hv[side] = v; }
                }
            }
        }
        uint32_t cell = TNO_EMPTY, vi[4] = {0, 0, 0, 0};
        float c[3] = {0, 0, 0};
        if (fid[0] != TNO_EMPTY && fid[1] != TNO_EMPTY &&
            get_common_tetrahedra(face_tets + 2 * (size_t)fid[0], face_tets + 2 * (size_t)fid[1], &cell)) {
            float c0[3], c1[3];
            combine_indices(faces + 3 * (size_t)fid[0], faces + 3 * (size_t)fid[1], hu[0], hv[0], hu[1], hv[1], vi, c0, c1);
            const float m = ht[1] / (ht[0] + ht[1]);
            for (int k = 0; k < 3; ++k) c[k] = c0[k] * m + c1[k] * (1 - m);
        } else {
            cell = TNO_EMPTY;
        }
        tets[i] = cell;
        memcpy(bary + 3 * i, c, sizeof c);
        memcpy(verts + 4 * i, vi, sizeof vi);
    }
    return 0;
}

/* post-process caller-supplied sorted hit lists (crafted tie / duplicate cases) */
int tno_postprocess(const uint32_t *faces, const uint32_t *face_tets, uint64_t R, uint32_t M,
                    const uint32_t *hit_count, const uint32_t *hit_ids, const float *hit_t,
                    const float *hit_uv, uint32_t *num_visited, uint32_t *visited, float *bary,
                    float *dist, uint32_t *verts) {
    for (uint64_t r = 0; r < R; ++r) {
        uint32_t n = hit_count[r];
        if (n > M) return 4;
        uint32_t *t = visited + r * M;
        float *dl = dist + r * M * 2;
        float *bcs = bary + r * M * 6;
        for (uint32_t j = 0; j < n; ++j) {
            t[j] = hit_ids[r * M + j];
            dl[2 * j] = hit_t[r * M + j]; dl[2 * j + 1] = 0.0f;
            bcs[6 * j] = hit_uv[2 * (r * M + j)]; bcs[6 * j + 1] = hit_uv[2 * (r * M + j) + 1];
            bcs[6 * j + 2] = bcs[6 * j + 3] = bcs[6 * j + 4] = bcs[6 * j + 5] = 0.0f;
        }
        num_visited[r] = post_process_row(n, M, faces, face_tets, t, dl, bcs, verts + r * M * 4);
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* find_visited_cells: tetrahedra_tracer.cu:115-160, py_binding.cpp:163-216   */
/* ------------------------------------------------------------------------- */
int tno_find_matched_cells(uint64_t R, uint64_t S, uint64_t M, const uint32_t *num_visited,
                           const uint32_t *visited, const float *dist /*[R,M,2]*/,
                           const float *bary /*[R,M,2,3]*/, const float *distances /*[R,S]*/,
                           const uint32_t *verts /*[R,M,4]*/, uint32_t *cells_out /*[R,S]*/,
                           uint32_t *verts_out /*[R,S,4]*/, uint8_t *mask_out /*[R,S]*/,
                           float *bary_out /*[R,S,3]*/) {
    /* defaults: py_binding.cpp:188-191 */
    memset(mask_out, 0, R * S);
    memset(cells_out, 0xFF, R * S * 4);
    memset(verts_out, 0xFF, R * S * 16);
    memset(bary_out, 0, R * S * 12);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)R; ++i) {
        uint32_t p = 0;
        for (uint64_t j = 0; j < S; ++j) {
            const float cur = distances[i * S + j];
            while (p < num_visited[i] && dist[2 * (i * M + p) + 1] < cur) p++;
            if (p >= num_visited[i]) break;
            const float tin = dist[2 * (i * M + p)], tout = dist[2 * (i * M + p) + 1];
            if (tin <= cur) {
                mask_out[i * S + j] = 1;
                cells_out[i * S + j] = visited[i * M + p];
                memcpy(verts_out + 4 * (i * S + j), verts + 4 * (i * M + p), 16);
                const float mult = (cur - tin) / (tout - tin);
                const float *c1 = bary + 6 * (i * M + p), *c2 = c1 + 3;
                for (int k = 0; k < 3; ++k) bary_out[3 * (i * S + j) + k] = (1 - mult) * c1[k] + mult * c2[k];
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* interpolate_values fwd / bwd: tetrahedra_tracer.cu:195-248                 */
/* field [Fd, V] feature-major; out [Fd, n] (py_binding.cpp:320 returns the    */
/* moveaxis(0,-1) view of it); grad_in is the [Fd, n] transposed copy (:369)   */
/* ------------------------------------------------------------------------- */
int tno_interpolate_values(uint32_t D, uint32_t V, uint32_t n, uint32_t Fd, const uint32_t *vi,
                           const float *bc, const float *field, float *out) {
    if (!(D == 2 || D == 3 || D == 4 || D == 6)) return 5;
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < (int64_t)Fd; ++j) {
        for (uint32_t i = 0; i < n; ++i) {
            float o = 0, w = 0;
            for (uint32_t k = 0; k + 1 < D; ++k) {
                const float wk = bc[(size_t)i * (D - 1) + k];
                const uint32_t v = vi[(size_t)i * D + k + 1];
                if (v != TNO_EMPTY) o += wk * field[(size_t)j * V + v];
                w += wk;
            }
            if (vi[(size_t)i * D] != TNO_EMPTY) o += (1.0f - w) * field[(size_t)j * V + vi[(size_t)i * D]];
            out[(size_t)j * n + i] = o;
        }
    }
    return 0;
}

int tno_interpolate_values_backward(uint32_t D, uint32_t V, uint32_t n, uint32_t Fd,
                                    const uint32_t *vi, const float *bc,
                                    const float *grad_in /*[Fd,n]*/, float *grad_field /*[Fd,V]*/) {
    if (!(D == 2 || D == 3 || D == 4 || D == 6)) return 5;
    memset(grad_field, 0, (size_t)Fd * V * sizeof(float));
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < (int64_t)Fd; ++j) {
        for (uint32_t i = 0; i < n; ++i) {
            const float g = grad_in[(size_t)j * n + i];
            float w = 0;
            for (uint32_t k = 0; k + 1 < D; ++k) {
                const float wk = bc[(size_t)i * (D - 1) + k];
                const uint32_t v = vi[(size_t)i * D + k + 1];
                if (v != TNO_EMPTY) grad_field[(size_t)j * V + v] += wk * g;
                w += wk;
            }
            if (vi[(size_t)i * D] != TNO_EMPTY) grad_field[(size_t)j * V + vi[(size_t)i * D]] += (1 - w) * g;
        }
    }
    return 0;
}

int tno_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
