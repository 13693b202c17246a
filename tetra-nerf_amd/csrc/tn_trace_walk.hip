// tn_trace_walk.hip -- adjacency-walk trace path: ONE LANE PER RAY.
//
// A Delaunay tetrahedralisation has a convex hull, so a ray's faces form one chain
// f0 < f1 < ... < fn in which consecutive faces bound the tetrahedron between them.  Instead
// of collecting all hits through a BVH and sorting them (the reference's structure,
// src/optix/optix_trace_rays.cu:268-331 + :78-108), a lane walks the chain: find the two hull
// faces the ray's line crosses, then step tet -> neighbour tet through 64-byte records specialised
// by entry face (WalkVar, tn_common.h), producing the faces already in order.  Per step: one
// dependent 64-B record, ONE vertex shear, three edge functions against the carried entry face to
// pick the exit, the exit face's three edge functions in its stored order, one (t,u,v).
//
// Parity by construction: every (t,u,v) is computed by the same expression tree, in the face's
// STORED vertex order, as the general path / the oracle (tri_finish in tn_device.h); the
// emitted segment is bit-identical to what sort + post_process_tetrahedra yield whenever the
// chain is "certified":
//   (S1) t strictly increases along the chain   (sorted order == chain order, no id tie-breaks)
//   (S2) no two consecutive gaps below eps      (reference phase 1 is then a no-op and phase 2
//                                                pairs j with j+1, dropping pairs < eps;
//                                                DESIGN.md "clean chain")
//   (S3) exactly two hull faces are crossed, every tet on the way has exactly two crossed
//        faces, no edge function is exactly 0, every recorded t is in (0, 1e16), < M-1 faces.
// A ray that violates a condition is handed over: when only the ORDER of a sound chain is uncertified (S1 /
// S2; 97 % of the hand-overs) it goes to `rewalk_list` -- k_walk_collect walks the chain again recording raw
// hits into the ray's own rows and k_postprocess_rows (tn_trace_general.hip) sorts and pairs them literally --
// otherwise to `fallback_list`, re-traced by the BVH all-hits kernel.  Both rewrite the ray's rows, so the
// union is oracle-identical.
//
// Memory behaviour: the per-step segment stores are per-lane (rows are 26 KB apart, written two segments at a
// time with 16-B stores); the constant tail of every row (about 88 % of all bytes at M=512) is streamed by
// k_fill_tails with 16-byte stores, one wave per row span; blockIdx is remapped so each XCD owns runs of 16
// consecutive blocks (4096 neighbouring rays) and its L2 keeps the tets they cross.
#include "tn_device.h"
#include "tn_kernels.h"

namespace tn {

namespace {

constexpr int WALK_BLOCK = 256;
constexpr uint32_t XCD_GROUP = 16;  // consecutive blocks per XCD run (4096 rays)
constexpr uint32_t MAX_WALK_STEPS = 1u << 20;

// Selects are written as bit tests on purpose (v_cndmask): an `i == 0 ? a : i == 1 ? b : ...` chain is turned into a
// switch by the optimiser and then lowered to exec-masked branches -- a dozen of those per step cost more than the
// arithmetic.

// fill dwords [start, end) of `base` with `value`; base 16-byte aligned.  Wave-cooperative.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void fill_dwords(uint32_t *__restrict__ base, uint32_t start, uint32_t end, uint32_t value, int lane) {
    const uint32_t a0 = (start + 3u) & ~3u;  // first 16-B aligned dword
    const uint32_t head_end = a0 < end ? a0 : end;
    if (start + lane < head_end) base[start + lane] = value;
    if (a0 >= end) return;
    const uint32_t a1 = end & ~3u;
    u32x4 *b4 = reinterpret_cast<u32x4 *>(base);
    const u32x4 v4 = {value, value, value, value};
    for (uint32_t i = (a0 >> 2) + lane; i < (a1 >> 2); i += 64) {
        b4[i] = v4;  // plain stores: nontemporal ones measured slower for this pure write stream
    }
    if (a1 + lane < end) base[a1 + lane] = value;
}

__device__ __forceinline__ float sel4f(float a, float b, float c, float d, uint32_t i) {
    const bool b0 = (i & 1u) != 0, b1 = (i & 2u) != 0;
    const float lo = b0 ? b : a, hi = b0 ? d : c;
    return b1 ? hi : lo;
}

// a walk variant record as four 16-B quads: q0 = (pn.xyz, orig), q1 = vid, q2 = (nb0..2, code_hi), q3 = (fid0..2, code_lo)
struct Var { uint4 q0, q1, q2, q3; };
__device__ __forceinline__ Var load_var(const WalkVar *vars, uint32_t c) {
    const uint4 *r = reinterpret_cast<const uint4 *>(vars + c);
    Var v;
    v.q0 = r[0]; v.q1 = r[1]; v.q2 = r[2]; v.q3 = r[3];
    return v;
}
__device__ __forceinline__ uint32_t sel3u(const uint4 &v, uint32_t i) {  // i in 0..2 (3 -> z)
    const bool b0 = (i & 1u) != 0, b1 = (i & 2u) != 0;
    const uint32_t lo = b0 ? v.y : v.x;
    return b1 ? v.z : lo;
}
__device__ __forceinline__ SV selsv(const SV &p0, const SV &p1, const SV &p2, const SV &p3, uint32_t i) {
    SV r;
    r.x = sel4f(p0.x, p1.x, p2.x, p3.x, i); r.y = sel4f(p0.y, p1.y, p2.y, p3.y, i); r.z = sel4f(p0.z, p1.z, p2.z, p3.z, i);
    return r;
}

}  // namespace

// The walk runs on entry-face-specialised records (WalkVar, tn_common.h): nothing of the entry face is permuted
// or recomputed, one vertex is sheared per step, three edge functions against it decide the exit, and the exit
// face's edge functions are evaluated directly in its stored order (E(P,Q) == -E(Q,P) bitwise, so they equal the
// shared ones): bit-identical hits for 30 % fewer instructions than a per-tet record with dynamic selects.
__global__ __launch_bounds__(WALK_BLOCK) void k_trace_walk(WalkParams p) {
    const TraceParams &t = p.t;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t M = t.M;

    // XCD-aware block remap: hardware places block b on XCD b % 8; give each XCD a contiguous band
    // Consecutive blocks trace neighbouring rays that cross the same tets, so each XCD gets RUNS
    // of XCD_GROUP consecutive blocks (its L2 keeps their tets) while the runs still interleave
    // across the frame (an XCD owning one contiguous band would own all the misses or all the
    // long rays).
    const uint32_t nblk = (uint32_t)((t.num_items + WALK_BLOCK - 1) / WALK_BLOCK);
    uint32_t lb = blockIdx.x;
    if (!(p.debug & 4u)) {
        const uint32_t G = (p.debug & 8u) ? (nblk + 7) / 8 : XCD_GROUP;
        const uint32_t super = blockIdx.x / (8 * G), rem = blockIdx.x % (8 * G);
        lb = super * 8 * G + (rem & 7) * G + (rem >> 3);
    }
    if (lb >= nblk) return;
    const size_t ray = (size_t)lb * WALK_BLOCK + threadIdx.x;
    const bool active = ray < t.num_items;
    const size_t rr = active ? ray : 0;

    const float ox = t.origins[3 * rr], oy = t.origins[3 * rr + 1], oz = t.origins[3 * rr + 2];
    const float dx = t.dirs[3 * rr], dy = t.dirs[3 * rr + 1], dz = t.dirs[3 * rr + 2];
    const RayPre rp = ray_pre(ox, oy, oz, dx, dy, dz);

    bool flag = false;  // ray must be re-traced by the general path
    uint32_t why = 0;   // first reason (1..12), counted in stats[4 + why]

    // ------------------------------------------------------------------ hull crossing search
    // Wave-uniform traversal of the (small) hull BVH: a node is visited if ANY lane's line hits
    // its padded box; box / triangle data are read through uniform (scalar) loads, every lane
    // tests its own ray.  No stack: the tree has a fixed depth (<= 3 internal levels).
    uint32_t nhull = 0;
    uint32_t hf0 = TN_EMPTY, hf1 = TN_EMPTY, hc0 = 0, hc1 = 0, he0 = 0, he1 = 0, hs0 = 0, hs1 = 0;
    float ht0 = 0.f, ht1 = 0.f;
    auto hull_face = [&](const SV &A, const SV &B, const SV &C, uint32_t fid, uint32_t rec, uint32_t loc, uint32_t slot) {
        const float U = edge_f(B, C), V = edge_f(C, A), W = edge_f(A, B);
        const bool mixed = (U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f);
        if (!mixed) {
            // crossed (or degenerate: a zero edge function -> the general path decides)
            const float det = (U + V) + W;
            if (U == 0.0f || V == 0.0f || W == 0.0f || det == 0.0f) { flag = true; why = 1; }
            const float T = (U * A.z + V * B.z) + W * C.z;
            const float tt = T / det;
            if (nhull == 0) { hf0 = fid; ht0 = tt; hc0 = rec; he0 = loc; hs0 = slot; }
            else if (nhull == 1) { hf1 = fid; ht1 = tt; hc1 = rec; he1 = loc; hs1 = slot; }
            nhull++;
        }
    };
    {
        // Per-lane stackless traversal of the threaded hull tree (DFS pre-order, skip links):
        // ray-independent visiting order, every crossing of the ray's LINE is found.  Works for
        // incoherent batches (random training rays) as well as for camera frames.
        const float ix = safe_inv(dx), iy = safe_inv(dy), iz = safe_inv(dz);
        const float pad = 16.0f * 1.1920929e-7f * (fmaxf(fabsf(ox), fmaxf(fabsf(oy), fabsf(oz))) + p.scene_max);
        uint32_t i = active ? 0u : p.n_hull_nodes;
        while (i < p.n_hull_nodes) {
            const float4 a = p.hull_nodes[2 * (size_t)i], b = p.hull_nodes[2 * (size_t)i + 1];
            if (!line_box(ox, oy, oz, ix, iy, iz, a.x, a.y, a.z, b.x, b.y, b.z, pad)) { i = __float_as_uint(a.w); continue; }
            const uint32_t leaf = __float_as_uint(b.w);
            if (leaf != TN_EMPTY) {
                const uint32_t first = leaf >> 3, cnt = leaf & 7u;
                for (uint32_t k = 0; k < cnt; ++k) {
                    const float4 *tp = p.hull_tris + 3 * (size_t)(first + k);
                    const float4 v0 = tp[0], v1 = tp[1], v2 = tp[2];
                    hull_face(shear(rp, v0.x, v0.y, v0.z), shear(rp, v1.x, v1.y, v1.z), shear(rp, v2.x, v2.y, v2.z),
                              __float_as_uint(v0.w), __float_as_uint(v1.w), __float_as_uint(v2.w), first + k);
                }
            }
            i = i + 1;
        }
    }
    if (nhull != 0 && nhull != 2) { flag = true; why = 2; }
    if (nhull == 2 && !(ht0 < ht1 || ht1 < ht0)) { flag = true; why = 3; }  // equal or NaN
    if (!active) { flag = false; nhull = 0; }

    // ------------------------------------------------------------------ the walk
    // Segment stores go through LDS: a lane appends its segments to a 4-slot buffer (cells[4] | dist[4][2] |
    // bary[4][6] | verts[4][4] = 13 x 16 B), and whenever a lane's buffer is full the WAVE writes it out, four rays
    // per store instruction, lane i copying 16-B chunk i % 13 of ray i / 13: consecutive lanes hit consecutive
    // addresses, so a ray's 208 B leave as ~5 line transactions instead of 14 scattered 16-B ones (the per-lane
    // stores were the larger half of the walk: 1.64 ms -> 0.85 ms without them).  All 64 lanes stay in the loop
    // until the last ray of the wave is done so that they can help.
    extern __shared__ __attribute__((aligned(16))) uint32_t seg_lds[];
    uint32_t *mybuf = seg_lds + (size_t)threadIdx.x * 52;
    const size_t wave_ray0 = (size_t)lb * WALK_BLOCK + (size_t)wave * 64;

    uint32_t nseg = 0;
    uint32_t slot_start = 0, f_end = 0;  // entry hull triangle / hull exit face, for the re-walk of an uncertified chain
    bool alive = nhull == 2 && !flag;
    const bool first0 = ht0 < ht1;
    const uint32_t f_out = first0 ? hf1 : hf0;
    uint32_t fid_in = first0 ? hf0 : hf1;                        // id of the face the current tet was entered through
    uint32_t c = 4u * (first0 ? hc0 : hc1) + (first0 ? he0 : he1);  // variant = (tet record, entry face)
    if (!alive) c = 0;
    slot_start = first0 ? hs0 : hs1; f_end = f_out;
    // The entry face in its STORED order: sheared vertices A,B,C and edge functions U=E(B,C), V=E(C,A), W=E(A,B).
    // From here on they are carried: the exit face of a step, evaluated in its stored order, is the entry face of
    // the next (same face-table entry), so per step only ONE vertex is sheared and three edge functions against
    // it decide the exit.
    SV A = {0.f, 0.f, 0.f}, B = A, C = A;
    if (alive) {
        const float4 *tp = p.hull_tris + 3 * (size_t)(first0 ? hs0 : hs1);
        const float4 v0 = tp[0], v1 = tp[1], v2 = tp[2];
        A = shear(rp, v0.x, v0.y, v0.z); B = shear(rp, v1.x, v1.y, v1.z); C = shear(rp, v2.x, v2.y, v2.z);
    }
    float Uc = edge_f(B, C), Vc = edge_f(C, A), Wc = edge_f(A, B);
    // state of the previous recorded (valid) hit
    bool have_prev = false, have_pp = false, pending_inv = false, had_special = false;
    float pt = 0.f, pu = 0.f, pv = 0.f, ppt = 0.f;
    uint32_t run = 0;  // current run of consecutive gaps below eps
    uint32_t nhits = 0;
    uint32_t steps = 0;
    Var cur = load_var(p.vars, c);
    if (alive) {
        // the entry hull face itself may be the first recorded hit
        float tt, uu, vv;
        if (tri_finish(Uc, Vc, Wc, A.z, B.z, C.z, tt, uu, vv)) { have_prev = true; pt = tt; pu = uu; pv = vv; nhits = 1; }
    }

    for (;;) {
        bool need_flush = false;
        if (alive) {
            // All checks of a step accumulate into `bad` (first reason kept) and are acted on ONCE at the
            // end of the step: one divergence point per step instead of a dozen.
            uint32_t bad = 0;
            const SV P = shear(rp, __uint_as_float(cur.q0.x), __uint_as_float(cur.q0.y), __uint_as_float(cur.q0.z));
            const float ea = edge_f(P, A), eb = edge_f(P, B), ec = edge_f(P, C);
            if (ea == 0.0f || eb == 0.0f || ec == 0.0f) bad = 5;
            // exit candidates: the faces opposite a {n,b,c}, b {n,c,a}, c {n,a,b}; a face is crossed iff its three
            // cyclic edge functions agree in sign: E(n,b), E(b,c) = Uc, E(c,n) = -ec, and cyclically
            const bool sa = ea > 0.0f, sb = eb > 0.0f, sc = ec > 0.0f;
            const bool su = Uc > 0.0f, sv = Vc > 0.0f, sw = Wc > 0.0f;
            const bool ha = (sb == su) && (su != sc);
            const bool hb = (sc == sv) && (sv != sa);
            const bool hc = (sa == sw) && (sw != sb);
            const uint32_t hmask = (ha ? 1u : 0u) | (hb ? 2u : 0u) | (hc ? 4u : 0u);
            if (!bad && __popc(hmask) != 1) bad = 6;
            const uint32_t x = (__ffs(hmask) - 1) & 3u;  // exit 0..2 (3 only together with bad)
            const uint32_t nb = sel3u(cur.q2, x);
            const uint32_t fx = sel3u(cur.q3, x);        // id of the exit face
            const bool last = nb == TN_EMPTY;
            // the next record is requested as soon as the exit is known
            const Var nxt = load_var(p.vars, (last || bad) ? c : nb);
            __builtin_amdgcn_sched_barrier(0);

            // the exit face in its stored order: 12-bit code of exit x out of the 36-bit word
            const bool x0 = (x & 1u) != 0, x1 = (x & 2u) != 0;
            const uint32_t w01 = x0 ? (cur.q3.w >> 12) : cur.q3.w;
            const uint32_t w2 = (cur.q3.w >> 24) | (cur.q2.w << 8);
            const uint32_t code = (x1 ? w2 : w01) & 0xFFFu;
            const SV A2 = selsv(P, A, B, C, code & 3u), B2 = selsv(P, A, B, C, (code >> 2) & 3u), C2 = selsv(P, A, B, C, (code >> 4) & 3u);
            const float U = edge_f(B2, C2), V = edge_f(C2, A2), W = edge_f(A2, B2);
            float ct = 0.f, cu = 0.f, cv = 0.f;
            const bool valid = tri_finish(U, V, W, A2.z, B2.z, C2.z, ct, cu, cv);

            bool do_emit = false;
            if (valid && have_prev) {
                const bool is_short = fabsf(pt - ct) < TN_EPS;
                bool ascending = ct > pt;
                if (ct == pt) ascending = fx > fid_in;  // exact tie: the sort orders the two faces by id
                if (ascending) {
                    // the face after an inverted pair must clear BOTH of its faces by eps
                    if (pending_inv && !(ct - ppt >= TN_EPS) && !bad) bad = 7;
                    pending_inv = false;
                    if (is_short) { if (++run >= 2) had_special = true; } else run = 0;   // (S2)
                } else {
                    // (S1) sorted order != chain order.  Certified only for an isolated pair closer
                    // than eps whose neighbours are at least eps away on both sides.
                    if ((!is_short || !have_pp || run > 0 || pending_inv || !(ct - ppt >= TN_EPS)) && !bad) bad = 7;
                    pending_inv = true;
                    had_special = true;
                }
                do_emit = !is_short;
            } else if (!valid && have_prev && !bad) {
                bad = 10;  // hit list is not a suffix of the chain
            }
            if (valid && ++nhits > M - 1 && !bad) bad = 9;  // more than M-1 faces

            if (do_emit && !bad) {
                // combine_indices: entry slot j <- position of its vertex in the exit face's stored order
                const float r0 = 1.0f - cu - cv;
                const uint32_t c0 = (code >> 6) & 3u, c1 = (code >> 8) & 3u, c2 = (code >> 10) & 3u;
                const uint32_t k = nseg & 3u;
                mybuf[k] = cur.q0.w;  // the caller's tet id
                *reinterpret_cast<float2 *>(mybuf + 4 + 2 * k) = make_float2(pt, ct);
                float2 *bp = reinterpret_cast<float2 *>(mybuf + 12 + 6 * k);
                bp[0] = make_float2(1.0f - pu - pv, pu);
                bp[1] = make_float2(pv, sel4f(r0, cu, cv, 0.f, c0));
                bp[2] = make_float2(sel4f(r0, cu, cv, 0.f, c1), sel4f(r0, cu, cv, 0.f, c2));
                *reinterpret_cast<uint4 *>(mybuf + 36 + 4 * k) = cur.q1;  // (n, a, b, c)
                nseg++;
                need_flush = (nseg & 3u) == 0;
            }
            if (valid) {
                have_pp = have_prev; ppt = pt;
                have_prev = true; pt = ct; pu = cu; pv = cv;
            }
            if (last && !bad) {
                if (fx != f_out) bad = 11;
                // Tie handling is certified away from the chain ends only: in a short chain the reference's
                // look-ahead can pair the two hull faces through their common EMPTY tet (get_common_tetrahedra,
                // optix_trace_rays.cu:22-37); a pair inverted at the very end has no following face to clear it.
                else if ((had_special && nhits <= 8) || pending_inv) bad = 8;
            }
            if (!bad && !last && ++steps > MAX_WALK_STEPS) bad = 12;
            if (bad) { flag = true; why = bad; alive = false; need_flush = false; }
            else if (last) alive = false;
            else {
                c = nb;
                cur = nxt;
                fid_in = fx;
                A = A2; B = B2; C = C2;
                Uc = U; Vc = V; Wc = W;
            }
        }
        // ---- cooperative write-out of the full buffers (wave-uniform)
        unsigned long long fm = __ballot(need_flush);
        if (fm) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the owners' LDS writes are done
            const uint32_t grp = (uint32_t)lane / 13u, chunk = (uint32_t)lane - 13u * grp;  // lanes 52..63 idle
            while (fm) {
                // the next (up to) four rays with a full buffer
                int src = -1;
#pragma unroll
                for (uint32_t g = 0; g < 4; ++g) {
                    if (fm) {
                        const int l = __ffsll(fm) - 1;
                        fm &= fm - 1;
                        if (grp == g) src = l;
                    }
                }
                // ds_bpermute reads only ACTIVE source lanes: every lane takes part in the exchange
                const uint32_t ns = (uint32_t)__shfl((int)nseg, src >= 0 ? src : 0);
                if (src >= 0 && grp < 4) {
                    const size_t row = (wave_ray0 + (size_t)src) * M + (ns - 4);   // first of the four slots
                    const uint4 data = *reinterpret_cast<const uint4 *>(seg_lds + ((size_t)wave * 64 + src) * 52 + 4 * chunk);
                    uint32_t *dst;
                    if (chunk == 0) dst = t.out_cells + row;
                    else if (chunk < 3) dst = reinterpret_cast<uint32_t *>(t.out_dist) + 2 * row + 4 * (chunk - 1);
                    else if (chunk < 9) dst = reinterpret_cast<uint32_t *>(t.out_bary) + 6 * row + 4 * (chunk - 3);
                    else dst = t.out_verts ? t.out_verts + 4 * row + 4 * (chunk - 9) : nullptr;
                    if (dst) *reinterpret_cast<uint4 *>(dst) = data;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // buffers are read before their owners refill them
        }
        if (__ballot(alive) == 0ull) break;
    }
    // ---- left-over segments (1..3 per ray): one ray per pass, lane w copies word w of its buffer
    {
        const uint32_t left = (active && !flag) ? (nseg & 3u) : 0u;
        unsigned long long lm = __ballot(left != 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        while (lm) {
            const int src = __ffsll(lm) - 1;
            lm &= lm - 1;
            const uint32_t nl = (uint32_t)__shfl((int)left, src), ns = (uint32_t)__shfl((int)nseg, src);
            const size_t row = (wave_ray0 + (size_t)src) * M + (ns - nl);
            const uint32_t w = (uint32_t)lane;
            if (w < 52) {
                const uint32_t val = seg_lds[((size_t)wave * 64 + src) * 52 + w];
                uint32_t *dst = nullptr;
                if (w < 4) { if (w < nl) dst = t.out_cells + row + w; }
                else if (w < 12) { if (w - 4 < 2 * nl) dst = reinterpret_cast<uint32_t *>(t.out_dist) + 2 * row + (w - 4); }
                else if (w < 36) { if (w - 12 < 6 * nl) dst = reinterpret_cast<uint32_t *>(t.out_bary) + 6 * row + (w - 12); }
                else if (w - 36 < 4 * nl && t.out_verts) dst = t.out_verts + 4 * row + (w - 36);
                if (dst) *dst = val;
            }
        }
    }

    // ------------------------------------------------------------------ fallback list + tails
    if (active) {
        if (flag) {
            // ordering problems only (7 / 8: the chain itself is sound, its sorted order is not certified;
            // 10: an invalid t inside the chain): the hit list is re-collected by walking the chain again
            // (k_walk_collect) and goes through the literal sort + pairing; anything else -> BVH all-hits path
            if (p.rewalk_list && (why == 7 || why == 8 || why == 10)) {
                const uint32_t slot = atomicAdd(p.rewalk_count, 1u);
                p.rewalk_list[slot] = make_uint4((uint32_t)(p.ray_base + ray), slot_start, f_end, 0u);
            } else {
                const uint32_t slot = atomicAdd(p.fallback_count, 1u);
                p.fallback_list[slot] = (uint32_t)(p.ray_base + ray);
            }
            if (t.stats) atomicAdd(&t.stats[4 + why], 1ull);
            p.walk_n[ray] = TN_EMPTY;
        } else {
            t.out_num[ray] = nseg;
            p.walk_n[ray] = nseg;
        }
    }
    if (!p.fused_tails) return;  // a separate k_fill_tails launch (other stream) writes the tails

    // wave-cooperative constant tails of the 64 rows this wave owns
    for (int i = 0; i < 64; ++i) {
        if (wave_ray0 + i >= t.num_items) break;
        const uint32_t n_i = __shfl(nseg, i);
        const bool fl_i = __shfl((int)flag, i) != 0;
        if (fl_i) continue;  // the general kernel rewrites the whole row
        const size_t r_i = wave_ray0 + i;
        fill_dwords(t.out_cells + r_i * M, n_i, M, TN_EMPTY, lane);
        fill_dwords(reinterpret_cast<uint32_t *>(t.out_dist + r_i * M * 2), 2 * n_i, 2 * M, 0u, lane);
        fill_dwords(reinterpret_cast<uint32_t *>(t.out_bary + r_i * M * 6), 6 * n_i, 6 * M, 0u, lane);
        if (t.out_verts) fill_dwords(t.out_verts + r_i * M * 4, 4 * n_i, 4 * M, TN_EMPTY, lane);
    }
}


// Re-walk of the chains whose ORDER the walk could not certify (reasons 7 / 8 / 10): lane per ray, same
// chain, same per-face arithmetic, but every valid hit (face id, t, u, v) is only RECORDED -- into the ray's
// own output rows, used as scratch exactly like the reference's any-hit program does
// (optix_trace_rays.cu:310-326): ids -> visited row, t -> first M floats of the distance row, (u,v) -> first
// 2M floats of the barycentric row, count -> num_visited.  k_postprocess_rows then sorts and pairs them
// literally.  With two hull crossings, two crossed faces per tet and no zero edge function, the faces of the
// chain ARE the ray's all-hits set, so this equals the BVH path at a fraction of its cost; a chain that fails
// those checks here goes to the BVH list after all.
__global__ __launch_bounds__(WALK_BLOCK) void k_walk_collect(WalkParams p) {
    const TraceParams &t = p.t;
    const uint32_t M = t.M;
    const uint32_t n_items = *p.rewalk_count;
    // A lane walking ~200 dependent steps only pays off with enough lanes: below `rewalk_min` chains one
    // wavefront per ray through the BVH finishes sooner (measured: 1021 chains 0.39 ms vs 0.25 ms), so the
    // entries are handed to that list unchanged (decided on the device: no host round trip).
    const bool hand_over = n_items < p.rewalk_min;
    for (uint32_t it = blockIdx.x * WALK_BLOCK + threadIdx.x; it < n_items; it += gridDim.x * WALK_BLOCK) {
        const uint4 ent = p.rewalk_list[it];  // ray, hull triangle of the entry face, hull exit face id
        if (hand_over) {
            const uint32_t slot = atomicAdd(p.fallback_count, 1u);
            p.fallback_list[slot] = ent.x;
            t.out_num[ent.x] = TN_EMPTY;
            if (t.stats) atomicAdd(&t.stats[4 + 14], 1ull);
            continue;
        }
        const size_t ray = ent.x;
        const uint32_t f_out = ent.z;
        const RayPre rp = ray_pre(t.origins[3 * ray], t.origins[3 * ray + 1], t.origins[3 * ray + 2], t.dirs[3 * ray],
                                  t.dirs[3 * ray + 1], t.dirs[3 * ray + 2]);
        uint32_t *row_id = t.out_cells + ray * M;
        float *row_t = t.out_dist + ray * M * 2;
        float *row_uv = t.out_bary + ray * M * 6;
        uint32_t nhits = 0, steps = 0, bad = 0;
        auto record = [&](uint32_t fid, float tt, float uu, float vv) {
            if (nhits < M - 1) {
                row_id[nhits] = fid;
                row_t[nhits] = tt;
                *reinterpret_cast<float2 *>(row_uv + 2 * nhits) = make_float2(uu, vv);
            }
            nhits++;
        };
        // the entry hull face in its stored order (as in k_trace_walk)
        const float4 *tp = p.hull_tris + 3 * (size_t)ent.y;
        const float4 v0 = tp[0], v1 = tp[1], v2 = tp[2];
        SV A = shear(rp, v0.x, v0.y, v0.z), B = shear(rp, v1.x, v1.y, v1.z), C = shear(rp, v2.x, v2.y, v2.z);
        float Uc = edge_f(B, C), Vc = edge_f(C, A), Wc = edge_f(A, B);
        uint32_t c = 4u * __float_as_uint(v1.w) + __float_as_uint(v2.w);
        {
            float tt, uu, vv;
            if (tri_finish(Uc, Vc, Wc, A.z, B.z, C.z, tt, uu, vv)) record(__float_as_uint(v0.w), tt, uu, vv);
        }
        Var cur = load_var(p.vars, c);
        for (;;) {
            const SV P = shear(rp, __uint_as_float(cur.q0.x), __uint_as_float(cur.q0.y), __uint_as_float(cur.q0.z));
            const float ea = edge_f(P, A), eb = edge_f(P, B), ec = edge_f(P, C);
            if (ea == 0.0f || eb == 0.0f || ec == 0.0f) bad = 5;
            const bool sa = ea > 0.0f, sb = eb > 0.0f, sc = ec > 0.0f;
            const bool su = Uc > 0.0f, sv = Vc > 0.0f, sw = Wc > 0.0f;
            const bool ha = (sb == su) && (su != sc), hb = (sc == sv) && (sv != sa), hc = (sa == sw) && (sw != sb);
            const uint32_t hmask = (ha ? 1u : 0u) | (hb ? 2u : 0u) | (hc ? 4u : 0u);
            if (!bad && __popc(hmask) != 1) bad = 6;
            const uint32_t x = (__ffs(hmask) - 1) & 3u;
            const uint32_t nb = sel3u(cur.q2, x);
            const uint32_t fx = sel3u(cur.q3, x);
            const bool last = nb == TN_EMPTY;
            const Var nxt = load_var(p.vars, (last || bad) ? c : nb);  // requested before this step's stores
            __builtin_amdgcn_sched_barrier(0);
            if (bad) break;
            const bool x0 = (x & 1u) != 0, x1 = (x & 2u) != 0;
            const uint32_t w01 = x0 ? (cur.q3.w >> 12) : cur.q3.w;
            const uint32_t w2 = (cur.q3.w >> 24) | (cur.q2.w << 8);
            const uint32_t code = (x1 ? w2 : w01) & 0xFFFu;
            const SV A2 = selsv(P, A, B, C, code & 3u), B2 = selsv(P, A, B, C, (code >> 2) & 3u), C2 = selsv(P, A, B, C, (code >> 4) & 3u);
            const float U = edge_f(B2, C2), V = edge_f(C2, A2), W = edge_f(A2, B2);
            float tt, uu, vv;
            if (tri_finish(U, V, W, A2.z, B2.z, C2.z, tt, uu, vv)) record(fx, tt, uu, vv);
            if (last) {
                if (fx != f_out) bad = 11;
                break;
            }
            if (++steps > MAX_WALK_STEPS) { bad = 12; break; }
            c = nb;
            cur = nxt;
            A = A2; B = B2; C = C2;
            Uc = U; Vc = V; Wc = W;
        }
        if (!bad && nhits > M - 1) bad = 9;  // overflow: the BVH path keeps the M-1 nearest
        if (bad) {
            const uint32_t slot = atomicAdd(p.fallback_count, 1u);
            p.fallback_list[slot] = (uint32_t)ray;
            t.out_num[ray] = TN_EMPTY;  // k_postprocess_rows skips it; the BVH kernel rewrites the row
            if (t.stats) atomicAdd(&t.stats[4 + 14], 1ull);
        } else {
            t.out_num[ray] = nhits;
            if (t.stats) atomicAdd(&t.stats[4 + 13], 1ull);
        }
    }
}

void launch_walk_collect(const WalkParams &p, size_t max_items, hipStream_t stream) {
    if (max_items == 0) return;
    size_t blocks = (max_items + WALK_BLOCK - 1) / WALK_BLOCK;
    if (blocks > 256 * 4) blocks = 256 * 4;
    hipLaunchKernelGGL(k_walk_collect, dim3((unsigned)blocks), dim3(WALK_BLOCK), 0, stream, p);
}

// Constant tails of the rows the walk certified: slots [n, M) of the four row arrays.  Pure
// streaming stores (16 B per lane), one wave per ray per pass, XCD-banded like the walk so a
// row's lines are written by the XCD whose L2 already holds the row's segment lines.
__global__ __launch_bounds__(256) void k_fill_tails(size_t num_rays, uint32_t M, const uint32_t *__restrict__ walk_n,
                                                    uint32_t *__restrict__ out_cells, float *__restrict__ out_bary,
                                                    float *__restrict__ out_dist, uint32_t *__restrict__ out_verts) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t nwaves = (size_t)gridDim.x * 4;
    const uint32_t per = gridDim.x >> 3;  // gridDim.x is a multiple of 8
    const size_t lb = (size_t)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
    // contiguous span of rays per wave: consecutive rows are consecutive in memory
    const size_t span = (num_rays + nwaves - 1) / nwaves;
    const size_t r0 = (lb * 4 + wave) * span;
    const size_t r1 = r0 + span < num_rays ? r0 + span : num_rays;
    for (size_t r = r0; r < r1; ++r) {
        const uint32_t n = walk_n[r];
        if (n == TN_EMPTY) continue;  // re-traced by the general kernel, which writes the whole row
        fill_dwords(out_cells + r * M, n, M, TN_EMPTY, lane);
        fill_dwords(reinterpret_cast<uint32_t *>(out_dist + r * M * 2), 2 * n, 2 * M, 0u, lane);
        fill_dwords(reinterpret_cast<uint32_t *>(out_bary + r * M * 6), 6 * n, 6 * M, 0u, lane);
        if (out_verts) fill_dwords(out_verts + r * M * 4, 4 * n, 4 * M, TN_EMPTY, lane);
    }
}

void launch_fill_tails(size_t num_rays, uint32_t M, const uint32_t *walk_n, uint32_t *out_cells, float *out_bary,
                       float *out_dist, uint32_t *out_verts, hipStream_t stream, unsigned max_blocks) {
    if (num_rays == 0) return;
    size_t blocks = (num_rays + 3) / 4;           // >= one ray per wave
    // default: 2 blocks (8 waves) per CU -- enough to hold the write ceiling, and measured 3 % faster per launch
    // than 8 per CU because the BVH re-trace running beside the fill is less starved (profiles/r01_fill_grid.txt)
    const size_t cap = max_blocks ? max_blocks : 256 * 2;
    if (blocks > cap) blocks = cap;
    blocks = (blocks + 7) & ~(size_t)7;
    hipLaunchKernelGGL(k_fill_tails, dim3((unsigned)blocks), dim3(256), 0, stream, num_rays, M, walk_n, out_cells, out_bary,
                       out_dist, out_verts);
}

void launch_trace_walk(const WalkParams &p, hipStream_t stream) {
    if (p.t.num_items == 0) return;
    const uint32_t nblk = (uint32_t)((p.t.num_items + WALK_BLOCK - 1) / WALK_BLOCK);
    // grid padded so that both remaps (runs of XCD_GROUP blocks / one band per XCD) are bijections
    const uint32_t unit = 8 * ((p.debug & 8u) ? (nblk + 7) / 8 : XCD_GROUP);
    const uint32_t grid = (nblk + unit - 1) / unit * unit;
    hipLaunchKernelGGL(k_trace_walk, dim3(grid), dim3(WALK_BLOCK), WALK_BLOCK * 52 * sizeof(uint32_t), stream, p);
}

}  // namespace tn
