// tn_trace_walk.hip -- adjacency-walk trace path: ONE LANE PER RAY.
//
// A Delaunay tetrahedralisation has a convex hull, so a ray's faces form one chain
// f0 < f1 < ... < fn in which consecutive faces bound the tetrahedron between them.  Instead
// of collecting all hits through a BVH and sorting them (the reference's structure,
// src/optix/optix_trace_rays.cu:268-331 + :78-108), a lane walks the chain: find the two hull
// faces the ray's line crosses (k_hull_entry), then step tet -> neighbour tet through 64-byte records specialised
// by entry face (WalkVar, tn_common.h), producing the faces already in order.  Per step: one
// dependent record load, ONE vertex shear, three edge functions against the carried entry face to
// pick the exit, the exit face's three edge functions in its stored order, one (t,u,v).
//
// Parity by construction: every (t,u,v) is computed by the same expression tree, in the face's
// STORED vertex order, as the general path / the oracle (tri_finish in tn_device.h).  The walk
// sorts its rays into three classes:
//   certified   exactly two hull faces are crossed, every tet on the way has exactly two crossed
//               faces, no edge function is exactly 0, no vertex of a visited tet within rounding
//               distance of the ray, every recorded t in (0, 1e16), fewer than M faces, AND the order
//               is "clean": the hits ascend in the total order (t, face id), with gaps of ANY size -- or
//               contain ISOLATED pairs that are inverted / tied the wrong way round by less than eps whose
//               neighbours are at least eps away from both members.  For such a list the reference's
//               dedupe / pairing phases (optix_trace_rays.cu:124-257) provably reduce to "pair face k-1
//               with face k, drop the pairs shorter than eps":
//               * ascending list (sorted order == chain order).  Only chain neighbours share a tet (every
//                 tet of a sound chain has exactly two crossed faces), except the two hull faces, which
//                 "share" EMPTY.  Phase 1 at slot j looks at the following slots within eps of t_j: of
//                 those only the chain successor j+1 shares a tet with j, so j marks j+1 iff gap(j, j+1) <
//                 eps, and j is cleared iff it was marked (gap(j-1, j) < eps) and marks (gap(j, j+1) < eps):
//                 exactly the INTERIOR faces of every run of short gaps are cleared, the two ends of a run
//                 survive.  Phase 2 at a surviving j: if slot j+1 survives it is the chain successor and
//                 the pair is emitted iff its gap is >= eps; if slot j+1 was cleared (j opens a run of >= 2
//                 short gaps) the look-ahead skips the cleared slots and examines the next two surviving
//                 faces (and on while they are eps-chained), none of which shares a tet with j -- no
//                 emission, no swap.  Every pair with a gap >= eps has two surviving ends in adjacent
//                 slots, so the emitted set is { (k-1, k) : gap >= eps }, each from its own two slots.
//                 The one exception is the EMPTY == EMPTY quirk of get_common_tetrahedra: a look-ahead
//                 from the ENTRY hull face that reaches the exit hull face pairs the two.  The entry face
//                 only looks ahead when slot 1 was cleared, i.e. when the first TWO gaps are short: such
//                 rays are not certified (the lattice meshes of tests/test_parity_configs_gpu.py hit this).
//               * an isolated inverted pair: phase 1 only marks, phase 2 finds the partner of the face before
//                 the pair in the second slot it examines and the swap restores chain order.
//               * round 6, the ENDS of the chain (the hull is where slivers are: profiles/r06b_literal_reasons.txt -- 73-91 % of
//                 the literal rays of the bench meshes had their first violation there).  Stated in tests/cert_model.py, checked
//                 against the literal algorithm on crafted chains in tests/test_certification_rules.py:
//                 A  an isolated inverted pair at the very END of a chain of >= 4 hits: the face before the pair finds its
//                    partner in the second slot it examines and the swap restores chain order, nothing follows that could need
//                    clearing.  (3 hits: the entry hull face would examine the exit hull face first -> EMPTY == EMPTY.)
//                 B  a run of >= 2 short ascending gaps AT THE ENTRY face: phase 1 clears the run's interior; the entry face's
//                    look-ahead examines the run's last face and the face after it -- neither shares a tet with it -- and stops
//                    there iff the gap behind that face is long, which is required (two long ascending gaps after the run).
//                 C  the FIRST pair inverted by less than eps: after the sort hit 1 pairs with hit 0 (short: no segment) and hit 0
//                    then finds no partner, so the reference emits NO segment for hit 2 although it is long: the ray is
//                    certified with a flag that makes the segment writer drop that one segment (hit 2 clear of both by eps,
//                    then two more long ascending gaps: the same look-ahead bound).
//                 D  (generalises the isolated inverted pair) a CLUSTER -- hits each less than eps above the largest t before
//                    them -- that contains inversions is certified when it has at most THREE members, no member lies eps or
//                    more below an earlier one and every member is at least eps above the previous cluster: for every sorted
//                    order of such a triple the phases still emit exactly the chain pairs with a gap >= eps (exhaustive over the
//                    orders + 350,000 random chains against the literal algorithm; the first counter-examples are 4-clusters in
//                    which the first member sorts behind both of the last two -- the a-p-u-q-v-r-b example below is of that kind).
//                 The device code is struct OrderR6 below (clusters); OrderR5 keeps round 5's pairwise statement for the
//                 cross-check option cert_ends = 0.
//               Larger clusters that contain an inversion are NOT certified: a face whose two chain neighbours both sort
//               after it survives phase 1 inside the run, and the look-ahead of the run's last face can then
//               stop short of its partner (worked example in DESIGN.md section 2).
//   literal     the chain is sound but its order is not clean (an inversion inside a run of short gaps, an
//               inversion by eps or more, an end-of-chain pattern whose look-ahead bound is not met): the hits of
//               the log go through the literal sort + pairing (k_postprocess_log, tn_trace_general.hip).
//   fallback    anything else -> re-traced by the BVH all-hits kernel.
// The chain is the connected component of the hull faces in the set of crossed faces.  Every tet has 0 or 2 crossed
// faces for ANY rounded 2-D vertex positions (its boundary is a closed surface; edge functions are shared, so faces
// agree on every edge), hence the crossed faces form the chain plus, possibly, closed CYCLES the walk cannot see.  The
// rounded positions are the exact projections of a mesh whose vertices moved by at most delta = 7 * 2^-24 * (|o| +
// scene) perpendicular to the ray; ray casting through a properly embedded mesh yields one chain and no cycle, so a
// cycle needs a tet that the perturbation can INVERT: one whose smallest height is of the order of delta (a tet of
// height h loses at most a fraction 4 delta / h of its volume).  Round 3's aimed fuzzer (profiles/r03_hole_fuzz.py)
// produced such cycles -- 1.4e-5 of the rays aimed at edges and faces of meshes with vertex twins 1e-8..1e-6 apart and
// of a lattice jittered by 1e-7; never on a well-shaped mesh -- and every one of them passed within 0.13 delta of a mesh
// edge while all its vertices were far away (profiles/r03e_hole_analyse.txt): the needle faces between two nearly
// coincident long edges flip under the rounding, and the tets they bound are NOT on the chain, they only share the
// near edge with it.  Rule 8 therefore combines a mesh-side and a ray-side condition: the record of every tet
// carries (as an exponent byte) the second smallest of its vertices' star minima of the tet height -- a lower bound
// of the thinnest tet around its most suspicious edge (tn_build_core.h) -- and a ray leaves the walk when it passes
// within 8 delta of an edge of a tet whose value is below 32 delta.  Rule 4 (a vertex within the box padding = 32/7
// delta of the ray, L1 norm in the sheared plane) stays.  The count-only BVH cross-check (option verify_stride) and the
// fuzzer are the evidence that nothing is left: see DESIGN.md section 2.
//
// Memory behaviour: the walk does NOT touch the output rows.  Rows are 26 KB apart, so anything a lane stores
// into its own row is a scattered partial-line write (round 1: 1.52x write amplification, the stores were the
// larger half of the kernel).  Instead every recorded hit goes to a HIT LOG as one 16-byte entry
// {t, u, v, variant | exit << 30} at log[wave of 64 rays][hit index][lane] -- the 64 lanes of a wave store 1 KB of
// consecutive bytes per step, and the log is 16 B per hit instead of 52 B per segment.  k_write_segments turns
// the log of the certified rays into segment records (whole 128-byte lines, one wave per 8 rays, staged through LDS so
// that every store instruction writes contiguous runs), k_fill_rows_fine -- one short-lived block per row -- streams the
// constant tails [ceil32(n), M) of the certified rows after the segment writer (rounds 2-5 filled the last quarter of every
// row BESIDE the walk with persistent waves: option spec_fill = 1 / k_fill_range; since round 6's faster writer that overlap
// no longer pays).  The entry search is a kernel of its own (k_hull_entry: the hull's boxes and faces in LDS).
// blockIdx is remapped so each XCD owns runs of 16 consecutive blocks (4096 neighbouring rays) and its L2 keeps
// the tets they cross.
#include "tn_device.h"
#include "tn_kernels.h"

// Diagnostic builds only (make CXXFLAGS+=-DTN_WALK_DIAG=1; profiles/r06b_literal_reasons.py): which rule of the order test made
// a ray "literal" -- the FIRST violated one per ray, counted in g_walk_diag[1..8]; [0] = literal rays, [9..15] see below.
#ifndef TN_WALK_DIAG
#define TN_WALK_DIAG 0
#endif
#if TN_WALK_DIAG
__device__ unsigned long long g_walk_diag[16];
__device__ unsigned long long g_walk_time[8];   // 100 MHz ticks: [0] waves, [1] sum hull search, [2] sum walk loop, [3] max hull, [4] max loop, [5] max steps, [6] sum of per-wave max steps
extern "C" int tn_debug_walk_time(unsigned long long out[8], int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_walk_time), sizeof(unsigned long long) * 8) != hipSuccess) return 1;
    if (reset) { unsigned long long z[8] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_walk_time), z, sizeof z) != hipSuccess) return 1; }
    return 0;
}
extern "C" int tn_debug_walk_diag(unsigned long long out[16], int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_walk_diag), sizeof(unsigned long long) * 16) != hipSuccess) return 1;
    if (reset) { unsigned long long z[16] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_walk_diag), z, sizeof z) != hipSuccess) return 1; }
    return 0;
}
#endif

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

template <bool NT = false>
__device__ __forceinline__ void fill_dwords(uint32_t *__restrict__ base, uint32_t start, uint32_t end, uint32_t value, int lane) {
    const uint32_t a0 = (start + 3u) & ~3u;  // first 16-B aligned dword
    const uint32_t head_end = a0 < end ? a0 : end;
    if (start + lane < head_end) base[start + lane] = value;
    if (a0 >= end) return;
    const uint32_t a1 = end & ~3u;
    u32x4 *b4 = reinterpret_cast<u32x4 *>(base);
    const u32x4 v4 = {value, value, value, value};
    for (uint32_t i = (a0 >> 2) + lane; i < (a1 >> 2); i += 64) {
        // plain stores: nontemporal ones measured slower for a pure write stream running alone; NT = the fill that
        // streams BESIDE the walk (the walk's records then stay in the XCD's L2)
        if constexpr (NT) __builtin_nontemporal_store(v4, b4 + i);
        else b4[i] = v4;
    }
    if (a1 + lane < end) base[a1 + lane] = value;
}

__device__ __forceinline__ float sel4f(float a, float b, float c, float d, uint32_t i) {
    const bool b0 = (i & 1u) != 0, b1 = (i & 2u) != 0;
    const float lo = b0 ? b : a, hi = b0 ? d : c;
    return b1 ? hi : lo;
}

// The walk's part of a record (WalkHot: 32 bytes; tet id, vertex ids and face ids live in the tables of their own
// consumers), as SCALARS: with the neighbours / face ids kept in a uint4 the optimiser turns the select
// chain below into a dynamically indexed vector, parks the record in LDS and reads `nb` back with a ds_read -- an
// LDS round trip on the one dependent chain of the walk (record -> exit -> next record).
struct Var { float px, py, pz; uint32_t nb0, nb1, nb2, code_lo, code_hi; };
__device__ __forceinline__ Var load_var(const WalkHot *vars, uint32_t c) {   // two 16-byte loads of one 32-byte record
    const uint32_t *r = reinterpret_cast<const uint32_t *>(vars + c);
    const float4 q0 = *reinterpret_cast<const float4 *>(r);
    const uint4 q1 = *reinterpret_cast<const uint4 *>(r + 4);
    Var v;
    v.px = q0.x; v.py = q0.y; v.pz = q0.z; v.nb0 = __float_as_uint(q0.w);
    v.nb1 = q1.x; v.nb2 = q1.y; v.code_lo = q1.z; v.code_hi = q1.w;
    return v;
}
__device__ __forceinline__ uint32_t sel4u(uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t i) {
    const bool b0 = (i & 1u) != 0, b1 = (i & 2u) != 0;
    const uint32_t lo = b0 ? b : a, hi = b0 ? d : c;
    return b1 ? hi : lo;
}
__device__ __forceinline__ uint32_t sel3u(uint32_t v0, uint32_t v1, uint32_t v2, uint32_t i) {  // i in 0..2 (3 -> v2)
    const bool b0 = (i & 1u) != 0, b1 = (i & 2u) != 0;
    const uint32_t lo = b0 ? v1 : v0;
    return b1 ? v2 : lo;
}
__device__ __forceinline__ SV selsv(const SV &p0, const SV &p1, const SV &p2, const SV &p3, uint32_t i) {
    SV r;
    r.x = sel4f(p0.x, p1.x, p2.x, p3.x, i); r.y = sel4f(p0.y, p1.y, p2.y, p3.y, i); r.z = sel4f(p0.z, p1.z, p2.z, p3.z, i);
    return r;
}

}  // namespace

// ---- the ORDER TEST of the certification (header: "certified"), as a per-lane state machine over the recorded hits ----------
// Two statements of it exist on purpose.  OrderR6 is the product (round 6: clusters; rules A-D); OrderR5 is round 5's pairwise
// test (option cert_ends = 0): rays that R6 certifies and R5 does not come once from the segment writer and once from the literal
// pairing kernel -- two independent implementations -- and must give identical rows (tests/test_walk_gpu.py).  Both are written as
// selects: `vp` = this step recorded a hit and a previous one exists, `start` = this step recorded the ray's first hit.
namespace {
// OrderPair<false> = round 5's pairwise test; OrderPair<true> = the same with the end-of-chain rules A-C added (the form round 6
// measured first: cheaper per step than the cluster test, and what the tracer uses below WALK_TET_MIN_TETS tets;
// tests/cert_model.py: certify_pairwise).
template <bool ENDS>
struct OrderPair {
    bool ok = true, have_pp = false, prev_short = false, prev_inv = false, d2 = false;
    uint32_t pend = 0;
    float ppt = 0.f;
    __device__ __forceinline__ uint32_t step(bool valid, bool have_prev, float pt, float ct, bool tie_asc, uint32_t nhits) {
        const bool vp = valid && have_prev;
        const bool is_short = fabsf(pt - ct) < TN_EPS;
        const bool asc = (ct > pt) || (ct == pt && tie_asc);
        const bool clear2 = ct - ppt >= TN_EPS;
        // short + ascending: any run of them is fine (header) -- a run that starts at the entry face (pairs 1 and 2 both short:
        // nhits == 2 here) only if two long gaps follow it (rule B); short + inverted: isolated and clear of the face before -- as
        // the FIRST pair: the segment of hit 2 is lost, three clear long gaps must follow (rule C); after an inverted pair: a long
        // gap, clear of both of its members
        const bool pend_wait = pend >= 2u;
        const bool first_inv = is_short && !asc && !have_pp;
        const bool entry_run = is_short && asc && prev_short && nhits == 2u;
        const bool good = (is_short ? (!pend_wait && (asc ? !prev_inv : (!have_pp || (!prev_short && clear2))))
                                    : (asc && (!prev_inv || clear2))) &&
                          (ENDS || !(entry_run || first_inv));
        const uint32_t pend_long = (0x2810u >> (3u * pend)) & 7u;               // on a long gap: 0 2 0 4 2
        const uint32_t pend_next = is_short ? (entry_run ? 1u : (first_inv ? 3u : pend)) : pend_long;
        pend = vp ? pend_next : pend;
        d2 = d2 || (vp && first_inv);
        ok = ok && (!vp || good);
        prev_inv = vp ? (is_short && !asc) : prev_inv;
        prev_short = vp ? is_short : prev_short;
        have_pp = valid ? have_prev : have_pp;
        ppt = valid ? pt : ppt;
        return (vp && !good) ? 1u : 0u;
    }
    __device__ __forceinline__ bool needs_step(bool, bool, float) const { return true; }
    __device__ __forceinline__ void fast(bool, float) {}
    // rule A: an isolated inverted pair at the very end of a chain of >= 4 hits needs no following face; B / C: look-ahead settled
    __device__ __forceinline__ bool finish(uint32_t nhits) const { return ok && pend == 0u && !(prev_inv && (nhits < 4u || !ENDS)); }
    __device__ __forceinline__ bool drop2() const { return d2; }
    __device__ __forceinline__ uint32_t end_reason() const { return 8u; }
};
using OrderR5 = OrderPair<false>;
using OrderR5e = OrderPair<true>;

// Round 6: the same test on CLUSTERS.  A hit joins the current cluster iff it is less than eps above the cluster's largest t
// (so a new cluster starts with a gap of at least eps above EVERY member of the old one).  tests/cert_model.py is this state
// machine in Python, statement for statement; tests/test_certification_rules.py checks it against the literal algorithm.
//   * a cluster WITHOUT an inversion is an ascending run of short gaps: any length (header proof);
//   * a cluster WITH an inversion (adjacent hits in the wrong sorted order, exact ties by face id) is certified iff it has at most
//     THREE members, no member lies eps or more below an earlier one, and every member is at least eps above the previous
//     cluster (rule D; 2 members = round 2's isolated inverted pair).  Exhaustive over the sorted orders of isolated 3-clusters
//     and Monte-Carlo over 350,000 chains: the reference's phases then still emit exactly the chain pairs with a gap >= eps; the
//     first failures are 4-clusters in which the FIRST member sorts behind both of the last two;
//   * the ENDS of the chain: A an inverted PAIR as the last cluster of a chain of >= 4 hits; B an ascending first cluster of >= 3
//     hits when two long gaps follow; C the first cluster = an inverted pair: the reference loses the segment of hit 2 (drop2),
//     three long gaps must follow.  `pend` is the entry face's look-ahead: 0 none | 1 inside B's run | 2 one more long gap needed |
//     3 C: hit 2 pending | 4 two more long gaps needed; a hit that JOINS a cluster while pend >= 2 ends the certification.
struct OrderR6 {
    bool ok = true, cinv = false, cfirst = false, d2 = false;
    uint32_t cn = 0, pend = 0;
    float cmax = 0.f, prev_cmax = 0.f;
    __device__ __forceinline__ uint32_t step(bool valid, bool have_prev, float pt, float ct, bool tie_asc, uint32_t) {
        const bool vp = valid && have_prev, start = valid && !have_prev;
        const bool joins = !(ct - cmax >= TN_EPS);
        const bool asc = (ct > pt) || (ct == pt && tie_asc);
        const bool vj = vp && joins, vn = vp && !joins;
        const bool long_inv = cmax - ct >= TN_EPS;
        const bool cinv_n = cinv || !asc;
        const uint32_t cn_n = cn + 1u;
        const bool inv_ok = !long_inv && (cfirst ? cn_n == 2u : (cn_n <= 3u && ct - prev_cmax >= TN_EPS));
        const bool good = pend < 2u && (!cinv_n || inv_ok);
        ok = ok && (!vj || good);
        const bool rule_c = vj && cinv_n && cfirst && cn_n == 2u;
        const bool rule_b = vj && !cinv_n && cfirst && cn_n == 3u;
        d2 = d2 || rule_c;
        const uint32_t pend_new_cluster = (0x2810u >> (3u * pend)) & 7u;        // 0 2 0 4 2
        const uint32_t pend_old = pend;
        pend = vj ? (rule_c ? 3u : (rule_b ? 1u : pend)) : (vn ? pend_new_cluster : pend);
        const uint32_t why = (vj && !good) ? (pend_old >= 2u ? 1u : (long_inv ? 2u : (cfirst ? 5u : (cn_n > 3u ? 3u : 4u)))) : 0u;
        const bool fresh = vn || start;
        prev_cmax = vn ? cmax : prev_cmax;
        cmax = vj ? fmaxf(cmax, ct) : (fresh ? ct : cmax);
        cn = vj ? cn_n : (fresh ? 1u : cn);
        cinv = vj ? cinv_n : (fresh ? false : cinv);
        cfirst = start ? true : (vn ? false : cfirst);
        // (`why`, diagnostic builds: 1 joined while the entry look-ahead was pending | 2 eps or more below an earlier member | 3
        //  inverted cluster of more than three | 4 inverted cluster within eps of the previous one | 5 inverted first cluster)
        return why;
    }
    // 98 % of a wave's steps are "every lane's hit starts a new cluster, no look-ahead pending": then step() reduces to fast().
    // The kernel asks needs_step() per lane and takes the full statement only when ANY lane of the wave needs it (a wave-uniform
    // branch on a ballot): 277 -> ~245 VALU instructions per step, below round 5's pairwise test (250).
    __device__ __forceinline__ bool needs_step(bool valid, bool have_prev, float ct) const {
        return valid && (!have_prev || pend != 0u || !(ct - cmax >= TN_EPS));
    }
    __device__ __forceinline__ void fast(bool valid, float ct) {   // valid => a previous hit exists, the hit does not join, pend == 0
        prev_cmax = valid ? cmax : prev_cmax;
        cmax = valid ? ct : cmax;
        cn = valid ? 1u : cn;
        cinv = cinv && !valid;
        cfirst = cfirst && !valid;
    }
    __device__ __forceinline__ bool finish(uint32_t nhits) const {
        return ok && pend == 0u && (!cinv || (!cfirst && cn == 2u && nhits >= 4u));
    }
    __device__ __forceinline__ bool drop2() const { return d2; }
    __device__ __forceinline__ uint32_t end_reason() const { return pend ? 6u : 7u; }   // look-ahead unsettled | inverted last cluster
};
}  // namespace

// The walk runs on entry-face-specialised records (WalkVar, tn_common.h): nothing of the entry face is permuted
// or recomputed, one vertex is sheared per step, three edge functions against it decide the exit, and the exit
// face's edge functions are evaluated directly in its stored order (E(P,Q) == -E(Q,P) bitwise, so they equal the
// shared ones): bit-identical hits for 30 % fewer instructions than a per-tet record with dynamic selects.
// Entry search, a kernel of its own since round 6 (it was the first half of k_trace_walk: its registers -- six float4 of
// face data in flight -- cost the walk loop an occupancy step, 82 instead of 69 VGPRs).  Per ray one 16-byte HullEntry:
//   x = variant (record, entry face) the walk starts in, y = face id of the entry hull face,
//   z = triangle slot of that face | first flag reason << 24 | hull near-miss risk << 28 | state << 29 (0 miss, 1 walk, 2 hand over),
//   w = t of the OTHER crossed hull face (the chain must end there, bit for bit).
__global__ __launch_bounds__(WALK_BLOCK) void k_hull_entry(WalkParams p) {
    const TraceParams &t = p.t;

    // XCD-aware block remap: hardware places block b on XCD b % 8; give each XCD a contiguous band
    // Consecutive blocks trace neighbouring rays that cross the same tets, so each XCD gets RUNS
    // of XCD_GROUP consecutive blocks (its L2 keeps their tets) while the runs still interleave
    // across the frame (an XCD owning one contiguous band would own all the misses or all the
    // long rays).
    const uint32_t nblk = (uint32_t)((t.num_items + WALK_BLOCK - 1) / WALK_BLOCK);
    const uint32_t super = blockIdx.x / (8 * XCD_GROUP), rem = blockIdx.x % (8 * XCD_GROUP);
    const uint32_t lb = super * 8 * XCD_GROUP + (rem & 7) * XCD_GROUP + (rem >> 3);
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
    // rounding distance of a projected vertex: the box padding of the BVH path (tn_device.h: line_box)
    const float pad = 16.0f * 1.1920929e-7f * (fmaxf(fabsf(ox), fmaxf(fabsf(oy), fabsf(oz))) + p.scene_max);
    // Rule 8 (fold guard, see the header): delta = 7 * 2^-24 * (|o| + scene) bounds the error of a sheared 2-D vertex
    // position; a tet whose neighbourhood holds a tet thinner than 32 delta (exponent compare, conservative) AND one
    // of whose edges passes within 8 delta of the ray sends the ray to the BVH path.
    const float delta = 0.21875f * pad;               // 7/32 of the box padding
    const float kappa = 8.0f * delta;
    // RISK classes of the certification (round 5): the two guards above hand a ray over when it passes
    // within 8 delta of a hull edge / of an edge of a thin-neighbourhood tet.  What is not proved is that the degenerate
    // feature always lies THAT close to a tested edge, so the same tests with a wider band (option "risk_band": twice as wide
    // = 16 delta by default) mark a certified ray as "at risk": every such ray -- not one in 256 -- is re-counted by the BVH
    // cross-check (k_verify_counts).  bit 0: within 8 delta (hand over), bit 1: within the wide band (verify)
    const float kappa2 = p.risk_band * kappa;
    auto edge_band = [&](float e, const SV &X, const SV &Y) -> uint32_t {
        const float len = fabsf(X.x - Y.x) + fabsf(X.y - Y.y), ae = fabsf(e);
        return (ae <= kappa * len ? 1u : 0u) | (ae <= kappa2 * len ? 2u : 0u);
    };
    uint32_t risk = 0;   // bit 0: hull near-miss, bit 1: thin-neighbourhood near-miss
    uint32_t nhull = 0;
    uint32_t hf0 = TN_EMPTY, hf1 = TN_EMPTY, hc0 = 0, hc1 = 0, he0 = 0, he1 = 0, hs0 = 0, hs1 = 0;
    float ht0 = 0.f, ht1 = 0.f;
    auto hull_face = [&](const SV &A, const SV &B, const SV &C, uint32_t fid, uint32_t rec, uint32_t loc, uint32_t slot) {
        const float U = edge_f(B, C), V = edge_f(C, A), W = edge_f(A, B);
        const bool mixed = (U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f);
        // crossed (or degenerate: a zero edge function -> the general path decides).  Written as selects: with
        // conditional assignments the captured results end up in scratch memory behind computed pointers.
        const float det = (U + V) + W;
        if (!mixed && (U == 0.0f || V == 0.0f || W == 0.0f || det == 0.0f)) { flag = true; why = 1; }
        // hull graze guard: the line passes within 8 delta of an edge of a hull face near it (crossed or not).  A ray
        // that grazes the hull can have crossed faces in the rounded projection -- a closed cycle around the grazed
        // edge / vertex -- although it crosses no hull face at all (nhull == 0: the walk would certify a miss); round 3's
        // fuzzer found exactly these after rule 8 had removed the interior cycles (profiles/r03g_hole_classify.txt)
        const uint32_t hb = edge_band(U, B, C) | edge_band(V, C, A) | edge_band(W, A, B);
        if (hb & 1u) { flag = true; why = why ? why : 2u; }
        risk |= (hb >> 1) & 1u;
        const float T = (U * A.z + V * B.z) + W * C.z;
        const float tt = T / det;
        const bool s0 = !mixed && nhull == 0, s1 = !mixed && nhull == 1;
        hf0 = s0 ? fid : hf0; ht0 = s0 ? tt : ht0; hc0 = s0 ? rec : hc0; he0 = s0 ? loc : he0; hs0 = s0 ? slot : hs0;
        hf1 = s1 ? fid : hf1; ht1 = s1 ? tt : ht1; hc1 = s1 ? rec : hc1; he1 = s1 ? loc : he1; hs1 = s1 ? slot : hs1;
        nhull += mixed ? 0u : 1u;
    };
    // Round 6: the search was HALF of this kernel (100 MHz clock around both parts, profiles/r06y_walk_time.txt: 141 us of
    // hull search and 95 us of walk per wave on the C2 frame, 1.1 ms and 0.6 ms at 1M tets) although a hull has a few hundred
    // faces: every node and every leaf triangle of the threaded tree is a dependent round trip to L2, ~90 of them per ray.
    // For hulls of at most HULL_FLAT_MAX faces the boxes now live in LDS (flat two-level table, tn_common.h): a lane tests
    // all G <= 64 group boxes, the 8 leaf boxes of each group it hits and the 2 faces of each leaf it hits -- faces staged in
    // LDS as well (66 KB at 1024 faces): no dependent round trip to memory is left in the search.  The order in which
    // crossings are found is free (first0 = ht0 < ht1 sorts the two).
    const float ix = safe_inv(dx), iy = safe_inv(dy), iz = safe_inv(dz);
    if (p.n_hull_leaves) {
        extern __shared__ float4 s_hull[];
        const uint32_t nbox2 = 2u * (p.n_hull_groups + p.n_hull_leaves);
        for (uint32_t i = threadIdx.x; i < nbox2; i += WALK_BLOCK) s_hull[i] = p.hull_flat[i];
        float4 *s_tri = s_hull + nbox2;                         // the faces themselves: 48 bytes each
        for (uint32_t i = threadIdx.x; i < 3u * p.n_hull; i += WALK_BLOCK) s_tri[i] = p.hull_tris[i];
        __syncthreads();
        const float4 *s_leaf = s_hull + 2u * p.n_hull_groups;
        // boxes padded by TWICE the rounding distance: the graze guard of hull_face reaches 8 delta = 1.75 pad from an edge, and
        // every face that close to the line must be looked at (the 2-face leaves are tighter than the tree's 4-face ones)
        const float pad2 = 2.0f * pad;
        unsigned long long gm = 0ull;
        if (active)
            for (uint32_t g = 0; g < p.n_hull_groups; ++g) {     // uniform addresses: LDS broadcasts
                const float4 a = s_hull[2 * g], b = s_hull[2 * g + 1];
                gm |= line_box(ox, oy, oz, ix, iy, iz, a.x, a.y, a.z, b.x, b.y, b.z, pad2) ? 1ull << g : 0ull;
            }
        while (gm) {
            const uint32_t g = (uint32_t)__ffsll((long long)gm) - 1u;
            gm &= gm - 1ull;
            const uint32_t l0 = 8u * g;
            uint32_t lm = 0;
#pragma unroll
            for (uint32_t j = 0; j < 8; ++j) {
                const uint32_t l = l0 + j < p.n_hull_leaves ? l0 + j : p.n_hull_leaves - 1u;
                const float4 a = s_leaf[2 * l], b = s_leaf[2 * l + 1];
                lm |= (l0 + j < p.n_hull_leaves && line_box(ox, oy, oz, ix, iy, iz, a.x, a.y, a.z, b.x, b.y, b.z, pad2)) ? 1u << j : 0u;
            }
            while (lm) {
                const uint32_t l = l0 + (uint32_t)__ffs((int)lm) - 1u;
                lm &= lm - 1u;
                const uint32_t first = __float_as_uint(s_leaf[2 * l].w), cnt = __float_as_uint(s_leaf[2 * l + 1].w);
                const float4 *tp = s_tri + 3u * first;
                const float4 *tq = tp + (cnt > 1u ? 3 : 0);
                const float4 v0 = tp[0], v1 = tp[1], v2 = tp[2], w0 = tq[0], w1 = tq[1], w2 = tq[2];
                hull_face(shear(rp, v0.x, v0.y, v0.z), shear(rp, v1.x, v1.y, v1.z), shear(rp, v2.x, v2.y, v2.z),
                          __float_as_uint(v0.w), __float_as_uint(v1.w), __float_as_uint(v2.w), first);
                if (cnt > 1u)
                    hull_face(shear(rp, w0.x, w0.y, w0.z), shear(rp, w1.x, w1.y, w1.z), shear(rp, w2.x, w2.y, w2.z),
                              __float_as_uint(w0.w), __float_as_uint(w1.w), __float_as_uint(w2.w), first + 1u);
            }
        }
    } else {
        // Per-lane stackless traversal of the threaded hull tree (DFS pre-order, skip links):
        // ray-independent visiting order, every crossing of the ray's LINE is found.  Works for
        // incoherent batches (random training rays) as well as for camera frames.
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

    if (active) {
        const bool alive = nhull == 2 && !flag;
        const bool first0 = ht0 < ht1;
        uint4 e;
        e.x = alive ? 4u * (first0 ? hc0 : hc1) + (first0 ? he0 : he1) : 0u;   // variant = (tet record, entry face)
        e.y = first0 ? hf0 : hf1;
        e.z = (first0 ? hs0 : hs1) | (why << 24) | ((risk & 1u) << 28) | ((flag ? 2u : alive ? 1u : 0u) << 29);
        e.w = __float_as_uint(first0 ? ht1 : ht0);
        p.hull_entry[ray] = e;
    }
}

template <typename Order>
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
    const uint32_t super = blockIdx.x / (8 * XCD_GROUP), rem = blockIdx.x % (8 * XCD_GROUP);
    const uint32_t lb = super * 8 * XCD_GROUP + (rem & 7) * XCD_GROUP + (rem >> 3);
    if (lb >= nblk) return;
    const size_t ray = (size_t)lb * WALK_BLOCK + threadIdx.x;
    const bool active = ray < t.num_items;
    const size_t rr = active ? ray : 0;

    const float ox = t.origins[3 * rr], oy = t.origins[3 * rr + 1], oz = t.origins[3 * rr + 2];
    const float dx = t.dirs[3 * rr], dy = t.dirs[3 * rr + 1], dz = t.dirs[3 * rr + 2];
    const RayPre rp = ray_pre(ox, oy, oz, dx, dy, dz);

    bool flag = false;  // ray must be re-traced by the general path
    uint32_t why = 0;   // first reason (1..12), counted in stats[4 + why]
#if TN_WALK_DIAG
    const unsigned long long tick0 = wall_clock64();
#endif

    // ------------------------------------------------------------------ hull crossing search
    // Wave-uniform traversal of the (small) hull BVH: a node is visited if ANY lane's line hits
    // its padded box; box / triangle data are read through uniform (scalar) loads, every lane
    // tests its own ray.  No stack: the tree has a fixed depth (<= 3 internal levels).
    // rounding distance of a projected vertex: the box padding of the BVH path (tn_device.h: line_box)
    const float pad = 16.0f * 1.1920929e-7f * (fmaxf(fabsf(ox), fmaxf(fabsf(oy), fabsf(oz))) + p.scene_max);
    // Rule 8 (fold guard, see the header): delta = 7 * 2^-24 * (|o| + scene) bounds the error of a sheared 2-D vertex
    // position; a tet whose neighbourhood holds a tet thinner than 32 delta (exponent compare, conservative) AND one
    // of whose edges passes within 8 delta of the ray sends the ray to the BVH path.
    const float delta = 0.21875f * pad;               // 7/32 of the box padding
    const float kappa = 8.0f * delta;
    const uint32_t thin_exp = (__float_as_uint(32.0f * delta) >> 23) & 0xFFu;
    // RISK classes of the certification (round 5): the two guards above hand a ray over when it passes
    // within 8 delta of a hull edge / of an edge of a thin-neighbourhood tet.  What is not proved is that the degenerate
    // feature always lies THAT close to a tested edge, so the same tests with a wider band (option "risk_band": twice as wide
    // = 16 delta by default) mark a certified ray as "at risk": every such ray -- not one in 256 -- is re-counted by the BVH
    // cross-check (k_verify_counts).  bit 0: within 8 delta (hand over), bit 1: within the wide band (verify)
    const float kappa2 = p.risk_band * kappa;
    auto edge_band = [&](float e, const SV &X, const SV &Y) -> uint32_t {
        const float len = fabsf(X.x - Y.x) + fabsf(X.y - Y.y), ae = fabsf(e);
        return (ae <= kappa * len ? 1u : 0u) | (ae <= kappa2 * len ? 2u : 0u);
    };
    uint32_t risk = 0;   // bit 0: hull near-miss, bit 1: thin-neighbourhood near-miss
    // the entry found by k_hull_entry
    const uint4 ent = active ? p.hull_entry[ray] : make_uint4(0u, 0u, 0u, 0u);
    const uint32_t ent_state = ent.z >> 29;
    if (ent_state == 2u) { flag = true; why = (ent.z >> 24) & 15u; }
    risk = (ent.z >> 28) & 1u;
    // ------------------------------------------------------------------ the walk
    // Every recorded (valid) hit k of this ray is one 16-byte entry of the hit log, at
    // log[(wave of 64 rays) * M + k][lane]: the lanes of a wave store consecutive bytes.
    const size_t gw = (size_t)lb * (WALK_BLOCK / 64) + (size_t)wave;
    const size_t logoff = gw * (size_t)M * 64 + (size_t)lane;
    uint4 *mylog = p.hit_log + logoff;

    bool alive = ent_state == 1u;
    const uint32_t f_in0 = ent.y;
    const float t_out = __uint_as_float(ent.w);
    uint32_t c = alive ? ent.x : 0u;                                 // variant = (tet record, entry face)
    // The entry face in its STORED order: sheared vertices A,B,C and edge functions U=E(B,C), V=E(C,A), W=E(A,B).
    // From here on they are carried: the exit face of a step, evaluated in its stored order, is the entry face of
    // the next (same face-table entry), so per step only ONE vertex is sheared and three edge functions against
    // it decide the exit.
    SV A = {0.f, 0.f, 0.f}, B = A, C = A;
    if (alive) {
        const float4 *tp = p.hull_tris + 3 * (size_t)(ent.z & 0xFFFFFFu);
        const float4 v0 = tp[0], v1 = tp[1], v2 = tp[2];
        A = shear(rp, v0.x, v0.y, v0.z); B = shear(rp, v1.x, v1.y, v1.z); C = shear(rp, v2.x, v2.y, v2.z);
        // a vertex of the entry face within rounding distance of the ray (reason 4, see the header)
        if (fabsf(A.x) + fabsf(A.y) < pad || fabsf(B.x) + fabsf(B.y) < pad || fabsf(C.x) + fabsf(C.y) < pad) { flag = true; why = 4; alive = false; }
    }
    float Uc = edge_f(B, C), Vc = edge_f(C, A), Wc = edge_f(A, B);
    bool have_prev = false;   // a valid hit has been recorded
    Order ord;                // the order test (OrderR6; OrderR5 with option cert_ends = 0)
    float pt = 0.f;           // t of the previous recorded hit
    uint32_t nhits = 0, nshort = 0;
    uint32_t steps = 0;
#if TN_WALK_DIAG
    uint32_t lit_why = 0, n_viol = 0;
#endif
    Var cur = load_var(p.vars, c);
    if (alive && ((cur.code_hi >> 8) & 0xFFu) <= thin_exp) {
        const uint32_t eb0 = edge_band(Uc, B, C) | edge_band(Vc, C, A) | edge_band(Wc, A, B);
        if (eb0 & 1u) { flag = true; why = 8; alive = false; }
        risk |= eb0 & 2u;
    }
    if (alive) {
        // the entry hull face itself may be the first recorded hit (exit code 3 = "hull face id in the low bits")
        float tt, uu, vv;
        if (tri_finish(Uc, Vc, Wc, A.z, B.z, C.z, tt, uu, vv)) {
            ord.step(true, false, 0.f, tt, false, 0u);
            have_prev = true; pt = tt; nhits = 1;
            mylog[0] = make_uint4(__float_as_uint(tt), __float_as_uint(uu), __float_as_uint(vv), f_in0 | (3u << 30));
        }
    }

    // (round 6: issue priority by chord length -- s_setprio 3..0 by quarters of the hull diagonal, so that the longest rays of
    // a frame go first -- changed nothing: +-0.2 %, profiles/r06aa_prio_sweep.txt)
#if TN_WALK_DIAG
    const unsigned long long tick1 = wall_clock64();
#endif
    while (alive) {
        // All checks of a step accumulate into `bad` (first reason kept), straight-line: as nested ifs the checks
        // became a dozen exec-masked branches per step.
        uint32_t bad = 0;
        const SV P = shear(rp, cur.px, cur.py, cur.pz);
        const float ea = edge_f(P, A), eb = edge_f(P, B), ec = edge_f(P, C);
        bad = (fabsf(P.x) + fabsf(P.y) < pad) ? 4u : bad;                  // vertex within rounding distance of the ray
        bad = (!bad && (ea == 0.0f || eb == 0.0f || ec == 0.0f)) ? 5u : bad;
        // fold guard: a thin neighbourhood and one of the tet's NEW edges (n-a, n-b, n-c; the others were new edges of
        // a tet visited earlier, which carries the same flag when both end points of the edge have thin stars)
        const bool thin = ((cur.code_hi >> 8) & 0xFFu) <= thin_exp;
        const uint32_t band = edge_band(ea, P, A) | edge_band(eb, P, B) | edge_band(ec, P, C);
        bad = (!bad && thin && (band & 1u)) ? 8u : bad;
        risk |= thin ? (band & 2u) : 0u;
        // exit candidates: the faces opposite a {n,b,c}, b {n,c,a}, c {n,a,b}; a face is crossed iff its three
        // cyclic edge functions agree in sign: E(n,b), E(b,c) = Uc, E(c,n) = -ec, and cyclically
        const bool sa = ea > 0.0f, sb = eb > 0.0f, sc = ec > 0.0f;
        const bool su = Uc > 0.0f, sv = Vc > 0.0f, sw = Wc > 0.0f;
        const bool ha = (sb == su) && (su != sc);
        const bool hb = (sc == sv) && (sv != sa);
        const bool hc = (sa == sw) && (sw != sb);
        const uint32_t hmask = (ha ? 1u : 0u) | (hb ? 2u : 0u) | (hc ? 4u : 0u);
        bad = (!bad && __popc(hmask) != 1) ? 6u : bad;
        const uint32_t x = (__ffs(hmask) - 1) & 3u;  // exit 0..2 (3 only together with bad)
        const uint32_t nb = sel3u(cur.nb0, cur.nb1, cur.nb2, x);
        const bool last = nb == TN_EMPTY;
        // the next record is requested as soon as the exit is known
        const Var nxt = load_var(p.vars, (last || bad) ? c : nb);
        __builtin_amdgcn_sched_barrier(0);

        // the exit face in its stored order: 12-bit code of exit x out of the 36-bit word
        const bool x0 = (x & 1u) != 0, x1 = (x & 2u) != 0;
        const uint32_t w01 = x0 ? (cur.code_lo >> 12) : cur.code_lo;
        const uint32_t w2 = (cur.code_lo >> 24) | (cur.code_hi << 8);
        const uint32_t code = (x1 ? w2 : w01) & 0xFFFu;
        const SV A2 = selsv(P, A, B, C, code & 3u), B2 = selsv(P, A, B, C, (code >> 2) & 3u), C2 = selsv(P, A, B, C, (code >> 4) & 3u);
        const float U = edge_f(B2, C2), V = edge_f(C2, A2), W = edge_f(A2, B2);
        float ct = 0.f, cu = 0.f, cv = 0.f;
        const bool valid = tri_finish(U, V, W, A2.z, B2.z, C2.z, ct, cu, cv);

        // order of the pair (previous hit, this hit); see "certified" in the header.  Sorted order of the two = chain order; an
        // exact tie in t is ordered by face id: the previous recorded hit is this tet's entry face, and "id of exit x > id of the
        // entry face" is bit 16 + x of the record's code_hi
        if (__builtin_amdgcn_ballot_w64(ord.needs_step(valid, have_prev, ct)) != 0ull) {   // wave-uniform
            const uint32_t viol = ord.step(valid, have_prev, pt, ct, ((cur.code_hi >> (16u + x)) & 1u) != 0, nhits);
#if TN_WALK_DIAG
            lit_why = lit_why ? lit_why : viol;
            n_viol += viol ? 1u : 0u;
#else
            (void)viol;
#endif
            nshort += (valid && have_prev && fabsf(pt - ct) < TN_EPS) ? 1u : 0u;   // (a short pair always joins: counted here only)
        } else {
            ord.fast(valid, ct);
        }
        bad = (!bad && !valid && have_prev) ? 10u : bad;        // the hit list is not a suffix of the chain
        bad = (!bad && valid && nhits >= M - 1) ? 9u : bad;     // more than M-1 faces
        {
            // hit `nhits` of this ray: (t, u, v) in the face's stored order + the tet it closes (variant, exit).  The store is
            // UNCONDITIONAL (round 6): behind a branch the compiler ends every step with s_waitcnt vmcnt(0) -- the step then waits
            // for the acknowledgement of its own log store, which under the write stream of the fill running beside the walk takes
            // several times the record load's latency -- while a straight-line store leaves vmcnt(1) at the top of the loop (only
            // the next record must be back).  A step without a valid hit writes a dummy entry into slot `nhits`, which the next
            // valid hit overwrites and no reader looks at (readers stop at the ray's hit count); slot M - 1 takes the entries of
            // a ray that overflows (reason 9: its row comes from the BVH path).
            const uint32_t slot_k = nhits < M - 1 ? nhits : M - 1;
            mylog[(size_t)slot_k * 64] = make_uint4(__float_as_uint(ct), __float_as_uint(cu), __float_as_uint(cv), c | (x << 30));
        }
        nhits += valid ? 1u : 0u;
        pt = valid ? ct : pt;
        have_prev = have_prev || valid;
        // the chain must end in the other crossed hull face: its t was computed by the hull search with the same expression
        // tree, so the last valid hit carries it bit for bit
        bad = (!bad && last && valid && !(ct == t_out)) ? 11u : bad;
        steps++;
        bad = (!bad && !last && steps > MAX_WALK_STEPS) ? 12u : bad;
        flag = bad != 0;
        why = bad;
        alive = !bad && !last;
        // a lane that stops never reads its walk state again: advance unconditionally
        c = nb;
        cur = nxt;
        A = A2; B = B2; C = C2;
        Uc = U; Vc = V; Wc = W;
    }

#if TN_WALK_DIAG
    {
        const unsigned long long tick2 = wall_clock64();
        uint32_t ms = steps;
        for (int off = 32; off; off >>= 1) { const uint32_t o2 = __shfl_xor(ms, off); ms = o2 > ms ? o2 : ms; }
        if (lane == 0) {
            atomicAdd(&g_walk_time[0], 1ull);
            atomicAdd(&g_walk_time[1], tick1 - tick0); atomicAdd(&g_walk_time[2], tick2 - tick1);
            atomicMax(&g_walk_time[3], tick1 - tick0); atomicMax(&g_walk_time[4], tick2 - tick1);
            atomicMax(&g_walk_time[5], (unsigned long long)ms); atomicAdd(&g_walk_time[6], (unsigned long long)ms);
        }
    }
#endif
    // ------------------------------------------------------------------ classes, hand-over lists, hit counts
    const bool order_ok = ord.finish(nhits);
    const bool drop2 = ord.drop2();
    // every consecutive pair that is not short (rule C: minus the segment of hit 2, which the reference loses)
    const uint32_t nseg = nhits ? nhits - 1 - nshort - (drop2 ? 1u : 0u) : 0;
    const uint32_t wflag = drop2 ? 1u : 0u;
    if (active) {
        if (flag || (!order_ok && !p.literal_list)) {
            const uint32_t slot = atomicAdd(p.fallback_count, 1u);
            p.fallback_list[slot] = (uint32_t)(p.ray_base + ray);
            if (t.stats) atomicAdd(&t.stats[4 + (flag ? why : 7u)], 1ull);
            p.walk_n[ray] = TN_EMPTY;   // the BVH kernel writes the whole row
        } else if (!order_ok) {
#if TN_WALK_DIAG
            atomicAdd(&g_walk_diag[0], 1ull);
            atomicAdd(&g_walk_diag[lit_why ? lit_why : ord.end_reason()], 1ull);
            atomicAdd(&g_walk_diag[9], n_viol == 1 ? 1ull : 0ull);                 // rays with exactly one violation
            atomicAdd(&g_walk_diag[10], (unsigned long long)n_viol);               // violations in total
            atomicAdd(&g_walk_diag[12], (unsigned long long)nhits);               // hits of literal rays
#endif
            if (t.stats) atomicAdd(&t.stats[4 + 7], 1ull);
            const uint32_t slot = atomicAdd(p.literal_count, 1u);
            p.literal_list[slot] = make_uint2((uint32_t)ray, nhits);   // index within this walk launch (= log row)
            p.walk_n[ray] = TN_EMPTY;   // k_postprocess_log writes the whole row
        } else {
            p.walk_n[ray] = nhits | (wflag << 30);   // hits in the log (0 for a miss); bit 30: rule C (drop hit 2's segment)
            t.out_num[ray] = nseg;
            if (risk && p.risk_list) {  // certified, but inside the wide band of a guard: cross-checked, every one of them
                p.risk_list[atomicAdd(p.risk_count, 1u)] = (uint32_t)ray;
                if (t.stats) atomicAdd(&t.stats[24 + ((risk >> 1) & 1u)], 1ull);   // 24: hull near-miss (only), 25: thin neighbourhood
            }
        }
    }
}


// Hit log -> the SEGMENT part of the rows of the certified rays.  One wave works on EIGHT consecutive rays at a
// time: lane = (ray a = lane & 7, hit h = lane >> 3), so a load instruction fetches 8 hits of each of the 8 rays as
// 8 full 128-byte lines of the log (entries of neighbouring rays are neighbours in the log).  For a certified ray
// (header of this file) hits k-1 and k bound the tet recorded with hit k, and the pair is a segment unless it is
// shorter than eps; emitted slots are numbered by a per-ray prefix count over the wave ballot.  The tet id /
// vertex ids / combine_indices code come from the writer's 32-byte record -- of (tet, entry face) (WalkCold: ready-made) on
// meshes whose table the L2s hold, of the TET (WalkTet, round 4: a quarter of the table, (n, a, b, c) and the code derived
// per segment) on larger ones (tn_common.h) -- two 16-byte quads; bary_out is selected exactly as combine_indices does (optix_trace_rays.cu:39-75).  The slots between the
// last segment and the next multiple of 32 (where all four row arrays are on a 128-byte line boundary) get their
// tail constants here, so that k_fill_range starts every row on a line boundary and no line is written by two kernels.
//
// Stores: the 52-byte segment records of an iteration (8 rays x up to 8U slots) are staged in the wave's LDS region
// laid out [array][ray][slot] and read back with lane = (ray, consecutive dword / 8-byte / 16-byte unit of that ray's
// run): every store instruction writes contiguous runs per ray instead of 8 rows in 32..192-byte pieces (round 2b's
// ablation: the scattered stores were half of the direct-store kernel's time, profiles/r02b_writer_ablate.txt).
//
// Loads: the kernel is latency-bound (round 3 PMC, profiles/r03a_pmc_1.txt: its waves spend 66 % of their cycles in
// s_waitcnt).  Per iteration the chain is log entries -> walk records -> stores, and gfx950 retires loads and stores
// of a wave in issue order (one vmcnt), so a load issued after an iteration's stores is only "back" once those are
// acknowledged.  Hence the software pipeline: the entries of the NEXT iteration -- of this group, or of the wave's
// next group, whose hit counts were requested a whole group earlier -- are requested BEFORE this iteration's record
// loads and stores.
// Groups are dealt round-robin to the waves (the rays that miss the mesh are clustered).  Round 3 measured what bounds
// the kernel: its waves spend 66 % of their cycles in s_waitcnt and the average wave is alive for 55 % of the kernel
// (SQ_WAVE_CYCLES / SQ_WAVES, profiles/r03a_pmc_1.txt), yet neither more waves (U = 2: half the registers and LDS, 4
// waves per SIMD: 3-5 % slower per frame, profiles/r03b_writer.txt) nor a dynamic hand-out of the groups through an
// atomic counter (every wave busy to the end: 4-5 % slower, profiles/r03d_writer_ab.txt) help -- with all waves active
// each one waits longer: the kernel is bound by what the memory system delivers for its mix of 16-byte log reads, 32-byte
// record gathers and partial-line row writes (4.3-4.6 TB/s of raw traffic).  U = chunks of 8 hits per ray per iteration
// (4: 32 hits per ray in flight per wave).  The wave index is made wave-uniform (readfirstlane): group index, log base
// and row bases live in scalar registers.
namespace {
template <int U>
struct SW {
    static constexpr int SLOTS = 8 * U;                   // segment slots per ray per iteration
    static constexpr int STRIDE = SLOTS + 1;              // padded against bank conflicts between the 8 rays
    static constexpr int CELLS = 0;                       // [8][STRIDE] dwords
    static constexpr int DIST = 8 * STRIDE;               // [8][STRIDE][2]
    static constexpr int BARY = DIST + 8 * STRIDE * 2;    // [8][STRIDE][6]
    static constexpr int VERTS = BARY + 8 * STRIDE * 6;   // [8][STRIDE][4]
    static constexpr int META = VERTS + 8 * STRIDE * 4;   // [8] x {segments of this iteration, first slot}
    static constexpr int TOTAL = META + 16;               // U = 4: 3448 dwords = 13,792 B per wave; U = 2: 1784 = 7,136 B
};
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
}  // namespace

template <int U, bool PER_TET>
__global__ __launch_bounds__(256, 2) void k_write_segments(WriteParams q) {
    using W = SW<U>;
    __shared__ __attribute__((aligned(16))) uint32_t smem[4 * W::TOTAL];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint32_t *L = smem + wave * W::TOTAL;
    const uint32_t a = (uint32_t)lane & 7u, h = (uint32_t)lane >> 3;
    const uint32_t M = q.M;
    const size_t G = (q.num_rays + 7) / 8;                 // groups of 8 rays
    const size_t nwaves = (size_t)gridDim.x * 4;
    const unsigned long long raymask = 0x0101010101010101ull << a;

    auto hits_of = [&](size_t g) -> uint32_t {             // walk_n of ray 8g + a (TN_EMPTY: not this kernel's row)
        const size_t r = 8 * g + a;
        return (g < G && r < q.num_rays) ? q.walk_n[r] : TN_EMPTY;
    };
    auto log_of = [&](size_t g) -> const uint4 * {         // (scalar) entry k of ray 8g + a at [k * 64 + a]
        const size_t r0 = 8 * g;
        return q.hit_log + (r0 >> 6) * (size_t)M * 64 + (r0 & 63);
    };
    auto load_entries = [&](uint4 (&e)[U], const uint4 *lg, uint32_t nh, uint32_t c0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t k = c0 + 8 * u + h;
            e[u] = make_uint4(0u, 0u, 0u, 0u);
            if (k < nh) e[u] = lg[k * 64u + a];             // scalar base + 32-bit lane offset
        }
    };

    size_t g = (size_t)blockIdx.x * 4 + wave;
    if (g >= G) return;
    size_t g_next = g + nwaves, g_next2 = G;
    uint32_t nh_raw = hits_of(g);
    uint32_t nh_next_raw = hits_of(g_next);                // in flight during the whole first group
    uint4 e[U];
    load_entries(e, log_of(g), nh_raw == TN_EMPTY ? 0u : (nh_raw & 0x3FFFFFFFu), 0);
    for (; g < G; g = g_next, g_next = g_next2) {
        const bool skip = nh_raw == TN_EMPTY;   // literal / fallback ray (or padding): the row belongs to another kernel
        const uint32_t nh = skip ? 0u : (nh_raw & 0x3FFFFFFFu);
        const bool drop2 = !skip && (nh_raw >> 30) != 0u;   // rule C of the walk's order test: the reference loses hit 2's segment
        uint32_t mx = nh;                       // max over the 8 rays (the value is replicated over h)
#pragma unroll
        for (int off = 1; off < 8; off <<= 1) {
            const uint32_t o = (uint32_t)__shfl_xor((int)mx, off);
            mx = o > mx ? o : mx;
        }
        mx = __builtin_amdgcn_readfirstlane(mx);
        const uint4 *lg = log_of(g);
        // rows of the group: scalar bases (first slot of ray 8g) + 32-bit lane offsets (< 8 M slots)
        const size_t row0 = 8 * g * (size_t)M;
        uint32_t *const g_cells = q.out_cells + row0;
        float *const g_dist = q.out_dist + 2 * row0;
        float *const g_bary = q.out_bary + 6 * row0;
        uint32_t *const g_verts = q.out_verts ? q.out_verts + 4 * row0 : nullptr;
        const uint32_t row = a * M;
        // the group after the next: its hit counts are requested now, needed one group later
        g_next2 = g_next + nwaves;
        const uint32_t nh_next = nh_next_raw == TN_EMPTY ? 0u : (nh_next_raw & 0x3FFFFFFFu);
        const uint32_t nh_next2_raw = hits_of(g_next2);
        uint32_t nseg = 0;
        uint4 carry = make_uint4(0u, 0u, 0u, 0u);   // hit c0 - 1 of ray a
        uint32_t c0 = 0;
        do {
            // ---- request the next iteration's entries first (see the header comment)
            uint4 en[U];
            const bool more = c0 + 8 * U < mx;   // wave-uniform
            if (more) load_entries(en, lg, nh, c0 + 8 * U);
            else load_entries(en, log_of(g_next), g_next < G ? nh_next : 0u, 0);
            // ---- previous hit of every lane (lane - 8, or the last hit of the previous chunk), emission, slots
            const uint32_t base = nseg;
            uint4 pe[U];
            uint32_t slot[U];   // slot within this iteration (0 .. 8U-1), TN_EMPTY: no segment
            unsigned long long any = 0;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t k = c0 + 8 * u + h;
                pe[u].x = (uint32_t)__shfl_up((int)e[u].x, 8); pe[u].y = (uint32_t)__shfl_up((int)e[u].y, 8);
                pe[u].z = (uint32_t)__shfl_up((int)e[u].z, 8); pe[u].w = 0u;
                if (h == 0) pe[u] = carry;
                carry.x = (uint32_t)__shfl((int)e[u].x, (int)a + 56); carry.y = (uint32_t)__shfl((int)e[u].y, (int)a + 56);
                carry.z = (uint32_t)__shfl((int)e[u].z, (int)a + 56);
                const bool emit = k >= 1 && k < nh && !(fabsf(__uint_as_float(pe[u].x) - __uint_as_float(e[u].x)) < TN_EPS) && !(drop2 && k == 2u);
                const unsigned long long mall = __ballot(emit);
                const unsigned long long m = mall & raymask;
                any |= mall;
                slot[u] = emit ? (nseg - base) + (uint32_t)__popcll(m & lanemask_lt()) : TN_EMPTY;
                nseg += (uint32_t)__popcll(m);
            }
            if (any) {   // wave-uniform
                // ---- the writer's records of the emitted segments (32 bytes: WalkCold or WalkTet): vertex ids | tet id, codes
                uint4 qa[U], qv[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    qa[u] = make_uint4(0u, 0u, 0u, 0u); qv[u] = qa[u];
                    if (slot[u] != TN_EMPTY) {
                        const uint32_t c = e[u].w & 0x3FFFFFFFu;
                        const uint32_t *rec = PER_TET ? reinterpret_cast<const uint32_t *>(q.tets + (c >> 2))
                                                      : reinterpret_cast<const uint32_t *>(q.cold + c);
                        qv[u] = *reinterpret_cast<const uint4 *>(rec);       // PER_TET: vert[4]; else (n, a, b, c)
                        qa[u] = *reinterpret_cast<const uint4 *>(rec + 4);   // PER_TET: orig, perm, cmb_lo, cmb_hi; else orig, cmb
                    }
                }
                // ---- segment records -> LDS [array][ray][slot]
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (slot[u] != TN_EMPTY) {
                        const uint32_t at = a * W::STRIDE + slot[u];
                        const uint32_t x = e[u].w >> 30, ent = e[u].w & 3u;
                        uint32_t cmb;        // position of a / b / c in the exit face's stored order
                        uint4 vids = qv[u];  // (n, a, b, c)
                        if constexpr (PER_TET) {
                            // what the record of (tet, entry face) holds ready-made, derived from the tet's record
                            const unsigned long long lo64 = (unsigned long long)qa[u].z | ((unsigned long long)qa[u].w << 32);
                            const uint32_t c18 = ent == 3u ? ((uint32_t)(lo64 >> 54) | ((qa[u].y >> 24) << 10)) : (uint32_t)(lo64 >> (18u * ent));
                            cmb = c18 >> (6u * x);
                            const uint32_t pm = qa[u].y >> (6u * ent);
                            vids = make_uint4(sel4u(qv[u].x, qv[u].y, qv[u].z, qv[u].w, ent), sel4u(qv[u].x, qv[u].y, qv[u].z, qv[u].w, pm & 3u),
                                              sel4u(qv[u].x, qv[u].y, qv[u].z, qv[u].w, (pm >> 2) & 3u),
                                              sel4u(qv[u].x, qv[u].y, qv[u].z, qv[u].w, (pm >> 4) & 3u));
                        } else {
                            cmb = qa[u].y >> (6u * x);
                        }
                        const float pt = __uint_as_float(pe[u].x), pu = __uint_as_float(pe[u].y), pv = __uint_as_float(pe[u].z);
                        const float ct = __uint_as_float(e[u].x), cu = __uint_as_float(e[u].y), cv = __uint_as_float(e[u].z);
                        const float r0f = 1.0f - cu - cv;
                        const uint32_t k0 = cmb & 3u, k1 = (cmb >> 2) & 3u, k2 = (cmb >> 4) & 3u;
                        L[W::CELLS + at] = qa[u].x;
                        *reinterpret_cast<float2 *>(L + W::DIST + 2 * at) = make_float2(pt, ct);
                        float2 *bp = reinterpret_cast<float2 *>(L + W::BARY + 6 * at);
                        bp[0] = make_float2(1.0f - pu - pv, pu);
                        bp[1] = make_float2(pv, sel4f(r0f, cu, cv, 0.f, k0));
                        bp[2] = make_float2(sel4f(r0f, cu, cv, 0.f, k1), sel4f(r0f, cu, cv, 0.f, k2));
                        *reinterpret_cast<uint4 *>(L + W::VERTS + 4 * at) = vids;   // (n, a, b, c)
                    }
                }
                if (h == 0) *reinterpret_cast<uint2 *>(L + W::META + 2 * a) = make_uint2(nseg - base, base);
                wave_lds_fence();
                // ---- LDS -> rows: lane = (ray a2, unit d of that ray's run of this iteration).  ALL units are read back into
                //      registers first, then ALL stores are issued back to back (round 6): with a store group per read group the
                //      compiler put an s_waitcnt vmcnt(0) in front of every group's ds_read (read off the ISA: 24 full drains of
                //      the wave's memory queue per iteration, 8 now) -- ~20 serialised store round trips per iteration.  Kernel
                //      alone -15 % (C2 0.72 -> 0.61 ms, C4 0.95 -> 0.80 ms: profiles/r06c_lib_ab.txt)
                constexpr uint32_t RPI = 64 / W::SLOTS;       // rays per instruction for the one-unit-per-slot arrays
                constexpr uint32_t NQ = 8 / RPI;
                uint32_t s_sl[NQ], s_cell[NQ];
                float2 s_dist[NQ];
                uint4 s_vert[NQ];
#pragma unroll
                for (uint32_t qd = 0; qd < NQ; ++qd) {        // cell ids (4 B), distances (8 B), vertex ids (16 B) per slot
                    const uint32_t a2 = RPI * qd + (uint32_t)lane / W::SLOTS, d = (uint32_t)lane % W::SLOTS;
                    const uint2 meta = *reinterpret_cast<const uint2 *>(L + W::META + 2 * a2);
                    const uint32_t at = a2 * W::STRIDE + d;
                    s_sl[qd] = d < meta.x ? a2 * M + meta.y + d : TN_EMPTY;
                    s_cell[qd] = L[W::CELLS + at];
                    s_dist[qd] = *reinterpret_cast<const float2 *>(L + W::DIST + 2 * at);
                    s_vert[qd] = *reinterpret_cast<const uint4 *>(L + W::VERTS + 4 * at);
                }
                constexpr uint32_t UB = 3 * W::SLOTS;         // barycentrics: 3 x 8 B per slot
                constexpr uint32_t NB = 8 * UB / 64;
                uint32_t b_off[NB];
                float2 b_val[NB];
#pragma unroll
                for (uint32_t qd = 0; qd < NB; ++qd) {
                    const uint32_t gi = 64u * qd + (uint32_t)lane;
                    const uint32_t a2 = gi / UB, d = gi - UB * a2;
                    const uint2 meta = *reinterpret_cast<const uint2 *>(L + W::META + 2 * a2);
                    b_off[qd] = d < 3u * meta.x ? 6u * (a2 * M + meta.y) + 2u * d : TN_EMPTY;
                    b_val[qd] = *reinterpret_cast<const float2 *>(L + W::BARY + 6 * (a2 * W::STRIDE) + 2 * d);
                }
                wave_lds_fence();                             // the staging region is free for the next iteration
#pragma unroll
                for (uint32_t qd = 0; qd < NQ; ++qd) {
                    if (s_sl[qd] != TN_EMPTY) {
                        g_cells[s_sl[qd]] = s_cell[qd];
                        *reinterpret_cast<float2 *>(g_dist + 2u * s_sl[qd]) = s_dist[qd];
                        if (g_verts) *reinterpret_cast<uint4 *>(g_verts + 4u * s_sl[qd]) = s_vert[qd];
                    }
                }
#pragma unroll
                for (uint32_t qd = 0; qd < NB; ++qd)
                    if (b_off[qd] != TN_EMPTY) *reinterpret_cast<float2 *>(g_bary + b_off[qd]) = b_val[qd];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) e[u] = en[u];
            c0 += 8 * U;
        } while (c0 < mx);
        // tail constants up to the next multiple of 32 slots (line boundary of all four arrays)
        if (q.dense_tails && !skip) {
            uint32_t n32 = (nseg + 31u) & ~31u;
            if (n32 > M) n32 = M;
            for (uint32_t sl = nseg + h; sl < n32; sl += 8) {
                const uint32_t slot = row + sl;
                g_cells[slot] = TN_EMPTY;
                *reinterpret_cast<float2 *>(g_dist + 2u * slot) = make_float2(0.f, 0.f);
                float2 *bp = reinterpret_cast<float2 *>(g_bary + 6u * slot);
                bp[0] = make_float2(0.f, 0.f); bp[1] = make_float2(0.f, 0.f); bp[2] = make_float2(0.f, 0.f);
                if (g_verts) *reinterpret_cast<uint4 *>(g_verts + 4u * slot) = make_uint4(TN_EMPTY, TN_EMPTY, TN_EMPTY, TN_EMPTY);
            }
        }
        nh_raw = nh_next_raw;
        nh_next_raw = nh_next2_raw;
    }
}

void launch_write_segments(const WriteParams &q, hipStream_t stream, unsigned max_blocks) {
    if (q.num_rays == 0) return;
    size_t blocks = (q.num_rays + 31) / 32;        // one group of 8 rays per wave
    // grid = what is resident at once (2 blocks per CU at 192 VGPRs): the groups are dealt round-robin over it
    const size_t cap = max_blocks ? max_blocks : (size_t)256 * 2;
    if (blocks > cap) blocks = cap;
    if (q.tets) hipLaunchKernelGGL((k_write_segments<4, true>), dim3((unsigned)blocks), dim3(256), 0, stream, q);
    else hipLaunchKernelGGL((k_write_segments<4, false>), dim3((unsigned)blocks), dim3(256), 0, stream, q);
}

// Constant tails: pure streaming stores (16 B per lane, whole 128-byte lines), a contiguous span of rows per wave.
// This is the bulk of the bytes of a trace_rays call (88 % at M = 512) and runs at the write ceiling.  Two uses:
//   all_rows = 1: slots [k_split, M) of EVERY row -- needs nothing from the walk, so it streams beside it (speculative
//                 fill, tn_api.hip).  Rows of literal / fallback rays and rays with more than k_split segments are
//                 included: the kernels that write those slots are ordered behind this one.
//   all_rows = 0: slots [ceil32(n), k_split) of the certified rows (k_write_segments has written [0, ceil32(n))).
template <bool NT>
__global__ __launch_bounds__(256) void k_fill_range(size_t num_rays, uint32_t M, uint32_t all_rows, uint32_t k_split,
                                                    const uint32_t *__restrict__ walk_n, const uint32_t *__restrict__ out_num,
                                                    uint32_t *__restrict__ out_cells,
                                                    float *__restrict__ out_bary, float *__restrict__ out_dist,
                                                    uint32_t *__restrict__ out_verts) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // row span, row bases: scalar
    const size_t nwaves = (size_t)gridDim.x * 4;
    const size_t span = (num_rays + nwaves - 1) / nwaves;   // consecutive rows are consecutive in memory
    const size_t r0 = ((size_t)blockIdx.x * 4 + wave) * span;
    const size_t r1 = r0 + span < num_rays ? r0 + span : num_rays;
    for (size_t r = r0; r < r1; ++r) {
        uint32_t lo = k_split, hi = M;
        if (!all_rows) {
            if (walk_n[r] == TN_EMPTY) continue;  // literal / fallback ray: those kernels write the whole row
            lo = (out_num[r] + 31u) & ~31u;
            if (lo > M) lo = M;
            hi = k_split;
        }
        if (lo >= hi) continue;
        fill_dwords<NT>(out_cells + r * M, lo, hi, TN_EMPTY, lane);
        fill_dwords<NT>(reinterpret_cast<uint32_t *>(out_dist + r * M * 2), 2 * lo, 2 * hi, 0u, lane);
        fill_dwords<NT>(reinterpret_cast<uint32_t *>(out_bary + r * M * 6), 6 * lo, 6 * hi, 0u, lane);
        if (out_verts) fill_dwords<NT>(out_verts + r * M * 4, 4 * lo, 4 * hi, TN_EMPTY, lane);
    }
}

// The same rows with the work cut FINE: one block per row, its four waves take 6-8 KB each (cells + distances | first half
// of the barycentrics | second half | vertex ids), six to eight 1 KB store instructions per wave and the block is gone.
// Rows in flight form one moving window per array and the dispatcher balances the channels: with long-lived waves that
// own fixed spans of rows the same 17 GB took 2.40 ... 3.07 ms depending on WHERE the driver had put the pages (fresh
// allocations of the same rows in one process, profiles/r06s_placement.txt), torch's own one-store-per-thread fill of the
// same pages 2.43 ... 2.50 ms (profiles/r06s_torch_fill.txt).
template <bool NT>
__global__ __launch_bounds__(256) void k_fill_rows_fine(size_t num_rays, uint32_t M, uint32_t all_rows, uint32_t k_split,
                                                        const uint32_t *__restrict__ walk_n, const uint32_t *__restrict__ out_num,
                                                        uint32_t *__restrict__ out_cells,
                                                        float *__restrict__ out_bary, float *__restrict__ out_dist,
                                                        uint32_t *__restrict__ out_verts) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const size_t r = blockIdx.x;
    uint32_t lo = k_split, hi = M;
    if (!all_rows) {
        if (walk_n[r] == TN_EMPTY) return;        // literal / fallback ray: those kernels write the whole row
        lo = (out_num[r] + 31u) & ~31u;
        if (lo > M) lo = M;
        hi = k_split;
    }
    if (lo >= hi) return;
    const uint32_t mid = 6u * lo + ((3u * (hi - lo) + 3u) & ~3u);   // 16-byte aligned when lo is
    uint32_t *bary = reinterpret_cast<uint32_t *>(out_bary + r * M * 6);
    if (wave == 0) {
        fill_dwords<NT>(out_cells + r * M, lo, hi, TN_EMPTY, lane);
        fill_dwords<NT>(reinterpret_cast<uint32_t *>(out_dist + r * M * 2), 2 * lo, 2 * hi, 0u, lane);
    } else if (wave == 1) {
        fill_dwords<NT>(bary, 6 * lo, mid < 6 * hi ? mid : 6 * hi, 0u, lane);
    } else if (wave == 2) {
        if (mid < 6 * hi) fill_dwords<NT>(bary, mid, 6 * hi, 0u, lane);
    } else if (out_verts) {
        fill_dwords<NT>(out_verts + r * M * 4, 4 * lo, 4 * hi, TN_EMPTY, lane);
    }
}

// The same rows as ONE LINEAR STREAM PER ARRAY, the arrays one after the other, one 16-byte store per thread and the thread
// is gone (round 6, the last of the fill experiments and the first to explain them).  Fresh allocations of the same rows
// in one process (same virtual addresses, new physical pages) moved every fill that writes the four arrays IN STEP -- rows
// dealt to persistent waves, a block per row, even torch-style one-store blocks interleaved 1 : 2 : 6 : 4 -- between 5.6 and
// 7.1 TB/s, while a single linear stream (this order; torch's own fill) stays at 7.0 - 7.15 TB/s wherever the pages lie
// (profiles/r06s_flat_fill.txt, r06s_torch_fill.txt, r06s_placement.txt).
//   block -> (array, 256 consecutive 16-byte units of it); a row holds M/4, M/2, 3M/2, M units (cells, distances,
//   barycentrics, vertex ids): powers of two and 3 x a power of two, so the row of a unit costs a shift (and a division by 3).
template <bool NT, bool UNI>
__global__ __launch_bounds__(256) void k_fill_linear(size_t num_rays, uint32_t M, uint32_t all_rows, uint32_t k_split, uint32_t log2_m4,
                                                     unsigned long long n_bary, unsigned long long n_verts, unsigned long long n_dist,
                                                     const uint32_t *__restrict__ walk_n, const uint32_t *__restrict__ out_num,
                                                     uint32_t *__restrict__ out_cells, float *__restrict__ out_bary,
                                                     float *__restrict__ out_dist, uint32_t *__restrict__ out_verts) {
    // n_bary / n_verts / n_dist: first block of the NEXT array (blocks are dealt bary, verts, dist, cells: largest first)
    unsigned long long b = blockIdx.x;
    u32x4 *base; uint32_t w4, sh; uint32_t val; bool three = false;     // w4 = 16-byte units per 4 slots; units per row = (M/4 << sh) (x3)
    if (b < n_bary) { base = reinterpret_cast<u32x4 *>(out_bary); w4 = 6; sh = 1; three = true; val = 0u; }
    else if (b < n_verts) { b -= n_bary; base = reinterpret_cast<u32x4 *>(out_verts); w4 = 4; sh = 2; val = TN_EMPTY; }
    else if (b < n_dist) { b -= n_verts; base = reinterpret_cast<u32x4 *>(out_dist); w4 = 2; sh = 1; val = 0u; }
    else { b -= n_dist; base = reinterpret_cast<u32x4 *>(out_cells); w4 = 1; sh = 0; val = TN_EMPTY; }
    const unsigned long long g = b * 256ull + threadIdx.x;             // unit of the array
    const uint32_t q = (uint32_t)(g >> (log2_m4 + sh));                // (x3 arrays: the row's third; else the row)
    const uint32_t row = three ? q / 3u : q;
    if (row >= num_rays) return;
    const uint32_t upr = (three ? 3u : 1u) << (log2_m4 + sh);          // units per row
    const uint32_t off = (uint32_t)(g - (unsigned long long)row * upr);
    uint32_t lo = k_split, hi = M;
    if (!all_rows) {
        // (a per-row lookup in front of the one store: the wave then lives for a load latency per KB and the fill is bound by
        // THAT -- 6.0 instead of 3.3 ms per C2 frame; four chunks per block with the lookups up front: 4.2 ms and the pure
        // fill falls to 6.0 - 6.5 TB/s; the lookup through the scalar cache (this code): 5.4 ms; a block per row with the
        // arrays one after the other: 4.1 ms, steadier (3.97 - 4.21) but never below the block-per-row fill of all four
        // arrays (3.24 - 4.02); profiles/r06ac_linear_sweep*.txt, r06ae_*.txt.  So the tracer's tail fill, which needs
        // the lookup, stays with k_fill_rows_fine, and this kernel serves tn_fill_rows and option fill_blocks = -2)
        uint32_t wn, n;
        if constexpr (UNI) {   // M >= 256: a wave's 64 units lie in one row -> the lookup goes through the scalar cache
            const uint32_t rs = (uint32_t)__builtin_amdgcn_readfirstlane((int)row);
            wn = walk_n[rs]; n = out_num[rs];
        } else {
            wn = walk_n[row]; n = out_num[row];
        }
        if (wn == TN_EMPTY) return;               // literal / fallback ray: those kernels write the whole row
        lo = (n + 31u) & ~31u;
        if (lo > M) lo = M;
        hi = k_split;
    }
    // lo, hi are multiples of 4 slots: [lo, hi) slots = [lo / 4 * w4, hi / 4 * w4) units
    if (off < (lo >> 2) * w4 || off >= (hi >> 2) * w4) return;
    const u32x4 v4 = {val, val, val, val};
    if constexpr (NT) __builtin_nontemporal_store(v4, base + g);
    else base[g] = v4;
}

void launch_fill_range(size_t num_rays, uint32_t M, bool all_rows, const uint32_t *walk_n, const uint32_t *out_num,
                       uint32_t *out_cells, float *out_bary, float *out_dist, uint32_t *out_verts, hipStream_t stream,
                       uint32_t k_split, bool nontemporal, unsigned max_blocks) {
    if (num_rays == 0) return;
    if (max_blocks == FILL_LINEAR) {
        if (M < 4 || (M & (M - 1)) != 0) throw Error("k_fill_linear: M must be a power of two >= 4");
        uint32_t log2_m4 = 0;
        while ((4u << log2_m4) < M) ++log2_m4;
        const size_t cap_rows = 0x200000u;          // rows per launch: 2^21 rows x 13 M / 1024 blocks (1.1e8 at M = 4096) stays below 2^31
        for (size_t base = 0; base < num_rays; base += cap_rows) {
            const size_t n = num_rays - base < cap_rows ? num_rays - base : cap_rows;
            const unsigned long long u_cells = (unsigned long long)n * (M / 4);
            auto blocks_of = [](unsigned long long units) { return (units + 255ull) / 256ull; };
            const unsigned long long nb = blocks_of(6 * u_cells), nv = nb + (out_verts ? blocks_of(4 * u_cells) : 0ull),
                                     nd = nv + blocks_of(2 * u_cells), total = nd + blocks_of(u_cells);
            if (total > 0x7FFFFFFFull) throw Error("k_fill_linear: grid too large");
            auto args = [&](auto kern) {
                hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(256), 0, stream, n, M, all_rows ? 1u : 0u, k_split, log2_m4, nb, nv, nd,
                                   walk_n ? walk_n + base : nullptr, out_num ? out_num + base : nullptr, out_cells + base * M,
                                   out_bary + base * M * 6, out_dist + base * M * 2, out_verts ? out_verts + base * M * 4 : nullptr);
            };
            if (M >= 256) { if (nontemporal) args(k_fill_linear<true, true>); else args(k_fill_linear<false, true>); }
            else { if (nontemporal) args(k_fill_linear<true, false>); else args(k_fill_linear<false, false>); }
        }
        return;
    }
    if (max_blocks == FILL_FINE) {
        for (size_t base = 0; base < num_rays; base += 0x40000000u) {     // grid.x limit
            const size_t n = num_rays - base < 0x40000000u ? num_rays - base : 0x40000000u;
            if (nontemporal)
                hipLaunchKernelGGL(k_fill_rows_fine<true>, dim3((unsigned)n), dim3(256), 0, stream, n, M, all_rows ? 1u : 0u, k_split,
                                   walk_n ? walk_n + base : nullptr, out_num ? out_num + base : nullptr, out_cells + base * M,
                                   out_bary + base * M * 6, out_dist + base * M * 2, out_verts ? out_verts + base * M * 4 : nullptr);
            else
                hipLaunchKernelGGL(k_fill_rows_fine<false>, dim3((unsigned)n), dim3(256), 0, stream, n, M, all_rows ? 1u : 0u, k_split,
                                   walk_n ? walk_n + base : nullptr, out_num ? out_num + base : nullptr, out_cells + base * M,
                                   out_bary + base * M * 6, out_dist + base * M * 2, out_verts ? out_verts + base * M * 4 : nullptr);
        }
        return;
    }
    size_t blocks = (num_rays + 3) / 4;           // >= one ray per wave
    // after the writer: 2 blocks (8 waves) per CU hold the write ceiling, and the latency-bound kernels running beside
    // the fill are less starved than with 8 per CU (profiles/r01_fill_grid.txt); beside the walk: 2048 blocks
    // (profiles/r02p_specfill2.txt)
    const size_t cap = max_blocks ? max_blocks : (all_rows ? 2048 : 256 * 2);
    if (blocks > cap) blocks = cap;
    if (nontemporal)
        hipLaunchKernelGGL(k_fill_range<true>, dim3((unsigned)blocks), dim3(256), 0, stream, num_rays, M, all_rows ? 1u : 0u, k_split,
                           walk_n, out_num, out_cells, out_bary, out_dist, out_verts);
    else
        hipLaunchKernelGGL(k_fill_range<false>, dim3((unsigned)blocks), dim3(256), 0, stream, num_rays, M, all_rows ? 1u : 0u, k_split,
                           walk_n, out_num, out_cells, out_bary, out_dist, out_verts);
}

// 64-byte build records -> the three consumer tables (tn_common.h: WalkHot / WalkCold / WalkFid)
__global__ __launch_bounds__(256) void k_split_walk_records(size_t n4, const WalkVar *__restrict__ vars, WalkHot *__restrict__ hot,
                                                            WalkCold *__restrict__ cold, WalkTet *__restrict__ tets,
                                                            WalkFid *__restrict__ fidt) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const WalkVar v = vars[i];
    WalkHot h;
    h.pn[0] = v.pn[0]; h.pn[1] = v.pn[1]; h.pn[2] = v.pn[2];
    h.nb0 = v.nb[0]; h.nb1 = v.nb[1]; h.nb2 = v.nb[2]; h.code_lo = v.code_lo; h.code_hi = v.code_hi;
    WalkFid f;
    f.fid[0] = v.fid0; f.fid[1] = v.fid1; f.fid[2] = v.fid2; f.pad = 0;
    hot[i] = h; fidt[i] = f;
    if (cold) {
        WalkCold c;
        for (int k = 0; k < 4; ++k) c.vid[k] = v.vid[k];
        c.orig = v.orig;
        const unsigned long long code = (unsigned long long)v.code_lo | ((unsigned long long)(v.code_hi & 0xFu) << 32);
        c.cmb = (uint32_t)((code >> 6) & 63u) | ((uint32_t)((code >> 18) & 63u) << 6) | ((uint32_t)((code >> 30) & 63u) << 12);
        c.pad0 = 0; c.pad1 = 0;
        cold[i] = c;
    }
    if (tets && (i & 3u) == 0) {
        // the tet's record from its four entry-face records: vert[e] = the vertex opposite face e (vid[0] of record e); the
        // local index of a / b / c of entry e = any j with vert[j] == vid[m] (a degenerate tet may list a vertex twice: either
        // index names the same vertex id, and only ids are ever read through it)
        WalkTet tt;
        WalkVar w[4];
        w[0] = v; w[1] = vars[i + 1]; w[2] = vars[i + 2]; w[3] = vars[i + 3];
        for (int e = 0; e < 4; ++e) tt.vert[e] = w[e].vid[0];
        tt.orig = v.orig;
        uint32_t perm = 0;
        unsigned long long lo64 = 0;
        uint32_t top = 0;
        for (uint32_t e = 0; e < 4; ++e) {
            for (uint32_t m = 0; m < 3; ++m) {
                uint32_t j = 0;
                for (uint32_t k = 0; k < 4; ++k)
                    if (tt.vert[3 - k] == w[e].vid[m + 1]) j = 3 - k;   // first match wins
                perm |= j << (6 * e + 2 * m);
            }
            const unsigned long long code = (unsigned long long)w[e].code_lo | ((unsigned long long)(w[e].code_hi & 0xFu) << 32);
            const unsigned long long c18 = ((code >> 6) & 63ull) | (((code >> 18) & 63ull) << 6) | (((code >> 30) & 63ull) << 12);
            if (e < 3) lo64 |= c18 << (18 * e);
            else { lo64 |= c18 << 54; top = (uint32_t)(c18 >> 10); }
        }
        tt.perm = perm | (top << 24);
        tt.cmb_lo = (uint32_t)lo64; tt.cmb_hi = (uint32_t)(lo64 >> 32);
        tets[i >> 2] = tt;
    }
}
void launch_split_walk_records(size_t n4, const WalkVar *vars, WalkHot *hot, WalkCold *cold, WalkTet *tets, WalkFid *fidt, hipStream_t stream) {
    if (n4 == 0) return;
    hipLaunchKernelGGL(k_split_walk_records, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, n4, vars, hot, cold, tets, fidt);
}

void launch_trace_walk(const WalkParams &p, hipStream_t stream, size_t lds_reserve) {
    if (p.t.num_items == 0) return;
    const uint32_t nblk = (uint32_t)((p.t.num_items + WALK_BLOCK - 1) / WALK_BLOCK);
    // grid padded so that the remap (runs of XCD_GROUP blocks per XCD) is a bijection
    const uint32_t unit = 8 * XCD_GROUP;
    const uint32_t grid = (nblk + unit - 1) / unit * unit;
    // entry search: the flat hull table + the faces in LDS (up to 18 + 48 KB at HULL_FLAT_MAX faces)
    const size_t need = p.n_hull_leaves ? (size_t)(p.n_hull_groups + p.n_hull_leaves) * 32 + (size_t)p.n_hull * 48 : 0;
    // above 64 KB of dynamic LDS only by opt-in, per device (a process may hold tracers on several): set whenever it is needed
    if (need > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_hull_entry), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)((HULL_FLAT_MAX / 2 + HULL_FLAT_MAX / 16) * 32 + HULL_FLAT_MAX * 48)) != hipSuccess)
        throw Error("k_hull_entry: the device refused " + std::to_string(need) + " bytes of dynamic LDS");
    hipLaunchKernelGGL(k_hull_entry, dim3(grid), dim3(WALK_BLOCK), need, stream, p);
    if (lds_reserve > 64 * 1024) lds_reserve = 64 * 1024;
    if (p.cert_ends == 1u) hipLaunchKernelGGL(k_trace_walk<OrderR6>, dim3(grid), dim3(WALK_BLOCK), lds_reserve, stream, p);
    else if (p.cert_ends == 3u) hipLaunchKernelGGL(k_trace_walk<OrderR5e>, dim3(grid), dim3(WALK_BLOCK), lds_reserve, stream, p);
    else hipLaunchKernelGGL(k_trace_walk<OrderR5>, dim3(grid), dim3(WALK_BLOCK), lds_reserve, stream, p);
}

}  // namespace tn
