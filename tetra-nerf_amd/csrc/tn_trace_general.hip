// tn_trace_general.hip -- general all-hits trace path: ONE WAVEFRONT PER RAY.
//
// Replaces, for one ray per 64-lane wave:
//   __raygen__rg + optixTrace + __anyhit__ms   (src/optix/optix_trace_rays.cu:268-331)
//   bitonic_sort                                (:78-108)
//   post_process_tetrahedra                     (:110-266)
// of the reference.  Structure (all in LDS, nothing spills to the output rows):
//   1. wave-cooperative traversal of the 64-ary BVH: one lane per child box / per
//      triangle, __ballot + prefix popcount to push children / append hits;
//   2. bitonic sort of the <= M-1 hits on the 64-bit key (t bits << 32 | face id) --
//      a total order, so the result is independent of traversal order;
//   3. dedupe + pairing: a wave-parallel fast branch when the sorted list is a clean
//      chain (every consecutive pair shares a tetrahedron, no two consecutive gaps
//      below eps -- then the reference's phases reduce to "pair j with j+1, drop pairs
//      shorter than eps"), else the literal serial algorithm on lane 0;
//   4. coalesced row writes: segments by consecutive lanes, then the tail fill
//      (visited/verts = 0xFFFFFFFF, bary/dist = 0) so every output byte is written once.
#include "tn_device.h"
#include "tn_kernels.h"

namespace tn {

namespace {

struct WaveSmem {
    uint64_t *key;   // [M]   t bits << 32 | face id
    float *hu;       // [M]
    float *hv;       // [M]
    uint2 *hft;      // [M]   face -> tets of the sorted hit
    uint32_t *stack; // [STACK_CAP]
    uint8_t *mark;   // [M]
    uint8_t *emitf;  // [M]   slot j emits the segment (slot j, slot j+1)
};

__device__ __forceinline__ WaveSmem carve(char *smem, uint32_t M) {
    WaveSmem s;
    s.key = reinterpret_cast<uint64_t *>(smem);
    s.hft = reinterpret_cast<uint2 *>(s.key + M);
    s.hu = reinterpret_cast<float *>(s.hft + (M < 32 ? 32 : M));  // hft doubles as the leaf list (>= 64 entries)
    s.hv = s.hu + M;
    s.stack = reinterpret_cast<uint32_t *>(s.hv + M);
    s.mark = reinterpret_cast<uint8_t *>(s.stack + STACK_CAP);
    s.emitf = s.mark + M;
    return s;
}

__device__ __forceinline__ void wave_sync() { __syncthreads(); }  // 1-wave workgroups

// keep-the-nearest insertion once the hit buffer is full (rare): replace the current
// maximum key if the new key is smaller.  Final content = the M-1 smallest keys.
__device__ void insert_overflow(WaveSmem &s, uint32_t cap, uint64_t k, float u, float v, int lane) {
    // wave-parallel argmax over cap entries
    uint64_t best = 0;
    uint32_t bidx = 0;
    for (uint32_t i = lane; i < cap; i += 64) {
        const uint64_t x = s.key[i];
        if (x >= best) { best = x; bidx = i; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint64_t ob = __shfl_xor(best, off);
        const uint32_t oi = __shfl_xor(bidx, off);
        if (ob > best || (ob == best && oi > bidx)) { best = ob; bidx = oi; }
    }
    if (k < best && lane == 0) { s.key[bidx] = k; s.hu[bidx] = u; s.hv[bidx] = v; }
    wave_sync();
}

// ---- stage 3+4: dedupe / pairing / row write for one ray; hits sorted in LDS -------------
__device__ void postprocess_and_write(const WaveSmem &s, uint32_t nh, uint32_t M, const uint32_t *__restrict__ faces,
                                      const uint32_t *__restrict__ face_tets, uint32_t *__restrict__ out_num,
                                      uint32_t *__restrict__ out_cells, float *__restrict__ out_bary,
                                      float *__restrict__ out_dist, uint32_t *__restrict__ out_verts,
                                      unsigned long long *stats, int lane, bool compact = false) {
    // face -> tets of each sorted hit: random 8-byte reads, all of a lane's requests (eight per 512 hits) in flight at once
    for (uint32_t base = 0; base < nh; base += 512) {
        uint2 ft[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint32_t j = base + 64 * q + lane;
            if (j < nh) ft[q] = *reinterpret_cast<const uint2 *>(face_tets + 2 * (size_t)(uint32_t)s.key[j]);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint32_t j = base + 64 * q + lane;
            if (j < nh) { s.hft[j] = ft[q]; s.mark[j] = 0; }
        }
    }
    wave_sync();

    // clean-chain test
    bool bad = false;
    for (uint32_t j = lane; j + 1 < nh; j += 64) {
        uint32_t c;
        const float t0 = __uint_as_float((uint32_t)(s.key[j] >> 32));
        const float t1 = __uint_as_float((uint32_t)(s.key[j + 1] >> 32));
        if (!common_tet(s.hft[j], s.hft[j + 1], c)) bad = true;
        if (j + 2 < nh) {
            const float t2 = __uint_as_float((uint32_t)(s.key[j + 2] >> 32));
            if (fabsf(t1 - t0) < TN_EPS && fabsf(t2 - t1) < TN_EPS) bad = true;
        }
    }
    const bool clean = (__ballot(bad) == 0ull);

    // Decide which slots j emit the segment (slot j, slot j+1): in parallel for a clean chain, else by
    // the literal serial algorithm on lane 0.  The serial pass touches LDS only: the reference's swap
    // step moves a matched face to slot j+1 and never touches slots <= j again, so once the pass is over
    // every emitted segment is exactly (final slot j, final slot j+1) and the row data can be produced by
    // all lanes in parallel below.
    if (clean) {
        for (uint32_t j = lane; j < nh; j += 64) {
            bool em = false;
            if (j + 1 < nh) {
                const float t0 = __uint_as_float((uint32_t)(s.key[j] >> 32));
                const float t1 = __uint_as_float((uint32_t)(s.key[j + 1] >> 32));
                em = fabsf(t0 - t1) >= TN_EPS;
            }
            s.emitf[j] = em ? 1 : 0;
        }
    } else {
        // The literal algorithm, LOCALISED.  Its two serial phases only do something around "anomalies":
        //  * phase 1 (optix_trace_rays.cu:124-159) at slot j looks at the following slots within eps of t_j, so all its
        //    marks and clears stay inside j's maximal run of consecutive sub-eps gaps ("eps-chain"; the slot after a run
        //    is at least eps away from every member and real -- the first slot of a run is never cleared: it has no
        //    earlier slot in its window that could mark it).  Runs are disjoint, so ONE LANE PER RUN executes the literal
        //    loop of its run, all runs at once.
        //  * phase 2 (:188-257) at slot j is a plain "common step" (emit (j, j+1) iff the gap is >= eps; nothing moves)
        //    whenever slots j and j+1 are real faces sharing a tetrahedron.  Only other steps look ahead and swap, and a
        //    swap touches slots up to j+off.  Lane 0 therefore executes the literal step only from each anomalous slot
        //    on, and keeps going while the current slot is anomalous or at or below the highest slot a swap has touched;
        //    every other step takes the plain result computed by all lanes in parallel.
        // Same results as running both phases over all slots (round 2a did: ~260 serial iterations x 2 per ray).
        if (stats && lane == 0) atomicAdd(&stats[2], 1ull);
        auto Tk = [](uint64_t k) { return __uint_as_float((uint32_t)(k >> 32)); };
        auto T = [&](uint32_t j) { return Tk(s.key[j]); };
        auto ID = [&](uint32_t j) { return (uint32_t)s.key[j]; };
        // ---- phase 1: runs of sub-eps gaps -> list of run starts (in the traversal stack, free by now)
        uint32_t nruns = 0;
        for (uint32_t base = 0; base < nh; base += 64) {
            const uint32_t j = base + lane;
            bool start = false;
            if (j + 1 < nh) {
                const bool sj = fabsf(T(j + 1) - T(j)) < TN_EPS;
                const bool sp = j > 0 && fabsf(T(j) - T(j - 1)) < TN_EPS;
                start = sj && !sp;
            }
            const uint64_t m = __ballot(start);
            if (start) {
                const uint32_t slot = nruns + __popcll(m & lanemask_lt());
                if (slot < (uint32_t)STACK_CAP) s.stack[slot] = j;
            }
            nruns += __popcll(m);
        }
        wave_sync();
        const bool runs_fit = nruns <= (uint32_t)STACK_CAP;   // always at M <= 768; otherwise lane 0 walks all slots
        auto phase1_at = [&](uint32_t j) {
            if (ID(j) == TN_EMPTY) return;
            const float dn = T(j);
            bool clear_self = false;
            for (uint32_t off = 1; j + off < nh && (ID(j + off) == TN_EMPTY || fabsf(T(j + off) - dn) < TN_EPS); ++off) {
                uint32_t c;
                if (ID(j + off) != TN_EMPTY && common_tet(s.hft[j], s.hft[j + off], c)) {
                    if (ID(j) != ID(j + off)) clear_self = true;
                    if (s.mark[j + off]) s.key[j + off] = (s.key[j + off] & 0xFFFFFFFF00000000ull) | TN_EMPTY;
                    else s.mark[j + off] = 1;
                }
            }
            if (clear_self && s.mark[j]) s.key[j] = (s.key[j] & 0xFFFFFFFF00000000ull) | TN_EMPTY;
            s.mark[j] = 0;
        };
        if (runs_fit) {
            for (uint32_t ci = lane; ci < nruns; ci += 64) {
                for (uint32_t j = s.stack[ci];; ++j) {
                    phase1_at(j);
                    if (!(j + 1 < nh && fabsf(T(j + 1) - T(j)) < TN_EPS)) break;   // j was the last slot of the run
                }
            }
        } else if (lane == 0) {
            for (uint32_t j = 0; j + 1 < nh; ++j) phase1_at(j);
        }
        wave_sync();
        // ---- phase 2: plain steps by all lanes, anomalous slots -> bit mask (in the stack region)
        unsigned long long *amask = reinterpret_cast<unsigned long long *>(s.stack);   // [ceil(nh / 64)] <= 64 words
        for (uint32_t base = 0; base < nh; base += 64) {
            const uint32_t j = base + lane;
            bool anomalous = false;
            uint8_t em = 0;
            if (j + 1 < nh && ID(j) != TN_EMPTY) {
                uint32_t c;
                const bool plain = ID(j + 1) != TN_EMPTY && common_tet(s.hft[j], s.hft[j + 1], c);
                anomalous = !plain;
                em = (plain && fabsf(T(j) - T(j + 1)) >= TN_EPS) ? 1 : 0;
            }
            if (j < nh) s.emitf[j] = em;
            const uint64_t m = __ballot(anomalous);
            if (lane == 0) amask[base >> 6] = m;
        }
        wave_sync();
        if (lane == 0) {
            const uint32_t nwords = (nh + 63) >> 6;
            auto next_anomaly = [&](uint32_t from) -> uint32_t {   // first anomalous slot >= from (nh: none)
                for (uint32_t w = from >> 6; w < nwords; ++w) {
                    unsigned long long m = amask[w];
                    if (w == (from >> 6)) m &= ~0ull << (from & 63);
                    if (m) return (w << 6) + (uint32_t)__ffsll(m) - 1;
                }
                return nh;
            };
            uint32_t touched = 0;   // highest slot a swap has written
            uint32_t j = next_anomaly(0);
            while (j < nh) {
                s.emitf[j] = 0;
                if (ID(j) != TN_EMPTY) {
                    uint32_t cell;
                    if (j + 1 < nh && ID(j + 1) != TN_EMPTY && common_tet(s.hft[j], s.hft[j + 1], cell)) {
                        // common step: the next slot is a real face sharing a tetrahedron (off = 1, no swap)
                        if (fabsf(T(j) - T(j + 1)) >= TN_EPS) s.emitf[j] = 1;
                    } else {
                        const uint2 orig = s.hft[j];
                        float dn = T(j);
                        uint32_t real_off = 1;
                        for (uint32_t off = 1; j + off < nh && (real_off < 3 || ID(j + off) == TN_EMPTY || fabsf(T(j + off) - dn) < TN_EPS); ++off) {
                            if (ID(j + off) == TN_EMPTY) continue;
                            if (common_tet(orig, s.hft[j + off], cell)) {
                                if (fabsf(T(j) - T(j + off)) >= TN_EPS) s.emitf[j] = 1;
                                if (off > 1) {
                                    // swap(dl, first bary record, id) of slots j+off and j+1 (:244-250)
                                    const uint64_t k0 = s.key[j + off]; s.key[j + off] = s.key[j + 1]; s.key[j + 1] = k0;
                                    const float u0 = s.hu[j + off]; s.hu[j + off] = s.hu[j + 1]; s.hu[j + 1] = u0;
                                    const float v0 = s.hv[j + off]; s.hv[j + off] = s.hv[j + 1]; s.hv[j + 1] = v0;
                                    const uint2 f0 = s.hft[j + off]; s.hft[j + off] = s.hft[j + 1]; s.hft[j + 1] = f0;
                                    const uint8_t m0 = s.mark[j + off]; s.mark[j + off] = s.mark[j + 1]; s.mark[j + 1] = m0;
                                    touched = j + off > touched ? j + off : touched;
                                }
                                break;
                            }
                            dn = T(j + off);
                            real_off++;
                        }
                    }
                }
                ++j;
                if (j > touched) j = next_anomaly(j);   // slots j, j+1 are untouched originals: plain until the next anomaly
            }
        }
    }
    wave_sync();

    // parallel emission of the flagged pairs (slot j, slot j+1), in slot order
    uint32_t nseg = 0;
    struct FaceIds { uint32_t a[3], b[3]; bool emit; };
    auto face_ids = [&](uint32_t base) {   // vertex ids (stored order) of the two faces of the segment slot base + lane emits
        FaceIds r;
        const uint32_t j = base + lane;
        r.emit = j + 1 < nh && s.emitf[j] != 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) r.a[k] = r.b[k] = 0u;
        if (r.emit) {
            const uint32_t f0 = (uint32_t)s.key[j], f1 = (uint32_t)s.key[j + 1];
#pragma unroll
            for (int k = 0; k < 3; ++k) { r.a[k] = faces[3 * (size_t)f0 + k]; r.b[k] = faces[3 * (size_t)f1 + k]; }
        }
        return r;
    };
    FaceIds nxt = face_ids(0);
    for (uint32_t base = 0; base + 1 < nh; base += 64) {
        const uint32_t j = base + lane;
        const FaceIds cur = nxt;
        if (base + 65 < nh) nxt = face_ids(base + 64);   // requested before this iteration's rows are computed and stored
        const bool emit = cur.emit;
        const uint64_t m = __ballot(emit);
        if (emit) {
            const uint32_t slot = nseg + __popcll(m & lanemask_lt());
            const float t0 = __uint_as_float((uint32_t)(s.key[j] >> 32));
            const float t1 = __uint_as_float((uint32_t)(s.key[j + 1] >> 32));
            uint32_t cell = TN_EMPTY;
            common_tet(s.hft[j], s.hft[j + 1], cell);
            const uint32_t id1[3] = {cur.a[0], cur.a[1], cur.a[2]};
            const uint32_t id2[3] = {cur.b[0], cur.b[1], cur.b[2]};
            uint32_t vi[4];
            float b1[3], b2[3];
            combine_indices(id1, id2, s.hu[j], s.hv[j], s.hu[j + 1], s.hv[j + 1], vi, b1, b2);
            out_cells[slot] = cell;
            *reinterpret_cast<float2 *>(out_dist + 2 * (size_t)slot) = make_float2(t0, t1);
            float2 *bp = reinterpret_cast<float2 *>(out_bary + 6 * (size_t)slot);
            bp[0] = make_float2(b1[0], b1[1]);
            bp[1] = make_float2(b1[2], b2[0]);
            bp[2] = make_float2(b2[1], b2[2]);
            if (out_verts) *reinterpret_cast<uint4 *>(out_verts + 4 * (size_t)slot) = make_uint4(vi[0], vi[1], vi[2], vi[3]);
        }
        nseg += __popcll(m);
    }

    // tail fill: every remaining byte of the rows, coalesced (compact rows: nothing beyond the segments -- a 4096-ray
    // training batch otherwise writes 109 MB of constants that its consumers never read)
    for (uint32_t j = nseg + lane; j < (compact ? nseg : M); j += 64) {
        out_cells[j] = TN_EMPTY;
        *reinterpret_cast<float2 *>(out_dist + 2 * (size_t)j) = make_float2(0.f, 0.f);
        float2 *bp = reinterpret_cast<float2 *>(out_bary + 6 * (size_t)j);
        bp[0] = make_float2(0.f, 0.f); bp[1] = make_float2(0.f, 0.f); bp[2] = make_float2(0.f, 0.f);
        if (out_verts) *reinterpret_cast<uint4 *>(out_verts + 4 * (size_t)j) = make_uint4(TN_EMPTY, TN_EMPTY, TN_EMPTY, TN_EMPTY);
    }
    if (lane == 0) *out_num = nseg;
}

__device__ void sort_hits(const WaveSmem &s, uint32_t nh, int lane) {
    uint32_t Np = 1;
    while (Np < nh) Np <<= 1;
    for (uint32_t i = nh + lane; i < Np; i += 64) s.key[i] = ~0ull;
    wave_sync();
    for (uint32_t k = 2; k <= Np; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t tix = lane; tix < (Np >> 1); tix += 64) {
                // index of the lower element of the tix-th compare-exchange pair
                const uint32_t i = ((tix & ~(j - 1)) << 1) | (tix & (j - 1));
                const uint32_t ij = i | j;
                const uint64_t a = s.key[i], b = s.key[ij];
                const bool up = (i & k) == 0;
                if ((a > b) == up && a != b) {
                    s.key[i] = b; s.key[ij] = a;
                    const float ua = s.hu[i], ub = s.hu[ij]; s.hu[i] = ub; s.hu[ij] = ua;
                    const float va = s.hv[i], vb = s.hv[ij]; s.hv[i] = vb; s.hv[ij] = va;
                }
            }
            wave_sync();
        }
    }
}

// The hits the walk logged come in chain order: sorted on (t, face id) except where a gap is below the resolution of t
// (a tie, an inversion of a few ulps) -- which is why the ray is here.  Odd-even transposition passes until a whole pass
// swaps nothing: one pass when the order is already right, a few when neighbours are exchanged; the keys are distinct (a
// chain crosses a face once), so the result is the bitonic network's.  A chain that is still unsorted after 8 passes
// gets the network (option "literal_sort_passes", default 8; tests run 0 = always the network and 1).
__device__ void sort_logged_hits(const WaveSmem &s, uint32_t nh, int lane, uint32_t max_passes) {
    for (uint32_t pass = 0; pass < max_passes; ++pass) {
        bool swapped = false;
#pragma unroll
        for (uint32_t par = 0; par < 2; ++par) {
            for (uint32_t i = 2 * (uint32_t)lane + par; i + 1 < nh; i += 128) {
                const uint64_t a = s.key[i], b = s.key[i + 1];
                if (a > b) {
                    s.key[i] = b; s.key[i + 1] = a;
                    const float ua = s.hu[i], ub = s.hu[i + 1]; s.hu[i] = ub; s.hu[i + 1] = ua;
                    const float va = s.hv[i], vb = s.hv[i + 1]; s.hv[i] = vb; s.hv[i + 1] = va;
                    swapped = true;
                }
            }
            wave_sync();
        }
        if (__ballot(swapped) == 0ull) return;
    }
    sort_hits(s, nh, lane);
}

}  // namespace

// All faces of the mesh the ray (o, d) hits with 0 < t < 1e16, into the LDS hit arrays (unsorted).
// Returns the number kept (<= cap; beyond cap the nearest are kept).
// defer: on the first hit beyond `cap` stop the traversal (overflow = true) instead of keeping the nearest cap hits --
// the caller hands the ray to a launch with larger LDS arrays
__device__ uint32_t collect_hits(const WideBvh &bvh, WaveSmem &s, uint32_t M, uint32_t cap, float ox, float oy, float oz,
                                 float dx, float dy, float dz, unsigned long long *stats, int lane, bool &overflow,
                                 bool defer = false) {
    const RayPre rp = ray_pre(ox, oy, oz, dx, dy, dz);
    const float ix = safe_inv(dx), iy = safe_inv(dy), iz = safe_inv(dz);
    const float pad = 16.0f * 1.1920929e-7f * (fmaxf(fabsf(ox), fmaxf(fabsf(oy), fabsf(oz))) + bvh.scene_max);
        uint32_t nh = 0;      // hits stored (wave-uniform)
        overflow = false;
        bool aborted = false; // wave-uniform

        // One leaf: 64 triangles against the ray, hits appended to the LDS hit arrays.
        auto test_leaf = [&](float a0, float a1, float a2, float b0, float b1, float b2, float c0, float c1, float c2,
                             uint32_t fid) {
            const SV A = shear(rp, a0, a1, a2), B = shear(rp, b0, b1, b2), C = shear(rp, c0, c1, c2);
            float t = 0.f, u = 0.f, v = 0.f;
            const bool hit = (fid != TN_EMPTY) && tri_hit_sv(A, B, C, t, u, v);
            const uint64_t m = __ballot(hit);
            const uint32_t c = __popcll(m);
            if (c) {
                const uint64_t k = ((uint64_t)__float_as_uint(t) << 32) | fid;
                if (nh + c <= cap) {
                    if (hit) {
                        const uint32_t slot = nh + __popcll(m & lanemask_lt());
                        s.key[slot] = k; s.hu[slot] = u; s.hv[slot] = v;
                    }
                    nh += c;
                } else if (defer) {
                    overflow = true;
                    aborted = true;
                } else {
                    overflow = true;
                    wave_sync();
                    uint64_t mm = m;
                    while (mm) {
                        const int src = __ffsll((unsigned long long)mm) - 1;
                        mm &= mm - 1;
                        const uint64_t ks = __shfl(k, src);
                        const float us = __shfl(u, src), vs = __shfl(v, src);
                        if (nh < cap) {
                            if (lane == 0) { s.key[nh] = ks; s.hu[nh] = us; s.hv[nh] = vs; }
                            nh++;
                            wave_sync();
                        } else {
                            insert_overflow(s, cap, ks, us, vs, lane);
                        }
                    }
                }
            }
        };
        // Leaves whose box the line crosses are first collected (their order is irrelevant: the hits are
        // sorted afterwards), then tested in a loop that requests leaf i+1's triangles before testing leaf
        // i -- the dependent pop -> load -> test chain of a stack traversal becomes one load latency per
        // ray instead of one per leaf.  The list lives in the LDS area of `hft` (unused until the sort).
        uint32_t *leaf_list = reinterpret_cast<uint32_t *>(s.hft);
        const uint32_t leaf_cap = 2 * (M < 32 ? 32 : M);  // >= 64: one node's children always fit after a flush
        uint32_t nleaf = 0;
        auto run_leaves = [&]() {
            wave_sync();
            if (nleaf == 0) return;
            if (lane == 0 && stats) atomicAdd(&stats[23], (unsigned long long)nleaf);
            // one wave instruction tests G = 64 / leaf_w crossed leaves: lane = (leaf of the group, triangle slot)
            const uint32_t LW = bvh.leaf_w, LS = bvh.leaf_shift, G = 64u >> LS;
            const uint32_t sub = (uint32_t)lane >> LS, in = (uint32_t)lane & (LW - 1);
            const uint32_t ngroups = (nleaf + G - 1) / G;
            auto fetch = [&](uint32_t gi, float (&d)[9], uint32_t &fid) {
                const uint32_t e = gi * G + sub;
                fid = TN_EMPTY;
#pragma unroll
                for (int k = 0; k < 9; ++k) d[k] = 0.f;
                if (e < nleaf) {
                    const size_t li = leaf_list[e];
                    const float *tr = bvh.leaf_tri + li * (9 * (size_t)LW) + in;
#pragma unroll
                    for (int k = 0; k < 9; ++k) d[k] = tr[k * LW];
                    fid = bvh.leaf_id[li * LW + in];
                }
            };
            // three leaves deep: leaf i + 2 is requested before leaf i is tested (a small batch is one wavefront per ray
            // with every ray resident at once, so a ray's own chain of round trips is the time of the call)
            float cur[9], nx1[9], nx2[9];
            uint32_t cfid, f1 = TN_EMPTY, f2 = TN_EMPTY;
            fetch(0, cur, cfid);
            if (ngroups > 1) fetch(1, nx1, f1);
            for (uint32_t i = 0; i < ngroups && !aborted; ++i) {
                if (i + 2 < ngroups) fetch(i + 2, nx2, f2);
                test_leaf(cur[0], cur[1], cur[2], cur[3], cur[4], cur[5], cur[6], cur[7], cur[8], cfid);
#pragma unroll
                for (int k = 0; k < 9; ++k) { cur[k] = nx1[k]; nx1[k] = nx2[k]; }
                cfid = f1; f1 = f2;
            }
            nleaf = 0;
            wave_sync();
        };

        // Internal nodes: the NEXT node is popped and its boxes requested before the current one is tested (the visiting
        // order is irrelevant: every crossed leaf is wanted), so a node costs one round trip per two instead of one each.
        auto load_node = [&](uint32_t idx, float (&bx)[6], uint32_t &ch) {
            const float *b = bvh.boxes + (size_t)idx * (6 * WIDE) + lane;
#pragma unroll
            for (int k = 0; k < 6; ++k) bx[k] = b[k * WIDE];
            ch = bvh.child[(size_t)idx * WIDE + lane];
        };
        uint32_t sp = 0;      // stack of internal nodes (wave-uniform)
        float cbx[6], nbx[6];
        uint32_t cch, nch = TN_EMPTY;
        load_node(0u, cbx, cch);   // root = internal node 0
        bool have_cur = true;
        while (have_cur && !aborted) {
            bool have_nxt = sp > 0;
            if (have_nxt) {
                const uint32_t nidx = s.stack[sp - 1];
                sp--;
                wave_sync();  // everyone has read the top before it can be overwritten
                load_node(nidx, nbx, nch);
            }
            if (lane == 0 && stats) atomicAdd(&stats[22], 1ull);
            const bool hit = cch != TN_EMPTY && line_box(ox, oy, oz, ix, iy, iz, cbx[0], cbx[1], cbx[2], cbx[3], cbx[4], cbx[5], pad);
            const bool to_leaf = hit && (cch >> 31) != 0, to_node = hit && (cch >> 31) == 0;
            const uint64_t ml = __ballot(to_leaf), mn = __ballot(to_node);
            if (nleaf + __popcll(ml) > leaf_cap) run_leaves();  // flush a full list (tiny M only)
            if (to_leaf) leaf_list[nleaf + __popcll(ml & lanemask_lt())] = cch & 0x7FFFFFFFu;
            nleaf += __popcll(ml);
            if (to_node) s.stack[sp + __popcll(mn & lanemask_lt())] = cch;
            sp += __popcll(mn);
            wave_sync();
            if (!have_nxt && sp > 0) {   // nothing was waiting: take one of the children just pushed
                const uint32_t nidx = s.stack[sp - 1];
                sp--;
                wave_sync();
                load_node(nidx, nbx, nch);
                have_nxt = true;
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) cbx[k] = nbx[k];
            cch = nch;
            have_cur = have_nxt;
        }
        if (!aborted) run_leaves();
        wave_sync();
        return nh;
}

#if TN_WALK_DIAG
__device__ unsigned long long g_gen_time[8];   // 100 MHz ticks: [0] rays, [1] traversal + leaf tests, [2] sort, [3] pairing + row write, [4] nodes, [5] leaves, [6] hits
extern "C" int tn_debug_general_time(unsigned long long out[8], int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gen_time), sizeof(unsigned long long) * 8) != hipSuccess) return 1;
    if (reset) { unsigned long long z[8] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_gen_time), z, sizeof z) != hipSuccess) return 1; }
    return 0;
}
#endif
__global__ __launch_bounds__(64) void k_trace_general(TraceParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    const uint32_t M = p.M;
    // LDS arrays of C <= M entries (lds_cap): a small batch is latency-bound and wants every ray resident at once -- at
    // M = 512 the full arrays (14.8 KB) admit 11 wavefronts per CU, 256-entry arrays (8 KB) admit 20.  A ray with more
    // than C - 1 hits is handed to the overflow list (a second launch with full arrays).
    const uint32_t C = p.lds_cap && p.lds_cap < M ? p.lds_cap : M;
    const bool defer = C < M;
    WaveSmem s = carve(smem, C);
    const uint32_t cap = C - 1;  // at most M-1 hits are kept (optix_trace_rays.cu:312-315)

    const size_t n_items = p.item_count ? (size_t)*p.item_count : p.num_items;
    for (size_t it = blockIdx.x; it < n_items; it += gridDim.x) {
        const size_t ray = p.ray_list ? (size_t)p.ray_list[it] : it;
        bool overflow = false;
#if TN_WALK_DIAG
        const unsigned long long tk0 = wall_clock64();
#endif
        uint32_t nh = collect_hits(p.bvh, s, C, cap, p.origins[3 * ray], p.origins[3 * ray + 1], p.origins[3 * ray + 2],
                                   p.dirs[3 * ray], p.dirs[3 * ray + 1], p.dirs[3 * ray + 2],
                                   nullptr,
                                   lane, overflow, defer);
#if TN_WALK_DIAG
        const unsigned long long tk1 = wall_clock64();
#endif
        if (overflow && defer) {
            if (lane == 0) p.overflow_list[atomicAdd(p.overflow_count, 1u)] = (uint32_t)ray;
            continue;
        }
        if (overflow && lane == 0 && p.stats) atomicAdd(&p.stats[3], 1ull);

        sort_hits(s, nh, lane);
#if TN_WALK_DIAG
        const unsigned long long tk2 = wall_clock64();
#endif
        postprocess_and_write(s, nh, M, p.faces, p.face_tets, p.out_num + ray, p.out_cells + ray * M,
                              p.out_bary + ray * M * 6, p.out_dist + ray * M * 2,
                              p.out_verts ? p.out_verts + ray * M * 4 : nullptr, p.stats, lane, p.compact_rows != 0);
        wave_sync();
#if TN_WALK_DIAG
        if (lane == 0) {
            const unsigned long long tk3 = wall_clock64();
            atomicAdd(&g_gen_time[0], 1ull); atomicAdd(&g_gen_time[1], tk1 - tk0); atomicAdd(&g_gen_time[2], tk2 - tk1);
            atomicAdd(&g_gen_time[3], tk3 - tk2); atomicAdd(&g_gen_time[6], (unsigned long long)nh);
        }
#endif
    }
}

// post-process only (test aid, tn_postprocess_hits): rows of caller-supplied sorted hits
__global__ __launch_bounds__(64) void k_postprocess_hits(TraceParams p, const uint32_t *hit_count,
                                                         const uint32_t *hit_ids, const float *hit_t,
                                                         const float *hit_uv) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    WaveSmem s = carve(smem, p.M);
    const int lane = threadIdx.x;
    const uint32_t M = p.M;
    for (size_t ray = blockIdx.x; ray < p.num_items; ray += gridDim.x) {
        const uint32_t nh = hit_count[ray] < M ? hit_count[ray] : M;
        for (uint32_t j = lane; j < nh; j += 64) {
            s.key[j] = ((uint64_t)__float_as_uint(hit_t[ray * M + j]) << 32) | hit_ids[ray * M + j];
            s.hu[j] = hit_uv[2 * (ray * M + j)];
            s.hv[j] = hit_uv[2 * (ray * M + j) + 1];
        }
        wave_sync();
        postprocess_and_write(s, nh, M, p.faces, p.face_tets, p.out_num + ray, p.out_cells + ray * M,
                              p.out_bary + ray * M * 6, p.out_dist + ray * M * 2,
                              p.out_verts ? p.out_verts + ray * M * 4 : nullptr, p.stats, lane);
        wave_sync();
    }
}

// Literal sort + pairing of the hits the walk LOGGED for the rays whose chain is sound but whose order it does not
// certify (a gap below eps, a tie, an inversion): hit k of launch item r is the 16-byte log entry
// ((r / 64) * M + k) * 64 + r % 64 = {t, u, v, variant | exit << 30}; its face id is WalkFid::fid[exit] of the variant
// (exit code 3: the entry hull face, id in the low bits).  The chain's faces are the ray's all-hits set (two hull
// crossings, two crossed faces per tet, no zero edge function), so this equals the BVH path -- sort on (t, face id),
// then the reference's phases literally (optix_trace_rays.cu:110-266) -- without a traversal.
__global__ __launch_bounds__(64) void k_postprocess_log(TraceParams p, const WalkFid *__restrict__ fidt,
                                                        const uint4 *__restrict__ hit_log,
                                                        const uint2 *__restrict__ literal_list,
                                                        const uint32_t *__restrict__ literal_count) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    WaveSmem s = carve(smem, p.M);
    const int lane = threadIdx.x;
    const uint32_t M = p.M;
    const size_t n_items = *literal_count;
    for (size_t it = blockIdx.x; it < n_items; it += gridDim.x) {
        const uint2 ent = literal_list[it];
        const size_t ray = ent.x;
        const uint32_t nh = ent.y < M ? ent.y : M - 1;
        const uint4 *lg = hit_log + (ray >> 6) * (size_t)M * 64 + (ray & 63);
        // two dependent random reads per hit (log entry -> face id of its variant): a lane's eight log entries are
        // requested together, then its eight face ids, then everything goes to LDS
        for (uint32_t base = 0; base < nh; base += 512) {
            uint4 e[8];
            uint32_t fid[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const uint32_t j = base + 64 * q + lane;
                if (j < nh) e[q] = lg[(size_t)j * 64];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const uint32_t j = base + 64 * q + lane;
                if (j < nh) {
                    const uint32_t x = e[q].w >> 30, lo = e[q].w & 0x3FFFFFFFu;
                    fid[q] = x == 3u ? lo : fidt[lo].fid[x];
                }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const uint32_t j = base + 64 * q + lane;
                if (j < nh) {
                    s.key[j] = ((uint64_t)e[q].x << 32) | fid[q];
                    s.hu[j] = __uint_as_float(e[q].y);
                    s.hv[j] = __uint_as_float(e[q].z);
                }
            }
        }
        wave_sync();
        sort_logged_hits(s, nh, lane, p.sort_passes);
        postprocess_and_write(s, nh, M, p.faces, p.face_tets, p.out_num + ray, p.out_cells + ray * M,
                              p.out_bary + ray * M * 6, p.out_dist + ray * M * 2,
                              p.out_verts ? p.out_verts + ray * M * 4 : nullptr, p.stats, lane, p.compact_rows != 0);
        if (lane == 0 && p.stats) atomicAdd(&p.stats[4 + 13], 1ull);
        wave_sync();
    }
}

// Global cross-check of the walk's certification (option "verify_stride", tests / fuzzing / paranoid callers): the
// walk only sees the connected component of crossed faces that contains the two hull faces; a second component
// (the star of a vertex whose rounded projection lands exactly on the ray while the chain passes
// beside it) is invisible to it and only excluded by the vertex-proximity rule (reason 4).  The BVH all-hits traversal
// sees EVERY face the triangle routine accepts, so for every stride-th ray the walk certified the number of faces the
// BVH finds (count only: no sort, no pairing) must equal the number of hits the walk logged.  A mismatch is counted
// (reason 14) and the ray is handed to the BVH kernel like any other fallback ray -- the kernels that write rows are
// launched after this one.  reason 15 counts the rays checked.
// LATE form (late != 0; the default schedule of a one-chunk call, tn_api.hip): the check runs on a side stream BESIDE
// the segment writer and the tail fill instead of in front of them, so walk_n is left alone (the row of a mismatching ray is
// written as certified) and the ray goes to a list of its own, which one more BVH launch re-traces -- whole rows -- after
// everything else of the call has been joined.
__global__ __launch_bounds__(64) void k_verify_counts(TraceParams p, uint32_t stride, uint32_t *__restrict__ walk_n,
                                                      uint32_t *__restrict__ fallback_list, uint32_t *__restrict__ fallback_count,
                                                      size_t ray_base, uint32_t late, uint32_t inject,
                                                      const uint32_t *__restrict__ ray_list, const uint32_t *__restrict__ list_count) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    WaveSmem s = carve(smem, p.M);
    const int lane = threadIdx.x;
    // every stride-th ray (the blind sample), or exactly the rays of a list (the walk's risk list: its count lives on the device)
    const size_t n_checks = ray_list ? (size_t)*list_count : (p.num_items + stride - 1) / stride;
    const int c_checked = ray_list ? 26 : 4 + 15, c_bad = ray_list ? 27 : 4 + 14;
    for (size_t it = blockIdx.x; it < n_checks; it += gridDim.x) {
        const size_t ray = ray_list ? (size_t)ray_list[it] : it * stride;
        if (ray_list && stride && ray % stride == 0) continue;   // the blind sample checks (and counts) this one
        const uint32_t wn_raw = walk_n[ray];
        if (wn_raw == TN_EMPTY) continue;      // literal / fallback ray: not certified, nothing to verify
        const uint32_t wn = wn_raw & 0x3FFFFFFFu;   // (bit 30: the walk's rule C flag for the segment writer)
        bool overflow = false;
        const uint32_t nh = collect_hits(p.bvh, s, p.M, p.M - 1, p.origins[3 * ray], p.origins[3 * ray + 1], p.origins[3 * ray + 2],
                                         p.dirs[3 * ray], p.dirs[3 * ray + 1], p.dirs[3 * ray + 2], nullptr, lane, overflow);
        if (lane == 0) {
            if (p.stats) atomicAdd(&p.stats[c_checked], 1ull);
            // inject (tests): every checked ray is treated as a mismatch, so that the hand-over -- incl. the late form's
            // re-trace of rows the writer and the fills have already written -- is exercised although no real mismatch exists
            if (nh != wn || overflow || inject) {
                if (!late) walk_n[ray] = TN_EMPTY;   // the segment writer and the fill skip the row: the BVH kernel writes it
                fallback_list[atomicAdd(fallback_count, 1u)] = (uint32_t)(ray_base + ray);
                if (p.stats) atomicAdd(&p.stats[c_bad], 1ull);
            }
        }
        wave_sync();
    }
}

size_t trace_general_smem_bytes(uint32_t M) {
    return (size_t)M * (8 + 4 + 4) + (size_t)(M < 32 ? 32 : M) * 8 + STACK_CAP * 4 + 2 * (size_t)M;
}

// M = 4096 asks for ~108 KB of dynamic LDS per wave: beyond the 64 KB a kernel gets without the attribute
template <typename K>
static size_t wave_smem(K kernel, uint32_t M) {
    const size_t bytes = trace_general_smem_bytes(M);
    if (bytes > 64 * 1024) allow_dynamic_lds(reinterpret_cast<const void *>(kernel), bytes);
    return bytes;
}


// trace_rays_triangles: the sorted all-hits list itself, no pairing
// (src/optix/optix_trace_rays_triangles.cu:50-87).  Slots >= count: 0 in every array -- what the reference's
// torch::zeros outputs hold there (py_binding.cpp:90-94; it never writes them apart from sort padding).
__global__ __launch_bounds__(64) void k_trace_triangles(TraceParams p, uint32_t *out_ids, float *out_t, float *out_uv,
                                                        uint32_t *out_v3) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    WaveSmem s = carve(smem, p.M);
    const int lane = threadIdx.x;
    const uint32_t M = p.M;
    for (size_t ray = blockIdx.x; ray < p.num_items; ray += gridDim.x) {
        bool overflow = false;
        const uint32_t nh = collect_hits(p.bvh, s, M, M - 1, p.origins[3 * ray], p.origins[3 * ray + 1], p.origins[3 * ray + 2],
                                         p.dirs[3 * ray], p.dirs[3 * ray + 1], p.dirs[3 * ray + 2], nullptr, lane, overflow);
        sort_hits(s, nh, lane);
        for (uint32_t j = lane; j < M; j += 64) {
            const bool ok = j < nh;
            const uint32_t id = ok ? (uint32_t)s.key[j] : 0u;
            const size_t q = ray * M + j;
            out_ids[q] = id;
            out_t[q] = ok ? __uint_as_float((uint32_t)(s.key[j] >> 32)) : 0.f;
            out_uv[2 * q] = ok ? s.hu[j] : 0.f;
            out_uv[2 * q + 1] = ok ? s.hv[j] : 0.f;
            for (int k = 0; k < 3; ++k) out_v3[3 * q + k] = ok ? p.faces[3 * (size_t)id + k] : 0u;
        }
        if (lane == 0) p.out_num[ray] = nh;
        wave_sync();
    }
}

// find_tetrahedra: closest face hit along +x and -x, common tetrahedron, blended barycentrics
// (src/optix/optix_find_tetrahedra.cu:84-212).  Closest = smallest (t, face id).
__global__ __launch_bounds__(64) void k_find_tetrahedra(TraceParams p, const float *__restrict__ points, uint32_t *out_tet,
                                                        float *out_bary, uint32_t *out_verts) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    WaveSmem s = carve(smem, p.M);
    const int lane = threadIdx.x;
    const uint32_t M = p.M;
    for (size_t pt = blockIdx.x; pt < p.num_items; pt += gridDim.x) {
        const float ox = points[3 * pt], oy = points[3 * pt + 1], oz = points[3 * pt + 2];
        uint32_t fid[2] = {TN_EMPTY, TN_EMPTY};
        float ht[2] = {0.f, 0.f}, hu[2] = {0.f, 0.f}, hv[2] = {0.f, 0.f};
        for (int side = 0; side < 2; ++side) {
            bool overflow = false;
            const uint32_t nh = collect_hits(p.bvh, s, M, M - 1, ox, oy, oz, side == 0 ? 1.0f : -1.0f, 0.0f, 0.0f, nullptr, lane, overflow);
            wave_sync();
            uint64_t best = ~0ull;
            uint32_t bidx = 0;
            for (uint32_t j = lane; j < nh; j += 64) if (s.key[j] < best) { best = s.key[j]; bidx = j; }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const uint64_t ob = __shfl_xor(best, off);
                const uint32_t oi = __shfl_xor(bidx, off);
                if (ob < best) { best = ob; bidx = oi; }
            }
            if (nh) {
                fid[side] = (uint32_t)best; ht[side] = __uint_as_float((uint32_t)(best >> 32));
                hu[side] = s.hu[bidx]; hv[side] = s.hv[bidx];
            }
            wave_sync();
        }
        if (lane == 0) {
            uint32_t cell = TN_EMPTY;
            uint32_t vi[4] = {0, 0, 0, 0};
            float c[3] = {0.f, 0.f, 0.f};
            if (fid[0] != TN_EMPTY && fid[1] != TN_EMPTY) {
                const uint2 t0 = *reinterpret_cast<const uint2 *>(p.face_tets + 2 * (size_t)fid[0]);
                const uint2 t1 = *reinterpret_cast<const uint2 *>(p.face_tets + 2 * (size_t)fid[1]);
                if (common_tet(t0, t1, cell)) {
                    const uint32_t id1[3] = {p.faces[3 * (size_t)fid[0]], p.faces[3 * (size_t)fid[0] + 1], p.faces[3 * (size_t)fid[0] + 2]};
                    const uint32_t id2[3] = {p.faces[3 * (size_t)fid[1]], p.faces[3 * (size_t)fid[1] + 1], p.faces[3 * (size_t)fid[1] + 2]};
                    float c0[3], c1[3];
                    combine_indices(id1, id2, hu[0], hv[0], hu[1], hv[1], vi, c0, c1);
                    const float m = ht[1] / (ht[0] + ht[1]);
                    for (int k = 0; k < 3; ++k) c[k] = c0[k] * m + c1[k] * (1 - m);
                } else {
                    cell = TN_EMPTY;
                }
            }
            out_tet[pt] = cell;
            for (int k = 0; k < 3; ++k) out_bary[3 * pt + k] = c[k];
            for (int k = 0; k < 4; ++k) out_verts[4 * pt + k] = vi[k];
        }
        wave_sync();
    }
}

void launch_trace_triangles(const TraceParams &p, uint32_t *out_ids, float *out_t, float *out_uv, uint32_t *out_v3,
                            hipStream_t stream) {
    if (p.num_items == 0) return;
    const size_t max_blocks = 256 * 16;
    const unsigned grid = (unsigned)(p.num_items < max_blocks ? p.num_items : max_blocks);
    hipLaunchKernelGGL(k_trace_triangles, dim3(grid), dim3(64), wave_smem(k_trace_triangles, p.M), stream, p, out_ids, out_t, out_uv, out_v3);
}

void launch_find_tetrahedra(const TraceParams &p, const float *points, uint32_t *out_tet, float *out_bary,
                            uint32_t *out_verts, hipStream_t stream) {
    if (p.num_items == 0) return;
    const size_t max_blocks = 256 * 16;
    const unsigned grid = (unsigned)(p.num_items < max_blocks ? p.num_items : max_blocks);
    hipLaunchKernelGGL(k_find_tetrahedra, dim3(grid), dim3(64), wave_smem(k_find_tetrahedra, p.M), stream, p, points, out_tet, out_bary, out_verts);
}

void launch_trace_general(const TraceParams &p, hipStream_t stream) {
    if (p.num_items == 0) return;
    const size_t smem = wave_smem(k_trace_general, p.lds_cap && p.lds_cap < p.M ? p.lds_cap : p.M);
    const size_t max_blocks = 256 * 16;
    const unsigned grid = (unsigned)(p.num_items < max_blocks ? p.num_items : max_blocks);
    hipLaunchKernelGGL(k_trace_general, dim3(grid), dim3(64), smem, stream, p);
}

void launch_verify_counts(const TraceParams &p, uint32_t stride, uint32_t *walk_n, uint32_t *fallback_list, uint32_t *fallback_count,
                          size_t ray_base, hipStream_t stream, bool late, bool inject, const uint32_t *ray_list,
                          const uint32_t *list_count, size_t max_list) {
    if (p.num_items == 0 || (stride == 0 && !ray_list)) return;
    const size_t smem = wave_smem(k_verify_counts, p.M);
    // (the risk list is short -- a fraction of a per cent of the rays -- but its length is only known on the device: the grid
    //  of the blind sample strides over it; blocks beyond the list exit at once)
    const size_t n_checks = ray_list ? std::min<size_t>(max_list, 256 * 16) : (p.num_items + stride - 1) / stride, max_blocks = 256 * 16;
    if (n_checks == 0) return;
    hipLaunchKernelGGL(k_verify_counts, dim3((unsigned)(n_checks < max_blocks ? n_checks : max_blocks)), dim3(64), smem, stream, p, stride,
                       walk_n, fallback_list, fallback_count, ray_base, late ? 1u : 0u, inject ? 1u : 0u, ray_list, list_count);
}

void launch_postprocess_log(const TraceParams &p, const WalkFid *fidt, const uint4 *hit_log, const uint2 *literal_list,
                            const uint32_t *literal_count, size_t max_items, hipStream_t stream) {
    if (max_items == 0) return;
    const size_t max_blocks = 256 * 16;
    const unsigned grid = (unsigned)(max_items < max_blocks ? max_items : max_blocks);
    hipLaunchKernelGGL(k_postprocess_log, dim3(grid), dim3(64), wave_smem(k_postprocess_log, p.M), stream, p, fidt, hit_log,
                       literal_list, literal_count);
}

void launch_postprocess_hits(const TraceParams &p, const uint32_t *hit_count, const uint32_t *hit_ids,
                             const float *hit_t, const float *hit_uv, hipStream_t stream) {
    if (p.num_items == 0) return;
    const size_t smem = wave_smem(k_postprocess_hits, p.M);
    const size_t max_blocks = 256 * 16;
    const unsigned grid = (unsigned)(p.num_items < max_blocks ? p.num_items : max_blocks);
    hipLaunchKernelGGL(k_postprocess_hits, dim3(grid), dim3(64), smem, stream, p, hit_count, hit_ids, hit_t, hit_uv);
}

}  // namespace tn
