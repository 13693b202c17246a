// tn_build.hip -- load_tetrahedra's structures built ON THE DEVICE.
//
// The reference builds its face table with a single-threaded std::unordered_map after a blocking D2H of the mesh and
// hands the triangles to optixAccelBuild (src/tetrahedra_tracer.cpp:21-71, 244-340).  Round 1/2a did the equivalent
// on one host core (tn_mesh.cpp: 0.25 s at 300k tets, 1.0 s at 1M).  Here the mesh never leaves the device:
//
//   face table   every (tet, local face) "sighting" goes into an open-addressing table keyed by its sorted vertex
//                triple (atomicCAS on the slot; the second sighting of a key pairs up with the slot's owner; a third
//                is the reference's "shared by more than two tetrahedra" error).  A face's FIRST sighting is the
//                smaller sighting index of its pair, so an exclusive scan over "is first sighting" numbers the faces
//                in exactly the reference's first-seen order -- no sort, and independent of the thread schedule.
//   walk records Morton codes of the tet centroids (same arithmetic as the host build) -> stable radix sort ->
//                one thread per (record, entry face) derives the 64-byte WalkVar through the SAME function the host
//                build uses (core::make_walk_var): the array is bit-identical to tn_mesh.cpp's.
//   hull tree    the few hundred hull faces are compacted (face-id order), downloaded (48 B each) and threaded by the
//                host routine (build_hull_from_info) -- identical by construction.
//   face BVH     the median-split tree's SHAPE depends on the face count only, so the host lays out the nodes and the
//                device fills them: per level one segmented reduction (centroid bounds -> split axis per segment), one
//                radix sort of (segment, coordinate) keys; boxes bottom-up; 64-wide collapse level by level with
//                core::collapse_node (the host build's greedy opening, one thread per wide node).  Same tree as the
//                host build up to ties in the median splits and the numbering of the wide nodes.
//
// Blocking like the reference's load (a few small D2H reads: counts, the hull faces, the child rows for the traversal
// stack bound); rocPRIM provides the radix sort and the scan.
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "tn_build.h"

namespace tn {

namespace {

constexpr int BT = 256;
constexpr int MAX_WIDE_LEVELS = 16;
inline unsigned grid_for(size_t n) { return (unsigned)((n + BT - 1) / BT); }
__device__ __forceinline__ size_t gid() { return (size_t)blockIdx.x * blockDim.x + threadIdx.x; }

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off));
    return v;
}

// vertex ids in range?  max |coordinate| over the referenced vertices (the box padding of the trace kernels)
__global__ __launch_bounds__(BT) void k_cells_check_max(size_t n4, const uint32_t *__restrict__ cells, uint32_t V,
                                                        const float *__restrict__ xyz, uint32_t *flags, uint32_t *smax_bits) {
    const size_t i = gid();
    float m = 0.f;
    if (i < n4) {
        const uint32_t v = cells[i];
        if (v >= V) atomicOr(flags, core::FLAG_CELL_OOB);
        else m = fmaxf(fabsf(xyz[3 * (size_t)v]), fmaxf(fabsf(xyz[3 * (size_t)v + 1]), fabsf(xyz[3 * (size_t)v + 2])));
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(smax_bits, __float_as_uint(m));   // non-negative floats order as uints
}

__global__ __launch_bounds__(BT) void k_face_hash(size_t n4, const uint32_t *__restrict__ cells, uint32_t *slot, uint64_t cap_mask,
                                                  uint32_t *partner, uint32_t *flags) {
    const size_t i = gid();
    if (i < n4) core::face_hash_insert((uint32_t)i, cells, slot, cap_mask, partner, flags);
}

__global__ __launch_bounds__(BT) void k_face_first(size_t n4, const uint32_t *__restrict__ partner, uint32_t *__restrict__ first) {
    const size_t i = gid();
    if (i < n4) first[i] = core::face_is_first((uint32_t)i, partner) ? 1u : 0u;
}

__global__ __launch_bounds__(BT) void k_face_emit(size_t n4, const uint32_t *__restrict__ cells, const uint32_t *__restrict__ partner,
                                                  const uint32_t *__restrict__ first, const uint32_t *__restrict__ fidx,
                                                  uint32_t *faces, uint32_t *face_tets, uint32_t *tet_face) {
    const size_t i = gid();
    if (i < n4 && first[i]) core::face_emit((uint32_t)i, fidx[i], cells, partner, faces, face_tets, tet_face);
}

__global__ __launch_bounds__(BT) void k_hull_flag(size_t F, const uint32_t *__restrict__ face_tets, uint32_t *__restrict__ hflag) {
    const size_t f = gid();
    if (f < F) hflag[f] = face_tets[2 * f + 1] == TN_EMPTY ? 1u : 0u;
}
__global__ __launch_bounds__(BT) void k_hull_info(size_t F, const uint32_t *__restrict__ hflag, const uint32_t *__restrict__ hidx,
                                                  const uint32_t *__restrict__ faces, const uint32_t *__restrict__ face_tets,
                                                  const uint32_t *__restrict__ tet_face, const uint32_t *__restrict__ rec_of_tet,
                                                  const float *__restrict__ xyz, uint32_t *info, uint32_t *flags) {
    const size_t f = gid();
    if (f < F && hflag[f]) core::hull_face_info((uint32_t)f, faces, face_tets, tet_face, rec_of_tet, xyz, info + 12 * (size_t)hidx[f], flags);
}

// bounds[0..2] = min, bounds[3..5] = max, as order-preserving uints
__global__ void k_init_bounds(uint32_t *bounds, size_t nseg) {
    const size_t i = gid();
    if (i < nseg * 6) bounds[i] = (i % 6) < 3 ? core::float_ordered(INFINITY) : core::float_ordered(-INFINITY);
}
__device__ __forceinline__ void bounds_update(uint32_t *b, const float c[3], bool active, bool uniform) {
    if (uniform) {   // the whole wave updates the same segment: reduce first, 6 atomics per wave
        float lo[3], hi[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) { lo[a] = wave_min(c[a]); hi[a] = wave_max(c[a]); }
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int a = 0; a < 3; ++a) { atomicMin(b + a, core::float_ordered(lo[a])); atomicMax(b + 3 + a, core::float_ordered(hi[a])); }
        }
    } else if (active) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { atomicMin(b + a, core::float_ordered(c[a])); atomicMax(b + 3 + a, core::float_ordered(c[a])); }
    }
}

__global__ __launch_bounds__(BT) void k_tet_bounds(size_t T, const uint32_t *__restrict__ cells, const float *__restrict__ xyz, uint32_t *bounds) {
    const size_t i = gid();
    const bool active = i < T;
    float c[3] = {0.f, 0.f, 0.f};
    if (active) core::tet_centroid((uint32_t)i, cells, xyz, c);
    const bool uniform = __ballot(!active) == 0ull;
    bounds_update(bounds, c, active, uniform);
}
__global__ __launch_bounds__(BT) void k_tet_codes(size_t T, const uint32_t *__restrict__ cells, const float *__restrict__ xyz,
                                                  const uint32_t *__restrict__ bounds, uint64_t *codes, uint32_t *idx) {
    const size_t i = gid();
    if (i >= T) return;
    float c[3], lo[3], hi[3];
    core::tet_centroid((uint32_t)i, cells, xyz, c);
    for (int a = 0; a < 3; ++a) { lo[a] = core::ordered_float(bounds[a]); hi[a] = core::ordered_float(bounds[3 + a]); }
    codes[i] = core::morton63(c, lo, hi);
    idx[i] = (uint32_t)i;
}
__global__ __launch_bounds__(BT) void k_inverse(size_t n, const uint32_t *__restrict__ order, uint32_t *__restrict__ inv) {
    const size_t r = gid();
    if (r < n) inv[order[r]] = (uint32_t)r;
}
__global__ __launch_bounds__(BT) void k_iota(size_t n, uint32_t *p) {
    const size_t i = gid();
    if (i < n) p[i] = (uint32_t)i;
}

__global__ __launch_bounds__(BT) void k_walk_vars(size_t n4, const uint32_t *__restrict__ order, const uint32_t *__restrict__ rec_of_tet,
                                                  const uint32_t *__restrict__ cells, const float *__restrict__ xyz,
                                                  const uint32_t *__restrict__ tet_face, const uint32_t *__restrict__ faces,
                                                  const uint32_t *__restrict__ face_tets, WalkVar *vars, uint32_t *flags) {
    const size_t i = gid();
    if (i >= n4) return;
    const WalkVar v = core::walk_var_of((uint32_t)(i >> 2), (uint32_t)(i & 3), order, rec_of_tet, cells, xyz, tet_face, faces, face_tets, flags);
    const uint4 *src = reinterpret_cast<const uint4 *>(&v);
    uint4 *dst = reinterpret_cast<uint4 *>(vars + i);
    dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3];
}

// thin-neighbourhood exponent of every tet (tn_build_core.h): star minima of the smallest tet height, then the patch of
// the four walk records of the tet
__global__ __launch_bounds__(BT) void k_fill_u32(size_t n, uint32_t *p, uint32_t v) {
    const size_t i = gid();
    if (i < n) p[i] = v;
}
__global__ __launch_bounds__(BT) void k_tet_thin(size_t T, const uint32_t *__restrict__ cells, const float *__restrict__ xyz, uint32_t *vmin) {
    const size_t i = gid();
    if (i >= T) return;
    const uint32_t *c = cells + 4 * i;
    float p[4][3];
    for (int k = 0; k < 4; ++k) for (int a = 0; a < 3; ++a) p[k][a] = xyz[3 * (size_t)c[k] + a];
    const uint32_t bits = core::tet_min_height_bits(p);
    for (int k = 0; k < 4; ++k) core::atomic_min_u32(vmin + c[k], bits);
}
__global__ __launch_bounds__(BT) void k_thin_patch(size_t T, const uint32_t *__restrict__ cells, const uint32_t *__restrict__ vmin,
                                                   const uint32_t *__restrict__ rec_of_tet, WalkVar *vars) {
    const size_t i = gid();
    if (i >= T) return;
    const uint32_t *c = cells + 4 * i;
    const uint32_t e = core::thin_exponent(vmin[c[0]], vmin[c[1]], vmin[c[2]], vmin[c[3]]);
    for (uint32_t k = 0; k < 4; ++k) vars[4 * (size_t)rec_of_tet[i] + k].code_hi |= e << core::THIN_SHIFT;
}

// ------------------------------------------------------------------ face BVH
__global__ __launch_bounds__(BT) void k_face_boxes(size_t F, const uint32_t *__restrict__ faces, const float *__restrict__ xyz, float *fb, float *cen) {
    const size_t f = gid();
    if (f < F) core::face_box((uint32_t)f, faces, xyz, fb + 6 * f, cen + 3 * f);
}

// position i -> its segment of this level, from its segment of the previous level (child_first / split_pos per previous
// segment: the first of its one or two successors and the position where the second one starts, ~0 if it did not split;
// both null at level 0: everything is segment 0); centroid bounds per segment
__global__ __launch_bounds__(BT) void k_seg_bounds(size_t n, const uint32_t *__restrict__ order, const uint32_t *__restrict__ child_first,
                                                   const uint32_t *__restrict__ split_pos, const float *__restrict__ cen,
                                                   uint32_t *__restrict__ seg_of, uint32_t *segb) {
    const size_t i = gid();
    const bool active = i < n;
    uint32_t s = TN_EMPTY;
    float c[3] = {0.f, 0.f, 0.f};
    if (active) {
        s = 0;
        if (child_first) {
            const uint32_t sp = seg_of[i];
            s = child_first[sp] + ((uint32_t)i >= split_pos[sp] ? 1u : 0u);
        }
        seg_of[i] = s;
        const size_t f = order[i];
        c[0] = cen[3 * f]; c[1] = cen[3 * f + 1]; c[2] = cen[3 * f + 2];
    }
    // segmented reduction over the wave: a segment is a run of consecutive positions, so its lanes are consecutive.
    // Inclusive segmented min / max scan by shuffles; the last lane of each run adds the run's result with 6 atomics
    // (2-3 runs per wave at the deep levels instead of 64 x 6 same-address atomics).
    const int lane = threadIdx.x & 63;
    float lo[3] = {c[0], c[1], c[2]}, hi[3] = {c[0], c[1], c[2]};
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t so = (uint32_t)__shfl_up((int)s, off);
        const bool take = lane >= off && so == s;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float l2 = __shfl_up(lo[a], off), h2 = __shfl_up(hi[a], off);
            lo[a] = take ? fminf(lo[a], l2) : lo[a];
            hi[a] = take ? fmaxf(hi[a], h2) : hi[a];
        }
    }
    const uint32_t snext = (uint32_t)__shfl_down((int)s, 1);
    if (active && (lane == 63 || snext != s)) {
        uint32_t *b = segb + 6 * (size_t)s;
#pragma unroll
        for (int a = 0; a < 3; ++a) { atomicMin(b + a, core::float_ordered(lo[a])); atomicMax(b + 3 + a, core::float_ordered(hi[a])); }
    }
}
__global__ __launch_bounds__(BT) void k_seg_keys(size_t n, const uint32_t *__restrict__ order, const uint32_t *__restrict__ seg_of,
                                                 const uint32_t *__restrict__ segb, const float *__restrict__ cen, uint64_t *keys) {
    const size_t i = gid();
    if (i >= n) return;
    const uint32_t s = seg_of[i];
    float clo[3], chi[3];
    for (int a = 0; a < 3; ++a) { clo[a] = core::ordered_float(segb[6 * (size_t)s + a]); chi[a] = core::ordered_float(segb[6 * (size_t)s + 3 + a]); }
    const int ax = core::split_axis(clo, chi);
    keys[i] = ((uint64_t)s << 32) | core::float_ordered(cen[3 * (size_t)order[i] + ax]);
}

// boxes of the binary nodes of one level (deepest level first): a leaf from its faces, an internal node from its children
__global__ __launch_bounds__(BT) void k_node_boxes(uint32_t first_node, uint32_t n_nodes, const core::BinNode *__restrict__ bn,
                                                   const uint32_t *__restrict__ order, const float *__restrict__ fb, float *node_lo, float *node_hi) {
    const size_t t = gid();
    if (t >= n_nodes) return;
    const size_t k = first_node + t;
    const core::BinNode nd = bn[k];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    if (nd.left < 0) {
        for (uint32_t i = nd.first; i < nd.first + nd.count; ++i) {
            const float *b = fb + 6 * (size_t)order[i];
            for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], b[a]); hi[a] = fmaxf(hi[a], b[3 + a]); }
        }
    } else {
        for (int a = 0; a < 3; ++a) {
            lo[a] = fminf(node_lo[3 * (size_t)nd.left + a], node_lo[3 * (size_t)nd.right + a]);
            hi[a] = fmaxf(node_hi[3 * (size_t)nd.left + a], node_hi[3 * (size_t)nd.right + a]);
        }
    }
    for (int a = 0; a < 3; ++a) { node_lo[3 * k + a] = lo[a]; node_hi[3 * k + a] = hi[a]; }
}

// 64 threads = 64 / leaf_w leaves: SoA triangle blocks of leaf_w slots + face ids
__global__ __launch_bounds__(64) void k_leaf_soa(uint32_t n_leaves, uint32_t leaf_w, uint32_t leaf_shift, const uint32_t *__restrict__ leaf_nodes,
                                                 const core::BinNode *__restrict__ bn, const uint32_t *__restrict__ order,
                                                 const uint32_t *__restrict__ faces, const float *__restrict__ xyz, float *leaf_tri,
                                                 uint32_t *leaf_id) {
    const size_t l = (size_t)blockIdx.x * (64u >> leaf_shift) + (threadIdx.x >> leaf_shift);
    const uint32_t i = threadIdx.x & (leaf_w - 1);
    if (l >= n_leaves) return;
    const core::BinNode nd = bn[leaf_nodes[l]];
    uint32_t fid = TN_EMPTY;
    float v[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (i < nd.count) {
        fid = order[nd.first + i];
        const uint32_t *f = faces + 3 * (size_t)fid;
        for (int q = 0; q < 3; ++q)
            for (int k = 0; k < 3; ++k) v[q * 3 + k] = xyz[3 * (size_t)f[q] + k];
    }
    leaf_id[l * leaf_w + i] = fid;
    for (int q = 0; q < 9; ++q) leaf_tri[(l * 9 + q) * leaf_w + i] = v[q];
}

// one thread per wide node of level `lev` (node ids [snap[lev], snap[lev + 1])): greedy opening of its binary subtree
__global__ __launch_bounds__(BT) void k_collapse(uint32_t lev, const uint32_t *__restrict__ snap, uint32_t *wide_sub, uint32_t *counter,
                                                 uint32_t cap, const core::BinNode *__restrict__ bn, const float *__restrict__ node_lo,
                                                 const float *__restrict__ node_hi, float *boxes, uint32_t *child, uint32_t *flags) {
    const size_t w = (size_t)snap[lev] + gid();
    if (w >= snap[lev + 1]) return;
    const core::BinTreeView tree{bn, node_lo, node_hi};
    int kids[WIDE];
    const int nk = core::collapse_node((int)wide_sub[w], tree, kids);
    for (int i = 0; i < WIDE; ++i) {
        uint32_t ch = TN_EMPTY;
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        if (i < nk) {
            const int k = kids[i];
            for (int a = 0; a < 3; ++a) { lo[a] = node_lo[3 * (size_t)k + a]; hi[a] = node_hi[3 * (size_t)k + a]; }
            if (bn[k].left < 0) ch = 0x80000000u | (uint32_t)bn[k].leaf;
            else {
                const uint32_t cw = atomicAdd(counter, 1u);
                if (cw < cap) { wide_sub[cw] = (uint32_t)k; ch = cw; }
                else atomicOr(flags, core::FLAG_INTERNAL);
            }
        }
        for (int a = 0; a < 3; ++a) {
            boxes[(w * 6 + a) * WIDE + i] = lo[a];
            boxes[(w * 6 + 3 + a) * WIDE + i] = hi[a];
        }
        child[w * WIDE + i] = ch;
    }
}
__global__ void k_snap(uint32_t lev, uint32_t *snap, const uint32_t *counter) {
    if (gid() == 0) snap[lev + 2] = *counter;
}

struct Temp {   // rocPRIM scratch, grown on demand
    DevBuf<char> buf;
    void *need(size_t bytes) {
        if (bytes > buf.n) buf.alloc(bytes + bytes / 4 + 256);
        return buf.p;
    }
};

void sort_pairs(Temp &tmp, const uint64_t *kin, uint64_t *kout, const uint32_t *vin, uint32_t *vout, size_t n, unsigned end_bit,
                hipStream_t s) {
    size_t bytes = 0;
    TN_HIP(rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, n, 0u, end_bit, s));
    void *p = tmp.need(bytes);
    TN_HIP(rocprim::radix_sort_pairs(p, bytes, kin, kout, vin, vout, n, 0u, end_bit, s));
}
void exclusive_scan(Temp &tmp, const uint32_t *in, uint32_t *out, size_t n, hipStream_t s) {
    size_t bytes = 0;
    TN_HIP(rocprim::exclusive_scan(nullptr, bytes, in, out, 0u, n, rocprim::plus<uint32_t>(), s));
    void *p = tmp.need(bytes);
    TN_HIP(rocprim::exclusive_scan(p, bytes, in, out, 0u, n, rocprim::plus<uint32_t>(), s));
}
template <typename T>
T read_back(const T *dev, hipStream_t s) {
    T h{};
    TN_HIP(hipMemcpyAsync(&h, dev, sizeof(T), hipMemcpyDeviceToHost, s));
    TN_HIP(hipStreamSynchronize(s));
    return h;
}
unsigned bit_length(uint64_t v) { unsigned b = 0; while (v) { ++b; v >>= 1; } return b; }

}  // namespace

void device_build(size_t V, size_t T, const float *xyz, const uint32_t *cells, hipStream_t s, BuildTargets out, BuildInfo &info,
                  uint32_t leaf_w) {
    if (leaf_w != 16 && leaf_w != 32 && leaf_w != 64) throw Error("leaf width must be 16, 32 or 64");
    const uint32_t leaf_shift = leaf_w == 16 ? 4u : (leaf_w == 32 ? 5u : 6u);
    if (T == 0) throw Error("device_build needs at least one tetrahedron");
    const size_t n4 = 4 * T;
    Temp tmp;
    DevBuf<uint32_t> flags;   // [0] error flags, [1] max |coordinate| bits, [2..8) centroid bounds of the tets
    flags.alloc(8);
    TN_HIP(hipMemsetAsync(flags.p, 0, 8 * sizeof(uint32_t), s));

    // ------------------------------------------------------------ face table
    hipLaunchKernelGGL(k_cells_check_max, dim3(grid_for(n4)), dim3(BT), 0, s, n4, cells, (uint32_t)V, xyz, flags.p, flags.p + 1);
    if (read_back(flags.p, s) & core::FLAG_CELL_OOB) throw Error("cells contains a vertex index that is out of bounds");
    size_t cap = 16;
    while (cap < 8 * T + 16) cap <<= 1;
    DevBuf<uint32_t> slot, partner, first, fidx, tet_face;
    slot.alloc(cap); partner.alloc(n4); first.alloc(n4); fidx.alloc(n4); tet_face.alloc(n4);
    TN_HIP(hipMemsetAsync(slot.p, 0xFF, cap * sizeof(uint32_t), s));
    TN_HIP(hipMemsetAsync(partner.p, 0xFF, n4 * sizeof(uint32_t), s));
    hipLaunchKernelGGL(k_face_hash, dim3(grid_for(n4)), dim3(BT), 0, s, n4, cells, slot.p, (uint64_t)(cap - 1), partner.p, flags.p);
    hipLaunchKernelGGL(k_face_first, dim3(grid_for(n4)), dim3(BT), 0, s, n4, partner.p, first.p);
    exclusive_scan(tmp, first.p, fidx.p, n4, s);
    const uint32_t err = read_back(flags.p, s);
    if (err & core::FLAG_TRIPLE_FACE) throw Error("A triangle is shared by more than two tetrahedra!");
    const size_t F = (size_t)read_back(fidx.p + (n4 - 1), s) + read_back(first.p + (n4 - 1), s);
    slot.release();
    out.faces.alloc(3 * F);
    out.face_tets.alloc(2 * F);
    hipLaunchKernelGGL(k_face_emit, dim3(grid_for(n4)), dim3(BT), 0, s, n4, cells, partner.p, first.p, fidx.p, out.faces.p,
                       out.face_tets.p, tet_face.p);
    // hull faces in face-id order
    DevBuf<uint32_t> hflag, hidx;
    hflag.alloc(F); hidx.alloc(F);
    hipLaunchKernelGGL(k_hull_flag, dim3(grid_for(F)), dim3(BT), 0, s, F, out.face_tets.p, hflag.p);
    exclusive_scan(tmp, hflag.p, hidx.p, F, s);
    const size_t n_hull = (size_t)read_back(hidx.p + (F - 1), s) + read_back(hflag.p + (F - 1), s);

    // ------------------------------------------------------------ Morton order of the tets, walk records
    DevBuf<uint64_t> codes, codes2;
    DevBuf<uint32_t> idx, order, rec_of_tet;
    codes.alloc(T); codes2.alloc(T); idx.alloc(T); order.alloc(T); rec_of_tet.alloc(T);
    hipLaunchKernelGGL(k_init_bounds, dim3(1), dim3(BT), 0, s, flags.p + 2, (size_t)1);
    hipLaunchKernelGGL(k_tet_bounds, dim3(grid_for(T)), dim3(BT), 0, s, T, cells, xyz, flags.p + 2);
    hipLaunchKernelGGL(k_tet_codes, dim3(grid_for(T)), dim3(BT), 0, s, T, cells, xyz, flags.p + 2, codes.p, idx.p);
    sort_pairs(tmp, codes.p, codes2.p, idx.p, order.p, T, 63u, s);
    hipLaunchKernelGGL(k_inverse, dim3(grid_for(T)), dim3(BT), 0, s, T, order.p, rec_of_tet.p);
    out.vars.alloc(n4);
    hipLaunchKernelGGL(k_walk_vars, dim3(grid_for(n4)), dim3(BT), 0, s, n4, order.p, rec_of_tet.p, cells, xyz, tet_face.p,
                       out.faces.p, out.face_tets.p, out.vars.p, flags.p);
    {
        DevBuf<uint32_t> vmin;
        vmin.alloc(V);
        hipLaunchKernelGGL(k_fill_u32, dim3(grid_for(V)), dim3(BT), 0, s, V, vmin.p, 0x7F800000u);   // +inf
        hipLaunchKernelGGL(k_tet_thin, dim3(grid_for(T)), dim3(BT), 0, s, T, cells, xyz, vmin.p);
        hipLaunchKernelGGL(k_thin_patch, dim3(grid_for(T)), dim3(BT), 0, s, T, cells, vmin.p, rec_of_tet.p, out.vars.p);
        TN_HIP(hipStreamSynchronize(s));   // vmin is freed at the end of this scope
    }

    // ------------------------------------------------------------ hull tree (host threading of the downloaded faces)
    std::vector<float> hinfo(n_hull * 12);
    if (n_hull) {
        DevBuf<uint32_t> dinfo;
        dinfo.alloc(n_hull * 12);
        hipLaunchKernelGGL(k_hull_info, dim3(grid_for(F)), dim3(BT), 0, s, F, hflag.p, hidx.p, out.faces.p, out.face_tets.p, tet_face.p,
                           rec_of_tet.p, xyz, dinfo.p, flags.p);
        TN_HIP(hipMemcpyAsync(hinfo.data(), dinfo.p, hinfo.size() * sizeof(float), hipMemcpyDeviceToHost, s));
        TN_HIP(hipStreamSynchronize(s));
    }
    codes.release(); codes2.release(); idx.release();

    // ------------------------------------------------------------ face BVH
    // host: shape of the tree (while the device works on the records)
    std::vector<core::BinNode> bn;
    std::vector<std::vector<uint32_t>> frontier;
    std::vector<uint32_t> level_start, leaf_nodes;
    build_bin_topology(F, bn, frontier, level_start, leaf_nodes, leaf_w);
    const size_t nn = bn.size(), n_leaves = leaf_nodes.size();
    HostHullBvh hth;
    build_hull_from_info(hinfo, hth);
    out.hull_nodes.upload(hth.nodes_and_flat());
    out.hull_tris.upload(hth.tris);

    DevBuf<core::BinNode> dbn;
    dbn.upload(bn);
    DevBuf<uint32_t> dleaf_nodes;
    dleaf_nodes.upload(leaf_nodes);
    DevBuf<float> fb, cen, node_lo, node_hi;
    fb.alloc(6 * F); cen.alloc(3 * F); node_lo.alloc(3 * nn); node_hi.alloc(3 * nn);
    hipLaunchKernelGGL(k_face_boxes, dim3(grid_for(F)), dim3(BT), 0, s, F, out.faces.p, xyz, fb.p, cen.p);
    DevBuf<uint32_t> ord_a, ord_b, seg_of, segb;
    DevBuf<uint64_t> keys_a, keys_b;
    ord_a.alloc(F); ord_b.alloc(F); seg_of.alloc(F); keys_a.alloc(F); keys_b.alloc(F);
    hipLaunchKernelGGL(k_iota, dim3(grid_for(F)), dim3(BT), 0, s, F, ord_a.p);
    const size_t split_rounds = frontier.size() - 1;
    {
        // per split round l >= 1: for every segment of round l - 1 its first successor and the position of the second
        std::vector<uint32_t> child_first, split_pos;
        std::vector<size_t> offs(split_rounds, 0);
        size_t max_seg = 1;
        for (size_t l = 0; l < split_rounds; ++l) {
            max_seg = std::max(max_seg, frontier[l].size());
            if (l == 0) continue;
            offs[l] = child_first.size();
            uint32_t running = 0;
            for (uint32_t k : frontier[l - 1]) {
                child_first.push_back(running);
                const bool split = bn[k].left >= 0 && bn[k].level + 1 == (uint32_t)l;   // it split in round l - 1 -> l
                split_pos.push_back(split ? bn[bn[k].right].first : 0xFFFFFFFFu);
                running += split ? 2u : 1u;
            }
        }
        DevBuf<uint32_t> d_child_first, d_split_pos;
        if (!child_first.empty()) { d_child_first.upload(child_first); d_split_pos.upload(split_pos); }
        segb.alloc(6 * max_seg);
        for (size_t l = 0; l < split_rounds; ++l) {
            const uint32_t nseg = (uint32_t)frontier[l].size();
            hipLaunchKernelGGL(k_init_bounds, dim3(grid_for((size_t)nseg * 6)), dim3(BT), 0, s, segb.p, (size_t)nseg);
            hipLaunchKernelGGL(k_seg_bounds, dim3(grid_for(F)), dim3(BT), 0, s, F, ord_a.p, l ? d_child_first.p + offs[l] : nullptr,
                               l ? d_split_pos.p + offs[l] : nullptr, cen.p, seg_of.p, segb.p);
            hipLaunchKernelGGL(k_seg_keys, dim3(grid_for(F)), dim3(BT), 0, s, F, ord_a.p, seg_of.p, segb.p, cen.p, keys_a.p);
            sort_pairs(tmp, keys_a.p, keys_b.p, ord_a.p, ord_b.p, F, 32u + bit_length(nseg - 1), s);
            ord_a.swap(ord_b);
        }
        TN_HIP(hipStreamSynchronize(s));   // d_child_first / d_split_pos go out of scope
    }
    for (size_t l = level_start.size() - 1; l-- > 0;) {
        const uint32_t first_node = level_start[l], cnt = level_start[l + 1] - level_start[l];
        hipLaunchKernelGGL(k_node_boxes, dim3(grid_for(cnt)), dim3(BT), 0, s, first_node, cnt, dbn.p, ord_a.p, fb.p, node_lo.p, node_hi.p);
    }
    out.bvh.leaf_tri.alloc(n_leaves * 9 * leaf_w);
    out.bvh.leaf_id.alloc(n_leaves * leaf_w);
    {
        const unsigned per_block = 64u >> leaf_shift;
        hipLaunchKernelGGL(k_leaf_soa, dim3((unsigned)((n_leaves + per_block - 1) / per_block)), dim3(64), 0, s, (uint32_t)n_leaves, leaf_w,
                           leaf_shift, dleaf_nodes.p, dbn.p, ord_a.p, out.faces.p, xyz, out.bvh.leaf_tri.p, out.bvh.leaf_id.p);
    }
    // collapse, level by level
    const uint32_t wcap = (uint32_t)nn;
    DevBuf<float> boxes_tmp;
    DevBuf<uint32_t> child_tmp, wide_sub, ctl;   // ctl: [0] counter, [1 ..] snap
    boxes_tmp.alloc((size_t)wcap * 6 * WIDE); child_tmp.alloc((size_t)wcap * WIDE); wide_sub.alloc(wcap);
    ctl.alloc(MAX_WIDE_LEVELS + 4);
    {
        std::vector<uint32_t> h(MAX_WIDE_LEVELS + 4, 0u);
        h[0] = 1u; h[1] = 0u; h[2] = 1u;   // one node (the root), level 0 = [0, 1)
        TN_HIP(hipMemcpyAsync(ctl.p, h.data(), h.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        TN_HIP(hipMemsetAsync(wide_sub.p, 0, sizeof(uint32_t), s));   // root of the binary tree
        TN_HIP(hipStreamSynchronize(s));                              // h goes out of scope
    }
    for (uint32_t lev = 0; lev < (uint32_t)MAX_WIDE_LEVELS; ++lev) {
        hipLaunchKernelGGL(k_collapse, dim3(grid_for(wcap)), dim3(BT), 0, s, lev, ctl.p + 1, wide_sub.p, ctl.p, wcap, dbn.p, node_lo.p,
                           node_hi.p, boxes_tmp.p, child_tmp.p, flags.p);
        hipLaunchKernelGGL(k_snap, dim3(1), dim3(64), 0, s, lev, ctl.p + 1, ctl.p);
    }
    std::vector<uint32_t> hctl(MAX_WIDE_LEVELS + 4);
    TN_HIP(hipMemcpyAsync(hctl.data(), ctl.p, hctl.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    TN_HIP(hipStreamSynchronize(s));
    const size_t n_wide = hctl[0];
    if (hctl[1 + MAX_WIDE_LEVELS] != n_wide) throw Error("face BVH too deep (more than 16 levels of 64-wide nodes)");
    if (read_back(flags.p, s) & core::FLAG_INTERNAL) throw Error("internal: inconsistent adjacency in the device build");
    std::vector<uint32_t> hchild(n_wide * WIDE);
    TN_HIP(hipMemcpyAsync(hchild.data(), child_tmp.p, hchild.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    out.bvh.boxes.alloc(n_wide * 6 * WIDE);
    out.bvh.child.alloc(n_wide * WIDE);
    TN_HIP(hipMemcpyAsync(out.bvh.boxes.p, boxes_tmp.p, n_wide * 6 * WIDE * sizeof(float), hipMemcpyDeviceToDevice, s));
    TN_HIP(hipMemcpyAsync(out.bvh.child.p, child_tmp.p, n_wide * WIDE * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
    TN_HIP(hipStreamSynchronize(s));

    info.F = (uint32_t)F;
    info.n_hull = (uint32_t)n_hull;
    info.n_hull_nodes = (uint32_t)(hth.nodes.size() / 8);
    info.max_stack = wide_bvh_max_stack(hchild.data(), n_wide);
    {
        const uint32_t bits = read_back(flags.p + 1, s);
        std::memcpy(&info.scene_max, &bits, 4);
    }
    out.bvh.set_view(n_wide, info.scene_max, leaf_w);
}

}  // namespace tn
