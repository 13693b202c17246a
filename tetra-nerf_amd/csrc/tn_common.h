// tn_common.h -- shared declarations of libtetranerf_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#define TN_EMPTY 0xFFFFFFFFu
#define TN_EPS 1e-6f  // tie window of the pairing stage (reference: optix_trace_rays.cu:8)

namespace tn {

void set_error(const std::string &msg);

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

#define TN_HIP(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            throw tn::Error(std::string(#expr) + " failed: " + hipGetErrorString(e_) + " (" + \
                            __FILE__ + ":" + std::to_string(__LINE__) + ")");                \
    } while (0)

// Kernels that ask for more than 64 KB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize, and the
// attribute is per DEVICE: remembered per (call site, device) so that a process driving several GPUs sets it on each.
struct PerDeviceOnce {
    unsigned long long done = 0;  // bit d = set on device d (racing threads at worst set it twice)
    template <typename Fn>
    void run(Fn &&fn) {
        int dev = 0;
        TN_HIP(hipGetDevice(&dev));
        const unsigned long long bit = 1ull << (dev & 63);
        if (__atomic_load_n(&done, __ATOMIC_ACQUIRE) & bit) return;
        fn();
        __atomic_fetch_or(&done, bit, __ATOMIC_RELEASE);
    }
};
inline void allow_dynamic_lds(const void *kernel, size_t bytes) {
    TN_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
}

constexpr int WIDE = 64;            // BVH branching factor = wavefront width
constexpr int STACK_CAP = 64 * 6;   // traversal stack entries per wave

// Wide (64-ary) BVH over the unique faces.  Built by recursive median splits of the face centroids
// (compact leaves of 33..64 triangles) and collapsed to 64-wide internal nodes.  Leaves store the
// triangle vertices inline, SoA, so a wave reads a leaf with 10 coalesced 256-B loads; an internal node
// is 6 SoA box rows + 1 row of child references (bit 31 = leaf index, TN_EMPTY = no child).
struct WideBvh {
    const float *leaf_tri;      // [n_leaves][9][leaf_w]  v0.xyz v1.xyz v2.xyz
    const uint32_t *leaf_id;    // [n_leaves][leaf_w]     face id or TN_EMPTY
    uint32_t leaf_w;            // faces per leaf block: 16, 32 or 64 (64 / leaf_w leaves are tested per wave instruction)
    uint32_t leaf_shift;        // log2(leaf_w)
    const float *boxes;         // [n_nodes][6][64]   lo.xyz hi.xyz of the children
    const uint32_t *child;      // [n_nodes][64]      child reference
    uint32_t n_nodes;           // node 0 is the root
    float scene_max;            // max |coordinate| over the mesh vertices
};

// Per-tet record of the adjacency walk: one 128-byte line.
//   vert[k]  vertex ids (cell order);  pos[k] their positions
//   nbr[k]   tet behind the face opposite local vertex k (TN_EMPTY on the hull)
//   face[k]  id of that face in the face table
//   perm     per face k, 3x2 bits: local vertex index (0..3) of the face's 1st/2nd/3rd
//            STORED vertex (first-seen triple), bits [6k, 6k+6)
//   back     per face k, 2 bits: the local index of that face in the neighbour tet
// Records are stored in Morton order of the tet centroids (consecutive steps of a walk and neighbouring
// rays touch neighbouring lines); nbr[] are record indices, `orig` is the caller's tet id.
struct alignas(128) TetRec {
    uint32_t vert[4];
    uint32_t nbr[4];
    uint32_t face[4];
    float pos[4][3];
    uint32_t perm;
    uint32_t back;
    uint32_t orig;     // index of this tet in the caller's `cells` (records are stored in Morton order)
    uint32_t euv[2];   // per face k, 12 bits at [12*(k&1)] of euv[k>>1]: three 4-bit codes (U,V,W of the
                       // face in stored order): bits 0-2 = edge pair index (01,02,03,12,13,23), bit 3 = negate
    uint32_t cmb[3];   // per ordered face pair (entry e, exit x), 6 bits at 6*(3e + x - (x>e)): for each entry-face
                       // slot j the position (0..2) of that vertex in the exit face's stored order, 3 = absent
};
static_assert(sizeof(TetRec) == 128, "TetRec must be one 128-B line");

// Entry-face-specialised walk record: one per (tet record r, entry local face e), index 4r + e, 64 B.
// Local numbering {0: n, 1: a, 2: b, 3: c}: n = the vertex opposite the entry face, (a,b,c) = the entry face's
// STORED triple -- exactly the order the previous step left the face's sheared vertices and edge functions in,
// so nothing of the entry face is permuted or recomputed.  Exit x in {0,1,2} = the face opposite a / b / c.
struct alignas(64) WalkVar {
    // quad 0 + 1 + 2: what the walk reads (three 16-byte loads)
    float pn[3];       // position of n
    uint32_t fid0;     // face id of exit 0
    uint32_t nb[3];    // variant entered through exit x (TN_EMPTY: hull)
    uint32_t fid1;     // face id of exit 1
    // quad 2 + 3: what the segment writer reads (two 16-byte loads)
    uint32_t orig;     // the caller's tet id
    uint32_t code_lo;  // per exit x, 12 bits at 12x: p0 p1 p2 (2 bits each: the exit face's stored order in local
                       // numbers), then c0 c1 c2 (2 bits each: position of a / b / c in that order, 3 = absent)
    uint32_t code_hi;  // bits 32..35 of the code word
    uint32_t fid2;     // face id of exit 2
    uint32_t vid[4];   // vertex ids n, a, b, c (= vertex_indices of a segment entered through this face)
#if defined(__HIPCC__)
    __host__ __device__
#endif
    uint32_t fid(uint32_t x) const { return x == 0 ? fid0 : (x == 1 ? fid1 : fid2); }   // dword 3 + 4x of the record
#if defined(__HIPCC__)
    __host__ __device__
#endif
    void set_fid(uint32_t x, uint32_t f) { if (x == 0) fid0 = f; else if (x == 1) fid1 = f; else fid2 = f; }
};
static_assert(sizeof(WalkVar) == 64, "WalkVar must be 64 bytes");

// What the kernels read: the 64-byte record (the build's product; kept as the unit of the host / device / emulation
// equality checks) is de-interleaved at the end of load_tetrahedra into three tables by consumer, so that each kernel's
// random accesses fetch only bytes it uses (round 3: the walk read 48 and the segment writer 32 of every 64-byte record,
// i.e. 37 % / 25 % of each 128-byte line; the tables are 2x / 2x / 4x smaller than the combined one for the XCD's 4 MiB L2):
//   WalkHot  [4T] 32 B  walk:            position of n, the three neighbour variants, the 36-bit code + thin exponent
//   WalkCold [4T] 32 B  segment writer, meshes whose table fits the L2s (up to WALK_TET_MIN_TETS tets): vertex ids
//                       (n, a, b, c), the caller's tet id, the combine code of the 3 exits -- nothing to derive per segment
//   WalkTet  [T]  32 B  segment writer, larger meshes (round 4): ONE record per tet -- a quarter of the table, four
//                       Morton-neighbouring tets per 128-byte line instead of one.  The tet's vertex ids in tet-local
//                       order, the caller's tet id, and per entry face e what WalkCold bakes into the record of (tet, e):
//                       the local indices of its a / b / c (6 bits at 6e of `perm`; n = local vertex e) and the combine code
//                       of its three exits (18 bits at 18e of the 72-bit field cmb_lo | cmb_hi << 32 | perm[31:24] << 64);
//                       the writer derives (n, a, b, c) and the code per segment (~40 VALU instructions).  Measured on one
//                       box, interleaved (profiles/r04d_lib_ab.txt): 2^20 incoherent rays on 1M tets 18.96 -> 16.94 ms
//                       (-10.7 %: the per-entry table is 129 MB there and every hit fetched a 64-byte sector beyond the L2
//                       for 32 bytes), but +3 % on the 100k / 300k frames, whose tables the L2s hold -- hence both.
//                       The log entry names (tet, e, exit) either way.
//   WalkFid  [4T] 16 B  literal pairing: face id of exit 0 / 1 / 2 (total order (t, face id) of the sort)
struct alignas(32) WalkHot { float pn[3]; uint32_t nb0; uint32_t nb1, nb2, code_lo, code_hi; };
struct alignas(32) WalkCold { uint32_t vid[4]; uint32_t orig, cmb, pad0, pad1; };   // cmb: per exit x 6 bits at 6x (c0 c1 c2)
constexpr uint32_t WALK_TET_MIN_TETS = 500000;   // from here on the writer reads the per-tet table (see above)
struct alignas(32) WalkTet {
    uint32_t vert[4]; uint32_t orig, perm, cmb_lo, cmb_hi;
#if defined(__HIPCC__)
    __host__ __device__
#endif
    uint32_t cmb(uint32_t e) const {   // per exit x of entry face e: 6 bits at 6x (c0 c1 c2: position of a / b / c in the exit face's stored order)
        const unsigned long long lo64 = (unsigned long long)cmb_lo | ((unsigned long long)cmb_hi << 32);
        const uint32_t v = e == 3u ? ((uint32_t)(lo64 >> 54) | ((perm >> 24) << 10)) : (uint32_t)(lo64 >> (18u * e));
        return v & 0x3FFFFu;
    }
#if defined(__HIPCC__)
    __host__ __device__
#endif
    uint32_t vid(uint32_t e, uint32_t m) const {   // vertex ids (n, a, b, c) of a segment entered through face e, m = 0..3
        return m == 0 ? vert[e & 3u] : vert[(perm >> (6u * e + 2u * (m - 1u))) & 3u];
    }
};
struct alignas(16) WalkFid { uint32_t fid[3]; uint32_t pad; };
static_assert(sizeof(WalkHot) == 32 && sizeof(WalkCold) == 32 && sizeof(WalkTet) == 32 && sizeof(WalkFid) == 16, "split walk records");

struct DeviceMesh {
    const float *xyz = nullptr;       // borrowed [V,3]
    const uint32_t *cells = nullptr;  // borrowed [T,4]
    uint32_t V = 0, T = 0, F = 0;
    uint32_t *faces = nullptr;      // [F,3] first-seen unsorted triple
    uint32_t *face_tets = nullptr;  // [F,2]
    WideBvh bvh{};                  // over all faces
    // adjacency walk
    const WalkHot *hot = nullptr;   // [4T] entry-face-specialised records, split by consumer (see WalkHot)
    const WalkCold *cold = nullptr; // [4T] the writer's table of meshes below WALK_TET_MIN_TETS tets, else null
    const WalkTet *tets = nullptr;  // [T] one record per tet (see WalkTet), else null
    const WalkFid *fidt = nullptr;
    const float4 *hull_nodes = nullptr;  // threaded binary BVH over the hull faces (2 float4 per node)
    const float4 *hull_tris = nullptr;   // 3 float4 per hull face
    uint32_t n_hull_nodes = 0;
    uint32_t n_hull = 0;
};

// host-side build products (tn_mesh.cpp)
struct HostMesh {
    std::vector<uint32_t> faces;      // 3F
    std::vector<uint32_t> face_tets;  // 2F
    float scene_max = 0.f;
};

struct HostWideBvh {
    std::vector<float> leaf_tri;
    std::vector<uint32_t> leaf_id;
    std::vector<float> boxes;
    std::vector<uint32_t> child;
    uint32_t max_stack = 1;  // worst-case entries of the traversal stack (must stay <= STACK_CAP)
    uint32_t leaf_w = WIDE;  // faces per leaf block
};

struct HostHullBvh {
    std::vector<float> nodes;  // [n_nodes][8]: lo.xyz, skip | hi.xyz, leaf (first<<3|count, or ~0)
    std::vector<float> tris;   // [n_hull][12]: v0.xyz, face id | v1.xyz, tet record | v2.xyz, local face  (Morton order)
    // Flat two-level box table over the same Morton-ordered faces (round 6, the walk's LDS hull search; only for hulls of at
    // most HULL_FLAT_MAX faces, else empty): [G groups][8] then [L leaves][8], each lo.xyz, first | hi.xyz, count -- a leaf
    // = 2 consecutive faces (first = face slot), a group = 8 consecutive leaves (first = leaf index)
    std::vector<float> flat;
    // what is uploaded as the device's hull_nodes array: the threaded tree, then the flat table
    std::vector<float> nodes_and_flat() const { std::vector<float> v(nodes); v.insert(v.end(), flat.begin(), flat.end()); return v; }
};
constexpr uint32_t HULL_FLAT_MAX = 1024;   // faces: 512 leaves + 64 groups = 18 KB of LDS per walk block
inline uint32_t hull_flat_leaves(uint32_t n_hull) { return n_hull && n_hull <= HULL_FLAT_MAX ? (n_hull + 1u) / 2u : 0u; }
inline uint32_t hull_flat_groups(uint32_t n_hull) { return (hull_flat_leaves(n_hull) + 7u) / 8u; }
// `recs` / `rec_of_tet` (record index of each original tet) from build_tet_records: every hull face carries
// the record index of its tet and its local face index, so the walk starts without further lookups
void build_hull_threaded(const float *xyz, const uint32_t *faces, const uint32_t *face_tets,
                         const std::vector<uint32_t> &ids, const std::vector<TetRec> &recs,
                         const std::vector<uint32_t> &rec_of_tet, HostHullBvh &out);

// the same from 12 floats per hull face in ascending face-id order (v0.xyz, face id | v1.xyz, tet record | v2.xyz, local face)
void build_hull_from_info(const std::vector<float> &info, HostHullBvh &out);
uint32_t wide_bvh_max_stack(const uint32_t *child, size_t n_nodes);

// first-seen face table; throws tn::Error("A triangle is shared by more than two tetrahedra!")
void build_face_table(size_t T, const uint32_t *cells, HostMesh &out);
// wide BVH over the faces listed in `ids` (global face ids)
void build_wide_bvh(const float *xyz, const uint32_t *faces, const std::vector<uint32_t> &ids,
                    HostWideBvh &out, uint32_t leaf_w = WIDE);
// adjacency records
void build_tet_records(size_t T, const uint32_t *cells, const float *xyz, const HostMesh &hm,
                       std::vector<TetRec> &out, std::vector<uint32_t> &rec_of_tet);

// entry-face-specialised walk records from the (Morton-ordered) tet records
void build_walk_variants(const std::vector<TetRec> &recs, std::vector<WalkVar> &out);

}  // namespace tn
