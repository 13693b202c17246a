// tn_kernels.h -- host-callable launchers of the HIP kernels.
#pragma once
#include "tn_common.h"

namespace tn {

struct TraceParams {
    const float *origins;      // [R,3]
    const float *dirs;         // [R,3]
    const uint32_t *faces;     // [F,3]
    const uint32_t *face_tets; // [F,2]
    WideBvh bvh;
    uint32_t *out_num;         // [R]
    uint32_t *out_cells;       // [R,M]
    float *out_bary;           // [R,M,2,3]
    float *out_dist;           // [R,M,2]
    uint32_t *out_verts;       // [R,M,4] or null
    uint32_t M;
    size_t num_items;          // rays (or entries of ray_list) to process
    const uint32_t *ray_list;  // optional indirection: item -> ray index
    const uint32_t *item_count; // optional device-side item count (overrides num_items in-kernel)
    uint32_t lds_cap;          // k_trace_general: entries of the LDS hit arrays (0 = M); rays with more hits go to overflow_list
    uint32_t *overflow_list;   // [num_items] / [1]
    uint32_t *overflow_count;
    unsigned long long *stats; // [4] device counters or null
    uint32_t compact_rows;     // 1: slots >= num_visited are left unwritten (TN_TRACE_COMPACT_ROWS); 0: every slot of a row
    uint32_t sort_passes;      // k_postprocess_log: odd-even passes over the logged hits before the bitonic network takes over
};

// general all-hits path, one wavefront per ray (tn_trace_general.hip)
void launch_trace_general(const TraceParams &p, hipStream_t stream);
void launch_postprocess_hits(const TraceParams &p, const uint32_t *hit_count, const uint32_t *hit_ids,
                             const float *hit_t, const float *hit_uv, hipStream_t stream);
size_t trace_general_smem_bytes(uint32_t M);
void launch_trace_triangles(const TraceParams &p, uint32_t *out_ids, float *out_t, float *out_uv, uint32_t *out_v3,
                            hipStream_t stream);
void launch_find_tetrahedra(const TraceParams &p, const float *points, uint32_t *out_tet, float *out_bary,
                            uint32_t *out_verts, hipStream_t stream);

// de-interleaves the 64-byte build records into the three consumer tables (tn_trace_walk.hip)
// exactly one of cold / tets is non-null (tn_common.h: WALK_TET_MIN_TETS)
void launch_split_walk_records(size_t n4, const WalkVar *vars, WalkHot *hot, WalkCold *cold, WalkTet *tets, WalkFid *fidt, hipStream_t stream);

// adjacency walk, one lane per ray (tn_trace_walk.hip)
struct WalkParams {
    TraceParams t;
    const WalkHot *vars;       // entry-face-specialised records of the walk: the walk's 32 bytes (k_trace_walk)
    float scene_max;           // max |coordinate| of the mesh (box padding)
    const float4 *hull_nodes;  // threaded per-lane hull tree
    const float4 *hull_tris;
    uint32_t n_hull_nodes;
    const float4 *hull_flat;   // flat two-level box table of the hull (HostHullBvh::flat): groups, then leaves; staged in LDS
    uint32_t n_hull_groups, n_hull_leaves;   // 0: use the threaded tree
    uint32_t n_hull;           // hull faces
    uint4 *hull_entry;         // [num_items] k_hull_entry -> k_trace_walk (layout: tn_trace_walk.hip)
    uint32_t *fallback_list;   // [R] rays for the BVH all-hits kernel (global ray ids)
    uint32_t *fallback_count;  // [1]
    uint2 *literal_list;       // [num_items] {ray index within this launch, hits in the log}: sound chains whose ORDER is
    uint32_t *literal_count;   // [1]            not certified -> literal sort + pairing of the logged hits
    uint32_t *walk_n;          // [num_items] hits logged per certified ray (0 = miss), TN_EMPTY = literal / fallback
    uint4 *hit_log;            // [ceil(num_items / 64)][M][64] x {t, u, v, variant | exit << 30}: hit k of ray r at
                               // ((r / 64) * M + k) * 64 + r % 64 (a wave's 64 lanes store 1 KB of consecutive bytes per
                               // step); exit code 3 = entry hull face, its face id in the low 30 bits
    size_t ray_base;           // global index of item 0 (rays are traced in chunks when the log would be too large)
    uint32_t *risk_list;       // [num_items] certified rays inside the WIDE band (risk_band x 8 delta: 16 delta by default) of a certification guard: every one of
    uint32_t *risk_count;      // [1]          them is cross-checked (k_verify_counts); null: not collected
    float risk_band;           // width of that band in units of the guards' own 8 delta (tn option "risk_band")
    uint32_t cert_ends;        // the order test: 0 round 5's pairwise test, 3 the same + the end-of-chain rules A-C, 1 round 6's cluster test (A-D)
};
// lds_reserve: bytes of (unused) dynamic LDS per block = an occupancy limit (160 KB / lds_reserve blocks per CU), 0 = none
void launch_trace_walk(const WalkParams &p, hipStream_t stream, size_t lds_reserve = 0);
// literal sort + pairing of the logged hits of the rays in literal_list (tn_trace_general.hip); rows of launch item i
// are p.out_*[i] (the TraceParams of the same walk launch)
void launch_postprocess_log(const TraceParams &p, const WalkFid *fidt, const uint4 *hit_log, const uint2 *literal_list,
                            const uint32_t *literal_count, size_t max_items, hipStream_t stream);

// count-only BVH cross-check of every stride-th certified ray (tn_trace_general.hip: k_verify_counts); p = the
// TraceParams of the walk launch; mismatching rays are appended to the fallback list (global ids: ray_base + index)
// late: mismatching rays only go to the list (their rows are re-traced at the end of the call), walk_n stays
// ray_list / list_count (nullable): check exactly those rays (indices within this launch; the count lives on the device; the grid is
// sized for max_list) instead of every stride-th one -- the walk's risk list
void launch_verify_counts(const TraceParams &p, uint32_t stride, uint32_t *walk_n, uint32_t *fallback_list, uint32_t *fallback_count,
                          size_t ray_base, hipStream_t stream, bool late = false, bool inject = false, const uint32_t *ray_list = nullptr,
                          const uint32_t *list_count = nullptr, size_t max_list = 0);

// hit log -> rows of the rays the walk certified (walk_n[ray] != TN_EMPTY): k_write_segments writes the segment records
// + the tail constants up to the next multiple of 32 slots (a 128-byte line boundary in all four row arrays);
// k_fill_range streams the rest of the constant tails.  Every byte is written once.
struct WriteParams {
    size_t num_rays;
    uint32_t M;
    uint32_t dense_tails;      // 0: slots >= num_visited are left unwritten (non-reference option)
    const uint32_t *walk_n;    // hits in the log; TN_EMPTY: the row belongs to the literal / BVH kernels
    const uint4 *hit_log;
    const WalkCold *cold;      // the segment writer's table: one 32-byte record per (tet, entry face) ...
    const WalkTet *tets;       // ... or per tet (exactly one of the two is non-null)
    uint32_t *out_cells;
    float *out_bary;
    float *out_dist;
    uint32_t *out_verts;       // nullable
};
void launch_write_segments(const WriteParams &q, hipStream_t stream, unsigned max_blocks = 0);
// all_rows: slots [k_split, M) of every row; otherwise slots [ceil32(out_num[r]), k_split) of the certified rows
constexpr unsigned FILL_FINE = 0xFFFFFFFFu;   // max_blocks value: one block per row (k_fill_rows_fine)
constexpr unsigned FILL_LINEAR = 0xFFFFFFFEu; // max_blocks value: one linear stream per array, one store per thread (k_fill_linear)
void launch_fill_range(size_t num_rays, uint32_t M, bool all_rows, const uint32_t *walk_n, const uint32_t *out_num,
                       uint32_t *out_cells, float *out_bary, float *out_dist, uint32_t *out_verts, hipStream_t stream,
                       uint32_t k_split, bool nontemporal, unsigned max_blocks = 0);

// sample -> segment matching (tn_match.hip)
void launch_find_matched_cells(size_t R, size_t S, size_t M, const uint32_t *num_visited,
                               const uint32_t *visited, const float *dist, const float *bary,
                               const float *distances, const uint32_t *verts, uint32_t *cells_out,
                               uint32_t *verts_out, uint8_t *mask_out, float *bary_out, hipStream_t stream,
                               const uint32_t *ray_index = nullptr, const uint32_t *count = nullptr);

// barycentric gather and its adjoint (tn_interp.hip); throws on unsupported D
void launch_interpolate_values(uint32_t D, uint32_t V, uint32_t n, uint32_t Fd, const uint32_t *vi,
                               const float *bc, const float *field, float *result, hipStream_t stream);
void launch_interpolate_values_backward(uint32_t D, uint32_t V, uint32_t n, uint32_t Fd, const uint32_t *vi,
                                        const float *bc, const float *grad_in, bool rows_major, float *field_grad,
                                        hipStream_t stream);

// the same on a VERTEX-MAJOR field [V, Fd] / into a vertex-major gradient (accumulated; the caller zeroes it): no
// per-call transposition and no temporaries -- what a caller that keeps a vertex-major shadow of the field uses
void launch_interpolate_values_vm(uint32_t D, uint32_t n, uint32_t Fd, const uint32_t *vi, const float *bc,
                                  const float *fieldT, float *result, hipStream_t stream);
void launch_interpolate_values_backward_vm_det(uint32_t D, uint32_t V, uint32_t n, uint32_t Fd, const uint32_t *vi, const float *bc,
                                               const float *grad_rows, float *gradT, hipStream_t stream);
void launch_interpolate_values_backward_vm(uint32_t D, uint32_t n, uint32_t Fd, const uint32_t *vi, const float *bc,
                                           const float *grad_rows, float *gradT, hipStream_t stream);

// shallow MLP + heads (tn_mlp.hip); all weights in nn.Linear layout [out, in] row-major, fp32
struct MlpWeights {
    const float *w1, *b1;  // [128,64],  [128]   mlp_base layer 0
    const float *w2, *b2;  // [128,128], [128]   mlp_base layer 1
    const float *w3, *b3;  // [128,128], [128]   mlp_base layer 2 (+ ReLU out)
    const float *wd, *bd;  // [1,128],   [1]     density head (+ softplus)
    const float *wh, *bh;  // [128,155], [128]   mlp_head: columns = [dir encoding 27 | base 128]
    const float *wr, *br;  // [3,128],   [3]     rgb head (+ sigmoid)
};
// The packed forms of one set of weights (tn_mlp_set_weights packs them once per parameter version) + the scratch the
// kernels need beside them.  Everything is device memory owned by the tn_mlp handle (tn_api.hip).
struct MlpPacks {
    const float *pk_plain;     // fp32 MFMA forward, layer-1 K order of a [64, n] feature-major input
    const float *pk_gather;    // the same with layer-1 K order of the fused gather (forward, render pass, backward)
    const float *pt;           // transposed weights in accumulator order (backward)
    const uint4 *blob;         // bf16x3 pieces (forward, mode 1)
    const float *wenc;         // [128][28] the direction encoding's columns of mlp_head (head_ray_term)
    float *hterm;              // [rays][128] the head layer's per-ray term of the current call: Wh[:, :27] enc(dir) + ray_bias
    float *enc;                // [rays][mlp_enc_floats_per_ray()] direction encodings of the current call
    const float *ray_bias;     // per CALL (set by the entry point, never stored): [rays][128] added to the head layer's
                               // pre-activation (appearance embedding, tn_mlp_common.h: add_ray_bias), or null
    float *grad_scratch;       // [mlp_param_grad_scratch_floats()] per-block partial sums of the parameter gradients
};
size_t mlp_pack_floats();              // tn_mlp.hip
size_t mlp_backward_pack_floats();     // tn_mlp_bwd.hip
size_t mlp_x3_blob_u4();               // tn_mlp_x3.hip
size_t mlp_enc_floats_per_ray();       // 32: covers the fp32 kernels' 28 and the bf16x3 kernel's 32
void launch_mlp_pack(const MlpWeights &w, float *pk, bool gather_l1, hipStream_t stream);
void launch_pack_wenc(const MlpWeights &w, float *wenc, hipStream_t stream);   // [128 * 28] floats
void launch_mlp_pack_t(const MlpWeights &w, float *pt, hipStream_t stream);
void launch_mlp_pack_x3(const MlpWeights &w, uint4 *blob, hipStream_t stream);
// feats != null: input is the [64, n] feature buffer; feats == null: the kernel gathers the features itself from
// (vi [n,4], bc [n,3], fieldT [V, 64] VERTEX-major)
// count (nullable): device-side number of RAYS (n, num_rays are then upper bounds the grid is sized for)
void launch_mlp_forward(size_t n, uint32_t samples_per_ray, size_t num_rays, const float *feats, const uint32_t *vi,
                        const float *bc, const float *fieldT, const float *dirs, const MlpPacks &w, float *sigma, float *rgb,
                        hipStream_t stream, const uint32_t *count = nullptr);
// the same on the bf16 matrix cores with 3-way operand splitting (tn_mlp_x3.hip)
void launch_mlp_forward_x3(size_t n, uint32_t samples_per_ray, size_t num_rays, const float *feats, const uint32_t *vi,
                           const float *bc, const float *fieldT, const float *dirs, const MlpPacks &w, float *sigma, float *rgb,
                           hipStream_t stream, const uint32_t *count = nullptr);
// Training (tn_mlp.hip: TRAIN variant of the forward kernel, tn_mlp_bwd.hip, tn_mlp_grad.hip).  The training forward SAVES
// the layer inputs and the ReLU masks; the backward kernel runs the reverse network from the masks alone (no recompute);
// the parameter-gradient GEMMs contract the saved inputs with the gradients it leaves.  Device memory owned by the caller;
// the [F, n] tensors are QUAD-major, [F / 4][n][4] floats (tn_mlp_common.h).
struct MlpBackwardBuffers {
    float *x0;                 // [64, n]   gathered features (layer-1 input)                              forward -> grads
    float *h1, *h2, *h3, *h4;  // [128, n]  layer outputs after ReLU (inputs of the next layer)             forward -> grads
    unsigned long long *masks; // [4, n, 2] ReLU masks of h1..h4: bit j of word (layer, sample, half) = slot j   forward -> backward
    float *d1, *d2, *d3, *d4;  // [128, n]  gradients w.r.t. the pre-activations of layers 1, 2, 3 and the head layer   backward -> grads
    float *dhead;              // [4, n]    d sigma_raw, d rgb_raw[0..2]                                   backward -> grads
    float *dx0;                // [n, 64]   gradient w.r.t. the gathered features, sample-major rows       backward -> gather adjoint
};
// launch_mlp_forward in fp32 with GATHER, which also fills x0, h1..h4 and masks of `save`
void launch_mlp_forward_train(size_t n, uint32_t samples_per_ray, size_t num_rays, const uint32_t *vi, const float *bc,
                              const float *fieldT, const float *dirs, const MlpPacks &w, float *sigma, float *rgb,
                              const MlpBackwardBuffers &save, hipStream_t stream);
// dX chain from the saved masks and the forward's OUTPUTS sigma [n] / rgb [n, 3] (softplus' = 1 - exp(-sigma),
// sigmoid' = rgb (1 - rgb)); d_sigma [n], d_rgb [n, 3]; fills d1..d4, dhead, dx0
void launch_mlp_backward(size_t n, const float *sigma, const float *rgb, const MlpPacks &w, const float *d_sigma, const float *d_rgb,
                         const MlpBackwardBuffers &b, hipStream_t stream);
// gradient of the per-ray head bias: out [rays, 128] = sum over the ray's samples of d4 (the gradient w.r.t. the head
// layer's pre-activation, left by launch_mlp_backward)
void launch_ray_head_grad(size_t n, uint32_t samples_per_ray, const float *d4, float *out, hipStream_t stream);
// parameter gradients (tn_mlp_grad.hip), ACCUMULATED into the twelve tensors (nn.Linear layout: w1 [128,64], b1, w2, b2, w3,
// b3 [128..], wd [1,128], bd [1], wh [128,155], bh, wr [3,128], br [3]) from the buffers launch_mlp_backward left for the
// same samples; bit-reproducible (no atomics).  dirs: the ray directions of the chunk.
struct MlpParamGrads { float *w1, *b1, *w2, *b2, *w3, *b3, *wd, *bd, *wh, *bh, *wr, *br; };
size_t mlp_param_grad_scratch_floats();
void launch_mlp_param_grads(size_t n, uint32_t samples_per_ray, const float *dirs, const MlpPacks &w, const MlpBackwardBuffers &b,
                            const MlpParamGrads &g, hipStream_t stream);
// background colour of the RGB renderer (RGBRenderer.combine_rgb: comp + background (1 - accumulation)) and its evaluation-mode
// behaviour (RGBRenderer.forward when not training: nan_to_num of the sample colours, result clamped to [0, 1])
struct Background { float r, g, b; int clamp; };
__host__ __device__ __forceinline__ float nan_to_num(float x) {
    return x != x ? 0.f : (x > 3.4028234663852886e38f ? 3.4028234663852886e38f : (x < -3.4028234663852886e38f ? -3.4028234663852886e38f : x));
}
// adjoint of launch_composite: d sigma [R,S], d rgb [R,S,3] from the gradients of the rendered rgb / accumulation
void launch_composite_backward(size_t R, uint32_t S, const float *sigma, const float *rgb, const float *edges, Background background,
                               const float *d_out_rgb, const float *d_out_acc, float *d_sigma, float *d_rgb, hipStream_t stream);
void launch_transpose(const float *in, float *out, uint32_t rows, uint32_t cols, hipStream_t stream);
// everything between trace_rays and the frame as ONE persistent launch (tn_render_rays.hip): coarse sampler -> match -> gather +
// MLP (density) -> weights -> PDF sampler -> match -> gather + MLP + heads -> composite, scattered into out_* (arrays over ALL rays)
// at the hitting rays ray_index[0 .. *count) (count null: r_max); S_fine = 0: one pass.  dirs [R_all, 3], ray_bias [R_all, 128] or
// null are indexed by ray.  scratch: render_rays_scratch_floats(...) floats, laid out by the RenderRaysLayout it fills.
struct RenderRaysLayout { uint32_t T; size_t per_block, o_edges_f, o_hterm, o_vi, o_bc, o_sigma, o_rgb, o_enc; };
size_t render_rays_scratch_floats(size_t r_max, uint32_t S, uint32_t S_fine, bool has_bias, unsigned grid, RenderRaysLayout &L);
void launch_render_rays(const uint32_t *num_visited, const float *dist, const float *bary, const uint32_t *verts, uint32_t M,
                        const uint32_t *ray_index, const uint32_t *count, size_t r_max, uint32_t S, uint32_t S_fine, bool biased,
                        const float *lin, const float *u_table, float hist_pad, float eps, const float *fieldT, const float *dirs,
                        const float *ray_bias, const MlpPacks &w, Background background, float *out_rgb, float *out_acc, float *out_depth,
                        float *scratch, const RenderRaysLayout &L, unsigned grid, hipStream_t stream,
                        unsigned long long *prof = nullptr /* [8] debug: 100 MHz ticks per phase kind, summed over blocks */,
                        int mode = 0 /* 0: fp32 MFMA, 1: bf16x3 (the MLP phases run x3::forward_group) */);
// ray samplers (tn_samplers.hip): one wavefront per hitting ray, trace rows read in place through ray_index
// count (nullable, every launcher below): the number of hitting rays lives on the device; r / R is the upper bound the grid is sized for
void launch_sample_coarse(size_t r, uint32_t S, uint32_t M, const uint32_t *ray_index, const uint32_t *num_visited, const float *hit_dist,
                          const float *lin, const float *t_rand, bool biased, float *edges, float *near_far, hipStream_t stream,
                          const uint32_t *count = nullptr);
void launch_sample_pdf(size_t r, uint32_t S, uint32_t num_fine, const float *edges, const float *weights, const float *near_far,
                       const float *u_table, const float *u_rand, float histogram_padding, float eps, float *out, hipStream_t stream,
                       const uint32_t *count = nullptr);
// ray_index (nullable): out_rgb / out_acc / out_depth are arrays over ALL rays of the call and row q is written at ray_index[q]
void launch_composite(size_t R, uint32_t S, const float *sigma, const float *rgb, const float *edges, Background background,
                      float *out_rgb, float *out_acc, float *out_depth, float *out_weights, hipStream_t stream,
                      const uint32_t *ray_index = nullptr, const uint32_t *count = nullptr);
// stable partition of the rays by num_visited > 0 (tn_samplers.hip): order [R], *count, padded [R] (nullable: order with the
// entries beyond count replaced by order[0]); scratch: compact_scratch_u32(R) uint32
size_t compact_scratch_u32(size_t R);
void launch_compact_hits(size_t R, const uint32_t *num_visited, uint32_t *order, uint32_t *count, uint32_t *padded, uint32_t *scratch,
                         hipStream_t stream);

// uint32-indexed gather / EMA scatter (tn_uint32.hip); elem_size 4 = f32, 8 = f64
void launch_gather_uint32(int elem_size, uint32_t num_values, uint32_t num_indices, const uint32_t *indices,
                          const void *values, void *result, hipStream_t stream);
void launch_scatter_ema_uint32(int elem_size, uint32_t num_result, uint32_t num_indices, const uint32_t *indices,
                               double decay, const void *values, void *result, hipStream_t stream);

}  // namespace tn
