/*
 * tetranerf_hip.h -- C-ABI of libtetranerf_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the ray -> tetrahedra hot path of jkulhanek/tetra-nerf.
 * Every entry point replaces one native function the reference's pybind11 module
 * (`tetranerf_cpp_extension`, /root/reference/src/py_binding.cpp:433-449) binds; the
 * reference interface each one stands in for is cited at the declaration.
 *
 * Conventions
 *   - plain C, raw DEVICE pointers and sizes, no torch / HIP types in signatures;
 *     `stream` is a hipStream_t passed as void* (NULL = the null stream);
 *   - every function returns 0 on success, non-zero on error; the message is available
 *     from tn_last_error() (thread-local) and mirrors the reference's exception text where
 *     callers can observe it (src/utils/exception.h:164-181 -> Python RuntimeError);
 *   - all work is enqueued on `stream`; no call synchronises the device except
 *     tn_load_tetrahedra (one-off build, like TetrahedraStructure::build's blocking copies,
 *     src/tetrahedra_tracer.cpp:255-259);
 *   - the caller owns every input/output buffer.  The tracer owns only what it derives
 *     from the mesh (face tables, acceleration structures, scratch); the vertex and cell
 *     buffers are borrowed for the tracer's lifetime (src/tetrahedra_tracer.h:300-303);
 *   - output buffers need NOT be initialised: each kernel writes every byte of its outputs
 *     exactly once (the reference memsets them first, py_binding.cpp:53-57,188-191).
 *   - indices are uint32 with 0xFFFFFFFF = "empty" (int32 -1 on the torch side).
 */
#ifndef TETRANERF_HIP_H
#define TETRANERF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tn_tracer *tn_tracer_t;

/* thread-local text of the last error raised on this thread ("" if none) */
const char *tn_last_error(void);

/* library / build identification: "tetranerf_hip <version> abi <TN_ABI_VERSION> gfx950" */
const char *tn_version(void);

/* Number of this header's binary interface: raised whenever an exported signature changes (round 5 added count / ray_index
 * pointers to tn_find_matched_cells_indexed, tn_mlp_forward_gather, tn_composite, tn_sample_coarse, tn_sample_pdf and removed
 * tn_render_pass).  A binding compares tn_abi_version() of the library it loaded with the TN_ABI_VERSION it was written
 * against and refuses a mismatch (tetra-nerf_amd/_lib.py does) instead of calling through shifted arguments. */
#define TN_ABI_VERSION 6
int tn_abi_version(void);

/* TetrahedraTracer::TetrahedraTracer(int8_t device)   src/tetrahedra_tracer.h:284-378,
 *                                                       src/tetrahedra_tracer.cpp:90-135
 * PyTetrahedraTracer ctor                               src/py_binding.cpp:30-36 */
int tn_tracer_create(int device, tn_tracer_t *out);

/* TetrahedraTracer::~TetrahedraTracer                  src/tetrahedra_tracer.cpp:178-189 */
int tn_tracer_destroy(tn_tracer_t tracer);

/* TetrahedraTracer::load_tetrahedra -> TetrahedraStructure::build
 *   src/tetrahedra_tracer.h:304-309, src/tetrahedra_tracer.cpp:244-340;
 *   face table = convert_tetrahedra_to_triangles, src/tetrahedra_tracer.cpp:45-71
 * xyz   f32 [V,3] device, cells u32 [T,4] device (borrowed).
 * The structures are built ON THE DEVICE from these buffers (csrc/tn_build.hip; option "gpu_build" = 0 selects the
 * single-threaded host build of csrc/tn_mesh.cpp, which produces identical face tables, walk records and hull tree).
 * Blocking, like the reference's.
 * Errors: "A triangle is shared by more than two tetrahedra!", "cells contains a vertex index that is out of bounds" */
int tn_load_tetrahedra(tn_tracer_t tracer, size_t num_vertices, size_t num_cells,
                       const float *xyz, const uint32_t *cells, void *stream);

/* number of unique faces of the loaded mesh (0 before load) */
size_t tn_num_faces(tn_tracer_t tracer);

/* copies the face tables to HOST buffers (test/debug aid; faces u32 [F,3] in first-seen
 * order with the unsorted first-seen triple, face_tets u32 [F,2]) */
int tn_get_faces(tn_tracer_t tracer, uint32_t *faces_host, uint32_t *face_tets_host);

/* test aid: copies one structure load_tetrahedra built to a HOST buffer.  which: 0 faces, 1 face_tets, 2 walk records
 * (64 B each), 3 hull nodes, 4 hull triangles, 5 BVH child rows, 6 BVH boxes, 7 BVH leaf ids, 8 BVH leaf triangles.
 * *bytes receives the size in bytes; dst may be NULL (size query). */
int tn_get_build_table(tn_tracer_t tracer, int which, void *dst, size_t *bytes);

/* TetrahedraTracer::trace_rays                         src/tetrahedra_tracer.h:311-331,
 *   TraceRaysPipeline::trace_rays                       src/tetrahedra_tracer.cpp:137-176,
 *   device programs                                      src/optix/optix_trace_rays.cu:268-331
 *   launch-parameter block `Params`                     src/optix_types.h:1-14
 * origins, directions f32 [R,3]; max_ray_triangles = M must be a power of two
 *   ("max_ray_triangles must be a power of 2.", py_binding.cpp:44-47).
 * num_visited u32 [R]; visited u32 [R,M]; bary f32 [R,M,2,3]; dist f32 [R,M,2];
 * verts u32 [R,M,4] (nullable, cf. optix_trace_rays.cu:220).
 * Slots >= num_visited[r]: visited/verts = 0xFFFFFFFF, bary/dist = 0. */
int tn_trace_rays(tn_tracer_t tracer, size_t num_rays, uint32_t max_ray_triangles,
                  const float *origins, const float *directions, uint32_t *num_visited,
                  uint32_t *visited, float *bary, float *dist, uint32_t *verts, void *stream);

/* tn_trace_rays with per-call flags (no reference counterpart: the reference always materialises the dense rows).
 * TN_TRACE_COMPACT_ROWS: slots >= num_visited[r] are left UNWRITTEN (and rows of rays that miss the mesh untouched but
 * for num_visited[r] = 0) -- for consumers that read the rows only through num_visited (tn_sample_*, tn_render_rays,
 * tn_find_matched_cells_indexed): 52 B per segment instead of 52*M B per ray.  A per-call argument rather than a tracer
 * option, so that threads sharing a tracer (nerfstudio's viewer and trainer share the model) cannot see each other's
 * choice.  Calls on ONE tracer handle are serialised inside the library (a per-tracer mutex around the host section: the
 * tracer's scratch buffers, counters, side streams and events are shared state); their kernels queue on the streams. */
#define TN_TRACE_COMPACT_ROWS 1u
int tn_trace_rays_ex(tn_tracer_t tracer, size_t num_rays, uint32_t max_ray_triangles,
                     const float *origins, const float *directions, uint32_t *num_visited,
                     uint32_t *visited, float *bary, float *dist, uint32_t *verts, uint32_t flags, void *stream);

/* TetrahedraTracer::trace_rays_triangles                src/tetrahedra_tracer.h:333-352,
 *   device programs                                      src/optix/optix_trace_rays_triangles.cu:50-115
 * the sorted all-hits list of each ray, without the pairing stage (not called by the model).
 * visited u32 [R,M] face ids; bary f32 [R,M,2] = (u,v); dist f32 [R,M] = t; verts u32 [R,M,3] = the face's
 * stored vertex triple.  Slots >= num_visited[r]: 0 in every array (the reference's torch::zeros defaults). */
int tn_trace_rays_triangles(tn_tracer_t tracer, size_t num_rays, uint32_t max_ray_triangles,
                            const float *origins, const float *directions, uint32_t *num_visited,
                            uint32_t *visited, float *bary, float *dist, uint32_t *verts, void *stream);

/* TetrahedraTracer::find_tetrahedra                      src/tetrahedra_tracer.h:354-366,
 *   device programs                                      src/optix/optix_find_tetrahedra.cu:84-212
 * point location by two closest-hit rays (+x, -x): tetrahedra u32 [N] (0xFFFFFFFF = outside),
 * bary f32 [N,3] (weights of verts[1..3]; verts[0] has 1 - sum), verts u32 [N,4]; not found: bary/verts = 0. */
int tn_find_tetrahedra(tn_tracer_t tracer, size_t num_points, const float *positions, uint32_t *tetrahedra,
                       float *bary, uint32_t *verts, void *stream);

/* find_matched_cells                                   src/tetrahedra_tracer.h:380-393,
 *                                                       src/tetrahedra_tracer.cu:115-193
 * (PyTetrahedraTracer::find_visited_cells, src/py_binding.cpp:163-216; `cells` of the
 *  reference signature is unused there and omitted here)
 * distances f32 [R,S] ascending per ray.  Outputs: cells_out u32 [R,S] (default
 * 0xFFFFFFFF), verts_out u32 [R,S,4] (0xFFFFFFFF), mask_out u8 [R,S] (0), bary_out
 * f32 [R,S,3] (0). */
int tn_find_matched_cells(size_t num_rays, size_t num_samples, size_t max_visited_cells,
                          const uint32_t *num_visited, const uint32_t *visited,
                          const float *dist, const float *bary, const float *distances,
                          const uint32_t *verts, uint32_t *cells_out, uint32_t *verts_out,
                          uint8_t *mask_out, float *bary_out, void *stream);

/* The same for a SUBSET of the traced rays without compacting their rows first (addition): sample row r of
 * distances / outputs ([R,S...]) is matched against trace row ray_index[r] of the [*,M] arrays.  The model
 * compacts the 26 KB rows of the hitting rays with boolean indexing (model.py:560-567); this reads them in place. */
int tn_find_matched_cells_indexed(size_t num_rays, size_t num_samples, size_t max_visited_cells,
                                  const uint32_t *ray_index, const uint32_t *num_visited, const uint32_t *visited,
                                  const float *dist, const float *bary, const float *distances,
                                  const uint32_t *verts, uint32_t *cells_out, uint32_t *verts_out,
                                  uint8_t *mask_out, float *bary_out, const uint32_t *count /* see tn_compact_hits */,
                                  void *stream);

/* Compaction of the hitting rays ON THE DEVICE (addition; the reference compacts with boolean indexing, model.py:540-567 --
 * a device -> host synchronisation per call).  A stable partition of the rays of a trace call by num_visited > 0:
 * order u32 [R]: order[0 .. *count) = the rays that hit the mesh, in ray order, order[*count .. R) = the others, in ray order;
 * count u32 [1] stays in DEVICE memory; padded u32 [R] (nullable) = order with the entries from *count on replaced by
 * order[0] (a full-size index whose tail names a valid ray: what a padded, sync-free batch is gathered with).
 * scratch: 2 * ceil(R / 2048) uint32 of device memory (scratch_len = its length).
 * `count` ARGUMENTS of the entry points below (tn_sample_coarse, tn_sample_pdf, tn_find_matched_cells_indexed,
 * tn_mlp_forward_gather, tn_composite, tn_render_rays): a nullable device pointer to the number of rays to process; when it is
 * given, the size argument (num_hit_rays / num_rays / n) is only the UPPER BOUND the launch is sized for and the outputs are
 * allocated for -- rows from *count on are left unwritten -- so that nothing on the host ever waits for the ray count. */
int tn_compact_hits(size_t num_rays, const uint32_t *num_visited, uint32_t *order, uint32_t *count, uint32_t *padded,
                    uint32_t *scratch, size_t scratch_len, void *stream);

/* interpolate_values<D>                                src/tetrahedra_tracer.h:395-402,
 *                                                       src/tetrahedra_tracer.cu:195-221,250-266
 * vertex_indices u32 [n,D], barycentric f32 [n,D-1], field f32 [F,V] feature-major,
 * result f32 [F,n].  D in {2,3,4,6}, else
 * "Unsupported interpolation dimension with value <D>" (py_binding.cpp:258-276). */
int tn_interpolate_values(uint32_t interpolation_dim, uint32_t num_vertices,
                          uint32_t num_values, uint32_t field_dim,
                          const uint32_t *vertex_indices, const float *barycentric,
                          const float *field, float *result, void *stream);

/* interpolate_values_backward<D>                       src/tetrahedra_tracer.h:404-411,
 *                                                       src/tetrahedra_tracer.cu:223-248,268-290
 * grad_in f32 [F,n] (the transposed contiguous copy py_binding.cpp:369 makes),
 * field_grad_out f32 [F,V]: fully written (zero + scatter-add). */
int tn_interpolate_values_backward(uint32_t interpolation_dim, uint32_t num_vertices,
                                   uint32_t num_values, uint32_t field_dim,
                                   const uint32_t *vertex_indices, const float *barycentric,
                                   const float *grad_in, float *field_grad_out, void *stream);

/* The same with the gradient in the layout autograd hands over, grad_rows f32 [n,F] (sample-major):
 * saves the transposed copy of py_binding.cpp:369 (addition to the reference surface). */
int tn_interpolate_values_backward_rows(uint32_t interpolation_dim, uint32_t num_vertices,
                                        uint32_t num_values, uint32_t field_dim,
                                        const uint32_t *vertex_indices, const float *barycentric,
                                        const float *grad_rows, float *field_grad_out, void *stream);

/* Vertex-major variants (additions): the reference keeps its field feature-major [F,V] (checkpoint layout,
 * model.py:269-271), which makes every vertex row a 4-byte gather; these take a [V,F] shadow copy that the caller
 * refreshes once per field version with tn_transpose_f32 -- no per-call O(V) transposition and no temporaries.
 *   tn_transpose_f32: in [rows, cols] -> out [cols, rows]
 *   tn_interpolate_values_vm: field_vm f32 [V,F]; result f32 [F,n] as tn_interpolate_values
 *   tn_interpolate_values_backward_vm: grad_rows f32 [n,F]; field_grad_vm f32 [V,F] is ACCUMULATED into
 *     (zero it first; the adjoint of tn_transpose_f32 brings it back to [F,V]) */
int tn_transpose_f32(uint32_t rows, uint32_t cols, const float *in, float *out, void *stream);
int tn_interpolate_values_vm(uint32_t interpolation_dim, uint32_t num_values, uint32_t field_dim,
                             const uint32_t *vertex_indices, const float *barycentric, const float *field_vm,
                             float *result, void *stream);
int tn_interpolate_values_backward_vm(uint32_t interpolation_dim, uint32_t num_values, uint32_t field_dim,
                                      const uint32_t *vertex_indices, const float *barycentric,
                                      const float *grad_rows, float *field_grad_vm, void *stream);
/* The same WITHOUT float atomics (addition; the reference's kernel, tetrahedra_tracer.cu:223-247, adds with atomicAdd and so
 * does tn_interpolate_values_backward_vm): the (sample, vertex) pairs are sorted by vertex and every gradient element is
 * summed by one writer in a fixed order -- bit-identical from run to run, 3-5x the time.  Needs num_vertices. */
int tn_interpolate_values_backward_vm_det(uint32_t interpolation_dim, uint32_t num_vertices, uint32_t num_values,
                                          uint32_t field_dim, const uint32_t *vertex_indices, const float *barycentric,
                                          const float *grad_rows, float *field_grad_vm, void *stream);

/* Test aid: run only the dedupe / pairing / tail-fill stage
 * (post_process_tetrahedra, src/optix/optix_trace_rays.cu:110-266) on caller-supplied
 * sorted hit rows: hit_count u32 [R], hit_ids u32 [R,M], hit_t f32 [R,M], hit_uv f32 [R,M,2]. */
int tn_postprocess_hits(tn_tracer_t tracer, size_t num_rays, uint32_t max_ray_triangles,
                        const uint32_t *hit_count, const uint32_t *hit_ids, const float *hit_t,
                        const float *hit_uv, uint32_t *num_visited, uint32_t *visited,
                        float *bary, float *dist, uint32_t *verts, void *stream);

/* the same on caller-supplied face tables (faces u32 [F,3], face_tets u32 [F,2]) instead of a loaded mesh's:
 * lets hit lists that do not come from a mesh at hand -- the reference's tests/test_sort.py vectors -- be paired */
int tn_postprocess_hits_tables(int device, size_t num_rays, uint32_t max_ray_triangles, const uint32_t *faces,
                               const uint32_t *face_tets, const uint32_t *hit_count, const uint32_t *hit_ids,
                               const float *hit_t, const float *hit_uv, uint32_t *num_visited, uint32_t *visited,
                               float *bary, float *dist, uint32_t *verts, void *stream);

/* per-call statistics of the last tn_trace_rays on this tracer (host values; forces a
 * stream sync).  stats[0] = rays certified by the adjacency walk, stats[1] = all other rays (literal pairing of
 * their logged hits, or re-traced by the general all-hits path), stats[2] = rays whose post-process ran the serial
 * literal branch, stats[3] = rays that overflowed M-1 hits. */
int tn_trace_stats(tn_tracer_t tracer, uint64_t stats[4]);

/* diagnostic: why the adjacency walk did not certify rays of the last tn_trace_rays.
 * reasons[k], k = 1..12: 1 zero edge function / zero determinant on a hull face, 2 not exactly two
 * hull crossings, 3 equal hull distances, 4 a vertex of a visited tet within rounding distance of the ray,
 * 5 zero edge function, 6 not exactly two crossed faces in a tet, 7 a gap below eps / a tie / an inversion in t
 * (the chain is sound, its order is not certified), 9 more than M-1 faces, 10 invalid t after a valid one,
 * 11 exit face mismatch, 12 step limit, 8 fold guard: a tet whose neighbourhood holds a tet thinner than 32 rounding
 * distances AND one of whose edges passes within 8 rounding distances of the ray (csrc/tn_trace_walk.hip header).
 * Reason 7 rays keep their logged hits, which go through the literal sort + pairing (reasons[13] counts them);
 * all others are re-traced through the BVH all-hits path.  With option "verify_stride": reasons[15] = certified rays
 * cross-checked against a count-only BVH traversal, reasons[14] = those whose face count differed from the walk's
 * (re-traced through the BVH path; never observed). */
int tn_trace_flag_reasons(tn_tracer_t tracer, uint64_t reasons[16]);

/* The cross-check of the walk's certification in the last tn_trace_rays (host values; forces a stream sync).  Two populations of
 * CERTIFIED rays are re-counted by a count-only BVH all-hits traversal (a differing count re-traces the ray through the BVH path):
 *   the blind sample   every verify_stride-th ray:                      out[0] = stride, out[1] = checked, out[2] = mismatches
 *   the risk classes   EVERY ray inside the wide band (16 rounding distances by default; the guards that hand a ray over act at 8) of
 *                      a hull edge (out[3] = rays of that class) or of an edge of a thin-neighbourhood tet (out[4]):
 *                      out[5] = checked (those not already in the blind sample), out[6] = mismatches
 * See csrc/tn_trace_walk.hip (edge_band) and DESIGN.md section 2.  Option "verify_risk" = 0 switches the risk classes off. */
int tn_trace_cross_check(tn_tracer_t tracer, uint64_t out[8]);

/* Per-kernel breakdown of the last one-chunk walk call traced with option "timing" = 1 (measurement aid; bench.py prints it):
 * with that option the kernels of a call are enqueued on the CALLER's stream in program order with a timing event after each
 * (a normal call overlaps them on four streams, so its parts do not add up to its duration).  ms[0..7] = speculative tail fill,
 * adjacency walk, BVH re-trace of the fallback rays, count cross-check, segment writer, literal pairing of the logged hits,
 * tail fill, re-trace of cross-check mismatches.  Waits for the call to finish. */
int tn_trace_timings(tn_tracer_t tracer, float ms[8]);

/* The constant tails of the dense reference rows (py_binding.cpp:53-57: torch::zeros / full(-1) of the five outputs) for slots
 * [first_slot & ~31, M) of EVERY row: visited / verts = 0xFFFFFFFF, bary / dist = 0.  The tracer's own tail-fill kernel as a
 * stand-alone op: what a caller that traced with TN_TRACE_COMPACT_ROWS runs if it later needs dense rows, and what bench.py
 * times to learn the write rate THIS box sustains (the ceiling of a trace_rays call, whose bytes are 88 % constant tails). */
int tn_fill_rows(size_t num_rays, uint32_t max_ray_triangles, uint32_t first_slot, uint32_t *visited, float *bary, float *dist,
                 uint32_t *verts, void *stream);

/* knobs ("walk" and "gpu_build" also through the environment: TETRANERF_HIP_WALK, TETRANERF_HIP_GPU_BUILD):
 *   "walk"    1 = adjacency-walk fast path with general-path fallback (default),
 *             0 = general all-hits path for every ray, 2 = walk for any batch size
 *   "walk_min_rays"  smallest batch the walk is used for (default 12288; 8192 from 2M tets, 6144 from 4M tets on until
 *             the option is set; below it one wavefront per
 *             ray through the wide BVH has the lower latency).  Whatever the options say, the walk path needs
 *             max_ray_triangles >= 4 (16-byte stores into the rows); 1 and 2 take the BVH path
 *   "dense_tails"  1 (default) = every slot of the [R,M] rows is written, as the reference does;
 *             0 = slots >= num_visited[r] of walked rows are left UNWRITTEN (non-reference: for callers
 *             that only read rows through num_visited, e.g. tn_find_matched_cells; saves ~88 % of the bytes).
 *             Tracer-wide; prefer the per-call flag TN_TRACE_COMPACT_ROWS of tn_trace_rays_ex
 *   "literal" 1 (default) = rays whose order the walk cannot certify have their logged hits sorted and paired
 *             literally; 0 = they are re-traced through the BVH all-hits path (cross-check of the two paths)
 *   "spec_fill"  1 = the last quarter / half of every row (slots no ray, or hardly any ray, of this mesh reaches) is filled
 *             beside the walk on a stream of its own (the schedule of rounds 2-5); 0 (default since round 6) = the whole
 *             tail fill after the segment writer.  "spec_k0" = first speculatively filled slot (multiple of 32; 0 = the
 *             rule of tn_api.hip) -- tests force it low so that rays of every class overwrite speculatively filled slots;
 *             "spec_blocks" (512) = grid of that fill, "walk_lds_kb" (26) = dynamic LDS reserved per walk block beside it
 *   "fill_blocks"  grid of the tail fill: -1 (default) = one block per row (k_fill_rows_fine), -2 = one linear stream per
 *             array with one store per thread (k_fill_linear: what tn_fill_rows uses), n > 0 = n blocks of persistent waves
 *   "writer_blocks"  grid of the segment writer (0 = default: 2 blocks per CU, what is resident at once)
 *   "hull_flat"  1 (default) = the walk's entry search (k_hull_entry) goes through the flat box table staged in LDS when the
 *             hull has at most 1024 faces; 0 = the threaded hull tree for every hull size (tests, A/B)
 *   "cert_ends"  statement of the walk's order test: 0 round 5's pairwise test, 3 the same + the end-of-chain rules A-C,
 *             1 the cluster test with rules A-D, 2 (default) = by mesh size.  Moves rays between the segment writer and the
 *             literal pairing kernel, never a byte of a row
 *   "log_cap_mb"  cap of the hit log in MiB (0 = default: a quarter of the free device memory, at most 24 GiB);
 *             larger calls are processed in ray chunks
 *   "gpu_build"  1 (default) = load_tetrahedra builds its structures on the device; 0 = single-threaded host build
 *   "leaf_width" 16 (default) / 32 / 64: faces per leaf of the face BVH (applies at the next tn_load_tetrahedra); the BVH
 *             path tests 64 / leaf_width crossed leaves per wave instruction
 *   "small_lds"  1 (default) = batches below walk_min_rays use LDS hit arrays sized for the mesh (every ray resident at
 *             once) and re-trace the rays with more hits in a second launch; "lds_cap" forces their size (tests)
 *   "verify_stride"  k > 0 (default 256): every k-th ray
 *             the walk certified is cross-checked: a count-only BVH all-hits
 *             traversal must find exactly the faces the walk logged, otherwise the ray is re-traced through the BVH path and
 *             counted in tn_trace_flag_reasons()[14].  A one-chunk call runs the check beside the row writers (< 1 % of the
 *             call; linear in the rays checked: tests and the fuzzer run it at 1); 0 = off.
 *             "verify_inject" 1 = every checked ray counts as a mismatch (tests of the hand-over)
 *   "literal_sort_passes"  (default 8) odd-even transposition passes over the nearly sorted hits the walk logged for a ray
 *             whose order it does not certify, before the bitonic network takes over (same result: distinct keys; tests run 0 and 1)
 *   "verify_risk"  1 (default) = every certified ray of the RISK classes is cross-checked as well (tn_trace_cross_check); 0 = only
 *             the blind sample.  "risk_band" (default 2) = width of the classes' band in units of the guards' own 8 rounding
 *             distances (2: rays between 8 and 16; measured cost in DESIGN.md section 2)
 *   "timing"  1 = serialise the kernels of a one-chunk walk call on the caller's stream with timing events (tn_trace_timings);
 *             0 (default) = the overlapped four-stream schedule
 *   "writer_table"  0 (default) = the segment writer's record table by mesh size (one record per (tet, entry face) below
 *             500k tets, one per tet above), 1 / 2 force either (applies at the next tn_load_tetrahedra; tests, A/B)
 * Unknown names are an error. */
int tn_set_option(tn_tracer_t tracer, const char *name, int value);

/* gather_uint32<T> / scatter_ema_uint32<T>              src/tetrahedra_tracer.cu:30-113,
 *                                                       src/py_binding.cpp:374-431
 * (exported by the reference, not called by its model).  elem_size 4 = float32, 8 = float64.
 * gather: result[i] = values[indices[i]];  scatter: result[k] = result[k]*decay + (1-decay)*values[i], k = indices[i],
 * applied atomically per element (compare-and-swap).  Out-of-range indices are skipped. */
int tn_gather_uint32(int elem_size, uint32_t num_values, uint32_t num_indices, const uint32_t *indices,
                     const void *values, void *result, void *stream);
int tn_scatter_ema_uint32(int elem_size, uint32_t num_result, uint32_t num_indices, const uint32_t *indices,
                          double decay, const void *values, void *result, void *stream);

/* ---- shallow MLP + volume render (inference forward) ------------------------------------------
 * The arithmetic of these two lives in nerfstudio, not in /root/reference; the call sites are
 * tetranerf/nerfstudio/model.py:414-455 (modules), :602-621 (MLP + heads), :632-638 (weights and
 * renderers).  Weights use the nn.Linear layout [out, in], row-major fp32, i.e. what a reference
 * checkpoint holds (mlp_base.layers.{0,1,2}, field_output_density.net, mlp_head.layers.0,
 * field_output_color.net). */
typedef struct tn_mlp_weights {
    const float *w1, *b1; /* [128,64],  [128] */
    const float *w2, *b2; /* [128,128], [128] */
    const float *w3, *b3; /* [128,128], [128] */
    const float *wd, *bd; /* [1,128],   [1]    density head  */
    const float *wh, *bh; /* [128,155], [128]  mlp_head: columns [direction encoding 27 | base 128] */
    const float *wr, *br; /* [3,128],   [3]    rgb head      */
} tn_mlp_weights;

/* A handle owns the packed forms of ONE set of weights (per kernel family: fp32-MFMA forward with / without the fused
 * gather, bf16x3 pieces, transposed for the backward pass) and the small per-call scratch (direction encodings): the
 * entry points below neither pack nor allocate.  tn_mlp_set_weights packs; call it once per parameter VERSION (after
 * an optimiser step, after loading a checkpoint).  One handle serves one stream at a time. */
typedef struct tn_mlp *tn_mlp_t;
int tn_mlp_create(int device, tn_mlp_t *out);
int tn_mlp_destroy(tn_mlp_t mlp);
int tn_mlp_set_weights(tn_mlp_t mlp, const tn_mlp_weights *weights, void *stream);

/* Arithmetic `mode` of the two forward entry points (per call):
 *   0  v_mfma_f32_32x32x2_f32: an exact fp32 fma chain (157 TFLOP/s peak) -- what every parity claim refers to;
 *   1  "bf16x3": v_mfma_f32_32x32x16_bf16 on operands split into three bf16 pieces, six partial products per
 *      multiply, fp32 accumulation: dropped terms < 2^-24 of a product at 2.67x fewer matrix-core cycles (opt-in).
 * feats f32 [64, n] feature-major (the buffer tn_interpolate_values writes), dirs f32 [n/samples_per_ray, 3]
 * (one direction per ray; samples of a ray are consecutive) -> sigma f32 [n] (softplus), rgb f32 [n,3] (sigmoid).
 * rgb == NULL: density only (mlp_base + density head: the coarse pass of the model, model.py:577-581); dirs unused. */
int tn_mlp_forward(tn_mlp_t mlp, size_t n, uint32_t samples_per_ray, const float *feats, const float *dirs, int mode,
                   float *sigma, float *rgb, void *stream);

/* the same with the barycentric gather fused in (tn_interpolate_values<4> + tn_mlp_forward without the
 * [64,n] intermediate): vertex_indices u32 [n,4], barycentric f32 [n,3], field_vm f32 [V,64] VERTEX-major
 * (tn_transpose_f32 of the [64,V] parameter, refreshed by the caller once per field version) */
/* ray_head_bias f32 [n / samples_per_ray, 128] or NULL (here and in tn_mlp_forward_gather_train; tn_render_rays takes rows over ALL rays): a
 * per-RAY vector added to the pre-activation of mlp_head -- the reference's appearance embedding (model.py:437-447,
 * 608-620: head input = cat[encoded_dir, base, embedded_appearance], the embedding constant along a ray), whose E columns
 * of the head GEMM collapse to c_ray = Wh[:, 155:] emb(camera of the ray), an [rays, E] x [E, 128] product the caller
 * makes (and differentiates: tn_mlp_ray_head_grad returns dL/dc_ray).  wh of tn_mlp_weights stays [128, 155]. */
int tn_mlp_forward_gather(tn_mlp_t mlp, size_t n, uint32_t samples_per_ray, const uint32_t *vertex_indices,
                          const float *barycentric, const float *field_vm, const float *dirs, int mode, float *sigma,
                          float *rgb, const float *ray_head_bias, const uint32_t *count /* device-side number of RAYS, nullable */,
                          void *stream);

/* background colour of nerfstudio's RGBRenderer as the reference model configures / overrides it (model.py:466,504-518;
 * `renderers.BACKGROUND_COLOR_OVERRIDE`): comp_rgb + background (1 - accumulation).  clamp != 0 = the renderer's
 * evaluation mode (RGBRenderer.forward when not training): nan_to_num of the sample colours, result clamped to [0, 1].
 * A NULL pointer means white without clamp (the shipped configurations in training mode). */
typedef struct tn_rgb_background { float r, g, b; int clamp; } tn_rgb_background;
/* EVERYTHING between tn_trace_rays and the frame as ONE persistent launch (reference span
 * tetranerf/nerfstudio/model.py:531-662 in evaluation mode): coarse sampler (uniform or biased) -> find_visited_cells ->
 * interpolate_values + mlp_base + density head -> get_weights -> PDFSampler (include_original) -> find_visited_cells ->
 * interpolate_values + mlp_base + heads -> get_weights + RGB / accumulation / median-depth renderers, written into the frame.
 * The ray set is ray_index u32 [num_hit_rays_max] = tn_compact_hits' `order` and its SIZE lives on the device (count u32 [1];
 * NULL: all num_hit_rays_max entries) -- nothing on the host waits for the trace.  The trace rows are read in place; dirs f32
 * [R,3] and ray_head_bias f32 [R,128] (nullable) are arrays over ALL rays, indexed by ray; out_rgb f32 [R,3], out_acc f32 [R],
 * out_depth f32 [R] are written at the hitting rays' own rows (pre-fill them with the background values).
 * num_fine = 0: one pass over the coarse samples.  linspace f32 [S+1], u_table f32 [num_fine+1] (bin-centred quantiles),
 * histogram_padding / eps as in tn_sample_coarse / tn_sample_pdf.  fp32 MFMA arithmetic; every stage runs the same device
 * function as the stand-alone entry points, so the frame is bit-identical to tn_sample_coarse -> tn_find_matched_cells_indexed
 * -> tn_mlp_forward_gather -> tn_composite -> tn_sample_pdf -> ... (csrc/tn_render_rays.hip).
 * Needs max(2 M, 3 S + num_fine + 3) * 32 B <= 160 KB of LDS (M <= 2048 at the shipped sample counts). */
int tn_render_rays(tn_mlp_t mlp, uint32_t max_ray_triangles, const uint32_t *num_visited, const float *hit_distances,
                   const float *barycentric, const uint32_t *vertex_indices, const uint32_t *ray_index, const uint32_t *count,
                   size_t num_hit_rays_max, uint32_t num_samples, uint32_t num_fine, int biased, const float *linspace,
                   const float *u_table, float histogram_padding, float eps, const float *field_vm, const float *dirs,
                   const tn_rgb_background *background, float *out_rgb, float *out_acc, float *out_depth,
                   const float *ray_head_bias, void *stream);
/* ... with the arithmetic of the MLP phases as a per-call argument like tn_mlp_forward's `mode` (round 6): 0 = fp32 MFMA (what
 * tn_render_rays runs), 1 = bf16x3 (split-operand bf16 MFMA, csrc/tn_mlp_x3.hip; opt-in): then the frame is bit-identical to the
 * kernel chain run with mode 1.  Same reference span (model.py:531-662). */
int tn_render_rays_ex(tn_mlp_t mlp, uint32_t max_ray_triangles, const uint32_t *num_visited, const float *hit_distances,
                   const float *barycentric, const uint32_t *vertex_indices, const uint32_t *ray_index, const uint32_t *count,
                   size_t num_hit_rays_max, uint32_t num_samples, uint32_t num_fine, int biased, const float *linspace,
                   const float *u_table, float histogram_padding, float eps, const float *field_vm, const float *dirs,
                   const tn_rgb_background *background, float *out_rgb, float *out_acc, float *out_depth,
                   const float *ray_head_bias, int mode, void *stream);

/* ---- ray samplers between tn_trace_rays and the render passes (model.py:111-192, 549-557, 582-586; nerfstudio's
 * UniformSampler / PDFSampler for the parts the reference imports).  One wavefront per HITTING ray; the trace rows are
 * read in place through ray_index u32 [r] (as in tn_render_rays).
 * tn_sample_coarse: near / far of every hitting ray (near_far f32 [r,2]: first t_in, last t_out, model.py:531-544) and
 *   its num_samples + 1 coarse bin edges (edges f32 [r, S+1], euclidean): linspace f32 [S+1] = the spacing bins
 *   (torch.linspace(0, 1, S+1): the caller's table, so that both sides use the same values); t_rand f32 [r, S+1] uniform
 *   draws = training-mode stratified bins (model.py:166-175), NULL = evaluation; biased != 0: the edges are re-mapped so
 *   that every visited tetrahedron receives the same share of the samples (map_from_real_distances_to_biased_with_bounds,
 *   model.py:111-122).
 * tn_sample_pdf: PDFSampler with include_original (model.py:463,584): inverse-CDF samples of the coarse weights f32 [r,S]
 *   (+ histogram_padding, eps as in nerfstudio: 0.01, 1e-5) at the num_fine + 1 quantiles u_table f32 [num_fine+1]
 *   (evaluation: bin centres; training: bin starts + u_rand f32 [r, num_fine+1] / (num_fine+1), u_rand NULL otherwise),
 *   merged with the coarse edges: edges_out f32 [r, S + num_fine + 2], sorted, euclidean. */
int tn_sample_coarse(size_t num_hit_rays, uint32_t num_samples, uint32_t max_ray_triangles, const uint32_t *ray_index,
                     const uint32_t *num_visited, const float *hit_distances, const float *linspace, const float *t_rand,
                     int biased, float *edges, float *near_far, const uint32_t *count, void *stream);
int tn_sample_pdf(size_t num_hit_rays, uint32_t num_samples, uint32_t num_fine, const float *edges, const float *weights,
                  const float *near_far, const float *u_table, const float *u_rand, float histogram_padding, float eps,
                  float *edges_out, const uint32_t *count, void *stream);

/* RaySamples.get_weights + RGB (background blend) / accumulation / median-depth renderers.
 * sigma f32 [R,S], rgb f32 [R,S,3], edges f32 [R,S+1] (bin edges: starts = edges[:, :-1], ends = edges[:, 1:]);
 * out_rgb f32 [R,3], out_acc f32 [R], out_depth f32 [R], out_weights f32 [R,S] (nullable).
 * rgb == NULL and out_rgb == NULL: only out_weights is written (get_weights of the coarse pass, model.py:582).
 * ray_index u32 [R] (nullable): out_rgb / out_acc / out_depth are then arrays over ALL rays of the trace call and row q is
 * written at ray_index[q] (the scatter of model.py:640-662 into the frame, without an index_put pass). */
int tn_composite(size_t num_rays, uint32_t num_samples, const float *sigma, const float *rgb, const float *edges,
                 const tn_rgb_background *background, float *out_rgb, float *out_acc, float *out_depth,
                 float *out_weights, const uint32_t *ray_index, const uint32_t *count, void *stream);

/* ---- training: the MLP node and the composite node (PyTorch autograd in the reference: the trainer
 * back-propagates through nerfstudio's MLP / renderers, model.py:602-638).  Three calls per batch of n samples, all on
 * the handle's weights, fp32 MFMA:
 *   tn_mlp_forward_gather_train  = tn_mlp_forward_gather (mode 0) that also SAVES what the backward pass needs:
 *       x0 [64,n] gathered features, h1..h4 [128,n] layer outputs after ReLU (the operands of the weight-gradient
 *       GEMMs) and masks [4,n,2] u64 = the ReLU masks of h1..h4 (all the dX chain needs);
 *   tn_mlp_backward  runs the reverse network from the masks and the forward's OUTPUTS sigma [n] / rgb [n,3]
 *       (softplus' = 1 - exp(-sigma), sigmoid' = rgb (1 - rgb)) -- nothing is recomputed -- given d_sigma f32 [n],
 *       d_rgb f32 [n,3]; fills d1..d4 [128,n] (gradients w.r.t. the pre-activations of mlp_base layers 0..2 and of
 *       mlp_head), dhead [4,n] = d sigma_raw, d rgb_raw[0..2] (four plain rows), and dx0 [n,64] = the gradient of the
 *       gathered features as SAMPLE-major rows (feed it to tn_interpolate_values_backward_vm / _rows);
 *   The [F,n] tensors x0, h1..h4, d1..d4 are opaque to the caller and QUAD-major: [F/4][n][4] floats, element (feature
 *   f, sample s) at ((f / 4) n + s) 4 + f % 4 -- 16-byte stores for the kernels that produce them, 16-byte tile loads for
 *   the GEMMs that consume them.
 *   tn_mlp_param_grads  ACCUMULATES the gradients of the twelve parameter tensors (tn_mlp_grads: fp32 gradient buffers
 *       in nn.Linear layout, zeroed by the caller before the first batch) from those buffers: dW_l = d_l (input of
 *       layer l)^T as sample-streaming fp32-MFMA GEMMs (the 27 direction-encoding columns of mlp_head and the density
 *       head's vector ride along the mlp_head GEMM), the rgb head / bias sums as one bandwidth-bound pass; per-block
 *       partial sums are added in a fixed order (no atomics): bit-reproducible.  dirs f32 [n / samples_per_ray, 3].
 * All buffers are device memory owned by the caller (4.4 KB per sample in total). */
typedef struct tn_mlp_backward_buffers {
    float *x0, *h1, *h2, *h3, *h4;   /* forward -> param_grads */
    void *masks;                     /* forward -> backward: u64 [4, n, 2] */
    float *d1, *d2, *d3, *d4, *dhead, *dx0;   /* backward -> param_grads / gather adjoint */
} tn_mlp_backward_buffers;
int tn_mlp_forward_gather_train(tn_mlp_t mlp, size_t n, uint32_t samples_per_ray, const uint32_t *vertex_indices,
                                const float *barycentric, const float *field_vm, const float *dirs, float *sigma, float *rgb,
                                const tn_mlp_backward_buffers *buffers, const float *ray_head_bias, void *stream);
int tn_mlp_backward(tn_mlp_t mlp, size_t n, const float *sigma, const float *rgb, const float *d_sigma, const float *d_rgb,
                    const tn_mlp_backward_buffers *buffers, void *stream);
/* gradient of the per-ray head bias after tn_mlp_backward: d_ray_head_bias f32 [n / samples_per_ray, 128] = the sum over
 * each ray's samples of d4 (bit-reproducible) */
int tn_mlp_ray_head_grad(size_t n, uint32_t samples_per_ray, const tn_mlp_backward_buffers *buffers, float *d_ray_head_bias,
                         void *stream);
typedef struct tn_mlp_grads { /* same shapes as tn_mlp_weights */
    float *w1, *b1, *w2, *b2, *w3, *b3, *wd, *bd, *wh, *bh, *wr, *br;
} tn_mlp_grads;
int tn_mlp_param_grads(tn_mlp_t mlp, size_t n, uint32_t samples_per_ray, const float *dirs,
                       const tn_mlp_backward_buffers *buffers, const tn_mlp_grads *grads, void *stream);

/* adjoint of tn_composite w.r.t. sigma [R,S] and rgb [R,S,3], given the gradients of the rendered rgb [R,3] and
 * accumulation [R] (either nullable); the median depth carries no gradient. */
int tn_composite_backward(size_t num_rays, uint32_t num_samples, const float *sigma, const float *rgb, const float *edges,
                          const tn_rgb_background *background, const float *d_out_rgb, const float *d_out_acc,
                          float *d_sigma, float *d_rgb, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* TETRANERF_HIP_H */
