// tn_samplers.hip -- the ray samplers between trace_rays and the render passes, as device kernels (row a14 / f1).
//
// Reference: tetranerf/nerfstudio/model.py:111-192 (map_from_real_distances_to_biased_with_bounds + TetrahedraSampler),
// :549-557 (coarse sampling), :582-586 (PDF sampling); nerfstudio's UniformSampler / PDFSampler for the parts the
// reference imports (restated in tetra-nerf_amd/render.py: uniform_sample_bins, biased_sample_bins, pdf_sample_bins --
// those PyTorch statements are the parity definition; the kernels here evaluate the same expressions per element in
// the same order, only the reductions (sum of the padded weights, prefix sums of pdf / segment lengths) are wave scans
// instead of torch's reduction trees: results agree to fp32 round-off of those sums).
//
// One wavefront per HITTING ray (ray_index names its row in the trace outputs, which are read in place):
//   k_sample_coarse  near / far of the ray (model.py:531-544) + the S+1 coarse bin edges: linspace or train-mode
//                    stratified bins (model.py:166-175), mapped to euclidean distances (spacing_to_euclidean, :177) and --
//                    biased = 1 -- re-mapped onto the visited segments (:111-122)
//   k_sample_pdf     inverse-CDF samples of the padded coarse weights at num_fine + 1 quantiles, merged with the coarse
//                    edges (two sorted lists: merge by rank instead of a sort) and mapped to euclidean distances
// so that a render is trace -> [sampler -> pass] x 2 with no PyTorch operator in between.
#include "tn_ray_ops.h"

namespace tn {

using namespace rayops;

namespace {

constexpr int SB = 256;   // 4 rays per block

}  // namespace

// lin [S+1] = torch.linspace(0, 1, S+1) (made once by the caller: the same table the PyTorch statement uses);
// t_rand [r, S+1] uniform draws or null (evaluation); out: edges [r, S+1], near_far [r, 2]
__global__ __launch_bounds__(SB) void k_sample_coarse(size_t r, uint32_t S, uint32_t M, const uint32_t *__restrict__ ray_index,
                                                      const uint32_t *__restrict__ num_visited, const float *__restrict__ hit_dist,
                                                      const float *__restrict__ lin, const float *__restrict__ t_rand, int biased,
                                                      float *__restrict__ edges, float *__restrict__ near_far,
                                                      const uint32_t *__restrict__ count) {
    extern __shared__ float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (count) r = *count;      // device-side number of hitting rays (the grid was sized for an upper bound)
    float *cum = smem + (size_t)wave * (M + 1);          // biased: cum[i] = start + sum of the first i segment lengths
    for (size_t q = (size_t)blockIdx.x * (SB / 64) + wave; q < r; q += (size_t)gridDim.x * (SB / 64)) {
        const size_t ray = ray_index[q];
        ray_sample_coarse(S, M, ray, num_visited[ray], hit_dist, lin, t_rand ? t_rand + q * (size_t)(S + 1) : nullptr, biased,
                          edges + q * (size_t)(S + 1), near_far + 2 * q, cum, lane);
    }
}

// edges [r, S+1] euclidean coarse edges, weights [r, S] coarse weights, near_far [r, 2];
// u_table [nb] (nb = num_fine + 1): evaluation = the bin-centred quantiles, training = the bin starts, to which
// u_rand [r, nb] / nb is added; out [r, S + nb + 1] = the merged, sorted edges mapped back to euclidean distances
__global__ __launch_bounds__(SB) void k_sample_pdf(size_t r, uint32_t S, uint32_t nb, const float *__restrict__ edges,
                                                   const float *__restrict__ weights, const float *__restrict__ near_far,
                                                   const float *__restrict__ u_table, const float *__restrict__ u_rand,
                                                   float histogram_padding, float eps, float *__restrict__ out,
                                                   const uint32_t *__restrict__ count) {
    extern __shared__ float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (count) r = *count;
    float *lds = smem + (size_t)wave * pdf_lds_floats(S, nb);
    for (size_t q = (size_t)blockIdx.x * (SB / 64) + wave; q < r; q += (size_t)gridDim.x * (SB / 64))
        ray_sample_pdf(S, nb, edges + q * (size_t)(S + 1), weights + q * (size_t)S, near_far[2 * q], near_far[2 * q + 1], u_table,
                       u_rand ? u_rand + q * (size_t)nb : nullptr, histogram_padding, eps, out + q * (size_t)(S + 1 + nb), lds, lane);
}

// ---- compaction of the hitting rays on the device (replaces torch.nonzero / boolean indexing, model.py:540-567: no host
// synchronisation).  A stable partition of the rays by num_visited > 0: order[0 .. count) = the rays that hit the mesh, in ray
// order, order[count .. R) = the others, in ray order; *count stays in device memory -- the samplers, the matcher, the MLP
// and the composite kernels take its address and are launched over R.  Two small launches: per-block hit counts, then every
// block sums the counts before it (a few hundred values) and writes its rays.
namespace {
constexpr int CP_BLOCK = 256, CP_PER = 8, CP_RAYS = CP_BLOCK * CP_PER;   // 2048 consecutive rays per block, 8 per thread

__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *sm /*[CP_BLOCK / 64 + 1]*/, uint32_t &total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)inc, off);
        if (lane >= off) inc += o;
    }
    if (lane == 63) sm[wave] = inc;
    __syncthreads();
    uint32_t before = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < CP_BLOCK / 64; ++w) {
        const uint32_t x = sm[w];
        before += w < wave ? x : 0u;
        tot += x;
    }
    total = tot;
    __syncthreads();
    return before + inc - v;
}
}  // namespace

__global__ __launch_bounds__(CP_BLOCK) void k_count_hits(size_t R, const uint32_t *__restrict__ num_visited, uint32_t *__restrict__ block_hits,
                                                         uint32_t *__restrict__ block_first) {
    __shared__ uint32_t sm[CP_BLOCK / 64 + 1];
    __shared__ uint32_t first;
    if (threadIdx.x == 0) first = TN_EMPTY;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * CP_RAYS + (size_t)threadIdx.x * CP_PER;
    uint32_t c = 0, mine = TN_EMPTY;
#pragma unroll
    for (int i = 0; i < CP_PER; ++i) {
        const size_t ray = base + i;
        const bool hit = ray < R && num_visited[ray] > 0;
        if (hit && mine == TN_EMPTY) mine = (uint32_t)ray;
        c += hit ? 1u : 0u;
    }
    if (mine != TN_EMPTY) atomicMin(&first, mine);
    uint32_t total;
    block_excl_scan(c, sm, total);
    if (threadIdx.x == 0) { block_hits[blockIdx.x] = total; block_first[blockIdx.x] = first; }
}

__global__ __launch_bounds__(CP_BLOCK) void k_write_order(size_t R, const uint32_t *__restrict__ num_visited, const uint32_t *__restrict__ block_hits,
                                                          const uint32_t *__restrict__ block_first, uint32_t *__restrict__ order,
                                                          uint32_t *__restrict__ count, uint32_t *__restrict__ padded) {
    __shared__ uint32_t sm[CP_BLOCK / 64 + 1];
    __shared__ uint32_t s_first;
    if (threadIdx.x == 0) s_first = TN_EMPTY;
    __syncthreads();
    // hits in the blocks before this one, hits in all blocks, the first hitting ray of the call
    uint32_t before = 0, all = 0, first = TN_EMPTY;
    for (uint32_t b = threadIdx.x; b < gridDim.x; b += CP_BLOCK) {
        const uint32_t x = block_hits[b];
        before += b < blockIdx.x ? x : 0u;
        all += x;
        const uint32_t f = block_first[b];
        first = f < first ? f : first;
    }
    if (first != TN_EMPTY) atomicMin(&s_first, first);
    uint32_t t0, t1;
    block_excl_scan(before, sm, t0);
    block_excl_scan(all, sm, t1);
    const uint32_t hits_before = t0, total = t1;
    const uint32_t pad_ray = s_first == TN_EMPTY ? 0u : s_first;   // (no ray hits: every entry is padding and names ray 0)
    if (blockIdx.x == 0 && threadIdx.x == 0) *count = total;
    const size_t base = (size_t)blockIdx.x * CP_RAYS + (size_t)threadIdx.x * CP_PER;
    bool hit[CP_PER];
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < CP_PER; ++i) {
        const size_t ray = base + i;
        hit[i] = ray < R && num_visited[ray] > 0;
        c += hit[i] ? 1u : 0u;
    }
    uint32_t tot_block;
    uint32_t h = hits_before + block_excl_scan(c, sm, tot_block);   // hitting rays before this thread's first ray
#pragma unroll
    for (int i = 0; i < CP_PER; ++i) {
        const size_t ray = base + i;
        if (ray >= R) break;
        const size_t pos = hit[i] ? (size_t)h : (size_t)total + (ray - h);   // misses: rank among the misses = ray - hits before it
        order[pos] = (uint32_t)ray;
        if (padded) padded[pos] = hit[i] ? (uint32_t)ray : pad_ray;
        h += hit[i] ? 1u : 0u;
    }
}

size_t compact_scratch_u32(size_t R) { return 2 * ((R + CP_RAYS - 1) / CP_RAYS); }

void launch_compact_hits(size_t R, const uint32_t *num_visited, uint32_t *order, uint32_t *count, uint32_t *padded, uint32_t *scratch,
                         hipStream_t stream) {
    const size_t blocks = (R + CP_RAYS - 1) / CP_RAYS;
    if (R == 0) { (void)hipMemsetAsync(count, 0, sizeof(uint32_t), stream); return; }
    hipLaunchKernelGGL(k_count_hits, dim3((unsigned)blocks), dim3(CP_BLOCK), 0, stream, R, num_visited, scratch, scratch + blocks);
    hipLaunchKernelGGL(k_write_order, dim3((unsigned)blocks), dim3(CP_BLOCK), 0, stream, R, num_visited, scratch, scratch + blocks, order,
                       count, padded);
}

void launch_sample_coarse(size_t r, uint32_t S, uint32_t M, const uint32_t *ray_index, const uint32_t *num_visited, const float *hit_dist,
                          const float *lin, const float *t_rand, bool biased, float *edges, float *near_far, hipStream_t stream,
                          const uint32_t *count) {
    if (r == 0) return;
    const size_t smem = biased ? (size_t)(SB / 64) * (M + 1) * sizeof(float) : 0;
    if (smem > 64 * 1024) throw Error("sample_coarse: max_ray_triangles too large for the biased sampler");
    const size_t blocks = (r + SB / 64 - 1) / (SB / 64);
    hipLaunchKernelGGL(k_sample_coarse, dim3((unsigned)(blocks < 256 * 16 ? blocks : 256 * 16)), dim3(SB), smem, stream, r, S, M, ray_index,
                       num_visited, hit_dist, lin, t_rand, biased ? 1 : 0, edges, near_far, count);
}

void launch_sample_pdf(size_t r, uint32_t S, uint32_t num_fine, const float *edges, const float *weights, const float *near_far,
                       const float *u_table, const float *u_rand, float histogram_padding, float eps, float *out, hipStream_t stream,
                       const uint32_t *count) {
    if (r == 0) return;
    const uint32_t nb = num_fine + 1;
    const size_t smem = (size_t)(SB / 64) * (2 * (S + 1) + nb) * sizeof(float);
    if (smem > 64 * 1024) throw Error("sample_pdf: too many samples per ray");
    const size_t blocks = (r + SB / 64 - 1) / (SB / 64);
    hipLaunchKernelGGL(k_sample_pdf, dim3((unsigned)(blocks < 256 * 16 ? blocks : 256 * 16)), dim3(SB), smem, stream, r, S, nb, edges, weights,
                       near_far, u_table, u_rand, histogram_padding, eps, out, count);
}

}  // namespace tn
