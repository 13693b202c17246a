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
#include "tn_device.h"
#include "tn_kernels.h"

namespace tn {

namespace {

__device__ __forceinline__ float wave_incl_scan(float v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float o = __shfl_up(v, off);
        if (lane >= off) v += o;
    }
    return v;
}
// running maximum over an LDS array of n floats (wave-cooperative): makes a list that is sorted up to rounding
// (an inversion of an ulp between neighbours) non-decreasing, so that the merge by rank below is a permutation
__device__ __forceinline__ void lds_running_max(float *a, uint32_t n, int lane) {
    float carry = -INFINITY;
    for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t k = base + lane;
        float v = k < n ? a[k] : -INFINITY;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const float o = __shfl_up(v, off);
            if (lane >= off) v = fmaxf(v, o);
        }
        v = fmaxf(v, carry);
        if (k < n) a[k] = v;
        carry = __shfl(v, 63);
    }
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ void lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int SB = 256;   // 4 rays per block

}  // namespace

// lin [S+1] = torch.linspace(0, 1, S+1) (made once by the caller: the same table the PyTorch statement uses);
// t_rand [r, S+1] uniform draws or null (evaluation); out: edges [r, S+1], near_far [r, 2]
__global__ __launch_bounds__(SB) void k_sample_coarse(size_t r, uint32_t S, uint32_t M, const uint32_t *__restrict__ ray_index,
                                                      const uint32_t *__restrict__ num_visited, const float *__restrict__ hit_dist,
                                                      const float *__restrict__ lin, const float *__restrict__ t_rand, int biased,
                                                      float *__restrict__ edges, float *__restrict__ near_far) {
    extern __shared__ float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *cum = smem + (size_t)wave * (M + 1);          // biased: cum[i] = start + sum of the first i segment lengths
    for (size_t q = (size_t)blockIdx.x * (SB / 64) + wave; q < r; q += (size_t)gridDim.x * (SB / 64)) {
        const size_t ray = ray_index[q];
        const uint32_t nb = num_visited[ray];
        const float2 *row = reinterpret_cast<const float2 *>(hit_dist + ray * (size_t)M * 2);
        // (a ray that misses the mesh has no row -- with compact rows not even a written one: the sync-free training path
        // names such rays only when a whole batch misses; their samples are discarded, they just have to be finite)
        const float near = nb ? row[0].x : 0.0f;
        const float far = nb ? row[nb - 1].y : 1.0f;
        if (lane == 0) { near_far[2 * q] = near; near_far[2 * q + 1] = far; }
        if (biased) {
            // lengths (clamped at 0: the cell -1 closing segments) and their running sum from the first entry point
            float carry = near;   // bounds_start = hit_distances[..., 0, 0]
            if (lane == 0) cum[0] = carry;
            for (uint32_t base = 0; base < nb; base += 64) {
                const uint32_t k = base + lane;
                float len = 0.f;
                if (k < nb) { const float2 s = row[k]; len = fmaxf(s.y - s.x, 0.f); }
                const float inc = wave_incl_scan(len, lane);
                if (k < nb) cum[k + 1] = carry + inc;
                carry += __shfl(inc, 63);
            }
            lds_sync();
        }
        const float fnb = (float)nb;
        for (uint32_t j = lane; j <= S; j += 64) {
            float b = lin[j];
            if (t_rand) {   // stratified: every edge jittered between the centres of its two neighbouring bins
                const float lower = j == 0 ? lin[0] : (lin[j] + lin[j - 1]) / 2.0f;
                const float upper = j == S ? lin[S] : (lin[j + 1] + lin[j]) / 2.0f;
                b = lower + (upper - lower) * t_rand[q * (size_t)(S + 1) + j];
            }
            float e = b * far + (1.0f - b) * near;
            if (biased && nb) {
                float rest = (e - near) / (far - near) * fnb;
                float iv = floorf(rest);
                iv = fminf(iv, fnb - 1.0f);
                iv = fmaxf(iv, 0.0f);
                rest = rest - iv;
                const uint32_t i = (uint32_t)iv;
                const float2 s = row[i];
                e = cum[i] + fmaxf(s.y - s.x, 0.f) * rest;
            }
            edges[q * (size_t)(S + 1) + j] = e;
        }
        if (biased) lds_sync();
    }
}

// edges [r, S+1] euclidean coarse edges, weights [r, S] coarse weights, near_far [r, 2];
// u_table [nb] (nb = num_fine + 1): evaluation = the bin-centred quantiles, training = the bin starts, to which
// u_rand [r, nb] / nb is added; out [r, S + nb + 1] = the merged, sorted edges mapped back to euclidean distances
__global__ __launch_bounds__(SB) void k_sample_pdf(size_t r, uint32_t S, uint32_t nb, const float *__restrict__ edges,
                                                   const float *__restrict__ weights, const float *__restrict__ near_far,
                                                   const float *__restrict__ u_table, const float *__restrict__ u_rand,
                                                   float histogram_padding, float eps, float *__restrict__ out) {
    extern __shared__ float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t per = (S + 1) + (S + 1) + nb;
    float *cdf = smem + (size_t)wave * per;      // [S+1]
    float *sp = cdf + (S + 1);                   // [S+1] spacing edges
    float *nw = sp + (S + 1);                    // [nb]  new bins
    for (size_t q = (size_t)blockIdx.x * (SB / 64) + wave; q < r; q += (size_t)gridDim.x * (SB / 64)) {
        const float near = near_far[2 * q], far = near_far[2 * q + 1];
        const float *w = weights + q * (size_t)S;
        const float *e = edges + q * (size_t)(S + 1);
        for (uint32_t j = lane; j <= S; j += 64) sp[j] = (e[j] - near) / (far - near);
        // padded weights -> pdf -> cdf
        float part = 0.f;
        for (uint32_t j = lane; j < S; j += 64) part += w[j] + histogram_padding;
        float wsum = wave_sum(part);
        const float padding = fmaxf(eps - wsum, 0.f);
        const float add = padding / (float)S;
        wsum = wsum + padding;
        float carry = 0.f;
        if (lane == 0) cdf[0] = 0.f;
        for (uint32_t base = 0; base < S; base += 64) {
            const uint32_t j = base + lane;
            const float pdf = j < S ? ((w[j] + histogram_padding) + add) / wsum : 0.f;
            const float inc = wave_incl_scan(pdf, lane);
            if (j < S) cdf[j + 1] = fminf(1.0f, carry + inc);
            carry += __shfl(inc, 63);
        }
        lds_sync();
        // inverse CDF at the quantiles
        for (uint32_t k = lane; k < nb; k += 64) {
            float u = u_table[k];
            if (u_rand) u = u + u_rand[q * (size_t)nb + k] / (float)nb;
            // searchsorted(cdf, u, side = "right"): number of entries <= u
            uint32_t lo = 0, hi = S + 1;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
            }
            const uint32_t below = lo == 0 ? 0u : (lo - 1 > S ? S : lo - 1), above = lo > S ? S : lo;
            const float c0 = cdf[below], c1 = cdf[above], b0 = sp[below], b1 = sp[above];
            float t = (u - c0) / (c1 - c0);
            if (!(t == t)) t = 0.f;                                   // nan_to_num(., 0)
            if (t == INFINITY) t = 3.4028234663852886e38f;            // nan_to_num maps +-inf to the finite extremes
            if (t == -INFINITY) t = -3.4028234663852886e38f;
            t = fminf(fmaxf(t, 0.f), 1.f);
            nw[k] = b0 + t * (b1 - b0);
        }
        lds_sync();
        // both lists are sorted up to rounding (the biased mapping and the inverse CDF are monotone functions evaluated
        // in fp32): enforce it, then merge by rank (coarse edges first on ties) and map back to euclidean distances
        lds_running_max(sp, S + 1, lane);
        lds_running_max(nw, nb, lane);
        lds_sync();
        float *o = out + q * (size_t)(S + 1 + nb);
        for (uint32_t j = lane; j <= S; j += 64) {
            const float v = sp[j];
            uint32_t lo = 0, hi = nb;                                 // new bins strictly below v
            while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (nw[mid] < v) lo = mid + 1; else hi = mid; }
            o[j + lo] = v * far + (1.0f - v) * near;
        }
        for (uint32_t k = lane; k < nb; k += 64) {
            const float v = nw[k];
            uint32_t lo = 0, hi = S + 1;                              // coarse edges <= v
            while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (sp[mid] <= v) lo = mid + 1; else hi = mid; }
            o[k + lo] = v * far + (1.0f - v) * near;
        }
        lds_sync();
    }
}

void launch_sample_coarse(size_t r, uint32_t S, uint32_t M, const uint32_t *ray_index, const uint32_t *num_visited, const float *hit_dist,
                          const float *lin, const float *t_rand, bool biased, float *edges, float *near_far, hipStream_t stream) {
    if (r == 0) return;
    const size_t smem = biased ? (size_t)(SB / 64) * (M + 1) * sizeof(float) : 0;
    if (smem > 64 * 1024) throw Error("sample_coarse: max_ray_triangles too large for the biased sampler");
    const size_t blocks = (r + SB / 64 - 1) / (SB / 64);
    hipLaunchKernelGGL(k_sample_coarse, dim3((unsigned)(blocks < 256 * 16 ? blocks : 256 * 16)), dim3(SB), smem, stream, r, S, M, ray_index,
                       num_visited, hit_dist, lin, t_rand, biased ? 1 : 0, edges, near_far);
}

void launch_sample_pdf(size_t r, uint32_t S, uint32_t num_fine, const float *edges, const float *weights, const float *near_far,
                       const float *u_table, const float *u_rand, float histogram_padding, float eps, float *out, hipStream_t stream) {
    if (r == 0) return;
    const uint32_t nb = num_fine + 1;
    const size_t smem = (size_t)(SB / 64) * (2 * (S + 1) + nb) * sizeof(float);
    if (smem > 64 * 1024) throw Error("sample_pdf: too many samples per ray");
    const size_t blocks = (r + SB / 64 - 1) / (SB / 64);
    hipLaunchKernelGGL(k_sample_pdf, dim3((unsigned)(blocks < 256 * 16 ? blocks : 256 * 16)), dim3(SB), smem, stream, r, S, nb, edges, weights,
                       near_far, u_table, u_rand, histogram_padding, eps, out);
}

}  // namespace tn
