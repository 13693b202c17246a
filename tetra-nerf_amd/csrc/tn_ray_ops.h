// tn_ray_ops.h -- the per-RAY stages between trace_rays and the frame (coarse sampler, PDF sampler, composite) as wave-level
// device functions: ONE wavefront works on one ray.  The stand-alone kernels (k_sample_coarse / k_sample_pdf in
// tn_samplers.hip, k_composite in tn_mlp.hip) and the persistent render kernel (tn_render_rays.hip) call the SAME functions,
// so a ray's values do not depend on which launch shape produced them.
// Reference: tetranerf/nerfstudio/model.py:111-192, 531-557, 582-586, 632-662 (+ nerfstudio's UniformSampler / PDFSampler /
// RaySamples.get_weights / renderers, restated in tetra-nerf_amd/render.py -- those PyTorch statements are the parity definition).
#pragma once
#include "tn_device.h"
#include "tn_kernels.h"

namespace tn {
namespace rayops {

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
// CH independent inclusive wave scans (sum / running maximum), step by step TOGETHER: a scan is a chain of six dependent
// cross-lane moves (ds_bpermute: an LDS-crossbar round trip each), and a wave that owns its ray alone -- the persistent render
// kernel runs 8 waves per CU -- has nothing to fill the gaps with but the other chunks' chains.  Per chunk the operations and
// their order are exactly those of wave_incl_scan: same bits.
template <int CH>
__device__ __forceinline__ void wave_incl_scan_multi(float (&v)[CH], int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        float o[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) o[c] = __shfl_up(v[c], off);
#pragma unroll
        for (int c = 0; c < CH; ++c)
            if (lane >= off) v[c] += o[c];
    }
}
template <int CH>
__device__ __forceinline__ void wave_incl_max_multi(float (&v)[CH], int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        float o[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) o[c] = __shfl_up(v[c], off);
#pragma unroll
        for (int c = 0; c < CH; ++c)
            if (lane >= off) v[c] = fmaxf(v[c], o[c]);
    }
}
// running maximum over an LDS array of n <= 64 CH floats, the chunks' scans interleaved (same values as lds_running_max)
template <int CH>
__device__ __forceinline__ void lds_running_max_multi(float *a, uint32_t n, int lane) {
    float v[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) { const uint32_t k = 64u * c + lane; v[c] = k < n ? a[k] : -INFINITY; }
    wave_incl_max_multi<CH>(v, lane);
    float carry = -INFINITY;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const uint32_t k = 64u * c + lane;
        const float m = fmaxf(v[c], carry);
        if (k < n) a[k] = m;
        carry = __shfl(m, 63);
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

// Coarse sampler of one hitting ray (model.py:531-557, 111-122, 166-177): near / far + the S + 1 bin edges.
// `ray` = its row in the trace outputs; lin [S+1] = linspace(0, 1, S+1); t_row [S+1] uniform draws of THIS ray or null
// (evaluation); cum: M + 1 floats of LDS owned by the wave (biased only); edges [S+1], near_far [2] of THIS ray.
// ray_sample_coarse_nf: nb = num_visited[ray] and the ray's near / far are the CALLER's (the persistent kernel loads them for a
// whole tile at once); edges2 (nullable): a second copy of the edges, e.g. in the wave's LDS for the matcher that follows.
__device__ __forceinline__ void ray_sample_coarse_nf(uint32_t S, uint32_t M, size_t ray, uint32_t nb, float near, float far,
                                                     const float *__restrict__ hit_dist, const float *__restrict__ lin,
                                                     const float *__restrict__ t_row, int biased, float *__restrict__ edges,
                                                     float *edges2, float *cum, int lane) {
    const float2 *row = reinterpret_cast<const float2 *>(hit_dist + ray * (size_t)M * 2);
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
        if (t_row) {   // stratified: every edge jittered between the centres of its two neighbouring bins
            const float lower = j == 0 ? lin[0] : (lin[j] + lin[j - 1]) / 2.0f;
            const float upper = j == S ? lin[S] : (lin[j + 1] + lin[j]) / 2.0f;
            b = lower + (upper - lower) * t_row[j];
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
        edges[j] = e;
        if (edges2) edges2[j] = e;
    }
    if (biased || edges2) lds_sync();
}

// the same with near / far read from the ray's row (the stand-alone kernel); near_far [2] of THIS ray
__device__ __forceinline__ void ray_sample_coarse(uint32_t S, uint32_t M, size_t ray, uint32_t nb,
                                                  const float *__restrict__ hit_dist, const float *__restrict__ lin,
                                                  const float *__restrict__ t_row, int biased, float *__restrict__ edges,
                                                  float *__restrict__ near_far, float *cum, int lane) {
    const float2 *row = reinterpret_cast<const float2 *>(hit_dist + ray * (size_t)M * 2);
    // (a ray that misses the mesh has no row -- with compact rows not even a written one: the sync-free training path
    // names such rays only when a whole batch misses; their samples are discarded, they just have to be finite)
    const float near = nb ? row[0].x : 0.0f;
    const float far = nb ? row[nb - 1].y : 1.0f;
    if (lane == 0) { near_far[0] = near; near_far[1] = far; }
    ray_sample_coarse_nf(S, M, ray, nb, near, far, hit_dist, lin, t_row, biased, edges, nullptr, cum, lane);
}

// The binary searches of searchsorted / the merge by rank, for the CH items a lane owns AT ONCE: the classic lo / hi / mid loop
// with a fixed trip count (K >= the longest search; a finished item idles), so every item visits exactly the mids the plain
// loop visits -- same result on any array, sorted or not -- while the LDS reads of the CH items overlap instead of forming one
// chain of K dependent round trips per item (a wave that owns its ray alone has nothing else to hide them behind).
// LE: count of entries <= v (searchsorted side = "right"); otherwise: entries < v.
template <int CH, bool LE>
__device__ __forceinline__ void multi_search(const float *arr, uint32_t N, const float (&v)[CH], const bool (&on)[CH], uint32_t (&res)[CH]) {
    uint32_t lo[CH], hi[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) { lo[c] = 0; hi[c] = N; }
    const int K = N ? 32 - __clz((int)N) : 0;     // a range of N halves to nothing in at most floor(log2 N) + 1 steps
    const uint32_t last = N ? N - 1 : 0;
    for (int it = 0; it < K; ++it) {
        // straight-line (selects, no exec-masked branches): the CH reads of a step are issued together
        uint32_t mid[CH];
        float a[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) { mid[c] = (lo[c] + hi[c]) >> 1; a[c] = arr[mid[c] < last ? mid[c] : last]; }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const bool act = on[c] && lo[c] < hi[c];
            const bool take = LE ? (a[c] <= v[c]) : (a[c] < v[c]);
            lo[c] = (act && take) ? mid[c] + 1 : lo[c];
            hi[c] = (act && !take) ? mid[c] : hi[c];
        }
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) res[c] = lo[c];
}

// floats of LDS ray_sample_pdf needs per wave
__host__ __device__ constexpr size_t pdf_lds_floats(uint32_t S, uint32_t nb) { return 2 * (size_t)(S + 1) + nb; }

// nerfstudio's PDFSampler (include_original) for one ray (model.py:582-586): e [S+1] euclidean coarse edges, w [S] coarse
// weights -> out [S + 1 + nb] merged, sorted euclidean edges (nb = num_fine + 1).  u_table [nb]: evaluation = the bin-centred
// quantiles, training = the bin starts, to which u_row [nb] / nb is added (u_row null otherwise).  lds: pdf_lds_floats(S, nb).
// ray_sample_pdf_chunks<CH>: S + 1 <= 64 CH and nb <= 64 CH -- every load of the ray requested before its first use, the
// chunks' scans and the lane's binary searches interleaved (see wave_incl_scan_multi / multi_search); ray_sample_pdf_loops:
// any size, plain loops.  Same expressions in the same order: same bits.
template <int CH>
__device__ __forceinline__ void ray_sample_pdf_chunks(uint32_t S, uint32_t nb, const float *__restrict__ e, const float *__restrict__ w,
                                                      float near, float far, const float *__restrict__ u_table,
                                                      const float *__restrict__ u_row, float histogram_padding, float eps,
                                                      float *__restrict__ o, float *lds, int lane, float *o2 = nullptr) {
    float *cdf = lds;                // [S+1]
    float *sp = cdf + (S + 1);       // [S+1] spacing edges
    float *nw = sp + (S + 1);        // [nb]  new bins
    float ev[CH], wv[CH], uv[CH];
    bool on[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const uint32_t j = 64u * c + lane;
        ev[c] = j <= S ? e[j] : 0.f;
        wv[c] = j < S ? w[j] : 0.f;
        on[c] = j < nb;
        uv[c] = 0.f;
        if (on[c]) {
            float u = u_table[j];
            if (u_row) u = u + u_row[j] / (float)nb;
            uv[c] = u;
        }
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const uint32_t j = 64u * c + lane;
        if (j <= S) sp[j] = (ev[c] - near) / (far - near);
    }
    // padded weights -> pdf -> cdf
    float part = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const uint32_t j = 64u * c + lane;
        if (j < S) part += wv[c] + histogram_padding;
    }
    float wsum = wave_sum(part);
    const float padding = fmaxf(eps - wsum, 0.f);
    const float add = padding / (float)S;
    wsum = wsum + padding;
    float inc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const uint32_t j = 64u * c + lane;
        inc[c] = j < S ? ((wv[c] + histogram_padding) + add) / wsum : 0.f;
    }
    wave_incl_scan_multi<CH>(inc, lane);
    float carry = 0.f;
    if (lane == 0) cdf[0] = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const uint32_t j = 64u * c + lane;
        if (j < S) cdf[j + 1] = fminf(1.0f, carry + inc[c]);
        carry += __shfl(inc[c], 63);
    }
    lds_sync();
    // inverse CDF at the quantiles
    uint32_t pos[CH];
    multi_search<CH, true>(cdf, S + 1, uv, on, pos);         // searchsorted(cdf, u, side = "right"): number of entries <= u
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const uint32_t k = 64u * c + lane, lo = pos[c];
        const float u = uv[c];
        const uint32_t below = lo == 0 ? 0u : (lo - 1 > S ? S : lo - 1), above = lo > S ? S : lo;
        const float c0 = cdf[below], c1 = cdf[above], b0 = sp[below], b1 = sp[above];
        float t = (u - c0) / (c1 - c0);
        if (!(t == t)) t = 0.f;                                   // nan_to_num(., 0)
        if (t == INFINITY) t = 3.4028234663852886e38f;            // nan_to_num maps +-inf to the finite extremes
        if (t == -INFINITY) t = -3.4028234663852886e38f;
        t = fminf(fmaxf(t, 0.f), 1.f);
        if (on[c]) nw[k] = b0 + t * (b1 - b0);
    }
    lds_sync();
    // both lists are sorted up to rounding (the biased mapping and the inverse CDF are monotone functions evaluated
    // in fp32): enforce it, then merge by rank (coarse edges first on ties) and map back to euclidean distances
    lds_running_max_multi<CH>(sp, S + 1, lane);
    lds_running_max_multi<CH>(nw, nb, lane);
    lds_sync();
    float vv[CH];
    bool on2[CH];
    uint32_t rk[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) { const uint32_t j = 64u * c + lane; on2[c] = j <= S; vv[c] = on2[c] ? sp[j] : 0.f; }
    multi_search<CH, false>(nw, nb, vv, on2, rk);            // new bins strictly below v
#pragma unroll
    for (int c = 0; c < CH; ++c)
        if (on2[c]) {
            const float ev2 = vv[c] * far + (1.0f - vv[c]) * near;
            o[64u * c + lane + rk[c]] = ev2;
            if (o2) o2[64u * c + lane + rk[c]] = ev2;
        }
#pragma unroll
    for (int c = 0; c < CH; ++c) { const uint32_t k = 64u * c + lane; on2[c] = k < nb; vv[c] = on2[c] ? nw[k] : 0.f; }
    multi_search<CH, true>(sp, S + 1, vv, on2, rk);          // coarse edges <= v
#pragma unroll
    for (int c = 0; c < CH; ++c)
        if (on2[c]) {
            const float ev2 = vv[c] * far + (1.0f - vv[c]) * near;
            o[64u * c + lane + rk[c]] = ev2;
            if (o2) o2[64u * c + lane + rk[c]] = ev2;
        }
    lds_sync();
}

__device__ __forceinline__ void ray_sample_pdf_loops(uint32_t S, uint32_t nb, const float *__restrict__ e, const float *__restrict__ w,
                                                     float near, float far, const float *__restrict__ u_table,
                                                     const float *__restrict__ u_row, float histogram_padding, float eps,
                                                     float *__restrict__ o, float *lds, int lane) {
    float *cdf = lds;                // [S+1]
    float *sp = cdf + (S + 1);       // [S+1] spacing edges
    float *nw = sp + (S + 1);        // [nb]  new bins
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
        if (u_row) u = u + u_row[k] / (float)nb;
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
    lds_running_max(sp, S + 1, lane);
    lds_running_max(nw, nb, lane);
    lds_sync();
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

__device__ __forceinline__ void ray_sample_pdf(uint32_t S, uint32_t nb, const float *__restrict__ e, const float *__restrict__ w,
                                               float near, float far, const float *__restrict__ u_table,
                                               const float *__restrict__ u_row, float histogram_padding, float eps,
                                               float *__restrict__ o, float *lds, int lane, float *o2 = nullptr) {
    // o2 (nullable): a second copy of the merged edges (the wave's LDS, for the matcher that follows); returns through *o2
    // only in the chunk forms -- the caller falls back to `o` otherwise (pdf_writes_second_copy)
    const uint32_t big = S + 1 > nb ? S + 1 : nb;       // wave-uniform dispatch on the chunks a lane owns
    if (big <= 64 * 3) ray_sample_pdf_chunks<3>(S, nb, e, w, near, far, u_table, u_row, histogram_padding, eps, o, lds, lane, o2);
    else if (big <= 64 * 5) ray_sample_pdf_chunks<5>(S, nb, e, w, near, far, u_table, u_row, histogram_padding, eps, o, lds, lane, o2);
    else ray_sample_pdf_loops(S, nb, e, w, near, far, u_table, u_row, histogram_padding, eps, o, lds, lane);
}
__host__ __device__ constexpr bool pdf_writes_second_copy(uint32_t S, uint32_t nb) { return (S + 1 > nb ? S + 1 : nb) <= 64 * 5; }

// RaySamples.get_weights + RGB (background blend) / accumulation / median-depth renderers of one ray (model.py:632-662):
// sigma [S], rgb [S,3] (null: weights only), e [S+1]; out_rgb3 / out_acc / out_depth: where THIS ray's results go (null: not
// written); out_w [S] (nullable; global or LDS).  Lanes stride the samples; exclusive scan of sigma * delta.
// The chunks of 64 samples form a serial chain only through their CARRIES (two scalars per chunk); their loads and their two
// wave scans are independent.  A wave that owns a ray alone (8 waves per CU in the persistent render kernel) would otherwise
// walk 2 x 6 dependent cross-lane moves per chunk with nothing to overlap them with: groups of CH chunks are processed
// together -- all loads first, the CH scans of sigma * delta interleaved, the carries in order, the CH scans of the weights
// interleaved, the median search in order.  Per sample the expressions and their order are unchanged: same bits for any CH.
template <int CH>
__device__ __forceinline__ void ray_composite_chunks(uint32_t S, const float *__restrict__ sigma, const float *__restrict__ rgb,
                                                     const float *__restrict__ e, const Background &background,
                                                     float *__restrict__ out_rgb3, float *__restrict__ out_acc,
                                                     float *__restrict__ out_depth, float *out_w, int lane) {
    float carry = 0.f;       // sum of sigma*delta of all previous samples
    float cw = 0.f;          // running sum of weights (for the median depth)
    float r0 = 0.f, r1 = 0.f, r2 = 0.f, accw = 0.f;
    float depth = 0.f;
    bool found = false;
    for (uint32_t base0 = 0; base0 < S; base0 += 64 * CH) {
        float stv[CH], env[CH], dd[CH], inc[CH], k0[CH], k1[CH], k2[CH];
        bool ok[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const uint32_t j = base0 + 64u * c + lane;
            ok[c] = j < S;
            const size_t q = ok[c] ? j : S - 1;
            stv[c] = e[q]; env[c] = e[q + 1];
            const float sg = sigma[q];
            k0[c] = k1[c] = k2[c] = 0.f;
            if (rgb && ok[c]) { k0[c] = rgb[3 * q]; k1[c] = rgb[3 * q + 1]; k2[c] = rgb[3 * q + 2]; }
            dd[c] = ok[c] ? (env[c] - stv[c]) * sg : 0.f;
            inc[c] = dd[c];
        }
        wave_incl_scan_multi<CH>(inc, lane);      // inclusive scans of dd over the wave, one per chunk
        float w[CH], winc[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const float excl = carry + (inc[c] - dd[c]);
            float wi = (1.0f - expf(-dd[c])) * expf(-excl);
            if (!(wi == wi) || !ok[c]) wi = 0.f;  // nan_to_num
            w[c] = wi;
            winc[c] = wi;
            carry += __shfl(inc[c], 63);
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const uint32_t j = base0 + 64u * c + lane;
            if (out_w && ok[c]) out_w[j] = w[c];
            if (rgb) {
                float c0 = k0[c], c1 = k1[c], c2 = k2[c];
                if (background.clamp) { c0 = nan_to_num(c0); c1 = nan_to_num(c1); c2 = nan_to_num(c2); }
                r0 += w[c] * c0; r1 += w[c] * c1; r2 += w[c] * c2;
            }
            accw += w[c];
        }
        wave_incl_scan_multi<CH>(winc, lane);     // median depth: cumulative weights
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const float cum = cw + winc[c];
            const uint64_t m = __ballot(ok[c] && cum >= 0.5f);
            if (!found && m) {                    // first sample whose cumulative weight reaches 0.5
                const int src = __ffsll((unsigned long long)m) - 1;
                depth = __shfl(0.5f * (stv[c] + env[c]), src);
                found = true;
            }
            cw += __shfl(winc[c], 63);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        r0 += __shfl_xor(r0, off); r1 += __shfl_xor(r1, off); r2 += __shfl_xor(r2, off); accw += __shfl_xor(accw, off);
    }
    if (!found) depth = 0.5f * (e[S - 1] + e[S]);  // searchsorted clamps to the last sample
    if (lane == 0 && out_rgb3) {
        float o0 = r0 + background.r * (1.0f - accw), o1 = r1 + background.g * (1.0f - accw), o2 = r2 + background.b * (1.0f - accw);
        if (background.clamp) { o0 = fminf(fmaxf(o0, 0.f), 1.f); o1 = fminf(fmaxf(o1, 0.f), 1.f); o2 = fminf(fmaxf(o2, 0.f), 1.f); }
        out_rgb3[0] = o0; out_rgb3[1] = o1; out_rgb3[2] = o2;
        out_acc[0] = accw;
        out_depth[0] = depth;
    }
}

__device__ __forceinline__ void ray_composite(uint32_t S, const float *__restrict__ sigma, const float *__restrict__ rgb,
                                              const float *__restrict__ e, const Background &background, float *__restrict__ out_rgb3,
                                              float *__restrict__ out_acc, float *__restrict__ out_depth, float *out_w, int lane) {
    // wave-uniform dispatch on the ray's chunk count (chunks beyond S cost scans of zeros: the groups are sized to the
    // shipped sample counts -- 128 / 256 / 257 / 513 -- with little waste)
    if (S <= 64 * 2) ray_composite_chunks<2>(S, sigma, rgb, e, background, out_rgb3, out_acc, out_depth, out_w, lane);
    else if (S <= 64 * 5) ray_composite_chunks<5>(S, sigma, rgb, e, background, out_rgb3, out_acc, out_depth, out_w, lane);
    else ray_composite_chunks<9>(S, sigma, rgb, e, background, out_rgb3, out_acc, out_depth, out_w, lane);
}

}  // namespace rayops
}  // namespace tn
