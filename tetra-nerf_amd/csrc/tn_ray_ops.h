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
__device__ __forceinline__ void ray_sample_coarse(uint32_t S, uint32_t M, size_t ray, const uint32_t *__restrict__ num_visited,
                                                  const float *__restrict__ hit_dist, const float *__restrict__ lin,
                                                  const float *__restrict__ t_row, int biased, float *__restrict__ edges,
                                                  float *__restrict__ near_far, float *cum, int lane) {
    const uint32_t nb = num_visited[ray];
    const float2 *row = reinterpret_cast<const float2 *>(hit_dist + ray * (size_t)M * 2);
    // (a ray that misses the mesh has no row -- with compact rows not even a written one: the sync-free training path
    // names such rays only when a whole batch misses; their samples are discarded, they just have to be finite)
    const float near = nb ? row[0].x : 0.0f;
    const float far = nb ? row[nb - 1].y : 1.0f;
    if (lane == 0) { near_far[0] = near; near_far[1] = far; }
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
    }
    if (biased) lds_sync();
}

// floats of LDS ray_sample_pdf needs per wave
__host__ __device__ constexpr size_t pdf_lds_floats(uint32_t S, uint32_t nb) { return 2 * (size_t)(S + 1) + nb; }

// nerfstudio's PDFSampler (include_original) for one ray (model.py:582-586): e [S+1] euclidean coarse edges, w [S] coarse
// weights -> out [S + 1 + nb] merged, sorted euclidean edges (nb = num_fine + 1).  u_table [nb]: evaluation = the bin-centred
// quantiles, training = the bin starts, to which u_row [nb] / nb is added (u_row null otherwise).  lds: pdf_lds_floats(S, nb).
__device__ __forceinline__ void ray_sample_pdf(uint32_t S, uint32_t nb, const float *__restrict__ e, const float *__restrict__ w,
                                               float near, float far, const float *__restrict__ u_table,
                                               const float *__restrict__ u_row, float histogram_padding, float eps,
                                               float *__restrict__ o, float *lds, int lane) {
    float *cdf = lds;                // [S+1]
    float *sp = cdf + (S + 1);       // [S+1] spacing edges
    float *nw = sp + (S + 1);        // [nb]  new bins
    // the ray's edges and weights: every load requested before the first use (one round trip instead of one per chunk and
    // pass); WCH chunks of 64 cover S <= 576, longer rays take the generic loops
    constexpr int WCH = 9;
    if (S <= 64 * WCH) {
        float ev[WCH], wv[WCH];
#pragma unroll
        for (int c = 0; c < WCH; ++c) {
            ev[c] = wv[c] = 0.f;
            if (64u * c <= S) {                 // wave-uniform
                const uint32_t j = 64u * c + lane;
                if (j <= S) ev[c] = e[j];
                if (j < S) wv[c] = w[j];
            }
        }
#pragma unroll
        for (int c = 0; c < WCH; ++c) {
            const uint32_t j = 64u * c + lane;
            if (64u * c <= S && j <= S) sp[j] = (ev[c] - near) / (far - near);
        }
        // padded weights -> pdf -> cdf
        float part = 0.f;
#pragma unroll
        for (int c = 0; c < WCH; ++c) {
            const uint32_t j = 64u * c + lane;
            if (64u * c < S && j < S) part += wv[c] + histogram_padding;
        }
        float wsum = wave_sum(part);
        const float padding = fmaxf(eps - wsum, 0.f);
        const float add = padding / (float)S;
        wsum = wsum + padding;
        float carry = 0.f;
        if (lane == 0) cdf[0] = 0.f;
#pragma unroll
        for (int c = 0; c < WCH; ++c) {
            if (64u * c >= S) break;            // wave-uniform
            const uint32_t j = 64u * c + lane;
            const float pdf = j < S ? ((wv[c] + histogram_padding) + add) / wsum : 0.f;
            const float inc = wave_incl_scan(pdf, lane);
            if (j < S) cdf[j + 1] = fminf(1.0f, carry + inc);
            carry += __shfl(inc, 63);
        }
    } else {
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
    // both lists are sorted up to rounding (the biased mapping and the inverse CDF are monotone functions evaluated
    // in fp32): enforce it, then merge by rank (coarse edges first on ties) and map back to euclidean distances
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

// RaySamples.get_weights + RGB (background blend) / accumulation / median-depth renderers of one ray (model.py:632-662):
// sigma [S], rgb [S,3] (null: weights only), e [S+1]; out_rgb3 / out_acc / out_depth: where THIS ray's results go (null: not
// written); out_w [S] (nullable).  Lanes stride the samples; exclusive scan of sigma * delta.
__device__ __forceinline__ void ray_composite(uint32_t S, const float *__restrict__ sigma, const float *__restrict__ rgb,
                                              const float *__restrict__ e, const Background &background, float *__restrict__ out_rgb3,
                                              float *__restrict__ out_acc, float *__restrict__ out_depth, float *__restrict__ out_w,
                                              int lane) {
    float carry = 0.f;       // sum of sigma*delta of all previous samples
    float cw = 0.f;          // running sum of weights (for the median depth)
    float r0 = 0.f, r1 = 0.f, r2 = 0.f, accw = 0.f;
    float depth = 0.f;
    bool found = false;
    // The chunks of 64 samples are a serial chain (two wave scans each, carried sums), their LOADS are not: a wave that owns
    // a ray alone (8 waves per CU in the persistent render kernel) would pay one memory round trip per chunk, so the values of
    // up to CH chunks (576 samples: both shipped configurations in one go) are requested before the first scan.
    constexpr int CH = 9;
    for (uint32_t base0 = 0; base0 < S; base0 += 64 * CH) {
        float stv[CH], env[CH], sgv[CH], k0[CH], k1[CH], k2[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            stv[c] = env[c] = sgv[c] = k0[c] = k1[c] = k2[c] = 0.f;
            if (base0 + 64u * c < S) {          // wave-uniform
                const uint32_t j = base0 + 64u * c + lane;
                const size_t q = j < S ? j : S - 1;
                stv[c] = e[q]; env[c] = e[q + 1]; sgv[c] = sigma[q];
                if (rgb) { k0[c] = rgb[3 * q]; k1[c] = rgb[3 * q + 1]; k2[c] = rgb[3 * q + 2]; }
            }
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (base0 + 64u * c >= S) break;    // wave-uniform
            const uint32_t j = base0 + 64u * c + lane;
            const bool ok = j < S;
            const size_t q = ok ? j : S - 1;
            const float st = stv[c], en = env[c];
            const float dd = ok ? (en - st) * sgv[c] : 0.f;
            // inclusive scan of dd over the wave
            float inc = dd;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const float o = __shfl_up(inc, off);
                if (lane >= off) inc += o;
            }
            const float excl = carry + (inc - dd);
            float w = (1.0f - expf(-dd)) * expf(-excl);
            if (!(w == w) || !ok) w = 0.f;  // nan_to_num
            if (out_w && ok) out_w[q] = w;
            if (rgb) {
                float c0 = k0[c], c1 = k1[c], c2 = k2[c];
                if (background.clamp) { c0 = nan_to_num(c0); c1 = nan_to_num(c1); c2 = nan_to_num(c2); }
                r0 += w * c0; r1 += w * c1; r2 += w * c2;
            }
            accw += w;
            // median depth: first sample whose cumulative weight reaches 0.5
            float winc = w;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const float o = __shfl_up(winc, off);
                if (lane >= off) winc += o;
            }
            const float cum = cw + winc;
            const uint64_t m = __ballot(ok && cum >= 0.5f);
            if (!found && m) {
                const int src = __ffsll((unsigned long long)m) - 1;
                depth = __shfl(0.5f * (st + en), src);
                found = true;
            }
            carry += __shfl(inc, 63);
            cw += __shfl(winc, 63);
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

}  // namespace rayops
}  // namespace tn
