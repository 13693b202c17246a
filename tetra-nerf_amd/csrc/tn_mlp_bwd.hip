// tn_mlp_bwd.hip -- training adjoint of the shallow MLP + heads on fp32 MFMA: the dX chain.
//
// Replaces what PyTorch autograd does for the reference's model in training
// (tetranerf/nerfstudio/model.py:602-630: mlp_base, density / rgb heads, mlp_head; the trainer back-propagates
// through them with cuBLAS GEMMs and elementwise kernels).
//
//   k_mlp_backward.  Same dataflow as the forward kernel (tn_mlp.hip): one wavefront owns 32 samples, the gradient
//       of a layer's pre-activations lives in the MFMA accumulators in the layout the next GEMM's B operand wants: the
//       transposed weights W^T are packed with their K axis (the layer's OUTPUT features) in accumulator order, so d_pre
//       of layer l+1 is fed straight back as the B operand of W_{l+1}^T exactly like an activation in the forward pass.
//       It recomputes NOTHING: the training forward (tn_mlp.hip, TRAIN) saved the ReLU masks of the four hidden layers
//       (64 bits per lane and layer), and softplus' / sigmoid' follow from the forward's OUTPUTS
//       (softplus'(x) = 1 - exp(-softplus(x)), sigmoid' = y (1 - y)).  Round 3a recomputed the forward pass inside this
//       kernel (1864 MFMAs per 32 samples; now 896).
//       Outputs: the pre-activation gradients d_pre1..4 QUAD-major ([F/4][n][4], tn_mlp_common.h: a lane's four consecutive
//       features = one 16-byte store, 512 contiguous bytes per half-wave) and d_sigma_raw / d_rgb_raw as four plain rows
//       [4, n], for the weight-gradient GEMMs (tn_mlp_grad.hip); and d_x0 [n, 64] = the gradient of the gathered features
//       as sample-major rows, which tn_interpolate_values_backward_vm scatters into the field.
//
// FLOPs per fine sample: forward 122.6 k, dX 114.7 k, dW 122.4 k.
#include "tn_mlp_common.h"

namespace tn {

using namespace mlp;

namespace {

// ---- transposed packs: [k-step over the layer's OUTPUT features (accumulator order)][tile of INPUT features][lane]
constexpr int OTI1 = FD / 32;                                   // input tiles of layer 1
constexpr size_t tfloats(int tiles) { return (size_t)KSH * tiles * 64; }
constexpr size_t OFFT_H = 0;                                    // Wh[:, 27:]^T  (128 -> 128); behind it the density vector
constexpr size_t N_TH = tfloats(OT) + DVEC + CVEC;              //   (accumulator order) and the rgb head vectors (K order)
constexpr size_t OFFT_3 = OFFT_H + N_TH;
constexpr size_t OFFT_2 = OFFT_3 + tfloats(OT);
constexpr size_t OFFT_1 = OFFT_2 + tfloats(OT);
constexpr size_t PACKT_FLOATS = OFFT_1 + tfloats(OTI1);

__global__ void k_mlp_pack_t(MlpWeights w, float *__restrict__ pt) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= PACKT_FLOATS) return;
    float v = 0.f;
    auto at = [](size_t j, int tiles, int &ks, int &t, int &row, int &h) {
        const int lane = (int)(j & 63);
        row = lane & 31; h = lane >> 5; t = (int)((j >> 6) % tiles); ks = (int)(j / (64 * (size_t)tiles));
    };
    int ks, t, row, h;
    if (i < OFFT_3) {
        const size_t j = i - OFFT_H;
        if (j < tfloats(OT)) { at(j, OT, ks, t, row, h); v = w.wh[(size_t)acc_k(ks, h) * (ENC + HID) + ENC + 32 * t + row]; }
        else if (j < tfloats(OT) + DVEC) { const int jj = (int)(j - tfloats(OT)); if (jj < 128) v = w.wd[acc_k(jj & 63, jj >> 6)]; }
        else {
            const int jj = (int)(j - tfloats(OT) - DVEC);
            if (jj < 384) v = w.wr[(size_t)(jj >> 7) * HID + acc_k(jj & 63, (jj >> 6) & 1)];
        }
    } else if (i < OFFT_2) {
        at(i - OFFT_3, OT, ks, t, row, h); v = w.w3[(size_t)acc_k(ks, h) * HID + 32 * t + row];
    } else if (i < OFFT_1) {
        at(i - OFFT_2, OT, ks, t, row, h); v = w.w2[(size_t)acc_k(ks, h) * HID + 32 * t + row];
    } else {
        at(i - OFFT_1, OTI1, ks, t, row, h); v = w.w1[(size_t)acc_k(ks, h) * FD + 32 * t + row];
    }
    pt[i] = v;
}

struct BwdIn {
    const unsigned long long *masks;   // [4, n, 2] ReLU masks of h1..h4 (tn_mlp.hip: TRAIN)
    const float *sigma, *rgb;          // the forward's outputs [n], [n, 3]
    const float *d_sigma, *d_rgb;      // [n], [n, 3]
};
struct BwdOut {
    float *d1, *d2, *d3, *d4;  // [128, n] gradients w.r.t. the pre-activations of layers 1, 2, 3 and the head layer
    float *dhead;              // [4, n]   d sigma_raw, d rgb_raw[0..2]
    float *dx0;                // [n, 64]  gradient of the gathered features, SAMPLE-major rows (what the gather adjoint reads)
};

// 4 waves per block, one per SIMD (64 gradient values + 64 accumulators + the masks per lane; the 8-wave shape of the
// forward kernel would need them in 256 registers and spills).  A wave alone on its SIMD has nobody to hide its waits, so:
//   * the four weight stages of a group ping-pong between two LDS buffers: stage l + 1 is requested (async global -> LDS)
//     when GEMM l starts and has the whole GEMM to land; ONE barrier per layer;
//   * the stores of a GEMM's input (d4..d1: 16 quads per lane and tensor) are issued between the MFMAs of the first 32
//     k-steps of the GEMM that consumes it, so that they have drained when the next
//     `s_waitcnt vmcnt(0)` comes (on gfx9 stores and loads share that counter);
//   * the masks and head gradients of the NEXT group are requested a GEMM ahead.
constexpr int BWD_BLOCK = 256;
constexpr int PASS_FLOATS = BWD_BLOCK * 4;                                        // one async copy per thread = 16 bytes
constexpr int passes(size_t floats) { return (int)((floats + PASS_FLOATS - 1) / PASS_FLOATS); }   // stage copies: whole passes
constexpr int cmax(int a, int b) { return a > b ? a : b; }
constexpr int P_TH = passes(N_TH), P_T = passes(tfloats(OT)), P_T1 = passes(tfloats(OTI1));
constexpr bool TR_LDS = BWD_BLOCK == 256;                                        // d x0 rows through LDS (4 waves: it fits behind the last stage)
constexpr size_t TR_FLOATS = TR_LDS ? (BWD_BLOCK / 64) * 32 * 65 : 0;
constexpr size_t BUF_A = PASS_FLOATS * (size_t)cmax(P_TH, P_T);                  // stages 0, 2: Wh^T (+ vectors), W2^T
constexpr size_t BUF_B = cmax(PASS_FLOATS * cmax(P_T, P_T1), (int)(tfloats(OTI1) + TR_FLOATS));   // stages 1, 3: W3^T, W1^T (+ transposition)
static_assert((BUF_A + BUF_B) * sizeof(float) <= 160 * 1024, "two weight stages must fit the CU's LDS");
static_assert(OFFT_H + PASS_FLOATS * (size_t)P_TH <= PACKT_FLOATS, "whole-pass copy of the first stage stays inside the pack");
static_assert(OFFT_1 + PASS_FLOATS * (size_t)P_T1 <= PACKT_FLOATS, "whole-pass copy of the last stage stays inside the pack");

// PASSES x 4 KB of a packed layer -> LDS, every thread the same number of async loads (whole passes: what lies behind the
// layer in the pack lands in the buffer's padding)
template <int PASSES>
__device__ __forceinline__ void stage_fixed(float *lds, const float *__restrict__ src) {
    const float4 *s4 = reinterpret_cast<const float4 *>(src);
    float4 *d4 = reinterpret_cast<float4 *>(lds);
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave0 = __builtin_amdgcn_readfirstlane(threadIdx.x & ~63u);
#pragma unroll
    for (int it = 0; it < PASSES; ++it) {
        const uint32_t base = (uint32_t)it * BWD_BLOCK + wave0;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(s4 + base + lane),
                                         (__attribute__((address_space(3))) void *)(d4 + base), 16, 0, 0);
    }
}
// every vector-memory operation of this wave has completed (its share of the staged weights has landed, its stores have
// drained), then the block barrier
__device__ __forceinline__ void layer_top() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

}  // namespace

__global__ __launch_bounds__(BWD_BLOCK) void k_mlp_backward(size_t n, BwdIn in, const float *__restrict__ pt, BwdOut o) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *bufA = reinterpret_cast<float *>(smem), *bufB = bufA + BUF_A;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5;
    constexpr size_t GROUP = (BWD_BLOCK / 64) * 32;
    const size_t ngroups = (n + GROUP - 1) / GROUP;
    // the sample of this lane in group gg, clamped: lanes beyond the end recompute sample n - 1 and store the same values
    // to the same places as its owner
    auto sample_of = [&](size_t gg) {
        const size_t s = gg * GROUP + (size_t)wave * 32 + (lane & 31);
        return s < n ? s : n - 1;
    };
    // what a group needs of its samples before the first GEMM: four mask words and the head gradients
    struct Head { unsigned long long m1, m2, m3, m4; float dsr, dr0, dr1, dr2; };
    auto load_head = [&](size_t gg) {
        const size_t sc = sample_of(gg);
        Head q;
        q.m1 = in.masks[(0 * n + sc) * 2 + h]; q.m2 = in.masks[(1 * n + sc) * 2 + h];
        q.m3 = in.masks[(2 * n + sc) * 2 + h]; q.m4 = in.masks[(3 * n + sc) * 2 + h];
        // softplus(beta = 1, threshold = 20): derivative sigmoid(raw) = 1 - exp(-softplus(raw)) (1 beyond the threshold,
        // where the forward returned raw itself and 1 - exp(-raw) rounds to 1)
        q.dsr = in.d_sigma[sc] * -expm1f(-in.sigma[sc]);
        // rgb = sigmoid(c): d c = d rgb * rgb * (1 - rgb)
        const float y0 = in.rgb[3 * sc], y1 = in.rgb[3 * sc + 1], y2 = in.rgb[3 * sc + 2];
        q.dr0 = in.d_rgb[3 * sc] * (y0 * (1.0f - y0));
        q.dr1 = in.d_rgb[3 * sc + 1] * (y1 * (1.0f - y1));
        q.dr2 = in.d_rgb[3 * sc + 2] * (y2 * (1.0f - y2));
        return q;
    };

    stage_fixed<P_TH>(bufA, pt + OFFT_H);
    Head hd = load_head(blockIdx.x);

    for (size_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const size_t s = g * GROUP + (size_t)wave * 32 + (lane & 31);
        const size_t sc = s < n ? s : n - 1;     // columns of the lanes beyond the end: their owner's, same values
        float bin[KSH];
        f32x16 acc[OT];
        const unsigned long long m1 = hd.m1, m2 = hd.m2, m3 = hd.m3;
        const float dsr = hd.dsr;
        if (h == 0 && s < n) {
            o.dhead[s] = dsr;
            o.dhead[n + s] = hd.dr0; o.dhead[2 * n + s] = hd.dr1; o.dhead[3 * n + s] = hd.dr2;
        }
        // ---- d h3 = Wh[:, 27:]^T d_pre4 + wd * d sigma_raw, masked (A); d_pre4 = ReLU'(h4) Wr^T d c
        layer_top();
        stage_fixed<P_T>(bufB, pt + OFFT_3);
        {
            const float *cv = bufA + tfloats(OT) + DVEC;
            const float *w0 = cv + 64 * h, *w1 = cv + 128 + 64 * h, *w2 = cv + 256 + 64 * h;
#pragma unroll
            for (int j = 0; j < KSH; ++j) {
                const float v = (w0[j] * hd.dr0 + w1[j] * hd.dr1) + w2[j] * hd.dr2;
                bin[j] = ((hd.m4 >> j) & 1ull) ? v : 0.f;
            }
        }
        // the next group's masks and head gradients
        const Head hn = load_head(g + gridDim.x < ngroups ? g + gridDim.x : g);
        zero_acc(acc);
        gemm_steps_store<KSH, 0, OT, KSH>(acc, bin, bufA, lane, quad_ptr(o.d4, n, sc, h), 2 * n);
        {
            const float *dv = bufA + tfloats(OT) + 64 * h;
#pragma unroll
            for (int t = 0; t < OT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] += dv[t * 16 + r] * dsr;
            masked_to_bin(acc, m3, bin);
        }
        // ---- d h2 = W3^T d_pre3 (B), d h1 = W2^T d_pre2 (A)
        layer_top();
        stage_fixed<P_T>(bufA, pt + OFFT_2);
        zero_acc(acc);
        gemm_steps_store<KSH, 0, OT, KSH>(acc, bin, bufB, lane, quad_ptr(o.d3, n, sc, h), 2 * n);
        masked_to_bin(acc, m2, bin);
        layer_top();
        stage_fixed<P_T1>(bufB, pt + OFFT_1);
        zero_acc(acc);
        gemm_steps_store<KSH, 0, OT, KSH>(acc, bin, bufA, lane, quad_ptr(o.d2, n, sc, h), 2 * n);
        masked_to_bin(acc, m1, bin);
        // ---- d x0 = W1^T d_pre1  (64 input features = 2 tiles) (B); the next group's first stage goes to A meanwhile
        layer_top();
        stage_fixed<P_TH>(bufA, pt + OFFT_H);
        {
            f32x16 acc2[OTI1];
            zero_acc(acc2);
            gemm_steps_store<KSH, 0, OTI1, KSH>(acc2, bin, bufB, lane, quad_ptr(o.d1, n, sc, h), 2 * n);
            // d x0 leaves as SAMPLE-major rows [n, 64] (the gather adjoint reads a sample's gradient as one 256-byte line):
            // through this wave's slice of the tail of buffer B ([32 samples][65]: conflict-free both ways), each sample's
            // 64 values then go out as one coalesced store
            if constexpr (TR_LDS) {
                float *tr = bufB + tfloats(OTI1) + (size_t)wave * (32 * 65);
                {
                    float *col = tr + (lane & 31) * 65 + 4 * h;
#pragma unroll
                    for (int j = 0; j < OTI1 * 16; ++j) col[32 * (j >> 4) + (j & 3) + 8 * ((j >> 2) & 3)] = acc2[j >> 4][j & 15];
                }
                const size_t s0 = g * GROUP + (size_t)wave * 32;
#pragma unroll 8
                for (int i = 0; i < 32; ++i)
                    if (s0 + i < n) o.dx0[(s0 + i) * FD + lane] = tr[i * 65 + lane];
            } else if (s < n) {
                // every lane writes its 8 quads (4 consecutive features each) into its sample's row
                float *row = o.dx0 + s * FD + 4 * h;
#pragma unroll
                for (int j = 0; j < OTI1 * 16; j += 4)
                    *reinterpret_cast<float4 *>(row + 32 * (j >> 4) + 8 * ((j >> 2) & 3)) =
                        make_float4(acc2[j >> 4][j & 15], acc2[j >> 4][(j & 15) + 1], acc2[j >> 4][(j & 15) + 2], acc2[j >> 4][(j & 15) + 3]);
            }
        }
        hd = hn;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the last (unused) stage copy must not outlive the block's LDS
}

// Adjoint of k_composite (RaySamples.get_weights + RGB / accumulation renderers, model.py:632-638; the median depth has
// no gradient): one wavefront per ray.  With dd_i = delta_i sigma_i, T_i = exp(-sum_{k<i} dd_k), w_i = (1 - exp(-dd_i)) T_i
// and a_i = dL/dw_i = g_rgb . c_i - bg sum(g_rgb) + g_acc:
//     dL/d sigma_i = delta_i (a_i T_{i+1} - sum_{k>i} a_k w_k),      dL/d c_i = w_i g_rgb.
// The samples are swept from the far end (suffix sums by wave scans, carried across chunks of 64); the prefix of dd is
// the total minus the suffix.  Samples whose weight is not finite get zero gradients (nan_to_num in the forward).
__global__ __launch_bounds__(64) void k_composite_backward(size_t R, uint32_t S, const float *__restrict__ sigma,
                                                           const float *__restrict__ rgb, const float *__restrict__ edges,
                                                           Background background, const float *__restrict__ g_rgb,
                                                           const float *__restrict__ g_acc, float *__restrict__ d_sigma,
                                                           float *__restrict__ d_rgb) {
    const int lane = threadIdx.x;
    for (size_t ray = blockIdx.x; ray < R; ray += gridDim.x) {
        const float *e = edges + ray * (S + 1);
        const float gr = g_rgb ? g_rgb[3 * ray] : 0.f, gg = g_rgb ? g_rgb[3 * ray + 1] : 0.f, gb = g_rgb ? g_rgb[3 * ray + 2] : 0.f;
        const float ga = g_acc ? g_acc[ray] : 0.f;
        const float a_const = ga - ((background.r * gr + background.g * gg) + background.b * gb);
        // total of dd
        float tot = 0.f;
        for (uint32_t j = lane; j < S; j += 64) tot += (e[j + 1] - e[j]) * sigma[ray * S + j];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off);
        float carry_dd = 0.f, carry_aw = 0.f;   // suffix sums over the chunks already processed (farther samples)
        const uint32_t nchunks = (S + 63) / 64;
        for (uint32_t c = nchunks; c-- > 0;) {
            const uint32_t j = c * 64 + lane;
            const bool ok = j < S;
            const size_t q = ray * S + (ok ? j : S - 1);
            const float delta = ok ? e[j + 1] - e[j] : 0.f;
            const float dd = ok ? delta * sigma[q] : 0.f;
            // inclusive suffix sum of dd within the chunk
            float suf = dd;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const float o2 = __shfl_down(suf, off);
                if (lane + off < 64) suf += o2;
            }
            const float excl = tot - (suf + carry_dd);          // sum_{k<j} dd_k
            const float Ti = expf(-excl), Tn = expf(-(excl + dd));
            float w = (1.0f - expf(-dd)) * Ti;
            const bool fin = ok && (w == w) && fabsf(w) <= 3.0e38f;
            if (!fin) w = 0.f;
            const float c0 = rgb[3 * q], c1 = rgb[3 * q + 1], c2 = rgb[3 * q + 2];
            const float ai = ((gr * c0 + gg * c1) + gb * c2) + a_const;
            const float aw = fin ? ai * w : 0.f;
            float sufaw = aw;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const float o2 = __shfl_down(sufaw, off);
                if (lane + off < 64) sufaw += o2;
            }
            const float later = (sufaw - aw) + carry_aw;        // sum_{k>j} a_k w_k
            if (ok) {
                float ds = delta * (ai * Tn - later);
                if (!fin || !(ds == ds)) ds = 0.f;
                d_sigma[q] = ds;
                d_rgb[3 * q] = w * gr; d_rgb[3 * q + 1] = w * gg; d_rgb[3 * q + 2] = w * gb;
            }
            carry_dd += __shfl(suf, 0);
            carry_aw += __shfl(sufaw, 0);
        }
    }
}

void launch_composite_backward(size_t R, uint32_t S, const float *sigma, const float *rgb, const float *edges, Background background,
                               const float *d_out_rgb, const float *d_out_acc, float *d_sigma, float *d_rgb, hipStream_t stream) {
    if (R == 0 || S == 0) return;
    const unsigned grid = (unsigned)(R < 256u * 32u ? R : 256u * 32u);
    hipLaunchKernelGGL(k_composite_backward, dim3(grid), dim3(64), 0, stream, R, S, sigma, rgb, edges, background, d_out_rgb, d_out_acc,
                       d_sigma, d_rgb);
}

size_t mlp_backward_pack_floats() { return PACKT_FLOATS; }

void launch_mlp_pack_t(const MlpWeights &w, float *pt, hipStream_t stream) {
    hipLaunchKernelGGL(k_mlp_pack_t, dim3((unsigned)((PACKT_FLOATS + 255) / 256)), dim3(256), 0, stream, w, pt);
}

void launch_mlp_backward(size_t n, const float *sigma, const float *rgb, const MlpPacks &w, const float *d_sigma, const float *d_rgb,
                         const MlpBackwardBuffers &b, hipStream_t stream) {
    if (n == 0) return;
    const size_t smem = (BUF_A + BUF_B) * sizeof(float);
    static PerDeviceOnce lds_attr;
    lds_attr.run([&] { allow_dynamic_lds(reinterpret_cast<const void *>(k_mlp_backward), smem); });
    const size_t group = (BWD_BLOCK / 64) * 32;
    const size_t ngroups = (n + group - 1) / group;
    const unsigned grid = (unsigned)(ngroups < 256 ? ngroups : 256);  // one 4-wave block per CU
    hipLaunchKernelGGL(k_mlp_backward, dim3(grid), dim3(BWD_BLOCK), smem, stream, n,
                       BwdIn{b.masks, sigma, rgb, d_sigma, d_rgb}, w.pt, BwdOut{b.d1, b.d2, b.d3, b.d4, b.dhead, b.dx0});
}

// Gradient of the per-ray head bias (tn_mlp_common.h: add_ray_bias; the appearance embedding of model.py:608-620):
// out[ray][f] = sum over the ray's S samples of d4[f][sample].  d4 is quad-major [32][n][4]: one 256-thread block per ray,
// thread = (quad, one of 8 consecutive samples): every 8 threads read 128 contiguous bytes per step; the 8 partial sums are
// combined by shuffles in a fixed order (bit-reproducible).
__global__ __launch_bounds__(256) void k_ray_head_grad(size_t n, uint32_t S, const float *__restrict__ d4, float *__restrict__ out) {
    const uint32_t quad = threadIdx.x >> 3, sub = threadIdx.x & 7;
    const size_t rays = n / S;
    const float4 *src = reinterpret_cast<const float4 *>(d4) + (size_t)quad * n;
    for (size_t ray = blockIdx.x; ray < rays; ray += gridDim.x) {
        const size_t s0 = ray * S;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (uint32_t j = sub; j < S; j += 8) {
            const float4 v = src[s0 + j];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
#pragma unroll
        for (int off = 1; off < 8; off <<= 1) {
            a.x += __shfl_xor(a.x, off); a.y += __shfl_xor(a.y, off); a.z += __shfl_xor(a.z, off); a.w += __shfl_xor(a.w, off);
        }
        if (sub == 0) *reinterpret_cast<float4 *>(out + ray * HID + 4 * quad) = a;
    }
}

void launch_ray_head_grad(size_t n, uint32_t samples_per_ray, const float *d4, float *out, hipStream_t stream) {
    if (n == 0 || samples_per_ray == 0) return;
    const size_t rays = n / samples_per_ray;
    hipLaunchKernelGGL(k_ray_head_grad, dim3((unsigned)(rays < 256 * 8 ? rays : 256 * 8)), dim3(256), 0, stream, n, samples_per_ray, d4, out);
}

}  // namespace tn
