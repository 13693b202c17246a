// tn_mlp.hip -- fused shallow MLP + heads of the Tetra-NeRF model on fp32 MFMA, and the per-ray
// volume-render composite (inference forward).
//
// Replaces the PyTorch/cuBLAS chain of the reference's model
//   mlp_base 64->128->128->128 (ReLU, ReLU out), density head 128->1 + softplus,
//   NeRFEncoding(dir) ++ base -> mlp_head 155->128 ReLU, rgb head 128->3 + sigmoid
//   (tetranerf/nerfstudio/model.py:414-455 built, :602-621 run), and
//   RaySamples.get_weights + RGB/accumulation/depth renderers (:632-638).
//
// This part of the path really is a dense GEMM (122,624 FLOP per sample, K in {64,128,155}), so it
// runs on the matrix cores -- in fp32, because the parity bar is 1e-5: v_mfma_f32_32x32x2_f32 is an
// exact fp32 fma chain at the fp32 vector rate (157 TFLOP/s peak).
//
// Dataflow (the point of the design): one wavefront owns a tile of 32 samples and computes the
// TRANSPOSED products  Y^T[out][sample] = W[out][k] * X^T[k][sample].  The MFMA C/D layout of a
// 32x32 tile is  col = lane & 31 (= sample), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (= feature), and
// the B operand wants lane (h = lane>>5, s = lane&31) to hold X^T[k_h][s] -- so accumulator register r
// of a tile can be fed STRAIGHT BACK as the B operand of the next layer if that layer consumes its K
// indices in the order {(r&3)+8*(r>>2), (r&3)+8*(r>>2)+4}.  The weights (A operands) are pre-permuted to
// exactly that K order once per call (k_mlp_pack), so activations never leave the registers between
// layers: no LDS round trip, no cross-lane shuffles, no HBM traffic besides the [64,n] input and the
// 16 B/sample output.  Weights are staged per layer in LDS (<= 80 KB) and shared by the 4 waves of a
// block; every MFMA reads its A operand as one conflict-free 256-B ds_read_b32.
#include "tn_mlp_common.h"

namespace tn {

using namespace mlp;


namespace {

__global__ void k_mlp_pack(MlpWeights w, float *__restrict__ pk, int gather_l1) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= PACK_FLOATS) return;
    float v = 0.f;
    auto split = [](size_t j, int tiles, int &ks, int &ot, int &lane) {
        lane = (int)(j & 63); ot = (int)((j >> 6) % tiles); ks = (int)(j / (64 * (size_t)tiles));
    };
    int ks, ot, lane;
    if (i < OFF_W2) {                       // layer 1: natural K order (input from memory)
        split(i - OFF_W1, OT, ks, ot, lane);
        const int o = 32 * ot + (lane & 31), h = lane >> 5;
        // k-step ks consumes features (2ks, 2ks+1) when the input is read from the [64,n] buffer, and
        // (ks, 32+ks) when the wave gathers it itself (each half-wave owns 32 contiguous features)
        v = ks < KS1 ? w.w1[(size_t)o * FD + (gather_l1 ? 32 * h + ks : 2 * ks + h)] : (h == 0 ? w.b1[o] : 0.f);
    } else if (i < OFF_W3) {
        split(i - OFF_W2, OT, ks, ot, lane);
        const int o = 32 * ot + (lane & 31), h = lane >> 5;
        v = ks < KSH ? w.w2[(size_t)o * HID + acc_k(ks, h)] : (h == 0 ? w.b2[o] : 0.f);
    } else if (i < OFF_W3 + lfloats(KSH, OT)) {
        split(i - OFF_W3, OT, ks, ot, lane);
        const int o = 32 * ot + (lane & 31), h = lane >> 5;
        v = ks < KSH ? w.w3[(size_t)o * HID + acc_k(ks, h)] : (h == 0 ? w.b3[o] : 0.f);
    } else if (i < OFF_WHEAD) {             // density head vector behind layer 3
        const int j = (int)(i - OFF_W3 - lfloats(KSH, OT));
        if (j < 128) v = w.wd[acc_k(j & 63, j >> 6)];
        else if (j == 128) v = w.bd[0];
    } else if (i < OFF_WHEAD + lfloats(HEAD_KS, OT)) {  // head [enc(27) | base(128)] -> 128
        split(i - OFF_WHEAD, OT, ks, ot, lane);
        const int row = lane & 31, h = lane >> 5;
        const int o = 32 * ot + row;
        const size_t base = (size_t)o * (ENC + HID);
        if (ks < KSE) { const int k = 2 * ks + h; v = k < ENC ? w.wh[base + k] : 0.f; }
        else if (ks < HEAD_KS) v = w.wh[base + ENC + acc_k(ks - KSE, h)];
        else v = h == 0 ? w.bh[o] : 0.f;
    } else {                                // rgb head vectors behind the head layer
        const int j = (int)(i - OFF_WHEAD - lfloats(HEAD_KS, OT));
        if (j < 384) v = w.wr[(size_t)(j >> 7) * HID + acc_k(j & 63, (j >> 6) & 1)];
        else if (j < 387) v = w.br[j - 384];
    }
    pk[i] = v;
}

// direction encoding per ray (padded to 28): NeRFEncoding(3, 4 freqs 2^linspace(0,4,4), include_input)
__global__ void k_dir_encoding(size_t R, const float *__restrict__ dirs, float *__restrict__ enc) {
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const float two_pi = 6.283185307179586f, half_pi = 1.5707963267948966f;
    const float freqs[4] = {1.0f, 2.5198421478271484f, 6.349603652954102f, 16.0f};  // fp32(2**(4*i/3))
    float *e = enc + r * ENC_PAD;
    for (int c = 0; c < 3; ++c) {
        const float x = two_pi * dirs[3 * r + c];
        for (int f = 0; f < 4; ++f) {
            const float s = x * freqs[f];
            e[c * 4 + f] = sinf(s);
            e[12 + c * 4 + f] = sinf(s + half_pi);
        }
        e[24 + c] = dirs[3 * r + c];
    }
    e[27] = 0.f;
}

}  // namespace

// 8 waves (BLOCK = 512) share each staged layer, one block per CU, two waves per SIMD: while one waits for a weight copy, at
// a barrier or in an epilogue the other one's MFMAs run.  (Round 2's 4-wave / two-blocks-per-CU variant with the head layer
// staged in two halves measured neutral, profiles/r02o_mlp_block.txt, and is gone.)
// TRAIN: the layer inputs x0, h1..h4 (quad-major [F/4][n][4], what the weight-gradient GEMMs contract) and the ReLU masks (all the
// dX kernel needs) are saved on the way -- the backward pass recomputes nothing (round 3a recomputed the whole forward
// inside the dX kernel: 2.2 of its 5 ms).
struct FwdSave { float *x0, *h1, *h2, *h3, *h4; unsigned long long *masks; };
template <bool GATHER, bool DENSITY_ONLY, int BLOCK = MLP_BLOCK, bool TRAIN = false>
__global__ __launch_bounds__(BLOCK, 2) void k_mlp_forward(size_t n, uint32_t samples_per_ray, const float *__restrict__ feats,
                                                           const uint32_t *__restrict__ vi, const float *__restrict__ bc,
                                                           const float *__restrict__ fieldT,
                                                           const float *__restrict__ enc, const float *__restrict__ pk,
                                                           float *__restrict__ sigma, float *__restrict__ rgb, FwdSave sv,
                                                           const float *__restrict__ ray_bias) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *lds = reinterpret_cast<float *>(smem);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5;
    constexpr size_t GROUP = (BLOCK / 64) * 32;
    const size_t ngroups = (n + GROUP - 1) / GROUP;

    for (size_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const size_t s = g * GROUP + (size_t)wave * 32 + (lane & 31);
        const size_t sc = s < n ? s : n - 1;  // clamped: out-of-range lanes compute a duplicate, store nothing
        float bin[KSH];

        // ---- layer 1: 64 -> 128, B operands straight from the feature-major input [64, n]
        __syncthreads();
        stage_weights<BLOCK>(lds, pk + OFF_W1, lfloats(KS1, OT));
        if constexpr (!GATHER) {
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) bin[ks] = feats[(size_t)(2 * ks + h) * n + sc];
        } else {
            // fused barycentric gather (interpolate_values<4>, same summation order => same bits as the
            // stand-alone op): this lane produces features 32h .. 32h+31 of its sample straight into the
            // B-operand registers; the [64, n] feature buffer never exists.
            const uint4 v4 = *reinterpret_cast<const uint4 *>(vi + 4 * sc);
            const float b0 = bc[3 * sc], b1 = bc[3 * sc + 1], b2 = bc[3 * sc + 2];
            const float w0 = 1.0f - ((b0 + b1) + b2);
            const uint32_t vv[4] = {v4.y, v4.z, v4.w, v4.x};
            const float ww[4] = {b0, b1, b2, w0};
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) bin[ks] = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (vv[k] != TN_EMPTY) {
                    const float4 *row = reinterpret_cast<const float4 *>(fieldT + (size_t)vv[k] * FD + 32 * h);
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const float4 x = row[q];
                        bin[4 * q] += ww[k] * x.x; bin[4 * q + 1] += ww[k] * x.y;
                        bin[4 * q + 2] += ww[k] * x.z; bin[4 * q + 3] += ww[k] * x.w;
                    }
                }
            }
        }
        stage_wait();
        {
            f32x16 acc[OT];
            zero_acc(acc);
            // TRAIN: every GEMM's input leaves for HBM under the GEMM's own MFMAs (lanes beyond the end store their
            // duplicate of sample n - 1 where its owner stores it)
            if constexpr (TRAIN) gemm_steps_store<KS1, 0, OT, KS1>(acc, bin, lds, lane, quad_ptr_x0(sv.x0, n, sc, h), n);
            else gemm_steps<KS1, 0, OT>(acc, bin, lds, lane);
            bias_step<KS1, OT>(acc, lds, lane);
            relu_to_bin(acc, bin);
        }
        auto save_mask = [&](int layer) {
            if constexpr (TRAIN) sv.masks[((size_t)layer * n + sc) * 2 + h] = mask_of(bin);
        };
        save_mask(0);
        // ---- layers 2, 3: 128 -> 128, accumulators fed back as B operands
        __syncthreads();
        stage_weights<BLOCK>(lds, pk + OFF_W2, lfloats(KSH, OT));
        stage_wait();
        {
            f32x16 acc[OT];
            zero_acc(acc);
            if constexpr (TRAIN) gemm_steps_store<KSH, 0, OT, KSH>(acc, bin, lds, lane, quad_ptr(sv.h1, n, sc, h), 2 * n);
            else gemm_steps<KSH, 0, OT>(acc, bin, lds, lane);
            bias_step<KSH, OT>(acc, lds, lane);
            relu_to_bin(acc, bin);
        }
        save_mask(1);
        __syncthreads();
        stage_weights<BLOCK>(lds, pk + OFF_W3, N_W3);
        stage_wait();
        {
            f32x16 acc[OT];
            zero_acc(acc);
            if constexpr (TRAIN) gemm_steps_store<KSH, 0, OT, KSH>(acc, bin, lds, lane, quad_ptr(sv.h2, n, sc, h), 2 * n);
            else gemm_steps<KSH, 0, OT>(acc, bin, lds, lane);
            bias_step<KSH, OT>(acc, lds, lane);
            relu_to_bin(acc, bin);  // mlp_base out_activation = ReLU
        }
        save_mask(2);
        {
            // density head 128 -> 1 + softplus on the VALU (the vector rides behind layer 3's weights)
            const float *dv = lds + lfloats(KSH, OT);
            const float raw = head_dot(dv + 64 * h, bin) + dv[128];
            const float sp = raw > 20.0f ? raw : log1pf(expf(raw));  // torch softplus(beta=1, threshold=20)
            if (h == 0 && s < n) sigma[s] = sp;
        }
        if constexpr (DENSITY_ONLY) continue;  // coarse pass of the model (model.py:577-581)
        // ---- head [enc(27) | base(128)] -> 128 ReLU
        __syncthreads();
        stage_weights<BLOCK>(lds, pk + OFF_WHEAD, N_WHEAD);
        stage_wait();
        {
            f32x16 acc[OT];
            zero_acc(acc);
            const float *e = enc + (sc / samples_per_ray) * ENC_PAD;
#pragma unroll
            for (int ks = 0; ks < KSE; ++ks) {
                const float b = e[2 * ks + h];
                const float *wrow = lds + (size_t)ks * OT * 64 + lane;
#pragma unroll
                for (int t = 0; t < OT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wrow[t * 64], b, acc[t], 0, 0, 0);
            }
            if constexpr (TRAIN) gemm_steps_store<KSH, KSE, OT, KSH>(acc, bin, lds, lane, quad_ptr(sv.h3, n, sc, h), 2 * n);
            else gemm_steps<KSH, KSE, OT>(acc, bin, lds, lane);
            bias_step<HEAD_KS, OT>(acc, lds, lane);
            if (ray_bias) add_ray_bias(acc, ray_bias + (sc / samples_per_ray) * HID, h);   // wave-uniform test
            relu_to_bin(acc, bin);
        }
        if constexpr (TRAIN) store_bin(sv.h4, n, sc, bin, h);   // the last layer's output has no GEMM to hide under
        save_mask(3);
        {
            // rgb head 128 -> 3 + sigmoid on the VALU
            const float *cv = lds + lfloats(HEAD_KS, OT);
            const float c0 = head_dot(cv + 64 * h, bin) + cv[384];
            const float c1 = head_dot(cv + 128 + 64 * h, bin) + cv[385];
            const float c2 = head_dot(cv + 256 + 64 * h, bin) + cv[386];
            if (h == 0 && s < n) {
                rgb[3 * s] = 1.0f / (1.0f + expf(-c0));
                rgb[3 * s + 1] = 1.0f / (1.0f + expf(-c1));
                rgb[3 * s + 2] = 1.0f / (1.0f + expf(-c2));
            }
        }
    }
}

// Per-ray composite: one wavefront per ray, lanes stride the samples; exclusive scan of sigma*delta.
__global__ __launch_bounds__(64) void k_composite(size_t R, uint32_t S, const float *__restrict__ sigma,
                                                  const float *__restrict__ rgb, const float *__restrict__ edges,
                                                  Background background, float *__restrict__ out_rgb,
                                                  float *__restrict__ out_acc, float *__restrict__ out_depth,
                                                  float *__restrict__ out_weights) {
    const int lane = threadIdx.x;
    for (size_t ray = blockIdx.x; ray < R; ray += gridDim.x) {
        const float *e = edges + ray * (S + 1);
        float carry = 0.f;       // sum of sigma*delta of all previous samples
        float cw = 0.f;          // running sum of weights (for the median depth)
        float r0 = 0.f, r1 = 0.f, r2 = 0.f, accw = 0.f;
        float depth = 0.f;
        bool found = false;
        for (uint32_t base = 0; base < S; base += 64) {
            const uint32_t j = base + lane;
            const bool ok = j < S;
            const size_t q = ray * S + (ok ? j : S - 1);
            const float st = e[ok ? j : S - 1], en = e[(ok ? j : S - 1) + 1];
            const float dd = ok ? (en - st) * sigma[q] : 0.f;
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
            if (out_weights && ok) out_weights[q] = w;
            if (rgb) {
                float c0 = rgb[3 * q], c1 = rgb[3 * q + 1], c2 = rgb[3 * q + 2];
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
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            r0 += __shfl_xor(r0, off); r1 += __shfl_xor(r1, off); r2 += __shfl_xor(r2, off); accw += __shfl_xor(accw, off);
        }
        if (!found) depth = 0.5f * (e[S - 1] + e[S]);  // searchsorted clamps to the last sample
        if (lane == 0 && out_rgb) {
            float o0 = r0 + background.r * (1.0f - accw), o1 = r1 + background.g * (1.0f - accw), o2 = r2 + background.b * (1.0f - accw);
            if (background.clamp) { o0 = fminf(fmaxf(o0, 0.f), 1.f); o1 = fminf(fmaxf(o1, 0.f), 1.f); o2 = fminf(fmaxf(o2, 0.f), 1.f); }
            out_rgb[3 * ray] = o0; out_rgb[3 * ray + 1] = o1; out_rgb[3 * ray + 2] = o2;
            out_acc[ray] = accw;
            out_depth[ray] = depth;
        }
    }
}

size_t mlp_pack_floats() { return PACK_FLOATS + 1024; }   // + slack: the training kernel copies whole 4 KB passes (tn_mlp_bwd.hip)


void launch_mlp_pack(const MlpWeights &w, float *pk, bool gather_l1, hipStream_t stream) {
    hipLaunchKernelGGL(k_mlp_pack, dim3((unsigned)((PACK_FLOATS + 255) / 256)), dim3(256), 0, stream, w, pk, gather_l1 ? 1 : 0);
}

void launch_dir_encoding(size_t num_rays, const float *dirs, float *enc, hipStream_t stream) {
    if (num_rays == 0) return;
    hipLaunchKernelGGL(k_dir_encoding, dim3((unsigned)((num_rays + 255) / 256)), dim3(256), 0, stream, num_rays, dirs, enc);
}

size_t mlp_enc_floats_per_ray() { return 32; }

void launch_mlp_forward(size_t n, uint32_t samples_per_ray, size_t num_rays, const float *feats, const uint32_t *vi,
                        const float *bc, const float *fieldT, const float *dirs, const MlpPacks &w, float *sigma, float *rgb,
                        hipStream_t stream) {
    if (n == 0) return;
    const bool gather = feats == nullptr;
    const bool density_only = rgb == nullptr;  // coarse pass: no colour head, no direction encoding
    if (density_only) num_rays = 0;
    const float *pk = gather ? w.pk_gather : w.pk_plain;
    float *enc = w.enc;
    if (num_rays)
        hipLaunchKernelGGL(k_dir_encoding, dim3((unsigned)((num_rays + 255) / 256)), dim3(256), 0, stream, num_rays, dirs, enc);
    // one 8-wave block per CU (4-wave blocks, two per CU, measured neutral: profiles/r02o_mlp_block.txt)
    const size_t smem = MAX_STAGE_FLOATS * sizeof(float);  // the largest staged layer
    static PerDeviceOnce lds_attr;
    lds_attr.run([&] {
        allow_dynamic_lds(reinterpret_cast<const void *>(k_mlp_forward<false, false>), smem);
        allow_dynamic_lds(reinterpret_cast<const void *>(k_mlp_forward<true, false>), smem);
        allow_dynamic_lds(reinterpret_cast<const void *>(k_mlp_forward<false, true>), smem);
        allow_dynamic_lds(reinterpret_cast<const void *>(k_mlp_forward<true, true>), smem);
    });
    const size_t group = (MLP_BLOCK / 64) * 32;
    const size_t ngroups = (n + group - 1) / group;
    const unsigned grid = (unsigned)(ngroups < 256 ? ngroups : 256);
#define TN_MLP_LAUNCH(G, D)                                                                                         \
    hipLaunchKernelGGL((k_mlp_forward<G, D>), dim3(grid), dim3(MLP_BLOCK), smem, stream, n, samples_per_ray, feats, vi, bc, \
                       fieldT, enc, pk, sigma, rgb, FwdSave{}, w.ray_bias)
    if (gather && density_only) TN_MLP_LAUNCH(true, true);
    else if (gather) TN_MLP_LAUNCH(true, false);
    else if (density_only) TN_MLP_LAUNCH(false, true);
    else TN_MLP_LAUNCH(false, false);
#undef TN_MLP_LAUNCH
}

void launch_mlp_forward_train(size_t n, uint32_t samples_per_ray, size_t num_rays, const uint32_t *vi, const float *bc,
                              const float *fieldT, const float *dirs, const MlpPacks &w, float *sigma, float *rgb,
                              const MlpBackwardBuffers &save, hipStream_t stream) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_dir_encoding, dim3((unsigned)((num_rays + 255) / 256)), dim3(256), 0, stream, num_rays, dirs, w.enc);
    const size_t smem = MAX_STAGE_FLOATS * sizeof(float);
    static PerDeviceOnce lds_attr;
    lds_attr.run([&] { allow_dynamic_lds(reinterpret_cast<const void *>(k_mlp_forward<true, false, MLP_BLOCK, true>), smem); });
    const size_t group = (MLP_BLOCK / 64) * 32;
    const size_t ngroups = (n + group - 1) / group;
    const unsigned grid = (unsigned)(ngroups < 256 ? ngroups : 256);
    hipLaunchKernelGGL((k_mlp_forward<true, false, MLP_BLOCK, true>), dim3(grid), dim3(MLP_BLOCK), smem, stream, n, samples_per_ray,
                       (const float *)nullptr, vi, bc, fieldT, w.enc, w.pk_gather, sigma, rgb,
                       FwdSave{save.x0, save.h1, save.h2, save.h3, save.h4, save.masks}, w.ray_bias);
}

void launch_composite(size_t R, uint32_t S, const float *sigma, const float *rgb, const float *edges, Background background,
                      float *out_rgb, float *out_acc, float *out_depth, float *out_weights, hipStream_t stream) {
    if (R == 0 || S == 0) return;
    const unsigned grid = (unsigned)(R < 256u * 32u ? R : 256u * 32u);
    hipLaunchKernelGGL(k_composite, dim3(grid), dim3(64), 0, stream, R, S, sigma, rgb, edges, background, out_rgb, out_acc,
                       out_depth, out_weights);
}

}  // namespace tn
