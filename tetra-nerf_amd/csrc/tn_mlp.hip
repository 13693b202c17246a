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
#include "tn_mlp_fwd.h"
#include "tn_ray_ops.h"

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
    } else if (i < OFF_WHEAD + lfloats(HEAD_KS, OT)) {  // head: the 128 base columns of [enc(27) | base(128)] -> 128
        split(i - OFF_WHEAD, OT, ks, ot, lane);
        const int row = lane & 31, h = lane >> 5;
        const int o = 32 * ot + row;
        const size_t base = (size_t)o * (ENC + HID);
        if (ks < HEAD_KS) v = w.wh[base + ENC + acc_k(ks, h)];
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

// the encoding's 27 columns of mlp_head, [128][ENC_PAD] (column 27 zero): the operand of head_ray_term
__global__ void k_pack_wenc(MlpWeights w, float *__restrict__ wenc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= HID * ENC_PAD) return;
    const int o = i / ENC_PAD, k = i % ENC_PAD;
    wenc[i] = k < ENC ? w.wh[(size_t)o * (ENC + HID) + k] : 0.f;
}

// hterm[r][o] = Wh[o, :27] . enc(dir_r) (+ the caller's per-ray bias: the appearance embedding): what the head layer adds per RAY
__global__ __launch_bounds__(256) void k_head_ray_term(size_t R, const float *__restrict__ enc, const float *__restrict__ wenc,
                                                       const float *__restrict__ ray_bias, float *__restrict__ hterm) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * HID) return;
    const size_t r = i / HID;
    const int o = (int)(i % HID);
    float t = head_ray_term(wenc + o * ENC_PAD, enc + r * ENC_PAD);
    if (ray_bias) t += ray_bias[i];
    hterm[i] = t;
}

}  // namespace

// 8 waves (BLOCK = 512) share each staged layer, one block per CU, two waves per SIMD: while one waits for a weight copy, at
// a barrier or in an epilogue the other one's MFMAs run.  (Round 2's 4-wave / two-blocks-per-CU variant with the head layer
// staged in two halves measured neutral, profiles/r02o_mlp_block.txt, and is gone.)  The loop body lives in tn_mlp_fwd.h
// (mlp_forward_group): the persistent render kernel (tn_render_rays.hip) runs the same code on its tiles.
template <bool GATHER, bool DENSITY_ONLY, int BLOCK = MLP_BLOCK, bool TRAIN = false>
__global__ __launch_bounds__(BLOCK, 2) void k_mlp_forward(size_t n, uint32_t samples_per_ray, const float *__restrict__ feats,
                                                           const uint32_t *__restrict__ vi, const float *__restrict__ bc,
                                                           const float *__restrict__ fieldT,
                                                           const float *__restrict__ hterm, const float *__restrict__ pk,
                                                           float *__restrict__ sigma, float *__restrict__ rgb, FwdSave sv,
                                                           const uint32_t *__restrict__ count) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *lds = reinterpret_cast<float *>(smem);
    // count (nullable): the number of RAYS lives on the device (sync-free callers launch over an upper bound)
    if (count) n = (size_t)*count * samples_per_ray;
    constexpr size_t GROUP = (BLOCK / 64) * 32;
    const size_t ngroups = (n + GROUP - 1) / GROUP;
    if constexpr (TRAIN) {
        FwdCarry cy;
#pragma unroll
        for (int j = 0; j < KSH; ++j) cy.h4[j] = 0.f;
        cy.p = nullptr; cy.m = nullptr;
        for (size_t g = blockIdx.x; g < ngroups; g += gridDim.x)
            mlp_forward_group<GATHER, DENSITY_ONLY, BLOCK, TRAIN>(lds, g, n, samples_per_ray, feats, vi, bc, fieldT, hterm, pk, sigma, rgb, sv, &cy);
        flush_carry(cy, n);
    } else {
        for (size_t g = blockIdx.x; g < ngroups; g += gridDim.x)
            mlp_forward_group<GATHER, DENSITY_ONLY, BLOCK, TRAIN>(lds, g, n, samples_per_ray, feats, vi, bc, fieldT, hterm, pk, sigma, rgb, sv);
    }
}

// Per-ray composite: one wavefront per ray, lanes stride the samples; exclusive scan of sigma*delta.
__global__ __launch_bounds__(64) void k_composite(size_t R, uint32_t S, const float *__restrict__ sigma,
                                                  const float *__restrict__ rgb, const float *__restrict__ edges,
                                                  Background background, float *__restrict__ out_rgb,
                                                  float *__restrict__ out_acc, float *__restrict__ out_depth,
                                                  float *__restrict__ out_weights, const uint32_t *__restrict__ ray_index,
                                                  const uint32_t *__restrict__ count) {
    const int lane = threadIdx.x;
    if (count) R = *count;
    for (size_t ray = blockIdx.x; ray < R; ray += gridDim.x) {
        const size_t dst = ray_index ? (size_t)ray_index[ray] : ray;   // (scatter into the frame buffers of all rays)
        // (one chunk at a time: 39 registers, 8 waves per SIMD hide the scans' latency here; the persistent render kernel, 8 waves
        //  per CU, interleaves up to 9 chunks -- same bits for any grouping, tn_ray_ops.h)
        rayops::ray_composite_chunks<1>(S, sigma + ray * S, rgb ? rgb + 3 * ray * S : nullptr, edges + ray * (S + 1), background,
                              out_rgb ? out_rgb + 3 * dst : nullptr, out_acc ? out_acc + dst : nullptr,
                              out_depth ? out_depth + dst : nullptr, out_weights ? out_weights + ray * S : nullptr, lane);
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

void launch_pack_wenc(const MlpWeights &w, float *wenc, hipStream_t stream) {
    hipLaunchKernelGGL(k_pack_wenc, dim3((HID * ENC_PAD + 255) / 256), dim3(256), 0, stream, w, wenc);
}

// the head layer's per-ray term of a call: direction encodings (w.enc) -> w.hterm (+ w.ray_bias)
void launch_head_ray_term(size_t num_rays, const float *dirs, const MlpPacks &w, hipStream_t stream) {
    if (num_rays == 0) return;
    hipLaunchKernelGGL(k_dir_encoding, dim3((unsigned)((num_rays + 255) / 256)), dim3(256), 0, stream, num_rays, dirs, w.enc);
    hipLaunchKernelGGL(k_head_ray_term, dim3((unsigned)((num_rays * HID + 255) / 256)), dim3(256), 0, stream, num_rays, w.enc, w.wenc,
                       w.ray_bias, w.hterm);
}

size_t mlp_enc_floats_per_ray() { return 32; }

void launch_mlp_forward(size_t n, uint32_t samples_per_ray, size_t num_rays, const float *feats, const uint32_t *vi,
                        const float *bc, const float *fieldT, const float *dirs, const MlpPacks &w, float *sigma, float *rgb,
                        hipStream_t stream, const uint32_t *count) {
    if (n == 0) return;
    const bool gather = feats == nullptr;
    const bool density_only = rgb == nullptr;  // coarse pass: no colour head, no direction encoding
    if (density_only) num_rays = 0;
    const float *pk = gather ? w.pk_gather : w.pk_plain;
    launch_head_ray_term(num_rays, dirs, w, stream);
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
                       fieldT, w.hterm, pk, sigma, rgb, FwdSave{}, count)
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
    launch_head_ray_term(num_rays, dirs, w, stream);
    const size_t smem = MAX_STAGE_FLOATS * sizeof(float);
    static PerDeviceOnce lds_attr;
    lds_attr.run([&] { allow_dynamic_lds(reinterpret_cast<const void *>(k_mlp_forward<true, false, MLP_BLOCK, true>), smem); });
    const size_t group = (MLP_BLOCK / 64) * 32;
    const size_t ngroups = (n + group - 1) / group;
    const unsigned grid = (unsigned)(ngroups < 256 ? ngroups : 256);
    hipLaunchKernelGGL((k_mlp_forward<true, false, MLP_BLOCK, true>), dim3(grid), dim3(MLP_BLOCK), smem, stream, n, samples_per_ray,
                       (const float *)nullptr, vi, bc, fieldT, w.hterm, w.pk_gather, sigma, rgb,
                       FwdSave{save.x0, save.h1, save.h2, save.h3, save.h4, save.masks}, (const uint32_t *)nullptr);
}

void launch_composite(size_t R, uint32_t S, const float *sigma, const float *rgb, const float *edges, Background background,
                      float *out_rgb, float *out_acc, float *out_depth, float *out_weights, hipStream_t stream,
                      const uint32_t *ray_index, const uint32_t *count) {
    if (R == 0 || S == 0) return;
    const unsigned grid = (unsigned)(R < 256u * 32u ? R : 256u * 32u);
    hipLaunchKernelGGL(k_composite, dim3(grid), dim3(64), 0, stream, R, S, sigma, rgb, edges, background, out_rgb, out_acc,
                       out_depth, out_weights, ray_index, count);
}

}  // namespace tn
