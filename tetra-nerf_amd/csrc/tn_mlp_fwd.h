// tn_mlp_fwd.h -- one 256-sample group of the fused forward MLP (gather + mlp_base + density head [+ mlp_head + rgb head]) as a
// device function: the loop body of k_mlp_forward (tn_mlp.hip), shared with the persistent render kernel (tn_render_rays.hip),
// which runs the SAME instruction stream on its tiles -- results are bit-identical by construction.  See tn_mlp.hip's header
// for the dataflow.
#pragma once
#include "tn_mlp_common.h"

namespace tn {
namespace mlp {

// TRAIN: the layer inputs x0, h1..h4 (quad-major [F/4][n][4], what the weight-gradient GEMMs contract) and the ReLU masks (all the
// dX kernel needs) are saved on the way -- the backward pass recomputes nothing (round 3a recomputed the whole forward
// inside the dX kernel: 2.2 of its 5 ms).
struct FwdSave { float *x0, *h1, *h2, *h3, *h4; unsigned long long *masks; };

// group g = samples [g * GROUP, (g + 1) * GROUP) of n; every thread of the block calls it (it contains the block barriers of
// the weight stages).  TRAIN: cy carries h4 and the place of its mask from one group of the block to the next (FwdCarry,
// tn_mlp_common.h: cy->p == nullptr before the first group; the caller stores the last group's with flush_carry).  lds: MAX_STAGE_FLOATS floats.  hterm [rays][128]: the head layer's per-ray term (unused when DENSITY_ONLY).
template <bool GATHER, bool DENSITY_ONLY, int BLOCK, bool TRAIN>
static __device__ __forceinline__ void mlp_forward_group(float *lds, size_t g, size_t n, uint32_t samples_per_ray,
                                                         const float *__restrict__ feats, const uint32_t *__restrict__ vi,
                                                         const float *__restrict__ bc, const float *__restrict__ fieldT,
                                                         const float *__restrict__ hterm, const float *__restrict__ pk,
                                                         float *__restrict__ sigma, float *__restrict__ rgb, const FwdSave &sv,
                                                         FwdCarry *cy = nullptr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5;
    constexpr size_t GROUP = (BLOCK / 64) * 32;
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
        if constexpr (TRAIN) {
            // (the first group of a block has nothing to carry: it stores zeros where its own h4 and mask go -- the same lane
            //  stores the values there later, and a wave's stores to one address land in program order)
            if (!cy->p) { cy->p = quad_ptr(sv.h4, n, sc, h); cy->m = sv.masks + ((size_t)3 * n + sc) * 2 + h; }
            gemm_steps_store_carry<KS1, OT>(acc, bin, lds, lane, quad_ptr_x0(sv.x0, n, sc, h), n, *cy, 2 * n);
        } else gemm_steps<KS1, 0, OT>(acc, bin, lds, lane);
        bias_step<KS1, OT>(acc, lds, lane);
        relu_to_bin(acc, bin);
    }
    auto mask_ptr = [&](int layer) { return sv.masks + ((size_t)layer * n + sc) * 2 + h; };   // TRAIN only
    // ---- layers 2, 3: 128 -> 128, accumulators fed back as B operands
    __syncthreads();
    stage_weights<BLOCK>(lds, pk + OFF_W2, lfloats(KSH, OT));
    stage_wait();
    {
        f32x16 acc[OT];
        zero_acc(acc);
        if constexpr (TRAIN) gemm_steps_store<KSH, 0, OT, KSH, true>(acc, bin, lds, lane, quad_ptr(sv.h1, n, sc, h), 2 * n, mask_ptr(0));
        else gemm_steps<KSH, 0, OT>(acc, bin, lds, lane);
        bias_step<KSH, OT>(acc, lds, lane);
        relu_to_bin(acc, bin);
    }
    __syncthreads();
    stage_weights<BLOCK>(lds, pk + OFF_W3, N_W3);
    stage_wait();
    {
        f32x16 acc[OT];
        zero_acc(acc);
        if constexpr (TRAIN) gemm_steps_store<KSH, 0, OT, KSH, true>(acc, bin, lds, lane, quad_ptr(sv.h2, n, sc, h), 2 * n, mask_ptr(1));
        else gemm_steps<KSH, 0, OT>(acc, bin, lds, lane);
        bias_step<KSH, OT>(acc, lds, lane);
        relu_to_bin(acc, bin);  // mlp_base out_activation = ReLU
    }
    {
        // density head 128 -> 1 + softplus on the VALU (the vector rides behind layer 3's weights)
        const float *dv = lds + lfloats(KSH, OT);
        const float raw = head_dot(dv + 64 * h, bin) + dv[128];
        const float sp = raw > 20.0f ? raw : log1pf(expf(raw));  // torch softplus(beta=1, threshold=20)
        if (h == 0 && s < n) sigma[s] = sp;
    }
    if constexpr (DENSITY_ONLY) return;  // coarse pass of the model (model.py:577-581)
    // ---- head [enc(27) | base(128)] -> 128 ReLU: the 128 base columns as a GEMM, the encoding's 27 columns (constant along a
    //      ray) as the per-ray vector the caller made (hterm = Wh[:, :27] enc(dir) + the appearance embedding's bias, if any)
    __syncthreads();
    stage_weights<BLOCK>(lds, pk + OFF_WHEAD, N_WHEAD);
    stage_wait();
    {
        f32x16 acc[OT];
        zero_acc(acc);
        if constexpr (TRAIN) gemm_steps_store<KSH, 0, OT, KSH, true>(acc, bin, lds, lane, quad_ptr(sv.h3, n, sc, h), 2 * n, mask_ptr(2));
        else gemm_steps<KSH, 0, OT>(acc, bin, lds, lane);
        bias_step<HEAD_KS, OT>(acc, lds, lane);
        add_ray_bias(acc, hterm + (sc / samples_per_ray) * HID, h);
        relu_to_bin(acc, bin);
    }
    if constexpr (TRAIN) {   // h4 and its mask leave under the next group's first GEMM (or with flush_carry)
#pragma unroll
        for (int j = 0; j < KSH; ++j) cy->h4[j] = bin[j];
        cy->p = quad_ptr(sv.h4, n, sc, h);
        cy->m = mask_ptr(3);
    }
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

// after a block's last group: the carried h4 and its mask
static __device__ __forceinline__ void flush_carry(const FwdCarry &cy, size_t n) {
    if (!cy.p) return;
    float4 *p = cy.p;
#pragma unroll
    for (int g = 0; g < KSH / 4; ++g) {
        *p = make_float4(cy.h4[4 * g], cy.h4[4 * g + 1], cy.h4[4 * g + 2], cy.h4[4 * g + 3]);
        p += 2 * n;
    }
    *cy.m = mask_of(cy.h4);
}

}  // namespace mlp
}  // namespace tn
