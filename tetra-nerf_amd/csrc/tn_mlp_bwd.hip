// tn_mlp_bwd.hip -- training adjoint of the shallow MLP + heads on fp32 MFMA.
//
// Replaces what PyTorch autograd does for the reference's model in training
// (tetranerf/nerfstudio/model.py:602-630: mlp_base, density / rgb heads, mlp_head; the trainer back-propagates
// through them with cuBLAS GEMMs and elementwise kernels).  Two kernels:
//
//   k_mlp_backward  "dX chain".  Same dataflow as the forward kernel (tn_mlp.hip): one wavefront owns 32 samples,
//       activations live in the MFMA accumulators in the layout the next layer's B operand wants.  It first
//       RECOMPUTES the forward pass (gather -> 4 layers; nothing was saved by the forward), keeping of every hidden
//       layer only the ReLU mask (64 bits per lane) and writing the activation itself to HBM, then runs the
//       reverse network on the matrix cores: the transposed weights W^T are packed with their K axis (the layer's
//       OUTPUT features) in accumulator order, so d_pre of layer l+1 is fed straight back as the B operand of
//       W_{l+1}^T exactly like an activation in the forward pass.  softplus' / sigmoid' / ReLU masks are applied
//       in registers; the two narrow heads run on the VALU as in the forward kernel.
//       Outputs, all FEATURE-MAJOR [F, n] (a register of the wave = one feature of 32 consecutive samples = one
//       128-byte line): the layer inputs x0, h1, h2, h3, h4 and the pre-activation gradients d_pre1..4 (for the
//       weight gradients), d_sigma_raw / d_rgb_raw, and d_x0 [64, n] = the gradient of the gathered features,
//       which tn_interpolate_values_backward scatters into the field.
//   k_dw_gemm  weight gradients dW[out, in] += A[out, n] * B[in, n]^T with K = the sample axis streamed once from
//       HBM (both operands are read as whole lines), 32x32x2 fp32 MFMA tiles, per-block partial sums added with
//       float atomics; the bias gradient (row sums of A) rides along.  At 128 x 128 it needs 1 KB of operands per
//       131 kFLOP: balanced between HBM and the fp32 MFMA peak.
//
// FLOPs per fine sample: forward 122.6 k, recompute 122.6 k - heads, dX 114.7 k, dW 122.4 k.
#include "tn_mlp_common.h"

namespace tn {

using namespace mlp;

namespace {

// ---- transposed packs: [k-step over the layer's OUTPUT features (accumulator order)][tile of INPUT features][lane]
constexpr int OTI1 = FD / 32;                                   // input tiles of layer 1
constexpr size_t tfloats(int tiles) { return (size_t)KSH * tiles * 64; }
constexpr size_t OFFT_H = 0;                                    // Wh[:, 27:]^T  (128 -> 128), density vector behind it
constexpr size_t N_TH = tfloats(OT) + DVEC;
constexpr size_t OFFT_3 = OFFT_H + N_TH;
constexpr size_t OFFT_2 = OFFT_3 + tfloats(OT);
constexpr size_t OFFT_1 = OFFT_2 + tfloats(OT);
constexpr size_t PACKT_FLOATS = OFFT_1 + tfloats(OTI1);
constexpr size_t PACK_SLACK = 1024;   // floats behind the forward pack that a whole-pass stage copy may read (tn_mlp.hip allocates them)

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
        else { const int jj = (int)(j - tfloats(OT)); if (jj < 128) v = w.wd[acc_k(jj & 63, jj >> 6)]; }
    } else if (i < OFFT_2) {
        at(i - OFFT_3, OT, ks, t, row, h); v = w.w3[(size_t)acc_k(ks, h) * HID + 32 * t + row];
    } else if (i < OFFT_1) {
        at(i - OFFT_2, OT, ks, t, row, h); v = w.w2[(size_t)acc_k(ks, h) * HID + 32 * t + row];
    } else {
        at(i - OFFT_1, OTI1, ks, t, row, h); v = w.w1[(size_t)acc_k(ks, h) * FD + 32 * t + row];
    }
    pt[i] = v;
}

struct BwdBuffers {
    float *x0;                 // [64, n]  gathered features
    float *h1, *h2, *h3, *h4;  // [128, n] layer outputs after ReLU
    float *d1, *d2, *d3, *d4;  // [128, n] gradients w.r.t. the pre-activations of layers 1, 2, 3 and the head layer
    float *dhead;              // [4, n]   d sigma_raw, d rgb_raw[0..2]
    float *dx0;                // [n, 64]  gradient of the gathered features, SAMPLE-major rows (what the gather adjoint reads)
};

// bin slot j of half-wave h holds feature acc_k(j, h) = 32 (j >> 4) + (j & 3) + 8 ((j >> 2) & 3) + 4 h: from slot to slot
// the feature grows by 1, or by 5 after every fourth slot -- the feature-major stores walk one pointer with two strides
// (64 independent row addresses would be hoisted out of the sample loop and spill).
template <int COUNT>
__device__ __forceinline__ void store_slots(float *__restrict__ dst, size_t n, size_t s, bool ok, const float (&vals)[COUNT], int h) {
    if (!ok) return;
    float *p = dst + (size_t)(4 * h) * n + s;
    const size_t n1 = n, n5 = 5 * n;
#pragma unroll
    for (int j = 0; j < COUNT; ++j) {
        *p = vals[j];
        p += ((j & 3) == 3) ? n5 : n1;
    }
}
__device__ __forceinline__ void store_bin(float *__restrict__ dst, size_t n, size_t s, bool ok, const float (&bin)[KSH], int h) {
    store_slots<KSH>(dst, n, s, ok, bin, h);
}

__device__ __forceinline__ unsigned long long mask_of(const float (&bin)[KSH]) {
    unsigned long long m = 0;
#pragma unroll
    for (int j = 0; j < KSH; ++j) m |= (unsigned long long)(bin[j] > 0.f ? 1u : 0u) << j;
    return m;
}

template <int TILES>
__device__ __forceinline__ void masked_to_bin(const f32x16 (&acc)[TILES], unsigned long long m, float (&bin)[KSH]) {
#pragma unroll
    for (int t = 0; t < OT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) bin[t * 16 + r] = ((m >> (t * 16 + r)) & 1ull) ? acc[t][r] : 0.f;
}

// 4 waves per block = one per SIMD: the kernel keeps 64 activations, 64 accumulators, four 64-bit ReLU masks and the head
// gradients live at once, more than the 256 registers a wave gets at two waves per SIMD (the forward kernel's shape);
// alone on its SIMD a wave has the whole 512-entry file (VGPRs + AGPRs), and one wave per SIMD already reaches the
// fp32 MFMA issue rate -- PROVIDED nothing it waits for is on the critical path, because no other wave fills the gap:
//   * the eight weight stages of a group ping-pong between two LDS buffers: stage l + 1 is requested (async global -> LDS)
//     when GEMM l starts and has the whole GEMM to land; ONE barrier per layer (it says both "everybody is done with the
//     buffer about to be overwritten" and "everybody's share of this layer's weights has landed");
//   * the feature-major stores of a GEMM's input (x0, h1..h3, d4..d1: 64 x 128-byte lines per wave and tensor) are issued
//     between the MFMAs of the first 32 k-steps of the GEMM that consumes it, so that they have drained long before the
//     next `s_waitcnt vmcnt(0)` (on gfx9 stores and loads share that counter: round 3a's kernel waited for the 64 stores of
//     every layer before it could even request the next layer's weights);
//   * the vertex ids / weights of the NEXT group are requested half a group ahead.
// (round 3a, profiles/r03m_train_kernel_stats.txt: 5.5 ms per 2.1 M samples = 57 % of the MFMA-bound time.)
constexpr int BWD_BLOCK = 256;
constexpr int passes(size_t floats) { return (int)((floats + 1023) / 1024); }   // stage copies: whole 4 KB passes
constexpr int cmax(int a, int b) { return a > b ? a : b; }
constexpr int P_W1 = passes(lfloats(KS1, OT)), P_W2 = passes(lfloats(KSH, OT)), P_W3 = passes(N_W3), P_WH = passes(N_WHEAD);
constexpr int P_TH = passes(N_TH), P_T = passes(tfloats(OT)), P_T1 = passes(tfloats(OTI1));
constexpr size_t BUF_A = 1024 * (size_t)cmax(cmax(P_W1, P_W3), cmax(P_TH, P_T));    // stages 0, 2, 4, 6
constexpr size_t BUF_B = 1024 * (size_t)cmax(cmax(P_W2, P_WH), cmax(P_T, P_T1));    // stages 1, 3, 5, 7
static_assert((BUF_A + BUF_B) * sizeof(float) <= 160 * 1024, "two weight stages must fit the CU's LDS");
static_assert(OFF_WHEAD + 1024 * (size_t)P_WH <= PACK_FLOATS + PACK_SLACK, "the last stage copy over-reads into the pack's slack");
static_assert(OFFT_1 + 1024 * (size_t)P_T1 <= PACKT_FLOATS, "transposed pack");
static_assert(tfloats(OTI1) + (BWD_BLOCK / 64) * 32 * 65 <= BUF_B, "d x0 transposition behind the last stage");

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

// gemm_steps (tn_mlp_common.h) + the feature-major stores of the B operand: two values after each of the first NST / 2
// k-steps.  LINEAR: consecutive values are n floats apart (x0); otherwise the accumulator order of store_slots.
template <int KS, int KS0, int TILES, int NST, bool LINEAR>
__device__ __forceinline__ void gemm_steps_store(f32x16 (&acc)[TILES], const float (&bin)[KSH], const float *lds, int lane,
                                                 float *__restrict__ p, size_t n) {
    const size_t n5 = 5 * n;
    float a[TILES], an[TILES];
    const float *w0 = lds + (size_t)KS0 * TILES * 64 + lane;
#pragma unroll
    for (int t = 0; t < TILES; ++t) a[t] = w0[t * 64];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        if (ks + 1 < KS) {
            const float *wrow = lds + (size_t)(KS0 + ks + 1) * TILES * 64 + lane;
#pragma unroll
            for (int t = 0; t < TILES; ++t) an[t] = wrow[t * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < TILES; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], bin[ks], acc[t], 0, 0, 0);
        if (2 * ks < NST) {
#pragma unroll
            for (int j = 2 * ks; j < 2 * ks + 2; ++j) {
                *p = bin[j];
                p += (LINEAR || (j & 3) != 3) ? n : n5;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < TILES; ++t) a[t] = an[t];
    }
}

}  // namespace

__global__ __launch_bounds__(BWD_BLOCK) void k_mlp_backward(size_t n, uint32_t samples_per_ray, const uint32_t *__restrict__ vi,
                                                            const float *__restrict__ bc, const float *__restrict__ fieldT,
                                                            const float *__restrict__ enc, const float *__restrict__ pk,
                                                            const float *__restrict__ pt, const float *__restrict__ d_sigma,
                                                            const float *__restrict__ d_rgb, BwdBuffers o) {
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

    stage_fixed<P_W1>(bufA, pk + OFF_W1);
    size_t sc = sample_of(blockIdx.x);
    uint4 v4 = *reinterpret_cast<const uint4 *>(vi + 4 * sc);
    float b0 = bc[3 * sc], b1 = bc[3 * sc + 1], b2 = bc[3 * sc + 2];

    for (size_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
        float bin[KSH];
        unsigned long long m1, m2, m3, m4;
        f32x16 acc[OT];

        // ================= forward recompute =================
        // ---- fused barycentric gather (same summation order as interpolate_values) -> x0
        {
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
        // per-sample inputs of the later layers, requested now
        const float dsg = d_sigma[sc];
        const float drg0 = d_rgb[3 * sc], drg1 = d_rgb[3 * sc + 1], drg2 = d_rgb[3 * sc + 2];
        float ev[KSE];
        {
            const float *e = enc + (sc / samples_per_ray) * ENC_PAD;
#pragma unroll
            for (int ks = 0; ks < KSE; ++ks) ev[ks] = e[2 * ks + h];
        }
        // ---- layer 1 (weights: buffer A)
        layer_top();
        stage_fixed<P_W2>(bufB, pk + OFF_W2);
        zero_acc(acc);
        gemm_steps_store<KS1, 0, OT, KS1, true>(acc, bin, bufA, lane, o.x0 + (size_t)(32 * h) * n + sc, n);
        bias_step<KS1, OT>(acc, bufA, lane);
        relu_to_bin(acc, bin);
        m1 = mask_of(bin);
        // ---- layer 2 (B)
        layer_top();
        stage_fixed<P_W3>(bufA, pk + OFF_W3);
        zero_acc(acc);
        gemm_steps_store<KSH, 0, OT, KSH, false>(acc, bin, bufB, lane, o.h1 + (size_t)(4 * h) * n + sc, n);
        bias_step<KSH, OT>(acc, bufB, lane);
        relu_to_bin(acc, bin);
        m2 = mask_of(bin);
        // ---- layer 3 (A) + density head
        layer_top();
        stage_fixed<P_WH>(bufB, pk + OFF_WHEAD);
        zero_acc(acc);
        gemm_steps_store<KSH, 0, OT, KSH, false>(acc, bin, bufA, lane, o.h2 + (size_t)(4 * h) * n + sc, n);
        bias_step<KSH, OT>(acc, bufA, lane);
        relu_to_bin(acc, bin);
        m3 = mask_of(bin);
        float dsr;  // d L / d sigma_raw
        {
            const float *dv = bufA + lfloats(KSH, OT);
            const float raw = head_dot(dv + 64 * h, bin) + dv[128];
            // softplus(beta = 1, threshold = 20): derivative sigmoid(raw), 1 beyond the threshold
            const float ds = raw > 20.0f ? 1.0f : 1.0f / (1.0f + expf(-raw));
            dsr = dsg * ds;
        }
        // ---- head [enc(27) | base(128)] -> 128 ReLU (B), rgb head
        layer_top();
        stage_fixed<P_TH>(bufA, pt + OFFT_H);
        zero_acc(acc);
#pragma unroll
        for (int ks = 0; ks < KSE; ++ks) {
            const float *wrow = bufB + (size_t)ks * OT * 64 + lane;
#pragma unroll
            for (int t = 0; t < OT; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wrow[t * 64], ev[ks], acc[t], 0, 0, 0);
        }
        gemm_steps_store<KSH, KSE, OT, KSH, false>(acc, bin, bufB, lane, o.h3 + (size_t)(4 * h) * n + sc, n);
        bias_step<HEAD_KS, OT>(acc, bufB, lane);
        relu_to_bin(acc, bin);
        m4 = mask_of(bin);
        // ================= backward =================
        float d4v[KSH];
        {
            // rgb head: rgb = sigmoid(c), d c = d rgb * rgb * (1 - rgb); d h4 = Wr^T d c, masked by ReLU'(h4)
            const float *cv = bufB + lfloats(HEAD_KS, OT);
            const float drg[3] = {drg0, drg1, drg2};
            float drr[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float raw = head_dot(cv + 128 * c + 64 * h, bin) + cv[384 + c];
                const float y = 1.0f / (1.0f + expf(-raw));
                drr[c] = drg[c] * (y * (1.0f - y));
            }
            if (h == 0) {
                o.dhead[sc] = dsr;
                o.dhead[n + sc] = drr[0]; o.dhead[2 * n + sc] = drr[1]; o.dhead[3 * n + sc] = drr[2];
            }
            const float *w0 = cv + 64 * h, *w1 = cv + 128 + 64 * h, *w2 = cv + 256 + 64 * h;
#pragma unroll
            for (int j = 0; j < KSH; ++j) {
                const float v = (w0[j] * drr[0] + w1[j] * drr[1]) + w2[j] * drr[2];
                d4v[j] = ((m4 >> j) & 1ull) ? v : 0.f;
            }
        }
        // ---- d h3 = Wh[:, 27:]^T d_pre4 + wd * d sigma_raw, masked (A).  h4 leaves only now: its stores have this GEMM to drain
        layer_top();
        stage_fixed<P_T>(bufB, pt + OFFT_3);
        store_bin(o.h4, n, sc, true, bin, h);
#pragma unroll
        for (int j = 0; j < KSH; ++j) bin[j] = d4v[j];
        // the next group's sample descriptors
        const size_t scn = sample_of(g + gridDim.x < ngroups ? g + gridDim.x : g);
        const uint4 v4n = *reinterpret_cast<const uint4 *>(vi + 4 * scn);
        const float b0n = bc[3 * scn], b1n = bc[3 * scn + 1], b2n = bc[3 * scn + 2];
        zero_acc(acc);
        gemm_steps_store<KSH, 0, OT, KSH, false>(acc, bin, bufA, lane, o.d4 + (size_t)(4 * h) * n + sc, n);
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
        gemm_steps_store<KSH, 0, OT, KSH, false>(acc, bin, bufB, lane, o.d3 + (size_t)(4 * h) * n + sc, n);
        masked_to_bin(acc, m2, bin);
        layer_top();
        stage_fixed<P_T1>(bufB, pt + OFFT_1);
        zero_acc(acc);
        gemm_steps_store<KSH, 0, OT, KSH, false>(acc, bin, bufA, lane, o.d2 + (size_t)(4 * h) * n + sc, n);
        masked_to_bin(acc, m1, bin);
        // ---- d x0 = W1^T d_pre1  (64 input features = 2 tiles) (B); the next group's first layer goes to A meanwhile
        layer_top();
        stage_fixed<P_W1>(bufA, pk + OFF_W1);
        {
            f32x16 acc2[OTI1];
            zero_acc(acc2);
            gemm_steps_store<KSH, 0, OTI1, KSH, false>(acc2, bin, bufB, lane, o.d1 + (size_t)(4 * h) * n + sc, n);
            // d x0 leaves as SAMPLE-major rows [n, 64] (the gather adjoint reads a sample's gradient as one 256-byte line;
            // round 3a wrote it feature-major and transposed 0.5 GB per iteration): through this wave's slice of the free
            // tail of buffer B ([32 samples][65]: conflict-free both ways), each sample's 64 values then go out as one
            // coalesced store
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
        }
        sc = scn; v4 = v4n; b0 = b0n; b1 = b1n; b2 = b2n;
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
                                                           float background, const float *__restrict__ g_rgb,
                                                           const float *__restrict__ g_acc, float *__restrict__ d_sigma,
                                                           float *__restrict__ d_rgb) {
    const int lane = threadIdx.x;
    for (size_t ray = blockIdx.x; ray < R; ray += gridDim.x) {
        const float *e = edges + ray * (S + 1);
        const float gr = g_rgb ? g_rgb[3 * ray] : 0.f, gg = g_rgb ? g_rgb[3 * ray + 1] : 0.f, gb = g_rgb ? g_rgb[3 * ray + 2] : 0.f;
        const float ga = g_acc ? g_acc[ray] : 0.f;
        const float a_const = ga - background * ((gr + gg) + gb);
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

void launch_composite_backward(size_t R, uint32_t S, const float *sigma, const float *rgb, const float *edges, float background,
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

void launch_mlp_backward(size_t n, uint32_t samples_per_ray, const uint32_t *vi, const float *bc, const float *field_vm,
                         const float *dirs, const MlpPacks &w, const float *d_sigma, const float *d_rgb,
                         const MlpBackwardBuffers &b, hipStream_t stream) {
    if (n == 0) return;
    const size_t num_rays = n / samples_per_ray;
    const float *pk = w.pk_gather, *pt = w.pt;
    float *enc = w.enc;
    launch_dir_encoding(num_rays, dirs, enc, stream);
    const size_t smem = (BUF_A + BUF_B) * sizeof(float);
    static PerDeviceOnce lds_attr;
    lds_attr.run([&] { allow_dynamic_lds(reinterpret_cast<const void *>(k_mlp_backward), smem); });
    const size_t group = (BWD_BLOCK / 64) * 32;
    const size_t ngroups = (n + group - 1) / group;
    const unsigned grid = (unsigned)(ngroups < 256 ? ngroups : 256);  // one 4-wave block per CU
    BwdBuffers o{b.x0, b.h1, b.h2, b.h3, b.h4, b.d1, b.d2, b.d3, b.d4, b.dhead, b.dx0};
    hipLaunchKernelGGL(k_mlp_backward, dim3(grid), dim3(BWD_BLOCK), smem, stream, n, samples_per_ray, vi, bc, field_vm, enc, pk, pt,
                       d_sigma, d_rgb, o);
}

}  // namespace tn
