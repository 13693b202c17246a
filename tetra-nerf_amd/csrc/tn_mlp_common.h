// tn_mlp_common.h -- geometry of the packed MLP layers and the device helpers shared by the forward kernels
// (tn_mlp.hip) and the training kernels (tn_mlp_bwd.hip).  See the header of tn_mlp.hip for the dataflow.
#pragma once
#include "tn_device.h"
#include "tn_kernels.h"

namespace tn {
namespace mlp {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int HID = 128;          // hidden width
constexpr int FD = 64;            // field dim
constexpr int ENC = 27;           // direction encoding width
constexpr int ENC_PAD = 28;       // padded to an even K
constexpr int KS1 = FD / 2;       // k-steps of layer 1
constexpr int KSH = HID / 2;      // k-steps over 128 features held in accumulators
constexpr int KSE = ENC_PAD / 2;  // k-steps over the direction encoding
constexpr int OT = HID / 32;      // output tiles of a hidden layer
constexpr int MLP_BLOCK = 512;    // 8 waves share one staged layer: 256 samples per group

// One staged layer = [k-steps + 1][tiles][64 lanes] floats; the extra last k-step carries the bias
// (A = bias for the lower half-wave, 0 for the upper; B = 1.0), so the bias add is part of the GEMM.
// The head layer has a 5th output tile whose row 0 is the density head (it consumes the same B
// operands, the mlp_base output); the rgb head is a 1-tile layer (rows 0..2).
struct LayerGeom { int ksteps, tiles; size_t off; };
constexpr size_t lfloats(int ksteps, int tiles) { return (size_t)(ksteps + 1) * tiles * 64; }
// The head layer (mlp_head: [enc(27) | base(128)] -> 128) as an MFMA GEMM covers only its 128 base columns (round 5): the
// direction encoding is constant along a ray, so its 27 columns collapse to ONE vector per ray, t_ray = Wh[:, :27] enc(dir)
// -- 27 multiply-adds per output and ray instead of 14 MFMA k-steps (of 242 in the network) per 32 samples -- which is
// added to the accumulators together with the appearance embedding's per-ray bias (add_ray_bias).  head_ray_term below is
// the one expression both the stand-alone kernel (k_head_ray_term, tn_mlp.hip) and the persistent render kernel evaluate.
constexpr int HEAD_KS = KSH;
// The two narrow heads (density 128 -> 1 on the mlp_base output, rgb 128 -> 3 on the head output) are NOT
// MFMA layers: as 32-row tiles they would spend 130 of 1098 MFMAs per 32 samples on 4 useful rows.  Their
// weights ride behind the layer that produces their input ([half][64] floats in the lane's K order + bias)
// and each lane does its half of the dot products on the otherwise idle VALU (64 fma per output).
constexpr size_t DVEC = 2 * 64 + 4;        // density: wd in K order per half, bd, pad
constexpr size_t CVEC = 3 * 2 * 64 + 4;    // rgb: wr rows in K order per half, br, pad
constexpr size_t OFF_W1 = 0;
constexpr size_t OFF_W2 = OFF_W1 + lfloats(KS1, OT);
constexpr size_t OFF_W3 = OFF_W2 + lfloats(KSH, OT);
constexpr size_t N_W3 = lfloats(KSH, OT) + DVEC;
constexpr size_t OFF_WHEAD = OFF_W3 + N_W3;
constexpr size_t N_WHEAD = lfloats(HEAD_KS, OT) + CVEC;
constexpr size_t PACK_FLOATS = OFF_WHEAD + N_WHEAD;
constexpr size_t MAX_STAGE_FLOATS = N_WHEAD;

__host__ __device__ constexpr int acc_feature(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
// k index consumed by k-step `ks` (0..63) of a layer whose input lives in accumulators, half h
__host__ __device__ constexpr int acc_k(int ks, int h) { return 32 * (ks >> 4) + acc_feature(ks & 15, h); }

// Packed layer -> LDS with the async global->LDS path (global_load_lds_dwordx4: no staging registers, all
// of a thread's loads in flight at once; a load-wait-write loop exposes one L2 latency per 8 KB).  The LDS
// destination of a wave is uniform base + lane * 16, which is exactly a linear copy.
template <int BLOCK = MLP_BLOCK>
static __device__ __forceinline__ void stage_weights(float *lds, const float *__restrict__ src, size_t n_floats) {
    const float4 *s4 = reinterpret_cast<const float4 *>(src);
    float4 *d4 = reinterpret_cast<float4 *>(lds);
    const uint32_t n16 = (uint32_t)(n_floats / 4), lane = threadIdx.x & 63;
    const uint32_t wave0 = __builtin_amdgcn_readfirstlane(threadIdx.x & ~63u);
    for (uint32_t base = wave0; base < n16; base += BLOCK) {
        const uint32_t i = base + lane;
        if (i < n16)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(s4 + i),
                                             (__attribute__((address_space(3))) void *)(d4 + base), 16, 0, 0);
    }
}
static __device__ __forceinline__ void stage_wait() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// acc[t] += W_staged[k-steps KS0 .. KS0+KS) * bin[BIN0 .. BIN0+KS)
// The A operands (staged weights) of k-step ks+1 are requested from LDS BEFORE the MFMAs of k-step ks are issued: left to
// itself the compiler reads them into the same registers right before their use, and a wave that is alone on its SIMD
// (the training kernel) then idles for one LDS latency per k-step (round 3: the dX kernel ran at 52 % of the MFMA rate).
// The sched_barrier keeps the scheduler from hoisting hundreds of reads (register blow-up) or sinking these.
template <int KS, int KS0, int TILES, int BIN0 = 0>
static __device__ __forceinline__ void gemm_steps(f32x16 (&acc)[TILES], const float (&bin)[KSH], const float *lds, int lane) {
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
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], bin[BIN0 + ks], acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < TILES; ++t) a[t] = an[t];
    }
}

// Layout of the training buffers ("quad-major"): [F / 4][n][4] floats -- element (feature f, sample s) at
// ((f / 4) n + s) 4 + f % 4.  A lane holds its sample's features in groups of four consecutive ones (slots 4g .. 4g + 3 of
// half-wave h = features 32 (g >> 2) + 8 (g & 3) + 4 h + {0..3} = quad 8 (g >> 2) + 2 (g & 3) + h), so a group leaves as ONE
// 16-byte store per lane, 512 contiguous bytes per half-wave (feature-major [F][n] needed four 4-byte stores: the store
// instructions themselves, not the bytes, were what saving cost the training kernels), and the weight-gradient GEMMs
// fetch their tiles as 16-byte loads.
//
// gemm_steps + the stores of the B operand (the training kernels save every GEMM's input): one quad after every second
// k-step of the first NST / 2, i.e. in the shadow of MFMAs and drained long before the GEMM ends.  p: this lane's first
// quad; qstride: distance of consecutive quads in float4 units (2 n in accumulator order, n for x0).
// MASK: the operand is a ReLU output whose mask the dX kernel wants (slot j: bit j, stored as masks[(layer * n + sample) * 2 +
// half]); its two VALU operations per value ride under the MFMAs of k-step j as well (round 5's ablation, profiles/
// r05m_save_ablate.txt: computed between two GEMMs, where the matrix pipe of BOTH waves of a SIMD idles -- the stage
// barriers keep them in phase -- the four masks cost 0.10 of the 0.30 ms the saves add to a 2.1 M-sample forward).
static __device__ __forceinline__ uint32_t relu_bit(float v, int j) {
    const uint32_t b = __float_as_uint(v);      // a ReLU output is >= +0: "positive" = "bit pattern not zero"
    return (b < 1u ? b : 1u) << (j & 31);
}
template <int KS, int KS0, int TILES, int NST, bool MASK = false>
static __device__ __forceinline__ void gemm_steps_store(f32x16 (&acc)[TILES], const float (&bin)[KSH], const float *lds, int lane,
                                                        float4 *__restrict__ p, size_t qstride,
                                                        unsigned long long *__restrict__ mask_out = nullptr) {
    float a[TILES], an[TILES];
    uint32_t lo = 0, hi = 0;
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
        if ((ks & 1) == 0 && 2 * ks < NST) {
            *p = make_float4(bin[2 * ks], bin[2 * ks + 1], bin[2 * ks + 2], bin[2 * ks + 3]);
            p += qstride;
        }
        if constexpr (MASK) {
            if (ks < 32) lo |= relu_bit(bin[ks], ks);
            else hi |= relu_bit(bin[ks], ks);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < TILES; ++t) a[t] = an[t];
    }
    if constexpr (MASK) *mask_out = ((unsigned long long)hi << 32) | lo;
}
// The training forward's LAST saved tensor, h4 (the head layer's ReLU output), and its mask have no GEMM of their own group
// to leave under (the rgb head runs on the VALU): stored right away they cost as much as h1..h3 together (0.09 ms of the same
// 0.30).  They are carried in registers into the NEXT group's layer-1 GEMM instead, whose 32 k-steps have room beside the 8 quads
// of x0: one quad of the carried h4 after every odd k-step, two of its mask bits per k-step.
struct FwdCarry { float h4[KSH]; float4 *p; unsigned long long *m; };
template <int KS, int TILES>
static __device__ __forceinline__ void gemm_steps_store_carry(f32x16 (&acc)[TILES], const float (&bin)[KSH], const float *lds, int lane,
                                                              float4 *__restrict__ p, size_t qstride, const FwdCarry &cy, size_t cstride) {
    static_assert(KS == 32 && KSH == 64, "16 carried quads on the odd k-steps, two mask bits per k-step");
    float a[TILES], an[TILES];
    uint32_t lo = 0, hi = 0;
    float4 *cp = cy.p;
    const float *w0 = lds + lane;
#pragma unroll
    for (int t = 0; t < TILES; ++t) a[t] = w0[t * 64];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        if (ks + 1 < KS) {
            const float *wrow = lds + (size_t)(ks + 1) * TILES * 64 + lane;
#pragma unroll
            for (int t = 0; t < TILES; ++t) an[t] = wrow[t * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < TILES; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], bin[ks], acc[t], 0, 0, 0);
        if ((ks & 1) == 0 && 2 * ks < KS) {
            *p = make_float4(bin[2 * ks], bin[2 * ks + 1], bin[2 * ks + 2], bin[2 * ks + 3]);
            p += qstride;
        }
        if (ks & 1) {
            const int q = ks >> 1;
            *cp = make_float4(cy.h4[4 * q], cy.h4[4 * q + 1], cy.h4[4 * q + 2], cy.h4[4 * q + 3]);
            cp += cstride;
        }
        if (ks < 16) lo |= relu_bit(cy.h4[2 * ks], 2 * ks) | relu_bit(cy.h4[2 * ks + 1], 2 * ks + 1);
        else hi |= relu_bit(cy.h4[2 * ks], 2 * ks) | relu_bit(cy.h4[2 * ks + 1], 2 * ks + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < TILES; ++t) a[t] = an[t];
    }
    *cy.m = ((unsigned long long)hi << 32) | lo;
}
// this lane's first quad of a [128 / 4][n][4] tensor in accumulator order / of the [64 / 4][n][4] gathered features
static __device__ __forceinline__ float4 *quad_ptr(float *base, size_t n, size_t s, int h) {
    return reinterpret_cast<float4 *>(base) + ((size_t)h * n + s);
}
static __device__ __forceinline__ float4 *quad_ptr_x0(float *base, size_t n, size_t s, int h) {
    return reinterpret_cast<float4 *>(base) + ((size_t)(8 * h) * n + s);
}

template <int STEP, int TILES>
static __device__ __forceinline__ void bias_step(f32x16 (&acc)[TILES], const float *lds, int lane) {
    const float *wrow = lds + (size_t)STEP * TILES * 64 + lane;
#pragma unroll
    for (int t = 0; t < TILES; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wrow[t * 64], 1.0f, acc[t], 0, 0, 0);
}

// Per-RAY bias of the head layer (the appearance embedding of the reference's model, model.py:437-447,608-620: the head
// input is cat[encoded_dir, base, embedded_appearance] and the embedding is constant along a ray, so its E columns of the
// head GEMM collapse to one vector per ray, c = Wh[:, 155:] emb(camera of the ray), which the caller computes -- an
// [rays, E] x [E, 128] product -- and the kernels add to the head pre-activation before the ReLU).  row = the ray's 128
// floats; a lane adds the 64 features it holds: acc[t][4 q + j] is feature 32 t + 8 q + 4 h + j (acc_feature).
template <int TILES>
static __device__ __forceinline__ void add_ray_bias(f32x16 (&acc)[TILES], const float *__restrict__ row, int h) {
#pragma unroll
    for (int t = 0; t < OT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = *reinterpret_cast<const float4 *>(row + 32 * t + 8 * q + 4 * h);
            acc[t][4 * q] += v.x; acc[t][4 * q + 1] += v.y; acc[t][4 * q + 2] += v.z; acc[t][4 * q + 3] += v.w;
        }
}

// t[o] = sum_k wenc[o][k] enc[k], k = 0 .. 26 in this order (wenc: [128][ENC_PAD] = Wh[:, :27], column 27 zero)
static __device__ __forceinline__ float head_ray_term(const float *__restrict__ wenc_row, const float *enc) {
    float a = 0.f;
#pragma unroll
    for (int k = 0; k < ENC; ++k) a = a + wenc_row[k] * enc[k];
    return a;
}

template <int TILES>
static __device__ __forceinline__ void zero_acc(f32x16 (&acc)[TILES]) {
#pragma unroll
    for (int t = 0; t < TILES; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
}

// This lane's half of a 128-long dot product with the activations it holds (wl: 64 floats in the lane's K
// order, broadcast LDS reads), plus the other half-wave's half.
static __device__ __forceinline__ float head_dot(const float *wl, const float (&bin)[KSH]) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float4 w4 = reinterpret_cast<const float4 *>(wl)[i];
        a0 = __builtin_fmaf(w4.x, bin[4 * i], a0);
        a1 = __builtin_fmaf(w4.y, bin[4 * i + 1], a1);
        a2 = __builtin_fmaf(w4.z, bin[4 * i + 2], a2);
        a3 = __builtin_fmaf(w4.w, bin[4 * i + 3], a3);
    }
    const float part = (a0 + a1) + (a2 + a3);
    return part + __shfl_xor(part, 32);
}

template <int TILES>
static __device__ __forceinline__ void relu_to_bin(const f32x16 (&acc)[TILES], float (&bin)[KSH]) {
#pragma unroll
    for (int t = 0; t < OT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) bin[t * 16 + r] = fmaxf(acc[t][r], 0.f);
}


// all 64 values a lane holds of a [128 / 4][n][4] tensor (see above): 16 quads, 2 n float4 apart
static __device__ __forceinline__ void store_bin(float *__restrict__ dst, size_t n, size_t s, const float (&bin)[KSH], int h) {
    float4 *p = quad_ptr(dst, n, s, h);
#pragma unroll
    for (int g = 0; g < KSH / 4; ++g) {
        *p = make_float4(bin[4 * g], bin[4 * g + 1], bin[4 * g + 2], bin[4 * g + 3]);
        p += 2 * n;
    }
}
// ReLU mask of the 64 activations a lane holds (slot j: bit j); stored as masks[(layer * n + sample) * 2 + half]
static __device__ __forceinline__ unsigned long long mask_of(const float (&bin)[KSH]) {
    // the values are ReLU outputs (>= +0): "positive" = "bit pattern not zero"; min(bits, 1) << j | word: two VALU
    // operations per value
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        lo |= (__float_as_uint(bin[j]) < 1u ? __float_as_uint(bin[j]) : 1u) << j;
        hi |= (__float_as_uint(bin[32 + j]) < 1u ? __float_as_uint(bin[32 + j]) : 1u) << j;
    }
    return ((unsigned long long)hi << 32) | lo;
}
// bin[j] = bit j of m ? acc[j] : 0, as bit arithmetic (sign-extended bit field AND the value: two VALU operations)
template <int TILES>
static __device__ __forceinline__ void masked_to_bin(const f32x16 (&acc)[TILES], unsigned long long m, float (&bin)[KSH]) {
    const uint32_t lo = (uint32_t)m, hi = (uint32_t)(m >> 32);
#pragma unroll
    for (int t = 0; t < OT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = t * 16 + r;
            const uint32_t sel = (uint32_t)__builtin_amdgcn_sbfe((int)(j < 32 ? lo : hi), j & 31, 1);   // 0 or 0xFFFFFFFF
            bin[j] = __uint_as_float(__float_as_uint(acc[t][r]) & sel);
        }
}

}  // namespace mlp

// packers / encoders of tn_mlp.hip, used by the training path as well
void launch_dir_encoding(size_t num_rays, const float *dirs, float *enc, hipStream_t stream);

}  // namespace tn
