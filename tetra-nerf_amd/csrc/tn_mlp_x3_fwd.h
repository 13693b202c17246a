// tn_mlp_x3_fwd.h -- the bf16x3 forward of ONE group of 256 samples (8 waves x 32) as a device function: the loop body of
// k_mlp_forward_x3 (tn_mlp_x3.hip, where the arithmetic is described) and of the MLP phases of the persistent render kernel in
// its bf16x3 mode (tn_render_rays.hip, round 6) -- the same code, so the two produce identical bits.
#pragma once
#include "tn_device.h"
#include "tn_kernels.h"

namespace tn {
namespace x3 {

using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int HID = 128, FD = 64, ENC = 27, ENC32 = 32;
constexpr int X3_BLOCK = 512;

// sizes in 16-byte units
constexpr size_t wu4(int steps, int tiles) { return (size_t)steps * tiles * 3 * 64; }
constexpr size_t bu4(int tiles) { return (size_t)tiles * 8; }  // bias: [tile][half][16] floats
// The narrow heads (density 128 -> 1, rgb 128 -> 3) run on the VALU as in tn_mlp.hip: their fp32 vectors
// ([half][64] floats in the lane's K order + bias) ride behind the layer that produces their input.
constexpr size_t DVEC_U4 = (2 * 64 + 4) / 4, CVEC_U4 = (3 * 2 * 64 + 4) / 4;
constexpr size_t N_L1 = wu4(4, 4) + bu4(4);
constexpr size_t N_L2 = wu4(8, 4) + bu4(4);
constexpr size_t N_L3 = N_L2 + DVEC_U4;
constexpr size_t N_HEAD = wu4(2, 4) + wu4(8, 4) + bu4(4) + CVEC_U4;
constexpr size_t O_L1 = 0, O_L2 = O_L1 + N_L1, O_L3 = O_L2 + N_L2, O_HEAD = O_L3 + N_L3, N_BLOB = O_HEAD + N_HEAD;
constexpr size_t MAX_STAGE_U4 = N_HEAD > N_L3 ? N_HEAD : N_L3;

__host__ __device__ constexpr int acc_feature(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
// input feature consumed by K slot (step q, half h, element j) of a layer fed from accumulators
__host__ __device__ constexpr int acc_k(int q, int h, int j) { return 32 * (q >> 1) + acc_feature(8 * (q & 1) + j, h); }

static __device__ __forceinline__ uint32_t pk_bf16(float a, float b) {  // a -> low half, round to nearest even
    const f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

// 8 fp32 values -> three packed-bf16 operand registers quadruples (hi, mid, lo)
static __device__ __forceinline__ void split8(const float *v, uint4 &hi, uint4 &mid, uint4 &lo) {
    uint32_t H[4], Mi[4], L[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float a = v[2 * p], b = v[2 * p + 1];
        const uint32_t ph = pk_bf16(a, b);
        const float ra = a - __uint_as_float(ph << 16), rb = b - __uint_as_float(ph & 0xFFFF0000u);
        const uint32_t pm = pk_bf16(ra, rb);
        const float sa = ra - __uint_as_float(pm << 16), sb = rb - __uint_as_float(pm & 0xFFFF0000u);
        H[p] = ph; Mi[p] = pm; L[p] = pk_bf16(sa, sb);
    }
    hi = make_uint4(H[0], H[1], H[2], H[3]);
    mid = make_uint4(Mi[0], Mi[1], Mi[2], Mi[3]);
    lo = make_uint4(L[0], L[1], L[2], L[3]);
}

static __device__ __forceinline__ f32x16 mma(const uint4 &a, const uint4 &b, const f32x16 &c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

struct B3 { uint4 h, m, l; };  // the three bf16 pieces of 8 K-values of a lane

// one K = 16 step of NT output tiles: acc[t] += W[t](hi,mid,lo) x B(hi,mid,lo), six partial products,
// small terms first; two tiles are interleaved so that consecutive MFMAs never share an accumulator
template <int NT, int TILES>
static __device__ __forceinline__ void x3_mma(f32x16 (&acc)[TILES], const uint4 *wl, const B3 &b, int lane) {
#pragma unroll
    for (int t = 0; t + 1 < NT; t += 2) {
        const uint4 *w0 = wl + (size_t)t * 192 + lane, *w1 = w0 + 192;
        const uint4 ah0 = w0[0], am0 = w0[64], al0 = w0[128];
        const uint4 ah1 = w1[0], am1 = w1[64], al1 = w1[128];
        acc[t] = mma(al0, b.h, acc[t]);     acc[t + 1] = mma(al1, b.h, acc[t + 1]);
        acc[t] = mma(ah0, b.l, acc[t]);     acc[t + 1] = mma(ah1, b.l, acc[t + 1]);
        acc[t] = mma(am0, b.m, acc[t]);     acc[t + 1] = mma(am1, b.m, acc[t + 1]);
        acc[t] = mma(am0, b.h, acc[t]);     acc[t + 1] = mma(am1, b.h, acc[t + 1]);
        acc[t] = mma(ah0, b.m, acc[t]);     acc[t + 1] = mma(ah1, b.m, acc[t + 1]);
        acc[t] = mma(ah0, b.h, acc[t]);     acc[t + 1] = mma(ah1, b.h, acc[t + 1]);
    }
    if constexpr (NT & 1) {
        constexpr int t = NT - 1;
        const uint4 *w0 = wl + (size_t)t * 192 + lane;
        const uint4 ah0 = w0[0], am0 = w0[64], al0 = w0[128];
        acc[t] = mma(al0, b.h, acc[t]);
        acc[t] = mma(ah0, b.l, acc[t]);
        acc[t] = mma(am0, b.m, acc[t]);
        acc[t] = mma(am0, b.h, acc[t]);
        acc[t] = mma(ah0, b.m, acc[t]);
        acc[t] = mma(ah0, b.h, acc[t]);
    }
}

// STEPS consecutive K = 16 steps over bin[0 .. 8*STEPS): the operand split of step q+1 (VALU) is issued
// in the same scheduling region as the MFMAs of step q, so it runs in their shadow; the sched_barrier
// between regions keeps the A-operand reads of later steps from being hoisted (registers).
template <int STEPS, int NT, int TILES>
static __device__ __forceinline__ void x3_steps(f32x16 (&acc)[TILES], const uint4 *wl, const float *bin, int lane) {
    B3 cur, nxt;
    split8(bin, cur.h, cur.m, cur.l);
#pragma unroll
    for (int q = 0; q < STEPS; ++q) {
        if (q + 1 < STEPS) split8(bin + 8 * (q + 1), nxt.h, nxt.m, nxt.l);
        x3_mma<NT>(acc, wl + (size_t)q * NT * 192, cur, lane);
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
    }
}

template <int TILES>
static __device__ __forceinline__ void init_bias(f32x16 (&acc)[TILES], const uint4 *bias, int h) {
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
        const float4 *b = reinterpret_cast<const float4 *>(bias) + (t * 2 + h) * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 x = b[q];
            acc[t][4 * q] = x.x; acc[t][4 * q + 1] = x.y; acc[t][4 * q + 2] = x.z; acc[t][4 * q + 3] = x.w;
        }
    }
}

static __device__ __forceinline__ float head_dot(const float *wl, const float (&bin)[64]) {
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
static __device__ __forceinline__ void relu_to_bin(const f32x16 (&acc)[TILES], float (&bin)[64]) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) bin[t * 16 + r] = fmaxf(acc[t][r], 0.f);
}

// Layer blob -> LDS with the async global->LDS path (global_load_lds_dwordx4: no staging registers, all
// of a thread's loads in flight at once; a load-wait-write loop exposes one L2 latency per 8 KB).  The LDS
// destination of a wave is uniform base + lane * 16, which is exactly a linear copy.
static __device__ __forceinline__ void stage(uint4 *lds, const uint4 *__restrict__ src, uint32_t n16) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave0 = __builtin_amdgcn_readfirstlane(threadIdx.x & ~63u);
    for (uint32_t base = wave0; base < n16; base += X3_BLOCK) {
        const uint32_t i = base + lane;
        if (i < n16)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + i),
                                             (__attribute__((address_space(3))) void *)(lds + base), 16, 0, 0);
    }
}
static __device__ __forceinline__ void stage_wait() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}


// One group: samples g * 256 + wave * 32 + (lane & 31) of n; lds: MAX_STAGE_U4 uint4.  enc f32 [rays][32]: the direction encoding
// of the sample's ray (ray = sample / samples_per_ray), ray_bias f32 [rays][128] or null; blob: k_mlp_pack_x3's.  All 512 threads
// of the block call it together (block barriers inside).
template <bool GATHER, bool DENSITY_ONLY>
static __device__ __forceinline__ void forward_group(uint4 *lds, size_t g, size_t n, uint32_t samples_per_ray, const float *__restrict__ feats,
                                                     const uint32_t *__restrict__ vi, const float *__restrict__ bc,
                                                     const float *__restrict__ fieldT, const float *__restrict__ enc,
                                                     const uint4 *__restrict__ blob, float *__restrict__ sigma, float *__restrict__ rgb,
                                                     const float *__restrict__ ray_bias) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5;
    constexpr size_t GROUP = (X3_BLOCK / 64) * 32;
    const size_t s = g * GROUP + (size_t)wave * 32 + (lane & 31);
    const size_t sc = s < n ? s : n - 1;
    float bin[64];

    // ---- layer 1: this lane supplies features 32h .. 32h+31 of its sample
    __syncthreads();
    stage(lds, blob + O_L1, N_L1);
    if constexpr (!GATHER) {  // B operands straight from the feature-major input [64, n]
#pragma unroll
        for (int i = 0; i < 32; ++i) bin[i] = feats[(size_t)(32 * h + i) * n + sc];
    } else {
        const uint4 v4 = *reinterpret_cast<const uint4 *>(vi + 4 * sc);
        const float b0 = bc[3 * sc], b1 = bc[3 * sc + 1], b2 = bc[3 * sc + 2];
        const float w0 = 1.0f - ((b0 + b1) + b2);
        const uint32_t vv[4] = {v4.y, v4.z, v4.w, v4.x};
        const float ww[4] = {b0, b1, b2, w0};
#pragma unroll
        for (int i = 0; i < 32; ++i) bin[i] = 0.f;
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
        f32x16 acc[4];
        init_bias(acc, lds + wu4(4, 4), h);
        x3_steps<4, 4>(acc, lds, bin, lane);
        relu_to_bin(acc, bin);
    }
    // ---- layers 2, 3
#pragma unroll
    for (int l = 0; l < 2; ++l) {
        __syncthreads();
        stage(lds, blob + (l == 0 ? O_L2 : O_L3), l == 0 ? N_L2 : N_L3);
        stage_wait();
        f32x16 acc[4];
        init_bias(acc, lds + wu4(8, 4), h);
        x3_steps<8, 4>(acc, lds, bin, lane);
        relu_to_bin(acc, bin);
    }
    {
        // density head 128 -> 1 + softplus on the VALU, fp32 (vector behind layer 3's blob)
        const float *dv = reinterpret_cast<const float *>(lds + N_L2);
        const float raw = head_dot(dv + 64 * h, bin) + dv[128];
        const float sp = raw > 20.0f ? raw : log1pf(expf(raw));
        if (h == 0 && s < n) sigma[s] = sp;
    }
    if constexpr (DENSITY_ONLY) return;
    // ---- head [enc(27) | base(128)] -> 128 ReLU
    __syncthreads();
    stage(lds, blob + O_HEAD, N_HEAD);
    stage_wait();
    {
        f32x16 acc[4];
        init_bias(acc, lds + wu4(2, 4) + wu4(8, 4), h);
        const float *e = enc + (sc / samples_per_ray) * ENC32;
        float ev[16];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float4 e0 = *reinterpret_cast<const float4 *>(e + 16 * q + 8 * h);
            const float4 e1 = *reinterpret_cast<const float4 *>(e + 16 * q + 8 * h + 4);
            ev[8 * q] = e0.x; ev[8 * q + 1] = e0.y; ev[8 * q + 2] = e0.z; ev[8 * q + 3] = e0.w;
            ev[8 * q + 4] = e1.x; ev[8 * q + 5] = e1.y; ev[8 * q + 6] = e1.z; ev[8 * q + 7] = e1.w;
        }
        x3_steps<2, 4>(acc, lds, ev, lane);
        x3_steps<8, 4>(acc, lds + wu4(2, 4), bin, lane);
        if (ray_bias) {   // per-ray head bias (appearance embedding; tn_mlp_common.h: add_ray_bias), wave-uniform test
            const float *row = ray_bias + (sc / samples_per_ray) * HID;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = *reinterpret_cast<const float4 *>(row + 32 * t + 8 * q + 4 * h);
                    acc[t][4 * q] += v.x; acc[t][4 * q + 1] += v.y; acc[t][4 * q + 2] += v.z; acc[t][4 * q + 3] += v.w;
                }
        }
        relu_to_bin(acc, bin);
    }
    {
        // rgb head 128 -> 3 + sigmoid on the VALU, fp32
        const float *cv = reinterpret_cast<const float *>(lds + wu4(2, 4) + wu4(8, 4) + bu4(4));
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

}  // namespace x3
}  // namespace tn
