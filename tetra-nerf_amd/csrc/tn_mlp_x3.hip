// tn_mlp_x3.hip -- the shallow MLP + heads on the bf16 matrix cores at fp32 accuracy ("bf16x3").
//
// Optional mode of tn_mlp_forward / tn_mlp_forward_gather (tn_mlp_set_mode(1)); the default stays the
// exact fp32 MFMA kernel of tn_mlp.hip.  fp32 MFMA runs at the vector rate (157 TFLOP/s); the bf16
// MFMA (v_mfma_f32_32x32x16_bf16) is 16x faster.  Every fp32 operand is split into three bf16 pieces
//     x = x_hi + x_mid + x_lo,   x_hi = bf16(x), x_mid = bf16(x - x_hi), x_lo = bf16(x - x_hi - x_mid)
// (both subtractions are exact in fp32; the three pieces carry 24+ significant bits), and a product is
// evaluated as the six partial products of order <= 2^-16
//     w*x ~= w_lo*x_hi + w_hi*x_lo + w_mid*x_mid + w_mid*x_hi + w_hi*x_mid + w_hi*x_hi
// each exact in the MFMA's fp32 accumulator; the dropped terms are below 2^-24 |w x|, i.e. under one
// fp32 rounding of the product.  6 bf16 MFMAs (K = 16) replace 8 fp32 MFMAs (K = 2): 2.67x fewer
// matrix-core cycles for the same 1e-5 parity bar (tests/test_render_gpu.py checks it against float64).
//
// Dataflow as in tn_mlp.hip: a wavefront owns 32 samples, computes Y^T = W * X^T, and feeds the
// accumulators of one layer straight back as the B operand of the next.  With K = 16 per instruction
// lane (h, s) supplies 8 K-values per step: registers (tile q>>1, r = 8(q&1) .. +7) of its accumulators
// at step q -- the weights are packed to that K order (k_mlp_pack_x3).  A and B operands use the same
// (half-wave, element) -> k assignment, so the packing only has to agree with itself.  The bias is the
// accumulators' initial value.  Per layer the three weight pieces are staged in LDS
// ([step][tile][piece][lane] x 16 B, <= 120 KB for the head layer) and shared by the 8 waves of a block.
#include "tn_device.h"
#include "tn_kernels.h"
#include "tn_mlp_x3_fwd.h"

namespace tn {

namespace {

using namespace x3;

// ---- weight packing -------------------------------------------------------------------------------

// weight of (segment, output tile, row, K slot); segments: 0 L1, 1 L2, 2 L3, 3 head/encoding steps,
// 4 head/base steps
__device__ float x3_weight(const MlpWeights &w, int seg, int tile, int row, int q, int h, int j) {
    const int o = 32 * tile + row;
    switch (seg) {
        case 0: return w.w1[(size_t)o * FD + 32 * h + 8 * q + j];
        case 1: return w.w2[(size_t)o * HID + acc_k(q, h, j)];
        case 2: return w.w3[(size_t)o * HID + acc_k(q, h, j)];
        case 3: { const int e = 16 * q + 8 * h + j; return e < ENC ? w.wh[(size_t)o * (ENC + HID) + e] : 0.f; }
        default: return w.wh[(size_t)o * (ENC + HID) + ENC + acc_k(q, h, j)];
    }
}

__device__ float x3_bias(const MlpWeights &w, int layer, int tile, int h, int r) {
    const int o = 32 * tile + acc_feature(r, h);
    switch (layer) {
        case 0: return w.b1[o];
        case 1: return w.b2[o];
        case 2: return w.b3[o];
        default: return w.bh[o];
    }
}

struct Seg { int seg, steps, tiles; size_t off; };

__global__ void k_mlp_pack_x3(MlpWeights w, uint4 *__restrict__ blob) {
    const Seg segs[5] = {{0, 4, 4, O_L1}, {1, 8, 4, O_L2}, {2, 8, 4, O_L3}, {3, 2, 4, O_HEAD}, {4, 8, 4, O_HEAD + wu4(2, 4)}};
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int s = 0; s < 5; ++s) {
        const size_t cnt = (size_t)segs[s].steps * segs[s].tiles * 64;
        if (i < cnt) {
            const int lane = (int)(i & 63), st = (int)(i >> 6), tile = st % segs[s].tiles, q = st / segs[s].tiles;
            float v[8];
            for (int j = 0; j < 8; ++j) v[j] = x3_weight(w, segs[s].seg, tile, lane & 31, q, lane >> 5, j);
            uint4 hi, mid, lo;
            split8(v, hi, mid, lo);
            uint4 *dst = blob + segs[s].off + (size_t)st * 192 + lane;
            dst[0] = hi; dst[64] = mid; dst[128] = lo;
            return;
        }
        i -= cnt;
    }
    // biases: [tile][half][16] floats behind each layer's weights
    const size_t boff[4] = {O_L1 + wu4(4, 4), O_L2 + wu4(8, 4), O_L3 + wu4(8, 4), O_HEAD + wu4(2, 4) + wu4(8, 4)};
    for (int l = 0; l < 4; ++l) {
        if (i < 128) {
            const int r = (int)(i & 15), h = (int)((i >> 4) & 1), tile = (int)(i >> 5);
            reinterpret_cast<float *>(blob + boff[l])[i] = x3_bias(w, l, tile, h, r);
            return;
        }
        i -= 128;
    }
    // head vectors (fp32): density behind layer 3, rgb behind the head layer; K order of bin[]: feature
    // 32*(i>>4) + acc_feature(i&15, h) for element i of half h
    float *dv = reinterpret_cast<float *>(blob + O_L3 + N_L2);
    float *cv = reinterpret_cast<float *>(blob + O_HEAD + wu4(2, 4) + wu4(8, 4) + bu4(4));
    if (i < 132) {
        const int j = (int)i;
        dv[j] = j < 128 ? w.wd[32 * ((j & 63) >> 4) + acc_feature(j & 15, j >> 6)] : (j == 128 ? w.bd[0] : 0.f);
        return;
    }
    i -= 132;
    if (i < 388) {
        const int j = (int)i;
        cv[j] = j < 384 ? w.wr[(size_t)(j >> 7) * HID + 32 * ((j & 63) >> 4) + acc_feature(j & 15, (j >> 6) & 1)]
                        : (j < 387 ? w.br[j - 384] : 0.f);
    }
}
constexpr size_t PACK_THREADS = (4 * 4 + 8 * 4 + 8 * 4 + 2 * 4 + 8 * 4) * 64 + 4 * 128 + 132 + 388;

// direction encoding per ray, padded to 32 (same arithmetic as k_dir_encoding of tn_mlp.hip)
__global__ void k_dir_encoding32(size_t R, const float *__restrict__ dirs, float *__restrict__ enc) {
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const float two_pi = 6.283185307179586f, half_pi = 1.5707963267948966f;
    const float freqs[4] = {1.0f, 2.5198421478271484f, 6.349603652954102f, 16.0f};
    float *e = enc + r * ENC32;
    for (int c = 0; c < 3; ++c) {
        const float x = two_pi * dirs[3 * r + c];
        for (int f = 0; f < 4; ++f) {
            const float s = x * freqs[f];
            e[c * 4 + f] = sinf(s);
            e[12 + c * 4 + f] = sinf(s + half_pi);
        }
        e[24 + c] = dirs[3 * r + c];
    }
    for (int k = ENC; k < ENC32; ++k) e[k] = 0.f;
}

}  // namespace

template <bool GATHER, bool DENSITY_ONLY>
__global__ __launch_bounds__(X3_BLOCK) void k_mlp_forward_x3(size_t n, uint32_t samples_per_ray, const float *__restrict__ feats,
                                                             const uint32_t *__restrict__ vi, const float *__restrict__ bc,
                                                             const float *__restrict__ fieldT, const float *__restrict__ enc,
                                                             const uint4 *__restrict__ blob, float *__restrict__ sigma,
                                                             float *__restrict__ rgb, const float *__restrict__ ray_bias,
                                                             const uint32_t *__restrict__ count) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint4 *lds = reinterpret_cast<uint4 *>(smem);
    if (count) n = (size_t)*count * samples_per_ray;   // device-side ray count (sync-free callers: n = the upper bound)
    constexpr size_t GROUP = (X3_BLOCK / 64) * 32;
    const size_t ngroups = (n + GROUP - 1) / GROUP;
    for (size_t g = blockIdx.x; g < ngroups; g += gridDim.x)
        forward_group<GATHER, DENSITY_ONLY>(lds, g, n, samples_per_ray, feats, vi, bc, fieldT, enc, blob, sigma, rgb, ray_bias);
}

size_t mlp_x3_blob_u4() { return N_BLOB; }

void launch_mlp_pack_x3(const MlpWeights &w, uint4 *blob, hipStream_t stream) {
    hipLaunchKernelGGL(k_mlp_pack_x3, dim3((unsigned)((PACK_THREADS + 255) / 256)), dim3(256), 0, stream, w, blob);
}

void launch_mlp_forward_x3(size_t n, uint32_t samples_per_ray, size_t num_rays, const float *feats, const uint32_t *vi,
                           const float *bc, const float *fieldT, const float *dirs, const MlpPacks &w, float *sigma, float *rgb,
                           hipStream_t stream, const uint32_t *count) {
    if (n == 0) return;
    const bool gather = feats == nullptr;
    const bool density_only = rgb == nullptr;
    if (density_only) num_rays = 0;
    const uint4 *blob = w.blob;
    float *enc = w.enc;
    if (num_rays)
        hipLaunchKernelGGL(k_dir_encoding32, dim3((unsigned)((num_rays + 255) / 256)), dim3(256), 0, stream, num_rays, dirs, enc);
    const size_t smem = MAX_STAGE_U4 * sizeof(uint4);  // head layer: 120 KB of weight pieces + bias + rgb vectors
    static PerDeviceOnce lds_attr;
    lds_attr.run([&] {
        allow_dynamic_lds(reinterpret_cast<const void *>(k_mlp_forward_x3<false, false>), smem);
        allow_dynamic_lds(reinterpret_cast<const void *>(k_mlp_forward_x3<true, false>), smem);
        allow_dynamic_lds(reinterpret_cast<const void *>(k_mlp_forward_x3<false, true>), smem);
        allow_dynamic_lds(reinterpret_cast<const void *>(k_mlp_forward_x3<true, true>), smem);
    });
    const size_t group = (X3_BLOCK / 64) * 32;
    const size_t ngroups = (n + group - 1) / group;
    const unsigned grid = (unsigned)(ngroups < 256 ? ngroups : 256);
#define TN_X3_LAUNCH(G, D)                                                                                            \
    hipLaunchKernelGGL((k_mlp_forward_x3<G, D>), dim3(grid), dim3(X3_BLOCK), smem, stream, n, samples_per_ray, feats, vi, bc, \
                       fieldT, enc, blob, sigma, rgb, w.ray_bias, count)
    if (gather && density_only) TN_X3_LAUNCH(true, true);
    else if (gather) TN_X3_LAUNCH(true, false);
    else if (density_only) TN_X3_LAUNCH(false, true);
    else TN_X3_LAUNCH(false, false);
#undef TN_X3_LAUNCH
}

}  // namespace tn
