// tn_mlp_grad.hip -- parameter gradients of the shallow MLP + heads from the buffers k_mlp_backward left (tn_mlp_bwd.hip).
//
// Replaces what PyTorch autograd does for the reference's model in training (tetranerf/nerfstudio/model.py:602-630: the
// weight / bias gradients of mlp_base, mlp_head and the two field heads; the trainer gets them from cuBLAS GEMMs and
// reductions).  All twelve gradients of a chunk of samples come from FIVE streaming kernels + their reductions:
//
//   k_dw_gemm<4, true>   A = d4, B = [h3 | enc(ray of the sample)]:  d Wh[:, 27:], d Wh[:, :27], d bh, and on the VALU
//                        d wd[f] = sum_s d sigma_raw[s] h3[f][s]  (h3's tile is in LDS anyway)
//   k_dw_gemm<4, false>  A = d3, B = h2: d W3, d b3;   A = d2, B = h1: d W2, d b2
//   k_dw_gemm<2, false>  A = d1, B = x0: d W1, d b1
//   k_rgb_head_grad      d wr[c][f] = sum_s d rgb_raw[c][s] h4[f][s], d bd, d br   (bandwidth-bound: 528 B per sample)
//
// k_dw_gemm: dW[128, 32 NB] = A[128, n] B[32 NB, n]^T with K = the sample axis streamed once from HBM (the operands are
// quad-major, [F/4][n][4], tn_mlp_common.h: a thread fetches four feature rows of one sample as one 16-byte load, a
// half-wave 512 contiguous bytes), 32x32x2 fp32 MFMA tiles.  A block owns a contiguous slice of samples and walks it in
// steps of 32: global -> registers (issued one step ahead) -> LDS -> MFMA operands.  The LDS rows hold the even samples
// of the step followed by the odd ones (row stride 36 floats), so that lane (row, k parity) fetches its 16 operand values
// of a step with four conflict-free 16-byte reads, and tile c + 1's operands are requested before tile c's MFMAs issue.
// Two blocks per CU (8 waves): while one block writes its tile or waits at a barrier the other one's MFMAs run.
// Round 3 (profiles/r03l_train_kernel_stats.txt): the first version -- one block per CU, 4-byte LDS reads issued right
// before their MFMA -- ran at 38 % of the fp32 MFMA rate and took 26 % of a training iteration.
//
// No atomics: every block writes its partial sums to its own slot of a scratch buffer and k_reduce_partials adds the slots
// in a fixed order INTO the caller's gradient tensors (nn.Linear layout): gradients are bit-reproducible from run to run.
#include "tn_mlp_common.h"

namespace tn {

using namespace mlp;

namespace {

constexpr int DW_GRID = 512;          // two blocks per CU
constexpr int LD = 36;                // LDS row stride (floats): 16-byte aligned, LD / 4 odd -> conflict-free b128 reads

struct DwArgs {
    const float *A;        // [128, n]
    const float *B;        // [32 NBM, n]
    const float *enc;      // EXTRA: [rays, ENC_PAD] direction encodings -> a further tile of 32 B rows (28..31 zero)
    const float *dh;       // EXTRA: d sigma_raw [n]
    uint32_t spr;          // samples per ray
    float *part;           // [gridDim.x][128 * 32 NB + 256]: dW tile, row sums of A, (EXTRA) the d wd vector
};

template <int NBM, bool EXTRA>
__global__ __launch_bounds__(256, 2) void k_dw_gemm(DwArgs g, size_t n, uint32_t slice) {
    constexpr int NB = NBM + (EXTRA ? 1 : 0);
    constexpr int RA = 128, RBM = 32 * NBM, RB = 32 * NB;
    __shared__ __attribute__((aligned(16))) float As[RA * LD];
    __shared__ __attribute__((aligned(16))) float Bs[RB * LD];
    __shared__ __attribute__((aligned(16))) float dhs[32];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int col = tid & 31, row0 = tid >> 5;            // staging: sample of the step, quad (4 feature rows) within a pass of 8
    const int cpos = (col & 1) * 16 + (col >> 1);         // even samples first, then the odd ones
    const size_t s_begin = (size_t)blockIdx.x * slice;
    const size_t s_end = s_begin + slice < n ? s_begin + slice : n;
    float *part = g.part + (size_t)blockIdx.x * (RA * RB + 256);
    constexpr int QA = RA / 32, QBM = RBM / 32;           // quads per thread: the tiles are [F / 4][n][4] (tn_mlp_common.h)

    f32x16 acc[NB];
#pragma unroll
    for (int c = 0; c < NB; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float rsum = 0.f, dv = 0.f;
    if (s_begin < n) {
        float4 ra[QA], rb[QBM];
        float re[EXTRA ? 4 : 1] = {}, rdh = 0.f;
        uint32_t e_ray = 0, e_rem = 0;   // EXTRA: ray of this thread's sample, offset of the sample within it
        const float4 *A4 = reinterpret_cast<const float4 *>(g.A), *B4 = reinterpret_cast<const float4 *>(g.B);
        auto fetch = [&](size_t s0) {
            const size_t sidx = s0 + col;
            const bool in = sidx < s_end;
            const size_t sc = in ? sidx : s_end - 1;      // clamped: loads stay unconditional, A (and dh) are zeroed
#pragma unroll
            for (int p = 0; p < QA; ++p) ra[p] = A4[(size_t)(8 * p + row0) * n + sc];
#pragma unroll
            for (int p = 0; p < QBM; ++p) rb[p] = B4[(size_t)(8 * p + row0) * n + sc];
            if constexpr (EXTRA) {
                // the encoding of the sample's ray: a thread's sample advances by 32 per step, so its ray changes every
                // spr / 32 steps -- the four values stay in registers and are re-read only then (per step: one division and
                // four gathers less; 0.87 -> 0.7x ms per 2.1 M samples, profiles/r04r_dw_ablate.txt).  Lanes beyond the
                // slice keep what they have (their A rows are zero).
                bool reload = false;
                if (s0 == s_begin) {
                    e_ray = (uint32_t)sc / g.spr;                 // n < 2^32 (checked by the launcher)
                    e_rem = (uint32_t)sc - e_ray * g.spr;
                    reload = true;
                } else if (in) {
                    e_rem += 32u;
                    while (e_rem >= g.spr) { e_rem -= g.spr; ++e_ray; reload = true; }
                }
                if (reload) {
                    const float *e = g.enc + (size_t)e_ray * ENC_PAD;
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const int j = 8 * p + row0;
                        re[p] = j < ENC_PAD ? e[j < ENC_PAD ? j : 0] : 0.f;
                    }
                }
                rdh = row0 == 0 ? g.dh[sc] : 0.f;
                if (!in) rdh = 0.f;
            }
            if (!in) {
#pragma unroll
                for (int p = 0; p < QA; ++p) ra[p] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        auto put4 = [&](float *tile, int quad, const float4 &v) {
            float *q = tile + (4 * quad) * LD + cpos;
            q[0] = v.x; q[LD] = v.y; q[2 * LD] = v.z; q[3 * LD] = v.w;
        };
        fetch(s_begin);
        const int m = lane & 31, kk = lane >> 5;
        const float4 *arow = reinterpret_cast<const float4 *>(As + (32 * w + m) * LD + kk * 16);
        const float4 *brow = reinterpret_cast<const float4 *>(Bs + m * LD + kk * 16);
        const int vf = tid >> 1, vh = tid & 1;             // d wd: feature row, sample parity
        for (size_t s0 = s_begin; s0 < s_end; s0 += 32) {
            __syncthreads();   // the previous step's reads of the tiles are done
#pragma unroll
            for (int p = 0; p < QA; ++p) put4(As, 8 * p + row0, ra[p]);
#pragma unroll
            for (int p = 0; p < QBM; ++p) put4(Bs, 8 * p + row0, rb[p]);
            if constexpr (EXTRA) {
#pragma unroll
                for (int p = 0; p < 4; ++p) Bs[(RBM + 8 * p + row0) * LD + cpos] = re[p];
                if (row0 == 0) dhs[cpos] = rdh;
            }
            __syncthreads();
            if (s0 + 32 < s_end) fetch(s0 + 32);
            float4 a4[4], b4[2][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) a4[q] = arow[q];
#pragma unroll
            for (int q = 0; q < 4; ++q) b4[0][q] = brow[q];
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                if (c + 1 < NB) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) b4[(c + 1) & 1][q] = brow[(c + 1) * 32 * LD / 4 + q];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 a = a4[q], b = b4[c & 1][q];
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[c], 0, 0, 0);
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[c], 0, 0, 0);
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[c], 0, 0, 0);
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[c], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) rsum += (a4[q].x + a4[q].y) + (a4[q].z + a4[q].w);
            if (EXTRA) {
                const float4 *hr = reinterpret_cast<const float4 *>(Bs + vf * LD + vh * 16);
                const float4 *dr = reinterpret_cast<const float4 *>(dhs + vh * 16);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 x = hr[q], d = dr[q];
                    dv += (x.x * d.x + x.y * d.y) + (x.z * d.z + x.w * d.w);
                }
            }
        }
    }
    // partial sums of this block (zeros when the block had no samples: the reduction adds every slot)
    const int hh = lane >> 5;
#pragma unroll
    for (int c = 0; c < NB; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            part[(size_t)(32 * w + acc_feature(r, hh)) * RB + 32 * c + (lane & 31)] = acc[c][r];
    rsum += __shfl_xor(rsum, 32);
    if (lane < 32) part[RA * RB + 32 * w + lane] = rsum;
    if (EXTRA) {
        dv += __shfl_xor(dv, 1);
        if ((tid & 1) == 0) part[RA * RB + 128 + (tid >> 1)] = dv;
    } else if (tid < 128) {
        part[RA * RB + 128 + tid] = 0.f;
    }
}

// out[map(i)] += sum over the blocks' slots, in a fixed order (consecutive threads read consecutive floats of a slot).  Layout of a slot: [128][RB] tile, [128] row sums, [128] vector.
struct ReduceArgs {
    const float *part;
    uint32_t nslots, RB, RBM;
    float *dW; uint32_t ldw;    // columns [0, RBM) of the tile -> dW[row * ldw + col]
    float *dWe; uint32_t ne;    // columns RBM + e, e < ne   -> dWe[row * ldw + e]       (nullable)
    float *db;                  // [128] row sums                                         (nullable)
    float *dvec;                // [128] vector                                           (nullable)
};
__global__ __launch_bounds__(256) void k_reduce_partials(ReduceArgs a) {
    // 64 outputs per block, the slots dealt over 4 thread rows; the four row sums are added in a fixed order
    __shared__ float red[4][64];
    const uint32_t count = 128 * a.RB + 256;
    const uint32_t x = threadIdx.x & 63, y = threadIdx.x >> 6;
    const uint32_t i = blockIdx.x * 64 + x;
    float s0 = 0.f, s1 = 0.f;
    if (i < count) {
        const float *p = a.part + i;
        uint32_t b = y;
        for (; b + 4 < a.nslots; b += 8) { s0 += p[(size_t)b * count]; s1 += p[(size_t)(b + 4) * count]; }
        if (b < a.nslots) s0 += p[(size_t)b * count];
    }
    red[y][x] = s0 + s1;
    __syncthreads();
    if (y != 0 || i >= count) return;
    float *dst = nullptr;
    if (i < 128 * a.RB) {
        const uint32_t row = i / a.RB, colx = i % a.RB;
        if (colx < a.RBM) dst = a.dW + (size_t)row * a.ldw + colx;
        else if (a.dWe && colx - a.RBM < a.ne) dst = a.dWe + (size_t)row * a.ldw + (colx - a.RBM);
    } else if (i < 128 * a.RB + 128) {
        if (a.db) dst = a.db + (i - 128 * a.RB);
    } else if (a.dvec) {
        dst = a.dvec + (i - 128 * a.RB - 128);
    }
    if (dst) *dst += (red[0][x] + red[1][x]) + (red[2][x] + red[3][x]);
}

// d wr[c][f] = sum_s dhead[1 + c][s] h4[f][s]; d bd = sum_s dhead[0][s]; d br[c] = sum_s dhead[1 + c][s].
// A block owns a slice of samples; wave w streams rows 32 w .. 32 w + 31 of h4 in 64-sample columns (one coalesced
// 256-byte load per row) against the three gradient rows held in registers; 96 per-lane partial sums, reduced across
// the wave once at the end.  Slot layout: [3][128] + [4].
constexpr int RGB_SLOT = 3 * 128 + 4;
__global__ __launch_bounds__(256) void k_rgb_head_grad(size_t n, uint32_t slice, const float *__restrict__ dhead,
                                                       const float *__restrict__ h4, float *__restrict__ part_all) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t s_begin = (size_t)blockIdx.x * slice;
    const size_t s_end = s_begin + slice < n ? s_begin + slice : n;
    float acc[32][3];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i][0] = acc[i][1] = acc[i][2] = 0.f;
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    const float4 *rows = reinterpret_cast<const float4 *>(h4) + (size_t)(8 * w) * n;   // [F / 4][n][4]: quads 8 w .. 8 w + 7
    for (size_t s0 = s_begin; s0 < s_end; s0 += 128) {     // two columns of 64 samples per trip: 16 row loads in flight
        float4 x[2][8];
        float dd[2][3];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const size_t s = s0 + 64 * u + lane;
            const bool in = s < s_end;
            const size_t sc = in ? s : s_end - 1;
            float d0 = dhead[sc], d1 = dhead[n + sc], d2 = dhead[2 * n + sc], d3 = dhead[3 * n + sc];
            if (!in) d0 = d1 = d2 = d3 = 0.f;
            t0 += d0; t1 += d1; t2 += d2; t3 += d3;
            dd[u][0] = d1; dd[u][1] = d2; dd[u][2] = d3;
#pragma unroll
            for (int i = 0; i < 8; ++i) x[u][i] = rows[(size_t)i * n + sc];
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float xv[4] = {x[u][i].x, x[u][i].y, x[u][i].z, x[u][i].w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    acc[4 * i + c][0] += xv[c] * dd[u][0]; acc[4 * i + c][1] += xv[c] * dd[u][1]; acc[4 * i + c][2] += xv[c] * dd[u][2];
                }
            }
    }
    float *part = part_all + (size_t)blockIdx.x * RGB_SLOT;
#pragma unroll
    for (int i = 0; i < 32; ++i)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = acc[i][c];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            if (lane == 0) part[c * 128 + 32 * w + i] = v;
        }
    if (w == 0) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            t0 += __shfl_xor(t0, off); t1 += __shfl_xor(t1, off); t2 += __shfl_xor(t2, off); t3 += __shfl_xor(t3, off);
        }
        if (lane == 0) { part[384] = t0; part[385] = t1; part[386] = t2; part[387] = t3; }
    }
}
__global__ __launch_bounds__(256) void k_reduce_rgb(const float *__restrict__ part, uint32_t nslots, float *__restrict__ gwr,
                                                    float *__restrict__ gbd, float *__restrict__ gbr) {
    __shared__ float red[4][64];
    const uint32_t x = threadIdx.x & 63, y = threadIdx.x >> 6;
    const uint32_t i = blockIdx.x * 64 + x;
    float s = 0.f;
    if (i < RGB_SLOT)
        for (uint32_t b = y; b < nslots; b += 4) s += part[(size_t)b * RGB_SLOT + i];
    red[y][x] = s;
    __syncthreads();
    if (y != 0 || i >= RGB_SLOT) return;
    s = (red[0][x] + red[1][x]) + (red[2][x] + red[3][x]);
    if (i < 384) gwr[i] += s;
    else if (i == 384) gbd[0] += s;
    else gbr[i - 385] += s;
}

struct Slicing { unsigned grid; uint32_t slice; };
Slicing slicing(size_t n, uint32_t unit) {
    // at most DW_GRID blocks, slices a multiple of `unit` samples
    size_t slice = (n + DW_GRID - 1) / DW_GRID;
    slice = (slice + unit - 1) / unit * unit;
    if (slice > 0xFFFFFFFFull) throw Error("param_grads: chunk too large");
    const size_t grid = (n + slice - 1) / slice;
    return {(unsigned)grid, (uint32_t)slice};
}

template <int NBM, bool EXTRA>
void run_dw(size_t n, const DwArgs &g, ReduceArgs r, hipStream_t stream) {
    const Slicing sl = slicing(n, 32);
    hipLaunchKernelGGL((k_dw_gemm<NBM, EXTRA>), dim3(sl.grid), dim3(256), 0, stream, g, n, sl.slice);
    r.part = g.part; r.nslots = sl.grid; r.RBM = 32 * NBM; r.RB = 32 * (NBM + (EXTRA ? 1 : 0));
    const uint32_t count = 128 * r.RB + 256;
    hipLaunchKernelGGL(k_reduce_partials, dim3((count + 63) / 64), dim3(256), 0, stream, r);
}

}  // namespace

size_t mlp_param_grad_scratch_floats() { return (size_t)DW_GRID * (128 * 160 + 256); }

void launch_mlp_param_grads(size_t n, uint32_t samples_per_ray, const float *dirs, const MlpPacks &w, const MlpBackwardBuffers &b,
                            const MlpParamGrads &g, hipStream_t stream) {
    if (n == 0) return;
    if (n > 0xFFFFFFFFull) throw Error("param_grads: more than 2^32 samples per call");
    launch_dir_encoding(n / samples_per_ray, dirs, w.enc, stream);
    float *part = w.grad_scratch;
    // mlp_head: [enc(27) | base(128)] -> 128, and the density head's weight vector
    run_dw<4, true>(n, DwArgs{b.d4, b.h3, w.enc, b.dhead, samples_per_ray, part},
                    ReduceArgs{nullptr, 0, 0, 0, g.wh + ENC, ENC + HID, g.wh, ENC, g.bh, g.wd}, stream);
    run_dw<4, false>(n, DwArgs{b.d3, b.h2, nullptr, nullptr, 0, part},
                     ReduceArgs{nullptr, 0, 0, 0, g.w3, HID, nullptr, 0, g.b3, nullptr}, stream);
    run_dw<4, false>(n, DwArgs{b.d2, b.h1, nullptr, nullptr, 0, part},
                     ReduceArgs{nullptr, 0, 0, 0, g.w2, HID, nullptr, 0, g.b2, nullptr}, stream);
    run_dw<2, false>(n, DwArgs{b.d1, b.x0, nullptr, nullptr, 0, part},
                     ReduceArgs{nullptr, 0, 0, 0, g.w1, FD, nullptr, 0, g.b1, nullptr}, stream);
    const Slicing sl = slicing(n, 128);
    hipLaunchKernelGGL(k_rgb_head_grad, dim3(sl.grid), dim3(256), 0, stream, n, sl.slice, b.dhead, b.h4, part);
    hipLaunchKernelGGL(k_reduce_rgb, dim3((RGB_SLOT + 63) / 64), dim3(256), 0, stream, part, sl.grid, g.wr, g.bd, g.br);
}

}  // namespace tn
