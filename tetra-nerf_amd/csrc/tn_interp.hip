// tn_interp.hip -- barycentric feature gather (interpolate_values) and its adjoint.
//
// Replaces interpolate_values_kernel<D> / interpolate_values_backward_kernel<D>
// (src/tetrahedra_tracer.cu:195-248): there a thread is one (sample, feature) pair on a
// (n/1024, field_dim) grid, so indices and weights are re-read field_dim times and every
// field access is a lone 4-byte gather into a feature-major [Fd, V] table.
//
// Here the table is first transposed to vertex-major [V, Fd] (one 256-byte line per vertex
// at Fd = 64; the table is L2/MALL resident), then ONE WAVEFRONT processes 64 samples with
// LANE = FEATURE: per sample the D vertex rows are read as coalesced 256-B loads (ids and
// weights broadcast from the owning lane by v_readlane), results go through a padded LDS
// tile and leave as 256-B row stores into the reference's [Fd, n] result layout.
// The per-element summation order of the reference is kept, so the forward result is
// bit-identical to the CPU oracle:
//     out = ((b0*f[v1] + b1*f[v2]) + ...) + (1 - ((b0+b1)+...)) * f[v0],  EMPTY ids skipped.
// Backward: LANE = FEATURE; 64 lanes add to 64 consecutive floats of a vertex-major gradient (one
// cache line per atomic instruction), with run-length combining of consecutive samples that hit the
// same vertex tuple; the result is transposed back to [Fd, V].
#include <cstdlib>
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "tn_device.h"
#include "tn_kernels.h"

namespace tn {

namespace {

constexpr int TS = 64;   // samples per tile
constexpr int TP = 65;   // padded LDS row (floats)

// [rows, cols] -> [cols, rows], 64x64 tiles through LDS.  1-D grid over the tiles (row-tile major): neither dimension
// is bounded by the 65535 blocks of gridDim.y (a [64, V] field with V > 4.19 M vertices has that many column tiles).
__global__ __launch_bounds__(256) void k_transpose(const float *__restrict__ in, float *__restrict__ out,
                                                   uint32_t rows, uint32_t cols) {
    __shared__ float tile[64][TP];
    const uint32_t tiles_c = (cols + 63) / 64;
    const uint32_t c0 = (blockIdx.x % tiles_c) * 64, r0 = (blockIdx.x / tiles_c) * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4)
        if (r0 + r < rows && c0 + tx < cols) tile[r][tx] = in[(size_t)(r0 + r) * cols + c0 + tx];
    __syncthreads();
    for (int c = ty; c < 64; c += 4)
        if (c0 + c < cols && r0 + tx < rows) out[(size_t)(c0 + c) * rows + r0 + tx] = tile[tx][c];
}

// Forward: one wavefront = 32 samples; lane (h = lane >> 5, s = lane & 31) produces the 32 features
// 32h .. 32h+31 (of each 64-feature block) of sample s.  Every vertex row is read as 8 contiguous
// 16-B loads per lane (128 B), every result store of the wave is two fully used 128-B segments of the
// reference's [Fd, n] layout -- no LDS, no transposition, ~50 VGPRs.
template <int D>
__global__ __launch_bounds__(256) void k_interp_fwd(uint32_t n, uint32_t Fd, const uint32_t *__restrict__ vi,
                                                    const float *__restrict__ bc, const float *__restrict__ fieldT,
                                                    float *__restrict__ result) {
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const uint32_t ntiles = (n + 31) / 32;
    const uint32_t wave0 = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t nwaves = gridDim.x * (blockDim.x >> 6);
    for (uint32_t tix = wave0; tix < ntiles; tix += nwaves) {
        const uint32_t s = tix * 32 + (lane & 31);
        const bool ok = s < n;
        uint32_t v[D];
        float b[D - 1];
#pragma unroll
        for (int k = 0; k < D; ++k) v[k] = ok ? vi[(size_t)s * D + k] : TN_EMPTY;
#pragma unroll
        for (int k = 0; k < D - 1; ++k) b[k] = ok ? bc[(size_t)s * (D - 1) + k] : 0.f;
        float w = 0.f;
#pragma unroll
        for (int k = 0; k < D - 1; ++k) w += b[k];
        const float w0 = 1.0f - w;
        for (uint32_t f0 = 32 * h; f0 < Fd; f0 += 64) {
            float out[32];
#pragma unroll
            for (int j = 0; j < 32; ++j) out[j] = 0.f;
            // reference order: the D-1 weighted vertices first, the implicit-weight vertex last
#pragma unroll
            for (int k = 0; k < D; ++k) {
                const int kk = k < D - 1 ? k + 1 : 0;
                const float wk = k < D - 1 ? b[k < D - 1 ? k : 0] : w0;
                if (v[kk] != TN_EMPTY) {
                    const float *row = fieldT + (size_t)v[kk] * Fd + f0;
                    if (f0 + 32 <= Fd && (Fd & 3) == 0) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const float4 x = reinterpret_cast<const float4 *>(row)[q];
                            out[4 * q] += wk * x.x; out[4 * q + 1] += wk * x.y;
                            out[4 * q + 2] += wk * x.z; out[4 * q + 3] += wk * x.w;
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 32; ++j)
                            if (f0 + j < Fd) out[j] += wk * row[j];
                    }
                }
            }
            if (ok) {
#pragma unroll
                for (int j = 0; j < 32; ++j)
                    if (f0 + j < Fd) result[(size_t)(f0 + j) * n + s] = out[j];
            }
        }
    }
}

// Forward, Fd = 64 (the model's field): one wavefront = 64 samples in 8 iterations of 8 samples.  Lane (q = lane & 7,
// s = lane >> 3) owns features {4q..4q+3} and {32+4q..32+4q+3} of samples base + 8s + it: the 8 lanes of a sample read
// a vertex row as two instructions of 8 x 16 B = one full 128-byte line each (8 line requests per sample for its
// 4 vertices; the 32-samples-per-wave kernel above issues 64, one per lane per 16-byte piece, and was bound by the
// L1 request rate, not by HBM), 8 row loads are in flight per lane, and every feature row of the reference's
// [Fd, n] result receives 8 consecutive samples = 32 B per lane, 256 contiguous bytes per wave.  Same summation
// order as the reference -> bit-identical to the oracle.
template <int D>
__global__ __launch_bounds__(256) void k_interp_fwd64(uint32_t n, const uint32_t *__restrict__ vi,
                                                      const float *__restrict__ bc, const float *__restrict__ fieldT,
                                                      float *__restrict__ result) {
    const int lane = threadIdx.x & 63;
    const uint32_t q = (uint32_t)lane & 7u, sg = (uint32_t)lane >> 3;
    const uint32_t ntiles = (n + 63) / 64;
    const uint32_t wave0 = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t nwaves = gridDim.x * (blockDim.x >> 6);
    for (uint32_t tix = wave0; tix < ntiles; tix += nwaves) {
        const uint32_t base = tix * 64 + 8 * sg;
        float acc[8][8];
        // the rows of the previous sample: consecutive samples of a ray lie in the same or in adjacent tetrahedra, and the
        // vertex rows (1 KB per sample between the L2 and the CU) are what bounds this kernel, so a row that repeats is
        // taken from registers
        uint32_t pv[D];
        float4 p0[D], p1[D];
#pragma unroll
        for (int k = 0; k < D; ++k) { pv[k] = TN_EMPTY; p0[k] = make_float4(0.f, 0.f, 0.f, 0.f); p1[k] = p0[k]; }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[it][j] = 0.f;
            const uint32_t smp = base + it;
            const bool ok = smp < n;
            uint32_t v[D];
            float b[D - 1];
#pragma unroll
            for (int k = 0; k < D; ++k) v[k] = ok ? vi[(size_t)smp * D + k] : TN_EMPTY;
#pragma unroll
            for (int k = 0; k < D - 1; ++k) b[k] = ok ? bc[(size_t)smp * (D - 1) + k] : 0.f;
            float w = 0.f;
#pragma unroll
            for (int k = 0; k < D - 1; ++k) w += b[k];
            const float w0 = 1.0f - w;
            // the rows of this sample: from the previous sample's registers wherever the vertex was used there in ANY slot
            // (the next tetrahedron along a ray shares three of its four vertices, in whatever order the mesh lists them;
            // round 3's counters: 3.4 row loads per sample with same-slot reuse only), otherwise from the field
            float4 n0[D], n1[D];
#pragma unroll
            for (int k = 0; k < D; ++k) {
                n0[k] = make_float4(0.f, 0.f, 0.f, 0.f); n1[k] = n0[k];
                if (v[k] != TN_EMPTY) {
                    bool found = false;
#pragma unroll
                    for (int c = 0; c < D; ++c)
                        if (v[k] == pv[c]) { n0[k] = p0[c]; n1[k] = p1[c]; found = true; }
                    if (!found) {
                        const float4 *row = reinterpret_cast<const float4 *>(fieldT + (size_t)v[k] * 64);
                        n0[k] = row[q]; n1[k] = row[8 + q];
                    }
                }
            }
            // reference order: the D-1 weighted vertices first, the implicit-weight vertex last
#pragma unroll
            for (int k = 0; k < D; ++k) {
                const int kk = k < D - 1 ? k + 1 : 0;
                const float wk = k < D - 1 ? b[k < D - 1 ? k : 0] : w0;
                if (v[kk] != TN_EMPTY) {
                    const float4 x0 = n0[kk], x1 = n1[kk];
                    acc[it][0] += wk * x0.x; acc[it][1] += wk * x0.y; acc[it][2] += wk * x0.z; acc[it][3] += wk * x0.w;
                    acc[it][4] += wk * x1.x; acc[it][5] += wk * x1.y; acc[it][6] += wk * x1.z; acc[it][7] += wk * x1.w;
                }
            }
#pragma unroll
            for (int k = 0; k < D; ++k) { pv[k] = v[k]; p0[k] = n0[k]; p1[k] = n1[k]; }
        }
        const bool wide = (n & 3u) == 0 && base + 8 <= n;   // 16-byte aligned rows, all 8 samples valid
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t f = (j < 4 ? 4 * q + j : 32 + 4 * q + (j - 4));
            float *dst = result + (size_t)f * n + base;
            if (wide) {
                reinterpret_cast<float4 *>(dst)[0] = make_float4(acc[0][j], acc[1][j], acc[2][j], acc[3][j]);
                reinterpret_cast<float4 *>(dst)[1] = make_float4(acc[4][j], acc[5][j], acc[6][j], acc[7][j]);
            } else {
#pragma unroll
                for (int it = 0; it < 8; ++it)
                    if (base + it < n) dst[it] = acc[it][j];
            }
        }
    }
}

// Round 6, measured and dropped (profiles/r06g_ops_ab.txt; the kernel is in the history, commit "wip: ... straight-line gather
// kernel draft"): the ISA of k_interp_fwd64 above has 558 branches and 41 s_waitcnt vmcnt(0) per 64-sample tile (the "register
// or load" selects of the carry become branches around loads), so a STRAIGHT-LINE form was tried: every load unconditional (clamped
// index, EMPTY ids read row 0 and are discarded by selects), no carry, all 64 row requests of a tile in flight (324 registers,
// 7 drains).  In-process A/B, bit-identical: 250 -> 278 us (4096 x 513 samples), 1927 -> 2300 us (65,536 x 256): 11-19 % SLOWER.
// The carry's saved L2 -> L1 row traffic (1 KB per sample without it) is worth more than the drains cost: round 3's reading of the
// counters (this op sits on the L2 -> L1 return path) stands.

// Backward: one wavefront = 64 consecutive samples, LANE = FEATURE.  The incoming gradient is read
// sample-major ([n, Fd] rows: one coalesced 256-B load per sample at Fd = 64); vertex ids and weights
// are wave-uniform and come through scalar loads; consecutive samples with the same vertex tuple
// (same tetrahedron along the ray) are combined in registers; when the tuple changes the sums of the vertices that
// stay are carried over, and each flush is one atomic instruction whose 64 lanes hit 64 consecutive floats of the
// vertex-major gradient (one or two cache lines).
template <int D>
__global__ __launch_bounds__(256) void k_interp_bwd(uint32_t n, uint32_t Fd, const uint32_t *__restrict__ vi,
                                                    const float *__restrict__ bc,
                                                    const float *__restrict__ grad_rows, float *__restrict__ gradT) {
    constexpr int B = 8;  // samples whose gradient rows are in flight together
    const int lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    const uint32_t nwaves = gridDim.x * (blockDim.x >> 6);
    const uint32_t ntiles = (n + TS - 1) / TS;
    for (uint32_t tix = wave; tix < ntiles; tix += nwaves) {
        const uint32_t base = tix * TS;
        const uint32_t cnt = n - base < TS ? n - base : TS;
        for (uint32_t f0 = 0; f0 < Fd; f0 += 64) {
            const uint32_t f = f0 + lane;
            const bool fok = f < Fd;
            uint32_t cur[D];
            float acc[D];
#pragma unroll
            for (int k = 0; k < D; ++k) { cur[k] = TN_EMPTY; acc[k] = 0.f; }
            for (uint32_t s0 = 0; s0 < cnt; s0 += B) {
                float g[B];
#pragma unroll
                for (int j = 0; j < B; ++j)
                    g[j] = (s0 + j < cnt && fok) ? grad_rows[(size_t)(base + s0 + j) * Fd + f] : 0.f;
#pragma unroll
                for (int j = 0; j < B; ++j) {
                    if (s0 + j >= cnt) break;
                    const size_t i = base + s0 + j;
                    uint32_t v[D];
                    float wgt[D];
                    float w = 0.f;
                    bool same = true;
#pragma unroll
                    for (int k = 0; k < D; ++k) { v[k] = vi[i * D + k]; same = same && (v[k] == cur[k]); }
#pragma unroll
                    for (int k = 0; k < D - 1; ++k) { wgt[k + 1] = bc[i * (D - 1) + k]; w += wgt[k + 1]; }
                    wgt[0] = 1.0f - w;
                    if (!same) {  // wave-uniform
                        // the next tetrahedron along a ray shares all but one of its vertices, in whatever order the mesh
                        // lists them: the sums of the vertices that stay MOVE to their new slots, only the others are
                        // flushed (one atomic instruction per step instead of D; the per-sample path stays slot-wise)
                        float nacc[D];
                        uint32_t carried = 0;
#pragma unroll
                        for (int a = 0; a < D; ++a) {
                            nacc[a] = 0.f;
                            bool taken = v[a] == TN_EMPTY;
#pragma unroll
                            for (int c = 0; c < D; ++c)
                                if (!taken && cur[c] == v[a] && !((carried >> c) & 1u)) { nacc[a] = acc[c]; carried |= 1u << c; taken = true; }
                        }
#pragma unroll
                        for (int c = 0; c < D; ++c)
                            if (!((carried >> c) & 1u) && cur[c] != TN_EMPTY && fok) atomicAdd(&gradT[(size_t)cur[c] * Fd + f], acc[c]);
#pragma unroll
                        for (int k = 0; k < D; ++k) { cur[k] = v[k]; acc[k] = nacc[k]; }
                    }
#pragma unroll
                    for (int k = 0; k < D; ++k) acc[k] += wgt[k] * g[j];
                }
            }
#pragma unroll
            for (int k = 0; k < D; ++k)
                if (cur[k] != TN_EMPTY && fok) atomicAdd(&gradT[(size_t)cur[k] * Fd + f], acc[k]);
        }
    }
}

// stream-ordered temporary, released when the scope is left (also when a launch check throws)
struct AsyncBuf {
    float *p = nullptr;
    hipStream_t s;
    AsyncBuf(size_t floats, hipStream_t stream) : s(stream) {
        if (floats) TN_HIP(hipMallocAsync((void **)&p, floats * sizeof(float), stream));
    }
    ~AsyncBuf() { if (p) (void)hipFreeAsync(p, s); }
    AsyncBuf(const AsyncBuf &) = delete;
    AsyncBuf &operator=(const AsyncBuf &) = delete;
};

// fieldT: the field vertex-major [V, Fd]
template <int D>
void run_fwd_vm(uint32_t n, uint32_t Fd, const uint32_t *vi, const float *bc, const float *fieldT, float *result,
                hipStream_t stream) {
    if (Fd == 64) {
        const uint32_t nblocks = ((n + 63) / 64 + 3) / 4;  // 4 waves (256 samples) per block
        const unsigned grid = nblocks < 256u * 16u ? nblocks : 256u * 16u;
        hipLaunchKernelGGL(k_interp_fwd64<D>, dim3(grid), dim3(256), 0, stream, n, vi, bc, fieldT, result);
        return;
    }
    const uint32_t nblocks = ((n + 31) / 32 + 3) / 4;  // 4 waves (128 samples) per block
    const unsigned grid = nblocks < 256u * 16u ? nblocks : 256u * 16u;
    hipLaunchKernelGGL(k_interp_fwd<D>, dim3(grid), dim3(256), 0, stream, n, Fd, vi, bc, fieldT, result);
}

// gradT: vertex-major [V, Fd] gradient, ACCUMULATED into; rows: gradient rows [n, Fd]
template <int D>
void run_bwd_vm(uint32_t n, uint32_t Fd, const uint32_t *vi, const float *bc, const float *rows, float *gradT,
                hipStream_t stream) {
    if (n == 0) return;
    const uint32_t nblocks = ((n + TS - 1) / TS + 3) / 4;  // 4 waves per block
    const unsigned grid = nblocks < 256u * 32u ? nblocks : 256u * 32u;
    hipLaunchKernelGGL(k_interp_bwd<D>, dim3(grid), dim3(256), 0, stream, n, Fd, vi, bc, rows, gradT);
}

template <int D>
void run_fwd(uint32_t V, uint32_t n, uint32_t Fd, const uint32_t *vi, const float *bc, const float *field,
             float *result, hipStream_t stream) {
    AsyncBuf fieldT((size_t)V * Fd, stream);
    launch_transpose(field, fieldT.p, Fd, V, stream);
    run_fwd_vm<D>(n, Fd, vi, bc, fieldT.p, result, stream);
}

// rows_major: grad_in is [n, Fd] (what autograd hands over for a contiguous [..., Fd] gradient);
// otherwise the reference's native layout [Fd, n] (tetrahedra_tracer.h:404-411), transposed first.
template <int D>
void run_bwd(uint32_t V, uint32_t n, uint32_t Fd, const uint32_t *vi, const float *bc, const float *grad_in,
             bool rows_major, float *field_grad, hipStream_t stream) {
    AsyncBuf gradT((size_t)V * Fd, stream);
    TN_HIP(hipMemsetAsync(gradT.p, 0, (size_t)V * Fd * sizeof(float), stream));
    if (n) {
        AsyncBuf rows(rows_major ? 0 : (size_t)n * Fd, stream);
        if (!rows_major) {
            launch_transpose(grad_in, rows.p, Fd, n, stream);
            grad_in = rows.p;
        }
        run_bwd_vm<D>(n, Fd, vi, bc, grad_in, gradT.p, stream);
    }
    launch_transpose(gradT.p, field_grad, V, Fd, stream);
}

// ---- the adjoint WITHOUT float atomics (opt-in: bit-reproducible field gradients) -------------------------------------
// The (sample, vertex slot) pairs are sorted by vertex (stable radix sort: inside a vertex they stay in sample order), then
// one block per vertex adds its pairs' weighted gradient rows in a FIXED order: wave w of the block takes the w-th quarter
// of the vertex's run, lane = feature, sequential over the run; the four partial sums are combined as (p0 + p1) + (p2 + p3).
// One writer per gradient element: the result does not depend on scheduling.  ~3-5x the time of the atomic kernel.
__global__ void k_iota(uint32_t n, uint32_t *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = i;
}
__global__ void k_run_bounds(uint32_t n, const uint32_t *__restrict__ keys, uint32_t V, uint32_t *__restrict__ start,
                             uint32_t *__restrict__ end) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t k = keys[i];
    if (k >= V) return;                                   // TN_EMPTY (unmatched slot) sorts last
    if (i == 0 || keys[i - 1] != k) start[k] = i;
    if (i + 1 == n || keys[i + 1] != k) end[k] = i + 1;
}
template <int D>
__global__ __launch_bounds__(256) void k_interp_bwd_det(uint32_t V, uint32_t Fd, const uint32_t *__restrict__ start,
                                                        const uint32_t *__restrict__ end, const uint32_t *__restrict__ pairs,
                                                        const float *__restrict__ bc, const float *__restrict__ grad_rows,
                                                        float *__restrict__ gradT) {
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    for (uint32_t v = blockIdx.x; v < V; v += gridDim.x) {
        const uint32_t st = start[v], en = end[v];
        if (en <= st) continue;                            // block-uniform
        const uint32_t len = en - st;
        const uint32_t a = st + (uint32_t)(((uint64_t)len * wave) / 4), b = st + (uint32_t)(((uint64_t)len * (wave + 1)) / 4);
        for (uint32_t f0 = 0; f0 < Fd; f0 += 64) {
            const uint32_t f = f0 + lane;
            const bool fok = f < Fd;
            float acc = 0.f;
            for (uint32_t i0 = a; i0 < b; i0 += 8) {
                float g[8], w[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    g[j] = 0.f; w[j] = 0.f;
                    if (i0 + j < b) {
                        const uint32_t p = pairs[i0 + j];           // wave-uniform
                        const uint32_t smp = p / D, k = p % D;
                        if (k == 0) {
                            float t = 0.f;
#pragma unroll
                            for (int c = 0; c < D - 1; ++c) t += bc[(size_t)smp * (D - 1) + c];
                            w[j] = 1.0f - t;
                        } else {
                            w[j] = bc[(size_t)smp * (D - 1) + (k - 1)];
                        }
                        if (fok) g[j] = grad_rows[(size_t)smp * Fd + f];
                    }
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) acc += w[j] * g[j];   // (slots beyond the run add w = 0 times g = 0)
            }
            part[wave][lane] = acc;
            __syncthreads();
            if (wave == 0 && fok) gradT[(size_t)v * Fd + f] += (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
            __syncthreads();
        }
    }
}

template <int D>
void run_bwd_det(uint32_t V, uint32_t n, uint32_t Fd, const uint32_t *vi, const float *bc, const float *grad_rows, float *gradT,
                 hipStream_t stream) {
    const size_t np = (size_t)n * D;
    if (np > 0xFFFFFFFFull) throw Error("interpolate_values backward (deterministic): too many samples");
    // ONE stream-ordered allocation per call (keys | values | iota | run bounds | rocprim's scratch); the pool keeps it cached
    // between calls, so a training step that takes this path every iteration does not touch the allocator after the first
    size_t bytes = 0;
    TN_HIP(rocprim::radix_sort_pairs(nullptr, bytes, vi, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, np, 0u, 32u, stream));
    const size_t npa = (np + 63) & ~(size_t)63, nb = (2 * (size_t)V + 63) & ~(size_t)63;
    AsyncBuf all(3 * npa + nb + (bytes + 3) / 4 + 64, stream);
    uint32_t *ks = reinterpret_cast<uint32_t *>(all.p), *vs = ks + npa, *io = vs + npa;
    uint32_t *st = io + npa, *en = st + V;
    void *tmp = st + nb;
    hipLaunchKernelGGL(k_iota, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, stream, (uint32_t)np, io);
    TN_HIP(rocprim::radix_sort_pairs(tmp, bytes, vi, ks, io, vs, np, 0u, 32u, stream));
    TN_HIP(hipMemsetAsync(st, 0, 2 * (size_t)V * sizeof(uint32_t), stream));
    hipLaunchKernelGGL(k_run_bounds, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, stream, (uint32_t)np, ks, V, st, en);
    const unsigned grid = V < 256u * 64u ? V : 256u * 64u;
    hipLaunchKernelGGL(k_interp_bwd_det<D>, dim3(grid), dim3(256), 0, stream, V, Fd, st, en, vs, bc, grad_rows, gradT);
}

}  // namespace

void launch_interpolate_values_backward_vm_det(uint32_t D, uint32_t V, uint32_t n, uint32_t Fd, const uint32_t *vi, const float *bc,
                                               const float *grad_rows, float *gradT, hipStream_t stream) {
    if (n == 0 || Fd == 0 || V == 0) return;
    switch (D) {
        case 2: run_bwd_det<2>(V, n, Fd, vi, bc, grad_rows, gradT, stream); break;
        case 3: run_bwd_det<3>(V, n, Fd, vi, bc, grad_rows, gradT, stream); break;
        case 4: run_bwd_det<4>(V, n, Fd, vi, bc, grad_rows, gradT, stream); break;
        case 6: run_bwd_det<6>(V, n, Fd, vi, bc, grad_rows, gradT, stream); break;
        default: throw Error("Unsupported interpolation dimension with value " + std::to_string(D));
    }
}

void launch_transpose(const float *in, float *out, uint32_t rows, uint32_t cols, hipStream_t stream) {
    if (rows == 0 || cols == 0) return;
    const size_t tiles = (size_t)((cols + 63) / 64) * ((rows + 63) / 64);
    if (tiles > 0x7FFFFFFFull) throw Error("transpose: matrix too large");
    hipLaunchKernelGGL(k_transpose, dim3((unsigned)tiles), dim3(256), 0, stream, in, out, rows, cols);
}

void launch_interpolate_values(uint32_t D, uint32_t V, uint32_t n, uint32_t Fd, const uint32_t *vi,
                               const float *bc, const float *field, float *result, hipStream_t stream) {
    if (n == 0 || Fd == 0) return;
    switch (D) {
        case 2: run_fwd<2>(V, n, Fd, vi, bc, field, result, stream); break;
        case 3: run_fwd<3>(V, n, Fd, vi, bc, field, result, stream); break;
        case 4: run_fwd<4>(V, n, Fd, vi, bc, field, result, stream); break;
        case 6: run_fwd<6>(V, n, Fd, vi, bc, field, result, stream); break;
        default: throw Error("Unsupported interpolation dimension with value " + std::to_string(D));
    }
}

void launch_interpolate_values_vm(uint32_t D, uint32_t n, uint32_t Fd, const uint32_t *vi, const float *bc,
                                  const float *fieldT, float *result, hipStream_t stream) {
    if (n == 0 || Fd == 0) return;
    switch (D) {
        case 2: run_fwd_vm<2>(n, Fd, vi, bc, fieldT, result, stream); break;
        case 3: run_fwd_vm<3>(n, Fd, vi, bc, fieldT, result, stream); break;
        case 4: run_fwd_vm<4>(n, Fd, vi, bc, fieldT, result, stream); break;
        case 6: run_fwd_vm<6>(n, Fd, vi, bc, fieldT, result, stream); break;
        default: throw Error("Unsupported interpolation dimension with value " + std::to_string(D));
    }
}

void launch_interpolate_values_backward_vm(uint32_t D, uint32_t n, uint32_t Fd, const uint32_t *vi, const float *bc,
                                           const float *grad_rows, float *gradT, hipStream_t stream) {
    if (n == 0 || Fd == 0) return;
    switch (D) {
        case 2: run_bwd_vm<2>(n, Fd, vi, bc, grad_rows, gradT, stream); break;
        case 3: run_bwd_vm<3>(n, Fd, vi, bc, grad_rows, gradT, stream); break;
        case 4: run_bwd_vm<4>(n, Fd, vi, bc, grad_rows, gradT, stream); break;
        case 6: run_bwd_vm<6>(n, Fd, vi, bc, grad_rows, gradT, stream); break;
        default: throw Error("Unsupported interpolation dimension with value " + std::to_string(D));
    }
}

void launch_interpolate_values_backward(uint32_t D, uint32_t V, uint32_t n, uint32_t Fd, const uint32_t *vi,
                                        const float *bc, const float *grad_in, bool rows_major, float *field_grad,
                                        hipStream_t stream) {
    if (Fd == 0 || V == 0) return;
    switch (D) {
        case 2: run_bwd<2>(V, n, Fd, vi, bc, grad_in, rows_major, field_grad, stream); break;
        case 3: run_bwd<3>(V, n, Fd, vi, bc, grad_in, rows_major, field_grad, stream); break;
        case 4: run_bwd<4>(V, n, Fd, vi, bc, grad_in, rows_major, field_grad, stream); break;
        case 6: run_bwd<6>(V, n, Fd, vi, bc, grad_in, rows_major, field_grad, stream); break;
        default: throw Error("Unsupported interpolation dimension with value " + std::to_string(D));
    }
}

}  // namespace tn
