// tn_render_rays.hip -- everything between trace_rays and the frame as ONE persistent launch:
//   coarse sampler (model.py:531-557, 111-192) -> sample / segment matching (find_visited_cells, src/tetrahedra_tracer.cu:115-160)
//   -> barycentric gather (interpolate_values, :195-221) + mlp_base + density head (model.py:577-581) -> get_weights (:582)
//   -> PDF sampler (:582-586) -> matching -> gather + mlp_base + heads (:602-621) -> weights + RGB / accumulation / median depth
//   renderers, scattered into the frame (:632-662)
// on the trace rows of the hitting rays IN PLACE, for a ray set whose SIZE lives on the device (tn_compact_hits): nothing on the
// host waits for the trace, and a render is trace_rays + compaction + this launch.
//
// Round 2-4's one-launch pass (tn_render.hip) interleaved the match and the composite with the MFMA layers of every
// 256-sample step; it was 4-6 % slower than the chain of separate kernels, because whatever one wave does between two layer
// barriers (the composite scan of the step's ray, the staging of its segments) is waited for by the seven others.  This kernel
// keeps the three kinds of work APART IN TIME inside one launch instead:
//   * a block (8 waves, one per CU, persistent) owns a contiguous range of hitting rays and works on it in TILES of up to T rays;
//   * per tile, the per-ray stages (samplers, matcher, composite) run as wave-per-ray phases on all 8 waves -- each wave owns
//     the rays t = wave, wave + 8, ... of the tile in EVERY phase, so consecutive ray phases need no block barrier -- and hand
//     their results over through a per-block scratch area in global memory (44 B per sample, re-used tile after tile);
//   * between them the MLP phases run mlp_forward_group (tn_mlp_fwd.h) -- the very loop body of k_mlp_forward -- over the
//     tile's samples as one contiguous stream: no instruction of another stage between two MFMA layers.
// The ray phases cost a few dependent round trips per ray (~1 % of a tile's time, measured in DESIGN.md section 4.5b); the
// kernel chain pays the same stages as separate launches over the whole chunk plus its intermediates' allocations.
// Every stage is the SAME device function the stand-alone kernels call (tn_ray_ops.h, tn_mlp_fwd.h; the matcher restates
// k_find_matched's expressions), so the frame is bit-identical to the kernel chain's (tests/test_render_gpu.py).
#include <cstdlib>

#include "tn_mlp_fwd.h"
#include "tn_mlp_x3_fwd.h"
#include "tn_ray_ops.h"

namespace tn {

using namespace mlp;
using namespace rayops;

namespace {

// Run-time diagnostics that CHANGE RESULTS (phase skipping) or add work (phase clocks) exist only in a diagnostic build
// (make CXXFLAGS+=-DTN_RENDER_DIAG=1, as profiles/r05q_mlp_phase_only.py / r05n_prof_overlap.sh need): the product kernel
// carries neither the tests nor the environment variables.
#ifndef TN_RENDER_DIAG
#define TN_RENDER_DIAG 0
#endif
constexpr bool DIAG = TN_RENDER_DIAG != 0;
// first float of the coarse edges in a wave's LDS region during ray phase 1: behind the matcher's tin / pmax [2 M] AND behind the
// 28 floats ray_dir_encoding writes at the start of the region (M = 4 / 8: 2 M < 28 -- the encoding used to overwrite the edges)
__host__ __device__ inline size_t phase1_edges_offset(uint32_t M) { return 2 * (size_t)M > 28 ? 2 * (size_t)M : 28; }

struct RenderRaysParams {
    uint32_t skip;   // DIAG builds only (TETRANERF_HIP_RENDER_SKIP): bit 0 / 1 / 2 = leave out ray phase 1 / 2 / 3 -- the MLP phases then run on the
                     // sample placement the previous launch left in the scratch: their time alone (profiles/r05q_mlp_phase_only.py)

    // trace rows (outputs of tn_trace_rays, read in place)
    const uint32_t *num_visited;   // [R_all]
    const float *dist;             // [R_all, M, 2]
    const float *bary;             // [R_all, M, 2, 3]
    const uint32_t *verts;         // [R_all, M, 4]
    uint32_t M;
    const uint32_t *ray_index;     // [r_max] hitting rays first (tn_compact_hits order)
    const uint32_t *count;         // device-side number of hitting rays (null: r_max)
    size_t r_max;
    uint32_t S, S_fine;            // coarse samples; fine samples added by the PDF sampler (0: one pass)
    int biased;
    const float *lin;              // [S + 1] linspace(0, 1, S + 1)
    const float *u_table;          // [S_fine + 1] bin-centred quantiles (S_fine > 0)
    float hist_pad, eps;
    const float *fieldT;           // [V, 64]
    const float *dirs;             // [R_all, 3]
    const float *ray_bias;         // [R_all, 128] or null
    const float *wenc;             // [128][28] the direction encoding's columns of mlp_head (head_ray_term)
    const float *pk;               // packed weights (gather order)
    const uint4 *blob;             // bf16x3 mode (round 6): k_mlp_pack_x3's weight pieces
    Background bg;
    float *out_rgb, *out_acc, *out_depth;   // [R_all, 3], [R_all], [R_all]: written at the ray's own row
    // per-block scratch (global memory), offsets in floats
    float *scratch;
    size_t per_block;
    uint32_t T;                    // tile capacity in rays
    size_t o_edges_f, o_hterm, o_vi, o_bc, o_sigma, o_rgb, o_enc;   // (edges_c at 0; o_enc: bf16x3 mode, [T][32] direction encodings)
    uint32_t region;               // floats of LDS per wave for the ray phases
    unsigned long long *prof;      // [8] DIAG builds only (TETRANERF_HIP_RENDER_PROFILE=1): 100 MHz ticks per phase kind, summed over blocks
};

// one ray's samples (the bin centres of e[0 .. S]) against its segments: vi [S] x 4 ids, bc [S] x 3 weights.  The expressions
// are k_find_matched's (tn_match.hip); only vertex ids and barycentrics are produced (the MLP kernel reads nothing else).
// tin / pmax: 2 M floats of LDS owned by the wave.
// n = num_visited[src] (the kernel loads the counts of a whole tile at once: no dependent load here); dv0: the ray's first 512
// segment bounds as load_bounds() requested them (the caller issues that ahead of the sampler / composite of the same ray, so
// that the rows are back when the matcher starts); e: global or LDS.
__device__ __forceinline__ void load_bounds(uint32_t M, size_t src, uint32_t n, const RenderRaysParams &p, float2 (&dv)[8], int lane) {
    if (n > M) n = M;
    const float2 *drow = reinterpret_cast<const float2 *>(p.dist) + src * M;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const uint32_t j = 64u * c + lane;
        dv[c] = make_float2(0.f, -INFINITY);
        if (j < n) dv[c] = drow[j];
    }
}

template <int UM>
__device__ __forceinline__ void ray_match(uint32_t S, uint32_t M, size_t src, uint32_t n, const RenderRaysParams &p, const float *e,
                                          uint32_t *__restrict__ vi_out, float *__restrict__ bc_out, float *tin, float *pmax, int lane,
                                          const float2 (&dv0)[8]) {
    if (n > M) n = M;
    const float2 *drow = reinterpret_cast<const float2 *>(p.dist) + src * M;
    // do the sample distances ascend?  (distance j = centre of bin j, as the callers of find_visited_cells compute it.)  The
    // loads of a whole group of 64 * UM samples are requested together, ahead of the bounds': the wave owns its ray alone and
    // every dependent round trip is exposed.
    bool bad = false;
    for (uint32_t base = 0; base + 1 < S; base += 64 * UM) {
        float a0[UM], a1[UM], a2[UM];
#pragma unroll
        for (int u = 0; u < UM; ++u) {
            const uint32_t j = base + 64 * u + lane;
            a0[u] = a1[u] = a2[u] = 0.f;
            if (j + 1 < S) { a0[u] = e[j]; a1[u] = e[j + 1]; a2[u] = e[j + 2]; }
        }
#pragma unroll
        for (int u = 0; u < UM; ++u) {
            const uint32_t j = base + 64 * u + lane;
            if (j + 1 < S) bad |= !((a1[u] + a0[u]) / 2.0f <= (a2[u] + a1[u]) / 2.0f);
        }
    }
    // stage bounds + inclusive running max of t_out (wave scans over chunks of 64; the rows of up to 8 chunks requested at once,
    // their scans interleaved: rayops::wave_incl_max_multi)
    float carry = -INFINITY;
    for (uint32_t base0 = 0; base0 < n; base0 += 512) {
        float2 dv[8];
        float mx[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const uint32_t j = base0 + 64u * c + lane;
            dv[c] = dv0[c];
            if (base0) { dv[c] = make_float2(0.f, -INFINITY); if (j < n) dv[c] = drow[j]; }   // (rays with more than 512 segments)
            mx[c] = dv[c].y;
        }
        wave_incl_max_multi<8>(mx, lane);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const uint32_t j = base0 + 64u * c + lane;
            const float m = fmaxf(mx[c], carry);
            if (j < n) { tin[j] = dv[c].x; pmax[j] = m; }
            carry = __shfl(m, 63);
        }
    }
    const bool ascending = (__ballot(bad) == 0ull);
    lds_sync();

    if (ascending) {
        uint32_t top = 1;                       // largest power of two <= n (0 for n == 0)
        while ((top << 1) <= n && (top << 1) != 0) top <<= 1;
        if (n == 0) top = 0;
        for (uint32_t base = 0; base < S; base += 64 * UM) {
            float cur[UM];
            uint32_t pp[UM];
#pragma unroll
            for (int u = 0; u < UM; ++u) {
                const uint32_t j = base + 64 * u + lane;
                cur[u] = j < S ? (e[j + 1] + e[j]) / 2.0f : 0.f;
                pp[u] = 0;
            }
            // pp = number of segments whose running-max t_out is below the sample = first pp with pmax[pp] >= cur.  Straight-line
            // (clamped reads + selects): the UM reads of a step are issued together
            const uint32_t nlast = n ? n - 1 : 0;
            for (uint32_t bit = top; bit > 0; bit >>= 1) {
                float pv[UM];
#pragma unroll
                for (int u = 0; u < UM; ++u) { const uint32_t k = pp[u] + bit - 1; pv[u] = pmax[k < nlast ? k : nlast]; }
#pragma unroll
                for (int u = 0; u < UM; ++u) pp[u] = (pp[u] + bit <= n && pv[u] < cur[u]) ? pp[u] + bit : pp[u];
            }
            bool mk[UM];
            uint4 vv[UM];
            float t_in[UM], t_out[UM];
            float2 q0[UM], q1[UM], q2[UM];
            float tv[UM];
#pragma unroll
            for (int u = 0; u < UM; ++u) tv[u] = tin[pp[u] < nlast ? pp[u] : nlast];
#pragma unroll
            for (int u = 0; u < UM; ++u) {
                const uint32_t j = base + 64 * u + lane;
                mk[u] = false; vv[u] = make_uint4(TN_EMPTY, TN_EMPTY, TN_EMPTY, TN_EMPTY);
                t_in[u] = 0.f; t_out[u] = 1.f; q0[u] = q1[u] = q2[u] = make_float2(0.f, 0.f);
                if (j < S && pp[u] < n && tv[u] <= cur[u]) {
                    const size_t g = src * M + pp[u];
                    mk[u] = true;
                    t_in[u] = tv[u]; t_out[u] = drow[pp[u]].y;
                    vv[u] = *reinterpret_cast<const uint4 *>(p.verts + 4 * g);
                    const float2 *bp = reinterpret_cast<const float2 *>(p.bary + 6 * g);
                    q0[u] = bp[0]; q1[u] = bp[1]; q2[u] = bp[2];  // c1.xyz = q0.x q0.y q1.x ; c2.xyz = q1.y q2.x q2.y
                }
            }
#pragma unroll
            for (int u = 0; u < UM; ++u) {
                const uint32_t j = base + 64 * u + lane;
                if (j >= S) continue;
                float b0 = 0.f, b1 = 0.f, b2 = 0.f;
                if (mk[u]) {
                    const float mult = (cur[u] - t_in[u]) / (t_out[u] - t_in[u]);
                    b0 = (1 - mult) * q0[u].x + mult * q1[u].y;
                    b1 = (1 - mult) * q0[u].y + mult * q2[u].x;
                    b2 = (1 - mult) * q1[u].x + mult * q2[u].y;
                }
                *reinterpret_cast<uint4 *>(vi_out + 4 * (size_t)j) = vv[u];
                bc_out[3 * (size_t)j] = b0; bc_out[3 * (size_t)j + 1] = b1; bc_out[3 * (size_t)j + 2] = b2;
            }
        }
    } else {
        // defaults everywhere, then the literal pointer walk on lane 0 (src/tetrahedra_tracer.cu:129-160)
        for (uint32_t j = lane; j < S; j += 64) {
            *reinterpret_cast<uint4 *>(vi_out + 4 * (size_t)j) = make_uint4(TN_EMPTY, TN_EMPTY, TN_EMPTY, TN_EMPTY);
            bc_out[3 * (size_t)j] = 0.f; bc_out[3 * (size_t)j + 1] = 0.f; bc_out[3 * (size_t)j + 2] = 0.f;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            uint32_t pos = 0;
            for (uint32_t j = 0; j < S; ++j) {
                const float cur = (e[j + 1] + e[j]) / 2.0f;
                while (pos < n && drow[pos].y < cur) pos++;
                if (pos >= n) break;
                const float2 hd = drow[pos];
                if (hd.x <= cur) {
                    const size_t g = src * M + pos;
                    for (int k = 0; k < 4; ++k) vi_out[4 * (size_t)j + k] = p.verts[4 * g + k];
                    const float mult = (cur - hd.x) / (hd.y - hd.x);
                    for (int k = 0; k < 3; ++k)
                        bc_out[3 * (size_t)j + k] = (1 - mult) * p.bary[6 * g + k] + mult * p.bary[6 * g + 3 + k];
                }
            }
        }
    }
    lds_sync();
}

// direction encoding of one ray (k_dir_encoding's expressions, lane-parallel): NeRFEncoding(3, 4 freqs 2^linspace(0,4,4), include_input)
__device__ __forceinline__ void ray_dir_encoding(const float *__restrict__ d3, float *__restrict__ e, int lane) {
    const float two_pi = 6.283185307179586f, half_pi = 1.5707963267948966f;
    const float freqs[4] = {1.0f, 2.5198421478271484f, 6.349603652954102f, 16.0f};  // fp32(2**(4*i/3))
    if (lane < 12) {
        const int c = lane >> 2, f = lane & 3;
        const float x = two_pi * d3[c];
        const float s = x * (f == 0 ? freqs[0] : (f == 1 ? freqs[1] : (f == 2 ? freqs[2] : freqs[3])));
        e[c * 4 + f] = sinf(s);
        e[12 + c * 4 + f] = sinf(s + half_pi);
    } else if (lane < 15) {
        e[24 + (lane - 12)] = d3[lane - 12];
    } else if (lane == 15) {
        e[27] = 0.f;
    }
}

// global-memory hand-over between two ray phases of the SAME wave (a lane reads what another lane of its wave wrote)
__device__ __forceinline__ void wave_global_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

}  // namespace

// One 8-wave block per CU, like k_mlp_forward.  Measured and dropped (profiles/r05h_render_ab.txt): 4-wave blocks, two per CU
// (the staged head layer is 68 KB since the encoding's columns left it), started half a tile apart so that one block's ray
// phases fall into the other one's MLP phases -- the MLP phases of 4-wave blocks (128-sample groups: twice the weight staging
// and barriers per sample) lost more than the overlap gained: 3.6 % behind the kernel chain instead of 1.3 %.
constexpr int RR_BLOCK = MLP_BLOCK;

// X3 (round 6): the MLP phases in the bf16x3 arithmetic (x3::forward_group, the loop body of k_mlp_forward_x3: same bits as the
// bf16x3 kernel chain).  The head layer of that arithmetic takes the direction encoding as two k-steps of its GEMM, not as a
// per-ray term, so ray phase 1 leaves the ray's 32-float encoding (and its appearance bias row, if any) in the tile's scratch.
template <bool FINE, bool X3>
__global__ __launch_bounds__(RR_BLOCK, 2) void k_render_rays(RenderRaysParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *lds = reinterpret_cast<float *>(smem);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NW = RR_BLOCK / 64;
    const size_t r = p.count ? (size_t)*p.count : p.r_max;
    // Block b owns the hitting rays b, b + G, b + 2 G, ... (G = gridDim.x), its i-th ray = entry i G + b of the list: the
    // blocks advance at the same pace, so at any moment the whole chip works on a window of ~G consecutive hitting rays -- the
    // neighbouring image pixels whose tetrahedra share vertices -- as k_mlp_forward's grid-stride over 256-sample groups does.
    // (Measured against contiguous per-block ranges, the first version: no difference on the bench frame, whose 3.8 MB field
    // every XCD's L2 holds; kept for meshes beyond the L2s.)
    const size_t G = gridDim.x, blk = blockIdx.x;
    if (r <= blk) return;                                   // block-uniform
    const size_t q0 = 0, q1 = (r - blk + G - 1) / G;        // local ray indices of this block
    const uint32_t S = p.S, M = p.M;
    const uint32_t nb = p.S_fine + 1;
    const uint32_t Sf = FINE ? S + nb : S;                  // samples of the final pass
    float *sc = p.scratch + (size_t)blockIdx.x * p.per_block;
    float *edges_c = sc, *edges_f = sc + p.o_edges_f, *hterm = sc + p.o_hterm;
    uint32_t *vi = reinterpret_cast<uint32_t *>(sc + p.o_vi);
    float *bc = sc + p.o_bc, *sigma = sc + p.o_sigma, *rgb = sc + p.o_rgb;
    float *enc_t = sc + p.o_enc;                            // X3: [T][32]
    float *wl = lds + (size_t)wave * p.region;              // this wave's LDS for the ray phases (aliases the weight stage)

    const uint32_t nrays = (uint32_t)(q1 - q0);
    const uint32_t ntiles = (nrays + p.T - 1) / p.T;
    const uint32_t tile = (nrays + ntiles - 1) / ntiles;    // even tiles: one partial MLP group per tile at most
    constexpr size_t GROUP = (size_t)NW * 32;

    // phase profile (debug): thread 0 adds the ticks of a phase straight to the global counters when it ends; nothing is kept
    // in registers across the MLP phases (per-thread accumulators cost 14 VGPRs alive through the whole kernel: spills)
    unsigned long long t_prev = 0;
    if constexpr (DIAG) if (p.prof && threadIdx.x == 0) { t_prev = wall_clock64(); atomicAdd(&p.prof[5], 1ull); }
    auto tick = [&](int k) {      // phase k ends here (nothing in the product build)
        if constexpr (DIAG) if (p.prof && threadIdx.x == 0) { const unsigned long long t = wall_clock64(); atomicAdd(&p.prof[k], t - t_prev); t_prev = t; }
    };
    const uint32_t skip = DIAG ? p.skip : 0u;
    // per-wave LDS for ray phase 2: [coarse weights: S floats][PDF sampler], re-used by the matcher afterwards
    const uint32_t w_floats = (S + 3u) & ~3u;
    for (size_t tq = q0; tq < q1;) {
        const uint32_t nt = (uint32_t)(q1 - tq < tile ? q1 - tq : tile);
        // This wave's rays of the tile are t = wave + NW i, i = 0, 1, ...: lane i holds ray i's row index, segment count and --
        // after phase 1 -- near / far, for all three ray phases (T <= 64 NW): a ray phase starts without a dependent load chain
        // (ray id -> count -> rows), which a wave that owns its ray alone would pay in full, three times per ray.
        uint32_t l_ray = 0, l_nv = 0;
        float l_near = 0.f, l_far = 1.f;
        if ((uint32_t)wave + (uint32_t)NW * lane < nt) {
            l_ray = p.ray_index[(tq + wave + (size_t)NW * lane) * G + blk];
            l_nv = p.num_visited[l_ray];
        }
        if ((uint32_t)wave + (uint32_t)NW * lane < nt) {     // (after l_nv: near / far as ray_sample_coarse reads them)
            const float2 *row = reinterpret_cast<const float2 *>(p.dist) + (size_t)l_ray * M;
            l_near = l_nv ? row[0].x : 0.0f;
            l_far = l_nv ? row[l_nv - 1].y : 1.0f;
        }
        // ---- ray phase 1: coarse sampler -> matcher (+ the head layer's per-ray term) of this wave's rays.  The edges reach
        //      the matcher through the wave's LDS (and global memory for the later phases): no store -> load round trip
        float *el = wl + phase1_edges_offset(M);             // [S + 1] behind the matcher's tin / pmax and the direction encoding
        if (!(skip & 1u)) for (uint32_t t = wave, i = 0; t < nt; t += NW, ++i) {
            const size_t ray = (uint32_t)__builtin_amdgcn_readlane((int)l_ray, (int)i);
            const uint32_t nv = (uint32_t)__builtin_amdgcn_readlane((int)l_nv, (int)i);
            const float near = __shfl(l_near, (int)i), far = __shfl(l_far, (int)i);
            float2 dv[8];
            load_bounds(M, ray, nv, p, dv, lane);            // back by the time the matcher wants them
            float *e = edges_c + (size_t)t * (S + 1);
            ray_sample_coarse_nf(S, M, ray, nv, near, far, p.dist, p.lin, nullptr, p.biased, e, el, wl, lane);
            // the head layer's per-ray term: Wh[:, :27] enc(dir) + the appearance embedding's bias (k_head_ray_term's expression)
            ray_dir_encoding(p.dirs + 3 * ray, wl, lane);   // (28 floats of the wave's LDS: the sampler is done with `cum`)
            lds_sync();
            if constexpr (X3) {
                // the encoding itself, padded to 32 (k_dir_encoding32's layout), and the ray's bias row
                if (lane < 32) enc_t[(size_t)t * 32 + lane] = lane < 27 ? wl[lane] : 0.f;
                if (p.ray_bias) {
                    hterm[(size_t)t * HID + lane] = p.ray_bias[ray * HID + lane];
                    hterm[(size_t)t * HID + 64 + lane] = p.ray_bias[ray * HID + 64 + lane];
                }
            } else {
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int o = lane + 64 * half;
                    float tv = head_ray_term(p.wenc + o * ENC_PAD, wl);
                    if (p.ray_bias) tv += p.ray_bias[ray * HID + o];
                    hterm[(size_t)t * HID + o] = tv;
                }
            }
            lds_sync();
            if (S <= 256) ray_match<4>(S, M, ray, nv, p, el, vi + 4 * (size_t)t * S, bc + 3 * (size_t)t * S, wl, wl + M, lane, dv);
            else ray_match<9>(S, M, ray, nv, p, el, vi + 4 * (size_t)t * S, bc + 3 * (size_t)t * S, wl, wl + M, lane, dv);
        }
        __syncthreads();
        tick(0);
        if constexpr (FINE) {
            // ---- MLP phase 1: density only over the tile's nt * S coarse samples
            {
                const size_t n = (size_t)nt * S, ngroups = (n + GROUP - 1) / GROUP;
                for (size_t g = 0; g < ngroups; ++g)
if constexpr (X3) x3::forward_group<true, true>(reinterpret_cast<uint4 *>(lds), g, n, S, nullptr, vi, bc, p.fieldT, nullptr, p.blob, sigma, nullptr, nullptr);
                    else                     mlp_forward_group<true, true, RR_BLOCK, false>(lds, g, n, S, nullptr, vi, bc, p.fieldT, nullptr, p.pk, sigma, nullptr,
                                                                   FwdSave{});
            }
            __syncthreads();
            tick(1);
            // ---- ray phase 2: get_weights -> PDF sampler -> matcher of the fine samples.  The coarse weights go from the
            //      composite to the sampler through the wave's LDS (the kernel chain writes and re-reads them in HBM)
            // LDS of the wave: [coarse weights S | PDF sampler] then -- the matcher re-uses that part for tin / pmax -- the merged
            // fine edges behind both
            const size_t o_merged = std::max<size_t>(2 * (size_t)M, w_floats + pdf_lds_floats(S, nb));
            const bool lds_edges = pdf_writes_second_copy(S, nb);
            if (!(skip & 2u)) for (uint32_t t = wave, i = 0; t < nt; t += NW, ++i) {
                const size_t ray = (uint32_t)__builtin_amdgcn_readlane((int)l_ray, (int)i);
                const uint32_t nv = (uint32_t)__builtin_amdgcn_readlane((int)l_nv, (int)i);
                const float near = __shfl(l_near, (int)i), far = __shfl(l_far, (int)i);
                float2 dv[8];
                load_bounds(M, ray, nv, p, dv, lane);        // in flight during the composite and the sampler
                const float *e = edges_c + (size_t)t * (S + 1);
                ray_composite(S, sigma + (size_t)t * S, nullptr, e, p.bg, nullptr, nullptr, nullptr, wl, lane);
                lds_sync();
                float *ef = edges_f + (size_t)t * (Sf + 1);
                float *efl = wl + o_merged;
                ray_sample_pdf(S, nb, e, wl, near, far, p.u_table, nullptr, p.hist_pad, p.eps, ef, wl + w_floats, lane, efl);
                const float *em = efl;
                if (!lds_edges) { wave_global_sync(); em = ef; }
                if (Sf <= 256) ray_match<4>(Sf, M, ray, nv, p, em, vi + 4 * (size_t)t * Sf, bc + 3 * (size_t)t * Sf, wl, wl + M, lane, dv);
                else ray_match<9>(Sf, M, ray, nv, p, em, vi + 4 * (size_t)t * Sf, bc + 3 * (size_t)t * Sf, wl, wl + M, lane, dv);
            }
            __syncthreads();
            tick(2);
        }
        // ---- MLP phase 2: the whole network over the tile's nt * Sf samples
        {
            const size_t n = (size_t)nt * Sf, ngroups = (n + GROUP - 1) / GROUP;
            for (size_t g = 0; g < ngroups; ++g)
                if constexpr (X3) x3::forward_group<true, false>(reinterpret_cast<uint4 *>(lds), g, n, Sf, nullptr, vi, bc, p.fieldT, enc_t, p.blob, sigma, rgb,
                                                                 p.ray_bias ? hterm : nullptr);
                else mlp_forward_group<true, false, RR_BLOCK, false>(lds, g, n, Sf, nullptr, vi, bc, p.fieldT, hterm, p.pk, sigma, rgb, FwdSave{});
        }
        __syncthreads();
        tick(3);
        // ---- ray phase 3: weights + renderers, scattered into the frame
        const float *ee = FINE ? edges_f : edges_c;
        if (!(skip & 4u)) for (uint32_t t = wave, i = 0; t < nt; t += NW, ++i) {
            const size_t ray = (uint32_t)__builtin_amdgcn_readlane((int)l_ray, (int)i);
            ray_composite(Sf, sigma + (size_t)t * Sf, rgb + 3 * (size_t)t * Sf, ee + (size_t)t * (Sf + 1), p.bg, p.out_rgb + 3 * ray,
                          p.out_acc + ray, p.out_depth + ray, nullptr, lane);
        }
        tick(4);
        tq += nt;
        // (the next tile's first ray phase writes edges_c / hterm / vi / bc: all of them last read before the barrier above;
        //  its LDS use starts after this wave's own composite; sigma / rgb are next written after two more barriers)
    }
}

size_t render_rays_scratch_floats(size_t r_max, uint32_t S, uint32_t S_fine, bool has_bias, unsigned grid, RenderRaysLayout &L) {
    auto al = [](size_t x) { return (x + 63) & ~(size_t)63; };   // 256-byte aligned pieces (and blocks): a group's 4 KB of vertex ids start on a line
    const uint32_t nb = S_fine + 1;
    const uint32_t Sf = S_fine ? S + nb : S;
    const size_t per_ray = (size_t)(S + 1) + (S_fine ? (size_t)(Sf + 1) : 0) + HID + 32 + (size_t)Sf * 11;
    const size_t rays_per_block = (r_max + grid - 1) / grid;
    // Scratch budget per block = tile size.  Measured (profiles/r05k_scratch_sweep.txt): a chip-wide working set of 256 x 4 MB
    // costs the coarse-only render 1.4 % against 256 x 0.5 MB (address translation: 256 private windows), but small tiles end in
    // a partial MLP group each -- free when the samples per ray are a multiple of the group (256: the coarse-only pass), 0.4 % at
    // 78 rays x 513 samples, 1.1 % at 39: hence 512 KB where tiles cannot end in a partial group, 4 MB otherwise.
    // TETRANERF_HIP_RENDER_SCRATCH_KB overrides (sweeps).
    static const size_t env_kb = [] { const char *v = std::getenv("TETRANERF_HIP_RENDER_SCRATCH_KB"); return v && *v ? (size_t)std::atol(v) : (size_t)0; }();
    const size_t budget_kb = env_kb ? env_kb : ((Sf % 256u == 0 && (!S_fine || S % 256u == 0)) ? (size_t)512 : (size_t)4096);
    size_t T = (budget_kb << 10) / (per_ray * sizeof(float));
    if (T < 8) T = 8;
    if (T > 64 * (RR_BLOCK / 64)) T = 64 * (RR_BLOCK / 64);        // a wave keeps its rays' ids in one register, lane i = ray i
    if (T > rays_per_block) T = rays_per_block;
    if (T < 1) T = 1;
    L.T = (uint32_t)T;
    size_t o = al(T * (S + 1));
    L.o_edges_f = o; o = al(o + (S_fine ? T * (Sf + 1) : 0));
    L.o_hterm = o; o = al(o + T * HID);
    L.o_vi = o; o = al(o + T * Sf * 4);
    L.o_bc = o; o = al(o + T * Sf * 3);
    L.o_sigma = o; o = al(o + T * Sf);
    L.o_rgb = o; o = al(o + T * Sf * 3);
    L.o_enc = o; o = al(o + T * 32);                      // (bf16x3 mode)
    L.per_block = o;
    return o * grid;
}

void launch_render_rays(const uint32_t *num_visited, const float *dist, const float *bary, const uint32_t *verts, uint32_t M,
                        const uint32_t *ray_index, const uint32_t *count, size_t r_max, uint32_t S, uint32_t S_fine, bool biased,
                        const float *lin, const float *u_table, float hist_pad, float eps, const float *fieldT, const float *dirs,
                        const float *ray_bias, const MlpPacks &w, Background background, float *out_rgb, float *out_acc, float *out_depth,
                        float *scratch, const RenderRaysLayout &L, unsigned grid, hipStream_t stream, unsigned long long *prof, int mode) {
    if (r_max == 0) return;
    const uint32_t nb = S_fine + 1;
    // per wave: phase 1 = [tin / pmax (or the biased sampler's cum): 2 M][coarse edges: S + 1]; phase 2 = [coarse weights |
    // PDF sampler, re-used as tin / pmax][merged fine edges: S + nb + 1]
    const size_t w_fl = ((size_t)S + 3) & ~(size_t)3;
    const size_t region = (std::max<size_t>(phase1_edges_offset(M) + (S + 1), S_fine ? std::max<size_t>(2 * (size_t)M, w_fl + pdf_lds_floats(S, nb)) +
                                                                                  (size_t)S + nb + 1 : 0) + 3) & ~(size_t)3;
    const size_t lds_floats = std::max<size_t>(mode ? 4 * x3::MAX_STAGE_U4 : MAX_STAGE_FLOATS, (RR_BLOCK / 64) * region);
    const size_t smem = lds_floats * sizeof(float);
    if (smem > 160 * 1024) throw Error("render_rays: max_ray_triangles / samples per ray too large for the per-wave LDS regions");
    RenderRaysParams p{};
    if constexpr (DIAG) { const char *v = std::getenv("TETRANERF_HIP_RENDER_SKIP"); p.skip = v && *v ? (uint32_t)std::atoi(v) : 0u; }
    p.num_visited = num_visited; p.dist = dist; p.bary = bary; p.verts = verts; p.M = M;
    p.ray_index = ray_index; p.count = count; p.r_max = r_max;
    p.S = S; p.S_fine = S_fine; p.biased = biased ? 1 : 0;
    p.lin = lin; p.u_table = u_table; p.hist_pad = hist_pad; p.eps = eps;
    p.fieldT = fieldT; p.dirs = dirs; p.ray_bias = ray_bias; p.wenc = w.wenc; p.pk = w.pk_gather; p.bg = background;
    p.out_rgb = out_rgb; p.out_acc = out_acc; p.out_depth = out_depth;
    p.scratch = scratch; p.per_block = L.per_block; p.T = L.T;
    p.o_edges_f = L.o_edges_f; p.o_hterm = L.o_hterm; p.o_vi = L.o_vi; p.o_bc = L.o_bc;
    p.o_sigma = L.o_sigma; p.o_rgb = L.o_rgb; p.o_enc = L.o_enc;
    p.blob = w.blob;
    p.region = (uint32_t)region;
    p.prof = DIAG ? prof : nullptr;
    static PerDeviceOnce lds_attr;
    lds_attr.run([&] {
        allow_dynamic_lds(reinterpret_cast<const void *>(k_render_rays<false, false>), 160 * 1024);
        allow_dynamic_lds(reinterpret_cast<const void *>(k_render_rays<true, false>), 160 * 1024);
        allow_dynamic_lds(reinterpret_cast<const void *>(k_render_rays<false, true>), 160 * 1024);
        allow_dynamic_lds(reinterpret_cast<const void *>(k_render_rays<true, true>), 160 * 1024);
    });
    if (mode) {
        if (S_fine) hipLaunchKernelGGL((k_render_rays<true, true>), dim3(grid), dim3(RR_BLOCK), smem, stream, p);
        else hipLaunchKernelGGL((k_render_rays<false, true>), dim3(grid), dim3(RR_BLOCK), smem, stream, p);
        return;
    }
    if (S_fine) hipLaunchKernelGGL((k_render_rays<true, false>), dim3(grid), dim3(RR_BLOCK), smem, stream, p);
    else hipLaunchKernelGGL((k_render_rays<false, false>), dim3(grid), dim3(RR_BLOCK), smem, stream, p);
}

}  // namespace tn
