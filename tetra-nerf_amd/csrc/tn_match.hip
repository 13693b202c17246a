// tn_match.hip -- find_visited_cells: match S sorted sample distances of a ray against its
// <= M sorted segments and lerp the entry/exit barycentrics.
//
// Replaces find_matched_cells_kernel (src/tetrahedra_tracer.cu:115-160; 16-thread blocks,
// one thread per ray, serial over samples).  Here ONE WAVEFRONT owns a ray: the segment
// bounds are staged in LDS, the per-sample two-pointer merge becomes an independent binary
// search (valid because the sample distances ascend: the sequential pointer of the reference
// equals "first segment whose t_out >= d"), and all outputs are written by
// consecutive lanes (coalesced), including the defaults the reference gets from
// torch::full/zeros (src/py_binding.cpp:188-191).  Rays whose distances do not ascend take a
// literal serial branch on lane 0.
#include "tn_ray_ops.h"

namespace tn {

// Round 5: the per-ray chain of dependent round trips (row index -> count -> bounds, chunk by chunk -> distances, twice ->
// per 256 samples: search -> gather -> store) is what the op waits for at every size.  Now: the NEXT
// ray's row index and count are requested while this one is matched; the bounds rows of up to 8 chunks and the distances of a
// whole group of 64 UM samples (with the neighbours the ascending test needs) are requested together; the chunks' running-max
// scans run interleaved (rayops::wave_incl_max_multi); the binary lifting reads LDS through clamped indices and selects, so
// the UM reads of a step overlap.  Same expressions, same outputs.
template <int UM>
__global__ __launch_bounds__(64) void k_find_matched(size_t R, uint32_t S, uint32_t M,
                                                     const uint32_t *__restrict__ num_visited,
                                                     const uint32_t *__restrict__ visited,
                                                     const float *__restrict__ dist,
                                                     const float *__restrict__ bary,
                                                     const float *__restrict__ distances,
                                                     const uint32_t *__restrict__ verts,
                                                     uint32_t *__restrict__ cells_out,
                                                     uint32_t *__restrict__ verts_out,
                                                     uint8_t *__restrict__ mask_out,
                                                     float *__restrict__ bary_out,
                                                     const uint32_t *__restrict__ ray_index, const uint32_t *__restrict__ count) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (count) R = *count;      // device-side number of (hitting) rays; the grid was sized for an upper bound
    float *tin = reinterpret_cast<float *>(smem);  // [M]
    float *pmax = tin + M;                         // [M] running max of t_out
    const int lane = threadIdx.x;

    // row of the trace outputs a sample row belongs to (ray_index: the caller kept the trace rows of ALL rays and matches a
    // subset -- no compacted copy of the 26 KB rows) and its segment count, one ray ahead
    size_t ray = blockIdx.x;
    size_t src_next = 0;
    uint32_t n_next = 0;
    if (ray < R) {
        src_next = ray_index ? (size_t)ray_index[ray] : ray;
        n_next = num_visited[src_next];
    }
    for (; ray < R; ray += gridDim.x) {
        const size_t src = src_next;
        uint32_t n = n_next;
        if (ray + gridDim.x < R) {
            src_next = ray_index ? (size_t)ray_index[ray + gridDim.x] : ray + gridDim.x;
            n_next = num_visited[src_next];
        }
        if (n > M) n = M;
        const float2 *drow = reinterpret_cast<const float2 *>(dist) + src * M;
        const float *srow = distances + ray * S;
        // do the sample distances ascend?
        bool bad = false;
        for (uint32_t base = 0; base + 1 < S; base += 64 * UM) {
            float a0[UM], a1[UM];
#pragma unroll
            for (int u = 0; u < UM; ++u) {
                const uint32_t j = base + 64 * u + lane;
                a0[u] = a1[u] = 0.f;
                if (j + 1 < S) { a0[u] = srow[j]; a1[u] = srow[j + 1]; }
            }
#pragma unroll
            for (int u = 0; u < UM; ++u) {
                const uint32_t j = base + 64 * u + lane;
                if (j + 1 < S) bad |= !(a0[u] <= a1[u]);
            }
        }
        // stage bounds + inclusive running max of t_out (wave scans, chunks of 64)
        float carry = -INFINITY;
        for (uint32_t base0 = 0; base0 < n; base0 += 512) {
            float2 dv[8];
            float mx[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const uint32_t j = base0 + 64u * c + lane;
                dv[c] = make_float2(0.f, -INFINITY);
                if (j < n) dv[c] = drow[j];
                mx[c] = dv[c].y;
            }
            rayops::wave_incl_max_multi<8>(mx, lane);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const uint32_t j = base0 + 64u * c + lane;
                const float m = fmaxf(mx[c], carry);
                if (j < n) { tin[j] = dv[c].x; pmax[j] = m; }
                carry = __shfl(m, 63);
            }
        }
        const bool ascending = (__ballot(bad) == 0ull);
        __syncthreads();

        if (ascending) {
            // UM sample chunks of 64 per iteration: their searches (LDS) and segment gathers (global) are
            // independent, so the dependent chain search -> gather -> store is paid once per 64 UM samples
            uint32_t top = 1;                       // largest power of two <= n (0 for n == 0)
            while ((top << 1) <= n && (top << 1) != 0) top <<= 1;
            if (n == 0) top = 0;
            const uint32_t nlast = n ? n - 1 : 0;
            for (uint32_t base = 0; base < S; base += 64 * UM) {
                float cur[UM];
                uint32_t p[UM];
#pragma unroll
                for (int u = 0; u < UM; ++u) {
                    const uint32_t j = base + 64 * u + lane;
                    cur[u] = j < S ? srow[j] : 0.f;
                    p[u] = 0;
                }
                // p = number of segments whose running-max t_out is below the sample = first p with pmax[p] >= cur
                for (uint32_t bit = top; bit > 0; bit >>= 1) {
                    float pv[UM];
#pragma unroll
                    for (int u = 0; u < UM; ++u) { const uint32_t k = p[u] + bit - 1; pv[u] = pmax[k < nlast ? k : nlast]; }
#pragma unroll
                    for (int u = 0; u < UM; ++u) p[u] = (p[u] + bit <= n && pv[u] < cur[u]) ? p[u] + bit : p[u];
                }
                uint8_t mk[UM];
                uint32_t cell[UM];
                uint4 vv[UM];
                float t_in[UM], t_out[UM], tv[UM];
                float2 q0[UM], q1[UM], q2[UM];
#pragma unroll
                for (int u = 0; u < UM; ++u) tv[u] = tin[p[u] < nlast ? p[u] : nlast];
#pragma unroll
                for (int u = 0; u < UM; ++u) {
                    const uint32_t j = base + 64 * u + lane;
                    mk[u] = 0; cell[u] = TN_EMPTY; vv[u] = make_uint4(TN_EMPTY, TN_EMPTY, TN_EMPTY, TN_EMPTY);
                    t_in[u] = 0.f; t_out[u] = 1.f; q0[u] = q1[u] = q2[u] = make_float2(0.f, 0.f);
                    if (j < S && p[u] < n && tv[u] <= cur[u]) {
                        const size_t g = src * M + p[u];
                        mk[u] = 1;
                        t_in[u] = tv[u]; t_out[u] = drow[p[u]].y;
                        cell[u] = visited[g];
                        vv[u] = *reinterpret_cast<const uint4 *>(verts + 4 * g);
                        const float2 *bp = reinterpret_cast<const float2 *>(bary + 6 * g);
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
                    const size_t o = ray * S + j;
                    mask_out[o] = mk[u];
                    cells_out[o] = cell[u];
                    *reinterpret_cast<uint4 *>(verts_out + 4 * o) = vv[u];
                    bary_out[3 * o] = b0; bary_out[3 * o + 1] = b1; bary_out[3 * o + 2] = b2;
                }
            }
        } else {
            // defaults everywhere, then the literal pointer walk on lane 0
            for (uint32_t j = lane; j < S; j += 64) {
                const size_t o = ray * S + j;
                mask_out[o] = 0;
                cells_out[o] = TN_EMPTY;
                *reinterpret_cast<uint4 *>(verts_out + 4 * o) = make_uint4(TN_EMPTY, TN_EMPTY, TN_EMPTY, TN_EMPTY);
                bary_out[3 * o] = 0.f; bary_out[3 * o + 1] = 0.f; bary_out[3 * o + 2] = 0.f;
            }
            __syncthreads();
            if (lane == 0) {
                uint32_t p = 0;
                for (uint32_t j = 0; j < S; ++j) {
                    const float cur = srow[j];
                    while (p < n && drow[p].y < cur) p++;
                    if (p >= n) break;
                    const float2 hd = drow[p];
                    if (hd.x <= cur) {
                        const size_t g = src * M + p, o = ray * S + j;
                        mask_out[o] = 1;
                        cells_out[o] = visited[g];
                        for (int k = 0; k < 4; ++k) verts_out[4 * o + k] = verts[4 * g + k];
                        const float mult = (cur - hd.x) / (hd.y - hd.x);
                        for (int k = 0; k < 3; ++k)
                            bary_out[3 * o + k] = (1 - mult) * bary[6 * g + k] + mult * bary[6 * g + 3 + k];
                    }
                }
            }
        }
        __syncthreads();
    }
}

void launch_find_matched_cells(size_t R, size_t S, size_t M, const uint32_t *num_visited,
                               const uint32_t *visited, const float *dist, const float *bary,
                               const float *distances, const uint32_t *verts, uint32_t *cells_out,
                               uint32_t *verts_out, uint8_t *mask_out, float *bary_out, hipStream_t stream,
                               const uint32_t *ray_index, const uint32_t *count) {
    if (R == 0 || S == 0) return;
    const size_t smem = 2 * M * sizeof(float);
    const size_t max_blocks = 256 * 32;
    const unsigned grid = (unsigned)(R < max_blocks ? R : max_blocks);
    // the samples of a ray in ONE group where they fit: 64 UM per iteration of the search / gather / store chain
    if (S <= 256)
        hipLaunchKernelGGL(k_find_matched<4>, dim3(grid), dim3(64), smem, stream, R, (uint32_t)S, (uint32_t)M,
                           num_visited, visited, dist, bary, distances, verts, cells_out, verts_out, mask_out, bary_out, ray_index, count);
    else
        hipLaunchKernelGGL(k_find_matched<5>, dim3(grid), dim3(64), smem, stream, R, (uint32_t)S, (uint32_t)M,
                           num_visited, visited, dist, bary, distances, verts, cells_out, verts_out, mask_out, bary_out, ray_index, count);
}

}  // namespace tn
