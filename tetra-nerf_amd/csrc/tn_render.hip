// tn_render.hip -- one render PASS of the Tetra-NeRF model as ONE launch:
//   sample -> segment matching (find_visited_cells, src/tetrahedra_tracer.cu:115-160)
//   -> barycentric feature gather (interpolate_values, src/tetrahedra_tracer.cu:195-221)
//   -> mlp_base + density head [+ direction encoding ++ mlp_head + rgb head] (model.py:414-455, 577-621)
//   -> RaySamples.get_weights [+ RGB over background / accumulation / median depth] (model.py:632-662)
// on the trace rows of the hitting rays IN PLACE (ray_index), from the [r, S+1] bin edges a sampler produced.
// Nothing per-sample goes through HBM except the 4-byte coarse weights the PDF sampler needs: the unfused path
// writes and re-reads 33 B (match outputs) + 16 B (sigma, rgb) per sample and launches three kernels per pass.
//
// Work decomposition = tn_mlp.hip's: a block of 8 wavefronts processes 256 consecutive samples per step, one
// wavefront = 32 samples x 2 feature halves, weights staged per layer in LDS, activations fed back from the
// accumulators (see tn_mlp.hip).  What is new around the four layers:
//   * a block owns a contiguous range of RAYS and walks their samples as one stream, so a ray never straddles two
//     blocks and the per-ray scans need no inter-block hand-over;
//   * per step the segments (t_in and the running maximum of t_out, as k_find_matched) of the <= 6 rays the step
//     touches are staged in LDS; every sample does the matcher's binary lifting there, reads its segment's vertex ids
//     and entry / exit barycentrics straight from the trace rows, lerps them and gathers its features into the
//     B-operand registers (same expression trees as tn_match.hip / tn_interp.hip).  The match of step g + 1 is
//     software-pipelined into step g (its loads ride with the weight copies of layer 3 and of the head layer), so a
//     step starts with the feature gather like tn_mlp.hip's kernel and the match adds no exposed round trip;
//   * sigma * delta (and the colour) of the samples go to an LDS exchange ring holding two steps; one wavefront per
//     ray runs the composite scan (k_composite's expressions) over the ray's 256-sample chunks (4 samples per lane) --
//     aligned to the RAY's first sample, each processed in the step that completes it, so that a ray's result does not
//     depend on where it sits in the batch -- carrying the state of the one ray that continues into the next step through an LDS slot;
//     finished rays write rgb / accumulation / depth (or every sample its weight).
// Preconditions (checked by the host entry): 64 <= S, M <= 512, bin edges non-decreasing per ray.
#include "tn_mlp_common.h"

namespace tn {

using namespace mlp;

namespace {

constexpr int RP_PIECES = 6;        // rays a step of 256 samples can touch when S >= 64
constexpr int RP_GROUP = 256;
constexpr int RP_RING = 2 * RP_GROUP;

struct RenderPassParams {
    const uint32_t *num_visited;   // [R_all]
    const float *dist;             // [R_all, M, 2]
    const float *bary;             // [R_all, M, 2, 3]
    const uint32_t *verts;         // [R_all, M, 4]
    const uint32_t *ray_index;     // [r] trace row of hitting ray q
    const uint32_t *nv_hit;        // [r] num_visited[ray_index[q]] (gathered once per launch: no dependent load chain in the pipeline)
    const float *edges;            // [r, S + 1]
    const float *fieldT;           // [V, 64]
    const float *enc;              // [r, 28] or null (density only)
    const float *ray_bias;         // [r, 128] per-ray head bias (appearance embedding) or null
    const float *pk;               // packed weights (k_mlp_pack, gather order)
    float *out_weights;            // [r, S] or null
    float *out_rgb, *out_acc, *out_depth;   // [R_all, 3], [R_all], [R_all] or null: written at ray_index[q]
    size_t r;
    uint32_t S, M;
    Background background;
};

struct RayState { float carry, acc, r0, r1, r2, depth; uint32_t found, pad; };

__global__ void k_gather_counts(size_t r, const uint32_t *__restrict__ ray_index, const uint32_t *__restrict__ num_visited,
                                uint32_t *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < r) out[i] = num_visited[ray_index[i]];
}

}  // namespace

// what a lane carries from the match of its sample to the gather: the segment's vertex ids, the sample's barycentrics
// and its bin edges
struct Matched { uint4 v4; float b0, b1, b2, e0, e1; };

template <bool DENSITY_ONLY>
__global__ __launch_bounds__(MLP_BLOCK) void k_render_pass(RenderPassParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *lds = reinterpret_cast<float *>(smem);                 // staged layer
    float *seg_t = lds + MAX_STAGE_FLOATS;                        // [RP_PIECES][M] t_in
    float *seg_p = seg_t + (size_t)RP_PIECES * p.M;               // [RP_PIECES][M] running max of t_out
    uint32_t *seg_n = reinterpret_cast<uint32_t *>(seg_p + (size_t)RP_PIECES * p.M);   // [8]
    // composite exchange: a ring of TWO steps (block-local sample index & 511), because the composite scans a ray in
    // chunks of 64 samples aligned to the RAY's first sample -- a chunk may straddle two steps and is processed in the
    // step that completes it.  Ray-aligned chunks make a ray's result independent of where the ray sits in the batch.
    float *c_dd = reinterpret_cast<float *>(seg_n + 8);           // [512] sigma * delta
    float *c_mid = c_dd + RP_RING;                                // [512] bin centre
    float *c_rgb = c_mid + RP_RING;                               // [3][512]
    RayState *open = reinterpret_cast<RayState *>(c_rgb + 3 * RP_RING);   // [2]: written by step g into [g & 1]

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5;
    const uint32_t S = p.S, M = p.M;
    const size_t q0 = p.r * blockIdx.x / gridDim.x;             // r < 2^32, gridDim <= 256: no overflow
    const size_t q1 = p.r * (blockIdx.x + 1) / gridDim.x;
    const uint32_t stream_len = (uint32_t)((q1 - q0) * S);
    const uint32_t ngroups = (stream_len + RP_GROUP - 1) / RP_GROUP;

    // The match of step g + 1 is software-pipelined into step g: its segments are staged (and its bin edges requested)
    // while layer 3's weights are in flight, its search runs and its segment records are requested while the head
    // layer's weights are in flight, the lerp consumes them after the head layer.  Step g itself then starts with the
    // feature gather, exactly like tn_mlp.hip's fused-gather kernel: the match adds no exposed round trip.
    auto step_geometry = [&](uint32_t g, uint32_t &gs, uint32_t &ge, uint32_t &qa, uint32_t &P) {
        gs = g * RP_GROUP; ge = gs + RP_GROUP < stream_len ? gs + RP_GROUP : stream_len;
        qa = gs / S; P = (ge - 1) / S - qa + 1;
    };
    auto sample_of = [&](uint32_t g, uint32_t &ql, uint32_t &j) {   // this lane's sample of step g (clamped: duplicates store nothing)
        uint32_t gs, ge, qa, P;
        step_geometry(g, gs, ge, qa, P);
        const uint32_t sl = gs + (uint32_t)wave * 32 + ((uint32_t)lane & 31u);
        const uint32_t slc = sl < ge ? sl : ge - 1;
        ql = slc / S; j = slc - ql * S;
    };
    // part A: segments of the rays of step g -> LDS (t_in, running max of t_out); this lane's bin edges requested
    auto match_stage = [&](uint32_t g, Matched &m) {
        uint32_t gs, ge, qa, P;
        step_geometry(g, gs, ge, qa, P);
        for (uint32_t piece = wave; piece < P; piece += MLP_BLOCK / 64) {
            const size_t ray = p.ray_index[q0 + qa + piece];
            uint32_t n = p.num_visited[ray];
            if (n > M) n = M;
            const float2 *drow = reinterpret_cast<const float2 *>(p.dist) + ray * M;
            float carry = -INFINITY;
            for (uint32_t base = 0; base < n; base += 64) {
                const uint32_t jj = base + lane;
                float2 d = make_float2(0.f, -INFINITY);
                if (jj < n) d = drow[jj];
                float mx = d.y;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const float o = __shfl_up(mx, off);
                    if (lane >= off) mx = fmaxf(mx, o);
                }
                mx = fmaxf(mx, carry);
                if (jj < n) { seg_t[(size_t)piece * M + jj] = d.x; seg_p[(size_t)piece * M + jj] = mx; }
                carry = __shfl(mx, 63);
            }
            if (lane == 0) seg_n[piece] = n;
        }
        uint32_t ql, j;
        sample_of(g, ql, j);
        const size_t q = q0 + ql;
        m.e0 = p.edges[q * (S + 1) + j]; m.e1 = p.edges[q * (S + 1) + j + 1];
    };
    // The same, split for the software pipeline (one ray per wavefront: a step touches <= 6 rays, a block has 8 waves):
    // the ray's row index and segment count are requested a layer early, the rows of its segment bounds (<= 8 chunks of
    // 64, M <= 512) at the start of the next layer, the scan + LDS stores run at that layer's end.
    struct SegLoad { size_t ray; uint32_t n; bool on; float2 d[8]; };
    auto seg_issue_count = [&](uint32_t g, SegLoad &sl, Matched &m) {
        uint32_t gs, ge, qa, P;
        step_geometry(g, gs, ge, qa, P);
        sl.on = (uint32_t)wave < P;
        sl.ray = 0; sl.n = 0;
        if (sl.on) { sl.ray = p.ray_index[q0 + qa + wave]; sl.n = p.nv_hit[q0 + qa + wave]; }
        uint32_t ql, j;
        sample_of(g, ql, j);
        const size_t q = q0 + ql;
        m.e0 = p.edges[q * (S + 1) + j]; m.e1 = p.edges[q * (S + 1) + j + 1];
    };
    auto seg_issue_rows = [&](SegLoad &sl) {
        if (sl.n > M) sl.n = M;
        const float2 *drow = reinterpret_cast<const float2 *>(p.dist) + sl.ray * M;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const uint32_t jj = (uint32_t)c * 64 + lane;
            sl.d[c] = make_float2(0.f, -INFINITY);
            if (sl.on && jj < sl.n) sl.d[c] = drow[jj];
        }
    };
    auto seg_store = [&](const SegLoad &sl) {
        if (!sl.on) return;                         // wave-uniform
        float carry = -INFINITY;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if ((uint32_t)c * 64 >= sl.n) break;     // wave-uniform
            const uint32_t jj = (uint32_t)c * 64 + lane;
            float mx = sl.d[c].y;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const float o = __shfl_up(mx, off);
                if (lane >= off) mx = fmaxf(mx, o);
            }
            mx = fmaxf(mx, carry);
            if (jj < sl.n) { seg_t[(size_t)wave * M + jj] = sl.d[c].x; seg_p[(size_t)wave * M + jj] = mx; }
            carry = __shfl(mx, 63);
        }
        if (lane == 0) seg_n[wave] = sl.n;
    };
    // part B (after a barrier): the matcher's binary lifting in LDS; the segment's record requested
    struct Pending { float2 q0f, q1f, q2f; float t_in, t_out, cur; bool hit; };
    auto match_search = [&](uint32_t g, Matched &m, Pending &pd) {
        uint32_t gs, ge, qa, P, ql, j;
        step_geometry(g, gs, ge, qa, P);
        sample_of(g, ql, j);
        const uint32_t piece = ql - qa;
        const uint32_t n = seg_n[piece];
        const float *pm = seg_p + (size_t)piece * M, *ti = seg_t + (size_t)piece * M;
        pd.cur = (m.e1 + m.e0) / 2.0f;
        uint32_t pos = 0;
        for (uint32_t bit = n ? (1u << (31 - __clz((int)n))) : 0u; bit > 0; bit >>= 1)
            if (pos + bit <= n && pm[pos + bit - 1] < pd.cur) pos += bit;
        m.v4 = make_uint4(TN_EMPTY, TN_EMPTY, TN_EMPTY, TN_EMPTY);
        pd.hit = pos < n && ti[pos] <= pd.cur;
        pd.t_in = 0.f; pd.t_out = 1.f; pd.q0f = pd.q1f = pd.q2f = make_float2(0.f, 0.f);
        if (pd.hit) {
            const size_t gi = (size_t)p.ray_index[q0 + ql] * M + pos;
            pd.t_in = ti[pos]; pd.t_out = p.dist[2 * gi + 1];
            m.v4 = *reinterpret_cast<const uint4 *>(p.verts + 4 * gi);
            const float2 *bp = reinterpret_cast<const float2 *>(p.bary + 6 * gi);
            pd.q0f = bp[0]; pd.q1f = bp[1]; pd.q2f = bp[2];   // c1.xyz = q0.x q0.y q1.x ; c2.xyz = q1.y q2.x q2.y
        }
    };
    // part C: lerp of the entry / exit barycentrics (tn_match.hip's expression)
    auto match_finish = [&](Matched &m, const Pending &pd) {
        m.b0 = m.b1 = m.b2 = 0.f;
        if (pd.hit) {
            const float mult = (pd.cur - pd.t_in) / (pd.t_out - pd.t_in);
            m.b0 = (1 - mult) * pd.q0f.x + mult * pd.q1f.y;
            m.b1 = (1 - mult) * pd.q0f.y + mult * pd.q2f.x;
            m.b2 = (1 - mult) * pd.q1f.x + mult * pd.q2f.y;
        }
    };

    // ---- composite of step g2: one wavefront per ray of the step; ray-aligned chunks of 256 samples (4 consecutive
    //      samples per lane: a lane-local prefix + ONE wave scan per 256 samples -- the cross-lane shuffles are what a scan
    //      costs), each chunk processed in the step that holds its last sample.  Called one step LATE, in the shadow of
    //      the next step's layer-2 weight copy (the exchange ring still holds the two steps a chunk can straddle).
    auto composite_step = [&](uint32_t g) {
        uint32_t gs, ge, qa, P;
        step_geometry(g, gs, ge, qa, P);
        for (uint32_t piece2 = wave; piece2 < P; piece2 += MLP_BLOCK / 64) {
        const uint32_t qq = qa + piece2;
        const uint32_t ray_start = qq * S, ray_end = ray_start + S;
        uint32_t k = gs > ray_start ? (gs - ray_start) / RP_GROUP : 0;   // chunks 0 .. k-1 were completed by earlier steps
        RayState st = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0u, 0u};
        if (k > 0) st = open[(g + 1) & 1];       // the one ray that continues from the previous step
        bool done = false, any = false;
        for (;; ++k) {
            const uint32_t cb = ray_start + RP_GROUP * k;
            const uint32_t ce = cb + RP_GROUP < ray_end ? cb + RP_GROUP : ray_end;
            if (ce - 1 >= ge) break;             // completed by a later step
            float dd[4], w[4];
            uint32_t li[4];
            bool ok[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t idx = cb + 4u * (uint32_t)lane + (uint32_t)i;
                ok[i] = idx < ce;
                li[i] = (ok[i] ? idx : ce - 1) & (RP_RING - 1);
                dd[i] = ok[i] ? c_dd[li[i]] : 0.f;
            }
            const float p1 = dd[0], p2 = p1 + dd[1], p3 = p2 + dd[2], tot = p3 + dd[3];
            float inc = tot;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const float o = __shfl_up(inc, off);
                if (lane >= off) inc += o;
            }
            const float ex0 = st.carry + (inc - tot);
            const float excl[4] = {ex0, ex0 + p1, ex0 + p2, ex0 + p3};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float wi = (1.0f - expf(-dd[i])) * expf(-excl[i]);
                if (!(wi == wi) || !ok[i]) wi = 0.f;   // nan_to_num
                w[i] = wi;
                if (p.out_weights && ok[i]) p.out_weights[(q0 + qq) * S + (cb - ray_start) + 4u * (uint32_t)lane + (uint32_t)i] = wi;
            }
            if constexpr (!DENSITY_ONLY) {
                float l0 = 0.f, l1 = 0.f, l2 = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float k0 = c_rgb[li[i]], k1 = c_rgb[RP_RING + li[i]], k2 = c_rgb[2 * RP_RING + li[i]];
                    if (p.background.clamp) { k0 = nan_to_num(k0); k1 = nan_to_num(k1); k2 = nan_to_num(k2); }
                    l0 += w[i] * k0; l1 += w[i] * k1; l2 += w[i] * k2;
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) { l0 += __shfl_xor(l0, off); l1 += __shfl_xor(l1, off); l2 += __shfl_xor(l2, off); }
                st.r0 += l0; st.r1 += l1; st.r2 += l2;
            }
            const float c1 = w[0], c2 = c1 + w[1], c3 = c2 + w[2], wtot = c3 + w[3];
            float winc = wtot;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const float o = __shfl_up(winc, off);
                if (lane >= off) winc += o;
            }
            const float cum0 = st.acc + (winc - wtot);
            const float cum[4] = {cum0 + c1, cum0 + c2, cum0 + c3, cum0 + wtot};
            // median depth: the first sample whose cumulative weight reaches 0.5
            float mymid = 0.f;
            bool mine = false;
#pragma unroll
            for (int i = 3; i >= 0; --i)
                if (ok[i] && cum[i] >= 0.5f) { mine = true; mymid = c_mid[li[i]]; }
            const uint64_t m = __ballot(mine);
            if (!st.found && m) {
                const int src = __ffsll((unsigned long long)m) - 1;
                st.depth = __shfl(mymid, src);
                st.found = 1u;
            }
            st.carry += __shfl(inc, 63);
            st.acc += __shfl(winc, 63);
            any = true;
            if (ce == ray_end) { done = true; break; }
        }
        if (done) {
            if (!DENSITY_ONLY && p.out_rgb && lane == 0) {
                const size_t ray = p.ray_index[q0 + qq];
                if (!st.found) st.depth = c_mid[(ray_end - 1) & (RP_RING - 1)];   // searchsorted clamps to the last sample
                float o0 = st.r0 + p.background.r * (1.0f - st.acc), o1 = st.r1 + p.background.g * (1.0f - st.acc), o2 = st.r2 + p.background.b * (1.0f - st.acc);
                if (p.background.clamp) { o0 = fminf(fmaxf(o0, 0.f), 1.f); o1 = fminf(fmaxf(o1, 0.f), 1.f); o2 = fminf(fmaxf(o2, 0.f), 1.f); }
                p.out_rgb[3 * ray] = o0; p.out_rgb[3 * ray + 1] = o1; p.out_rgb[3 * ray + 2] = o2;
                p.out_acc[ray] = st.acc;
                p.out_depth[ray] = st.depth;
            }
        } else if (any && lane == 0) {
            open[g & 1] = st;
        }
    }
    };

    Matched cur, nxt;
    if (ngroups) {
        Pending pd;
        match_stage(0, cur);
        __syncthreads();
        match_search(0, cur, pd);
        match_finish(cur, pd);
    }
    nxt = cur;

    for (uint32_t g = 0; g < ngroups; ++g) {
        uint32_t gs, ge, qa, P;
        step_geometry(g, gs, ge, qa, P);
        const bool more = g + 1 < ngroups;
        const uint32_t sl = gs + (uint32_t)wave * 32 + ((uint32_t)lane & 31u);
        uint32_t ql, jdummy;
        sample_of(g, ql, jdummy);
        const size_t q = q0 + ql;
        const float e0 = cur.e0, e1 = cur.e1;
        __syncthreads();
        stage_weights(lds, p.pk + OFF_W1, lfloats(KS1, OT));
        // ---- barycentric gather of this lane's sample (tn_interp.hip's summation order)
        float bin[KSH];
        {
            const float w0 = 1.0f - ((cur.b0 + cur.b1) + cur.b2);
            const uint32_t vv[4] = {cur.v4.y, cur.v4.z, cur.v4.w, cur.v4.x};
            const float ww[4] = {cur.b0, cur.b1, cur.b2, w0};
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) bin[ks] = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (vv[k] != TN_EMPTY) {
                    const float4 *row = reinterpret_cast<const float4 *>(p.fieldT + (size_t)vv[k] * FD + 32 * h);
#pragma unroll
                    for (int qq = 0; qq < 8; ++qq) {
                        const float4 x = row[qq];
                        bin[4 * qq] += ww[k] * x.x; bin[4 * qq + 1] += ww[k] * x.y;
                        bin[4 * qq + 2] += ww[k] * x.z; bin[4 * qq + 3] += ww[k] * x.w;
                    }
                }
            }
        }
        stage_wait();
        // (every stage_wait drains the wave's outstanding loads, so the loads of the NEXT step's match are issued right
        //  after one and consumed a whole MFMA layer later)
        SegLoad sgl;
        sgl.on = false; sgl.ray = 0; sgl.n = 0;
        if (more) seg_issue_count(g + 1, sgl, nxt);
        {
            f32x16 acc[OT];
            zero_acc(acc);
            gemm_steps<KS1, 0, OT>(acc, bin, lds, lane);
            bias_step<KS1, OT>(acc, lds, lane);
            relu_to_bin(acc, bin);
        }
        __syncthreads();
        stage_weights(lds, p.pk + OFF_W2, lfloats(KSH, OT));
        if (g > 0) composite_step(g - 1);        // LDS + shuffles only: rides with the weight copy
        stage_wait();
        if (more) seg_issue_rows(sgl);
        {
            f32x16 acc[OT];
            zero_acc(acc);
            gemm_steps<KSH, 0, OT>(acc, bin, lds, lane);
            bias_step<KSH, OT>(acc, lds, lane);
            relu_to_bin(acc, bin);
        }
        if (more) seg_store(sgl);                 // visible to everyone after the next barrier
        __syncthreads();
        stage_weights(lds, p.pk + OFF_W3, N_W3);
        stage_wait();
        Pending pd;
        pd.hit = false; pd.t_in = 0.f; pd.t_out = 1.f; pd.cur = 0.f; pd.q0f = pd.q1f = pd.q2f = make_float2(0.f, 0.f);
        if (more) match_search(g + 1, nxt, pd);   // LDS search; the segment's record is in flight during layer 3
        {
            f32x16 acc[OT];
            zero_acc(acc);
            gemm_steps<KSH, 0, OT>(acc, bin, lds, lane);
            bias_step<KSH, OT>(acc, lds, lane);
            relu_to_bin(acc, bin);
        }
        {
            const float *dv = lds + lfloats(KSH, OT);
            const float raw = head_dot(dv + 64 * h, bin) + dv[128];
            const float sp = raw > 20.0f ? raw : log1pf(expf(raw));
            if (h == 0) {
                const uint32_t li = sl & (RP_RING - 1);
                c_dd[li] = (e1 - e0) * sp;
                c_mid[li] = 0.5f * (e0 + e1);
            }
        }
        if (more) match_finish(nxt, pd);
        if constexpr (!DENSITY_ONLY) {
            __syncthreads();
            stage_weights(lds, p.pk + OFF_WHEAD, N_WHEAD);
            stage_wait();
            {
                f32x16 acc[OT];
                zero_acc(acc);
                const float *e = p.enc + q * ENC_PAD;
#pragma unroll
                for (int ks = 0; ks < KSE; ++ks) {
                    const float b = e[2 * ks + h];
                    const float *wrow = lds + (size_t)ks * OT * 64 + lane;
#pragma unroll
                    for (int t = 0; t < OT; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wrow[t * 64], b, acc[t], 0, 0, 0);
                }
                gemm_steps<KSH, KSE, OT>(acc, bin, lds, lane);
                bias_step<HEAD_KS, OT>(acc, lds, lane);
                if (p.ray_bias) add_ray_bias(acc, p.ray_bias + q * HID, h);
                relu_to_bin(acc, bin);
            }
            const float *cv = lds + lfloats(HEAD_KS, OT);
            const float c0 = head_dot(cv + 64 * h, bin) + cv[384];
            const float c1 = head_dot(cv + 128 + 64 * h, bin) + cv[385];
            const float c2 = head_dot(cv + 256 + 64 * h, bin) + cv[386];
            if (h == 0) {
                const uint32_t li = sl & (RP_RING - 1);
                c_rgb[li] = 1.0f / (1.0f + expf(-c0));
                c_rgb[RP_RING + li] = 1.0f / (1.0f + expf(-c1));
                c_rgb[2 * RP_RING + li] = 1.0f / (1.0f + expf(-c2));
            }
        }
        cur = nxt;
    }
    if (ngroups) {
        __syncthreads();
        composite_step(ngroups - 1);
    }
}

void launch_render_pass(const uint32_t *num_visited, const float *dist, const float *bary, const uint32_t *verts, uint32_t M,
                        const uint32_t *ray_index, size_t r, uint32_t S, const float *edges, const float *fieldT,
                        const float *dirs, const MlpPacks &w, Background background, float *out_weights, float *out_rgb,
                        float *out_acc, float *out_depth, hipStream_t stream) {
    if (r == 0) return;
    if (S < 64) throw Error("render_pass needs at least 64 samples per ray");
    if (M > 512) throw Error("render_pass supports max_ray_triangles <= 512");
    if ((size_t)S * ((r + 255) / 256 + 1) >= 0xFFFFFFFFull) throw Error("render_pass: too many samples per block");
    const bool density_only = dirs == nullptr;
    if (!density_only && !(out_rgb && out_acc && out_depth)) throw Error("render_pass: colour pass without output buffers");
    const float *pk = w.pk_gather;
    float *enc = density_only ? nullptr : w.enc;
    uint32_t *nvh = w.nvh;
    hipLaunchKernelGGL(k_gather_counts, dim3((unsigned)((r + 255) / 256)), dim3(256), 0, stream, r, ray_index, num_visited, nvh);
    if (!density_only) launch_dir_encoding(r, dirs, enc, stream);
    RenderPassParams p{};
    p.num_visited = num_visited; p.dist = dist; p.bary = bary; p.verts = verts; p.ray_index = ray_index; p.edges = edges;
    p.nv_hit = nvh; p.fieldT = fieldT; p.enc = enc; p.ray_bias = density_only ? nullptr : w.ray_bias; p.pk = pk; p.out_weights = out_weights; p.out_rgb = out_rgb; p.out_acc = out_acc;
    p.out_depth = out_depth; p.r = r; p.S = S; p.M = M; p.background = background;
    auto smem_for = [](size_t m) { return (MAX_STAGE_FLOATS + 2 * (size_t)RP_PIECES * m + 8 + 5 * RP_RING) * sizeof(float) + 2 * sizeof(RayState); };
    const size_t smem = smem_for(M);
    static PerDeviceOnce lds_attr;
    lds_attr.run([&] {
        allow_dynamic_lds(reinterpret_cast<const void *>(k_render_pass<false>), smem_for(512));
        allow_dynamic_lds(reinterpret_cast<const void *>(k_render_pass<true>), smem_for(512));
    });
    const unsigned grid = (unsigned)(r < 256 ? r : 256);   // one 8-wave block per CU, each owns a range of rays
    if (density_only) hipLaunchKernelGGL(k_render_pass<true>, dim3(grid), dim3(MLP_BLOCK), smem, stream, p);
    else hipLaunchKernelGGL(k_render_pass<false>, dim3(grid), dim3(MLP_BLOCK), smem, stream, p);
}

}  // namespace tn
