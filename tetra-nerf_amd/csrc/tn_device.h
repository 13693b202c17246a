// tn_device.h -- device-side geometry primitives shared by the trace kernels (gfx950).
//
// The triangle routine below DEFINES a hit for this library (it stands in for OptiX's
// built-in triangle intersection, reference call sites src/optix/optix_trace_rays.cu:280-292,
// 311-326).  It is a watertight edge-function test in a ray-aligned sheared space, fp32,
// one IEEE rounding per operation (the library is compiled with -ffp-contract=off and HIP's
// default correctly-rounded fp32 division), with a double-precision retry when an edge
// function is exactly zero.  A float product difference fl(fl(ab) - fl(cd)) is either 0 or
// carries the exact sign of ab - cd, so all edge-function signs are exact with respect to the
// (rounded) sheared 2-D vertex positions, and E(P,Q) == -E(Q,P) bitwise: faces sharing an
// edge always agree on which side of it the ray passes.
#pragma once
#include "tn_common.h"

namespace tn {

struct RayPre {
    int kx, ky, kz;
    float Sx, Sy, Sz;
    float ox, oy, oz;
};

// component k of (x,y,z) as two v_cndmask on the index bits (a `k == 0 ? .. : k == 1 ? ..` chain is
// turned into a switch and lowered to exec-masked branches)
__device__ __forceinline__ float pick(float x, float y, float z, int k) {
    const bool b0 = (k & 1) != 0, b1 = (k & 2) != 0;
    const float lo = b0 ? y : x;
    return b1 ? z : lo;
}

__device__ __forceinline__ RayPre ray_pre(float ox, float oy, float oz, float dx, float dy, float dz) {
    RayPre r;
    int kz = 0;
    float m = fabsf(dx);
    if (fabsf(dy) > m) { kz = 1; m = fabsf(dy); }
    if (fabsf(dz) > m) { kz = 2; }
    int kx = kz + 1; if (kx == 3) kx = 0;
    int ky = kx + 1; if (ky == 3) ky = 0;
    const float dkz = pick(dx, dy, dz, kz);
    if (dkz < 0.0f) { const int t = kx; kx = ky; ky = t; }
    r.kx = kx; r.ky = ky; r.kz = kz;
    r.Sx = pick(dx, dy, dz, kx) / dkz;
    r.Sy = pick(dx, dy, dz, ky) / dkz;
    r.Sz = 1.0f / dkz;
    r.ox = ox; r.oy = oy; r.oz = oz;
    return r;
}

// vertex -> sheared ray space: (x', y') transverse, z' = scaled distance along the ray
struct SV {
    float x, y, z;
};

__device__ __forceinline__ SV shear(const RayPre &r, float px, float py, float pz) {
    const float ax = px - r.ox, ay = py - r.oy, az = pz - r.oz;
    const float akx = pick(ax, ay, az, r.kx), aky = pick(ax, ay, az, r.ky), akz = pick(ax, ay, az, r.kz);
    SV s;
    s.x = akx - r.Sx * akz;
    s.y = aky - r.Sy * akz;
    s.z = r.Sz * akz;
    return s;
}

// edge function of the directed edge P -> Q:  E(P,Q) = Q.x*P.y - Q.y*P.x  (== -E(Q,P) bitwise)
__device__ __forceinline__ float edge_f(const SV &P, const SV &Q) { return Q.x * P.y - Q.y * P.x; }
__device__ __forceinline__ float edge_d(const SV &P, const SV &Q) {
    return (float)((double)Q.x * (double)P.y - (double)Q.y * (double)P.x);
}

// given the three edge functions U=E(B,C), V=E(C,A), W=E(A,B) (after the zero retry):
// hit decision + (t,u,v).  u,v = weights of B and C (OptiX convention).
__device__ __forceinline__ bool tri_finish(float U, float V, float W, float Az, float Bz, float Cz,
                                           float &t, float &u, float &v) {
    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return false;
    const float det = (U + V) + W;
    if (det == 0.0f) return false;
    const float T = (U * Az + V * Bz) + W * Cz;
    const float tt = T / det;
    if (!(tt > 0.0f && tt < 1e16f)) return false;  // tmin 0 / tmax 1e16: optix_trace_rays.cu:284-285
    t = tt;
    u = V / det;
    v = W / det;
    return true;
}

__device__ __forceinline__ bool tri_hit_sv(const SV &A, const SV &B, const SV &C, float &t, float &u, float &v) {
    float U = edge_f(B, C), V = edge_f(C, A), W = edge_f(A, B);
    if (U == 0.0f || V == 0.0f || W == 0.0f) {
        U = edge_d(B, C); V = edge_d(C, A); W = edge_d(A, B);
    }
    return tri_finish(U, V, W, A.z, B.z, C.z, t, u, v);
}

// Line-vs-padded-box test used for BVH culling.  Conservative with respect to tri_hit: the
// 2-D hit decision is exact for vertex positions perturbed by <= 6*2^-24*(|o|+|v|)_inf, so a
// box padded by 16*2^-23*(|o|_inf + scene_max) (passed in as `pad`) contains every face the
// triangle routine can accept.  The whole line is tested (no clipping to t>0): t is decided
// per face by tri_finish only.
__device__ __forceinline__ bool line_box(float ox, float oy, float oz, float ix, float iy, float iz,
                                         float lx, float ly, float lz, float hx, float hy, float hz, float pad) {
    float tn_ = -INFINITY, tf_ = INFINITY;
    {
        const float a = ((lx - ox) - pad) * ix, b = ((hx - ox) + pad) * ix;
        const float mn = a < b ? a : b, mx = a < b ? b : a;
        if (mn > tn_) tn_ = mn;
        if (mx < tf_) tf_ = mx;
    }
    {
        const float a = ((ly - oy) - pad) * iy, b = ((hy - oy) + pad) * iy;
        const float mn = a < b ? a : b, mx = a < b ? b : a;
        if (mn > tn_) tn_ = mn;
        if (mx < tf_) tf_ = mx;
    }
    {
        const float a = ((lz - oz) - pad) * iz, b = ((hz - oz) + pad) * iz;
        const float mn = a < b ? a : b, mx = a < b ? b : a;
        if (mn > tn_) tn_ = mn;
        if (mx < tf_) tf_ = mx;
    }
    const float slack = 4.0f * 1.1920929e-7f * (fabsf(tn_) + fabsf(tf_));
    return (tn_ <= tf_ + slack) || !(tn_ == tn_) || !(tf_ == tf_);
}

__device__ __forceinline__ float safe_inv(float d) {
    if (fabsf(d) < 1e-30f) d = (__float_as_uint(d) >> 31) ? -1e-30f : 1e-30f;
    return 1.0f / d;
}

// get_common_tetrahedra, reference optix_trace_rays.cu:22-37 (check order is observable)
__device__ __forceinline__ bool common_tet(uint2 a, uint2 b, uint32_t &cell) {
    if (a.x == b.x) { cell = a.x; return true; }
    if (a.x == b.y) { cell = a.x; return true; }
    if (a.y == b.x) { cell = a.y; return true; }
    if (a.y == b.y) { cell = a.y; return true; }
    return false;
}

// combine_indices, reference optix_trace_rays.cu:39-75
__device__ __forceinline__ void combine_indices(const uint32_t id1[3], const uint32_t id2[3], float u1, float v1,
                                                float u2, float v2, uint32_t out4[4], float bc1[3], float bc2[3]) {
    out4[0] = 0; out4[1] = id1[0]; out4[2] = id1[1]; out4[3] = id1[2];
    bc1[0] = 1.0f - u1 - v1; bc1[1] = u1; bc1[2] = v1;
    const float ref2[3] = {1.0f - u2 - v2, u2, v2};
    bc2[0] = 0.0f; bc2[1] = 0.0f; bc2[2] = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        bool was_break = false;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (!was_break && id1[j] == id2[i]) { bc2[j] = ref2[i]; was_break = true; }
        }
        if (!was_break) out4[0] = id2[i];
    }
}

__device__ __forceinline__ uint64_t lanemask_lt() {
    const unsigned lane = __lane_id();
    return lane == 0 ? 0ull : (~0ull >> (64 - lane));
}

}  // namespace tn
