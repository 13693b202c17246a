// tn_uint32.hip -- the two uint32-indexed helper ops the reference exports for its (dormant) occupancy
// field: gather_uint32 and scatter_ema_uint32 (src/tetrahedra_tracer.cu:30-113).  Not on the model path.
// The EMA update is a compare-and-swap loop on the value's bit pattern; unlike the reference it keeps
// the full precision of the new value (the reference routes the bit pattern through a float temporary,
// :61-66,77, which only passes its test through the tolerance).
#include "tn_common.h"
#include "tn_kernels.h"

namespace tn {

namespace {

template <typename T>
__global__ void k_gather_u32(uint32_t num_values, uint32_t num_indices, const uint32_t *__restrict__ indices,
                             const T *__restrict__ values, T *__restrict__ result) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= num_indices) return;
    const uint32_t k = indices[i];
    if (k >= num_values) return;  // out-of-range indices leave the slot untouched, as in the reference
    result[i] = values[k];
}

__device__ __forceinline__ void atomic_ema(float *addr, float decay, float update) {
    unsigned int *a = reinterpret_cast<unsigned int *>(addr);
    unsigned int old = *a, assumed;
    do {
        assumed = old;
        const float nv = __uint_as_float(assumed) * decay + (1 - decay) * update;
        old = atomicCAS(a, assumed, __float_as_uint(nv));
    } while (assumed != old);  // integer comparison: no hang on NaN
}

__device__ __forceinline__ void atomic_ema(double *addr, double decay, double update) {
    unsigned long long *a = reinterpret_cast<unsigned long long *>(addr);
    unsigned long long old = *a, assumed;
    do {
        assumed = old;
        const double nv = __longlong_as_double((long long)assumed) * decay + (1 - decay) * update;
        old = atomicCAS(a, assumed, (unsigned long long)__double_as_longlong(nv));
    } while (assumed != old);
}

template <typename T>
__global__ void k_scatter_ema_u32(uint32_t num_result, uint32_t num_indices, const uint32_t *__restrict__ indices, T decay,
                                  const T *__restrict__ values, T *__restrict__ result) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= num_indices) return;
    const uint32_t k = indices[i];
    if (k >= num_result) return;
    atomic_ema(&result[k], decay, values[i]);
}

}  // namespace

void launch_gather_uint32(int elem_size, uint32_t num_values, uint32_t num_indices, const uint32_t *indices,
                          const void *values, void *result, hipStream_t stream) {
    if (num_indices == 0) return;
    const unsigned grid = (num_indices + 255) / 256;
    if (elem_size == 4)
        hipLaunchKernelGGL(k_gather_u32<float>, dim3(grid), dim3(256), 0, stream, num_values, num_indices, indices,
                           (const float *)values, (float *)result);
    else if (elem_size == 8)
        hipLaunchKernelGGL(k_gather_u32<double>, dim3(grid), dim3(256), 0, stream, num_values, num_indices, indices,
                           (const double *)values, (double *)result);
    else throw Error("self must be a tensor of a floating-point type");
}

void launch_scatter_ema_uint32(int elem_size, uint32_t num_result, uint32_t num_indices, const uint32_t *indices,
                               double decay, const void *values, void *result, hipStream_t stream) {
    if (num_indices == 0) return;
    const unsigned grid = (num_indices + 255) / 256;
    if (elem_size == 4)
        hipLaunchKernelGGL(k_scatter_ema_u32<float>, dim3(grid), dim3(256), 0, stream, num_result, num_indices, indices,
                           (float)decay, (const float *)values, (float *)result);
    else if (elem_size == 8)
        hipLaunchKernelGGL(k_scatter_ema_u32<double>, dim3(grid), dim3(256), 0, stream, num_result, num_indices, indices, decay,
                           (const double *)values, (double *)result);
    else throw Error("self must be a tensor of a floating-point type");
}

}  // namespace tn
