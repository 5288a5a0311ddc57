// Shared host/device helpers for libvaecap (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

namespace vc {

// ---- error plumbing: every C-ABI entry returns 0 or a hipError_t / VC_E* code
enum { VC_OK = 0, VC_EINVAL = 10001, VC_EWORKSPACE = 10002 };

char* last_error_buf();  // thread-local, defined in api.hip

inline int fail(int code, const char* fmt, const char* a = "", long b = 0, long c = 0) {
    snprintf(last_error_buf(), 512, fmt, a, b, c);
    return code;
}

#define VC_CHECK_ARG(cond, what)                                              \
    do {                                                                      \
        if (!(cond)) return vc::fail(vc::VC_EINVAL, "%s: invalid argument: " what, __func__); \
    } while (0)

inline int launch_status(const char* fn) {
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) return 0;
    snprintf(last_error_buf(), 512, "%s: kernel launch failed: %s", fn, hipGetErrorString(e));
    return (int)e;
}

#define VC_LAUNCH_CHECK()                                                     \
    do {                                                                      \
        int s__ = vc::launch_status(__func__);                                \
        if (s__) return s__;                                                  \
    } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// wave64 sum
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// block-wide sum for blockDim.x == NT (multiple of 64); result valid in every thread.
template <int NT>
__device__ __forceinline__ float block_sum(float v, float* sh /* >= NT/64 floats */) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) t += sh[i];  // fixed order -> deterministic
    return t;
}
template <int NT>
__device__ __forceinline__ float block_max(float v, float* sh) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    float t = sh[0];
#pragma unroll
    for (int i = 1; i < NT / 64; ++i) t = fmaxf(t, sh[i]);
    return t;
}

// gemm.hip: split-K partial products for consumers that reduce them in their own kernel (lstm.hip)
int gemm_partials_f32(hipStream_t st, int ta, int tb, int M, int N, int K, const float* A, long lda, const float* B, long ldb,
                      float* ws, size_t ws_bytes, int max_splits, int* splits_out);
size_t gemm_partials_bytes(int M, int N, int K, int max_splits);
int gemm_default_precision();   // the deprecated process-wide default of vc_gemm_set_precision (0 unless a caller set it)

}  // namespace vc
