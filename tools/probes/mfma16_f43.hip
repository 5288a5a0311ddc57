// Instruction-mix probe for a Winograd F(4x4,3x3) forward on v_mfma_f32_16x16x4_f32 (32-cycle issue rate), two workgroups of four
// waves per CU (two waves per SIMD), NA four-register accumulators per wave.  Between every two MFMAs: F scalar VALU operations
// (v_fma_f32) and one ds_read_b128 every RD-th MFMA (RD == 0: none).  Rows: the F(2x2,3x3) kernel's mix (32 accumulators, ~1.3 VALU
// and 0.375 reads per MFMA) against F(4x4,3x3) on 16 tiles x 16 channels (36 accumulators, 4.7 VALU and 0.5 reads per MFMA).
//   hipcc -w --offload-arch=gfx950 -O3 -std=c++17 tools/probes/mfma16_f43.hip -o build/probes/mfma16_f43
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define SB() __builtin_amdgcn_sched_barrier(0)
#ifndef OPK
#define OPK 1
#endif
__device__ __forceinline__ float vop(float acc, float x, float b) {
#if OPK == 0
    return acc + x;                       // v_add_f32 (VOP2)
#elif OPK == 1
    return __builtin_fmaf(x, b, acc);     // v_fma_f32 / v_fmac_f32 with a register multiplier
#else
    return __builtin_fmaf(x, 4.0f, acc);  // inline-constant multiplier
#endif
}

template <int NA, int F10, int RD>   // F10 = VALU operations per MFMA x 10 (47 = 4.7)
__global__ __launch_bounds__(256, 2) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[16384];   // 64 KB: two workgroups per CU
    f32x4 acc[NA];
#pragma unroll
    for (int p = 0; p < NA; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = a + i;
    float4 ld[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) ld[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = a;
    __syncthreads();
    const float* lp = &lds[(threadIdx.x & 63) * 4];
    for (int it = 0; it < iters; ++it) {
        int fdone = 0;
#pragma unroll
        for (int m = 0; m < NA; ++m) {
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m], 0, 0, 0);
            SB();
            const int ftarget = (m + 1) * F10 / 10;
#pragma unroll
            for (; fdone < ftarget; ++fdone) v[fdone & 15] = vop(v[fdone & 15], v[(fdone + 5) & 15], b);
            if (RD > 0 && (m % RD) == RD - 1) ld[(m / RD) & 3] = *reinterpret_cast<const float4*>(lp + ((m / RD) & 15) * 256);
            SB();
        }
        a += ld[0].x + ld[1].y + ld[2].z + ld[3].w;
    }
    float s = 0.f;
#pragma unroll
    for (int p = 0; p < NA; ++p) s += acc[p][0] + acc[p][1] + acc[p][2] + acc[p][3];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NA, int F10, int RD>
__global__ __launch_bounds__(256, 2) void k32(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[16384];
    f32x16 acc[NA];
#pragma unroll
    for (int p = 0; p < NA; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = a + i;
    float4 ld[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) ld[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = a;
    __syncthreads();
    const float* lp = &lds[(threadIdx.x & 63) * 4];
    for (int it = 0; it < iters; ++it) {
        int fdone = 0;
#pragma unroll
        for (int m = 0; m < NA; ++m) {
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m], 0, 0, 0);
            SB();
            const int ftarget = (m + 1) * F10 / 10;
#pragma unroll
            for (; fdone < ftarget; ++fdone) v[fdone & 15] = vop(v[fdone & 15], v[(fdone + 5) & 15], b);
            if (RD > 0 && (m % RD) == RD - 1) ld[(m / RD) & 3] = *reinterpret_cast<const float4*>(lp + ((m / RD) & 15) * 256);
            SB();
        }
        a += ld[0].x + ld[1].y + ld[2].z + ld[3].w;
    }
    float s = 0.f;
#pragma unroll
    for (int p = 0; p < NA; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[p][r];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NA, int F10, int RD>
static void run32(float* out, double ghz) {
    const int iters = 32000 / NA;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k32<NA, F10, RD>), dim3(512), dim3(256), 0, 0, out, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k32<NA, F10, RD>), dim3(512), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)iters * NA * 2;
    const double per = ms * 1e-3 * ghz * 1e9 / mf;
    printf("32x32x2 acc/wave %2d  VALU/MFMA %.1f  ds_read_b128 every %d: %.3f ms, %.1f cycles per MFMA per SIMD (%.1f %% of the 64-cycle rate)\n",
           NA, F10 / 10.0, RD, ms, per, 6400.0 / per);
}

template <int NA, int F10, int RD>
static void run(float* out, double ghz, const char* what) {
    const int iters = 64000 / NA;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NA, F10, RD>), dim3(512), dim3(256), 0, 0, out, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NA, F10, RD>), dim3(512), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)iters * NA * 2;   // MFMAs per SIMD (two waves)
    const double per = ms * 1e-3 * ghz * 1e9 / mf;
    printf("acc/wave %2d  VALU/MFMA %.1f  ds_read_b128 every %d: %.3f ms, %.1f cycles per MFMA per SIMD (%.1f %% of the 32-cycle rate)  %s\n",
           NA, F10 / 10.0, RD, ms, per, 3200.0 / per, what);
}

int main() {
    float* out;
    hipMalloc(&out, 512 * 256 * 4);
    int khz = 0;
    hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    const double ghz = khz / 1e6;
    run<32, 0, 0>(out, ghz, "bare MFMAs");
    run<36, 0, 0>(out, ghz, "bare MFMAs");
    run<32, 13, 0>(out, ghz, "F(2,3) VALU only");
    run<32, 13, 3>(out, ghz, "F(2,3) mix");
    run<36, 24, 2>(out, ghz, "F(4,3) mix with packed-equivalent VALU count");
    run<36, 30, 2>(out, ghz, "");
    run<36, 40, 2>(out, ghz, "");
    run<36, 47, 0>(out, ghz, "F(4,3) VALU only");
    run<36, 47, 2>(out, ghz, "F(4,3) mix");
    run<36, 55, 2>(out, ghz, "F(4,3) mix + address / staging overhead");
    run<36, 47, 1>(out, ghz, "F(4,3) mix, twice the LDS reads");
    printf("-- the same VALU and LDS work per MFMA-cycle on v_mfma_f32_32x32x2_f32 (64-cycle rate): 8 accumulators of 16 registers\n");
    run32<8, 0, 0>(out, ghz);
    run32<8, 26, 0>(out, ghz);
    run32<8, 26, 1>(out, ghz);
    run32<8, 94, 1>(out, ghz);
    return 0;
}
