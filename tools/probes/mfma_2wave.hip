// Does a SECOND wave per SIMD hide the filler instructions beside v_mfma_f32_32x32x2_f32?  Companion of mfma_fillers.hip (one wave per
// SIMD, sixteen accumulators, 512 registers): here a workgroup has WAVES x 4 waves (WAVES per SIMD), each wave owns 16 / WAVES
// accumulators and runs the same gap pattern (F other instructions pinned between every two MFMAs).  Total MFMAs per SIMD are equal
// in every configuration, so the TFLOP/s column compares directly.
//   hipcc -w --offload-arch=gfx950 -O3 -std=c++17 tools/probes/mfma_2wave.hip -o build/probes/mfma_2wave
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define SB() __builtin_amdgcn_sched_barrier(0)

// KIND 0: scalar v_add_f32, 1: ds_read_b128, 2: alternate add / ds_read, 3: v_pk_add_f32, 5: the Winograd forward mix per 16 MFMAs
// (16 v_pk_add-equivalents as 32 scalar adds when F == 2, 8 ds_read_b128)
template <int WAVES, int F, int KIND>
__global__ __launch_bounds__(256 * WAVES, 1) void k(float* out, int iters) {
    constexpr int NA = 16 / WAVES;
    __shared__ __attribute__((aligned(16))) float lds[8192];
    f32x16 acc[NA];
#pragma unroll
    for (int p = 0; p < NA; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = a + i;
    f32x2 w2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) w2[i] = f32x2{a + i, a - i};
    const f32x2 b2 = f32x2{b, b * 0.5f};
    float4 ld[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) ld[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    lds[threadIdx.x & 255] = a;
    __syncthreads();
    const float* lp = &lds[(threadIdx.x & 63) * 4];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 16 / NA; ++rep) {
#pragma unroll
            for (int m = 0; m < NA; ++m) {
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m], 0, 0, 0);
                SB();
#pragma unroll
                for (int f = 0; f < F; ++f) {
                    const int j = (rep * NA + m) * F + f;
                    const bool lds_op = KIND == 1 || (KIND == 2 && (f & 1)) || (KIND == 5 && f == F - 1 && (m & 1));
                    if (lds_op) ld[j & 3] = *reinterpret_cast<const float4*>(lp + (j & 7) * 256);
                    else if (KIND == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(w2[j & 7]) : "v"(b2));
                    else v[j & 7] += b;
                }
                SB();
            }
        }
        a += ld[0].x + ld[1].y + ld[2].z + ld[3].w;
    }
    float s = 0.f;
#pragma unroll
    for (int p = 0; p < NA; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[p][r];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i] + w2[i][0] + w2[i][1];
    out[blockIdx.x * 256 * WAVES + threadIdx.x] = s;
}

template <int WAVES, int F, int KIND>
static void run(float* out, double ghz) {
    const int iters = 4000 / WAVES;   // every wave runs 16 MFMAs per iteration: the same MFMA total per SIMD in every configuration
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<WAVES, F, KIND>), dim3(256), dim3(256 * WAVES), 0, 0, out, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<WAVES, F, KIND>), dim3(256), dim3(256 * WAVES), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)iters * 16.0 * WAVES;   // MFMAs per SIMD
    const double per = ms * 1e-3 * ghz * 1e9 / mf;
    const char* kn = KIND == 0 ? "v_add     " : KIND == 1 ? "ds_read128" : KIND == 2 ? "add+ds    " : KIND == 3 ? "v_pk_add  " : "wino mix  ";
    printf("waves/SIMD %d acc/wave %2d  %s x%d per gap: %.3f ms, %.1f cycles per MFMA per SIMD (%.1f %% of the 64-cycle rate), %.1f TFLOP/s\n",
           WAVES, 16 / WAVES, kn, F, ms, per, 6400.0 / per, 256.0 * 4 * mf * 4096.0 / (ms * 1e-3) / 1e12);
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 1024 * 4);
    int khz = 0;
    hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    const double ghz = khz / 1e6;
#define ROW(F, K) run<1, F, K>(out, ghz); run<2, F, K>(out, ghz); run<4, F, K>(out, ghz);
    ROW(0, 0) ROW(1, 0) ROW(2, 0) ROW(3, 0) ROW(4, 0)
    ROW(1, 3) ROW(2, 3)
    ROW(2, 2) ROW(4, 2)
    ROW(2, 5) ROW(3, 5) ROW(4, 5)
    return 0;
}
