// What does a lone wave per SIMD hide beside v_mfma_f32_32x32x2_f32?  One workgroup of four waves per CU (512 registers per lane: sixteen
// 32x32 accumulators in AGPRs, as the Winograd kernels), per iteration sixteen MFMAs on sixteen different accumulators with F other
// instructions pinned between every two of them (sched_barrier): scalar fp32 adds, ds_read_b128 or a mix.  Prints cycles per MFMA
// (wall clock x the clock the chip reports) -- 64 = the matrix pipe's issue rate.
//   hipcc -w --offload-arch=gfx950 -O3 -std=c++17 tools/probes/mfma_fillers.hip -o build/probes/mfma_fillers
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define SB() __builtin_amdgcn_sched_barrier(0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int F, int KIND>   // KIND 0: VALU adds, 1: ds_read_b128, 2: alternate, 3: packed v_pk_add_f32, 4: v_fma_f32
__global__ __launch_bounds__(256, 1) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    f32x16 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p)
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
    lds[threadIdx.x] = a;
    __syncthreads();
    const float* lp = &lds[(threadIdx.x & 63) * 4];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m], 0, 0, 0);
            SB();
#pragma unroll
            for (int f = 0; f < F; ++f) {
                const bool lds_op = KIND == 1 || (KIND == 2 && (f & 1));
                if (lds_op) ld[(m * F + f) & 3] = *reinterpret_cast<const float4*>(lp + ((m * F + f) & 7) * 256);
                else if (KIND == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(w2[(m * F + f) & 7]) : "v"(b2));
                else if (KIND == 4) v[(m * F + f) & 7] = __builtin_fmaf(v[(m * F + f) & 7], b, a);
                else v[(m * F + f) & 7] += b;
            }
            SB();
        }
        a += ld[0].x + ld[1].y + ld[2].z + ld[3].w;
    }
    float s = 0.f;
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[p][r];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i] + w2[i][0] + w2[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int F, int KIND>
static void run(float* out, double ghz) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<F, KIND>), dim3(256), dim3(256), 0, 0, out, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<F, KIND>), dim3(256), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double per = ms * 1e-3 * ghz * 1e9 / (iters * 16.0);
    printf("%s fillers per gap %d: %.3f ms, %.1f cycles per MFMA at %.2f GHz (%.1f %% of the 64-cycle issue rate), %.1f TFLOP/s\n",
           KIND == 0 ? "VALU add " : KIND == 1 ? "ds_read128" : KIND == 2 ? "mixed     " : KIND == 3 ? "v_pk_add  " : "v_fma     ", F, ms, per, ghz, 6400.0 / per, 256.0 * 4 * iters * 16 * 4096.0 / (ms * 1e-3) / 1e12);
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 256 * 4);
    int khz = 0;
    hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    const double ghz = khz / 1e6;
    run<0, 0>(out, ghz); run<1, 0>(out, ghz); run<2, 0>(out, ghz); run<3, 0>(out, ghz); run<4, 0>(out, ghz); run<6, 0>(out, ghz); run<8, 0>(out, ghz);
    run<1, 1>(out, ghz); run<2, 1>(out, ghz); run<4, 1>(out, ghz);
    run<2, 2>(out, ghz); run<4, 2>(out, ghz); run<6, 2>(out, ghz);
    run<1, 3>(out, ghz); run<2, 3>(out, ghz); run<4, 3>(out, ghz);
    run<1, 4>(out, ghz); run<2, 4>(out, ghz); run<4, 4>(out, ghz);
    return 0;
}
