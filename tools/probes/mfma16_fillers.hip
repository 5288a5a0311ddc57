// The same question as mfma_fillers.hip for v_mfma_f32_16x16x4_f32 (32 cycles per SIMD): 64 accumulators of four registers (256 AGPRs), one
// wave per SIMD, per iteration 32 MFMAs on 32 different accumulators; between them a mix that a Winograd kernel with 16-tile blocks x 64
// output channels would have: R ds_read_b64 and V packed adds per 32 MFMAs.
//   hipcc -w --offload-arch=gfx950 -O3 -std=c++17 tools/probes/mfma16_fillers.hip -o build/probes/mfma16_fillers
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define SB() __builtin_amdgcn_sched_barrier(0)

template <int R, int V>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    f32x4 acc[64];
#pragma unroll
    for (int p = 0; p < 64; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
    f32x2 w2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) w2[i] = f32x2{a + i, a - i};
    const f32x2 b2 = f32x2{b, b * 0.5f};
    float2 ld[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) ld[i] = make_float2(0.f, 0.f);
    lds[threadIdx.x] = a;
    __syncthreads();
    const float* lp = &lds[(threadIdx.x & 63) * 2];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int m = 0; m < 32; ++m) {
                acc[h * 32 + m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[h * 32 + m], 0, 0, 0);
                SB();
                if (m < R) ld[m & 3] = *reinterpret_cast<const float2*>(lp + (m & 15) * 128);
                if (m >= 32 - V) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(w2[m & 7]) : "v"(b2));
                SB();
            }
        a += ld[0].x + ld[1].y + ld[2].x + ld[3].y;
    }
    float s = 0.f;
#pragma unroll
    for (int p = 0; p < 64; ++p) s += acc[p][0] + acc[p][1] + acc[p][2] + acc[p][3];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += w2[i][0] + w2[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int R, int V>
static void run(float* out, double ghz) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<R, V>), dim3(256), dim3(256), 0, 0, out, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<R, V>), dim3(256), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("16x16x4: per 32 MFMAs %2d ds_read_b64 + %2d v_pk_add_f32: %.3f ms, %.1f TFLOP/s\n", R, V, ms, 256.0 * 4 * iters * 64 * 2048.0 / (ms * 1e-3) / 1e12);
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 256 * 4);
    run<0, 0>(out, 2.4); run<24, 0>(out, 2.4); run<0, 8>(out, 2.4); run<24, 8>(out, 2.4); run<24, 16>(out, 2.4); run<32, 32>(out, 2.4);
    return 0;
}
