// What does the f32 matrix pipe SUSTAIN on this chip?  (round 6)  The dense f32 MFMA peak of MI355X_MICROARCH.md, 157.3 TFLOP/s, is
// 256 CUs x 4 SIMDs x 64 MACs per clock x 2.4 GHz; every f32 MFMA kernel of this repository has been observed at 2.1-2.4 GHz, the lower
// the busier its matrix pipe (profiles/r06_gemm_decode_pmc.md: 70.7 % busy at 2.14 GHz, 50.9 % at 2.37).  This probe runs NOTHING but
// back-to-back v_mfma_f32_32x32x2_f32 (four independent accumulator chains per wave, one or two waves per SIMD, every CU) for tens of
// milliseconds -- long enough for the power management to settle -- on (a) constant operands and (b) operands that change every MFMA
// and carry random mantissas, and prints the sustained TFLOP/s: the ceiling a perfect f32 kernel would be priced against in practice.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_sustained.hip -o /tmp/mfma_sustained && /tmp/mfma_sustained
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <bool REAL>
__global__ __launch_bounds__(512, 1) void sustain(const float* __restrict__ src, float* __restrict__ out, int iters) {
    f32x16 acc[4];
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0;
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) {   // eight operand pairs per lane: random values (REAL) or one constant
        a[i] = REAL ? src[(blockIdx.x * 512 + threadIdx.x) * 16 + i] : 1.f;
        b[i] = REAL ? src[(blockIdx.x * 512 + threadIdx.x) * 16 + 8 + i] : 2.f;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + c) & 7], b[(u + 2 * c) & 7], acc[c], 0, 0, 0);
    }
    float s = 0;
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
    int dev = 0;
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, dev);
    const int cus = p.multiProcessorCount;
    float *src, *out;
    const size_t n = (size_t)cus * 512 * 16;
    hipMalloc(&src, n * 4);
    hipMalloc(&out, (size_t)cus * 512 * 4);
    float* h = (float*)malloc(n * 4);
    srand(1);
    for (size_t i = 0; i < n; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
    hipMemcpy(src, h, n * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int waves = 4; waves <= 8; waves += 4)
        for (int real = 0; real < 2; ++real)
            for (int rep = 0; rep < 2; ++rep) {
                const int iters = rep ? 60000 : 6000;   // ~4 ms and ~40 ms
                auto launch = [&](int it) {
                    if (real) hipLaunchKernelGGL(sustain<true>, dim3(cus), dim3(64 * waves), 0, 0, src, out, it);
                    else hipLaunchKernelGGL(sustain<false>, dim3(cus), dim3(64 * waves), 0, 0, src, out, it);
                };
                launch(100);
                hipDeviceSynchronize();
                hipEventRecord(e0);
                launch(iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                const double flop = (double)cus * waves * iters * 32.0 * 2.0 * 32 * 32 * 2;   // 32 MFMAs per iteration, 2 x 32 x 32 x 2 flop each
                printf("%d CUs, %d waves per CU, %s operands, %7.2f ms: %6.1f TFLOP/s = %.3f of 157.3 (%.2f GHz-equivalent at 64 MACs per SIMD and clock)\n", cus, waves,
                       real ? "random, changing every MFMA" : "constant                   ", ms, flop / ms * 1e-9, flop / ms * 1e-9 / 157.3,
                       flop / ms * 1e-9 / 157.3 * 2.4);
            }
    return 0;
}
