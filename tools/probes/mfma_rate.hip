// Issue-rate probe for the f32 MFMA forms at one and two waves per SIMD (hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ __launch_bounds__(256, 1) void k16(float* out, int iters, float a, float b) {
    f32x4 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = f32x4{0, 0, 0, 0};
    float av = a + threadIdx.x, bv = b;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 32; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[c], 0, 0, 0);
    }
    float s = 0;
    for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CHAINS>
__global__ __launch_bounds__(256, 1) void k32(float* out, int iters, float a, float b) {
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0;
    float av = a + threadIdx.x, bv = b;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 32; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[c], 0, 0, 0);
    }
    float s = 0;
    for (int c = 0; c < CHAINS; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <class F>
static void run(const char* name, F launch, double flop_per_mfma, int chains, int threads, int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 2000;
    launch(10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch(iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)iters * 32 * chains;  // MFMAs per wave
    const double waves = (double)blocks * threads / 64;
    printf("%-34s %8.3f ms  %6.1f ns per MFMA per wave  %7.1f TFLOP/s\n", name, ms, ms * 1e6 / mf, mf * waves * flop_per_mfma / ms * 1e-9);
}
int main() {
    float* out;
    hipMalloc(&out, 1 << 22);
    const int B = 256;
    run("16x16x4 1 chain  4 waves/CU", [&](int it) { hipLaunchKernelGGL(k16<1>, dim3(B), dim3(256), 0, 0, out, it, 1.f, 2.f); }, 2048, 1, 256, B);
    run("16x16x4 2 chains 4 waves/CU", [&](int it) { hipLaunchKernelGGL(k16<2>, dim3(B), dim3(256), 0, 0, out, it, 1.f, 2.f); }, 2048, 2, 256, B);
    run("16x16x4 4 chains 4 waves/CU", [&](int it) { hipLaunchKernelGGL(k16<4>, dim3(B), dim3(256), 0, 0, out, it, 1.f, 2.f); }, 2048, 4, 256, B);
    run("16x16x4 2 chains 8 waves/CU", [&](int it) { hipLaunchKernelGGL(k16<2>, dim3(B), dim3(512), 0, 0, out, it, 1.f, 2.f); }, 2048, 2, 512, B);
    run("32x32x2 1 chain  4 waves/CU", [&](int it) { hipLaunchKernelGGL(k32<1>, dim3(B), dim3(256), 0, 0, out, it, 1.f, 2.f); }, 4096, 1, 256, B);
    run("32x32x2 2 chains 4 waves/CU", [&](int it) { hipLaunchKernelGGL(k32<2>, dim3(B), dim3(256), 0, 0, out, it, 1.f, 2.f); }, 4096, 2, 256, B);
    run("32x32x2 2 chains 8 waves/CU", [&](int it) { hipLaunchKernelGGL(k32<2>, dim3(B), dim3(512), 0, 0, out, it, 1.f, 2.f); }, 4096, 2, 512, B);
    return 0;
}
