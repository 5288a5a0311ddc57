// Can a VALU-only wave run beside an MFMA-only wave on the SAME SIMD without slowing its matrix stream?  (mfma16_f43.hip shows that
// inside one instruction stream every VALU operation beside v_mfma_f32_16x16x4_f32 costs 4-8 cycles of matrix time, with one or two
// such waves per SIMD.)  512-thread workgroups, one per CU: waves 0-3 (one per SIMD) issue only MFMAs (36 accumulators) and one
// ds_read_b128 every RD-th MFMA; their partners, waves 4-7, issue only v_fma_f32 (R x 10 per ten MFMAs of the partner, in total) and
// one ds_write_b128 per WR operations.  Every wave stamps s_memtime around its loop.
//   hipcc -w --offload-arch=gfx950 -O3 -std=c++17 tools/probes/mfma_specialised.hip -o build/probes/mfma_specialised
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define SB() __builtin_amdgcn_sched_barrier(0)
constexpr int NA = 36;

template <int R10, int RD, int WR, int PRIO>
__global__ __launch_bounds__(512, 1) void k(float* out, unsigned long long* cyc, int iters, int miters) {
    __shared__ __attribute__((aligned(16))) float lds[16384];
    const int wave = threadIdx.x >> 6;
    float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
    for (int i = threadIdx.x; i < 16384; i += 512) lds[i] = a;
    __syncthreads();
    float s = 0.f;
    unsigned long long t0, t1;
    if (wave < 4) {
        f32x4 acc[NA];
#pragma unroll
        for (int p = 0; p < NA; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
        float4 ld[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) ld[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* lp = &lds[(threadIdx.x & 63) * 4];
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < miters; ++it) {
#pragma unroll
            for (int m = 0; m < NA; ++m) {
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m], 0, 0, 0);
                SB();
                if (RD > 0 && (m % RD) == RD - 1) ld[(m / RD) & 3] = *reinterpret_cast<const float4*>(lp + ((m / RD) & 15) * 256);
                SB();
            }
            a += ld[0].x + ld[1].y + ld[2].z + ld[3].w;
        }
        t1 = __builtin_readcyclecounter();
#pragma unroll
        for (int p = 0; p < NA; ++p) s += acc[p][0] + acc[p][1] + acc[p][2] + acc[p][3];
    } else {
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = a + i;
        float* wp = &lds[8192 + (threadIdx.x & 63) * 4];
        constexpr int NV = NA * R10 / 10;   // operations per iteration of the partner
        if (PRIO == 1) __builtin_amdgcn_s_setprio(1);
        if (PRIO == 3) __builtin_amdgcn_s_setprio(3);
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                v[j & 15] = __builtin_fmaf(v[(j + 5) & 15], b, v[j & 15]);
                if (WR > 0 && (j % WR) == WR - 1) {
                    SB();
                    *reinterpret_cast<float4*>(wp + ((j / WR) & 7) * 256) = make_float4(v[0], v[1], v[2], v[3]);
                    SB();
                }
            }
        }
        t1 = __builtin_readcyclecounter();
#pragma unroll
        for (int i = 0; i < 16; ++i) s += v[i];
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int R10, int RD, int WR, int PRIO>
static void run(float* out, unsigned long long* cyc, double ghz) {
    const int iters = 1500;
    std::vector<unsigned long long> h(256 * 8);
    auto launch = [&](int it, int mit, double& cm, double& cv) {
        hipLaunchKernelGGL((k<R10, RD, WR, PRIO>), dim3(256), dim3(512), 0, 0, out, cyc, it, mit);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        cm = cv = 0;
        for (int b = 0; b < 256; ++b)
            for (int w = 0; w < 8; ++w) (w < 4 ? cm : cv) += (double)h[b * 8 + w];
        cm /= 1024.0; cv /= 1024.0;
    };
    double cm, cv, m_alone, dummy, v_alone;
    launch(50, 50, cm, cv);
    launch(0, iters, m_alone, dummy);      // MFMA waves alone (their partners exit at once)
    launch(iters, 0, dummy, v_alone);      // VALU waves alone
    launch(iters, iters, cm, cv);          // together
    const double nm = (double)iters * NA, nv = (double)iters * (NA * R10 / 10);
    // ticks of the constant-rate counter; the matrix rate of the MFMA waves alone calibrates it (measured 36.9 cycles per MFMA)
    const double cyc_per_tick = 36.9 / (m_alone / nm);
    printf("VALU/MFMA %4.1f read/%d write/%d prio %d | alone: %.2f cyc/VALU | together: MFMA waves %.1f cyc/MFMA (%.0f %% of alone)",
           R10 / 10.0, RD, WR, PRIO, v_alone / nv * cyc_per_tick, cm / nm * cyc_per_tick, 100.0 * m_alone / cm);
    if (cv > cm) {   // VALU waves outlast: operations issued while the MFMA waves ran
        const double done = nv - (cv - cm) / (v_alone / nv);
        printf(", VALU waves issued %.2f operations per MFMA beside them\n", done / nm);
    } else {
        printf(", VALU waves done at %.0f %% of the MFMA waves' time: %.2f cyc/VALU\n", 100.0 * cv / cm, cv / nv * cyc_per_tick);
    }
}

int main() {
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&cyc, 256 * 8 * 8);
    const double ghz = 2.4;
    run<47, 2, 0, 0>(out, cyc, ghz);
    run<120, 2, 0, 0>(out, cyc, ghz);
    run<13, 2, 0, 1>(out, cyc, ghz);
    run<26, 2, 0, 1>(out, cyc, ghz);
    run<47, 2, 0, 1>(out, cyc, ghz);
    run<47, 2, 19, 1>(out, cyc, ghz);
    run<120, 2, 19, 1>(out, cyc, ghz);
    run<47, 2, 19, 3>(out, cyc, ghz);
    run<120, 2, 19, 3>(out, cyc, ghz);
    return 0;
}
