// Probe: how fast can the Adam update stream its seven arrays (p, g, m, v read; p, m, v written = 28 B per parameter)?
//   hipcc --offload-arch=gfx950 -O3 tools/probes/adam_stream.hip -o build/probes/adam_stream && ./build/probes/adam_stream
// Variants: groups of 16 bytes in flight per thread (2 = csrc/optim.hip), non-temporal loads / stores, grid size.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int G, bool NT>
__global__ __launch_bounds__(256) void adam_probe(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n4) {
    float4* p4 = (float4*)p; const float4* g4 = (const float4*)g; float4* m4 = (float4*)m; float4* v4 = (float4*)v;
    const long stride = (long)gridDim.x * 256;
    auto ld = [](const float4* q) -> float4 {
        if (NT) { float4 r; r.x = __builtin_nontemporal_load(&q->x); r.y = __builtin_nontemporal_load(&q->y); r.z = __builtin_nontemporal_load(&q->z); r.w = __builtin_nontemporal_load(&q->w); return r; }
        return *q;
    };
    auto st = [](float4* q, const float4& r) {
        if (NT) { __builtin_nontemporal_store(r.x, &q->x); __builtin_nontemporal_store(r.y, &q->y); __builtin_nontemporal_store(r.z, &q->z); __builtin_nontemporal_store(r.w, &q->w); }
        else *q = r;
    };
    auto upd = [](float& pp, float gg, float& mm, float& vv) {
        mm = 0.8f * mm + 0.2f * gg;
        vv = 0.999f * vv + 0.001f * gg * gg;
        pp -= 1e-5f * mm / (sqrtf(vv) + 1e-8f);
    };
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += G * stride) {
        float4 P[G], M[G], V[G], Gr[G];
#pragma unroll
        for (int k = 0; k < G; ++k) { const long j = i + k * stride; if (j < n4) { P[k] = ld(p4 + j); M[k] = ld(m4 + j); V[k] = ld(v4 + j); Gr[k] = ld(g4 + j); } }
#pragma unroll
        for (int k = 0; k < G; ++k) {
            const long j = i + k * stride;
            if (j < n4) {
                upd(P[k].x, Gr[k].x, M[k].x, V[k].x); upd(P[k].y, Gr[k].y, M[k].y, V[k].y); upd(P[k].z, Gr[k].z, M[k].z, V[k].z); upd(P[k].w, Gr[k].w, M[k].w, V[k].w);
                st(p4 + j, P[k]); st(m4 + j, M[k]); st(v4 + j, V[k]);
            }
        }
    }
}

template <int G, bool NT>
static void run(const char* name, float* p, float* g, float* m, float* v, long n, int grid) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((adam_probe<G, NT>), dim3(grid), dim3(256), 0, 0, p, g, m, v, n / 4);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((adam_probe<G, NT>), dim3(grid), dim3(256), 0, 0, p, g, m, v, n / 4);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("%-28s grid %5d: %.3f ms  %.2f TB/s\n", name, grid, ms, 28.0 * n / ms / 1e9);
}

int main() {
    const long n = 119545856L + 14714688L;   // the VGG16 parameters
    float *p, *g, *m, *v;
    hipMalloc(&p, n * 4); hipMalloc(&g, n * 4); hipMalloc(&m, n * 4); hipMalloc(&v, n * 4);
    hipMemset(p, 0, n * 4); hipMemset(g, 0, n * 4); hipMemset(m, 0, n * 4); hipMemset(v, 0, n * 4);
    for (int grid : {1024, 2048, 4096, 8192}) {
        run<1, false>("1 group", p, g, m, v, n, grid);
        run<2, false>("2 groups (optim.hip)", p, g, m, v, n, grid);
        run<4, false>("4 groups", p, g, m, v, n, grid);
        run<2, true>("2 groups, non-temporal", p, g, m, v, n, grid);
        run<4, true>("4 groups, non-temporal", p, g, m, v, n, grid);
    }
    return 0;
}
