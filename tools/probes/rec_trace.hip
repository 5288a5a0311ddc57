// Phase timeline of the LSTM recurrence step kernels (s_memtime stamps per wave), one launch each at N rows.
//   hipcc -w --offload-arch=gfx950 -O3 -std=c++17 -I include -I vae_captioning_amd/csrc tools/probes/rec_trace.hip \
//         -L vae_captioning_amd/lib -lvaecap -Wl,-rpath,'$ORIGIN/../../vae_captioning_amd/lib' -o build/probes/rec_trace
#define VC_REC_TRACE 1
#include "../../vae_captioning_amd/csrc/lstm.hip"  // (the other csrc files are compiled into the probe too: no libvaecap.so, whose copies of these kernels would shadow the traced ones)
#include <vector>
#include <algorithm>

static void report(const char* name, const std::vector<unsigned long long>& tr, int wgs, int nst) {
    // per stamp: median over (workgroup, wave) of (stamp - stamp 0 of the same wave); also the earliest stamp 0 and the latest end
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int w = 0; w < wgs * 4; ++w) {
        if (tr[w * 32] && tr[w * 32] < t0) t0 = tr[w * 32];
        if (tr[w * 32 + 30] > t1) t1 = tr[w * 32 + 30];
    }
    printf("%s: first start -> last end %.2f us (100 MHz s_memtime ticks assumed)\n", name, (t1 - t0) / 100.0);
    const int order[] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 28, 29, 30};
    for (int k : order) {
        std::vector<double> v, st;
        for (int w = 0; w < wgs * 4; ++w)
            if (tr[w * 32 + k]) { v.push_back((tr[w * 32 + k] - tr[w * 32]) / 100.0); st.push_back((tr[w * 32 + k] - t0) / 100.0); }
        if (v.empty()) continue;
        std::sort(v.begin(), v.end());
        std::sort(st.begin(), st.end());
        printf("  stamp %2d: since own start median %6.2f us (min %6.2f max %6.2f); since launch median %6.2f max %6.2f\n", k, v[v.size() / 2], v.front(), v.back(),
               st[st.size() / 2], st.back());
    }
    (void)nst;
}

int main(int argc, char** argv) {
    using namespace vc;
    const int N = argc > 1 ? atoi(argv[1]) : 320, H = 512;
    float *h, *c, *Wh, *g, *co, *ho, *whp, *dG, *dH, *dC, *act;
    int* lens;
    hipMalloc(&h, N * H * 4); hipMalloc(&c, N * H * 4); hipMalloc(&Wh, H * 4 * H * 4); hipMalloc(&g, (size_t)N * 4 * H * 4);
    hipMalloc(&co, N * H * 4); hipMalloc(&ho, N * H * 4); hipMalloc(&whp, H * 4 * H * 4); hipMalloc(&lens, N * 4);
    hipMalloc(&dG, (size_t)N * 4 * H * 4); hipMalloc(&dH, N * H * 4); hipMalloc(&dC, N * H * 4); hipMalloc(&act, (size_t)N * 4 * H * 4);
    hipMemset(h, 0, N * H * 4); hipMemset(c, 0, N * H * 4); hipMemset(Wh, 0, H * 4 * H * 4); hipMemset(g, 0, (size_t)N * 4 * H * 4);
    hipMemset(dG, 0, (size_t)N * 4 * H * 4); hipMemset(dH, 0, N * H * 4); hipMemset(dC, 0, N * H * 4); hipMemset(act, 0, (size_t)N * 4 * H * 4);
    std::vector<int> l(N, 100);
    hipMemcpy(lens, l.data(), N * 4, hipMemcpyHostToDevice);
    unsigned long long* tr;
    const int WG = 256;
    hipMalloc(&tr, WG * 4 * 32 * 8);
    std::vector<unsigned long long> host(WG * 4 * 32);
    for (int dir = 0; dir < 2; ++dir) {
        for (int rep = 0; rep < 3; ++rep) {
            unsigned long long* on = rep == 2 ? tr : nullptr;
            { hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(vc::g_rec_trace), &on, sizeof(on)); if (e) printf("symbol: %s\n", hipGetErrorString(e)); }
            hipMemset(tr, 0, WG * 4 * 32 * 8);
            if (dir == 0) {
                hipLaunchKernelGGL(lstm_rec_pack_fwd_kernel, dim3(1024), dim3(256), 0, 0, Wh, H, (float4*)whp);
                LstmFwdArgs a{h, c, Wh, g, lens, co, ho, N, H, 1};
                for (int k = 0; k < 3; ++k) { int rc = rec_fwd(0, a, whp); if (rc) printf("rec_fwd rc %d %s\n", rc, last_error_buf()); }
            } else {
                hipLaunchKernelGGL(lstm_rec_pack_bwd_kernel, dim3(1024), dim3(256), 0, 0, Wh, H, (float4*)whp);
                LstmBwdArgs a{dG, Wh, lens, nullptr, dH, dC, act, c, co, g, N, H, 1, 0};
                for (int k = 0; k < 3; ++k) { int rc = rec_bwd(0, a, whp); if (rc) printf("rec_bwd rc %d %s\n", rc, last_error_buf()); }
            }
            { hipError_t e = hipDeviceSynchronize(); if (e) printf("sync: %s\n", hipGetErrorString(e)); }
        }
        hipMemcpy(host.data(), tr, WG * 4 * 32 * 8, hipMemcpyDeviceToHost);
        report(dir == 0 ? "fwd" : "bwd", host, WG, 32);
    }
    return 0;
}
