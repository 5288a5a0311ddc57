// Phase timeline of conv_wino_kernel (cycle-counter stamps per wave) on one VGG layer shape, forward, 64 images.
//   cd vae_captioning_amd/csrc && hipcc -w --offload-arch=gfx950 -O3 -std=c++17 -DVC_WINO_TRACE -I ../../include -I . \
//       ../../tools/probes/wino_trace.hip api.hip -o ../../build/probes/wino_trace     (no libvaecap.so: its copies of the kernels would shadow these)
#include "../../vae_captioning_amd/csrc/conv_wino.hip"
#include <algorithm>
#include <vector>

int main(int argc, char** argv) {
    using namespace vc;
    const int B = 64, H = argc > 1 ? atoi(argv[1]) : 56, C = argc > 2 ? atoi(argv[2]) : 256, N = argc > 3 ? atoi(argv[3]) : 256;
    const size_t px = (size_t)B * H * H;
    float *x, *w, *vp, *bias, *y;
    hipMalloc(&x, px * C * 4); hipMalloc(&y, px * N * 4); hipMalloc(&w, 9 * C * N * 4); hipMalloc(&vp, 16 * C * N * 4); hipMalloc(&bias, N * 4);
    hipMemset(x, 0, px * C * 4); hipMemset(w, 0, 9 * C * N * 4); hipMemset(bias, 0, N * 4);
    vc_conv3x3_wino_pack_f32(0, C, N, w, 0, vp);
    const int wgs = 40000;
    unsigned long long* tr;
    hipMalloc(&tr, (size_t)wgs * 4 * 8 * 8);
    for (int rep = 0; rep < 3; ++rep) {
        unsigned long long* on = rep == 2 ? tr : nullptr;
        hipMemcpyToSymbol(HIP_SYMBOL(vc::g_wino_trace), &on, sizeof(on));
        hipMemset(tr, 0, (size_t)wgs * 4 * 8 * 8);
        int rc = vc_conv3x3_wino_fwd_f32(0, B, H, H, C, N, x, vp, bias, y, nullptr, 1);
        if (rc) printf("rc %d %s\n", rc, vc_last_error());
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h((size_t)wgs * 4 * 8);
    hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost);
    printf("wino fwd %dx%d %d->%d (%d chunks of 128 MFMAs = %d MFMA cycles per wave): cycles since the wave's own start (median / p10 / p90 over waves)\n", H, H, C, N, C / 16,
           C / 16 * 128 * 64);
    const char* names[] = {"start", "geometry done (before the first loads)", "first half stored to LDS", "first barrier passed", "first unit prepared (loop starts)", "main loop done",
                           "epilogue done"};
    for (int k = 1; k < 7; ++k) {
        std::vector<double> v;
        for (size_t w0 = 0; w0 < (size_t)wgs * 4; ++w0)
            if (h[w0 * 8] && h[w0 * 8 + k]) v.push_back((double)(h[w0 * 8 + k] - h[w0 * 8]));
        if (v.empty()) continue;
        std::sort(v.begin(), v.end());
        printf("  %-42s %9.0f %9.0f %9.0f   (%zu waves)\n", names[k], v[v.size() / 2], v[v.size() / 10], v[v.size() * 9 / 10], v.size());
    }
    return 0;
}
