// Phase timeline of conv_patch_kernel (s_memtime stamps per wave) on one VGG layer shape, forward, 64 images.
//   cd vae_captioning_amd/csrc && hipcc -w --offload-arch=gfx950 -O3 -std=c++17 -DVC_PATCH_TRACE -I ../../include -I . \
//       ../../tools/probes/patch_trace.hip api.hip -o ../../build/probes/patch_trace     (no libvaecap.so: its copies of the kernels would shadow these)
#include "../../vae_captioning_amd/csrc/conv_patch.hip"
#include <algorithm>
#include <vector>

int main(int argc, char** argv) {
    using namespace vc;
    const int B = 64, H = argc > 1 ? atoi(argv[1]) : 224, C = argc > 2 ? atoi(argv[2]) : 64, N = argc > 3 ? atoi(argv[3]) : 64;
    const int dgrad = argc > 4 ? atoi(argv[4]) : 0;
    const size_t px = (size_t)B * H * H;
    float *x, *w, *wp, *bias, *y, *tw;
    hipMalloc(&x, px * C * 4); hipMalloc(&y, px * N * 4); hipMalloc(&w, 9 * C * N * 4); hipMalloc(&wp, 9 * C * N * 4); hipMalloc(&bias, N * 4);
    hipMemset(x, 0, px * C * 4); hipMemset(w, 0, 9 * C * N * 4); hipMemset(bias, 0, N * 4);
    const size_t twb = vc_conv3x3_packed_workspace_bytes(B, H, H, C, N, 0);
    hipMalloc(&tw, twb + 16);
    vc_conv3x3_pack_f32(0, C, N, w, 0, wp);
    const PatchPlan p = plan_patch(B, H, H, C, N);
    const int wgs = 40000;
    unsigned long long* tr;
    hipMalloc(&tr, (size_t)wgs * 4 * 8 * 8);
    for (int rep = 0; rep < 3; ++rep) {
        unsigned long long* on = rep == 2 ? tr : nullptr;
        hipMemcpyToSymbol(HIP_SYMBOL(vc::g_patch_trace), &on, sizeof(on));
        hipMemset(tr, 0, (size_t)wgs * 4 * 8 * 8);
        int rc = dgrad ? vc_conv3x3_dgrad_packed_f32(0, B, H, H, N, C, x, wp, y, y, tw, twb) : vc_conv3x3_fwd_packed_f32(0, B, H, H, C, N, x, wp, bias, y, 1, tw, twb);
        if (rc) printf("rc %d %s\n", rc, vc_last_error());
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h((size_t)wgs * 4 * 8);
    hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost);
    printf("conv %dx%d %d->%d TN=%d scheme=%d: per wave, cycles since the wave's own start (median / p10 / p90 over waves of the MAIN launch)\n", H, H, C, N, p.TN, p.scheme);
    const char* names[] = {"start", "geometry done (before first loads)", "first patch stored", "first barrier passed", "main loop done", "epilogue done"};
    for (int k = 1; k < 6; ++k) {
        std::vector<double> v;
        for (size_t w0 = 0; w0 < (size_t)wgs * 4; ++w0)
            if (h[w0 * 8] && h[w0 * 8 + k]) v.push_back((double)(h[w0 * 8 + k] - h[w0 * 8]));
        if (v.empty()) continue;
        std::sort(v.begin(), v.end());
        printf("  %-38s %9.0f %9.0f %9.0f   (%zu waves)\n", names[k], v[v.size() / 2], v[v.size() / 10], v[v.size() * 9 / 10], v.size());
    }
    return 0;
}
