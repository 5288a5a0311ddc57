// Winograd F(3x3, 2x2) weight gradient of the 3x3 convolutions for gfx950 (utils/image_embeddings.py:36-212, backward of
// tf.nn.conv2d w.r.t. the filter); the forward / data gradient F(2x2, 3x3) kernels and the derivation are in conv_wino.hip.
#include <stdlib.h>
#include "conv_wino_wgrad_kernel.h"

namespace vc {

// ---- weight gradient: F(3x3, 2x2) ---------------------------------------------------------------------------------------------
//   dW[ky][kx][c][n] = sum_pixels x[p + (ky-1, kx-1)][c] dy[p][n]
//                    = A'^T [ sum_tiles (B^T d B) (.) (G' e G'^T) ] A'      d = the tile's 4 x 4 input patch, e = its 2 x 2 dy values,
//   G' = [1 0; 1 1; 1 -1; 0 1], A'^T = [1 1/2 1/2 0; 0 1/2 -1/2 0; 0 1/2 1/2 -1] (B^T as in the forward; checked numerically, tests/).
// The contraction index of the sixteen position products S_p[c][n] = sum_t U_p[t][c] E_p[t][n] is the TILE: a wave owns 32 input
// channels x 32 output channels x sixteen positions (256 accumulator registers), a workgroup 64 x 64, and walks a range of blocks
// (the forward's tile blocks: one block = one K-chunk of TBH/2 * TBW MFMA steps, two tiles per step -- lane half lh takes tile row
// 2r + lh, consecutive steps move along x so that two of the four patch columns' vertical transforms carry over).  Both operands are
// transformed in registers from LDS (x: halo patch [pixel][64 c]; dy: [pixel][64 n]; double-buffered, refilled for the next block under
// the MFMAs); raw position sums go to the workspace per K split, wino_wgrad_reduce_kernel sums the splits in fixed order, applies
// A'^T . A' and writes dw (and db = sum of E_(1,1) = sum of dy, gathered on the way).
// sum of the K splits (fixed order) + dW = A'^T S A' (+ db).  A block owns 64 consecutive (c, n) elements: thread (q, e) sums positions
// 4 q .. 4 q + 3 of element e over the splits (256-byte rows of the workspace), the sums meet in LDS, threads (ky, e) transform.
__global__ __launch_bounds__(256) void wino_wgrad_reduce_kernel(const float* __restrict__ ws, int nsplit, int C, int N, float* __restrict__ dw,
                                                                float* __restrict__ db, int accumulate) {
    __shared__ float S[16][64];
    const long CN = (long)C * N;
    if (db && blockIdx.x == gridDim.x - 1) {
        const float* bw = ws + (long)nsplit * 16 * CN;
        for (int i = threadIdx.x; i < N; i += 256) {
            float v = 0.f;
            for (int z = 0; z < nsplit; ++z) v += bw[(long)z * N + i];
            db[i] = accumulate ? db[i] + v : v;
        }
        return;
    }
    const int e = threadIdx.x & 63, q = threadIdx.x >> 6;
    const long i = (long)blockIdx.x * 64 + e;   // CN % 64 == 0
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    const float* src = ws + (long)(4 * q) * CN + i;
#pragma unroll 4
    for (int z = 0; z < nsplit; ++z) {
        const float* sz = src + (long)z * 16 * CN;
        v0 += sz[0]; v1 += sz[CN]; v2 += sz[2 * CN]; v3 += sz[3 * CN];
    }
    S[4 * q][e] = v0; S[4 * q + 1][e] = v1; S[4 * q + 2][e] = v2; S[4 * q + 3][e] = v3;
    __syncthreads();
    if (q < 3) {   // output row ky = q: R[ky][nu] over xi, then the same over nu
        const int ky = q;
        float R[4];
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) {
            const float h = 0.5f * (S[4 + nu][e] + S[8 + nu][e]), d = 0.5f * (S[4 + nu][e] - S[8 + nu][e]);
            R[nu] = ky == 0 ? S[nu][e] + h : ky == 1 ? d : h - S[12 + nu][e];
        }
        const float h = 0.5f * (R[1] + R[2]), d = 0.5f * (R[1] - R[2]);
        const float w0 = R[0] + h, w1 = d, w2 = h - R[3];
        float* o = dw + (long)(ky * 3) * CN + i;
        if (accumulate) { o[0] += w0; o[CN] += w1; o[2 * CN] += w2; }
        else { o[0] = w0; o[CN] = w1; o[2 * CN] = w2; }
    }
}

struct WinoWgPlan {
    bool ok;
    int shape;   // 0: 4 x 8, 1: 4 x 7, 2: 2 x 14 tiles per block
    WinoGeom g;
    int ncb, nnb, nsplit, cps;
};

static WinoWgPlan plan_wino_wgrad(int B, int H, int W, int C, int N) {
    WinoWgPlan p;
    p.ok = false; p.shape = 0; p.ncb = p.nnb = p.nsplit = p.cps = 0;
    WinoGeom& g = p.g;
    g.B = B; g.H = H; g.W = W; g.C = C; g.N = N;
    if (B <= 0 || H < 4 || W < 2 || (H & 1) || (W & 1) || C <= 0 || N <= 0 || C % 64 || N % 64) return p;
    if ((long)B * H * W * (long)(C > N ? C : N) * 4 > 0x7fffffffL) return p;
    const int TW = W / 2, TH = H / 2;
    static const int shapes[3][2] = {{4, 8}, {4, 7}, {2, 14}};
    double best = 0.0;
    for (int k = 0; k < 3; ++k) {
        const int tbh = shapes[k][0], tbw = shapes[k][1];
        const double eff = (double)TW * TH / ((double)cdiv(TW, tbw) * cdiv(TH, tbh) * tbh * tbw);
        if (eff > best + 1e-9) { best = eff; p.shape = k; }
    }
    g.TBH = shapes[p.shape][0]; g.TBW = shapes[p.shape][1];
    g.PW = 2 * g.TBW + 2; g.PH = 2 * g.TBH + 2;
    g.bx_n = cdiv(TW, g.TBW); g.by_n = cdiv(TH, g.TBH);
    g.blocks_img = g.bx_n * g.by_n;
    if ((long)B * g.blocks_img > 0x3fffffffL) return p;
    g.nblocks = B * g.blocks_img;
    p.ncb = C / 64; p.nnb = N / 64;
    int ns = cdiv(256, p.ncb * p.nnb);
    if (ns > g.nblocks) ns = g.nblocks;
    p.cps = cdiv(g.nblocks, ns);
    p.nsplit = cdiv(g.nblocks, p.cps);
    p.ok = true;
    return p;
}

static size_t wino_wgrad_ws(const WinoWgPlan& p) {
    return p.ok ? ((size_t)p.nsplit * 16 * p.g.C * p.g.N + (size_t)p.nsplit * p.g.N) * sizeof(float) : 0;
}

}  // namespace vc

extern "C" int vc_conv3x3_wino_wgrad_supported(int B, int H, int W, int Cin, int Cout) {
    const int nb = vc::wino_images_per_launch(B, H, W, Cin, Cout);
    return nb > 0 && vc::plan_wino_wgrad(nb, H, W, Cin, Cout).ok ? 1 : 0;
}

// the split count is not monotone in the block count (nsplit = cdiv(nblocks, cdiv(nblocks, ns))): the remainder launch of a call that
// is cut into image ranges may need MORE workspace than the full ranges -- the reported size is the maximum over both plans
static size_t wino_wgrad_ws_call(int B, int H, int W, int Cin, int Cout) {
    const int per = vc::wino_images_per_launch(B, H, W, Cin, Cout);
    if (per <= 0) return 0;
    size_t need = vc::wino_wgrad_ws(vc::plan_wino_wgrad(per, H, W, Cin, Cout));
    if (B % per) {
        const size_t r = vc::wino_wgrad_ws(vc::plan_wino_wgrad(B % per, H, W, Cin, Cout));
        if (r > need) need = r;
    }
    return need;
}

extern "C" size_t vc_conv3x3_wino_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
    return wino_wgrad_ws_call(B, H, W, Cin, Cout);
}

extern "C" int vc_conv3x3_wino_wgrad_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* x, const float* dy, float* dw,
                                         float* db, int accumulate, float* ws, size_t ws_bytes) {
    using namespace vc;
    const int per = wino_images_per_launch(B, H, W, Cin, Cout);
    VC_CHECK_ARG(per > 0 && plan_wino_wgrad(per, H, W, Cin, Cout).ok, "unsupported shape (vc_conv3x3_wino_wgrad_supported)");
    VC_CHECK_ARG(x && dy && dw && ws, "null pointer");
    VC_CHECK_ARG(waligned16(x) && waligned16(dy) && waligned16(ws), "pointers must be 16-byte aligned");
    // every image range is validated BEFORE the first launch: a failure must not leave dw / db half accumulated
    VC_CHECK_ARG(B % per == 0 || plan_wino_wgrad(B % per, H, W, Cin, Cout).ok, "unsupported shape (vc_conv3x3_wino_wgrad_supported)");
    if (ws_bytes < wino_wgrad_ws_call(B, H, W, Cin, Cout))
        return fail(VC_EWORKSPACE, "%s: workspace too small (%ld < %ld bytes)", __func__, (long)ws_bytes, (long)wino_wgrad_ws_call(B, H, W, Cin, Cout));
    for (int b0 = 0; b0 < B; b0 += per) {   // image ranges of < 2 GiB; the later ranges accumulate into dw / db
        const int nbi = B - b0 < per ? B - b0 : per;
        const WinoWgPlan p = plan_wino_wgrad(nbi, H, W, Cin, Cout);
        VC_CHECK_ARG(p.ok, "unsupported shape (vc_conv3x3_wino_wgrad_supported)");
        if (ws_bytes < wino_wgrad_ws(p)) return fail(VC_EWORKSPACE, "%s: workspace too small (%ld < %ld bytes)", __func__, (long)ws_bytes, (long)wino_wgrad_ws(p));
        WinoWgArgs a;
        a.g = p.g; a.x = x + (size_t)b0 * H * W * Cin; a.dy = dy + (size_t)b0 * H * W * Cout; a.ws = ws;
        a.ncb = p.ncb; a.nnb = p.nnb; a.nsplit = p.nsplit; a.cps = p.cps;
        static const int xcd_on = getenv("VC_WGRAD_XCD") ? atoi(getenv("VC_WGRAD_XCD")) : 1;   // (A/B runs)
        a.xcd = xcd_on;
        int rc = p.shape == 0 ? launch_wino_wgrad_4x8((hipStream_t)stream, a) : p.shape == 1 ? launch_wino_wgrad_4x7((hipStream_t)stream, a)
                                                                                             : launch_wino_wgrad_2x14((hipStream_t)stream, a);
        if (rc) return rc;
        const int grid = (int)((long)Cin * Cout / 64);
        hipLaunchKernelGGL(wino_wgrad_reduce_kernel, dim3(grid + (db ? 1 : 0)), dim3(256), 0, (hipStream_t)stream, ws, p.nsplit, Cin, Cout, dw, db,
                           accumulate || b0 > 0);
        rc = launch_status(__func__);
        if (rc) return rc;
    }
    return 0;
}
