// Winograd F(3x3, 2x2) weight gradient of the 3x3 convolutions for gfx950 (utils/image_embeddings.py:36-212, backward of
// tf.nn.conv2d w.r.t. the filter); the forward / data gradient F(2x2, 3x3) kernels and the derivation are in conv_wino.hip.
#include <stdlib.h>
#include "conv_wino.h"

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
struct WinoWgArgs {
    WinoGeom g;          // C = input channels (x), N = output channels (dy)
    const float* x;      // [P, C]
    const float* dy;     // [P, N]
    float* ws;           // [split][16][C][N] position sums, then [split][N] bias partials
    int ncb, nnb, nsplit, cps;
};

template <int TBH, int TBW>
__global__ __launch_bounds__(256, 1) void wino_wgrad_kernel(WinoWgArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int PW = 2 * TBW + 2, PH = 2 * TBH + 2, DW = 2 * TBW, DH = 2 * TBH, NS = (TBH / 2) * TBW;
    constexpr int XPF = 180 * 64, BUF = XPF + 128 * 64;   // floats: x patch, dy tile block; two buffers
    static_assert(PW * PH <= 180 && DW * DH <= 128 && NS <= 16 && NS >= 8, "block shape");
    const WinoGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int cw = wave >> 1, nw = wave & 1;
    const int cn = blockIdx.x % (a.ncb * a.nnb), split = blockIdx.x / (a.ncb * a.nnb);
    const int cb = cn / a.nnb, nb = cn - cb * a.nnb;
    const int C = g.C, N = g.N;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)((long)g.B * g.H * g.W * C * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, (int)((long)g.B * g.H * g.W * N * 4), 0x00020000);

    const int blk0 = split * a.cps;
    int nch = g.nblocks - blk0;
    if (nch > a.cps) nch = a.cps;

    // staging slots: x patch 180 pixels x 16 channel quads = 2880 float4 (slots 0..11 of a thread), dy 128 pixels x 16 quads = 2048
    // (slots 12..19); slot i of a thread: float4 index tid + 256 i (x) / tid + 256 (i - 12) (dy): pixel index / 16, quad index % 16
    float4 st[10];
    int by0 = 0, bx0 = 0, bimg = 0;   // current block to LOAD (uniform)
    auto set_block = [&](int blk) {
        const unsigned b = (unsigned)blk / (unsigned)g.blocks_img, rem = (unsigned)blk - b * (unsigned)g.blocks_img;
        const unsigned by = rem / (unsigned)g.bx_n, bx = rem - by * (unsigned)g.bx_n;
        bimg = (int)b; by0 = (int)by * DH; bx0 = (int)bx * DW;   // first output pixel of the block
    };
    auto gload1 = [&](int i, int k) {   // slot i into st[k]
        if (i < 12) {
            const int s = tid + 256 * i, pix = s >> 4, quad = s & 15;
            const int py = pix / PW, px = pix - py * PW;
            const int y = by0 - 1 + py, x = bx0 - 1 + px;
            const bool ok = s < 180 * 16 && pix < PW * PH && (unsigned)y < (unsigned)g.H && (unsigned)x < (unsigned)g.W;
            const unsigned off = ok ? (unsigned)((((bimg * g.H + y) * g.W + x) * C + cb * 64 + quad * 4) * 4) : WOOB;
            st[k] = wbufload(rx, off, 0);
        } else {
            const int s = tid + 256 * (i - 12), pix = s >> 4, quad = s & 15;
            const int py = pix / DW, px = pix - py * DW;
            const int y = by0 + py, x = bx0 + px;
            const bool ok = pix < DW * DH && y < g.H && x < g.W;
            const unsigned off = ok ? (unsigned)((((bimg * g.H + y) * g.W + x) * N + nb * 64 + quad * 4) * 4) : WOOB;
            st[k] = wbufload(ry, off, 0);
        }
    };
    auto lstore1 = [&](int buf, int i, int k) {
        if (i < 12) {
            if (i < 11 || tid + 256 * i < 180 * 16) *reinterpret_cast<float4*>(&smem[buf * BUF + (tid + 256 * i) * 4]) = st[k];
        } else {
            *reinterpret_cast<float4*>(&smem[buf * BUF + XPF + (tid + 256 * (i - 12)) * 4]) = st[k];
        }
    };

    f32x16 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
    float dbacc = 0.f;

    // operands of one step: ua[p] = U_p[tile][c], eb[p] = E_p[tile][n]; dv / ev: raw reads; tc[j][xi]: vertical transform of patch column j
    float ua[2][16], eb[2][16], dv[4][4], ev[2][2], tc[4][4], tv[2][4];
    const int xbase = ((2 * lh) * PW) * 64 + cw * 32 + li;          // + buf * BUF + ((4 r + i) * PW + 2 tx + j) * 64
    const int ybase = XPF + ((2 * lh) * DW) * 64 + nw * 32 + li;    // + buf * BUF + ((4 r + a) * DW + 2 tx + b) * 64
    // micro-operation k of preparing step sn (from LDS buffer `buf`) into operand set ob
    auto prep = [&](int buf, int sn, int ob, int k) {
        const int r = sn / TBW, tx = sn - r * TBW;
        const bool fresh = tx == 0;
        const int nread = fresh ? 16 : 8;
        if (k < nread) {
            const int idx = fresh ? k : 8 + k, j = idx >> 2, i = idx & 3;
            dv[i][j] = smem[buf * BUF + xbase + ((4 * r + i) * PW + 2 * tx + j) * 64];
            return;
        }
        k -= nread;
        if (k < 4) {
            const int aa = k >> 1, bb = k & 1;
            ev[aa][bb] = smem[buf * BUF + ybase + ((4 * r + aa) * DW + 2 * tx + bb) * 64];
            return;
        }
        k -= 4;
        if (k < nread) {   // vertical transforms of the new patch columns (the two older ones carry over from the previous step)
            const int idx = fresh ? k : 8 + k, j = idx >> 2, xi = idx & 3;
            if (!fresh && k == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) { tc[0][q] = tc[2][q]; tc[1][q] = tc[3][q]; }
            }
            tc[j][xi] = xi == 0 ? dv[0][j] - dv[2][j] : xi == 1 ? dv[1][j] + dv[2][j] : xi == 2 ? dv[2][j] - dv[1][j] : dv[1][j] - dv[3][j];
            return;
        }
        k -= nread;
        if (k < 16) {
            const int xi = k >> 2, nu = k & 3;
            ua[ob][k] = nu == 0 ? tc[0][xi] - tc[2][xi] : nu == 1 ? tc[1][xi] + tc[2][xi] : nu == 2 ? tc[2][xi] - tc[1][xi] : tc[1][xi] - tc[3][xi];
            return;
        }
        k -= 16;
        if (k < 4) {       // vertical G' of dy column bb = k >> 1: rows 1 (sum), 2 (difference); rows 0 and 3 are e[0][bb], e[1][bb]
            const int bb = k >> 1;
            if (k & 1) tv[bb][2] = ev[0][bb] - ev[1][bb];
            else { tv[bb][1] = ev[0][bb] + ev[1][bb]; tv[bb][0] = ev[0][bb]; tv[bb][3] = ev[1][bb]; }
            return;
        }
        k -= 4;
        if (k < 8) {
            const int xi = k >> 1;
            if (k & 1) eb[ob][xi * 4 + 2] = tv[0][xi] - tv[1][xi];
            else { eb[ob][xi * 4 + 1] = tv[0][xi] + tv[1][xi]; eb[ob][xi * 4 + 0] = tv[0][xi]; eb[ob][xi * 4 + 3] = tv[1][xi]; }
            return;
        }
        k -= 8;
        if (k == 0) dbacc += eb[ob][5];   // E_(1,1) = the sum of the tile's four dy values
    };
#define WSB() __builtin_amdgcn_sched_barrier(0)
    if (nch > 0) {
        set_block(blk0);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int k = 0; k < 10; ++k) gload1(10 * h + k, k);
#pragma unroll
            for (int k = 0; k < 10; ++k) lstore1(0, 10 * h + k, k);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 65; ++k) prep(0, 0, 0, k);
        WSB();
    }
    for (int ci = 0; ci < nch; ++ci) {
        const int buf = ci & 1;
        const bool more = ci + 1 < nch;
        if (more) set_block(blk0 + ci + 1);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int ob = s & 1;
            const bool last_step = s == NS - 1;
            if (last_step && more) __syncthreads();   // this buffer's last reads are behind every wave, the other buffer is written
            const bool nxt = !last_step || more;
            const int sn = last_step ? 0 : s + 1, nbuf = last_step ? buf ^ 1 : buf, nob = (NS & 1) && last_step ? ob ^ 1 : ob ^ 1;
            const int total = (sn % TBW) == 0 ? 65 : 49, per = (total + 15) / 16;
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[ob][m], eb[ob][m], acc[m], 0, 0, 0);
                WSB();
                if (nxt) {
#pragma unroll
                    for (int k2 = 0; k2 < 5; ++k2)
                        if (k2 < per && m * per + k2 < total) prep(nbuf, sn, nob, m * per + k2);
                }
                if (more && m < 10) {   // the next block's data: two batches of ten slots, loaded early, written a few steps later
                    if (s == 0) gload1(m, m);
                    if (s == NS / 2 - 2) lstore1(buf ^ 1, m, m);
                    if (s == NS / 2 - 1) gload1(10 + m, m);
                    if (s == NS - 3) lstore1(buf ^ 1, 10 + m, m);
                }
                WSB();
            }
        }
    }
#undef WSB
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");

    // raw position sums of this split: acc[p][r] = S_p[c0 + 8 (r >> 2) + 4 lh + (r & 3)][n0 + li]
    const int c0 = cb * 64 + cw * 32, n0 = nb * 64 + nw * 32;
    float* o = a.ws + (long)split * 16 * C * N + (long)n0 + li;
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[((long)p * C + c0 + 8 * (r >> 2) + 4 * lh + (r & 3)) * N] = acc[p][r];
    if (cb == 0 && cw == 0) {
        const float v = dbacc + __shfl_xor(dbacc, 32, 64);
        if (lh == 0) a.ws[(long)a.nsplit * 16 * C * N + (long)split * N + n0 + li] = v;
    }
}

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

constexpr int WINO_WG_LDS_BYTES = 2 * (180 * 64 + 128 * 64) * 4;

template <int TBH, int TBW>
static int launch_wino_wgrad(hipStream_t st, const WinoWgArgs& a) {
    static int once = [] {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wino_wgrad_kernel<TBH, TBW>), hipFuncAttributeMaxDynamicSharedMemorySize, WINO_WG_LDS_BYTES);
        return e == hipSuccess ? 0 : fail((int)e, "%s: hipFuncSetAttribute failed", "wino wgrad kernel");
    }();
    if (once) return once;
    hipLaunchKernelGGL((wino_wgrad_kernel<TBH, TBW>), dim3(a.ncb * a.nnb * a.nsplit), dim3(256), WINO_WG_LDS_BYTES, st, a);
    return launch_status("conv wino wgrad");
}

}  // namespace vc

extern "C" int vc_conv3x3_wino_wgrad_supported(int B, int H, int W, int Cin, int Cout) { return vc::plan_wino_wgrad(B, H, W, Cin, Cout).ok ? 1 : 0; }

extern "C" size_t vc_conv3x3_wino_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
    return vc::wino_wgrad_ws(vc::plan_wino_wgrad(B, H, W, Cin, Cout));
}

extern "C" int vc_conv3x3_wino_wgrad_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* x, const float* dy, float* dw,
                                         float* db, int accumulate, float* ws, size_t ws_bytes) {
    using namespace vc;
    const WinoWgPlan p = plan_wino_wgrad(B, H, W, Cin, Cout);
    VC_CHECK_ARG(p.ok, "unsupported shape (vc_conv3x3_wino_wgrad_supported)");
    VC_CHECK_ARG(x && dy && dw && ws, "null pointer");
    VC_CHECK_ARG(waligned16(x) && waligned16(dy) && waligned16(ws), "pointers must be 16-byte aligned");
    if (ws_bytes < wino_wgrad_ws(p)) return fail(VC_EWORKSPACE, "%s: workspace too small (%ld < %ld bytes)", __func__, (long)ws_bytes, (long)wino_wgrad_ws(p));
    WinoWgArgs a;
    a.g = p.g; a.x = x; a.dy = dy; a.ws = ws; a.ncb = p.ncb; a.nnb = p.nnb; a.nsplit = p.nsplit; a.cps = p.cps;
    int rc = p.shape == 0 ? launch_wino_wgrad<4, 8>((hipStream_t)stream, a) : p.shape == 1 ? launch_wino_wgrad<4, 7>((hipStream_t)stream, a)
                                                                                           : launch_wino_wgrad<2, 14>((hipStream_t)stream, a);
    if (rc) return rc;
    const int grid = (int)((long)Cin * Cout / 64);
    hipLaunchKernelGGL(wino_wgrad_reduce_kernel, dim3(grid + (db ? 1 : 0)), dim3(256), 0, (hipStream_t)stream, ws, p.nsplit, Cin, Cout, dw, db, accumulate);
    return launch_status(__func__);
}
