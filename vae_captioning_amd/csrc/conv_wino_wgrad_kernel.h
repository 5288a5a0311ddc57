// Winograd F(3x3, 2x2) weight-gradient kernel template (see conv_wino_wgrad.hip for the derivation and the host side).  Each block
// shape is instantiated in its own translation unit (conv_wino_wgrad_<shape>.hip): a fully unrolled K-chunk of 256 MFMAs with their
// hand-placed fillers takes hipcc about a minute per instantiation.
#pragma once
#include "conv_wino.h"

// `make ablate-wgrad` builds the kernels with WG_ABL = a bit mask that REMOVES parts of the main loop (results wrong; timing only):
// 1 transform arithmetic, 2 LDS operand reads, 4 staging (global loads + LDS writes)
#ifndef WG_ABL
#define WG_ABL 0
#endif

namespace vc {

constexpr int WG_XPL = 724, WG_YPL = 516;   // LDS plane pitches (floats) of the x patch (>= 4 * 180) and the dy block (>= 4 * 128)

struct WinoWgArgs {
    WinoGeom g;          // C = input channels (x), N = output channels (dy)
    const float* x;      // [B][C/4][H][W][4] (the C4 activation layout, vaecap.h)
    const float* dy;     // [B][N/4][H][W][4]
    float* ws;           // [split][16][C][N] position sums, then [split][N] bias partials
    int ncb, nnb, nsplit, cps;
};

template <int TBH, int TBW>
__global__ __launch_bounds__(256, 1) void wino_wgrad_kernel(WinoWgArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int PW = 2 * TBW + 2, PH = 2 * TBH + 2, DW = 2 * TBW, DH = 2 * TBH, NS = (TBH / 2) * TBW;
    // LDS: per buffer the x halo patch and the dy block as sixteen CHANNEL-QUAD PLANES [quad][pixel][4] -- the staging writes are
    // 16-byte pieces of consecutive pixels (what the C4 layout delivers: consecutive lanes = consecutive pixels of a patch row), the operand
    // reads (one float per lane: channel li of 32, ds_read_b32, banks mod 32) are conflict-free because the plane pitches are 4 mod 8 dwords x 4
    constexpr int NPX = PW * PH, NDY = DW * DH;
    constexpr int XPL = WG_XPL, YPL = WG_YPL;            // plane pitches in floats (724 = 20 mod 32, 516 = 4 mod 32)
    constexpr int XPF = 16 * XPL, BUF = XPF + 16 * YPL;   // floats: x patch, dy tile block; two buffers
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

    // staging slots: x patch 16 channel quads x NPX <= 180 pixels = 2880 float4 (slots 0..11 of a thread), dy 16 quads x NDY <= 128 pixels
    // (slots 12..19); slot i of a thread: float4 index s = tid + 256 i (x) / tid + 256 (i - 12) (dy): quad s / pixels, pixel s % pixels
    float4 st[10];
    int by0 = 0, bx0 = 0, bimg = 0;   // current block to LOAD (uniform)
    auto set_block = [&](int blk) {
        const unsigned b = (unsigned)blk / (unsigned)g.blocks_img, rem = (unsigned)blk - b * (unsigned)g.blocks_img;
        const unsigned by = rem / (unsigned)g.bx_n, bx = rem - by * (unsigned)g.bx_n;
        bimg = (int)b; by0 = (int)by * DH; bx0 = (int)bx * DW;   // first output pixel of the block
    };
    auto gload1 = [&](int i, int k) {   // slot i into st[k]
        if (i < 12) {
            const int s = tid + 256 * i, quad = s / NPX, pix = s - quad * NPX;
            const int py = pix / PW, px = pix - py * PW;
            const int y = by0 - 1 + py, x = bx0 - 1 + px;
            const bool ok = s < 16 * NPX && (unsigned)y < (unsigned)g.H && (unsigned)x < (unsigned)g.W;
            const unsigned off = ok ? (unsigned)(((((bimg * (C >> 2) + cb * 16 + quad) * g.H + y) * g.W + x)) * 16) : WOOB;
            st[k] = wbufload(rx, off, 0);
        } else {
            const int s = tid + 256 * (i - 12), quad = s / NDY, pix = s - quad * NDY;
            const int py = pix / DW, px = pix - py * DW;
            const int y = by0 + py, x = bx0 + px;
            const bool ok = s < 16 * NDY && y < g.H && x < g.W;
            const unsigned off = ok ? (unsigned)(((((bimg * (N >> 2) + nb * 16 + quad) * g.H + y) * g.W + x)) * 16) : WOOB;
            st[k] = wbufload(ry, off, 0);
        }
    };
    auto lstore1 = [&](int buf, int i, int k) {
        if (i < 12) {
            const int s = tid + 256 * i, quad = s / NPX, pix = s - quad * NPX;
            if (s < 16 * NPX) *reinterpret_cast<float4*>(&smem[buf * BUF + quad * XPL + pix * 4]) = st[k];
        } else {
            const int s = tid + 256 * (i - 12), quad = s / NDY, pix = s - quad * NDY;
            if (s < 16 * NDY) *reinterpret_cast<float4*>(&smem[buf * BUF + XPF + quad * YPL + pix * 4]) = st[k];
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
    const int xbase = ((2 * lh) * PW) * 4 + (cw * 8 + (li >> 2)) * XPL + (li & 3);          // + buf * BUF + ((4 r + i) * PW + 2 tx + j) * 4
    const int ybase = XPF + ((2 * lh) * DW) * 4 + (nw * 8 + (li >> 2)) * YPL + (li & 3);    // + buf * BUF + ((4 r + a) * DW + 2 tx + b) * 4
    // micro-operation k of preparing step sn (from LDS buffer `buf`) into operand set ob
    auto prep = [&](int buf, int sn, int ob, int k) {
        const int r = sn / TBW, tx = sn - r * TBW;
        const bool fresh = tx == 0;
        const int nread = fresh ? 16 : 8;
        if (k < nread) {
            const int idx = fresh ? k : 8 + k, j = idx >> 2, i = idx & 3;
            if (WG_ABL & 2) return;
            dv[i][j] = smem[buf * BUF + xbase + ((4 * r + i) * PW + 2 * tx + j) * 4];
            return;
        }
        k -= nread;
        if (k < 4) {
            const int aa = k >> 1, bb = k & 1;
            if (WG_ABL & 2) return;
            ev[aa][bb] = smem[buf * BUF + ybase + ((4 * r + aa) * DW + 2 * tx + bb) * 4];
            return;
        }
        k -= 4;
        if (WG_ABL & 1) return;
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
                if (more && m < 10 && !(WG_ABL & 4)) {   // the next block's data: two batches of ten slots, loaded early, written a few steps later
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

constexpr int WINO_WG_LDS_BYTES = 2 * 16 * (WG_XPL + WG_YPL) * 4;   // 158 720

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


int launch_wino_wgrad_4x8(hipStream_t st, const WinoWgArgs& a);
int launch_wino_wgrad_4x7(hipStream_t st, const WinoWgArgs& a);
int launch_wino_wgrad_2x14(hipStream_t st, const WinoWgArgs& a);

}  // namespace vc
