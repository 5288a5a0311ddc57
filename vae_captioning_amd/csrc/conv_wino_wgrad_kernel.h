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
    static_assert(PW * PH <= 180 && DW * DH <= 128 && NS <= 16 && NS >= 13, "block shape (the staging schedule runs to step 11, before the barrier of step NS - 1)");
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

    // staging slots: a thread owns ONE pixel of the x halo patch (pixel tid < NPX <= 180) and ONE pixel of the dy block (pixel tid & 127
    // < NDY <= 128, channel-quad parity tid >> 7): per block TWO voffsets (the pixel's 16 bytes in channel-quad plane 0 of the tile, or
    // WOOB) and the quad as a SCALAR plane offset -- slot i < 16: x quad i; slot 16 + i, i < 8: dy quad 2 i + (tid >> 7).  (Slots that were
    // float4 index tid + 256 i of [quad][pixel] cost two divisions and a bounds check per slot and block: + 0.4 ms per step, round 4.)
    float4 st[8];
    const int xpy = tid / PW, xpx = tid - xpy * PW;                 // (compile-time divisors)
    const int ypix = tid & 127, ypy = ypix / DW, ypx = ypix - ypy * DW;
    const bool xlive = tid < NPX, ylive = ypix < NDY;
    const unsigned plane_b = (unsigned)g.H * (unsigned)g.W * 16u;   // bytes of one channel-quad plane of an image
    const int xq0 = cb * 16, yq0 = nb * 16 + (tid >> 7);
    const int xst = tid * 4, yst = XPF + (tid >> 7) * YPL + ypix * 4;   // LDS float index of the thread's pixel in plane 0 (+ buf * BUF + quad * pitch)
    unsigned xvoff = WOOB, yvoff = WOOB;   // of the block to LOAD
    auto set_block = [&](int blk) {
        const unsigned b = (unsigned)blk / (unsigned)g.blocks_img, rem = (unsigned)blk - b * (unsigned)g.blocks_img;
        const unsigned by = rem / (unsigned)g.bx_n, bx = rem - by * (unsigned)g.bx_n;
        const int y = (int)by * DH - 1 + xpy, x = (int)bx * DW - 1 + xpx;
        xvoff = (xlive && (unsigned)y < (unsigned)g.H && (unsigned)x < (unsigned)g.W)
                    ? (unsigned)((((int)b * (C >> 2) + xq0) * g.H + y) * g.W + x) * 16u : WOOB;
        const int yy = (int)by * DH + ypy, yx = (int)bx * DW + ypx;
        yvoff = (ylive && yy < g.H && yx < g.W) ? (unsigned)((((int)b * (N >> 2) + yq0) * g.H + yy) * g.W + yx) * 16u : WOOB;
    };
    auto gload1 = [&](int i, int k) {   // slot i into st[k]
        if (i < 16) st[k] = wbufload(rx, xvoff, (unsigned)i * plane_b);
        else st[k] = wbufload(ry, yvoff, (unsigned)(2 * (i - 16)) * plane_b);
    };
    auto lstore1 = [&](int buf, int i, int k) {
        if (i < 16) {
            if (xlive) *reinterpret_cast<float4*>(&smem[buf * BUF + xst + i * XPL]) = st[k];
        } else {
            if (ylive) *reinterpret_cast<float4*>(&smem[buf * BUF + yst + 2 * (i - 16) * YPL]) = st[k];
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
        for (int h = 0; h < 3; ++h) {
#pragma unroll
            for (int k = 0; k < 8; ++k) gload1(8 * h + k, k);
#pragma unroll
            for (int k = 0; k < 8; ++k) lstore1(0, 8 * h + k, k);
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
                if (more && m < 8 && !(WG_ABL & 4)) {   // the next block's data: three batches of eight slots, each loaded three steps before it is written
                    if (s == 0) gload1(m, m);
                    if (s == 3) lstore1(buf ^ 1, m, m);
                    if (s == 4) gload1(8 + m, m);
                    if (s == 7) lstore1(buf ^ 1, 8 + m, m);
                    if (s == 8) gload1(16 + m, m);
                    if (s == 11) lstore1(buf ^ 1, 16 + m, m);
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
