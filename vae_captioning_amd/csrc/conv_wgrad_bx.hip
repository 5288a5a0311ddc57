// Weight gradient of the 3x3 convolutions on the bf16 matrix pipe of gfx950 ("split-bf16": a = hi + lo, three products, f32
// accumulators -- gemm_bf16x3_core.h), utils/image_embeddings.py:36-212 (backward of tf.nn.conv2d w.r.t. the filter).
//
//   dW[ky][kx][c][n] = sum_{b, y, x} X[b][y + ky - 1][x + kx - 1][c] dY[b][y][x][n]
//                    = sum_{b, y', x} X[b][y'][x + kx - 1][c] dY[b][y' - ky + 1][x][n]
//
// a DIRECT product (no Winograd transform: on this pipe a MAC costs 3/16 of an f32 MFMA MAC, and the transform arithmetic of
// F(3x3,2x2) would be the bound) whose contraction index is the PIXEL.  The kernel is built round what a lane can keep in
// registers:
//   * v_mfma_f32_32x32x16_bf16 takes, per lane, eight consecutive k of ONE row of A / column of B.  With k = sixteen pixels of an
//     image row, a lane owns one channel (A: x channel c0 + lane % 32, B: dy channel n0 + lane % 32) and the eight pixels
//     8 (lane / 32) .. + 7 -- it reads them from an LDS image of the C4 layout as it arrives from HBM ([channel quad][pixel][4]:
//     staging is a plain 16-byte copy) and splits them IN REGISTERS (v_cvt_pk_bf16_f32, widen, subtract, v_cvt_pk_bf16_f32).
//   * second form of the sum: the chunk (image row y', sixteen columns) multiplies ONE x row, shifted three ways (kx), with
//     THREE dy rows (ky).  The x row is read with a one-pixel halo (ten values per lane) and split once; the kx = 1 operand is
//     the same packed pairs shifted by sixteen bits (v_alignbit_b32), kx = 2 the pairs one register further.  The dy rows roll:
//     walking down a block, a chunk needs one new dy row and keeps two.
//   * a wave owns 32 input channels x 64 output channels x the nine taps = eighteen 32 x 32 accumulators (288 registers: one
//     wave per SIMD, 512 registers each), so that a split x row feeds 54 MFMAs and a split dy row 27.  A workgroup = four waves
//     = 64 x 64 channels x two halves of a block of eight image rows x sixteen columns.
// Per chunk and wave: 54 MFMAs (1728 cycles) against ~26 LDS reads and ~100 VALU operations, placed by hand between the
// MFMAs (`ops`), one chunk ahead of their use.  Blocks are staged through two LDS images (one barrier per block: 216 MFMAs per
// wave), global loads issued one to two chunks before their LDS writes.
// Image rows are stacked with ONE zero row between images (global row g = b (H + 1) + y): a block of eight rows may span two
// images, and the 14- and 28-row layers lose 1 / (H + 1) of their rows instead of an eighth.  Columns come in sixteens (a 14-,
// 28- or 56-wide layer carries 12.5 % dead k).
// K is split over the blocks (workgroups = channel tiles x splits >= 256); the two row halves of a workgroup are added in the LDS, its raw
// sums go to the workspace, wgrad_bx_reduce_kernel adds the nsplit partials in fixed order (deterministic) and forms db from the raw f32 dy sums.
#include <stdlib.h>
#include <type_traits>
#include "conv_wino.h"
#include "gemm_bf16x3_core.h"

// `make wbabl`: WB_ABL = bit mask that REMOVES parts (results wrong; timing only): 1 split arithmetic, 2 LDS operand reads,
// 4 staging (global loads + LDS writes), 8 MFMAs, 16 everything between the MFMAs (operands frozen after the first chunk)
#ifndef WB_ABL
#define WB_ABL 0
#endif

namespace vc {

constexpr int WB_XPL = 580, WB_YPL = 644;   // LDS plane pitches in floats: >= 4 * (8 * 18), 4 * (10 * 16), both = 4 mod 32
constexpr int WB_XPF = 16 * WB_XPL, WB_BUF = WB_XPF + 16 * WB_YPL;
constexpr int WB_LDS_BYTES = 2 * WB_BUF * 4;   // 156 672

struct WgBxArgs {
    const float* x;      // [B][C/4][H][W][4]
    const float* dy;     // [B][N/4][H][W][4]
    float* ws;           // [nsplit][9][C][N] raw sums, then [2 nsplit][N] dy sums
    int B, H, W, C, N;
    int bx_n, nblocks, grows;   // blocks per block row; blocks; stacked rows B (H + 1) - 1
    unsigned m_bx_n, one_bx_n, m_h1;   // wino_magic of bx_n (one_bx_n = ~0u when bx_n == 1) and of H + 1
    int ncb, nnb, nsplit, cps, xcd;
};

#define WSB() __builtin_amdgcn_sched_barrier(0)

__global__ __launch_bounds__(256, 1) void wgrad_bx_kernel(WgBxArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int XPL = WB_XPL, YPL = WB_YPL, XPF = WB_XPF, BUF = WB_BUF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int cw = wave & 1, kh = wave >> 1;   // input-channel half of the tile; rows 4 kh .. 4 kh + 3 of a block
    const int wid = a.xcd ? xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int cn = wid % (a.ncb * a.nnb), split = wid / (a.ncb * a.nnb);
    const int cb = cn / a.nnb, nb = cn - cb * a.nnb;
    const int C = a.C, N = a.N, H = a.H, W = a.W;
    const __amdgpu_buffer_rsrc_t rx_ = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)((long)a.B * H * W * C * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t ry_ = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, (int)((long)a.B * H * W * N * 4), 0x00020000);
    const int blk0 = split * a.cps;
    int nch = a.nblocks - blk0;
    if (nch > a.cps) nch = a.cps;
    if (nch < 0) nch = 0;

    // ---- staging: nineteen 16-byte slots per thread and block, four address families ----------------------------------------------
    //   s 0..7   x patch pixel tid & 127 (of 8 x 18 = 144), channel quads 2 s + (tid >> 7)
    //   s 8      x patch pixel 128 + (tid & 15), quad tid >> 4
    //   s 9..16  dy patch pixel tid & 127 (of 10 x 16 = 160), quads 2 (s - 9) + (tid >> 7)
    //   s 17, 18 dy patch pixel 128 + (tid & 31), quads (tid >> 5) + 8 (s - 17)
    // x patch pixel (pr, pc) = image row 8 by + pr, column 16 bx - 1 + pc; dy patch pixel = row 8 by - 1 + pr, column 16 bx + pc
    const unsigned plane_b = (unsigned)H * (unsigned)W * 16u;
    const int p0 = tid & 127, p1 = 128 + (tid & 15), p3 = 128 + (tid & 31);
    const int f_pr[4] = {p0 / 18, p1 / 18, p0 >> 4, p3 >> 4};
    const int f_pc[4] = {p0 % 18 - 1, p1 % 18 - 1, p0 & 15, p3 & 15};
    const int f_q[4] = {cb * 16 + (tid >> 7), cb * 16 + (tid >> 4), nb * 16 + (tid >> 7), nb * 16 + (tid >> 5)};
    const int f_lds[4] = {(tid >> 7) * XPL + p0 * 4, (tid >> 4) * XPL + p1 * 4, XPF + (tid >> 7) * YPL + p0 * 4, XPF + (tid >> 5) * YPL + p3 * 4};
    unsigned voff[4] = {WOOB, WOOB, WOOB, WOOB};
    // (branch-free on purpose: a division "m ? __umulhi(n, m) : n" or a guarded address is compiled to a branch, and a branch inside the
    // MFMA stream is a scheduling barrier with an s_cbranch in front of it)
    auto bdiv = [&](unsigned n, unsigned m, unsigned one) -> unsigned { return (__umulhi(n, m) & ~one) | (n & one); };   // one = ~0u: divisor 1
    auto set_family = [&](int f, int blk, bool live) {
        const unsigned by = bdiv((unsigned)blk, a.m_bx_n, a.one_bx_n), bx = (unsigned)blk - by * (unsigned)a.bx_n;
        const int g = (int)by * 8 + f_pr[f] - (f >= 2 ? 1 : 0), col = (int)bx * 16 + f_pc[f];
        const unsigned b = __umulhi((unsigned)max(g, 0), a.m_h1);   // H + 1 >= 2
        const int y = g - (int)b * (H + 1);
        const unsigned ok = 0u - (unsigned)((int)live & (int)((unsigned)g < (unsigned)a.grows) & (int)(y < H) & (int)((unsigned)col < (unsigned)W));
        const int ch4 = f < 2 ? (C >> 2) : (N >> 2);
        const unsigned addr = (unsigned)((((int)b * ch4 + f_q[f]) * H + y) * W + col) * 16u;
        voff[f] = (addr & ok) | (WOOB & ~ok);
    };
    auto gload = [&](int s) -> float4 {
        if (s < 8) return wbufload(rx_, voff[0], (unsigned)(2 * s) * plane_b);
        if (s == 8) return wbufload(rx_, voff[1], 0u);
        if (s < 17) return wbufload(ry_, voff[2], (unsigned)(2 * (s - 9)) * plane_b);
        return wbufload(ry_, voff[3], (unsigned)(8 * (s - 17)) * plane_b);
    };
    auto lstore = [&](int dst, int s, const float4& v) {   // dst = float offset of the image being filled
        float* p = s < 8 ? smem + dst + f_lds[0] + 2 * s * XPL : s == 8 ? smem + dst + f_lds[1] : s < 17 ? smem + dst + f_lds[2] + 2 * (s - 9) * YPL : smem + dst + f_lds[3] + 8 * (s - 17) * YPL;
        *reinterpret_cast<float4*>(p) = v;
    };
    float4 st[7];

    // ---- operands -------------------------------------------------------------------------------------------------------------
    // lane (li, lh): x channel cw * 32 + li = quad cw * 8 + li / 4, component li % 4; pixels 8 lh + i of patch row 4 kh + R
    const int xb = (cw * 8 + (li >> 2)) * XPL + (li & 3) + ((4 * kh) * 18 + 8 * lh) * 4;
    const int yb = XPF + (li >> 2) * YPL + (li & 3) + ((4 * kh) * 16 + 8 * lh) * 4;   // + ng * 8 * YPL + (lr * 16 + k) * 4
    unsigned ph[5], pl[5];
    u32x4 xh[3], xl[3], xnh[3], xnl[3];
    u32x4 dyh[3][2], dyl[3][2];
    float dbs[2] = {0.f, 0.f};
    // A PAIR of adjacent pixels is the unit of operand preparation: one ds_read2_b32, then the split (p, q) -> packed (hi, lo) in four
    // STAGES, each behind a different MFMA (a dependent chain of six VALU instructions behind one MFMA outlasts it):
    //   0: hi = bf16(p, q)   1: widen hi   2: r = (p, q) - float(hi)   3: lo = bf16(r)  (+ the pair's share of db)
    // -- scalar subtractions on purpose: a v_pk_add_f32 beside MFMAs costs ~13 cycles beyond its issue slot (MI355X_MICROARCH.md,
    // "price of one filler"); the file is compiled with -fno-slp-vectorize so that hipcc does not re-pack them.
    // rawx / rawy / tmp* are indexed by compile-time constants: registers that live from a pair's read to its last stage.
    float rawx[4][5][2], tmpx[4][5][2], rawy[7][8][2], tmpy[7][8][2];
    auto st_hi = [&](float p, float q) -> unsigned {
        if (WB_ABL & 1) return __float_as_uint(p);
        return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{p, q}, bf16x2));
    };
    auto st_lo = [&](float p, float q) -> unsigned {
        if (WB_ABL & 1) return __float_as_uint(q);
        return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{p, q}, bf16x2));
    };
    // x pair m of patch row `row` (task id t: distinct temporaries per row in flight); sg = -1: read, 0..3: stages
    auto xpair = [&](int t, int img, int row, int m, int sg) {
        if (sg == -1) {
            if (WB_ABL & 2) { rawx[t][m][0] = (float)(m + row); rawx[t][m][1] = (float)(m - row); return; }
            rawx[t][m][0] = smem[img + xb + (row * 18 + 2 * m) * 4];
            rawx[t][m][1] = smem[img + xb + (row * 18 + 2 * m + 1) * 4];
        } else if (sg == 0) ph[m] = st_hi(rawx[t][m][0], rawx[t][m][1]);
        else if (sg == 1) {
            if (!(WB_ABL & 1)) { tmpx[t][m][0] = __uint_as_float(ph[m] << 16); tmpx[t][m][1] = __uint_as_float(ph[m] & 0xffff0000u); }
        } else if (sg == 2) {
            if (!(WB_ABL & 1)) { tmpx[t][m][0] = rawx[t][m][0] - tmpx[t][m][0]; tmpx[t][m][1] = rawx[t][m][1] - tmpx[t][m][1]; }
            else { tmpx[t][m][0] = rawx[t][m][1]; tmpx[t][m][1] = rawx[t][m][1]; }
        } else pl[m] = st_lo(tmpx[t][m][0], tmpx[t][m][1]);
    };
    auto mkx = [&](int k) {   // the three kx operands of the next chunk from the five packed pairs, in six parts
        if (k == 0) { xnh[1][0] = __builtin_amdgcn_alignbit(ph[1], ph[0], 16); xnh[1][1] = __builtin_amdgcn_alignbit(ph[2], ph[1], 16); }
        else if (k == 1) { xnh[1][2] = __builtin_amdgcn_alignbit(ph[3], ph[2], 16); xnh[1][3] = __builtin_amdgcn_alignbit(ph[4], ph[3], 16); }
        else if (k == 2) { xnl[1][0] = __builtin_amdgcn_alignbit(pl[1], pl[0], 16); xnl[1][1] = __builtin_amdgcn_alignbit(pl[2], pl[1], 16); }
        else if (k == 3) { xnl[1][2] = __builtin_amdgcn_alignbit(pl[3], pl[2], 16); xnl[1][3] = __builtin_amdgcn_alignbit(pl[4], pl[3], 16); }
        else if (k == 4) { xnh[0] = u32x4{ph[0], ph[1], ph[2], ph[3]}; xnh[2] = u32x4{ph[1], ph[2], ph[3], ph[4]}; }
        else { xnl[0] = u32x4{pl[0], pl[1], pl[2], pl[3]}; xnl[2] = u32x4{pl[1], pl[2], pl[3], pl[4]}; }
    };
    // dy pair p (channel group p / 4, pair p % 4) of patch row lr (relative to 4 kh) into register slot `slot`; task id t
    auto ypair = [&](int t, int img, int lr, int p, int sg, int slot, bool centre) {
        const int ng = p >> 2, m = p & 3;
        if (sg == -1) {
            if (WB_ABL & 2) { rawy[t][p][0] = (float)(p + lr); rawy[t][p][1] = (float)(p - lr); return; }
            rawy[t][p][0] = smem[img + yb + ng * 8 * YPL + (lr * 16 + 2 * m) * 4];
            rawy[t][p][1] = smem[img + yb + ng * 8 * YPL + (lr * 16 + 2 * m + 1) * 4];
        } else if (sg == 0) dyh[slot][ng][m] = st_hi(rawy[t][p][0], rawy[t][p][1]);
        else if (sg == 1) {
            if (!(WB_ABL & 1)) { tmpy[t][p][0] = __uint_as_float(dyh[slot][ng][m] << 16); tmpy[t][p][1] = __uint_as_float(dyh[slot][ng][m] & 0xffff0000u); }
        } else if (sg == 2) {
            if (!(WB_ABL & 1)) { tmpy[t][p][0] = rawy[t][p][0] - tmpy[t][p][0]; tmpy[t][p][1] = rawy[t][p][1] - tmpy[t][p][1]; }
            else { tmpy[t][p][0] = rawy[t][p][1]; tmpy[t][p][1] = rawy[t][p][1]; }
        } else {
            dyl[slot][ng][m] = st_lo(tmpy[t][p][0], tmpy[t][p][1]);
            if (centre) dbs[ng] += rawy[t][p][0] + rawy[t][p][1];   // db: every real dy row is the centre row of exactly one wave
        }
    };
    // a row's pairs p0 .. p0 + np - 1 as a pipeline: pair p starts (stage 0) at interval S + II (p - p0), its read three intervals earlier
    auto xrow = [&](int N, int S, int II, int t, int img, int row) {
#pragma unroll
        for (int m = 0; m < 5; ++m) {
            const int d = N - (S + II * m);
            if (d == -3) xpair(t, img, row, m, -1);
            else if (d >= 0 && d < 4) xpair(t, img, row, m, d);
        }
    };
    auto yrow = [&](int N, int S, int II, int p0, int np, int t, int img, int lr, int slot, bool centre) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int d = N - (S + II * (p - p0));
            if (p < p0 || p >= p0 + np) continue;
            if (d == -3) ypair(t, img, lr, p, -1, slot, centre);
            else if (d >= 0 && d < 4) ypair(t, img, lr, p, d, slot, centre);
        }
    };

    f32x16 acc[2][9];
#pragma unroll
    for (int ng = 0; ng < 2; ++ng)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ng][t][r] = 0.f;

    int cur = 0, nxt = BUF;   // float offsets of the two LDS images

    // ---- prologue: block 0 into image 0, batch A of block 1 into registers, the operands of chunk 0 ----------------------------
#pragma unroll
    for (int f = 0; f < 4; ++f) set_family(f, blk0, nch > 0);
#pragma unroll
    for (int h = 0; h < 3; ++h) {
        const int s0 = h == 0 ? 0 : h == 1 ? 7 : 13, ns = h == 0 ? 7 : 6;
#pragma unroll
        for (int i = 0; i < 7; ++i)
            if (i < ns) st[i] = gload(s0 + i);
#pragma unroll
        for (int i = 0; i < 7; ++i)
            if (i < ns) lstore(cur, s0 + i, st[i]);
    }
#pragma unroll
    for (int f = 0; f < 4; ++f) set_family(f, blk0 + 1, nch > 1);
#pragma unroll
    for (int i = 0; i < 7; ++i) st[i] = gload(i);
    __syncthreads();
#pragma unroll
    for (int sg = -1; sg < 4; ++sg)
#pragma unroll
        for (int m = 0; m < 5; ++m) xpair(0, cur, 0, m, sg);
#pragma unroll
    for (int k = 0; k < 6; ++k) mkx(k);
#pragma unroll
    for (int t = 0; t < 3; ++t) { xh[t] = xnh[t]; xl[t] = xnl[t]; }
#pragma unroll
    for (int sg = -1; sg < 4; ++sg) {
#pragma unroll
        for (int i = 0; i < 8; ++i) ypair(0, cur, 0, i, sg, 0, false);
#pragma unroll
        for (int i = 0; i < 4; ++i) ypair(1, cur, 1, i, sg, 1, true);   // pairs 4..7 of row 1: intervals 3..15 of the block (below)
        if (WB_ABL & 16) {
#pragma unroll
            for (int i = 4; i < 8; ++i) ypair(1, cur, 1, i, sg, 1, true);
#pragma unroll
            for (int i = 0; i < 8; ++i) ypair(2, cur, 2, i, sg, 2, true);
        }
    }
    WSB();

    // ---- the operations placed behind MFMA N = 54 J + n (0..215) of a block ------------------------------------------------------
    // chunk J multiplies x row J (xh / xl) with dy rows lr = J (ky 2), J + 1 (ky 1), J + 2 (ky 0) in register slots lr % 3, ky 2 first,
    // so a dy row may be (re)written into its slot from interval 54 (lr - 3) + 18 on and must be complete by 54 (lr - 2) + 36:
    //   row   slot   written during        from
    //   1'    1      [3, 16) pairs 4..7    this image (the row's pairs 0..3 were split at the end of the previous block)
    //   2     2      [4, 36)               this image
    //   3     0      [37, 83)              this image
    //   4     1      [84, 130)             this image
    //   5     2      [132, 178)            this image
    //   0'    0      [181, 213)            the other image (complete behind the barrier at 162): the NEXT block's row 0
    //   1'    1      [199, 215) pairs 0..3 the other image
    // x row J + 1 (five pairs + the kx operands) is prepared inside chunk J, the next block's row 0 inside chunk 3.  At most ~5
    // instructions land behind one MFMA (what a 32-cycle MFMA hides for a single wave, MI355X_MICROARCH.md).
    // Staging of the next block into the other image (free since the previous barrier): three batches of 16-byte slots, each loaded
    // ~60 MFMAs (~2000 cycles) before it is written; the addresses of the block after next are formed behind batch C's loads.
    auto ops = [&](int J, int n, int blk_next2, bool live_next2) {
        const int N = 54 * J + n;
        if (WB_ABL & 16) return;   // MFMAs only, on the operands of the first chunk (real data: the matrix pipe's power draw, and clock, depend on it)
        yrow(N, 3, 3, 4, 4, 1, cur, 1, 1, true);
        yrow(N, 4, 4, 0, 8, 2, cur, 2, 2, true);
        yrow(N, 37, 6, 0, 8, 3, cur, 3, 0, true);
        yrow(N, 84, 6, 0, 8, 4, cur, 4, 1, true);
        yrow(N, 132, 6, 0, 8, 5, cur, 5, 2, false);
        yrow(N, 181, 4, 0, 8, 6, nxt, 0, 0, false);
        yrow(N, 199, 4, 0, 4, 0, nxt, 1, 1, true);
        xrow(N, 17, 4, 1, cur, 1);
        if (N >= 38 && N < 44) mkx(N - 38);
        xrow(N, 57, 4, 2, cur, 2);
        if (N >= 78 && N < 84) mkx(N - 78);
        xrow(N, 111, 4, 3, cur, 3);
        if (N >= 132 && N < 138) mkx(N - 132);
        xrow(N, 165, 4, 0, nxt, 0);
        if (N >= 186 && N < 192) mkx(N - 186);
        if (!(WB_ABL & 4)) {
            if (N >= 30 && N < 37) lstore(nxt, N - 30, st[N - 30]);              // batch A (slots 0..6), loaded at 192.. of the previous block
            if (N >= 40 && N < 46) st[N - 40] = gload(7 + N - 40);              // batch B (7..12)
            if (N >= 100 && N < 106) lstore(nxt, 7 + N - 100, st[N - 100]);
            if (N >= 106 && N < 112) st[N - 106] = gload(13 + N - 106);         // batch C (13..18)
            if (N >= 140 && N < 148 && !(N & 1)) set_family((N - 140) >> 1, blk_next2, live_next2);
            if (N >= 155 && N < 161) lstore(nxt, 13 + N - 155, st[N - 155]);
            if (N >= 192 && N < 199) st[N - 192] = gload(N - 192);              // batch A of the block after next
        }
    };
    // eighteen accumulators are 288 registers against 256 AGPRs: left to itself the register allocator moves whole accumulators between
    // AGPR tuples and VGPRs round every MFMA (1764 v_accvgpr moves and 486 spills in the first build) -- the register class is pinned
    // per accumulator instead: sixteen in AGPRs, the last two (ng 1, taps 7 and 8) in VGPRs.  An accumulator is only ever the C AND D
    // of its MFMAs (no nops needed between them) and is read once, behind the s_nop of the epilogue.
    auto mfma = [&](f32x16& c, const u32x4& av, const u32x4& bv, bool in_agpr) {
        if (in_agpr) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(av), "v"(bv));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(av), "v"(bv));
    };
    auto chunk = [&](int J, int blk_next2, bool live_next2) {
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const int ky = 2 - g, slot = (J + g) % 3;
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int ng = 0; ng < 2; ++ng)
#pragma unroll
                    for (int tx = 0; tx < 3; ++tx) {
                        const int n = g * 18 + t * 6 + ng * 3 + tx;
                        if (!(WB_ABL & 8)) mfma(acc[ng][ky * 3 + tx], t == 1 ? xl[tx] : xh[tx], t == 2 ? dyl[slot][ng] : dyh[slot][ng], ng * 9 + ky * 3 + tx < 16);
                        WSB();
                        ops(J, n, blk_next2, live_next2);
                        WSB();
                    }
        }
        if (!(WB_ABL & 16)) {
#pragma unroll
            for (int t = 0; t < 3; ++t) { xh[t] = xnh[t]; xl[t] = xnl[t]; }
        }
    };

    for (int ci = 0; ci < nch; ++ci) {
        chunk(0, 0, false);
        chunk(1, 0, false);
        chunk(2, blk0 + ci + 2, ci + 2 < nch);
        __syncthreads();   // the other image is complete; this one is behind every wave
        chunk(3, 0, false);
        const int t = cur; cur = nxt; nxt = t;
    }
#undef WSB
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");

    // ---- the two row halves meet in the LDS (waves kh = 1 write their accumulators, waves kh = 0 add them: half the workspace traffic
    // and half the reduce kernel's reads), then the raw sums of this split go out:
    //   acc[ng][tap][r] = S[tap][c0 + 8 (r >> 2) + 4 lh + (r & 3)][n0 + 32 ng + li]
    __syncthreads();   // every wave is behind its last LDS read
    float4* xch = reinterpret_cast<float4*>(smem) + (size_t)cw * (2 * 9 * 4 * 64) + lane;   // [cw][ng][tap][register quad][lane] float4: 147 456 bytes
    if (kh == 1) {
#pragma unroll
        for (int ng = 0; ng < 2; ++ng)
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    xch[((ng * 9 + t) * 4 + q) * 64] = make_float4(acc[ng][t][4 * q], acc[ng][t][4 * q + 1], acc[ng][t][4 * q + 2], acc[ng][t][4 * q + 3]);
    }
    __syncthreads();
    const int c0 = cb * 64 + cw * 32, n0 = nb * 64;
    if (kh == 0) {
        float* o = a.ws + (long)split * 9 * (long)C * N + n0 + li;
#pragma unroll
        for (int ng = 0; ng < 2; ++ng)
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = xch[((ng * 9 + t) * 4 + q) * 64];
                    float* oq = o + ((long)t * C + c0 + 8 * q + 4 * lh) * N + 32 * ng;
                    oq[0] = acc[ng][t][4 * q] + v.x;
                    oq[N] = acc[ng][t][4 * q + 1] + v.y;
                    oq[2 * (long)N] = acc[ng][t][4 * q + 2] + v.z;
                    oq[3 * (long)N] = acc[ng][t][4 * q + 3] + v.w;
                }
    }
    if (cb == 0 && cw == 0) {   // dy sums: one row of N per (split, row half)
        const long z = (long)split * 2 + kh;
#pragma unroll
        for (int ng = 0; ng < 2; ++ng) {
            float v = dbs[ng];
            v += __shfl_xor(v, 32, 64);
            if (lh == 0) a.ws[(long)a.nsplit * 9 * C * N + z * N + n0 + 32 * ng + li] = v;
        }
    }
}

// dw (+)= sum of the nsplit partials (fixed order); db = sum of the 2 nsplit dy-sum rows.  One float4 of [9][C][N] per thread.
__global__ __launch_bounds__(256) void wgrad_bx_reduce_kernel(const float* __restrict__ ws, int nz, long n4, int N, float* __restrict__ dw,
                                                              float* __restrict__ db, int accumulate) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) {
        const float4* src = reinterpret_cast<const float4*>(ws) + i;
        float4 s = src[0];
        for (int z = 1; z < nz; ++z) {
            const float4 v = src[(long)z * n4];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        float4* o = reinterpret_cast<float4*>(dw) + i;
        if (accumulate) { const float4 p = *o; s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w; }
        *o = s;
    } else if (db) {
        const long j = i - n4;
        if (j < N) {
            const float* bw = ws + (long)nz * n4 * 4;
            float v = 0.f;
            for (int z = 0; z < 2 * nz; ++z) v += bw[(long)z * N + j];
            db[j] = accumulate ? db[j] + v : v;
        }
    }
}

struct WgBxPlan {
    bool ok;
    int bx_n, by_n, nblocks, grows, ncb, nnb, nsplit, cps;
};

static WgBxPlan plan_wgrad_bx(int B, int H, int W, int C, int N) {
    WgBxPlan p;
    p.ok = false;
    p.bx_n = p.by_n = p.nblocks = p.grows = p.ncb = p.nnb = p.nsplit = p.cps = 0;
    if (B <= 0 || H < 1 || W < 1 || C <= 0 || N <= 0 || C % 64 || N % 64) return p;
    if ((long)B * H * W * (long)(C > N ? C : N) * 4 > 0x7fffffffL) return p;
    if ((long)B * (H + 1) > 0x0fffffffL) return p;
    p.grows = B * (H + 1) - 1;
    p.bx_n = cdiv(W, 16);
    p.by_n = cdiv(p.grows, 8);
    p.nblocks = p.bx_n * p.by_n;
    p.ncb = C / 64; p.nnb = N / 64;
    const int target = 256;   // one workgroup per CU
    int ns = cdiv(target, p.ncb * p.nnb);
    if (ns > p.nblocks) ns = p.nblocks;
    p.cps = cdiv(p.nblocks, ns);
    p.nsplit = cdiv(p.nblocks, p.cps);
    p.ok = true;
    return p;
}

static size_t wgrad_bx_ws(const WgBxPlan& p, int C, int N) {
    return p.ok ? ((size_t)p.nsplit * 9 * C * N + (size_t)p.nsplit * 2 * N) * sizeof(float) : 0;
}

static size_t wgrad_bx_ws_call(int B, int H, int W, int Cin, int Cout) {
    const int per = wino_images_per_launch(B, H, W, Cin, Cout);
    if (per <= 0) return 0;
    size_t need = wgrad_bx_ws(plan_wgrad_bx(per, H, W, Cin, Cout), Cin, Cout);
    if (B % per) {
        const size_t r = wgrad_bx_ws(plan_wgrad_bx(B % per, H, W, Cin, Cout), Cin, Cout);
        if (r > need) need = r;
    }
    return need;
}

}  // namespace vc

extern "C" int vc_conv3x3_bx_wgrad_supported(int B, int H, int W, int Cin, int Cout) {
    const int nb = vc::wino_images_per_launch(B, H, W, Cin, Cout);
    return nb > 0 && vc::plan_wgrad_bx(nb, H, W, Cin, Cout).ok ? 1 : 0;
}

extern "C" size_t vc_conv3x3_bx_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
    return vc::wgrad_bx_ws_call(B, H, W, Cin, Cout);
}

extern "C" int vc_conv3x3_bx_wgrad_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* x, const float* dy, float* dw,
                                       float* db, int accumulate, float* ws, size_t ws_bytes) {
    using namespace vc;
    const int per = wino_images_per_launch(B, H, W, Cin, Cout);
    VC_CHECK_ARG(per > 0 && plan_wgrad_bx(per, H, W, Cin, Cout).ok, "unsupported shape (vc_conv3x3_bx_wgrad_supported)");
    VC_CHECK_ARG(x && dy && dw && ws, "null pointer");
    VC_CHECK_ARG(waligned16(x) && waligned16(dy) && waligned16(ws) && waligned16(dw), "pointers must be 16-byte aligned");
    VC_CHECK_ARG(B % per == 0 || plan_wgrad_bx(B % per, H, W, Cin, Cout).ok, "unsupported shape (vc_conv3x3_bx_wgrad_supported)");
    if (ws_bytes < wgrad_bx_ws_call(B, H, W, Cin, Cout))
        return fail(VC_EWORKSPACE, "%s: workspace too small (%ld < %ld bytes)", __func__, (long)ws_bytes, (long)wgrad_bx_ws_call(B, H, W, Cin, Cout));
    static int once = [] {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_bx_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, WB_LDS_BYTES);
        return e == hipSuccess ? 0 : fail((int)e, "%s: hipFuncSetAttribute failed", "bx wgrad kernel");
    }();
    if (once) return once;
    for (int b0 = 0; b0 < B; b0 += per) {   // image ranges of < 2 GiB; the later ranges accumulate into dw / db
        const int nbi = B - b0 < per ? B - b0 : per;
        const WgBxPlan p = plan_wgrad_bx(nbi, H, W, Cin, Cout);
        WgBxArgs a;
        a.x = x + (size_t)b0 * H * W * Cin; a.dy = dy + (size_t)b0 * H * W * Cout; a.ws = ws;
        a.B = nbi; a.H = H; a.W = W; a.C = Cin; a.N = Cout;
        a.bx_n = p.bx_n; a.nblocks = p.nblocks; a.grows = p.grows;
        a.m_bx_n = wino_magic(p.bx_n); a.one_bx_n = p.bx_n == 1 ? 0xffffffffu : 0u; a.m_h1 = wino_magic(H + 1);
        a.ncb = p.ncb; a.nnb = p.nnb; a.nsplit = p.nsplit; a.cps = p.cps;
        static const int xcd_on = getenv("VC_WGRAD_XCD") ? atoi(getenv("VC_WGRAD_XCD")) : 1;
        a.xcd = xcd_on;
        hipLaunchKernelGGL(wgrad_bx_kernel, dim3(p.ncb * p.nnb * p.nsplit), dim3(256), WB_LDS_BYTES, (hipStream_t)stream, a);
        int rc = launch_status(__func__);
        if (rc) return rc;
        const long n4 = 9L * Cin * Cout / 4;
        const long nthreads = n4 + (db ? Cout : 0);
        hipLaunchKernelGGL(wgrad_bx_reduce_kernel, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ws, p.nsplit, n4, Cout, dw,
                           db, (accumulate || b0 > 0) ? 1 : 0);
        rc = launch_status(__func__);
        if (rc) return rc;
    }
    return 0;
}
