// Winograd F(3x3, 2x2) weight-gradient kernel template (see conv_wino_wgrad.hip for the derivation and the host side).  Each block
// shape is instantiated in its own translation unit (conv_wino_wgrad_<shape>.hip): a fully unrolled K-chunk of 256 MFMAs with their
// hand-placed fillers takes hipcc about a minute per instantiation.
#pragma once
#include <type_traits>
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
    int xcd;             // 1: workgroup order by XCD (below)
};

template <int TBH, int TBW>
__global__ __launch_bounds__(512, 2) void wino_wgrad_kernel(WinoWgArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int PW = 2 * TBW + 2, PHT = 2 * TBH + 2, DW = 2 * TBW, DH = 2 * TBH, NS = (TBH / 2) * TBW;
    // LDS: per buffer the x halo patch and the dy block as sixteen CHANNEL-QUAD PLANES [quad][pixel][4] -- the staging writes are
    // 16-byte pieces of consecutive pixels (what the C4 layout delivers: consecutive lanes = consecutive pixels of a patch row), the operand
    // reads (one float per lane: channel li of 32, ds_read_b32) spread over the banks because the plane pitches are 4 mod 8 dwords x 4
    constexpr int NPX = PW * PHT, NDY = DW * DH;
    constexpr int XPL = WG_XPL, YPL = WG_YPL;            // plane pitches in floats (724 = 20 mod 32, 516 = 4 mod 32)
    constexpr int XPF = 16 * XPL, BUF = XPF + 16 * YPL;   // floats: x patch, dy tile block; two buffers
    static_assert(PW * PHT <= 180 && DW * DH <= 128 && NS <= 16 && NS >= 13, "block shape (the staging schedule runs to step 11, before the barrier of step NS - 1)");
    const WinoGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    // EIGHT waves, two per SIMD: wave = (input-channel tile cw, output-channel tile nw, position half ph) owns 32 x 32 channels x the
    // eight positions p = 4 xi + nu with xi (the vertical index) in {2 ph, 2 ph + 1} = eight 32 x 32 accumulators = 128 registers.
    // Round 3's four waves (one per SIMD, sixteen positions = 256 accumulator registers each) left every LDS-read and barrier wait of a
    // wave exposed -- the ablation (profiles/r04_wgrad_ablation.txt) charges 21 % of the kernel to the operand reads and their waits, 4 %
    // to staging, and a third fewer transform operations alone (positions split over FOUR waves, 256 registers each) bought 1.8 % -- so
    // the positions are split to HALVE THE ACCUMULATORS: a second wave per SIMD issues MFMAs while the first waits.  A wave needs three
    // of the four patch rows and runs half of the x transform (vertical 2 of 4 outputs per column, horizontal 8 of 16) and half of its
    // tile's dy transform.  Raw position sums go to the workspace as before; nothing is exchanged between the waves.
    const int cw = wave >> 2, nw = (wave >> 1) & 1, ph = wave & 1;
    // Workgroup order: the hardware deals consecutive workgroups round the eight XCDs, each with its own L2; logical id = xcd_remap(...)
    // gives XCD i a CONTIGUOUS range of (split, tile) ids -- whole K splits (x and dy of a split are then fetched by one XCD only), or,
    // on the 64-tile layers, half the tiles of one split (four input-channel tiles x all eight output-channel tiles) -- instead of one
    // residue class mod 8 = one output-channel tile of EVERY split, which fetched x eight times (profiles/r04_traffic_layers.md).
    const int wid = a.xcd ? xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int cn = wid % (a.ncb * a.nnb), split = wid / (a.ncb * a.nnb);
    const int cb = cn / a.nnb, nb = cn - cb * a.nnb;
    const int C = g.C, N = g.N;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)((long)g.B * g.H * g.W * C * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, (int)((long)g.B * g.H * g.W * N * 4), 0x00020000);

    const int blk0 = split * a.cps;
    int nch = g.nblocks - blk0;
    if (nch > a.cps) nch = a.cps;

    // staging slots: a thread owns ONE pixel of the x halo patch (pixel tid & 255 < NPX <= 180, channel-quad parity tid >> 8) and ONE pixel
    // of the dy block (pixel tid & 127 < NDY <= 128, channel quad mod 4 = tid >> 7): per block TWO voffsets (the pixel's 16 bytes in the
    // thread's first channel-quad plane, or WOOB) and the further quads as SCALAR plane offsets -- slot i < 8: x quad 2 i + (tid >> 8);
    // slot 8 + i, i < 4: dy quad 4 i + (tid >> 7).  (Slots that were float4 index tid + 256 i of [quad][pixel] cost two divisions and a
    // bounds check per slot and block: + 0.4 ms per step, round 4.)
    float4 st[4];
    const int xpix = tid & 255, xpy = xpix / PW, xpx = xpix - xpy * PW;   // (compile-time divisors)
    const int ypix = tid & 127, ypy = ypix / DW, ypx = ypix - ypy * DW;
    const bool xlive = xpix < NPX, ylive = ypix < NDY;
    const unsigned plane_b = (unsigned)g.H * (unsigned)g.W * 16u;   // bytes of one channel-quad plane of an image
    const int xq0 = cb * 16 + (tid >> 8), yq0 = nb * 16 + (tid >> 7);
    const int xst = (tid >> 8) * XPL + xpix * 4, yst = XPF + (tid >> 7) * YPL + ypix * 4;   // LDS float index of the thread's first plane (+ buf * BUF + further quads)
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
        if (i < 8) st[k] = wbufload(rx, xvoff, (unsigned)(2 * i) * plane_b);
        else st[k] = wbufload(ry, yvoff, (unsigned)(4 * (i - 8)) * plane_b);
    };
    auto lstore1 = [&](int buf, int i, int k) {
        if (i < 8) {
            if (xlive) *reinterpret_cast<float4*>(&smem[buf * BUF + xst + 2 * i * XPL]) = st[k];
        } else {
            if (ylive) *reinterpret_cast<float4*>(&smem[buf * BUF + yst + 4 * (i - 8) * YPL]) = st[k];
        }
    };

    f32x16 acc[8];
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
    float dbacc = 0.f;

    const int xbase = ((2 * lh) * PW) * 4 + (cw * 8 + (li >> 2)) * XPL + (li & 3);          // + buf * BUF + ((4 r + i) * PW + 2 tx + j) * 4
    const int ybase = XPF + ((2 * lh) * DW) * 4 + (nw * 8 + (li >> 2)) * YPL + (li & 3);    // + buf * BUF + ((4 r + a) * DW + 2 tx + b) * 4

    auto body = [&](auto phc) {
        constexpr int PH = decltype(phc)::value;
        // operands of one step, double-buffered by ob: ua[ob][4 xl + nu] = U_p[tile][c] (xl = xi - 2 PH); the dy side keeps its raw values
        // e, the vertical terms v and the two computed horizontal terms per xl -- the MFMA takes whichever register holds E_p (no copies)
        float ua[2][8], dv[3][4], tc[4][2];
        float ey[2][2][2], vy[2][2], hy[2][2][2];   // [ob]: e[aa][bb]; v[bb]; h[xl][0: sum, 1: difference]
        auto evalue = [&](int ob, int pl) -> float {   // E_p, p = 4 (2 PH + xl) + nu, from the registers above
            const int xl = pl >> 2, nu = pl & 3;
            // row term of bb: PH 0: xi 0 -> e[0][bb], xi 1 -> v[bb] = e0 + e1;  PH 1: xi 2 -> v[bb] = e0 - e1, xi 3 -> e[1][bb]
            auto row = [&](int bb) -> float { return PH == 0 ? (xl == 0 ? ey[ob][0][bb] : vy[ob][bb]) : (xl == 0 ? vy[ob][bb] : ey[ob][1][bb]); };
            return nu == 0 ? row(0) : nu == 3 ? row(1) : hy[ob][xl][nu - 1];
        };
        // micro-operation k of preparing step sn (from LDS buffer `buf`) into operand set ob
        auto prep = [&](int buf, int sn, int ob, int k) {
            const int r = sn / TBW, tx = sn - r * TBW;
            const bool fresh = tx == 0;
            const int ncol = fresh ? 4 : 2, nread = 3 * ncol;
            if (k < nread) {   // patch rows PH .. PH + 2 of the new columns
                const int idx = fresh ? k : 6 + k, j = idx / 3, i = idx % 3;
                if (WG_ABL & 2) return;
                dv[i][j] = smem[buf * BUF + xbase + ((4 * r + PH + i) * PW + 2 * tx + j) * 4];
                return;
            }
            k -= nread;
            if (k < 4) {
                const int aa = k >> 1, bb = k & 1;
                if (WG_ABL & 2) return;
                ey[ob][aa][bb] = smem[buf * BUF + ybase + ((4 * r + aa) * DW + 2 * tx + bb) * 4];
                return;
            }
            k -= 4;
            if (WG_ABL & 1) return;
            if (k < 2 * ncol) {   // vertical transforms of the new patch columns (the two older ones carry over from the previous step)
                const int idx = fresh ? k : 4 + k, j = idx >> 1, xl = idx & 1;
                if (!fresh && k == 0) { tc[0][0] = tc[2][0]; tc[0][1] = tc[2][1]; tc[1][0] = tc[3][0]; tc[1][1] = tc[3][1]; }
                // B^T rows: xi 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3; dv[i] = patch row PH + i
                if (PH == 0) tc[j][xl] = xl == 0 ? dv[0][j] - dv[2][j] : dv[1][j] + dv[2][j];
                else tc[j][xl] = xl == 0 ? dv[1][j] - dv[0][j] : dv[0][j] - dv[2][j];
                return;
            }
            k -= 2 * ncol;
            if (k < 8) {
                const int xl = k >> 2, nu = k & 3;
                ua[ob][k] = nu == 0 ? tc[0][xl] - tc[2][xl] : nu == 1 ? tc[1][xl] + tc[2][xl] : nu == 2 ? tc[2][xl] - tc[1][xl] : tc[1][xl] - tc[3][xl];
                return;
            }
            k -= 8;
            if (k < 2) {   // vertical G' of dy column bb: the one computed term of this position half
                vy[ob][k] = PH == 0 ? ey[ob][0][k] + ey[ob][1][k] : ey[ob][0][k] - ey[ob][1][k];
                return;
            }
            k -= 2;
            if (k < 4) {   // horizontal G': sum and difference of the two row terms
                const int xl = k >> 1, sd = k & 1;
                const float r0 = PH == 0 ? (xl == 0 ? ey[ob][0][0] : vy[ob][0]) : (xl == 0 ? vy[ob][0] : ey[ob][1][0]);
                const float r1 = PH == 0 ? (xl == 0 ? ey[ob][0][1] : vy[ob][1]) : (xl == 0 ? vy[ob][1] : ey[ob][1][1]);
                hy[ob][xl][sd] = sd == 0 ? r0 + r1 : r0 - r1;
                return;
            }
            k -= 4;
            if (PH == 0 && k == 0) dbacc += hy[ob][1][0];   // E_(1,1) = the sum of the tile's four dy values
        };
#define WSB() __builtin_amdgcn_sched_barrier(0)
        constexpr int TOT_F = 12 + 4 + 8 + 8 + 2 + 4 + 1, TOT_C = 6 + 4 + 4 + 8 + 2 + 4 + 1;   // micro-operations of a fresh / a carried step
        if (nch > 0) {
            set_block(blk0);
#pragma unroll
            for (int h = 0; h < 3; ++h) {
#pragma unroll
                for (int k = 0; k < 4; ++k) gload1(4 * h + k, k);
#pragma unroll
                for (int k = 0; k < 4; ++k) lstore1(0, 4 * h + k, k);
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < TOT_F; ++k) prep(0, 0, 0, k);
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
                const int sn = last_step ? 0 : s + 1, nbuf = last_step ? buf ^ 1 : buf, nob = ob ^ 1;
                const int total = (sn % TBW) == 0 ? TOT_F : TOT_C, per = (total + 7) / 8;
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[ob][m], evalue(ob, m), acc[m], 0, 0, 0);
                    WSB();
                    if (nxt) {
#pragma unroll
                        for (int k2 = 0; k2 < 5; ++k2)
                            if (k2 < per && m * per + k2 < total) prep(nbuf, sn, nob, m * per + k2);
                    }
                    if (more && m < 4 && !(WG_ABL & 4)) {   // the next block's data: three batches of four slots, each loaded three steps before it is written
                        if (s == 0) gload1(m, m);
                        if (s == 3) lstore1(buf ^ 1, m, m);
                        if (s == 4) gload1(4 + m, m);
                        if (s == 7) lstore1(buf ^ 1, 4 + m, m);
                        if (s == 8) gload1(8 + m, m);
                        if (s == 11) lstore1(buf ^ 1, 8 + m, m);
                    }
                    WSB();
                }
            }
        }
#undef WSB
        asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");

        // raw position sums of this split: acc[pl][r] = S_p[c0 + 8 (r >> 2) + 4 lh + (r & 3)][n0 + li], p = 8 PH + pl
        const int c0 = cb * 64 + cw * 32, n0 = nb * 64 + nw * 32;
        float* o = a.ws + (long)split * 16 * C * N + (long)n0 + li;
#pragma unroll
        for (int pl = 0; pl < 8; ++pl)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[((long)(8 * PH + pl) * C + c0 + 8 * (r >> 2) + 4 * lh + (r & 3)) * N] = acc[pl][r];
        if (PH == 0 && cb == 0 && cw == 0) {
            const float v = dbacc + __shfl_xor(dbacc, 32, 64);
            if (lh == 0) a.ws[(long)a.nsplit * 16 * C * N + (long)split * N + n0 + li] = v;
        }
    };
    if (ph) body(std::integral_constant<int, 1>{});
    else body(std::integral_constant<int, 0>{});
}

constexpr int WINO_WG_LDS_BYTES = 2 * 16 * (WG_XPL + WG_YPL) * 4;   // 158 720

template <int TBH, int TBW>
static int launch_wino_wgrad(hipStream_t st, const WinoWgArgs& a) {
    static int once = [] {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wino_wgrad_kernel<TBH, TBW>), hipFuncAttributeMaxDynamicSharedMemorySize, WINO_WG_LDS_BYTES);
        return e == hipSuccess ? 0 : fail((int)e, "%s: hipFuncSetAttribute failed", "wino wgrad kernel");
    }();
    if (once) return once;
    hipLaunchKernelGGL((wino_wgrad_kernel<TBH, TBW>), dim3(a.ncb * a.nnb * a.nsplit), dim3(512), WINO_WG_LDS_BYTES, st, a);
    return launch_status("conv wino wgrad");
}


int launch_wino_wgrad_4x8(hipStream_t st, const WinoWgArgs& a);
int launch_wino_wgrad_4x7(hipStream_t st, const WinoWgArgs& a);
int launch_wino_wgrad_2x14(hipStream_t st, const WinoWgArgs& a);

}  // namespace vc
