// 3x3 / stride 1 / SAME convolution, DIRECT form, on the bf16 matrix pipe, second kernel: the activation operand is split IN REGISTERS.
// Forward and data gradient of utils/image_embeddings.py:36-212 in the split-bf16 arithmetic of gemm_bf16x3_core.h (a = hi + lo, three
// v_mfma_f32_32x32x16_bf16 products, f32 accumulators); f32 activations in the C4 layout.  The opt-in mode of
// vc_gemm_set_precision(1) / Trainer(precision="bf16x3"), NOT the reference's tf.float32 arithmetic.
//
//   D[co, pixel] += W_tap[co, ci] . X[ci, pixel + tap]      A = weights (row = output channel), B = activations (column = pixel)
//
// What conv_bx.hip (the first kernel) pays for -- splitting the patch into an LDS image, a fragment read of BOTH operands for every
// tap, one barrier per tap in an eight-wave workgroup whose phases do not overlap -- is removed the way conv_wgrad_bx.hip removed it:
//   * one wave per SIMD (four per workgroup), 512 registers: a wave keeps the weights of a k-step (16 input channels x 9 taps x 64
//     output channels = 36 fragments, 144 registers) RESIDENT and eight 32 x 32 accumulators (4 output rows x 32 pixels x 64 channels);
//   * the patch goes to the LDS as it arrives from HBM (C4: [quad][pixel][4], a straight 16-byte copy); a lane owns one pixel column
//     and reads, per WINDOW (input row i of 6, horizontal tap tx of 3), the two quads of its pixel with two ds_read_b128 and splits
//     the eight values in registers (four pairs x four stages, each stage behind a different MFMA); a window feeds the MFMAs of up
//     to three output rows (vertical taps) x two channel tiles x three split terms = 6 / 12 / 18 MFMAs;
//   * per k-step and wave: 216 MFMAs against 432 VALU instructions, 72 ds_read_b128 and 19 + 19 staging instructions, every one of
//     them placed behind a specific MFMA (`ops`); ONE barrier per k-step, issued where the wave has read the last window of the
//     current LDS image -- the first windows and weights of the next image are read behind it, under the k-step's last MFMAs.
// Geometry: lane = pixel (dx = lane % LW of LW = 32 or 16 columns; LW = 16: lanes 16..31 are the same columns four rows lower), rows
// are padded global rows b (H + 1) + y (conv_bx.hip), a workgroup tile = 16 x 32 or 32 x 16 pixels x 64 output channels, persistent
// workgroups walk (tile, k-step) as one flat sequence, channel tile slowest.  Epilogue: bias + ReLU (forward) or the ReLU mask of
// the layer's input (data gradient), 16-byte stores straight from the accumulators (a register quad = four consecutive channels).
#include <stdlib.h>
#include <type_traits>
#include <utility>
#include "conv_wino.h"
#include "gemm_bf16x3_core.h"

// `make c2abl`: C2_ABL = bit mask that REMOVES parts (results wrong; timing only): 1 split arithmetic, 2 window reads, 4 staging,
// 8 weight-fragment reads, 16 everything between the MFMAs (operands frozen after the first k-step: the matrix pipe on real data)
#ifndef C2_ABL
#define C2_ABL 0
#endif

namespace vc {

constexpr int C2_NPIX = 612;                     // patch pixels: 18 x 34 (LW 32) or 34 x 18 (LW 16)
constexpr int C2_XPL = C2_NPIX * 4;              // floats per channel-quad plane
constexpr int C2_XBYTES = 4 * C2_XPL * 4;        // 39 168: four quads = one k-step of 16 channels
constexpr int C2_WBYTES = 9 * 2 * 2 * 1024;      // 36 864: [tap][channel tile][hi | lo][lane][16 B]
constexpr int C2_BUF = C2_XBYTES + C2_WBYTES;    // 76 032
constexpr int C2_LDS = 2 * C2_BUF;               // 152 064

struct ConvBx2Args {
    const float* x;      // [B][C/4][H][W][4]
    const char* wp;      // packed weights [N / 64][C / 16][C2_WBYTES] (vc_conv3x3_bx2_pack_f32)
    float* out;          // [B][N/4][H][W][4]
    const float* aux;    // forward: bias [N] or null; data gradient: ReLU source in the layout of out, or null
    int B, H, W, C, N, relu;
    int col_tiles, ptiles, ntiles, grows;
    unsigned m_hp1, m_coltiles, one_coltiles, m_ptiles, one_ptiles;
};

#define C2SB() __builtin_amdgcn_sched_barrier(0)

// The k-step's 216 MFMAs as compile-time data: window w = (input row i = w / 3, horizontal tap tx = w % 3) owns 6 / 12 / 18 / 18 / 12 / 6
// consecutive MFMAs (rows 0..5: one, two, three, three, two, one vertical taps x two channel tiles x three split terms); windows
// 18..20 are the next k-step's 0..2.  Everything that depends on the MFMA index is resolved with `if constexpr` in generic lambdas
// (std::integral_constant arguments, expanded by c2_for): left to the unroller, the same schedule did not compile in half an hour.
constexpr int c2_wstart(int w) {
    const int i = w / 3, tx = w % 3;
    const int base = i == 0 ? 0 : i == 1 ? 18 : i == 2 ? 54 : i == 3 ? 108 : i == 4 ? 162 : i == 5 ? 198 : 216;
    const int len = (i == 0 || i >= 5) ? 6 : (i == 1 || i == 4) ? 12 : 18;
    return base + tx * len;
}
struct C2Slot { int w, i, tx, k, term, ty, T, r; };
constexpr C2Slot c2_slot(int n) {
    int w = 17;
    while (w > 0 && n < c2_wstart(w)) --w;
    const int i = w / 3, tx = w % 3;
    const int nty = i < 2 ? i + 1 : i > 3 ? 6 - i : 3, ty0 = i > 3 ? i - 3 : 0;
    const int m = n - c2_wstart(w);
    const int term = m / (2 * nty), ty = ty0 + (m % (2 * nty)) / 2, T = m & 1;
    return C2Slot{w, i, tx, w % 3, term, ty, T, i - ty};
}
template <class F, int... I>
__device__ __forceinline__ void c2_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void c2_for(F&& f) { c2_for_impl(f, std::make_integer_sequence<int, N>{}); }

template <int KIND, int LW>
__global__ __launch_bounds__(256, 1) void conv_bx2_kernel(ConvBx2Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem2[];
    constexpr int PW = LW + 2, RWV = 4 * (32 / LW), TR = 4 * RWV;   // patch width; rows per wave; rows per workgroup tile
    static_assert((TR + 2) * PW == C2_NPIX, "patch size");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int dyl = li / LW, dx = li % LW;
    const int H = a.H, W = a.W, C = a.C, N = a.N;
    const int nk = C >> 4;
    const unsigned plane_b = (unsigned)H * (unsigned)W * 16u;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)((long)a.B * H * W * C * 4), 0x00020000);
    auto bdiv = [&](unsigned n, unsigned m, unsigned one) -> unsigned { return (__umulhi(n, m) & ~one) | (n & one); };

    struct Tile { int co_t, pr0, x0; };
    auto decode = [&](int id) {
        const unsigned co_t = bdiv((unsigned)id, a.m_ptiles, a.one_ptiles), pt = (unsigned)id - co_t * a.ptiles;
        const unsigned row_t = bdiv(pt, a.m_coltiles, a.one_coltiles), col_t = pt - row_t * a.col_tiles;
        return Tile{(int)co_t, (int)row_t * TR, (int)col_t * LW};
    };

    // ---- staging: per k-step 2448 patch pieces (slot s = tid + 256 u: quad s / 612, pixel s % 612 -> LDS byte 16 s) and 2304 weight
    // pieces (a linear copy), 16 bytes each: ten + nine per thread, in three batches of registers
    int s_py[10], s_px[10], s_q[10];
#pragma unroll
    for (int u = 0; u < 10; ++u) {
        const int s = tid + 256 * u, q = s / C2_NPIX, pix = s - q * C2_NPIX;
        s_q[u] = q; s_py[u] = pix / PW; s_px[u] = pix - s_py[u] * PW;
    }
    unsigned pvoff[10];
    unsigned soff = 0;          // of the k-step being staged
    const char* wsrc = a.wp;    // of the k-step being staged (this thread's first piece)
    auto set_patch = [&](const Tile& t, bool live) {
#pragma unroll
        for (int u = 0; u < 10; ++u) {
            const int g = t.pr0 - 1 + s_py[u], col = t.x0 - 1 + s_px[u];
            const unsigned b = __umulhi((unsigned)max(g, 0), a.m_hp1);
            const int y = g - (int)b * (H + 1);
            const unsigned ok = 0u - (unsigned)((int)live & (int)(tid + 256 * u < 4 * C2_NPIX) & (int)((unsigned)g < (unsigned)a.grows) & (int)(y < H) & (int)((unsigned)col < (unsigned)W));
            const unsigned addr = (unsigned)((((int)b * (C >> 2) + s_q[u]) * H + y) * W + col) * 16u;
            pvoff[u] = (addr & ok) | (WOOB & ~ok);
        }
    };
    u32x4 st[7];
    auto gload = [&](int s) -> u32x4 {   // slot 0..9: patch; 10..18: weights
        if (s < 10) { const float4 v = wbufload(rx, pvoff[s], soff); return __builtin_bit_cast(u32x4, v); }
        return *reinterpret_cast<const u32x4*>(wsrc + (size_t)(s - 10) * 4096);
    };
    auto lstore = [&](int img, int s, const u32x4& v) {   // img: byte offset of the LDS image being filled
        if (s < 10) { if (tid + 256 * s < 4 * C2_NPIX) *reinterpret_cast<u32x4*>(smem2 + img + (tid + 256 * s) * 16) = v; }
        else *reinterpret_cast<u32x4*>(smem2 + img + C2_XBYTES + (tid + 256 * (s - 10)) * 16) = v;
    };

    // ---- operands --------------------------------------------------------------------------------------------------------------
    // lane (li, lh): pixel column dx of the wave's rows; input channels 8 lh .. 8 lh + 7 of the k-step = quads 2 lh, 2 lh + 1
    const int xl = (2 * lh) * C2_XPL * 4 + ((wave * RWV + 4 * dyl) * PW + dx) * 16;   // byte offset; window (i, tx): + (i PW + tx) 16; second quad + XPL 4
    const int al = C2_XBYTES + lane * 16;                                             // A fragment ((tap * 2 + T) * 2 + part) * 1024 + this
    u32x4 Afr[3][3][2][2];   // [ty][tx][channel tile][hi | lo]
    u32x4 Bop[3][2];         // three windows in flight: [w % 3][hi | lo]
    float raw[3][8], tmp[3][8];
    auto aread = [&](int img, int ty, int tx) {   // the four fragments of one tap
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                if (C2_ABL & 8) { Afr[ty][tx][T][part] = u32x4{(unsigned)(ty + tx), (unsigned)lane, 0x3f803f80u, (unsigned)(T + part)}; continue; }
                Afr[ty][tx][T][part] = *reinterpret_cast<const u32x4*>(smem2 + img + al + ((((ty * 3 + tx) * 2 + T) * 2 + part) << 10));
            }
    };
    // window w = (input row i = w / 3, horizontal tap tx = w % 3), register set w % 3: sg -1 = read, then pair p (0..3) stage 0..3
    auto wread = [&](int img, int w) {
        const int i = (w % 18) / 3, tx = w % 3, k = w % 3;
        if (C2_ABL & 2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) raw[k][e] = (float)(e + w);
            return;
        }
        const float4 q0 = *reinterpret_cast<const float4*>(smem2 + img + xl + (i * PW + tx) * 16);
        const float4 q1 = *reinterpret_cast<const float4*>(smem2 + img + xl + C2_XPL * 4 + (i * PW + tx) * 16);
        raw[k][0] = q0.x; raw[k][1] = q0.y; raw[k][2] = q0.z; raw[k][3] = q0.w;
        raw[k][4] = q1.x; raw[k][5] = q1.y; raw[k][6] = q1.z; raw[k][7] = q1.w;
    };
    auto wstage = [&](int w, int p, int sg) {
        const int k = w % 3;
        const float v0 = raw[k][2 * p], v1 = raw[k][2 * p + 1];
        if (C2_ABL & 1) {
            if (sg == 0) Bop[k][0][p] = __float_as_uint(v0);
            if (sg == 3) Bop[k][1][p] = __float_as_uint(v1);
            return;
        }
        if (sg == 0) Bop[k][0][p] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v0, v1}, bf16x2));
        else if (sg == 1) { tmp[k][2 * p] = __uint_as_float(Bop[k][0][p] << 16); tmp[k][2 * p + 1] = __uint_as_float(Bop[k][0][p] & 0xffff0000u); }
        else if (sg == 2) { tmp[k][2 * p] = v0 - tmp[k][2 * p]; tmp[k][2 * p + 1] = v1 - tmp[k][2 * p + 1]; }
        else Bop[k][1][p] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{tmp[k][2 * p], tmp[k][2 * p + 1]}, bf16x2));
    };

    f32x16 acc[4][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int T = 0; T < 2; ++T)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[r][T][e] = 0.f;
    };
    // epilogue of tile t: acc[r][T][4 j + e] = channel 64 co_t + 32 T + 8 j + 4 lh + e of pixel (row pr0 + wave RWV + r + 4 dyl, column x0 + dx)
    auto store_tile = [&](const Tile& t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int g = t.pr0 + wave * RWV + r + 4 * dyl, xx = t.x0 + dx;
            const unsigned b = __umulhi((unsigned)g, a.m_hp1);
            const int y = g - (int)b * (H + 1);
            if (g >= a.grows || y >= H || xx >= W) continue;
            const size_t pix = ((size_t)b * (size_t)(N >> 2) * H + y) * W + xx;   // + quad * H * W, in 16-byte elements
#pragma unroll
            for (int T = 0; T < 2; ++T)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int co = t.co_t * 64 + 32 * T + 8 * j + 4 * lh;
                    float4 v = make_float4(acc[r][T][4 * j], acc[r][T][4 * j + 1], acc[r][T][4 * j + 2], acc[r][T][4 * j + 3]);
                    const size_t e = (pix + (size_t)(co >> 2) * H * W) * 4;
                    if (KIND == 0) {
                        if (a.aux) {
                            const float4 bb = *reinterpret_cast<const float4*>(a.aux + co);
                            v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
                        }
                        if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    } else if (a.aux) {
                        const float4 s = *reinterpret_cast<const float4*>(a.aux + e);
                        v.x = s.x > 0.f ? v.x : 0.f; v.y = s.y > 0.f ? v.y : 0.f; v.z = s.z > 0.f ? v.z : 0.f; v.w = s.w > 0.f ? v.w : 0.f;
                    }
                    *reinterpret_cast<float4*>(a.out + e) = v;
                }
        }
    };

    // ---- the staged stream: (tile, k-step) two steps ahead of the MFMAs -------------------------------------------------------------
    int tile = blockIdx.x;
    if (tile >= a.ntiles) return;
    Tile cur = decode(tile);
    int ks = 0;
    int t2 = tile, k2 = 0;   // the k-step whose descriptor (pvoff, soff, wsrc) is current
    auto describe = [&](bool newtile) {   // descriptor of (t2, k2)
        const bool live = t2 < a.ntiles;
        const Tile tt = decode(live ? t2 : tile);
        if (newtile) set_patch(tt, live);
        soff = (unsigned)(4 * k2) * plane_b;
        wsrc = a.wp + ((size_t)tt.co_t * nk + k2) * C2_WBYTES + (size_t)tid * 16;
    };
    auto advance = [&]() {   // (t2, k2) <- the following k-step of the flat sequence
        const bool wrap = k2 + 1 >= nk;
        k2 = wrap ? 0 : k2 + 1;
        t2 = wrap ? t2 + (int)gridDim.x : t2;
        describe(wrap);
    };
    int cimg = C2_BUF, nimg = 0;   // byte offsets of the LDS image of the current / the next k-step (the prologue fills `nimg`, then swaps)
    describe(true);
#pragma unroll
    for (int h = 0; h < 3; ++h) {
        const int s0 = h == 0 ? 0 : h == 1 ? 7 : 13, ns = h == 0 ? 7 : 6;
#pragma unroll
        for (int i = 0; i < 7; ++i)
            if (i < ns) st[i] = gload(s0 + i);
#pragma unroll
        for (int i = 0; i < 7; ++i)
            if (i < ns) lstore(nimg, s0 + i, st[i]);
    }
    advance();
#pragma unroll
    for (int i = 0; i < 7; ++i) st[i] = gload(i);   // batch A of the second k-step
    zero_acc();
    __syncthreads();
    // ---- the operations behind MFMA n (0..215) of a k-step ---------------------------------------------------------------------------
    // A window is read 13 intervals before its first MFMA S(w) and its four pairs enter the four split stages two intervals apart
    // from S(w) - 10.  Windows 18..20 are the next k-step's 0..2, read from the other image behind the barrier (at 198, after window
    // 17's read at 197); windows 0..2 finish, in the first intervals, the stages they entered as 18..20 of the previous k-step.
    auto ops_win = [&](auto nc, auto wminc, auto wmaxc) {
        constexpr int n = decltype(nc)::value, wmin = decltype(wminc)::value, wmax = decltype(wmaxc)::value;
        c2_for<21>([&](auto wc) {
            constexpr int w = decltype(wc)::value, S = c2_wstart(w);
            if constexpr (w >= wmin && w <= wmax) {
                if constexpr (n == S - 13) wread(w >= 18 ? nimg : cimg, w);
                if constexpr (n - (S - 10) >= 0 && n - (S - 10) < 4) wstage(w, 0, n - (S - 10));
                if constexpr (n - (S - 8) >= 0 && n - (S - 8) < 4) wstage(w, 1, n - (S - 8));
                if constexpr (n - (S - 6) >= 0 && n - (S - 6) < 4) wstage(w, 2, n - (S - 6));
                if constexpr (n - (S - 4) >= 0 && n - (S - 4) < 4) wstage(w, 3, n - (S - 4));
            }
        });
        if constexpr (n >= 200 && n < 203) aread(nimg, 0, n - 200);   // vertical tap 0 of the next k-step (its registers are free from 162 on)
    };
    // ---- prologue, second part: what intervals 200..215 of a previous k-step would have prepared from this image ------------------
    c2_for<16>([&](auto jc) { ops_win(std::integral_constant<int, 200 + decltype(jc)::value>{}, std::integral_constant<int, 18>{}, std::integral_constant<int, 20>{}); });
    if (C2_ABL & 16) {   // (frozen operands: complete windows 0..2 and every fragment)
        c2_for<12>([&](auto jc) { ops_win(jc, std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{}); });
#pragma unroll
        for (int ty = 1; ty < 3; ++ty)
#pragma unroll
            for (int tx = 0; tx < 3; ++tx) aread(nimg, ty, tx);
    }
    { const int t = cimg; cimg = nimg; nimg = t; }
    C2SB();

    auto ops = [&](auto nc) {
        constexpr int n = decltype(nc)::value;
        if constexpr ((C2_ABL & 16) != 0) return;
        ops_win(nc, std::integral_constant<int, 0>{}, std::integral_constant<int, 20>{});
        // weight fragments: vertical tap 1 of this k-step (needed from 18), tap 2 (from 54)
        if constexpr (n < 3) aread(cimg, 1, n);
        if constexpr (n >= 20 && n < 23) aread(cimg, 2, n - 20);
        if constexpr (!(C2_ABL & 4)) {
            if constexpr (n >= 40 && n < 47) lstore(nimg, n - 40, st[n - 40]);              // batch A (slots 0..6) of the next k-step, loaded at 204.. of the previous one
            if constexpr (n >= 47 && n < 53) st[n - 47] = gload(7 + n - 47);              // batch B (7..12)
            if constexpr (n >= 110 && n < 116) lstore(nimg, 7 + n - 110, st[n - 110]);
            if constexpr (n >= 116 && n < 122) st[n - 116] = gload(13 + n - 116);         // batch C (13..18)
            if constexpr (n >= 180 && n < 186) lstore(nimg, 13 + n - 180, st[n - 180]);
            if constexpr (n == 190) advance();                                            // descriptor of the k-step after next
            if constexpr (n >= 204 && n < 211) st[n - 204] = gload(n - 204);
        }
    };

    for (;;) {
        c2_for<216>([&](auto nc) {
            constexpr int n = decltype(nc)::value;
            constexpr C2Slot si = c2_slot(n);
            if constexpr (n == 198) __syncthreads();   // every wave has read the last window of this image; the other image is complete
            acc[si.r][si.T] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Afr[si.ty][si.tx][si.T][si.term == 1 ? 1 : 0]),
                                                                       __builtin_bit_cast(bf16x8, Bop[si.k][si.term == 2 ? 1 : 0]), acc[si.r][si.T], 0, 0, 0);
            C2SB();
            ops(nc);
            C2SB();
        });
        { const int t = cimg; cimg = nimg; nimg = t; }
        if (++ks == nk) {
            store_tile(cur);
            tile += (int)gridDim.x;
            if (tile >= a.ntiles) break;
            cur = decode(tile);
            ks = 0;
            zero_acc();
        }
    }
}
#undef C2SB

// w [3,3,Cin,Cout] HWIO -> [co block of 64][k-step of 16][tap][T][hi | lo][lane][8 bf16]; transpose: rows = Cin (the data gradient's output
// channels), k = Cout, taps flipped.  One thread per (block, k-step, tap, T, lane).
__global__ __launch_bounds__(256) void conv_bx2_pack_kernel(const float* __restrict__ w, int Cin, int Cout, int transpose, char* __restrict__ wp) {
    const int rows = transpose ? Cin : Cout, kdim = transpose ? Cout : Cin;
    const int nk = kdim >> 4;
    const long total = (long)(rows >> 6) * nk * 9 * 2 * 64;
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= total) return;
    const int lane = (int)(id & 63), T = (int)((id >> 6) & 1);
    const long rest = id >> 7;
    const int tap = (int)(rest % 9);
    const long r2 = rest / 9;
    const int ks = (int)(r2 % nk), cb = (int)(r2 / nk);
    const int li = lane & 31, lh = lane >> 5;
    const int row = cb * 64 + T * 32 + li, k0 = ks * 16 + lh * 8;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
        v[j] = transpose ? w[((long)(8 - tap) * Cin + row) * Cout + k0 + j] : w[((long)tap * Cin + k0 + j) * Cout + row];
    u32x4 hi, lo;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        unsigned h, l;
        split_pair(v[2 * p], v[2 * p + 1], h, l);
        hi[p] = h; lo[p] = l;
    }
    char* dst = wp + ((size_t)cb * nk + ks) * C2_WBYTES + (size_t)((tap * 2 + T) * 2) * 1024 + (size_t)lane * 16;
    *reinterpret_cast<u32x4*>(dst) = hi;
    *reinterpret_cast<u32x4*>(dst + 1024) = lo;
}

struct ConvBx2Plan {
    bool ok;
    int lw, col_tiles, row_tiles, ptiles, ntiles, grows;
};

static ConvBx2Plan plan_conv_bx2(int B, int H, int W, int C, int N) {   // C = contraction channels, N = produced channels
    ConvBx2Plan p;
    p.ok = false; p.lw = 32; p.col_tiles = p.row_tiles = p.ptiles = p.ntiles = p.grows = 0;
    if (B <= 0 || H < 1 || W < 1 || C <= 0 || N <= 0 || C % 16 || N % 64) return p;
    if ((long)B * H * W * (long)(C > N ? C : N) * 4 > 0x7fffffffL) return p;
    if ((long)B * (H + 1) > 0x0fffffffL) return p;
    p.grows = B * (H + 1) - 1;
    static const int force = getenv("VC_CONVBX2_LW") ? atoi(getenv("VC_CONVBX2_LW")) : 0;
    // 32 columns x 16 rows or 16 columns x 32 rows per workgroup: the shape that wastes fewer pixel slots
    double best = -1.0;
    for (int lw = 32; lw >= 16; lw >>= 1) {
        if (force && lw != force) continue;
        const int tr = lw == 32 ? 16 : 32;
        const double eff = (double)W * p.grows / ((double)cdiv(W, lw) * lw * (double)cdiv(p.grows, tr) * tr);
        if (eff > best + 1e-9) { best = eff; p.lw = lw; }
    }
    const int tr = p.lw == 32 ? 16 : 32;
    p.col_tiles = cdiv(W, p.lw);
    p.row_tiles = cdiv(p.grows, tr);
    p.ptiles = p.col_tiles * p.row_tiles;
    p.ntiles = p.ptiles * (N / 64);
    p.ok = true;
    return p;
}

template <int KIND>
static int launch_conv_bx2(hipStream_t st, int B, int H, int W, int C, int N, const float* in, const void* wp, const float* aux, float* out, int relu) {
    const ConvBx2Plan p = plan_conv_bx2(B, H, W, C, N);
    ConvBx2Args a;
    a.x = in; a.wp = (const char*)wp; a.out = out; a.aux = aux;
    a.B = B; a.H = H; a.W = W; a.C = C; a.N = N; a.relu = relu;
    a.col_tiles = p.col_tiles; a.ptiles = p.ptiles; a.ntiles = p.ntiles; a.grows = p.grows;
    a.m_hp1 = wino_magic(H + 1);
    a.m_coltiles = wino_magic(p.col_tiles); a.one_coltiles = p.col_tiles == 1 ? 0xffffffffu : 0u;
    a.m_ptiles = wino_magic(p.ptiles); a.one_ptiles = p.ptiles == 1 ? 0xffffffffu : 0u;
    static const int cus = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    static const int grid_env = getenv("VC_CONVBX2_GRID") ? atoi(getenv("VC_CONVBX2_GRID")) : 0;
    const int grid = p.ntiles < (grid_env > 0 ? grid_env : cus) ? p.ntiles : (grid_env > 0 ? grid_env : cus);
    if (p.lw == 32) {
        static int once = [] {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_bx2_kernel<KIND, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, C2_LDS);
            return e == hipSuccess ? 0 : fail((int)e, "%s: hipFuncSetAttribute failed", "conv_bx2 kernel");
        }();
        if (once) return once;
        hipLaunchKernelGGL((conv_bx2_kernel<KIND, 32>), dim3(grid), dim3(256), C2_LDS, st, a);
    } else {
        static int once = [] {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_bx2_kernel<KIND, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, C2_LDS);
            return e == hipSuccess ? 0 : fail((int)e, "%s: hipFuncSetAttribute failed", "conv_bx2 kernel");
        }();
        if (once) return once;
        hipLaunchKernelGGL((conv_bx2_kernel<KIND, 16>), dim3(grid), dim3(256), C2_LDS, st, a);
    }
    return launch_status("conv bx2");
}

}  // namespace vc

extern "C" int vc_conv3x3_bx2_supported(int B, int H, int W, int Cin, int Cout, int dgrad) {
    return vc::plan_conv_bx2(B, H, W, dgrad ? Cout : Cin, dgrad ? Cin : Cout).ok ? 1 : 0;
}

extern "C" size_t vc_conv3x3_bx2_pack_bytes(int Cin, int Cout) {
    if (Cin <= 0 || Cout <= 0 || Cin % 16 || Cout % 16) return 0;
    return (size_t)9 * Cin * Cout * 4;   // hi + lo bf16 per element: both orientations have the same size
}

extern "C" int vc_conv3x3_bx2_pack_f32(void* stream, int Cin, int Cout, const float* w, int transpose, void* wp) {
    using namespace vc;
    VC_CHECK_ARG(w && wp, "null pointer");
    VC_CHECK_ARG(Cin > 0 && Cout > 0 && (transpose ? (Cin % 64 == 0 && Cout % 16 == 0) : (Cout % 64 == 0 && Cin % 16 == 0)),
                 "produced channels must be a multiple of 64, contraction channels of 16");
    VC_CHECK_ARG(waligned16(wp), "wp must be 16-byte aligned");
    const int rows = transpose ? Cin : Cout, kdim = transpose ? Cout : Cin;
    const long total = (long)(rows >> 6) * (kdim >> 4) * 9 * 2 * 64;
    hipLaunchKernelGGL(conv_bx2_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, Cin, Cout, transpose, (char*)wp);
    return launch_status(__func__);
}

extern "C" int vc_conv3x3_bx2_fwd_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* x, const void* wp, const float* bias,
                                      float* y, int relu) {
    using namespace vc;
    VC_CHECK_ARG(plan_conv_bx2(B, H, W, Cin, Cout).ok, "unsupported shape (vc_conv3x3_bx2_supported)");
    VC_CHECK_ARG(x && wp && y, "null pointer");
    VC_CHECK_ARG(waligned16(x) && waligned16(wp) && waligned16(y) && (!bias || waligned16(bias)), "pointers must be 16-byte aligned");
    return launch_conv_bx2<0>((hipStream_t)stream, B, H, W, Cin, Cout, x, wp, bias, y, relu);
}

extern "C" int vc_conv3x3_bx2_dgrad_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* dy, const void* wpt, const float* relu_src,
                                        float* dx) {
    using namespace vc;
    VC_CHECK_ARG(plan_conv_bx2(B, H, W, Cout, Cin).ok, "unsupported shape (vc_conv3x3_bx2_supported)");
    VC_CHECK_ARG(dy && wpt && dx, "null pointer");
    VC_CHECK_ARG(waligned16(dy) && waligned16(wpt) && waligned16(dx) && (!relu_src || waligned16(relu_src)), "pointers must be 16-byte aligned");
    return launch_conv_bx2<1>((hipStream_t)stream, B, H, W, Cout, Cin, dy, wpt, relu_src, dx, 0);
}
