// PROTOTYPE (not on the trainer's path): Winograd F(4x4,3x3) forward / data gradient in f32 on the one-wave-per-SIMD recipe of
// conv_wgrad_bx.hip / conv_bx2.hip -- utils/image_embeddings.py:36-212, the same arithmetic class as conv_wino4.hip (f32 MFMAs, f32
// transforms).  Built to answer one question: does a lone wave per SIMD with register-resident accumulators for ALL 36 positions, its
// input transform in registers and every instruction at a compile-time place reach a higher matrix-pipe duty than conv_wino4's 46-54 %?
//
//   Y = A^T [ sum_ci (G g G^T) (.) (B^T d B) ] A          v_mfma_f32_16x16x4_f32: rows = output channels (A = U), columns = tiles (B = V)
//
// A wave = 16 tiles (4 x 4) x 32 output channels x 36 positions = 72 accumulators of four registers (288: the first 64 pinned in AGPRs,
// eight in VGPRs); the output transform is register math in one lane (no exchange between waves).  A lane = (tile, channel pair 2g, 2g+1
// of the eight-channel step): per channel it reads its raw 6 x 6 patch from the LDS image [channel][patch row][pixel] (a b128 + a b64
// per row; staging transposes the C4 pieces in registers), transforms it in registers (twelve 1-D transforms of fourteen operations) WHILE
// the previous channel's 72 MFMAs run, and reads the weights U[p] of both channel tiles and both channels as one b128 per position.
// Workgroup = four waves = an 8 x 8-tile block (32 x 32 pixels, patch 34 x 34 at pitch 36: conflict-free for a wave's 4 x 4 tiles) x 32
// output channels; two LDS images of 76 KB (patches 40 KB + weights 36 KB per eight channels); weights arrive by LDS-DMA.
// Two barriers per step of 144 MFMAs: the image of the next step is written during the first pass (behind B1: every wave is done with
// it) and read from the second pass on (behind B2).
#include <stdlib.h>
#include <type_traits>
#include <utility>
#include "conv_wino.h"

// R4_ABL (timing only; make r4abl): 1 no staging after the prologue (loads, LDS-DMA, LDS writes), 2 no transform arithmetic, 4 no weight
// reads after the prologue, 8 no patch reads, 16 one barrier per step instead of two
#ifndef R4_ABL
#define R4_ABL 0
#endif

namespace vc {

constexpr int R4_PP = 36, R4_CP = 1280;               // patch row pitch, channel plane pitch (floats; 34 x 36 = 1224 -> a multiple of 64)
constexpr int R4_XBYTES = 8 * R4_CP * 4;              // 40 960
constexpr int R4_WBYTES = 36 * 64 * 16;               // 36 864: [position][lane][channel tile 2][channel 2]
constexpr int R4_BUF = R4_XBYTES + R4_WBYTES;         // 77 824
constexpr int R4_LDS = 2 * R4_BUF;                    // 155 648

struct R4Args {
    const float* x;      // [B][C/4][H][W][4]
    const char* wp;      // [N / 32][C / 8][R4_WBYTES]
    float* y;            // [B][N/4][H][W][4]
    const float* aux;    // forward: bias or null; data gradient: ReLU source (layout of y) or null
    int B, H, W, C, N, relu;
    int bx_n, blocks_img, nblocks;
    unsigned m_bx_n, one_bx_n, m_blocks_img, one_blocks_img;
};

template <class F, int... I>
__device__ __forceinline__ void r4_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void r4_for(F&& f) { r4_for_impl(f, std::make_integer_sequence<int, N>{}); }

typedef float r4f4 __attribute__((ext_vector_type(4)));
#define R4SB() __builtin_amdgcn_sched_barrier(0)

template <int KIND>
__global__ __launch_bounds__(256, 1) void conv_wino4r_kernel(R4Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem4[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = lane & 15, g = lane >> 4, ty = t >> 2, tx = t & 3, wy = wave >> 1, wx = wave & 1;
    const int H = a.H, W = a.W, C = a.C, N = a.N, nk = C >> 3;
    auto bdiv = [&](unsigned n, unsigned m, unsigned one) -> unsigned { return (__umulhi(n, m) & ~one) | (n & one); };
    const unsigned nt = (unsigned)blockIdx.x / (unsigned)a.nblocks, blk = (unsigned)blockIdx.x - nt * (unsigned)a.nblocks;
    const unsigned b = bdiv(blk, a.m_blocks_img, a.one_blocks_img), rem = blk - b * (unsigned)a.blocks_img;
    const unsigned by = bdiv(rem, a.m_bx_n, a.one_bx_n), bx = rem - by * (unsigned)a.bx_n;
    const int y0 = (int)by * 32, x0 = (int)bx * 32;
    const unsigned plane_b = (unsigned)H * (unsigned)W * 16u;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)((long)a.B * H * W * C * 4), 0x00020000);

    // ---- staging: patch slot s = tid + 256 u (u < 3; 612 slots = 2 quads x 34 rows x 9 groups of four pixels): four 16-byte loads (the four
    // pixels, four channels each), written as four 16-byte pieces (the four channels, four pixels each); weights: nine LDS-DMA pieces
    unsigned pvoff[3][4];
    int plds[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int s = tid + 256 * u, q = s / 306, r2 = s - 306 * q, row = r2 / 9, grp = r2 - 9 * row;
        const int iy = y0 - 1 + row;
        plds[u] = ((4 * q) * R4_CP + row * R4_PP + 4 * grp) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ix = x0 - 1 + 4 * grp + j;
            const unsigned ok = 0u - (unsigned)((int)(s < 612) & (int)((unsigned)iy < (unsigned)H) & (int)((unsigned)ix < (unsigned)W) & (int)(4 * grp + j < 34));
            const unsigned addr = (unsigned)((((int)b * (C >> 2) + q) * H + iy) * W + ix) * 16u;
            pvoff[u][j] = (addr & ok) | (WOOB & ~ok);
        }
    }
    r4f4 st[3][4];
    auto pload = [&](int u, int j, int ks) {
        const float4 v = wbufload(rx, pvoff[u][j], (unsigned)(2 * ks) * plane_b);
        st[u][j] = r4f4{v.x, v.y, v.z, v.w};
    };
    auto pstore = [&](int img, int u, int c) {   // channel c of the slot's quad: pixels j = 0..3
        if (tid + 256 * u < 612)
            *reinterpret_cast<r4f4*>(smem4 + img + plds[u] + c * R4_CP * 4) = r4f4{st[u][0][c], st[u][1][c], st[u][2][c], st[u][3][c]};
    };
    const char* wbase = a.wp + (size_t)nt * nk * R4_WBYTES + (size_t)tid * 16;
    auto wdma = [&](int img, int u, int ks) {   // piece tid + 256 u of step ks -> LDS (wave-uniform base + lane * 16)
        const char* src = wbase + (size_t)ks * R4_WBYTES + (size_t)u * 4096;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(smem4 + img + R4_XBYTES + (wave * 64 + 256 * u) * 16), 16, 0, 0);
    };

    // ---- operands -------------------------------------------------------------------------------------------------------------
    const int pl = ((2 * g) * R4_CP + (4 * (4 * wy + ty)) * R4_PP + 4 * (4 * wx + tx)) * 4;   // byte offset of the lane's patch, channel 2 g; + e CP 4; + (r PP + c) 4
    const int ul = R4_XBYTES + lane * 16;                                                    // + p * 1024
    float Vc[36], T[36], raw[2][6];
    r4f4 U[4];
    auto uread = [&](int img, int p) { U[p & 3] = *reinterpret_cast<const r4f4*>(smem4 + img + ul + p * 1024); };
    auto rread = [&](int img, int e, int r, int half) {   // half 0: pixels 0..3 (b128), 1: pixels 4, 5 (b64) of patch row r -> raw[r & 1]
        const char* pp = smem4 + img + pl + e * R4_CP * 4 + r * R4_PP * 4;
        if (half == 0) {
            const r4f4 v = *reinterpret_cast<const r4f4*>(pp);
            raw[r & 1][0] = v[0]; raw[r & 1][1] = v[1]; raw[r & 1][2] = v[2]; raw[r & 1][3] = v[3];
        } else {
            const float2 v = *reinterpret_cast<const float2*>(pp + 16);
            raw[r & 1][4] = v.x; raw[r & 1][5] = v.y;
        }
    };
    // B^T d (fourteen operations in five parts): rows of B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
    float ts = 0.f, tu = 0.f, tv = 0.f, tw = 0.f, txx = 0.f, tyy = 0.f;
    auto bt1d = [&](const float (&d)[6], float (&o)[6], int part) {
        if (part == 0) { ts = d[1] + d[2]; tu = d[3] + d[4]; tv = d[1] - d[2]; }
        else if (part == 1) { tw = d[3] - d[4]; txx = d[3] - d[1]; tyy = d[4] - d[2]; }
        else if (part == 2) { o[0] = fmaf(4.f, d[0], fmaf(-5.f, d[2], d[4])); o[5] = fmaf(4.f, d[1], fmaf(-5.f, d[3], d[5])); }
        else if (part == 3) { o[1] = fmaf(-4.f, ts, tu); o[2] = fmaf(4.f, tv, -tw); }
        else { o[3] = fmaf(2.f, txx, tyy); o[4] = fmaf(-2.f, txx, tyy); }
    };
    auto rowtf = [&](int r, int part) {   // raw row r -> T[r][.] (horizontal transform)
        float o[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) o[c] = T[6 * r + c];
        bt1d(raw[r & 1], o, part);
#pragma unroll
        for (int c = 0; c < 6; ++c) T[6 * r + c] = o[c];
    };
    float cin[6];
    auto coltf = [&](int c, int part) {   // T[.][c] -> T[.][c] in place (vertical transform): the inputs are captured in part 0
        if (part == 0) {
#pragma unroll
            for (int r = 0; r < 6; ++r) cin[r] = T[6 * r + c];
        }
        float o[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) o[r] = T[6 * r + c];
        bt1d(cin, o, part);
#pragma unroll
        for (int r = 0; r < 6; ++r) T[6 * r + c] = o[r];
    };

    r4f4 acc[36][2];
#pragma unroll
    for (int p = 0; p < 36; ++p)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) acc[p][ct] = r4f4{0.f, 0.f, 0.f, 0.f};
    auto mfma = [&](r4f4& c, float av, float bv, bool in_agpr) {
        if (in_agpr) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(c) : "v"(av), "v"(bv));
        else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c) : "v"(av), "v"(bv));
    };

    // ---- prologue: step 0 into image 0; channel 2 g of it transformed; the first weights ------------------------------------------
    int cur = 0, nxt = R4_BUF;
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) pload(u, j, 0);
#pragma unroll
    for (int u = 0; u < 9; ++u) wdma(cur, u, 0);
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int c = 0; c < 4; ++c) pstore(cur, u, c);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        rread(cur, 0, r, 0); rread(cur, 0, r, 1);
#pragma unroll
        for (int part = 0; part < 5; ++part) rowtf(r, part);
    }
#pragma unroll
    for (int c = 0; c < 6; ++c)
#pragma unroll
        for (int part = 0; part < 5; ++part) coltf(c, part);
#pragma unroll
    for (int p = 0; p < 36; ++p) Vc[p] = T[p];
#pragma unroll
    for (int p = 0; p < 3; ++p) uread(cur, p);
    R4SB();

    // ---- one pass = the 72 MFMAs of one channel (position p = m / 2, channel tile m % 2).  Behind them: the weights three positions
    // ahead (a ring of four b128), the NEXT channel's patch rows (row r read at 5 r, 5 r + 1 into raw[r & 1]; its horizontal transform in
    // five parts from 5 r + 4; the six vertical transforms from 36) and, in the first pass of a step, the staging of the next step.
    auto pass = [&](auto ec, int ksn) {
        constexpr int e = decltype(ec)::value;
        const int rimg = e == 0 ? cur : nxt;      // the next channel: 2 g + 1 of this step's image, or 2 g of the next step's
        r4_for<72>([&](auto mc) {
            constexpr int m = decltype(mc)::value, p = m >> 1, ct = m & 1;
            mfma(acc[p][ct], U[p & 3][ct * 2 + e], Vc[p], p * 2 + ct < 64);
            R4SB();
            if constexpr (!(R4_ABL & 4)) {
                if constexpr (ct == 1 && p + 3 < 36) uread(cur, p + 3);
                if constexpr (m == 67 || m == 69 || m == 71) uread(rimg, (m - 67) >> 1);   // positions 0..2 of the next pass
            }
            r4_for<6>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                if constexpr (!(R4_ABL & 8)) {
                    if constexpr (m == 5 * r) rread(rimg, 1 - e, r, 0);
                    if constexpr (m == 5 * r + 1) rread(rimg, 1 - e, r, 1);
                }
                if constexpr (!(R4_ABL & 2)) {
                    if constexpr (m >= 5 * r + 4 && m < 5 * r + 9) rowtf(r, m - (5 * r + 4));
                    if constexpr (m >= 36 + 5 * r && m < 41 + 5 * r) coltf(r, m - (36 + 5 * r));
                }
            });
            if constexpr (e == 0 && !(R4_ABL & 1)) {   // staging of step ksn into the other image (free since B1, published by B2)
                if constexpr (m >= 2 && m < 14) pload((m - 2) >> 2, (m - 2) & 3, ksn);
                if constexpr (m >= 14 && m < 23) wdma(nxt, m - 14, ksn);
                if constexpr (m >= 52 && m < 64) pstore(nxt, (m - 52) >> 2, (m - 52) & 3);
            }
            R4SB();
        });
#pragma unroll
        for (int p = 0; p < 36; ++p) Vc[p] = T[p];
    };
    for (int ks = 0; ks < nk; ++ks) {
        if (!(R4_ABL & 16)) __syncthreads();                  // B1: every wave is done with the other image
        pass(std::integral_constant<int, 0>{}, ks + 1 < nk ? ks + 1 : ks);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the LDS-DMA pieces have landed)
        __syncthreads();                                      // B2: the other image is complete
        pass(std::integral_constant<int, 1>{}, 0);
        const int sw = cur; cur = nxt; nxt = sw;
    }
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");

    // ---- output transform A^T M A per (channel tile, accumulator register = output channel), then bias / ReLU (or the ReLU mask) and
    // 16-byte stores: a register quad is four consecutive channels of one pixel.  A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
    auto at1d = [&](const float (&mm)[6], float (&o)[4]) {
        const float pp = mm[1] + mm[2], qq = mm[1] - mm[2], rr = mm[3] + mm[4], ss = mm[3] - mm[4];
        o[0] = mm[0] + pp + rr;
        o[1] = fmaf(2.f, ss, qq);
        o[2] = fmaf(4.f, rr, pp);
        o[3] = fmaf(8.f, ss, qq) + mm[5];
    };
    const int oy0 = y0 + 4 * (4 * wy + ty), ox0 = x0 + 4 * (4 * wx + tx);
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        float Y[4][16];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float tm[4][6];   // vertical: A^T over r' for every c'
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                float col[6], o[4];
#pragma unroll
                for (int rp = 0; rp < 6; ++rp) col[rp] = acc[6 * rp + c][ct][r];
                at1d(col, o);
#pragma unroll
                for (int i = 0; i < 4; ++i) tm[i][c] = o[i];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float o[4];
                at1d(tm[i], o);
#pragma unroll
                for (int j = 0; j < 4; ++j) Y[r][4 * i + j] = o[j];
            }
        }
        const int co = (int)nt * 32 + 16 * ct + 4 * g;
        float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
        if (KIND == 0 && a.aux) bb = *reinterpret_cast<const float4*>(a.aux + co);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int oy = oy0 + i, ox = ox0 + j;
                if (oy >= H || ox >= W) continue;
                const size_t eo = ((((size_t)b * (size_t)(N >> 2) + (size_t)(co >> 2)) * H + oy) * W + ox) * 4;
                float4 v = make_float4(Y[0][4 * i + j] + bb.x, Y[1][4 * i + j] + bb.y, Y[2][4 * i + j] + bb.z, Y[3][4 * i + j] + bb.w);
                if (KIND == 0) {
                    if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                } else if (a.aux) {
                    const float4 sv = *reinterpret_cast<const float4*>(a.aux + eo);
                    v.x = sv.x > 0.f ? v.x : 0.f; v.y = sv.y > 0.f ? v.y : 0.f; v.z = sv.z > 0.f ? v.z : 0.f; v.w = sv.w > 0.f ? v.w : 0.f;
                }
                *reinterpret_cast<float4*>(a.y + eo) = v;
            }
    }
}
#undef R4SB

// w [3,3,Cin,Cout] HWIO -> [N / 32][C / 8][position 6 r' + c'][lane][channel tile][channel]: U = G g G^T of (contraction channel
// 8 ks + 2 (lane / 16) + e, produced channel 32 nt + 16 ct + lane % 16); transpose: C = Cout, N = Cin, taps flipped (data gradient)
__global__ __launch_bounds__(256) void conv_wino4r_pack_kernel(const float* __restrict__ w, int Ci, int Co, int transpose, float* __restrict__ out) {
    const int C = transpose ? Co : Ci, N = transpose ? Ci : Co, nk = C >> 3;
    const long total = (long)C * N;
    const float G[6][3] = {{0.25f, 0.f, 0.f}, {-1.f / 6, -1.f / 6, -1.f / 6}, {-1.f / 6, 1.f / 6, -1.f / 6},
                           {1.f / 24, 1.f / 12, 1.f / 6}, {1.f / 24, -1.f / 12, 1.f / 6}, {0.f, 0.f, 1.f}};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i / N), n = (int)(i % N);
        float gk[3][3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
                gk[ky][kx] = transpose ? w[((long)((2 - ky) * 3 + (2 - kx)) * Ci + n) * Co + c] : w[((long)(ky * 3 + kx) * Ci + c) * Co + n];
        float tg[6][3];   // G g
#pragma unroll
        for (int u = 0; u < 6; ++u)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) tg[u][kx] = G[u][0] * gk[0][kx] + G[u][1] * gk[1][kx] + G[u][2] * gk[2][kx];
        const int ntile = n >> 5, ct = (n >> 4) & 1, ln = (n & 15) + 16 * ((c & 7) >> 1), ks = c >> 3, e = c & 1;
        float* o = out + ((long)ntile * nk + ks) * (R4_WBYTES / 4) + ln * 4 + ct * 2 + e;
#pragma unroll
        for (int u = 0; u < 6; ++u)
#pragma unroll
            for (int v = 0; v < 6; ++v) o[(6 * u + v) * 256] = tg[u][0] * G[v][0] + tg[u][1] * G[v][1] + tg[u][2] * G[v][2];
    }
}

static bool plan_wino4r(int B, int H, int W, int C, int N) {
    if (B <= 0 || H < 1 || W < 1 || C <= 0 || N <= 0 || C % 8 || N % 32) return false;
    if ((long)B * H * W * (long)(C > N ? C : N) * 4 > 0x7fffffffL) return false;
    if ((long)B * cdiv(H, 32) * cdiv(W, 32) * (N / 32) > 0x7fffffffL) return false;
    return true;
}

template <int KIND>
static int launch_wino4r(hipStream_t st, int B, int H, int W, int C, int N, const float* in, const void* wp, const float* aux, float* out, int relu) {
    R4Args a;
    a.x = in; a.wp = (const char*)wp; a.y = out; a.aux = aux;
    a.B = B; a.H = H; a.W = W; a.C = C; a.N = N; a.relu = relu;
    a.bx_n = cdiv(W, 32); a.blocks_img = a.bx_n * cdiv(H, 32); a.nblocks = B * a.blocks_img;
    a.m_bx_n = wino_magic(a.bx_n); a.one_bx_n = a.bx_n == 1 ? 0xffffffffu : 0u;
    a.m_blocks_img = wino_magic(a.blocks_img); a.one_blocks_img = a.blocks_img == 1 ? 0xffffffffu : 0u;
    static int once = [] {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino4r_kernel<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, R4_LDS);
        return e == hipSuccess ? 0 : fail((int)e, "%s: hipFuncSetAttribute failed", "conv_wino4r kernel");
    }();
    if (once) return once;
    hipLaunchKernelGGL((conv_wino4r_kernel<KIND>), dim3((unsigned)((N / 32) * a.nblocks)), dim3(256), R4_LDS, st, a);
    return launch_status("conv wino4r");
}

}  // namespace vc

extern "C" int vc_conv3x3_wino4r_supported(int B, int H, int W, int Cin, int Cout, int dgrad) {
    return vc::plan_wino4r(B, H, W, dgrad ? Cout : Cin, dgrad ? Cin : Cout) ? 1 : 0;
}

extern "C" size_t vc_conv3x3_wino4r_pack_bytes(int Cin, int Cout) {
    if (Cin <= 0 || Cout <= 0 || Cin % 8 || Cout % 8) return 0;
    return (size_t)36 * Cin * Cout * 4;
}

extern "C" int vc_conv3x3_wino4r_pack_f32(void* stream, int Cin, int Cout, const float* w, int transpose, float* wp) {
    using namespace vc;
    VC_CHECK_ARG(w && wp, "null pointer");
    VC_CHECK_ARG(Cin > 0 && Cout > 0 && (transpose ? (Cin % 32 == 0 && Cout % 8 == 0) : (Cout % 32 == 0 && Cin % 8 == 0)),
                 "produced channels must be a multiple of 32, contraction channels of 8");
    VC_CHECK_ARG(waligned16(wp), "wp must be 16-byte aligned");
    const long total = (long)Cin * Cout;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(conv_wino4r_pack_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, Cin, Cout, transpose, wp);
    return launch_status(__func__);
}

extern "C" int vc_conv3x3_wino4r_fwd_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* x, const float* wp, const float* bias,
                                         float* y, int relu) {
    using namespace vc;
    VC_CHECK_ARG(plan_wino4r(B, H, W, Cin, Cout), "unsupported shape (vc_conv3x3_wino4r_supported)");
    VC_CHECK_ARG(x && wp && y, "null pointer");
    VC_CHECK_ARG(waligned16(x) && waligned16(wp) && waligned16(y) && (!bias || waligned16(bias)), "pointers must be 16-byte aligned");
    return launch_wino4r<0>((hipStream_t)stream, B, H, W, Cin, Cout, x, wp, bias, y, relu);
}

extern "C" int vc_conv3x3_wino4r_dgrad_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* dy, const float* wpt, const float* relu_src,
                                           float* dx) {
    using namespace vc;
    VC_CHECK_ARG(plan_wino4r(B, H, W, Cout, Cin), "unsupported shape (vc_conv3x3_wino4r_supported)");
    VC_CHECK_ARG(dy && wpt && dx, "null pointer");
    VC_CHECK_ARG(waligned16(dy) && waligned16(wpt) && waligned16(dx) && (!relu_src || waligned16(relu_src)), "pointers must be 16-byte aligned");
    return launch_wino4r<1>((hipStream_t)stream, B, H, W, Cout, Cin, dy, wpt, relu_src, dx, 0);
}
