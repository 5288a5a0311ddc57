// 3x3 / stride 1 / SAME convolution by Winograd's minimal filtering F(2x2, 3x3) for gfx950: forward and data gradient of
// utils/image_embeddings.py:36-212 in fp32 with 2.25x fewer multiplications than the direct form (conv_patch.hip).
//
//   Y = A^T [ sum_c (G g_c G^T) (.) (B^T d_c B) ] A          per 2 x 2 output tile: d = its 4 x 4 input patch, g = the 3 x 3 filter
//
// The sixteen positions of the transformed domain are sixteen independent products M_p[n][tile] = sum_c V_p[c][n] U_p[c][tile];
// everything between the NHWC input and the NHWC output stays inside the kernel:
//   * a wave owns a BLOCK of up to 32 tiles (TBH x TBW tiles = 2 TBH x 2 TBW output pixels of one image) and all sixteen positions
//     for 32 output columns: sixteen 32 x 32 MFMA accumulators = 256 registers (AGPRs), one wave per SIMD;
//   * the workgroup (four waves = four blocks x the same 32 columns) stages, per 16-channel chunk, the halo patches of its blocks
//     and the chunk's transformed weights (32 KB, contiguous in the packed layout) in LDS;
//   * the input transform B^T d B runs in registers on the MFMA's B-operand layout (lane = tile, lane half = 8 of the 16 channels):
//     eight ds_read_b128 + 32 float4 additions per sixteen MFMAs, prepared one position row ahead of the MFMAs that consume it;
//   * the weights are transformed once per optimiser step (vc_conv3x3_wino_pack_f32; transpose 1 = flipped taps, transposed
//     channels for the data gradient);
//   * a lane ends with all sixteen positions of ONE tile x 16 output columns, so the output transform A^T M A, the bias, the ReLU /
//     ReLU mask, the 16-byte stores and the fused 2 x 2 max-pool (a pooling window IS a Winograd tile) are register math.
// Rounding: the transforms add at most four fp32 terms with coefficients 1, 1/2, 1/4; results agree with the direct form to a few
// 1e-7 of the tensor maximum times sqrt(K) (tests/test_gpu_conv_wino.py holds both to the same fp64 oracle).
#include <stdlib.h>
#include "gemm_core.h"
#include "vaecap.h"

namespace vc {

typedef unsigned int wu32x4 __attribute__((ext_vector_type(4)));

enum { WK_FWD = 0, WK_DGRAD = 1 };
constexpr int WCH = 16;                          // channels per chunk
constexpr int WPITCH = WCH + 4;                  // floats per patch pixel in LDS
constexpr int WPIX = 180;                        // patch pixels per block: (2 TBH + 2)(2 TBW + 2) <= 180
constexpr int WBLK = WPIX * WPITCH;              // floats per block patch
constexpr int WSLOTS = 12;                       // float4 patch slots per thread: 4 blocks x 180 pixels x 4 channel quads <= 256 x 12
constexpr int WV_FLOATS = 16 * 4 * 32 * 4;       // one chunk of transformed weights: [p 16][quad 4][n 32][e 4]
constexpr int WP_OFF = WV_FLOATS;                // LDS: weights first (their ds_read offsets stay below the 64 KB immediate range), then the patches
constexpr int WINO_LDS_BYTES = (WP_OFF + 4 * WBLK) * 4;
constexpr unsigned WOOB = 0x80000000u;

struct WinoGeom {
    int B, H, W, C, N;
    int TBH, TBW;          // tiles per block (rows, columns); TBH * TBW <= 32
    int PW, PH;            // halo patch of a block in pixels: 2 TBW + 2, 2 TBH + 2
    int bx_n, by_n;        // blocks per image row / column
    int blocks_img;
    int nblocks;           // B * blocks_img
};

struct WinoArgs {
    WinoGeom g;
    const float* x;     // [P, C]
    const float* wp;    // packed [N/32][C/16][16][4][32][4]
    float* out;         // [P, N]
    const float* aux;   // fwd: bias [N] or null; dgrad: ReLU source [P, N] or null
    float* pool;        // fwd: also max_pool2x2(out) [B, H/2, W/2, N] (null: none)
    int relu;
    int tiles_n, ntiles, nchunks;
    int dbg;            // experiments (VC_WINO_DBG): 1 = no global loads in the loop, 2 = no LDS restaging, 4 = no MFMAs
};

__device__ __forceinline__ float4 wbufload(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    wu32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
    return *reinterpret_cast<float4*>(&v);
}
__device__ __forceinline__ float wcomp(const float4& v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; }
__device__ __forceinline__ float4 f4add(const float4& a, const float4& b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4sub(const float4& a, const float4& b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

template <int KIND, bool POOL>
__global__ __launch_bounds__(256, 1) void conv_wino_kernel(WinoArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const WinoGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int id = xcd_remap(blockIdx.x, a.ntiles);
    const int tm = id / a.tiles_n, nt = id - tm * a.tiles_n, n0 = nt * 32;
    const int C = g.C, N = g.N;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)((long)g.B * g.H * g.W * C * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.wp, 0, 16 * C * N * 4, 0x00020000);

    // patch slots of this thread: slot s = tid + 256 i = (block s / 720, patch pixel (s % 720) / 4, channel quad s % 4)
    unsigned voff[WSLOTS];
#pragma unroll
    for (int i = 0; i < WSLOTS; ++i) {
        const int s = tid + 256 * i;
        const int blk = s / (4 * WPIX), r = s - blk * (4 * WPIX), pix = r >> 2, quad = r & 3;
        const int gb = tm * 4 + blk;
        voff[i] = WOOB;
        if (blk < 4 && gb < g.nblocks && pix < g.PH * g.PW) {
            const int b = gb / g.blocks_img, rem = gb - b * g.blocks_img;
            const int by = rem / g.bx_n, bx = rem - by * g.bx_n;
            const int py = pix / g.PW, px = pix - py * g.PW;
            const int y = by * 2 * g.TBH - 1 + py, x = bx * 2 * g.TBW - 1 + px;
            if ((unsigned)y < (unsigned)g.H && (unsigned)x < (unsigned)g.W)
                voff[i] = (unsigned)((((long)(b * g.H + y) * g.W + x) * C + quad * 4) * 4);
        }
    }
    const int pst = WP_OFF + (tid >> 2) * WPITCH + (tid & 3) * 4;   // LDS float index of slot 0; slot i is 64 pixels = 64 * WPITCH floats further
    const unsigned vsrc = (unsigned)(((long)nt * a.nchunks) * WV_FLOATS * 4) + (unsigned)tid * 16u;  // weights: chunk c at + c * 32 KB, piece i at + i * 4 KB

    // this lane's tile inside its wave's block, its 4 x 4 patch origin in LDS, its weight fragment origin
    const int ntl = g.TBH * g.TBW;
    const int jt = li < ntl ? li : 0;
    const int tyl = jt / g.TBW, txl = jt - tyl * g.TBW;
    const int abase = WP_OFF + wave * WBLK + ((2 * tyl) * g.PW + 2 * txl) * WPITCH + lh * 8;   // + q * 4 + (i * PW + j) * WPITCH
    const int rowp = g.PW * WPITCH;
    const int vbase = (2 * lh * 32 + li) * 4;                               // + (p * 4 + q) * 128

    f32x16 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

    float4 pr[WSLOTS], vr[8];
    auto gload = [&](int ch) {
#pragma unroll
        for (int i = 0; i < WSLOTS; ++i) pr[i] = wbufload(rx, voff[i], (unsigned)ch * (WCH * 4));
#pragma unroll
        for (int i = 0; i < 8; ++i) vr[i] = wbufload(rw, vsrc + (unsigned)i * 4096u, (unsigned)ch * (WV_FLOATS * 4));
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < WSLOTS; ++i)
            if (i < WSLOTS - 1 || tid + 256 * i < 4 * 4 * WPIX) *reinterpret_cast<float4*>(&smem[pst + i * 64 * WPITCH]) = pr[i];
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<float4*>(&smem[(tid + 256 * i) * 4]) = vr[i];
    };

    // unit (q, xi): position row xi of channel quad q.  U[xi][nu] = (B^T d B)[xi][nu], B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1].
    // The software pipeline is written out and pinned with sched_barriers (hipcc otherwise sinks the LDS reads next to their first
    // use and batches the MFMAs behind the additions): while the sixteen MFMAs of unit u run, the twelve ds_read_b128 of unit u + 1
    // are issued first and its 32 additions follow one quad per MFMA.
    // Units run in the order xi = 0, 2, 1, 3 so that every patch row is read once per channel quad: rows 0 and 2 for xi = 0, row 1 for
    // xi = 2, nothing for xi = 1, row 3 for xi = 3 (sixteen patch reads + sixteen weight reads per 64 MFMAs).
    float4 ur[2][4], vf[2][4];
    float4 dr[4][4], tt[4];
    auto rd = [&](int u, int buf) {   // the patch rows unit u is the first to need (four pixels each) + its four weight fragments
        const int q = u >> 2, xi = (u & 3) == 1 ? 2 : (u & 3) == 2 ? 1 : (u & 3);
        const float* pq = &smem[abase + q * 4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (xi == 0) { dr[0][j] = *reinterpret_cast<const float4*>(pq + j * WPITCH); dr[2][j] = *reinterpret_cast<const float4*>(pq + 2 * rowp + j * WPITCH); }
            if (xi == 2) dr[1][j] = *reinterpret_cast<const float4*>(pq + rowp + j * WPITCH);
            if (xi == 3) dr[3][j] = *reinterpret_cast<const float4*>(pq + 3 * rowp + j * WPITCH);
        }
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) vf[buf][nu] = *reinterpret_cast<const float4*>(&smem[vbase + ((xi * 4 + nu) * 4 + q) * 128]);
    };
    auto tstep = [&](int u, int buf, int k) {   // eight steps of four additions
        const int xi = (u & 3) == 1 ? 2 : (u & 3) == 2 ? 1 : (u & 3);
        if (k < 4) tt[k] = xi == 0 ? f4sub(dr[0][k], dr[2][k]) : xi == 1 ? f4add(dr[1][k], dr[2][k]) : xi == 2 ? f4sub(dr[2][k], dr[1][k]) : f4sub(dr[1][k], dr[3][k]);
        if (k == 4) ur[buf][0] = f4sub(tt[0], tt[2]);
        if (k == 5) ur[buf][1] = f4add(tt[1], tt[2]);
        if (k == 6) ur[buf][2] = f4sub(tt[2], tt[1]);
        if (k == 7) ur[buf][3] = f4sub(tt[1], tt[3]);
    };
    auto mf = [&](int u, int buf, int m) {
        const int xi = (u & 3) == 1 ? 2 : (u & 3) == 2 ? 1 : (u & 3), e = m >> 2, nu = m & 3;
        acc[xi * 4 + nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcomp(vf[buf][nu], e), wcomp(ur[buf][nu], e), acc[xi * 4 + nu], 0, 0, 0);
    };
#define WSB() __builtin_amdgcn_sched_barrier(0)

    gload(0);
    lstore();
    __syncthreads();
    for (int ch = 0; ch < a.nchunks; ++ch) {
        const bool more = ch + 1 < a.nchunks;
        if (more && !(a.dbg & 1)) gload(ch + 1);
        rd(0, 0);
#pragma unroll
        for (int k = 0; k < 8; ++k) tstep(0, 0, k);
        WSB();
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int b = u & 1, nb = b ^ 1;
            if (u + 1 < 8) {
                rd(u + 1, nb);
                WSB();
                mf(u, b, 0); mf(u, b, 1);
                WSB();
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    tstep(u + 1, nb, k);
                    WSB();
                    mf(u, b, 2 + k);
                    WSB();
                }
#pragma unroll
                for (int m = 10; m < 16; ++m) mf(u, b, m);
            } else {
#pragma unroll
                for (int m = 0; m < 16; ++m) mf(u, b, m);
            }
            WSB();
        }
        if (more && !(a.dbg & 2)) {
            __syncthreads();
            lstore();
            __syncthreads();
        }
    }
#undef WSB
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");

    // ---- output transform + epilogue: acc[p][r] = M_p[column n0 + 8 (r >> 2) + 4 lh + (r & 3)][tile li]
    const int gb = tm * 4 + wave;
    const bool blk_ok = gb < g.nblocks && li < ntl;
    const int gbc = gb < g.nblocks ? gb : 0;
    const int b = gbc / g.blocks_img, rem = gbc - b * g.blocks_img;
    const int by = rem / g.bx_n, bx = rem - by * g.bx_n;
    const int y0 = (by * g.TBH + tyl) * 2, x0 = (bx * g.TBW + txl) * 2;
    const bool ok00 = blk_ok && y0 < g.H && x0 < g.W, ok01 = ok00 && x0 + 1 < g.W, ok10 = ok00 && y0 + 1 < g.H, ok11 = ok10 && x0 + 1 < g.W;
    const long p00 = ((long)(b * g.H + y0) * g.W + x0) * N;
    const long rowN = (long)g.W * N;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
        const int col = n0 + 8 * rg + 4 * lh;
        float4 Y[2][2];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float s[2][4];
#pragma unroll
            for (int nu = 0; nu < 4; ++nu) {
                const float m0 = acc[nu][4 * rg + k], m1 = acc[4 + nu][4 * rg + k], m2 = acc[8 + nu][4 * rg + k], m3 = acc[12 + nu][4 * rg + k];
                s[0][nu] = m0 + m1 + m2;
                s[1][nu] = m1 - m2 - m3;
            }
#pragma unroll
            for (int aa = 0; aa < 2; ++aa) {
                const float y0v = s[aa][0] + s[aa][1] + s[aa][2], y1v = s[aa][1] - s[aa][2] - s[aa][3];
                if (k == 0) { Y[aa][0].x = y0v; Y[aa][1].x = y1v; }
                if (k == 1) { Y[aa][0].y = y0v; Y[aa][1].y = y1v; }
                if (k == 2) { Y[aa][0].z = y0v; Y[aa][1].z = y1v; }
                if (k == 3) { Y[aa][0].w = y0v; Y[aa][1].w = y1v; }
            }
        }
        if (KIND == WK_FWD) {
            if (a.aux) {
                const float4 bv = *reinterpret_cast<const float4*>(a.aux + col);
#pragma unroll
                for (int aa = 0; aa < 2; ++aa)
#pragma unroll
                    for (int bb = 0; bb < 2; ++bb) Y[aa][bb] = f4add(Y[aa][bb], bv);
            }
            if (a.relu) {
#pragma unroll
                for (int aa = 0; aa < 2; ++aa)
#pragma unroll
                    for (int bb = 0; bb < 2; ++bb) {
                        float4& v = Y[aa][bb];
                        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                    }
            }
        } else if (a.aux) {
#pragma unroll
            for (int aa = 0; aa < 2; ++aa)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) {
                    const bool ok = aa == 0 ? (bb == 0 ? ok00 : ok01) : (bb == 0 ? ok10 : ok11);
                    const float4 m = ok ? *reinterpret_cast<const float4*>(a.aux + p00 + aa * rowN + bb * N + col) : f4zero();
                    float4& v = Y[aa][bb];
                    if (!(m.x > 0.f)) v.x = 0.f;
                    if (!(m.y > 0.f)) v.y = 0.f;
                    if (!(m.z > 0.f)) v.z = 0.f;
                    if (!(m.w > 0.f)) v.w = 0.f;
                }
        }
        if (ok00) *reinterpret_cast<float4*>(a.out + p00 + col) = Y[0][0];
        if (ok01) *reinterpret_cast<float4*>(a.out + p00 + N + col) = Y[0][1];
        if (ok10) *reinterpret_cast<float4*>(a.out + p00 + rowN + col) = Y[1][0];
        if (ok11) *reinterpret_cast<float4*>(a.out + p00 + rowN + N + col) = Y[1][1];
        if (POOL && ok11) {
            float4 m;
            m.x = fmaxf(fmaxf(Y[0][0].x, Y[0][1].x), fmaxf(Y[1][0].x, Y[1][1].x));
            m.y = fmaxf(fmaxf(Y[0][0].y, Y[0][1].y), fmaxf(Y[1][0].y, Y[1][1].y));
            m.z = fmaxf(fmaxf(Y[0][0].z, Y[0][1].z), fmaxf(Y[1][0].z, Y[1][1].z));
            m.w = fmaxf(fmaxf(Y[0][0].w, Y[0][1].w), fmaxf(Y[1][0].w, Y[1][1].w));
            *reinterpret_cast<float4*>(a.pool + ((long)(b * (g.H >> 1) + (y0 >> 1)) * (g.W >> 1) + (x0 >> 1)) * N + col) = m;
        }
    }
}

// w [3][3][Ci][Co] (HWIO) -> V = G g G^T, G = [1 0 0; 1/2 1/2 1/2; 1/2 -1/2 1/2; 0 0 1], packed [N/32][C/16][p 16][quad 4][n 32][e 4]:
//   transpose 0 (forward):        C = Ci, N = Co, g[ky][kx] = w[ky][kx][c][n]
//   transpose 1 (data gradient):  C = Co, N = Ci, g[ky][kx] = w[2 - ky][2 - kx][n][c]
__global__ __launch_bounds__(256) void wino_pack_kernel(const float* __restrict__ w, int Ci, int Co, int transpose, float* __restrict__ out) {
    const int C = transpose ? Co : Ci, N = transpose ? Ci : Co;
    const long total = (long)C * N;
    const int nchunks = C / WCH;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        // consecutive threads: consecutive n of one c when not transposed (w rows are [c][n]), consecutive c of one n when transposed
        const int c = transpose ? (int)(i % C) : (int)(i / N), n = transpose ? (int)(i / C) : (int)(i % N);
        float gk[3][3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
                gk[ky][kx] = transpose ? w[((long)((2 - ky) * 3 + (2 - kx)) * Ci + n) * Co + c] : w[((long)(ky * 3 + kx) * Ci + c) * Co + n];
        float t[4][3];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            t[0][kx] = gk[0][kx];
            t[1][kx] = 0.5f * (gk[0][kx] + gk[1][kx] + gk[2][kx]);
            t[2][kx] = 0.5f * (gk[0][kx] - gk[1][kx] + gk[2][kx]);
            t[3][kx] = gk[2][kx];
        }
        const int nt = n >> 5, nl = n & 31, ch = c / WCH, quad = (c % WCH) >> 2, e = c & 3;
        float* o = out + ((long)nt * nchunks + ch) * WV_FLOATS + (quad * 32 + nl) * 4 + e;
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) {
            const float v0 = t[xi][0], v1 = 0.5f * (t[xi][0] + t[xi][1] + t[xi][2]), v2 = 0.5f * (t[xi][0] - t[xi][1] + t[xi][2]), v3 = t[xi][2];
            o[(xi * 4 + 0) * 512] = v0;
            o[(xi * 4 + 1) * 512] = v1;
            o[(xi * 4 + 2) * 512] = v2;
            o[(xi * 4 + 3) * 512] = v3;
        }
    }
}

// ---- host side ----------------------------------------------------------------------------------
// Block shapes (tiles): wide images take 4 x 8, the 56-wide layers 4 x 7, the 28-wide 2 x 14, the 14-wide 4 x 7 (two blocks per image,
// the second one half empty); in general the widest TBW <= 16 that divides the tile columns, with TBH = 32 / TBW (at most 8).
static bool plan_wino(int B, int H, int W, int C, int N, WinoGeom& g) {
    g.B = B; g.H = H; g.W = W; g.C = C; g.N = N;
    if (B <= 0 || H < 2 || W < 2 || (H & 1) || (W & 1) || C <= 0 || N <= 0 || C % WCH || N % 32) return false;
    if ((long)B * H * W * (long)(C > N ? C : N) * 4 > 0x7fffffffL || 16L * C * N * 4 > 0x7fffffffL) return false;
    const int TW = W / 2, TH = H / 2;
    int best = 0;
    double best_eff = 0.0;
    for (int tbw = 1; tbw <= 16 && tbw <= TW; ++tbw) {
        int tbh = 32 / tbw;
        if (tbh > 8) tbh = 8;
        if (tbh > TH) tbh = TH;
        if ((2 * tbh + 2) * (2 * tbw + 2) > WPIX) continue;
        const long slots = (long)cdiv(TW, tbw) * cdiv(TH, tbh) * 32;
        const double eff = (double)TW * TH / (double)slots;
        if (eff > best_eff + 1e-9) { best_eff = eff; best = tbw; }
    }
    if (!best) return false;
    g.TBW = best;
    g.TBH = 32 / best > 8 ? 8 : 32 / best;
    if (g.TBH > TH) g.TBH = TH;
    g.PW = 2 * g.TBW + 2; g.PH = 2 * g.TBH + 2;
    g.bx_n = cdiv(TW, g.TBW); g.by_n = cdiv(TH, g.TBH);
    g.blocks_img = g.bx_n * g.by_n;
    if ((long)B * g.blocks_img > 0x3fffffffL) return false;
    g.nblocks = B * g.blocks_img;
    return true;
}

static int wino_attr() {
    static int once = [] {
        hipError_t e = hipSuccess;
        auto set = [&](const void* f) { if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, WINO_LDS_BYTES); };
        set(reinterpret_cast<const void*>(conv_wino_kernel<WK_FWD, false>));
        set(reinterpret_cast<const void*>(conv_wino_kernel<WK_FWD, true>));
        set(reinterpret_cast<const void*>(conv_wino_kernel<WK_DGRAD, false>));
        return e == hipSuccess ? 0 : fail((int)e, "%s: hipFuncSetAttribute failed", "conv wino kernel");
    }();
    return once;
}

template <int KIND, bool POOL>
static int launch_wino(hipStream_t st, WinoArgs& a) {
    int rc = wino_attr();
    if (rc) return rc;
    a.tiles_n = a.g.N / 32;
    a.nchunks = a.g.C / WCH;
    a.ntiles = cdiv(a.g.nblocks, 4) * a.tiles_n;
    static const int dbg = getenv("VC_WINO_DBG") ? atoi(getenv("VC_WINO_DBG")) : 0;
    a.dbg = dbg;
    hipLaunchKernelGGL((conv_wino_kernel<KIND, POOL>), dim3(a.ntiles), dim3(256), WINO_LDS_BYTES, st, a);
    return launch_status("conv wino");
}

static bool waligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace vc

extern "C" int vc_conv3x3_wino_supported(int B, int H, int W, int Cin, int Cout, int dgrad) {
    vc::WinoGeom g;
    return (dgrad ? vc::plan_wino(B, H, W, Cout, Cin, g) : vc::plan_wino(B, H, W, Cin, Cout, g)) ? 1 : 0;
}

extern "C" int vc_conv3x3_wino_pack_f32(void* stream, int Cin, int Cout, const float* w, int transpose, float* wp) {
    using namespace vc;
    const int C = transpose ? Cout : Cin, N = transpose ? Cin : Cout;
    VC_CHECK_ARG(C > 0 && N > 0 && C % WCH == 0 && N % 32 == 0, "gathered channels % 16 == 0 and output channels % 32 == 0 required");
    VC_CHECK_ARG(w && wp && waligned16(wp), "null or misaligned pointer");
    const long total = (long)C * N;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(wino_pack_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, Cin, Cout, transpose, wp);
    return launch_status(__func__);
}

extern "C" int vc_conv3x3_wino_fwd_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* x, const float* wp,
                                       const float* bias, float* y, float* ypool, int relu) {
    using namespace vc;
    WinoArgs a;
    VC_CHECK_ARG(plan_wino(B, H, W, Cin, Cout, a.g), "unsupported shape (vc_conv3x3_wino_supported)");
    VC_CHECK_ARG(x && wp && y, "null pointer");
    VC_CHECK_ARG(waligned16(x) && waligned16(wp) && waligned16(y) && waligned16(bias) && waligned16(ypool), "pointers must be 16-byte aligned");
    a.x = x; a.wp = wp; a.out = y; a.aux = bias; a.relu = relu; a.pool = ypool;
    return ypool ? launch_wino<WK_FWD, true>((hipStream_t)stream, a) : launch_wino<WK_FWD, false>((hipStream_t)stream, a);
}

extern "C" int vc_conv3x3_wino_dgrad_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* dy, const float* wpt,
                                         const float* relu_src, float* dx) {
    using namespace vc;
    WinoArgs a;
    VC_CHECK_ARG(plan_wino(B, H, W, Cout, Cin, a.g), "unsupported shape (vc_conv3x3_wino_supported)");
    VC_CHECK_ARG(dy && wpt && dx, "null pointer");
    VC_CHECK_ARG(waligned16(dy) && waligned16(wpt) && waligned16(dx) && waligned16(relu_src), "pointers must be 16-byte aligned");
    a.x = dy; a.wp = wpt; a.out = dx; a.aux = relu_src; a.relu = 0; a.pool = nullptr;
    return launch_wino<WK_DGRAD, false>((hipStream_t)stream, a);
}
