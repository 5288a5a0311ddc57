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
constexpr int WSLOTS = 6;                        // float4 patch slots per thread and half: 4 blocks x 180 pixels x 2 channel quads <= 256 x 6
constexpr int WV_FLOATS = 2 * 16 * 2 * 32 * 4;   // one chunk of transformed weights: [half 2][p 16][lane half 2][n 32][e 4]
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
    const float* wp;    // packed [N/32][C/16][2][16][2][32][4]
    float* out;         // [P, N]
    const float* aux;   // fwd: bias [N] or null; dgrad: ReLU source [P, N] or null
    float* pool;        // fwd: also max_pool2x2(out) [B, H/2, W/2, N] (null: none)
    int relu;
    int tiles_n, ntiles, nchunks;
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

    // A chunk (16 channels) is staged as two HALVES: half q = channels [8 q, 8 q + 8) of every patch pixel + the weights of those
    // channels (lane half lh works on channel quad 2 q + lh).  While the MFMAs of one half run, the other half's region of the LDS is
    // refilled for the next half-phase, so that LDS writes, global loads and MFMAs overlap all the time; one barrier per half-phase.
    // patch slots of this thread: slot s = tid + 256 i = (patch pixel s / 2 of the workgroup's 4 x 180, quad 2 q + (s & 1))
    unsigned voff[WSLOTS];
#pragma unroll
    for (int i = 0; i < WSLOTS; ++i) {
        const unsigned pl = (unsigned)(tid >> 1) + 128u * i;
        const unsigned blk = pl / WPIX, pix = pl - blk * WPIX;   // (32-bit unsigned divisions: a signed or 64-bit one costs ~100 VALU)
        const unsigned gb = (unsigned)tm * 4u + blk;
        voff[i] = WOOB;
        if (blk < 4 && gb < (unsigned)g.nblocks && pix < (unsigned)(g.PH * g.PW)) {
            const unsigned b = gb / (unsigned)g.blocks_img, rem = gb - b * (unsigned)g.blocks_img;
            const unsigned by = rem / (unsigned)g.bx_n, bx = rem - by * (unsigned)g.bx_n;
            const unsigned py = pix / (unsigned)g.PW, px = pix - py * (unsigned)g.PW;
            const int y = (int)(by * 2u * g.TBH + py) - 1, x = (int)(bx * 2u * g.TBW + px) - 1;
            if ((unsigned)y < (unsigned)g.H && (unsigned)x < (unsigned)g.W)
                voff[i] = (((b * (unsigned)g.H + (unsigned)y) * (unsigned)g.W + (unsigned)x) * (unsigned)C + (unsigned)(tid & 1) * 4u) * 4u;
        }
    }
    const int pst = WP_OFF + (tid >> 1) * WPITCH + (tid & 1) * 4;   // LDS float index of slot 0 of half 0; slot i is 128 pixels further, half 1 eight floats
    const unsigned vsrc = (unsigned)(((long)nt * a.nchunks) * WV_FLOATS * 4) + (unsigned)tid * 16u;  // weights: half-phase h at + h * 16 KB, piece i at + i * 4 KB

    // this lane's tile inside its wave's block, its 4 x 4 patch origin in LDS, its weight fragment origin
    const int ntl = g.TBH * g.TBW;
    const int jt = li < ntl ? li : 0;
    const int tyl = jt / g.TBW, txl = jt - tyl * g.TBW;
    const int abase = WP_OFF + wave * WBLK + ((2 * tyl) * g.PW + 2 * txl) * WPITCH + lh * 4;   // + q * 8 + (i * PW + j) * WPITCH
    const int rowp = g.PW * WPITCH;
    const int vbase = (lh * 32 + li) * 4;                                                      // + q * 4096 + p * 256

    const int gb = tm * 4 + wave;
    const bool blk_ok = gb < g.nblocks && li < ntl;
    const int gbc = gb < g.nblocks ? gb : 0;
    const int b = gbc / g.blocks_img, rem = gbc - b * g.blocks_img;
    const int by = rem / g.bx_n, bx = rem - by * g.bx_n;
    const int y0 = (by * g.TBH + tyl) * 2, x0 = (bx * g.TBW + txl) * 2;
    const bool ok00 = blk_ok && y0 < g.H && x0 < g.W, ok01 = ok00 && x0 + 1 < g.W, ok10 = ok00 && y0 + 1 < g.H, ok11 = ok10 && x0 + 1 < g.W;
    const long p00 = ((long)(b * g.H + y0) * g.W + x0) * N;
    const long rowN = (long)g.W * N;
    unsigned mbits[2] = {0xffffffffu, 0xffffffffu};
    f32x16 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

    float4 st[WSLOTS + 4];           // staging registers: six patch slots + four weight pieces of ONE half
    auto gload1 = [&](int hp, int i) {   // the i-th of the ten global loads of half-phase hp's data
        if (i < WSLOTS) st[i] = wbufload(rx, voff[i], (unsigned)hp * 32u);
        else st[i] = wbufload(rw, vsrc + (unsigned)(i - WSLOTS) * 4096u, (unsigned)hp * 16384u);
    };
    auto lstore = [&](int q, int i) {    // the i-th of the ten LDS writes of a half
        if (i < WSLOTS) {
            if (i < WSLOTS - 1 || tid + 256 * i < 2 * 4 * WPIX) *reinterpret_cast<float4*>(&smem[pst + q * 8 + i * 128 * WPITCH]) = st[i];
        } else {
            *reinterpret_cast<float4*>(&smem[q * 4096 + (tid + 256 * (i - WSLOTS)) * 4]) = st[i];
        }
    };

    // unit (q, xi): position row xi of half q.  U[xi][nu] = (B^T d B)[xi][nu], B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1].
    // Units run in the order xi = 0, 2, 1, 3 so that every patch row is read once per half: rows 0 and 2 for xi = 0, row 1 for
    // xi = 2, nothing for xi = 1, row 3 for xi = 3 (sixteen patch reads + sixteen weight reads per 64 MFMAs).
    // One wave per SIMD: nothing else feeds the matrix pipe while this wave issues anything that is not an MFMA, so the software
    // pipeline is written out gap by gap and pinned with sched_barriers -- consecutive MFMAs go to DIFFERENT accumulators (k-step
    // outer, column inner), every gap between two MFMAs carries at most five other instructions (MI355X_MICROARCH.md: what a
    // single wave hides per MFMA), the additions stay scalar (-fno-slp-vectorize: packed fp32 adds cost more beside MFMAs).
    float4 ur[2][4], vf[2][4];
    float4 dr[4][4], tt[4];
    auto xi_of = [](int u4) { return u4 == 1 ? 2 : u4 == 2 ? 1 : u4; };
    auto rdp = [&](int q, int u4, int j) {   // pixel j of the patch rows unit u4 is the first to need
        const int xi = xi_of(u4);
        const float* pq = &smem[abase + q * 8 + j * WPITCH];
        if (xi == 0) { dr[0][j] = *reinterpret_cast<const float4*>(pq); dr[2][j] = *reinterpret_cast<const float4*>(pq + 2 * rowp); }
        if (xi == 2) dr[1][j] = *reinterpret_cast<const float4*>(pq + rowp);
        if (xi == 3) dr[3][j] = *reinterpret_cast<const float4*>(pq + 3 * rowp);
    };
    auto rdv = [&](int q, int u4, int buf, int nu) { vf[buf][nu] = *reinterpret_cast<const float4*>(&smem[vbase + q * 4096 + (xi_of(u4) * 4 + nu) * 256]); };
    auto tstep = [&](int u4, int buf, int k) {   // eight steps of four additions
        const int xi = xi_of(u4);
        if (k < 4) tt[k] = xi == 0 ? f4sub(dr[0][k], dr[2][k]) : xi == 1 ? f4add(dr[1][k], dr[2][k]) : xi == 2 ? f4sub(dr[2][k], dr[1][k]) : f4sub(dr[1][k], dr[3][k]);
        if (k == 4) ur[buf][0] = f4sub(tt[0], tt[2]);
        if (k == 5) ur[buf][1] = f4add(tt[1], tt[2]);
        if (k == 6) ur[buf][2] = f4sub(tt[2], tt[1]);
        if (k == 7) ur[buf][3] = f4sub(tt[1], tt[3]);
    };
    auto mf = [&](int u4, int buf, int m) {
        const int xi = xi_of(u4), e = m >> 2, nu = m & 3;
        acc[xi * 4 + nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcomp(vf[buf][nu], e), wcomp(ur[buf][nu], e), acc[xi * 4 + nu], 0, 0, 0);
    };
#define WSB() __builtin_amdgcn_sched_barrier(0)
    // one half-phase = the four units of half q of the current chunk; the data of the NEXT half-phase (the other half's LDS region) is
    // loaded from global memory during unit 0 and written to the LDS during unit 2; the barrier at the head of unit 3 closes both this
    // half's last reads and those writes, so unit 3 already prepares the first unit of the next half-phase.
    auto half = [&](int q, bool more, int nexthp) {
#pragma unroll
        for (int u4 = 0; u4 < 4; ++u4) {
            const int b = u4 & 1, nb = b ^ 1;
            const bool nxt = u4 < 3 || more;
            const int nq = u4 < 3 ? q : q ^ 1, nu4 = (u4 + 1) & 3;
            if (u4 == 3 && more) __syncthreads();
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                mf(u4, b, m);
                WSB();
                if (nxt) {
                    if (m < 4) rdp(nq, nu4, m);                       // gaps 0..3: one patch pixel each (two rows for xi = 0)
                    else if (m < 6) { rdv(nq, nu4, nb, 2 * (m - 4)); rdv(nq, nu4, nb, 2 * (m - 4) + 1); }
                    else if (m < 14) tstep(nu4, nb, m - 6);           // gaps 6..13: four additions each
                }
                if (more && m < 10) {                                 // staging of the next half-phase: one operation per gap
                    if (u4 == 0) gload1(nexthp, m);
                    if (u4 == 2) lstore(q ^ 1, m);
                }
                WSB();
            }
        }
    };

#pragma unroll
    for (int i = 0; i < WSLOTS + 4; ++i) gload1(0, i);
    // data gradient: the ReLU mask of this lane's 2 x 2 pixels x 16 columns as 64 bits, loaded while the first patch is in flight
    if (KIND == WK_DGRAD && a.aux) {
        mbits[0] = mbits[1] = 0u;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
#pragma unroll
            for (int aa = 0; aa < 2; ++aa)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) {
                    const bool ok = aa == 0 ? (bb == 0 ? ok00 : ok01) : (bb == 0 ? ok10 : ok11);
                    const float4 m = ok ? *reinterpret_cast<const float4*>(a.aux + p00 + aa * rowN + bb * N + n0 + 8 * rg + 4 * lh) : f4zero();
                    const unsigned bits = (m.x > 0.f ? 1u : 0u) | (m.y > 0.f ? 2u : 0u) | (m.z > 0.f ? 4u : 0u) | (m.w > 0.f ? 8u : 0u);
                    mbits[rg >> 1] |= bits << (16 * (rg & 1) + 8 * aa + 4 * bb);
                }
    }
#pragma unroll
    for (int i = 0; i < WSLOTS + 4; ++i) lstore(0, i);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) rdp(0, 0, j);
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) rdv(0, 0, 0, nu);
#pragma unroll
    for (int k = 0; k < 8; ++k) tstep(0, 0, k);
    WSB();
    for (int ch = 0; ch < a.nchunks; ++ch) {
        half(0, true, 2 * ch + 1);
        half(1, ch + 1 < a.nchunks, 2 * ch + 2);
    }
#undef WSB
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");

    // ---- output transform + epilogue: acc[p][r] = M_p[column n0 + 8 (r >> 2) + 4 lh + (r & 3)][tile li]
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
                    const unsigned mb = mbits[rg >> 1] >> (16 * (rg & 1) + 8 * aa + 4 * bb);
                    float4& v = Y[aa][bb];
                    if (!(mb & 1u)) v.x = 0.f;
                    if (!(mb & 2u)) v.y = 0.f;
                    if (!(mb & 4u)) v.z = 0.f;
                    if (!(mb & 8u)) v.w = 0.f;
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

// w [3][3][Ci][Co] (HWIO) -> V = G g G^T, G = [1 0 0; 1/2 1/2 1/2; 1/2 -1/2 1/2; 0 0 1], packed [N/32][C/16][half 2][p 16][lane half 2][n 32][e 4] (channel = 16 chunk + 8 half + 4 lane half + e):
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
        const int nt = n >> 5, nl = n & 31, ch = c / WCH, quad = (c % WCH) >> 2, e = c & 3;   // quad = 2 * half + lane half
        float* o = out + ((long)nt * nchunks + ch) * WV_FLOATS + (quad >> 1) * 4096 + ((quad & 1) * 32 + nl) * 4 + e;
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) {
            const float v0 = t[xi][0], v1 = 0.5f * (t[xi][0] + t[xi][1] + t[xi][2]), v2 = 0.5f * (t[xi][0] - t[xi][1] + t[xi][2]), v3 = t[xi][2];
            o[(xi * 4 + 0) * 256] = v0;
            o[(xi * 4 + 1) * 256] = v1;
            o[(xi * 4 + 2) * 256] = v2;
            o[(xi * 4 + 3) * 256] = v3;
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
    hipLaunchKernelGGL((conv_wino_kernel<KIND, POOL>), dim3(a.ntiles), dim3(256), WINO_LDS_BYTES, st, a);
    return launch_status("conv wino");
}

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
