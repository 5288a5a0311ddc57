// Winograd F(2x2, 3x3) forward / data gradient, second tiling (see conv_wino.hip for the algorithm): v_mfma_f32_16x16x4_f32 tiles, a wave
// owns a block of up to SIXTEEN 2x2 tiles x 64 output channels x the sixteen positions (64 accumulators of four registers = 256 AGPRs).
// Why: with one wave per SIMD every VALU instruction between two MFMAs costs matrix-pipe time (tools/probes/mfma_fillers.hip).  Here a
// transformed input value feeds FOUR MFMAs (four 16-channel column tiles) instead of one, so the input transform is 8 packed additions
// per 32 MFMAs (tools/probes/mfma16_fillers.hip: 137 TFLOP/s with this mix, 118 with the 32x32x2 kernel's), and 4 x 4-tile blocks fit
// the 56-wide layers exactly (the 32-tile blocks waste an eighth there).
//   * workgroup = four waves = four blocks x the same 64 columns; per 16-channel chunk the LDS holds the blocks' halo patches
//     (10 x 10 pixels, pitch 20 floats; in odd tile rows the two channel quads of a half are swapped on the global side, which makes the
//     ds_read_b64 of a 4 x 4-tile block conflict-free) and the chunk's transformed weights [half 2][p 16][column tile 4][k group 4][n 16][e 2];
//   * lane = (tile j = lane % 16, k group g = lane / 16): the MFMA's four k are the four lane groups, a half (8 channels) is two
//     k-steps (e), lane group g works on channels 8 q + 2 g + e: float2 per patch pixel, float2 per weight fragment;
//   * unit = (half q, position row xi): 4 positions x 4 column tiles x 2 k-steps = 32 MFMAs on 16 different accumulators; the next
//     unit's patch reads (4 or 8 ds_read_b64), sixteen weight reads and eight float2 additions go one per gap;
//   * staging / half-phase pipeline, epilogue (output transform, bias, ReLU / mask bits, fused max-pool): as conv_wino.hip.
#include <stdlib.h>
#include "conv_wino.h"

namespace vc {

typedef float w16f2 __attribute__((ext_vector_type(2)));

enum { W16_FWD = 0, W16_DGRAD = 1 };
constexpr int W16_PITCH = 20;                    // floats per patch pixel in LDS (16 channels + 4)
constexpr int W16_PIX = 100;                     // patch pixels per block: (2 TBH + 2)(2 TBW + 2) <= 100
constexpr int W16_BLK = W16_PIX * W16_PITCH;
constexpr int W16_PSLOTS = 4;                    // float4 patch slots per thread and half: 4 blocks x 100 pixels x 2 quads <= 256 x 4
constexpr int W16_VSLOTS = 8;                    // float4 weight pieces per thread and half: 32 KB
constexpr int W16_SLOTS = W16_PSLOTS + W16_VSLOTS;
constexpr int W16_VHALF = 16 * 4 * 4 * 16 * 2;   // floats of one half of a chunk's weights: [p 16][column tile 4][k group 4][n 16][e 2]
constexpr int W16_POFF = 2 * W16_VHALF;          // LDS: weights first, then the patches
constexpr int WINO16_LDS_BYTES = (W16_POFF + 4 * W16_BLK) * 4;

struct Wino16Args {
    WinoGeom g;         // TBH * TBW <= 16
    const float* x;     // [P, C]
    const float* wp;    // packed [N/64][C/16][half 2][p 16][column tile 4][k group 4][n 16][e 2]
    float* out;         // [P, N]
    const float* aux;   // fwd: bias [N] or null; dgrad: ReLU source [P, N] or null
    float* pool;
    int relu;
    int tiles_n, ntiles, nchunks;
};

__device__ __forceinline__ float4 w16_f4add(const float4& a, const float4& b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

template <int KIND, bool POOL>
__global__ __launch_bounds__(256, 1) void conv_wino16_kernel(Wino16Args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const WinoGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lj = lane & 15, lg = lane >> 4;
    const int id = xcd_remap(blockIdx.x, a.ntiles);
    const int tm = id / a.tiles_n, nt = id - tm * a.tiles_n, n0 = nt * 64;
    const int C = g.C, N = g.N;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)((long)g.B * g.H * g.W * C * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.wp, 0, 16 * C * N * 4, 0x00020000);

    // patch slots of this thread: slot s = tid + 256 i = (patch pixel s / 2 of the workgroup's 4 x 100, channel quad 2 q + (s & 1))
    unsigned voff[W16_PSLOTS];
#pragma unroll
    for (int i = 0; i < W16_PSLOTS; ++i) {
        const unsigned pl = (unsigned)(tid >> 1) + 128u * i;
        const unsigned blk = pl / W16_PIX, pix = pl - blk * W16_PIX;
        const unsigned gb = (unsigned)tm * 4u + blk;
        voff[i] = WOOB;
        if (blk < 4 && gb < (unsigned)g.nblocks && pix < (unsigned)(g.PH * g.PW)) {
            const unsigned b = gb / (unsigned)g.blocks_img, rem = gb - b * (unsigned)g.blocks_img;
            const unsigned by = rem / (unsigned)g.bx_n, bx = rem - by * (unsigned)g.bx_n;
            const unsigned py = pix / (unsigned)g.PW, px = pix - py * (unsigned)g.PW;
            const int y = (int)(by * 2u * g.TBH + py) - 1, x = (int)(bx * 2u * g.TBW + px) - 1;
            if ((unsigned)y < (unsigned)g.H && (unsigned)x < (unsigned)g.W)
                voff[i] = (((b * (unsigned)g.H + (unsigned)y) * (unsigned)g.W + (unsigned)x) * (unsigned)C + (((unsigned)tid ^ (py >> 1)) & 1u) * 4u) * 4u;
        }
    }
    const int pst = W16_POFF + (tid >> 1) * W16_PITCH + (tid & 1) * 4;   // slot 0 of half 0; slot i is 128 pixels further, half 1 eight floats
    const unsigned vsrc = (unsigned)(((long)nt * a.nchunks) * (2 * W16_VHALF) * 4) + (unsigned)tid * 16u;   // half-phase h at + h * 32 KB, piece i at + i * 4 KB

    const int ntl = g.TBH * g.TBW;
    const int jt = lj < ntl ? lj : 0;
    const int tyl = jt / g.TBW, txl = jt - tyl * g.TBW;
    const int abase0 = W16_POFF + wave * W16_BLK + ((2 * tyl) * g.PW + 2 * txl) * W16_PITCH + 2 * (lg & 1);
    int aq[2];   // [row pair i >> 1]: this lane's float2 (channels 8 q + 2 lg + e) sits in quad (lg >> 1) ^ (tile-row parity) of the half
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) aq[pr] = abase0 + (((lg >> 1) ^ (tyl & 1) ^ pr) & 1) * 4;
    const int rowp = g.PW * W16_PITCH;
    const int vbase = lg * 32 + lj * 2;   // + q * VHALF + p * 512 + ct * 128: a wave's 64 float2 of one fragment are 512 contiguous bytes

    const int gb = tm * 4 + wave;
    const bool blk_ok = gb < g.nblocks && lj < ntl;
    const int gbc = gb < g.nblocks ? gb : 0;
    const int b = gbc / g.blocks_img, rem = gbc - b * g.blocks_img;
    const int by = rem / g.bx_n, bx = rem - by * g.bx_n;
    const int y0 = (by * g.TBH + tyl) * 2, x0 = (bx * g.TBW + txl) * 2;
    const bool ok00 = blk_ok && y0 < g.H && x0 < g.W, ok01 = ok00 && x0 + 1 < g.W, ok10 = ok00 && y0 + 1 < g.H, ok11 = ok10 && x0 + 1 < g.W;
    const long p00 = ((long)(b * g.H + y0) * g.W + x0) * N;
    const long rowN = (long)g.W * N;
    unsigned mbits[2] = {0xffffffffu, 0xffffffffu};

    f32x4 acc[16][4];   // [position][column tile]: M_p[n0 + 16 ct + 4 lg + r][tile lj]
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[p][ct] = f32x4{0.f, 0.f, 0.f, 0.f};

    float4 st[W16_SLOTS];
    auto gload1 = [&](int hp, int i) {
        if (i < W16_PSLOTS) st[i] = wbufload(rx, voff[i], (unsigned)hp * 32u);
        else st[i] = wbufload(rw, vsrc + (unsigned)(i - W16_PSLOTS) * 4096u, (unsigned)hp * (W16_VHALF * 4));
    };
    auto lstore = [&](int q, int i) {
        if (i < W16_PSLOTS) {
            if (i < W16_PSLOTS - 1 || tid + 256 * i < 2 * 4 * W16_PIX) *reinterpret_cast<float4*>(&smem[pst + q * 8 + i * 128 * W16_PITCH]) = st[i];
        } else {
            *reinterpret_cast<float4*>(&smem[q * W16_VHALF + (tid + 256 * (i - W16_PSLOTS)) * 4]) = st[i];
        }
    };

    // unit (q, xi), xi in the order 0, 2, 1, 3 (each patch row read once per half)
    w16f2 ur[2][4], vf[2][4][4];
    w16f2 dr[4][4], tt[4];
    auto xi_of = [](int u4) { return u4 == 1 ? 2 : u4 == 2 ? 1 : u4; };
    auto rdp = [&](int q, int u4, int k) {   // k-th patch read of the rows unit u4 is the first to need (xi = 0: eight, xi = 2 / 3: four)
        const int xi = xi_of(u4);
        const int j = k & 3;
        const float* p0 = &smem[aq[0] + q * 8 + j * W16_PITCH];   // patch rows 0, 1
        const float* p1 = &smem[aq[1] + q * 8 + j * W16_PITCH];   // patch rows 2, 3
        if (xi == 0 && k < 4) dr[0][j] = *reinterpret_cast<const w16f2*>(p0);
        if (xi == 0 && k >= 4) dr[2][j] = *reinterpret_cast<const w16f2*>(p1 + 2 * rowp);
        if (xi == 2 && k < 4) dr[1][j] = *reinterpret_cast<const w16f2*>(p0 + rowp);
        if (xi == 3 && k < 4) dr[3][j] = *reinterpret_cast<const w16f2*>(p1 + 3 * rowp);
    };
    auto rdv = [&](int q, int u4, int buf, int k) {   // k = 4 nu + ct
        const int nu = k >> 2, ct = k & 3;
        vf[buf][nu][ct] = *reinterpret_cast<const w16f2*>(&smem[vbase + q * W16_VHALF + (xi_of(u4) * 4 + nu) * 512 + ct * 128]);
    };
    auto tstep = [&](int u4, int buf, int k) {   // eight float2 additions
        const int xi = xi_of(u4);
        if (k < 4) tt[k] = xi == 0 ? dr[0][k] - dr[2][k] : xi == 1 ? dr[1][k] + dr[2][k] : xi == 2 ? dr[2][k] - dr[1][k] : dr[1][k] - dr[3][k];
        if (k == 4) ur[buf][0] = tt[0] - tt[2];
        if (k == 5) ur[buf][1] = tt[1] + tt[2];
        if (k == 6) ur[buf][2] = tt[2] - tt[1];
        if (k == 7) ur[buf][3] = tt[1] - tt[3];
    };
    auto mf = [&](int u4, int buf, int m) {     // m = 16 e + 4 nu + ct: sixteen different accumulators in a row
        const int xi = xi_of(u4), e = m >> 4, nu = (m >> 2) & 3, ct = m & 3;
        // Inline asm with the accumulator constrained to AGPRs: as a builtin (64 four-register values, or slices of sixteen 16-register
        // tuples) hipcc 7.2 keeps part of the accumulators in VGPRs and shuffles ~280 v_accvgpr_read / write / mov per chunk through the
        // loop.  The price: the compiler's hazard recogniser does not see these MFMAs.  Measured: without any wait state the first
        // column tile of every position comes out wrong (the MFMA whose B operand differs from its predecessor's); one s_nop 1 in
        // front of those MFMAs (or of all of them, -2.5 %) gives results equal to the builtin form on every test shape.
        if (ct == 0)
            asm volatile("s_nop 1\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[xi * 4 + nu][ct]) : "v"(vf[buf][nu][ct][e]), "v"(ur[buf][nu][e]));
        else
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[xi * 4 + nu][ct]) : "v"(vf[buf][nu][ct][e]), "v"(ur[buf][nu][e]));
    };
#define WSB() __builtin_amdgcn_sched_barrier(0)
    auto half = [&](int q, bool more, int nexthp) {
#pragma unroll
        for (int u4 = 0; u4 < 4; ++u4) {
            const int bsel = u4 & 1, nb = bsel ^ 1;
            const bool nxt = u4 < 3 || more;
            const int nq = u4 < 3 ? q : q ^ 1, nu4 = (u4 + 1) & 3;
            if (u4 == 3 && more) __syncthreads();
#pragma unroll
            for (int m = 0; m < 32; ++m) {
                mf(u4, bsel, m);
                WSB();
                if (nxt) {
                    if (m < 8) rdp(nq, nu4, m);                    // gaps 0..7: patch reads
                    else if (m < 24) rdv(nq, nu4, nb, m - 8);      // gaps 8..23: the sixteen weight fragments
                    else tstep(nu4, nb, m - 24);                   // gaps 24..31: one float2 addition each
                }
                if (more && m < W16_SLOTS) {
                    if (u4 == 0) gload1(nexthp, m);
                    if (u4 == 2) lstore(q ^ 1, m);
                }
                WSB();
            }
        }
    };

#pragma unroll
    for (int i = 0; i < W16_SLOTS; ++i) gload1(0, i);
    if (KIND == W16_DGRAD && a.aux) {   // ReLU mask of this lane's 2 x 2 pixels x 16 columns as 64 bits
        mbits[0] = mbits[1] = 0u;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int aa = 0; aa < 2; ++aa)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) {
                    const bool ok = aa == 0 ? (bb == 0 ? ok00 : ok01) : (bb == 0 ? ok10 : ok11);
                    const float4 m = ok ? *reinterpret_cast<const float4*>(a.aux + p00 + aa * rowN + bb * N + n0 + 16 * ct + 4 * lg) : f4zero();
                    const unsigned bits = (m.x > 0.f ? 1u : 0u) | (m.y > 0.f ? 2u : 0u) | (m.z > 0.f ? 4u : 0u) | (m.w > 0.f ? 8u : 0u);
                    mbits[ct >> 1] |= bits << (16 * (ct & 1) + 8 * aa + 4 * bb);
                }
    }
#pragma unroll
    for (int i = 0; i < W16_SLOTS; ++i) lstore(0, i);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) rdp(0, 0, k);
#pragma unroll
    for (int k = 0; k < 16; ++k) rdv(0, 0, 0, k);
#pragma unroll
    for (int k = 0; k < 8; ++k) tstep(0, 0, k);
    WSB();
    for (int ch = 0; ch < a.nchunks; ++ch) {
        const bool more = ch + 1 < a.nchunks;
        half(0, true, 2 * ch + 1);
        half(1, more, 2 * ch + 2);
        // (inline-asm MFMAs: hipcc does not know their latency, and it copies accumulators right behind the loop exit)
        if (!more) asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    }
#undef WSB
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) asm volatile("" : "+a"(acc[p][ct]));   // every accumulator's first read sits behind the wait

    // ---- output transform + epilogue: acc[p][ct][r] = M_p[column n0 + 16 ct + 4 lg + r][tile lj]
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        const int col = n0 + 16 * ct + 4 * lg;
        float4 Y[2][2];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float s[2][4];
#pragma unroll
            for (int nu = 0; nu < 4; ++nu) {
                const float m0 = acc[nu][ct][k], m1 = acc[4 + nu][ct][k], m2 = acc[8 + nu][ct][k], m3 = acc[12 + nu][ct][k];
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
        if (KIND == W16_FWD) {
            if (a.aux) {
                const float4 bv = *reinterpret_cast<const float4*>(a.aux + col);
#pragma unroll
                for (int aa = 0; aa < 2; ++aa)
#pragma unroll
                    for (int bb = 0; bb < 2; ++bb) Y[aa][bb] = w16_f4add(Y[aa][bb], bv);
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
                    const unsigned mb = mbits[ct >> 1] >> (16 * (ct & 1) + 8 * aa + 4 * bb);
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

// w [3][3][Ci][Co] (HWIO) -> V = G g G^T packed [N/64][C/16][half 2][p 16][column tile 4][k group 4][n 16][e 2], channel = 16 chunk + 8 half + 2 g + e
//   transpose 0 (forward): C = Ci, N = Co;  transpose 1 (data gradient): C = Co, N = Ci, flipped taps
__global__ __launch_bounds__(256) void wino16_pack_kernel(const float* __restrict__ w, int Ci, int Co, int transpose, float* __restrict__ out) {
    const int C = transpose ? Co : Ci, N = transpose ? Ci : Co;
    const long total = (long)C * N;
    const int nchunks = C / 16;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
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
        const int nt = n >> 6, nl = n & 63, ch = c >> 4, cc = c & 15, q = cc >> 3, gg = (cc & 7) >> 1, e = cc & 1;
        float* o = out + (((long)nt * nchunks + ch) * 2 + q) * W16_VHALF + (nl >> 4) * 128 + gg * 32 + (nl & 15) * 2 + e;
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) {
            o[(xi * 4 + 0) * 512] = t[xi][0];
            o[(xi * 4 + 1) * 512] = 0.5f * (t[xi][0] + t[xi][1] + t[xi][2]);
            o[(xi * 4 + 2) * 512] = 0.5f * (t[xi][0] - t[xi][1] + t[xi][2]);
            o[(xi * 4 + 3) * 512] = t[xi][2];
        }
    }
}

// blocks of at most 16 tiles whose halo patch fits 100 pixels: 4 x 4 wherever the tile grid divides by four, else the best fit
static bool plan_wino16(int B, int H, int W, int C, int N, WinoGeom& g) {
    g.B = B; g.H = H; g.W = W; g.C = C; g.N = N;
    if (B <= 0 || H < 2 || W < 2 || (H & 1) || (W & 1) || C <= 0 || N <= 0 || C % 16 || N % 64) return false;
    if ((long)B * H * W * (long)(C > N ? C : N) * 4 > 0x7fffffffL || 16L * C * N * 4 > 0x7fffffffL) return false;
    const int TW = W / 2, TH = H / 2;
    int best = 0, best_h = 0;
    double best_eff = 0.0;
    for (int tbw = 1; tbw <= 16 && tbw <= TW; ++tbw) {
        int tbh = 16 / tbw;
        if (tbh > TH) tbh = TH;
        if ((2 * tbh + 2) * (2 * tbw + 2) > W16_PIX) continue;
        const double eff = (double)TW * TH / ((double)cdiv(TW, tbw) * cdiv(TH, tbh) * 16.0);
        if (eff > best_eff + 1e-9) { best_eff = eff; best = tbw; best_h = tbh; }
    }
    if (!best) return false;
    g.TBW = best; g.TBH = best_h;
    g.PW = 2 * g.TBW + 2; g.PH = 2 * g.TBH + 2;
    g.bx_n = cdiv(TW, g.TBW); g.by_n = cdiv(TH, g.TBH);
    g.blocks_img = g.bx_n * g.by_n;
    if ((long)B * g.blocks_img > 0x3fffffffL) return false;
    g.nblocks = B * g.blocks_img;
    return true;
}

static int wino16_attr() {
    static int once = [] {
        hipError_t e = hipSuccess;
        auto set = [&](const void* f) { if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, WINO16_LDS_BYTES); };
        set(reinterpret_cast<const void*>(conv_wino16_kernel<W16_FWD, false>));
        set(reinterpret_cast<const void*>(conv_wino16_kernel<W16_FWD, true>));
        set(reinterpret_cast<const void*>(conv_wino16_kernel<W16_DGRAD, false>));
        return e == hipSuccess ? 0 : fail((int)e, "%s: hipFuncSetAttribute failed", "conv wino16 kernel");
    }();
    return once;
}

template <int KIND, bool POOL>
static int launch_wino16(hipStream_t st, Wino16Args& a) {
    int rc = wino16_attr();
    if (rc) return rc;
    a.tiles_n = a.g.N / 64;
    a.nchunks = a.g.C / 16;
    a.ntiles = cdiv(a.g.nblocks, 4) * a.tiles_n;
    hipLaunchKernelGGL((conv_wino16_kernel<KIND, POOL>), dim3(a.ntiles), dim3(256), WINO16_LDS_BYTES, st, a);
    return launch_status("conv wino16");
}

}  // namespace vc

extern "C" int vc_conv3x3_wino16_supported(int B, int H, int W, int Cin, int Cout, int dgrad) {
    vc::WinoGeom g;
    return (dgrad ? vc::plan_wino16(B, H, W, Cout, Cin, g) : vc::plan_wino16(B, H, W, Cin, Cout, g)) ? 1 : 0;
}

extern "C" int vc_conv3x3_wino16_pack_f32(void* stream, int Cin, int Cout, const float* w, int transpose, float* wp) {
    using namespace vc;
    const int C = transpose ? Cout : Cin, N = transpose ? Cin : Cout;
    VC_CHECK_ARG(C > 0 && N > 0 && C % 16 == 0 && N % 64 == 0, "gathered channels % 16 == 0 and output channels % 64 == 0 required");
    VC_CHECK_ARG(w && wp && waligned16(wp), "null or misaligned pointer");
    const long total = (long)C * N;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(wino16_pack_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, Cin, Cout, transpose, wp);
    return launch_status(__func__);
}

extern "C" int vc_conv3x3_wino16_fwd_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* x, const float* wp,
                                         const float* bias, float* y, float* ypool, int relu) {
    using namespace vc;
    Wino16Args a;
    VC_CHECK_ARG(plan_wino16(B, H, W, Cin, Cout, a.g), "unsupported shape (vc_conv3x3_wino16_supported)");
    VC_CHECK_ARG(x && wp && y, "null pointer");
    VC_CHECK_ARG(waligned16(x) && waligned16(wp) && waligned16(y) && waligned16(bias) && waligned16(ypool), "pointers must be 16-byte aligned");
    a.x = x; a.wp = wp; a.out = y; a.aux = bias; a.relu = relu; a.pool = ypool;
    return ypool ? launch_wino16<W16_FWD, true>((hipStream_t)stream, a) : launch_wino16<W16_FWD, false>((hipStream_t)stream, a);
}

extern "C" int vc_conv3x3_wino16_dgrad_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* dy, const float* wpt,
                                           const float* relu_src, float* dx) {
    using namespace vc;
    Wino16Args a;
    VC_CHECK_ARG(plan_wino16(B, H, W, Cout, Cin, a.g), "unsupported shape (vc_conv3x3_wino16_supported)");
    VC_CHECK_ARG(dy && wpt && dx, "null pointer");
    VC_CHECK_ARG(waligned16(dy) && waligned16(wpt) && waligned16(dx) && waligned16(relu_src), "pointers must be 16-byte aligned");
    a.x = dy; a.wp = wpt; a.out = dx; a.aux = relu_src; a.relu = 0; a.pool = nullptr;
    return launch_wino16<W16_DGRAD, false>((hipStream_t)stream, a);
}
