// 3x3 / stride 1 / SAME convolution by Winograd's minimal filtering F(2x2, 3x3) for gfx950: forward and data gradient of
// utils/image_embeddings.py:36-212 in fp32 with 2.25x fewer multiplications than the direct form.
//
//   Y = A^T [ sum_c (G g_c G^T) (.) (B^T d_c B) ] A          per 2 x 2 output tile: d = its 4 x 4 input patch, g = the 3 x 3 filter
//
// The sixteen positions of the transformed domain are sixteen independent products M_p[n][tile] = sum_c V_p[c][n] U_p[c][tile]; the
// input transform B^T d B, the position products (MFMA) and the output transform A^T M A run in ONE kernel, nothing transformed touches
// HBM; the weights V = G g G^T are transformed once per optimiser step (vc_conv3x3_wino_pack_f32; transpose 1 = flipped taps,
// transposed channels for the data gradient).  Rounding: the transforms add at most four fp32 terms with coefficients 1, 1/2, 1/4;
// results agree with the direct form to a few 1e-7 of the tensor maximum times sqrt(K) (tests/test_gpu_conv_wino.py, fp64 oracle).
//
// Round-3 kernel: TWO independent workgroups per CU (two waves per SIMD) on v_mfma_f32_16x16x4_f32 tiles (round 2 ran one wave per
// SIMD on sixteen 32 x 32 accumulators; profiles/r03_wino_fwd_dgrad_round2_kernel.txt keeps its per-layer times).
//
// Why (tools/probes/mfma_2wave.hip, profiles/r03_mfma_2wave.txt): with ONE wave per SIMD -- sixteen 32 x 32
// accumulators = 256 registers -- every VALU instruction of the input transform between two MFMAs costs matrix-pipe time (64-68 %
// MFMA-busy), and the ~16 000 cycles of a tile's prologue / epilogue run with nothing else resident.  A second wave on the SIMD takes
// both: its MFMAs issue while the first wave adds, loads or stores (the probe: 2 adds + LDS reads per MFMA 75.6 % -> 84.7 % of the
// issue rate).  Halving the accumulators without splitting a tile's sixteen positions over two waves (which would need an exchange
// before the output transform) means a smaller MFMA tile:
//   * a wave owns a block of up to SIXTEEN 2x2 tiles (4 x 4 tiles, or 2 x 7 on the 28 / 14-wide layers) x 32 output channels x the
//     sixteen positions = 32 accumulators of four registers = 128 registers; 4 x 4 blocks tile the 224 / 112 / 56-wide layers exactly
//     (the 32-tile blocks wasted an eighth of the 56-wide layers);
//   * lane = (tile j = lane % 16, k group g = lane / 16): the MFMA's four k are the four lane groups; a half-phase (8 input channels)
//     is two k-steps e, lane group g works on channels 8 q + 2 g + e -- float2 per patch pixel, one float4 (column tile ct x e) per
//     position of the transformed weights;
//   * workgroup = four waves = four blocks x the same 32 output channels; per 16-channel chunk the LDS holds the blocks' halo patches
//     (<= 100 pixels, pitch 20 floats, quads of odd tile rows swapped on the global side: conflict-free ds_read_b64) and the chunk's
//     transformed weights [half 2][p 16][g 4][n 16][ct 2][e 2]: 64 KB per workgroup, two workgroups per CU;
//   * half-phase staging, output transform, bias in the accumulator of position (1, 1), ReLU / ReLU mask bits (32 per lane), fused
//     2 x 2 max-pool (a pooling window IS a Winograd tile): register math in the epilogue, no LDS, no separate pooling pass.
#include <stdlib.h>
#include "conv_wino.h"

// `make -C vae_captioning_amd/csrc ablate` builds this file with W2_ABL = a bit mask that REMOVES parts of the main loop (results are then wrong;
// timing only): 1 transform additions, 2 patch reads, 4 weight-fragment reads, 8 staging (global loads + LDS writes), 16 barriers,
// 32 patch loads contiguous over the lanes
#ifndef W2_ABL
#define W2_ABL 0
#endif

namespace vc {

typedef float w2f2 __attribute__((ext_vector_type(2)));

enum { W2_FWD = 0, W2_DGRAD = 1 };
// LDS patch of a block: [half q 2][k group g 4][patch row R][pixel pair CP][pixel & 1][e 2] -- the unit a lane reads is 16 bytes = two
// horizontally adjacent pixels x its two channels (8 q + 2 g + e), so a tile's patch row is TWO ds_read_b128 (pixel-major with
// ds_read2_b64: twice the LDS cycles and 2-way bank conflicts).  Planes of W2_PLANE units (16 bytes each) per g, rows of g.P units:
// P = 6 (4 x 4 tiles) and P = 12 (2 x 7 tiles) make the sixteen lanes of every ds_read_b128 lane group hit sixteen different bank quads
// (plan_wino2 searches P; tools/probes/... none needed: SQ_LDS_BANK_CONFLICT = 0 on those shapes)
constexpr int W2_PIX = 100;                      // patch pixels per block: (2 TBH + 2)(2 TBW + 2) <= 100
constexpr int W2_PLANE = 80;                     // 16-byte units per (half, g) plane: PH * P <= 80
constexpr int W2_GSTR = W2_PLANE * 4;            // floats between the planes of g and g + 1
constexpr int W2_QSTR = 4 * W2_GSTR;             // floats between the two halves
constexpr int W2_BLK = 2 * W2_QSTR;              // floats per block patch (10 KB)
constexpr int W2_PSLOTS = 4;                     // float4 patch slots per thread and half: 4 blocks x 100 pixels x 2 quads <= 256 x 4
constexpr int W2_VSLOTS = 4;                     // float4 weight pieces per thread and half: 16 KB
constexpr int W2_SLOTS = W2_PSLOTS + W2_VSLOTS;
constexpr int W2_VHALF = 16 * 4 * 16 * 2 * 2;    // floats of one half of a chunk's weights: [p 16][g 4][n 16][ct 2][e 2]
constexpr int W2_POFF = 2 * W2_VHALF;            // LDS: weights first, then the patches
constexpr int WINO2_LDS_BYTES = (W2_POFF + 4 * W2_BLK) * 4;   // 64 KB: two workgroups per CU

struct Wino2Args {
    WinoGeom g;         // TBH * TBW <= 16
    const float* x;     // [B][C/4][H][W][4] (the C4 activation layout, vaecap.h)
    const float* wp;    // packed [N/32][C/16][half 2][p 16][g 4][n 16][ct 2][e 2]
    float* out;         // [B][N/4][H][W][4]
    const float* aux;   // fwd: bias [N] or null; dgrad: ReLU source, layout of out, or null
    float* pool;        // fwd: also max_pool2x2(out) (null: none)
    unsigned* pbits;    // fwd + pool: MaxPoolGrad routing codes [B][N/4][H/2][W/2] half-words (null: not wanted): per pooled element 4 bits =
                        // position of the first maximum of its 2 x 2 window (row-major) | 4 if that maximum is > 0; a half-word = a channel quad
    unsigned* mask;     // [workgroups][256 threads]: (out > 0) of each lane's 2 x 2 pixels x 8 columns as 32 bits -- written by the forward
                        // (null: not wanted), read by the data gradient of the NEXT layer instead of relu_src (same shape => same lanes)
    int relu;
    int tiles_n, ntiles, nchunks;
};

template <int KIND, bool POOL>
__global__ __launch_bounds__(256, 2) void conv_wino2_kernel(Wino2Args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const WinoGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lj = lane & 15, lg = lane >> 4;
    const int id = xcd_remap(blockIdx.x, a.ntiles);
    const int tm = id / a.tiles_n, nt = id - tm * a.tiles_n, n0 = nt * 32;
    const int C = g.C, N = g.N;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)((long)g.B * g.H * g.W * C * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.wp, 0, 16 * C * N * 4, 0x00020000);

    // patch slots of this thread: slot s = tid + 256 i = (patch pixel s / 2 of the workgroup's 4 x 100, channel quad 2 q + (s & 1))
    unsigned voff[W2_PSLOTS];
    int pst[W2_PSLOTS];   // LDS float index of the slot's first channel pair in half 0
#pragma unroll
    for (int i = 0; i < W2_PSLOTS; ++i) {
        const unsigned pl = (unsigned)(tid >> 1) + 128u * i;
        const unsigned blk = pl / W2_PIX, pix = pl - blk * W2_PIX;
        const unsigned gb = (unsigned)tm * 4u + blk;
        const unsigned gbc = gb < (unsigned)g.nblocks ? gb : 0u;
        const unsigned b = wino_div(gbc, g.m_blocks_img), rem = gbc - b * (unsigned)g.blocks_img;
        const unsigned by = wino_div(rem, g.m_bx_n), bx = rem - by * (unsigned)g.bx_n;
        const unsigned py = wino_div(pix, g.m_pw), px = pix - py * (unsigned)g.PW;
        const int y = (int)(by * 2u * g.TBH + py) - 1, x = (int)(bx * 2u * g.TBW + px) - 1;
        const bool ok = blk < 4 && gb < (unsigned)g.nblocks && pix < (unsigned)(g.PH * g.PW) && (unsigned)y < (unsigned)g.H && (unsigned)x < (unsigned)g.W;
        // C4 layout [B][C/4][H][W][4]: channel quad 2 h + (tid & 1) of half-phase h is a plane (soffset h * 2 planes)
        const unsigned off = ((((b * (unsigned)(C >> 2) + ((unsigned)tid & 1u)) * (unsigned)g.H + (unsigned)y) * (unsigned)g.W + (unsigned)x)) * 16u;
        voff[i] = ok ? off : WOOB;
        // (timing only) the patch loads made contiguous over the lanes: what a channel-blocked activation layout would present to the L1
        if (W2_ABL & 32) voff[i] = (unsigned)(((unsigned)tm * 1024u + (unsigned)tid + 256u * i) * 16u) % (unsigned)(g.B * g.H * g.W * C * 4 - 4096);
        // the slot's float4 = channels 4 (tid & 1) .. + 3 of the half = k groups 2 (tid & 1) (.xy) and 2 (tid & 1) + 1 (.zw)
        // (slots past the workgroup's patches -- a fifth block, pixels past PH x PW -- land on unit 79 of block 0, which no plane uses: PH * P <= 79)
        pst[i] = (blk < 4 && pix < (unsigned)(g.PH * g.PW))
                     ? W2_POFF + (int)blk * W2_BLK + (int)(2u * ((unsigned)tid & 1u)) * W2_GSTR + (int)((py * (unsigned)g.P + (px >> 1)) * 4u + (px & 1u) * 2u)
                     : W2_POFF + (W2_PLANE - 1) * 4;
    }
    const unsigned vsrc = (unsigned)(((long)nt * a.nchunks) * (2 * W2_VHALF) * 4) + (unsigned)tid * 16u;   // half-phase h at + h * 16 KB, piece i at + i * 4 KB

    const int ntl = g.TBH * g.TBW;
    const int jt = lj < ntl ? lj : 0;
    const int tyl = (int)wino_div((unsigned)jt, g.m_tbw), txl = jt - tyl * g.TBW;
    int ab[4];   // LDS float index of this lane's first pixel pair in patch rows 2 tyl + r (half 0); the second pair is 4 floats further
#pragma unroll
    for (int r = 0; r < 4; ++r) ab[r] = W2_POFF + wave * W2_BLK + lg * W2_GSTR + ((2 * tyl + r) * g.P + txl) * 4;
    const int vbase = (lg * 16 + lj) * 4;   // + q * VHALF + p * 256: a wave's 64 float4 of one position are 1 KB of consecutive LDS

    const int gb = tm * 4 + wave;
    const bool blk_ok = gb < g.nblocks && lj < ntl;
    const int gbc = gb < g.nblocks ? gb : 0;
    const int b = (int)wino_div((unsigned)gbc, g.m_blocks_img), rem = gbc - b * g.blocks_img;
    const int by = (int)wino_div((unsigned)rem, g.m_bx_n), bx = rem - by * g.bx_n;
    const int y0 = (by * g.TBH + tyl) * 2, x0 = (bx * g.TBW + txl) * 2;
    const bool ok00 = blk_ok && y0 < g.H && x0 < g.W, ok01 = ok00 && x0 + 1 < g.W, ok10 = ok00 && y0 + 1 < g.H, ok11 = ok10 && x0 + 1 < g.W;
    // output in the C4 layout: the lane's two float4 (ct = 0, 1) are channel quads n0 / 4 + 2 lg + ct = two planes
    const long pl_f = (long)g.H * g.W * 4;   // floats per channel-quad plane
    const long p00 = (((long)(b * (N >> 2) + (n0 >> 2) + 2 * lg) * g.H + y0) * g.W + x0) * 4;
    const long rowN = (long)g.W * 4;
    unsigned mbits = 0xffffffffu;

    // [position][column tile]: M_p[n0 + 8 lg + 4 ct + r][tile lj] -- row m = 4 lg + r of column tile ct stands for output channel
    // 8 (m >> 2) + 4 ct + (m & 3) of the workgroup's 32 (wino2_pack_kernel), so a lane's two float4 are 32 consecutive bytes of a pixel and
    // four lanes write a full 128-byte line (column = 16 ct + m: two 64-byte pieces per pixel)
    f32x4 acc[16][2];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) acc[p][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    // forward: the bias rides in the accumulator of position (1, 1) -- A^T's column 1 is (1, 1), so A^T M A adds M_(1,1) to all four
    // outputs of the tile; its load overlaps the first patch loads and the epilogue has no load left
    if (KIND == W2_FWD && a.aux) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const float4 bv = *reinterpret_cast<const float4*>(a.aux + n0 + 8 * lg + 4 * ct);
            acc[5][ct] = f32x4{bv.x, bv.y, bv.z, bv.w};
        }
    }

    float4 st[W2_SLOTS];
    const unsigned plane2_b = (unsigned)g.H * (unsigned)g.W * 32u;   // bytes of two channel-quad planes = the eight channels of a half-phase
    auto gload1 = [&](int hp, int i) {
        if (i < W2_PSLOTS) st[i] = wbufload(rx, voff[i], (unsigned)hp * plane2_b);
        else st[i] = wbufload(rw, vsrc + (unsigned)(i - W2_PSLOTS) * 4096u, (unsigned)hp * (W2_VHALF * 4));
    };
    auto lstore = [&](int q, int i) {
        if (i < W2_PSLOTS) {
            *reinterpret_cast<float2*>(&smem[pst[i] + q * W2_QSTR]) = make_float2(st[i].x, st[i].y);
            *reinterpret_cast<float2*>(&smem[pst[i] + q * W2_QSTR + W2_GSTR]) = make_float2(st[i].z, st[i].w);
        } else {
            *reinterpret_cast<float4*>(&smem[q * W2_VHALF + (tid + 256 * (i - W2_PSLOTS)) * 4]) = st[i];
        }
    };

    // unit (q, xi), xi in the order 0, 2, 1, 3 (each patch row read once per half): 4 positions x 2 column tiles x 2 k-steps = 16 MFMAs
    // ur / vf are SINGLE-buffered: the MFMAs of a unit run position by position (nu-major), so the fragments of position nu are dead
    // after its four MFMAs and the next unit's values for nu are written right behind them (in-order issue: the MFMA has read its operands)
    w2f2 ur[4];
    float4 vf[4];
    w2f2 dr[4][4], tt[4];
    auto xi_of = [](int u4) { return u4 == 1 ? 2 : u4 == 2 ? 1 : u4; };
    auto rdp = [&](int q, int u4, int k) {   // k-th patch read (one ds_read_b128 = two pixels) of the rows unit u4 is the first to need: xi = 0 four, xi = 2 / 3 two
        const int xi = xi_of(u4);
        if (W2_ABL & 2) return;
        const int c = k & 1;
        const int row = xi == 0 ? (k < 2 ? 0 : 2) : xi == 2 ? 1 : 3;
        if (xi == 1 || (xi != 0 && k >= 2)) return;
        const float4 v = *reinterpret_cast<const float4*>(&smem[ab[row] + q * W2_QSTR + c * 4]);
        dr[row][2 * c] = w2f2{v.x, v.y};
        dr[row][2 * c + 1] = w2f2{v.z, v.w};
    };
    auto rdv = [&](int q, int u4, int nu) {
        if (W2_ABL & 4) return;
        vf[nu] = *reinterpret_cast<const float4*>(&smem[vbase + q * W2_VHALF + (xi_of(u4) * 4 + nu) * 256]);
    };
    // eight steps of two SCALAR additions (component by component, -fno-slp-vectorize for this file): with two waves per SIMD a
    // v_pk_add_f32 beside the partner's MFMAs costs more matrix-pipe time than two v_add_f32 (tools/probes/mfma_2wave.hip: one packed
    // add per MFMA 82 % of the issue rate with one, two or four waves per SIMD; two scalar adds 78 % -> 87 % with a second wave)
    auto sub2 = [](const w2f2& a, const w2f2& b) { w2f2 r; r.x = a.x - b.x; r.y = a.y - b.y; return r; };
    auto add2 = [](const w2f2& a, const w2f2& b) { w2f2 r; r.x = a.x + b.x; r.y = a.y + b.y; return r; };
    auto tstep = [&](int u4, int k) {
        const int xi = xi_of(u4);
        if (W2_ABL & 1) return;
        if (k < 4) tt[k] = xi == 0 ? sub2(dr[0][k], dr[2][k]) : xi == 1 ? add2(dr[1][k], dr[2][k]) : xi == 2 ? sub2(dr[2][k], dr[1][k]) : sub2(dr[1][k], dr[3][k]);
        if (k == 4) ur[0] = sub2(tt[0], tt[2]);
        if (k == 5) ur[1] = add2(tt[1], tt[2]);
        if (k == 6) ur[2] = sub2(tt[2], tt[1]);
        if (k == 7) ur[3] = sub2(tt[1], tt[3]);
    };
    auto mf = [&](int u4, int m) {     // m = 4 nu + 2 e + ct: an accumulator comes back after two MFMAs (64 cycles >= the 40 of a dependent issue)
        const int xi = xi_of(u4), nu = m >> 2, e = (m >> 1) & 1, ct = m & 1;
        const float4& v = vf[nu];
        const float av = ct == 0 ? (e == 0 ? v.x : v.y) : (e == 0 ? v.z : v.w);
        acc[xi * 4 + nu][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, ur[nu][e], acc[xi * 4 + nu][ct], 0, 0, 0);
    };
#define WSB() __builtin_amdgcn_sched_barrier(0)
    // one half-phase h = the four units of half q = h & 1.  Staging runs two half-phases ahead: the data of half-phase h + 2 is requested
    // during unit 3 (global loads: three units = ~50 MFMAs of latency cover before anybody needs the registers), the data of h + 1 --
    // requested one half-phase earlier -- is written to the other half's LDS region during unit 2, and the barrier at the head of
    // unit 3 closes this half's last reads and those writes.  st_more / ld_more are compile-time (the last chunk is peeled): no
    // branches inside the MFMA stream.
    auto half = [&](int q, bool st_more, bool ld_more, int ldhp) {
#pragma unroll
        for (int u4 = 0; u4 < 4; ++u4) {
            const bool nxt = u4 < 3 || st_more;
            const int nq = u4 < 3 ? q : q ^ 1, nu4 = (u4 + 1) & 3;
            if (u4 == 3 && st_more && !(W2_ABL & 16)) __syncthreads();
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                mf(u4, m);
                WSB();
                if (nxt) {   // the next unit's operands
                    if (m < 4) rdp(nq, nu4, m);                                    // gaps 0..3: patch reads
                    else if (m >= 8 && m < 12) tstep(nu4, m - 8);                  // gaps 8..11: the vertical half of the transform (tt), >= 4 MFMAs behind its reads
                    else if (m >= 12) tstep(nu4, m - 8);                           // gaps 12..15: ur[0..3], each behind its position's last MFMA
                    if ((m & 3) == 3) rdv(nq, nu4, m >> 2);                        // gaps 3, 7, 11, 15: the weight fragment of the position just finished
                }
                if (!(W2_ABL & 8)) {   // staging: LDS writes behind the unit's patch reads (gaps 4..7 and 8..11), global loads in the first eight gaps
                    if (u4 == 2 && st_more && m >= 4 && m < 4 + W2_SLOTS) lstore(q ^ 1, m - 4);
                    if (u4 == 3 && ld_more && m < W2_SLOTS) gload1(ldhp, m);
                }
                WSB();
            }
        }
    };

#pragma unroll
    for (int i = 0; i < W2_SLOTS; ++i) gload1(0, i);
    if (KIND == W2_DGRAD && a.mask) {   // the producer's forward left the mask as bits in this kernel's lane order: one 4-byte load
        mbits = a.mask[(size_t)id * 256 + tid];
    } else if (KIND == W2_DGRAD && a.aux) {   // ReLU mask of this lane's 2 x 2 pixels x 8 columns as 32 bits
        mbits = 0u;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int aa = 0; aa < 2; ++aa)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) {
                    const bool ok = aa == 0 ? (bb == 0 ? ok00 : ok01) : (bb == 0 ? ok10 : ok11);
                    const float4 m = ok ? *reinterpret_cast<const float4*>(a.aux + p00 + ct * pl_f + aa * rowN + bb * 4) : f4zero();
                    const unsigned bits = (m.x > 0.f ? 1u : 0u) | (m.y > 0.f ? 2u : 0u) | (m.z > 0.f ? 4u : 0u) | (m.w > 0.f ? 8u : 0u);
                    mbits |= bits << (16 * ct + 8 * aa + 4 * bb);
                }
    }
#pragma unroll
    for (int i = 0; i < W2_SLOTS; ++i) lstore(0, i);
#pragma unroll
    for (int i = 0; i < W2_SLOTS; ++i) gload1(1, i);   // half-phase 1 (always exists: a chunk is two half-phases)
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) rdp(0, 0, k);
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) rdv(0, 0, nu);
#pragma unroll
    for (int k = 0; k < 8; ++k) tstep(0, k);
    WSB();
    for (int ch = 0; ch + 1 < a.nchunks; ++ch) {
        half(0, true, true, 2 * ch + 2);
        half(1, true, true, 2 * ch + 3);
    }
    half(0, true, false, 0);    // the last chunk: its second half is in the registers already, nothing left to request
    half(1, false, false, 0);
#undef WSB

    // ---- output transform + epilogue: acc[p][ct][r] = M_p[column n0 + 8 lg + 4 ct + r][tile lj]
    unsigned obits = 0u;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        const int col = n0 + 8 * lg + 4 * ct;
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
        if (KIND == W2_FWD) {   // (the bias is already in the accumulators)
            if (a.relu) {
#pragma unroll
                for (int aa = 0; aa < 2; ++aa)
#pragma unroll
                    for (int bb = 0; bb < 2; ++bb) {
                        float4& v = Y[aa][bb];
                        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                    }
            }
        } else if (a.aux || a.mask) {
#pragma unroll
            for (int aa = 0; aa < 2; ++aa)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) {
                    const unsigned mb = mbits >> (16 * ct + 8 * aa + 4 * bb);
                    float4& v = Y[aa][bb];
                    if (!(mb & 1u)) v.x = 0.f;
                    if (!(mb & 2u)) v.y = 0.f;
                    if (!(mb & 4u)) v.z = 0.f;
                    if (!(mb & 8u)) v.w = 0.f;
                }
        }
        if (KIND == W2_FWD && a.mask) {
#pragma unroll
            for (int aa = 0; aa < 2; ++aa)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) {
                    const float4& v = Y[aa][bb];
                    const unsigned bits = (v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u);
                    obits |= bits << (16 * ct + 8 * aa + 4 * bb);
                }
        }
        float* oq = a.out + p00 + ct * pl_f;
        if (ok00) *reinterpret_cast<float4*>(oq) = Y[0][0];
        if (ok01) *reinterpret_cast<float4*>(oq + 4) = Y[0][1];
        if (ok10) *reinterpret_cast<float4*>(oq + rowN) = Y[1][0];
        if (ok11) *reinterpret_cast<float4*>(oq + rowN + 4) = Y[1][1];
        if (POOL && ok11) {
            float4 m;
            m.x = fmaxf(fmaxf(Y[0][0].x, Y[0][1].x), fmaxf(Y[1][0].x, Y[1][1].x));
            m.y = fmaxf(fmaxf(Y[0][0].y, Y[0][1].y), fmaxf(Y[1][0].y, Y[1][1].y));
            m.z = fmaxf(fmaxf(Y[0][0].z, Y[0][1].z), fmaxf(Y[1][0].z, Y[1][1].z));
            m.w = fmaxf(fmaxf(Y[0][0].w, Y[0][1].w), fmaxf(Y[1][0].w, Y[1][1].w));
            const size_t pp = (((size_t)(b * (N >> 2) + (col >> 2)) * (g.H >> 1) + (y0 >> 1)) * (g.W >> 1) + (x0 >> 1)) * 4;
            *reinterpret_cast<float4*>(a.pool + pp) = m;
            if (a.pbits) {   // where MaxPoolGrad will send the gradient (vc_maxpool2x2_bwd_bits_f32): first maximum in row-major order, valid if > 0
                auto code = [](float v00, float v01, float v10, float v11, float mx) -> unsigned {
                    return (v00 == mx ? 0u : v01 == mx ? 1u : v10 == mx ? 2u : 3u) | (mx > 0.f ? 4u : 0u);
                };
                const unsigned c16 = code(Y[0][0].x, Y[0][1].x, Y[1][0].x, Y[1][1].x, m.x) | code(Y[0][0].y, Y[0][1].y, Y[1][0].y, Y[1][1].y, m.y) << 4 |
                                     code(Y[0][0].z, Y[0][1].z, Y[1][0].z, Y[1][1].z, m.z) << 8 | code(Y[0][0].w, Y[0][1].w, Y[1][0].w, Y[1][1].w, m.w) << 12;
                reinterpret_cast<unsigned short*>(a.pbits)[pp >> 2] = (unsigned short)c16;   // one half-word per (channel quad, pooled pixel)
            }
        }
    }
    if (KIND == W2_FWD && a.mask) a.mask[(size_t)id * 256 + tid] = obits;
}

// w [3][3][Ci][Co] (HWIO) -> V = G g G^T, G = [1 0 0; 1/2 1/2 1/2; 1/2 -1/2 1/2; 0 0 1], packed
// [N/32][C/16][half 2][p 16][g 4][n 16][ct 2][e 2] (channel = 16 chunk + 8 half + 2 g + e, column = 32 nt + 8 (n >> 2) + 4 ct + (n & 3)):
//   transpose 0 (forward):        C = Ci, N = Co, g[ky][kx] = w[ky][kx][c][n]
//   transpose 1 (data gradient):  C = Co, N = Ci, g[ky][kx] = w[2 - ky][2 - kx][n][c]
__global__ __launch_bounds__(256) void wino2_pack_kernel(const float* __restrict__ w, int Ci, int Co, int transpose, float* __restrict__ out) {
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
        const int nt = n >> 5, ct = (n >> 2) & 1, nn = ((n & 31) >> 3) * 4 + (n & 3), ch = c >> 4, cc = c & 15, q = cc >> 3, gg = (cc & 7) >> 1, e = cc & 1;
        float* o = out + (((long)nt * nchunks + ch) * 2 + q) * W2_VHALF + (gg * 16 + nn) * 4 + ct * 2 + e;
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) {
            o[(xi * 4 + 0) * 256] = t[xi][0];
            o[(xi * 4 + 1) * 256] = 0.5f * (t[xi][0] + t[xi][1] + t[xi][2]);
            o[(xi * 4 + 2) * 256] = 0.5f * (t[xi][0] - t[xi][1] + t[xi][2]);
            o[(xi * 4 + 3) * 256] = t[xi][2];
        }
    }
}

// ---- host side ----------------------------------------------------------------------------------
// Pitch P (pixel pairs per patch row in the LDS) with the fewest bank conflicts of the kernel's ds_read_b128 patch reads: a wave64
// ds_read_b128 is served in four groups of sixteen lanes (MI355X_MICROARCH.md, LDS), a group is conflict-free when its sixteen 16-byte
// units fall into sixteen different bank quads (unit index mod 16).  Lane = (tile lj = lane % 16, k group lg = lane / 16) reads unit
// lg * W2_PLANE + (2 ty + r) * P + tx + c.  0: the patch does not fit a plane.
static int wino2_row_pitch_search(int TBH, int TBW);
static int wino2_row_pitch(int TBH, int TBW) {   // memoised: the search costs ~0.1 ms and every launch plans its geometry
    static int cache[17][17];   // 0 = not searched yet, -1 = does not fit; block shapes have TBH * TBW <= 16
    if (TBH < 1 || TBW < 1 || TBH > 16 || TBW > 16) return 0;
    int& c = cache[TBH][TBW];
    if (c == 0) { const int p = wino2_row_pitch_search(TBH, TBW); c = p > 0 ? p : -1; }
    return c > 0 ? c : 0;
}
static int wino2_row_pitch_search(int TBH, int TBW) {
    static const int groups[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27}, {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                      {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59}, {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
    const int PW = 2 * TBW + 2, PH = 2 * TBH + 2, ntl = TBH * TBW;
    int best = 0, best_cost = 1 << 30;
    for (int P = PW / 2; PH * P <= W2_PLANE - 1; ++P) {   // (unit W2_PLANE - 1 stays free: the dump slot of the staging writes)
        int cost = 0;
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 2; ++c)
                for (int gi = 0; gi < 4; ++gi) {
                    int units[16], worst = 1;
                    for (int k = 0; k < 16; ++k) {
                        const int lane = groups[gi][k], lj = lane & 15, lg = lane >> 4, jt = lj < ntl ? lj : 0;
                        units[k] = lg * W2_PLANE + (2 * (jt / TBW) + r) * P + jt % TBW + c;
                    }
                    for (int k = 0; k < 16; ++k) {   // distinct units on the bank quad of unit k (equal addresses broadcast)
                        int ways = 0;
                        for (int j = 0; j < 16; ++j) {
                            if ((units[j] & 15) != (units[k] & 15)) continue;
                            bool seen = false;
                            for (int i = 0; i < j; ++i) seen = seen || units[i] == units[j];
                            if (!seen) ++ways;
                        }
                        if (ways > worst) worst = ways;
                    }
                    cost += worst;
                }
        if (cost < best_cost) { best_cost = cost; best = P; }
    }
    return best;
}

// blocks of at most 16 tiles whose halo patch fits 100 pixels: 4 x 4 wherever the tile grid divides by four (224 / 112 / 56 wide),
// 2 x 7 on the 28 / 14-wide layers, in general the shape with the fewest empty slots
static bool plan_wino2(int B, int H, int W, int C, int N, WinoGeom& g) {
    g.B = B; g.H = H; g.W = W; g.C = C; g.N = N;
    if (B <= 0 || H < 2 || W < 2 || (H & 1) || (W & 1) || C <= 0 || N <= 0 || C % 16 || N % 32) return false;
    if ((long)B * H * W * (long)(C > N ? C : N) * 4 > 0x7fffffffL || 16L * C * N * 4 > 0x7fffffffL) return false;
    const int TW = W / 2, TH = H / 2;
    int best = 0, best_h = 0;
    double best_eff = 0.0;
    for (int tbw = 1; tbw <= 16 && tbw <= TW; ++tbw) {
        int tbh = 16 / tbw;
        if (tbh > TH) tbh = TH;
        if ((2 * tbh + 2) * (2 * tbw + 2) > W2_PIX) continue;
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
    if ((long)g.nblocks * g.blocks_img >= 0x100000000L) return false;   // the reciprocal divisions are exact below 2^32 / divisor
    g.m_blocks_img = wino_magic(g.blocks_img); g.m_bx_n = wino_magic(g.bx_n); g.m_pw = wino_magic(g.PW); g.m_tbw = wino_magic(g.TBW);
    g.P = wino2_row_pitch(g.TBH, g.TBW);
    return g.P > 0;
}

static int wino2_attr() {
    static int once = [] {
        hipError_t e = hipSuccess;
        auto set = [&](const void* f) { if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, WINO2_LDS_BYTES); };
        set(reinterpret_cast<const void*>(conv_wino2_kernel<W2_FWD, false>));
        set(reinterpret_cast<const void*>(conv_wino2_kernel<W2_FWD, true>));
        set(reinterpret_cast<const void*>(conv_wino2_kernel<W2_DGRAD, false>));
        return e == hipSuccess ? 0 : fail((int)e, "%s: hipFuncSetAttribute failed", "conv wino2 kernel");
    }();
    return once;
}

template <int KIND, bool POOL>
static int launch_wino2(hipStream_t st, Wino2Args& a) {
    int rc = wino2_attr();
    if (rc) return rc;
    a.tiles_n = a.g.N / 32;
    a.nchunks = a.g.C / 16;
    a.ntiles = cdiv(a.g.nblocks, 4) * a.tiles_n;
    hipLaunchKernelGGL((conv_wino2_kernel<KIND, POOL>), dim3(a.ntiles), dim3(256), WINO2_LDS_BYTES, st, a);
    return launch_status("conv wino2");
}

}  // namespace vc

// ---- C ABI -----------------------------------------------------------------------------------------
extern "C" int vc_conv3x3_wino_supported(int B, int H, int W, int Cin, int Cout, int dgrad) {
    vc::WinoGeom g;
    const int nb = vc::wino_images_per_launch(B, H, W, Cin, Cout);
    return nb > 0 && (dgrad ? vc::plan_wino2(nb, H, W, Cout, Cin, g) : vc::plan_wino2(nb, H, W, Cin, Cout, g)) ? 1 : 0;
}

extern "C" int vc_conv3x3_wino_single_launch_supported(int B, int H, int W, int Cin, int Cout) {
    return vc::wino_images_per_launch(B, H, W, Cin, Cout) >= B ? 1 : 0;
}

extern "C" int vc_conv3x3_wino_pack_f32(void* stream, int Cin, int Cout, const float* w, int transpose, float* wp) {
    using namespace vc;
    const int C = transpose ? Cout : Cin, N = transpose ? Cin : Cout;
    VC_CHECK_ARG(C > 0 && N > 0 && C % 16 == 0 && N % 32 == 0, "gathered channels % 16 == 0 and output channels % 32 == 0 required");
    VC_CHECK_ARG(w && wp && waligned16(wp), "null or misaligned pointer");
    const long total = (long)Cin * Cout;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(wino2_pack_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, Cin, Cout, transpose, wp);
    return launch_status(__func__);
}

namespace vc {
// one launch over nb images; C = gathered channels, N = produced channels
static int wino2_launch(hipStream_t st, int kind, int nb, int H, int W, int C, int N, const float* x, const float* wp, float* out, const float* aux,
                        float* pool, unsigned* mask, int relu, unsigned* pbits = nullptr) {
    Wino2Args a;
    if (!plan_wino2(nb, H, W, C, N, a.g)) return fail(VC_EINVAL, "%s: unsupported shape (vc_conv3x3_wino_supported)", "conv wino");
    a.x = x; a.wp = wp; a.out = out; a.aux = aux; a.pool = pool; a.mask = mask; a.relu = relu; a.pbits = pbits;
    if (kind == W2_DGRAD) return launch_wino2<W2_DGRAD, false>(st, a);
    return pool ? launch_wino2<W2_FWD, true>(st, a) : launch_wino2<W2_FWD, false>(st, a);
}
}  // namespace vc

extern "C" int vc_conv3x3_wino_fwd_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* x, const float* wp,
                                       const float* bias, float* y, float* ypool, int relu) {
    using namespace vc;
    VC_CHECK_ARG(x && wp && y, "null pointer");
    VC_CHECK_ARG(waligned16(x) && waligned16(wp) && waligned16(y) && waligned16(bias) && waligned16(ypool), "pointers must be 16-byte aligned");
    const int per = wino_images_per_launch(B, H, W, Cin, Cout);
    VC_CHECK_ARG(per > 0 && vc_conv3x3_wino_supported(B, H, W, Cin, Cout, 0), "unsupported shape (vc_conv3x3_wino_supported)");
    for (int b0 = 0; b0 < B; b0 += per) {   // image ranges of < 2 GiB (one launch for every VGG16 layer up to 160 images)
        const int nb = B - b0 < per ? B - b0 : per;
        const int rc = wino2_launch((hipStream_t)stream, W2_FWD, nb, H, W, Cin, Cout, x + (size_t)b0 * H * W * Cin, wp, y + (size_t)b0 * H * W * Cout, bias,
                                    ypool ? ypool + (size_t)b0 * (H / 2) * (W / 2) * Cout : nullptr, nullptr, relu);
        if (rc) return rc;
    }
    return 0;
}

extern "C" size_t vc_conv3x3_wino_pool_words(int B, int H, int W, int C) { return (size_t)B * (H / 2) * (W / 2) * (C / 8); }

extern "C" int vc_conv3x3_wino_fwd_pool_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* x, const float* wp,
                                            const float* bias, float* y, float* ypool, uint32_t* pool_bits) {
    using namespace vc;
    VC_CHECK_ARG(x && wp && y && ypool && pool_bits, "null pointer");
    VC_CHECK_ARG(waligned16(x) && waligned16(wp) && waligned16(y) && waligned16(bias) && waligned16(ypool), "pointers must be 16-byte aligned");
    const int per = wino_images_per_launch(B, H, W, Cin, Cout);
    VC_CHECK_ARG(per > 0 && vc_conv3x3_wino_supported(B, H, W, Cin, Cout, 0), "unsupported shape (vc_conv3x3_wino_supported)");
    for (int b0 = 0; b0 < B; b0 += per) {
        const int nb = B - b0 < per ? B - b0 : per;
        const size_t po = (size_t)b0 * (H / 2) * (W / 2) * Cout;
        const int rc = wino2_launch((hipStream_t)stream, W2_FWD, nb, H, W, Cin, Cout, x + (size_t)b0 * H * W * Cin, wp, y + (size_t)b0 * H * W * Cout, bias,
                                    ypool + po, nullptr, 1, pool_bits + po / 8);
        if (rc) return rc;
    }
    return 0;
}

extern "C" size_t vc_conv3x3_wino_mask_words(int B, int H, int W, int C) {
    vc::WinoGeom g;
    if (!vc::plan_wino2(B, H, W, 16, C, g)) return 0;
    return (size_t)vc::cdiv(g.nblocks, 4) * (C / 32) * 256;
}

extern "C" int vc_conv3x3_wino_fwd_mask_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* x, const float* wp,
                                            const float* bias, float* y, int relu, uint32_t* mask_out) {
    using namespace vc;
    VC_CHECK_ARG(x && wp && y && mask_out, "null pointer");
    VC_CHECK_ARG(waligned16(x) && waligned16(wp) && waligned16(y) && waligned16(bias) && waligned16(mask_out), "pointers must be 16-byte aligned");
    VC_CHECK_ARG(wino_images_per_launch(B, H, W, Cin, Cout) >= B && vc_conv3x3_wino_supported(B, H, W, Cin, Cout, 0),
                 "unsupported shape, or more images than one launch takes (vc_conv3x3_wino_single_launch_supported): the mask bits are per tile of ONE launch");
    return wino2_launch((hipStream_t)stream, W2_FWD, B, H, W, Cin, Cout, x, wp, y, bias, nullptr, mask_out, relu);
}

extern "C" int vc_conv3x3_wino_dgrad_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* dy, const float* wpt,
                                         const float* relu_src, float* dx) {
    using namespace vc;
    VC_CHECK_ARG(dy && wpt && dx, "null pointer");
    VC_CHECK_ARG(waligned16(dy) && waligned16(wpt) && waligned16(dx) && waligned16(relu_src), "pointers must be 16-byte aligned");
    const int per = wino_images_per_launch(B, H, W, Cin, Cout);
    VC_CHECK_ARG(per > 0 && vc_conv3x3_wino_supported(B, H, W, Cin, Cout, 1), "unsupported shape (vc_conv3x3_wino_supported)");
    for (int b0 = 0; b0 < B; b0 += per) {
        const int nb = B - b0 < per ? B - b0 : per;
        const int rc = wino2_launch((hipStream_t)stream, W2_DGRAD, nb, H, W, Cout, Cin, dy + (size_t)b0 * H * W * Cout, wpt, dx + (size_t)b0 * H * W * Cin,
                                    relu_src ? relu_src + (size_t)b0 * H * W * Cin : nullptr, nullptr, nullptr, 0);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int vc_conv3x3_wino_dgrad_bits_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* dy, const float* wpt,
                                              const uint32_t* mask_bits, float* dx) {
    using namespace vc;
    VC_CHECK_ARG(dy && wpt && dx && mask_bits, "null pointer");
    VC_CHECK_ARG(waligned16(dy) && waligned16(wpt) && waligned16(dx) && waligned16(mask_bits), "pointers must be 16-byte aligned");
    VC_CHECK_ARG(wino_images_per_launch(B, H, W, Cin, Cout) >= B && vc_conv3x3_wino_supported(B, H, W, Cin, Cout, 1),
                 "unsupported shape, or more images than one launch takes (vc_conv3x3_wino_single_launch_supported)");
    return wino2_launch((hipStream_t)stream, W2_DGRAD, B, H, W, Cout, Cin, dy, wpt, dx, nullptr, nullptr, const_cast<uint32_t*>(mask_bits), 0);
}
