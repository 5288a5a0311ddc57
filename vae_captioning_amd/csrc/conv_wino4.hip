// 3x3 / stride 1 / SAME convolution by Winograd's F(4x4, 3x3) for gfx950: forward and data gradient of
// utils/image_embeddings.py:36-212 in fp32 with FOUR times fewer multiplications than the direct form (F(2x2,3x3), conv_wino.hip: 2.25).
//
//   Y = A^T [ sum_c (G g_c G^T) (.) (B^T d_c B) ] A     per 4 x 4 output tile: d = its 6 x 6 input patch, g = the 3 x 3 filter
//
// Since round 4 every 3x3 layer of VGG16 behind conv1_1 runs here (vc_conv3x3_wino4_preferred; profiles/r04_wino4_layers.txt).
// Why this kernel looks the way it does (tools/probes/mfma16_f43.hip, mfma_specialised.hip; HISTORY.md section 4g): on gfx950 the fp32
// MFMA rate equals the fp32 vector rate and on one SIMD a VALU instruction and the matrix pipe do NOT overlap -- every VALU operation
// costs 4-8 cycles of matrix time -- so a Winograd kernel's speed is its instruction count per MFMA.  F(4x4,3x3) spends 36 / 16 = 2.25
// MFMAs per output where F(2x2,3x3) spends 4, but its input transform is heavier (144 operations per tile and channel against 32);
// the kernel is built around that count: 12-operation 1-D transforms (FMA with the constants 4, 5, 2), a lane owns ONE tile and ONE
// input channel of a four-channel phase, and (round 4) the TWO WAVES OF A BLOCK SPLIT THE 36 POSITIONS: wave `hf` takes the horizontal
// indices v = 3 hf .. 3 hf + 2, runs half of the horizontal pass (6 operations per patch row) and three column transforms -- 72
// operations per phase instead of 144 --, multiplies its eighteen positions against BOTH groups of sixteen output channels (the same 36
// MFMAs per phase, one B operand feeding two MFMAs), and the partial outputs meet once, in the epilogue.
//
// Structure: workgroup = four waves = TWO blocks of sixteen 4 x 4 tiles x 32 output channels, wave = (block, position half); 36
// accumulators of four registers per wave, two workgroups per CU.  A phase = four input channels = ONE k-step.  LDS per phase: the
// blocks' patches as four channel planes and the phase's transformed weights [group 2][position / 4][g 4][n 16][position % 4] (a lane's
// fragment of four positions = one ds_read_b128, 1 KB contiguous per wave); both double-buffered.  A block is either a 4 x 4 SQUARE of
// tiles with a shared 18 x 18 halo patch ([g 4][row 18][pixel, pitch 20]: a lane's patch row = ds_read_b128 + ds_read_b64) or, where
// squares would leave slots empty (the 56- and 28-wide layers), sixteen CONSECUTIVE tiles in raster order with a 6 x 6 patch each
// (template flag LT below).  Epilogue: A^T over u in registers, the wave's three terms of the sum over v, one exchange of sixteen
// float4 per lane through the LDS, then wave `hf` holds the finished outputs of channel group hf -- bias (it rode in the accumulator
// of position (1, 1)), ReLU, mask bits, 2 x 2 max-pool + routing codes are register math; the outputs themselves pass through the LDS once
// more (the wave's own 16 KB, no barrier between its write and its read) so that a store instruction writes pixel ROWS of the C4 planes.
// Rounding: the transforms' constants (4, 5, 8; 1/4, 1/6, 1/24 in the weights) cost about a decimal digit against F(2x2,3x3):
// ~1e-5 of the tensor maximum (tests/test_gpu_conv_wino4.py against the fp64 oracle).
#include <stdlib.h>
#include <type_traits>
#include "conv_wino.h"

// `make wino4abl W4FLAGS=-DW4_ABL=n` builds this file with W4_ABL = a bit mask that REMOVES parts of the main loop (results are then wrong; timing only):
// 1 transform arithmetic, 2 patch-row reads, 4 weight-fragment reads, 8 staging (global loads + LDS writes), 16 barriers, 32 the patch loads wrapped into a cache-resident 1 MB window (same pattern), 64 no output stores, 128 no exchange of the partial outputs (NOT a valid ablation: the compiler then drops the MFMAs of the unsent channel group), 256 the output stores wrapped into a cache-resident 1 MB window, 512 no global loads of the staging (its LDS writes stay), 1024 no LDS writes of the staging (its loads stay), 2048 (MODE 2) no loads of the transformed input
#ifndef W4_ABL
#define W4_ABL 0
#endif
namespace vc {

enum { W4_FWD = 0, W4_DGRAD = 1 };
constexpr int W4_PITCH = 20;                 // floats per patch row (18 pixels + 2)
constexpr int W4_PLANE = 384;                // floats per channel plane: 18 x 20 = 360, padded to 6 x 64 (bank-quad aligned planes)
constexpr int W4_BLKF = 4 * W4_PLANE;        // one block's patch of a phase: four channel planes
constexpr int W4_NBLK = 2;                   // blocks per workgroup
constexpr int W4_PBUF = W4_NBLK * W4_BLKF;   // 12 KB
constexpr int W4_WGRP = 9 * 256;             // floats of one channel group's weights of a phase: [pq 9][g 4][n 16][pp 4]
constexpr int W4_WBUF = 2 * W4_WGRP;         // 18 KB
constexpr int W4_POFF = 2 * W4_WBUF;         // LDS: two weight buffers, then two patch buffers
constexpr int W4_DUMP = W4_POFF + 360;       // plane padding of block 0: where the staging slots past the data write
constexpr int W4_MAIN_BYTES = (2 * W4_WBUF + 2 * W4_PBUF) * 4;     // 61 440: the main loop's buffers
constexpr int WINO4_LDS_BYTES = 4 * 16 * 64 * 16;                  // 65 536: the epilogue's exchange of partial outputs (four waves x 16 float4 per lane); two workgroups per CU
static_assert(W4_MAIN_BYTES <= WINO4_LDS_BYTES, "LDS");
// LINEAR TILES (template flag LT): a block = sixteen CONSECUTIVE 4 x 4 tiles of the launch in raster order (image, tile row, tile
// column) instead of a 4 x 4 square of them.  On the 56- and 28-wide layers (14 x 14 and 7 x 7 tiles per image) square blocks leave 23 %
// of their tile slots -- and of the MFMAs -- outside the image; linear blocks fill every slot (196 and 49 tiles per image, 64 images:
// whole blocks).  The price is the patch: every tile stages its own 6 x 6 pixels (576 per block against the shared 18 x 18 = 324), as
// four planes [tile 16][row 6][pixel 6] of 578 floats (pitch 2 mod 4: the two lane groups a ds_read_b64 serves together then hit
// different banks); a patch row = three ds_read_b64.  Everything behind the patch -- transforms, MFMAs, weights, epilogue -- is the same.
constexpr int W4L_PLANE = 578;
constexpr int W4L_BLKF = 4 * W4L_PLANE;
constexpr int W4L_PBUF = W4_NBLK * W4L_BLKF;                       // 18.5 KB
constexpr int W4L_DUMP = W4_POFF + 2 * W4L_PBUF;                   // sixteen floats behind the patch buffers
constexpr int W4L_NPIX = W4_NBLK * 576;
constexpr int WINO4L_LDS_BYTES = (2 * W4_WBUF + 2 * W4L_PBUF + 16) * 4;   // 73 920: two workgroups per CU
static_assert(WINO4L_LDS_BYTES >= WINO4_LDS_BYTES, "the exchange area");
constexpr int W4_WPH = W4_WBUF * 4;          // bytes of a phase's packed weights per 32-channel tile
constexpr int W4_NPIX = W4_NBLK * 324;       // patch pixels of a workgroup

struct Wino4Geom {
    int B, H, W, C, N;
    int bx_n, by_n, blocks_img, nblocks;     // blocks of 16 x 16 output pixels (linear tiles: blocks_img unused, nblocks = ceil(tiles / 16))
    unsigned m_blocks_img, m_bx_n;
    int lt, tw, tiles_img, ntiles_all;       // linear tiles: tiles per image row / per image / in the launch
    unsigned m_tiles_img, m_tw;
};

struct Wino4Args {
    Wino4Geom g;
    const float* x;      // [B][C/4][H][W][4] (the C4 activation layout, vaecap.h)
    const float* wp;     // packed [N/32][C/4][group 2][pq 9][g 4][n 16][pp 4]
    float* out;          // [B][N/4][H][W][4]
    const float* aux;    // fwd: bias [N] or null; dgrad: ReLU source, layout of out, or null
    float* pool;         // fwd: also max_pool2x2(out) [B][N/4][H/2][W/2][4] (null: none)
    unsigned* pbits;     // fwd + pool: MaxPoolGrad routing codes, the format of conv_wino.hip ([B][N/4][H/2][W/2] half-words, 4 bits per pooled element:
                         // position of the first maximum of its window | 4 if it is > 0); a lane owns four channels = one half-word
    unsigned* mask;      // [workgroups][256 threads][2]: (out > 0) of each lane's 4 x 4 pixels x 4 channels as 64 bits -- written by the forward
                         // (null: not wanted), read by the data gradient of the NEXT layer instead of the float source (same shape => same lanes)
    int relu;
    int tiles_n, ntiles, nphases;
    int tg, per_chunk;   // workgroup order: [chunk of tg channel tiles][block pair][channel tile of the chunk] (tg divides tiles_n; per_chunk = block pairs * tg)
};

// one 1-D input transform B^T (6 -> 6) in twelve operations:
//   o0 = 4 d0 - 5 d2 + d4      o1 = (d4 - 4 d2) + (d3 - 4 d1)     o2 = (d4 - 4 d2) - (d3 - 4 d1)
//   o5 = 4 d1 - 5 d3 + d5      o3 = (d4 - d2) + 2 (d3 - d1)       o4 = (d4 - d2) - 2 (d3 - d1)
// w4_bstep_v (columns, into the MFMA operands), step by step so that the caller can spread the steps over MFMA gaps: output o_j is not
// written before step 2 j -- a column transform running ONE step per gap beside the twelve MFMAs of the previous column (six positions
// x two channel groups) writes o_j over the operand that the MFMAs of position j have consumed (single-buffered U).
__device__ __forceinline__ void w4_bstep_v(int k, const float& d0, const float& d1, const float& d2, const float& d3, const float& d4, const float& d5,
                                           float* o, float* t) {
    if (k == 0) t[0] = __builtin_fmaf(-4.f, d2, d4);
    if (k == 1) t[1] = __builtin_fmaf(-4.f, d1, d3);
    if (k == 2) t[4] = __builtin_fmaf(-5.f, d2, d4);
    if (k == 3) o[0] = __builtin_fmaf(4.f, d0, t[4]);
    if (k == 4) t[2] = d4 - d2;
    if (k == 5) o[1] = t[0] + t[1];
    if (k == 6) t[3] = d3 - d1;
    if (k == 7) t[4] = __builtin_fmaf(-5.f, d3, d5);
    if (k == 8) o[2] = t[0] - t[1];
    if (k == 9) o[3] = __builtin_fmaf(2.f, t[3], t[2]);
    if (k == 10) o[4] = __builtin_fmaf(-2.f, t[3], t[2]);
    if (k == 11) o[5] = __builtin_fmaf(4.f, d1, t[4]);
}
// w4_hstep<HF> (rows, from the raw pixels): the HALF of the row transform a wave needs -- HF = 0: outputs o0 o1 o2 (horizontal indices
// v = 0, 1, 2), HF = 1: o3 o4 o5 (v = 3, 4, 5) -- in six operations; the raw inputs are dead after step 3.
template <int HF>
__device__ __forceinline__ void w4_hstep(int k, const float* d, float* o, float* t) {
    if (HF == 0) {
        if (k == 0) t[0] = __builtin_fmaf(-4.f, d[2], d[4]);
        if (k == 1) t[1] = __builtin_fmaf(-4.f, d[1], d[3]);
        if (k == 2) t[2] = __builtin_fmaf(-5.f, d[2], d[4]);
        if (k == 3) o[0] = __builtin_fmaf(4.f, d[0], t[2]);
        if (k == 4) o[1] = t[0] + t[1];
        if (k == 5) o[2] = t[0] - t[1];
    } else {
        if (k == 0) t[0] = d[4] - d[2];
        if (k == 1) t[1] = d[3] - d[1];
        if (k == 2) t[2] = __builtin_fmaf(-5.f, d[3], d[5]);
        if (k == 3) o[2] = __builtin_fmaf(4.f, d[1], t[2]);
        if (k == 4) o[0] = __builtin_fmaf(2.f, t[1], t[0]);
        if (k == 5) o[1] = __builtin_fmaf(-2.f, t[1], t[0]);
    }
}

// MODE 0: square blocks, 1: linear tile blocks, 2 (W4_VIN): linear tile blocks whose input arrives TRANSFORMED -- a.x = V, the
// B^T d B of every (tile, input channel) in the B-operand order of the MFMAs, written once per layer and pass by wino4_xform_kernel
// below.  The main loop then has no patch staging, no patch LDS traffic and no transform arithmetic: a lane's B operands of a phase are
// five 16-byte loads straight into registers (1 KB contiguous per wave and load), everything else -- weights through the LDS, the 36
// MFMAs, the epilogue with its bias / ReLU / mask bits / pool / routing codes -- is the code of MODE 1.  Why: on gfx950 the f32 MFMA
// shares its issue port with the f32 VALU (HISTORY.md 4g) and the fused kernel's 2.3-3.9 VALU + 1.0-1.4 LDS instructions per MFMA are
// what holds it at 46-57 % MFMA-busy; here the input transform is paid ONCE per input channel instead of once per 32-output-channel
// tile (16 x on the 512-channel layers) and at HBM speed (V = 2.25 x the activation bytes, written + read once).
enum { W4_SQUARE = 0, W4_LINEAR = 1, W4_VIN = 2 };
template <int KIND, bool POOL, int MODE>
__global__ __launch_bounds__(256, 2) void conv_wino4_kernel(Wino4Args a) {
    constexpr bool LT = MODE != W4_SQUARE, VIN = MODE == W4_VIN;
    constexpr int NPS = VIN ? 1 : LT ? 5 : 3;                          // patch slots per thread and phase
    constexpr int PBUF = LT ? W4L_PBUF : W4_PBUF, PLANE = LT ? W4L_PLANE : W4_PLANE;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const Wino4Geom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lj = lane & 15, lg = lane >> 4;
    // wave = 2 block + half: the two waves of a block split the 36 POSITIONS (half 0: horizontal indices v = 0..2, half 1: v = 3..5) and
    // each multiplies its eighteen against BOTH channel groups; the epilogue exchanges partial outputs and wave `hf` finishes channel group hf
    const int wb = wave >> 1, hf = wave & 1;
    const int id = xcd_remap(blockIdx.x, a.ntiles);
    const int chunk = id / a.per_chunk, cr = id - chunk * a.per_chunk;
    const int tm = cr / a.tg, nt = chunk * a.tg + (cr - tm * a.tg);
    const int C = g.C, N = g.N;
    const int nc0 = nt * 32 + hf * 16 + 4 * lg;   // the four output channels this lane FINISHES

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, VIN ? (int)((long)g.nblocks * C * 2304) : (int)((long)g.B * g.H * g.W * C * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.wp, 0, (int)(36L * C * N * 4), 0x00020000);

    // patch slots: slot s = tid + 256 j; one float4 = the phase's four channels of a pixel.  Square blocks: s < 648 = (block s / 324, patch
    // pixel s % 324 of its 18 x 18); linear tiles: s < 1152 = (block s / 576, tile (s % 576) / 36, pixel of its 6 x 6)
    unsigned voff[NPS];
    int pst[NPS];
#pragma unroll
    for (int j = 0; j < NPS; ++j) {
        const unsigned s = (unsigned)tid + 256u * j;
        if constexpr (VIN) {   // no patches: the wave's B operands come from V[block][phase][pq 9][g 4][tile 16][pp 4] (voff: the lane's 16 bytes of fragment 0)
            voff[j] = (((unsigned)(tm * W4_NBLK + (wave >> 1)) * (unsigned)a.nphases * 9u + 4u * (unsigned)(wave & 1)) * 256u + (unsigned)lane * 4u) * 4u;
            pst[j] = 0;
        } else if constexpr (LT) {
            const unsigned blk = s >= 576u ? 1u : 0u, rem = s - blk * 576u;
            const unsigned tl = rem / 36u, pp = rem - tl * 36u, py = pp / 6u, px = pp - py * 6u;
            const unsigned T = ((unsigned)tm * W4_NBLK + blk) * 16u + tl;
            const bool live = s < (unsigned)W4L_NPIX && T < (unsigned)g.ntiles_all;
            const unsigned Tc = live ? T : 0u;
            const unsigned b = wino_div(Tc, g.m_tiles_img), r2 = Tc - b * (unsigned)g.tiles_img;
            const unsigned tyy = wino_div(r2, g.m_tw), txx = r2 - tyy * (unsigned)g.tw;
            const int y = (int)(tyy * 4u + py) - 1, x = (int)(txx * 4u + px) - 1;
            const bool ok = live && (unsigned)y < (unsigned)g.H && (unsigned)x < (unsigned)g.W;
            voff[j] = ok ? (((b * (unsigned)(C >> 2) * (unsigned)g.H + (unsigned)y) * (unsigned)g.W + (unsigned)x)) * 16u : WOOB;
            pst[j] = s < (unsigned)W4L_NPIX ? W4_POFF + (int)blk * W4L_BLKF + (int)(tl * 36u + pp) : W4L_DUMP + (tid & 15);
        } else {
            const unsigned blk = s >= 324u ? 1u : 0u, pix = s - blk * 324u;
            const unsigned py = pix / 18u, px = pix - py * 18u;
            const unsigned gb = (unsigned)tm * W4_NBLK + blk;
            const bool live = s < (unsigned)W4_NPIX && gb < (unsigned)g.nblocks;
            const unsigned gbc = live ? gb : 0u;
            const unsigned b = wino_div(gbc, g.m_blocks_img), rem = gbc - b * (unsigned)g.blocks_img;
            const unsigned by = wino_div(rem, g.m_bx_n), bx = rem - by * (unsigned)g.bx_n;
            const int y = (int)(by * 16u + py) - 1, x = (int)(bx * 16u + px) - 1;
            const bool ok = live && (unsigned)y < (unsigned)g.H && (unsigned)x < (unsigned)g.W;
            // C4 layout [B][C/4][H][W][4]: the pixel's four channels of phase h are 16 bytes at channel plane h (soffset h * plane bytes);
            // consecutive lanes = consecutive pixels of a patch row = consecutive 16-byte pieces (288-byte runs)
            voff[j] = ok ? (((b * (unsigned)(C >> 2) * (unsigned)g.H + (unsigned)y) * (unsigned)g.W + (unsigned)x)) * 16u : WOOB;
            pst[j] = s < (unsigned)W4_NPIX ? W4_POFF + (int)blk * W4_BLKF + (int)py * W4_PITCH + (int)px : W4_DUMP + (tid & 15);
        }
        if ((W4_ABL & 32) && voff[j] != WOOB) voff[j] &= 0xfffffu;   // timing only: the same access pattern inside a 1 MB window (cache-resident patches)
    }
    // weight slots: piece i of the phase's 18 KB = float4 tid + 256 i < 1152
    const unsigned vsrc = (unsigned)nt * (unsigned)a.nphases * (unsigned)W4_WPH + (unsigned)tid * 16u;
    const bool w4ok = tid < 128;   // piece 4: only the first half of the workgroup has one

    const int ty = lj >> 2, tx = lj & 3;
    // + buffer + r * (row pitch): the lane's patch row r in plane lg
    const int rbase = LT ? W4_POFF + wb * W4L_BLKF + lg * W4L_PLANE + lj * 36 : W4_POFF + wb * W4_BLKF + lg * W4_PLANE + (4 * ty) * W4_PITCH + 4 * tx;
    const int vbase = (lg * 16 + lj) * 4;                                                      // + group * WGRP + wq * WBUF + pq * 256

    // M_p[channel 32 nt + 16 gi + 4 lg + r][tile lj] for the wave's positions p = 18 hf + pos, pos = 6 vl + u (u: vertical index, vl: the
    // wave's horizontal index v - 3 hf) and BOTH channel groups gi
    f32x4 acc[2][18];
#pragma unroll
    for (int gi = 0; gi < 2; ++gi)
#pragma unroll
        for (int p = 0; p < 18; ++p) acc[gi][p] = f32x4{0.f, 0.f, 0.f, 0.f};
    // forward: the bias rides in the accumulator of position (1, 1) (half 0): column 1 of A^T is (1, 1, 1, 1), so A^T M A adds M_(1,1) to
    // all sixteen outputs of the tile
    if (KIND == W4_FWD && a.aux && hf == 0) {
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
            const float4 bv = *reinterpret_cast<const float4*>(a.aux + nt * 32 + gi * 16 + 4 * lg);
            acc[gi][7] = f32x4{bv.x, bv.y, bv.z, bv.w};
        }
    }

    // this lane's 4 x 4 output pixels
    const int gbo = tm * W4_NBLK + wb;
    bool blk_ok;
    int ob, y0, x0;
    if constexpr (LT) {
        const int T = gbo * 16 + lj;
        blk_ok = T < g.ntiles_all;
        const int Tc = blk_ok ? T : 0;
        ob = (int)wino_div((unsigned)Tc, g.m_tiles_img);
        const int r2 = Tc - ob * g.tiles_img, tyy = (int)wino_div((unsigned)r2, g.m_tw);
        y0 = 4 * tyy; x0 = 4 * (r2 - tyy * g.tw);
    } else {
        blk_ok = gbo < g.nblocks;
        const int gboc = blk_ok ? gbo : 0;
        ob = (int)wino_div((unsigned)gboc, g.m_blocks_img);
        const int orem = gboc - ob * g.blocks_img;
        const int oby = (int)wino_div((unsigned)orem, g.m_bx_n), obx = orem - oby * g.bx_n;
        y0 = oby * 16 + 4 * ty; x0 = obx * 16 + 4 * tx;
    }
    // output in the C4 layout: this lane's channel quad nc0 / 4 is one plane; a tile row = 64 consecutive bytes
    const long p00 = (((long)(ob * (N >> 2) + (nc0 >> 2)) * g.H + y0) * g.W + x0) * 4;
    // data gradient: the ReLU mask of the lane's outputs as 64 bits (bit 16 aa + 4 bb + c), from the producer's forward (one 8-byte
    // load) or from the float activation (sixteen loads that overlap the first patch loads)
    unsigned mb0 = 0xffffffffu, mb1 = 0xffffffffu;
    if (KIND == W4_DGRAD && a.mask) {
        const uint2 m = *reinterpret_cast<const uint2*>(a.mask + ((size_t)(tm * a.tiles_n + nt) * 256 + tid) * 2);   // by (block pair, channel tile): independent of the workgroup order
        mb0 = m.x; mb1 = m.y;
    } else if (KIND == W4_DGRAD && a.aux) {
        mb0 = mb1 = 0u;
#pragma unroll
        for (int aa = 0; aa < 4; ++aa)
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
                const bool ok = blk_ok && y0 + aa < g.H && x0 + bb < g.W;
                const float4 m = ok ? *reinterpret_cast<const float4*>(a.aux + p00 + (aa * g.W + bb) * 4) : f4zero();
                const unsigned bits = (m.x > 0.f ? 1u : 0u) | (m.y > 0.f ? 2u : 0u) | (m.z > 0.f ? 4u : 0u) | (m.w > 0.f ? 8u : 0u);
                if (aa < 2) mb0 |= bits << (16 * aa + 4 * bb);
                else mb1 |= bits << (16 * (aa - 2) + 4 * bb);
            }
    }

    const unsigned plane_b = (unsigned)g.H * (unsigned)g.W * 16u;   // bytes of one channel-quad plane of an image
    float4 Y[4][4];   // the lane's finished outputs (channel group hf)

    // ---- main loop + partial output transform, compiled once per position half
    auto body = [&](auto hfc) {
        constexpr int HF = decltype(hfc)::value;
        float4 st[5];               // staging registers: the patch slots, then the five weight pieces
        float HA[6][3], HB[6][3];   // [patch row r][local horizontal index vl]: rows transformed horizontally (the wave's half of B^T over a row)
        float U[6];                 // one column of B^T d B = the B operands of twelve MFMAs
        float4 vf[2][2];            // weight fragments [slot][channel group]: fragment f = positions 4 (4 HF + f) .. + 3 of the packed order
        float rr[2][6];             // raw patch rows in flight (row r in set r & 1)
        float tv[5], th[3];
        if (W4_ABL) {   // ablated builds read registers nobody wrote: give them values
#pragma unroll
            for (int i = 0; i < 5; ++i) st[i] = make_float4(1.f, 2.f, 3.f, 4.f);
#pragma unroll
            for (int i = 0; i < 2; ++i) { vf[i][0] = make_float4(1.f + lane, 2.f, 3.f, 4.f); vf[i][1] = make_float4(2.f + lane, 2.f, 3.f, 4.f); }
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) { HA[r][c] = 1.f + lane + r; HB[r][c] = 2.f + lane + c; rr[c & 1][r] = 0.5f * lane; U[r] = 1.f * lane; }
        }
        auto pload = [&](int i, int hp) { if (!(W4_ABL & (8 | 512))) st[i] = wbufload(rx, voff[i], (unsigned)((W4_ABL & 32) ? (hp & 1) : hp) * plane_b); };
        auto pput = [&](const float4& v, int i, int pq) {   // patch slot i into patch buffer pq: one pixel, four channel planes
            float* d = &smem[pst[i] + pq * PBUF];
            d[0] = v.x; d[PLANE] = v.y; d[2 * PLANE] = v.z; d[3 * PLANE] = v.w;
        };
        auto wput = [&](const float4& v, int i, int wq) {   // weight piece i into weight buffer wq
            const int dst = (i == 4 && !w4ok) ? (VIN ? W4_POFF : LT ? W4L_DUMP : W4_DUMP) + 4 * (tid & 3) : wq * W4_WBUF + (tid + 256 * i) * 4;
            *reinterpret_cast<float4*>(&smem[dst]) = v;
        };
        auto keep = [&](const float4& v) { asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w)); };   // (ablation 1024: the loads stay, their LDS writes go)
        auto pstore = [&](int i, int pq) { if (W4_ABL & 1024) keep(st[i]); else if (!(W4_ABL & 8)) pput(st[i], i, pq); };
        auto wload = [&](int si, int i, int hp) { if (!(W4_ABL & (8 | 512))) st[si] = wbufload(rw, (i == 4 && !w4ok) ? WOOB : vsrc + (unsigned)i * 4096u, (unsigned)hp * (unsigned)W4_WPH); };
        auto wstore = [&](int si, int i, int wq) { if (W4_ABL & 1024) keep(st[si]); else if (!(W4_ABL & 8)) wput(st[si], i, wq); };
        auto rdrow = [&](int off, int r) {   // off: float offset of the patch buffer whose rows are read; row r lands in set r & 1
            if (W4_ABL & 2) return;
            float* d = rr[r & 1];
            if constexpr (LT) {
                const float2 v0 = *reinterpret_cast<const float2*>(&smem[rbase + off + r * 6]);
                const float2 v1 = *reinterpret_cast<const float2*>(&smem[rbase + off + r * 6 + 2]);
                const float2 v2 = *reinterpret_cast<const float2*>(&smem[rbase + off + r * 6 + 4]);
                d[0] = v0.x; d[1] = v0.y; d[2] = v1.x; d[3] = v1.y; d[4] = v2.x; d[5] = v2.y;
            } else {
                const float4 v = *reinterpret_cast<const float4*>(&smem[rbase + off + r * W4_PITCH]);
                const float2 w = *reinterpret_cast<const float2*>(&smem[rbase + off + r * W4_PITCH + 4]);
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w; d[4] = w.x; d[5] = w.y;
            }
        };
        auto rdfrag = [&](int f, int slot, int wq) {   // fragment f of weight buffer wq, both channel groups
            if (W4_ABL & 4) return;
#pragma unroll
            for (int gi = 0; gi < 2; ++gi)
                vf[slot][gi] = *reinterpret_cast<const float4*>(&smem[vbase + gi * W4_WGRP + wq * W4_WBUF + (4 * HF + f) * 256]);
        };
#define WSB() __builtin_amdgcn_sched_barrier(0)

        // ---- MODE 2 (pre-transformed input): the same 36 MFMAs with their B operands in registers.  Vc / Vn: this / the next phase's five
        // fragments (fragment f = positions 4 (4 HF + f) .. + 3, the weights' numbering); the loads of phase h + 1 are issued at gaps 0 .. 4 of
        // phase h, AHEAD of the weight pieces (gaps 5 .. 9 -> LDS at 20 .. 24): vmcnt retires in order, so by the time the weights are
        // written both sets have arrived and nothing waits at the top of the next phase.
        if constexpr (VIN) {
            float4 Va[5], Vb[5];
            const unsigned vstep = 9u * 1024u;   // bytes of one phase of a block
            auto vload = [&](float4 (&Vn)[5], int i, int hp) { if (!(W4_ABL & 2048)) Vn[i] = wbufload(rx, voff[0] + (unsigned)i * 1024u, (unsigned)((W4_ABL & 32) ? (hp & 1) : hp) * vstep); };
            if (W4_ABL & 2048) {
#pragma unroll
                for (int i = 0; i < 5; ++i) Va[i] = Vb[i] = make_float4(1.f + lane, 2.f, 3.f, 0.5f * i);
            }
            auto vphase = [&](auto qc, float4 (&Vc)[5], float4 (&Vn)[5], bool nxt, int h) {
                constexpr int q = decltype(qc)::value;
#pragma unroll
                for (int m = 0; m < 36; ++m) {
                    const int pos = m >> 1, gi = m & 1;
                    {
                        const int pp = 18 * HF + pos, f = (pp >> 2) - 4 * HF;
                        const float4& fr = vf[(f + q) & 1][gi];
                        const float av = (pp & 3) == 0 ? fr.x : (pp & 3) == 1 ? fr.y : (pp & 3) == 2 ? fr.z : fr.w;
                        const float4& vb = Vc[f];
                        const float bv = (pp & 3) == 0 ? vb.x : (pp & 3) == 1 ? vb.y : (pp & 3) == 2 ? vb.z : vb.w;
                        acc[gi][pos] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[gi][pos], 0, 0, 0);
                    }
                    WSB();
                    {
                        constexpr int first = HF ? 0 : 1;
                        if (m >= first && m <= first + 24 && ((m - first) & 7) == 0) {
                            const int f = (m - first) / 8 + 1;
                            rdfrag(f, (f + q) & 1, q);
                        }
                    }
                    if (nxt && m == 33) rdfrag(0, (q ^ 1) & 1, q ^ 1);
                    if (m < 5 && nxt) vload(Vn, m, h + 1);
                    if (m >= 5 && m < 10 && nxt) wload(m - 5, m - 5, h + 1);
                    if (m >= 20 && m < 25 && nxt) wstore(m - 20, m - 20, q ^ 1);
                    if (m == 28 && nxt && !(W4_ABL & 16)) __syncthreads();
                    WSB();
                }
            };
            {   // prologue: phase 0's weights into the LDS, its B operands into registers
                float4 pr[5];
#pragma unroll
                for (int i = 0; i < 5; ++i) pr[i] = wbufload(rw, (i == 4 && !w4ok) ? WOOB : vsrc + (unsigned)i * 4096u, 0u);
#pragma unroll
                for (int i = 0; i < 5; ++i) vload(Va, i, 0);
#pragma unroll
                for (int i = 0; i < 5; ++i) wput(pr[i], i, 0);
            }
            __syncthreads();
            rdfrag(0, 0, 0);
            WSB();
            using Q0 = std::integral_constant<int, 0>;
            using Q1 = std::integral_constant<int, 1>;
            int h = 0;
            for (; h + 2 < a.nphases; h += 2) {
                vphase(Q0{}, Va, Vb, true, h);
                vphase(Q1{}, Vb, Va, true, h + 1);
            }
            vphase(Q0{}, Va, Vb, true, h);
            vphase(Q1{}, Vb, Va, false, h + 1);
        } else {
        // One phase (index h, parity q) = 36 MFMAs: the wave's eighteen positions x two channel groups, column by column of Hc (twelve
        // MFMAs per column).  Beside them the next phase's rows are read (patch buffer q ^ 1) and transformed into Hn (gaps 1 .. 24), the
        // next column is transformed vertically (one step per gap), the next phase's weights are staged into weight buffer q ^ 1 and the
        // patches of the phase after it into patch buffer q; the barrier at gap 28 publishes the writes and closes this phase's reads.
        auto phase = [&](auto qc, float (&Hc)[6][3], float (&Hn)[6][3], bool stP, bool stW, int h) {
            constexpr int q = decltype(qc)::value;
            const bool nxt = stW;   // a next phase exists exactly when its weights are still to be staged
            const bool rd0ok = stP;
            constexpr int rdo = (q ^ 1) * PBUF, rd0 = q * PBUF;
#pragma unroll
            for (int m = 0; m < 36; ++m) {
                const int pos = m >> 1, gi = m & 1, vl = pos / 6, u = pos % 6;
                {   // packed position p = 18 HF + pos lives in fragment f = p / 4 - 4 HF, slot (f + q) & 1 (five fragments per phase: the parity flips with the phase)
                    const int pp = 18 * HF + pos, f = (pp >> 2) - 4 * HF;
                    const float4& fr = vf[(f + q) & 1][gi];
                    const float av = (pp & 3) == 0 ? fr.x : (pp & 3) == 1 ? fr.y : (pp & 3) == 2 ? fr.z : fr.w;
                    acc[gi][pos] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, U[u], acc[gi][pos], 0, 0, 0);
                }
                WSB();
                // (A) vertical transform of the next column, one step per gap, written over the operands just consumed
                if (W4_ABL & 1) {
                } else if (vl < 2) {
                    w4_bstep_v(m % 12, Hc[0][vl + 1], Hc[1][vl + 1], Hc[2][vl + 1], Hc[3][vl + 1], Hc[4][vl + 1], Hc[5][vl + 1], U, tv);
                } else if (nxt) {
                    w4_bstep_v(m % 12, Hn[0][0], Hn[1][0], Hn[2][0], Hn[3][0], Hn[4][0], Hn[5][0], U, tv);
                }
                // (B) horizontal transform of the next phase's rows: 36 steps over gaps 1 .. 24 (row r: gaps 4 r + 1 .. 4 r + 4, done
                // before its column 0 is wanted at gap 24)
                if (nxt && m >= 1 && m <= 24 && !(W4_ABL & 1)) {
#pragma unroll
                    for (int S = (m - 1) * 3 / 2; S < m * 3 / 2; ++S) w4_hstep<HF>(S % 6, rr[(S / 6) & 1], Hn[S / 6], th);
                }
                // (C) row reads: rows 0 and 1 were read at the end of the previous phase; the raw row r is dead after gap 4 r + 3, row
                // r + 2 is read into its registers at gap 4 r + 4; rows 0, 1 of the phase after the next at gaps 34, 35 (behind the barrier)
                if (nxt && m >= 4 && m <= 16 && (m & 3) == 0) rdrow(rdo, m / 4 + 1);
                if (rd0ok && m == 34) rdrow(rd0, 0);
                if (rd0ok && m == 35) rdrow(rd0, 1);
                // (D) weight fragments: fragment f >= 1 of this phase once the slot's previous fragment is consumed; fragment 0 of the
                // next phase at gap 33.  Half 0 uses fragment f at MFMAs 8 f .. 8 f + 7, half 1 at 8 f - 4 .. 8 f + 3.
                {
                    constexpr int first = HF ? 0 : 1;
                    if (m >= first && m <= first + 24 && ((m - first) & 7) == 0) {
                        const int f = (m - first) / 8 + 1;
                        rdfrag(f, (f + q) & 1, q);
                    }
                }
                if (nxt && m == 33) rdfrag(0, (q ^ 1) & 1, q ^ 1);
                // (E) staging through shared registers.  Square blocks: group A = the three patch slots of phase h + 2 + weight piece 0 of
                // phase h + 1 (loaded at gaps 0 .. 3, written at 12 .. 15), group B = weight pieces 1 .. 4 (16 .. 19 -> 24 .. 27).  Linear tiles:
                // the five patch slots (0 .. 4 -> 11 .. 15), then the five weight pieces (16 .. 20 -> 23 .. 27).
                if constexpr (!LT) {
                    if (m < 3 && stP) pload(m, h + 2);
                    if (m == 3 && stW) wload(3, 0, h + 1);
                    if (m >= 12 && m < 15 && stP) pstore(m - 12, q);
                    if (m == 15 && stW) wstore(3, 0, q ^ 1);
                    if (m >= 16 && m < 20 && stW) wload(m - 16, m - 15, h + 1);
                    if (m >= 24 && m < 28 && stW) wstore(m - 24, m - 23, q ^ 1);
                } else {   // (own registers for the weight pieces, both groups in flight together: measured, no gain)
                    if (m < 5 && stP) pload(m, h + 2);
                    if (m >= 11 && m < 16 && stP) pstore(m - 11, q);
                    if (m >= 16 && m < 21 && stW) wload(m - 16, m - 16, h + 1);
                    if (m >= 23 && m < 28 && stW) wstore(m - 23, m - 23, q ^ 1);
                }
                // (F)
                if (m == 28 && nxt && !(W4_ABL & 16)) __syncthreads();
                WSB();
            }
        };

        // ---- prologue: phases 0 (patches + weights) and 1 (patches) into the LDS, rows of phase 0 transformed.  All eleven loads are
        // issued before the first store: one memory round trip instead of three (a workgroup of conv1_2 lives for sixteen phases only)
        if (!(W4_ABL & 8)) {
            float4 pr[2 * NPS + 5];
#pragma unroll
            for (int i = 0; i < NPS; ++i) pr[i] = wbufload(rx, voff[i], 0u);
#pragma unroll
            for (int i = 0; i < 5; ++i) pr[NPS + i] = wbufload(rw, (i == 4 && !w4ok) ? WOOB : vsrc + (unsigned)i * 4096u, 0u);
#pragma unroll
            for (int i = 0; i < NPS; ++i) pr[NPS + 5 + i] = wbufload(rx, voff[i], plane_b);
#pragma unroll
            for (int i = 0; i < NPS; ++i) pput(pr[i], i, 0);
#pragma unroll
            for (int i = 0; i < 5; ++i) wput(pr[NPS + i], i, 0);
#pragma unroll
            for (int i = 0; i < NPS; ++i) pput(pr[NPS + 5 + i], i, 1);
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            rdrow(0, r);
#pragma unroll
            for (int k = 0; k < 6; ++k) w4_hstep<HF>(k, rr[r & 1], HA[r], th);
        }
#pragma unroll
        for (int k = 0; k < 12; ++k) w4_bstep_v(k, HA[0][0], HA[1][0], HA[2][0], HA[3][0], HA[4][0], HA[5][0], U, tv);
        rdfrag(0, 0, 0);
        rdrow(PBUF, 0);   // rows 0, 1 of phase 1
        rdrow(PBUF, 1);
        __syncthreads();     // (nobody stages over rows that somebody still reads)
        WSB();

        using Q0 = std::integral_constant<int, 0>;
        using Q1 = std::integral_constant<int, 1>;
        int h = 0;
        for (; h + 2 < a.nphases; h += 2) {
            phase(Q0{}, HA, HB, true, true, h);
            phase(Q1{}, HB, HA, true, true, h + 1);
        }
        phase(Q0{}, HA, HB, false, true, h);
        phase(Q1{}, HB, HA, false, false, h + 1);
        }   // (MODE 0 / 1 main loop)
#undef WSB

        // ---- output transform A^T M A (6 x 6 -> 4 x 4): rows of A^T
        //   y0 = m0 + m1 + m2 + m3 + m4   y1 = (m1 - m2) + 2 (m3 - m4)   y2 = (m1 + m2) + 4 (m3 + m4)   y3 = (m1 - m2) + 8 (m3 - m4) + m5
        // The vertical pass (over u) is whole inside a wave; of the horizontal pass (over v) a wave holds three of the six terms: PARTIAL
        // outputs, summed with the other half's through the LDS -- each wave sends the partials of the channel group it does not finish.
        __syncthreads();   // (the main loop's LDS reads are behind every wave)
        float4* xch = reinterpret_cast<float4*>(smem);
        auto partial = [&](int gi, float4 (&Pq)[4][4]) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float T[4][3];
#pragma unroll
                for (int vl = 0; vl < 3; ++vl) {
                    const float m0 = acc[gi][6 * vl + 0][r], m1 = acc[gi][6 * vl + 1][r], m2 = acc[gi][6 * vl + 2][r], m3 = acc[gi][6 * vl + 3][r],
                                m4 = acc[gi][6 * vl + 4][r], m5 = acc[gi][6 * vl + 5][r];
                    const float s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
                    T[0][vl] = m0 + s1 + s2;
                    T[1][vl] = __builtin_fmaf(2.f, d2, d1);
                    T[2][vl] = __builtin_fmaf(4.f, s2, s1);
                    T[3][vl] = __builtin_fmaf(8.f, d2, d1) + m5;
                }
#pragma unroll
                for (int aa = 0; aa < 4; ++aa) {
                    float o0, o1, o2, o3;
                    if (HF == 0) {   // v = 0, 1, 2: m0, m1, m2 of the row
                        const float s1 = T[aa][1] + T[aa][2], d1 = T[aa][1] - T[aa][2];
                        o0 = T[aa][0] + s1; o1 = d1; o2 = s1; o3 = d1;
                    } else {         // v = 3, 4, 5: m3, m4, m5
                        const float s2 = T[aa][0] + T[aa][1], d2 = T[aa][0] - T[aa][1];
                        o0 = s2; o1 = 2.f * d2; o2 = 4.f * s2; o3 = __builtin_fmaf(8.f, d2, T[aa][2]);
                    }
                    if (r == 0) { Pq[aa][0].x = o0; Pq[aa][1].x = o1; Pq[aa][2].x = o2; Pq[aa][3].x = o3; }
                    if (r == 1) { Pq[aa][0].y = o0; Pq[aa][1].y = o1; Pq[aa][2].y = o2; Pq[aa][3].y = o3; }
                    if (r == 2) { Pq[aa][0].z = o0; Pq[aa][1].z = o1; Pq[aa][2].z = o2; Pq[aa][3].z = o3; }
                    if (r == 3) { Pq[aa][0].w = o0; Pq[aa][1].w = o1; Pq[aa][2].w = o2; Pq[aa][3].w = o3; }
                }
            }
        };
        {
            float4 S[4][4];
            partial(1 - HF, S);
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (!(W4_ABL & 128)) xch[(wave * 16 + k) * 64 + lane] = S[k >> 2][k & 3];
        }
        partial(HF, Y);
        if (!(W4_ABL & 128)) __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (W4_ABL & 128) break;
            const float4 o = xch[((wave ^ 1) * 16 + k) * 64 + lane];
            float4& y = Y[k >> 2][k & 3];
            // (always half 0's partial + half 1's, so that both waves of a block round alike)
            if (HF == 0) { y.x = y.x + o.x; y.y = y.y + o.y; y.z = y.z + o.z; y.w = y.w + o.w; }
            else { y.x = o.x + y.x; y.y = o.y + y.y; y.z = o.z + y.z; y.w = o.w + y.w; }
        }
    };
    if (hf) body(std::integral_constant<int, 1>{});
    else body(std::integral_constant<int, 0>{});

    // ---- epilogue
    unsigned ob0 = 0u, ob1 = 0u;
#pragma unroll
    for (int aa = 0; aa < 4; ++aa)
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            const bool ok = blk_ok && y0 + aa < g.H && x0 + bb < g.W;
            float4& v = Y[aa][bb];
            if (KIND == W4_FWD) {   // (the bias is already in the accumulators)
                if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                if (a.mask) {
                    const unsigned bits = (v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u);
                    if (aa < 2) ob0 |= bits << (16 * aa + 4 * bb);
                    else ob1 |= bits << (16 * (aa - 2) + 4 * bb);
                }
            } else if (a.aux || a.mask) {
                const unsigned mb = (aa < 2 ? mb0 >> (16 * aa + 4 * bb) : mb1 >> (16 * (aa - 2) + 4 * bb));
                if (!(mb & 1u)) v.x = 0.f;
                if (!(mb & 2u)) v.y = 0.f;
                if (!(mb & 4u)) v.z = 0.f;
                if (!(mb & 8u)) v.w = 0.f;
            }
            (void)ok;
        }
    if constexpr (!LT) {
        // Square blocks: the wave's 16 x 16 pixels x four channel quads leave through its 16 KB of the exchange area so that a store
        // instruction writes four 256-byte ROWS (lane = (quad, pixel of the row)) instead of 64 scattered 16-byte pieces -- the scattered
        // form cost 0.08 ms of conv1_2's 0.90 even with every store hitting the cache (W4_ABL=256).  slot(quad, y, x) = quad * 256 + 16 y +
        // ((x + 2 (y >> 2)) & 15): the rotation keeps the eight lanes of a ds_write_b128 group (four tile columns x two tile rows) on four
        // bank quads (2-way, hidden behind the instruction's own 13 cycles); a row read is 256 contiguous bytes, rotated.
        __syncthreads();   // the partner wave has read this wave's partial outputs
        float4* T = reinterpret_cast<float4*>(smem) + wave * 1024;
#pragma unroll
        for (int aa = 0; aa < 4; ++aa)
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) T[lg * 256 + (4 * ty + aa) * 16 + ((4 * tx + bb + 2 * ty) & 15)] = Y[aa][bb];
        const int by0 = y0 - 4 * ty, bx0 = x0 - 4 * tx, xx = lj;   // (lane = 16 quad + pixel: quad = lg, pixel = lj)
        const long pb = (((long)(ob * (N >> 2) + (nc0 >> 2)) * g.H + by0) * g.W + bx0 + xx) * 4;
        const bool okx = blk_ok && bx0 + xx < g.W;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float4 v = T[lg * 256 + i * 16 + ((xx + 2 * (i >> 2)) & 15)];
            const long po = pb + (long)i * g.W * 4;
            if (okx && by0 + i < g.H && !((W4_ABL & 64) && v.x != 12345.f)) *reinterpret_cast<float4*>(a.out + ((W4_ABL & 256) ? (po & 0x3fffcL) : po)) = v;   // (256: timing only, every store inside a 1 MB window)
        }
    } else {
        // Linear tile blocks: sixteen consecutive tiles in raster order = runs of up to `tw` adjacent tiles of one tile row.  The same
        // hand-over: slot(quad, aa, tile, bb) = ((quad * 4 + aa) * 16 + tile) * 4 + ((bb + (tile >> 1)) & 3) (the rotation spreads the eight
        // lanes of a ds_write_b128 group over all banks); store instruction (quad, aa): lane = 4 tile + bb, i.e. the pixel row aa of all
        // sixteen tiles -- one to three contiguous runs (896 bytes per tile row at 56 pixels) instead of 64 pieces.
        __syncthreads();
        float4* T = reinterpret_cast<float4*>(smem) + wave * 1024;
#pragma unroll
        for (int aa = 0; aa < 4; ++aa)
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) T[((lg * 4 + aa) * 16 + lj) * 4 + ((bb + (lj >> 1)) & 3)] = Y[aa][bb];
        const int rt = lane >> 2, rb = lane & 3;          // the tile and pixel column this lane STORES
        const int Tr = gbo * 16 + rt;
        const bool rok = Tr < g.ntiles_all;
        const int Trc = rok ? Tr : 0;
        const int rob = (int)wino_div((unsigned)Trc, g.m_tiles_img);
        const int rr2 = Trc - rob * g.tiles_img, rty = (int)wino_div((unsigned)rr2, g.m_tw);
        const int ry0 = 4 * rty, rx = 4 * (rr2 - rty * g.tw) + rb;
        const long pr = (((long)(rob * (N >> 2) + (nt * 8 + hf * 4)) * g.H + ry0) * g.W + rx) * 4;   // quad 0, row 0
        const bool rokx = rok && rx < g.W;
        const long planef = (long)g.H * g.W * 4;
        const int rslot = rt * 4 + ((rb + (rt >> 1)) & 3);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int aa = 0; aa < 4; ++aa) {
                const float4 v = T[(q * 4 + aa) * 64 + rslot];
                const long po = pr + q * planef + (long)aa * g.W * 4;
                if (rokx && ry0 + aa < g.H && !((W4_ABL & 64) && v.x != 12345.f)) *reinterpret_cast<float4*>(a.out + ((W4_ABL & 256) ? (po & 0x3fffcL) : po)) = v;
            }
    }
    if (KIND == W4_FWD && a.mask) *reinterpret_cast<uint2*>(a.mask + ((size_t)(tm * a.tiles_n + nt) * 256 + tid) * 2) = make_uint2(ob0, ob1);
    if (POOL) {   // a 4 x 4 tile is 2 x 2 pooling windows: register math, no LDS, no separate pooling pass
        const int HP = g.H >> 1, WP = g.W >> 1;
#pragma unroll
        for (int pa = 0; pa < 2; ++pa)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) {
                const float4 &v00 = Y[2 * pa][2 * pc], &v01 = Y[2 * pa][2 * pc + 1], &v10 = Y[2 * pa + 1][2 * pc], &v11 = Y[2 * pa + 1][2 * pc + 1];
                if (!(blk_ok && y0 + 2 * pa + 1 < g.H && x0 + 2 * pc + 1 < g.W)) continue;
                float4 m;
                m.x = fmaxf(fmaxf(v00.x, v01.x), fmaxf(v10.x, v11.x));
                m.y = fmaxf(fmaxf(v00.y, v01.y), fmaxf(v10.y, v11.y));
                m.z = fmaxf(fmaxf(v00.z, v01.z), fmaxf(v10.z, v11.z));
                m.w = fmaxf(fmaxf(v00.w, v01.w), fmaxf(v10.w, v11.w));
                const size_t pp = (((size_t)(ob * (N >> 2) + (nc0 >> 2)) * HP + (y0 >> 1) + pa) * WP + (x0 >> 1) + pc) * 4;
                *reinterpret_cast<float4*>(a.pool + pp) = m;
                if (a.pbits) {   // where MaxPoolGrad will send the gradient (vc_maxpool2x2_bwd_bits_f32): first maximum in row-major order, valid if > 0
                    auto code = [](float a00, float a01, float a10, float a11, float mx) -> unsigned {
                        return (a00 == mx ? 0u : a01 == mx ? 1u : a10 == mx ? 2u : 3u) | (mx > 0.f ? 4u : 0u);
                    };
                    const unsigned c16 = code(v00.x, v01.x, v10.x, v11.x, m.x) | code(v00.y, v01.y, v10.y, v11.y, m.y) << 4 |
                                         code(v00.z, v01.z, v10.z, v11.z, m.z) << 8 | code(v00.w, v01.w, v10.w, v11.w, m.w) << 12;
                    reinterpret_cast<unsigned short*>(a.pbits)[pp >> 2] = (unsigned short)c16;   // one half-word per (channel quad, pooled pixel): [B][N/4][H/2][W/2]
                }
            }
    }
}

// w [3][3][Ci][Co] (HWIO) -> V = G g G^T (6 x 6), G = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1],
// packed [N/32][C/4][group 2][pq 9][g 4][n 16][pp 4]: channel c = 4 phase + g, column = 32 nt + 16 group + n, position p = 6 v + u = 4 pq + pp
//   transpose 0 (forward):        C = Ci, N = Co, g[ky][kx] = w[ky][kx][c][n]
//   transpose 1 (data gradient):  C = Co, N = Ci, g[ky][kx] = w[2 - ky][2 - kx][n][c]
__global__ __launch_bounds__(256) void wino4_pack_kernel(const float* __restrict__ w, int Ci, int Co, int transpose, float* __restrict__ out) {
    const int C = transpose ? Co : Ci, N = transpose ? Ci : Co;
    const long total = (long)C * N;
    const int nph = C / 4;
    const float G[6][3] = {{0.25f, 0.f, 0.f}, {-1.f / 6, -1.f / 6, -1.f / 6}, {-1.f / 6, 1.f / 6, -1.f / 6},
                           {1.f / 24, 1.f / 12, 1.f / 6}, {1.f / 24, -1.f / 12, 1.f / 6}, {0.f, 0.f, 1.f}};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = transpose ? (int)(i % C) : (int)(i / N), n = transpose ? (int)(i / C) : (int)(i % N);
        float gk[3][3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
                gk[ky][kx] = transpose ? w[((long)((2 - ky) * 3 + (2 - kx)) * Ci + n) * Co + c] : w[((long)(ky * 3 + kx) * Ci + c) * Co + n];
        float t[6][3];   // G g
#pragma unroll
        for (int u = 0; u < 6; ++u)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) t[u][kx] = G[u][0] * gk[0][kx] + G[u][1] * gk[1][kx] + G[u][2] * gk[2][kx];
        const int nt = n >> 5, grp = (n >> 4) & 1, nn = n & 15, ph = c >> 2, gg = c & 3;
        float* o = out + (((long)nt * nph + ph) * 2 + grp) * W4_WGRP + (gg * 16 + nn) * 4;
#pragma unroll
        for (int u = 0; u < 6; ++u)
#pragma unroll
            for (int v = 0; v < 6; ++v) {
                const int p = 6 * v + u;
                o[(p >> 2) * 256 + (p & 3)] = t[u][0] * G[v][0] + t[u][1] * G[v][1] + t[u][2] * G[v][2];
            }
    }
}


// ---- MODE 2's input: V[block][phase][pq 9][g 4][tile 16][pp 4] = B^T d B of tile 16 block + tile, channel 4 phase + g, position 4 pq + pp
// (p = 6 v + u, v horizontal) -- per (block, phase) exactly the image a wave's five fragment loads walk.  One wave = one (block, phase):
// a lane = (g, tile) gathers its channel's 6 x 6 patch (36 4-byte loads; the four g lanes of a tile cover a pixel's 16 bytes, the six
// columns a row's lines), runs the SAME twelve-operation 1-D transforms in the same order as the fused kernel (rows, then columns: the
// results are bit-identical to what MODE 0 / 1 feed their MFMAs) and stores nine float4: 1 KB contiguous per wave and store.  HBM-bound:
// reads the activation once (+ halos from the L2), writes 2.25 x its bytes.
struct Wino4XArgs {
    const float* x;
    float* V;
    int B, H, W, C, nph, nblocks, tw, tiles_img, ntiles_all;
    unsigned m_tiles_img, m_tw;
};
__global__ __launch_bounds__(256) void wino4_xform_kernel(Wino4XArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lj = lane & 15, lg = lane >> 4;
    const int blk = blockIdx.x * 4 + wave, ph = blockIdx.y;
    if (blk >= a.nblocks) return;
    const int T = blk * 16 + lj;
    const bool live = T < a.ntiles_all;
    const unsigned Tc = live ? (unsigned)T : 0u;
    const unsigned b = wino_div(Tc, a.m_tiles_img), r2 = Tc - b * (unsigned)a.tiles_img;
    const unsigned tyy = wino_div(r2, a.m_tw), txx = r2 - tyy * (unsigned)a.tw;
    const int y0 = (int)(tyy * 4u) - 1, x0 = (int)(txx * 4u) - 1;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)((long)a.B * a.H * a.W * a.C * 4), 0x00020000);
    const unsigned plane = (b * (unsigned)(a.C >> 2) + (unsigned)ph) * (unsigned)a.H;
    float Hh[6][6];   // [row r][horizontal index v]
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        float d[6], t[3];
        const int y = y0 + r;
        const bool oky = live && (unsigned)y < (unsigned)a.H;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int x = x0 + j;
            const unsigned off = (oky && (unsigned)x < (unsigned)a.W) ? (((plane + (unsigned)y) * (unsigned)a.W + (unsigned)x) * 4u + (unsigned)lg) * 4u : WOOB;
            d[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, (int)off, 0, 0));
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) w4_hstep<0>(k, d, &Hh[r][0], t);
#pragma unroll
        for (int k = 0; k < 6; ++k) w4_hstep<1>(k, d, &Hh[r][3], t);
    }
    float o[36];   // o[6 v + u]
#pragma unroll
    for (int v = 0; v < 6; ++v) {
        float t[5];
#pragma unroll
        for (int k = 0; k < 12; ++k) w4_bstep_v(k, Hh[0][v], Hh[1][v], Hh[2][v], Hh[3][v], Hh[4][v], Hh[5][v], &o[6 * v], t);
    }
    float4* dst = reinterpret_cast<float4*>(a.V) + ((size_t)blk * a.nph + ph) * 9 * 64 + lane;
#pragma unroll
    for (int pq = 0; pq < 9; ++pq) dst[pq * 64] = make_float4(o[4 * pq], o[4 * pq + 1], o[4 * pq + 2], o[4 * pq + 3]);
}

static bool plan_wino4(int B, int H, int W, int C, int N, Wino4Geom& g) {
    g.B = B; g.H = H; g.W = W; g.C = C; g.N = N;
    if (B <= 0 || H < 4 || W < 4 || C <= 0 || N <= 0 || C % 8 || N % 32) return false;   // (any H, W: stores and pooling windows past the image are masked)
    if ((long)B * H * W * (long)(C > N ? C : N) * 4 > 0x7fffffffL || 36L * C * N * 4 > 0x7fffffffL) return false;
    g.bx_n = cdiv(W, 16); g.by_n = cdiv(H, 16);
    g.blocks_img = g.bx_n * g.by_n;
    if ((long)B * g.blocks_img > 0x3fffffffL) return false;
    g.nblocks = B * g.blocks_img;
    if ((long)g.nblocks * g.blocks_img >= 0x100000000L) return false;
    g.m_blocks_img = wino_magic(g.blocks_img); g.m_bx_n = wino_magic(g.bx_n);
    // linear tiles where they need fewer blocks than the square ones (VGG16: the 56- and 28-wide layers; a 14-wide image is one block either way)
    const int th = cdiv(H, 4);
    g.tw = cdiv(W, 4); g.tiles_img = th * g.tw; g.ntiles_all = B * g.tiles_img;
    g.m_tiles_img = wino_magic(g.tiles_img); g.m_tw = wino_magic(g.tw);
    g.lt = ((long)B * g.tiles_img < 0x7ffffff0L && cdiv(g.ntiles_all, 16) < g.nblocks) ? 1 : 0;
    if (g.lt) g.nblocks = cdiv(g.ntiles_all, 16);
    return true;
}

static int wino4_attr() {
    static int once = [] {
        hipError_t e = hipSuccess;
        auto set = [&](const void* f) { if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, WINO4_LDS_BYTES); };
        set(reinterpret_cast<const void*>(conv_wino4_kernel<W4_FWD, false, false>));
        set(reinterpret_cast<const void*>(conv_wino4_kernel<W4_FWD, true, false>));
        set(reinterpret_cast<const void*>(conv_wino4_kernel<W4_DGRAD, false, false>));
        auto setl = [&](const void* f) { if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, WINO4L_LDS_BYTES); };
        setl(reinterpret_cast<const void*>(conv_wino4_kernel<W4_FWD, false, true>));
        setl(reinterpret_cast<const void*>(conv_wino4_kernel<W4_FWD, true, true>));
        setl(reinterpret_cast<const void*>(conv_wino4_kernel<W4_DGRAD, false, true>));
        set(reinterpret_cast<const void*>(conv_wino4_kernel<W4_FWD, false, W4_VIN>));
        set(reinterpret_cast<const void*>(conv_wino4_kernel<W4_FWD, true, W4_VIN>));
        set(reinterpret_cast<const void*>(conv_wino4_kernel<W4_DGRAD, false, W4_VIN>));
        return e == hipSuccess ? 0 : fail((int)e, "%s: hipFuncSetAttribute failed", "conv wino4 kernel");
    }();
    return once;
}

// one launch over nb images; C = gathered channels, N = produced channels
static bool plan_wino4v(int B, int H, int W, int C, int N, Wino4Geom& g);
static int wino4_launch(hipStream_t st, int kind, int nb, int H, int W, int C, int N, const float* x, const float* wp, float* out, const float* aux,
                        float* pool, unsigned* mask, int relu, unsigned* pbits = nullptr, float* vws = nullptr) {
    Wino4Args a;
    if (!(vws ? plan_wino4v(nb, H, W, C, N, a.g) : plan_wino4(nb, H, W, C, N, a.g))) return fail(VC_EINVAL, "%s: unsupported shape (vc_conv3x3_wino4_supported / vc_conv3x3_wino4v_supported)", "conv wino4");
    int rc = wino4_attr();
    if (rc) return rc;
    if (vws) {   // MODE 2: transform the input once (x -> V), then run the main loop on V
        Wino4XArgs xa{x, vws, nb, H, W, C, C / 4, a.g.nblocks, a.g.tw, a.g.tiles_img, a.g.ntiles_all, a.g.m_tiles_img, a.g.m_tw};
        hipLaunchKernelGGL(wino4_xform_kernel, dim3(cdiv(a.g.nblocks, 4), C / 4), dim3(256), 0, st, xa);
        rc = launch_status("conv wino4v transform");
        if (rc) return rc;
        x = vws;
    }
    a.x = x; a.wp = wp; a.out = out; a.aux = aux; a.pool = pool; a.mask = mask; a.pbits = pbits; a.relu = relu;
    a.tiles_n = N / 32;
    a.nphases = C / 4;
    a.ntiles = cdiv(a.g.nblocks, W4_NBLK) * a.tiles_n;
    // Workgroup order (the kernel's id -> (block pair, channel tile)): the 64 workgroups an XCD holds at a time share its 4 MB L2; with all
    // N / 32 channel tiles of a block pair side by side (the order up to round 4) the 512-channel layers keep 16 x 2.4 MB of packed weights
    // in flight per XCD and re-fetch them for every four block pairs: 12-13 x the algorithmic bytes on conv4_x.  Chunks of FOUR channel
    // tiles (16 block pairs x 4 tiles per XCD) balance patch and weight re-fetches: 10.1 -> 6.6 GB per forward pass over the twelve
    // layers, data gradient 9.9 -> 7.0 (profiles/r04_traffic_layers.md; 8: 7.7, 2: 8.3); the time does not change (the Infinity Cache
    // serves the re-fetches either way).  VC_WINO4_TG=<n>: another chunk width, 0 = all tiles of a block pair together.
    static const int tg_env = getenv("VC_WINO4_TG") ? atoi(getenv("VC_WINO4_TG")) : 4;
    a.tg = (tg_env > 0 && a.tiles_n % tg_env == 0) ? tg_env : a.tiles_n;
    a.per_chunk = cdiv(a.g.nblocks, W4_NBLK) * a.tg;
    if (vws) {
        if (kind == W4_DGRAD) hipLaunchKernelGGL((conv_wino4_kernel<W4_DGRAD, false, W4_VIN>), dim3(a.ntiles), dim3(256), WINO4_LDS_BYTES, st, a);
        else if (pool) hipLaunchKernelGGL((conv_wino4_kernel<W4_FWD, true, W4_VIN>), dim3(a.ntiles), dim3(256), WINO4_LDS_BYTES, st, a);
        else hipLaunchKernelGGL((conv_wino4_kernel<W4_FWD, false, W4_VIN>), dim3(a.ntiles), dim3(256), WINO4_LDS_BYTES, st, a);
    } else if (a.g.lt) {
        if (kind == W4_DGRAD) hipLaunchKernelGGL((conv_wino4_kernel<W4_DGRAD, false, true>), dim3(a.ntiles), dim3(256), WINO4L_LDS_BYTES, st, a);
        else if (pool) hipLaunchKernelGGL((conv_wino4_kernel<W4_FWD, true, true>), dim3(a.ntiles), dim3(256), WINO4L_LDS_BYTES, st, a);
        else hipLaunchKernelGGL((conv_wino4_kernel<W4_FWD, false, true>), dim3(a.ntiles), dim3(256), WINO4L_LDS_BYTES, st, a);
    } else if (kind == W4_DGRAD) hipLaunchKernelGGL((conv_wino4_kernel<W4_DGRAD, false, false>), dim3(a.ntiles), dim3(256), WINO4_LDS_BYTES, st, a);
    else if (pool) hipLaunchKernelGGL((conv_wino4_kernel<W4_FWD, true, false>), dim3(a.ntiles), dim3(256), WINO4_LDS_BYTES, st, a);
    else hipLaunchKernelGGL((conv_wino4_kernel<W4_FWD, false, false>), dim3(a.ntiles), dim3(256), WINO4_LDS_BYTES, st, a);
    return launch_status("conv wino4");
}

// MODE 2's geometry: linear tile blocks, and only where the fused kernel's launch on the same shape puts the same tile in the same lane
// (its linear blocks, or one 4 x 4-tile square per image): the ReLU mask bits of the two forms are then interchangeable.
static bool plan_wino4v(int B, int H, int W, int C, int N, Wino4Geom& g) {
    if (!plan_wino4(B, H, W, C, N, g)) return false;
    if (!(g.lt || (g.tw == 4 && cdiv(H, 4) == 4))) return false;
    g.lt = 1;
    g.nblocks = cdiv(g.ntiles_all, 16);
    return (long)g.nblocks * C * 2304 <= 0x7fffffffL;
}

}  // namespace vc

// ---- C ABI (the entries mirror conv_wino.hip's, argument for argument) ------------------------------
extern "C" int vc_conv3x3_wino4_supported(int B, int H, int W, int Cin, int Cout, int dgrad) {
    vc::Wino4Geom g;
    const int nb = vc::wino_images_per_launch(B, H, W, Cin, Cout);
    return nb > 0 && (dgrad ? vc::plan_wino4(nb, H, W, Cout, Cin, g) : vc::plan_wino4(nb, H, W, Cin, Cout, g)) ? 1 : 0;
}

// 1 when this kernel is the faster Winograd form for the layer: its 16 x 16-pixel blocks must cover the image at least 0.75 x as well as
// the F(2x2,3x3) kernel's blocks of sixteen 2 x 2 tiles do.  Decided on the training step itself (cfg4 at 64 images, three streams, ms per
// step).  Round 3 (each wave transformed the whole 6 x 6 patch; profiles/r03_wino4_layers.txt): the rule was 0.85 -- the 56-wide conv3_x (77 %
// coverage against 100 %) lost 0.14 ms on F(4x4,3x3).  Round 4 (the two waves of a block split the positions, half the transform each):
// conv3_x gains 2-12 % kernel by kernel and 0.16 ms in the step (28.74 -> 28.58, profiles/r04_wino4_layers.txt), so every VGG16 layer
// behind conv1_1 now runs this kernel; VC_WINO4_MIN_COVERAGE overrides the ratio for A/B runs.
extern "C" int vc_conv3x3_wino4_preferred(int B, int H, int W, int Cin, int Cout) {
    if (!vc_conv3x3_wino4_supported(B, H, W, Cin, Cout, 0) || !vc_conv3x3_wino4_supported(B, H, W, Cin, Cout, 1) || (H & 1) || (W & 1)) return 0;
    // share of the tile slots that hold pixels of the image: linear tiles leave only the padding INSIDE the 4 x 4 tiles of the right / bottom edge
    const double e4 = (double)H * W / ((double)vc::cdiv(H, 4) * vc::cdiv(W, 4) * 16.0);
    static const double ratio = getenv("VC_WINO4_MIN_COVERAGE") ? atof(getenv("VC_WINO4_MIN_COVERAGE")) : 0.75;   // (A/B runs of the rule)
    return e4 >= ratio * vc::wino2_coverage(H, W) ? 1 : 0;
}

extern "C" int vc_conv3x3_wino4_pack_f32(void* stream, int Cin, int Cout, const float* w, int transpose, float* wp) {
    using namespace vc;
    const int C = transpose ? Cout : Cin, N = transpose ? Cin : Cout;
    VC_CHECK_ARG(C > 0 && N > 0 && C % 8 == 0 && N % 32 == 0, "gathered channels % 8 == 0 and output channels % 32 == 0 required");
    VC_CHECK_ARG(w && wp && waligned16(wp), "null or misaligned pointer");
    const long total = (long)Cin * Cout;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(wino4_pack_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, Cin, Cout, transpose, wp);
    return launch_status(__func__);
}

extern "C" int vc_conv3x3_wino4_fwd_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* x, const float* wp,
                                        const float* bias, float* y, float* ypool, int relu) {
    using namespace vc;
    VC_CHECK_ARG(x && wp && y, "null pointer");
    VC_CHECK_ARG(waligned16(x) && waligned16(wp) && waligned16(y) && waligned16(bias) && waligned16(ypool), "pointers must be 16-byte aligned");
    VC_CHECK_ARG(!ypool || !((H | W) & 1), "the fused max-pool needs even H and W");
    const int per = wino_images_per_launch(B, H, W, Cin, Cout);
    VC_CHECK_ARG(per > 0 && vc_conv3x3_wino4_supported(B, H, W, Cin, Cout, 0), "unsupported shape (vc_conv3x3_wino4_supported)");
    for (int b0 = 0; b0 < B; b0 += per) {   // image ranges of < 2 GiB
        const int nb = B - b0 < per ? B - b0 : per;
        const int rc = wino4_launch((hipStream_t)stream, W4_FWD, nb, H, W, Cin, Cout, x + (size_t)b0 * H * W * Cin, wp, y + (size_t)b0 * H * W * Cout, bias,
                                    ypool ? ypool + (size_t)b0 * (H / 2) * (W / 2) * Cout : nullptr, nullptr, relu);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int vc_conv3x3_wino4_fwd_pool_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* x, const float* wp,
                                             const float* bias, float* y, float* ypool, uint32_t* pool_bits) {
    using namespace vc;
    VC_CHECK_ARG(x && wp && y && ypool && pool_bits, "null pointer");
    VC_CHECK_ARG(waligned16(x) && waligned16(wp) && waligned16(y) && waligned16(bias) && waligned16(ypool), "pointers must be 16-byte aligned");
    VC_CHECK_ARG(!((H | W) & 1), "the fused max-pool needs even H and W");
    const int per = wino_images_per_launch(B, H, W, Cin, Cout);
    VC_CHECK_ARG(per > 0 && vc_conv3x3_wino4_supported(B, H, W, Cin, Cout, 0), "unsupported shape (vc_conv3x3_wino4_supported)");
    for (int b0 = 0; b0 < B; b0 += per) {
        const int nb = B - b0 < per ? B - b0 : per;
        const size_t po = (size_t)b0 * (H / 2) * (W / 2) * Cout;
        const int rc = wino4_launch((hipStream_t)stream, W4_FWD, nb, H, W, Cin, Cout, x + (size_t)b0 * H * W * Cin, wp, y + (size_t)b0 * H * W * Cout, bias,
                                    ypool + po, nullptr, 1, pool_bits + po / 8);
        if (rc) return rc;
    }
    return 0;
}

extern "C" size_t vc_conv3x3_wino4_mask_words(int B, int H, int W, int C) {
    vc::Wino4Geom g;
    if (!vc::plan_wino4(B, H, W, 8, C, g)) return 0;
    return (size_t)vc::cdiv(B * g.blocks_img, vc::W4_NBLK) * (C / 32) * 512;   // (the square-block count: never less than the linear one)
}

extern "C" int vc_conv3x3_wino4_fwd_mask_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* x, const float* wp,
                                             const float* bias, float* y, int relu, uint32_t* mask_out) {
    using namespace vc;
    VC_CHECK_ARG(x && wp && y && mask_out, "null pointer");
    VC_CHECK_ARG(waligned16(x) && waligned16(wp) && waligned16(y) && waligned16(bias) && waligned16(mask_out), "pointers must be 16-byte aligned");
    VC_CHECK_ARG(wino_images_per_launch(B, H, W, Cin, Cout) >= B && vc_conv3x3_wino4_supported(B, H, W, Cin, Cout, 0),
                 "unsupported shape, or more images than one launch takes (vc_conv3x3_wino_single_launch_supported): the mask bits are per tile of ONE launch");
    return wino4_launch((hipStream_t)stream, W4_FWD, B, H, W, Cin, Cout, x, wp, y, bias, nullptr, mask_out, relu);
}

extern "C" int vc_conv3x3_wino4_dgrad_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* dy, const float* wpt,
                                          const float* relu_src, float* dx) {
    using namespace vc;
    VC_CHECK_ARG(dy && wpt && dx, "null pointer");
    VC_CHECK_ARG(waligned16(dy) && waligned16(wpt) && waligned16(dx) && waligned16(relu_src), "pointers must be 16-byte aligned");
    const int per = wino_images_per_launch(B, H, W, Cin, Cout);
    VC_CHECK_ARG(per > 0 && vc_conv3x3_wino4_supported(B, H, W, Cin, Cout, 1), "unsupported shape (vc_conv3x3_wino4_supported)");
    for (int b0 = 0; b0 < B; b0 += per) {
        const int nb = B - b0 < per ? B - b0 : per;
        const int rc = wino4_launch((hipStream_t)stream, W4_DGRAD, nb, H, W, Cout, Cin, dy + (size_t)b0 * H * W * Cout, wpt, dx + (size_t)b0 * H * W * Cin,
                                    relu_src ? relu_src + (size_t)b0 * H * W * Cin : nullptr, nullptr, nullptr, 0);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int vc_conv3x3_wino4_dgrad_bits_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* dy, const float* wpt,
                                               const uint32_t* mask_bits, float* dx) {
    using namespace vc;
    VC_CHECK_ARG(dy && wpt && dx && mask_bits, "null pointer");
    VC_CHECK_ARG(waligned16(dy) && waligned16(wpt) && waligned16(dx) && waligned16(mask_bits), "pointers must be 16-byte aligned");
    VC_CHECK_ARG(wino_images_per_launch(B, H, W, Cin, Cout) >= B && vc_conv3x3_wino4_supported(B, H, W, Cin, Cout, 1),
                 "unsupported shape, or more images than one launch takes (vc_conv3x3_wino_single_launch_supported)");
    return wino4_launch((hipStream_t)stream, W4_DGRAD, B, H, W, Cout, Cin, dy, wpt, dx, nullptr, nullptr, const_cast<uint32_t*>(mask_bits), 0);
}

// ---- the same operations with the input transformed ONCE into a workspace (MODE 2 above): vc_conv3x3_wino4v_* = vc_conv3x3_wino4_* + (vws,
// vws_bytes); same packed weights, same outputs, same mask bits / routing codes (the shapes it takes are the ones where the lanes coincide).
extern "C" int vc_conv3x3_wino4v_supported(int B, int H, int W, int Cin, int Cout, int dgrad) {
    vc::Wino4Geom g;
    if (vc::wino_images_per_launch(B, H, W, Cin, Cout) < B) return 0;   // one launch only
    return (dgrad ? vc::plan_wino4v(B, H, W, Cout, Cin, g) : vc::plan_wino4v(B, H, W, Cin, Cout, g)) ? 1 : 0;
}
// 1 where the pre-transformed form is the faster one for a launch over B images.  Measured per layer (tools/experiments/wino4v_try.py,
// profiles/r06_wino4v_layers.txt; fused -> transform + main kernel): it wins where V (2.25 x the gathered activation, written and read
// once) is small beside the MFMA work it unloads -- the 28- and 14-wide layers (conv4_x, conv5_x: x1.05-1.19) -- and loses on the 56-wide
// ones (conv3_x: 0.73-0.89, V = 231-462 MB per half batch).  VC_WINO4V_MAX_PIXELS overrides the image-size bound (0 = never).
extern "C" int vc_conv3x3_wino4v_preferred(int B, int H, int W, int Cin, int Cout, int dgrad) {
    static const long maxpix = getenv("VC_WINO4V_MAX_PIXELS") ? atol(getenv("VC_WINO4V_MAX_PIXELS")) : 28 * 28;
    return (long)H * W <= maxpix && (dgrad ? Cout : Cin) >= 256 && vc_conv3x3_wino4v_supported(B, H, W, Cin, Cout, dgrad) ? 1 : 0;
}
extern "C" size_t vc_conv3x3_wino4v_workspace_bytes(int B, int H, int W, int C) {
    vc::Wino4Geom g;
    if (!vc::plan_wino4v(B, H, W, C, 32, g)) return 0;
    return (size_t)g.nblocks * C * 2304;
}
#define VC_W4V_ARGS(C_)                                                                                                       \
    VC_CHECK_ARG(vws && waligned16(vws) && vws_bytes >= vc_conv3x3_wino4v_workspace_bytes(B, H, W, C_) && vws_bytes > 0,      \
                 "workspace missing or too small (vc_conv3x3_wino4v_workspace_bytes), or unsupported shape")
extern "C" int vc_conv3x3_wino4v_fwd_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* x, const float* wp, const float* bias,
                                         float* y, float* ypool, int relu, float* vws, size_t vws_bytes) {
    using namespace vc;
    VC_CHECK_ARG(x && wp && y, "null pointer");
    VC_CHECK_ARG(waligned16(x) && waligned16(wp) && waligned16(y) && waligned16(bias) && waligned16(ypool), "pointers must be 16-byte aligned");
    VC_CHECK_ARG(!ypool || !((H | W) & 1), "the fused max-pool needs even H and W");
    VC_CHECK_ARG(vc_conv3x3_wino4v_supported(B, H, W, Cin, Cout, 0), "unsupported shape (vc_conv3x3_wino4v_supported)");
    VC_W4V_ARGS(Cin);
    return wino4_launch((hipStream_t)stream, W4_FWD, B, H, W, Cin, Cout, x, wp, y, bias, ypool, nullptr, relu, nullptr, vws);
}
extern "C" int vc_conv3x3_wino4v_fwd_pool_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* x, const float* wp, const float* bias,
                                              float* y, float* ypool, uint32_t* pool_bits, float* vws, size_t vws_bytes) {
    using namespace vc;
    VC_CHECK_ARG(x && wp && y && ypool && pool_bits, "null pointer");
    VC_CHECK_ARG(waligned16(x) && waligned16(wp) && waligned16(y) && waligned16(bias) && waligned16(ypool), "pointers must be 16-byte aligned");
    VC_CHECK_ARG(!((H | W) & 1), "the fused max-pool needs even H and W");
    VC_CHECK_ARG(vc_conv3x3_wino4v_supported(B, H, W, Cin, Cout, 0), "unsupported shape (vc_conv3x3_wino4v_supported)");
    VC_W4V_ARGS(Cin);
    return wino4_launch((hipStream_t)stream, W4_FWD, B, H, W, Cin, Cout, x, wp, y, bias, ypool, nullptr, 1, pool_bits, vws);
}
extern "C" int vc_conv3x3_wino4v_fwd_mask_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* x, const float* wp, const float* bias,
                                              float* y, int relu, uint32_t* mask_out, float* vws, size_t vws_bytes) {
    using namespace vc;
    VC_CHECK_ARG(x && wp && y && mask_out, "null pointer");
    VC_CHECK_ARG(waligned16(x) && waligned16(wp) && waligned16(y) && waligned16(bias) && waligned16(mask_out), "pointers must be 16-byte aligned");
    VC_CHECK_ARG(vc_conv3x3_wino4v_supported(B, H, W, Cin, Cout, 0), "unsupported shape (vc_conv3x3_wino4v_supported)");
    VC_W4V_ARGS(Cin);
    return wino4_launch((hipStream_t)stream, W4_FWD, B, H, W, Cin, Cout, x, wp, y, bias, nullptr, mask_out, relu, nullptr, vws);
}
extern "C" int vc_conv3x3_wino4v_dgrad_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* dy, const float* wpt, const float* relu_src,
                                           float* dx, float* vws, size_t vws_bytes) {
    using namespace vc;
    VC_CHECK_ARG(dy && wpt && dx, "null pointer");
    VC_CHECK_ARG(waligned16(dy) && waligned16(wpt) && waligned16(dx) && waligned16(relu_src), "pointers must be 16-byte aligned");
    VC_CHECK_ARG(vc_conv3x3_wino4v_supported(B, H, W, Cin, Cout, 1), "unsupported shape (vc_conv3x3_wino4v_supported)");
    VC_W4V_ARGS(Cout);
    return wino4_launch((hipStream_t)stream, W4_DGRAD, B, H, W, Cout, Cin, dy, wpt, dx, relu_src, nullptr, nullptr, 0, nullptr, vws);
}
extern "C" int vc_conv3x3_wino4v_dgrad_bits_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* dy, const float* wpt,
                                                const uint32_t* mask_bits, float* dx, float* vws, size_t vws_bytes) {
    using namespace vc;
    VC_CHECK_ARG(dy && wpt && dx && mask_bits, "null pointer");
    VC_CHECK_ARG(waligned16(dy) && waligned16(wpt) && waligned16(dx) && waligned16(mask_bits), "pointers must be 16-byte aligned");
    VC_CHECK_ARG(vc_conv3x3_wino4v_supported(B, H, W, Cin, Cout, 1), "unsupported shape (vc_conv3x3_wino4v_supported)");
    VC_W4V_ARGS(Cout);
    return wino4_launch((hipStream_t)stream, W4_DGRAD, B, H, W, Cout, Cin, dy, wpt, dx, nullptr, nullptr, const_cast<uint32_t*>(mask_bits), 0, nullptr, vws);
}
