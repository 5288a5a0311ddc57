// 3x3 / stride 1 / SAME convolution, DIRECT form, on the bf16 matrix pipe ("bf16x3"): forward and data gradient of
// utils/image_embeddings.py:36-212 with f32 activations in HBM (the C4 layout of the Winograd kernels), f32 accumulators and
// split-bf16 operands -- a = a_hi + a_lo, three v_mfma_f32_32x32x16_bf16 products (hi.hi + hi.lo + lo.hi) per k-step, exactly
// the arithmetic of gemm_bf16x3_core.h.  NOT the reference's arithmetic (tf.float32 conv2d): the opt-in mode of
// vc_gemm_set_precision(1) / Trainer(precision="bf16x3"), reported on its own bench lines.
//
// Why direct and not Winograd here: at 16x the f32 MFMA rate the multiplications are nearly free and a kernel is bound by what it
// stages and transforms per MFMA.  The direct form needs NO transform, and -- unlike a GEMM -- stages an activation once for nine
// taps: per 32-channel slab a workgroup splits its halo patch ONCE (three VALU operations per element) into the LDS image
// [patch pixel][32 k: hi | lo] and then runs nine taps against it, each tap only streaming 18 KB of weights that were split when
// they were packed (once per optimiser step).  A tap is a ROW OFFSET into the patch image: the MFMA's B operand (column = pixel)
// of output pixel (r, c) at tap (ty, tx) is patch row (r + ty) PW + (c + tx), so all nine taps read the same bytes.
//
//   D[co, pixel] += W_tap[co, ci] . X[ci, pixel + tap]      A = weights (row = output channel), B = patch (column = pixel)
//
// With the output channels as the M dimension a lane's accumulator quad (reg & 3) is FOUR CONSECUTIVE CHANNELS of one pixel = one
// 16-byte element of a C4 plane, and the 32 lanes of a pixel run are consecutive pixels: the epilogue stores straight from registers.
//
// Geometry.  An MFMA pixel tile ("run") = RPT rows x RW columns of pixels, RPT RW <= 32 (224-wide: 1 x 32; 112: 2 x 16; 56: 4 x 8;
// 28: 1 x 28; 14: 2 x 14 -- the last two leave four lanes idle).  Rows are PADDED GLOBAL rows pr = b (H + 1) + y: one all-zero row
// between consecutive images, so a tile may run across image boundaries (the zero row is the bottom halo of one image and the top
// halo of the next) and the deep layers lose 1 / (H + 1) of their rows instead of a quarter of their tile slots.  A workgroup =
// eight waves = NWP pixel groups x NWC channel groups, a wave = 64 channels x two runs (four 32 x 32 accumulators); workgroup tile =
// 64 NWC channels x 2 NWP runs stacked vertically (N = 64: 1 x 8 waves; else 2 x 4: 128 channels x 256 pixels -- the weights are
// the dominant traffic, 4 / (pixels per tile) bytes per MAC).  Order: channel tile slowest, so the workgroups an XCD holds share
// one tile of weights in its L2.
//
// LDS: patch image (PH PW rows x 144 B, single: rewritten between two barriers once per slab) + two weight images (64 NWC rows x
// 144 B, double-buffered: ONE barrier per tap).  Row pitch 144 B = nine 16-byte slots: the sixteen lanes of a ds_read_b128 group
// fall on sixteen different slots when their rows are consecutive (gemm_bf16x3_core.h).
#include <stdlib.h>
#include "conv_wino.h"
#include "gemm_bf16x3_core.h"

// CB_ABL (timing only, `make cbabl`; never shipped): 1 no weight loads after the first, 2 no patch loads / splits / writes after the
// first, 4 no MFMAs, 8 no output stores, 16 no fragment reads, 32 no barriers
#ifndef CB_ABL
#define CB_ABL 0
#endif
#ifndef CB_KPRE   // 2: both k-steps' fragment reads of a tap ahead of its MFMAs (measured: slower); 1: per k-step
#define CB_KPRE 1
#endif
namespace vc {

constexpr int CB_PITCH = BX_PITCH;   // 144
// patch staging slots (float4) per thread and slab, by (channel groups, waves)
constexpr int cb_maxp(int nwc, int nw) { return nw == 8 ? (nwc == 1 ? 10 : 6) : (nwc == 1 ? 11 : 7); }

struct ConvBxArgs {
    const float* x;      // [B][C/4][H][W][4]
    const char* wp;      // packed weights [N / TN][C / 32][9 taps][TN rows][144 B] (vc_conv3x3_bx_pack_f32)
    float* out;          // [B][N/4][H][W][4]
    const float* aux;    // forward: bias [N] or null; data gradient: ReLU source in the layout of out, or null
    int B, H, W, C, N;
    int relu;
    int RW, RPT, TR;     // run = RPT rows x RW columns; TR = padded rows per workgroup tile
    int PW, NPIX;        // patch: (TR + 2) rows x PW = RW + 2 columns
    int col_tiles, ptiles, ntiles;
    unsigned m_rw, m_pw, m_hp1, m_coltiles, m_ptiles, m_npix;   // wino_magic of RW, PW, H + 1, col_tiles, ptiles, NPIX
};

// KIND 0: forward (bias, ReLU); 1: data gradient (ReLU mask from aux).  NW = 8 waves, one workgroup per CU; or 4 waves (half the pixel
// tile), TWO workgroups per CU: twice the tiles for the small deep layers (conv5_x at 32 images: 120 eight-wave tiles for 256 CUs),
// and two workgroups that are not barrier-locked to each other overlap one's staging with the other's MFMAs
template <int KIND, int NWC, int NW>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 2 : 1) void conv_bx_kernel(ConvBxArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int CB_NT = NW * 64;
    constexpr int NWP = NW / NWC, TN = 64 * NWC, RUNS = 2 * NWP;
    constexpr int WBYTES = TN * CB_PITCH;                  // one tap's weight image
    constexpr int WPIECES = WBYTES / 16, WSLOTS = (WPIECES + CB_NT - 1) / CB_NT;
    constexpr int MAXP = cb_maxp(NWC, NW);                 // patch slots (float4) per thread and slab
    char* Ws = smem;                                       // two weight images
    char* Ps = smem + 2 * WBYTES;                          // the patch image
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wc = wave % NWC, wp = wave / NWC;
    const int li = lane & 31, lh = lane >> 5;
    const int H = a.H, W = a.W, C = a.C, N = a.N;
    const unsigned plane_b = (unsigned)H * (unsigned)W * 16u;   // bytes of one channel-quad plane
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)((long)a.B * H * W * C * 4), 0x00020000);

    // PERSISTENT workgroups: tile ids blockIdx.x, + gridDim.x, ... as ONE flat sequence of (tile, slab, tap) steps -- the patch of a
    // tile's first slab is prefetched under the last taps of the previous tile exactly like the next slab's, and the epilogue's stores
    // drain under the next tile's MFMAs (with one workgroup per CU a tile-per-launch kernel exposed both: conv1_2's tiles spent 24 of
    // their 37 us outside the eighteen taps).  Channel tile slowest: the workgroups in flight share one tile of weights in L2.
    struct Tile { int co_t, pr0, x0; };
    auto decode = [&](int id) {
        const unsigned co_t = wino_div((unsigned)id, a.m_ptiles), pt = (unsigned)id - co_t * a.ptiles;
        const unsigned row_t = wino_div(pt, a.m_coltiles), col_t = pt - row_t * a.col_tiles;
        return Tile{(int)co_t, (int)row_t * a.TR, (int)col_t * a.RW};
    };

    // ---- staging slots of the patch: slot s = tid + 512 u -> (channel quad q = s / NPIX, patch pixel s % NPIX); consecutive lanes =
    // consecutive pixels of a patch row of ONE plane (16-byte pieces, contiguous in the C4 layout)
    const int nslots = a.NPIX * 8;
    unsigned pvoff[MAXP];
    int plds[MAXP], pinfo[MAXP];   // pinfo = patch row << 12 | patch column << 4 | channel quad
#pragma unroll
    for (int u = 0; u < MAXP; ++u) {
        const unsigned s = (unsigned)(tid + u * CB_NT);
        plds[u] = -1; pinfo[u] = 0;
        if ((int)s < nslots) {
            const unsigned q = wino_div(s, a.m_npix), pix = s - q * a.NPIX;
            const unsigned py = wino_div(pix, a.m_pw), px = pix - py * a.PW;
            plds[u] = (int)pix * CB_PITCH + (int)q * 8;
            pinfo[u] = (int)((py << 12) | (px << 4) | q);
        }
    }
    auto set_patch = [&](const Tile& t) {   // global offsets of the thread's slots for tile t (WOOB: outside every image -> zeros)
#pragma unroll
        for (int u = 0; u < MAXP; ++u) {
            pvoff[u] = WOOB;
            const int pr = t.pr0 - 1 + (pinfo[u] >> 12), xx = t.x0 - 1 + ((pinfo[u] >> 4) & 255);
            if (plds[u] >= 0 && pr >= 0 && xx >= 0 && xx < W) {
                const unsigned b = wino_div((unsigned)pr, a.m_hp1), y = (unsigned)pr - b * (unsigned)(H + 1);
                if ((int)b < a.B && (int)y < H)
                    pvoff[u] = (((b * (unsigned)(C >> 2) + (unsigned)(pinfo[u] & 15)) * (unsigned)H + y) * (unsigned)W + (unsigned)xx) * 16u;
            }
        }
    };
    float4 preg[MAXP];
    auto pload = [&](int slab) {
        const unsigned soff = (unsigned)(slab * 8) * plane_b;
#pragma unroll
        for (int u = 0; u < MAXP; ++u)
            if (u * CB_NT < nslots) preg[u] = wbufload(rx, pvoff[u], soff);   // (uniform bound: the slots of whole 512-thread rounds)
    };
    auto pput = [&]() {
#pragma unroll
        for (int u = 0; u < MAXP; ++u) {
            if (u * CB_NT >= nslots || plds[u] < 0) continue;
            u32x2 hi, lo;
            split_quad(preg[u].x, preg[u].y, preg[u].z, preg[u].w, hi, lo);
            *reinterpret_cast<u32x2*>(Ps + plds[u]) = hi;
            *reinterpret_cast<u32x2*>(Ps + plds[u] + 64) = lo;
        }
    };
    // ---- weights of one (channel tile, slab, tap): a linear copy of WBYTES
    const int nslab = C / 32, nsteps = nslab * 9;
    // THREE register sets, set = tap % 3: the weights of step s are loaded at the top of step s - 3 and written to the LDS at the end
    // of step s - 1 -- with one set the load issued at the top of a step was waited for at its end, one MFMA phase (~0.7 us) later:
    // shorter than an L2 round trip under load (ablation: 16 % of the kernel).
    u32x4 wreg[3][WSLOTS];
    auto wload = [&](int set, int co_t, int step) {   // step = slab * 9 + tap
        const char* src = a.wp + ((size_t)co_t * nsteps + step) * WBYTES;
#pragma unroll
        for (int u = 0; u < WSLOTS; ++u)
            if (tid + u * CB_NT < WPIECES) wreg[set][u] = *reinterpret_cast<const u32x4*>(src + (size_t)(tid + u * CB_NT) * 16);
    };
    auto wput = [&](int set, int buf) {
#pragma unroll
        for (int u = 0; u < WSLOTS; ++u)
            if (tid + u * CB_NT < WPIECES) *reinterpret_cast<u32x4*>(Ws + buf * WBYTES + (tid + u * CB_NT) * 16) = wreg[set][u];
    };

    // ---- fragment addresses: lane -> (row lr, column lc) of its run; idle lanes of a 28- / 14-wide run read lane 0's pixel
    const unsigned lr = wino_div((unsigned)li, a.m_rw), lc = (unsigned)li - lr * a.RW;
    const bool lane_live = (int)lr < a.RPT;
    int prow[2];
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        const int rn = wp * 2 + tn;
        prow[tn] = lane_live ? ((rn * a.RPT + (int)lr) * a.PW + (int)lc) : (rn * a.RPT * a.PW);
    }
    const int arow = (wc * 2) * 32 + li;   // + 32 tm

    f32x16 acc[2][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;
    };
    // epilogue of tile t: acc[tm][tn][4 q + e] = channel co0 + 32 tm + 8 q + 4 lh + e of pixel (run wp * 2 + tn, lane li)
    auto store_tile = [&](const Tile& t) {
        const int co0 = t.co_t * TN + wc * 64;
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            const int pr = t.pr0 + (wp * 2 + tn) * a.RPT + (int)lr, xx = t.x0 + (int)lc;
            const unsigned b = wino_div((unsigned)pr, a.m_hp1), y = (unsigned)pr - b * (unsigned)(H + 1);
            if (!lane_live || (int)b >= a.B || (int)y >= H || xx >= W) continue;
            const size_t pix = ((size_t)b * (size_t)(N >> 2) * H + y) * W + xx;   // + quad * H * W, in 16-byte elements
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int co = co0 + 32 * tm + 8 * q + 4 * lh;
                    float4 v = make_float4(acc[tm][tn][4 * q], acc[tm][tn][4 * q + 1], acc[tm][tn][4 * q + 2], acc[tm][tn][4 * q + 3]);
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

    int tile = blockIdx.x;
    if (tile >= a.ntiles) return;
    Tile cur = decode(tile);
    set_patch(cur);
    zero_acc();
    pload(0);
    {
        const int ntile0 = tile + (int)gridDim.x;   // (nsteps >= 9: the first three steps are this tile's)
        (void)ntile0;
        wload(0, cur.co_t, 0);
        wload(1, cur.co_t, 1);
        wload(2, cur.co_t, 2);
    }
    pput();
    wput(0, 0);
    __syncthreads();
    int buf = 0;
    for (;;) {
        const int ntile = tile + (int)gridDim.x;
        const bool has_next = ntile < a.ntiles;
        Tile nxt = cur;
        if (has_next) nxt = decode(ntile);
        int step = 0;
        for (int slab = 0; slab < nslab; ++slab) {
#pragma unroll 1
            for (int tap3 = 0; tap3 < 3; ++tap3) {
#pragma unroll
                for (int j = 0; j < 3; ++j, ++step) {
                    const int tap = 3 * tap3 + j;
                    const bool last = step + 1 == nsteps;           // the tile's last step
                    const bool more = !last || has_next;
                    // set j held this step's weights: they went to the LDS at the end of the previous step -> reload it for step + 3
                    // (ONE load site per register set: two sites that fill the same registers from different branches make hipcc copy the
                    // loaded values where the branches meet -- behind an s_waitcnt vmcnt(0) right after the loads were issued)
                    if (!(CB_ABL & 1)) {
                        const int s3 = step + 3;
                        const bool in_tile = s3 < nsteps;
                        if (in_tile || has_next) wload(j, in_tile ? cur.co_t : nxt.co_t, in_tile ? s3 : s3 - nsteps);
                    }
                    if (tap == 5 && !(CB_ABL & 2)) {   // three taps ahead of the slab (or tile) boundary
                        const bool tile_end = slab + 1 >= nslab;
                        if (tile_end && has_next) set_patch(nxt);
                        if (!tile_end || has_next) pload(tile_end ? 0 : slab + 1);
                    }
                    const int ty = tap3, tx = j;
                    const char* Wc = Ws + buf * WBYTES + arow * CB_PITCH + lh * 16;
                    const char* Pc = Ps + (ty * a.PW + tx) * CB_PITCH + lh * 16;
                    // NWC == 2: all sixteen fragment reads of the tap first, then its 24 MFMAs -- the second k-step's reads land under the
                    // first's MFMAs; NWC == 1 (ten patch slots per thread): one k-step at a time, the registers do not allow more
                    constexpr int KPRE = (CB_KPRE == 2 && NWC == 2) ? 2 : 1;
#pragma unroll
                    for (int k0 = 0; k0 < 2; k0 += KPRE) {
                        bf16x8 ah[KPRE][2], al[KPRE][2], bh[KPRE][2], bl[KPRE][2];
#pragma unroll
                        for (int kq = 0; kq < KPRE; ++kq) {
                            const int kk = k0 + kq;
#pragma unroll
                            for (int tm = 0; tm < 2; ++tm) {
                                const char* p = Wc + tm * 32 * CB_PITCH + kk * 32;
                                if (CB_ABL & 16) {
                                    const u32x4 c = {(unsigned)(tm + kk), (unsigned)lane, 0x3f803f80u, (unsigned)tap};
                                    ah[kq][tm] = __builtin_bit_cast(bf16x8, c); al[kq][tm] = __builtin_bit_cast(bf16x8, c + 1u);
                                    continue;
                                }
                                ah[kq][tm] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p));
                                al[kq][tm] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p + 64));
                            }
#pragma unroll
                            for (int tn = 0; tn < 2; ++tn) {
                                const char* p = Pc + prow[tn] * CB_PITCH + kk * 32;
                                if (CB_ABL & 16) {
                                    const u32x4 c = {(unsigned)(tn + kk), (unsigned)lane, 0x3f803f80u, (unsigned)tap};
                                    bh[kq][tn] = __builtin_bit_cast(bf16x8, c); bl[kq][tn] = __builtin_bit_cast(bf16x8, c + 2u);
                                    continue;
                                }
                                bh[kq][tn] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p));
                                bl[kq][tn] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p + 64));
                            }
                        }
                        if (KPRE == 2) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int kq = 0; kq < KPRE; ++kq) {
                            if (CB_ABL & 4) {
#pragma unroll
                                for (int t2 = 0; t2 < 2; ++t2) asm volatile("" ::"v"(ah[kq][t2]), "v"(al[kq][t2]), "v"(bh[kq][t2]), "v"(bl[kq][t2]));
                                continue;
                            }
#pragma unroll
                            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                                for (int tn = 0; tn < 2; ++tn)
                                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[kq][tm], bh[kq][tn], acc[tm][tn], 0, 0, 0);
#pragma unroll
                            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                                for (int tn = 0; tn < 2; ++tn)
                                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[kq][tm], bl[kq][tn], acc[tm][tn], 0, 0, 0);
#pragma unroll
                            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                                for (int tn = 0; tn < 2; ++tn)
                                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[kq][tm], bh[kq][tn], acc[tm][tn], 0, 0, 0);
                        }
                    }
                    if (more) {
                        if (!(CB_ABL & 1)) wput((j + 1) % 3, buf ^ 1);   // the next step's weights (that image was last read one barrier ago)
                        if (tap == 8 && !(CB_ABL & 2)) {  // slab / tile boundary: the patch image is single -- every wave must be done reading it
                            if (!(CB_ABL & 32)) __syncthreads();
                            pput();
                        }
                        if (!(CB_ABL & 32)) __syncthreads();
                        buf ^= 1;
                    }
                }
            }
        }
        if (!(CB_ABL & 8) || !has_next) store_tile(cur);
        if (!has_next) break;
        zero_acc();
        tile = ntile;
        cur = nxt;
    }
}

// packed[((ct * (C/32) + slab) * 9 + tap) * TN + row][144 B]: row = output channel ct TN + row of the product, k = input channel
// 32 slab + 0..31 as bf16 hi (64 B) then lo (64 B), 16 B of padding.  transpose = 0: the forward's weights w[tap][ci][co];
// 1: the data gradient's -- rows = the forward's INPUT channels, k = its output channels, taps flipped (w[8 - tap][row][k]).
__global__ __launch_bounds__(256) void conv_bx_pack_kernel(const float* __restrict__ w, int C, int N, int transpose, int TN, char* __restrict__ out) {
    const int Cr = transpose ? C : N;   // rows (produced channels)
    const int Ck = transpose ? N : C;   // contracted channels
    const long total = (long)(Cr / TN) * (Ck / 32) * 9 * TN * 9;   // 16-byte pieces: 9 per row
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int piece = (int)(i % 9);
        long t = i / 9;
        const int row = (int)(t % TN); t /= TN;
        const int tap = (int)(t % 9); t /= 9;
        const int slab = (int)(t % (Ck / 32));
        const int ct = (int)(t / (Ck / 32));
        u32x4 v = {0u, 0u, 0u, 0u};
        if (piece < 8) {
            const int lo = piece >> 2, k0 = slab * 32 + (piece & 3) * 8, r = ct * TN + row;
            unsigned o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float f[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int k = k0 + 2 * j + e;
                    f[e] = transpose ? w[((long)(8 - tap) * C + r) * N + k] : w[((long)tap * C + k) * N + r];
                }
                unsigned h, l;
                split_pair(f[0], f[1], h, l);
                o[j] = lo ? l : h;
            }
            v = u32x4{o[0], o[1], o[2], o[3]};
        }
        *reinterpret_cast<u32x4*>(out + i * 16) = v;
    }
}

struct ConvBxPlan {
    int nwc, nw, RW, RPT;
};
static bool plan_conv_bx_nw(int B, int H, int W, int C, int N, int nw, ConvBxPlan& p);
static bool plan_conv_bx(int B, int H, int W, int C, int N, ConvBxPlan& p) {
    static const int force = getenv("VC_CONVBX_WAVES") ? atoi(getenv("VC_CONVBX_WAVES")) : 0;   // experiments: 4 / 8
    ConvBxPlan p8, p4;
    const bool ok8 = plan_conv_bx_nw(B, H, W, C, N, 8, p8), ok4 = plan_conv_bx_nw(B, H, W, C, N, 4, p4);
    if (!ok8 && !ok4) return false;
    bool use4 = !ok8;
    if (ok8 && ok4) {
        if (force == 4) use4 = true;
        else if (force == 8) use4 = false;
        else {
            // eight-wave tiles unless they leave CUs idle: under ~0.8 tiles per CU the four-wave form (twice the tiles, two per CU) wins
            const long t8 = (long)cdiv(W, p8.RW) * cdiv((long)B * (H + 1), 2 * (8 / p8.nwc) * p8.RPT) * (N / (64 * p8.nwc));
            use4 = t8 < 200;
        }
    }
    p = use4 ? p4 : p8;
    return true;
}
static bool plan_conv_bx_nw(int B, int H, int W, int C, int N, int nw, ConvBxPlan& p) {
    if (B < 1 || H < 2 || W < 2 || C % 32 || C < 32 || !(N == 64 || (N >= 128 && N % 128 == 0))) return false;
    p.nwc = N == 64 ? 1 : 2;
    p.nw = nw;
    const int CB_NT = nw * 64;
    // run shape: the candidate with the fewest idle lanes / columns
    const int cand[5][2] = {{32, 1}, {16, 2}, {8, 4}, {28, 1}, {14, 2}};
    double best = 0;
    p.RW = 0;
    for (int i = 0; i < 5; ++i) {
        const int rw = cand[i][0], rpt = cand[i][1];
        if (rw > W && !(rw == 32 && W >= 17)) continue;
        const int runs = 2 * (nw / p.nwc);
        if ((runs * rpt + 2) * (rw + 2) * 8 > cb_maxp(p.nwc, nw) * CB_NT) continue;   // the patch must fit the staging slots
        const double eff = (double)W / (double)(cdiv(W, rw) * rw) * (double)(rw * rpt) / 32.0;
        if (eff > best + 1e-9) { best = eff; p.RW = rw; p.RPT = rpt; }
    }
    return p.RW != 0;
}

template <int KIND>
static int launch_conv_bx(hipStream_t st, int B, int H, int W, int C, int N, const float* x, const void* wp, const float* aux, float* out, int relu,
                          const char* fn) {
    ConvBxPlan p;
    if (!plan_conv_bx(B, H, W, C, N, p)) return fail(VC_EINVAL, "%s: unsupported shape (contracted channels %% 32, produced channels 64 or a multiple of 128; ask vc_conv3x3_bx_supported)", fn);
    const int per = wino_images_per_launch(B, H, W, C, N);
    if (per < 1) return fail(VC_EINVAL, "%s: one image exceeds the 2 GiB offset range", fn);
    for (int b0 = 0; b0 < B; b0 += per) {
        const int nb = B - b0 < per ? B - b0 : per;
        ConvBxArgs a;
        a.x = x + (size_t)b0 * H * W * C; a.wp = (const char*)wp; a.out = out + (size_t)b0 * H * W * N;
        a.aux = (KIND == 1 && aux) ? aux + (size_t)b0 * H * W * N : aux;
        a.B = nb; a.H = H; a.W = W; a.C = C; a.N = N; a.relu = relu;
        const int runs = 2 * (p.nw / p.nwc), CB_NT = p.nw * 64;
        a.RW = p.RW; a.RPT = p.RPT; a.TR = runs * p.RPT;
        a.PW = p.RW + 2; a.NPIX = (a.TR + 2) * a.PW;
        a.col_tiles = cdiv(W, p.RW);
        const int row_tiles = cdiv((long)nb * (H + 1), a.TR);
        a.ptiles = a.col_tiles * row_tiles;
        a.ntiles = a.ptiles * (N / (64 * p.nwc));
        a.m_rw = wino_magic(a.RW); a.m_pw = wino_magic(a.PW); a.m_hp1 = wino_magic(H + 1);
        a.m_coltiles = wino_magic(a.col_tiles); a.m_ptiles = wino_magic(a.ptiles); a.m_npix = wino_magic(a.NPIX);
        const int lds = 2 * 64 * p.nwc * CB_PITCH + a.NPIX * CB_PITCH;
        // persistent grid: one workgroup per CU, every workgroup the same number of tiles (+- 1); VC_CONVBX_GRID overrides (experiments)
        static const int grid_env = getenv("VC_CONVBX_GRID") ? atoi(getenv("VC_CONVBX_GRID")) : 0;
        static const int cus = [] {
            int dev = 0, n = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
            return n;
        }();
        const int wgs_per_cu = (p.nw == 4 && 2 * lds <= 160 * 1024) ? 2 : 1;
        int grid = grid_env > 0 ? grid_env : cus * wgs_per_cu;
        if (grid > a.ntiles) grid = a.ntiles;
        if (a.NPIX * 8 > cb_maxp(p.nwc, p.nw) * CB_NT || lds > 160 * 1024) return fail(VC_EINVAL, "%s: patch does not fit", fn);
        auto go = [&](auto kern) {
            static bool done = false;   // (the generic lambda is instantiated once per kernel type: one flag each)
            if (!done) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); done = true; }
            hipLaunchKernelGGL(kern, dim3(grid), dim3(CB_NT), lds, st, a);
        };
        if (p.nw == 8) { if (p.nwc == 1) go(conv_bx_kernel<KIND, 1, 8>); else go(conv_bx_kernel<KIND, 2, 8>); }
        else { if (p.nwc == 1) go(conv_bx_kernel<KIND, 1, 4>); else go(conv_bx_kernel<KIND, 2, 4>); }
        const int rc = launch_status(fn);
        if (rc) return rc;
    }
    return 0;
}

}  // namespace vc

using namespace vc;

extern "C" int vc_conv3x3_bx_supported(int B, int H, int W, int Cin, int Cout, int dgrad) {
    ConvBxPlan p;
    const int C = dgrad ? Cout : Cin, N = dgrad ? Cin : Cout;   // contracted / produced channels of the launch
    if (!plan_conv_bx(B, H, W, C, N, p)) return 0;
    const int runs = 2 * (p.nw / p.nwc), TR = runs * p.RPT, npix = (TR + 2) * (p.RW + 2);
    if (npix * 8 > cb_maxp(p.nwc, p.nw) * p.nw * 64) return 0;
    if (2 * 64 * p.nwc * CB_PITCH + npix * CB_PITCH > 160 * 1024) return 0;
    return wino_images_per_launch(B, H, W, C, N) >= 1 ? 1 : 0;
}

extern "C" size_t vc_conv3x3_bx_pack_bytes(int Cin, int Cout) { return (size_t)9 * Cin * Cout / 32 * CB_PITCH; }

extern "C" int vc_conv3x3_bx_pack_f32(void* stream, int Cin, int Cout, const float* w, int transpose, void* wp) {
    VC_CHECK_ARG(w && wp && Cin % 32 == 0 && Cout % 32 == 0 && Cin >= 32 && Cout >= 32, "channels must be multiples of 32");
    const int rows = transpose ? Cin : Cout;
    VC_CHECK_ARG(rows == 64 || (rows >= 128 && rows % 128 == 0), "produced channels must be 64 or a multiple of 128");
    VC_CHECK_ARG(waligned16(wp), "wp must be 16-byte aligned");
    const int TN = rows == 64 ? 64 : 128;
    const long total = (long)9 * Cin * Cout / 32 * 9;
    int blocks = cdiv(total, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(conv_bx_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, Cin, Cout, transpose, TN, (char*)wp);
    VC_LAUNCH_CHECK();
    return 0;
}

extern "C" int vc_conv3x3_bx_fwd_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* x, const void* wp, const float* bias,
                                     float* y, int relu) {
    VC_CHECK_ARG(x && wp && y && B > 0, "null operand");
    VC_CHECK_ARG(waligned16(x) && waligned16(y) && waligned16(wp) && (!bias || waligned16(bias)), "operands must be 16-byte aligned");
    return launch_conv_bx<0>((hipStream_t)stream, B, H, W, Cin, Cout, x, wp, bias, y, relu, __func__);
}

extern "C" int vc_conv3x3_bx_dgrad_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* dy, const void* wpt, const float* relu_src,
                                       float* dx) {
    VC_CHECK_ARG(dy && wpt && dx && B > 0, "null operand");
    VC_CHECK_ARG(waligned16(dy) && waligned16(dx) && waligned16(wpt) && (!relu_src || waligned16(relu_src)), "operands must be 16-byte aligned");
    return launch_conv_bx<1>((hipStream_t)stream, B, H, W, Cout, Cin, dy, wpt, relu_src, dx, 0, __func__);
}
