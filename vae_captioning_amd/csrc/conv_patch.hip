// 3x3 / stride 1 / SAME convolution with LDS-staged input PATCHES for gfx950 (forward and data gradient of
// utils/image_embeddings.py:36-212; the data gradient is the same kernel on dy with flipped, transposed weights).
//
//   out[p, n] = sum_{tap, c} in[p + off(tap), c] * Wp[tap][c][n]          M = B*H*W pixels, N columns, K = 9*C
//
// What differs from the implicit-GEMM kernel of conv.hip (which stages an im2col K-tile per tap: every input element
// travels global -> LDS nine times, two barriers per 32-deep K-tile):
//   * A operand: a workgroup stages the halo PATCH of its 128 output pixels for a 32-channel chunk ONCE
//     (one buffer_load_dwordx4 per slot, zero-filled outside the image by the buffer bounds check, no per-tap
//     address arithmetic) and walks the nine taps as constant LDS offsets: two barriers per 288-deep K range
//     (576 MFMAs per wave) instead of two per 32;
//   * B operand: the weights are pre-packed [tap][C/4][N][4] (vc_conv3x3_pack_f32, once per optimiser step) so that
//     a lane's four consecutive k values of one output column are ONE 16-byte load; every wave loads its own B
//     fragments straight into registers (512 contiguous bytes per half-wave), one tap ahead - no LDS, no barrier;
//   * all per-tap addressing is scalar: buffer voffset = per-slot constant, soffset = f(tap, chunk) in an SGPR.
// Pixel tilings (the MFMA row m of a tile -> output pixel):
//   SUB  : four 4 x 8 sub-tiles per workgroup, each with its own 6 x 10 halo patch (W % 8 == 0, H % 4 == 0:
//          the 224 / 112 / 56 layers); sub-tiles are image-local, so pooling windows never straddle a tile;
//   FLAT : 128 consecutive pixels of the flattened [B*H*W] order; the patch is every image row the range touches
//          +-1, row pitch W + 2, with one zero row between two images (the 28 / 14 layers).
#include <stdlib.h>
#include "gemm_core.h"
#include "vaecap.h"

namespace vc {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

enum { PATCH_SUB = 0, PATCH_FLAT = 1 };
enum { PK_FWD = 0, PK_DGRAD = 1 };
constexpr int PITCH = 36;            // floats per patch pixel in LDS (32 channels + 4: conflict-free ds_read_b128 over 16 rows)
constexpr unsigned OOB = 0x80000000u;  // buffer voffset that always fails the bounds check (tensors are < 2 GiB): loads 0

struct PatchGeom {
    int B, H, W, C, N;
    int PW;            // patch row pitch in pixels (SUB: 10, FLAT: W + 2)
    int R;             // FLAT: row slots of the patch
    int subs_x, subs_img;  // SUB: sub-tiles per image row, per image
    long nsubs;        // SUB: B * subs_img
    long P;            // B*H*W
};

struct PatchArgs {
    PatchGeom g;
    const float* x;    // [P, C]
    const float* wp;   // packed [9][C/4][N][4]
    float* out;        // [P, N]
    const float* aux;  // fwd: bias [N] or null; dgrad: ReLU source [P, N] or null
    int relu;
    int tiles_n, ntiles, tile0;
    int nchunks;       // C / 32
    // K-split tail launch (blockIdx.y = split): chunk range of a split and the raw partial sums [split][tail tile][128][BN]
    int chunks_per_split;
    float* tail_ws;
    float* pool;       // forward, SUB tiling only: also write max_pool2x2(out) [B, H/2, W/2, N] (null: no pooling)
    // data gradient of the convolution BEHIND a 2x2 max-pool: the result is the gradient w.r.t. the pooled tensor; with up_y / up_dx
    // set it is routed straight to the arg-max positions of the pool's input up_y [B, 2H, 2W, N] (+ its ReLU gradient) and written
    // to up_dx [B, 2H, 2W, N] -- MaxPoolGrad + ReluGrad without their own launch and without materialising the pooled gradient
    const float* up_y;
    float* up_dx;
};

// MaxPoolGrad (first maximum in (dy,dx) scan order wins; TF-sem.) fused with the ReluGrad of the convolution that produced the
// window a, b (row 0), c, d (row 1): the same rule as conv.hip's maxpool_bwd_kernel
__device__ __forceinline__ void unpool_route(float a, float b, float c, float d, float g, float& oa, float& ob, float& oc, float& od) {
    const float m = fmaxf(fmaxf(a, b), fmaxf(c, d));
    const int w = (a == m) ? 0 : (b == m) ? 1 : (c == m) ? 2 : 3;
    const float v = !(m > 0.f) ? 0.f : g;
    oa = w == 0 ? v : 0.f; ob = w == 1 ? v : 0.f; oc = w == 2 ? v : 0.f; od = w == 3 ? v : 0.f;
}
__device__ __forceinline__ void unpool_store(const float* __restrict__ yb, float* __restrict__ db, long rowstride, int N, const float4& g) {
    const float4 a = *reinterpret_cast<const float4*>(yb), b = *reinterpret_cast<const float4*>(yb + N);
    const float4 c = *reinterpret_cast<const float4*>(yb + rowstride), d = *reinterpret_cast<const float4*>(yb + rowstride + N);
    float4 oa, ob, oc, od;
    unpool_route(a.x, b.x, c.x, d.x, g.x, oa.x, ob.x, oc.x, od.x);
    unpool_route(a.y, b.y, c.y, d.y, g.y, oa.y, ob.y, oc.y, od.y);
    unpool_route(a.z, b.z, c.z, d.z, g.z, oa.z, ob.z, oc.z, od.z);
    unpool_route(a.w, b.w, c.w, d.w, g.w, oa.w, ob.w, oc.w, od.w);
    *reinterpret_cast<float4*>(db) = oa; *reinterpret_cast<float4*>(db + N) = ob;
    *reinterpret_cast<float4*>(db + rowstride) = oc; *reinterpret_cast<float4*>(db + rowstride + N) = od;
}

// ---- tile geometry ------------------------------------------------------------------------------
template <int SCHEME>
struct TileMap {
    const PatchGeom& g;
    int tm;
    // FLAT
    long p0;
    int g0, b0, nb0;
    __device__ __forceinline__ TileMap(const PatchGeom& g_, int tm_) : g(g_), tm(tm_) {
        if (SCHEME == PATCH_FLAT) {
            p0 = (long)tm * 128;
            g0 = (int)(p0 / g.W);
            b0 = g0 / g.H;
            nb0 = g.H - (g0 - b0 * g.H);
        }
    }
    // output pixel of tile row r (-1: beyond the tensor)
    __device__ __forceinline__ long out_pixel(int r) const {
        if (SCHEME == PATCH_SUB) {  // (32-bit unsigned arithmetic: sub-tile and pixel counts are < 2^31; 64-bit divisions cost ~100 VALU each)
            const unsigned s = (unsigned)tm * 4u + (unsigned)(r >> 5);
            if (s >= (unsigned)g.nsubs) return -1;
            const unsigned b = s / (unsigned)g.subs_img, rem = s - b * (unsigned)g.subs_img;
            const unsigned sy = rem / (unsigned)g.subs_x, sx = rem - sy * (unsigned)g.subs_x;
            const unsigned y = sy * 4u + ((unsigned)(r & 31) >> 3), x = sx * 8u + (unsigned)(r & 7);
            return (long)((b * (unsigned)g.H + y) * (unsigned)g.W + x);
        } else {
            const long p = p0 + r;
            return p < g.P ? p : -1;
        }
    }
    // patch pixel (LDS index) of the CENTRE tap of tile row m
    __device__ __forceinline__ int centre(int m) const {
        if (SCHEME == PATCH_SUB) {
            return (m >> 5) * 60 + (((m & 31) >> 3) + 1) * 10 + (m & 7) + 1;
        } else {
            long p = p0 + m;
            if (p >= g.P) p = g.P - 1;
            const int gr = (int)(p / g.W), x = (int)(p - (long)gr * g.W);
            const int bb = gr / g.H;
            return ((gr - g0) + 1 + (bb - b0)) * g.PW + x + 1;
        }
    }
    // global pixel index stored in patch slot `pix` (-1: zero)
    __device__ __forceinline__ long source(int pix) const {
        if (SCHEME == PATCH_SUB) {
            if (pix >= 240) return -1;
            const int j = pix / 60, r60 = pix - j * 60;
            const int py = r60 / 10, px = r60 - py * 10;
            const unsigned s = (unsigned)tm * 4u + (unsigned)j;
            if (s >= (unsigned)g.nsubs) return -1;
            const unsigned b = s / (unsigned)g.subs_img, rem = s - b * (unsigned)g.subs_img;
            const unsigned sy = rem / (unsigned)g.subs_x, sx = rem - sy * (unsigned)g.subs_x;
            const int y = (int)(sy * 4u) + py - 1, x = (int)(sx * 8u) + px - 1;
            if ((unsigned)y >= (unsigned)g.H || (unsigned)x >= (unsigned)g.W) return -1;
            return (long)((b * (unsigned)g.H + (unsigned)y) * (unsigned)g.W + (unsigned)x);
        } else {
            const int rs = pix / g.PW, x = pix - rs * g.PW - 1;
            if (rs >= g.R || (unsigned)x >= (unsigned)g.W) return -1;
            int gr, bexp;
            if (rs <= nb0) { gr = g0 + rs - 1; bexp = b0; }
            else if (rs == nb0 + 1) return -1;       // the zero row between two images
            else { gr = g0 + rs - 2; bexp = b0 + 1; }
            if (gr < 0 || gr >= g.B * g.H || gr / g.H != bexp) return -1;
            return (long)gr * g.W + x;
        }
    }
};

__device__ __forceinline__ float4 bufload(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
    return *reinterpret_cast<float4*>(&v);
}

__device__ __forceinline__ float comp(const float4& v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; }

// value of lane ^ 1 (quad_perm [1,0,3,2]) / of lane ^ 8 (rotation by 8 inside the 16-lane DPP row): VALU modifiers, no LDS
__device__ __forceinline__ float dpp_xor1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_ror8(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));
}

template <int TN>
using PatchCfg = TileCfg<2, 2, 2, TN>;  // 4 waves, wave tile 64 x (TN*32): block 128 x (TN*64)

template <int SCHEME>
struct PatchLds {
    static constexpr int NVP = SCHEME == PATCH_SUB ? 8 : 9;  // float4 patch slots per thread (256 threads x 8 channel quads)
    static constexpr int PIX = NVP * 32;
    template <int TN>
    static constexpr int bytes() {
        return (PIX * PITCH > PatchCfg<TN>::EPI_FLOATS ? PIX * PITCH : PatchCfg<TN>::EPI_FLOATS) * 4;
    }
};

// tools/probes/patch_trace.hip compiles this file with VC_PATCH_TRACE: every wave stamps s_memtime at phase edges
#ifdef VC_PATCH_TRACE
__device__ unsigned long long* g_patch_trace = nullptr;  // [workgroups][4 waves][8 stamps]
#define PATCH_STAMP(k)                                                                                              \
    do {                                                                                                            \
        if (g_patch_trace && (threadIdx.x & 63) == 0)                                                               \
            g_patch_trace[((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (k)] = __builtin_readcyclecounter();    \
    } while (0)
#else
#define PATCH_STAMP(k)
#endif

template <int TN, int SCHEME, int KIND, bool POOL = false, bool UNPOOL = false>
__global__ __launch_bounds__(256, TN == 1 ? 3 : 2) void conv_patch_kernel(PatchArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using CFG = PatchCfg<TN>;
    constexpr int NVP = PatchLds<SCHEME>::NVP;
    const PatchGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
    PATCH_STAMP(0);
    const int id = a.tile0 + xcd_remap(blockIdx.x, a.ntiles);
    const int tmi = id / a.tiles_n, n0 = (id - tmi * a.tiles_n) * CFG::BN;
    const TileMap<SCHEME> map(g, tmi);
    const int C = g.C, N = g.N;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)(g.P * C * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.wp, 0, 9 * C * N * 4, 0x00020000);

    // patch slots of this thread: slot u = (patch pixel (tid >> 3) + 32 u, channel quad tid & 7)
    unsigned voff[NVP];
#pragma unroll
    for (int u = 0; u < NVP; ++u) {
        const long gp = map.source((tid >> 3) + 32 * u);
        voff[u] = gp >= 0 ? (unsigned)((gp * C + (tid & 7) * 4) * 4) : OOB;
    }
    const int pst = (tid >> 3) * PITCH + (tid & 7) * 4;  // LDS float index of slot 0; slot u is 32*PITCH further
    // A fragment bases: tile row (wm*2 + t)*32 + li, lane half lh takes channels [16 lh, 16 lh + 16) of the chunk
    int abase[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) abase[t] = map.centre((wm * 2 + t) * 32 + li) * PITCH + lh * 16;
    // B fragment: packed weights, column n0 + (wn*TN + t)*32 + li, channel quad 4 lh + q of the chunk
    unsigned voffb[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t) voffb[t] = (unsigned)(((long)(4 * lh) * N + n0 + (wn * TN + t) * 32 + li) * 16);
    const unsigned qstride = (unsigned)N * 16u;  // bytes between channel quads

    // Accumulators start at the BIAS of their output column (forward, main launch: the K-split tail launch writes raw partial sums
    // and its reduce kernel adds the bias); the data gradient packs the ReLU mask of its rows into one bit per accumulator element
    // here, while the first patch is in flight -- the epilogue then has no load on its critical path.
    f32x16 acc[2][TN];
    acc_zero<CFG>(acc);
    unsigned mbits[TN];
#pragma unroll
    for (int u = 0; u < TN; ++u) mbits[u] = 0xffffffffu;
    if (!a.tail_ws && a.aux) {
        if (KIND == PK_FWD) {
#pragma unroll
            for (int u = 0; u < TN; ++u)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 bv = *reinterpret_cast<const float4*>(a.aux + n0 + (wn * TN + u) * 32 + 8 * q + 4 * lh);
#pragma unroll
                    for (int t = 0; t < 2; ++t) { acc[t][u][4 * q] = bv.x; acc[t][u][4 * q + 1] = bv.y; acc[t][u][4 * q + 2] = bv.z; acc[t][u][4 * q + 3] = bv.w; }
                }
        } else {
#pragma unroll
            for (int u = 0; u < TN; ++u) mbits[u] = 0u;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const long p = map.out_pixel((wm * 2 + t) * 32 + li);
                const long pc = p >= 0 ? p : 0;
#pragma unroll
                for (int u = 0; u < TN; ++u)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 m = *reinterpret_cast<const float4*>(a.aux + pc * N + n0 + (wn * TN + u) * 32 + 8 * q + 4 * lh);
                        const unsigned bits = (m.x > 0.f ? 1u : 0u) | (m.y > 0.f ? 2u : 0u) | (m.z > 0.f ? 4u : 0u) | (m.w > 0.f ? 8u : 0u);
                        mbits[u] |= bits << (16 * t + 4 * q);
                    }
            }
        }
    }

    int cb = 0, ce = a.nchunks;
    if (a.tail_ws) {
        cb = blockIdx.y * a.chunks_per_split;
        ce = cb + a.chunks_per_split < a.nchunks ? cb + a.chunks_per_split : a.nchunks;
    }

    float4 pr[NVP];
    float4 b0[TN][4], b1[TN][4];
    auto pload = [&](int ch) {
#pragma unroll
        for (int u = 0; u < NVP; ++u) pr[u] = bufload(rx, voff[u], (unsigned)ch * 128u);
    };
    auto pstore = [&]() {
#pragma unroll
        for (int u = 0; u < NVP; ++u) *reinterpret_cast<float4*>(&smem[pst + u * 32 * PITCH]) = pr[u];
    };
    auto bload = [&](float4 (&b)[TN][4], int tap, int ch) {
        const unsigned s0 = (unsigned)(tap * (C >> 2) + ch * 8) * qstride;
#pragma unroll
        for (int t = 0; t < TN; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) b[t][q] = bufload(rw, voffb[t], s0 + (unsigned)q * qstride);
    };
    // one tap of one chunk: 32-deep contraction, A fragments read from the patch at a constant offset
    auto compute = [&](const float4 (&b)[TN][4], int tapoff) {
        float fa[2][2][4];
        auto frag = [&](int c, int buf) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const float4 v = *reinterpret_cast<const float4*>(&smem[abase[t] + tapoff + c * 4]);
                fa[buf][t][0] = v.x; fa[buf][t][1] = v.y; fa[buf][t][2] = v.z; fa[buf][t][3] = v.w;
            }
        };
        // (sched_barrier: hipcc otherwise sinks every load to just before its first use -- it minimises registers --
        //  which exposes the L2 / LDS latency once per 16 MFMAs; the barriers pin the software pipeline as written)
        frag(0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c + 1 < 4) frag(c + 1, (c + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int u = 0; u < TN; ++u)
                        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(b[u][c], e), fa[c & 1][t][e], acc[t][u], 0, 0, 0);
        }
    };
    const int PWp = (SCHEME == PATCH_SUB ? 10 : g.PW) * PITCH;  // floats per patch row (SUB: compile-time -> immediate offsets)
    auto tapoff = [&](int tap) { return (tap / 3 - 1) * PWp + (tap % 3 - 1) * PITCH; };

    PATCH_STAMP(1);
    if (cb < ce) {
        pload(cb);
        bload(b0, 0, cb);
        pstore();
        PATCH_STAMP(2);
        __syncthreads();
    }
    PATCH_STAMP(3);
    for (int ch = cb; ch < ce; ++ch) {
        const bool more = ch + 1 < ce;
        bload(b1, 1, ch); compute(b0, tapoff(0));
        bload(b0, 2, ch); compute(b1, tapoff(1));
        bload(b1, 3, ch); compute(b0, tapoff(2));
        bload(b0, 4, ch); compute(b1, tapoff(3));
        bload(b1, 5, ch); compute(b0, tapoff(4));
        bload(b0, 6, ch); compute(b1, tapoff(5));
        bload(b1, 7, ch); compute(b0, tapoff(6));
        bload(b0, 8, ch);
        if (more) pload(ch + 1);       // next chunk's patch: in flight under the last two taps
        compute(b1, tapoff(7));
        if (more) bload(b1, 0, ch + 1);
        compute(b0, tapoff(8));
        if (more) {
            __syncthreads();           // every wave is done reading this chunk's patch
            pstore();
#pragma unroll
            for (int t = 0; t < TN; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) b0[t][q] = b1[t][q];
            __syncthreads();
        } else {
            // (hipcc 7.2 has scheduled register copies of a loop exit straight behind the last 16-pass MFMA without its wait states,
            //  see gemm notes in DESIGN.md: the epilogue below reads the accumulators right away)
            asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
        }
    }

    PATCH_STAMP(4);
    // The weights are the MFMA's A operand, the patch its B operand: acc[t][u][r] = D[output column 32 u + (r&3) + 8 (r>>2) + 4 lh]
    // [tile row (pixel) 32 t + li] -- a lane owns ONE pixel and, in every group of four accumulator registers, four CONSECUTIVE
    // output columns: bias / ReLU / ReLU mask and the 16-byte stores come straight from the accumulators, with no LDS transposition and
    // no workgroup barrier (round 2's first version transposed through LDS: ~8 % of a 64-channel tile).
    if (a.tail_ws) {  // raw partial sums of this split: [split][tail tile][128][BN]
        float* o = a.tail_ws + ((long)blockIdx.y * a.ntiles + (id - a.tile0)) * (128 * CFG::BN);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < TN; ++u)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(o + ((wm * 2 + t) * 32 + li) * CFG::BN + (wn * TN + u) * 32 + 8 * q + 4 * lh) =
                        make_float4(acc[t][u][4 * q], acc[t][u][4 * q + 1], acc[t][u][4 * q + 2], acc[t][u][4 * q + 3]);
        return;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const long p = map.out_pixel((wm * 2 + t) * 32 + li);
        const bool live = p >= 0;
        const long pc = live ? p : 0;
        float* pool_row = nullptr;
        bool pool_lane = false;
        const float* up_y = nullptr;
        float* up_dx = nullptr;
        long up_stride = 0;
        if (UNPOOL) {  // top-left pixel of this (pooled) pixel's window in the [B, 2H, 2W, N] tensors
            const long prow = pc / g.W;                      // b*H + y
            const long base = ((2 * prow) * (2L * g.W) + 2 * (pc - prow * g.W)) * N;
            up_y = a.up_y + base;
            up_dx = a.up_dx + base;
            up_stride = 2L * g.W * N;
        }
        if (POOL) {
            // 2x2 / stride 2 max-pool (utils/image_embeddings.py:59-63 ...): the 32 pixels of this MFMA tile are one 4 x 8 sub-tile, pixel
            // li = 8 row + column, so a pooling window is the lanes {li, li ^ 1, li ^ 8}: two DPP max steps; max commutes with the bias
            // add and the ReLU (both monotone)
            const long prow = pc / g.W;                      // b*H + y (H even): the pooled row is prow / 2
            const int px = (int)(pc - prow * g.W) >> 1;
            pool_row = a.pool + ((prow >> 1) * (g.W >> 1) + px) * N;
            pool_lane = live && (li & 9) == 0;
        }
        if (UNPOOL) {
            // all sixteen window loads of a 32-column strip in flight together (the registers of the weight / patch pipeline are free
            // here), then route + store: one memory latency per strip instead of one per register quad
#pragma unroll
            for (int u = 0; u < TN; ++u) {
                float4 wa[4], wb[4], wc[4], wd[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float* yq = up_y + n0 + (wn * TN + u) * 32 + 8 * q + 4 * lh;
                    wa[q] = live ? *reinterpret_cast<const float4*>(yq) : f4zero();
                    wb[q] = live ? *reinterpret_cast<const float4*>(yq + N) : f4zero();
                    wc[q] = live ? *reinterpret_cast<const float4*>(yq + up_stride) : f4zero();
                    wd[q] = live ? *reinterpret_cast<const float4*>(yq + up_stride + N) : f4zero();
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float* dq = up_dx + n0 + (wn * TN + u) * 32 + 8 * q + 4 * lh;
                    const float4 gv = make_float4(acc[t][u][4 * q], acc[t][u][4 * q + 1], acc[t][u][4 * q + 2], acc[t][u][4 * q + 3]);
                    float4 oa, ob, oc, od;
                    unpool_route(wa[q].x, wb[q].x, wc[q].x, wd[q].x, gv.x, oa.x, ob.x, oc.x, od.x);
                    unpool_route(wa[q].y, wb[q].y, wc[q].y, wd[q].y, gv.y, oa.y, ob.y, oc.y, od.y);
                    unpool_route(wa[q].z, wb[q].z, wc[q].z, wd[q].z, gv.z, oa.z, ob.z, oc.z, od.z);
                    unpool_route(wa[q].w, wb[q].w, wc[q].w, wd[q].w, gv.w, oa.w, ob.w, oc.w, od.w);
                    if (live) {
                        *reinterpret_cast<float4*>(dq) = oa; *reinterpret_cast<float4*>(dq + N) = ob;
                        *reinterpret_cast<float4*>(dq + up_stride) = oc; *reinterpret_cast<float4*>(dq + up_stride + N) = od;
                    }
                }
            }
            continue;
        }
#pragma unroll
        for (int u = 0; u < TN; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = n0 + (wn * TN + u) * 32 + 8 * q + 4 * lh;
                float4 v = make_float4(acc[t][u][4 * q], acc[t][u][4 * q + 1], acc[t][u][4 * q + 2], acc[t][u][4 * q + 3]);
                if (KIND == PK_FWD) {  // (the bias is already in the accumulators)
                    if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                } else {               // ReluGrad of the layer that produced this convolution's input (mask bits packed before the main loop)
                    const unsigned mb = mbits[u] >> (16 * t + 4 * q);
                    if (!(mb & 1u)) v.x = 0.f;
                    if (!(mb & 2u)) v.y = 0.f;
                    if (!(mb & 4u)) v.z = 0.f;
                    if (!(mb & 8u)) v.w = 0.f;
                }
                if (live) *reinterpret_cast<float4*>(a.out + p * N + col) = v;
                if (POOL) {
                    float4 m = v;
                    m.x = fmaxf(m.x, dpp_xor1(m.x)); m.y = fmaxf(m.y, dpp_xor1(m.y)); m.z = fmaxf(m.z, dpp_xor1(m.z)); m.w = fmaxf(m.w, dpp_xor1(m.w));
                    m.x = fmaxf(m.x, dpp_ror8(m.x)); m.y = fmaxf(m.y, dpp_ror8(m.y)); m.z = fmaxf(m.z, dpp_ror8(m.z)); m.w = fmaxf(m.w, dpp_ror8(m.w));
                    if (pool_lane) *reinterpret_cast<float4*>(pool_row + col) = m;
                }
            }
    }
    PATCH_STAMP(5);
}

// Sum of the tail launch's K splits (fixed order) + the epilogue of the main launch.
template <int SCHEME, int KIND, bool UNPOOL = false>
__global__ __launch_bounds__(256) void patch_tail_reduce_kernel(PatchArgs a, int splits, int BN) {
    const PatchGeom& g = a.g;
    const int qpr = BN >> 2;                         // float4 per tile row
    const long per_split = (long)a.ntiles * 128 * qpr;
    const float4* ws = reinterpret_cast<const float4*>(a.tail_ws);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < per_split; i += (long)gridDim.x * 256) {
        float4 v = ws[i];
        for (int z = 1; z < splits; ++z) {
            const float4 t = ws[(long)z * per_split + i];
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        const int cq = (int)(i % qpr);
        const long tr = i / qpr;
        const int r = (int)(tr & 127);
        const int id = a.tile0 + (int)(tr >> 7);
        const int tmi = id / a.tiles_n, n0 = (id - tmi * a.tiles_n) * BN;
        const TileMap<SCHEME> map(g, tmi);
        const long p = map.out_pixel(r);
        if (p < 0) continue;
        const int col = n0 + cq * 4;
        if (KIND == PK_FWD) {
            if (a.aux) {
                const float4 bv = *reinterpret_cast<const float4*>(a.aux + col);
                v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
            }
            if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        } else if (a.aux) {
            const float4 m = *reinterpret_cast<const float4*>(a.aux + p * g.N + col);
            if (!(m.x > 0.f)) v.x = 0.f;
            if (!(m.y > 0.f)) v.y = 0.f;
            if (!(m.z > 0.f)) v.z = 0.f;
            if (!(m.w > 0.f)) v.w = 0.f;
        }
        if (UNPOOL) {
            const long prow = p / g.W;
            const long base = ((2 * prow) * (2L * g.W) + 2 * (p - prow * g.W)) * g.N + col;
            unpool_store(a.up_y + base, a.up_dx + base, 2L * g.W * g.N, g.N, v);
            continue;
        }
        *reinterpret_cast<float4*>(a.out + p * g.N + col) = v;
        if (KIND == PK_FWD && SCHEME == PATCH_SUB && a.pool && !(r & 1) && !(r & 8)) {  // top-left pixel of a pooling window
            // raw sums of the three other pixels of the window: rows r+1, r+8, r+9 of the same tile
            float4 m = ws[i];
            for (int z = 1; z < splits; ++z) { const float4 t = ws[(long)z * per_split + i]; m.x += t.x; m.y += t.y; m.z += t.z; m.w += t.w; }
            const long offs[3] = {(long)qpr, 8L * qpr, 9L * qpr};
            for (int k = 0; k < 3; ++k) {
                float4 s = ws[i + offs[k]];
                for (int z = 1; z < splits; ++z) { const float4 t = ws[(long)z * per_split + i + offs[k]]; s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
                m.x = fmaxf(m.x, s.x); m.y = fmaxf(m.y, s.y); m.z = fmaxf(m.z, s.z); m.w = fmaxf(m.w, s.w);
            }
            if (a.aux) {
                const float4 bv = *reinterpret_cast<const float4*>(a.aux + col);
                m.x += bv.x; m.y += bv.y; m.z += bv.z; m.w += bv.w;
            }
            if (a.relu) { m.x = fmaxf(m.x, 0.f); m.y = fmaxf(m.y, 0.f); m.z = fmaxf(m.z, 0.f); m.w = fmaxf(m.w, 0.f); }
            const long prow = p / g.W;
            const int px = (int)(p - prow * g.W) >> 1;
            *reinterpret_cast<float4*>(a.pool + ((prow >> 1) * (g.W >> 1) + px) * g.N + col) = m;
        }
    }
}

// w [9][Ci][Co] (HWIO) -> packed [tap][C/4][N][4]:
//   transpose 0 (forward):        C = Ci, N = Co, out[tap][ci/4][co][ci%4] = w[tap][ci][co]
//   transpose 1 (data gradient):  C = Co, N = Ci, out[tap][co/4][ci][co%4] = w[8 - tap][ci][co]   (flipped taps)
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ w, int Ci, int Co, int transpose,
                                                           float4* __restrict__ out) {
    const int C = transpose ? Co : Ci, N = transpose ? Ci : Co;
    const long total = 9L * (C >> 2) * N;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int n = (int)(i % N);
        const long t = i / N;
        const int cq = (int)(t % (C >> 2)), tap = (int)(t / (C >> 2));
        float4 v;
        if (!transpose) {
            const float* s = w + ((long)tap * Ci + cq * 4) * Co + n;
            v = make_float4(s[0], s[Co], s[2L * Co], s[3L * Co]);
        } else {
            v = *reinterpret_cast<const float4*>(w + ((long)(8 - tap) * Ci + n) * Co + cq * 4);
        }
        out[i] = v;
    }
}

// ---- host side ----------------------------------------------------------------------------------
struct PatchPlan {
    int scheme;      // -1: shape not supported by the patch kernels
    int TN;
    PatchGeom g;
    int tiles_m, tiles_n;
};

constexpr int PATCH_CUS = 256;

// Estimated time of a launch of T tiles of nch 32-channel chunks, in units of one chunk of ONE tile (see the measurements at
// plan_patch_tail): whole rounds of one tile per CU + the cheapest way to run the remaining tiles (unsplit, or their chunk range
// split into `cps`-chunk pieces that cover all CUs again + partial-sum traffic + one more launch).
static double patch_launch_cost(int T, int nch, int* cps_out) {
    const int tail = T % PATCH_CUS;
    double best = tail ? nch + 0.2 : 0.0;
    int best_cps = nch;
    if (tail && nch >= 2)
        for (int cps = 1; cps < nch; ++cps) {
            const int splits = cdiv(nch, cps);
            const double cost = cdiv((long)tail * splits, PATCH_CUS) * (cps + 0.2) + 0.002 * splits * tail + 0.6;
            if (cost < best - 1e-9) { best = cost; best_cps = cps; }
        }
    if (cps_out) *cps_out = best_cps;
    return (T / PATCH_CUS) * (nch + 0.2) + best;
}

static PatchPlan plan_patch(int B, int H, int W, int C, int N) {
    PatchPlan p;
    p.scheme = -1;
    PatchGeom& g = p.g;
    g.B = B; g.H = H; g.W = W; g.C = C; g.N = N;
    g.P = (long)B * H * W;
    g.R = 0; g.subs_x = 0; g.subs_img = 0; g.nsubs = 0; g.PW = 0;
    if (B <= 0 || H <= 0 || W <= 0 || C % 32 || N % 64 || C <= 0 || N <= 0) return p;
    if (g.P * (long)(C > N ? C : N) * 4 > 0x7fffffffL) return p;  // 32-bit buffer offsets, OOB marker at 2 GiB
    if (W % 8 == 0 && H % 4 == 0) {
        p.scheme = PATCH_SUB;
        g.PW = 10;
        g.subs_x = W / 8;
        g.subs_img = (H / 4) * (W / 8);
        g.nsubs = (long)B * g.subs_img;
        p.tiles_m = cdiv(g.nsubs, 4);
    } else {
        const int rows = (W - 1 + 127) / W + 1;  // image rows 128 consecutive pixels can touch
        g.PW = W + 2;
        g.R = rows + 3;                           // + halo above / below + one zero row between two images
        if ((long)H * W < 128 || g.R * g.PW > PatchLds<PATCH_FLAT>::PIX) return p;
        p.scheme = PATCH_FLAT;
        p.tiles_m = cdiv(g.P, 128);
    }
    // 128 x 128 or 128 x 64 tiles: both run at the same steady-state rate once a tile has >= 4 chunks (measured: conv4_2 146.3 vs
    // 146.8 TFLOP/s), so the narrower tile is taken whenever its finer launch quantum wastes less of the last round (conv5_x at 64
    // images: 392 wide tiles = 1.53 per CU -> 126 TFLOP/s; 784 narrow tiles = 3 rounds + a 16-tile split tail -> 141).  A chunk of
    // a wide tile costs two units.  With C = 64 (two chunks per tile) the wide tile's fewer prologues win.
    p.TN = N % 128 == 0 ? 2 : 1;
    if (p.TN == 2 && C >= 128) {
        const double wide = 2.0 * patch_launch_cost(p.tiles_m * (N / 128), C / 32, nullptr);
        const double narrow = patch_launch_cost(p.tiles_m * (N / 64), C / 32, nullptr);
        if (narrow < wide - 1e-9) p.TN = 1;
    }
    if (const char* e = getenv("VC_PATCH_TN")) {  // experiments only (tools/microbench.py)
        if (atoi(e) == 1) p.TN = 1;
        if (atoi(e) == 2 && N % 128 == 0) p.TN = 2;
    }
    p.tiles_n = N / (p.TN * 64);
    return p;
}

// Launch geometry.  Measured on MI355X (tools/microbench.py convsweep, conv4_2 shape, tiles of 16 chunks): a launch of T
// tiles takes ceil(T / 256) x 0.2625 ms -- ONE workgroup already keeps its CU's matrix pipes ~94 % busy (a lone tile
// 0.278 ms, two co-resident tiles 0.525 ms), so the scheduling quantum is one tile per CU, not one per resident slot:
// 248 tiles 0.278 ms, 296 tiles 0.524 ms, 492 tiles 0.528 ms, 516 tiles 0.787 ms.  The main launch therefore takes
// floor(T / 256) * 256 tiles; the remaining tiles run as a second launch whose K range (the 32-channel chunks) is split
// so that the short workgroups again cover all 256 CUs evenly, and patch_tail_reduce_kernel sums the splits in fixed
// order and applies the epilogue.  The split is chosen by the cost model below (unit: the time of one chunk).
struct PatchTail {
    int main_tiles, tail_tiles, splits, cps;
};

static PatchTail plan_patch_tail(const PatchPlan& p) {
    PatchTail t;
    const int T = p.tiles_m * p.tiles_n;
    const int nch = p.g.C / 32;
    const int tail = T % PATCH_CUS;
    t.main_tiles = T; t.tail_tiles = 0; t.splits = 1; t.cps = nch;
    int cps = nch;
    patch_launch_cost(T, nch, &cps);
    if (tail == 0 || cps == nch) return t;
    t.cps = cps;
    t.splits = cdiv(nch, cps);
    t.main_tiles = T - tail;
    t.tail_tiles = tail;
    return t;
}

static size_t patch_workspace(const PatchPlan& p) {
    if (p.scheme < 0) return 0;
    const PatchTail t = plan_patch_tail(p);
    return t.tail_tiles ? (size_t)t.splits * t.tail_tiles * 128 * (p.TN * 64) * sizeof(float) : 0;
}

template <int TN, int SCHEME, int KIND, bool POOL = false, bool UNPOOL = false>
static int launch_patch(hipStream_t st, const PatchPlan& p, PatchArgs& a, float* ws, size_t ws_bytes) {
    const PatchTail t = plan_patch_tail(p);
    constexpr int smem = PatchLds<SCHEME>::template bytes<TN>();
    a.tiles_n = p.tiles_n;
    a.nchunks = p.g.C / 32;
    a.tile0 = 0; a.tail_ws = nullptr; a.chunks_per_split = 0;
    const size_t need = (size_t)t.splits * t.tail_tiles * 128 * (TN * 64) * sizeof(float);
    if (t.tail_tiles == 0 || !ws || ws_bytes < need) {
        a.ntiles = p.tiles_m * p.tiles_n;
        hipLaunchKernelGGL((conv_patch_kernel<TN, SCHEME, KIND, POOL, UNPOOL>), dim3(a.ntiles, 1), dim3(256), smem, st, a);
        return launch_status("conv patch");
    }
    if (t.main_tiles > 0) {
        a.ntiles = t.main_tiles;
        hipLaunchKernelGGL((conv_patch_kernel<TN, SCHEME, KIND, POOL, UNPOOL>), dim3(a.ntiles, 1), dim3(256), smem, st, a);
        if (int e = launch_status("conv patch")) return e;
    }
    PatchArgs d = a;
    d.tile0 = t.main_tiles; d.ntiles = t.tail_tiles; d.chunks_per_split = t.cps; d.tail_ws = ws;
    hipLaunchKernelGGL((conv_patch_kernel<TN, SCHEME, KIND, false, false>), dim3(d.ntiles, t.splits), dim3(256), smem, st, d);
    if (int e = launch_status("conv patch tail")) return e;
    const long items = (long)t.tail_tiles * 128 * (TN * 16);
    int grid = (int)((items + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL((patch_tail_reduce_kernel<SCHEME, KIND, UNPOOL>), dim3(grid), dim3(256), 0, st, d, t.splits, TN * 64);
    return launch_status("conv patch tail reduce");
}

template <int KIND>
static int dispatch_patch(hipStream_t st, const PatchPlan& p, PatchArgs& a, float* ws, size_t ws_bytes) {
    if (KIND == PK_DGRAD && a.up_y) {  // MaxPoolGrad + ReluGrad fused into the data gradient's epilogue
        if (p.scheme == PATCH_SUB)
            return p.TN == 2 ? launch_patch<2, PATCH_SUB, PK_DGRAD, false, true>(st, p, a, ws, ws_bytes) : launch_patch<1, PATCH_SUB, PK_DGRAD, false, true>(st, p, a, ws, ws_bytes);
        return p.TN == 2 ? launch_patch<2, PATCH_FLAT, PK_DGRAD, false, true>(st, p, a, ws, ws_bytes) : launch_patch<1, PATCH_FLAT, PK_DGRAD, false, true>(st, p, a, ws, ws_bytes);
    }
    if (KIND == PK_FWD && a.pool)  // fused 2x2 max-pool: 4 x 8 sub-tile tiling only (checked by the caller)
        return p.TN == 2 ? launch_patch<2, PATCH_SUB, PK_FWD, true>(st, p, a, ws, ws_bytes) : launch_patch<1, PATCH_SUB, PK_FWD, true>(st, p, a, ws, ws_bytes);
    if (p.scheme == PATCH_SUB)
        return p.TN == 2 ? launch_patch<2, PATCH_SUB, KIND>(st, p, a, ws, ws_bytes) : launch_patch<1, PATCH_SUB, KIND>(st, p, a, ws, ws_bytes);
    return p.TN == 2 ? launch_patch<2, PATCH_FLAT, KIND>(st, p, a, ws, ws_bytes) : launch_patch<1, PATCH_FLAT, KIND>(st, p, a, ws, ws_bytes);
}


// =====================================================================================================================
// Weight gradient with LDS-staged patches:  dW[tap][ci][co] = sum_p x[p + off(tap)][ci] * dy[p][co]
//   M = (tap, ci), N = co, K = pixels.  A workgroup owns 64 input channels x ALL NINE taps x 64 output channels
//   (36 MFMA tiles of 32 x 32: wave w = (ci half w >> 1, co half w & 1) keeps the nine taps of its 32 x 32 block in
//   nine accumulators) and walks a range of TR x TC pixel sub-tiles = K-tiles: 4 x 8 for the 224 / 112 / 56-wide layers, 4 x 7 for
//   the 28-wide and 2 x 14 for the 14-wide ones (round 2 first gave those a FLAT kernel -- 32 consecutive pixels of the flattened
//   order with a row-pitched patch of 180 pixels and a per-k-step index table: 124-127 TFLOP/s; the 6 x 9 / 4 x 16 patches of an
//   exact 2-D tiling reach 139-141 / 134 and that kernel was deleted).  Per K-tile it stages the (TR+2) x (TC+2) halo patch of x
//   (60 pixels x 64 channels for 4 x 8) and the dy tile (TR*TC x 64) ONCE and issues 9 x TR*TC/2 MFMAs per wave:
//   2.25 x the MFMAs per barrier pair of the implicit-GEMM weight gradient (whose M-tile is one tap: every tap re-stages
//   its shifted copy of x), with 6 instead of 8 loads per thread and every LDS address = lane base + immediate.
//   The nine taps read the SAME patch, so x travels global -> LDS once per (pixel, co-tile) instead of nine times.
//   Split-K over sub-tile ranges into [split][9*Cin*Cout] partial sums, reduced in fixed order (deterministic);
//   the bias gradient is summed from the staged dy tiles by the ci-tile-0 workgroups.
// =====================================================================================================================
struct WgradPatchArgs {
    const float* x;      // [P, Cin]
    const float* dy;     // [P, Cout]
    float* ws;           // [splits][9*Cin*Cout]
    float* bias_ws;      // [splits][Cout] or null
    int B, H, W, Cin, Cout;
    int subs_x, subs_y;  // sub-tiles per image row / column
    long nsubs;          // B * subs_x * subs_y
    int tiles_co;        // Cout / 64
    int subs_per_split;
};

// K-tile = TR x TC pixels (4 x 8 for the 224 / 112 / 56-wide layers, 4 x 7 for the 28-wide, 2 x 14 for the 14-wide ones): lane half
// lh covers the TR/2 rows starting at row (TR/2) lh, TR*TC/2 MFMA k-steps per tile
template <int TR, int TC>
struct WgTile {
    static constexpr int PW = TC + 2, PP = (TR + 2) * PW;  // patch row pitch / pixels
    static constexpr int NP = TR * TC, KS = NP / 2;        // pixels / k-steps per K-tile
    static constexpr int XP = ((PP * 16 + 255) / 256) * 256 * 4;  // floats of the x patch area (every staging slot has a home)
    static constexpr int NVX = (PP * 16 + 255) / 256;
    static constexpr int LDS_BYTES = (XP + 32 * 64) * 4;
    static_assert(TR % 2 == 0 && NP <= 32 && NVX <= 4, "tile shape");
};

template <int TR, int TC>
__global__ __launch_bounds__(256, 2) void wgrad_patch_kernel(WgradPatchArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using G = WgTile<TR, TC>;
    constexpr int WG_XP = G::XP;
    float* Xp = smem;            // [(TR+2)(TC+2)][64]
    float* Dy = smem + WG_XP;    // [32][64] (rows >= TR*TC stay zero)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int cit = blockIdx.x / a.tiles_co, cot = blockIdx.x - cit * a.tiles_co;
    const int ci0 = cit * 64, co0 = cot * 64;
    const int W = a.W, Cin = a.Cin, Cout = a.Cout;
    const long P = (long)a.B * a.H * W;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)(P * Cin * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, (int)(P * Cout * 4), 0x00020000);

    // staging slots: x patch PP px x 16 channel quads (up to 4 float4 per thread, slots past the patch load zeros), dy 32 x 16 (2)
    int cx[G::NVX], fl[G::NVX];
#pragma unroll
    for (int u = 0; u < G::NVX; ++u) {
        const int f = tid + 256 * u, pix = f >> 4, q = f & 15;
        const int py = pix / G::PW, px = pix - py * G::PW;
        cx[u] = (((py - 1) * W + (px - 1)) * Cin + ci0 + q * 4) * 4;
        fl[u] = (py == 0 ? 1 : 0) | (py == TR + 1 ? 2 : 0) | (px == 0 ? 4 : 0) | (px == TC + 1 ? 8 : 0) | (pix >= G::PP ? 16 : 0);
    }
    unsigned cd[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int f = tid + 256 * u, pix = f >> 4, q = f & 15;
        cd[u] = pix < G::NP ? (unsigned)((((pix / TC) * W + (pix % TC)) * Cout + co0 + q * 4) * 4) : OOB;
    }
    const int xst = tid * 4;  // LDS float index of x slot 0 (slot u: + 1024 u), dy slot 0 (slot u: + 1024 u)

    // fragment bases (bytes): A = x patch, row i = ci, lane half lh covers pixels KS lh .. KS lh + KS - 1 (TR/2 rows) of the K-tile
    const int abase = (((TR / 2) * lh * G::PW) * 64 + (wave >> 1) * 32 + li) * 4;
    const int bbase = ((G::KS * lh) * 64 + (wave & 1) * 32 + li) * 4 + WG_XP * 4;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    long s = (long)blockIdx.y * a.subs_per_split;
    const long s_end = s + a.subs_per_split < a.nsubs ? s + a.subs_per_split : a.nsubs;
    const int subs_img = a.subs_x * a.subs_y;
    int b = (int)(s / subs_img), rem = (int)(s - (long)b * subs_img);
    int sy = rem / a.subs_x, sx = rem - sy * a.subs_x;

    // staging registers of TWO K-tiles: the loads run two tiles ahead of the MFMAs (the 64-channel layers stream x and dy from HBM
    // with no re-use between workgroups; one tile ahead = 9216 MFMA cycles did not always cover that latency)
    float4 pr[2][G::NVX], dr[2][2];
    auto issue = [&](int h) {
        const int pb = (b * a.H + sy * TR) * W + sx * TC;  // first pixel of the sub-tile
        const int em = 16 | (sy == 0 ? 1 : 0) | (sy == a.subs_y - 1 ? 2 : 0) | (sx == 0 ? 4 : 0) | (sx == a.subs_x - 1 ? 8 : 0);
        const int pbx = pb * Cin * 4;
#pragma unroll
        for (int u = 0; u < G::NVX; ++u) pr[h][u] = bufload(rx, (fl[u] & em) ? OOB : (unsigned)(cx[u] + pbx), 0);
        const unsigned pbd = (unsigned)pb * (unsigned)Cout * 4u;
#pragma unroll
        for (int u = 0; u < 2; ++u) dr[h][u] = bufload(rd, cd[u], pbd);
    };
    auto advance = [&]() {  // the ISSUE cursor (s, b, sy, sx)
        ++s;
        if (++sx == a.subs_x) { sx = 0; if (++sy == a.subs_y) { sy = 0; ++b; } }
    };
    // bias gradient = column sums of dy, taken from the staging REGISTERS (slot = pixel tid >> 4 (+16), columns 4 (tid & 15) ..+3):
    // eight VALU adds per K-tile in the shadow of the MFMAs, no LDS traffic; the 16 threads of a column quad are combined at the end
    const bool do_bias = a.bias_ws != nullptr && cit == 0;
    float4 csum = f4zero();

    long todo = s_end - s;  // K-tiles of this split
    if (s < s_end) { issue(0); advance(); }
    if (s < s_end) { issue(1); advance(); }
    while (todo > 0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {  // (register set h holds the tile that is consumed now; it is refilled with the tile two ahead)
            if (todo <= 0) break;
            --todo;
            __syncthreads();  // every wave is done with the previous tile
#pragma unroll
            for (int u = 0; u < G::NVX; ++u) *reinterpret_cast<float4*>(&Xp[xst + 1024 * u]) = pr[h][u];
#pragma unroll
            for (int u = 0; u < 2; ++u) *reinterpret_cast<float4*>(&Dy[xst + 1024 * u]) = dr[h][u];
            if (do_bias) {
                csum.x += dr[h][0].x + dr[h][1].x; csum.y += dr[h][0].y + dr[h][1].y; csum.z += dr[h][0].z + dr[h][1].z; csum.w += dr[h][0].w + dr[h][1].w;
            }
            __syncthreads();
            if (s < s_end) { issue(h); advance(); }  // the tile two ahead: its loads fly under this tile's and the next tile's MFMAs
            // KS k-steps: k-step ks covers pixel KS lh + ks of the K-tile = (row (TR/2) lh + ks / TC, column ks % TC)
            float fa[2][9], fb[2];
            auto frag = [&](int ks, int buf) {
                const char* sm = reinterpret_cast<const char*>(smem);
                fb[buf] = *reinterpret_cast<const float*>(sm + bbase + ks * 256);
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int dyy = t / 3 - 1, dxx = t % 3 - 1;
                    fa[buf][t] = *reinterpret_cast<const float*>(sm + abase + (((ks / TC + 1 + dyy) * G::PW + ks % TC + 1 + dxx) * 64) * 4);
                }
            };
            frag(0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks) {
                if (ks + 1 < G::KS) frag(ks + 1, (ks + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[ks & 1][t], fb[ks & 1], acc[t], 0, 0, 0);
            }
        }
    }

    // ---- partial sums of this split: ws[z][(tap*Cin + ci)*Cout + co]; C/D layout: col = lane & 31, row = (r&3) + 8 (r>>2) + 4 lh
    float* out = a.ws + (long)blockIdx.y * 9 * Cin * Cout;
    const int col = co0 + (wave & 1) * 32 + li;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = ci0 + (wave >> 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            out[((long)t * Cin + ci) * Cout + col] = acc[t][r];
        }
    if (do_bias) {
        __syncthreads();
        *reinterpret_cast<float4*>(&smem[tid * 4]) = csum;  // [pixel group tid >> 4][column 4 (tid & 15) + e]
        __syncthreads();
        if (tid < 64) {
            float t = 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) t += smem[g * 64 + tid];  // fixed order
            a.bias_ws[(long)blockIdx.y * Cout + co0 + tid] = t;
        }
    }
}

// dw = sum over the K splits of ws[split][MN] (fixed order; 16-byte accesses, MN % 4 == 0); the last block does the same for the
// bias gradient's partials (Cout floats per split) when db is given -- one launch per layer
__global__ __launch_bounds__(256) void wgrad_patch_reduce_kernel(const float* __restrict__ ws, int splits, long MN, float* __restrict__ dw,
                                                                 const float* __restrict__ bias_ws, int Cout, float* __restrict__ db,
                                                                 int accumulate) {
    if (db && blockIdx.x == gridDim.x - 1) {
        for (int i = threadIdx.x; i < Cout; i += 256) {
            float v = 0.f;
            for (int z = 0; z < splits; ++z) v += bias_ws[(long)z * Cout + i];
            db[i] = accumulate ? db[i] + v : v;
        }
        return;
    }
    const long Q = MN >> 2;
    const int nb = db ? gridDim.x - 1 : gridDim.x;
    const float4* w4 = reinterpret_cast<const float4*>(ws);
    float4* o4 = reinterpret_cast<float4*>(dw);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < Q; i += (long)nb * 256) {
        float4 v = w4[i];
        for (int z = 1; z < splits; ++z) {
            const float4 t = w4[(long)z * Q + i];
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        if (accumulate) {
            const float4 t = o4[i];
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        o4[i] = v;
    }
}

struct WgradPatchPlan {
    bool ok;
    int tr, tc;      // K-tile shape: 4 x 8, 4 x 7 (28-wide layers) or 2 x 14 (14-wide layers)
    int splits, sps, tiles;
    long nsubs;      // K-tiles
};

static WgradPatchPlan plan_wgrad_patch(int B, int H, int W, int Cin, int Cout) {
    WgradPatchPlan p;
    p.ok = false; p.splits = 0; p.sps = 0; p.tiles = 0; p.nsubs = 0; p.tr = 4; p.tc = 8;
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || Cin % 64 || Cout % 64) return p;
    const long P = (long)B * H * W;
    if (P * (long)(Cin > Cout ? Cin : Cout) * 4 > 0x7fffffffL) return p;
    if (W % 8 == 0 && H % 4 == 0) {
        p.nsubs = (long)B * (H / 4) * (W / 8);
    } else if (W % 7 == 0 && H % 4 == 0) {   // 28 x 28: 4 x 7 K-tiles (6 x 9 patch)
        p.tr = 4; p.tc = 7;
        p.nsubs = (long)B * (H / 4) * (W / 7);
    } else if (W % 14 == 0 && H % 2 == 0) {  // 14 x 14: 2 x 14 K-tiles (4 x 16 patch)
        p.tr = 2; p.tc = 14;
        p.nsubs = (long)B * (H / 2) * (W / 14);
    } else {
        return p;
    }
    p.tiles = (Cin / 64) * (Cout / 64);
    long slots = 512;                        // two workgroups per CU
    if (const char* e = getenv("VC_WGRAD_SLOTS")) slots = atol(e) > 0 ? atol(e) : 512;  // experiments only
    long splits = slots / p.tiles;
    if (splits > 256) splits = 256;          // (bounds the partial-sum traffic of the few-tile layers)
    if (splits > p.nsubs / 8) splits = p.nsubs / 8;  // >= 8 K-tiles per split
    if (splits < 1) splits = 1;
    p.sps = (int)((p.nsubs + splits - 1) / splits);
    p.splits = (int)((p.nsubs + p.sps - 1) / p.sps);
    p.ok = true;
    return p;
}

}  // namespace vc

using namespace vc;

extern "C" int vc_conv3x3_patch_supported(int B, int H, int W, int Cin, int Cout, int dgrad) {
    return plan_patch(B, H, W, dgrad ? Cout : Cin, dgrad ? Cin : Cout).scheme >= 0 ? 1 : 0;
}

extern "C" size_t vc_conv3x3_packed_workspace_bytes(int B, int H, int W, int Cin, int Cout, int dgrad) {
    return patch_workspace(plan_patch(B, H, W, dgrad ? Cout : Cin, dgrad ? Cin : Cout));
}

extern "C" int vc_conv3x3_pack_f32(void* stream, int Cin, int Cout, const float* w, int transpose, float* wp) {
    VC_CHECK_ARG(w && wp && Cin > 0 && Cout > 0 && Cin % 4 == 0 && Cout % 4 == 0, "channel counts must be multiples of 4");
    VC_CHECK_ARG((((uintptr_t)w | (uintptr_t)wp) & 15) == 0, "w / wp must be 16-byte aligned");
    const long total = 9L * Cin * Cout / 4;
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, Cin, Cout, transpose, (float4*)wp);
    VC_LAUNCH_CHECK();
    return 0;
}

extern "C" int vc_conv3x3_fwd_packed_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* x, const float* wp,
                                         const float* bias, float* y, int relu, float* ws, size_t ws_bytes) {
    VC_CHECK_ARG(x && wp && y, "null pointer");
    const PatchPlan p = plan_patch(B, H, W, Cin, Cout);
    if (p.scheme < 0) return fail(VC_EINVAL, "%s: shape not supported by the patch kernel (vc_conv3x3_patch_supported)", __func__);
    PatchArgs a;
    a.g = p.g; a.x = x; a.wp = wp; a.out = y; a.aux = bias; a.relu = relu; a.pool = nullptr; a.up_y = nullptr; a.up_dx = nullptr;
    return dispatch_patch<PK_FWD>((hipStream_t)stream, p, a, ws, ws_bytes);
}

extern "C" int vc_conv3x3_fwd_pool_packed_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* x, const float* wp,
                                              const float* bias, float* y, float* ypool, int relu, float* ws, size_t ws_bytes) {
    VC_CHECK_ARG(x && wp && y && ypool, "null pointer");
    const PatchPlan p = plan_patch(B, H, W, Cin, Cout);
    if (p.scheme != PATCH_SUB) return fail(VC_EINVAL, "%s: the fused max-pool needs the 4 x 8 sub-tile tiling (W %% 8 == 0, H %% 4 == 0, Cin %% 32 == 0, Cout %% 64 == 0)", __func__);
    PatchArgs a;
    a.g = p.g; a.x = x; a.wp = wp; a.out = y; a.aux = bias; a.relu = relu; a.pool = ypool; a.up_y = nullptr; a.up_dx = nullptr;
    return dispatch_patch<PK_FWD>((hipStream_t)stream, p, a, ws, ws_bytes);
}

extern "C" int vc_conv3x3_dgrad_packed_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* dy, const float* wpt,
                                           const float* relu_src, float* dx, float* ws, size_t ws_bytes) {
    VC_CHECK_ARG(dy && wpt && dx, "null pointer");
    const PatchPlan p = plan_patch(B, H, W, Cout, Cin);
    if (p.scheme < 0) return fail(VC_EINVAL, "%s: shape not supported by the patch kernel (vc_conv3x3_patch_supported)", __func__);
    PatchArgs a;
    a.g = p.g; a.x = dy; a.wp = wpt; a.out = dx; a.aux = relu_src; a.relu = 0; a.pool = nullptr; a.up_y = nullptr; a.up_dx = nullptr;
    return dispatch_patch<PK_DGRAD>((hipStream_t)stream, p, a, ws, ws_bytes);
}

extern "C" int vc_conv3x3_dgrad_unpool_packed_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* dy, const float* wpt,
                                                  const float* y_prepool, float* dx_prepool, float* ws, size_t ws_bytes) {
    VC_CHECK_ARG(dy && wpt && y_prepool && dx_prepool, "null pointer");
    VC_CHECK_ARG((((uintptr_t)y_prepool | (uintptr_t)dx_prepool) & 15) == 0, "y_prepool / dx_prepool must be 16-byte aligned");
    const PatchPlan p = plan_patch(B, H, W, Cout, Cin);
    if (p.scheme < 0) return fail(VC_EINVAL, "%s: shape not supported by the patch kernel (vc_conv3x3_patch_supported)", __func__);
    if ((long)B * 4 * H * W * Cin * 4 > 0x7fffffffL * 4L) return fail(VC_EINVAL, "%s: pre-pool tensor too large", __func__);
    PatchArgs a;
    a.g = p.g; a.x = dy; a.wp = wpt; a.out = nullptr; a.aux = nullptr; a.relu = 0; a.pool = nullptr; a.up_y = y_prepool; a.up_dx = dx_prepool;
    return dispatch_patch<PK_DGRAD>((hipStream_t)stream, p, a, ws, ws_bytes);
}

extern "C" int vc_conv3x3_wgrad_patch_supported(int B, int H, int W, int Cin, int Cout) {
    return plan_wgrad_patch(B, H, W, Cin, Cout).ok ? 1 : 0;
}

extern "C" size_t vc_conv3x3_wgrad_patch_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
    const WgradPatchPlan p = plan_wgrad_patch(B, H, W, Cin, Cout);
    return p.ok ? (size_t)p.splits * (9L * Cin * Cout + Cout) * sizeof(float) : 0;
}

extern "C" int vc_conv3x3_wgrad_patch_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* x, const float* dy,
                                          float* dw, float* db, int accumulate, float* ws, size_t ws_bytes) {
    VC_CHECK_ARG(x && dy && dw, "null pointer");
    const WgradPatchPlan p = plan_wgrad_patch(B, H, W, Cin, Cout);
    if (!p.ok) return fail(VC_EINVAL, "%s: shape not supported by the patch kernel (vc_conv3x3_wgrad_patch_supported)", __func__);
    const long MN = 9L * Cin * Cout;
    if (!ws || ws_bytes < (size_t)p.splits * (MN + Cout) * sizeof(float))
        return fail(VC_EWORKSPACE, "%s: workspace too small (need vc_conv3x3_wgrad_patch_workspace_bytes)", __func__);
    WgradPatchArgs a;
    a.x = x; a.dy = dy; a.ws = ws; a.bias_ws = db ? ws + (size_t)p.splits * MN : nullptr;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.subs_x = W / p.tc; a.subs_y = H / p.tr; a.nsubs = p.nsubs; a.tiles_co = Cout / 64; a.subs_per_split = p.sps;
    if (p.tc == 7)
        hipLaunchKernelGGL((wgrad_patch_kernel<4, 7>), dim3(p.tiles, p.splits), dim3(256), (WgTile<4, 7>::LDS_BYTES), (hipStream_t)stream, a);
    else if (p.tc == 14)
        hipLaunchKernelGGL((wgrad_patch_kernel<2, 14>), dim3(p.tiles, p.splits), dim3(256), (WgTile<2, 14>::LDS_BYTES), (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL((wgrad_patch_kernel<4, 8>), dim3(p.tiles, p.splits), dim3(256), (WgTile<4, 8>::LDS_BYTES), (hipStream_t)stream, a);
    VC_LAUNCH_CHECK();
    int grid = (int)((MN / 4 + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(wgrad_patch_reduce_kernel, dim3(grid + (db ? 1 : 0)), dim3(256), 0, (hipStream_t)stream, ws, p.splits, MN, dw,
                       a.bias_ws, Cout, db, accumulate);
    VC_LAUNCH_CHECK();
    return 0;
}
