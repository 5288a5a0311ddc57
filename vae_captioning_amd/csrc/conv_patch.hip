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
};

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
        if (SCHEME == PATCH_SUB) {
            const long s = (long)tm * 4 + (r >> 5);
            if (s >= g.nsubs) return -1;
            const int b = (int)(s / g.subs_img), rem = (int)(s - (long)b * g.subs_img);
            const int sy = rem / g.subs_x, sx = rem - sy * g.subs_x;
            const int y = sy * 4 + ((r & 31) >> 3), x = sx * 8 + (r & 7);
            return ((long)b * g.H + y) * g.W + x;
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
            const long s = (long)tm * 4 + j;
            if (s >= g.nsubs) return -1;
            const int b = (int)(s / g.subs_img), rem = (int)(s - (long)b * g.subs_img);
            const int sy = rem / g.subs_x, sx = rem - sy * g.subs_x;
            const int y = sy * 4 + py - 1, x = sx * 8 + px - 1;
            if ((unsigned)y >= (unsigned)g.H || (unsigned)x >= (unsigned)g.W) return -1;
            return ((long)b * g.H + y) * g.W + x;
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

template <int TN, int SCHEME, int KIND>
__global__ __launch_bounds__(256, 2) void conv_patch_kernel(PatchArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using CFG = PatchCfg<TN>;
    constexpr int NVP = PatchLds<SCHEME>::NVP;
    const PatchGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
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

    f32x16 acc[2][TN];
    acc_zero<CFG>(acc);

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
                        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c & 1][t][e], comp(b[u][c], e), acc[t][u], 0, 0, 0);
        }
    };
    const int PWp = (SCHEME == PATCH_SUB ? 10 : g.PW) * PITCH;  // floats per patch row (SUB: compile-time -> immediate offsets)
    auto tapoff = [&](int tap) { return (tap / 3 - 1) * PWp + (tap % 3 - 1) * PITCH; };

    if (cb < ce) {
        pload(cb);
        bload(b0, 0, cb);
        pstore();
        __syncthreads();
    }
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
        }
    }

    if (a.tail_ws) {  // raw partial sums of this split: [split][tail tile][128][BN]
        float* o = a.tail_ws + ((long)blockIdx.y * a.ntiles + (id - a.tile0)) * (128 * CFG::BN);
        epilogue_rows<CFG>(acc, smem, [&](int r, int cc, float4 v) { *reinterpret_cast<float4*>(o + r * CFG::BN + cc) = v; });
        return;
    }
    // epilogue: each call handles 4 consecutive columns of one tile row
    epilogue_rows<CFG>(acc, smem, [&](int r, int cc, float4 v) {
        const long p = map.out_pixel(r);
        if (p < 0) return;
        const int col = n0 + cc;
        if (KIND == PK_FWD) {
            if (a.aux) {
                const float4 bv = *reinterpret_cast<const float4*>(a.aux + col);
                v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
            }
            if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        } else if (a.aux) {  // ReluGrad of the layer that produced this convolution's input
            const float4 m = *reinterpret_cast<const float4*>(a.aux + p * N + col);
            if (!(m.x > 0.f)) v.x = 0.f;
            if (!(m.y > 0.f)) v.y = 0.f;
            if (!(m.z > 0.f)) v.z = 0.f;
            if (!(m.w > 0.f)) v.w = 0.f;
        }
        *reinterpret_cast<float4*>(a.out + p * N + col) = v;
    });
}

// Sum of the tail launch's K splits (fixed order) + the epilogue of the main launch.
template <int SCHEME, int KIND>
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
        *reinterpret_cast<float4*>(a.out + p * g.N + col) = v;
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
    p.TN = N % 128 == 0 ? 2 : 1;
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
constexpr int PATCH_CUS = 256;

struct PatchTail {
    int main_tiles, tail_tiles, splits, cps;
};

static PatchTail plan_patch_tail(const PatchPlan& p) {
    PatchTail t;
    const int T = p.tiles_m * p.tiles_n;
    const int nch = p.g.C / 32;
    const int tail = T % PATCH_CUS;
    t.main_tiles = T; t.tail_tiles = 0; t.splits = 1; t.cps = nch;
    if (tail == 0 || nch < 2) return t;
    // cost(cps) = rounds of short workgroups x (chunks each + prologue / epilogue) + partial-sum traffic + one more launch
    double best = nch + 0.2;  // unsplit: one tile on `tail` CUs
    int best_cps = nch;
    for (int cps = 1; cps < nch; ++cps) {
        const int splits = cdiv(nch, cps);
        const double cost = cdiv((long)tail * splits, PATCH_CUS) * (cps + 0.2) + 0.002 * splits * tail + 0.6;
        if (cost < best - 1e-9) { best = cost; best_cps = cps; }
    }
    if (best_cps == nch) return t;
    t.cps = best_cps;
    t.splits = cdiv(nch, best_cps);
    t.main_tiles = T - tail;
    t.tail_tiles = tail;
    return t;
}

static size_t patch_workspace(const PatchPlan& p) {
    if (p.scheme < 0) return 0;
    const PatchTail t = plan_patch_tail(p);
    return t.tail_tiles ? (size_t)t.splits * t.tail_tiles * 128 * (p.TN * 64) * sizeof(float) : 0;
}

template <int TN, int SCHEME, int KIND>
static int launch_patch(hipStream_t st, const PatchPlan& p, PatchArgs& a, float* ws, size_t ws_bytes) {
    const PatchTail t = plan_patch_tail(p);
    constexpr int smem = PatchLds<SCHEME>::template bytes<TN>();
    a.tiles_n = p.tiles_n;
    a.nchunks = p.g.C / 32;
    a.tile0 = 0; a.tail_ws = nullptr; a.chunks_per_split = 0;
    const size_t need = (size_t)t.splits * t.tail_tiles * 128 * (TN * 64) * sizeof(float);
    if (t.tail_tiles == 0 || !ws || ws_bytes < need) {
        a.ntiles = p.tiles_m * p.tiles_n;
        hipLaunchKernelGGL((conv_patch_kernel<TN, SCHEME, KIND>), dim3(a.ntiles, 1), dim3(256), smem, st, a);
        return launch_status("conv patch");
    }
    if (t.main_tiles > 0) {
        a.ntiles = t.main_tiles;
        hipLaunchKernelGGL((conv_patch_kernel<TN, SCHEME, KIND>), dim3(a.ntiles, 1), dim3(256), smem, st, a);
        if (int e = launch_status("conv patch")) return e;
    }
    PatchArgs d = a;
    d.tile0 = t.main_tiles; d.ntiles = t.tail_tiles; d.chunks_per_split = t.cps; d.tail_ws = ws;
    hipLaunchKernelGGL((conv_patch_kernel<TN, SCHEME, KIND>), dim3(d.ntiles, t.splits), dim3(256), smem, st, d);
    if (int e = launch_status("conv patch tail")) return e;
    const long items = (long)t.tail_tiles * 128 * (TN * 16);
    int grid = (int)((items + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL((patch_tail_reduce_kernel<SCHEME, KIND>), dim3(grid), dim3(256), 0, st, d, t.splits, TN * 64);
    return launch_status("conv patch tail reduce");
}

template <int KIND>
static int dispatch_patch(hipStream_t st, const PatchPlan& p, PatchArgs& a, float* ws, size_t ws_bytes) {
    if (p.scheme == PATCH_SUB)
        return p.TN == 2 ? launch_patch<2, PATCH_SUB, KIND>(st, p, a, ws, ws_bytes) : launch_patch<1, PATCH_SUB, KIND>(st, p, a, ws, ws_bytes);
    return p.TN == 2 ? launch_patch<2, PATCH_FLAT, KIND>(st, p, a, ws, ws_bytes) : launch_patch<1, PATCH_FLAT, KIND>(st, p, a, ws, ws_bytes);
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
    a.g = p.g; a.x = x; a.wp = wp; a.out = y; a.aux = bias; a.relu = relu;
    return dispatch_patch<PK_FWD>((hipStream_t)stream, p, a, ws, ws_bytes);
}

extern "C" int vc_conv3x3_dgrad_packed_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* dy, const float* wpt,
                                           const float* relu_src, float* dx, float* ws, size_t ws_bytes) {
    VC_CHECK_ARG(dy && wpt && dx, "null pointer");
    const PatchPlan p = plan_patch(B, H, W, Cout, Cin);
    if (p.scheme < 0) return fail(VC_EINVAL, "%s: shape not supported by the patch kernel (vc_conv3x3_patch_supported)", __func__);
    PatchArgs a;
    a.g = p.g; a.x = dy; a.wp = wpt; a.out = dx; a.aux = relu_src; a.relu = 0;
    return dispatch_patch<PK_DGRAD>((hipStream_t)stream, p, a, ws, ws_bytes);
}
