// VGG16 feature-extractor kernels (utils/image_embeddings.py:26-238) for gfx950:
// 3x3 / stride 1 / SAME convolution as an implicit GEMM on the fp32 MFMA tile engine
// (forward, data gradient, weight gradient), 2x2/2 max-pool forward/backward, input
// preprocessing.  Layout: NHWC activations, HWIO kernels (the reference's TF layout).
//
//   forward : out[p, co]  = sum_{tap, ci} in[p + off(tap), ci] * W[tap, ci, co]      M = B*H*W, N = Cout, K = 9*Cin
//   dgrad   : din[p, ci]  = sum_{tap, co} dout[p - off(tap), co] * W[tap, ci, co]    M = B*H*W, N = Cin,  K = 9*Cout
//   wgrad   : dW[tap,ci,co] = sum_p in[p + off(tap), ci] * dout[p, co]               M = 9*Cin, N = Cout, K = B*H*W
// The A operand is never materialised (no im2col): the tile loader gathers the shifted
// NHWC rows directly (zero outside the image) into the LDS image, where the 9 taps of a
// pixel tile are re-read from L1/L2.  Channel counts must be multiples of 4 (conv1_1's 3
// input channels are zero-padded to 4 by the caller; vc_vgg_preprocess_f32 emits NHWC4).
#include <type_traits>
#include "gemm_core.h"
#include "vaecap.h"

namespace vc {

struct ConvGeom {
    int B, H, W, Cin, Cout;
    int lc_in, lc_out;  // log2 of Cin / Cout (powers of two)
    int HW;
    long P;  // B*H*W
};

// ---- gathering loaders (slot protocol of gemm_core.h) ------------------------------------
// forward / dgrad A operand: MK image, rows = pixels, k = tap*C + c over a [P, C] NHWC tensor,
// shifted by sign * off(tap).  The pixel decode (two divisions) happens once per slot.
struct LoadPixelsMK {
    const float* x;
    ConvGeom g;
    int lc;    // log2(C) of the gathered tensor
    int sign;  // +1 forward, -1 dgrad
    const float* pb[MAXNV];
    int py[MAXNV], px[MAXNV], ko[MAXNV];
    __device__ __forceinline__ void init(int u, int row, int kofs) {
        ko[u] = kofs;
        if (row >= g.P) { pb[u] = nullptr; py[u] = px[u] = 0; return; }
        const int b = row / g.HW, rem = row - b * g.HW;
        py[u] = rem / g.W;
        px[u] = rem - py[u] * g.W;
        pb[u] = x + ((long)row << lc);
    }
    __device__ __forceinline__ float4 load(int u, int k0) const {
        const int k = k0 + ko[u];
        const int tap = k >> lc;
        if (!pb[u] || tap >= 9) return f4zero();
        const int c = k & ((1 << lc) - 1);
        const int dy = (tap / 3 - 1) * sign, dx = (tap % 3 - 1) * sign;
        if ((unsigned)(py[u] + dy) >= (unsigned)g.H || (unsigned)(px[u] + dx) >= (unsigned)g.W) return f4zero();
        return *reinterpret_cast<const float4*>(pb[u] + (long)((dy * g.W + dx) << lc) + c);
    }
};
// wgrad A operand: KM image, rows m = tap*Cin + ci (4 consecutive ci), k = pixel.  The tap of a slot is
// fixed; the pixel -> (y, x) decode per K-tile uses a float reciprocal with a +-1 correction (exact
// for pixel indices < 2^24; larger tensors take the integer-division path).
__device__ __forceinline__ int fastdiv(int n, int d, float inv) {
    int q = (int)((float)n * inv);
    const int r = n - q * d;
    q += (r >= d) - (r < 0);
    return q;
}
struct LoadPixelsKM {
    const float* x;
    ConvGeom g;
    float invW, invH;
    const float* pb[MAXNV];   // x + shift(tap)*Cin + ci  (add pixel*Cin)
    int dy[MAXNV], dx[MAXNV], ko[MAXNV];
    __device__ __forceinline__ void init(int u, int m, int kofs) {
        ko[u] = kofs;
        const int tap = m >> g.lc_in;
        if (tap >= 9) { pb[u] = nullptr; dy[u] = dx[u] = 0; return; }
        dy[u] = tap / 3 - 1;
        dx[u] = tap % 3 - 1;
        pb[u] = x + (long)((dy[u] * g.W + dx[u]) << g.lc_in) + (m & (g.Cin - 1));
    }
    __device__ __forceinline__ float4 load(int u, int k0) const {
        const int p = k0 + ko[u];
        if (!pb[u] || p >= g.P) return f4zero();
        int rowi, yy;
        if (g.P < (1 << 24)) {
            rowi = fastdiv(p, g.W, invW);
            yy = rowi - fastdiv(rowi, g.H, invH) * g.H;
        } else {
            rowi = p / g.W;
            yy = rowi % g.H;
        }
        const int xx = p - rowi * g.W;
        if ((unsigned)(yy + dy[u]) >= (unsigned)g.H || (unsigned)(xx + dx[u]) >= (unsigned)g.W) return f4zero();
        return *reinterpret_cast<const float4*>(pb[u] + ((long)p << g.lc_in));
    }
};
// dgrad B operand: MK image, rows n = ci, k = tap*Cout + co  ->  W[tap][ci][co]
struct LoadWeightsT {
    const float* w;
    ConvGeom g;
    const float* q[MAXNV];
    int ko[MAXNV];
    __device__ __forceinline__ void init(int u, int n, int kofs) {
        q[u] = n < g.Cin ? w + ((long)n << g.lc_out) : nullptr;
        ko[u] = kofs;
    }
    __device__ __forceinline__ float4 load(int u, int k0) const {
        const int k = k0 + ko[u];
        const int tap = k >> g.lc_out;
        if (!q[u] || tap >= 9) return f4zero();
        return *reinterpret_cast<const float4*>(q[u] + (((long)tap * g.Cin) << g.lc_out) + (k & (g.Cout - 1)));
    }
};

enum { CONV_FWD = 0, CONV_DGRAD = 1, CONV_WGRAD = 2 };

struct ConvArgs {
    ConvGeom g;
    const float* a;      // fwd: x, dgrad: dy, wgrad: x
    const float* b;      // fwd: w, dgrad: w,  wgrad: dy
    float* out;          // fwd: y, dgrad: dx, wgrad: split-K workspace
    const float* aux;    // fwd: bias, dgrad: relu source (x; may be null)
    int relu;
    int tiles_n, ntiles, kchunk;
    float* bias_ws;      // wgrad: [splits, Cout] partial column sums of dy (bias gradient), or null
    // fwd / dgrad "tail" launch (the last partial round of tiles, K split so that it fills the chip once):
    int tile0;           // first tile of this launch (tiles before it belong to the main launch)
    float* tail_ws;      // null: normal epilogue; else raw partial sums [split][tail_rows][N]
    long tail_row0;      // first output row of the tail
    long tail_rows;
};

// Column sums of the dy tile staged in LDS (KM image Bs[k][BN+4]); thread t owns column t % BN and the
// k rows [part*PER, part*PER + PER), part = t / BN.
constexpr int colsum_parts(int groups) { return groups >= 32 ? 32 : groups >= 16 ? 16 : groups >= 8 ? 8 : groups >= 4 ? 4 : groups >= 2 ? 2 : 1; }

template <class CFG>
struct ColsumHook {
    float* acc;
    bool on;
    __device__ __forceinline__ void operator()(const float*, const float* Bs) const {
        if (!on) return;
        constexpr int PARTS = colsum_parts(CFG::NT / CFG::BN), PER = 32 / PARTS;
        const int col = threadIdx.x % CFG::BN, part = threadIdx.x / CFG::BN;
        if (part >= PARTS) return;  // thread counts that are 3 x BN: the third group idles
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < PER; ++k) s += Bs[(part * PER + k) * (CFG::BN + 4) + col];
        *acc += s;
    }
};

static inline int grid_for(long work_items, int per_block = 256, int cap = 4096) {
    long b = (work_items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (int)b;
}

// Tail launch: the split's raw partial sums of this tile -> tail_ws[split][row - tail_row0][col]
template <class CFG>
__device__ __forceinline__ void tail_store(f32x16 (&acc)[CFG::TM][CFG::TN], float* smem, const ConvArgs& c, int m0, int n0, int ncols) {
    float* out = c.tail_ws + (long)blockIdx.y * c.tail_rows * ncols;
    epilogue_rows<CFG>(acc, smem, [&](int r, int cc, float4 v) {
        const long row = m0 + r;
        const int col = n0 + cc;
        if (row >= c.g.P || col >= ncols) return;
        *reinterpret_cast<float4*>(out + (row - c.tail_row0) * ncols + col) = v;
    });
}

// Sum of the tail's K splits (fixed order) + the epilogue the main launch applies in registers.
template <int KIND>
__global__ __launch_bounds__(256) void conv_tail_reduce_kernel(const float4* __restrict__ ws, int splits, long rows, int nc4, long row0,
                                                               const float* __restrict__ aux, int relu, float4* __restrict__ out) {
    const long total = rows * nc4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        float4 v = ws[i];
        for (int z = 1; z < splits; ++z) {
            const float4 t = ws[(long)z * total + i];
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        const long o = row0 * nc4 + i;
        if (KIND == CONV_FWD) {
            if (aux) {
                const float4 bv = reinterpret_cast<const float4*>(aux)[i % nc4];
                v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
            }
            if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        } else if (aux) {
            const float4 m = reinterpret_cast<const float4*>(aux)[o];
            if (!(m.x > 0.f)) v.x = 0.f;
            if (!(m.y > 0.f)) v.y = 0.f;
            if (!(m.z > 0.f)) v.z = 0.f;
            if (!(m.w > 0.f)) v.w = 0.f;
        }
        out[o] = v;
    }
}

template <class CFG, int KIND>
__global__ __launch_bounds__(CFG::NT) void conv_kernel(ConvArgs c) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const ConvGeom& g = c.g;
    const int id = c.tile0 + xcd_remap(blockIdx.x, c.ntiles);
    const int m0 = (id / c.tiles_n) * CFG::BM;
    const int n0 = (id % c.tiles_n) * CFG::BN;
    f32x16 acc[CFG::TM][CFG::TN];
    acc_zero<CFG>(acc);
    if (KIND == CONV_FWD) {
        LoadPixelsMK la;
        la.x = c.a; la.g = g; la.lc = g.lc_in; la.sign = 1;
        LoadKM<true> lb;
        lb.p = c.b; lb.ld = g.Cout; lb.R = g.Cout; lb.K = 9 * g.Cin;
        const int Kf = 9 * g.Cin;
        const int kb = c.tail_ws ? blockIdx.y * c.kchunk : 0;
        const int ke = c.tail_ws ? (kb + c.kchunk < Kf ? kb + c.kchunk : Kf) : Kf;
        mfma_mainloop<CFG, MODE_MK, MODE_KM>(acc, la, lb, m0, n0, kb, ke, smem);
        if (c.tail_ws) {
            tail_store<CFG>(acc, smem, c, m0, n0, g.Cout);
            return;
        }
        epilogue_rows<CFG>(acc, smem, [&](int r, int cc, float4 v) {
            const long row = m0 + r;
            const int col = n0 + cc;
            if (row >= g.P || col >= g.Cout) return;  // Cout % 4 == 0: whole quads
            if (c.aux) {
                const float4 bv = *reinterpret_cast<const float4*>(c.aux + col);
                v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
            }
            if (c.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            *reinterpret_cast<float4*>(c.out + row * g.Cout + col) = v;
        });
    } else if (KIND == CONV_DGRAD) {
        LoadPixelsMK la;
        la.x = c.a; la.g = g; la.lc = g.lc_out; la.sign = -1;
        LoadWeightsT lb;
        lb.w = c.b; lb.g = g;
        const int Kf = 9 * g.Cout;
        const int kb = c.tail_ws ? blockIdx.y * c.kchunk : 0;
        const int ke = c.tail_ws ? (kb + c.kchunk < Kf ? kb + c.kchunk : Kf) : Kf;
        mfma_mainloop<CFG, MODE_MK, MODE_MK>(acc, la, lb, m0, n0, kb, ke, smem);
        if (c.tail_ws) {
            tail_store<CFG>(acc, smem, c, m0, n0, g.Cin);
            return;
        }
        epilogue_rows<CFG>(acc, smem, [&](int r, int cc, float4 v) {
            const long row = m0 + r;
            const int col = n0 + cc;
            if (row >= g.P || col >= g.Cin) return;
            if (c.aux) {  // ReluGrad of the layer that produced this convolution's input
                const float4 m = *reinterpret_cast<const float4*>(c.aux + row * g.Cin + col);
                if (!(m.x > 0.f)) v.x = 0.f;
                if (!(m.y > 0.f)) v.y = 0.f;
                if (!(m.z > 0.f)) v.z = 0.f;
                if (!(m.w > 0.f)) v.w = 0.f;
            }
            *reinterpret_cast<float4*>(c.out + row * g.Cin + col) = v;
        });
    } else {
        const int M = 9 * g.Cin;
        const long kb = (long)blockIdx.y * c.kchunk;
        const long ke = kb + c.kchunk < g.P ? kb + c.kchunk : g.P;
        LoadPixelsKM la;
        la.x = c.a; la.g = g; la.invW = 1.0f / (float)g.W; la.invH = 1.0f / (float)g.H;
        LoadKM<true> lb;
        lb.p = c.b; lb.ld = g.Cout; lb.R = g.Cout; lb.K = (int)g.P;
        float csum = 0.f;
        const bool do_bias = c.bias_ws != nullptr && m0 == 0;  // one m-tile per (n-tile, split) sums dy's columns
        ColsumHook<CFG> hook{&csum, do_bias};
        mfma_mainloop<CFG, MODE_KM, MODE_KM, LoadPixelsKM, LoadKM<true>, 0, ColsumHook<CFG>>(acc, la, lb, m0, n0, (int)kb, (int)ke, smem, hook);
        if (do_bias) {  // combine the NT/BN partial sums of each column (fixed order) and store the split's partial
            constexpr int PARTS = colsum_parts(CFG::NT / CFG::BN);
            __syncthreads();
            smem[threadIdx.x] = csum;
            __syncthreads();
            if (threadIdx.x < CFG::BN && n0 + (int)threadIdx.x < g.Cout) {
                float t = 0.f;
#pragma unroll
                for (int q = 0; q < PARTS; ++q) t += smem[q * CFG::BN + threadIdx.x];
                c.bias_ws[(long)blockIdx.y * g.Cout + n0 + threadIdx.x] = t;
            }
        }
        float* out = c.out + (long)blockIdx.y * M * g.Cout;
        epilogue_rows<CFG>(acc, smem, [&](int r, int cc, float4 v) {
            const int row = m0 + r, col = n0 + cc;
            if (row >= M || col >= g.Cout) return;
            *reinterpret_cast<float4*>(out + (long)row * g.Cout + col) = v;
        });
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, int splits, long MN,
                                                           float* __restrict__ dw, int accumulate) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < MN; i += (long)gridDim.x * 256) {
        float v = 0.f;
        for (int z = 0; z < splits; ++z) v += ws[(long)z * MN + i];
        dw[i] = accumulate ? dw[i] + v : v;
    }
}

using ConvCfgWide = TileCfg<2, 2, 2, 2>;    // 128 x 128
using ConvCfgNarrow = TileCfg<4, 1, 2, 2>;  // 256 x 64, 256 threads (weight gradients with <= 64 output channels)
using ConvCfgN2 = TileCfg<2, 2, 2, 1>;      // 128 x 64, waves 64 x 32: forward / data gradient of the 64-channel layers (116-120 VGPRs,
                                            // 4 workgroups per CU; the 256 x 64 tile needed 184-196 VGPRs = 2 per CU: conv1_2 108 -> 115 TFLOP/s)

static int ilog2_exact(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return (1 << l) == v ? l : -1;
}

static int make_geom(ConvGeom& g, int B, int H, int W, int Cin, int Cout) {
    g.B = B; g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout;
    g.lc_in = ilog2_exact(Cin);
    g.lc_out = ilog2_exact(Cout);
    g.HW = H * W;
    g.P = (long)B * H * W;
    if (g.lc_in < 2 || g.lc_out < 2) return 1;            // powers of two >= 4
    if (g.P * (long)(Cin > Cout ? Cin : Cout) > 0x7fffffffffffL) return 1;
    if (g.P > 0x7fffffff - 512) return 1;                  // pixel index is an int in the tile engine
    return 0;
}

struct WgradPlan {
    int splits, kchunk;
};
// 9*64 = 576 rows = 3 x 192 exactly: the 64-input-channel layers (conv1_2, conv2_1) get 192-row tiles
// (256- / 128-row tiles would compute 768 / 640 rows: 25 % / 10 % of the MFMAs wasted)
static int wgrad_bm(int Cin, int Cout) { return 9 * Cin <= 64 ? 64 : (Cin == 64 ? 192 : (Cout <= 64 ? 256 : 128)); }
static int wgrad_bn(int Cin, int Cout) { return Cout <= 64 ? 64 : 128; }

static WgradPlan plan_wgrad(const ConvGeom& g) {
    const long tiles = (long)cdiv(9 * g.Cin, wgrad_bm(g.Cin, g.Cout)) * cdiv(g.Cout, wgrad_bn(g.Cin, g.Cout));
    // Workgroups per launch = a whole number of rounds of resident workgroups, never a round and a bit (the ragged
    // second round ran at a third of the occupancy: 1024-ish workgroups on 768 slots cost conv2_2 / conv3_x 10-17 %).
    // 128 x 128 tiles run 3 per CU (768 slots): two rounds; Cin == 64 (192-row tiles): one round - the 192 x 64 kernel
    // (152 VGPRs) runs 3 per CU = 768, the 192 x 128 kernel (212 VGPRs) 2 per CU = 512.
    // (two rounds for the small outputs; one for the 512-channel layers, whose 9.4 MB partial sums make the reduce
    // launch cost as much as the second round gains)
    const long big_out = 9L * g.Cin * g.Cout >= (1L << 21);
    long splits = (g.Cin == 64 ? (g.Cout <= 64 ? 768 : 512) : (big_out ? 768 : 1536)) / tiles;
    const long maxs = g.P / 512 > 0 ? g.P / 512 : 1;  // >= 16 K-tiles per split
    if (splits > maxs) splits = maxs;
    if (splits > 256) splits = 256;
    if (splits < 1) splits = 1;
    WgradPlan p;
    p.kchunk = cdiv(cdiv(g.P, splits), 32) * 32;
    p.splits = cdiv(g.P, p.kchunk);
    return p;
}

using ConvCfgSmall = TileCfg<2, 2, 1, 1>;    //  64 x  64: conv1_1's weight gradient (M = 9*4 rows)
using ConvCfgW192n = TileCfg<2, 2, 3, 1>;    // 192 x  64, waves 96 x 32: conv1_2's weight gradient (576 x 64)
using ConvCfgW192w = TileCfg<2, 2, 3, 2>;    // 192 x 128, waves 96 x 64: conv2_1's weight gradient (576 x 128)

template <class C, int KIND>
static void launch_cfg(hipStream_t st, ConvArgs& c, int Mrows, int Ncols, int splits) {
    c.tiles_n = cdiv(Ncols, C::BN);
    c.ntiles = cdiv(Mrows, C::BM) * c.tiles_n;
    c.tile0 = 0; c.tail_ws = nullptr; c.tail_row0 = 0; c.tail_rows = 0;
    hipLaunchKernelGGL((conv_kernel<C, KIND>), dim3(c.ntiles, splits), dim3(C::NT), C::SMEM_BYTES, st, c);
}

// Forward / data-gradient launches are cut into whole rounds of resident workgroups: a launch of 2.04 rounds costs 2.33
// (measured on conv4_2 at 64 images: 1520 tiles 1.83 ms, 1544 tiles 2.13 ms -- the ragged last round runs one
// workgroup per CU at a third of the chip's rate).  The main launch takes floor(rounds) x slots tiles (whole tile rows);
// the remaining tile rows run as a second launch with K split so that it is again one full round of (short)
// workgroups, and conv_tail_reduce_kernel sums the splits in fixed order and applies the epilogue.
struct TailPlan {
    int main_tiles, tail_tiles, splits, kchunk;
    long row0, rows;
};

template <class C>
static TailPlan plan_tail(long P, int Ncols, int K, int slots) {
    TailPlan t;
    const int tiles_n = cdiv(Ncols, C::BN);
    const int tiles_m = (int)cdiv(P, (long)C::BM);
    const int T = tiles_m * tiles_n;
    t.main_tiles = T; t.tail_tiles = 0; t.splits = 1; t.kchunk = 0; t.row0 = P; t.rows = 0;
    const int full = (T / slots) * slots;
    if (full == T) return t;
    const int main_m = full / tiles_n;  // whole tile rows (0: less than one round of tiles -> all of them are split)
    const int tail = T - main_m * tiles_n;
    // whole launch below one round (conv5): ~1200 short workgroups balance the CUs better than 392 long ones (gemm.hip
    // plan_gemm has the measurements); a tail after full rounds fills exactly one more round
    int S = main_m == 0 ? (1200 + tail / 2) / tail : slots / tail;
    const int ktiles = cdiv(K, 32);
    if (S > ktiles / (main_m == 0 ? 16 : 4)) S = ktiles / (main_m == 0 ? 16 : 4);  // >= 4 K-tiles per split of a tail, >= 16 when everything is split
    if (S > 32) S = 32;
    if (S < 2) return t;
    t.kchunk = cdiv(ktiles, S) * 32;
    t.splits = cdiv(K, t.kchunk);
    t.main_tiles = main_m * tiles_n;
    t.tail_tiles = tail;
    t.row0 = (long)main_m * C::BM;
    t.rows = P - t.row0;
    return t;
}

template <class C, int KIND>
static int launch_rounds(hipStream_t st, ConvArgs& c, long P, int Ncols, int K, int slots, float* ws, size_t ws_bytes) {
    const TailPlan t = plan_tail<C>(P, Ncols, K, slots);
    c.tiles_n = cdiv(Ncols, C::BN);
    c.tile0 = 0; c.tail_ws = nullptr; c.tail_row0 = 0; c.tail_rows = 0; c.kchunk = 0;
    const size_t need = (size_t)t.splits * t.rows * Ncols * sizeof(float);
    if (t.tail_tiles == 0 || !ws || ws_bytes < need) {  // no ragged round, or no workspace given: one launch
        c.ntiles = (int)cdiv(P, (long)C::BM) * c.tiles_n;
        hipLaunchKernelGGL((conv_kernel<C, KIND>), dim3(c.ntiles, 1), dim3(C::NT), C::SMEM_BYTES, st, c);
        return launch_status("conv");
    }
    if (t.main_tiles > 0) {
        c.ntiles = t.main_tiles;
        hipLaunchKernelGGL((conv_kernel<C, KIND>), dim3(c.ntiles, 1), dim3(C::NT), C::SMEM_BYTES, st, c);
        if (int e = launch_status("conv")) return e;
    }
    ConvArgs d = c;
    d.tile0 = t.main_tiles; d.ntiles = t.tail_tiles; d.kchunk = t.kchunk; d.tail_ws = ws; d.tail_row0 = t.row0; d.tail_rows = t.rows;
    hipLaunchKernelGGL((conv_kernel<C, KIND>), dim3(d.ntiles, t.splits), dim3(C::NT), C::SMEM_BYTES, st, d);
    if (int e = launch_status("conv tail")) return e;
    const int nc4 = Ncols / 4;
    hipLaunchKernelGGL((conv_tail_reduce_kernel<KIND>), dim3(grid_for(t.rows * nc4)), dim3(256), 0, st, (const float4*)ws, t.splits, t.rows,
                       nc4, t.row0, c.aux, c.relu, (float4*)c.out);
    return launch_status("conv tail reduce");
}

// workgroups per round: the 128 x 128 kernels run 3 per CU (164-168 VGPRs)
constexpr int SLOTS_WIDE = 768, SLOTS_N2 = 768;  // (N2: 768 measured best of 768 / 1024 / 1280)

template <int KIND>
static size_t rounds_workspace(long P, int Ncols, int K) {
    const TailPlan t = Ncols <= 64 ? plan_tail<ConvCfgN2>(P, Ncols, K, SLOTS_N2) : plan_tail<ConvCfgWide>(P, Ncols, K, SLOTS_WIDE);
    return t.tail_tiles ? (size_t)t.splits * t.rows * Ncols * sizeof(float) : 0;
}

template <int KIND>
static int launch_fwd_dgrad(hipStream_t st, ConvArgs& c, long P, int Ncols, int K, float* ws, size_t ws_bytes) {
    if (Ncols <= 64) return launch_rounds<ConvCfgN2, KIND>(st, c, P, Ncols, K, SLOTS_N2, ws, ws_bytes);
    return launch_rounds<ConvCfgWide, KIND>(st, c, P, Ncols, K, SLOTS_WIDE, ws, ws_bytes);
}

static void launch_wgrad(hipStream_t st, ConvArgs& c, int Mrows, int Ncols, int splits) {
    constexpr int KIND = CONV_WGRAD;
    if (Mrows <= 64) {
        launch_cfg<ConvCfgSmall, KIND>(st, c, Mrows, Ncols, splits);
    } else if (Mrows == 576) {
        if (Ncols <= 64) launch_cfg<ConvCfgW192n, KIND>(st, c, Mrows, Ncols, splits);
        else launch_cfg<ConvCfgW192w, KIND>(st, c, Mrows, Ncols, splits);
    } else if (Ncols <= 64) {
        launch_cfg<ConvCfgNarrow, KIND>(st, c, Mrows, Ncols, splits);
    } else {  // (halving the tile for the few-tile conv5 layers was measured slower: 0.70 vs 0.62 ms)
        launch_cfg<ConvCfgWide, KIND>(st, c, Mrows, Ncols, splits);
    }
}

// ---- max-pool 2x2 / stride 2 (utils/image_embeddings.py:59-63 ...) ------------------------
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float4* __restrict__ x, int B, int H, int W, int C4,
                                                          float4* __restrict__ y) {
    const int Ho = H >> 1, Wo = W >> 1;
    const long total = (long)B * Ho * Wo * C4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C4);
        long p = i / C4;
        const int xo = (int)(p % Wo);
        p /= Wo;
        const int yo = (int)(p % Ho);
        const long b = p / Ho;
        const long base = ((b * H + 2 * yo) * W + 2 * xo) * C4 + c;
        const float4 a = x[base], b1 = x[base + C4], c1 = x[base + (long)W * C4], d = x[base + (long)W * C4 + C4];
        float4 m;
        m.x = fmaxf(fmaxf(a.x, b1.x), fmaxf(c1.x, d.x));
        m.y = fmaxf(fmaxf(a.y, b1.y), fmaxf(c1.y, d.y));
        m.z = fmaxf(fmaxf(a.z, b1.z), fmaxf(c1.z, d.z));
        m.w = fmaxf(fmaxf(a.w, b1.w), fmaxf(c1.w, d.w));
        y[i] = m;
    }
}

// MaxPoolGrad (first maximum in (dy,dx) scan order wins; TF-sem.) fused with the ReluGrad of
// the convolution that produced x:  dx = (x is the window's first max && x > 0) ? dy : 0
__device__ __forceinline__ void route1(float a, float b, float c, float d, float g, int relu, float& oa, float& ob, float& oc,
                                       float& od) {
    const float m = fmaxf(fmaxf(a, b), fmaxf(c, d));
    const int w = (a == m) ? 0 : (b == m) ? 1 : (c == m) ? 2 : 3;
    const float v = (relu && !(m > 0.f)) ? 0.f : g;
    oa = w == 0 ? v : 0.f; ob = w == 1 ? v : 0.f; oc = w == 2 ? v : 0.f; od = w == 3 ? v : 0.f;
}
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float4* __restrict__ x, const float4* __restrict__ dy, int B,
                                                          int H, int W, int C4, int relu, float4* __restrict__ dx) {
    const int Ho = H >> 1, Wo = W >> 1;
    const long total = (long)B * Ho * Wo * C4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C4);
        long p = i / C4;
        const int xo = (int)(p % Wo);
        p /= Wo;
        const int yo = (int)(p % Ho);
        const long b = p / Ho;
        const long base = ((b * H + 2 * yo) * W + 2 * xo) * C4 + c;
        const long i1 = base + C4, i2 = base + (long)W * C4, i3 = i2 + C4;
        const float4 a = x[base], b1 = x[i1], c1 = x[i2], d = x[i3], g = dy[i];
        float4 oa, ob, oc, od;
        route1(a.x, b1.x, c1.x, d.x, g.x, relu, oa.x, ob.x, oc.x, od.x);
        route1(a.y, b1.y, c1.y, d.y, g.y, relu, oa.y, ob.y, oc.y, od.y);
        route1(a.z, b1.z, c1.z, d.z, g.z, relu, oa.z, ob.z, oc.z, od.z);
        route1(a.w, b1.w, c1.w, d.w, g.w, relu, oa.w, ob.w, oc.w, od.w);
        dx[base] = oa; dx[i1] = ob; dx[i2] = oc; dx[i3] = od;
    }
}

// MaxPoolGrad + ReluGrad from the routing codes the pooled Winograd forward left (vc_conv3x3_wino_fwd_pool_f32): per pooled element four
// bits = position of the window's first maximum | 4 if that maximum is > 0.  Reads the pooled gradient and 2 bytes of codes per pooled
// pixel and channel quad instead of the whole pre-pool activation: 1.3 instead of 2.25 tensor passes.
// C4 activation layout: codes [planes = B * C/4][H/2][W/2] half-words (one per channel quad and pooled pixel),
// dy [planes][H/2][W/2][4], dx [planes][H][W][4].  A thread owns one pooled pixel of one plane: it writes 2 x 32 consecutive bytes, consecutive
// threads consecutive pieces (full lines).
__global__ __launch_bounds__(256) void maxpool_bwd_bits_c4_kernel(const unsigned short* __restrict__ bits, const float4* __restrict__ dy, long planes, int H,
                                                                  int W, float4* __restrict__ dx) {
    const int Ho = H >> 1, Wo = W >> 1;
    const long total = planes * Ho * Wo;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int xo = (int)(i % Wo);
        const long p = i / Wo;   // plane * Ho + yo
        const long base = (2 * p) * W + 2 * xo;
        const unsigned cd = bits[i];
        const float4 g = dy[i];
        float4 o[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            o[w].x = ((cd & 7u) == (4u | w)) ? g.x : 0.f;
            o[w].y = (((cd >> 4) & 7u) == (4u | w)) ? g.y : 0.f;
            o[w].z = (((cd >> 8) & 7u) == (4u | w)) ? g.z : 0.f;
            o[w].w = (((cd >> 12) & 7u) == (4u | w)) ? g.w : 0.f;
        }
        dx[base] = o[0]; dx[base + 1] = o[1]; dx[base + W] = o[2]; dx[base + W + 1] = o[3];
    }
}

// NHWC [B,H,W,C] <-> C4 [B][C/4][H][W][4] (vaecap.h: the activation layout of the Winograd kernels); one float4 per thread, the C4 side coalesced
__global__ __launch_bounds__(256) void nhwc_c4_kernel(const float4* __restrict__ in, long B, long HW, int C4, int to_c4, float4* __restrict__ out) {
    const long total = B * HW * C4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {   // i = C4-side index (b, q, p)
        const long p = i % HW;
        const long t = i / HW;
        const int q = (int)(t % C4);
        const long b = t / C4;
        const long j = (b * HW + p) * C4 + q;   // NHWC-side index
        if (to_c4) out[i] = in[j];
        else out[j] = in[i];
    }
}

// images - mean_rgb (utils/image_embeddings.py:31-34), RGB -> NHWC4 (4th channel zero)
__global__ __launch_bounds__(256) void preprocess_kernel(const float* __restrict__ img, long P, float4* __restrict__ out) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < P; i += (long)gridDim.x * 256)
        out[i] = make_float4(img[i * 3] - 123.68f, img[i * 3 + 1] - 116.779f, img[i * 3 + 2] - 103.939f, 0.f);
}

// the same from uint8 pixels (what the reference's HDF5 holds, preprocess.py:27-28; the feed's cast to float32 happens here): a thread
// converts FOUR pixels = three 4-byte words in, four 16-byte vectors out (P % 4 == 0; bytes of a word are little-endian pixels)
__global__ __launch_bounds__(256) void preprocess_u8_kernel(const uint32_t* __restrict__ img, long Q, float4* __restrict__ out) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < Q; i += (long)gridDim.x * 256) {
        const uint32_t a = img[3 * i], b = img[3 * i + 1], c = img[3 * i + 2];
        const float r0 = (float)(a & 255u), g0 = (float)((a >> 8) & 255u), b0 = (float)((a >> 16) & 255u);
        const float r1 = (float)(a >> 24), g1 = (float)(b & 255u), b1 = (float)((b >> 8) & 255u);
        const float r2 = (float)((b >> 16) & 255u), g2 = (float)(b >> 24), b2 = (float)(c & 255u);
        const float r3 = (float)((c >> 8) & 255u), g3 = (float)((c >> 16) & 255u), b3 = (float)(c >> 24);
        out[4 * i] = make_float4(r0 - 123.68f, g0 - 116.779f, b0 - 103.939f, 0.f);
        out[4 * i + 1] = make_float4(r1 - 123.68f, g1 - 116.779f, b1 - 103.939f, 0.f);
        out[4 * i + 2] = make_float4(r2 - 123.68f, g2 - 116.779f, b2 - 103.939f, 0.f);
        out[4 * i + 3] = make_float4(r3 - 123.68f, g3 - 116.779f, b3 - 103.939f, 0.f);
    }
}

// dst[o][c][i] = c < c_src ? src[o][c][i] : 0   (pad or truncate the middle dimension)
__global__ __launch_bounds__(256) void pad_dim_kernel(const float* __restrict__ src, long outer, int c_src, int c_dst, int inner,
                                                      float* __restrict__ dst) {
    const long total = outer * c_dst * inner;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int in = (int)(i % inner);
        const long t = i / inner;
        const int c = (int)(t % c_dst);
        const long o = t / c_dst;
        dst[i] = c < c_src ? src[(o * c_src + c) * inner + in] : 0.f;
    }
}

}  // namespace vc

using namespace vc;

extern "C" size_t vc_conv3x3_fwd_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
    ConvGeom g;
    if (make_geom(g, B, H, W, Cin, Cout)) return 0;
    return rounds_workspace<CONV_FWD>(g.P, Cout, 9 * Cin);
}

extern "C" size_t vc_conv3x3_dgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
    ConvGeom g;
    if (make_geom(g, B, H, W, Cin, Cout)) return 0;
    return rounds_workspace<CONV_DGRAD>(g.P, Cin, 9 * Cout);
}

extern "C" int vc_conv3x3_fwd_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* x, const float* w,
                                  const float* bias, float* y, int relu, float* ws, size_t ws_bytes) {
    ConvArgs c;
    VC_CHECK_ARG(x && w && y && B > 0 && H > 0 && W > 0, "bad argument");
    if (make_geom(c.g, B, H, W, Cin, Cout)) return fail(VC_EINVAL, "%s: Cin/Cout must be powers of two >= 4", __func__);
    c.a = x; c.b = w; c.out = y; c.aux = bias; c.relu = relu; c.kchunk = 0; c.bias_ws = nullptr;
    return launch_fwd_dgrad<CONV_FWD>((hipStream_t)stream, c, c.g.P, Cout, 9 * Cin, ws, ws_bytes);
}

extern "C" int vc_conv3x3_dgrad_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* dy, const float* w,
                                    const float* relu_src, float* dx, float* ws, size_t ws_bytes) {
    ConvArgs c;
    VC_CHECK_ARG(dy && w && dx && B > 0 && H > 0 && W > 0, "bad argument");
    if (make_geom(c.g, B, H, W, Cin, Cout)) return fail(VC_EINVAL, "%s: Cin/Cout must be powers of two >= 4", __func__);
    c.a = dy; c.b = w; c.out = dx; c.aux = relu_src; c.relu = 0; c.kchunk = 0; c.bias_ws = nullptr;
    return launch_fwd_dgrad<CONV_DGRAD>((hipStream_t)stream, c, c.g.P, Cin, 9 * Cout, ws, ws_bytes);
}

extern "C" size_t vc_conv3x3_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
    ConvGeom g;
    if (make_geom(g, B, H, W, Cin, Cout)) return 0;
    WgradPlan p = plan_wgrad(g);
    return (size_t)p.splits * (9L * Cin * Cout + Cout) * sizeof(float);
}

extern "C" int vc_conv3x3_wgrad_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* x, const float* dy,
                                    float* dw, float* db, int accumulate, float* ws, size_t ws_bytes) {
    ConvArgs c;
    VC_CHECK_ARG(x && dy && dw && B > 0 && H > 0 && W > 0, "bad argument");
    if (make_geom(c.g, B, H, W, Cin, Cout)) return fail(VC_EINVAL, "%s: Cin/Cout must be powers of two >= 4", __func__);
    WgradPlan p = plan_wgrad(c.g);
    const long MN = 9L * Cin * Cout;
    if (!ws || ws_bytes < (size_t)p.splits * (MN + Cout) * sizeof(float))
        return fail(VC_EWORKSPACE, "%s: workspace too small (need vc_conv3x3_wgrad_workspace_bytes)", __func__);
    c.a = x; c.b = dy; c.out = ws; c.aux = nullptr; c.relu = 0; c.kchunk = p.kchunk;
    c.bias_ws = db ? ws + (size_t)p.splits * MN : nullptr;
    launch_wgrad((hipStream_t)stream, c, 9 * Cin, Cout, p.splits);
    VC_LAUNCH_CHECK();
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(grid_for(MN)), dim3(256), 0, (hipStream_t)stream, ws, p.splits, MN, dw, accumulate);
    VC_LAUNCH_CHECK();
    if (db) {
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(grid_for(Cout)), dim3(256), 0, (hipStream_t)stream, c.bias_ws, p.splits, (long)Cout, db, accumulate);
        VC_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int vc_maxpool2x2_fwd_f32(void* stream, int B, int H, int W, int C, const float* x, float* y) {
    VC_CHECK_ARG(x && y && B > 0 && H > 0 && W > 0 && C > 0 && H % 2 == 0 && W % 2 == 0 && C % 4 == 0, "even H/W, C % 4 == 0 required");
    const long total = (long)B * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const float4*)x, B, H, W, C / 4, (float4*)y);
    VC_LAUNCH_CHECK();
    return 0;
}

extern "C" int vc_maxpool2x2_bwd_f32(void* stream, int B, int H, int W, int C, const float* x, const float* dy, float* dx,
                                     int relu_grad) {
    VC_CHECK_ARG(x && dy && dx && B > 0 && H > 0 && W > 0 && C > 0 && H % 2 == 0 && W % 2 == 0 && C % 4 == 0, "even H/W, C % 4 == 0 required");
    const long total = (long)B * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const float4*)x, (const float4*)dy, B, H, W, C / 4, relu_grad, (float4*)dx);
    VC_LAUNCH_CHECK();
    return 0;
}

extern "C" int vc_maxpool2x2_bwd_bits_f32(void* stream, int B, int H, int W, int C, const uint32_t* pool_bits, const float* dy, float* dx) {
    VC_CHECK_ARG(pool_bits && dy && dx && B > 0 && H > 0 && W > 0 && C > 0 && H % 2 == 0 && W % 2 == 0 && C % 4 == 0, "even H/W, C % 4 == 0 required");
    const long planes = (long)B * (C / 4), total = planes * (H / 2) * (W / 2);
    hipLaunchKernelGGL(maxpool_bwd_bits_c4_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)pool_bits, (const float4*)dy,
                       planes, H, W, (float4*)dx);
    VC_LAUNCH_CHECK();
    return 0;
}

extern "C" int vc_nhwc_to_c4_f32(void* stream, int B, int H, int W, int C, const float* nhwc, float* c4) {
    VC_CHECK_ARG(nhwc && c4 && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "C % 4 == 0 required");
    hipLaunchKernelGGL(nhwc_c4_kernel, dim3(grid_for((long)B * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream, (const float4*)nhwc, (long)B, (long)H * W, C / 4, 1, (float4*)c4);
    VC_LAUNCH_CHECK();
    return 0;
}

extern "C" int vc_c4_to_nhwc_f32(void* stream, int B, int H, int W, int C, const float* c4, float* nhwc) {
    VC_CHECK_ARG(nhwc && c4 && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "C % 4 == 0 required");
    hipLaunchKernelGGL(nhwc_c4_kernel, dim3(grid_for((long)B * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream, (const float4*)c4, (long)B, (long)H * W, C / 4, 0, (float4*)nhwc);
    VC_LAUNCH_CHECK();
    return 0;
}

extern "C" int vc_vgg_preprocess_f32(void* stream, const float* images, int B, int H, int W, float* out_nhwc4) {
    VC_CHECK_ARG(images && out_nhwc4 && B > 0 && H > 0 && W > 0, "bad argument");
    const long P = (long)B * H * W;
    hipLaunchKernelGGL(preprocess_kernel, dim3(grid_for(P)), dim3(256), 0, (hipStream_t)stream, images, P, (float4*)out_nhwc4);
    VC_LAUNCH_CHECK();
    return 0;
}

extern "C" int vc_vgg_preprocess_u8(void* stream, const void* images_u8, int B, int H, int W, float* out_nhwc4) {
    VC_CHECK_ARG(images_u8 && out_nhwc4 && B > 0 && H > 0 && W > 0, "bad argument");
    const long P = (long)B * H * W;
    VC_CHECK_ARG(P % 4 == 0 && (((uintptr_t)images_u8) & 3) == 0 && (((uintptr_t)out_nhwc4) & 15) == 0, "B*H*W must be a multiple of 4; images 4-byte, output 16-byte aligned");
    hipLaunchKernelGGL(preprocess_u8_kernel, dim3(grid_for(P / 4)), dim3(256), 0, (hipStream_t)stream, (const uint32_t*)images_u8, P / 4, (float4*)out_nhwc4);
    VC_LAUNCH_CHECK();
    return 0;
}

extern "C" int vc_pad_dim_f32(void* stream, const float* src, long outer, int c_src, int c_dst, int inner, float* dst) {
    VC_CHECK_ARG(src && dst && outer > 0 && c_src > 0 && c_dst > 0 && inner > 0, "bad argument");
    hipLaunchKernelGGL(pad_dim_kernel, dim3(grid_for(outer * c_dst * inner)), dim3(256), 0, (hipStream_t)stream, src, outer, c_src, c_dst, inner, dst);
    VC_LAUNCH_CHECK();
    return 0;
}
