// vc_gemm_f32: C = op(A).op(B) (+bias) (relu) (+=C), fp32 in / fp32 accumulate on MFMA.
// Replaces the TF matmul / tf.layers.dense call sites of the reference
// (main.py:94,108; vae_model/encoder.py:60-65,78-81,94-97; vae_model/decoder.py:111,127-129;
//  utils/image_embeddings.py:223,234) and every backward GEMM tf.gradients derives from them.
#include <type_traits>
#include "gemm_core.h"
#include "gemm_bf16x3_core.h"
#include "vaecap.h"
#ifdef VC_MICROBENCH
#include "vaecap_microbench.h"
#endif

namespace vc {

struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    float* ws;
    long lda, ldb, ldc;
    int M, N, K;
    int tiles_n, ntiles;
    int kchunk;  // K range per split (multiple of 32)
    int splits;
    int flags;
    int tile0;       // first tile of this launch (tail launches start after the main launch's tiles)
    long ws_row0;    // split partials are stored as ws[split][row - ws_row0][N] with ws_rows rows per split
    long ws_rows;
};

// PREC 0: f32 MFMA (v_mfma_f32_32x32x2_f32); 1: split-bf16, three v_mfma_f32_32x32x16_bf16 per k-step (gemm_bf16x3_core.h)
template <class CFG, int AM, int BMD, bool VEC, int ABL, int PREC>
__device__ __forceinline__ void gemm_body(const GemmArgs& g, float* smem) {
    int id = xcd_remap(blockIdx.x, g.ntiles);
    int tm_, tn_;
    if (PREC == 1) {
        // Grouped order: the ~96 workgroups an XCD holds at a time cover GM tile rows x ~12 tile columns instead of one row x all
        // columns, so BOTH operand panels of the window stay in its 4 MB L2 (at 16x the f32 MFMA rate this kernel is bound by what
        // it stages, and a [K, 10000] B operand streamed once per tile row was half of that).  Same tiles, same arithmetic.
        constexpr int GM = 8;
        const int tiles_m = g.ntiles / g.tiles_n, per = GM * g.tiles_n;
        const int grp = id / per, r = id - grp * per;
        const int gm = min(GM, tiles_m - grp * GM);
        tm_ = grp * GM + r % gm;
        tn_ = r / gm;
    } else {
        tm_ = id / g.tiles_n;
        tn_ = id % g.tiles_n;
    }
    const int m0 = (tm_ + g.tile0 / g.tiles_n) * CFG::BM;
    const int n0 = tn_ * CFG::BN;
    const int kb = blockIdx.y * g.kchunk;
    const int ke = min(g.K, kb + g.kchunk);
    f32x16 acc[CFG::TM][CFG::TN];
    acc_zero<CFG>(acc);
    typename std::conditional<AM == MODE_MK, LoadMK<VEC>, LoadKM<VEC>>::type la;
    la.p = g.A; la.ld = g.lda; la.R = g.M; la.K = g.K;
    typename std::conditional<BMD == MODE_MK, LoadMK<VEC>, LoadKM<VEC>>::type lb;
    lb.p = g.B; lb.ld = g.ldb; lb.R = g.N; lb.K = g.K;
#ifdef VC_MICROBENCH  // ablation variants exist only in the microbenchmark build (make microbench), never in libvaecap.so
    if (ABL == 8) mfma_mainloop_db<CFG, AM, BMD>(acc, la, lb, m0, n0, kb, ke, smem);
    else
#endif
    if (PREC == 1) {
        constexpr bool DB = CFG::NT == 512;   // the one-workgroup-per-CU tile: two LDS images
        if (VEC && g.K >= 32) mfma_mainloop_bf16x3<CFG, AM, BMD, true, DB>(acc, la, lb, m0, n0, kb, ke, smem);
        else mfma_mainloop_bf16x3<CFG, AM, BMD, false, DB>(acc, la, lb, m0, n0, kb, ke, smem);
    }
    else mfma_mainloop<CFG, AM, BMD, decltype(la), decltype(lb), ABL>(acc, la, lb, m0, n0, kb, ke, smem);
    const bool split = g.splits > 1;
    float* out = split ? g.ws + ((long)blockIdx.y * g.ws_rows - g.ws_row0) * g.N : g.C;
    const long ldo = split ? g.N : g.ldc;
    const bool vec_out = ((ldo & 3) == 0) && ((((uintptr_t)out) & 15) == 0);
    epilogue_rows<CFG>(acc, smem, [&](int r, int c, float4 v) {
        const int row = m0 + r, col = n0 + c;
        if (row >= g.M || col >= g.N) return;
        float* p = out + (long)row * ldo + col;
        const bool full = vec_out && col + 3 < g.N;
        if (!split) {
            if (g.bias) {
                v.x += g.bias[col];
                if (col + 1 < g.N) v.y += g.bias[col + 1];
                if (col + 2 < g.N) v.z += g.bias[col + 2];
                if (col + 3 < g.N) v.w += g.bias[col + 3];
            }
            if (g.flags & VC_GEMM_ACCUMULATE) {
                if (full) {
                    const float4 o = *reinterpret_cast<const float4*>(p);
                    v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                } else {
                    v.x += p[0];
                    if (col + 1 < g.N) v.y += p[1];
                    if (col + 2 < g.N) v.z += p[2];
                    if (col + 3 < g.N) v.w += p[3];
                }
            }
            if (g.flags & VC_GEMM_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        }
        if (full) {
            *reinterpret_cast<float4*>(p) = v;
        } else {
            p[0] = v.x;
            if (col + 1 < g.N) p[1] = v.y;
            if (col + 2 < g.N) p[2] = v.z;
            if (col + 3 < g.N) p[3] = v.w;
        }
    });
}

template <class CFG, int AM, int BMD, bool VEC, int ABL = 0>
__global__ __launch_bounds__(CFG::NT) void gemm_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    gemm_body<CFG, AM, BMD, VEC, ABL, 0>(g, smem);
}
// the split-bf16 form of the same kernel (own symbol: its register budget is set for THREE workgroups per CU -- the loop waits on
// one K-tile of global loads per iteration, and a third resident workgroup covers that latency)
#ifndef VC_BX_WAVES
#define VC_BX_WAVES 3
#endif
template <class CFG, int AM, int BMD, bool VEC>
__global__ __launch_bounds__(CFG::NT, CFG::NT == 512 ? 2 : VC_BX_WAVES) void gemm_bx_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    gemm_body<CFG, AM, BMD, VEC, 0, 1>(g, smem);
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int splits, long MN, int N,
                                                           float* __restrict__ C, long ldc, const float* __restrict__ bias,
                                                           int flags) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < MN; i += (long)gridDim.x * 256) {
        float v = 0.f;
        for (int z = 0; z < splits; ++z) v += ws[(long)z * MN + i];  // fixed order: deterministic
        const int col = (int)(i % N);
        const long row = i / N;
        if (bias) v += bias[col];
        float* p = C + row * ldc + col;
        if (flags & VC_GEMM_ACCUMULATE) v += *p;
        if (flags & VC_GEMM_RELU) v = fmaxf(v, 0.f);
        *p = v;
    }
}

// the same with 16-byte accesses (N % 4 == 0, ldc % 4 == 0, 16-byte aligned C / bias / ws): same summation order per element
__global__ __launch_bounds__(256) void splitk_reduce4_kernel(const float4* __restrict__ ws, int splits, long Q, int NQ,
                                                            float* __restrict__ C, long ldc, const float* __restrict__ bias,
                                                            int flags) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < Q; i += (long)gridDim.x * 256) {
        float4 v = ws[i];
        for (int z = 1; z < splits; ++z) {
            const float4 t = ws[(long)z * Q + i];
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        const int cq = (int)(i % NQ);
        const long row = i / NQ;
        if (bias) {
            const float4 b = *reinterpret_cast<const float4*>(bias + cq * 4);
            v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        }
        float4* p = reinterpret_cast<float4*>(C + row * ldc + cq * 4);
        if (flags & VC_GEMM_ACCUMULATE) {
            const float4 o = *p;
            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
        if (flags & VC_GEMM_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        *p = v;
    }
}

static void launch_splitk_reduce(hipStream_t st, const float* ws, int splits, long MN, int N, float* C, long ldc, const float* bias, int flags) {
    auto al = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    if ((N & 3) == 0 && (ldc & 3) == 0 && al(ws) && al(C) && (!bias || al(bias))) {
        const long Q = MN >> 2;
        int blocks = cdiv(Q, 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(splitk_reduce4_kernel, dim3(blocks), dim3(256), 0, st, reinterpret_cast<const float4*>(ws), splits, Q, N >> 2, C, ldc, bias, flags);
        return;
    }
    int blocks = cdiv(MN, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, ws, splits, MN, N, C, ldc, bias, flags);
}

struct GemmPlan {
    bool big;  // 128x128 tile, else 64x64
    bool skinny;  // M <= 64: 64 x 128 tiles (four waves of 64 x 32) -- a 128-row tile would spend half its MFMAs on zero rows, and the
                  // 64-row fc products (13 GFLOP over 411 MB of weights) are then MFMA-bound at twice their HBM time
    int splits, kchunk, tiles_m, tiles_n;
    // whole rounds (128x128 tiles, no global split-K): the tile rows beyond the last full round of 768 resident workgroups
    // run as a second launch with K split `tail_splits` ways (see conv.hip launch_rounds for the measurement behind it)
    int main_m, tail_splits, tail_kchunk;
};

static GemmPlan plan_gemm(int M, int N, int K) {
    GemmPlan p;
    p.skinny = false;
    if (M <= 64 && N >= 256 && K >= 512) {
        p.skinny = true; p.big = false;
        p.tiles_m = 1; p.tiles_n = cdiv(N, 128);
        long sa = 768 / p.tiles_n;           // one round of resident workgroups
        const long maxs = K / 256;           // >= 8 K-tiles per split
        if (sa > maxs) sa = maxs;
        if (sa > 64) sa = 64;
        if (sa < 1) sa = 1;
        p.kchunk = cdiv(cdiv(K, (int)sa), 32) * 32;
        p.splits = cdiv(K, p.kchunk);
        p.main_m = 1; p.tail_splits = 1; p.tail_kchunk = 0;
        return p;
    }
    // Measured on MI355X (tools/microbench.py): 128x128 tiles win when there are >= 1.5 per CU, and for
    // small outputs with a deep K (weight gradients) when K is split into chunks of >= 512 so that every
    // CU gets 2-3 of them; in between (e.g. 512 x 10000 x 25600) 64x64 tiles with ~5 workgroups per CU
    // beat 128x128 with a ragged 2.5 per CU.
    const long t128 = (long)cdiv(M, 128) * cdiv(N, 128);
    long sa = 1;
    if (t128 <= 128) {
        sa = 768 / t128;  // one whole round of the 768 resident 128 x 128 workgroups (a ragged 640 cost ~15 %)
        const long maxs = K / 512;
        if (sa > maxs) sa = maxs;
        if (sa > 64) sa = 64;
        if (sa < 1) sa = 1;
    } else if (t128 < 768) {
        // Less than one round of 128 x 128 tiles: K is split so that ~1200 shorter workgroups share the chip (finer
        // granularity evens out the 1-vs-2 workgroups per CU of an unsplit launch), as long as a split keeps >= 20
        // K-tiles.  Measured (tools/microbench.py mid): 6400 x 512 x 10000  83 -> 100 TFLOP/s (S = 5..6),
        // 512 x 10000 x 6400  91 -> 102 (S = 4), 12800 x 512 x 10000  104 -> 116 (S = 3), 512 x 10000 x 25600  106 -> 121.
        sa = (1200 + t128 / 2) / t128;
        const long maxs = K / 640;
        if (sa > maxs) sa = maxs;
        if (sa < 1) sa = 1;
    }
    p.big = t128 >= 384 || (t128 <= 128 && t128 * sa >= 256) || (t128 > 128 && sa > 1);
    const int b = p.big ? 128 : 64;
    p.tiles_m = cdiv(M, b);
    p.tiles_n = cdiv(N, b);
    const long tiles = (long)p.tiles_m * p.tiles_n;
    int splits = (int)sa;
    if (!p.big) {
        splits = 1;
        if (tiles < 192) {
            splits = (int)((512 + tiles - 1) / tiles);
            const int maxs = K / 256;  // keep >= 8 K-tiles per split
            if (splits > maxs) splits = maxs;
            if (splits > 64) splits = 64;
            if (splits < 1) splits = 1;
        }
    }
    int kchunk = cdiv(cdiv(K, splits), 32) * 32;
    if (kchunk < 32) kchunk = 32;
    p.splits = cdiv(K, kchunk);
    p.kchunk = kchunk;
    p.main_m = p.tiles_m; p.tail_splits = 1; p.tail_kchunk = 0;
    if (p.big && p.splits == 1) {
        const int slots = 768, T = p.tiles_m * p.tiles_n;
        const int full = (T / slots) * slots;
        if (full > 0 && full < T) {
            const int main_m = full / p.tiles_n;
            const int tail = T - main_m * p.tiles_n;
            const int ktiles = cdiv(K, 32);
            int S = slots / tail;
            if (S > ktiles / 16) S = ktiles / 16;  // >= 16 K-tiles per split (K = 512: any split of the tail measured slower than none)
            if (S > 32) S = 32;
            if (S >= 2) {
                p.tail_kchunk = cdiv(ktiles, S) * 32;
                p.tail_splits = cdiv(K, p.tail_kchunk);
                p.main_m = main_m;
            }
        }
    }
    return p;
}

template <class CFG, int AM, int BMD, bool VEC, int PREC>
static void launch_gemm(hipStream_t st, const GemmArgs& g) {
    dim3 grid(g.ntiles, g.splits);
    if (PREC == 1) {
        constexpr int LDS = CFG::NT == 512 ? 2 * (CFG::A_FLOATS + CFG::B_FLOATS) * 4 : CFG::SMEM_BYTES;   // 256 x 256: two images, 144 KB
        static_assert(LDS >= CFG::SMEM_BYTES && LDS <= 160 * 1024, "LDS budget");
        if (LDS > 65536) {   // set at every launch: a host-side attribute write, valid whichever device is current (cheap beside a 256 x 256-tile launch)
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bx_kernel<CFG, AM, BMD, VEC>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
            if (e != hipSuccess) { fail((int)e, "%s: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed", "gemm_bx_kernel"); return; }
        }
        hipLaunchKernelGGL((gemm_bx_kernel<CFG, AM, BMD, VEC>), grid, dim3(CFG::NT), LDS, st, g);
    } else hipLaunchKernelGGL((gemm_kernel<CFG, AM, BMD, VEC>), grid, dim3(CFG::NT), CFG::SMEM_BYTES, st, g);
}

template <class CFG, bool VEC, int PREC>
static void dispatch_modes_p(hipStream_t st, const GemmArgs& g, int ta, int tb) {
    // ta: A stored [K,M] (row-contiguous) -> KM.  tb == 0: B stored [K,N] -> KM; tb: B stored [N,K] -> MK.
    if (!ta && !tb) launch_gemm<CFG, MODE_MK, MODE_KM, VEC, PREC>(st, g);
    else if (!ta && tb) launch_gemm<CFG, MODE_MK, MODE_MK, VEC, PREC>(st, g);
    else if (ta && !tb) launch_gemm<CFG, MODE_KM, MODE_KM, VEC, PREC>(st, g);
    else launch_gemm<CFG, MODE_KM, MODE_MK, VEC, PREC>(st, g);
}
template <class CFG, bool VEC>
static void dispatch_modes(hipStream_t st, const GemmArgs& g, int ta, int tb, int prec = 0) {
    if (prec == 1) dispatch_modes_p<CFG, VEC, 1>(st, g, ta, tb);
    else dispatch_modes_p<CFG, VEC, 0>(st, g, ta, tb);
}

using Cfg128 = TileCfg<2, 2, 2, 2>;
using Cfg64 = TileCfg<2, 2, 1, 1>;
using CfgSkinny = TileCfg<1, 4, 2, 1>;  // 64 x 128: four waves side by side, each 64 rows x 32 columns
using CfgBx256 = TileCfg<2, 4, 4, 2>;   // split-bf16 only: 256 x 256, eight waves of 128 x 64 (half the staged bytes per MAC of 128 x 128)

// Split-bf16 plan for large products: 256 x 256 tiles, ONE workgroup per CU.  At 16x the f32 MFMA rate the kernel is bound by
// the bytes it stages (measured: the 128 x 128 form saturates at ~7-8 TB/s of L2 -> LDS traffic, 16 MACs per byte), so the tile
// is as large as the accumulator file allows (32 MACs per byte).  K is split only to fill the last round of 256 workgroups.
struct BxPlan {
    bool use;
    int tiles_m, tiles_n, splits, kchunk;
};
static BxPlan plan_bx256(int M, int N, int K) {
    BxPlan p;
    p.use = false;
    if (M < 192 || N < 192) return p;
    p.tiles_m = cdiv(M, 256); p.tiles_n = cdiv(N, 256);
    const long t = (long)p.tiles_m * p.tiles_n;
    int best = 1;
    double beff = 0;
    const int maxs = K / 1024 < 1 ? 1 : (K / 1024 > 16 ? 16 : K / 1024);
    for (int s = 1; s <= maxs; ++s) {
        const long w = t * s;
        const double eff = (double)w / (double)(cdiv(w, 256) * 256L) - 0.02 * (s - 1);   // (a split costs a workspace round trip)
        if (eff > beff + 1e-9) { beff = eff; best = s; }
    }
    if (t * best < 200) return p;   // under one round: the 128 x 128 plan spreads the work better
    p.splits = best;
    p.kchunk = cdiv(cdiv(K, best), 32) * 32;
    p.splits = cdiv(K, p.kchunk);
    p.use = true;
    return p;
}

// Split-K partial products only (no reduce, no bias): ws[s][M][N] = op(A) op(B) over the s-th K range, 64 x 64 tiles.
// For consumers that sum the partials themselves (the LSTM gate kernels).  Returns the number of splits written
// (chosen so that tiles x splits is about one round of resident workgroups, each split >= 2 K-tiles, <= max_splits).
struct PartialPlan {
    bool big;
    int splits, kchunk;
};

static PartialPlan plan_partials(int M, int N, int K, int max_splits) {
    PartialPlan p;
    const long t128 = (long)cdiv(M, 128) * cdiv(N, 128);
    p.big = false;  // 128 x 128 tiles measured slower for the LSTM step products at N = 320 and N = 1280 (+10-20 % per step)
    (void)t128;
    const long tiles = p.big ? t128 : (long)cdiv(M, 64) * cdiv(N, 64);
    long S = 768 / tiles;
    const int min_k = p.big ? 128 : 64;
    if (S > K / min_k) S = K / min_k;
    if (S > max_splits) S = max_splits;
    if (S < 1) S = 1;
    p.kchunk = cdiv(cdiv(K, (int)S), 32) * 32;
    p.splits = cdiv(K, p.kchunk);
    return p;
}

int gemm_partials_f32(hipStream_t st, int ta, int tb, int M, int N, int K, const float* A, long lda, const float* B, long ldb,
                      float* ws, size_t ws_bytes, int max_splits, int* splits_out) {
    const PartialPlan pp = plan_partials(M, N, K, max_splits);
    const int b = pp.big ? 128 : 64;
    const int tiles_m = cdiv(M, b), tiles_n = cdiv(N, b);
    const int splits = pp.splits, kchunk = pp.kchunk;
    if (!ws || ws_bytes < (size_t)splits * M * N * sizeof(float)) return fail(VC_EWORKSPACE, "%s: workspace too small", __func__);
    GemmArgs g;
    g.A = A; g.B = B; g.C = ws; g.bias = nullptr; g.ws = ws;
    g.lda = lda; g.ldb = ldb; g.ldc = N; g.M = M; g.N = N; g.K = K;
    g.tiles_n = tiles_n; g.ntiles = tiles_m * tiles_n; g.kchunk = kchunk; g.splits = splits; g.flags = 0;
    g.tile0 = 0; g.ws_row0 = 0; g.ws_rows = M;
    auto al = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    const bool vec = al(A) && al(B) && (lda % 4 == 0) && (ldb % 4 == 0) && ((ta ? M : K) % 4 == 0) && ((tb ? K : N) % 4 == 0);
    if (pp.big) {
        if (vec) dispatch_modes<Cfg128, true>(st, g, ta, tb); else dispatch_modes<Cfg128, false>(st, g, ta, tb);
    } else {
        if (vec) dispatch_modes<Cfg64, true>(st, g, ta, tb); else dispatch_modes<Cfg64, false>(st, g, ta, tb);
    }
    *splits_out = splits;
    return launch_status(__func__);
}

size_t gemm_partials_bytes(int M, int N, int K, int max_splits) {
    return (size_t)plan_partials(M, N, K, max_splits).splits * M * N * sizeof(float);
}

}  // namespace vc

extern "C" size_t vc_gemm_workspace_bytes(int M, int N, int K) {
    vc::GemmPlan p = vc::plan_gemm(M, N, K);
    size_t bx = 0;   // the split-bf16 plan may split K differently: the query covers both precisions
    {
        const vc::BxPlan b = vc::plan_bx256(M, N, K);
        if (b.use && b.splits > 1) bx = (size_t)b.splits * M * N * sizeof(float);
    }
    if (bx) {
        size_t f = 0;
        if (p.splits > 1) f = (size_t)p.splits * M * N * sizeof(float);
        else if (p.tail_splits > 1) f = (size_t)p.tail_splits * (M - (long)p.main_m * 128) * N * sizeof(float);
        return f > bx ? f : bx;
    }
    if (p.splits > 1) return (size_t)p.splits * M * N * sizeof(float);
    if (p.tail_splits > 1) return (size_t)p.tail_splits * (M - (long)p.main_m * 128) * N * sizeof(float);
    return 0;
}

namespace vc {
// DEPRECATED process-wide default (vc_gemm_set_precision): what a vc_gemm_f32 call WITHOUT the VC_GEMM_BF16X3 flag computes with.  The
// product path never sets it: engine.CaptionEngine / trainer.VggEngine pass their precision with every call (ABI 4).
static int g_gemm_precision = 0;
int gemm_default_precision() { return g_gemm_precision; }
}

extern "C" int vc_gemm_set_precision(int mode) {
    VC_CHECK_ARG(mode == 0 || mode == 1, "0 = f32 MFMA, 1 = split-bf16 (bf16x3)");
    vc::g_gemm_precision = mode;
    return 0;
}
extern "C" int vc_gemm_get_precision(void) { return vc::g_gemm_precision; }

static int gemm_impl(int prec, const char* fn, void* stream, int ta, int tb, int M, int N, int K, const float* A, long lda, const float* B,
                     long ldb, float* C, long ldc, const float* bias, int flags, float* ws, size_t ws_bytes);

extern "C" int vc_gemm_f32(void* stream, int ta, int tb, int M, int N, int K, const float* A, long lda, const float* B,
                           long ldb, float* C, long ldc, const float* bias, int flags, float* ws, size_t ws_bytes) {
    return gemm_impl((flags & VC_GEMM_BF16X3) ? 1 : vc::g_gemm_precision, __func__, stream, ta, tb, M, N, K, A, lda, B, ldb, C, ldc, bias, flags & ~VC_GEMM_BF16X3, ws, ws_bytes);
}
extern "C" int vc_gemm_bf16x3_f32(void* stream, int ta, int tb, int M, int N, int K, const float* A, long lda, const float* B,
                                  long ldb, float* C, long ldc, const float* bias, int flags, float* ws, size_t ws_bytes) {
    return gemm_impl(1, __func__, stream, ta, tb, M, N, K, A, lda, B, ldb, C, ldc, bias, flags & ~VC_GEMM_BF16X3, ws, ws_bytes);
}

// (argument / launch checks of the shared body report the ENTRY's name)
#define VC_GEMM_ARG(cond, what)                                               \
    do {                                                                      \
        if (!(cond)) return vc::fail(vc::VC_EINVAL, "%s: invalid argument: " what, fn); \
    } while (0)
#define VC_GEMM_LAUNCHED()                                                    \
    do {                                                                      \
        int s__ = vc::launch_status(fn);                                      \
        if (s__) return s__;                                                  \
    } while (0)

static int gemm_impl(int prec, const char* fn, void* stream, int ta, int tb, int M, int N, int K, const float* A, long lda, const float* B,
                     long ldb, float* C, long ldc, const float* bias, int flags, float* ws, size_t ws_bytes) {
    using namespace vc;
    VC_GEMM_ARG(M >= 0 && N >= 0 && K >= 0, "negative dimension");
    if (M == 0 || N == 0) return 0;
    VC_GEMM_ARG(A && B && C, "null operand");
    VC_GEMM_ARG(lda >= (ta ? M : K) && ldb >= (tb ? K : N) && ldc >= N, "leading dimension too small");
    hipStream_t st = (hipStream_t)stream;
    auto al = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    const bool vec = al(A) && al(B) && (lda % 4 == 0) && (ldb % 4 == 0) && ((ta ? M : K) % 4 == 0) &&
                     ((tb ? K : N) % 4 == 0);
    if (prec == 1) {
        const BxPlan b = plan_bx256(M, N, K);
        if (b.use) {
            if (b.splits > 1 && (!ws || ws_bytes < (size_t)b.splits * M * N * sizeof(float)))
                return fail(VC_EWORKSPACE, "%s: workspace too small (need vc_gemm_workspace_bytes)", fn);
            GemmArgs g;
            g.A = A; g.B = B; g.C = C; g.bias = bias; g.ws = ws;
            g.lda = lda; g.ldb = ldb; g.ldc = ldc;
            g.M = M; g.N = N; g.K = K;
            g.tiles_n = b.tiles_n; g.ntiles = b.tiles_m * b.tiles_n;
            g.kchunk = b.kchunk; g.splits = b.splits; g.flags = flags;
            g.tile0 = 0; g.ws_row0 = 0; g.ws_rows = M;
            if (vec) dispatch_modes_p<CfgBx256, true, 1>(st, g, ta, tb); else dispatch_modes_p<CfgBx256, false, 1>(st, g, ta, tb);
            VC_GEMM_LAUNCHED();
            if (b.splits > 1) {
                launch_splitk_reduce(st, ws, b.splits, (long)M * N, N, C, ldc, bias, flags);
                VC_GEMM_LAUNCHED();
            }
            return 0;
        }
    }
    GemmPlan p = plan_gemm(M, N, K);
    if (p.splits > 1 && (!ws || ws_bytes < (size_t)p.splits * M * N * sizeof(float)))
        return fail(VC_EWORKSPACE, "%s: workspace too small (need vc_gemm_workspace_bytes)", fn);
    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.ws = ws;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.M = M; g.N = N; g.K = K;
    g.tiles_n = p.tiles_n; g.ntiles = p.tiles_m * p.tiles_n;
    g.kchunk = p.kchunk; g.splits = p.splits; g.flags = flags;
    g.tile0 = 0; g.ws_row0 = 0; g.ws_rows = M;
    const long tail_row0 = (long)p.main_m * 128;
    const bool tail = p.tail_splits > 1 && ws && ws_bytes >= (size_t)p.tail_splits * (M - tail_row0) * N * sizeof(float);
    if (tail) g.ntiles = p.main_m * p.tiles_n;  // whole rounds; the remaining tile rows follow as a K-split launch
    if (p.skinny) {
        if (vec) dispatch_modes<CfgSkinny, true>(st, g, ta, tb, prec); else dispatch_modes<CfgSkinny, false>(st, g, ta, tb, prec);
    } else if (p.big) {
        if (vec) dispatch_modes<Cfg128, true>(st, g, ta, tb, prec); else dispatch_modes<Cfg128, false>(st, g, ta, tb, prec);
    } else {
        if (vec) dispatch_modes<Cfg64, true>(st, g, ta, tb, prec); else dispatch_modes<Cfg64, false>(st, g, ta, tb, prec);
    }
    VC_GEMM_LAUNCHED();
    if (p.splits > 1) {
        launch_splitk_reduce(st, ws, p.splits, (long)M * N, N, C, ldc, bias, flags);
        VC_GEMM_LAUNCHED();
    }
    if (tail) {
        GemmArgs t = g;
        t.tile0 = p.main_m * p.tiles_n;
        t.ntiles = (p.tiles_m - p.main_m) * p.tiles_n;
        t.kchunk = p.tail_kchunk; t.splits = p.tail_splits;
        t.ws_row0 = tail_row0; t.ws_rows = M - tail_row0;
        if (vec) dispatch_modes<Cfg128, true>(st, t, ta, tb, prec); else dispatch_modes<Cfg128, false>(st, t, ta, tb, prec);
        VC_GEMM_LAUNCHED();
        launch_splitk_reduce(st, ws, t.splits, t.ws_rows * N, N, C + tail_row0 * ldc, ldc, bias, flags);
        VC_GEMM_LAUNCHED();
    }
    return 0;
}

#ifdef VC_MICROBENCH
// Benchmark-only entry (tools/microbench.py ablate, built by `make microbench` into libvaecap_microbench.so): the NN 128x128
// kernel with parts of its main loop ablated (results are then meaningless); variant 0 == vc_gemm_f32's kernel for aligned NN operands.
extern "C" int vc_debug_gemm_ablate_f32(void* stream, int variant, int M, int N, int K, const float* A, const float* B, float* C) {
    using namespace vc;
    VC_CHECK_ARG(A && B && C && M % 128 == 0 && N % 128 == 0 && K % 32 == 0, "debug entry: multiples of 128/128/32 only");
    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.bias = nullptr; g.ws = nullptr;
    g.lda = K; g.ldb = N; g.ldc = N; g.M = M; g.N = N; g.K = K;
    g.tiles_n = N / 128; g.ntiles = (M / 128) * (N / 128); g.kchunk = K; g.splits = 1; g.flags = 0;
    g.tile0 = 0; g.ws_row0 = 0; g.ws_rows = M;
    dim3 grid(g.ntiles, 1);
    hipStream_t st = (hipStream_t)stream;
#define VC_ABL(V) case V: hipLaunchKernelGGL((gemm_kernel<Cfg128, MODE_MK, MODE_KM, true, V>), grid, dim3(Cfg128::NT), Cfg128::SMEM_BYTES, st, g); break;
    if (variant == 8) {
        hipLaunchKernelGGL((gemm_kernel<Cfg128, MODE_MK, MODE_KM, true, 8>), grid, dim3(Cfg128::NT), 2 * Cfg128::SMEM_BYTES, st, g);
        VC_LAUNCH_CHECK();
        return 0;
    }
    switch (variant) { VC_ABL(0) VC_ABL(1) VC_ABL(2) VC_ABL(3) VC_ABL(4) VC_ABL(6) VC_ABL(7) default: return fail(VC_EINVAL, "%s: unknown variant", __func__); }
#undef VC_ABL
    VC_LAUNCH_CHECK();
    return 0;
}
#endif  // VC_MICROBENCH
