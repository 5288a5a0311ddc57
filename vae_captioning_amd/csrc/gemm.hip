// vc_gemm_f32: C = op(A).op(B) (+bias) (relu) (+=C), fp32 in / fp32 accumulate on MFMA.
// Replaces the TF matmul / tf.layers.dense call sites of the reference
// (main.py:94,108; vae_model/encoder.py:60-65,78-81,94-97; vae_model/decoder.py:111,127-129;
//  utils/image_embeddings.py:223,234) and every backward GEMM tf.gradients derives from them.
#include <type_traits>
#include "gemm_core.h"
#include "vaecap.h"

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
};

template <class CFG, int AM, int BMD, bool VEC>
__global__ __launch_bounds__(CFG::NT) void gemm_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int id = xcd_remap(blockIdx.x, g.ntiles);
    const int m0 = (id / g.tiles_n) * CFG::BM;
    const int n0 = (id % g.tiles_n) * CFG::BN;
    const int kb = blockIdx.y * g.kchunk;
    const int ke = min(g.K, kb + g.kchunk);
    f32x16 acc[CFG::TM][CFG::TN];
    acc_zero<CFG>(acc);
    typename std::conditional<AM == MODE_MK, LoadMK<VEC>, LoadKM<VEC>>::type la{g.A, g.lda, g.M, g.K};
    typename std::conditional<BMD == MODE_MK, LoadMK<VEC>, LoadKM<VEC>>::type lb{g.B, g.ldb, g.N, g.K};
    mfma_mainloop<CFG, AM, BMD>(acc, la, lb, m0, n0, kb, ke, smem);
    AccCoord<CFG> co;
    const bool split = g.splits > 1;
    float* out = split ? g.ws + (long)blockIdx.y * g.M * g.N : g.C;
    const long ldo = split ? g.N : g.ldc;
#pragma unroll
    for (int tn = 0; tn < CFG::TN; ++tn) {
        const int col = n0 + co.col(tn);
        if (col >= g.N) continue;
        const float bv = (!split && g.bias) ? g.bias[col] : 0.f;
#pragma unroll
        for (int tm = 0; tm < CFG::TM; ++tm) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + co.row(tm, r);
                if (row >= g.M) continue;
                float v = acc[tm][tn][r] + bv;
                float* p = out + (long)row * ldo + col;
                if (!split) {
                    if (g.flags & VC_GEMM_ACCUMULATE) v += *p;
                    if (g.flags & VC_GEMM_RELU) v = fmaxf(v, 0.f);
                }
                *p = v;
            }
        }
    }
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

struct GemmPlan {
    bool big;  // 128x128 tile, else 64x64
    int splits, kchunk, tiles_m, tiles_n;
};

static GemmPlan plan_gemm(int M, int N, int K) {
    GemmPlan p;
    const long t128 = (long)cdiv(M, 128) * cdiv(N, 128);
    p.big = t128 >= 160;
    const int b = p.big ? 128 : 64;
    p.tiles_m = cdiv(M, b);
    p.tiles_n = cdiv(N, b);
    const long tiles = (long)p.tiles_m * p.tiles_n;
    int splits = 1;
    if (tiles < 192) {
        splits = (int)((512 + tiles - 1) / tiles);
        const int maxs = K / 256;  // keep >= 8 K-tiles per split
        if (splits > maxs) splits = maxs;
        if (splits > 64) splits = 64;
        if (splits < 1) splits = 1;
    }
    int kchunk = cdiv(cdiv(K, splits), 32) * 32;
    if (kchunk < 32) kchunk = 32;
    p.splits = cdiv(K, kchunk);
    p.kchunk = kchunk;
    return p;
}

template <class CFG, int AM, int BMD, bool VEC>
static void launch_gemm(hipStream_t st, const GemmArgs& g) {
    dim3 grid(g.ntiles, g.splits);
    hipLaunchKernelGGL((gemm_kernel<CFG, AM, BMD, VEC>), grid, dim3(CFG::NT), CFG::SMEM_BYTES, st, g);
}

template <class CFG, bool VEC>
static void dispatch_modes(hipStream_t st, const GemmArgs& g, int ta, int tb) {
    // ta: A stored [K,M] (row-contiguous) -> KM.  tb == 0: B stored [K,N] -> KM; tb: B stored [N,K] -> MK.
    if (!ta && !tb) launch_gemm<CFG, MODE_MK, MODE_KM, VEC>(st, g);
    else if (!ta && tb) launch_gemm<CFG, MODE_MK, MODE_MK, VEC>(st, g);
    else if (ta && !tb) launch_gemm<CFG, MODE_KM, MODE_KM, VEC>(st, g);
    else launch_gemm<CFG, MODE_KM, MODE_MK, VEC>(st, g);
}

using Cfg128 = TileCfg<2, 2, 2, 2>;
using Cfg64 = TileCfg<2, 2, 1, 1>;

}  // namespace vc

extern "C" size_t vc_gemm_workspace_bytes(int M, int N, int K) {
    vc::GemmPlan p = vc::plan_gemm(M, N, K);
    return p.splits > 1 ? (size_t)p.splits * M * N * sizeof(float) : 0;
}

extern "C" int vc_gemm_f32(void* stream, int ta, int tb, int M, int N, int K, const float* A, long lda, const float* B,
                           long ldb, float* C, long ldc, const float* bias, int flags, float* ws, size_t ws_bytes) {
    using namespace vc;
    VC_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, "negative dimension");
    if (M == 0 || N == 0) return 0;
    VC_CHECK_ARG(A && B && C, "null operand");
    VC_CHECK_ARG(lda >= (ta ? M : K) && ldb >= (tb ? K : N) && ldc >= N, "leading dimension too small");
    hipStream_t st = (hipStream_t)stream;
    GemmPlan p = plan_gemm(M, N, K);
    if (p.splits > 1 && (!ws || ws_bytes < (size_t)p.splits * M * N * sizeof(float)))
        return fail(VC_EWORKSPACE, "%s: workspace too small (need vc_gemm_workspace_bytes)", __func__);
    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.ws = ws;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.M = M; g.N = N; g.K = K;
    g.tiles_n = p.tiles_n; g.ntiles = p.tiles_m * p.tiles_n;
    g.kchunk = p.kchunk; g.splits = p.splits; g.flags = flags;
    auto al = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    const bool vec = al(A) && al(B) && (lda % 4 == 0) && (ldb % 4 == 0) && ((ta ? M : K) % 4 == 0) &&
                     ((tb ? K : N) % 4 == 0);
    if (p.big) {
        if (vec) dispatch_modes<Cfg128, true>(st, g, ta, tb); else dispatch_modes<Cfg128, false>(st, g, ta, tb);
    } else {
        if (vec) dispatch_modes<Cfg64, true>(st, g, ta, tb); else dispatch_modes<Cfg64, false>(st, g, ta, tb);
    }
    VC_LAUNCH_CHECK();
    if (p.splits > 1) {
        const long MN = (long)M * N;
        int blocks = cdiv(MN, 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, ws, p.splits, MN, N, C, ldc, bias, flags);
        VC_LAUNCH_CHECK();
    }
    return 0;
}
