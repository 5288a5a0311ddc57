// f32-in / f32-accumulate MFMA tile engine for gfx950 (v_mfma_f32_32x32x2_f32).
//
// One workgroup = WM x WN waves (64 lanes each); each wave owns TM x TN MFMA tiles of
// 32x32, so the block tile is BM = WM*TM*32 by BN = WN*TN*32, K-step BK = 32.
//
// MFMA 32x32x2 f32 operand map (cdna_hip_programming.md section 3): lane l supplies
// A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31]; C/D: col = l&31,
// row = (reg&3) + 8*(reg>>2) + 4*(l>>5).  The contraction order inside a K-step is
// free as long as A and B agree, so lane-half h = l>>5 takes k in [16h, 16h+16): each
// lane then reads 16 CONTIGUOUS k values of its row -> ds_read_b128 for operands
// staged K-contiguous.
//
// LDS images (single buffer; the next K-tile is prefetched into registers while the
// current one is consumed -- 2 barriers per K-tile, 3-4 workgroups per CU hide them):
//   MK image: S[row][36]       for operands whose global layout is K-contiguous
//             (row stride 36 dwords: the 16 rows of a ds_read_b128 lane group tile
//              all 64 banks, conflict-free; 144 B keeps 16-B alignment)
//   KM image: S[k][R + 4]      for operands whose global layout is row-contiguous
//             (fragment reads are ds_read_b32, 32 consecutive dwords per half-wave)
// LDS bandwidth is irrelevant here: one 64-cycle MFMA consumes 2 dwords per lane.
#pragma once
#include "common.h"

namespace vc {

enum OperandMode { MODE_MK = 0, MODE_KM = 1 };

template <int WM_, int WN_, int TM_, int TN_>
struct TileCfg {
    static constexpr int WM = WM_, WN = WN_, TM = TM_, TN = TN_;
    static constexpr int BM = WM * TM * 32, BN = WN * TN * 32, BK = 32;
    static constexpr int NT = WM * WN * 64;
    static constexpr int A_FLOATS = BM * 36;  // >= 32*(BM+4)
    static constexpr int B_FLOATS = BN * 36;
    static constexpr int SMEM_BYTES = (A_FLOATS + B_FLOATS) * 4;
};

// ---------------------------------------------------------------------------
// Plain global-memory operand loaders.  load(r, k): 4 consecutive elements
//   MK: rows r, K-contiguous:   elements (r, k..k+3)      address p[r*ld + k]
//   KM: rows k, R-contiguous:   elements (r..r+3, k)      address p[k*ld + r]
// VEC = pointer 16-B aligned, ld % 4 == 0, extent-along-vector % 4 == 0.
// ---------------------------------------------------------------------------
template <bool VEC>
struct LoadMK {
    const float* p;
    long ld;
    int R, K;
    __device__ __forceinline__ float4 load(int r, int k) const {
        if (r >= R) return f4zero();
        const float* q = p + (long)r * ld + k;
        if (VEC) {
            if (k >= K) return f4zero();
            return *reinterpret_cast<const float4*>(q);
        }
        float4 v;
        v.x = (k + 0 < K) ? q[0] : 0.f;
        v.y = (k + 1 < K) ? q[1] : 0.f;
        v.z = (k + 2 < K) ? q[2] : 0.f;
        v.w = (k + 3 < K) ? q[3] : 0.f;
        return v;
    }
};

template <bool VEC>
struct LoadKM {
    const float* p;
    long ld;
    int R, K;
    __device__ __forceinline__ float4 load(int r, int k) const {
        if (k >= K) return f4zero();
        const float* q = p + (long)k * ld + r;
        if (VEC) {
            if (r >= R) return f4zero();
            return *reinterpret_cast<const float4*>(q);
        }
        float4 v;
        v.x = (r + 0 < R) ? q[0] : 0.f;
        v.y = (r + 1 < R) ? q[1] : 0.f;
        v.z = (r + 2 < R) ? q[2] : 0.f;
        v.w = (r + 3 < R) ? q[3] : 0.f;
        return v;
    }
};

// ---------------------------------------------------------------------------
// Staging helpers: thread -> (row, k) slots of one operand tile of ROWS x 32.
// ---------------------------------------------------------------------------
template <int ROWS, int NT, int MODE>
struct Stage {
    static constexpr int NV = ROWS * 8 / NT;  // float4 per thread per K-tile
    static_assert(ROWS * 8 % NT == 0, "tile/threads mismatch");
    // MK: slot f -> row f/8, k-quad f%8.   KM: slot f -> k-row f/(ROWS/4), row-quad f%(ROWS/4)
    static __device__ __forceinline__ int row(int f) { return MODE == MODE_MK ? (f >> 3) : ((f % (ROWS / 4)) * 4); }
    static __device__ __forceinline__ int kof(int f) { return MODE == MODE_MK ? ((f & 7) * 4) : (f / (ROWS / 4)); }
    static __device__ __forceinline__ int lds(int f) {
        return MODE == MODE_MK ? (row(f) * 36 + kof(f)) : (kof(f) * (ROWS + 4) + row(f));
    }
};

// ---------------------------------------------------------------------------
// The main loop.  acc[TM][TN] accumulates  sum_k A(m0+.., k) * B(n0+.., k)  for
// k in [k_begin, k_end)  (k_begin % 32 == 0).
// ---------------------------------------------------------------------------
template <class CFG, int AMODE, int BMODE, class ALoader, class BLoader>
__device__ __forceinline__ void mfma_mainloop(f32x16 (&acc)[CFG::TM][CFG::TN], const ALoader& A, const BLoader& B,
                                              int m0, int n0, int k_begin, int k_end, float* smem) {
    constexpr int TM = CFG::TM, TN = CFG::TN, BM = CFG::BM, BN = CFG::BN, NT = CFG::NT;
    using SA = Stage<BM, NT, AMODE>;
    using SB = Stage<BN, NT, BMODE>;
    float* As = smem;
    float* Bs = smem + CFG::A_FLOATS;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / CFG::WN, wn = wave % CFG::WN;
    const int li = lane & 31, lh = lane >> 5;

    float4 ra[SA::NV], rb[SB::NV];
    auto gload = [&](int k0) {
#pragma unroll
        for (int u = 0; u < SA::NV; ++u) {
            const int f = tid + u * NT;
            ra[u] = A.load(m0 + SA::row(f), k0 + SA::kof(f));
        }
#pragma unroll
        for (int u = 0; u < SB::NV; ++u) {
            const int f = tid + u * NT;
            rb[u] = B.load(n0 + SB::row(f), k0 + SB::kof(f));
        }
    };
    if (k_begin < k_end) gload(k_begin);
    for (int k0 = k_begin; k0 < k_end; k0 += 32) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < SA::NV; ++u) *reinterpret_cast<float4*>(&As[SA::lds(tid + u * NT)]) = ra[u];
#pragma unroll
        for (int u = 0; u < SB::NV; ++u) *reinterpret_cast<float4*>(&Bs[SB::lds(tid + u * NT)]) = rb[u];
        __syncthreads();
        if (k0 + 32 < k_end) gload(k0 + 32);
#pragma unroll
        for (int c = 0; c < 2; ++c) {  // two chunks of 8 MFMA k-steps
            float a[TM][8], b[TN][8];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const int r = (wm * TM + tm) * 32 + li;
                if (AMODE == MODE_MK) {
                    const float4 v0 = *reinterpret_cast<const float4*>(&As[r * 36 + lh * 16 + c * 8]);
                    const float4 v1 = *reinterpret_cast<const float4*>(&As[r * 36 + lh * 16 + c * 8 + 4]);
                    a[tm][0] = v0.x; a[tm][1] = v0.y; a[tm][2] = v0.z; a[tm][3] = v0.w;
                    a[tm][4] = v1.x; a[tm][5] = v1.y; a[tm][6] = v1.z; a[tm][7] = v1.w;
                } else {
#pragma unroll
                    for (int s = 0; s < 8; ++s) a[tm][s] = As[(lh * 16 + c * 8 + s) * (BM + 4) + r];
                }
            }
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const int r = (wn * TN + tn) * 32 + li;
                if (BMODE == MODE_MK) {
                    const float4 v0 = *reinterpret_cast<const float4*>(&Bs[r * 36 + lh * 16 + c * 8]);
                    const float4 v1 = *reinterpret_cast<const float4*>(&Bs[r * 36 + lh * 16 + c * 8 + 4]);
                    b[tn][0] = v0.x; b[tn][1] = v0.y; b[tn][2] = v0.z; b[tn][3] = v0.w;
                    b[tn][4] = v1.x; b[tn][5] = v1.y; b[tn][6] = v1.z; b[tn][7] = v1.w;
                } else {
#pragma unroll
                    for (int s = 0; s < 8; ++s) b[tn][s] = Bs[(lh * 16 + c * 8 + s) * (BN + 4) + r];
                }
            }
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][s], b[tn][s], acc[tm][tn], 0, 0, 0);
        }
    }
}

template <class CFG>
__device__ __forceinline__ void acc_zero(f32x16 (&acc)[CFG::TM][CFG::TN]) {
#pragma unroll
    for (int tm = 0; tm < CFG::TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < CFG::TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;
}

// Coordinates of accumulator element (tm, tn, reg) of the calling lane inside the block tile.
template <class CFG>
struct AccCoord {
    int wm, wn, li, lh;
    __device__ __forceinline__ AccCoord() {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        wm = wave / CFG::WN;
        wn = wave % CFG::WN;
        li = lane & 31;
        lh = lane >> 5;
    }
    __device__ __forceinline__ int row(int tm, int reg) const {
        return (wm * CFG::TM + tm) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
    }
    __device__ __forceinline__ int col(int tn) const { return (wn * CFG::TN + tn) * 32 + li; }
};

// XCD-aware remap of a linear workgroup id (cdna_hip_programming.md T1, bijective form):
// consecutive logical ids land on the same XCD so neighbouring tiles share its L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

}  // namespace vc
