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
    static constexpr int EPI_FLOATS = WM * WN * 32 * (TN * 32 + 4);  // epilogue_rows' per-wave transpose strips
    static constexpr int SMEM_FLOATS = A_FLOATS + B_FLOATS > EPI_FLOATS ? A_FLOATS + B_FLOATS : EPI_FLOATS;
    static constexpr int SMEM_BYTES = SMEM_FLOATS * 4;
};

// ---------------------------------------------------------------------------
// Operand loaders.  A thread owns NV "slots" of an operand tile; slot u always covers the same
// tile row and the same K offset, so everything that does not depend on the K-tile is computed
// ONCE in init() and kept in registers (u is a compile-time index after unrolling):
//   init(u, row, kofs)   row = global row of the slot, kofs = its K offset inside a tile
//   load(u, k0)          4 consecutive elements for the K-tile starting at k0 (zero outside)
// MK: operand stored K-contiguous (the 4 elements run along K): address p[row*ld + k]
// KM: operand stored row-contiguous (the 4 elements run along rows): address p[k*ld + row]
// VEC = pointer 16-B aligned, ld % 4 == 0, extent-along-vector % 4 == 0.
// ---------------------------------------------------------------------------
constexpr int MAXNV = 8;

template <bool VEC>
struct LoadMK {
    const float* p;
    long ld;
    int R, K;
    const float* q[MAXNV];
    int ko[MAXNV];
    __device__ __forceinline__ void init(int u, int row, int kofs) {
        q[u] = row < R ? p + (long)row * ld + kofs : nullptr;
        ko[u] = kofs;
    }
    __device__ __forceinline__ float4 load(int u, int k0) const {
        const int k = k0 + ko[u];
        if (!q[u] || k >= K) return f4zero();
        const float* s = q[u] + k0;
        if (VEC) return *reinterpret_cast<const float4*>(s);
        float4 v;
        v.x = s[0];
        v.y = (k + 1 < K) ? s[1] : 0.f;
        v.z = (k + 2 < K) ? s[2] : 0.f;
        v.w = (k + 3 < K) ? s[3] : 0.f;
        return v;
    }
};

template <bool VEC>
struct LoadKM {
    const float* p;
    long ld;
    int R, K;
    const float* q[MAXNV];
    int ko[MAXNV], rr[MAXNV];
    __device__ __forceinline__ void init(int u, int row, int kofs) {
        q[u] = row < R ? p + (long)kofs * ld + row : nullptr;
        ko[u] = kofs;
        rr[u] = row;
    }
    __device__ __forceinline__ float4 load(int u, int k0) const {
        if (!q[u] || k0 + ko[u] >= K) return f4zero();
        const float* s = q[u] + (long)k0 * ld;
        if (VEC) return *reinterpret_cast<const float4*>(s);
        const int r = rr[u];
        float4 v;
        v.x = s[0];
        v.y = (r + 1 < R) ? s[1] : 0.f;
        v.z = (r + 2 < R) ? s[2] : 0.f;
        v.w = (r + 3 < R) ? s[3] : 0.f;
        return v;
    }
};

// ---------------------------------------------------------------------------
// Staging helpers: thread -> (row, k) slots of one operand tile of ROWS x 32.
// ---------------------------------------------------------------------------
template <int ROWS, int NT, int MODE>
struct Stage {
    static constexpr int NV = (ROWS * 8 + NT - 1) / NT;  // float4 per thread per K-tile
    // thread counts that do not divide the tile (192 / 384 threads on a 64 / 128-row operand): the last
    // slot exists only for the first ROWS*8 - (NV-1)*NT threads
    static constexpr bool RAGGED = (ROWS * 8) % NT != 0;
    static __device__ __forceinline__ bool has(int f) { return !RAGGED || f < ROWS * 8; }
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
// ABL (ablation bits, debugging/benchmark only, see tools/microbench.py): 1 = no global loads after
// the prologue, 2 = no LDS stores / barriers, 4 = no LDS fragment reads.  ABL = 0 is the product path.
// Hook: called by every thread once per K-tile right after the tile became visible in LDS (As, Bs);
// lets a kernel fold an extra reduction over a staged operand into the loop (conv wgrad: bias gradient).
struct NoTileHook {
    __device__ __forceinline__ void operator()(const float*, const float*) const {}
};

template <class CFG, int AMODE, int BMODE, class ALoader, class BLoader, int ABL = 0, class Hook = NoTileHook>
__device__ __forceinline__ void mfma_mainloop(f32x16 (&acc)[CFG::TM][CFG::TN], ALoader A, BLoader B,
                                              int m0, int n0, int k_begin, int k_end, float* smem, Hook hook = Hook()) {
    constexpr int TM = CFG::TM, TN = CFG::TN, BM = CFG::BM, BN = CFG::BN, NT = CFG::NT;
    using SA = Stage<BM, NT, AMODE>;
    using SB = Stage<BN, NT, BMODE>;
    float* As = smem;
    float* Bs = smem + CFG::A_FLOATS;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / CFG::WN, wn = wave % CFG::WN;
    const int li = lane & 31, lh = lane >> 5;

    static_assert(SA::NV <= MAXNV && SB::NV <= MAXNV, "too many slots per thread");
    float4 ra[SA::NV], rb[SB::NV];
#pragma unroll
    for (int u = 0; u < SA::NV; ++u)
        if (SA::has(tid + u * NT)) A.init(u, m0 + SA::row(tid + u * NT), SA::kof(tid + u * NT));
#pragma unroll
    for (int u = 0; u < SB::NV; ++u)
        if (SB::has(tid + u * NT)) B.init(u, n0 + SB::row(tid + u * NT), SB::kof(tid + u * NT));
    auto gload = [&](int k0) {
#pragma unroll
        for (int u = 0; u < SA::NV; ++u)
            if (SA::has(tid + u * NT)) ra[u] = A.load(u, k0);
#pragma unroll
        for (int u = 0; u < SB::NV; ++u)
            if (SB::has(tid + u * NT)) rb[u] = B.load(u, k0);
    };
    if (k_begin < k_end) gload(k_begin);
    float a[TM][8], b[TN][8];
    if (ABL & 4) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) a[tm][s] = ra[0].x + s + tm;
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) b[tn][s] = rb[0].y + s - tn;
        }
    }
    for (int k0 = k_begin; k0 < k_end; k0 += 32) {
        if (!(ABL & 2) || k0 == k_begin) {
            __syncthreads();
#pragma unroll
            for (int u = 0; u < SA::NV; ++u)
                if (SA::has(tid + u * NT)) *reinterpret_cast<float4*>(&As[SA::lds(tid + u * NT)]) = ra[u];
#pragma unroll
            for (int u = 0; u < SB::NV; ++u)
                if (SB::has(tid + u * NT)) *reinterpret_cast<float4*>(&Bs[SB::lds(tid + u * NT)]) = rb[u];
            __syncthreads();
        }
        if (!(ABL & 1) && k0 + 32 < k_end) gload(k0 + 32);
        hook(As, Bs);
        if (ABL & 4) {
#pragma unroll
            for (int s = 0; s < 16; ++s)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][s & 7], b[tn][s & 7], acc[tm][tn], 0, 0, 0);
        } else {
            // Four chunks of 4 MFMA k-steps; the fragments of chunk c+1 are read from LDS while the 16 MFMAs of chunk c
            // run (two fragment sets, 32 VGPRs in all: the same as one 8-step set).
            float fa[2][TM][4], fb[2][TN][4];
            auto frag = [&](int c, int buf) {
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) {
                    const int r = (wm * TM + tm) * 32 + li;
                    if (AMODE == MODE_MK) {
                        const float4 v = *reinterpret_cast<const float4*>(&As[r * 36 + lh * 16 + c * 4]);
                        fa[buf][tm][0] = v.x; fa[buf][tm][1] = v.y; fa[buf][tm][2] = v.z; fa[buf][tm][3] = v.w;
                    } else {
#pragma unroll
                        for (int s = 0; s < 4; ++s) fa[buf][tm][s] = As[(lh * 16 + c * 4 + s) * (BM + 4) + r];
                    }
                }
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    const int r = (wn * TN + tn) * 32 + li;
                    if (BMODE == MODE_MK) {
                        const float4 v = *reinterpret_cast<const float4*>(&Bs[r * 36 + lh * 16 + c * 4]);
                        fb[buf][tn][0] = v.x; fb[buf][tn][1] = v.y; fb[buf][tn][2] = v.z; fb[buf][tn][3] = v.w;
                    } else {
#pragma unroll
                        for (int s = 0; s < 4; ++s) fb[buf][tn][s] = Bs[(lh * 16 + c * 4 + s) * (BN + 4) + r];
                    }
                }
            };
            frag(0, 0);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (c + 1 < 4) frag(c + 1, (c + 1) & 1);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn)
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c & 1][tm][s], fb[c & 1][tn][s], acc[tm][tn], 0, 0, 0);
            }
        }
    }
}

#ifdef VC_MICROBENCH  // measured and not adopted; kept for tools/microbench.py ablate only
// Double-buffered variant: ONE barrier per K-tile.  Tile t is consumed from LDS buffer t&1 while the
// global loads of tile t+1 are in flight; they are written to the other buffer after the MFMAs.
// (A wave can only reach the store of iteration t after every wave passed the barrier of iteration
// t-1, i.e. finished reading that buffer in iteration t-1.)  Needs 2 * CFG::SMEM_BYTES of LDS.
template <class CFG, int AMODE, int BMODE, class ALoader, class BLoader>
__device__ __forceinline__ void mfma_mainloop_db(f32x16 (&acc)[CFG::TM][CFG::TN], ALoader A, BLoader B,
                                                 int m0, int n0, int k_begin, int k_end, float* smem) {
    constexpr int TM = CFG::TM, TN = CFG::TN, BM = CFG::BM, BN = CFG::BN, NT = CFG::NT;
    constexpr int BUF = CFG::A_FLOATS + CFG::B_FLOATS;
    using SA = Stage<BM, NT, AMODE>;
    using SB = Stage<BN, NT, BMODE>;
    static_assert(!SA::RAGGED && !SB::RAGGED, "double-buffered variant: thread count must divide the tiles");
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / CFG::WN, wn = wave % CFG::WN;
    const int li = lane & 31, lh = lane >> 5;
    float4 ra[SA::NV], rb[SB::NV];
#pragma unroll
    for (int u = 0; u < SA::NV; ++u) A.init(u, m0 + SA::row(tid + u * NT), SA::kof(tid + u * NT));
#pragma unroll
    for (int u = 0; u < SB::NV; ++u) B.init(u, n0 + SB::row(tid + u * NT), SB::kof(tid + u * NT));
    auto gload = [&](int k0) {
#pragma unroll
        for (int u = 0; u < SA::NV; ++u) ra[u] = A.load(u, k0);
#pragma unroll
        for (int u = 0; u < SB::NV; ++u) rb[u] = B.load(u, k0);
    };
    auto sstore = [&](float* As, float* Bs) {
#pragma unroll
        for (int u = 0; u < SA::NV; ++u) *reinterpret_cast<float4*>(&As[SA::lds(tid + u * NT)]) = ra[u];
#pragma unroll
        for (int u = 0; u < SB::NV; ++u) *reinterpret_cast<float4*>(&Bs[SB::lds(tid + u * NT)]) = rb[u];
    };
    if (k_begin >= k_end) return;
    gload(k_begin);
    sstore(smem, smem + CFG::A_FLOATS);
    __syncthreads();
    int cur = 0;
    for (int k0 = k_begin; k0 < k_end; k0 += 32) {
        const bool more = k0 + 32 < k_end;
        if (more) gload(k0 + 32);
        const float* As = smem + cur * BUF;
        const float* Bs = As + CFG::A_FLOATS;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
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
        if (more) {
            float* An = smem + (cur ^ 1) * BUF;
            sstore(An, An + CFG::A_FLOATS);
        }
        __syncthreads();
        cur ^= 1;
    }
}

#endif  // VC_MICROBENCH

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

// ---------------------------------------------------------------------------
// Row-vectorised epilogue.  The MFMA accumulator layout gives a lane ONE column and 16 scattered
// rows, i.e. 4-byte global accesses in 128-B row segments.  Here every wave transposes its tiles
// through a private LDS strip so that a lane owns 4 CONSECUTIVE columns: 16-B global accesses,
// TN*128 contiguous bytes per row, 4x fewer memory instructions (cf. cdna_hip_programming.md T21).
// f(row, col, v): row / col relative to the workgroup tile, col % 4 == 0, v = 4 consecutive columns.
// Must be called by all threads (contains __syncthreads); smem is reused (>= WM*WN*32*(TN*32+4) floats).
// ---------------------------------------------------------------------------
struct NoStripHook {
    __device__ __forceinline__ void operator()(int, const float*, int) const {}
};
// g(tm, strip, LD): optional hook called by every lane after tile row-block tm of its wave has been written to the wave's LDS
// strip (32 rows x TN*32 columns, row pitch LD floats) and streamed out through f -- lets a kernel derive a second output from
// the whole 32-row block (e.g. a fused 2x2 max-pool).
template <class CFG, class F, class G = NoStripHook>
__device__ __forceinline__ void epilogue_rows(f32x16 (&acc)[CFG::TM][CFG::TN], float* smem, F f, G g = G()) {
    constexpr int TM = CFG::TM, TN = CFG::TN;
    constexpr int LD = TN * 32 + 4;        // floats per staged row (16-B aligned, rows 4 banks apart)
    constexpr int QPR = TN * 8;            // float4 per row
    constexpr int RPP = 64 / QPR;          // rows per pass
    static_assert(CFG::WM * CFG::WN * 32 * LD <= CFG::SMEM_FLOATS, "epilogue strip does not fit the tile LDS");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / CFG::WN, wn = wave % CFG::WN;
    const int li = lane & 31, lh = lane >> 5;
    float* T = smem + wave * 32 * LD;
    __syncthreads();  // every wave is done reading the operand tiles
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * lh) * LD + tn * 32 + li] = acc[tm][tn][r];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 32 / RPP; ++j) {
            const int row = j * RPP + lane / QPR, cq = lane % QPR;
            const float4 v = *reinterpret_cast<const float4*>(&T[row * LD + cq * 4]);
            f((wm * TM + tm) * 32 + row, wn * TN * 32 + cq * 4, v);
        }
        g(tm, T, LD);
        __syncthreads();
    }
}

// XCD-aware remap of a linear workgroup id (cdna_hip_programming.md T1, bijective form):
// consecutive logical ids land on the same XCD so neighbouring tiles share its L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

}  // namespace vc
