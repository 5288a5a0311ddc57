// Split-bf16 ("bf16x3") main loop for the GEMM tile engine of gemm_core.h: f32 operands in HBM, f32 accumulators, the
// products on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16: 16x the f32 MFMA rate on gfx950, MI355X_MICROARCH.md).
//
//   a = a_hi + a_lo + O(2^-18 |a|),  a_hi = bf16(a) (round to nearest even), a_lo = bf16(a - a_hi)
//   a.b ~= a_hi.b_hi + a_hi.b_lo + a_lo.b_hi        (a_lo.b_lo ~ 2^-18 |a.b| is dropped)
//
// every bf16 x bf16 product is exact in f32 and the sums run in the f32 accumulator, so the result differs from the f32 MFMA
// path by ~1e-5 of sum|a.b| at worst, ~3e-6 typically (measured in tests/test_gpu_bf16x3.py) -- the same class as the
// F(4x4,3x3) Winograd rounding the convolution path is held to.  Three MFMAs of 16384 MACs in 3 x 32 cycles against sixteen
// f32 MFMAs of 2048 MACs in 16 x 64: 10.7x fewer matrix-pipe cycles per MAC.
//
// The split happens ONCE per staged element, between the global load and the LDS write (three VALU operations per element:
// v_cvt_pk_bf16_f32, widen + subtract, v_cvt_pk_bf16_f32), so every operand keeps its f32 HBM layout and the kernel is a
// drop-in for gemm_kernel: same arguments, same tile plan, same split-K workspace, same epilogue.
//
// LDS image of an operand tile of ROWS x 32 (k): row r at byte 144 r = [hi: 32 bf16 | lo: 32 bf16 | 16 B pad].  An MFMA
// fragment (lane l: row l & 31, k = 8 (l >> 5) .. + 7 of a 16-deep step) is ONE ds_read_b128; with a row pitch of nine 16-byte
// slots the sixteen lanes of a read group fall on sixteen different slots (conflict-free: the same argument as gemm_core.h's
// 36-float pitch).  The staging writes are 8-byte pieces (four k of one row, hi and lo): a float4 along K for a K-contiguous
// operand, one component of four float4 (an in-register 4 x 4 transpose) for a row-contiguous one; the lane -> (row, k quad)
// maps below keep the sixteen lanes of a ds_write_b64 group on 32 different banks.
#pragma once
#include <type_traits>
#include "gemm_core.h"

namespace vc {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// (x, y) -> packed bf16 pairs hi = bf16(x, y), lo = bf16((x, y) - float(hi))
__device__ __forceinline__ void split_pair(float x, float y, unsigned& hi, unsigned& lo) {
    const f32x2 v = {x, y};
    const bf16x2 h = __builtin_convertvector(v, bf16x2);
    const f32x2 r = v - __builtin_convertvector(h, f32x2);
    const bf16x2 l = __builtin_convertvector(r, bf16x2);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}
// four consecutive k of one row -> the 8-byte hi piece and the 8-byte lo piece
__device__ __forceinline__ void split_quad(float a, float b, float c, float d, u32x2& hi, u32x2& lo) {
    unsigned h0, l0, h1, l1;
    split_pair(a, b, h0, l0);
    split_pair(c, d, h1, l1);
    hi = u32x2{h0, h1};
    lo = u32x2{l0, l1};
}

constexpr int BX_PITCH = 144;  // bytes per LDS row: 64 hi + 64 lo + 16 pad

// K-contiguous operand (MK): slot f (one float4 = four k of one row) -> k quad f & 7, row from f >> 3 with its three low bits
// rotated so that the two rows inside a 16-lane write group are FOUR apart (4 x 144 B = 16 banks: the group covers 32 banks).
template <int ROWS, int NT>
struct BxStageMK {
    static constexpr int NV = (ROWS * 8 + NT - 1) / NT;
    static __device__ __forceinline__ bool has(int f) { return (ROWS * 8) % NT == 0 || f < ROWS * 8; }
    static __device__ __forceinline__ int kq(int f) { return f & 7; }
    static __device__ __forceinline__ int row(int f) {
        const int t = f >> 3;
        return (t & ~7) | ((t & 1) << 2) | ((t >> 1) & 3);
    }
};
// Row-contiguous operand (KM): a thread owns ONE block of four k x four rows (four float4 along the rows, k = 4 kq + j):
// k quad = lane & 7, row quad = (lane >> 3) + 8 wave.  A load instruction touches eight k rows x 128 contiguous bytes.
template <int ROWS, int NT>
struct BxStageKM {
    static constexpr int NV = 4;
    static __device__ __forceinline__ bool has(int tid) { return 4 * rq(tid) < ROWS; }
    static __device__ __forceinline__ int kq(int tid) { return tid & 7; }
    static __device__ __forceinline__ int rq(int tid) { return ((tid & 63) >> 3) + 8 * (tid >> 6); }
};

template <class CFG, int AMODE, int BMODE, class ALoader, class BLoader>
__device__ __forceinline__ void mfma_mainloop_bf16x3(f32x16 (&acc)[CFG::TM][CFG::TN], ALoader A, BLoader B,
                                                     int m0, int n0, int k_begin, int k_end, float* smem) {
    constexpr int TM = CFG::TM, TN = CFG::TN, BM = CFG::BM, BN = CFG::BN, NT = CFG::NT;
    static_assert(BM * BX_PITCH <= CFG::A_FLOATS * 4 && BN * BX_PITCH <= CFG::B_FLOATS * 4, "bf16x3 image must fit the f32 tile's LDS");
    static_assert(NT == 256, "stage maps assume four waves");
    using MA = BxStageMK<BM, NT>;
    using KA = BxStageKM<BM, NT>;
    using MB = BxStageMK<BN, NT>;
    using KB = BxStageKM<BN, NT>;
    constexpr int NVA = AMODE == MODE_MK ? MA::NV : KA::NV;
    constexpr int NVB = BMODE == MODE_MK ? MB::NV : KB::NV;
    static_assert(NVA <= MAXNV && NVB <= MAXNV, "too many slots per thread");
    char* As = reinterpret_cast<char*>(smem);
    char* Bs = reinterpret_cast<char*>(smem + CFG::A_FLOATS);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / CFG::WN, wn = wave % CFG::WN;
    const int li = lane & 31, lh = lane >> 5;

    float4 ra[NVA], rb[NVB];
    const bool hasA = AMODE == MODE_MK ? true : KA::has(tid);
    const bool hasB = BMODE == MODE_MK ? true : KB::has(tid);
#pragma unroll
    for (int u = 0; u < NVA; ++u) {
        if (AMODE == MODE_MK) { if (MA::has(tid + u * NT)) A.init(u, m0 + MA::row(tid + u * NT), MA::kq(tid + u * NT) * 4); }
        else if (hasA) A.init(u, m0 + 4 * KA::rq(tid), 4 * KA::kq(tid) + u);
    }
#pragma unroll
    for (int u = 0; u < NVB; ++u) {
        if (BMODE == MODE_MK) { if (MB::has(tid + u * NT)) B.init(u, n0 + MB::row(tid + u * NT), MB::kq(tid + u * NT) * 4); }
        else if (hasB) B.init(u, n0 + 4 * KB::rq(tid), 4 * KB::kq(tid) + u);
    }
    auto gload = [&](int k0) {
#pragma unroll
        for (int u = 0; u < NVA; ++u)
            if (AMODE == MODE_MK ? MA::has(tid + u * NT) : hasA) ra[u] = A.load(u, k0);
#pragma unroll
        for (int u = 0; u < NVB; ++u)
            if (BMODE == MODE_MK ? MB::has(tid + u * NT) : hasB) rb[u] = B.load(u, k0);
    };
    // split + LDS write of one operand's staged registers
    auto put = [&](char* S, const float4* r, auto mk_tag, auto MKS, auto KMS, bool has_km) {
        constexpr bool MK = decltype(mk_tag)::value;
        using SM = decltype(MKS);
        using SK = decltype(KMS);
        if constexpr (MK) {
#pragma unroll
            for (int u = 0; u < SM::NV; ++u) {
                const int f = tid + u * NT;
                if (!SM::has(f)) continue;
                u32x2 hi, lo;
                split_quad(r[u].x, r[u].y, r[u].z, r[u].w, hi, lo);
                char* p = S + SM::row(f) * BX_PITCH + SM::kq(f) * 8;
                *reinterpret_cast<u32x2*>(p) = hi;
                *reinterpret_cast<u32x2*>(p + 64) = lo;
            }
        } else {
            if (!has_km) return;
            char* p = S + 4 * SK::rq(tid) * BX_PITCH + SK::kq(tid) * 8;
            u32x2 hi, lo;
            split_quad(r[0].x, r[1].x, r[2].x, r[3].x, hi, lo);
            *reinterpret_cast<u32x2*>(p) = hi; *reinterpret_cast<u32x2*>(p + 64) = lo;
            split_quad(r[0].y, r[1].y, r[2].y, r[3].y, hi, lo);
            *reinterpret_cast<u32x2*>(p + BX_PITCH) = hi; *reinterpret_cast<u32x2*>(p + BX_PITCH + 64) = lo;
            split_quad(r[0].z, r[1].z, r[2].z, r[3].z, hi, lo);
            *reinterpret_cast<u32x2*>(p + 2 * BX_PITCH) = hi; *reinterpret_cast<u32x2*>(p + 2 * BX_PITCH + 64) = lo;
            split_quad(r[0].w, r[1].w, r[2].w, r[3].w, hi, lo);
            *reinterpret_cast<u32x2*>(p + 3 * BX_PITCH) = hi; *reinterpret_cast<u32x2*>(p + 3 * BX_PITCH + 64) = lo;
        }
    };
    if (k_begin < k_end) gload(k_begin);
    for (int k0 = k_begin; k0 < k_end; k0 += 32) {
        __syncthreads();
        put(As, ra, std::integral_constant<bool, AMODE == MODE_MK>(), MA(), KA(), hasA);
        put(Bs, rb, std::integral_constant<bool, BMODE == MODE_MK>(), MB(), KB(), hasB);
        __syncthreads();
        if (k0 + 32 < k_end) gload(k0 + 32);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const char* p = As + ((wm * TM + tm) * 32 + li) * BX_PITCH + (kk * 2 + lh) * 16;
                ah[tm] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p));
                al[tm] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p + 64));
            }
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const char* p = Bs + ((wn * TN + tn) * 32 + li) * BX_PITCH + (kk * 2 + lh) * 16;
                bh[tn] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p));
                bl[tn] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p + 64));
            }
            // the two small terms first, then hi.hi: three independent accumulators between two uses of the same one
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[tm], bh[tn], acc[tm][tn], 0, 0, 0);
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bl[tn], acc[tm][tn], 0, 0, 0);
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bh[tn], acc[tm][tn], 0, 0, 0);
        }
    }
}

}  // namespace vc
