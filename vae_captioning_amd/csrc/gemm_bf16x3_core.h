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

// BX_ABL (timing only, `make bxabl`; never shipped): 1 no global loads after the prologue, 2 no split arithmetic, 4 no LDS writes,
// 8 no barriers, 16 no fragment reads, 32 no MFMAs
#ifndef BX_ABL
#define BX_ABL 0
#endif

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// (x, y) -> packed bf16 pairs hi = bf16(x, y), lo = bf16((x, y) - float(hi))
__device__ __forceinline__ void split_pair(float x, float y, unsigned& hi, unsigned& lo) {
    if (BX_ABL & 2) { hi = __float_as_uint(x); lo = __float_as_uint(y); return; }
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

// Branch-free operand fetch for ALIGNED operands (VEC): every slot loads unconditionally from an always-valid address (rows
// clamped once, at init) and the zero fill of out-of-range slots is applied where the value is consumed (put), from a mask.  The
// loaders of gemm_core.h return "zero or the loaded value" per slot, which makes hipcc branch around every load and wait
// vmcnt(0) at each merge -- the loads of the next K-tile were then waited for BEFORE the MFMAs of the current one
// (cdna_hip_programming.md, "three .s-level traps", item c): 0.50 -> 0.31 ms on 4096^3 in the no-loads ablation.
// The last, partial K-tile is SHIFTED BACK to [K - 32, K) (always in range; needs K >= 32) and the slots in front of the tile's
// real start -- k already accumulated -- are zero-filled like the rows outside the operand, so no address is ever clamped per k.
struct BxFetchMK {   // K-contiguous: slot = (row, four consecutive k at kofs)
    const float* base[MAXNV];   // p + clamped row * ld + kofs
    int ko[MAXNV];
    unsigned rowok;             // bit u: slot u's row is inside the operand
    __device__ __forceinline__ void init(const float* p, long ld, int R, int u, int row, int kofs) {
        rowok |= (row < R ? 1u : 0u) << u;
        base[u] = p + (long)min(row, R - 1) * ld + kofs;
        ko[u] = kofs;
    }
    // ks = start of the (possibly shifted) tile: a workgroup-uniform value
    __device__ __forceinline__ float4 fetch(int u, int ks, long) const { return *reinterpret_cast<const float4*>(base[u] + ks); }
    __device__ __forceinline__ unsigned ok(int u, int shift) const { return ((rowok >> u) & (ko[u] >= shift ? 1u : 0u)) << u; }
};
struct BxFetchKM {   // row-contiguous: slot = (k index kofs, four consecutive rows)
    const float* base[MAXNV];   // p + kofs * ld + clamped row
    int ko[MAXNV];
    unsigned rowok;
    __device__ __forceinline__ void init(const float* p, long ld, int R, int u, int row, int kofs) {
        rowok |= (row < R ? 1u : 0u) << u;
        base[u] = p + (long)kofs * ld + min(row, R - 4);
        ko[u] = kofs;
    }
    __device__ __forceinline__ float4 fetch(int u, int, long ks_ld) const { return *reinterpret_cast<const float4*>(base[u] + ks_ld); }
    __device__ __forceinline__ unsigned ok(int u, int shift) const { return ((rowok >> u) & (ko[u] >= shift ? 1u : 0u)) << u; }
};

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
// k quad = lane & 7, row quad = (lane >> 3) + 8 wave (four waves cover 128 rows, eight waves 256).  A load instruction touches
// eight k rows x 128 contiguous bytes.
template <int ROWS, int NT>
struct BxStageKM {
    static constexpr int NV = 4;
    static __device__ __forceinline__ bool has(int tid) { return ROWS * 2 >= NT || 4 * rq(tid) < ROWS; }   // (compile-time true when every thread owns a block)
    static __device__ __forceinline__ int kq(int tid) { return tid & 7; }
    static __device__ __forceinline__ int rq(int tid) { return ((tid & 63) >> 3) + 8 * (tid >> 6); }
};

// DB: two LDS images (2 x CFG::SMEM_BYTES): the next K-tile is written while slower waves still read the current one -- ONE barrier
// per K-tile, and a wave's split + write phase overlaps the other waves' MFMAs (the 256 x 256 tile runs one workgroup per CU: with
// a single image all eight waves leave the matrix pipe idle together).
template <class CFG, int AMODE, int BMODE, bool VEC, bool DB, class ALoader, class BLoader>
__device__ __forceinline__ void mfma_mainloop_bf16x3(f32x16 (&acc)[CFG::TM][CFG::TN], ALoader A, BLoader B,
                                                     int m0, int n0, int k_begin, int k_end, float* smem) {
    constexpr int TM = CFG::TM, TN = CFG::TN, BM = CFG::BM, BN = CFG::BN, NT = CFG::NT;
    static_assert(BM * BX_PITCH <= CFG::A_FLOATS * 4 && BN * BX_PITCH <= CFG::B_FLOATS * 4, "bf16x3 image must fit the f32 tile's LDS");
    static_assert((NT == 256 && BM <= 128 && BN <= 128) || (NT == 512 && BM <= 256 && BN <= 256), "row-contiguous stage map: one 4 x 4 block per thread");
    using MA = BxStageMK<BM, NT>;
    using KA = BxStageKM<BM, NT>;
    using MB = BxStageMK<BN, NT>;
    using KB = BxStageKM<BN, NT>;
    constexpr int NVA = AMODE == MODE_MK ? MA::NV : KA::NV;
    constexpr int NVB = BMODE == MODE_MK ? MB::NV : KB::NV;
    static_assert(NVA <= MAXNV && NVB <= MAXNV, "too many slots per thread");
    char* As = reinterpret_cast<char*>(smem);
    char* Bs = reinterpret_cast<char*>(smem + CFG::A_FLOATS);
    constexpr int BUF_BYTES = (CFG::A_FLOATS + CFG::B_FLOATS) * 4;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / CFG::WN, wn = wave % CFG::WN;
    const int li = lane & 31, lh = lane >> 5;

    float4 ra[NVA], rb[NVB];
    const bool hasA = AMODE == MODE_MK ? true : KA::has(tid);
    const bool hasB = BMODE == MODE_MK ? true : KB::has(tid);
    auto slotA = [&](int u) { return AMODE == MODE_MK ? MA::has(tid + u * NT) : hasA; };
    auto slotB = [&](int u) { return BMODE == MODE_MK ? MB::has(tid + u * NT) : hasB; };
    // aligned operands: branch-free fetchers (above); otherwise the element-wise guarded loaders of gemm_core.h
    typename std::conditional<AMODE == MODE_MK, BxFetchMK, BxFetchKM>::type FA;
    typename std::conditional<BMODE == MODE_MK, BxFetchMK, BxFetchKM>::type FB;
    FA.rowok = 0; FB.rowok = 0;
#pragma unroll
    for (int u = 0; u < NVA; ++u) {
        if (!slotA(u)) continue;
        const int row = AMODE == MODE_MK ? m0 + MA::row(tid + u * NT) : m0 + 4 * KA::rq(tid);
        const int kof = AMODE == MODE_MK ? MA::kq(tid + u * NT) * 4 : 4 * KA::kq(tid) + u;
        if (VEC) FA.init(A.p, A.ld, A.R, u, row, kof); else A.init(u, row, kof);
    }
#pragma unroll
    for (int u = 0; u < NVB; ++u) {
        if (!slotB(u)) continue;
        const int row = BMODE == MODE_MK ? n0 + MB::row(tid + u * NT) : n0 + 4 * KB::rq(tid);
        const int kof = BMODE == MODE_MK ? MB::kq(tid + u * NT) * 4 : 4 * KB::kq(tid) + u;
        if (VEC) FB.init(B.p, B.ld, B.R, u, row, kof); else B.init(u, row, kof);
    }
    const bool interior = VEC && m0 + BM <= A.R && n0 + BN <= B.R;   // workgroup-uniform
    unsigned okA = ~0u, okB = ~0u;   // edge tiles: bit u = slot u of the staged K-tile holds operand data (else zero fill)
    bool all_ok = true;              // uniform: the staged K-tile needs no zero fill
    auto gload = [&](int k0) {
        if (!VEC) {
#pragma unroll
            for (int u = 0; u < NVA; ++u) if (slotA(u)) ra[u] = A.load(u, k0);
#pragma unroll
            for (int u = 0; u < NVB; ++u) if (slotB(u)) rb[u] = B.load(u, k0);
            return;
        }
        // ONE code path for interior and edge tiles (two paths that load into the same registers make hipcc copy the loaded values
        // at the merge, i.e. wait for the loads right after issuing them): uniform tile start, zero fill only when needed
        const int ks = min(k0, A.K - 32), shift = k0 - ks;
        all_ok = interior && shift == 0;
        okA = 0; okB = 0;
#pragma unroll
        for (int u = 0; u < NVA; ++u) if (slotA(u)) { ra[u] = FA.fetch(u, ks, (long)ks * A.ld); okA |= FA.ok(u, shift); }
#pragma unroll
        for (int u = 0; u < NVB; ++u) if (slotB(u)) { rb[u] = FB.fetch(u, ks, (long)ks * B.ld); okB |= FB.ok(u, shift); }
    };
    // split + LDS write of one operand's staged registers (masked: zero fill of the slots whose ok bit is clear)
    auto put = [&](char* S, const float4* r0, unsigned ok, bool masked, auto mk_tag, auto MKS, auto KMS, bool has_km) {
        constexpr bool MK = decltype(mk_tag)::value;
        using SM = decltype(MKS);
        using SK = decltype(KMS);
        constexpr int NV = MK ? SM::NV : SK::NV;
        float4 r[NV];
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            r[u] = r0[u];
            if (masked && !((ok >> u) & 1u)) r[u] = f4zero();
        }
        if constexpr (MK) {
#pragma unroll
            for (int u = 0; u < SM::NV; ++u) {
                const int f = tid + u * NT;
                if (!SM::has(f)) continue;
                u32x2 hi, lo;
                split_quad(r[u].x, r[u].y, r[u].z, r[u].w, hi, lo);
                char* p = S + SM::row(f) * BX_PITCH + SM::kq(f) * 8;
                if (BX_ABL & 4) { asm volatile("" ::"v"(hi), "v"(lo)); continue; }
                *reinterpret_cast<u32x2*>(p) = hi;
                *reinterpret_cast<u32x2*>(p + 64) = lo;
            }
        } else {
            if (!has_km) return;
            char* p = S + 4 * SK::rq(tid) * BX_PITCH + SK::kq(tid) * 8;
            u32x2 hi, lo;
            auto wr = [&](char* q) {
                if (BX_ABL & 4) { asm volatile("" ::"v"(hi), "v"(lo)); return; }
                *reinterpret_cast<u32x2*>(q) = hi; *reinterpret_cast<u32x2*>(q + 64) = lo;
            };
            split_quad(r[0].x, r[1].x, r[2].x, r[3].x, hi, lo);
            wr(p);
            split_quad(r[0].y, r[1].y, r[2].y, r[3].y, hi, lo);
            wr(p + BX_PITCH);
            split_quad(r[0].z, r[1].z, r[2].z, r[3].z, hi, lo);
            wr(p + 2 * BX_PITCH);
            split_quad(r[0].w, r[1].w, r[2].w, r[3].w, hi, lo);
            wr(p + 3 * BX_PITCH);
        }
    };
    auto put_both = [&](bool masked, int buf) {
        put(As + buf * BUF_BYTES, ra, okA, masked, std::integral_constant<bool, AMODE == MODE_MK>(), MA(), KA(), hasA);
        put(Bs + buf * BUF_BYTES, rb, okB, masked, std::integral_constant<bool, BMODE == MODE_MK>(), MB(), KB(), hasB);
    };
    // Loads at the TOP of an iteration, their split + LDS write at its BOTTOM: no loaded value crosses the loop's back edge (hipcc
    // otherwise copies components of the first staged register at the back edge, behind an s_waitcnt on a load it has just issued).
    if (k_begin >= k_end) return;
    gload(k_begin);
    if (!VEC || all_ok) put_both(false, 0); else put_both(true, 0);
    if (!(BX_ABL & 8)) __syncthreads();
    int cur = 0;
    for (int k0 = k_begin; k0 < k_end; k0 += 32) {
        const bool more = k0 + 32 < k_end;
        if (!(BX_ABL & 1) && more) gload(k0 + 32);
        const char* Ac = As + cur * BUF_BYTES;
        const char* Bc = Bs + cur * BUF_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const char* p = Ac + ((wm * TM + tm) * 32 + li) * BX_PITCH + (kk * 2 + lh) * 16;
                if (BX_ABL & 16) {
                    const u32x4 c = {(unsigned)(tm + kk), (unsigned)lane, 0x3f803f80u, (unsigned)k0};
                    ah[tm] = __builtin_bit_cast(bf16x8, c); al[tm] = __builtin_bit_cast(bf16x8, c + 1u);
                    continue;
                }
                ah[tm] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p));
                al[tm] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p + 64));
            }
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const char* p = Bc + ((wn * TN + tn) * 32 + li) * BX_PITCH + (kk * 2 + lh) * 16;
                if (BX_ABL & 16) {
                    const u32x4 c = {(unsigned)(tn + kk), (unsigned)lane, 0x3f803f80u, (unsigned)k0};
                    bh[tn] = __builtin_bit_cast(bf16x8, c); bl[tn] = __builtin_bit_cast(bf16x8, c + 2u);
                    continue;
                }
                bh[tn] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p));
                bl[tn] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p + 64));
            }
            if (BX_ABL & 32) {
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) asm volatile("" ::"v"(ah[tm]), "v"(al[tm]));
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) asm volatile("" ::"v"(bh[tn]), "v"(bl[tn]));
                continue;
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
        if (more) {
            if (!DB && !(BX_ABL & 8)) __syncthreads();   // every wave has read this K-tile's fragments
            if (DB) cur ^= 1;                             // (the other image was last read one barrier ago)
            if (!VEC || all_ok) put_both(false, cur); else put_both(true, cur);
            if (!(BX_ABL & 8)) __syncthreads();           // the next K-tile is visible
        }
    }
}

}  // namespace vc
