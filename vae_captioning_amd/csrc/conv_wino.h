// Shared by conv_wino.hip (forward / data gradient) and conv_wino_wgrad.hip (weight gradient): tile-block geometry and the buffer load.
#pragma once
#include <stdlib.h>
#include "gemm_core.h"
#include "vaecap.h"

namespace vc {

typedef unsigned int wu32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned WOOB = 0x80000000u;   // buffer voffset that always fails the bounds check (tensors are < 2 GiB): loads 0

struct WinoGeom {
    int B, H, W, C, N;
    int TBH, TBW;          // tiles per block (rows, columns); TBH * TBW <= 32
    int PW, PH;            // halo patch of a block in pixels: 2 TBW + 2, 2 TBH + 2
    int bx_n, by_n;        // blocks per image row / column
    int blocks_img;
    int nblocks;           // B * blocks_img
    unsigned m_blocks_img, m_bx_n, m_pw, m_tbw;   // ceil(2^32 / d) of the kernel's divisors (wino_magic): n / d == __umulhi(n, m) while n * d < 2^32
    int P;                 // forward / data gradient only: pitch of a patch row in the LDS, in pixel PAIRS (>= PW / 2; chosen bank-conflict-free)
};

static inline unsigned wino_magic(int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ULL + (unsigned)d - 1) / (unsigned)d); }   // 0: divisor 1
__device__ __forceinline__ unsigned wino_div(unsigned n, unsigned m) { return m ? __umulhi(n, m) : n; }

__device__ __forceinline__ float4 wbufload(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    wu32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
    return *reinterpret_cast<float4*>(&v);
}

// fraction of the F(2x2,3x3) kernel's tile slots that hold real tiles on an H x W image (its blocks: at most sixteen 2 x 2 tiles whose
// halo patch fits 100 pixels -- the search of plan_wino2 in conv_wino.hip)
static inline double wino2_coverage(int H, int W) {
    const int TW = W / 2, TH = H / 2;
    double best = 0.0;
    for (int tbw = 1; tbw <= 16 && tbw <= TW; ++tbw) {
        int tbh = 16 / tbw;
        if (tbh > TH) tbh = TH;
        if (tbh < 1 || (2 * tbh + 2) * (2 * tbw + 2) > 100) continue;
        const double eff = (double)TW * TH / ((double)((TW + tbw - 1) / tbw) * ((TH + tbh - 1) / tbh) * 16.0);
        if (eff > best) best = eff;
    }
    return best;
}

static inline bool waligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// The kernels address their tensors with 32-bit buffer offsets (< 2 GiB per launch): a call on more images is cut into launches
// over image ranges.  Images per launch for [B, H, W, max(C, N)] floats (VC_WINO_MAX_BYTES: tests force the cut on small shapes).
static inline int wino_images_per_launch(int B, int H, int W, int C, int N) {
    static const long cap = getenv("VC_WINO_MAX_BYTES") ? atol(getenv("VC_WINO_MAX_BYTES")) : 0x7fffffffL;
    const long per = (long)H * W * (long)(C > N ? C : N) * 4;
    long n = per > 0 ? cap / per : 0;
    if (n > B) n = B;
    return (int)n;   // 0: one image alone is too large
}

}  // namespace vc
