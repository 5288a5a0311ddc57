// 3x3 / stride 1 / SAME convolution by Winograd's minimal filtering F(2x2, 3x3) for gfx950: forward and data gradient of
// utils/image_embeddings.py:36-212 in fp32 with 2.25x fewer multiplications than the direct form (conv_patch.hip).
//
//   Y = A^T [ sum_c (G g_c G^T) (.) (B^T d_c B) ] A          per 2 x 2 output tile: d = its 4 x 4 input patch, g = the 3 x 3 filter
//
// The sixteen positions of the transformed domain are sixteen independent products M_p[n][tile] = sum_c V_p[c][n] U_p[c][tile];
// everything between the NHWC input and the NHWC output stays inside the kernel:
//   * a wave owns a BLOCK of up to 32 tiles (TBH x TBW tiles = 2 TBH x 2 TBW output pixels of one image) and all sixteen positions
//     for 32 output columns: sixteen 32 x 32 MFMA accumulators = 256 registers (AGPRs), one wave per SIMD;
//   * the workgroup (four waves = four blocks x the same 32 columns) stages, per 16-channel chunk, the halo patches of its blocks
//     and the chunk's transformed weights (32 KB, contiguous in the packed layout) in LDS;
//   * the input transform B^T d B runs in registers on the MFMA's B-operand layout (lane = tile, lane half = 8 of the 16 channels):
//     eight ds_read_b128 + 32 float4 additions per sixteen MFMAs, prepared one position row ahead of the MFMAs that consume it;
//   * the weights are transformed once per optimiser step (vc_conv3x3_wino_pack_f32; transpose 1 = flipped taps, transposed
//     channels for the data gradient);
//   * a lane ends with all sixteen positions of ONE tile x 16 output columns, so the output transform A^T M A, the bias, the ReLU /
//     ReLU mask, the 16-byte stores and the fused 2 x 2 max-pool (a pooling window IS a Winograd tile) are register math.
// Rounding: the transforms add at most four fp32 terms with coefficients 1, 1/2, 1/4; results agree with the direct form to a few
// 1e-7 of the tensor maximum times sqrt(K) (tests/test_gpu_conv_wino.py holds both to the same fp64 oracle).
#include <stdlib.h>
#include "conv_wino.h"

namespace vc {


enum { WK_FWD = 0, WK_DGRAD = 1 };
constexpr int WCH = 16;                          // channels per chunk
constexpr int WPITCH = WCH + 4;                  // floats per patch pixel in LDS
constexpr int WPIX = 180;                        // patch pixels per block: (2 TBH + 2)(2 TBW + 2) <= 180
constexpr int WBLK = WPIX * WPITCH;              // floats per block patch
constexpr int WSLOTS = 6;                        // float4 patch slots per thread and half: 4 blocks x 180 pixels x 2 channel quads <= 256 x 6
constexpr int WV_FLOATS = 2 * 16 * 2 * 32 * 4;   // one chunk of transformed weights: [half 2][p 16][lane half 2][n 32][e 4]
constexpr int WP_OFF = WV_FLOATS;                // LDS: weights first (their ds_read offsets stay below the 64 KB immediate range), then the patches
constexpr int WINO_LDS_BYTES = (WP_OFF + 4 * WBLK) * 4;


struct WinoArgs {
    WinoGeom g;
    const float* x;     // [P, C]
    const float* wp;    // packed [N/32][C/16][2][16][2][32][4]
    float* out;         // [P, N]
    const float* aux;   // fwd: bias [N] or null; dgrad: ReLU source [P, N] or null
    float* pool;        // fwd: also max_pool2x2(out) [B, H/2, W/2, N] (null: none)
    unsigned* mask;     // [ntiles][256 threads][2]: (out > 0) of each lane's 2 x 2 pixels x 16 columns as 64 bits -- written by the forward
                        // (null: not wanted), read by the data gradient of the NEXT layer instead of relu_src (same shape => same tiles / lanes)
    int relu;
    int tiles_n, ntiles, nchunks;
};

__device__ __forceinline__ float wcomp(const float4& v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; }
__device__ __forceinline__ float4 f4add(const float4& a, const float4& b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4sub(const float4& a, const float4& b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

// tools/probes/wino_trace.hip compiles this file with VC_WINO_TRACE: every wave stamps the cycle counter at phase edges
#ifdef VC_WINO_TRACE
__device__ unsigned long long* g_wino_trace = nullptr;  // [workgroups][4 waves][8 stamps]
#define WINO_STAMP(k)                                                                                              \
    do {                                                                                                           \
        if (g_wino_trace && (threadIdx.x & 63) == 0)                                                               \
            g_wino_trace[((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (k)] = __builtin_readcyclecounter();    \
    } while (0)
#else
#define WINO_STAMP(k)
#endif

template <int KIND, bool POOL>
__global__ __launch_bounds__(256, 1) void conv_wino_kernel(WinoArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const WinoGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    WINO_STAMP(0);
    const int id = xcd_remap(blockIdx.x, a.ntiles);
    const int tm = id / a.tiles_n, nt = id - tm * a.tiles_n, n0 = nt * 32;
    const int C = g.C, N = g.N;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)((long)g.B * g.H * g.W * C * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.wp, 0, 16 * C * N * 4, 0x00020000);

    // A chunk (16 channels) is staged as two HALVES: half q = channels [8 q, 8 q + 8) of every patch pixel + the weights of those
    // channels (lane half lh works on channel quad 2 q + lh).  While the MFMAs of one half run, the other half's region of the LDS is
    // refilled for the next half-phase, so that LDS writes, global loads and MFMAs overlap all the time; one barrier per half-phase.
    // patch slots of this thread: slot s = tid + 256 i = (patch pixel s / 2 of the workgroup's 4 x 180, quad 2 q + (s & 1))
    unsigned voff[WSLOTS];
#pragma unroll
    for (int i = 0; i < WSLOTS; ++i) {
        const unsigned pl = (unsigned)(tid >> 1) + 128u * i;
        const unsigned blk = pl / WPIX, pix = pl - blk * WPIX;   // (32-bit unsigned divisions: a signed or 64-bit one costs ~100 VALU)
        const unsigned gb = (unsigned)tm * 4u + blk;
        // (branch-free: the reciprocal divisions are cheap, and divergent branches here cost more than the arithmetic they skip)
        const unsigned gbc = gb < (unsigned)g.nblocks ? gb : 0u;
        const unsigned b = wino_div(gbc, g.m_blocks_img), rem = gbc - b * (unsigned)g.blocks_img;
        const unsigned by = wino_div(rem, g.m_bx_n), bx = rem - by * (unsigned)g.bx_n;
        const unsigned py = wino_div(pix, g.m_pw), px = pix - py * (unsigned)g.PW;
        const int y = (int)(by * 2u * g.TBH + py) - 1, x = (int)(bx * 2u * g.TBW + px) - 1;
        const bool ok = blk < 4 && gb < (unsigned)g.nblocks && pix < (unsigned)(g.PH * g.PW) && (unsigned)y < (unsigned)g.H && (unsigned)x < (unsigned)g.W;
        const unsigned off = (((b * (unsigned)g.H + (unsigned)y) * (unsigned)g.W + (unsigned)x) * (unsigned)C + (((unsigned)tid ^ (py >> 1)) & 1u) * 4u) * 4u;
        voff[i] = ok ? off : WOOB;
    }
    // LDS patch layout: pixel pitch 20 floats = four channel quads + pad; in patch rows py with (py >> 1) odd the two quads of a half
    // are SWAPPED (done on the global side: the staging thread fetches the other quad, its LDS address stays uniform) -- this shifts
    // every second tile row by four banks, so that the sixteen lanes of a ds_read_b128 group (two or four tile rows of 8 / 4 tiles)
    // hit sixteen different bank quads (without it: 2-way conflicts, SQ_LDS_BANK_CONFLICT > SQ_ACTIVE_INST_LDS)
    const int pst = WP_OFF + (tid >> 1) * WPITCH + (tid & 1) * 4;   // LDS float index of slot 0 of half 0; slot i is 128 pixels further, half 1 eight floats
    const unsigned vsrc = (unsigned)(((long)nt * a.nchunks) * WV_FLOATS * 4) + (unsigned)tid * 16u;  // weights: half-phase h at + h * 16 KB, piece i at + i * 4 KB

    // this lane's tile inside its wave's block, its 4 x 4 patch origin in LDS, its weight fragment origin
    const int ntl = g.TBH * g.TBW;
    const int jt = li < ntl ? li : 0;
    const int tyl = (int)wino_div((unsigned)jt, g.m_tbw), txl = jt - tyl * g.TBW;
    const int abase0 = WP_OFF + wave * WBLK + ((2 * tyl) * g.PW + 2 * txl) * WPITCH;           // + (i * PW + j) * WPITCH + slot * 4
    int aq[2][2];   // [half q][row pair i >> 1]: abase0 + 4 * slot of this lane's quad 2 q + lh in patch rows 2 tyl + i
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) aq[q][pr] = abase0 + (2 * q + (lh ^ (tyl & 1) ^ pr)) * 4;
    const int rowp = g.PW * WPITCH;
    const int vbase = (lh * 32 + li) * 4;                                                      // + q * 4096 + p * 256

    const int gb = tm * 4 + wave;
    const bool blk_ok = gb < g.nblocks && li < ntl;
    const int gbc = gb < g.nblocks ? gb : 0;
    const int b = (int)wino_div((unsigned)gbc, g.m_blocks_img), rem = gbc - b * g.blocks_img;
    const int by = (int)wino_div((unsigned)rem, g.m_bx_n), bx = rem - by * g.bx_n;
    const int y0 = (by * g.TBH + tyl) * 2, x0 = (bx * g.TBW + txl) * 2;
    const bool ok00 = blk_ok && y0 < g.H && x0 < g.W, ok01 = ok00 && x0 + 1 < g.W, ok10 = ok00 && y0 + 1 < g.H, ok11 = ok10 && x0 + 1 < g.W;
    const long p00 = ((long)(b * g.H + y0) * g.W + x0) * N;
    const long rowN = (long)g.W * N;
    unsigned mbits[2] = {0xffffffffu, 0xffffffffu};
    f32x16 acc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[5][r] = 0.f;   // (the other fifteen start with a zero-addend MFMA, see mf)
    // forward: the bias rides in the accumulator of position (1, 1) -- A^T's column 1 is (1, 1), so A^T M A adds M_(1,1) to all four
    // outputs of the tile; its loads overlap the first patch loads and the epilogue has no load left
    if (KIND == WK_FWD && a.aux) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const float4 bv = *reinterpret_cast<const float4*>(a.aux + n0 + 8 * rg + 4 * lh);
            acc[5][4 * rg] = bv.x; acc[5][4 * rg + 1] = bv.y; acc[5][4 * rg + 2] = bv.z; acc[5][4 * rg + 3] = bv.w;
        }
    }

    float4 st[WSLOTS + 4];           // staging registers: six patch slots + four weight pieces of ONE half
    auto gload1 = [&](int hp, int i) {   // the i-th of the ten global loads of half-phase hp's data
        if (i < WSLOTS) st[i] = wbufload(rx, voff[i], (unsigned)hp * 32u);
        else st[i] = wbufload(rw, vsrc + (unsigned)(i - WSLOTS) * 4096u, (unsigned)hp * 16384u);
    };
    auto lstore = [&](int q, int i) {    // the i-th of the ten LDS writes of a half
        if (i < WSLOTS) {
            if (i < WSLOTS - 1 || tid + 256 * i < 2 * 4 * WPIX) *reinterpret_cast<float4*>(&smem[pst + q * 8 + i * 128 * WPITCH]) = st[i];
        } else {
            *reinterpret_cast<float4*>(&smem[q * 4096 + (tid + 256 * (i - WSLOTS)) * 4]) = st[i];
        }
    };

    // unit (q, xi): position row xi of half q.  U[xi][nu] = (B^T d B)[xi][nu], B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1].
    // Units run in the order xi = 0, 2, 1, 3 so that every patch row is read once per half: rows 0 and 2 for xi = 0, row 1 for
    // xi = 2, nothing for xi = 1, row 3 for xi = 3 (sixteen patch reads + sixteen weight reads per 64 MFMAs).
    // One wave per SIMD: nothing else feeds the matrix pipe while this wave issues anything that is not an MFMA, so the software
    // pipeline is written out gap by gap and pinned with sched_barriers -- consecutive MFMAs go to DIFFERENT accumulators (k-step
    // outer, column inner), every gap between two MFMAs carries at most five other instructions.  tools/probes/mfma_fillers.hip
    // measures what a lone wave hides beside this MFMA: LDS reads are free, every VALU instruction in a gap costs matrix-pipe time
    // (0 / 1 / 2 / 4 adds per gap: 135 / 125 / 118 / 115 TFLOP/s), a packed v_pk_add_f32 costs as much as a scalar add -- so the
    // float4 additions are left to the SLP vectoriser (two packed adds per float4).
    float4 ur[2][4], vf[2][4];
    float4 dr[4][4], tt[4];
    auto xi_of = [](int u4) { return u4 == 1 ? 2 : u4 == 2 ? 1 : u4; };
    auto rdp = [&](int q, int u4, int j) {   // pixel j of the patch rows unit u4 is the first to need
        const int xi = xi_of(u4);
        const float* p0 = &smem[aq[q][0] + j * WPITCH];   // patch rows 0, 1
        const float* p1 = &smem[aq[q][1] + j * WPITCH];   // patch rows 2, 3
        if (xi == 0) { dr[0][j] = *reinterpret_cast<const float4*>(p0); dr[2][j] = *reinterpret_cast<const float4*>(p1 + 2 * rowp); }
        if (xi == 2) dr[1][j] = *reinterpret_cast<const float4*>(p0 + rowp);
        if (xi == 3) dr[3][j] = *reinterpret_cast<const float4*>(p1 + 3 * rowp);
    };
    auto rdv = [&](int q, int u4, int buf, int nu) { vf[buf][nu] = *reinterpret_cast<const float4*>(&smem[vbase + q * 4096 + (xi_of(u4) * 4 + nu) * 256]); };
    auto tstep = [&](int u4, int buf, int k) {   // eight steps of four additions
        const int xi = xi_of(u4);
        if (k < 4) tt[k] = xi == 0 ? f4sub(dr[0][k], dr[2][k]) : xi == 1 ? f4add(dr[1][k], dr[2][k]) : xi == 2 ? f4sub(dr[2][k], dr[1][k]) : f4sub(dr[1][k], dr[3][k]);
        if (k == 4) ur[buf][0] = f4sub(tt[0], tt[2]);
        if (k == 5) ur[buf][1] = f4add(tt[1], tt[2]);
        if (k == 6) ur[buf][2] = f4sub(tt[2], tt[1]);
        if (k == 7) ur[buf][3] = f4sub(tt[1], tt[3]);
    };
    auto mf = [&](int u4, int buf, int m, bool first) {
        const int xi = xi_of(u4), e = m >> 2, nu = m & 3;
        // the first MFMA of an accumulator (first half-phase of the tile, k-step 0) takes the constant 0 as its addend: no zero-fill of 240
        // accumulator registers per tile (position (1, 1) starts at the bias instead)
        if (first && e == 0 && xi * 4 + nu != 5) {
            f32x16 z;
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = 0.f;
            acc[xi * 4 + nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcomp(vf[buf][nu], e), wcomp(ur[buf][nu], e), z, 0, 0, 0);
        } else {
            acc[xi * 4 + nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcomp(vf[buf][nu], e), wcomp(ur[buf][nu], e), acc[xi * 4 + nu], 0, 0, 0);
        }
    };
#define WSB() __builtin_amdgcn_sched_barrier(0)
    // one half-phase = the four units of half q of the current chunk; the data of the NEXT half-phase (the other half's LDS region) sits
    // in the staging registers since the previous half-phase's unit 3 (three units = ~3000 cycles of load latency) and is written to
    // the LDS during unit 2; the barrier at the head of unit 3 closes both this half's last reads and those writes, so unit 3 already
    // prepares the first unit of the next half-phase and re-loads the staging registers.
    auto half = [&](int q, bool more, bool more2, int nexthp, bool first) {
#pragma unroll
        for (int u4 = 0; u4 < 4; ++u4) {
            const int b = u4 & 1, nb = b ^ 1;
            const bool nxt = u4 < 3 || more;
            const int nq = u4 < 3 ? q : q ^ 1, nu4 = (u4 + 1) & 3;
            if (u4 == 3 && more) __syncthreads();
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                mf(u4, b, m, first);
                WSB();
                if (nxt) {
                    if (m < 4) rdp(nq, nu4, m);                       // gaps 0..3: one patch pixel each (two rows for xi = 0)
                    else if (m < 6) { rdv(nq, nu4, nb, 2 * (m - 4)); rdv(nq, nu4, nb, 2 * (m - 4) + 1); }
                    else if (m < 14) tstep(nu4, nb, m - 6);           // gaps 6..13: four additions each
                }
                if (more && m < 10) {                                 // staging of the next half-phase: one operation per gap
                    if (u4 == 0) gload1(nexthp, m);
                    if (u4 == 2) lstore(q ^ 1, m);
                }
                WSB();
            }
        }
    };

    WINO_STAMP(1);
#pragma unroll
    for (int i = 0; i < WSLOTS + 4; ++i) gload1(0, i);
    // data gradient: the ReLU mask of this lane's 2 x 2 pixels x 16 columns as 64 bits, loaded while the first patch is in flight
    if (KIND == WK_DGRAD && a.mask) {   // the producer's forward left the mask as bits in this kernel's lane order: one 8-byte load
        const uint2 mb = *reinterpret_cast<const uint2*>(a.mask + ((size_t)id * 256 + tid) * 2);
        mbits[0] = mb.x; mbits[1] = mb.y;
    } else if (KIND == WK_DGRAD && a.aux) {
        mbits[0] = mbits[1] = 0u;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
#pragma unroll
            for (int aa = 0; aa < 2; ++aa)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) {
                    const bool ok = aa == 0 ? (bb == 0 ? ok00 : ok01) : (bb == 0 ? ok10 : ok11);
                    const float4 m = ok ? *reinterpret_cast<const float4*>(a.aux + p00 + aa * rowN + bb * N + n0 + 8 * rg + 4 * lh) : f4zero();
                    const unsigned bits = (m.x > 0.f ? 1u : 0u) | (m.y > 0.f ? 2u : 0u) | (m.z > 0.f ? 4u : 0u) | (m.w > 0.f ? 8u : 0u);
                    mbits[rg >> 1] |= bits << (16 * (rg & 1) + 8 * aa + 4 * bb);
                }
    }
#pragma unroll
    for (int i = 0; i < WSLOTS + 4; ++i) lstore(0, i);
    WINO_STAMP(2);
    __syncthreads();
    WINO_STAMP(3);
#pragma unroll
    for (int j = 0; j < 4; ++j) rdp(0, 0, j);
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) rdv(0, 0, 0, nu);
#pragma unroll
    for (int k = 0; k < 8; ++k) tstep(0, 0, k);
    WSB();
    WINO_STAMP(4);
    {   // chunk 0: its first half-phase starts the accumulators
        const bool more = 1 < a.nchunks;
        half(0, true, more, 1, true);
        half(1, more, more, 2, false);
    }
    for (int ch = 1; ch < a.nchunks; ++ch) {
        const bool more = ch + 1 < a.nchunks;
        half(0, true, more, 2 * ch + 1, false);
        half(1, more, more, 2 * ch + 2, false);
    }
#undef WSB
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    WINO_STAMP(5);

    // ---- output transform + epilogue: acc[p][r] = M_p[column n0 + 8 (r >> 2) + 4 lh + (r & 3)][tile li]
    unsigned obits[2] = {0u, 0u};
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
        const int col = n0 + 8 * rg + 4 * lh;
        float4 Y[2][2];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float s[2][4];
#pragma unroll
            for (int nu = 0; nu < 4; ++nu) {
                const float m0 = acc[nu][4 * rg + k], m1 = acc[4 + nu][4 * rg + k], m2 = acc[8 + nu][4 * rg + k], m3 = acc[12 + nu][4 * rg + k];
                s[0][nu] = m0 + m1 + m2;
                s[1][nu] = m1 - m2 - m3;
            }
#pragma unroll
            for (int aa = 0; aa < 2; ++aa) {
                const float y0v = s[aa][0] + s[aa][1] + s[aa][2], y1v = s[aa][1] - s[aa][2] - s[aa][3];
                if (k == 0) { Y[aa][0].x = y0v; Y[aa][1].x = y1v; }
                if (k == 1) { Y[aa][0].y = y0v; Y[aa][1].y = y1v; }
                if (k == 2) { Y[aa][0].z = y0v; Y[aa][1].z = y1v; }
                if (k == 3) { Y[aa][0].w = y0v; Y[aa][1].w = y1v; }
            }
        }
        if (KIND == WK_FWD) {   // (the bias is already in the accumulators)
            if (a.relu) {
#pragma unroll
                for (int aa = 0; aa < 2; ++aa)
#pragma unroll
                    for (int bb = 0; bb < 2; ++bb) {
                        float4& v = Y[aa][bb];
                        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                    }
            }
        } else if (a.aux || a.mask) {
#pragma unroll
            for (int aa = 0; aa < 2; ++aa)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) {
                    const unsigned mb = mbits[rg >> 1] >> (16 * (rg & 1) + 8 * aa + 4 * bb);
                    float4& v = Y[aa][bb];
                    if (!(mb & 1u)) v.x = 0.f;
                    if (!(mb & 2u)) v.y = 0.f;
                    if (!(mb & 4u)) v.z = 0.f;
                    if (!(mb & 8u)) v.w = 0.f;
                }
        }
        if (KIND == WK_FWD && a.mask) {
#pragma unroll
            for (int aa = 0; aa < 2; ++aa)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) {
                    const float4& v = Y[aa][bb];
                    const unsigned bits = (v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u);
                    obits[rg >> 1] |= bits << (16 * (rg & 1) + 8 * aa + 4 * bb);
                }
        }
        if (ok00) *reinterpret_cast<float4*>(a.out + p00 + col) = Y[0][0];
        if (ok01) *reinterpret_cast<float4*>(a.out + p00 + N + col) = Y[0][1];
        if (ok10) *reinterpret_cast<float4*>(a.out + p00 + rowN + col) = Y[1][0];
        if (ok11) *reinterpret_cast<float4*>(a.out + p00 + rowN + N + col) = Y[1][1];
        if (POOL && ok11) {
            float4 m;
            m.x = fmaxf(fmaxf(Y[0][0].x, Y[0][1].x), fmaxf(Y[1][0].x, Y[1][1].x));
            m.y = fmaxf(fmaxf(Y[0][0].y, Y[0][1].y), fmaxf(Y[1][0].y, Y[1][1].y));
            m.z = fmaxf(fmaxf(Y[0][0].z, Y[0][1].z), fmaxf(Y[1][0].z, Y[1][1].z));
            m.w = fmaxf(fmaxf(Y[0][0].w, Y[0][1].w), fmaxf(Y[1][0].w, Y[1][1].w));
            *reinterpret_cast<float4*>(a.pool + ((long)(b * (g.H >> 1) + (y0 >> 1)) * (g.W >> 1) + (x0 >> 1)) * N + col) = m;
        }
    }
    if (KIND == WK_FWD && a.mask) *reinterpret_cast<uint2*>(a.mask + ((size_t)id * 256 + tid) * 2) = make_uint2(obits[0], obits[1]);
    WINO_STAMP(6);
}

// w [3][3][Ci][Co] (HWIO) -> V = G g G^T, G = [1 0 0; 1/2 1/2 1/2; 1/2 -1/2 1/2; 0 0 1], packed [N/32][C/16][half 2][p 16][lane half 2][n 32][e 4] (channel = 16 chunk + 8 half + 4 lane half + e):
//   transpose 0 (forward):        C = Ci, N = Co, g[ky][kx] = w[ky][kx][c][n]
//   transpose 1 (data gradient):  C = Co, N = Ci, g[ky][kx] = w[2 - ky][2 - kx][n][c]
__global__ __launch_bounds__(256) void wino_pack_kernel(const float* __restrict__ w, int Ci, int Co, int transpose, float* __restrict__ out) {
    const int C = transpose ? Co : Ci, N = transpose ? Ci : Co;
    const long total = (long)C * N;
    const int nchunks = C / WCH;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        // consecutive threads: consecutive n of one c when not transposed (w rows are [c][n]), consecutive c of one n when transposed
        const int c = transpose ? (int)(i % C) : (int)(i / N), n = transpose ? (int)(i / C) : (int)(i % N);
        float gk[3][3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
                gk[ky][kx] = transpose ? w[((long)((2 - ky) * 3 + (2 - kx)) * Ci + n) * Co + c] : w[((long)(ky * 3 + kx) * Ci + c) * Co + n];
        float t[4][3];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            t[0][kx] = gk[0][kx];
            t[1][kx] = 0.5f * (gk[0][kx] + gk[1][kx] + gk[2][kx]);
            t[2][kx] = 0.5f * (gk[0][kx] - gk[1][kx] + gk[2][kx]);
            t[3][kx] = gk[2][kx];
        }
        const int nt = n >> 5, nl = n & 31, ch = c / WCH, quad = (c % WCH) >> 2, e = c & 3;   // quad = 2 * half + lane half
        float* o = out + ((long)nt * nchunks + ch) * WV_FLOATS + (quad >> 1) * 4096 + ((quad & 1) * 32 + nl) * 4 + e;
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) {
            const float v0 = t[xi][0], v1 = 0.5f * (t[xi][0] + t[xi][1] + t[xi][2]), v2 = 0.5f * (t[xi][0] - t[xi][1] + t[xi][2]), v3 = t[xi][2];
            o[(xi * 4 + 0) * 256] = v0;
            o[(xi * 4 + 1) * 256] = v1;
            o[(xi * 4 + 2) * 256] = v2;
            o[(xi * 4 + 3) * 256] = v3;
        }
    }
}

// ---- host side ----------------------------------------------------------------------------------
// Block shapes (tiles): wide images take 4 x 8, the 56-wide layers 4 x 7, the 28-wide 2 x 14, the 14-wide 4 x 7 (two blocks per image,
// the second one half empty); in general the widest TBW <= 16 that divides the tile columns, with TBH = 32 / TBW (at most 8).
static bool plan_wino(int B, int H, int W, int C, int N, WinoGeom& g) {
    g.B = B; g.H = H; g.W = W; g.C = C; g.N = N;
    if (B <= 0 || H < 2 || W < 2 || (H & 1) || (W & 1) || C <= 0 || N <= 0 || C % WCH || N % 32) return false;
    if ((long)B * H * W * (long)(C > N ? C : N) * 4 > 0x7fffffffL || 16L * C * N * 4 > 0x7fffffffL) return false;
    const int TW = W / 2, TH = H / 2;
    int best = 0;
    double best_eff = 0.0;
    for (int tbw = 1; tbw <= 16 && tbw <= TW; ++tbw) {
        int tbh = 32 / tbw;
        if (tbh > 8) tbh = 8;
        if (tbh > TH) tbh = TH;
        if ((2 * tbh + 2) * (2 * tbw + 2) > WPIX) continue;
        const long slots = (long)cdiv(TW, tbw) * cdiv(TH, tbh) * 32;
        const double eff = (double)TW * TH / (double)slots;
        if (eff > best_eff + 1e-9) { best_eff = eff; best = tbw; }
    }
    if (!best) return false;
    g.TBW = best;
    g.TBH = 32 / best > 8 ? 8 : 32 / best;
    if (g.TBH > TH) g.TBH = TH;
    g.PW = 2 * g.TBW + 2; g.PH = 2 * g.TBH + 2;
    g.bx_n = cdiv(TW, g.TBW); g.by_n = cdiv(TH, g.TBH);
    g.blocks_img = g.bx_n * g.by_n;
    if ((long)B * g.blocks_img > 0x3fffffffL) return false;
    g.nblocks = B * g.blocks_img;
    if ((long)g.nblocks * g.blocks_img >= 0x100000000L) return false;   // the reciprocal divisions are exact below 2^32 / divisor
    g.m_blocks_img = wino_magic(g.blocks_img); g.m_bx_n = wino_magic(g.bx_n); g.m_pw = wino_magic(g.PW); g.m_tbw = wino_magic(g.TBW);
    return true;
}

static int wino_attr() {
    static int once = [] {
        hipError_t e = hipSuccess;
        auto set = [&](const void* f) { if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, WINO_LDS_BYTES); };
        set(reinterpret_cast<const void*>(conv_wino_kernel<WK_FWD, false>));
        set(reinterpret_cast<const void*>(conv_wino_kernel<WK_FWD, true>));
        set(reinterpret_cast<const void*>(conv_wino_kernel<WK_DGRAD, false>));
        return e == hipSuccess ? 0 : fail((int)e, "%s: hipFuncSetAttribute failed", "conv wino kernel");
    }();
    return once;
}

template <int KIND, bool POOL>
static int launch_wino(hipStream_t st, WinoArgs& a) {
    int rc = wino_attr();
    if (rc) return rc;
    a.tiles_n = a.g.N / 32;
    a.nchunks = a.g.C / WCH;
    a.ntiles = cdiv(a.g.nblocks, 4) * a.tiles_n;
    hipLaunchKernelGGL((conv_wino_kernel<KIND, POOL>), dim3(a.ntiles), dim3(256), WINO_LDS_BYTES, st, a);
    return launch_status("conv wino");
}


}  // namespace vc

extern "C" int vc_conv3x3_wino_supported(int B, int H, int W, int Cin, int Cout, int dgrad) {
    vc::WinoGeom g;
    const int nb = vc::wino_images_per_launch(B, H, W, Cin, Cout);
    if (nb > 0 && vc::wino_version() == 2) return (dgrad ? vc::wino2_plan_ok(nb, H, W, Cout, Cin) : vc::wino2_plan_ok(nb, H, W, Cin, Cout)) ? 1 : 0;
    return nb > 0 && (dgrad ? vc::plan_wino(nb, H, W, Cout, Cin, g) : vc::plan_wino(nb, H, W, Cin, Cout, g)) ? 1 : 0;
}

extern "C" int vc_conv3x3_wino_single_launch_supported(int B, int H, int W, int Cin, int Cout) {
    return vc::wino_images_per_launch(B, H, W, Cin, Cout) >= B ? 1 : 0;
}

extern "C" int vc_conv3x3_wino_pack_f32(void* stream, int Cin, int Cout, const float* w, int transpose, float* wp) {
    using namespace vc;
    const int C = transpose ? Cout : Cin, N = transpose ? Cin : Cout;
    VC_CHECK_ARG(C > 0 && N > 0 && C % WCH == 0 && N % 32 == 0, "gathered channels % 16 == 0 and output channels % 32 == 0 required");
    VC_CHECK_ARG(w && wp && waligned16(wp), "null or misaligned pointer");
    if (wino_version() == 2) return wino2_pack((hipStream_t)stream, Cin, Cout, w, transpose, wp);
    const long total = (long)C * N;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(wino_pack_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, Cin, Cout, transpose, wp);
    return launch_status(__func__);
}

extern "C" int vc_conv3x3_wino_fwd_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* x, const float* wp,
                                       const float* bias, float* y, float* ypool, int relu) {
    using namespace vc;
    VC_CHECK_ARG(x && wp && y, "null pointer");
    VC_CHECK_ARG(waligned16(x) && waligned16(wp) && waligned16(y) && waligned16(bias) && waligned16(ypool), "pointers must be 16-byte aligned");
    const int per = wino_images_per_launch(B, H, W, Cin, Cout);
    VC_CHECK_ARG(per > 0, "unsupported shape (vc_conv3x3_wino_supported)");
    for (int b0 = 0; b0 < B; b0 += per) {   // image ranges of < 2 GiB (one launch for every VGG16 layer up to 160 images)
        const int nb = B - b0 < per ? B - b0 : per;
        if (wino_version() == 2) {
            const int rc = wino2_launch((hipStream_t)stream, 0, nb, H, W, Cin, Cout, x + (size_t)b0 * H * W * Cin, wp, y + (size_t)b0 * H * W * Cout, bias,
                                        ypool ? ypool + (size_t)b0 * (H / 2) * (W / 2) * Cout : nullptr, nullptr, relu);
            if (rc) return rc;
            continue;
        }
        WinoArgs a;
        VC_CHECK_ARG(plan_wino(nb, H, W, Cin, Cout, a.g), "unsupported shape (vc_conv3x3_wino_supported)");
        a.x = x + (size_t)b0 * H * W * Cin; a.wp = wp; a.out = y + (size_t)b0 * H * W * Cout; a.aux = bias; a.relu = relu; a.mask = nullptr;
        a.pool = ypool ? ypool + (size_t)b0 * (H / 2) * (W / 2) * Cout : nullptr;
        const int rc = ypool ? launch_wino<WK_FWD, true>((hipStream_t)stream, a) : launch_wino<WK_FWD, false>((hipStream_t)stream, a);
        if (rc) return rc;
    }
    return 0;
}

extern "C" size_t vc_conv3x3_wino_mask_words(int B, int H, int W, int C) {
    if (vc::wino_version() == 2) return vc::wino2_mask_words(B, H, W, C);
    vc::WinoGeom g;
    if (!vc::plan_wino(B, H, W, 16, C, g)) return 0;
    return (size_t)vc::cdiv(g.nblocks, 4) * (C / 32) * 256 * 2;
}

extern "C" int vc_conv3x3_wino_fwd_mask_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* x, const float* wp,
                                            const float* bias, float* y, int relu, uint32_t* mask_out) {
    using namespace vc;
    WinoArgs a;
    VC_CHECK_ARG(x && wp && y && mask_out, "null pointer");
    VC_CHECK_ARG(waligned16(x) && waligned16(wp) && waligned16(y) && waligned16(bias) && waligned16(mask_out), "pointers must be 16-byte aligned");
    if (wino_version() == 2) {
        VC_CHECK_ARG(wino_images_per_launch(B, H, W, Cin, Cout) >= B && wino2_plan_ok(B, H, W, Cin, Cout),
                     "unsupported shape, or more images than one launch takes (vc_conv3x3_wino_single_launch_supported): the mask bits are per tile of ONE launch");
        return wino2_launch((hipStream_t)stream, 0, B, H, W, Cin, Cout, x, wp, y, bias, nullptr, mask_out, relu);
    }
    VC_CHECK_ARG(wino_images_per_launch(B, H, W, Cin, Cout) >= B && plan_wino(B, H, W, Cin, Cout, a.g),
                 "unsupported shape, or more images than one launch takes (vc_conv3x3_wino_single_launch_supported): the mask bits are per tile of ONE launch");
    a.x = x; a.wp = wp; a.out = y; a.aux = bias; a.relu = relu; a.pool = nullptr; a.mask = mask_out;
    return launch_wino<WK_FWD, false>((hipStream_t)stream, a);
}

extern "C" int vc_conv3x3_wino_dgrad_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* dy, const float* wpt,
                                         const float* relu_src, float* dx) {
    using namespace vc;
    VC_CHECK_ARG(dy && wpt && dx, "null pointer");
    VC_CHECK_ARG(waligned16(dy) && waligned16(wpt) && waligned16(dx) && waligned16(relu_src), "pointers must be 16-byte aligned");
    const int per = wino_images_per_launch(B, H, W, Cin, Cout);
    VC_CHECK_ARG(per > 0, "unsupported shape (vc_conv3x3_wino_supported)");
    for (int b0 = 0; b0 < B; b0 += per) {
        const int nb = B - b0 < per ? B - b0 : per;
        if (wino_version() == 2) {
            const int rc = wino2_launch((hipStream_t)stream, 1, nb, H, W, Cout, Cin, dy + (size_t)b0 * H * W * Cout, wpt, dx + (size_t)b0 * H * W * Cin,
                                        relu_src ? relu_src + (size_t)b0 * H * W * Cin : nullptr, nullptr, nullptr, 0);
            if (rc) return rc;
            continue;
        }
        WinoArgs a;
        VC_CHECK_ARG(plan_wino(nb, H, W, Cout, Cin, a.g), "unsupported shape (vc_conv3x3_wino_supported)");
        a.x = dy + (size_t)b0 * H * W * Cout; a.wp = wpt; a.out = dx + (size_t)b0 * H * W * Cin;
        a.aux = relu_src ? relu_src + (size_t)b0 * H * W * Cin : nullptr; a.relu = 0; a.pool = nullptr; a.mask = nullptr;
        const int rc = launch_wino<WK_DGRAD, false>((hipStream_t)stream, a);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int vc_conv3x3_wino_dgrad_bits_f32(void* stream, int B, int H, int W, int Cin, int Cout, const float* dy, const float* wpt,
                                              const uint32_t* mask_bits, float* dx) {
    using namespace vc;
    WinoArgs a;
    VC_CHECK_ARG(dy && wpt && dx && mask_bits, "null pointer");
    VC_CHECK_ARG(waligned16(dy) && waligned16(wpt) && waligned16(dx) && waligned16(mask_bits), "pointers must be 16-byte aligned");
    if (wino_version() == 2) {
        VC_CHECK_ARG(wino_images_per_launch(B, H, W, Cin, Cout) >= B && wino2_plan_ok(B, H, W, Cout, Cin),
                     "unsupported shape, or more images than one launch takes (vc_conv3x3_wino_single_launch_supported)");
        return wino2_launch((hipStream_t)stream, 1, B, H, W, Cout, Cin, dy, wpt, dx, nullptr, nullptr, const_cast<uint32_t*>(mask_bits), 0);
    }
    VC_CHECK_ARG(wino_images_per_launch(B, H, W, Cin, Cout) >= B && plan_wino(B, H, W, Cout, Cin, a.g),
                 "unsupported shape, or more images than one launch takes (vc_conv3x3_wino_single_launch_supported)");
    a.x = dy; a.wp = wpt; a.out = dx; a.aux = nullptr; a.relu = 0; a.pool = nullptr; a.mask = const_cast<uint32_t*>(mask_bits);
    return launch_wino<WK_DGRAD, false>((hipStream_t)stream, a);
}
