// Fused LSTM time-step kernels (forward and backward-through-time) for gfx950.
//
// Reference: utils/rnn_model.py:23-51 builds MultiRNNCell[DropoutWrapper(LSTMCell(H))];
// it is stepped at vae_model/encoder.py:46,48,49-55 and vae_model/decoder.py:100,102,113,
// 116-121.  TF LSTMCell (TF-sem.): g = [x,h].W + b; i,j,f,o = split(g,4);
// c' = sigmoid(f+1)*c + sigmoid(i)*tanh(j); h' = sigmoid(o)*tanh(c').
//
// MI355X design: the eight per-gate GEMMs of a step collapse into TWO MFMA GEMMs --
//   (1) the input projection of ALL steps at once,  G = X[T*N,E].Wx + b  (vc_gemm_f32),
//   (2) per step the recurrent  h[N,H].Wh[H,4H]  fused with the gate math in its epilogue.
// In (2) a workgroup owns 32 hidden units x all four gates: its 128 tile columns are
// {i,j,f,o} x 32 units, so the four accumulators of a lane hold the four gates of the same
// (row, unit) and the gate nonlinearity, cell update, length mask and state write are pure
// per-lane register math -- no LDS round trip, no second kernel.
#include "gemm_core.h"
#include "gemm_bf16x3_core.h"
#include "vaecap.h"

namespace vc {

// B operand of the forward step: Wh [H, 4H] row-major viewed as tile columns
// c -> gate c/32, unit u0 + c%32.
struct LoadWhGates {
    const float* p;  // Wh
    int H, u0;
    const float* q[MAXNV];
    int ko[MAXNV];
    __device__ __forceinline__ void init(int u, int c, int kofs) {
        q[u] = p + (long)kofs * 4 * H + (c >> 5) * H + u0 + (c & 31);
        ko[u] = kofs;
    }
    __device__ __forceinline__ float4 load(int u, int k0) const {
        if (k0 + ko[u] >= H) return f4zero();
        return *reinterpret_cast<const float4*>(q[u] + (long)k0 * 4 * H);
    }
};

struct LstmFwdArgs {
    const float* h_prev;  // [N,H]
    const float* c_prev;  // [N,H]
    const float* Wh;      // [H,4H]
    float* gact;          // [N,4H] in: x-projection + bias; out: gate activations i,j,f,o
    const int32_t* lens;  // [N] effective lengths
    float* c_out;
    float* h_out;
    int N, H, t;
};

template <class CFG>
__global__ __launch_bounds__(CFG::NT) void lstm_step_fwd_kernel(LstmFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int m0 = blockIdx.x * CFG::BM;
    const int u0 = blockIdx.y * 32;
    f32x16 acc[1][4];
    acc_zero<CFG>(acc);
    LoadMK<true> la;
    la.p = a.h_prev; la.ld = a.H; la.R = a.N; la.K = a.H;
    LoadWhGates lb;
    lb.p = a.Wh; lb.H = a.H; lb.u0 = u0;
    mfma_mainloop<CFG, MODE_MK, MODE_KM>(acc, la, lb, m0, 0, 0, a.H, smem);
    AccCoord<CFG> co;
    const int u = u0 + co.li;
    const int H = a.H;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + co.row(0, r);
        if (row >= a.N) continue;
        float* g = a.gact + (long)row * 4 * H + u;
        const float gi = acc[0][0][r] + g[0];
        const float gj = acc[0][1][r] + g[H];
        const float gf = acc[0][2][r] + g[2 * H];
        const float go = acc[0][3][r] + g[3 * H];
        const float i = sigmoidf_(gi), j = tanhf(gj), f = sigmoidf_(gf + 1.0f), o = sigmoidf_(go);
        const long si = (long)row * H + u;
        const float cp = a.c_prev[si];
        const float c = f * cp + i * j;
        const float h = o * tanhf(c);
        g[0] = i; g[H] = j; g[2 * H] = f; g[3 * H] = o;
        const bool active = a.t < a.lens[row];  // dynamic_rnn: state copied through past the length
        a.c_out[si] = active ? c : cp;
        a.h_out[si] = active ? h : a.h_prev[si];
    }
}

struct LstmBwdArgs {
    const float* dG_next;  // [N,4H] gate gradients of step t+1 (unused when first)
    const float* Wh;       // [H,4H]
    const int32_t* lens;
    const float* dh_ext;   // [N,H] external gradient w.r.t. hs[t+1], may be null
    float* dH_run;         // [N,H] in/out: total gradient w.r.t. hs[t+2] -> hs[t+1]
    float* dC_run;         // [N,H] in/out: gradient w.r.t. cs[t+1] -> cs[t]
    const float* act;      // [N,4H] gate activations of step t
    const float* c_prev;   // cs[t]
    const float* c_cur;    // cs[t+1]
    float* dG;             // [N,4H] out: gate pre-activation gradients of step t
    int N, H, t, first;
};

template <class CFG>
__global__ __launch_bounds__(CFG::NT) void lstm_step_bwd_kernel(LstmBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int m0 = blockIdx.x * CFG::BM;
    const int n0 = blockIdx.y * CFG::BN;
    const int H = a.H;
    f32x16 acc[CFG::TM][CFG::TN];
    acc_zero<CFG>(acc);
    if (!a.first) {
        LoadMK<true> la, lb;  // (dG.Wh^T)[row,u] = sum_k dG[row,k] * Wh[u,k]
        la.p = a.dG_next; la.ld = 4L * H; la.R = a.N; la.K = 4 * H;
        lb.p = a.Wh; lb.ld = 4L * H; lb.R = H; lb.K = 4 * H;
        mfma_mainloop<CFG, MODE_MK, MODE_MK>(acc, la, lb, m0, n0, 0, 4 * H, smem);
    }
    AccCoord<CFG> co;
#pragma unroll
    for (int tn = 0; tn < CFG::TN; ++tn) {
        const int u = n0 + co.col(tn);
        if (u >= H) continue;
#pragma unroll
        for (int tm = 0; tm < CFG::TM; ++tm) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + co.row(tm, r);
                if (row >= a.N) continue;
                const long si = (long)row * H + u;
                const int len = a.lens[row];
                float dh = a.dH_run[si];
                if (!a.first && (a.t + 1 < len)) dh = acc[tm][tn][r];  // step t+1 was active
                if (a.dh_ext) dh += a.dh_ext[si];
                a.dH_run[si] = dh;
                const float* ac = a.act + (long)row * 4 * H + u;
                float* dg = a.dG + (long)row * 4 * H + u;
                if (a.t < len) {
                    const float i = ac[0], j = ac[H], f = ac[2 * H], o = ac[3 * H];
                    const float tc = tanhf(a.c_cur[si]);
                    const float dct = a.dC_run[si] + dh * o * (1.f - tc * tc);
                    dg[0] = dct * j * i * (1.f - i);
                    dg[H] = dct * i * (1.f - j * j);
                    dg[2 * H] = dct * a.c_prev[si] * f * (1.f - f);
                    dg[3 * H] = dh * tc * o * (1.f - o);
                    a.dC_run[si] = dct * f;
                } else {
                    dg[0] = 0.f; dg[H] = 0.f; dg[2 * H] = 0.f; dg[3 * H] = 0.f;
                }
            }
        }
    }
}

// ---- split form: the recurrent GEMM through vc_gemm_f32 (64x64 tiles + split-K fill the chip even
// at N = 320 rows, where the fused kernel above has only 50 workgroups) followed by these
// element-wise gate kernels.  Same arithmetic, same buffers; chosen by the sequence drivers.
// part: `nsplit` split-K partial products h_{t-1}.Wh [N, 4H] each (fixed summation order), or null when the recurrent
// product has already been accumulated into gact.
__global__ __launch_bounds__(256) void lstm_gates_fwd_kernel(float* __restrict__ gact, const float* __restrict__ c_prev,
                                                             const float* __restrict__ h_prev, const int32_t* __restrict__ lens,
                                                             float* __restrict__ c_out, float* __restrict__ h_out, int N, int H, int t,
                                                             const float* __restrict__ part, int nsplit) {
    const long total = (long)N * H;
    const long pstride = (long)N * 4 * H;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int row = (int)(i / H), u = (int)(i % H);
        float* g = gact + (long)row * 4 * H + u;
        float p0 = g[0], p1 = g[H], p2 = g[2 * H], p3 = g[3 * H];
        if (part) {
            const float* q = part + (long)row * 4 * H + u;
            float r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f;
            for (int z = 0; z < nsplit; ++z, q += pstride) { r0 += q[0]; r1 += q[H]; r2 += q[2 * H]; r3 += q[3 * H]; }
            p0 += r0; p1 += r1; p2 += r2; p3 += r3;
        }
        const float ig = sigmoidf_(p0), jg = tanhf(p1), fg = sigmoidf_(p2 + 1.0f), og = sigmoidf_(p3);
        const float cp = c_prev[i];
        const float c = fg * cp + ig * jg;
        const float h = og * tanhf(c);
        g[0] = ig; g[H] = jg; g[2 * H] = fg; g[3 * H] = og;
        const bool active = t < lens[row];
        c_out[i] = active ? c : cp;
        h_out[i] = active ? h : h_prev[i];
    }
}

__global__ __launch_bounds__(256) void lstm_gates_bwd_kernel(const float* __restrict__ rec, const int32_t* __restrict__ lens,
                                                             const float* __restrict__ dh_ext, float* __restrict__ dH_run,
                                                             float* __restrict__ dC_run, const float* __restrict__ act,
                                                             const float* __restrict__ c_prev, const float* __restrict__ c_cur,
                                                             float* __restrict__ dG, int N, int H, int t, int first, int nsplit) {
    const long total = (long)N * H;
    for (long si = (long)blockIdx.x * 256 + threadIdx.x; si < total; si += (long)gridDim.x * 256) {
        const int row = (int)(si / H), u = (int)(si % H);
        const int len = lens[row];
        float dh = dH_run[si];
        if (!first && (t + 1 < len)) {  // rec: nsplit split-K partials of dG[t+1].Wh^T, summed in fixed order
            dh = rec[si];
            for (int z = 1; z < nsplit; ++z) dh += rec[(long)z * total + si];
        }
        if (dh_ext) dh += dh_ext[si];
        dH_run[si] = dh;
        const float* ac = act + (long)row * 4 * H + u;
        float* dg = dG + (long)row * 4 * H + u;
        if (t < len) {
            const float i = ac[0], j = ac[H], f = ac[2 * H], o = ac[3 * H];
            const float tc = tanhf(c_cur[si]);
            const float dct = dC_run[si] + dh * o * (1.f - tc * tc);
            dg[0] = dct * j * i * (1.f - i);
            dg[H] = dct * i * (1.f - j * j);
            dg[2 * H] = dct * c_prev[si] * f * (1.f - f);
            dg[3 * H] = dh * tc * o * (1.f - o);
            dC_run[si] = dct * f;
        } else {
            dg[0] = 0.f; dg[H] = 0.f; dg[2 * H] = 0.f; dg[3 * H] = 0.f;
        }
    }
}

typedef unsigned int lstm_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 lbuf(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    lstm_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
    return *reinterpret_cast<float4*>(&v);
}
__device__ __forceinline__ float lcomp(const float4& v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; }

// ---- register-operand recurrence kernels (H = 512) ------------------------------------------------------------------------
// One launch per time step, ONE workgroup per CU: workgroup (ug, rg) owns a narrow column slice of the step -- forward: 8 hidden
// units x 4 gates = 32 gate columns, backward: 16 hidden units -- for a block of rows, and its four waves split K.  The slice of Wh
// a wave needs is pre-packed in MFMA-operand order (one 16-byte load per lane and four k) and sits in registers: forward, the whole
// [128 k x 32 columns] quarter stays resident for the launch (64 VGPRs); backward, [128 k x 16 units] blocks stream through a
// two-slot register ring.  The other operand (rows of h_{t-1}, or of dG[t+1]) is the one with many bytes: each wave streams ITS K
// range of 16-row tiles global -> registers (coalesced 512-byte row segments) -> a wave-private LDS tile -> MFMA fragments, software
// pipelined one tile ahead, with no workgroup barrier in the contraction (DS operations of one wave execute in order).  Tiles are
// 16 x 16 (v_mfma_f32_16x16x4_f32): 320 rows / 4 row groups = 80 rows = five whole tiles, where 32 x 32 tiles would pad to 96.
// The four K-partial tiles meet in LDS (each wave re-uses its own staging area) and the gate math runs on (row, unit) pairs.
//
// k order inside a 128-deep block: lane group kq = lane >> 4 of the 16x16x4 operand layout takes floats kbase(kq) + 4 j + e
// (j = 0..7 the 16-byte read, e its element), kbase = {0, 64, 32, 96}: the two lane groups a ds_read_b128 services together then
// differ by 64 floats = the same banks, and the 16 rows (pitch 132 floats) spread over all 64 banks -> conflict-free fragment reads.
__device__ __forceinline__ int rec_kbase(int kq) { return ((kq & 1) << 6) | ((kq >> 1) << 5); }
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// ---- the split-bf16 ("bf16x3") form of the recurrence kernels (template flag BX; chosen by vc_gemm_set_precision(1) in the sequence
// drivers).  The operand orders are the SAME: a lane's pair of 16-byte fragments (j = 2 m, 2 m + 1) holds EIGHT consecutive k of its
// lane group -- exactly one v_mfma_f32_16x16x32_bf16 operand.  The row operand is split into (hi, lo) in registers right before its
// MFMAs; the packed Wh holds, in the same two float4 slots, the eight hi halves and the eight lo halves (split by the pack kernel).
// Three bf16 MFMAs of 16 cycles replace eight f32 MFMAs of 32: the matrix time of a step drops ~5x (it was 40 % of a 1280-row step,
// profiles/r05_lstm_pmc.md); everything else -- streaming, LDS round trip, K-split reduction, gate math -- is unchanged.
__device__ __forceinline__ void rec_split8(const float4& p, const float4& q, bf16x8& hi, bf16x8& lo) {
    unsigned h[4], l[4];
    split_pair(p.x, p.y, h[0], l[0]);
    split_pair(p.z, p.w, h[1], l[1]);
    split_pair(q.x, q.y, h[2], l[2]);
    split_pair(q.z, q.w, h[3], l[3]);
    hi = __builtin_bit_cast(bf16x8, u32x4{h[0], h[1], h[2], h[3]});
    lo = __builtin_bit_cast(bf16x8, u32x4{l[0], l[1], l[2], l[3]});
}
__device__ __forceinline__ bf16x8 rec_bits(const float4& v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ f32x4 mfma16bx(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
// pack side: the pair (j & ~1, j | 1) of a lane's slots = eight values v[0..7]; slot j gets the hi halves (j even) or the lo halves
__device__ __forceinline__ float4 rec_pack_pair(const float (&v)[8], int odd) {
    unsigned h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) split_pair(v[2 * i], v[2 * i + 1], h[i], l[i]);
    const u32x4 o = odd ? u32x4{l[0], l[1], l[2], l[3]} : u32x4{h[0], h[1], h[2], h[3]};
    return __builtin_bit_cast(float4, o);
}

// tools/probes/rec_trace.hip compiles this file with VC_REC_TRACE: wave 0.. of every workgroup stamps s_memtime at phase edges
#ifdef VC_REC_TRACE
__device__ unsigned long long* g_rec_trace = nullptr;  // [workgroups][4 waves][32 stamps]
#define REC_STAMP(k)                                                                                                          \
    do {                                                                                                                      \
        if (g_rec_trace && (threadIdx.x & 63) == 0)                                                                           \
            g_rec_trace[((blockIdx.y * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6)) * 32 + (k)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define REC_STAMP(k)
#endif

constexpr int REC_PITCH = 132;                       // floats per staged row (128 k + 4 pad)
constexpr int REC_TILE = 16 * REC_PITCH;             // one 16-row tile
constexpr int REC_LDS_BYTES = 4 * 2 * REC_TILE * 4;  // 4 waves x 2 tiles = 67 584 B

// forward: out[(((ug*4 + w)*2 + ct)*8 + j)*64 + lane] = Wh[k .. k+3][col], k = 128 w + kbase(lane>>4) + 4 j,
// col = gate*H + 8 ug + unit with (gate, unit) = ((16 ct + (lane&15)) >> 3, & 7)
__global__ __launch_bounds__(256) void lstm_rec_pack_fwd_kernel(const float* __restrict__ Wh, int H, float4* __restrict__ out, int bx) {
    const int total = (H / 8) * 4 * 2 * 8 * 64;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int lane = i & 63, j = (i >> 6) & 7, ct = (i >> 9) & 1, w = (i >> 10) & 3, ug = i >> 12;
        const int lc = ct * 16 + (lane & 15);
        const int col = (lc >> 3) * H + ug * 8 + (lc & 7);
        if (bx) {
            const float* s = Wh + (long)(128 * w + rec_kbase(lane >> 4) + 4 * (j & ~1)) * 4 * H + col;
            float v[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = s[(long)r * 4 * H];
            out[i] = rec_pack_pair(v, j & 1);
            continue;
        }
        const int k = 128 * w + rec_kbase(lane >> 4) + 4 * j;
        const float* s = Wh + (long)k * 4 * H + col;
        out[i] = make_float4(s[0], s[4L * H], s[8L * H], s[12L * H]);
    }
}

// backward: out[(((ug*4 + w)*4 + sb)*8 + j)*64 + lane] = Wh[16 ug + (lane&15)][k .. k+3], k = 512 w + 128 sb + kbase(lane>>4) + 4 j
__global__ __launch_bounds__(256) void lstm_rec_pack_bwd_kernel(const float* __restrict__ Wh, int H, float4* __restrict__ out, int bx) {
    const int total = (H / 16) * 4 * 4 * 8 * 64;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int lane = i & 63, j = (i >> 6) & 7, sb = (i >> 9) & 3, w = (i >> 11) & 3, ug = i >> 13;
        if (bx) {
            const float* s = Wh + (long)(ug * 16 + (lane & 15)) * 4 * H + 512 * w + 128 * sb + rec_kbase(lane >> 4) + 4 * (j & ~1);
            float v[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = s[r];
            out[i] = rec_pack_pair(v, j & 1);
            continue;
        }
        const int k = 512 * w + 128 * sb + rec_kbase(lane >> 4) + 4 * j;
        out[i] = *reinterpret_cast<const float4*>(Wh + (long)(ug * 16 + (lane & 15)) * 4 * H + k);
    }
}

// Row tile slot i of a workgroup works on row tile (i + rot) % RT of the pass, rot = its column-slice index % RT: the 64 (32)
// workgroups that share a row block then walk its tiles in different orders instead of all asking L2 for the same lines at once.
template <int RT>
__device__ __forceinline__ int rec_rot(int i, int rot) {
    const int r = i + rot;
    return r >= RT ? r - RT : r;
}

// The contraction of one pass: acc[i][ct] += A[row0 + 16 i .., wave K range] . B, units u = sb*RT + i in order.
// A: buffer resource over the row-major operand (rows past its end read as zeros), `pitch` bytes per row, kofs = byte offset of the
// wave's K range in a row.  B: NSB == 1: resident fragments bres; else streamed from bp (1 KB per (sb, j), lane-linear).
// CTS != 0: column tile ct of the streamed B operand lives CTS float4 behind tile 0 (two 16-unit slices of the backward pack, which is
// laid out per slice), else the tiles of a sub-block are consecutive.
template <int NSB, int CT, int RT, bool BX = false, int CTS = 0>
__device__ __forceinline__ void rec_contract(f32x4 (&acc)[RT][CT], const __amdgpu_buffer_rsrc_t ra, const unsigned pitch, const int row0,
                                             const unsigned kofs, const float4 (&bres)[CT][8], const float4* __restrict__ bp, float* As,
                                             const int lane, const int rot) {
    constexpr int U = NSB * RT;
    const int rsub = lane >> 5, seg = lane & 31;
    const unsigned v0 = (unsigned)(row0 + rsub) * pitch + kofs + seg * 16;
    float* wr = As + rsub * REC_PITCH + seg * 4;
    const float* rd = As + (lane & 15) * REC_PITCH + rec_kbase(lane >> 4);
    float4 g[2][8], fr[2][8], bs[2][CT][8];
    auto gload = [&](int u) {  // unit u's 16 rows x 128 k: two rows (2 x 512 contiguous bytes) per wave instruction
        const int sb = u / RT, i = u % RT;
#pragma unroll
        for (int q = 0; q < 8; ++q) g[u & 1][q] = lbuf(ra, v0 + (unsigned)(16 * rec_rot<RT>(i, rot) + 2 * q) * pitch + sb * 512, 0);
    };
    auto stage = [&](int u) {  // registers -> this wave's LDS tile (u & 1)
        float* w = wr + (u & 1) * REC_TILE;
#pragma unroll
        for (int q = 0; q < 8; ++q) *reinterpret_cast<float4*>(w + 2 * q * REC_PITCH) = g[u & 1][q];
    };
    auto frags = [&](int u) {
        const float* r = rd + (u & 1) * REC_TILE;
#pragma unroll
        for (int j = 0; j < 8; ++j) fr[u & 1][j] = *reinterpret_cast<const float4*>(r + 4 * j);
    };
    auto bload = [&](int sb) {
        if constexpr (NSB > 1) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int j = 0; j < 8; ++j) bs[sb & 1][ct][j] = CTS ? bp[(sb * 8 + j) * 64 + ct * CTS] : bp[((sb * CT + ct) * 8 + j) * 64];
        }
    };
    // software pipeline: global loads run TWO units ahead of the MFMAs, the LDS round trip one unit ahead
    REC_STAMP(1);
    gload(0);
    bload(0);
    if (U > 1) gload(1);
    __builtin_amdgcn_sched_barrier(0);
    stage(0);
    if (U > 2) gload(2);
    frags(0);
    __builtin_amdgcn_sched_barrier(0);
    REC_STAMP(2);
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int sb = u / RT, i = u % RT;
        const bool nextb = NSB > 1 && i == 0 && sb + 1 < NSB;
        if (u + 1 < U) {
            stage(u + 1);
            if (u + 3 < U) gload(u + 3);  // into the registers stage(u + 1) has just emptied
            frags(u + 1);
        }
        if (nextb) bload(sb + 1);
        if constexpr (BX) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                bf16x8 ah, al;
                rec_split8(fr[u & 1][2 * m], fr[u & 1][2 * m + 1], ah, al);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const bf16x8 bh = rec_bits(NSB > 1 ? bs[sb & 1][ct][2 * m] : bres[ct][2 * m]);
                    const bf16x8 bl = rec_bits(NSB > 1 ? bs[sb & 1][ct][2 * m + 1] : bres[ct][2 * m + 1]);
                    acc[i][ct] = mfma16bx(al, bh, acc[i][ct]);
                    acc[i][ct] = mfma16bx(ah, bl, acc[i][ct]);
                    acc[i][ct] = mfma16bx(ah, bh, acc[i][ct]);
                }
            }
        } else {
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const float bv = NSB > 1 ? lcomp(bs[sb & 1][ct][j], e) : lcomp(bres[ct][j], e);
                    acc[i][ct] = mfma16(lcomp(fr[u & 1][j], e), bv, acc[i][ct]);
                }
        }
        // One wave per SIMD: nothing else fills the matrix pipe while this wave issues its LDS / memory instructions, so they are
        // spread between the unit's MFMAs (CT MFMAs, then one of: 8 LDS writes, 8 LDS reads, 8 global loads, 8 CT B loads).
        // (BX: twelve CT short MFMAs per unit instead of 32 CT -- the compiler's own interleave is kept.)
        if (!BX && u + 1 < U) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, CT, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, CT, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            if (u + 3 < U) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, CT, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
            }
        }
        if (!BX && nextb) {
#pragma unroll
            for (int q = 0; q < 8 * CT; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        REC_STAMP(3 + u);
    }
}

// K-partial tiles of a wave -> its own LDS area: red[row][RP], row = 16 i + 4 (lane >> 4) + v, column = 16 ct + (lane & 15)
template <int CT, int RT, int RP>
__device__ __forceinline__ void rec_spill(const f32x4 (&acc)[RT][CT], float* red, const int lane, const int rot) {
    float* p = red + (4 * (lane >> 4)) * RP + (lane & 15);
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        float* pi = p + 16 * rec_rot<RT>(i, rot) * RP;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int v = 0; v < 4; ++v) pi[v * RP + 16 * ct] = acc[i][ct][v];
    }
}

// sigmoid / tanh of the recurrence kernels: v_exp_f32 + v_rcp_f32 (1 ulp each; |error| < 3e-7 absolute, saturating correctly)
__device__ __forceinline__ float rsigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float rtanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float& at(float4& v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; }
__device__ __forceinline__ float at(const float4& v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; }

// Gate math of both kernels: one thread per (row, four consecutive units) item, every global operand of the item is loaded BEFORE
// the contraction (none depends on it), so the step's tail is LDS reads, arithmetic and 16-byte stores only.
template <int RT, bool BX = false>
__global__ __launch_bounds__(256, 1) void lstm_rec_fwd_kernel(LstmFwdArgs a, const float4* __restrict__ whp, int rows_wg) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int RP = 36;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ug = blockIdx.x, N = a.N, rot = ug % RT;
    constexpr int H = 512;
    const int rbeg = blockIdx.y * rows_wg, rend = min(N, rbeg + rows_wg);
    REC_STAMP(0);
    float4 bres[2][8];
    {
        const float4* bp = whp + (size_t)(ug * 4 + wave) * (2 * 8 * 64) + lane;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int j = 0; j < 8; ++j) bres[ct][j] = bp[(ct * 8 + j) * 64];
    }
    const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc((void*)a.h_prev, 0, N * H * 4, 0x00020000);
    float* As = smem + wave * (2 * REC_TILE);
    const int rr = tid >> 1, hf = tid & 1;  // item: row rr of the pass, units 8 ug + 4 hf .. + 3  (32 RT items <= 160 threads)
    for (int row0 = rbeg; row0 < rend; row0 += 16 * RT) {
        const int row = row0 + rr;
        const bool has = tid < 32 * RT && row < rend;
        const int rowc = min(row, N - 1);  // (threads without an item load a valid row too: no branch, no copies behind the loads)
        float4 gx[4];
        const long si = (long)rowc * H + ug * 8 + 4 * hf;
        float* gp = a.gact + (long)rowc * 4 * H + ug * 8 + 4 * hf;
#pragma unroll
        for (int g = 0; g < 4; ++g) gx[g] = ld4(gp + g * H);
        const float4 cp = ld4(a.c_prev + si), hp = ld4(a.h_prev + si);
        const int len = a.lens[rowc];
        f32x4 acc[RT][2];
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) acc[i][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (row0 != rbeg) __syncthreads();  // the previous pass's gate math has read every wave's partials
        rec_contract<1, 2, RT, BX>(acc, rh, H * 4, row0, wave * 512, bres, nullptr, As, lane, rot);
        rec_spill<2, RT, RP>(acc, As, lane, rot);
        REC_STAMP(28);
        __syncthreads();
        REC_STAMP(29);
        if (has) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 s = f4zero();
#pragma unroll
                for (int w = 0; w < 4; ++w) {  // fixed order
                    const float4 v = ld4(smem + w * (2 * REC_TILE) + rr * RP + g * 8 + 4 * hf);
                    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
                }
                gx[g].x += s.x; gx[g].y += s.y; gx[g].z += s.z; gx[g].w += s.w;
            }
            float4 c4, h4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float i = rsigmoid(at(gx[0], e)), j = rtanh(at(gx[1], e)), f = rsigmoid(at(gx[2], e) + 1.0f), o = rsigmoid(at(gx[3], e));
                const float c = f * at(cp, e) + i * j;
                const float h = o * rtanh(c);
                at(gx[0], e) = i; at(gx[1], e) = j; at(gx[2], e) = f; at(gx[3], e) = o;
                const bool active = a.t < len;  // dynamic_rnn: state copied through past the length
                at(c4, e) = active ? c : at(cp, e);
                at(h4, e) = active ? h : at(hp, e);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) st4(gp + g * H, gx[g]);
            st4(a.c_out + si, c4);
            st4(a.h_out + si, h4);
        }
        REC_STAMP(30);
    }
}

// CT = 2: 32 hidden units (two 16-unit column tiles) per workgroup -- half the column slices, so half the re-reads of dG[t+1] through the
// L2 (profiles/r05_lstm_pmc.md: what bounds the step at 1280 rows), and twice the MFMAs per staged row tile.
template <int RT, bool BX = false, int CT = 1>
__global__ __launch_bounds__(256, 1) void lstm_rec_bwd_kernel(LstmBwdArgs a, const float4* __restrict__ whp, int rows_wg) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int RP = 16 * CT + 4, UW = 16 * CT, QW = 4 * CT;   // partial-tile row pitch; units and unit quads per workgroup
    static_assert(16 * RT * RP <= 2 * REC_TILE, "a wave's partial tile lives in its two staging tiles");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ug = blockIdx.x, N = a.N, rot = ug % RT;
    constexpr int H = 512, K = 4 * H;
    constexpr int ITEMS = 64 * RT * CT, ITERS = (ITEMS + 255) / 256;  // item: (row of the pass, four consecutive units of the 16 CT)
    const int rbeg = blockIdx.y * rows_wg, rend = min(N, rbeg + rows_wg);
    REC_STAMP(0);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)a.dG_next, 0, a.first ? 0 : N * K * 4, 0x00020000);
    const float4* bp = whp + (size_t)(ug * CT * 4 + wave) * (4 * 8 * 64) + lane;   // (slice ug * CT; slice ug * CT + 1 lies 4 x 2048 float4 behind)
    float* As = smem + wave * (2 * REC_TILE);
    const float4 nob[CT][8] = {};
    struct Item {
        float4 ac[4], cc, cpv, dc, dh, ext;
        int len;
    };
    for (int row0 = rbeg; row0 < rend; row0 += 16 * RT) {
        auto item_ok = [&](int it) { return tid + 256 * it < ITEMS && row0 + (tid + 256 * it) / QW < rend; };
        auto fetch = [&](int it, Item& m) {
            const int p = tid + 256 * it, row = min(row0 + p / QW, N - 1);  // (clamped: threads without an item load a valid row)
            const long si = (long)row * H + ug * UW + 4 * (p % QW);
            const float* ac = a.act + (long)row * 4 * H + ug * UW + 4 * (p % QW);
#pragma unroll
            for (int g = 0; g < 4; ++g) m.ac[g] = ld4(ac + g * H);
            m.cc = ld4(a.c_cur + si);
            m.cpv = ld4(a.c_prev + si);
            m.dc = ld4(a.dC_run + si);
            m.dh = ld4(a.dH_run + si);
            m.ext = ld4((a.dh_ext ? a.dh_ext : a.dH_run) + si);  // (used only when dh_ext is given)
            m.len = a.lens[row];
        };
        auto finish = [&](int it, Item& m) {
            const int p = tid + 256 * it, rr = p / QW, row = row0 + rr;
            const long si = (long)row * H + ug * UW + 4 * (p % QW);
            float4 dh = m.dh;
            if (!a.first && (a.t + 1 < m.len)) {  // step t+1 was active: its recurrent gradient replaces the carried one
                float4 s = f4zero();
#pragma unroll
                for (int w = 0; w < 4; ++w) {  // fixed order
                    const float4 v = ld4(smem + w * (2 * REC_TILE) + rr * RP + 4 * (p % QW));
                    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
                }
                dh = s;
            }
            if (a.dh_ext) { dh.x += m.ext.x; dh.y += m.ext.y; dh.z += m.ext.z; dh.w += m.ext.w; }
            st4(a.dH_run + si, dh);
            float4 dg[4], dcn = m.dc;
#pragma unroll
            for (int g = 0; g < 4; ++g) dg[g] = f4zero();
            if (a.t < m.len) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float i = at(m.ac[0], e), j = at(m.ac[1], e), f = at(m.ac[2], e), o = at(m.ac[3], e);
                    const float tc = rtanh(at(m.cc, e));
                    const float dhe = at(dh, e);
                    const float dct = at(m.dc, e) + dhe * o * (1.f - tc * tc);
                    at(dg[0], e) = dct * j * i * (1.f - i);
                    at(dg[1], e) = dct * i * (1.f - j * j);
                    at(dg[2], e) = dct * at(m.cpv, e) * f * (1.f - f);
                    at(dg[3], e) = dhe * tc * o * (1.f - o);
                    at(dcn, e) = dct * f;
                }
                st4(a.dC_run + si, dcn);
            }
            float* dgp = a.dG + (long)row * 4 * H + ug * UW + 4 * (p % QW);
#pragma unroll
            for (int g = 0; g < 4; ++g) st4(dgp + g * H, dg[g]);
        };
        Item m0;
        const bool ok0 = item_ok(0);
        fetch(0, m0);
        if (!a.first) {
            f32x4 acc[RT][CT];
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) acc[i][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (row0 != rbeg) __syncthreads();
            rec_contract<4, CT, RT, BX, CT == 1 ? 0 : 4 * 4 * 8 * 64>(acc, rg, K * 4, row0, wave * 2048, nob, bp, As, lane, rot);
            rec_spill<CT, RT, RP>(acc, As, lane, rot);
            REC_STAMP(28);
            __syncthreads();
            REC_STAMP(29);
        }
        if (ok0) finish(0, m0);
#pragma unroll
        for (int it = 1; it < ITERS; ++it)
            if (item_ok(it)) {
                Item m;
                fetch(it, m);
                finish(it, m);
            }
        REC_STAMP(30);
    }
}

// ---- forward recurrence for many rows (N > 640): eight waves, 16 hidden units x 4 gates per workgroup ---------------------------
// Above ~640 rows the four-wave kernel needs several passes per workgroup and re-reads the h rows once per 8-unit column slice
// (64 slices).  This variant halves that: 32 column slices of 16 units (64 gate columns, four 16-wide MFMA column tiles: a patch
// fragment feeds four MFMAs), EIGHT waves split K (64 k each, so the resident Wh fragments still fit: 4 x 4 float4 = 64 VGPRs),
// 32 x 8 workgroups of 160 rows at N = 1280 = two passes of five 16-row tiles.  The eight partial tiles meet in LDS in two halves
// (gates i, j then f, o: 8 x 80 x 36 floats = 92 KB each).
constexpr int R8_PITCH = 68;                 // floats per staged row (64 k + 4 pad)
constexpr int R8_TILE = 16 * R8_PITCH;       // one 16-row tile
constexpr int R8_WAVE = 80 * 36;             // floats per wave: its partial half tile (>= its two staging tiles)
constexpr int R8_LDS_BYTES = 8 * R8_WAVE * 4;

// out[((((ug*8 + w)*4 + ct)*4 + j)*64 + lane] = Wh[k .. k+3][col], k = 64 w + 16 (lane>>4) + 4 j, col = ct*H + 16 ug + (lane&15)
__global__ __launch_bounds__(256) void lstm_rec8_pack_kernel(const float* __restrict__ Wh, int H, float4* __restrict__ out, int bx) {
    const int total = (H / 16) * 8 * 4 * 4 * 64;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int lane = i & 63, j = (i >> 6) & 3, ct = (i >> 8) & 3, w = (i >> 10) & 7, ug = i >> 13;
        const int col = ct * H + ug * 16 + (lane & 15);
        if (bx) {
            const float* s = Wh + (long)(64 * w + 16 * (lane >> 4) + 4 * (j & ~1)) * 4 * H + col;
            float v[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = s[(long)r * 4 * H];
            out[i] = rec_pack_pair(v, j & 1);
            continue;
        }
        const int k = 64 * w + 16 * (lane >> 4) + 4 * j;
        const float* s = Wh + (long)k * 4 * H + col;
        out[i] = make_float4(s[0], s[4L * H], s[8L * H], s[12L * H]);
    }
}

template <bool BX>
__global__ __launch_bounds__(512, 1) void lstm_rec8_fwd_kernel(LstmFwdArgs a, const float4* __restrict__ whp, int rows_wg) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int RT = 5, H = 512, RP = 36;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ug = blockIdx.x, N = a.N, rot = ug % RT;
    const int rbeg = blockIdx.y * rows_wg, rend = min(N, rbeg + rows_wg);
    float4 bres[4][4];
    {
        const float4* bp = whp + (size_t)(ug * 8 + wave) * (4 * 4 * 64) + lane;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int j = 0; j < 4; ++j) bres[ct][j] = bp[(ct * 4 + j) * 64];
    }
    const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc((void*)a.h_prev, 0, N * H * 4, 0x00020000);
    float* As = smem + wave * R8_WAVE;
    // staging: lane -> (row lane >> 4 of a group of four, 16-byte segment lane & 15); fragments: lane -> (row lane & 15, k group lane >> 4)
    const int rsub = lane >> 4, seg = lane & 15;
    float* wr = As + rsub * R8_PITCH + seg * 4;
    const float* rd = As + (lane & 15) * R8_PITCH + 16 * (lane >> 4);
    const int rr = tid >> 2, quad = tid & 3;  // gate item: row rr of the pass, units 16 ug + 4 quad .. + 3 (320 items)
    for (int row0 = rbeg; row0 < rend; row0 += 16 * RT) {
        const int row = row0 + rr;
        const bool has = tid < 64 * RT && row < rend;
        const int rowc = min(row, N - 1);
        float4 gx[4];
        const long si = (long)rowc * H + ug * 16 + 4 * quad;
        float* gp = a.gact + (long)rowc * 4 * H + ug * 16 + 4 * quad;
#pragma unroll
        for (int g = 0; g < 4; ++g) gx[g] = ld4(gp + g * H);
        const float4 cp = ld4(a.c_prev + si), hp = ld4(a.h_prev + si);
        const int len = a.lens[rowc];
        f32x4 acc[RT][4];
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[i][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (row0 != rbeg) __syncthreads();  // the previous pass's gate math has read every wave's partials
        // contraction: tile slot i works on row tile (i + rot) % RT; loads two tiles ahead, LDS round trip one tile ahead
        const unsigned v0 = (unsigned)(row0 + rsub) * (H * 4) + wave * 256 + seg * 16;
        float4 g[2][4], fr[2][4];
        auto gload = [&](int u) {
#pragma unroll
            for (int q = 0; q < 4; ++q) g[u & 1][q] = lbuf(rh, v0 + (unsigned)(16 * rec_rot<RT>(u, rot) + 4 * q) * (H * 4), 0);
        };
        auto stage = [&](int u) {
            float* w = wr + (u & 1) * R8_TILE;
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(w + 4 * q * R8_PITCH) = g[u & 1][q];
        };
        auto frags = [&](int u) {
            const float* r = rd + (u & 1) * R8_TILE;
#pragma unroll
            for (int j = 0; j < 4; ++j) fr[u & 1][j] = *reinterpret_cast<const float4*>(r + 4 * j);
        };
        gload(0);
        gload(1);
        __builtin_amdgcn_sched_barrier(0);
        stage(0);
        gload(2);
        frags(0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < RT; ++u) {
            if (u + 1 < RT) {
                stage(u + 1);
                if (u + 3 < RT) gload(u + 3);
                frags(u + 1);
            }
            if constexpr (BX) {
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    bf16x8 ah, al;
                    rec_split8(fr[u & 1][2 * m], fr[u & 1][2 * m + 1], ah, al);
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) {
                        const bf16x8 bh = rec_bits(bres[ct][2 * m]), bl = rec_bits(bres[ct][2 * m + 1]);
                        acc[u][ct] = mfma16bx(al, bh, acc[u][ct]);
                        acc[u][ct] = mfma16bx(ah, bl, acc[u][ct]);
                        acc[u][ct] = mfma16bx(ah, bh, acc[u][ct]);
                    }
                }
            } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) acc[u][ct] = mfma16(lcomp(fr[u & 1][j], e), lcomp(bres[ct][j], e), acc[u][ct]);
            }
            if (!BX && u + 1 < RT) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                if (u + 3 < RT) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // the eight K-partial tiles meet in LDS, gates (i, j) first, then (f, o): red[wave][row][36], column = 16 (gate & 1) + unit
        float4 sums[4];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (half) __syncthreads();
            {
                float* p = As + (4 * (lane >> 4)) * RP + (lane & 15);
#pragma unroll
                for (int i = 0; i < RT; ++i) {
                    float* pi = p + 16 * rec_rot<RT>(i, rot) * RP;
#pragma unroll
                    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                        for (int v = 0; v < 4; ++v) pi[v * RP + 16 * c2] = acc[i][2 * half + c2][v];
                }
            }
            __syncthreads();
            if (has) {
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) {
                    float4 s = f4zero();
#pragma unroll
                    for (int w = 0; w < 8; ++w) {  // fixed order
                        const float4 v = ld4(smem + w * R8_WAVE + rr * RP + 16 * c2 + 4 * quad);
                        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
                    }
                    sums[2 * half + c2] = s;
                }
            }
        }
        if (has) {
#pragma unroll
            for (int g = 0; g < 4; ++g) { gx[g].x += sums[g].x; gx[g].y += sums[g].y; gx[g].z += sums[g].z; gx[g].w += sums[g].w; }
            float4 c4, h4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float i = rsigmoid(at(gx[0], e)), j = rtanh(at(gx[1], e)), f = rsigmoid(at(gx[2], e) + 1.0f), o = rsigmoid(at(gx[3], e));
                const float c = f * at(cp, e) + i * j;
                const float h = o * rtanh(c);
                at(gx[0], e) = i; at(gx[1], e) = j; at(gx[2], e) = f; at(gx[3], e) = o;
                const bool active = a.t < len;
                at(c4, e) = active ? c : at(cp, e);
                at(h4, e) = active ? h : at(hp, e);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) st4(gp + g * H, gx[g]);
            st4(a.c_out + si, c4);
            st4(a.h_out + si, h4);
        }
    }
}

// Step kernel choice (vc_lstm_set_mode): 2 (default) = auto -- the register-operand recurrence kernels where H == 512 (forward:
// the four-wave 8-unit kernel up to 400 rows, the eight-wave 16-unit kernel above; backward any N), else the split form; 3: the
// same (kept for callers of round 2's first half, when auto still took the split form above 640 rows); 1: split-K GEMM + gate
// kernels; 0: round-1 fused kernels.  Measured per step at H = 512 (marginal cost inside the sequence drivers, tools/microbench.py
// lstm; N = 160 / 320 / 640 / 1280): forward 11.5 / 17.1 / 26.2 / 47.9 us against 18.4 / 22.0 / 32.6 / 55.3 us for the split form
// (four-wave kernel alone: 31.9 / 58.9 at 640 / 1280), backward 18.4 / 25.1 / 42.5 / 80.3 against 24.3 / 33.7 / 50.9 / 88.7 us (the
// step kernels alone, rocprofv3: 13.0 us forward and 11.8 us backward at N = 320, where round 2 started from a 13.4 us split-K
// GEMM + 8.2 us gate kernel backward).
static int g_lstm_mode = 2;   // DEPRECATED process-wide default (vc_lstm_set_mode): calls choose with VC_LSTM_KERNELS(k) in their flags (ABI 4)
static bool rec_ok(int N, int H) { return H == 512 && (long)(N + 96) * 4 * H * 4 < 0x7fffffffL; }
static int lstm_env(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
static bool rec8_rows(int N) { return N > 400; }  // forward: the eight-wave 16-unit kernel above (19.7 vs 17.1 us at 320 rows, 26.2 vs 31.9 at 640, 47.9 vs 58.9 at 1280)
constexpr size_t REC_PACK_BYTES = (size_t)512 * 2048 * sizeof(float);

static int rec_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 256;
        cus = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
    }
    return cus;
}

template <class K>
static int rec_lds(K kern) {  // 66 KB of dynamic LDS: above the 64 KB a kernel may use without opting in
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, REC_LDS_BYTES);
    return e == hipSuccess ? 0 : fail((int)e, "%s: hipFuncSetAttribute failed", "lstm recurrence kernel");
}

// row groups: as many as keep one workgroup per CU (UG column slices x RG <= CUs), at least 16 rows each
static int rec_row_groups(int N, int UG) {
    int rg = rec_cus() / UG;   // (two workgroups per CU measured slower: HISTORY.md section R)
    if (rg < 1) rg = 1;
    const int cap = cdiv(N, 16);
    return rg < cap ? rg : cap;
}

// bx: the split-bf16 kernels (whp must then come from the pack kernels called with bx = 1)
static bool rec_bx(int flags) { return (flags & VC_LSTM_BF16X3) || gemm_default_precision() == 1; }
// which step kernels a sequence call runs: the call's own choice (VC_LSTM_KERNELS(k) in its flags), else the deprecated process-wide default
static int seq_mode(int flags) { const int k = flags & 7; return (k >= 1 && k <= 4) ? k - 1 : g_lstm_mode; }
static int seq_gemm_flags(int flags) { return (flags & VC_LSTM_BF16X3) ? VC_GEMM_BF16X3 : 0; }

static int rec_fwd(hipStream_t st, const LstmFwdArgs& a, const float* whp, bool bx = false) {
    static int once = rec_lds(lstm_rec_fwd_kernel<5>) | rec_lds(lstm_rec_fwd_kernel<3>) | rec_lds(lstm_rec_fwd_kernel<5, true>) | rec_lds(lstm_rec_fwd_kernel<3, true>);
    if (once) return once;
    const int RG = rec_row_groups(a.N, 64), rows = cdiv(a.N, RG);
    const dim3 g(64, RG);
    if (rows > 48) {
        if (bx) hipLaunchKernelGGL((lstm_rec_fwd_kernel<5, true>), g, dim3(256), REC_LDS_BYTES, st, a, (const float4*)whp, rows);
        else hipLaunchKernelGGL(lstm_rec_fwd_kernel<5>, g, dim3(256), REC_LDS_BYTES, st, a, (const float4*)whp, rows);
    } else {
        if (bx) hipLaunchKernelGGL((lstm_rec_fwd_kernel<3, true>), g, dim3(256), REC_LDS_BYTES, st, a, (const float4*)whp, rows);
        else hipLaunchKernelGGL(lstm_rec_fwd_kernel<3>, g, dim3(256), REC_LDS_BYTES, st, a, (const float4*)whp, rows);
    }
    return launch_status("lstm rec fwd");
}

static int rec8_fwd(hipStream_t st, const LstmFwdArgs& a, const float* whp, bool bx = false) {
    static int once = [] {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_rec8_fwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, R8_LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_rec8_fwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, R8_LDS_BYTES);
        return e == hipSuccess ? 0 : fail((int)e, "%s: hipFuncSetAttribute failed", "lstm rec8 kernel");
    }();
    if (once) return once;
    const int RG = rec_row_groups(a.N, 32), rows = cdiv(a.N, RG);
    if (bx) hipLaunchKernelGGL(lstm_rec8_fwd_kernel<true>, dim3(32, RG), dim3(512), R8_LDS_BYTES, st, a, (const float4*)whp, rows);
    else hipLaunchKernelGGL(lstm_rec8_fwd_kernel<false>, dim3(32, RG), dim3(512), R8_LDS_BYTES, st, a, (const float4*)whp, rows);
    return launch_status("lstm rec8 fwd");
}

// (the LDS attribute of a variant is set the first time THAT variant launches: a process that only ever runs one of the eight
// instantiations touches one)
template <class K>
static int rec_lds_once(K kern, int& state) {
    if (state < 0) state = rec_lds(kern);
    return state;
}
#define REC_BWD_LAUNCH(RT, BX, CT, grid)                                                                                     \
    do {                                                                                                                     \
        static int st_ = -1;                                                                                                 \
        const int rc_ = rec_lds_once(lstm_rec_bwd_kernel<RT, BX, CT>, st_);                                                  \
        if (rc_) return rc_;                                                                                                 \
        hipLaunchKernelGGL((lstm_rec_bwd_kernel<RT, BX, CT>), grid, dim3(256), REC_LDS_BYTES, st, a, (const float4*)whp, rows); \
    } while (0)

static int rec_bwd(hipStream_t st, const LstmBwdArgs& a, const float* whp, bool bx = false) {
    // many rows: 32 units per workgroup (16 column slices x 16 row groups at 1280 rows: ONE pass of five row tiles per workgroup instead
    // of two, each dG[t+1] row read by 16 workgroups instead of 32).  Measured (tools/microbench.py lstm, 1280 rows): split-bf16 54.8 ->
    // 48.5 us per step, f32 81.4 -> 77.6; in the f32 steps it is small but repeatable (round 6, same box, sixteen- / 32-unit form:
    // cfg2 13.566 / 13.544 -> 13.519 / 13.515 ms, cfg3 14.935 -> 14.842), so both precisions take it from 600 rows on.
    // VC_LSTM_BWD_CT2_ROWS: from how many rows on (0: never; A/B runs).
    static const int ct2_rows = lstm_env("VC_LSTM_BWD_CT2_ROWS", 600);
    if (ct2_rows > 0 && a.N >= ct2_rows) {
        const int RG = rec_row_groups(a.N, 16), rows = cdiv(a.N, RG);
        const dim3 g(16, RG);
        if (rows > 48) {
            if (bx) REC_BWD_LAUNCH(5, true, 2, g); else REC_BWD_LAUNCH(5, false, 2, g);
        } else {
            if (bx) REC_BWD_LAUNCH(3, true, 2, g); else REC_BWD_LAUNCH(3, false, 2, g);
        }
        return launch_status("lstm rec bwd");
    }
    const int RG = rec_row_groups(a.N, 32), rows = cdiv(a.N, RG);
    const dim3 g(32, RG);
    if (rows > 48) {
        if (bx) REC_BWD_LAUNCH(5, true, 1, g); else REC_BWD_LAUNCH(5, false, 1, g);
    } else {
        if (bx) REC_BWD_LAUNCH(3, true, 1, g); else REC_BWD_LAUNCH(3, false, 1, g);
    }
    return launch_status("lstm rec bwd");
}
#undef REC_BWD_LAUNCH

using FwdCfg128 = TileCfg<4, 1, 1, 4>;  // 128 rows x (4 gates x 32 units), 256 threads
using FwdCfg64 = TileCfg<2, 1, 1, 4>;   //  64 rows,                       128 threads
using BwdCfg = TileCfg<2, 2, 1, 1>;     // 64 x 64

static int step_fwd(hipStream_t st, const LstmFwdArgs& a) {
    if (a.N > 640) {
        hipLaunchKernelGGL((lstm_step_fwd_kernel<FwdCfg128>), dim3(cdiv(a.N, 128), a.H / 32), dim3(FwdCfg128::NT),
                           FwdCfg128::SMEM_BYTES, st, a);
    } else {
        hipLaunchKernelGGL((lstm_step_fwd_kernel<FwdCfg64>), dim3(cdiv(a.N, 64), a.H / 32), dim3(FwdCfg64::NT),
                           FwdCfg64::SMEM_BYTES, st, a);
    }
    return launch_status("vc_lstm_step_fwd_f32");
}

static int step_bwd(hipStream_t st, const LstmBwdArgs& a) {
    hipLaunchKernelGGL((lstm_step_bwd_kernel<BwdCfg>), dim3(cdiv(a.N, 64), cdiv(a.H, 64)), dim3(BwdCfg::NT),
                       BwdCfg::SMEM_BYTES, st, a);
    return launch_status("vc_lstm_step_bwd_f32");
}

static bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace vc

extern "C" int vc_lstm_step_fwd_f32(void* stream, int N, int H, int t, const float* h_prev, const float* c_prev,
                                    const float* Wh, float* gact, const int32_t* lens_eff, float* c_out, float* h_out) {
    using namespace vc;
    VC_CHECK_ARG(N > 0 && H > 0 && H % 32 == 0, "H must be a positive multiple of 32");
    VC_CHECK_ARG(h_prev && c_prev && Wh && gact && lens_eff && c_out && h_out, "null pointer");
    VC_CHECK_ARG(aligned16(h_prev) && aligned16(Wh), "h_prev / Wh must be 16-byte aligned");
    LstmFwdArgs a{h_prev, c_prev, Wh, gact, lens_eff, c_out, h_out, N, H, t};
    return step_fwd((hipStream_t)stream, a);
}

extern "C" int vc_lstm_step_bwd_f32(void* stream, int N, int H, int t, int first, const float* dG_next, const float* Wh,
                                    const int32_t* lens_eff, const float* dh_ext, float* dH_run, float* dC_run,
                                    const float* act, const float* c_prev, const float* c_cur, float* dG) {
    using namespace vc;
    VC_CHECK_ARG(N > 0 && H > 0 && H % 32 == 0, "H must be a positive multiple of 32");
    VC_CHECK_ARG(Wh && lens_eff && dH_run && dC_run && act && c_prev && c_cur && dG, "null pointer");
    VC_CHECK_ARG(first || dG_next, "dG_next required unless first");
    VC_CHECK_ARG(aligned16(Wh) && (first || aligned16(dG_next)), "Wh / dG_next must be 16-byte aligned");
    LstmBwdArgs a{dG_next, Wh, lens_eff, dh_ext, dH_run, dC_run, act, c_prev, c_cur, dG, N, H, t, first};
    return step_bwd((hipStream_t)stream, a);
}

// Single steps on the recurrence kernel (generation: the decoder advances one token at a time and Wh does not change between
// steps): pack Wh once, then step.  H == 512 only (ask vc_lstm_step_packed_supported); whp: 2 * H * 4H floats (the operand
// orders of the four-wave and of the eight-wave kernel; the step picks by N).
extern "C" int vc_lstm_step_packed_supported(int N, int H) { return vc::rec_ok(N, H) ? 1 : 0; }

extern "C" int vc_lstm_pack_wh_f32(void* stream, int H, const float* Wh, float* whp) {
    using namespace vc;
    VC_CHECK_ARG(H == 512 && Wh && whp, "H == 512 required (vc_lstm_step_packed_supported)");
    VC_CHECK_ARG(aligned16(Wh) && aligned16(whp), "Wh / whp must be 16-byte aligned");
    // (single steps stay on the f32 kernels whatever the process-wide precision: a packed buffer outlives mode changes)
    hipLaunchKernelGGL(lstm_rec_pack_fwd_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, Wh, H, (float4*)whp, 0);
    hipLaunchKernelGGL(lstm_rec8_pack_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, Wh, H, (float4*)(whp + (size_t)H * 4 * H), 0);
    return launch_status(__func__);
}

extern "C" int vc_lstm_step_fwd_packed_f32(void* stream, int N, int H, int t, const float* h_prev, const float* c_prev, const float* whp,
                                           float* gact, const int32_t* lens_eff, float* c_out, float* h_out) {
    using namespace vc;
    VC_CHECK_ARG(N > 0 && rec_ok(N, H), "unsupported shape (vc_lstm_step_packed_supported)");
    VC_CHECK_ARG(h_prev && c_prev && whp && gact && lens_eff && c_out && h_out, "null pointer");
    VC_CHECK_ARG(aligned16(h_prev) && aligned16(c_prev) && aligned16(whp) && aligned16(gact) && aligned16(c_out) && aligned16(h_out),
                 "state / gate buffers must be 16-byte aligned");
    LstmFwdArgs a{h_prev, c_prev, nullptr, gact, lens_eff, c_out, h_out, N, H, t};
    return rec8_rows(N) ? rec8_fwd((hipStream_t)stream, a, whp + (size_t)H * 4 * H) : rec_fwd((hipStream_t)stream, a, whp);
}

extern "C" int vc_lstm_set_mode(int split) {
    vc::g_lstm_mode = (split < 0 || split > 3) ? 2 : split;
    return 0;
}

extern "C" size_t vc_lstm_seq_workspace_bytes(int T, int N, int E, int H) {
    size_t w = vc_gemm_workspace_bytes(E, 4 * H, T * N);
    size_t w6 = vc::gemm_partials_bytes(N, 4 * H, H, 8);
    size_t w7 = vc::gemm_partials_bytes(N, H, 4 * H, 16);
    if (w6 > w) w = w6;
    if (w7 > w) w = w7;
    size_t w2 = vc_gemm_workspace_bytes(H, 4 * H, T * N);
    size_t w3 = vc_gemm_workspace_bytes(T * N, E, 4 * H);
    size_t w4 = vc_gemm_workspace_bytes(T * N, 4 * H, E);
    size_t w5 = vc_colsum_workspace_bytes(T * N, 4 * H);
    size_t m = w;
    const size_t wp = (size_t)H * 4 * H * sizeof(float);  // Wh in operand order for the recurrence kernels
    if (wp > m) m = wp;
    if (w2 > m) m = w2;
    if (w3 > m) m = w3;
    if (w4 > m) m = w4;
    if (w5 > m) m = w5;
    return m;
}

// Whole sequence forward: act = X.Wx + b for all T steps (one GEMM), then T fused steps.
// cs[0], hs[0] must hold the initial state (zeros for the reference's zero_state).
extern "C" int vc_lstm_seq_fwd_f32(void* stream, int T, int N, int E, int H, const float* X, const float* W,
                                   const float* b, const int32_t* lens_eff, float* act, float* cs, float* hs, float* ws,
                                   size_t ws_bytes, int flags) {
    using namespace vc;
    const int mode = seq_mode(flags), gf = seq_gemm_flags(flags);
    VC_CHECK_ARG(T > 0 && N > 0 && E > 0 && H > 0 && H % 32 == 0, "bad dimensions (H % 32 == 0 required)");
    VC_CHECK_ARG(X && W && b && lens_eff && act && cs && hs, "null pointer");
    const float* Wx = W;
    const float* Wh = W + (long)E * 4 * H;
    int rc = vc_gemm_f32(stream, 0, 0, T * N, 4 * H, E, X, E, Wx, 4 * H, act, 4 * H, b, gf, ws, ws_bytes);
    if (rc) return rc;
    const long NH = (long)N * H;
    const int eg = (int)((NH + 255) / 256) < 2048 ? (int)((NH + 255) / 256) : 2048;
    const bool rec = (mode == 3 || mode == 2) && rec_ok(N, H) && ws && ws_bytes >= REC_PACK_BYTES;
    const bool rec8 = rec && rec8_rows(N);
    const bool bx = rec && rec_bx(flags);   // split-bf16 recurrence (VC_LSTM_BF16X3)
    if (rec) {  // Wh in the MFMA-operand layout of the step kernel, once per sequence (4 MB at H = 512)
        if (rec8) hipLaunchKernelGGL(lstm_rec8_pack_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, Wh, H, (float4*)ws, bx ? 1 : 0);
        else hipLaunchKernelGGL(lstm_rec_pack_fwd_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, Wh, H, (float4*)ws, bx ? 1 : 0);
        rc = launch_status(__func__);
        if (rc) return rc;
    }
    for (int t = 0; t < T; ++t) {
        float* g = act + (long)t * N * 4 * H;
        if (rec) {
            LstmFwdArgs a{hs + t * NH, cs + t * NH, Wh, g, lens_eff, cs + (t + 1) * NH, hs + (t + 1) * NH, N, H, t};
            rc = rec8 ? rec8_fwd((hipStream_t)stream, a, ws, bx) : rec_fwd((hipStream_t)stream, a, ws, bx);
        } else if (mode) {
            int ns = 1;  // recurrent product as split-K partials in ws; the gate kernel sums them (no separate reduce launch)
            rc = gemm_partials_f32((hipStream_t)stream, 0, 0, N, 4 * H, H, hs + t * NH, H, Wh, 4 * H, ws, ws_bytes, 8, &ns);
            if (rc) return rc;
            hipLaunchKernelGGL(lstm_gates_fwd_kernel, dim3(eg), dim3(256), 0, (hipStream_t)stream, g, cs + t * NH, hs + t * NH, lens_eff,
                               cs + (t + 1) * NH, hs + (t + 1) * NH, N, H, t, ws, ns);
            rc = launch_status(__func__);
        } else {
            rc = vc_lstm_step_fwd_f32(stream, N, H, t, hs + t * NH, cs + t * NH, Wh, g, lens_eff, cs + (t + 1) * NH, hs + (t + 1) * NH);
        }
        if (rc) return rc;
    }
    return 0;
}

// Whole sequence backward.  dhs_ext: [T+1,N,H] external gradient w.r.t. every state hs[t]
// (may be null); dH_run / dC_run: [N,H] scratch that must hold the gradient w.r.t. the
// final state on entry (zeros, or the encoder's d h_T) and holds d(hs[1]), d(cs[0]) on exit.
// Outputs: dG [T,N,4H], dX [T,N,E], dW [E+H,4H], db [4H].
// The same pass in two calls, for callers that keep the weight gradients off the chain that the data gradient feeds (they have
// no consumer before the optimiser, so they may run on another stream under whatever follows dX):
//   vc_lstm_seq_bwd_data_f32     the recurrence (dG for every step) and dX = dG.Wx^T
//   vc_lstm_seq_bwd_weights_f32  dWx = X^T.dG, dWh = hs[0:T]^T.dG, db = colsum(dG), from the dG the first call left
// vc_lstm_seq_bwd_f32 is the first followed by the second on one stream.
extern "C" int vc_lstm_seq_bwd_weights_f32(void* stream, int T, int N, int E, int H, const float* X, const float* hs, const float* dG,
                                           float* dW, float* db, float* ws, size_t ws_bytes, int flags) {
    VC_CHECK_ARG(T > 0 && N > 0 && E > 0 && H > 0 && H % 32 == 0, "bad dimensions (H % 32 == 0 required)");
    VC_CHECK_ARG(X && hs && dG && dW && db, "null pointer");
    const int gf = vc::seq_gemm_flags(flags);
    int rc = vc_gemm_f32(stream, 1, 0, E, 4 * H, T * N, X, E, dG, 4 * H, dW, 4 * H, nullptr, gf, ws, ws_bytes);
    if (rc) return rc;
    rc = vc_gemm_f32(stream, 1, 0, H, 4 * H, T * N, hs, H, dG, 4 * H, dW + (long)E * 4 * H, 4 * H, nullptr, gf, ws, ws_bytes);
    if (rc) return rc;
    return vc_colsum_f32(stream, dG, T * N, 4 * H, 4 * H, db, 0, ws, ws_bytes);
}

extern "C" int vc_lstm_seq_bwd_f32(void* stream, int T, int N, int E, int H, const float* X, const float* W,
                                   const int32_t* lens_eff, const float* act, const float* cs, const float* hs,
                                   const float* dhs_ext, float* dH_run, float* dC_run, float* dG, float* dX, float* dW,
                                   float* db, float* ws, size_t ws_bytes, int flags) {
    VC_CHECK_ARG(dW && db, "null pointer");
    int rc = vc_lstm_seq_bwd_data_f32(stream, T, N, E, H, W, lens_eff, act, cs, dhs_ext, dH_run, dC_run, dG, dX, ws, ws_bytes, flags);
    if (rc) return rc;
    return vc_lstm_seq_bwd_weights_f32(stream, T, N, E, H, X, hs, dG, dW, db, ws, ws_bytes, flags);
}

extern "C" int vc_lstm_seq_bwd_data_f32(void* stream, int T, int N, int E, int H, const float* W, const int32_t* lens_eff,
                                        const float* act, const float* cs, const float* dhs_ext, float* dH_run, float* dC_run,
                                        float* dG, float* dX, float* ws, size_t ws_bytes, int flags) {
    using namespace vc;
    const int mode = seq_mode(flags), gf = seq_gemm_flags(flags);
    VC_CHECK_ARG(T > 0 && N > 0 && E > 0 && H > 0 && H % 32 == 0, "bad dimensions (H % 32 == 0 required)");
    VC_CHECK_ARG(W && lens_eff && act && cs && dH_run && dC_run && dG && dX, "null pointer");
    const float* Wx = W;
    const float* Wh = W + (long)E * 4 * H;
    const long NH = (long)N * H, NG = (long)N * 4 * H;
    int rc;
    const int eg = (int)((NH + 255) / 256) < 2048 ? (int)((NH + 255) / 256) : 2048;
    // split form: the recurrent product dG[t+1].Wh^T as split-K partials in ws, summed by the gate kernel
    const float* rec = ws;
    const bool recb = (mode == 3 || mode == 2) && rec_ok(N, H) && ws && ws_bytes >= REC_PACK_BYTES;
    const bool bx = recb && rec_bx(flags);
    if (recb) {  // Wh^T slices in operand order; ws is free again for the GEMMs below once the step loop has run
        hipLaunchKernelGGL(lstm_rec_pack_bwd_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, Wh, H, (float4*)ws, bx ? 1 : 0);
        rc = launch_status(__func__);
        if (rc) return rc;
    }
    for (int t = T - 1; t >= 0; --t) {
        const int first = (t == T - 1);
        const float* ext = dhs_ext ? dhs_ext + (t + 1) * NH : nullptr;
        if (recb) {
            LstmBwdArgs a{first ? nullptr : dG + (t + 1) * NG, Wh, lens_eff, ext, dH_run, dC_run, act + t * NG, cs + t * NH, cs + (t + 1) * NH,
                          dG + t * NG, N, H, t, first};
            rc = rec_bwd((hipStream_t)stream, a, ws, bx);
        } else if (mode) {
            int ns = 1;
            if (!first) {
                rc = gemm_partials_f32((hipStream_t)stream, 0, 1, N, H, 4 * H, dG + (t + 1) * NG, 4 * H, Wh, 4 * H, ws, ws_bytes, 16, &ns);
                if (rc) return rc;
            }
            hipLaunchKernelGGL(lstm_gates_bwd_kernel, dim3(eg), dim3(256), 0, (hipStream_t)stream, rec, lens_eff, ext, dH_run, dC_run,
                               act + t * NG, cs + t * NH, cs + (t + 1) * NH, dG + t * NG, N, H, t, first, ns);
            rc = launch_status(__func__);
        } else {
            rc = vc_lstm_step_bwd_f32(stream, N, H, t, first, first ? nullptr : dG + (t + 1) * NG, Wh, lens_eff, ext, dH_run, dC_run,
                                      act + t * NG, cs + t * NH, cs + (t + 1) * NH, dG + t * NG);
        }
        if (rc) return rc;
    }
    return vc_gemm_f32(stream, 0, 1, T * N, E, 4 * H, dG, 4 * H, Wx, 4 * H, dX, E, nullptr, gf, ws, ws_bytes);
}
