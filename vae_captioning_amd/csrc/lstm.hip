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

// ---- wide fused step kernels (default for H % 512 == 0) ----------------------------------------------------------------
// One launch per time step, no LDS staging and no barrier in the contraction: a workgroup of EIGHT waves owns a
// (TM*32 rows) x (16 hidden units x 4 gates) block of the step and splits K eight ways; every wave loads its A fragments
// (rows of h_{t-1}, 64 contiguous bytes per lane) and B fragments (Wh pre-packed [k/4][4H][4], so that a lane's four
// consecutive k of one gate column are ONE 16-byte load) straight into registers -- all of a wave's loads are in flight
// before its first MFMA -- and the eight partial tiles meet in LDS, where the gate math runs on the (row, unit) pairs.
// 320 workgroups at N = 320 (the split form needed a split-K GEMM launch + a gate launch per step: 23 / 35 us per
// forward / backward step; the round-1 fused kernel had 50 workgroups of four waves marching through K = 512 serially).
typedef unsigned int lstm_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 lbuf(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    lstm_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
    return *reinterpret_cast<float4*>(&v);
}
__device__ __forceinline__ float lcomp(const float4& v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; }

// Wh [H, 4H] row-major -> [H/4][4H][4]
__global__ __launch_bounds__(256) void lstm_pack_wh_kernel(const float* __restrict__ Wh, int H, float4* __restrict__ out) {
    const long total = (long)(H >> 2) * 4 * H;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int col = (int)(i % (4 * H));
        const long kq = i / (4 * H);
        const float* s = Wh + kq * 4 * 4 * H + col;
        out[i] = make_float4(s[0], s[4L * H], s[8L * H], s[12L * H]);
    }
}

template <int TM>
__global__ __launch_bounds__(512, 1) void lstm_wide_fwd_kernel(LstmFwdArgs a, const float* __restrict__ whp) {
    extern __shared__ __attribute__((aligned(16))) float smem[];  // [8 waves][2][32 rows][32 cols] = 64 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int m0 = blockIdx.x * (TM * 32), u0 = blockIdx.y * 16;
    const int H = a.H, N = a.N;
    const int KW = H >> 3;          // K range of a wave
    const int kb0 = wave * KW;
    const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc((void*)a.h_prev, 0, N * H * 4, 0x00020000);  // rows >= N read as 0
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)whp, 0, H * 4 * H * 4, 0x00020000);
    // B: column of lane li in tile tn = gate (2 tn + (li >> 4)), unit u0 + (li & 15); k quad (kb0 + 32 b + 16 lh + 4 q) / 4
    unsigned vb[2];
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) vb[tn] = (unsigned)((((long)(kb0 >> 2) + 4 * lh) * 4 * H + (2 * tn + (li >> 4)) * H + u0 + (li & 15)) * 16);
    unsigned va[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t) va[t] = (unsigned)(((long)(m0 + t * 32 + li) * H + kb0 + 16 * lh) * 4);
    constexpr int NB = 2;  // 32-deep K blocks per wave at H = 512 (KW = 64); general: KW / 32, handled by the loop below
    f32x16 acc[TM][2];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;
    for (int kb = 0; kb < KW; kb += 32 * NB) {
        float4 fa[NB][TM][4], fb[NB][2][4];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) fa[b][t][q] = lbuf(rh, va[t], (unsigned)((kb + 32 * b + 4 * q) * 4));
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int q = 0; q < 4; ++q) fb[b][tn][q] = lbuf(rw, vb[tn], (unsigned)(((kb + 32 * b) / 4 + q) * 4 * H * 16));
        }
        __builtin_amdgcn_sched_barrier(0);  // every load of the K range is in flight before the first MFMA
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int t = 0; t < TM; ++t)
#pragma unroll
                        for (int tn = 0; tn < 2; ++tn)
                            acc[t][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(lcomp(fa[b][t][q], e), lcomp(fb[b][tn][q], e), acc[t][tn], 0, 0, 0);
    }
    // partial tiles -> LDS, one 32-row tile at a time (64 KB): red[wave][tn][row][col]; then the gate math on its (row, unit) pairs
    float* red = smem + wave * 2048;
#pragma unroll
    for (int t = 0; t < TM; ++t) {
    if (t) __syncthreads();
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[tn * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + li] = acc[t][tn][r];
    __syncthreads();
    {
        const int p = tid;  // 32 rows x 16 units = 512 pairs = one per thread
        const int r32 = p >> 4, un = p & 15;
        const int row = m0 + t * 32 + r32;
        if (row >= N) continue;
        float g4[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) sum += smem[w * 2048 + (g >> 1) * 1024 + r32 * 32 + (g & 1) * 16 + un];  // fixed order
            g4[g] = sum;
        }
        const int u = u0 + un;
        float* gp = a.gact + (long)row * 4 * H + u;
        const float gi = g4[0] + gp[0], gj = g4[1] + gp[H], gf = g4[2] + gp[2 * H], go = g4[3] + gp[3 * H];
        const float i = sigmoidf_(gi), j = tanhf(gj), f = sigmoidf_(gf + 1.0f), o = sigmoidf_(go);
        const long si = (long)row * H + u;
        const float cp = a.c_prev[si];
        const float c = f * cp + i * j;
        const float h = o * tanhf(c);
        gp[0] = i; gp[H] = j; gp[2 * H] = f; gp[3 * H] = o;
        const bool active = a.t < a.lens[row];
        a.c_out[si] = active ? c : cp;
        a.h_out[si] = active ? h : a.h_prev[si];
    }
    }
}

// backward step: (dG[t+1] . Wh^T)[row, u] = sum_k dG[row, k] Wh[u, k], k over the 4H gate columns: both operands are
// K-contiguous in memory (no packing); block = TM*32 rows x 32 units, eight waves split K = 4H, register ring of two
// 64-deep K ranges per wave (the next range's loads fly under the current range's 32*TM MFMAs).
template <int TM>
__global__ __launch_bounds__(512, 1) void lstm_wide_bwd_kernel(LstmBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];  // [8 waves][TM][32][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int m0 = blockIdx.x * (TM * 32), n0 = blockIdx.y * 32;
    const int H = a.H, N = a.N, K = 4 * H;
    const int KW = K >> 3, kb0 = wave * KW;
    f32x16 acc[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    if (!a.first) {
        const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)a.dG_next, 0, N * K * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.Wh, 0, H * K * 4, 0x00020000);
        unsigned va[TM];
#pragma unroll
        for (int t = 0; t < TM; ++t) va[t] = (unsigned)(((long)(m0 + t * 32 + li) * K + kb0 + 16 * lh) * 4);
        const unsigned vb = (unsigned)(((long)(n0 + li) * K + kb0 + 16 * lh) * 4);
        float4 fa[2][2][TM][4], fb[2][2][4];  // [ring slot][32-deep block][..]
        auto issue = [&](int slot, int kb) {
#pragma unroll
            for (int b = 0; b < 2; ++b) {
#pragma unroll
                for (int t = 0; t < TM; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) fa[slot][b][t][q] = lbuf(rg, va[t], (unsigned)((kb + 32 * b + 4 * q) * 4));
#pragma unroll
                for (int q = 0; q < 4; ++q) fb[slot][b][q] = lbuf(rw, vb, (unsigned)((kb + 32 * b + 4 * q) * 4));
            }
        };
        auto compute = [&](int slot) {
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int t = 0; t < TM; ++t)
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(lcomp(fa[slot][b][t][q], e), lcomp(fb[slot][b][q], e), acc[t], 0, 0, 0);
        };
        issue(0, 0);
        if (KW > 64) issue(1, 64);
        __builtin_amdgcn_sched_barrier(0);
        for (int kb = 0; kb < KW; kb += 128) {
            compute(0);
            if (kb + 128 < KW) issue(0, kb + 128);
            __builtin_amdgcn_sched_barrier(0);
            if (kb + 64 < KW) compute(1);
            if (kb + 192 < KW) issue(1, kb + 192);
            __builtin_amdgcn_sched_barrier(0);
        }
        float* red = smem + wave * (TM * 1024);
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[t * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + li] = acc[t][r];
    }
    __syncthreads();
    for (int p = tid; p < TM * 32 * 32; p += 512) {
        const int rr = p >> 5, un = p & 31;
        const int row = m0 + rr, u = n0 + un;
        if (row >= N || u >= H) continue;
        const long si = (long)row * H + u;
        const int len = a.lens[row];
        float dh = a.dH_run[si];
        if (!a.first && (a.t + 1 < len)) {  // step t+1 was active: its recurrent gradient replaces the carried one
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) sum += smem[w * (TM * 1024) + (rr >> 5) * 1024 + (rr & 31) * 32 + un];
            dh = sum;
        }
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

// 2 (default) = auto: wide fused FORWARD step kernel for N <= 640 rows where H % 512 == 0, else the split form; 3: wide kernels
// wherever supported (forward and backward); 1: split-K GEMM + gate kernels; 0: round-1 fused kernels
static int g_lstm_mode = 2;
static bool wide_ok(int N, int H) { return H % 512 == 0 && (long)N * 4 * H * 4 < 0x7fffffffL; }  // (a wave's K range = H/8 = whole 64-deep blocks)

static int wide_fwd(hipStream_t st, const LstmFwdArgs& a, const float* whp) {
    if (a.N > 640)
        hipLaunchKernelGGL(lstm_wide_fwd_kernel<2>, dim3(cdiv(a.N, 64), a.H / 16), dim3(512), 8 * 2 * 1024 * 4, st, a, whp);
    else
        hipLaunchKernelGGL(lstm_wide_fwd_kernel<1>, dim3(cdiv(a.N, 32), a.H / 16), dim3(512), 8 * 2 * 1024 * 4, st, a, whp);
    return launch_status("lstm wide fwd");
}

static int wide_bwd(hipStream_t st, const LstmBwdArgs& a) {
    if (a.N > 640)
        hipLaunchKernelGGL(lstm_wide_bwd_kernel<2>, dim3(cdiv(a.N, 64), a.H / 32), dim3(512), 8 * 2 * 1024 * 4, st, a);
    else
        hipLaunchKernelGGL(lstm_wide_bwd_kernel<1>, dim3(cdiv(a.N, 32), a.H / 32), dim3(512), 8 * 1024 * 4, st, a);
    return launch_status("lstm wide bwd");
}


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
    const size_t wp = (size_t)H * 4 * H * sizeof(float);  // packed Wh of the wide fused step kernels
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
                                   size_t ws_bytes) {
    using namespace vc;
    VC_CHECK_ARG(T > 0 && N > 0 && E > 0 && H > 0 && H % 32 == 0, "bad dimensions (H % 32 == 0 required)");
    VC_CHECK_ARG(X && W && b && lens_eff && act && cs && hs, "null pointer");
    const float* Wx = W;
    const float* Wh = W + (long)E * 4 * H;
    int rc = vc_gemm_f32(stream, 0, 0, T * N, 4 * H, E, X, E, Wx, 4 * H, act, 4 * H, b, 0, ws, ws_bytes);
    if (rc) return rc;
    const long NH = (long)N * H;
    const int eg = (int)((NH + 255) / 256) < 2048 ? (int)((NH + 255) / 256) : 2048;
    // auto (2): the wide forward kernel up to 640 rows (measured, H = 512: N = 160 8.6 us vs 18.6 us per step for GEMM + gates,
    // N = 320 17 vs 23 us; at N = 1280 both ~55 us: every workgroup re-reads its 128 KB slice of Wh from L2 each step)
    const bool wide = (g_lstm_mode == 3 || (g_lstm_mode == 2 && N <= 640)) && wide_ok(N, H) && ws && ws_bytes >= (size_t)H * 4 * H * sizeof(float);
    if (wide) {  // Wh in the MFMA-operand layout, once per sequence (4 MB at H = 512)
        hipLaunchKernelGGL(lstm_pack_wh_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, Wh, H, (float4*)ws);
        rc = launch_status(__func__);
        if (rc) return rc;
    }
    for (int t = 0; t < T; ++t) {
        float* g = act + (long)t * N * 4 * H;
        if (wide) {
            LstmFwdArgs a{hs + t * NH, cs + t * NH, Wh, g, lens_eff, cs + (t + 1) * NH, hs + (t + 1) * NH, N, H, t};
            rc = wide_fwd((hipStream_t)stream, a, ws);
        } else if (g_lstm_mode) {
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
extern "C" int vc_lstm_seq_bwd_f32(void* stream, int T, int N, int E, int H, const float* X, const float* W,
                                   const int32_t* lens_eff, const float* act, const float* cs, const float* hs,
                                   const float* dhs_ext, float* dH_run, float* dC_run, float* dG, float* dX, float* dW,
                                   float* db, float* ws, size_t ws_bytes) {
    using namespace vc;
    VC_CHECK_ARG(T > 0 && N > 0 && E > 0 && H > 0 && H % 32 == 0, "bad dimensions (H % 32 == 0 required)");
    VC_CHECK_ARG(X && W && lens_eff && act && cs && hs && dH_run && dC_run && dG && dX && dW && db, "null pointer");
    const float* Wx = W;
    const float* Wh = W + (long)E * 4 * H;
    const long NH = (long)N * H, NG = (long)N * 4 * H;
    int rc;
    const int eg = (int)((NH + 255) / 256) < 2048 ? (int)((NH + 255) / 256) : 2048;
    // split form: the recurrent product dG[t+1].Wh^T as split-K partials in ws, summed by the gate kernel
    const float* rec = ws;
    for (int t = T - 1; t >= 0; --t) {
        const int first = (t == T - 1);
        const float* ext = dhs_ext ? dhs_ext + (t + 1) * NH : nullptr;
        if (g_lstm_mode == 3 && wide_ok(N, H)) {  // (the wide backward kernel is not faster than GEMM + gates: 22.8 vs 21 us at N = 320, 71 vs 45 us at N = 1280)
            LstmBwdArgs a{first ? nullptr : dG + (t + 1) * NG, Wh, lens_eff, ext, dH_run, dC_run, act + t * NG, cs + t * NH, cs + (t + 1) * NH,
                          dG + t * NG, N, H, t, first};
            rc = wide_bwd((hipStream_t)stream, a);
        } else if (g_lstm_mode) {
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
    // dWx = X^T.dG, dWh = hs[0:T]^T.dG, db = colsum(dG), dX = dG.Wx^T
    rc = vc_gemm_f32(stream, 1, 0, E, 4 * H, T * N, X, E, dG, 4 * H, dW, 4 * H, nullptr, 0, ws, ws_bytes);
    if (rc) return rc;
    rc = vc_gemm_f32(stream, 1, 0, H, 4 * H, T * N, hs, H, dG, 4 * H, dW + (long)E * 4 * H, 4 * H, nullptr, 0, ws, ws_bytes);
    if (rc) return rc;
    rc = vc_colsum_f32(stream, dG, T * N, 4 * H, 4 * H, db, 0, ws, ws_bytes);
    if (rc) return rc;
    return vc_gemm_f32(stream, 0, 1, T * N, E, 4 * H, dG, 4 * H, Wx, 4 * H, dX, E, nullptr, 0, ws, ws_bytes);
}
