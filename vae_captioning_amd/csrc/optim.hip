// Optimiser-side kernels (ops/optimizers.py): global-norm clip, Adam / SGD / Momentum
// over flat parameter buffers, device-resident step scalars (so a captured hipGraph can be
// replayed without host-side argument changes), and the Philox4x32-10 generator that
// replaces TF's random_normal / dropout streams.
//
// HBM-bound: Adam touches 4 reads + 3 writes x 4 B = 28 B per parameter.
#include "common.h"
#include <stdlib.h>
#include "vaecap.h"

namespace vc {

static inline int grid_for(long work_items, int per_block = 256, int cap = 2048) {
    long b = (work_items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (int)b;
}

constexpr int SUMSQ_BLOCKS = 512;

// Workgroups of an Adam launch (grid-stride, all resident at once; 256 / 512 / 1024 measured no better: HISTORY.md section R)
static inline int adam_grid(long n) { return grid_for(n / 4 + 1, 256, 2048); }

// partial[b] = sum over a fixed grid-stride slice of x^2: deterministic for a fixed n.
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ x, long n, float* __restrict__ partial) {
    __shared__ float sh[4];
    float s = 0.f;
    const long n4 = n >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 v = x4[i];
        s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    for (long i = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) s += x[i] * x[i];
    s = block_sum<256>(s, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// tf.clip_by_global_norm (ops/optimizers.py:15-16; TF-sem.): norm = sqrt(sum), scale =
// clip * min(1/norm, 1/clip).  out[0] = norm, out[1] = scale.
__global__ __launch_bounds__(256) void clip_finalize_kernel(const float* __restrict__ partial, int np,
                                                            float clip, float* __restrict__ out) {
    __shared__ float sh[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < np; i += 256) s += partial[i];
    s = block_sum<256>(s, sh);
    if (threadIdx.x == 0) {
        const float norm = sqrtf(s);
        out[0] = norm;
        out[1] = norm > 0.f ? clip * fminf(1.f / norm, 1.f / clip) : 1.f;
    }
}

// Device-resident step bookkeeping.  step[0] = global_step BEFORE this step (the value the
// reference feeds to `anneal`, main.py:237-238).  Writes
//   s[0] Adam lr_t (non-CNN)   = lr * sqrt(1-b2^t)/(1-b1^t), t = step+1   (TF-sem.)
//   s[1] annealing coefficient = (tanh((step - 1000*ann_param)/1000)+1)/2 or 1 (main.py:163-170)
//   s[2] staircase-decayed lr  = lr * 0.5^floor(step/decay_steps)        (ops/optimizers.py:24-31)
//   s[3] Adam lr_t (CNN), s[4] staircase-decayed CNN lr
// then increments step.
__global__ void step_update_kernel(int32_t* step, float* s, float lr, float cnn_lr, float beta1, float beta2,
                                   float ann_param, int ann_on, int decay_steps) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int gs = step[0];
    const float t = (float)(gs + 1);
    const float corr = sqrtf(1.f - powf(beta2, t)) / (1.f - powf(beta1, t));
    s[0] = lr * corr;
    s[1] = ann_on ? (tanhf(((float)gs - 1000.f * ann_param) / 1000.f) + 1.f) * 0.5f : 1.f;
    const float dec = powf(0.5f, (float)(decay_steps > 0 ? gs / decay_steps : 0));
    s[2] = lr * dec;
    s[3] = cnn_lr * corr;
    s[4] = cnn_lr * dec;
    step[0] = gs + 1;
}

// tf.train.AdamOptimizer.apply_gradients (ops/optimizers.py:37-40,72-75; TF-sem.):
//   g' = g*scale (+ l2*p);  m = b1*m + (1-b1)*g';  v = b2*v + (1-b2)*g'^2;  p -= lr_t*m/(sqrt(v)+eps)
// SUMSQ: also leaves sum(p_new^2) of the block in sumsq_partial[blockIdx.x] -- the regulariser's loss term of the NEXT step
// (main.py:69-74 sums w^2 over the cnn/* variables) without another 0.5 GB pass over the parameters.
template <bool SUMSQ>
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long n, const float* __restrict__ lr_t,
                                                   const float* __restrict__ scale, float beta1, float beta2, float eps,
                                                   float l2, float* __restrict__ sumsq_partial) {
    __shared__ float sh[4];
    float s2 = 0.f;
    const float lr = lr_t[0];
    const float sc = scale ? scale[0] : 1.f;
    const long n4 = n >> 2;
    float4* p4 = reinterpret_cast<float4*>(p);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
    auto upd = [&](float& pp, float gg, float& mm, float& vv) {
        gg = gg * sc + l2 * pp;
        mm = beta1 * mm + (1.f - beta1) * gg;
        vv = beta2 * vv + (1.f - beta2) * gg * gg;
        pp -= lr * mm / (sqrtf(vv) + eps);
        if (SUMSQ) s2 += pp * pp;
    };
    // two independent 16-byte groups per thread and iteration: eight loads in flight before the first use (HBM-bound: 28 B / parameter)
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + stride < n4; i += 2 * stride) {
        const long j = i + stride;
        float4 pa = p4[i], ma = m4[i], va = v4[i];
        const float4 ga = g4[i];
        float4 pb = p4[j], mb = m4[j], vb = v4[j];
        const float4 gb = g4[j];
        upd(pa.x, ga.x, ma.x, va.x); upd(pa.y, ga.y, ma.y, va.y);
        upd(pa.z, ga.z, ma.z, va.z); upd(pa.w, ga.w, ma.w, va.w);
        upd(pb.x, gb.x, mb.x, vb.x); upd(pb.y, gb.y, mb.y, vb.y);
        upd(pb.z, gb.z, mb.z, vb.z); upd(pb.w, gb.w, mb.w, vb.w);
        p4[i] = pa; m4[i] = ma; v4[i] = va;
        p4[j] = pb; m4[j] = mb; v4[j] = vb;
    }
    if (i < n4) {
        float4 pp = p4[i], mm = m4[i], vv = v4[i];
        const float4 gg = g4[i];
        upd(pp.x, gg.x, mm.x, vv.x); upd(pp.y, gg.y, mm.y, vv.y);
        upd(pp.z, gg.z, mm.z, vv.z); upd(pp.w, gg.w, mm.w, vv.w);
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
    }
    for (long i = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) upd(p[i], g[i], m[i], v[i]);
    if (SUMSQ) {
        s2 = block_sum<256>(s2, sh);
        if (threadIdx.x == 0) sumsq_partial[blockIdx.x] = s2;
    }
}

// GradientDescentOptimizer: p -= lr * (g*scale + l2*p)
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, long n,
                                                  const float* __restrict__ lr, const float* __restrict__ scale, float l2) {
    const float a = lr[0], sc = scale ? scale[0] : 1.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) p[i] -= a * (g[i] * sc + l2 * p[i]);
}

// MomentumOptimizer (TF-sem.): a = mom*a + g'; p -= lr*a.  row_mask (per row of width E):
// the sparse variant only touches rows present in the batch's indices.
__global__ __launch_bounds__(256) void momentum_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ a,
                                                       long n, const float* __restrict__ lr, const float* __restrict__ scale,
                                                       float mom, float l2, const float* __restrict__ row_mask, int E) {
    const float al = lr[0], sc = scale ? scale[0] : 1.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        if (row_mask && row_mask[i / E] == 0.f) continue;
        const float gg = g[i] * sc + l2 * p[i];
        const float aa = mom * a[i] + gg;
        a[i] = aa;
        p[i] -= al * aa;
    }
}

// ---- Philox4x32-10 (Salmon et al. 2011), counter = (quad index lo, hi, offset lo, hi) -----
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t (&o)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

// mode 0: raw uint32; 1: N(0,1) via Box-Muller; 2: Bernoulli(keep) as 0/1 floats; 3: uniform [0,1)
// `step` (nullable, device): added to the high KEY word so that a captured graph draws a fresh
// stream every replay without any host-side argument change.  (Counter words stay (element quad,
// offset): callers put a stream id into offset >> 32, so step must not share a word with it --
// stream k at step s+1 would repeat stream k+1 at step s.)
__global__ __launch_bounds__(256) void philox_kernel(void* __restrict__ out, long n, uint64_t seed, uint64_t offset, int mode,
                                                     float keep, const int32_t* __restrict__ step) {
    const long nq = (n + 3) >> 2;
    const uint32_t k1 = (uint32_t)(seed >> 32) + (step ? (uint32_t)step[0] : 0u);
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < nq; q += (long)gridDim.x * 256) {
        uint32_t r[4];
        philox4x32_10((uint32_t)q, (uint32_t)((uint64_t)q >> 32), (uint32_t)offset, (uint32_t)(offset >> 32), (uint32_t)seed, k1, r);
        float f[4];
        if (mode == 1) {
            const float u1 = ((float)r[0] + 1.0f) * 2.3283064365386963e-10f;  // (0,1]
            const float u2 = (float)r[1] * 2.3283064365386963e-10f;
            const float u3 = ((float)r[2] + 1.0f) * 2.3283064365386963e-10f;
            const float u4 = (float)r[3] * 2.3283064365386963e-10f;
            const float ra = sqrtf(-2.f * __logf(u1)), rb = sqrtf(-2.f * __logf(u3));
            float s, c;
            __sincosf(6.283185307179586f * u2, &s, &c);
            f[0] = ra * c; f[1] = ra * s;
            __sincosf(6.283185307179586f * u4, &s, &c);
            f[2] = rb * c; f[3] = rb * s;
        } else if (mode == 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) f[j] = ((float)r[j] * 2.3283064365386963e-10f < keep) ? 1.f : 0.f;
        } else if (mode == 3) {
#pragma unroll
            for (int j = 0; j < 4; ++j) f[j] = (float)(r[j] >> 8) * 5.9604644775390625e-08f;  // 24 bits -> [0, 1)
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long i = q * 4 + j;
            if (i < n) {
                if (mode == 0) reinterpret_cast<uint32_t*>(out)[i] = r[j];
                else reinterpret_cast<float*>(out)[i] = f[j];
            }
        }
    }
}

}  // namespace vc

using namespace vc;

extern "C" int vc_sumsq_blocks(void) { return SUMSQ_BLOCKS; }

extern "C" int vc_sumsq_partial_f32(void* stream, const float* x, long n, float* partial) {
    VC_CHECK_ARG(x && partial && n >= 0 && (((uintptr_t)x & 15) == 0), "bad argument (x must be 16-byte aligned)");
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(SUMSQ_BLOCKS), dim3(256), 0, (hipStream_t)stream, x, n, partial);
    VC_LAUNCH_CHECK();
    return 0;
}

extern "C" int vc_clip_finalize_f32(void* stream, const float* partial, int n_partial, float clip, float* out_norm_scale) {
    VC_CHECK_ARG(partial && out_norm_scale && n_partial > 0 && clip > 0.f, "bad argument");
    hipLaunchKernelGGL(clip_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, n_partial, clip, out_norm_scale);
    VC_LAUNCH_CHECK();
    return 0;
}

extern "C" int vc_step_update(void* stream, int32_t* step, float* scalars, float lr, float cnn_lr, float beta1, float beta2,
                              float ann_param, int ann_on, int decay_steps) {
    VC_CHECK_ARG(step && scalars, "null pointer");
    hipLaunchKernelGGL(step_update_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, step, scalars, lr, cnn_lr, beta1, beta2, ann_param, ann_on, decay_steps);
    VC_LAUNCH_CHECK();
    return 0;
}

extern "C" int vc_adam_f32(void* stream, float* p, const float* g, float* m, float* v, long n, const float* lr_t,
                           const float* scale, float beta1, float beta2, float eps, float l2) {
    VC_CHECK_ARG(p && g && m && v && lr_t && n >= 0, "bad argument");
    VC_CHECK_ARG((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "buffers must be 16-byte aligned");
    if (n == 0) return 0;
    hipLaunchKernelGGL(adam_kernel<false>, dim3(adam_grid(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr_t, scale, beta1, beta2, eps, l2, nullptr);
    VC_LAUNCH_CHECK();
    return 0;
}

extern "C" int vc_adam_blocks(long n) { return adam_grid(n); }

extern "C" int vc_adam_sumsq_f32(void* stream, float* p, const float* g, float* m, float* v, long n, const float* lr_t,
                                 const float* scale, float beta1, float beta2, float eps, float l2, float* sumsq_partial) {
    VC_CHECK_ARG(p && g && m && v && lr_t && sumsq_partial && n > 0, "bad argument");
    VC_CHECK_ARG((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "buffers must be 16-byte aligned");
    hipLaunchKernelGGL(adam_kernel<true>, dim3(adam_grid(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr_t, scale, beta1, beta2, eps, l2, sumsq_partial);
    VC_LAUNCH_CHECK();
    return 0;
}

extern "C" int vc_sgd_f32(void* stream, float* p, const float* g, long n, const float* lr, const float* scale, float l2) {
    VC_CHECK_ARG(p && g && lr && n >= 0, "bad argument");
    if (n == 0) return 0;
    hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, n, lr, scale, l2);
    VC_LAUNCH_CHECK();
    return 0;
}

extern "C" int vc_momentum_f32(void* stream, float* p, const float* g, float* accum, long n, const float* lr,
                               const float* scale, float momentum, float l2, const float* row_mask, int E) {
    VC_CHECK_ARG(p && g && accum && lr && n >= 0 && (!row_mask || E > 0), "bad argument");
    if (n == 0) return 0;
    hipLaunchKernelGGL(momentum_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, accum, n, lr, scale, momentum, l2, row_mask, E > 0 ? E : 1);
    VC_LAUNCH_CHECK();
    return 0;
}

static int philox_launch(void* stream, void* out, long n, uint64_t seed, uint64_t offset, int mode, float keep,
                         const int32_t* step) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(philox_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, out, n, seed, offset, mode, keep, step);
    return launch_status("vc_philox");
}

extern "C" int vc_philox_u32(void* stream, uint32_t* out, long n, uint64_t seed, uint64_t offset, const int32_t* step) {
    VC_CHECK_ARG(out && n >= 0, "bad argument");
    return philox_launch(stream, out, n, seed, offset, 0, 0.f, step);
}
extern "C" int vc_philox_normal_f32(void* stream, float* out, long n, uint64_t seed, uint64_t offset, const int32_t* step) {
    VC_CHECK_ARG(out && n >= 0, "bad argument");
    return philox_launch(stream, out, n, seed, offset, 1, 0.f, step);
}
extern "C" int vc_philox_uniform_f32(void* stream, float* out, long n, uint64_t seed, uint64_t offset, const int32_t* step) {
    VC_CHECK_ARG(out && n >= 0, "bad argument");
    return philox_launch(stream, out, n, seed, offset, 3, 0.f, step);
}
extern "C" int vc_philox_bernoulli_f32(void* stream, float* out, long n, float keep, uint64_t seed, uint64_t offset,
                                       const int32_t* step) {
    VC_CHECK_ARG(out && n >= 0 && keep > 0.f && keep <= 1.f, "bad argument");
    return philox_launch(stream, out, n, seed, offset, 2, keep, step);
}
