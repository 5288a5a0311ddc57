// Loss-side kernels: masked sparse softmax cross-entropy (forward + gradient in one pass
// over the logits), reparameterised Gaussian sample, KL terms for the Normal / GMM / AG
// priors and their gradients, and the 90-component head mixing of the GMM / AG encoders.
//
// Reference: main.py:118-177 (KL, masked CE, lower bound), vae_model/encoder.py:59-109
// (heads, zs.Normal sample), vae_model/decoder.py:109-110 (the z buffer is consumed as a
// raw [N, S*L] reshape of [S, N, L] -- quirk Q1 -- i.e. the SAME memory, no kernel).
#include "common.h"
#include "vaecap.h"

namespace vc {

static inline int grid_for(long work_items, int per_block = 256, int cap = 2048) {
    long b = (work_items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (int)b;
}

// ---------------------------------------------------------------------------------------
// tf.nn.sparse_softmax_cross_entropy_with_logits + mask + div (main.py:152-158, Q8).
// One workgroup per row; the row (V <= 12288, row pitch % 4 == 0) lives in registers, so the
// logits are read ONCE from HBM and overwritten in place by d(loss)/d(logits):
//   ce = logsumexp(x) - x[label]; mask = (label != 0); row_loss = ce * mask
//   dlogits = (softmax(x) - onehot(label)) * mask * gscale / den[0]
// HBM traffic = 4 B read + 4 B written per logit (the algorithmic minimum for an unfused
// softmax-CE with gradient).
// ---------------------------------------------------------------------------------------
constexpr int XENT_MAXQ = 12;  // float4 per thread

template <bool WRITE_GRAD>
__global__ __launch_bounds__(256) void xent_reg_kernel(float* __restrict__ logits, const int32_t* __restrict__ labels,
                                                       int V, long ld, const float* __restrict__ den, float gscale,
                                                       float* __restrict__ row_loss) {
    __shared__ float sh[4];
    const long row = blockIdx.x;
    float4* p = reinterpret_cast<float4*>(logits + row * ld);
    const int V4 = (V + 3) >> 2;  // V % 4 != 0 (the reference's observed 11313): the row's last quad reaches into the ld padding
    const int label = labels[row];
    float4 x[XENT_MAXQ];
    float mx = -INFINITY;
#pragma unroll
    for (int q = 0; q < XENT_MAXQ; ++q) {
        const int i = threadIdx.x + q * 256;
        if (i < V4) {
            x[q] = p[i];
            if (i * 4 + 3 >= V) {  // columns >= V do not exist: -inf drops them from the max and the sum (exp -> 0, gradient 0)
                if (i * 4 + 1 >= V) x[q].y = -INFINITY;
                if (i * 4 + 2 >= V) x[q].z = -INFINITY;
                x[q].w = -INFINITY;
            }
            mx = fmaxf(mx, fmaxf(fmaxf(x[q].x, x[q].y), fmaxf(x[q].z, x[q].w)));
        }
    }
    mx = block_max<256>(mx, sh);
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < XENT_MAXQ; ++q) {
        const int i = threadIdx.x + q * 256;
        if (i < V4) {
            x[q].x = __expf(x[q].x - mx); x[q].y = __expf(x[q].y - mx);
            x[q].z = __expf(x[q].z - mx); x[q].w = __expf(x[q].w - mx);
            s += (x[q].x + x[q].y) + (x[q].z + x[q].w);
        }
    }
    s = block_sum<256>(s, sh);
    const bool live = label != 0;
    if (threadIdx.x == 0) {
        float l = 0.f;
        if (live && label > 0 && label < V) l = __logf(s) + mx - logits[row * ld + label];
        row_loss[row] = l;
    }
    if (WRITE_GRAD) {
        __syncthreads();  // thread 0 has read logits[label] before anyone overwrites it
        const float k = live ? gscale / den[0] : 0.f;
        const float inv = k / s;
#pragma unroll
        for (int q = 0; q < XENT_MAXQ; ++q) {
            const int i = threadIdx.x + q * 256;
            if (i < V4) {
                float4 d = make_float4(x[q].x * inv, x[q].y * inv, x[q].z * inv, x[q].w * inv);
                const int c = i * 4;
                if (label >= c && label < c + 4) {
                    if (label == c) d.x -= k; else if (label == c + 1) d.y -= k; else if (label == c + 2) d.z -= k; else d.w -= k;
                }
                p[i] = d;
            }
        }
    }
}

// Generic V: three passes over the row (passes 2 and 3 hit L2).
template <bool WRITE_GRAD>
__global__ __launch_bounds__(256) void xent_mem_kernel(float* __restrict__ logits, const int32_t* __restrict__ labels,
                                                       int V, long ld, const float* __restrict__ den, float gscale,
                                                       float* __restrict__ row_loss) {
    __shared__ float sh[4];
    const long row = blockIdx.x;
    float* p = logits + row * ld;
    const int label = labels[row];
    float mx = -INFINITY;
    for (int c = threadIdx.x; c < V; c += 256) mx = fmaxf(mx, p[c]);
    mx = block_max<256>(mx, sh);
    float s = 0.f;
    for (int c = threadIdx.x; c < V; c += 256) s += __expf(p[c] - mx);
    s = block_sum<256>(s, sh);
    const bool live = label != 0;
    if (threadIdx.x == 0) {
        float l = 0.f;
        if (live && label > 0 && label < V) l = __logf(s) + mx - p[label];
        row_loss[row] = l;
    }
    if (WRITE_GRAD) {
        __syncthreads();
        const float k = live ? gscale / den[0] : 0.f;
        const float inv = k / s;
        for (int c = threadIdx.x; c < V; c += 256) {
            float d = __expf(p[c] - mx) * inv;
            if (c == label) d -= k;
            p[c] = d;
        }
    }
}

// softmax probabilities per row (gen mode: tf.nn.softmax, vae_model/decoder.py:140,142)
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ x, int V, long ld,
                                                           float* __restrict__ y, long ldy) {
    __shared__ float sh[4];
    const float* p = x + (long)blockIdx.x * ld;
    float* q = y + (long)blockIdx.x * ldy;
    float mx = -INFINITY;
    for (int c = threadIdx.x; c < V; c += 256) mx = fmaxf(mx, p[c]);
    mx = block_max<256>(mx, sh);
    float s = 0.f;
    for (int c = threadIdx.x; c < V; c += 256) s += __expf(p[c] - mx);
    s = block_sum<256>(s, sh);
    const float inv = 1.f / s;
    for (int c = threadIdx.x; c < V; c += 256) q[c] = __expf(p[c] - mx) * inv;
}

// The same with the row held in registers (V <= 12288, 16-byte aligned pitches): one read of the logits instead of three, 16-byte
// accesses (34 -> ~12 us for 640 rows x 10 000 columns)
__global__ __launch_bounds__(256) void softmax_rows_reg_kernel(const float* __restrict__ x, int V, long ld,
                                                               float* __restrict__ y, long ldy) {
    __shared__ float sh[4];
    const float4* p = reinterpret_cast<const float4*>(x + (long)blockIdx.x * ld);
    float4* q = reinterpret_cast<float4*>(y + (long)blockIdx.x * ldy);
    const int Q = V >> 2;
    float4 r[12];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const int c = threadIdx.x + 256 * i;
        if (c < Q) {
            r[i] = p[c];
            mx = fmaxf(fmaxf(mx, fmaxf(r[i].x, r[i].y)), fmaxf(r[i].z, r[i].w));
        }
    }
    mx = block_max<256>(mx, sh);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const int c = threadIdx.x + 256 * i;
        if (c < Q) {
            r[i].x = __expf(r[i].x - mx); r[i].y = __expf(r[i].y - mx); r[i].z = __expf(r[i].z - mx); r[i].w = __expf(r[i].w - mx);
            s += (r[i].x + r[i].y) + (r[i].z + r[i].w);
        }
    }
    s = block_sum<256>(s, sh);
    const float inv = 1.f / s;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const int c = threadIdx.x + 256 * i;
        if (c < Q) q[c] = make_float4(r[i].x * inv, r[i].y * inv, r[i].z * inv, r[i].w * inv);
    }
}

// Categorical sample per row from logits / temperature (tf.multinomial, vae_model/decoder.py:137-138) by
// inverse CDF with an INJECTED uniform u[row] in [0,1): index = first i with cumsum(softmax)[i] > u.
// (TF draws with its own Gumbel/Philox stream, which cannot be matched; the distribution is the same.)
__global__ __launch_bounds__(256) void multinomial_rows_kernel(const float* __restrict__ logits, int V, long ld, float inv_temp,
                                                               const float* __restrict__ u, int32_t* __restrict__ out) {
    __shared__ float sh[4];
    __shared__ float part[256];
    const float* p = logits + (long)blockIdx.x * ld;
    float mx = -INFINITY;
    for (int c = threadIdx.x; c < V; c += 256) mx = fmaxf(mx, p[c] * inv_temp);
    mx = block_max<256>(mx, sh);
    // contiguous chunk per thread so that the scan order is the index order
    const int per = (V + 255) / 256;
    const int c0 = threadIdx.x * per, c1 = min(V, c0 + per);
    float s = 0.f;
    for (int c = c0; c < c1; ++c) s += __expf(p[c] * inv_temp - mx);
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = 0.f;
        for (int i = 0; i < 256; ++i) tot += part[i];
        const float target = u[blockIdx.x] * tot;
        float run = 0.f;
        int t = 0;
        for (; t < 255; ++t) {
            if (run + part[t] > target) break;
            run += part[t];
        }
        int idx = min(V - 1, t * per);
        for (int c = t * per; c < min(V, (t + 1) * per); ++c) {
            run += __expf(p[c] * inv_temp - mx);
            idx = c;
            if (run > target) break;
        }
        out[blockIdx.x] = idx;
    }
}

// ---------------------------------------------------------------------------------------
// z[s,n,l] = mean[n,l] + std[n,l] * eps[s,n,l]   (zs.Normal, n_samples=S; encoder.py:108-109)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sample_kernel(const float* __restrict__ mean, const float* __restrict__ std_,
                                                     const float* __restrict__ eps, long NL, long total,
                                                     float* __restrict__ z) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long j = i % NL;
        z[i] = mean[j] + std_[j] * eps[i];
    }
}

// Data-parallel form of the sample (quirk Q1 under sharding, see dp.py): the z_rnn input rows of this
// rank are the flat range q in [q0, q0+nq) of the GLOBAL [S, Ng, L] sample tensor, q = s*Ng + n.
//   z[(q-q0), l] = mean_g[q % Ng, l] + std_g[q % Ng, l] * eps[(q-q0), l]
// With Ng = N, q0 = 0, nq = S*N it is exactly sample_kernel.
__global__ __launch_bounds__(256) void sample_mixed_kernel(const float* __restrict__ mean_g, const float* __restrict__ std_g,
                                                           const float* __restrict__ eps, int Ng, int L, long q0, long total,
                                                           float* __restrict__ z) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long q = q0 + i / L;
        const long j = (q % Ng) * L + i % L;
        z[i] = mean_g[j] + std_g[j] * eps[i];
    }
}
// ... and of its gradient: per global row n, the sums over this rank's q == n (mod Ng):
//   dmean_part[n,l] = sum dz[(q-q0), l],   dstd_part[n,l] = sum dz[(q-q0), l] * eps[(q-q0), l]
// (rows that do not occur get zeros); the partials are then reduce-scattered over the ranks.
__global__ __launch_bounds__(256) void latent_sums_mixed_kernel(const float* __restrict__ dz, const float* __restrict__ eps,
                                                                int Ng, int L, long q0, long nq, float* __restrict__ dmean,
                                                                float* __restrict__ dstd) {
    const long NL = (long)Ng * L;
    for (long j = (long)blockIdx.x * 256 + threadIdx.x; j < NL; j += (long)gridDim.x * 256) {
        const long n = j / L;
        const int l = (int)(j % L);
        long q = q0 + ((n - q0 % Ng) % Ng + Ng) % Ng;  // first q >= q0 with q % Ng == n
        float dm = 0.f, ds = 0.f;
        for (; q < q0 + nq; q += Ng) {
            const float g = dz[(q - q0) * L + l];
            dm += g;
            ds += g * eps[(q - q0) * L + l];
        }
        dmean[j] = dm;
        dstd[j] = ds;
    }
}

// ---------------------------------------------------------------------------------------
// KL per row.  mode 0 (Normal / GMM, main.py:120-124,131-135):
//     -0.5 * sum_l (1 + log(std^2 + 1e-5) - mean^2 - std^2)
// mode 1 (AG, main.py:140-145):
//     -0.5 * sum_l [0.5 + log(std+1e-5) - log(0.1+1e-5) - ((mean-mu_p)^2 + std^2)/(2*0.1^2+1e-7)]
// One wave per row.
// ---------------------------------------------------------------------------------------
#define VC_AG_SIGMA 0.1f
__global__ __launch_bounds__(256) void kl_rows_kernel(const float* __restrict__ mean, const float* __restrict__ std_,
                                                      const float* __restrict__ mu_p, int N, int L, int mode,
                                                      float* __restrict__ row_kl) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const int lane = threadIdx.x & 63;
    float s = 0.f;
    for (int l = lane; l < L; l += 64) {
        const float m = mean[(long)row * L + l], sd = std_[(long)row * L + l];
        if (mode == 0) {
            s += 1.f + logf(sd * sd + 1e-5f) - m * m - sd * sd;
        } else {
            const float d = m - mu_p[(long)row * L + l];
            s += 0.5f + logf(sd + 1e-5f) - logf(VC_AG_SIGMA + 1e-5f) - (d * d + sd * sd) / (2.f * VC_AG_SIGMA * VC_AG_SIGMA + 1e-7f);
        }
    }
    s = wave_sum(s);
    if (lane == 0) row_kl[row] = -0.5f * s;
}

// ---------------------------------------------------------------------------------------
// Gradient w.r.t. (mean, std) of   [sum over the S samples through z]  +  kl_w * KL:
//   dmean = sum_s dz[s] + dKL/dmean ;  dstd = sum_s dz[s]*eps[s] + dKL/dstd
// kl_w = ann[0] * kl_scale (device annealing coefficient x host constant: 1/10, and 1/N for
// the batch-mean Normal KL).  out_logstd: return d/d(logstd) = dstd * std (Normal heads).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void latent_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ eps,
                                                         const float* __restrict__ mean, const float* __restrict__ std_,
                                                         const float* __restrict__ mu_p, const float* __restrict__ ann,
                                                         float kl_scale, int S, long NL, int mode, int out_logstd,
                                                         float* __restrict__ dmean, float* __restrict__ dstd) {
    const float w = (ann ? ann[0] : 1.f) * kl_scale;
    for (long j = (long)blockIdx.x * 256 + threadIdx.x; j < NL; j += (long)gridDim.x * 256) {
        // S == 0: dmean / dstd already hold the sums over the samples (vc_latent_sums_mixed_f32)
        float dm = S ? 0.f : dmean[j], ds = S ? 0.f : dstd[j];
        int s = 0;
        for (; s + 8 <= S; s += 8) {   // sixteen loads in flight, the sums in the same order as the plain loop (a thread per element is all
            float g[8], e[8];          // the parallelism N * L = 48 000 gives: with one load pair at a time the launch took 65 us at cfg4)
#pragma unroll
            for (int k = 0; k < 8; ++k) { g[k] = dz[(long)(s + k) * NL + j]; e[k] = eps[(long)(s + k) * NL + j]; }
#pragma unroll
            for (int k = 0; k < 8; ++k) { dm += g[k]; ds += g[k] * e[k]; }
        }
        for (; s < S; ++s) {
            const float g = dz[(long)s * NL + j];
            dm += g;
            ds += g * eps[(long)s * NL + j];
        }
        const float m = mean[j], sd = std_[j];
        if (mode == 0) {
            dm += w * m;
            ds -= w * (sd / (sd * sd + 1e-5f) - sd);
        } else {
            const float den = 2.f * VC_AG_SIGMA * VC_AG_SIGMA + 1e-7f;
            dm += w * (m - mu_p[j]) / den;
            ds -= 0.5f * w * (1.f / (sd + 1e-5f) - 2.f * sd / den);
        }
        dmean[j] = dm;
        dstd[j] = out_logstd ? ds * sd : ds;
    }
}

// ---------------------------------------------------------------------------------------
// GMM / AG heads.  heads [N, 2*K*L]: columns [k*L + l] = component means tm[n,k,l],
// columns [K*L + k*L + l] = log-stds tl[n,k,l]  (vae_model/encoder.py:71-107).
//   AG : mean = sum_k c[n,k]*tm[n,k,:],  std = sum_k c[n,k]*exp(tl[n,k,:])   (:105-107)
//   GMM: mean = tm[n, idx[n], :],        std = exp(tl[n, idx[n], :])         (:87-88)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void heads_mix_fwd_kernel(const float* __restrict__ heads, const float* __restrict__ c,
                                                            const int32_t* __restrict__ idx, int N, int K, int L,
                                                            float* __restrict__ mean, float* __restrict__ std_) {
    const long NL = (long)N * L;
    for (long j = (long)blockIdx.x * 256 + threadIdx.x; j < NL; j += (long)gridDim.x * 256) {
        const long n = j / L;
        const int l = (int)(j % L);
        const float* h = heads + n * 2 * K * L;
        float m = 0.f, s = 0.f;
        if (idx) {
            const int k = idx[n];
            m = h[k * L + l];
            s = __expf(h[K * L + k * L + l]);
        } else {
            for (int k = 0; k < K; ++k) {
                const float ck = c[n * K + k];
                if (ck != 0.f) {
                    m += ck * h[k * L + l];
                    s += ck * __expf(h[K * L + k * L + l]);
                }
            }
        }
        mean[j] = m;
        std_[j] = s;
    }
}

__global__ __launch_bounds__(256) void heads_mix_bwd_kernel(const float* __restrict__ heads, const float* __restrict__ c,
                                                            const int32_t* __restrict__ idx, const float* __restrict__ dmean,
                                                            const float* __restrict__ dstd, int N, int K, int L,
                                                            float* __restrict__ dheads) {
    const long total = (long)N * K * L;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long n = i / ((long)K * L);
        const int kl = (int)(i % ((long)K * L));
        const int k = kl / L, l = kl % L;
        const float w = idx ? (idx[n] == k ? 1.f : 0.f) : c[n * K + k];
        const long o = n * 2 * K * L;
        float gm = 0.f, gs = 0.f;
        if (w != 0.f) {
            gm = w * dmean[n * L + l];
            gs = w * dstd[n * L + l] * __expf(heads[o + (long)K * L + kl]);
        }
        dheads[o + kl] = gm;
        dheads[o + (long)K * L + kl] = gs;
    }
}

// Scalars of one step (main.py:156-177): out[0] rec_loss = ce_num/ce_den (+ reg), out[1] mean KL,
// out[2] lower_bound = rec + ann*kld/10 (mean over rows for the AG vector loss, what main.py:247-251
// prints), out[3] annealing coefficient.  All inputs are device scalars.
__global__ void loss_finalize_kernel(const float* ce_num, const float* ce_den, const float* reg, float reg_scale,
                                     const float* kl_sum, float inv_n, const float* ann, float* out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float rec = ce_num[0] / ce_den[0];
    if (reg) rec += reg[0] * reg_scale;
    const float a = ann ? ann[0] : 1.f;
    const float kld = kl_sum ? kl_sum[0] * inv_n : 0.f;
    out[0] = rec;
    out[1] = kld;
    out[2] = kl_sum ? rec + a * kld / 10.f : rec;
    out[3] = a;
}

}  // namespace vc

using namespace vc;

extern "C" int vc_loss_finalize_f32(void* stream, const float* ce_num, const float* ce_den, const float* reg_sumsq,
                                    float reg_scale, const float* kl_sum, float inv_n, const float* ann, float* out4) {
    VC_CHECK_ARG(ce_num && ce_den && out4, "null pointer");
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ce_num, ce_den, reg_sumsq, reg_scale, kl_sum, inv_n, ann, out4);
    VC_LAUNCH_CHECK();
    return 0;
}

extern "C" int vc_softmax_xent_f32(void* stream, float* logits, const int32_t* labels, long rows, int V, long ld,
                                   const float* den, float gscale, float* row_loss, int write_grad) {
    VC_CHECK_ARG(logits && labels && row_loss && rows >= 0 && V > 0 && ld >= V, "bad argument");
    VC_CHECK_ARG(!write_grad || den, "den required for the gradient");
    if (rows == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    // register kernel: rows of whole float4 quads -- V % 4 == 0, or a row pitch padded to a multiple of 4 (ld >= roundup(V, 4))
    const bool reg = (ld % 4 == 0) && (ld >= (long)((V + 3) / 4) * 4) && (V <= XENT_MAXQ * 256 * 4) && (((uintptr_t)logits & 15) == 0);
    dim3 g((unsigned)rows), b(256);
    if (reg) {
        if (write_grad) hipLaunchKernelGGL(xent_reg_kernel<true>, g, b, 0, st, logits, labels, V, ld, den, gscale, row_loss);
        else hipLaunchKernelGGL(xent_reg_kernel<false>, g, b, 0, st, logits, labels, V, ld, den, gscale, row_loss);
    } else {
        if (write_grad) hipLaunchKernelGGL(xent_mem_kernel<true>, g, b, 0, st, logits, labels, V, ld, den, gscale, row_loss);
        else hipLaunchKernelGGL(xent_mem_kernel<false>, g, b, 0, st, logits, labels, V, ld, den, gscale, row_loss);
    }
    VC_LAUNCH_CHECK();
    return 0;
}

extern "C" int vc_softmax_rows_f32(void* stream, const float* x, long rows, int V, long ld, float* y, long ldy) {
    VC_CHECK_ARG(x && y && rows >= 0 && V > 0 && ld >= V && ldy >= V, "bad argument");
    if (rows == 0) return 0;
    if ((V & 3) == 0 && V <= 12288 && (ld & 3) == 0 && (ldy & 3) == 0 && ((((uintptr_t)x) | ((uintptr_t)y)) & 15) == 0)
        hipLaunchKernelGGL(softmax_rows_reg_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, V, ld, y, ldy);
    else
        hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, V, ld, y, ldy);
    VC_LAUNCH_CHECK();
    return 0;
}

extern "C" int vc_multinomial_rows_f32(void* stream, const float* logits, long rows, int V, long ld, float temperature,
                                       const float* u, int32_t* out) {
    VC_CHECK_ARG(logits && u && out && rows >= 0 && V > 0 && ld >= V && temperature > 0.f, "bad argument");
    if (rows == 0) return 0;
    hipLaunchKernelGGL(multinomial_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, logits, V, ld, 1.0f / temperature, u, out);
    VC_LAUNCH_CHECK();
    return 0;
}

extern "C" int vc_latent_sample_f32(void* stream, int S, int N, int L, const float* mean, const float* std_,
                                    const float* eps, float* z) {
    VC_CHECK_ARG(mean && std_ && eps && z && S > 0 && N > 0 && L > 0, "bad argument");
    const long NL = (long)N * L, total = NL * S;
    hipLaunchKernelGGL(sample_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, mean, std_, eps, NL, total, z);
    VC_LAUNCH_CHECK();
    return 0;
}

extern "C" int vc_latent_sample_mixed_f32(void* stream, int Ng, int L, long q0, long nq, const float* mean_g,
                                          const float* std_g, const float* eps, float* z) {
    VC_CHECK_ARG(mean_g && std_g && eps && z && Ng > 0 && L > 0 && q0 >= 0 && nq > 0, "bad argument");
    const long total = nq * L;
    hipLaunchKernelGGL(sample_mixed_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, mean_g, std_g, eps, Ng, L, q0, total, z);
    VC_LAUNCH_CHECK();
    return 0;
}

extern "C" int vc_latent_sums_mixed_f32(void* stream, int Ng, int L, long q0, long nq, const float* dz, const float* eps,
                                        float* dmean_part, float* dstd_part) {
    VC_CHECK_ARG(dz && eps && dmean_part && dstd_part && Ng > 0 && L > 0 && q0 >= 0 && nq > 0, "bad argument");
    hipLaunchKernelGGL(latent_sums_mixed_kernel, dim3(grid_for((long)Ng * L)), dim3(256), 0, (hipStream_t)stream, dz, eps, Ng, L, q0, nq, dmean_part, dstd_part);
    VC_LAUNCH_CHECK();
    return 0;
}

extern "C" int vc_kl_rows_f32(void* stream, int N, int L, int mode, const float* mean, const float* std_,
                              const float* mu_p, float* row_kl) {
    VC_CHECK_ARG(mean && std_ && row_kl && N > 0 && L > 0 && (mode == 0 || (mode == 1 && mu_p)), "bad argument");
    hipLaunchKernelGGL(kl_rows_kernel, dim3(cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream, mean, std_, mu_p, N, L, mode, row_kl);
    VC_LAUNCH_CHECK();
    return 0;
}

extern "C" int vc_latent_bwd_f32(void* stream, int S, int N, int L, int mode, int out_logstd, const float* dz,
                                 const float* eps, const float* mean, const float* std_, const float* mu_p,
                                 const float* ann, float kl_scale, float* dmean, float* dstd) {
    VC_CHECK_ARG(mean && std_ && dmean && dstd && S >= 0 && (S == 0 || (dz && eps)) && N > 0 && L > 0, "bad argument");
    VC_CHECK_ARG(mode == 0 || (mode == 1 && mu_p), "mu_p required for the AG prior");
    const long NL = (long)N * L;
    hipLaunchKernelGGL(latent_bwd_kernel, dim3(grid_for(NL)), dim3(256), 0, (hipStream_t)stream, dz, eps, mean, std_, mu_p, ann,
                       kl_scale, S, NL, mode, out_logstd, dmean, dstd);
    VC_LAUNCH_CHECK();
    return 0;
}

extern "C" int vc_heads_mix_fwd_f32(void* stream, int N, int K, int L, const float* heads, const float* c_i,
                                    const int32_t* idx, float* mean, float* std_) {
    VC_CHECK_ARG(heads && mean && std_ && (c_i || idx) && N > 0 && K > 0 && L > 0, "bad argument");
    hipLaunchKernelGGL(heads_mix_fwd_kernel, dim3(grid_for((long)N * L)), dim3(256), 0, (hipStream_t)stream, heads, c_i, idx, N, K, L, mean, std_);
    VC_LAUNCH_CHECK();
    return 0;
}

extern "C" int vc_heads_mix_bwd_f32(void* stream, int N, int K, int L, const float* heads, const float* c_i,
                                    const int32_t* idx, const float* dmean, const float* dstd, float* dheads) {
    VC_CHECK_ARG(heads && dmean && dstd && dheads && (c_i || idx) && N > 0 && K > 0 && L > 0, "bad argument");
    hipLaunchKernelGGL(heads_mix_bwd_kernel, dim3(grid_for((long)N * K * L)), dim3(256), 0, (hipStream_t)stream, heads, c_i, idx, dmean, dstd, N, K, L, dheads);
    VC_LAUNCH_CHECK();
    return 0;
}
