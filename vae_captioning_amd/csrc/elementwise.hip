// HBM-bound data-movement kernels: embedding gather / scatter-add, column sums (bias
// gradients), dropout, ReLU backward, row tiling, reductions.  All are coalesced,
// 16-B vectorised where the layout allows, and sized as grid-stride loops over <= 2048
// workgroups (cdna_hip_programming.md guideline 11).
#include "common.h"
#include "vaecap.h"

namespace vc {

static inline int grid_for(long work_items, int per_block = 256, int cap = 2048) {
    long b = (work_items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (int)b;
}

// ---- embedding ---------------------------------------------------------------------
// tf.nn.embedding_lookup, vae_model/encoder.py:31-36, vae_model/decoder.py:77-83.
template <bool VEC>
__global__ __launch_bounds__(256) void gather_kernel(const float* __restrict__ table, const int32_t* __restrict__ ids,
                                                     long rows, int E, int vocab, float* __restrict__ out) {
    if (VEC) {
        const int E4 = E >> 2;
        const long total = rows * E4;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
            const long r = i / E4;
            const int e = (int)(i % E4);
            int id = ids[r];
            id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
            reinterpret_cast<float4*>(out)[i] = reinterpret_cast<const float4*>(table + (long)id * E)[e];
        }
    } else {
        const long total = rows * E;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
            const long r = i / E;
            int id = ids[r];
            id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
            out[i] = table[(long)id * E + (i % E)];
        }
    }
}

// Gradient of the lookup = IndexedSlices scatter-add (TF-sem.).  fp32 hardware atomics:
// the order in which duplicate tokens are summed is not fixed (sum is order-dependent at
// the 1e-7 level only).
__global__ __launch_bounds__(256) void scatter_add_kernel(float* __restrict__ dtable, const int32_t* __restrict__ ids,
                                                          long rows, int E, int vocab, const float* __restrict__ dX) {
    const long total = rows * E;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / E;
        const int id = ids[r];
        if (id < 0 || id >= vocab) continue;
        const float v = dX[i];
        if (v != 0.f) unsafeAtomicAdd(dtable + (long)id * E + (i % E), v);
    }
}

// Deterministic alternative: the caller supplies a stable argsort of the ids (`order`) and the
// per-output-row segment starts; one wave owns one output row and sums its positions in order
// (no atomics, no memset: rows with an empty segment are written as zeros).  order == NULL means
// the identity (used for the second level of a two-level sum: hot tokens such as <PAD>/<BOS>/<EOS>
// are first reduced in sub-segments of <= 32 positions, then the partial rows are summed).
template <bool VEC>
__global__ __launch_bounds__(256) void segment_grad_kernel(float* __restrict__ dtable, const int32_t* __restrict__ order,
                                                           const int32_t* __restrict__ seg_start, int E, int nrows,
                                                           const float* __restrict__ dX) {
    const int lane = threadIdx.x & 63;
    for (int v = blockIdx.x * 4 + (threadIdx.x >> 6); v < nrows; v += gridDim.x * 4) {
        const int s0 = seg_start[v], s1 = seg_start[v + 1];
        if (VEC) {
            for (int e = lane * 4; e < E; e += 256) {
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int j = s0; j < s1; ++j) {
                    const long r = order ? order[j] : j;
                    const float4 x = *reinterpret_cast<const float4*>(dX + r * E + e);
                    acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
                }
                *reinterpret_cast<float4*>(dtable + (long)v * E + e) = acc;
            }
        } else {
            for (int e = lane; e < E; e += 64) {
                float acc = 0.f;
                for (int j = s0; j < s1; ++j) acc += dX[(long)(order ? order[j] : j) * E + e];
                dtable[(long)v * E + e] = acc;
            }
        }
    }
}

// ---- inverted index for the deterministic embedding gradient, built ON DEVICE (was numpy argsort + bincount on the
// host inside every set_batch).  A stable counting sort of the R token ids in three launches:
//   count : block (vocabulary block, segment s of the positions): thread v counts its id in segment s   -> cnt[s][v]
//   scan  : three small launches (per-id totals; one-workgroup exclusive scans through LDS; per-id finish) -> first slot of
//           every (segment, id), the sub-segment table seg1 (chunks of <= `chunk` positions, never crossing an id boundary)
//           and seg2 (sub-segment range of each id)
//   place : same grid as count: thread v walks segment s again and writes the positions of its id in order
// Every thread compares its id against the segment's ids streamed through LDS (broadcast reads): R/segments x V
// comparisons per launch, a few microseconds at R = 25600, V = 10000.  Integer work, no atomics: bit-identical to
// engine.embedding_grad_index (numpy) by construction.
constexpr int IDX_TILE = 1024;

template <bool PLACE>
__global__ __launch_bounds__(256) void embidx_scan_ids_kernel(const int32_t* __restrict__ ids, long R, int vocab, long seg_len,
                                                              int32_t* __restrict__ cnt /* [segs][vocab]: counts, or first slots */,
                                                              int32_t* __restrict__ order) {
    __shared__ __attribute__((aligned(16))) int32_t tile[IDX_TILE];
    const int v = blockIdx.x * 256 + threadIdx.x;
    const long p0 = (long)blockIdx.y * seg_len;
    const long p1 = p0 + seg_len < R ? p0 + seg_len : R;
    int32_t cur = 0;
    if (PLACE && v < vocab) cur = cnt[(long)blockIdx.y * vocab + v];
    for (long base = p0; base < p1; base += IDX_TILE) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < IDX_TILE / 256; ++u) {
            const long p = base + threadIdx.x + 256 * u;
            int32_t id = -1;
            if (p < p1) {
                id = ids[p];
                id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
            }
            tile[threadIdx.x + 256 * u] = id;
        }
        __syncthreads();
        const int n = (int)(p1 - base < IDX_TILE ? p1 - base : IDX_TILE);
        for (int j = 0; j < n; j += 4) {
            const int4 q = *reinterpret_cast<const int4*>(&tile[j]);  // same address in every lane: LDS broadcast
            if (PLACE) {
                if (q.x == v) order[cur++] = (int32_t)(base + j);
                if (q.y == v) order[cur++] = (int32_t)(base + j + 1);
                if (q.z == v) order[cur++] = (int32_t)(base + j + 2);
                if (q.w == v) order[cur++] = (int32_t)(base + j + 3);
            } else {
                cur += (q.x == v) + (q.y == v) + (q.z == v) + (q.w == v);
            }
        }
    }
    if (!PLACE && v < vocab) cnt[(long)blockIdx.y * vocab + v] = cur;
}

// scan, stage 1 (thread = id, coalesced): per-id totals and sub-segment counts; cnt[s][v] becomes the id-local prefix over segments
__global__ __launch_bounds__(256) void embidx_totals_kernel(int32_t* __restrict__ cnt, int segs, int vocab, int chunk,
                                                            int32_t* __restrict__ tot, int32_t* __restrict__ nsub) {
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= vocab) return;
    int32_t t = 0;
    for (int s = 0; s < segs; ++s) {
        const int32_t c = cnt[(long)s * vocab + v];
        cnt[(long)s * vocab + v] = t;
        t += c;
    }
    tot[v] = t;
    nsub[v] = (t + chunk - 1) / chunk;
}

// scan, stage 2 (one workgroup): exclusive scans over the ids of the totals (-> first slot of each id) and of the sub-segment
// counts (-> seg2), through LDS so that global accesses stay coalesced
__global__ __launch_bounds__(1024) void embidx_scan_kernel(const int32_t* __restrict__ tot, const int32_t* __restrict__ nsub, int vocab,
                                                           int32_t* __restrict__ starts, int32_t* __restrict__ seg2) {
    extern __shared__ int32_t sh[];  // [2][vocab] + [2][1024]
    int32_t* a = sh;
    int32_t* b = sh + vocab;
    int32_t* pa = sh + 2 * vocab;
    int32_t* pb = pa + 1024;
    const int t = threadIdx.x;
    for (int v = t; v < vocab; v += 1024) { a[v] = tot[v]; b[v] = nsub[v]; }
    __syncthreads();
    const int per = (vocab + 1023) / 1024;
    const int v0 = t * per, v1 = v0 + per < vocab ? v0 + per : vocab;
    int32_t sa = 0, sb = 0;
    for (int v = v0; v < v1; ++v) { sa += a[v]; sb += b[v]; }
    pa[t] = sa; pb[t] = sb;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int32_t xa = t >= o ? pa[t - o] : 0, xb = t >= o ? pb[t - o] : 0;
        __syncthreads();
        pa[t] += xa; pb[t] += xb;
        __syncthreads();
    }
    int32_t ea = pa[t] - sa, eb = pb[t] - sb;
    for (int v = v0; v < v1; ++v) {
        const int32_t ca = a[v], cb = b[v];
        a[v] = ea; b[v] = eb;
        ea += ca; eb += cb;
    }
    __syncthreads();
    for (int v = t; v < vocab; v += 1024) { starts[v] = a[v]; seg2[v] = b[v]; }
    if (t == 0) seg2[vocab] = pb[1023];
}

// scan, stage 3 (thread = id): first slot of every (segment, id); the id's sub-segment boundaries; unused sub-segments -> empty
__global__ __launch_bounds__(256) void embidx_finish_kernel(int32_t* __restrict__ cnt, int segs, int vocab, long R, int chunk, int nsub_max,
                                                            const int32_t* __restrict__ tot, const int32_t* __restrict__ starts,
                                                            int32_t* __restrict__ seg1, const int32_t* __restrict__ seg2) {
    const int v = blockIdx.x * 256 + threadIdx.x;
    const int32_t nsub = seg2[vocab];
    for (int k = nsub + v; k <= nsub_max; k += gridDim.x * 256) seg1[k] = (int32_t)R;  // closes the last sub-segment; the rest are empty
    if (v >= vocab) return;
    const int32_t st = starts[v];
    for (int s = 0; s < segs; ++s) cnt[(long)s * vocab + v] += st;
    const int32_t ns = (tot[v] + chunk - 1) / chunk, sb = seg2[v];
    for (int w = 0; w < ns; ++w) seg1[sb + w] = st + w * chunk;
}

__global__ __launch_bounds__(256) void mark_rows_kernel(float* __restrict__ touched, const int32_t* __restrict__ ids,
                                                        long n, int vocab) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int id = ids[i];
        if (id >= 0 && id < vocab) touched[id] = 1.f;
    }
}

// ---- column sums (bias gradients): two deterministic stages --------------------------
// Stage 1: grid (cols/64, chunks); a workgroup owns 64 columns x one contiguous row chunk and
// reads it as 16 row-lanes x 16 float4 lanes (256 B per row segment, fully coalesced).
constexpr int COLSUM_MAX_CHUNKS = 1024;
template <bool VEC>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, long rows, int cols, long ld,
                                                             float* __restrict__ part) {
    __shared__ float sh[16][65];
    const long per = (rows + gridDim.y - 1) / gridDim.y;
    const long r0 = (long)blockIdx.y * per;
    const long r1 = r0 + per < rows ? r0 + per : rows;
    if (VEC) {
        const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
        const int c = blockIdx.x * 64 + cl * 4;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < cols)
            for (long r = r0 + rl; r < r1; r += 16) {
                const float4 v = *reinterpret_cast<const float4*>(x + r * ld + c);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
        sh[rl][cl * 4 + 0] = s.x; sh[rl][cl * 4 + 1] = s.y; sh[rl][cl * 4 + 2] = s.z; sh[rl][cl * 4 + 3] = s.w;
    } else {
        const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
        const int c = blockIdx.x * 64 + cl;
        float s = 0.f;
        if (c < cols)
            for (long r = r0 + rl; r < r1; r += 4) s += x[r * ld + c];
        sh[rl][cl] = s;
        if (rl == 0)
            for (int k = 4; k < 16; ++k) sh[k][cl] = 0.f;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int c = blockIdx.x * 64 + threadIdx.x;
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sh[k][threadIdx.x];
        if (c < cols) part[(long)blockIdx.y * cols + c] = t;
    }
}
// Stage 2: a workgroup owns 64 columns; 4 row-lanes each sum every 4th chunk partial, then the four
// lane sums are combined in a fixed order.
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, int chunks, int cols,
                                                           float* __restrict__ out, int accumulate) {
    __shared__ float sh[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float s = 0.f;
    if (c < cols)
        for (int k = rl; k < chunks; k += 4) s += part[(long)k * cols + c];
    sh[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && c < cols) {
        const float t = (sh[0][cl] + sh[1][cl]) + (sh[2][cl] + sh[3][cl]);
        out[c] = accumulate ? out[c] + t : t;
    }
}

// ---- dropout / relu ------------------------------------------------------------------
// tf.nn.dropout (TF-sem.): y = x * mask / keep.  vae_model/decoder.py:85-87,
// utils/rnn_model.py:45-46, utils/image_embeddings.py:225-226,236-237.
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, const float* __restrict__ mask,
                                                      float inv_keep, long n, float* __restrict__ y) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] = x[i] * mask[i] * inv_keep;
}
// dx = dy * (y > 0) [* mask / keep]  -- ReluGrad (+ dropout backward)
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                       const float* __restrict__ mask, float inv_keep, long n,
                                                       float* __restrict__ dx) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float v = y[i] > 0.f ? dy[i] : 0.f;
        if (mask) v *= mask[i] * inv_keep;
        dx[i] = v;
    }
}

// ---- row tiling (main.py:84-89) and its gradient ---------------------------------------
__global__ __launch_bounds__(256) void tile_rows_kernel(const float* __restrict__ x, long B, int nc, int E,
                                                        float* __restrict__ y) {
    const long total = B * nc * E;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long row = i / E;
        y[i] = x[(row / nc) * E + (i % E)];
    }
}
__global__ __launch_bounds__(256) void segment_sum_kernel(const float* __restrict__ y, long B, int nc, int E,
                                                          float* __restrict__ x, int accumulate) {
    const long total = B * E;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long b = i / E;
        const int e = (int)(i % E);
        float s = 0.f;
        for (int j = 0; j < nc; ++j) s += y[(b * nc + j) * E + e];
        x[i] = accumulate ? x[i] + s : s;
    }
}

// ---- misc ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void exp_kernel(const float* __restrict__ x, long n, float* __restrict__ y) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] = __expf(x[i]);
}
__global__ __launch_bounds__(256) void axpy_kernel(float a, const float* __restrict__ x, long n, float* __restrict__ y) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] += a * x[i];
}

// single-workgroup fixed-order sum: out = scale * sum(x) (+ out)
__global__ __launch_bounds__(1024) void reduce_sum_kernel(const float* __restrict__ x, long n, float scale,
                                                          float* __restrict__ out, int accumulate) {
    __shared__ float sh[16];
    float s = 0.f;
    for (long i = threadIdx.x; i < n; i += 1024) s += x[i];
    s = block_sum<1024>(s, sh);
    if (threadIdx.x == 0) out[0] = accumulate ? out[0] + scale * s : scale * s;
}
__global__ __launch_bounds__(1024) void count_nonzero_kernel(const int32_t* __restrict__ x, long n, float* __restrict__ out) {
    __shared__ float sh[16];
    float s = 0.f;
    for (long i = threadIdx.x; i < n; i += 1024) s += (x[i] != 0) ? 1.f : 0.f;
    s = block_sum<1024>(s, sh);
    if (threadIdx.x == 0) out[0] = s;
}

// argmax per row: first maximum (np.argmax tie rule), one workgroup per row.
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ x, int cols, long ld,
                                                          int32_t* __restrict__ out) {
    __shared__ float sv[256];
    __shared__ int si[256];
    const float* p = x + (long)blockIdx.x * ld;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = threadIdx.x; c < cols; c += 256) {
        const float v = p[c];
        if (v > bv) { bv = v; bi = c; }
    }
    sv[threadIdx.x] = bv;
    si[threadIdx.x] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            const float v2 = sv[threadIdx.x + o];
            const int i2 = si[threadIdx.x + o];
            if (v2 > sv[threadIdx.x] || (v2 == sv[threadIdx.x] && i2 < si[threadIdx.x])) {
                sv[threadIdx.x] = v2;
                si[threadIdx.x] = i2;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = si[0];
}

// top-k per row in (value descending, index ascending) order == the first k entries of a STABLE
// sort on -p (vae_model/decoder.py:273-276).  k passes; pass j finds the best element strictly after
// the (j-1)-th winner in that order, so the data are never modified.
__global__ __launch_bounds__(256) void topk_rows_kernel(const float* __restrict__ x, int cols, long ld, int k,
                                                        float* __restrict__ out_val, int32_t* __restrict__ out_idx) {
    __shared__ float sv[256];
    __shared__ int si[256];
    const float* p = x + (long)blockIdx.x * ld;
    float lv = INFINITY;
    int li = -1;
    for (int j = 0; j < k; ++j) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int c = threadIdx.x; c < cols; c += 256) {
            const float v = p[c];
            const bool eligible = (v < lv) || (v == lv && c > li);
            if (eligible && (v > bv || (v == bv && c < bi))) { bv = v; bi = c; }
        }
        sv[threadIdx.x] = bv;
        si[threadIdx.x] = bi;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (threadIdx.x < o) {
                const float v2 = sv[threadIdx.x + o];
                const int i2 = si[threadIdx.x + o];
                if (v2 > sv[threadIdx.x] || (v2 == sv[threadIdx.x] && i2 < si[threadIdx.x])) { sv[threadIdx.x] = v2; si[threadIdx.x] = i2; }
            }
            __syncthreads();
        }
        lv = sv[0];
        li = si[0];
        if (threadIdx.x == 0) {
            out_val[(long)blockIdx.x * k + j] = lv;
            out_idx[(long)blockIdx.x * k + j] = li;
        }
        __syncthreads();
    }
}

// The same result in ONE pass over the row for k <= 8: every thread keeps the best eight of its elements in registers (sorted,
// earlier index first among equal values -- a thread visits its indices in increasing order), then the block merges the 256 sorted
// lists through LDS: k rounds of "best head" with the (value desc, index asc) order.  (The k-pass kernel above read the row k times
// and ran a 256-wide LDS tree per pass: 74 us for 640 rows x 10 000 columns, k = 5.)
__global__ __launch_bounds__(256) void topk_rows_small_kernel(const float* __restrict__ x, int cols, long ld, int k,
                                                              float* __restrict__ out_val, int32_t* __restrict__ out_idx) {
    __shared__ float lv[8][256];
    __shared__ int li[8][256];
    __shared__ float wv[4];
    __shared__ int wi[4], wt[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* p = x + (long)blockIdx.x * ld;
    float tv[8];
    int ti[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { tv[j] = -INFINITY; ti[j] = 0x7fffffff; }
    auto offer = [&](float v, int c) {
        if (v > tv[7]) {  // (equal to the last kept value: the kept one has the smaller index)
            tv[7] = v; ti[7] = c;
#pragma unroll
            for (int j = 7; j > 0; --j)
                if (tv[j] > tv[j - 1]) {  // strict: an equal earlier element stays in front
                    const float a = tv[j]; tv[j] = tv[j - 1]; tv[j - 1] = a;
                    const int b = ti[j]; ti[j] = ti[j - 1]; ti[j - 1] = b;
                }
        }
    };
    if (((ld & 3) == 0) && ((((uintptr_t)x) & 15) == 0)) {
        const int c4 = cols >> 2;
        for (int q = tid; q < c4; q += 256) {
            const float4 v = *reinterpret_cast<const float4*>(p + 4 * q);
            offer(v.x, 4 * q); offer(v.y, 4 * q + 1); offer(v.z, 4 * q + 2); offer(v.w, 4 * q + 3);
        }
        for (int c = 4 * c4 + tid; c < cols; c += 256) offer(p[c], c);   // (ragged tail: indices above every vector index of this thread)
    } else {
        for (int c = tid; c < cols; c += 256) offer(p[c], c);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { lv[j][tid] = tv[j]; li[j][tid] = ti[j]; }
    int head = 0;
    for (int j = 0; j < k; ++j) {
        float bv = head < 8 ? lv[head][tid] : -INFINITY;
        int bi = head < 8 ? li[head][tid] : 0x7fffffff;
        int bt = tid;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float v2 = __shfl_xor(bv, o, 64);
            const int i2 = __shfl_xor(bi, o, 64), t2 = __shfl_xor(bt, o, 64);
            if (v2 > bv || (v2 == bv && i2 < bi)) { bv = v2; bi = i2; bt = t2; }
        }
        if (lane == 0) { wv[wave] = bv; wi[wave] = bi; wt[wave] = bt; }
        __syncthreads();
        bv = wv[0]; bi = wi[0]; bt = wt[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (wv[w] > bv || (wv[w] == bv && wi[w] < bi)) { bv = wv[w]; bi = wi[w]; bt = wt[w]; }
        if (tid == bt) ++head;
        if (tid == 0) {
            out_val[(long)blockIdx.x * k + j] = bv;
            out_idx[(long)blockIdx.x * k + j] = bi;
        }
        __syncthreads();
    }
}

// One decoder round's three row moves in one launch (beam search, vae_model/decoder.py:254-262: every new beam continues the LSTM state
// of its parent and feeds its last word): cg[r] = c[parent[r]], hg[r] = h[parent[r]] (H floats each) and, when xproj is given,
// gact[r] = xproj[tok[r]] (G floats: the word's input projection E.Wx + b, looked up instead of multiplied).  16-byte accesses.
__global__ __launch_bounds__(256) void beam_gather_kernel(const float4* __restrict__ c, const float4* __restrict__ h, const int32_t* __restrict__ parent,
                                                          int rows, int H4, float4* __restrict__ cg, float4* __restrict__ hg,
                                                          const float4* __restrict__ xproj, const int32_t* __restrict__ tok, int vocab, int G4,
                                                          float4* __restrict__ gact) {
    const long per = 2L * H4 + (xproj ? G4 : 0);
    const long total = (long)rows * per;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int r = (int)(i / per);
        const int e = (int)(i - (long)r * per);
        if (e < 2 * H4) {
            int pr = parent[r];
            pr = pr < 0 ? 0 : (pr >= rows ? rows - 1 : pr);
            if (e < H4) cg[(long)r * H4 + e] = c[(long)pr * H4 + e];
            else hg[(long)r * H4 + e - H4] = h[(long)pr * H4 + e - H4];
        } else {
            int id = tok[r];
            id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
            gact[(long)r * G4 + e - 2 * H4] = xproj[(long)id * G4 + e - 2 * H4];
        }
    }
}

// softmax + top-k of a row in ONE read of the logits (beam search, vae_model/decoder.py:248-276: probs = softmax(logits), then the
// beam_size most probable words by a stable sort on -p).  The probabilities are formed by exactly the expressions of
// softmax_rows_reg_kernel / softmax_rows_kernel (loss.hip: same maximum, same summation order, p = exp(l - max) * (1 / sum)) and offered
// to the per-thread lists of topk_rows_small_kernel in the same column order, so (p, index) out equal vc_softmax_rows_f32 followed by
// vc_topk_rows_f32 bit for bit -- without writing the [rows, V] probabilities (25.6 MB per round at 640 rows) and reading them back.
//
// The register form's selection: instead of every thread keeping a sorted list of its best eight of 40 -- an insertion path the wave
// takes whenever ANY of its lanes inserts, i.e. for nearly every element -- the workgroup first agrees on a threshold that at least k
// probabilities reach (the k-th largest of one wave's 64 thread maxima, the largest such value of the four waves), collects the few
// elements that reach it into an LDS list, and ONE wave picks the k best of those (value descending, index ascending: the order of the
// stable sort).  A row with more than TOPK_CAND elements at the threshold (many equal probabilities) takes the list path below.
constexpr int TOPK_CAND = 256;
template <bool REG>
__global__ __launch_bounds__(256) void softmax_topk_rows_kernel(const float* __restrict__ x, int V, long ld, int k,
                                                                float* __restrict__ out_val, int32_t* __restrict__ out_idx) {
    __shared__ float sh[4];
    __shared__ float lv[8][256];
    __shared__ int li[8][256];
    __shared__ float wv[4];
    __shared__ int wi[4], wt[4];
    __shared__ int ncand;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float tv[8];
    int ti[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { tv[j] = -INFINITY; ti[j] = 0x7fffffff; }
    auto offer = [&](float v, int c) {
        if (v > tv[7]) {
            tv[7] = v; ti[7] = c;
#pragma unroll
            for (int j = 7; j > 0; --j)
                if (tv[j] > tv[j - 1]) {
                    const float a = tv[j]; tv[j] = tv[j - 1]; tv[j - 1] = a;
                    const int b = ti[j]; ti[j] = ti[j - 1]; ti[j - 1] = b;
                }
        }
    };
    if (REG) {
        const float4* p = reinterpret_cast<const float4*>(x + (long)blockIdx.x * ld);
        const int Q = V >> 2;
        float4 r[12];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const int c = tid + 256 * i;
            if (c < Q) {
                r[i] = p[c];
                mx = fmaxf(fmaxf(mx, fmaxf(r[i].x, r[i].y)), fmaxf(r[i].z, r[i].w));
            }
        }
        mx = block_max<256>(mx, sh);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const int c = tid + 256 * i;
            if (c < Q) {
                r[i].x = __expf(r[i].x - mx); r[i].y = __expf(r[i].y - mx); r[i].z = __expf(r[i].z - mx); r[i].w = __expf(r[i].w - mx);
                s += (r[i].x + r[i].y) + (r[i].z + r[i].w);
            }
        }
        s = block_sum<256>(s, sh);
        const float inv = 1.f / s;
        {
            if (tid == 0) ncand = 0;
            float tm = -INFINITY;
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                const int c = tid + 256 * i;
                if (c < Q) {
                    r[i].x *= inv; r[i].y *= inv; r[i].z *= inv; r[i].w *= inv;
                    tm = fmaxf(fmaxf(tm, fmaxf(r[i].x, r[i].y)), fmaxf(r[i].z, r[i].w));
                }
            }
            float kth = -INFINITY;
            for (int j = 0; j < k; ++j) {   // the k-th largest thread maximum of this wave: k maxima, each knocked out of ONE lane
                kth = wave_max(tm);
                const unsigned long long holders = __ballot(tm == kth);
                if (lane == __builtin_ctzll(holders)) tm = -INFINITY;
            }
            if (lane == 0) wv[wave] = kth;
            __syncthreads();
            const float thr = fmaxf(fmaxf(wv[0], wv[1]), fmaxf(wv[2], wv[3]));
            auto cand = [&](float v, int c) {
                if (v >= thr) {
                    const int slot = atomicAdd(&ncand, 1);
                    if (slot < TOPK_CAND) { lv[0][slot] = v; li[0][slot] = c; }
                }
            };
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                const int c = tid + 256 * i;
                if (c < Q) { cand(r[i].x, 4 * c); cand(r[i].y, 4 * c + 1); cand(r[i].z, 4 * c + 2); cand(r[i].w, 4 * c + 3); }
            }
            __syncthreads();
            const int nc = ncand;
            if (nc <= TOPK_CAND) {
                if (wave != 0) return;
                float cv[TOPK_CAND / 64];
                int ci[TOPK_CAND / 64];
#pragma unroll
                for (int q = 0; q < TOPK_CAND / 64; ++q) {
                    const int sidx = lane + 64 * q;
                    cv[q] = sidx < nc ? lv[0][sidx] : -INFINITY;
                    ci[q] = sidx < nc ? li[0][sidx] : 0x7fffffff;
                }
                for (int j = 0; j < k; ++j) {
                    float bv = cv[0];
                    int bi = ci[0];
#pragma unroll
                    for (int q = 1; q < TOPK_CAND / 64; ++q)
                        if (cv[q] > bv || (cv[q] == bv && ci[q] < bi)) { bv = cv[q]; bi = ci[q]; }
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) {
                        const float v2 = __shfl_xor(bv, o, 64);
                        const int i2 = __shfl_xor(bi, o, 64);
                        if (v2 > bv || (v2 == bv && i2 < bi)) { bv = v2; bi = i2; }
                    }
#pragma unroll
                    for (int q = 0; q < TOPK_CAND / 64; ++q)
                        if (ci[q] == bi) { cv[q] = -INFINITY; ci[q] = 0x7fffffff; }   // (indices are unique: exactly one lane holds the winner)
                    if (lane == 0) {
                        out_val[(long)blockIdx.x * k + j] = bv;
                        out_idx[(long)blockIdx.x * k + j] = bi;
                    }
                }
                return;
            }
            __syncthreads();   // (the list path reuses lv / li)
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                const int c = tid + 256 * i;
                if (c < Q) { offer(r[i].x, 4 * c); offer(r[i].y, 4 * c + 1); offer(r[i].z, 4 * c + 2); offer(r[i].w, 4 * c + 3); }
            }
        }
    } else {
        const float* p = x + (long)blockIdx.x * ld;
        float mx = -INFINITY;
        for (int c = tid; c < V; c += 256) mx = fmaxf(mx, p[c]);
        mx = block_max<256>(mx, sh);
        float s = 0.f;
        for (int c = tid; c < V; c += 256) s += __expf(p[c] - mx);
        s = block_sum<256>(s, sh);
        const float inv = 1.f / s;
        for (int c = tid; c < V; c += 256) offer(__expf(p[c] - mx) * inv, c);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { lv[j][tid] = tv[j]; li[j][tid] = ti[j]; }
    int head = 0;
    for (int j = 0; j < k; ++j) {
        float bv = head < 8 ? lv[head][tid] : -INFINITY;
        int bi = head < 8 ? li[head][tid] : 0x7fffffff;
        int bt = tid;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float v2 = __shfl_xor(bv, o, 64);
            const int i2 = __shfl_xor(bi, o, 64), t2 = __shfl_xor(bt, o, 64);
            if (v2 > bv || (v2 == bv && i2 < bi)) { bv = v2; bi = i2; bt = t2; }
        }
        if (lane == 0) { wv[wave] = bv; wi[wave] = bi; wt[wave] = bt; }
        __syncthreads();
        bv = wv[0]; bi = wi[0]; bt = wt[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (wv[w] > bv || (wv[w] == bv && wi[w] < bi)) { bv = wv[w]; bi = wi[w]; bt = wt[w]; }
        if (tid == bt) ++head;
        if (tid == 0) {
            out_val[(long)blockIdx.x * k + j] = bv;
            out_idx[(long)blockIdx.x * k + j] = bi;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void fill_kernel(float* __restrict__ x, long n, float v) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) x[i] = v;
}

}  // namespace vc

using namespace vc;

extern "C" int vc_topk_rows_f32(void* stream, const float* x, long rows, int cols, long ld, int k, float* out_val,
                                int32_t* out_idx) {
    VC_CHECK_ARG(x && out_val && out_idx && rows >= 0 && cols > 0 && ld >= cols && k > 0 && k <= cols, "bad argument");
    if (rows == 0) return 0;
    if (k <= 8)
        hipLaunchKernelGGL(topk_rows_small_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, cols, ld, k, out_val, out_idx);
    else
        hipLaunchKernelGGL(topk_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, cols, ld, k, out_val, out_idx);
    VC_LAUNCH_CHECK();
    return 0;
}

extern "C" int vc_beam_gather_f32(void* stream, const float* c, const float* h, const int32_t* parent, int rows, int H, float* cg, float* hg,
                                  const float* xproj, const int32_t* tok, int vocab, int G, float* gact) {
    VC_CHECK_ARG(c && h && parent && cg && hg && rows > 0 && H > 0 && H % 4 == 0, "bad argument (H % 4 == 0)");
    VC_CHECK_ARG(!xproj || (tok && gact && vocab > 0 && G > 0 && G % 4 == 0), "xproj needs tok, gact, vocab and G % 4 == 0");
    VC_CHECK_ARG(((((uintptr_t)c | (uintptr_t)h | (uintptr_t)cg | (uintptr_t)hg | (uintptr_t)xproj | (uintptr_t)gact)) & 15) == 0, "pointers must be 16-byte aligned");
    const long total = (long)rows * (2L * (H / 4) + (xproj ? G / 4 : 0));
    hipLaunchKernelGGL(beam_gather_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float4*>(c),
                       reinterpret_cast<const float4*>(h), parent, rows, H / 4, reinterpret_cast<float4*>(cg), reinterpret_cast<float4*>(hg),
                       reinterpret_cast<const float4*>(xproj), tok, vocab, G / 4, reinterpret_cast<float4*>(gact));
    VC_LAUNCH_CHECK();
    return 0;
}

extern "C" int vc_softmax_topk_rows_f32(void* stream, const float* logits, long rows, int V, long ld, int k, float* out_p, int32_t* out_idx) {
    VC_CHECK_ARG(logits && out_p && out_idx && rows >= 0 && V > 0 && ld >= V && k > 0 && k <= 8 && k <= V, "bad argument (k <= 8)");
    if (rows == 0) return 0;
    // (the register form under the conditions vc_softmax_rows_f32 takes it: the probabilities then round alike)
    if ((V & 3) == 0 && V <= 12288 && (ld & 3) == 0 && (((uintptr_t)logits) & 15) == 0)
        hipLaunchKernelGGL(softmax_topk_rows_kernel<true>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, logits, V, ld, k, out_p, out_idx);
    else
        hipLaunchKernelGGL(softmax_topk_rows_kernel<false>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, logits, V, ld, k, out_p, out_idx);
    VC_LAUNCH_CHECK();
    return 0;
}

extern "C" int vc_fill_f32(void* stream, float* x, long n, float value) {
    VC_CHECK_ARG(x && n >= 0, "bad argument");
    if (n == 0) return 0;
    hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, n, value);
    VC_LAUNCH_CHECK();
    return 0;
}

extern "C" int vc_embedding_gather_f32(void* stream, const float* table, const int32_t* ids, long rows, int E, int vocab,
                                       float* out) {
    VC_CHECK_ARG(table && ids && out && rows >= 0 && E > 0 && vocab > 0, "bad argument");
    if (rows == 0) return 0;
    const bool vec = (E % 4 == 0) && (((uintptr_t)table | (uintptr_t)out) & 15) == 0;
    if (vec)
        hipLaunchKernelGGL(gather_kernel<true>, dim3(grid_for(rows * (E / 4))), dim3(256), 0, (hipStream_t)stream, table, ids, rows, E, vocab, out);
    else
        hipLaunchKernelGGL(gather_kernel<false>, dim3(grid_for(rows * E)), dim3(256), 0, (hipStream_t)stream, table, ids, rows, E, vocab, out);
    VC_LAUNCH_CHECK();
    return 0;
}

extern "C" int vc_embedding_scatter_add_f32(void* stream, float* dtable, const int32_t* ids, long rows, int E, int vocab,
                                            const float* dX) {
    VC_CHECK_ARG(dtable && ids && dX && rows >= 0 && E > 0 && vocab > 0, "bad argument");
    if (rows == 0) return 0;
    hipLaunchKernelGGL(scatter_add_kernel, dim3(grid_for(rows * E)), dim3(256), 0, (hipStream_t)stream, dtable, ids, rows, E, vocab, dX);
    VC_LAUNCH_CHECK();
    return 0;
}

extern "C" int vc_embedding_grad_sorted_f32(void* stream, float* dtable, const int32_t* order, const int32_t* seg_start,
                                           int E, int nrows, const float* dX) {
    VC_CHECK_ARG(dtable && seg_start && dX && E > 0 && nrows > 0, "bad argument");
    const bool vec = (E % 4 == 0) && ((((uintptr_t)dtable | (uintptr_t)dX) & 15) == 0);
    if (vec)
        hipLaunchKernelGGL(segment_grad_kernel<true>, dim3(grid_for((long)nrows * 64, 256, 8192)), dim3(256), 0, (hipStream_t)stream, dtable, order, seg_start, E, nrows, dX);
    else
        hipLaunchKernelGGL(segment_grad_kernel<false>, dim3(grid_for((long)nrows * 64, 256, 8192)), dim3(256), 0, (hipStream_t)stream, dtable, order, seg_start, E, nrows, dX);
    VC_LAUNCH_CHECK();
    return 0;
}

static int embidx_segs(long R) {
    long s = (R + 511) / 512;
    return (int)(s < 1 ? 1 : (s > 32 ? 32 : s));
}

extern "C" size_t vc_embedding_index_workspace_bytes(long R, int vocab) {
    return ((size_t)embidx_segs(R) + 3) * (size_t)vocab * sizeof(int32_t);  // cnt[segs][vocab] + totals + sub-segment counts + first slots
}

extern "C" size_t vc_embedding_index_max_subsegments(long R, int vocab, int chunk) {
    return chunk > 0 ? (size_t)(R / chunk + vocab) : 0;
}

// largest vocabulary the single-workgroup scan takes (its LDS table: 8 bytes per id + 8 KB of the 160 KB); callers with more ids build
// the index on the host (engine.embedding_grad_index) -- engine.set_batch does
extern "C" int vc_embedding_index_max_vocab(void) { return (160 * 1024 - 8192) / 8; }

extern "C" int vc_embedding_grad_index(void* stream, const int32_t* ids, long R, int vocab, int chunk, int32_t* order,
                                       int32_t* seg1, int32_t* seg2, int32_t* ws, size_t ws_bytes) {
    VC_CHECK_ARG(ids && order && seg1 && seg2 && R > 0 && vocab > 0 && chunk > 0 && R < (1L << 31), "bad argument");
    const int segs = embidx_segs(R);
    if (!ws || ws_bytes < ((size_t)segs + 3) * vocab * sizeof(int32_t))
        return fail(VC_EWORKSPACE, "%s: workspace too small (need vc_embedding_index_workspace_bytes)", __func__);
    VC_CHECK_ARG(vocab <= vc_embedding_index_max_vocab(), "vocabulary too large for the single-workgroup scan (vc_embedding_index_max_vocab)");
    int32_t* tot = ws + (size_t)segs * vocab;
    int32_t* nsb = tot + vocab;
    int32_t* starts = nsb + vocab;
    const long seg_len = ((R + segs - 1) / segs + 3) / 4 * 4;
    const int nsub_max = (int)(R / chunk + vocab);
    const dim3 grid(cdiv(vocab, 256), segs);
    hipLaunchKernelGGL(embidx_scan_ids_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, ids, R, vocab, seg_len, ws, order);
    VC_LAUNCH_CHECK();
    hipLaunchKernelGGL(embidx_totals_kernel, dim3(grid.x), dim3(256), 0, (hipStream_t)stream, ws, segs, vocab, chunk, tot, nsb);
    VC_LAUNCH_CHECK();
    if ((size_t)vocab * 8 + 8192 > 64 * 1024) {  // more than the default 64 KB of dynamic LDS
        const hipError_t e = hipFuncSetAttribute((const void*)embidx_scan_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)vocab * 8 + 8192));
        if (e != hipSuccess) return fail((int)e, "%s: hipFuncSetAttribute (%ld bytes of dynamic LDS) failed", __func__, (long)vocab * 8 + 8192);
    }
    hipLaunchKernelGGL(embidx_scan_kernel, dim3(1), dim3(1024), (size_t)vocab * 8 + 8192, (hipStream_t)stream, tot, nsb, vocab, starts, seg2);
    VC_LAUNCH_CHECK();
    hipLaunchKernelGGL(embidx_finish_kernel, dim3(grid.x), dim3(256), 0, (hipStream_t)stream, ws, segs, vocab, R, chunk, nsub_max, tot, starts, seg1, seg2);
    VC_LAUNCH_CHECK();
    hipLaunchKernelGGL(embidx_scan_ids_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, ids, R, vocab, seg_len, ws, order);
    VC_LAUNCH_CHECK();
    return 0;
}

extern "C" int vc_mark_rows_f32(void* stream, float* touched, const int32_t* ids, long n, int vocab) {
    VC_CHECK_ARG(touched && ids && n >= 0 && vocab > 0, "bad argument");
    if (n == 0) return 0;
    hipLaunchKernelGGL(mark_rows_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, touched, ids, n, vocab);
    VC_LAUNCH_CHECK();
    return 0;
}

static int colsum_chunks(long rows, int cols) {
    const int gx = cdiv(cols, 64);
    long chunks = rows / 64;
    const long cap = 2048 / gx > 1 ? 2048 / gx : 1;
    if (chunks > cap) chunks = cap;
    if (chunks > COLSUM_MAX_CHUNKS) chunks = COLSUM_MAX_CHUNKS;
    if (gx < 8 && chunks > 256) chunks = 256;  // narrow matrices: keep the serial second stage short
    if (chunks < 1) chunks = 1;
    return (int)chunks;
}

extern "C" size_t vc_colsum_workspace_bytes(long rows, int cols) {
    return (size_t)colsum_chunks(rows, cols) * cols * sizeof(float);
}

extern "C" int vc_colsum_f32(void* stream, const float* x, long rows, int cols, long ld, float* out, int accumulate,
                             float* ws, size_t ws_bytes) {
    VC_CHECK_ARG(x && out && rows >= 0 && cols > 0 && ld >= cols, "bad argument");
    const int chunks = colsum_chunks(rows, cols);
    if (!ws || ws_bytes < (size_t)chunks * cols * sizeof(float))
        return fail(VC_EWORKSPACE, "%s: workspace too small (need vc_colsum_workspace_bytes)", __func__);
    const bool vec = (cols % 4 == 0) && (ld % 4 == 0) && (((uintptr_t)x & 15) == 0);
    if (vec)
        hipLaunchKernelGGL(colsum_partial_kernel<true>, dim3(cdiv(cols, 64), chunks), dim3(256), 0, (hipStream_t)stream, x, rows, cols, ld, ws);
    else
        hipLaunchKernelGGL(colsum_partial_kernel<false>, dim3(cdiv(cols, 64), chunks), dim3(256), 0, (hipStream_t)stream, x, rows, cols, ld, ws);
    VC_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(cols, 64)), dim3(256), 0, (hipStream_t)stream, ws, chunks, cols, out, accumulate);
    VC_LAUNCH_CHECK();
    return 0;
}

extern "C" int vc_dropout_f32(void* stream, const float* x, const float* mask, float keep, long n, float* y) {
    VC_CHECK_ARG(x && mask && y && n >= 0 && keep > 0.f, "bad argument");
    if (n == 0) return 0;
    hipLaunchKernelGGL(dropout_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, mask, 1.0f / keep, n, y);
    VC_LAUNCH_CHECK();
    return 0;
}

extern "C" int vc_relu_bwd_f32(void* stream, const float* dy, const float* y, const float* mask, float keep, long n,
                               float* dx) {
    VC_CHECK_ARG(dy && y && dx && n >= 0 && keep > 0.f, "bad argument");
    if (n == 0) return 0;
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dy, y, mask, 1.0f / keep, n, dx);
    VC_LAUNCH_CHECK();
    return 0;
}

extern "C" int vc_tile_rows_f32(void* stream, const float* x, long B, int nc, int E, float* y) {
    VC_CHECK_ARG(x && y && B >= 0 && nc > 0 && E > 0, "bad argument");
    if (B == 0) return 0;
    hipLaunchKernelGGL(tile_rows_kernel, dim3(grid_for(B * nc * E)), dim3(256), 0, (hipStream_t)stream, x, B, nc, E, y);
    VC_LAUNCH_CHECK();
    return 0;
}

extern "C" int vc_segment_sum_rows_f32(void* stream, const float* y, long B, int nc, int E, float* x, int accumulate) {
    VC_CHECK_ARG(x && y && B >= 0 && nc > 0 && E > 0, "bad argument");
    if (B == 0) return 0;
    hipLaunchKernelGGL(segment_sum_kernel, dim3(grid_for(B * E)), dim3(256), 0, (hipStream_t)stream, y, B, nc, E, x, accumulate);
    VC_LAUNCH_CHECK();
    return 0;
}

extern "C" int vc_exp_f32(void* stream, const float* x, long n, float* y) {
    VC_CHECK_ARG(x && y && n >= 0, "bad argument");
    if (n == 0) return 0;
    hipLaunchKernelGGL(exp_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, n, y);
    VC_LAUNCH_CHECK();
    return 0;
}

extern "C" int vc_axpy_f32(void* stream, float a, const float* x, long n, float* y) {
    VC_CHECK_ARG(x && y && n >= 0, "bad argument");
    if (n == 0) return 0;
    hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, a, x, n, y);
    VC_LAUNCH_CHECK();
    return 0;
}

extern "C" int vc_reduce_sum_f32(void* stream, const float* x, long n, float scale, float* out, int accumulate) {
    VC_CHECK_ARG(x && out && n >= 0, "bad argument");
    hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, n, scale, out, accumulate);
    VC_LAUNCH_CHECK();
    return 0;
}

extern "C" int vc_count_nonzero_i32(void* stream, const int32_t* x, long n, float* out) {
    VC_CHECK_ARG(x && out && n >= 0, "bad argument");
    hipLaunchKernelGGL(count_nonzero_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, n, out);
    VC_LAUNCH_CHECK();
    return 0;
}

extern "C" int vc_argmax_rows_f32(void* stream, const float* x, long rows, int cols, long ld, int32_t* out) {
    VC_CHECK_ARG(x && out && rows >= 0 && cols > 0 && ld >= cols, "bad argument");
    if (rows == 0) return 0;
    hipLaunchKernelGGL(argmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, cols, ld, out);
    VC_LAUNCH_CHECK();
    return 0;
}
