// Beam-search bookkeeping on device (vae_model/decoder.py:238-300 + utils/top_n.py:4-43).
//
// One thread per image replays, for its image, exactly what the reference's Python does after every
// decoder step: walk the image's live beams in the order TopN.extract() returned them (the heap ARRAY order),
// walk each beam's top-`beam_size` words in descending probability, skip p < 1e-12, and push the extended
// caption into the image's `complete` (word == <EOS>, score = logprob / len**len_norm_f) or `partial`
// (score = logprob) TopN.  TopN is a min-heap keyed by score only, driven with heapq's heappush /
// heappushpop; ties are therefore resolved by heapq's sift order, which this file reproduces move for move
// (CPython Lib/heapq.py: _siftdown, _siftup).  Each word's log-probability is a float32 log (np.log of a float32
// probability), accumulated in double like the reference's Python / numpy float64 sums.
//
// Nothing here needs the host between decoder steps: the kernel also emits, for the next step, each new
// beam's parent row (whose LSTM state it continues) and its last token.
#include "common.h"
#include "vaecap.h"

namespace vc {

constexpr int BEAM_MAX = 16;

struct BeamItem {
    double score, logprob;
    int parent, tok, len, slot;
};

__device__ __forceinline__ bool item_lt(const BeamItem& a, const BeamItem& b) { return a.score < b.score; }

// A TopN heap of one image in the LDS, stored field by field with the IMAGE as the fastest index ([position][image]): the 32 walking
// threads of a workgroup step through their heaps side by side, and with one 32-byte item after another per image (512 bytes between
// two images' items) every access of the 32 lanes fell on the same banks -- a 32-way conflict on each of the ~50 item moves of a push.
constexpr int BEAM_IMAGES = 32;                 // images per workgroup
struct HeapRef {
    double* sc;     // [BEAM_MAX][BEAM_IMAGES] score, then logprob
    double* lp;
    int4* meta;     // [BEAM_MAX][BEAM_IMAGES] (parent, tok, len, slot)
    __device__ __forceinline__ BeamItem get(int pos) const {
        BeamItem it;
        it.score = sc[pos * BEAM_IMAGES];
        it.logprob = lp[pos * BEAM_IMAGES];
        const int4 m = meta[pos * BEAM_IMAGES];
        it.parent = m.x; it.tok = m.y; it.len = m.z; it.slot = m.w;
        return it;
    }
    __device__ __forceinline__ void put(int pos, const BeamItem& it) const {
        sc[pos * BEAM_IMAGES] = it.score;
        lp[pos * BEAM_IMAGES] = it.logprob;
        meta[pos * BEAM_IMAGES] = make_int4(it.parent, it.tok, it.len, it.slot);
    }
    __device__ __forceinline__ double score(int pos) const { return sc[pos * BEAM_IMAGES]; }
};

// CPython Lib/heapq.py _siftdown / _siftup, move for move (comparisons by score only)
__device__ __forceinline__ void sift_down(const HeapRef& heap, int startpos, int pos) {
    const BeamItem newitem = heap.get(pos);
    while (pos > startpos) {
        const int parentpos = (pos - 1) >> 1;
        if (newitem.score < heap.score(parentpos)) {
            heap.put(pos, heap.get(parentpos));
            pos = parentpos;
            continue;
        }
        break;
    }
    heap.put(pos, newitem);
}

__device__ __forceinline__ void sift_up(const HeapRef& heap, int n, int pos) {
    const int startpos = pos;
    const BeamItem newitem = heap.get(pos);
    int childpos = 2 * pos + 1;
    while (childpos < n) {
        const int rightpos = childpos + 1;
        if (rightpos < n && !(heap.score(childpos) < heap.score(rightpos))) childpos = rightpos;
        heap.put(pos, heap.get(childpos));
        pos = childpos;
        childpos = 2 * pos + 1;
    }
    heap.put(pos, newitem);
    sift_down(heap, startpos, pos);
}

// TopN.push: returns the slot field of the item that left the heap (the popped root, or the rejected newcomer), -1 if none
__device__ __forceinline__ int topn_push(const HeapRef& heap, int& count, int cap, const BeamItem& item) {
    if (count < cap) {
        heap.put(count, item);
        ++count;
        sift_down(heap, 0, count - 1);
        return -1;
    }
    if (count > 0 && heap.score(0) < item.score) {
        const int freed = heap.get(0).slot;
        heap.put(0, item);
        sift_up(heap, count, 0);
        return freed;
    }
    return item.slot;
}

struct BeamArgs {
    int B, n, k, Lmax, eos;
    double len_norm_f;
    const float* tv;
    const int32_t* ti;
    int32_t *pcount, *ccount, *p_len, *c_len, *c_slot, *c_free;
    double *p_score, *p_logprob, *c_score, *c_logprob;
    const int32_t* sent_cur;
    int32_t *sent_next, *c_sent, *parent, *tok;
};

// The two heaps of an image live in LDS (32 images x 2 x 16 items x 32 B = 32 KB, HeapRef above): as private arrays their run-time
// indexing went through scratch memory, one L2 round trip per heap move (65 us per decoder step for 128 images x 5 beams).
// Round 6: the heap walk stays one thread per image (it IS sequential: heapq's sift order decides ties), but the token copies it
// used to do itself -- a kept beam's sentence into the next round's buffer, a finished caption into its pool slot: up to
// beam x (length - 1) dependent load / store pairs per thread, 41 us per round at 30 tokens -- are only RECORDED by the walk and carried
// out afterwards by eight threads per image (a workgroup = 32 images = 256 threads, thread t of an image copies tokens t, t + 8, ...).
// A finished caption is recorded PER POOL SLOT: a slot that was freed and taken again within the round keeps its last writer's record,
// which is the caption the sequential code left there.
constexpr int BEAM_THREADS = 8 * BEAM_IMAGES;   // 256
constexpr int BEAM_SLOTS = BEAM_MAX + 1;        // pool slots of finished captions per image


__global__ __launch_bounds__(BEAM_THREADS) void beam_update_kernel(BeamArgs a) {
    __shared__ double h_sc[2][BEAM_MAX][BEAM_IMAGES], h_lp[2][BEAM_MAX][BEAM_IMAGES];
    __shared__ int4 h_meta[2][BEAM_MAX][BEAM_IMAGES];
    __shared__ int part_n[BEAM_IMAGES];
    __shared__ short crec_src[BEAM_IMAGES][BEAM_SLOTS], crec_len0[BEAM_IMAGES][BEAM_SLOTS];   // len0 < 0: no caption recorded for the slot this round
    __shared__ int crec_tok[BEAM_IMAGES][BEAM_SLOTS];
    // the round's candidates (top-k probabilities and words of every live beam) and the beams' running sums, fetched by all 256 threads
    // (eight per image, coalesced) before the walk: read one by one inside it, each was a dependent global load -- 2 x beam x k round
    // trips of ~0.5 us per image and round (25 of the kernel's 31 us at beam 5).  Up to beam x k = 64 candidates; beyond, the walk reads global memory.
    constexpr int BEAM_PRE = 64;
    __shared__ float pre_p[BEAM_IMAGES][BEAM_PRE];
    __shared__ int pre_i[BEAM_IMAGES][BEAM_PRE];
    __shared__ double pre_lp[BEAM_IMAGES][BEAM_MAX];
    __shared__ int pre_len[BEAM_IMAGES][BEAM_MAX];
    const int n = a.n, L = a.Lmax;
    const bool pre = n * a.k <= BEAM_PRE;
    {
        const int li = threadIdx.x >> 3, sub = threadIdx.x & 7, b = blockIdx.x * BEAM_IMAGES + li;
        if (b < a.B) {
            if (pre)
                for (int q = sub; q < n * a.k; q += 8) {
                    pre_p[li][q] = a.tv[(long)b * n * a.k + q];
                    pre_i[li][q] = a.ti[(long)b * n * a.k + q];
                }
            for (int q = sub; q < n; q += 8) {
                pre_lp[li][q] = a.p_logprob[(long)b * n + q];
                pre_len[li][q] = a.p_len[(long)b * n + q];
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < BEAM_IMAGES) {   // ---- the walk: one thread per image
        const int li = threadIdx.x, b = blockIdx.x * BEAM_IMAGES + li;
        part_n[li] = 0;
        for (int q = 0; q < BEAM_SLOTS; ++q) crec_len0[li][q] = -1;
        if (b < a.B) {
            const int np = a.pcount[b];
            for (int j = 0; j < n; ++j) {  // defaults for slots that stay empty: continue row b*n with token 0 (ignored)
                a.parent[b * n + j] = b * n;
                a.tok[b * n + j] = 0;
            }
            if (np != 0) {  // (0: every beam of this image has ended)
                const HeapRef part{&h_sc[0][0][li], &h_lp[0][0][li], &h_meta[0][0][li]};
                const HeapRef comp{&h_sc[1][0][li], &h_lp[1][0][li], &h_meta[1][0][li]};
                int hn = 0, cn = a.ccount[b];
                for (int j = 0; j < cn; ++j) {
                    BeamItem it;
                    it.score = a.c_score[b * n + j];
                    it.logprob = a.c_logprob[b * n + j];
                    it.len = a.c_len[b * n + j];
                    it.slot = a.c_slot[b * n + j];
                    it.parent = it.tok = 0;
                    comp.put(j, it);
                }
                int freemask = a.c_free[b];
                for (int i = 0; i < np; ++i) {
                    const long row = (long)b * n + i;
                    const double lp0 = pre_lp[li][i];
                    const int len0 = pre_len[li][i];
                    for (int j = 0; j < a.k; ++j) {
                        const float pw = pre ? pre_p[li][i * a.k + j] : a.tv[row * a.k + j];
                        if ((double)pw < 1e-12) continue;  // decoder.py:279: float32 p against the Python float 1e-12
                        BeamItem it;
                        it.tok = pre ? pre_i[li][i * a.k + j] : a.ti[row * a.k + j];
                        it.parent = i;
                        it.len = len0 + 1;
                        it.logprob = lp0 + (double)logf(pw);  // decoder.py:282: np.log of a float32 is a float32; the SUM is a float64
                        it.score = it.logprob;
                        it.slot = -1;
                        if (it.tok == a.eos) {
                            if (a.len_norm_f > 0) it.score = it.logprob / pow((double)it.len, a.len_norm_f);
                            // take a free pool slot, record the caption for it, give the slot back if the heap does not keep it
                            int s = 0;
                            while (!((freemask >> s) & 1)) ++s;
                            freemask &= ~(1 << s);
                            it.slot = s;
                            crec_src[li][s] = (short)i; crec_len0[li][s] = (short)len0; crec_tok[li][s] = it.tok;
                            const int freed = topn_push(comp, cn, n, it);
                            if (freed >= 0) freemask |= 1 << freed;
                        } else {
                            topn_push(part, hn, n, it);
                        }
                    }
                }
                for (int j = 0; j < hn; ++j) {
                    const long o = (long)b * n + j;
                    const BeamItem it = part.get(j);
                    a.p_score[o] = it.score;
                    a.p_logprob[o] = it.logprob;
                    a.p_len[o] = it.len;
                    a.parent[o] = b * n + it.parent;
                    a.tok[o] = it.tok;
                }
                a.pcount[b] = hn;
                for (int j = 0; j < cn; ++j) {
                    const long o = (long)b * n + j;
                    const BeamItem it = comp.get(j);
                    a.c_score[o] = it.score;
                    a.c_logprob[o] = it.logprob;
                    a.c_len[o] = it.len;
                    a.c_slot[o] = it.slot;
                }
                a.ccount[b] = cn;
                a.c_free[b] = freemask;
                part_n[li] = hn;
            }
        }
    }
    __syncthreads();
    // ---- the copies: eight threads per image
    const int li = threadIdx.x >> 3, sub = threadIdx.x & 7, b = blockIdx.x * BEAM_IMAGES + li;
    if (b >= a.B) return;
    const int32_t* cur = a.sent_cur + (long)b * n * L;
    for (int q = 0; q <= n; ++q) {   // finished captions, per pool slot
        const int len0 = crec_len0[li][q], i = crec_src[li][q];
        if (len0 < 0) continue;
        int32_t* dst = a.c_sent + ((long)b * (n + 1) + q) * L;
        for (int t = sub; t <= len0; t += 8) dst[t] = t < len0 ? cur[i * L + t] : crec_tok[li][q];
    }
    const int hn = part_n[li];
    int32_t* nxt = a.sent_next + (long)b * n * L;
    for (int j = 0; j < hn; ++j) {
        const int4 m = h_meta[0][j][li];   // (parent, tok, len, slot) of the kept beam in heap-array position j
        const int len = m.z, src = m.x;
        for (int t = sub; t < len; t += 8) nxt[j * L + t] = t < len - 1 ? cur[src * L + t] : m.y;
    }
}

// greedy / sampled decoding (vae_model/decoder.py:186-194: the loop ends at the stop word): done[b] |= (tok[b] == eos); pending[0] = rows not done
// yet.  One workgroup; lets a captured chunk of decoder steps report "has every image emitted <EOS>" in 4 bytes.
__global__ __launch_bounds__(256) void eos_track_kernel(const int32_t* __restrict__ tok, int B, int eos, int32_t* __restrict__ done,
                                                        float* __restrict__ pending) {
    __shared__ float sh[4];
    float open = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) {
        const int d = done[b] | (tok[b] == eos ? 1 : 0);
        done[b] = d;
        open += d ? 0.f : 1.f;
    }
    open = block_sum<256>(open, sh);
    if (threadIdx.x == 0) pending[0] = open;
}

}  // namespace vc

extern "C" int vc_eos_track_i32(void* stream, const int32_t* tok, int B, int eos, int32_t* done, float* pending) {
    using namespace vc;
    VC_CHECK_ARG(tok && done && pending && B > 0, "bad argument");
    hipLaunchKernelGGL(eos_track_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, tok, B, eos, done, pending);
    VC_LAUNCH_CHECK();
    return 0;
}

extern "C" int vc_beam_update(void* stream, int B, int beam, int Lmax, int eos, double len_norm_f, const float* top_p,
                              const int32_t* top_i, int32_t* pcount, int32_t* ccount, double* p_score, double* p_logprob,
                              int32_t* p_len, const int32_t* sent_cur, int32_t* sent_next, double* c_score, double* c_logprob,
                              int32_t* c_len, int32_t* c_slot, int32_t* c_free, int32_t* c_sent, int32_t* parent, int32_t* tok) {
    using namespace vc;
    VC_CHECK_ARG(B > 0 && beam > 0 && beam <= BEAM_MAX && Lmax > 1, "beam size must be 1..16");
    VC_CHECK_ARG(top_p && top_i && pcount && ccount && p_score && p_logprob && p_len && sent_cur && sent_next && c_score &&
                 c_logprob && c_len && c_slot && c_free && c_sent && parent && tok, "null pointer");
    BeamArgs a;
    a.B = B; a.n = beam; a.k = beam; a.Lmax = Lmax; a.eos = eos; a.len_norm_f = len_norm_f;
    a.tv = top_p; a.ti = top_i; a.pcount = pcount; a.ccount = ccount; a.p_len = p_len; a.c_len = c_len; a.c_slot = c_slot;
    a.c_free = c_free; a.p_score = p_score; a.p_logprob = p_logprob; a.c_score = c_score; a.c_logprob = c_logprob;
    a.sent_cur = sent_cur; a.sent_next = sent_next; a.c_sent = c_sent; a.parent = parent; a.tok = tok;
    hipLaunchKernelGGL(beam_update_kernel, dim3(cdiv(B, BEAM_IMAGES)), dim3(BEAM_THREADS), 0, (hipStream_t)stream, a);
    VC_LAUNCH_CHECK();
    return 0;
}
