// Beam-search bookkeeping on device (vae_model/decoder.py:238-300 + utils/top_n.py:4-43).
//
// One wave per image replays, for its image, exactly what the reference's Python does after every
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

// ---- a value that every lane of the wave holds alike, and arrays spread over the lanes (element p in lane p)
__device__ __forceinline__ int lane_get(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ int lane_set(int old, int l, int x) { return (int)threadIdx.x == l ? x : old; }   // (x: the same in every lane)
__device__ __forceinline__ double lane_get(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double lane_set(double old, int l, double x) {
    return (int)threadIdx.x == l ? x : old;
}

// A TopN heap of one image IN REGISTERS: one wave works on one image, and heap position p is lane p of six registers.  The walk below
// is sequential (heapq's sift order decides ties) and every lane runs it alike; `heap[pos]` with the wave-uniform pos is a
// v_readlane / a one-lane select -- a few cycles -- where the LDS heaps of rounds 4-6 paid one ~120-cycle round trip per dependent access,
// ~20 of them per push (1.2 us per candidate, 33-40 us per round at beam 5: the longest latency-bound kernel of a decode round).
struct WaveHeap {
    double sc, lp;
    int par, tok, len, slot;
    __device__ __forceinline__ BeamItem get(int pos) const {
        BeamItem it;
        it.score = lane_get(sc, pos); it.logprob = lane_get(lp, pos);
        it.parent = lane_get(par, pos); it.tok = lane_get(tok, pos); it.len = lane_get(len, pos); it.slot = lane_get(slot, pos);
        return it;
    }
    __device__ __forceinline__ void put(int pos, const BeamItem& it) {
        sc = lane_set(sc, pos, it.score); lp = lane_set(lp, pos, it.logprob);
        par = lane_set(par, pos, it.parent); tok = lane_set(tok, pos, it.tok); len = lane_set(len, pos, it.len); slot = lane_set(slot, pos, it.slot);
    }
    __device__ __forceinline__ double score(int pos) const { return lane_get(sc, pos); }
};

// CPython Lib/heapq.py _siftdown / _siftup, move for move (comparisons by score only).  `newitem` is the item heapq has just stored at
// `pos` (heappush: appended at the end; heappushpop: written over the root): it is carried in scalars and stored once, where it settles.
__device__ __forceinline__ void sift_down(WaveHeap& heap, int startpos, int pos, const BeamItem& newitem) {
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

__device__ __forceinline__ void sift_up(WaveHeap& heap, int n, int pos, const BeamItem& newitem) {
    const int startpos = pos;
    int childpos = 2 * pos + 1;
    while (childpos < n) {
        const int rightpos = childpos + 1;
        if (rightpos < n && !(heap.score(childpos) < heap.score(rightpos))) childpos = rightpos;
        heap.put(pos, heap.get(childpos));
        pos = childpos;
        childpos = 2 * pos + 1;
    }
    sift_down(heap, startpos, pos, newitem);
}

// TopN.push: returns the slot field of the item that left the heap (the popped root, or the rejected newcomer), -1 if none
__device__ __forceinline__ int topn_push(WaveHeap& heap, int& count, int cap, const BeamItem& item) {
    if (count < cap) {
        ++count;
        sift_down(heap, 0, count - 1, item);
        return -1;
    }
    if (count > 0 && heap.score(0) < item.score) {
        const int freed = lane_get(heap.slot, 0);
        sift_up(heap, count, 0, item);
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

// One wave per image.  Per round:
//   1. in parallel, lane q prepares candidate q (beam q / k, its j-th word): float32 log of the word's probability added to the beam's
//      float64 running sum, the length-normalised score if the word is <EOS> -- the transcendental work of the whole round at once;
//   2. the walk over the candidates in the reference's order (beams in heap-array order, words by descending probability), heaps in
//      registers (WaveHeap): a kept beam's sentence and a finished caption are only RECORDED (parent + last word; per pool slot);
//   3. the token copies by the 64 lanes: each kept beam's sentence into the next round's buffer, each recorded caption into its pool
//      slot (a slot freed and taken again within the round keeps its last writer's record, which is the caption the sequential
//      code left there).
__global__ __launch_bounds__(64) void beam_update_kernel(BeamArgs a) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int n = a.n, k = a.k, L = a.Lmax;
    const int np = __builtin_amdgcn_readfirstlane(a.pcount[b]);
    if (lane < n) {   // defaults for slots that stay empty: continue row b*n with token 0 (ignored)
        a.parent[b * n + lane] = b * n;
        a.tok[b * n + lane] = 0;
    }
    if (np == 0) return;   // every beam of this image has ended
    WaveHeap part{0.0, 0.0, 0, 0, 0, -1}, comp{0.0, 0.0, 0, 0, 0, -1};
    int hn = 0, cn = __builtin_amdgcn_readfirstlane(a.ccount[b]);
    if (lane < cn) {
        comp.sc = a.c_score[b * n + lane];
        comp.lp = a.c_logprob[b * n + lane];
        comp.len = a.c_len[b * n + lane];
        comp.slot = a.c_slot[b * n + lane];
    }
    int freemask = __builtin_amdgcn_readfirstlane(a.c_free[b]);
    int rec_src = 0, rec_len0 = -1, rec_tok = 0;   // lane s: the caption recorded for pool slot s this round (len0 < 0: none)
    const int total = np * k;
    for (int base = 0; base < total; base += 64) {
        // ---- 1. candidate base + lane
        const int q = base + lane;
        const bool have = q < total;
        const int i_q = have ? q / k : 0;
        const long row = (long)b * n + i_q;
        const float pw = have ? a.tv[(long)b * n * k + q] : 0.f;
        const int tok_q = have ? a.ti[(long)b * n * k + q] : 0;
        const int len0_q = a.p_len[row];
        const double lp_q = a.p_logprob[row] + (double)logf(pw);   // decoder.py:282: np.log of a float32 is a float32; the SUM is a float64
        double sc_q = lp_q;
        if (tok_q == a.eos && a.len_norm_f > 0) sc_q = lp_q / pow((double)(len0_q + 1), a.len_norm_f);
        const int skip_q = (!have || (double)pw < 1e-12) ? 1 : 0;   // decoder.py:279: float32 p against the Python float 1e-12
        // ---- 2. the walk
        const int cnt = total - base < 64 ? total - base : 64;
        for (int c = 0; c < cnt; ++c) {
            if (lane_get(skip_q, c)) continue;
            BeamItem it;
            it.tok = lane_get(tok_q, c);
            it.parent = (base + c) / k;
            const int len0 = lane_get(len0_q, c);
            it.len = len0 + 1;
            it.logprob = lane_get(lp_q, c);
            it.score = lane_get(sc_q, c);
            it.slot = -1;
            if (it.tok == a.eos) {
                // take a free pool slot, record the caption for it, give the slot back if the heap does not keep it
                const int s = __builtin_ctz(freemask);
                freemask &= ~(1 << s);
                it.slot = s;
                rec_src = lane_set(rec_src, s, it.parent); rec_len0 = lane_set(rec_len0, s, len0); rec_tok = lane_set(rec_tok, s, it.tok);
                const int freed = topn_push(comp, cn, n, it);
                if (freed >= 0) freemask |= 1 << freed;
            } else {
                topn_push(part, hn, n, it);
            }
        }
    }
    if (lane < hn) {
        const long o = (long)b * n + lane;
        a.p_score[o] = part.sc;
        a.p_logprob[o] = part.lp;
        a.p_len[o] = part.len;
        a.parent[o] = b * n + part.par;
        a.tok[o] = part.tok;
    }
    if (lane < cn) {
        const long o = (long)b * n + lane;
        a.c_score[o] = comp.sc;
        a.c_logprob[o] = comp.lp;
        a.c_len[o] = comp.len;
        a.c_slot[o] = comp.slot;
    }
    if (lane == 0) {
        a.pcount[b] = hn;
        a.ccount[b] = cn;
        a.c_free[b] = freemask;
    }
    // ---- 3. the copies
    const int32_t* cur = a.sent_cur + (long)b * n * L;
    for (int s = 0; s <= n; ++s) {   // finished captions, per pool slot
        const int len0 = lane_get(rec_len0, s);
        if (len0 < 0) continue;
        const int i = lane_get(rec_src, s), tk = lane_get(rec_tok, s);
        int32_t* dst = a.c_sent + ((long)b * (n + 1) + s) * L;
        for (int t = lane; t <= len0; t += 64) dst[t] = t < len0 ? cur[i * L + t] : tk;
    }
    int32_t* nxt = a.sent_next + (long)b * n * L;
    for (int j = 0; j < hn; ++j) {
        const int len = lane_get(part.len, j), src = lane_get(part.par, j), tk = lane_get(part.tok, j);
        for (int t = lane; t < len; t += 64) nxt[j * L + t] = t < len - 1 ? cur[src * L + t] : tk;
    }
}

// ---- beam x beam <= 64 (beam <= 8): every candidate of the round has a lane of its own for the whole kernel, so the heaps hold
// (score, id) PAIRS -- three registers per move instead of eight -- and an item's other fields are read where they already are: id < 64
// is candidate id (lane id of the registers step 1 filled), id >= 64 a complete caption carried in from earlier rounds (lane id - 64 of
// the registers loaded from c_*).  Same pushes, same sift moves, same order; what leaves the kernel is gathered by id at the end.
struct SlimHeap {
    double sc;
    int id;
    __device__ __forceinline__ double score(int pos) const { return lane_get(sc, pos); }
    __device__ __forceinline__ void move(int dst, int src) {   // heap[dst] = heap[src]
        const double s = lane_get(sc, src);
        const int i = lane_get(id, src);
        sc = lane_set(sc, dst, s); id = lane_set(id, dst, i);
    }
    __device__ __forceinline__ void put(int pos, double s, int i) { sc = lane_set(sc, pos, s); id = lane_set(id, pos, i); }
    // CPython Lib/heapq.py _siftdown / _siftup, move for move, the new item (s, i) carried in scalars
    __device__ __forceinline__ void sift_down(int startpos, int pos, double s, int i) {
        while (pos > startpos) {
            const int parentpos = (pos - 1) >> 1;
            if (s < score(parentpos)) {
                move(pos, parentpos);
                pos = parentpos;
                continue;
            }
            break;
        }
        put(pos, s, i);
    }
    __device__ __forceinline__ void sift_up(int n, int pos, double s, int i) {
        const int startpos = pos;
        int childpos = 2 * pos + 1;
        while (childpos < n) {
            const int rightpos = childpos + 1;
            if (rightpos < n && !(score(childpos) < score(rightpos))) childpos = rightpos;
            move(pos, childpos);
            pos = childpos;
            childpos = 2 * pos + 1;
        }
        sift_down(startpos, pos, s, i);
    }
    // TopN.push: the id that left the heap (the popped root, or the rejected newcomer), -1 if none
    __device__ __forceinline__ int push(int& count, int cap, double s, int i) {
        if (count < cap) {
            ++count;
            sift_down(0, count - 1, s, i);
            return -1;
        }
        if (count > 0 && score(0) < s) {
            const int out = lane_get(id, 0);
            sift_up(count, 0, s, i);
            return out;
        }
        return i;
    }
};

__global__ __launch_bounds__(64) void beam_update_slim_kernel(BeamArgs a) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int n = a.n, k = a.k, L = a.Lmax;
    const int np = __builtin_amdgcn_readfirstlane(a.pcount[b]);
    if (lane < n) {   // defaults for slots that stay empty: continue row b*n with token 0 (ignored)
        a.parent[b * n + lane] = b * n;
        a.tok[b * n + lane] = 0;
    }
    if (np == 0) return;   // every beam of this image has ended
    int hn = 0, cn = __builtin_amdgcn_readfirstlane(a.ccount[b]);
    // complete captions carried in (lane j < cn): they start as the complete heap, ids 64 + j
    double cc_lp = 0.0;
    int cc_len = 0, cc_slot = -1;
    SlimHeap part{0.0, 0}, comp{0.0, 0};
    if (lane < cn) {
        comp.sc = a.c_score[b * n + lane];
        comp.id = 64 + lane;
        cc_lp = a.c_logprob[b * n + lane];
        cc_len = a.c_len[b * n + lane];
        cc_slot = a.c_slot[b * n + lane];
    }
    int freemask = __builtin_amdgcn_readfirstlane(a.c_free[b]);
    // ---- 1. candidate `lane`: beam lane / k, its (lane % k)-th word
    const int total = np * k;
    const bool have = lane < total;
    const int i_q = have ? lane / k : 0;
    const long row = (long)b * n + i_q;
    const float pw = have ? a.tv[(long)b * n * k + lane] : 0.f;
    const int tok_q = have ? a.ti[(long)b * n * k + lane] : 0;
    const int len0_q = a.p_len[row];
    const double lp_q = a.p_logprob[row] + (double)logf(pw);   // decoder.py:282: np.log of a float32 is a float32; the SUM is a float64
    double sc_q = lp_q;
    if (tok_q == a.eos && a.len_norm_f > 0) sc_q = lp_q / pow((double)(len0_q + 1), a.len_norm_f);
    const int skip_q = (!have || (double)pw < 1e-12) ? 1 : 0;   // decoder.py:279: float32 p against the Python float 1e-12
    int slot_q = -1;     // pool slot of candidate `lane`, if it is a finished caption the walk gave one
    int rec_cand = -1;   // lane s: the candidate whose caption pool slot s receives this round (-1: none)
    // ---- 2. the walk
    for (int c = 0; c < total; ++c) {
        if (lane_get(skip_q, c)) continue;
        const double s = lane_get(sc_q, c);
        if (lane_get(tok_q, c) == a.eos) {
            // take a free pool slot, record the caption for it, give the slot back if the heap does not keep it
            const int sl = __builtin_ctz(freemask);
            freemask &= ~(1 << sl);
            slot_q = lane_set(slot_q, c, sl);
            rec_cand = lane_set(rec_cand, sl, c);
            const int out = comp.push(cn, n, s, c);
            if (out >= 0) freemask |= 1 << (out < 64 ? lane_get(slot_q, out) : lane_get(cc_slot, out - 64));
        } else {
            part.push(hn, n, s, c);
        }
    }
    // ---- what the heaps hold, gathered by id (lane j: heap position j)
    {   // (the exchanges run with every lane active; the stores are per heap position)
        const int id = lane < hn ? part.id : 0;
        const double lp = __shfl(lp_q, id, 64);
        const int len0 = __shfl(len0_q, id, 64), tk = __shfl(tok_q, id, 64);
        if (lane < hn) {
            const long o = (long)b * n + lane;
            a.p_score[o] = part.sc;
            a.p_logprob[o] = lp;
            a.p_len[o] = len0 + 1;
            a.parent[o] = b * n + id / k;
            a.tok[o] = tk;
        }
    }
    {
        const int id = lane < cn ? comp.id : 0;
        const int cq = id < 64 ? id : 0, cj = id < 64 ? 0 : id - 64;
        const double lp_new = __shfl(lp_q, cq, 64), lp_old = __shfl(cc_lp, cj, 64);
        const int len_new = __shfl(len0_q, cq, 64) + 1, len_old = __shfl(cc_len, cj, 64);
        const int slot_new = __shfl(slot_q, cq, 64), slot_old = __shfl(cc_slot, cj, 64);
        if (lane < cn) {
            const long o = (long)b * n + lane;
            a.c_score[o] = comp.sc;
            a.c_logprob[o] = id < 64 ? lp_new : lp_old;
            a.c_len[o] = id < 64 ? len_new : len_old;
            a.c_slot[o] = id < 64 ? slot_new : slot_old;
        }
    }
    if (lane == 0) {
        a.pcount[b] = hn;
        a.ccount[b] = cn;
        a.c_free[b] = freemask;
    }
    // ---- 3. the copies
    const int32_t* cur = a.sent_cur + (long)b * n * L;
    for (int sl = 0; sl <= n; ++sl) {   // finished captions, per pool slot
        const int c = lane_get(rec_cand, sl);
        if (c < 0) continue;
        const int len0 = lane_get(len0_q, c), i = c / k;
        int32_t* dst = a.c_sent + ((long)b * (n + 1) + sl) * L;
        for (int t = lane; t <= len0; t += 64) dst[t] = t < len0 ? cur[i * L + t] : a.eos;
    }
    int32_t* nxt = a.sent_next + (long)b * n * L;
    for (int j = 0; j < hn; ++j) {
        const int id = lane_get(part.id, j);
        const int len = lane_get(len0_q, id) + 1, src = id / k, tk = lane_get(tok_q, id);
        for (int t = lane; t < len; t += 64) nxt[j * L + t] = t < len - 1 ? cur[src * L + t] : tk;
    }
}

// vae_model/decoder.py:238-247 for every image at once: the state the first round's vc_beam_update reads (twenty fills, two state
// gathers and two index ramps as torch / library launches before: ~0.2 ms of a 5 ms call at 128 images)
__global__ __launch_bounds__(256) void beam_init_kernel(BeamArgs a, int bos, int H, const float* __restrict__ c_in, const float* __restrict__ h_in,
                                                        float* __restrict__ c_out, float* __restrict__ h_out, int32_t* __restrict__ sent_cur) {
    const long M = (long)a.B * a.n, stride = (long)gridDim.x * 256, t0 = (long)blockIdx.x * 256 + threadIdx.x;
    for (long i = t0; i < M * H; i += stride) {
        const long src = (i / H) / a.n * H + i % H;
        c_out[i] = c_in[src];
        h_out[i] = h_in[src];
    }
    for (long i = t0; i < M * a.Lmax; i += stride) {
        sent_cur[i] = bos;
        a.sent_next[i] = 0;
    }
    for (long i = t0; i < (long)a.B * (a.n + 1) * a.Lmax; i += stride) a.c_sent[i] = 0;
    for (long i = t0; i < M; i += stride) {
        a.p_score[i] = 0.0; a.p_logprob[i] = 0.0; a.p_len[i] = 1;
        a.c_score[i] = 0.0; a.c_logprob[i] = 0.0; a.c_len[i] = 0; a.c_slot[i] = 0;
        a.parent[i] = (int)i; a.tok[i] = bos;
    }
    for (long i = t0; i < a.B; i += stride) {
        a.pcount[i] = 1; a.ccount[i] = 0; a.c_free[i] = (1 << (a.n + 1)) - 1;
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

extern "C" int vc_beam_init(void* stream, int B, int beam, int Lmax, int bos, int H, const float* c_in, const float* h_in, float* c_out,
                            float* h_out, int32_t* pcount, int32_t* ccount, double* p_score, double* p_logprob, int32_t* p_len,
                            int32_t* sent_cur, int32_t* sent_next, double* c_score, double* c_logprob, int32_t* c_len, int32_t* c_slot,
                            int32_t* c_free, int32_t* c_sent, int32_t* parent, int32_t* tok) {
    using namespace vc;
    VC_CHECK_ARG(B > 0 && beam > 0 && beam <= BEAM_MAX && Lmax > 1 && H > 0, "beam size must be 1..16");
    VC_CHECK_ARG(c_in && h_in && c_out && h_out && pcount && ccount && p_score && p_logprob && p_len && sent_cur && sent_next && c_score &&
                 c_logprob && c_len && c_slot && c_free && c_sent && parent && tok, "null pointer");
    BeamArgs a;
    a.B = B; a.n = beam; a.k = beam; a.Lmax = Lmax; a.eos = 0; a.len_norm_f = 0; a.tv = nullptr; a.ti = nullptr;
    a.pcount = pcount; a.ccount = ccount; a.p_len = p_len; a.c_len = c_len; a.c_slot = c_slot;
    a.c_free = c_free; a.p_score = p_score; a.p_logprob = p_logprob; a.c_score = c_score; a.c_logprob = c_logprob;
    a.sent_cur = sent_cur; a.sent_next = sent_next; a.c_sent = c_sent; a.parent = parent; a.tok = tok;
    const long work = (long)B * beam * (H > Lmax ? H : Lmax);
    hipLaunchKernelGGL(beam_init_kernel, dim3((unsigned)(cdiv(work, 256L) < 1024 ? cdiv(work, 256L) : 1024)), dim3(256), 0, (hipStream_t)stream,
                       a, bos, H, c_in, h_in, c_out, h_out, sent_cur);
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
    if (beam * beam <= 64) hipLaunchKernelGGL(beam_update_slim_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(beam_update_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, a);
    VC_LAUNCH_CHECK();
    return 0;
}
