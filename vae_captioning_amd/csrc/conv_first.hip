// conv1_1 of the VGG16 (utils/image_embeddings.py:36-48: 3x3, stride 1, SAME, 3 -> 64 channels, bias, ReLU) -- forward and weight
// gradient for gfx950.  (It has no data gradient: the images are not trained.)
//
// With K = 27 the layer is 0.6 % of the network's multiply-adds but writes (forward) / reads (weight gradient) the largest
// activation of the net, [B, 224, 224, 64] fp32 = 822 MB at 64 images: both kernels are HBM-bound (0.87 GB -> ~0.14 ms at
// 6.3 TB/s) and are built around 16-byte / full-line accesses of that tensor; the 32-wide K-tiles of the general kernels
// (conv.hip) spent 0.52 + 0.35 ms on them.
//
//   forward:  D[channel][pixel] = sum_k W^T[channel][k] . patch[k][pixel], k = 3 tap + c (27, padded to 28 = 14 MFMA k-steps).
//             The OUTPUT CHANNELS are the M dimension: a lane of v_mfma_f32_32x32x2_f32 then owns one pixel and, in each group of four
//             accumulator registers, four CONSECUTIVE channels -> 16-byte LDS writes into a wave-private [32 pixels][64 channels]
//             tile, read back so that each global store instruction covers 1 KB of consecutive memory (the bias rides in the
//             padding row k = 27 of the contraction; ReLU in registers; no workgroup barrier).  A wave owns 32 consecutive pixels of an image row; the 3 x 3 x RGBX neighbourhood comes as five
//             16-byte loads per lane (lane-half h takes k in [14 h, 14 h + 14): taps 4 h .. 4 h + 4), zero outside the image through
//             the buffer out-of-range bit; the weights (28 x 64) stay in registers for the whole launch.
//   weight gradient:  D[m][n] = sum_pixels patch[pixel][m] . dy[pixel][n], m = 3 tap + c, plus row m = 27 of ones = the bias
//             gradient; one 32 x 64 accumulator pair per wave over its share of the pixels, summed over the waves of a workgroup in
//             LDS and over the workgroups by a second kernel in a fixed order (deterministic).
#include "gemm_core.h"
#include "vaecap.h"

namespace vc {

typedef unsigned int c1_u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned C1_OOB = 0x80000000u;

__device__ __forceinline__ float4 c1_load4(__amdgpu_buffer_rsrc_t r, unsigned voff) {
    c1_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0);
    return *reinterpret_cast<float4*>(&v);
}
__device__ __forceinline__ float c1_load1(__amdgpu_buffer_rsrc_t r, unsigned voff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, 0, 0));
}
__device__ __forceinline__ float c1_comp(const float4& v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; }

struct Conv1Args {
    const float* x4;    // [B, H, W, 4] mean-subtracted RGB + one zero channel (vc_vgg_preprocess_f32)
    const float* w;     // [3, 3, 3, 64] HWIO
    const float* bias;  // [64]
    float* y;           // [B][16][H][W][4] (the C4 activation layout, vaecap.h)
    const float* dy;    // weight gradient: layout of y
    float* ws;          // weight gradient: [workgroups][28][64] partial sums
    int B, H, W, relu;
    int groups;         // B * H * W / 32
    int segs;           // W / 32
    unsigned* mask;     // forward, optional: (y > 0) as bits in the lane order of conv_wino4.hip's data gradient with N = 64 produced channels
                        // ([workgroup id = 2 (block / 2) + column tile][256 threads][2 words], bit 16 aa + 4 bb + c of a lane's 4 x 4 pixels x 4 channels)
    int blocks_img, bx_n;   // 16 x 16-pixel blocks per image / per image row (H, W multiples of 16)
};

__global__ __launch_bounds__(256, 3) void conv1_fwd_kernel(Conv1Args a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int H = a.H, W = a.W;
    // A operand: W^T[channel 32 tm + li][k = 14 lh + s]; the padding row k = 27 carries the BIAS (its B operand is 1.0)
    float wa[2][14];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int s = 0; s < 14; ++s) {
            const int k = 14 * lh + s;
            wa[tm][s] = k < 27 ? a.w[k * 64 + tm * 32 + li] : a.bias[tm * 32 + li];
        }
    // slot j of a lane holds tap j + 4 lh: its row / column displacement
    int dyl[5], dxl[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int t0 = j, t1 = j + 4;
        dyl[j] = lh ? t1 / 3 - 1 : t0 / 3 - 1;
        dxl[j] = lh ? t1 % 3 - 1 : t0 % 3 - 1;
    }
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x4, 0, a.B * H * W * 16, 0x00020000);
    const int nw = gridDim.x * 4;
    float4 slot[2][5];
    auto fetch = [&](int g, float4 (&sl)[5]) {
        const int row = g / a.segs, x = (g - row * a.segs) * 32 + li, y = row % H;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const bool ok = ((unsigned)(y + dyl[j]) < (unsigned)H) & ((unsigned)(x + dxl[j]) < (unsigned)W);
            const unsigned off = (unsigned)((row + dyl[j]) * W + x + dxl[j]) * 16u;
            sl[j] = c1_load4(rx, ok ? off : C1_OOB);
        }
    };
    int g = blockIdx.x * 4 + wave;
    if (g < a.groups) fetch(g, slot[0]);
    int cur = 0;
    for (; g < a.groups; g += nw) {
        const int gn = g + nw;
        if (cur == 0) {
            if (gn < a.groups) fetch(gn, slot[1]);
        } else {
            if (gn < a.groups) fetch(gn, slot[0]);
        }
        f32x16 acc[2];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][r] = 0.f;
        auto mac = [&](const float4 (&sl)[5]) {
#pragma unroll
            for (int s = 0; s < 14; ++s) {
                // lane-half 0: k = s -> tap s / 3 (slot s / 3), channel s % 3; half 1: k = 14 + s -> tap (14 + s) / 3 = slot (s + 2) / 3
                const float v0 = c1_comp(sl[s / 3], s % 3);
                const float v1 = s == 13 ? 1.f : c1_comp(sl[s == 13 ? 4 : (s + 2) / 3], (s + 2) % 3);  // (k = 27: the bias row)
                const float bv = lh ? v1 : v0;
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) acc[tm] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[tm][s], bv, acc[tm], 0, 0, 0);
            }
        };
        if (cur == 0) mac(slot[0]); else mac(slot[1]);
        cur ^= 1;
        // lane (li, lh) holds pixel li, channels 32 tm + 8 q + 4 lh .. + 3 = channel quad 8 tm + 2 q + lh: in the C4 layout
        // ([B][16][H][W][4]) the 32 lanes of a half write 32 consecutive pixels of ONE plane -- every store instruction is two runs of
        // 512 consecutive bytes straight from the accumulators (the NHWC form needed a transpose through the LDS for full lines)
        const int row = g / a.segs, xs = (g - row * a.segs) * 32 + li, bi = row / H;
        float* yp = a.y + (((long)(row + bi * 15 * H + lh * H)) * W + xs) * 4;
        const long qstep = (long)H * W * 4;   // floats per channel-quad plane
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 v = make_float4(acc[tm][4 * q], acc[tm][4 * q + 1], acc[tm][4 * q + 2], acc[tm][4 * q + 3]);
                if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                *reinterpret_cast<float4*>(yp + (8 * tm + 2 * q) * qstep) = v;
                if (a.mask) {
                    // the ReLU mask of conv1_2's data gradient as bits (that kernel then loads 8 bytes per lane instead of sixteen float4):
                    // this lane's nibble = (v > 0) of its pixel and channel quad 8 tm + 2 q + lh; the four pixels of a tile row are four
                    // neighbouring lanes -> one half-word per (tile, quad, tile row aa), written by the lane of the row's first pixel
                    unsigned nb4 = (v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u);
                    nb4 |= (unsigned)__shfl_down((int)nb4, 1, 64) << 4;
                    nb4 |= (unsigned)__shfl_down((int)nb4, 2, 64) << 8;
                    if ((li & 3) == 0) {
                        const int yy = row - bi * H;
                        const int gb = bi * a.blocks_img + (yy >> 4) * a.bx_n + (xs >> 4);
                        const int thr = ((gb & 1) * 2 + (q >> 1)) * 64 + (2 * (q & 1) + lh) * 16 + ((yy >> 2) & 3) * 4 + ((xs >> 2) & 3);
                        const long word = ((long)((gb >> 1) * 2 + tm) * 256 + thr) * 2 + ((yy >> 1) & 1);
                        reinterpret_cast<unsigned short*>(a.mask)[word * 2 + (yy & 1)] = (unsigned short)(nb4 & 0xffffu);
                    }
                }
            }
    }
}

// weight gradient, partial sums per workgroup: ws[wg][m][n], m = 3 tap + c (27 rows) and m = 27 = sum of dy (bias gradient)
__global__ __launch_bounds__(256) void conv1_wgrad_kernel(Conv1Args a) {
    __shared__ __attribute__((aligned(16))) float red[4][32 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int H = a.H, W = a.W;
    // A row m = li: tap m / 3, channel m % 3; m == 27: ones; m > 27: zeros
    const int tap = li / 3, ch = li - 3 * tap;
    const int dym = tap / 3 - 1, dxm = tap % 3 - 1;
    const bool patch_row = li < 27;
    const float constant = li == 27 ? 1.f : 0.f;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x4, 0, a.B * H * W * 16, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.B * H * W * 256, 0x00020000);
    f32x16 acc[2];
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tn][r] = 0.f;
    const int nw = gridDim.x * 4;
    for (int g = blockIdx.x * 4 + wave; g < a.groups; g += nw) {
        const int row = g / a.segs, x0 = (g - row * a.segs) * 32, y = row % H;
        const bool oky = patch_row & ((unsigned)(y + dym) < (unsigned)H);
        // lane-half lh takes pixels x0 + 16 lh + s
        const int xb = x0 + 16 * lh + dxm;
        const unsigned abase = (unsigned)((row + dym) * W + xb) * 16u + (unsigned)ch * 4u;
        // dy in the C4 layout: channel 32 tn + li = component li & 3 of channel quad 8 tn + li / 4 (a plane per quad, 16 bytes per pixel)
        const int bi = row / H;
        const unsigned bbase = (unsigned)(((row + bi * 15 * H + (li >> 2) * H) * W + x0 + 16 * lh) * 16 + (li & 3) * 4);
        const unsigned tnstep = (unsigned)(8 * H * W * 16);
        float av[16], bv[2][16];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const bool ok = oky & ((unsigned)(xb + s) < (unsigned)W);
            av[s] = c1_load1(rx, ok ? abase + (unsigned)s * 16u : C1_OOB);
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) bv[tn][s] = c1_load1(rd, bbase + (unsigned)s * 16u + (unsigned)tn * tnstep);
        }
        __builtin_amdgcn_sched_barrier(0);  // every load of the group is in flight before its first MFMA
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float am = patch_row ? av[s] : constant;
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) acc[tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(am, bv[tn][s], acc[tn], 0, 0, 0);
        }
    }
    // D: lane holds column n = 32 tn + li of rows m = (r&3) + 8 (r>>2) + 4 lh
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave][((r & 3) + 8 * (r >> 2) + 4 * lh) * 64 + tn * 32 + li] = acc[tn][r];
    __syncthreads();
    float* o = a.ws + (long)blockIdx.x * (28 * 64);
    for (int i = tid; i < 28 * 64; i += 256) o[i] = ((red[0][i] + red[1][i]) + red[2][i]) + red[3][i];  // fixed order
}

// ws[parts][1792] -> dw / db: a block owns 32 outputs, its eight waves-halves sum every eighth partial and meet in LDS; all in a
// fixed order (deterministic)
__global__ __launch_bounds__(256) void conv1_wgrad_reduce_kernel(const float* __restrict__ ws, int parts, float* __restrict__ dw,
                                                                 float* __restrict__ db, int accumulate) {
    __shared__ float part[8][32];
    const int o = threadIdx.x & 31, pg = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + o;  // < 1792 = 56 blocks x 32
    float s0 = 0.f, s1 = 0.f;
    int p = pg;
    for (; p + 8 < parts; p += 16) {
        s0 += ws[(long)p * 1792 + i];
        s1 += ws[(long)(p + 8) * 1792 + i];
    }
    if (p < parts) s0 += ws[(long)p * 1792 + i];
    part[pg][o] = s0 + s1;
    __syncthreads();
    if (pg) return;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += part[k][o];
    if (i < 27 * 64) dw[i] = accumulate ? dw[i] + s : s;
    else if (db) db[i - 27 * 64] = accumulate ? db[i - 27 * 64] + s : s;
}

constexpr int C1_WGRAD_WGS = 1024;

// a LAUNCH addresses [B,H,W,64] floats with 32-bit offsets (< 2 GiB); a call on more images is cut into launches over image ranges
static int conv1_images_per_launch(int B, int H, int W) {
    const long per = (long)H * W * 256;
    long n = per > 0 ? 0x7fffffffL / per : 0;
    return (int)(n > B ? B : n);
}
static bool conv1_ok(int B, int H, int W) { return B > 0 && H > 0 && W > 0 && W % 32 == 0 && conv1_images_per_launch(B, H, W) > 0; }

}  // namespace vc

// 1 when the specialised conv1_1 kernels handle this geometry (W % 32 == 0, one image's activation < 2 GiB), else callers use vc_conv3x3_*_f32
extern "C" int vc_conv1_supported(int B, int H, int W) { return vc::conv1_ok(B, H, W) ? 1 : 0; }

extern "C" size_t vc_conv1_wgrad_workspace_bytes(void) { return (size_t)vc::C1_WGRAD_WGS * 28 * 64 * sizeof(float); }

// y = relu?(conv3x3(x4[..., :3], w) + bias): x4 [B,H,W,4] (fourth channel ignored), w [3,3,3,64] HWIO, y [B,H,W,64]
extern "C" int vc_conv1_fwd_f32(void* stream, int B, int H, int W, const float* x4, const float* w, const float* bias, float* y, int relu) {
    using namespace vc;
    VC_CHECK_ARG(conv1_ok(B, H, W), "unsupported geometry (W % 32 == 0 and activation < 2 GiB required; see vc_conv1_supported)");
    VC_CHECK_ARG(x4 && w && bias && y, "null pointer");
    VC_CHECK_ARG((((uintptr_t)x4 | (uintptr_t)y | (uintptr_t)bias) & 15) == 0, "x4 / y / bias must be 16-byte aligned");
    const int per = conv1_images_per_launch(B, H, W);
    for (int b0 = 0; b0 < B; b0 += per) {
        const int nb = B - b0 < per ? B - b0 : per;
        Conv1Args a{};
        a.x4 = x4 + (size_t)b0 * H * W * 4; a.w = w; a.bias = bias; a.y = y + (size_t)b0 * H * W * 64; a.B = nb; a.H = H; a.W = W; a.relu = relu;
        a.segs = W / 32; a.groups = nb * H * a.segs;
        int wgs = cdiv(a.groups, 4);
        if (wgs > 2048) wgs = 2048;
        hipLaunchKernelGGL(conv1_fwd_kernel, dim3(wgs), dim3(256), 0, (hipStream_t)stream, a);
        const int rc = launch_status(__func__);
        if (rc) return rc;
    }
    return 0;
}

// The same forward, also leaving (y > 0) as bits for the F(4x4,3x3) data gradient of the NEXT layer (conv1_2: Cin = 64):
// mask_out = vc_conv3x3_wino4_mask_words(B, H, W, 64) words in the layout vc_conv3x3_wino4_fwd_mask_f32 writes, to be passed to
// vc_conv3x3_wino4_dgrad_bits_f32(B, H, W, 64, Cout, ...) -- bit-identical to its float-mask form on y.  One launch only (the bits
// are per tile of ONE launch of B images), H and W multiples of 16 (whole blocks: every half-word is written).
extern "C" int vc_conv1_fwd_mask_f32(void* stream, int B, int H, int W, const float* x4, const float* w, const float* bias, float* y,
                                     uint32_t* mask_out) {
    using namespace vc;
    VC_CHECK_ARG(conv1_ok(B, H, W) && conv1_images_per_launch(B, H, W) >= B && H % 16 == 0 && W % 16 == 0,
                 "unsupported geometry (one launch, H % 16 == 0, W % 32 == 0 required)");
    VC_CHECK_ARG(x4 && w && bias && y && mask_out, "null pointer");
    VC_CHECK_ARG((((uintptr_t)x4 | (uintptr_t)y | (uintptr_t)bias | (uintptr_t)mask_out) & 15) == 0, "x4 / y / bias / mask_out must be 16-byte aligned");
    Conv1Args a{};
    a.x4 = x4; a.w = w; a.bias = bias; a.y = y; a.B = B; a.H = H; a.W = W; a.relu = 1;
    a.segs = W / 32; a.groups = B * H * a.segs;
    a.mask = mask_out; a.bx_n = W / 16; a.blocks_img = (H / 16) * a.bx_n;
    int wgs = cdiv(a.groups, 4);
    if (wgs > 2048) wgs = 2048;
    hipLaunchKernelGGL(conv1_fwd_kernel, dim3(wgs), dim3(256), 0, (hipStream_t)stream, a);
    return launch_status(__func__);
}

// dw [3,3,3,64] (+)= sum over pixels of patch x dy; db [64] (+)= sum of dy (may be null); dy is the gradient w.r.t. the layer's
// pre-activation (the ReLU gradient is applied by the producer, as for vc_conv3x3_wgrad_f32)
extern "C" int vc_conv1_wgrad_f32(void* stream, int B, int H, int W, const float* x4, const float* dy, float* dw, float* db, int accumulate,
                                  float* ws, size_t ws_bytes) {
    using namespace vc;
    VC_CHECK_ARG(conv1_ok(B, H, W), "unsupported geometry (W % 32 == 0 and activation < 2 GiB required; see vc_conv1_supported)");
    VC_CHECK_ARG(x4 && dy && dw, "null pointer");
    if (!ws || ws_bytes < vc_conv1_wgrad_workspace_bytes()) return fail(VC_EWORKSPACE, "%s: workspace too small (vc_conv1_wgrad_workspace_bytes)", __func__);
    const int per = conv1_images_per_launch(B, H, W);
    for (int b0 = 0; b0 < B; b0 += per) {   // the later image ranges accumulate into dw / db
        const int nb = B - b0 < per ? B - b0 : per;
        Conv1Args a{};
        a.x4 = x4 + (size_t)b0 * H * W * 4; a.dy = dy + (size_t)b0 * H * W * 64; a.ws = ws; a.B = nb; a.H = H; a.W = W;
        a.segs = W / 32; a.groups = nb * H * a.segs;
        int wgs = cdiv(a.groups, 4);
        if (wgs > C1_WGRAD_WGS) wgs = C1_WGRAD_WGS;
        hipLaunchKernelGGL(conv1_wgrad_kernel, dim3(wgs), dim3(256), 0, (hipStream_t)stream, a);
        int rc = launch_status(__func__);
        if (rc) return rc;
        hipLaunchKernelGGL(conv1_wgrad_reduce_kernel, dim3(56), dim3(256), 0, (hipStream_t)stream, ws, wgs, dw, db, accumulate || b0 > 0);
        rc = launch_status(__func__);
        if (rc) return rc;
    }
    return 0;
}
