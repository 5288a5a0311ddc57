/* Benchmark-only entry points.  NOT part of libvaecap.so: they exist in libvaecap_microbench.so, built with -DVC_MICROBENCH by
 * `make -C vae_captioning_amd/csrc microbench` and used by tools/microbench.py only. */
#ifndef VAECAP_MICROBENCH_H
#define VAECAP_MICROBENCH_H
#ifdef __cplusplus
extern "C" {
#endif
/* The NN 128x128 GEMM kernel with main-loop stages ablated (bit 1: no global loads, 2: no LDS stores / barriers, 4: no LDS reads,
 * 8: double-buffered LDS variant); output is meaningless for variant != 0 and != 8. */
int vc_debug_gemm_ablate_f32(void* stream, int variant, int M, int N, int K, const float* A, const float* B, float* C);
#ifdef __cplusplus
}
#endif
#endif
