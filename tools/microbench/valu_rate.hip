#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float s) {
  float t = threadIdx.x * 1e-3f;
  if (MODE == 0) {
    float a0 = t, a1 = t + 1, a2 = t + 2, a3 = t + 3, a4 = t + 4, a5 = t + 5, a6 = t + 6, a7 = t + 7;
    for (int i = 0; i < iters; ++i) {
      a0 = fmaf(a0, s, 1.0f); a1 = fmaf(a1, s, 1.0f); a2 = fmaf(a2, s, 1.0f); a3 = fmaf(a3, s, 1.0f);
      a4 = fmaf(a4, s, 1.0f); a5 = fmaf(a5, s, 1.0f); a6 = fmaf(a6, s, 1.0f); a7 = fmaf(a7, s, 1.0f);
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  } else if (MODE == 1) {
    v2f a0 = {t, t + 1}, a1 = {t + 2, t + 3}, a2 = {t + 4, t + 5}, a3 = {t + 6, t + 7}, a4 = {t + 8, t + 9}, a5 = {t + 10, t + 11},
        a6 = {t + 12, t + 13}, a7 = {t + 14, t + 15};
    v2f ss = {s, s}, one = {1.0f, 1.0f};
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_elementwise_fma(a0, ss, one); a1 = __builtin_elementwise_fma(a1, ss, one);
      a2 = __builtin_elementwise_fma(a2, ss, one); a3 = __builtin_elementwise_fma(a3, ss, one);
      a4 = __builtin_elementwise_fma(a4, ss, one); a5 = __builtin_elementwise_fma(a5, ss, one);
      a6 = __builtin_elementwise_fma(a6, ss, one); a7 = __builtin_elementwise_fma(a7, ss, one);
    }
    v2f r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[blockIdx.x * 256 + threadIdx.x] = r.x + r.y;
  } else {  // min/max chain
    float a0 = t, a1 = t + 1, a2 = t + 2, a3 = t + 3, a4 = t + 4, a5 = t + 5, a6 = t + 6, a7 = t + 7;
    for (int i = 0; i < iters; ++i) {
      a0 = fmaxf(a0 * s, 1.0f); a1 = fminf(a1 + s, 100.0f); a2 = fmaxf(a2 * s, 1.0f); a3 = fminf(a3 + s, 100.0f);
      a4 = fmaxf(a4 * s, 1.0f); a5 = fminf(a5 + s, 100.0f); a6 = fmaxf(a6 * s, 1.0f); a7 = fminf(a7 + s, 100.0f);
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  }
}
template <int MODE>
void run(const char* name, double flop_per_iter_lane) {
  float* d;
  const int blocks = 256 * 16, iters = 20000;
  hipMalloc(&d, blocks * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int r = 0; r < 3; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 0.999f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (r == 2) printf("%s: %.2f ms, %.1f TFLOP/s, %.1f G wave-instr/s\n", name, ms, flop_per_iter_lane * iters * blocks * 256 / ms / 1e9,
                       8.0 * (MODE == 2 ? 2 : 1) * iters * blocks * 4 / ms / 1e6);
  }
  hipFree(d);
}
int main() {
  run<0>("v_fma_f32 x8", 16);
  run<1>("v_pk_fma_f32 x8", 32);
  run<2>("mul+max / add+min x8", 16);
  return 0;
}
