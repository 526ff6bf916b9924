#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t mul24(uint32_t a, uint32_t b) { return (a & 0xffffffu) * (b & 0xffffffu); }  // v_mul_u32_u24
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
  } else if (MODE == 3 || MODE == 4 || MODE == 5) {  // integer multiplies (the counter RNG) and a transcendental
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const uint32_t m = 0x9E3779B1u + (uint32_t)iters;
    float f0 = t + 1, f1 = t + 2, f2 = t + 3, f3 = t + 4, f4 = t + 5, f5 = t + 6, f6 = t + 7, f7 = t + 8;
    for (int i = 0; i < iters; ++i) {
      if (MODE == 3) {
        a0 *= m; a1 *= m; a2 *= m; a3 *= m; a4 *= m; a5 *= m; a6 *= m; a7 *= m;  // v_mul_lo_u32
      } else if (MODE == 4) {
        a0 = mul24(a0, m); a1 = mul24(a1, m); a2 = mul24(a2, m);
        a3 = mul24(a3, m); a4 = mul24(a4, m); a5 = mul24(a5, m);
        a6 = mul24(a6, m); a7 = mul24(a7, m);  // v_mul_u32_u24
      } else {
        f0 = __builtin_amdgcn_rcpf(f0); f1 = __builtin_amdgcn_rcpf(f1); f2 = __builtin_amdgcn_rcpf(f2); f3 = __builtin_amdgcn_rcpf(f3);
        f4 = __builtin_amdgcn_rcpf(f4); f5 = __builtin_amdgcn_rcpf(f5); f6 = __builtin_amdgcn_rcpf(f6); f7 = __builtin_amdgcn_rcpf(f7);
      }
    }
    out[blockIdx.x * 256 + threadIdx.x] = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7) + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
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
  run<3>("v_mul_lo_u32 x8 (flop column: instructions)", 8);
  run<4>("v_mul_u32_u24 x8 (flop column: instructions)", 8);
  run<5>("v_rcp_f32 x8 (flop column: instructions)", 8);
  return 0;
}
