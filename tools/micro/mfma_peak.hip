// Sustained v_mfma_f32_32x32x2_f32 rate with no memory traffic: the practical ceiling for the f32
// GEMM kernels (power / clock limited), to read roofline fractions against.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ void __launch_bounds__(256) mfma_loop(float* out, int iters, float a0, float b0) {
  f32x16 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c)
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  float a = a0 + threadIdx.x, b = b0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < CHAINS; ++c)
    for (int r = 0; r < 16; ++r) s += acc[c][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int CHAINS>
static void run(int wg_per_cu, int iters) {
  float* out;
  const int blocks = 256 * wg_per_cu;
  hipMalloc(&out, blocks * 256 * sizeof(float));
  hipEvent_t s, e;
  hipEventCreate(&s);
  hipEventCreate(&e);
  hipLaunchKernelGGL(mfma_loop<CHAINS>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f);
  hipDeviceSynchronize();
  hipEventRecord(s);
  hipLaunchKernelGGL(mfma_loop<CHAINS>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f);
  hipEventRecord(e);
  hipEventSynchronize(e);
  float ms;
  hipEventElapsedTime(&ms, s, e);
  const double flops = (double)blocks * 4 * iters * 4 * CHAINS * 4096.0;
  printf("chains %d, %d WG/CU, %d iters: %.3f ms, %.1f TFLOP/s\n", CHAINS, wg_per_cu, iters, ms, flops / ms / 1e9);
  hipFree(out);
}

int main() {
  run<4>(1, 20000);
  run<4>(2, 20000);
  run<1>(2, 40000);
  run<1>(4, 40000);
  run<4>(2, 200000);
  return 0;
}
