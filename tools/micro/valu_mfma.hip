// Do VALU instructions and MFMAs of co-resident waves overlap?  Per K step each wave issues its
// tile's MFMAs (as in the GEMM core, fragments from LDS) plus NV independent integer VALU ops.
//
// Measured on MI355X: 128x128 tile 153.5 TFLOP/s with 0 extra ops, 150.7 with 512; 128x32 tile 150.4
// with 0, 140.2 with 256.  They overlap: vector ALU work of the loaders is not what holds the conv
// kernels at 0.5-0.75 of the MFMA peak.
#include "../../disentangled-vae_amd/csrc/gemm_core.h"
#include <stdio.h>

template <class T, int NV>
__global__ void __launch_bounds__(256, 2) k(float* out, int iters, int seed) {
  __shared__ __attribute__((aligned(16))) float smem[2 * T::STAGE_FLOATS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / T::WGN, wn = wave % T::WGN;
  for (int i = threadIdx.x; i < 2 * T::STAGE_FLOATS; i += 256) smem[i] = (float)(i % 7) * 0.125f;
  __syncthreads();
  f32x16 acc[T::TM][T::TN];
  zero_acc<T>(acc);
  unsigned x[8];
  for (int j = 0; j < 8; ++j) x[j] = seed + j * 77 + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    mma_stage<T>(smem, smem + T::A_FLOATS, acc, wm, wn, lane);
#pragma unroll
    for (int v = 0; v < NV / 8; ++v)
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = x[j] * 1664525u + 1013904223u;
    __syncthreads();
  }
  float s = 0.f;
  for_each_acc<T>(acc, [&](int, int, float v) { s += v; });
  unsigned xs = 0;
  for (int j = 0; j < 8; ++j) xs ^= x[j];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)xs;
}

template <class T, int NV>
static void run(const char* name, float* out, int blocks) {
  const int iters = 4000;
  hipEvent_t s, e;
  hipEventCreate(&s);
  hipEventCreate(&e);
  hipLaunchKernelGGL((k<T, NV>), dim3(blocks), dim3(256), 0, 0, out, iters, 1);
  hipDeviceSynchronize();
  hipEventRecord(s);
  hipLaunchKernelGGL((k<T, NV>), dim3(blocks), dim3(256), 0, 0, out, iters, 1);
  hipEventRecord(e);
  hipEventSynchronize(e);
  float ms;
  hipEventElapsedTime(&ms, s, e);
  const double flops = (double)blocks * iters * 2.0 * T::BM * T::BN * 32;
  printf("%-10s +%4d VALU ops per K step: %8.3f ms  %6.1f TFLOP/s\n", name, NV, ms, flops / ms / 1e9);
}

int main() {
  float* out;
  hipMalloc(&out, 4096 * 256 * 4);
  typedef Tile<128, 128, 2, 2, true, true> T128;
  typedef Tile<128, 32, 4, 1, true, true> T32;
  run<T128, 0>("128x128", out, 512);
  run<T128, 128>("128x128", out, 512);
  run<T128, 256>("128x128", out, 512);
  run<T128, 512>("128x128", out, 512);
  run<T32, 0>("128x32", out, 768);
  run<T32, 64>("128x32", out, 768);
  run<T32, 128>("128x32", out, 768);
  run<T32, 256>("128x32", out, 768);
  return 0;
}
