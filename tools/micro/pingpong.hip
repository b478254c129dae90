// Ping-pong K loop: one workgroup of eight waves = two groups of four; while one group multiplies
// (its waves own the matrix pipes of the four SIMDs), the other does everything else (wait for its
// loads, transform, LDS stores, next loads), then they swap.  Two workgroup barriers per K step.
// Prices the idea against core_steps.hip's mode 3 on the same L2-resident operands.
//
// Measured on MI355X: 137 / 130 / 119 TFLOP/s for 2x(128x128) / 2x(128x64) / 2x(128x32) here (the two
// groups share the B tile), against 130 / 115 / 105-110 for the plain loop in core_steps.hip.  Built
// into the real kernels (each group on its own tile and LDS region) it LOST: GEMM 83-98 instead of
// 90-110 TFLOP/s, weight gradient 58-72 instead of 91-100.  With only one wave per SIMD multiplying,
// every wait on an LDS fragment read is an idle matrix pipe; two co-resident waves that multiply at
// the same time fill each other's bubbles, which is worth more than the strict alternation.  Kept as
// a record of the experiment; the product kernels use one group per workgroup, two workgroups per CU.
#include "../../disentangled-vae_amd/csrc/gemm_core.h"
#include <stdio.h>

__device__ __forceinline__ int ltid() { return threadIdx.x & 255; }

// KC loader for one group's share: V4 chunks per thread, rows row0.. of a [rows][K] matrix
template <int V4>
struct GLoader {
  const float* rowp[V4];
  __device__ __forceinline__ GLoader(const float* base, int64_t ld, int row0) {
#pragma unroll
    for (int v = 0; v < V4; ++v) {
      const int L = v * 256 + ltid();
      rowp[v] = base + (int64_t)(row0 + (L >> 3)) * ld + 4 * (L & 7);
    }
  }
  __device__ __forceinline__ void issue(int kt, f32x4 (&r)[V4]) {
#pragma unroll
    for (int v = 0; v < V4; ++v) r[v] = ldg4(rowp[v] + kt * 32);
  }
};
template <int V4, int LD>
__device__ __forceinline__ void gstore(float* lds, const f32x4 (&r)[V4]) {
#pragma unroll
  for (int v = 0; v < V4; ++v) {
    const int L = v * 256 + ltid();
    *reinterpret_cast<f32x4*>(lds + (L >> 3) * LD + 4 * (L & 7)) = r[v];
  }
}

template <class T>
__global__ void __launch_bounds__(512, 1) pp_kernel(const float* A, const float* B, float* out, int iters, int K) {
  constexpr int STAGE = 2 * T::A_FLOATS + T::B_FLOATS;
  constexpr int BV = T::B_V4 >= 2 ? T::B_V4 / 2 : 1;
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];
  const int grp = threadIdx.x >> 8;
  const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3;
  const int wm = wave / T::WGN, wn = wave % T::WGN;
  for (int i = threadIdx.x; i < 2 * STAGE; i += 512) smem[i] = (float)(i % 7) * 0.125f;
  __syncthreads();
  f32x16 acc[T::TM][T::TN];
  zero_acc<T>(acc);
  GLoader<T::A_V4> la(A, K, 0);
  const bool loads_b = T::B_V4 >= 2 || grp == 0;
  GLoader<BV> lb(B, K, T::B_V4 >= 2 ? grp * (T::BN / 2) : 0);
  f32x4 ra[T::A_V4], rb[BV];
  const int ktiles = K / 32;
  int kt = 0;
  la.issue(kt, ra);
  if (loads_b) lb.issue(kt, rb);
  if (++kt == ktiles) kt = 0;
  int cur = 0;
  float* aoff = smem + grp * T::A_FLOATS;
  float* boff = smem + 2 * T::A_FLOATS + (T::B_V4 >= 2 ? grp * (T::BN / 2) * T::BLD : 0);
  for (int it = 0; it < iters; ++it) {
    const float* st = smem + cur * STAGE;
    float* nx = smem + (cur ^ 1) * STAGE;
    if (grp == 0) {
      mma_stage<T>(st, st + 2 * T::A_FLOATS, acc, wm, wn, lane);
      __syncthreads();
      gstore<T::A_V4, T::ALD>(aoff + (cur ^ 1) * STAGE, ra);
      if (loads_b) gstore<BV, T::BLD>(boff + (cur ^ 1) * STAGE, rb);
      la.issue(kt, ra);
      if (loads_b) lb.issue(kt, rb);
      __syncthreads();
    } else {
      gstore<T::A_V4, T::ALD>(aoff + (cur ^ 1) * STAGE, ra);
      if (loads_b) gstore<BV, T::BLD>(boff + (cur ^ 1) * STAGE, rb);
      la.issue(kt, ra);
      if (loads_b) lb.issue(kt, rb);
      __syncthreads();
      mma_stage<T>(st + T::A_FLOATS, st + 2 * T::A_FLOATS, acc, wm, wn, lane);
      __syncthreads();
    }
    (void)nx;
    if (++kt == ktiles) kt = 0;
    cur ^= 1;
  }
  float s = 0.f;
  for_each_acc<T>(acc, [&](int, int, float v) { s += v; });
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <class T>
static void run(const char* name, const float* A, const float* B, float* out, int blocks, int iters, int K) {
  hipEvent_t s, e;
  hipEventCreate(&s);
  hipEventCreate(&e);
  hipLaunchKernelGGL((pp_kernel<T>), dim3(blocks), dim3(512), 0, 0, A, B, out, iters, K);
  hipDeviceSynchronize();
  hipEventRecord(s);
  hipLaunchKernelGGL((pp_kernel<T>), dim3(blocks), dim3(512), 0, 0, A, B, out, iters, K);
  hipEventRecord(e);
  hipEventSynchronize(e);
  float ms;
  hipEventElapsedTime(&ms, s, e);
  const double flops = (double)blocks * iters * 2.0 * (2 * T::BM) * T::BN * 32;
  printf("%-16s ping-pong blocks %4d: %8.3f ms  %6.1f TFLOP/s\n", name, blocks, ms, flops / ms / 1e9);
}

int main() {
  float *A, *B, *out;
  hipMalloc(&A, 256 * 1024 * 4);
  hipMalloc(&B, 128 * 1024 * 4);
  hipMalloc(&out, 4096 * 512 * 4);
  hipMemset(A, 0, 256 * 1024 * 4);
  hipMemset(B, 0, 128 * 1024 * 4);
  run<Tile<128, 128, 2, 2, true, true>>("2x(128x128)", A, B, out, 256, 4000, 1024);
  run<Tile<128, 64, 4, 1, true, true>>("2x(128x64)", A, B, out, 256, 4000, 1024);
  run<Tile<128, 32, 4, 1, true, true>>("2x(128x32)", A, B, out, 256, 4000, 1024);
  return 0;
}
