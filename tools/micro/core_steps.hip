// What the f32 GEMM core loses to each ingredient of its K step, measured without the rest:
//   mode 0: mma_stage only (LDS reads + MFMA), operands staged once
//   mode 1: + one __syncthreads per K step
//   mode 2: + LDS stores of a register tile per K step (no global loads)
//   mode 3: + global loads (plain KC loaders on a small matrix that stays in L2)
#include "../../disentangled-vae_amd/csrc/gemm_core.h"
#include <stdio.h>

template <class T, int MODE>
__global__ void __launch_bounds__(256, 2) core_kernel(const float* A, const float* B, float* out, int iters, int K) {
  __shared__ __attribute__((aligned(16))) float smem[2 * T::STAGE_FLOATS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / T::WGN, wn = wave % T::WGN;
  for (int i = threadIdx.x; i < 2 * T::STAGE_FLOATS; i += 256) smem[i] = (float)(i % 7) * 0.125f;
  __syncthreads();
  f32x16 acc[T::TM][T::TN];
  zero_acc<T>(acc);
  f32x4 ra[T::A_V4], rb[T::B_V4];
  for (int v = 0; v < T::A_V4; ++v) ra[v] = (f32x4){1.f, 2.f, 3.f, 4.f};
  for (int v = 0; v < T::B_V4; ++v) rb[v] = (f32x4){1.f, 2.f, 3.f, 4.f};
  KCLoader<T::A_V4> la(A, K, T::BM, K, 0);
  KCLoader<T::B_V4> lb(B, K, T::BN, K, 0);
  f32x4 xa[T::A_V4], xb[T::B_V4];
  float sink = 0.f;
  const int ktiles = K / 32;
  int cur = 0, kt = 0;
  for (int it = 0; it < iters; ++it) {
    if (MODE >= 3) {
      la.issue(kt, MODE == 6 || MODE == 7 ? xa : ra);
      lb.issue(kt, MODE == 6 || MODE == 7 ? xb : rb);
      if (++kt == ktiles) kt = 0;
    }
    const float* st = smem + cur * T::STAGE_FLOATS;
    mma_stage<T>(st, st + T::A_FLOATS, acc, wm, wn, lane);
    if (MODE == 6) {   // loads issued, consumed only by a cheap reduction after the multiply
      for (int v = 0; v < T::A_V4; ++v) sink += xa[v].x;
      for (int v = 0; v < T::B_V4; ++v) sink += xb[v].x;
    }
    if (MODE == 7) {   // loads issued and finished, but constants are what goes to LDS
      la.finish(xa);
      lb.finish(xb);
      for (int v = 0; v < T::A_V4; ++v) sink += xa[v].x + xa[v].w;
      for (int v = 0; v < T::B_V4; ++v) sink += xb[v].x + xb[v].w;
    }
    if (MODE >= 2) {
      float* nx = smem + (cur ^ 1) * T::STAGE_FLOATS;
      if (MODE >= 3 && MODE < 6) {
        la.finish(ra);
        lb.finish(rb);
      }
      store_a_tile<T>(nx, ra);
      store_b_tile<T>(nx + T::A_FLOATS, rb);
    }
    if (MODE >= 1) __syncthreads();
    if (MODE >= 2) cur ^= 1;
  }
  float s = 0.f;
  for_each_acc<T>(acc, [&](int, int, float v) { s += v; });
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// mode 4: loads issued two K steps ahead (two register sets); mode 5: same, with the address math
// and the LDS stores placed between the four MFMA groups of the step
template <class T>
__device__ __forceinline__ void mma_q(const float* As, const float* Bs, f32x16 (&acc)[T::TM][T::TN], int wm, int wn,
                                      int lane, int q) {
  const int l31 = lane & 31, h = lane >> 5;
  float a[T::TM][4], b[T::TN][4];
#pragma unroll
  for (int i = 0; i < T::TM; ++i) {
    const int row = (wm * T::TM + i) * 32 + l31;
    const float4 v = *reinterpret_cast<const float4*>(As + row * T::ALD + 8 * q + 4 * h);
    a[i][0] = v.x; a[i][1] = v.y; a[i][2] = v.z; a[i][3] = v.w;
  }
#pragma unroll
  for (int j = 0; j < T::TN; ++j) {
    const int row = (wn * T::TN + j) * 32 + l31;
    const float4 v = *reinterpret_cast<const float4*>(Bs + row * T::BLD + 8 * q + 4 * h);
    b[j][0] = v.x; b[j][1] = v.y; b[j][2] = v.z; b[j][3] = v.w;
  }
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
      for (int j = 0; j < T::TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
}

template <class T, int MODE>
__global__ void __launch_bounds__(256, 2) core2_kernel(const float* A, const float* B, float* out, int iters, int K) {
  __shared__ __attribute__((aligned(16))) float smem[2 * T::STAGE_FLOATS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / T::WGN, wn = wave % T::WGN;
  for (int i = threadIdx.x; i < 2 * T::STAGE_FLOATS; i += 256) smem[i] = (float)(i % 7) * 0.125f;
  __syncthreads();
  f32x16 acc[T::TM][T::TN];
  zero_acc<T>(acc);
  f32x4 ra0[T::A_V4], rb0[T::B_V4], ra1[T::A_V4], rb1[T::B_V4];
  KCLoader<T::A_V4> la0(A, K, T::BM, K, 0), la1(A, K, T::BM, K, 0);
  KCLoader<T::B_V4> lb0(B, K, T::BN, K, 0), lb1(B, K, T::BN, K, 0);
  const int ktiles = K / 32;
  la1.issue(0, ra1);
  lb1.issue(0, rb1);
  int cur = 0, kt = 1;
  for (int it = 0; it < iters; it += 2) {
    // even step: loads into set 0, store set 1
    {
      const float* st = smem + cur * T::STAGE_FLOATS;
      float* nx = smem + (cur ^ 1) * T::STAGE_FLOATS;
      if (MODE == 4) {
        la0.issue(kt, ra0);
        lb0.issue(kt, rb0);
        mma_stage<T>(st, st + T::A_FLOATS, acc, wm, wn, lane);
        la1.finish(ra1);
        lb1.finish(rb1);
        store_a_tile<T>(nx, ra1);
        store_b_tile<T>(nx + T::A_FLOATS, rb1);
      } else {
        mma_q<T>(st, st + T::A_FLOATS, acc, wm, wn, lane, 0);
        la0.issue(kt, ra0);
        lb0.issue(kt, rb0);
        mma_q<T>(st, st + T::A_FLOATS, acc, wm, wn, lane, 1);
        la1.finish(ra1);
        store_a_tile<T>(nx, ra1);
        mma_q<T>(st, st + T::A_FLOATS, acc, wm, wn, lane, 2);
        lb1.finish(rb1);
        store_b_tile<T>(nx + T::A_FLOATS, rb1);
        mma_q<T>(st, st + T::A_FLOATS, acc, wm, wn, lane, 3);
      }
      if (++kt == ktiles) kt = 0;
      __syncthreads();
      cur ^= 1;
    }
    {
      const float* st = smem + cur * T::STAGE_FLOATS;
      float* nx = smem + (cur ^ 1) * T::STAGE_FLOATS;
      if (MODE == 4) {
        la1.issue(kt, ra1);
        lb1.issue(kt, rb1);
        mma_stage<T>(st, st + T::A_FLOATS, acc, wm, wn, lane);
        la0.finish(ra0);
        lb0.finish(rb0);
        store_a_tile<T>(nx, ra0);
        store_b_tile<T>(nx + T::A_FLOATS, rb0);
      } else {
        mma_q<T>(st, st + T::A_FLOATS, acc, wm, wn, lane, 0);
        la1.issue(kt, ra1);
        lb1.issue(kt, rb1);
        mma_q<T>(st, st + T::A_FLOATS, acc, wm, wn, lane, 1);
        la0.finish(ra0);
        store_a_tile<T>(nx, ra0);
        mma_q<T>(st, st + T::A_FLOATS, acc, wm, wn, lane, 2);
        lb0.finish(rb0);
        store_b_tile<T>(nx + T::A_FLOATS, rb0);
        mma_q<T>(st, st + T::A_FLOATS, acc, wm, wn, lane, 3);
      }
      if (++kt == ktiles) kt = 0;
      __syncthreads();
      cur ^= 1;
    }
  }
  float s = 0.f;
  for_each_acc<T>(acc, [&](int, int, float v) { s += v; });
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <class T, int MODE>
static void run2(const char* name, const float* A, const float* B, float* out, int blocks, int iters, int K) {
  hipEvent_t s, e;
  hipEventCreate(&s);
  hipEventCreate(&e);
  hipLaunchKernelGGL((core2_kernel<T, MODE>), dim3(blocks), dim3(256), 0, 0, A, B, out, iters, K);
  hipDeviceSynchronize();
  hipEventRecord(s);
  hipLaunchKernelGGL((core2_kernel<T, MODE>), dim3(blocks), dim3(256), 0, 0, A, B, out, iters, K);
  hipEventRecord(e);
  hipEventSynchronize(e);
  float ms;
  hipEventElapsedTime(&ms, s, e);
  const double flops = (double)blocks * iters * 2.0 * T::BM * T::BN * 32;
  printf("%-14s mode %d blocks %4d: %8.3f ms  %6.1f TFLOP/s\n", name, MODE, blocks, ms, flops / ms / 1e9);
}

template <class T, int MODE>
static void run(const char* name, const float* A, const float* B, float* out, int blocks, int iters, int K) {
  hipEvent_t s, e;
  hipEventCreate(&s);
  hipEventCreate(&e);
  hipLaunchKernelGGL((core_kernel<T, MODE>), dim3(blocks), dim3(256), 0, 0, A, B, out, iters, K);
  hipDeviceSynchronize();
  hipEventRecord(s);
  hipLaunchKernelGGL((core_kernel<T, MODE>), dim3(blocks), dim3(256), 0, 0, A, B, out, iters, K);
  hipEventRecord(e);
  hipEventSynchronize(e);
  float ms;
  hipEventElapsedTime(&ms, s, e);
  const double flops = (double)blocks * iters * 2.0 * T::BM * T::BN * 32;
  printf("%-14s mode %d blocks %4d: %8.3f ms  %6.1f TFLOP/s\n", name, MODE, blocks, ms, flops / ms / 1e9);
}

template <class T>
static void all(const char* name, const float* A, const float* B, float* out, int per_cu) {
  const int iters = 4000, K = 1024;
  run<T, 0>(name, A, B, out, 256 * per_cu, iters, K);
  run<T, 1>(name, A, B, out, 256 * per_cu, iters, K);
  run<T, 2>(name, A, B, out, 256 * per_cu, iters, K);
  run<T, 3>(name, A, B, out, 256 * per_cu, iters, K);
  run<T, 6>(name, A, B, out, 256 * per_cu, iters, K);
  run<T, 7>(name, A, B, out, 256 * per_cu, iters, K);
  if (T::AKC && T::BKC) {
    run2<T, 4>(name, A, B, out, 256 * per_cu, iters, K);
    run2<T, 5>(name, A, B, out, 256 * per_cu, iters, K);
  }
}

int main() {
  float *A, *B, *out;
  hipMalloc(&A, 128 * 1024 * 4);
  hipMalloc(&B, 128 * 1024 * 4);
  hipMalloc(&out, 4096 * 256 * 4);
  hipMemset(A, 0, 128 * 1024 * 4);
  hipMemset(B, 0, 128 * 1024 * 4);
  all<Tile<128, 128, 2, 2, true, true>>("T128x128 kc/kc", A, B, out, 2);
  all<Tile<128, 128, 2, 2, false, false>>("T128x128 rc/rc", A, B, out, 2);
  all<Tile<128, 64, 4, 1, true, true>>("T128x64 kc/kc", A, B, out, 2);
  all<Tile<128, 32, 4, 1, true, true>>("T128x32 kc/kc", A, B, out, 3);
  all<Tile<64, 128, 2, 2, false, false>>("T64x128 rc/rc", A, B, out, 3);
  return 0;
}
