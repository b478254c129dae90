// Shared helpers for the gfx950 kernels of the video-VAE hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <errno.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DVAE_WAVE 64
#define LRELU_SLOPE 0.2f

// activation codes shared with the host side
#define DVAE_ACT_NONE 0
#define DVAE_ACT_LRELU 1
#define DVAE_ACT_SIGMOID 2

void dvae_set_error(const char* fmt, ...);

// operand precision of the MFMA GEMM / convolution kernels (process-wide switch)
#define DVAE_PRECISION_F32 0
#define DVAE_PRECISION_BF16 1
#define DVAE_PRECISION_F32_SPLIT 2
int dvae_mfma_precision();

#define DVAE_REQUIRE(cond, ...)                 \
  do {                                          \
    if (!(cond)) {                              \
      dvae_set_error(__VA_ARGS__);              \
      return -EINVAL;                           \
    }                                           \
  } while (0)

#define DVAE_LAUNCH_CHECK(name)                                              \
  do {                                                                       \
    hipError_t e__ = hipGetLastError();                                      \
    if (e__ != hipSuccess) {                                                 \
      dvae_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return -EIO;                                                           \
    }                                                                        \
  } while (0)

// Launch tracing (see trace.hip): every launch of the library goes through DVAE_LAUNCH; a launcher may
// declare the algorithmic work of its next launch (FLOPs of a GEMM-shaped kernel, bytes an HBM-bound
// kernel has to move) with dvae_trace_work just before it.
#define DVAE_UNIT_NONE 0
#define DVAE_UNIT_FLOP 1
#define DVAE_UNIT_BYTE 2
extern int g_dvae_trace_on;
void dvae_trace_pre(hipStream_t s);
void dvae_trace_post(const void* fn, hipStream_t s);
void dvae_trace_work(double work, int unit);
#define DVAE_WORK(work, unit)                               \
  do {                                                      \
    if (g_dvae_trace_on) dvae_trace_work((double)(work), unit); \
  } while (0)
#define DVAE_LAUNCH(kernel, grid, block, lds, stream, ...)                 \
  do {                                                                     \
    if (g_dvae_trace_on) dvae_trace_pre(stream);                           \
    hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);     \
    if (g_dvae_trace_on) dvae_trace_post((const void*)(kernel), stream);   \
  } while (0)

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

__device__ __forceinline__ float lrelu(float x) { return x > 0.f ? x : x * LRELU_SLOPE; }

// affine (+ optional leaky-relu) applied to activations as they are loaded: the BatchNorm of the
// producing layer folded to y*scale+shift.  scale == nullptr means "use the value as stored".
__device__ __forceinline__ float4 affine_act4(float4 v, const float* __restrict__ scale,
                                              const float* __restrict__ shift, int c, int act) {
  if (scale) {
    const float4 s = *reinterpret_cast<const float4*>(scale + c);
    const float4 b = *reinterpret_cast<const float4*>(shift + c);
    v.x = fmaf(v.x, s.x, b.x);
    v.y = fmaf(v.y, s.y, b.y);
    v.z = fmaf(v.z, s.z, b.z);
    v.w = fmaf(v.w, s.w, b.w);
  }
  if (act == DVAE_ACT_LRELU) {
    v.x = lrelu(v.x);
    v.y = lrelu(v.y);
    v.z = lrelu(v.z);
    v.w = lrelu(v.w);
  }
  return v;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// Sum over a 256-thread block; result valid in thread 0.  `red` holds >= 4 doubles.
__device__ __forceinline__ double block_sum_256(double v, double* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) r = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  return r;
}
