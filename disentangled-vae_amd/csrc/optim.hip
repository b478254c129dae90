// Multi-tensor gradient norm, clip and Adam for the video-VAE trainer.
// Tensor tables travel as kernel arguments (no device-side table to keep in sync), at most
// MT_MAX tensors per launch; each workgroup owns one fixed-size chunk of one tensor.
#include "common.h"
#include "../../include/dvae_hip.h"

#define ST(s) ((hipStream_t)(s))
#define MT_MAX 48
#define MT_CHUNK 8192  // elements per workgroup (256 threads x 8 float4)

struct MTTable {
  void* p[MT_MAX];
  void* g[MT_MAX];
  void* m[MT_MAX];
  void* v[MT_MAX];
  int64_t numel[MT_MAX];
  int chunk_start[MT_MAX + 1];  // first workgroup of each tensor
  int n;
};

__device__ __forceinline__ int mt_find(const MTTable& t, int blk) {
  int lo = 0, hi = t.n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (t.chunk_start[mid] <= blk) lo = mid; else hi = mid - 1;
  }
  return lo;
}

__global__ void __launch_bounds__(256)
mt_sumsq_kernel(MTTable t, double* __restrict__ partials, int part_offset) {
  __shared__ double red[4];
  const int ti = mt_find(t, blockIdx.x);
  const int64_t base = (int64_t)(blockIdx.x - t.chunk_start[ti]) * MT_CHUNK;
  const int64_t n = t.numel[ti];
  const float* g = (const float*)t.g[ti];
  double s = 0.0;
  const int64_t end = base + MT_CHUNK < n ? base + MT_CHUNK : n;
  if ((((uintptr_t)g) & 15) == 0) {
    for (int64_t i = base + threadIdx.x * 4; i < end; i += 1024) {
      if (i + 4 <= end) {
        const float4 v = *reinterpret_cast<const float4*>(g + i);
        s += (double)(v.x * v.x + v.y * v.y) + (double)(v.z * v.z + v.w * v.w);
      } else {
        for (int64_t j = i; j < end; ++j) s += (double)g[j] * (double)g[j];
      }
    }
  } else {
    for (int64_t i = base + threadIdx.x; i < end; i += 256) s += (double)g[i] * (double)g[i];
  }
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) partials[part_offset + blockIdx.x] = s;
}

__global__ void __launch_bounds__(256)
norm_finalize_kernel(const double* __restrict__ partials, int nparts, float max_norm, float* __restrict__ out) {
  __shared__ double red[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) s += partials[i];
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) {
    const float norm = (float)sqrt(s);
    out[0] = norm;
    float coef = 1.f;
    if (max_norm > 0.f) {
      coef = max_norm / (norm + 1e-6f);
      if (coef > 1.f) coef = 1.f;
    }
    out[1] = coef;
  }
}

__global__ void __launch_bounds__(256)
mt_scale_kernel(MTTable t, const float* __restrict__ coef) {
  const int ti = mt_find(t, blockIdx.x);
  const int64_t base = (int64_t)(blockIdx.x - t.chunk_start[ti]) * MT_CHUNK;
  const int64_t n = t.numel[ti];
  float* g = (float*)t.g[ti];
  const float c = coef[0];
  const int64_t end = base + MT_CHUNK < n ? base + MT_CHUNK : n;
  for (int64_t i = base + threadIdx.x; i < end; i += 256) g[i] *= c;
}

// dst tensor i <- src tensor i (p = destinations, g = sources): 16-byte pieces where both ends allow it
__global__ void __launch_bounds__(256)
mt_copy_kernel(MTTable t) {
  const int ti = mt_find(t, blockIdx.x);
  const int64_t base = (int64_t)(blockIdx.x - t.chunk_start[ti]) * MT_CHUNK;
  const int64_t n = t.numel[ti];
  float* __restrict__ d = (float*)t.p[ti];
  const float* __restrict__ s = (const float*)t.g[ti];
  const int64_t end = base + MT_CHUNK < n ? base + MT_CHUNK : n;
  if ((((uintptr_t)d | (uintptr_t)s) & 15) == 0) {
    const int64_t end4 = base + ((end - base) & ~(int64_t)3);
    for (int64_t i = base + 4 * threadIdx.x; i < end4; i += 1024)
      *reinterpret_cast<float4*>(d + i) = *reinterpret_cast<const float4*>(s + i);
    for (int64_t i = end4 + threadIdx.x; i < end; i += 256) d[i] = s[i];
  } else {
    for (int64_t i = base + threadIdx.x; i < end; i += 256) d[i] = s[i];
  }
}

// `dyn` (optional, device): {step count, learning rate, 1 - beta1^step, sqrt(1 - beta2^step)}, kept up
// to date by adam_advance_kernel.  When given, the step size and the bias corrections are read from it,
// so a captured graph replays correctly as the step count and the learning rate move; the host-side
// scalars are ignored.
__global__ void __launch_bounds__(256)
mt_adam_kernel(MTTable t, float lr, float beta1, float beta2, float eps, float wd, float bc1, float bc2_sqrt,
               const float* __restrict__ grad_coef, const float* __restrict__ dyn) {
  if (dyn) {
    lr = dyn[1];
    bc1 = dyn[2];
    bc2_sqrt = dyn[3];
  }
  const int ti = mt_find(t, blockIdx.x);
  const int64_t base = (int64_t)(blockIdx.x - t.chunk_start[ti]) * MT_CHUNK;
  const int64_t n = t.numel[ti];
  float* p = (float*)t.p[ti];
  const float* g = (const float*)t.g[ti];
  float* m = (float*)t.m[ti];
  float* v = (float*)t.v[ti];
  const float gc = grad_coef ? grad_coef[0] : 1.f;
  const float step_size = lr / bc1;
  const int64_t end = base + MT_CHUNK < n ? base + MT_CHUNK : n;
  const bool vec = ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0;
#define ADAM_ONE(P, G, Mm, V)                                   \
  {                                                             \
    float gg = (G) * gc;                                        \
    if (wd != 0.f) gg = fmaf(wd, (P), gg);                      \
    (Mm) = fmaf(beta1, (Mm), (1.f - beta1) * gg);               \
    (V) = fmaf(beta2, (V), (1.f - beta2) * gg * gg);            \
    const float denom = sqrtf(V) / bc2_sqrt + eps;              \
    (P) = (P) - step_size * ((Mm) / denom);                     \
  }
  if (vec) {
    for (int64_t i = base + threadIdx.x * 4; i < end; i += 1024) {
      if (i + 4 <= end) {
        float4 pv = *reinterpret_cast<float4*>(p + i);
        const float4 gv = *reinterpret_cast<const float4*>(g + i);
        float4 mv = *reinterpret_cast<float4*>(m + i);
        float4 vv = *reinterpret_cast<float4*>(v + i);
        ADAM_ONE(pv.x, gv.x, mv.x, vv.x)
        ADAM_ONE(pv.y, gv.y, mv.y, vv.y)
        ADAM_ONE(pv.z, gv.z, mv.z, vv.z)
        ADAM_ONE(pv.w, gv.w, mv.w, vv.w)
        *reinterpret_cast<float4*>(p + i) = pv;
        *reinterpret_cast<float4*>(m + i) = mv;
        *reinterpret_cast<float4*>(v + i) = vv;
      } else {
        for (int64_t j = i; j < end; ++j) ADAM_ONE(p[j], g[j], m[j], v[j])
      }
    }
  } else {
    for (int64_t i = base + threadIdx.x; i < end; i += 256) ADAM_ONE(p[i], g[i], m[i], v[i])
  }
#undef ADAM_ONE
}

static int fill_table(MTTable& t, int64_t first, int64_t n, const int64_t* numel) {
  int blocks = 0;
  t.n = 0;
  for (int64_t i = first; i < n && t.n < MT_MAX; ++i) {
    t.numel[t.n] = numel[i];
    t.chunk_start[t.n] = blocks;
    blocks += (int)ceil_div64(numel[i] > 0 ? numel[i] : 1, MT_CHUNK);
    ++t.n;
  }
  t.chunk_start[t.n] = blocks;
  return blocks;
}

static int64_t total_chunks(int64_t n, const int64_t* numel) {
  int64_t c = 0;
  for (int64_t i = 0; i < n; ++i) c += ceil_div64(numel[i] > 0 ? numel[i] : 1, MT_CHUNK);
  return c;
}

extern "C" int64_t dvae_grad_norm_ws_floats(int64_t n, const int64_t* numel) { return 2 * total_chunks(n, numel) + 2; }

extern "C" int dvae_grad_norm_f32(int64_t n, const float* const* grads, const int64_t* numel, float max_norm,
                                  float* norm_out, float* ws, dvae_stream_t stream) {
  DVAE_REQUIRE(n > 0, "grad_norm: no tensors");
  DVAE_REQUIRE(((uintptr_t)ws & 7) == 0, "grad_norm: workspace must be 8-byte aligned");
  int part = 0;
  for (int64_t first = 0; first < n; first += MT_MAX) {
    MTTable t;
    const int blocks = fill_table(t, first, n, numel);
    for (int i = 0; i < t.n; ++i) t.g[i] = (void*)grads[first + i];
    { double el = 0; for (int i = 0; i < t.n; ++i) el += (double)t.numel[i]; DVAE_WORK(4.0 * el, DVAE_UNIT_BYTE); }
    DVAE_LAUNCH(mt_sumsq_kernel, dim3(blocks), dim3(256), 0, ST(stream), t, (double*)ws, part);
    DVAE_LAUNCH_CHECK("grad_norm");
    part += blocks;
  }
  DVAE_LAUNCH(norm_finalize_kernel, dim3(1), dim3(256), 0, ST(stream), (const double*)ws, part, max_norm,
                     norm_out);
  DVAE_LAUNCH_CHECK("grad_norm_finalize");
  return 0;
}

extern "C" int dvae_scale_tensors_f32(int64_t n, float* const* tensors, const int64_t* numel, const float* coef,
                                      dvae_stream_t stream) {
  DVAE_REQUIRE(n > 0 && coef != nullptr, "scale_tensors: no tensors or no coefficient");
  for (int64_t first = 0; first < n; first += MT_MAX) {
    MTTable t;
    const int blocks = fill_table(t, first, n, numel);
    for (int i = 0; i < t.n; ++i) t.g[i] = (void*)tensors[first + i];
    DVAE_LAUNCH(mt_scale_kernel, dim3(blocks), dim3(256), 0, ST(stream), t, coef);
    DVAE_LAUNCH_CHECK("scale_tensors");
  }
  return 0;
}

extern "C" int dvae_copy_tensors_f32(int64_t n, float* const* dst, const float* const* src, const int64_t* numel,
                                     dvae_stream_t stream) {
  DVAE_REQUIRE(n > 0, "copy_tensors: no tensors");
  for (int64_t first = 0; first < n; first += MT_MAX) {
    MTTable t;
    const int blocks = fill_table(t, first, n, numel);
    for (int i = 0; i < t.n; ++i) {
      t.p[i] = (void*)dst[first + i];
      t.g[i] = (void*)src[first + i];
    }
    { double el = 0; for (int i = 0; i < t.n; ++i) el += (double)t.numel[i]; DVAE_WORK(8.0 * el, DVAE_UNIT_BYTE); }
    DVAE_LAUNCH(mt_copy_kernel, dim3(blocks), dim3(256), 0, ST(stream), t);
    DVAE_LAUNCH_CHECK("copy_tensors");
  }
  return 0;
}

// The corrections are formed in double from the double-precision betas and rounded to fp32 once, which is
// what a host that computes 1 - beta^t in double and passes it as a float argument does: a replayed step
// and an eagerly launched one see the same bits.
__global__ void adam_advance_kernel(float* dyn, double beta1, double beta2) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const float step = dyn[0] + 1.f;
    dyn[0] = step;
    dyn[2] = (float)(1.0 - pow(beta1, (double)step));
    dyn[3] = sqrtf((float)(1.0 - pow(beta2, (double)step)));
  }
}

extern "C" int dvae_adam_advance_f32(float* dyn, double beta1, double beta2, dvae_stream_t stream) {
  DVAE_REQUIRE(dyn != nullptr, "adam_advance: null state");
  DVAE_REQUIRE(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0, "adam_advance: betas must be in [0, 1)");
  DVAE_LAUNCH(adam_advance_kernel, dim3(1), dim3(64), 0, ST(stream), dyn, beta1, beta2);
  DVAE_LAUNCH_CHECK("adam_advance");
  return 0;
}

extern "C" int dvae_adam_step_f32(int64_t n, float* const* params, const float* const* grads, float* const* exp_avg,
                                  float* const* exp_avg_sq, const int64_t* numel, float lr, float beta1, float beta2,
                                  float eps, float weight_decay, float bias_correction1, float bias_correction2,
                                  const float* grad_coef, const float* dyn, dvae_stream_t stream) {
  DVAE_REQUIRE(n > 0, "adam_step: no tensors");
  DVAE_REQUIRE(dyn != nullptr || (bias_correction1 > 0.f && bias_correction2 > 0.f),
               "adam_step: bias corrections must be positive");
  const float bc2_sqrt = sqrtf(bias_correction2);
  for (int64_t first = 0; first < n; first += MT_MAX) {
    MTTable t;
    const int blocks = fill_table(t, first, n, numel);
    for (int i = 0; i < t.n; ++i) {
      t.p[i] = (void*)params[first + i];
      t.g[i] = (void*)grads[first + i];
      t.m[i] = (void*)exp_avg[first + i];
      t.v[i] = (void*)exp_avg_sq[first + i];
    }
    { double el = 0; for (int i = 0; i < t.n; ++i) el += (double)t.numel[i]; DVAE_WORK(28.0 * el, DVAE_UNIT_BYTE); }
    DVAE_LAUNCH(mt_adam_kernel, dim3(blocks), dim3(256), 0, ST(stream), t, lr, beta1, beta2, eps, weight_decay,
                       bias_correction1, bc2_sqrt, grad_coef, dyn);
    DVAE_LAUNCH_CHECK("adam_step");
  }
  return 0;
}
