// HBM-bound kernels of the video-VAE hot path: content/motion split, losses, BatchNorm glue,
// activation backward.  All fp32 in memory; reductions accumulate in fp64 and use a fixed
// two-stage order (per-workgroup partials, then one workgroup), so results are run-to-run stable.
#include "common.h"
#include "../../include/dvae_hip.h"
#include <stdarg.h>
#include <stdio.h>
#include <mutex>

static thread_local char g_err[512] = "";

void dvae_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* dvae_last_error(void) { return g_err; }
extern "C" int dvae_version(void) { return 2; }

static int g_mfma_precision = DVAE_PRECISION_F32;
int dvae_mfma_precision() { return g_mfma_precision; }
extern "C" int dvae_get_mfma_precision(void) { return g_mfma_precision; }
extern "C" int dvae_set_mfma_precision(int precision) {
  DVAE_REQUIRE(precision == DVAE_PRECISION_F32 || precision == DVAE_PRECISION_BF16 ||
                   precision == DVAE_PRECISION_F32_SPLIT,
               "set_mfma_precision: unknown precision %d", precision);
  g_mfma_precision = precision;
  return 0;
}

#define ST(s) ((hipStream_t)(s))
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
static bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// ------------------------------------------------------------------------------------------------
// temporal mean + frame difference: one pass over the clip, 16 B per lane, T loads in flight.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
split_mean_diff_kernel(const float4* __restrict__ video, float4* __restrict__ mean,
                       float4* __restrict__ diff, int64_t total, int T, int64_t P4) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int64_t b = idx / P4, p = idx - b * P4;
  const float4* src = video + b * T * P4 + p;
  float4* d = diff + b * (T - 1) * P4 + p;
  float4 prev = src[0];
  float4 sum = prev;
#pragma unroll 4
  for (int t = 1; t < T; ++t) {
    const float4 cur = src[(int64_t)t * P4];
    d[(int64_t)(t - 1) * P4] = make_float4(cur.x - prev.x, cur.y - prev.y, cur.z - prev.z, cur.w - prev.w);
    sum.x += cur.x; sum.y += cur.y; sum.z += cur.z; sum.w += cur.w;
    prev = cur;
  }
  const float inv = (float)T;
  mean[idx] = make_float4(sum.x / inv, sum.y / inv, sum.z / inv, sum.w / inv);
}

extern "C" int dvae_split_mean_diff_f32(const float* video, float* mean, float* diff, int64_t B, int64_t T,
                                        int64_t P, dvae_stream_t stream) {
  DVAE_REQUIRE(B > 0 && T >= 1 && P > 0 && P % 4 == 0, "split_mean_diff: need B>0, T>=1, P%%4==0 (B=%ld T=%ld P=%ld)",
               (long)B, (long)T, (long)P);
  const int64_t total = B * (P / 4);
  DVAE_WORK(4.0 * ((double)B * T * P + (double)B * P + (double)B * (T - 1) * P), DVAE_UNIT_BYTE);
  DVAE_LAUNCH(split_mean_diff_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, ST(stream),
                     (const float4*)video, (float4*)mean, (float4*)diff, total, (int)T, P / 4);
  DVAE_LAUNCH_CHECK("split_mean_diff");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Moving-MNIST renderer: every frame is the max-composite of num_digits 28x28 (digit_size) patches at
// integer positions that the host simulated (seeded trajectories stay on the host so the draws are
// those of numpy's generator).  One thread writes four horizontally adjacent pixels; bound by the
// write of the clip.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
render_clips_kernel(const float* __restrict__ bank, const int* __restrict__ digit_idx, const int* __restrict__ pos,
                    float4* __restrict__ video, int64_t total4, int T, int nd, int S, int D) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const int s4 = S / 4;
  const int px0 = (int)(i % s4) * 4;
  const int py = (int)((i / s4) % S);
  const int64_t frame = i / ((int64_t)s4 * S);          // b*T + t
  const int64_t b = frame / T;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  for (int d = 0; d < nd; ++d) {
    const int x = pos[(frame * nd + d) * 2 + 0], y = pos[(frame * nd + d) * 2 + 1];
    const int dy = py - y;
    if (dy < 0 || dy >= D) continue;
    const float* img = bank + ((int64_t)digit_idx[b * nd + d] * D + dy) * D;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int dx = px0 + k - x;
      if (dx >= 0 && dx < D) v[k] = fmaxf(v[k], img[dx]);
    }
  }
  video[i] = make_float4(v[0], v[1], v[2], v[3]);
}

extern "C" int dvae_render_clips_f32(const float* bank, int64_t bank_size, const int32_t* digit_idx,
                                     const int32_t* positions, float* video, int64_t B, int64_t T, int64_t num_digits,
                                     int64_t frame_size, int64_t digit_size, dvae_stream_t stream) {
  DVAE_REQUIRE(B > 0 && T > 0 && num_digits > 0 && bank_size > 0, "render_clips: empty");
  DVAE_REQUIRE(frame_size % 4 == 0 && digit_size > 0 && digit_size <= frame_size,
               "render_clips: frame_size must be a multiple of 4 and >= digit_size (frame %ld digit %ld)",
               (long)frame_size, (long)digit_size);
  DVAE_REQUIRE((((uintptr_t)video) & 15) == 0, "render_clips: video must be 16-byte aligned");
  const int64_t total4 = B * T * frame_size * (frame_size / 4);
  DVAE_LAUNCH(render_clips_kernel, dim3((unsigned)ceil_div64(total4, 256)), dim3(256), 0, ST(stream), bank,
                     (const int*)digit_idx, (const int*)positions, (float4*)video, total4, (int)T, (int)num_digits,
                     (int)frame_size, (int)digit_size);
  DVAE_LAUNCH_CHECK("render_clips");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// reconstruction loss
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float recon_elem(float r, float x, int kind) {
  if (kind == 0) {
    const float e = r - x;
    return e * e;
  }
  const float lp = fmaxf(logf(r), -100.f);
  const float lq = fmaxf(logf(1.f - r), -100.f);
  return -(x * lp + (1.f - x) * lq);
}

#define RL_MAX_BLOCKS 2048

__global__ void __launch_bounds__(256)
recon_loss_fwd_kernel(const float4* __restrict__ recon, const float4* __restrict__ target,
                      double* __restrict__ partials, int64_t total, int T, int64_t P4, int kind, int with_temporal) {
  __shared__ double red[4];
  double s_rec = 0.0, s_tmp = 0.0;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t b = idx / P4, p = idx - b * P4;
    const float4* r = recon + b * T * P4 + p;
    const float4* x = target + b * T * P4 + p;
    float4 rp = r[0], xp = x[0];
    float acc = recon_elem(rp.x, xp.x, kind) + recon_elem(rp.y, xp.y, kind) + recon_elem(rp.z, xp.z, kind) +
                recon_elem(rp.w, xp.w, kind);
    float acct = 0.f;
    for (int t = 1; t < T; ++t) {
      const float4 rc = r[(int64_t)t * P4], xc = x[(int64_t)t * P4];
      acc += recon_elem(rc.x, xc.x, kind) + recon_elem(rc.y, xc.y, kind) + recon_elem(rc.z, xc.z, kind) +
             recon_elem(rc.w, xc.w, kind);
      if (with_temporal) {
        const float d0 = (rc.x - rp.x) - (xc.x - xp.x), d1 = (rc.y - rp.y) - (xc.y - xp.y);
        const float d2 = (rc.z - rp.z) - (xc.z - xp.z), d3 = (rc.w - rp.w) - (xc.w - xp.w);
        acct += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
      }
      rp = rc;
      xp = xc;
    }
    s_rec += (double)acc;
    s_tmp += (double)acct;
  }
  const double a = block_sum_256(s_rec, red);
  const double b2 = block_sum_256(s_tmp, red);
  if (threadIdx.x == 0) {
    partials[2 * blockIdx.x] = a;
    partials[2 * blockIdx.x + 1] = b2;
  }
}

__global__ void __launch_bounds__(256)
recon_loss_finalize_kernel(const double* __restrict__ partials, int nparts, float* __restrict__ out, double B,
                           double T, double P) {
  __shared__ double red[4];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) {
    a += partials[2 * i];
    b += partials[2 * i + 1];
  }
  a = block_sum_256(a, red);
  b = block_sum_256(b, red);
  if (threadIdx.x == 0) {
    const float per_pixel = (float)(a / (B * T * P));
    out[0] = per_pixel;
    out[1] = per_pixel * (float)(T * P);
    out[2] = T > 1 ? (float)(b / (B * (T - 1) * P)) * (float)((T - 1) * P) : 0.f;
  }
}

static int recon_blocks(int64_t B, int64_t P) {
  int64_t n = ceil_div64(B * (P / 4), 256);
  return (int)(n < RL_MAX_BLOCKS ? n : RL_MAX_BLOCKS);
}

extern "C" int64_t dvae_recon_loss_ws_floats(int64_t B, int64_t T, int64_t P) {
  (void)T;
  return 4 * (int64_t)RL_MAX_BLOCKS + 4;  // doubles, 2 per block
}

extern "C" int dvae_recon_loss_fwd_f32(const float* recon, const float* target, float* out, float* ws, int64_t B,
                                       int64_t T, int64_t P, int kind, int with_temporal, dvae_stream_t stream) {
  DVAE_REQUIRE(B > 0 && T >= 1 && P > 0 && P % 4 == 0, "recon_loss_fwd: need P%%4==0");
  DVAE_REQUIRE(kind == 0 || kind == 1, "recon_loss_fwd: kind must be 0 (mse) or 1 (bce)");
  DVAE_REQUIRE(((uintptr_t)ws & 7) == 0, "recon_loss_fwd: workspace must be 8-byte aligned");
  const int nb = recon_blocks(B, P);
  DVAE_WORK(8.0 * (double)B * T * P, DVAE_UNIT_BYTE);                      // recon and target, once each
  DVAE_LAUNCH(recon_loss_fwd_kernel, dim3(nb), dim3(256), 0, ST(stream), (const float4*)recon,
                     (const float4*)target, (double*)ws, B * (P / 4), (int)T, P / 4, kind, with_temporal);
  DVAE_LAUNCH_CHECK("recon_loss_fwd");
  DVAE_LAUNCH(recon_loss_finalize_kernel, dim3(1), dim3(256), 0, ST(stream), (const double*)ws, nb, out,
                     (double)B, (double)T, (double)P);
  DVAE_LAUNCH_CHECK("recon_loss_finalize");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// reconstruction metrics of the evaluation loop (reference: evaluation/metrics.py:14-22, :48-50) in
// one pass over the two clips: a thread walks one float4 column of a clip through time, so every
// element is read once and the previous frame stays in registers.
//   out[0] = mean((recon - target)^2)          out[1] = 10 log10(max_val^2 / out[0])  (+inf when 0)
//   out[2] = mean(|recon[:,1:] - recon[:,:-1]|)  (NaN when T == 1: the mean of no elements, as in torch)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
recon_metrics_kernel(const float4* __restrict__ recon, const float4* __restrict__ target,
                     double* __restrict__ partials, int64_t total, int T, int64_t P4) {
  __shared__ double red[4];
  double s_sq = 0.0, s_ab = 0.0;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t b = idx / P4, p = idx - b * P4;
    const float4* r = recon + b * T * P4 + p;
    const float4* x = target + b * T * P4 + p;
    float4 rp = r[0];
    const float4 x0 = x[0];
    float sq = (rp.x - x0.x) * (rp.x - x0.x) + (rp.y - x0.y) * (rp.y - x0.y) + (rp.z - x0.z) * (rp.z - x0.z) +
               (rp.w - x0.w) * (rp.w - x0.w);
    float ab = 0.f;
    for (int t = 1; t < T; ++t) {
      const float4 rc = r[(int64_t)t * P4], xc = x[(int64_t)t * P4];
      sq += (rc.x - xc.x) * (rc.x - xc.x) + (rc.y - xc.y) * (rc.y - xc.y) + (rc.z - xc.z) * (rc.z - xc.z) +
            (rc.w - xc.w) * (rc.w - xc.w);
      ab += fabsf(rc.x - rp.x) + fabsf(rc.y - rp.y) + fabsf(rc.z - rp.z) + fabsf(rc.w - rp.w);
      rp = rc;
    }
    s_sq += (double)sq;
    s_ab += (double)ab;
  }
  const double a = block_sum_256(s_sq, red);
  const double b2 = block_sum_256(s_ab, red);
  if (threadIdx.x == 0) {
    partials[2 * blockIdx.x] = a;
    partials[2 * blockIdx.x + 1] = b2;
  }
}

__global__ void __launch_bounds__(256)
recon_metrics_finalize_kernel(const double* __restrict__ partials, int nparts, float* __restrict__ out, double B,
                              double T, double P, double max_val) {
  __shared__ double red[4];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) {
    a += partials[2 * i];
    b += partials[2 * i + 1];
  }
  a = block_sum_256(a, red);
  b = block_sum_256(b, red);
  if (threadIdx.x == 0) {
    const float mse = (float)(a / (B * T * P));        // rounded to fp32 first, as F.mse_loss(...).item() is
    out[0] = mse;
    out[1] = mse == 0.f ? __builtin_inff() : (float)(10.0 * log10(max_val * max_val / (double)mse));
    out[2] = T > 1 ? (float)(b / (B * (T - 1) * P)) : __builtin_nanf("");   // mean of no elements
  }
}

extern "C" int dvae_recon_metrics_f32(const float* recon, const float* target, float* out, float* ws, int64_t B,
                                      int64_t T, int64_t P, float max_val, dvae_stream_t stream) {
  DVAE_REQUIRE(B > 0 && T >= 1 && P > 0 && P % 4 == 0, "recon_metrics: need P%%4==0 (P=%ld)", (long)P);
  DVAE_REQUIRE(((((uintptr_t)recon) | ((uintptr_t)target)) & 15) == 0, "recon_metrics: clips must be 16-byte aligned");
  DVAE_REQUIRE(ws != nullptr && ((uintptr_t)ws & 7) == 0, "recon_metrics: workspace must be 8-byte aligned");
  const int nb = recon_blocks(B, P);
  DVAE_LAUNCH(recon_metrics_kernel, dim3(nb), dim3(256), 0, ST(stream), (const float4*)recon,
                     (const float4*)target, (double*)ws, B * (P / 4), (int)T, P / 4);
  DVAE_LAUNCH_CHECK("recon_metrics");
  DVAE_LAUNCH(recon_metrics_finalize_kernel, dim3(1), dim3(256), 0, ST(stream), (const double*)ws, nb, out,
                     (double)B, (double)T, (double)P, (double)max_val);
  DVAE_LAUNCH_CHECK("recon_metrics_finalize");
  return 0;
}

// ---- the same metrics for callers that cannot allocate -------------------------------------------
// The per-workgroup partial sums live in a file-scope device array instead of a caller-provided
// workspace.  Every GPU holds its own copy of the array at its own address: the address is looked up
// once per device (hipGetSymbolAddress with that device current) and only that pointer is ever handed
// to a kernel -- the array's name in host code is a host-side stand-in, not a device address.
#define NOALLOC_MAX_DEVICES 64
static __device__ __align__(256) double scratch[2 * RL_MAX_BLOCKS];          // 32 KiB per GPU
static double* g_scratch_ptr[NOALLOC_MAX_DEVICES];
static std::mutex g_scratch_mutex;

// device pointer of `scratch` on the current device (nullptr + error message on failure)
static double* scratch_on_current_device() {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess || dev < 0 || dev >= NOALLOC_MAX_DEVICES) {
    dvae_set_error("recon_metrics_noalloc: cannot identify the current device (%s, device %d, at most %d supported)",
                   hipGetErrorString(e), dev, NOALLOC_MAX_DEVICES);
    return nullptr;
  }
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (g_scratch_ptr[dev] == nullptr) {
    void* p = nullptr;
    e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess || p == nullptr) {
      dvae_set_error("recon_metrics_noalloc: hipGetSymbolAddress failed on device %d: %s", dev, hipGetErrorString(e));
      return nullptr;
    }
    g_scratch_ptr[dev] = (double*)p;
  }
  return g_scratch_ptr[dev];
}

extern "C" int64_t dvae_recon_metrics_noalloc_max_elements(void) { return DVAE_NOALLOC_MAX_ELEMENTS; }

// Looks the scratch array's address up on the current device now, so that the first dvae_recon_metrics_noalloc_f32
// on this device makes no runtime call besides its two launches (a symbol look-up may load the code object, which a
// stream under graph capture does not allow).
extern "C" int dvae_recon_metrics_noalloc_init(void) { return scratch_on_current_device() ? 0 : -EIO; }

extern "C" int dvae_recon_metrics_noalloc_f32(const float* recon, const float* target, float* out, int64_t B,
                                              int64_t T, int64_t P, float max_val, dvae_stream_t stream) {
  DVAE_REQUIRE(B > 0 && T >= 1 && P > 0 && P % 4 == 0, "recon_metrics_noalloc: need P%%4==0 (P=%ld)", (long)P);
  DVAE_REQUIRE(B <= DVAE_NOALLOC_MAX_ELEMENTS / T / P,
               "recon_metrics_noalloc: B*T*P = %ld*%ld*%ld elements per clip tensor exceeds the limit of %ld; use "
               "dvae_recon_metrics_f32 with a caller-provided workspace",
               (long)B, (long)T, (long)P, (long)DVAE_NOALLOC_MAX_ELEMENTS);
  DVAE_REQUIRE(recon && target && out, "recon_metrics_noalloc: null argument");
  DVAE_REQUIRE(((((uintptr_t)recon) | ((uintptr_t)target)) & 15) == 0,
               "recon_metrics_noalloc: clips must be 16-byte aligned");
  double* ws = scratch_on_current_device();
  if (ws == nullptr) return -EIO;
  // every partial the finalize kernel reads is written by this launch first: nothing to clear
  const int nb = recon_blocks(B, P);
  DVAE_LAUNCH(recon_metrics_kernel, dim3(nb), dim3(256), 0, ST(stream), (const float4*)recon,
                     (const float4*)target, ws, B * (P / 4), (int)T, P / 4);
  DVAE_LAUNCH_CHECK("recon_metrics_noalloc");
  DVAE_LAUNCH(recon_metrics_finalize_kernel, dim3(1), dim3(256), 0, ST(stream), (const double*)ws, nb, out,
                     (double)B, (double)T, (double)P, (double)max_val);
  DVAE_LAUNCH_CHECK("recon_metrics_noalloc_finalize");
  return 0;
}

__device__ __forceinline__ float recon_grad_elem(float r, float x, int kind) {
  if (kind == 0) return 2.f * (r - x);
  // F.binary_cross_entropy backward: (r - x) / max((1 - r) * r, 1e-12)
  return (r - x) / fmaxf((1.f - r) * r, 1e-12f);
}

__global__ void sum_partials_kernel(const double* __restrict__ partials, int n, float* out);   // defined below

// LOGIT: the reconstruction is the sigmoid output of the decoder's last layer and the caller wants the gradient
// with respect to that layer's pre-activation: every element is multiplied by r(1-r) on its way out (the same
// float operations, in the same order, as a separate sigmoid-backward pass over the stored gradient), and the
// workgroup's sum of what it wrote goes to partials[blockIdx.x] (the last layer's bias gradient).
template <bool LOGIT>
__global__ void __launch_bounds__(256)
recon_loss_bwd_kernel(const float4* __restrict__ recon, const float4* __restrict__ target,
                      const float* __restrict__ gscale, float4* __restrict__ drecon, int64_t total, int T,
                      int64_t P4, int kind, float w_rec, float w_tmp, double* __restrict__ partials) {
  __shared__ double red[4];
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  double sum = 0.0;
  if (idx < total) {
    const float g = gscale ? gscale[0] : 1.f;
    const float wr = w_rec * g, wt = w_tmp * g;
    const int64_t b = idx / P4, p = idx - b * P4;
    const float4* r = recon + b * T * P4 + p;
    const float4* x = target + b * T * P4 + p;
    float4* d = drecon + b * T * P4 + p;
    float4 rc = r[0], xc = x[0];
    float4 dprev = make_float4(0.f, 0.f, 0.f, 0.f);  // (dr - dx) between t-1 and t
    for (int t = 0; t < T; ++t) {
      float4 rn = rc, xn = xc;
      float4 dnext = make_float4(0.f, 0.f, 0.f, 0.f);
      if (t + 1 < T) {
        rn = r[(int64_t)(t + 1) * P4];
        xn = x[(int64_t)(t + 1) * P4];
        if (wt != 0.f)
          dnext = make_float4((rn.x - rc.x) - (xn.x - xc.x), (rn.y - rc.y) - (xn.y - xc.y),
                              (rn.z - rc.z) - (xn.z - xc.z), (rn.w - rc.w) - (xn.w - xc.w));
      }
      float4 o;
      o.x = wr * recon_grad_elem(rc.x, xc.x, kind) + wt * (dprev.x - dnext.x);
      o.y = wr * recon_grad_elem(rc.y, xc.y, kind) + wt * (dprev.y - dnext.y);
      o.z = wr * recon_grad_elem(rc.z, xc.z, kind) + wt * (dprev.z - dnext.z);
      o.w = wr * recon_grad_elem(rc.w, xc.w, kind) + wt * (dprev.w - dnext.w);
      if (LOGIT) {
        o.x = o.x * rc.x * (1.f - rc.x);
        o.y = o.y * rc.y * (1.f - rc.y);
        o.z = o.z * rc.z * (1.f - rc.z);
        o.w = o.w * rc.w * (1.f - rc.w);
        sum += ((double)o.x + (double)o.y) + ((double)o.z + (double)o.w);
      }
      d[(int64_t)t * P4] = o;
      dprev = dnext;
      rc = rn;
      xc = xn;
    }
  }
  if (LOGIT) {
    sum = block_sum_256(sum, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = sum;
  }
}

extern "C" int dvae_recon_loss_bwd_f32(const float* recon, const float* target, const float* gscale, float* drecon,
                                       int64_t B, int64_t T, int64_t P, int kind, float temporal_weight,
                                       dvae_stream_t stream) {
  DVAE_REQUIRE(B > 0 && T >= 1 && P > 0 && P % 4 == 0, "recon_loss_bwd: need P%%4==0");
  DVAE_REQUIRE(kind == 0 || kind == 1, "recon_loss_bwd: kind must be 0 (mse) or 1 (bce)");
  const int64_t total = B * (P / 4);
  // out[1] = sum(l)/B ; out[2] = sum(d^2)/B
  const float w_rec = 1.f / (float)B;
  const float w_tmp = (T > 1) ? 2.f * temporal_weight / (float)B : 0.f;
  DVAE_WORK(12.0 * (double)B * T * P, DVAE_UNIT_BYTE);                     // recon, target in; gradient out
  DVAE_LAUNCH(recon_loss_bwd_kernel<false>, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, ST(stream),
                     (const float4*)recon, (const float4*)target, gscale, (float4*)drecon, total, (int)T, P / 4,
                     kind, w_rec, w_tmp, (double*)nullptr);
  DVAE_LAUNCH_CHECK("recon_loss_bwd");
  return 0;
}

extern "C" int64_t dvae_recon_loss_bwd_logit_ws_floats(int64_t B, int64_t P) {
  return 2 * ceil_div64(B * (P / 4), 256);      // one double per workgroup
}

extern "C" int dvae_recon_loss_bwd_logit_f32(const float* recon, const float* target, const float* gscale,
                                             float* dlogit, float* dbias, float* ws, int64_t B, int64_t T, int64_t P,
                                             int kind, float temporal_weight, dvae_stream_t stream) {
  DVAE_REQUIRE(B > 0 && T >= 1 && P > 0 && P % 4 == 0, "recon_loss_bwd_logit: need P%%4==0");
  DVAE_REQUIRE(kind == 0 || kind == 1, "recon_loss_bwd_logit: kind must be 0 (mse) or 1 (bce)");
  DVAE_REQUIRE(ws != nullptr && ((uintptr_t)ws & 7) == 0, "recon_loss_bwd_logit: workspace must be 8-byte aligned");
  const int64_t total = B * (P / 4);
  const int64_t nb = ceil_div64(total, 256);
  DVAE_REQUIRE(nb < (int64_t)1 << 31, "recon_loss_bwd_logit: too many elements");
  const float w_rec = 1.f / (float)B;
  const float w_tmp = (T > 1) ? 2.f * temporal_weight / (float)B : 0.f;
  DVAE_WORK(12.0 * (double)B * T * P, DVAE_UNIT_BYTE);
  DVAE_LAUNCH(recon_loss_bwd_kernel<true>, dim3((unsigned)nb), dim3(256), 0, ST(stream), (const float4*)recon,
                     (const float4*)target, gscale, (float4*)dlogit, total, (int)T, P / 4, kind, w_rec, w_tmp,
                     (double*)ws);
  DVAE_LAUNCH_CHECK("recon_loss_bwd_logit");
  if (dbias) {
    DVAE_LAUNCH(sum_partials_kernel, dim3(1), dim3(256), 0, ST(stream), (const double*)ws, (int)nb, dbias);
    DVAE_LAUNCH_CHECK("recon_loss_bwd_logit_sum");
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------
// reparameterisation and KL (tiny tensors: [B, latent])
// ------------------------------------------------------------------------------------------------
__global__ void reparam_fwd_kernel(const float* mu, const float* logvar, const float* eps, float* z, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) z[i] = mu[i] + eps[i] * expf(0.5f * logvar[i]);
}

__global__ void reparam_bwd_kernel(const float* dz, const float* logvar, const float* eps, float* dmu,
                                   float* dlogvar, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const float g = dz[i];
    dmu[i] = g;
    dlogvar[i] = g * eps[i] * 0.5f * expf(0.5f * logvar[i]);
  }
}

extern "C" int dvae_reparam_fwd_f32(const float* mu, const float* logvar, const float* eps, float* z, int64_t n,
                                    dvae_stream_t stream) {
  DVAE_REQUIRE(n > 0, "reparam_fwd: empty");
  DVAE_LAUNCH(reparam_fwd_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, ST(stream), mu, logvar,
                     eps, z, n);
  DVAE_LAUNCH_CHECK("reparam_fwd");
  return 0;
}

extern "C" int dvae_reparam_bwd_f32(const float* dz, const float* logvar, const float* eps, float* dmu,
                                    float* dlogvar, int64_t n, dvae_stream_t stream) {
  DVAE_REQUIRE(n > 0, "reparam_bwd: empty");
  DVAE_LAUNCH(reparam_bwd_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, ST(stream), dz, logvar,
                     eps, dmu, dlogvar, n);
  DVAE_LAUNCH_CHECK("reparam_bwd");
  return 0;
}

// One workgroup of 1024 threads (the sums are two numbers; a second launch to combine partials would
// cost more than it saves), four independent partial sums per thread so that the loads overlap.
__global__ void __launch_bounds__(1024)
kl_fwd_kernel(const float* __restrict__ mu, const float* __restrict__ logvar, float* __restrict__ out, int64_t B,
              int64_t D, float free_bits) {
  __shared__ double red[2][16];
  double s_floor[4] = {0.0, 0.0, 0.0, 0.0}, s_raw[4] = {0.0, 0.0, 0.0, 0.0};
  const int64_t n = B * D;
  for (int64_t i0 = threadIdx.x; i0 < n; i0 += 4096) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t i = i0 + 1024 * u;
      if (i < n) {
        const float m = mu[i], lv = logvar[i];
        const float kl = -0.5f * (1.f + lv - m * m - expf(lv));
        s_raw[u] += (double)kl;
        s_floor[u] += (double)((free_bits > 0.f) ? fmaxf(kl, free_bits) : kl);
      }
    }
  }
  const double a = wave_sum((s_floor[0] + s_floor[1]) + (s_floor[2] + s_floor[3]));
  const double b = wave_sum((s_raw[0] + s_raw[1]) + (s_raw[2] + s_raw[3]));
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = a;
    red[1][threadIdx.x >> 6] = b;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double sf = 0.0, sr = 0.0;
    for (int w = 0; w < 16; ++w) {
      sf += red[0][w];
      sr += red[1][w];
    }
    out[0] = (float)(sf / (double)B);
    out[1] = (float)(sr / (double)B);
    out[2] = (float)(sf / (double)n);
  }
}

__global__ void kl_bwd_kernel(const float* mu, const float* logvar, const float* gscale, float coef,
                              const float* coef_dev, float inv_b, float* dmu, float* dlogvar, int64_t n,
                              float free_bits) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  // the weight of the term: a launch constant, or (coef_dev) a device scalar that may change between replays
  const float c = ((coef_dev ? coef_dev[0] : coef) * inv_b) * (gscale ? gscale[0] : 1.f);
  const float m = mu[i], lv = logvar[i];
  const float e = expf(lv);
  float gm = c * m, gl = c * 0.5f * (e - 1.f);
  if (free_bits > 0.f) {
    const float kl = -0.5f * (1.f + lv - m * m - e);
    if (!(kl > free_bits)) {  // the floor is active: maximum() passes no gradient to kl
      gm = 0.f;
      gl = 0.f;
    }
  }
  dmu[i] = gm;
  dlogvar[i] = gl;
}

extern "C" int dvae_kl_fwd_f32(const float* mu, const float* logvar, float* out, int64_t B, int64_t D,
                               float free_bits, dvae_stream_t stream) {
  DVAE_REQUIRE(B > 0 && D > 0, "kl_fwd: empty");
  DVAE_LAUNCH(kl_fwd_kernel, dim3(1), dim3(1024), 0, ST(stream), mu, logvar, out, B, D, free_bits);
  DVAE_LAUNCH_CHECK("kl_fwd");
  return 0;
}

extern "C" int dvae_kl_bwd_f32(const float* mu, const float* logvar, const float* gscale, float coef,
                               const float* coef_dev, float* dmu, float* dlogvar, int64_t B, int64_t D,
                               float free_bits, dvae_stream_t stream) {
  DVAE_REQUIRE(B > 0 && D > 0, "kl_bwd: empty");
  const int64_t n = B * D;
  DVAE_LAUNCH(kl_bwd_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, ST(stream), mu, logvar,
                     gscale, coef, coef_dev, 1.f / (float)B, dmu, dlogvar, n, free_bits);
  DVAE_LAUNCH_CHECK("kl_bwd");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// The objective in two launches: the pass over the clip (recon_loss_fwd_kernel: per-workgroup partial sums
// of the reconstruction and temporal terms), then ONE workgroup that finishes everything small -- sums the
// partials, computes the KL terms of one or two latents, applies the weights and writes every entry of the
// loss dictionary.  out layout 0 (one latent, 7 floats): total, recon per pixel, recon, temporal, kl, kl raw,
// kl per dim.  Layout 1 (two latents, 12 floats): total, kl, recon per pixel, recon, temporal, content kl
// (mean, raw, per dim), motion kl (same), independence.
// ------------------------------------------------------------------------------------------------
struct LossFinishArgs {
  const double* partials; int nparts;
  double B, T, P;
  const float* mu[2]; const float* lv[2]; int64_t D[2];
  float beta[2];
  const float* beta_dev;     // device scalar that replaces beta[p] for the latents named in beta_dev_mask, or nullptr
  int beta_dev_mask;
  int nkl;
  float free_bits, temporal_weight, indep_weight;
  const float* indep;        // value of the independence term (device scalar) or nullptr
  int layout;
  float* out;
  float* total_out;          // the total once more, as a tensor of its own (nullptr: not wanted)
};

__global__ void __launch_bounds__(1024) vae_loss_finish_kernel(LossFinishArgs a) {
  __shared__ double red[6][16];
  const int wave = threadIdx.x >> 6;
  // One workgroup on an otherwise idle chip: what this kernel costs is the number of memory round trips a thread
  // makes one after the other, so every loop below requests a batch of independent loads before it uses any.
  double s_rec = 0.0, s_tmp = 0.0;
  {
    for (int i0 = threadIdx.x; i0 < a.nparts; i0 += 4 * 1024) {
      double r[4], t[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + 1024 * u;
        r[u] = i < a.nparts ? a.partials[2 * i] : 0.0;
        t[u] = i < a.nparts ? a.partials[2 * i + 1] : 0.0;
      }
      s_rec += (r[0] + r[1]) + (r[2] + r[3]);
      s_tmp += (t[0] + t[1]) + (t[2] + t[3]);
    }
  }
  double kf[2] = {0.0, 0.0}, kr[2] = {0.0, 0.0};
  const int64_t rows = (int64_t)a.B;
  for (int p = 0; p < a.nkl; ++p) {
    const int64_t n = rows * a.D[p];
    double s_floor = 0.0, s_raw = 0.0;
    const bool quads = n % 4 == 0 && ((((uintptr_t)a.mu[p]) | ((uintptr_t)a.lv[p])) & 15) == 0;
    if (quads) {
      const float4* mu4 = reinterpret_cast<const float4*>(a.mu[p]);
      const float4* lv4 = reinterpret_cast<const float4*>(a.lv[p]);
      const int64_t n4 = n / 4;
      constexpr int U = 8;                     // 8 x 1024 threads x 4 values: a 256 x 128 latent in one pass
      for (int64_t i0 = threadIdx.x; i0 < n4; i0 += U * 1024) {
        float4 m[U], l[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int64_t i = i0 + 1024 * u;
          m[u] = i < n4 ? mu4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
          l[u] = i < n4 ? lv4[i] : make_float4(0.f, 0.f, 0.f, 0.f);       // kl(0, 0) = 0 ...
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (i0 + 1024 * u < n4) {                                         // ... but the floor of a free-bits term is not
            const float mm[4] = {m[u].x, m[u].y, m[u].z, m[u].w}, ll[4] = {l[u].x, l[u].y, l[u].z, l[u].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float kl = -0.5f * (1.f + ll[e] - mm[e] * mm[e] - expf(ll[e]));
              s_raw += (double)kl;
              s_floor += (double)((a.free_bits > 0.f) ? fmaxf(kl, a.free_bits) : kl);
            }
          }
        }
      }
    } else {
      for (int64_t i = threadIdx.x; i < n; i += 1024) {
        const float m = a.mu[p][i], lv = a.lv[p][i];
        const float kl = -0.5f * (1.f + lv - m * m - expf(lv));
        s_raw += (double)kl;
        s_floor += (double)((a.free_bits > 0.f) ? fmaxf(kl, a.free_bits) : kl);
      }
    }
    kf[p] = s_floor;
    kr[p] = s_raw;
  }
  const double v[6] = {s_rec, s_tmp, kf[0], kr[0], kf[1], kr[1]};
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    const double w = wave_sum(v[q]);
    if ((threadIdx.x & 63) == 0) red[q][wave] = w;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t[6];
    for (int q = 0; q < 6; ++q) {
      double acc = 0.0;
      for (int w = 0; w < 16; ++w) acc += red[q][w];
      t[q] = acc;
    }
    const float per_pixel = (float)(t[0] / (a.B * a.T * a.P));
    const float rec = per_pixel * (float)(a.T * a.P);
    const float tmp = a.T > 1 ? (float)(t[1] / (a.B * (a.T - 1) * a.P)) * (float)((a.T - 1) * a.P) : 0.f;
    float kl_mean[2] = {0.f, 0.f}, kl_raw[2] = {0.f, 0.f}, kl_dim[2] = {0.f, 0.f};
    for (int p = 0; p < a.nkl; ++p) {
      kl_mean[p] = (float)(t[2 + 2 * p] / a.B);
      kl_raw[p] = (float)(t[3 + 2 * p] / a.B);
      kl_dim[p] = (float)(t[2 + 2 * p] / (a.B * (double)a.D[p]));
    }
    float total;
    float beta[2] = {a.beta[0], a.beta[1]};
    if (a.beta_dev) {
      const float bd = a.beta_dev[0];
      if (a.beta_dev_mask & 1) beta[0] = bd;
      if (a.beta_dev_mask & 2) beta[1] = bd;
    }
    if (a.layout == 0) {
      float s = 0.f;
      s += 1.f * rec;
      s += beta[0] * kl_mean[0];
      s += a.temporal_weight * tmp;
      total = s;
      a.out[0] = total; a.out[1] = per_pixel; a.out[2] = rec; a.out[3] = tmp;
      a.out[4] = kl_mean[0]; a.out[5] = kl_raw[0]; a.out[6] = kl_dim[0];
    } else {
      float kl = 0.f;
      kl += beta[0] * kl_mean[0];
      kl += beta[1] * kl_mean[1];
      const float ind = a.indep ? a.indep[0] : 0.f;
      float s = 0.f;
      s += 1.f * rec;
      s += 1.f * kl;
      s += a.temporal_weight * tmp;
      s += a.indep_weight * ind;
      total = s;
      a.out[0] = total; a.out[1] = kl; a.out[2] = per_pixel; a.out[3] = rec; a.out[4] = tmp;
      a.out[5] = kl_mean[0]; a.out[6] = kl_raw[0]; a.out[7] = kl_dim[0];
      a.out[8] = kl_mean[1]; a.out[9] = kl_raw[1]; a.out[10] = kl_dim[1];
      a.out[11] = ind;
    }
    if (a.total_out) a.total_out[0] = total;
  }
}

extern "C" int dvae_vae_loss_fwd_f32(const float* recon, const float* target, int64_t B, int64_t T, int64_t P, int kind,
                                     int with_temporal, const float* mu0, const float* logvar0, int64_t D0,
                                     float beta0, const float* mu1, const float* logvar1, int64_t D1, float beta1,
                                     const float* beta_dev, int beta_dev_mask, float free_bits,
                                     float temporal_weight, const float* indep, float indep_weight, float* out,
                                     float* total_out, float* ws, dvae_stream_t stream) {
  DVAE_REQUIRE(B > 0 && T >= 1 && P > 0 && P % 4 == 0, "vae_loss_fwd: need P%%4==0");
  DVAE_REQUIRE(kind == 0 || kind == 1, "vae_loss_fwd: kind must be 0 (mse) or 1 (bce)");
  DVAE_REQUIRE(recon && target && mu0 && logvar0 && out && ws && D0 > 0, "vae_loss_fwd: null argument");
  DVAE_REQUIRE((mu1 == nullptr) == (logvar1 == nullptr) && (mu1 == nullptr || D1 > 0), "vae_loss_fwd: the second latent needs mu, logvar and its width");
  DVAE_REQUIRE(((uintptr_t)ws & 7) == 0, "vae_loss_fwd: workspace must be 8-byte aligned");
  DVAE_REQUIRE((beta_dev_mask & ~3) == 0 && (beta_dev_mask == 0 || beta_dev), "vae_loss_fwd: beta_dev_mask names latents 0 and 1 and needs beta_dev");
  const int nb = recon_blocks(B, P);
  DVAE_WORK(8.0 * (double)B * T * P, DVAE_UNIT_BYTE);                      // recon and target, once each
  DVAE_LAUNCH(recon_loss_fwd_kernel, dim3(nb), dim3(256), 0, ST(stream), (const float4*)recon,
                     (const float4*)target, (double*)ws, B * (P / 4), (int)T, P / 4, kind, with_temporal);
  DVAE_LAUNCH_CHECK("vae_loss_fwd");
  LossFinishArgs a;
  a.partials = (const double*)ws; a.nparts = nb;
  a.B = (double)B; a.T = (double)T; a.P = (double)P;
  a.mu[0] = mu0; a.lv[0] = logvar0; a.D[0] = D0; a.beta[0] = beta0;
  a.mu[1] = mu1; a.lv[1] = logvar1; a.D[1] = mu1 ? D1 : 0; a.beta[1] = beta1;
  a.beta_dev = beta_dev_mask ? beta_dev : nullptr; a.beta_dev_mask = beta_dev_mask;
  a.nkl = mu1 ? 2 : 1;
  a.free_bits = free_bits; a.temporal_weight = temporal_weight; a.indep_weight = indep_weight;
  a.indep = indep; a.layout = mu1 ? 1 : 0; a.out = out; a.total_out = total_out;
  DVAE_LAUNCH(vae_loss_finish_kernel, dim3(1), dim3(1024), 0, ST(stream), a);
  DVAE_LAUNCH_CHECK("vae_loss_finish");
  return 0;
}

// KL gradients of one or two latents in one launch (coef_i = beta_i; the 1/B of the batch mean is applied here)
__global__ void kl_bwd2_kernel(const float* mu0, const float* lv0, float* dmu0, float* dlv0, int64_t n0, float c0,
                               const float* mu1, const float* lv1, float* dmu1, float* dlv1, int64_t n1, float c1,
                               const float* coef_dev, int coef_dev_mask, float inv_b, const float* gscale,
                               float free_bits) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const float* mu = mu0; const float* lvp = lv0; float* dmu = dmu0; float* dlv = dlv0; float coef = c0;
  int which = 1;
  if (i >= n0) {
    i -= n0;
    if (i >= n1) return;
    mu = mu1; lvp = lv1; dmu = dmu1; dlv = dlv1; coef = c1;
    which = 2;
  }
  if (coef_dev_mask & which) coef = coef_dev[0];
  const float c = (coef * inv_b) * (gscale ? gscale[0] : 1.f);
  const float m = mu[i], lv = lvp[i];
  const float e = expf(lv);
  float gm = c * m, gl = c * 0.5f * (e - 1.f);
  if (free_bits > 0.f) {
    const float kl = -0.5f * (1.f + lv - m * m - e);
    if (!(kl > free_bits)) {  // the floor is active: maximum() passes no gradient to kl
      gm = 0.f;
      gl = 0.f;
    }
  }
  dmu[i] = gm;
  dlv[i] = gl;
}

extern "C" int dvae_kl_bwd2_f32(const float* mu0, const float* logvar0, float coef0, float* dmu0, float* dlogvar0,
                                int64_t D0, const float* mu1, const float* logvar1, float coef1, float* dmu1,
                                float* dlogvar1, int64_t D1, const float* coef_dev, int coef_dev_mask,
                                const float* gscale, int64_t B, float free_bits, dvae_stream_t stream) {
  DVAE_REQUIRE(B > 0 && D0 > 0 && mu0 && logvar0 && dmu0 && dlogvar0, "kl_bwd2: empty or null");
  DVAE_REQUIRE(mu1 == nullptr || (logvar1 && dmu1 && dlogvar1 && D1 > 0), "kl_bwd2: the second latent is incomplete");
  DVAE_REQUIRE((coef_dev_mask & ~3) == 0 && (coef_dev_mask == 0 || coef_dev), "kl_bwd2: coef_dev_mask names latents 0 and 1 and needs coef_dev");
  const int64_t n0 = B * D0, n1 = mu1 ? B * D1 : 0;
  DVAE_LAUNCH(kl_bwd2_kernel, dim3((unsigned)ceil_div64(n0 + n1, 256)), dim3(256), 0, ST(stream), mu0, logvar0, dmu0,
                     dlogvar0, n0, coef0, mu1, logvar1, dmu1, dlogvar1, n1, coef1, coef_dev, coef_dev_mask,
                     1.f / (float)B, gscale, free_bits);
  DVAE_LAUNCH_CHECK("kl_bwd2");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// independence term.  ws layout: corr[Dc*Dm] | mean_c[Dc] | inv_c[Dc] | mean_m[Dm] | inv_m[Dm] | part[Dc]
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
indep_stats_kernel(const float* __restrict__ z, float* __restrict__ mean, float* __restrict__ inv, int64_t B,
                   int64_t D) {
  // one workgroup per column
  __shared__ double red[4];
  const int64_t j = blockIdx.x;
  double s = 0.0;
  for (int64_t b = threadIdx.x; b < B; b += 256) s += (double)z[b * D + j];
  s = block_sum_256(s, red);
  __shared__ double mu_s;
  if (threadIdx.x == 0) mu_s = s / (double)B;
  __syncthreads();
  const double mu = mu_s;
  double q = 0.0;
  for (int64_t b = threadIdx.x; b < B; b += 256) {
    const double d = (double)z[b * D + j] - mu;
    q += d * d;
  }
  q = block_sum_256(q, red);
  if (threadIdx.x == 0) {
    const float sd = (float)sqrt(q / (double)(B > 1 ? B - 1 : 1));
    mean[j] = (float)mu;
    inv[j] = 1.f / (sd + 1e-8f);
  }
}

__global__ void __launch_bounds__(256)
indep_corr_kernel(const float* __restrict__ zc, const float* __restrict__ zm, float* __restrict__ ws, int64_t B,
                  int64_t Dc, int64_t Dm) {
  // one workgroup per content column i.  Thread (bs, jl): batch rows b = bs, bs+8, ... and motion columns
  // j = jl, jl+32, ... (a row of zm is read as contiguous 128-byte pieces); the eight row slices meet in LDS.
  __shared__ double red[4];
  __shared__ float partial[8][256 + 1];
  float* corr = ws;
  const float* mean_c = ws + Dc * Dm;
  const float* inv_c = mean_c + Dc;
  const float* mean_m = inv_c + Dc;
  const float* inv_m = mean_m + Dm;
  float* part = const_cast<float*>(inv_m) + Dm;
  const int64_t i = blockIdx.x;
  const float mc = mean_c[i], ic = inv_c[i];
  const int bs = threadIdx.x >> 5, jl = threadIdx.x & 31;
  constexpr int JK = 8;                      // columns per thread: Dm <= 256
  float mm[JK], im[JK], acc[JK];
#pragma unroll
  for (int k = 0; k < JK; ++k) {
    const int64_t j = jl + 32 * k;
    mm[k] = j < Dm ? mean_m[j] : 0.f;
    im[k] = j < Dm ? inv_m[j] : 0.f;
    acc[k] = 0.f;
  }
  for (int64_t b = bs; b < B; b += 8) {
    const float xc = (zc[b * Dc + i] - mc) * ic;
#pragma unroll
    for (int k = 0; k < JK; ++k) {
      const int64_t j = jl + 32 * k;
      if (j < Dm) acc[k] += xc * ((zm[b * Dm + j] - mm[k]) * im[k]);
    }
  }
#pragma unroll
  for (int k = 0; k < JK; ++k) partial[bs][jl + 32 * k] = acc[k];
  __syncthreads();
  double sq = 0.0;
  for (int64_t j = threadIdx.x; j < Dm; j += 256) {
    float a = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) a += partial[q][j];
    const float c = a / (float)B;
    corr[i * Dm + j] = c;
    sq += (double)c * (double)c;
  }
  sq = block_sum_256(sq, red);
  if (threadIdx.x == 0) part[i] = (float)sq;
}

__global__ void __launch_bounds__(256)
indep_finalize_kernel(const float* __restrict__ part, float* __restrict__ out, int64_t Dc, int64_t Dm) {
  __shared__ double red[4];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < Dc; i += 256) s += (double)part[i];
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) out[0] = (float)(s / (double)(Dc * Dm));
}

// gradient for one side.  X is the side being differentiated [B,Dx], Y the other [B,Dy];
// corr is indexed corr[ix*sx + iy*sy].
__global__ void __launch_bounds__(256)
indep_bwd_kernel(const float* __restrict__ X, const float* __restrict__ Y, const float* __restrict__ corr,
                 int64_t sx, int64_t sy, const float* __restrict__ mean_x, const float* __restrict__ inv_x,
                 const float* __restrict__ mean_y, const float* __restrict__ inv_y,
                 const float* __restrict__ gscale, float coef, float* __restrict__ dX, int64_t B, int64_t Dx,
                 int64_t Dy) {
  // one workgroup per column i of X.  Eight adjacent lanes share a batch row: each takes a contiguous eighth of
  // the other side's columns (its correlation coefficients and column statistics stay in registers), so a row
  // is read as one contiguous piece and the workgroup covers 32 rows per pass.
  extern __shared__ float gx_s[];  // B floats
  __shared__ double red[4];
  const int64_t i = blockIdx.x;
  const float g = coef * (gscale ? gscale[0] : 1.f);
  const float mx = mean_x[i], ix = inv_x[i];
  const float gfac = 2.f / (float)(Dx * Dy) / (float)B;
  constexpr int JMAX = 32;                   // columns per lane: Dy <= 256
  const int js = threadIdx.x & 7, rb = threadIdx.x >> 3;
  const int per = (int)((Dy + 7) / 8), j0 = js * per;
  float cj[JMAX], myj[JMAX];
#pragma unroll
  for (int u = 0; u < JMAX; ++u) {
    const int j = j0 + u;
    const bool ok = u < per && j < Dy;
    cj[u] = ok ? corr[i * sx + j * sy] * inv_y[j] : 0.f;      // coefficient times 1/std of column j
    myj[u] = ok ? mean_y[j] : 0.f;
  }
  double s1 = 0.0, s2 = 0.0;
  for (int64_t b0 = 0; b0 < B; b0 += 32) {
    const int64_t b = b0 + rb;
    float acc = 0.f;
    if (b < B) {
#pragma unroll
      for (int u = 0; u < JMAX; ++u)
        if (u < per && j0 + u < Dy) acc += cj[u] * (Y[b * Dy + j0 + u] - myj[u]);
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) acc += __shfl_xor(acc, o);
    if (js == 0 && b < B) {
      const float gx = acc * gfac;  // dL/d xtilde_bi
      gx_s[b] = gx;
      s1 += (double)gx;
      s2 += (double)gx * (double)(X[b * Dx + i] - mx);
    }
  }
  s1 = block_sum_256(s1, red);
  __shared__ double s1_s, s2_s;
  if (threadIdx.x == 0) s1_s = s1;
  s2 = block_sum_256(s2, red);
  if (threadIdx.x == 0) s2_s = s2;
  __syncthreads();
  const float gbar = (float)(s1_s / (double)B);
  // std s = 1/ix - 1e-8 ; d s / d x_b = (x_b - mu) / ((B-1) s)
  const float sd = 1.f / ix - 1e-8f;
  const float k2 = (B > 1 && sd > 0.f) ? (float)s2_s * ix * ix / ((float)(B - 1) * sd) : 0.f;
  for (int64_t b = threadIdx.x; b < B; b += 256) {
    const float v = (gx_s[b] - gbar) * ix - (X[b * Dx + i] - mx) * k2;
    dX[b * Dx + i] = g * v;
  }
}

extern "C" int64_t dvae_indep_ws_floats(int64_t Dc, int64_t Dm) { return Dc * Dm + 3 * Dc + 2 * Dm; }

extern "C" int dvae_indep_fwd_f32(const float* zc, const float* zm, float* out, float* ws, int64_t B, int64_t Dc,
                                  int64_t Dm, dvae_stream_t stream) {
  DVAE_REQUIRE(B > 0 && Dc > 0 && Dm > 0, "indep_fwd: empty");
  DVAE_REQUIRE(Dm <= 256, "indep_fwd: motion latent sizes above 256 are not supported (Dm=%ld)", (long)Dm);
  float* mean_c = ws + Dc * Dm;
  float* inv_c = mean_c + Dc;
  float* mean_m = inv_c + Dc;
  float* inv_m = mean_m + Dm;
  float* part = inv_m + Dm;
  DVAE_LAUNCH(indep_stats_kernel, dim3((unsigned)Dc), dim3(256), 0, ST(stream), zc, mean_c, inv_c, B, Dc);
  DVAE_LAUNCH(indep_stats_kernel, dim3((unsigned)Dm), dim3(256), 0, ST(stream), zm, mean_m, inv_m, B, Dm);
  DVAE_LAUNCH(indep_corr_kernel, dim3((unsigned)Dc), dim3(256), 0, ST(stream), zc, zm, ws, B, Dc, Dm);
  DVAE_LAUNCH(indep_finalize_kernel, dim3(1), dim3(256), 0, ST(stream), part, out, Dc, Dm);
  DVAE_LAUNCH_CHECK("indep_fwd");
  return 0;
}

extern "C" int dvae_indep_bwd_f32(const float* zc, const float* zm, const float* ws, const float* gscale, float coef,
                                  float* dzc, float* dzm, int64_t B, int64_t Dc, int64_t Dm, dvae_stream_t stream) {
  DVAE_REQUIRE(B > 0 && Dc > 0 && Dm > 0, "indep_bwd: empty");
  DVAE_REQUIRE(B * 4 <= 64 * 1024, "indep_bwd: batch %ld too large for the LDS staging buffer", (long)B);
  DVAE_REQUIRE(Dc <= 256 && Dm <= 256, "indep_bwd: latent sizes above 256 are not supported (Dc=%ld Dm=%ld)", (long)Dc,
               (long)Dm);
  const float* corr = ws;
  const float* mean_c = ws + Dc * Dm;
  const float* inv_c = mean_c + Dc;
  const float* mean_m = inv_c + Dc;
  const float* inv_m = mean_m + Dm;
  DVAE_LAUNCH(indep_bwd_kernel, dim3((unsigned)Dc), dim3(256), (size_t)B * 4, ST(stream), zc, zm, corr, Dm,
                     (int64_t)1, mean_c, inv_c, mean_m, inv_m, gscale, coef, dzc, B, Dc, Dm);
  DVAE_LAUNCH(indep_bwd_kernel, dim3((unsigned)Dm), dim3(256), (size_t)B * 4, ST(stream), zm, zc, corr,
                     (int64_t)1, Dm, mean_m, inv_m, mean_c, inv_c, gscale, coef, dzm, B, Dm, Dc);
  DVAE_LAUNCH_CHECK("indep_bwd");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// BatchNorm glue
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
bn_finalize_kernel(const float* __restrict__ partials, int64_t nparts, double M, double Mrep, int64_t C,
                   const float* __restrict__ gamma, const float* __restrict__ beta, float* running_mean,
                   float* running_var, int64_t* nbt, float momentum, float eps, float* scale, float* shift,
                   float* mean_out, float* invstd_out) {
  __shared__ double red[4];
  const int64_t c = blockIdx.x;
  double s1 = 0.0, s2 = 0.0;
  for (int64_t p = threadIdx.x; p < nparts; p += 256) {
    s1 += (double)partials[(p * 2) * C + c];
    s2 += (double)partials[(p * 2 + 1) * C + c];
  }
  s1 = block_sum_256(s1, red);
  s2 = block_sum_256(s2, red);
  if (threadIdx.x == 0) {
    const double mean = s1 / M;
    double var = s2 / M - mean * mean;
    if (var < 0.0) var = 0.0;
    const double invstd = 1.0 / sqrt(var + (double)eps);
    const float sc = (float)((double)gamma[c] * invstd);
    scale[c] = sc;
    shift[c] = (float)((double)beta[c] - mean * (double)gamma[c] * invstd);
    mean_out[c] = (float)mean;
    invstd_out[c] = (float)invstd;
    if (running_mean) {
      const double unb = Mrep > 1.0 ? var * (Mrep / (Mrep - 1.0)) : var;
      running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * mean);
      running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unb);
    }
    if (nbt && c == 0) nbt[0] += 1;
  }
}

extern "C" int dvae_bn_finalize_rep_f32(const float* partials, int64_t nparts, int64_t M, int64_t replicas, int64_t C,
                                        const float* gamma, const float* beta, float* running_mean,
                                        float* running_var, int64_t* num_batches_tracked, float momentum, float eps,
                                        float* scale, float* shift, float* mean, float* invstd, dvae_stream_t stream) {
  DVAE_REQUIRE(nparts > 0 && M > 0 && C > 0 && replicas >= 1, "bn_finalize: empty");
  DVAE_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_finalize: running buffers come in pairs");
  DVAE_LAUNCH(bn_finalize_kernel, dim3((unsigned)C), dim3(256), 0, ST(stream), partials, nparts, (double)M,
                     (double)M * (double)replicas, C, gamma, beta, running_mean, running_var, num_batches_tracked,
                     momentum, eps, scale, shift, mean, invstd);
  DVAE_LAUNCH_CHECK("bn_finalize");
  return 0;
}

extern "C" int dvae_bn_finalize_f32(const float* partials, int64_t nparts, int64_t M, int64_t C, const float* gamma,
                                    const float* beta, float* running_mean, float* running_var,
                                    int64_t* num_batches_tracked, float momentum, float eps, float* scale,
                                    float* shift, float* mean, float* invstd, dvae_stream_t stream) {
  return dvae_bn_finalize_rep_f32(partials, nparts, M, 1, C, gamma, beta, running_mean, running_var,
                                  num_batches_tracked, momentum, eps, scale, shift, mean, invstd, stream);
}

__global__ void bn_eval_affine_kernel(const float* gamma, const float* beta, const float* rm, const float* rv,
                                      float eps, float* scale, float* shift, int64_t C) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float inv = 1.f / sqrtf(rv[c] + eps);
  const float sc = gamma[c] * inv;
  scale[c] = sc;
  shift[c] = beta[c] - rm[c] * sc;
}

extern "C" int dvae_bn_eval_affine_f32(const float* gamma, const float* beta, const float* running_mean,
                                       const float* running_var, float eps, float* scale, float* shift, int64_t C,
                                       dvae_stream_t stream) {
  DVAE_REQUIRE(C > 0, "bn_eval_affine: empty");
  DVAE_LAUNCH(bn_eval_affine_kernel, dim3((unsigned)ceil_div64(C, 256)), dim3(256), 0, ST(stream), gamma, beta,
                     running_mean, running_var, eps, scale, shift, C);
  DVAE_LAUNCH_CHECK("bn_eval_affine");
  return 0;
}

__global__ void __launch_bounds__(256)
affine_act_kernel(const float4* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                  int act, float4* __restrict__ out, int64_t total4, int C4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C4) * 4;
    out[i] = affine_act4(x[i], scale, shift, c, act);
  }
}

extern "C" int dvae_affine_act_f32(const float* x, const float* scale, const float* shift, int act, float* out,
                                   int64_t M, int64_t C, dvae_stream_t stream) {
  DVAE_REQUIRE(M > 0 && C > 0 && C % 4 == 0, "affine_act: need C%%4==0");
  const int64_t total4 = M * C / 4;
  int64_t nb = ceil_div64(total4, 256);
  if (nb > 4096) nb = 4096;
  DVAE_WORK(8.0 * (double)M * C, DVAE_UNIT_BYTE);
  DVAE_LAUNCH(affine_act_kernel, dim3((unsigned)nb), dim3(256), 0, ST(stream), (const float4*)x, scale, shift,
                     act, (float4*)out, total4, (int)(C / 4));
  DVAE_LAUNCH_CHECK("affine_act");
  return 0;
}

// ---- column sums over [M, C] with an element transform ------------------------------------------
// Thread layout: TX column groups (4 columns each) x TY rows; grid.y tiles the column groups,
// grid.x strides the rows.  Each workgroup writes one row of partials[grid.x][NV][C].
#define CS_MAX_ROWBLOCKS 1024

struct ColsumGeom {
  int tx, ty, gy;
  int64_t gx;
};

static ColsumGeom colsum_geom(int64_t M, int64_t C) {
  ColsumGeom g;
  const int64_t c4 = C / 4;
  int tx = 1;
  while (tx < c4 && tx < 256) tx <<= 1;
  g.tx = tx;
  g.ty = 256 / tx;
  g.gy = (int)ceil_div64(c4, tx);
  int64_t gx = ceil_div64(M, (int64_t)g.ty * 8);
  if (gx > CS_MAX_ROWBLOCKS) gx = CS_MAX_ROWBLOCKS;
  if (gx < 1) gx = 1;
  g.gx = gx;
  return g;
}

// Geometry of the one-launch form (partial rows AND their sum in the same launch: the workgroup of a column group
// that finishes last adds the group's partial rows): column groups of at most 128 columns, so that the adding-up
// is a few tens of KB per group and spread over as many workgroups as there are groups, and at most 1024
// workgroups in all.
static ColsumGeom colsum_geom_fused(int64_t M, int64_t C) {
  ColsumGeom g;
  const int64_t c4 = C / 4;
  int tx = 1;
  while (tx < c4 && tx < 32) tx <<= 1;
  g.tx = tx;
  g.ty = 256 / tx;
  g.gy = (int)ceil_div64(c4, tx);
  int64_t gx = ceil_div64(M, (int64_t)g.ty * 8);
  int64_t cap = CS_MAX_ROWBLOCKS / g.gy;
  if (cap < 1) cap = 1;
  if (gx > cap) gx = cap;
  if (gx < 1) gx = 1;
  g.gx = gx;
  return g;
}
#define CS_MAX_GROUPS 1024     // column groups of one launch = arrival counters it uses

extern "C" int64_t dvae_colsum_ws_floats(int64_t M, int64_t C) {
  (void)M;
  return 2 * (int64_t)CS_MAX_ROWBLOCKS * C + 2 * C;       // up to 1024 partial rows of two values per column
}

extern "C" int64_t dvae_colsum_tickets(void) { return CS_MAX_GROUPS; }

// MODE 0: BN backward stats: v1 = dz, v2 = dz * xhat   (NV = 2)
// MODE 1: bias/activation backward: dpre = dout*act'(out) written to dst (which may be du itself), v1 = dpre (NV = 1)
template <int MODE>
__global__ void __launch_bounds__(256)
colsum_kernel(const float* du, const float* __restrict__ y, const float* __restrict__ scale,
              const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ invstd,
              int act, float* __restrict__ partials, int64_t M, int64_t C, int tx, int ty, float* dst,
              unsigned* tickets, int fin, float invM, float* out0, float* out1, float* out2, float* out3) {
  // tickets == nullptr: the partial rows are the result.  Otherwise one launch does it all: fin 0 writes the column
  // sums to out0 (and out1 for the second value); fin 1 (MODE 0) writes the BatchNorm-backward sums and the two
  // coefficient vectors of bn_bwd_coeffs_kernel: out0 = dbeta, out1 = dgamma, out2 = A, out3 = B.
  constexpr int NV = MODE == 0 ? 2 : 1;
  __shared__ float red[256 * 4 * NV];
  __shared__ int s_last;
  const int cx = threadIdx.x % tx, ry = threadIdx.x / tx;
  const int64_t c4 = (int64_t)blockIdx.y * tx + cx;
  const bool col_ok = c4 * 4 < C;
  const int64_t c = c4 * 4;
  float a1[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f};
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f), mu = sh, is = sc;
  if (MODE == 0 && col_ok) {
    sc = *reinterpret_cast<const float4*>(scale + c);
    sh = *reinterpret_cast<const float4*>(shift + c);
    mu = *reinterpret_cast<const float4*>(mean + c);
    is = *reinterpret_cast<const float4*>(invstd + c);
  }
  if (col_ok) {
    // one row's contribution; rows are added in ascending order whatever the unrolling, so the sums do not depend on it
    auto add_row = [&](int64_t r, float4 g, const float4 yv) {
      if (MODE == 0) {
        const float z0 = fmaf(yv.x, sc.x, sh.x), z1 = fmaf(yv.y, sc.y, sh.y), z2 = fmaf(yv.z, sc.z, sh.z),
                    z3 = fmaf(yv.w, sc.w, sh.w);
        const float d0 = z0 > 0.f ? g.x : g.x * LRELU_SLOPE, d1 = z1 > 0.f ? g.y : g.y * LRELU_SLOPE,
                    d2 = z2 > 0.f ? g.z : g.z * LRELU_SLOPE, d3 = z3 > 0.f ? g.w : g.w * LRELU_SLOPE;
        a1[0] += d0; a1[1] += d1; a1[2] += d2; a1[3] += d3;
        a2[0] += d0 * ((yv.x - mu.x) * is.x);
        a2[1] += d1 * ((yv.y - mu.y) * is.y);
        a2[2] += d2 * ((yv.z - mu.z) * is.z);
        a2[3] += d3 * ((yv.w - mu.w) * is.w);
      } else {
        if (act == DVAE_ACT_LRELU) {
          g.x = yv.x > 0.f ? g.x : g.x * LRELU_SLOPE;
          g.y = yv.y > 0.f ? g.y : g.y * LRELU_SLOPE;
          g.z = yv.z > 0.f ? g.z : g.z * LRELU_SLOPE;
          g.w = yv.w > 0.f ? g.w : g.w * LRELU_SLOPE;
        }
        if (act == DVAE_ACT_LRELU || dst != du) *reinterpret_cast<float4*>(dst + r * C + c) = g;
        a1[0] += g.x; a1[1] += g.y; a1[2] += g.z; a1[3] += g.w;
      }
    };
    const bool need_y = MODE == 0 || act == DVAE_ACT_LRELU;
    const int64_t step = (int64_t)gridDim.x * ty;
    int64_t r = (int64_t)blockIdx.x * ty + ry;
    // four rows' loads in flight per thread: with one, the loop runs at the latency of a load, not at bandwidth
    for (; r + 3 * step < M; r += 4 * step) {
      float4 g[4], yv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        g[u] = *reinterpret_cast<const float4*>(du + (r + u * step) * C + c);
        yv[u] = need_y ? *reinterpret_cast<const float4*>(y + (r + u * step) * C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) add_row(r + u * step, g[u], yv[u]);
    }
    for (; r < M; r += step) {
      const float4 g = *reinterpret_cast<const float4*>(du + r * C + c);
      const float4 yv = need_y ? *reinterpret_cast<const float4*>(y + r * C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      add_row(r, g, yv);
    }
  }
  // reduce over ry
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    red[(threadIdx.x * NV + 0) * 4 + k] = a1[k];
    if (NV == 2) red[(threadIdx.x * NV + 1) * 4 + k] = a2[k];
  }
  __syncthreads();
  // the partial rows live behind a buffer descriptor: in the one-launch form they are handed to another workgroup
  // inside the launch, so they leave as 16-byte write-through stores and are read back past the L1 (cache policy
  // bit 4 of a raw buffer access); a plain descriptor store otherwise
  const auto prsrc = __builtin_amdgcn_make_buffer_rsrc(partials, 0, 0x7fffffff, 0x00020000);
  const int aux = tickets ? 16 : 0;
  if (ry == 0 && col_ok) {
    for (int v = 0; v < NV; ++v) {
      f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int j = 0; j < ty; ++j) {
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] += red[(((j * tx) + cx) * NV + v) * 4 + k];
      }
      const unsigned off = (unsigned)((((int64_t)blockIdx.x * NV + v) * C + c) * 4);
      if (aux) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, s), prsrc, off, 0, 16);
      else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, s), prsrc, off, 0, 0);
    }
  }
  if (tickets == nullptr) return;
  // every storing wave drains its stores, the workgroup meets, one lane draws the group's ticket: the workgroup that
  // draws the last one adds up the group's rows (in row order, whoever it is: the result does not depend on it)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(tickets + blockIdx.y, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = old + 1u == gridDim.x;
  }
  __syncthreads();
  if (!s_last) return;
  double acc[NV][4];
#pragma unroll
  for (int v = 0; v < NV; ++v)
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[v][k] = 0.0;
  if (col_ok) {
    for (int64_t p = ry; p < (int64_t)gridDim.x; p += ty) {
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const unsigned off = (unsigned)(((p * NV + v) * C + c) * 4);
        const f32x4 q = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prsrc, off, 0, 16));
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[v][k] += (double)q[k];
      }
    }
  }
  double* dred = reinterpret_cast<double*>(red);           // 256 * 4 * NV floats = room for 256 * 2 * NV doubles
#pragma unroll
  for (int v = 0; v < NV; ++v) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      __syncthreads();
      dred[threadIdx.x * 2 + 0] = acc[v][2 * half];
      dred[threadIdx.x * 2 + 1] = acc[v][2 * half + 1];
      __syncthreads();
      if (ry == 0) {
        double s0 = 0.0, s1 = 0.0;
        for (int j = 0; j < ty; ++j) {
          s0 += dred[(j * tx + cx) * 2 + 0];
          s1 += dred[(j * tx + cx) * 2 + 1];
        }
        acc[v][2 * half] = s0;
        acc[v][2 * half + 1] = s1;
      }
    }
  }
  if (ry == 0 && col_ok) {
    if (fin == 0) {
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        float* o = v == 0 ? out0 : out1;
        if (o) *reinterpret_cast<float4*>(o + c) = make_float4((float)acc[v][0], (float)acc[v][1], (float)acc[v][2], (float)acc[v][3]);
      }
    } else if (NV == 2) {
      const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, muv[4] = {mu.x, mu.y, mu.z, mu.w}, isv[4] = {is.x, is.y, is.z, is.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float db = (float)acc[0][k], dg = (float)acc[NV - 1][k];
        out0[c + k] = db;
        out1[c + k] = dg;
        const float a = -scv[k] * isv[k] * (dg * invM);
        out2[c + k] = a;
        out3[c + k] = -scv[k] * (db * invM) - a * muv[k];
      }
    }
  }
  // the counter is back at zero for the next launch that uses it (launches that share counters are ordered on one stream)
  if (threadIdx.x == 0) __hip_atomic_store(tickets + blockIdx.y, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// sums partials[nparts][NV][C] -> out_v[C]; optional BN-backward constants
__global__ void __launch_bounds__(256)
colsum_finalize_kernel(const float* __restrict__ partials, int64_t nparts, int NV, int64_t C, float* out0,
                       float* out1) {
  __shared__ double red[4];
  const int64_t c = blockIdx.x;
  for (int v = 0; v < NV; ++v) {
    double s = 0.0;
    for (int64_t p = threadIdx.x; p < nparts; p += 256) s += (double)partials[(p * NV + v) * C + c];
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) {
      float* o = v == 0 ? out0 : out1;
      if (o) o[c] = (float)s;
    }
  }
}

// Batch sums of a BatchNorm backward and the two vectors that fold them into the elementwise step:
// dx = scale * (dz - dbeta/M - xhat * dgamma/M) = scale * dz + A * y + B with
// A = -scale * invstd * dgamma / M and B = -scale * dbeta / M - A * mean.
__global__ void __launch_bounds__(256)
bn_bwd_coeffs_kernel(const float* __restrict__ partials, int64_t nparts, int64_t C, const float* __restrict__ scale,
                     const float* __restrict__ mean, const float* __restrict__ invstd, float invM,
                     float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ coef_a,
                     float* __restrict__ coef_b) {
  __shared__ double red[4];
  const int64_t c = blockIdx.x;
  double s1 = 0.0, s2 = 0.0;
  for (int64_t p = threadIdx.x; p < nparts; p += 256) {
    s1 += (double)partials[(p * 2 + 0) * C + c];
    s2 += (double)partials[(p * 2 + 1) * C + c];
  }
  s1 = block_sum_256(s1, red);
  s2 = block_sum_256(s2, red);
  if (threadIdx.x == 0) {
    const float db = (float)s1, dg = (float)s2;
    dbeta[c] = db;
    dgamma[c] = dg;
    const float a = -scale[c] * invstd[c] * (dg * invM);
    coef_a[c] = a;
    coef_b[c] = -scale[c] * (db * invM) - a * mean[c];
  }
}

extern "C" int dvae_bn_bwd_coeffs_f32(const float* partials, int64_t nparts, const float* scale, const float* mean,
                                      const float* invstd, float* dgamma, float* dbeta, float* coef_a, float* coef_b,
                                      int64_t M, int64_t C, dvae_stream_t stream) {
  DVAE_REQUIRE(M > 0 && C > 0 && nparts > 0, "bn_bwd_coeffs: empty");
  DVAE_REQUIRE(partials && scale && mean && invstd && dgamma && dbeta && coef_a && coef_b, "bn_bwd_coeffs: null argument");
  DVAE_LAUNCH(bn_bwd_coeffs_kernel, dim3((unsigned)C), dim3(256), 0, ST(stream), partials, nparts, C, scale, mean,
                     invstd, 1.f / (float)M, dgamma, dbeta, coef_a, coef_b);
  DVAE_LAUNCH_CHECK("bn_bwd_coeffs");
  return 0;
}

// dx = scale * (dz - dbeta/M - xhat * dgamma/M), dz = du * lrelu'(scale*y + shift), in place over du.
// The launch makes the grid stride a multiple of C/4, so a thread meets the same four channels in
// every iteration and keeps their six constants in registers.
template <bool ADJ>
__global__ void __launch_bounds__(256)
bn_bwd_apply_kernel(float4* __restrict__ du, const float4* __restrict__ y, const float* __restrict__ scale,
                    const float* __restrict__ shift, const float* __restrict__ mean,
                    const float* __restrict__ invstd, const float* __restrict__ dbeta,
                    const float* __restrict__ dgamma, float invM, int64_t total4, int C4) {
  const int64_t first = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
  const int c = (int)(first % C4) * 4;
  const float4 sc = *reinterpret_cast<const float4*>(scale + c);
  const float4 sh = *reinterpret_cast<const float4*>(shift + c);
  const float4 mu = *reinterpret_cast<const float4*>(mean + c);
  const float4 is = *reinterpret_cast<const float4*>(invstd + c);
  float4 k1 = *reinterpret_cast<const float4*>(dbeta + c);      // dbeta/M
  float4 k2 = *reinterpret_cast<const float4*>(dgamma + c);     // dgamma/M
  k1.x *= invM; k1.y *= invM; k1.z *= invM; k1.w *= invM;
  k2.x *= invM; k2.y *= invM; k2.z *= invM; k2.w *= invM;
#define BN_ONE(f)                                                         \
  {                                                                       \
    const float z = fmaf(yv.f, sc.f, sh.f);                               \
    const float dz = z > 0.f ? g.f : g.f * LRELU_SLOPE;                   \
    const float xh = (yv.f - mu.f) * is.f;                                \
    o.f = sc.f * (dz - k1.f - xh * k2.f);                                 \
  }
  int64_t i = first;
  if (ADJ) {
    // a workgroup takes four adjacent 4 KB pieces per round (256 % C4 == 0 keeps a thread on its channels)
    int64_t j = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    for (; j + 768 < total4; j += 4 * stride) {
      float4 gs[4], ys[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        gs[u] = du[j + u * 256];
        ys[u] = y[j + u * 256];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float4 g = gs[u], yv = ys[u];
        float4 o;
        BN_ONE(x) BN_ONE(y) BN_ONE(z) BN_ONE(w)
        du[j + u * 256] = o;
      }
    }
    // the ragged last round of the one workgroup that meets the end
    for (int u = 0; u < 4; ++u)
      if (j + u * 256 < total4) {
        const float4 g = du[j + u * 256];
        const float4 yv = y[j + u * 256];
        float4 o;
        BN_ONE(x) BN_ONE(y) BN_ONE(z) BN_ONE(w)
        du[j + u * 256] = o;
      }
    i = total4;
  }
  // four elements' loads in flight per thread before the first is used
  for (; i + 3 * stride < total4; i += 4 * stride) {
    float4 gs[4], ys[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      gs[u] = du[i + u * stride];
      ys[u] = y[i + u * stride];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float4 g = gs[u], yv = ys[u];
      float4 o;
      BN_ONE(x) BN_ONE(y) BN_ONE(z) BN_ONE(w)
      du[i + u * stride] = o;
    }
  }
  for (; i < total4; i += stride) {
    const float4 g = du[i];
    const float4 yv = y[i];
    float4 o;
    BN_ONE(x) BN_ONE(y) BN_ONE(z) BN_ONE(w)
    du[i] = o;
  }
#undef BN_ONE
}

// grid of the apply kernel: at most 4096 workgroups, and 256 * gridDim.x a multiple of C/4
static unsigned bn_apply_blocks(int64_t total4, int64_t C4) {
  int64_t nb = ceil_div64(total4, 256);
  if (nb > 4096) nb = 4096;
  int64_t unit = C4;                     // smallest nb with (256 nb) % C4 == 0 is C4 / gcd(256, C4)
  for (int64_t a = 256, b = C4; b != 0;) {
    const int64_t t = a % b;
    a = b;
    b = t;
    unit = C4 / a;
  }
  nb = ceil_div64(nb, unit) * unit;
  return (unsigned)nb;
}

// adjacent pieces per workgroup need a thread to stay on its four channels from one 256-element piece to the next
static bool bn_apply_adj(int64_t C4) { return 256 % C4 == 0; }
#define DVAE_LAUNCH_BN_APPLY(nb, ...)                                                         \
  if (bn_apply_adj(C / 4)) DVAE_LAUNCH(bn_bwd_apply_kernel<true>, dim3(nb), __VA_ARGS__);   \
  else DVAE_LAUNCH(bn_bwd_apply_kernel<false>, dim3(nb), __VA_ARGS__)

#define CS_NOFIN (unsigned*)nullptr, 0, 0.f, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr

static bool colsum_one_launch(const unsigned* tickets, int64_t C) {
  return tickets != nullptr && ceil_div64(C / 4, 32) <= CS_MAX_GROUPS && (((uintptr_t)tickets) & 3) == 0;
}

extern "C" int dvae_bn_bwd_f32(float* du, const float* y, const float* scale, const float* shift, const float* mean,
                               const float* invstd, const float* gamma, float* dgamma, float* dbeta, float* ws,
                               unsigned* tickets, int64_t M, int64_t C, dvae_stream_t stream) {
  (void)gamma;  // scale already equals gamma*invstd
  DVAE_REQUIRE(M > 0 && C > 0 && C % 4 == 0, "bn_bwd: need C%%4==0 (C=%ld)", (long)C);
  DVAE_REQUIRE(aligned16(ws) && aligned16(dbeta) && aligned16(dgamma), "bn_bwd: ws, dbeta, dgamma must be 16-byte aligned");
  DVAE_WORK(8.0 * (double)M * C, DVAE_UNIT_BYTE);                          // du and y, once each
  if (colsum_one_launch(tickets, C)) {
    const ColsumGeom g = colsum_geom_fused(M, C);
    DVAE_LAUNCH(colsum_kernel<0>, dim3((unsigned)g.gx, g.gy), dim3(256), 0, ST(stream), (const float*)du, y, scale, shift,
                       mean, invstd, 0, ws, M, C, g.tx, g.ty, (float*)nullptr, tickets, 0, 0.f, dbeta, dgamma,
                       (float*)nullptr, (float*)nullptr);
    DVAE_LAUNCH_CHECK("bn_bwd_reduce");
  } else {
    const ColsumGeom g = colsum_geom(M, C);
    DVAE_LAUNCH(colsum_kernel<0>, dim3((unsigned)g.gx, g.gy), dim3(256), 0, ST(stream), (const float*)du, y, scale, shift,
                       mean, invstd, 0, ws, M, C, g.tx, g.ty, (float*)nullptr, CS_NOFIN);
    DVAE_LAUNCH_CHECK("bn_bwd_reduce");
    DVAE_LAUNCH(colsum_finalize_kernel, dim3((unsigned)C), dim3(256), 0, ST(stream), (const float*)ws, g.gx, 2, C,
                       dbeta, dgamma);
    DVAE_LAUNCH_CHECK("bn_bwd_finalize");
  }
  const int64_t total4 = M * C / 4;
  DVAE_WORK(12.0 * (double)M * C, DVAE_UNIT_BYTE);                         // du, y in; dy out
  DVAE_LAUNCH_BN_APPLY(bn_apply_blocks(total4, C / 4), dim3(256), 0, ST(stream), (float4*)du, (const float4*)y,
                     scale, shift, mean, invstd, (const float*)dbeta, (const float*)dgamma, 1.f / (float)M, total4,
                     (int)(C / 4));
  DVAE_LAUNCH_CHECK("bn_bwd_apply");
  return 0;
}

// step 1 alone: the partial rows [rows][2][C] of (sum dz, sum dz*xhat), for dvae_bn_bwd_coeffs_f32
extern "C" int64_t dvae_bn_bwd_reduce_rows(int64_t M, int64_t C) { return colsum_geom(M, C).gx; }

extern "C" int dvae_bn_bwd_reduce_f32(const float* du, const float* y, const float* scale, const float* shift,
                                      const float* mean, const float* invstd, float* partials, int64_t M, int64_t C,
                                      dvae_stream_t stream) {
  DVAE_REQUIRE(M > 0 && C > 0 && C % 4 == 0, "bn_bwd_reduce: need C%%4==0 (C=%ld)", (long)C);
  const ColsumGeom g = colsum_geom(M, C);
  DVAE_WORK(8.0 * (double)M * C, DVAE_UNIT_BYTE);                          // du and y, once each
  DVAE_LAUNCH(colsum_kernel<0>, dim3((unsigned)g.gx, g.gy), dim3(256), 0, ST(stream), du, y,
                     scale, shift, mean, invstd, 0, partials, M, C, g.tx, g.ty, (float*)nullptr, CS_NOFIN);
  DVAE_LAUNCH_CHECK("bn_bwd_reduce");
  return 0;
}

// dvae_bn_bwd_reduce_f32 + dvae_bn_bwd_coeffs_f32 in one launch
extern "C" int dvae_bn_bwd_reduce_coeffs_f32(const float* du, const float* y, const float* scale, const float* shift,
                                             const float* mean, const float* invstd, float* dgamma, float* dbeta,
                                             float* coef_a, float* coef_b, float* ws, unsigned* tickets, int64_t M,
                                             int64_t C, dvae_stream_t stream) {
  DVAE_REQUIRE(M > 0 && C > 0 && C % 4 == 0, "bn_bwd_reduce_coeffs: need C%%4==0 (C=%ld)", (long)C);
  DVAE_REQUIRE(du && y && scale && shift && mean && invstd && dgamma && dbeta && coef_a && coef_b && ws,
               "bn_bwd_reduce_coeffs: null argument");
  DVAE_REQUIRE(colsum_one_launch(tickets, C) && aligned16(ws), "bn_bwd_reduce_coeffs: needs the arrival counters (dvae_colsum_tickets() "
               "zeroed 32-bit words) and a 16-byte aligned workspace");
  const ColsumGeom g = colsum_geom_fused(M, C);
  DVAE_WORK(8.0 * (double)M * C, DVAE_UNIT_BYTE);                          // du and y, once each
  DVAE_LAUNCH(colsum_kernel<0>, dim3((unsigned)g.gx, g.gy), dim3(256), 0, ST(stream), du, y, scale, shift, mean, invstd,
                     0, ws, M, C, g.tx, g.ty, (float*)nullptr, tickets, 1, 1.f / (float)M, dbeta, dgamma, coef_a, coef_b);
  DVAE_LAUNCH_CHECK("bn_bwd_reduce_coeffs");
  return 0;
}

extern "C" int dvae_bn_bwd_from_partials_f32(float* du, const float* y, const float* scale, const float* shift,
                                             const float* mean, const float* invstd, const float* partials,
                                             int64_t nparts, float* dgamma, float* dbeta, int64_t M, int64_t C,
                                             dvae_stream_t stream) {
  DVAE_REQUIRE(M > 0 && C > 0 && C % 4 == 0 && nparts > 0, "bn_bwd_from_partials: need C%%4==0 and partial rows (C=%ld)",
               (long)C);
  DVAE_LAUNCH(colsum_finalize_kernel, dim3((unsigned)C), dim3(256), 0, ST(stream), partials, nparts, 2, C, dbeta,
                     dgamma);
  DVAE_LAUNCH_CHECK("bn_bwd_finalize");
  const int64_t total4 = M * C / 4;
  DVAE_WORK(12.0 * (double)M * C, DVAE_UNIT_BYTE);                         // du, y in; dy out
  DVAE_LAUNCH_BN_APPLY(bn_apply_blocks(total4, C / 4), dim3(256), 0, ST(stream), (float4*)du, (const float4*)y,
                     scale, shift, mean, invstd, (const float*)dbeta, (const float*)dgamma, 1.f / (float)M, total4,
                     (int)(C / 4));
  DVAE_LAUNCH_CHECK("bn_bwd_apply");
  return 0;
}

extern "C" int dvae_bias_act_bwd_to_f32(const float* dout, float* dpre, const float* out, int act, float* dbias,
                                        float* ws, unsigned* tickets, int64_t M, int64_t N, dvae_stream_t stream) {
  DVAE_REQUIRE(M > 0 && N > 0 && N % 4 == 0, "bias_act_bwd: need N%%4==0 (N=%ld)", (long)N);
  DVAE_REQUIRE(act == DVAE_ACT_NONE || act == DVAE_ACT_LRELU, "bias_act_bwd: unsupported activation %d", act);
  DVAE_REQUIRE(dout && dpre, "bias_act_bwd: null gradient");
  if (act == DVAE_ACT_NONE && dbias == nullptr && dpre == dout) return 0;
  DVAE_WORK((act == DVAE_ACT_LRELU ? 12.0 : dpre != dout ? 8.0 : 4.0) * (double)M * N, DVAE_UNIT_BYTE);
  if (M <= 64) {
    // a handful of rows (latent heads at small batch): one launch; beyond that the two-stage form is faster.  Narrow column groups (at most 32 x 4 columns per workgroup) so that a workgroup
    // walks the rows with as many of them side by side as its 256 threads allow, and writes the sums itself.
    const int64_t c4 = N / 4;
    int tx = 1;
    while (tx < c4 && tx < 32) tx <<= 1;
    const int ty = 256 / tx;
    DVAE_LAUNCH(colsum_kernel<1>, dim3(1, (unsigned)ceil_div64(c4, tx)), dim3(256), 0, ST(stream), dout, out,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, act,
                       dbias ? dbias : ws, M, N, tx, ty, dpre, CS_NOFIN);
    DVAE_LAUNCH_CHECK("bias_act_bwd");
    return 0;
  }
  if (dbias && colsum_one_launch(tickets, N) && aligned16(ws) && aligned16(dbias)) {
    // partial rows and their sum in one launch
    const ColsumGeom g = colsum_geom_fused(M, N);
    DVAE_LAUNCH(colsum_kernel<1>, dim3((unsigned)g.gx, g.gy), dim3(256), 0, ST(stream), dout, out,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, act, ws,
                       M, N, g.tx, g.ty, dpre, tickets, 0, 0.f, dbias, (float*)nullptr, (float*)nullptr, (float*)nullptr);
    DVAE_LAUNCH_CHECK("bias_act_bwd");
    return 0;
  }
  const ColsumGeom g = colsum_geom(M, N);
  DVAE_LAUNCH(colsum_kernel<1>, dim3((unsigned)g.gx, g.gy), dim3(256), 0, ST(stream), dout, out,
                     (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, act, ws,
                     M, N, g.tx, g.ty, dpre, CS_NOFIN);
  DVAE_LAUNCH_CHECK("bias_act_bwd");
  if (dbias) {
    DVAE_LAUNCH(colsum_finalize_kernel, dim3((unsigned)N), dim3(256), 0, ST(stream), (const float*)ws, g.gx, 1,
                       N, dbias, (float*)nullptr);
    DVAE_LAUNCH_CHECK("bias_act_bwd_finalize");
  }
  return 0;
}

extern "C" int dvae_bias_act_bwd_f32(float* dout, const float* out, int act, float* dbias, float* ws, unsigned* tickets,
                                     int64_t M, int64_t N, dvae_stream_t stream) {
  return dvae_bias_act_bwd_to_f32(dout, dout, out, act, dbias, ws, tickets, M, N, stream);
}

__global__ void __launch_bounds__(256)
sigmoid_bwd_kernel(const float* __restrict__ dr, const float* __restrict__ r, float* __restrict__ dl,
                   double* __restrict__ partials, int64_t n) {
  __shared__ double red[4];
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float rv = r[i];
    const float v = dr[i] * rv * (1.f - rv);
    dl[i] = v;
    s += (double)v;
  }
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

__global__ void __launch_bounds__(256)
sum_partials_kernel(const double* __restrict__ partials, int n, float* out) {
  __shared__ double red[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += partials[i];
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) out[0] = (float)s;
}

extern "C" int dvae_sigmoid_bwd_f32(const float* drecon, const float* recon, float* dlogit, float* dbias, float* ws,
                                    int64_t n, dvae_stream_t stream) {
  DVAE_REQUIRE(n > 0, "sigmoid_bwd: empty");
  DVAE_REQUIRE(((uintptr_t)ws & 7) == 0, "sigmoid_bwd: workspace must be 8-byte aligned");
  int64_t nb = ceil_div64(n, 256 * 8);
  if (nb > 1024) nb = 1024;
  DVAE_WORK(12.0 * (double)n, DVAE_UNIT_BYTE);
  DVAE_LAUNCH(sigmoid_bwd_kernel, dim3((unsigned)nb), dim3(256), 0, ST(stream), drecon, recon, dlogit,
                     (double*)ws, n);
  DVAE_LAUNCH_CHECK("sigmoid_bwd");
  if (dbias) {
    DVAE_LAUNCH(sum_partials_kernel, dim3(1), dim3(256), 0, ST(stream), (const double*)ws, (int)nb, dbias);
    DVAE_LAUNCH_CHECK("sigmoid_bwd_sum");
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------
// layout helpers: 3-D permutation (weight re-ordering between the reference's NCHW flatten order and
// the NHWC order the kernels use; batch-major <-> time-major sequences) and a weighted sum of scalars
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
permute3_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t D0, int64_t D1, int64_t D2, int p0,
                int p1, int p2) {
  // out dims = (D[p0], D[p1], D[p2]); threads walk the output linearly
  const int64_t d[3] = {D0, D1, D2};
  const int64_t istr[3] = {D1 * D2, D2, 1};
  const int64_t o1 = d[p1], o2 = d[p2];
  const int64_t total = D0 * D1 * D2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t c = i % o2, b = (i / o2) % o1, a = i / (o2 * o1);
    out[i] = in[a * istr[p0] + b * istr[p1] + c * istr[p2]];
  }
}

// (0,2,1): D0 independent D1 x D2 transposes through 32x32 LDS tiles, both sides coalesced
__global__ void __launch_bounds__(256)
transpose_inner_kernel(const float* __restrict__ in, float* __restrict__ out, int D1, int D2, int t1, int t2) {
  __shared__ float tile[32][33];
  const int b = blockIdx.x;
  const int a = b / (t1 * t2), r = b - a * (t1 * t2);
  const int i0 = (r / t2) * 32, j0 = (r % t2) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const float* src = in + (int64_t)a * D1 * D2;
  float* dst = out + (int64_t)a * D1 * D2;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = i0 + ty + 8 * k, j = j0 + tx;
    if (i < D1 && j < D2) tile[ty + 8 * k][tx] = src[(int64_t)i * D2 + j];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int j = j0 + ty + 8 * k, i = i0 + tx;
    if (i < D1 && j < D2) dst[(int64_t)j * D1 + i] = tile[tx][ty + 8 * k];
  }
}

// (1,0,2) with rows of whole float4: out[b][a][:] = in[a][b][:]
__global__ void __launch_bounds__(256)
swap_outer_kernel(const float4* __restrict__ in, float4* __restrict__ out, int D0, int D1, int V, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / V;
    const int c = (int)(i - row * V);
    const int a = (int)(row % D0), b = (int)(row / D0);          // output row (b, a)
    out[i] = in[((int64_t)a * D1 + b) * V + c];
  }
}

extern "C" int dvae_permute3_f32(const float* in, float* out, int64_t D0, int64_t D1, int64_t D2, int p0, int p1,
                                 int p2, dvae_stream_t stream) {
  DVAE_REQUIRE(D0 > 0 && D1 > 0 && D2 > 0, "permute3: empty");
  DVAE_REQUIRE(p0 >= 0 && p0 < 3 && p1 >= 0 && p1 < 3 && p2 >= 0 && p2 < 3 && p0 != p1 && p0 != p2 && p1 != p2,
               "permute3: (%d,%d,%d) is not a permutation", p0, p1, p2);
  if (p0 == 0 && p1 == 2 && p2 == 1 && D1 > 1 && D2 > 1 && D1 < (1 << 30) && D2 < (1 << 30)) {
    const int64_t t1 = ceil_div64(D1, 32), t2 = ceil_div64(D2, 32);
    if (D0 * t1 * t2 < (1LL << 31) && t1 * t2 < (1LL << 31)) {
      DVAE_LAUNCH(transpose_inner_kernel, dim3((unsigned)(D0 * t1 * t2)), dim3(256), 0, ST(stream), in, out,
                         (int)D1, (int)D2, (int)t1, (int)t2);
      DVAE_LAUNCH_CHECK("permute3");
      return 0;
    }
  }
  if (p0 == 1 && p1 == 0 && p2 == 2 && D2 % 4 == 0 && D0 < (1 << 30) && D1 < (1 << 30) && D2 < (1LL << 32) &&
      (((uintptr_t)in | (uintptr_t)out) & 15) == 0) {
    const int64_t total = D0 * D1 * (D2 / 4);
    int64_t nbv = ceil_div64(total, 256);
    if (nbv > 8192) nbv = 8192;
    DVAE_LAUNCH(swap_outer_kernel, dim3((unsigned)nbv), dim3(256), 0, ST(stream), (const float4*)in, (float4*)out,
                       (int)D0, (int)D1, (int)(D2 / 4), total);
    DVAE_LAUNCH_CHECK("permute3");
    return 0;
  }
  int64_t nb = ceil_div64(D0 * D1 * D2, 256);
  if (nb > 4096) nb = 4096;
  DVAE_LAUNCH(permute3_kernel, dim3((unsigned)nb), dim3(256), 0, ST(stream), in, out, D0, D1, D2, p0, p1, p2);
  DVAE_LAUNCH_CHECK("permute3");
  return 0;
}

struct WSumArgs {
  const float* src[8];
  float coef[8];
  int n;
};

__global__ void weighted_sum_kernel(WSumArgs a, float* out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < a.n; ++i) s += a.coef[i] * a.src[i][0];
    out[0] = s;
  }
}

extern "C" int dvae_weighted_sum_f32(int n, const float* const* terms, const float* coefs, float* out,
                                     dvae_stream_t stream) {
  DVAE_REQUIRE(n > 0 && n <= 8, "weighted_sum: 1..8 terms (got %d)", n);
  WSumArgs a;
  a.n = n;
  for (int i = 0; i < n; ++i) {
    a.src[i] = terms[i];
    a.coef[i] = coefs[i];
  }
  DVAE_LAUNCH(weighted_sum_kernel, dim3(1), dim3(64), 0, ST(stream), a, out);
  DVAE_LAUNCH_CHECK("weighted_sum");
  return 0;
}
