// LSTM recurrence kernels.  The recurrence is latency bound (B rows x 4H gate columns x H deep per
// step, strictly sequential in time), so the work of one step is spread over as many SIMDs as the
// problem allows: a workgroup owns a 32-row x 32-column output tile (for the forward step: 4 gates x 8
// hidden units, so the cell update is local to the workgroup) and its four waves split K between
// them, each feeding v_mfma_f32_32x32x2_f32 from its own quarter of every 128-deep K chunk; the four
// partial tiles are summed through LDS, then the sigmoid/tanh cell update is applied in the same
// kernel.  Steps are separate launches on one stream (captured into a graph by the caller): on this
// chip a kernel boundary is cheaper than a grid-wide barrier inside a persistent launch.
#include "common.h"
#include "../../include/dvae_hip.h"

#define ST(s) ((hipStream_t)(s))
#define KS_CH 128           // K chunk staged per iteration
#define KS_LDA (KS_CH + 4)  // k-contiguous image: [32 rows][KS_CH + 4]
#define KS_LDB_RC 36        // row-contiguous image: [KS_CH k][32 + 4]
#define KS_RED_LD 40

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// ---- staging helpers (256 threads move a 32 x 128 operand chunk = 4 float4 each) ---------------
// k-contiguous source: row pointer + k
__device__ __forceinline__ void ks_fetch_kc(const float* __restrict__ base, int64_t ld, int64_t row0, int64_t rows,
                                            int64_t k0, int64_t K, float4 (&r)[4]) {
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int L = v * 256 + threadIdx.x;
    const int row = L >> 5, ch = L & 31;
    const int64_t gr = row0 + row, k = k0 + 4 * ch;
    r[v] = (base != nullptr && gr < rows && k < K) ? *reinterpret_cast<const float4*>(base + gr * ld + k)
                                                   : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
__device__ __forceinline__ void ks_store_kc(float* lds, const float4 (&r)[4]) {
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int L = v * 256 + threadIdx.x;
    *reinterpret_cast<float4*>(lds + (L >> 5) * KS_LDA + 4 * (L & 31)) = r[v];
  }
}
// gate-row gather for the forward step: tile row c (0..31) is weight row (c>>3)*H + j0 + (c&7)
__device__ __forceinline__ void ks_fetch_gate_rows(const float* __restrict__ w, int64_t ld, int64_t H, int64_t j0,
                                                   int64_t k0, int64_t K, float4 (&r)[4]) {
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int L = v * 256 + threadIdx.x;
    const int c = L >> 5, ch = L & 31;
    const int64_t wr = (int64_t)(c >> 3) * H + j0 + (c & 7), k = k0 + 4 * ch;
    r[v] = (w != nullptr && k < K) ? *reinterpret_cast<const float4*>(w + wr * ld + k)
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
// row-contiguous source [K][N]: chunk rows k0..k0+127, columns n0..n0+31
__device__ __forceinline__ void ks_fetch_rc(const float* __restrict__ base, int64_t ld, int64_t n0, int64_t N,
                                            int64_t k0, int64_t K, float4 (&r)[4]) {
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int L = v * 256 + threadIdx.x;
    const int kr = L >> 3, c4 = L & 7;
    const int64_t k = k0 + kr, n = n0 + 4 * c4;
    r[v] = (k < K && n < N) ? *reinterpret_cast<const float4*>(base + k * ld + n) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
__device__ __forceinline__ void ks_store_rc(float* lds, const float4 (&r)[4]) {
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int L = v * 256 + threadIdx.x;
    *reinterpret_cast<float4*>(lds + (L >> 3) * KS_LDB_RC + 4 * (L & 7)) = r[v];
  }
}

template <bool BKC>
__device__ __forceinline__ void ks_mma(const float* __restrict__ As, const float* __restrict__ Bs, f32x16& acc,
                                       int wave, int lane) {
  const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int kk = 32 * wave + 8 * q + 4 * h;
    const float4 av = *reinterpret_cast<const float4*>(As + l31 * KS_LDA + kk);
    float b0, b1, b2, b3;
    if (BKC) {
      const float4 bv = *reinterpret_cast<const float4*>(Bs + l31 * KS_LDA + kk);
      b0 = bv.x; b1 = bv.y; b2 = bv.z; b3 = bv.w;
    } else {
      b0 = Bs[(kk + 0) * KS_LDB_RC + l31];
      b1 = Bs[(kk + 1) * KS_LDB_RC + l31];
      b2 = Bs[(kk + 2) * KS_LDB_RC + l31];
      b3 = Bs[(kk + 3) * KS_LDB_RC + l31];
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b2, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b3, acc, 0, 0, 0);
  }
}

// write the wave's partial tile to red[wave][32][KS_RED_LD]
__device__ __forceinline__ void ks_spill(float* red, const f32x16& acc, int wave, int lane) {
  const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
    red[(wave * 32 + row) * KS_RED_LD + l31] = acc[r];
  }
}

#define KS_STAGE_KC (32 * KS_LDA)
#define KS_STAGE_RC (KS_CH * KS_LDB_RC)

// ------------------------------------------------------------------------------------------------
// forward step
// ------------------------------------------------------------------------------------------------
struct LstmFwdArgs {
  const float* xproj; int64_t ld_xproj;
  const float* x; int64_t ld_x;
  const float* w_ih; int64_t I;
  const float* h_prev; int64_t ld_h;
  const float* c_prev; int64_t ld_c;
  const float* w_hh;
  const float* b_ih;
  const float* b_hh;
  float* gates_out;
  float* h_out; int64_t ld_hout;
  float* c_out; int64_t ld_cout;
  int64_t B, H;
};

__global__ void __launch_bounds__(256) lstm_step_fwd_kernel(LstmFwdArgs g) {
  __shared__ __attribute__((aligned(16))) float smem[4 * KS_STAGE_KC];  // 2 stages x (A, B)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t j0 = (int64_t)blockIdx.x * 8, b0 = (int64_t)blockIdx.y * 32;
  const int nx = (g.x != nullptr) ? (int)((g.I + KS_CH - 1) / KS_CH) : 0;
  const int nh = (g.h_prev != nullptr) ? (int)((g.H + KS_CH - 1) / KS_CH) : 0;
  const int nchunks = nx + nh;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  float4 ra[4], rb[4];
  auto fetch = [&](int c) {
    if (c < nx) {
      ks_fetch_kc(g.x, g.ld_x, b0, g.B, (int64_t)c * KS_CH, g.I, ra);
      ks_fetch_gate_rows(g.w_ih, g.I, g.H, j0, (int64_t)c * KS_CH, g.I, rb);
    } else {
      ks_fetch_kc(g.h_prev, g.ld_h, b0, g.B, (int64_t)(c - nx) * KS_CH, g.H, ra);
      ks_fetch_gate_rows(g.w_hh, g.H, g.H, j0, (int64_t)(c - nx) * KS_CH, g.H, rb);
    }
  };
  if (nchunks > 0) {
    fetch(0);
    ks_store_kc(smem, ra);
    ks_store_kc(smem + KS_STAGE_KC, rb);
  }
  __syncthreads();
  int cur = 0;
  for (int c = 0; c < nchunks; ++c) {
    const bool more = c + 1 < nchunks;
    if (more) fetch(c + 1);
    const float* st = smem + cur * 2 * KS_STAGE_KC;
    ks_mma<true>(st, st + KS_STAGE_KC, acc, wave, lane);
    if (more) {
      float* nxp = smem + (cur ^ 1) * 2 * KS_STAGE_KC;
      ks_store_kc(nxp, ra);
      ks_store_kc(nxp + KS_STAGE_KC, rb);
    }
    __syncthreads();
    cur ^= 1;
  }
  float* red = smem;
  ks_spill(red, acc, wave, lane);
  __syncthreads();
  // cell update: thread -> (row b, hidden unit jj)
  const int bl = threadIdx.x >> 3, jj = threadIdx.x & 7;
  const int64_t b = b0 + bl, j = j0 + jj;
  if (b < g.B) {
    float pre[4];
#pragma unroll
    for (int gt = 0; gt < 4; ++gt) {
      const int col = gt * 8 + jj;
      float s = red[(0 * 32 + bl) * KS_RED_LD + col] + red[(1 * 32 + bl) * KS_RED_LD + col] +
                red[(2 * 32 + bl) * KS_RED_LD + col] + red[(3 * 32 + bl) * KS_RED_LD + col];
      const int64_t gc = (int64_t)gt * g.H + j;
      if (g.xproj) s += g.xproj[b * g.ld_xproj + gc];
      if (g.b_ih) s += g.b_ih[gc] + g.b_hh[gc];
      pre[gt] = s;
    }
    const float ig = sigmoidf_(pre[0]), fg = sigmoidf_(pre[1]), gg = tanhf(pre[2]), og = sigmoidf_(pre[3]);
    const float cp = g.c_prev ? g.c_prev[b * g.ld_c + j] : 0.f;
    const float cn = fg * cp + ig * gg;
    const float hn = og * tanhf(cn);
    float* go = g.gates_out + b * 4 * g.H;
    go[0 * g.H + j] = ig;
    go[1 * g.H + j] = fg;
    go[2 * g.H + j] = gg;
    go[3 * g.H + j] = og;
    g.c_out[b * g.ld_cout + j] = cn;
    g.h_out[b * g.ld_hout + j] = hn;
  }
}

extern "C" int dvae_lstm_step_fwd_f32(const float* xproj, int64_t ld_xproj, const float* x, int64_t ld_x,
                                      const float* w_ih, int64_t I, const float* h_prev, int64_t ld_h,
                                      const float* c_prev, int64_t ld_c, const float* w_hh, const float* b_ih,
                                      const float* b_hh, float* gates_out, float* h_out, int64_t ld_hout,
                                      float* c_out, int64_t ld_cout, int64_t B, int64_t H, dvae_stream_t stream) {
  DVAE_REQUIRE(B > 0 && H > 0 && H % 8 == 0, "lstm_step_fwd: H must be a multiple of 8 (H=%ld)", (long)H);
  DVAE_REQUIRE((x == nullptr) == (w_ih == nullptr), "lstm_step_fwd: x and w_ih come together");
  DVAE_REQUIRE(x == nullptr || (I > 0 && I % 4 == 0 && ld_x % 4 == 0 && aligned16(x) && aligned16(w_ih)),
               "lstm_step_fwd: x/w_ih need I%%4==0, ld_x%%4==0 and 16-byte alignment");
  DVAE_REQUIRE(h_prev == nullptr || (ld_h % 4 == 0 && aligned16(h_prev)), "lstm_step_fwd: h_prev needs ld_h%%4==0 and 16-byte alignment");
  DVAE_REQUIRE(w_hh != nullptr && aligned16(w_hh), "lstm_step_fwd: w_hh must be 16-byte aligned");
  DVAE_REQUIRE((b_ih == nullptr) == (b_hh == nullptr), "lstm_step_fwd: biases come in pairs");
  DVAE_REQUIRE(gates_out && h_out && c_out, "lstm_step_fwd: outputs must be non-null");
  LstmFwdArgs g{xproj, ld_xproj, x, ld_x, w_ih, I, h_prev, ld_h, c_prev, ld_c, w_hh, b_ih, b_hh,
                gates_out, h_out, ld_hout, c_out, ld_cout, B, H};
  const dim3 grid((unsigned)(H / 8), (unsigned)ceil_div64(B, 32));
  hipLaunchKernelGGL(lstm_step_fwd_kernel, grid, dim3(256), 0, ST(stream), g);
  DVAE_LAUNCH_CHECK("lstm_step_fwd");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// backward: pointwise gate gradients, then skinny GEMMs against the weights
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
lstm_bwd_pointwise_kernel(const float* __restrict__ dh, int64_t ld_dh, const float* __restrict__ dc, int64_t ld_dc,
                          const float* __restrict__ gates, const float* __restrict__ c_prev, int64_t ld_cprev,
                          const float* __restrict__ c, int64_t ld_c, float* __restrict__ dgates,
                          float* __restrict__ dc_prev, int64_t ld_dcprev, int64_t B, int64_t H) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= B * H) return;
  const int64_t b = idx / H, j = idx - b * H;
  const float* gt = gates + b * 4 * H;
  const float ig = gt[j], fg = gt[H + j], gg = gt[2 * H + j], og = gt[3 * H + j];
  const float tc = tanhf(c[b * ld_c + j]);
  const float dhv = dh[b * ld_dh + j];
  const float dct = (dc ? dc[b * ld_dc + j] : 0.f) + dhv * og * (1.f - tc * tc);
  const float cp = c_prev ? c_prev[b * ld_cprev + j] : 0.f;
  float* dg = dgates + b * 4 * H;
  dg[j] = dct * gg * ig * (1.f - ig);
  dg[H + j] = dct * cp * fg * (1.f - fg);
  dg[2 * H + j] = dct * ig * (1.f - gg * gg);
  dg[3 * H + j] = dhv * tc * og * (1.f - og);
  dc_prev[b * ld_dcprev + j] = dct * fg;
}

extern "C" int dvae_lstm_step_bwd_pointwise_f32(const float* dh, int64_t ld_dh, const float* dc, int64_t ld_dc,
                                                const float* gates, const float* c_prev, int64_t ld_cprev,
                                                const float* c, int64_t ld_c, float* dgates, float* dc_prev,
                                                int64_t ld_dcprev, int64_t B, int64_t H, dvae_stream_t stream) {
  DVAE_REQUIRE(B > 0 && H > 0, "lstm_step_bwd_pointwise: empty");
  DVAE_REQUIRE(dh && gates && c && dgates && dc_prev, "lstm_step_bwd_pointwise: null argument");
  hipLaunchKernelGGL(lstm_bwd_pointwise_kernel, dim3((unsigned)ceil_div64(B * H, 256)), dim3(256), 0, ST(stream), dh,
                     ld_dh, dc, ld_dc, gates, c_prev, ld_cprev, c, ld_c, dgates, dc_prev, ld_dcprev, B, H);
  DVAE_LAUNCH_CHECK("lstm_step_bwd_pointwise");
  return 0;
}

struct SkinnyArgs {
  const float* a; int64_t ld_a;
  const float* w;
  const float* w2;
  float* out; int64_t ld_out;
  int64_t B, N, K;
  int accumulate;
};

__global__ void __launch_bounds__(256) skinny_gemm_nn_kernel(SkinnyArgs g) {
  __shared__ __attribute__((aligned(16))) float smem[2 * (KS_STAGE_KC + KS_STAGE_RC)];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t n0 = (int64_t)blockIdx.x * 32, b0 = (int64_t)blockIdx.y * 32;
  const int per = (int)((g.K + KS_CH - 1) / KS_CH);
  const int nchunks = per * (g.w2 ? 2 : 1);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float4 ra[4], rb[4];
  auto fetch = [&](int c) {
    const int seg = c / per, cc = c - seg * per;
    ks_fetch_kc(g.a, g.ld_a, b0, g.B, (int64_t)cc * KS_CH, g.K, ra);
    ks_fetch_rc(seg == 0 ? g.w : g.w2, g.N, n0, g.N, (int64_t)cc * KS_CH, g.K, rb);
  };
  fetch(0);
  ks_store_kc(smem, ra);
  ks_store_rc(smem + KS_STAGE_KC, rb);
  __syncthreads();
  int cur = 0;
  for (int c = 0; c < nchunks; ++c) {
    const bool more = c + 1 < nchunks;
    if (more) fetch(c + 1);
    const float* st = smem + cur * (KS_STAGE_KC + KS_STAGE_RC);
    ks_mma<false>(st, st + KS_STAGE_KC, acc, wave, lane);
    if (more) {
      float* nxp = smem + (cur ^ 1) * (KS_STAGE_KC + KS_STAGE_RC);
      ks_store_kc(nxp, ra);
      ks_store_rc(nxp + KS_STAGE_KC, rb);
    }
    __syncthreads();
    cur ^= 1;
  }
  float* red = smem;
  ks_spill(red, acc, wave, lane);
  __syncthreads();
  // 32 x 32 outputs, 4 per thread
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int i = e * 256 + threadIdx.x;
    const int row = i >> 5, col = i & 31;
    const int64_t b = b0 + row, n = n0 + col;
    if (b < g.B && n < g.N) {
      float s = red[(0 * 32 + row) * KS_RED_LD + col] + red[(1 * 32 + row) * KS_RED_LD + col] +
                red[(2 * 32 + row) * KS_RED_LD + col] + red[(3 * 32 + row) * KS_RED_LD + col];
      float* p = g.out + b * g.ld_out + n;
      if (g.accumulate) s += *p;
      *p = s;
    }
  }
}

extern "C" int dvae_skinny_gemm_nn_f32(const float* dg, int64_t ld_dg, const float* w, const float* w2, float* out,
                                       int64_t ld_out, int64_t B, int64_t N, int64_t K, int accumulate,
                                       dvae_stream_t stream) {
  DVAE_REQUIRE(B > 0 && N > 0 && K > 0, "skinny_gemm_nn: empty");
  DVAE_REQUIRE(K % 4 == 0 && N % 4 == 0 && ld_dg % 4 == 0, "skinny_gemm_nn: K, N, ld_dg must be multiples of 4");
  DVAE_REQUIRE(aligned16(dg) && aligned16(w) && (w2 == nullptr || aligned16(w2)), "skinny_gemm_nn: 16-byte alignment");
  SkinnyArgs g{dg, ld_dg, w, w2, out, ld_out, B, N, K, accumulate};
  const dim3 grid((unsigned)ceil_div64(N, 32), (unsigned)ceil_div64(B, 32));
  hipLaunchKernelGGL(skinny_gemm_nn_kernel, grid, dim3(256), 0, ST(stream), g);
  DVAE_LAUNCH_CHECK("skinny_gemm_nn");
  return 0;
}
