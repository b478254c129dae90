// LSTM recurrence kernels.  The recurrence is latency bound (B rows x 4H gate columns x H deep per
// step, strictly sequential in time), so one step is shaped for the shortest path through a single
// launch, not for operand reuse: the output is cut into as many small tiles as there are compute units
// (forward: 32 rows x [4 gates x 8 hidden units], so the cell update is local to the workgroup;
// backward: 16 x 16), the waves of a workgroup split K between them, and every wave loads its operand
// fragments straight from memory (L2 resident: weights and one step's state) into the register layout
// of v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32 -- all loads of a wave are in flight at once, no
// staging through LDS, one barrier.  The partial tiles are summed through LDS and the sigmoid/tanh
// cell update (forward) or the accumulation into dh (backward) happens in the same kernel.  These
// per-step kernels serve stacked decoders and shapes the one-launch kernels do not admit.
//
// Whole recurrences (second half of this file) run as ONE launch where every workgroup can be resident:
// no grid-wide barrier -- only the H/8 workgroups that serve one group of 32 batch rows hand data to each
// other, through 16-byte write-through stores, one arrival counter per group and L1-bypassing loads.
#include "common.h"
#include <mutex>
#include <cstdlib>
#include "../../include/dvae_hip.h"

#define ST(s) ((hipStream_t)(s))
#define KS_RED_LD 40

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// NF consecutive 16-byte pieces of one k-contiguous operand row, `stride` floats apart, starting at
// k0: piece q covers k0 + stride*q .. +3.  Branch-free (clamped address, value zeroed afterwards) so
// that the compiler keeps all the loads of a wave in flight together.  K%4 == 0.
template <int NF>
__device__ __forceinline__ void frag_load(const float* __restrict__ rowp, bool rowok, int64_t k0, int stride,
                                          int64_t K, f32x4 (&r)[NF]) {
#pragma unroll
  for (int q = 0; q < NF; ++q) {
    const int64_t kk = k0 + stride * q;
    const bool ok = rowok && kk < K;
    const f32x4 v = *reinterpret_cast<const f32x4*>(rowp + (ok ? kk : 0));
    r[q] = ok ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
}

// write the wave's partial 32x32 tile to red[wave][32][KS_RED_LD]
__device__ __forceinline__ void ks_spill(float* red, const f32x16& acc, int wave, int lane) {
  const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
    red[(wave * 32 + row) * KS_RED_LD + l31] = acc[r];
  }
}

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

#define FWD_WAVES 4     // the cell update below maps 256 threads onto 32 rows x 8 hidden units
__global__ void __launch_bounds__(64 * FWD_WAVES) lstm_step_fwd_kernel(LstmFwdArgs g) {
  __shared__ float red[FWD_WAVES * 32 * KS_RED_LD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int64_t j0 = (int64_t)blockIdx.x * 8, b0 = (int64_t)blockIdx.y * 32;
  // lane l31 feeds batch row b0 + l31 as the A operand and gate row (l31>>3)*H + j0 + (l31&7) as B
  const int64_t row = b0 + l31;
  const bool rowok = row < g.B;
  const int64_t rc = rowok ? row : g.B - 1;
  const int64_t wr = (int64_t)(l31 >> 3) * g.H + j0 + (l31 & 7);

  // operands of the cell update (thread -> row bl, hidden unit jj): requested first, so that they
  // arrive while the products are formed
  const int bl = threadIdx.x >> 3, jj = threadIdx.x & 7;
  const int64_t b = b0 + bl, j = j0 + jj;
  // Unconditional loads (absent operands read a dummy address and are dropped at the point of
  // use), so that nothing here makes the wave wait before the operand loads below are issued.
  const int64_t bb = b < g.B ? b : g.B - 1;
  const float* xpp = g.xproj ? g.xproj + bb * g.ld_xproj + j : g.w_hh;
  const float* bip = g.b_ih ? g.b_ih + j : g.w_hh;
  const float* bhp = g.b_ih ? g.b_hh + j : g.w_hh;
  const int64_t xstep = g.xproj ? g.H : 0, bstep = g.b_ih ? g.H : 0;
  float raw_x[4], raw_bi[4], raw_bh[4];
#pragma unroll
  for (int gt = 0; gt < 4; ++gt) {
    raw_x[gt] = xpp[gt * xstep];
    raw_bi[gt] = bip[gt * bstep];
    raw_bh[gt] = bhp[gt * bstep];
  }
  const float raw_c = *(g.c_prev ? g.c_prev + bb * g.ld_c + j : g.w_hh);

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int seg = 0; seg < 2; ++seg) {
    const float* A = seg == 0 ? g.x : g.h_prev;
    if (A == nullptr) continue;
    const int64_t K = seg == 0 ? g.I : g.H;
    const float* ap = A + rc * (seg == 0 ? g.ld_x : g.ld_h);
    const float* bp = (seg == 0 ? g.w_ih : g.w_hh) + wr * K;
    // a wave takes 64 consecutive k per pass: eight MFMA quads, each fed by one 16-byte load per operand
    for (int64_t k0 = (int64_t)wave * 64; k0 < K; k0 += 64 * FWD_WAVES) {
      f32x4 av[8], bv[8];
      frag_load<8>(ap, rowok, k0 + 4 * h, 8, K, av);
      frag_load<8>(bp, true, k0 + 4 * h, 8, K, bv);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][0], bv[q][0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][1], bv[q][1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][2], bv[q][2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][3], bv[q][3], acc, 0, 0, 0);
      }
    }
  }
  ks_spill(red, acc, wave, lane);
  __syncthreads();
  if (b < g.B) {
    float pre[4];
#pragma unroll
    for (int gt = 0; gt < 4; ++gt) {
      const int col = gt * 8 + jj;
      const float s = red[(0 * 32 + bl) * KS_RED_LD + col] + red[(1 * 32 + bl) * KS_RED_LD + col] +
                      red[(2 * 32 + bl) * KS_RED_LD + col] + red[(3 * 32 + bl) * KS_RED_LD + col];
      pre[gt] = (s + (g.xproj ? raw_x[gt] : 0.f)) + (g.b_ih ? raw_bi[gt] + raw_bh[gt] : 0.f);
    }
    const float cp = g.c_prev ? raw_c : 0.f;
    const float ig = sigmoidf_(pre[0]), fg = sigmoidf_(pre[1]), gg = tanhf(pre[2]), og = sigmoidf_(pre[3]);
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
  DVAE_WORK(2.0 * (double)B * 4 * H * ((x ? I : 0) + (h_prev ? H : 0)), DVAE_UNIT_FLOP);
  DVAE_LAUNCH(lstm_step_fwd_kernel, grid, dim3(64 * FWD_WAVES), 0, ST(stream), g);
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
  DVAE_LAUNCH(lstm_bwd_pointwise_kernel, dim3((unsigned)ceil_div64(B * H, 256)), dim3(256), 0, ST(stream), dh,
                     ld_dh, dc, ld_dc, gates, c_prev, ld_cprev, c, ld_c, dgates, dc_prev, ld_dcprev, B, H);
  DVAE_LAUNCH_CHECK("lstm_step_bwd_pointwise");
  return 0;
}

struct SkinnyArgs {
  const float* a; int64_t ld_a;
  const float* wt; int64_t ld_wt;
  float* out; int64_t ld_out;
  int64_t B, N, K;
  int accumulate;
};

#define SK_WAVES 8     // waves of a workgroup, K split between them
#define SK_NF 8        // 16-byte pieces per operand and pass: a wave covers 16*SK_NF consecutive k per pass
#define SK_RED_LD 17

// out[16 x 16 tile] (+)= a[16 rows][K] * wt[16 rows][K]^T.  Lane (i = lane&15, q = lane>>4) holds, per
// 16-deep group, a[row i][k..k+3] and wt[col i][k..k+3] with k = group + 4q: component m of the two
// 16-byte pieces is the operand pair of one v_mfma_f32_16x16x4_f32 (its four k are group + 4q + m).
__global__ void __launch_bounds__(64 * SK_WAVES) skinny_gemm_nt_kernel(SkinnyArgs g) {
  __shared__ float red[SK_WAVES * 16 * SK_RED_LD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int64_t n0 = (int64_t)blockIdx.x * 16, b0 = (int64_t)blockIdx.y * 16;
  const bool rowok = b0 + i < g.B, colok = n0 + i < g.N;
  const float* ap = g.a + (rowok ? b0 + i : g.B - 1) * g.ld_a;
  const float* bp = g.wt + (colok ? n0 + i : g.N - 1) * g.ld_wt;
  // the value to accumulate onto is requested first (threads 0..255 own one output each)
  const int orow = (threadIdx.x >> 4) & 15, ocol = threadIdx.x & 15;
  const bool owner = threadIdx.x < 256 && b0 + orow < g.B && n0 + ocol < g.N;
  float* const op = g.out + (b0 + orow) * g.ld_out + n0 + ocol;
  float prev = 0.f;
  if (owner && g.accumulate) prev = *op;
  // two accumulators: consecutive MFMAs of a wave do not wait for each other's result
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f}, acc2 = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int64_t k0 = (int64_t)wave * 16 * SK_NF; k0 < g.K; k0 += 16 * SK_NF * SK_WAVES) {
    f32x4 av[SK_NF], bv[SK_NF];
    frag_load<SK_NF>(ap, rowok, k0 + 4 * q, 16, g.K, av);
    frag_load<SK_NF>(bp, colok, k0 + 4 * q, 16, g.K, bv);
#pragma unroll
    for (int f = 0; f < SK_NF; ++f) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[f][0], bv[f][0], acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[f][1], bv[f][1], acc2, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[f][2], bv[f][2], acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[f][3], bv[f][3], acc2, 0, 0, 0);
    }
  }
  acc += acc2;
  // D: row = 4q + r, column = i
#pragma unroll
  for (int r = 0; r < 4; ++r) red[(wave * 16 + 4 * q + r) * SK_RED_LD + i] = acc[r];
  __syncthreads();
  if (owner) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < SK_WAVES; ++w) s += red[(w * 16 + orow) * SK_RED_LD + ocol];
    *op = s + prev;
  }
}

// One whole backward step in a launch: the skinny product above finishes dh_{t-1} (+= dG_t * W), and the
// thread that owns an element of it takes it straight through the cell's pointwise backward of step
// t-1 (what lstm_bwd_pointwise_kernel does), writing that step's gate gradients and dc.  The
// recurrence's critical path is then one kernel per time step instead of two.
struct LstmBwdStepArgs {
  SkinnyArgs mm;                 // out = dh_{t-1} [B,H], accumulated in place
  const float* a2; int64_t ld_a2;          // optional second product onto the same output: a2 [B,K2] * wt2 [H,K2]^T
  const float* wt2; int64_t ld_wt2;
  int64_t K2;
  const float* gates;            // activated gates of step t-1 [B,4H]
  const float* c_prev; int64_t ld_cprev;   // c_{t-2} (nullptr: zero state)
  const float* c; int64_t ld_c;            // c_{t-1}
  const float* dc; int64_t ld_dc;          // dc carried into step t-1
  float* dgates;                 // [B,4H] out
  float* dc_prev; int64_t ld_dcprev;       // out
};

__global__ void __launch_bounds__(64 * SK_WAVES) lstm_bwd_step_kernel(LstmBwdStepArgs a) {
  __shared__ float red[SK_WAVES * 16 * SK_RED_LD];
  const SkinnyArgs& g = a.mm;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int64_t n0 = (int64_t)blockIdx.x * 16, b0 = (int64_t)blockIdx.y * 16;
  const bool rowok = b0 + i < g.B, colok = n0 + i < g.N;
  const float* ap = g.a + (rowok ? b0 + i : g.B - 1) * g.ld_a;
  const float* bp = g.wt + (colok ? n0 + i : g.N - 1) * g.ld_wt;
  // everything the owner of an output needs besides the product is requested first
  const int orow = (threadIdx.x >> 4) & 15, ocol = threadIdx.x & 15;
  const bool owner = threadIdx.x < 256 && b0 + orow < g.B && n0 + ocol < g.N;
  const int64_t b = owner ? b0 + orow : 0, j = owner ? n0 + ocol : 0, H = g.N;
  float* const op = g.out + b * g.ld_out + j;
  const float prev = *op;
  const float* gt = a.gates + b * 4 * H;
  const float ig = gt[j], fg = gt[H + j], gg = gt[2 * H + j], og = gt[3 * H + j];
  const float cv = a.c[b * a.ld_c + j];
  const float dcin = *(a.dc ? a.dc + b * a.ld_dc + j : a.c);
  const float cpv = *(a.c_prev ? a.c_prev + b * a.ld_cprev + j : a.c);

  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f}, acc2 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int seg = 0; seg < 2; ++seg) {
    // the second product (a stacked decoder: both gradients that reach one hidden state) continues the sums
    const float* sa = seg == 0 ? ap : (a.a2 ? a.a2 + (rowok ? b0 + i : g.B - 1) * a.ld_a2 : nullptr);
    const float* sb = seg == 0 ? bp : (a.a2 ? a.wt2 + (colok ? n0 + i : g.N - 1) * a.ld_wt2 : nullptr);
    const int64_t sK = seg == 0 ? g.K : a.K2;
    if (sa == nullptr) break;
    for (int64_t k0 = (int64_t)wave * 16 * SK_NF; k0 < sK; k0 += 16 * SK_NF * SK_WAVES) {
      f32x4 av[SK_NF], bv[SK_NF];
      frag_load<SK_NF>(sa, rowok, k0 + 4 * q, 16, sK, av);
      frag_load<SK_NF>(sb, colok, k0 + 4 * q, 16, sK, bv);
#pragma unroll
      for (int f = 0; f < SK_NF; ++f) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[f][0], bv[f][0], acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[f][1], bv[f][1], acc2, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[f][2], bv[f][2], acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[f][3], bv[f][3], acc2, 0, 0, 0);
      }
    }
  }
  acc += acc2;
#pragma unroll
  for (int r = 0; r < 4; ++r) red[(wave * 16 + 4 * q + r) * SK_RED_LD + i] = acc[r];
  __syncthreads();
  if (owner) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < SK_WAVES; ++w) s += red[(w * 16 + orow) * SK_RED_LD + ocol];
    const float dhv = s + prev;
    *op = dhv;
    const float tc = tanhf(cv);
    const float dct = (a.dc ? dcin : 0.f) + dhv * og * (1.f - tc * tc);
    const float cp = a.c_prev ? cpv : 0.f;
    float* dg = a.dgates + b * 4 * H;
    dg[j] = dct * gg * ig * (1.f - ig);
    dg[H + j] = dct * cp * fg * (1.f - fg);
    dg[2 * H + j] = dct * ig * (1.f - gg * gg);
    dg[3 * H + j] = dhv * tc * og * (1.f - og);
    a.dc_prev[b * a.ld_dcprev + j] = dct * fg;
  }
}

extern "C" int dvae_lstm_bwd_step_f32(const float* dg_t, int64_t ld_dg, const float* wt, int64_t ld_wt, float* dh_prev,
                                      int64_t ld_dh, const float* dc, int64_t ld_dc, const float* gates,
                                      const float* c_prev, int64_t ld_cprev, const float* c, int64_t ld_c,
                                      float* dgates, float* dc_prev, int64_t ld_dcprev, int64_t B, int64_t H,
                                      int64_t K, dvae_stream_t stream) {
  DVAE_REQUIRE(B > 0 && H > 0 && K > 0, "lstm_bwd_step: empty");
  DVAE_REQUIRE(dg_t && wt && dh_prev && gates && c && dgates && dc_prev, "lstm_bwd_step: null argument");
  DVAE_REQUIRE(K % 4 == 0 && ld_dg % 4 == 0 && ld_wt % 4 == 0, "lstm_bwd_step: K, ld_dg, ld_wt must be multiples of 4");
  DVAE_REQUIRE(aligned16(dg_t) && aligned16(wt), "lstm_bwd_step: 16-byte alignment");
  LstmBwdStepArgs a{SkinnyArgs{dg_t, ld_dg, wt, ld_wt, dh_prev, ld_dh, B, H, K, 1}, nullptr, 0, nullptr, 0, 0,
                    gates, c_prev, ld_cprev, c, ld_c, dc, ld_dc, dgates, dc_prev, ld_dcprev};
  const dim3 grid((unsigned)ceil_div64(H, 16), (unsigned)ceil_div64(B, 16));
  DVAE_WORK(2.0 * (double)B * H * K, DVAE_UNIT_FLOP);
  DVAE_LAUNCH(lstm_bwd_step_kernel, grid, dim3(64 * SK_WAVES), 0, ST(stream), a);
  DVAE_LAUNCH_CHECK("lstm_bwd_step");
  return 0;
}

extern "C" int dvae_lstm_bwd_step2_f32(const float* dg_a, int64_t ld_dga, const float* wt_a, int64_t ld_wta, int64_t Ka,
                                       const float* dg_b, int64_t ld_dgb, const float* wt_b, int64_t ld_wtb, int64_t Kb,
                                       float* dh_prev, int64_t ld_dh, const float* dc, int64_t ld_dc,
                                       const float* gates, const float* c_prev, int64_t ld_cprev, const float* c,
                                       int64_t ld_c, float* dgates, float* dc_prev, int64_t ld_dcprev, int64_t B,
                                       int64_t H, dvae_stream_t stream) {
  DVAE_REQUIRE(B > 0 && H > 0 && Ka > 0, "lstm_bwd_step2: empty");
  DVAE_REQUIRE(dg_a && wt_a && dh_prev && gates && c && dgates && dc_prev, "lstm_bwd_step2: null argument");
  DVAE_REQUIRE((dg_b == nullptr) == (wt_b == nullptr) && (dg_b == nullptr || Kb > 0), "lstm_bwd_step2: the second product needs both operands");
  DVAE_REQUIRE(Ka % 4 == 0 && ld_dga % 4 == 0 && ld_wta % 4 == 0 && aligned16(dg_a) && aligned16(wt_a),
               "lstm_bwd_step2: K and leading dimensions must be multiples of 4, operands 16-byte aligned");
  DVAE_REQUIRE(dg_b == nullptr || (Kb % 4 == 0 && ld_dgb % 4 == 0 && ld_wtb % 4 == 0 && aligned16(dg_b) && aligned16(wt_b)),
               "lstm_bwd_step2: K and leading dimensions must be multiples of 4, operands 16-byte aligned");
  LstmBwdStepArgs a{SkinnyArgs{dg_a, ld_dga, wt_a, ld_wta, dh_prev, ld_dh, B, H, Ka, 1}, dg_b, ld_dgb, wt_b, ld_wtb,
                    dg_b ? Kb : 0, gates, c_prev, ld_cprev, c, ld_c, dc, ld_dc, dgates, dc_prev, ld_dcprev};
  const dim3 grid((unsigned)ceil_div64(H, 16), (unsigned)ceil_div64(B, 16));
  DVAE_WORK(2.0 * (double)B * H * (Ka + (dg_b ? Kb : 0)), DVAE_UNIT_FLOP);
  DVAE_LAUNCH(lstm_bwd_step_kernel, grid, dim3(64 * SK_WAVES), 0, ST(stream), a);
  DVAE_LAUNCH_CHECK("lstm_bwd_step2");
  return 0;
}

extern "C" int dvae_skinny_gemm_nt_f32(const float* a, int64_t ld_a, const float* wt, int64_t ld_wt, float* out,
                                       int64_t ld_out, int64_t B, int64_t N, int64_t K, int accumulate,
                                       dvae_stream_t stream) {
  DVAE_REQUIRE(B > 0 && N > 0 && K > 0, "skinny_gemm_nt: empty");
  DVAE_REQUIRE(a && wt && out, "skinny_gemm_nt: null argument");
  DVAE_REQUIRE(K % 4 == 0 && ld_a % 4 == 0 && ld_wt % 4 == 0, "skinny_gemm_nt: K, ld_a, ld_wt must be multiples of 4");
  DVAE_REQUIRE(aligned16(a) && aligned16(wt), "skinny_gemm_nt: 16-byte alignment");
  SkinnyArgs g{a, ld_a, wt, ld_wt, out, ld_out, B, N, K, accumulate};
  const dim3 grid((unsigned)ceil_div64(N, 16), (unsigned)ceil_div64(B, 16));
  DVAE_WORK(2.0 * (double)B * N * K, DVAE_UNIT_FLOP);
  DVAE_LAUNCH(skinny_gemm_nt_kernel, grid, dim3(64 * SK_WAVES), 0, ST(stream), g);
  DVAE_LAUNCH_CHECK("skinny_gemm_nt");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Whole recurrence in ONE launch (forward).
//
// The batch rows are independent, so the work splits into groups of 32 rows, and a group into H/8
// workgroups that each own eight hidden units (4 x 8 = 32 gate columns).  A workgroup keeps its slice of
// the recurrent weights in registers for all T steps (each of its eight waves holds 32 columns x H/8 of
// the K dimension in MFMA operand layout), keeps the cell state of its 32 x 8 elements in registers, and
// per step needs only the group's previous hidden state [32 x H] from the other workgroups of its group.
// That hand-off goes through global memory inside the launch:
//   producer  the new 32 x 8 block of h is collected in LDS and stored write-through (sc1) by one wave
//             instruction of 16-byte stores (1 KB, whole 128-byte lines); that wave drains its stores
//             (s_waitcnt vmcnt(0)) and one lane of it adds 1 to the group's arrival counter;
//   consumer  one lane polls the counter with agent-scope loads until all H/8 arrivals of the step are
//             in, the workgroup passes a barrier, and then every wave fetches its part of h with
//             L1-bypassing (sc1) 16-byte loads only.
// Two buffers alternate: a workgroup can only write step t+1 after every workgroup of its group has
// written step t, i.e. has finished reading step t-1.  Groups never wait for each other.  Workgroups of a
// group are numbered eight apart so that they tend to share an XCD's L2 (speed only).  Every wait is
// bounded: after LSTM_SEQ_TIMEOUT_TICKS without progress a workgroup raises the launch's timeout word
// and leaves; the others see the word and leave too, so the grid always drains.  A workgroup that leaves
// fills what it has not yet produced of its outputs with NaN, so the failure is in the data (loss and
// gradients turn NaN) besides the counter below.  All workgroups of a launch must be resident together:
// the host asks the runtime how many fit the device (seq_resident_workgroups) and refuses larger grids.
// ------------------------------------------------------------------------------------------------
#define SEQ_WAVES 8
#define SEQ_MAX_CHAINS 2
#define SEQ_CNT_STRIDE 64                    // unsigned words between the counters of two groups (256 B)
#define LSTM_SEQ_TIMEOUT_TICKS 20000000LL    // of the 100 MHz wall clock: 200 ms

struct LstmSeqChain {
  const float* xproj;        // [T][B][4H] input projections without bias, or nullptr
  const float* w_rec;        // [4H][H]
  const float* b_a;          // [4H] or nullptr
  const float* b_b;          // [4H] or nullptr (comes with b_a)
  const float* h_init; int64_t ld_hinit;     // [B][..] or nullptr (zeros)
  const float* c_init; int64_t ld_cinit;     // [B][..] or nullptr (zeros)
  float* gates;              // [T][B][4H] activated gates, out
  float* hseq; int64_t ld_h; int64_t step_h; // h of step t at hseq + t*step_h + b*ld_h + j
  float* cseq;               // [T][B][H]
  int reverse;               // walk t = T-1 .. 0
};

struct LstmSeqArgs {
  LstmSeqChain ch[SEQ_MAX_CHAINS];
  int nchains;
  int T;
  int64_t B;
  int H;
  int groups_per_chain;      // ceil(B / 32)
  unsigned* sync;            // [0] timeout word, then one arrival counter per group, SEQ_CNT_STRIDE apart
  float* xch;                // exchange: [group][2][H/8 slices][32 rows][8]
  int debug_drop_slice;      // -1; test knob: workgroups of this slice leave at once, as if they never became resident
};

typedef __attribute__((address_space(1))) unsigned gu32_t;

__device__ __forceinline__ f32x4 load_x4_sc1(const float* p) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// 16-byte write-through store; the trailing nop keeps the data registers until the store has read them
__device__ __forceinline__ void store_x4_sc1(float* p, f32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}

// Arrival counters and the timeout word start every launch at zero.  A small kernel in front of the launch, not
// a memset: replayed from a captured graph, a memset node was seen to land while the kernel behind it was
// already running (workgroups then read a non-zero timeout word and gave up).
__global__ void seq_zero_kernel(unsigned* p, int n) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = 0u;
}
static int seq_clear(float* ws, int groups, hipStream_t stream) {
  hipLaunchKernelGGL(seq_zero_kernel, dim3(1), dim3(256), 0, stream, (unsigned*)ws, SEQ_CNT_STRIDE * (1 + groups));
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// Bounded waits that ran out, counted per GPU and never cleared by a launch (the per-launch word in the
// workspace is): read back by dvae_lstm_seq_timeouts() at a point where the caller synchronises anyway.
static __device__ unsigned g_seq_timeouts;

template <int KW>      // K per wave = H / 8: 16, 32 or 64
__global__ void __launch_bounds__(64 * SEQ_WAVES, SEQ_WAVES / 2) lstm_seq_fwd_kernel(LstmSeqArgs g) {
  __shared__ float red[SEQ_WAVES * 32 * KS_RED_LD];
  __shared__ __attribute__((aligned(16))) float hx[32 * 8];
  __shared__ int s_abort;
  constexpr int NQ = KW / 2;                 // MFMAs per wave and step; also units per lane half
  const int H = g.H, NS = H / 8;
  const int ngroups = g.nchains * g.groups_per_chain;
  // workgroups of one group are eight apart in blockIdx
  const int per8 = 8 * NS;
  const int group = (int)(blockIdx.x % 8) + 8 * (int)(blockIdx.x / per8);
  const int slice = (int)(blockIdx.x % per8) / 8;
  if (group >= ngroups || slice == g.debug_drop_slice) return;
  const LstmSeqChain& c = g.ch[group / g.groups_per_chain];
  const int64_t b0 = (int64_t)(group % g.groups_per_chain) * 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  unsigned* const tmo = g.sync;
  unsigned* const cnt = g.sync + SEQ_CNT_STRIDE * (1 + group);
  float* const xch = g.xch + (int64_t)group * 2 * H * 32;
  if (threadIdx.x == 0) s_abort = 0;

  // this wave's K range and this lane's units in it: unit(i) = kbase + NQ*hh + i
  const int kbase = wave * KW + NQ * hh;
  // recurrent weights -> registers: B operand of MFMA i is W[gate row of column l31][unit(i)]
  const int64_t wrow = (int64_t)(l31 >> 3) * H + slice * 8 + (l31 & 7);
  float wreg[NQ];
  {
    const float* wp = c.w_rec + wrow * H + kbase;
#pragma unroll
    for (int q = 0; q < NQ / 4; ++q) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(wp + 4 * q);
      wreg[4 * q + 0] = v[0]; wreg[4 * q + 1] = v[1]; wreg[4 * q + 2] = v[2]; wreg[4 * q + 3] = v[3];
    }
  }
  // cell-update role (threads 0..255): row bl, unit jj; the cell state stays in a register
  const int bl = threadIdx.x >> 3, jj = threadIdx.x & 7;
  const bool cell = threadIdx.x < 256;
  const int64_t b = b0 + bl, j = (int64_t)slice * 8 + jj;
  const bool bok = cell && b < g.B;
  float bias[4] = {0.f, 0.f, 0.f, 0.f};
  float cstate = 0.f;
  if (cell) {
    if (c.b_a) {
#pragma unroll
      for (int gt = 0; gt < 4; ++gt) bias[gt] = c.b_a[gt * H + j] + c.b_b[gt * H + j];
    }
    if (bok && c.c_init) cstate = c.c_init[b * c.ld_cinit + j];
  }
  const int64_t arow = b0 + l31;
  const bool arow_ok = arow < g.B;
  __syncthreads();

  for (int step = 0; step < g.T; ++step) {
    const int t = c.reverse ? g.T - 1 - step : step;
    // operands of the cell update that do not depend on the hand-off: requested before the wait
    float xp[4] = {0.f, 0.f, 0.f, 0.f};
    if (bok && c.xproj) {
      const float* xr = c.xproj + ((int64_t)t * g.B + b) * 4 * H + j;
#pragma unroll
      for (int gt = 0; gt < 4; ++gt) xp[gt] = xr[gt * H];
    }
    f32x4 av[NQ / 4];
    bool have_h = true;
    if (step == 0) {
      if (c.h_init && arow_ok) {
        const float* hp = c.h_init + arow * c.ld_hinit + kbase;
#pragma unroll
        for (int q = 0; q < NQ / 4; ++q) av[q] = *reinterpret_cast<const f32x4*>(hp + 4 * q);
      } else {
#pragma unroll
        for (int q = 0; q < NQ / 4; ++q) av[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        have_h = c.h_init != nullptr;
      }
    } else {
      if (threadIdx.x == 0) {
        const unsigned target = (unsigned)NS * (unsigned)step;
        const long long t0 = wall_clock64();
        bool ok = true;
        while (__hip_atomic_load((gu32_t*)cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
          __builtin_amdgcn_s_sleep(1);
          if (wall_clock64() - t0 > LSTM_SEQ_TIMEOUT_TICKS ||
              __hip_atomic_load((gu32_t*)tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
            ok = false;
            break;
          }
        }
        if (!ok) {
          __hip_atomic_store((gu32_t*)tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          atomicAdd(&g_seq_timeouts, 1u);
          s_abort = 1;
        }
      }
      __syncthreads();
      if (s_abort) {
        if (bok) {
          for (int s2 = step; s2 < g.T; ++s2) {
            const int t2 = c.reverse ? g.T - 1 - s2 : s2;
            c.hseq[(int64_t)t2 * c.step_h + b * c.ld_h + j] = __builtin_nanf("");
            c.cseq[((int64_t)t2 * g.B + b) * H + j] = __builtin_nanf("");
          }
        }
        return;
      }
      // h of the previous step: unit u lives in slice u/8 at [row][u%8]
      const float* src = xch + (int64_t)((step - 1) & 1) * H * 32;
#pragma unroll
      for (int q = 0; q < NQ / 4; ++q) {
        const int u = kbase + 4 * q;
        av[q] = load_x4_sc1(src + ((int64_t)(u >> 3) * 32 + l31) * 8 + (u & 7));
      }
      // the loads above are invisible to the compiler's wait counters: wait here, and make every use of
      // the registers depend on this statement
      if constexpr (NQ / 4 == 2)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(av[0]), "+v"(av[1]) : : "memory");
      else if constexpr (NQ / 4 == 4)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3]) : : "memory");
      else
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3]), "+v"(av[4]), "+v"(av[5]), "+v"(av[6]), "+v"(av[7])
                     : : "memory");
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (have_h) {
#pragma unroll
      for (int q = 0; q < NQ / 4; ++q) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][0], wreg[4 * q + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][1], wreg[4 * q + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][2], wreg[4 * q + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][3], wreg[4 * q + 3], acc, 0, 0, 0);
      }
    }
    ks_spill(red, acc, wave, lane);
    __syncthreads();
    float ig_o = 0.f, fg_o = 0.f, gg_o = 0.f, og_o = 0.f, hn_o = 0.f;
    if (cell) {
      float pre[4];
#pragma unroll
      for (int gt = 0; gt < 4; ++gt) {
        const int col = gt * 8 + jj;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < SEQ_WAVES; ++w) s += red[(w * 32 + bl) * KS_RED_LD + col];
        pre[gt] = (s + xp[gt]) + bias[gt];
      }
      const float ig = sigmoidf_(pre[0]), fg = sigmoidf_(pre[1]), gg = tanhf(pre[2]), og = sigmoidf_(pre[3]);
      cstate = fg * cstate + ig * gg;
      const float hn = og * tanhf(cstate);
      ig_o = ig; fg_o = fg; gg_o = gg; og_o = og; hn_o = hn;
      hx[bl * 8 + jj] = bok ? hn : 0.f;
    }
    if (step + 1 < g.T) {
      // the hand-off is the critical path.  The new 32 x 8 block of h goes to the group as ONE wave instruction
      // of 16-byte write-through stores (1 KB, whole lines), then one lane signals for the workgroup: arrivals
      // on one word are served one after the other, so there is one per workgroup and step.
      __syncthreads();
      if (wave == 0) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(&hx[lane * 4]);
        float* dst = xch + (int64_t)(step & 1) * H * 32 + (int64_t)slice * 256 + lane * 4;
        store_x4_sc1(dst, v);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add((gu32_t*)cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    // the step's own outputs behind it
    if (bok) {
      float* go = c.gates + ((int64_t)t * g.B + b) * 4 * H;
      go[0 * H + j] = ig_o;
      go[1 * H + j] = fg_o;
      go[2 * H + j] = gg_o;
      go[3 * H + j] = og_o;
      c.cseq[((int64_t)t * g.B + b) * H + j] = cstate;
      c.hseq[(int64_t)t * c.step_h + b * c.ld_h + j] = hn_o;
    }
    // (the next step's barrier behind the poll keeps `red` from being overwritten while it is read)
  }
}

extern "C" int64_t dvae_lstm_seq_ws_floats(int64_t nchains, int64_t B, int64_t H) {
  const int64_t groups = nchains * ceil_div64(B, 32);
  return SEQ_CNT_STRIDE * (1 + groups) + groups * 2 * H * 32;
}

extern "C" int dvae_lstm_seq_fwd_f32(int64_t nchains, const float* const* xproj, const float* const* w_rec,
                                     const float* const* b_a, const float* const* b_b, const float* const* h_init,
                                     int64_t ld_hinit, const float* const* c_init, int64_t ld_cinit,
                                     float* const* gates, float* const* hseq, int64_t ld_h, int64_t step_h,
                                     float* const* cseq, const int* reverse, int64_t T, int64_t B, int64_t H,
                                     float* ws, dvae_stream_t stream) {
  DVAE_REQUIRE(dvae_lstm_seq_supported(nchains, T, B, H),
               "lstm_seq_fwd: unsupported shape (chains=%ld T=%ld B=%ld H=%ld): H must be 128, 256 or 512 and "
               "chains*ceil(B/32)*(H/8) at most 512", (long)nchains, (long)T, (long)B, (long)H);
  DVAE_REQUIRE(ws != nullptr && (((uintptr_t)ws) & 255) == 0, "lstm_seq_fwd: workspace must be 256-byte aligned");
  LstmSeqArgs g;
  g.nchains = (int)nchains; g.T = (int)T; g.B = B; g.H = (int)H;
  g.groups_per_chain = (int)ceil_div64(B, 32);
  for (int i = 0; i < nchains; ++i) {
    LstmSeqChain& c = g.ch[i];
    c.xproj = xproj ? xproj[i] : nullptr;
    c.w_rec = w_rec[i];
    c.b_a = b_a ? b_a[i] : nullptr;
    c.b_b = b_b ? b_b[i] : nullptr;
    c.h_init = h_init ? h_init[i] : nullptr; c.ld_hinit = ld_hinit;
    c.c_init = c_init ? c_init[i] : nullptr; c.ld_cinit = ld_cinit;
    c.gates = gates[i]; c.hseq = hseq[i]; c.ld_h = ld_h; c.step_h = step_h; c.cseq = cseq[i];
    c.reverse = reverse ? reverse[i] : 0;
    DVAE_REQUIRE(c.w_rec && c.gates && c.hseq && c.cseq, "lstm_seq_fwd: null argument (chain %d)", i);
    DVAE_REQUIRE((c.b_a == nullptr) == (c.b_b == nullptr), "lstm_seq_fwd: biases come in pairs");
    DVAE_REQUIRE(aligned16(c.w_rec) && (c.h_init == nullptr || (aligned16(c.h_init) && ld_hinit % 4 == 0)),
                 "lstm_seq_fwd: w_rec and h_init must be 16-byte aligned, ld_hinit a multiple of 4");
  }
  const int groups = g.nchains * g.groups_per_chain;
  g.sync = (unsigned*)ws;
  g.xch = ws + SEQ_CNT_STRIDE * (1 + groups);
  {
    // test knob: the workgroups of one slice never show up, which is what a launch that is not fully resident
    // looks like to the others (they run into the bound of their first wait)
    const char* drop = getenv("DVAE_LSTM_SEQ_DEBUG_DROP_SLICE");
    g.debug_drop_slice = drop ? atoi(drop) : -1;
  }
  if (seq_clear(ws, groups, ST(stream)) != 0) {
    dvae_set_error("lstm_seq_fwd: clearing the counters failed: %s", hipGetErrorString(hipGetLastError()));
    return -EIO;
  }
  const int NS = (int)(H / 8);
  const unsigned nblocks = (unsigned)(8 * NS * ((groups + 7) / 8));
  DVAE_WORK(2.0 * (double)nchains * T * B * 4 * H * H, DVAE_UNIT_FLOP);
  if (H == 128) DVAE_LAUNCH((lstm_seq_fwd_kernel<16>), dim3(nblocks), dim3(64 * SEQ_WAVES), 0, ST(stream), g);
  else if (H == 256) DVAE_LAUNCH((lstm_seq_fwd_kernel<32>), dim3(nblocks), dim3(64 * SEQ_WAVES), 0, ST(stream), g);
  else DVAE_LAUNCH((lstm_seq_fwd_kernel<64>), dim3(nblocks), dim3(64 * SEQ_WAVES), 0, ST(stream), g);
  DVAE_LAUNCH_CHECK("lstm_seq_fwd");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Whole recurrence in ONE launch (backward through time).
//
// Same partition as the forward launch: a group of 32 batch rows is served by H/8 workgroups, and a
// workgroup owns eight hidden units: their gate gradients of every step, their carried cell gradient
// (in a register) and the 32 rows of the recurrent weights that belong to their gates (in registers, as
// MFMA B operands).  The hidden-state gradient of the previous step is dh += dG * W, a sum over ALL gate
// columns, so here the workgroups exchange partial products instead of states: after the pointwise
// step a workgroup multiplies its 32 x 32 block of dG (through LDS) with its 32 x H weight rows and
// writes the 32 x H partial, ordered by destination workgroup, to the exchange area; the owner of a
// hidden-unit slice adds the H/8 partials of its slice in workgroup order (fixed order: run-to-run
// identical bits) to the gradient that arrives from outside.  Arrival counters, the two alternating
// exchange buffers, the bounded waits and the residency requirement are those of the forward launch.
// ------------------------------------------------------------------------------------------------
#define SEQ_A_LD 36

struct LstmSeqBwdChain {
  const float* w_rec;        // [4H][H]
  const float* gates;        // [T][B][4H] activated gates of the forward pass
  const float* cseq;         // [T][B][H]
  const float* c_init; int64_t ld_cinit;     // cell state before the chain's first step, or nullptr (zeros)
  const float* dh_ext; int64_t ld_dh; int64_t step_dh;   // gradient reaching h of step t from outside the chain
  float* dgates;             // [T][B][4H] gradient wrt the gate pre-activations, out
  float* dh_init; int64_t ld_dhinit;         // gradient wrt the hidden state before the first step, or nullptr
  float* dc_init; int64_t ld_dcinit;         // gradient wrt the cell state before the first step (with dh_init)
  int reverse;               // the forward pass walked t = T-1 .. 0
};

struct LstmSeqBwdArgs {
  LstmSeqBwdChain ch[SEQ_MAX_CHAINS];
  int nchains;
  int T;
  int64_t B;
  int groups_per_chain;
  int accumulate_dh_init;    // dh_init += instead of =
  unsigned* sync;
  float* xch;                // [group][2][writer H/8][destination H/8][32 rows][8]
};

typedef __attribute__((address_space(1))) float gf32_t;

// thread 0 waits until the group's counter has reached `target`; returns false for the whole workgroup
// when the wait ran into its bound or another workgroup raised the timeout word
__device__ __forceinline__ bool seq_wait(unsigned* cnt, unsigned* tmo, unsigned target, int* s_abort) {
  if (threadIdx.x == 0) {
    const long long t0 = wall_clock64();
    bool ok = true;
    while (__hip_atomic_load((gu32_t*)cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (wall_clock64() - t0 > LSTM_SEQ_TIMEOUT_TICKS ||
          __hip_atomic_load((gu32_t*)tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
        ok = false;
        break;
      }
    }
    if (!ok) {
      __hip_atomic_store((gu32_t*)tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      atomicAdd(&g_seq_timeouts, 1u);
      *s_abort = 1;
    }
  }
  __syncthreads();
  return *s_abort == 0;
}

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#define SEQ_AUX_SC1 16      // cache-policy bits of a raw buffer access: write-through store / L1-bypassing load

template <int HH>      // hidden size: 128, 256 or 512
__global__ void __launch_bounds__(64 * SEQ_WAVES, SEQ_WAVES / 2) lstm_seq_bwd_kernel(LstmSeqBwdArgs g) {
  __shared__ __attribute__((aligned(16))) float As[32 * SEQ_A_LD];                 // dG block, MFMA A operand
  __shared__ __attribute__((aligned(16))) float Ts[SEQ_WAVES * 32 * SEQ_A_LD];     // per wave: product tile on its way out
  __shared__ __attribute__((aligned(16))) float red[SEQ_WAVES * 32 * 8];           // per wave: partial sums on their way in
  __shared__ int s_abort;
  constexpr int H = HH, NS = HH / 8;
  constexpr int NTILE = HH / 32;                            // 32-column tiles of the partial product
  constexpr int NT = (NTILE + SEQ_WAVES - 1) / SEQ_WAVES;   // tiles per wave
  constexpr int NW = NS / SEQ_WAVES;                        // writers whose partials one wave fetches
  const int ngroups = g.nchains * g.groups_per_chain;
  const int per8 = 8 * NS;
  const int group = (int)(blockIdx.x % 8) + 8 * (int)(blockIdx.x / per8);
  const int slice = (int)(blockIdx.x % per8) / 8;
  if (group >= ngroups) return;
  const LstmSeqBwdChain& c = g.ch[group / g.groups_per_chain];
  const int64_t b0 = (int64_t)(group % g.groups_per_chain) * 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  unsigned* const tmo = g.sync;
  unsigned* const cnt = g.sync + SEQ_CNT_STRIDE * (1 + group);
  // the group's exchange area [2][writer][destination slice][32 rows][8], addressed through a buffer
  // descriptor: 16-byte write-through stores and L1-bypassing loads that the compiler counts itself
  constexpr unsigned XBUF = (unsigned)NS * 32u * H;         // floats per buffer
  const auto xrsrc = __builtin_amdgcn_make_buffer_rsrc(g.xch + (int64_t)group * 2 * XBUF, 0, 2 * XBUF * 4, 0x00020000);
  if (threadIdx.x == 0) s_abort = 0;

  // weight rows of this workgroup's 32 gate columns: local k = 8*gate + unit; lane half hh feeds k = 16*hh + i
  // to MFMA i, as row (k/8)*H + 8*slice + k%8 of w_rec and column 32*tile + l31
  float wreg[NT][16];
#pragma unroll
  for (int tt = 0; tt < NT; ++tt) {
    const int tile = wave + SEQ_WAVES * tt;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int k = 16 * hh + i;
      wreg[tt][i] = tile < NTILE ? c.w_rec[((int64_t)(k >> 3) * H + slice * 8 + (k & 7)) * H + tile * 32 + l31] : 0.f;
    }
  }
  const int bl = threadIdx.x >> 3, jj = threadIdx.x & 7;
  const bool cell = threadIdx.x < 256;
  const int64_t b = b0 + bl, j = (int64_t)slice * 8 + jj;
  const bool bok = cell && b < g.B;
  const int xrow = lane >> 1, xhalf = lane & 1;              // exchange role: 16 bytes of row xrow
  float dcarry = 0.f;
  __syncthreads();

  // sum over all writers of this slice's 32 x 8 block in `buf`: every wave fetches NW writers' blocks
  // (one 1-KB block per wave instruction), adds them in writer order and leaves its sum in `red`
  auto gather = [&](int buf) {
    u32x4 v[NW];
#pragma unroll
    for (int q = 0; q < NW; ++q) {
      const unsigned w = (unsigned)(wave * NW + q);
      const unsigned off = ((unsigned)buf * XBUF + w * 32u * H + (unsigned)slice * 256u + (unsigned)lane * 4u) * 4u;
      v[q] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, off, 0, SEQ_AUX_SC1);
    }
    f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < NW; ++q) s += __builtin_bit_cast(f32x4, v[q]);
    *reinterpret_cast<f32x4*>(&red[(wave * 32 + xrow) * 8 + xhalf * 4]) = s;
    __syncthreads();
  };
  auto gathered = [&]() {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < SEQ_WAVES; ++w) s += red[(w * 32 + bl) * 8 + jj];
    return s;
  };

  for (int step = 0; step < g.T; ++step) {
    const int t = c.reverse ? step : g.T - 1 - step;
    const bool has_prev = step + 1 < g.T;
    const int tp = c.reverse ? t + 1 : t - 1;
    // everything the pointwise step needs besides the exchanged sum: requested before the wait
    float ig = 0.f, fg = 0.f, gg = 0.f, og = 0.f, cv = 0.f, cpv = 0.f, dhv = 0.f;
    if (bok) {
      const float* gt = c.gates + ((int64_t)t * g.B + b) * 4 * H + j;
      ig = gt[0]; fg = gt[H]; gg = gt[2 * H]; og = gt[3 * H];
      cv = c.cseq[((int64_t)t * g.B + b) * H + j];
      if (has_prev) cpv = c.cseq[((int64_t)tp * g.B + b) * H + j];
      else if (c.c_init) cpv = c.c_init[b * c.ld_cinit + j];
      dhv = c.dh_ext[(int64_t)t * c.step_dh + b * c.ld_dh + j];
    }
    if (step > 0) {
      if (!seq_wait(cnt, tmo, (unsigned)NS * (unsigned)step, &s_abort)) {
        if (bok) {
          for (int s2 = step; s2 < g.T; ++s2) {
            const int t2 = c.reverse ? s2 : g.T - 1 - s2;
            float* dg = c.dgates + ((int64_t)t2 * g.B + b) * 4 * H + j;
            dg[0] = dg[H] = dg[2 * H] = dg[3 * H] = __builtin_nanf("");
          }
          if (c.dh_init) c.dh_init[b * c.ld_dhinit + j] = __builtin_nanf("");
          if (c.dc_init) c.dc_init[b * c.ld_dcinit + j] = __builtin_nanf("");
        }
        return;
      }
      gather((step - 1) & 1);
      if (cell) dhv += gathered();
    }
    float dg0 = 0.f, dg1 = 0.f, dg2 = 0.f, dg3 = 0.f;
    if (cell) {
      const float tc = tanhf(cv);
      const float dct = dcarry + dhv * og * (1.f - tc * tc);
      dg0 = dct * gg * ig * (1.f - ig);
      dg1 = dct * cpv * fg * (1.f - fg);
      dg2 = dct * ig * (1.f - gg * gg);
      dg3 = dhv * tc * og * (1.f - og);
      dcarry = dct * fg;
      if (!bok) dg0 = dg1 = dg2 = dg3 = 0.f;
      As[bl * SEQ_A_LD + 0 * 8 + jj] = dg0;
      As[bl * SEQ_A_LD + 1 * 8 + jj] = dg1;
      As[bl * SEQ_A_LD + 2 * 8 + jj] = dg2;
      As[bl * SEQ_A_LD + 3 * 8 + jj] = dg3;
    }
    const bool product = has_prev || c.dh_init != nullptr;
    if (product) {
      __syncthreads();
      f32x4 av[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) av[q] = *reinterpret_cast<const f32x4*>(&As[l31 * SEQ_A_LD + 16 * hh + 4 * q]);
      float* const tsw = Ts + wave * 32 * SEQ_A_LD;
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        const int tile = wave + SEQ_WAVES * tt;
        if (tile < NTILE) {
          f32x16 acc;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][0], wreg[tt][4 * q + 0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][1], wreg[tt][4 * q + 1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][2], wreg[tt][4 * q + 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][3], wreg[tt][4 * q + 3], acc, 0, 0, 0);
          }
          // through the wave's own LDS tile into destination order: columns 8q..8q+7 of the tile belong to
          // destination slice 4*tile + q, whose [32 rows][8] block is 1 KB = one 16-byte store per lane
#pragma unroll
          for (int r = 0; r < 16; ++r) tsw[((r & 3) + 8 * (r >> 2) + 4 * hh) * SEQ_A_LD + l31] = acc[r];
          // (same wave wrote and reads: LDS operations of a wave complete in order)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(&tsw[xrow * SEQ_A_LD + q * 8 + xhalf * 4]);
            const unsigned off = ((unsigned)(step & 1) * XBUF + (unsigned)slice * 32u * H +
                                  (unsigned)(tile * 4 + q) * 256u + (unsigned)lane * 4u) * 4u;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), xrsrc, off, 0, SEQ_AUX_SC1);
          }
        }
      }
      // one arrival per workgroup (arrivals on one word are served one after the other): every wave drains
      // its stores, the workgroup meets, one lane signals
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_fetch_add((gu32_t*)cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // the step's own output goes out behind the hand-off
    if (bok) {
      float* dg = c.dgates + ((int64_t)t * g.B + b) * 4 * H + j;
      dg[0] = dg0; dg[H] = dg1; dg[2 * H] = dg2; dg[3 * H] = dg3;
    }
    // (As and red are rewritten only behind the next step's wait, whose barrier every wave reaches after its reads)
  }
  if (c.dh_init != nullptr) {
    if (!seq_wait(cnt, tmo, (unsigned)NS * (unsigned)g.T, &s_abort)) {
      if (bok) {
        c.dh_init[b * c.ld_dhinit + j] = __builtin_nanf("");
        if (c.dc_init) c.dc_init[b * c.ld_dcinit + j] = __builtin_nanf("");
      }
      return;
    }
    gather((g.T - 1) & 1);
    if (bok) {
      const float s = gathered();
      float* dst = c.dh_init + b * c.ld_dhinit + j;
      *dst = g.accumulate_dh_init ? *dst + s : s;
      if (c.dc_init) c.dc_init[b * c.ld_dcinit + j] = dcarry;
    }
  }
}

extern "C" int64_t dvae_lstm_seq_bwd_ws_floats(int64_t nchains, int64_t B, int64_t H) {
  const int64_t groups = nchains * ceil_div64(B, 32);
  return SEQ_CNT_STRIDE * (1 + groups) + groups * 2 * (H / 8) * 32 * H;
}

extern "C" int dvae_lstm_seq_bwd_f32(int64_t nchains, const float* const* w_rec, const float* const* gates,
                                     const float* const* cseq, const float* const* c_init, int64_t ld_cinit,
                                     const float* const* dh_ext, int64_t ld_dh, int64_t step_dh,
                                     float* const* dgates, float* const* dh_init, int64_t ld_dhinit,
                                     float* const* dc_init, int64_t ld_dcinit, int accumulate_dh_init,
                                     const int* reverse, int64_t T, int64_t B, int64_t H, float* ws,
                                     dvae_stream_t stream) {
  DVAE_REQUIRE(dvae_lstm_seq_supported(nchains, T, B, H),
               "lstm_seq_bwd: unsupported shape (chains=%ld T=%ld B=%ld H=%ld): H must be 128, 256 or 512 and "
               "chains*ceil(B/32)*(H/8) at most 512", (long)nchains, (long)T, (long)B, (long)H);
  DVAE_REQUIRE(ws != nullptr && (((uintptr_t)ws) & 255) == 0, "lstm_seq_bwd: workspace must be 256-byte aligned");
  DVAE_REQUIRE(w_rec && gates && cseq && dh_ext && dgates, "lstm_seq_bwd: null argument");
  LstmSeqBwdArgs g;
  g.nchains = (int)nchains; g.T = (int)T; g.B = B;
  g.groups_per_chain = (int)ceil_div64(B, 32);
  g.accumulate_dh_init = accumulate_dh_init;
  for (int i = 0; i < nchains; ++i) {
    LstmSeqBwdChain& c = g.ch[i];
    c.w_rec = w_rec[i]; c.gates = gates[i]; c.cseq = cseq[i];
    c.c_init = c_init ? c_init[i] : nullptr; c.ld_cinit = ld_cinit;
    c.dh_ext = dh_ext[i]; c.ld_dh = ld_dh; c.step_dh = step_dh;
    c.dgates = dgates[i];
    c.dh_init = dh_init ? dh_init[i] : nullptr; c.ld_dhinit = ld_dhinit;
    c.dc_init = dc_init ? dc_init[i] : nullptr; c.ld_dcinit = ld_dcinit;
    c.reverse = reverse ? reverse[i] : 0;
    DVAE_REQUIRE(c.w_rec && c.gates && c.cseq && c.dh_ext && c.dgates, "lstm_seq_bwd: null argument (chain %d)", i);
    DVAE_REQUIRE(c.dc_init == nullptr || c.dh_init != nullptr, "lstm_seq_bwd: dc_init comes with dh_init");
  }
  const int groups = g.nchains * g.groups_per_chain;
  g.sync = (unsigned*)ws;
  g.xch = ws + SEQ_CNT_STRIDE * (1 + groups);
  if (seq_clear(ws, groups, ST(stream)) != 0) {
    dvae_set_error("lstm_seq_bwd: clearing the counters failed: %s", hipGetErrorString(hipGetLastError()));
    return -EIO;
  }
  const int NS = (int)(H / 8);
  const unsigned nblocks = (unsigned)(8 * NS * ((groups + 7) / 8));
  DVAE_WORK(2.0 * (double)nchains * T * B * 4 * H * H, DVAE_UNIT_FLOP);
  if (H == 128) DVAE_LAUNCH((lstm_seq_bwd_kernel<128>), dim3(nblocks), dim3(64 * SEQ_WAVES), 0, ST(stream), g);
  else if (H == 256) DVAE_LAUNCH((lstm_seq_bwd_kernel<256>), dim3(nblocks), dim3(64 * SEQ_WAVES), 0, ST(stream), g);
  else DVAE_LAUNCH((lstm_seq_bwd_kernel<512>), dim3(nblocks), dim3(64 * SEQ_WAVES), 0, ST(stream), g);
  DVAE_LAUNCH_CHECK("lstm_seq_bwd");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Stacked autoregressive decoder (two LSTM layers, models/decoders.py:110-137) in ONE launch, forward.
//
// Cell (t, l) computes   pre = x W_ih_l^T + h_{l,t-1} W_hh_l^T + b,   x = h_{1,t-1} for layer 0 (the top layer's
// previous output; the learned first input at t = 0) and x = h_{0,t} for layer 1.  The cells form one chain
// (0,0) (0,1) (1,0) (1,1) ...; partition, hand-off and residency are those of lstm_seq_fwd_kernel: a workgroup
// owns eight hidden units of BOTH layers, keeps the four 32 x H weight slices in registers (64 per lane at
// H = 256) and both cell states in registers, and every cell publishes its 32 x 8 block of h to the group
// (one buffer pair per layer) behind one arrival counter that counts cells.
// ------------------------------------------------------------------------------------------------
#define DEC_L 2

struct LstmDecArgs {
  const float* w_ih[DEC_L]; const float* w_hh[DEC_L];     // [4H][H] each
  const float* b_ih[DEC_L]; const float* b_hh[DEC_L];     // [4H] each
  const float* h0; int64_t ld_h0;                         // [B][L*H]: layer l starts from columns l*H ..
  const float* c0; int64_t ld_c0;
  const float* init;                                      // [H] first input of layer 0, the same for every row
  float* gates;                                           // [L][T][B][4H]
  float* hs;                                              // [L][T][B][H]
  float* cs;                                              // [L][T][B][H]
  int T;
  int64_t B;
  int H;
  int groups;
  unsigned* sync;
  float* xch;                                             // [group][L][2][H/8 slices][32 rows][8]
};

template <int KW>      // K per wave and operand = H / 8: 16 or 32
__global__ void __launch_bounds__(64 * SEQ_WAVES) lstm_dec_fwd_kernel(LstmDecArgs g) {
  __shared__ float red[SEQ_WAVES * 32 * KS_RED_LD];
  __shared__ __attribute__((aligned(16))) float hx[32 * 8];
  __shared__ int s_abort;
  constexpr int NQ = KW / 2;
  const int H = g.H, NS = H / 8;
  const int per8 = 8 * NS;
  const int group = (int)(blockIdx.x % 8) + 8 * (int)(blockIdx.x / per8);
  const int slice = (int)(blockIdx.x % per8) / 8;
  if (group >= g.groups) return;
  const int64_t b0 = (int64_t)group * 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  unsigned* const tmo = g.sync;
  unsigned* const cnt = g.sync + SEQ_CNT_STRIDE * (1 + group);
  float* const xch = g.xch + (int64_t)group * DEC_L * 2 * H * 32;
  if (threadIdx.x == 0) s_abort = 0;

  const int kbase = wave * KW + NQ * hh;
  const int64_t wrow = (int64_t)(l31 >> 3) * H + slice * 8 + (l31 & 7);
  float wx[DEC_L][NQ], wh[DEC_L][NQ];
#pragma unroll
  for (int l = 0; l < DEC_L; ++l)
#pragma unroll
    for (int q = 0; q < NQ / 4; ++q) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(g.w_ih[l] + wrow * H + kbase + 4 * q);
      const f32x4 b = *reinterpret_cast<const f32x4*>(g.w_hh[l] + wrow * H + kbase + 4 * q);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        wx[l][4 * q + e] = a[e];
        wh[l][4 * q + e] = b[e];
      }
    }
  const int bl = threadIdx.x >> 3, jj = threadIdx.x & 7;
  const bool cell = threadIdx.x < 256;
  const int64_t b = b0 + bl, j = (int64_t)slice * 8 + jj;
  const bool bok = cell && b < g.B;
  float bias[DEC_L][4], cstate[DEC_L];
#pragma unroll
  for (int l = 0; l < DEC_L; ++l) {
#pragma unroll
    for (int gt = 0; gt < 4; ++gt) bias[l][gt] = cell ? g.b_ih[l][gt * H + j] + g.b_hh[l][gt * H + j] : 0.f;
    cstate[l] = bok ? g.c0[b * g.ld_c0 + l * H + j] : 0.f;
  }
  const int64_t arow = b0 + l31;
  const bool arow_ok = arow < g.B;
  const int64_t arc = arow_ok ? arow : g.B - 1;
  __syncthreads();

  const int ncells = g.T * DEC_L;
  // The recurrent half of a cell's product does not wait for the cell before it: cell s needs the output of cell
  // s-1 as its input x, and that same tensor is the previous hidden state of cell s+1 (layer 1 at step t follows
  // layer 0 at step t, whose x was h_{1,t-1}; layer 0 at step t+1 follows layer 1 at step t, whose x was h_{0,t}).
  // So a cell multiplies x with its W_ih on the critical path and, behind its hand-off, the same x with the NEXT
  // cell's W_hh; only the two cells of step 0 take their previous state from h0.
  auto mul = [&](const f32x4 (&a)[NQ / 4], const float (&w)[NQ], f32x16 acc) {
#pragma unroll
    for (int q = 0; q < NQ / 4; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][e], w[4 * q + e], acc, 0, 0, 0);
    return acc;
  };
  auto h0_rows = [&](int l, f32x4 (&a)[NQ / 4]) {
#pragma unroll
    for (int q = 0; q < NQ / 4; ++q) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(g.h0 + arc * g.ld_h0 + l * H + kbase + 4 * q);
      a[q] = arow_ok ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };
  f32x16 zero16;
#pragma unroll
  for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
  f32x16 rec;                                   // h_prev W_hh^T of the cell about to run
  {
    f32x4 ah[NQ / 4];
    h0_rows(0, ah);
    rec = mul(ah, wh[0], zero16);
  }
#pragma unroll 1
  for (int t = 0; t < g.T; ++t) {
#pragma unroll
    for (int l = 0; l < DEC_L; ++l) {
      const int s = t * DEC_L + l;
      if (s > 0) {
        if (!seq_wait(cnt, tmo, (unsigned)NS * (unsigned)s, &s_abort)) {
          if (bok) {
            for (int s2 = s; s2 < ncells; ++s2) {
              const int64_t row2 = ((int64_t)(s2 % DEC_L) * g.T + s2 / DEC_L) * g.B + b;
              g.hs[row2 * H + j] = __builtin_nanf("");
              g.cs[row2 * H + j] = __builtin_nanf("");
            }
          }
          return;
        }
      }
      // x: the cell below, or the top layer's previous output (the learned first input at the very start)
      f32x4 ax[NQ / 4];
      if (l == 0 && t == 0) {
#pragma unroll
        for (int q = 0; q < NQ / 4; ++q) ax[q] = *reinterpret_cast<const f32x4*>(g.init + kbase + 4 * q);
      } else {
        const float* src = xch + (int64_t)((l == 0 ? (DEC_L - 1) * 2 + ((t - 1) & 1) : (l - 1) * 2 + (t & 1))) * H * 32;
#pragma unroll
        for (int q = 0; q < NQ / 4; ++q) {
          const int u = kbase + 4 * q;
          ax[q] = load_x4_sc1(src + ((int64_t)(u >> 3) * 32 + l31) * 8 + (u & 7));
        }
        // the write-through-coherent loads above are invisible to the compiler's wait counters
        if constexpr (NQ / 4 == 2)
          asm volatile("s_waitcnt vmcnt(0)" : "+v"(ax[0]), "+v"(ax[1]) : : "memory");
        else
          asm volatile("s_waitcnt vmcnt(0)" : "+v"(ax[0]), "+v"(ax[1]), "+v"(ax[2]), "+v"(ax[3]) : : "memory");
      }
      const f32x16 acc = mul(ax, wx[l], rec);
      ks_spill(red, acc, wave, lane);
      __syncthreads();
      float ig_o = 0.f, fg_o = 0.f, gg_o = 0.f, og_o = 0.f, hn_o = 0.f;
      if (cell) {
        float pre[4];
#pragma unroll
        for (int gt = 0; gt < 4; ++gt) {
          const int col = gt * 8 + jj;
          float sum = 0.f;
#pragma unroll
          for (int w = 0; w < SEQ_WAVES; ++w) sum += red[(w * 32 + bl) * KS_RED_LD + col];
          pre[gt] = sum + bias[l][gt];
        }
        ig_o = sigmoidf_(pre[0]); fg_o = sigmoidf_(pre[1]); gg_o = tanhf(pre[2]); og_o = sigmoidf_(pre[3]);
        cstate[l] = fg_o * cstate[l] + ig_o * gg_o;
        hn_o = og_o * tanhf(cstate[l]);
        hx[bl * 8 + jj] = bok ? hn_o : 0.f;
      }
      if (s + 1 < ncells) {
        __syncthreads();
        if (wave == 0) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(&hx[lane * 4]);
          store_x4_sc1(xch + (int64_t)(l * 2 + (t & 1)) * H * 32 + (int64_t)slice * 256 + lane * 4, v);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (lane == 0) __hip_atomic_fetch_add((gu32_t*)cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      if (bok) {
        const int64_t row = ((int64_t)l * g.T + t) * g.B + b;
        float* go = g.gates + row * 4 * H;
        go[0 * H + j] = ig_o;
        go[1 * H + j] = fg_o;
        go[2 * H + j] = gg_o;
        go[3 * H + j] = og_o;
        g.cs[row * H + j] = cstate[l];
        g.hs[row * H + j] = hn_o;
      }
      // behind the hand-off: the recurrent half of the next cell (layer 1 - l)
      if (s + 1 < ncells) {
        if (s == 0) {
          f32x4 ah[NQ / 4];
          h0_rows(1, ah);
          rec = mul(ah, wh[1], zero16);
        } else {
          rec = mul(ax, wh[DEC_L - 1 - l], zero16);
        }
      }
    }
  }
}

extern "C" int64_t dvae_lstm_dec2_ws_floats(int64_t B, int64_t H) {
  const int64_t groups = ceil_div64(B, 32);
  // counters, then the larger of the forward (hidden states) and backward (partial products) exchange areas
  return SEQ_CNT_STRIDE * (1 + groups) + groups * DEC_L * 2 * 2 * (H / 8) * 32 * H;
}

extern "C" int dvae_lstm_dec2_fwd_f32(const float* const* w_ih, const float* const* w_hh, const float* const* b_ih,
                                      const float* const* b_hh, const float* h0, int64_t ld_h0, const float* c0,
                                      int64_t ld_c0, const float* init, float* gates, float* hs, float* cs, int64_t T,
                                      int64_t B, int64_t H, float* ws, dvae_stream_t stream) {
  DVAE_REQUIRE(dvae_lstm_dec2_supported(T, B, H),
               "lstm_dec2_fwd: unsupported shape (T=%ld B=%ld H=%ld): H must be 128 or 256 and ceil(B/32)*(H/8) at most 256",
               (long)T, (long)B, (long)H);
  DVAE_REQUIRE(w_ih && w_hh && b_ih && b_hh && h0 && c0 && init && gates && hs && cs, "lstm_dec2_fwd: null argument");
  DVAE_REQUIRE(ws != nullptr && (((uintptr_t)ws) & 255) == 0, "lstm_dec2_fwd: workspace must be 256-byte aligned");
  DVAE_REQUIRE(aligned16(h0) && ld_h0 % 4 == 0 && aligned16(init), "lstm_dec2_fwd: h0 and init must be 16-byte aligned");
  LstmDecArgs g;
  for (int l = 0; l < DEC_L; ++l) {
    g.w_ih[l] = w_ih[l]; g.w_hh[l] = w_hh[l]; g.b_ih[l] = b_ih[l]; g.b_hh[l] = b_hh[l];
    DVAE_REQUIRE(g.w_ih[l] && g.w_hh[l] && g.b_ih[l] && g.b_hh[l] && aligned16(g.w_ih[l]) && aligned16(g.w_hh[l]),
                 "lstm_dec2_fwd: weights of layer %d missing or not 16-byte aligned", l);
  }
  g.h0 = h0; g.ld_h0 = ld_h0; g.c0 = c0; g.ld_c0 = ld_c0; g.init = init;
  g.gates = gates; g.hs = hs; g.cs = cs; g.T = (int)T; g.B = B; g.H = (int)H;
  g.groups = (int)ceil_div64(B, 32);
  g.sync = (unsigned*)ws;
  g.xch = ws + SEQ_CNT_STRIDE * (1 + g.groups);
  if (seq_clear(ws, g.groups, ST(stream)) != 0) {
    dvae_set_error("lstm_dec2_fwd: clearing the counters failed: %s", hipGetErrorString(hipGetLastError()));
    return -EIO;
  }
  const int NS = (int)(H / 8);
  const unsigned nblocks = (unsigned)(8 * NS * ((g.groups + 7) / 8));
  DVAE_WORK(2.0 * (double)DEC_L * T * B * 4 * H * 2 * H, DVAE_UNIT_FLOP);
  if (H == 128) DVAE_LAUNCH((lstm_dec_fwd_kernel<16>), dim3(nblocks), dim3(64 * SEQ_WAVES), 0, ST(stream), g);
  else DVAE_LAUNCH((lstm_dec_fwd_kernel<32>), dim3(nblocks), dim3(64 * SEQ_WAVES), 0, ST(stream), g);
  DVAE_LAUNCH_CHECK("lstm_dec2_fwd");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Stacked decoder, backward through the chain of cells in ONE launch: (T-1,1) (T-1,0) (T-2,1) ... (0,0).
//
// A hidden state h_{l,t} receives at most two products: A, the input path of the cell it feeds as x (layer 1 at
// the same step for l = 0; layer 0 at the next step for the top layer), and B, its own layer's recurrent path
// from step t+1.  Each cell therefore publishes two 32 x H partial products (dG W_ih_l and dG W_hh_l, as in
// lstm_seq_bwd_kernel) into the buffers of their destination cells -- kind (A/B) x destination layer x parity of
// the destination step -- and gathers its own two before its pointwise step.  What leaves the chain at its end
// (destination step -1) is the gradient of the initial hidden states and of the first input.
// ------------------------------------------------------------------------------------------------
struct LstmDecBwdArgs {
  const float* w_ih[DEC_L]; const float* w_hh[DEC_L];     // [4H][H]
  const float* gates;                                     // [L][T][B][4H]
  const float* cs;                                        // [L][T][B][H]
  const float* c0; int64_t ld_c0;                         // [B][L*H]
  const float* dstates; int64_t ld_ds; int64_t step_ds;   // gradient reaching h_{L-1,t}[b][j]: + t*step_ds + b*ld_ds + j
  float* dgates;                                          // [L][T][B][4H]
  float* dh0; int64_t ld_dh0;                             // [B][L*H]
  float* dc0; int64_t ld_dc0;                             // [B][L*H]
  float* dx0;                                             // [B][H]: gradient wrt layer 0's first input, per row
  int T;
  int64_t B;
  int groups;
  unsigned* sync;
  float* xch;                                             // [group][kind][layer][parity][writer][dest][32][8]
};

template <int HH>      // hidden size: 128 or 256 (one 32-column tile per wave at most)
__global__ void __launch_bounds__(64 * SEQ_WAVES) lstm_dec_bwd_kernel(LstmDecBwdArgs g) {
  __shared__ __attribute__((aligned(16))) float As[32 * SEQ_A_LD];
  __shared__ __attribute__((aligned(16))) float Ts[SEQ_WAVES * 32 * SEQ_A_LD];
  __shared__ __attribute__((aligned(16))) float red[SEQ_WAVES * 32 * 8];
  __shared__ int s_abort;
  constexpr int H = HH, NS = HH / 8, NTILE = HH / 32, NW = NS / SEQ_WAVES;
  static_assert(NTILE <= SEQ_WAVES, "one product tile per wave");
  const int per8 = 8 * NS;
  const int group = (int)(blockIdx.x % 8) + 8 * (int)(blockIdx.x / per8);
  const int slice = (int)(blockIdx.x % per8) / 8;
  if (group >= g.groups) return;
  const int64_t b0 = (int64_t)group * 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  unsigned* const tmo = g.sync;
  unsigned* const cnt = g.sync + SEQ_CNT_STRIDE * (1 + group);
  constexpr unsigned XBUF = (unsigned)NS * 32u * H;         // floats per buffer
  constexpr unsigned NBUF = 2 * DEC_L * 2;
  const auto xrsrc = __builtin_amdgcn_make_buffer_rsrc(g.xch + (int64_t)group * NBUF * XBUF, 0, NBUF * XBUF * 4, 0x00020000);
  auto buf_of = [](int kind, int layer, int parity) { return (unsigned)((kind * DEC_L + layer) * 2 + (parity & 1)); };
  if (threadIdx.x == 0) s_abort = 0;

  // weight rows of this workgroup's 32 gate columns, both matrices of both layers (see lstm_seq_bwd_kernel)
  const bool has_tile = wave < NTILE;
  float wx[DEC_L][16], wh[DEC_L][16];
#pragma unroll
  for (int l = 0; l < DEC_L; ++l)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int k = 16 * hh + i;
      const int64_t off = ((int64_t)(k >> 3) * H + slice * 8 + (k & 7)) * H + wave * 32 + l31;
      wx[l][i] = has_tile ? g.w_ih[l][off] : 0.f;
      wh[l][i] = has_tile ? g.w_hh[l][off] : 0.f;
    }
  const int bl = threadIdx.x >> 3, jj = threadIdx.x & 7;
  const bool cell = threadIdx.x < 256;
  const int64_t b = b0 + bl, j = (int64_t)slice * 8 + jj;
  const bool bok = cell && b < g.B;
  const int xrow = lane >> 1, xhalf = lane & 1;
  float dcarry[DEC_L] = {0.f, 0.f};
  __syncthreads();

  // sum over all writers of this slice's 32 x 8 block of up to two buffers (fixed order), left in `red`
  auto gather = [&](bool use_a, unsigned buf_a, bool use_b, unsigned buf_b) {
    u32x4 va[NW], vb[NW];
#pragma unroll
    for (int q = 0; q < NW; ++q) {
      const unsigned w = (unsigned)(wave * NW + q);
      const unsigned off = (w * 32u * H + (unsigned)slice * 256u + (unsigned)lane * 4u) * 4u;
      va[q] = use_a ? __builtin_amdgcn_raw_buffer_load_b128(xrsrc, buf_a * XBUF * 4u + off, 0, SEQ_AUX_SC1) : (u32x4){0u, 0u, 0u, 0u};
      vb[q] = use_b ? __builtin_amdgcn_raw_buffer_load_b128(xrsrc, buf_b * XBUF * 4u + off, 0, SEQ_AUX_SC1) : (u32x4){0u, 0u, 0u, 0u};
    }
    f32x4 sum = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < NW; ++q) sum += __builtin_bit_cast(f32x4, va[q]);
#pragma unroll
    for (int q = 0; q < NW; ++q) sum += __builtin_bit_cast(f32x4, vb[q]);
    *reinterpret_cast<f32x4*>(&red[(wave * 32 + xrow) * 8 + xhalf * 4]) = sum;
    __syncthreads();
  };
  auto gathered = [&]() {
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < SEQ_WAVES; ++w) sum += red[(w * 32 + bl) * 8 + jj];
    return sum;
  };
  // this wave's 32 x 32 tile of dG * (rows of W) into destination order and out (see lstm_seq_bwd_kernel)
  auto product = [&](const f32x4 (&av)[4], const float (&w)[16], unsigned buf) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][e], w[4 * q + e], acc, 0, 0, 0);
    float* const tsw = Ts + wave * 32 * SEQ_A_LD;
#pragma unroll
    for (int r = 0; r < 16; ++r) tsw[((r & 3) + 8 * (r >> 2) + 4 * hh) * SEQ_A_LD + l31] = acc[r];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(&tsw[xrow * SEQ_A_LD + q * 8 + xhalf * 4]);
      const unsigned off = (buf * XBUF + (unsigned)slice * 32u * H + (unsigned)(wave * 4 + q) * 256u + (unsigned)lane * 4u) * 4u;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), xrsrc, off, 0, SEQ_AUX_SC1);
    }
  };

  const int T = g.T;
#pragma unroll 1
  for (int t = T - 1; t >= 0; --t) {
#pragma unroll
    for (int l = DEC_L - 1; l >= 0; --l) {
      const int sigma = (T - 1 - t) * DEC_L + (DEC_L - 1 - l);
      float ig = 0.f, fg = 0.f, gg = 0.f, og = 0.f, cv = 0.f, cpv = 0.f, dhv = 0.f;
      const int64_t row = ((int64_t)l * T + t) * g.B + b;
      if (bok) {
        const float* gt = g.gates + row * 4 * H + j;
        ig = gt[0]; fg = gt[H]; gg = gt[2 * H]; og = gt[3 * H];
        cv = g.cs[row * H + j];
        cpv = t > 0 ? g.cs[(row - g.B) * H + j] : g.c0[b * g.ld_c0 + l * H + j];
        if (l == DEC_L - 1) dhv = g.dstates[(int64_t)t * g.step_ds + b * g.ld_ds + j];
      }
      if (sigma > 0) {
        if (!seq_wait(cnt, tmo, (unsigned)NS * (unsigned)sigma, &s_abort)) {
          if (bok) {
            for (int s2 = sigma; s2 < T * DEC_L; ++s2) {
              const int t2 = T - 1 - s2 / DEC_L, l2 = DEC_L - 1 - s2 % DEC_L;
              float* dg = g.dgates + (((int64_t)l2 * T + t2) * g.B + b) * 4 * H + j;
              dg[0] = dg[H] = dg[2 * H] = dg[3 * H] = __builtin_nanf("");
            }
#pragma unroll
            for (int l2 = 0; l2 < DEC_L; ++l2) {
              g.dh0[b * g.ld_dh0 + l2 * H + j] = __builtin_nanf("");
              g.dc0[b * g.ld_dc0 + l2 * H + j] = __builtin_nanf("");
            }
            g.dx0[b * H + j] = __builtin_nanf("");
          }
          return;
        }
        const bool use_a = !(l == DEC_L - 1 && t == T - 1), use_b = t < T - 1;
        gather(use_a, buf_of(0, l, t), use_b, buf_of(1, l, t));
        if (cell) dhv += gathered();
      }
      float dg0 = 0.f, dg1 = 0.f, dg2 = 0.f, dg3 = 0.f;
      if (cell) {
        const float tc = tanhf(cv);
        const float dct = dcarry[l] + dhv * og * (1.f - tc * tc);
        dg0 = dct * gg * ig * (1.f - ig);
        dg1 = dct * cpv * fg * (1.f - fg);
        dg2 = dct * ig * (1.f - gg * gg);
        dg3 = dhv * tc * og * (1.f - og);
        dcarry[l] = dct * fg;
        if (!bok) dg0 = dg1 = dg2 = dg3 = 0.f;
        As[bl * SEQ_A_LD + 0 * 8 + jj] = dg0;
        As[bl * SEQ_A_LD + 1 * 8 + jj] = dg1;
        As[bl * SEQ_A_LD + 2 * 8 + jj] = dg2;
        As[bl * SEQ_A_LD + 3 * 8 + jj] = dg3;
      }
      __syncthreads();
      // Input path: to the cell that fed this one its x -- the NEXT cell of the chain, which waits for it.
      // Recurrent path: to this layer's previous step, two cells on; it is written behind this cell's arrival and
      // is covered by the next cell's (a wave drains all its stores before its workgroup signals).  The chain's
      // last cell has no next arrival and writes both in front of its own.
      const bool last = sigma + 1 == T * DEC_L;
      f32x4 av[4];
      if (has_tile) {
#pragma unroll
        for (int q = 0; q < 4; ++q) av[q] = *reinterpret_cast<const f32x4*>(&As[l31 * SEQ_A_LD + 16 * hh + 4 * q]);
        product(av, wx[l], l > 0 ? buf_of(0, l - 1, t) : buf_of(0, DEC_L - 1, t - 1));
        if (last) product(av, wh[l], buf_of(1, l, t - 1));
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_fetch_add((gu32_t*)cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (bok) {
        float* dg = g.dgates + row * 4 * H + j;
        dg[0] = dg0; dg[H] = dg1; dg[2 * H] = dg2; dg[3 * H] = dg3;
      }
      if (has_tile && !last) product(av, wh[l], buf_of(1, l, t - 1));
    }
  }
  // what left the chain at its end: destination step -1 (odd parity)
  if (!seq_wait(cnt, tmo, (unsigned)NS * (unsigned)(T * DEC_L), &s_abort)) {
    if (bok) {
#pragma unroll
      for (int l2 = 0; l2 < DEC_L; ++l2) {
        g.dh0[b * g.ld_dh0 + l2 * H + j] = __builtin_nanf("");
        g.dc0[b * g.ld_dc0 + l2 * H + j] = __builtin_nanf("");
      }
      g.dx0[b * H + j] = __builtin_nanf("");
    }
    return;
  }
#pragma unroll
  for (int l = 0; l < DEC_L; ++l) {
    gather(false, 0u, true, buf_of(1, l, -1));
    if (bok) {
      g.dh0[b * g.ld_dh0 + l * H + j] = gathered();
      g.dc0[b * g.ld_dc0 + l * H + j] = dcarry[l];
    }
    __syncthreads();
  }
  gather(true, buf_of(0, DEC_L - 1, -1), false, 0u);
  if (bok) g.dx0[b * H + j] = gathered();
}

extern "C" int dvae_lstm_dec2_bwd_f32(const float* const* w_ih, const float* const* w_hh, const float* gates,
                                      const float* cs, const float* c0, int64_t ld_c0, const float* dstates,
                                      int64_t ld_ds, int64_t step_ds, float* dgates, float* dh0, int64_t ld_dh0,
                                      float* dc0, int64_t ld_dc0, float* dx0, int64_t T, int64_t B, int64_t H,
                                      float* ws, dvae_stream_t stream) {
  DVAE_REQUIRE(dvae_lstm_dec2_supported(T, B, H),
               "lstm_dec2_bwd: unsupported shape (T=%ld B=%ld H=%ld): H must be 128 or 256 and ceil(B/32)*(H/8) at most 256",
               (long)T, (long)B, (long)H);
  DVAE_REQUIRE(w_ih && w_hh && gates && cs && c0 && dstates && dgates && dh0 && dc0 && dx0, "lstm_dec2_bwd: null argument");
  DVAE_REQUIRE(ws != nullptr && (((uintptr_t)ws) & 255) == 0, "lstm_dec2_bwd: workspace must be 256-byte aligned");
  LstmDecBwdArgs g;
  for (int l = 0; l < DEC_L; ++l) {
    g.w_ih[l] = w_ih[l]; g.w_hh[l] = w_hh[l];
    DVAE_REQUIRE(g.w_ih[l] && g.w_hh[l], "lstm_dec2_bwd: weights of layer %d missing", l);
  }
  g.gates = gates; g.cs = cs; g.c0 = c0; g.ld_c0 = ld_c0;
  g.dstates = dstates; g.ld_ds = ld_ds; g.step_ds = step_ds;
  g.dgates = dgates; g.dh0 = dh0; g.ld_dh0 = ld_dh0; g.dc0 = dc0; g.ld_dc0 = ld_dc0; g.dx0 = dx0;
  g.T = (int)T; g.B = B; g.groups = (int)ceil_div64(B, 32);
  g.sync = (unsigned*)ws;
  g.xch = ws + SEQ_CNT_STRIDE * (1 + g.groups);
  if (seq_clear(ws, g.groups, ST(stream)) != 0) {
    dvae_set_error("lstm_dec2_bwd: clearing the counters failed: %s", hipGetErrorString(hipGetLastError()));
    return -EIO;
  }
  const int NS = (int)(H / 8);
  const unsigned nblocks = (unsigned)(8 * NS * ((g.groups + 7) / 8));
  DVAE_WORK(2.0 * (double)DEC_L * T * B * 4 * H * 2 * H, DVAE_UNIT_FLOP);
  if (H == 128) DVAE_LAUNCH((lstm_dec_bwd_kernel<128>), dim3(nblocks), dim3(64 * SEQ_WAVES), 0, ST(stream), g);
  else DVAE_LAUNCH((lstm_dec_bwd_kernel<256>), dim3(nblocks), dim3(64 * SEQ_WAVES), 0, ST(stream), g);
  DVAE_LAUNCH_CHECK("lstm_dec2_bwd");
  return 0;
}

// Number of workgroups, on the current GPU and since the library was loaded, that gave up a bounded wait inside one
// of the one-launch recurrence kernels (their outputs are then invalid).  Synchronises with the device.
extern "C" int dvae_lstm_seq_timeouts_clear(void) {
  const unsigned zero = 0;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_seq_timeouts), &zero, sizeof(zero), 0, hipMemcpyHostToDevice) != hipSuccess) {
    dvae_set_error("lstm_seq_timeouts_clear: writing the counter failed: %s", hipGetErrorString(hipGetLastError()));
    return -EIO;
  }
  return 0;
}

extern "C" int64_t dvae_lstm_seq_timeouts(void) {
  unsigned v = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_seq_timeouts), sizeof(v), 0, hipMemcpyDeviceToHost) != hipSuccess) {
    dvae_set_error("lstm_seq_timeouts: reading the counter failed: %s", hipGetErrorString(hipGetLastError()));
    return -EIO;
  }
  return (int64_t)v;
}

// ------------------------------------------------------------------------------------------------
// Residency.  The one-launch kernels exchange data between workgroups inside the launch, so every workgroup of a
// grid must be on the chip at the same time.  How many fit is a property of the device and of the compiled
// kernels, so it is asked of the runtime (compute units x co-resident workgroups per compute unit, the smaller of
// the forward and the backward kernel), once per device and hidden size, instead of assuming 256 compute units.
// ------------------------------------------------------------------------------------------------
static int seq_occupancy(const void* kernel) {
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, 64 * SEQ_WAVES, 0) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

// kind 0: lstm_seq_fwd / lstm_seq_bwd (H = 128, 256, 512); kind 1: the stacked decoder's pair (H = 128, 256).
// Returns {compute units, workgroups per compute unit}, {0, 0} without a usable device.
static void seq_residency(int kind, int64_t H, int* cus, int* per_cu) {
  static std::mutex mu;
  static int cache[16][2][3][2];            // [device][kind][H index] -> {cus, per_cu}; 0 = not asked yet
  *cus = 0; *per_cu = 0;
  const int hi = H == 128 ? 0 : H == 256 ? 1 : H == 512 ? 2 : -1;
  int dev = -1;
  if (hi < 0 || kind < 0 || kind > 1 || (kind == 1 && hi == 2)) return;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) { (void)hipGetLastError(); return; }
  std::lock_guard<std::mutex> lock(mu);
  int* e = cache[dev][kind][hi];
  if (e[0] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); return; }
    int f = 0, b = 0;
    if (kind == 0) {
      f = hi == 0 ? seq_occupancy((const void*)lstm_seq_fwd_kernel<16>) : hi == 1 ? seq_occupancy((const void*)lstm_seq_fwd_kernel<32>)
                                                                                  : seq_occupancy((const void*)lstm_seq_fwd_kernel<64>);
      b = hi == 0 ? seq_occupancy((const void*)lstm_seq_bwd_kernel<128>) : hi == 1 ? seq_occupancy((const void*)lstm_seq_bwd_kernel<256>)
                                                                                   : seq_occupancy((const void*)lstm_seq_bwd_kernel<512>);
    } else {
      f = hi == 0 ? seq_occupancy((const void*)lstm_dec_fwd_kernel<16>) : seq_occupancy((const void*)lstm_dec_fwd_kernel<32>);
      b = hi == 0 ? seq_occupancy((const void*)lstm_dec_bwd_kernel<128>) : seq_occupancy((const void*)lstm_dec_bwd_kernel<256>);
    }
    e[0] = n;
    e[1] = f < b ? f : b;
  }
  *cus = e[0]; *per_cu = e[1];
}

extern "C" int64_t dvae_lstm_resident_workgroups(int kind, int64_t H, int64_t max_per_cu) {
  int cus = 0, per_cu = 0;
  seq_residency(kind, H, &cus, &per_cu);
  if (max_per_cu > 0 && per_cu > max_per_cu) per_cu = (int)max_per_cu;
  return (int64_t)cus * per_cu;
}

extern "C" int dvae_lstm_seq_supported(int64_t nchains, int64_t T, int64_t B, int64_t H) {
  if (nchains < 1 || nchains > SEQ_MAX_CHAINS || T < 1 || B < 1) return 0;
  if (H != 128 && H != 256 && H != 512) return 0;
  // every workgroup must be resident at once (at most two of these 512-thread workgroups per compute unit)
  return nchains * ceil_div64(B, 32) * (H / 8) <= dvae_lstm_resident_workgroups(0, H, 2);
}

extern "C" int dvae_lstm_dec2_supported(int64_t T, int64_t B, int64_t H) {
  if (T < 1 || B < 1 || (H != 128 && H != 256)) return 0;
  return ceil_div64(B, 32) * (H / 8) <= dvae_lstm_resident_workgroups(1, H, 1);   // one workgroup per compute unit
}
