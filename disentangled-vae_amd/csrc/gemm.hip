// Linear-layer GEMMs (forward NT, data-gradient NN, weight-gradient TN) on the shared f32-MFMA core.
#include "gemm_core.h"
#include "../../include/dvae_hip.h"

#define ST(s) ((hipStream_t)(s))

enum { G_NT = 0, G_NN = 1, G_TN = 2 };

struct GemmArgs {
  const float* A;
  const float* B;
  const float* bias;
  float* C;
  float* slab;  // split-K partials [split][M][N] or nullptr
  int64_t lda, ldb, ldc;
  int64_t M, N, K;
  int act, accumulate, ktiles_per_split;
};

template <class T, int KIND, bool BF>
__global__ void __launch_bounds__(256, 2) gemm_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float smem[2 * Core<T, BF>::STAGE];
  const int64_t m0 = (int64_t)blockIdx.y * T::BM, n0 = (int64_t)blockIdx.x * T::BN;
  const int ktiles = (int)((g.K + 31) / 32);
  const int kt_begin = blockIdx.z * g.ktiles_per_split;
  int kt_end = kt_begin + g.ktiles_per_split;
  if (kt_end > ktiles) kt_end = ktiles;

  f32x16 acc[T::TM][T::TN];
  zero_acc<T>(acc);

  if (KIND == G_NT) {
    KCLoader<T::A_V4> la(g.A, g.lda, g.M, g.K, m0);
    KCLoader<T::B_V4> lb(g.B, g.ldb, g.N, g.K, n0);
    gemm_mainloop<T, BF>(la, lb, kt_begin, kt_end, smem, acc);
  } else if (KIND == G_NN) {
    KCLoader<T::A_V4> la(g.A, g.lda, g.M, g.K, m0);
    RCLoader<T::B_V4, T::BN, BF> lb(g.B, g.ldb, g.N, g.K, n0);
    gemm_mainloop<T, BF>(la, lb, kt_begin, kt_end, smem, acc);
  } else {
    RCLoader<T::A_V4, T::BM, BF> la(g.A, g.lda, g.M, g.K, m0);
    RCLoader<T::B_V4, T::BN, BF> lb(g.B, g.ldb, g.N, g.K, n0);
    gemm_mainloop<T, BF>(la, lb, kt_begin, kt_end, smem, acc);
  }

  if (g.slab) {
    float* dst = g.slab + (int64_t)blockIdx.z * g.M * g.N;
    for_each_acc<T>(acc, [&](int r, int c, float v) {
      const int64_t m = m0 + r, n = n0 + c;
      if (m < g.M && n < g.N) dst[m * g.N + n] = v;
    });
  } else {
    for_each_acc<T>(acc, [&](int r, int c, float v) {
      const int64_t m = m0 + r, n = n0 + c;
      if (m < g.M && n < g.N) {
        float o = v;
        if (g.bias) o += g.bias[n];
        if (g.act == DVAE_ACT_LRELU) o = lrelu(o);
        float* p = g.C + m * g.ldc + n;
        if (g.accumulate) o += *p;
        *p = o;
      }
    });
  }
}

__global__ void __launch_bounds__(256)
splitk_reduce_kernel(const float* __restrict__ slab, int split, const float* __restrict__ bias, int act,
                     int accumulate, float* __restrict__ C, int64_t ldc, int64_t M, int64_t N) {
  const int64_t total = M * N;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    float s = 0.f;
    for (int z = 0; z < split; ++z) s += slab[(int64_t)z * total + i];
    const int64_t m = i / N, n = i - m * N;
    if (bias) s += bias[n];
    if (act == DVAE_ACT_LRELU) s = lrelu(s);
    float* p = C + m * ldc + n;
    if (accumulate) s += *p;
    *p = s;
  }
}

static int choose_split(int64_t tiles, int64_t ktiles) {
  int64_t s = 512 / (tiles > 0 ? tiles : 1);
  if (s > ktiles / 4) s = ktiles / 4;
  if (s > 32) s = 32;
  if (s < 1) s = 1;
  return (int)s;
}

struct Plan {
  bool small_m;
  int64_t tm, tn;
  int split, ktiles_per_split;
};

static Plan make_plan(int64_t M, int64_t N, int64_t K) {
  Plan p;
  p.small_m = M <= 32;
  const int64_t BM = p.small_m ? 32 : 128, BN = 128;
  p.tm = ceil_div64(M, BM);
  p.tn = ceil_div64(N, BN);
  const int64_t ktiles = ceil_div64(K, 32);
  p.split = choose_split(p.tm * p.tn, ktiles);
  p.ktiles_per_split = (int)ceil_div64(ktiles, p.split);
  p.split = (int)ceil_div64(ktiles, p.ktiles_per_split);
  return p;
}

extern "C" int64_t dvae_gemm_ws_floats(int64_t M, int64_t N, int64_t K) {
  const Plan p = make_plan(M, N, K);
  return p.split > 1 ? (int64_t)p.split * M * N : 4;
}

template <int KIND>
static int launch_gemm(const char* name, GemmArgs g, float* ws, hipStream_t stream) {
  const Plan p = make_plan(g.M, g.N, g.K);
  g.ktiles_per_split = p.ktiles_per_split;
  g.slab = p.split > 1 ? ws : nullptr;
  if (p.split > 1 && ws == nullptr) {
    dvae_set_error("%s: split-K needs a workspace of %ld floats", name, (long)((int64_t)p.split * g.M * g.N));
    return -EINVAL;
  }
  const dim3 grid((unsigned)p.tn, (unsigned)p.tm, (unsigned)p.split);
  constexpr bool AKC = KIND != G_TN, BKC = KIND == G_NT;
  const bool bf = dvae_mfma_precision() == DVAE_PRECISION_BF16;
  if (p.small_m) {
    if (bf) hipLaunchKernelGGL((gemm_kernel<Tile<32, 128, 1, 4, AKC, BKC>, KIND, true>), grid, dim3(256), 0, stream, g);
    else hipLaunchKernelGGL((gemm_kernel<Tile<32, 128, 1, 4, AKC, BKC>, KIND, false>), grid, dim3(256), 0, stream, g);
  } else {
    if (bf) hipLaunchKernelGGL((gemm_kernel<Tile<128, 128, 2, 2, AKC, BKC>, KIND, true>), grid, dim3(256), 0, stream, g);
    else hipLaunchKernelGGL((gemm_kernel<Tile<128, 128, 2, 2, AKC, BKC>, KIND, false>), grid, dim3(256), 0, stream, g);
  }
  DVAE_LAUNCH_CHECK(name);
  if (p.split > 1) {
    int64_t nb = ceil_div64(g.M * g.N, 256);
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, stream, (const float*)ws, p.split,
                       g.bias, g.act, g.accumulate, g.C, g.ldc, g.M, g.N);
    DVAE_LAUNCH_CHECK(name);
  }
  return 0;
}

static bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

extern "C" int dvae_gemm_nt_f32(const float* A, int64_t lda, const float* B, int64_t ldb, const float* bias, int act,
                                float* C, int64_t ldc, int64_t M, int64_t N, int64_t K, float* ws,
                                dvae_stream_t stream) {
  DVAE_REQUIRE(M > 0 && N > 0 && K > 0, "gemm_nt: empty problem");
  DVAE_REQUIRE(K % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0, "gemm_nt: K, lda, ldb must be multiples of 4 (K=%ld lda=%ld ldb=%ld)",
               (long)K, (long)lda, (long)ldb);
  DVAE_REQUIRE(aligned16(A) && aligned16(B), "gemm_nt: A and B must be 16-byte aligned");
  DVAE_REQUIRE(act == DVAE_ACT_NONE || act == DVAE_ACT_LRELU, "gemm_nt: unsupported activation %d", act);
  GemmArgs g{A, B, bias, C, nullptr, lda, ldb, ldc, M, N, K, act, 0, 0};
  return launch_gemm<G_NT>("gemm_nt", g, ws, ST(stream));
}

extern "C" int dvae_gemm_nn_f32(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                                int64_t M, int64_t N, int64_t K, int accumulate, float* ws, dvae_stream_t stream) {
  DVAE_REQUIRE(M > 0 && N > 0 && K > 0, "gemm_nn: empty problem");
  DVAE_REQUIRE(K % 4 == 0 && N % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0,
               "gemm_nn: K, N, lda, ldb must be multiples of 4 (K=%ld N=%ld)", (long)K, (long)N);
  DVAE_REQUIRE(aligned16(A) && aligned16(B), "gemm_nn: A and B must be 16-byte aligned");
  GemmArgs g{A, B, nullptr, C, nullptr, lda, ldb, ldc, M, N, K, DVAE_ACT_NONE, accumulate, 0};
  return launch_gemm<G_NN>("gemm_nn", g, ws, ST(stream));
}

extern "C" int dvae_gemm_tn_f32(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                                int64_t M, int64_t N, int64_t K, int accumulate, float* ws, dvae_stream_t stream) {
  DVAE_REQUIRE(M > 0 && N > 0 && K > 0, "gemm_tn: empty problem");
  DVAE_REQUIRE(M % 4 == 0 && N % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0,
               "gemm_tn: M, N, lda, ldb must be multiples of 4 (M=%ld N=%ld)", (long)M, (long)N);
  DVAE_REQUIRE(aligned16(A) && aligned16(B), "gemm_tn: A and B must be 16-byte aligned");
  GemmArgs g{A, B, nullptr, C, nullptr, lda, ldb, ldc, M, N, K, DVAE_ACT_NONE, accumulate, 0};
  return launch_gemm<G_TN>("gemm_tn", g, ws, ST(stream));
}
