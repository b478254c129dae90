// Linear-layer GEMMs (forward NT, data-gradient NN, weight-gradient TN) on the shared f32-MFMA core.
#include "gemm_core.h"
#include "../../include/dvae_hip.h"

#define ST(s) ((hipStream_t)(s))

enum { G_NT = 0, G_NN = 1, G_TN = 2 };
// Build-time switches, each the shipped choice of an A/B that -D...=0 takes back:
// fp32 NT products on whole tiles keep two K tiles of loads in flight, like the NN and TN products
#ifndef DVAE_NT_AHEAD2
#define DVAE_NT_AHEAD2 1
#endif
// whole tiles take loaders without address selects and zero masks (KCLoader / RCLoader <..., WHOLE>)
#ifndef DVAE_GEMM_WHOLE_TILES
#define DVAE_GEMM_WHOLE_TILES 1
#endif
// bf16 operands: three workgroups per CU (the output tile leaves through LDS in two runs of rows)
#ifndef DVAE_BF16_WG3
#define DVAE_BF16_WG3 1
#endif

static bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

struct GemmArgs {
  const float* A;
  const float* B;
  const float* bias;
  float* C;
  float* slab;  // split-K partials [split][M][N] or nullptr
  int64_t lda, ldb, ldc;
  int64_t M, N, K;
  int act, accumulate, ktiles_per_split;
  int tm0;      // first row of tiles of this launch (a product may be split over two launches)
  int wide_rows;  // rows of the output (and of the partial slabs) may be written 16 bytes at a time
  int group_n;    // the column tiles of one row of tiles (and K range) run on one XCD, next to each other
};

// Diagnostic build only (-DDVAE_GEMM_STAMP=1, never the shipped library): every workgroup's shader-clock and
// real-time ticks across its main loop, for the clock the chip holds under this load
// (shader ticks / 100 MHz ticks x 100 MHz), read back with dvae_debug_read_stamps (tools/debug/clock_probe.py).
#ifndef DVAE_GEMM_STAMP
#define DVAE_GEMM_STAMP 0
#endif
#if DVAE_GEMM_STAMP
#define STAMP_WGS 4096
__device__ unsigned long long dvae_stamps[2 * STAMP_WGS];
extern "C" int dvae_debug_read_stamps(unsigned long long* host, int n) {
  if (n > 2 * STAMP_WGS) n = 2 * STAMP_WGS;
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(dvae_stamps), sizeof(unsigned long long) * n);
}
#endif

template <class T, int KIND, int BF>      // BF: PREC_F32 / PREC_BF16 / PREC_SPLIT
__global__ void __launch_bounds__(256, ((DVAE_GEMM_ONE_STAGE && KIND == G_NT && BF == PREC_F32) || (BF == PREC_BF16 && DVAE_BF16_WG3)) ? 3 : 2) gemm_kernel(GemmArgs g) {
  // the K loop's stages, which also carry the output tile on its way out: whole when it fits, else in EPI_PASSES
  // runs of rows (the bf16 stages are 40 KB, the 128 x 128 tile 68 KB: two runs, and a third workgroup per CU)
  constexpr int LOOP_FLOATS = ((DVAE_GEMM_ONE_STAGE && KIND == G_NT) ? 1 : 2) * Core<T, BF>::STAGE;
  constexpr int TILE_FLOATS = T::BM * (T::BN + 4);
  constexpr int EPI_PASSES = (BF == PREC_BF16 && DVAE_BF16_WG3 && TILE_FLOATS > LOOP_FLOATS) ? 2 : 1;
  constexpr int SMEM_FLOATS = (BF == PREC_BF16 && TILE_FLOATS / EPI_PASSES > LOOP_FLOATS) ? TILE_FLOATS / EPI_PASSES : LOOP_FLOATS;
  __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];
  // Workgroups go to the eight XCDs in turn (linear id mod 8) and every XCD has its own L2.  Numbered as launched,
  // the few column tiles that share one A tile land on different XCDs and each fetches that tile for itself; with
  // group_n they are numbered eight apart instead: (row of tiles, K range) g = 8 * (slot / columns) + XCD, column =
  // slot % columns.  Groups beyond the last multiple of eight keep the plain numbering.
  unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (g.group_n) {
    const unsigned tn = gridDim.x, groups = gridDim.y * gridDim.z;
    const unsigned id = blockIdx.x + tn * (blockIdx.y + gridDim.y * blockIdx.z);
    if (id < (groups & ~7u) * tn) {
      const unsigned slot = id >> 3, grp = (slot / tn) * 8 + (id & 7);
      bx = slot % tn;
      by = grp % gridDim.y;
      bz = grp / gridDim.y;
    }
  }
  const int64_t m0 = (int64_t)(by + g.tm0) * T::BM, n0 = (int64_t)bx * T::BN;
  const int ktiles = (int)((g.K + 31) / 32);
  const int kt_begin = bz * g.ktiles_per_split;
  int kt_end = kt_begin + g.ktiles_per_split;
  if (kt_end > ktiles) kt_end = ktiles;

  f32x16 acc[T::TM][T::TN];
  zero_acc<T>(acc);

#if DVAE_GEMM_STAMP
  const unsigned long long st0 = __builtin_amdgcn_s_memtime(), sr0 = __builtin_amdgcn_s_memrealtime();
#endif
  // whole tile, whole K tiles: loaders without address selects and zero masks (see KCLoader).  With bf16 operands
  // 8-13 % of a large product; with fp32 operands 5-10 % of the NN and TN products, whose loads run two K tiles
  // ahead, and 7-13 % of the NT products when they take that loop as well (with the general loaders it costs them
  // registers and gains nothing; with these loaders alone, nothing).
  const bool whole = m0 + T::BM <= g.M && n0 + T::BN <= g.N && g.K % 32 == 0;
#define GEMM_RUN(W)                                                                                    \
  if (KIND == G_NT) {                                                                                  \
    KCLoader<T::A_V4, W> la(g.A, g.lda, g.M, g.K, m0);                                                 \
    KCLoader<T::B_V4, W> lb(g.B, g.ldb, g.N, g.K, n0);                                                 \
    if (DVAE_NT_AHEAD2 && BF == PREC_F32 && W) gemm_mainloop_ahead2<T, BF>(la, lb, kt_begin, kt_end, smem, acc); \
    else gemm_mainloop<T, BF>(la, lb, kt_begin, kt_end, smem, acc);                                    \
  } else if (KIND == G_NN) {                                                                           \
    KCLoader<T::A_V4, W> la(g.A, g.lda, g.M, g.K, m0);                                                 \
    RCLoader<T::B_V4, T::BN, Core<T, BF>::KADJ, W> lb(g.B, g.ldb, g.N, g.K, n0);                       \
    if (BF == PREC_F32) gemm_mainloop_ahead2<T, BF>(la, lb, kt_begin, kt_end, smem, acc);              \
    else gemm_mainloop<T, BF>(la, lb, kt_begin, kt_end, smem, acc);                                    \
  } else {                                                                                             \
    RCLoader<T::A_V4, T::BM, Core<T, BF>::KADJ, W> la(g.A, g.lda, g.M, g.K, m0);                       \
    RCLoader<T::B_V4, T::BN, Core<T, BF>::KADJ, W> lb(g.B, g.ldb, g.N, g.K, n0);                       \
    if (BF == PREC_F32) gemm_mainloop_ahead2<T, BF>(la, lb, kt_begin, kt_end, smem, acc);              \
    else gemm_mainloop<T, BF>(la, lb, kt_begin, kt_end, smem, acc);                                    \
  }
  if (DVAE_GEMM_WHOLE_TILES && (BF == PREC_BF16 || KIND != G_NT || DVAE_NT_AHEAD2) && whole) {
    GEMM_RUN(true)
  } else {
    GEMM_RUN(false)
  }
#undef GEMM_RUN

#if DVAE_GEMM_STAMP
  {
    const unsigned long long st1 = __builtin_amdgcn_s_memtime(), sr1 = __builtin_amdgcn_s_memrealtime();
    const unsigned id = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if (threadIdx.x == 0 && id < STAMP_WGS) {
      dvae_stamps[2 * id] = st1 - st0;
      dvae_stamps[2 * id + 1] = sr1 - sr0;
    }
  }
#endif
  // Epilogue.  The accumulators hold four consecutive ROWS of one column per lane, so written straight from
  // registers a tile leaves as 64 four-byte stores per lane; when every workgroup of a round reaches its epilogue at
  // the same time (equal tiles) that store tail is not hidden by anybody's multiplies.  With whole, aligned rows the
  // tile goes through the (now free) LDS stages instead and leaves as 16-byte stores along the rows: a quarter of
  // the store instructions, 512 contiguous bytes per row and wave instruction.  Same values, same arithmetic.
  constexpr int TLD = T::BN + 4;
  constexpr int RUN = T::BM / EPI_PASSES;                 // rows of the tile that travel through LDS together
  constexpr bool fits = SMEM_FLOATS >= RUN * TLD;
  float* dst = g.slab ? g.slab + (int64_t)bz * g.M * g.N : g.C;
  const int64_t ldd = g.slab ? g.N : g.ldc;
  const bool wide = fits && g.wide_rows && m0 + T::BM <= g.M && n0 + T::BN <= g.N;
  if (wide) {
    constexpr int C4 = T::BN / 4, PER = RUN * C4 / 256;
    static_assert(256 % C4 == 0, "a thread keeps its columns for every piece it writes");
    const int c4 = threadIdx.x % C4;
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!g.slab && g.bias) bias4 = *reinterpret_cast<const float4*>(g.bias + n0 + 4 * c4);
#pragma unroll
    for (int pass = 0; pass < EPI_PASSES; ++pass) {
      if (pass > 0) __syncthreads();
      for_each_acc<T>(acc, [&](int r, int c, float v) {
        if (EPI_PASSES == 1 || r / RUN == pass) smem[(r - pass * RUN) * TLD + c] = v;
      });
      __syncthreads();
#pragma unroll
      for (int it = 0; it < PER; ++it) {
        const int row = (it * 256 + threadIdx.x) / C4;
        float4 o = *reinterpret_cast<const float4*>(smem + row * TLD + 4 * c4);
        float4* p = reinterpret_cast<float4*>(dst + (m0 + pass * RUN + row) * ldd + n0 + 4 * c4);
        if (!g.slab) {
          o.x += bias4.x; o.y += bias4.y; o.z += bias4.z; o.w += bias4.w;
          if (g.act == DVAE_ACT_LRELU) { o.x = lrelu(o.x); o.y = lrelu(o.y); o.z = lrelu(o.z); o.w = lrelu(o.w); }
          if (g.accumulate) {
            const float4 q = *p;
            o.x += q.x; o.y += q.y; o.z += q.z; o.w += q.w;
          }
        }
        *p = o;
      }
    }
  } else if (g.slab) {
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

// Sum of the split-K partials [split][M][N] plus the epilogue.  Four slabs are read per pass with
// independent loads (a one-slab-at-a-time loop is a chain of dependent memory latencies) and, when the
// row length allows, four columns per thread; the order of the additions is fixed.
template <int V>
__global__ void __launch_bounds__(256)
splitk_reduce_kernel(const float* __restrict__ slab, int split, const float* __restrict__ bias, int act,
                     int accumulate, float* __restrict__ C, int64_t ldc, int64_t M, int64_t N) {
  typedef float vec __attribute__((ext_vector_type(V)));
  const int64_t total = M * N, totalv = total / V;
  for (int64_t iv = (int64_t)blockIdx.x * 256 + threadIdx.x; iv < totalv; iv += (int64_t)gridDim.x * 256) {
    const int64_t i = iv * V;
    vec s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int z = 0;
    for (; z + 4 <= split; z += 4) {
      s0 += *reinterpret_cast<const vec*>(slab + (int64_t)z * total + i);
      s1 += *reinterpret_cast<const vec*>(slab + (int64_t)(z + 1) * total + i);
      s2 += *reinterpret_cast<const vec*>(slab + (int64_t)(z + 2) * total + i);
      s3 += *reinterpret_cast<const vec*>(slab + (int64_t)(z + 3) * total + i);
    }
    for (; z < split; ++z) s0 += *reinterpret_cast<const vec*>(slab + (int64_t)z * total + i);
    const vec s = (s0 + s1) + (s2 + s3);
    const int64_t m = i / N, n = i - m * N;
    float* p = C + m * ldc + n;
#pragma unroll
    for (int e = 0; e < V; ++e) {
      float o = s[e];
      if (bias) o += bias[n + e];
      if (act == DVAE_ACT_LRELU) o = lrelu(o);
      if (accumulate) o += p[e];
      p[e] = o;
    }
  }
}

// Small products (a few hundred 16x16 output tiles, e.g. the latent heads at batch size): a tiled
// launch would be a handful of workgroups walking K in 32-deep steps plus a split-K reduction.  Here a
// workgroup owns one 16x16 tile, its eight waves split K between them and load their operand fragments
// straight from memory into the register layout of v_mfma_f32_16x16x4_f32 (lane = row or column x
// k quad; 16-byte loads along a k-contiguous operand, four 4-byte loads down a row-contiguous one),
// all loads of a pass in flight together; the partial tiles meet in LDS, where bias, activation and
// accumulation are applied.  One launch, no workspace.
#define SG_WAVES 8
#define SG_NF 4          // k groups of sixteen per wave and pass
#define SG_RED_LD 17
// BF: operands rounded to bf16 (the bf16 mode's contract), products still exact in the f32 MFMA.
template <int KIND, bool BF>
__global__ void __launch_bounds__(64 * SG_WAVES) small_gemm_kernel(GemmArgs g) {
  __shared__ float red[SG_WAVES * 16 * SG_RED_LD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int64_t n0 = (int64_t)blockIdx.x * 16, m0 = (int64_t)blockIdx.y * 16;
  const bool rowok = m0 + i < g.M, colok = n0 + i < g.N;
  const int64_t row = rowok ? m0 + i : g.M - 1, col = colok ? n0 + i : g.N - 1;
  // the value to accumulate onto is requested first (threads 0..255 own one output each)
  const int orow = (threadIdx.x >> 4) & 15, ocol = threadIdx.x & 15;
  const bool owner = threadIdx.x < 256 && m0 + orow < g.M && n0 + ocol < g.N;
  float* const op = g.C + (m0 + orow) * g.ldc + n0 + ocol;
  float prev = 0.f, bias = 0.f;
  if (owner && g.accumulate) prev = *op;
  if (owner && g.bias) bias = g.bias[n0 + ocol];

  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f}, acc2 = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int64_t k0 = (int64_t)wave * 16 * SG_NF; k0 < g.K; k0 += 16 * SG_NF * SG_WAVES) {
    f32x4 av[SG_NF], bv[SG_NF];
#pragma unroll
    for (int f = 0; f < SG_NF; ++f) {
      const int64_t k = k0 + 16 * f + 4 * q;       // this lane's four k of group f
      if (KIND != G_TN) {                          // A[m][k], k-contiguous (K % 4 == 0)
        const bool ok = rowok && k < g.K;
        const f32x4 v = ldg4(g.A + row * g.lda + (ok ? k : 0));
        av[f] = zero_unless(v, ok);
      } else {                                     // A[k][m]
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool ok = rowok && k + e < g.K;
          const float v = g.A[(ok ? k + e : 0) * g.lda + row];
          av[f][e] = ok ? v : 0.f;
        }
      }
      if (KIND == G_NT) {                          // B[n][k], k-contiguous
        const bool ok = colok && k < g.K;
        const f32x4 v = ldg4(g.B + col * g.ldb + (ok ? k : 0));
        bv[f] = zero_unless(v, ok);
      } else {                                     // B[k][n]
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool ok = colok && k + e < g.K;
          const float v = g.B[(ok ? k + e : 0) * g.ldb + col];
          bv[f][e] = ok ? v : 0.f;
        }
      }
    }
    if (BF) {
#pragma unroll
      for (int f = 0; f < SG_NF; ++f) {
        av[f] = round_bf16(av[f]);
        bv[f] = round_bf16(bv[f]);
      }
    }
    // two accumulators: consecutive MFMAs of a wave do not wait for each other's result
#pragma unroll
    for (int f = 0; f < SG_NF; ++f) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[f][0], bv[f][0], acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[f][1], bv[f][1], acc2, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[f][2], bv[f][2], acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[f][3], bv[f][3], acc2, 0, 0, 0);
    }
  }
  acc += acc2;
  // D: row = 4q + r, column = i
#pragma unroll
  for (int r = 0; r < 4; ++r) red[(wave * 16 + 4 * q + r) * SG_RED_LD + i] = acc[r];
  __syncthreads();
  if (owner) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < SG_WAVES; ++w) s += red[(w * 16 + orow) * SG_RED_LD + ocol];
    s += bias;
    if (g.act == DVAE_ACT_LRELU) s = lrelu(s);
    *op = s + prev;
  }
}

// the small-product kernel pays when its tiles are few and the operand rows they re-read stay small
static bool small_gemm_shape(int64_t M, int64_t N, int64_t K) {
  const int64_t tiles = ceil_div64(M, 16) * ceil_div64(N, 16);
  return tiles <= 1024 && tiles * K <= (1 << 20);
}

// Split of K over workgroups.  The chip holds 512 workgroups of this kernel at a time, so a launch
// costs (rounds of 512) x (k-tiles per workgroup); each split adds a pass over the M x N partials,
// priced in the same unit: one round of one k-tile is ~0.54 GFLOP ~ 4.9 us, which is the time HBM
// needs for ~5 M floats of slab traffic.
static int choose_split(int64_t tiles, int64_t ktiles, int64_t mn, int64_t slots) {
  if (tiles >= slots) return 1;
  // floats of slab traffic that cost as much as one round of one k-tile (with bf16 operands a k-tile is ~1.2 us)
  const double slab_unit = slots > 512 ? 1.2e6 : 5.0e6;
  int64_t smax = ktiles / 4;
  if (smax > 32) smax = 32;
  int best = 1;
  double best_cost = (double)ktiles;            // one round, no partials
  for (int64_t s = 2; s <= smax; ++s) {
    const int64_t rounds = (tiles * s + slots - 1) / slots;
    const double cost = (double)rounds * (double)((ktiles + s - 1) / s) + (double)(s + 1) * (double)mn / slab_unit;
    if (cost < best_cost * 0.9) {
      best_cost = cost;
      best = (int)s;
    }
  }
  return best;
}

// Whether the column tiles of a row of tiles should share an XCD (GemmArgs::group_n), by the operand bytes that
// cross the fabric either way: numbered as launched, A is fetched by min(columns, 8) XCDs and B by 8 / columns of
// them; grouped, A is fetched once and B by all eight, once per round of 512 workgroups.
static bool group_column_tiles(int64_t M, int64_t N, int64_t K, int64_t tn, int64_t workgroups) {
  static const bool off = [] { const char* v = getenv("DVAE_GEMM_GROUP_N"); return v && v[0] == '0'; }();
  if (off || tn < 2 || tn > 8) return false;
  const double a = (double)M * (double)K, b = (double)N * (double)K;
  const double rounds = (double)ceil_div64(workgroups, 512);
  return a + 8.0 * rounds * b < a * (double)tn + b * (8.0 / (double)tn);
}

struct Plan {
  bool small_m;
  int64_t tm, tn;
  int split, ktiles_per_split;
};

// workgroups the chip holds at a time: two per CU, three with bf16 operands (smaller LDS stages)
static int64_t resident_slots() { return (DVAE_BF16_WG3 && dvae_mfma_precision() == DVAE_PRECISION_BF16) ? 768 : 512; }

static Plan make_plan(int64_t M, int64_t N, int64_t K) {
  Plan p;
  p.small_m = M <= 32;
  const int64_t BM = p.small_m ? 32 : 128, BN = 128;
  p.tm = ceil_div64(M, BM);
  p.tn = ceil_div64(N, BN);
  const int64_t ktiles = ceil_div64(K, 32);
  p.split = choose_split(p.tm * p.tn, ktiles, M * N, resident_slots());
  p.ktiles_per_split = (int)ceil_div64(ktiles, p.split);
  p.split = (int)ceil_div64(ktiles, p.ktiles_per_split);
  return p;
}

extern "C" int64_t dvae_gemm_ws_floats(int64_t M, int64_t N, int64_t K) {
  if (small_gemm_shape(M, N, K)) return 4;
  const Plan p = make_plan(M, N, K);
  return p.split > 1 ? (int64_t)p.split * M * N : 4;
}

template <int KIND>
static int launch_gemm(const char* name, GemmArgs g, float* ws, hipStream_t stream) {
  if (small_gemm_shape(g.M, g.N, g.K)) {
    const dim3 grid((unsigned)ceil_div64(g.N, 16), (unsigned)ceil_div64(g.M, 16));
    DVAE_WORK(2.0 * (double)g.M * g.N * g.K, DVAE_UNIT_FLOP);
    if (dvae_mfma_precision() == DVAE_PRECISION_BF16)
      DVAE_LAUNCH((small_gemm_kernel<KIND, true>), grid, dim3(64 * SG_WAVES), 0, stream, g);
    else
      DVAE_LAUNCH((small_gemm_kernel<KIND, false>), grid, dim3(64 * SG_WAVES), 0, stream, g);
    DVAE_LAUNCH_CHECK(name);
    return 0;
  }
  const Plan p = make_plan(g.M, g.N, g.K);
  g.ktiles_per_split = p.ktiles_per_split;
  g.slab = p.split > 1 ? ws : nullptr;
  if (p.split > 1 && ws == nullptr) {
    dvae_set_error("%s: split-K needs a workspace of %ld floats", name, (long)((int64_t)p.split * g.M * g.N));
    return -EINVAL;
  }
  const dim3 grid((unsigned)p.tn, (unsigned)p.tm, (unsigned)p.split);
  constexpr bool AKC = KIND != G_TN, BKC = KIND == G_NT;
  const int prec = dvae_mfma_precision();
  typedef Tile<32, 128, 1, 4, AKC, BKC> Skinny;
  typedef Tile<128, 128, 2, 2, AKC, BKC> Big;
  int64_t rows = g.M;                  // rows of C the next launch computes
#define GEMM_LAUNCH(TILE, KIND_, GRID)                                                                   \
  {                                                                                                      \
    DVAE_WORK(2.0 * (double)rows * g.N * g.K, DVAE_UNIT_FLOP);                                           \
    if (prec == DVAE_PRECISION_BF16)                                                                     \
      DVAE_LAUNCH((gemm_kernel<TILE, KIND_, PREC_BF16>), GRID, dim3(256), 0, stream, g);          \
    else if (prec == DVAE_PRECISION_F32_SPLIT)                                                           \
      DVAE_LAUNCH((gemm_kernel<TILE, KIND_, PREC_SPLIT>), GRID, dim3(256), 0, stream, g);         \
    else                                                                                                 \
      DVAE_LAUNCH((gemm_kernel<TILE, KIND_, PREC_F32>), GRID, dim3(256), 0, stream, g);           \
  }
  g.tm0 = 0;
  {
    static const bool narrow_only = [] { const char* v = getenv("DVAE_GEMM_NARROW_STORES"); return v && v[0] == '1'; }();
    float* out = g.slab ? g.slab : g.C;
    const int64_t ldo = g.slab ? g.N : g.ldc;
    g.wide_rows = !narrow_only && ldo % 4 == 0 && aligned16(out) && (g.slab != nullptr || g.bias == nullptr || aligned16(g.bias)) &&
                  (g.slab == nullptr || (g.M * g.N) % 4 == 0);
  }
  // tiles that fill the 512 workgroup slots a whole number of times and then leave between a fifth and
  // seven tenths of a round: the remaining rows of tiles go to a second launch with 64-wide tiles,
  // which is a shorter last round than a mostly idle one
  const int64_t tiles = p.tm * p.tn, full = tiles / 512, rest = tiles % 512;
  if (KIND == G_NT && !p.small_m && p.split == 1 && full >= 1 && 5 * rest >= 512 && 10 * rest <= 7 * 512) {
    const int64_t head_tm = full * 512 / p.tn;
    if (head_tm > 0 && head_tm < p.tm) {
      typedef Tile<128, 128, 2, 2, true, true> Wide;
      typedef Tile<128, 64, 4, 1, true, true> Narrow;
      const dim3 head((unsigned)p.tn, (unsigned)head_tm, 1);
      const dim3 tail((unsigned)ceil_div64(g.N, 64), (unsigned)(p.tm - head_tm), 1);
      rows = head_tm * 128;
      GEMM_LAUNCH(Wide, G_NT, head)
      g.tm0 = (int)head_tm;
      rows = g.M - head_tm * 128;
      GEMM_LAUNCH(Narrow, G_NT, tail)
      DVAE_LAUNCH_CHECK(name);
      return 0;
    }
  }
  // A launch takes as long as the busiest CU: ceil(tiles / 256) tiles of 128 x 128.  Tiles half as wide halve the
  // unit (at a little less reuse per tile), which shortens launches of one to two tiles per CU: 304 tiles are
  // two rounds of wide tiles but three half-rounds of narrow ones.
  typedef Tile<128, 64, 4, 1, AKC, BKC> Half;
  const int64_t wide_rounds = (tiles + 255) / 256, half_rounds = (2 * tiles + 255) / 256;
  const bool narrow = !p.small_m && p.split == 1 && g.N % 64 == 0 &&
                      1.08 * 0.5 * (double)half_rounds < 0.95 * (double)wide_rounds;
  if (p.small_m) GEMM_LAUNCH(Skinny, KIND, grid)
  else if (narrow) GEMM_LAUNCH(Half, KIND, dim3((unsigned)ceil_div64(g.N, 64), (unsigned)p.tm, 1))
  else {
    g.group_n = group_column_tiles(g.M, g.N, g.K, p.tn, p.tn * p.tm * p.split);
    GEMM_LAUNCH(Big, KIND, grid)
  }
#undef GEMM_LAUNCH
  DVAE_LAUNCH_CHECK(name);
  if (p.split > 1) {
    // four columns per thread when rows are whole float4s in the slab (16-byte aligned with its M*N
    // stride) -- the output pointer is only ever addressed per element
    const bool v4 = g.N % 4 == 0 && aligned16(ws);
    int64_t nb = ceil_div64(g.M * g.N / (v4 ? 4 : 1), 256);
    if (nb > 2048) nb = 2048;
    if (v4)
      DVAE_LAUNCH(splitk_reduce_kernel<4>, dim3((unsigned)nb), dim3(256), 0, stream, (const float*)ws, p.split,
                         g.bias, g.act, g.accumulate, g.C, g.ldc, g.M, g.N);
    else
      DVAE_LAUNCH(splitk_reduce_kernel<1>, dim3((unsigned)nb), dim3(256), 0, stream, (const float*)ws, p.split,
                         g.bias, g.act, g.accumulate, g.C, g.ldc, g.M, g.N);
    DVAE_LAUNCH_CHECK(name);
  }
  return 0;
}


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
