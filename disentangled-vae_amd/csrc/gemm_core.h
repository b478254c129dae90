// LDS-tiled f32-MFMA GEMM core shared by the Linear, convolution and LSTM kernels.
//
// One workgroup = 256 threads = 4 waves computes a BM x BN tile of C = A * B, K in steps of 32,
// on v_mfma_f32_32x32x2_f32 (exact fp32, one rounding per product).  Global -> registers -> LDS
// staging with the next tile's loads in flight while the current one is multiplied; two LDS stages,
// one barrier per K step.
//
// Each operand tile lives in LDS in one of two images, chosen by how it is contiguous in memory:
//   KC ("k-contiguous"):   lds[row][BK+4]   filled by 16-B stores along k, read by ds_read_b128
//   RC ("row-contiguous"): lds[k][rows+4]   filled by 16-B stores along the row index, read by b32
// The MFMA sums over k, so any k order works as long as A and B agree: inside each block of eight
// k values lane half h (lane>>5) owns k = 4h..4h+3, which is what lets a KC operand be fetched with
// one 16-B LDS read per lane for four consecutive MFMAs.
//
// bf16-operand variant (Core<T, true>): the same loaders and the same fp32 accumulators, but the
// operands are rounded to bf16 (round-to-nearest-even) on their way into LDS and multiplied by
// v_mfma_f32_32x32x16_bf16.  Both LDS images are then k-contiguous rows of 32 bf16 (+8 pad), read
// with one 16-B LDS read per lane per MFMA; a row-contiguous operand is transposed in registers,
// which needs the thread's chunks of one K tile to sit on consecutive k (KADJ loaders).
#pragma once
#include "common.h"

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Staging registers are native 4-vectors, not HIP's float4 struct: arrays of the struct that are
// read component-wise across elements (the bf16 transpose) were left in scratch memory by the compiler.
__device__ __forceinline__ f32x4 ldg4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

template <int BM_, int BN_, int WGM_, int WGN_, bool AKC_, bool BKC_>
struct Tile {
  static constexpr int BM = BM_, BN = BN_, BK = 32, WGM = WGM_, WGN = WGN_;
  static constexpr bool AKC = AKC_, BKC = BKC_;
  static constexpr int TM = BM / (WGM * 32), TN = BN / (WGN * 32);
  static constexpr int ALD = AKC ? BK + 4 : BM + 4;
  static constexpr int BLD = BKC ? BK + 4 : BN + 4;
  static constexpr int A_FLOATS = (AKC ? BM : BK) * ALD;
  static constexpr int B_FLOATS = (BKC ? BN : BK) * BLD;
  static constexpr int STAGE_FLOATS = A_FLOATS + B_FLOATS;
  static constexpr int SMEM_BYTES = 2 * STAGE_FLOATS * 4;
  static constexpr int A_V4 = BM * BK / 4 / 256;  // float4 per thread per A tile
  static constexpr int B_V4 = BN * BK / 4 / 256;
  static_assert(WGM * WGN == 4, "four waves per workgroup");
  static_assert(TM >= 1 && TN >= 1 && A_V4 >= 1 && B_V4 >= 1, "tile too small");
};

// position of float4 number v of this thread inside a KC tile: (row, 4-float chunk of k)
__device__ __forceinline__ void kc_pos(int v, int& row, int& chunk) {
  const int L = v * 256 + threadIdx.x;
  row = L >> 3;
  chunk = L & 7;
}
// position inside an RC tile with R rows: (k row, 4-float chunk of the row index).  Chunk v of a
// thread sits 1024/R k rows after chunk v-1, or (KADJ) on the next k row.
template <int R, bool KADJ = false>
__device__ __forceinline__ void rc_pos(int v, int& kr, int& c4) {
  constexpr int V4 = R / 32;
  c4 = threadIdx.x % (R / 4);
  const int kg = threadIdx.x / (R / 4);
  kr = KADJ ? V4 * kg + v : v * (1024 / R) + kg;
}

template <class T>
__device__ __forceinline__ void store_a_tile(float* lds, const f32x4 (&r)[T::A_V4]) {
#pragma unroll
  for (int v = 0; v < T::A_V4; ++v) {
    if (T::AKC) {
      int row, ch;
      kc_pos(v, row, ch);
      *reinterpret_cast<f32x4*>(lds + row * T::ALD + 4 * ch) = r[v];
    } else {
      int kr, c4;
      rc_pos<T::BM>(v, kr, c4);
      *reinterpret_cast<f32x4*>(lds + kr * T::ALD + 4 * c4) = r[v];
    }
  }
}

template <class T>
__device__ __forceinline__ void store_b_tile(float* lds, const f32x4 (&r)[T::B_V4]) {
#pragma unroll
  for (int v = 0; v < T::B_V4; ++v) {
    if (T::BKC) {
      int row, ch;
      kc_pos(v, row, ch);
      *reinterpret_cast<f32x4*>(lds + row * T::BLD + 4 * ch) = r[v];
    } else {
      int kr, c4;
      rc_pos<T::BN>(v, kr, c4);
      *reinterpret_cast<f32x4*>(lds + kr * T::BLD + 4 * c4) = r[v];
    }
  }
}

// multiply one staged K step (32 deep) into the accumulators
template <class T>
__device__ __forceinline__ void mma_stage(const float* __restrict__ As, const float* __restrict__ Bs,
                                          f32x16 (&acc)[T::TM][T::TN], int wm, int wn, int lane) {
  const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
  for (int q = 0; q < T::BK / 8; ++q) {
    float a[T::TM][4], b[T::TN][4];
#pragma unroll
    for (int i = 0; i < T::TM; ++i) {
      const int row = (wm * T::TM + i) * 32 + l31;
      if (T::AKC) {
        const float4 v = *reinterpret_cast<const float4*>(As + row * T::ALD + 8 * q + 4 * h);
        a[i][0] = v.x; a[i][1] = v.y; a[i][2] = v.z; a[i][3] = v.w;
      } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) a[i][s] = As[(8 * q + 4 * h + s) * T::ALD + row];
      }
    }
#pragma unroll
    for (int j = 0; j < T::TN; ++j) {
      const int row = (wn * T::TN + j) * 32 + l31;
      if (T::BKC) {
        const float4 v = *reinterpret_cast<const float4*>(Bs + row * T::BLD + 8 * q + 4 * h);
        b[j][0] = v.x; b[j][1] = v.y; b[j][2] = v.z; b[j][3] = v.w;
      } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) b[j][s] = Bs[(8 * q + 4 * h + s) * T::BLD + row];
      }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int j = 0; j < T::TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
  }
}

template <class T>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[T::TM][T::TN]) {
#pragma unroll
  for (int i = 0; i < T::TM; ++i)
#pragma unroll
    for (int j = 0; j < T::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
}

// ---- bf16-operand staging ------------------------------------------------------------------
// the nearest bf16-representable values (round to nearest even), as floats
__device__ __forceinline__ f32x4 round_bf16(f32x4 v) {
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  const bf16x2 a = __builtin_convertvector((f32x2_){v[0], v[1]}, bf16x2);
  const bf16x2 b = __builtin_convertvector((f32x2_){v[2], v[3]}, bf16x2);
  const f32x2_ fa = __builtin_convertvector(a, f32x2_), fb = __builtin_convertvector(b, f32x2_);
  return (f32x4){fa[0], fa[1], fb[0], fb[1]};
}

__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
  const f32x2 v = {lo, hi};
  const bf16x2 b = __builtin_convertvector(v, bf16x2);   // v_cvt_pk_bf16_f32, round to nearest even
  uint32_t u;
  __builtin_memcpy(&u, &b, 4);
  return u;
}

#define BF_LD 20   // dwords per LDS row: 32 bf16 + 8 pad = 80 B (16-B reads stay aligned and conflict-free)

// A row holds its 32 bf16 as four 16-B blocks (block b = k 8b..8b+7), stored at block position
// b ^ ((row >> 4) & 3).  The XOR spreads the transposing stores of a row-contiguous operand (lanes
// that hold rows 4 apart at the same k) over all banks; readers undo it with the same XOR.
__device__ __forceinline__ int bf_swz(int row) { return (row >> 4) & 3; }

// one operand tile (ROWS rows x 32 k) into its bf16 image; KC: chunks are 4 consecutive k of one row;
// RC (KADJ loader order): chunk v holds rows 4*c4..4*c4+3 at k = V4*kg + v
template <int ROWS, int V4, bool KC>
__device__ __forceinline__ void store_tile_bf16(uint32_t* lds, const f32x4 (&r)[V4]) {
  if (KC) {
#pragma unroll
    for (int v = 0; v < V4; ++v) {
      int row, ch;
      kc_pos(v, row, ch);
      uint2 w;
      w.x = pack_bf16(r[v].x, r[v].y);
      w.y = pack_bf16(r[v].z, r[v].w);
      *reinterpret_cast<uint2*>(lds + row * BF_LD + 2 * (ch ^ (2 * bf_swz(row)))) = w;
    }
  } else {
    static_assert(KC || V4 == ROWS / 32, "row-contiguous tile: V4 chunks per thread");
    const int c4 = threadIdx.x % (ROWS / 4), kg = threadIdx.x / (ROWS / 4);
    const f32x4 r0 = r[0], r1 = r[1 % V4], r2 = r[2 % V4], r3 = r[3 % V4];
    // component C of the chunks = row 4*c4 + J at the thread's k values
#define BF_ROW(J, C)                                                                                      \
  {                                                                                                       \
    const int row = 4 * c4 + J;                                                                           \
    const int sw = 2 * bf_swz(row); /* in units of 4 k (8 B) */                                           \
    if (V4 == 4) { /* k = 4kg..4kg+3: one 8-B quad */                                                     \
      uint2 w;                                                                                            \
      w.x = pack_bf16(r0.C, r1.C);                                                                        \
      w.y = pack_bf16(r2.C, r3.C);                                                                        \
      *reinterpret_cast<uint2*>(lds + row * BF_LD + 2 * (kg ^ sw)) = w;                                   \
    } else if (V4 == 2) { /* k = 2kg, 2kg+1: dword kg of the row */                                       \
      lds[row * BF_LD + 2 * ((kg >> 1) ^ sw) + (kg & 1)] = pack_bf16(r0.C, r1.C);                         \
    } else { /* k = kg: one halfword */                                                                   \
      reinterpret_cast<uint16_t*>(lds)[row * (2 * BF_LD) + 4 * ((kg >> 2) ^ sw) + (kg & 3)] =             \
          (uint16_t)pack_bf16(r0.C, 0.f);                                                                 \
    }                                                                                                     \
  }
    BF_ROW(0, x)
    BF_ROW(1, y)
    BF_ROW(2, z)
    BF_ROW(3, w)
#undef BF_ROW
  }
}

template <class T>
__device__ __forceinline__ void mma_stage_bf16(const uint32_t* __restrict__ As, const uint32_t* __restrict__ Bs,
                                               f32x16 (&acc)[T::TM][T::TN], int wm, int wn, int lane) {
  const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    bf16x8 a[T::TM], b[T::TN];
#pragma unroll
    for (int i = 0; i < T::TM; ++i) {
      const int row = (wm * T::TM + i) * 32 + l31;
      a[i] = *reinterpret_cast<const bf16x8*>(As + row * BF_LD + 4 * ((2 * q + h) ^ bf_swz(row)));
    }
#pragma unroll
    for (int j = 0; j < T::TN; ++j) {
      const int row = (wn * T::TN + j) * 32 + l31;
      b[j] = *reinterpret_cast<const bf16x8*>(Bs + row * BF_LD + 4 * ((2 * q + h) ^ bf_swz(row)));
    }
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
      for (int j = 0; j < T::TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
  }
}

// ---- fp32 products from bf16 MFMAs (PREC_SPLIT) -------------------------------------------------
// An fp32 value is split into three bf16 terms hi + mid + lo (each the nearest bf16 of what the terms
// before it leave over; the first two remainders are exact in fp32, the last rounding is below 2^-25
// of the value), so a product a*b is the sum of nine bf16 x bf16 products, each exact in the fp32
// accumulator.  The six largest are formed (hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi); the three
// dropped ones (mid*lo, lo*mid, lo*lo) are below 2^-24 of |a*b|, the size of fp32's own rounding.
// Operands stay fp32 in memory and in LDS (the fp32 images and loaders are used unchanged); a lane
// splits its fragments after reading them, eight k per operand row for v_mfma_f32_32x32x16_bf16.
// one pair of fp32 values -> the packed bf16 pairs of its three terms
__device__ __forceinline__ float bf_lo_as_float(uint32_t pk) { return __uint_as_float(pk << 16); }
__device__ __forceinline__ float bf_hi_as_float(uint32_t pk) { return __uint_as_float(pk & 0xffff0000u); }

__device__ __forceinline__ void split_pair(float a0, float a1, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
  const f32x2 a = {a0, a1};
  const uint32_t h = pack_bf16(a[0], a[1]);
  const f32x2 r = a - (f32x2){bf_lo_as_float(h), bf_hi_as_float(h)};         // exact
  const uint32_t m = pack_bf16(r[0], r[1]);
  const f32x2 t = r - (f32x2){bf_lo_as_float(m), bf_hi_as_float(m)};         // exact
  hi = h;
  mid = m;
  lo = pack_bf16(t[0], t[1]);
}

// Raw fp32 fragments of the 16-deep group g of a staged K step: eight k per operand row and lane.
// Which eight k of a group a lane holds is free as long as A and B agree: lane half h takes
// k = 8q + 4h .. +3 for the group's two q.  Fragment f < TM is A's row block f, the others are B's.
template <class T>
__device__ __forceinline__ void load_group(const float* __restrict__ As, const float* __restrict__ Bs, int g, int wm,
                                           int wn, int l31, int h, float (&v)[T::TM + T::TN][8]) {
#pragma unroll
  for (int f = 0; f < T::TM + T::TN; ++f) {
    const bool isa = f < T::TM;
    const int row = ((isa ? wm * T::TM + f : wn * T::TN + (f - T::TM))) * 32 + l31;
    const float* S = isa ? As : Bs;
    const int ld = isa ? T::ALD : T::BLD;
    const bool kc = isa ? T::AKC : T::BKC;
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
      const int q = 2 * g + qq;
      if (kc) {
        const float4 t = *reinterpret_cast<const float4*>(S + row * ld + 8 * q + 4 * h);
        v[f][4 * qq] = t.x; v[f][4 * qq + 1] = t.y; v[f][4 * qq + 2] = t.z; v[f][4 * qq + 3] = t.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[f][4 * qq + e] = S[(8 * q + 4 * h + e) * ld + row];
      }
    }
  }
}

// split terms of a group's fragments: w[term][fragment][pair], term 0 = hi, 1 = mid, 2 = lo
template <int NF>
struct SplitFrags {
  uint32_t w[3][NF][4];
  __device__ __forceinline__ bf16x8 get(int term, int f) const {
    bf16x8 o;
    __builtin_memcpy(&o, w[term][f], 16);
    return o;
  }
};

// One staged K step (32 deep, fp32 images) as two 16-deep bf16 MFMA groups, six products each, smallest
// terms first.  A wave issues in program order and an MFMA holds it for the time of eight vector
// instructions, so the second group's splitting is written between the first group's MFMAs, one pair of
// values per MFMA, with scheduling fences that keep the compiler from regrouping them; only the first
// group's split runs with the matrix pipe idle.
template <class T>
__device__ __forceinline__ void mma_stage_split(const float* __restrict__ As, const float* __restrict__ Bs,
                                                f32x16 (&acc)[T::TM][T::TN], int wm, int wn, int lane) {
  static_assert(T::BK == 32, "two 16-deep groups per K step");
  constexpr int NF = T::TM + T::TN, NM = 6 * T::TM * T::TN, NT = 4 * NF;
  constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
  const int l31 = lane & 31, h = lane >> 5;
  float v[NF][8];
  SplitFrags<NF> s0, s1;
  load_group<T>(As, Bs, 0, wm, wn, l31, h, v);
#pragma unroll
  for (int t = 0; t < NT; ++t)
    split_pair(v[t / 4][2 * (t % 4)], v[t / 4][2 * (t % 4) + 1], s0.w[0][t / 4][t % 4], s0.w[1][t / 4][t % 4],
               s0.w[2][t / 4][t % 4]);
  load_group<T>(As, Bs, 1, wm, wn, l31, h, v);
  __builtin_amdgcn_sched_barrier(0);
  constexpr int PER = (NT + NM - 1) / NM;          // pairs to split per MFMA
#pragma unroll
  for (int n = 0; n < NM; ++n) {
    const int ty = n / (T::TM * T::TN), i = (n % (T::TM * T::TN)) / T::TN, j = n % T::TN;
    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(s0.get(PA[ty], i), s0.get(PB[ty], T::TM + j), acc[i][j], 0, 0, 0);
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int t = n * PER + u;
      if (t < NT)
        split_pair(v[t / 4][2 * (t % 4)], v[t / 4][2 * (t % 4) + 1], s1.w[0][t / 4][t % 4], s1.w[1][t / 4][t % 4],
                   s1.w[2][t / 4][t % 4]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int n = 0; n < NM; ++n) {
    const int ty = n / (T::TM * T::TN), i = (n % (T::TM * T::TN)) / T::TN, j = n % T::TN;
    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(s1.get(PA[ty], i), s1.get(PB[ty], T::TM + j), acc[i][j], 0, 0, 0);
  }
}

// The operand precisions behind one interface.  STAGE is in 4-byte units.
#define PREC_F32 0
#define PREC_BF16 1
#define PREC_SPLIT 2
template <class T, int PREC>
struct Core;

template <class T>
struct Core<T, PREC_SPLIT> {
  static constexpr int STAGE = T::STAGE_FLOATS;
  static constexpr bool KADJ = false;
  static __device__ __forceinline__ void store_a(float* st, const f32x4 (&r)[T::A_V4]) { store_a_tile<T>(st, r); }
  static __device__ __forceinline__ void store_b(float* st, const f32x4 (&r)[T::B_V4]) {
    store_b_tile<T>(st + T::A_FLOATS, r);
  }
  static __device__ __forceinline__ void mma(const float* st, f32x16 (&acc)[T::TM][T::TN], int wm, int wn, int lane) {
    mma_stage_split<T>(st, st + T::A_FLOATS, acc, wm, wn, lane);
  }
};

template <class T>
struct Core<T, PREC_F32> {
  static constexpr int STAGE = T::STAGE_FLOATS;
  static constexpr bool KADJ = false;
  static __device__ __forceinline__ void store_a(float* st, const f32x4 (&r)[T::A_V4]) { store_a_tile<T>(st, r); }
  static __device__ __forceinline__ void store_b(float* st, const f32x4 (&r)[T::B_V4]) {
    store_b_tile<T>(st + T::A_FLOATS, r);
  }
  static __device__ __forceinline__ void mma(const float* st, f32x16 (&acc)[T::TM][T::TN], int wm, int wn, int lane) {
    mma_stage<T>(st, st + T::A_FLOATS, acc, wm, wn, lane);
  }
};

template <class T>
struct Core<T, PREC_BF16> {
  static constexpr int A_DW = T::BM * BF_LD, B_DW = T::BN * BF_LD;
  static constexpr int STAGE = A_DW + B_DW;
  static constexpr bool KADJ = true;
  static __device__ __forceinline__ void store_a(float* st, const f32x4 (&r)[T::A_V4]) {
    store_tile_bf16<T::BM, T::A_V4, T::AKC>(reinterpret_cast<uint32_t*>(st), r);
  }
  static __device__ __forceinline__ void store_b(float* st, const f32x4 (&r)[T::B_V4]) {
    store_tile_bf16<T::BN, T::B_V4, T::BKC>(reinterpret_cast<uint32_t*>(st) + A_DW, r);
  }
  static __device__ __forceinline__ void mma(const float* st, f32x16 (&acc)[T::TM][T::TN], int wm, int wn, int lane) {
    const uint32_t* s = reinterpret_cast<const uint32_t*>(st);
    mma_stage_bf16<T>(s, s + A_DW, acc, wm, wn, lane);
  }
};

// The multiply stage runs at a raised issue priority, so that a co-resident workgroup's staging instructions do
// not delay this wave's MFMAs (1-3 % on the NT / NN GEMMs and the small conv layers, neutral elsewhere).
// -DDVAE_MMA_PRIO=0 builds without it.
#ifndef DVAE_MMA_PRIO
#define DVAE_MMA_PRIO 1
#endif
// experiment: the plain main loop on ONE LDS stage and two barriers per K step (three workgroups per CU)
#ifndef DVAE_GEMM_ONE_STAGE
#define DVAE_GEMM_ONE_STAGE 0
#endif
#if DVAE_MMA_PRIO
#define MMA_PRIO_UP() __builtin_amdgcn_s_setprio(DVAE_MMA_PRIO)
#define MMA_PRIO_DOWN() __builtin_amdgcn_s_setprio(0)
#else
#define MMA_PRIO_UP()
#define MMA_PRIO_DOWN()
#endif

// Main loop over K tiles [kt_begin, kt_end).  LA / LB provide
//   void issue(int kt, float4 (&regs)[V4])   start the global loads of K tile kt; branch-free, and
//                                            nothing in it may depend on a loaded value
//   void finish(float4 (&regs)[V4])          load-time transform and zero mask, applied just before
//                                            the registers go to LDS
// so the loads of tile kt+1 stay in flight for the whole multiply of tile kt.
template <class T, int BF, class LA, class LB>
__device__ __forceinline__ void gemm_mainloop(LA& la, LB& lb, int kt_begin, int kt_end, float* smem,
                                              f32x16 (&acc)[T::TM][T::TN]) {
  typedef Core<T, BF> C;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / T::WGN, wn = wave % T::WGN;
  f32x4 ra[T::A_V4], rb[T::B_V4];
  if (kt_begin >= kt_end) return;
  la.issue(kt_begin, ra);
  lb.issue(kt_begin, rb);
  la.finish(ra);
  lb.finish(rb);
  C::store_a(smem, ra);
  C::store_b(smem, rb);
  __syncthreads();
  int cur = 0;
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const bool more = kt + 1 < kt_end;
    if (more) {
      la.issue(kt + 1, ra);
      lb.issue(kt + 1, rb);
    }
    MMA_PRIO_UP();
    C::mma(smem + cur * C::STAGE, acc, wm, wn, lane);
    MMA_PRIO_DOWN();
#if DVAE_GEMM_ONE_STAGE
    __syncthreads();          // one LDS stage: every wave has read it before it is overwritten
    if (more) {
      float* nx = smem;
#else
    if (more) {
      float* nx = smem + (cur ^ 1) * C::STAGE;
#endif
      la.finish(ra);
      lb.finish(rb);
      C::store_a(nx, ra);
      C::store_b(nx, rb);
    }
    __syncthreads();
#if !DVAE_GEMM_ONE_STAGE
    cur ^= 1;
#endif
  }
}

// The same loop with the loads issued two K tiles ahead (two register sets): for tiles whose multiply
// phase (16-32 MFMAs per wave) is shorter than the memory latency.  Loaders additionally provide
//   Tag tag() / void set_tag(Tag)    the per-issue state finish() depends on
template <class T, int BF, class LA, class LB>
__device__ __forceinline__ void gemm_mainloop_ahead2(LA& la, LB& lb, int kt_begin, int kt_end, float* smem,
                                                     f32x16 (&acc)[T::TM][T::TN]) {
  typedef Core<T, BF> C;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / T::WGN, wn = wave % T::WGN;
  const int n = kt_end - kt_begin;
  if (n <= 0) return;
  f32x4 ra0[T::A_V4], rb0[T::B_V4], ra1[T::A_V4], rb1[T::B_V4];
  typename LA::Tag ta0, ta1;
  typename LB::Tag tb0, tb1;
  la.issue(kt_begin, ra0);
  lb.issue(kt_begin, rb0);
  ta0 = la.tag();
  tb0 = lb.tag();
  ta1 = ta0;
  tb1 = tb0;
  if (n > 1) {
    la.issue(kt_begin + 1, ra1);
    lb.issue(kt_begin + 1, rb1);
    ta1 = la.tag();
    tb1 = lb.tag();
  }
  la.set_tag(ta0);
  lb.set_tag(tb0);
  la.finish(ra0);
  lb.finish(rb0);
  C::store_a(smem, ra0);
  C::store_b(smem, rb0);
  __syncthreads();
  // step i: tile i is in LDS stage i&1, tile i+1 in register set (i+1)&1; set i&1 is free for tile i+2
#define AHEAD2_STEP(I, RA_FREE, RB_FREE, TA_FREE, TB_FREE, RA_NEXT, RB_NEXT, TA_NEXT, TB_NEXT, SIDX)   \
  {                                                                                                   \
    if ((I) + 2 < n) {                                                                                \
      la.issue(kt_begin + (I) + 2, RA_FREE);                                                          \
      lb.issue(kt_begin + (I) + 2, RB_FREE);                                                          \
      TA_FREE = la.tag();                                                                             \
      TB_FREE = lb.tag();                                                                             \
    }                                                                                                 \
    MMA_PRIO_UP();                                                                                    \
    C::mma(smem + (SIDX) * C::STAGE, acc, wm, wn, lane);                                             \
    MMA_PRIO_DOWN();                                                                                  \
    if ((I) + 1 < n) {                                                                                \
      float* nx = smem + ((SIDX) ^ 1) * C::STAGE;                                                     \
      la.set_tag(TA_NEXT);                                                                            \
      lb.set_tag(TB_NEXT);                                                                            \
      la.finish(RA_NEXT);                                                                             \
      lb.finish(RB_NEXT);                                                                             \
      C::store_a(nx, RA_NEXT);                                                                        \
      C::store_b(nx, RB_NEXT);                                                                        \
    }                                                                                                 \
    __syncthreads();                                                                                  \
  }
  for (int i = 0; i < n; i += 2) {
    AHEAD2_STEP(i, ra0, rb0, ta0, tb0, ra1, rb1, ta1, tb1, 0)
    if (i + 1 < n) AHEAD2_STEP(i + 1, ra1, rb1, ta1, tb1, ra0, rb0, ta0, tb0, 1)
  }
#undef AHEAD2_STEP
}

// visit every accumulator element of this lane: f(row_in_tile, col_in_tile, value&)
template <class T, class F>
__device__ __forceinline__ void for_each_acc(f32x16 (&acc)[T::TM][T::TN], F f) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / T::WGN, wn = wave % T::WGN;
  const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
  for (int i = 0; i < T::TM; ++i)
#pragma unroll
    for (int j = 0; j < T::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (wm * T::TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int col = (wn * T::TN + j) * 32 + l31;
        f(row, col, (float)acc[i][j][r]);
      }
}

// Plain strided-matrix loaders ---------------------------------------------------------------
// Out-of-range chunks are read from `base` itself (always valid and 16-byte aligned) and zeroed in
// finish(), which keeps issue() free of branches.
__device__ __forceinline__ f32x4 zero_unless(f32x4 v, bool keep) {
  return keep ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
}

// KC: tile rows are matrix rows (R of them starting at row0), k along the contiguous dimension.
// Everything that does not move with the K tile (row pointer, row validity) is set up once.
// WHOLE: the caller vouches that every row of the tile exists and that K is a multiple of 32; issue() is then
// nothing but the loads and finish() nothing at all (with bf16 operands the multiplies of a K tile are so short
// that the address selects and zero masks of the general form are a fifth of the step).
template <int V4, bool WHOLE = false>
struct KCLoader {
  const float* base;
  const float* rowp[V4];
  int64_t K;
  int kofs;
  unsigned rowok, mask;
  __device__ __forceinline__ KCLoader(const float* base_, int64_t ld, int64_t rows, int64_t K_, int64_t row0)
      : base(base_), K(K_) {
    rowok = 0;
    kofs = 4 * (threadIdx.x & 7);
#pragma unroll
    for (int v = 0; v < V4; ++v) {
      int row, ch;
      kc_pos(v, row, ch);
      const int64_t gr = row0 + row;
      const bool ok = gr < rows;
      rowp[v] = base + (ok ? gr * ld : 0) + kofs;
      rowok |= (unsigned)ok << v;
    }
  }
  __device__ __forceinline__ void issue(int kt, f32x4 (&r)[V4]) { issue_at(kt * 32, r); }
  // the K tile that starts at k0 (callers that walk K in an order of their own)
  __device__ __forceinline__ void issue_at(int k0, f32x4 (&r)[V4]) {
    if (WHOLE) {
#pragma unroll
      for (int v = 0; v < V4; ++v) r[v] = ldg4(rowp[v] + k0);
      return;
    }
    mask = (k0 + kofs < K) ? rowok : 0u;
#pragma unroll
    for (int v = 0; v < V4; ++v) {
      const float* p = (mask >> v) & 1u ? rowp[v] + k0 : base;
      r[v] = ldg4(p);
    }
  }
  __device__ __forceinline__ void finish(f32x4 (&r)[V4]) const {
    if (WHOLE) return;
#pragma unroll
    for (int v = 0; v < V4; ++v) r[v] = zero_unless(r[v], (mask >> v) & 1u);
  }
  // what finish() needs from the matching issue(), for main loops that keep two K tiles in flight
  struct Tag { unsigned mask; };
  __device__ __forceinline__ Tag tag() const { return Tag{mask}; }
  __device__ __forceinline__ void set_tag(const Tag& t) { mask = t.mask; }
};

// RC: tile "rows" index the contiguous dimension of memory (R of them starting at col0); k walks
// the strided dimension.  256 % (R/4) == 0, so a thread keeps its column for every chunk it loads.
template <int V4, int R, bool KADJ = false, bool WHOLE = false>
struct RCLoader {
  static_assert(256 % (R / 4) == 0, "column position must not depend on the chunk number");
  const float* base;
  const float* colp;
  int64_t ld, K;
  int kr0;
  bool colok;
  unsigned mask;
  __device__ __forceinline__ RCLoader(const float* base_, int64_t ld_, int64_t cols, int64_t K_, int64_t col0)
      : base(base_), ld(ld_), K(K_) {
    kr0 = threadIdx.x / (R / 4);
    const int64_t c = col0 + 4 * (threadIdx.x % (R / 4));
    colok = c < cols;
    colp = base + (colok ? (int64_t)((KADJ ? V4 : 1) * kr0) * ld + c : 0);
  }
  __device__ __forceinline__ void issue(int kt, f32x4 (&r)[V4]) {
    if (WHOLE) {
#pragma unroll
      for (int v = 0; v < V4; ++v) r[v] = ldg4(colp + (int64_t)(kt * 32 + (KADJ ? v : v * (1024 / R))) * ld);
      return;
    }
    mask = 0;
#pragma unroll
    for (int v = 0; v < V4; ++v) {
      // k index of chunk v = kk + (KADJ ? V4 : 1) * kr0; colp already points at row kr0
      const int kk = kt * 32 + (KADJ ? v : v * (1024 / R));
      const bool ok = colok && kk + (KADJ ? V4 : 1) * kr0 < K;
      const float* p = ok ? colp + (int64_t)kk * ld : base;
      r[v] = ldg4(p);
      mask |= (unsigned)ok << v;
    }
  }
  __device__ __forceinline__ void finish(f32x4 (&r)[V4]) const {
    if (WHOLE) return;
#pragma unroll
    for (int v = 0; v < V4; ++v) r[v] = zero_unless(r[v], (mask >> v) & 1u);
  }
  // what finish() needs from the matching issue(), for main loops that keep two K tiles in flight
  struct Tag { unsigned mask; };
  __device__ __forceinline__ Tag tag() const { return Tag{mask}; }
  __device__ __forceinline__ void set_tag(const Tag& t) { mask = t.mask; }
};
