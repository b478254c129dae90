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
#pragma once
#include "common.h"

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
// position inside an RC tile with R rows: (k row, 4-float chunk of the row index)
template <int R>
__device__ __forceinline__ void rc_pos(int v, int& kr, int& c4) {
  const int L = v * 256 + threadIdx.x;
  kr = L / (R / 4);
  c4 = L % (R / 4);
}

template <class T>
__device__ __forceinline__ void store_a_tile(float* lds, const float4 (&r)[T::A_V4]) {
#pragma unroll
  for (int v = 0; v < T::A_V4; ++v) {
    if (T::AKC) {
      int row, ch;
      kc_pos(v, row, ch);
      *reinterpret_cast<float4*>(lds + row * T::ALD + 4 * ch) = r[v];
    } else {
      int kr, c4;
      rc_pos<T::BM>(v, kr, c4);
      *reinterpret_cast<float4*>(lds + kr * T::ALD + 4 * c4) = r[v];
    }
  }
}

template <class T>
__device__ __forceinline__ void store_b_tile(float* lds, const float4 (&r)[T::B_V4]) {
#pragma unroll
  for (int v = 0; v < T::B_V4; ++v) {
    if (T::BKC) {
      int row, ch;
      kc_pos(v, row, ch);
      *reinterpret_cast<float4*>(lds + row * T::BLD + 4 * ch) = r[v];
    } else {
      int kr, c4;
      rc_pos<T::BN>(v, kr, c4);
      *reinterpret_cast<float4*>(lds + kr * T::BLD + 4 * c4) = r[v];
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

// Main loop over K tiles [kt_begin, kt_end).  LA / LB provide
//   void issue(int kt, float4 (&regs)[V4])   start the global loads of K tile kt; branch-free, and
//                                            nothing in it may depend on a loaded value
//   void finish(float4 (&regs)[V4])          load-time transform and zero mask, applied just before
//                                            the registers go to LDS
// so the loads of tile kt+1 stay in flight for the whole multiply of tile kt.
template <class T, class LA, class LB>
__device__ __forceinline__ void gemm_mainloop(LA& la, LB& lb, int kt_begin, int kt_end, float* smem,
                                              f32x16 (&acc)[T::TM][T::TN]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / T::WGN, wn = wave % T::WGN;
  float4 ra[T::A_V4], rb[T::B_V4];
  if (kt_begin >= kt_end) return;
  la.issue(kt_begin, ra);
  lb.issue(kt_begin, rb);
  la.finish(ra);
  lb.finish(rb);
  store_a_tile<T>(smem, ra);
  store_b_tile<T>(smem + T::A_FLOATS, rb);
  __syncthreads();
  int cur = 0;
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const bool more = kt + 1 < kt_end;
    if (more) {
      la.issue(kt + 1, ra);
      lb.issue(kt + 1, rb);
    }
    const float* st = smem + cur * T::STAGE_FLOATS;
    mma_stage<T>(st, st + T::A_FLOATS, acc, wm, wn, lane);
    if (more) {
      float* nx = smem + (cur ^ 1) * T::STAGE_FLOATS;
      la.finish(ra);
      lb.finish(rb);
      store_a_tile<T>(nx, ra);
      store_b_tile<T>(nx + T::A_FLOATS, rb);
    }
    __syncthreads();
    cur ^= 1;
  }
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
__device__ __forceinline__ float4 zero_unless(float4 v, bool keep) {
  return keep ? v : make_float4(0.f, 0.f, 0.f, 0.f);
}

// KC: tile rows are matrix rows (R of them starting at row0), k along the contiguous dimension.
// Everything that does not move with the K tile (row pointer, row validity) is set up once.
template <int V4>
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
  __device__ __forceinline__ void issue(int kt, float4 (&r)[V4]) {
    const int k0 = kt * 32;
    mask = (k0 + kofs < K) ? rowok : 0u;
#pragma unroll
    for (int v = 0; v < V4; ++v) {
      const float* p = (mask >> v) & 1u ? rowp[v] + k0 : base;
      r[v] = *reinterpret_cast<const float4*>(p);
    }
  }
  __device__ __forceinline__ void finish(float4 (&r)[V4]) const {
#pragma unroll
    for (int v = 0; v < V4; ++v) r[v] = zero_unless(r[v], (mask >> v) & 1u);
  }
};

// RC: tile "rows" index the contiguous dimension of memory (R of them starting at col0); k walks
// the strided dimension.  256 % (R/4) == 0, so a thread keeps its column for every chunk it loads.
template <int V4, int R>
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
    colp = base + (colok ? (int64_t)kr0 * ld + c : 0);
  }
  __device__ __forceinline__ void issue(int kt, float4 (&r)[V4]) {
    mask = 0;
#pragma unroll
    for (int v = 0; v < V4; ++v) {
      const int kk = kt * 32 + v * (1024 / R);       // uniform part of the k index
      const bool ok = colok && kk + kr0 < K;
      const float* p = ok ? colp + (int64_t)kk * ld : base;
      r[v] = *reinterpret_cast<const float4*>(p);
      mask |= (unsigned)ok << v;
    }
  }
  __device__ __forceinline__ void finish(float4 (&r)[V4]) const {
#pragma unroll
    for (int v = 0; v < V4; ++v) r[v] = zero_unless(r[v], (mask >> v) & 1u);
  }
};
