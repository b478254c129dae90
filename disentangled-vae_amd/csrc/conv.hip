// 4x4 / stride 2 / pad 1 convolution family as implicit GEMMs on the shared f32-MFMA core.
//
//   conv forward         y[m=(n,ho,wo)][co] = sum_{kh,kw,ci} act(x)[n,2ho-1+kh,2wo-1+kw,ci] * wp[co][kh][kw][ci]
//   transposed forward   four sub-pixel phases (ph,pw); for each, a 2x2-tap conv at input resolution:
//                        y[n,2a+ph,2b+pw][co] = sum_{th,tw,ci} act(x)[n,a+ph-th,b+pw-tw,ci] * wp[ph,pw][co][th][tw][ci]
//   weight gradient      dw[c1][c2][kh][kw] = sum_{m=(n,a,b)} P[m][c1] * Q[n,2a-1+kh,2b-1+kw][c2]
//
// Activations are NHWC, so a K slice (fixed tap, run of input channels) is contiguous in memory.
// The previous layer's BatchNorm + LeakyReLU is applied while the operand is loaded; zero padding
// is applied after that transform.  The forward kernels also emit per-workgroup partial sums of the
// raw output and of its square per output channel (BatchNorm statistics), each workgroup walking
// several M tiles so the number of partial rows stays bounded.
//
// Spatial sizes must be powers of two (64x64 and 128x128 frames and their halvings).
#include "gemm_core.h"
#include "../../include/dvae_hip.h"

#define ST(s) ((hipStream_t)(s))
#define CONV_MAX_PARTS 1024
// the tiled forward kernels' filter loader without selects and masks when every column tile lies inside the filter
// (-DDVAE_CONV_WHOLE_B=0: always the general loader)
#ifndef DVAE_CONV_WHOLE_B
#define DVAE_CONV_WHOLE_B 1
#endif
// timing-only build (-DDVAE_CONV_EPILOGUE_PROBE=1): the tiled forward kernels without their output stores and
// without the loads of the BatchNorm-backward operand, i.e. what the epilogue's memory traffic costs
#ifndef DVAE_CONV_EPILOGUE_PROBE
#define DVAE_CONV_EPILOGUE_PROBE 0
#endif

static inline int ilog2(int64_t v) {
  int l = 0;
  while ((1LL << l) < v) ++l;
  return l;
}
static inline bool is_pow2(int64_t v) { return v > 0 && (v & (v - 1)) == 0; }
static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// ------------------------------------------------------------------------------------------------
// weight packing
// ------------------------------------------------------------------------------------------------
__global__ void pack_conv_w_kernel(const float* __restrict__ w, float* __restrict__ wp, int64_t A, int64_t Bc) {
  const int64_t total = A * Bc * 16;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    // destination index i = (a*16 + tap)*Bc + b
    const int64_t b = i % Bc, t = (i / Bc) % 16, a = i / (Bc * 16);
    wp[i] = w[(a * Bc + b) * 16 + t];
  }
}

__global__ void pack_deconv_w_kernel(const float* __restrict__ w, float* __restrict__ wp, int64_t Kc, int64_t Nc) {
  const int64_t per_phase = Nc * 4 * Kc;
  const int64_t total = 4 * per_phase;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    // destination index i = ((phase*Nc + n)*4 + tap)*Kc + k
    const int64_t k = i % Kc, tap = (i / Kc) % 4, n = (i / (Kc * 4)) % Nc, phase = i / per_phase;
    const int ph = (int)(phase >> 1), pw = (int)(phase & 1), th = (int)(tap >> 1), tw = (int)(tap & 1);
    const int kh = ph == 0 ? (th == 0 ? 1 : 3) : (th == 0 ? 0 : 2);
    const int kw = pw == 0 ? (tw == 0 ? 1 : 3) : (tw == 0 ? 0 : 2);
    wp[i] = w[((k * Nc + n) * 4 + kh) * 4 + kw];
  }
}

extern "C" int dvae_pack_conv_w_f32(const float* w, float* wp, int64_t A, int64_t Bc, dvae_stream_t stream) {
  DVAE_REQUIRE(A > 0 && Bc > 0, "pack_conv_w: empty");
  int64_t nb = ceil_div64(A * Bc * 16, 256);
  if (nb > 2048) nb = 2048;
  DVAE_LAUNCH(pack_conv_w_kernel, dim3((unsigned)nb), dim3(256), 0, ST(stream), w, wp, A, Bc);
  DVAE_LAUNCH_CHECK("pack_conv_w");
  return 0;
}

extern "C" int dvae_pack_deconv_w_f32(const float* w, float* wp, int64_t Kc, int64_t Nc, dvae_stream_t stream) {
  DVAE_REQUIRE(Kc > 0 && Nc > 0, "pack_deconv_w: empty");
  int64_t nb = ceil_div64(Kc * Nc * 16, 256);
  if (nb > 2048) nb = 2048;
  DVAE_LAUNCH(pack_deconv_w_kernel, dim3((unsigned)nb), dim3(256), 0, ST(stream), w, wp, Kc, Nc);
  DVAE_LAUNCH_CHECK("pack_deconv_w");
  return 0;
}

// Several weights in one launch: a step packs every convolution weight of an encoder or decoder (both layouts when
// a backward follows), and at small batch sizes sixteen few-microsecond launches cost more than the packing.
#define PACK_MAX 16
#define PACK_PER_BLOCK 2048          // elements a workgroup packs
struct PackTable {
  int first_block[PACK_MAX + 1];
  const float* w[PACK_MAX];
  float* wp[PACK_MAX];
  int64_t d0[PACK_MAX], d1[PACK_MAX];
  int transposed[PACK_MAX];
  int n;
};

__global__ void __launch_bounds__(256) pack_many_kernel(PackTable t) {
  int it = 0;
  while (it + 1 < t.n && (int)blockIdx.x >= t.first_block[it + 1]) ++it;
  const int64_t lo = (int64_t)(blockIdx.x - t.first_block[it]) * PACK_PER_BLOCK;
  const float* __restrict__ w = t.w[it];
  float* __restrict__ wp = t.wp[it];
  const int64_t d0 = t.d0[it], d1 = t.d1[it];
  const int64_t total = d0 * d1 * 16 < lo + PACK_PER_BLOCK ? d0 * d1 * 16 : lo + PACK_PER_BLOCK;
  if (!t.transposed[it]) {
    for (int64_t i = lo + threadIdx.x; i < total; i += 256) {
      const int64_t b = i % d1, tp = (i / d1) % 16, a = i / (d1 * 16);
      wp[i] = w[(a * d1 + b) * 16 + tp];
    }
  } else {
    const int64_t per_phase = d1 * 4 * d0;
    for (int64_t i = lo + threadIdx.x; i < total; i += 256) {
      const int64_t k = i % d0, tap = (i / d0) % 4, n = (i / (d0 * 4)) % d1, phase = i / per_phase;
      const int ph = (int)(phase >> 1), pw = (int)(phase & 1), th = (int)(tap >> 1), tw = (int)(tap & 1);
      const int kh = ph == 0 ? (th == 0 ? 1 : 3) : (th == 0 ? 0 : 2);
      const int kw = pw == 0 ? (tw == 0 ? 1 : 3) : (tw == 0 ? 0 : 2);
      wp[i] = w[((k * d1 + n) * 4 + kh) * 4 + kw];
    }
  }
}

extern "C" int dvae_pack_w_many_f32(int n, const float* const* w, float* const* wp, const int64_t* d0,
                                    const int64_t* d1, const int* transposed, dvae_stream_t stream) {
  DVAE_REQUIRE(n >= 0 && w && wp && d0 && d1 && transposed, "pack_w_many: null table");
  for (int i = 0; i < n; ++i)
    DVAE_REQUIRE(w[i] && wp[i] && d0[i] > 0 && d1[i] > 0, "pack_w_many: item %d is empty", i);
  for (int first = 0; first < n; first += PACK_MAX) {
    PackTable t;
    t.n = n - first < PACK_MAX ? n - first : PACK_MAX;
    t.first_block[0] = 0;
    for (int i = 0; i < t.n; ++i) {
      const int64_t blocks = ceil_div64(d0[first + i] * d1[first + i] * 16, PACK_PER_BLOCK);
      DVAE_REQUIRE(blocks < (1 << 24), "pack_w_many: item %d is too large", first + i);
      t.first_block[i + 1] = t.first_block[i] + (int)blocks;
      t.w[i] = w[first + i];
      t.wp[i] = wp[first + i];
      t.d0[i] = d0[first + i];
      t.d1[i] = d1[first + i];
      t.transposed[i] = transposed[first + i];
    }
    DVAE_LAUNCH(pack_many_kernel, dim3((unsigned)t.first_block[t.n]), dim3(256), 0, ST(stream), t);
    DVAE_LAUNCH_CHECK("pack_w_many");
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------
// forward kernels
// ------------------------------------------------------------------------------------------------
struct ConvArgs {
  const float* x;
  const float* in_scale;
  const float* in_shift;
  const float* wp;
  float* y;
  float* stats;  // [parts][2][Cout] or nullptr
  int64_t M;     // GEMM rows per phase
  int in_act;
  int H, W;      // input spatial size
  int lh, lw;    // log2 of the row-index grid (conv: Ho,Wo ; deconv: H,W)
  int Cin, Cout, K;
  int mtiles;
  int lc;      // log2(Cin) when Cin is a power of two, else -1
  int cin32;   // Cin % 32 == 0: a K tile never straddles two filter taps
  // BatchNorm-backward statistics instead of forward statistics (bn_y != nullptr): the output v is a
  // gradient w.r.t. the activation a = lrelu(bn_y*bn_scale + bn_shift) of a tensor bn_y shaped like
  // the output; the partial rows then hold sum(dz) and sum(dz*xhat), dz = v*lrelu'(.), xhat =
  // (bn_y - bn_mean)*bn_invstd.  The output itself is written unchanged.
  const float* bn_y;
  const float* bn_scale;
  const float* bn_shift;
  const float* bn_mean;
  const float* bn_invstd;
  // input not yet through its BatchNorm backward (conv_rows_kernel, BNIN): raw output of that layer,
  // the two folded coefficient vectors, and where the finished gradient rows go
  const float* in_y;
  const float* in_a;
  const float* in_b;
  float* dx_out;
  int mt0;          // tiled kernels: first M tile of this launch (a layer may be split over two launches)
  int part0;        // tiled kernels: first statistics partial row of this launch
  // tiled kernels, K split over gridDim.z (few M tiles: every workgroup takes kt_per K tiles and writes its partial
  // output, in output layout, to slab + blockIdx.z * slab_stride; conv_slab_reduce_kernel sums them); nullptr: no split
  float* slab;
  int64_t slab_stride;
  int kt_per;
  float* ws;        // scratch the caller brought (may be nullptr) and its size
  int64_t ws_floats;
  // bf16 storage (bf16-MFMA mode only; row-ring and single-channel kernels): which of the tensors above hold
  // bfloat16 elements instead of floats, DVAE_ST_* bits
  int st;
};

// Tensors that a flag of ConvArgs::st / WgradArgs::st marks are arrays of bfloat16 (same element indexing, half
// the bytes); their pointers keep the float type of the struct and are only ever used through these helpers.
// (DVAE_ST_X: x / P, DVAE_ST_Y: y / Q, DVAE_ST_BNY: bn_y / p_y, DVAE_ST_INY: in_y, DVAE_ST_DX: dx_out)
// The kernels take the flags as a template argument ST, one instantiation per combination a training step
// produces: with a run-time flag the 8-byte and the 16-byte load of one value sit on two sides of a branch and
// write the same registers, and the compiler then orders every such load behind all loads in flight.
__device__ __forceinline__ f32x4 unpack_bf16x4(uint2 w) {
  f32x4 v;
  v[0] = __uint_as_float(w.x << 16);
  v[1] = __uint_as_float(w.x & 0xffff0000u);
  v[2] = __uint_as_float(w.y << 16);
  v[3] = __uint_as_float(w.y & 0xffff0000u);
  return v;
}
// four consecutive elements at element offset o of a tensor stored as floats, or (b16) as bfloat16.  The load
// hands back what memory holds (four floats, or the two dwords of four bfloat16 in [0] and [1]) and widen4s makes
// the values of it where they are used: a conversion beside the load would wait for the load there, and the row
// kernels issue their loads a whole row of MFMAs ahead of the use.
__device__ __forceinline__ f32x4 ld4s(const float* base, int64_t o, bool b16) {
  if (b16) {
    const uint2 w = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + o);
    return (f32x4){__uint_as_float(w.x), __uint_as_float(w.y), 0.f, 0.f};
  }
  return ldg4(base + o);
}
__device__ __forceinline__ f32x4 widen4s(const f32x4& raw, bool b16) {
  return b16 ? unpack_bf16x4(make_uint2(__float_as_uint(raw[0]), __float_as_uint(raw[1]))) : raw;
}
__device__ __forceinline__ void st4s(float* base, int64_t o, const f32x4& v, bool b16) {
  if (b16) {
    uint2 w;
    w.x = pack_bf16(v[0], v[1]);
    w.y = pack_bf16(v[2], v[3]);
    *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(base) + o) = w;
  } else {
    *reinterpret_cast<f32x4*>(base + o) = v;
  }
}

// Order of the K tiles when a K tile never straddles two filter taps (Cin % 32 == 0): blocks of 32 input channels
// outermost, the filter taps inside a block.  A frame's pixel is read by four taps (by four output pixels); taps
// outermost, those four reads of one 128-byte piece lie a whole pass over the tile's input apart (64-256 KB per
// workgroup, more than its share of L2) and the layer's input crosses the fabric three to four times.  Inside a
// block the conv's sixteen taps go by parity class (kh & 1, kw & 1), the four taps of a class touching the same
// pixels, so the re-reads come within four K tiles.  Operand B follows through conv_walk_k0.
template <int MODE>
__device__ __forceinline__ int conv_walk_tap(int kt) {
  if (MODE == 1) return kt & 3;
  return (int)((0xFD75EC64B931A820ull >> (4 * (kt & 15))) & 15u);      // 0 2 8 10 | 1 3 9 11 | 4 6 12 14 | 5 7 13 15
}
template <int MODE>
__device__ __forceinline__ int conv_walk_block(int kt) { return MODE == 1 ? kt >> 2 : kt >> 4; }
template <int MODE>
__device__ __forceinline__ int conv_walk_k0(const ConvArgs& g, int kt) {
  return g.cin32 ? conv_walk_tap<MODE>(kt) * g.Cin + conv_walk_block<MODE>(kt) * 32 : kt * 32;
}

// A operand of the forward implicit GEMMs.  MODE 0: conv (16 taps, input pixel 2*o-1+k);
// MODE 1: transposed conv phase (4 taps, input pixel a+ph-th).  C1: single input channel.
//
// Per M tile each thread records, for each of its rows, the element offset of the row's anchor pixel
// and one bit per filter tap saying whether that tap falls inside the frame.  A K tile then costs a
// bit test and one address add per 16-byte load.  issue() only starts loads (masked-out chunks read
// x[0..3], which is always there, and are zeroed in finish()); the BatchNorm scale/shift of the K
// tile's channels ride along as two more loads in flight.  K tiles may be visited in any order.
// NOXF: the caller vouches that the input takes no BatchNorm affine and no activation on its way in
template <int V4, int MODE, bool C1, bool NOXF = false>
struct ConvALoader {
  const ConvArgs& g;
  int ph, pw;
  int64_t off0[V4];       // element offset of (n, h0, w0, channel 0)
  unsigned tapmask[V4];
  int h0[V4], w0[V4];     // single-channel path only
  unsigned mask;
  float4 sc, sh;
  __device__ __forceinline__ ConvALoader(const ConvArgs& g_, int ph_, int pw_) : g(g_), ph(ph_), pw(pw_) {}
  __device__ __forceinline__ void set_tile(int64_t m0) {
#pragma unroll
    for (int v = 0; v < V4; ++v) {
      int row, ch;
      kc_pos(v, row, ch);
      const int64_t m = m0 + row;
      const int b = (int)(m & ((1 << g.lw) - 1));
      const int a = (int)((m >> g.lw) & ((1 << g.lh) - 1));
      const int64_t n = m >> (g.lw + g.lh);
      unsigned rowbits = 0, colbits = 0, bits;
      int hh, ww;
      if (MODE == 0) {
        hh = 2 * a - 1;
        ww = 2 * b - 1;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          rowbits |= (unsigned)(hh + d >= 0 && hh + d < g.H) << (4 * d);
          colbits |= (unsigned)(ww + d >= 0 && ww + d < g.W) << d;
        }
        bits = rowbits * colbits;          // bit 4*dh + dw
      } else {
        hh = a + ph;
        ww = b + pw;
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          rowbits |= (unsigned)(hh - d >= 0 && hh - d < g.H) << (2 * d);
          colbits |= (unsigned)(ww - d >= 0 && ww - d < g.W) << d;
        }
        bits = rowbits * colbits;          // bit 2*th + tw
      }
      tapmask[v] = m < g.M ? bits : 0u;
      off0[v] = ((n * g.H + hh) * g.W + ww) * g.Cin;
      h0[v] = hh;
      w0[v] = ww;
    }
  }
  __device__ __forceinline__ void issue(int kt, f32x4 (&r)[V4]) {
    const int ch = threadIdx.x & 7;
    mask = 0;
    if (C1) {
      // K = 16 (conv only): k = kh*4 + kw, the chunk is one filter row
      const int kh = ch;
      const bool kok = kt == 0 && kh < 4;
#pragma unroll
      for (int v = 0; v < V4; ++v) {
        float t[4];
        const unsigned bits = kok ? (tapmask[v] >> (4 * (kh & 3))) & 15u : 0u;
        const float* p = g.x + off0[v] + kh * g.W;
#pragma unroll
        for (int kw = 0; kw < 4; ++kw) {
          const bool eok = (bits >> kw) & 1u;
          t[kw] = *(eok ? p + kw : g.x);
        }
        mask |= bits << (4 * v);
        r[v] = (f32x4){t[0], t[1], t[2], t[3]};
      }
      return;
    }
    int tp, ci;
    bool kok;
    if (g.cin32) {
      tp = conv_walk_tap<MODE>(kt);
      ci = conv_walk_block<MODE>(kt) * 32 + 4 * ch;
      kok = kt * 32 < g.K;
    } else {
      const int k = kt * 32 + 4 * ch;
      tp = g.lc >= 0 ? k >> g.lc : k / g.Cin;
      ci = k - tp * g.Cin;
      kok = k < g.K;
    }
    const int tapoff = MODE == 0 ? (tp >> 2) * g.W + (tp & 3) : -((tp >> 1) * g.W + (tp & 1));
    const int toff = tapoff * g.Cin + ci;
    if (!NOXF && g.in_scale) {
      const int cs = kok ? ci : 0;
      sc = *reinterpret_cast<const float4*>(g.in_scale + cs);
      sh = *reinterpret_cast<const float4*>(g.in_shift + cs);
    }
#pragma unroll
    for (int v = 0; v < V4; ++v) {
      const bool eok = kok && ((tapmask[v] >> tp) & 1u);
      const float* p = eok ? g.x + (off0[v] + toff) : g.x;
      r[v] = ldg4(p);
      mask |= (unsigned)eok << v;
    }
  }
  struct Tag { unsigned mask; float4 sc, sh; };
  __device__ __forceinline__ Tag tag() const { return Tag{mask, sc, sh}; }
  __device__ __forceinline__ void set_tag(const Tag& t) { mask = t.mask; sc = t.sc; sh = t.sh; }
  __device__ __forceinline__ void finish(f32x4 (&r)[V4]) const {
    if (C1) {
#pragma unroll
      for (int v = 0; v < V4; ++v) {
        r[v].x = (mask >> (4 * v + 0)) & 1u ? r[v].x : 0.f;
        r[v].y = (mask >> (4 * v + 1)) & 1u ? r[v].y : 0.f;
        r[v].z = (mask >> (4 * v + 2)) & 1u ? r[v].z : 0.f;
        r[v].w = (mask >> (4 * v + 3)) & 1u ? r[v].w : 0.f;
      }
      return;
    }
#pragma unroll
    for (int v = 0; v < V4; ++v) {
      f32x4 val = r[v];
      if (!NOXF && g.in_scale) {
        val.x = fmaf(val.x, sc.x, sh.x); val.y = fmaf(val.y, sc.y, sh.y);
        val.z = fmaf(val.z, sc.z, sh.z); val.w = fmaf(val.w, sc.w, sh.w);
      }
      if (!NOXF && g.in_act == DVAE_ACT_LRELU) {
        val.x = lrelu(val.x); val.y = lrelu(val.y); val.z = lrelu(val.z); val.w = lrelu(val.w);
      }
      r[v] = zero_unless(val, (mask >> v) & 1u);
    }
  }
};

// B operand (packed filter [Cout][K]) read in the K order of the A operand
template <int V4, int MODE, bool WHOLE = false>
struct ConvBLoader {
  KCLoader<V4, WHOLE> l;
  const ConvArgs& g;
  __device__ __forceinline__ ConvBLoader(const ConvArgs& g_, const float* base, int64_t n0)
      : l(base, (int64_t)g_.K, (int64_t)g_.Cout, (int64_t)g_.K, n0), g(g_) {}
  __device__ __forceinline__ void issue(int kt, f32x4 (&r)[V4]) { l.issue_at(conv_walk_k0<MODE>(g, kt), r); }
  __device__ __forceinline__ void finish(f32x4 (&r)[V4]) const { l.finish(r); }
  typedef typename KCLoader<V4, WHOLE>::Tag Tag;
  __device__ __forceinline__ Tag tag() const { return l.tag(); }
  __device__ __forceinline__ void set_tag(const Tag& t) { l.set_tag(t); }
};

// BF: PREC_F32, PREC_BF16 or PREC_SPLIT.  LEAN >= 1: every column tile lies inside the filter and K is whole K tiles,
// so the filter loader needs no address selects and no zero masks; LEAN == 2: the input also takes no transform
template <class T, int MODE, bool C1, int BF, bool BNS, int LEAN = 0>
__global__ void __launch_bounds__(256, 2) conv_fwd_kernel(ConvArgs g) {
  __shared__ __attribute__((aligned(16))) float smem[2 * Core<T, BF>::STAGE];
  // transposed conv: the four sub-pixel phases of one M tile share its input tile.  Workgroups go to
  // the eight XCDs in turn (id mod 8), so with a multiple of eight tile columns the four are numbered
  // eight apart (same XCD, same L2); otherwise they are adjacent and meet in the last-level cache.
  const int gx = MODE == 1 ? (int)(gridDim.x >> 2) : (int)gridDim.x;
  int phase = MODE == 1 ? (int)(blockIdx.x & 3) : 0;
  int bx = MODE == 1 ? (int)(blockIdx.x >> 2) : (int)blockIdx.x;
  if (MODE == 1 && (gx & 7) == 0) {
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
    phase = j & 3;
    bx = (j >> 2) * 8 + x;
  }
  // conv with several column tiles: the column tiles of one M tile read the same input; numbered eight apart they
  // run on one XCD next to each other instead of a whole grid row of workgroups apart
  int by = (int)blockIdx.y;
  if (MODE == 0 && gridDim.y > 1 && (gx & 7) == 0) {
    const unsigned id = blockIdx.x + (unsigned)gx * blockIdx.y, slot = id >> 3;
    by = (int)(slot % gridDim.y);
    bx = (int)(slot / gridDim.y) * 8 + (int)(id & 7);
  }
  const int ph = phase >> 1, pw = phase & 1;
  const int64_t n0 = (int64_t)by * T::BN;
  const int ktiles = (g.K + 31) / 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / T::WGN, wn = wave % T::WGN;
  const int l31 = lane & 31;

  ConvALoader<T::A_V4, MODE, C1, LEAN == 2> la(g, ph, pw);
  ConvBLoader<T::B_V4, MODE, (LEAN >= 1)> lb(g, g.wp + (int64_t)phase * g.Cout * g.K, n0);

  float s1[T::TN], s2[T::TN];
#pragma unroll
  for (int j = 0; j < T::TN; ++j) s1[j] = s2[j] = 0.f;
  const bool split = g.slab != nullptr;
  const int kt_begin = split ? (int)blockIdx.z * g.kt_per : 0;
  const int kt_end = split ? (kt_begin + g.kt_per < ktiles ? kt_begin + g.kt_per : ktiles) : ktiles;
  float* const yout = split ? g.slab + (int64_t)blockIdx.z * g.slab_stride : g.y;

  for (int mt = g.mt0 + bx; mt < g.mtiles; mt += gx) {
    const int64_t m0 = (int64_t)mt * T::BM;
    la.set_tile(m0);
    f32x16 acc[T::TM][T::TN];
    zero_acc<T>(acc);
    gemm_mainloop<T, BF>(la, lb, kt_begin, kt_end, smem, acc);

    // store raw output
    for_each_acc<T>(acc, [&](int r, int c, float v) {
      const int64_t m = m0 + r, n = n0 + c;
      if (m < g.M && n < g.Cout) {
        int64_t orow;
        if (MODE == 0) {
          orow = m;
        } else {
          const int b = (int)(m & ((1 << g.lw) - 1));
          const int a = (int)((m >> g.lw) & ((1 << g.lh) - 1));
          const int64_t img = m >> (g.lw + g.lh);
          orow = (img * (2 * g.H) + (2 * a + ph)) * (2 * g.W) + (2 * b + pw);
        }
#if DVAE_CONV_EPILOGUE_PROBE
        if (v == 123.456f) yout[orow * g.Cout + n] = v;
#else
        yout[orow * g.Cout + n] = v;
#endif
      }
    });
    if (g.stats && !BNS && !split) {
#pragma unroll
      for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int j = 0; j < T::TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = acc[i][j][r];
            s1[j] += v;
            s2[j] += v * v;
          }
    } else if (BNS) {
      const int h = lane >> 5;
#pragma unroll
      for (int j = 0; j < T::TN; ++j) {
        const int64_t n = n0 + (wn * T::TN + j) * 32 + l31;
        const bool nok = n < g.Cout;
        const float bsc = nok ? g.bn_scale[n] : 0.f, bsh = nok ? g.bn_shift[n] : 0.f;
        const float bmu = nok ? g.bn_mean[n] : 0.f, bis = nok ? g.bn_invstd[n] : 0.f;
#pragma unroll
        for (int i = 0; i < T::TM; ++i) {
          float yv[16];
          bool ok[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int64_t m = m0 + (wm * T::TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            int64_t orow = m;
            if (MODE == 1) {
              const int b = (int)(m & ((1 << g.lw) - 1));
              const int a = (int)((m >> g.lw) & ((1 << g.lh) - 1));
              const int64_t img = m >> (g.lw + g.lh);
              orow = (img * (2 * g.H) + (2 * a + ph)) * (2 * g.W) + (2 * b + pw);
            }
            ok[r] = nok && m < g.M;
#if DVAE_CONV_EPILOGUE_PROBE
            yv[r] = ok[r] ? 0.5f : 0.f;
#else
            yv[r] = ok[r] ? g.bn_y[orow * g.Cout + n] : 0.f;
#endif
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float z = fmaf(yv[r], bsc, bsh);
            const float v = acc[i][j][r];
            const float dz = ok[r] ? (z > 0.f ? v : v * LRELU_SLOPE) : 0.f;
            s1[j] += dz;
            s2[j] += dz * ((yv[r] - bmu) * bis);
          }
        }
      }
    }
  }

  if (g.stats && !split) {
    // lanes l and l+32 hold the same column; waves with equal wn hold the same columns
    float* red = smem;  // [4 waves][TN*32][2]
    __syncthreads();
#pragma unroll
    for (int j = 0; j < T::TN; ++j) {
      const float a = s1[j] + __shfl_xor(s1[j], 32);
      const float b = s2[j] + __shfl_xor(s2[j], 32);
      if (lane < 32) {
        red[((wave * T::TN + j) * 32 + l31) * 2 + 0] = a;
        red[((wave * T::TN + j) * 32 + l31) * 2 + 1] = b;
      }
    }
    __syncthreads();
    if (threadIdx.x < T::BN) {
      const int col = threadIdx.x;
      const int cwn = col / (T::TN * 32), j = (col / 32) % T::TN, l = col & 31;
      float a = 0.f, b = 0.f;
      for (int m = 0; m < T::WGM; ++m) {
        const int w = m * T::WGN + cwn;
        a += red[((w * T::TN + j) * 32 + l) * 2 + 0];
        b += red[((w * T::TN + j) * 32 + l) * 2 + 1];
      }
      const int64_t n = n0 + col;
      if (n < g.Cout) {
        const int64_t part = g.part0 + (int64_t)phase * gx + bx;
        g.stats[(part * 2 + 0) * g.Cout + n] = a;
        g.stats[(part * 2 + 1) * g.Cout + n] = b;
      }
    }
  }
}

// Single-channel input (first encoder layer; data gradient of the final decoder layer): K is only the
// sixteen filter taps, so the layer is bound by the write of its output.  No LDS staging: every wave
// takes sixteen consecutive output pixels at a time, gathers their 4x4 input windows straight from the
// frame (lane = pixel x filter column, one load per filter row) as the B operand of
// v_mfma_f32_16x16x4_f32, keeps the filter [Cout x 16] in registers as the A operand, and gets four
// consecutive output channels of one pixel per lane, i.e. 16-B stores that tile each pixel's channel
// vector.  BatchNorm partial sums are kept per lane and reduced once per workgroup.
template <int NCB, bool BNS, int ST = 0>
__global__ void __launch_bounds__(256) conv_c1_kernel(ConvArgs g) {
  __shared__ float red[4][NCB * 16][2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i16 = lane & 15, kq = lane >> 4;
  float wa[NCB][4];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = cb * 16 + i16;
      wa[cb][r] = co < g.Cout ? g.wp[co * 16 + r * 4 + kq] : 0.f;      // tap (kh = r, kw = kq)
    }
  float s1[NCB][4], s2[NCB][4];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int r = 0; r < 4; ++r) s1[cb][r] = s2[cb][r] = 0.f;
  // BatchNorm-backward statistics: per-channel constants of this lane's four channels per block
  float bsc[NCB][4], bsh[NCB][4], bmu[NCB][4], bis[NCB][4];
  if (BNS) {
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = cb * 16 + 4 * kq + r;
        const bool ok = co < g.Cout;
        bsc[cb][r] = ok ? g.bn_scale[co] : 0.f;
        bsh[cb][r] = ok ? g.bn_shift[co] : 0.f;
        bmu[cb][r] = ok ? g.bn_mean[co] : 0.f;
        bis[cb][r] = ok ? g.bn_invstd[co] : 0.f;
      }
  }
  const int64_t ngroups = (g.M + 15) / 16;
  const bool vec_store = (g.Cout & 3) == 0;
  constexpr bool ybf = (ST & DVAE_ST_Y) != 0, bnybf = (ST & DVAE_ST_BNY) != 0;       // (the host sets them for vec_store shapes only)
  // two pixel groups per pass: the gathers of both are issued before either is multiplied and stored
  constexpr int U = 2;
  for (int64_t grp0 = (int64_t)blockIdx.x * 4 + wave; grp0 < ngroups; grp0 += (int64_t)gridDim.x * 4 * U) {
    float xvu[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t m = (grp0 + (int64_t)u * gridDim.x * 4) * 16 + i16;
      const bool mok = m < g.M;
      const int b = (int)(m & ((1 << g.lw) - 1));
      const int a = (int)((m >> g.lw) & ((1 << g.lh) - 1));
      const int64_t n = mok ? m >> (g.lw + g.lh) : 0;
      const float* img = g.x + n * g.H * g.W;
      const int wi = 2 * b - 1 + kq;
      const bool wok = mok && wi >= 0 && wi < g.W;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int hi = 2 * a - 1 + r;
        const bool ok = wok && hi >= 0 && hi < g.H;
        const float v = *(ok ? img + hi * g.W + wi : g.x);
        xvu[u][r] = ok ? v : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
    const int64_t m = (grp0 + (int64_t)u * gridDim.x * 4) * 16 + i16;
    const bool mok = m < g.M;
    const float (&xv)[4] = xvu[u];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[cb][r], xv[r], acc, 0, 0, 0);
      // D: row = output channel cb*16 + 4*kq + r, column = pixel i16
      const int co0 = cb * 16 + 4 * kq;
      if (mok && co0 < g.Cout) {
        float* dst = g.y + m * g.Cout + co0;
        if (vec_store) {
          st4s(g.y, m * g.Cout + co0, acc, ybf);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (co0 + r < g.Cout) dst[r] = acc[r];
        }
      }
      if (!BNS) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s1[cb][r] += acc[r];
          s2[cb][r] += acc[r] * acc[r];
        }
      } else if (mok && co0 < g.Cout) {
        // BatchNorm-backward statistics of the layer whose activation this gradient belongs to
        float yv[4] = {0.f, 0.f, 0.f, 0.f};
        const float* yp = g.bn_y + m * g.Cout + co0;
        if (vec_store) {
          const f32x4 t = widen4s(ld4s(g.bn_y, m * g.Cout + co0, bnybf), bnybf);
          yv[0] = t[0]; yv[1] = t[1]; yv[2] = t[2]; yv[3] = t[3];
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (co0 + r < g.Cout) yv[r] = yp[r];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float z = fmaf(yv[r], bsc[cb][r], bsh[cb][r]);
          const float dz = co0 + r < g.Cout ? (z > 0.f ? acc[r] : acc[r] * LRELU_SLOPE) : 0.f;
          s1[cb][r] += dz;
          s2[cb][r] += dz * ((yv[r] - bmu[cb][r]) * bis[cb][r]);
        }
      }
    }
    }
  }
  if (g.stats) {
    // the sixteen lanes with equal kq hold different pixels of the same channels
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float a1 = s1[cb][r], a2 = s2[cb][r];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          a1 += __shfl_xor(a1, o);
          a2 += __shfl_xor(a2, o);
        }
        if (i16 == 0) {
          red[wave][cb * 16 + 4 * kq + r][0] = a1;
          red[wave][cb * 16 + 4 * kq + r][1] = a2;
        }
      }
    __syncthreads();
    if (threadIdx.x < NCB * 16 && threadIdx.x < g.Cout) {
      const int co = threadIdx.x;
      const float a1 = (red[0][co][0] + red[1][co][0]) + (red[2][co][0] + red[3][co][0]);
      const float a2 = (red[0][co][1] + red[1][co][1]) + (red[2][co][1] + red[3][co][1]);
      g.stats[((int64_t)blockIdx.x * 2 + 0) * g.Cout + co] = a1;
      g.stats[((int64_t)blockIdx.x * 2 + 1) * g.Cout + co] = a2;
    }
  }
}

// Element type and multiply step of the row-ring kernels.  A lane's MFMA operand piece is four consecutive
// channels (or pixels) of one row: with f32 operands those are the operands of four v_mfma_f32_16x16x4_f32
// (k = kq of each), with bf16 operands they are ONE 8-byte operand of v_mfma_f32_16x16x16_bf16 (k = 4kq..4kq+3),
// so the bf16 variants keep their rings and their register-resident filters as bf16 (rounded to nearest even
// once, on the way in) and issue a quarter of the MFMAs at twice the rate each.
typedef short s16x4 __attribute__((ext_vector_type(4)));
template <bool BF>
struct RowOps;
template <>
struct RowOps<false> {
  typedef float elem;
  typedef f32x4 frag;
  static constexpr int PAD = 4;       // elements: 16 bytes
  static __device__ __forceinline__ frag make(const f32x4& v) { return v; }
};
template <>
struct RowOps<true> {
  typedef unsigned short elem;
  typedef s16x4 frag;
  static constexpr int PAD = 8;       // elements: 16 bytes
  static __device__ __forceinline__ frag make(const f32x4& v) {
    uint2 w;
    w.x = pack_bf16(v[0], v[1]);
    w.y = pack_bf16(v[2], v[3]);
    return __builtin_bit_cast(s16x4, w);
  }
  static __device__ __forceinline__ f32x4 mma(const frag& a, const frag& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
  }
};
// the weight-gradient kernels' K is pixels: a lane's four pixels 4kq..4kq+3 of one channel, read one by one
__device__ __forceinline__ s16x4 bf_quad(float k0, float k1, float k2, float k3) {
  uint2 w;
  w.x = pack_bf16(k0, k1);
  w.y = pack_bf16(k2, k3);
  return __builtin_bit_cast(s16x4, w);
}

// The weight-gradient row kernels in the 'fp32_split' mode: both operands of a product come out of LDS as fp32
// quads (a lane's four pixels of one channel), so a lane splits each quad into its three bf16 terms there and
// issues the six largest of the nine cross products on the bf16 MFMA, smallest first (gemm_core.h: split_pair).
// Products with one accumulator are six dependent MFMAs; callers run a term over all their accumulators before
// the next term.
struct SplitQuad {
  s16x4 t[3];      // hi, mid, lo
};
__device__ __forceinline__ SplitQuad split_quad(float k0, float k1, float k2, float k3) {
  uint32_t h0, m0, l0, h1, m1, l1;
  split_pair(k0, k1, h0, m0, l0);
  split_pair(k2, k3, h1, m1, l1);
  SplitQuad q;
  q.t[0] = __builtin_bit_cast(s16x4, make_uint2(h0, h1));
  q.t[1] = __builtin_bit_cast(s16x4, make_uint2(m0, m1));
  q.t[2] = __builtin_bit_cast(s16x4, make_uint2(l0, l1));
  return q;
}
// term pairs (a, b) of the six products, smallest first
__device__ constexpr int SPLIT_TA[6] = {2, 0, 1, 1, 0, 0};
__device__ constexpr int SPLIT_TB[6] = {0, 2, 1, 0, 1, 0};

// Transposed convolution with few channels (Cin*Cout <= 2048) on frames at least sixteen pixels wide:
// too little work per pixel for a tiled GEMM whose K loop is four taps deep.  A workgroup is four waves
// = the four sub-pixel phases; each wave keeps its phase's whole filter [Cout][4 taps][Cin] in
// registers as the A operands of v_mfma_f32_16x16x4_f32 and owns, with its workgroup, a contiguous run
// of input rows.  Every input row is fetched from memory once, with full-line coalesced loads, gets
// the producing layer's BatchNorm + LeakyReLU once, and is written into a ring of four row slots in
// LDS whose end pixels are zero: the rows an output row pair needs are then read by all four waves as
// MFMA B operands (lane = pixel x channel quad, 16-byte LDS reads), column taps are address offsets,
// the frame's left/right edges need no masks, and taps above/below the frame are skipped per wave.
// A lane ends up with four consecutive output channels of one output pixel: 16-byte stores that tile
// the pixel's channel vector.  The row after next is in flight while the current one is multiplied;
// one barrier per row.  Statistics (forward sums, or the BatchNorm-backward sums of ConvArgs::bn_y)
// stay in registers and leave as one partial row per wave.
#define ROWS_WMAX 32
// BNIN: the input is a gradient that still has to go through its own layer's BatchNorm backward
// (x = du, in_y = that layer's raw output, dx = scale*dz + A*y + B formed on the way into LDS and
// written to dx_out for the rows the workgroup owns), as in conv_rows_kernel.
template <int CIN, int COUT, int WAVES_PER_EU, bool XF, bool BNS, bool BNIN, bool BF, int ST = 0>
__global__ void __launch_bounds__(256, WAVES_PER_EU) deconv_rows_kernel(ConvArgs g) {
  typedef RowOps<BF> R;
  typedef typename R::frag frag_t;
  constexpr int NU = CIN / 16, NCB = COUT / 16;
  constexpr int PS = CIN + R::PAD;                  // pixel stride in LDS: a lane's reads without bank conflicts
  constexpr int C4 = CIN / 4;                       // four-channel pieces per pixel
  constexpr int NLD = (ROWS_WMAX * C4 + 255) / 256; // 16-byte pieces of a row per thread, at most
  __shared__ __attribute__((aligned(16))) typename R::elem ring[4][(ROWS_WMAX + 2) * PS];
  const int lane = threadIdx.x & 63, phase = threadIdx.x >> 6, ph = phase >> 1, pw = phase & 1;
  const int i16 = lane & 15, kq = lane >> 4;
  const int W = g.W, H = g.H;
  static_assert(BF || ST == 0, "bf16 storage goes with bf16 MFMA operands");
  constexpr bool xbf = (ST & DVAE_ST_X) != 0, ybf = (ST & DVAE_ST_Y) != 0, bnybf = (ST & DVAE_ST_BNY) != 0,
                 inybf = (ST & DVAE_ST_INY) != 0, dxbf = (ST & DVAE_ST_DX) != 0;
  // component m of wa[cb][t][u] multiplies channel 16u + 4kq + m of tap t into output channel 16cb + i16
  frag_t wa[NCB][4][NU];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int u = 0; u < NU; ++u)
        wa[cb][t][u] = R::make(ldg4(g.wp + (((int64_t)phase * COUT + 16 * cb + i16) * 4 + t) * CIN + 16 * u + 4 * kq));
  // loader role: this thread's pieces of a row are pieces threadIdx.x + 256 j; their channel offset is
  // the same for every j (C4 divides 256)
  const int lc4 = threadIdx.x % C4;
  f32x4 sc = (f32x4){1.f, 1.f, 1.f, 1.f}, sh = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float slope = g.in_act == DVAE_ACT_LRELU ? LRELU_SLOPE : 1.f;
  if ((XF || BNIN) && g.in_scale) {
    sc = ldg4(g.in_scale + 4 * lc4);
    sh = ldg4(g.in_shift + 4 * lc4);
  }
  f32x4 ca = sh, cb2 = sh;
  if (BNIN) {
    ca = ldg4(g.in_a + 4 * lc4);
    cb2 = ldg4(g.in_b + 4 * lc4);
  }
  const int row_pieces = W * C4;
  f32x4 bsc[NCB], bsh[NCB], bmu[NCB], bis[NCB];
  if (BNS) {
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      bsc[cb] = ldg4(g.bn_scale + 16 * cb + 4 * kq);
      bsh[cb] = ldg4(g.bn_shift + 16 * cb + 4 * kq);
      bmu[cb] = ldg4(g.bn_mean + 16 * cb + 4 * kq);
      bis[cb] = ldg4(g.bn_invstd + 16 * cb + 4 * kq);
    }
  }
  f32x4 s1[NCB], s2[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) s1[cb] = s2[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // zero pixels left and right of every row slot
  for (int i = threadIdx.x; i < 4 * 2 * PS; i += 256) {
    const int slot = i / (2 * PS), r = i % (2 * PS);
    ring[slot][(r < PS ? 0 : (W + 1) * PS) + r % PS] = 0;
  }

  // Rows are numbered over the whole batch (row = image * H + a): consecutive rows are consecutive in
  // memory.  The grid has one workgroup per partial-statistics row the caller allocated; the first
  // g.mtiles of them share out the rows in contiguous runs, the rest write zero statistics.
  const int64_t rows = g.M >> g.lw;
  const int64_t per = (rows + g.mtiles - 1) / g.mtiles;
  const int64_t r0 = (int)blockIdx.x < g.mtiles ? blockIdx.x * per : rows;
  const int64_t r1 = r0 + per < rows ? r0 + per : rows;

  f32x4 raw[NLD], rawy[BNIN ? NLD : 1];
  auto fetch = [&](int64_t row) {            // issue the loads of a row (nothing if it does not exist)
    const bool rok = row >= 0 && row < rows;
    const int64_t off = (rok ? row : 0) * (int64_t)row_pieces * 4;
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int f = threadIdx.x + 256 * j;
      const int64_t o = off + 4 * (rok && f < row_pieces ? f : 0);
      raw[j] = ld4s(g.x, o, xbf);
      if (BNIN) rawy[j] = ld4s(g.in_y, o, inybf);
    }
  };
  auto deposit = [&](int64_t row) {          // transform and write the fetched row into its slot
    typename R::elem* dst = ring[row & 3];
    const bool owned = BNIN && row >= r0 && row < r1;
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int f = threadIdx.x + 256 * j;
      if (f < row_pieces) {
        f32x4 v = widen4s(raw[j], xbf);
        if (BNIN) {
          const f32x4 y = widen4s(rawy[j], inybf);
          const f32x4 z = y * sc + sh;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float dz = z[e] > 0.f ? v[e] : v[e] * LRELU_SLOPE;
            v[e] = fmaf(sc[e], dz, fmaf(ca[e], y[e], cb2[e]));
          }
          if (owned) st4s(g.dx_out, row * (int64_t)row_pieces * 4 + 4 * f, v, dxbf);
        } else if (XF) {
          const f32x4 z = v * sc + sh;       // slope <= 1: lrelu(z) = max(z, slope z)
          const f32x4 zs = z * slope;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(z[e], zs[e]);
        }
        *reinterpret_cast<frag_t*>(dst + (1 + f / C4) * PS + 4 * lc4) = R::make(v);
      }
    }
  };
  if (r0 < r1) {
    fetch(r0 - 1);
    deposit(r0 - 1);
    fetch(r0);
    deposit(r0);
    fetch(r0 + 1);
    deposit(r0 + 1);
  }
  __syncthreads();
  for (int64_t r = r0; r < r1; ++r) {
    fetch(r + 2);
    const int a = (int)r & (H - 1);
    for (int j = 0; j < W / 16; ++j) {
      const int64_t orow = (2 * r + ph) * (2 * W) + 2 * (16 * j + i16) + pw;
      f32x4 yv[NCB];
      if (BNS) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) yv[cb] = ld4s(g.bn_y, orow * COUT + 16 * cb + 4 * kq, bnybf);
      }
      f32x4 acc[NCB][2];
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) acc[cb][0] = acc[cb][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int th = 0; th < 2; ++th) {
        const int arow = a + ph - th;
        if (arow < 0 || arow >= H) continue;          // the tap row is above or below the frame
        const typename R::elem* rowp = ring[(r + ph - th) & 3];
        frag_t xv[2][NU];
#pragma unroll
        for (int tw = 0; tw < 2; ++tw)
#pragma unroll
          for (int u = 0; u < NU; ++u)
            xv[tw][u] = *reinterpret_cast<const frag_t*>(rowp + (1 + 16 * j + i16 + pw - tw) * PS + 16 * u + 4 * kq);
#pragma unroll
        for (int tw = 0; tw < 2; ++tw)
#pragma unroll
          for (int u = 0; u < NU; ++u) {
            if constexpr (BF) {
#pragma unroll
              for (int cb = 0; cb < NCB; ++cb)
                acc[cb][u & 1] = R::mma(wa[cb][2 * th + tw][u], xv[tw][u], acc[cb][u & 1]);
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb)
                  acc[cb][e & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[cb][2 * th + tw][u][e], xv[tw][u][e],
                                                                        acc[cb][e & 1], 0, 0, 0);
            }
          }
      }
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        // D: row = output channel 16cb + 4kq + r, column = pixel 16j + i16
        const f32x4 v = acc[cb][0] + acc[cb][1];
        st4s(g.y, orow * COUT + 16 * cb + 4 * kq, v, ybf);
        if (!BNS) {
          s1[cb] += v;
          s2[cb] += v * v;
        } else {
          const f32x4 yw = widen4s(yv[cb], bnybf);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float z = fmaf(yw[e], bsc[cb][e], bsh[cb][e]);
            const float dz = z > 0.f ? v[e] : v[e] * LRELU_SLOPE;
            s1[cb][e] += dz;
            s2[cb][e] += dz * ((yw[e] - bmu[cb][e]) * bis[cb][e]);
          }
        }
      }
    }
    deposit(r + 2);       // slot (r+2)&3 is not among the three this row reads
    __syncthreads();
  }
  if (g.stats) {
    // the sixteen lanes with equal kq hold different pixels of the same channels
    const int64_t part = (int64_t)phase * gridDim.x + blockIdx.x;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a1 = s1[cb][e], a2 = s2[cb][e];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          a1 += __shfl_xor(a1, o);
          a2 += __shfl_xor(a2, o);
        }
        if (i16 == 0) {
          g.stats[(part * 2 + 0) * COUT + 16 * cb + 4 * kq + e] = a1;
          g.stats[(part * 2 + 1) * COUT + 16 * cb + 4 * kq + e] = a2;
        }
      }
  }
}

// The same row-ring scheme one size up (Cin*Cout <= 8192, e.g. 128 -> 64 on 8-pixel-wide frames): a
// phase's filter no longer fits one wave, so a workgroup takes ONE sub-pixel phase (the four
// workgroups of a row run are placed on one XCD and share their input rows through its L2) and its four waves hold sixteen
// output channels each ([16][4 taps][Cin] = 128 registers per lane at Cin = 128).  A group of sixteen
// pixels is two rows of eight, so the row a lane reads depends on the lane: rows outside the frame are
// redirected to a ninth, all-zero slot instead of being skipped.
template <int CIN, int WAVES_PER_EU, bool XF, bool BNS, bool BF, int ST = 0>
__global__ void __launch_bounds__(256, WAVES_PER_EU) deconv_rows8_kernel(ConvArgs g) {
  static_assert(BF || ST == 0, "bf16 storage goes with bf16 MFMA operands");
  static_assert((ST & ~DVAE_ST_Y) == 0, "this kernel only writes bf16");
  typedef RowOps<BF> R;
  typedef typename R::frag frag_t;
  constexpr int COUT = 64, W = 8, NU = CIN / 16;
  constexpr int PS = CIN + R::PAD, C4 = CIN / 4;
  constexpr int SLOT = (W + 2) * PS;
  constexpr int NLD = (W * C4 + 255) / 256;
  __shared__ __attribute__((aligned(16))) typename R::elem ring[9 * SLOT];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // The four phases of a row run read the same input rows.  Workgroups go to the eight XCDs in turn
  // (id mod 8), so with a multiple of eight runs the four are numbered eight apart: same XCD, same L2.
  const int gx = gridDim.x >> 2;
  int phase = blockIdx.x & 3, bx = blockIdx.x >> 2;
  if ((gx & 7) == 0) {
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
    phase = j & 3;
    bx = (j >> 2) * 8 + x;
  }
  const int ph = phase >> 1, pw = phase & 1;
  const int i16 = lane & 15, kq = lane >> 4;
  const int H = g.H;
  // component m of wa[t][u] multiplies channel 16u + 4kq + m of tap t into output channel 16 wave + i16
  frag_t wa[4][NU];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int u = 0; u < NU; ++u)
      wa[t][u] = R::make(ldg4(g.wp + (((int64_t)phase * COUT + 16 * wave + i16) * 4 + t) * CIN + 16 * u + 4 * kq));
  const int lc4 = threadIdx.x % C4;
  f32x4 sc = (f32x4){1.f, 1.f, 1.f, 1.f}, sh = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float slope = g.in_act == DVAE_ACT_LRELU ? LRELU_SLOPE : 1.f;
  if (XF && g.in_scale) {
    sc = ldg4(g.in_scale + 4 * lc4);
    sh = ldg4(g.in_shift + 4 * lc4);
  }
  f32x4 bsc, bsh, bmu, bis;
  if (BNS) {
    bsc = ldg4(g.bn_scale + 16 * wave + 4 * kq);
    bsh = ldg4(g.bn_shift + 16 * wave + 4 * kq);
    bmu = ldg4(g.bn_mean + 16 * wave + 4 * kq);
    bis = ldg4(g.bn_invstd + 16 * wave + 4 * kq);
  }
  f32x4 s1 = (f32x4){0.f, 0.f, 0.f, 0.f}, s2 = (f32x4){0.f, 0.f, 0.f, 0.f};

  // zero pixels left and right of the eight row slots, and the whole ninth slot
  for (int i = threadIdx.x; i < 8 * 2 * PS; i += 256) {
    const int slot = i / (2 * PS), r = i % (2 * PS);
    ring[slot * SLOT + (r < PS ? 0 : (W + 1) * PS) + r % PS] = 0;
  }
  for (int i = threadIdx.x; i < SLOT; i += 256) ring[8 * SLOT + i] = 0;

  // Input rows are numbered over the whole batch; a step takes the row pair (2p, 2p+1), which never
  // straddles two frames (H is even), and reads rows 2p-1 .. 2p+2.  gx workgroups per phase, one per
  // partial-statistics row; the first g.mtiles of them share out the pairs in contiguous runs.
  const int64_t rows = g.M >> 3, pairs = rows >> 1;
  const int64_t per = (pairs + g.mtiles - 1) / g.mtiles;
  const int64_t p0 = bx < g.mtiles ? bx * per : pairs;
  const int64_t p1 = p0 + per < pairs ? p0 + per : pairs;
  constexpr int row_pieces = W * C4;

  auto fetch = [&](int64_t row, f32x4 (&raw)[NLD]) {
    const bool rok = row >= 0 && row < rows;
    const float* src = g.x + (rok ? row : 0) * (int64_t)row_pieces * 4;
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int f = threadIdx.x + 256 * j;
      raw[j] = ldg4(src + 4 * (rok && f < row_pieces ? f : 0));
    }
  };
  auto deposit = [&](int64_t row, const f32x4 (&raw)[NLD]) {
    typename R::elem* dst = ring + (row & 7) * SLOT;
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int f = threadIdx.x + 256 * j;
      if (f < row_pieces) {
        f32x4 v = raw[j];
        if (XF) {
          const f32x4 z = v * sc + sh;       // slope <= 1: lrelu(z) = max(z, slope z)
          const f32x4 zs = z * slope;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(z[e], zs[e]);
        }
        *reinterpret_cast<frag_t*>(dst + (1 + f / C4) * PS + 4 * lc4) = R::make(v);
      }
    }
  };
  f32x4 rawa[NLD], rawb[NLD];
  if (p0 < p1) {
    fetch(2 * p0 - 1, rawa);
    fetch(2 * p0, rawb);
    deposit(2 * p0 - 1, rawa);
    deposit(2 * p0, rawb);
    fetch(2 * p0 + 1, rawa);
    fetch(2 * p0 + 2, rawb);
    deposit(2 * p0 + 1, rawa);
    deposit(2 * p0 + 2, rawb);
  }
  __syncthreads();
  const int b = i16 & 7;
  for (int64_t p = p0; p < p1; ++p) {
    fetch(2 * p + 3, rawa);
    fetch(2 * p + 4, rawb);
    const int64_t arow = 2 * p + (i16 >> 3);                // this lane's input row
    const int a = (int)arow & (H - 1);
    const int64_t orow = (2 * arow + ph) * (2 * W) + 2 * b + pw;
    f32x4 yv;
    if (BNS) yv = ldg4(g.bn_y + orow * COUT + 16 * wave + 4 * kq);
    f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int th = 0; th < 2; ++th) {
      const int ia = a + ph - th;
      const bool inside = ia >= 0 && ia < H;
      const typename R::elem* rowp = ring + (inside ? (int)((arow + ph - th) & 7) : 8) * SLOT;
      frag_t xv[2][NU];
#pragma unroll
      for (int tw = 0; tw < 2; ++tw)
#pragma unroll
        for (int u = 0; u < NU; ++u)
          xv[tw][u] = *reinterpret_cast<const frag_t*>(rowp + (1 + b + pw - tw) * PS + 16 * u + 4 * kq);
#pragma unroll
      for (int tw = 0; tw < 2; ++tw)
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          if constexpr (BF) {
            acc[u & 1] = R::mma(wa[2 * th + tw][u], xv[tw][u], acc[u & 1]);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              acc[e & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[2 * th + tw][u][e], xv[tw][u][e], acc[e & 1], 0, 0, 0);
          }
        }
    }
    // D: row = output channel 16 wave + 4kq + e, column = pixel i16
    const f32x4 v = acc[0] + acc[1];
    st4s(g.y, orow * COUT + 16 * wave + 4 * kq, v, (ST & DVAE_ST_Y) != 0);
    if (!BNS) {
      s1 += v;
      s2 += v * v;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float z = fmaf(yv[e], bsc[e], bsh[e]);
        const float dz = z > 0.f ? v[e] : v[e] * LRELU_SLOPE;
        s1[e] += dz;
        s2[e] += dz * ((yv[e] - bmu[e]) * bis[e]);
      }
    }
    deposit(2 * p + 3, rawa);       // slots of rows 2p+3, 2p+4 are not among the four this step reads
    deposit(2 * p + 4, rawb);
    __syncthreads();
  }
  if (g.stats) {
    const int64_t part = (int64_t)phase * gx + bx;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float a1 = s1[e], a2 = s2[e];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        a1 += __shfl_xor(a1, o);
        a2 += __shfl_xor(a2, o);
      }
      if (i16 == 0) {
        g.stats[(part * 2 + 0) * COUT + 16 * wave + 4 * kq + e] = a1;
        g.stats[(part * 2 + 1) * COUT + 16 * wave + 4 * kq + e] = a2;
      }
    }
  }
}

// Stride-2 convolution with few channels (Cin <= 32) on frames at least 32 pixels wide, by the same
// scheme: a wave keeps the filter of sixteen output channels [16][16 taps][Cin] in registers (the
// workgroup has Cout/16 waves), input rows go through an LDS ring once (eight slots: four rows feed an
// output row, two more are in flight), and the output row's sixteen-pixel groups take their B operands
// from LDS.  A row slot holds the even and the odd input columns apart, so that the sixteen lanes of a
// read (output pixels ob..ob+15, input column 2ob-1+kw) touch consecutive pixels: the odd columns are
// preceded by a zero pixel (column -1) and the even ones followed by one (column W).
//
// BNIN: the input is the gradient of a BatchNorm + LeakyReLU layer's activation that has not been
// taken through the BatchNorm backward yet.  x is du, in_y the layer's raw output, and a row becomes
// dx = scale * dz + A * y + B (dz = du * lrelu'(scale * y + shift); A, B fold the batch sums, see
// dvae_bn_bwd_coeffs_f32) on its way into LDS -- the elementwise BatchNorm-backward pass over the
// tensor disappears.  The rows a workgroup owns (not its halo rows) are also written to dx_out for
// the weight-gradient kernel.
template <int CIN, int COUT, int WAVES_PER_EU, bool XF, bool BNS, bool BNIN, bool BF, int ST = 0>
__global__ void __launch_bounds__(COUT * 4, WAVES_PER_EU) conv_rows_kernel(ConvArgs g) {
  typedef RowOps<BF> R;
  typedef typename R::frag frag_t;
  constexpr int NU = CIN / 16, NT = COUT * 4;
  constexpr int PS = CIN + R::PAD;                  // pixel stride in LDS: a lane's reads without bank conflicts
  constexpr int C4 = CIN / 4;                       // four-channel pieces per pixel
  constexpr int WMAX = BNIN ? 32 : 64;              // widest frame (the BNIN variant holds two tensors' rows)
  constexpr int NLD = (WMAX * C4 + NT - 1) / NT;    // 16-byte pieces of a row per thread, at most
  extern __shared__ __attribute__((aligned(16))) unsigned char ring_bytes[];
  typename R::elem* const ring = reinterpret_cast<typename R::elem*>(ring_bytes);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i16 = lane & 15, kq = lane >> 4;
  const int W = g.W, H = g.H, Wo = W >> 1, Ho = H >> 1;
  static_assert(BF || ST == 0, "bf16 storage goes with bf16 MFMA operands");
  constexpr bool xbf = (ST & DVAE_ST_X) != 0, ybf = (ST & DVAE_ST_Y) != 0, bnybf = (ST & DVAE_ST_BNY) != 0,
                 inybf = (ST & DVAE_ST_INY) != 0, dxbf = (ST & DVAE_ST_DX) != 0;
  const int half = Wo + 1;                          // pixels of one parity, with its zero pixel
  const int slot_floats = 2 * half * PS;
  // component m of wa[t][u] multiplies channel 16u + 4kq + m of tap t into output channel 16 wave + i16
  frag_t wa[16][NU];
#pragma unroll
  for (int t = 0; t < 16; ++t)
#pragma unroll
    for (int u = 0; u < NU; ++u)
      wa[t][u] = R::make(ldg4(g.wp + ((int64_t)(16 * wave + i16) * 16 + t) * CIN + 16 * u + 4 * kq));
  const int lc4 = threadIdx.x % C4;
  f32x4 sc = (f32x4){1.f, 1.f, 1.f, 1.f}, sh = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float slope = g.in_act == DVAE_ACT_LRELU ? LRELU_SLOPE : 1.f;
  if ((XF || BNIN) && g.in_scale) {
    sc = ldg4(g.in_scale + 4 * lc4);
    sh = ldg4(g.in_shift + 4 * lc4);
  }
  f32x4 ca = sh, cb = sh;
  if (BNIN) {
    ca = ldg4(g.in_a + 4 * lc4);
    cb = ldg4(g.in_b + 4 * lc4);
  }
  const int row_pieces = W * C4;
  f32x4 bsc, bsh, bmu, bis;
  if (BNS) {
    bsc = ldg4(g.bn_scale + 16 * wave + 4 * kq);
    bsh = ldg4(g.bn_shift + 16 * wave + 4 * kq);
    bmu = ldg4(g.bn_mean + 16 * wave + 4 * kq);
    bis = ldg4(g.bn_invstd + 16 * wave + 4 * kq);
  }
  f32x4 s1 = (f32x4){0.f, 0.f, 0.f, 0.f}, s2 = (f32x4){0.f, 0.f, 0.f, 0.f};

  // the zero pixels of every slot: even columns' at index Wo, odd columns' at index 0 of their half
  for (int i = threadIdx.x; i < 8 * 2 * PS; i += NT) {
    const int slot = i / (2 * PS), r = i % (2 * PS);
    ring[slot * slot_floats + (r < PS ? Wo * PS : half * PS) + r % PS] = 0;
  }

  // Output rows are numbered over the whole batch (row = image * Ho + oa); output row r reads input
  // rows 2r-1 .. 2r+2 of the same numbering.  The grid has one workgroup per partial-statistics row
  // the caller allocated; the first g.mtiles share out the rows in contiguous runs.
  const int64_t rows = g.M >> g.lw, rows_in = 2 * rows;
  const int64_t per = (rows + g.mtiles - 1) / g.mtiles;
  const int64_t r0 = (int)blockIdx.x < g.mtiles ? blockIdx.x * per : rows;
  const int64_t r1 = r0 + per < rows ? r0 + per : rows;

  struct Raw {
    f32x4 v[NLD], y[BNIN ? NLD : 1];
  };
  auto fetch = [&](int64_t row, Raw& raw) {      // issue the loads of an input row (if it exists)
    const bool rok = row >= 0 && row < rows_in;
    const int64_t off = (rok ? row : 0) * (int64_t)row_pieces * 4;
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int f = threadIdx.x + NT * j;
      const int64_t o = off + 4 * (rok && f < row_pieces ? f : 0);
      raw.v[j] = ld4s(g.x, o, xbf);
      if (BNIN) raw.y[j] = ld4s(g.in_y, o, inybf);
    }
  };
  auto deposit = [&](int64_t row, const Raw& raw) {   // transform, split by column parity, store
    typename R::elem* dst = ring + (row & 7) * slot_floats;
    const bool owned = BNIN && row >= 2 * r0 && row < 2 * r1;
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int f = threadIdx.x + NT * j;
      if (f < row_pieces) {
        f32x4 v = widen4s(raw.v[j], xbf);
        if (BNIN) {
          const f32x4 y = widen4s(raw.y[j], inybf);
          const f32x4 z = y * sc + sh;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float dz = z[e] > 0.f ? v[e] : v[e] * LRELU_SLOPE;
            v[e] = fmaf(sc[e], dz, fmaf(ca[e], y[e], cb[e]));
          }
          if (owned) st4s(g.dx_out, row * (int64_t)row_pieces * 4 + 4 * f, v, dxbf);
        } else if (XF) {
          const f32x4 z = v * sc + sh;       // slope <= 1: lrelu(z) = max(z, slope z)
          const f32x4 zs = z * slope;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(z[e], zs[e]);
        }
        const int px = f / C4;
        const int at = (px & 1) ? half + 1 + (px >> 1) : (px >> 1);
        *reinterpret_cast<frag_t*>(dst + at * PS + 4 * lc4) = R::make(v);
      }
    }
  };
  Raw rawa, rawb;
  if (r0 < r1) {
    fetch(2 * r0 - 1, rawa);
    fetch(2 * r0, rawb);
    deposit(2 * r0 - 1, rawa);
    deposit(2 * r0, rawb);
    fetch(2 * r0 + 1, rawa);
    fetch(2 * r0 + 2, rawb);
    deposit(2 * r0 + 1, rawa);
    deposit(2 * r0 + 2, rawb);
  }
  __syncthreads();
  for (int64_t r = r0; r < r1; ++r) {
    fetch(2 * r + 3, rawa);
    fetch(2 * r + 4, rawb);
    const int oa = (int)r & (Ho - 1);
    for (int j = 0; j < Wo / 16; ++j) {
      const int ob = 16 * j + i16;
      const int64_t orow = r * Wo + ob;
      f32x4 yv = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (BNS) yv = ld4s(g.bn_y, orow * COUT + 16 * wave + 4 * kq, bnybf);
      f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int kh = 0; kh < 4; ++kh) {
        const int hin = 2 * oa - 1 + kh;
        if (hin < 0 || hin >= H) continue;          // the tap row is above or below the frame
        const typename R::elem* rowp = ring + ((2 * r - 1 + kh) & 7) * slot_floats;
        // input column 2ob-1+kw: kw 0 -> odd[ob-1], 1 -> even[ob], 2 -> odd[ob], 3 -> even[ob+1]
        const int at[4] = {half + ob, ob, half + ob + 1, ob + 1};
        frag_t xv[4][NU];
#pragma unroll
        for (int kw = 0; kw < 4; ++kw)
#pragma unroll
          for (int u = 0; u < NU; ++u)
            xv[kw][u] = *reinterpret_cast<const frag_t*>(rowp + at[kw] * PS + 16 * u + 4 * kq);
#pragma unroll
        for (int kw = 0; kw < 4; ++kw)
#pragma unroll
          for (int u = 0; u < NU; ++u) {
            if constexpr (BF) {
              acc[(kw * NU + u) & 1] = R::mma(wa[4 * kh + kw][u], xv[kw][u], acc[(kw * NU + u) & 1]);
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                acc[e & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[4 * kh + kw][u][e], xv[kw][u][e], acc[e & 1], 0, 0, 0);
            }
          }
      }
      // D: row = output channel 16 wave + 4kq + e, column = output pixel ob
      const f32x4 v = acc[0] + acc[1];
      st4s(g.y, orow * COUT + 16 * wave + 4 * kq, v, ybf);
      if (!BNS) {
        s1 += v;
        s2 += v * v;
      } else {
        const f32x4 yw = widen4s(yv, bnybf);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float z = fmaf(yw[e], bsc[e], bsh[e]);
          const float dz = z > 0.f ? v[e] : v[e] * LRELU_SLOPE;
          s1[e] += dz;
          s2[e] += dz * ((yw[e] - bmu[e]) * bis[e]);
        }
      }
    }
    deposit(2 * r + 3, rawa);       // slots of rows 2r+3, 2r+4 are not among the four this row reads
    deposit(2 * r + 4, rawb);
    __syncthreads();
  }
  if (g.stats) {
    // the sixteen lanes with equal kq hold different pixels of the same channels
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float a1 = s1[e], a2 = s2[e];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        a1 += __shfl_xor(a1, o);
        a2 += __shfl_xor(a2, o);
      }
      if (i16 == 0) {
        g.stats[((int64_t)blockIdx.x * 2 + 0) * COUT + 16 * wave + 4 * kq + e] = a1;
        g.stats[((int64_t)blockIdx.x * 2 + 1) * COUT + 16 * wave + 4 * kq + e] = a2;
      }
    }
  }
}

extern "C" int64_t dvae_conv_stat_parts(int64_t M_rows) {
  (void)M_rows;
  return 4 * CONV_MAX_PARTS;
}

// number of stat partial rows a forward launch writes
static int64_t conv_parts_written(int64_t M, int phases) {
  // all tile shapes use BM = 128
  int64_t mt = ceil_div64(M, 128);
  if (mt > CONV_MAX_PARTS) mt = CONV_MAX_PARTS;
  return mt * phases;
}

// K split of the tiled kernels: how many ways (0: no split).  For layers whose 128-wide tiles fill between a quarter
// and half of the chip's 512 workgroup slots and whose K loop is long enough to share out.  (Fewer tiles than that
// are better served by narrower tiles, which already give every compute unit a workgroup: measured 68 vs 76 us and
// 36 vs 51 us with the split at 64 and 8 tiles; at 160 tiles the split takes 129 us against 170.)
static int conv_split_ways(int64_t M, int phases, int64_t Cout, int64_t K) {
  if (Cout < 128 || Cout % 4 != 0 || 256 % (Cout / 4) != 0) return 0;
  const int64_t tiles128 = ceil_div64(M, 128) * phases * ceil_div64(Cout, 128);
  const int64_t ktiles = ceil_div64(K, 32);
  if (tiles128 <= 128 || tiles128 > 256 || ktiles < 32) return 0;
  int64_t S = 512 / tiles128;
  if (S > 4) S = 4;
  if (S > ktiles / 16) S = ktiles / 16;     // a share shorter than 16 K tiles costs more in prologue, epilogue and
                                            // summing than the fuller chip returns (measured: 8 per share, +10 %)
  while (S >= 2 && (S - 1) * ceil_div64(ktiles, S) >= ktiles) --S;    // no empty share
  return S >= 2 ? (int)S : 0;
}

// Sum of the K-split partial outputs (fixed order), written to y; with stats != nullptr also the forward
// BatchNorm partial sums (sum, sum of squares per channel) of the rows a workgroup handles, one partial row per
// workgroup.  Thread = four channels of every (256 / (Cout/4))-th row of the workgroup's run of rows.
__global__ void __launch_bounds__(256)
conv_slab_reduce_kernel(const float4* __restrict__ slab, int S, int64_t n4, int C4, float4* __restrict__ y,
                        float* __restrict__ stats, int64_t rows, int64_t rows_per_part) {
  __shared__ float red[2][256][4];
  const int c4 = threadIdx.x % C4, rsub = threadIdx.x / C4, RG = 256 / C4;
  const int64_t r0 = blockIdx.x * rows_per_part;
  const int64_t r1 = r0 + rows_per_part < rows ? r0 + rows_per_part : rows;
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
  for (int64_t r = r0 + rsub; r < r1; r += RG) {
    const int64_t idx = r * C4 + c4;
    float4 v = slab[idx];
    for (int z = 1; z < S; ++z) {
      const float4 w = slab[(int64_t)z * n4 + idx];
      v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
    }
    y[idx] = v;
    s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
    s2.x += v.x * v.x; s2.y += v.y * v.y; s2.z += v.z * v.z; s2.w += v.w * v.w;
  }
  if (stats == nullptr) return;
  red[0][threadIdx.x][0] = s1.x; red[0][threadIdx.x][1] = s1.y; red[0][threadIdx.x][2] = s1.z; red[0][threadIdx.x][3] = s1.w;
  red[1][threadIdx.x][0] = s2.x; red[1][threadIdx.x][1] = s2.y; red[1][threadIdx.x][2] = s2.z; red[1][threadIdx.x][3] = s2.w;
  __syncthreads();
  const int Cout = 4 * C4;
  for (int i = threadIdx.x; i < 2 * Cout; i += 256) {
    const int which = i / Cout, c = i % Cout;
    float a = 0.f;
    for (int q = 0; q < RG; ++q) a += red[which][q * C4 + (c >> 2)][c & 3];
    stats[((int64_t)blockIdx.x * 2 + which) * Cout + c] = a;
  }
}

// The row kernels on bf16-stored tensors: one instantiation per combination that a training step produces
// (false: none for this kernel variant and these flags).
template <int CI, int CO, int WPE, bool XF, bool BNS, bool BNIN>
static bool launch_deconv_rows_st(const ConvArgs& g, dim3 grid, hipStream_t stream) {
  if constexpr (CI == 64 && CO == 32 && !BNS && !BNIN) {
    // the decoder's last block writes its raw output as bf16
    if (g.st == DVAE_ST_Y) {
      DVAE_LAUNCH((deconv_rows_kernel<CI, CO, WPE, XF, BNS, BNIN, true, DVAE_ST_Y>), grid, dim3(256), 0, stream, g);
      return true;
    }
    // ... and reads the block before it as bf16
    if (XF && g.st == (DVAE_ST_X | DVAE_ST_Y)) {
      DVAE_LAUNCH((deconv_rows_kernel<CI, CO, WPE, true, BNS, BNIN, true, DVAE_ST_X | DVAE_ST_Y>), grid, dim3(256), 0,
                  stream, g);
      return true;
    }
  }
  if constexpr (CI == 32 && CO == 16 && BNS && BNIN) {
    // data gradient of the motion encoder's second convolution: reads the first layer's raw output (bf16) for the
    // BatchNorm-backward sums and writes the gradient of its activation as bf16
    if (g.st == (DVAE_ST_Y | DVAE_ST_BNY)) {
      DVAE_LAUNCH((deconv_rows_kernel<CI, CO, WPE, XF, BNS, BNIN, true, DVAE_ST_Y | DVAE_ST_BNY>), grid, dim3(256), 0,
                  stream, g);
      return true;
    }
  }
  return false;
}
template <int CI, int CO, int WPE, bool XF, bool BNS, bool BNIN>
static bool launch_conv_rows_st(const ConvArgs& g, dim3 grid, size_t lds, hipStream_t stream) {
  if constexpr (CI == 32 && CO == 64 && BNS && BNIN) {
    // its data gradient: bf16 du and y in, bf16 dL/dy out
    if (g.st == (DVAE_ST_X | DVAE_ST_INY | DVAE_ST_DX)) {
      DVAE_LAUNCH((conv_rows_kernel<CI, CO, WPE, XF, BNS, BNIN, true, DVAE_ST_X | DVAE_ST_INY | DVAE_ST_DX>), grid,
                  dim3(CO * 4), lds, stream, g);
      return true;
    }
    // ... with the block below stored as bf16 too (its raw output read for the sums, the gradient of its activation written)
    constexpr int ALL = DVAE_ST_X | DVAE_ST_INY | DVAE_ST_DX | DVAE_ST_BNY | DVAE_ST_Y;
    if (g.st == ALL) {
      DVAE_LAUNCH((conv_rows_kernel<CI, CO, WPE, XF, BNS, BNIN, true, ALL>), grid, dim3(CO * 4), lds, stream, g);
      return true;
    }
  }
  if constexpr (CI == 16 && CO == 32 && XF && !BNS && !BNIN) {
    // the motion encoder's second convolution reads the first layer's raw output as bf16
    if (g.st == DVAE_ST_X) {
      DVAE_LAUNCH((conv_rows_kernel<CI, CO, WPE, XF, BNS, BNIN, true, DVAE_ST_X>), grid, dim3(CO * 4), lds, stream, g);
      return true;
    }
  }
  return false;
}

template <int MODE>
static int launch_conv_fwd(const char* name, ConvArgs g, int phases, hipStream_t stream) {
  const bool c1 = g.Cin == 1;
  const bool bf = dvae_mfma_precision() == DVAE_PRECISION_BF16;
  const bool sp = dvae_mfma_precision() == DVAE_PRECISION_F32_SPLIT;     // tiled kernels without fused BN-backward sums
  const bool bns = g.bn_y != nullptr;
#define LAUNCH(TILE, C1FLAG)                                                                         \
  {                                                                                                  \
    if (mt_end < 0) {                                                                                \
      g.mt0 = 0;                                                                                     \
      g.mtiles = (int)ceil_div64(g.M, TILE::BM);                                                     \
    } else {                                                                                         \
      g.mt0 = mt_begin;                                                                              \
      g.mtiles = mt_end;                                                                             \
    }                                                                                                \
    int gx = g.mtiles - g.mt0 < gx_cap ? g.mtiles - g.mt0 : gx_cap;                                  \
    dim3 grid(gx * phases, (unsigned)ceil_div64(g.Cout, TILE::BN), (unsigned)ksplit);                \
    {                                                                                                \
      int64_t rows_ = (int64_t)g.mtiles * TILE::BM;                                                  \
      rows_ = (rows_ < g.M ? rows_ : g.M) - (int64_t)g.mt0 * TILE::BM;                               \
      if (C1FLAG) DVAE_WORK(4.0 * ((double)rows_ * 4 + (double)rows_ * g.Cout), DVAE_UNIT_BYTE);     \
      else DVAE_WORK(2.0 * (double)rows_ * phases * g.Cout * g.K, DVAE_UNIT_FLOP);                   \
    }                                                                                                \
    const bool bw_ = DVAE_CONV_WHOLE_B && !(C1FLAG) && g.Cout % TILE::BN == 0 && g.K % 32 == 0;        \
    if (bns && !(C1FLAG)) {                                                                          \
      if (bf)                                                                                        \
        CONV_FWD_LAUNCH(TILE, C1FLAG, (C1FLAG) ? PREC_F32 : PREC_BF16, !(C1FLAG))                    \
      else                                                                                           \
        CONV_FWD_LAUNCH(TILE, C1FLAG, PREC_F32, !(C1FLAG))                                           \
    } else if (bf && !(C1FLAG))                                                                      \
      CONV_FWD_LAUNCH(TILE, C1FLAG, (C1FLAG) ? PREC_F32 : PREC_BF16, false)                          \
    else if (sp && !(C1FLAG))                                                                        \
      CONV_FWD_LAUNCH(TILE, C1FLAG, (C1FLAG) ? PREC_F32 : PREC_SPLIT, false)                         \
    else                                                                                             \
      CONV_FWD_LAUNCH(TILE, C1FLAG, PREC_F32, false)                                                 \
  }
#define CONV_FWD_LAUNCH(TILE, C1FLAG, PREC_, BNS_)                                                             \
  {                                                                                                            \
    if (bw_ && g.in_scale == nullptr && g.in_act == DVAE_ACT_NONE)                                              \
      DVAE_LAUNCH((conv_fwd_kernel<TILE, MODE, C1FLAG, PREC_, BNS_, (C1FLAG) ? 0 : 2>), grid, dim3(256), 0, stream, g); \
    else if (bw_) DVAE_LAUNCH((conv_fwd_kernel<TILE, MODE, C1FLAG, PREC_, BNS_, (C1FLAG) ? 0 : 1>), grid, dim3(256), 0, stream, g); \
    else DVAE_LAUNCH((conv_fwd_kernel<TILE, MODE, C1FLAG, PREC_, BNS_, 0>), grid, dim3(256), 0, stream, g);      \
  }
  int mt_begin = 0, mt_end = -1;       // M-tile range of the next LAUNCH (-1: the whole layer)
  int ksplit = 1;                      // gridDim.z of the next LAUNCH
  int gx_cap = CONV_MAX_PARTS;         // most workgroups along M (= statistics partial rows per phase) of the next LAUNCH
  typedef Tile<128, 128, 2, 2, true, true> T128;
  typedef Tile<128, 64, 4, 1, true, true> T64;
  typedef Tile<128, 32, 4, 1, true, true> T32;
#define ROWS_K(CI, CO, WPE, XF_, BNS_, BNIN_)                                                            \
  {                                                                                                      \
    if (bf && g.st != 0) {                                                                               \
      if (!launch_deconv_rows_st<CI, CO, WPE, XF_, BNS_, BNIN_>(g, grid, stream)) {                      \
        dvae_set_error("%s: no row kernel for this combination of bf16-stored tensors (0x%x)", name, g.st); \
        return -EINVAL;                                                                                  \
      }                                                                                                  \
    } else if (bf) DVAE_LAUNCH((deconv_rows_kernel<CI, CO, WPE, XF_, BNS_, BNIN_, true>), grid, dim3(256), 0, stream, g); \
    else DVAE_LAUNCH((deconv_rows_kernel<CI, CO, WPE, XF_, BNS_, BNIN_, false>), grid, dim3(256), 0, stream, g);   \
  }
#define ROWS_LAUNCH(CI, CO, WPE, RESIDENT)                                                               \
  if (MODE == 1 && g.Cin == CI && g.Cout == CO && g.W >= 16 && g.W <= ROWS_WMAX &&                 \
      aligned16(g.y) && (!bns || aligned16(g.bn_y)) &&                                                    \
      (!g.in_scale || (aligned16(g.in_scale) && aligned16(g.in_shift)))) {                                \
    const dim3 grid((unsigned)(conv_parts_written(g.M, phases) / phases));                                \
    g.mtiles = (int)grid.x < 256 * RESIDENT ? (int)grid.x : 256 * RESIDENT;                               \
    const bool xf = g.in_scale != nullptr || g.in_act != DVAE_ACT_NONE;                                   \
    DVAE_WORK(2.0 * (double)g.M * phases * g.Cout * g.K, DVAE_UNIT_FLOP);                                 \
    if (g.in_y != nullptr && bns)                                                                         \
      ROWS_K(CI, CO, WPE, false, true, true)                                                              \
    else if (g.in_y != nullptr) {                                                                         \
      dvae_set_error("%s: a BatchNorm-backward input needs the fused statistics of the layer below", name); \
      return -EINVAL;                                                                                     \
    } else if (bns && xf)                                                                                 \
      ROWS_K(CI, CO, WPE, true, true, false)                                                              \
    else if (bns)                                                                                         \
      ROWS_K(CI, CO, WPE, false, true, false)                                                             \
    else if (xf)                                                                                          \
      ROWS_K(CI, CO, WPE, true, false, false)                                                             \
    else                                                                                                  \
      ROWS_K(CI, CO, WPE, false, false, false)                                                            \
    DVAE_LAUNCH_CHECK(name);                                                                              \
    return 0;                                                                                             \
  }
  ROWS_LAUNCH(32, 16, 2, 4)
  ROWS_LAUNCH(64, 32, 2, 2)
#undef ROWS_LAUNCH
#undef ROWS_K
  if (MODE == 1 && g.in_y != nullptr) {
    dvae_set_error("%s: no kernel takes a BatchNorm-backward input at this shape (see dvae_deconv4s2_bnin_supported)", name);
    return -EINVAL;
  }
  if (MODE == 1 && g.Cin == 128 && g.Cout == 64 && g.W == 8 && g.H >= 2 && aligned16(g.y) &&
      (!bns || aligned16(g.bn_y)) && (!g.in_scale || (aligned16(g.in_scale) && aligned16(g.in_shift)))) {
    const unsigned gx = (unsigned)(conv_parts_written(g.M, phases) / phases);
    g.mtiles = (int)gx < 128 ? (int)gx : 128;      // 4 phases x 128 = two workgroups per compute unit
    const bool xf = g.in_scale != nullptr || g.in_act != DVAE_ACT_NONE;
    const dim3 grid(gx * 4);
    DVAE_WORK(2.0 * (double)g.M * phases * g.Cout * g.K, DVAE_UNIT_FLOP);
#define ROWS8_K(XF_, BNS_)                                                                              \
  {                                                                                                     \
    if (bf && g.st != 0) {                                                                              \
      if (XF_ && !(BNS_) && g.st == DVAE_ST_Y)                                                          \
        DVAE_LAUNCH((deconv_rows8_kernel<128, 2, true, false, true, DVAE_ST_Y>), grid, dim3(256), 0, stream, g); \
      else {                                                                                            \
        dvae_set_error("%s: the 128 -> 64 row kernel writes a bf16 output in its plain forward form only", name); \
        return -EINVAL;                                                                                 \
      }                                                                                                 \
    } else if (bf) DVAE_LAUNCH((deconv_rows8_kernel<128, 2, XF_, BNS_, true>), grid, dim3(256), 0, stream, g); \
    else DVAE_LAUNCH((deconv_rows8_kernel<128, 2, XF_, BNS_, false>), grid, dim3(256), 0, stream, g);   \
  }
    if (bns && xf) ROWS8_K(true, true)
    else if (bns) ROWS8_K(false, true)
    else if (xf) ROWS8_K(true, false)
    else ROWS8_K(false, false)
#undef ROWS8_K
    DVAE_LAUNCH_CHECK(name);
    return 0;
  }
  if (MODE == 1 && g.st != 0) {
    dvae_set_error("%s: bf16 storage is only taken by the row kernels (64 -> 32 and 32 -> 16 channels on frames 16 to "
                   "32 pixels wide, 128 -> 64 on 8-pixel-wide frames)", name);
    return -EINVAL;
  }
#define CROWS_K(CI, CO, WPE, XF_, BNS_, BNIN_)                                                           \
  {                                                                                                      \
    if (bf && g.st != 0) {                                                                               \
      if (!launch_conv_rows_st<CI, CO, WPE, XF_, BNS_, BNIN_>(g, grid, lds, stream)) {                   \
        dvae_set_error("%s: no row kernel for this combination of bf16-stored tensors (0x%x)", name, g.st); \
        return -EINVAL;                                                                                  \
      }                                                                                                  \
    } else if (bf) DVAE_LAUNCH((conv_rows_kernel<CI, CO, WPE, XF_, BNS_, BNIN_, true>), grid, dim3(CO * 4), lds, stream, g); \
    else DVAE_LAUNCH((conv_rows_kernel<CI, CO, WPE, XF_, BNS_, BNIN_, false>), grid, dim3(CO * 4), lds, stream, g);   \
  }
#define CONV_ROWS_LAUNCH(CI, CO, WPE, RESIDENT)                                                           \
  if (MODE == 0 && g.Cin == CI && g.Cout == CO && g.W >= 32 && g.W <= 64 && g.H >= 2 &&            \
      aligned16(g.y) && (!bns || aligned16(g.bn_y)) &&                                                    \
      (!g.in_scale || (aligned16(g.in_scale) && aligned16(g.in_shift)))) {                                \
    const dim3 grid((unsigned)conv_parts_written(g.M, 1));                                                \
    g.mtiles = (int)grid.x < 256 * RESIDENT ? (int)grid.x : 256 * RESIDENT;                               \
    const size_t lds = (size_t)8 * 2 * (g.W / 2 + 1) * (bf ? (CI + 8) * 2 : (CI + 4) * 4);                \
    const bool xf = g.in_scale != nullptr || g.in_act != DVAE_ACT_NONE;                                   \
    DVAE_WORK(2.0 * (double)g.M * g.Cout * g.K, DVAE_UNIT_FLOP);                                          \
    if (g.in_y != nullptr && bns)                                                                         \
      CROWS_K(CI, CO, WPE, false, true, true) \
    else if (g.in_y != nullptr) {                                                                         \
      dvae_set_error("%s: a BatchNorm-backward input needs the fused statistics of the layer below", name); \
      return -EINVAL;                                                                                     \
    } else if (bns && xf)                                                                                 \
      CROWS_K(CI, CO, WPE, true, true, false) \
    else if (bns)                                                                                         \
      CROWS_K(CI, CO, WPE, false, true, false) \
    else if (xf)                                                                                          \
      CROWS_K(CI, CO, WPE, true, false, false) \
    else                                                                                                  \
      CROWS_K(CI, CO, WPE, false, false, false) \
    DVAE_LAUNCH_CHECK(name);                                                                              \
    return 0;                                                                                             \
  }
  CONV_ROWS_LAUNCH(32, 64, 2, 2)
  CONV_ROWS_LAUNCH(16, 32, 2, 4)
#undef CONV_ROWS_LAUNCH
#undef CROWS_K
  if (g.in_y != nullptr) {
    dvae_set_error("%s: no kernel takes a BatchNorm-backward input at this shape (see dvae_conv4s2_bnin_supported)", name);
    return -EINVAL;
  }
  if (g.st != 0 && !(c1 && MODE == 0 && g.Cout <= 64 && g.Cout % 4 == 0 && aligned16(g.y) && (!bns || aligned16(g.bn_y)) &&
                     (g.st & ~(DVAE_ST_Y | DVAE_ST_BNY)) == 0)) {
    dvae_set_error("%s: bf16 storage is only taken by the few-channel row kernels and, for the output and the "
                   "BatchNorm-backward tensor, by the single-channel-input kernel", name);
    return -EINVAL;
  }
  if (c1 && MODE == 0 && g.Cout <= 64 && aligned16(g.y) && (!bns || aligned16(g.bn_y))) {
    // one partial-statistics row per workgroup, as many rows as the caller sized the buffer for
    const dim3 grid((unsigned)conv_parts_written(g.M, 1));
#define C1_LAUNCH(NCB)                                                                          \
  DVAE_WORK(4.0 * (double)g.M * 4 + ((g.st & DVAE_ST_Y) ? 2.0 : 4.0) * (double)g.M * g.Cout +            \
            (bns ? ((g.st & DVAE_ST_BNY) ? 2.0 : 4.0) * (double)g.M * g.Cout : 0.0), DVAE_UNIT_BYTE);     \
  if (g.st != 0) {                                                                              \
    if (NCB == 2 && bns && g.st == (DVAE_ST_Y | DVAE_ST_BNY))                                   \
      DVAE_LAUNCH((conv_c1_kernel<2, true, DVAE_ST_Y | DVAE_ST_BNY>), grid, dim3(256), 0, stream, g); \
    else if (NCB == 1 && !bns && g.st == DVAE_ST_Y)                                             \
      DVAE_LAUNCH((conv_c1_kernel<1, false, DVAE_ST_Y>), grid, dim3(256), 0, stream, g);        \
    else {                                                                                      \
      dvae_set_error("%s: the single-channel-input kernel takes bf16 storage for y and bn_y together (32 channels) " \
                     "or for a plain output of up to 16 channels", name);                       \
      return -EINVAL;                                                                           \
    }                                                                                           \
  } else if (bns) DVAE_LAUNCH((conv_c1_kernel<NCB, true>), grid, dim3(256), 0, stream, g);       \
  else DVAE_LAUNCH((conv_c1_kernel<NCB, false>), grid, dim3(256), 0, stream, g);
    if (g.Cout <= 16) {
      C1_LAUNCH(1)
    } else if (g.Cout <= 32) {
      C1_LAUNCH(2)
    } else {
      C1_LAUNCH(4)
    }
#undef C1_LAUNCH
  } else if (c1) {
    if (g.Cout > 32) LAUNCH(T64, true) else LAUNCH(T32, true)
  } else {
    const int ways = (g.ws != nullptr && !bns) ? conv_split_ways(g.M, phases, g.Cout, g.K) : 0;
    const int64_t out_floats = g.M * phases * g.Cout;
    if (ways >= 2 && (int64_t)ways * out_floats <= g.ws_floats && aligned16(g.ws) && aligned16(g.y)) {
      float* const stats_out = g.stats;
      g.slab = g.ws;
      g.slab_stride = out_floats;
      g.kt_per = (int)ceil_div64(ceil_div64(g.K, 32), ways);
      ksplit = ways;
      LAUNCH(T128, false)
      DVAE_LAUNCH_CHECK(name);
      const int64_t parts = conv_parts_written(g.M, phases), rows = g.M * phases;
      DVAE_LAUNCH(conv_slab_reduce_kernel, dim3((unsigned)parts), dim3(256), 0, stream, (const float4*)g.ws, ways,
                  out_floats / 4, g.Cout / 4, (float4*)g.y, stats_out, rows, ceil_div64(rows, parts));
      DVAE_LAUNCH_CHECK(name);
      return 0;
    }
    // the widest tile that still gives every compute unit a workgroup (all tiles are 128 rows high)
    const int64_t row_tiles = ceil_div64(g.M, 128) * phases;
    int bn = g.Cout > 64 ? 128 : (g.Cout > 32 ? 64 : 32);
    while (bn > 32 && row_tiles * ceil_div64(g.Cout, bn) < 256) bn >>= 1;
    // 128-wide tiles that fill the chip's 512 workgroup slots a whole number of times and then leave
    // between a fifth and seven tenths of a round: the remainder goes to a second launch with 64-wide
    // tiles, which is a shorter last round than a mostly idle one.  With statistics the two launches share the
    // layer's partial rows out in proportion to their M tiles: as many rows in all as one launch writes, the
    // second launch's numbered after the first's.
    const int64_t tiles128 = row_tiles * ceil_div64(g.Cout, 128);
    const int64_t full = tiles128 / 512, rest = tiles128 % 512;
    static const bool split_stats = [] { const char* v = getenv("DVAE_CONV_SPLIT_STATS"); return !(v && v[0] == '0'); }();
    if (bn == 128 && (g.stats == nullptr || (split_stats && !bns)) && full >= 1 && 5 * rest >= 512 && 10 * rest <= 7 * 512) {
      const int64_t per_mt = phases * ceil_div64(g.Cout, 128);          // workgroups per M tile
      const int split_mt = (int)(full * 512 / per_mt);
      const int all_mt = (int)ceil_div64(g.M, 128);
      const int64_t parts = conv_parts_written(g.M, phases) / phases;  // partial rows per phase of the whole layer
      int gx1 = (int)(parts * split_mt / all_mt);
      if (gx1 < 1) gx1 = 1;
      const int gx2 = (int)parts - gx1;
      if (split_mt > 0 && split_mt < all_mt && gx1 <= split_mt && gx2 >= 1 && gx2 <= all_mt - split_mt) {
        mt_begin = 0;
        mt_end = split_mt;
        gx_cap = gx1;
        g.part0 = 0;
        LAUNCH(T128, false)
        mt_begin = split_mt;
        mt_end = all_mt;
        gx_cap = gx2;
        g.part0 = gx1 * phases;
        LAUNCH(T64, false)
        DVAE_LAUNCH_CHECK(name);
        return 0;
      }
    }
    if (bn == 128) {
      LAUNCH(T128, false)
    } else if (bn == 64) {
      LAUNCH(T64, false)
    } else {
      LAUNCH(T32, false)
    }
  }
#undef LAUNCH
#undef CONV_FWD_LAUNCH
  DVAE_LAUNCH_CHECK(name);
  return 0;
}

struct BnBwdStat {   // see ConvArgs::bn_y
  const float* y;
  const float* scale;
  const float* shift;
  const float* mean;
  const float* invstd;
};
struct BnBwdInput {  // see ConvArgs::in_y
  const float* y;
  const float* a;
  const float* b;
  float* dx_out;
};

static int conv4s2_fwd_impl(const char* name, const float* x, const float* in_scale, const float* in_shift, int in_act,
                            const float* wp, float* y, float* stat_partials, int64_t N, int64_t H, int64_t W,
                            int64_t Cin, int64_t Cout, const BnBwdStat& bn, hipStream_t stream,
                            const BnBwdInput& bi = BnBwdInput{nullptr, nullptr, nullptr, nullptr},
                            float* ws = nullptr, int64_t ws_floats = 0, int storage = 0) {
  DVAE_REQUIRE(N > 0 && Cout > 0, "%s: empty", name);
  DVAE_REQUIRE(storage == 0 || dvae_mfma_precision() == DVAE_PRECISION_BF16, "%s: bf16 storage needs the bf16 MFMA mode", name);
  DVAE_REQUIRE(is_pow2(H) && is_pow2(W) && H >= 2 && W >= 2, "%s: H and W must be powers of two >= 2 (H=%ld W=%ld)", name,
               (long)H, (long)W);
  DVAE_REQUIRE(Cin == 1 || Cin % 4 == 0, "%s: Cin must be 1 or a multiple of 4 (Cin=%ld)", name, (long)Cin);
  DVAE_REQUIRE(!(Cin == 1 && (in_scale != nullptr || in_act != DVAE_ACT_NONE)),
               "%s: no load-time transform on single-channel input", name);
  DVAE_REQUIRE(aligned16(x) && aligned16(wp), "%s: x and wp must be 16-byte aligned", name);
  DVAE_REQUIRE(N * H * W * Cin < (1LL << 40), "%s: tensor too large", name);
  DVAE_REQUIRE(bn.y == nullptr || (stat_partials && bn.scale && bn.shift && bn.mean && bn.invstd),
               "%s: BatchNorm-backward statistics need the partials buffer and all four per-channel vectors", name);
  ConvArgs g;
  g.x = x; g.in_scale = in_scale; g.in_shift = in_shift; g.wp = wp; g.y = y; g.stats = stat_partials;
  g.in_act = in_act; g.H = (int)H; g.W = (int)W; g.lh = ilog2(H / 2); g.lw = ilog2(W / 2);
  g.Cin = (int)Cin; g.Cout = (int)Cout; g.K = (int)(16 * Cin);
  g.M = N * (H / 2) * (W / 2);
  g.lc = is_pow2(Cin) ? ilog2(Cin) : -1;
  g.cin32 = Cin % 32 == 0;
  g.bn_y = bn.y; g.bn_scale = bn.scale; g.bn_shift = bn.shift; g.bn_mean = bn.mean; g.bn_invstd = bn.invstd;
  g.in_y = bi.y; g.in_a = bi.a; g.in_b = bi.b; g.dx_out = bi.dx_out;
  g.slab = nullptr; g.slab_stride = 0; g.kt_per = 0; g.ws = ws; g.ws_floats = ws_floats; g.part0 = 0; g.st = storage;
  DVAE_REQUIRE(bi.y == nullptr || (in_scale && in_shift && bi.a && bi.b && bi.dx_out && aligned16(bi.y) &&
                                   aligned16(bi.a) && aligned16(bi.b) && aligned16(bi.dx_out)),
               "%s: a BatchNorm-backward input needs y, scale, shift, A, B and dx_out, 16-byte aligned", name);
  return launch_conv_fwd<0>(name, g, 1, stream);
}

static int deconv4s2_fwd_impl(const char* name, const float* x, const float* in_scale, const float* in_shift,
                              int in_act, const float* wp, float* y, float* stat_partials, int64_t N, int64_t H,
                              int64_t W, int64_t Cin, int64_t Cout, const BnBwdStat& bn, hipStream_t stream,
                              const BnBwdInput& bi = BnBwdInput{nullptr, nullptr, nullptr, nullptr},
                              float* ws = nullptr, int64_t ws_floats = 0, int storage = 0) {
  DVAE_REQUIRE(N > 0 && Cout > 0, "%s: empty", name);
  DVAE_REQUIRE(storage == 0 || dvae_mfma_precision() == DVAE_PRECISION_BF16, "%s: bf16 storage needs the bf16 MFMA mode", name);
  DVAE_REQUIRE(is_pow2(H) && is_pow2(W), "%s: H and W must be powers of two (H=%ld W=%ld)", name, (long)H, (long)W);
  DVAE_REQUIRE(Cin % 4 == 0, "%s: Cin must be a multiple of 4 (Cin=%ld)", name, (long)Cin);
  DVAE_REQUIRE(aligned16(x) && aligned16(wp), "%s: x and wp must be 16-byte aligned", name);
  DVAE_REQUIRE(bn.y == nullptr || (stat_partials && bn.scale && bn.shift && bn.mean && bn.invstd),
               "%s: BatchNorm-backward statistics need the partials buffer and all four per-channel vectors", name);
  ConvArgs g;
  g.x = x; g.in_scale = in_scale; g.in_shift = in_shift; g.wp = wp; g.y = y; g.stats = stat_partials;
  g.in_act = in_act; g.H = (int)H; g.W = (int)W; g.lh = ilog2(H); g.lw = ilog2(W);
  g.Cin = (int)Cin; g.Cout = (int)Cout; g.K = (int)(4 * Cin);
  g.M = N * H * W;
  g.lc = is_pow2(Cin) ? ilog2(Cin) : -1;
  g.cin32 = Cin % 32 == 0;
  g.bn_y = bn.y; g.bn_scale = bn.scale; g.bn_shift = bn.shift; g.bn_mean = bn.mean; g.bn_invstd = bn.invstd;
  g.in_y = bi.y; g.in_a = bi.a; g.in_b = bi.b; g.dx_out = bi.dx_out;
  g.slab = nullptr; g.slab_stride = 0; g.kt_per = 0; g.ws = ws; g.ws_floats = ws_floats; g.part0 = 0; g.st = storage;
  DVAE_REQUIRE(bi.y == nullptr || (in_scale && in_shift && bi.a && bi.b && bi.dx_out && aligned16(bi.y) &&
                                   aligned16(bi.a) && aligned16(bi.b) && aligned16(bi.dx_out)),
               "%s: a BatchNorm-backward input needs y, scale, shift, A, B and dx_out, 16-byte aligned", name);
  return launch_conv_fwd<1>(name, g, 4, stream);
}

extern "C" int dvae_conv4s2_fwd_f32(const float* x, const float* in_scale, const float* in_shift, int in_act,
                                    const float* wp, float* y, float* stat_partials, int64_t N, int64_t H, int64_t W,
                                    int64_t Cin, int64_t Cout, dvae_stream_t stream) {
  return conv4s2_fwd_impl("conv4s2_fwd", x, in_scale, in_shift, in_act, wp, y, stat_partials, N, H, W, Cin, Cout,
                          BnBwdStat{nullptr, nullptr, nullptr, nullptr, nullptr}, ST(stream));
}

extern "C" int dvae_deconv4s2_fwd_f32(const float* x, const float* in_scale, const float* in_shift, int in_act,
                                      const float* wp, float* y, float* stat_partials, int64_t N, int64_t H,
                                      int64_t W, int64_t Cin, int64_t Cout, dvae_stream_t stream) {
  return deconv4s2_fwd_impl("deconv4s2_fwd", x, in_scale, in_shift, in_act, wp, y, stat_partials, N, H, W, Cin, Cout,
                            BnBwdStat{nullptr, nullptr, nullptr, nullptr, nullptr}, ST(stream));
}

extern "C" int64_t dvae_conv_fwd_ws_floats(int transposed, int64_t N, int64_t H, int64_t W, int64_t Cin,
                                           int64_t Cout) {
  const int64_t M = transposed ? N * H * W : N * (H / 2) * (W / 2);
  const int phases = transposed ? 4 : 1;
  const int ways = Cin == 1 ? 0 : conv_split_ways(M, phases, Cout, transposed ? 4 * Cin : 16 * Cin);
  return (int64_t)ways * M * phases * Cout;
}

extern "C" int dvae_conv4s2_fwd_ws_f32(const float* x, const float* in_scale, const float* in_shift, int in_act,
                                       const float* wp, float* y, float* stat_partials, int64_t N, int64_t H,
                                       int64_t W, int64_t Cin, int64_t Cout, float* ws, int64_t ws_floats,
                                       int storage, dvae_stream_t stream) {
  return conv4s2_fwd_impl("conv4s2_fwd", x, in_scale, in_shift, in_act, wp, y, stat_partials, N, H, W, Cin, Cout,
                          BnBwdStat{nullptr, nullptr, nullptr, nullptr, nullptr}, ST(stream),
                          BnBwdInput{nullptr, nullptr, nullptr, nullptr}, ws, ws_floats, storage);
}

extern "C" int dvae_deconv4s2_fwd_ws_f32(const float* x, const float* in_scale, const float* in_shift, int in_act,
                                         const float* wp, float* y, float* stat_partials, int64_t N, int64_t H,
                                         int64_t W, int64_t Cin, int64_t Cout, float* ws, int64_t ws_floats,
                                         int storage, dvae_stream_t stream) {
  return deconv4s2_fwd_impl("deconv4s2_fwd", x, in_scale, in_shift, in_act, wp, y, stat_partials, N, H, W, Cin, Cout,
                            BnBwdStat{nullptr, nullptr, nullptr, nullptr, nullptr}, ST(stream),
                            BnBwdInput{nullptr, nullptr, nullptr, nullptr}, ws, ws_floats, storage);
}

extern "C" int dvae_conv4s2_dgrad_bnstat_f32(const float* x, const float* wp, float* y, float* stat_partials,
                                             int64_t N, int64_t H, int64_t W, int64_t Cin, int64_t Cout,
                                             const float* bn_y, const float* bn_scale, const float* bn_shift,
                                             const float* bn_mean, const float* bn_invstd, int storage,
                                             dvae_stream_t stream) {
  DVAE_REQUIRE(bn_y != nullptr, "conv4s2_dgrad_bnstat: bn_y is null");
  return conv4s2_fwd_impl("conv4s2_dgrad_bnstat", x, nullptr, nullptr, DVAE_ACT_NONE, wp, y, stat_partials, N, H, W,
                          Cin, Cout, BnBwdStat{bn_y, bn_scale, bn_shift, bn_mean, bn_invstd}, ST(stream),
                          BnBwdInput{nullptr, nullptr, nullptr, nullptr}, nullptr, 0, storage);
}

extern "C" int dvae_conv4s2_bnin_supported(int64_t H, int64_t W, int64_t Cin, int64_t Cout) {
  (void)H;
  return Cin == 32 && Cout == 64 && W == 32;
}

extern "C" int dvae_conv4s2_dgrad_bnstat_bnin_f32(const float* du_in, const float* in_y, const float* in_scale,
                                                  const float* in_shift, const float* in_a, const float* in_b,
                                                  float* dx_out, const float* wp, float* y, float* stat_partials,
                                                  int64_t N, int64_t H, int64_t W, int64_t Cin, int64_t Cout,
                                                  const float* bn_y, const float* bn_scale, const float* bn_shift,
                                                  const float* bn_mean, const float* bn_invstd, int storage,
                                                  dvae_stream_t stream) {
  DVAE_REQUIRE(bn_y != nullptr && in_y != nullptr, "conv4s2_dgrad_bnstat_bnin: bn_y / in_y is null");
  DVAE_REQUIRE(dvae_conv4s2_bnin_supported(H, W, Cin, Cout), "conv4s2_dgrad_bnstat_bnin: unsupported shape or precision");
  return conv4s2_fwd_impl("conv4s2_dgrad_bnstat_bnin", du_in, in_scale, in_shift, DVAE_ACT_NONE, wp, y, stat_partials, N,
                          H, W, Cin, Cout, BnBwdStat{bn_y, bn_scale, bn_shift, bn_mean, bn_invstd}, ST(stream),
                          BnBwdInput{in_y, in_a, in_b, dx_out}, nullptr, 0, storage);
}

extern "C" int dvae_deconv4s2_dgrad_bnstat_f32(const float* x, const float* wp, float* y, float* stat_partials,
                                               int64_t N, int64_t H, int64_t W, int64_t Cin, int64_t Cout,
                                               const float* bn_y, const float* bn_scale, const float* bn_shift,
                                               const float* bn_mean, const float* bn_invstd, dvae_stream_t stream) {
  DVAE_REQUIRE(bn_y != nullptr, "deconv4s2_dgrad_bnstat: bn_y is null");
  return deconv4s2_fwd_impl("deconv4s2_dgrad_bnstat", x, nullptr, nullptr, DVAE_ACT_NONE, wp, y, stat_partials, N, H,
                            W, Cin, Cout, BnBwdStat{bn_y, bn_scale, bn_shift, bn_mean, bn_invstd}, ST(stream));
}

extern "C" int dvae_deconv4s2_bnin_supported(int64_t H, int64_t W, int64_t Cin, int64_t Cout) {
  (void)H;
  return Cin == 32 && Cout == 16 && (W == 16 || W == 32);
}

extern "C" int dvae_deconv4s2_dgrad_bnstat_bnin_f32(const float* du_in, const float* in_y, const float* in_scale,
                                                    const float* in_shift, const float* in_a, const float* in_b,
                                                    float* dx_out, const float* wp, float* y, float* stat_partials,
                                                    int64_t N, int64_t H, int64_t W, int64_t Cin, int64_t Cout,
                                                    const float* bn_y, const float* bn_scale, const float* bn_shift,
                                                    const float* bn_mean, const float* bn_invstd, int storage,
                                                    dvae_stream_t stream) {
  DVAE_REQUIRE(bn_y != nullptr && in_y != nullptr, "deconv4s2_dgrad_bnstat_bnin: bn_y / in_y is null");
  DVAE_REQUIRE(dvae_deconv4s2_bnin_supported(H, W, Cin, Cout), "deconv4s2_dgrad_bnstat_bnin: unsupported shape or precision");
  return deconv4s2_fwd_impl("deconv4s2_dgrad_bnstat_bnin", du_in, in_scale, in_shift, DVAE_ACT_NONE, wp, y, stat_partials,
                            N, H, W, Cin, Cout, BnBwdStat{bn_y, bn_scale, bn_shift, bn_mean, bn_invstd}, ST(stream),
                            BnBwdInput{in_y, in_a, in_b, dx_out}, nullptr, 0, storage);
}

// Host helper used by the Python side through dvae_conv_stat_parts is an upper bound; the exact
// number of rows written is needed by bn_finalize, so expose it.
extern "C" int64_t dvae_conv_stat_parts_written(int64_t M_rows_per_phase, int phases) {
  return conv_parts_written(M_rows_per_phase, phases);
}

// ------------------------------------------------------------------------------------------------
// weight gradient
// ------------------------------------------------------------------------------------------------
struct WgradArgs {
  const float* P;
  const float* p_scale;
  const float* p_shift;
  const float* Q;
  const float* q_scale;
  const float* q_shift;
  float* slab;  // [split][C1][16*C2]
  int64_t Mtot;  // N*Hs*Ws
  int p_act, q_act;
  int Hs, Ws, lh, lw;
  int C1, C2, NC;  // NC = 16*C2
  int ktiles_per_split;
  // P not yet through its BatchNorm backward (single-channel-Q kernel only): P is du, p_y the layer's
  // raw output, p_scale/p_shift its forward affine, and P becomes scale*dz + A*y + B as it is loaded
  const float* p_y;
  const float* p_a;
  const float* p_b;
  // the same for Q (wgrad_rows8_kernel only); the finished rows of dL/dy are also written to dx_out
  const float* q_y;
  const float* q_a;
  const float* q_b;
  float* dx_out;
  int st;       // bf16 storage of P (DVAE_ST_X), Q (DVAE_ST_Y), p_y (DVAE_ST_BNY): see ConvArgs::st
};

// Both operands of the weight-gradient GEMM are row-contiguous tiles whose column position is
// fixed per thread for the whole K loop (256 % (R/4) == 0), so everything that depends on the column
// (channel, filter tap, pixel offset, which border rows the tap leaves the frame on) is computed
// once; per K tile only the pixel index m moves.  With Hs, Ws powers of two the Q pixel of (m, tap)
// sits at element ((4m - 2b) + dh*WQ + dw) * C2 of Q, b = m mod Ws.
template <int V4, int R, bool KADJ>
struct WgradPLoader {
  const WgradArgs& g;
  int c;          // channel of this thread's 4-float chunk
  int kr0;
  bool col_ok;
  unsigned mask;
  float4 sc, sh;
  __device__ __forceinline__ WgradPLoader(const WgradArgs& g_, int64_t c0) : g(g_) {
    c = (int)c0 + 4 * (threadIdx.x % (R / 4));
    kr0 = threadIdx.x / (R / 4);
    col_ok = c < g.C1;
    sc = make_float4(1.f, 1.f, 1.f, 1.f);
    sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col_ok && g.p_scale) {
      sc = *reinterpret_cast<const float4*>(g.p_scale + c);
      sh = *reinterpret_cast<const float4*>(g.p_shift + c);
    }
  }
  __device__ __forceinline__ void issue(int kt, f32x4 (&r)[V4]) {
    mask = 0;
#pragma unroll
    for (int v = 0; v < V4; ++v) {
      const unsigned m = (unsigned)(kt * 32) + (unsigned)(KADJ ? V4 * kr0 + v : v * (1024 / R) + kr0);
      const bool eok = col_ok && m < (unsigned)g.Mtot;
      const float* p = eok ? g.P + ((uint64_t)m * (unsigned)g.C1 + (unsigned)c) : g.P;
      r[v] = ldg4(p);
      mask |= (unsigned)eok << v;
    }
  }
  struct Tag { unsigned mask; };
  __device__ __forceinline__ Tag tag() const { return Tag{mask}; }
  __device__ __forceinline__ void set_tag(const Tag& t) { mask = t.mask; }
  __device__ __forceinline__ void finish(f32x4 (&r)[V4]) const {
#pragma unroll
    for (int v = 0; v < V4; ++v) {
      f32x4 val = r[v];
      if (g.p_scale) {
        val.x = fmaf(val.x, sc.x, sh.x); val.y = fmaf(val.y, sc.y, sh.y);
        val.z = fmaf(val.z, sc.z, sh.z); val.w = fmaf(val.w, sc.w, sh.w);
      }
      if (g.p_act == DVAE_ACT_LRELU) {
        val.x = lrelu(val.x); val.y = lrelu(val.y); val.z = lrelu(val.z); val.w = lrelu(val.w);
      }
      r[v] = zero_unless(val, (mask >> v) & 1u);
    }
  }
};

template <int V4, int R, bool KADJ>
struct WgradQLoader {
  const WgradArgs& g;
  int kr0;
  int coff;        // (dh*WQ + dw)*C2 + c2
  int bad_a, bad_b;  // the border row / column on which this tap leaves the frame (-1: never)
  bool col_ok;
  unsigned mask;
  float4 sc, sh;
  __device__ __forceinline__ WgradQLoader(const WgradArgs& g_, int64_t n0) : g(g_) {
    const int col = (int)n0 + 4 * (threadIdx.x % (R / 4));
    kr0 = threadIdx.x / (R / 4);
    col_ok = col < g.NC;
    const int tp = col / g.C2;
    const int c2 = col - tp * g.C2;
    const int dh = (tp >> 2) - 1, dw = (tp & 3) - 1;
    coff = (dh * 2 * g.Ws + dw) * g.C2 + c2;
    bad_a = dh < 0 ? 0 : (dh == 2 ? g.Hs - 1 : -1);
    bad_b = dw < 0 ? 0 : (dw == 2 ? g.Ws - 1 : -1);
    sc = make_float4(1.f, 1.f, 1.f, 1.f);
    sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col_ok && g.q_scale) {
      sc = *reinterpret_cast<const float4*>(g.q_scale + c2);
      sh = *reinterpret_cast<const float4*>(g.q_shift + c2);
    }
  }
  __device__ __forceinline__ void issue(int kt, f32x4 (&r)[V4]) {
    mask = 0;
#pragma unroll
    for (int v = 0; v < V4; ++v) {
      const unsigned m = (unsigned)(kt * 32) + (unsigned)(KADJ ? V4 * kr0 + v : v * (1024 / R) + kr0);
      const int b = (int)(m & (unsigned)(g.Ws - 1));
      const int a = (int)((m >> g.lw) & (unsigned)(g.Hs - 1));
      const bool eok = col_ok && m < (unsigned)g.Mtot && a != bad_a && b != bad_b;
      const unsigned e = 4u * m - 2u * (unsigned)b;
      const float* p = eok ? g.Q + ((int64_t)((uint64_t)e * (unsigned)g.C2) + coff) : g.Q;
      r[v] = ldg4(p);
      mask |= (unsigned)eok << v;
    }
  }
  struct Tag { unsigned mask; };
  __device__ __forceinline__ Tag tag() const { return Tag{mask}; }
  __device__ __forceinline__ void set_tag(const Tag& t) { mask = t.mask; }
  __device__ __forceinline__ void finish(f32x4 (&r)[V4]) const {
#pragma unroll
    for (int v = 0; v < V4; ++v) {
      f32x4 val = r[v];
      if (g.q_scale) {
        val.x = fmaf(val.x, sc.x, sh.x); val.y = fmaf(val.y, sc.y, sh.y);
        val.z = fmaf(val.z, sc.z, sh.z); val.w = fmaf(val.w, sc.w, sh.w);
      }
      if (g.q_act == DVAE_ACT_LRELU) {
        val.x = lrelu(val.x); val.y = lrelu(val.y); val.z = lrelu(val.z); val.w = lrelu(val.w);
      }
      r[v] = zero_unless(val, (mask >> v) & 1u);
    }
  }
};

template <class T, int PREC>      // PREC_F32, PREC_BF16 or PREC_SPLIT
__global__ void __launch_bounds__(256, 2) conv_wgrad_kernel(WgradArgs g) {
  constexpr int BF = PREC;
  constexpr bool BFL = PREC == PREC_BF16;       // the loaders' bf16 chunk order
  __shared__ __attribute__((aligned(16))) float smem[2 * Core<T, BF>::STAGE];
  const int64_t m0 = (int64_t)blockIdx.y * T::BM, n0 = (int64_t)blockIdx.x * T::BN;
  const int ktiles = (int)((g.Mtot + 31) / 32);
  const int kt_begin = blockIdx.z * g.ktiles_per_split;
  int kt_end = kt_begin + g.ktiles_per_split;
  if (kt_end > ktiles) kt_end = ktiles;
  f32x16 acc[T::TM][T::TN];
  zero_acc<T>(acc);
  WgradPLoader<T::A_V4, T::BM, BFL> la(g, m0);
  WgradQLoader<T::B_V4, T::BN, BFL> lb(g, n0);
  // two K tiles of loads in flight: the narrow tiles multiply for less time than a load takes (+5 % on
  // the 64x128 tile), and the 128x128 tile gains 3-4 % with its light loaders (190 registers, no spill);
  // the bf16 variant of the wide tile keeps one tile in flight
  if (T::TM * T::TN <= 2 || PREC == PREC_F32) gemm_mainloop_ahead2<T, BF>(la, lb, kt_begin, kt_end, smem, acc);
  else gemm_mainloop<T, BF>(la, lb, kt_begin, kt_end, smem, acc);
  float* dst = g.slab + (int64_t)blockIdx.z * g.C1 * g.NC;
  for_each_acc<T>(acc, [&](int r, int c, float v) {
    const int64_t m = m0 + r, n = n0 + c;
    if (m < g.C1 && n < g.NC) dst[m * g.NC + n] = v;
  });
}

// Single-channel Q (first encoder conv, final decoder layer): dw[c1][tap] = sum_m P[m][c1] * q(m,tap).
// Sixteen taps make this a reduction bound by the read of P, not a GEMM worth tiling through LDS:
// each wave streams its own pixels straight from memory into v_mfma_f32_16x16x4_f32 operands
// (A = P^T: 16 channels x 4 pixels, exactly how P lies in memory; B = 4 pixels x 16 taps gathered
// from the single-channel image), accumulates [channels][taps] in registers over its whole pixel
// range and the four waves' partials are summed through LDS once at the end.
#define WG1_MAX_CB 8   // up to 128 channels
template <int NCB, int U, bool BNP, int ST = 0>
__global__ void __launch_bounds__(256) conv_wgrad_c1_kernel(WgradArgs g) {
  __shared__ float red[4 * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i16 = lane & 15, kq = lane >> 4;
  constexpr int ncb = NCB;
  // P stored as bf16 with 32 channels: row i of channel block cb stands for channel 2i + cb instead of 16cb + i, so
  // that one dword load brings a lane both of its channels (2-byte loads cost as much per element as dwords)
  constexpr bool PAIR = (ST & DVAE_ST_X) != 0 && NCB == 2;
  // ... with 16 channels two neighbouring lanes fetch the same dword and take a half each (P and p_y alike)
  constexpr bool HALF = (ST & DVAE_ST_X) != 0 && NCB == 1;
  static_assert(!PAIR || !BNP, "bf16 P with 32 channels: the variant without a BatchNorm-backward P");
  static_assert(!HALF || (ST & DVAE_ST_BNY) != 0 || !BNP, "bf16 P with 16 channels: p_y is bf16 as well");
  static_assert((ST & DVAE_ST_X) != 0 || ST == 0, "p_y is only stored as bf16 together with P");
  auto chan = [](int cb, int i) { return PAIR ? 2 * i + cb : cb * 16 + i; };
  const int HQ = 2 * g.Hs, WQ = 2 * g.Ws;
  const int kh = i16 >> 2, kw = i16 & 3;       // this lane's tap as a B-operand column
  f32x4 acc[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) acc[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float psc[NCB], psh[NCB], pca[NCB], pcb[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
    const int c = chan(cb, i16);
    const bool ok = cb < ncb && c < g.C1 && g.p_scale != nullptr;
    psc[cb] = ok ? g.p_scale[c] : 1.f;
    psh[cb] = ok ? g.p_shift[c] : 0.f;
    pca[cb] = BNP && ok ? g.p_a[c] : 0.f;
    pcb[cb] = BNP && ok ? g.p_b[c] : 0.f;
  }
  const int64_t nblk = (g.Mtot + 3) / 4;        // blocks of 4 pixels
  const int64_t wid = (int64_t)blockIdx.x * 4 + wave, nw = (int64_t)gridDim.x * 4;
  // U pixel blocks per iteration: all their loads are issued before the first MFMA
  for (int64_t blk = wid; blk < nblk; blk += U * nw) {
    float qv[U], pv[U][NCB], yv[U][BNP ? NCB : 1];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t m = (blk + u * nw) * 4 + kq;
      const bool mok = (blk + u * nw) < nblk && m < g.Mtot;
      qv[u] = 0.f;
      if (mok) {
        const int b = (int)(m & ((1 << g.lw) - 1));
        const int a = (int)((m >> g.lw) & ((1 << g.lh) - 1));
        const int64_t n = m >> (g.lw + g.lh);
        const int hi = 2 * a - 1 + kh, wi = 2 * b - 1 + kw;
        if (hi >= 0 && hi < HQ && wi >= 0 && wi < WQ) qv[u] = g.Q[(n * HQ + hi) * WQ + wi];
      }
      if constexpr (PAIR) {
        pv[u][0] = mok ? __uint_as_float(reinterpret_cast<const unsigned*>(g.P)[m * 16 + i16]) : 0.f;   // (C1 == 32)
      } else if constexpr (HALF) {
        pv[u][0] = mok ? __uint_as_float(reinterpret_cast<const unsigned*>(g.P)[m * 8 + (i16 >> 1)]) : 0.f;   // (C1 == 16)
        if (BNP) yv[u][0] = mok ? __uint_as_float(reinterpret_cast<const unsigned*>(g.p_y)[m * 8 + (i16 >> 1)]) : 0.f;
      } else {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
          pv[u][cb] = 0.f;
          if (cb < ncb) {
            const int c = cb * 16 + i16;
            if (mok && c < g.C1) {
              pv[u][cb] = g.P[m * g.C1 + c];
              if (BNP) yv[u][cb] = g.p_y[m * g.C1 + c];
            }
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t m = (blk + u * nw) * 4 + kq;
      const bool mok = (blk + u * nw) < nblk && m < g.Mtot;
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        if (cb < ncb) {
          auto half_of = [&](float raw) {       // this lane's bf16 of the dword it shares with its neighbour
            const unsigned w = __float_as_uint(raw);
            return __uint_as_float((i16 & 1) ? (w & 0xffff0000u) : (w << 16));
          };
          float v = PAIR ? __uint_as_float(cb == 0 ? __float_as_uint(pv[u][0]) << 16 : __float_as_uint(pv[u][0]) & 0xffff0000u)
                    : HALF ? half_of(pv[u][0]) : pv[u][cb];
          if (mok && chan(cb, i16) < g.C1) {
            if (BNP) {
              const float y = HALF ? half_of(yv[u][cb]) : yv[u][cb], zz = fmaf(y, psc[cb], psh[cb]);
              const float dz = zz > 0.f ? v : v * LRELU_SLOPE;
              v = fmaf(psc[cb], dz, fmaf(pca[cb], y, pcb[cb]));
            } else {
              if (g.p_scale) v = fmaf(v, psc[cb], psh[cb]);
              if (g.p_act == DVAE_ACT_LRELU) v = lrelu(v);
            }
          } else {
            v = 0.f;
          }
          acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(v, qv[u], acc[cb], 0, 0, 0);
        }
      }
    }
  }
  // D layout: column (tap) = lane & 15, row (channel in block) = (lane >> 4) * 4 + reg.
  // Sum the four waves' partial tiles through LDS, one 16-channel block at a time.
  float* dst = g.slab + (int64_t)blockIdx.x * g.C1 * 16;
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
    if (cb < ncb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(wave * 16 + kq * 4 + r) * 16 + i16] = acc[cb][r];
      __syncthreads();
      const int c = chan(cb, threadIdx.x >> 4), tap = threadIdx.x & 15;
      if (c < g.C1) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) s += red[(w * 16 + (threadIdx.x >> 4)) * 16 + tap];
        dst[c * 16 + tap] = s;
      }
      __syncthreads();
    }
  }
}

// dw[(c1*C2 + c2)*16 + tap] = sum_z slab[z][c1][tap*C2 + c2]; threads walk the slab layout so the
// reads of every slab are contiguous
// Sum of the partial gradients.  Thread -> one gradient element (consecutive threads read consecutive
// floats of a slab), blockIdx.y -> a run of `zchunk` slabs.  With one run the sums go to dw in the torch
// layout; with several (many slabs of a small gradient: a single run would be one long chain of
// dependent loads in a handful of threads) each run leaves a slab-shaped partial in `part`, and a
// second launch over those finishes.  The order of the additions is fixed either way.
__global__ void __launch_bounds__(256)
wgrad_reduce_kernel(const float* __restrict__ slab, int split, int zchunk, float* __restrict__ part,
                    float* __restrict__ dw, int64_t C1, int64_t C2) {
  const int64_t total = C1 * C2 * 16;
  const int z0 = (int)blockIdx.y * zchunk, z1 = z0 + zchunk < split ? z0 + zchunk : split;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int z = z0;
    for (; z + 4 <= z1; z += 4) {
      s0 += slab[(int64_t)z * total + i];
      s1 += slab[(int64_t)(z + 1) * total + i];
      s2 += slab[(int64_t)(z + 2) * total + i];
      s3 += slab[(int64_t)(z + 3) * total + i];
    }
    for (; z < z1; ++z) s0 += slab[(int64_t)z * total + i];
    const float v = (s0 + s1) + (s2 + s3);
    if (gridDim.y == 1) {
      const int64_t c2 = i % C2, tap = (i / C2) & 15, c1 = i / (16 * C2);
      dw[(c1 * C2 + c2) * 16 + tap] = v;
    } else {
      part[(int64_t)blockIdx.y * total + i] = v;
    }
  }
}

// slabs per run of the first stage (0: a single stage), and the number of runs
#define WGRAD_REDUCE_RUNS 32
static int wgrad_reduce_runs(int split, int64_t total) {
  // worth it when the single-stage launch would have few threads and long chains
  return (split >= 128 && total <= 65536) ? WGRAD_REDUCE_RUNS : 1;
}

// Weight gradient with few channels (C1*C2 <= 2048): the whole gradient [C1][16 taps][C2] fits in the
// accumulators of one workgroup, so a workgroup keeps it there and streams a contiguous run of P rows
// past it (K = pixels).  Rows of P and of Q go through LDS rings once, in their memory layout
// [pixel][channel] with the producing layers' transforms applied on the way in; a lane reads its MFMA
// operands with 4-byte LDS reads (A: P[pixel 4kq+e][c1], B: Q[pixel of the tap][c2], pixel strides
// chosen so that the four pixel quads of a read fall into different banks), the zero pixels left and
// right of every Q row stand for the frame's edges, and taps above/below the frame are skipped.  Wave
// w owns the sixteen rows 16 (w % NRB) of the gradient and an equal share of the taps.  Each workgroup
// writes its partial gradient as one slab for wgrad_reduce_kernel.
template <int C1, int C2, int WAVES_PER_EU, bool XP, bool XQ, int BF, int ST = 0>      // BF: 0 fp32, 1 bf16, 2 fp32 from bf16 terms
__global__ void __launch_bounds__(256, WAVES_PER_EU) wgrad_rows_kernel(WgradArgs g) {
  constexpr int NRB = C1 / 16;            // row blocks of the gradient
  constexpr int CPARTS = 4 / NRB;         // the taps are shared between this many waves per row block
  constexpr int DHN = 4 / CPARTS;         // filter rows per wave
  constexpr int NCB = C2 / 16;
  constexpr int PSP = C1 + 4, PSQ = C2 + 2;
  constexpr int P4 = C1 / 4, Q4 = C2 / 4;
  constexpr int NLP = (32 * P4 + 255) / 256, NLQ = (64 * Q4 + 255) / 256;   // Ws <= 32
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int Ws = g.Ws, Hs = g.Hs, WQ = 2 * Ws;
  const int pslot = Ws * PSP, qslot = (WQ + 2) * PSQ;
  float* const pring = lds;                // 4 slots
  float* const qring = lds + 4 * pslot;    // 8 slots
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i16 = lane & 15, kq = lane >> 4;
  const int rb = wave % NRB, dh0 = (wave / NRB) * DHN;

  const int lp4 = threadIdx.x % P4, lq4 = threadIdx.x % Q4;
  f32x4 psc = (f32x4){1.f, 1.f, 1.f, 1.f}, psh = (f32x4){0.f, 0.f, 0.f, 0.f}, qsc = psc, qsh = psh;
  const float pslope = g.p_act == DVAE_ACT_LRELU ? LRELU_SLOPE : 1.f;
  const float qslope = g.q_act == DVAE_ACT_LRELU ? LRELU_SLOPE : 1.f;
  if (XP && g.p_scale) {
    psc = ldg4(g.p_scale + 4 * lp4);
    psh = ldg4(g.p_shift + 4 * lp4);
  }
  if (XQ && g.q_scale) {
    qsc = ldg4(g.q_scale + 4 * lq4);
    qsh = ldg4(g.q_shift + 4 * lq4);
  }
  // zero pixels left (column -1) and right (column WQ) of every Q slot
  for (int i = threadIdx.x; i < 8 * 2 * PSQ; i += 256) {
    const int slot = i / (2 * PSQ), r = i % (2 * PSQ);
    qring[slot * qslot + (r < PSQ ? 0 : (WQ + 1) * PSQ) + r % PSQ] = 0.f;
  }

  f32x4 acc[DHN][4][NCB];
#pragma unroll
  for (int a = 0; a < DHN; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int c = 0; c < NCB; ++c) acc[a][b][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // P rows are numbered over the whole batch (row = image * Hs + a); P row r meets Q rows 2r-1 .. 2r+2
  const int64_t rows = g.Mtot >> g.lw, qrows = 2 * rows;
  const int64_t per = (rows + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = blockIdx.x * per;
  const int64_t r1 = r0 + per < rows ? r0 + per : rows;
  const int p_pieces = Ws * P4, q_pieces = WQ * Q4;

  auto transform = [](f32x4 v, const f32x4& sc, const f32x4& sh, float slope) {
    const f32x4 z = v * sc + sh;         // slope <= 1: lrelu(z) = max(z, slope z)
    const f32x4 zs = z * slope;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaxf(z[e], zs[e]);
    return v;
  };
  static_assert(BF == 1 || ST == 0, "bf16 storage goes with bf16 MFMA operands");
  constexpr bool pbf = (ST & DVAE_ST_X) != 0, qbf = (ST & DVAE_ST_Y) != 0;
  auto fetch_p = [&](int64_t row, f32x4 (&raw)[NLP]) {
    const bool rok = row >= 0 && row < rows;
    const int64_t src = (rok ? row : 0) * (int64_t)p_pieces * 4;
#pragma unroll
    for (int j = 0; j < NLP; ++j) {
      const int f = threadIdx.x + 256 * j;
      raw[j] = ld4s(g.P, src + 4 * (rok && f < p_pieces ? f : 0), pbf);
    }
  };
  auto deposit_p = [&](int64_t row, const f32x4 (&raw)[NLP]) {
    float* dst = pring + (row & 3) * pslot;
#pragma unroll
    for (int j = 0; j < NLP; ++j) {
      const int f = threadIdx.x + 256 * j;
      if (f < p_pieces) {
        const f32x4 w = widen4s(raw[j], pbf);
        const f32x4 v = XP ? transform(w, psc, psh, pslope) : w;
        *reinterpret_cast<f32x4*>(dst + (f / P4) * PSP + 4 * lp4) = v;
      }
    }
  };
  auto fetch_q = [&](int64_t row, f32x4 (&raw)[NLQ]) {
    const bool rok = row >= 0 && row < qrows;
    const int64_t src = (rok ? row : 0) * (int64_t)q_pieces * 4;
#pragma unroll
    for (int j = 0; j < NLQ; ++j) {
      const int f = threadIdx.x + 256 * j;
      raw[j] = ld4s(g.Q, src + 4 * (rok && f < q_pieces ? f : 0), qbf);
    }
  };
  auto deposit_q = [&](int64_t row, const f32x4 (&raw)[NLQ]) {
    float* dst = qring + (row & 7) * qslot;
#pragma unroll
    for (int j = 0; j < NLQ; ++j) {
      const int f = threadIdx.x + 256 * j;
      if (f < q_pieces) {
        const f32x4 w = widen4s(raw[j], qbf);
        const f32x4 v = XQ ? transform(w, qsc, qsh, qslope) : w;
        float* d = dst + (1 + f / Q4) * PSQ + 4 * lq4;       // 8-byte aligned (PSQ is even)
        *reinterpret_cast<float2*>(d) = make_float2(v[0], v[1]);
        *reinterpret_cast<float2*>(d + 2) = make_float2(v[2], v[3]);
      }
    }
  };
  f32x4 rawp[NLP], rawq0[NLQ], rawq1[NLQ];
  if (r0 < r1) {
    fetch_p(r0, rawp);
    fetch_q(2 * r0 - 1, rawq0);
    fetch_q(2 * r0, rawq1);
    deposit_p(r0, rawp);
    deposit_q(2 * r0 - 1, rawq0);
    deposit_q(2 * r0, rawq1);
    fetch_q(2 * r0 + 1, rawq0);
    fetch_q(2 * r0 + 2, rawq1);
    deposit_q(2 * r0 + 1, rawq0);
    deposit_q(2 * r0 + 2, rawq1);
  }
  __syncthreads();
  for (int64_t r = r0; r < r1; ++r) {
    fetch_p(r + 1, rawp);
    fetch_q(2 * r + 3, rawq0);
    fetch_q(2 * r + 4, rawq1);
    const int a = (int)r & (Hs - 1);
    const float* prow = pring + (r & 3) * pslot;
    for (int j = 0; j < Ws / 16; ++j) {
      const int b0 = 16 * j + 4 * kq;              // this lane's pixels are b0 + e
      float av[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) av[e] = prow[(b0 + e) * PSP + 16 * rb + i16];
#pragma unroll
      for (int dl = 0; dl < DHN; ++dl) {
        const int dh = dh0 + dl, hin = 2 * a - 1 + dh;
        if (hin < 0 || hin >= 2 * Hs) continue;    // the tap row is above or below the frame
        const float* qrow = qring + ((2 * r - 1 + dh) & 7) * qslot;
        // Q pixel of (b, dw) is column 2b-1+dw, stored at index 2b+dw
        if constexpr (BF == 2) {
          const SplitQuad af = split_quad(av[0], av[1], av[2], av[3]);
          // two taps at a time: four (two) independent accumulators per term, few split operands alive
#pragma unroll
          for (int dw0 = 0; dw0 < 4; dw0 += 2) {
            SplitQuad bs[2][NCB];
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
              for (int cb = 0; cb < NCB; ++cb) {
                const float* qa = qrow + (2 * b0 + dw0 + d) * PSQ + 16 * cb + i16;
                bs[d][cb] = split_quad(qa[0], qa[2 * PSQ], qa[4 * PSQ], qa[6 * PSQ]);
              }
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
              for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb)
                  acc[dl][dw0 + d][cb] =
                      RowOps<true>::mma(af.t[SPLIT_TA[t]], bs[d][cb].t[SPLIT_TB[t]], acc[dl][dw0 + d][cb]);
          }
          continue;
        }
        if constexpr (BF == 1) {
          // the lane's four pixels are the four k of one bf16 MFMA operand
          const s16x4 af = bf_quad(av[0], av[1], av[2], av[3]);
#pragma unroll
          for (int dw = 0; dw < 4; ++dw) {
            float bq[4][NCB];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int cb = 0; cb < NCB; ++cb) bq[e][cb] = qrow[(2 * (b0 + e) + dw) * PSQ + 16 * cb + i16];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
              acc[dl][dw][cb] = RowOps<true>::mma(af, bf_quad(bq[0][cb], bq[1][cb], bq[2][cb], bq[3][cb]), acc[dl][dw][cb]);
          }
          continue;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float bv[4][NCB];
#pragma unroll
          for (int dw = 0; dw < 4; ++dw)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) bv[dw][cb] = qrow[(2 * (b0 + e) + dw) * PSQ + 16 * cb + i16];
#pragma unroll
          for (int dw = 0; dw < 4; ++dw)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
              acc[dl][dw][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[dw][cb], acc[dl][dw][cb], 0, 0, 0);
        }
      }
    }
    deposit_p(r + 1, rawp);
    deposit_q(2 * r + 3, rawq0);
    deposit_q(2 * r + 4, rawq1);
    __syncthreads();
  }
  // D: row = c1 16rb + 4kq + e, column = c2 16cb + i16 of tap 4dh + dw
  float* slab = g.slab + (int64_t)blockIdx.x * C1 * g.NC;
#pragma unroll
  for (int dl = 0; dl < DHN; ++dl)
#pragma unroll
    for (int dw = 0; dw < 4; ++dw)
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          slab[(int64_t)(16 * rb + 4 * kq + e) * g.NC + (4 * (dh0 + dl) + dw) * C2 + 16 * cb + i16] = acc[dl][dw][cb][e];
}

// The same for C1 = 128, C2 = 64 on 8-pixel-wide P (the gradient is four times what one workgroup's
// accumulators hold): a workgroup takes one filter row dh (blockIdx.x & 3), i.e. the columns of taps
// 4dh .. 4dh+3, so for a P row it needs exactly one Q row (2a-1+dh).  A step is a pair of P rows
// (sixteen pixels = one K group); Q rows that fall outside the frame are deposited as zeros.  Wave w owns
// gradient rows 32w .. 32w+31.  The four workgroups of a split fill one slab between them.
//
// BNQ: Q is a gradient that has not been through its layer's BatchNorm backward (Q = du, q_y the
// layer's raw output; a row becomes scale*dz + A*y + B on its way into LDS).  Every Q row is met by
// exactly two filter rows; the workgroups of filter rows 1 and 2 between them meet each row once and
// write the finished rows to dx_out for the data-gradient kernel.
template <bool XP, bool XQ, bool BNQ, int BF, int ST = 0>      // BF: 0 fp32, 1 bf16, 2 fp32 from bf16 terms
__global__ void __launch_bounds__(256, 2) wgrad_rows8_kernel(WgradArgs g) {
  static_assert(BF == 1 || ST == 0, "bf16 storage goes with bf16 MFMA operands");
  static_assert(ST == 0 || (BNQ && ST == (DVAE_ST_Y | DVAE_ST_INY)), "bf16 storage: Q and q_y of the BatchNorm-backward form");
  constexpr bool qbf = (ST & DVAE_ST_Y) != 0, qybf = (ST & DVAE_ST_INY) != 0;
  constexpr int C1 = 128, C2 = 64, Ws = 8, WQ = 16, NCB = C2 / 16;
  constexpr int PSP = C1 + 4, PSQ = C2 + 2, P4 = C1 / 4, Q4 = C2 / 4;
  constexpr int PROW = Ws * PSP, QROW = (WQ + 2) * PSQ;
  __shared__ __attribute__((aligned(16))) float pbuf[2][2][PROW];
  __shared__ __attribute__((aligned(16))) float qbuf[2][2][QROW];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i16 = lane & 15, kq = lane >> 4;
  // The four filter rows of a split read the same P rows and overlapping Q rows.  Workgroups go to the
  // eight XCDs in turn (id mod 8), so with a number of splits that is a multiple of eight the four are
  // numbered eight apart: same XCD, same L2, dispatched together.
  const int nsplit = gridDim.x >> 2;
  int dh = blockIdx.x & 3, z = blockIdx.x >> 2;
  if ((nsplit & 7) == 0) {
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
    dh = j & 3;
    z = (j >> 2) * 8 + x;
  }
  const int Hs = g.Hs;

  const int lp4 = threadIdx.x % P4, lq4 = threadIdx.x % Q4;
  f32x4 psc = (f32x4){1.f, 1.f, 1.f, 1.f}, psh = (f32x4){0.f, 0.f, 0.f, 0.f}, qsc = psc, qsh = psh;
  const float pslope = g.p_act == DVAE_ACT_LRELU ? LRELU_SLOPE : 1.f;
  const float qslope = g.q_act == DVAE_ACT_LRELU ? LRELU_SLOPE : 1.f;
  if (XP && g.p_scale) {
    psc = ldg4(g.p_scale + 4 * lp4);
    psh = ldg4(g.p_shift + 4 * lp4);
  }
  if ((XQ || BNQ) && g.q_scale) {
    qsc = ldg4(g.q_scale + 4 * lq4);
    qsh = ldg4(g.q_shift + 4 * lq4);
  }
  f32x4 qca = qsh, qcb = qsh;
  if (BNQ) {
    qca = ldg4(g.q_a + 4 * lq4);
    qcb = ldg4(g.q_b + 4 * lq4);
  }
  const bool writer = BNQ && (dh == 1 || dh == 2);
  // zero pixels left (column -1) and right (column WQ) of the four Q row buffers
  for (int i = threadIdx.x; i < 4 * 2 * PSQ; i += 256) {
    const int slot = i / (2 * PSQ), r = i % (2 * PSQ);
    (&qbuf[0][0][0])[slot * QROW + (r < PSQ ? 0 : (WQ + 1) * PSQ) + r % PSQ] = 0.f;
  }
  f32x4 acc[2][4][NCB];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int c = 0; c < NCB; ++c) acc[a][b][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // P rows are numbered over the whole batch; a pair (2p, 2p+1) never straddles two frames
  const int64_t rows = g.Mtot >> 3, pairs = rows >> 1;
  const int64_t per = (pairs + nsplit - 1) / nsplit;
  const int64_t p0 = z * per;
  const int64_t p1 = p0 + per < pairs ? p0 + per : pairs;

  auto transform = [](f32x4 v, const f32x4& sc, const f32x4& sh, float slope) {
    const f32x4 zz = v * sc + sh;         // slope <= 1: lrelu(z) = max(z, slope z)
    const f32x4 zs = zz * slope;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaxf(zz[e], zs[e]);
    return v;
  };
  // a thread moves one 16-byte piece of each of the pair's two P rows and two Q rows
  struct Raw {
    f32x4 p[2], q[2], qy[BNQ ? 2 : 1];
    bool qok[2];
    int64_t qoff[2];
  };
  auto fetch = [&](int64_t p, Raw& raw) {
    const bool pok = p < p1;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int64_t prow = 2 * p + h;
      const int a = (int)prow & (Hs - 1);
      const int hin = 2 * a - 1 + dh;
      raw.qok[h] = pok && hin >= 0 && hin < 2 * Hs;
      raw.p[h] = ldg4(g.P + (pok ? prow * (int64_t)(Ws * C1) : 0) + 4 * threadIdx.x);
      raw.qoff[h] = (raw.qok[h] ? (2 * prow - 1 + dh) * (int64_t)(WQ * C2) : 0) + 4 * threadIdx.x;
      raw.q[h] = ld4s(g.Q, raw.qoff[h], qbf);
      if (BNQ) raw.qy[h] = ld4s(g.q_y, raw.qoff[h], qybf);
    }
  };
  auto deposit = [&](int64_t p, const Raw& raw) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const f32x4 pv = XP ? transform(raw.p[h], psc, psh, pslope) : raw.p[h];
      *reinterpret_cast<f32x4*>(&pbuf[p & 1][h][(threadIdx.x / P4) * PSP + 4 * lp4]) = pv;
      f32x4 qv = XQ ? transform(raw.q[h], qsc, qsh, qslope) : widen4s(raw.q[h], qbf);
      if (BNQ) {
        const f32x4 y = widen4s(raw.qy[h], qybf);
        const f32x4 zz = y * qsc + qsh;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float dz = zz[e] > 0.f ? qv[e] : qv[e] * LRELU_SLOPE;
          qv[e] = fmaf(qsc[e], dz, fmaf(qca[e], y[e], qcb[e]));
        }
        if (writer && raw.qok[h]) *reinterpret_cast<f32x4*>(g.dx_out + raw.qoff[h]) = qv;
      }
      qv = zero_unless(qv, raw.qok[h]);
      float* d = &qbuf[p & 1][h][(1 + threadIdx.x / Q4) * PSQ + 4 * lq4];     // 8-byte aligned (PSQ is even)
      *reinterpret_cast<float2*>(d) = make_float2(qv[0], qv[1]);
      *reinterpret_cast<float2*>(d + 2) = make_float2(qv[2], qv[3]);
    }
  };
  Raw raw;
  if (p0 < p1) {
    fetch(p0, raw);
    deposit(p0, raw);
  }
  __syncthreads();
  for (int64_t p = p0; p < p1; ++p) {
    fetch(p + 1, raw);
    if constexpr (BF == 2) {
      const int h = kq >> 1, b0 = (4 * kq) & 7;
      const float* prow = pbuf[p & 1][h];
      const float* qrow = qbuf[p & 1][h];
      SplitQuad af[2];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const float* pa = prow + b0 * PSP + 16 * (2 * wave + r) + i16;
        af[r] = split_quad(pa[0], pa[PSP], pa[2 * PSP], pa[3 * PSP]);
      }
#pragma unroll
      for (int dw = 0; dw < 4; ++dw) {
        SplitQuad bs[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
          const float* qa = qrow + (2 * b0 + dw) * PSQ + 16 * cb + i16;
          bs[cb] = split_quad(qa[0], qa[2 * PSQ], qa[4 * PSQ], qa[6 * PSQ]);
        }
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int r = 0; r < 2; ++r)
              acc[r][dw][cb] = RowOps<true>::mma(af[r].t[SPLIT_TA[t]], bs[cb].t[SPLIT_TB[t]], acc[r][dw][cb]);
      }
    }
    if constexpr (BF == 1) {
      // the lane's four pixels 4kq..4kq+3 (one row of the pair: 4kq+3 < 8 or 4kq >= 8) are the four k of one
      // bf16 MFMA operand
      const int h = kq >> 1, b0 = (4 * kq) & 7;
      const float* prow = pbuf[p & 1][h];
      const float* qrow = qbuf[p & 1][h];
      s16x4 af[2];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const float* pa = prow + b0 * PSP + 16 * (2 * wave + r) + i16;
        af[r] = bf_quad(pa[0], pa[PSP], pa[2 * PSP], pa[3 * PSP]);
      }
#pragma unroll
      for (int dw = 0; dw < 4; ++dw)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
          const float* qa = qrow + (2 * b0 + dw) * PSQ + 16 * cb + i16;
          const s16x4 bfq = bf_quad(qa[0], qa[2 * PSQ], qa[4 * PSQ], qa[6 * PSQ]);
#pragma unroll
          for (int r = 0; r < 2; ++r) acc[r][dw][cb] = RowOps<true>::mma(af[r], bfq, acc[r][dw][cb]);
        }
    }
#pragma unroll
    for (int e = 0; e < (BF != 0 ? 0 : 4); ++e) {
      const int pos = 4 * kq + e, h = pos >> 3, b = pos & 7;     // this lane's pixel of the K group
      const float* prow = pbuf[p & 1][h];
      const float* qrow = qbuf[p & 1][h];
      float av[2], bv[4][NCB];
#pragma unroll
      for (int r = 0; r < 2; ++r) av[r] = prow[b * PSP + 16 * (2 * wave + r) + i16];
      // Q pixel of (b, dw) is column 2b-1+dw, stored at index 2b+dw
#pragma unroll
      for (int dw = 0; dw < 4; ++dw)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) bv[dw][cb] = qrow[(2 * b + dw) * PSQ + 16 * cb + i16];
#pragma unroll
      for (int dw = 0; dw < 4; ++dw)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
          for (int r = 0; r < 2; ++r)
            acc[r][dw][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r], bv[dw][cb], acc[r][dw][cb], 0, 0, 0);
    }
    if (p + 1 < p1) deposit(p + 1, raw);     // the other buffer: nobody reads it during this step
    __syncthreads();
  }
  // D: row = c1 16(2 wave + r) + 4kq + e, column = c2 16cb + i16 of tap 4dh + dw
  float* slab = g.slab + (int64_t)z * C1 * g.NC;
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int dw = 0; dw < 4; ++dw)
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          slab[(int64_t)(16 * (2 * wave + r) + 4 * kq + e) * g.NC + (4 * dh + dw) * C2 + 16 * cb + i16] = acc[r][dw][cb][e];
}

// And for C1 = 256, C2 = 128 on 4x4 P (sixteen times one workgroup's accumulators): a workgroup takes
// ONE filter tap (dh, dw), and a step is one whole frame -- its sixteen P pixels are exactly one K
// group, and the tap needs exactly sixteen of the frame's 8x8 Q pixels, (2a-1+dh, 2b-1+dw), gathered
// into LDS pixel by pixel (zeros where the tap leaves the frame).  Wave w owns gradient rows
// 64w .. 64w+63.  The sixteen workgroups of a split are numbered so that they land on one XCD close
// together in time and share the frame through that XCD's L2.
template <bool XP, bool XQ, int BF>      // BF: 0 fp32, 1 bf16, 2 fp32 from bf16 terms
__global__ void __launch_bounds__(256, 2) wgrad_img4_kernel(WgradArgs g) {
  constexpr int C1 = 256, C2 = 128, NCB = C2 / 16;
  constexpr int PSP = C1 + 4, PSQ = C2 + 4, P4 = C1 / 4, Q4 = C2 / 4;
  __shared__ __attribute__((aligned(16))) float pbuf[2][16 * PSP];
  __shared__ __attribute__((aligned(16))) float qbuf[2][16 * PSQ];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i16 = lane & 15, kq = lane >> 4;
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int tap = jj & 15, dh = tap >> 2, dw = tap & 3;
  const int z = (jj >> 4) * 8 + xcd, nsplit = gridDim.x >> 4;

  const int lp4 = threadIdx.x % P4, lq4 = threadIdx.x % Q4;
  f32x4 psc = (f32x4){1.f, 1.f, 1.f, 1.f}, psh = (f32x4){0.f, 0.f, 0.f, 0.f}, qsc = psc, qsh = psh;
  const float pslope = g.p_act == DVAE_ACT_LRELU ? LRELU_SLOPE : 1.f;
  const float qslope = g.q_act == DVAE_ACT_LRELU ? LRELU_SLOPE : 1.f;
  if (XP && g.p_scale) {
    psc = ldg4(g.p_scale + 4 * lp4);
    psh = ldg4(g.p_shift + 4 * lp4);
  }
  if (XQ && g.q_scale) {
    qsc = ldg4(g.q_scale + 4 * lq4);
    qsh = ldg4(g.q_shift + 4 * lq4);
  }
  f32x4 acc[4][NCB];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < NCB; ++c) acc[a][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int64_t frames = g.Mtot >> 4;
  const int64_t per = (frames + nsplit - 1) / nsplit;
  const int64_t n0 = z * per;
  const int64_t n1 = n0 + per < frames ? n0 + per : frames;

  auto transform = [](f32x4 v, const f32x4& sc, const f32x4& sh, float slope) {
    const f32x4 zz = v * sc + sh;         // slope <= 1: lrelu(z) = max(z, slope z)
    const f32x4 zs = zz * slope;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaxf(zz[e], zs[e]);
    return v;
  };
  // this thread's two Q pieces: pixel (a, b) of the 4x4 grid -> Q pixel (2a-1+dh, 2b-1+dw) of the 8x8 frame
  int qoff[2];
  bool qin[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int pi = (threadIdx.x + 256 * j) / Q4, a = pi >> 2, b = pi & 3;
    const int hi = 2 * a - 1 + dh, wi = 2 * b - 1 + dw;
    qin[j] = hi >= 0 && hi < 8 && wi >= 0 && wi < 8;
    qoff[j] = qin[j] ? (hi * 8 + wi) * C2 + 4 * lq4 : 0;
  }
  struct Raw {
    f32x4 p[4], q[2];
  };
  auto fetch = [&](int64_t n, Raw& raw) {
    const int64_t nn = n < n1 ? n : n0;
    const float* ps = g.P + nn * (16 * C1);
    const float* qs = g.Q + nn * (64 * C2);
#pragma unroll
    for (int j = 0; j < 4; ++j) raw.p[j] = ldg4(ps + 4 * (threadIdx.x + 256 * j));
#pragma unroll
    for (int j = 0; j < 2; ++j) raw.q[j] = ldg4(qs + qoff[j]);
  };
  auto deposit = [&](int64_t n, const Raw& raw) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 pv = XP ? transform(raw.p[j], psc, psh, pslope) : raw.p[j];
      *reinterpret_cast<f32x4*>(&pbuf[n & 1][((threadIdx.x + 256 * j) / P4) * PSP + 4 * lp4]) = pv;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      f32x4 qv = XQ ? transform(raw.q[j], qsc, qsh, qslope) : raw.q[j];
      qv = zero_unless(qv, qin[j]);
      *reinterpret_cast<f32x4*>(&qbuf[n & 1][((threadIdx.x + 256 * j) / Q4) * PSQ + 4 * lq4]) = qv;
    }
  };
  Raw raw;
  if (n0 < n1) {
    fetch(n0, raw);
    deposit(n0, raw);
  }
  __syncthreads();
  for (int64_t n = n0; n < n1; ++n) {
    fetch(n + 1, raw);
    const float* pp = pbuf[n & 1];
    const float* qq = qbuf[n & 1];
    if constexpr (BF == 2) {
      SplitQuad af[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float* pa = pp + 4 * kq * PSP + 16 * (4 * wave + r) + i16;
        af[r] = split_quad(pa[0], pa[PSP], pa[2 * PSP], pa[3 * PSP]);
      }
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        const float* qa = qq + 4 * kq * PSQ + 16 * cb + i16;
        const SplitQuad bs = split_quad(qa[0], qa[PSQ], qa[2 * PSQ], qa[3 * PSQ]);
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[r][cb] = RowOps<true>::mma(af[r].t[SPLIT_TA[t]], bs.t[SPLIT_TB[t]], acc[r][cb]);
      }
    }
    if constexpr (BF == 1) {
      // the lane's four pixels 4kq..4kq+3 are the four k of one bf16 MFMA operand
      s16x4 af[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float* pa = pp + 4 * kq * PSP + 16 * (4 * wave + r) + i16;
        af[r] = bf_quad(pa[0], pa[PSP], pa[2 * PSP], pa[3 * PSP]);
      }
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        const float* qa = qq + 4 * kq * PSQ + 16 * cb + i16;
        const s16x4 bfq = bf_quad(qa[0], qa[PSQ], qa[2 * PSQ], qa[3 * PSQ]);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r][cb] = RowOps<true>::mma(af[r], bfq, acc[r][cb]);
      }
    }
#pragma unroll
    for (int e = 0; e < (BF != 0 ? 0 : 4); ++e) {
      const int pos = 4 * kq + e;                 // this lane's pixel of the K group
      float av[4], bv[NCB];
#pragma unroll
      for (int r = 0; r < 4; ++r) av[r] = pp[pos * PSP + 16 * (4 * wave + r) + i16];
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) bv[cb] = qq[pos * PSQ + 16 * cb + i16];
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          acc[r][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r], bv[cb], acc[r][cb], 0, 0, 0);
    }
    if (n + 1 < n1) deposit(n + 1, raw);      // the other buffer: nobody reads it during this step
    __syncthreads();
  }
  // D: row = c1 16(4 wave + r) + 4kq + e, column = c2 16cb + i16 of this workgroup's tap
  float* slab = g.slab + (int64_t)z * C1 * g.NC;
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        slab[(int64_t)(16 * (4 * wave + r) + 4 * kq + e) * g.NC + tap * C2 + 16 * cb + i16] = acc[r][cb][e];
}

struct WgradPlan {
  int bm;       // 32, 64 or 128; 0 = single-channel streaming kernel; -1 / -2 / -3 = row-ring kernels
  int64_t tm, tn;
  int split, ktiles_per_split;
};

static bool wgrad_rows_shape(int64_t C1, int64_t C2, int64_t Ws) {
  return ((C1 == 64 && C2 == 32) || (C1 == 32 && C2 == 16)) && (Ws == 16 || Ws == 32);
}

static WgradPlan wgrad_plan(int64_t Mtot, int64_t Ws, int64_t C1, int64_t C2, int64_t Hs = 0) {
  WgradPlan p;
  if (C1 == 256 && C2 == 128 && Ws == 4 && Hs == 4 && Mtot >= 16 * 32) {
    // sixteen workgroups (filter taps) per slab; 16 x 32 = two workgroups per compute unit
    p.bm = -3;
    p.tm = p.tn = 1;
    p.split = 32;
    p.ktiles_per_split = 0;
    return p;
  }
  if (C2 == 1 && C1 <= 16 * WG1_MAX_CB) {
    p.bm = 0;
    p.tm = p.tn = 1;
    int64_t blocks = ceil_div64(Mtot, 4 * 4 * 16);      // >= 16 four-pixel blocks per wave
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    p.split = (int)blocks;
    p.ktiles_per_split = 0;
    return p;
  }
  if (C1 == 128 && C2 == 64 && Ws == 8 && Mtot >= 16) {
    // four workgroups (filter rows) per slab; 4 x 128 = two workgroups per compute unit
    p.bm = -2;
    p.tm = p.tn = 1;
    const int64_t pairs = Mtot / 16;
    p.split = (int)(pairs < 128 ? pairs : 128);
    p.ktiles_per_split = 0;
    return p;
  }
  if (wgrad_rows_shape(C1, C2, Ws)) {
    // one slab per workgroup; as many workgroups as are resident at once
    p.bm = -1;
    p.tm = p.tn = 1;
    const int64_t rows = Mtot / Ws, resident = 256 * (C1 == 64 ? 2 : 4);
    p.split = (int)(rows < resident ? rows : resident);
    p.ktiles_per_split = 0;
    return p;
  }
  const int64_t NC = 16 * C2;
  p.bm = C1 <= 32 ? 32 : (C1 <= 64 ? 64 : 128);
  p.tm = ceil_div64(C1, p.bm);
  p.tn = ceil_div64(NC, 128);
  const int64_t ktiles = ceil_div64(Mtot, 32);
  int64_t s = 768 / (p.tm * p.tn);
  if (s > ktiles / 8) s = ktiles / 8;
  if (s > 256) s = 256;
  if (s < 1) s = 1;
  p.ktiles_per_split = (int)ceil_div64(ktiles, s);
  p.split = (int)ceil_div64(ktiles, p.ktiles_per_split);
  return p;
}

extern "C" int64_t dvae_conv_wgrad_ws_floats(int64_t N, int64_t Hs, int64_t Ws, int64_t C1, int64_t C2) {
  const WgradPlan p = wgrad_plan(N * Hs * Ws, Ws, C1, C2, Hs);
  // the slabs, then the partial sums of a two-stage reduction
  const int runs = wgrad_reduce_runs(p.split, C1 * C2 * 16);
  return ((int64_t)p.split + (runs > 1 ? runs : 0)) * C1 * C2 * 16;
}

static int conv4s2_wgrad_impl(const float* P, const float* p_scale, const float* p_shift, int p_act,
                              const float* Q, const float* q_scale, const float* q_shift, int q_act,
                              float* dw, float* ws, int64_t N, int64_t Hs, int64_t Ws, int64_t C1, int64_t C2,
                              const float* p_y, const float* p_a, const float* p_b, dvae_stream_t stream,
                              const float* q_y = nullptr, const float* q_a = nullptr, const float* q_b = nullptr,
                              float* dx_out = nullptr, int storage = 0) {
  DVAE_REQUIRE(N > 0 && C1 > 0 && C2 > 0, "conv4s2_wgrad: empty");
  DVAE_REQUIRE(is_pow2(Hs) && is_pow2(Ws), "conv4s2_wgrad: Hs and Ws must be powers of two");
  DVAE_REQUIRE(C1 % 4 == 0, "conv4s2_wgrad: C1 must be a multiple of 4 (C1=%ld)", (long)C1);
  DVAE_REQUIRE(C2 == 1 || C2 % 4 == 0, "conv4s2_wgrad: C2 must be 1 or a multiple of 4 (C2=%ld)", (long)C2);
  DVAE_REQUIRE(!(C2 == 1 && (q_scale != nullptr || q_act != DVAE_ACT_NONE)),
               "conv4s2_wgrad: no load-time transform on a single-channel Q");
  DVAE_REQUIRE(!(C2 == 1 && C1 > 16 * WG1_MAX_CB), "conv4s2_wgrad: single-channel Q supports C1 <= %d", 16 * WG1_MAX_CB);
  DVAE_REQUIRE(aligned16(P) && aligned16(Q) && ws != nullptr, "conv4s2_wgrad: P, Q must be 16-byte aligned, ws non-null");
  DVAE_REQUIRE(N * Hs * Ws < (1LL << 30), "conv4s2_wgrad: more than 2^30 pixels");
  WgradArgs g;
  g.P = P; g.p_scale = p_scale; g.p_shift = p_shift; g.p_act = p_act;
  g.Q = Q; g.q_scale = q_scale; g.q_shift = q_shift; g.q_act = q_act;
  g.slab = ws; g.Mtot = N * Hs * Ws; g.Hs = (int)Hs; g.Ws = (int)Ws; g.lh = ilog2(Hs); g.lw = ilog2(Ws);
  g.C1 = (int)C1; g.C2 = (int)C2; g.NC = (int)(16 * C2);
  g.p_y = p_y; g.p_a = p_a; g.p_b = p_b;
  g.q_y = q_y; g.q_a = q_a; g.q_b = q_b; g.dx_out = dx_out; g.st = storage;
  DVAE_REQUIRE(p_y == nullptr || (C2 == 1 && p_scale && p_shift && p_a && p_b),
               "conv4s2_wgrad: a BatchNorm-backward P needs a single-channel Q and scale, shift, A, B");
  const WgradPlan p = wgrad_plan(g.Mtot, Ws, C1, C2, Hs);
  g.ktiles_per_split = p.ktiles_per_split;
  hipStream_t s = ST(stream);
  const bool wbf = dvae_mfma_precision() == DVAE_PRECISION_BF16;
  // 'fp32_split' in the three row kernels whose operands meet in LDS: correct, and slower than their fp32 MFMA form
  // on this part (0.87 / 0.83 / 0.71 ms against 0.69 / 0.77 / 0.73 at batch 256): the lanes hold four k per
  // operand, which is v_mfma_f32_16x16x16_bf16 -- half the bf16 rate of gfx950's 16x16x32 form -- so six of them
  // save a quarter of the matrix time of four fp32 MFMAs, and the two dozen vector instructions that split a quad
  // cost more than that.  Only on request (DVAE_SPLIT_ROW_WGRAD=1); by default these kernels stay on the fp32 MFMA
  // in that mode.
  const bool wsp = dvae_mfma_precision() == DVAE_PRECISION_F32_SPLIT && [] {
    const char* v = getenv("DVAE_SPLIT_ROW_WGRAD");
    return v != nullptr && v[0] == '1';
  }();
  DVAE_REQUIRE(storage == 0 || (wbf && ((p.bm == -1 && (storage == DVAE_ST_Y || storage == (DVAE_ST_X | DVAE_ST_Y))) ||
                                        (p.bm == -2 && storage == (DVAE_ST_Y | DVAE_ST_INY) && g.q_y != nullptr) ||
                                        (p.bm == 0 && storage == DVAE_ST_X && C1 == 32 && g.p_y == nullptr) ||
                                        (p.bm == 0 && storage == (DVAE_ST_X | DVAE_ST_BNY) && C1 == 16 && g.p_y != nullptr))),
               "conv4s2_wgrad: bf16 storage needs the bf16 MFMA mode and is only taken by the row kernels (Q) and the "
               "single-channel-Q kernel (P with 32 channels; P and p_y with 16)");
  // a row-ring kernel in the variant of the current MFMA precision
#define WG_BF(NAME, LDS, ...)                                                                           \
  {                                                                                                     \
    if (wbf) DVAE_LAUNCH((NAME<__VA_ARGS__, 1>), grid, dim3(256), LDS, s, g);                           \
    else if (wsp) DVAE_LAUNCH((NAME<__VA_ARGS__, 2>), grid, dim3(256), LDS, s, g);                      \
    else DVAE_LAUNCH((NAME<__VA_ARGS__, 0>), grid, dim3(256), LDS, s, g);                               \
  }
  DVAE_REQUIRE(q_y == nullptr || (p.bm == -2 && q_scale && q_shift && q_a && q_b && dx_out && aligned16(q_y) &&
                                  aligned16(q_a) && aligned16(q_b) && aligned16(dx_out)),
               "conv4s2_wgrad: a BatchNorm-backward Q is only taken where dvae_conv4s2_wgrad_bnq_supported says so, "
               "with y, scale, shift, A, B and dx_out, 16-byte aligned");
  if (C2 == 1) DVAE_WORK(((storage & DVAE_ST_X) ? 2.0 : 4.0) * (double)g.Mtot * C1 * (g.p_y ? 2 : 1) + 4.0 * (double)g.Mtot * 4, DVAE_UNIT_BYTE);       // streams P (and Q) once
  else DVAE_WORK(2.0 * (double)g.Mtot * C1 * 16 * C2, DVAE_UNIT_FLOP);
  if (p.bm == -3) {
    DVAE_REQUIRE((!p_scale || (aligned16(p_scale) && aligned16(p_shift))) && (!q_scale || (aligned16(q_scale) && aligned16(q_shift))),
                 "conv4s2_wgrad: scale/shift vectors must be 16-byte aligned");
    const dim3 grid((unsigned)p.split * 16);
    const bool xp = p_scale != nullptr || p_act != DVAE_ACT_NONE, xq = q_scale != nullptr || q_act != DVAE_ACT_NONE;
    if (xp && xq) WG_BF(wgrad_img4_kernel, 0, true, true)
    else if (xp) WG_BF(wgrad_img4_kernel, 0, true, false)
    else if (xq) WG_BF(wgrad_img4_kernel, 0, false, true)
    else WG_BF(wgrad_img4_kernel, 0, false, false)
  } else if (p.bm == -2) {
    DVAE_REQUIRE(Hs >= 2, "conv4s2_wgrad: Hs must be at least 2 at this shape");
    DVAE_REQUIRE((!p_scale || (aligned16(p_scale) && aligned16(p_shift))) && (!q_scale || (aligned16(q_scale) && aligned16(q_shift))),
                 "conv4s2_wgrad: scale/shift vectors must be 16-byte aligned");
    const dim3 grid((unsigned)p.split * 4);
    const bool xp = p_scale != nullptr || p_act != DVAE_ACT_NONE, xq = q_scale != nullptr || q_act != DVAE_ACT_NONE;
    if (g.q_y != nullptr && g.st != 0) {           // (bf16 mode, Q and q_y: checked above)
      if (xp) DVAE_LAUNCH((wgrad_rows8_kernel<true, false, true, true, DVAE_ST_Y | DVAE_ST_INY>), grid, dim3(256), 0, s, g);
      else DVAE_LAUNCH((wgrad_rows8_kernel<false, false, true, true, DVAE_ST_Y | DVAE_ST_INY>), grid, dim3(256), 0, s, g);
    } else if (g.q_y != nullptr) {
      if (xp) WG_BF(wgrad_rows8_kernel, 0, true, false, true)
      else WG_BF(wgrad_rows8_kernel, 0, false, false, true)
    } else if (xp && xq) WG_BF(wgrad_rows8_kernel, 0, true, true, false)
    else if (xp) WG_BF(wgrad_rows8_kernel, 0, true, false, false)
    else if (xq) WG_BF(wgrad_rows8_kernel, 0, false, true, false)
    else WG_BF(wgrad_rows8_kernel, 0, false, false, false)
  } else if (p.bm == -1) {
    DVAE_REQUIRE((!p_scale || (aligned16(p_scale) && aligned16(p_shift))) && (!q_scale || (aligned16(q_scale) && aligned16(q_shift))),
                 "conv4s2_wgrad: scale/shift vectors must be 16-byte aligned");
    const dim3 grid((unsigned)p.split);
    const bool xp = p_scale != nullptr || p_act != DVAE_ACT_NONE, xq = q_scale != nullptr || q_act != DVAE_ACT_NONE;
#define WGR_LAUNCH(A, B)                                                                                  \
  {                                                                                                       \
    const size_t ldsb = (size_t)(4 * Ws * (A + 4) + 8 * (2 * Ws + 2) * (B + 2)) * sizeof(float);          \
    if (g.st != 0) {                                                                                      \
      if (A == 64 && wbf && xp && !xq && g.st == DVAE_ST_Y)                                               \
        DVAE_LAUNCH((wgrad_rows_kernel<64, 32, 2, true, false, true, DVAE_ST_Y>), grid, dim3(256), ldsb, s, g); \
      else if (A == 64 && wbf && xp && !xq && g.st == (DVAE_ST_X | DVAE_ST_Y))                            \
        DVAE_LAUNCH((wgrad_rows_kernel<64, 32, 2, true, false, true, DVAE_ST_X | DVAE_ST_Y>), grid, dim3(256), ldsb, s, g); \
      else if (A == 32 && wbf && !xp && xq && g.st == DVAE_ST_Y)                                          \
        DVAE_LAUNCH((wgrad_rows_kernel<32, 16, 2, false, true, true, DVAE_ST_Y>), grid, dim3(256), ldsb, s, g); \
      else {                                                                                              \
        dvae_set_error("conv4s2_wgrad: the row kernels take a bf16 Q in bf16 mode (64 x 32 with a transformed P, " \
                       "32 x 16 with a transformed Q)");                                                  \
        return -EINVAL;                                                                                   \
      }                                                                                                   \
    } else if (xp && xq) WG_BF(wgrad_rows_kernel, ldsb, A, B, 2, true, true) \
    else if (xp) WG_BF(wgrad_rows_kernel, ldsb, A, B, 2, true, false) \
    else if (xq) WG_BF(wgrad_rows_kernel, ldsb, A, B, 2, false, true) \
    else WG_BF(wgrad_rows_kernel, ldsb, A, B, 2, false, false) \
  }
    if (C1 == 64) WGR_LAUNCH(64, 32) else WGR_LAUNCH(32, 16)
#undef WGR_LAUNCH
#undef WG_BF
  } else if (p.bm == 0) {
    const int ncb = (int)((C1 + 15) / 16);
    const dim3 grid((unsigned)p.split);
    if (g.p_y != nullptr && g.st != 0)             // (C1 == 16, P and p_y bf16: checked above)
      DVAE_LAUNCH((conv_wgrad_c1_kernel<1, 8, true, DVAE_ST_X | DVAE_ST_BNY>), grid, dim3(256), 0, s, g);
    else if (g.p_y != nullptr) {
      if (ncb <= 1) DVAE_LAUNCH((conv_wgrad_c1_kernel<1, 8, true>), grid, dim3(256), 0, s, g);
      else if (ncb <= 2) DVAE_LAUNCH((conv_wgrad_c1_kernel<2, 4, true>), grid, dim3(256), 0, s, g);
      else if (ncb <= 4) DVAE_LAUNCH((conv_wgrad_c1_kernel<4, 2, true>), grid, dim3(256), 0, s, g);
      else DVAE_LAUNCH((conv_wgrad_c1_kernel<8, 1, true>), grid, dim3(256), 0, s, g);
    } else if (g.st != 0)
      DVAE_LAUNCH((conv_wgrad_c1_kernel<2, 8, false, DVAE_ST_X>), grid, dim3(256), 0, s, g);      // (C1 == 32, checked above)
    else if (ncb <= 1) DVAE_LAUNCH((conv_wgrad_c1_kernel<1, 8, false>), grid, dim3(256), 0, s, g);
    else if (ncb <= 2) DVAE_LAUNCH((conv_wgrad_c1_kernel<2, 8, false>), grid, dim3(256), 0, s, g);
    else if (ncb <= 4) DVAE_LAUNCH((conv_wgrad_c1_kernel<4, 4, false>), grid, dim3(256), 0, s, g);
    else DVAE_LAUNCH((conv_wgrad_c1_kernel<8, 2, false>), grid, dim3(256), 0, s, g);
  } else {
    const dim3 grid((unsigned)p.tn, (unsigned)p.tm, (unsigned)p.split);
    const bool bf = dvae_mfma_precision() == DVAE_PRECISION_BF16;
#define WG_LAUNCH(TILE)                                                                     \
  if (bf) DVAE_LAUNCH((conv_wgrad_kernel<TILE, PREC_BF16>), grid, dim3(256), 0, s, g);     \
  else if (dvae_mfma_precision() == DVAE_PRECISION_F32_SPLIT)                              \
    DVAE_LAUNCH((conv_wgrad_kernel<TILE, PREC_SPLIT>), grid, dim3(256), 0, s, g);          \
  else DVAE_LAUNCH((conv_wgrad_kernel<TILE, PREC_F32>), grid, dim3(256), 0, s, g);
    typedef Tile<32, 128, 1, 4, false, false> W32;
    typedef Tile<64, 128, 2, 2, false, false> W64;
    typedef Tile<128, 128, 2, 2, false, false> W128;
    if (p.bm == 32) {
      WG_LAUNCH(W32)
    } else if (p.bm == 64) {
      WG_LAUNCH(W64)
    } else {
      WG_LAUNCH(W128)
    }
#undef WG_LAUNCH
  }
  DVAE_LAUNCH_CHECK("conv4s2_wgrad");
  int64_t nb = ceil_div64(C1 * C2 * 16, 256);
  if (nb > 2048) nb = 2048;
  const int runs = wgrad_reduce_runs(p.split, C1 * C2 * 16);
  if (runs > 1) {
    float* part = ws + (int64_t)p.split * C1 * C2 * 16;
    const int zchunk = (p.split + runs - 1) / runs, nruns = (p.split + zchunk - 1) / zchunk;
    DVAE_LAUNCH(wgrad_reduce_kernel, dim3((unsigned)nb, (unsigned)nruns), dim3(256), 0, s, (const float*)ws,
                       p.split, zchunk, part, dw, C1, C2);
    DVAE_LAUNCH(wgrad_reduce_kernel, dim3((unsigned)nb, 1), dim3(256), 0, s, (const float*)part, nruns, nruns,
                       part, dw, C1, C2);
  } else {
    DVAE_LAUNCH(wgrad_reduce_kernel, dim3((unsigned)nb, 1), dim3(256), 0, s, (const float*)ws, p.split, p.split,
                       ws, dw, C1, C2);
  }
  DVAE_LAUNCH_CHECK("conv4s2_wgrad_reduce");
  return 0;
}

extern "C" int dvae_conv4s2_wgrad_f32(const float* P, const float* p_scale, const float* p_shift, int p_act,
                                      const float* Q, const float* q_scale, const float* q_shift, int q_act,
                                      float* dw, float* ws, int64_t N, int64_t Hs, int64_t Ws, int64_t C1, int64_t C2,
                                      int storage, dvae_stream_t stream) {
  return conv4s2_wgrad_impl(P, p_scale, p_shift, p_act, Q, q_scale, q_shift, q_act, dw, ws, N, Hs, Ws, C1, C2, nullptr,
                            nullptr, nullptr, stream, nullptr, nullptr, nullptr, nullptr, storage);
}

extern "C" int dvae_conv4s2_wgrad_bnq_supported(int64_t N, int64_t Hs, int64_t Ws, int64_t C1, int64_t C2) {
  return wgrad_plan(N * Hs * Ws, Ws, C1, C2, Hs).bm == -2;
}

extern "C" int dvae_conv4s2_wgrad_bnq_f32(const float* P, const float* p_scale, const float* p_shift, int p_act,
                                          const float* du, const float* q_y, const float* q_scale,
                                          const float* q_shift, const float* q_a, const float* q_b, float* dx_out,
                                          float* dw, float* ws, int64_t N, int64_t Hs, int64_t Ws, int64_t C1,
                                          int64_t C2, int storage, dvae_stream_t stream) {
  DVAE_REQUIRE(q_y != nullptr, "conv4s2_wgrad_bnq: q_y is null");
  return conv4s2_wgrad_impl(P, p_scale, p_shift, p_act, du, q_scale, q_shift, DVAE_ACT_NONE, dw, ws, N, Hs, Ws, C1, C2,
                            nullptr, nullptr, nullptr, stream, q_y, q_a, q_b, dx_out, storage);
}

extern "C" int dvae_conv4s2_wgrad_bnp_f32(const float* du, const float* p_y, const float* p_scale, const float* p_shift,
                                          const float* p_a, const float* p_b, const float* Q, float* dw, float* ws,
                                          int64_t N, int64_t Hs, int64_t Ws, int64_t C1, int storage,
                                          dvae_stream_t stream) {
  DVAE_REQUIRE(p_y != nullptr, "conv4s2_wgrad_bnp: p_y is null");
  return conv4s2_wgrad_impl(du, p_scale, p_shift, DVAE_ACT_NONE, Q, nullptr, nullptr, DVAE_ACT_NONE, dw, ws, N, Hs, Ws,
                            C1, 1, p_y, p_a, p_b, stream, nullptr, nullptr, nullptr, nullptr, storage);
}

// ------------------------------------------------------------------------------------------------
// final decoder layer: ConvTranspose2d(Cin -> 1) + bias + sigmoid.
//
// Written in scatter form, every input pixel contributes x[pixel][:] . w[:][kh][kw] to the sixteen
// output positions (2a-1+kh, 2b-1+kw): contributions[16 taps][pixels] = W^T[16 x Cin] * X^T[Cin x
// pixels], a GEMM with exactly sixteen rows, which v_mfma_f32_16x16x4_f32 computes for sixteen pixels
// at a time while x is read once, 16 B per lane, straight into the B operand.  A workgroup owns a
// band of R input rows of one frame (plus one halo row on each side), parks the sixteen tap products
// of every pixel in LDS, then each output position sums the four that reach it in a fixed order
// (bit-reproducible, no atomics), applies bias + sigmoid and the 2R output rows are written out
// contiguously.  Bound by the read of x.
// ------------------------------------------------------------------------------------------------
#define DF_MAX_CB 16   // Cin <= 256
struct DeconvFinalArgs {
  const float* x;
  const float* in_scale;
  const float* in_shift;
  const float* w;
  const float* bias;
  float* y;
  int in_act, out_act;
  int H, W, lw, Cin, R, bands;
};

__global__ void __launch_bounds__(256) deconv_final_kernel(DeconvFinalArgs g) {
  extern __shared__ __attribute__((aligned(16))) float contrib[];   // [(R+2)*W pixels][16 taps]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j16 = lane & 15, q = lane >> 4;
  const int W = g.W, H = g.H, WO = 2 * W;
  const int64_t n = blockIdx.x / g.bands;
  const int a0 = (int)(blockIdx.x % g.bands) * g.R;
  const int ncb = (g.Cin + 15) / 16;
  // A operand: w[ci][tap] for tap = lane&15 and the four channels 4q..4q+3 of every 16-channel block
  float wa[DF_MAX_CB][4];
#pragma unroll
  for (int cb = 0; cb < DF_MAX_CB; ++cb)
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) {
      const int ci = cb * 16 + 4 * q + s2;
      wa[cb][s2] = (cb < ncb && ci < g.Cin) ? g.w[ci * 16 + j16] : 0.f;
    }
  __syncthreads();
  const int rows_in = g.R + 2;                   // halo row above and below
  const int nblk = (rows_in * W + 15) / 16;
  for (int blk = wave; blk < nblk; blk += 4) {
    const int p = blk * 16 + j16;                // pixel of this lane inside the band (with halo)
    const int ar = p >> g.lw, b = p & (W - 1);
    const int a = a0 - 1 + ar;
    const bool ok = ar < rows_in && a >= 0 && a < H;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* px = g.x + ((n * H + (ok ? a : 0)) * W + b) * g.Cin;
#pragma unroll
    for (int cb = 0; cb < DF_MAX_CB; ++cb) {
      if (cb < ncb) {
        const int ci = cb * 16 + 4 * q;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok && ci < g.Cin) {
          v = *reinterpret_cast<const float4*>(px + ci);
          v = affine_act4(v, g.in_scale, g.in_shift, ci, g.in_act);
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[cb][0], v.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[cb][1], v.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[cb][2], v.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[cb][3], v.w, acc, 0, 0, 0);
      }
    }
    // D: column = pixel j16, row = tap = 4q + r  ->  this lane holds kh = q, kw = 0..3 of its pixel
    if (ar < rows_in)
      *reinterpret_cast<float4*>(contrib + (int64_t)p * 16 + 4 * q) =
          ok ? make_float4(acc[0], acc[1], acc[2], acc[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
  // gather: output (ho, wo) = sum over the 2x2 input pixels that reach it, in a fixed order
  const float bv = g.bias ? g.bias[0] : 0.f;
  float* yo = g.y + (n * 2 * H + 2 * a0) * WO;
  for (int i = threadIdx.x; i < 2 * g.R * WO; i += 256) {
    const int orow = i / WO, wo = i - orow * WO;
    const int ho = 2 * a0 + orow;
    // rows: ho even -> (a = ho/2, kh = 1), (a = ho/2 - 1, kh = 3); ho odd -> (a = (ho+1)/2, kh = 0), (a = (ho-1)/2, kh = 2)
    const int ah = (ho + 1) >> 1, khh = (ho & 1) ? 0 : 1;
    const int bh = (wo + 1) >> 1, kwh = (wo & 1) ? 0 : 1;
    float v = bv;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
      const int a = ah - dy, kh = khh + 2 * dy;
      if (a < 0 || a >= H) continue;
      const int ar = a - (a0 - 1);
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int b = bh - dx, kw = kwh + 2 * dx;
        if (b < 0 || b >= W) continue;
        v += contrib[(ar * W + b) * 16 + kh * 4 + kw];
      }
    }
    if (g.out_act == DVAE_ACT_SIGMOID) v = 1.f / (1.f + expf(-v));
    yo[i] = v;
  }
}

// The same kernel for a channel count known at compile time (Cin = 16 NCB): the tap weights take 4 NCB
// registers instead of 64, and a wave keeps the loads of U pixel blocks in flight at once.
template <int NCB, int U, bool XB = false>
__global__ void __launch_bounds__(256) deconv_final_fixed_kernel(DeconvFinalArgs g) {
  extern __shared__ __attribute__((aligned(16))) float contrib[];   // [(R+2)*W pixels][16 taps]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j16 = lane & 15, q = lane >> 4;
  const int W = g.W, H = g.H, WO = 2 * W;
  const int64_t n = blockIdx.x / g.bands;
  const int a0 = (int)(blockIdx.x % g.bands) * g.R;
  float wa[NCB][4];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) wa[cb][s2] = g.w[(cb * 16 + 4 * q + s2) * 16 + j16];
  float4 sc[NCB], sh[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
    sc[cb] = g.in_scale ? *reinterpret_cast<const float4*>(g.in_scale + cb * 16 + 4 * q) : make_float4(1.f, 1.f, 1.f, 1.f);
    sh[cb] = g.in_scale ? *reinterpret_cast<const float4*>(g.in_shift + cb * 16 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int rows_in = g.R + 2;                   // halo row above and below
  const int nblk = (rows_in * W + 15) / 16;
  for (int blk0 = wave; blk0 < nblk; blk0 += 4 * U) {
    f32x4 v[U][NCB];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int p = (blk0 + 4 * u) * 16 + j16;
      const int ar = p >> g.lw, b = p & (W - 1);
      const int a = a0 - 1 + ar;
      ok[u] = ar < rows_in && a >= 0 && a < H;
      const int64_t px = ((n * H + (ok[u] ? a : 0)) * W + b) * (NCB * 16) + 4 * q;
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) v[u][cb] = ld4s(g.x, px + cb * 16, XB);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int p = (blk0 + 4 * u) * 16 + j16;
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        const f32x4 t4 = widen4s(v[u][cb], XB);
        float4 t = make_float4(t4[0], t4[1], t4[2], t4[3]);
        if (g.in_scale) {
          t.x = fmaf(t.x, sc[cb].x, sh[cb].x); t.y = fmaf(t.y, sc[cb].y, sh[cb].y);
          t.z = fmaf(t.z, sc[cb].z, sh[cb].z); t.w = fmaf(t.w, sc[cb].w, sh[cb].w);
        }
        if (g.in_act == DVAE_ACT_LRELU) { t.x = lrelu(t.x); t.y = lrelu(t.y); t.z = lrelu(t.z); t.w = lrelu(t.w); }
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[cb][0], t.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[cb][1], t.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[cb][2], t.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[cb][3], t.w, acc, 0, 0, 0);
      }
      if ((p >> g.lw) < rows_in)
        *reinterpret_cast<float4*>(contrib + (int64_t)p * 16 + 4 * q) =
            ok[u] ? make_float4(acc[0], acc[1], acc[2], acc[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __syncthreads();
  // gather: output (ho, wo) = sum over the 2x2 input pixels that reach it, in a fixed order
  const float bv = g.bias ? g.bias[0] : 0.f;
  float* yo = g.y + (n * 2 * H + 2 * a0) * WO;
  for (int i = threadIdx.x; i < 2 * g.R * WO; i += 256) {
    const int orow = i / WO, wo = i - orow * WO;
    const int ho = 2 * a0 + orow;
    const int ah = (ho + 1) >> 1, khh = (ho & 1) ? 0 : 1;
    const int bh = (wo + 1) >> 1, kwh = (wo & 1) ? 0 : 1;
    float v = bv;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
      const int a = ah - dy, kh = khh + 2 * dy;
      if (a < 0 || a >= H) continue;
      const int ar = a - (a0 - 1);
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int b = bh - dx, kw = kwh + 2 * dx;
        if (b < 0 || b >= W) continue;
        v += contrib[(ar * W + b) * 16 + kh * 4 + kw];
      }
    }
    if (g.out_act == DVAE_ACT_SIGMOID) v = 1.f / (1.f + expf(-v));
    yo[i] = v;
  }
}

extern "C" int dvae_deconv4s2_final_fwd_f32(const float* x, const float* in_scale, const float* in_shift, int in_act,
                                            const float* w, const float* bias, int out_act, float* y, int64_t N,
                                            int64_t H, int64_t W, int64_t Cin, int storage, dvae_stream_t stream) {
  DVAE_REQUIRE(N > 0, "deconv4s2_final_fwd: empty");
  DVAE_REQUIRE(storage == 0 || (storage == DVAE_ST_X && Cin == 32 &&
                                (in_scale == nullptr || (aligned16(in_scale) && aligned16(in_shift)))),
               "deconv4s2_final_fwd: bf16 storage is taken for the input, with 32 channels");
  DVAE_REQUIRE(is_pow2(H) && is_pow2(W), "deconv4s2_final_fwd: H and W must be powers of two");
  DVAE_REQUIRE(Cin % 4 == 0 && Cin <= 16 * DF_MAX_CB, "deconv4s2_final_fwd: Cin must be a multiple of 4, at most %d (Cin=%ld)",
               16 * DF_MAX_CB, (long)Cin);
  DVAE_REQUIRE(aligned16(x), "deconv4s2_final_fwd: x must be 16-byte aligned");
  DVAE_REQUIRE(W <= 128, "deconv4s2_final_fwd: W too large for the LDS band image");
  DeconvFinalArgs g;
  g.x = x; g.in_scale = in_scale; g.in_shift = in_shift; g.w = w; g.bias = bias; g.y = y;
  g.in_act = in_act; g.out_act = out_act; g.H = (int)H; g.W = (int)W; g.lw = ilog2(W); g.Cin = (int)Cin;
  g.R = H >= 16 ? 16 : (int)H;     // rows per band: the two halo rows are read twice, so not too few
  g.bands = (int)(H / g.R);
  const size_t lds = (size_t)(g.R + 2) * W * 16 * 4;
  const bool vec_affine = in_scale == nullptr || (aligned16(in_scale) && aligned16(in_shift));
  DVAE_WORK((storage ? 2.0 : 4.0) * (double)N * H * W * Cin + 4.0 * (double)N * 4 * H * W, DVAE_UNIT_BYTE);   // read x, write y
  if (Cin == 32 && vec_affine && storage)
    DVAE_LAUNCH((deconv_final_fixed_kernel<2, 4, true>), dim3((unsigned)(N * g.bands)), dim3(256), lds, ST(stream), g);
  else if (Cin == 32 && vec_affine)
    DVAE_LAUNCH((deconv_final_fixed_kernel<2, 4>), dim3((unsigned)(N * g.bands)), dim3(256), lds, ST(stream), g);
  else if (Cin == 64 && vec_affine)
    DVAE_LAUNCH((deconv_final_fixed_kernel<4, 2>), dim3((unsigned)(N * g.bands)), dim3(256), lds, ST(stream), g);
  else
    DVAE_LAUNCH(deconv_final_kernel, dim3((unsigned)(N * g.bands)), dim3(256), lds, ST(stream), g);
  DVAE_LAUNCH_CHECK("deconv4s2_final_fwd");
  return 0;
}
