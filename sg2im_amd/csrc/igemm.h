// Implicit-GEMM engine on the gfx950 fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
// One 256-thread workgroup (4 wavefronts of 64, arranged 2x2) owns a BM x BN output
// tile.  The reduction runs in chunks of BK = 32; operand tiles are staged through
// LDS (one buffer per operand; the next chunk is prefetched into registers while the
// current one is on the matrix pipe, and stored between two barriers).  Two LDS layouts:
//   * "m-major"  [rows][BK+4]  - the reduction index is contiguous in global memory
//                                (activations NHWC along channels, weight rows);
//                                fragments are fetched with ds_read_b128.
//   * "k-major"  [BK][rows+4]  - the reduction index is the global row (weight-gradient
//                                and data-gradient operands); fragments via ds_read_b32.
// The fp32 MFMA consumes one A and one B scalar per lane per K=2 step:
//   lane l: A[i = l&31][k = l>>5],  B[k = l>>5][j = l&31]   (cdna_hip_programming.md section 3)
// Inside a BK chunk the k order is permuted (kperm) so an m-major lane reads four
// consecutive k with one 16-byte LDS read; both operands use the same permutation,
// which only reorders the (commutative up to rounding) fp32 accumulation.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sg2im {

constexpr int BK = 32;
constexpr int MLD = BK + 4;          // m-major LDS row stride (floats); 144 B keeps 16-B alignment
constexpr int KPAD = 4;              // k-major LDS row pad (floats)
constexpr int NTHREADS = 256;

typedef float f32x16 __attribute__((ext_vector_type(16)));

// One input "source": a dense NHWC tensor (or a row matrix when H=W=1) that supplies
// C consecutive channels of a virtual channel-concatenated operand.
struct Src {
  const float* p;          // base pointer
  const long long* gidx;   // optional row gather (row-matrix geometry only)
  const float* scale;      // optional fused per-channel affine + leaky (pending BN/act)
  const float* shift;
  float slope;             // leaky slope applied after the affine (1.0f = identity)
  int C;                   // channels supplied by this source
  int ld;                  // floats between consecutive pixels / rows
  int up;                  // log2 of nearest-neighbour upsampling (0 or 1)
  int bf;                  // 1: the tensor holds bfloat16 (sg2im_src.dtype; `ld` counts ELEMENTS either way) - the bf16
                           // halo'd kernels only (conv_halo.h, wgrad_halo.h)
};

struct ConvGeom {
  Src s0, s1, s2, s3;      // unused sources have C == 0 (named members: a dynamically indexed
                           // array inside a by-value kernel argument is demoted to scratch)
  int nsrc;
  int Ctot;                // sum of src[i].C  (channels of the virtual concat = K per tap)
  int Wtap;                // floats per tap of a WEIGHT row (>= Ctot; sg2im_conv_desc.weight_channels)
  int NB, H, W;            // batch and *logical* input size (after upsampling)
  int Ho, Wo;              // output size
  int KH, KW, stride, pad;
};

struct Epi {
  float* C;                // destination
  long long ldc;           // floats between destination rows
  const float* bias;       // per-column bias or nullptr
  float slope;             // leaky slope of the fused output activation (1 = none, 0 = ReLU)
  int accumulate;          // 1: C += result
  float* ws;               // split-K partials [nsplit][M][N]; used when nsplit > 1
  int nsplit;
  int col_ctot, col_wtap;  // weight gradients of a layer whose weight rows hold col_wtap > col_ctot floats per
                           // tap: GEMM column n = tap * col_ctot + c lands in destination column
                           // tap * col_wtap + c (0, 0: identity)
  // data gradients that flow into an activation (sg2im_conv2d_backward_data_act): the finished value is multiplied
  // by leaky'(act) = (act[row][n] > 0 ? 1 : mask_slope), act = the ACTIVATED output of the layer the gradient is
  // taken with respect to (same rows / columns as the destination) - the arithmetic of act_bwd_kernel (norm.hip),
  // applied by the epilogue or, with split-K, by the finish
  const float* mask;
  long long ld_mask;
  float mask_slope;
  int out_bf;              // 1: C holds bfloat16 (sg2im_conv_desc.out_dtype): the finished value is rounded (RNE) when it is
                           // stored; launches without split-K and without accumulate only (the entry points check)
};
__device__ __forceinline__ int epi_col(const Epi& e, int n) {
  return e.col_wtap ? (n / e.col_ctot) * e.col_wtap + n % e.col_ctot : n;
}

__device__ __forceinline__ int kperm(int s, int h) { return 8 * (s >> 2) + 4 * h + (s & 3); }

// field-by-field select: a whole-struct conditional copy out of the kernarg segment is
// lowered to memcpy into scratch
// (by-value sel4: a ternary over lvalues selects the *address* and again indexes scratch)
template <typename T> __device__ __forceinline__ T sel4(int s, T a, T b, T c, T d) {
  return s == 0 ? a : s == 1 ? b : s == 2 ? c : d;
}
#define SG2IM_PICK(f) sel4(s, g.s0.f, g.s1.f, g.s2.f, g.s3.f)
__device__ __forceinline__ Src pick_src(const ConvGeom& g, int s) {
  Src S;
  S.p = SG2IM_PICK(p); S.gidx = SG2IM_PICK(gidx); S.scale = SG2IM_PICK(scale); S.shift = SG2IM_PICK(shift);
  S.slope = SG2IM_PICK(slope); S.C = SG2IM_PICK(C); S.ld = SG2IM_PICK(ld); S.up = SG2IM_PICK(up); S.bf = SG2IM_PICK(bf);
  return S;
}
#undef SG2IM_PICK

// The Src block of source `s`, fetched from the kernel-argument segment with scalar loads.
// ConvGeom must sit at offset 0 of the kernel's (single, by-value) parameter struct.  Unlike
// pick_src this keeps no copy of the four sources in SGPRs, so it is the form to use where the
// index changes inside the main loop.
__device__ __forceinline__ Src kernarg_src(int s) {
  typedef __attribute__((address_space(4))) const Src* KSrc;
  const KSrc k = (KSrc)__builtin_amdgcn_kernarg_segment_ptr() + s;
  Src S;
  S.p = k->p; S.gidx = k->gidx; S.scale = k->scale; S.shift = k->shift;
  S.slope = k->slope; S.C = k->C; S.ld = k->ld; S.up = k->up; S.bf = k->bf;
  return S;
}

__device__ __forceinline__ float leaky(float v, float slope) { return v > 0.f ? v : v * slope; }

// channel c of the virtual concat -> (source index, channel within the source)
__device__ __forceinline__ void locate_channel(const ConvGeom& g, int c, int& s, int& cs) {
  const int e0 = g.s0.C, e1 = e0 + g.s1.C, e2 = e1 + g.s2.C;
  if (c < e0) { s = 0; cs = c; }
  else if (c < e1) { s = 1; cs = c - e0; }
  else if (c < e2) { s = 2; cs = c - e1; }
  else { s = 3; cs = c - e2; }
}

// ---------------------------------------------------------------------------
// LDS image sizes
// ---------------------------------------------------------------------------
template <int ROWS, bool KMAJOR> struct LdsTile {
  static constexpr int FLOATS = KMAJOR ? BK * (ROWS + KPAD) : ROWS * MLD;
};

// registers -> LDS.  Thread mapping (both images): NV = ROWS/32 float4 per thread.
//   m-major: col4 = tid & 7,            row = (tid >> 3) + 32 * i
//   k-major: col4 = tid % (ROWS/4),     krow = tid / (ROWS/4) + (1024/ROWS) * i
template <int ROWS, bool KMAJOR>
__device__ __forceinline__ void store_tile(float* lds, const float4 (&r)[ROWS / 32], int tid) {
  constexpr int NV = ROWS / 32;
  if (!KMAJOR) {
    const int col4 = tid & 7, r0 = tid >> 3;
    #pragma unroll
    for (int i = 0; i < NV; ++i)
      *reinterpret_cast<float4*>(lds + (r0 + 32 * i) * MLD + 4 * col4) = r[i];
  } else {
    constexpr int Q = ROWS / 4;
    const int col4 = tid % Q, k0 = tid / Q;
    #pragma unroll
    for (int i = 0; i < NV; ++i)
      *reinterpret_cast<float4*>(lds + (k0 + (1024 / ROWS) * i) * (ROWS + KPAD) + 4 * col4) = r[i];
  }
}

// ---------------------------------------------------------------------------
// one BK chunk of MFMAs for a wave owning TM x TN 32x32 tiles
// ---------------------------------------------------------------------------
// the A / B fragments of one BK chunk for a wave owning TM x TN 32x32 tiles
template <int BM, int BN> struct Frags { float a[BM / 64][16], b[BN / 64][16]; };

template <int BM, int BN, bool AK, bool BKM>
__device__ __forceinline__ void read_frags(const float* __restrict__ As, const float* __restrict__ Bs,
                                           int wm0, int wn0, int lane, Frags<BM, BN>& f) {
  constexpr int TM = BM / 64, TN = BN / 64;
  const int i = lane & 31, h = lane >> 5;
  #pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    if (!AK) {
      const float* row = As + (wm0 + tm * 32 + i) * MLD + 4 * h;
      #pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 v = *reinterpret_cast<const float4*>(row + 8 * g);
        f.a[tm][4 * g + 0] = v.x; f.a[tm][4 * g + 1] = v.y; f.a[tm][4 * g + 2] = v.z; f.a[tm][4 * g + 3] = v.w;
      }
    } else {
      #pragma unroll
      for (int s = 0; s < 16; ++s) f.a[tm][s] = As[kperm(s, h) * (BM + KPAD) + wm0 + tm * 32 + i];
    }
  }
  #pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    if (!BKM) {
      const float* row = Bs + (wn0 + tn * 32 + i) * MLD + 4 * h;
      #pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 v = *reinterpret_cast<const float4*>(row + 8 * g);
        f.b[tn][4 * g + 0] = v.x; f.b[tn][4 * g + 1] = v.y; f.b[tn][4 * g + 2] = v.z; f.b[tn][4 * g + 3] = v.w;
      }
    } else {
      #pragma unroll
      for (int s = 0; s < 16; ++s) f.b[tn][s] = Bs[kperm(s, h) * (BN + KPAD) + wn0 + tn * 32 + i];
    }
  }
}

template <int BM, int BN>
__device__ __forceinline__ void mma_frags(const Frags<BM, BN>& f, f32x16 (&acc)[BM / 64][BN / 64]) {
  constexpr int TM = BM / 64, TN = BN / 64;
  #pragma unroll
  for (int s = 0; s < 16; ++s) {
    #pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      #pragma unroll
      for (int tn = 0; tn < TN; ++tn)
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[tm][s], f.b[tn][s], acc[tm][tn], 0, 0, 0);
    }
  }
}

// one BK chunk of MFMAs: fragments from LDS, then the MFMA block
template <int BM, int BN, bool AK, bool BKM>
__device__ __forceinline__ void mma_chunk(const float* __restrict__ As, const float* __restrict__ Bs,
                                          int wm0, int wn0, int lane,
                                          f32x16 (&acc)[BM / 64][BN / 64]) {
  Frags<BM, BN> f;
  read_frags<BM, BN, AK, BKM>(As, Bs, wm0, wn0, lane, f);
  mma_frags<BM, BN>(f, acc);
}

// ---------------------------------------------------------------------------
// epilogue: C/D layout of the 32x32 MFMA: col = lane & 31,
// row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
// ---------------------------------------------------------------------------
struct IdentityRow { __device__ __forceinline__ long long operator()(int m) const { return m; } };

// `rowmap` turns a tile-local GEMM row into the destination row (identity except for the
// stride-2 parity classes of the data gradient and the 2-D pixel patches of conv_halo.h); split-K
// partials are only remapped with WSMAP (patches: [split][M][N] over the destination rows).
// FULL: every tile row is a real row (the 128-pixel patches of conv_halo.h) - no row bound test.
// The workgroup-uniform cases (split-K partials / bfloat16 destination / accumulate / plain store) are decided ONCE per
// column fragment and each has its own branch-free store loop: as one loop with the tests inside, the 32 stores of a
// thread were 32 x ~80 instructions of exec-mask juggling - as many VALU instructions as the whole main loop of a bf16
// launch (round 6, found in the ISA after SQ_INSTS_VALU / SQ_INSTS_MFMA = 9.4 on the bf16 forward kernel).
template <int BM, int BN, typename RowMap = IdentityRow, bool WSMAP = false, bool MASK = false, bool FULL = false>
__device__ __forceinline__ void epilogue(const Epi& e, int M, int N, int nlimit, int m0, int n0, int wm0,
                                         int wn0, int lane, int split,
                                         const f32x16 (&acc)[BM / 64][BN / 64], RowMap rowmap = RowMap()) {
  constexpr int TM = BM / 64, TN = BN / 64;
  const int j = lane & 31, h = lane >> 5;
  auto row_of = [&](int tm, int r) __attribute__((always_inline)) { return m0 + wm0 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h; };
  #pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int n = n0 + wn0 + tn * 32 + j;
    if (n >= nlimit) continue;
    if (e.nsplit > 1) {                                   // split-K partials [split][M][N]
      float* const wsn = e.ws + (long long)split * M * N + n;
      #pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        #pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = row_of(tm, r);
          if (!FULL && m >= M) continue;
          const long long mw = WSMAP ? (long long)rowmap(m) : (long long)m;
          wsn[mw * N] = acc[tm][tn][r];
        }
      }
      continue;
    }
    const float bv = e.bias ? e.bias[n] : 0.f;
    const int ncol = epi_col(e, n);
    if (e.out_bf) {                                       // bfloat16 destination (no accumulate: the entry points check)
      __bf16* const cb = reinterpret_cast<__bf16*>(e.C) + ncol;
      #pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        #pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = row_of(tm, r);
          if (!FULL && m >= M) continue;
          const long long drow = rowmap(m);
          float v = leaky(acc[tm][tn][r] + bv, e.slope);
          if (MASK) v *= e.mask[drow * e.ld_mask + n] > 0.f ? 1.f : e.mask_slope;
          cb[drow * e.ldc] = (__bf16)v;
        }
      }
    } else if (e.accumulate) {
      float* const cf = e.C + ncol;
      #pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        #pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = row_of(tm, r);
          if (!FULL && m >= M) continue;
          const long long drow = rowmap(m);
          float v = leaky(acc[tm][tn][r] + bv, e.slope);
          if (MASK) v *= e.mask[drow * e.ld_mask + n] > 0.f ? 1.f : e.mask_slope;
          cf[drow * e.ldc] += v;
        }
      }
    } else {
      float* const cf = e.C + ncol;
      #pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        #pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = row_of(tm, r);
          if (!FULL && m >= M) continue;
          const long long drow = rowmap(m);
          float v = leaky(acc[tm][tn][r] + bv, e.slope);
          if (MASK) v *= e.mask[drow * e.ld_mask + n] > 0.f ? 1.f : e.mask_slope;
          cf[drow * e.ldc] = v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// BatchNorm reductions in the GEMM epilogue (sg2im_conv2d_forward_bn / sg2im_conv2d_backward_data_bn):
// the workgroup that owns a BM x BN output tile also owns BM rows of BN per-channel sums.  Per column the
// partial sums of the tile's 2 (wave rows) x 2 (lane halves) row groups are combined through LDS in a
// fixed order and written to tile partials [K][N][tiles] (the tile index fastest: the finish reads a channel's
// partials as one contiguous run), which a one-launch finish (norm.hip) reduces in double - the same two-stage,
// atomics-free scheme as the standalone statistics kernels, minus their pass over the tensor.  `lds` is the
// operand image (free after the main loop), >= 9 * BN floats.
// ---------------------------------------------------------------------------
struct StatSink {
  float* partial;          // [3][N][tiles] (forward: pivot, sum d, sum d^2) or [2][N][tiles] (backward: sum du, sum du*xhat)
  int tiles;               // row tiles of the launch = the stride between two channels' partials
  const int* count;        // optional: only the first count[0] * unit rows are real (padded row batches)
  int unit;
  // backward only: the normalised layer's pre-BN output and its statistics
  const float* y; long long ld_y;
  const float* mean; const float* invstd; const float* scale; const float* shift;
  float slope;
  int pool2, H, W;         // pool2: this launch's rows are pixels (n, h, w) of an H x W map at TWICE y's resolution
  int y_bf;                // backward: y holds bfloat16 (sg2im_bn_bwd.y_dtype)
};

__device__ __forceinline__ int live_limit(const StatSink& ss, int M) {
  if (!ss.count) return M;
  const long long t = (long long)ss.count[0] * ss.unit;
  return t < M ? (int)t : M;
}

// forward: statistics of the values the epilogue stored, v = leaky(acc + bias), as pivot-shifted sums with the
// tile's first row as the pivot (sum d, sum d^2 with d = v - pivot: free of the cancellation of E[x^2] - mean^2)
template <int BM, int BN, bool FULL = false>
__device__ __forceinline__ void epilogue_stats(const Epi& e, const StatSink& ss, int M, int N, int m0, int n0, int wm0,
                                               int wn0, int lane, int tid, int tile,
                                               const f32x16 (&acc)[BM / 64][BN / 64], float* lds) {
  constexpr int TM = BM / 64, TN = BN / 64;
  const int j = lane & 31, h = lane >> 5;
  const int wave_m = wm0 / (BM / 2);
  float* piv = lds;                  // [BN]
  float* red = lds + BN;             // [2][4][BN]
  const int Mlive = live_limit(ss, M);
  __syncthreads();                   // (every wave is done with the operand image)
  if (wave_m == 0 && h == 0) {
    #pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int col = wn0 + tn * 32 + j, n = n0 + col;
      const float bv = (n < N && e.bias) ? e.bias[n] : 0.f;
      piv[col] = leaky(acc[0][tn][0] + bv, e.slope);
    }
  }
  __syncthreads();
  #pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int col = wn0 + tn * 32 + j, n = n0 + col;
    const float bv = (n < N && e.bias) ? e.bias[n] : 0.f;
    const float pv = piv[col];
    float s0 = 0.f, s1 = 0.f;
    #pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      #pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm0 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (FULL || m < Mlive) {
          const float d = leaky(acc[tm][tn][r] + bv, e.slope) - pv;
          s0 += d; s1 = fmaf(d, d, s1);
        }
      }
    }
    red[(wave_m * 2 + h) * BN + col] = s0;
    red[(4 + wave_m * 2 + h) * BN + col] = s1;
  }
  __syncthreads();
  for (int col = tid; col < BN; col += NTHREADS) {
    const int n = n0 + col;
    if (n >= N) continue;
    const float s0 = ((red[col] + red[BN + col]) + red[2 * BN + col]) + red[3 * BN + col];
    const float s1 = ((red[4 * BN + col] + red[5 * BN + col]) + red[6 * BN + col]) + red[7 * BN + col];
    float* dst = ss.partial + (size_t)n * ss.tiles + tile;
    const size_t plane = (size_t)N * ss.tiles;
    dst[0] = piv[col]; dst[plane] = s0; dst[2 * plane] = s1;
  }
}

// backward: this launch's result is gz = d(loss)/d(activated output of a BatchNorm'd layer).  With
// u = scale * y + shift (the normalised, pre-activation value) and du = gz * leaky'(u):
//   sum du  and  sum du * (y - mean) * invstd   per channel - what sg2im_bn_act_backward's first pass computes.
// pool2: the rows are at twice y's resolution (nearest-upsample backward): by linearity the sums of the 2x2-pooled
// gradient equal the sums over the fine pixels with y read at the coarse pixel.
// rowmap: tile-local row -> row of the launch's result tensor (identity, or a conv_halo.h pixel patch)
template <bool B> struct BoolC { static constexpr bool value = B; };
// (row of the BatchNorm'd layer's y for a result row at TWICE its resolution: generic form by division; a row map that
// knows the pixel's coordinates - conv_halo.h PatchRow - provides pool2() and saves the two divisions per element)
template <typename RowMap> struct HasPool2 { static constexpr bool value = false; };
template <typename RowMap>
__device__ __forceinline__ long long pooled_row(const RowMap& rowmap, int m, const StatSink& ss) {
  if constexpr (HasPool2<RowMap>::value) return rowmap.pool2(m);
  else {
    const int HW = ss.H * ss.W;
    const int mg = (int)rowmap(m);
    const int nb = mg / HW, rem = mg - nb * HW;
    const int hi = rem / ss.W, wi = rem - hi * ss.W;
    return ((long long)nb * (ss.H >> 1) + (hi >> 1)) * (ss.W >> 1) + (wi >> 1);
  }
}

template <int BM, int BN, typename RowMap = IdentityRow, bool FULL = false>
__device__ __forceinline__ void epilogue_bnbwd(const StatSink& ss, int M, int N, int m0, int n0, int wm0, int wn0,
                                               int lane, int tid, int tile,
                                               const f32x16 (&acc)[BM / 64][BN / 64], float* lds, RowMap rowmap = RowMap()) {
  constexpr int TM = BM / 64, TN = BN / 64;
  const int j = lane & 31, h = lane >> 5;
  const int wave_m = wm0 / (BM / 2);
  float* red = lds;                  // [2][4][BN]
  const int Mlive = live_limit(ss, M);
  float s0[TN], s1[TN], sc[TN], sh[TN], mu[TN], is[TN];
  bool okn[TN];
  #pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int n = n0 + wn0 + tn * 32 + j;
    const bool ok = n < N;
    okn[tn] = ok;
    s0[tn] = 0.f; s1[tn] = 0.f;
    sc[tn] = ok ? ss.scale[n] : 0.f; sh[tn] = ok ? ss.shift[n] : 0.f;
    mu[tn] = ok ? ss.mean[n] : 0.f; is[tn] = ok ? ss.invstd[n] : 0.f;
  }
  // the two workgroup-uniform facts (rows at twice y's resolution / y stored as bfloat16) pick one of four branch-free
  // accumulation loops
  auto sums = [&](auto pool_c, auto bf_c) __attribute__((always_inline)) {
    constexpr bool POOL = decltype(pool_c)::value, YBF = decltype(bf_c)::value;
    #pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      #pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm0 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (!FULL && m >= Mlive) continue;
        const long long row = POOL ? pooled_row(rowmap, m, ss) : (long long)rowmap(m);
        #pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          const int n = n0 + wn0 + tn * 32 + j;
          if (!okn[tn]) continue;
          const float yv = YBF ? (float)reinterpret_cast<const __bf16*>(ss.y)[row * ss.ld_y + n] : ss.y[row * ss.ld_y + n];
          const float du = acc[tm][tn][r] * (fmaf(yv, sc[tn], sh[tn]) > 0.f ? 1.f : ss.slope);
          s0[tn] += du; s1[tn] = fmaf(du, (yv - mu[tn]) * is[tn], s1[tn]);
        }
      }
    }
  };
  using T_ = BoolC<true>; using F_ = BoolC<false>;
  if (ss.pool2) { if (ss.y_bf) sums(T_{}, T_{}); else sums(T_{}, F_{}); }
  else { if (ss.y_bf) sums(F_{}, T_{}); else sums(F_{}, F_{}); }
  __syncthreads();                   // (every wave is done with the operand image)
  #pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int col = wn0 + tn * 32 + j;
    red[(wave_m * 2 + h) * BN + col] = s0[tn];
    red[(4 + wave_m * 2 + h) * BN + col] = s1[tn];
  }
  __syncthreads();
  for (int col = tid; col < BN; col += NTHREADS) {
    const int n = n0 + col;
    if (n >= N) continue;
    const float a = ((red[col] + red[BN + col]) + red[2 * BN + col]) + red[3 * BN + col];
    const float b = ((red[4 * BN + col] + red[5 * BN + col]) + red[6 * BN + col]) + red[7 * BN + col];
    float* dst = ss.partial + (size_t)n * ss.tiles + tile;
    dst[0] = a; dst[(size_t)N * ss.tiles] = b;
  }
}

// ---------------------------------------------------------------------------
// bf16 operand path (compute_dtype 1): the SAME loaders / staging registers / pipeline, but the
// operands are rounded to bf16 (RNE, v_cvt_pk_bf16_f32) when they are written to LDS and multiplied
// with v_mfma_f32_32x32x16_bf16 (fp32 accumulate) - 16x the fp32 matrix rate, half the LDS bytes.
// Tensors in HBM stay fp32 (activations, gradients, master weights), so BatchNorm statistics, losses
// and Adam are untouched.  LDS images hold bf16:
//   m-major  [rows][BK + 8]       80-byte rows: a lane's 8 consecutive k (one MFMA operand) = one
//                                 ds_read_b128, conflict free (rows 20 banks apart)
//   k-major  [BK][rows + 32]      read with ds_read_b64_tr_b16 - the hardware 4x16 transpose read
//                                 (probed: tools/_src/tr_probe.hip): the 16 lanes of a group pass the
//                                 addresses of 4 rows x 4 four-element pieces and receive 4 consecutive
//                                 k of ONE column each; two reads give the 8 k of an operand.  Rows
//                                 (rows+32)*2 B apart -> the 4 rows of a read sit on disjoint banks.
// Operand layout of the MFMA: lane l: A[i = l & 31][k = 8 (l >> 5) + 0..7], B[k = same][j = l & 31];
// a BK = 32 chunk is two K = 16 steps.
// ---------------------------------------------------------------------------
typedef __bf16 bf16_t;
typedef bf16_t bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16_t bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
constexpr int MLDH = BK + 8;         // m-major bf16 row stride (elements)
constexpr int KPADH = 32;            // k-major bf16 row pad (elements)

template <int ROWS, bool KMAJOR> struct LdsTileH {
  static constexpr int HALFS = KMAJOR ? BK * (ROWS + KPADH) : ROWS * MLDH;
};

// LDS bytes of one operand image
template <bool BF, int ROWS, bool KMAJOR> struct TileBytes {
  static constexpr int value = BF ? LdsTileH<ROWS, KMAJOR>::HALFS * 2 : LdsTile<ROWS, KMAJOR>::FLOATS * 4;
};

// four bfloat16 (the 8 bytes a bf16-storage loader fetched into .x / .y of a float4 register quad) -> four floats
__device__ __forceinline__ float4 unpack_bf16x4(const float4& raw) {
  const unsigned lo = __float_as_uint(raw.x), hi = __float_as_uint(raw.y);
  return make_float4(__uint_as_float(lo << 16), __uint_as_float(lo & 0xffff0000u),
                     __uint_as_float(hi << 16), __uint_as_float(hi & 0xffff0000u));
}

__device__ __forceinline__ bf16x4 to_bf16x4(const float4& v) {
  const f32x4v f = {v.x, v.y, v.z, v.w};
  return __builtin_convertvector(f, bf16x4);
}

// registers -> LDS, same thread mapping as store_tile, 8 bytes per float4
template <int ROWS, bool KMAJOR>
__device__ __forceinline__ void store_tile_h(bf16_t* lds, const float4 (&r)[ROWS / 32], int tid) {
  constexpr int NV = ROWS / 32;
  if (!KMAJOR) {
    const int col4 = tid & 7, r0 = tid >> 3;
    #pragma unroll
    for (int i = 0; i < NV; ++i)
      *reinterpret_cast<bf16x4*>(lds + (r0 + 32 * i) * MLDH + 4 * col4) = to_bf16x4(r[i]);
  } else {
    constexpr int Q = ROWS / 4;
    const int col4 = tid % Q, k0 = tid / Q;
    #pragma unroll
    for (int i = 0; i < NV; ++i)
      *reinterpret_cast<bf16x4*>(lds + (k0 + (1024 / ROWS) * i) * (ROWS + KPADH) + 4 * col4) = to_bf16x4(r[i]);
  }
}

template <int BM, int BN> struct FragsH { bf16x8 a[BM / 64][2], b[BN / 64][2]; };

// one operand (8 consecutive k of row/column `rc0 + (lane & 31)`, K step `step`) from a k-major image
template <int ROWS>
__device__ __forceinline__ bf16x8 read_tr(const bf16_t* img, int rc0, int step, int lane) {
  typedef __attribute__((address_space(3))) bf16x4* LdsPtr;
  const int p = lane & 15, g = lane >> 4;
  const bf16_t* a = img + (16 * step + 8 * (g >> 1) + (p >> 2)) * (ROWS + KPADH) + rc0 + 16 * (g & 1) + 4 * (p & 3);
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LdsPtr)a);
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LdsPtr)(a + 4 * (ROWS + KPADH)));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int BM, int BN, bool AK, bool BKM>
__device__ __forceinline__ void read_frags_h(const bf16_t* __restrict__ As, const bf16_t* __restrict__ Bs,
                                             int wm0, int wn0, int lane, FragsH<BM, BN>& f) {
  constexpr int TM = BM / 64, TN = BN / 64;
  const int i = lane & 31, h = lane >> 5;
  #pragma unroll
  for (int st = 0; st < 2; ++st) {
    #pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      if (!AK) f.a[tm][st] = *reinterpret_cast<const bf16x8*>(As + (wm0 + tm * 32 + i) * MLDH + 16 * st + 8 * h);
      else f.a[tm][st] = read_tr<BM>(As, wm0 + tm * 32, st, lane);
    }
    #pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      if (!BKM) f.b[tn][st] = *reinterpret_cast<const bf16x8*>(Bs + (wn0 + tn * 32 + i) * MLDH + 16 * st + 8 * h);
      else f.b[tn][st] = read_tr<BN>(Bs, wn0 + tn * 32, st, lane);
    }
  }
}

template <int BM, int BN>
__device__ __forceinline__ void mma_frags_h(const FragsH<BM, BN>& f, f32x16 (&acc)[BM / 64][BN / 64]) {
  constexpr int TM = BM / 64, TN = BN / 64;
  #pragma unroll
  for (int st = 0; st < 2; ++st) {
    #pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      #pragma unroll
      for (int tn = 0; tn < TN; ++tn)
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[tm][st], f.b[tn][st], acc[tm][tn], 0, 0, 0);
    }
  }
}

// Software pipeline shared by the three kernels: the global loads of the next K-chunk are kept
// in flight in registers while the matrix cores work on the chunk staged in LDS.
//   load(it)  : global -> registers          stage(live) : registers -> LDS
//   mma(0) / mma(1): fragments of the LDS image -> registers / the BK chunk of MFMAs
// The main loop shared by the three kernels - ONE register set, ONE LDS image per operand and a
// branch-free loop body: the loads of chunk i+1, the MFMAs of chunk i and the LDS stores of chunk i+1
// are one basic block, so the compiler can slot the loader's address arithmetic into the 64-cycle
// shadows of the MFMAs (a wave issues ~8 other instructions per fp32 MFMA for free).  The last chunk
// is fetched and staged a second time instead of guarding the tail with branches; that copy is never
// read (`live` = false for it).  A single LDS image means two barriers per chunk (every wave must be
// done reading chunk i before it is overwritten) but half the LDS, i.e. twice the resident workgroups.
// [Round 2 measured and dropped: a two-image / two-register-set loop for the small tiles, a 512-thread
// ping-pong form, two chunks per barrier interval, staggered workgroup starts and a direct-to-LDS
// loop - none faster on the training step; see DESIGN.md section 4.1 and profiles/r2_*_ab.log.]
template <typename Load, typename Stage, typename Mma>
__device__ __forceinline__ void k_pipeline(int it_begin, int it_end, Load load, Stage stage, Mma mma) {
  const int n = it_end - it_begin;
  if (n <= 0) return;
  load(it_begin);
  stage(true);
  __syncthreads();
  #pragma unroll 1
  for (int i = 0; i < n; ++i) {
    const int nxt = i + 1 < n ? i + 1 : n - 1;
    load(it_begin + nxt);
    // keep the global loads ahead of the MFMA block: left alone, the scheduler sinks them
    // to just before their first use (end of the block) and the wave stalls on vmcnt
    __builtin_amdgcn_sched_barrier(0);
    mma(0); mma(1);
    __syncthreads();
    stage(i + 1 < n);
    __syncthreads();
  }
}

// wave placement inside the block tile: 2 x 2 wavefronts
template <int BM, int BN>
__device__ __forceinline__ void wave_origin(int tid, int& wm0, int& wn0, int& lane) {
  const int wave = tid >> 6;
  lane = tid & 63;
  wm0 = (wave >> 1) * (BM / 2);
  wn0 = (wave & 1) * (BN / 2);
}

}  // namespace sg2im
