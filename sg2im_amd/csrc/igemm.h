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
  S.slope = SG2IM_PICK(slope); S.C = SG2IM_PICK(C); S.ld = SG2IM_PICK(ld); S.up = SG2IM_PICK(up);
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
  S.slope = k->slope; S.C = k->C; S.ld = k->ld; S.up = k->up;
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
// stride-2 parity classes of the data gradient); split-K partials are not remapped.
template <int BM, int BN, typename RowMap = IdentityRow>
__device__ __forceinline__ void epilogue(const Epi& e, int M, int N, int nlimit, int m0, int n0, int wm0,
                                         int wn0, int lane, int split,
                                         const f32x16 (&acc)[BM / 64][BN / 64], RowMap rowmap = RowMap()) {
  constexpr int TM = BM / 64, TN = BN / 64;
  const int j = lane & 31, h = lane >> 5;
  #pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int n = n0 + wn0 + tn * 32 + j;
    if (n >= nlimit) continue;
    const float bv = (e.nsplit == 1 && e.bias) ? e.bias[n] : 0.f;
    const int ncol = epi_col(e, n);
    #pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      #pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm0 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m >= M) continue;
        float v = acc[tm][tn][r];
        if (e.nsplit > 1) {
          e.ws[((long long)split * M + m) * N + n] = v;
        } else {
          v = leaky(v + bv, e.slope);
          float* dst = e.C + rowmap(m) * e.ldc + ncol;
          if (e.accumulate) v += *dst;
          *dst = v;
        }
      }
    }
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
//   load(it, set)        : global -> register set     stage(set, buf, live) : registers -> LDS buffer
//   mma(0, buf) / (1, buf): fragments of LDS buffer `buf` -> registers / the BK chunk of MFMAs
// (register sets / LDS buffers are selected by compile-time tags so the register arrays are
// never dynamically indexed, which would demote them to scratch; `live` is false for the
// re-staged copy of the last chunk, see k_pipeline_d1)
template <int I> struct IC { static constexpr int value = I; };

#ifndef SG2IM_SMALL_TILE_DEPTH
#define SG2IM_SMALL_TILE_DEPTH 1   // (2 measured: no gain on the tiny GEMMs, fewer resident waves)
#endif
#ifndef SG2IM_LDS_STAGES
#define SG2IM_LDS_STAGES 1
#endif
constexpr int LDS_STAGES = SG2IM_LDS_STAGES;

#ifndef SG2IM_SCHED_FENCE
#define SG2IM_SCHED_FENCE 0      // (see SG2IM_LAUNDER in conv.hip: measured slower)
#endif
#ifndef SG2IM_ABL
#define SG2IM_ABL 0        // timing-only ablations (1: no loads/stores in the loop, 2: no barriers)
#endif
// (Measured and dropped, round 2: delaying the co-resident workgroups of a CU by 1-3k cycles at
// kernel start - by linear block id or by the hardware wave slot - to de-phase their loader / MFMA
// phases changed nothing, 86.9 vs 86.0-86.7 TFLOP/s forward over the layer table.)
// Depth-1 variant: ONE register set and a branch-free loop body - the loads of chunk i+1,
// the MFMAs of chunk i and the LDS stores of chunk i+1 are one basic block, so the compiler
// can slot the loader's address arithmetic into the 64-cycle shadows of the MFMAs (a wave
// issues ~8 other instructions per fp32 MFMA for free).  The last chunk is fetched and
// staged a second time instead of guarding the tail with branches; that copy is never read.
template <typename Load, typename Stage, typename Mma>
__device__ __forceinline__ void k_pipeline_d1(int it_begin, int it_end, Load load, Stage stage, Mma mma) {
  const int n = it_end - it_begin;
  if (n <= 0) return;
  load(it_begin, IC<0>());
  stage(IC<0>(), 0, true);
  __syncthreads();
  int cur = 0;
  #pragma unroll 1
  for (int i = 0; i < n; ++i) {
    const int nxt = i + 1 < n ? i + 1 : n - 1;
#if !(SG2IM_ABL & 1)
    load(it_begin + nxt, IC<0>());
#endif
    // keep the global loads ahead of the MFMA block: left alone, the scheduler sinks them
    // to just before their first use (end of the block) and the wave stalls on vmcnt
    __builtin_amdgcn_sched_barrier(0);
    if (LDS_STAGES == 2) {
      mma(0, cur); mma(1, cur);
      stage(IC<0>(), cur ^ 1, i + 1 < n);
      __syncthreads();
      cur ^= 1;
    } else {
      // single LDS image (half the LDS -> twice the resident workgroups): every wave must
      // be done reading chunk i before it is overwritten, hence the second barrier
      mma(0, 0); mma(1, 0);
#if SG2IM_SCHED_FENCE
      __builtin_amdgcn_sched_barrier(0);       // (experiment: keep stage(i+1) out of the MFMA block)
#endif
#if !(SG2IM_ABL & 2)
      __syncthreads();
#endif
#if !(SG2IM_ABL & 1)
      stage(IC<0>(), 0, i + 1 < n);
#endif
#if !(SG2IM_ABL & 2)
      __syncthreads();
#endif
    }
  }
}
// Depth-2 variant (used for the 64x64 tile): TWO register sets and TWO LDS images, one barrier
// per chunk - two chunks of global loads stay in flight.  The small-tile launches are tiny
// GEMMs with few workgroups per CU, where nothing else hides the load -> store -> barrier ->
// MFMA latency chain of each chunk (measured ~1.2 us per chunk with the depth-1 loop).
template <typename Load, typename Stage, typename Mma>
__device__ __forceinline__ void k_pipeline_d2(int it_begin, int it_end, Load load, Stage stage, Mma mma) {
  const int n = it_end - it_begin;
  if (n <= 0) return;
  load(it_begin, IC<0>());
  if (n > 1) load(it_begin + 1, IC<1>());
  stage(IC<0>(), 0, true);
  __syncthreads();
  // ONE mma call site (runtime LDS buffer index): with the MFMA block instantiated twice
  // hipcc gives each copy its own accumulator registers and doubles the AGPR budget.
  // The loads / stages exist twice (wave-uniform branch on the chunk parity) so that each
  // copy addresses its register set statically.
  #pragma unroll 1
  for (int i = 0; i < n; ++i) {
    const int par = i & 1;
    // (the distinct asm markers keep hipcc from merging the two branch bodies back into
    // one copy that selects the register set through a pointer, i.e. through scratch)
    if (i + 2 < n) {
      if (par == 0) {
        asm volatile("; k_pipeline: load set 0" ::: "memory");
        load(it_begin + i + 2, IC<0>());
        asm volatile("; k_pipeline: load set 0 done" ::: "memory");
      } else {
        asm volatile("; k_pipeline: load set 1" ::: "memory");
        load(it_begin + i + 2, IC<1>());
        asm volatile("; k_pipeline: load set 1 done" ::: "memory");
      }
    }
    mma(0, par); mma(1, par);
    if (i + 1 < n) {
      if (par == 0) {
        asm volatile("; k_pipeline: stage set 1" ::: "memory");
        stage(IC<1>(), 1, true);
        asm volatile("; k_pipeline: stage set 1 done" ::: "memory");
      } else {
        asm volatile("; k_pipeline: stage set 0" ::: "memory");
        stage(IC<0>(), 0, true);
        asm volatile("; k_pipeline: stage set 0 done" ::: "memory");
      }
    }
    __syncthreads();
  }
}

// Ping-pong variant: a 512-thread workgroup = two 256-thread HALVES, each with its own output tile,
// LDS image and register set, running the depth-1 loop one barrier apart: while half 0 is in its
// MFMA block, half 1 converts / stores its next chunk to LDS and issues the global loads of the one
// after, and vice versa - the two waves that share a SIMD are, by construction, never both in their
// loader phase (which left the matrix pipe idle: time per K chunk was N x MFMA block + one full
// loader phase for N = 1..4 co-resident 256-thread workgroups, i.e. the loader phases of co-resident
// waves coincided instead of hiding under each other's MFMAs).  Every s_barrier is workgroup-wide
// (all 8 waves); the halves execute the same NUMBER of barriers, half 1 offset by one.
//   barrier #     half 0                          half 1
//   0             (prologue: chunk 0 staged, chunk 1 in registers - both halves)
//   1             MFMA(0)                          -
//   2             stage(1), load(2)                MFMA(0)
//   3             MFMA(1)                          stage(1), load(2)
//   ...
// `half` must be wave-uniform and the K range identical for both halves.
template <typename Load, typename Stage, typename Mma>
__device__ __forceinline__ void k_pipeline_pp(int half, int it_begin, int it_end, Load load, Stage stage, Mma mma) {
  const int n = it_end - it_begin;
  if (n <= 0) return;
  load(it_begin, IC<0>());
  stage(IC<0>(), 0, true);
  load(it_begin + (n > 1 ? 1 : 0), IC<0>());
  __syncthreads();                                  // # 0
  if (half) __syncthreads();                        // # 1 (half 1 sits out half 0's first MFMA block)
  #pragma unroll 1
  for (int i = 0; i < n; ++i) {
    mma(0, 0); mma(1, 0);
    __syncthreads();
    // chunk i+1 to LDS (after the last chunk: a copy that is never read), then the loads of chunk
    // i+2 - a whole MFMA block + loader phase ahead of their use
    stage(IC<0>(), 0, i + 1 < n);
    load(it_begin + (i + 2 < n ? i + 2 : n - 1), IC<0>());
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
  }
  if (!half) __syncthreads();                       // (same barrier count as half 1)
}

// Paired variant: TWO K chunks per barrier interval (an effective K step of 64 with the BK = 32
// loaders, LDS images and fragment reads): chunks 2i / 2i+1 sit in LDS images 0 / 1, their
// successors in register sets 0 / 1.  Per pair: one load block, 2 x (fragment reads + MFMA block),
// barrier, two stages, barrier - half the barriers and half the store -> barrier -> read latency
// chains per MFMA of the depth-1 loop, for twice its LDS and staging registers.  An odd chunk
// count is padded with an all-zero LDS image in FRONT (zfill(1) - one wasted MFMA block per
// workgroup) so that the MFMA blocks exist only inside the loop body: a second call site makes
// hipcc give each copy its own accumulator registers.
template <typename Load, typename Stage, typename Mma, typename Zfill>
__device__ __forceinline__ void k_pipeline_x2(int it_begin, int it_end, Load load, Stage stage, Mma mma, Zfill zfill) {
  const int n = it_end - it_begin;
  if (n <= 0) return;
  int pos = it_begin;
  load(pos, IC<0>());
  stage(IC<0>(), 0, true);
  if (n & 1) {
    zfill(1);
    pos += 1;
  } else {
    load(pos + 1, IC<1>());
    stage(IC<1>(), 1, true);
    pos += 2;
  }
  __syncthreads();
  const int pairs = (n + 1) >> 1;
  #pragma unroll 1
  for (int i = 0; i < pairs; ++i, pos += 2) {
    // (after the last pair: the final chunk is fetched and staged twice more, copies that are never
    // read - the loaders' cursors do not advance past index it_end - 1)
    const bool more = pos < it_end;
    load(more ? pos : it_end - 1, IC<0>()); load(more ? pos + 1 : it_end - 1, IC<1>());
    __builtin_amdgcn_sched_barrier(0);
    mma(0, 0); mma(1, 0);
    mma(0, 1); mma(1, 1);
    __syncthreads();
    stage(IC<0>(), 0, more); stage(IC<1>(), 1, more);
    __syncthreads();
  }
}

// PD = 1: single LDS image / one register set (large tiles, occupancy bound)
// PD = 2: double LDS image / two register sets (small tiles, latency bound)
// PD = 3: paired chunks (k_pipeline_x2)
template <int PD, typename Load, typename Stage, typename Mma, typename Zfill>
__device__ __forceinline__ void k_pipeline(int it_begin, int it_end, Load load, Stage stage, Mma mma, Zfill zfill) {
  if constexpr (PD == 3) k_pipeline_x2(it_begin, it_end, load, stage, mma, zfill);
  else if constexpr (PD == 2) k_pipeline_d2(it_begin, it_end, load, stage, mma);
  else k_pipeline_d1(it_begin, it_end, load, stage, mma);
}

#ifndef SG2IM_X2
#define SG2IM_X2 0         // bit 0: 128x128, bit 1: 128x64 / 64x128, bit 2: 64x64 tiles use the paired loop
// (measured, profiles/r2_paired_chunk_ab.log: 128x128 forward +2 %, weight gradient -2 %, 128x64 tiles
// -5 % - the lost resident workgroup costs what the saved barriers gain; step 10.29 -> 10.34 / 10.6 ms. OFF.)
#endif
// pipeline depth / LDS images per tile shape
template <int BM, int BN> struct TilePipe {
  static constexpr bool X2 = (BM * BN == 128 * 128) ? (SG2IM_X2 & 1) != 0
                           : (BM * BN == 128 * 64) ? (SG2IM_X2 & 2) != 0 : (SG2IM_X2 & 4) != 0;
  static constexpr int DEPTH = X2 ? 3 : (BM * BN <= 64 * 64) ? SG2IM_SMALL_TILE_DEPTH : 1;
  static constexpr int LDS_IMAGES = DEPTH >= 2 ? 2 : LDS_STAGES;
};

// wave placement inside the block tile: 2 x 2 wavefronts
template <int BM, int BN>
__device__ __forceinline__ void wave_origin(int tid, int& wm0, int& wn0, int& lane) {
  const int wave = tid >> 6;
  lane = tid & 63;
  wm0 = (wave >> 1) * (BM / 2);
  wn0 = (wave & 1) * (BN / 2);
}

}  // namespace sg2im
