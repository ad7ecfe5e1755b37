// Convolution / linear layers as implicit GEMM on the fp32 matrix cores.
//
//   conv_fwd    out[pix][co]        = sum_{tap,c} X[pix (+) tap][c] * W[co][tap][c]     (NT)
//   conv_dgrad  dX[inpix][c]        = sum_{tap,co} dY[inpix (-) tap][co] * W[co][tap][c] (NN)
//   conv_wgrad  dW[co][tap][c]      = sum_{pix} dY[pix][co] * X[pix (+) tap][c]          (TN)
//
// Activations are NHWC, weights are [Cout][KH][KW][Ctot] (the physical layout of a
// torch channels_last (Cout,Cin,KH,KW) parameter).  A linear layer is the 1x1 case on
// a [rows,1,1,K] tensor.  The X operand is a *virtual* tensor: the channel concat of
// up to four sources, each optionally nearest-upsampled x2, row-gathered, and with a
// pending per-channel affine + LeakyReLU (the previous layer's BatchNorm + activation)
// applied on the fly - so torch.cat / F.upsample / obj_vecs[idx] / BN-apply of the
// reference (sg2im/crn.py:58-64,107; sg2im/graph.py:77-82; sg2im/layers.py:166-169)
// are never materialised.
#include <algorithm>
#define SG2IM_GEMM_TU 1
#include "launch_count.h"
#include "gcn_persist.h"
#include "igemm.h"
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include "sg2im_hip.h"
#include "bn_final.h"

namespace sg2im {

struct FwdParams {
  ConvGeom g;
  const float* Wt;
  int Cout;
  int M;            // NB*Ho*Wo
  int iters;        // total K iterations
  int nch;          // K chunks per tap (VEC=4)
  Epi e;
  StatSink st;      // ST kernels: BatchNorm statistics of the output tile (igemm.h epilogue_stats)
};

struct DgradParams {
  ConvGeom g;       // geometry of the *forward* conv; g.s0 describes dY: p, C=Cout, ld
  const float* Wt;
  int c_begin, Nc;  // input-channel range [c_begin, c_begin+Nc) produced by this launch
  int M;            // NB*H*W (rows of the largest parity class when parity != 0)
  int iters, nch;   // (parity == 0) total K iterations / chunks per tap
  int parity;       // 1: stride-2 parity decomposition, blockIdx.z = class (no split-K)
  Epi e;
  StatSink st;      // ST kernels: BatchNorm-backward sums of the output tile (igemm.h epilogue_bnbwd)
};

struct WgradParams {
  ConvGeom g;
  const float* dY;  // [NB*Ho*Wo][ldy]
  int ldy;
  int Cout;
  int P;            // NB*Ho*Wo (reduction length)
  int iters;        // ceil(P / BK)
  int ntile_c;      // column tiles per tap (VEC=4)
  int ntiles_n, ntiles_m;   // tile counts (the launch is a 1-D, XCD-swizzled grid)
  Epi e;
  float* dbias;     // optional [Cout]: column sums of dY, produced by the column-tile-0 workgroups
  float* ws_bias;   // split-K partials of dbias: [nsplit][Cout] behind the dW partials
  int background;   // host side only: sg2im_conv_desc.launch_hints & SG2IM_HINT_BACKGROUND
};

// ---------------------------------------------------------------------------
// virtual-tensor element access
// ---------------------------------------------------------------------------
__device__ __forceinline__ long long pixel_row(const ConvGeom& g, const Src& S, int n, int hi, int wi) {
  if (S.gidx) return S.gidx[n];
  return ((long long)n * (g.H >> S.up) + (hi >> S.up)) * (g.W >> S.up) + (wi >> S.up);
}

__device__ __forceinline__ float4 load4(const ConvGeom& g, const Src& S, int n, int hi, int wi, int c) {
  float4 v = *reinterpret_cast<const float4*>(S.p + pixel_row(g, S, n, hi, wi) * S.ld + c);
  if (S.scale) {
    const float4 sc = *reinterpret_cast<const float4*>(S.scale + c);
    const float4 sh = *reinterpret_cast<const float4*>(S.shift + c);
    v.x = leaky(fmaf(v.x, sc.x, sh.x), S.slope);
    v.y = leaky(fmaf(v.y, sc.y, sh.y), S.slope);
    v.z = leaky(fmaf(v.z, sc.z, sh.z), S.slope);
    v.w = leaky(fmaf(v.w, sc.w, sh.w), S.slope);
  }
  return v;
}

__device__ __forceinline__ float load1(const ConvGeom& g, const Src& S, int n, int hi, int wi, int c) {
  float v = S.p[pixel_row(g, S, n, hi, wi) * S.ld + c];
  if (S.scale) v = leaky(fmaf(v, S.scale[c], S.shift[c]), S.slope);
  return v;
}

__device__ __forceinline__ void zero_acc(f32x16& a) {
  #pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = 0.f;
}

__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// K-chunk `ch` (within one tap) -> source, its first concat channel, chunk base
__device__ __forceinline__ void locate_chunk(const ConvGeom& g, int ch, int& s, int& cstart, int& cb) {
  const int n0 = (g.s0.C + BK - 1) / BK, n1 = (g.s1.C + BK - 1) / BK, n2 = (g.s2.C + BK - 1) / BK;
  if (ch < n0) { s = 0; cstart = 0; }
  else if (ch < n0 + n1) { s = 1; cstart = g.s0.C; ch -= n0; }
  else if (ch < n0 + n1 + n2) { s = 2; cstart = g.s0.C + g.s1.C; ch -= n0 + n1; }
  else { s = 3; cstart = g.s0.C + g.s1.C + g.s2.C; ch -= n0 + n1 + n2; }
  cb = ch * BK;
}

#define SG2IM_ZERO_ACC()                                   \
  Frags<BM, BN> frags;                                     \
  FragsH<BM, BN> fragsh;                                   \
  f32x16 acc[BM / 64][BN / 64];                            \
  _Pragma("unroll") for (int a_ = 0; a_ < BM / 64; ++a_)   \
    _Pragma("unroll") for (int b_ = 0; b_ < BN / 64; ++b_) zero_acc(acc[a_][b_]);

// ---------------------------------------------------------------------------
// Fast-path loader conventions (VEC == 4).  Global loads are issued BRANCH-FREE: an
// out-of-range row / channel chunk reads element 0 of its tensor (always mapped) and is
// zeroed later through a validity mask.  The pending per-channel affine + LeakyReLU of the
// source is not applied in the load either: its scale/shift vectors are fetched alongside
// (identity constants when the source has none) and applied when the register set is
// written to LDS, one K-chunk later.  This keeps every load of a chunk in flight at once -
// hipcc otherwise wraps each conditional load in an exec-mask branch followed by
// `s_waitcnt vmcnt(0)`, which serialises a full memory latency per tile row.
// Offsets are 32-bit element offsets (tensors < 2^32 elements).
// ---------------------------------------------------------------------------
__device__ float k_ones4[4] = {1.f, 1.f, 1.f, 1.f};
__device__ float k_zeros4[4] = {0.f, 0.f, 0.f, 0.f};

struct Aff { float4 sc, sh; float slope; };

// scalar (VEC = 1) element of the virtual tensor, branch-free like the vector path
__device__ __forceinline__ float load1_bf(const ConvGeom& g, const Src& S, bool ok, int n, int hi, int wi, int c) {
  const int Hs = g.H >> S.up, Ws = g.W >> S.up;
  unsigned row;
  if (S.gidx) row = (unsigned)S.gidx[ok ? n : 0];
  else row = (unsigned)((n * Hs + (hi >> S.up)) * Ws + (wi >> S.up));
  const float v = S.p[ok ? row * (unsigned)S.ld + (unsigned)c : 0u];
  const bool has = ok && S.scale != nullptr;
  const float sc = *(has ? S.scale + c : k_ones4), sh = *(has ? S.shift + c : k_zeros4);
  const float r = leaky(fmaf(v, sc, sh), has ? S.slope : 1.f);
  return ok ? r : 0.f;
}

// leaky for 0 <= slope <= 1 (check_desc enforces it for source slopes): max(v, v * slope) - two VALU
// instead of compare + multiply + select, bit-identical (v > 0: v * slope <= v; v < 0: v * slope >= v)
__device__ __forceinline__ float leaky01(float v, float slope) { return __builtin_fmaxf(v, v * slope); }

// Pending affine + LeakyReLU + validity mask of one float4, on the packed-fp32 VALU (v_pk_fma_f32 /
// v_pk_mul_f32: two lanes of work per instruction): 2 fma + 2 mul + 4 max + 2 mul(mask) instead of
// 4 + 4 + 4 + 4 selects.  The mask is a multiplication by 0 / 1 (the masked-off load fetched element 0
// of its tensor, finite).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 apply_aff(float4 v, const Aff& a, bool ok) {
  const f32x2 mk = ok ? f32x2{1.f, 1.f} : f32x2{0.f, 0.f};
  const f32x2 sl = {a.slope, a.slope};
  f32x2 lo = f32x2{v.x, v.y} * f32x2{a.sc.x, a.sc.y} + f32x2{a.sh.x, a.sh.y};
  f32x2 hi = f32x2{v.z, v.w} * f32x2{a.sc.z, a.sc.w} + f32x2{a.sh.z, a.sh.w};
  const f32x2 lo2 = lo * sl, hi2 = hi * sl;
  lo = f32x2{__builtin_fmaxf(lo.x, lo2.x), __builtin_fmaxf(lo.y, lo2.y)} * mk;
  hi = f32x2{__builtin_fmaxf(hi.x, hi2.x), __builtin_fmaxf(hi.y, hi2.y)} * mk;
  return make_float4(lo.x, lo.y, hi.x, hi.y);
}

// 16 bytes through a raw buffer resource: an offset at or beyond the resource's size reads zeros - rows
// that are out of range need no select afterwards (operands without a pending affine: dY).  kOobByte is
// out of range for every tensor below 2 GiB (the entry points check).  The descriptor is built by hand
// and the load is the LLVM intrinsic itself: this ROCm's __builtin_amdgcn_raw_buffer_load_b128 lowers to
// a ONE-dword load whose value is splat over the four lanes (tools/_src/bufload_probe.hip).
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ f32x4 llvm_raw_buffer_load_f32x4(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
typedef i32x4 BufRsrc;
constexpr unsigned kOobByte = 0x80000000u;
__device__ __forceinline__ BufRsrc rsrc_of(const float* p, unsigned bytes) {
  const unsigned long long a = (unsigned long long)p;
  BufRsrc r;
  r.x = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
  r.y = __builtin_amdgcn_readfirstlane((int)(a >> 32));              // (stride 0: raw buffer)
  r.z = __builtin_amdgcn_readfirstlane((int)bytes);
  r.w = 0x00020000;                                                  // (32-bit data format, no swizzle)
  return r;
}
__device__ __forceinline__ float4 ld4_buf(BufRsrc r, unsigned byte_off) {
  const f32x4 v = llvm_raw_buffer_load_f32x4(r, (int)byte_off, 0, 0);
  return make_float4(v.x, v.y, v.z, v.w);
}

// 16 bytes at a 32-bit BYTE offset from a wave-uniform base: the address is formed by the load itself
// (global_load_dwordx4 v, v_off, s[base]) instead of a 64-bit add per lane (tensors < 4 GiB: check_desc)
__device__ __forceinline__ float4 ld4_off(const float* base, unsigned elem_off) {
  unsigned byte_off = elem_off << 2;
  asm volatile("" : "+v"(byte_off));     // (keeps hipcc from turning the select on the offset into a select of two 64-bit addresses)
  return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + (size_t)byte_off);
}

// bfloat16 storage (Src::bf, dY of a bf16-storage layer): FOUR bf16 = 8 bytes per lane, returned in .x / .y of the same
// float4 register quad the fp32 loaders fill (unpack_bf16x4 at staging time, igemm.h); element offsets as above
typedef float f32x2g __attribute__((ext_vector_type(2)));
__device__ f32x2g llvm_raw_buffer_load_f32x2(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v2f32");
__device__ __forceinline__ float4 ld2h_buf(BufRsrc r, unsigned byte_off) {
  const f32x2g v = llvm_raw_buffer_load_f32x2(r, (int)byte_off, 0, 0);
  return make_float4(v.x, v.y, 0.f, 0.f);
}
__device__ __forceinline__ float4 ld2h_off(const float* base, unsigned elem_off) {
  unsigned byte_off = elem_off << 1;
  asm volatile("" : "+v"(byte_off));
  const f32x2g v = *reinterpret_cast<const f32x2g*>(reinterpret_cast<const char*>(base) + (size_t)byte_off);
  return make_float4(v.x, v.y, 0.f, 0.f);
}

__device__ __forceinline__ void fetch_aff(Aff& a, const Src& S, int c, bool cok) {
  const bool has = S.scale != nullptr && cok;
  const float* scp = has ? S.scale + c : k_ones4;
  const float* shp = has ? S.shift + c : k_zeros4;
  a.sc = *reinterpret_cast<const float4*>(scp);
  a.sh = *reinterpret_cast<const float4*>(shp);
  a.slope = has ? S.slope : 1.f;
}

template <int NVA, int NVB> struct RegSet {
  float4 a[NVA], b[NVB];
  Aff aff;
  unsigned ma, mb;          // validity bits of the A / B rows
};

// ---------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------
// H: bf16 operand path (igemm.h): operands rounded to bf16 on their way into LDS, v_mfma_f32_32x32x16_bf16
// ST: the epilogue also reduces the tile's BatchNorm statistics (launches without split-K only)
template <int BM, int BN, int VEC, bool GATHER, bool H = false, bool ST = false>
__global__ __launch_bounds__(NTHREADS) void conv_fwd_kernel(const FwdParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NVA = BM / 32, NVB = BN / 32;
  constexpr int AF = TileBytes<H, BM, false>::value / 4, BF = TileBytes<H, BN, false>::value / 4;   // (in floats)
  const ConvGeom& g = p.g;
  const int tid = threadIdx.x;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN, split = blockIdx.z;
  const int per = (p.iters + p.e.nsplit - 1) / p.e.nsplit;
  const int it_begin = split * per;
  const int it_end = min(p.iters, it_begin + per);
  const int col4 = tid & 7, r0 = tid >> 3;
  const int ldw = g.KH * g.KW * g.Wtap;
  const int Ktot = g.KH * g.KW * g.Ctot;

  int rn[NVA], rhb[NVA], rwb[NVA];
  unsigned wrow[NVB], bmask = 0;
  {
    const int HoWo = g.Ho * g.Wo;
    #pragma unroll
    for (int i = 0; i < NVA; ++i) {
      const int m = m0 + r0 + 32 * i;
      if (m < p.M) {
        const int n = m / HoWo, rem = m - n * HoWo;
        const int ho = rem / g.Wo, wo = rem - ho * g.Wo;
        rn[i] = n; rhb[i] = ho * g.stride - g.pad; rwb[i] = wo * g.stride - g.pad;
      } else { rn[i] = -1; rhb[i] = 0; rwb[i] = 0; }
    }
    #pragma unroll
    for (int i = 0; i < NVB; ++i) {
      const int n = n0 + r0 + 32 * i;
      const bool ok = n < p.Cout;
      wrow[i] = ok ? (unsigned)n * (unsigned)ldw : 0u;
      bmask |= (ok ? 1u : 0u) << i;
    }
  }

  // Wave-uniform cursor over the reduction order (tap, source, channel chunk).  The loader is
  // called with consecutive chunk indices, so the cursor is decoded once and then ADVANCED -
  // a per-chunk decode (two integer divisions, a four-way source select of eight fields) was
  // ~120 scalar instructions and ~20 branches per chunk in front of the loads.
  int q_tap = 0, q_kh = 0, q_kw = 0, q_s = 0, q_cstart = 0, q_cb = 0;
  static_assert(offsetof(FwdParams, g) == 0 && offsetof(ConvGeom, s0) == 0, "kernarg_src layout");
  Src q_S = kernarg_src(0);
  if (VEC == 4 && it_begin < it_end) {
    // (reduction order: tap outer, (source, channel chunk) inner.  Tap INNER - the taps re-reading one
    // 32-channel slab back to back - was measured in round 2: fewer bytes fetched, not faster.)
    q_tap = it_begin / p.nch;
    locate_chunk(g, it_begin - q_tap * p.nch, q_s, q_cstart, q_cb);
    q_S = kernarg_src(q_s);
    q_kh = q_tap / g.KW; q_kw = q_tap - q_kh * g.KW;
  }

  typedef RegSet<NVA, NVB> RS;
  RS rs0;
  unsigned roff[NVA], rmask = 0;      // per A row: element offset of its pixel for the current (tap, source), validity
  bool q_dirty = true;
  auto load_into = [&](int it, RS& r) {
    if (VEC == 4) {
      const int tap = q_tap, kh = q_kh, kw = q_kw, cstart = q_cstart, cb = q_cb;
      const Src S = q_S;
      bool next_dirty;
      {
        // branch-free advance (scalar selects; the source block is re-read with scalar loads
        // every chunk) so that the loop body stays one basic block.  The last chunk is
        // re-issued instead of advancing past the end.
        const bool adv = it + 1 < it_end;
        const int ncb = q_cb + BK;
        const bool wrap_s = adv && ncb >= q_S.C;                 // next source
        const bool wrap_t = wrap_s && q_s + 1 == g.nsrc;         // next tap
        const bool wrap_w = wrap_t && q_kw + 1 == g.KW;          // next kernel row
        q_cb = wrap_s ? 0 : (adv ? ncb : q_cb);
        q_cstart = wrap_t ? 0 : (wrap_s ? q_cstart + q_S.C : q_cstart);
        q_s = wrap_t ? 0 : (wrap_s ? q_s + 1 : q_s);
        q_tap += wrap_t ? 1 : 0;
        q_kw = wrap_w ? 0 : (wrap_t ? q_kw + 1 : q_kw);
        q_kh += wrap_w ? 1 : 0;
        next_dirty = wrap_s;                                     // (the source or the tap moves on)
        if (next_dirty) q_S = kernarg_src(q_s);                  // (scalar loads of the source block)
      }
      const int c = cb + 4 * col4;
      const bool cok = c < S.C;
      fetch_aff(r.aff, S, c, cok);
      // Row geometry (bounds test, pixel offset) depends on the tap and the source only: recomputed
      // when either changed (wave-uniform branch - every nch-th chunk in the tap-outer order), otherwise
      // a row's offset just moves on by the channel chunk.  This arithmetic sits in FRONT of the
      // chunk's global loads, i.e. it is not covered by the wave's own MFMA block.
      if (q_dirty) {
        const int Hs = g.H >> S.up, Ws = g.W >> S.up;
        rmask = 0;
        #pragma unroll
        for (int i = 0; i < NVA; ++i) {
          const int hi = rhb[i] + kh, wi = rwb[i] + kw;
          const bool ok = rn[i] >= 0 && (unsigned)hi < (unsigned)g.H && (unsigned)wi < (unsigned)g.W;
          rmask |= (ok ? 1u : 0u) << i;
          unsigned pixel;
          if (GATHER) pixel = S.gidx ? (unsigned)S.gidx[ok ? rn[i] : 0] : (unsigned)(ok ? rn[i] : 0);
          else pixel = ok ? (unsigned)((rn[i] * Hs + (hi >> S.up)) * Ws + (wi >> S.up)) : 0u;
          roff[i] = pixel * (unsigned)S.ld;
        }
      }
      q_dirty = next_dirty;
      const unsigned ma = cok ? rmask : 0u;
      #pragma unroll
      for (int i = 0; i < NVA; ++i) {
        const unsigned off = ((ma >> i & 1u) ? roff[i] + (unsigned)c : 0u);
        r.a[i] = ld4_off(S.p, off);
      }
      const unsigned wcol = (unsigned)(tap * g.Wtap + cstart + c);
      const unsigned mb = cok ? bmask : 0u;
      #pragma unroll
      for (int i = 0; i < NVB; ++i) {
        const unsigned off = ((mb >> i & 1u) ? wrow[i] + wcol : 0u);
        r.b[i] = ld4_off(p.Wt, off);
      }
      r.ma = ma; r.mb = mb;
    } else {
      float av[NVA][4], bv[NVB][4];
      #pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = it * BK + 4 * col4 + j;
        const bool kok = k < Ktot;
        const int kk = kok ? k : 0;
        const int tap = kk / g.Ctot, c = kk - tap * g.Ctot;
        int s, cs;
        locate_channel(g, c, s, cs);
        const Src S = pick_src(g, s);
        const int kh = tap / g.KW, kw = tap - kh * g.KW;
        #pragma unroll
        for (int i = 0; i < NVA; ++i) {
          const int hi = rhb[i] + kh, wi = rwb[i] + kw;
          const bool ok = kok && rn[i] >= 0 && (unsigned)hi < (unsigned)g.H && (unsigned)wi < (unsigned)g.W;
          av[i][j] = load1_bf(g, S, ok, rn[i], hi, wi, cs);
        }
        #pragma unroll
        for (int i = 0; i < NVB; ++i) {
          const int n = n0 + r0 + 32 * i;
          const bool wok = kok && n < p.Cout;
          const float wv = p.Wt[wok ? (unsigned)n * (unsigned)ldw + (unsigned)(tap * g.Wtap + c) : 0u];
          bv[i][j] = wok ? wv : 0.f;
        }
      }
      #pragma unroll
      for (int i = 0; i < NVA; ++i) r.a[i] = make_float4(av[i][0], av[i][1], av[i][2], av[i][3]);
      #pragma unroll
      for (int i = 0; i < NVB; ++i) r.b[i] = make_float4(bv[i][0], bv[i][1], bv[i][2], bv[i][3]);
      r.aff.sc = make_float4(1.f, 1.f, 1.f, 1.f); r.aff.sh = zero4(); r.aff.slope = 1.f;
      r.ma = ~0u; r.mb = ~0u;
    }
  };
  auto stage_from = [&](RS& r) {
    float4 ta[NVA], tb[NVB];
    #pragma unroll
    for (int i = 0; i < NVA; ++i) ta[i] = apply_aff(r.a[i], r.aff, (r.ma >> i & 1u) != 0);
    #pragma unroll
    // (no select on the weight operand: where its mask is off - output channel >= Cout, channel chunk
    // beyond the source - the load fetched W[0..3], finite, and either the A operand of the same k is
    // zero or the accumulator column is never stored)
    for (int i = 0; i < NVB; ++i) tb[i] = r.b[i];
    if constexpr (H) {
      store_tile_h<BM, false>(reinterpret_cast<bf16_t*>(smem), ta, tid);
      store_tile_h<BN, false>(reinterpret_cast<bf16_t*>(smem + AF), tb, tid);
    } else {
      store_tile<BM, false>(smem, ta, tid);
      store_tile<BN, false>(smem + AF, tb, tid);
    }
  };

  SG2IM_ZERO_ACC()
  int wm0, wn0, lane;
  wave_origin<BM, BN>(tid, wm0, wn0, lane);

  auto do_load = [&](int it) { load_into(it, rs0); };
  auto do_stage = [&](bool) { stage_from(rs0); };
  auto do_mma = [&](int phase) {
    if constexpr (H) {
      if (phase == 0) read_frags_h<BM, BN, false, false>(reinterpret_cast<const bf16_t*>(smem),
                                                         reinterpret_cast<const bf16_t*>(smem + AF), wm0, wn0, lane, fragsh);
      else mma_frags_h<BM, BN>(fragsh, acc);
    } else {
      if (phase == 0) read_frags<BM, BN, false, false>(smem, smem + AF, wm0, wn0, lane, frags);
      else mma_frags<BM, BN>(frags, acc);
    }
  };
  k_pipeline(it_begin, it_end, do_load, do_stage, do_mma);
  epilogue<BM, BN>(p.e, p.M, p.Cout, p.Cout, m0, n0, wm0, wn0, lane, split, acc);
  if constexpr (ST) {
    if (p.e.nsplit == 1) epilogue_stats<BM, BN>(p.e, p.st, p.M, p.Cout, m0, n0, wm0, wn0, lane, tid, blockIdx.y, acc, smem);
  }
}

// ---------------------------------------------------------------------------
// data gradient (transposed convolution of dY with the same weights)
// VA: vector width of the dY loads (4 needs Cout % 4 == 0), VB: of the weight loads
// (4 needs Ctot, c_begin, c_count % 4 == 0).  VA == 1 implies the flat-K enumeration.
//
// parity mode (stride 2): an input pixel (h, w) only sees the taps with
// kh = (h + pad) mod 2 (mod 2) - the other taps hit no output pixel.  The rows are
// therefore split into the 4 classes (h mod 2, w mod 2), blockIdx.z = class, and each
// class runs a dense stride-1 problem over its KH/2 x KW/2 live taps: 4x fewer MFMAs
// than sweeping all taps with zero operands.
// ---------------------------------------------------------------------------
struct ParityRow {
  int H, W, Hc, Wc, ph, pw;
  __device__ __forceinline__ long long operator()(int m) const {
    const int hw = Hc * Wc;
    const int n = m / hw, rem = m - n * hw;
    const int hp = rem / Wc, wp = rem - hp * Wc;
    return ((long long)n * H + 2 * hp + ph) * W + 2 * wp + pw;
  }
};

template <int BM, int BN, int VA, int VB, bool H = false, bool ST = false>
__global__ __launch_bounds__(NTHREADS) void conv_dgrad_kernel(const DgradParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NVA = BM / 32, NVB = BN / 32;
  constexpr int AF = TileBytes<H, BM, false>::value / 4, BF = TileBytes<H, BN, true>::value / 4;
  const ConvGeom& g = p.g;
  const int tid = threadIdx.x;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int col4 = tid & 7, r0 = tid >> 3;
  const int taps = g.KH * g.KW;
  const int ldw = taps * g.Wtap;
  const int Cout = g.s0.C, ldy = g.s0.ld;
  const float* dY = g.s0.p;

  // row space, live taps and K range of this workgroup
  int split = 0, ph = 0, pw = 0, Hc = g.H, Wc = g.W, kh0 = 0, kw0 = 0, nkw = g.KW, kstep = 1;
  int M = p.M, iters = p.iters, nch = p.nch;
  int cls = 0;
  if (p.parity) {
    cls = blockIdx.z & 3;                 // grid.z = 4 classes x nsplit
    split = blockIdx.z >> 2;
    ph = cls >> 1; pw = cls & 1;
    Hc = (g.H - ph + 1) >> 1; Wc = (g.W - pw + 1) >> 1;
    M = g.NB * Hc * Wc;
    kh0 = (ph + g.pad) & 1; kw0 = (pw + g.pad) & 1; kstep = 2;
    const int nkh = (g.KH - kh0 + 1) >> 1;
    nkw = (g.KW - kw0 + 1) >> 1;
    nch = (Cout + BK - 1) / BK;
    iters = nkh * nkw * nch;
    if (m0 >= M) return;
  } else {
    split = blockIdx.z;
  }
  const int per = (iters + p.e.nsplit - 1) / p.e.nsplit;
  const int it_begin = split * per;
  const int it_end = min(iters, it_begin + per);
  const int Ktot = taps * Cout;

  // A rows are *input* pixels (n, h, w); stored pre-shifted by the padding
  int rn[NVA], rh[NVA], rw[NVA];
  {
    const int HW = Hc * Wc;
    #pragma unroll
    for (int i = 0; i < NVA; ++i) {
      const int m = m0 + r0 + 32 * i;
      if (m < M) {
        const int n = m / HW, rem = m - n * HW;
        const int hp = rem / Wc, wp = rem - hp * Wc;
        rn[i] = n; rh[i] = hp * kstep + ph + g.pad; rw[i] = wp * kstep + pw + g.pad;
      } else { rn[i] = -1; rh[i] = 0; rw[i] = 0; }
    }
  }
  constexpr int Q = BN / 4;
  const int bcol4 = tid % Q, bk0 = tid / Q;
  const int nn = n0 + 4 * bcol4;
  const bool nok4 = nn < p.Nc;

  // output pixel hit by input row i through tap (kh, kw): element row index or invalid
  auto out_pixel = [&](int i, int kh, int kw, unsigned& row) -> bool {
    const int nh = rh[i] - kh, nw = rw[i] - kw;      // = ho*stride, wo*stride
    int ho = nh, wo = nw;
    bool ok = rn[i] >= 0 && nh >= 0 && nw >= 0;
    if (g.stride != 1) {
      ho = nh / g.stride; wo = nw / g.stride;
      ok = ok && ho * g.stride == nh && wo * g.stride == nw;
    }
    ok = ok && ho < g.Ho && wo < g.Wo;
    row = ok ? (unsigned)((rn[i] * g.Ho + ho) * g.Wo + wo) : 0u;
    return ok;
  };

  typedef RegSet<NVA, NVB> RS;
  RS rs0;
  // wave-uniform cursor over (live tap, output-channel chunk), advanced chunk by chunk
  int q_th = 0, q_tw = 0, q_cb = 0;
  if (VA == 4 && it_begin < it_end) {
    const int t = it_begin / nch;
    q_cb = (it_begin - t * nch) * BK;
    q_th = t / nkw; q_tw = t - q_th * nkw;
  }
  // (VA == 4) per A row: element offset of the output pixel the current tap hits, validity - recomputed
  // when the tap moves on (every nch-th chunk), see the forward kernel; dY is read through a buffer
  // resource, out-of-range rows come back as zeros
  unsigned droff[NVA], dmask = 0;
  bool d_dirty = true;
  const BufRsrc rsY = rsrc_of(dY, (unsigned)(g.NB * g.Ho * g.Wo) * (unsigned)ldy * 4u);
  auto load_into = [&](int it, RS& r) {
    if (VA == 4) {
      const int cb = q_cb;
      const int kh = kh0 + kstep * q_th, kw = kw0 + kstep * q_tw;
      const int tap = kh * g.KW + kw;
      if (d_dirty) {
        dmask = 0;
        #pragma unroll
        for (int i = 0; i < NVA; ++i) {
          unsigned row;
          const bool ok = out_pixel(i, kh, kw, row);
          dmask |= (ok ? 1u : 0u) << i;
          droff[i] = row * (unsigned)ldy;
        }
      }
      d_dirty = false;
      if (it + 1 < it_end) {
        q_cb += BK;
        if (q_cb >= Cout) { q_cb = 0; d_dirty = true; if (++q_tw == nkw) { q_tw = 0; ++q_th; } }
      }
      const int co = cb + 4 * col4;
      const bool cok = co < Cout;
      const unsigned ma = cok ? dmask : 0u;
      #pragma unroll
      for (int i = 0; i < NVA; ++i)
        r.a[i] = ld4_buf(rsY, (ma >> i & 1u) ? (droff[i] + (unsigned)co) << 2 : kOobByte);
      const unsigned wcol = (unsigned)(tap * g.Wtap + p.c_begin);
      unsigned mb = 0;
      #pragma unroll
      for (int i = 0; i < NVB; ++i) {
        const int cok2 = cb + bk0 + (1024 / BN) * i;
        const bool rok = cok2 < Cout;
        const unsigned base = rok ? (unsigned)cok2 * (unsigned)ldw + wcol : 0u;
        if (VB == 4) {
          const bool ok = rok && nok4;
          mb |= (ok ? 1u : 0u) << i;
          r.b[i] = ld4_off(p.Wt, ok ? base + (unsigned)nn : 0u);
        } else {
          float bv[4];
          #pragma unroll
          for (int j = 0; j < 4; ++j) {
            const bool ok = rok && nn + j < p.Nc;
            const float v = p.Wt[ok ? base + (unsigned)(nn + j) : 0u];
            bv[j] = ok ? v : 0.f;
          }
          mb |= 1u << i;
          r.b[i] = make_float4(bv[0], bv[1], bv[2], bv[3]);
        }
      }
      r.ma = ma; r.mb = mb;
    } else {
      float av[NVA][4];
      #pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = it * BK + 4 * col4 + j;
        const bool kok = k < Ktot;
        const int tap = kok ? k / Cout : 0, co = kok ? k - tap * Cout : 0;
        const int kh = tap / g.KW, kw = tap - kh * g.KW;
        #pragma unroll
        for (int i = 0; i < NVA; ++i) {
          unsigned row;
          const bool ok = out_pixel(i, kh, kw, row) && kok;
          const float v = dY[ok ? row * (unsigned)ldy + (unsigned)co : 0u];
          av[i][j] = ok ? v : 0.f;
        }
      }
      #pragma unroll
      for (int i = 0; i < NVA; ++i) r.a[i] = make_float4(av[i][0], av[i][1], av[i][2], av[i][3]);
      #pragma unroll
      for (int i = 0; i < NVB; ++i) {
        const int k = it * BK + bk0 + (1024 / BN) * i;
        const bool kok = k < Ktot;
        const int tap = kok ? k / Cout : 0, co = kok ? k - tap * Cout : 0;
        const unsigned base = (unsigned)co * (unsigned)ldw + (unsigned)(tap * g.Wtap + p.c_begin);
        float bv[4];
        #pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool ok = kok && nn + j < p.Nc;
          const float v = p.Wt[ok ? base + (unsigned)(nn + j) : 0u];
          bv[j] = ok ? v : 0.f;
        }
        r.b[i] = make_float4(bv[0], bv[1], bv[2], bv[3]);
      }
      r.ma = ~0u; r.mb = ~0u;
    }
  };
  auto stage_from = [&](RS& r) {
    float4 ta[NVA], tb[NVB];
    #pragma unroll
    for (int i = 0; i < NVA; ++i) ta[i] = r.a[i];      // (VA == 4: invalid rows were read as zeros; VA == 1: zeroed in the loader)
    #pragma unroll
    // (no select on the weight operand, as in the forward kernel: dY is zero for k >= Cout and columns
    // >= Nc are never stored)
    for (int i = 0; i < NVB; ++i) tb[i] = r.b[i];
    if constexpr (H) {
      store_tile_h<BM, false>(reinterpret_cast<bf16_t*>(smem), ta, tid);
      store_tile_h<BN, true>(reinterpret_cast<bf16_t*>(smem + AF), tb, tid);
    } else {
      store_tile<BM, false>(smem, ta, tid);
      store_tile<BN, true>(smem + AF, tb, tid);
    }
  };

  SG2IM_ZERO_ACC()
  int wm0, wn0, lane;
  wave_origin<BM, BN>(tid, wm0, wn0, lane);

  auto do_load = [&](int it) { load_into(it, rs0); };
  auto do_stage = [&](bool) { stage_from(rs0); };
  auto do_mma = [&](int phase) {
    if constexpr (H) {
      if (phase == 0) read_frags_h<BM, BN, false, true>(reinterpret_cast<const bf16_t*>(smem),
                                                        reinterpret_cast<const bf16_t*>(smem + AF), wm0, wn0, lane, fragsh);
      else mma_frags_h<BM, BN>(fragsh, acc);
    } else {
      if (phase == 0) read_frags<BM, BN, false, true>(smem, smem + AF, wm0, wn0, lane, frags);
      else mma_frags<BM, BN>(frags, acc);
    }
  };
  k_pipeline(it_begin, it_end, do_load, do_stage, do_mma);
  if (p.parity) {
    // split-K slabs of the parity form: [split][class][p.M rows][Nc], finished (and mapped to the
    // interleaved destination rows) by splitk_finish_parity_kernel
    Epi e2 = p.e;
    if (e2.nsplit > 1) e2.ws += ((size_t)split * 4 + cls) * (size_t)p.M * p.Nc;
    epilogue<BM, BN>(e2, M, p.Nc, p.Nc, m0, n0, wm0, wn0, lane, 0, acc, ParityRow{g.H, g.W, Hc, Wc, ph, pw});
  } else {
    if (!ST && p.e.mask != nullptr && p.e.nsplit == 1)      // (workgroup-uniform: sg2im_conv2d_backward_data_act)
      epilogue<BM, BN, IdentityRow, false, true>(p.e, M, p.Nc, p.Nc, m0, n0, wm0, wn0, lane, split, acc);
    else
      epilogue<BM, BN>(p.e, M, p.Nc, p.Nc, m0, n0, wm0, wn0, lane, split, acc);
    if constexpr (ST) {
      if (p.e.nsplit == 1) epilogue_bnbwd<BM, BN>(p.st, M, p.Nc, m0, n0, wm0, wn0, lane, tid, blockIdx.y, acc, smem);
    }
  }
}

// ---------------------------------------------------------------------------
// weight gradient
// ---------------------------------------------------------------------------
// (the body is a device function of (params, block coordinates) so that the grouped launch below can run it
// for one of several problems)
// FR ("fast rows", VEC == 4, no gathers): stride 1 and Wo % BK == 0, so the BK output pixels of a K chunk lie
// in ONE output row - the chunk's (n, ho) are wave-uniform and a B row is (chunk's first column + its fixed
// k): no per-row coordinate state, ~6 instead of ~20 VALU per row and chunk.
template <int BM, int BN, int VEC, bool GATHER, bool H, bool FR = false>
__device__ __forceinline__ void conv_wgrad_body(const WgradParams& p, const int bidx, const int bidy, const int bidz) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NVA = BM / 32, NVB = BN / 32;
  constexpr int AF = TileBytes<H, BM, true>::value / 4, BF = TileBytes<H, BN, true>::value / 4;
  const ConvGeom& g = p.g;
  const int tid = threadIdx.x;
  // (an XCD-pinned 1-D grid - every column tile of a reduction slice on one XCD's L2 - was
  // measured: no gain on the large layers, so the plain 3-D grid stays)
  const int ntile_x = bidx, mtile = bidy, split = bidz;
  const int m0 = mtile * BM;
  const int per = (p.iters + p.e.nsplit - 1) / p.e.nsplit;
  const int it_begin = split * per;
  const int it_end = min(p.iters, it_begin + per);
  const int taps = g.KH * g.KW;
  const int Ntot = taps * g.Ctot;
  const int HoWo = g.Ho * g.Wo;

  // Column tiles run over the FLAT (tap, channel) axis of dW, so a tile may straddle taps;
  // a thread's float4 never does (Ctot % 4 == 0) and keeps its own (tap, source, channel).
  const int n0 = ntile_x * BN;
  constexpr int QA = BM / 4, QB = BN / 4;
  const int acol4 = tid % QA, ak0 = tid / QA;
  const int bcol4 = tid % QB, bk0 = tid / QB;
  const int aco = m0 + 4 * acol4;
  const bool aok = aco < p.Cout;

  // per-thread B column(s): fixed for the whole reduction
  int bs = 0, bcs = 0;
  const int bcol = n0 + 4 * bcol4;
  const bool bok = (VEC == 4) && bcol < Ntot;
  const int btap = bok ? bcol / g.Ctot : 0;
  locate_channel(g, bok ? bcol - btap * g.Ctot : 0, bs, bcs);
  const Src BS = pick_src(g, bs);
  const int bkh = btap / g.KW, bkw = btap - bkh * g.KW;
  const int Hs = g.H >> BS.up, Ws = g.W >> BS.up;
  Aff baff;
  fetch_aff(baff, BS, bcs, bok);
  int js[4], jcs[4], jkh[4], jkw[4];
  bool jok[4];
  if (VEC != 4) {
    #pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + 4 * bcol4 + j;
      jok[j] = n < Ntot;
      const int tap = jok[j] ? n / g.Ctot : 0, c = jok[j] ? n - tap * g.Ctot : 0;
      locate_channel(g, c, js[j], jcs[j]);
      jkh[j] = tap / g.KW; jkw[j] = tap - jkh[j] * g.KW;
    }
  }
  // the B rows are output pixels: (n, ho, wo) of row pix = it*BK + bk0 + (1024/BN)*i is
  // advanced incrementally (BK pixels per chunk) instead of two integer divisions per row
  // (kept pre-multiplied: hb = ho * stride - pad, wb = wo * stride - pad; the wrap tests compare against the
  // equally transformed limits, so a chunk costs no multiplication per row)
  int bn_[NVB], bhb[NVB], bwb[NVB];
  const int dn = BK / HoWo, drem = BK - dn * HoWo, dh = drem / g.Wo, dw = drem - dh * g.Wo;
  const int dhs = dh * g.stride, dws = dw * g.stride, WoS = g.Wo * g.stride, HoS = g.Ho * g.stride;
  const int wlim = WoS - g.pad, hlim = HoS - g.pad;
  // FR: wave-uniform cursor of the chunk's first pixel, per-row constants
  int c_n = 0, c_ho = 0, c_wo = 0, kx[NVB];
  const int dkh = bkh - g.pad;
  #pragma unroll
  for (int i = 0; i < NVB; ++i) {
    const int pix = it_begin * BK + bk0 + (1024 / BN) * i;
    bn_[i] = pix / HoWo;
    const int rem = pix - bn_[i] * HoWo;
    const int ho = rem / g.Wo, wo = rem - ho * g.Wo;
    bhb[i] = ho * g.stride - g.pad; bwb[i] = wo * g.stride - g.pad;
    kx[i] = bk0 + (1024 / BN) * i + bkw - g.pad;
  }
  if (FR) {
    const int pix0 = it_begin * BK;
    c_n = pix0 / HoWo;
    const int rem = pix0 - c_n * HoWo;
    c_ho = rem / g.Wo; c_wo = rem - c_ho * g.Wo;
  }

  typedef RegSet<NVA, NVB> RS;
  RS rs0;
  const BufRsrc rsY = rsrc_of(p.dY, (unsigned)p.P * (unsigned)p.ldy * 4u);
  auto load_into = [&](int it, RS& r) {
    unsigned ma = 0, mb = 0;
    #pragma unroll
    for (int i = 0; i < NVA; ++i) {
      const int pix = it * BK + ak0 + (1024 / BM) * i;
      if (VEC == 4) {
        const bool ok = aok && pix < p.P;
        ma |= (ok ? 1u : 0u) << i;
        r.a[i] = ld4_buf(rsY, ok ? ((unsigned)pix * (unsigned)p.ldy + (unsigned)aco) << 2 : kOobByte);   // (zeros when !ok)
      } else {
        float v[4];
        #pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool ok = pix < p.P && aco + j < p.Cout;
          const float t = p.dY[ok ? (unsigned)pix * (unsigned)p.ldy + (unsigned)(aco + j) : 0u];
          v[j] = ok ? t : 0.f;
        }
        r.a[i] = make_float4(v[0], v[1], v[2], v[3]);
        ma |= 1u << i;
      }
    }
    // (the pipelines re-issue the last chunk instead of branching around the tail: the row coordinates
    // are not advanced past it, so pix < P keeps implying n < NB)
    const bool adv = it + 1 < it_end;
    const int a_dn = adv ? dn : 0, a_dhs = adv ? dhs : 0, a_dws = adv ? dws : 0;
    if (FR && VEC == 4 && !GATHER) {
      const int hv = c_ho + dkh;
      const bool okh = bok && it * BK < p.P && (unsigned)hv < (unsigned)g.H;     // (P % BK == 0: the whole chunk)
      const unsigned rowbase = (unsigned)((c_n * Hs + (hv >> BS.up)) * Ws);
      #pragma unroll
      for (int i = 0; i < NVB; ++i) {
        const int wi = c_wo + kx[i];
        const bool ok = okh && (unsigned)wi < (unsigned)g.W;
        mb |= (ok ? 1u : 0u) << i;
        r.b[i] = ld4_off(BS.p, ok ? (rowbase + (unsigned)(wi >> BS.up)) * (unsigned)BS.ld + (unsigned)bcs : 0u);
      }
      if (adv) {                                     // (wave-uniform: scalar unit)
        c_wo += BK;
        if (c_wo >= g.Wo) { c_wo = 0; if (++c_ho >= g.Ho) { c_ho = 0; ++c_n; } }
      }
    } else
    #pragma unroll
    for (int i = 0; i < NVB; ++i) {
      const int pix = it * BK + bk0 + (1024 / BN) * i;
      const int n = bn_[i], hb = bhb[i], wb = bwb[i];
      if (VEC == 4) {
        const int hi = hb + bkh, wi = wb + bkw;
        const bool ok = bok && pix < p.P && (unsigned)hi < (unsigned)g.H && (unsigned)wi < (unsigned)g.W;
        mb |= (ok ? 1u : 0u) << i;
        unsigned row;
        if (GATHER) row = BS.gidx ? (unsigned)BS.gidx[ok ? n : 0] : (unsigned)(ok ? n : 0);
        else row = ok ? (unsigned)((n * Hs + (hi >> BS.up)) * Ws + (wi >> BS.up)) : 0u;
        r.b[i] = ld4_off(BS.p, ok ? row * (unsigned)BS.ld + (unsigned)bcs : 0u);
      } else {
        float e[4];
        #pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int hi = hb + jkh[j], wi = wb + jkw[j];
          const bool ok = pix < p.P && jok[j] && (unsigned)hi < (unsigned)g.H && (unsigned)wi < (unsigned)g.W;
          e[j] = load1_bf(g, pick_src(g, js[j]), ok, n, hi, wi, jcs[j]);
        }
        r.b[i] = make_float4(e[0], e[1], e[2], e[3]);
        mb |= 1u << i;
      }
      // advance this row by BK pixels for the next chunk
      int w2 = wb + a_dws, h2 = hb + a_dhs, n2 = n + a_dn;
      if (w2 >= wlim) { w2 -= WoS; h2 += g.stride; }
      if (h2 >= hlim) { h2 -= HoS; ++n2; }
      bwb[i] = w2; bhb[i] = h2; bn_[i] = n2;
    }
    r.ma = ma; r.mb = mb;
  };
  // bias gradient: this thread's dY values all belong to output channels aco..aco+3
  const bool want_db = p.dbias != nullptr && ntile_x == 0;
  float4 dbs = zero4();
  auto stage_from = [&](RS& r, bool live) {
    float4 ta[NVA], tb[NVB];
    #pragma unroll
    for (int i = 0; i < NVA; ++i) ta[i] = r.a[i];        // (rows beyond the last pixel / channel were read as zeros)
    if (want_db && live) {
      #pragma unroll
      for (int i = 0; i < NVA; ++i) { dbs.x += ta[i].x; dbs.y += ta[i].y; dbs.z += ta[i].z; dbs.w += ta[i].w; }
    }
    #pragma unroll
    for (int i = 0; i < NVB; ++i) {
      if (VEC == 4) tb[i] = apply_aff(r.b[i], baff, (r.mb >> i & 1u) != 0);
      else tb[i] = r.b[i];
    }
    if constexpr (H) {
      store_tile_h<BM, true>(reinterpret_cast<bf16_t*>(smem), ta, tid);
      store_tile_h<BN, true>(reinterpret_cast<bf16_t*>(smem + AF), tb, tid);
    } else {
      store_tile<BM, true>(smem, ta, tid);
      store_tile<BN, true>(smem + AF, tb, tid);
    }
  };

  SG2IM_ZERO_ACC()
  int wm0, wn0, lane;
  wave_origin<BM, BN>(tid, wm0, wn0, lane);

  auto do_load = [&](int it) { load_into(it, rs0); };
  auto do_stage = [&](bool live) { stage_from(rs0, live); };
  auto do_mma = [&](int phase) {
    if constexpr (H) {
      if (phase == 0) read_frags_h<BM, BN, true, true>(reinterpret_cast<const bf16_t*>(smem),
                                                       reinterpret_cast<const bf16_t*>(smem + AF), wm0, wn0, lane, fragsh);
      else mma_frags_h<BM, BN>(fragsh, acc);
    } else {
      if (phase == 0) read_frags<BM, BN, true, true>(smem, smem + AF, wm0, wn0, lane, frags);
      else mma_frags<BM, BN>(frags, acc);
    }
  };
  k_pipeline(it_begin, it_end, do_load, do_stage, do_mma);
  epilogue<BM, BN>(p.e, p.Cout, Ntot, Ntot, m0, n0, wm0, wn0, lane, split, acc);
  // (workgroup-uniform condition: every wave must reach the barriers below)
  if (p.dbias != nullptr && bidx == 0) {
    // the 256 / QA threads that share a channel quad hold sums over disjoint pixel rows:
    // combine them through LDS in thread order (fixed order -> reproducible)
    constexpr int GROUPS = NTHREADS / QA;
    float4* red = reinterpret_cast<float4*>(smem);            // [GROUPS][QA]  (the operand image is free now)
    __syncthreads();                                          // (every wave is done with its last chunk)
    if (want_db) red[ak0 * QA + acol4] = dbs;
    __syncthreads();
    if (want_db && tid < QA && aco < p.Cout) {
      float4 t = red[tid];
      for (int k = 1; k < GROUPS; ++k) {
        const float4 u = red[k * QA + tid];
        t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
      }
      const float tv[4] = {t.x, t.y, t.z, t.w};
      float* dst = p.e.nsplit > 1 ? p.ws_bias + (size_t)split * p.Cout : p.dbias;
      for (int j = 0; j < 4; ++j) {
        if (aco + j >= p.Cout) break;
        if (p.e.nsplit > 1 || !p.e.accumulate) dst[aco + j] = tv[j];
        else dst[aco + j] += tv[j];
      }
    }
  }
}

template <int BM, int BN, int VEC, bool GATHER, bool H = false, bool FR = false>
__global__ __launch_bounds__(NTHREADS) void conv_wgrad_kernel(const WgradParams p) {
  conv_wgrad_body<BM, BN, VEC, GATHER, H, FR>(p, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Grouped launch: up to four independent weight-gradient problems (the four linear layers of a
// GraphTripleConv layer, whose weight gradients are leaves of the backward graph) as ONE grid - the
// dependent chain of small launches they sit in gets 1 launch + 1 finish instead of 4 + 4.  Workgroup b
// belongs to problem i with first[i] <= b < first[i + 1]; inside it the usual (column tile, row tile, split).
// The problem is selected by a static if-chain: indexing the by-value argument dynamically would demote
// it to scratch.  64x64 tiles, float4 loaders, row-gather capable sources.
constexpr int kGroupMax = 4;
struct WgradGroup { WgradParams p[kGroupMax]; int first[kGroupMax + 1]; };
__global__ __launch_bounds__(NTHREADS) void conv_wgrad_group_kernel(const WgradGroup g) {
  const int b = blockIdx.x;
#define SG2IM_GROUP_CASE(i)                                                        \
  if (b < g.first[i + 1]) {                                                        \
    const int l = b - g.first[i];                                                  \
    const int tiles = g.p[i].ntiles_n * g.p[i].ntiles_m;                           \
    const int sp = l / tiles, t = l - sp * tiles;                                  \
    const int ty = t / g.p[i].ntiles_n, tx = t - ty * g.p[i].ntiles_n;             \
    conv_wgrad_body<64, 64, 4, true, false>(g.p[i], tx, ty, sp);                   \
    return;                                                                        \
  }
  SG2IM_GROUP_CASE(0) SG2IM_GROUP_CASE(1) SG2IM_GROUP_CASE(2) SG2IM_GROUP_CASE(3)
#undef SG2IM_GROUP_CASE
}

// ---------------------------------------------------------------------------
// split-K finish: C = act(sum_s ws[s] + bias) (+ C)
// ---------------------------------------------------------------------------
// SL "split lanes" share one output: lane l adds splits l, l+SL, ... (ascending), the SL partials
// are then added in lane order - a fixed order, so the result does not depend on scheduling.
// SL > 1 is for tiny outputs reduced over many splits, where one thread per output would walk
// hundreds of dependent-latency loads.
__global__ void splitk_finish_kernel(const float* __restrict__ ws, int nsplit, long long MN, int N,
                                     float* __restrict__ C, long long ldc, const float* __restrict__ bias,
                                     float slope, int accumulate, int SL,
                                     const float* __restrict__ ws2, float* __restrict__ C2, int N2,
                                     int col_ctot, int col_wtap, const float* __restrict__ mask, long long ld_mask,
                                     float mask_slope) {
  // second, tiny reduction riding along (weight-gradient launches: the bias gradient partials)
  __shared__ float part[256];
  if (C2 != nullptr && blockIdx.x == 0) {
    constexpr int SLB = 8, PERB = 256 / SLB;      // 8 split lanes x 32 columns per round
    const int jl = threadIdx.x % PERB, sl2 = threadIdx.x / PERB;
    for (int base = 0; base < N2; base += PERB) {
      const int j = base + jl;
      float v = 0.f;
      if (j < N2)
        for (int s = sl2; s < nsplit; s += SLB) v += ws2[(size_t)s * N2 + j];
      part[threadIdx.x] = v;
      __syncthreads();
      if (sl2 == 0 && j < N2) {
        for (int l = 1; l < SLB; ++l) v += part[l * PERB + jl];
        C2[j] = accumulate ? C2[j] + v : v;
      }
      __syncthreads();
    }
  }
  const int per = 256 / SL;                       // outputs per block
  const int ol = threadIdx.x % per, sl = threadIdx.x / per;
  for (long long base = (long long)blockIdx.x * per; base < MN; base += (long long)gridDim.x * per) {
    const long long idx = base + ol;
    float v = 0.f;
    if (idx < MN) {
      // (eight loads in flight per thread: the loop is a chain of load latencies; the additions keep their order)
      #pragma unroll 8
      for (int s = sl; s < nsplit; s += SL) v += ws[(long long)s * MN + idx];
    }
    if (SL > 1) {
      __syncthreads();
      part[threadIdx.x] = v;
      __syncthreads();
      if (sl == 0) for (int l = 1; l < SL; ++l) v += part[l * per + ol];
    }
    if (sl == 0 && idx < MN) {
      const long long m = idx / N;
      const int n = (int)(idx - m * N);
      if (bias) v += bias[n];
      v = leaky(v, slope);
      if (mask) v *= mask[m * ld_mask + n] > 0.f ? 1.f : mask_slope;                          // (Epi::mask)
      float* dst = C + m * ldc + (col_wtap ? (n / col_ctot) * col_wtap + n % col_ctot : n);   // (Epi::col_*)
      if (accumulate) v += *dst;
      *dst = v;
    }
  }
}

// float4 form of the SL == 1 case (N, ldc multiples of 4, 16-byte aligned buffers): one thread
// per four adjacent outputs of a row; same ascending split order per output.
struct FinishArgs {
  const float* ws; int nsplit; long long MN; int N; float* C; long long ldc; const float* bias; float slope;
  int accumulate; const float* ws2; float* C2; int N2;
};
struct FinishMask { const float* act; long long ld; float slope; };        // Epi::mask for the finish kernels (act: may be null)
__device__ __forceinline__ void splitk_finish_v4_body(const float* __restrict__ ws, int nsplit, long long MN, int N,
                                                      float* __restrict__ C, long long ldc,
                                                      const float* __restrict__ bias, float slope, int accumulate,
                                                      const float* __restrict__ ws2, float* __restrict__ C2, int N2,
                                                      const int blk, const int nblk, const FinishMask fm = FinishMask{nullptr, 0, 1.f}) {
  if (C2 != nullptr && blk == 0) {
    __shared__ float part[256];
    constexpr int SLB = 8, PERB = 256 / SLB;
    const int jl = threadIdx.x % PERB, sl2 = threadIdx.x / PERB;
    for (int base = 0; base < N2; base += PERB) {
      const int j = base + jl;
      float v = 0.f;
      if (j < N2)
        for (int s = sl2; s < nsplit; s += SLB) v += ws2[(size_t)s * N2 + j];
      part[threadIdx.x] = v;
      __syncthreads();
      if (sl2 == 0 && j < N2) {
        for (int l = 1; l < SLB; ++l) v += part[l * PERB + jl];
        C2[j] = accumulate ? C2[j] + v : v;
      }
      __syncthreads();
    }
  }
  const long long Q = MN >> 2;
  const float4* __restrict__ w4 = reinterpret_cast<const float4*>(ws);
  for (long long q = (long long)blk * blockDim.x + threadIdx.x; q < Q; q += (long long)nblk * blockDim.x) {
    float4 v = w4[q];
    for (int s = 1; s < nsplit; ++s) {
      const float4 u = w4[(long long)s * Q + q];
      v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    const long long idx = q << 2;
    const long long m = idx / N;
    const int n = (int)(idx - m * N);
    if (bias) {
      const float4 b = *reinterpret_cast<const float4*>(bias + n);
      v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    }
    v.x = leaky(v.x, slope); v.y = leaky(v.y, slope); v.z = leaky(v.z, slope); v.w = leaky(v.w, slope);
    if (fm.act) {
      const float4 a = *reinterpret_cast<const float4*>(fm.act + m * fm.ld + n);
      v.x *= a.x > 0.f ? 1.f : fm.slope; v.y *= a.y > 0.f ? 1.f : fm.slope;
      v.z *= a.z > 0.f ? 1.f : fm.slope; v.w *= a.w > 0.f ? 1.f : fm.slope;
    }
    float4* dst = reinterpret_cast<float4*>(C + m * ldc + n);
    if (accumulate) { const float4 o = *dst; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
    *dst = v;
  }
}

__global__ void splitk_finish_v4_kernel(const float* __restrict__ ws, int nsplit, long long MN, int N,
                                        float* __restrict__ C, long long ldc, const float* __restrict__ bias,
                                        float slope, int accumulate,
                                        const float* __restrict__ ws2, float* __restrict__ C2, int N2, const FinishMask fm) {
  splitk_finish_v4_body(ws, nsplit, MN, N, C, ldc, bias, slope, accumulate, ws2, C2, N2, blockIdx.x, gridDim.x, fm);
}

// the finishes of a grouped launch (conv_wgrad_group_kernel) in one grid
struct FinishGroup { FinishArgs a[kGroupMax]; int first[kGroupMax + 1]; };
__global__ void splitk_finish_v4_group_kernel(const FinishGroup g) {
  const int b = blockIdx.x;
#define SG2IM_GROUP_CASE(i)                                                                                  \
  if (b < g.first[i + 1]) {                                                                                  \
    const FinishArgs& a = g.a[i];                                                                            \
    splitk_finish_v4_body(a.ws, a.nsplit, a.MN, a.N, a.C, a.ldc, a.bias, a.slope, a.accumulate, a.ws2, a.C2, \
                          a.N2, b - g.first[i], g.first[i + 1] - g.first[i]);                                \
    return;                                                                                                  \
  }
  SG2IM_GROUP_CASE(0) SG2IM_GROUP_CASE(1) SG2IM_GROUP_CASE(2) SG2IM_GROUP_CASE(3)
#undef SG2IM_GROUP_CASE
}

// Split-K finish that ALSO produces the BatchNorm tile partials of its output (see igemm.h epilogue_stats /
// epilogue_bnbwd - the launches with split-K have no finished values in their epilogue).  Workgroup
// (blockIdx.x, blockIdx.y) owns the rows [x * per, (x + 1) * per) of the column slab [128 y, 128 (y + 1)):
// thread -> (column quad tx, row lane ty), rows strided by TR, the row lanes of a quad combined through LDS in a
// fixed order.  Splits are added in ascending order, as in splitk_finish_v4_kernel.  MODE 1: forward statistics of C = leaky(sum + bias) as (pivot, sum d, sum d^2)
// with the block's first row as the pivot; MODE 2: backward sums (sum du, sum du * xhat) of C = sum.
template <int MODE>
__global__ __launch_bounds__(256) void splitk_finish_stats_kernel(const float* __restrict__ ws, int nsplit, long long M, int N,
                                                                  float* __restrict__ C, long long ldc,
                                                                  const float* __restrict__ bias, float slope, long long per,
                                                                  const StatSink ss) {
  __shared__ float4 red4[2 * 256];
  constexpr int SLABQ = 32;                      // column quads per slab
  const int CQ = N >> 2;
  const int q_lo = blockIdx.y * SLABQ;
  const int TQ = CQ - q_lo < SLABQ ? CQ - q_lo : SLABQ, TR = 256 / TQ;
  const int tx = threadIdx.x % TQ, ty = threadIdx.x / TQ;
  const long long r0 = (long long)blockIdx.x * per;
  const long long r1 = r0 + per < M ? r0 + per : M;
  long long live = M;
  if (ss.count) { const long long t = (long long)ss.count[0] * ss.unit; live = t < M ? t : M; }
  const long long MN = M * N;
  const long long HW = (long long)ss.H * ss.W;
  {
    const int n = 4 * (q_lo + tx);
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) b4 = *reinterpret_cast<const float4*>(bias + n);
    auto finished = [&](long long r) -> float4 {
      float4 v = *reinterpret_cast<const float4*>(ws + r * N + n);
      for (int sp = 1; sp < nsplit; ++sp) {
        const float4 u = *reinterpret_cast<const float4*>(ws + (long long)sp * MN + r * N + n);
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
      }
      v.x = leaky(v.x + b4.x, slope); v.y = leaky(v.y + b4.y, slope);
      v.z = leaky(v.z + b4.z, slope); v.w = leaky(v.w + b4.w, slope);
      return v;
    };
    float4 pv = make_float4(0.f, 0.f, 0.f, 0.f), sc = pv, sh = pv, mu = pv, is = pv;
    if (MODE == 1) pv = finished(r0);          // (every row lane of the quad recomputes it: cache hits)
    else {
      sc = *reinterpret_cast<const float4*>(ss.scale + n); sh = *reinterpret_cast<const float4*>(ss.shift + n);
      mu = *reinterpret_cast<const float4*>(ss.mean + n); is = *reinterpret_cast<const float4*>(ss.invstd + n);
    }
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    if (ty < TR) {
      for (long long r = r0 + ty; r < r1; r += TR) {
        const float4 v = finished(r);
        *reinterpret_cast<float4*>(C + r * ldc + n) = v;
        if (r >= live) continue;
        if (MODE == 1) {
          const float4 d = make_float4(v.x - pv.x, v.y - pv.y, v.z - pv.z, v.w - pv.w);
          s0.x += d.x; s0.y += d.y; s0.z += d.z; s0.w += d.w;
          s1.x = fmaf(d.x, d.x, s1.x); s1.y = fmaf(d.y, d.y, s1.y); s1.z = fmaf(d.z, d.z, s1.z); s1.w = fmaf(d.w, d.w, s1.w);
        } else {
          long long row = r;
          if (ss.pool2) {
            const long long nb = r / HW; const int rem = (int)(r - nb * HW);
            const int hi = rem / ss.W, wi = rem - hi * ss.W;
            row = (nb * (ss.H >> 1) + (hi >> 1)) * (ss.W >> 1) + (wi >> 1);
          }
          const float4 yv = *reinterpret_cast<const float4*>(ss.y + row * ss.ld_y + n);
          const float4 du = make_float4(v.x * (fmaf(yv.x, sc.x, sh.x) > 0.f ? 1.f : ss.slope),
                                        v.y * (fmaf(yv.y, sc.y, sh.y) > 0.f ? 1.f : ss.slope),
                                        v.z * (fmaf(yv.z, sc.z, sh.z) > 0.f ? 1.f : ss.slope),
                                        v.w * (fmaf(yv.w, sc.w, sh.w) > 0.f ? 1.f : ss.slope));
          s0.x += du.x; s0.y += du.y; s0.z += du.z; s0.w += du.w;
          s1.x = fmaf(du.x, (yv.x - mu.x) * is.x, s1.x); s1.y = fmaf(du.y, (yv.y - mu.y) * is.y, s1.y);
          s1.z = fmaf(du.z, (yv.z - mu.z) * is.z, s1.z); s1.w = fmaf(du.w, (yv.w - mu.w) * is.w, s1.w);
        }
      }
    }
    if (TR > 1) {
      __syncthreads();
      red4[threadIdx.x] = s0; red4[256 + threadIdx.x] = s1;
      __syncthreads();
      if (ty == 0) {
        for (int t = 1; t < TR; ++t) {
          const float4 a = red4[t * TQ + tx], b = red4[256 + t * TQ + tx];
          s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
          s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
        }
      }
    }
    if (ty == 0) {
      // tile partials [K][N][tiles], see igemm.h StatSink
      const size_t plane = (size_t)N * ss.tiles;
      float* dst = ss.partial + (size_t)n * ss.tiles + blockIdx.x;
      const float a0[4] = {pv.x, pv.y, pv.z, pv.w}, a1[4] = {s0.x, s0.y, s0.z, s0.w}, a2[4] = {s1.x, s1.y, s1.z, s1.w};
      #pragma unroll
      for (int j = 0; j < 4; ++j) {
        float* d = dst + (size_t)j * ss.tiles;
        if (MODE == 1) { d[0] = a0[j]; d[plane] = a1[j]; d[2 * plane] = a2[j]; }
        else { d[0] = a1[j]; d[plane] = a2[j]; }
      }
    }
  }
}

// Data gradient w.r.t. a few (<= 4) input channels - the RGB input of the discriminators'
// first convolution.  An MFMA tile would be 64 columns wide for 3 useful ones (120 us at the
// bench shape); this is a plain gather: one thread per input pixel, the weights of the
// requested channels in LDS as [tap][co][NC], each live tap a dot product over co.
template <int NC>
__global__ void conv_dgrad_fewc_kernel(const float* __restrict__ dY, int ldy, const float* __restrict__ Wt,
                                       int Cout, int Ctot, int c_begin, int NB, int H, int W, int Ho, int Wo,
                                       int KH, int KW, int stride, int pad, float* __restrict__ dx,
                                       long long ld_dx, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) float wsh[];
  const int taps = KH * KW;
  for (int idx = threadIdx.x; idx < taps * Cout * NC; idx += blockDim.x) {
    const int c = idx % NC, co = (idx / NC) % Cout, tap = idx / (NC * Cout);
    wsh[idx] = Wt[((long long)co * taps + tap) * Ctot + c_begin + c];
  }
  __syncthreads();
  const long long pix = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= (long long)NB * H * W) return;
  const int w = (int)(pix % W), h = (int)((pix / W) % H), n = (int)(pix / ((long long)W * H));
  float acc[NC];
  #pragma unroll
  for (int c = 0; c < NC; ++c) acc[c] = 0.f;
  const bool v4 = (Cout % 4 == 0) && (ldy % 4 == 0) && !((uintptr_t)dY & 15);
  for (int kh = 0; kh < KH; ++kh) {
    const int hh = h + pad - kh;
    if (hh < 0 || hh % stride) continue;
    const int ho = hh / stride;
    if (ho >= Ho) continue;
    for (int kw = 0; kw < KW; ++kw) {
      const int ww = w + pad - kw;
      if (ww < 0 || ww % stride) continue;
      const int wo = ww / stride;
      if (wo >= Wo) continue;
      const float* row = dY + (((long long)n * Ho + ho) * Wo + wo) * ldy;
      const float* wt = wsh + (kh * KW + kw) * Cout * NC;
      if (v4) {
        for (int co = 0; co < Cout; co += 4) {
          const float4 g4 = *reinterpret_cast<const float4*>(row + co);
          const float gv[4] = {g4.x, g4.y, g4.z, g4.w};
          #pragma unroll
          for (int j = 0; j < 4; ++j)
            #pragma unroll
            for (int c = 0; c < NC; ++c) acc[c] = fmaf(gv[j], wt[(co + j) * NC + c], acc[c]);
        }
      } else {
        for (int co = 0; co < Cout; ++co)
          #pragma unroll
          for (int c = 0; c < NC; ++c) acc[c] = fmaf(row[co], wt[co * NC + c], acc[c]);
      }
    }
  }
  float* dst = dx + pix * ld_dx;
  #pragma unroll
  for (int c = 0; c < NC; ++c) dst[c] = accumulate ? dst[c] + acc[c] : acc[c];
}

// The same gather with FOUR LANES PER PIXEL for the geometries whose pixels see at most 2 x 2 taps (kernel 4 /
// stride 2, kernel 3 / stride 2, kernel 2 / stride 1 ...).  One thread per pixel is a chain of 64 dependent
// 16-byte loads on 3 waves per SIMD (55 us for the crops of the object discriminator, profiles/r4_conv_layers.log);
// here lane q of a pixel's quad owns the output channels {16 i + 4 q + j}: a quad's load is one contiguous 64-byte
// piece of the dY row, the (up to) 16 loads of the four taps are issued before the first multiply (branch-free:
// a tap outside the output reads row 0 and its values are replaced by zeros), four times the waves.  The quad's partial sums
// are combined by two xor shuffles (fixed order).
template <int NC>
__global__ __launch_bounds__(256) void conv_dgrad_fewc_quad_kernel(
    const float* __restrict__ dY, int ldy, const float* __restrict__ Wt, int Cout, int Ctot, int c_begin, int NB,
    int H, int W, int Ho, int Wo, int KH, int KW, int stride, int pad, float* __restrict__ dx, long long ld_dx,
    int accumulate) {
  extern __shared__ __attribute__((aligned(16))) float wsh[];
  const int taps = KH * KW;
  for (int idx = threadIdx.x; idx < taps * Cout * NC; idx += blockDim.x) {
    const int c = idx % NC, co = (idx / NC) % Cout, tap = idx / (NC * Cout);
    wsh[idx] = Wt[((long long)co * taps + tap) * Ctot + c_begin + c];
  }
  __syncthreads();
  const long long total = (long long)NB * H * W;
  const long long pix_raw = (long long)blockIdx.x * 64 + (threadIdx.x >> 2);
  const bool live = pix_raw < total;
  const long long pix = live ? pix_raw : total - 1;
  const int q = threadIdx.x & 3;
  const int w = (int)(pix % W), h = (int)((pix / W) % H), n = (int)(pix / ((long long)W * H));
  // the taps that reach this pixel: kh = (h + pad) mod stride + stride a, a = 0, 1 (same for kw)
  const int kh0 = (h + pad) % stride, kw0 = (w + pad) % stride;
  const float* rowp[4];
  const float* wtp[4];
  bool tv[4];
  #pragma unroll
  for (int a = 0; a < 2; ++a)
    #pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int kh = kh0 + stride * a, kw = kw0 + stride * b;
      const int ho = (h + pad - kh) / stride, wo = (w + pad - kw) / stride;       // (exact when the tap is live)
      const bool ok = kh < KH && kw < KW && h + pad - kh >= 0 && w + pad - kw >= 0 && ho < Ho && wo < Wo;
      tv[2 * a + b] = ok;
      rowp[2 * a + b] = dY + (ok ? (((long long)n * Ho + ho) * Wo + wo) * ldy : 0);
      wtp[2 * a + b] = wsh + (ok ? (kh * KW + kw) : 0) * Cout * NC;
    }
  float acc[NC];
  #pragma unroll
  for (int c = 0; c < NC; ++c) acc[c] = 0.f;
  for (int co0 = 0; co0 < Cout; co0 += 64) {
    float4 g[4][4];
    #pragma unroll
    for (int t = 0; t < 4; ++t)
      #pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int co = co0 + 16 * i + 4 * q;
        g[t][i] = *reinterpret_cast<const float4*>(rowp[t] + (co < Cout ? co : 0));
      }
    #pragma unroll
    for (int t = 0; t < 4; ++t)
      #pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int co = co0 + 16 * i + 4 * q;
        const bool on = tv[t] && co < Cout;
        const float gv[4] = {on ? g[t][i].x : 0.f, on ? g[t][i].y : 0.f, on ? g[t][i].z : 0.f, on ? g[t][i].w : 0.f};
        const float* wt = wtp[t] + (co < Cout ? co : 0) * NC;
        #pragma unroll
        for (int j = 0; j < 4; ++j)
          #pragma unroll
          for (int c = 0; c < NC; ++c) acc[c] = fmaf(gv[j], wt[j * NC + c], acc[c]);
      }
  }
  #pragma unroll
  for (int c = 0; c < NC; ++c) {
    acc[c] += __shfl_xor(acc[c], 1);
    acc[c] += __shfl_xor(acc[c], 2);
  }
  if (live && q == 0) {
    float* dst = dx + pix * ld_dx;
    #pragma unroll
    for (int c = 0; c < NC; ++c) dst[c] = accumulate ? dst[c] + acc[c] : acc[c];
  }
}

// finish of the stride-2 parity data gradient: slabs [split][class][Mmax][N]; class (ph, pw) row m
// is destination pixel (n, 2 hp + ph, 2 wp + pw)
__global__ void splitk_finish_parity_kernel(const float* __restrict__ ws, int nsplit, int Mmax, int N,
                                            float* __restrict__ C, long long ldc, int accumulate,
                                            int NB, int H, int W) {
  const long long per_split = 4LL * Mmax * N;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < per_split;
       idx += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(idx % N);
    const long long t = idx / N;
    const int m = (int)(t % Mmax), cls = (int)(t / Mmax);
    const int ph = cls >> 1, pw = cls & 1;
    const int Hc = (H - ph + 1) >> 1, Wc = (W - pw + 1) >> 1;
    if (m >= NB * Hc * Wc) continue;
    float v = 0.f;
    for (int s = 0; s < nsplit; ++s) v += ws[(long long)s * per_split + idx];
    const ParityRow map{H, W, Hc, Wc, ph, pw};
    float* dst = C + map(m) * ldc + n;
    if (accumulate) v += *dst;
    *dst = v;
  }
}

}  // namespace sg2im
#include "conv_halo.h"
#include "conv_fewout.h"
#include "wgrad_halo.h"
namespace sg2im {

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
static int g_num_cu = 256;
static const bool g_fewc_scalar = getenv("SG2IM_FEWC_SCALAR") != nullptr;   // (A/B knob: one thread per pixel)
static const bool g_plan_debug = getenv("SG2IM_PLAN_DEBUG") != nullptr;   // print the launch plans
// LDS request of a "background" weight gradient (sg2im_conv_desc.launch_hints bit 0): 56 KB = at most two
// workgroups per CU.  [measured, profiles/r2_deferred_wgrad_ab.log: 3 resident (no padding) 9.88, 2 resident
// 9.74, 1 resident (84 KB) 10.0 ms per training step]
static const size_t g_bg_lds = 56 * 1024;
static const bool g_plan_tune = getenv("SG2IM_PLAN_TUNE") != nullptr;     // honour SG2IM_FORCE_PLAN
// A/B knob: 0 = the *_bn entry points run conv + the standalone BatchNorm reduction passes (the round-2 launches)
static const bool g_fuse_bn = !(getenv("SG2IM_FUSE_BN") && atoi(getenv("SG2IM_FUSE_BN")) == 0);
// SG2IM_FEWOUT=0: the 1 x 1 convolutions with <= 4 output channels back on the implicit-GEMM kernels (A/B knob, conv_fewout.h)
static const bool g_fewout = !(getenv("SG2IM_FEWOUT") && atoi(getenv("SG2IM_FEWOUT")) == 0);

template <typename K>
static hipError_t ensure_lds(K kernel, size_t bytes) {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

static Src* src_at(ConvGeom& g, int i) { return i == 0 ? &g.s0 : i == 1 ? &g.s1 : i == 2 ? &g.s2 : &g.s3; }

static bool geom_vec4(ConvGeom& g) {
  for (int i = 0; i < g.nsrc; ++i) {
    const Src& s = *src_at(g, i);
    if (s.C % 4 || s.ld % 4 || ((uintptr_t)s.p & 15)) return false;
    if (s.scale && (((uintptr_t)s.scale & 15) || ((uintptr_t)s.shift & 15))) return false;
  }
  return g.Ctot % 4 == 0;
}

static void fill_geom(ConvGeom& g, const sg2im_conv_desc* d) {
  g.nsrc = d->nsrc;
  g.Ctot = 0;
  for (int i = 0; i < 4; ++i) {
    Src& s = *src_at(g, i);
    if (i < d->nsrc) {
      const sg2im_src& q = d->src[i];
      s.p = q.data; s.gidx = q.gather; s.scale = q.scale; s.shift = q.shift; s.slope = q.slope;
      s.C = q.channels; s.ld = q.ld; s.up = q.upsample_log2; s.bf = q.dtype == 1;
      g.Ctot += q.channels;
    } else {
      s = Src{nullptr, nullptr, nullptr, nullptr, 1.f, 0, 0, 0, 0};
    }
  }
  g.NB = d->batch; g.H = d->in_h; g.W = d->in_w; g.Ho = d->out_h; g.Wo = d->out_w;
  g.KH = d->kh; g.KW = d->kw; g.stride = d->stride; g.pad = d->pad;
  g.Wtap = d->weight_channels > 0 ? d->weight_channels : g.Ctot;
}

static int check_desc(const sg2im_conv_desc* d) {
  if (!d || d->nsrc < 1 || d->nsrc > 4) return 1;
  if (d->stride < 1 || d->kh < 1 || d->kw < 1) return 1;
  if (d->compute_dtype != 0 && d->compute_dtype != 1) return 1;
  if (d->weight_channels != 0) {          // (weight rows wider than the sources: at least their channel sum)
    int ct = 0;
    for (int i = 0; i < d->nsrc; ++i) ct += d->src[i].channels;
    if (d->weight_channels < ct) return 1;
  }
  const int eh = (d->in_h + 2 * d->pad - d->kh) / d->stride + 1;
  const int ew = (d->in_w + 2 * d->pad - d->kw) / d->stride + 1;
  if (eh != d->out_h || ew != d->out_w) return 1;
  for (int i = 0; i < d->nsrc; ++i) {
    if (!d->src[i].data || d->src[i].channels < 1 || d->src[i].ld < d->src[i].channels) return 1;
    // (the loaders form addresses from 32-bit byte offsets and evaluate the pending LeakyReLU as max(v, v * slope))
    if (!(d->src[i].slope >= 0.f && d->src[i].slope <= 1.f)) return 1;
    if (!d->src[i].gather &&
        (double)d->batch * (d->in_h >> d->src[i].upsample_log2) * (d->in_w >> d->src[i].upsample_log2) * d->src[i].ld * 4.0 >= 4294967296.0)
      return 1;
    if (d->src[i].upsample_log2 < 0 || d->src[i].upsample_log2 > 1) return 1;
    if (d->src[i].gather && (d->in_h != 1 || d->in_w != 1)) return 1;
    if (d->src[i].upsample_log2 && ((d->in_h & 1) || (d->in_w & 1))) return 1;
  }
  return 0;
}

// ---- tile / split-K choice ------------------------------------------------------------
// Candidates: 0 -> 128x128, 1 -> 128x64, 2 -> 64x64, 3 -> 64x128.  A launch should put >= 2 workgroups
// on each of the 256 CUs; small problems get there through smaller tiles and split-K
// (partials in the workspace, combined by splitk_finish_kernel).
struct Plan { int tile, bm, bn, nsplit; long long tiles; };

static const int kBM[4] = {128, 128, 64, 64}, kBN[4] = {128, 64, 64, 128};
// relative MFMA efficiency of the tile shapes per pass (forward, data gradient, weight gradient);
// these and the constants in launch_cost are fitted to tools/plan_sweep.py measurements
enum { PASS_FWD = 0, PASS_DGRAD = 1, PASS_WGRAD = 2 };
static const double kEff[3][4] = {{1.0, 0.92, 0.80, 0.935}, {1.0, 0.92, 0.895, 0.966}, {1.0, 0.90, 0.80, 0.90}};

// Cost model of one launch, in cycles of a CU's MFMA pipes.  The kernels are MFMA bound, so a
// CU that is handed b workgroups needs b x (chunks x cycles-per-chunk): what matters is the
// block count of the BUSIEST CU, ceil(blocks / #CU) - 2.05 blocks per CU cost as much as 3.
// Fewer than ~3 co-resident workgroups leave the loader latency exposed (`hide`); split-K pays
// for the partial-sum round trip through the workspace and the finish launch.
#ifndef SG2IM_FIN0
#define SG2IM_FIN0 9000.0     // (round 2 sweep of the whole step: 3800 -> 10.42, 9000 -> 10.26, 18000 -> 10.42, 36000 -> 10.92 ms)
#endif
#ifndef SG2IM_FINBW
#define SG2IM_FINBW 2500.0
#endif
static const int kOcc[4] = {3, 4, 6, 4};                  // resident workgroups per CU (VGPR/LDS limited)
static const double g_fin0 = SG2IM_FIN0;
static const double g_finbw = SG2IM_FINBW;
static double launch_cost(int pass, int t, long long tiles, int ns, int iters, long long MN) {
  const long long blocks = tiles * ns;
  const int per = (iters + ns - 1) / ns;
  const double chunk = (double)kBM[t] * kBN[t] * BK * 2.0 / 256.0 / kEff[pass][t];
  const double fixed = 1700.0;                            // prologue + epilogue of a workgroup
  const long long rounds = (blocks + g_num_cu - 1) / g_num_cu;
  const double resident = std::min<double>((double)blocks / g_num_cu, kOcc[t]);
  const double hide = resident >= 2.9 ? 1.0 : resident >= 1.9 ? 0.925 : 0.51;
  double c = (double)rounds * (per * chunk + fixed) / hide;
  if (ns > 1) c += g_fin0 + (double)ns * (double)MN * 8.0 / g_finbw;
  return c;
}

static int split_for(int pass, int t, long long tiles, int iters, long long MN, size_t ws_bytes, int min_iters,
                     double* cost_out = nullptr) {
  long long cap = std::max(1, iters / min_iters);
  cap = std::min<long long>(cap, MN > 0 ? std::max<long long>(1, (long long)(ws_bytes / sizeof(float)) / MN) : 1);
  // (tiny outputs - the weight gradients of the RGB layers - are latency bound and may be
  // spread much wider than the usual cap)
  cap = std::min<long long>(cap, MN <= 16384 ? 512 : 64);
  int best = 1;
  double best_c = launch_cost(pass, t, tiles, 1, iters, MN);
  for (int ns = 2; ns <= cap; ++ns) {
    const int per = (iters + ns - 1) / ns;
    if ((iters + per - 1) / per != ns) continue;          // every split non-empty
    const double c = launch_cost(pass, t, tiles, ns, iters, MN);
    if (c < best_c) { best_c = c; best = ns; }
  }
  if (cost_out) *cost_out = best_c;
  return best;
}

// ntn(bn): number of N tiles for tile width bn (wgrad tiles N per tap)
template <typename NT>
static Plan make_plan(int pass, long long M, long long N, int iters, long long MN, size_t ws_bytes, bool can_split,
                      int min_iters, bool only64, NT ntn) {
  Plan best{2, 64, 64, 1, 0};
  double best_cost = -1.0;
  for (int t = 0; t < 4; ++t) {
    if (only64 && t != 2) continue;
    const long long tm = (M + kBM[t] - 1) / kBM[t], tn = ntn(kBN[t]);
    const long long tiles = tm * tn;
    double cost;
    int ns = 1;
    if (can_split) ns = split_for(pass, t, tiles, iters, MN, ws_bytes, min_iters, &cost);
    else cost = launch_cost(pass, t, tiles, 1, iters, MN);
    if (g_plan_debug) fprintf(stderr, "[sg2im plan]   tile %dx%d tiles=%lld ns=%d cost=%.0f\n", kBM[t], kBN[t], tiles, ns, cost);
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = Plan{t, kBM[t], kBN[t], ns, tiles}; }
  }
  if (g_plan_tune) {                 // tools/plan_sweep.py: force "tile,nsplit" for this call
    const char* f = getenv("SG2IM_FORCE_PLAN");
    int t = 0, ns = 1;
    if (f && sscanf(f, "%d,%d", &t, &ns) == 2 && t >= 0 && t < 4 && !(only64 && t != 2)) {
      const long long tiles = ((M + kBM[t] - 1) / kBM[t]) * ntn(kBN[t]);
      long long cap = can_split ? std::max(1, iters / min_iters) : 1;
      if (MN > 0) cap = std::min<long long>(cap, std::max<long long>(1, (long long)(ws_bytes / sizeof(float)) / MN));
      ns = (int)std::max<long long>(1, std::min<long long>(ns, cap));
      const int per = (iters + ns - 1) / ns;
      ns = (iters + per - 1) / per;
      best = Plan{t, kBM[t], kBN[t], ns, tiles};
    }
  }
  if (g_plan_debug) fprintf(stderr, "[sg2im plan] M=%lld N=%lld iters=%d -> %dx%d x%d\n", M, N, iters, best.bm, best.bn, best.nsplit);
  return best;
}

static hipError_t finish_split(const Epi& e, long long M, int N, hipStream_t st,
                               const float* ws2 = nullptr, float* C2 = nullptr, int N2 = 0) {
  if (e.nsplit <= 1) return hipSuccess;
#ifdef SG2IM_PROBE_SKIP_FINISH
  // timing probe only (WRONG results): what would the step cost if the split-K finish launches - all of them (1) or the
  // weight gradients' (2: they pass a bias-partial pointer or write dW rows) - did not exist?  tools/r6_probe_finish.sh
  { static const int mode = getenv("SG2IM_SKIP_FINISH_PROBE") ? atoi(getenv("SG2IM_SKIP_FINISH_PROBE")) : 0;
    if (mode == 1 || (mode == 2 && e.accumulate)) return hipSuccess; }
#endif
  const long long MN = M * N;
  int SL = 1;
  // (up to 8 workgroups per CU: the loop over the splits is a chain of load latencies, not bandwidth - 256 splits of a
  // 64 x 576 weight gradient took 19 us with 2 split lanes)
  while (SL < 16 && 2 * SL <= e.nsplit / 4 && (MN * SL + 255) / 256 < 8 * g_num_cu) SL *= 2;
  const bool v4 = SL == 1 && N % 4 == 0 && e.ldc % 4 == 0 && !((uintptr_t)e.ws & 15) && !((uintptr_t)e.C & 15) &&
                  (!e.bias || !((uintptr_t)e.bias & 15)) && e.col_wtap == 0 &&
                  (!e.mask || (e.ld_mask % 4 == 0 && !((uintptr_t)e.mask & 15)));
  if (v4) {
    const int blocks4 = (int)std::min<long long>((MN / 4 + 255) / 256, 4096);
    SG2IM_LAUNCH(splitk_finish_v4_kernel, dim3(blocks4), dim3(256), 0, st, e.ws, e.nsplit, MN, N, e.C, e.ldc,
                       e.bias, e.slope, e.accumulate, ws2, C2, N2, (FinishMask{e.mask, e.ld_mask, e.mask_slope}));
    return hipGetLastError();
  }
  const int per = 256 / SL;
  const int blocks = (int)std::min<long long>((MN + per - 1) / per, 4096);
  SG2IM_LAUNCH(splitk_finish_kernel, dim3(blocks), dim3(256), 0, st, e.ws, e.nsplit, MN, N, e.C, e.ldc,
                     e.bias, e.slope, e.accumulate, SL, ws2, C2, N2, e.col_ctot, e.col_wtap, e.mask, e.ld_mask, e.mask_slope);
  return hipGetLastError();
}

// Per-instantiation "attributes set" flags.  sg2im_init() sets every one of them up front, so that
// no hipFuncSetAttribute call is left for a first launch that may happen inside a stream capture;
// a caller that skipped sg2im_init() still gets them lazily.
template <int BM, int BN, int VEC, bool GATHER> bool g_fwd_ready = false;
template <int BM, int BN, int VA, int VB> bool g_dgrad_ready = false;
template <int BM, int BN, int VEC, bool GATHER> bool g_wgrad_ready = false;

template <int BM, int BN> constexpr size_t fwd_lds() {
  return (LdsTile<BM, false>::FLOATS + LdsTile<BN, false>::FLOATS) * sizeof(float);
}
template <int BM, int BN> constexpr size_t dgrad_lds() {
  return (LdsTile<BM, false>::FLOATS + LdsTile<BN, true>::FLOATS) * sizeof(float);
}
template <int BM, int BN> constexpr size_t wgrad_lds() {
  return (LdsTile<BM, true>::FLOATS + LdsTile<BN, true>::FLOATS) * sizeof(float);
}

template <int BM, int BN, int VEC, bool GATHER> static hipError_t prepare_fwd() {
  if (g_fwd_ready<BM, BN, VEC, GATHER>) return hipSuccess;
  const hipError_t e = ensure_lds(conv_fwd_kernel<BM, BN, VEC, GATHER>, fwd_lds<BM, BN>());
  if (e == hipSuccess) g_fwd_ready<BM, BN, VEC, GATHER> = true;
  return e;
}
template <int BM, int BN, int VA, int VB> static hipError_t prepare_dgrad() {
  if (g_dgrad_ready<BM, BN, VA, VB>) return hipSuccess;
  const hipError_t e = ensure_lds(conv_dgrad_kernel<BM, BN, VA, VB>, dgrad_lds<BM, BN>());
  if (e == hipSuccess) g_dgrad_ready<BM, BN, VA, VB> = true;
  return e;
}
template <int BM, int BN, int VEC, bool GATHER> static hipError_t prepare_wgrad() {
  if (g_wgrad_ready<BM, BN, VEC, GATHER>) return hipSuccess;
  const hipError_t e = ensure_lds(conv_wgrad_kernel<BM, BN, VEC, GATHER>, std::max(wgrad_lds<BM, BN>(), g_bg_lds));
  if (e == hipSuccess) g_wgrad_ready<BM, BN, VEC, GATHER> = true;
  return e;
}

template <int BM, int BN, int VEC, bool GATHER>
static hipError_t launch_fwd_g(FwdParams& p, hipStream_t st) {
  constexpr size_t lds = fwd_lds<BM, BN>();
  { hipError_t e = prepare_fwd<BM, BN, VEC, GATHER>(); if (e != hipSuccess) return e; }
  dim3 grid((p.Cout + BN - 1) / BN, (p.M + BM - 1) / BM, p.e.nsplit);
  SG2IM_LAUNCH((conv_fwd_kernel<BM, BN, VEC, GATHER>), grid, dim3(NTHREADS), lds, st, p);
  return hipGetLastError();
}

static bool any_gather(ConvGeom& g) { for (int i = 0; i < g.nsrc; ++i) if (src_at(g, i)->gidx) return true; return false; }

template <int BM, int BN, int VEC>
static hipError_t launch_fwd(FwdParams& p, hipStream_t st) {
  if (VEC == 4 && any_gather(p.g)) return launch_fwd_g<BM, BN, VEC, true>(p, st);     // (row gathers: tiny GEMMs)
  return launch_fwd_g<BM, BN, VEC, false>(p, st);
}

template <int BM, int BN, int VA, int VB>
static hipError_t launch_dgrad(DgradParams& p, hipStream_t st) {
  constexpr size_t lds = dgrad_lds<BM, BN>();
  { hipError_t e = prepare_dgrad<BM, BN, VA, VB>(); if (e != hipSuccess) return e; }
  dim3 grid((p.Nc + BN - 1) / BN, (p.M + BM - 1) / BM, p.parity ? 4 * p.e.nsplit : p.e.nsplit);
  SG2IM_LAUNCH((conv_dgrad_kernel<BM, BN, VA, VB>), grid, dim3(NTHREADS), lds, st, p);
  return hipGetLastError();
}

// (two LDS images of the halo'd weight-gradient kernel: 84-88 KB, above the 64 KB a launch may ask for unprepared)
static bool g_wgrad_halo_ready = false;
static hipError_t prepare_wgrad_halo() {
  if (g_wgrad_halo_ready) return hipSuccess;
  hipError_t e = ensure_lds(conv_wgrad_halo_kernel<4, 16>, wgrad_halo_lds<4, 16>());
  if (e == hipSuccess) e = ensure_lds(conv_wgrad_halo_kernel<8, 8>, wgrad_halo_lds<8, 8>());
#define SG2IM_WGH(XB_, YB_) if (e == hipSuccess) e = ensure_lds(conv_wgrad_halo_h_kernel<XB_, YB_>, wgrad_halo_h_lds())
  SG2IM_WGH(0, false); SG2IM_WGH(1, false); SG2IM_WGH(2, false); SG2IM_WGH(0, true); SG2IM_WGH(1, true); SG2IM_WGH(2, true);
#undef SG2IM_WGH
  if (e == hipSuccess) g_wgrad_halo_ready = true;
  return e;
}

template <int BM, int BN> bool g_wgrad_fr_ready = false;
template <int BM, int BN> static hipError_t prepare_wgrad_fr() {
  if (g_wgrad_fr_ready<BM, BN>) return hipSuccess;
  const hipError_t e = ensure_lds(conv_wgrad_kernel<BM, BN, 4, false, false, true>,
                                  std::max(wgrad_lds<BM, BN>(), g_bg_lds));
  if (e == hipSuccess) g_wgrad_fr_ready<BM, BN> = true;
  return e;
}

template <int BM, int BN, int VEC, bool GATHER>
static hipError_t launch_wgrad_g(WgradParams& p, int ntiles_n, hipStream_t st) {
  constexpr size_t lds = wgrad_lds<BM, BN>();
  { hipError_t e = prepare_wgrad<BM, BN, VEC, GATHER>(); if (e != hipSuccess) return e; }
  p.ntiles_n = ntiles_n;
  p.ntiles_m = (p.Cout + BM - 1) / BM;
  dim3 grid(p.ntiles_n, p.ntiles_m, p.e.nsplit);
  // background launch: cap the resident workgroups per CU of the large-tile kernels so that small kernels of
  // a concurrent stream always find a free slot
  const size_t lds_req = (p.background && BM * BN > 64 * 64) ? std::max(lds, g_bg_lds) : lds;
  if constexpr (VEC == 4 && !GATHER) {
    // "fast rows" form: the BK pixels of every K chunk lie in one output row (see conv_wgrad_body)
    if (p.g.stride == 1 && p.g.Wo % BK == 0) {
      { hipError_t e = prepare_wgrad_fr<BM, BN>(); if (e != hipSuccess) return e; }
      SG2IM_LAUNCH((conv_wgrad_kernel<BM, BN, 4, false, false, true>), grid, dim3(NTHREADS), lds_req, st, p);
      return hipGetLastError();
    }
  }
  SG2IM_LAUNCH((conv_wgrad_kernel<BM, BN, VEC, GATHER>), grid, dim3(NTHREADS), lds_req, st, p);
  return hipGetLastError();
}

template <int BM, int BN, int VEC>
static hipError_t launch_wgrad(WgradParams& p, int ntiles_n, hipStream_t st) {
  if (VEC == 4 && any_gather(p.g)) return launch_wgrad_g<BM, BN, VEC, true>(p, ntiles_n, st);
  return launch_wgrad_g<BM, BN, VEC, false>(p, ntiles_n, st);
}

// bf16 operand path (desc->compute_dtype == 1): vectorised, gather-free launches only (spatial convs);
// everything else stays on the fp32 kernels.  Largest LDS image: 128x128 m-major = 2 x 10 KB.
template <int BM, int BN>
static hipError_t launch_fwd_h(FwdParams& p, hipStream_t st) {
  constexpr size_t lds = TileBytes<true, BM, false>::value + TileBytes<true, BN, false>::value;
  dim3 grid((p.Cout + BN - 1) / BN, (p.M + BM - 1) / BM, p.e.nsplit);
  SG2IM_LAUNCH((conv_fwd_kernel<BM, BN, 4, false, true>), grid, dim3(NTHREADS), lds, st, p);
  return hipGetLastError();
}
template <int BM, int BN>
static hipError_t launch_dgrad_h(DgradParams& p, hipStream_t st) {
  constexpr size_t lds = TileBytes<true, BM, false>::value + TileBytes<true, BN, true>::value;
  dim3 grid((p.Nc + BN - 1) / BN, (p.M + BM - 1) / BM, p.parity ? 4 * p.e.nsplit : p.e.nsplit);
  SG2IM_LAUNCH((conv_dgrad_kernel<BM, BN, 4, 4, true>), grid, dim3(NTHREADS), lds, st, p);
  return hipGetLastError();
}
template <int BM, int BN>
static hipError_t launch_wgrad_h(WgradParams& p, int ntiles_n, hipStream_t st) {
  constexpr size_t lds = TileBytes<true, BM, true>::value + TileBytes<true, BN, true>::value;
  p.ntiles_n = ntiles_n;
  p.ntiles_m = (p.Cout + BM - 1) / BM;
  dim3 grid(p.ntiles_n, p.ntiles_m, p.e.nsplit);
  const size_t lds_req = (p.background && BM * BN > 64 * 64) ? std::max(lds, std::min<size_t>(g_bg_lds, 64 * 1024)) : lds;
  SG2IM_LAUNCH((conv_wgrad_kernel<BM, BN, 4, false, true>), grid, dim3(NTHREADS), lds_req, st, p);
  return hipGetLastError();
}

// ST launches: the epilogue also reduces BatchNorm tile partials (no split-K; float4 loaders, no row gathers)
template <int BM, int BN, bool H>
static hipError_t launch_fwd_st(FwdParams& p, hipStream_t st) {
  constexpr size_t lds = TileBytes<H, BM, false>::value + TileBytes<H, BN, false>::value;
  dim3 grid((p.Cout + BN - 1) / BN, (p.M + BM - 1) / BM, 1);
  SG2IM_LAUNCH((conv_fwd_kernel<BM, BN, 4, false, H, true>), grid, dim3(NTHREADS), lds, st, p);
  return hipGetLastError();
}
template <int BM, int BN, bool H>
static hipError_t launch_dgrad_st(DgradParams& p, hipStream_t st) {
  constexpr size_t lds = TileBytes<H, BM, false>::value + TileBytes<H, BN, true>::value;
  dim3 grid((p.Nc + BN - 1) / BN, (p.M + BM - 1) / BM, 1);
  SG2IM_LAUNCH((conv_dgrad_kernel<BM, BN, 4, 4, H, true>), grid, dim3(NTHREADS), lds, st, p);
  return hipGetLastError();
}

// rows per workgroup / row blocks / column slabs of splitk_finish_stats_kernel: ~8 workgroups per CU, at least 8
// rows each, at most as many row blocks as the partial buffer holds (K floats per row block and column)
static void finish_stats_grid(long long M, int N, int K, size_t partial_floats, int* nblk, long long* per, int* nslab) {
  *nslab = (N / 4 + 31) / 32;
  long long want = std::max<long long>(1, 2048 / *nslab);
  const long long cap = std::max<long long>(1, (long long)(partial_floats / ((size_t)K * N)));
  want = std::min<long long>(std::min<long long>(want, cap), 4096);
  *per = std::max<long long>(8, (M + want - 1) / want);
  *nblk = (int)((M + *per - 1) / *per);
}

static bool al16p(const void* q) { return ((uintptr_t)q & 15) == 0; }

// ---- halo'd-tile kernels (conv_halo.h): 3x3 / stride 1 / pad 1, float4 loaders, fp32 ----
// A/B knob: 0 = every convolution on the first-generation per-tap kernels
static const bool g_halo = !(getenv("SG2IM_HALO") && atoi(getenv("SG2IM_HALO")) == 0);
// A/B knob: 0 = weight gradients of the 3x3 convolutions on the per-tap kernel, 1 (default) = the halo'd-tile kernel
// where it measured faster, 2 = wherever its geometry allows (the parity tests run every eligible shape through it)
static const int g_wgrad_halo = getenv("SG2IM_WGRAD_HALO") ? atoi(getenv("SG2IM_WGRAD_HALO")) : 1;
struct HaloPlan { int rt, ct, bn, nsplit, patches; };

static bool halo_geometry(const sg2im_conv_desc* d) {
  return g_halo && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1 &&
         d->in_h < 32768 && d->in_w < 32768;
}

// patch shape for an H x W map, N-tile width and split-K (over whole 32-channel chunks) for `ncols` output columns
static bool halo_plan(int NB, int H, int W, int ncols, int nchunks, size_t ws_bytes, bool can_split, HaloPlan* pl) {
  // (the squarest patch first: 8 x 16 has the smallest halo - 180 pixels for 128 outputs against 204 / 264 - and,
  // at 35 KB of LDS with 64-wide column tiles, four resident workgroups per CU instead of three: the 64-channel
  // 64 x 64 layers went 92 -> 114 TFLOP/s forward with it, profiles/r3_halo_layers.log)
  if (W % 16 == 0 && H % 8 == 0) { pl->rt = 8; pl->ct = 16; }
  else if (W % 32 == 0 && H % 4 == 0) { pl->rt = 4; pl->ct = 32; }
  else if (W % 64 == 0 && H % 2 == 0) { pl->rt = 2; pl->ct = 64; }
  else return false;
  const long long M = (long long)NB * H * W;
  pl->patches = (int)(M / 128);
  if (pl->patches < 1 || ncols < 32 || nchunks < 1) return false;
  // (64-wide column tiles only: the 128-wide form - 2 wavefronts per SIMD at ~200 registers - measured 90 against
  // 113 TFLOP/s on the 128-column data gradients of the 64 x 64 layers; the template keeps the parameter)
  pl->bn = 64;
  const long long blocks = (long long)pl->patches * ((ncols + pl->bn - 1) / pl->bn);
  int ns = 1;
  if (can_split && blocks < (3 * g_num_cu) / 2) {
    ns = (int)std::min<long long>(nchunks, (2 * g_num_cu + blocks - 1) / blocks);
    ns = (int)std::min<long long>(ns, std::max<long long>(1, (long long)(ws_bytes / sizeof(float)) / (M * ncols)));
    const int per = (nchunks + ns - 1) / ns;
    ns = (nchunks + per - 1) / per;
  }
  pl->nsplit = std::max(1, ns);
  if (g_plan_debug) fprintf(stderr, "[sg2im halo] M=%lld N=%d chunks=%d -> patch %dx%d bn=%d x%d\n", M, ncols, nchunks, pl->rt, pl->ct, pl->bn, pl->nsplit);
  return true;
}

// A/B knobs of the bf16 halo'd kernels: SG2IM_HALO_TG = taps per staging group of the weight-mirror kernels (1 or 3:
// conv_halo.h; default 3 - the nine-tap form measured slower and is not instantiated), SG2IM_HALO_WB = 0 ignores the mirror
static const bool g_halo_wb = !(getenv("SG2IM_HALO_WB") && atoi(getenv("SG2IM_HALO_WB")) == 0);
static const int g_halo_tg = getenv("SG2IM_HALO_TG") ? atoi(getenv("SG2IM_HALO_TG")) : 3;
template <int RT, int CT, int BN, bool DG, bool ST, bool H, int TG, bool WB, bool AB>
static hipError_t launch_halo_v(HaloParams& p, const HaloPlan& pl, hipStream_t st) {
  constexpr size_t lds = halo_lds<RT, CT, BN, DG, H, TG>();
  static_assert(lds <= 64 * 1024, "the instantiated forms fit the default dynamic-LDS limit");
  dim3 grid((p.N + BN - 1) / BN, pl.patches, pl.nsplit);
  SG2IM_LAUNCH((conv_halo_kernel<RT, CT, BN, DG, ST, H, TG, WB, AB>), grid, dim3(NTHREADS), lds, st, p);
  return hipGetLastError();
}
template <int RT, int CT, int BN, bool DG, bool ST, bool H>
static hipError_t launch_halo_t(HaloParams& p, const HaloPlan& pl, hipStream_t st) {
  if constexpr (H) {
    // AB: bfloat16 storage on the A side (forward: any source; data gradient: dY)
    bool ab = DG ? p.dy_bf != 0 : false;
    if (!DG) { ConvGeom& g = p.g; for (int i = 0; i < g.nsrc; ++i) ab = ab || src_at(g, i)->bf; }
    const bool wb = g_halo_wb && p.Wh != nullptr;
    if (ab) {
      if (wb && g_halo_tg == 3) return launch_halo_v<RT, CT, BN, DG, ST, true, 3, true, true>(p, pl, st);
      if (wb) return launch_halo_v<RT, CT, BN, DG, ST, true, 1, true, true>(p, pl, st);
      return launch_halo_v<RT, CT, BN, DG, ST, true, 1, false, true>(p, pl, st);
    }
    if (wb && g_halo_tg == 3) return launch_halo_v<RT, CT, BN, DG, ST, true, 3, true, false>(p, pl, st);
    if (wb) return launch_halo_v<RT, CT, BN, DG, ST, true, 1, true, false>(p, pl, st);
  }
  return launch_halo_v<RT, CT, BN, DG, ST, H, 1, false, false>(p, pl, st);
}
// hb: bf16 operands (sg2im_conv_desc.compute_dtype 1)
template <bool DG, bool ST>
static hipError_t launch_halo(HaloParams& p, const HaloPlan& pl, hipStream_t st, bool hb = false) {
#define SG2IM_HALO_CASE(RT_, CT_) \
  if (pl.ct == CT_) return hb ? launch_halo_t<RT_, CT_, 64, DG, ST, true>(p, pl, st) : launch_halo_t<RT_, CT_, 64, DG, ST, false>(p, pl, st)
  SG2IM_HALO_CASE(2, 64); SG2IM_HALO_CASE(4, 32); SG2IM_HALO_CASE(8, 16);
#undef SG2IM_HALO_CASE
  return hipErrorInvalidValue;
}

__global__ void init_probe_kernel(int* flag) { if (flag) flag[0] = 1; }

}  // namespace sg2im

using namespace sg2im;

extern "C" {

// One-time set-up of everything the launchers would otherwise do lazily on a first launch:
// the dynamic-LDS attribute of every implicit-GEMM instantiation and the loading of this
// library's code object (one empty launch on the null stream, synchronised).  After it the
// entry points make no HIP call other than kernel launches / async memsets on `stream`, so
// a first launch may happen inside a stream capture.  Idempotent; call it before capturing.
int sg2im_init(void) {
  static bool done = false;
  if (done) return SG2IM_OK;
  hipError_t e = hipSuccess;
#define SG2IM_PREP(call) do { if (e == hipSuccess) e = (call); } while (0)
#define SG2IM_PREP_TILES(fn, ...)                                                   \
  SG2IM_PREP((fn<128, 128, __VA_ARGS__>())); SG2IM_PREP((fn<128, 64, __VA_ARGS__>())); \
  SG2IM_PREP((fn<64, 64, __VA_ARGS__>())); SG2IM_PREP((fn<64, 128, __VA_ARGS__>()))
  SG2IM_PREP_TILES(prepare_fwd, 4, false);
  SG2IM_PREP_TILES(prepare_fwd, 4, true);
  SG2IM_PREP((prepare_fwd<64, 64, 1, false>()));
  SG2IM_PREP_TILES(prepare_dgrad, 4, 4);
  SG2IM_PREP((prepare_dgrad<64, 64, 4, 1>())); SG2IM_PREP((prepare_dgrad<128, 64, 4, 1>()));
  SG2IM_PREP((prepare_dgrad<64, 64, 1, 1>()));
  SG2IM_PREP_TILES(prepare_wgrad, 4, false);
  SG2IM_PREP((prepare_wgrad_fr<128, 128>())); SG2IM_PREP((prepare_wgrad_fr<128, 64>()));
  SG2IM_PREP((prepare_wgrad_fr<64, 64>())); SG2IM_PREP((prepare_wgrad_fr<64, 128>()));
  SG2IM_PREP_TILES(prepare_wgrad, 4, true);
  SG2IM_PREP((prepare_wgrad<64, 64, 1, false>()));
#undef SG2IM_PREP_TILES
#undef SG2IM_PREP
  if (e == hipSuccess) e = ensure_lds(conv_wgrad_group_kernel, wgrad_lds<64, 64>());
  if (e == hipSuccess) e = prepare_wgrad_halo();
  if (e == hipSuccess) e = gcn::prepare();          // the persistent GraphTripleConv-stack kernels (gcn_persist.hip)
  if (e != hipSuccess) return SG2IM_ERR_HIP;
  SG2IM_LAUNCH(init_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)0, (int*)nullptr);
  if (hipGetLastError() != hipSuccess || hipStreamSynchronize((hipStream_t)0) != hipSuccess) return SG2IM_ERR_HIP;
  done = true;
  return SG2IM_OK;
}

static int bn_fwd_standalone(const sg2im_bn_fwd* bn, const float* out, long long rows, int cout, long long ld_out,
                             hipStream_t stream) {
  return sg2im_bn_stats(out, rows, cout, ld_out, bn->gamma, bn->beta, bn->eps, bn->momentum, bn->training,
                        bn->running_mean, bn->running_var, bn->num_batches_tracked, bn->unbiased_rows, bn->mean,
                        bn->invstd, bn->scale, bn->shift, bn->partial, bn->count, bn->count_unit, stream);
}

// bn != nullptr: followed by the BatchNorm statistics of `out` (sg2im_conv2d_forward_bn)
static int conv_forward_impl(const sg2im_conv_desc* d, const float* weight, int cout, const float* bias,
                             float out_slope, float* out, long long ld_out, int accumulate,
                             float* workspace, size_t workspace_bytes, const sg2im_bn_fwd* bn, hipStream_t stream) {
  if (check_desc(d) || !weight || !out || cout < 1) return SG2IM_ERR_ARG;
  FwdParams p;
  fill_geom(p.g, d);
  p.Wt = weight; p.Cout = cout;
  p.M = d->batch * d->out_h * d->out_w;
  p.st = StatSink{};
  if (p.M == 0) return bn ? SG2IM_ERR_ARG : SG2IM_OK;
  const bool v4 = geom_vec4(p.g) && !((uintptr_t)weight & 15);
  const int taps = d->kh * d->kw;
  if (v4) {
    p.nch = 0;
    for (int i = 0; i < p.g.nsrc; ++i) p.nch += (src_at(p.g, i)->C + BK - 1) / BK;
    p.iters = taps * p.nch;
  } else {
    p.nch = 0;
    p.iters = (taps * p.g.Ctot + BK - 1) / BK;
  }
  // BatchNorm statistics of the output from the same launches: in the conv epilogue (no split-K) or in the
  // split-K finish; anything that does not qualify runs the standalone statistics pass afterwards
  const bool bn_train = bn && bn->training;
  const bool st_ok = g_fuse_bn && bn_train && v4 && !any_gather(p.g) && !accumulate && bn->partial && al16p(bn->partial);
  hipError_t err;
  HaloPlan hp;
  int nsplit = 0;
  // bfloat16 STORAGE (sg2im_src.dtype / sg2im_conv_desc.out_dtype): the bf16 halo'd kernel only, without split-K (the
  // finish kernels read and write fp32) - anything else is refused, never silently misread
  bool sb = d->out_dtype == 1;
  for (int i = 0; i < d->nsrc; ++i) sb = sb || d->src[i].dtype == 1;
  if (sb && (d->compute_dtype != 1 || accumulate || (bn && bn->count))) return SG2IM_ERR_ARG;
  if (v4 && !any_gather(p.g) && !accumulate && halo_geometry(d) && p.g.Wtap % 4 == 0 &&
      halo_plan(d->batch, d->in_h, d->in_w, cout, p.nch, workspace_bytes, workspace != nullptr && !sb, &hp)) {
    HaloParams q;
    q.g = p.g; q.Wt = weight; q.Wh = (const bf16_t*)d->weight_bf16; q.N = cout; q.c_begin = 0; q.nchunks = p.nch;
    q.tiles_x = d->in_w / hp.ct; q.tiles_y = d->in_h / hp.rt; q.M = p.M;
    q.e = Epi{out, ld_out, bias, out_slope, 0, workspace, hp.nsplit};
    q.e.out_bf = d->out_dtype == 1;
    q.dy_bf = 0;
    q.st = StatSink{};
    p.e = q.e;
    nsplit = hp.nsplit;
    if (sb && bn_train && !(st_ok && (size_t)hp.patches * 3 * cout <= bn->partial_floats)) return SG2IM_ERR_ARG;
    if (sb && bn && !bn_train) {       // eval-mode BatchNorm: no statistics to reduce, the folded affine comes from the running ones
      if (launch_halo<false, false>(q, hp, stream, true) != hipSuccess) return SG2IM_ERR_HIP;
      return bn_fwd_standalone(bn, (const float*)nullptr, p.M, cout, ld_out, stream);
    }
    if (st_ok && !bn->count && hp.nsplit == 1 && (size_t)hp.patches * 3 * cout <= bn->partial_floats) {
      q.st.partial = bn->partial; q.st.tiles = hp.patches;
      if (launch_halo<false, true>(q, hp, stream, d->compute_dtype == 1) != hipSuccess) return SG2IM_ERR_HIP;
      return bn_stats_finish_tiles(bn->partial, hp.patches, 128, p.M, cout, bn, stream);
    }
    err = launch_halo<false, false>(q, hp, stream, d->compute_dtype == 1);
  } else {
  if (sb) return SG2IM_ERR_ARG;
  const Plan pl = make_plan(PASS_FWD, p.M, cout, p.iters, (long long)p.M * cout, workspace_bytes, workspace != nullptr, 4, !v4,
                            [&](int bn) { return (long long)(cout + bn - 1) / bn; });
  p.e = Epi{out, ld_out, bias, out_slope, accumulate, workspace, pl.nsplit};
  nsplit = pl.nsplit;
  if (st_ok && pl.nsplit == 1 && (size_t)((p.M + pl.bm - 1) / pl.bm) * 3 * cout <= bn->partial_floats) {
    p.st.partial = bn->partial; p.st.count = bn->count; p.st.unit = bn->count_unit;
    p.st.tiles = (p.M + pl.bm - 1) / pl.bm;
    const bool hb = d->compute_dtype == 1;
#define SG2IM_ST(BM_, BN_) (hb ? launch_fwd_st<BM_, BN_, true>(p, stream) : launch_fwd_st<BM_, BN_, false>(p, stream))
    err = pl.tile == 0 ? SG2IM_ST(128, 128) : pl.tile == 1 ? SG2IM_ST(128, 64) : pl.tile == 2 ? SG2IM_ST(64, 64) : SG2IM_ST(64, 128);
#undef SG2IM_ST
    if (err != hipSuccess) return SG2IM_ERR_HIP;
    return bn_stats_finish_tiles(bn->partial, (p.M + pl.bm - 1) / pl.bm, pl.bm, p.M, cout, bn, stream);
  }
  if (v4 && d->compute_dtype == 1 && !any_gather(p.g)) {
    err = pl.tile == 0 ? launch_fwd_h<128, 128>(p, stream) : pl.tile == 1 ? launch_fwd_h<128, 64>(p, stream)
        : pl.tile == 2 ? launch_fwd_h<64, 64>(p, stream) : launch_fwd_h<64, 128>(p, stream);
  } else if (v4) {
    err = pl.tile == 0 ? launch_fwd<128, 128, 4>(p, stream) : pl.tile == 1 ? launch_fwd<128, 64, 4>(p, stream)
        : pl.tile == 2 ? launch_fwd<64, 64, 4>(p, stream) : launch_fwd<64, 128, 4>(p, stream);
  } else {
    err = launch_fwd<64, 64, 1>(p, stream);
  }
  }
  if (err != hipSuccess) return SG2IM_ERR_HIP;
  if (st_ok && nsplit > 1 && cout % 4 == 0 && ld_out % 4 == 0 && al16p(out) && al16p(workspace) && (!bias || al16p(bias))) {
    int nblk, nslab; long long per;
    finish_stats_grid(p.M, cout, 3, bn->partial_floats, &nblk, &per, &nslab);
    if ((size_t)nblk * 3 * cout <= bn->partial_floats) {
      StatSink ss{};
      ss.partial = bn->partial; ss.count = bn->count; ss.unit = bn->count_unit; ss.tiles = nblk;
      SG2IM_LAUNCH(splitk_finish_stats_kernel<1>, dim3(nblk, nslab), dim3(256), 0, stream, workspace, nsplit, (long long)p.M,
                         cout, out, ld_out, bias, out_slope, per, ss);
      if (hipGetLastError() != hipSuccess) return SG2IM_ERR_HIP;
      return bn_stats_finish_tiles(bn->partial, nblk, per, p.M, cout, bn, stream);
    }
  }
  if (finish_split(p.e, p.M, cout, stream) != hipSuccess) return SG2IM_ERR_HIP;
  return bn ? bn_fwd_standalone(bn, out, p.M, cout, ld_out, stream) : SG2IM_OK;
}

int sg2im_conv_halo_unsplit(int batch, int h, int w, int cols, int chunks) {
  HaloPlan hp;
  if (!g_halo || batch < 1 || h < 1 || w < 1 || cols < 1 || chunks < 1) return 0;
  if (!halo_plan(batch, h, w, cols, chunks, (size_t)1 << 40, true, &hp)) return 0;
  return hp.nsplit == 1 ? 1 : 0;
}

int sg2im_conv2d_forward(const sg2im_conv_desc* d, const float* weight, int cout, const float* bias,
                         float out_slope, float* out, long long ld_out, int accumulate,
                         float* workspace, size_t workspace_bytes, hipStream_t stream) {
  return conv_forward_impl(d, weight, cout, bias, out_slope, out, ld_out, accumulate, workspace, workspace_bytes, nullptr,
                           stream);
}

int sg2im_conv2d_forward_bn(const sg2im_conv_desc* d, const float* weight, int cout, const float* bias,
                            float out_slope, float* out, long long ld_out, float* workspace, size_t workspace_bytes,
                            const sg2im_bn_fwd* bn, hipStream_t stream) {
  if (!bn || !bn->mean || !bn->invstd || !bn->scale || !bn->shift) return SG2IM_ERR_ARG;
  if (bn->training && !bn->partial) return SG2IM_ERR_ARG;
  if (!bn->training && (!bn->running_mean || !bn->running_var)) return SG2IM_ERR_ARG;
  return conv_forward_impl(d, weight, cout, bias, out_slope, out, ld_out, 0, workspace, workspace_bytes, bn, stream);
}

// bb != nullptr: followed by the reductions + coefficient set-up of a BatchNorm backward over dx
// (sg2im_conv2d_backward_data_bn)
// am: optional activation mask of the result (sg2im_conv2d_backward_data_act; never together with bb)
struct ActMask { const float* act; long long ld; float slope; };
static int conv_dgrad_impl(const sg2im_conv_desc* d, const float* weight, int cout, const float* dy,
                           int ld_dy, int c_begin, int c_count, float* dx, long long ld_dx,
                           int accumulate, float* workspace, size_t workspace_bytes, const sg2im_bn_bwd* bb,
                           hipStream_t stream, const ActMask* am = nullptr) {
  if (!d || !weight || !dy || !dx || cout < 1 || c_count < 1 || ld_dy < cout) return SG2IM_ERR_ARG;
  if (am && (bb || accumulate || !am->act || am->ld < c_count)) return SG2IM_ERR_ARG;
  if (check_desc(d)) return SG2IM_ERR_ARG;
  DgradParams p;
  p.st = StatSink{};
  // geometry (and Ctot) of the forward conv; s0 is then re-purposed to carry dY
  ConvGeom& g = p.g;
  fill_geom(g, d);
  g.nsrc = 1;
  for (int i = 0; i < 4; ++i) *src_at(g, i) = Src{nullptr, nullptr, nullptr, nullptr, 1.f, 0, 0, 0, 0};
  g.s0.p = dy; g.s0.C = cout; g.s0.ld = ld_dy;
  if (c_begin < 0 || c_begin + c_count > g.Ctot) return SG2IM_ERR_ARG;
  // (dY is read through a buffer resource with an out-of-range offset of 2 GiB for invalid rows)
  if ((double)d->batch * d->out_h * d->out_w * ld_dy * 4.0 >= 2147483648.0) return SG2IM_ERR_ARG;
  p.Wt = weight; p.c_begin = c_begin; p.Nc = c_count;
  const long long Mfull = (long long)d->batch * d->in_h * d->in_w;
  if (Mfull == 0) return bb ? SG2IM_ERR_ARG : SG2IM_OK;
  const int taps = d->kh * d->kw;
  // the standalone first two passes of the BatchNorm backward over the finished dx (launches that cannot fold them)
  auto bn_after = [&]() -> int {
    if (!bb) return SG2IM_OK;
    const int f = bb->pool2 ? 2 : 1;
    return bn_bwd_standalone(dx, ld_dx, bb->pool2, d->batch, d->in_h / f, d->in_w / f, c_count, bb, stream);
  };
  // launches that cannot apply the activation mask themselves (few-channel gathers, the stride-2 parity form): the
  // elementwise pass over the finished dx (in place; needs a dense dx)
  auto mask_after = [&]() -> int {
    if (!am) return SG2IM_OK;
    if (ld_dx != c_count) return SG2IM_ERR_ARG;
    return sg2im_act_backward(dx, ld_dx, 0, d->batch, d->in_h, d->in_w, am->act, am->ld, c_count, am->slope, dx, stream);
  };
  // 1 x 1 with at most four OUTPUT channels (output_conv[2], mask_net's last layer): one elementwise pass (conv_fewout.h)
  // (bf16 operands: cout == 4 is a shape the bf16 mode is SPECIFIED to round - oracle conv2d rd / rw - so it stays on the
  // matrix-core path there)
  if (g_fewout && taps == 1 && d->stride == 1 && d->pad == 0 && (cout < 4 || (cout == 4 && d->compute_dtype == 0)) && !bb && !accumulate && !(c_count & 3) &&
      !(c_begin & 3) && !(g.Wtap & 3) && !(ld_dx & 3) && !((uintptr_t)dx & 15) && !((uintptr_t)weight & 15) &&
      (!am || (!(am->ld & 3) && !((uintptr_t)am->act & 15)))) {
    const int c4n = c_count >> 2;
    const dim3 fgrid((unsigned)((Mfull * c4n + 255) / 256));
#define SG2IM_FEWOUT_DG(CO_)                                                                                               \
    SG2IM_LAUNCH((conv1x1_fewout_dgrad_kernel<CO_>), fgrid, dim3(256), 0, stream, dy, ld_dy, weight, g.Wtap, c_begin, c4n, \
                       Mfull, dx, ld_dx, am ? am->act : nullptr, am ? am->ld : 0LL, am ? am->slope : 1.f)
    if (cout == 1) SG2IM_FEWOUT_DG(1); else if (cout == 2) SG2IM_FEWOUT_DG(2); else if (cout == 3) SG2IM_FEWOUT_DG(3);
    else SG2IM_FEWOUT_DG(4);
#undef SG2IM_FEWOUT_DG
    return hipGetLastError() == hipSuccess ? SG2IM_OK : SG2IM_ERR_HIP;
  }
  if (c_count <= 4 && (size_t)taps * cout * c_count * sizeof(float) <= 48 * 1024) {
    const size_t lds = (size_t)taps * cout * c_count * sizeof(float);
    // at most 2 x 2 taps per pixel, 16-byte dY loads
    const bool quad = (d->kh + d->stride - 1) / d->stride <= 2 && (d->kw + d->stride - 1) / d->stride <= 2 &&
                      cout % 4 == 0 && ld_dy % 4 == 0 && !((uintptr_t)dy & 15) && !g_fewc_scalar;
    if (quad) {
      dim3 qgrid((unsigned)((Mfull + 63) / 64));
#define SG2IM_FEWQ(NC)                                                                                        \
      SG2IM_LAUNCH((conv_dgrad_fewc_quad_kernel<NC>), qgrid, dim3(256), lds, stream, dy, ld_dy, weight, cout, \
                         g.Wtap, c_begin, d->batch, d->in_h, d->in_w, d->out_h, d->out_w, d->kh, d->kw,      \
                         d->stride, d->pad, dx, ld_dx, accumulate)
      if (c_count == 1) SG2IM_FEWQ(1); else if (c_count == 2) SG2IM_FEWQ(2); else if (c_count == 3) SG2IM_FEWQ(3);
      else SG2IM_FEWQ(4);
#undef SG2IM_FEWQ
      if (am) return hipGetLastError() == hipSuccess ? mask_after() : SG2IM_ERR_HIP;
      return hipGetLastError() == hipSuccess ? bn_after() : SG2IM_ERR_HIP;
    }
    dim3 grid((unsigned)((Mfull + 255) / 256));
#define SG2IM_FEWC(NC)                                                                                        \
    SG2IM_LAUNCH((conv_dgrad_fewc_kernel<NC>), grid, dim3(256), lds, stream, dy, ld_dy, weight, cout,   \
                       g.Wtap, c_begin, d->batch, d->in_h, d->in_w, d->out_h, d->out_w, d->kh, d->kw,        \
                       d->stride, d->pad, dx, ld_dx, accumulate)
    if (c_count == 1) SG2IM_FEWC(1); else if (c_count == 2) SG2IM_FEWC(2); else if (c_count == 3) SG2IM_FEWC(3);
    else SG2IM_FEWC(4);
#undef SG2IM_FEWC
    if (am) return hipGetLastError() == hipSuccess ? mask_after() : SG2IM_ERR_HIP;
    return hipGetLastError() == hipSuccess ? bn_after() : SG2IM_ERR_HIP;
  }
  const bool va4 = (cout % 4 == 0) && (ld_dy % 4 == 0) && !((uintptr_t)dy & 15);
  const bool vb4 = (g.Ctot % 4 == 0) && (c_begin % 4 == 0) && (c_count % 4 == 0) && !((uintptr_t)weight & 15);
  // 3x3 / stride 1 / pad 1: the halo'd-tile kernel (conv_halo.h)
  HaloPlan hp;
  // bfloat16 storage of dy / dx / the BatchNorm'd layer's y: the bf16 halo'd kernel, no split-K (see conv_forward_impl)
  const bool sbd = d->dy_dtype == 1 || d->out_dtype == 1 || (bb && bb->y_dtype == 1);
  const bool va4h = d->dy_dtype == 1 ? ((cout % 4 == 0) && (ld_dy % 4 == 0) && !((uintptr_t)dy & 7)) : va4;
  if (sbd && (d->compute_dtype != 1 || accumulate || am)) return SG2IM_ERR_ARG;
  if (va4h && vb4 && !accumulate && halo_geometry(d) && g.Wtap % 4 == 0 &&
      halo_plan(d->batch, d->in_h, d->in_w, c_count, (cout + BK - 1) / BK, workspace_bytes, workspace != nullptr && !sbd, &hp)) {
    HaloParams q;
    q.g = g; q.Wt = weight; q.Wh = (const bf16_t*)d->weight_bf16; q.N = c_count; q.c_begin = c_begin; q.nchunks = (cout + BK - 1) / BK;
    q.tiles_x = d->in_w / hp.ct; q.tiles_y = d->in_h / hp.rt; q.M = (int)Mfull;
    q.e = Epi{dx, ld_dx, nullptr, 1.f, 0, workspace, hp.nsplit};
    q.e.out_bf = d->out_dtype == 1;
    q.dy_bf = d->dy_dtype == 1;
    if (am) { q.e.mask = am->act; q.e.ld_mask = am->ld; q.e.mask_slope = am->slope; }     // (epilogue, or the split-K finish)
    q.st = StatSink{};
    const long long bn_rows_h = bb ? (bb->pool2 ? Mfull / 4 : Mfull) : 0;
    const bool st_h = g_fuse_bn && bb && !bb->count && bb->partial && al16p(bb->partial) &&
                      (!bb->pool2 || (d->in_h % 2 == 0 && d->in_w % 2 == 0));
    StatSink ss{};
    if (st_h) {
      ss.partial = bb->partial; ss.y = bb->y; ss.ld_y = bb->ld_y; ss.mean = bb->mean; ss.invstd = bb->invstd;
      ss.scale = bb->scale; ss.shift = bb->shift; ss.slope = bb->slope; ss.pool2 = bb->pool2; ss.H = d->in_h; ss.W = d->in_w;
      ss.y_bf = bb->y_dtype == 1;
    }
    if (sbd && bb && !(st_h && (size_t)hp.patches * 2 * c_count <= bb->partial_floats)) return SG2IM_ERR_ARG;
    if (st_h && hp.nsplit == 1 && (size_t)hp.patches * 2 * c_count <= bb->partial_floats) {
      q.st = ss; q.st.tiles = hp.patches;
      if (launch_halo<true, true>(q, hp, stream, d->compute_dtype == 1) != hipSuccess) return SG2IM_ERR_HIP;
      return bn_bwd_finish_tiles(bb->partial, hp.patches, bn_rows_h, c_count, bb, stream);
    }
    if (launch_halo<true, false>(q, hp, stream, d->compute_dtype == 1) != hipSuccess) return SG2IM_ERR_HIP;
    if (st_h && hp.nsplit > 1 && ld_dx % 4 == 0 && al16p(dx) && al16p(workspace) && bb->ld_y % 4 == 0 && al16p(bb->y) &&
        al16p(bb->mean) && al16p(bb->invstd) && al16p(bb->scale) && al16p(bb->shift)) {
      int nblk, nslab; long long per;
      finish_stats_grid(Mfull, c_count, 2, bb->partial_floats, &nblk, &per, &nslab);
      if ((size_t)nblk * 2 * c_count <= bb->partial_floats) {
        ss.tiles = nblk;
        SG2IM_LAUNCH(splitk_finish_stats_kernel<2>, dim3(nblk, nslab), dim3(256), 0, stream, workspace, hp.nsplit, Mfull,
                           c_count, dx, ld_dx, (const float*)nullptr, 1.f, per, ss);
        if (hipGetLastError() != hipSuccess) return SG2IM_ERR_HIP;
        return bn_bwd_finish_tiles(bb->partial, nblk, bn_rows_h, c_count, bb, stream);
      }
    }
    if (finish_split(q.e, Mfull, c_count, stream) != hipSuccess) return SG2IM_ERR_HIP;
    return bn_after();
  }
  if (sbd) return SG2IM_ERR_ARG;
  // stride-2 parity decomposition: needs the chunked (VA=4) K enumeration; split-K partials of
  // this form are laid out per class and finished by splitk_finish_parity_kernel
  p.parity = (d->stride == 2 && va4 && d->kh >= 2 && d->kw >= 2) ? 1 : 0;
  if (va4) { p.nch = (cout + BK - 1) / BK; p.iters = taps * p.nch; }
  else { p.nch = 0; p.iters = (taps * cout + BK - 1) / BK; }
  long long Mrows = Mfull;
  int iters_eff = p.iters;
  if (p.parity) {
    Mrows = (long long)d->batch * ((d->in_h + 1) / 2) * ((d->in_w + 1) / 2);     // largest class
    iters_eff = ((d->kh + 1) / 2) * ((d->kw + 1) / 2) * p.nch;
  }
  p.M = (int)Mrows;
  const long long MNslab = p.parity ? 4 * Mrows * c_count : Mfull * c_count;      // floats per split
  Plan pl = make_plan(PASS_DGRAD, Mrows, c_count, iters_eff, MNslab, workspace_bytes,
                      workspace != nullptr, 4, !va4,
                      // (the four parity classes are launched together: 4x the workgroups)
                      [&](int bn) { return (long long)(c_count + bn - 1) / bn * (p.parity ? 4 : 1); });
  if (va4 && !vb4 && (pl.tile == 0 || pl.tile == 3)) {      // narrow scalar-B outputs: 64-wide tiles only
    const int t = pl.tile == 0 ? 1 : 2;
    pl = Plan{t, kBM[t], kBN[t], 1, 0};
    if (workspace)
      pl.nsplit = split_for(PASS_DGRAD, t, ((Mrows + pl.bm - 1) / pl.bm) * ((c_count + 63) / 64) * (p.parity ? 4 : 1),
                            iters_eff, MNslab, workspace_bytes, 4);
  }
  p.e = Epi{dx, ld_dx, nullptr, 1.f, accumulate, workspace, pl.nsplit};
  if (am && !p.parity) { p.e.mask = am->act; p.e.ld_mask = am->ld; p.e.mask_slope = am->slope; }
  hipError_t err;
  // BatchNorm-backward sums of dx from the same launches (data-gradient epilogue or split-K finish)
  const long long bn_rows = bb ? (bb->pool2 ? Mfull / 4 : Mfull) : 0;
  const bool st_ok = g_fuse_bn && bb && va4 && vb4 && !p.parity && !accumulate && bb->partial && al16p(bb->partial) &&
                     (!bb->pool2 || (d->in_h % 2 == 0 && d->in_w % 2 == 0));
  StatSink ss{};
  if (st_ok) {
    ss.partial = bb->partial; ss.count = bb->count; ss.unit = bb->count_unit * (bb->pool2 ? 4 : 1);
    ss.y = bb->y; ss.ld_y = bb->ld_y; ss.mean = bb->mean; ss.invstd = bb->invstd; ss.scale = bb->scale; ss.shift = bb->shift;
    ss.slope = bb->slope; ss.pool2 = bb->pool2; ss.H = d->in_h; ss.W = d->in_w;
  }
  if (st_ok && pl.nsplit == 1 && (size_t)((Mfull + pl.bm - 1) / pl.bm) * 2 * c_count <= bb->partial_floats) {
    p.st = ss;
    p.st.tiles = (int)((Mfull + pl.bm - 1) / pl.bm);
    const bool hb = d->compute_dtype == 1;
#define SG2IM_ST(BM_, BN_) (hb ? launch_dgrad_st<BM_, BN_, true>(p, stream) : launch_dgrad_st<BM_, BN_, false>(p, stream))
    err = pl.tile == 0 ? SG2IM_ST(128, 128) : pl.tile == 1 ? SG2IM_ST(128, 64) : pl.tile == 2 ? SG2IM_ST(64, 64) : SG2IM_ST(64, 128);
#undef SG2IM_ST
    if (err != hipSuccess) return SG2IM_ERR_HIP;
    return bn_bwd_finish_tiles(bb->partial, (int)((Mfull + pl.bm - 1) / pl.bm), bn_rows, c_count, bb, stream);
  }
  if (va4 && vb4 && d->compute_dtype == 1) {
    err = pl.tile == 0 ? launch_dgrad_h<128, 128>(p, stream) : pl.tile == 1 ? launch_dgrad_h<128, 64>(p, stream)
        : pl.tile == 2 ? launch_dgrad_h<64, 64>(p, stream) : launch_dgrad_h<64, 128>(p, stream);
  } else if (va4 && vb4) {
    err = pl.tile == 0 ? launch_dgrad<128, 128, 4, 4>(p, stream) : pl.tile == 1 ? launch_dgrad<128, 64, 4, 4>(p, stream)
        : pl.tile == 2 ? launch_dgrad<64, 64, 4, 4>(p, stream) : launch_dgrad<64, 128, 4, 4>(p, stream);
  } else if (va4) {
    err = pl.tile == 2 ? launch_dgrad<64, 64, 4, 1>(p, stream) : launch_dgrad<128, 64, 4, 1>(p, stream);
  } else {
    err = launch_dgrad<64, 64, 1, 1>(p, stream);
  }
  if (err != hipSuccess) return SG2IM_ERR_HIP;
  if (p.parity && pl.nsplit > 1) {
    const long long per_split = 4LL * p.M * c_count;
    const int blocks = (int)std::min<long long>((per_split + 255) / 256, 4096);
    SG2IM_LAUNCH(splitk_finish_parity_kernel, dim3(blocks), dim3(256), 0, stream, workspace, pl.nsplit, p.M,
                       c_count, dx, ld_dx, accumulate, d->batch, d->in_h, d->in_w);
    if (am) return hipGetLastError() == hipSuccess ? mask_after() : SG2IM_ERR_HIP;
    return hipGetLastError() == hipSuccess ? bn_after() : SG2IM_ERR_HIP;
  }
  if (p.parity && am) return mask_after();                  // (parity form without split-K: its epilogue has no mask)
  if (st_ok && pl.nsplit > 1 && ld_dx % 4 == 0 && al16p(dx) && al16p(workspace) && bb->ld_y % 4 == 0 && al16p(bb->y) &&
      al16p(bb->mean) && al16p(bb->invstd) && al16p(bb->scale) && al16p(bb->shift)) {
    int nblk, nslab; long long per;
    finish_stats_grid(Mfull, c_count, 2, bb->partial_floats, &nblk, &per, &nslab);
    if ((size_t)nblk * 2 * c_count <= bb->partial_floats) {
      ss.tiles = nblk;
      SG2IM_LAUNCH(splitk_finish_stats_kernel<2>, dim3(nblk, nslab), dim3(256), 0, stream, workspace, pl.nsplit, Mfull, c_count,
                         dx, ld_dx, (const float*)nullptr, 1.f, per, ss);
      if (hipGetLastError() != hipSuccess) return SG2IM_ERR_HIP;
      return bn_bwd_finish_tiles(bb->partial, nblk, bn_rows, c_count, bb, stream);
    }
  }
  if (finish_split(p.e, Mfull, c_count, stream) != hipSuccess) return SG2IM_ERR_HIP;
  return bn_after();
}

int sg2im_conv2d_backward_data(const sg2im_conv_desc* d, const float* weight, int cout, const float* dy,
                               int ld_dy, int c_begin, int c_count, float* dx, long long ld_dx,
                               int accumulate, float* workspace, size_t workspace_bytes,
                               hipStream_t stream) {
  return conv_dgrad_impl(d, weight, cout, dy, ld_dy, c_begin, c_count, dx, ld_dx, accumulate, workspace, workspace_bytes,
                         nullptr, stream);
}

int sg2im_conv2d_backward_data_bn(const sg2im_conv_desc* d, const float* weight, int cout, const float* dy,
                                  int ld_dy, int c_begin, int c_count, float* dx, long long ld_dx,
                                  float* workspace, size_t workspace_bytes, const sg2im_bn_bwd* bb, hipStream_t stream) {
  if (!bb || !bb->y || !bb->mean || !bb->invstd || !bb->scale || !bb->shift || !bb->coef || !bb->partial) return SG2IM_ERR_ARG;
  if (bb->ld_y < c_count) return SG2IM_ERR_ARG;
  return conv_dgrad_impl(d, weight, cout, dy, ld_dy, c_begin, c_count, dx, ld_dx, 0, workspace, workspace_bytes, bb, stream);
}

int sg2im_conv2d_backward_data_act(const sg2im_conv_desc* d, const float* weight, int cout, const float* dy,
                                   int ld_dy, int c_begin, int c_count, float* dx, long long ld_dx, const float* act,
                                   long long ld_act, float slope, float* workspace, size_t workspace_bytes,
                                   hipStream_t stream) {
  const ActMask am{act, ld_act, slope};
  return conv_dgrad_impl(d, weight, cout, dy, ld_dy, c_begin, c_count, dx, ld_dx, 0, workspace, workspace_bytes, nullptr,
                         stream, &am);
}

int sg2im_conv2d_backward_weight(const sg2im_conv_desc* d, const float* dy, int ld_dy, int cout,
                                 float* dweight, float* dbias, int accumulate, float* workspace,
                                 size_t workspace_bytes, hipStream_t stream) {
  if (check_desc(d) || !dy || !dweight || cout < 1 || ld_dy < cout) return SG2IM_ERR_ARG;
  WgradParams p;
  fill_geom(p.g, d);
  p.dY = dy; p.ldy = ld_dy; p.Cout = cout;
  p.background = d->launch_hints & SG2IM_HINT_BACKGROUND;
  p.P = d->batch * d->out_h * d->out_w;
  if ((double)p.P * ld_dy * 4.0 >= 2147483648.0) return SG2IM_ERR_ARG;     // (buffer-resource reads of dY, see above)
  const int taps = d->kh * d->kw;
  const int Ntot = taps * p.g.Ctot;
  if (p.P == 0) {
    if (!accumulate && hipMemsetAsync(dweight, 0, sizeof(float) * (size_t)cout * taps * p.g.Wtap, stream) != hipSuccess)
      return SG2IM_ERR_HIP;
    if (!accumulate && dbias && hipMemsetAsync(dbias, 0, sizeof(float) * (size_t)cout, stream) != hipSuccess)
      return SG2IM_ERR_HIP;
    return SG2IM_OK;
  }
  // 1 x 1 with at most four OUTPUT channels over ONE plain NHWC source: a two-stage column reduction (conv_fewout.h)
  {
    const Src& s0 = p.g.s0;
    const int C4 = s0.C >> 2;
    if (g_fewout && taps == 1 && d->stride == 1 && d->pad == 0 && (cout < 4 || (cout == 4 && d->compute_dtype == 0)) && p.g.nsrc == 1 && !s0.gidx && !s0.up &&
        !(s0.C & 3) && C4 >= 4 && C4 <= 64 && !(C4 & (C4 - 1)) && !(s0.ld & 3) && !((uintptr_t)s0.p & 15) &&
        (!s0.scale || (s0.shift && !((uintptr_t)s0.scale & 15) && !((uintptr_t)s0.shift & 15))) && workspace) {
      const long long M = p.P;
      long long nblk = std::min<long long>(1024, (M + 127) / 128);
      const long long per = (M + nblk - 1) / nblk;
      nblk = (M + per - 1) / per;
      if (sizeof(float) * (size_t)nblk * cout * (s0.C + 1) <= workspace_bytes) {
#define SG2IM_FEWOUT_WG(CO_)                                                                                             \
        SG2IM_LAUNCH((conv1x1_fewout_wgrad_kernel<CO_>), dim3((unsigned)nblk), dim3(256), 0, stream, s0.p, (long long)s0.ld, \
                           s0.scale, s0.shift, s0.slope, C4, dy, ld_dy, M, per, (int)nblk, workspace)
        if (cout == 1) SG2IM_FEWOUT_WG(1); else if (cout == 2) SG2IM_FEWOUT_WG(2); else if (cout == 3) SG2IM_FEWOUT_WG(3);
        else SG2IM_FEWOUT_WG(4);
#undef SG2IM_FEWOUT_WG
        const int nel = cout * s0.C + cout;
        SG2IM_LAUNCH(conv1x1_fewout_wgrad_finish_kernel, dim3(nel), dim3(64), 0, stream, workspace, (int)nblk,
                           cout, s0.C, dweight, p.g.Wtap, dbias, accumulate);
        return hipGetLastError() == hipSuccess ? SG2IM_OK : SG2IM_ERR_HIP;
      }
    }
  }
  p.iters = (p.P + BK - 1) / BK;
  // bfloat16 storage of a source or of dy: the bf16 halo'd weight-gradient kernel only (else refused, see below)
  bool sbw = d->dy_dtype == 1;
  for (int i = 0; i < d->nsrc; ++i) sbw = sbw || d->src[i].dtype == 1;
  const bool v4 = geom_vec4(p.g) && (cout % 4 == 0) && (ld_dy % 4 == 0) && !((uintptr_t)dy & (d->dy_dtype == 1 ? 7 : 15));
  const int Ctot = p.g.Ctot;
  // (room for the bias-gradient partials of up to 512 splits is kept behind the dW partials)
  const size_t bias_room = dbias ? sizeof(float) * 512 * (size_t)cout : 0;
  const bool can_split = workspace != nullptr && workspace_bytes > bias_room;
  // 3x3 / stride 1 / pad 1 over maps that 64-pixel patches tile: the halo'd-tile kernel (wgrad_halo.h)
  // (bf16 operands, round 5: conv_wgrad_halo_h_kernel, 4 x 16 patches only; SG2IM_WGRAD_HALO_BF16=0: the per-tap bf16 kernels)
  static const bool g_wgrad_halo_h = !(getenv("SG2IM_WGRAD_HALO_BF16") && atoi(getenv("SG2IM_WGRAD_HALO_BF16")) == 0);
  const bool hb = d->compute_dtype == 1;
  if (sbw && !hb) return SG2IM_ERR_ARG;
  if (g_wgrad_halo && v4 && !any_gather(p.g) && halo_geometry(d) &&
      ((d->in_w % 16 == 0 && d->in_h % 4 == 0) || (!hb && d->in_w % 8 == 0 && d->in_h % 8 == 0)) && (!hb || g_wgrad_halo_h)) {
    const bool wide = d->in_w % 16 == 0;
    WgHaloParams q;
    q.g = p.g; q.dY = dy; q.ldy = ld_dy; q.Cout = cout; q.dy_bf = d->dy_dtype == 1;
    q.tiles_x = d->in_w / (wide ? 16 : 8); q.tiles_y = d->in_h / (wide ? 4 : 8);
    q.npatch = d->batch * q.tiles_x * q.tiles_y;
    const int ncb = (Ctot + 63) / 64, nkb = (cout + 63) / 64;
    // K split over whole patches.  One workgroup per CU is resident (324 registers per lane), so the launch runs in
    // ceil(blocks / #CU) rounds of `per` patches each (+ ~1 patch worth of prologue / epilogue, + the finish launch
    // growing with the split count): take the split count that minimises that
    const long long tiles = (long long)ncb * nkb;
    long long cap = can_split ? std::max<long long>(1, (long long)((workspace_bytes - bias_room) / sizeof(float)) / ((long long)cout * Ntot)) : 1;
    cap = std::min<long long>(std::min<long long>(cap, 512), q.npatch);
    long long ns = 1;
    double best = -1.0;
    for (long long t = 1; t <= cap; ++t) {
      const long long per = (q.npatch + t - 1) / t;
      if ((q.npatch + per - 1) / per != t) continue;            // every split non-empty
      const long long rounds = (tiles * t + g_num_cu - 1) / g_num_cu;
      const double cost = (double)rounds * ((double)per + 1.0) + 0.02 * (double)t;
      if (best < 0 || cost < best) { best = cost; ns = t; }
    }
    q.per = (int)((q.npatch + ns - 1) / ns);
    // [measured, profiles/r3_wgrad_halo.log] it wins where a workgroup gets enough patches to amortise its prologue and
    // its 147 KB of partial sums, and loses on the widest concats (dY is re-read once per 64-channel block):
    // m4.conv0 97 -> 106, the 64-channel 64x64 layers 76 -> 89, m2.conv1 84 -> 94, m1.conv1 80 -> 92 TFLOP/s;
    // m2.conv0 (672 channels) 97 -> 93, m1.conv0 (1184) 87 -> 86, mask_net's 8x8 layer (4 patches each) 67 -> 61
    // bf16: the alternative re-reads and re-activates the fp32 input once per tap for 1/8 of the MFMA instructions - the
    // halo'd form wherever a workgroup gets at least two patches
    const bool halo_pays = sbw || (hb ? q.per >= 2 : (Ctot <= 512 && q.per >= 6));
    const int nsplit = (q.npatch + q.per - 1) / q.per;
    q.e = Epi{dweight, (long long)taps * p.g.Wtap, nullptr, 1.f, accumulate, workspace, nsplit};
    if (p.g.Wtap != Ctot) { q.e.col_ctot = Ctot; q.e.col_wtap = p.g.Wtap; }
    q.dbias = dbias;
    q.ws_bias = nsplit > 1 ? workspace + (size_t)nsplit * cout * Ntot : nullptr;
    if (halo_pays || g_wgrad_halo >= 2) {
      if (g_plan_debug) fprintf(stderr, "[sg2im wgrad halo] Cout=%d Ctot=%d patches=%d -> %s blocks %dx%d x%d\n", cout, Ctot, q.npatch,
                                wide ? "4x16" : "8x8", ncb, nkb, nsplit);
      dim3 grid(ncb, nkb, nsplit);
      if (prepare_wgrad_halo() != hipSuccess) return SG2IM_ERR_HIP;
      if (hb) {
        // storage of the sources: all float32 / all bfloat16 / mixed (template forms of the kernel, wgrad_halo.h)
        int nbf = 0;
        for (int i = 0; i < p.g.nsrc; ++i) nbf += src_at(p.g, i)->bf ? 1 : 0;
        const int xb = nbf == 0 ? 0 : (nbf == p.g.nsrc ? 1 : 2);
#define SG2IM_WGH(XB_, YB_) SG2IM_LAUNCH((conv_wgrad_halo_h_kernel<XB_, YB_>), grid, dim3(NTHREADS), wgrad_halo_h_lds(), stream, q)
        if (q.dy_bf) { if (xb == 0) SG2IM_WGH(0, true); else if (xb == 1) SG2IM_WGH(1, true); else SG2IM_WGH(2, true); }
        else { if (xb == 0) SG2IM_WGH(0, false); else if (xb == 1) SG2IM_WGH(1, false); else SG2IM_WGH(2, false); }
#undef SG2IM_WGH
      }
      else if (wide) SG2IM_LAUNCH((conv_wgrad_halo_kernel<4, 16>), grid, dim3(NTHREADS), (wgrad_halo_lds<4, 16>()), stream, q);
      else SG2IM_LAUNCH((conv_wgrad_halo_kernel<8, 8>), grid, dim3(NTHREADS), (wgrad_halo_lds<8, 8>()), stream, q);
      if (hipGetLastError() != hipSuccess) return SG2IM_ERR_HIP;
      return finish_split(q.e, cout, Ntot, stream, q.ws_bias, dbias, cout) == hipSuccess ? SG2IM_OK : SG2IM_ERR_HIP;
    }
  }
  if (sbw) return SG2IM_ERR_ARG;       // (bf16 storage outside the bf16 halo'd kernel's geometry: not supported)
  const Plan pl = make_plan(PASS_WGRAD, cout, Ntot, p.iters, (long long)cout * Ntot, can_split ? workspace_bytes - bias_room : 0,
                            can_split, 2, !v4, [&](int bn) { return (long long)(Ntot + bn - 1) / bn; });
  p.ntile_c = 0;
  const int ntiles_n = (Ntot + pl.bn - 1) / pl.bn;
  p.e = Epi{dweight, (long long)taps * p.g.Wtap, nullptr, 1.f, accumulate, workspace, pl.nsplit};
  if (p.g.Wtap != Ctot) { p.e.col_ctot = Ctot; p.e.col_wtap = p.g.Wtap; }     // (see Epi: per-tap destination stride)
  p.dbias = dbias;
  p.ws_bias = pl.nsplit > 1 ? workspace + (size_t)pl.nsplit * cout * Ntot : nullptr;
  hipError_t err;
  if (v4 && d->compute_dtype == 1 && !any_gather(p.g)) {
    err = pl.tile == 0 ? launch_wgrad_h<128, 128>(p, ntiles_n, stream)
        : pl.tile == 1 ? launch_wgrad_h<128, 64>(p, ntiles_n, stream)
        : pl.tile == 2 ? launch_wgrad_h<64, 64>(p, ntiles_n, stream)
                       : launch_wgrad_h<64, 128>(p, ntiles_n, stream);
  } else if (v4) {
    err = pl.tile == 0 ? launch_wgrad<128, 128, 4>(p, ntiles_n, stream)
        : pl.tile == 1 ? launch_wgrad<128, 64, 4>(p, ntiles_n, stream)
        : pl.tile == 2 ? launch_wgrad<64, 64, 4>(p, ntiles_n, stream)
                       : launch_wgrad<64, 128, 4>(p, ntiles_n, stream);
  } else {
    err = launch_wgrad<64, 64, 1>(p, ntiles_n, stream);
  }
  if (err != hipSuccess) return SG2IM_ERR_HIP;
  return finish_split(p.e, cout, Ntot, stream, p.ws_bias, dbias, cout) == hipSuccess ? SG2IM_OK : SG2IM_ERR_HIP;
}

// Up to kGroupMax weight gradients (+ their bias gradients) as one launch + one finish launch: the four
// linear layers of a GraphTripleConv layer.  Every problem must be float4-loadable (channels, ld_dy, cout
// multiples of 4, 16-byte aligned buffers) and small enough for 64x64 tiles to be the right shape; returns
// SG2IM_ERR_ARG without launching anything otherwise (the caller then uses sg2im_conv2d_backward_weight).
int sg2im_conv2d_backward_weight_group(int n, const sg2im_conv_desc* const* descs, const float* const* dys,
                                       const int* ld_dys, const int* couts, float* const* dweights,
                                       float* const* dbiases, int accumulate, float* workspace,
                                       size_t workspace_bytes, hipStream_t stream) {
  if (n < 1 || n > kGroupMax || !descs || !dys || !ld_dys || !couts || !dweights || !dbiases) return SG2IM_ERR_ARG;
  WgradGroup wg;
  FinishGroup fg;
  int blocks = 0, fblocks = 0;
  size_t ws_off = 0;                                   // floats
  bool any_split = false;
  wg.first[0] = fg.first[0] = 0;
  for (int i = 0; i < kGroupMax; ++i) {
    if (i >= n) { wg.p[i] = wg.p[0]; wg.first[i + 1] = wg.first[i]; fg.a[i] = FinishArgs{}; fg.first[i + 1] = fg.first[i]; continue; }
    const sg2im_conv_desc* d = descs[i];
    const int cout = couts[i], ld_dy = ld_dys[i];
    if (check_desc(d) || !dys[i] || !dweights[i] || cout < 1 || ld_dy < cout || d->compute_dtype != 0) return SG2IM_ERR_ARG;
    WgradParams& p = wg.p[i];
    fill_geom(p.g, d);
    p.dY = dys[i]; p.ldy = ld_dy; p.Cout = cout; p.background = 0;
    p.P = d->batch * d->out_h * d->out_w;
    if (p.P == 0 || (double)p.P * ld_dy * 4.0 >= 2147483648.0) return SG2IM_ERR_ARG;
    const bool v4 = geom_vec4(p.g) && (cout % 4 == 0) && (ld_dy % 4 == 0) && !((uintptr_t)dys[i] & 15) &&
                    !((uintptr_t)dweights[i] & 15);
    if (!v4 || p.g.Wtap != p.g.Ctot) return SG2IM_ERR_ARG;
    const int taps = d->kh * d->kw, Ntot = taps * p.g.Ctot;
    p.iters = (p.P + BK - 1) / BK;
    const size_t bias_room = dbiases[i] ? 512 * (size_t)cout : 0;                 // floats
    const size_t avail = workspace && workspace_bytes / sizeof(float) > ws_off + bias_room
                             ? workspace_bytes / sizeof(float) - ws_off - bias_room : 0;
    const bool can_split = avail > 0;
    const Plan pl = make_plan(PASS_WGRAD, cout, Ntot, p.iters, (long long)cout * Ntot, avail * sizeof(float), can_split, 2,
                              true, [&](int bn) { return (long long)(Ntot + bn - 1) / bn; });
    p.ntile_c = 0;
    p.ntiles_n = (Ntot + 63) / 64;
    p.ntiles_m = (cout + 63) / 64;
    float* ws_i = workspace ? workspace + ws_off : nullptr;
    p.e = Epi{dweights[i], (long long)Ntot, nullptr, 1.f, accumulate, ws_i, pl.nsplit};
    p.dbias = dbiases[i];
    p.ws_bias = pl.nsplit > 1 ? ws_i + (size_t)pl.nsplit * cout * Ntot : nullptr;
    blocks += p.ntiles_n * p.ntiles_m * pl.nsplit;
    wg.first[i + 1] = blocks;
    FinishArgs& a = fg.a[i];
    a = FinishArgs{};
    if (pl.nsplit > 1) {
      any_split = true;
      const long long MN = (long long)cout * Ntot;
      if (Ntot % 4 || ((uintptr_t)ws_i & 15)) return SG2IM_ERR_ARG;
      a = FinishArgs{ws_i, pl.nsplit, MN, Ntot, dweights[i], (long long)Ntot, nullptr, 1.f, accumulate, p.ws_bias, dbiases[i], cout};
      fblocks += (int)std::min<long long>((MN / 4 + 255) / 256, 1024);
      ws_off += ((size_t)pl.nsplit * cout * Ntot + (dbiases[i] ? (size_t)pl.nsplit * cout : 0) + 3) / 4 * 4;
    }
    fg.first[i + 1] = fblocks;
  }
  { hipError_t e = prepare_wgrad<64, 64, 4, true>(); if (e != hipSuccess) return SG2IM_ERR_HIP; }
  static bool group_ready = false;
  const size_t glds = wgrad_lds<64, 64>();
  if (!group_ready) {
    if (ensure_lds(conv_wgrad_group_kernel, glds) != hipSuccess) return SG2IM_ERR_HIP;
    group_ready = true;
  }
  SG2IM_LAUNCH(conv_wgrad_group_kernel, dim3(blocks), dim3(NTHREADS), glds, stream, wg);
  if (hipGetLastError() != hipSuccess) return SG2IM_ERR_HIP;
  if (any_split) {
    SG2IM_LAUNCH(splitk_finish_v4_group_kernel, dim3(fblocks), dim3(256), 0, stream, fg);
    if (hipGetLastError() != hipSuccess) return SG2IM_ERR_HIP;
  }
  return SG2IM_OK;
}

}  // extern "C"
