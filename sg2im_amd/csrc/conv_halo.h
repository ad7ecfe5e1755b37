// 3x3 / stride 1 / pad 1 convolutions (forward and data gradient) with a HALO'D INPUT TILE in LDS.
//
// The first-generation kernels (conv.hip) stage one BM x 32 operand tile per (tap, 32-channel chunk): every
// input element is fetched from global memory - and run through the pending BatchNorm affine + LeakyReLU -
// once per tap, nine times in all, and that loader work (address arithmetic, loads, packed VALU, LDS stores)
// in front of every 32 MFMAs is what kept them at ~0.6 of the fp32 matrix peak.
// Here a workgroup owns a 2-D PATCH of RT x CT = 128 output pixels of one image.  Per 32-channel chunk it
// stages the (RT + 2) x (CT + 2) pixel halo of the patch ONCE - loaded once, activated once - and issues the
// MFMAs of all nine taps from it: a tap is just a different base address of the A-fragment reads
// (halo pixel (r + kh, c + kw) for output pixel (r, c)).  Per MFMA the loader does 1/4.4 (2 x 64 patch) to
// 1/6.4 (8 x 16) of the global loads, affine VALU and LDS stores of the per-tap form; only the weights are
// still staged per (chunk, tap).
//
//   forward        out[n][y][x][co] = sum_{kh,kw,c} X[n][y + kh - 1][x + kw - 1][c] * W[co][kh][kw][c]
//   data gradient  dX[n][y][x][c]   = sum_{kh,kw,co} dY[n][y - kh + 1][x - kw + 1][co] * W[co][kh][kw][c]
// (the same kernel: the data gradient reads the halo at the mirrored tap offset and the weights k-major).
//
// LDS: halo image [halo pixel][32 + 4] floats (the m-major layout of igemm.h: ds_read_b128 fragments, conflict
// free) + one weight tile.  Pipeline: weights of the next (chunk, tap) and the halo of the next chunk are
// prefetched into registers while the matrix cores work; the nine taps are unrolled, so every tap's halo
// offset is an immediate of the fragment reads.
#pragma once
#include "igemm.h"
#include <utility>

#ifndef SG2IM_HALO_WAVES64
#define SG2IM_HALO_WAVES64 4
#endif

namespace sg2im {

// compile-time loop over the nine taps: f(TapC<0>{}), ..., f(TapC<8>{}) - the tap / register-slot index of the nine-tap
// staging is a constant in every use, so the per-tap register arrays never become indexable memory
template <int T> struct TapC { static constexpr int v = T; };
template <typename F, int... I> __device__ __forceinline__ void for_seq(F f, std::integer_sequence<int, I...>) { (f(TapC<I>{}), ...); }
template <int N, typename F> __device__ __forceinline__ void for_n(F f) { for_seq(f, std::make_integer_sequence<int, N>{}); }

struct HaloParams {
  ConvGeom g;           // forward: the conv's sources; data gradient: s0 = dY (C = Cout, ld = ld_dy)
  const float* Wt;
  const bf16_t* Wh;     // optional bf16 mirror of Wt, same layout (sg2im_conv_desc.weight_bf16; the WB kernels read it)
  int N;                // output columns: forward Cout, data gradient c_count
  int c_begin;          // data gradient: first input channel of the produced range
  int nchunks;          // 32-channel chunks of the reduction (forward: over all sources; dgrad: ceil(Cout / 32))
  int tiles_x, tiles_y; // patches per image along x / y
  int M;                // NB * H * W
  int dy_bf;            // data gradient: dY holds bfloat16 (sg2im_conv_desc.dy_dtype)
  Epi e;
  StatSink st;
};

// Which patch column a tile-local output pixel q = 16 r + c of a 16-wide patch works on: (c - 2 r) mod 16.
// Why (round 6, found with SQ_LDS_BANK_CONFLICT = one conflict cycle per MFMA in the 8 x 16 kernels, fp32 and bf16):
// a ds_read_b128 of the A fragments serves lanes {0-3, 12-15, 20-27} (and {4-11, 16-19, 28-31}) in one LDS pass, i.e. 8
// pixels of patch row r and 8 of row r + 1.  The bank window of a halo pixel depends on (pixel index mod 16) for both
// image strides (36 floats, 40 bf16), and a halo row is 18 pixels: row r + 1 sits 2 classes further, so with the
// identity mapping columns {0-3, 12-15} of one row meet columns {4-11} + 2 = {6-13} of the next - two 2-way conflicts
// per read, for every tap.  Rotating the columns by -2 per row makes the class of lane i equal to (i mod 16) + const:
// conflict free.  A bijection inside each patch row; the epilogue's row map applies the same rotation.  32- and
// 64-wide patches keep a fragment's 32 pixels in one row: nothing to fix.
template <int CT> __device__ __forceinline__ int patch_col(int r, int c) { return CT == 16 ? ((c - 2 * r) & 15) : c; }

// tile-local output pixel q (row-major inside the patch, columns rotated as above) -> row of the NHWC result
template <int CT> struct PatchRow {
  int nb, H, W, y0, x0;
  __device__ __forceinline__ long long operator()(int q) const {
    return ((long long)nb * H + y0 + q / CT) * W + x0 + patch_col<CT>(q / CT, q % CT);
  }
  // the same pixel's row in a map of HALF the resolution (igemm.h epilogue_bnbwd, pool2)
  __device__ __forceinline__ long long pool2(int q) const {
    return ((long long)nb * (H >> 1) + ((y0 + q / CT) >> 1)) * (W >> 1) + ((x0 + patch_col<CT>(q / CT, q % CT)) >> 1);
  }
};
template <int CT> struct HasPool2<PatchRow<CT>> { static constexpr bool value = true; };

// wave-uniform cursor over the 32-channel chunks of the virtual channel concat
struct ChunkCursor {
  Src S;                // current source
  int s, cb, cstart;    // source index, chunk's first channel inside the source, source's first concat channel
};

// (64-wide column tiles are held to 128 registers: four resident wavefronts per SIMD, as many workgroups per CU
// as their 35-47 KB of LDS allow)
// H: bf16 operand path (igemm.h): both LDS images hold bf16 (halo [pixel][32 + 8], weights [BN][32 + 8] resp.
// k-major [32][BN + 32] read with the transposing ds_read_b64_tr_b16), v_mfma_f32_32x32x16_bf16, fp32 accumulate
// T9 (bf16 only, round 6): the weight slices of ALL NINE taps of a 32-channel chunk are staged at once (9 tap images,
// 46 / 55 KB): one barrier pair and one batch of 24 global loads per thread per CHUNK - 36 MFMAs per wave between two
// barriers instead of 4.  The per-tap form of the bf16 path was bound by the latency of the two weight loads it had
// in flight per tap (4 MFMAs = 128 matrix cycles to cover an L2 round trip) and by 18 barriers per chunk.  Same
// order of accumulation (chunk, tap, K step): bit-identical results.  ~61-70 KB of LDS: two workgroups per CU.
// WB (bf16 only, round 6): the weight slices come from a bf16 MIRROR of the weight tensor (HaloParams::Wh, refreshed
// once per step from the fp32 master weights): one 16-byte load of 8 elements per thread and tap that goes to LDS as it
// is, instead of two fp32 loads + conversions.  The weights are 3/4 of the bytes a 64-column workgroup pulls through
// L2 (73.7 of 97 KB per chunk), and at bf16 matrix rates that stream - ~11 TB/s over the chip - is what bounds these
// kernels.  Same values (RNE either way): bit-identical results.
// TG = taps per staging group: 1 (rounds 3-5: a barrier pair per tap), 3 (one kernel row: 12 MFMAs per wave between
// barriers, three weight pieces in flight per thread, ~30 KB of LDS: four workgroups per CU) or 9 (T9 above).
// AB (bf16 only, round 6): the A side may hold bfloat16 STORAGE - forward: per source (Src::bf, a wave-uniform branch per
// chunk), data gradient: dY (HaloParams::dy_bf).  A separate instantiation so that the fp32-storage loaders keep their
// branch-free basic blocks.
template <int RT, int CT, int BN, bool DG, bool ST, bool H = false, int TG = 1, bool WB = false, bool AB = false>
// (the fp32 4 x 32 data-gradient form - 7 halo float4 per thread, maps that 8 x 16 patches do not tile - needs 130
// registers: three waves per SIMD instead of a spilled offset that is reloaded every chunk)
__global__ __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(TG == 9 ? 2 : (BN == 64 && CT <= 32) ? ((RT == 4 && ((DG && !H) || (!DG && TG == 3))) ? 3 : SG2IM_HALO_WAVES64) : 2)))
void conv_halo_kernel(const HaloParams p) {
  static_assert(TG == 1 || (H && (TG == 3 || TG == 9)), "multi-tap staging exists for the bf16 operand path only");
  static_assert(!AB || H, "bfloat16 storage is read by the bf16 operand path only");
  static_assert(!WB || (H && BN == 64), "the weight mirror is bf16: bf16 operand path, 64-column tiles");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int BM = RT * CT;
  static_assert(BM == 128, "a patch is 128 output pixels");
  constexpr int HWD = CT + 2, HP = (RT + 2) * HWD;            // halo width / halo pixels
  constexpr int NA = (HP * 8 + NTHREADS - 1) / NTHREADS;      // float4 of the halo image per thread
  constexpr int NVB = BN / 32;
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int AF = H ? HP * MLDH / 2 : HP * MLD;            // (floats)
  float* const As = smem;
  float* const Bs = smem + AF;
  bf16_t* const Ash = reinterpret_cast<bf16_t*>(smem);
  bf16_t* const Bsh = reinterpret_cast<bf16_t*>(smem + AF);
  const ConvGeom& g = p.g;
  const int tid = threadIdx.x;
  const int s_wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (wave-uniform: a scalar register)
  const int n0 = blockIdx.x * BN, split = blockIdx.z;
  int tile = blockIdx.y;
  const int tx = tile % p.tiles_x; tile /= p.tiles_x;
  const int ty = tile % p.tiles_y;
  const int nb = tile / p.tiles_y;
  const int y0 = ty * RT, x0 = tx * CT;
  const int per = (p.nchunks + p.e.nsplit - 1) / p.e.nsplit;
  const int c_lo = split * per;
  const int c_hi = min(p.nchunks, c_lo + per);
  const int col4 = tid & 7, r0 = tid >> 3;
  const int ldw = 9 * g.Wtap;
  const int Cout = g.s0.C;                                    // (DG)

  // ---- A loader: thread -> halo float4 (pixel r0 + 32 j, channels 4 col4 .. 4 col4 + 3 of the chunk) ----
  unsigned amask = 0;                                         // bit j: the halo pixel exists and lies inside the image
  #pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int hp = r0 + 32 * j;
    const int hr = hp / HWD, hc = hp - hr * HWD;
    const int ay = y0 - 1 + hr, ax = x0 - 1 + hc;
    const bool ok = hp < HP && (unsigned)ay < (unsigned)g.H && (unsigned)ax < (unsigned)g.W;
    amask |= (ok ? 1u : 0u) << j;
  }
  unsigned aoff[NA];                                          // element offset of the pixel in the current source
  // (recomputed only when the source changes - a handful of times per workgroup - so the pixel coordinates are
  // re-derived from the halo index instead of being kept in registers)
  auto pixel_offsets = [&](const Src& S) {
    const int Hs = g.H >> S.up, Ws = g.W >> S.up;
    #pragma unroll
    for (int j = 0; j < NA; ++j) {
      int hp = r0 + 32 * j;
      asm volatile("" : "+v"(hp));    // (opaque: keeps the compiler from holding the prologue's hr / hc of every j live
                                      // across the main loop for this rarely executed block - spilled registers otherwise)
      const int hr = hp / HWD, hc = hp - hr * HWD;
      const int ay = (amask >> j & 1u) ? y0 - 1 + hr : 0, ax = (amask >> j & 1u) ? x0 - 1 + hc : 0;
      aoff[j] = (unsigned)((nb * Hs + (ay >> S.up)) * Ws + (ax >> S.up)) * (unsigned)S.ld;
    }
  };

  static_assert(offsetof(HaloParams, g) == 0 && offsetof(ConvGeom, s0) == 0, "kernarg_src layout");
  // chunk cursor `nx` = the chunk whose halo is prefetched next; `cu` = the chunk on the matrix cores
  ChunkCursor nx;
  nx.s = 0; nx.cb = 0; nx.cstart = 0; nx.S = kernarg_src(0);
  if (!DG) {
    // decode c_lo
    int ch = c_lo;
    for (;;) {
      const int nc = (nx.S.C + BK - 1) / BK;
      if (ch < nc || nx.s + 1 >= g.nsrc) break;
      ch -= nc; nx.cstart += nx.S.C; nx.s += 1; nx.S = kernarg_src(nx.s);
    }
    nx.cb = ch * BK;
  } else {
    nx.cb = c_lo * BK;
  }
  auto advance = [&](ChunkCursor& c) {
    c.cb += BK;
    if (!DG && c.cb >= c.S.C && c.s + 1 < g.nsrc) {
      c.cstart += c.S.C; c.s += 1; c.cb = 0; c.S = kernarg_src(c.s);
    }
  };

  float4 ra[NA];
  Aff aaff;
  unsigned ramask = 0;
  const BufRsrc rsY = rsrc_of(g.s0.p, (unsigned)p.M * (unsigned)g.s0.ld * (AB ? 2u : 4u));     // (DG: dY; out-of-range offset -> zeros)
  int off_src = -1;
  int ra_bf = 0;                                              // the staged halo registers hold bf16 pairs (wave-uniform)
  auto load_A = [&](const ChunkCursor& c) {
    const int ch = c.cb + 4 * col4;
    const bool cok = ch < c.S.C;
    if (off_src != c.s) { pixel_offsets(c.S); off_src = c.s; }        // (wave-uniform: the source changed)
    if (DG) {
      if constexpr (AB) {
        #pragma unroll
        for (int j = 0; j < NA; ++j)
          ra[j] = ld2h_buf(rsY, ((amask >> j & 1u) && cok) ? (aoff[j] + (unsigned)ch) << 1 : kOobByte);
        ra_bf = 1;
        return;
      }
      #pragma unroll
      for (int j = 0; j < NA; ++j)
        ra[j] = ld4_buf(rsY, ((amask >> j & 1u) && cok) ? (aoff[j] + (unsigned)ch) << 2 : kOobByte);
    } else {
      fetch_aff(aaff, c.S, ch, cok);
      ramask = cok ? amask : 0u;
      if constexpr (AB) {
        if (c.S.bf) {                                               // (wave-uniform: the source's storage type)
          #pragma unroll
          for (int j = 0; j < NA; ++j)
            ra[j] = ld2h_off(c.S.p, (ramask >> j & 1u) ? aoff[j] + (unsigned)ch : 0u);
          ra_bf = 1;
          return;
        }
      }
      ra_bf = 0;
      #pragma unroll
      for (int j = 0; j < NA; ++j)
        ra[j] = ld4_off(c.S.p, (ramask >> j & 1u) ? aoff[j] + (unsigned)ch : 0u);
    }
  };
  auto stage_A = [&]() {
    #pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int hp = r0 + 32 * j;
      float4 v = ra[j];
      if constexpr (AB) { if (DG || ra_bf) v = unpack_bf16x4(v); }
      if (!DG) v = apply_aff(v, aaff, (ramask >> j & 1u) != 0);
      if (NA * 32 <= HP || hp < HP) {
        if constexpr (H) *reinterpret_cast<bf16x4*>(Ash + hp * MLDH + 4 * col4) = to_bf16x4(v);
        else *reinterpret_cast<float4*>(As + hp * MLD + 4 * col4) = v;
      }
    }
  };

  // ---- B loader: the weight tile of one (chunk, tap) ----
  constexpr int NT = TG;                                      // tap images staged at once
  constexpr int TAPH = DG ? BK * (BN + KPADH) : BN * MLDH;    // bf16 elements of one tap image
  float4 rb[WB ? 1 : NT][NVB];
  f32x4v rbw[WB ? NT : 1];                                    // (WB: one 16-byte piece per tap; a VECTOR type - a float4
                                                              // struct copied memory-to-memory keeps the array in scratch)
  // WB: thread -> one 16-byte piece (8 bf16) of the tap image.  forward: row = tid >> 2 (output channel), piece =
  // tid & 3 (8 channels of the chunk); data gradient: k row = tid >> 3 (output channel of the chunk), piece = tid & 7
  // (8 input channels)
  const int hrow = DG ? tid >> 3 : tid >> 2, hpc = DG ? tid & 7 : tid & 3;
  // forward: m-major rows = output channels n0 + r0 + 32 i, k = channels of the chunk
  unsigned wrow[NVB];
  #pragma unroll
  for (int i = 0; i < NVB; ++i) {
    const int n = n0 + r0 + 32 * i;
    wrow[i] = n < p.N ? (unsigned)n * (unsigned)ldw : 0u;
  }
  // data gradient: k-major rows = output channels of the chunk, columns = input channels n0 + 4 bcol4 ..
  constexpr int QB = BN / 4;
  const int bcol4 = tid % QB, bk0 = tid / QB;
  const int nn = n0 + 4 * bcol4;
  const bool nok4 = nn < p.N;
  // slot: the register set / tap image the slice goes to (T9: = tap; per-tap form: 0)
  auto load_B = [&](const ChunkCursor& c, int tap, auto slotc) __attribute__((always_inline)) {
    constexpr int slot = decltype(slotc)::v;
    if constexpr (WB) {
      unsigned off;       // bf16 element offset into the mirror
      bool ok;
      if (!DG) {
        const int ch = c.cb + 8 * hpc, n = n0 + hrow;
        ok = ch < c.S.C && n < p.N;
        off = (unsigned)n * (unsigned)ldw + (unsigned)(tap * g.Wtap + c.cstart + ch);
      } else {
        const int co = c.cb + hrow, n8 = n0 + 8 * hpc;
        ok = co < Cout && n8 < p.N;
        off = (unsigned)co * (unsigned)ldw + (unsigned)(tap * g.Wtap + p.c_begin + n8);
      }
      // (16 bytes at a 2-byte-element offset: ld4_off takes float-element offsets - half of it, the base cast)
      unsigned byte_off = ok ? off << 1 : 0u;
      asm volatile("" : "+v"(byte_off));
      rbw[slot] = *reinterpret_cast<const f32x4v*>(reinterpret_cast<const char*>(p.Wh) + (size_t)byte_off);
      // (no select on a masked-off piece: it read element 0 - finite - and meets zeros of the A operand, or lands in a
      // column that is never stored, like the fp32 loader's)
      return;
    } else
    if (!DG) {
      const int ch = c.cb + 4 * col4;
      const bool cok = ch < c.S.C;
      const unsigned wcol = (unsigned)(tap * g.Wtap + c.cstart + ch);
      #pragma unroll
      for (int i = 0; i < NVB; ++i) {
        const bool ok = cok && (n0 + r0 + 32 * i) < p.N;
        rb[slot][i] = ld4_off(p.Wt, ok ? wrow[i] + wcol : 0u);      // (no select: see conv_fwd_kernel)
      }
    } else {
      const unsigned wcol = (unsigned)(tap * g.Wtap + p.c_begin + nn);
      #pragma unroll
      for (int i = 0; i < NVB; ++i) {
        const int co = c.cb + bk0 + (1024 / BN) * i;
        const bool ok = nok4 && co < Cout;
        rb[slot][i] = ld4_off(p.Wt, ok ? (unsigned)co * (unsigned)ldw + wcol : 0u);
      }
    }
  };
  auto stage_B = [&](auto slotc) __attribute__((always_inline)) {
    constexpr int slot = decltype(slotc)::v;
    if constexpr (WB) {
      bf16_t* const dst = Bsh + slot * TAPH + (DG ? hrow * (BN + KPADH) : hrow * MLDH) + 8 * hpc;
      *reinterpret_cast<f32x4v*>(dst) = rbw[slot];
    } else if constexpr (H) {
      if (!DG) store_tile_h<BN, false>(Bsh + slot * TAPH, rb[slot], tid);
      else store_tile_h<BN, true>(Bsh + slot * TAPH, rb[slot], tid);
    } else {
      if (!DG) store_tile<BN, false>(Bs, rb[slot], tid);
      else store_tile<BN, true>(Bs, rb[slot], tid);
    }
  };

  // ---- fragments ----
  int wm0, wn0, lane;
  wave_origin<BM, BN>(tid, wm0, wn0, lane);
  const int li = lane & 31, lh = lane >> 5;
  int apix[TM];                                               // halo pixel of this lane's output pixel, per row fragment
  #pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int q = wm0 + tm * 32 + li;
    apix[tm] = (q / CT) * HWD + patch_col<CT>(q / CT, q % CT);
  }
  Frags<BM, BN> f;
  FragsH<BM, BN> fh;
  f32x16 acc[TM][TN];
  #pragma unroll
  for (int a_ = 0; a_ < TM; ++a_)
    #pragma unroll
    for (int b_ = 0; b_ < TN; ++b_) zero_acc(acc[a_][b_]);

  auto mma_tap = [&](int tapoff, auto slotc) __attribute__((always_inline)) {
    constexpr int slot = decltype(slotc)::v;
    if constexpr (H) {
      const bf16_t* const Bt = Bsh + slot * TAPH;
      #pragma unroll
      for (int stp = 0; stp < 2; ++stp) {
        #pragma unroll
        for (int tm = 0; tm < TM; ++tm)
          fh.a[tm][stp] = *reinterpret_cast<const bf16x8*>(Ash + (apix[tm] + tapoff) * MLDH + 16 * stp + 8 * lh);
        #pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          if (!DG) fh.b[tn][stp] = *reinterpret_cast<const bf16x8*>(Bt + (wn0 + tn * 32 + li) * MLDH + 16 * stp + 8 * lh);
          else fh.b[tn][stp] = read_tr<BN>(Bt, wn0 + tn * 32, stp, lane);
        }
      }
      mma_frags_h<BM, BN>(fh, acc);
      return;
    }
    #pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const float* row = As + (apix[tm] + tapoff) * MLD + 4 * lh;
      #pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(row + 8 * q);
        f.a[tm][4 * q + 0] = v.x; f.a[tm][4 * q + 1] = v.y; f.a[tm][4 * q + 2] = v.z; f.a[tm][4 * q + 3] = v.w;
      }
    }
    #pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      if (!DG) {
        const float* row = Bs + (wn0 + tn * 32 + li) * MLD + 4 * lh;
        #pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 v = *reinterpret_cast<const float4*>(row + 8 * q);
          f.b[tn][4 * q + 0] = v.x; f.b[tn][4 * q + 1] = v.y; f.b[tn][4 * q + 2] = v.z; f.b[tn][4 * q + 3] = v.w;
        }
      } else {
        #pragma unroll
        for (int sI = 0; sI < 16; ++sI) f.b[tn][sI] = Bs[kperm(sI, lh) * (BN + KPAD) + wn0 + tn * 32 + li];
      }
    }
    mma_frags<BM, BN>(f, acc);
  };

  // ---- main loop: chunks outer, the nine taps unrolled ----
  if constexpr (TG > 1) {
    constexpr int G = 9 / TG;                                   // staging groups per chunk
    if (c_lo < c_hi) {
      ChunkCursor cu = nx;
      load_A(cu);
      for_n<TG>([&](auto t) __attribute__((always_inline)) { load_B(cu, decltype(t)::v, t); });
      stage_A();
      for_n<TG>([&](auto t) __attribute__((always_inline)) { stage_B(t); });
      __syncthreads();
      #pragma unroll 1
      for (int ch = c_lo; ch < c_hi; ++ch) {
        const bool more = ch + 1 < c_hi;
        nx = cu;
        if (more) advance(nx);
        load_A(nx);
        for_n<G>([&](auto gc) __attribute__((always_inline)) {
          constexpr int gi = decltype(gc)::v;
          // the weights of the next group (the first of the next chunk behind the last) are in flight during the MFMAs
          for_n<TG>([&](auto t) __attribute__((always_inline)) {
            if constexpr (gi + 1 < G) load_B(cu, (gi + 1) * TG + decltype(t)::v, t); else load_B(nx, decltype(t)::v, t);
          });
          __builtin_amdgcn_sched_barrier(0);
          for_n<TG>([&](auto t) __attribute__((always_inline)) {
            constexpr int tap = gi * TG + decltype(t)::v, kh = tap / 3, kw = tap - 3 * kh;
            mma_tap(DG ? (2 - kh) * HWD + (2 - kw) : kh * HWD + kw, t);
          });
          __syncthreads();
          if constexpr (gi == G - 1) stage_A();
          for_n<TG>([&](auto t) __attribute__((always_inline)) { stage_B(t); });
          __syncthreads();
        });
        cu = nx;
      }
    }
  } else if (c_lo < c_hi) {
    ChunkCursor cu = nx;
    load_A(cu);
    load_B(cu, 0, TapC<0>{});
    stage_A();
    stage_B(TapC<0>{});
    __syncthreads();
    #pragma unroll 1
    for (int ch = c_lo; ch < c_hi; ++ch) {
      // halo of the next chunk (the last chunk is fetched again instead of branching around the tail)
      const bool more = ch + 1 < c_hi;
      nx = cu;
      if (more) advance(nx);
      load_A(nx);
      #pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int kh = tap / 3, kw = tap - 3 * kh;
        if (tap < 8) load_B(cu, tap + 1, TapC<0>{}); else load_B(nx, 0, TapC<0>{});
        // keep the global loads ahead of the MFMA block (see k_pipeline)
        __builtin_amdgcn_sched_barrier(0);
        mma_tap(DG ? (2 - kh) * HWD + (2 - kw) : kh * HWD + kw, TapC<0>{});
        __syncthreads();
        stage_B(TapC<0>{});
        if (tap == 8) stage_A();
        __syncthreads();
      }
      cu = nx;
    }
  }

  // The epilogue's thread coordinates are RE-DERIVED here (lane from mbcnt, the wave's tile origin from a scalar)
  // instead of being kept alive across the main loop: at the 128-register budget of the 64-column forms that cost
  // two or three spilled registers (scratch stores in the prologue, reloads here).
  const int e_lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const int e_wm0 = (s_wave >> 1) * (BM / 2), e_wn0 = (s_wave & 1) * (BN / 2);
  const int e_tid = s_wave * 64 + e_lane;
  // (the patch origin too: workgroup-uniform, but the divisions leave it in vector registers, and held across the loop it
  // was spilled at the 128-register budget - re-derived from an opaque copy of the block index)
  int e_tile = blockIdx.y, e_nx = p.tiles_x, e_ny = p.tiles_y;
  asm volatile("" : "+s"(e_tile), "+s"(e_nx), "+s"(e_ny));       // (the divisions' reciprocals are not carried across the loop either)
  const int e_tx = e_tile % e_nx; e_tile /= e_nx;
  const int e_ty = e_tile % e_ny;
  const PatchRow<CT> rowmap{e_tile / e_ny, g.H, g.W, e_ty * RT, e_tx * CT};
  if (DG && p.e.mask != nullptr && p.e.nsplit == 1)       // (workgroup-uniform; the activation mask of sg2im_conv2d_backward_data_act)
    epilogue<BM, BN, PatchRow<CT>, true, true, true>(p.e, p.M, p.N, p.N, 0, n0, e_wm0, e_wn0, e_lane, split, acc, rowmap);
  else
    epilogue<BM, BN, PatchRow<CT>, true, false, true>(p.e, p.M, p.N, p.N, 0, n0, e_wm0, e_wn0, e_lane, split, acc, rowmap);
  if constexpr (ST) {
    if (p.e.nsplit == 1) {
      if (!DG) epilogue_stats<BM, BN, true>(p.e, p.st, BM, p.N, 0, n0, e_wm0, e_wn0, e_lane, e_tid, blockIdx.y, acc, smem);
      else epilogue_bnbwd<BM, BN, PatchRow<CT>, true>(p.st, BM, p.N, 0, n0, e_wm0, e_wn0, e_lane, e_tid, blockIdx.y, acc, smem, rowmap);
    }
  }
}

template <int RT, int CT, int BN, bool DG, bool H = false, int TG = 1> constexpr size_t halo_lds() {     // (WB: as its H form)
  return H ? ((size_t)(RT + 2) * (CT + 2) * MLDH + TG * (DG ? (size_t)BK * (BN + KPADH) : (size_t)BN * MLDH)) * 2
           : ((size_t)(RT + 2) * (CT + 2) * MLD + (DG ? (size_t)BK * (BN + KPAD) : (size_t)BN * MLD)) * sizeof(float);
}

}  // namespace sg2im
