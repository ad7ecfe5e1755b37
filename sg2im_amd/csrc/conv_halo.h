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

#ifndef SG2IM_HALO_WAVES64
#define SG2IM_HALO_WAVES64 4
#endif

namespace sg2im {

struct HaloParams {
  ConvGeom g;           // forward: the conv's sources; data gradient: s0 = dY (C = Cout, ld = ld_dy)
  const float* Wt;
  int N;                // output columns: forward Cout, data gradient c_count
  int c_begin;          // data gradient: first input channel of the produced range
  int nchunks;          // 32-channel chunks of the reduction (forward: over all sources; dgrad: ceil(Cout / 32))
  int tiles_x, tiles_y; // patches per image along x / y
  int M;                // NB * H * W
  Epi e;
  StatSink st;
};

// tile-local output pixel q (row-major inside the patch) -> row of the NHWC result
template <int CT> struct PatchRow {
  int nb, H, W, y0, x0;
  __device__ __forceinline__ long long operator()(int q) const {
    return ((long long)nb * H + y0 + q / CT) * W + x0 + q % CT;
  }
};

// wave-uniform cursor over the 32-channel chunks of the virtual channel concat
struct ChunkCursor {
  Src S;                // current source
  int s, cb, cstart;    // source index, chunk's first channel inside the source, source's first concat channel
};

// (64-wide column tiles are held to 128 registers: four resident wavefronts per SIMD, as many workgroups per CU
// as their 35-47 KB of LDS allow)
// H: bf16 operand path (igemm.h): both LDS images hold bf16 (halo [pixel][32 + 8], weights [BN][32 + 8] resp.
// k-major [32][BN + 32] read with the transposing ds_read_b64_tr_b16), v_mfma_f32_32x32x16_bf16, fp32 accumulate
template <int RT, int CT, int BN, bool DG, bool ST, bool H = false>
// (the fp32 4 x 32 data-gradient form - 7 halo float4 per thread, maps that 8 x 16 patches do not tile - needs 130
// registers: three waves per SIMD instead of a spilled offset that is reloaded every chunk)
__global__ __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu((BN == 64 && CT <= 32) ? ((RT == 4 && DG && !H) ? 3 : SG2IM_HALO_WAVES64) : 2)))
void conv_halo_kernel(const HaloParams p) {
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
  const BufRsrc rsY = rsrc_of(g.s0.p, (unsigned)p.M * (unsigned)g.s0.ld * 4u);     // (DG: dY; out-of-range offset -> zeros)
  int off_src = -1;
  auto load_A = [&](const ChunkCursor& c) {
    const int ch = c.cb + 4 * col4;
    const bool cok = ch < c.S.C;
    if (off_src != c.s) { pixel_offsets(c.S); off_src = c.s; }        // (wave-uniform: the source changed)
    if (DG) {
      #pragma unroll
      for (int j = 0; j < NA; ++j)
        ra[j] = ld4_buf(rsY, ((amask >> j & 1u) && cok) ? (aoff[j] + (unsigned)ch) << 2 : kOobByte);
    } else {
      fetch_aff(aaff, c.S, ch, cok);
      ramask = cok ? amask : 0u;
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
      if (!DG) v = apply_aff(v, aaff, (ramask >> j & 1u) != 0);
      if (NA * 32 <= HP || hp < HP) {
        if constexpr (H) *reinterpret_cast<bf16x4*>(Ash + hp * MLDH + 4 * col4) = to_bf16x4(v);
        else *reinterpret_cast<float4*>(As + hp * MLD + 4 * col4) = v;
      }
    }
  };

  // ---- B loader: the weight tile of one (chunk, tap) ----
  float4 rb[NVB];
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
  auto load_B = [&](const ChunkCursor& c, int tap) {
    if (!DG) {
      const int ch = c.cb + 4 * col4;
      const bool cok = ch < c.S.C;
      const unsigned wcol = (unsigned)(tap * g.Wtap + c.cstart + ch);
      #pragma unroll
      for (int i = 0; i < NVB; ++i) {
        const bool ok = cok && (n0 + r0 + 32 * i) < p.N;
        rb[i] = ld4_off(p.Wt, ok ? wrow[i] + wcol : 0u);      // (no select: see conv_fwd_kernel)
      }
    } else {
      const unsigned wcol = (unsigned)(tap * g.Wtap + p.c_begin + nn);
      #pragma unroll
      for (int i = 0; i < NVB; ++i) {
        const int co = c.cb + bk0 + (1024 / BN) * i;
        const bool ok = nok4 && co < Cout;
        rb[i] = ld4_off(p.Wt, ok ? (unsigned)co * (unsigned)ldw + wcol : 0u);
      }
    }
  };
  auto stage_B = [&]() {
    if constexpr (H) {
      if (!DG) store_tile_h<BN, false>(Bsh, rb, tid);
      else store_tile_h<BN, true>(Bsh, rb, tid);
    } else {
      if (!DG) store_tile<BN, false>(Bs, rb, tid);
      else store_tile<BN, true>(Bs, rb, tid);
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
    apix[tm] = (q / CT) * HWD + q % CT;
  }
  Frags<BM, BN> f;
  FragsH<BM, BN> fh;
  f32x16 acc[TM][TN];
  #pragma unroll
  for (int a_ = 0; a_ < TM; ++a_)
    #pragma unroll
    for (int b_ = 0; b_ < TN; ++b_) zero_acc(acc[a_][b_]);

  auto mma_tap = [&](int tapoff) {
    if constexpr (H) {
      #pragma unroll
      for (int stp = 0; stp < 2; ++stp) {
        #pragma unroll
        for (int tm = 0; tm < TM; ++tm)
          fh.a[tm][stp] = *reinterpret_cast<const bf16x8*>(Ash + (apix[tm] + tapoff) * MLDH + 16 * stp + 8 * lh);
        #pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          if (!DG) fh.b[tn][stp] = *reinterpret_cast<const bf16x8*>(Bsh + (wn0 + tn * 32 + li) * MLDH + 16 * stp + 8 * lh);
          else fh.b[tn][stp] = read_tr<BN>(Bsh, wn0 + tn * 32, stp, lane);
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
  if (c_lo < c_hi) {
    ChunkCursor cu = nx;
    load_A(cu);
    load_B(cu, 0);
    stage_A();
    stage_B();
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
        if (tap < 8) load_B(cu, tap + 1); else load_B(nx, 0);
        // keep the global loads ahead of the MFMA block (see k_pipeline)
        __builtin_amdgcn_sched_barrier(0);
        mma_tap(DG ? (2 - kh) * HWD + (2 - kw) : kh * HWD + kw);
        __syncthreads();
        stage_B();
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
  const PatchRow<CT> rowmap{nb, g.H, g.W, y0, x0};
  if (DG && p.e.mask != nullptr && p.e.nsplit == 1)       // (workgroup-uniform; the activation mask of sg2im_conv2d_backward_data_act)
    epilogue<BM, BN, PatchRow<CT>, true, true>(p.e, p.M, p.N, p.N, 0, n0, e_wm0, e_wn0, e_lane, split, acc, rowmap);
  else
    epilogue<BM, BN, PatchRow<CT>, true>(p.e, p.M, p.N, p.N, 0, n0, e_wm0, e_wn0, e_lane, split, acc, rowmap);
  if constexpr (ST) {
    if (p.e.nsplit == 1) {
      if (!DG) epilogue_stats<BM, BN>(p.e, p.st, BM, p.N, 0, n0, e_wm0, e_wn0, e_lane, e_tid, blockIdx.y, acc, smem);
      else epilogue_bnbwd<BM, BN, PatchRow<CT>>(p.st, BM, p.N, 0, n0, e_wm0, e_wn0, e_lane, e_tid, blockIdx.y, acc, smem, rowmap);
    }
  }
}

template <int RT, int CT, int BN, bool DG, bool H = false> constexpr size_t halo_lds() {
  return H ? ((size_t)(RT + 2) * (CT + 2) * MLDH + (DG ? (size_t)BK * (BN + KPADH) : (size_t)BN * MLDH)) * 2
           : ((size_t)(RT + 2) * (CT + 2) * MLD + (DG ? (size_t)BK * (BN + KPAD) : (size_t)BN * MLD)) * sizeof(float);
}

}  // namespace sg2im
