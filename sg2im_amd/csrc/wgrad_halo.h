// Weight gradient of the 3x3 / stride 1 / pad 1 convolutions with a HALO'D INPUT TILE in LDS (the transpose of
// conv_halo.h; reference under autograd: nn.Conv2d.weight.grad in crn.py:41-47,79-86).
//
//   dW[k][kh][kw][c] = sum_{n,y,x} dY[n][y][x][k] * A[n][y + kh - 1][x + kw - 1][c]        (A: the activated input)
//
// The first-generation kernel (conv_wgrad_kernel) tiles the FLAT (tap, channel) axis: a 64 x 64 or 64 x 128 tile
// re-loads and re-activates the (shifted) input pixels once per tap and gets 16-32 MFMAs out of every staged
// 32-pixel chunk - too few to cover the latency of the next chunk's loads on the 64-channel layers (76 TFLOP/s),
// and m4.conv0 fetched 1.1 GB per launch for 0.18 GB of operands (profiles/r2_pmc_hbm_traffic_m4conv0.txt; this kernel:
// 0.30 GB, profiles/r3_pmc_hbm_traffic_m4conv0.txt).
// Here a workgroup owns 64 output channels x 64 input channels x ALL NINE TAPS and walks a range of RT x CT = 64
// pixel patches: per patch it stages the 64 x 64 dY tile and the (RT + 2) x (CT + 2) halo of the input ONCE
// (loaded once, activated once) and issues 9 x 32 MFMAs per wave from them - a tap is an immediate offset of the
// B-fragment reads.  Waves: 2 (halves of the output channels) x 2 (halves of the input channels); 144 accumulator
// registers per lane.  Both LDS images are [pixel][64] with the upper / lower 32 columns swapped on odd pixels:
// the two pixels of an MFMA's K pair then sit on disjoint banks without padding.
#pragma once
#include "igemm.h"

namespace sg2im {

struct WgHaloParams {
  ConvGeom g;             // the forward convolution's sources (offset 0: kernarg_src)
  const float* dY;        // [NB * H * W][ldy]
  int ldy, Cout;
  int tiles_x, tiles_y;   // patches per image
  int npatch, per;        // patches in all, patches per K split
  Epi e;                  // C = dW (ldc = 9 * Wtap floats per row), ws = split-K partials [nsplit][Cout][9 * Ctot]
  float* dbias;           // optional: the bias gradient (column sums of dY), produced by the workgroups of channel block 0
  float* ws_bias;         // its split-K partials [nsplit][Cout]
  int dy_bf;              // dY holds bfloat16 (sg2im_conv_desc.dy_dtype; conv_wgrad_halo_h_kernel only)
};

template <int RT, int CT>
__global__ __launch_bounds__(NTHREADS) void conv_wgrad_halo_kernel(const WgHaloParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int PX = RT * CT;                       // pixels of a patch = K of one staging step
  static_assert(PX == 64 && CT % 2 == 0, "64-pixel patches, even width");
  constexpr int HWD = CT + 2, HP = (RT + 2) * HWD;  // halo width / halo pixels
  constexpr int NX = (HP + 15) / 16;                // halo float4 per thread (16 threads span the 64 channels)
  constexpr int NY = PX / 16;                       // dY float4 per thread
  constexpr int LBUF = (PX + HP) * 64;              // floats of one LDS image: Ys [PX][64] + Xs [HP][64]
  const ConvGeom& g = p.g;
  const int tid = threadIdx.x;
  const int cb0 = blockIdx.x * 64, k0 = blockIdx.y * 64, split = blockIdx.z;
  const int p_lo = split * p.per, p_hi = min(p.npatch, p_lo + p.per);
  const int col4 = tid & 15, r0 = tid >> 4;

  // ---- input operand: this thread's four channels belong to one source for the whole launch ----
  const int xc = cb0 + 4 * col4;
  const bool xok = xc < g.Ctot;
  int xs_ = 0, xcs = 0;
  locate_channel(g, xok ? xc : 0, xs_, xcs);
  const Src XS = pick_src(g, xs_);
  Aff xaff;
  fetch_aff(xaff, XS, xcs, xok);
  const int Hs = g.H >> XS.up, Ws = g.W >> XS.up;
  // ---- dY operand ----
  const int yk = k0 + 4 * col4;
  const bool yok = yk < p.Cout;
  const BufRsrc rsY = rsrc_of(p.dY, (unsigned)(g.NB * g.H * g.W) * (unsigned)p.ldy * 4u);

  float4 rx[NX], ry[NY];
  unsigned rxm = 0;
  int l_nb = 0, l_y0 = 0, l_x0 = 0;                 // patch being loaded
  auto begin_patch = [&](int pt) {
    int t = pt;
    const int tx = t % p.tiles_x; t /= p.tiles_x;
    const int ty = t % p.tiles_y;
    l_nb = t / p.tiles_y; l_y0 = ty * RT; l_x0 = tx * CT;
    rxm = 0;
  };
  auto load_x = [&](int j) {                        // halo float4 j of the patch
    const int hp = r0 + 16 * j;
    const int hr = hp / HWD, hc = hp - hr * HWD;
    const int ay = l_y0 - 1 + hr, ax = l_x0 - 1 + hc;
    const bool ok = xok && hp < HP && (unsigned)ay < (unsigned)g.H && (unsigned)ax < (unsigned)g.W;
    rxm |= (ok ? 1u : 0u) << j;
    // (branch-free: a pixel outside the image loads its clamped neighbour and is zeroed through the mask when it is
    // staged - a branch here would split the stage into basic blocks the MFMAs cannot be interleaved with)
    const int ayc = min(max(ay, 0), g.H - 1), axc = min(max(ax, 0), g.W - 1);
    const unsigned off = (unsigned)((l_nb * Hs + (ayc >> XS.up)) * Ws + (axc >> XS.up)) * (unsigned)XS.ld + (unsigned)xcs;
    rx[j] = ld4_off(XS.p, off);
  };
  auto load_y = [&](int j) {                        // dY float4 j
    const int q = r0 + 16 * j;
    const int pix = (l_nb * g.H + l_y0 + q / CT) * g.W + l_x0 + q % CT;
    ry[j] = ld4_buf(rsY, yok ? ((unsigned)pix * (unsigned)p.ldy + (unsigned)yk) << 2 : kOobByte);       // (zeros when !yok)
  };
  const bool want_db = p.dbias != nullptr && blockIdx.x == 0;
  float4 dbs = zero4();
  auto stage_x = [&](float* Xs, int j) {
    const int hp = r0 + 16 * j;
    if (NX * 16 <= HP || hp < HP) {
      const float4 v = apply_aff(rx[j], xaff, (rxm >> j & 1u) != 0);
      *reinterpret_cast<float4*>(Xs + hp * 64 + ((4 * col4) ^ ((hp & 1) << 5))) = v;
    }
  };
  auto stage_y = [&](float* Ys, int j) {
    const int q = r0 + 16 * j;
    if (want_db) { dbs.x += ry[j].x; dbs.y += ry[j].y; dbs.z += ry[j].z; dbs.w += ry[j].w; }
    *reinterpret_cast<float4*>(Ys + q * 64 + ((4 * col4) ^ ((q & 1) << 5))) = ry[j];
  };

  // ---- fragments: lane (i, h) of wave (kwv, cwv) ----
  const int wave = tid >> 6, lane = tid & 63;
  const int kwv = wave & 1, cwv = wave >> 1;
  const int li = lane & 31, lh = lane >> 5;
  f32x16 acc[9];
  #pragma unroll
  for (int t = 0; t < 9; ++t) zero_acc(acc[t]);
  // K pair s of the patch = pixels 2 s + lh: row s / (CT / 2), column 2 (s % (CT / 2)) + lh.
  // A: Ys[q][kcol ^ 32 lh];  B(tap): Xs[halo pixel][ccol ^ 32 ((lh + kw) & 1)]   (HWD is even)
  const int ya_off = lh * 64 + ((kwv * 32 + li) ^ (lh << 5));
  const int xb0_off = PX * 64 + lh * 64 + ((cwv * 32 + li) ^ (lh << 5));             // halo columns j even
  const int xb1_off = PX * 64 + lh * 64 + ((cwv * 32 + li) ^ ((lh ^ 1) << 5));       // j odd
  // One STAGE = (patch row r, kernel row kh): the CT / 2 K pairs of the row against the three taps of kernel row kh
  // = 3 CT / 2 MFMAs from CT / 2 dY values and the CT + 1 values of halo row r + kh this lane half touches
  // (column j + lh, j = c + kw).  The fragments of the NEXT stage are read from LDS before the MFMAs of the
  // current one are issued (register double buffer, stages fully unrolled): with one wavefront per SIMD nothing
  // else would cover the LDS latency.  The global loads of the next patch and their staging into the OTHER LDS
  // image are spread over the stages (a few float4 per stage), next to the MFMAs instead of in front of them.
  constexpr int NST = 3 * RT;                          // stages per patch
  static_assert(NX + NY + 1 <= NST, "the loads and the stores of the loader pieces take disjoint stages");
  auto mma_patch = [&](const float* cur, float* nxt, bool more) {
    constexpr int SR = CT / 2;                         // K pairs per patch row
    const float* const ya = cur + ya_off;
    const float* const xb0 = cur + xb0_off;
    const float* const xb1 = cur + xb1_off;
    float a[2][SR], b[2][CT + 1];
    auto read_a = [&](float (&dst)[SR], int r) {
      #pragma unroll
      for (int u = 0; u < SR; ++u) dst[u] = ya[(r * CT + 2 * u) * 64];
    };
    auto read_b = [&](float (&dst)[CT + 1], int hrow) {
      #pragma unroll
      for (int j = 0; j <= CT; ++j) dst[j] = ((j & 1) ? xb1 : xb0)[(hrow * HWD + j) * 64];
    };
    read_a(a[0], 0);
    read_b(b[0], 0);
    #pragma unroll
    for (int st = 0; st < NST; ++st) {
      const int r = st / 3, kh = st - 3 * r;
      if (st + 1 < NST) {
        const int r2 = (st + 1) / 3, kh2 = (st + 1) - 3 * r2;
        if (kh2 == 0) read_a(a[r2 & 1], r2);
        read_b(b[(st + 1) & 1], r2 + kh2);
      }
      // loader pieces: loads in the first half of the patch, LDS stores of the arrived data in the second
      constexpr int NLS = (NX + NY + 1) / 2;           // stages that issue loads / stores (two pieces each)
      constexpr int FIRST_STORE = NST - NLS;
      if (st < NLS) {
        #pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int piece = 2 * st + k;
          if (piece < NX) load_x(piece);
          else if (piece < NX + NY) load_y(piece - NX);
        }
      }
      if (more && st >= FIRST_STORE) {
        #pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int piece = 2 * (st - FIRST_STORE) + k;
          if (piece < NX) stage_x(nxt + PX * 64, piece);
          else if (piece < NX + NY) stage_y(nxt, piece - NX);
        }
      }
      #pragma unroll
      for (int u = 0; u < SR; ++u)
        #pragma unroll
        for (int kw = 0; kw < 3; ++kw)
          acc[3 * kh + kw] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r & 1][u], b[st & 1][2 * u + kw], acc[3 * kh + kw], 0, 0, 0);
      // issue order inside the stage: the LDS reads of the next fragments first, then every MFMA followed by a few
      // of the loader's VALU / SALU / memory instructions - an in-order wave overlaps them only when they are
      // interleaved (one wavefront per SIMD: nobody else fills the matrix pipe while this wave does address math)
      __builtin_amdgcn_sched_group_barrier(0x100, CT + 1 + SR, 0);
      #pragma unroll
      for (int k = 0; k < 3 * SR; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x006, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x230, 1, 0);       // (a global load or an LDS store)
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  if (p_lo < p_hi) {
    begin_patch(p_lo);
    #pragma unroll
    for (int j = 0; j < NX; ++j) load_x(j);
    #pragma unroll
    for (int j = 0; j < NY; ++j) load_y(j);
    #pragma unroll
    for (int j = 0; j < NX; ++j) stage_x(smem + PX * 64, j);
    #pragma unroll
    for (int j = 0; j < NY; ++j) stage_y(smem, j);
    __syncthreads();
    #pragma unroll 1
    for (int pt = p_lo; pt < p_hi; ++pt) {
      const bool more = pt + 1 < p_hi;
      begin_patch(more ? pt + 1 : pt);                 // (the last patch is fetched again instead of branching around the tail)
      const int w = (pt - p_lo) & 1;
      mma_patch(smem + w * LBUF, smem + (w ^ 1) * LBUF, more);
      __syncthreads();                                 // the other image is complete; this one is free for the patch after next
    }
  }

  // ---- epilogue: per tap a 32 x 32 fragment of dW; C/D layout col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) ----
  const int Ntot = 9 * g.Ctot;
  const int c = cb0 + cwv * 32 + li;
  if (c < g.Ctot) {
    #pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int n = tap * g.Ctot + c;
      #pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = k0 + kwv * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m >= p.Cout) continue;
        float v = acc[tap][r];
        if (p.e.nsplit > 1) {
          p.e.ws[((long long)split * p.Cout + m) * Ntot + n] = v;
        } else {
          float* dst = p.e.C + (long long)m * p.e.ldc + epi_col(p.e, n);
          if (p.e.accumulate) v += *dst;
          *dst = v;
        }
      }
    }
  }
  // bias gradient (workgroup-uniform condition): the 16 threads that share a channel quad hold sums over disjoint
  // pixels - combined through LDS in thread order (fixed order: reproducible)
  if (p.dbias != nullptr && blockIdx.x == 0) {
    float4* red = reinterpret_cast<float4*>(smem);     // [16][16]  (the operand images are free now)
    __syncthreads();
    red[r0 * 16 + col4] = dbs;
    __syncthreads();
    if (tid < 16 && k0 + 4 * tid < p.Cout) {
      float4 t = red[tid];
      for (int k = 1; k < 16; ++k) {
        const float4 u = red[k * 16 + tid];
        t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
      }
      const float tv[4] = {t.x, t.y, t.z, t.w};
      float* dst = p.e.nsplit > 1 ? p.ws_bias + (size_t)split * p.Cout : p.dbias;
      for (int j = 0; j < 4; ++j) {
        const int k = k0 + 4 * tid + j;
        if (k >= p.Cout) break;
        if (p.e.nsplit > 1 || !p.e.accumulate) dst[k] = tv[j];
        else dst[k] += tv[j];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// bf16 operand form (sg2im_conv_desc.compute_dtype = 1; round 5).  Until round 4 the bf16 mode's weight gradients ran
// on the per-tap kernel, which re-reads and re-activates the fp32 input once per tap.  Same ownership as above - a
// workgroup owns 64 output channels x 64 input channels x all nine taps and walks 4 x 16 pixel patches - but both LDS
// images hold bfloat16, PIXEL-major ([pixel][64 + 32] elements: the k-major layout of igemm.h, a pixel row = 192 bytes),
// and a patch ROW of 16 pixels is exactly one K = 16 step of v_mfma_f32_32x32x16_bf16:
//   A (dY)    = the transposing read ds_read_b64_tr_b16 of pixels 16 r .. 16 r + 15, columns = this wave's 32 output channels
//   B (input) = the same read of halo pixels (r + kh) * 18 + kw .. + 15, columns = this wave's 32 input channels
// - a tap is a different FIRST ROW of the B read.  Per patch and wave: 4 A fragments, 18 B fragments (one per halo row
// and kernel column, shared by the up to three (patch row, kernel row) pairs that meet in that halo row) and 36 MFMAs
// against the fp32 kernel's 288.  Tensors in HBM stay fp32: operands are rounded (RNE) when they are staged, the pending
// BatchNorm affine + LeakyReLU of the input is applied in fp32 before the rounding, the bias gradient is summed from the
// fp32 dY.  Two LDS images (2 x 33 KB): the next patch's global loads are in flight while this one is on the matrix
// cores; ~2 workgroups per CU cover each other's staging (no hand-interleaved schedule as in the fp32 kernel - with 8x
// fewer MFMA instructions per byte this kernel is bound by its loads, not by issue slots).
// ---------------------------------------------------------------------------------------------------------------
constexpr int WGH_LD = 64 + KPADH;                    // bf16 elements between two pixel rows of an image

__device__ __forceinline__ bf16x8 read_tr_at(const bf16_t* img, int row0, int col0, int lane) {
  typedef __attribute__((address_space(3))) bf16x4* LdsPtr;
  const int p = lane & 15, g = lane >> 4;
  const bf16_t* a = img + (row0 + 8 * (g >> 1) + (p >> 2)) * WGH_LD + col0 + 16 * (g & 1) + 4 * (p & 3);
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LdsPtr)a);
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LdsPtr)(a + 4 * WGH_LD));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

#ifndef SG2IM_WGH_WAVES
#define SG2IM_WGH_WAVES 1         // wavefronts per SIMD the register budget is cut for.  1: 180 + 144 registers, no spill -
#endif                            // measured faster than 2 (256 registers, 3 reloads per patch): bf16 step 4.39-4.40 vs 4.42-4.43 ms
// XB: storage of the input sources - 0 every source float32, 1 every source bfloat16, 2 mixed (per-lane loads: a
// 64-channel block may straddle two sources); YB: dY holds bfloat16.  Template parameters, not run-time flags: a branch
// inside the loaders splits the stage into basic blocks and cost 30-45 % on every layer (round 6, measured).
template <int XB, bool YB>
__global__ __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(SG2IM_WGH_WAVES))) void conv_wgrad_halo_h_kernel(const WgHaloParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int RT = 4, CT = 16;
  constexpr int PX = RT * CT;                       // 64 pixels = 4 K steps of 16
  constexpr int HWD = CT + 2, HP = (RT + 2) * HWD;  // 18 x 6 = 108 halo pixels
  constexpr int NX = (HP + 15) / 16;                // halo float4 per thread (16 threads span the 64 channels)
  constexpr int NY = PX / 16;                       // dY float4 per thread
  constexpr int IMG = (PX + HP) * WGH_LD;           // bf16 elements of one image set: Ys [PX] + Xs [HP]
  bf16_t* const lds = reinterpret_cast<bf16_t*>(smem);
  const ConvGeom& g = p.g;
  const int tid = threadIdx.x;
  const int cb0 = blockIdx.x * 64, k0 = blockIdx.y * 64, split = blockIdx.z;
  const int p_lo = split * p.per, p_hi = min(p.npatch, p_lo + p.per);
  const int col4 = tid & 15, r0 = tid >> 4;

  // ---- input operand: this thread's four channels belong to one source for the whole launch ----
  const int xc = cb0 + 4 * col4;
  const bool xok = xc < g.Ctot;
  int xs_ = 0, xcs = 0;
  locate_channel(g, xok ? xc : 0, xs_, xcs);
  const Src XS = pick_src(g, xs_);
  Aff xaff;
  fetch_aff(xaff, XS, xcs, xok);
  const int Hs = g.H >> XS.up, Ws = g.W >> XS.up;
  const int yk = k0 + 4 * col4;
  const bool yok = yk < p.Cout;
  const BufRsrc rsY = rsrc_of(p.dY, (unsigned)(g.NB * g.H * g.W) * (unsigned)p.ldy * (YB ? 2u : 4u));
  const bool xbf = XB == 1 || (XB == 2 && XS.bf != 0);   // (per thread: its four channels' source holds bfloat16)

  float4 rx[NX], ry[NY];
  unsigned rxm = 0;
  auto load_patch = [&](int pt) {
    int t = pt;
    const int tx = t % p.tiles_x; t /= p.tiles_x;
    const int ty = t % p.tiles_y;
    const int nb = t / p.tiles_y, y0 = ty * RT, x0 = tx * CT;
    rxm = 0;
    #pragma unroll
    for (int j = 0; j < NX; ++j) {
      const int hp = r0 + 16 * j;
      const int hr = hp / HWD, hc = hp - hr * HWD;
      const int ay = y0 - 1 + hr, ax = x0 - 1 + hc;
      const bool ok = xok && hp < HP && (unsigned)ay < (unsigned)g.H && (unsigned)ax < (unsigned)g.W;
      rxm |= (ok ? 1u : 0u) << j;
      const int ayc = min(max(ay, 0), g.H - 1), axc = min(max(ax, 0), g.W - 1);
      const unsigned xo = (unsigned)((nb * Hs + (ayc >> XS.up)) * Ws + (axc >> XS.up)) * (unsigned)XS.ld + (unsigned)xcs;
      if constexpr (XB == 0) rx[j] = ld4_off(XS.p, xo);
      else if constexpr (XB == 1) rx[j] = ld2h_off(XS.p, xo);
      else {
        // mixed block: ONE 16-byte load per lane at a per-lane BYTE offset (a bfloat16 lane fetches 8 elements and uses
        // the first four) - no per-lane branch around the loads.  A bfloat16 source must therefore be readable 8 bytes
        // past its last element (include/sg2im_hip.h; sg2im_amd.functional._new_s allocates the slack).
        unsigned bo = xbf ? xo << 1 : xo << 2;
        asm volatile("" : "+v"(bo));
        rx[j] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(XS.p) + (size_t)bo);
      }
    }
    #pragma unroll
    for (int j = 0; j < NY; ++j) {
      const int q = r0 + 16 * j;
      const int pix = (nb * g.H + y0 + q / CT) * g.W + x0 + q % CT;
      if constexpr (YB) ry[j] = ld2h_buf(rsY, yok ? ((unsigned)pix * (unsigned)p.ldy + (unsigned)yk) << 1 : kOobByte);
      else ry[j] = ld4_buf(rsY, yok ? ((unsigned)pix * (unsigned)p.ldy + (unsigned)yk) << 2 : kOobByte);     // (zeros when !yok)
    }
  };
  const bool want_db = p.dbias != nullptr && blockIdx.x == 0;
  float4 dbs = zero4();
  auto stage_patch = [&](bf16_t* img) {
    #pragma unroll
    for (int j = 0; j < NY; ++j) {
      const int q = r0 + 16 * j;
      if constexpr (YB) ry[j] = unpack_bf16x4(ry[j]);
      if (want_db) { dbs.x += ry[j].x; dbs.y += ry[j].y; dbs.z += ry[j].z; dbs.w += ry[j].w; }     // (the bias gradient sums the fp32 dY)
      *reinterpret_cast<bf16x4*>(img + q * WGH_LD + 4 * col4) = to_bf16x4(ry[j]);
    }
    #pragma unroll
    for (int j = 0; j < NX; ++j) {
      const int hp = r0 + 16 * j;
      if (NX * 16 <= HP || hp < HP)
        *reinterpret_cast<bf16x4*>(img + (PX + hp) * WGH_LD + 4 * col4) =
          to_bf16x4(apply_aff(XB == 0 ? rx[j] : XB == 1 ? unpack_bf16x4(rx[j]) : (xbf ? unpack_bf16x4(rx[j]) : rx[j]), xaff, (rxm >> j & 1u) != 0));
    }
  };

  // ---- fragments: wave (kwv, cwv) owns 32 output channels x 32 input channels x 9 taps ----
  const int wave = tid >> 6, lane = tid & 63;
  const int kwv = wave & 1, cwv = wave >> 1;
  const int li = lane & 31, lh = lane >> 5;
  f32x16 acc[9];
  #pragma unroll
  for (int t = 0; t < 9; ++t) zero_acc(acc[t]);
  auto mma_patch = [&](const bf16_t* img) {
    const bf16_t* const Ys = img;
    const bf16_t* const Xs = img + PX * WGH_LD;
    bf16x8 a[RT];
    #pragma unroll
    for (int r = 0; r < RT; ++r) a[r] = read_tr_at(Ys, CT * r, kwv * 32, lane);
    #pragma unroll
    for (int hr = 0; hr < RT + 2; ++hr) {           // halo row hr meets patch row r under kernel row kh = hr - r
      #pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const bf16x8 b = read_tr_at(Xs, hr * HWD + kw, cwv * 32, lane);
        #pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int r = hr - kh;
          if (r >= 0 && r < RT)
            acc[3 * kh + kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[r], b, acc[3 * kh + kw], 0, 0, 0);
        }
      }
    }
  };

  if (p_lo < p_hi) {
    load_patch(p_lo);
    stage_patch(lds);
    __syncthreads();
    #pragma unroll 1
    for (int pt = p_lo; pt < p_hi; ++pt) {
      const bool more = pt + 1 < p_hi;
      const int w = (pt - p_lo) & 1;
      if (more) load_patch(pt + 1);                  // (in flight while this patch is on the matrix cores)
      __builtin_amdgcn_sched_barrier(0);
      mma_patch(lds + w * IMG);
      if (more) stage_patch(lds + (w ^ 1) * IMG);
      __syncthreads();                               // the other image is complete; this one is free for the patch after next
    }
  }

  // ---- epilogue: as conv_wgrad_halo_kernel (same C/D layout: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)) ----
  const int Ntot = 9 * g.Ctot;
  const int c = cb0 + cwv * 32 + li;
  if (c < g.Ctot) {
    #pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int n = tap * g.Ctot + c;
      #pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = k0 + kwv * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m >= p.Cout) continue;
        float v = acc[tap][r];
        if (p.e.nsplit > 1) {
          p.e.ws[((long long)split * p.Cout + m) * Ntot + n] = v;
        } else {
          float* dst = p.e.C + (long long)m * p.e.ldc + epi_col(p.e, n);
          if (p.e.accumulate) v += *dst;
          *dst = v;
        }
      }
    }
  }
  if (p.dbias != nullptr && blockIdx.x == 0) {
    float4* red = reinterpret_cast<float4*>(smem);     // [16][16]  (the operand images are free now)
    __syncthreads();
    red[r0 * 16 + col4] = dbs;
    __syncthreads();
    if (tid < 16 && k0 + 4 * tid < p.Cout) {
      float4 t = red[tid];
      for (int k = 1; k < 16; ++k) {
        const float4 u = red[k * 16 + tid];
        t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
      }
      const float tv[4] = {t.x, t.y, t.z, t.w};
      float* dst = p.e.nsplit > 1 ? p.ws_bias + (size_t)split * p.Cout : p.dbias;
      for (int j = 0; j < 4; ++j) {
        const int k = k0 + 4 * tid + j;
        if (k >= p.Cout) break;
        if (p.e.nsplit > 1 || !p.e.accumulate) dst[k] = tv[j];
        else dst[k] += tv[j];
      }
    }
  }
}

constexpr size_t wgrad_halo_h_lds() {                 // two image sets of (64 + 108) pixels x 96 bf16
  return 2 * (size_t)(64 + 108) * WGH_LD * sizeof(bf16_t);
}

template <int RT, int CT> constexpr size_t wgrad_halo_lds() {          // two images
  return 2 * ((size_t)RT * CT + (size_t)(RT + 2) * (CT + 2)) * 64 * sizeof(float);
}

}  // namespace sg2im
