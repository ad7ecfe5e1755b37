// Weight gradient of 3x3 / stride 1 / pad 1 convolutions with a ROW-HALO'D input tile:
//   dW[co][kh][kw][c] = sum_pix dY[pix][co] * X[pix + (kh - 1, kw - 1)][c]
// The first-generation kernel (conv.hip conv_wgrad_body) tiles the flat (tap, channel) column axis: a 64-channel
// layer's 576 columns are nine 64-column tiles, each of which re-reads - and re-activates - the input pixels of
// its K chunk for its own tap.  Here a column tile is ONE kernel row kh x the three taps kw = 0, 1, 2 x a block
// of 64 channels (192 columns): the 32 output pixels of a K chunk lie in one output row ("fast rows":
// Wo % 32 == 0), so the three taps read the same input row shifted by one pixel - the workgroup stages the
// 34 pixels [wo - 1, wo + 32] x 64 channels once, activates them once, and forms the three taps' B fragments
// from that image at pixel offsets 0, 1, 2.  Per MFMA the loader moves 2.1 instead of 4.5 float4 per thread
// and runs the pending affine on 0.47x the data; dY is staged once for three taps instead of once per tap.
// 64 output channels per workgroup (2 x 2 wavefronts, 32 x 96 per wave: 48 accumulators).
#pragma once
#include "igemm.h"

namespace sg2im {

struct Wgrad3Params {
  ConvGeom g;
  const float* dY;      // [NB*Ho*Wo][ldy]
  int ldy;
  int Cout;
  int P;                // NB*Ho*Wo (reduction length), a multiple of 32
  int iters;            // P / 32
  int ncb;              // 64-channel blocks of the virtual concat: ceil(Ctot / 64)
  Epi e;                // C = dW, ldc = 9 * Wtap; split-K partials [split][Cout][9 * Ctot]
  float* dbias;         // optional [Cout]
  float* ws_bias;       // [nsplit][Cout] behind the dW partials
  int background;       // host side only
};

constexpr int W3_BM = 64, W3_CB = 64, W3_BN = 3 * W3_CB, W3_PIX = BK + 2;
constexpr int W3_ALD = W3_BM + KPAD, W3_BLD = W3_CB + KPAD;
constexpr size_t kWgrad3Lds = (size_t)(BK * W3_ALD + W3_PIX * W3_BLD) * sizeof(float);

__global__ __launch_bounds__(NTHREADS) void conv_wgrad3_kernel(const Wgrad3Params p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const As = smem;                         // dY chunk, k-major [32 pixels][64 + 4]
  float* const Bs = smem + BK * W3_ALD;           // input row halo, k-major [34 pixels][64 + 4]
  const ConvGeom& g = p.g;
  const int tid = threadIdx.x;
  const int kh = blockIdx.x % 3, cb = blockIdx.x / 3;          // kernel row, 64-channel block
  const int m0 = blockIdx.y * W3_BM, split = blockIdx.z;
  const int per = (p.iters + p.e.nsplit - 1) / p.e.nsplit;
  const int it_begin = split * per;
  const int it_end = min(p.iters, it_begin + per);
  const int HoWo = g.Ho * g.Wo;

  // A loader: dY, thread -> (channel quad acol4, pixel rows ak0 + 16 i)
  constexpr int QA = W3_BM / 4;
  const int acol4 = tid % QA, ak0 = tid / QA;
  const int aco = m0 + 4 * acol4;
  const bool aok = aco < p.Cout;
  const BufRsrc rsY = rsrc_of(p.dY, (unsigned)p.P * (unsigned)p.ldy * 4u);

  // B loader: thread -> (channel quad cq of the block, pixel slots ps, ps + 16 and - for ps < 2 - 32 + ps)
  const int cq = tid & 15, ps = tid >> 4;
  const int bc = cb * W3_CB + 4 * cq;                          // channel of the virtual concat
  const bool bok = bc < g.Ctot;
  int bs = 0, bcs = 0;
  locate_channel(g, bok ? bc : 0, bs, bcs);
  const Src BS = pick_src(g, bs);
  const int Hs = g.H >> BS.up, Ws = g.W >> BS.up;
  Aff baff;
  fetch_aff(baff, BS, bcs, bok);
  // wave-uniform cursor of the chunk's first output pixel
  int c_n, c_ho, c_wo;
  {
    const int pix0 = it_begin * BK;
    c_n = pix0 / HoWo;
    const int rem = pix0 - c_n * HoWo;
    c_ho = rem / g.Wo; c_wo = rem - c_ho * g.Wo;
  }

  float4 ra[2], rb[3];
  unsigned rbm = 0;
  auto load = [&](int it) {
    #pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pix = it * BK + ak0 + 16 * i;
      ra[i] = ld4_buf(rsY, aok ? ((unsigned)pix * (unsigned)p.ldy + (unsigned)aco) << 2 : kOobByte);
    }
    const int hi = c_ho + kh - 1;
    const bool okh = bok && (unsigned)hi < (unsigned)g.H;
    const unsigned rowbase = (unsigned)((c_n * Hs + ((okh ? hi : 0) >> BS.up)) * Ws);
    unsigned m = 0;
    #pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int px = ps + 16 * i;                               // pixel slot 0 .. 33 (i == 2: only ps < 2)
      const int wi = c_wo - 1 + px;
      const bool ok = okh && (i < 2 || ps < 2) && (unsigned)wi < (unsigned)g.W;
      m |= (ok ? 1u : 0u) << i;
      rb[i] = ld4_off(BS.p, ok ? (rowbase + (unsigned)(wi >> BS.up)) * (unsigned)BS.ld + (unsigned)bcs : 0u);
    }
    rbm = m;
    if (it + 1 < it_end) {                                      // (wave-uniform: scalar unit)
      c_wo += BK;
      if (c_wo >= g.Wo) { c_wo = 0; if (++c_ho >= g.Ho) { c_ho = 0; ++c_n; } }
    }
  };
  const bool want_db = p.dbias != nullptr && blockIdx.x == 0;
  float4 dbs = zero4();
  auto stage = [&](bool live) {
    #pragma unroll
    for (int i = 0; i < 2; ++i) {
      *reinterpret_cast<float4*>(As + (ak0 + 16 * i) * W3_ALD + 4 * acol4) = ra[i];
      if (want_db && live) { dbs.x += ra[i].x; dbs.y += ra[i].y; dbs.z += ra[i].z; dbs.w += ra[i].w; }
    }
    #pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float4 v = apply_aff(rb[i], baff, (rbm >> i & 1u) != 0);
      if (i < 2 || ps < 2) *reinterpret_cast<float4*>(Bs + (ps + 16 * i) * W3_BLD + 4 * cq) = v;
    }
  };

  const int wave = tid >> 6, lane = tid & 63;
  const int wm0 = (wave >> 1) * 32, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  float fa[16], fb[3][16];
  f32x16 acc[3];
  #pragma unroll
  for (int t = 0; t < 3; ++t) zero_acc(acc[t]);
  auto mma = [&](int phase) {
    if (phase == 0) {
      #pragma unroll
      for (int s = 0; s < 16; ++s) fa[s] = As[kperm(s, lh) * W3_ALD + wm0 + li];
      #pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int ct = 3 * wn + t;                              // column tile: tap kw = ct >> 1, channel half ct & 1
        #pragma unroll
        for (int s = 0; s < 16; ++s) fb[t][s] = Bs[(kperm(s, lh) + (ct >> 1)) * W3_BLD + (ct & 1) * 32 + li];
      }
    } else {
      #pragma unroll
      for (int s = 0; s < 16; ++s)
        #pragma unroll
        for (int t = 0; t < 3; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s], fb[t][s], acc[t], 0, 0, 0);
    }
  };
  k_pipeline(it_begin, it_end, load, stage, mma);

  // epilogue: GEMM column n = tap * Ctot + channel (the flat axis of the split-K partials and of epi_col)
  {
    const Epi& e = p.e;
    const int Ntot = 9 * g.Ctot;
    #pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int ct = 3 * wn + t;
      const int ch = cb * W3_CB + (ct & 1) * 32 + li;
      if (ch >= g.Ctot) continue;
      const int n = (kh * 3 + (ct >> 1)) * g.Ctot + ch;
      const int ncol = epi_col(e, n);
      #pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m >= p.Cout) continue;
        float v = acc[t][r];
        if (e.nsplit > 1) {
          e.ws[((long long)split * p.Cout + m) * Ntot + n] = v;
        } else {
          float* dst = e.C + (long long)m * e.ldc + ncol;
          if (e.accumulate) v += *dst;
          *dst = v;
        }
      }
    }
  }
  if (p.dbias != nullptr && blockIdx.x == 0) {
    // the 256 / QA threads that share a channel quad hold sums over disjoint pixel rows: combined through LDS in
    // thread order (fixed order -> reproducible), as in conv_wgrad_body
    constexpr int GROUPS = NTHREADS / QA;
    float4* red = reinterpret_cast<float4*>(smem);
    __syncthreads();
    red[ak0 * QA + acol4] = dbs;
    __syncthreads();
    if (tid < QA && aco < p.Cout) {
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

}  // namespace sg2im
