// A discriminator CNN of 'CK-X-S' tokens with BatchNorm (reference sg2im/layers.py:129-213 as the discriminators use
// it, sg2im/discriminators.py:25-45,48-64: conv, then [BatchNorm2d, LeakyReLU, conv] ...) FORWARD as ONE persistent
// launch: stages separated by the grid barrier of gcn_persist.hip.
//
// Why: between the refinement network's forward and backward pass the generator loss runs both discriminators over
// the generated images - two dependent chains of ~20 launches of 5-30 us each with the chip nearly idle (0.26 ms
// forward).  Per layer i the launch path needs conv (+ split-K finish), statistics partial /
// finish launches; here:
//   conv_i      y_i = conv(act_{i-1}) + b_i as an implicit GEMM over 32 x 64 tiles (rows = output pixels, K = taps x
//               channels); act_{i-1} = leaky(scale_{i-1} y_{i-1} + shift_{i-1}) is applied by the operand loader (the
//               normalised tensor never exists, as on the launch path); per 32-row tile and channel the epilogue leaves
//               (pivot, sum (y - pivot), sum (y - pivot)^2) of the tile's LIVE rows      [barrier]
//   finish_i    a wavefront per channel combines the tile partials (Chan et al., in double, fixed order) into mean /
//               invstd / scale / shift and moves the running statistics `training` times    [barrier]
// The last convolution has no BatchNorm behind it.  Same arithmetic per element as the launch path's kernels
// (bn_stats_final_tiles_kernel) up to the summation order inside a 32-row tile.
//
// Residency: every workgroup must be resident (grid barrier).  A workgroup takes 98 KB of LDS = one per CU, and the
// generator loss runs D_obj and D_img side by side on two streams: the launcher takes at most HALF the CUs per launch,
// so that two such launches can always be resident together (two whole-chip persistent kernels dispatched
// concurrently can starve each other of CUs for good).
#define SG2IM_PERSIST_HELPERS_ONLY
#include "gcn_persist.hip"         // (opens namespace sg2im::gcn and leaves it open)

// ---- operand loaders of a convolution -------------------------------------------------------------------------
// A[m][k], m = (n, ho, wo), k = (kh * KW + kw) * Cin + ci over an NHWC tensor; Cin a multiple of 32: a 32-wide K chunk
// lies inside one tap.  Optional pending per-channel affine + leaky ReLU (the BatchNorm + activation in front of this
// convolution); taps outside the image are zeros (AFTER the affine: padding pads the activated tensor).
struct RowsConv {
  const float* X; const float* scale; const float* shift; float slope;
  int Cin, KW, W, H, stride, pad, cpt;          // cpt: chunks per tap = Cin / 32
  int base[4], hi0[4], wi0[4];
  __device__ __forceinline__ void init(const float* X_, int H_, int W_, int Cin_, int KW_, int stride_, int pad_, int Ho, int Wo,
                                       const float* scale_, const float* shift_, float slope_, int m0, int M, const Lane& L) {
    X = X_; H = H_; W = W_; Cin = Cin_; KW = KW_; stride = stride_; pad = pad_; cpt = Cin_ >> 5;
    scale = scale_; shift = shift_; slope = slope_;
    #pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = min(m0 + L.r8 + 8 * j, M - 1);
      const int wo = m % Wo, t = m / Wo, ho = t % Ho, n = t / Ho;
      hi0[j] = ho * stride - pad; wi0[j] = wo * stride - pad;
      base[j] = n * H * W;
    }
  }
  __device__ __forceinline__ void chunk(int c, v4f (&d)[4]) const {
    const int tap = c / cpt, ci = ((c - tap * cpt) << 5);            // (wave-uniform)
    const int kh = tap / KW, kw = tap - kh * KW;
    const int c4 = (threadIdx.x & 7) << 2;
    bool ok[4];
    #pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int hi = hi0[j] + kh, wi = wi0[j] + kw;
      ok[j] = (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;
      const int hc = min(max(hi, 0), H - 1), wc = min(max(wi, 0), W - 1);
      d[j] = ld4(X + (base[j] + hc * W + wc) * Cin + ci + c4);
    }
    if (scale) {
      const v4f sc = ld4(scale + ci + c4), sh = ld4(shift + ci + c4);
      #pragma unroll
      for (int j = 0; j < 4; ++j) {
        v4f v = d[j] * sc + sh;
        v.x = v.x > 0.f ? v.x : v.x * slope; v.y = v.y > 0.f ? v.y : v.y * slope;
        v.z = v.z > 0.f ? v.z : v.z * slope; v.w = v.w > 0.f ? v.w : v.w * slope;
        d[j] = v;
      }
    }
    #pragma unroll
    for (int j = 0; j < 4; ++j) d[j] = ok[j] ? d[j] : zero4();
  }
};

// The same for a few input channels (the RGB input of the first convolution: K = KH * KW * Cin = 48 for a 4 x 4
// kernel, at most 64 = two chunks): scalar loads; which (kh, kw, ci) a lane's four elements of a chunk are does not
// depend on the row, so the divisions are done once per tile.  No pending affine.
struct RowsConvFew {
  const float* X;
  int Cin, W, H;
  int base[4], hi0[4], wi0[4];
  int kh0[4], kw0[4], ko0[4], kh1[4], kw1[4], ko1[4];           // chunk 0 / 1, element q: tap row, tap column, (kh W + kw) Cin + ci
  bool v0[4], v1[4];
  __device__ __forceinline__ void init(const float* X_, int H_, int W_, int Cin_, int KH, int KW, int stride, int pad, int Ho, int Wo,
                                       int m0, int M, const Lane& L) {
    X = X_; H = H_; W = W_; Cin = Cin_;
    const int KWC = KW * Cin_, K = KH * KWC;
    #pragma unroll
    for (int q = 0; q < 4; ++q) {
      int k = 4 * L.c4 + q;
      v0[q] = k < K; k = min(k, K - 1);
      kh0[q] = k / KWC; kw0[q] = (k - kh0[q] * KWC) / Cin_; ko0[q] = (kh0[q] * W_ + kw0[q]) * Cin_ + (k - kh0[q] * KWC - kw0[q] * Cin_);
      k = 32 + 4 * L.c4 + q;
      v1[q] = k < K; k = min(k, K - 1);
      kh1[q] = k / KWC; kw1[q] = (k - kh1[q] * KWC) / Cin_; ko1[q] = (kh1[q] * W_ + kw1[q]) * Cin_ + (k - kh1[q] * KWC - kw1[q] * Cin_);
    }
    #pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = min(m0 + L.r8 + 8 * j, M - 1);
      const int wo = m % Wo, t = m / Wo, ho = t % Ho, n = t / Ho;
      hi0[j] = ho * stride - pad; wi0[j] = wo * stride - pad;
      base[j] = ((n * H_ + hi0[j]) * W_ + wi0[j]) * Cin_;
    }
  }
  __device__ __forceinline__ void chunk(int c, v4f (&d)[4]) const {
    const bool first = c == 0;                                       // (wave-uniform; K <= 64: chunks 0 and 1 only)
    #pragma unroll
    for (int j = 0; j < 4; ++j) {
      float e[4];
      #pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int kh = first ? kh0[q] : kh1[q], kw = first ? kw0[q] : kw1[q], ko = first ? ko0[q] : ko1[q];
        const bool ok = (first ? v0[q] : v1[q]) && (unsigned)(hi0[j] + kh) < (unsigned)H && (unsigned)(wi0[j] + kw) < (unsigned)W;
        const float v = X[ok ? base[j] + ko : 0];
        e[q] = ok ? v : 0.f;
      }
      d[j] = v4f{e[0], e[1], e[2], e[3]};
    }
  }
};

// m-major weights W[row0 + r][32 c + ..] with a K bound (K = 48: the second chunk is half empty), rows clamped
struct RowsMK {
  const float* X; int off[4]; int K;
  __device__ __forceinline__ void init(const float* X_, int ld, int row0, int nrows, int K_, const Lane& L) {
    X = X_; K = K_;
    #pragma unroll
    for (int j = 0; j < 4; ++j) off[j] = min(row0 + L.r8 + 8 * j, nrows - 1) * ld;
  }
  __device__ __forceinline__ void chunk(int c, v4f (&d)[4]) const {
    const int k0 = 32 * c + ((threadIdx.x & 7) << 2);
    if ((K & 3) == 0) {                                               // (rows 16-byte aligned: whole pieces in or out)
      #pragma unroll
      for (int j = 0; j < 4; ++j) d[j] = k0 < K ? ld4(X + off[j] + min(k0, K - 4)) : zero4();
    } else {
      #pragma unroll
      for (int j = 0; j < 4; ++j) {
        float e[4];
        #pragma unroll
        for (int q = 0; q < 4; ++q) e[q] = k0 + q < K ? X[off[j] + min(k0 + q, K - 1)] : 0.f;
        d[j] = v4f{e[0], e[1], e[2], e[3]};
      }
    }
  }
};

struct EpiBiasStore {              // out[m0 + row][n0 + col] = v + bias
  float* out; const float* bias; int ldo, m0, n0, M, N;
  __device__ __forceinline__ void operator()(int row, int col, v4f v) const {
    const int m = m0 + row, n = n0 + col;
    if (m >= M || n >= N) return;
    if (bias) v = v + ld4(bias + n);
    st4(out + m * ldo + n, v);
  }
};

struct DiscFwdArgs {
  sg2im_disc_stack s;              // (first: layers are fetched from the kernel-argument segment)
  unsigned* sync;
};

__device__ __forceinline__ sg2im_disc_layer fetch_disc_layer(int l) {
  typedef __attribute__((address_space(4))) const char* KPtr;
  typedef __attribute__((address_space(4))) const sg2im_disc_layer* KLayer;
  const KLayer k = (KLayer)((KPtr)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(sg2im_disc_stack, layer)) + l;
  sg2im_disc_layer L;
  L.weight = k->weight; L.bias = k->bias; L.out = k->out;
  L.gamma = k->gamma; L.beta = k->beta; L.running_mean = k->running_mean; L.running_var = k->running_var;
  L.num_batches_tracked = k->num_batches_tracked;
  L.mean = k->mean; L.invstd = k->invstd; L.scale = k->scale; L.shift = k->shift; L.partial = k->partial;
  L.cin = k->cin; L.cout = k->cout; L.kh = k->kh; L.kw = k->kw; L.stride = k->stride; L.pad = k->pad;
  L.in_h = k->in_h; L.in_w = k->in_w; L.out_h = k->out_h; L.out_w = k->out_w;
  return L;
}

// one 32 x 64 (NB = 2) or 32 x 32 tile of a convolution + its statistics partial
template <int NB, bool FEW>
__device__ __forceinline__ void disc_conv_tile(const sg2im_disc_layer& Ly, const float* x, const float* scale, const float* shift,
                                               float slope, int batch, long long live, int tile, int nrb, float* smem, const Lane& L) {
  const int cb = tile / nrb, rb = tile - cb * nrb;
  const int m0 = rb << 5, n0 = cb * 32 * NB;
  const int M = batch * Ly.out_h * Ly.out_w, K = Ly.kh * Ly.kw * Ly.cin;
  RowsMK w[NB];
  #pragma unroll
  for (int nb = 0; nb < NB; ++nb) w[nb].init(Ly.weight, K, n0 + 32 * nb, Ly.cout, K, L);
  EpiBiasStore epi;
  epi.out = Ly.out; epi.bias = Ly.bias; epi.ldo = Ly.cout; epi.m0 = m0; epi.n0 = n0; epi.M = M; epi.N = Ly.cout;
  if (FEW) {
    RowsConvFew a;
    a.init(x, Ly.in_h, Ly.in_w, Ly.cin, Ly.kh, Ly.kw, Ly.stride, Ly.pad, Ly.out_h, Ly.out_w, m0, M, L);
    tile_gemm<NB, false, false>(a, w, (K + 31) >> 5, smem, L, epi);
  } else {
    RowsConv a;
    a.init(x, Ly.in_h, Ly.in_w, Ly.cin, Ly.kw, Ly.stride, Ly.pad, Ly.out_h, Ly.out_w, scale, shift, slope, m0, M, L);
    tile_gemm<NB, false, false>(a, w, K >> 5, smem, L, epi);
  }
  if (Ly.partial) {
    // statistics of the tile's live rows, per column: the value that was stored, recomputed from the four wave partials
    // (same order of additions as tile_gemm's epilogue) + bias; pivot = the tile's first row
    const int col = threadIdx.x;
    if (col < 32 * NB && n0 + col < Ly.cout) {
      const float* red = smem + kStageFloats;
      const float b = Ly.bias ? Ly.bias[n0 + col] : 0.f;
      const long long nlive = live - m0;                           // rows of this tile that are real
      const int rows = (int)(nlive < 0 ? 0 : nlive > 32 ? 32 : nlive);
      float pv = 0.f, s = 0.f, ss = 0.f;
      for (int r = 0; r < rows; ++r) {
        const float* p = red + r * kRedLd + col;
        const float v = (((p[0] + p[32 * kRedLd]) + p[64 * kRedLd]) + p[96 * kRedLd]) + b;
        if (r == 0) pv = v;
        const float dlt = v - pv;
        s += dlt; ss += dlt * dlt;
      }
      const size_t plane = (size_t)Ly.cout * nrb;
      float* dst = Ly.partial + (size_t)(n0 + col) * nrb + rb;
      dst[0] = pv; dst[plane] = s; dst[2 * plane] = ss;
    }
  }
}

// finish of the tile partials of one BatchNorm: a wavefront per channel (the arithmetic of bn_stats_final_tiles_kernel)
__device__ __forceinline__ void disc_bn_finish(const sg2im_disc_layer& Ly, int nblk, long long rows, long long live, float eps,
                                               float momentum, int updates, const Lane& L) {
  const int C = Ly.cout;
  if (blockIdx.x == 0 && threadIdx.x == 0 && Ly.num_batches_tracked) *Ly.num_batches_tracked += updates;
  for (int c = blockIdx.x * 4 + L.wave; c < C; c += gridDim.x * 4) {
    const size_t plane = (size_t)C * nblk;
    const float* q = Ly.partial + (size_t)c * nblk;
    const double P = (double)q[0];
    double sn = 0.0, sm = 0.0, sq = 0.0;
    for (int t = L.lane; t < nblk; t += 64) {
      long long nt = live - (long long)t * 32;
      if (nt <= 0) continue;
      if (nt > 32) nt = 32;
      const double n = (double)nt, s = (double)q[plane + t], ss = (double)q[2 * plane + t];
      const double mt = ((double)q[t] - P) + s / n;
      double m2 = ss - s * s / n;
      if (m2 < 0.0) m2 = 0.0;
      sn += n; sm += n * mt; sq += m2 + n * mt * mt;
    }
    #pragma unroll
    for (int off = 32; off > 0; off >>= 1) { sn += __shfl_xor(sn, off); sm += __shfl_xor(sm, off); sq += __shfl_xor(sq, off); }
    if (L.lane != 0) continue;
    const double N = sn > 0.0 ? sn : 1.0;
    const double dm = sm / N;
    const double mu = P + dm;
    double var = sq / N - dm * dm;
    if (var < 0.0) var = 0.0;
    if (Ly.running_mean) {
      const double nu = (double)(live > 0 ? live : 1);
      const double unbiased = nu > 1.0 ? var * nu / (nu - 1.0) : var;
      float rm = Ly.running_mean[c], rv = Ly.running_var[c];
      for (int u = 0; u < updates; ++u) {
        rm = (float)((1.0 - momentum) * rm + momentum * mu);
        rv = (float)((1.0 - momentum) * rv + momentum * unbiased);
      }
      Ly.running_mean[c] = rm; Ly.running_var[c] = rv;
    }
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    const float g = Ly.gamma ? Ly.gamma[c] : 1.f, b = Ly.beta ? Ly.beta[c] : 0.f;
    Ly.mean[c] = (float)mu; Ly.invstd[c] = is;
    const float sc = g * is;
    Ly.scale[c] = sc; Ly.shift[c] = b - (float)mu * sc;
    (void)rows;
  }
}

__global__ __launch_bounds__(kThreads) void disc_stack_fwd_kernel(const DiscFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  Sync sy = {};
  if (threadIdx.x == 0) sync_begin(sy, a.sync);
  const Lane LN = my_lane();
  const int nl = a.s.n_layers, batch = a.s.batch;
  const float* x = a.s.x;
  const float* scale = nullptr; const float* shift = nullptr;
  for (int l = 0; l < nl; ++l) {
    const sg2im_disc_layer Ly = fetch_disc_layer(l);
    const long long per = (long long)Ly.out_h * Ly.out_w;
    const long long rows = (long long)batch * per;
    long long live = rows;
    if (a.s.count) { const long long t = (long long)a.s.count[0] * a.s.count_unit * per; live = t < rows ? t : rows; }
    const int nrb = (int)((rows + 31) >> 5);
    const bool few = (Ly.cin & 31) != 0;
    {                                                           // (cout is a multiple of 64: 32 x 64 tiles)
      const TileWalk tw(nrb * (Ly.cout >> 6));
      if (few) {
        for (int t = tw.first; t < tw.end; t += tw.step) disc_conv_tile<2, true>(Ly, x, scale, shift, a.s.slope, batch, live, t, nrb, smem, LN);
      } else {
        for (int t = tw.first; t < tw.end; t += tw.step) disc_conv_tile<2, false>(Ly, x, scale, shift, a.s.slope, batch, live, t, nrb, smem, LN);
      }
    }
    if (Ly.partial) {
      grid_barrier(sy);
      disc_bn_finish(Ly, nrb, rows, live, a.s.eps, a.s.momentum, a.s.training, LN);
    }
    if (l + 1 < nl) grid_barrier(sy);
    x = Ly.out; scale = Ly.scale; shift = Ly.shift;
    if (!Ly.partial) { scale = nullptr; shift = nullptr; }
  }
  if (threadIdx.x == 0) stamp(sy);
}

static bool g_disc_ready = false;
static int g_disc_cus = 0;

static hipError_t disc_prepare() {
  if (g_disc_ready) return hipSuccess;
  int dev = 0, cus = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (e == hipSuccess)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(disc_stack_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)kLdsBytes);
  if (e == hipSuccess) { g_disc_cus = cus > 0 ? cus : 1; g_disc_ready = true; }
  return e;
}

static bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static bool disc_ok(const sg2im_disc_stack* S) {
  if (!S || S->n_layers < 1 || S->n_layers > SG2IM_DISC_MAX_LAYERS || S->batch < 1 || !S->x || S->training < 1) return false;
  int h = 0, w = 0, c = 0;
  const long long lim = 1ll << 31;
  for (int l = 0; l < S->n_layers; ++l) {
    const sg2im_disc_layer& L = S->layer[l];
    if (L.cin < 1 || L.cout < 64 || (L.cout & 63) || L.kh < 1 || L.kw < 1 || L.stride < 1 || L.pad < 0) return false;
    if (L.out_h != (L.in_h + 2 * L.pad - L.kh) / L.stride + 1 || L.out_w != (L.in_w + 2 * L.pad - L.kw) / L.stride + 1) return false;
    if (L.out_h < 1 || L.out_w < 1) return false;
    if (l > 0 && (L.in_h != h || L.in_w != w || L.cin != c)) return false;        // reads the previous layer's output
    if (l > 0 && (L.cin & 31)) return false;                                      // vector loader: 32-wide K chunks inside a tap
    if (l == 0 && (L.cin & 31) && L.kh * L.kw * L.cin > 64) return false;           // scalar loader: at most two K chunks
    h = L.out_h; w = L.out_w; c = L.cout;
    if (!L.weight || !L.out || !al16(L.weight) || !al16(L.out) || (L.bias && !al16(L.bias))) return false;
    if ((long long)S->batch * L.in_h * L.in_w * L.cin >= lim || (long long)S->batch * L.out_h * L.out_w * L.cout >= lim ||
        (long long)L.cout * L.kh * L.kw * L.cin >= lim)
      return false;
    const bool bn = L.partial != nullptr;
    if (bn != (l + 1 < S->n_layers)) return false;                                // BatchNorm behind every convolution but the last
    if (bn && (!L.mean || !L.invstd || !L.scale || !L.shift || !al16(L.scale) || !al16(L.shift))) return false;
    if (bn && ((L.running_mean == nullptr) != (L.running_var == nullptr))) return false;
  }
  return true;
}

}  // namespace gcn
}  // namespace sg2im

using namespace sg2im;

extern "C" {

int sg2im_disc_stack_supported(int n_layers, const int* cin, const int* cout, const int* ksize) {
  if (n_layers < 1 || n_layers > SG2IM_DISC_MAX_LAYERS || !cin || !cout || !ksize) return 0;
  for (int l = 0; l < n_layers; ++l) {
    if (cout[l] < 64 || (cout[l] & 63) || ksize[l] < 1) return 0;
    if (l == 0 ? ((cin[l] & 31) && ksize[l] * ksize[l] * cin[l] > 64) : (cin[l] & 31)) return 0;
  }
  return 1;
}

size_t sg2im_disc_stack_partial_floats(int batch, int out_h, int out_w, int cout) {
  return 3 * (size_t)cout * (((size_t)batch * out_h * out_w + 31) / 32);
}

int sg2im_disc_stack_forward(const sg2im_disc_stack* S, void* sync, size_t sync_bytes, hipStream_t stream) {
  if (!gcn::disc_ok(S) || !sync || sync_bytes < sg2im_gconv_stack_sync_bytes() || !gcn::al16(sync)) return SG2IM_ERR_ARG;
  if (gcn::disc_prepare() != hipSuccess) return SG2IM_ERR_HIP;
  if (hipMemsetAsync(sync, 0, sg2im_gconv_stack_sync_bytes(), stream) != hipSuccess) return SG2IM_ERR_HIP;
  gcn::DiscFwdArgs a;
  std::memcpy(&a.s, S, sizeof(*S));
  a.sync = static_cast<unsigned*>(sync);
  // at most half the CUs: two of these launches (D_obj and D_img of the generator loss) must be able to be resident
  // together, whatever order the hardware dispatches their workgroups in
  long long most = 1;
  for (int l = 0; l < S->n_layers; ++l) {
    const sg2im_disc_layer& L = S->layer[l];
    most = std::max(most, (((long long)S->batch * L.out_h * L.out_w + 31) / 32) * ((L.cout + 63) / 64));
  }
  int grid = (int)std::min<long long>(std::max(1, gcn::g_disc_cus / 2), most);
  if (grid >= 8) grid &= ~7;
  SG2IM_LAUNCH(gcn::disc_stack_fwd_kernel, dim3(grid), dim3(gcn::kThreads), gcn::kLdsBytes, stream, a);
  return hipGetLastError() == hipSuccess ? SG2IM_OK : SG2IM_ERR_HIP;
}

}  // extern "C"
