// BatchNorm2d statistics / backward, pooling, layout conversions and other HBM-bound
// NHWC helpers of the hot path.  All tensors are [rows][C] with the channel index
// contiguous, so a workgroup maps threadIdx -> (channel, row group): global accesses
// are coalesced along channels and per-channel reductions run down the rows, finishing
// through LDS and a per-workgroup partial that a tiny second kernel reduces in double.
#include <algorithm>
#include <hip/hip_runtime.h>
#include "launch_count.h"
#include "sg2im_hip.h"
#include "bn_final.h"

namespace sg2im {

#ifndef SG2IM_RED_BLOCKS
#define SG2IM_RED_BLOCKS 256
#endif
constexpr int RED_BLOCKS = SG2IM_RED_BLOCKS;   // row blocks of the two-stage per-channel reductions (callers size `partial` for 1024)

__device__ __forceinline__ float leakyf(float v, float slope) { return v > 0.f ? v : v * slope; }

// Padded row batches (sg2im_amd/bucketing.py): a launch is sized for `rows` rows but only the
// first count[0] * unit of them are real - the rest is padding that must not enter a statistic.
// count == nullptr: every row is real.
__device__ __forceinline__ long long live_rows(long long rows, const int* __restrict__ count, int unit) {
  if (!count) return rows;
  const long long t = (long long)count[0] * unit;
  return t < rows ? t : rows;
}

// thread -> (channel lane tx, row group ty); channels covered: tx, tx + TC, ...
struct ChanMap { int TC, TR, tx, ty; };
__device__ __forceinline__ ChanMap chan_map(int C) {
  ChanMap m;
  m.TC = C < (int)blockDim.x ? C : (int)blockDim.x;
  m.TR = blockDim.x / m.TC;
  m.tx = threadIdx.x % m.TC;
  m.ty = threadIdx.x / m.TC;      // ty >= TR -> idle thread
  return m;
}

// ---------------------------------------------------------------------------
// generic per-channel sums of up to two row-wise quantities
// ---------------------------------------------------------------------------
template <typename F>
__device__ __forceinline__ void channel_reduce2(int C, long long rows, float* partial, F f) {
  // f(row, c, s0, s1) accumulates into s0/s1.  partial: [gridDim.x][2][C]
  extern __shared__ float red[];
  const ChanMap m = chan_map(C);
  const long long per = (rows + gridDim.x - 1) / gridDim.x;
  const long long r0 = (long long)blockIdx.x * per;
  const long long r1 = r0 + per < rows ? r0 + per : rows;
  for (int c = m.tx; c < C; c += m.TC) {
    float s0 = 0.f, s1 = 0.f;
    if (m.ty < m.TR)
      for (long long r = r0 + m.ty; r < r1; r += m.TR) f(r, c, s0, s1);
    // reduce over ty through LDS (only needed when TR > 1; c loop runs once then)
    if (m.TR > 1) {
      red[threadIdx.x] = s0; red[blockDim.x + threadIdx.x] = s1;
      __syncthreads();
      if (m.ty == 0) {
        for (int t = 1; t < m.TR; ++t) { s0 += red[t * m.TC + m.tx]; s1 += red[blockDim.x + t * m.TC + m.tx]; }
      }
      __syncthreads();
    }
    if (m.ty == 0) {
      partial[((long long)blockIdx.x * 2 + 0) * C + c] = s0;
      partial[((long long)blockIdx.x * 2 + 1) * C + c] = s1;
    }
  }
}

// second stage: one wavefront per channel strides over the row-block partials and finishes
// with a shuffle tree in double (fixed order -> deterministic); every lane gets the totals
__device__ __forceinline__ void wave_partial_sums(const float* __restrict__ partial, int nblk, int C, int c,
                                                  double& s0, double& s1) {
  const int lane = threadIdx.x & 63;
  double a = 0.0, b = 0.0;
  for (int i = lane; i < nblk; i += 64) {
    a += (double)partial[((long long)i * 2) * C + c];
    b += (double)partial[((long long)i * 2 + 1) * C + c];
  }
  #pragma unroll
  for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); }
  s0 = a; s1 = b;
}

__global__ void bn_stats_partial_kernel(const float* __restrict__ x, long long rows, int C, long long ld,
                                        float* __restrict__ partial, const int* __restrict__ count, int unit) {
  rows = live_rows(rows, count, unit);
  // shifted sums: d = x - x[row 0] keeps var = E[d^2] - E[d]^2 free of the catastrophic
  // cancellation a large |mean| / std ratio causes in E[x^2] - mean^2
  channel_reduce2(C, rows, partial, [&](long long r, int c, float& s0, float& s1) {
    const float v = x[r * ld + c] - x[c];
    s0 += v; s1 = fmaf(v, v, s1);
  });
}

__global__ void bn_stats_final_kernel(const float* __restrict__ x, const float* __restrict__ partial, int nblk, long long rows,
                                      long long unbiased_rows, int C,
                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                      float eps, float momentum, int training,
                                      float* __restrict__ running_mean, float* __restrict__ running_var,
                                      long long* __restrict__ nbt, float* __restrict__ mean,
                                      float* __restrict__ invstd, float* __restrict__ scale,
                                      float* __restrict__ shift, const int* __restrict__ count, int unit) {
  const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);     // one wavefront per channel
  if (c == 0 && (threadIdx.x & 63) == 0 && training && nbt) *nbt += training;      // (training = n: the pass counts n times)
  if (count) {
    const long long live = live_rows(rows, count, unit);
    if (unbiased_rows > 0) unbiased_rows = unbiased_rows / rows * live;   // (a whole multiple of rows)
    rows = live > 0 ? live : 1;
  }
  if (c >= C) return;
  double mu, var;
  if (training) {
    double s, ss;
    wave_partial_sums(partial, nblk, C, c, s, ss);
    const double dm = s / (double)rows;                 // mean of the pivot-shifted values
    mu = (double)x[c] + dm;
    var = ss / (double)rows - dm * dm;
    if (var < 0.0) var = 0.0;
    if (running_mean && (threadIdx.x & 63) == 0) {
      const double nu = (double)(unbiased_rows > 0 ? unbiased_rows : rows);
      const double unbiased = nu > 1.0 ? var * nu / (nu - 1.0) : var;
      float rm = running_mean[c], rv = running_var[c];
      for (int u = 0; u < training; ++u) {              // the same rounding as n passes over the same batch
        rm = (float)((1.0 - momentum) * rm + momentum * mu);
        rv = (float)((1.0 - momentum) * rv + momentum * unbiased);
      }
      running_mean[c] = rm; running_var[c] = rv;
    }
  } else {
    mu = running_mean[c]; var = running_var[c];
  }
  if ((threadIdx.x & 63) != 0) return;
  const float is = (float)(1.0 / sqrt(var + (double)eps));
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  mean[c] = (float)mu; invstd[c] = is;
  const float sc = g * is;
  scale[c] = sc; shift[c] = b - (float)mu * sc;
}

// Finish of statistics whose tile partials came out of a GEMM epilogue / split-K finish (conv.hip): tile t
// covers the rows [t * per, (t + 1) * per) and holds (its own pivot p_t, sum d, sum d^2) with d = x - p_t,
// stored [3][C][nblk] (a channel's partials are one contiguous run: coalesced reads).
// Per tile: mean_t = p_t + s_t / n_t, M2_t = q_t - s_t^2 / n_t; combined in double relative to tile 0's pivot
// (Chan et al.): one wavefront per channel, lanes stride over the tiles, xor-tree at the end (fixed order).
__global__ void bn_stats_final_tiles_kernel(const float* __restrict__ partial, int nblk, long long per, long long rows,
                                            long long unbiased_rows, int C,
                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                            float eps, float momentum, int updates,
                                            float* __restrict__ running_mean, float* __restrict__ running_var,
                                            long long* __restrict__ nbt, float* __restrict__ mean,
                                            float* __restrict__ invstd, float* __restrict__ scale,
                                            float* __restrict__ shift, const int* __restrict__ count, int unit) {
  const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);     // one wavefront per channel
  const int lane = threadIdx.x & 63;
  if (c == 0 && lane == 0 && nbt) *nbt += updates;
  long long live = live_rows(rows, count, unit);
  if (count && unbiased_rows > 0) unbiased_rows = unbiased_rows / rows * live;   // (a whole multiple of rows)
  if (c >= C) return;
  const size_t plane = (size_t)C * nblk;
  const float* __restrict__ q = partial + (size_t)c * nblk;
  const double P = (double)q[0];                        // tile 0's pivot
  double sn = 0.0, sm = 0.0, sq = 0.0;
  auto add_tile = [&](int t, float pv, float sv, float ssv) {
    long long nt = live - (long long)t * per;
    if (nt <= 0) return;
    if (nt > per) nt = per;
    const double n = (double)nt, s = (double)sv, ss = (double)ssv;
    const double mt = ((double)pv - P) + s / n;
    double m2 = ss - s * s / n;
    if (m2 < 0.0) m2 = 0.0;
    sn += n; sm += n * mt; sq += m2 + n * mt * mt;
  };
  if ((nblk & 3) == 0) {            // four tiles per lane and load (a channel's partials are contiguous, 16-byte aligned)
    const float4* __restrict__ q4 = reinterpret_cast<const float4*>(q);
    const float4* __restrict__ s4 = reinterpret_cast<const float4*>(q + plane);
    const float4* __restrict__ ss4 = reinterpret_cast<const float4*>(q + 2 * plane);
    for (int t4 = lane; t4 < (nblk >> 2); t4 += 64) {
      const float4 a = q4[t4], b = s4[t4], c4 = ss4[t4];
      add_tile(4 * t4 + 0, a.x, b.x, c4.x); add_tile(4 * t4 + 1, a.y, b.y, c4.y);
      add_tile(4 * t4 + 2, a.z, b.z, c4.z); add_tile(4 * t4 + 3, a.w, b.w, c4.w);
    }
  } else {
    for (int t = lane; t < nblk; t += 64) add_tile(t, q[t], q[plane + t], q[2 * plane + t]);
  }
  #pragma unroll
  for (int off = 32; off > 0; off >>= 1) { sn += __shfl_xor(sn, off); sm += __shfl_xor(sm, off); sq += __shfl_xor(sq, off); }
  if (lane != 0) return;
  const double N = sn > 0.0 ? sn : 1.0;
  const double dm = sm / N;
  const double mu = P + dm;
  double var = sq / N - dm * dm;
  if (var < 0.0) var = 0.0;
  if (running_mean) {
    const double nu = (double)(unbiased_rows > 0 ? unbiased_rows : (live > 0 ? live : 1));
    const double unbiased = nu > 1.0 ? var * nu / (nu - 1.0) : var;
    float rm = running_mean[c], rv = running_var[c];
    for (int u = 0; u < updates; ++u) {                 // (sg2im_bn_fwd.training = n: see bn_stats_final_kernel)
      rm = (float)((1.0 - momentum) * rm + momentum * mu);
      rv = (float)((1.0 - momentum) * rv + momentum * unbiased);
    }
    running_mean[c] = rm; running_var[c] = rv;
  }
  const float is = (float)(1.0 / sqrt(var + (double)eps));
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  mean[c] = (float)mu; invstd[c] = is;
  const float sc = g * is;
  scale[c] = sc; shift[c] = b - (float)mu * sc;
}

// dz source: plain [rows][ld_g] or 2x2 sum of an upsampled-resolution tensor
struct GradSrc { const float* g; long long ld; int pool2, h, w; };
__device__ __forceinline__ float read_dz(const GradSrc& s, long long r, int c) {
  if (!s.pool2) return s.g[r * s.ld + c];
  const long long hw = (long long)s.h * s.w;
  const long long n = r / hw; const int rem = (int)(r - n * hw);
  const int y = rem / s.w, x = rem - y * s.w;
  const long long W2 = 2LL * s.w;
  const float* p = s.g + ((n * 2 * s.h + 2 * y) * W2 + 2 * x) * s.ld + c;
  return (p[0] + p[s.ld]) + (p[W2 * s.ld] + p[(W2 + 1) * s.ld]);
}

__global__ void bn_bwd_partial_kernel(GradSrc gs, const float* __restrict__ y, long long ld_y, long long rows,
                                      int C, const float* __restrict__ mean, const float* __restrict__ invstd,
                                      const float* __restrict__ scale, const float* __restrict__ shift,
                                      float slope, float* __restrict__ partial, const int* __restrict__ count,
                                      int unit) {
  rows = live_rows(rows, count, unit);
  channel_reduce2(C, rows, partial, [&](long long r, int c, float& s0, float& s1) {
    const float yv = y[r * ld_y + c];
    const float u = fmaf(yv, scale[c], shift[c]);
    const float du = read_dz(gs, r, c) * (u > 0.f ? 1.f : slope);
    s0 += du; s1 = fmaf(du, (yv - mean[c]) * invstd[c], s1);
  });
}

// coef[0][c] = a, coef[1][c] = k1, coef[2][c] = k0 with dy = a*du + k1*y + k0
__global__ void bn_bwd_final_kernel(const float* __restrict__ partial, int nblk, long long rows, int C,
                                    const float* __restrict__ gamma, const float* __restrict__ mean,
                                    const float* __restrict__ invstd, int training, float* __restrict__ dgamma,
                                    float* __restrict__ dbeta, int accumulate, float* __restrict__ coef,
                                    const int* __restrict__ count, int unit) {
  const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);     // one wavefront per channel
  if (c >= C) return;
  if (count) { rows = live_rows(rows, count, unit); if (rows < 1) rows = 1; }
  double s, sx;
  wave_partial_sums(partial, nblk, C, c, s, sx);
  if ((threadIdx.x & 63) != 0) return;
  if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)sx;
  if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)s;
  const double g = gamma ? gamma[c] : 1.0, is = invstd[c], mu = mean[c];
  const double a = g * is;
  double k1 = 0.0, k0 = 0.0;
  if (training) {
    const double M = (double)rows;
    k1 = -a * sx * is / M;
    k0 = -a * s / M - k1 * mu;
  }
  coef[c] = (float)a; coef[C + c] = (float)k1; coef[2 * C + c] = (float)k0;
}

// the same finish for tile partials [2][C][nblk] out of a data-gradient epilogue / split-K finish (conv.hip)
__global__ void bn_bwd_final_tiles_kernel(const float* __restrict__ partial, int nblk, long long rows, int C,
                                          const float* __restrict__ gamma, const float* __restrict__ mean,
                                          const float* __restrict__ invstd, int training, float* __restrict__ dgamma,
                                          float* __restrict__ dbeta, int accumulate, float* __restrict__ coef,
                                          const int* __restrict__ count, int unit) {
  const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);     // one wavefront per channel
  if (c >= C) return;
  if (count) { rows = live_rows(rows, count, unit); if (rows < 1) rows = 1; }
  const int lane = threadIdx.x & 63;
  const float* __restrict__ q = partial + (size_t)c * nblk;
  const size_t plane = (size_t)C * nblk;
  double s = 0.0, sx = 0.0;
  if ((nblk & 3) == 0) {
    const float4* __restrict__ q4 = reinterpret_cast<const float4*>(q);
    const float4* __restrict__ x4 = reinterpret_cast<const float4*>(q + plane);
    for (int t4 = lane; t4 < (nblk >> 2); t4 += 64) {
      const float4 a = q4[t4], b = x4[t4];
      s += ((double)a.x + (double)a.y) + ((double)a.z + (double)a.w);
      sx += ((double)b.x + (double)b.y) + ((double)b.z + (double)b.w);
    }
  } else {
    for (int t = lane; t < nblk; t += 64) { s += (double)q[t]; sx += (double)q[plane + t]; }
  }
  #pragma unroll
  for (int off = 32; off > 0; off >>= 1) { s += __shfl_xor(s, off); sx += __shfl_xor(sx, off); }
  if (lane != 0) return;
  if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)sx;
  if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)s;
  const double g = gamma ? gamma[c] : 1.0, is = invstd[c], mu = mean[c];
  const double a = g * is;
  double k1 = 0.0, k0 = 0.0;
  if (training) {
    const double M = (double)rows;
    k1 = -a * sx * is / M;
    k0 = -a * s / M - k1 * mu;
  }
  coef[c] = (float)a; coef[C + c] = (float)k1; coef[2 * C + c] = (float)k0;
}

__global__ void bn_bwd_apply_kernel(GradSrc gs, const float* __restrict__ y, long long ld_y, long long rows,
                                    int C, const float* __restrict__ scale, const float* __restrict__ shift,
                                    float slope, const float* __restrict__ coef, float* __restrict__ dy,
                                    const int* __restrict__ count, int unit) {
  const long long total = rows * C;
  const long long live = live_rows(rows, count, unit);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / C; const int c = (int)(i - r * C);
    if (r >= live) { dy[i] = 0.f; continue; }        // padding row: no gradient
    const float yv = y[r * ld_y + c];
    const float u = fmaf(yv, scale[c], shift[c]);
    const float du = read_dz(gs, r, c) * (u > 0.f ? 1.f : slope);
    dy[i] = fmaf(coef[c], du, fmaf(coef[C + c], yv, coef[2 * C + c]));
  }
}

__global__ void act_bwd_kernel(GradSrc gs, const float* __restrict__ y, long long ld_y, long long rows, int C,
                               float slope, float* __restrict__ dx) {
  const long long total = rows * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / C; const int c = (int)(i - r * C);
    dx[i] = read_dz(gs, r, c) * (y[r * ld_y + c] > 0.f ? 1.f : slope);
  }
}

// out = leaky_slope(scale[c] * x + shift[c]): a normalisation + activation that has to be
// materialised (BatchNorm1d + ReLU of an MLP whose output feeds a gather / pooling, not a GEMM)
__global__ void affine_act_kernel(const float* __restrict__ x, long long ld_x, long long rows, int C,
                                  const float* __restrict__ scale, const float* __restrict__ shift, float slope,
                                  float* __restrict__ out, long long ld_out) {
  const long long total = rows * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / C; const int c = (int)(i - r * C);
    const float v = fmaf(x[r * ld_x + c], scale[c], shift[c]);
    out[r * ld_out + c] = v > 0.f ? v : v * slope;
  }
}

// float4 form (C, ld_x, ld_out multiples of 4, 16-byte aligned): the materialisation pass of the
// refinement network's activations when the direct-to-LDS convolution loop (igemm2.h) is on
__global__ void affine_act_v4_kernel(const float* __restrict__ x, long long ld_x, long long rows, int C,
                                     const float* __restrict__ scale, const float* __restrict__ shift, float slope,
                                     float* __restrict__ out, long long ld_out) {
  const int CQ = C >> 2;
  const long long total = rows * CQ;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / CQ; const int c = 4 * (int)(i - r * CQ);
    const float4 v = *reinterpret_cast<const float4*>(x + r * ld_x + c);
    const float4 sc = *reinterpret_cast<const float4*>(scale + c), sh = *reinterpret_cast<const float4*>(shift + c);
    float4 o;
    o.x = leakyf(fmaf(v.x, sc.x, sh.x), slope); o.y = leakyf(fmaf(v.y, sc.y, sh.y), slope);
    o.z = leakyf(fmaf(v.z, sc.z, sh.z), slope); o.w = leakyf(fmaf(v.w, sc.w, sh.w), slope);
    *reinterpret_cast<float4*>(out + r * ld_out + c) = o;
  }
}

// ---- InstanceNorm2d (affine=False, no running statistics; reference sg2im/layers.py:27-28) ----
// One workgroup owns the (image n, 64-channel block) slice of an NHWC tensor: kInRows row lanes x
// 64 channel lanes, partial sums combined through LDS in a fixed order (deterministic).
constexpr int kInRows = 8;

template <int NV>
__device__ inline void in_reduce(float (&v)[NV], float (*red)[kInRows][64], int rl, int cl) {
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; ++k) red[k][rl][cl] = v[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < kInRows; ++r) s += red[k][r][cl];
    v[k] = s;
  }
}

// scale[n][c] = 1/sqrt(var+eps), shift[n][c] = -mean*scale (biased variance over the HW pixels)
__global__ void __launch_bounds__(kInRows * 64)
instnorm_stats_kernel(const float* __restrict__ x, int HW, int C, float eps, float* __restrict__ scale,
                      float* __restrict__ shift) {
  __shared__ float red[1][kInRows][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl, n = blockIdx.y;
  const bool live = c < C;
  const float* xp = x + (long long)n * HW * C + (live ? c : 0);
  float a[1] = {0.f};
  if (live) for (int r = rl; r < HW; r += kInRows) a[0] += xp[(long long)r * C];
  in_reduce<1>(a, red, rl, cl);
  const float mean = a[0] / (float)HW;
  float q[1] = {0.f};
  if (live) for (int r = rl; r < HW; r += kInRows) { const float d = xp[(long long)r * C] - mean; q[0] = fmaf(d, d, q[0]); }
  in_reduce<1>(q, red, rl, cl);
  if (live && rl == 0) {
    const float is = 1.f / sqrtf(q[0] / (float)HW + eps);
    scale[(long long)n * C + c] = is;
    shift[(long long)n * C + c] = -mean * is;
  }
}

// out = leaky_slope(scale[n][c] * x + shift[n][c])
__global__ void instnorm_act_kernel(const float* __restrict__ x, long long total, int HW, int C,
                                    const float* __restrict__ scale, const float* __restrict__ shift, float slope,
                                    float* __restrict__ out) {
  const long long per = (long long)HW * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long n = i / per; const int c = (int)(i % C);
    const float v = fmaf(x[i], scale[n * C + c], shift[n * C + c]);
    out[i] = v > 0.f ? v : v * slope;
  }
}

// dyn = gradient w.r.t. the normalised value yn = scale*x + shift  ->  gradient w.r.t. x:
// dx = scale * (dyn - mean_hw(dyn) - yn * mean_hw(dyn * yn));  dx may alias dyn.
__global__ void __launch_bounds__(kInRows * 64)
instnorm_backward_kernel(const float* dyn, const float* __restrict__ x, int HW, int C,
                         const float* __restrict__ scale, const float* __restrict__ shift, float* dx) {
  __shared__ float red[2][kInRows][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl, n = blockIdx.y;
  const bool live = c < C;
  const long long base = (long long)n * HW * C + (live ? c : 0);
  const float sc = live ? scale[(long long)n * C + c] : 0.f, sh = live ? shift[(long long)n * C + c] : 0.f;
  float a[2] = {0.f, 0.f};
  if (live) for (int r = rl; r < HW; r += kInRows) {
    const float g = dyn[base + (long long)r * C], yn = fmaf(x[base + (long long)r * C], sc, sh);
    a[0] += g; a[1] = fmaf(g, yn, a[1]);
  }
  in_reduce<2>(a, red, rl, cl);
  const float m0 = a[0] / (float)HW, m1 = a[1] / (float)HW;
  if (live) for (int r = rl; r < HW; r += kInRows) {
    const long long o = base + (long long)r * C;
    const float yn = fmaf(x[o], sc, sh);
    dx[o] = sc * (dyn[o] - m0 - yn * m1);
  }
}

// ---- the spatial tokens of build_cnn architecture strings (reference sg2im/layers.py:184-196) ----
// out[b][y][x][c] = alpha * in[b][y/f][x/f][c]   (nn.Upsample nearest; backward of a sum / average pool)
__global__ void resample_up_kernel(const float* __restrict__ x, int B, int H, int W, int C, int f, int Ho, int Wo,
                                   float alpha, float* __restrict__ out) {
  const long long total = (long long)B * Ho * Wo * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C); long long t = i / C;
    const int xo = (int)(t % Wo); t /= Wo;
    const int yo = (int)(t % Ho); const long long n = t / Ho;
    const int yi = yo / f, xi = xo / f;
    out[i] = (yi < H && xi < W) ? alpha * x[((n * H + yi) * W + xi) * C + c] : 0.f;
  }
}

// out[b][yo][xo][c] = alpha * sum of the f x f window (floor(H/f) x floor(W/f) outputs: nn.AvgPool2d with
// alpha = 1/f^2; backward of the nearest upsample with alpha = 1)
__global__ void pool_sum_kernel(const float* __restrict__ x, int B, int H, int W, int C, int f, float alpha,
                                float* __restrict__ out) {
  const int Ho = H / f, Wo = W / f;
  const long long total = (long long)B * Ho * Wo * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C); long long t = i / C;
    const int xo = (int)(t % Wo); t /= Wo;
    const int yo = (int)(t % Ho); const long long n = t / Ho;
    float s = 0.f;
    for (int dy = 0; dy < f; ++dy)
      for (int dx = 0; dx < f; ++dx)
        s += x[((n * H + yo * f + dy) * W + xo * f + dx) * C + c];
    out[i] = alpha * s;
  }
}

// nn.MaxPool2d(f, f): the FIRST maximum of the row-major window scan wins (ATen's `val > max`)
__device__ inline int maxpool_argmax(const float* __restrict__ x, long long n, int H, int W, int C, int f, int yo,
                                     int xo, int c, float& best) {
  int arg = 0; best = x[((n * H + yo * f) * W + xo * f) * C + c];
  for (int k = 1; k < f * f; ++k) {
    const int dy = k / f, dx = k - dy * f;
    const float v = x[((n * H + yo * f + dy) * W + xo * f + dx) * C + c];
    if (v > best || v != v) { best = v; arg = k; }
  }
  return arg;
}

__global__ void maxpool_fwd_kernel(const float* __restrict__ x, int B, int H, int W, int C, int f,
                                   float* __restrict__ out) {
  const int Ho = H / f, Wo = W / f;
  const long long total = (long long)B * Ho * Wo * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C); long long t = i / C;
    const int xo = (int)(t % Wo); t /= Wo;
    const int yo = (int)(t % Ho); const long long n = t / Ho;
    float best;
    maxpool_argmax(x, n, H, W, C, f, yo, xo, c, best);
    out[i] = best;
  }
}

// dx has the input's shape: the window's gradient goes to its arg-max, every other position gets 0
__global__ void maxpool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, int B, int H, int W,
                                   int C, int f, float* __restrict__ dx) {
  const int Ho = H / f, Wo = W / f;
  const long long total = (long long)B * H * W * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C); long long t = i / C;
    const int xi = (int)(t % W); t /= W;
    const int yi = (int)(t % H); const long long n = t / H;
    const int yo = yi / f, xo = xi / f;
    float g = 0.f;
    if (yo < Ho && xo < Wo) {
      float best;
      const int arg = maxpool_argmax(x, n, H, W, C, f, yo, xo, c, best);
      if (arg == (yi - yo * f) * f + (xi - xo * f)) g = dy[((n * Ho + yo) * Wo + xo) * C + c];
    }
    dx[i] = g;
  }
}

// out = leaky_slope(x) and out = a + b (the residual sum of ResidualBlock, sg2im/layers.py:112-117)
__global__ void leaky_fwd_kernel(const float* __restrict__ x, long long n, float slope, float* __restrict__ out) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float v = x[i];
    out[i] = v > 0.f ? v : v * slope;
  }
}

__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, long long n,
                           float* __restrict__ out) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    out[i] = a[i] + b[i];
}

__global__ void avgpool_kernel(const float* __restrict__ x, int B, int H, int W, int C, int f,
                               float* __restrict__ out) {
  const int Ho = H / f, Wo = W / f;
  const long long total = (long long)B * Ho * Wo * C;
  const float inv = 1.f / (float)(f * f);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C); long long t = i / C;
    const int xo = (int)(t % Wo); t /= Wo;
    const int yo = (int)(t % Ho); const long long n = t / Ho;
    float s = 0.f;
    for (int dy = 0; dy < f; ++dy)
      for (int dx = 0; dx < f; ++dx)
        s += x[((n * H + yo * f + dy) * W + xo * f + dx) * C + c];
    out[i] = s * inv;
  }
}

struct PyramidArgs { const float* lvl[8]; long long ld[8]; int f[8]; int n; };
__global__ void pyramid_bwd_kernel(PyramidArgs a, int B, int H, int W, int C, float* __restrict__ out,
                                   long long ld_out) {
  const long long total = (long long)B * H * W * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C); long long t = i / C;
    const int x = (int)(t % W); t /= W;
    const int y = (int)(t % H); const long long n = t / H;
    float s = 0.f;
    #pragma unroll
    for (int l = 0; l < 8; ++l) {
      if (l < a.n) {
        const int f = a.f[l];
        const int h = H / f, w = W / f;
        s += a.lvl[l][((n * h + y / f) * w + x / f) * a.ld[l] + c] * (1.f / (float)(f * f));
      }
    }
    out[((n * H + y) * W + x) * ld_out + c] = s;
  }
}

__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, int B, int C, int H, int W,
                                    float* __restrict__ dst, long long ld, int coff) {
  const long long total = (long long)B * C * H * W;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C); long long t = i / C;     // enumerate in NHWC order (coalesced writes)
    const int x = (int)(t % W); t /= W;
    const int y = (int)(t % H); const long long n = t / H;
    dst[((n * H + y) * W + x) * ld + coff + c] = src[((n * C + c) * H + y) * W + x];
  }
}

__global__ void nhwc_to_nchw_kernel(const float* __restrict__ src, long long ld, int coff, int B, int C,
                                    int H, int W, float* __restrict__ dst) {
  const long long total = (long long)B * C * H * W;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % W); long long t = i / W;     // enumerate in NCHW order (coalesced writes)
    const int y = (int)(t % H); t /= H;
    const int c = (int)(t % C); const long long n = t / C;
    dst[i] = src[((n * H + y) * W + x) * ld + coff + c];
  }
}

__global__ void gap_fwd_kernel(const float* __restrict__ x, int B, int HW, int C, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  const int n = i / C, c = i - n * C;
  float s = 0.f;
  for (int p = 0; p < HW; ++p) s += x[((long long)n * HW + p) * C + c];
  out[i] = s / (float)HW;
}

__global__ void gap_bwd_kernel(const float* __restrict__ dout, int B, int HW, int C, float* __restrict__ dx) {
  const long long total = (long long)B * HW * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C); const long long n = i / ((long long)HW * C);
    dx[i] = dout[n * C + c] / (float)HW;
  }
}

__global__ void sigmoid_fwd_kernel(const float* __restrict__ x, long long n, float* __restrict__ y) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    y[i] = 1.f / (1.f + expf(-x[i]));
}

__global__ void sigmoid_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, long long n,
                                   float* __restrict__ dx) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float s = y[i];
    dx[i] = dy[i] * s * (1.f - s);
  }
}

__global__ void colsum_partial_kernel(const float* __restrict__ x, long long rows, int C, long long ld,
                                      float* __restrict__ partial) {
  channel_reduce2(C, rows, partial, [&](long long r, int c, float& s0, float& s1) { s0 += x[r * ld + c]; });
}

__global__ void colsum_final_kernel(const float* __restrict__ partial, int nblk, int C, float* __restrict__ out,
                                    int accumulate) {
  const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);     // one wavefront per channel
  if (c >= C) return;
  double s, unused;
  wave_partial_sums(partial, nblk, C, c, s, unused);
  if ((threadIdx.x & 63) == 0) out[c] = (accumulate ? out[c] : 0.f) + (float)s;
}

// ---------------------------------------------------------------------------
// 16-byte variants (C % 4 == 0, 16-byte aligned rows): one thread owns a channel QUAD,
// 4x fewer memory instructions and 4x the bytes in flight per lane - these passes are
// HBM-bound and the scalar forms above reach only ~25 % of the bandwidth.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

__device__ __forceinline__ float4 read_dz4(const GradSrc& s, long long r, int c) {
  if (!s.pool2) return ld4(s.g + r * s.ld + c);
  const long long hw = (long long)s.h * s.w;
  const long long n = r / hw; const int rem = (int)(r - n * hw);
  const int y = rem / s.w, x = rem - y * s.w;
  const long long W2 = 2LL * s.w;
  const float* p = s.g + ((n * 2 * s.h + 2 * y) * W2 + 2 * x) * s.ld + c;
  return f4add(f4add(ld4(p), ld4(p + s.ld)), f4add(ld4(p + W2 * s.ld), ld4(p + (W2 + 1) * s.ld)));
}

// f(row, c, s0, s1) accumulates float4 quads; partial layout identical to channel_reduce2
template <typename F>
__device__ __forceinline__ void channel_reduce2_v4(int C, long long rows, float* partial, F f) {
  extern __shared__ float red[];                       // 2 * 4 * blockDim.x floats
  const int CQ = C >> 2;
  const int TQ = CQ < (int)blockDim.x ? CQ : (int)blockDim.x;
  const int TR = blockDim.x / TQ;
  const int tx = threadIdx.x % TQ, ty = threadIdx.x / TQ;
  const long long per = (rows + gridDim.x - 1) / gridDim.x;
  const long long r0 = (long long)blockIdx.x * per;
  const long long r1 = r0 + per < rows ? r0 + per : rows;
  for (int q = tx; q < CQ; q += TQ) {
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    if (ty < TR)
      for (long long r = r0 + ty; r < r1; r += TR) f(r, 4 * q, s0, s1);
    if (TR > 1) {
      float4* red4 = reinterpret_cast<float4*>(red);
      red4[threadIdx.x] = s0; red4[blockDim.x + threadIdx.x] = s1;
      __syncthreads();
      if (ty == 0) {
        for (int t = 1; t < TR; ++t) { s0 = f4add(s0, red4[t * TQ + tx]); s1 = f4add(s1, red4[blockDim.x + t * TQ + tx]); }
      }
      __syncthreads();
    }
    if (ty == 0) {
      *reinterpret_cast<float4*>(partial + ((long long)blockIdx.x * 2 + 0) * C + 4 * q) = s0;
      *reinterpret_cast<float4*>(partial + ((long long)blockIdx.x * 2 + 1) * C + 4 * q) = s1;
    }
  }
}

__global__ void bn_stats_partial_v4_kernel(const float* __restrict__ x, long long rows, int C, long long ld,
                                           float* __restrict__ partial, const int* __restrict__ count, int unit) {
  rows = live_rows(rows, count, unit);
  channel_reduce2_v4(C, rows, partial, [&](long long r, int c, float4& s0, float4& s1) {
    const float4 t = ld4(x + r * ld + c), p = ld4(x + c);          // (pivot = row 0, see the scalar form)
    const float4 v = make_float4(t.x - p.x, t.y - p.y, t.z - p.z, t.w - p.w);
    s0 = f4add(s0, v);
    s1.x = fmaf(v.x, v.x, s1.x); s1.y = fmaf(v.y, v.y, s1.y); s1.z = fmaf(v.z, v.z, s1.z); s1.w = fmaf(v.w, v.w, s1.w);
  });
}

__global__ void colsum_partial_v4_kernel(const float* __restrict__ x, long long rows, int C, long long ld,
                                         float* __restrict__ partial) {
  channel_reduce2_v4(C, rows, partial, [&](long long r, int c, float4& s0, float4& s1) { s0 = f4add(s0, ld4(x + r * ld + c)); });
}

__device__ __forceinline__ float du1(float dz, float yv, float sc, float sh, float slope) {
  return dz * (fmaf(yv, sc, sh) > 0.f ? 1.f : slope);
}

__global__ void bn_bwd_partial_v4_kernel(GradSrc gs, const float* __restrict__ y, long long ld_y, long long rows,
                                         int C, const float* __restrict__ mean, const float* __restrict__ invstd,
                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                         float slope, float* __restrict__ partial, const int* __restrict__ count,
                                         int unit) {
  rows = live_rows(rows, count, unit);
  channel_reduce2_v4(C, rows, partial, [&](long long r, int c, float4& s0, float4& s1) {
    const float4 yv = ld4(y + r * ld_y + c), dz = read_dz4(gs, r, c);
    const float4 sc = ld4(scale + c), sh = ld4(shift + c), mu = ld4(mean + c), is = ld4(invstd + c);
    const float4 du = make_float4(du1(dz.x, yv.x, sc.x, sh.x, slope), du1(dz.y, yv.y, sc.y, sh.y, slope),
                                  du1(dz.z, yv.z, sc.z, sh.z, slope), du1(dz.w, yv.w, sc.w, sh.w, slope));
    s0 = f4add(s0, du);
    s1.x = fmaf(du.x, (yv.x - mu.x) * is.x, s1.x); s1.y = fmaf(du.y, (yv.y - mu.y) * is.y, s1.y);
    s1.z = fmaf(du.z, (yv.z - mu.z) * is.z, s1.z); s1.w = fmaf(du.w, (yv.w - mu.w) * is.w, s1.w);
  });
}

__global__ void bn_bwd_apply_v4_kernel(GradSrc gs, const float* __restrict__ y, long long ld_y, long long rows,
                                       int C, const float* __restrict__ scale, const float* __restrict__ shift,
                                       float slope, const float* __restrict__ coef, float* __restrict__ dy,
                                       const int* __restrict__ count, int unit) {
  const int CQ = C >> 2;
  const long long total = rows * CQ;
  const long long live = live_rows(rows, count, unit);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / CQ; const int c = 4 * (int)(i - r * CQ);
    if (r >= live) { *reinterpret_cast<float4*>(dy + r * C + c) = make_float4(0.f, 0.f, 0.f, 0.f); continue; }
    const float4 yv = ld4(y + r * ld_y + c), dz = read_dz4(gs, r, c);
    const float4 sc = ld4(scale + c), sh = ld4(shift + c);
    const float4 a = ld4(coef + c), k1 = ld4(coef + C + c), k0 = ld4(coef + 2 * C + c);
    float4 o;
    o.x = fmaf(a.x, du1(dz.x, yv.x, sc.x, sh.x, slope), fmaf(k1.x, yv.x, k0.x));
    o.y = fmaf(a.y, du1(dz.y, yv.y, sc.y, sh.y, slope), fmaf(k1.y, yv.y, k0.y));
    o.z = fmaf(a.z, du1(dz.z, yv.z, sc.z, sh.z, slope), fmaf(k1.z, yv.z, k0.z));
    o.w = fmaf(a.w, du1(dz.w, yv.w, sc.w, sh.w, slope), fmaf(k1.w, yv.w, k0.w));
    *reinterpret_cast<float4*>(dy + r * C + c) = o;
  }
}

// the same with bfloat16 STORAGE of any of the three tensors (sg2im_bn_backward_apply_ex): g (dz), y, dy hold fp32 or
// bf16 independently; the arithmetic is fp32, the result is rounded (RNE) when dy holds bf16
__device__ __forceinline__ float4 ld4x(const void* base, long long elem, int bf) {
  if (!bf) return ld4(reinterpret_cast<const float*>(base) + elem);
  const uint2 r = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + elem);
  return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u),
                     __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u));
}
__global__ void bn_bwd_apply_x_kernel(const void* __restrict__ g, long long ld_g, int pool2, int h, int w, int g_bf,
                                      const void* __restrict__ y, long long ld_y, int y_bf, long long rows, int C,
                                      const float* __restrict__ scale, const float* __restrict__ shift, float slope,
                                      const float* __restrict__ coef, void* __restrict__ dy, int dy_bf) {
  const int CQ = C >> 2;
  const long long total = rows * CQ;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / CQ; const int c = 4 * (int)(i - r * CQ);
    const float4 yv = ld4x(y, r * ld_y + c, y_bf);
    float4 dz;
    if (!pool2) dz = ld4x(g, r * ld_g + c, g_bf);
    else {
      const long long hw = (long long)h * w;
      const long long n = r / hw; const int rem = (int)(r - n * hw);
      const int yy = rem / w, xx = rem - yy * w;
      const long long W2 = 2LL * w;
      const long long e0 = ((n * 2 * h + 2 * yy) * W2 + 2 * xx) * ld_g + c;
      dz = f4add(f4add(ld4x(g, e0, g_bf), ld4x(g, e0 + ld_g, g_bf)), f4add(ld4x(g, e0 + W2 * ld_g, g_bf), ld4x(g, e0 + (W2 + 1) * ld_g, g_bf)));
    }
    const float4 sc = ld4(scale + c), sh = ld4(shift + c);
    const float4 a = ld4(coef + c), k1 = ld4(coef + C + c), k0 = ld4(coef + 2 * C + c);
    float4 o;
    o.x = fmaf(a.x, du1(dz.x, yv.x, sc.x, sh.x, slope), fmaf(k1.x, yv.x, k0.x));
    o.y = fmaf(a.y, du1(dz.y, yv.y, sc.y, sh.y, slope), fmaf(k1.y, yv.y, k0.y));
    o.z = fmaf(a.z, du1(dz.z, yv.z, sc.z, sh.z, slope), fmaf(k1.z, yv.z, k0.z));
    o.w = fmaf(a.w, du1(dz.w, yv.w, sc.w, sh.w, slope), fmaf(k1.w, yv.w, k0.w));
    if (!dy_bf) *reinterpret_cast<float4*>(reinterpret_cast<float*>(dy) + r * C + c) = o;
    else {
      typedef __bf16 bfx4 __attribute__((ext_vector_type(4)));
      typedef float fx4 __attribute__((ext_vector_type(4)));
      const fx4 f = {o.x, o.y, o.z, o.w};
      *reinterpret_cast<bfx4*>(reinterpret_cast<__bf16*>(dy) + r * C + c) = __builtin_convertvector(f, bfx4);
    }
  }
}

__global__ void act_bwd_v4_kernel(GradSrc gs, const float* __restrict__ y, long long ld_y, long long rows, int C,
                                  float slope, float* __restrict__ dx) {
  const int CQ = C >> 2;
  const long long total = rows * CQ;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / CQ; const int c = 4 * (int)(i - r * CQ);
    const float4 yv = ld4(y + r * ld_y + c), dz = read_dz4(gs, r, c);
    float4 o;
    o.x = dz.x * (yv.x > 0.f ? 1.f : slope); o.y = dz.y * (yv.y > 0.f ? 1.f : slope);
    o.z = dz.z * (yv.z > 0.f ? 1.f : slope); o.w = dz.w * (yv.w > 0.f ? 1.f : slope);
    *reinterpret_cast<float4*>(dx + r * C + c) = o;
  }
}

__global__ void avgpool_v4_kernel(const float* __restrict__ x, int B, int H, int W, int C, int f,
                                  float* __restrict__ out) {
  const int Ho = H / f, Wo = W / f, CQ = C >> 2;
  const long long total = (long long)B * Ho * Wo * CQ;
  const float inv = 1.f / (float)(f * f);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = 4 * (int)(i % CQ); long long t = i / CQ;
    const int xo = (int)(t % Wo); t /= Wo;
    const int yo = (int)(t % Ho); const long long n = t / Ho;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int dy = 0; dy < f; ++dy)
      for (int dx = 0; dx < f; ++dx)
        s = f4add(s, ld4(x + ((n * H + yo * f + dy) * W + xo * f + dx) * C + c));
    *reinterpret_cast<float4*>(out + 4 * i) = make_float4(s.x * inv, s.y * inv, s.z * inv, s.w * inv);
  }
}

// One image row (n, y) per workgroup: the per-level row bases, factors and weights are wave-uniform
// (scalar), an item only divides its x by the level's factor - a shift when the factor is a power of
// two.  (The grid-stride form decoded (n, y, x, c) and y / f, x / f, H / f, W / f with ~15 integer
// divisions per float4: 99 us for the 5-level 64x64x128 batch-32 pyramid against ~20 us of traffic.)
__global__ void pyramid_bwd_v4_kernel(PyramidArgs a, int B, int H, int W, int C, float* __restrict__ out,
                                      long long ld_out) {
  const int CQ = C >> 2;
  const int n = blockIdx.x / H, y = blockIdx.x - n * H;
  const bool cq_pow2 = (CQ & (CQ - 1)) == 0;
  const int cq_shift = __builtin_ctz(CQ);
  float* const orow = out + ((long long)n * H + y) * W * ld_out;
  for (int j = threadIdx.x; j < W * CQ; j += blockDim.x) {
    const int x = cq_pow2 ? j >> cq_shift : j / CQ;
    const int c = 4 * (j - x * CQ);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    #pragma unroll
    for (int l = 0; l < 8; ++l) {
      if (l < a.n) {
        const int f = a.f[l];
        const int h = H / f, w = W / f;
        const int xl = (f & (f - 1)) == 0 ? x >> __builtin_ctz(f) : x / f;
        const float4 v = ld4(a.lvl[l] + (((long long)n * h + y / f) * w + xl) * a.ld[l] + c);
        const float k = 1.f / (float)(f * f);
        s.x += v.x * k; s.y += v.y * k; s.z += v.z * k; s.w += v.w * k;
      }
    }
    *reinterpret_cast<float4*>(orow + (long long)x * ld_out + c) = s;
  }
}

static inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

static inline int red_blocks(long long rows) {
  return (int)std::max<long long>(1, std::min<long long>(RED_BLOCKS, (rows + 31) / 32));
}
static inline int ew_blocks(long long total) {
  return (int)std::max<long long>(1, std::min<long long>((total + 255) / 256, 8192));
}
static inline int ok_or(hipError_t e) { return e == hipSuccess ? SG2IM_OK : SG2IM_ERR_HIP; }

// sg2im_debug_mark_gemm_end: a caller-supplied event recorded between the GEMM launches of a *_bn entry point and
// its BatchNorm finish launch (per host thread), so that a timer can attribute the two parts separately
static thread_local hipEvent_t t_gemm_end_event = nullptr;
static thread_local int* t_gemm_end_flag = nullptr;
static void mark_gemm_end(hipStream_t stream) {
  if (!t_gemm_end_event) return;
  if (hipEventRecord(t_gemm_end_event, stream) == hipSuccess && t_gemm_end_flag) *t_gemm_end_flag = 1;
}

int bn_stats_finish_tiles(const float* partial, int nblk, long long per, long long rows, int channels,
                          const sg2im_bn_fwd* a, hipStream_t stream) {
  mark_gemm_end(stream);
  SG2IM_LAUNCH(bn_stats_final_tiles_kernel, dim3((channels + 3) / 4), dim3(256), 0, stream, partial, nblk, per, rows,
                     a->unbiased_rows, channels, a->gamma, a->beta, a->eps, a->momentum, a->training > 1 ? a->training : 1,
                     a->running_mean, a->running_var, a->num_batches_tracked, a->mean, a->invstd, a->scale, a->shift, a->count, a->count_unit);
  return ok_or(hipGetLastError());
}

int bn_bwd_finish_tiles(const float* partial, int nblk, long long rows, int channels, const sg2im_bn_bwd* a,
                        hipStream_t stream) {
  mark_gemm_end(stream);
  SG2IM_LAUNCH(bn_bwd_final_tiles_kernel, dim3((channels + 3) / 4), dim3(256), 0, stream, partial, nblk, rows, channels,
                     a->gamma, a->mean, a->invstd, a->training, a->dgamma, a->dbeta, a->accumulate, a->coef, a->count,
                     a->count_unit);
  return ok_or(hipGetLastError());
}

int bn_bwd_standalone(const float* g, long long ld_g, int pool2, int batch, int h, int w, int channels,
                      const sg2im_bn_bwd* a, hipStream_t stream) {
  const long long rows = (long long)batch * h * w;
  if (rows < 1 || !a->partial) return SG2IM_ERR_ARG;
  const GradSrc gs{g, ld_g, pool2, h, w};
  int nblk = red_blocks(rows);
  // (the caller's partial buffer may be smaller than the usual 2 * C * RED_BLOCKS floats)
  if ((size_t)nblk * 2 * channels > a->partial_floats) nblk = (int)std::max<size_t>(1, a->partial_floats / ((size_t)2 * channels));
  const bool v4 = channels % 4 == 0 && ld_g % 4 == 0 && a->ld_y % 4 == 0 && al16(g) && al16(a->y) &&
                  al16(a->partial) && al16(a->mean) && al16(a->invstd) && al16(a->scale) && al16(a->shift);
  if (v4)
    SG2IM_LAUNCH(bn_bwd_partial_v4_kernel, dim3(nblk), dim3(256), 8 * 256 * sizeof(float), stream, gs, a->y, a->ld_y,
                       rows, channels, a->mean, a->invstd, a->scale, a->shift, a->slope, a->partial, a->count, a->count_unit);
  else
    SG2IM_LAUNCH(bn_bwd_partial_kernel, dim3(nblk), dim3(256), 2 * 256 * sizeof(float), stream, gs, a->y, a->ld_y, rows,
                       channels, a->mean, a->invstd, a->scale, a->shift, a->slope, a->partial, a->count, a->count_unit);
  if (hipGetLastError() != hipSuccess) return SG2IM_ERR_HIP;
  SG2IM_LAUNCH(bn_bwd_final_kernel, dim3((channels + 3) / 4), dim3(256), 0, stream, a->partial, nblk, rows, channels,
                     a->gamma, a->mean, a->invstd, a->training, a->dgamma, a->dbeta, a->accumulate, a->coef, a->count,
                     a->count_unit);
  return ok_or(hipGetLastError());
}


// fp32 -> bfloat16 (RNE), 8 elements per thread: the weight mirror of sg2im_conv_desc.weight_bf16
typedef __bf16 cast_bf16x8 __attribute__((ext_vector_type(8)));
typedef float cast_f32x8 __attribute__((ext_vector_type(8)));
__global__ void cast_f32_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, long long n8, long long n) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
    const float4 a = reinterpret_cast<const float4*>(src)[2 * i], b = reinterpret_cast<const float4*>(src)[2 * i + 1];
    const cast_f32x8 f = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    reinterpret_cast<cast_bf16x8*>(dst)[i] = __builtin_convertvector(f, cast_bf16x8);
  }
  if (blockIdx.x == 0 && threadIdx.x < (unsigned)(n - 8 * n8)) dst[8 * n8 + threadIdx.x] = (__bf16)src[8 * n8 + threadIdx.x];
}

}  // namespace sg2im

using namespace sg2im;

extern "C" {

int sg2im_bn_stats(const float* x, long long rows, int channels, long long ld, const float* gamma,
                   const float* beta, float eps, float momentum, int training, float* running_mean,
                   float* running_var, long long* num_batches_tracked, long long unbiased_rows,
                   float* mean, float* invstd, float* scale, float* shift, float* partial,
                   const int* count, int count_unit, hipStream_t stream) {
  if (channels < 1 || rows < 1 || !mean || !invstd || !scale || !shift) return SG2IM_ERR_ARG;
  if (training && (!x || !partial)) return SG2IM_ERR_ARG;
  if (!training && (!running_mean || !running_var)) return SG2IM_ERR_ARG;
  int nblk = 0;
  if (training) {
    nblk = red_blocks(rows);
    if (channels % 4 == 0 && ld % 4 == 0 && al16(x) && al16(partial))
      SG2IM_LAUNCH(bn_stats_partial_v4_kernel, dim3(nblk), dim3(256), 8 * 256 * sizeof(float), stream, x, rows,
                         channels, ld, partial, count, count_unit);
    else
      SG2IM_LAUNCH(bn_stats_partial_kernel, dim3(nblk), dim3(256), 2 * 256 * sizeof(float), stream, x, rows,
                         channels, ld, partial, count, count_unit);
  }
  SG2IM_LAUNCH(bn_stats_final_kernel, dim3((channels + 3) / 4), dim3(256), 0, stream, x, partial, nblk, rows,
                     unbiased_rows, channels, gamma, beta, eps, momentum, training, running_mean, running_var,
                     num_batches_tracked, mean, invstd, scale, shift, count, count_unit);
  return ok_or(hipGetLastError());
}

int sg2im_bn_act_backward(const float* g, long long ld_g, int pool2, int batch, int h, int w,
                          const float* y, long long ld_y, int channels, const float* gamma,
                          const float* mean, const float* invstd, const float* scale,
                          const float* shift, float slope, int training, float* dy,
                          float* dgamma, float* dbeta, int accumulate, float* partial,
                          const int* count, int count_unit, hipStream_t stream) {
  if (!g || !y || !dy || !partial || channels < 1 || !mean || !invstd || !scale || !shift) return SG2IM_ERR_ARG;
  if ((long long)batch * h * w < 1) return SG2IM_ERR_ARG;
  sg2im_bn_bwd a;
  a.y = y; a.ld_y = ld_y; a.pool2 = pool2; a.gamma = gamma; a.mean = mean; a.invstd = invstd; a.scale = scale;
  a.shift = shift; a.slope = slope; a.training = training; a.dgamma = dgamma; a.dbeta = dbeta; a.accumulate = accumulate;
  a.coef = partial + (size_t)2 * channels * RED_BLOCKS;     // 3*C floats behind the partials
  a.partial = partial; a.partial_floats = (size_t)2 * channels * RED_BLOCKS; a.count = count; a.count_unit = count_unit;
  const int rc = bn_bwd_standalone(g, ld_g, pool2, batch, h, w, channels, &a, stream);
  if (rc != SG2IM_OK) return rc;
  return sg2im_bn_backward_apply(g, ld_g, pool2, batch, h, w, y, ld_y, channels, scale, shift, slope, a.coef, dy, count,
                                 count_unit, stream);
}

int sg2im_debug_mark_gemm_end(hipEvent_t event, int* recorded) {
  sg2im::t_gemm_end_event = event;
  sg2im::t_gemm_end_flag = event ? recorded : nullptr;
  return SG2IM_OK;
}

int sg2im_bn_backward_apply(const float* g, long long ld_g, int pool2, int batch, int h, int w, const float* y,
                            long long ld_y, int channels, const float* scale, const float* shift, float slope,
                            const float* coef, float* dy, const int* count, int count_unit, hipStream_t stream) {
  if (!g || !y || !dy || !coef || channels < 1 || !scale || !shift) return SG2IM_ERR_ARG;
  const long long rows = (long long)batch * h * w;
  if (rows < 1) return SG2IM_ERR_ARG;
  const GradSrc gs{g, ld_g, pool2, h, w};
  const bool v4 = channels % 4 == 0 && ld_g % 4 == 0 && ld_y % 4 == 0 && al16(g) && al16(y) && al16(dy) &&
                  al16(scale) && al16(shift) && al16(coef);
  if (v4)
    SG2IM_LAUNCH(bn_bwd_apply_v4_kernel, dim3(ew_blocks(rows * channels / 4)), dim3(256), 0, stream, gs, y, ld_y,
                       rows, channels, scale, shift, slope, coef, dy, count, count_unit);
  else
    SG2IM_LAUNCH(bn_bwd_apply_kernel, dim3(ew_blocks(rows * channels)), dim3(256), 0, stream, gs, y, ld_y, rows,
                       channels, scale, shift, slope, coef, dy, count, count_unit);
  return ok_or(hipGetLastError());
}

int sg2im_bn_backward_apply_ex(const void* g, long long ld_g, int pool2, int batch, int h, int w, const void* y,
                               long long ld_y, int channels, const float* scale, const float* shift, float slope,
                               const float* coef, void* dy, int g_dtype, int y_dtype, int dy_dtype, hipStream_t stream) {
  if (!g || !y || !dy || !coef || channels < 1 || !scale || !shift) return SG2IM_ERR_ARG;
  if ((g_dtype | y_dtype | dy_dtype) & ~1) return SG2IM_ERR_ARG;
  const long long rows = (long long)batch * h * w;
  if (rows < 1) return SG2IM_ERR_ARG;
  if (channels % 4 || ld_g % 4 || ld_y % 4 || ((uintptr_t)g & 15) || ((uintptr_t)y & 15) || ((uintptr_t)dy & 15) || !al16(scale) ||
      !al16(shift) || !al16(coef))
    return SG2IM_ERR_ARG;
  SG2IM_LAUNCH(bn_bwd_apply_x_kernel, dim3(ew_blocks(rows * channels / 4)), dim3(256), 0, stream, g, ld_g, pool2, h, w, g_dtype,
                     y, ld_y, y_dtype, rows, channels, scale, shift, slope, coef, dy, dy_dtype);
  return ok_or(hipGetLastError());
}

int sg2im_act_backward(const float* g, long long ld_g, int pool2, int batch, int h, int w,
                       const float* y, long long ld_y, int channels, float slope, float* dx,
                       hipStream_t stream) {
  if (!g || !y || !dx || channels < 1) return SG2IM_ERR_ARG;
  const long long rows = (long long)batch * h * w;
  if (rows == 0) return SG2IM_OK;
  const GradSrc gs{g, ld_g, pool2, h, w};
  if (channels % 4 == 0 && ld_g % 4 == 0 && ld_y % 4 == 0 && al16(g) && al16(y) && al16(dx))
    SG2IM_LAUNCH(act_bwd_v4_kernel, dim3(ew_blocks(rows * channels / 4)), dim3(256), 0, stream, gs, y, ld_y, rows,
                       channels, slope, dx);
  else
    SG2IM_LAUNCH(act_bwd_kernel, dim3(ew_blocks(rows * channels)), dim3(256), 0, stream, gs, y, ld_y, rows,
                       channels, slope, dx);
  return ok_or(hipGetLastError());
}

int sg2im_affine_act_forward(const float* x, long long ld_x, long long rows, int channels, const float* scale,
                             const float* shift, float slope, float* out, long long ld_out,
                             hipStream_t stream) {
  if (!x || !scale || !shift || !out || channels < 1 || rows < 0) return SG2IM_ERR_ARG;
  if (rows == 0) return SG2IM_OK;
  if (channels % 4 == 0 && ld_x % 4 == 0 && ld_out % 4 == 0 && al16(x) && al16(out) && al16(scale) && al16(shift))
    SG2IM_LAUNCH(affine_act_v4_kernel, dim3(ew_blocks(rows * channels / 4)), dim3(256), 0, stream, x, ld_x, rows,
                       channels, scale, shift, slope, out, ld_out);
  else
    SG2IM_LAUNCH(affine_act_kernel, dim3(ew_blocks(rows * channels)), dim3(256), 0, stream, x, ld_x, rows, channels,
                       scale, shift, slope, out, ld_out);
  return ok_or(hipGetLastError());
}

int sg2im_cast_f32_to_bf16(const float* src, void* dst, long long n, hipStream_t stream) {
  if (!src || !dst || n < 0 || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return SG2IM_ERR_ARG;
  if (n == 0) return SG2IM_OK;
  SG2IM_LAUNCH(cast_f32_bf16_kernel, dim3(ew_blocks(std::max<long long>(1, n / 8))), dim3(256), 0, stream, src, (__bf16*)dst,
                     n / 8, n);
  return ok_or(hipGetLastError());
}

int sg2im_instnorm_stats(const float* x, int batch, int hw, int channels, float eps, float* scale, float* shift,
                         hipStream_t stream) {
  if (!x || !scale || !shift || batch < 0 || hw < 1 || channels < 1) return SG2IM_ERR_ARG;
  if (batch == 0) return SG2IM_OK;
  SG2IM_LAUNCH(instnorm_stats_kernel, dim3((channels + 63) / 64, batch), dim3(kInRows * 64), 0, stream, x, hw,
                     channels, eps, scale, shift);
  return ok_or(hipGetLastError());
}

int sg2im_instnorm_act_forward(const float* x, int batch, int hw, int channels, const float* scale,
                               const float* shift, float slope, float* out, hipStream_t stream) {
  if (!x || !scale || !shift || !out || batch < 0 || hw < 1 || channels < 1) return SG2IM_ERR_ARG;
  const long long total = (long long)batch * hw * channels;
  if (total == 0) return SG2IM_OK;
  SG2IM_LAUNCH(instnorm_act_kernel, dim3(ew_blocks(total)), dim3(256), 0, stream, x, total, hw, channels, scale,
                     shift, slope, out);
  return ok_or(hipGetLastError());
}

int sg2im_instnorm_backward(const float* dyn, const float* x, int batch, int hw, int channels, const float* scale,
                            const float* shift, float* dx, hipStream_t stream) {
  if (!dyn || !x || !scale || !shift || !dx || batch < 0 || hw < 1 || channels < 1) return SG2IM_ERR_ARG;
  if (batch == 0) return SG2IM_OK;
  SG2IM_LAUNCH(instnorm_backward_kernel, dim3((channels + 63) / 64, batch), dim3(kInRows * 64), 0, stream, dyn,
                     x, hw, channels, scale, shift, dx);
  return ok_or(hipGetLastError());
}

int sg2im_resample_nearest_up(const float* x, int batch, int h, int w, int channels, int factor, int out_h,
                              int out_w, float alpha, float* out, hipStream_t stream) {
  if (!x || !out || factor < 1 || out_h < 0 || out_w < 0 || channels < 1) return SG2IM_ERR_ARG;
  const long long total = (long long)batch * out_h * out_w * channels;
  if (total == 0) return SG2IM_OK;
  SG2IM_LAUNCH(resample_up_kernel, dim3(ew_blocks(total)), dim3(256), 0, stream, x, batch, h, w, channels, factor,
                     out_h, out_w, alpha, out);
  return ok_or(hipGetLastError());
}

int sg2im_pool_sum_forward(const float* x, int batch, int h, int w, int channels, int factor, float alpha,
                           float* out, hipStream_t stream) {
  if (!x || !out || factor < 1 || channels < 1) return SG2IM_ERR_ARG;
  const long long total = (long long)batch * (h / factor) * (w / factor) * channels;
  if (total == 0) return SG2IM_OK;
  SG2IM_LAUNCH(pool_sum_kernel, dim3(ew_blocks(total)), dim3(256), 0, stream, x, batch, h, w, channels, factor,
                     alpha, out);
  return ok_or(hipGetLastError());
}

int sg2im_maxpool_forward(const float* x, int batch, int h, int w, int channels, int factor, float* out,
                          hipStream_t stream) {
  if (!x || !out || factor < 1 || channels < 1) return SG2IM_ERR_ARG;
  const long long total = (long long)batch * (h / factor) * (w / factor) * channels;
  if (total == 0) return SG2IM_OK;
  SG2IM_LAUNCH(maxpool_fwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, stream, x, batch, h, w, channels, factor,
                     out);
  return ok_or(hipGetLastError());
}

int sg2im_maxpool_backward(const float* x, const float* dy, int batch, int h, int w, int channels, int factor,
                           float* dx, hipStream_t stream) {
  if (!x || !dy || !dx || factor < 1 || channels < 1) return SG2IM_ERR_ARG;
  const long long total = (long long)batch * h * w * channels;
  if (total == 0) return SG2IM_OK;
  SG2IM_LAUNCH(maxpool_bwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, stream, x, dy, batch, h, w, channels,
                     factor, dx);
  return ok_or(hipGetLastError());
}

int sg2im_leaky_forward(const float* x, long long n, float slope, float* out, hipStream_t stream) {
  if (!x || !out || n < 0) return SG2IM_ERR_ARG;
  if (n == 0) return SG2IM_OK;
  SG2IM_LAUNCH(leaky_fwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, stream, x, n, slope, out);
  return ok_or(hipGetLastError());
}

int sg2im_add_forward(const float* a, const float* b, long long n, float* out, hipStream_t stream) {
  if (!a || !b || !out || n < 0) return SG2IM_ERR_ARG;
  if (n == 0) return SG2IM_OK;
  SG2IM_LAUNCH(add_kernel, dim3(ew_blocks(n)), dim3(256), 0, stream, a, b, n, out);
  return ok_or(hipGetLastError());
}

int sg2im_avgpool_forward(const float* x, int batch, int h, int w, int channels, int factor,
                          float* out, hipStream_t stream) {
  if (!x || !out || factor < 1 || h % factor || w % factor) return SG2IM_ERR_ARG;
  const long long total = (long long)batch * (h / factor) * (w / factor) * channels;
  if (total == 0) return SG2IM_OK;
  if (channels % 4 == 0 && al16(x) && al16(out))
    SG2IM_LAUNCH(avgpool_v4_kernel, dim3(ew_blocks(total / 4)), dim3(256), 0, stream, x, batch, h, w, channels,
                       factor, out);
  else
    SG2IM_LAUNCH(avgpool_kernel, dim3(ew_blocks(total)), dim3(256), 0, stream, x, batch, h, w, channels, factor, out);
  return ok_or(hipGetLastError());
}

int sg2im_pyramid_backward(const float* const* dlevels, const int* factors, const long long* lds,
                           int n_levels, int batch, int h, int w, int channels, float* dlayout,
                           long long ld_out, hipStream_t stream) {
  if (!dlevels || !factors || !lds || n_levels < 1 || n_levels > 8 || !dlayout) return SG2IM_ERR_ARG;
  PyramidArgs a;
  a.n = n_levels;
  for (int l = 0; l < 8; ++l) {
    a.lvl[l] = l < n_levels ? dlevels[l] : nullptr;
    a.ld[l] = l < n_levels ? lds[l] : 0;
    a.f[l] = l < n_levels ? factors[l] : 1;
    if (l < n_levels && (factors[l] < 1 || h % factors[l] || w % factors[l])) return SG2IM_ERR_ARG;
  }
  const long long total = (long long)batch * h * w * channels;
  if (total == 0) return SG2IM_OK;
  bool v4 = channels % 4 == 0 && ld_out % 4 == 0 && al16(dlayout);
  for (int l = 0; l < n_levels; ++l) v4 = v4 && lds[l] % 4 == 0 && al16(dlevels[l]);
  if (v4)
    SG2IM_LAUNCH(pyramid_bwd_v4_kernel, dim3((unsigned)(batch * h)), dim3(256), 0, stream, a, batch, h, w,
                       channels, dlayout, ld_out);
  else
    SG2IM_LAUNCH(pyramid_bwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, stream, a, batch, h, w, channels,
                       dlayout, ld_out);
  return ok_or(hipGetLastError());
}

int sg2im_nchw_to_nhwc(const float* src, int batch, int channels, int h, int w, float* dst,
                       long long ld_dst, int c_offset, hipStream_t stream) {
  if (!src || !dst) return SG2IM_ERR_ARG;
  const long long total = (long long)batch * channels * h * w;
  if (total == 0) return SG2IM_OK;
  SG2IM_LAUNCH(nchw_to_nhwc_kernel, dim3(ew_blocks(total)), dim3(256), 0, stream, src, batch, channels, h, w,
                     dst, ld_dst, c_offset);
  return ok_or(hipGetLastError());
}

int sg2im_nhwc_to_nchw(const float* src, long long ld_src, int c_offset, int batch, int channels,
                       int h, int w, float* dst, hipStream_t stream) {
  if (!src || !dst) return SG2IM_ERR_ARG;
  const long long total = (long long)batch * channels * h * w;
  if (total == 0) return SG2IM_OK;
  SG2IM_LAUNCH(nhwc_to_nchw_kernel, dim3(ew_blocks(total)), dim3(256), 0, stream, src, ld_src, c_offset, batch,
                     channels, h, w, dst);
  return ok_or(hipGetLastError());
}

int sg2im_gap_forward(const float* x, int batch, int hw, int channels, float* out, hipStream_t stream) {
  if (!x || !out || hw < 1) return SG2IM_ERR_ARG;
  if (batch * channels == 0) return SG2IM_OK;
  SG2IM_LAUNCH(gap_fwd_kernel, dim3((batch * channels + 255) / 256), dim3(256), 0, stream, x, batch, hw, channels, out);
  return ok_or(hipGetLastError());
}

int sg2im_gap_backward(const float* dout, int batch, int hw, int channels, float* dx, hipStream_t stream) {
  if (!dout || !dx || hw < 1) return SG2IM_ERR_ARG;
  const long long total = (long long)batch * hw * channels;
  if (total == 0) return SG2IM_OK;
  SG2IM_LAUNCH(gap_bwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, stream, dout, batch, hw, channels, dx);
  return ok_or(hipGetLastError());
}

int sg2im_sigmoid_forward(const float* x, long long n, float* y, hipStream_t stream) {
  if (!x || !y) return SG2IM_ERR_ARG;
  if (n == 0) return SG2IM_OK;
  SG2IM_LAUNCH(sigmoid_fwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, stream, x, n, y);
  return ok_or(hipGetLastError());
}

int sg2im_sigmoid_backward(const float* y, const float* dy, long long n, float* dx, hipStream_t stream) {
  if (!y || !dy || !dx) return SG2IM_ERR_ARG;
  if (n == 0) return SG2IM_OK;
  SG2IM_LAUNCH(sigmoid_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, stream, y, dy, n, dx);
  return ok_or(hipGetLastError());
}

int sg2im_column_sum(const float* x, long long rows, int cols, long long ld, float* out, int accumulate,
                     float* partial, hipStream_t stream) {
  if (!x || !out || !partial || cols < 1) return SG2IM_ERR_ARG;
  if (rows == 0) {
    if (!accumulate && hipMemsetAsync(out, 0, sizeof(float) * cols, stream) != hipSuccess) return SG2IM_ERR_HIP;
    return SG2IM_OK;
  }
  const int nblk = red_blocks(rows);
  if (cols % 4 == 0 && ld % 4 == 0 && al16(x) && al16(partial))
    SG2IM_LAUNCH(colsum_partial_v4_kernel, dim3(nblk), dim3(256), 8 * 256 * sizeof(float), stream, x, rows, cols, ld,
                       partial);
  else
    SG2IM_LAUNCH(colsum_partial_kernel, dim3(nblk), dim3(256), 2 * 256 * sizeof(float), stream, x, rows, cols, ld,
                       partial);
  SG2IM_LAUNCH(colsum_final_kernel, dim3((cols + 3) / 4), dim3(256), 0, stream, partial, nblk, cols, out, accumulate);
  return ok_or(hipGetLastError());
}

}  // extern "C"
