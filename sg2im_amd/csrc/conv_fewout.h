// 1 x 1 convolutions with at most four OUTPUT channels - the refinement network's output_conv[2] (64 -> 3 image
// channels, crn.py:84) and mask_net's last layer (128 -> 1 mask logit, model.py:105): backward passes without the
// matrix cores.  K = cout <= 4 leaves a 64 x 64 MFMA tile >= 94 % padding and the split-K plan of a 131 072-row
// reduction needs a finish launch: the generic kernels take 24 / 48 us (data / weight gradient of output_conv[2])
// and 50 / 126 us under contention (mask_net, inside the tail that ends a VG-style step) for what is one pass over
// 33.5 - 42 MB.  Both are HBM-bound elementwise / reduction kernels here (fp32 arithmetic, as the scalar-loader
// implicit-GEMM launches they replace: a dY with 1 or 3 channels never took the bf16 path).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sg2im {

// dx[m][c] = sum_o dy[m][o] * W[o][c_begin + c]  (an fmaf chain over o = 0 .. CO-1 starting from +0: the value the
// fp32 MFMA kernel produces), optionally * leaky'(act[m][c]) - sg2im_conv2d_backward_data_act's mask arithmetic.
// thread = (row, float4 of channels); c4n = c_count / 4
template <int CO>
__global__ __launch_bounds__(256) void conv1x1_fewout_dgrad_kernel(const float* __restrict__ dy, int ld_dy,
                                                                   const float* __restrict__ W, int wrow, int c_begin,
                                                                   int c4n, long long M, float* __restrict__ dx,
                                                                   long long ld_dx, const float* __restrict__ act,
                                                                   long long ld_act, float slope) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long m = idx / c4n;
  const int c4 = (int)(idx - m * c4n);
  if (m >= M) return;
  float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
  #pragma unroll
  for (int o = 0; o < CO; ++o) {
    const float d = dy[m * ld_dy + o];
    const float4 w = *reinterpret_cast<const float4*>(W + (long long)o * wrow + c_begin + 4 * c4);
    r.x = fmaf(d, w.x, r.x); r.y = fmaf(d, w.y, r.y); r.z = fmaf(d, w.z, r.z); r.w = fmaf(d, w.w, r.w);
  }
  if (act) {
    const float4 a = *reinterpret_cast<const float4*>(act + m * ld_act + 4 * c4);
    r.x *= a.x > 0.f ? 1.f : slope; r.y *= a.y > 0.f ? 1.f : slope;
    r.z *= a.z > 0.f ? 1.f : slope; r.w *= a.w > 0.f ? 1.f : slope;
  }
  *reinterpret_cast<float4*>(dx + m * ld_dx + 4 * c4) = r;
}

// dW[o][c] = sum_m dy[m][o] * x'[m][c],  dB[o] = sum_m dy[m][o]   (x' = leaky(x * scale + shift) when an affine is pending)
// Stage 1: workgroup b reduces the rows [b * per, (b + 1) * per): thread = (float4 lane c4 of C4 = C / 4, row group rg
// of R = 256 / C4), rows strided by R, then the R row groups are summed through LDS in order -> part[b][o][C] and,
// behind the nblk * CO * C weight partials, bpart[b][o].  Stage 2 sums the blocks in order.  No atomics: reproducible.
template <int CO>
__global__ __launch_bounds__(256) void conv1x1_fewout_wgrad_kernel(const float* __restrict__ x, long long ld_x,
                                                                   const float* __restrict__ scale,
                                                                   const float* __restrict__ shift, float slope, int C4,
                                                                   const float* __restrict__ dy, int ld_dy, long long M,
                                                                   long long per, int nblk, float* __restrict__ part) {
  __shared__ float4 red[256 * CO];
  __shared__ float bred[64 * CO];
  const int tid = threadIdx.x;
  const int c4 = tid % C4, rg = tid / C4, R = 256 / C4;
  const long long r0 = (long long)blockIdx.x * per, r1 = (r0 + per < M) ? r0 + per : M;
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (scale) { sc = *reinterpret_cast<const float4*>(scale + 4 * c4); sh = *reinterpret_cast<const float4*>(shift + 4 * c4); }
  float4 acc[CO];
  float bacc[CO];
  #pragma unroll
  for (int o = 0; o < CO; ++o) { acc[o] = make_float4(0.f, 0.f, 0.f, 0.f); bacc[o] = 0.f; }
  #pragma unroll 4
  for (long long m = r0 + rg; m < r1; m += R) {
    float4 v = *reinterpret_cast<const float4*>(x + m * ld_x + 4 * c4);
    if (scale) {
      v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y); v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
      v.x = fmaxf(v.x, v.x * slope); v.y = fmaxf(v.y, v.y * slope); v.z = fmaxf(v.z, v.z * slope); v.w = fmaxf(v.w, v.w * slope);
    }
    #pragma unroll
    for (int o = 0; o < CO; ++o) {
      const float d = dy[m * ld_dy + o];
      acc[o].x = fmaf(d, v.x, acc[o].x); acc[o].y = fmaf(d, v.y, acc[o].y);
      acc[o].z = fmaf(d, v.z, acc[o].z); acc[o].w = fmaf(d, v.w, acc[o].w);
      bacc[o] += d;
    }
  }
  #pragma unroll
  for (int o = 0; o < CO; ++o) {
    red[(rg * CO + o) * C4 + c4] = acc[o];
    if (c4 == 0) bred[rg * CO + o] = bacc[o];
  }
  __syncthreads();
  const int C = 4 * C4;
  for (int e = tid; e < CO * C4; e += 256) {
    const int o = e / C4, k = e - o * C4;
    float4 s = red[o * C4 + k];
    for (int g = 1; g < R; ++g) {
      const float4 t = red[(g * CO + o) * C4 + k];
      s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    *reinterpret_cast<float4*>(part + ((long long)blockIdx.x * CO + o) * C + 4 * k) = s;
  }
  if (tid < CO) {
    float s = bred[tid];
    for (int g = 1; g < R; ++g) s += bred[g * CO + tid];
    part[(long long)nblk * CO * C + (long long)blockIdx.x * CO + tid] = s;
  }
}

// one wavefront per element e: e < CO * C: dweight[o * wrow + c] (+)= sum_b part[b][o][c];  e >= CO * C: the bias sums.
// Lanes stride over the blocks, then a fixed xor tree.
__global__ __launch_bounds__(64) void conv1x1_fewout_wgrad_finish_kernel(const float* __restrict__ part, int nblk, int CO,
                                                                         int C, float* __restrict__ dweight, int wrow,
                                                                         float* __restrict__ dbias, int accumulate) {
  const int e = blockIdx.x, lane = threadIdx.x;
  const int nw = CO * C;
  const float* __restrict__ q = e < nw ? part + e : part + (long long)nblk * nw + (e - nw);
  const int stride = e < nw ? nw : CO;
  float s = 0.f;
  #pragma unroll 4
  for (int b = lane; b < nblk; b += 64) s += q[(long long)b * stride];
  #pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if (lane != 0) return;
  if (e < nw) {
    float* dst = dweight + (long long)(e / C) * wrow + (e % C);
    *dst = (accumulate ? *dst : 0.f) + s;
  } else if (dbias) {
    dbias[e - nw] = (accumulate ? dbias[e - nw] : 0.f) + s;
  }
}

}  // namespace sg2im
