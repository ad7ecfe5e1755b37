// Loss kernels (reference sg2im/losses.py, scripts/train.py:387-412) and the fused Adam
// update (torch.optim.Adam defaults, scripts/train.py:426-443).  Every loss kernel
// produces the scalar loss AND d(loss)/d(input) in one pass over the input (HBM-bound,
// one read + one write); the scalar is reduced deterministically: per-workgroup partials
// in a fixed slot, summed in double by a single-thread epilogue kernel.
#include <algorithm>
#include <cmath>
#include <hip/hip_runtime.h>
#include "launch_count.h"
#include "sg2im_hip.h"

namespace sg2im {

constexpr int LOSS_BLOCKS = 256;

__device__ __forceinline__ float block_sum(float v) {
  __shared__ float ws[4];
  #pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) ws[wave] = v;
  __syncthreads();
  float s = 0.f;
  if (threadIdx.x == 0) for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += ws[w];
  return s;    // valid on thread 0
}

// Padded row batches (sg2im_amd/bucketing.py): only the first count[0] * unit of the n elements /
// rows are real; the mean runs over those and the padding gets a zero gradient.  count == nullptr:
// everything is real.
__device__ __forceinline__ long long live_count(long long n, const int* __restrict__ count, int unit) {
  if (!count) return n;
  const long long t = (long long)count[0] * unit;
  return t < n ? (t > 0 ? t : 1) : n;
}

enum { L_L1 = 0, L_MSE = 1, L_BCE = 2, L_BCE_PROB = 3, L_MEAN = 4, L_LSGAN = 5 };

template <int KIND>
__global__ void elementwise_loss_kernel(const float* __restrict__ x, const float* __restrict__ t, long long n,
                                        float target, float weight, float* __restrict__ grad,
                                        float* __restrict__ partial, const int* __restrict__ count, int unit) {
  float s = 0.f;
  const long long live = live_count(n, count, unit);
  const float gscale = (float)((double)weight / (double)live);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    if (i >= live) { if (grad) grad[i] = 0.f; continue; }
    const float v = x[i];
    float l, g;
    if (KIND == L_L1) {
      const float d = v - t[i];
      l = fabsf(d); g = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    } else if (KIND == L_MSE) {
      const float d = v - t[i];
      l = d * d; g = 2.f * d;
    } else if (KIND == L_BCE) {
      // losses.py:55-56: max(x,0) - x*t + log(1 + exp(-|x|))
      const float e = expf(-fabsf(v));
      l = fmaxf(v, 0.f) - v * target + logf(1.f + e);
      const float sig = v >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
      g = sig - target;
    } else if (KIND == L_BCE_PROB) {
      // F.binary_cross_entropy on probabilities (train.py:408): logs clamped at -100,
      // gradient (p - t) / max(p (1 - p), 1e-12) as ATen computes it
      const float y = t[i];
      l = -(y * fmaxf(logf(v), -100.f) + (1.f - y) * fmaxf(logf(1.f - v), -100.f));
      g = (v - y) / fmaxf((1.f - v) * v, 1e-12f);
    } else if (KIND == L_MEAN) {
      // WGAN terms (losses.py:106-124): +-mean(scores); `target` carries the sign
      l = target * v; g = target;
    } else {
      // LSGAN (losses.py:127-145): mse(sigmoid(x), target)
      const float sg = 1.f / (1.f + expf(-v));
      const float d = sg - target;
      l = d * d; g = 2.f * d * sg * (1.f - sg);
    }
    s += l;
    if (grad) grad[i] = g * gscale;
  }
  s = block_sum(s);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// one wavefront: lane l adds partial[l], partial[l + 64], ... (ascending), the 64 lane sums are
// then folded by a fixed butterfly - a fixed order, reproducible run to run
__global__ void loss_final_kernel(const float* __restrict__ partial, int n, double weight, long long total,
                                  const int* __restrict__ count, int unit, float* __restrict__ loss) {
  const int lane = threadIdx.x;
  const double scale = weight / (double)live_count(total, count, unit);
  double s = 0.0;
  for (int i = lane; i < n; i += 64) s += partial[i];
  #pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) loss[0] = (float)(s * scale);
}

// one wavefront per row
__global__ void cross_entropy_kernel(const float* __restrict__ scores, int rows, int C,
                                     const long long* __restrict__ labels, float weight,
                                     float* __restrict__ grad, float* __restrict__ partial,
                                     const int* __restrict__ count, int unit) {
  const int r = blockIdx.x, lane = threadIdx.x;
  const long long live = live_count(rows, count, unit);
  const float gscale = (float)((double)weight / (double)live);
  if (r >= live) {                                   // padding row: no loss, no gradient
    if (lane == 0) partial[r] = 0.f;
    if (grad) for (int c = lane; c < C; c += 64) grad[(long long)r * C + c] = 0.f;
    return;
  }
  const float* s = scores + (long long)r * C;
  float mx = -INFINITY;
  for (int c = lane; c < C; c += 64) mx = fmaxf(mx, s[c]);
  #pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
  float sum = 0.f;
  for (int c = lane; c < C; c += 64) sum += expf(s[c] - mx);
  #pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
  const int y = (int)labels[r];
  const float lse = mx + logf(sum);
  if (lane == 0) partial[r] = lse - s[y];
  if (grad) {
    for (int c = lane; c < C; c += 64) {
      const float p = expf(s[c] - mx) / sum;
      grad[(long long)r * C + c] = (p - (c == y ? 1.f : 0.f)) * gscale;
    }
  }
}

__global__ void scale_by_scalar_kernel(const float* __restrict__ x, const float* __restrict__ a, long long n,
                                       float* __restrict__ y) {
  const float s = a[0];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    y[i] = x[i] * s;
}

struct ScalarPtrs { const float* p[8]; };

// out = p[0][0] + p[1][0] + ... in that order (the total of the weighted loss terms)
__global__ void sum_scalars_kernel(ScalarPtrs t, int n, float* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = t.p[0][0];
    for (int i = 1; i < n; ++i) s += t.p[i][0];
    out[0] = s;
  }
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long long n, float one_minus_b1, float b2, float one_minus_b2,
                            float step_size, float inv_bc2_sqrt, float eps, float gscale) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const float gi = g[i] * gscale;
    const float mi = m[i] + (gi - m[i]) * one_minus_b1;          // exp_avg.lerp_(grad, 1-beta1)
    const float vi = v[i] * b2 + gi * gi * one_minus_b2;         // mul_(beta2).addcmul_(g, g, 1-beta2)
    const float denom = sqrtf(vi) * inv_bc2_sqrt + eps;
    p[i] = p[i] - step_size * (mi / denom);
    m[i] = mi; v[i] = vi;
  }
}

// Guarded variant: the step counter lives on the device and only advances when the
// guard scalar (the generator's total loss) is finite - the device-side form of the
// reference's "if not math.isfinite(total_loss): continue" (scripts/train.py:553-555),
// which needs a host sync there.  state[0] = step count (as float), state[1] = step_size,
// state[2] = 1/sqrt(bias_correction2), state[3] = 1.0 if this step is applied else 0.0.
__global__ void adam_prepare_kernel(float* __restrict__ state, const float* __restrict__ guard, float lr,
                                    float b1, float b2) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const bool ok = guard ? isfinite(guard[0]) : true;
    if (ok) {
      const float t = state[0] + 1.f;
      state[0] = t;
      const double bc1 = 1.0 - pow((double)b1, (double)t);
      const double bc2 = 1.0 - pow((double)b2, (double)t);
      state[1] = (float)((double)lr / bc1);
      state[2] = (float)(1.0 / sqrt(bc2));
    }
    state[3] = ok ? 1.f : 0.f;
  }
}

__global__ void adam_guarded_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                    float* __restrict__ v, long long n, float one_minus_b1, float b2,
                                    float one_minus_b2, const float* __restrict__ state, float eps, float gscale) {
  if (state[3] == 0.f) return;
  const float step_size = state[1], inv_bc2_sqrt = state[2];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const float gi = g[i] * gscale;
    const float mi = m[i] + (gi - m[i]) * one_minus_b1;
    const float vi = v[i] * b2 + gi * gi * one_minus_b2;
    const float denom = sqrtf(vi) * inv_bc2_sqrt + eps;
    p[i] = p[i] - step_size * (mi / denom);
    m[i] = mi; v[i] = vi;
  }
}

static inline int ok_or(hipError_t e) { return e == hipSuccess ? SG2IM_OK : SG2IM_ERR_HIP; }

template <int KIND>
static int run_elementwise(const float* x, const float* t, long long n, float target, float weight,
                           float* loss, float* grad, float* partial, hipStream_t stream,
                           const int* count = nullptr, int unit = 1) {
  constexpr bool needs_t = KIND == L_L1 || KIND == L_MSE || KIND == L_BCE_PROB;
  if (!x || !loss || !partial || n < 1 || (needs_t && !t)) return SG2IM_ERR_ARG;
  const int blocks = (int)std::min<long long>(LOSS_BLOCKS, (n + 255) / 256);
  SG2IM_LAUNCH((elementwise_loss_kernel<KIND>), dim3(blocks), dim3(256), 0, stream, x, t, n, target,
                     weight, grad, partial, count, unit);
  SG2IM_LAUNCH(loss_final_kernel, dim3(1), dim3(64), 0, stream, partial, blocks, (double)weight, n, count, unit, loss);
  return ok_or(hipGetLastError());
}

}  // namespace sg2im

using namespace sg2im;

extern "C" {

int sg2im_l1_loss(const float* pred, const float* target, long long n, float weight, float* loss,
                  float* grad, float* partial, hipStream_t stream) {
  return run_elementwise<L_L1>(pred, target, n, 0.f, weight, loss, grad, partial, stream);
}

int sg2im_mse_loss(const float* pred, const float* target, long long n, float weight, float* loss,
                   float* grad, float* partial, const int* count, int count_unit, hipStream_t stream) {
  return run_elementwise<L_MSE>(pred, target, n, 0.f, weight, loss, grad, partial, stream, count, count_unit);
}

int sg2im_bce_logits_loss(const float* x, long long n, float target, float weight, float* loss,
                          float* grad, float* partial, const int* count, int count_unit, hipStream_t stream) {
  return run_elementwise<L_BCE>(x, nullptr, n, target, weight, loss, grad, partial, stream, count, count_unit);
}

int sg2im_gan_score_loss(const float* x, long long n, int kind, float target, float weight, float* loss,
                         float* grad, float* partial, const int* count, int count_unit, hipStream_t stream) {
  if (kind == 0) return run_elementwise<L_BCE>(x, nullptr, n, target, weight, loss, grad, partial, stream, count, count_unit);
  if (kind == 1) return run_elementwise<L_MEAN>(x, nullptr, n, target, weight, loss, grad, partial, stream, count, count_unit);
  if (kind == 2) return run_elementwise<L_LSGAN>(x, nullptr, n, target, weight, loss, grad, partial, stream, count, count_unit);
  return SG2IM_ERR_ARG;
}

int sg2im_bce_prob_loss(const float* prob, const float* target, long long n, float weight, float* loss,
                        float* grad, float* partial, const int* count, int count_unit, hipStream_t stream) {
  return run_elementwise<L_BCE_PROB>(prob, target, n, 0.f, weight, loss, grad, partial, stream, count, count_unit);
}

int sg2im_cross_entropy_loss(const float* scores, int rows, int classes, const long long* labels,
                             float weight, float* loss, float* grad, float* partial,
                             const int* count, int count_unit, hipStream_t stream) {
  if (!scores || !labels || !loss || !partial || rows < 1 || classes < 1) return SG2IM_ERR_ARG;
  SG2IM_LAUNCH(cross_entropy_kernel, dim3(rows), dim3(64), 0, stream, scores, rows, classes, labels,
                     weight, grad, partial, count, count_unit);
  SG2IM_LAUNCH(loss_final_kernel, dim3(1), dim3(64), 0, stream, partial, rows, (double)weight, (long long)rows,
                     count, count_unit, loss);
  return ok_or(hipGetLastError());
}

int sg2im_scale_by_scalar(const float* x, const float* a_dev, long long n, float* y, hipStream_t stream) {
  if (!x || !a_dev || !y) return SG2IM_ERR_ARG;
  if (n == 0) return SG2IM_OK;
  const int blocks = (int)std::min<long long>((n + 255) / 256, 4096);
  SG2IM_LAUNCH(scale_by_scalar_kernel, dim3(blocks), dim3(256), 0, stream, x, a_dev, n, y);
  return ok_or(hipGetLastError());
}

int sg2im_sum_scalars(const float* const* terms, int n, float* out, hipStream_t stream) {
  if (!terms || !out || n < 1 || n > 8) return SG2IM_ERR_ARG;
  ScalarPtrs t;
  for (int i = 0; i < 8; ++i) t.p[i] = i < n ? terms[i] : nullptr;
  for (int i = 0; i < n; ++i) if (!t.p[i]) return SG2IM_ERR_ARG;
  SG2IM_LAUNCH(sum_scalars_kernel, dim3(1), dim3(64), 0, stream, t, n, out);
  return ok_or(hipGetLastError());
}

int sg2im_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n,
                    float lr, float beta1, float beta2, float eps, int step, float grad_scale,
                    hipStream_t stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || step < 1) return SG2IM_ERR_ARG;
  if (n == 0) return SG2IM_OK;
  const double bc1 = 1.0 - std::pow((double)beta1, step);
  const double bc2 = 1.0 - std::pow((double)beta2, step);
  const int blocks = (int)std::min<long long>((n + 255) / 256, 16384);
  SG2IM_LAUNCH(adam_kernel, dim3(blocks), dim3(256), 0, stream, param, grad, exp_avg, exp_avg_sq, n,
                     1.f - beta1, beta2, 1.f - beta2, (float)((double)lr / bc1), (float)(1.0 / std::sqrt(bc2)), eps,
                     grad_scale);
  return ok_or(hipGetLastError());
}

}  // extern "C"

extern "C" int sg2im_adam_step_guarded(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                                       long long n, float lr, float beta1, float beta2, float eps,
                                       float grad_scale, float* state, const float* guard,
                                       hipStream_t stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || !state) return SG2IM_ERR_ARG;
  SG2IM_LAUNCH(sg2im::adam_prepare_kernel, dim3(1), dim3(64), 0, stream, state, guard, lr, beta1, beta2);
  if (n > 0) {
    const int blocks = (int)std::min<long long>((n + 255) / 256, 16384);
    SG2IM_LAUNCH(sg2im::adam_guarded_kernel, dim3(blocks), dim3(256), 0, stream, param, grad, exp_avg,
                       exp_avg_sq, n, 1.f - beta1, beta2, 1.f - beta2, state, eps, grad_scale);
  }
  return hipGetLastError() == hipSuccess ? SG2IM_OK : SG2IM_ERR_HIP;
}

// The two halves of sg2im_adam_step_guarded as separate calls: sg2im_adam_prepare_guarded advances the step counter /
// bias corrections in `state` once per optimiser step (or marks the step as skipped), sg2im_adam_apply_guarded applies
// the update to ANY slice of the arena - a part of the arena whose gradients are complete early can then be updated
// while the rest of the backward pass is still running.
extern "C" int sg2im_adam_prepare_guarded(float lr, float beta1, float beta2, float* state, const float* guard,
                                          hipStream_t stream) {
  if (!state) return SG2IM_ERR_ARG;
  SG2IM_LAUNCH(sg2im::adam_prepare_kernel, dim3(1), dim3(64), 0, stream, state, guard, lr, beta1, beta2);
  return hipGetLastError() == hipSuccess ? SG2IM_OK : SG2IM_ERR_HIP;
}

extern "C" int sg2im_adam_apply_guarded(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n,
                                        float beta1, float beta2, float eps, float grad_scale, const float* state,
                                        hipStream_t stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || !state || n < 0) return SG2IM_ERR_ARG;
  if (n == 0) return SG2IM_OK;
  const int blocks = (int)std::min<long long>((n + 255) / 256, 16384);
  SG2IM_LAUNCH(sg2im::adam_guarded_kernel, dim3(blocks), dim3(256), 0, stream, param, grad, exp_avg, exp_avg_sq, n,
                     1.f - beta1, beta2, 1.f - beta2, state, eps, grad_scale);
  return hipGetLastError() == hipSuccess ? SG2IM_OK : SG2IM_ERR_HIP;
}
