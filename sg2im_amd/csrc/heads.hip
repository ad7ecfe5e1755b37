// Two linear heads on the same row matrix in ONE launch per direction: the object discriminator's real / fake score
// and its class logits (reference sg2im/discriminators.py:66-75: real_classifier (1024 -> 1) and obj_classifier
// (1024 -> num_objects) both read the pooled feature vector), and the sum of their input gradients.
//
//   y1 = x W1^T + b1   [rows][n1]          y2 = x W2^T + b2   [rows][n2]
//   dx = g1 W1 + g2 W2 [rows][k]
//
// As GEMMs these are a 1-column and a ~180-column problem over a few hundred rows: two tile launches + two split-K
// finishes forward (45 us between the refinement network's forward and backward, profiles/r4_step_kernel_sequence.txt),
// two data gradients + a finish + an add backward (55 us).  Here a workgroup owns RT rows (staged in LDS) and a block
// of output columns: latency-bound either way, but one launch.  Fixed summation order (no atomics).
// The weight / bias gradients stay with the implicit-GEMM family (sg2im_conv2d_backward_weight_group).
#define SG2IM_GEMM_TU 1      // (counted with the GEMM family: these launches replace GEMM launches)
#include <hip/hip_runtime.h>
#include "sg2im_hip.h"
#include "launch_count.h"

namespace sg2im {
namespace heads {

constexpr int RT = 8;            // rows per workgroup
constexpr int CT = 32;           // forward: output columns per workgroup (8 per wave)
constexpr int KT = 256;          // data gradient: dx columns per workgroup (one per thread)
constexpr int THREADS = 256;
constexpr int MAXK = 1536;       // forward: x rows staged in LDS, RT * MAXK floats = 48 KB
constexpr int MAXN = 2048;       // data gradient: gradient rows staged in LDS, RT * MAXN floats

typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ const float* head_row(const float* W1, int n1, const float* W2, int n, int K) {
  return n < n1 ? W1 + (long long)n * K : W2 + (long long)(n - n1) * K;
}

// forward: workgroup (row block, column block); wave w takes 8 columns of [W1; W2], its 64 lanes split K in float4
// pieces.  A wave issues ALL its weight loads of a 256-wide K slab (8 columns x 16 bytes per lane) before the first
// multiply - the first version walked the columns one dependent load at a time and took 80 us - then RT x 8 partial
// dot products per lane, xor-tree over the lanes at the end.
__global__ __launch_bounds__(THREADS) void two_heads_fwd_kernel(
    const float* __restrict__ x, long long ldx, int rows, int K,
    const float* __restrict__ W1, const float* __restrict__ b1, int n1,
    const float* __restrict__ W2, const float* __restrict__ b2, int n2,
    float* __restrict__ y1, long long ld1, float* __restrict__ y2, long long ld2) {
  extern __shared__ __attribute__((aligned(16))) float xs[];        // [RT][K]
  const int r0 = blockIdx.x * RT;
  const int k4 = K >> 2;
  for (int i = threadIdx.x; i < RT * k4; i += THREADS) {
    const int r = i / k4, c = i - r * k4;
    v4f v = {0.f, 0.f, 0.f, 0.f};
    if (r0 + r < rows) v = *reinterpret_cast<const v4f*>(x + (long long)(r0 + r) * ldx + 4 * c);
    *reinterpret_cast<v4f*>(xs + r * K + 4 * c) = v;
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int N = n1 + n2;
  constexpr int CW = CT / (THREADS / 64);                            // columns per wave
  const int nb = blockIdx.y * CT + wave * CW;
  if (nb >= N) return;
  const float* wrow[CW];
  #pragma unroll
  for (int j = 0; j < CW; ++j) wrow[j] = head_row(W1, n1, W2, min(nb + j, N - 1), K);     // (clamped: stored only if < N)
  float acc[CW][RT];
  #pragma unroll
  for (int j = 0; j < CW; ++j)
    #pragma unroll
    for (int r = 0; r < RT; ++r) acc[j][r] = 0.f;
  for (int c = lane; c < k4; c += 64) {
    v4f w[CW];
    #pragma unroll
    for (int j = 0; j < CW; ++j) w[j] = *reinterpret_cast<const v4f*>(wrow[j] + 4 * c);
    #pragma unroll
    for (int r = 0; r < RT; ++r) {
      const v4f v = *reinterpret_cast<const v4f*>(xs + r * K + 4 * c);
      #pragma unroll
      for (int j = 0; j < CW; ++j)
        acc[j][r] = fmaf(w[j].x, v.x, fmaf(w[j].y, v.y, fmaf(w[j].z, v.z, fmaf(w[j].w, v.w, acc[j][r]))));
    }
  }
  #pragma unroll
  for (int off = 32; off > 0; off >>= 1)
    #pragma unroll
    for (int j = 0; j < CW; ++j)
      #pragma unroll
      for (int r = 0; r < RT; ++r) acc[j][r] += __shfl_xor(acc[j][r], off);
  // lane j * RT + r stores (column nb + j, row r0 + r)
  #pragma unroll
  for (int j = 0; j < CW; ++j)
    #pragma unroll
    for (int r = 0; r < RT; ++r)
      if (lane == j * RT + r) {
        const int n = nb + j;
        if (n < N && r0 + r < rows) {
          if (n < n1) y1[(long long)(r0 + r) * ld1 + n] = acc[j][r] + (b1 ? b1[n] : 0.f);
          else y2[(long long)(r0 + r) * ld2 + (n - n1)] = acc[j][r] + (b2 ? b2[n - n1] : 0.f);
        }
      }
}

// data gradient: workgroup (row block, block of 256 dx columns); the RT gradient rows [g1 | g2] in LDS as [n][RT];
// a thread owns ONE dx column and walks the n1 + n2 weight rows (a row's 256 values = one coalesced 1 KB load per
// workgroup), 16 loads in flight, head 1 first, rows in ascending order
__global__ __launch_bounds__(THREADS) void two_heads_dgrad_kernel(
    const float* __restrict__ g1, long long ldg1, const float* __restrict__ g2, long long ldg2, int rows, int K,
    const float* __restrict__ W1, int n1, const float* __restrict__ W2, int n2,
    float* __restrict__ dx, long long lddx) {
  extern __shared__ __attribute__((aligned(16))) float gs[];        // [n1 + n2][RT]
  const int r0 = blockIdx.x * RT;
  const int N = n1 + n2;
  for (int i = threadIdx.x; i < N * RT; i += THREADS) {
    const int n = i / RT, r = i - n * RT;
    float v = 0.f;
    if (r0 + r < rows) v = n < n1 ? g1[(long long)(r0 + r) * ldg1 + n] : g2[(long long)(r0 + r) * ldg2 + (n - n1)];
    gs[i] = v;
  }
  __syncthreads();
  const int k = blockIdx.y * KT + threadIdx.x;
  if (k >= K) return;
  float acc[RT];
  #pragma unroll
  for (int r = 0; r < RT; ++r) acc[r] = 0.f;
  constexpr int U = 16;
  int n = 0;
  for (; n + U <= N; n += U) {
    float w[U];
    #pragma unroll
    for (int u = 0; u < U; ++u) w[u] = head_row(W1, n1, W2, n + u, K)[k];
    #pragma unroll
    for (int u = 0; u < U; ++u) {
      const v4f ga = *reinterpret_cast<const v4f*>(gs + (n + u) * RT);
      const v4f gb = *reinterpret_cast<const v4f*>(gs + (n + u) * RT + 4);
      acc[0] = fmaf(w[u], ga.x, acc[0]); acc[1] = fmaf(w[u], ga.y, acc[1]);
      acc[2] = fmaf(w[u], ga.z, acc[2]); acc[3] = fmaf(w[u], ga.w, acc[3]);
      acc[4] = fmaf(w[u], gb.x, acc[4]); acc[5] = fmaf(w[u], gb.y, acc[5]);
      acc[6] = fmaf(w[u], gb.z, acc[6]); acc[7] = fmaf(w[u], gb.w, acc[7]);
    }
  }
  for (; n < N; ++n) {
    const float w = head_row(W1, n1, W2, n, K)[k];
    #pragma unroll
    for (int r = 0; r < RT; ++r) acc[r] = fmaf(w, gs[n * RT + r], acc[r]);
  }
  #pragma unroll
  for (int r = 0; r < RT; ++r)
    if (r0 + r < rows) dx[(long long)(r0 + r) * lddx + k] = acc[r];
}

static_assert(RT == 8, "the data-gradient kernel reads the eight rows' gradients as two float4");

static bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace heads
}  // namespace sg2im

using namespace sg2im::heads;

extern "C" {

int sg2im_two_heads_supported(int k, int n1, int n2) {
  return k >= 4 && k % 4 == 0 && k <= MAXK && n1 >= 1 && n2 >= 1 && n1 + n2 <= MAXN;
}

int sg2im_two_heads_forward(const float* x, long long ldx, int rows, int k, const float* w1, const float* b1, int n1,
                            const float* w2, const float* b2, int n2, float* y1, long long ld1, float* y2,
                            long long ld2, hipStream_t stream) {
  if (!sg2im_two_heads_supported(k, n1, n2) || rows < 0 || !w1 || !w2 || ldx % 4 != 0 || ldx < k || ld1 < n1 || ld2 < n2 ||
      !al16(x) || !al16(w1) || !al16(w2))
    return SG2IM_ERR_ARG;
  if (rows == 0) return SG2IM_OK;
  if (!x || !y1 || !y2) return SG2IM_ERR_ARG;
  SG2IM_LAUNCH(two_heads_fwd_kernel, dim3((rows + RT - 1) / RT, (n1 + n2 + CT - 1) / CT), dim3(THREADS), (size_t)RT * k * sizeof(float), stream,
               x, ldx, rows, k, w1, b1, n1, w2, b2, n2, y1, ld1, y2, ld2);
  return hipGetLastError() == hipSuccess ? SG2IM_OK : SG2IM_ERR_HIP;
}

int sg2im_two_heads_backward_data(const float* g1, long long ldg1, const float* g2, long long ldg2, int rows, int k,
                                  const float* w1, int n1, const float* w2, int n2, float* dx, long long lddx,
                                  hipStream_t stream) {
  if (!sg2im_two_heads_supported(k, n1, n2) || rows < 0 || !w1 || !w2 || lddx % 4 != 0 || lddx < k || ldg1 < n1 ||
      ldg2 < n2 || !al16(dx) || !al16(w1) || !al16(w2))
    return SG2IM_ERR_ARG;
  if (rows == 0) return SG2IM_OK;
  if (!g1 || !g2 || !dx) return SG2IM_ERR_ARG;
  SG2IM_LAUNCH(two_heads_dgrad_kernel, dim3((rows + RT - 1) / RT, (k + KT - 1) / KT), dim3(THREADS), (size_t)(n1 + n2) * RT * sizeof(float),
               stream, g1, ldg1, g2, ldg2, rows, k, w1, n1, w2, n2, dx, lddx);
  return hipGetLastError() == hipSuccess ? SG2IM_OK : SG2IM_ERR_HIP;
}

}  // extern "C"
