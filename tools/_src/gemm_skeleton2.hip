// Sandbox for the implicit-GEMM main-loop STRUCTURE on gfx950 (tools/gemm_skeleton2.py): the full
// per-chunk instruction mix of conv_fwd_kernel<128,64> (32 fp32 MFMAs, 12 fragment ds_read_b128,
// 6 ds_write_b128, 6 L2-resident global float4 loads, ~168 VALU in 8 independent chains, ~118 SALU in
// 4 chains, 2 barriers) arranged in different ways.  MODE:
//   0  phases in sequence (what csrc/igemm.h k_pipeline_d1 does): loads + loader math, MFMA block,
//      barrier, LDS stores, barrier
//   1  as 0, co-resident workgroups start 2048 x slot cycles apart (slot = HW wave id & 3)
//   2  as 0, s_setprio 3 around the MFMA block
//   3  loader math cut into 16 slices placed between the MFMA pairs (software interleave inside a wave)
//   4  512-thread ping-pong: two halves one barrier apart (k_pipeline_pp)
//   5  as 3, plus the LDS stores of the previous chunk's registers also sliced into the MFMA block
//      (double LDS image, ONE barrier per chunk)
#include <hip/hip_runtime.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int MLD = 36;
constexpr int IMG = (128 + 64) * MLD;      // floats per LDS image

struct Fill { unsigned v[8]; int s[4]; };

__device__ __forceinline__ void fill_slice(Fill& f, int it, int j) {
  // ~10 VALU (8 chains) + ~7 SALU (4 chains)
  #pragma unroll
  for (int c = 0; c < 8; ++c) f.v[c] = f.v[c] * 1664525u + (unsigned)(c + j);
  f.v[0] ^= f.v[4]; f.v[1] += f.v[5];
  #pragma unroll
  for (int c = 0; c < 4; ++c) f.s[c] = f.s[c] * 3 + it + c;
  f.s[0] ^= f.s[2]; f.s[1] += f.s[3]; f.s[2] += j;
}

template <int MODE>
__global__ __launch_bounds__(MODE == 4 ? 512 : 256) void skel2_kernel(const float* __restrict__ g, float* __restrict__ sink,
                                                                       int iters) {
  extern __shared__ __attribute__((aligned(16))) float smem_all[];
  const int half = MODE == 4 ? (int)(threadIdx.x >> 8) : 0;
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
  float* smem = smem_all + half * IMG;
  const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 32;
  f32x16 acc0, acc1;
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  for (int i = tid; i < IMG * (MODE == 5 ? 2 : 1); i += 256) smem[i] = (float)(i & 7);
  Fill f;
  for (int c = 0; c < 8; ++c) f.v[c] = tid * (2 * c + 1);
  for (int c = 0; c < 4; ++c) f.s[c] = blockIdx.x + c;
  const float* gp = g + (size_t)((blockIdx.x * 2 + half) & 63) * 8192 + tid * 4;     // 64 x 32 KB regions: L2 resident
  float4 r[6];
  #pragma unroll
  for (int k = 0; k < 6; ++k) r[k] = *reinterpret_cast<const float4*>(gp + k * 1024);
  const int i_ = lane & 31, h = lane >> 5;
  const int col4 = tid & 7, r0 = tid >> 3;
  if (MODE == 1) {
    const unsigned slot = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 4) & 3u;
    for (unsigned q = 0; q < slot; ++q) __builtin_amdgcn_s_sleep(32);
  }
  __syncthreads();
  if (MODE == 4 && half) __syncthreads();

  auto read_frags = [&](const float* img, float (&a0)[16], float (&a1)[16], float (&b0)[16]) {
    const float* ra0 = img + (wm0 + i_) * MLD + 4 * h;
    const float* ra1 = img + (wm0 + 32 + i_) * MLD + 4 * h;
    const float* rb0 = img + 128 * MLD + (wn0 + i_) * MLD + 4 * h;
    #pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v0 = *reinterpret_cast<const float4*>(ra0 + 8 * q);
      const float4 v1 = *reinterpret_cast<const float4*>(ra1 + 8 * q);
      const float4 w0 = *reinterpret_cast<const float4*>(rb0 + 8 * q);
      a0[4 * q] = v0.x; a0[4 * q + 1] = v0.y; a0[4 * q + 2] = v0.z; a0[4 * q + 3] = v0.w;
      a1[4 * q] = v1.x; a1[4 * q + 1] = v1.y; a1[4 * q + 2] = v1.z; a1[4 * q + 3] = v1.w;
      b0[4 * q] = w0.x; b0[4 * q + 1] = w0.y; b0[4 * q + 2] = w0.z; b0[4 * q + 3] = w0.w;
    }
  };
  auto store_row = [&](float* img, int k, const float4& v) {
    *reinterpret_cast<float4*>(img + (r0 + 32 * k) * MLD + 4 * col4) = v;
  };

  for (int it = 0; it < iters; ++it) {
    float a0[16], a1[16], b0[16];
    if (MODE <= 2 || MODE == 4) {
      if (MODE != 4) {
        #pragma unroll
        for (int k = 0; k < 6; ++k) r[k] = *reinterpret_cast<const float4*>(gp + ((it + k) & 7) * 1024);
        #pragma unroll
        for (int j = 0; j < 16; ++j) fill_slice(f, it, j);
        __builtin_amdgcn_sched_barrier(0);
      }
      read_frags(smem, a0, a1, b0);
      if (MODE == 2) __builtin_amdgcn_s_setprio(3);
      #pragma unroll
      for (int s = 0; s < 16; ++s) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b0[s], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b0[s], acc1, 0, 0, 0);
      }
      if (MODE == 2) __builtin_amdgcn_s_setprio(0);
      __syncthreads();
      #pragma unroll
      for (int k = 0; k < 6; ++k) { float4 v = r[k]; v.x += (float)((f.v[k] ^ (unsigned)f.s[k & 3]) & 1u); store_row(smem, k, v); }
      if (MODE == 4) {
        #pragma unroll
        for (int k = 0; k < 6; ++k) r[k] = *reinterpret_cast<const float4*>(gp + ((it + k) & 7) * 1024);
        #pragma unroll
        for (int j = 0; j < 16; ++j) fill_slice(f, it, j);
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
    } else if (MODE == 3) {
      float4 rn[6];
      #pragma unroll
      for (int k = 0; k < 6; ++k) rn[k] = *reinterpret_cast<const float4*>(gp + ((it + k) & 7) * 1024);
      read_frags(smem, a0, a1, b0);
      #pragma unroll
      for (int s = 0; s < 16; ++s) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b0[s], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b0[s], acc1, 0, 0, 0);
        fill_slice(f, it, s);
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
      #pragma unroll
      for (int k = 0; k < 6; ++k) { float4 v = rn[k]; v.x += (float)((f.v[k] ^ (unsigned)f.s[k & 3]) & 1u); store_row(smem, k, v); }
      __syncthreads();
    } else {   // MODE 5
      float* cur = smem + (it & 1) * IMG;
      float* nxt = smem + ((it + 1) & 1) * IMG;
      float4 rn[6];
      #pragma unroll
      for (int k = 0; k < 6; ++k) rn[k] = *reinterpret_cast<const float4*>(gp + ((it + k) & 7) * 1024);
      read_frags(cur, a0, a1, b0);
      #pragma unroll
      for (int s = 0; s < 16; ++s) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b0[s], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b0[s], acc1, 0, 0, 0);
        fill_slice(f, it, s);
        if (s >= 2 && s < 14 && (s & 1) == 0) {          // the registers loaded one chunk ago -> the other LDS image
          const int k = (s - 2) >> 1;
          float4 v = r[k]; v.x += (float)((f.v[k] ^ (unsigned)f.s[k & 3]) & 1u); store_row(nxt, k, v);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      #pragma unroll
      for (int k = 0; k < 6; ++k) r[k] = rn[k];
      __syncthreads();
    }
  }
  if (MODE == 4 && !half) __syncthreads();
  float s = 0.f;
  for (int q = 0; q < 16; ++q) s += acc0[q] + acc1[q];
  unsigned fx = 0; for (int c = 0; c < 8; ++c) fx ^= f.v[c];
  if (s == 12345.678f) sink[0] = s + (float)fx + (float)(f.s[0] + f.s[1] + f.s[2] + f.s[3]);
}

template <int MODE>
static int launch2(const float* g, float* sink, int blocks, int iters, size_t extra_lds, hipStream_t st) {
  const size_t lds = IMG * sizeof(float) * (MODE == 4 || MODE == 5 ? 2 : 1) + extra_lds;
  hipFuncSetAttribute(reinterpret_cast<const void*>(skel2_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((skel2_kernel<MODE>), dim3(blocks), dim3(MODE == 4 ? 512 : 256), lds, st, g, sink, iters);
  return (int)hipGetLastError();
}

extern "C" int skel2_launch(int mode, const float* g, float* sink, int blocks, int iters, long long extra_lds, hipStream_t st) {
  switch (mode) {
    case 0: return launch2<0>(g, sink, blocks, iters, (size_t)extra_lds, st);
    case 1: return launch2<1>(g, sink, blocks, iters, (size_t)extra_lds, st);
    case 2: return launch2<2>(g, sink, blocks, iters, (size_t)extra_lds, st);
    case 3: return launch2<3>(g, sink, blocks, iters, (size_t)extra_lds, st);
    case 4: return launch2<4>(g, sink, blocks, iters, (size_t)extra_lds, st);
    case 5: return launch2<5>(g, sink, blocks, iters, (size_t)extra_lds, st);
  }
  return -1;
}
