// Prototype of the "v2" implicit-GEMM forward: operands go global -> LDS directly
// (buffer_load_dwordx4 ... lds: no staging registers, no ds_write, out-of-image rows zero-filled by the
// buffer bounds check), unpadded XOR-swizzled LDS image, two LDS buffers, ONE barrier per K chunk.
// Plain NHWC fp32 input (no pending affine), one source, stride 1.  tools/conv_v2.py compares it with
// sg2im_conv2d_forward on the refinement-network shapes.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BK = 32;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct V2Params {
  const float* X; const float* Wt; const float* bias; float* out;
  int C, ldx, Cout, ldw, NB, H, W, KH, KW, pad, M, nch;
  unsigned x_bytes, w_bytes;
  float slope;
  int dbg;       // timing-only: 1 = no DMA inside the loop, 2 = every DMA lane reads the same 1 KB (cache hits), 4 = no fragment reads
};

__device__ __forceinline__ float leaky(float v, float slope) { return v > 0.f ? v : v * slope; }

// NW waves per workgroup arranged (NW/2) x 2; wave tile (BM / (NW/2)) x (BN / 2)
template <int BM, int BN, int NW = 4>
__global__ __launch_bounds__(NW * 64) void conv_fwd_v2_kernel(const V2Params p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int TM = BM / (NW / 2) / 32, TN = BN / 64;
  constexpr int AF = BM * BK, BF = BN * BK, STAGE = AF + BF;
  constexpr int NA = BM / NW / 8, NB_ = BN / NW / 8;    // glds instructions per wave per chunk (8 rows each)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int lr = lane >> 3;                              // row within an 8-row glds group
  const int cswz = (lane & 7) ^ lr;                      // global 16-byte chunk this lane fetches (swizzle on the source)
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.X, 0, p.x_bytes, 0x27000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wt, 0, p.w_bytes, 0x27000);
  constexpr unsigned OOB = 0x80000000u;

  // A rows of this lane: r = (BM/4) * wave + 8 t + lr
  unsigned abase[NA], amask[NA];
  #pragma unroll
  for (int t = 0; t < NA; ++t) {
    const int m = m0 + (BM / NW) * wave + 8 * t + lr;
    unsigned mask = 0, base = 0;
    if (m < p.M) {
      const int hw = p.H * p.W;
      const int n = m / hw, rem = m - n * hw;
      const int ho = rem / p.W, wo = rem - ho * p.W;
      base = (unsigned)(((n * p.H + ho) * p.W + wo) * p.ldx);
      for (int kh = 0; kh < p.KH; ++kh)
        for (int kw = 0; kw < p.KW; ++kw) {
          const int hi = ho + kh - p.pad, wi = wo + kw - p.pad;
          if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W) mask |= 1u << (kh * p.KW + kw);
        }
    }
    abase[t] = base; amask[t] = mask;
  }
  unsigned bbase[NB_];
  bool bok[NB_];
  #pragma unroll
  for (int u = 0; u < NB_; ++u) {
    const int co = n0 + (BN / NW) * wave + 8 * u + lr;
    bok[u] = co < p.Cout;
    bbase[u] = (unsigned)(co * p.ldw);
  }
  const int iters = p.KH * p.KW * p.nch;

  auto issue = [&](int it, int buf) {
    const int tap = it / p.nch, cb = (it - tap * p.nch) * BK;
    const int kh = tap / p.KW, kw = tap - kh * p.KW;
    const int shift = (p.dbg & 2) ? 4 * cswz : ((kh - p.pad) * p.W + (kw - p.pad)) * p.ldx + cb + 4 * cswz;
    float* a_dst = smem + buf * STAGE + (BM / NW) * wave * BK;
    #pragma unroll
    for (int t = 0; t < NA; ++t) {
      const unsigned off = (p.dbg & 2) ? (unsigned)(shift + 32 * lr) * 4u : (amask[t] >> tap & 1u) ? (abase[t] + (unsigned)shift) * 4u : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(a_dst + 8 * t * BK), 16, off, 0, 0, 0);
    }
    const int wcol = tap * p.C + cb + 4 * cswz;
    float* b_dst = smem + buf * STAGE + AF + (BN / NW) * wave * BK;
    #pragma unroll
    for (int u = 0; u < NB_; ++u) {
      const unsigned off = (p.dbg & 2) ? (unsigned)(4 * cswz + 32 * lr) * 4u : bok[u] ? (bbase[u] + (unsigned)wcol) * 4u : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(b_dst + 8 * u * BK), 16, off, 0, 0, 0);
    }
  };

  f32x16 acc[TM][TN];
  #pragma unroll
  for (int a = 0; a < TM; ++a)
    #pragma unroll
    for (int b = 0; b < TN; ++b)
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const int wm0 = (wave >> 1) * (BM / (NW / 2)), wn0 = (wave & 1) * (BN / 2);
  const int i_ = lane & 31, h = lane >> 5, sw = i_ & 7;

  issue(0, 0);
  #pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    __syncthreads();                       // (vmcnt(0): chunk `it` has landed; every wave is done reading chunk it-1)
    const float* As = smem + (it & 1) * STAGE;
    const float* Bs = As + AF;
    float fa[TM][16], fb[TN][16];
    #pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const float* row = As + (wm0 + tm * 32 + i_) * BK;
      #pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 v = *reinterpret_cast<const float4*>(row + 4 * ((h + 2 * g) ^ sw));
        fa[tm][4 * g] = v.x; fa[tm][4 * g + 1] = v.y; fa[tm][4 * g + 2] = v.z; fa[tm][4 * g + 3] = v.w;
      }
    }
    #pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const float* row = Bs + (wn0 + tn * 32 + i_) * BK;
      #pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 v = *reinterpret_cast<const float4*>(row + 4 * ((h + 2 * g) ^ sw));
        fb[tn][4 * g] = v.x; fb[tn][4 * g + 1] = v.y; fb[tn][4 * g + 2] = v.z; fb[tn][4 * g + 3] = v.w;
      }
    }
    // ALL fragment reads first, THEN the next chunk's LDS-DMA, then the MFMAs: hipcc waits vmcnt(0) in
    // front of any ds_read that follows an LDS-DMA (it cannot prove the two do not alias)
    __builtin_amdgcn_sched_barrier(0);
    if (it + 1 < iters && !(p.dbg & 1)) issue(it + 1, (it + 1) & 1);
    __builtin_amdgcn_sched_barrier(0);
    #pragma unroll
    for (int s = 0; s < 16; ++s)
      #pragma unroll
      for (int tm = 0; tm < TM; ++tm)
        #pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[tm][s], fb[tn][s], acc[tm][tn], 0, 0, 0);
  }
  // epilogue
  const int j = lane & 31;
  #pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int n = n0 + wn0 + tn * 32 + j;
    if (n >= p.Cout) continue;
    const float bv = p.bias ? p.bias[n] : 0.f;
    #pragma unroll
    for (int tm = 0; tm < TM; ++tm)
      #pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm0 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m < p.M) p.out[(long long)m * p.Cout + n] = leaky(acc[tm][tn][r] + bv, p.slope);
      }
  }
}

extern "C" int conv_fwd_v2(const float* X, int C, int ldx, const float* Wt, int Cout, const float* bias, float* out,
                           int NB, int H, int W, int K, int pad, float slope, int tile, int dbg, hipStream_t st) {
  if (C % BK) return 1;
  V2Params p;
  p.dbg = dbg;
  p.X = X; p.Wt = Wt; p.bias = bias; p.out = out; p.C = C; p.ldx = ldx; p.Cout = Cout; p.ldw = K * K * C;
  p.NB = NB; p.H = H; p.W = W; p.KH = K; p.KW = K; p.pad = pad; p.M = NB * H * W; p.nch = C / BK; p.slope = slope;
  p.x_bytes = (unsigned)((size_t)NB * H * W * ldx * 4); p.w_bytes = (unsigned)((size_t)Cout * p.ldw * 4);
  if (tile == 0) {
    const size_t lds = 2 * (128 + 64) * BK * 4;
    hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fwd_v2_kernel<128, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((conv_fwd_v2_kernel<128, 64>), dim3((Cout + 63) / 64, (p.M + 127) / 128), dim3(256), lds, st, p);
  } else if (tile == 2) {
    const size_t lds = 2 * (256 + 64) * BK * 4;
    hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fwd_v2_kernel<256, 64, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((conv_fwd_v2_kernel<256, 64, 8>), dim3((Cout + 63) / 64, (p.M + 255) / 256), dim3(512), lds, st, p);
  } else if (tile == 3) {
    const size_t lds = 2 * (256 + 128) * BK * 4;
    hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fwd_v2_kernel<256, 128, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((conv_fwd_v2_kernel<256, 128, 8>), dim3((Cout + 127) / 128, (p.M + 255) / 256), dim3(512), lds, st, p);
  } else {
    const size_t lds = 2 * (128 + 128) * BK * 4;
    hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fwd_v2_kernel<128, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((conv_fwd_v2_kernel<128, 128>), dim3((Cout + 127) / 128, (p.M + 127) / 128), dim3(256), lds, st, p);
  }
  return (int)hipGetLastError();
}
