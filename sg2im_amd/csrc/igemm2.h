// Second-generation implicit-GEMM loop for the stride-1 convolutions of the refinement network
// (reference sg2im/crn.py:41-47,79-86): operands travel global memory -> LDS DIRECTLY
// (buffer_load_dwordx4 ... lds) - no staging registers, no ds_write pass, no loader arithmetic on
// the data; rows that fall outside the image (the zero padding of the convolution) or outside the
// tensor get an out-of-range buffer offset and are zero-filled by the buffer bounds check.
//
// Why (round 2 measurements, profiles/r2_gemm_loop_sandbox.log, r2_bench_bf16_*.json): the
// register-staged loop of igemm.h spends MFMA time + loader time ADDITIVELY (its loader phase - ~170
// VALU + ~120 SALU + 6 ds_write + 8 loads per K chunk - is instruction-issue bound and does not hide
// under the other waves' MFMAs: with the MFMAs made 16x cheaper (bf16) the same kernels still need 3.0
// of their 6.2 ms).  This loop has ~25 instructions of loader per chunk.
//
// Requirements (host side checks them, everything else stays on igemm.h): fp32, stride 1, every
// source plain (no pending affine, no row gather), channels % 32 == 0, tensors < 2 GB.
// The activated tensors the refinement network feeds into its convolutions are therefore
// MATERIALISED (sg2im_affine_act_forward) when this path is on, instead of applying the previous
// layer's BatchNorm + LeakyReLU inside the loader.
//
// Layout: the LDS image of an operand whose reduction index is contiguous in memory (activations along
// channels, weight rows) is [rows][32 floats] WITHOUT padding - the DMA writes lane-linear, 64 lanes x
// 16 B = 8 rows per instruction - and XOR-swizzled in 16-byte pieces: piece q of row r sits at position
// q ^ (r & 7).  The swizzle is applied on the SOURCE address (lane j of row r fetches piece j ^ (r & 7))
// and undone by the fragment reads (ds_read_b128 of piece q at q ^ (r & 7)): 8 consecutive rows then hit
// 8 different bank groups.  The data-gradient's weight operand (reduction index = weight row) is
// [32][BN] as in memory, read with ds_read_b32 (conflict free without padding).
// Two LDS buffers, ONE barrier per K chunk: [barrier (chunk i landed, everybody done with chunk i-1)]
// [all fragment reads of chunk i] [issue the DMA of chunk i+1] [MFMAs of chunk i] - the fragment reads
// must precede the DMA issue: hipcc waits vmcnt(0) in front of any LDS read that follows an LDS-DMA.
#pragma once
#include "igemm.h"

namespace sg2im {

typedef __attribute__((address_space(3))) void* lds_dst_t;

struct Src2 { const float* p; unsigned bytes; int C, ld, up; };

struct Conv2Params {
  Src2 s0, s1;             // forward: the (<= 2) channel-concatenated sources; data gradient: s0 = dY
  int nsrc;
  const float* Wt;         // [Cout][KH][KW][Ctot]
  unsigned w_bytes;
  int Ctot, ldw;           // ldw = KH * KW * Ctot
  int NB, H, W;            // spatial size (input == output: stride 1, "same" geometry checked by the host)
  int KH, KW, pad;
  int M;                   // NB * H * W rows
  int N;                   // forward: Cout; data gradient: number of input channels produced (c_count)
  int c_begin;             // data gradient: first input channel
  int Kd;                  // data gradient: Cout (reduction channels per tap)
  int nch;                 // K chunks per tap
  int iters;
  Epi e;
};

constexpr unsigned kOOB = 0x80000000u;

// MODE 0: forward (B = weight rows, reduction index contiguous); MODE 1: data gradient (B[k = co][n = c])
template <int BM, int BN, int NW, int MODE>
__global__ __launch_bounds__(NW * 64) void conv2_kernel(const Conv2Params p) {
  extern __shared__ __attribute__((aligned(16))) float smem2[];
  constexpr int WGM = NW / 2;                              // waves along M (x 2 along N)
  constexpr int TM = BM / WGM / 32, TN = BN / 64;
  constexpr int AF = BM * BK, BF = BN * BK, STAGE = AF + BF;
  constexpr int NA = BM / NW / 8;                          // A DMA instructions per wave per chunk (8 rows each)
  constexpr int QB = BN / 4;                               // MODE 1: lanes per B row
  constexpr int NB8 = MODE == 0 ? BN / NW / 8 : (BK * QB / 64) / NW;   // B DMA instructions per wave per chunk
  static_assert(NA >= 1 && NB8 >= 1, "tile too small for the wave count");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN, split = blockIdx.z;
  const int per = (p.iters + p.e.nsplit - 1) / p.e.nsplit;
  const int it_begin = split * per, it_end = min(p.iters, it_begin + per);
  const int lr = lane >> 3;
  const int cswz = (lane & 7) ^ lr;                        // 16-byte piece of the row this lane fetches
  const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.s0.p, 0, p.s0.bytes, 0x27000);
  const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.nsrc > 1 ? p.s1.p : p.s0.p), 0,
                                                                      p.nsrc > 1 ? p.s1.bytes : p.s0.bytes, 0x27000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wt, 0, p.w_bytes, 0x27000);

  // A rows of this lane: r = (BM / NW) * wave + 8 t + lr  ->  (image, y, x) and the tap validity mask
  int an[NA], ahw[NA];
  unsigned amask[NA];
  const int taps = p.KH * p.KW;
  #pragma unroll
  for (int t = 0; t < NA; ++t) {
    const int m = m0 + (BM / NW) * wave + 8 * t + lr;
    unsigned mask = 0;
    int n = 0, ho = 0, wo = 0;
    if (m < p.M) {
      const int hw = p.H * p.W;
      n = m / hw;
      const int rem = m - n * hw;
      ho = rem / p.W; wo = rem - ho * p.W;
      for (int tap = 0; tap < taps; ++tap) {
        const int kh = tap / p.KW, kw = tap - kh * p.KW;
        // forward: input pixel (ho + kh - pad, wo + kw - pad); data gradient: output pixel (h + pad - kh, w + pad - kw)
        const int hi = MODE == 0 ? ho + kh - p.pad : ho + p.pad - kh;
        const int wi = MODE == 0 ? wo + kw - p.pad : wo + p.pad - kw;
        if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W) mask |= 1u << tap;
      }
    }
    an[t] = n; ahw[t] = ho | (wo << 16); amask[t] = mask;
  }
  // B rows / pieces of this lane
  unsigned bbase[NB8];
  bool bok[NB8];
  #pragma unroll
  for (int u = 0; u < NB8; ++u) {
    if (MODE == 0) {
      const int co = n0 + (BN / NW) * wave + 8 * u + lr;
      bok[u] = co < p.N;
      bbase[u] = (unsigned)(co * p.ldw);
    } else {
      // instruction q = NB8 * wave + u covers k rows (64 / QB) * q + lane / QB, columns 4 * (lane % QB)
      const int col = n0 + 4 * (lane % QB);
      bok[u] = col < p.N;
      bbase[u] = (unsigned)(p.c_begin + col);
    }
  }

  auto issue = [&](int it, int buf) {
    // chunk `it` -> (tap, source, channel block)
    const int tap = it / p.nch;
    int ch = it - tap * p.nch;
    const int kh = tap / p.KW, kw = tap - kh * p.KW;
    const int dh = MODE == 0 ? kh - p.pad : p.pad - kh, dw = MODE == 0 ? kw - p.pad : p.pad - kw;
    const int n0ch = MODE == 0 ? p.s0.C / BK : p.nch;
    const bool second = MODE == 0 && ch >= n0ch;
    if (second) ch -= n0ch;
    const int cb = ch * BK;
    const int C = second ? p.s1.C : p.s0.C, ld = second ? p.s1.ld : p.s0.ld, up = second ? p.s1.up : p.s0.up;
    (void)C;
    const int Hs = p.H >> up, Ws = p.W >> up;
    float* a_dst = smem2 + buf * STAGE + (BM / NW) * wave * BK;
    #pragma unroll
    for (int t = 0; t < NA; ++t) {
      const int hi = (ahw[t] & 0xffff) + dh, wi = (ahw[t] >> 16) + dw;
      const unsigned pix = (unsigned)((an[t] * Hs + (hi >> up)) * Ws + (wi >> up));
      const unsigned off = (amask[t] >> tap & 1u) ? (pix * (unsigned)ld + (unsigned)(cb + 4 * cswz)) * 4u : kOOB;
      if (second) __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, (lds_dst_t)(a_dst + 8 * t * BK), 16, off, 0, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(r0, (lds_dst_t)(a_dst + 8 * t * BK), 16, off, 0, 0, 0);
    }
    if (MODE == 0) {
      const int wcol = tap * p.Ctot + (second ? p.s0.C : 0) + cb + 4 * cswz;
      float* b_dst = smem2 + buf * STAGE + AF + (BN / NW) * wave * BK;
      #pragma unroll
      for (int u = 0; u < NB8; ++u) {
        const unsigned off = bok[u] ? (bbase[u] + (unsigned)wcol) * 4u : kOOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_dst_t)(b_dst + 8 * u * BK), 16, off, 0, 0, 0);
      }
    } else {
      float* b_dst = smem2 + buf * STAGE + AF;
      #pragma unroll
      for (int u = 0; u < NB8; ++u) {
        const int q = NB8 * wave + u;
        const int k = (64 / QB) * q + lane / QB;                 // reduction row (output channel cb + k)
        const bool ok = bok[u] && cb + k < p.Kd;
        const unsigned off = ok ? ((unsigned)((cb + k) * p.ldw + tap * p.Ctot) + bbase[u]) * 4u : kOOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_dst_t)(b_dst + 64 * q * 4), 16, off, 0, 0, 0);
      }
    }
  };

  f32x16 acc[TM][TN];
  #pragma unroll
  for (int a = 0; a < TM; ++a)
    #pragma unroll
    for (int b = 0; b < TN; ++b)
      #pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const int wm0 = (wave >> 1) * (BM / WGM), wn0 = (wave & 1) * (BN / 2);
  const int i_ = lane & 31, h = lane >> 5, sw = i_ & 7;

  if (it_begin < it_end) issue(it_begin, 0);
  #pragma unroll 1
  for (int it = it_begin; it < it_end; ++it) {
    const int cur = (it - it_begin) & 1;
    __syncthreads();                       // vmcnt(0): chunk `it` has landed; every wave is done reading chunk it-1
    const float* As = smem2 + cur * STAGE;
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
      if (MODE == 0) {
        const float* row = Bs + (wn0 + tn * 32 + i_) * BK;
        #pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 v = *reinterpret_cast<const float4*>(row + 4 * ((h + 2 * g) ^ sw));
          fb[tn][4 * g] = v.x; fb[tn][4 * g + 1] = v.y; fb[tn][4 * g + 2] = v.z; fb[tn][4 * g + 3] = v.w;
        }
      } else {
        #pragma unroll
        for (int s = 0; s < 16; ++s) fb[tn][s] = Bs[kperm(s, h) * BN + wn0 + tn * 32 + i_];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (it + 1 < it_end) issue(it + 1, cur ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    #pragma unroll
    for (int s = 0; s < 16; ++s)
      #pragma unroll
      for (int tm = 0; tm < TM; ++tm)
        #pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[tm][s], fb[tn][s], acc[tm][tn], 0, 0, 0);
  }
  // epilogue (C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5))
  const int j = lane & 31;
  const Epi& e = p.e;
  #pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int n = n0 + wn0 + tn * 32 + j;
    if (n >= p.N) continue;
    const float bv = (e.nsplit == 1 && e.bias) ? e.bias[n] : 0.f;
    #pragma unroll
    for (int tm = 0; tm < TM; ++tm)
      #pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm0 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m >= p.M) continue;
        float v = acc[tm][tn][r];
        if (e.nsplit > 1) {
          e.ws[((long long)split * p.M + m) * p.N + n] = v;
        } else {
          v = leaky(v + bv, e.slope);
          float* dst = e.C + (long long)m * e.ldc + n;
          if (e.accumulate) v += *dst;
          *dst = v;
        }
      }
  }
}

}  // namespace sg2im
