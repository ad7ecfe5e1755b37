// The whole GraphTripleConv STACK (reference sg2im/graph.py:56-120 per layer, :123-144 the net; model.py:136-140)
// as ONE persistent launch per direction.
//
// Why: a layer is four small dependent GEMMs (M = a few hundred triples / objects) plus the CSR pool; as separate
// launches inside the multi-lane training graph every dependent launch costs ~5-10 us of dispatch latency, and the
// five layers (45 launches) took 0.48 ms at 7 TFLOP/s with the chip otherwise idle (profiles/r3_schedule_marks.txt).
// Here <= 256 resident workgroups (one per CU) walk the stages of all layers; two consecutive stages are separated by
// a grid barrier (XCD-hierarchical, MI355X_MICROARCH.md "barrier-xcd": per-XCC arrival counter -> the XCC's last
// arriver releases (buffer_wbl2) and arrives on the top counter -> acquires -> publishes the XCC's generation; every
// workgroup acquires (buffer_inv) before it reads what other CUs wrote).
//
// Forward stages of a layer (5 barriers):
//   A  h1     = relu([obj[s] | pred | obj[o]] W1a^T + b1a)       gather + concat in the operand loader (graph.py:73-83)
//   B  new_t  = relu(h1 W1b^T + b1b)                              (graph.py:83-89)
//   C  pooled = CSR pool of new_t (graph.py:92-114)               a wavefront per object row, the reference's
//                                                                 accumulation order (bit-exact, the arithmetic of
//                                                                 segment_sum_kernel)
//   D  h2     = relu(pooled W2a^T + b2a)
//   E  new_obj = relu(h2 W2b^T + b2b)                             (graph.py:118)
// Backward stages of a layer, last layer first (5 barriers; weight / bias gradients ride in the stage that has their
// operands, as extra tiles):
//   P1 dp3     = ((g_obj * relu'(new_obj)) W2b) * relu'(h2)       + dW2b, db2b
//   P2 dpooled = dp3 W2a                                          + dW2a, db2a
//   P3 dp1     = (dnt W1b) * relu'(h1),  dnt = [dpooled[s] / n_s | g_pred | dpooled[o] / n_o] * relu'(new_t) built in
//                the operand loader (the backward of the pool + concat, graph.py:98-114) and written out once
//   P4 d_triple = dp1 W1a                                         + dW1a, db1a, db1b
//   P5 d_obj   = CSR sum of d_triple's subject / object blocks    + dW1b      (d_obj, d_triple[:, Din:2Din] are the next
//                                                                              layer's g_obj, g_pred)
//
// GEMM tiling for tiny M: a workgroup owns a 32 x (32 NB) output tile and its four wavefronts SPLIT K between them
// (wave w takes the 32-wide K chunks w, w+4, ...), so a 32 x 64 x 512 tile is 128 MFMAs per wave instead of 512 on
// one; the four partial tiles are added in a fixed order through LDS (deterministic).  Waves share nothing inside the
// K loop: each stages ITS chunks - coalesced 16-byte global loads, 8 lanes per 128-byte row - into a wave-private LDS
// image and reads MFMA fragments back, so there is no workgroup barrier in the main loop; all chunks of a tile
// (<= 4 per wave for K <= 512) are in flight at once.  An operand whose reduction index is contiguous in memory
// (activations, nn.Linear weights in forward) is staged "m-major" [32 rows][36] and read with ds_read_b128 (the k
// permutation of igemm.h); one whose reduction index is the row (weights in the data gradients, both operands of a
// weight gradient) "k-major" [32 k][40] and read with ds_read_b32 - both conflict free.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#define SG2IM_GEMM_TU            // (a launch of this file counts as an implicit-GEMM-family launch)
#include "launch_count.h"
#include "gcn_persist.h"
#include "sg2im_hip.h"

namespace sg2im {
namespace gcn {

constexpr int kThreads = 256;
constexpr int kLdM = 36;                           // m-major image: floats between rows
constexpr int kLdK = 40;                           // k-major image: floats between k rows (4 * 40 = 32 mod 64)
constexpr int kImgFloats = 32 * kLdK;              // one staged 32 x 32 chunk (either form)
constexpr int kMaxNB = 2;
constexpr int kStageFloats = 4 * (1 + kMaxNB) * kImgFloats;        // 4 waves x (A + NB B blocks)
constexpr int kRedLd = 32 * kMaxNB + 8;            // 4 * kRedLd = 32 (mod 64): the two lane halves on disjoint banks
constexpr int kRedFloats = 4 * 32 * kRedLd;
constexpr size_t kLdsBytes = sizeof(float) * (kStageFloats + kRedFloats);

typedef float f32x16 __attribute__((ext_vector_type(16)));
// native 4-vector (HIP's float4 is a struct whose copies are emitted as memcpy, which keeps staging arrays in scratch)
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4f zero4() { return (v4f)(0.f); }
__device__ __forceinline__ v4f ld4(const float* p) { return *reinterpret_cast<const v4f*>(p); }
__device__ __forceinline__ void st4(float* p, const v4f& v) { *reinterpret_cast<v4f*>(p) = v; }
// x * relu'(y)
__device__ __forceinline__ v4f mask4(const v4f& x, const v4f& y) {
  v4f r;
  r.x = y.x > 0.f ? x.x : 0.f; r.y = y.y > 0.f ? x.y : 0.f; r.z = y.z > 0.f ? x.z : 0.f; r.w = y.w > 0.f ? x.w : 0.f;
  return r;
}

// ----------------------------------------------------------------------------------------------------------------
// grid barrier
// ----------------------------------------------------------------------------------------------------------------
// sync words (zeroed by the launcher before EVERY launch - a memset node when captured):
constexpr int kFlat = 0, kTop = 32, kError = 64, kMembers = 96, kXcc = 256, kXccStride = 64, kGenOff = 32;
// diagnostics: workgroup 0 stores the 100 MHz device clock at kernel start, before and after every barrier and at the
// end as 64-bit words from word kStamps on (sg2im_gconv_stack_stamps reads them from a host copy of the area)
constexpr int kStamps = 1536, kMaxStamps = 250;
// STICKY error count: the ONLY word of the area the per-launch memset leaves alone (the launcher clears the first
// kStickyWord words).  kError aborts the spins of the launch that timed out and is gone with the next launch's
// memset; this word keeps counting, so a grid that was not fully resident (another process on the GPU, a second
// persistent kernel on another stream) cannot go unnoticed: sg2im_gconv_stack_status() reports it and
// sg2im_amd.trainer polls it wherever it synchronises with the host anyway.
constexpr int kStickyWord = 2040;
static_assert(kStamps + 2 * kMaxStamps <= kStickyWord, "the stamps end below the sticky word");
constexpr unsigned long long kSpinTimeout = 200ull * 1000 * 100;      // 200 ms of the 100 MHz clock

struct Sync {
  unsigned* w;
  unsigned xcc, members, nxcc, epoch;
  int nstamp;
  bool censused;
};

__device__ __forceinline__ void stamp(Sync& sy) {
  if (blockIdx.x == 0 && sy.nstamp < kMaxStamps)
    reinterpret_cast<unsigned long long*>(sy.w + kStamps)[sy.nstamp++] = wall_clock64();
}

// every shared word is accessed as a GLOBAL (not flat) agent-scope atomic: relaxed sc1 loads / device-scope RMWs
typedef __attribute__((address_space(1))) unsigned gu32;
__device__ __forceinline__ unsigned ld_agent(unsigned* p) { return __hip_atomic_load((gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(unsigned* p, unsigned v) { __hip_atomic_store((gu32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned add_agent(unsigned* p, unsigned v) {
  return __hip_atomic_fetch_add((gu32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// every spin is bounded: a grid that is not fully resident (or a lost arrival) ends in an error word instead of a hang
__device__ __noinline__ bool spin_ge(unsigned* p, unsigned target, unsigned* err) {
  const unsigned long long t0 = wall_clock64();
  unsigned n = 0;
  for (;;) {
    if (ld_agent(p) >= target) return true;
    __builtin_amdgcn_s_sleep(1);
    if ((++n & 63) == 0) {
      if (ld_agent(err) != 0) return false;
      if (wall_clock64() - t0 > kSpinTimeout) {
        st_agent(err, 1u);
        add_agent(err - kError + kStickyWord, 1u);
        return false;
      }
    }
  }
}

// thread 0 only, at kernel start: join the census of its XCC (which workgroups share an L2 is only known at run time:
// the block -> XCD placement is not contractual) and announce itself on the flat counter
__device__ __forceinline__ void sync_begin(Sync& sy, unsigned* words) {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  sy.w = words; sy.xcc = x & 15u; sy.members = 0; sy.nxcc = 0; sy.epoch = 0; sy.censused = false; sy.nstamp = 0;
  add_agent(words + kMembers + sy.xcc, 1u);
  add_agent(words + kFlat, 1u);
  stamp(sy);
}

// Called by EVERY thread of EVERY workgroup the same number of times.  On return everything any workgroup stored
// before its call is visible to plain loads of every thread.
__device__ __forceinline__ void grid_barrier(Sync& sy) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's stores have reached the L2
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned* const w = sy.w;
    unsigned* const err = w + kError;
    stamp(sy);
    if (!sy.censused) {
      // the census is final once every workgroup has STARTED (long ago, by the first barrier): no data involved
      spin_ge(w + kFlat, gridDim.x, err);
      unsigned n = 0;
      for (int i = 0; i < 16; ++i) n += ld_agent(w + kMembers + i) != 0 ? 1u : 0u;
      sy.members = ld_agent(w + kMembers + sy.xcc);
      sy.nxcc = n;
      sy.censused = true;
    }
    const unsigned e = ++sy.epoch;
    unsigned* const cnt = w + kXcc + kXccStride * sy.xcc;
    unsigned* const gen = cnt + kGenOff;
    const unsigned old = add_agent(cnt, 1u);
    if (old + 1u == sy.members * e) {
      // last arriver of this XCC: every member's stores are in this XCD's L2 - write it back, then meet the others
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      add_agent(w + kTop, 1u);
      spin_ge(w + kTop, sy.nxcc * e, err);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      st_agent(gen, e);
    } else {
      spin_ge(gen, e, err);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    stamp(sy);
  }
  __syncthreads();
}

// ----------------------------------------------------------------------------------------------------------------
// operand loaders.  A staged chunk is 32 rows x 32 floats; lane -> the 16-byte piece c4 = lane & 7 of rows
// r8 + 8 j, j = 0..3 (8 lanes cover a row's 128 bytes).  For an m-major operand "row" is the tile row / column and
// the 32 floats are K chunk c; for a k-major operand "row" is the reduction index 32 c + r and the 32 floats are the
// tile's rows / columns.  Addresses are 32-bit element offsets off a wave-uniform base (the launcher checks every
// operand has < 2^31 elements); loaders are functors with always-inlined members (a lambda's operator() or a
// ternary over array ELEMENTS of a by-reference struct ends up indexing scratch).
// ----------------------------------------------------------------------------------------------------------------
struct Lane { int lane, wave, r8, c4; };
__device__ __forceinline__ Lane my_lane() {
  Lane L; L.lane = threadIdx.x & 63; L.wave = threadIdx.x >> 6; L.r8 = L.lane >> 3; L.c4 = L.lane & 7; return L;
}
template <typename T> __device__ __forceinline__ T sel3(int s, T a, T b, T c) { return s == 0 ? a : s == 1 ? b : c; }

// m-major: X[row0 + r][32 c + ..], rows clamped to [0, nrows)
struct RowsM {
  const float* X; int off[4];
  __device__ __forceinline__ void init(const float* X_, int ld, int row0, int nrows, const Lane& L) {
    X = X_;
    #pragma unroll
    for (int j = 0; j < 4; ++j) off[j] = min(row0 + L.r8 + 8 * j, nrows - 1) * ld + 4 * L.c4;
  }
  __device__ __forceinline__ void chunk(int c, v4f (&d)[4]) const {
    #pragma unroll
    for (int j = 0; j < 4; ++j) d[j] = ld4(X + off[j] + 32 * c);
  }
};

// m-major, masked: (G * relu'(Y))[row0 + r][32 c + ..]; G == nullptr: zeros
struct RowsMMasked {
  const float* G; const float* Y; int off[4];
  __device__ __forceinline__ void init(const float* G_, const float* Y_, int ld, int row0, int nrows, const Lane& L) {
    G = G_; Y = Y_;
    #pragma unroll
    for (int j = 0; j < 4; ++j) off[j] = min(row0 + L.r8 + 8 * j, nrows - 1) * ld + 4 * L.c4;
  }
  __device__ __forceinline__ void chunk(int c, v4f (&d)[4]) const {
    if (!G) {
      #pragma unroll
      for (int j = 0; j < 4; ++j) d[j] = zero4();
      return;
    }
    v4f g[4], y[4];
    #pragma unroll
    for (int j = 0; j < 4; ++j) { g[j] = ld4(G + off[j] + 32 * c); y[j] = ld4(Y + off[j] + 32 * c); }
    #pragma unroll
    for (int j = 0; j < 4; ++j) d[j] = mask4(g[j], y[j]);
  }
};

// the (subject, object) ids of a lane's four rows of a 32-triple block: the same in every layer, so a workgroup that
// keeps its row block from layer to layer (the usual case: identical stage shapes) fetches them once
struct TripleIdx { int m0; int s[4], o[4]; };
__device__ __forceinline__ void fetch_triple_idx(TripleIdx& ix, const long long* s_idx, const long long* o_idx, int m0, int M,
                                                 const Lane& L) {
  if (ix.m0 == m0) return;         // (workgroup-uniform)
  ix.m0 = m0;
  #pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = min(m0 + L.r8 + 8 * j, M - 1);
    ix.s[j] = (int)s_idx[m]; ix.o[j] = (int)o_idx[m];
  }
}

// m-major: [obj[s[t]] | pred[t] | obj[o[t]]], each `din` wide (graph.py:73-82)
struct RowsTriple {
  const float* obj; const float* pred;
  int os[4], op[4], oo[4];
  int cpd;                         // chunks per source = din / 32
  __device__ __forceinline__ void init(const float* obj_, int ld_obj, const float* pred_, int ld_pred, int din, int m0, int M,
                                       const TripleIdx& ix, const Lane& L) {
    obj = obj_; pred = pred_; cpd = din >> 5;
    #pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = min(m0 + L.r8 + 8 * j, M - 1);
      os[j] = ix.s[j] * ld_obj + 4 * L.c4;
      op[j] = m * ld_pred + 4 * L.c4;
      oo[j] = ix.o[j] * ld_obj + 4 * L.c4;
    }
  }
  __device__ __forceinline__ void chunk(int c, v4f (&d)[4]) const {
    const int src = c / cpd, cc = c - src * cpd;           // (wave-uniform)
    const float* base = src == 1 ? pred : obj;
    #pragma unroll
    for (int j = 0; j < 4; ++j) d[j] = ld4(base + sel3(src, os[j], op[j], oo[j]) + 32 * cc);
  }
};

// m-major: dnt[t] = [dpooled[s[t]] / n_s | g_pred[t] | dpooled[o[t]] / n_o] * relu'(new_t[t])  (the backward of the
// pool + concat of graph.py:98-114; same operations per element as gconv_pool_bwd_kernel: IEEE divide, mask).
// out != nullptr: the tile also writes the rows it builds (the weight gradient of net1's second layer reads them).
struct RowsDnt {
  const float* dp; const float* gp; const float* nt; float* out;
  // per-row state kept small (this loader sets the register high-water mark of the backward kernels): the triple's
  // subject / object rows and its own (clamped) row; the element offsets are re-formed per chunk (one mad each)
  int rs[4], ro[4], rm[4];
  float ds[4], dv[4];
  int H, ld_gp, ld_nt, c4x4, M, m0r;
  int cH, cHD;                     // chunk boundaries: [0, cH) subject block, [cH, cHD) predicate block, then object block
  __device__ __forceinline__ void init(const float* dpooled, int H_, const float* g_pred, int ld_gp_, const float* new_t, int ld_nt_,
                                       int Dout, const int* row_ptr, bool average, float* out_, int m0, int M_, const TripleIdx& ix,
                                       const Lane& L) {
    dp = dpooled; gp = g_pred; nt = new_t; out = out_; cH = H_ >> 5; cHD = (H_ + Dout) >> 5;
    H = H_; ld_gp = ld_gp_; ld_nt = ld_nt_; c4x4 = 4 * L.c4; M = M_; m0r = m0 + L.r8;
    #pragma unroll
    for (int j = 0; j < 4; ++j) {
      rs[j] = ix.s[j]; ro[j] = ix.o[j];
      rm[j] = min(m0r + 8 * j, M - 1);
      ds[j] = 1.f; dv[j] = 1.f;
      if (average) {
        ds[j] = (float)max(1, row_ptr[ix.s[j] + 1] - row_ptr[ix.s[j]]);
        dv[j] = (float)max(1, row_ptr[ix.o[j] + 1] - row_ptr[ix.o[j]]);
      }
    }
  }
  __device__ __forceinline__ void chunk(int c, v4f (&d)[4]) const {
    v4f y[4];
    #pragma unroll
    for (int j = 0; j < 4; ++j) y[j] = ld4(nt + rm[j] * ld_nt + c4x4 + 32 * c);
    if (c < cH) {
      #pragma unroll
      for (int j = 0; j < 4; ++j) d[j] = ld4(dp + rs[j] * H + c4x4 + 32 * c);
      #pragma unroll
      for (int j = 0; j < 4; ++j) { d[j].x = d[j].x / ds[j]; d[j].y = d[j].y / ds[j]; d[j].z = d[j].z / ds[j]; d[j].w = d[j].w / ds[j]; }
    } else if (c < cHD) {
      #pragma unroll
      for (int j = 0; j < 4; ++j) d[j] = gp ? ld4(gp + rm[j] * ld_gp + c4x4 + 32 * (c - cH)) : zero4();
    } else {
      #pragma unroll
      for (int j = 0; j < 4; ++j) d[j] = ld4(dp + ro[j] * H + c4x4 + 32 * (c - cHD));
      #pragma unroll
      for (int j = 0; j < 4; ++j) { d[j].x = d[j].x / dv[j]; d[j].y = d[j].y / dv[j]; d[j].z = d[j].z / dv[j]; d[j].w = d[j].w / dv[j]; }
    }
    #pragma unroll
    for (int j = 0; j < 4; ++j) {
      d[j] = mask4(d[j], y[j]);
      if (out && m0r + 8 * j < M) st4(out + rm[j] * ld_nt + c4x4 + 32 * c, d[j]);      // (dnt has new_t's shape and row stride)
    }
  }
};

// k-major: X[32 c + r][col0 + ..] for reduction rows < nrows (else zeros); the 32 columns col0.. must exist
struct RowsK {
  const float* X; int ld, col, nrows, r8;
  __device__ __forceinline__ void init(const float* X_, int ld_, int col0, int nrows_, const Lane& L) {
    X = X_; ld = ld_; col = col0 + 4 * L.c4; nrows = nrows_; r8 = L.r8;
  }
  __device__ __forceinline__ void chunk(int c, v4f (&d)[4]) const {
    #pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = 32 * c + r8 + 8 * j;
      const v4f v = ld4(X + min(m, nrows - 1) * ld + col);
      d[j] = m < nrows ? v : zero4();
    }
  }
};

// k-major, masked: (G * relu'(Y))[32 c + r][col0 + ..]; G == nullptr: zeros
struct RowsKMasked {
  const float* G; const float* Y; int ld, col, nrows, r8;
  __device__ __forceinline__ void init(const float* G_, const float* Y_, int ld_, int col0, int nrows_, const Lane& L) {
    G = G_; Y = Y_; ld = ld_; col = col0 + 4 * L.c4; nrows = nrows_; r8 = L.r8;
  }
  __device__ __forceinline__ void chunk(int c, v4f (&d)[4]) const {
    if (!G) {
      #pragma unroll
      for (int j = 0; j < 4; ++j) d[j] = zero4();
      return;
    }
    v4f g[4], y[4];
    #pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int o = min(32 * c + r8 + 8 * j, nrows - 1) * ld + col;
      g[j] = ld4(G + o); y[j] = ld4(Y + o);
    }
    #pragma unroll
    for (int j = 0; j < 4; ++j) d[j] = 32 * c + r8 + 8 * j < nrows ? mask4(g[j], y[j]) : zero4();
  }
};

// k-major: the gathered net1 input [obj[s[t]] | pred[t] | obj[o[t]]][32 c + r][col0 + ..]; a 32-column block lies in
// one of the three sources (din is a multiple of 32)
struct RowsKTriple {
  const float* base; const long long* idx; int ld, col, nrows, r8;
  __device__ __forceinline__ void init(const float* obj, int ld_obj, const float* pred, int ld_pred, const long long* s_idx,
                                       const long long* o_idx, int din, int col0, int nrows_, const Lane& L) {
    const int src = col0 / din;
    base = src == 1 ? pred : obj; ld = src == 1 ? ld_pred : ld_obj; idx = src == 0 ? s_idx : src == 1 ? nullptr : o_idx;
    col = col0 - src * din + 4 * L.c4; nrows = nrows_; r8 = L.r8;
  }
  __device__ __forceinline__ void chunk(int c, v4f (&d)[4]) const {
    int row[4];
    #pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = min(32 * c + r8 + 8 * j, nrows - 1);
      row[j] = idx ? (int)idx[m] : m;
    }
    #pragma unroll
    for (int j = 0; j < 4; ++j) {
      const v4f v = ld4(base + row[j] * ld + col);
      d[j] = 32 * c + r8 + 8 * j < nrows ? v : zero4();
    }
  }
};

// ----------------------------------------------------------------------------------------------------------------
// LDS staging and MFMA fragments
// ----------------------------------------------------------------------------------------------------------------
template <bool KMAJ>
__device__ __forceinline__ void stage_rows(float* img, const v4f (&r)[4], const Lane& L) {
  constexpr int LD = KMAJ ? kLdK : kLdM;
  #pragma unroll
  for (int j = 0; j < 4; ++j) st4(img + (L.r8 + 8 * j) * LD + 4 * L.c4, r[j]);
}

// lane l supplies element (i = l & 31, k = kperm(s, l >> 5)) of MFMA step s; kperm(s, h) = 8 (s >> 2) + 4 h + (s & 3)
template <bool KMAJ>
__device__ __forceinline__ void read_frag(const float* img, int lane, float (&f)[16]) {
  const int i = lane & 31, h = lane >> 5;
  if (!KMAJ) {
    const float* row = img + i * kLdM + 4 * h;
    #pragma unroll
    for (int g = 0; g < 4; ++g) {
      const v4f v = ld4(row + 8 * g);
      f[4 * g + 0] = v.x; f[4 * g + 1] = v.y; f[4 * g + 2] = v.z; f[4 * g + 3] = v.w;
    }
  } else {
    const float* col = img + 4 * h * kLdK + i;
    #pragma unroll
    for (int s = 0; s < 16; ++s) f[s] = col[(8 * (s >> 2) + (s & 3)) * kLdK];
  }
}

template <int NB> struct ChunkRegs { v4f a[4]; v4f b[NB][4]; };

template <int NB, typename RA, typename RB>
__device__ __forceinline__ void load_chunk(const RA& ra, const RB (&rb)[NB], int c, ChunkRegs<NB>& g) {
  ra.chunk(c, g.a);
  #pragma unroll
  for (int nb = 0; nb < NB; ++nb) rb[nb].chunk(c, g.b[nb]);
}

// registers -> this wave's LDS images -> MFMA fragments -> 16 NB MFMAs.  LDS operations of one wave execute in order
// and the images are private to the wave: no workgroup barrier; the wave-level fences only pin the compiler's order.
template <int NB, bool AK, bool BK>
__device__ __forceinline__ void mma_chunk(const ChunkRegs<NB>& g, float* img, const Lane& L, f32x16 (&acc)[NB]) {
  stage_rows<AK>(img, g.a, L);
  #pragma unroll
  for (int nb = 0; nb < NB; ++nb) stage_rows<BK>(img + (1 + nb) * kImgFloats, g.b[nb], L);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  float fa[16], fb[NB][16];
  read_frag<AK>(img, L.lane, fa);
  #pragma unroll
  for (int nb = 0; nb < NB; ++nb) read_frag<BK>(img + (1 + nb) * kImgFloats, L.lane, fb[nb]);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  #pragma unroll
  for (int s = 0; s < 16; ++s)
    #pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s], fb[nb][s], acc[nb], 0, 0, 0);
}

// Footprint of a kernel's tile machinery.  Full: up to four K chunks of a wave in flight (4 x 48 staging registers at
// NB = 2), 32 x 64 tiles, a reduction scratch next to the wave images - 98 KB of LDS, ~390-500 registers: one
// workgroup per CU, which needs WHOLE CUs.  Low (round 5): two chunks in flight, 32 x 32 tiles
// only, the reduction scratch ALIASES the wave images (one more workgroup barrier per tile) - 41 KB of LDS, <= 168
// registers: its workgroups co-reside with the refinement network's weight gradients, so the one-launch backward
// can run inside the captured iteration's tail instead of waiting for the weight-gradient lane to drain.
struct FullFootprint { static constexpr int MAXNB = 2, MAXCH = 4; static constexpr bool ALIAS = false; };
struct LowFootprint { static constexpr int MAXNB = 1, MAXCH = 2; static constexpr bool ALIAS = true; };
template <typename CFG> struct LdsPlan {
  static constexpr int kWaveFloats = (1 + CFG::MAXNB) * kImgFloats;
  static constexpr int kStage = 4 * kWaveFloats;
  static constexpr int kRedLd_ = 32 * CFG::MAXNB + 8;               // 4 * ld = 32 (mod 64): the two lane halves on disjoint banks
  static constexpr int kRed = 4 * 32 * kRedLd_;
  static constexpr int kRedOff = CFG::ALIAS ? 0 : kStage;
  static constexpr size_t kBytes = sizeof(float) * (CFG::ALIAS ? (kStage > kRed ? kStage : kRed) : kStage + kRed);
};
static_assert(LdsPlan<FullFootprint>::kBytes == kLdsBytes, "the full plan is the original layout");

// NCH chunks (cbase, cbase + 4, ...) of one wave: all their global loads are issued first.  The register sets are
// separate named objects (an array indexed by the chunk-in-batch number is not reliably promoted to registers), and
// the body is instantiated per chunk count: with the loads under run-time conditions the compiler's s_waitcnt
// placement has to assume the shortest path and ends up waiting for nearly all loads before the first MFMA.
template <int NB, int NCH, bool AK, bool BK, typename RA, typename RB>
__device__ __forceinline__ void batch(const RA& ra, const RB (&rb)[NB], int cbase, float* img, const Lane& L, f32x16 (&acc)[NB]) {
  ChunkRegs<NB> g0, g1, g2, g3;
  load_chunk<NB>(ra, rb, cbase, g0);
  if (NCH > 1) load_chunk<NB>(ra, rb, cbase + 4, g1);
  if (NCH > 2) load_chunk<NB>(ra, rb, cbase + 8, g2);
  if (NCH > 3) load_chunk<NB>(ra, rb, cbase + 12, g3);
  mma_chunk<NB, AK, BK>(g0, img, L, acc);
  if (NCH > 1) mma_chunk<NB, AK, BK>(g1, img, L, acc);
  if (NCH > 2) mma_chunk<NB, AK, BK>(g2, img, L, acc);
  if (NCH > 3) mma_chunk<NB, AK, BK>(g3, img, L, acc);
}

// One 32 x (32 NB) tile: acc = sum over nchunks K chunks of A-chunk x B-chunk (waves split the chunks), the four
// partials added in wave order through LDS, then epi(row, col, v4f) for every 4-column piece of the tile
// (row in [0, 32), col in [0, 32 NB) a multiple of 4).
template <int NB, bool AK, bool BK, typename CFG = FullFootprint, typename RA, typename RB, typename Epi>
__device__ __forceinline__ void tile_gemm(const RA& ra, const RB (&rb)[NB], int nchunks, float* smem, const Lane& L, const Epi& epi) {
  static_assert(NB <= CFG::MAXNB, "tile wider than the footprint's wave images");
  typedef LdsPlan<CFG> LP;
  constexpr int RLD = LP::kRedLd_;
  float* const img = smem + L.wave * LP::kWaveFloats;
  float* const red = smem + LP::kRedOff;
  f32x16 acc[NB];
  #pragma unroll
  for (int nb = 0; nb < NB; ++nb)
    #pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
  for (int cbase = L.wave; cbase < nchunks; cbase += 4 * CFG::MAXCH) {
    const int n = (nchunks - cbase + 3) >> 2;              // this wave's chunks from cbase on (wave-uniform)
    if (CFG::MAXCH >= 4 && n >= 4) batch<NB, 4, AK, BK>(ra, rb, cbase, img, L, acc);
    else if (CFG::MAXCH >= 4 && n == 3) batch<NB, 3, AK, BK>(ra, rb, cbase, img, L, acc);
    else if (n >= 2) batch<NB, 2, AK, BK>(ra, rb, cbase, img, L, acc);
    else batch<NB, 1, AK, BK>(ra, rb, cbase, img, L, acc);
  }
  __syncthreads();                                   // (every thread is done with the previous tile's `red`; ALIAS: and every
                                                     //  wave with its images, which `red` overwrites)
  {
    const int jc = L.lane & 31, h = L.lane >> 5;
    float* const mine = red + L.wave * 32 * RLD;
    #pragma unroll
    for (int nb = 0; nb < NB; ++nb)
      #pragma unroll
      for (int r = 0; r < 16; ++r) mine[((r & 3) + 8 * (r >> 2) + 4 * h) * RLD + 32 * nb + jc] = acc[nb][r];
  }
  __syncthreads();
  {
    const int row = threadIdx.x >> 3, col0 = (threadIdx.x & 7) * 4 * NB;
    #pragma unroll
    for (int q = 0; q < NB; ++q) {
      const int col = col0 + 4 * q;
      const float* p = red + row * RLD + col;
      const v4f v = ((ld4(p) + ld4(p + 32 * RLD)) + ld4(p + 64 * RLD)) + ld4(p + 96 * RLD);
      epi(row, col, v);
    }
  }
  if (CFG::ALIAS) __syncthreads();                   // (the next tile's wave images overwrite `red`)
}

// epilogues: (row, col) are tile-local
struct EpiBiasRelu {               // out[m0 + row][n0 + col] = relu(v + bias)
  float* out; const float* bias; int ldo, m0, n0, M, N;
  __device__ __forceinline__ void operator()(int row, int col, v4f v) const {
    const int m = m0 + row, n = n0 + col;
    if (m >= M || n >= N) return;
    if (bias) v = v + ld4(bias + n);
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    st4(out + m * ldo + n, v);
  }
};
struct EpiMaskStore {              // out[m][n] = v * relu'(act[m][n])  (act == nullptr: plain store); same row stride
  float* out; const float* act; int ldo, m0, n0, M, N;
  __device__ __forceinline__ void operator()(int row, int col, v4f v) const {
    const int m = m0 + row, n = n0 + col;
    if (m >= M || n >= N) return;
    if (act) v = mask4(v, ld4(act + m * ldo + n));
    st4(out + m * ldo + n, v);
  }
};
struct EpiAccum {                  // dW[i0 + row][j0 + col] (+)= v
  float* dw; int ld, i0, j0, NI, NJ, accumulate;
  __device__ __forceinline__ void operator()(int row, int col, v4f v) const {
    const int i = i0 + row, j = j0 + col;
    if (i >= NI || j >= NJ) return;
    float* p = dw + i * ld + j;
    if (accumulate) v = v + ld4(p);
    st4(p, v);
  }
};

// Tiles of a stage over the resident workgroups.  Column-tile-major order cut into contiguous runs per XCD (the
// dispatcher places block b on XCD b % 8 - a locality assumption only): an XCD's L2 then holds a few column tiles'
// weight rows instead of all of them.
struct TileWalk {
  int first, step, end;
  __device__ __forceinline__ TileWalk(int ntiles) {
    const int g = gridDim.x, b = blockIdx.x;
    if ((g & 7) == 0) {
      const int x = b & 7, i = b >> 3, per = (ntiles + 7) >> 3;
      first = x * per + i; step = g >> 3; end = min((x + 1) * per, ntiles);
    } else {
      first = b; step = g; end = ntiles;
    }
  }
};

// ---- forward tiles -------------------------------------------------------------------------------------------
// out[M][N] = relu(A W^T + bias), W [N][K] (nn.Linear layout).  KIND 0: A dense [M][K]; 1: the gathered triple input.
struct FwdStage {
  const float* a0; int ld0;                              // dense rows / object vectors
  const float* a1; int ld1;                              // predicate vectors
  const long long* s_idx; const long long* o_idx; int din;
  const float* W; const float* bias; float* out;
  int M, N, K;
};

template <int KIND, int NB>
__device__ __forceinline__ void fwd_tile(const FwdStage& st, int tile, int nrb, float* smem, TripleIdx& ix, const Lane& L) {
  const int cb = tile / nrb, rb = tile - cb * nrb;
  const int m0 = rb << 5, n0 = cb * 32 * NB;
  RowsM w[NB];
  #pragma unroll
  for (int nb = 0; nb < NB; ++nb) w[nb].init(st.W, st.K, n0 + 32 * nb, st.N, L);
  EpiBiasRelu epi;
  epi.out = st.out; epi.bias = st.bias; epi.ldo = st.N; epi.m0 = m0; epi.n0 = n0; epi.M = st.M; epi.N = st.N;
  if (KIND == 0) {
    RowsM a; a.init(st.a0, st.ld0, m0, st.M, L);
    tile_gemm<NB, false, false>(a, w, st.K >> 5, smem, L, epi);
  } else {
    fetch_triple_idx(ix, st.s_idx, st.o_idx, m0, st.M, L);
    RowsTriple a; a.init(st.a0, st.ld0, st.a1, st.ld1, st.din, m0, st.M, ix, L);
    tile_gemm<NB, false, false>(a, w, st.K >> 5, smem, L, epi);
  }
}

// NB = 2 (32 x 64 tiles: the A rows are staged once for two weight blocks) when 32 x 32 tiles would need more than
// one round over the resident workgroups
template <int KIND>
__device__ __forceinline__ void fwd_stage(const FwdStage& st, float* smem, TripleIdx& ix, const Lane& L) {
  const int nrb = (st.M + 31) >> 5;
  const int tiles1 = nrb * ((st.N + 31) >> 5);
  if (tiles1 > (int)gridDim.x && (st.N & 63) == 0) {
    const TileWalk tw(nrb * (st.N >> 6));
    for (int t = tw.first; t < tw.end; t += tw.step) fwd_tile<KIND, 2>(st, t, nrb, smem, ix, L);
  } else {
    const TileWalk tw(tiles1);
    for (int t = tw.first; t < tw.end; t += tw.step) fwd_tile<KIND, 1>(st, t, nrb, smem, ix, L);
  }
}

// Stage C / P5.  out[j][k] = (sum over row j's CSR entries, in order, from +0.0f, of src[e][k] (subject role: e < T)
// or src[e - T][ooff + k] (object role)) / max(1, #entries) - graph.py:92-114, the arithmetic of segment_sum_kernel.
// A wavefront per row; the first 8 entries' row pieces are independent loads issued back to back (slots past the
// row's count re-read its last entry and are dropped by a select: no branch between the loads), then added in entry
// order; longer rows finish in a plain loop.
__device__ __forceinline__ void pool_stage(const float* src, int ld, int ooff, const int* row_ptr, const int* entries, int T, int W,
                                           bool average, int O, float* out, const Lane& L) {
  const int W4 = W >> 2;
  for (int row = blockIdx.x * 4 + L.wave; row < O; row += gridDim.x * 4) {
    const int b = row_ptr[row], cnt = row_ptr[row + 1] - b;          // (wave-uniform)
    float* const dst = out + row * W;
    if (cnt == 0) {
      for (int p = L.lane; p < W4; p += 64) st4(dst + 4 * p, zero4());
      continue;
    }
    int off[8];
    #pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int id = entries[b + min(q, cnt - 1)];
      off[q] = id < T ? id * ld : (id - T) * ld + ooff;
    }
    const float d = (float)cnt;
    for (int p0 = 0; p0 < W4; p0 += 128) {
      const int pa = min(p0 + L.lane, W4 - 1), pb = min(p0 + 64 + L.lane, W4 - 1);
      v4f va[8], vb[8];
      #pragma unroll
      for (int q = 0; q < 8; ++q) { va[q] = ld4(src + off[q] + 4 * pa); vb[q] = ld4(src + off[q] + 4 * pb); }
      v4f sa = zero4(), sb = zero4();
      #pragma unroll
      for (int q = 0; q < 8; ++q) {
        const v4f ta = sa + va[q], tb = sb + vb[q];
        sa = q < cnt ? ta : sa; sb = q < cnt ? tb : sb;
      }
      for (int e = 8; e < cnt; ++e) {
        const int id = entries[b + e];
        const float* r = src + (id < T ? id * ld : (id - T) * ld + ooff);
        sa = sa + ld4(r + 4 * pa);
        sb = sb + ld4(r + 4 * pb);
      }
      if (average) {
        sa.x = sa.x / d; sa.y = sa.y / d; sa.z = sa.z / d; sa.w = sa.w / d;
        sb.x = sb.x / d; sb.y = sb.y / d; sb.z = sb.z / d; sb.w = sb.w / d;
      }
      if (p0 + L.lane < W4) st4(dst + 4 * pa, sa);
      if (p0 + 64 + L.lane < W4) st4(dst + 4 * pb, sb);
    }
  }
}

// layer l of the stack, fetched with scalar loads from the kernel-argument segment (a dynamically indexed array
// inside a by-value argument would be demoted to scratch); the stack must sit at offset 0 of the kernel's argument
__device__ __forceinline__ sg2im_gconv_stack_layer fetch_layer(int l) {
  typedef __attribute__((address_space(4))) const char* KPtr;
  typedef __attribute__((address_space(4))) const sg2im_gconv_stack_layer* KLayer;
  const KLayer k = (KLayer)((KPtr)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(sg2im_gconv_stack, layer)) + l;
  sg2im_gconv_stack_layer L;
  L.w1a = k->w1a; L.b1a = k->b1a; L.w1b = k->w1b; L.b1b = k->b1b; L.w2a = k->w2a; L.b2a = k->b2a; L.w2b = k->w2b; L.b2b = k->b2b;
  L.h1 = k->h1; L.new_t = k->new_t; L.pooled = k->pooled; L.h2 = k->h2; L.new_obj = k->new_obj;
  L.din = k->din; L.hidden = k->hidden; L.dout = k->dout; L.reserved = 0;
  return L;
}

struct FwdArgs {
  sg2im_gconv_stack s;             // (first: the layer table is read through the kernel-argument segment)
  unsigned* sync;
};

__global__ __launch_bounds__(kThreads) void gcn_stack_fwd_kernel(const FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  Sync sy = {};
  if (threadIdx.x == 0) sync_begin(sy, a.sync);
  const Lane LN = my_lane();
  TripleIdx ix;
  ix.m0 = -1;
  const int T = a.s.n_triples, O = a.s.n_objs, nl = a.s.n_layers;
  const float* obj = a.s.obj_vecs; int ld_obj = (int)a.s.ld_obj;
  const float* pred = a.s.pred_vecs; int ld_pred = (int)a.s.ld_pred;
  for (int l = 0; l < nl; ++l) {
    const sg2im_gconv_stack_layer L = fetch_layer(l);
    const int H = L.hidden, Dout = L.dout, NTc = 2 * H + Dout;
    FwdStage st;
    st.a0 = nullptr; st.ld0 = 0; st.a1 = nullptr; st.ld1 = 0; st.s_idx = a.s.s_idx; st.o_idx = a.s.o_idx; st.din = L.din;
    if (T > 0) {
      st.a0 = obj; st.ld0 = ld_obj; st.a1 = pred; st.ld1 = ld_pred;
      st.W = L.w1a; st.bias = L.b1a; st.out = L.h1; st.M = T; st.N = H; st.K = 3 * L.din;
      fwd_stage<1>(st, smem, ix, LN);
      grid_barrier(sy);
      st.a0 = L.h1; st.ld0 = H;
      st.W = L.w1b; st.bias = L.b1b; st.out = L.new_t; st.M = T; st.N = NTc; st.K = H;
      fwd_stage<0>(st, smem, ix, LN);
      grid_barrier(sy);
    }
    pool_stage(L.new_t, NTc, H + Dout, a.s.row_ptr, a.s.entries, T, H, a.s.average != 0, O, L.pooled, LN);
    grid_barrier(sy);
    st.a0 = L.pooled; st.ld0 = H;
    st.W = L.w2a; st.bias = L.b2a; st.out = L.h2; st.M = O; st.N = H; st.K = H;
    fwd_stage<0>(st, smem, ix, LN);
    grid_barrier(sy);
    st.a0 = L.h2; st.ld0 = H;
    st.W = L.w2b; st.bias = L.b2b; st.out = L.new_obj; st.M = O; st.N = Dout; st.K = H;
    fwd_stage<0>(st, smem, ix, LN);
    if (l + 1 < nl) grid_barrier(sy);
    obj = L.new_obj; ld_obj = Dout;
    pred = L.new_t + H; ld_pred = NTc;
  }
  if (threadIdx.x == 0) stamp(sy);
}

// ---- backward --------------------------------------------------------------------------------------------------
struct BwdArgs {
  sg2im_gconv_stack s;             // (first, see fetch_layer)
  sg2im_gconv_stack_grads g;
  // scratch carve-up: element offsets from g.scratch
  long long off_dpooled, off_dnt, off_dp1, off_dtriple0, off_dtriple1, off_dobj0, off_dobj1;
  unsigned* sync;
  // The stages this launch runs: [stage_lo, stage_hi) of the sequence layer nl-1: P1..P5, layer nl-2: P1..P5, ...
  // (5 per layer).  0 .. 5 nl = the whole backward in one co-resident launch; a launch of ONE stage executes no grid
  // barrier at all - the "staged" form: the same tiles (32 x 32, four wavefronts splitting K, the weight / bias gradients
  // riding as extra tiles) as 25 ordinary launches that need no co-residency and poll nothing.
  int stage_lo, stage_hi;
};

__device__ __forceinline__ sg2im_gconv_grads fetch_grads(int l) {
  typedef __attribute__((address_space(4))) const char* KPtr;
  typedef __attribute__((address_space(4))) const sg2im_gconv_grads* KG;
  const KG k = (KG)((KPtr)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(BwdArgs, g) + offsetof(sg2im_gconv_stack_grads, layer)) + l;
  sg2im_gconv_grads G;
  G.dw1a = k->dw1a; G.db1a = k->db1a; G.dw1b = k->dw1b; G.db1b = k->db1b; G.dw2a = k->dw2a; G.db2a = k->db2a; G.dw2b = k->dw2b;
  G.db2b = k->db2b; G.accumulate = k->accumulate;
  return G;
}

// data-gradient tile: out[M][N] = (A Wt) [* relu'(act)], Wt = W [K][N] k-major.  AKIND 0: A dense [M][K]; 1: G * relu'(Y);
// 2: the rebuilt dnt rows (RowsDnt)
struct DgradStage {
  const float* a; const float* y; int M, N, K;           // dense / masked A: row stride K
  const float* W; float* out; const float* act;          // out and act: row stride N
  // dnt
  const float* dpooled; const float* g_pred; int ld_gp; const float* new_t; int H, Dout; const int* row_ptr; int average; float* dnt;
  const long long* s_idx; const long long* o_idx;
};

template <int AKIND, typename CFG>
__device__ __forceinline__ void dgrad_tile(const DgradStage& st, int tile, int nrb, float* smem, TripleIdx& ix, const Lane& L) {
  const int cb = tile / nrb, rb = tile - cb * nrb;
  const int m0 = rb << 5, n0 = cb << 5;
  RowsK w[1];
  w[0].init(st.W, st.N, n0, st.K, L);
  EpiMaskStore epi;
  epi.out = st.out; epi.act = st.act; epi.ldo = st.N; epi.m0 = m0; epi.n0 = n0; epi.M = st.M; epi.N = st.N;
  if (AKIND == 0) {
    RowsM a; a.init(st.a, st.K, m0, st.M, L);
    tile_gemm<1, false, true, CFG>(a, w, st.K >> 5, smem, L, epi);
  } else if (AKIND == 1) {
    RowsMMasked a; a.init(st.a, st.y, st.K, m0, st.M, L);
    tile_gemm<1, false, true, CFG>(a, w, st.K >> 5, smem, L, epi);
  } else {
    fetch_triple_idx(ix, st.s_idx, st.o_idx, m0, st.M, L);
    RowsDnt a;
    a.init(st.dpooled, st.H, st.g_pred, st.ld_gp, st.new_t, st.K, st.Dout, st.row_ptr, st.average != 0, cb == 0 ? st.dnt : nullptr, m0,
           st.M, ix, L);
    tile_gemm<1, false, true, CFG>(a, w, st.K >> 5, smem, L, epi);
  }
}

// weight-gradient tile: dW[NI][NJ] (+)= dY^T X over R rows; dY [R][NI] (AKIND 1: G * relu'(Y)), X [R][NJ] (BKIND 1: the
// gathered net1 input).  32 x 64 tiles.
struct WgradStage {
  const float* dy; const float* y; int NI;
  const float* x; int NJ;
  int R;
  float* dw; int accumulate;
  // gathered X
  const float* obj; int ld_obj; const float* pred; int ld_pred; const long long* s_idx; const long long* o_idx; int din;
};
template <typename CFG>
__device__ __forceinline__ int wgrad_tiles(const WgradStage& st) {
  return st.dw ? ((st.NI + 31) >> 5) * ((st.NJ + 32 * CFG::MAXNB - 1) / (32 * CFG::MAXNB)) : 0;
}

// (32 x 64 tiles with the full footprint, 32 x 32 with the low one: WNB = CFG::MAXNB column blocks)
template <int AKIND, int BKIND, typename CFG>
__device__ __forceinline__ void wgrad_tile(const WgradStage& st, int tile, float* smem, const Lane& L) {
  constexpr int WNB = CFG::MAXNB;
  const int nib = (st.NI + 31) >> 5;
  const int jb = tile / nib, ib = tile - jb * nib;
  const int i0 = ib << 5, j0 = jb * 32 * WNB;
  const int nchunks = (st.R + 31) >> 5;
  EpiAccum epi;
  epi.dw = st.dw; epi.ld = st.NJ; epi.i0 = i0; epi.j0 = j0; epi.NI = st.NI; epi.NJ = st.NJ; epi.accumulate = st.accumulate;
  if (BKIND == 0) {
    RowsK x[WNB];
    #pragma unroll
    for (int q = 0; q < WNB; ++q) x[q].init(st.x, st.NJ, min(j0 + 32 * q, st.NJ - 32), st.R, L);
    if (AKIND == 0) { RowsK a; a.init(st.dy, st.NI, i0, st.R, L); tile_gemm<WNB, true, true, CFG>(a, x, nchunks, smem, L, epi); }
    else { RowsKMasked a; a.init(st.dy, st.y, st.NI, i0, st.R, L); tile_gemm<WNB, true, true, CFG>(a, x, nchunks, smem, L, epi); }
  } else {
    RowsKTriple x[WNB];
    #pragma unroll
    for (int q = 0; q < WNB; ++q)
      x[q].init(st.obj, st.ld_obj, st.pred, st.ld_pred, st.s_idx, st.o_idx, st.din, min(j0 + 32 * q, st.NJ - 32), st.R, L);
    RowsK a; a.init(st.dy, st.NI, i0, st.R, L);
    tile_gemm<WNB, true, true, CFG>(a, x, nchunks, smem, L, epi);
  }
}

// bias-gradient work item: db[n0 .. n0 + 32) (+)= column sums of dY [R][N] (MASK: of G * relu'(Y)) - fixed order:
// thread (g = tid >> 3, piece = tid & 7) sums rows g, g + 32, ... of its 4 columns, the 32 partial rows are then added in
// order
template <bool MASK, typename CFG>
__device__ __forceinline__ void colsum_tile(const float* dy, const float* y, int R, int N, int n0, float* db, int accumulate,
                                            float* smem) {
  const int g = threadIdx.x >> 3, piece = threadIdx.x & 7;
  const int col = n0 + 4 * piece;
  v4f s = zero4();
  if (dy)
    for (int r = g; r < R; r += 32) {
      const v4f v = ld4(dy + r * N + col);
      s = s + (MASK ? mask4(v, ld4(y + r * N + col)) : v);
    }
  float* const red = smem + LdsPlan<CFG>::kRedOff;
  __syncthreads();
  st4(red + g * 32 + 4 * piece, s);
  __syncthreads();
  if (threadIdx.x < 32) {
    float t = 0.f;
    #pragma unroll
    for (int r = 0; r < 32; ++r) t = t + red[r * 32 + threadIdx.x];
    float* p = db + n0 + threadIdx.x;
    *p = accumulate ? *p + t : t;
  }
  if (CFG::ALIAS) __syncthreads();                   // (the next tile's wave images overwrite `red`)
}

template <typename CFG>
__device__ __forceinline__ void gcn_stack_bwd_body(const BwdArgs& a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  Sync sy = {};
  if (threadIdx.x == 0) sync_begin(sy, a.sync);
  const Lane LN = my_lane();
  TripleIdx ix;
  ix.m0 = -1;
  const int T = a.s.n_triples, O = a.s.n_objs, nl = a.s.n_layers;
  const float* g_obj = a.g.g_obj;
  const float* g_pred = a.g.g_pred; int ld_gp = (int)a.g.ld_gpred;
  float* const dp3 = a.g.scratch;                           // [O][H]      (H = the widest hidden size: sized by the host)
  float* const dpooled = dp3 + a.off_dpooled;             // [O][H]
  float* const dnt = dp3 + a.off_dnt;                     // [T][2H + Dout]
  float* const dp1 = dp3 + a.off_dp1;                     // [T][H]
  int gs = 0;                                               // global stage index (workgroup-uniform)
  const int lo = a.stage_lo, hi = a.stage_hi;
  for (int l = nl - 1; l >= 0; --l, gs += 5) {
    if (gs + 5 <= lo || gs >= hi) {                         // (none of this layer's stages is in this launch)
      const sg2im_gconv_stack_layer Lq = fetch_layer(l);
      float* const dq = l == 0 ? a.g.d_triple : dp3 + (l & 1 ? a.off_dtriple1 : a.off_dtriple0);
      g_obj = l == 0 ? a.g.d_obj : dp3 + (l & 1 ? a.off_dobj1 : a.off_dobj0); g_pred = dq + Lq.din; ld_gp = 3 * Lq.din;
      continue;
    }
    const sg2im_gconv_stack_layer L = fetch_layer(l);
    const sg2im_gconv_grads G = fetch_grads(l);
    const int H = L.hidden, Dout = L.dout, Din = L.din, NTc = 2 * H + Dout;
    // this layer's inputs: the previous layer's outputs (layer 0: the stack's inputs)
    const float* xin = a.s.obj_vecs; int ld_x = (int)a.s.ld_obj;
    const float* pin = a.s.pred_vecs; int ld_p = (int)a.s.ld_pred;
    if (l > 0) {
      const sg2im_gconv_stack_layer P = fetch_layer(l - 1);
      xin = P.new_obj; ld_x = P.dout; pin = P.new_t + P.hidden; ld_p = 2 * P.hidden + P.dout;
    }
    // where this layer's input gradients go: ping-pong scratch, the caller's buffers for layer 0
    float* const d_triple = l == 0 ? a.g.d_triple : dp3 + (l & 1 ? a.off_dtriple1 : a.off_dtriple0);
    float* const d_obj = l == 0 ? a.g.d_obj : dp3 + (l & 1 ? a.off_dobj1 : a.off_dobj0);

    // ---- P1: dp3 = ((g_obj * relu'(new_obj)) W2b) * relu'(h2);  dW2b, db2b
    if (gs >= lo && gs < hi) {
      DgradStage st = {};
      st.a = g_obj; st.y = L.new_obj; st.M = O; st.N = H; st.K = Dout; st.W = L.w2b; st.out = dp3; st.act = L.h2;
      WgradStage wg = {};
      wg.dy = g_obj; wg.y = L.new_obj; wg.NI = Dout; wg.x = L.h2; wg.NJ = H; wg.R = O; wg.dw = G.dw2b; wg.accumulate = G.accumulate;
      const int nrb = (O + 31) >> 5, nd = nrb * (H >> 5), nw = wgrad_tiles<CFG>(wg), nc = G.db2b ? Dout >> 5 : 0;
      const TileWalk tw(nd + nw + nc);
      for (int t = tw.first; t < tw.end; t += tw.step) {
        if (t < nd) dgrad_tile<1, CFG>(st, t, nrb, smem, ix, LN);
        else if (t < nd + nw) wgrad_tile<1, 0, CFG>(wg, t - nd, smem, LN);
        else colsum_tile<true, CFG>(g_obj, L.new_obj, O, Dout, (t - nd - nw) << 5, G.db2b, G.accumulate, smem);
      }
    }
    if (gs >= lo && gs + 1 < hi) grid_barrier(sy);
    // ---- P2: dpooled = dp3 W2a;  dW2a, db2a
    if (gs + 1 >= lo && gs + 1 < hi) {
      DgradStage st = {};
      st.a = dp3; st.M = O; st.N = H; st.K = H; st.W = L.w2a; st.out = dpooled; st.act = nullptr;
      WgradStage wg = {};
      wg.dy = dp3; wg.NI = H; wg.x = L.pooled; wg.NJ = H; wg.R = O; wg.dw = G.dw2a; wg.accumulate = G.accumulate;
      const int nrb = (O + 31) >> 5, nd = T > 0 ? nrb * (H >> 5) : 0, nw = wgrad_tiles<CFG>(wg), nc = G.db2a ? H >> 5 : 0;
      const TileWalk tw(nd + nw + nc);
      for (int t = tw.first; t < tw.end; t += tw.step) {
        if (t < nd) dgrad_tile<0, CFG>(st, t, nrb, smem, ix, LN);
        else if (t < nd + nw) wgrad_tile<0, 0, CFG>(wg, t - nd, smem, LN);
        else colsum_tile<false, CFG>(dp3, nullptr, O, H, (t - nd - nw) << 5, G.db2a, G.accumulate, smem);
      }
    }
    if (T > 0) {
      if (gs + 1 >= lo && gs + 2 < hi) grid_barrier(sy);
      // ---- P3: dp1 = (dnt W1b) * relu'(h1), dnt rebuilt in the loader and written by the first column block
      if (gs + 2 >= lo && gs + 2 < hi) {
        DgradStage st = {};
        st.M = T; st.N = H; st.K = NTc; st.W = L.w1b; st.out = dp1; st.act = L.h1;
        st.dpooled = dpooled; st.g_pred = g_pred; st.ld_gp = ld_gp; st.new_t = L.new_t; st.H = H; st.Dout = Dout;
        st.row_ptr = a.s.row_ptr; st.average = a.s.average; st.dnt = dnt; st.s_idx = a.s.s_idx; st.o_idx = a.s.o_idx;
        const int nrb = (T + 31) >> 5;
        const TileWalk tw(nrb * (H >> 5));
        for (int t = tw.first; t < tw.end; t += tw.step) dgrad_tile<2, CFG>(st, t, nrb, smem, ix, LN);
      }
      if (gs + 2 >= lo && gs + 3 < hi) grid_barrier(sy);
      // ---- P4: d_triple = dp1 W1a;  dW1a, db1a, db1b
      if (gs + 3 >= lo && gs + 3 < hi) {
        DgradStage st = {};
        st.a = dp1; st.M = T; st.N = 3 * Din; st.K = H; st.W = L.w1a; st.out = d_triple; st.act = nullptr;
        WgradStage wg = {};
        wg.dy = dp1; wg.NI = H; wg.NJ = 3 * Din; wg.R = T; wg.dw = G.dw1a; wg.accumulate = G.accumulate;
        wg.obj = xin; wg.ld_obj = ld_x; wg.pred = pin; wg.ld_pred = ld_p; wg.s_idx = a.s.s_idx; wg.o_idx = a.s.o_idx; wg.din = Din;
        const int nrb = (T + 31) >> 5, nd = nrb * ((3 * Din) >> 5), nw = wgrad_tiles<CFG>(wg);
        const int nc1 = G.db1a ? H >> 5 : 0, nc2 = G.db1b ? NTc >> 5 : 0;
        const TileWalk tw(nd + nw + nc1 + nc2);
        for (int t = tw.first; t < tw.end; t += tw.step) {
          if (t < nd) dgrad_tile<0, CFG>(st, t, nrb, smem, ix, LN);
          else if (t < nd + nw) wgrad_tile<0, 1, CFG>(wg, t - nd, smem, LN);
          else if (t < nd + nw + nc1) colsum_tile<false, CFG>(dp1, nullptr, T, H, (t - nd - nw) << 5, G.db1a, G.accumulate, smem);
          else colsum_tile<false, CFG>(dnt, nullptr, T, NTc, (t - nd - nw - nc1) << 5, G.db1b, G.accumulate, smem);
        }
      }
      if (gs + 3 >= lo && gs + 4 < hi) grid_barrier(sy);
      // ---- P5: d_obj = CSR sum of d_triple's subject / object blocks;  dW1b
      if (gs + 4 >= lo && gs + 4 < hi) {
        WgradStage wg = {};
        wg.dy = dnt; wg.NI = NTc; wg.x = L.h1; wg.NJ = H; wg.R = T; wg.dw = G.dw1b; wg.accumulate = G.accumulate;
        const int nw = wgrad_tiles<CFG>(wg);
        const TileWalk tw(nw);
        for (int t = tw.first; t < tw.end; t += tw.step) wgrad_tile<0, 0, CFG>(wg, t, smem, LN);
        if (d_obj) pool_stage(d_triple, 3 * Din, 2 * Din, a.s.row_ptr, a.s.entries, T, Din, false, O, d_obj, LN);
      }
    }
    if (l > 0 && gs + 4 >= lo && gs + 5 < hi) grid_barrier(sy);
    g_obj = d_obj; g_pred = d_triple + Din; ld_gp = 3 * Din;
  }
  if (threadIdx.x == 0) stamp(sy);
}

__global__ __launch_bounds__(kThreads) void gcn_stack_bwd_kernel(const BwdArgs a) { gcn_stack_bwd_body<FullFootprint>(a); }
// <= 168 registers (three wavefronts per SIMD): a wavefront of this kernel fits next to a resident wavefront of the
// halo'd weight-gradient kernel (wgrad_halo.h: 324 of the SIMD's 512 registers per lane, 88 of the CU's 160 KB of
// LDS) or next to two of the per-tap kernel's (2 x 152).  The cap costs ~70 spilled registers - values that live
// ACROSS stages (layer pointers, lane geometry), saved / reloaded around the grid barriers and per tile, never
// inside a K loop (checked in the ISA with -gline-tables-only): the price of co-residency, 25 times per launch.
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(3)))
void gcn_stack_bwd_low_kernel(const BwdArgs a) { gcn_stack_bwd_body<LowFootprint>(a); }

static int g_cus = 0;

hipError_t prepare() {
  if (g_cus > 0) return hipSuccess;
  int dev = 0, cus = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (e == hipSuccess)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(gcn_stack_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)kLdsBytes);
  if (e == hipSuccess)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(gcn_stack_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)kLdsBytes);
  if (e == hipSuccess)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(gcn_stack_bwd_low_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)LdsPlan<LowFootprint>::kBytes);
  if (e == hipSuccess) g_cus = cus > 0 ? cus : 1;
  return e;
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static bool stack_ok(const sg2im_gconv_stack* S) {
  if (!S || S->n_layers < 1 || S->n_layers > SG2IM_GCONV_MAX_LAYERS || S->n_objs < 1 || S->n_triples < 0) return false;
  if (!S->obj_vecs || !S->row_ptr || !aligned16(S->obj_vecs) || (S->ld_obj & 3)) return false;
  if (S->n_triples > 0 && (!S->pred_vecs || !S->s_idx || !S->o_idx || !S->entries || !aligned16(S->pred_vecs) || (S->ld_pred & 3)))
    return false;
  const long long O = S->n_objs, T = S->n_triples, lim = 1ll << 31;
  if (O * S->ld_obj >= lim || T * S->ld_pred >= lim) return false;
  int din = 0;
  for (int l = 0; l < S->n_layers; ++l) {
    const sg2im_gconv_stack_layer& L = S->layer[l];
    if (L.din < 32 || L.hidden < 32 || L.dout < 32 || (L.din & 31) || (L.hidden & 31) || (L.dout & 31)) return false;
    if (l > 0 && L.din != din) return false;                      // a layer reads the previous layer's output
    din = L.dout;
    if (l == 0 && (S->ld_obj < L.din || (S->n_triples > 0 && S->ld_pred < L.din))) return false;
    const long long NTc = 2ll * L.hidden + L.dout;
    if (T * NTc >= lim || O * L.hidden >= lim || NTc * L.hidden >= lim || 3ll * L.din * L.hidden >= lim) return false;
    const void* ps[] = {L.w1a, L.w1b, L.w2a, L.w2b, L.pooled, L.h2, L.new_obj};
    for (const void* p : ps) if (!p || !aligned16(p)) return false;
    const void* bs[] = {L.b1a, L.b1b, L.b2a, L.b2b};
    for (const void* p : bs) if (p && !aligned16(p)) return false;
    if (S->n_triples > 0 && (!L.h1 || !L.new_t || !aligned16(L.h1) || !aligned16(L.new_t))) return false;
  }
  return true;
}

// resident grid: one workgroup per CU, fewer when no stage has that many 32 x 32 tiles
static int grid_for(const sg2im_gconv_stack* S, bool backward) {
  const long long T = S->n_triples, O = S->n_objs;
  long long most = 1;
  for (int l = 0; l < S->n_layers; ++l) {
    const sg2im_gconv_stack_layer& L = S->layer[l];
    const long long rt = (T + 31) / 32, ro = (O + 31) / 32, NTc = 2 * L.hidden + L.dout;
    most = std::max({most, rt * (L.hidden / 32), rt * (NTc / 32), ro * (L.hidden / 32), ro * (L.dout / 32), (O + 3) / 4});
    if (backward) most = std::max({most, (NTc / 32) * ((L.hidden + 63) / 64), rt * (3 * L.din / 32) + (L.hidden / 32) * ((3 * L.din + 63) / 64)});
  }
  int grid = (int)std::min<long long>(g_cus, most);
  // (probe knob: fewer resident workgroups for the BACKWARD launch - it then takes longer but leaves more of every CU
  // to the weight gradients it runs next to)
  static const int cap = getenv("SG2IM_GCN_BWD_GRID") ? atoi(getenv("SG2IM_GCN_BWD_GRID")) : 0;
  if (backward && cap > 0) grid = std::min(grid, cap);
  if (grid >= 8) grid &= ~7;
  return grid;
}

}  // namespace gcn
}  // namespace sg2im

using namespace sg2im;

extern "C" {

size_t sg2im_gconv_stack_sync_bytes(void) { return 8192; }

int sg2im_gconv_stack_supported(int din, int hidden, int dout) {
  return din >= 32 && hidden >= 32 && dout >= 32 && !(din & 31) && !(hidden & 31) && !(dout & 31);
}

int sg2im_gconv_stack_forward(const sg2im_gconv_stack* S, void* sync, size_t sync_bytes, hipStream_t stream) {
  if (!gcn::stack_ok(S) || !sync || sync_bytes < sg2im_gconv_stack_sync_bytes() || !gcn::aligned16(sync)) return SG2IM_ERR_ARG;
  if (gcn::prepare() != hipSuccess) return SG2IM_ERR_HIP;
  if (hipMemsetAsync(sync, 0, sizeof(unsigned) * gcn::kStickyWord, stream) != hipSuccess) return SG2IM_ERR_HIP;
  gcn::FwdArgs a;
  std::memcpy(&a.s, S, sizeof(*S));
  a.sync = static_cast<unsigned*>(sync);
  SG2IM_LAUNCH(gcn::gcn_stack_fwd_kernel, dim3(gcn::grid_for(S, false)), dim3(gcn::kThreads), gcn::kLdsBytes, stream, a);
  return hipGetLastError() == hipSuccess ? SG2IM_OK : SG2IM_ERR_HIP;
}

size_t sg2im_gconv_stack_backward_scratch(const sg2im_gconv_stack* S) {
  if (!S || S->n_layers < 1 || S->n_layers > SG2IM_GCONV_MAX_LAYERS) return 0;
  size_t H = 0, NT = 0, D = 0;
  for (int l = 0; l < S->n_layers; ++l) {
    H = std::max<size_t>(H, S->layer[l].hidden); NT = std::max<size_t>(NT, 2 * (size_t)S->layer[l].hidden + S->layer[l].dout);
    D = std::max<size_t>(D, S->layer[l].din);
  }
  const size_t O = S->n_objs, T = S->n_triples;
  return sizeof(float) * (2 * O * H + T * NT + T * H + 2 * T * 3 * D + 2 * O * D + 64);
}

int sg2im_gconv_stack_backward(const sg2im_gconv_stack* S, const sg2im_gconv_stack_grads* G, void* sync, size_t sync_bytes,
                               hipStream_t stream) {
  if (!gcn::stack_ok(S) || !G || !sync || sync_bytes < sg2im_gconv_stack_sync_bytes() || !gcn::aligned16(sync)) return SG2IM_ERR_ARG;
  if (S->n_triples < 1) return SG2IM_ERR_ARG;            // (no triples: use the per-layer entry points)
  if (!G->scratch || !gcn::aligned16(G->scratch) || G->scratch_bytes < sg2im_gconv_stack_backward_scratch(S)) return SG2IM_ERR_ARG;
  if (!G->d_triple || !gcn::aligned16(G->d_triple) || (G->d_obj && !gcn::aligned16(G->d_obj))) return SG2IM_ERR_ARG;
  if ((G->g_obj && !gcn::aligned16(G->g_obj)) || (G->g_pred && (!gcn::aligned16(G->g_pred) || (G->ld_gpred & 3)))) return SG2IM_ERR_ARG;
  for (int l = 0; l < S->n_layers; ++l) {
    const sg2im_gconv_grads& g = G->layer[l];
    const void* ps[] = {g.dw1a, g.db1a, g.dw1b, g.db1b, g.dw2a, g.db2a, g.dw2b, g.db2b};
    for (const void* p : ps) if (p && !gcn::aligned16(p)) return SG2IM_ERR_ARG;
  }
  if (gcn::prepare() != hipSuccess) return SG2IM_ERR_HIP;
  if (hipMemsetAsync(sync, 0, sizeof(unsigned) * gcn::kStickyWord, stream) != hipSuccess) return SG2IM_ERR_HIP;
  gcn::BwdArgs a;
  std::memcpy(&a.s, S, sizeof(*S));
  std::memcpy(&a.g, G, sizeof(*G));
  // scratch carve-up (element offsets from the base, every piece 16-byte aligned)
  size_t H = 0, NT = 0, D = 0;
  for (int l = 0; l < S->n_layers; ++l) {
    H = std::max<size_t>(H, S->layer[l].hidden); NT = std::max<size_t>(NT, 2 * (size_t)S->layer[l].hidden + S->layer[l].dout);
    D = std::max<size_t>(D, S->layer[l].din);
  }
  const size_t O = S->n_objs, T = S->n_triples;
  auto up4 = [](size_t n) { return (n + 3) / 4 * 4; };
  size_t off = up4(O * H);
  a.off_dpooled = (long long)off; off += up4(O * H);
  a.off_dnt = (long long)off; off += up4(T * NT);
  a.off_dp1 = (long long)off; off += up4(T * H);
  a.off_dtriple0 = (long long)off; off += up4(T * 3 * D);
  a.off_dtriple1 = (long long)off; off += up4(T * 3 * D);
  a.off_dobj0 = (long long)off; off += up4(O * D);
  a.off_dobj1 = (long long)off; off += up4(O * D);
  a.sync = static_cast<unsigned*>(sync);
  const int n_stages = 5 * S->n_layers;
  const bool staged = G->low_footprint >= 2, low = G->low_footprint == 1 || G->low_footprint == 2;
  for (int g = 0; g < (staged ? n_stages : 1); ++g) {
    a.stage_lo = staged ? g : 0;
    a.stage_hi = staged ? g + 1 : n_stages;
    if (low)
      SG2IM_LAUNCH(gcn::gcn_stack_bwd_low_kernel, dim3(gcn::grid_for(S, true)), dim3(gcn::kThreads),
                   gcn::LdsPlan<gcn::LowFootprint>::kBytes, stream, a);
    else
      SG2IM_LAUNCH(gcn::gcn_stack_bwd_kernel, dim3(gcn::grid_for(S, true)), dim3(gcn::kThreads), gcn::kLdsBytes, stream, a);
    if (hipGetLastError() != hipSuccess) return SG2IM_ERR_HIP;
  }
  return SG2IM_OK;
}

int sg2im_gconv_stack_stamps(const void* sync_host_copy, unsigned long long* out, int max_out) {
  // the device-clock stamps (100 MHz) workgroup 0 left in a sync area (host copy): kernel start, then (before,
  // after) each grid barrier, then the end; returns how many were copied
  if (!sync_host_copy || !out || max_out < 0) return 0;
  const unsigned long long* st = reinterpret_cast<const unsigned long long*>(static_cast<const unsigned*>(sync_host_copy) + gcn::kStamps);
  int n = 0;
  while (n < max_out && n < gcn::kMaxStamps && st[n] != 0) { out[n] = st[n]; ++n; }
  return n;
}

int sg2im_gconv_stack_status(const void* sync_host_copy) {
  // the error words of a sync area copied back to the host: 0 = every barrier of every launch that ever used this
  // area completed (bit 0: the last launch timed out; the rest: the sticky count of timed-out spins since the area
  // was created, shifted left by one)
  if (!sync_host_copy) return SG2IM_ERR_ARG;
  const unsigned* w = static_cast<const unsigned*>(sync_host_copy);
  const unsigned sticky = w[gcn::kStickyWord] > 0x3fffffffu ? 0x3fffffffu : w[gcn::kStickyWord];
  return (int)((w[gcn::kError] != 0 ? 1u : 0u) | (sticky << 1));
}

}  // extern "C"
