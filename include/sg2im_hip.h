/* sg2im_hip.h - C ABI of libsg2im_hip.so, the MI355X (gfx950) kernels behind the
 * sg2im training hot path.
 *
 * The reference (google/sg2im) has no FFI/plugin boundary: its hot path bottoms out in
 * ATen operator calls.  Each entry point below therefore cites the reference call site
 * (file:line under /root/reference) whose ATen operator sequence it replaces.  All
 * pointers are device pointers (HBM) unless stated; tensors are fp32, indices int64,
 * image-like activations are NHWC; every function is asynchronous on `stream`, keeps no
 * per-call state (the only process-wide state is the one-time kernel-attribute set-up that
 * sg2im_init() performs; it is not guarded against concurrent first calls from several host
 * threads - call sg2im_init() once before launching from more than one thread) and returns
 * SG2IM_OK (0) or an error code.  No torch types appear here; the
 * Python host (sg2im_amd/) binds this with ctypes.
 */
#ifndef SG2IM_HIP_H
#define SG2IM_HIP_H

#include <stddef.h>
#include <hip/hip_runtime_api.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SG2IM_OK 0
#define SG2IM_ERR_ARG 1   /* invalid argument (the reference would raise / assert) */
#define SG2IM_ERR_HIP 2   /* a HIP runtime call failed; see hipGetLastError() */

int sg2im_abi_version(void);   /* 11 (11: + sg2im_csr_build_triples; 10: bfloat16 storage, weight mirror) */

/* Statistics: kernels this library has launched (or recorded into a stream capture) so far in this process;
 * which = 0: all of them, 1: the implicit-GEMM family incl. its split-K finishes.  bench.py reads it around the
 * capture of one training iteration to report the launches per step of the graph-mode plan. */
unsigned long long sg2im_launch_count(int which);

/* One-time, idempotent set-up (kernel attributes of every implicit-GEMM instantiation, loading of
 * the library's code object).  Without it the same work happens lazily on first launches; with it
 * the entry points below only enqueue kernels / async memsets on `stream`, so they can be issued
 * inside a hipStreamBeginCapture region from the very first call (sg2im_amd/trainer.py captures
 * one hipGraph per batch-shape bucket).  The reference has no counterpart (eager ATen launches,
 * scripts/train.py:524-592). */
int sg2im_init(void);

/* ------------------------------------------------------------------------------------
 * Convolution / linear layers (implicit GEMM on the fp32 matrix cores).
 * Replaces nn.Conv2d / nn.Linear forward+backward as used by sg2im/crn.py:41-47,79-86,
 * sg2im/layers.py:178,221, sg2im/model.py:100,105, together with the torch.cat
 * (crn.py:63, graph.py:82, model.py:151), F.upsample (crn.py:107, model.py:98), row
 * gathers (graph.py:77-78, model.py:149-150) and the BatchNorm-apply + LeakyReLU of the
 * *previous* layer (layers.py:26,46), which are folded into the operand loader.
 *
 * A conv input is a virtual tensor: the channel concatenation of `nsrc` sources.
 * Weights are [cout][kh][kw][sum(channels)] (= a channels_last torch parameter).
 * A linear layer is kh=kw=1, in_h=in_w=out_h=out_w=1, batch=rows.
 * ---------------------------------------------------------------------------------- */
typedef struct sg2im_src {
  const float* data;        /* NHWC tensor [batch][in_h>>up][in_w>>up][ld] or rows [rows][ld] */
  const long long* gather;  /* optional: row r reads data[gather[r]] (linear geometry only) */
  const float* scale;       /* optional per-channel affine applied on load: v*scale+shift ... */
  const float* shift;
  float slope;              /* ... followed by leaky-relu with this slope (1 = identity) */
  int channels;
  int ld;                   /* floats between consecutive pixels / rows (>= channels) */
  int upsample_log2;        /* 0, or 1 = source is nearest-upsampled x2 on the fly */
  int dtype;                /* 0: `data` holds float; 1 (ABI 10): bfloat16 STORAGE - `data` points at bfloat16 elements (same
                             * NHWC layout, `ld` counts elements).  Only the bf16 halo'd 3x3 kernels read it (compute_dtype 1,
                             * stride 1, pad 1, maps that 128-pixel patches tile, no split-K: sg2im_conv_halo_unsplit);
                             * every other launch returns SG2IM_ERR_ARG instead of misreading the tensor.  A bfloat16 source
                             * must be readable 16 bytes past its last element (the weight gradient over sources of
                             * mixed types loads 16 bytes per lane where a bfloat16 lane needs 8). */
} sg2im_src;

typedef struct sg2im_conv_desc {
  sg2im_src src[4];
  int nsrc;
  int batch, in_h, in_w;    /* logical input size (after upsampling) */
  int out_h, out_w;
  int kh, kw, stride, pad;
  int compute_dtype;        /* 0: fp32 matrix cores (v_mfma_f32_32x32x2_f32, bit-equal to an fmaf chain);
                             * 1: operands rounded to bf16 (RNE) on their way into LDS, multiplied with
                             *    v_mfma_f32_32x32x16_bf16, fp32 accumulate - tensors in memory stay fp32.
                             *    Taken by vectorisable, gather-free launches (channels % 4 == 0, aligned);
                             *    the rest silently computes in fp32.  BASELINE.json configs[2..4]. */
  int launch_hints;         /* bit 0 (SG2IM_HINT_BACKGROUND), honoured by sg2im_conv2d_backward_weight: the launch
                             * is a leaf that runs next to latency-critical small kernels of another stream
                             * (sg2im_amd/trainer.py releases the refinement network's weight gradients under the
                             * layout / mask / graph-convolution backward chain) - its large-tile kernels then
                             * keep at most two workgroups resident per CU (padded LDS request) so that the other
                             * stream's workgroups always find a free slot.  Results are unaffected. */
  int weight_channels;      /* floats per tap of a weight row; 0 = the sum of the sources' channels.  Larger: the
                             * weight tensor has MORE input channels per tap than the sources supply and the
                             * convolution runs over the first sum(channels) of every tap - forward and
                             * backward_data read weight[co][tap][0 .. sum), backward_weight writes the same
                             * columns of dweight (row = kh * kw * weight_channels floats) and leaves the others
                             * untouched.  Use: the refinement network's first module concatenates a constant
                             * all-zero feature channel (crn.py:105) - 161 channels, not a multiple of 4, which
                             * would push the layer onto the scalar loaders; the zero channel contributes nothing
                             * forward and has an exactly zero weight gradient, so it is simply left out. */
  int out_dtype;            /* (ABI 10) 1: the result - `out` of sg2im_conv2d_forward[_bn], `dx` of sg2im_conv2d_backward_data[_bn]
                             * - is written as bfloat16 (RNE of the fp32 value that the fused BatchNorm reductions still see
                             * unrounded); `ld_out` / `ld_dx` count elements.  Same launches as sg2im_src.dtype = 1. */
  int dy_dtype;             /* (ABI 10) 1: `dy` of sg2im_conv2d_backward_data[_bn] / sg2im_conv2d_backward_weight holds
                             * bfloat16 (the weight gradient: the bf16 halo'd kernel, 3x3 over maps that 4 x 16 patches tile) */
  const void* weight_bf16;  /* optional (ABI 10): a bfloat16 copy of `weight` - same layout, element i = RNE(weight[i]),
                             * readable 16 bytes past its last element - e.g. made by sg2im_cast_f32_to_bf16 once per
                             * optimiser step.  With compute_dtype 1 the halo'd 3x3 kernels (forward, backward_data) then
                             * load their weight slices from it: half the bytes of the stream that bounds them, no
                             * conversion in the loader, bit-identical results.  NULL: the loaders round `weight`. */
} sg2im_conv_desc;
#define SG2IM_HINT_BACKGROUND 1

/* Limits (SG2IM_ERR_ARG otherwise): every source tensor below 4 GiB and dy below 2 GiB - the loaders form
 * addresses from 32-bit byte offsets off a wave-uniform base, dy is read through a buffer resource whose
 * out-of-range offset (2 GiB) makes invalid rows read as zeros; pending LeakyReLU slopes in [0, 1] (evaluated
 * as max(v, v * slope)). */
/* out[pix][co] = leaky_{out_slope}( conv(X, W)[pix][co] + bias[co] ) (+ out if accumulate) */
int sg2im_conv2d_forward(const sg2im_conv_desc* desc, const float* weight, int cout,
                         const float* bias, float out_slope, float* out, long long ld_out,
                         int accumulate, float* workspace, size_t workspace_bytes,
                         hipStream_t stream);
/* dX[inpix][c - c_begin] for concat channels c in [c_begin, c_begin + c_count), written with
 * row stride ld_dx.  `desc` is the forward descriptor (sources are not dereferenced). */
int sg2im_conv2d_backward_data(const sg2im_conv_desc* desc, const float* weight, int cout,
                               const float* dy, int ld_dy, int c_begin, int c_count, float* dx,
                               long long ld_dx, int accumulate, float* workspace,
                               size_t workspace_bytes, hipStream_t stream);
/* dX of sg2im_conv2d_backward_data (accumulate = 0) multiplied by leaky'_slope of the layer it flows into:
 *   dx[r][c] *= (act[r * ld_act + c] > 0 ? 1 : slope),   act = that layer's ACTIVATED output, rows / columns as dx
 * - autograd of  conv(leaky(y))  w.r.t. the pre-activation y (nn.ReLU / nn.LeakyReLU between two layers without a
 * norm: sg2im/layers.py:216-232 build_mlp, sg2im/graph.py:47-54 net1 / net2, sg2im/crn.py:79-86 output_conv) in the
 * launches of the data gradient: its epilogue or, with split-K, its finish applies the mask (the arithmetic of
 * sg2im_act_backward on the finished sum, bit for bit); launches that cannot (<= 4 input channels, the stride-2
 * parity form) run sg2im_act_backward over dx internally, which then must be dense (ld_dx == c_count). */
int sg2im_conv2d_backward_data_act(const sg2im_conv_desc* desc, const float* weight, int cout, const float* dy,
                                   int ld_dy, int c_begin, int c_count, float* dx, long long ld_dx, const float* act,
                                   long long ld_act, float slope, float* workspace, size_t workspace_bytes,
                                   hipStream_t stream);
/* dW[co][kh][kw][c] = sum_pix dY[pix][co] * X[pix (+) tap][c]  (+ dW if accumulate)
 * dbias (optional, may be NULL): dB[co] = sum_pix dY[pix][co]  (+ dB if accumulate) - the
 * bias gradient of the same layer, produced in the same pass over dY. */
int sg2im_conv2d_backward_weight(const sg2im_conv_desc* desc, const float* dy, int ld_dy, int cout,
                                 float* dweight, float* dbias, int accumulate, float* workspace,
                                 size_t workspace_bytes, hipStream_t stream);
/* Up to 4 weight gradients (+ bias gradients; dbiases[i] may be NULL) as ONE launch + ONE split-K finish launch:
 * the four linear layers of a GraphTripleConv layer (graph.py:60-71 under autograd) - leaves of the backward
 * graph that otherwise cost 4 + 4 launches inside a chain of small dependent kernels.  Results equal those of n
 * calls of sg2im_conv2d_backward_weight (compute_dtype 0) up to the fp32 summation order of the split-K plan
 * (always 64x64 tiles here); deterministic.  Every problem must take the float4 loaders (channels,
 * ld_dy, cout multiples of 4; 16-byte aligned buffers; weight_channels 0) - SG2IM_ERR_ARG with nothing launched
 * otherwise, the caller then issues the single calls.  The problems share `workspace` (disjoint slices). */
int sg2im_conv2d_backward_weight_group(int n, const sg2im_conv_desc* const* descs, const float* const* dys,
                                       const int* ld_dys, const int* couts, float* const* dweights,
                                       float* const* dbiases, int accumulate, float* workspace,
                                       size_t workspace_bytes, hipStream_t stream);
/* ------------------------------------------------------------------------------------
 * Convolution + the BatchNorm reductions that follow / precede it in ONE set of launches
 * (sg2im/crn.py:41-47: Conv2d -> BatchNorm2d -> LeakyReLU twice per refinement module; sg2im/layers.py:166-178;
 * sg2im/model.py:99-101).  Training-mode BatchNorm couples every convolution to a per-channel reduction over its
 * whole output; instead of a separate pass over the tensor the reductions ride in the GEMM: the workgroup that
 * owns an output tile (or, with split-K, the finish launch that sums the partials) also produces that tile's
 * per-channel partial sums, and one small launch finishes them in double.  No atomics, fixed order: results are
 * reproducible and equal sg2im_conv2d_forward + sg2im_bn_stats (resp. sg2im_conv2d_backward_data + the first two
 * passes of sg2im_bn_act_backward) up to the fp32 summation order of the statistics.  Launches that do not
 * qualify (scalar loaders, row gathers, the stride-2 parity form, <= 4 output channels) run exactly those
 * two calls internally.
 * ---------------------------------------------------------------------------------- */
typedef struct sg2im_bn_fwd {
  const float* gamma;       /* [cout] or NULL (1) */
  const float* beta;        /* [cout] or NULL (0) */
  float eps, momentum;
  int training;             /* 0: running statistics are used (no reduction at all); n >= 1: batch statistics, the
                               running statistics move as after n passes over this batch (see sg2im_bn_stats) */
  float* running_mean;      /* updated when training (may be NULL then) */
  float* running_var;
  long long* num_batches_tracked;
  long long unbiased_rows;  /* 0 = the output's rows; see sg2im_bn_stats */
  float* mean; float* invstd; float* scale; float* shift;   /* outputs, [cout] each */
  float* partial;           /* scratch, partial_floats floats: >= 3 * cout * ceil(rows / 64) takes every fused form,
                             * >= 2 * cout * 1024 is required (the standalone fallback) */
  size_t partial_floats;
  const int* count;         /* padded row batch: count[0] * count_unit real rows (NULL: all), see sg2im_bn_stats */
  int count_unit;
} sg2im_bn_fwd;
/* out = leaky_{out_slope}(conv(X, W) + bias) and the batch statistics of `out` (rows = batch * out_h * out_w,
 * channels = cout, row stride ld_out) as sg2im_bn_stats computes them */
int sg2im_conv2d_forward_bn(const sg2im_conv_desc* desc, const float* weight, int cout, const float* bias,
                            float out_slope, float* out, long long ld_out, float* workspace,
                            size_t workspace_bytes, const sg2im_bn_fwd* bn, hipStream_t stream);

typedef struct sg2im_bn_bwd {
  const float* y;           /* pre-normalisation output of the BatchNorm'd layer, [rows][ld_y], channels = c_count */
  long long ld_y;
  int pool2;                /* 1: dx (this launch's result) is at TWICE y's resolution - the layer's activated output
                             * was nearest-upsampled x2 into the convolution (crn.py:107): its gradient is the 2x2 sum */
  const float* gamma;       /* [C] or NULL */
  const float* mean; const float* invstd; const float* scale; const float* shift;   /* from the forward pass */
  float slope;              /* LeakyReLU slope of the activation behind the norm */
  int training;             /* 0: statistics are constants (eval-mode BatchNorm) */
  float* dgamma; float* dbeta;   /* [C] each, may be NULL; += when accumulate */
  int accumulate;
  float* coef;              /* out, float[3 * C]: dy = coef[0][c] * du + coef[1][c] * y + coef[2][c], the input of
                             * sg2im_bn_backward_apply */
  float* partial;           /* scratch, partial_floats floats: >= 2 * C * ceil(rows_of_dx / 64) takes every fused form */
  size_t partial_floats;
  const int* count;         /* padded row batch: count[0] * count_unit real rows OF y (NULL: all) */
  int count_unit;
  int y_dtype;              /* (ABI 10) 1: `y` holds bfloat16 (ld_y in elements); launches as sg2im_src.dtype = 1 */
} sg2im_bn_bwd;
/* dx = the data gradient of sg2im_conv2d_backward_data (accumulate = 0) for the channel range [c_begin,
 * c_begin + c_count), which is the gradient w.r.t. the ACTIVATED output z = leaky(scale * y + shift) of a
 * BatchNorm'd layer - plus that BatchNorm's backward reductions: dgamma, dbeta and the coefficients `coef`. */
int sg2im_conv2d_backward_data_bn(const sg2im_conv_desc* desc, const float* weight, int cout, const float* dy,
                                  int ld_dy, int c_begin, int c_count, float* dx, long long ld_dx,
                                  float* workspace, size_t workspace_bytes, const sg2im_bn_bwd* bn,
                                  hipStream_t stream);
/* third pass of sg2im_bn_act_backward on its own: dy[rows][C] = coef[0] * du + coef[1] * y + coef[2] with
 * du = dz * leaky'(scale * y + shift), dz read from g as in sg2im_bn_act_backward (pool2: 2x2 sums);
 * padding rows (count / count_unit) get 0 */
int sg2im_bn_backward_apply(const float* g, long long ld_g, int pool2, int batch, int h, int w, const float* y,
                            long long ld_y, int channels, const float* scale, const float* shift, float slope,
                            const float* coef, float* dy, const int* count, int count_unit, hipStream_t stream);

/* sg2im_bn_backward_apply with bfloat16 STORAGE of any of g (dz), y, dy (x_dtype 1; ld_* in elements; channels, ld_g, ld_y
 * multiples of 4, 16-byte aligned): fp32 arithmetic, dy rounded (RNE) when it holds bfloat16; no padded-batch count. */
int sg2im_bn_backward_apply_ex(const void* g, long long ld_g, int pool2, int batch, int h, int w, const void* y,
                               long long ld_y, int channels, const float* scale, const float* shift, float slope,
                               const float* coef, void* dy, int g_dtype, int y_dtype, int dy_dtype, hipStream_t stream);
/* 1 when the halo'd 3x3 kernels run a (batch, h, w) map with `cols` output columns and `chunks` 32-channel reduction
 * chunks WITHOUT split-K (the launches that can carry bfloat16 storage keep every CU busy on their own), else 0.  The
 * host side asks before it chooses the storage type of a layer's tensors (sg2im_amd.functional.RefinementFn). */
int sg2im_conv_halo_unsplit(int batch, int h, int w, int cols, int chunks);

/* out[n] = sum_m x[m][n] (+ out): bias gradients (autograd of Conv2d/Linear bias) */
/* partial: scratch float[2 * cols * 1024] */
int sg2im_column_sum(const float* x, long long rows, int cols, long long ld, float* out,
                     int accumulate, float* partial, hipStream_t stream);

/* ------------------------------------------------------------------------------------
 * Triple-indexed gather / scatter of GraphTripleConv (sg2im/graph.py:73-114) and the
 * embedding lookups (sg2im/model.py:131,133).
 * ---------------------------------------------------------------------------------- */
/* Stable CSR over destination rows.  Entries e in [0, n_a) come from keys_a (entry id e),
 * entries in [n_a, n_a+n_b) from keys_b (entry id n_a + index).  Row j lists its entry ids
 * in increasing order, i.e. all keys_a hits in index order, then all keys_b hits - the
 * accumulation order of the reference's two scatter_add calls (graph.py:98-99).
 * row_ptr: int[n_rows+1]; entries: int[n_a+n_b]; scratch: int[n_rows + n_a + n_b].
 * live_keys (device, may be NULL): only the first live_keys[0] keys of EACH key array take part - the
 * rest is the padding of a bucketed batch (see sg2im_bn_stats); entries[] is then only filled up to
 * row_ptr[n_rows]. */
int sg2im_csr_build(const long long* keys_a, int n_a, const long long* keys_b, int n_b, int n_rows,
                    int* row_ptr, int* entries, int* scratch, const int* live_keys, hipStream_t stream);
/* The pooling CSR of a GraphTripleConv stack straight from the (n_triples, 3) int64 triples tensor (graph.py:73-75
 * chunks it into s, p, o): sg2im_csr_build(s, T, o, T, n_rows, ...) AND the three columns as contiguous arrays
 * split[0..T) = s, split[T..2T) = p, split[2T..3T) = o, in one launch for the sizes of a training batch (the head of
 * the critical path of the step: three strided copies + the build before).  scratch / live_keys as above. */
int sg2im_csr_build_triples(const long long* triples, int n_triples, int n_rows, long long* split, int* row_ptr,
                            int* entries, int* scratch, const int* live_keys, hipStream_t stream);
/* out[j][0:width] = sum over row j's entries, in CSR order, starting from +0.0f, of
 *   src_a[e*ld_a + 0:width]            (e <  n_a)
 *   src_b[(e-n_a)*ld_b + 0:width]      (e >= n_a)
 * then, if average != 0, divided by max(1, #entries)  (graph.py:92-114: pooling sum/avg);
 * accumulate != 0 adds the result to out instead of overwriting it (gradient arenas).
 * Bit-exact w.r.t. the sequential fp32 order rule (oracle.gconv_pool_sequential). */
int sg2im_segment_sum(const float* src_a, long long ld_a, int n_a, const float* src_b, long long ld_b,
                      const int* row_ptr, const int* entries, int n_rows, int width, int average,
                      int accumulate, float* out, long long ld_out, hipStream_t stream);
/* dst[i][0:width] = src[idx[i]][0:width] (bit-exact copy) ; if row_ptr != NULL the row is
 * divided by max(1, row_ptr[idx[i]+1]-row_ptr[idx[i]])  (backward of the 'avg' pooling). */
int sg2im_gather_rows(const float* src, long long ld_src, const long long* idx, int n, int width,
                      const int* row_ptr, float* dst, long long ld_dst, hipStream_t stream);
/* Backward of the GraphTripleConv pooling block (graph.py:98-114 under autograd) in one launch:
 * d_new_t[t] = [ d_pooled[s[t]] (/ count) | g_pred[t] (zeros if NULL) | d_pooled[o[t]] (/ count) ] * leaky'_slope(new_t[t])
 * with count = max(1, row_ptr[r+1] - row_ptr[r]) when row_ptr != NULL ('avg' pooling), new_t the ACTIVATED net1
 * output [T][2*hidden + dout].  Equals sg2im_gather_rows x2 + sg2im_copy_2d + sg2im_act_backward, bit for bit. */
int sg2im_gconv_pool_backward(const float* d_pooled, long long ld_dp, const long long* s_idx,
                              const long long* o_idx, int n_triples, const int* row_ptr, const float* g_pred,
                              long long ld_gp, const float* new_t, long long ld_nt, int hidden, int dout,
                              float slope, float* d_new_t, long long ld_out, hipStream_t stream);
/* One whole GraphTripleConv layer (sg2im/graph.py:56-120) per call: the launch sequence of the entry points above -
 *   forward : gather + concat folded into net1's first GEMM (row-gathered sources), net1 (Linear-ReLU-Linear-ReLU),
 *             CSR pooling (sum / avg, the reference's accumulation order, bit-exact), net2;
 *   backward: the mirror image; the pooling backward + ReLU of net1's output is one launch, the four weight
 *             (+ bias) gradients are ONE grouped launch + one finish (sg2im_conv2d_backward_weight_group)
 * on `stream`, with caller-owned activation / scratch buffers (nothing is allocated, no state is kept).
 * Shapes: obj_vecs [n_objs][din], pred_vecs [n_triples][din]; w1a [hidden][3 din], w1b [2 hidden + dout][hidden],
 * w2a [hidden][hidden], w2b [dout][hidden] (nn.Linear layout, graph.py:60-71); h1 [T][hidden],
 * new_t [T][2 hidden + dout] (activated net1 output: columns [0, hidden) feed the subjects' pool,
 * [hidden, hidden + dout) ARE new_pred_vecs, the rest feeds the objects' pool), pooled / h2 [O][hidden],
 * new_obj [O][dout].  row_ptr / entries: sg2im_csr_build(s_idx, T, o_idx, T, n_objs, ...). */
typedef struct sg2im_gconv_layer {
  const float* obj_vecs; long long ld_obj;
  const float* pred_vecs; long long ld_pred;
  const long long* s_idx; const long long* o_idx;
  const int* row_ptr; const int* entries;
  int n_objs, n_triples, din, hidden, dout;
  int average;                         /* 1: 'avg' pooling (graph.py:101-112), 0: 'sum' */
  const float *w1a, *b1a, *w1b, *b1b, *w2a, *b2a, *w2b, *b2b;
} sg2im_gconv_layer;
typedef struct sg2im_gconv_grads {     /* parameter gradients (any may be NULL = not wanted); += when accumulate */
  float *dw1a, *db1a, *dw1b, *db1b, *dw2a, *db2a, *dw2b, *db2b;
  int accumulate;
} sg2im_gconv_grads;
int sg2im_gconv_layer_forward(const sg2im_gconv_layer* layer, float* h1, float* new_t, float* pooled, float* h2,
                              float* new_obj, float* workspace, size_t workspace_bytes, hipStream_t stream);
/* g_obj [O][dout] (NULL = zeros), g_pred [T][>= dout] with row stride ld_gpred (NULL = zeros): gradients w.r.t.
 * new_obj / new_pred_vecs.  d_triple [T][3 din] (out): gradient w.r.t. the gathered net1 input - its columns
 * [din, 2 din) are the gradient w.r.t. pred_vecs; d_obj [O][din] (out, may be NULL): gradient w.r.t. obj_vecs.
 * scratch: sg2im_gconv_layer_backward_scratch() bytes. */
size_t sg2im_gconv_layer_backward_scratch(int n_objs, int n_triples, int din, int hidden, int dout);
int sg2im_gconv_layer_backward(const sg2im_gconv_layer* layer, const float* h1, const float* new_t,
                               const float* pooled, const float* h2, const float* new_obj, const float* g_obj,
                               const float* g_pred, long long ld_gpred, float* d_triple, float* d_obj,
                               const sg2im_gconv_grads* grads, float* scratch, size_t scratch_bytes,
                               float* workspace, size_t workspace_bytes, hipStream_t stream);

/* The whole GraphTripleConv STACK - sg2im/model.py:136-140: `gconv` followed by `gconv_net`'s layers, each layer
 * sg2im/graph.py:56-120 - as ONE persistent launch per direction (csrc/gcn_persist.hip): <= one workgroup per CU
 * stays resident and walks the stages of all layers, separated by XCD-hierarchical grid barriers; gather + concat
 * and the CSR pool (bit-exact, the reference's accumulation order) run inside the GEMM operand loaders.
 * Layer l reads layer l-1's new_obj / new_t[:, hidden:hidden+dout] (layer 0: obj_vecs / pred_vecs) and writes its
 * activations h1 [T][hidden], new_t [T][2 hidden + dout], pooled [O][hidden], h2 [O][hidden], new_obj [O][dout]
 * (dense, caller-owned; what backward reads).  Requirements (else SG2IM_ERR_ARG - use the per-layer entry points):
 * din, hidden, dout multiples of 32 (sg2im_gconv_stack_supported), 16-byte aligned pointers, mlp_normalization
 * 'none'.  sync: >= sg2im_gconv_stack_sync_bytes() of device memory private to the call.  THE CALLER ZEROES THE WHOLE AREA
 * ONCE, when it allocates it (hipMemset; sg2im_amd.ops.sync_area uses torch.zeros); every launch then re-zeroes all
 * of it but ONE word on `stream` (see below); after completion word 64 is non-zero iff a grid barrier of THIS launch
 * timed out (never on a healthy device with the grid fully resident: every spin is bounded instead of hanging the
 * queue); word 2040 is a STICKY count of timed-out spins over all launches that ever used the area - the launcher's
 * memset leaves it alone, so a caller that only looks every now and then (sg2im_amd.trainer does, wherever it
 * synchronises with the host anyway) cannot miss a launch that produced garbage.  sg2im_gconv_stack_status() of a
 * host copy: 0 = healthy; bit 0 = the last launch timed out, bits 1.. = the sticky count.  Only ONE persistent
 * launch may be in flight per device at a time (two whole-chip resident grids can starve each other's barriers);
 * the Trainer issues the forward on its main lane only; its backward is chosen per configuration
 * (SG2IM_GCN_PERSIST_BWD=auto, sg2im_amd.trainer.Trainer._gcn_backward_mode): the one-launch low_footprint form where
 * the small-kernel tail ends the step (VG-style batches / a trained mask_net, either compute mode), layer by layer otherwise.
 * A barrier that times out (grid not fully resident: another process on the GPU) lets the launch carry on with
 * incomplete data - its results are garbage; the sticky word is what reports it (sg2im_gconv_stack_status, checked by
 * the Trainer wherever it synchronises with the host: Trainer.losses_to_host). */
#define SG2IM_GCONV_MAX_LAYERS 8
typedef struct sg2im_gconv_stack_layer {
  const float *w1a, *b1a, *w1b, *b1b, *w2a, *b2a, *w2b, *b2b;   /* nn.Linear layout, see sg2im_gconv_layer */
  float *h1, *new_t, *pooled, *h2, *new_obj;
  int din, hidden, dout, reserved;
} sg2im_gconv_stack_layer;
typedef struct sg2im_gconv_stack {
  const float* obj_vecs; long long ld_obj;
  const float* pred_vecs; long long ld_pred;
  const long long* s_idx; const long long* o_idx;
  const int* row_ptr; const int* entries;        /* sg2im_csr_build(s_idx, T, o_idx, T, n_objs, ...) */
  int n_objs, n_triples, n_layers;
  int average;                                   /* 1: 'avg' pooling, 0: 'sum' */
  sg2im_gconv_stack_layer layer[SG2IM_GCONV_MAX_LAYERS];
} sg2im_gconv_stack;
size_t sg2im_gconv_stack_sync_bytes(void);
int sg2im_gconv_stack_supported(int din, int hidden, int dout);
int sg2im_gconv_stack_forward(const sg2im_gconv_stack* stack, void* sync, size_t sync_bytes, hipStream_t stream);
/* Backward of the stack, ONE persistent launch: last layer first; per layer the data gradients of net2 / the pool +
 * concat (rebuilt in the operand loader) / net1, and the eight parameter gradients as extra tiles of the stage that
 * has their operands (+= when layer[l].accumulate; a NULL pointer skips that gradient).
 * g_obj [O][dout_last] dense (NULL = zeros), g_pred [T][>= dout_last] with row stride ld_gpred (NULL = zeros): gradients
 * w.r.t. the stack's outputs (new_obj of the last layer, its new_t[:, hidden:hidden+dout]).  d_triple [T][3 din_0]
 * (out, required): gradient w.r.t. layer 0's gathered net1 input - columns [din_0, 2 din_0) are the gradient w.r.t.
 * pred_vecs; d_obj [O][din_0] (out, may be NULL): gradient w.r.t. obj_vecs.  scratch:
 * sg2im_gconv_stack_backward_scratch() bytes.  Requires n_triples >= 1 (else SG2IM_ERR_ARG: use the per-layer calls). */
typedef struct sg2im_gconv_stack_grads {
  const float* g_obj; const float* g_pred; long long ld_gpred;
  float* d_triple; float* d_obj;
  float* scratch; size_t scratch_bytes;
  sg2im_gconv_grads layer[SG2IM_GCONV_MAX_LAYERS];
  int low_footprint;        /* 0: the kernel that wants WHOLE CUs (one workgroup per CU, ~390 registers, 98 KB of LDS: fastest
                             * on an otherwise idle GPU, 419 us at the bench shape); 1: <= 168 registers / 41 KB of LDS, 32 x 32
                             * tiles - its workgroups fit NEXT TO resident workgroups of other kernels, so the launch can
                             * run inside a busy multi-stream graph (sg2im_amd/trainer.py: under the refinement network's
                             * weight gradients) without waiting for whole CUs to drain.  Same results up to the fp32
                             * summation order of the weight gradients' tiles.
                             * 2 / 3 (ABI 11): STAGED - the same stages as 5 n_layers ordinary launches of kernel 1 / 0,
                             * one stage each: no grid barrier executes, nothing has to be co-resident and nothing polls
                             * (the sync area is only counted in).  Bit-identical to the one-launch form of the same kernel. */
  int reserved;
} sg2im_gconv_stack_grads;
size_t sg2im_gconv_stack_backward_scratch(const sg2im_gconv_stack* stack);
int sg2im_gconv_stack_backward(const sg2im_gconv_stack* stack, const sg2im_gconv_stack_grads* grads, void* sync,
                               size_t sync_bytes, hipStream_t stream);
int sg2im_gconv_stack_status(const void* sync_host_copy);
/* diagnostics: the 100 MHz device-clock stamps workgroup 0 left in the sync area (host copy): kernel start, then
 * (before, after) every grid barrier, then the end; returns the number copied into out[0..max_out) */
int sg2im_gconv_stack_stamps(const void* sync_host_copy, unsigned long long* out, int max_out);

/* ------------------------------------------------------------------------------------
 * Two nn.Linear heads over the same row matrix (sg2im/discriminators.py:66-75: AcDiscriminator's real_classifier and
 * obj_classifier both read the pooled 1024-vector) in one launch per direction:
 *   forward        y1 [rows][n1] = x w1^T + b1,  y2 [rows][n2] = x w2^T + b2        (b1 / b2 may be NULL)
 *   backward_data  dx [rows][k]  = g1 w1 + g2 w2     (what autograd sums from the two heads' input gradients)
 * w1 [n1][k], w2 [n2][k]: nn.Linear layout, 16-byte aligned; k % 4 == 0, k <= 1536, n1 + n2 <= 2048 (sg2im_two_heads_supported);
 * ldx / lddx multiples of 4.  Fixed summation order.  Weight / bias gradients: sg2im_conv2d_backward_weight[_group]
 * on the 1x1 geometry, as for every other nn.Linear.
 * ---------------------------------------------------------------------------------- */
int sg2im_two_heads_supported(int k, int n1, int n2);
int sg2im_two_heads_forward(const float* x, long long ldx, int rows, int k, const float* w1, const float* b1, int n1,
                            const float* w2, const float* b2, int n2, float* y1, long long ld1, float* y2,
                            long long ld2, hipStream_t stream);
int sg2im_two_heads_backward_data(const float* g1, long long ldg1, const float* g2, long long ldg2, int rows, int k,
                                  const float* w1, int n1, const float* w2, int n2, float* dx, long long lddx,
                                  hipStream_t stream);

/* Up to 16 plain device-to-device copies (sizes and addresses multiples of 4 bytes) in ONE launch: the hand-over
 * of a collated batch (scripts/train.py:514-519 `batch = [tensor.cuda() for tensor in batch]`) into the static
 * input buffers a captured iteration reads - images, object / triple / mask arrays and their padding, the two
 * row counts.  dst / src / bytes are HOST arrays of n entries. */
int sg2im_stage_batch(int n, void* const* dst, const void* const* src, const size_t* bytes, hipStream_t stream);
/* Diagnostics (bench.py's instrumented pass): while `event` is set (per host thread; NULL clears it), the *_bn
 * entry points record it on their stream between their GEMM launches (incl. a split-K finish) and their BatchNorm
 * finish launch and set *recorded = 1 - a timer bracketing the call can then attribute the two parts separately. */
int sg2im_debug_mark_gemm_end(hipEvent_t event, int* recorded);
/* Diagnostics: one single-thread launch that stores the device's constant-rate clock (wall_clock64(), 100 MHz) in
 * *slot when the stream reaches it - schedule marks inside a captured iteration (Trainer, SG2IM_MARKS=1). */
int sg2im_timestamp(unsigned long long* slot, hipStream_t stream);

/* dst[r][0:width] = src[r][0:width] for strided row matrices (the new_p column slice of the
 * net1 output, graph.py:88, travelling through backward) */
int sg2im_copy_2d(const float* src, long long ld_src, float* dst, long long ld_dst, long long rows,
                  int width, hipStream_t stream);

/* ------------------------------------------------------------------------------------
 * Scene layout (sg2im/layout.py:30-162) - boxes_to_layout / masks_to_layout fused:
 * grid construction, bilinear sampling (zeros padding) and the per-image scatter_add.
 *   layout[n][y][x][d] = sum_{o : obj_to_img[o]=n} vecs[o][d] * S_o(y,x)
 * S_o = bilinear sample of masks[o] (or of an all-ones 8x8 map when masks == NULL) at the
 * grid of layout.py:94-128.  Objects are visited in index order per image (img CSR).
 * masks: float [O][M][M] or, when masks_i64 != NULL, int64 (the GT masks, layout.py:87).
 * ---------------------------------------------------------------------------------- */
int sg2im_layout_forward(const float* vecs, long long ld_vecs, const float* boxes,
                         const float* masks, const long long* masks_i64, int mask_size,
                         const int* img_row_ptr, const int* img_entries, int n_images, int n_objs,
                         int dim, int height, int width, int align_corners, float* layout,
                         long long ld_layout, hipStream_t stream);
/* The same layout, the layout-noise channels (model.py:164-169: torch.cat([layout, noise], dim=1); `noise` is
 * the NCHW tensor [n_images][noise_dim][H][W], may be NULL with noise_dim 0) and the average-pool pyramid the
 * refinement network builds from it (crn.py:58-62: F.avg_pool2d(layout, factor) per module) in ONE pass:
 * levels[0] = NHWC [n_images][H][W][ld_levels] with channels [0, dim) = layout, [dim, dim + noise_dim) = noise;
 * levels[l] (1 <= l <= n_levels <= 4) = [n_images][H >> l][W >> l][ld_levels], the 2 x 2 mean of levels[l - 1]
 * (== sg2im_avgpool_forward(levels[l - 1], 2), bit for bit).  `levels` is a HOST array of n_levels + 1 device
 * pointers.  Requires dim % 32 == 0, noise_dim % 32 == 0, H % 16 == 0, W % 16 == 0 (SG2IM_ERR_ARG otherwise:
 * use sg2im_layout_forward + sg2im_nchw_to_nhwc + sg2im_avgpool_forward). */
int sg2im_layout_pyramid_forward(const float* vecs, long long ld_vecs, const float* boxes, const float* masks,
                                 const long long* masks_i64, int mask_size, const int* img_row_ptr,
                                 const int* img_entries, int n_images, int dim, const float* noise, int noise_dim,
                                 int height, int width, int align_corners, int n_levels, float* const* levels,
                                 long long ld_levels, hipStream_t stream);
/* d_vecs[o][d] = sum_{y,x} dlayout[n_o][y][x][d] * S_o(y,x)  (deterministic two-stage sum;
 * workspace: sg2im_layout_backward_workspace() bytes);  optional d_masks [O][M][M] for float
 * (predicted) masks: d_masks[o][i][j] = sum_{y,x} <dlayout[n_o][y][x], vecs[o]> * dS_o/dm_ij;
 * optional d_boxes [O][4]: the gradient w.r.t. the boxes through the sampling grid
 * (layout.py:117-127; Sg2ImModel.forward with boxes_gt=None lays out the PREDICTED boxes). */
size_t sg2im_layout_backward_workspace(int n_objs, int dim, int height, int width);
int sg2im_layout_backward(const float* dlayout, long long ld_dlayout, const float* vecs,
                          long long ld_vecs, const float* boxes, const float* masks,
                          const long long* masks_i64, int mask_size, const long long* obj_to_img,
                          const int* img_row_ptr, const int* img_entries, int n_images,
                          int n_objs, int dim, int height, int width, int align_corners,
                          float* d_vecs, long long ld_dvecs, float* d_masks, float* d_boxes,
                          float* workspace, hipStream_t stream);

/* d_vecs of the layout (the d_vecs part of sg2im_layout_backward) straight from the refinement network's PER-LEVEL
 * layout gradients: dlevels[l] [n_images][height / factors[l]][width / factors[l]][>= dim] with row stride lds[l] is the
 * gradient w.r.t. the layout average-pooled by factors[l] (crn.py:58-62; factors powers of two, <= 6 levels), so
 * d layout = sum_l upsample(dlevels[l]) / factors[l]^2 - what sg2im_pyramid_backward would write - is summed on the fly
 * and never materialised.  workspace: sg2im_layout_backward_workspace() bytes.  dim a multiple of 4. */
int sg2im_layout_backward_vecs_levels(const float* const* dlevels, const int* factors, const long long* lds, int n_levels,
                                      const float* boxes, const float* masks, const long long* masks_i64, int mask_size,
                                      const int* img_row_ptr, const int* img_entries, int n_images, int n_objs, int dim,
                                      int height, int width, int align_corners, float* d_vecs, long long ld_dvecs,
                                      float* workspace, hipStream_t stream);
/* d_masks / d_boxes of the layout (the other half of sg2im_layout_backward) from the same PER-LEVEL gradients:
 * G_o(y, x) = <d layout[obj_to_img[o], y, x, :], vecs[o]> is formed per image from one pass over the levels (summed on the
 * fly, as above), then transposed through the bilinear footprint into d_masks [O][M][M] (float masks only) and / or
 * d_boxes [O][4] (layout.py:60-61,87-88,117-127 through grid_sample's backward).  VG-style training, where mask_net is
 * trained through the layout (model.py:146-147,152-157): the step then never materialises the full-resolution layout
 * gradient.  dim a multiple of 4, <= 128; vecs 16-byte aligned with ld_vecs a multiple of 4 (SG2IM_ERR_ARG otherwise: the
 * caller materialises the gradient and calls sg2im_layout_backward).  workspace: sg2im_layout_backward_workspace() bytes. */
int sg2im_layout_backward_maps_levels(const float* const* dlevels, const int* factors, const long long* lds, int n_levels,
                                      const float* vecs, long long ld_vecs, const float* boxes, const float* masks,
                                      const long long* masks_i64, int mask_size, const int* img_row_ptr,
                                      const int* img_entries, int n_images, int n_objs, int dim, int height, int width,
                                      int align_corners, float* d_masks, float* d_boxes, float* workspace,
                                      hipStream_t stream);
/* Object crops for the object discriminator (sg2im/bilinear.py:28-132, 'cudnn' path):
 * crops[o] = bilinear sample of image obj_to_img[o] on linspace(2*x0-1, 2*x1-1, size).
 * imgs are NHWC [N][H][W][C] (row stride ld_img); crops NHWC [O][size][size][C]. */
int sg2im_crop_forward(const float* imgs, long long ld_img, int n_images, int height, int width,
                       int channels, const float* boxes, const long long* obj_to_img, int n_objs,
                       int size, int align_corners, float* crops, hipStream_t stream);
/* Transpose of the above without atomics, in a fixed summation order: per-object partial planes
 * (workspace: sg2im_crop_backward_workspace bytes) are summed per image pixel over the image's
 * objects in ascending order.  Every element of d_imgs [N][H][W] (row stride ld_dimg) is
 * written - zero where no crop touches it. */
size_t sg2im_crop_backward_workspace(int n_objs, int height, int width, int channels);
int sg2im_crop_backward(const float* d_crops, int n_images, int height, int width, int channels,
                        const float* boxes, const long long* obj_to_img, int n_objs, int size,
                        int align_corners, float* d_imgs, long long ld_dimg, float* workspace,
                        hipStream_t stream);

/* ------------------------------------------------------------------------------------
 * BatchNorm2d (training statistics), pooling, layout-pyramid helpers
 * (sg2im/layers.py:22-31, sg2im/crn.py:53-64, sg2im/model.py:94-106).
 * ---------------------------------------------------------------------------------- */
/* Per-channel batch statistics of x [rows][C] (row stride ld) -> mean, invstd, and the folded
 * affine scale = gamma*invstd, shift = beta - mean*scale that the conv loader applies.
 * training != 0: batch stats; running_mean/var/num_batches_tracked (may be NULL) are updated
 * with `momentum` like nn.BatchNorm2d.  training == 0: running stats are used instead.
 * training = n > 1: the running statistics (and the batch counter) move exactly as after n forward passes over
 * this same batch - the discriminators see the generated images twice per iteration with unchanged weights
 * (scripts/train.py:544-548 and :566-568 / :581-583), the second pass is not computed again.
 * unbiased_rows (0 = rows): sample count used for the unbiased running_var factor - mask_net
 * normalises a x2-upsampled tensor (model.py:98-99) whose statistics equal the source's.
 * partial: scratch float[2 * C * 1024].
 * count / count_unit (count may be NULL = every row is real): a padded row batch - only the first
 * count[0] * count_unit rows are real, the statistics run over those (sg2im_amd/bucketing.py: object /
 * triple axes padded to a bucket size so one captured hipGraph serves every batch of the bucket; the
 * true sizes live in device memory, which a graph replay re-reads). */
int sg2im_bn_stats(const float* x, long long rows, int channels, long long ld, const float* gamma,
                   const float* beta, float eps, float momentum, int training, float* running_mean,
                   float* running_var, long long* num_batches_tracked, long long unbiased_rows,
                   float* mean, float* invstd, float* scale, float* shift, float* partial,
                   const int* count, int count_unit, hipStream_t stream);
/* Backward through z = leaky_slope(scale*y+shift) and the batch statistics:
 *   dz is read from `g` [rows][ld_g] (channel offset already applied by the caller), or, when
 *   pool2 != 0, as the 2x2 sum of g laid out [batch][2h][2w][ld_g] (nearest-upsample backward).
 *   outputs: dy [rows][C] dense, dgamma[C], dbeta[C] (+= if accumulate).
 *   training == 0 -> statistics are constants (eval-mode BN).
 *   partial: scratch float[2 * C * 1024 + 3 * C].
 *   count / count_unit: as for sg2im_bn_stats; padding rows get dy = 0. */
int sg2im_bn_act_backward(const float* g, long long ld_g, int pool2, int batch, int h, int w,
                          const float* y, long long ld_y, int channels, const float* gamma,
                          const float* mean, const float* invstd, const float* scale,
                          const float* shift, float slope, int training, float* dy,
                          float* dgamma, float* dbeta, int accumulate, float* partial,
                          const int* count, int count_unit, hipStream_t stream);
/* out[r][c] = leaky_slope(scale[c] * x[r][c] + shift[c]): BatchNorm-apply + activation that must be
 * materialised - BatchNorm1d + ReLU inside build_mlp (sg2im/layers.py:216-232 with batch_norm='batch')
 * when the result feeds a gather / pooling instead of another GEMM's loader. */
int sg2im_affine_act_forward(const float* x, long long ld_x, long long rows, int channels, const float* scale,
                             const float* shift, float slope, float* out, long long ld_out,
                             hipStream_t stream);
/* dst[i] = bfloat16(src[i]) (round to nearest even), i < n; both pointers 16-byte aligned.  Makes the weight mirror of
 * sg2im_conv_desc.weight_bf16 from the fp32 master weights (sg2im_amd.optim.FlatParams.refresh_mirror: one launch over
 * the generator's parameter arena per iteration; the reference keeps fp32 weights only, scripts/train.py:418-423). */
int sg2im_cast_f32_to_bf16(const float* src, void* dst, long long n, hipStream_t stream);
/* InstanceNorm2d(channels) as get_normalization_2d(.., 'instance') builds it (sg2im/layers.py:27-28:
 * affine=False, no running statistics, eps 1e-5) on a dense NHWC tensor x[batch][hw][channels]:
 *   stats:    scale[n][c] = 1/sqrt(var_hw(x) + eps) (biased variance), shift[n][c] = -mean_hw(x) * scale
 *   forward:  out = leaky_slope(scale[n][c] * x + shift[n][c])        (norm + the activation after it)
 *   backward: dyn = gradient w.r.t. the normalised value (i.e. after the activation's backward);
 *             dx = scale * (dyn - mean_hw(dyn) - yn * mean_hw(dyn * yn)); dx may alias dyn.
 * Replaces autograd through nn.InstanceNorm2d at sg2im/crn.py:42-47 and sg2im/layers.py:166-168. */
int sg2im_instnorm_stats(const float* x, int batch, int hw, int channels, float eps, float* scale, float* shift,
                         hipStream_t stream);
int sg2im_instnorm_act_forward(const float* x, int batch, int hw, int channels, const float* scale,
                               const float* shift, float slope, float* out, hipStream_t stream);
int sg2im_instnorm_backward(const float* dyn, const float* x, int batch, int hw, int channels, const float* scale,
                            const float* shift, float* dx, hipStream_t stream);
/* The spatial / elementwise tokens of build_cnn architecture strings (sg2im/layers.py:184-196) and
 * ResidualBlock (sg2im/layers.py:88-117), dense NHWC tensors:
 *   resample_nearest_up: out[b][y][x][c] = alpha * x[b][y/f][x/f][c], out is (out_h, out_w) (0 outside the
 *                        f-scaled input) - nn.Upsample(scale_factor=f, 'nearest'); with alpha = 1/f^2 the
 *                        backward of AvgPool2d(f)
 *   pool_sum_forward:    out = alpha * (sum of each f x f window), floor(h/f) x floor(w/f) outputs -
 *                        nn.AvgPool2d(f, f) with alpha = 1/f^2; with alpha = 1 the backward of the upsample
 *   maxpool_forward / _backward: nn.MaxPool2d(f, f); the first maximum of the row-major window scan takes
 *                        the gradient (ATen's rule), dx has the input's shape
 *   leaky_forward:       out = x > 0 ? x : slope * x  (a LeakyReLU with no norm / GEMM to fuse it into)
 *   add_forward:         out = a + b  (the residual sum, layers.py:117) */
int sg2im_resample_nearest_up(const float* x, int batch, int h, int w, int channels, int factor, int out_h,
                              int out_w, float alpha, float* out, hipStream_t stream);
int sg2im_pool_sum_forward(const float* x, int batch, int h, int w, int channels, int factor, float alpha,
                           float* out, hipStream_t stream);
int sg2im_maxpool_forward(const float* x, int batch, int h, int w, int channels, int factor, float* out,
                          hipStream_t stream);
int sg2im_maxpool_backward(const float* x, const float* dy, int batch, int h, int w, int channels, int factor,
                           float* dx, hipStream_t stream);
int sg2im_leaky_forward(const float* x, long long n, float slope, float* out, hipStream_t stream);
int sg2im_add_forward(const float* a, const float* b, long long n, float* out, hipStream_t stream);
/* dx = g * leaky'(y): backward of a fused output activation (y is the activated output for
 * slope >= 0: sign(y) == sign(pre-activation)); pool2 as above. */
int sg2im_act_backward(const float* g, long long ld_g, int pool2, int batch, int h, int w,
                       const float* y, long long ld_y, int channels, float slope, float* dx,
                       hipStream_t stream);
/* NHWC average pooling by `factor` (crn.py:62) and its backward summed over the pyramid:
 * dlayout[n][y][x][c] (+)= sum_l dlevel_l[n][y/f_l][x/f_l][c] / f_l^2 for c < channels. */
int sg2im_avgpool_forward(const float* x, int batch, int h, int w, int channels, int factor,
                          float* out, hipStream_t stream);
int sg2im_pyramid_backward(const float* const* dlevels, const int* factors, const long long* lds,
                           int n_levels, int batch, int h, int w, int channels, float* dlayout,
                           long long ld_out, hipStream_t stream);
/* layout conversions at the API boundary (the reference is NCHW throughout) */
int sg2im_nchw_to_nhwc(const float* src, int batch, int channels, int h, int w, float* dst,
                       long long ld_dst, int c_offset, hipStream_t stream);
int sg2im_nhwc_to_nchw(const float* src, long long ld_src, int c_offset, int batch, int channels,
                       int h, int w, float* dst, hipStream_t stream);
/* GlobalAvgPool (sg2im/layers.py:83-86) over NHWC [batch][hw][C] and its backward */
int sg2im_gap_forward(const float* x, int batch, int hw, int channels, float* out, hipStream_t stream);
int sg2im_gap_backward(const float* dout, int batch, int hw, int channels, float* dx, hipStream_t stream);
/* sigmoid of mask scores (model.py:147) and its backward */
int sg2im_sigmoid_forward(const float* x, long long n, float* y, hipStream_t stream);
int sg2im_sigmoid_backward(const float* y, const float* dy, long long n, float* dx, hipStream_t stream);

/* ------------------------------------------------------------------------------------
 * Losses (sg2im/losses.py:39-103, scripts/train.py:387-412) - each writes the scalar loss
 * (already multiplied by `weight`) to loss[0] and d(loss)/d(input) to grad (may be NULL).
 * count / count_unit (count may be NULL): padded row batches (see sg2im_bn_stats) - the mean runs
 * over the first count[0] * count_unit elements (rows for the cross entropy), the rest gets grad 0.
 * ---------------------------------------------------------------------------------- */
int sg2im_l1_loss(const float* pred, const float* target, long long n, float weight, float* loss,
                  float* grad, float* partial, hipStream_t stream);
int sg2im_mse_loss(const float* pred, const float* target, long long n, float weight, float* loss,
                   float* grad, float* partial, const int* count, int count_unit, hipStream_t stream);
/* mean( max(x,0) - x*t + log(1+exp(-|x|)) ) with a constant target t (losses.py:39-57) */
int sg2im_bce_logits_loss(const float* x, long long n, float target, float weight, float* loss,
                          float* grad, float* partial, const int* count, int count_unit, hipStream_t stream);
/* GAN score terms against a constant target (sg2im/losses.py:72-145), mean over n scores:
 *   kind 0 'gan'   BCE-with-logits (same as sg2im_bce_logits_loss)
 *   kind 1 'wgan'  target * x          (target = -1: generator / real term, +1: fake term)
 *   kind 2 'lsgan' (sigmoid(x) - target)^2 */
int sg2im_gan_score_loss(const float* x, long long n, int kind, float target, float weight, float* loss,
                         float* grad, float* partial, const int* count, int count_unit, hipStream_t stream);
/* F.binary_cross_entropy(prob, target) on probabilities (mask loss, scripts/train.py:407-410):
 * logs clamped at -100, gradient (p - t) / max(p (1 - p), 1e-12) */
int sg2im_bce_prob_loss(const float* prob, const float* target, long long n, float weight, float* loss,
                        float* grad, float* partial, const int* count, int count_unit, hipStream_t stream);
/* mean_i( logsumexp(scores[i]) - scores[i][labels[i]] )  (F.cross_entropy, discriminators.py:74);
 * partial: scratch float[max(256, rows)] (256 floats for the element-wise losses above) */
int sg2im_cross_entropy_loss(const float* scores, int rows, int classes, const long long* labels,
                             float weight, float* loss, float* grad, float* partial,
                             const int* count, int count_unit, hipStream_t stream);
/* out[0] = terms[0][0] + terms[1][0] + ... (1 <= n <= 8, left to right): the total of the weighted loss
 * terms, scripts/train.py:387-412 `total_loss += ...`, :538-550; `terms` is a HOST array of device pointers */
int sg2im_sum_scalars(const float* const* terms, int n, float* out, hipStream_t stream);
/* y = a * x[0..n) with a read from device memory (chains an upstream scalar gradient) */
int sg2im_scale_by_scalar(const float* x, const float* a_dev, long long n, float* y, hipStream_t stream);

/* Fused Adam over a flat parameter arena (torch.optim.Adam defaults, scripts/train.py:426-443):
 * m = b1*m + (1-b1)*g; v = b2*v + (1-b2)*g*g; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps);
 * grad_scale multiplies g first (1/world_size after a sum all-reduce). */
int sg2im_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n,
                    float lr, float beta1, float beta2, float eps, int step, float grad_scale,
                    hipStream_t stream);

/* Same update with the step counter on the device: state = float[4] {step, step_size,
 * 1/sqrt(bias_correction2), applied}, zero-initialised by the caller.  If guard != NULL and
 * guard[0] is not finite the whole update (and the step counter) is skipped - the
 * device-side form of "if not math.isfinite(total_loss): continue" (scripts/train.py:553-555). */
int sg2im_adam_step_guarded(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                            long long n, float lr, float beta1, float beta2, float eps,
                            float grad_scale, float* state, const float* guard, hipStream_t stream);
/* The same in two calls: prepare (once per optimiser step: advances state[0], sets the bias corrections and the
 * "applied" flag - or marks the step skipped when guard[0] is not finite) and apply (the update of ANY slice
 * [param, param + n) of the arena, in any order, on any stream ordered after the prepare) - so that a part of the arena
 * whose gradients are complete early can be updated while the rest of the backward pass is still running. */
int sg2im_adam_prepare_guarded(float lr, float beta1, float beta2, float* state, const float* guard, hipStream_t stream);
int sg2im_adam_apply_guarded(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n,
                             float beta1, float beta2, float eps, float grad_scale, const float* state,
                             hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SG2IM_HIP_H */
