// One GraphTripleConv layer (reference sg2im/graph.py:56-120) behind ONE C entry point per direction:
// sg2im_gconv_layer_forward / sg2im_gconv_layer_backward run the layer's whole launch sequence - the gather +
// concat of (subject, predicate, object) vectors folded into net1's first GEMM, net1, the deterministic CSR
// pooling, net2 - resp. its mirror image with the four weight gradients as one grouped launch, on the caller's
// stream, with caller-provided activation / scratch buffers (no allocation, no state).  A host that is not
// Python binds two functions per layer instead of re-creating the sequence.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>
#include "sg2im_hip.h"

namespace {

// A/B knob (SG2IM_FUSE_ACT_BWD=0): the ReLU backward behind a data gradient as a launch of its own (rounds 1-4)
const bool g_fuse_act = !(getenv("SG2IM_FUSE_ACT_BWD") && atoi(getenv("SG2IM_FUSE_ACT_BWD")) == 0);

sg2im_conv_desc rows_desc(int rows) {
  sg2im_conv_desc d;
  std::memset(&d, 0, sizeof(d));
  d.batch = rows; d.in_h = d.in_w = d.out_h = d.out_w = 1;
  d.kh = d.kw = 1; d.stride = 1; d.pad = 0;
  return d;
}

void set_src(sg2im_conv_desc& d, int i, const float* p, int channels, long long ld, const long long* gather) {
  d.src[i].data = p; d.src[i].gather = gather; d.src[i].scale = nullptr; d.src[i].shift = nullptr;
  d.src[i].slope = 1.f; d.src[i].channels = channels; d.src[i].ld = (int)ld; d.src[i].upsample_log2 = 0;
  if (i + 1 > d.nsrc) d.nsrc = i + 1;
}

// net1's first Linear reads [obj[s], pred, obj[o]] per triple: three sources, two of them row-gathered
sg2im_conv_desc triple_desc(const sg2im_gconv_layer* L) {
  sg2im_conv_desc d = rows_desc(L->n_triples);
  set_src(d, 0, L->obj_vecs, L->din, L->ld_obj, L->s_idx);
  set_src(d, 1, L->pred_vecs, L->din, L->ld_pred, nullptr);
  set_src(d, 2, L->obj_vecs, L->din, L->ld_obj, L->o_idx);
  return d;
}

sg2im_conv_desc dense_desc(const float* x, int rows, int channels) {
  sg2im_conv_desc d = rows_desc(rows);
  set_src(d, 0, x, channels, channels, nullptr);
  return d;
}

bool layer_ok(const sg2im_gconv_layer* L) {
  return L && L->obj_vecs && L->row_ptr && L->n_objs >= 1 && L->n_triples >= 0 && L->din >= 1 && L->hidden >= 1 &&
         L->dout >= 1 && L->ld_obj >= L->din && (L->n_triples == 0 || (L->pred_vecs && L->s_idx && L->o_idx && L->entries &&
         L->ld_pred >= L->din)) && L->w1a && L->w1b && L->w2a && L->w2b;
}

}  // namespace

#define SG2IM_TRY(call) do { const int rc_ = (call); if (rc_ != SG2IM_OK) return rc_; } while (0)

extern "C" {

size_t sg2im_gconv_layer_backward_scratch(int n_objs, int n_triples, int din, int hidden, int dout) {
  (void)din;
  const size_t O = (size_t)n_objs, T = (size_t)n_triples, H = (size_t)hidden, NT = 2 * H + (size_t)dout;
  // dp4 [O][dout], dh2/dp3 [O][H], dpooled [O][H], d_new_t/dp2 [T][NT], dh1/dp1 [T][H]  (+ alignment slack)
  return (O * dout + 2 * O * H + T * NT + T * H + 64) * sizeof(float);
}

int sg2im_gconv_layer_forward(const sg2im_gconv_layer* L, float* h1, float* new_t, float* pooled, float* h2,
                              float* new_obj, float* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (!layer_ok(L) || !pooled || !h2 || !new_obj || (L->n_triples > 0 && (!h1 || !new_t))) return SG2IM_ERR_ARG;
  const int T = L->n_triples, O = L->n_objs, H = L->hidden, Dout = L->dout, NT = 2 * H + Dout;
  if (T > 0) {
    const sg2im_conv_desc d1 = triple_desc(L);
    SG2IM_TRY(sg2im_conv2d_forward(&d1, L->w1a, H, L->b1a, 0.f, h1, H, 0, workspace, workspace_bytes, stream));
    const sg2im_conv_desc d2 = dense_desc(h1, T, H);
    SG2IM_TRY(sg2im_conv2d_forward(&d2, L->w1b, NT, L->b1b, 0.f, new_t, NT, 0, workspace, workspace_bytes, stream));
  }
  // pooled[j] = sum / avg over row j's CSR entries: s-hits read new_t[:, :H], o-hits new_t[:, H + Dout:]
  if (T > 0)
    SG2IM_TRY(sg2im_segment_sum(new_t, NT, T, new_t + H + Dout, NT, L->row_ptr, L->entries, O, H, L->average, 0, pooled, H, stream));
  else if (hipMemsetAsync(pooled, 0, sizeof(float) * (size_t)O * H, stream) != hipSuccess) return SG2IM_ERR_HIP;
  const sg2im_conv_desc d3 = dense_desc(pooled, O, H);
  SG2IM_TRY(sg2im_conv2d_forward(&d3, L->w2a, H, L->b2a, 0.f, h2, H, 0, workspace, workspace_bytes, stream));
  const sg2im_conv_desc d4 = dense_desc(h2, O, H);
  SG2IM_TRY(sg2im_conv2d_forward(&d4, L->w2b, Dout, L->b2b, 0.f, new_obj, Dout, 0, workspace, workspace_bytes, stream));
  return SG2IM_OK;
}

int sg2im_gconv_layer_backward(const sg2im_gconv_layer* L, const float* h1, const float* new_t, const float* pooled,
                               const float* h2, const float* new_obj, const float* g_obj, const float* g_pred,
                               long long ld_gpred, float* d_triple, float* d_obj, const sg2im_gconv_grads* G,
                               float* scratch, size_t scratch_bytes, float* workspace, size_t workspace_bytes,
                               hipStream_t stream) {
  if (!layer_ok(L) || !pooled || !h2 || !new_obj || !G || !scratch) return SG2IM_ERR_ARG;
  const int T = L->n_triples, O = L->n_objs, Din = L->din, H = L->hidden, Dout = L->dout, NT = 2 * H + Dout;
  if (T > 0 && (!h1 || !new_t || !d_triple)) return SG2IM_ERR_ARG;
  if (scratch_bytes < sg2im_gconv_layer_backward_scratch(O, T, Din, H, Dout)) return SG2IM_ERR_ARG;
  // scratch carve-up (every piece 16-byte aligned: the sizes are padded to multiples of 4 floats)
  auto up4 = [](size_t n) { return (n + 3) / 4 * 4; };
  float* dp4 = scratch;
  float* dh2 = dp4 + up4((size_t)O * Dout);
  float* dpooled = dh2 + up4((size_t)O * H);
  float* dnt = dpooled + up4((size_t)O * H);
  float* dh1 = dnt + up4((size_t)T * NT);
  // ---- net2 ----
  if (g_obj) SG2IM_TRY(sg2im_act_backward(g_obj, Dout, 0, O, 1, 1, new_obj, Dout, Dout, 0.f, dp4, stream));
  else if (hipMemsetAsync(dp4, 0, sizeof(float) * (size_t)O * Dout, stream) != hipSuccess) return SG2IM_ERR_HIP;
  const sg2im_conv_desc d4 = dense_desc(h2, O, H);
  // dp3 = (dp4 W2b) * relu'(h2): the mask rides in the data gradient's launches (epilogue / split-K finish)
  if (g_fuse_act) {
    SG2IM_TRY(sg2im_conv2d_backward_data_act(&d4, L->w2b, Dout, dp4, Dout, 0, H, dh2, H, h2, H, 0.f, workspace, workspace_bytes, stream));
  } else {
    SG2IM_TRY(sg2im_conv2d_backward_data(&d4, L->w2b, Dout, dp4, Dout, 0, H, dh2, H, 0, workspace, workspace_bytes, stream));
    SG2IM_TRY(sg2im_act_backward(dh2, H, 0, O, 1, 1, h2, H, H, 0.f, dh2, stream));          // dp3 (in place)
  }
  const sg2im_conv_desc d3 = dense_desc(pooled, O, H);
  SG2IM_TRY(sg2im_conv2d_backward_data(&d3, L->w2a, H, dh2, H, 0, H, dpooled, H, 0, workspace, workspace_bytes, stream));
  sg2im_conv_desc d1 = triple_desc(L), d2 = dense_desc(h1, T, H);
  if (T > 0) {
    // ---- pooling backward + ReLU of net1's output in one launch: dp2 ----
    SG2IM_TRY(sg2im_gconv_pool_backward(dpooled, H, L->s_idx, L->o_idx, T, L->average ? L->row_ptr : nullptr, g_pred, ld_gpred,
                                        new_t, NT, H, Dout, 0.f, dnt, NT, stream));
    // ---- net1 ----
    if (g_fuse_act) {                                                                       // dp1 = (dnt W1b) * relu'(h1)
      SG2IM_TRY(sg2im_conv2d_backward_data_act(&d2, L->w1b, NT, dnt, NT, 0, H, dh1, H, h1, H, 0.f, workspace, workspace_bytes, stream));
    } else {
      SG2IM_TRY(sg2im_conv2d_backward_data(&d2, L->w1b, NT, dnt, NT, 0, H, dh1, H, 0, workspace, workspace_bytes, stream));
      SG2IM_TRY(sg2im_act_backward(dh1, H, 0, T, 1, 1, h1, H, H, 0.f, dh1, stream));        // dp1 (in place)
    }
    SG2IM_TRY(sg2im_conv2d_backward_data(&d1, L->w1a, H, dh1, H, 0, 3 * Din, d_triple, 3 * Din, 0, workspace, workspace_bytes,
                                         stream));
  }
  if (d_obj) {
    // gradient of the two row gathers: rows of d_triple's subject / object column blocks back to the objects
    if (T > 0)
      SG2IM_TRY(sg2im_segment_sum(d_triple, 3 * Din, T, d_triple + 2 * Din, 3 * Din, L->row_ptr, L->entries, O, Din, 0, 0, d_obj,
                                  Din, stream));
    else if (hipMemsetAsync(d_obj, 0, sizeof(float) * (size_t)O * Din, stream) != hipSuccess) return SG2IM_ERR_HIP;
  }
  // ---- the four weight (+ bias) gradients: leaves, one grouped launch + one finish when they qualify ----
  const int acc = G->accumulate;
  if (T > 0) {
    const sg2im_conv_desc* descs[4] = {&d4, &d3, &d2, &d1};
    const float* dys[4] = {dp4, dh2, dnt, dh1};
    const int lds[4] = {Dout, H, NT, H}, couts[4] = {Dout, H, NT, H};
    float* dws[4] = {G->dw2b, G->dw2a, G->dw1b, G->dw1a};
    float* dbs[4] = {G->db2b, G->db2a, G->db1b, G->db1a};
    bool all = true;
    for (int i = 0; i < 4; ++i) all = all && dws[i] != nullptr;
    int rc = SG2IM_ERR_ARG;
    if (all) rc = sg2im_conv2d_backward_weight_group(4, descs, dys, lds, couts, dws, dbs, acc, workspace, workspace_bytes, stream);
    if (rc == SG2IM_ERR_HIP) return rc;
    if (rc != SG2IM_OK)
      for (int i = 0; i < 4; ++i)
        if (dws[i]) SG2IM_TRY(sg2im_conv2d_backward_weight(descs[i], dys[i], lds[i], couts[i], dws[i], dbs[i], acc, workspace,
                                                           workspace_bytes, stream));
  } else {
    // no triples: only net2 saw data (every object pooled to zero)
    if (G->dw2b) SG2IM_TRY(sg2im_conv2d_backward_weight(&d4, dp4, Dout, Dout, G->dw2b, G->db2b, acc, workspace, workspace_bytes, stream));
    if (G->dw2a) SG2IM_TRY(sg2im_conv2d_backward_weight(&d3, dh2, H, H, G->dw2a, G->db2a, acc, workspace, workspace_bytes, stream));
    if (!acc) {
      if (G->dw1b && hipMemsetAsync(G->dw1b, 0, sizeof(float) * (size_t)NT * H, stream) != hipSuccess) return SG2IM_ERR_HIP;
      if (G->db1b && hipMemsetAsync(G->db1b, 0, sizeof(float) * (size_t)NT, stream) != hipSuccess) return SG2IM_ERR_HIP;
      if (G->dw1a && hipMemsetAsync(G->dw1a, 0, sizeof(float) * (size_t)H * 3 * Din, stream) != hipSuccess) return SG2IM_ERR_HIP;
      if (G->db1a && hipMemsetAsync(G->db1a, 0, sizeof(float) * (size_t)H, stream) != hipSuccess) return SG2IM_ERR_HIP;
    }
  }
  return SG2IM_OK;
}

}  // extern "C"
