// Internal interface between conv.hip and norm.hip: the finishes of BatchNorm reductions whose per-tile
// partial sums were produced by a GEMM epilogue / split-K finish (sg2im_conv2d_forward_bn,
// sg2im_conv2d_backward_data_bn).
#pragma once
#include "sg2im_hip.h"

namespace sg2im {

// partial: [3][channels][nblk] = (pivot, sum (x - pivot), sum (x - pivot)^2) of the rows [t * per, (t + 1) * per)
// of an output with `rows` rows -> mean / invstd / folded scale / shift (+ running statistics)
int bn_stats_finish_tiles(const float* partial, int nblk, long long per, long long rows, int channels,
                          const sg2im_bn_fwd* a, hipStream_t stream);
// partial: [2][channels][nblk] = (sum du, sum du * xhat) -> dgamma / dbeta / the coefficients of
// dy = a du + k1 y + k0;  rows: rows of the normalised tensor
int bn_bwd_finish_tiles(const float* partial, int nblk, long long rows, int channels, const sg2im_bn_bwd* a,
                        hipStream_t stream);
// the same from a finished gradient tensor g (the first two launches of sg2im_bn_act_backward)
int bn_bwd_standalone(const float* g, long long ld_g, int pool2, int batch, int h, int w, int channels,
                      const sg2im_bn_bwd* a, hipStream_t stream);

}  // namespace sg2im
