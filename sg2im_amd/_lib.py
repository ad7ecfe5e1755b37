"""ctypes binding of libsg2im_hip.so (C ABI declared in include/sg2im_hip.h).

The product path has no CPU fallback: if the library is missing or a call fails the
error is raised here, loudly.
"""
import ctypes
import os
from ctypes import c_float, c_int, c_longlong, c_size_t, c_void_p, POINTER, Structure

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('SG2IM_LIB') or os.path.join(_HERE, 'lib', 'libsg2im_hip.so')

SG2IM_OK, SG2IM_ERR_ARG, SG2IM_ERR_HIP = 0, 1, 2


class Sg2imHipError(RuntimeError):
  pass


class Src(Structure):
  _fields_ = [('data', c_void_p), ('gather', c_void_p), ('scale', c_void_p), ('shift', c_void_p),
              ('slope', c_float), ('channels', c_int), ('ld', c_int), ('upsample_log2', c_int), ('dtype', c_int)]


class ConvDesc(Structure):
  _fields_ = [('src', Src * 4), ('nsrc', c_int), ('batch', c_int), ('in_h', c_int), ('in_w', c_int),
              ('out_h', c_int), ('out_w', c_int), ('kh', c_int), ('kw', c_int), ('stride', c_int),
              ('pad', c_int), ('compute_dtype', c_int), ('launch_hints', c_int), ('weight_channels', c_int),
              ('out_dtype', c_int), ('dy_dtype', c_int), ('weight_bf16', c_void_p)]


class BnFwd(Structure):
  """sg2im_bn_fwd (include/sg2im_hip.h)"""
  _fields_ = [('gamma', c_void_p), ('beta', c_void_p), ('eps', c_float), ('momentum', c_float), ('training', c_int),
              ('running_mean', c_void_p), ('running_var', c_void_p), ('num_batches_tracked', c_void_p),
              ('unbiased_rows', c_longlong), ('mean', c_void_p), ('invstd', c_void_p), ('scale', c_void_p),
              ('shift', c_void_p), ('partial', c_void_p), ('partial_floats', c_size_t), ('count', c_void_p),
              ('count_unit', c_int)]


class BnBwd(Structure):
  """sg2im_bn_bwd (include/sg2im_hip.h)"""
  _fields_ = [('y', c_void_p), ('ld_y', c_longlong), ('pool2', c_int), ('gamma', c_void_p), ('mean', c_void_p),
              ('invstd', c_void_p), ('scale', c_void_p), ('shift', c_void_p), ('slope', c_float), ('training', c_int),
              ('dgamma', c_void_p), ('dbeta', c_void_p), ('accumulate', c_int), ('coef', c_void_p), ('partial', c_void_p),
              ('partial_floats', c_size_t), ('count', c_void_p), ('count_unit', c_int), ('y_dtype', c_int)]


class GconvLayer(Structure):
  """sg2im_gconv_layer (include/sg2im_hip.h)"""
  _fields_ = [('obj_vecs', c_void_p), ('ld_obj', c_longlong), ('pred_vecs', c_void_p), ('ld_pred', c_longlong),
              ('s_idx', c_void_p), ('o_idx', c_void_p), ('row_ptr', c_void_p), ('entries', c_void_p),
              ('n_objs', c_int), ('n_triples', c_int), ('din', c_int), ('hidden', c_int), ('dout', c_int),
              ('average', c_int)] + [(k, c_void_p) for k in ('w1a', 'b1a', 'w1b', 'b1b', 'w2a', 'b2a', 'w2b', 'b2b')]


SG2IM_GCONV_MAX_LAYERS = 8


class GconvStackLayer(Structure):
  """sg2im_gconv_stack_layer (include/sg2im_hip.h)"""
  _fields_ = [(k, c_void_p) for k in ('w1a', 'b1a', 'w1b', 'b1b', 'w2a', 'b2a', 'w2b', 'b2b', 'h1', 'new_t', 'pooled', 'h2',
                                      'new_obj')] + [('din', c_int), ('hidden', c_int), ('dout', c_int), ('reserved', c_int)]


class GconvStack(Structure):
  """sg2im_gconv_stack (include/sg2im_hip.h)"""
  _fields_ = [('obj_vecs', c_void_p), ('ld_obj', c_longlong), ('pred_vecs', c_void_p), ('ld_pred', c_longlong),
              ('s_idx', c_void_p), ('o_idx', c_void_p), ('row_ptr', c_void_p), ('entries', c_void_p),
              ('n_objs', c_int), ('n_triples', c_int), ('n_layers', c_int), ('average', c_int),
              ('layer', GconvStackLayer * SG2IM_GCONV_MAX_LAYERS)]


class GconvGrads(Structure):
  """sg2im_gconv_grads (include/sg2im_hip.h)"""
  _fields_ = [(k, c_void_p) for k in ('dw1a', 'db1a', 'dw1b', 'db1b', 'dw2a', 'db2a', 'dw2b', 'db2b')] + [('accumulate', c_int)]


class GconvStackGrads(Structure):
  """sg2im_gconv_stack_grads (include/sg2im_hip.h)"""
  _fields_ = [('g_obj', c_void_p), ('g_pred', c_void_p), ('ld_gpred', c_longlong), ('d_triple', c_void_p), ('d_obj', c_void_p),
              ('scratch', c_void_p), ('scratch_bytes', c_size_t), ('layer', GconvGrads * SG2IM_GCONV_MAX_LAYERS),
              ('low_footprint', c_int), ('reserved', c_int)]


_P, _I, _L, _F, _Z = c_void_p, c_int, c_longlong, c_float, c_size_t
_D = POINTER(ConvDesc)

# name -> argtypes (return type is int for all but the two noted)
_SIGNATURES = {
  'sg2im_abi_version': [],
  'sg2im_init': [],
  'sg2im_launch_count': [_I],
  'sg2im_conv2d_forward': [_D, _P, _I, _P, _F, _P, _L, _I, _P, _Z, _P],
  'sg2im_conv2d_backward_data': [_D, _P, _I, _P, _I, _I, _I, _P, _L, _I, _P, _Z, _P],
  'sg2im_conv2d_backward_data_act': [_D, _P, _I, _P, _I, _I, _I, _P, _L, _P, _L, _F, _P, _Z, _P],
  'sg2im_conv2d_forward_bn': [_D, _P, _I, _P, _F, _P, _L, _P, _Z, POINTER(BnFwd), _P],
  'sg2im_conv2d_backward_data_bn': [_D, _P, _I, _P, _I, _I, _I, _P, _L, _P, _Z, POINTER(BnBwd), _P],
  'sg2im_bn_backward_apply': [_P, _L, _I, _I, _I, _I, _P, _L, _I, _P, _P, _F, _P, _P, _P, _I, _P],
  'sg2im_conv2d_backward_weight': [_D, _P, _I, _I, _P, _P, _I, _P, _Z, _P],
  'sg2im_conv2d_backward_weight_group': [_I, POINTER(_D), POINTER(_P), POINTER(_I), POINTER(_I), POINTER(_P), POINTER(_P),
                                         _I, _P, _Z, _P],
  'sg2im_column_sum': [_P, _L, _I, _L, _P, _I, _P, _P],
  'sg2im_csr_build': [_P, _I, _P, _I, _I, _P, _P, _P, _P, _P],
  'sg2im_csr_build_triples': [_P, _I, _I, _P, _P, _P, _P, _P, _P],
  'sg2im_segment_sum': [_P, _L, _I, _P, _L, _P, _P, _I, _I, _I, _I, _P, _L, _P],
  'sg2im_gather_rows': [_P, _L, _P, _I, _I, _P, _P, _L, _P],
  'sg2im_gconv_pool_backward': [_P, _L, _P, _P, _I, _P, _P, _L, _P, _L, _I, _I, _F, _P, _L, _P],
  'sg2im_copy_2d': [_P, _L, _P, _L, _L, _I, _P],
  'sg2im_timestamp': [_P, _P],
  'sg2im_debug_mark_gemm_end': [_P, POINTER(c_int)],
  'sg2im_stage_batch': [_I, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_size_t), _P],
  'sg2im_gconv_layer_forward': [POINTER(GconvLayer), _P, _P, _P, _P, _P, _P, _Z, _P],
  'sg2im_gconv_layer_backward_scratch': [_I, _I, _I, _I, _I],
  'sg2im_gconv_layer_backward': [POINTER(GconvLayer), _P, _P, _P, _P, _P, _P, _P, _L, _P, _P, POINTER(GconvGrads), _P, _Z,
                                 _P, _Z, _P],
  'sg2im_gconv_stack_sync_bytes': [],
  'sg2im_gconv_stack_supported': [_I, _I, _I],
  'sg2im_gconv_stack_forward': [POINTER(GconvStack), _P, _Z, _P],
  'sg2im_gconv_stack_backward_scratch': [POINTER(GconvStack)],
  'sg2im_gconv_stack_backward': [POINTER(GconvStack), POINTER(GconvStackGrads), _P, _Z, _P],
  'sg2im_gconv_stack_status': [_P],
  'sg2im_gconv_stack_stamps': [_P, _P, _I],
  'sg2im_layout_forward': [_P, _L, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _P, _L, _P],
  'sg2im_layout_pyramid_forward': [_P, _L, _P, _P, _P, _I, _P, _P, _I, _I, _P, _I, _I, _I, _I, _I, POINTER(c_void_p), _L, _P],
  'sg2im_layout_backward_workspace': [_I, _I, _I, _I],
  'sg2im_layout_backward': [_P, _L, _P, _L, _P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _L, _P,
                            _P, _P, _P],
  'sg2im_layout_backward_vecs_levels': [POINTER(c_void_p), POINTER(c_int), POINTER(c_longlong), _I, _P, _P, _P, _I, _P, _P, _I,
                                        _I, _I, _I, _I, _I, _P, _L, _P, _P],
  'sg2im_layout_backward_maps_levels': [POINTER(c_void_p), POINTER(c_int), POINTER(c_longlong), _I, _P, _L, _P, _P, _P, _I, _P,
                                        _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P],
  'sg2im_crop_forward': [_P, _L, _I, _I, _I, _I, _P, _P, _I, _I, _I, _P, _P],
  'sg2im_crop_backward_workspace': [_I, _I, _I, _I],
  'sg2im_crop_backward': [_P, _I, _I, _I, _I, _P, _P, _I, _I, _I, _P, _L, _P, _P],
  'sg2im_bn_stats': [_P, _L, _I, _L, _P, _P, _F, _F, _I, _P, _P, _P, _L, _P, _P, _P, _P, _P, _P, _I, _P],
  'sg2im_bn_act_backward': [_P, _L, _I, _I, _I, _I, _P, _L, _I, _P, _P, _P, _P, _P, _F, _I, _P, _P, _P, _I,
                            _P, _P, _I, _P],
  'sg2im_affine_act_forward': [_P, _L, _L, _I, _P, _P, _F, _P, _L, _P],
  'sg2im_resample_nearest_up': [_P, _I, _I, _I, _I, _I, _I, _I, _F, _P, _P],
  'sg2im_pool_sum_forward': [_P, _I, _I, _I, _I, _I, _F, _P, _P],
  'sg2im_maxpool_forward': [_P, _I, _I, _I, _I, _I, _P, _P],
  'sg2im_maxpool_backward': [_P, _P, _I, _I, _I, _I, _I, _P, _P],
  'sg2im_leaky_forward': [_P, _L, _F, _P, _P],
  'sg2im_add_forward': [_P, _P, _L, _P, _P],
  'sg2im_instnorm_stats': [_P, _I, _I, _I, _F, _P, _P, _P],
  'sg2im_instnorm_act_forward': [_P, _I, _I, _I, _P, _P, _F, _P, _P],
  'sg2im_instnorm_backward': [_P, _P, _I, _I, _I, _P, _P, _P, _P],
  'sg2im_act_backward': [_P, _L, _I, _I, _I, _I, _P, _L, _I, _F, _P, _P],
  'sg2im_avgpool_forward': [_P, _I, _I, _I, _I, _I, _P, _P],
  'sg2im_pyramid_backward': [POINTER(c_void_p), POINTER(c_int), POINTER(c_longlong), _I, _I, _I, _I, _I, _P,
                             _L, _P],
  'sg2im_nchw_to_nhwc': [_P, _I, _I, _I, _I, _P, _L, _I, _P],
  'sg2im_nhwc_to_nchw': [_P, _L, _I, _I, _I, _I, _I, _P, _P],
  'sg2im_gap_forward': [_P, _I, _I, _I, _P, _P],
  'sg2im_gap_backward': [_P, _I, _I, _I, _P, _P],
  'sg2im_sigmoid_forward': [_P, _L, _P, _P],
  'sg2im_sigmoid_backward': [_P, _P, _L, _P, _P],
  'sg2im_l1_loss': [_P, _P, _L, _F, _P, _P, _P, _P],
  'sg2im_mse_loss': [_P, _P, _L, _F, _P, _P, _P, _P, _I, _P],
  'sg2im_bce_logits_loss': [_P, _L, _F, _F, _P, _P, _P, _P, _I, _P],
  'sg2im_gan_score_loss': [_P, _L, _I, _F, _F, _P, _P, _P, _P, _I, _P],
  'sg2im_bce_prob_loss': [_P, _P, _L, _F, _P, _P, _P, _P, _I, _P],
  'sg2im_cross_entropy_loss': [_P, _I, _I, _P, _F, _P, _P, _P, _P, _I, _P],
  'sg2im_scale_by_scalar': [_P, _P, _L, _P, _P],
  'sg2im_sum_scalars': [_P, _I, _P, _P],
  'sg2im_adam_step': [_P, _P, _P, _P, _L, _F, _F, _F, _F, _I, _F, _P],
  'sg2im_adam_step_guarded': [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _P, _P, _P],
  'sg2im_adam_prepare_guarded': [_F, _F, _F, _P, _P, _P],
  'sg2im_adam_apply_guarded': [_P, _P, _P, _P, _L, _F, _F, _F, _F, _P, _P],
  'sg2im_cast_f32_to_bf16': [_P, _P, _L, _P],
  'sg2im_bn_backward_apply_ex': [_P, _L, _I, _I, _I, _I, _P, _L, _I, _P, _P, _F, _P, _P, _I, _I, _I, _P],
  'sg2im_conv_halo_unsplit': [_I, _I, _I, _I, _I],
  'sg2im_two_heads_supported': [_I, _I, _I],
  'sg2im_two_heads_forward': [_P, _L, _I, _I, _P, _P, _I, _P, _P, _I, _P, _L, _P, _L, _P],
  'sg2im_two_heads_backward_data': [_P, _L, _P, _L, _I, _I, _P, _I, _P, _I, _P, _L, _P],
}
_RESTYPE = {'sg2im_layout_backward_workspace': c_size_t, 'sg2im_crop_backward_workspace': c_size_t,
            'sg2im_launch_count': ctypes.c_ulonglong, 'sg2im_gconv_layer_backward_scratch': c_size_t,
            'sg2im_gconv_stack_sync_bytes': c_size_t, 'sg2im_gconv_stack_backward_scratch': c_size_t}
EXPORTS = tuple(sorted(_SIGNATURES))

_lib = None


def load():
  """Load the shared library (once).  Raises Sg2imHipError when it is missing - there is
  deliberately no fallback implementation."""
  global _lib
  if _lib is not None:
    return _lib
  if not os.path.exists(LIB_PATH):
    raise Sg2imHipError(
      'libsg2im_hip.so not found at %s - build it with `python -m sg2im_amd.build` '
      '(or __graft_entry__.build()); the HIP path has no fallback' % LIB_PATH)
  # torch must own the process' HIP runtime: it bundles its own libamdhip64 and the device
  # pointers / streams we are handed come from that instance.  Importing torch first makes
  # our library's libamdhip64 dependency resolve to the already-loaded copy.
  import torch  # noqa: F401
  torch.cuda.is_available()
  lib = ctypes.CDLL(LIB_PATH)
  for name, argtypes in _SIGNATURES.items():
    fn = getattr(lib, name)          # AttributeError if the symbol is not exported
    fn.argtypes = argtypes
    fn.restype = _RESTYPE.get(name, c_int)
  _lib = lib
  return lib


_TRACE = os.environ.get('SG2IM_TRACE', '') == '1'
EAGER_EPOCH = 0      # number of EAGER launches made through this binding (captured ones do not count)
CAPTURING = False    # set by the Trainer around a stream capture: launches are recorded, not executed
_inited = False


def init():
  """sg2im_init(): kernel attributes + code-object load, once, before the first launch (so that a
  first launch may already sit inside a stream capture)."""
  global _inited
  if not _inited:
    if load().sg2im_init() != SG2IM_OK:
      raise Sg2imHipError('sg2im_init failed')
    _inited = True


def call(name, *args):
  """Invoke an int-returning entry point and raise on a non-zero status.  SG2IM_TRACE=1
  prints every call and synchronises after it (debugging aid: pins a device fault to an op)."""
  global EAGER_EPOCH
  if not _inited:
    init()
  if not CAPTURING:
    EAGER_EPOCH += 1          # see sg2im_amd/trainer.py::_graph_step (graph invalidation)
  rc = getattr(load(), name)(*args)
  if _TRACE:
    import sys
    import torch
    sys.stderr.write('[sg2im] %s\n' % name)
    sys.stderr.flush()
    if not torch.cuda.is_current_stream_capturing():
      torch.cuda.synchronize()
  if rc != SG2IM_OK:
    kind = {SG2IM_ERR_ARG: 'invalid argument', SG2IM_ERR_HIP: 'HIP runtime error'}.get(rc, 'error %d' % rc)
    raise Sg2imHipError('%s failed: %s' % (name, kind))
  return rc
