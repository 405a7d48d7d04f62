"""ctypes binding of libfsvit.so (C-ABI declared in include/fsvit.h).

There is deliberately NO fallback: if the shared library is missing or fails to load, importing
the engine raises — the product path never silently runs on anything but the HIP kernels.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libfsvit.so')

F32, BF16, F16, BF16X2, F16X2 = 0, 1, 2, 3, 4
ACT_NONE, ACT_GELU, ACT_LRELU = 0, 1, 2
HEAD_COS, HEAD_SQR, HEAD_DOT = 0, 1, 2
ERR_ARG, ERR_KEY, ERR_IMG_SIZE, ERR_WORKSPACE = -1, -2, -3, -4


class Tensor(C.Structure):
    _fields_ = [('name', C.c_char_p), ('data', C.POINTER(C.c_float)), ('ndim', C.c_int),
                ('shape', C.c_int64 * 4)]


class VisformerCfg(C.Structure):
    _fields_ = [('img_size', C.c_int), ('init_channels', C.c_int), ('embed_dim', C.c_int),
                ('depth', C.c_int * 3), ('num_heads', C.c_int), ('mlp_ratio', C.c_float),
                ('group', C.c_int), ('bn_eps', C.c_float)]


class VitCfg(C.Structure):
    _fields_ = [('img_size', C.c_int), ('patch_size', C.c_int), ('embed_dim', C.c_int), ('depth', C.c_int),
                ('num_heads', C.c_int), ('mlp_ratio', C.c_float), ('ln_eps', C.c_float)]


class ProfRec(C.Structure):
    _fields_ = [('layer', C.c_char * 48), ('kernel_id', C.c_int), ('launches', C.c_int), ('flops', C.c_double),
                ('ms', C.c_double)]


class Param(C.Structure):
    _fields_ = [('name', C.c_char_p), ('data', C.c_void_p), ('grad', C.c_void_p), ('numel', C.c_int64)]


# name -> (restype, argtypes); every symbol include/fsvit.h declares
_vp, _fp, _i, _f, _sz = C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_size_t
SIGNATURES = {
    'fsvit_last_error': (C.c_char_p, []),
    'fsvit_version': (C.c_char_p, []),
    'fsvit_visformer_create': (_i, [C.POINTER(VisformerCfg), C.POINTER(Tensor), _i, _i, C.POINTER(_vp)]),
    'fsvit_visformer_destroy': (None, [_vp]),
    'fsvit_visformer_out_dim': (_i, [_vp]),
    'fsvit_visformer_dtype': (_i, [_vp]),
    'fsvit_visformer_workspace_bytes': (_sz, [_vp, _i]),
    'fsvit_visformer_forward': (_i, [_vp, _fp, _i, _i, _i, _fp, _vp, _sz, _vp]),
    'fsvit_vit_create': (_i, [C.POINTER(VitCfg), C.POINTER(Tensor), _i, _i, C.POINTER(_vp)]),
    'fsvit_vit_destroy': (None, [_vp]),
    'fsvit_vit_out_dim': (_i, [_vp]),
    'fsvit_vit_workspace_bytes': (_sz, [_vp, _i]),
    'fsvit_vit_forward': (_i, [_vp, _fp, _i, _i, _i, _fp, _vp, _sz, _vp]),
    'fsvit_encoder_set_tap': (_i, [_vp, C.c_char_p, _vp, _sz]),
    'fsvit_encoder_profile_begin': (_i, [_vp]),
    'fsvit_encoder_profile_end': (_i, [_vp, C.POINTER(ProfRec), _i, C.POINTER(_i)]),
    'fsvit_kernel_name': (C.c_char_p, [_i, _i]),
    'fsvit_proto_head': (_i, [_fp, _fp, _i, _i, _i, _i, _i, _f, _i, _fp, _fp, _fp, _vp]),
    'fsvit_proto_head_devtemp': (_i, [_fp, _fp, _i, _i, _i, _i, _i, _fp, _i, _fp, _fp, _fp, _vp]),
    'fsvit_meta_baseline_forward': (_i, [_vp, _fp, _fp, _i, _i, _i, _i, _i, _i, _f, _i, _fp, _fp, _fp, _fp,
                                         _vp, _sz, _vp]),
    'fsvit_conv_gemm': (_i, [_vp, _vp, _fp, _vp, _fp, _vp] + [_i] * 15 + [_i, _vp]),
    'fsvit_conv_stem_tail': (_i, [_vp, _vp, _fp, _fp, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'fsvit_stage1_block': (_i, [_vp, _vp, _vp, _fp, _vp, _vp, _i, _vp]),
    'fsvit_stage1_block_hw': (_i, [_vp, _vp, _vp, _fp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'fsvit_visformer_last_tokens': (_i, [_vp, _vp, _sz, _i, _fp, _vp]),
    'fsvit_visformer_train_tokens': (_i, [_vp, _fp, _vp]),
    'fsvit_visformer_train_set_token_grad': (_i, [_vp, _fp]),
    'fsvit_linear_forward': (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _vp]),
    'fsvit_linear_backward': (_i, [_fp, _fp, _fp, _fp, _i, _fp, _fp, _i, _i, _i, _vp]),
    'fsvit_token_softlabel': (_i, [_fp, _fp, _i, _i, _i, _i, _i, C.c_double, _vp]),
    'fsvit_soft_target_ce': (_i, [_fp, _fp, _fp, _fp, _i, _i, _f, _vp]),
    'fsvit_row_normalize': (_i, [_fp, _fp, _fp, _i, _i, _vp]),
    'fsvit_row_normalize_backward': (_i, [_fp, _fp, _fp, _fp, _i, _i, _vp]),
    'fsvit_adamw_step': (_i, [_fp, _fp, _fp, _fp, _sz, _f, _f, _f, _f, _f, _i, _vp]),
    'fsvit_adamw_step_multi': (_i, [_vp, _i, _sz, _f, _f, _f, _f, _f, _i, _vp]),
    'fsvit_proj_mlp_rows': (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _i, _fp, _vp, _i, _fp, _i, _i, _i, _vp]),
    'fsvit_ln_linear_rows': (_i, [_vp, _vp, _vp, _i, _fp, _i, _i, _i, _f, _vp]),
    'fsvit_patch_embed2x2': (_i, [_vp, _vp, _vp, _i, _fp, _fp, _i, _i, _i, _i, _vp]),
    'fsvit_vit_block_tail': (_i, [_vp, _vp, _vp, _vp, _i, _i, _fp, _vp, _i, _fp, _vp, _i, _fp, _i, _i, _i, _f, _vp]),
    'fsvit_mlp_rows': (_i, [_vp, _vp, _vp, _i, _fp, _vp, _i, _fp, _i, _i, _i, _vp]),
    'fsvit_attention': (_i, [_vp, _vp, _i, _i, _i, _i, _f, _i, _vp]),
    'fsvit_qkv_attention': (_i, [_vp, _vp, _i, _fp, _vp, _i, _i, _i, _i, _i, _f, _vp]),
    'fsvit_vit_ln_qkv_attention': (_i, [_vp, _vp, _i, _fp, _vp, _i, _i, _i, _i, _i, _f, _f, _vp]),
    'fsvit_im2col27': (_i, [_fp, _vp, _i, _i, _i, _i, _vp]),
    'fsvit_stem_conv1': (_i, [_fp, _vp, _i, _fp, _vp, _vp, _i, _i, _i, _vp]),
    'fsvit_maxpool2_pos': (_i, [_vp, _fp, _vp, _i, _i, _i, _i, _i, _vp]),
    'fsvit_pool_affine': (_i, [_vp, _fp, _fp, _fp, _i, _i, _i, _i, _vp]),
    'fsvit_visformer_trainer_create': (_i, [C.POINTER(VisformerCfg), _i, C.POINTER(_vp)]),
    'fsvit_visformer_trainer_destroy': (None, [_vp]),
    'fsvit_visformer_trainer_workspace_bytes': (_sz, [_vp, C.POINTER(Param), _i, _i, _f]),
    'fsvit_visformer_train_forward': (_i, [_vp, C.POINTER(Param), _i, _fp, _i, _i, _i, _f, _fp, _fp, _vp, _sz, _vp]),
    'fsvit_visformer_train_backward': (_i, [_vp, C.POINTER(Param), _i, _fp, _vp]),
    'fsvit_vit_trainer_create': (_i, [C.POINTER(VitCfg), _i, C.POINTER(_vp)]),
    'fsvit_vit_trainer_destroy': (None, [_vp]),
    'fsvit_vit_trainer_droppath_calls': (_i, [_vp, _f]),
    'fsvit_vit_trainer_workspace_bytes': (_sz, [_vp, C.POINTER(Param), _i, _i, _f]),
    'fsvit_vit_train_forward': (_i, [_vp, C.POINTER(Param), _i, _fp, _i, _i, _i, _f, _fp, _fp, _vp, _sz, _vp]),
    'fsvit_vit_train_backward': (_i, [_vp, C.POINTER(Param), _i, _fp, _vp]),
    'fsvit_proto_head_backward': (_i, [_fp, _fp, _fp, _i, _i, _i, _i, _i, _f, _fp, _fp, _fp, _vp]),
    'fsvit_proto_head_backward_sqr': (_i, [_fp, _fp, _fp, _i, _i, _i, _i, _i, _f, _fp, _fp, _fp, _vp]),
    'fsvit_proto_head_ce': (_i, [_fp, _fp, _vp, _i, _i, _i, _i, _i, _f, _fp, _i, _fp, _fp, _fp, _fp, _fp, _vp, _vp]),
    'fsvit_proto_head_ce_backward': (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _f, _fp, _i, _fp, _fp, _fp, _vp, _vp]),
    'fsvit_proto_head_backward_devtemp': (_i, [_fp, _fp, _fp, _i, _i, _i, _i, _i, _fp, _i, _fp, _fp, _fp, _vp]),
    'fsvit_visformer_trainer_set_freeze_bn': (_i, [_vp, _i]),
    'fsvit_sgd_step_multi': (_i, [_vp, _i, _sz, _f, _f, _f, _i, _vp]),
    'fsvit_sgd_step': (_i, [_fp, _fp, _fp, _sz, _f, _f, _f, _i, _vp]),
    'fsvit_conv1x1_wgrad': (_i, [_vp, _vp, _fp, _i, _i, _i, _i, _vp]),
    'fsvit_conv3x3_wgrad': (_i, [_vp, _vp, _fp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'fsvit_gconv3x3': (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _i, _vp]),
    'fsvit_image_transform_gather': (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i,
                                          C.POINTER(C.c_float), C.POINTER(C.c_float), _fp, _vp]),
    'fsvit_attention_backward': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp]),
    'fsvit_sampler_draw': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
}

_lib = None


def load():
    """Load libfsvit.so once; raises ImportError (loudly) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f'{LIB_PATH} not found: build the HIP extension first '
            f'(python -c "import __graft_entry__ as g; g.build()" or make -C few-shot-vit_amd/csrc). '
            f'There is no CPU/PyTorch fallback for the fsvit hot path.')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class FsvitError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f'fsvit error {code}: {msg}')
        self.code = code


def check(rc):
    """Map the C return code to the exception the reference would raise for the same mistake."""
    if rc == 0:
        return
    msg = load().fsvit_last_error().decode()
    if rc == ERR_KEY:
        raise KeyError(msg)                 # load_state_dict / registry lookups
    if rc == ERR_IMG_SIZE:
        raise AssertionError(msg)           # PatchEmbed assert (visformer.py:283-284)
    if rc == ERR_ARG:
        raise ValueError(msg)
    raise FsvitError(rc, msg)
