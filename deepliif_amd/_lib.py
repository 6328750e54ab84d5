"""ctypes binding of libdeepliif_hip.so (include/deepliif_hip.h).

The library is the product: importing this module without the built .so raises, there is no CPU or PyTorch fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DEEPLIIF_AMD_LIB: another build of the same ABI (tools/ load libdeepliif_hip_dev.so, the -DDL_DEV_SWITCHES build with every A/B variant)
LIB_PATH = os.environ.get('DEEPLIIF_AMD_LIB') or os.path.join(_HERE, 'libdeepliif_hip.so')
# the same sources built with IEEE half as the 16-bit type (csrc/common.h, csrc/Makefile): the fp16 INFERENCE policy (engine.Precision 'fp16')
LIB_PATH_F16 = os.environ.get('DEEPLIIF_AMD_LIB_F16') or os.path.join(_HERE, 'libdeepliif_hip_f16.so')
HALF_BF16, HALF_FP16 = 0, 1

DL_F32, DL_BF16 = 0, 1
PREC_BF16, PREC_BF16X3 = 1, 3
ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3, 4
PAD_ZERO, PAD_REFLECT = 0, 1
NORM_INSTANCE, NORM_BATCH = 0, 1
LOSS_BCE_LOGITS, LOSS_MSE, LOSS_SMOOTH_L1, LOSS_L1, LOSS_LINEAR = 0, 1, 2, 3, 4
MAX_TAPS, MAX_PHASES = 64, 4
WGRAD_MULTI_MAX = 24
DL_VERSION = 114

i32 = C.c_int32


class ConvDesc(C.Structure):
    _fields_ = [('N', i32), ('Hi', i32), ('Wi', i32), ('Ci', i32), ('in_pstride', i32),
                ('Ho', i32), ('Wo', i32), ('Co', i32), ('out_pstride', i32),
                ('Hq', i32), ('Wq', i32), ('out_step', i32), ('in_step', i32), ('n_phase', i32),
                ('phase_oh', i32 * MAX_PHASES), ('phase_ow', i32 * MAX_PHASES),
                ('phase_tap_begin', i32 * (MAX_PHASES + 1)), ('phase_kbase', i32 * MAX_PHASES),
                ('tap_dh', C.c_int8 * MAX_TAPS), ('tap_dw', C.c_int8 * MAX_TAPS),
                ('pad_mode', i32), ('w_kstride', i32), ('w_rows', i32), ('act', i32),
                ('in_dtype', i32), ('out_dtype', i32), ('prec', i32), ('splitk', i32), ('in_act', i32), ('bias_n', i32), ('raw_out', i32), ('ci_real', i32), ('in_split', i32)]



class ConvBnStats(C.Structure):      # dl_conv_bnstats
    _fields_ = [('y', C.c_void_p), ('y_pstride', C.c_int32), ('act', C.c_int32),
                ('mean', C.c_void_p), ('rstd', C.c_void_p), ('scale', C.c_void_p), ('shift', C.c_void_p)]

class WgradDesc(C.Structure):
    _fields_ = [('N', i32), ('Hp', i32), ('Wp', i32), ('CAp', i32), ('p_pstride', i32),
                ('Hq', i32), ('Wq', i32), ('CBp', i32), ('q_pstride', i32),
                ('KH', i32), ('KW', i32), ('step', i32), ('pad', i32), ('pad_mode', i32),
                ('CA', i32), ('CB', i32), ('dtype', i32), ('prec', i32), ('splitk', i32), ('accumulate', i32),
                ('q_act', i32), ('p_act', i32), ('pad_w', i32), ('stack_kw', i32), ('p_split', i32), ('q_split', i32)]


class WgradReduceEntry(C.Structure):
    _fields_ = [('slab', C.c_void_p), ('grad', C.c_void_p),
                ('splitk', i32), ('CAp', i32), ('CBp', i32), ('J', i32), ('CA', i32), ('CB', i32), ('KK', i32), ('accumulate', i32), ('stack_kw', i32),
                ('block0', i32), ('nblocks', i32), ('kstride', i32)]


class PackDesc(C.Structure):
    _fields_ = [('A', i32), ('B', i32), ('KH', i32), ('KW', i32), ('row_is_a', i32),
                ('rows_real', i32), ('rows_pad', i32), ('Cc', i32), ('Cc_pad', i32), ('n_phase', i32),
                ('phase_tap_begin', i32 * (MAX_PHASES + 1)), ('phase_kbase', i32 * MAX_PHASES),
                ('tap_kh', C.c_int8 * MAX_TAPS), ('tap_kw', C.c_int8 * MAX_TAPS), ('kstride', i32), ('stack_kw', i32)]


class NormDesc(C.Structure):
    _fields_ = [('N', i32), ('H', i32), ('W', i32), ('Cp', i32), ('C', i32),
                ('y_pstride', i32), ('z_pstride', i32), ('r_pstride', i32),
                ('dtype', i32), ('scope', i32), ('act', i32), ('eps', C.c_float), ('momentum', C.c_float), ('ext_nchunks', i32)]


_vp, _f, _i, _i64 = C.c_void_p, C.c_float, C.c_int, C.c_int64

# every symbol include/deepliif_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    'dl_version': (_i, []),
    'dl_last_error': (C.c_char_p, []),
    'dl_switch_count': (_i, []),
    'dl_switch_name': (C.c_char_p, [_i]),
    'dl_switches_reload': (None, []),
    'dl_half_format': (_i, []),
    'dl_dev_build': (_i, []),
    'dl_conv_forward': (_i, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'dl_conv_stats_chunks': (_i, [C.POINTER(ConvDesc)]),
    'dl_conv_bnstats_chunks': (_i, [C.POINTER(ConvDesc)]),
    'dl_convt4_gather': (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _i, _vp, _i, _i, _vp]),
    'dl_pp_ws_bytes': (C.c_size_t, [_i, _i]),
    'dl_pp_kde_first_minimum': (_i, [_vp, _i, _i, _vp, _vp]),
    'dl_pp_cells': (_i, [_vp, C.c_size_t, _vp, C.c_size_t, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    'dl_pp_finish': (_i, [_vp, C.c_size_t, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, C.c_size_t, _vp, C.c_size_t, _vp]),
    'dl_conv_forward_bnstats': (_i, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, C.POINTER(ConvBnStats), _vp]),
    'dl_conv_kernel_name': (C.c_char_p, [C.POINTER(ConvDesc)]),
    'dl_conv_add_supported': (_i, [C.POINTER(ConvDesc)]),
    'dl_conv_forward_add': (_i, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    'dl_conv_wgrad': (_i, [C.POINTER(WgradDesc), _vp, _vp, _vp, _vp, _vp]),
    'dl_conv_wgrad_deferrable': (_i, [C.POINTER(WgradDesc)]),
    'dl_wgrad_slab_floats': (C.c_size_t, [C.POINTER(WgradDesc)]),
    'dl_conv_wgrad_slabs': (_i, [C.POINTER(WgradDesc), _vp, _vp, _vp, _vp, C.POINTER(WgradReduceEntry), _vp]),
    'dl_wgrad_reduce_batch': (_i, [_vp, _i, _i, _vp]),
    'dl_wgrad_plan': (_i, [C.POINTER(WgradDesc), C.POINTER(i32), C.POINTER(i32), C.POINTER(C.c_char_p)]),
    'dl_conv_wgrad_multi': (_i, [C.POINTER(WgradDesc), _i, _vp, _vp, _vp, _vp, C.POINTER(WgradReduceEntry), _vp]),
    'dl_pack_weights': (_i, [C.POINTER(PackDesc), _vp, _vp, _vp, _vp]),
    'dl_pack_job_bytes': (C.c_size_t, []),
    'dl_pack_job_fill': (_i, [C.POINTER(PackDesc), _vp, _vp, _vp, _vp]),
    'dl_pack_batch_blocks': (_i, [_vp, _i, _vp]),
    'dl_pack_weights_batch': (_i, [_vp, _vp, _i, _vp]),
    'dl_norm_ws_floats': (C.c_size_t, [C.POINTER(NormDesc)]),
    'dl_norm_forward': (_i, [C.POINTER(NormDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'dl_norm_backward': (_i, [C.POINTER(NormDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    'dl_act_forward': (_i, [_i, _i, _vp, _i, _vp, _i, _i64, _i, _vp]),
    'dl_act_backward': (_i, [_i, _i, _vp, _i, _vp, _i, _vp, _i, _i64, _i, _vp]),
    'dl_dropout': (_i, [_i, _vp, _i, _vp, _i, _i64, _i, _f, C.c_uint64, _vp]),
    'dl_axpby': (_i, [_i, _f, _vp, _i, _f, _vp, _i, _vp, _i, _i64, _i, _vp]),
    'dl_gate_forward': (_i, [_i, _vp, _i, _vp, _i, _vp, _i, _i64, _i, _vp]),
    'dl_gate_backward': (_i, [_i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i64, _i, _vp]),
    'dl_copy_channels': (_i, [_i, _vp, _i, _i, _vp, _i, _i, _i64, _i, _i, _vp]),
    'dl_channel_sum': (_i, [_i, _vp, _i, _i64, _i, _i, _vp, _i, _vp, _vp]),
    'dl_nchw_to_nhwc': (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _vp]),
    'dl_nhwc_to_nchw': (_i, [_i, _vp, _i, _i, _vp, _i, _i, _i, _i, _vp]),
    'dl_conv_narrow_supported': (_i, [_i, _i, _i, _i, _i, _i, _i, _i]),
    'dl_conv_narrow_forward': (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _i, _i, _vp]),
    'dl_conv_narrow_forward_x3': (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _i, _i, _vp]),
    'dl_shift_sum': (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _vp, _i, _i, _vp]),
    'dl_shift_stack': (_i, [_i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    'dl_reflect_fold': (_i, [_i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'dl_loss_ws_floats': (C.c_size_t, []),
    'dl_loss': (_i, [_i, _i, _vp, _i, _vp, _i, _f, _i64, _i, _i, _vp, _vp, _i, _f, _vp, _vp]),
    'dl_loss_acc': (_i, [_i, _i, _vp, _i, _vp, _i, _f, _i64, _i, _i, _vp, _f, _i, _vp, _i, _f, _vp, _vp]),
    'dl_upsample2_nearest': (_i, [_i, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp]),
    'dl_kldiv_ws_floats': (C.c_size_t, []),
    'dl_kldiv': (_i, [_i, _vp, _i, _vp, _i, _i64, _i, _i, _vp, _f, _i, _vp, _i, _f, _vp, _vp]),
    'dl_maxpool2_forward': (_i, [_i, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp]),
    'dl_maxpool2_backward': (_i, [_i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp]),
    'dl_adam_step': (_i, [_vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _i, _f, _vp]),
    'dl_adam_hyper': (_i, [_f, _f, _f, _f, _i, _f, _vp]),
    'dl_adam_step_dev': (_i, [_vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    'dl_tile_gather_u8': (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _i, _i, C.c_uint32, _vp, _i, _vp, _i, _i, _vp]),
    'dl_tile_gray_stats_u8': (_i, [_vp, _i64, _i, _i, _vp, _i, _i, _i, C.c_uint32, _vp, _vp]),
    'dl_tile_paste_u8': (_i, [_i, _vp, _i, _i, _vp, _i, _vp, _i64, _vp]),
    'dl_probe_mfma16': (_i, [_vp, _vp, _vp, _vp]),
    'dl_probe_trread': (_i, [_vp, _vp, _vp]),
    'dl_probe_mfma_sustained_elems': (C.c_size_t, [_i]),
    'dl_probe_mfma_sustained': (_i, [_vp, _i, _i, _vp, _vp]),
}

_libs = {}


class HipLibraryError(RuntimeError):
    pass


def load(half: str = 'bf16'):
    """Load (once) and return the ctypes library of one 16-bit format ('bf16': libdeepliif_hip.so, 'fp16': libdeepliif_hip_f16.so);
    raises HipLibraryError if it has not been built."""
    lib = _libs.get(half)
    if lib is not None:
        return lib
    if half not in ('bf16', 'fp16'):
        raise ValueError(f'unknown 16-bit format {half!r} (bf16 | fp16)')
    path = LIB_PATH if half == 'bf16' else LIB_PATH_F16
    if not os.path.exists(path):
        raise HipLibraryError(
            f'{path} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
            f'(or `make -C deepliif_amd/csrc`). deepliif_amd has no CPU / PyTorch fallback.')
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)           # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.dl_version() != DL_VERSION:
        raise HipLibraryError(f'{os.path.basename(path)} version {lib.dl_version()} != {DL_VERSION} (stale build)')
    if lib.dl_half_format() != (HALF_BF16 if half == 'bf16' else HALF_FP16):
        raise HipLibraryError(f'{path} was built for the other 16-bit format (dl_half_format() = {lib.dl_half_format()})')
    _libs[half] = lib
    return lib


def check(rc, what, lib=None):
    if rc != 0:
        msg = (lib or load()).dl_last_error().decode('utf-8', 'replace')
        raise HipLibraryError(f'{what} failed (rc={rc}): {msg}')
