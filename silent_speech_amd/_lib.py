"""ctypes binding of libsilent_speech_hip.so (C ABI declared in include/silent_speech_hip.h).

The product path has exactly one backend: the gfx950 shared library built by __graft_entry__.build()
(or `make -C silent_speech_amd/csrc`).  If it is missing, or a tensor is not on an AMD GPU, calls
raise -- there is no CPU fallback.  tests/ may inject a different handle (the host-emulator build of
the same kernel sources, tools/emu) through `use_library_for_testing`; nothing in the package does.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get('SS_AMD_LIBRARY') or os.path.join(_HERE, 'lib', 'libsilent_speech_hip.so')      # override: A/B runs of two builds on one box

SS_F32, SS_BF16, SS_F64, SS_F32X3 = 0, 1, 2, 3
ABI_VERSION = 9          # include/silent_speech_hip.h: SS_ABI_VERSION (struct layouts / signatures this binding was written against)
OP_KC, OP_OC = 0, 1


class RowMap(ctypes.Structure):
    _fields_ = [('base', ctypes.c_int64), ('batch_stride', ctypes.c_int64), ('row_stride', ctypes.c_int64),
                ('rows_per_batch', ctypes.c_int32)]


class GemmEpilogue(ctypes.Structure):
    _fields_ = [('bias', ctypes.c_void_p), ('gate', ctypes.c_void_p), ('gate_scale', ctypes.c_float),
                ('alpha', ctypes.c_float), ('relu', ctypes.c_int32), ('dropout_p', ctypes.c_float),
                ('seed', ctypes.c_uint64), ('rng_stream', ctypes.c_uint32), ('mode', ctypes.c_int32),
                ('col_mod', ctypes.c_int32), ('col_mul', ctypes.c_int32), ('col_div_mul', ctypes.c_int32),
                ('log_clamp', ctypes.c_float), ('c2', ctypes.c_void_p), ('cmap2', RowMap), ('col_stride2', ctypes.c_int64),
                ('col_sum', ctypes.c_void_p), ('col_sumsq', ctypes.c_void_p), ('col_shift', ctypes.c_void_p),
                ('planes_hi', ctypes.c_void_p), ('planes_lo', ctypes.c_void_p), ('planes_only', ctypes.c_int32),
                ('sign_out', ctypes.c_void_p), ('sign_pitch', ctypes.c_int64), ('gate_bits', ctypes.c_void_p), ('gate_bits_pitch', ctypes.c_int64)]


class DwJob(ctypes.Structure):
    _fields_ = [('A', ctypes.c_void_p), ('B', ctypes.c_void_p), ('C', ctypes.c_void_p), ('amap', RowMap), ('bmap', RowMap),
                ('ldc', ctypes.c_int64), ('M', ctypes.c_int32), ('N', ctypes.c_int32), ('K', ctypes.c_int32), ('flags', ctypes.c_int32)]


class ProfileRow(ctypes.Structure):
    _fields_ = [('name', ctypes.c_char * 64), ('calls', ctypes.c_int64), ('seconds', ctypes.c_double), ('flops', ctypes.c_double), ('bytes', ctypes.c_double)]


class PermuteJob(ctypes.Structure):
    _fields_ = [('inp', ctypes.c_void_p), ('out', ctypes.c_void_p), ('s0', ctypes.c_int64), ('s1', ctypes.c_int64), ('s2', ctypes.c_int64),
                ('o0', ctypes.c_int64), ('o1', ctypes.c_int64), ('d0', ctypes.c_int32), ('d1', ctypes.c_int32), ('d2', ctypes.c_int32),
                ('valid1', ctypes.c_int32), ('valid2', ctypes.c_int32), ('in_dtype', ctypes.c_int32), ('out_dtype', ctypes.c_int32),
                ('accumulate', ctypes.c_int32), ('scale', ctypes.c_float), ('first_block', ctypes.c_int32), ('nblocks', ctypes.c_int32),
                ('pad_', ctypes.c_int32)]


_P, _I, _F, _L = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int64
_U64, _U32 = ctypes.c_uint64, ctypes.c_uint32

# name -> argtypes; every function returns int (0 = ok) unless listed in _RESTYPES
SIGNATURES = {
    'ss_gemm': [_I, _I, _I, _I, _P, _P, _P, _I, _I, _I, ctypes.POINTER(RowMap), ctypes.POINTER(RowMap),
                ctypes.POINTER(RowMap), ctypes.POINTER(GemmEpilogue), _I, _P],
    'ss_gemm_dw_grouped': [_I, ctypes.POINTER(DwJob), _P],
    'ss_split_planes': [_P, _P, _P, _L, _P],
    'ss_gemm_planes': [_I, _P, _P, _P, _P, _P, _I, _I, _I, ctypes.POINTER(RowMap), ctypes.POINTER(RowMap), ctypes.POINTER(RowMap), ctypes.POINTER(GemmEpilogue), _P],
    'ss_permute3d': [_P, _I, _P, _I, _I, _I, _I, _L, _L, _L, _I, _I, _F, _I, _P],
    'ss_permute3d_batch': [_P, _P, _I, _I, _P],
    'ss_dtw_align': [_P, _P, _I, _I, _I, _P, _P, _P],
    'ss_dtw_align_skewed': [_P, _I, _P, _P, _P],
    'ss_reflect_pad_ragged': [_P, _P, _P, _P, _I, _I, _I, _L, _I, _P],
    'ss_concat_pad': [_P, _I, _P, _L, _I, _P],
    'ss_dtw_cumulative': [_I, _P, _L, _L, _I, _I, _P, _P, _P],
    'ss_relpos_attention_forward': [_I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _F, _U64, _U32, _P],
    'ss_relpos_attention_backward': [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _F, _U64, _U32, _P],
    'ss_relpos_attention_forward_p': [_I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _F, _U64, _U32, _P],
    'ss_relpos_attention_backward_p': [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _F, _U64, _U32, _P],
    'ss_relpos_attention_prepare_tables': [_P, _P, _I, _I, _I, _I, _F, _P],
    'ss_relpos_attention_x3_prepare_tables': [_P, _P, _I, _I, _I, _I, _F, _P],
    'ss_relpos_attention_x3_forward': [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _F, _U64, _U32, _P],
    'ss_relpos_attention_x3_backward': [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _F, _U64, _U32, _P],
    'ss_bn_stats_sums': [_I, _P, _I, _I, _I, _I, _P, _P, _P, _P],
    'ss_bn_finalize': [_P, ctypes.c_double, _I, _P, _P, _P, _P, _F, _F, _I, _P],
    'ss_bn_finalize_shift': [_P, _P, ctypes.c_double, _I, _P, _P, _P, _P, _F, _F, _I, _P],
    'ss_bn_apply': [_I, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P],
    'ss_bn_backward_sums': [_I, _P, _I, _P, _I, _P, _I, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P],
    'ss_bn_backward_apply': [_I, _P, _I, _P, _I, _P, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P, ctypes.c_double, _P, _I, _P, _I, _I, _I, _I, _I, _P, _P, _P],
    'ss_colsum': [_I, _P, _I, _I, _L, _P, _P, _P],
    'ss_add_dropout_layernorm_forward': [_I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _F, _U64, _U32, _P],
    'ss_add_dropout_layernorm_forward_planes': [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _F, _U64, _U32, _P],
    'ss_layernorm_backward': [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _U64, _U32, _P],
    'ss_layernorm_backward_bias': [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _U64, _U32, _P],
    'ss_layernorm_backward_ws': [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _F, _U64, _U32, _P],
    'ss_layernorm_backward_ws_planes': [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _F, _U64, _U32, _P],
    'ss_emg_prepare': [_I, _P, _P, _P, _I, _I, _I, _I, _P],
    'ss_frame_lse': [_P, _L, _I, _I, _I, _P, _P, _P],
    'ss_loss_index_tables': [_P, _I, _P, _P, _P, _P, _P, _P],
    'ss_voiced_loss': [_P, _L, _I, _I, _P, _P, _P, _P, _P, _P, _I, _F, _F, _P, _P, _P, _P],
    'ss_silent_cost_skewed': [_P, _L, _I, _P, _P, _P, _P, _I, _I, _I, _F, _P, _P, _P],
    'ss_silent_loss': [_P, _L, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _F, _F, _P, _P, _P, _P],
    'ss_phoneme_confusion': [_P, _P, _P, _P, _P, _I, _P, _P, _P, _I, _P, _I, _P],
    'ss_ctc_loss': [_P, _L, _I, _I, _P, _P, _I, _I, _L, _P, _P, _P, _P, _P, _P, _P],
    'ss_adamw_step': [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _F, _P],
    'ss_cast_f32': [_P, _P, _I, _L, _P],
    'ss_soft_clip': [_P, _P, _L, _I, _P, _P, _F, _F, _P],
    'ss_reflect_pad': [_P, _P, _I, _I, _I, _L, _P],
    'ss_iir_filtfilt': [_P, _P, _I, _I, _I, _P, _P, _L, _P],
    'ss_linear_resample': [_P, _P, _I, _I, ctypes.c_double, ctypes.c_double, _I, _P],
    'ss_iir_filtfilt_batch': [_P, _P, _P, _I, _I, _I, _P, _P, _L, _P],
    'ss_linear_resample_batch': [_P, _P, _P, _I, _I, ctypes.c_double, ctypes.c_double, _L, _P],
    'ss_stft_magnitude': [_P, _L, _I, _P, _L, _I, _P],
    'ss_stft_logmel_fft': [_P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _I, _F, _P, _L, _L, _L, _P],
}
_LP = ctypes.POINTER(ctypes.c_int64)
_HOST_FUNCS = {'ss_dtw_workspace_bytes': ([_I, _I, _LP, _LP, _LP], ctypes.c_int64),
               'ss_dtw_source': ([_I, _I, _L, _L], ctypes.c_int),
               'ss_bn_scratch_floats': ([_I, _I, _I], ctypes.c_int64),
               'ss_iir_filtfilt_workspace_bytes': ([_I, _I, _I], ctypes.c_int64),
               'ss_iir_filtfilt_batch_workspace_bytes': ([_P, _I, _I, _I, _I], ctypes.c_int64),
               'ss_colsum_scratch_floats': ([_I, _I], ctypes.c_int64),
               'ss_gemm_set_blocks_per_cu': ([_I], ctypes.c_int),
               'ss_gemm_last_kernel': ([], ctypes.c_int),
               'ss_gemm_set_option': ([_I, _I], ctypes.c_int),
               'ss_gemm_fuses_column_stats': ([_I, _I, _I, _I, _P, _I, _I, _I, ctypes.POINTER(RowMap), ctypes.POINTER(RowMap), ctypes.POINTER(RowMap),
                                               ctypes.POINTER(GemmEpilogue), _I], ctypes.c_int),
               'ss_gemm_sign_bits_supported': ([_I, _I, _I, _I, _P, _I, _I, _I, ctypes.POINTER(RowMap), ctypes.POINTER(RowMap), ctypes.POINTER(RowMap),
                                                ctypes.POINTER(GemmEpilogue), _I], ctypes.c_int),
               'ss_gemm_planes_supported': ([_I, _P, _I, _I, _I, ctypes.POINTER(RowMap), ctypes.POINTER(RowMap), ctypes.POINTER(RowMap), ctypes.POINTER(GemmEpilogue)], ctypes.c_int),
               'ss_gemm_dw_set_option': ([_I, _I], ctypes.c_int),
               'ss_plan_create': ([_P], _P), 'ss_plan_destroy': ([_P], None), 'ss_plan_slot_count': ([_P], ctypes.c_int),
               'ss_plan_slot_name': ([_P, _I], ctypes.c_char_p), 'ss_plan_bind': ([_P, _I, _P], ctypes.c_int),
               'ss_plan_set_option': ([_P, _I, _I], ctypes.c_int), 'ss_plan_set_reduce_hook': ([_P, _P, _P], ctypes.c_int),
               'ss_plan_set_event_hook': ([_P, _P, _P], ctypes.c_int), 'ss_plan_ctx_bytes': ([], ctypes.c_int64),
               'ss_plan_workspace_bytes': ([_P, _I, _I, _I], ctypes.c_int64),
               'ss_plan_forward': ([_P, _P, _P, _P, _L, _I, _I, _I, _I, _F, _U64, _P, _P, _P], ctypes.c_int),
               'ss_plan_backward': ([_P, _P, _P, _P, _P], ctypes.c_int),
               'ss_counters_add': ([_I, _P, _L, _P], ctypes.c_int),
               'ss_plan_profile': ([_P, _I], ctypes.c_int), 'ss_plan_profile_read': ([_P, _P, _I], ctypes.c_int),
               'ss_relpos_attention_needs_transposed': ([_I, _I, _I, _I], ctypes.c_int),
               'ss_relpos_attention_family': ([_I, _I, _I, _I], ctypes.c_int),
               'ss_relpos_attention_table_bytes': ([_I, _I, _I], ctypes.c_int64),
               'ss_relpos_attention_x3_supported': ([_I, _I, _I], ctypes.c_int),
               'ss_relpos_attention_x3_saved_bytes': ([_I, _I, _I, _I, _I], ctypes.c_int64),
               'ss_relpos_attention_x3_table_bytes': ([_I, _I, _I], ctypes.c_int64),
               'ss_relpos_attention_saved_bytes': ([_I, _I, _I, _I, _I, _I], ctypes.c_int64),
               'ss_layernorm_backward_scratch_floats': ([_I, _I], ctypes.c_int64)}
_RESTYPES = {'ss_last_error': ctypes.c_char_p, 'ss_target_arch': ctypes.c_char_p, 'ss_abi_version': ctypes.c_int}

_lib = None
_is_emulator = False


def _declare(lib):
    for name, restype in _RESTYPES.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = []
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int
    for name, (argtypes, restype) in _HOST_FUNCS.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = restype
    return lib


def load(path=None, _testing=False):
    global _lib, _is_emulator
    path = path or _LIB_PATH
    if not os.path.exists(path):
        raise RuntimeError('silent_speech_amd: %s not found -- build the gfx950 kernels first '
                           '(python -c "import __graft_entry__ as g; g.build()" or make -C silent_speech_amd/csrc). '
                           'There is no CPU fallback.' % path)
    cdll = ctypes.CDLL(path)
    cdll.ss_abi_version.restype = ctypes.c_int
    have = cdll.ss_abi_version()
    if have != ABI_VERSION:
        raise RuntimeError('silent_speech_amd: %s has ABI version %d, this binding needs %d -- rebuild it '
                           '(make -C silent_speech_amd/csrc); struct layouts differ between versions' % (path, have, ABI_VERSION))
    cdll.ss_target_arch.restype = ctypes.c_char_p
    arch = cdll.ss_target_arch()
    if arch != b'gfx950' and not _testing:
        # the product path has ONE backend; the host-emulator build of the kernel sources is test infrastructure and only enters through
        # use_library_for_testing (tests/backend.py) -- never through SS_AMD_LIBRARY or a default path
        raise RuntimeError('silent_speech_amd: %s targets %r, not gfx950 -- the package only loads the MI355X build '
                           '(SS_AMD_LIBRARY selects between gfx950 builds; the emulator library is for tests/ only)' % (path, arch))
    _lib = _declare(cdll)
    _is_emulator = arch != b'gfx950'
    return _lib


def use_library_for_testing(path):
    """tests/ only: run the same kernel sources on the host emulator (CPU tensors)."""
    return load(path, _testing=True)


def lib():
    if _lib is None:
        load()
    return _lib


def is_emulator():
    lib()
    return _is_emulator


def kernels_can_read(t):
    """True when the loaded library's kernels can dereference this tensor's memory: a GPU tensor for the product library (the only case outside
    tests/), a CPU tensor for the host-emulator build that tests/backend.py loads.  The one place where the package asks which of the two it has."""
    return bool(t.is_cuda) != is_emulator()


def kernel_device_for(t):
    """The device a host-side input has to be moved to before a kernel reads it: where it is, if the kernels can read it there; else the GPU."""
    return t.device if kernels_can_read(t) else torch.device('cuda')


def check(rc, what=''):
    if rc != 0:
        raise RuntimeError('%s failed: %s' % (what or 'silent_speech_hip call', lib().ss_last_error().decode()))


def dtype_code(dt):
    if dt == torch.float32:
        return SS_F32
    if dt == torch.bfloat16:
        return SS_BF16
    if dt == torch.float64:
        return SS_F64
    raise TypeError('unsupported dtype %s (float32, bfloat16; float64 for ss_dtw_cumulative only)' % dt)


def ptr(t):
    """Device pointer of a tensor, enforcing that the product library only ever sees GPU memory."""
    if t is None:
        return None
    if not kernels_can_read(t):
        if is_emulator():
            raise RuntimeError('emulator backend (tests only) needs CPU tensors')
        raise RuntimeError('silent_speech_amd: tensor is on %s; the HIP kernels need an AMD GPU tensor '
                           '(no CPU fallback exists)' % t.device)
    return ctypes.c_void_p(t.data_ptr())


def stream_of(t):
    if t is not None and t.is_cuda:
        return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
    return ctypes.c_void_p(0)
