"""ctypes binding of libfql_amd.so (include/fql_amd.h).

The product path has NO fallback: if the shared library is missing or a call fails, this module
raises.  Nothing here imports or calls the CPU oracle.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('FQL_AMD_LIB') or os.path.join(_HERE, 'libfql_amd.so')   # (override: kernel-variant builds under experiments/)

FQL_MAX_HIDDEN = 8
FQL_STREAM_LEGACY = 1   # (void*)1 == hipStreamLegacy: the legacy default stream (torch's default stream)
FQL_NUM_INFO = 13
FQL_OK, FQL_E_INVALID, FQL_E_NODEVICE, FQL_E_HIP, FQL_E_STATE, FQL_E_NOTFOUND = 0, -1, -2, -3, -4, -5


class FqlConfig(C.Structure):
    _fields_ = [
        ('obs_dim', C.c_int32), ('act_dim', C.c_int32),
        ('num_actor_hidden', C.c_int32), ('actor_hidden', C.c_int32 * FQL_MAX_HIDDEN),
        ('num_value_hidden', C.c_int32), ('value_hidden', C.c_int32 * FQL_MAX_HIDDEN),
        ('layer_norm', C.c_int32), ('actor_layer_norm', C.c_int32),
        ('lr', C.c_float), ('discount', C.c_float), ('tau', C.c_float), ('alpha', C.c_float),
        ('q_agg', C.c_int32), ('flow_steps', C.c_int32), ('normalize_q_loss', C.c_int32),
        ('batch_size', C.c_int32), ('precision', C.c_int32), ('encoder', C.c_int32), ('img_h', C.c_int32),
        ('img_w', C.c_int32), ('img_c', C.c_int32), ('reserved', C.c_int32 * 3),
    ]


class FqlNoise(C.Structure):
    _fields_ = [('eps1', C.c_void_p), ('x0', C.c_void_p), ('t', C.c_void_p), ('z', C.c_void_p), ('eps2', C.c_void_p)]


# every symbol include/fql_amd.h declares: (restype, argtypes)
_VP, _I, _I64, _U64, _SZ, _F = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_size_t, C.c_float
_BATCH = [_VP, _VP, _VP, _VP, _VP, _I]
SYMBOLS = {
    'fql_info_name': (C.c_char_p, [_I]),
    'fql_abi_version': (_I, []),
    'fql_default_config': (None, [C.POINTER(FqlConfig)]),
    'fql_last_error': (C.c_char_p, [_VP]),
    'fql_create': (_I, [C.POINTER(FqlConfig), _U64, C.POINTER(_VP)]),
    'fql_destroy': (_I, [_VP]),
    'fql_set_batch_size': (_I, [_VP, _I]),
    'fql_num_leaves': (_I, [_VP]),
    'fql_leaf_info': (_I, [_VP, _I, C.c_char_p, _I, C.POINTER(_I), C.POINTER(_I64)]),
    'fql_get_param': (_I, [_VP, C.c_char_p, _VP, _SZ]),
    'fql_set_param': (_I, [_VP, C.c_char_p, _VP, _SZ]),
    'fql_get_opt_state': (_I, [_VP, _I, C.c_char_p, _VP, _SZ]),
    'fql_set_opt_state': (_I, [_VP, _I, C.c_char_p, _VP, _SZ]),
    'fql_get_step': (_I, [_VP, C.POINTER(_I64), C.POINTER(_I64)]),
    'fql_set_step': (_I, [_VP, _I64, _I64]),
    'fql_update': (_I, [_VP] + _BATCH + [C.POINTER(FqlNoise), _VP, _VP]),
    'fql_update_begin': (_I, [_VP] + _BATCH + [C.POINTER(FqlNoise), _VP]),
    'fql_update_end': (_I, [_VP, _VP, _VP]),
    'fql_update_end_split': (_I, [_VP, _VP, _VP, _VP]),
    'fql_update_begin_split': (_I, [_VP] + _BATCH + [C.POINTER(FqlNoise), _VP, _VP]),
    'fql_update_from_dataset_begin_split': (_I, [_VP, _VP, _I, _I64, _I64, C.POINTER(FqlNoise), _VP, _VP]),
    'fql_grad_buckets': (_I, [_VP, C.POINTER(_SZ), C.POINTER(_SZ)]),
    'fql_grad_buffer': (_I, [_VP, C.POINTER(_VP), C.POINTER(_SZ)]),
    'fql_set_grad_scale': (_I, [_VP, _F]),
    'fql_set_rng_stream': (_I, [_VP, _U64]),
    'fql_info_enqueue': (_I, [_VP, _VP, C.POINTER(_U64)]),
    'fql_info_wait': (_I, [_VP, _U64, _VP]),
    'fql_total_loss': (_I, [_VP] + _BATCH + [C.POINTER(FqlNoise), C.POINTER(_F), C.POINTER(_F), _VP]),
    'fql_sample_actions': (_I, [_VP, _VP, _I, _VP, _U64, _VP, _VP]),
    'fql_flow_actions': (_I, [_VP, _VP, _VP, _I, _VP, _VP]),
    'fql_dataset_upload': (_I, [_VP, _I64, _I64, _VP, _VP, _VP, _VP, _VP]),
    'fql_dataset_add': (_I, [_VP, _VP, _VP, _F, _F, _VP]),
    'fql_dataset_size': (_I, [_VP, C.POINTER(_I64), C.POINTER(_I64)]),
    'fql_noise_from_jax_keys': (_I, [_VP, _VP, _I, _I, C.POINTER(FqlNoise), _VP]),
    'fql_dataset_reserve': (_I, [_VP, _I64]),
    'fql_dataset_add_frames': (_I, [_VP, _VP, _VP, _VP, _F, _F]),
    'fql_replay_create': (_I, [_VP, _I64]),
    'fql_replay_add': (_I, [_VP, _VP, _VP, _F, _F, _VP]),
    'fql_replay_add_frames': (_I, [_VP, _VP, _VP, _VP, _F, _F]),
    'fql_replay_size': (_I, [_VP, C.POINTER(_I64), C.POINTER(_I64)]),
    'fql_update_balanced': (_I, [_VP, _VP, _VP, _VP, _I, C.POINTER(FqlNoise), _VP, _VP]),
    'fql_update_from_dataset': (_I, [_VP, _VP, _I, _I64, _I64, C.POINTER(FqlNoise), _VP, _VP]),
    'fql_update_from_dataset_begin': (_I, [_VP, _VP, _I, _I64, _I64, C.POINTER(FqlNoise), _VP]),
    'fql_dataset_upload_frames': (_I, [_VP, _I64, _VP, _VP, _VP, _VP, _VP, _VP, _I, _F]),
    'fql_update_from_frames': (_I, [_VP, _VP, _VP, _I, _I64, _I64, C.POINTER(FqlNoise), _VP, _VP]),
    'fql_read_info': (_I, [_VP, C.POINTER(_F)]),
    'fql_synchronize': (_I, [_VP, C.POINTER(_I)]),
    'fql_stats': (_I, [_VP, C.POINTER(_I64), C.POINTER(_I64), C.POINTER(_I64)]),
    'fql_stream': (_VP, [_VP]),
}

_lib = None


def load():
    """Load libfql_amd.so and bind every declared symbol.  Raises if the extension is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
            '(hipcc --offload-arch=gfx950).  fql_amd has no CPU fallback.')
    lib = C.CDLL(LIB_PATH)
    lax = 'FQL_AMD_LIB' in os.environ    # experiments load older builds through the env override: bind what they export
    for name, (res, args) in SYMBOLS.items():
        if lax and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.fql_abi_version() != 1:
        raise RuntimeError('libfql_amd.so ABI version mismatch')
    _lib = lib
    return lib


class FqlError(RuntimeError):
    pass


def check(lib, handle, rc):
    if rc == FQL_OK:
        return
    msg = lib.fql_last_error(handle)
    msg = msg.decode() if msg else f'error {rc}'
    if rc == FQL_E_INVALID:
        raise ValueError(msg)
    if rc == FQL_E_NOTFOUND:
        raise KeyError(msg)
    raise FqlError(f'[{rc}] {msg}')
