"""ctypes binding of ``libnd_hip.so`` (C ABI declared in ``include/nd_hip.h``).

There is no CPU fallback: if the library is missing, cannot be loaded, or the device is not gfx950 the sampling
path raises.  PyTorch is used only for device memory, streams and graph capture.
"""
import ctypes
import os

_LIB = None
_LIB_PATH = os.environ.get('ND_HIP_LIB') or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libnd_hip.so')

# flags (mirror include/nd_hip.h)
CONV_IN_UP2X = 1
CONV_RES_UP2X = 2
CONV_SILU_OUT = 4
CONV_GN_SILU = 8
CONV_OUT_F32 = 16
DT_F32 = 0
DT_BF16 = 1
GN_SILU = 1
GN_POOL2 = 2
VAR_FIXED = 0
VAR_LEARNED = 1
VAR_LEARNED_INTERP = 2
COEF_COLS = 8
STEP_NO_CLIP = 1
STEP_PER_IMAGE = 2

_vp = ctypes.c_void_p
_i = ctypes.c_int
_f = ctypes.c_float
_i64 = ctypes.c_int64
_u64 = ctypes.c_uint64

# name -> argtypes; every function returns int status except the three listed in _SPECIAL
SIGNATURES = {
    'nd_timestep_embed': [_vp, _vp, _i, _i, _vp, _i, _vp],
    'nd_embedding_add_silu': [_vp, _vp, _vp, _i, _i, _i, _vp, _vp],
    'nd_conv_nhwc': [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _i,
                     _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp],
    'nd_conv1x1_stats_nhwc': [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp],
    'nd_conv_splitk_nhwc': [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _i,
                            _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp],
    'nd_conv3x3_winograd_splitk_nhwc': [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _i,
                                        _i, _i, _i, _i, _i, _i, _i, _vp, _vp],
    'nd_conv3x3_winograd_nhwc': [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _i,
                                 _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp],
    'nd_conv3x3_winograd_f4_nhwc': [_vp, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp],
    'nd_repack_conv_weight_winograd_f4': [_vp, _vp, _i, _i, _i, _vp],
    'nd_conv_winograd_f4_stats_rows': [_i, _i, _i, _i],
    'nd_conv_winograd_f4_splitk_stats_rows': [_i, _i, _i, _i],
    'nd_groupnorm_stats_from_partials': [_vp, _i, _i, _vp, _i, _i, _vp, _i, _i, _vp],
    'nd_conv3x3_winograd_stats_nhwc': [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _i,
                                       _i, _i, _i, _i, _i, _vp, _vp],
    'nd_conv3x3_winograd_vstats_nhwc': [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _i,
                                        _i, _i, _i, _i, _i, _i, _vp, _vp],
    'nd_groupnorm_channel_partials_nhwc': [_vp, _i, _i, _vp, _i, _i, _i, _vp],
    'nd_groupnorm_coeffs': [_vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp],
    'nd_groupnorm_coeffs_from_partials': [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _f, _vp],
    'nd_repack_conv_weight_winograd': [_vp, _vp, _i, _i, _vp],
    'nd_conv_direct_nhwc': [_vp, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    'nd_repack_conv_weight': [_vp, _vp, _i, _i, _i, _vp],
    'nd_groupnorm_stats_nhwc': [_vp, _i, _i, _vp, _i, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp],
    'nd_groupnorm_apply_nhwc': [_vp, _i, _i, _vp, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _i,
                                _i, _i, _i, _i, _f, _i, _i, _vp],
    'nd_groupnorm_fused_nhwc': [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _f, _i, _i, _vp],
    'nd_conv_bf16_nhwc': [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _i,
                          _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp],
    'nd_conv3x3_bf16_stats_nhwc': [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _i,
                                   _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp],
    'nd_conv1x1_bf16_stats_nhwc': [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _i,
                                   _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp],
    'nd_conv_bf16_stats_rows': [_i, _i, _i, _i, _i],
    'nd_conv_bf16_splitk_nhwc': [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _i,
                                 _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp],
    'nd_repack_conv_weight_bf16': [_vp, _vp, _i, _i, _i, _i, _vp],
    'nd_f32_to_bf16_rows': [_vp, _i, _vp, _i, _i, _i64, _vp],
    'nd_attention_bf16_nhwc': [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _vp],
    'nd_attention_nhwc': [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _vp],
    'nd_upsample2x_nhwc': [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp],
    'nd_avgpool2x_nhwc': [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp],
    'nd_space_to_depth2_nhwc': [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp],
    'nd_nchw_to_nhwc': [_vp, _vp, _i, _i, _i, _i, _vp],
    'nd_nhwc_to_nchw': [_vp, _vp, _i, _i, _i, _i, _vp],
    'nd_fill_timestep': [_vp, _vp, _vp, _i, _vp],
    'nd_step_advance': [_vp, _i, _vp],
    'nd_repack_conv_first_weight': [_vp, _vp, _i, _i, _vp],
    'nd_conv3x3_first_stats_rows': [_i, _i, _i, _i],
    'nd_conv3x3_first_nhwc': [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp],
    'nd_conv3x3_taps_gather_nhwc': [_vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    'nd_groupnorm_apply_coeffs_nhwc': [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp],
    'nd_copy_row_by_step': [_vp, _vp, _i, _i, _i64, _vp, _vp],
    'nd_ddim_step': [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _i, _f, _vp, _vp, _f, _vp, _i64, _u64, _vp, _u64, _i, _i, _i, _vp],
    'nd_ddpm_step': [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _i, _f, _vp, _vp, _i, _vp, _i64, _u64, _vp, _u64, _i, _i, _i, _vp],
    'nd_eps_log_var': [_vp, _i, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _vp],
    'nd_qsample_steps': [_vp, _vp, _vp, _i, _i64, _vp, _vp, _vp, _vp],
    'nd_checksum_segments': [_vp, _vp, _i, _vp, _vp],
    'nd_qsample': [_vp, _vp, _vp, _i64, _f, _f, _vp],
    'nd_to_uint8_hwc': [_vp, _i, _vp, _i, _i, _i, _i, _vp],
}
_SPECIAL = {
    'nd_version': ([], _i),
    'nd_conv_num_variants': ([], _i),
    'nd_conv_weight_floats': ([_i, _i, _i], _i64),
    'nd_conv_first_weight_floats': ([_i], _i64),
    'nd_conv_max_weight_read': ([_i, _i, _i, _i], _i64),
    'nd_conv_bf16_weight_elems': ([_i, _i, _i], _i64),
    'nd_conv_bf16_max_weight_read': ([_i, _i, _i, _i, _i], _i64),
    'nd_conv_bf16_splitk_workspace_floats': ([_i, _i, _i, _i, _i, _i, _i], _i64),
    'nd_conv_splitk_workspace_floats': ([_i, _i, _i, _i, _i, _i, _i], _i64),
    'nd_conv1x1_stats_rows': ([_i, _i, _i, _i], _i),
    'nd_conv_bf16_num_variants': ([], _i),
    'nd_conv_bf16_variant_layout': ([_i], _i),
    'nd_conv_bf16_variant_name': ([_i], ctypes.c_char_p),
    'nd_conv_bf16_variant_info': ([_i, ctypes.POINTER(_i), ctypes.POINTER(_i), ctypes.POINTER(_i)], _i),
    'nd_groupnorm_stats_blocks': ([_i, _i, _i, _i], _i),
    'nd_conv_winograd_weight_floats': ([_i, _i], _i64),
    'nd_conv_winograd_max_weight_read': ([_i, _i, _i], _i64),
    'nd_conv_winograd_num_variants': ([], _i),
    'nd_conv_winograd_stats_variant': ([], _i),
    'nd_conv_winograd_stats_rows': ([_i, _i, _i, _i], _i),
    'nd_conv_winograd_stats_floats': ([_i, _i, _i, _i, ctypes.POINTER(_i)], _i64),
    'nd_conv_winograd_variant_info': ([_i] + [ctypes.POINTER(_i)] * 5, _i),
    'nd_conv_winograd_variant_name': ([_i], ctypes.c_char_p),
    'nd_conv_winograd_f4_num_variants': ([], _i),
    'nd_conv_winograd_f4_variant_name': ([_i], ctypes.c_char_p),
    'nd_conv_winograd_f4_variant_info': ([_i] + [ctypes.POINTER(_i)] * 3, _i),
    'nd_conv_winograd_f4_weight_floats': ([_i, _i, _i], _i64),
    'nd_conv_winograd_f4_max_weight_read': ([_i, _i, _i], _i64),
    'nd_conv_select_variant': ([_i, _i, _i, _i, _i, _i, _i], _i),
    'nd_conv_variant_info': ([_i, ctypes.POINTER(_i), ctypes.POINTER(_i), ctypes.POINTER(_i)], _i),
    'nd_last_error': ([], ctypes.c_char_p),
    'nd_build_id': ([], ctypes.c_char_p),
    'nd_build_flags': ([], ctypes.c_char_p),
    'nd_device_arch': ([], ctypes.c_char_p),
}
EXPORTS = sorted(list(SIGNATURES) + list(_SPECIAL))


class NdHipError(RuntimeError):
    pass


def lib_path():
    return _LIB_PATH


def load():
    """Load (once) and return the ctypes library; raises NdHipError if it is not built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # PyTorch first: its wheel bundles its own HIP runtime (torch/lib/libamdhip64.so) and every device pointer this library
    # is handed comes from THAT runtime.  Loaded after torch, libnd_hip.so binds to the runtime already in the process;
    # loaded before it (python __graft_entry__.py smoke: build() then smoke() in one process), the system's
    # /opt/rocm/lib/libamdhip64.so.7 comes in first and the first call on a torch pointer fails (hipMemsetAsync: invalid)
    import torch  # noqa: F401
    if not os.path.exists(_LIB_PATH):
        raise NdHipError('libnd_hip.so not found at {} -- build it with `make -C nice-diffusion_amd` '
                         '(or `python -c "import __graft_entry__ as g; g.build()"`); there is no CPU fallback'
                         .format(_LIB_PATH))
    try:
        L = ctypes.CDLL(_LIB_PATH)
    except OSError as e:   # pragma: no cover
        raise NdHipError('cannot load {}: {}'.format(_LIB_PATH, e))
    for name, argtypes in SIGNATURES.items():
        fn = getattr(L, name)
        fn.argtypes = argtypes
        fn.restype = _i
    for name, (argtypes, restype) in _SPECIAL.items():
        fn = getattr(L, name)
        fn.argtypes = argtypes
        fn.restype = restype
    flags = L.nd_build_flags().decode()
    if flags and os.environ.get('ND_ALLOW_ABLATION') != '1':
        raise NdHipError('{} was compiled with timing-only / diagnostic macros ({}): such a build computes wrong results by '
                         'construction; set ND_ALLOW_ABLATION=1 to load it for a measurement'.format(_LIB_PATH, flags))
    _LIB = L
    return L


def build_id():
    """Source hash of the loaded library (+ its variant flags, if any): the stamp of every measured artefact."""
    L = load()
    flags = L.nd_build_flags().decode()
    return L.nd_build_id().decode() + ('+' + flags.replace(' ', ',') if flags else '')


def source_hash(root=None):
    """What the Makefile would stamp a library built from the sources in the tree with (nd_build_id of a current build)."""
    import glob
    import hashlib
    pkg = root or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = glob.glob(os.path.join(pkg, 'csrc', '*.hip')) + glob.glob(os.path.join(pkg, 'csrc', '*.h')) + \
        glob.glob(os.path.join(pkg, 'csrc', '*.inc')) + [os.path.join(pkg, '..', 'include', 'nd_hip.h')]
    h = hashlib.sha256()
    for f in sorted(files):
        h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


def last_error():
    return load().nd_last_error().decode('utf-8', 'replace')


def check(rc, what=''):
    if rc != 0:
        raise NdHipError('{} failed (rc={}): {}'.format(what or 'libnd_hip call', rc, last_error()))


def ptr(t):
    """Device pointer of a tensor (or None)."""
    return None if t is None else t.data_ptr()


def require_device(t, what='tensor'):
    if not t.is_cuda:
        raise NdHipError('{} lives on {}: this build runs the sampling path on an AMD GPU only (no CPU fallback)'
                         .format(what, t.device))


_ARCH_OK = {}


def require_gfx950(device_index):
    """Fail loudly on anything but gfx950 (the kernels use gfx950 MFMA / 160 KiB LDS)."""
    ok = _ARCH_OK.get(device_index)
    if ok is None:
        import torch
        with torch.cuda.device(device_index):
            arch = load().nd_device_arch().decode()
        ok = arch.startswith('gfx950')
        _ARCH_OK[device_index] = ok
        if not ok:
            raise NdHipError('device {} is "{}", libnd_hip.so is built for gfx950 only'.format(device_index, arch))
    elif not ok:
        raise NdHipError('device {} is not gfx950'.format(device_index))
