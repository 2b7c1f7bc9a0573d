"""ctypes binding of libssp_hip.so (C ABI: include/ssp_hip.h).

The product path has no CPU fallback: if the shared library is missing, or a kernel entry point reports an
error, this module raises.  Nothing here imports anything under oracle/.
"""
import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# (SSP_HIP_LIB: another build of the SAME library - e.g. the previous commit's kernels - for same-box A/B timing)
LIB_PATH = os.environ.get("SSP_HIP_LIB") or os.path.join(_HERE, "libssp_hip.so")


class SspError(RuntimeError):
    pass


_lib = None

P = c_void_p
I = c_int
F = c_float
L = c_int64

# name -> argtypes (all return int unless noted)
_SIGS = {
    "ssp_abi_version": [],
    "ssp_set_option": [c_char_p, I],
    "ssp_conv_fwd": [P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, P, L, P],
    "ssp_conv_fwd_affine": [P, P, P, P, P, F, I, I, I, I, I, I, I, I, I, P, L, P],
    "ssp_conv_stats_tile_m": [I, I, I, I, I, I, I],
    "ssp_conv_workspace_floats": [I, I, I, I, I, I, I],
    "ssp_conv_stats_tiles": [I, I, I, I, I, I, I],
    "ssp_conv_stats_floats": [I, I, I, I, I, I, I],
    "ssp_conv_plan_wino_tile": [I],
    "ssp_conv_wino_tiles": [I, I, I, I],
    "ssp_wino_filter_transform": [P, P, I, I, P],
    "ssp_wino_filter_transform_t": [P, P, I, I, I, P],
    "ssp_conv_dgrad": [P, P, P, I, I, I, I, I, I, I, I, I, I, P, L, P],
    "ssp_conv_dgrad_bnbwd": [P, P, P, I, I, I, I, I, I, I, I, I, P, L, P, I, P, P, P, P, F, P, I, P],
    "ssp_bn_act_bwd_partials": [P, I, P, I, P, I, P, P, P, P, I, I, I, I, F, I, P, I, I, P, P, P, P, P],
    "ssp_conv_wgrad": [P, P, P, I, I, I, I, I, I, I, I, P],
    "ssp_conv_wgrad_wino": [P, P, P, I, I, I, I, I, I, I, P, L, P],
    "ssp_conv_wgrad_wino_workspace_floats": [I, I, I, I, I],
    "ssp_conv_wgrad_wino_t": [P, P, P, I, I, I, I, I, I, I, I, P, L, P],
    "ssp_conv_wgrad_wino_workspace_floats_t": [I, I, I, I, I, I],
    "ssp_bn_fwd_finalize": [P, I, I, I, I, P, P, P, P, F, F, P, P, P, P, P],
    "ssp_bn_eval_prepare": [I, P, P, P, P, F, P, P, P, P, P],
    "ssp_bn_act_fwd": [P, I, P, I, P, P, I, I, I, I, I, F, P],
    "ssp_bn_bwd_blocks": [],
    "ssp_bn_act_bwd": [P, I, P, I, P, I, P, P, P, P, I, I, I, I, I, F, I, P, P, P, P, P, P],
    "ssp_bn_bwd_finalize": [P, I, I, L, I, I, P, P, P, P, P],
    "ssp_first_tile_pixels": [],
    "ssp_first_groups": [I, I, I],
    "ssp_first_fwd_stats": [P, P, P, I, I, I, P],
    "ssp_first_fwd_apply": [P, P, P, P, F, P, I, I, I, I, P],
    "ssp_first_conv_raw": [P, P, P, I, I, I, I, P],
    "ssp_first_bwd_reduce": [P, P, P, I, P, P, P, P, F, P, I, I, I, P],
    "ssp_first_bwd_wgrad": [P, P, P, I, P, P, P, P, P, P, F, P, P, L, I, I, I, P],
    "ssp_wino_input_transform_t": [P, I, P, I, I, I, I, I, P],
    "ssp_first_wgrad_workspace_floats": [I, I, I],
    "ssp_colsum": [P, I, L, I, P, P],
    "ssp_pose_errors": [P, I, P, P, P, I, I, P, P],
    "ssp_pts_diameter": [P, I, P, P, P],
    "ssp_sgd_step": [P, P, P, L, F, F, F, F, I, I, P],
    "ssp_nchw_to_nhwc": [P, P, I, I, I, I, I, I, P],
    "ssp_nhwc_to_nchw": [P, P, I, I, I, I, I, P],
    "ssp_u8hwc_to_nhwc": [P, P, I, I, I, I, I, I, P],
    "ssp_resample_u8": [P, I, I, I, I, P],
    "ssp_distort_u8": [P, P, L, P, I, P],
    "ssp_repack_fwd": [P, P, I, I, I, I, P],
    "ssp_repack_dgrad": [P, P, I, I, I, I, P],
    "ssp_repack_dgrad_packed": [P, P, I, I, I, I, P],
    "ssp_unpack_grad": [P, P, I, I, I, I, P],
    "ssp_reorg": [P, I, P, I, I, I, I, I, I, I, P],
    "ssp_copy_channels": [P, I, P, I, I, L, I, P],
    "ssp_maxpool_fwd": [P, I, P, I, I, I, I, I, P],
    "ssp_maxpool_bwd": [P, I, P, I, P, I, I, I, I, I, I, P],
    "ssp_region_loss": [P, P, I, P, P, P, I, I, I, I, I, I, F, F, F, F, F, I, I, P, I, P],
    "ssp_region_decode_argmax": [P, P, I, I, I, I, I, I, I, P],
    "ssp_region_decode_all": [P, P, I, I, I, I, I, I, P],
    "ssp_pnp_batched": [P, P, P, P, I, I, I, P],
    "ssp_prof_enable": [I],
    "ssp_prof_nkinds": [],
    "ssp_prof_collect": [P, P, P],
}

_RET64 = ('ssp_conv_workspace_floats', 'ssp_conv_wgrad_wino_workspace_floats', 'ssp_conv_wgrad_wino_workspace_floats_t',
          'ssp_conv_stats_floats', 'ssp_conv_wino_tiles', 'ssp_first_wgrad_workspace_floats')

PROF_KINDS = ("conv_fwd", "conv_dgrad", "conv_wgrad", "bn_act", "layout", "region", "optim", "first_block_fwd",
              "first_block_bwd", "wino_fwd", "wino_dgrad", "wino_wgrad", "onchip_fwd", "onchip_dgrad", "onchip_wgrad")


def csrc_digest():
    """sha1 over the kernel sources (csrc/*.hip, *.h, include/ssp_hip.h): lets a profile artefact say which kernels it
    measured (bench.py drops a traffic figure whose digest no longer matches)."""
    import glob
    import hashlib
    h = hashlib.sha1()
    files = sorted(glob.glob(os.path.join(_HERE, 'csrc', '*.hip')) + glob.glob(os.path.join(_HERE, 'csrc', '*.h')))
    files.append(os.path.join(os.path.dirname(_HERE), 'include', 'ssp_hip.h'))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, 'rb') as fp:
            h.update(fp.read())
    return h.hexdigest()


def exported_symbols():
    """Every entry point include/ssp_hip.h declares (used by the CPU-side ABI test)."""
    return ["ssp_last_error"] + list(_SIGS.keys())


def load():
    """Load the library (once).  Raises SspError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SspError(
            "libssp_hip.so not found at %s - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C singleshotpose_amd/csrc`; there is no CPU fallback for the HIP path" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    lib.ssp_last_error.restype = c_char_p
    lib.ssp_last_error.argtypes = []
    for name, args in _SIGS.items():
        fn = getattr(lib, name)
        fn.restype = c_int64 if name in _RET64 else c_int
        fn.argtypes = args
    _lib = lib
    return lib


PROF_MASK = [0]      # last mask handed to ssp_prof_enable (engine.Plan keeps its hipGraph replays off while launches are timed)


def call(name, *args):
    """Invoke an entry point; raise SspError with the library's message on a non-zero return."""
    lib = load()
    if name == 'ssp_prof_enable':
        PROF_MASK[0] = int(args[0])
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.ssp_last_error()
        raise SspError("%s failed (%d): %s" % (name, rc, msg.decode() if msg else "?"))
    return rc


def query(name, *args):
    """Entry points that return a value instead of a status (ssp_abi_version, ssp_conv_stats_tile_m, ...)."""
    return getattr(load(), name)(*args)
