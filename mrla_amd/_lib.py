"""ctypes binding of libmrla_hip.so (C ABI: include/mrla_hip.h).

The shared library is the product's only compute path for the MRLA operators: if it is missing the
import of any operator fails loudly with build instructions -- there is no Python/CPU fallback.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (MRLA_HIP_LIB: an experiment build of the same library -- scripts/build_variant.sh -- for A/B runs of whole programs)
LIB_PATH = os.environ.get("MRLA_HIP_LIB") or os.path.join(_HERE, "libmrla_hip.so")

ABI_VERSION = 5          # MRLA_ABI_VERSION of include/mrla_hip.h
OK, EINVAL, EUNSUPPORTED, EHIP = 0, -1, -2, -3
F32, BF16, F16 = 0, 1, 2
NCHW, NHWC = 0, 1
ACT_NONE, ACT_GELU = 0, 1
BN_NONE, BN_TRAIN, BN_EVAL = 0, 1, 2
FWD_MOMENTS, BWD_MOMENTS, TOKEN_PARTIALS, GEMM_MOMENTS = 8, 3, 15, 4

_ERR = {EINVAL: "invalid argument", EUNSUPPORTED: "unsupported shape/layout for the HIP kernels",
        EHIP: "HIP runtime error at kernel launch"}

_P, _I, _F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float

# name -> argument types (return type is always int).  Kept in one table so tests can check that every
# symbol declared in include/mrla_hip.h is exported and bound.
SIGNATURES = {
    "mrla_abi_version": [],
    "mrla_light_wgrad_rows": [_I] * 6,
    "mrla_light_stats_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "mrla_light_gate_fwd": [_P, _P, _P, _I, _P, _I, _I, _I, _I, _P],
    "mrla_light_bn_fwd": [_P, _P, _P, _P, _P, _P, _P, _I, _F, _F, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "mrla_light_apply_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "mrla_light_stats_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "mrla_light_bmom_splits": [_I] * 6,
    "mrla_light_mom_splits": [_I] * 6,
    "mrla_tuning_row_ranges": [_I],
    "mrla_light_bn_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "mrla_light_gate_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _P],
    "mrla_light_apply_bwd_pre_sums": [_I] * 6,
    "mrla_light_apply_bwd": [_P] * 15 + [_I] * 10 + [_P],
    "mrla_light_lean_supported": [_I] * 6,
    "mrla_light_stats_bwd_fused": [_P] * 8 + [_I] * 6 + [_P],
    "mrla_light_apply_bwd_fused": [_P] * 16 + [_I] * 8 + [_P],
    "mrla_light_pool_fused": [_P] * 6 + [_I] * 6 + [_P],
    "mrla_light_apply_fwd_fused": [_P] * 11 + [_I] * 8 + [_P],
    "mrla_light_stats_fwd_fused": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "mrla_base_tile_rows": [_I] * 6,
    "mrla_base_pmom_rows": [_I] * 6,
    "mrla_base_pool_value_fwd": [_P] * 8 + [_I] * 6 + [_P],
    "mrla_base_pmom_reduce": [_P, _P, _I, _I, _I, _I, _P],
    "mrla_base_dv_combine": [_P, _P, _P] + [_I] * 10 + [_P],
    "mrla_base_value_bwd_pre_sums": [_I] * 6,
    "mrla_base_value_bwd_dv": [_P] * 10 + [_I] * 7 + [_P],
    "mrla_base_gate_fwd": [_P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "mrla_base_attend_fwd": [_P, _P, _P, _P, _P, _P] + [_I] * 9 + [_P],
    "mrla_bn_stats_fwd": [_P, _P, _P, _P, _P, _P, _I, _F, _F, _P, _P, _P, _P, _I, _I, _I, _P],
    "mrla_bn_stats_fwd_rows": [_P, _P, _P, _P, _P, _I, _F, _F, _P, _P, _P, _P, _I, _I, _P],
    "mrla_base_tail_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "mrla_base_tail_stats_bwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "mrla_bn_stats_bwd": [_P, _P, _P, _P, _I, _I, _P, _P, _P, _I, _I, _I, _P],
    "mrla_base_attend_bwd": [_P] * 9 + [_I] * 8 + [_P],
    "mrla_base_gate_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P] + [_I] * 7 + [_P],
    "mrla_base_value_bwd": [_P] * 8 + [_I] * 11 + [_P],
    "mrla_token_norm_pool": [_P, _P, _P, _P, _F, _P, _P, _I, _I, _I, _I, _P],
    "mrla_token_apply_fwd": [_P] * 11 + [_I] * 6 + [_P],
    "mrla_token_part_rows": [_I] * 4,
    "mrla_token_apply_bwd": [_P] * 14 + [_I] * 5 + [_P],
    "mrla_token_gate_bwd": [_P] * 5 + [_I] + [_P] * 3 + [_I] * 5 + [_P],
    "mrla_token_ln_bwd": [_P] * 11 + [_I] * 5 + [_P],
    "mrla_token_base_supported": [_I] * 4,
    "mrla_token_base_value_fwd": [_P] * 6 + [_I] * 4 + [_P],
    "mrla_token_base_attend_fwd": [_P] * 8 + [_I] * 7 + [_P],
    "mrla_token_base_attend_bwd": [_P] * 4 + [_I] * 6 + [_P],
    "mrla_token_base_gate_bwd": [_P] * 8 + [_I] + [_P] * 3 + [_I] * 8 + [_P],
    "mrla_token_base_value_bwd": [_P] * 9 + [_I] * 4 + [_P],
    "mrla_bn_moment_rows": [_I] * 5,
    "mrla_bn_plane_moments": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "mrla_bn_act_fwd": [_P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P],
    "mrla_bn_plane_dmoments": [_P, _P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P],
    "mrla_bn_act_bwd": [_P, _P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P],
    "mrla_bn_pool_rows": [_I] * 6,
    "mrla_bn_relu_pool_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "mrla_bn_relu_pool_dmoments": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "mrla_bn_relu_pool_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "mrla_conv1x1_rows": [_I] * 4,
    "mrla_conv1x1_plan": [_I, _I, _I, _I, _I, _P],
    "mrla_conv1x1_wgrad_plan": [_I, _I, _I, _I, _P],
    "mrla_conv1x1_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "mrla_conv1x1_add_supported": [_I] * 4,
    "mrla_conv1x1_fwd_add": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "mrla_conv1x1_wgrad_rows": [_I] * 4,
    "mrla_conv1x1_wgrad": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "mrla_weight_bank_refresh": [_P, _I, _I, _P],
    "mrla_reduce_rows": [_P, _P, _I, _I, _P],
    "mrla_reduce_rows2": [_P, _P, _I, _I, _P, _P, _I, _I, _P],
    # sequence entry points (ABI 4): one call = the static launch sequence of a tail and direction
    "mrla_light_tail_fwd": [_P] * 6 + [_I] + [_P] * 6 + [_I, _F, _F] + [_P] * 6 + [_I] * 10 + [_P],
    "mrla_light_tail_bwd": [_P] * 5 + [_I] + [_P] * 7 + [_I] + [_P] * 5 + [_I] + [_P] * 8 + [_I] * 10 + [_P],
    "mrla_bn_fwd": [_P, _P, _I, _P, _P, _I, _P, _P, _P, _P, _I, _F, _F, _P, _I, _P] + [_I] * 6 + [_P],
    "mrla_bn_bwd": [_P] * 5 + [_I] * 4 + [_P, _P] + [_I] * 6 + [_P],
    "mrla_stem_fwd": [_P, _P, _P, _I, _P, _P, _P, _P, _I, _F, _F, _P, _P] + [_I] * 6 + [_P],
    "mrla_stem_bwd": [_P] * 5 + [_I, _I, _P, _P] + [_I] * 6 + [_P],
    "mrla_base_layer_fwd": [_P] * 6 + [_I] + [_P] * 5 + [_I, _F, _F] + [_P] * 9 + [_I] + [_P, _P] + [_I] * 9 + [_P],
    "mrla_base_layer_bwd": ([_P] * 5 + [_I] + [_P] * 6 + [_I] + [_P] * 6 + [_I] + [_P, _P] + [_I] + [_P] * 5 + [_I] + [_P] * 5
                            + [_I] * 12 + [_P]),
    "mrla_token_light_fwd": [_P] * 8 + [_I] + [_P, _P] + [_F] + [_P] * 4 + [_I] * 6 + [_P],
    "mrla_token_light_bwd": [_P] * 10 + [_I] + [_P] * 6 + [_I] + [_P] * 6 + [_I] * 6 + [_P],
}


class MrlaHipError(RuntimeError):
    pass


_lib = None


def load():
    """Load (once) and return the ctypes handle; raises MrlaHipError if the library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MrlaHipError(
            f"{LIB_PATH} not found: the MRLA HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` (or `make -C mrla_amd/csrc`) -- mrla_amd has no CPU/PyTorch fallback for its operators.")
    # torch FIRST: the library's DT_NEEDED libamdhip64.so.7 must resolve to the HIP runtime torch brought into the process.
    # Loaded before `import torch`, it pulls in /opt/rocm's copy instead, torch then binds to that one by SONAME, and launches
    # on torch's streams fail ("HIP runtime error at kernel launch": scripts/archive/r06_load_order_probe.py).
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch: fail loudly
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int
    if lib.mrla_abi_version() != ABI_VERSION:
        raise MrlaHipError(f"libmrla_hip.so reports ABI version {lib.mrla_abi_version()}, this binding is written against "
                           f"{ABI_VERSION} (MRLA_ABI_VERSION of include/mrla_hip.h); rebuild it")
    _lib = lib
    return lib


def check(rc, what):
    if rc != OK:
        raise MrlaHipError(f"{what}: {_ERR.get(rc, 'error')} (code {rc})")


def conv1x1_plan(m, k, n, addend=False):
    """(pixel blocks per workgroup, pipeline depth, workgroups, moment rows) of mrla_conv1x1_fwd[_add], or None."""
    out = (ctypes.c_int * 4)()
    rc = load().mrla_conv1x1_plan(m, k, n, BF16, int(addend), ctypes.cast(out, ctypes.c_void_p))
    return tuple(out) if rc == OK else None


def conv1x1_wgrad_plan(m, k, n):
    """(chunks per workgroup, LDS stages, tile n, tile k, splits, tiles) of mrla_conv1x1_wgrad, or None."""
    out = (ctypes.c_int * 6)()
    rc = load().mrla_conv1x1_wgrad_plan(m, k, n, BF16, ctypes.cast(out, ctypes.c_void_p))
    return tuple(out) if rc == OK else None


def call(name, *args):
    rc = getattr(load(), name)(*args)
    check(rc, name)
