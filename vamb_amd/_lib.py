"""ctypes binding of ``libvambhip.so`` (C ABI declared in ``include/vambhip.h``).

The shared library is built in-tree by ``__graft_entry__.build()`` / ``vamb_amd/csrc/build.py``.
There is NO fallback: if the library is missing or a call fails, the caller gets an exception.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (VAMBHIP_LIB_PATH: another build of the same ABI, for same-box A/B runs of two library builds -- tools/gpu)
LIB_PATH = os.environ.get("VAMBHIP_LIB_PATH") or os.path.join(_HERE, "libvambhip.so")
NBINS = 60
DENSITY_SCALE = 65536.0
HIST_SCALE = 256.0

VH_OK = 0
VH_ERR_INVALID = -1


class VambHipError(RuntimeError):
    pass


class ScanResult(ctypes.Structure):
    _fields_ = [("density_fx", ctypes.c_int64), ("hist_fx", ctypes.c_int64 * NBINS),
                ("n_within", ctypes.c_int64), ("n_lt", ctypes.c_int64)]


class ClusterInfo(ctypes.Structure):                                    # vh_cluster_info
    _fields_ = [("medoid", ctypes.c_int64), ("seed", ctypes.c_int64), ("n_members", ctypes.c_int64),
                ("kind", ctypes.c_int32), ("pad_", ctypes.c_int32), ("maximal_pvr", ctypes.c_double),
                ("observed_pvr", ctypes.c_double), ("radius", ctypes.c_double), ("successes", ctypes.c_int64),
                ("attempts", ctypes.c_int64), ("pvr_after", ctypes.c_double), ("successes_after", ctypes.c_int64),
                ("attempts_after", ctypes.c_int64), ("order_index_after", ctypes.c_int64)]


_lib = None

_i64 = ctypes.c_int64
_vp = ctypes.c_void_p
_int = ctypes.c_int
_f32 = ctypes.c_float
_pp = ctypes.POINTER(ctypes.c_void_p)

# name -> (restype, argtypes).  Must list every symbol declared in include/vambhip.h
# (tests/test_lib_abi.py cross-checks the header against this table and the .so).
SIGNATURES = {
    "vh_last_error": (ctypes.c_char_p, []),
    "vh_version": (ctypes.c_char_p, []),
    "vh_set_option": (_int, [ctypes.c_char_p, _i64]),
    "vh_unset_option": (_int, [ctypes.c_char_p]),
    "vh_get_option": (_int, [ctypes.c_char_p, ctypes.POINTER(_i64)]),
    "vh_set_option_string": (_int, [ctypes.c_char_p, ctypes.c_char_p]),
    "vh_selftest": (_int, [ctypes.POINTER(_int)]),
    "vh_device_count": (_int, [ctypes.POINTER(_int)]),
    "vh_set_device": (_int, [_int]),
    "vh_clu_create": (_int, [_vp, _vp, _i64, _int, _int, _vp, _pp]),
    "vh_clu_destroy": (_int, [_vp]),
    "vh_clu_rows": (_int, [_vp, ctypes.POINTER(_i64), ctypes.POINTER(_i64)]),
    "vh_clu_max_medoids": (_int, [_vp, ctypes.POINTER(_int)]),
    "vh_clu_scan": (_int, [_vp, _int, _vp, _vp, _vp]),
    "vh_clu_attach_comm": (_int, [_vp, _vp]),
    "vh_clu_scan_sharded": (_int, [_vp, _int, _vp, _vp]),
    "vh_clu_select_sharded": (_int, [_vp, _i64, _f32, _int, _vp, _vp, _i64, ctypes.POINTER(_i64)]),
    "vh_clu_scan_seq": (_int, [_vp, ctypes.POINTER(_i64)]),
    "vh_clu_scan_list": (_int, [_vp, _i64, _int, _vp, _i64, ctypes.POINTER(_i64)]),
    "vh_gen_create": (_int, [_vp, _vp, _i64, _int, _int, _int, ctypes.c_uint64, ctypes.c_double, _i64,
                             ctypes.POINTER(_vp)]),
    "vh_gen_create_sharded": (_int, [_vp, _vp, _i64, _int, _int, _int, ctypes.c_uint64, ctypes.c_double, _i64,
                                     ctypes.POINTER(_vp)]),
    "vh_gen_destroy": (_int, [_vp]),
    "vh_gen_next": (_int, [_vp, ctypes.POINTER(ClusterInfo), _vp, _i64]),
    "vh_gen_next_batch": (_int, [_vp, _int, ctypes.POINTER(ClusterInfo), _vp, _i64, ctypes.POINTER(_int)]),
    "vh_debug_find_threshold": (_int, [_vp, _i64, ctypes.c_double, ctypes.POINTER(_int), ctypes.POINTER(ctypes.c_double),
                                       ctypes.POINTER(ctypes.c_double)]),
    "vh_debug_pyrandom_sample": (_int, [ctypes.c_uint64, _int, _vp, _vp, _vp]),
    "vh_gen_state": (_int, [_vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_i64), ctypes.POINTER(_i64),
                            ctypes.POINTER(_i64)]),
    "vh_gen_live_rows": (_int, [_vp, ctypes.POINTER(_i64)]),
    "vh_gen_counters": (_int, [_vp, ctypes.POINTER(_i64), ctypes.POINTER(_i64), ctypes.POINTER(_i64),
                               ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_i64), ctypes.POINTER(_i64)]),
    "vh_clu_select": (_int, [_vp, _i64, _vp, _f32, _int, _vp, _i64, ctypes.POINTER(_i64)]),
    "vh_clu_remove": (_int, [_vp, _vp, _i64]),
    "vh_clu_pack": (_int, [_vp, ctypes.POINTER(_i64)]),
    "vh_clu_get_rows": (_int, [_vp, _vp, _i64, _vp]),
    "vh_clu_get_kept": (_int, [_vp, _vp]),
    "vh_clu_last_kernel_ms": (_int, [_vp, ctypes.POINTER(_f32)]),
    "vh_clu_set_timing": (_int, [_vp, _int]),
    "vh_vae_create": (_int, [_vp, _pp]),
    "vh_vae_destroy": (_int, [_vp]),
    "vh_vae_param_size": (_int, [_vp, ctypes.c_char_p, ctypes.POINTER(_i64)]),
    "vh_vae_set_param": (_int, [_vp, ctypes.c_char_p, _vp, _i64]),
    "vh_vae_get_param": (_int, [_vp, ctypes.c_char_p, _vp, _i64]),
    "vh_vae_get_grad": (_int, [_vp, ctypes.c_char_p, _vp, _i64]),
    "vh_vae_set_dataset": (_int, [_vp, _vp, _vp, _vp, _vp, _i64]),
    "vh_vae_train_step": (_int, [_vp, _vp, _i64, _vp, _vp, ctypes.POINTER(ctypes.c_double)]),
    "vh_vae_train_epoch": (_int, [_vp, _vp, _i64, _i64, ctypes.POINTER(ctypes.c_double)]),
    "vh_vae_forward": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vh_vae_encode": (_int, [_vp, _vp]),
    "vh_vae_opt_state": (_int, [_vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                                ctypes.POINTER(_i64)]),
    "vh_vae_opt_set_state": (_int, [_vp, ctypes.c_double, ctypes.c_double, _i64]),
    "vh_vae_reset_optimizer": (_int, [_vp]),
    "vh_vae_get_opt_moment": (_int, [_vp, ctypes.c_char_p, _int, _vp, _i64]),
    "vh_vae_set_opt_moment": (_int, [_vp, ctypes.c_char_p, _int, _vp, _i64]),
    "vh_dataset_create": (_int, [_vp, _vp, _vp, _vp, _i64, _int, ctypes.POINTER(_vp)]),
    "vh_dataset_destroy": (_int, [_vp]),
    "vh_vae_use_dataset": (_int, [_vp, _vp]),
    "vh_tnf_create": (_int, [_vp, _pp]),
    "vh_tnf_destroy": (_int, [_vp]),
    "vh_tnf_kmercounts": (_int, [_vp, _vp, _vp, _i64, _vp]),
    "vh_tnf_project": (_int, [_vp, _vp, _i64, _int, _vp]),
    "vh_prep_create": (_int, [_i64, _int, _pp]),
    "vh_prep_destroy": (_int, [_vp]),
    "vh_prep_upload": (_int, [_vp, _vp, _vp]),
    "vh_prep_column_sums": (_int, [_vp, _int, _vp, _vp]),
    "vh_prep_normalise_rows": (_int, [_vp, _vp, _vp, _int, _f32, _vp]),
    "vh_prep_zscore_tnf": (_int, [_vp, _vp, _vp]),
    "vh_prep_finish": (_int, [_vp, _vp, _vp, _pp]),
    "vh_dataset_shape": (_int, [_vp, ctypes.POINTER(_i64), ctypes.POINTER(_int)]),
    "vh_dataset_download": (_int, [_vp, _vp, _vp, _vp, _vp]),
    "vh_vae_train_epochs": (_int, [_vp, _i64, _i64, _i64, _i64, _vp]),
    "vh_vae_get_hidden": (_int, [_vp, _int, _vp, _i64]),
    "vh_vae_set_precision": (_int, [_vp, _int]),
    "vh_vae_set_probe": (_int, [_vp, _int, _int]),
    "vh_vae_probe_result": (_int, [_vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_i64),
                                   ctypes.POINTER(ctypes.c_double)]),
    "vh_comm_unique_id": (_int, [_vp]),
    "vh_comm_create": (_int, [_int, _int, _vp, _pp]),
    "vh_comm_destroy": (_int, [_vp]),
    "vh_comm_create_host": (_int, [_int, _int, _vp, _vp, _vp, _pp]),
    "vh_comm_info": (_int, [_vp, ctypes.POINTER(_int), ctypes.POINTER(_int), ctypes.POINTER(_int), ctypes.POINTER(_int)]),
    "vh_device_synchronize": (_int, []),
    "vh_vae_attach_comm": (_int, [_vp, _vp]),
    "vh_vae_set_syncbn": (_int, [_vp, _int]),
    "vh_vae_train_epoch_dp": (_int, [_vp, _vp, _i64, _i64, _i64, _vp, ctypes.POINTER(ctypes.c_double)]),
    "vh_debug_gemm": (_int, [_int, _int, _int, _vp, _vp, _vp, _vp, _int, _int, _int, _int,
                             ctypes.POINTER(_f32)]),
    "vh_debug_gemm16": (_int, [_int, _vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _int,
                               ctypes.POINTER(_f32)]),
    "vh_debug_gemm16_timeline": (_int, [_int, _int, _int, _int, _int, _vp, _int, _vp, _vp]),
    "vh_debug_scan_timeline": (_int, [_vp, _vp, _int, _vp, _vp]),
    "vh_debug_check_guards": (_int, [ctypes.c_char_p, _int, _vp]),
    "vh_vae_create_labelled": (_int, [_vp, _vp, _pp]),
    "vh_vae_set_optimizer": (_int, [_vp, _int, _f32]),
    "vh_vae_row_width": (_int, [_vp, ctypes.POINTER(ctypes.c_int32)]),
    "vh_vae_forward_rows": (_int, [_vp, _vp, _i64, _int, _vp, _vp, _vp, _vp]),
    "vh_vae_label_stats": (_int, [_vp, _i64, _vp]),
    "vh_dataset_set_labels": (_int, [_vp, _vp, _i64, ctypes.c_int32]),
    "vh_dataset_create_labels": (_int, [_vp, _i64, ctypes.c_int32, ctypes.POINTER(_vp)]),
    "vh_debug_gemm16_tn": (_int, [_vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _int, _int, _vp]),
    "vh_vae_set_hierarchy": (_int, [_vp, _vp, ctypes.c_int32]),
    "vh_vaevae_create": (_int, [_vp, _vp, _vp, _pp]),
    "vh_vaevae_destroy": (_int, [_vp]),
    "vh_vaevae_set_datasets": (_int, [_vp, _vp, _vp, _vp]),
    "vh_vaevae_train_step": (_int, [_vp, _vp, _i64, _vp, _vp, _vp]),
    "vh_vaevae_train_epoch": (_int, [_vp, _vp, _i64, _i64, _vp]),
    "vh_vaevae_get_grad": (_int, [_vp, _int, ctypes.c_char_p, _vp, _i64]),
}


def load():
    """Load libvambhip.so and type every entry point.  Raises if the library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VambHipError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ "
            "as g; g.build()'` (or `python vamb_amd/csrc/build.py`). There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here means the .so is stale
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


# Environment variable -> (library option, how the value is read).  The C library reads no environment: every
# constructor of a handle (VAE, scan backend, cluster generator) calls sync_env_options() first, so the variables
# keep working for A/B runs from the shell and for monkeypatch.setenv in the tests.
_ENV_OPTIONS = {
    "VAMBHIP_SCAN_LC": ("scan.column_loop", int),
    "VAMBHIP_SCAN_MFMA": ("scan.mfma", int),
    "VAMBHIP_SCAN_MFMA_ROWMAJOR": ("scan.mfma_rowmajor", int),
    "VAMBHIP_REFERENCE_ORDER": ("scan.reference_order", int),
    "VAMBHIP_SCAN_DBG": ("scan.debug", int),
    "VAMBHIP_SCAN_PUBLISH_SPLIT": ("scan.publish_split", int),
    "VAMBHIP_GEN_PROFILE": ("gen.profile", lambda v: 1),
    "VAMBHIP_NO_SPECULATION": ("gen.speculate", lambda v: 0),
    "VAMBHIP_SPEC_WINDOW": ("gen.spec_window", int),
    "VAMBHIP_GEN_PREFILL": ("gen.prefill", int),
    "VAMBHIP_GEN_INLINE_REMOVALS": ("gen.inline_removals", int),
    "VAMBHIP_GATHER_STAGE_BYTES": ("gen.gather_stage_bytes", int),
    "VAMBHIP_BIG_TILES": ("vae.big_tiles", int),
    "VAMBHIP_XCD_REMAP": ("vae.xcd_remap", int),
    "VAMBHIP_DW_WGS": ("vae.dw_workgroups", int),
    "VAMBHIP_SINGLE_STREAM": ("vae.single_stream", lambda v: 1),
    "VAMBHIP_FORK_EVENTS": ("vae.fork_events", lambda v: 1),
    "VAMBHIP_DEBUG_TIMING": ("vae.debug_timing", lambda v: 1),
    "VAMBHIP_VAE_FUSED_SKINNY": ("vae.fused_skinny", int),
    "VAMBHIP_VAE_FUSED_FINALIZE": ("vae.fused_finalize", int),
    "VAMBHIP_VAE_PREFETCH_BATCH": ("vae.prefetch_batch", int),
    "VAMBHIP_VAE_FORK_PLAN": ("vae.fork_plan", int),
    "VAMBHIP_VAE_PROBE_EVERY": ("vae.probe_every", int),
    "VAMBHIP_VAE_LOSS_FROM_DATASET": ("vae.loss_from_dataset", int),
    "VAMBHIP_VAE_PREFETCH_MAX_COLS": ("vae.prefetch_max_cols", int),
    "VAMBHIP_VAE_GEMM_PREFETCH": ("vae.gemm_prefetch", int),
    "VAMBHIP_VAE_GEMM_KGROUPS": ("vae.gemm_kgroups", int),
    "VAMBHIP_VAE_DW_PAIR": ("vae.dw_pair", int),
    "VAMBHIP_VAE_LOSS_REGISTERS": ("vae.loss_registers", int),
    "VAMBHIP_DEBUG_GUARD_BYTES": ("debug.guard_bytes", int),
}
_ENV_STRING_OPTIONS = {"VAMBHIP_RCCL": "comm.rccl_library", "ROCM_PATH": "comm.rocm_path"}
_explicit_options: dict = {}


def set_option(name: str, value) -> None:
    """Set a library option (include/vambhip.h lists them); it wins over the environment variables."""
    lib = load()
    if isinstance(value, str):
        check(lib.vh_set_option_string(name.encode(), value.encode()))
    else:
        check(lib.vh_set_option(name.encode(), int(value)))
    _explicit_options[name] = value


def get_option(name: str, default: int = 0) -> int:
    v = _i64(default)
    check(load().vh_get_option(name.encode(), ctypes.byref(v)))
    return v.value


def sync_env_options() -> None:
    lib = load()
    for var, (name, conv) in _ENV_OPTIONS.items():
        if name in _explicit_options:
            continue
        raw = os.environ.get(var)
        if raw is None or raw == "":
            check(lib.vh_unset_option(name.encode()))
        else:
            check(lib.vh_set_option(name.encode(), int(conv(raw))))
    for var, name in _ENV_STRING_OPTIONS.items():
        if name in _explicit_options:
            continue
        raw = os.environ.get(var)
        if raw:
            check(lib.vh_set_option_string(name.encode(), raw.encode()))


def check(status: int):
    if status == VH_OK:
        return
    msg = load().vh_last_error().decode("utf-8", "replace")
    if status == VH_ERR_INVALID:
        raise ValueError(msg)
    raise VambHipError(f"libvambhip status {status}: {msg}")


def device_count() -> int:
    n = _int(0)
    check(load().vh_device_count(ctypes.byref(n)))
    return n.value


_selftest_done = False
selftest_fallbacks = 0   # bit 0: row-major scan kernel switched off, bit 1: deep-prefetch / K-group GEMM tiles switched off


def require_gpu():
    """Fail loudly when no MI355X is visible (never silently fall back).  The first call of a process also runs the library's
    start-up self-test (``vh_selftest``, csrc/selftest.hip: the hand-scheduled kernels beside their compiler-scheduled twins;
    VAMBHIP_SELFTEST=0 skips it); options it had to switch off stay off for the process."""
    global _selftest_done, selftest_fallbacks
    n = device_count()
    if n < 1:
        raise VambHipError("no HIP device visible")
    if not _selftest_done:
        _selftest_done = True
        if os.environ.get("VAMBHIP_SELFTEST", "1") != "0":
            sync_env_options()
            mask = _int(0)
            check(load().vh_selftest(ctypes.byref(mask)))
            selftest_fallbacks = mask.value
            if mask.value & 1:   # (an entry in _explicit_options: sync_env_options leaves the option as the self-test set it)
                _explicit_options["scan.mfma_rowmajor"] = 0
            if mask.value & 2:
                _explicit_options["vae.gemm_prefetch"] = 1
                _explicit_options["vae.gemm_kgroups"] = 1
    return n


def ptr(a):
    if a is None:
        return None
    assert isinstance(a, np.ndarray) and a.flags.c_contiguous
    return a.ctypes.data_as(ctypes.c_void_p)
