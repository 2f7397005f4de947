"""ctypes binding of the C-ABI library (include/openibl_amd.h).

There is no CPU fallback: if the library cannot be loaded the import of the product path fails
with an explicit error, and every call that returns a non-zero status raises.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

from . import build as _build


c_void_p, c_int, c_size_t, c_char_p = C.c_void_p, C.c_int, C.c_size_t, C.c_char_p

# name -> (restype, argtypes); mirrors include/openibl_amd.h one to one.
SIGNATURES = {
    "oibl_abi_version": (c_int, []),
    "oibl_last_error": (c_char_p, []),
    "oibl_target_arch": (c_char_p, []),
    "oibl_elem_size": (c_size_t, [c_int]),
    "oibl_cast_f32_to_bf16": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "oibl_cast_bf16_to_f32": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "oibl_conv3x3_packed_bytes": (c_size_t, [c_int, c_int, c_int]),
    "oibl_pack_conv3x3_weights": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "oibl_conv3x3_nhwc": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                  c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "oibl_conv3x3_nhwc_flagged": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                          c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "oibl_conv3x3_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "oibl_conv3x3_nhwc_ws": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                     c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "oibl_conv1_1_nchw": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int,
                                  c_void_p, c_void_p]),
    "oibl_global_maxpool_nhwc": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "oibl_nhwc_to_nchw_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "oibl_nchw_f32_to_nhwc": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "oibl_vgg16_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "oibl_vgg16_conv5_forward": (c_int, [c_void_p, c_int, c_int, c_int, C.POINTER(c_void_p),
                                         C.POINTER(c_void_p), c_int, c_void_p, c_void_p,
                                         c_size_t, c_void_p]),
    "oibl_vgg16_conv5_forward_ev": (c_int, [c_void_p, c_int, c_int, c_int, C.POINTER(c_void_p),
                                            C.POINTER(c_void_p), c_int, c_void_p, c_void_p,
                                            c_size_t, c_void_p, c_void_p, c_void_p]),
    "oibl_vgg16_u8_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "oibl_vgg16_conv5_forward_u8": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p,
                                            C.POINTER(c_void_p), C.POINTER(c_void_p), c_int, c_void_p,
                                            c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    "oibl_vgg16_stem_bf16": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_void_p]),
    "oibl_vgg16_stem_x3": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p]),
    "oibl_vgg16_stem_mx": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p]),
    "oibl_netvlad_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "oibl_netvlad_forward": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                     c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_size_t,
                                     c_void_p]),
    "oibl_pca_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "oibl_pca_forward": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int,
                                 c_void_p, c_void_p, c_size_t, c_void_p]),
    "oibl_pca_pack_weight": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "oibl_pca_packed_supported": (c_int, [c_int, c_int, c_int]),
    "oibl_pca_forward_packed": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int,
                                        c_void_p, c_void_p, c_size_t, c_void_p]),
    "oibl_l2_normalize_rows": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "oibl_sum_l2_normalize": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "oibl_pairwise_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "oibl_pairwise_sqdist": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p,
                                     c_size_t, c_void_p, c_size_t, c_void_p]),
    "oibl_sqdist_topk_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "oibl_sqdist_topk": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                 c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "oibl_pairwise_st_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "oibl_pairwise_sqdist_st": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int,
                                        c_void_p, c_size_t, c_void_p, c_size_t, c_void_p]),
    "oibl_sqdist_topk_st_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "oibl_sqdist_topk_st": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                    c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                    c_void_p]),
    "oibl_x3_split_rows": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "oibl_x3_join_rows": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "oibl_mx_split_rows": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "oibl_mx_split_rows_flagged": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_void_p, c_void_p]),
    "oibl_mx_join_rows": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_int, c_void_p]),
    "oibl_match_operand_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "oibl_match_prepare": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "oibl_sqdist_topk_prepared_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "oibl_sqdist_topk_prepared": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int,
                                          c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_size_t, c_void_p]),
    "oibl_match_prepare_f16r": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "oibl_sqdist_topk_f16r_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "oibl_f16r_members": (c_int, [c_int]),
    "oibl_f16r_fused": (c_int, [c_int, c_int, c_int, c_int]),
    "oibl_f16r_filter_select": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                        c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                        c_void_p]),
    "oibl_f16r_keep_members": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                       c_int, c_void_p]),
    "oibl_f16r_rescore": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p,
                                  c_void_p, c_void_p, c_void_p]),
    "oibl_sqdist_topk_f16r": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                      c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                      c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_size_t, c_void_p]),
    "oibl_match_prepare_f16r_st": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "oibl_sqdist_topk_f16r_st_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "oibl_sqdist_topk_f16r_st": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                         c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_size_t, c_void_p]),
    "oibl_f16r_rescore_st": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int,
                                     c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "oibl_cast_f32_to_f16": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "oibl_cast_f16_to_f32": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "oibl_resize_bilinear_nchw": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int,
                                          c_void_p]),
    "oibl_row_topk": (c_int, [c_void_p, c_void_p, c_int, c_int, c_size_t, c_int, c_int, c_void_p,
                              c_void_p, c_void_p]),
    "oibl_row_argsort_workspace_bytes": (c_size_t, [c_int, c_int]),
    "oibl_row_argsort": (c_int, [c_void_p, c_int, c_int, c_size_t, c_void_p, c_void_p, c_void_p, c_size_t,
                                 c_void_p]),
    "oibl_cluster_means": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "oibl_first_hit_rank": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                    c_void_p, c_void_p]),
    "oibl_gemm_nt": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_size_t,
                             c_void_p]),
    "oibl_event_elapsed_ms": (c_int, [c_void_p, c_void_p, c_void_p]),
    "oibl_copy_words": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
}
# test hooks: exported ONLY by libopenibl_amd_dbg.so (the same sources compiled with -DOIBL_DEBUG_HOOKS);
# the value after the signature is the default the hook is reset to between tests (None: no reset)
_HOOKS = {"oibl_debug_set_regstage": (c_int, [c_int]),
          "oibl_debug_set_conv11_valu": (c_int, [c_int]),
          "oibl_debug_set_conv_tile": (c_int, [c_int]),
          "oibl_debug_set_conv_ablate": (c_int, [c_int]),
          "oibl_debug_set_conv_c64": (c_int, [c_int]),
          "oibl_debug_set_stem_fused": (c_int, [c_int]),
          "oibl_debug_mfma_peak": (c_int, [C.c_long, c_int, c_void_p, c_void_p]),
          "oibl_debug_set_match_ring": (c_int, [c_int]),
          "oibl_debug_set_ring_ablate": (c_int, [c_int]),
          "oibl_debug_set_ring_raster": (c_int, [c_int]),
          "oibl_debug_set_conv_korder": (c_int, [c_int]),
          "oibl_debug_set_mx_variant": (c_int, [c_int]),
          "oibl_debug_set_conv_splitk": (c_int, [c_int]),
          "oibl_debug_set_stem3_prio": (c_int, [c_int]),
          "oibl_debug_set_match_group": (c_int, [c_int]),
          "oibl_debug_set_match_splitk": (c_int, [c_int]),
          "oibl_debug_set_pca_small": (c_int, [c_int]),
          "oibl_debug_set_pca_stream": (c_int, [c_int]),
          "oibl_debug_set_mx_act_shift": (c_int, [c_int]),
          "oibl_debug_set_netvlad_slabs": (c_int, [c_int]),
          "oibl_debug_set_ring_bar1": (c_int, [c_int]),
          "oibl_debug_set_ring_stagger": (c_int, [c_int]),
          "oibl_debug_set_mx_splitk": (c_int, [c_int]),
          "oibl_debug_set_stem_u8": (c_int, [c_int]),
          "oibl_debug_set_match_bar1": (c_int, [c_int]),
          "oibl_debug_set_match_mx_early": (c_int, [c_int]),
          "oibl_debug_set_prof_buffer": (c_int, [c_void_p])}

ABI_VERSION = 3


class OpenIBLAmdError(RuntimeError):
    pass


def lib_path() -> Path:
    return _build.LIB_PATH


def debug_lib_path() -> Path:
    return _build.LIB_PATH_DBG


_LIBS = {"product": None, "debug": None}
_ACTIVE = "product"
_HOOK_DEFAULTS = {"oibl_debug_set_regstage": 0, "oibl_debug_set_conv11_valu": 0, "oibl_debug_set_conv_tile": 0,
                  "oibl_debug_set_conv_ablate": 0, "oibl_debug_set_conv_c64": 1, "oibl_debug_set_stem_fused": 1,
                  "oibl_debug_set_match_ring": 1, "oibl_debug_set_ring_ablate": 0, "oibl_debug_set_ring_raster": 0,
                  "oibl_debug_set_conv_korder": -1, "oibl_debug_set_mx_variant": 0, "oibl_debug_set_conv_splitk": 1,
                  "oibl_debug_set_stem3_prio": 0, "oibl_debug_set_match_group": 4, "oibl_debug_set_match_splitk": 1,
                  "oibl_debug_set_pca_small": 1, "oibl_debug_set_pca_stream": 1, "oibl_debug_set_mx_act_shift": 3, "oibl_debug_set_netvlad_slabs": 1,
                  "oibl_debug_set_ring_bar1": 1, "oibl_debug_set_ring_stagger": 0, "oibl_debug_set_match_bar1": 1, "oibl_debug_set_match_mx_early": 1, "oibl_debug_set_mx_splitk": 1, "oibl_debug_set_stem_u8": 1,
                  "oibl_debug_set_prof_buffer": None}


def _open(which: str):
    path = lib_path() if which == "product" else debug_lib_path()
    if os.environ.get("OPENIBL_AMD_NO_BUILD", "0") != "1":
        try:
            if not _build.is_current():
                _build.build(verbose=False)
        except Exception as e:  # a stale-but-present library is still usable — but say so
            if not path.exists():
                raise OpenIBLAmdError(
                    f"openibl_amd: the HIP extension is not built and cannot be built here: {e}")
            import warnings
            warnings.warn(f"openibl_amd: {path.name} is older than csrc/ and could not be rebuilt "
                          f"({e}); loading the stale library")
    if not path.exists():
        raise OpenIBLAmdError(
            f"openibl_amd: {path} is missing; run `python -m openibl_amd.build` (needs hipcc). "
            "There is no CPU fallback for this path.")
    try:
        lib = C.CDLL(str(path))
    except OSError as e:
        raise OpenIBLAmdError(f"openibl_amd: cannot load {path}: {e}")
    table = dict(SIGNATURES)
    if which == "debug":
        table.update(_HOOKS)
    for name, (res, args) in table.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise OpenIBLAmdError(f"openibl_amd: {path} does not export {name}")
        fn.restype = res
        fn.argtypes = args
    if lib.oibl_abi_version() != ABI_VERSION:
        raise OpenIBLAmdError(
            f"openibl_amd: ABI version {lib.oibl_abi_version()} != expected {ABI_VERSION}")
    return lib


def load():
    """The library every call goes through: libopenibl_amd.so — the product, no test hooks, no mutable
    process state — unless debug_hooks() switched this process to libopenibl_amd_dbg.so (same kernels,
    plus the oibl_debug_* hooks).  Builds first if the in-tree libraries are missing or stale and hipcc is
    available."""
    if _LIBS[_ACTIVE] is None:
        _LIBS[_ACTIVE] = _open(_ACTIVE)
    return _LIBS[_ACTIVE]


def debug_hooks():
    """Switch this process to the debug-hook library and return its handle (test infrastructure: variant
    tests, tests/gpu_* diagnostics).  Sticky until use_product_library()."""
    global _ACTIVE
    _ACTIVE = "debug"
    return load()


def use_product_library(reset_hooks: bool = True) -> None:
    """Back to the product library; the hooks of an already loaded debug library return to their defaults."""
    global _ACTIVE
    if reset_hooks and _LIBS["debug"] is not None:
        for name, dflt in _HOOK_DEFAULTS.items():
            getattr(_LIBS["debug"], name)(dflt)
    _ACTIVE = "product"


def active_library() -> str:
    return _ACTIVE


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().oibl_last_error().decode(errors="replace")
        raise OpenIBLAmdError(f"openibl_amd {what} failed (status {rc}): {msg}")
