"""ctypes binding of libocn_hip.so (include/ocn_hip.h).

The product path has NO fallback: if the shared library is missing or an entry returns a
non-zero status, this raises.  ``build()`` compiles it in-tree with hipcc for gfx950.
"""
from __future__ import annotations

import ctypes
import glob
import os
import subprocess
from ctypes import c_float, c_int32, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
LIB_PATH = os.environ.get("OCN_LIB_PATH", os.path.join(_HERE, "libocn_hip.so"))
CSRC = os.path.join(_HERE, "csrc")
INCLUDE = os.path.join(_ROOT, "include")

# name -> (restype, argtypes); must list every symbol include/ocn_hip.h declares
_P = c_void_p
ABI_VERSION = 9
SIGNATURES = {
    "ocn_abi_version": (c_int32, []),
    "ocn_scan_workspace_bytes": (c_int64, [c_int64]),
    "ocn_zero_regions": (c_int32, [_P, _P, c_int32, _P]),
    "ocn_check_edges": (c_int32, [_P, _P, c_int64, c_int64, c_int64, _P, _P]),
    "ocn_edge_offsets": (c_int32, [_P, _P, c_int64, _P, _P, _P]),
    "ocn_class_order": (c_int32, [_P, _P, _P, c_int64, _P, _P, _P, _P, _P, _P]),
    "ocn_scan_i32": (c_int32, [_P, c_int64, _P, _P, _P]),
    "ocn_order_workspace_bytes": (c_int64, [c_int64]),
    "ocn_order_by_node": (c_int32, [_P, c_int64, c_int64, _P, _P, _P]),
    "ocn_order_by_node_finish": (c_int32, [_P, c_int64, c_int64, _P, _P, _P]),
    "ocn_batch_prep": (c_int32, [_P, _P, c_int64, _P, _P, c_int64, _P, _P, _P, c_int32, _P]),
    "ocn_cn_flags": (c_int32, [_P, _P, _P, _P, _P, _P, _P, c_int64, _P, c_int64, _P, _P, _P, c_int64, c_int64, _P, _P, c_int64, _P,
                               _P, _P, _P, _P, _P, _P]),
    "ocn_chunk_offsets": (c_int32, [_P, _P, _P, _P, c_int64, _P, _P, _P]),
    "ocn_walk_chunk": (c_int32, []),
    "ocn_cn_walk_flags": (c_int32, [_P, _P, _P, _P, _P, _P, c_int64, _P, _P, _P, c_int64, _P, _P, c_int64, _P, _P, _P,
                                    _P, _P]),
    "ocn_neighbor_degree_sum": (c_int32, [_P, _P, c_int64, _P, _P]),
    "ocn_walk_prep_max_batch": (c_int32, []),
    "ocn_cn_flags_small_graph_cols": (c_int32, []),
    "ocn_walk_prep": (c_int32, [_P, _P, _P, _P, c_int64, c_int32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "ocn_cn_walk_group": (c_int32, [_P, _P, _P, _P, _P, c_int64, _P, _P, _P, _P, _P, _P, _P, c_int64, _P, _P, _P, _P]),
    "ocn_walk_rev_offsets": (c_int32, [_P, _P, _P, _P, _P, c_int64, _P, _P, _P]),
    "ocn_cn_weights_cn5": (c_int32, [_P, c_int64, _P, _P, c_int32, _P, _P]),
    "ocn_cn5_column_stats": (c_int32, [_P, c_int64, _P, _P]),
    "ocn_cn_colsum_workspace_bytes": (c_int64, [c_int64, c_int64]),
    "ocn_cn_colsum_exact": (c_int32, [_P, _P, _P, c_int64, _P, _P, _P, _P, c_int64, _P, c_int64, _P, _P, _P, _P, _P, _P, _P]),
    "ocn_cn_weights_cn7": (c_int32, [_P, c_int64, c_float, _P, _P, _P]),
    "ocn_cn_gather": (c_int32, [_P, _P, _P, _P, _P, c_int64, _P, _P, _P, _P, _P, c_int32, c_int64, _P, _P, _P, _P, _P,
                                _P, _P, _P, _P, _P]),
    "ocn_gather_schedule": (c_int32, [_P, c_int64, c_int64, _P, _P]),
    "ocn_coo_to_csr_workspace_bytes": (c_int64, [c_int64, c_int64, c_int32, c_int32]),
    "ocn_coo_to_csr": (c_int32, [_P, _P, c_int64, c_int64, c_int64, c_int32, c_int32, _P, _P, _P, _P, _P]),
    "ocn_wgrad_workspace_bytes": (c_int64, [c_int64, c_int32, c_int32]),
    "ocn_wgrad": (c_int32, [_P, c_int64, _P, c_int64, c_int64, c_int32, c_int32, _P, _P, _P, _P]),
    "ocn_cn_weights_cn6": (c_int32, [_P, _P, c_int64, _P, _P, _P, _P, _P, _P]),
    "ocn_cn_gather3": (c_int32, [_P, _P, _P, _P, _P, c_int64, _P, _P, _P, _P, _P, _P, _P, c_int32, _P, _P, _P, _P, _P]),
    "ocn_cn_gather3_backward": (c_int32, [_P, _P, _P, _P, _P, c_int64, _P, _P, _P, _P, _P, _P, _P, c_int32, _P, _P, _P, _P, _P, _P]),
    "ocn_cn_gather_backward": (c_int32, [_P, _P, _P, _P, _P, c_int64, _P, _P, _P, _P, _P, c_int32, _P, _P, _P, _P, _P]),
    "ocn_cn_gather_backward_det_workspace_bytes": (c_int64, [c_int64, c_int64, c_int64]),
    "ocn_cn_gather_backward_det": (c_int32, [_P, _P, _P, _P, c_int64, _P, _P, _P, c_int64, _P, _P, c_int64, c_int32, _P, _P, _P, _P,
                                             _P, _P]),
    "ocn_cn_gather_backward_det_keys_offset": (c_int64, [c_int64]),
    "ocn_cn_gather_backward_det_lists": (c_int32, [_P, _P, _P, _P, c_int64, _P, _P, c_int64, c_int64, _P, _P]),
    "ocn_ln_drop_relu_workspace_bytes": (c_int64, [c_int32]),
    "ocn_ln_drop_relu_forward": (c_int32, [_P, _P, _P, c_float, c_float, ctypes.c_uint64, c_int32, c_int64, c_int32, _P, _P, _P]),
    "ocn_ln_drop_relu_backward": (c_int32, [_P, _P, _P, _P, _P, c_float, ctypes.c_uint64, c_int32, c_int64, c_int32, _P, _P, _P, _P, _P]),
    "ocn_dropout_keep_mask": (c_int32, [ctypes.c_uint64, c_float, c_int64, _P, _P]),
    "ocn_mix3_workspace_bytes": (c_int64, []),
    "ocn_mix3_backward": (c_int32, [_P, _P, _P, _P, _P, c_int64, _P, _P, _P, _P, _P, _P]),
    "ocn_spmm_csr": (c_int32, [_P, _P, _P, c_int64, _P, c_int32, _P, _P, c_int32, c_int32, c_int32, _P, _P]),
    "ocn_deg_rsqrt": (c_int32, [_P, _P, c_int64, c_float, _P, _P]),
    "ocn_spgemm_max_cols": (c_int64, []),
    "ocn_spgemm_pattern_count": (c_int32, [_P, _P, c_int64, _P, _P, c_int64, _P, _P, c_int64, _P]),
    "ocn_spgemm_pattern_fill": (c_int32, [_P, _P, c_int64, _P, _P, c_int64, _P, _P, _P]),
    "ocn_spgemm_bit_rows": (c_int32, [_P, _P, c_int64, _P, _P, c_int64, _P, c_int64, _P, _P, c_int64, _P]),
    "ocn_dense_from_csr": (c_int32, [_P, _P, c_int64, c_int64, _P, _P, _P]),
    "ocn_dense_block_mm_bits": (c_int32, [_P, _P, c_int64, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32, _P, c_int64, _P]),
    "ocn_bitrows_from_csr": (c_int32, [_P, _P, c_int64, _P, c_int64, _P]),
    "ocn_bitrows_count": (c_int32, [_P, c_int64, c_int64, c_int64, _P, _P]),
    "ocn_bitrows_fill": (c_int32, [_P, c_int64, c_int64, c_int64, _P, _P, _P]),
    "ocn_rows_ln_relu": (c_int32, [_P, _P, _P, c_float, c_int32, c_int64, c_int32, _P, _P]),
    "ocn_fill_rows": (c_int32, [_P, c_int64, c_int32, _P, _P, c_int64, _P]),
    "ocn_combine3": (c_int32, [_P, _P, _P, _P, c_int64, _P, _P]),
    "ocn_linear_panel_bytes": (c_int64, [c_int32, c_int32]),
    "ocn_linear_split_weight": (c_int32, [_P, c_int32, c_int32, _P, _P]),
    "ocn_linear_bf16x6": (c_int32, [_P, c_int64, c_int32, _P, c_int32, _P, _P, _P, c_float, c_int32, _P, _P,
                                    _P, _P]),
    "ocn_linear_grouped": (c_int32, [_P, c_int32, c_int32, c_int32, _P]),
    "ocn_heads_nvec": (c_int32, []),
    "ocn_heads_nscal": (c_int32, []),
    "ocn_heads_scratch_bytes": (c_int64, [c_int32]),
    "ocn_heads_const_bytes": (c_int64, [c_int32]),
    "ocn_heads_panel_bytes": (c_int64, [c_int32, c_int32]),
    "ocn_heads_split_weight": (c_int32, [_P, c_int32, c_int32, c_float, _P, _P]),
    "ocn_heads_fused": (c_int32, [_P, _P]),
    "ocn_heads_small_batch": (c_int64, [c_int64]),
}


class OcnHipError(RuntimeError):
    pass


class OcnLinearGroup(ctypes.Structure):
    """Mirror of ``OcnLinearGroup`` in include/ocn_hip.h."""
    _fields_ = [("X", c_void_p), ("ldX", c_int64), ("M", c_int64), ("Wp", c_void_p), ("bias", c_void_p),
                ("gamma", c_void_p), ("beta", c_void_p), ("eps", c_float), ("relu", c_int32),
                ("scale", c_void_p), ("addend", c_void_p), ("ldAdd", c_int64), ("dotw", c_void_p),
                ("dotb", c_void_p), ("Y", c_void_p), ("ldY", c_int64), ("row_range", c_void_p),
                ("y_row_map", c_void_p), ("add_bcast", c_int32)]


class OcnHeadsArgs(ctypes.Structure):
    """Mirror of ``OcnHeadsArgs`` in include/ocn_hip.h."""
    _fields_ = [("x", c_void_p * 3), ("ldx", c_int64), ("B", c_int64), ("H", c_int32),
                ("p_first", c_void_p * 3), ("p_mid", c_void_p * 2), ("p_out", c_void_p * 3),
                ("vec", c_void_p), ("ranges", c_void_p), ("y_row_map", c_void_p), ("y", c_void_p),
                ("dump", c_void_p), ("cpark", c_void_p), ("scratch", c_void_p), ("eps", c_float), ("ln", c_int32), ("b_on_union", c_int32)]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def build_flags(extra_flags=()):
    """The compile flags of the product build (tests/test_host.py asserts that no experiment macro is among them)."""
    return ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17",
            # the reference's CPU kernels round the product and the sum separately; HIP's __fmul_rn /
            # __fadd_rn are plain * and + and would be contracted into FMAs under the default mode
            "-ffp-contract=off",
            # every unit's compile command is recorded in the library (.GCC.command.line): a build with an experiment macro
            # (-DOCN_X_*: timing ablations, several of which compute wrong results) can be told from the product build
            "-frecord-command-line",
            f"-I{INCLUDE}", f"-I{CSRC}", *extra_flags]


def build(force: bool = False, verbose: bool = False, extra_flags=(), out: str = None, jobs: int = None) -> str:
    """Compile csrc/*.hip -> ocn_amd/libocn_hip.so for gfx950 (hipcc cross-compiles without a GPU): one object per
    translation unit under csrc/_build/<flag hash>/ (rebuilt only when the unit, a header or the flags changed), compiled in
    parallel, then one link."""
    import hashlib
    from concurrent.futures import ThreadPoolExecutor
    out = out or LIB_PATH
    hdrs = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(INCLUDE, "ocn_hip.h")]
    deps = sources() + hdrs + [os.path.abspath(__file__)]
    if not force and not extra_flags and os.path.exists(out) and \
            os.path.getmtime(out) >= max(os.path.getmtime(d) for d in deps):
        return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = build_flags(extra_flags)
    tag = hashlib.sha1(" ".join(flags).encode()).hexdigest()[:10]
    bdir = os.path.join(CSRC, "_build", tag)
    os.makedirs(bdir, exist_ok=True)
    hdr_time = max(os.path.getmtime(h) for h in hdrs)
    todo, objs = [], []
    for src in sources():
        obj = os.path.join(bdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_time):
            todo.append([hipcc, *flags, "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    with ThreadPoolExecutor(max_workers=jobs or min(8, os.cpu_count() or 1)) as ex:
        list(ex.map(run, todo))
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs])
    return out


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OcnHipError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'`). "
                "ocn_amd has no CPU fallback.")
        # One HIP runtime per process: the torch wheel bundles its own libamdhip64 (same SONAME as
        # /opt/rocm's).  Importing torch first makes the loader hand that copy to libocn_hip.so too;
        # the other order would silently run torch on a different runtime (hipErrorNoDevice later).
        import torch  # noqa: F401
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)          # AttributeError if the symbol is not exported
            fn.restype, fn.argtypes = res, args
        if l.ocn_abi_version() != ABI_VERSION:
            raise OcnHipError("libocn_hip.so ABI version mismatch")
        _lib = l
    return _lib


def check(status: int, what: str) -> None:
    if status != 0:
        raise OcnHipError(f"{what} failed with status {status}"
                          + (" (invalid argument)" if status == -1 else ""))


def ptr(t) -> c_void_p:
    """Device pointer of a torch tensor (None -> NULL)."""
    return c_void_p(0 if t is None else t.data_ptr())


def stream_ptr() -> c_void_p:
    """torch's current HIP stream of the current device, as a raw hipStream_t.  Through the raw
    getter: ``torch.cuda.current_stream()`` walks is_available()/os.getenv on every call, which is
    most of a launch's host cost at nine launches per candidate batch."""
    import torch
    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    if raw is None:
        return c_void_p(torch.cuda.current_stream().cuda_stream)
    return c_void_p(raw(torch._C._cuda_getDevice()))
