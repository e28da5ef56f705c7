"""ctypes binding of the C ABI in include/reveal_amd.h.

One `Lib` per index width, mirroring the reference's two extension modules
(setup.py:19-32): libreveal_amd.so (reveallib) and libreveal_amd64.so
(reveallib64).  There is no Python or CPU fallback: a missing shared object is
an ImportError, a missing gfx950 device makes rv_new() fail.
"""
import ctypes
import importlib.util
import os
import sys
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

RV_T, RV_SA, RV_SAI, RV_LCP, RV_SO, RV_NSEP, RV_NODES = range(7)
K_SCAN_PAIR, K_SCAN_MULTI, K_SA_SORT, K_LCP, K_SPLIT, K_LABEL, K_BUBBLE, K_RADIX_SCATTER, K_RADIX_HIST, K_TEXT_ROUND, K_CASCADE, K_DIAG_TABLE, K_INIT_KEYS, K_PUBLISH = range(14)

c_i64p = ctypes.POINTER(ctypes.c_int64)
V = ctypes.c_void_p


class RvSub(ctypes.Structure):
    _fields_ = [("n", ctypes.c_int64), ("depth", ctypes.c_int32), ("nsamples", ctypes.c_int32),
                ("nnodes", ctypes.c_int32), ("parent", ctypes.c_int32), ("kind", ctypes.c_int32),
                ("reserved", ctypes.c_int32), ("nmums", ctypes.c_int64), ("nmembers", ctypes.c_int64)]


class RvAlignStats(ctypes.Structure):
    _fields_ = [("steps", ctypes.c_int64), ("splits", ctypes.c_int64), ("anchored_bp", ctypes.c_int64),
                ("levels", ctypes.c_int32), ("maxdepth", ctypes.c_int32), ("scanned_ranks", ctypes.c_int64),
                ("t_scan", ctypes.c_double), ("t_host", ctypes.c_double), ("t_split", ctypes.c_double),
                ("t_bubble", ctypes.c_double)]


TRACE_DTYPE = np.dtype([("key", np.int64), ("n", np.int64), ("depth", np.int32), ("nsamples", np.int32),
                        ("nnodes", np.int32), ("picked", np.int32), ("nmums", np.int64), ("l", np.uint32),
                        ("mn", np.int32), ("sp_min", np.int64), ("h_sa", np.uint64), ("h_lcp", np.uint64),
                        ("h_mums", np.uint64)], align=True)

# every symbol include/reveal_amd.h declares: (restype, argtypes)
_I, _L, _D = ctypes.c_int, ctypes.c_int64, ctypes.c_double
SYMBOLS = {
    "rv_last_error": (ctypes.c_char_p, []),
    "rv_abi_version": (_I, []),
    "rv_sa_bits": (_I, []),
    "rv_device_count": (_I, []),
    "rv_new": (V, [_I]),
    "rv_free": (None, [V]),
    "rv_add_sample": (_I, [V]),
    "rv_add_sequence": (_I, [V, ctypes.c_char_p, _L, c_i64p, c_i64p]),
    "rv_reset": (_I, [V]),
    "rv_reserve_text": (_I, [V, _L]),
    "rv_n": (_L, [V]),
    "rv_nsamples": (_I, [V]),
    "rv_nnodes": (_I, [V]),
    "rv_node_list": (_I, [V, V]),
    "rv_construct": (_I, [V, _I, ctypes.c_char_p, ctypes.c_char_p, _I]),
    "rv_upload": (_I, [V]),
    "rv_upload_again": (_I, [V]),
    "rv_get_array": (_L, [V, _I, V, _L]),
    "rv_getmums": (_L, [V, _I]),
    "rv_fetch_mums": (_I, [V, V, V, V, _L]),
    "rv_getmultimums": (_L, [V, _I, _I, _I, c_i64p]),
    "rv_fetch_multi": (_I, [V, V, V, V, V, V]),
    "rv_align_begin": (_I, [V, _I, _I]),
    "rv_frontier_size": (_I, [V]),
    "rv_frontier_scan": (_I, [V]),
    "rv_sub_skip_scan": (_I, [V, _I]),
    "rv_sub_info": (_I, [V, _I, ctypes.POINTER(RvSub)]),
    "rv_sub_nodes": (_I, [V, _I, V]),
    "rv_sub_mums": (_I, [V, _I, V, V, V, V, V]),
    "rv_sub_array": (_L, [V, _I, _I, V, _L]),
    "rv_sub_split": (_I, [V, _I, ctypes.c_uint32, _I, V, V, _I, V, _I, V, _I, V, _I]),
    "rv_frontier_commit": (_I, [V, V]),
    "rv_align_end": (_I, [V]),
    "rv_align_builtin": (_I, [V, _I, _I, ctypes.POINTER(RvAlignStats)]),
    "rv_align_builtin_until": (_I, [V, _I, _I, _I, ctypes.POINTER(RvAlignStats)]),
    "rv_align_builtin_resume": (_I, [V, ctypes.POINTER(RvAlignStats)]),
    "rv_align_builtin_continue": (_I, [V, _I, ctypes.POINTER(RvAlignStats)]),
    "rv_frontier_counts": (_I, [V, V]),
    "rv_frontier_export": (_I, [V, V, V, V]),
    "rv_frontier_pack": (_L, [V, V, _I, V, V, V, _I]),
    "rv_frontier_import": (_I, [V, _I, _I, ctypes.c_uint32, _I, _I, V, V, V, _L, V, V, V, _I]),
    "rv_frontier_seeds_export": (_L, [V, V, _I, V, _L]),
    "rv_frontier_seeds_import": (_I, [V, _I, V, _L]),
    "rv_maxlcp": (ctypes.c_uint32, [V]),
    "rv_cascade_info": (ctypes.c_int, [V, V]),
    "rv_cascade_why": (ctypes.c_char_p, [V]),
    "rv_anchor_count": (_L, [V, c_i64p]),
    "rv_fetch_anchors": (_I, [V, V, V, V]),
    "rv_set_result_buffers": (_I, [V, V, _L, V, _L, V, _L]),
    "rv_set_trace": (_I, [V, _I]),
    "rv_set_preselect": (_I, [V, _L]),
    "rv_set_option": (_I, [V, ctypes.c_char_p, _L]),
    "rv_set_launch_trace": (_I, [_I]),
    "rv_batch_new": (V, []),
    "rv_batch_add": (_I, [V, V]),
    "rv_batch_run": (_I, [V, _I, _I, _I, V, V]),
    "rv_batch_info": (_I, [V, V]),
    "rv_batch_free": (None, [V]),
    "rv_dev_alloc": (V, [_I, _L]),
    "rv_dev_free": (_I, [_I, V]),
    "rv_ipc_export": (_I, [_I, V, V]),
    "rv_ipc_open": (V, [_I, V]),
    "rv_ipc_close": (_I, [_I, V]),
    "rv_dev_copy": (_I, [_I, V, V, _L]),
    "rv_get_option": (_I, [V, ctypes.c_char_p, c_i64p]),
    "rv_option_count": (_I, []),
    "rv_option_name": (ctypes.c_char_p, [_I]),
    "rv_trace_count": (_L, [V]),
    "rv_fetch_trace": (_I, [V, V, _L]),
    "rv_clone": (V, [V]),
    "rv_sx_main": (V, [V]),
    "rv_sx_copy": (V, [V]),
    "rv_sx_free": (None, [V]),
    "rv_sx_info": (_I, [V, ctypes.POINTER(RvSub)]),
    "rv_sx_nodes": (_I, [V, V]),
    "rv_sx_array": (_L, [V, _I, V, _L]),
    "rv_sx_scan": (_L, [V, _I, _I, c_i64p]),
    "rv_sx_fetch": (_I, [V, V, V, V, V, V]),
    "rv_sx_split": (_I, [V, V, _I, V, _I, V, _I, V, _I, V]),
    "rv_sx_extract": (_I, [V, V, _I]),
    "rv_chain": (_L, [_L, _I, V, V, V, V, V, _L, _L, _I, V, V]),
    "rv_prof_enable": (_I, [V, _I]),
    "rv_prof_reset": (_I, [V]),
    "rv_prof_get": (_I, [V, _I, c_i64p, ctypes.POINTER(_D), ctypes.POINTER(_D)]),
    "rv_measure_bandwidth": (_I, [_I, _L, _I, ctypes.POINTER(_D), ctypes.POINTER(_D)]),
    "rv_sa_stats": (_I, [V] + [ctypes.POINTER(_I)] * 4 + [c_i64p, ctypes.POINTER(_I)]),
    "rv_sa_diag_table": (_I, [V]),
    "rv_sa_tail": (_I, [V, c_i64p]),
    "rv_set_picker": (_I, [V, _I, V]),
    "rv_picker_info": (_I, [V, c_i64p]),
    "rv_pick_chain": (_I, [V, _I, _L, V, V, V, V, V, _I, V, V, V, _I, V]),
    "rv_graph_import": (V, [_L, V, V, V, V, V, V, V, _L, V, V, V, V, V, V, _I, V, V, _I, V, _I]),
    "rv_graph_pick": (_I, [V, V, _I, _L, V, V, V, V, V, V, V, _I, V]),
    "rv_graph_align": (_I, [V, V, _L, V, V, ctypes.c_uint32, V, _I, V, V]),
    "rv_graph_align_fetch": (_I, [V, V]),
    "rv_graph_finish": (_I, [V]),
    "rv_set_graph_picker": (_I, [V, V, V]),
    "rv_graph_new": (V, []),
    "rv_graph_replay_begin": (V, [_I, V, V]),
    "rv_set_replay_graph": (_I, [V, V]),
    "rv_graph_add_linear": (_I, [V, _L, _L, _I]),
    "rv_graph_read_gfa": (_L, [V, V, V, V, _L, V]),
    "rv_graph_paths": (_I, [V, V]),
    "rv_graph_seal": (_I, [V]),
    "rv_gfa_parse": (V, [V, _L]),
    "rv_graph_adopt": (_L, [V, V, V, V, V]),
    "rv_gfa_parsed_free": (None, [V]),
    "rv_add_sequences": (_I, [V, V, _L, V, _L]),
    "rv_graph_node_kinds": (_I, [V, V]),
    "rv_graph_literal": (_I, [V]),
    "rv_graph_replay": (V, [_I, V, V, _L, V, V, V]),
    "rv_graph_error": (ctypes.c_char_p, [V]),
    "rv_graph_sizes": (_I, [V, c_i64p]),
    "rv_graph_export": (_I, [V] * 15),
    "rv_graph_prune": (_I, [V, V]),
    "rv_graph_gfa": (_L, [V, V, _I, V, V, V]),
    "rv_graph_free": (None, [V]),
    "rv_test_exclusive_sum_u32": (_I, [V, V, _L]),
    "rv_test_inclusive_max_u32": (_I, [V, V, _L]),
    "rv_test_radix_sort": (_I, [V, V, _L, _I, _I]),
    "rv_test_radix_time": (_I, [_L, _I, _I, _I, _I, ctypes.POINTER(_D), c_i64p]),
}


def _share_torch_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (same SONAME as /opt/rocm's, different file).  Two HIP
    runtimes in one process do not both get the GPU: whichever starts second reports "no HIP GPUs", and device pointers
    of one mean nothing to the other.  The hand-off of reveal_amd/shard.py passes torch device tensors (RCCL send/recv)
    to this library, so when torch is installed its copy is loaded first and the library binds to it, whatever the
    import order.  RV_SYSTEM_HIP=1 keeps the system runtime (then import torch before reveal_amd, or not at all)."""
    if os.environ.get("RV_SYSTEM_HIP") or "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(path):
        ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)


class Lib:
    def __init__(self, sa64=False):
        self.sa64 = sa64
        _share_torch_runtime()
        name = "libreveal_amd64.so" if sa64 else "libreveal_amd.so"
        self.path = os.path.join(os.environ.get("RV_LIB_DIR") or _HERE, name)     # RV_LIB_DIR: another build of the two libraries
        if not os.path.exists(self.path):
            raise ImportError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950); reveal_amd has no CPU fallback" % self.path)
        self.dll = ctypes.CDLL(self.path)
        for sym, (res, args) in SYMBOLS.items():
            fn = getattr(self.dll, sym)      # AttributeError = ABI mismatch: fail loudly
            fn.restype, fn.argtypes = res, args
        self.sa_t = np.int64 if sa64 else np.int32
        self.lcp_t = np.uint32 if sa64 else np.int32
        assert self.dll.rv_sa_bits() == (64 if sa64 else 32)
        # the one process-wide diagnostic (not a handle's switch, rv_set_launch_trace): RV_LAUNCH_TRACE in the environment when the library is loaded
        v = os.environ.get("RV_LAUNCH_TRACE")
        if v is not None and v.strip() not in ("", "0"):
            self.dll.rv_set_launch_trace(1)

    def err(self):
        return (self.dll.rv_last_error() or b"").decode(errors="replace")


_libs = {}
_device = 0


def set_device(i):
    """HIP device ordinal new index objects are created on (one process per GPU)."""
    global _device
    _device = int(i)


def device():
    return _device


def measure_bandwidth(nbytes=1 << 30, iters=10):
    """-> (read GB/s, copy GB/s counting read + write) of streaming kernels on the current device: the node's practical HBM ceiling"""
    lib = get(False)
    r, c = _D(0), _D(0)
    if lib.dll.rv_measure_bandwidth(device(), int(nbytes), int(iters), ctypes.byref(r), ctypes.byref(c)) != 0:
        raise RuntimeError(lib.err())
    return r.value, c.value


def set_launch_trace(on=True):
    """process-wide: every kernel launch of both libraries prints its source line and is waited for (finds the kernel behind a GPU fault)"""
    return [bool(get(w).dll.rv_set_launch_trace(1 if on else 0)) for w in (False, True)]


def get(sa64=False):
    if sa64 not in _libs:
        _libs[sa64] = Lib(sa64)
    return _libs[sa64]
