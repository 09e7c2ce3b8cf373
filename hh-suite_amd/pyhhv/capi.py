"""ctypes binding of libhhviterbi_hip.so (include/hhviterbi_hip.h) -- thin, no arithmetic here.

The library is the product; this module only marshals numpy arrays into the C ABI so that the
parity tests and bench.py can call it exactly the way a C/C++ host would.  It fails loudly when
the shared object is missing (there is no Python or CPU fallback).
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# hardware queues for the streams of the MAC length classes (hhv_create's comment): exported here because a host that
# loads torch has usually initialised HIP before the library sees its first call
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
LIB_PATH = os.environ.get("HHV_LIB", os.path.join(os.path.dirname(HERE), "lib", "libhhviterbi_hip.so"))

HHV_ALIGN_BACKTRACE = 1
HHV_ALIGN_CELLOFF = 2
HHV_STREAM_PAD = 256
REC_DW = 28

c_float_p = C.POINTER(C.c_float)
c_int_p = C.POINTER(C.c_int32)


class HhvParams(C.Structure):
    _fields_ = [("device", C.c_int32), ("local", C.c_int32), ("egq", C.c_float), ("egt", C.c_float),
                ("shift", C.c_float), ("corr", C.c_float), ("ssw", C.c_float), ("ss_mode", C.c_int32)]


RESULT_DTYPE = np.dtype([("score", np.float32), ("i2", np.int32), ("j2", np.int32), ("index", np.int32)])
HIT_DTYPE = np.dtype([("score", np.float32), ("viterbi_score", np.float32), ("score_ss", np.float32), ("index", np.int32),
                      ("i1", np.int32), ("j1", np.int32), ("i2", np.int32), ("j2", np.int32),
                      ("nsteps", np.int32), ("matched_cols", np.int32)])

# every symbol include/hhviterbi_hip.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "hhv_abi_version", "hhv_last_error", "hhv_record_bytes", "hhv_pack_profile", "hhv_fast_log2_tables",
    "hhv_create", "hhv_destroy", "hhv_set_params", "hhv_set_launch_policy", "hhv_set_fast_log2_tables", "hhv_set_query", "hhv_set_ss_tables", "hhv_set_query_ss", "hhv_set_ss_mode",
    "hhv_upload_templates", "hhv_upload_templates_ss", "hhv_adopt_device_stream",
    "hhv_upload_raw_templates", "hhv_rawset_free", "hhv_prepare_templates", "hhv_prep_params_check", "hhv_rawset_pav", "hhv_tset_records_of", "hhv_tset_download",
    "hhv_prefilter_upload_db", "hhv_prefilter_free_db", "hhv_prefilter_scores", "hhv_prefilter_first",
    "hhv_mac_realign", "hhv_mac_realign_hits", "hhv_mac_realign_tset", "hhv_mac_set_ss", "hhv_mac_celloff", "hhv_mac_path", "hhv_mac_posterior", "hhv_mac_set_lists", "hhv_mac_list", "hhv_macset_free",
    "hhv_prepare_subset", "hhv_rawdb_write", "hhv_rawdb_open", "hhv_rawset_size", "hhv_rawset_lengths",
    "hhv_db_write", "hhv_db_open", "hhv_tset_gather", "hhv_tset_free", "hhv_tset_size", "hhv_tset_cells", "hhv_tset_records", "hhv_align", "hhv_align_async",
    "hhv_sync", "hhv_check_error", "hhv_tset_set_neff", "hhv_hit_paths_packed", "hhv_stream", "hhv_last_kernel_ms", "hhv_set_celloff", "hhv_set_celloff_paths", "hhv_set_global_batch", "hhv_backtrace_matrix", "hhv_backtrace", "hhv_hits",
    "hhv_hit_path", "hhv_hit_path_pool", "hhv_topk", "hhv_device_count", "hhv_shard_plan", "hhv_segment_plan",
    "hhv_tset_set_global_ids", "hhv_merge_hits",
]


class HhvPrepParams(C.Structure):
    _fields_ = [("gapd", C.c_float), ("gape", C.c_float), ("gapf", C.c_float), ("gapg", C.c_float),
                ("gaph", C.c_float), ("gapi", C.c_float), ("gapb", C.c_float), ("pcm", C.c_int32),
                ("pca", C.c_float), ("pcb", C.c_float), ("pcc", C.c_float), ("columnscore", C.c_int32),
                ("pb", C.c_float * 20), ("R", C.c_float * 400)]


def prep_params(pb, R, gap=(0.15, 1.0, 0.6, 0.6, 0.6, 0.6, 1.0), pc=(2, 1.0, 1.5, 1.0), columnscore=1):
    """Defaults of the reference: src/hhdecl.cpp:64-67,74-80,98."""
    P = HhvPrepParams()
    (P.gapd, P.gape, P.gapf, P.gapg, P.gaph, P.gapi, P.gapb) = [float(x) for x in gap]
    P.pcm, P.pca, P.pcb, P.pcc = int(pc[0]), float(pc[1]), float(pc[2]), float(pc[3])
    P.columnscore = int(columnscore)
    P.pb[:] = [float(x) for x in np.asarray(pb).reshape(-1)]
    P.R[:] = [float(x) for x in np.asarray(R).reshape(-1)]
    return P


class HhvError(RuntimeError):
    pass


_lib = None
_libs = {}
FMA_LIB_PATH = os.path.join(os.path.dirname(HERE), "lib", "libhhviterbi_hip_fma.so")   # the opt-in fused-emission build


def load(path=None):
    """the product library (HHV_LIB or hh-suite_amd/lib/libhhviterbi_hip.so); path = another build of it (the opt-in
    libhhviterbi_hip_fma.so), loaded next to the default one"""
    global _lib
    if path is None and _lib is not None:
        return _lib
    if path is not None and path in _libs:
        return _libs[path]
    lib_path = LIB_PATH if path is None else path
    if not os.path.exists(lib_path):
        raise HhvError("%s not built (%s): run `make lib` / __graft_entry__.build()" % (os.path.basename(lib_path), lib_path))
    L = C.CDLL(lib_path)
    L.hhv_abi_version.restype = C.c_int
    L.hhv_last_error.restype = C.c_char_p
    L.hhv_record_bytes.restype = C.c_int32
    L.hhv_pack_profile.argtypes = [c_float_p, c_float_p, C.c_int32, C.c_int32, c_float_p]
    L.hhv_fast_log2_tables.argtypes = [c_float_p, c_float_p]
    L.hhv_device_count.argtypes = [c_int_p]
    L.hhv_shard_plan.argtypes = [C.c_int32, c_int_p, C.c_int32, c_int_p]
    L.hhv_segment_plan.argtypes = [C.c_int32, c_int_p, C.POINTER(C.c_int64), c_int_p]
    L.hhv_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(HhvParams)]
    L.hhv_destroy.argtypes = [C.c_void_p]
    L.hhv_destroy.restype = None
    L.hhv_set_query.argtypes = [C.c_void_p, c_float_p, c_float_p, C.c_int32]
    L.hhv_upload_templates.argtypes = [C.c_void_p, C.c_int32, c_int_p, C.POINTER(c_float_p), C.POINTER(c_float_p),
                                       C.POINTER(C.c_void_p)]
    c_i8pp = C.POINTER(C.c_void_p)
    L.hhv_upload_templates_ss.argtypes = [C.c_void_p, C.c_int32, c_int_p, C.POINTER(c_float_p), C.POINTER(c_float_p),
                                          c_i8pp, c_i8pp, c_i8pp, C.POINTER(C.c_void_p)]
    L.hhv_set_ss_tables.argtypes = [C.c_void_p, c_float_p, c_float_p, c_float_p]
    L.hhv_set_query_ss.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.hhv_set_ss_mode.argtypes = [C.c_void_p, C.c_int32]
    L.hhv_set_global_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.hhv_mac_set_ss.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.hhv_set_params.argtypes = [C.c_void_p, C.POINTER(HhvParams)]
    L.hhv_set_fast_log2_tables.argtypes = [C.c_void_p, c_float_p, c_float_p]
    L.hhv_hit_path_pool.argtypes = [C.c_void_p, C.c_void_p] + [C.POINTER(C.c_void_p)] * 5
    L.hhv_adopt_device_stream.argtypes = [C.c_void_p, C.c_int32, c_int_p, C.c_void_p, C.POINTER(C.c_void_p)]
    vpp = C.POINTER(C.c_void_p)
    L.hhv_upload_raw_templates.argtypes = [C.c_void_p, C.c_int32, c_int_p, C.POINTER(c_float_p), C.POINTER(c_float_p),
                                           C.POINTER(c_float_p), c_float_p, vpp, vpp, vpp, C.POINTER(C.c_void_p)]
    L.hhv_rawset_free.argtypes = [C.c_void_p]
    L.hhv_rawset_free.restype = None
    L.hhv_prepare_templates.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(HhvPrepParams), c_float_p, C.POINTER(C.c_void_p)]
    L.hhv_rawset_pav.argtypes = [C.c_void_p, C.c_void_p, c_float_p]
    L.hhv_tset_records_of.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, c_float_p]
    L.hhv_tset_download.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.hhv_prefilter_upload_db.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
    L.hhv_prefilter_free_db.argtypes = [C.c_void_p]
    L.hhv_prefilter_free_db.restype = None
    L.hhv_prefilter_scores.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                       C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]
    L.hhv_mac_realign.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_int32, C.c_float, C.c_float, C.POINTER(C.c_void_p), C.c_void_p]
    L.hhv_mac_realign_hits.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                       C.c_float, C.c_float, C.POINTER(C.c_void_p), C.c_void_p]
    L.hhv_mac_path.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.POINTER(C.c_int32)]
    L.hhv_mac_posterior.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    L.hhv_mac_celloff.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    L.hhv_mac_set_lists.argtypes = [C.c_void_p, C.c_int32]
    L.hhv_mac_list.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    L.hhv_mac_list.restype = C.c_int64
    L.hhv_macset_free.argtypes = [C.c_void_p]
    L.hhv_macset_free.restype = None
    L.hhv_db_write.argtypes = [C.c_char_p, C.c_int32, c_int_p, C.POINTER(c_float_p), C.POINTER(c_float_p), C.c_void_p,
                               C.c_void_p, C.c_void_p]
    L.hhv_db_open.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p)]
    L.hhv_tset_free.argtypes = [C.c_void_p]
    L.hhv_tset_free.restype = None
    L.hhv_tset_size.argtypes = [C.c_void_p]
    L.hhv_tset_size.restype = C.c_int32
    L.hhv_tset_cells.argtypes = [C.c_void_p, C.c_int32]
    L.hhv_tset_cells.restype = C.c_int64
    L.hhv_tset_records.argtypes = [C.c_void_p]
    L.hhv_tset_records.restype = C.c_int64
    L.hhv_align.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    L.hhv_align_async.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    L.hhv_sync.argtypes = [C.c_void_p]
    L.hhv_check_error.argtypes = [C.c_void_p]
    L.hhv_stream.argtypes = [C.c_void_p]
    L.hhv_stream.restype = C.c_void_p
    L.hhv_last_kernel_ms.argtypes = [C.c_void_p, c_float_p]
    L.hhv_set_celloff.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    L.hhv_set_celloff_paths.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]
    L.hhv_backtrace_matrix.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    L.hhv_hits.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    if hasattr(L, "hhv_backtrace"):  # (an A/B partner built from an older tree, tools/gpu_ab.sh, may lack the newest entries)
        L.hhv_backtrace.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.hhv_hit_path.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.c_void_p, c_int_p]
    L.hhv_topk.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p, c_int_p]
    L.hhv_tset_set_global_ids.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.hhv_merge_hits.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, c_int_p]
    if path is None:
        _lib = L
    else:
        _libs[path] = L
    return L


def device_count():
    """hhv_device_count: usable HIP devices (0 when there is none; no exception)."""
    n = C.c_int32(0)
    load().hhv_device_count(C.byref(n))
    return n.value


def shard_plan(lengths, n_shards):
    """hhv_shard_plan: shard of every template (the partition pyhhv.shard.shard_templates computes in Python)."""
    L = np.ascontiguousarray(lengths, dtype=np.int32)
    out = np.zeros(L.shape[0], dtype=np.int32)
    _check(load().hhv_shard_plan(L.shape[0], L.ctypes.data_as(c_int_p), int(n_shards), out.ctypes.data_as(c_int_p)))
    return out


def segment_plan(lengths):
    """hhv_segment_plan: the DP kernel's work queue for a template stream of these lengths -> (n_seg, [(first, end)] in draw
    order incl. the terminal entry).  Pure host function."""
    L = np.ascontiguousarray(lengths, dtype=np.int32)
    seg = np.zeros(2 * (L.shape[0] + 1), dtype=np.int64)
    n_seg = C.c_int32(0)
    _check(load().hhv_segment_plan(L.shape[0], L.ctypes.data_as(c_int_p), seg.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(n_seg)))
    return n_seg.value, seg[: 2 * (n_seg.value + 1)].reshape(-1, 2)


def _check(rc):
    if rc != 0:
        raise HhvError("hhv error %d: %s" % (rc, load().hhv_last_error().decode()))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def pack_profile(p, tr, index=-1):
    """Product packer (host only): index >= 0 -> (L+1, 28) header+columns, index < 0 -> (L, 28)."""
    p, tr = _f32(p), _f32(tr)
    L = p.shape[0] - 1
    out = np.zeros(((L + 1) if index >= 0 else L, REC_DW), dtype=np.float32)
    _check(load().hhv_pack_profile(p.ctypes.data_as(c_float_p), tr.ctypes.data_as(c_float_p), L, index,
                                   out.ctypes.data_as(c_float_p)))
    return out


def _row_pointers(block):
    """(n, ...) C-contiguous float32 block -> ctypes float*[n] with element k pointing at block[k] (no per-row Python object)."""
    assert block.dtype == np.float32 and block.flags.c_contiguous
    n = block.shape[0]
    addr = (block.ctypes.data + np.arange(n, dtype=np.uint64) * np.uint64(block.strides[0])).astype(np.uint64)
    arr = (c_float_p * n).from_buffer_copy(addr.tobytes())
    return arr


def db_write_blocks(path, P, T):
    """hhv_db_write for equal-length templates held in two blocks: P (n, L+1, 20), T (n, L+1, 7)."""
    n, L = P.shape[0], P.shape[1] - 1
    Ls = np.full(n, L, dtype=np.int32)
    _check(load().hhv_db_write(str(path).encode(), n, Ls.ctypes.data_as(c_int_p), _row_pointers(P), _row_pointers(T), None, None, None))
    return Ls


def db_write(path, tps, ttrs):
    """hhv_db_write: pack host profiles into a binary template database file (no device needed)."""
    tps = [_f32(a) for a in tps]
    ttrs = [_f32(a) for a in ttrs]
    n = len(tps)
    Ls = np.array([a.shape[0] - 1 for a in tps], dtype=np.int32)
    pp = (c_float_p * n)(*[a.ctypes.data_as(c_float_p) for a in tps])
    tt = (c_float_p * n)(*[a.ctypes.data_as(c_float_p) for a in ttrs])
    _check(load().hhv_db_write(path.encode(), n, Ls.ctypes.data_as(c_int_p), pp, tt, None, None, None))
    return Ls


def rawdb_write(path, fs, trs, neffs, neff_hmm):
    """hhv_rawdb_write: raw HMMs (as HMM::Read leaves them) -> raw template database file (no device needed)."""
    fs = [_f32(a) for a in fs]
    trs = [_f32(a) for a in trs]
    neffs = [_f32(a) for a in neffs]
    n = len(fs)
    Ls = np.array([a.shape[0] - 1 for a in trs], dtype=np.int32)
    nh = _f32(neff_hmm)
    ff = (c_float_p * n)(*[a.ctypes.data_as(c_float_p) for a in fs])
    tt = (c_float_p * n)(*[a.ctypes.data_as(c_float_p) for a in trs])
    nn = (c_float_p * n)(*[a.ctypes.data_as(c_float_p) for a in neffs])
    L = load()
    L.hhv_rawdb_write.argtypes = [C.c_char_p, C.c_int32, c_int_p, C.c_void_p, C.c_void_p, C.c_void_p, c_float_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p]
    _check(L.hhv_rawdb_write(str(path).encode(), n, Ls.ctypes.data_as(c_int_p), ff, tt, nn, nh.ctypes.data_as(c_float_p),
                             None, None, None))
    return Ls


def fast_log2_tables():
    lg2 = np.zeros(1025, dtype=np.float32)
    diff = np.zeros(1025, dtype=np.float32)
    _check(load().hhv_fast_log2_tables(lg2.ctypes.data_as(c_float_p), diff.ctypes.data_as(c_float_p)))
    return lg2, diff


class TemplateSet:
    def __init__(self, ctx, handle, Ls):
        self.ctx, self.h, self.L = ctx, handle, np.asarray(Ls, dtype=np.int32)
        self.n = len(self.L)

    def free(self):
        if self.h:
            self.ctx.lib.hhv_tset_free(self.h)
            self.h = None

    def cells(self):
        return int(self.ctx.lib.hhv_tset_cells(self.h, self.ctx.Lq))

    def records(self):
        return int(self.ctx.lib.hhv_tset_records(self.h))


class Context:
    """One hhv_ctx = the per-thread Viterbi object of the reference (src/hhviterbirunner.h:21-34)."""

    def __init__(self, local=0, egq=0.0, egt=0.0, shift=-0.03, corr=0.1, ssw=0.11, ss_mode=2, device=0, lib_path=None):
        self.lib = load(lib_path)
        self.par = HhvParams(int(device), int(local), float(egq), float(egt), float(shift), float(corr), float(ssw),
                             int(ss_mode))
        h = C.c_void_p()
        self._chk(self.lib.hhv_create(C.byref(h), C.byref(self.par)))
        self.h = h
        self.Lq = 0

    def close(self):
        if self.h:
            self.lib.hhv_destroy(self.h)
            self.h = None

    def _chk(self, rc):
        """status check against the library THIS context lives in (a second build loaded by path keeps its own error text)"""
        if rc != 0:
            raise HhvError("hhv error %d: %s" % (rc, self.lib.hhv_last_error().decode()))

    def set_launch_policy(self, pair_mode=-1, pair_swap=0, blocks_per_cu=0, trace_mode=-1):
        """hhv_set_launch_policy: pair_mode -1 library's choice / 0 one launch per strip / 1 pair launches wherever possible;
        trace_mode -1 by set size / 0 one lane per template / 1 one wavefront per template"""
        self._chk(self.lib.hhv_set_launch_policy(self.h, int(pair_mode), int(pair_swap), int(blocks_per_cu), int(trace_mode)))

    def set_query(self, p, tr):
        p, tr = _f32(p), _f32(tr)
        self.Lq = p.shape[0] - 1
        self._chk(self.lib.hhv_set_query(self.h, p.ctypes.data_as(c_float_p), tr.ctypes.data_as(c_float_p), self.Lq))

    def upload(self, tps, ttrs, t_ss=None):
        """t_ss: optional list of (ss_pred, ss_conf, ss_dssp) int8 arrays per template (entries may be None)."""
        tps = [_f32(a) for a in tps]
        ttrs = [_f32(a) for a in ttrs]
        n = len(tps)
        Ls = np.array([a.shape[0] - 1 for a in tps], dtype=np.int32)
        pp = (c_float_p * n)(*[a.ctypes.data_as(c_float_p) for a in tps])
        tt = (c_float_p * n)(*[a.ctypes.data_as(c_float_p) for a in ttrs])
        h = C.c_void_p()
        if t_ss is None:
            self._chk(self.lib.hhv_upload_templates(self.h, n, Ls.ctypes.data_as(c_int_p), pp, tt, C.byref(h)))
        else:
            keep = [[None if x is None else np.ascontiguousarray(x, dtype=np.int8) for x in t] for t in t_ss]
            arrs = []
            for col in range(3):
                arrs.append((C.c_void_p * n)(*[(k[col].ctypes.data if k[col] is not None else None) for k in keep]))
            self._chk(self.lib.hhv_upload_templates_ss(self.h, n, Ls.ctypes.data_as(c_int_p), pp, tt, arrs[0], arrs[1],
                                                    arrs[2], C.byref(h)))
        return TemplateSet(self, h, Ls)

    def upload_blocks(self, P, T):
        """hhv_upload_templates for equal-length templates held in two blocks: P (n, L+1, 20), T (n, L+1, 7) float32."""
        n, L = P.shape[0], P.shape[1] - 1
        Ls = np.full(n, L, dtype=np.int32)
        h = C.c_void_p()
        self._chk(self.lib.hhv_upload_templates(self.h, n, Ls.ctypes.data_as(c_int_p), _row_pointers(P), _row_pointers(T), C.byref(h)))
        return TemplateSet(self, h, Ls)

    def set_ss_tables(self, S73, S33, S37):
        S73, S33, S37 = _f32(S73).reshape(-1), _f32(S33).reshape(-1), _f32(S37).reshape(-1)
        assert S73.size == 352 and S33.size == 1936 and S37.size == 352
        self._chk(self.lib.hhv_set_ss_tables(self.h, S73.ctypes.data_as(c_float_p), S33.ctypes.data_as(c_float_p),
                                          S37.ctypes.data_as(c_float_p)))

    def set_query_ss(self, ss_pred=None, ss_conf=None, ss_dssp=None):
        a = [None if x is None else np.ascontiguousarray(x, dtype=np.int8) for x in (ss_pred, ss_conf, ss_dssp)]
        self._chk(self.lib.hhv_set_query_ss(self.h, *[(x.ctypes.data if x is not None else None) for x in a]))

    def set_ss_mode(self, mode):
        self._chk(self.lib.hhv_set_ss_mode(self.h, int(mode)))

    def upload_raw(self, fs, trs, neffs, neff_hmm):
        """hhv_upload_raw_templates: raw HMMs (f[(L+2),20], tr[(L+1),7], neff[(L+1),3], Neff_HMM) -> raw set handle."""
        fs = [_f32(a) for a in fs]
        trs = [_f32(a) for a in trs]
        neffs = [_f32(a) for a in neffs]
        n = len(fs)
        Ls = np.array([a.shape[0] - 1 for a in trs], dtype=np.int32)
        nh = _f32(neff_hmm)
        ff = (c_float_p * n)(*[a.ctypes.data_as(c_float_p) for a in fs])
        tt = (c_float_p * n)(*[a.ctypes.data_as(c_float_p) for a in trs])
        nn = (c_float_p * n)(*[a.ctypes.data_as(c_float_p) for a in neffs])
        h = C.c_void_p()
        self._chk(self.lib.hhv_upload_raw_templates(self.h, n, Ls.ctypes.data_as(c_int_p), ff, tt, nn,
                                                 nh.ctypes.data_as(c_float_p), None, None, None, C.byref(h)))
        return h, Ls

    def prepare_subset(self, raw, all_Ls, params, q_pav, ids):
        """hhv_prepare_subset -> new TemplateSet of the raw templates ids (in that order)."""
        q_pav = _f32(q_pav)
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        self.lib.hhv_prepare_subset.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                                C.POINTER(C.c_void_p)]
        h = C.c_void_p()
        self._chk(self.lib.hhv_prepare_subset(self.h, raw, C.byref(params), q_pav.ctypes.data, ids.ctypes.data, len(ids), C.byref(h)))
        return TemplateSet(self, h, np.asarray(all_Ls, dtype=np.int32)[ids])

    def rawdb_open(self, path):
        """hhv_rawdb_open -> (raw set handle, lengths): the raw database file straight into HBM."""
        self.lib.hhv_rawdb_open.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p)]
        self.lib.hhv_rawset_size.argtypes = [C.c_void_p]
        self.lib.hhv_rawset_lengths.argtypes = [C.c_void_p, C.c_void_p]
        h = C.c_void_p()
        self._chk(self.lib.hhv_rawdb_open(self.h, str(path).encode(), C.byref(h)))
        Ls = np.zeros(self.lib.hhv_rawset_size(h), dtype=np.int32)
        self._chk(self.lib.hhv_rawset_lengths(h, Ls.ctypes.data))
        return h, Ls

    def prepare(self, raw, Ls, params, q_pav, ts=None):
        """hhv_prepare_templates -> TemplateSet (created on the first call, refilled afterwards)."""
        q_pav = _f32(q_pav)
        h = C.c_void_p(ts.h.value) if ts is not None else C.c_void_p()
        self._chk(self.lib.hhv_prepare_templates(self.h, raw, C.byref(params), q_pav.ctypes.data_as(c_float_p), C.byref(h)))
        return ts if ts is not None else TemplateSet(self, h, Ls)

    def rawset_pav(self, raw, n):
        pav = np.zeros((n, 20), dtype=np.float32)
        self._chk(self.lib.hhv_rawset_pav(self.h, raw, pav.ctypes.data_as(c_float_p)))
        return pav

    def rawset_free(self, raw):
        self.lib.hhv_rawset_free(raw)

    def records_of(self, ts, k):
        out = np.zeros((int(ts.L[k]) + 1, REC_DW), dtype=np.float32)
        self._chk(self.lib.hhv_tset_records_of(self.h, ts.h, int(k), out.ctypes.data_as(c_float_p)))
        return out

    def gather(self, ts, ids):
        """hhv_tset_gather -> new TemplateSet with templates ids of ts (device-side copy)."""
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        self.lib.hhv_tset_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]
        h = C.c_void_p()
        self._chk(self.lib.hhv_tset_gather(self.h, ts.h, ids.ctypes.data, len(ids), C.byref(h)))
        return TemplateSet(self, h, np.asarray(ts.L, dtype=np.int32)[ids])

    def prefilter_upload_db(self, seqs, offsets):
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        h = C.c_void_p()
        self._chk(self.lib.hhv_prefilter_upload_db(self.h, len(offsets) - 1, seqs.ctypes.data, offsets.ctypes.data, C.byref(h)))
        return (h, len(offsets) - 1)

    def prefilter_first(self, db, profile, score_offset, log_qlen, bit_factor=4, smax_thresh=10, min_hits=100):
        """hhv_prefilter_first: first prefilter stage entirely on the device -> surviving ids, best first."""
        profile = np.ascontiguousarray(profile, dtype=np.uint8)
        ids = np.zeros(db[1], dtype=np.int32)
        n = C.c_int32()
        self.lib.hhv_prefilter_first.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_int32,
                                                 C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
        self._chk(self.lib.hhv_prefilter_first(self.h, db[0], profile.ctypes.data, profile.shape[1], int(score_offset),
                                            float(log_qlen), int(bit_factor), int(smax_thresh), int(min_hits), ids.ctypes.data,
                                            len(ids), C.byref(n)))
        return ids[:n.value]

    def prefilter_free_db(self, db):
        self.lib.hhv_prefilter_free_db(db[0])

    def prefilter_scores(self, db, profile, score_offset, gapped=False, gap_init=24, gap_extend=4, subset=None):
        profile = np.ascontiguousarray(profile, dtype=np.uint8)
        assert profile.shape[0] == 220
        n = db[1] if subset is None else len(subset)
        out = np.zeros(n, dtype=np.int32)
        sub = None if subset is None else np.ascontiguousarray(subset, dtype=np.int32)
        self._chk(self.lib.hhv_prefilter_scores(self.h, db[0], profile.ctypes.data, profile.shape[1], int(score_offset),
                                             int(bool(gapped)), int(gap_init), int(gap_extend),
                                             sub.ctypes.data if sub is not None else None, 0 if sub is None else len(sub),
                                             out.ctypes.data))
        return out

    def mac_set_lists(self, on):
        """hhv_mac_set_lists: keep the -o_matrices forward / backward lists of the following mac_realign* calls"""
        self._chk(self.lib.hhv_mac_set_lists(self.h, int(on)))

    def mac_realign(self, qp, q_tr_lin, tps, t_trs, celloffs=None, local=1, shift=-0.03, mact=0.3501):
        """hhv_mac_realign -> MacSet (hits structured array + path()/posterior() accessors)."""
        qp, q_tr_lin = _f32(qp), _f32(q_tr_lin)
        tps = [_f32(a) for a in tps]
        t_trs = [_f32(a) for a in t_trs]
        n = len(tps)
        Lq = qp.shape[0] - 1
        Lt = np.array([a.shape[0] - 1 for a in tps], dtype=np.int32)
        pp = (C.c_void_p * n)(*[a.ctypes.data for a in tps])
        tt = (C.c_void_p * n)(*[a.ctypes.data for a in t_trs])
        cc = None
        keep = None
        if celloffs is not None:
            keep = [None if m is None else np.ascontiguousarray(m, dtype=np.uint8) for m in celloffs]
            for k, m in enumerate(keep):
                assert m is None or m.shape == (Lq + 1, Lt[k] + 1)
            cc = (C.c_void_p * n)(*[None if m is None else m.ctypes.data for m in keep])
        hits = np.zeros(n, dtype=MAC_HIT_DTYPE)
        h = C.c_void_p()
        self._chk(self.lib.hhv_mac_realign(self.h, qp.ctypes.data, q_tr_lin.ctypes.data, Lq, n, Lt.ctypes.data, pp, tt, cc,
                                        int(local), shift, mact, C.byref(h), hits.ctypes.data))
        return MacSet(self.lib, h, hits, Lq, Lt)

    def mac_realign_tset(self, qp, q_tr_lin, ts, template_of, t_trs, inputs, local=1, shift=-0.03, mact=0.3501):
        """hhv_mac_realign_tset: profiles from the resident set ts (hit k = template template_of[k]); inputs as in
        mac_realign_hits."""
        qp, q_tr_lin = _f32(qp), _f32(q_tr_lin)
        t_trs = [_f32(a) for a in t_trs]
        n = len(t_trs)
        Lq = qp.shape[0] - 1
        tof = np.ascontiguousarray(template_of, dtype=np.int32)
        Lt = np.asarray(ts.L, dtype=np.int32)[tof]
        tt = (C.c_void_p * n)(*[a.ctypes.data for a in t_trs])
        arr = (MacInputStruct * n)()
        keep = []
        for k, (i1, j1, i2, j2, ns, vi, vj, xi, xj) in enumerate(inputs):
            vi, vj = np.ascontiguousarray(vi, np.int32), np.ascontiguousarray(vj, np.int32)
            xi, xj = np.ascontiguousarray(xi, np.int32), np.ascontiguousarray(xj, np.int32)
            keep += [vi, vj, xi, xj]
            arr[k] = MacInputStruct(i1, j1, i2, j2, ns, len(xi), vi.ctypes.data, vj.ctypes.data,
                                    xi.ctypes.data if len(xi) else None, xj.ctypes.data if len(xj) else None)
        self.lib.hhv_mac_realign_tset.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                                  C.c_void_p, C.c_int32, C.c_float, C.c_float, C.POINTER(C.c_void_p), C.c_void_p]
        hits = np.zeros(n, dtype=MAC_HIT_DTYPE)
        h = C.c_void_p()
        self._chk(self.lib.hhv_mac_realign_tset(self.h, qp.ctypes.data, q_tr_lin.ctypes.data, Lq, ts.h, n, tof.ctypes.data, tt,
                                             C.addressof(arr), 0, None, 0, None, int(local), shift, mact, C.byref(h),
                                             hits.ctypes.data))
        return MacSet(self.lib, h, hits, Lq, Lt)

    def mac_realign_hits(self, qp, q_tr_lin, tps, t_trs, inputs, qranges=(), tranges=(), local=1, shift=-0.03, mact=0.3501):
        """hhv_mac_realign_hits. inputs: list of (i1, j1, i2, j2, nsteps, i_steps, j_steps, excluded_i, excluded_j)."""
        qp, q_tr_lin = _f32(qp), _f32(q_tr_lin)
        tps = [_f32(a) for a in tps]
        t_trs = [_f32(a) for a in t_trs]
        n = len(tps)
        Lq = qp.shape[0] - 1
        Lt = np.array([a.shape[0] - 1 for a in tps], dtype=np.int32)
        pp = (C.c_void_p * n)(*[a.ctypes.data for a in tps])
        tt = (C.c_void_p * n)(*[a.ctypes.data for a in t_trs])
        arr = (MacInputStruct * n)()
        keep = []
        for k, (i1, j1, i2, j2, ns, vi, vj, xi, xj) in enumerate(inputs):
            vi, vj = np.ascontiguousarray(vi, np.int32), np.ascontiguousarray(vj, np.int32)
            xi, xj = np.ascontiguousarray(xi, np.int32), np.ascontiguousarray(xj, np.int32)
            keep += [vi, vj, xi, xj]
            arr[k] = MacInputStruct(i1, j1, i2, j2, ns, len(xi), vi.ctypes.data, vj.ctypes.data,
                                    xi.ctypes.data if len(xi) else None, xj.ctypes.data if len(xj) else None)
        qr = np.ascontiguousarray(qranges, np.int32).reshape(-1)
        trg = np.ascontiguousarray(tranges, np.int32).reshape(-1)
        hits = np.zeros(n, dtype=MAC_HIT_DTYPE)
        h = C.c_void_p()
        self._chk(self.lib.hhv_mac_realign_hits(self.h, qp.ctypes.data, q_tr_lin.ctypes.data, Lq, n, Lt.ctypes.data, pp, tt,
                                             C.addressof(arr), len(qr) // 2, qr.ctypes.data if len(qr) else None,
                                             len(trg) // 2, trg.ctypes.data if len(trg) else None, int(local), shift, mact,
                                             C.byref(h), hits.ctypes.data))
        return MacSet(self.lib, h, hits, Lq, Lt)

    def db_open(self, path, Ls):
        h = C.c_void_p()
        self._chk(self.lib.hhv_db_open(self.h, path.encode(), C.byref(h)))
        return TemplateSet(self, h, Ls)

    def adopt_device_stream(self, Ls, device_ptr):
        Ls = np.ascontiguousarray(Ls, dtype=np.int32)
        h = C.c_void_p()
        self._chk(self.lib.hhv_adopt_device_stream(self.h, len(Ls), Ls.ctypes.data_as(c_int_p), C.c_void_p(device_ptr),
                                                C.byref(h)))
        return TemplateSet(self, h, Ls)

    def align(self, ts, backtrace=False, celloff=False):
        flags = (HHV_ALIGN_BACKTRACE if backtrace else 0) | (HHV_ALIGN_CELLOFF if celloff else 0)
        out = np.zeros(ts.n, dtype=RESULT_DTYPE)
        self._chk(self.lib.hhv_align(self.h, ts.h, flags, out.ctypes.data))
        return out

    def align_async(self, ts, backtrace=False, celloff=False, d_out=None):
        flags = (HHV_ALIGN_BACKTRACE if backtrace else 0) | (HHV_ALIGN_CELLOFF if celloff else 0)
        self._chk(self.lib.hhv_align_async(self.h, ts.h, flags, C.c_void_p(d_out) if d_out else None))

    def sync(self):
        self._chk(self.lib.hhv_sync(self.h))

    def stream(self):
        return self.lib.hhv_stream(self.h)

    def last_kernel_ms(self):
        ms = C.c_float()
        self._chk(self.lib.hhv_last_kernel_ms(self.h, C.byref(ms)))
        return ms.value

    def set_celloff(self, ts, k, mask):
        if mask is None:
            self._chk(self.lib.hhv_set_celloff(self.h, ts.h, int(k), None))
            return
        m = np.ascontiguousarray(mask, dtype=np.uint8)
        assert m.shape == (self.Lq + 1, int(ts.L[k]) + 1)
        self._chk(self.lib.hhv_set_celloff(self.h, ts.h, int(k), m.ctypes.data))

    @staticmethod
    def pack_celloff_paths(paths):
        """paths = [(template, nsteps, i_steps, j_steps)] with 1-based step arrays as hit_path returns them (entries 1..nsteps
        are handed over) -> the arrays hhv_set_celloff_paths takes: (template_of, path_off, i, j)"""
        template_of = np.array([p[0] for p in paths], dtype=np.int32)
        off = np.zeros(len(paths) + 1, dtype=np.int64)
        for k, p in enumerate(paths):
            off[k + 1] = off[k] + int(p[1])
        pi = np.concatenate([np.asarray(p[2], dtype=np.int32)[1:int(p[1]) + 1] for p in paths]) if paths else np.zeros(0, np.int32)
        pj = np.concatenate([np.asarray(p[3], dtype=np.int32)[1:int(p[1]) + 1] for p in paths]) if paths else np.zeros(0, np.int32)
        return template_of, off, np.ascontiguousarray(pi), np.ascontiguousarray(pj)

    def set_celloff_paths_packed(self, ts, packed, qranges=(), tranges=()):
        template_of, off, pi, pj = packed
        qr = np.ascontiguousarray(np.asarray(qranges, dtype=np.int32).reshape(-1))
        tr = np.ascontiguousarray(np.asarray(tranges, dtype=np.int32).reshape(-1))
        self._chk(self.lib.hhv_set_celloff_paths(self.h, ts.h, len(template_of), template_of.ctypes.data, off.ctypes.data,
                                                 pi.ctypes.data, pj.ctypes.data, len(qr) // 2, qr.ctypes.data if len(qr) else None,
                                                 len(tr) // 2, tr.ctypes.data if len(tr) else None))

    def set_celloff_paths(self, ts, paths, qranges=(), tranges=()):
        """hhv_set_celloff_paths: paths = [(template, nsteps, i_steps, j_steps)] with 1-based step arrays as hit_path returns
        them (entries 1..nsteps are handed over); qranges / tranges = [(lo, hi)] of -excl / -template_excl"""
        self.set_celloff_paths_packed(ts, self.pack_celloff_paths(paths), qranges, tranges)

    def set_global_batch(self, ts, not_longest):
        """hhv_set_global_batch: not_longest[k] != 0 -> template k is shorter than the longest of its SIMD batch"""
        if not_longest is None:
            self._chk(self.lib.hhv_set_global_batch(self.h, ts.h, None))
            return
        f = np.ascontiguousarray(not_longest, dtype=np.uint8)
        assert f.shape == (ts.n,)
        self._chk(self.lib.hhv_set_global_batch(self.h, ts.h, f.ctypes.data))

    def backtrace_matrix(self, ts, k):
        out = np.zeros((self.Lq + 1, int(ts.L[k]) + 1), dtype=np.uint8)
        self._chk(self.lib.hhv_backtrace_matrix(self.h, ts.h, int(k), out.ctypes.data))
        return out

    def hits(self, ts, fetch=True):
        out = np.zeros(ts.n, dtype=HIT_DTYPE) if fetch else None
        self._chk(self.lib.hhv_hits(self.h, ts.h, out.ctypes.data if fetch else None))
        return out

    def backtrace(self, ts, k):
        """hhv_backtrace: (nsteps, matched_cols, i_steps, j_steps, states) of template k - Viterbi::Backtrace's BacktraceResult"""
        cap = self.Lq + int(ts.L[k]) + 2
        i_steps = np.zeros(cap, dtype=np.int32)
        j_steps = np.zeros(cap, dtype=np.int32)
        states = np.zeros(cap, dtype=np.int8)
        ns, mc = C.c_int32(), C.c_int32()
        self._chk(self.lib.hhv_backtrace(self.h, ts.h, int(k), cap, i_steps.ctypes.data_as(C.c_void_p), j_steps.ctypes.data_as(C.c_void_p),
                                      states.ctypes.data_as(C.c_void_p), C.byref(ns), C.byref(mc)))
        return ns.value, mc.value, i_steps, j_steps, states

    def hit_path(self, ts, k):
        cap = self.Lq + int(ts.L[k]) + 2
        i_steps = np.zeros(cap, dtype=np.int32)
        j_steps = np.zeros(cap, dtype=np.int32)
        states = np.zeros(cap, dtype=np.int8)
        S = np.zeros(cap, dtype=np.float32)
        ns = C.c_int32()
        self._chk(self.lib.hhv_hit_path(self.h, ts.h, int(k), cap, i_steps.ctypes.data, j_steps.ctypes.data,
                                     states.ctypes.data, S.ctypes.data, C.byref(ns)))
        return ns.value, i_steps, j_steps, states, S

    def hit_path_pool(self, ts):
        """hhv_hit_path_pool: (path_off[n+1], i_steps, j_steps, states, S) of all templates as numpy copies of the
        library's host mirror (entry path_off[k] + s = step s of template k, s = 1..nsteps)."""
        ptrs = [C.c_void_p() for _ in range(5)]
        self._chk(self.lib.hhv_hit_path_pool(self.h, ts.h, *[C.byref(p) for p in ptrs]))
        n = ts.n
        off = np.ctypeslib.as_array(C.cast(ptrs[0], C.POINTER(C.c_int64)), shape=(n + 1,)).copy()
        tot = int(off[n])
        i_steps = np.ctypeslib.as_array(C.cast(ptrs[1], C.POINTER(C.c_int32)), shape=(tot,)).copy()
        j_steps = np.ctypeslib.as_array(C.cast(ptrs[2], C.POINTER(C.c_int32)), shape=(tot,)).copy()
        states = np.ctypeslib.as_array(C.cast(ptrs[3], C.POINTER(C.c_int8)), shape=(tot,)).copy()
        S = np.ctypeslib.as_array(C.cast(ptrs[4], C.POINTER(C.c_float)), shape=(tot,)).copy()
        return off, i_steps, j_steps, states, S

    def hit_paths_packed(self, ts, hits):
        """hhv_hit_paths_packed: (off[n+1], i_steps u16, j_steps u16, states, S) - the paths of hhv_hits without the pool's unused
        capacity (entry off[k] + s = step s of template k, s = 0..nsteps, entry 0 all zero); numpy copies of the pinned buffer."""
        ptrs = [C.c_void_p() for _ in range(5)]
        self.lib.hhv_hit_paths_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p] + [C.POINTER(C.c_void_p)] * 5
        self._chk(self.lib.hhv_hit_paths_packed(self.h, ts.h, hits.ctypes.data, *[C.byref(p) for p in ptrs]))
        n = ts.n
        off = np.ctypeslib.as_array(C.cast(ptrs[0], C.POINTER(C.c_int64)), shape=(n + 1,)).copy()
        tot = int(off[n])
        i_steps = np.ctypeslib.as_array(C.cast(ptrs[1], C.POINTER(C.c_uint16)), shape=(tot,)).copy()
        j_steps = np.ctypeslib.as_array(C.cast(ptrs[2], C.POINTER(C.c_uint16)), shape=(tot,)).copy()
        states = np.ctypeslib.as_array(C.cast(ptrs[3], C.POINTER(C.c_int8)), shape=(tot,)).copy()
        S = np.ctypeslib.as_array(C.cast(ptrs[4], C.POINTER(C.c_float)), shape=(tot,)).copy()
        return off, i_steps, j_steps, states, S

    def set_neff(self, ts, q_neff, t_neff):
        """hhv_tset_set_neff: the diversities the reference's ranking key needs (topk(..., pvalue=True))"""
        t = _f32(t_neff)
        if t.shape[0] != ts.n:
            raise HhvError("set_neff: %d values for %d templates" % (t.shape[0], ts.n))
        self.lib.hhv_tset_set_neff.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
        self._chk(self.lib.hhv_tset_set_neff(self.h, ts.h, float(q_neff), t.ctypes.data))

    def topk(self, ts, k, d_out=None, fetch=True, raw=False, pvalue=False):
        out = np.zeros(k, dtype=HIT_DTYPE) if fetch else None
        n = C.c_int32()
        self._chk(self.lib.hhv_topk(self.h, ts.h, int(k), (1 if raw else 0) | (2 if pvalue else 0), out.ctypes.data if fetch else None,
                                 C.c_void_p(d_out) if d_out else None, C.byref(n)))
        return (out[:n.value] if fetch else None), n.value

    def set_global_ids(self, ts, ids):
        """ids: global template id of every entry of the shard (None = back to set indices); hhv_topk then reports them"""
        if ids is None:
            self._chk(self.lib.hhv_tset_set_global_ids(self.h, ts.h, None))
            return
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        if ids.shape[0] != ts.n:
            raise HhvError("set_global_ids: %d ids for %d templates" % (ids.shape[0], ts.n))
        self._chk(self.lib.hhv_tset_set_global_ids(self.h, ts.h, ids.ctypes.data_as(C.c_void_p)))

    def merge_hits(self, d_in, m, k, d_out=None, fetch=True, count=True):
        """d_in: device pointer to m hhv_hit records (gathered top-K lists, global ids) -> the k best, merged on the device.
        fetch=False, count=False: the merge is only enqueued on the context's stream (no host wait); returns (None, None)"""
        out = np.zeros(k, dtype=HIT_DTYPE) if fetch else None
        n = C.c_int32()
        want_n = fetch or count
        self._chk(self.lib.hhv_merge_hits(self.h, C.c_void_p(d_in), int(m), int(k), out.ctypes.data if fetch else None,
                                       C.c_void_p(d_out) if d_out else None, C.byref(n) if want_n else None))
        return (out[:n.value] if fetch else None), (n.value if want_n else None)


# ---- C++ runner (hh-suite_amd/host/viterbi_runner.cpp) through its C shim --------------------------
RUNNER_PATH = os.path.join(os.path.dirname(HERE), "lib", "libhhv_runner.so")
RUNNER_HIT_DTYPE = np.dtype([("entry", np.int32), ("irep", np.int32), ("lastrep", np.int32), ("score", np.float32),
                             ("i1", np.int32), ("j1", np.int32), ("i2", np.int32), ("j2", np.int32),
                             ("nsteps", np.int32), ("matched_cols", np.int32)])
_runner = None


def load_runner():
    global _runner
    if _runner is None:
        if not os.path.exists(RUNNER_PATH):
            raise HhvError("libhhv_runner.so not built (%s)" % RUNNER_PATH)
        load()
        _runner = C.CDLL(RUNNER_PATH)
        _runner.hhvr_alignment.argtypes = [C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                           C.c_int, C.c_int, C.c_float, C.c_char_p, C.c_char_p, c_float_p, c_float_p,
                                           C.c_int, C.c_int,
                                           c_int_p, C.POINTER(c_float_p), C.POINTER(c_float_p), C.c_void_p, C.c_int,
                                           C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    _runner.hhvr_linear_transitions.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    _runner.hhvr_linear_transitions.restype = None
    _runner.hhvr_mac_celloff.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_char_p, C.c_char_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    _runner.hhvr_mac_realign.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_float, C.c_int32, C.c_char_p, C.c_char_p,
                                         C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    _runner.hhvr_prefilter_db.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    _runner.hhvr_prefilter_select_first.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    _runner.hhvr_prefilter_select_second.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32,
                                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    _runner.hhvr_prefilter_profile.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    _runner.hhvr_read_context_library.argtypes = [C.c_char_p, C.c_void_p]
    _runner.hhvr_flog2.argtypes = [C.c_float]
    _runner.hhvr_flog2.restype = C.c_float
    _runner.hhvr_fpow2.argtypes = [C.c_float]
    _runner.hhvr_fpow2.restype = C.c_float
    return _runner


def runner_alignment(qp, qtr, tps, ttrs, loc=1, egq=0.0, egt=0.0, shift=-0.03, corr=0.1, ssw=0.11, ssm=2, altali=4,
                     smin=20.0, device=0, exclstr=None, template_exclstr=None):
    """hhv::ViterbiRunner::alignment -> (hits structured array, i_steps, j_steps, states, S) with one row per hit."""
    lib = load_runner()
    qp, qtr = _f32(qp), _f32(qtr)
    tps = [_f32(a) for a in tps]
    ttrs = [_f32(a) for a in ttrs]
    n = len(tps)
    Lq = qp.shape[0] - 1
    Ls = np.array([a.shape[0] - 1 for a in tps], dtype=np.int32)
    pp = (c_float_p * n)(*[a.ctypes.data_as(c_float_p) for a in tps])
    tt = (c_float_p * n)(*[a.ctypes.data_as(c_float_p) for a in ttrs])
    cap = n * altali
    pcap = Lq + int(Ls.max()) + 2
    hits = np.zeros(cap, dtype=RUNNER_HIT_DTYPE)
    i_s = np.zeros((cap, pcap), dtype=np.int32)
    j_s = np.zeros((cap, pcap), dtype=np.int32)
    st = np.zeros((cap, pcap), dtype=np.int8)
    S = np.zeros((cap, pcap), dtype=np.float32)
    m = lib.hhvr_alignment(device, loc, egq, egt, shift, corr, ssw, ssm, altali, smin,
                           exclstr.encode() if exclstr else None,
                           template_exclstr.encode() if template_exclstr else None, qp.ctypes.data_as(c_float_p),
                           qtr.ctypes.data_as(c_float_p), Lq, n, Ls.ctypes.data_as(c_int_p), pp, tt, hits.ctypes.data,
                           cap, pcap, i_s.ctypes.data, j_s.ctypes.data, st.ctypes.data, S.ctypes.data)
    if m < 0:
        raise HhvError("hhvr_alignment failed: %d: %s" % (m, load().hhv_last_error().decode()))
    return hits[:m], i_s[:m], j_s[:m], st[:m], S[:m]


class MacInputStruct(C.Structure):
    _fields_ = [("i1", C.c_int32), ("j1", C.c_int32), ("i2", C.c_int32), ("j2", C.c_int32), ("nsteps", C.c_int32),
                ("n_excluded", C.c_int32), ("i", C.c_void_p), ("j", C.c_void_p), ("excluded_i", C.c_void_p),
                ("excluded_j", C.c_void_p)]


MAC_HIT_DTYPE = np.dtype([("Pforward", "<f8"), ("sum_of_probs", "<f4"), ("i1", "<i4"), ("j1", "<i4"), ("i2", "<i4"),
                          ("j2", "<i4"), ("nsteps", "<i4"), ("matched_cols", "<i4"), ("reserved", "<i4")])


class MacSet:
    """Result of hhv_mac_realign: per-hit summaries on the host, paths and posterior matrices on the device."""

    def __init__(self, lib, h, hits, Lq, Lt):
        self.lib, self.h, self.hits, self.Lq, self.Lt = lib, h, hits, Lq, Lt

    def _chk(self, rc):
        if rc != 0:
            raise HhvError("hhv error %d: %s" % (rc, self.lib.hhv_last_error().decode()))

    def path(self, k):
        cap = int(self.hits["nsteps"][k]) + 1
        i_s = np.zeros(cap, np.int32)
        j_s = np.zeros(cap, np.int32)
        st = np.zeros(cap, np.int8)
        S = np.zeros(cap, np.float32)
        P = np.zeros(cap, np.float32)
        ns = C.c_int32()
        self._chk(self.lib.hhv_mac_path(self.h, k, cap, i_s.ctypes.data, j_s.ctypes.data, st.ctypes.data, S.ctypes.data,
                                     P.ctypes.data, C.byref(ns)))
        return i_s, j_s, st, S, P

    def celloff(self, k):
        out = np.zeros((self.Lq + 1, int(self.Lt[k]) + 1), np.uint8)
        self._chk(self.lib.hhv_mac_celloff(self.h, k, out.ctypes.data))
        return out

    def posterior(self, k):
        out = np.zeros((self.Lq + 1, int(self.Lt[k]) + 1), np.float32)
        self._chk(self.lib.hhv_mac_posterior(self.h, k, out.ctypes.data))
        return out

    def list(self, k, which):
        """hhv_mac_list: the reference's sparse forward (0) / backward (1) / posterior (2) list of hit k -> (i, j, value)"""
        n = self.lib.hhv_mac_list(self.h, k, which, 0, None, None, None)
        if n < 0:
            _check(int(n))
        li, lj, lv = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.float32)
        if n:
            m = self.lib.hhv_mac_list(self.h, k, which, n, li.ctypes.data, lj.ctypes.data, lv.ctypes.data)
            assert m == n, (m, n)
        return li, lj, lv

    def free(self):
        if self.h:
            self.lib.hhv_macset_free(self.h)
            self.h = None


# ---- hhv::Prefilter (host/prefilter.h): HHblits prefilter around the two GPU kernels -------------------------
PREFILTER_DEFAULTS = dict(gap_open=20, gap_extend=4, score_offset=50, bit_factor=4, smax_thresh=10, min_hits=100,
                          maxnumdb=20000, evalue_thresh=1000.0, evalue_coarse_thresh=100000.0)


def read_context_library(path):
    """hhv::ReadContextLibrary -> (219, 20) float64 central-column probabilities of a cs219 library file."""
    out = np.zeros((219, 20), dtype=np.float64)
    if load_runner().hhvr_read_context_library(str(path).encode(), out.ctypes.data) != 219:
        raise HhvError("cannot read context library %s" % path)
    return out


def prefilter_profile(q_p, pav, lib, score_offset=50, bit_factor=4):
    """hhv::PrefilterQueryProfile: q_p (Lq,20) rows p[0..Lq-1], pav (20,), lib (219,20) -> (220, Lq) uint8."""
    q_p, pav = _f32(q_p), _f32(pav)
    lib = np.ascontiguousarray(lib, dtype=np.float64)
    Lq = q_p.shape[0]
    out = np.zeros((220, Lq), dtype=np.uint8)
    load_runner().hhvr_prefilter_profile(q_p.ctypes.data, pav.ctypes.data, lib.ctypes.data, Lq, score_offset, bit_factor,
                                         out.ctypes.data)
    return out


def _prefilter_pars(kw):
    par = dict(PREFILTER_DEFAULTS)
    par.update(kw)
    ipar = np.array([par[k] for k in ("gap_open", "gap_extend", "score_offset", "bit_factor", "smax_thresh", "min_hits",
                                      "maxnumdb")], dtype=np.int32)
    dpar = np.array([par["evalue_thresh"], par["evalue_coarse_thresh"]], dtype=np.float64)
    return ipar, dpar


def prefilter_select_first(ungapped, lengths, Lq, **kw):
    """hhv::Prefilter::SelectFirst (host-only arithmetic): kernel scores -> ids that go on to Smith-Waterman."""
    ipar, dpar = _prefilter_pars(kw)
    ungapped = np.ascontiguousarray(ungapped, dtype=np.int32)
    lengths = np.ascontiguousarray(lengths, dtype=np.int32)
    out = np.zeros(len(lengths), dtype=np.int32)
    m = load_runner().hhvr_prefilter_select_first(ungapped.ctypes.data, lengths.ctypes.data, len(lengths), Lq, ipar.ctypes.data,
                                                  dpar.ctypes.data, out.ctypes.data)
    return out[:m]


def prefilter_select_second(sw, subset, lengths, Lq, **kw):
    """hhv::Prefilter::SelectSecond (host-only arithmetic): SW scores of subset -> (selected ids, e-values)."""
    ipar, dpar = _prefilter_pars(kw)
    sw = np.ascontiguousarray(sw, dtype=np.int32)
    subset = np.ascontiguousarray(subset, dtype=np.int32)
    lengths = np.ascontiguousarray(lengths, dtype=np.int32)
    ids = np.zeros(len(subset), dtype=np.int32)
    ev = np.zeros(len(subset), dtype=np.float64)
    m = load_runner().hhvr_prefilter_select_second(sw.ctypes.data, subset.ctypes.data, len(subset), lengths.ctypes.data,
                                                   len(lengths), Lq, ipar.ctypes.data, dpar.ctypes.data, ids.ctypes.data,
                                                   ev.ctypes.data)
    return ids[:m], ev[:m]


def prefilter_db(ctx, seqs, offsets, lib, q_p, pav, **kw):
    """hhv::Prefilter::prefilter_db -> (selected ids in reference order, e-values, number that passed stage 1)."""
    ipar, dpar = _prefilter_pars(kw)
    seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    lib = np.ascontiguousarray(lib, dtype=np.float64)
    q_p, pav = _f32(q_p), _f32(pav)
    n_db = len(offsets) - 1
    ids = np.zeros(n_db, dtype=np.int32)
    ev = np.zeros(n_db, dtype=np.float64)
    p1 = C.c_int32(0)
    m = load_runner().hhvr_prefilter_db(ctx.h, n_db, seqs.ctypes.data, offsets.ctypes.data, lib.ctypes.data, q_p.ctypes.data,
                                        pav.ctypes.data, q_p.shape[0], ipar.ctypes.data, dpar.ctypes.data, ids.ctypes.data,
                                        ev.ctypes.data, n_db, C.byref(p1))
    if m < 0:
        raise HhvError("hhvr_prefilter_db failed: %d: %s" % (m, load().hhv_last_error().decode()))
    return ids[:m], ev[:m], p1.value


# ---- hhv::PosteriorDecoderRunner (host/posterior_decoder.h): MAC realignment above hhv_mac_realign ----------------
def linear_transitions(tr_log2, is_query):
    """hhv::LinearTransitions: 2^tr (powf) + the boundary assignments of the realign stage."""
    tr = _f32(tr_log2)
    out = np.zeros_like(tr)
    load_runner().hhvr_linear_transitions(tr.ctypes.data, tr.shape[0] - 1, int(bool(is_query)), out.ctypes.data)
    return out


def _alt_lists(prev):
    off = np.zeros(len(prev) + 1, np.int32)
    for k, (a, _) in enumerate(prev):
        off[k + 1] = off[k] + len(a)
    pi = np.concatenate([np.asarray(a, np.int32) for a, _ in prev] + [np.zeros(0, np.int32)]).astype(np.int32)
    pj = np.concatenate([np.asarray(b, np.int32) for _, b in prev] + [np.zeros(0, np.int32)]).astype(np.int32)
    return off, np.ascontiguousarray(pi), np.ascontiguousarray(pj)


def mac_celloff(Lq, Lt, hit, prev=(), min_overlap=0, exclstr=None, template_exclstr=None):
    """hhv::MacCellOff. hit = (i1, j1, i2, j2, nsteps, i_steps, j_steps); prev = [(alt_i, alt_j), ...]."""
    i1, j1, i2, j2, ns, vi, vj = hit
    row = np.array([0, 1, i1, j1, i2, j2, ns], np.int32)
    vi = np.ascontiguousarray(vi, np.int32)
    vj = np.ascontiguousarray(vj, np.int32)
    off, pi, pj = _alt_lists(list(prev))
    mask = np.zeros((Lq + 1, Lt + 1), np.uint8)
    load_runner().hhvr_mac_celloff(Lq, Lt, min_overlap, exclstr.encode() if exclstr else None,
                                   template_exclstr.encode() if template_exclstr else None, row.ctypes.data, vi.ctypes.data,
                                   vj.ctypes.data, len(off) - 1, off.ctypes.data, pi.ctypes.data, pj.ctypes.data,
                                   mask.ctypes.data)
    return mask


def runner_mac_realign(ctx, qp, q_tr_lin, tps, t_trs, hits, loc=1, shift=-0.03, mact=0.3501, min_overlap=0, resident=None):
    """hhv::PosteriorDecoderRunner::executeComputation. hits: list of (entry, irep, i1, j1, i2, j2, nsteps, i_steps, j_steps).
    Returns (scalars[n][6] = nsteps,i1,j1,i2,j2,matched_cols; real[n][2] = Pforward,sum_of_probs; i, j, states, S, P)."""
    lib = load_runner()
    qp, q_tr_lin = _f32(qp), _f32(q_tr_lin)
    t_trs = [_f32(a) for a in t_trs]
    nt, nh = len(t_trs), len(hits)
    Lq = qp.shape[0] - 1
    Lt = np.array([a.shape[0] - 1 for a in t_trs], dtype=np.int32)
    if resident is None:
        tps = [_f32(a) for a in tps]
        pp = (C.c_void_p * nt)(*[a.ctypes.data for a in tps])
    else:
        pp = None   # profiles are read from the resident template set; hits' entry = index in it
    tt = (C.c_void_p * nt)(*[a.ctypes.data for a in t_trs])
    # optional 10th element of a hit: its template's index in the resident set when `entry` indexes a compact t_trs list
    rows = np.array([list(h[:7]) + [h[9] if len(h) > 9 else -1] for h in hits], dtype=np.int32).reshape(nh, 8)
    poff = np.zeros(nh + 1, np.int64)
    for k, h in enumerate(hits):          # nsteps < 0: resident hit, no path handed over
        poff[k + 1] = poff[k] + max(h[6], 0) + 1
    pi = np.zeros(poff[-1], np.int32)
    pj = np.zeros(poff[-1], np.int32)
    for k, h in enumerate(hits):
        if h[6] >= 0:
            pi[poff[k]:poff[k + 1]] = np.asarray(h[7], np.int32)[:h[6] + 1]
            pj[poff[k]:poff[k + 1]] = np.asarray(h[8], np.int32)[:h[6] + 1]
    pcap = Lq + int(Lt.max()) + 2
    sc = np.zeros((nh, 6), np.int32)
    re = np.zeros((nh, 2), np.float64)
    o_i = np.zeros((nh, pcap), np.int32)
    o_j = np.zeros((nh, pcap), np.int32)
    o_s = np.zeros((nh, pcap), np.int8)
    o_S = np.zeros((nh, pcap), np.float32)
    o_P = np.zeros((nh, pcap), np.float32)
    m = lib.hhvr_mac_realign(ctx.h, resident.h if resident is not None else None, loc, shift, mact, min_overlap, None, None, qp.ctypes.data, q_tr_lin.ctypes.data, Lq, nt,
                             Lt.ctypes.data, pp, tt, nh, rows.ctypes.data, poff.ctypes.data, pi.ctypes.data, pj.ctypes.data,
                             sc.ctypes.data, re.ctypes.data, pcap, o_i.ctypes.data, o_j.ctypes.data, o_s.ctypes.data,
                             o_S.ctypes.data, o_P.ctypes.data)
    if m < 0:
        raise HhvError("hhvr_mac_realign failed: %d: %s" % (m, load().hhv_last_error().decode()))
    return sc, re, o_i, o_j, o_s, o_S, o_P
