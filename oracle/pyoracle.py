"""ctypes loaders for the two CPU checkers -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

  Oracle  -> oracle/liboracle.so      (plain-C restatement, oracle/hhv_oracle.c)
  Ref     -> oracle/_ref/libhhref.so  (the reference's own translation units + ref_harness.cpp)
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "libhhref.so")

c_float_p = C.POINTER(C.c_float)
c_int_p = C.POINTER(C.c_int)
c_ubyte_p = C.POINTER(C.c_ubyte)
c_byte_p = C.POINTER(C.c_byte)


def _fp(a):
    return a.ctypes.data_as(c_float_p)


def _ip(a):
    return a.ctypes.data_as(c_int_p)


def _ubp(a):
    return a.ctypes.data_as(c_ubyte_p)


def _bp(a):
    return a.ctypes.data_as(c_byte_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _hits_buffers(N):
    return {"score": np.zeros(N, dtype=np.float32), "i2": np.zeros(N, dtype=np.int32), "j2": np.zeros(N, dtype=np.int32),
            "i1": np.zeros(N, dtype=np.int32), "j1": np.zeros(N, dtype=np.int32), "nsteps": np.zeros(N, dtype=np.int32),
            "matched_cols": np.zeros(N, dtype=np.int32), "hit_score": np.zeros(N, dtype=np.float32),
            "path_hash": np.zeros(N, dtype=np.uint64), "s_hash": np.zeros(N, dtype=np.uint64)}


def _hits_args(o):
    u64 = lambda a: a.ctypes.data_as(C.POINTER(C.c_ulonglong))
    return [_fp(o["score"]), _ip(o["i2"]), _ip(o["j2"]), _ip(o["i1"]), _ip(o["j1"]), _ip(o["nsteps"]),
            _ip(o["matched_cols"]), _fp(o["hit_score"]), u64(o["path_hash"]), u64(o["s_hash"])]


def path_hashes(path_off, i_steps, j_steps, states, S, nsteps):
    """The two checksums of ref_bench_hits / hho_bench_hits computed from a path pool (arrays concatenated per template at
    path_off[k], 1-based steps): returns (path_hash, s_hash) as uint64 arrays."""
    n = len(nsteps)
    ph = np.zeros(n, dtype=np.uint64)
    sh = np.zeros(n, dtype=np.uint64)
    nsteps = np.asarray(nsteps, dtype=np.int64)
    total = int(nsteps.sum())
    if total == 0:
        return ph, sh
    starts = np.asarray(path_off[:n], dtype=np.int64) + 1
    seg = np.repeat(np.arange(n), nsteps)
    first = np.cumsum(nsteps) - nsteps
    step = np.arange(total, dtype=np.int64) - np.repeat(first, nsteps) + 1
    idx = np.repeat(starts, nsteps) + step - 1
    w = (2 * step + 1).astype(np.uint64)
    with np.errstate(over="ignore"):
        a = (np.asarray(i_steps)[idx].astype(np.uint64) * np.uint64(1000003) + np.asarray(j_steps)[idx].astype(np.uint64)) * np.uint64(31) \
            + (np.asarray(states)[idx].astype(np.uint8)).astype(np.uint64)
        a = a * w
        b = np.asarray(S, dtype=np.float32)[idx].view(np.uint32).astype(np.uint64) * w
        nz = nsteps > 0
        ph[nz] = np.add.reduceat(a, first[nz])
        sh[nz] = np.add.reduceat(b, first[nz])
    return ph, sh


class HhoParams(C.Structure):
    _fields_ = [("local", C.c_int), ("egq", C.c_float), ("egt", C.c_float), ("shift", C.c_float),
                ("corr", C.c_float), ("ssw", C.c_float), ("ss_mode", C.c_int)]


class HhoSS(C.Structure):
    _fields_ = [("ss_hmm_mode", C.c_int),
                ("q_ss_pred", c_byte_p), ("q_ss_conf", c_byte_p), ("q_ss_dssp", c_byte_p),
                ("t_ss_pred", c_byte_p), ("t_ss_conf", c_byte_p), ("t_ss_dssp", c_byte_p),
                ("S73", c_float_p), ("S33", c_float_p), ("S37", c_float_p)]


def make_params(local=0, egq=0.0, egt=0.0, shift=-0.03, corr=0.1, ssw=0.11, ss_mode=2):
    """Defaults of the reference: src/hhdecl.cpp:86-98 (shift, corr, egq, egt, ssw, ssm)."""
    return dict(local=int(local), egq=float(egq), egt=float(egt), shift=float(shift), corr=float(corr),
                ssw=float(ssw), ss_mode=int(ss_mode))


class SSInfo:
    """Secondary-structure inputs for the ...AndSS variants (all int8 arrays of length L+1)."""

    def __init__(self, mode, q_pred, q_conf, q_dssp, S73, S33, S37):
        self.mode = int(mode)
        self.q_pred = np.ascontiguousarray(q_pred, dtype=np.int8)
        self.q_conf = np.ascontiguousarray(q_conf, dtype=np.int8)
        self.q_dssp = np.ascontiguousarray(q_dssp, dtype=np.int8)
        self.S73 = _f32(S73).reshape(8, 4, 11)
        self.S33 = _f32(S33).reshape(4, 11, 4, 11)
        self.S37 = _f32(S37).reshape(4, 11, 8)


class AlignOut:
    pass


class Oracle:
    def __init__(self, path=ORACLE_SO):
        self.lib = C.CDLL(path)
        L = self.lib
        L.hho_log2f4.restype = C.c_float
        L.hho_log2f4.argtypes = [C.c_float]
        L.hho_fast_log2.restype = C.c_float
        L.hho_fast_log2.argtypes = [C.c_float]
        L.hho_dot20_vec.restype = C.c_float
        L.hho_dot20_vec.argtypes = [c_float_p, c_float_p]
        L.hho_dot20_scalar.restype = C.c_float
        L.hho_dot20_scalar.argtypes = [c_float_p, c_float_p]
        L.hho_align.restype = C.c_int
        L.hho_align.argtypes = [C.POINTER(HhoParams), c_float_p, c_float_p, C.c_int, c_float_p, c_float_p, C.c_int,
                                C.c_int, c_ubyte_p, C.POINTER(HhoSS), c_float_p, c_int_p, c_int_p, c_ubyte_p]
        L.hho_backtrace.restype = C.c_int
        L.hho_backtrace.argtypes = [c_ubyte_p, C.c_int, C.c_int, C.c_int, c_int_p, c_int_p, c_byte_p, C.c_int,
                                    c_int_p, c_int_p]
        L.hho_score_for_backtrace.restype = C.c_int
        L.hho_score_for_backtrace.argtypes = [C.POINTER(HhoParams), c_float_p, c_float_p, C.POINTER(HhoSS), c_int_p,
                                              c_int_p, c_byte_p, C.c_int, C.c_float, c_float_p, c_float_p, c_float_p]
        L.hho_exclude_alignment.restype = C.c_int
        L.hho_exclude_alignment.argtypes = [C.c_int, C.c_int, c_int_p, c_int_p, C.c_int, c_ubyte_p]
        L.hho_bench_align.restype = C.c_double
        L.hho_bench_align.argtypes = [C.POINTER(HhoParams), c_float_p, c_float_p, C.c_int, C.c_int, c_int_p,
                                      C.POINTER(c_float_p), C.POINTER(c_float_p), C.c_int, c_float_p, c_int_p,
                                      c_int_p]

    # -- unit probes
    def set_emission_mode(self, mode):
        """0 = the reference's arithmetic (default); 2 = the engine's opt-in fused build (hho_set_emission_mode).  Process-wide:
        callers put it back to 0."""
        self.lib.hho_set_emission_mode(int(mode))

    def log2f4(self, x):
        return self.lib.hho_log2f4(C.c_float(x))

    def fast_log2(self, x):
        return self.lib.hho_fast_log2(C.c_float(x))

    def dot20_vec(self, q, t):
        q, t = _f32(q), _f32(t)
        return self.lib.hho_dot20_vec(_fp(q), _fp(t))

    def dot20_scalar(self, q, t):
        q, t = _f32(q), _f32(t)
        return self.lib.hho_dot20_scalar(_fp(q), _fp(t))

    @staticmethod
    def _ss_struct(ss, t_ss):
        if ss is None:
            return None, None
        tp, tc, td = (np.ascontiguousarray(a, dtype=np.int8) for a in t_ss)
        s = HhoSS(ss.mode, _bp(ss.q_pred), _bp(ss.q_conf), _bp(ss.q_dssp), _bp(tp), _bp(tc), _bp(td), _fp(ss.S73),
                  _fp(ss.S33), _fp(ss.S37))
        return s, (tp, tc, td)

    def align(self, par, qp, qtr, tp, ttr, Lbatch=None, celloff=None, ss=None, t_ss=None, want_bt=True,
              want_path=False):
        """One pair through the restatement; returns AlignOut(score, i2, j2, bt, [path fields])."""
        qp, qtr, tp, ttr = _f32(qp), _f32(qtr), _f32(tp), _f32(ttr)
        Lq, Lt = qp.shape[0] - 1, tp.shape[0] - 1
        Lb = Lt if Lbatch is None else int(Lbatch)
        P = HhoParams(**par)
        sst, keep = self._ss_struct(ss, t_ss)
        score = C.c_float()
        i2, j2 = C.c_int(), C.c_int()
        bt = np.zeros((Lq + 1, Lb + 1), dtype=np.uint8) if (want_bt or want_path) else None
        co = None
        if celloff is not None:
            co = np.ascontiguousarray(celloff, dtype=np.uint8)
            assert co.shape == (Lq + 1, Lt + 1)
        rc = self.lib.hho_align(C.byref(P), _fp(qp), _fp(qtr), Lq, _fp(tp), _fp(ttr), Lt, Lb,
                                _ubp(co) if co is not None else None, C.byref(sst) if sst is not None else None,
                                C.byref(score), C.byref(i2), C.byref(j2), _ubp(bt) if bt is not None else None)
        assert rc == 0, rc
        o = AlignOut()
        o.score, o.i2, o.j2, o.bt = np.float32(score.value), i2.value, j2.value, bt
        if want_path:
            cap = o.i2 + o.j2 + 2
            o.i_steps = np.zeros(cap, dtype=np.int32)
            o.j_steps = np.zeros(cap, dtype=np.int32)
            o.states = np.zeros(cap, dtype=np.int8)
            ns, mc = C.c_int(), C.c_int()
            rc = self.lib.hho_backtrace(_ubp(bt), Lb + 1, o.i2, o.j2, _ip(o.i_steps), _ip(o.j_steps), _bp(o.states),
                                        cap, C.byref(ns), C.byref(mc))
            assert rc == 0, rc
            o.nsteps, o.matched_cols = ns.value, mc.value
            o.S = np.zeros(cap, dtype=np.float32)
            hs, sss = C.c_float(), C.c_float()
            self.lib.hho_score_for_backtrace(C.byref(P), _fp(qp), _fp(tp), C.byref(sst) if sst is not None else None,
                                             _ip(o.i_steps), _ip(o.j_steps), _bp(o.states), o.nsteps,
                                             C.c_float(float(o.score)), _fp(o.S), C.byref(hs), C.byref(sss))
            o.hit_score, o.score_ss = np.float32(hs.value), np.float32(sss.value)
        return o

    def exclude_alignment(self, Lq, Lt, i_steps, j_steps, nsteps, mask=None):
        if mask is None:
            mask = np.zeros((Lq + 1, Lt + 1), dtype=np.uint8)
        i_steps = np.ascontiguousarray(i_steps, dtype=np.int32)
        j_steps = np.ascontiguousarray(j_steps, dtype=np.int32)
        self.lib.hho_exclude_alignment(Lq, Lt, _ip(i_steps), _ip(j_steps), int(nsteps), _ubp(mask))
        return mask

    def bench_align(self, par, qp, qtr, tps, ttrs, threads=1):
        qp, qtr = _f32(qp), _f32(qtr)
        tps = [_f32(a) for a in tps]
        ttrs = [_f32(a) for a in ttrs]
        N = len(tps)
        L = np.array([a.shape[0] - 1 for a in tps], dtype=np.int32)
        pp = (c_float_p * N)(*[_fp(a) for a in tps])
        tt = (c_float_p * N)(*[_fp(a) for a in ttrs])
        score = np.zeros(N, dtype=np.float32)
        i2 = np.zeros(N, dtype=np.int32)
        j2 = np.zeros(N, dtype=np.int32)
        P = HhoParams(**par)
        sec = self.lib.hho_bench_align(C.byref(P), _fp(qp), _fp(qtr), qp.shape[0] - 1, N, _ip(L), pp, tt,
                                       int(threads), _fp(score), _ip(i2), _ip(j2))
        return sec, score, i2, j2


    def bench_hits(self, par, qp, qtr, tps, ttrs, threads=1, replicate=True):
        """Align + Backtrace + ScoreForBacktrace for every template (single-length batches).  Returns a dict of arrays:
        sec, score, i2, j2, i1, j1, nsteps, matched_cols, hit_score, path_hash, s_hash (see oracle/ref_harness.cpp)."""
        qp, qtr = _f32(qp), _f32(qtr)
        tps = [_f32(a) for a in tps]
        ttrs = [_f32(a) for a in ttrs]
        N = len(tps)
        L = np.array([a.shape[0] - 1 for a in tps], dtype=np.int32)
        pp = (c_float_p * N)(*[_fp(a) for a in tps])
        tt = (c_float_p * N)(*[_fp(a) for a in ttrs])
        o = _hits_buffers(N)
        P = HhoParams(**par)
        self.lib.hho_bench_hits.restype = C.c_double
        o["sec"] = self.lib.hho_bench_hits(C.byref(P), _fp(qp), _fp(qtr), qp.shape[0] - 1, N, _ip(L), pp, tt, int(threads),
                                           *_hits_args(o))
        return o


class Ref:
    """The reference itself (Viterbi::Align & co) behind oracle/ref_harness.cpp."""

    def __init__(self, path=REF_SO):
        self.lib = C.CDLL(path)
        L = self.lib
        L.ref_vecsize.restype = C.c_int
        for n in ("ref_log2f4", "ref_fast_log2"):
            getattr(L, n).restype = C.c_float
            getattr(L, n).argtypes = [C.c_float]
        for n in ("ref_scalarprod20", "ref_scalarprod20vec"):
            getattr(L, n).restype = C.c_float
            getattr(L, n).argtypes = [c_float_p, c_float_p]
        L.ref_create.restype = C.c_void_p
        L.ref_create.argtypes = [C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_float,
                                 c_float_p, c_float_p, c_float_p]
        L.ref_destroy.argtypes = [C.c_void_p]
        L.ref_set_query.restype = C.c_int
        L.ref_set_query.argtypes = [C.c_void_p, c_float_p, c_float_p, C.c_int, c_byte_p, c_byte_p, c_byte_p]
        L.ref_align_batch.restype = C.c_int
        L.ref_align_batch.argtypes = [
            C.c_void_p, C.c_int, C.c_int, c_int_p, C.POINTER(c_float_p), C.POINTER(c_float_p),
            C.POINTER(c_byte_p), C.POINTER(c_byte_p), C.POINTER(c_byte_p), C.c_int, C.POINTER(c_ubyte_p),
            c_float_p, c_int_p, c_int_p, C.POINTER(c_ubyte_p), C.c_int, C.c_int, c_int_p, c_int_p,
            C.POINTER(c_int_p), C.POINTER(c_int_p), C.POINTER(c_byte_p), C.POINTER(c_float_p), c_float_p, c_float_p]
        L.ref_exclude_alignment.restype = C.c_int
        L.ref_exclude_alignment.argtypes = [C.c_int, C.c_int, c_int_p, c_int_p, C.c_int, c_ubyte_p]
        L.ref_bench_align.restype = C.c_double
        L.ref_bench_align.argtypes = [C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, c_float_p,
                                      c_float_p, C.c_int, C.c_int, c_int_p, C.POINTER(c_float_p),
                                      C.POINTER(c_float_p), C.c_int, c_float_p, c_int_p, c_int_p,
                                      C.POINTER(C.c_double)]
        self.V = L.ref_vecsize()

    def log2f4(self, x):
        return self.lib.ref_log2f4(C.c_float(x))

    def fast_log2(self, x):
        return self.lib.ref_fast_log2(C.c_float(x))

    def dot20_vec(self, q, t):
        q, t = _f32(q), _f32(t)
        return self.lib.ref_scalarprod20vec(_fp(q), _fp(t))

    def dot20_scalar(self, q, t):
        q, t = _f32(q), _f32(t)
        return self.lib.ref_scalarprod20(_fp(q), _fp(t))

    def align_batch(self, par, qp, qtr, tps, ttrs, replicate=False, celloffs=None, ss=None, t_sss=None,
                    want_bt=True, want_path=False):
        """One reference batch (<= V templates).  Returns a list of AlignOut (bt has Ltb+1 columns)."""
        qp, qtr = _f32(qp), _f32(qtr)
        tps = [_f32(a) for a in tps]
        ttrs = [_f32(a) for a in ttrs]
        n = len(tps)
        assert 1 <= n <= self.V
        Lq = qp.shape[0] - 1
        Ls = np.array([a.shape[0] - 1 for a in tps], dtype=np.int32)
        Ltb = int(Ls.max())
        maxres = max(Lq, Ltb) + 2
        if ss is not None:
            h = self.lib.ref_create(maxres, par["local"], par["egq"], par["egt"], par["corr"], par["shift"],
                                    par["ss_mode"], par["ssw"], _fp(ss.S73), _fp(ss.S33), _fp(ss.S37))
            self.lib.ref_set_query(h, _fp(qp), _fp(qtr), Lq, _bp(ss.q_pred), _bp(ss.q_conf), _bp(ss.q_dssp))
            mode = ss.mode
        else:
            h = self.lib.ref_create(maxres, par["local"], par["egq"], par["egt"], par["corr"], par["shift"],
                                    par["ss_mode"], par["ssw"], None, None, None)
            self.lib.ref_set_query(h, _fp(qp), _fp(qtr), Lq, None, None, None)
            mode = 0
        pp = (c_float_p * n)(*[_fp(a) for a in tps])
        tt = (c_float_p * n)(*[_fp(a) for a in ttrs])
        keep = []
        if t_sss is not None:
            arrs = [[np.ascontiguousarray(x, dtype=np.int8) for x in t] for t in t_sss]
            keep.append(arrs)
            sp = (c_byte_p * n)(*[_bp(a[0]) for a in arrs])
            sc = (c_byte_p * n)(*[_bp(a[1]) for a in arrs])
            sd = (c_byte_p * n)(*[_bp(a[2]) for a in arrs])
        else:
            sp = sc = sd = None
        co = None
        if celloffs is not None:
            cos = [None if m is None else np.ascontiguousarray(m, dtype=np.uint8) for m in celloffs]
            keep.append(cos)
            co = (c_ubyte_p * n)(*[(_ubp(m) if m is not None else None) for m in cos])
        score = np.zeros(n, dtype=np.float32)
        i2 = np.zeros(n, dtype=np.int32)
        j2 = np.zeros(n, dtype=np.int32)
        bts = [np.zeros((Lq + 1, Ltb + 1), dtype=np.uint8) for _ in range(n)]
        btp = (c_ubyte_p * n)(*[_ubp(b) for b in bts])
        cap = Lq + Ltb + 4
        nsteps = np.zeros(n, dtype=np.int32)
        mcols = np.zeros(n, dtype=np.int32)
        isl = [np.zeros(cap, dtype=np.int32) for _ in range(n)]
        jsl = [np.zeros(cap, dtype=np.int32) for _ in range(n)]
        stl = [np.zeros(cap, dtype=np.int8) for _ in range(n)]
        Sl = [np.zeros(cap, dtype=np.float32) for _ in range(n)]
        hit = np.zeros(n, dtype=np.float32)
        sss = np.zeros(n, dtype=np.float32)
        rc = self.lib.ref_align_batch(
            h, n, int(bool(replicate)), _ip(Ls), pp, tt, sp, sc, sd, mode, co, _fp(score), _ip(i2), _ip(j2),
            btp if (want_bt or want_path) else None, int(bool(want_path)), cap, _ip(nsteps), _ip(mcols),
            (c_int_p * n)(*[_ip(a) for a in isl]), (c_int_p * n)(*[_ip(a) for a in jsl]),
            (c_byte_p * n)(*[_bp(a) for a in stl]), (c_float_p * n)(*[_fp(a) for a in Sl]), _fp(hit), _fp(sss))
        self.lib.ref_destroy(h)
        assert rc == Ltb, rc
        outs = []
        for e in range(n):
            o = AlignOut()
            o.score, o.i2, o.j2, o.bt = score[e], int(i2[e]), int(j2[e]), bts[e]
            if want_path:
                o.nsteps, o.matched_cols = int(nsteps[e]), int(mcols[e])
                o.i_steps, o.j_steps, o.states, o.S = isl[e], jsl[e], stl[e], Sl[e]
                o.hit_score, o.score_ss = hit[e], sss[e]
            outs.append(o)
        return outs

    def exclude_alignment(self, Lq, Lt, i_steps, j_steps, nsteps):
        mask = np.zeros((Lq + 1, Lt + 1), dtype=np.uint8)
        i_steps = np.ascontiguousarray(i_steps, dtype=np.int32)
        j_steps = np.ascontiguousarray(j_steps, dtype=np.int32)
        self.lib.ref_exclude_alignment(Lq, Lt, _ip(i_steps), _ip(j_steps), int(nsteps), _ubp(mask))
        return mask

    def bench_align(self, par, qp, qtr, tps, ttrs, threads=1):
        """The reference's batch loop (V templates per Align call), timed.  Returns
        (wall_seconds, map_seconds_summed_over_threads, score, i2, j2)."""
        qp, qtr = _f32(qp), _f32(qtr)
        tps = [_f32(a) for a in tps]
        ttrs = [_f32(a) for a in ttrs]
        N = len(tps)
        L = np.array([a.shape[0] - 1 for a in tps], dtype=np.int32)
        maxres = max(qp.shape[0] - 1, int(L.max())) + 2
        pp = (c_float_p * N)(*[_fp(a) for a in tps])
        tt = (c_float_p * N)(*[_fp(a) for a in ttrs])
        score = np.zeros(N, dtype=np.float32)
        i2 = np.zeros(N, dtype=np.int32)
        j2 = np.zeros(N, dtype=np.int32)
        mapsec = C.c_double()
        sec = self.lib.ref_bench_align(maxres, par["local"], par["egq"], par["egt"], par["corr"], par["shift"],
                                       _fp(qp), _fp(qtr), qp.shape[0] - 1, N, _ip(L), pp, tt, int(threads),
                                       _fp(score), _ip(i2), _ip(j2), C.byref(mapsec))
        return sec, mapsec.value, score, i2, j2


    def bench_hits(self, par, qp, qtr, tps, ttrs, threads=1, replicate=False):
        """The reference's batch loop with Backtrace + ScoreForBacktrace (ref_bench_hits, oracle/ref_harness.cpp)."""
        qp, qtr = _f32(qp), _f32(qtr)
        tps = [_f32(a) for a in tps]
        ttrs = [_f32(a) for a in ttrs]
        N = len(tps)
        L = np.array([a.shape[0] - 1 for a in tps], dtype=np.int32)
        maxres = max(qp.shape[0] - 1, int(L.max())) + 2
        pp = (c_float_p * N)(*[_fp(a) for a in tps])
        tt = (c_float_p * N)(*[_fp(a) for a in ttrs])
        o = _hits_buffers(N)
        self.lib.ref_bench_hits.restype = C.c_double
        o["sec"] = self.lib.ref_bench_hits(C.c_int(maxres), C.c_int(par["local"]), C.c_float(par["egq"]), C.c_float(par["egt"]),
                                           C.c_float(par["corr"]), C.c_float(par["shift"]), _fp(qp), _fp(qtr),
                                           C.c_int(qp.shape[0] - 1), C.c_int(N), _ip(L), pp, tt, C.c_int(int(threads)),
                                           C.c_int(int(bool(replicate))), *_hits_args(o))
        return o


DEFAULT_GAP = np.array([0.15, 1.0, 0.6, 0.6, 0.6, 0.6, 1.0], dtype=np.float32)   # gapd gape gapf gapg gaph gapi gapb
DEFAULT_PC = np.array([2, 1.0, 1.5, 1.0], dtype=np.float32)                      # pcm pca pcb pcc


def _prep_common(f, tr, neff):
    f, tr, neff = _f32(f), _f32(tr), _f32(neff)
    L = tr.shape[0] - 1
    assert f.shape == (L + 2, 20) and neff.shape == (L + 1, 3)
    return f, tr, neff, L


def oracle_prepare(orc, role, f, tr, neff, neff_hmm, pb, R, q_pav=None, gap=DEFAULT_GAP, pc=DEFAULT_PC, columnscore=1):
    """hho_prepare: PrepareQueryHMM (role 0) / PrepareTemplateHMM (role 1) restated. -> (p[(L+2),20], tr, pav)."""
    f, tr, neff, L = _prep_common(f, tr, neff)
    pb, R = _f32(pb), _f32(R).reshape(-1)
    qp = _f32(np.zeros(20) if q_pav is None else q_pav)
    gap, pc = _f32(gap), _f32(pc)
    p = np.zeros((L + 2, 20), dtype=np.float32)
    tro = np.zeros((L + 1, 7), dtype=np.float32)
    pav = np.zeros(20, dtype=np.float32)
    fn = orc.lib.hho_prepare
    fn.restype = C.c_int
    fn.argtypes = [C.c_int, C.c_int, c_float_p, c_float_p, c_float_p, C.c_float, c_float_p, c_float_p, c_float_p,
                   c_float_p, c_float_p, C.c_int, c_float_p, c_float_p, c_float_p]
    rc = fn(role, L, _fp(f), _fp(tr), _fp(neff), float(neff_hmm), _fp(pb), _fp(R), _fp(qp), _fp(gap), _fp(pc),
            int(columnscore), _fp(p), _fp(tro), _fp(pav))
    assert rc == 0, rc
    return p, tro, pav


def ref_prepare(ref, role, f, tr, neff, neff_hmm, q_pav=None, gap=DEFAULT_GAP, pc=DEFAULT_PC, columnscore=1, pb=None):
    """The reference's own PrepareQueryHMM / PrepareTemplateHMM call sequence (oracle/ref_hmm_harness.cpp)."""
    f, tr, neff, L = _prep_common(f, tr, neff)
    qp = _f32(np.zeros(20) if q_pav is None else q_pav)
    gap, pc = _f32(gap), _f32(pc)
    p = np.zeros((L + 2, 20), dtype=np.float32)
    tro = np.zeros((L + 1, 7), dtype=np.float32)
    pav = np.zeros(20, dtype=np.float32)
    fn = ref.lib.ref_prepare_raw
    fn.restype = C.c_int
    fn.argtypes = [C.c_int, C.c_int, c_float_p, c_float_p, c_float_p, C.c_float, c_float_p, c_float_p, c_float_p,
                   C.c_int, c_float_p, c_float_p, c_float_p, c_float_p]
    pbo = None if pb is None else _f32(pb)
    rc = fn(role, L, _fp(f), _fp(tr), _fp(neff), float(neff_hmm), _fp(qp), _fp(gap), _fp(pc), int(columnscore),
            _fp(pbo) if pbo is not None else None, _fp(p), _fp(tro), _fp(pav))
    assert rc == 0, rc
    return p, tro, pav


def ref_substitution_matrix(ref):
    pb = np.zeros(20, dtype=np.float32)
    R = np.zeros((20, 20), dtype=np.float32)
    ref.lib.ref_substitution_matrix.argtypes = [c_float_p, c_float_p]
    ref.lib.ref_substitution_matrix(_fp(pb), _fp(R))
    return pb, R


def ref_read_hhm_raw(ref, path, maxres=25000):
    f = np.zeros((maxres + 2, 20), dtype=np.float32)
    tr = np.zeros((maxres + 1, 7), dtype=np.float32)
    neff = np.zeros((maxres + 1, 3), dtype=np.float32)
    nh, L = C.c_float(), C.c_int()
    pb = np.zeros(20, dtype=np.float32)
    fn = ref.lib.ref_read_hhm_raw
    fn.argtypes = [C.c_char_p, C.c_int, c_float_p, c_float_p, c_float_p, C.POINTER(C.c_float), C.POINTER(C.c_int),
                   c_float_p]
    rc = fn(path.encode(), maxres, _fp(f), _fp(tr), _fp(neff), C.byref(nh), C.byref(L), _fp(pb))
    assert rc == 0, rc
    L = L.value
    return f[:L + 2].copy(), tr[:L + 1].copy(), neff[:L + 1].copy(), np.float32(nh.value), pb


def have_ref():
    return os.path.exists(REF_SO)


def have_oracle():
    return os.path.exists(ORACLE_SO)


# ---- MAC realignment ------------------------------------------------------------------------
class MacOut:
    pass


def _mac_buffers(Lq, Lt):
    o = MacOut()
    n = (Lq + 1, Lt + 1)
    o.q_tr_lin = np.zeros((Lq + 1, 7), np.float32)
    o.t_tr_lin = np.zeros((Lt + 1, 7), np.float32)
    o.celloff = np.zeros(n, np.uint8)
    o.forward = np.zeros(n, np.float32)
    o.posterior = np.zeros(n, np.float32)
    o.bmm = np.zeros(n, np.uint8)
    o.scale = np.zeros(Lq + 2, np.float64)
    o.Pforward = C.c_double()
    o.scalars = np.zeros(6, np.int32)
    o.sum_of_probs = C.c_float()
    cap = Lq + Lt + 2
    o.i_steps = np.zeros(cap, np.int32)
    o.j_steps = np.zeros(cap, np.int32)
    o.states = np.zeros(cap, np.int8)
    o.S = np.zeros(cap, np.float32)
    o.P = np.zeros(cap, np.float32)
    return o


def _mac_finish(o):
    o.Pforward = o.Pforward.value
    o.sum_of_probs = np.float32(o.sum_of_probs.value)
    o.nsteps, o.i1, o.j1, o.i2, o.j2, o.matched_cols = [int(v) for v in o.scalars]
    return o


def _prev_lists(prev):
    off = np.zeros(len(prev) + 1, np.int32)
    for k, (a, _) in enumerate(prev):
        off[k + 1] = off[k] + len(a)
    pi = np.concatenate([np.asarray(a, np.int32) for a, _ in prev] + [np.zeros(0, np.int32)]).astype(np.int32)
    pj = np.concatenate([np.asarray(b, np.int32) for _, b in prev] + [np.zeros(0, np.int32)]).astype(np.int32)
    return off, np.ascontiguousarray(pi), np.ascontiguousarray(pj)


def ref_mac_realign(ref, qp, qtr, tp, ttr, vit, local=1, shift=-0.03, mact=0.3501, corr=0.1, min_overlap=0, prev=()):
    """PosteriorDecoder::realign of the reference on prepared tensors (log2 transitions) and a Viterbi hit `vit`
    (AlignOut with path).  prev: list of (alt_i, alt_j) of earlier MAC alignments of the same template."""
    qp, qtr, tp, ttr = _f32(qp), _f32(qtr), _f32(tp), _f32(ttr)
    Lq, Lt = qp.shape[0] - 1, tp.shape[0] - 1
    o = _mac_buffers(Lq, Lt)
    off, pi, pj = _prev_lists(list(prev))
    ns = vit.nsteps
    f = ref.lib.ref_mac_realign
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                  C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                  C.c_void_p, C.c_void_p] + [C.c_void_p] * 15
    vi = np.ascontiguousarray(vit.i_steps, np.int32)
    vj = np.ascontiguousarray(vit.j_steps, np.int32)
    rc = f(qp.ctypes.data, qtr.ctypes.data, Lq, tp.ctypes.data, ttr.ctypes.data, Lt, int(local), shift, mact, corr,
           min_overlap, int(vi[ns]), int(vj[ns]), vit.i2, vit.j2, ns, vi.ctypes.data, vj.ctypes.data, len(off) - 1,
           off.ctypes.data, pi.ctypes.data, pj.ctypes.data, o.q_tr_lin.ctypes.data, o.t_tr_lin.ctypes.data,
           o.celloff.ctypes.data, o.forward.ctypes.data, o.posterior.ctypes.data, o.bmm.ctypes.data, o.scale.ctypes.data,
           C.addressof(o.Pforward), o.scalars.ctypes.data, C.addressof(o.sum_of_probs), o.i_steps.ctypes.data,
           o.j_steps.ctypes.data, o.states.ctypes.data, o.S.ctypes.data, o.P.ctypes.data)
    assert rc == 0
    # what writeProfilesToHits attached to the hit: lists[w] = (i, j, value) of forward / backward / posterior, profiles[w]
    ll = ref.lib.ref_mac_last_list
    ll.restype = C.c_long
    ll.argtypes = [C.c_int, C.c_long, C.c_void_p]
    o.lists = []
    for w in range(3):
        n = ll(w, 0, None)
        t = np.zeros((n, 3), np.float32)
        assert ll(w, n, t.ctypes.data) == n
        o.lists.append((t[:, 0].astype(np.int32), t[:, 1].astype(np.int32), t[:, 2].copy()))
    o.profiles = []
    for w in range(2):
        pr = np.zeros(Lq + 1, np.float32)
        assert ref.lib.ref_mac_last_profile(w, C.c_void_p(pr.ctypes.data)) == Lq + 1
        o.profiles.append(pr)
    return _mac_finish(o)


def mac_plane_to_list(plane):
    """dense list plane (value where the reference has an entry, 0 elsewhere) -> (i, j, value) in the reference's order"""
    i, j = np.nonzero(plane[1:, 1:] != 0)
    return (i + 1).astype(np.int32), (j + 1).astype(np.int32), plane[i + 1, j + 1].astype(np.float32)


def mac_posterior_list(posterior, celloff, i_steps, j_steps, nsteps):
    """Hit::posterior_matrix (src/hhbacktracemac.cpp:82-108): posterior >= 0.01, finite, cell on - by then backtraceMAC has
    switched off the cells within two rows / columns of every path step (:149-154)"""
    p = posterior[1:, 1:]
    co = celloff.copy()
    Lq, Lt = co.shape[0] - 1, co.shape[1] - 1
    for s in range(1, nsteps + 1):
        i, j = int(i_steps[s]), int(j_steps[s])
        co[max(i - 2, 1):min(i + 2, Lq) + 1, j] = 1
        co[i, max(j - 2, 1):min(j + 2, Lt) + 1] = 1
    with np.errstate(invalid="ignore"):
        keep = (p >= np.float32(0.01)) & (co[1:, 1:] == 0) & np.isfinite(p)
    i, j = np.nonzero(keep)
    return (i + 1).astype(np.int32), (j + 1).astype(np.int32), p[i, j].astype(np.float32)


def mac_list_profile(lst, Lq):
    """Hit::forward_profile / backward_profile: float sums of a list's values per row, in list order (src/hhbacktracemac.cpp:64,79)"""
    out = np.zeros(Lq + 1, np.float32)
    for i, v in zip(lst[0], lst[2]):
        out[i] = np.float32(out[i] + v)
    return out


def oracle_mac_realign(orc, qp, q_tr_lin, tp, t_tr_lin, vit, local=1, shift=-0.03, mact=0.3501, min_overlap=0, prev=()):
    """The oracle's restatement of realign(): same outputs as ref_mac_realign; transitions are LINEAR here."""
    qp, q_tr_lin, tp, t_tr_lin = _f32(qp), _f32(q_tr_lin), _f32(tp), _f32(t_tr_lin)
    Lq, Lt = qp.shape[0] - 1, tp.shape[0] - 1
    o = _mac_buffers(Lq, Lt)
    o.q_tr_lin, o.t_tr_lin = q_tr_lin, t_tr_lin
    off, pi, pj = _prev_lists(list(prev))
    ns = vit.nsteps
    vi = np.ascontiguousarray(vit.i_steps, np.int32)
    vj = np.ascontiguousarray(vit.j_steps, np.int32)
    L = orc.lib
    V = C.c_void_p
    L.hho_mac_celloff.argtypes = [C.c_int] * 8 + [V, V, C.c_int, V, V, V, V]
    L.hho_mac_forward.argtypes = [V, V, C.c_int, V, V, C.c_int, C.c_int, C.c_float, V, V, V, V]
    L.hho_mac_backward.argtypes = [V, V, C.c_int, V, V, C.c_int, C.c_int, C.c_float, V, V, C.c_double, V]
    L.hho_mac_dp.argtypes = [V, V, C.c_int, C.c_int, C.c_int, C.c_float, V, V, V]
    L.hho_mac_backtrace.argtypes = [V, V, V, V, C.c_int, C.c_int, C.c_int, C.c_int] + [V] * 8
    assert L.hho_mac_celloff(Lq, Lt, min_overlap, int(vi[ns]), int(vj[ns]), vit.i2, vit.j2, ns, vi.ctypes.data, vj.ctypes.data,
                             len(off) - 1, off.ctypes.data, pi.ctypes.data, pj.ctypes.data, o.celloff.ctypes.data) == 0
    assert L.hho_mac_forward(qp.ctypes.data, q_tr_lin.ctypes.data, Lq, tp.ctypes.data, t_tr_lin.ctypes.data, Lt, int(local),
                             shift, o.celloff.ctypes.data, o.forward.ctypes.data, o.scale.ctypes.data,
                             C.addressof(o.Pforward)) == 0
    o.posterior[:] = o.forward
    # the -o_matrices lists as dense planes (value where the reference pushes an entry, 0 elsewhere)
    o.fwd_list = np.zeros((Lq + 1, Lt + 1), np.float32)
    o.bwd_list = np.zeros((Lq + 1, Lt + 1), np.float32)
    L.hho_mac_forward_list.argtypes = [V, C.c_int, C.c_int, V, C.c_double, V]
    L.hho_mac_backward_list.argtypes = [V, V, C.c_int, V, V, C.c_int, C.c_int, C.c_float, V, V, C.c_double, V, V]
    assert L.hho_mac_forward_list(o.forward.ctypes.data, Lq, Lt, o.scale.ctypes.data, o.Pforward.value, o.fwd_list.ctypes.data) == 0
    assert L.hho_mac_backward_list(qp.ctypes.data, q_tr_lin.ctypes.data, Lq, tp.ctypes.data, t_tr_lin.ctypes.data, Lt, int(local),
                                   shift, o.celloff.ctypes.data, o.scale.ctypes.data, o.Pforward.value, o.posterior.ctypes.data,
                                   o.bwd_list.ctypes.data) == 0
    i2, j2 = C.c_int(), C.c_int()
    assert L.hho_mac_dp(o.posterior.ctypes.data, o.celloff.ctypes.data, Lq, Lt, int(local), mact, o.bmm.ctypes.data,
                        C.addressof(i2), C.addressof(j2)) == 0
    ns_o, mc = C.c_int(), C.c_int()
    assert L.hho_mac_backtrace(o.bmm.ctypes.data, o.posterior.ctypes.data, qp.ctypes.data, tp.ctypes.data, Lq, Lt, i2.value,
                               j2.value, o.i_steps.ctypes.data, o.j_steps.ctypes.data, o.states.ctypes.data, o.S.ctypes.data,
                               o.P.ctypes.data, C.addressof(ns_o), C.addressof(mc), C.addressof(o.sum_of_probs)) == 0
    n = ns_o.value
    o.scalars[:] = [n, o.i_steps[n], o.j_steps[n], i2.value, j2.value, mc.value]
    return _mac_finish(o)
