"""The rows of SURVEY.md 8 chained the way hhblits chains them, product vs reference, on one synthetic database:

   prefilter (N3)  ->  PrepareTemplateHMM of the survivors (N2)  ->  Viterbi + backtrace (the hot path, A1-A10)
                   ->  MAC realignment of the best hits (N4)

Product: hhv::Prefilter / hhv_prepare_templates / hhv_align + hhv_hits / hhv::PosteriorDecoderRunner, every stage on the GPU.
Reference: Prefilter::prefilter_db, the HMM preparation chain, Viterbi::Align + Backtrace + ScoreForBacktrace and
PosteriorDecoder::realign, run in-process from the reference's own translation units (oracle/_ref).
Each stage consumes the PRODUCT's output of the stage before it, and is compared with the reference run on the same input."""
import os

import numpy as np
import pytest

import pyoracle as po
from pyhhv import synth

HERE = os.path.dirname(os.path.abspath(__file__))


def raw_homolog(seed, fq, trq, nq, L):
    """A raw HMM whose columns follow a window of the raw query (sparser / noisier), so that it scores well."""
    rng = np.random.default_rng(seed)
    f, tr, neff, nh = synth.make_raw_hmm(seed, L)
    Lq = fq.shape[0] - 2
    start = int(rng.integers(1, max(2, Lq - L)))
    for j in range(1, L + 1):
        i = start + j - 1
        if i <= Lq and rng.random() < 0.85:
            mix = 0.8 * fq[i].astype(np.float64) + 0.2 * f[j].astype(np.float64)
            f[j] = (mix / mix.sum()).astype(np.float32)
    return f, tr, neff, nh


@pytest.mark.gpu
def test_gpu_pipeline_matches_reference_chain(oracle, ref):
    from pyhhv import capi
    import test_prefilter as tpf
    z = np.load(os.path.join(HERE, "golden", "gonnet_pb_R.npz"))
    pb, R = z["pb"], z["R"]
    lib = np.load(os.path.join(HERE, "golden", "cs219_probs.npz"))["lib"]
    rng = np.random.default_rng(2024)

    # ---- query: raw -> prepared (host side in both worlds: PrepareQueryHMM is not part of the path)
    Lq = 180
    fq, trq, nq, nhq = synth.make_raw_hmm(7001, Lq)
    q_p, q_tr, q_pav = po.oracle_prepare(oracle, 0, fq, trq, nq, nhq, pb, R)
    qp = np.ascontiguousarray(q_p[:-1])            # (Lq+1, 20)

    # ---- database: 600 raw HMMs, every fifth one related to the query; cs219 sequences for the prefilter
    n_db = 600
    raws = []
    for k in range(n_db):
        L = int(rng.integers(30, 260))
        raws.append(raw_homolog(8000 + k, fq, trq, nq, L) if k % 5 == 0 else synth.make_raw_hmm(8000 + k, L))
    prof = capi.prefilter_profile(np.ascontiguousarray(qp[:-1]), q_pav, lib)
    seqs, offs, lens = tpf.make_db(prof, n_db, 99)   # sequences 0, 3, 6, ... follow the query profile

    c = capi.Context(local=1, shift=-0.03, corr=0.1, ss_mode=0)

    # ---- N3: prefilter
    pf_q = np.ascontiguousarray(qp[:-1])
    ids, ev, passed1 = capi.prefilter_db(c, seqs, offs, lib, pf_q, q_pav, min_hits=30)
    want_ids = tpf.ref_prefilter_db(ref, pf_q, q_pav, seqs, offs, min_hits=30)
    assert np.array_equal(ids, want_ids) and 30 <= len(ids) < n_db
    sel = [int(k) for k in ids]

    # ---- N2: the whole raw database is resident; the survivors are prepared on the device from their ids alone
    c.set_query(qp, q_tr)
    raw, Ls_all = c.upload_raw([r[0] for r in raws], [r[1] for r in raws], [r[2] for r in raws], [r[3] for r in raws])
    ts = c.prepare_subset(raw, Ls_all, capi.prep_params(pb, R), q_pav, ids)
    ref_p, ref_tr = [], []
    for pos, k in enumerate(sel):
        f, tr, neff, nh = raws[k]
        p, tro, _ = po.ref_prepare(ref, 1, f, tr, neff, nh, q_pav=q_pav, pb=pb)
        ref_p.append(np.ascontiguousarray(p[:-1]))
        ref_tr.append(tro)
        assert np.array_equal(c.records_of(ts, pos).view(np.uint32), capi.pack_profile(ref_p[-1], tro, index=pos).view(np.uint32)), k

    # ---- hot path: Viterbi + backtrace + hit scores on the device-prepared set
    c.align(ts, backtrace=True)
    hits = c.hits(ts)
    par = po.make_params(local=1, ss_mode=0)
    outs = [ref.align_batch(par, qp, q_tr, [ref_p[pos]], [ref_tr[pos]], want_path=True)[0] for pos in range(len(sel))]
    for pos, o in enumerate(outs):
        h = hits[pos]
        assert (h["i2"], h["j2"], h["nsteps"], h["i1"], h["j1"]) == (o.i2, o.j2, o.nsteps, o.i_steps[o.nsteps], o.j_steps[o.nsteps]), pos
        assert np.float32(h["score"]).tobytes() == np.float32(o.hit_score).tobytes(), pos

    # ---- N4: MAC realignment of the 12 best hits (Hit.score order), Viterbi alignments taken from the product
    # (profiles stay on the device: the runner reads them from the resident set; only the linear transitions of the
    #  realigned templates - powf on the host - and the Viterbi paths go in)
    best = [int(k) for k in np.argsort(-hits["score"], kind="stable")[:12]]
    q_lin = capi.linear_transitions(q_tr, True)
    t_lin_all = [capi.linear_transitions(t, False) for t in ref_tr]
    # half of the hits hand their Viterbi path over, the other half let the device take it from the set's trace results
    mac_in = []
    for e, pos in enumerate(best):
        if e % 2:
            mac_in.append((pos, 1, 0, 0, 0, 0, -1, None, None))
        else:
            ns, i_s, j_s, st, S = c.hit_path(ts, pos)
            h = hits[pos]
            mac_in.append((pos, 1, int(h["i1"]), int(h["j1"]), int(h["i2"]), int(h["j2"]), ns, i_s, j_s))
    sc, re, o_i, o_j, o_s, o_S, o_P = capi.runner_mac_realign(c, qp, q_lin, None, t_lin_all, mac_in, resident=ts)
    for e, pos in enumerate(best):
        r = po.ref_mac_realign(ref, qp, q_tr, ref_p[pos], ref_tr[pos], outs[pos], local=1)
        assert tuple(sc[e]) == (r.nsteps, r.i1, r.j1, r.i2, r.j2, r.matched_cols), pos
        n = r.nsteps
        assert np.array_equal(o_i[e, 1:n + 1], r.i_steps[1:n + 1]) and np.array_equal(o_j[e, 1:n + 1], r.j_steps[1:n + 1])
        assert o_P[e, 1:n + 1].tobytes() == r.P[1:n + 1].tobytes()
        assert np.float64(re[e, 0]).tobytes() == np.float64(r.Pforward).tobytes()
    c.rawset_free(raw)
    ts.free()
    c.close()
