"""hhv_set_celloff_paths (-m gpu): the masks of an alternative-alignment round built on the device from earlier paths (exclude_alignments ->
Viterbi::ExcludeAlignment, src/hhviterbirunner.cpp:152-164,273-289; src/hhviterbi.cpp:61-77) against the oracle's ExcludeAlignment
applied path by path.  Round 5 builds them entry by entry from the closed form of the union of the +-40 crosses (hhv_topk.hip
celloff_band_kernel); arrays that are not paths, and profiles beyond the LDS tables, take the cell-by-cell kernel.  Covered: one and
several paths per template (third and fourth rounds), paths of other templates in between, -excl / -template_excl ranges on top,
a "path" that jumps (the fallback), queries of one strip, two strips (two buffer planes) and the short-query arrays, local and
global - each time the masked DP that follows must give the oracle's end points, scores and backtrace bytes with the oracle's mask."""
import numpy as np
import pytest

from pyoracle import make_params

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("local", [0, 1])
@pytest.mark.parametrize("Lq", [300, 431, 150, 64])
def test_device_masks_equal_exclude_alignment(oracle, Lq, local):
    from pyhhv import capi, synth
    rng = np.random.default_rng(500 + Lq + local)
    par = make_params(local=local)
    qf, qtr = synth.make_query(91000 + Lq, Lq)
    tps, ttrs = [], []
    for k, L in enumerate([300, 120, 41, 2, 1, 500, 81, 230, 300, 64, 7, 160]):
        p, tr = synth.make_homolog(92000 + k, qf, L=L) if (k % 3 != 2 and L >= 2) else synth.make_template(92000 + k, L)
        tps.append(p)
        ttrs.append(tr)
    n = len(tps)
    c = capi.Context(local=local)
    c.set_query(qf, qtr)
    ts = c.upload(tps, ttrs)
    masks = [np.zeros((Lq + 1, p.shape[0]), dtype=np.uint8) for p in tps]
    paths = []   # (template, nsteps, i_steps, j_steps) of every round so far
    qr, tr_ = [(10, 14)], [(3, 5)]
    for rnd in range(3):
        if rnd == 0:
            res = c.align(ts, backtrace=True)
        else:
            use_ranges = rnd == 2
            order = list(rng.permutation(len(paths)))      # the paths of a template need not be neighbours in the list
            extra = []
            if rnd == 2:   # an array that is no path (steps that jump): must be masked cell by cell like the reference would
                e = 0
                ii = np.array([0, 200, 50, 120, 7], dtype=np.int32)
                jj = np.array([0, 30, 250, 100, 9], dtype=np.int32)
                ii, jj = np.minimum(ii, Lq), np.minimum(jj, tps[e].shape[0] - 1)
                extra.append((e, 4, ii, jj))
                oracle.exclude_alignment(Lq, tps[e].shape[0] - 1, ii, jj, 4, mask=masks[e])
            c.set_celloff_paths(ts, [paths[k] for k in order] + extra, qranges=qr if use_ranges else (), tranges=tr_ if use_ranges else ())
            want_masks = [m.copy() for m in masks]
            if use_ranges:
                for m in want_masks:
                    m[qr[0][0]:min(qr[0][1], Lq) + 1, 1:] = 1
                    m[1:, tr_[0][0]:min(tr_[0][1], m.shape[1] - 1) + 1] = 1
            res = c.align(ts, celloff=True)
            for e in range(n):
                a = oracle.align(par, qf, qtr, tps[e], ttrs[e], celloff=want_masks[e], want_path=True)
                assert (a.i2, a.j2) == (res["i2"][e], res["j2"][e]) and np.float32(a.score) == res["score"][e], (Lq, local, rnd, e)
                assert np.array_equal(c.backtrace_matrix(ts, e)[1:, 1:] & 0x7F, a.bt[1:, 1:] & 0x7F), (Lq, local, rnd, e)
        c.hits(ts)
        for e in range(n):
            ns, i_s, j_s, st, S = c.hit_path(ts, e)
            paths.append((e, ns, i_s.copy(), j_s.copy()))
            oracle.exclude_alignment(Lq, tps[e].shape[0] - 1, i_s, j_s, ns, mask=masks[e])
    ts.free()
    c.close()
