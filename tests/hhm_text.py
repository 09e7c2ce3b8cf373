"""Synthetic profile HMMs as .hhm TEXT, in the format HMM::WriteToFile produces and HMM::Read parses
(src/hhhmm.cpp:202-694, 2173-2300 of the reference): header, SEQ block, NULL line, HMM block with
-round(1000*log2 p) integers and '*' for zero.  Test data for the drop-in tests (tests/test_dropin_runner.py):
the reference's own reader turns these texts into HMM objects on both sides of the comparison."""
import numpy as np

# internal amino-acid index of the k-th letter of the alphabetically sorted file order (src/hhdecl.h:61)
S2A = [0, 4, 3, 6, 13, 7, 8, 9, 11, 10, 12, 2, 14, 5, 1, 15, 16, 19, 17, 18]
SORTED = "ACDEFGHIKLMNPQRSTVWY"
INTERNAL = "ARNDCQEGHILKMFPSTWYV"
NULL_LINE = [3706, 5728, 4211, 4064, 4839, 3729, 4763, 4308, 4069, 3323, 5509, 4640, 4464, 4937, 4285, 4423, 3815,
             3783, 6325, 4665]  # the background every hhmake-written file carries


def _rng(seed):
    return np.random.default_rng(seed)


def _val(p):
    if p <= 0:
        return "*"
    v = int(round(-1000.0 * np.log2(p)))
    return "*" if v >= 99999 else str(max(v, 0))


def random_columns(seed, L, sharp=6.0):
    r = _rng(seed)
    u = r.random((L, 20)) ** sharp
    u[u < 0.02] = 0.0
    u[np.arange(L), r.integers(0, 20, L)] += 0.3
    return u / u.sum(axis=1, keepdims=True)


def mutate_columns(seed, f, mut=0.3):
    """columns related to f: a fraction `mut` of them replaced by random ones (a detectable homolog)"""
    r = _rng(seed)
    g = f.copy()
    L = f.shape[0]
    repl = r.random(L) < mut
    g[repl] = random_columns(seed + 1, L)[repl]
    return g


def hhm_text(name, f, seed, ss=None, neff=None):
    """f: [L, 20] match-state frequencies (internal amino-acid order, rows sum to 1).
    ss: None or dict with any of 'dssp' (str over -HECSTGB), 'pred' (str over HEC), 'conf' (str over 0-9), length L."""
    r = _rng(seed ^ 0x5A5A)
    L = f.shape[0]
    cons = "".join(INTERNAL[int(np.argmax(f[i]))] for i in range(L))
    lines = ["HHsearch 1.6", "NAME  %s synthetic profile" % name, "FAM   ", "FILE  %s" % name, "COM   tests/hhm_text.py",
             "DATE  Thu Jan  1 00:00:00 2026", "LENG  %d match states, %d columns in multiple alignment" % (L, L),
             "FILT  10 out of 12 sequences passed filter (-id 90 -cov 0 -qid 0 -qsc -20.00 -diff 100)"]
    neff_hmm = float(neff) if neff is not None else float(1.0 + 9.0 * r.random())
    lines.append("NEFF  %.1f " % neff_hmm)
    lines.append("SEQ")
    if ss:
        if "dssp" in ss:
            lines += [">ss_dssp", ss["dssp"]]
        if "pred" in ss:
            lines += [">ss_pred", ss["pred"]]
        if "conf" in ss:
            lines += [">ss_conf", ss["conf"]]
    lines += [">Consensus", cons.lower(), ">%s synthetic profile" % name, cons, "#"]
    lines.append("NULL   " + "\t".join(str(v) for v in NULL_LINE) + "\t")
    lines.append("HMM    " + "\t".join(SORTED) + "\t")
    lines.append("       M->M\tM->I\tM->D\tI->M\tI->I\tD->M\tD->D\tNeff\tNeff_I\tNeff_D")
    lines.append("       0\t*\t*\t0\t*\t0\t*\t*\t*\t*\t")
    with np.errstate(divide="ignore"):
        v = np.where(f > 0, np.rint(-1000.0 * np.log2(np.where(f > 0, f, 1.0))), 99999).astype(np.int64)[:, S2A]
    for i in range(L):
        row = ["*" if x >= 99999 else str(x) for x in v[i].tolist()]
        lines.append("%s %-4d %s\t%d" % (cons[i], i + 1, "\t".join(row), i + 1))
        last = i == L - 1
        if last or r.random() < 0.3:  # no observed inserts / deletes in this column
            tr = ["0", "*", "*", "0" if last else "*", "*", "0" if last else "*", "*"]
            nI = nD = 0
        else:
            pI, pD = 0.002 + 0.06 * r.random(), 0.002 + 0.06 * r.random()
            pII, pDD = 0.1 + 0.6 * r.random(), 0.1 + 0.6 * r.random()
            tr = [_val(1 - pI - pD), _val(pI), _val(pD), _val(1 - pII), _val(pII), _val(1 - pDD), _val(pDD)]
            nI, nD = int(3000 * r.random()), int(3000 * r.random())
        nM = int(1000 * (1.0 + 9.0 * r.random()))
        lines.append("       " + "\t".join(tr) + "\t%d\t%d\t%d\t" % (nM, nI, nD))
        lines.append("")
    lines.append("//")
    return ("\n".join(lines) + "\n").encode()


def random_ss(seed, L):
    r = _rng(seed ^ 0x77)
    return {"dssp": "".join(r.choice(list("-HECSTGB"), L)), "pred": "".join(r.choice(list("HEC"), L)),
            "conf": "".join(r.choice(list("0123456789"), L))}


def hmmer3_text(name, f, seed, stars=True, effn=None):
    """The same kind of profile as HMMER3 TEXT (what hmmbuild writes and HMM::ReadHMMer3 parses, src/hhhmm.cpp:1208-1716): natural-log
    costs with five decimals, '*' for probability zero, a COMPO line (the reader OVERWRITES the process-wide background pb with it,
    :1399-1404), node 0, then match / insert / transition lines per node.  f: [L, 20] in the internal amino-acid order.
    stars: the '*' entries hmmbuild writes (node 0: d->d; node L: m->d, d->d) - the reader turns them into log2(0) = -inf."""
    r = _rng(seed ^ 0x3C3C)
    L = f.shape[0]

    def cost(p):
        return "*" if p <= 0 else "%.5f" % max(0.0, -np.log(p))

    def row(vals):
        return "  ".join(cost(float(p)) for p in vals)

    compo = f.mean(axis=0)[S2A]
    compo = 0.7 * compo / compo.sum() + 0.3 / 20
    ins = np.full(20, 0.05)
    lines = ["HMMER3/f [3.1b2 | February 2015]", "NAME  %s" % name, "LENG  %d" % L, "ALPH  amino", "NSEQ  %d" % (3 + seed % 20),
             "EFFN  %.6f" % (float(effn) if effn is not None else 0.6 + 4.0 * r.random()), "STATS LOCAL MSV      -10.0000  0.70000",
             "HMM          " + "        ".join(SORTED), "            m->m     m->i     m->d     i->m     i->i     d->m     d->d",
             "  COMPO   " + row(compo), "          " + row(ins)]

    def trans(last, first):
        pI, pD = 0.002 + 0.06 * r.random(), 0.002 + 0.06 * r.random()
        pII, pDD = 0.1 + 0.6 * r.random(), 0.1 + 0.6 * r.random()
        t = [1 - pI - pD, pI, pD, 1 - pII, pII, 1 - pDD, pDD]
        if last:
            t = [1 - pI, pI, 0.0 if stars else 1e-9, 1 - pII, pII, 1.0, 0.0 if stars else 1e-9]
        if first:
            t[5], t[6] = 1.0, (0.0 if stars else 1e-9)
        return "          " + "  ".join(cost(p) for p in t)

    lines.append(trans(False, True))
    fs = f[:, S2A]
    for i in range(L):
        lines.append("%7d   %s %6d - -" % (i + 1, row(np.maximum(fs[i], 1e-6)), i + 1))
        lines.append("          " + row(ins))
        lines.append(trans(i == L - 1, False))
    lines.append("//")
    return ("\n".join(lines) + "\n").encode()
