"""A model of the MAC dataflow kernels' synchronisation (hh-suite_amd/csrc/hhv_mac.hip, round 5), run under random schedules.

The forward / backward kernels are workgroups of eight wavefronts that hand each other units (row, strip) through LDS and wait on
monotone progress counters; every buffer is reused two rows later.  The GPU tests see the schedules the hardware happens to
produce; this test restates WHO WAITS FOR WHAT (the DF_WAIT calls of the kernels, cited below) and WHO READS / WRITES WHAT, and lets
a random scheduler interleave the waves: a cell that is read must hold the version the reader expects, a cell that is overwritten
must have been read by everybody who needs the old version, and nobody may wait forever.  It checks the protocol, not the
arithmetic - for every number of parallel-part waves the kernels support, strips per row from 1 to 7 and a handful of rows.
CPU only."""
import random

import pytest


class Cell:
    """one buffer slot: the tag of its content and how many reads of that content are still to come"""

    def __init__(self, name):
        self.name, self.tag, self.pending = name, None, 0

    def write(self, tag, readers):
        assert self.pending == 0, "%s: %r overwritten by %r with %d reader(s) still to come" % (self.name, self.tag, tag, self.pending)
        self.tag, self.pending = tag, readers

    def read(self, tag):
        assert self.tag == tag, "%s: read expects %r, holds %r" % (self.name, tag, self.tag)
        assert self.pending > 0, "%s: more reads of %r than the writer announced" % (self.name, tag)
        self.pending -= 1


class Cells(dict):
    def __missing__(self, key):
        c = self[key] = Cell(str(key))
        return c


def run(waves, counters, rng):
    """waves: generators yielding lists of (counter, need) to wait for (an empty list = a scheduling point)"""
    pending = {}
    for k, g in waves.items():
        try:
            pending[k] = next(g)
        except StopIteration:  # (a wave without a single unit)
            pass
    while pending:
        ready = [k for k, w in pending.items() if all(counters[c] >= n for c, n in w)]
        assert ready, "deadlock: %r" % {k: [(c, n, counters[c]) for c, n in w] for k, w in pending.items()}
        k = rng.choice(ready)
        try:
            pending[k] = next(waves[k])
        except StopIteration:
            del pending[k]


# ---- forward (hhv_mac_forward_df_kernel) --------------------------------------------------------------------------------
def forward(Lq, ns, NP, rng, local=True):
    cells, cnt = Cells(), {("P", w): 0 for w in range(NP)}
    cnt.update({("S", par, ch): 0 for par in (0, 1) for ch in (0, 1)})
    cnt.update({"T": 0, "R": 0})
    n_of = lambda w: (ns - w + NP - 1) // NP if ns > w else 0
    p_done = lambda i, s: (("P", s % NP), (i - 1) * n_of(s % NP) + s // NP + 1)   # P_DONE(i, s)
    has_row = lambda i: 1 <= i <= Lq

    def P(w):
        own = 0
        for i in range(1, Lq + 2):
            cur, prv = i & 1, (i & 1) ^ 1
            if i >= 2:
                # the end of row i-1: every other P wave has finished it (DF_WAIT(cnt + DF_P + o1, (i - 1) * N_OF(o1), ...))
                yield [(("P", o), (i - 1) * n_of(o)) for o in range(NP) if o != w]
                if i - 1 >= 2:
                    for o in range(NP):
                        if n_of(o):
                            cells["pmax", prv, o].read(i - 1)
                if w == 0:
                    cells["rring", prv].write(i - 1, 1)       # the total reads it at the end of its row
                    cnt["R"] = i - 1
            if i > Lq:
                break
            above = ((i - 2) >> 1) * ns
            for s in range(w, ns, NP):
                if i >= 2:
                    yield [(("S", prv, 0), above + s + 1), (("S", prv, 1), above + s + 1), ("T", (i - 2) * ns + s + 1)]
                    for t in ([s, s - 1] if s > 0 else [s]):   # row i-1 at this strip and the column left of it
                        cells["mm", prv, t].read(i - 1)
                        cells["gd", prv, t].read(("y", i - 1))
                        cells["im", prv, t].read(("y", i - 1))
                        cells["dgmi", prv, t].read(i - 1)
                else:
                    yield []
                nxt = has_row(i + 1)
                # readers of what this unit leaves: the unit below it and the one right of that (their "column left"), the
                # unit right of this one (mm of the column left of ITS strip), the total (mask, summand; F_MM of column Lt)
                below = (1 if nxt else 0) + (1 if nxt and s + 1 < ns else 0)
                cells["mask", cur, s].write(i, 3)                                  # two chain waves, the total
                cells["mm", cur, s].write(i, below + (1 if s + 1 < ns else 0) + (1 if (not local and s == ns - 1 and i < Lq) else 0))
                cells["dgmi", cur, s].write(i, below)
                cells["xb2", s].write(i, 1)
                yield []
                if s > 0:
                    yield [p_done(i, s - 1)]                                       # DF_WAIT(P_DONE(i, s - 1), dead)
                    cells["mm", cur, s - 1].read(i)
                cells["gd", cur, s].write(("c", i), 1)
                cells["im", cur, s].write(("c", i), 1)
                cells["xb0", s].write(i, 1)
                cells["xb1", s].write(i, 1)
                if s + NP >= ns:
                    # (the row's maximum: read by every P wave at the start of the next row when the row is >= 2)
                    cells["pmax", cur, w].write(i, NP if i >= 2 else 0)
                own += 1
                cnt[("P", w)] = own
                yield []

    def S(par, ch):
        done = 0
        key = "gd" if ch == 0 else "im"
        for i in range(1 if par else 2, Lq + 1, 2):
            cur = i & 1
            for s in range(ns):
                yield [p_done(i, s)]
                cells["mask", cur, s].read(i)
                cells[key, cur, s].read(("c", i))
                cells["xb%d" % ch, s].read(i)
                yield []
                nxt = has_row(i + 1)
                cells[key, cur, s].write(("y", i), (1 if nxt else 0) + (1 if nxt and s + 1 < ns else 0))
                done += 1
                cnt[("S", par, ch)] = done
                yield []

    def T():
        for i in range(1, Lq + 1):
            cur = i & 1
            for s in range(ns):
                yield [p_done(i, s)]
                cells["xb2", s].read(i)
                cells["mask", cur, s].read(i)
                if s == ns - 1:
                    yield [("R", i)]
                    cells["rring", cur].read(i)
                    if not local and i < Lq:
                        cells["mm", cur, s].read(i)
                cnt["T"] = (i - 1) * ns + s + 1
                yield []

    waves = {("P", w): P(w) for w in range(NP)}
    waves.update({("S", par, ch): S(par, ch) for par in (0, 1) for ch in (0, 1)})
    waves["T"] = T()
    run(waves, cnt, rng)
    left = {c.name: (c.tag, c.pending) for c in cells.values() if c.pending}
    assert not left, "reads announced but never made: %r" % left


# ---- backward (hhv_mac_backward_df_kernel) ------------------------------------------------------------------------------
def backward(Lq, ns, NP, rng):
    cells, cnt = Cells(), {("P", w): 0 for w in range(NP)}
    cnt.update({("S", par, ch): 0 for par in (0, 1) for ch in (0, 1)})
    cnt.update({("T", v): 0 for v in (0, 1)})
    n_of = lambda w: (ns - w + NP - 1) // NP if ns > w else 0
    p_done = lambda i, s: (("P", s % NP), (Lq - 1 - i) * n_of(s % NP) + s // NP + 1)
    t_done = lambda i, s: (("T", s & 1), (Lq - 1 - i) * ((ns + 1 - (s & 1)) >> 1) + (s >> 1) + 1)
    has_row = lambda i: 1 <= i <= Lq - 1
    # row Lq (written before the waves part): B_MM of every strip, read by the units of row Lq-1
    for s in range(ns):
        if has_row(Lq - 1):
            cells["mm", Lq & 1, s].write(("mm", Lq), 1 + (1 if s + 1 < ns else 0))
            cells["dgmi", Lq & 1, s].write(Lq, 1)
            cells["sf", (Lq - 1) & 1, s].write(Lq - 1, 1)    # the first row's F_MM values are staged by everybody before the loops

    def P(w):
        own = 0
        for i in range(Lq - 1, 0, -1):
            cur, prv = i & 1, (i & 1) ^ 1
            for s in range(w, ns, NP):
                if i <= Lq - 2:
                    # B_MM of row i+1 at this strip and the column right of it (the other posterior wave's); the other P
                    # wave's unit (i+1, s+1) has read its column of row i+2, which this unit overwrites
                    yield [t_done(i + 1, s)] + ([t_done(i + 1, s - 1)] if s > 0 else [])
                    if s + 1 < ns:
                        yield [p_done(i + 1, s + 1)]
                else:
                    yield []
                for t in ([s, s - 1] if s > 0 else [s]):
                    cells["mm", prv, t].read(("mm", i + 1))
                cells["dgmi", prv, s].read(i + 1)
                nxt = has_row(i - 1)
                if s == 0:
                    cells["prow", cur].write(i, min(2, ns))                        # each posterior wave that has units
                cells["mask", cur, s].write(i, 3)                                  # two chain waves, one posterior wave
                cells["mm", cur, s].write(("t0", i), 1)
                cells["dgmi", cur, s].write(i, 1 if nxt else 0)
                cells["gd", cur, s].write(("c", i), 1)
                cells["im", cur, s].write(("c", i), 1)
                cells["xb0", s].write(i, 1)
                cells["xb1", s].write(i, 1)
                cells["xb23", s].write(i, 1)
                if s + NP >= ns and nxt:
                    for t in range(w, ns, NP):                                     # F_MM of the next row, this wave's strips
                        cells["sf", prv, t].write(i - 1, 1)
                own += 1
                cnt[("P", w)] = own
                yield []

    def S(par, ch):
        done = 0
        key = "gd" if ch == 0 else "im"
        first = Lq - 1 if ((Lq - 1) & 1) == par else Lq - 2
        for i in range(first, 0, -2):
            cur = i & 1
            for s in range(ns):
                yield [p_done(i, s)]
                cells["mask", cur, s].read(i)
                cells[key, cur, s].read(("c", i))
                cells["xb%d" % ch, s].read(i)
                yield []
                cells[key, cur, s].write(("y", i), 1 + (1 if s + 1 < ns else 0))   # the posterior waves of (i, s) and (i, s+1)
                done += 1
                cnt[("S", par, ch)] = done
                yield []

    def P2(v):
        own = 0
        for i in range(Lq - 1, 0, -1):
            cur = i & 1
            for s in range(v, ns, 2):
                need = ((Lq - 1 - i) >> 1) * ns + s + 1
                yield [(("S", cur, 0), need), (("S", cur, 1), need)]
                if s == v:
                    cells["prow", cur].read(i)
                cells["sf", cur, s].read(i)
                cells["mask", cur, s].read(i)
                for t in ([s, s - 1] if s > 0 else [s]):
                    cells["gd", cur, t].read(("y", i))
                    cells["im", cur, t].read(("y", i))
                cells["mm", cur, s].read(("t0", i))
                cells["xb23", s].read(i)
                nxt = has_row(i - 1)
                cells["mm", cur, s].write(("mm", i), (1 if nxt else 0) + (1 if nxt and s + 1 < ns else 0))
                own += 1
                cnt[("T", v)] = own                                                # posted before the posterior is worked out
                yield []

    waves = {("P", w): P(w) for w in range(NP)}
    waves.update({("S", par, ch): S(par, ch) for par in (0, 1) for ch in (0, 1)})
    waves.update({("P2", v): P2(v) for v in (0, 1)})
    run(waves, cnt, rng)
    left = {c.name: (c.tag, c.pending) for c in cells.values() if c.pending}
    assert not left, "reads announced but never made: %r" % left


@pytest.mark.parametrize("NP", [2, 3, 4])
@pytest.mark.parametrize("ns", [1, 2, 3, 4, 5, 7])
def test_forward_protocol(NP, ns):
    for Lq in (1, 2, 3, 6):
        for seed in range(25):
            forward(Lq, ns, NP, random.Random(1000 * seed + 17 * Lq + ns), local=bool(seed & 1))


@pytest.mark.parametrize("NP", [2, 3, 4])
@pytest.mark.parametrize("ns", [1, 2, 3, 4, 5, 7])
def test_backward_protocol(NP, ns):
    for Lq in (1, 2, 3, 4, 7):
        for seed in range(25):
            backward(Lq, ns, NP, random.Random(1000 * seed + 17 * Lq + ns))


def test_the_model_notices_a_missing_wait():
    """the bug the soak found when the posterior wave was split: P waited for the posterior wave of strip s but not of s-1"""
    import inspect
    src = inspect.getsource(backward).replace("yield [t_done(i + 1, s)] + ([t_done(i + 1, s - 1)] if s > 0 else [])", "yield [t_done(i + 1, s)]")
    ns_ = {}
    exec(src, globals(), ns_)
    caught = 0
    for seed in range(200):
        try:
            ns_["backward"](5, 4, 2, random.Random(seed))
        except AssertionError:
            caught += 1
    assert caught > 0
