"""The hand-over protocol of the two-wave workgroups as a model (CPU only): a line-by-line Python restatement of what
`hhv_pair_kernel` (hh-suite_amd/csrc/hhv_stream_kernel.h: PairLds, WorkQueue::refill<W, PM>, the chunk-boundary block of
stream_body) does with its shared LDS words, run under random interleavings of the two wavefronts of several workgroups.
The device code itself is compared with the oracle and with the one-launch-per-strip path in tests/test_gpu_pair.py; the model
pins the PROTOCOL's invariants, which no finite number of hardware runs can cover:

* the second wave finds in FIFO slot (position mod 256) the row of exactly the position it asks for - for the step it is in
  and for the one it reads ahead - i.e. the first wave has written it and has not lapped it;
* neither wave ever waits for a condition only the other, equally waiting, wave could make true (no deadlock), whatever the
  relative speed of the two - so the bounded spins of `pair_wait` never run out;
* the second wave takes the segments in the order the first drew them, the 16-entry id ring is never lapped, and over all
  workgroups every record of the database is processed exactly once by either strip.
The segment lists come from the product's own planner (hhv_segment_plan); the refill / queue bookkeeping is the model of
tests/test_queue_model.py."""
import os
import sys

import numpy as np
import pytest

from test_queue_model import ArrayModel

W, C, LEAD = 64, 32, 1      # 64-lane arrays with the prefetched head (the only form the pair kernels exist in)
PAIR_FIFO = 256
INF = 0x7FFFFFFF
BLOCKED, STEP = "blocked", "step"


class PairLds:
    def __init__(self):
        self.fifo = {}              # slot -> (position, record)
        self.seg_id = [None] * 16
        self.seg_count = 0
        self.w0_done = 0
        self.w1_done = 0


class PairWave(ArrayModel):
    """one wavefront of a pair: role 1 = first strip (draws, publishes, writes the FIFO), role 2 = second strip"""

    def __init__(self, role, lds, segs, terminal, first_id, ticket):
        self.role, self.lds, self.ticket = role, lds, ticket
        self.draws = 0              # WorkQueue::draws: refill draws (the first segment is the workgroup's own number)
        self.other = None
        state = {"first": True}

        def draw():
            if state["first"]:
                state["first"] = False
                return first_id
            if self.role == 1:
                i = self.ticket()
                assert self.lds.seg_count - self.other.draws < 16, "segment id ring lapped"
                self.lds.seg_id[self.draws & 15] = i
                self.lds.seg_count = self.draws + 1
            else:
                assert self.lds.seg_count >= self.draws + 1   # (refill_gen has waited for it)
                i = self.lds.seg_id[self.draws & 15]
            self.draws += 1
            return i
        super().__init__(W, LEAD, segs, terminal, draw)

    def refill_gen(self, cc):
        cross = (not self.tail) and cc * self.C + self.C > self.J
        if cross and self.role == 2:
            while self.lds.seg_count < self.draws + 1:          # pair_wait(seg_count, draws + 1)
                yield BLOCKED
        self.refill(cc)

    def run_gen(self, processed):
        lds = self.lds
        yield from self.refill_gen(0)
        yield from self.refill_gen(1)
        if self.role == 2:
            while lds.w0_done < C + 1:                            # prologue: the first chunk's rows and the one read ahead
                yield BLOCKED
            assert lds.fifo[0][0] == 0
        M = self.end
        nchunks = -(-M // C)
        s_end = M + W - 1
        c = 0
        while c * C - LEAD < s_end:
            if c > 0:
                if self.role == 1:
                    if c * C - LEAD - (W - 1) > 0:
                        lds.w0_done = c * C - LEAD - (W - 1)
                    pmax = (c + 1) * C - LEAD - W
                    while pmax - (PAIR_FIFO - 1) > 0 and lds.w1_done < pmax - (PAIR_FIFO - 1):
                        yield BLOCKED
                else:
                    lds.w1_done = c * C - LEAD
                    while lds.w0_done < (c + 1) * C - LEAD + 1:
                        yield BLOCKED
                if c + 1 < nchunks:
                    yield from self.refill_gen(c + 1)
                if self.end != M:
                    M = self.end
                    s_end, nchunks = M + W - 1, -(-M // C)
                    if c * C - LEAD >= s_end:
                        break
            s_lo = c * C - LEAD if c > 0 else 0
            for s in range(s_lo, min((c + 1) * C - LEAD, s_end)):
                if self.role == 1:
                    r = s - (W - 1)                                # the last lane's position: its bottom row goes into the FIFO
                    if 0 <= r < M:
                        old = lds.fifo.get(r % PAIR_FIFO)
                        assert old is None or old[0] < self.other.consumed, ("FIFO slot overwritten before it was read", r, old)
                        lds.fifo[r % PAIR_FIFO] = (r, self.truth[r])
                    if 0 <= s < M:
                        processed.append(self.truth[s])
                else:
                    if 0 <= s < M:                                 # lane 0 takes the row requested a step ago
                        assert lds.fifo[s % PAIR_FIFO] == (s, self.truth[s]), ("wrong row", s, lds.fifo.get(s % PAIR_FIFO))
                    self.consumed = s + 1
                    if s + 1 < M:                                  # ... and requests the next one (every step, used or not)
                        assert lds.fifo.get((s + 1) % PAIR_FIFO, (None,))[0] == s + 1, ("row read ahead is not there", s + 1)
                yield STEP
            c += 1
        if self.role == 1:
            lds.w0_done = INF
        else:
            lds.w1_done = INF
        self.consumed = INF


def run_workgroups(segs, terminal, n_wg, rng, bias):
    """n_wg pair workgroups on one shared ticket counter, the 2 n_wg wavefronts advanced in random order; bias[w] = relative
    speed of wave w of every workgroup"""
    counter = {"next": n_wg}

    def ticket():
        i = counter["next"]
        counter["next"] += 1
        return i
    gens, waves, processed = [], [], []
    for k in range(n_wg):
        if k >= len(segs):
            continue
        lds = PairLds()
        pair = [PairWave(1, lds, segs, terminal, k, ticket), PairWave(2, lds, segs, terminal, k, ticket)]
        pair[0].other, pair[1].other = pair[1], pair[0]
        pair[1].consumed = 0
        pair[0].consumed = 0
        out = []
        processed.append(out)
        for w in pair:
            gens.append(w.run_gen(out if w.role == 1 else []))
            waves.append(w)
    alive = list(range(len(gens)))
    blocked = set()
    weight = np.array([bias[waves[g].role - 1] for g in range(len(gens))], dtype=float)
    while alive:
        p = weight[alive] / weight[alive].sum()
        g = alive[int(rng.choice(len(alive), p=p))]
        for _ in range(int(rng.integers(1, 1 + 40 * bias[waves[g].role - 1]))):   # a burst of the chosen wavefront
            try:
                status = next(gens[g])
            except StopIteration:
                alive.remove(g)
                blocked.clear()
                break
            if status == BLOCKED:
                blocked.add(g)
                assert len(blocked) < len(alive), "deadlock: every live wavefront waits"
                break
            blocked.clear()
    for k in range(0, len(waves), 2):
        assert waves[k].truth == waves[k + 1].truth                # both strips walked the same records
        assert waves[k].truth[-1] == terminal
    seen = [r for out in processed for r in out[:-1]]
    assert sorted(seen) == list(range(terminal))


@pytest.mark.parametrize("bias", [(1, 1), (20, 1), (1, 20)])
def test_pair_protocol_under_random_schedules(bias):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hh-suite_amd"))
    from pyhhv import capi
    rng = np.random.default_rng(7 + bias[0] * 3 + bias[1])
    cases = [np.full(40, 127), np.full(9, 300), rng.integers(1, 70, 400), np.array([20]), np.array([60, 30]),
             np.array([200, 200, 30]), np.concatenate([rng.integers(1, 400, 40), [1000, 1, 1]]), np.full(13, 128),
             rng.integers(1, 4, 1500), np.array([3000, 5, 2900])]
    for L in cases:
        L = np.asarray(L, dtype=np.int32)
        n_seg, seg = capi.segment_plan(L)
        segs = [tuple(int(x) for x in r) for r in seg[:n_seg]]
        terminal = int(seg[n_seg][0])
        for n_wg in (1, 3, n_seg + 1):
            run_workgroups(segs, terminal, n_wg, rng, bias)
