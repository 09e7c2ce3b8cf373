"""The work queue's bookkeeping as a model (CPU only): a line-by-line Python restatement of `WorkQueue` and of the refill /
step schedule of `hhv_stream_kernel` (hh-suite_amd/csrc/hhv_stream_kernel.h), run against the ground truth "position p of an
array's stream is record X".  It pins the ALGORITHM's invariants - the device code itself is compared with the oracle in
tests/test_gpu_queue.py:

* the ring (4 chunks of C = W / 2 records) always holds the record of every position a lane is at, although a chunk may be
  filled from two places of the template stream (one junction per chunk at most);
* `record_of(p)` - current delta, and the delta in front of the junction passed LAST - is right for every lane of every step:
  one junction of history is enough because segments have >= HHV_SEGMENT_MIN_RECORDS records and the refill runs less than
  three chunks ahead of the first lane;
* the stream ends exactly behind the terminal header of the last segment drawn; nothing behind it is processed.
The segment lists come from the product's own planner (hhv_segment_plan)."""
import numpy as np
import pytest

M_OPEN = 0x3FFFFFFF
RING_CHUNKS = 4


class ArrayModel:
    """one systolic array of W lanes: WorkQueue + ring + loop schedule, as in the kernel"""

    def __init__(self, W, lead, segs, terminal, draw):
        self.W, self.C, self.lead = W, W // 2, lead
        self.segs, self.terminal, self.draw = segs, terminal, draw
        self.ring = {}            # ring slot index -> (position, record) it holds
        self.truth = []           # record of every stream position, in order
        first, end = self.segs[self.draw()]
        self.delta, self.J, self.Jlast, self.dprev = first, end - first, 0, 0
        self.tail, self.end = False, M_OPEN
        self.truth += list(range(first, end))

    def refill(self, cc):
        C = self.C
        P0 = cc * C
        cross = (not self.tail) and P0 + C > self.J
        nxt, nlen = 0, 1
        if cross:
            i = self.draw()
            if i < len(self.segs):
                nxt, e = self.segs[i]
                nlen = e - nxt
                self.truth += list(range(nxt, e))
            else:
                nxt = self.terminal
                self.truth.append(self.terminal)
                self.end = self.J + 1
                self.tail = True
        for k in range(C):
            p = P0 + k
            rec = p + self.delta
            if cross and p >= self.J:
                rec += (nxt - self.J) - self.delta
            self.ring[p % (RING_CHUNKS * C)] = (p, rec)
        if cross:
            self.dprev, self.Jlast = self.delta, self.J
            self.delta = nxt - self.J
            self.J += nlen

    def record_of(self, p):
        return p + self.delta + ((self.dprev - self.delta) if p < self.Jlast else 0)

    def run(self):
        W, C, lead = self.W, self.C, self.lead
        self.refill(0)
        self.refill(1)
        M = self.end
        nchunks = -(-M // C)
        s_end = M + W - 1
        processed = []
        c = 0
        while c * C - lead < s_end:
            if c > 0:
                if c + 1 < nchunks:
                    self.refill(c + 1)
                if self.end != M:
                    M = self.end
                    s_end, nchunks = M + W - 1, -(-M // C)
                    if c * C - lead >= s_end:
                        break
            s_lo = c * C - lead if c > 0 else 0
            for s in range(s_lo, min((c + 1) * C - lead, s_end)):
                for g in range(W):
                    p = s - g
                    if 0 <= p < M:
                        pos, rec = self.ring[p % (RING_CHUNKS * C)]
                        assert pos == p, ("ring slot overwritten or not yet filled", W, s, g, p, pos)
                        assert rec == self.truth[p], ("ring holds the wrong record", W, s, g, p)
                        assert self.record_of(p) == self.truth[p], ("record_of", W, s, g, p, self.Jlast, self.J)
                        if g == 0:
                            processed.append(rec)
                # the head prefetch of the next step (LEAD = 1) reads position s + 1 of lane 0: it must be in the ring
                if lead and s + 1 < M:
                    assert self.ring[(s + 1) % (RING_CHUNKS * C)][0] == s + 1, ("prefetched head not in the ring", W, s)
            c += 1
        return processed


@pytest.mark.parametrize("W,lead", [(64, 1), (64, 0), (32, 0), (16, 0)])
def test_queue_bookkeeping_against_ground_truth(W, lead):
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hh-suite_amd"))
    from pyhhv import capi
    rng = np.random.default_rng(W + lead)
    cases = [np.full(24, 127), np.full(9, 300), rng.integers(1, 70, 300), np.array([20]), np.array([60, 30]),
             np.array([200, 200, 30]), np.concatenate([rng.integers(1, 400, 60), [1000, 1, 1]]), np.full(13, 128), np.full(17, 126)]
    for L in cases:
        L = np.asarray(L, dtype=np.int32)
        n_seg, seg = capi.segment_plan(L)
        segs = [tuple(int(x) for x in r) for r in seg[:n_seg]]
        terminal = int(seg[n_seg][0])
        for n_arrays in (1, 3, n_seg + 2):
            # the arrays draw in turn: array k starts with segment k, then whoever reaches a junction first draws next -
            # modelled by running the arrays one after the other on a shared counter (any order is a legal schedule)
            counter = {"next": n_arrays}

            def draw_from(start):
                state = {"first": True}

                def draw():
                    if state["first"]:
                        state["first"] = False
                        return start
                    i = counter["next"]
                    counter["next"] += 1
                    return i
                return draw
            seen = []
            for k in range(n_arrays):
                if k >= n_seg:
                    continue          # (WorkQueue::start: more arrays than segments -> nothing to do)
                a = ArrayModel(W, lead, segs, terminal, draw_from(k))
                got = a.run()
                assert got[-1] == terminal and terminal not in got[:-1]           # the stream ends with the terminal header
                seen += got[:-1]
            assert sorted(seen) == list(range(terminal))                            # every record exactly once, over all arrays
