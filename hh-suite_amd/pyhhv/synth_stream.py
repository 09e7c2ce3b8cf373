"""Synthetic prepared templates generated with torch straight into the packed record stream,
with the random streams SURVEY.md 8(d) prescribes for BASELINE's configs:

    template with GLOBAL id g:  seed = 0x5EED0000 + g  ->  splitmix64 (four outputs = the state)  ->  xoshiro256**
    uniforms:                   u = (x >> 40) * 2^-24          (top 24 bits of each 64-bit output)

Every template has its own stream, so its columns depend on its global id alone - not on the rank that holds it, the
shard it was assigned to or its place in the stream.  A database sharded over N ranks (hhv_shard_plan) is therefore the
same database as the one a single rank holds (bench.py --virtual-shards, tests/test_gpu_configs.py).

Draw order of a template: column j = 1..L, per column 28 uniforms - 20 for the profile column, 8 for the transitions
(2 used: pI and pD; SURVEY.md 8(d) fixes the other transitions).  The query (`query_np`) is one more such stream, seeded with
0x51000000 itself.  The streams are walked in lock step: one torch tensor element per template, one step per draw, 18 small integer
kernels per draw; the templates are visited in descending length order so that the active set is a prefix.

xoshiro256** / splitmix64: Blackman & Vigna, public domain reference implementations (restated here on int64 tensors
with logical right shifts emulated by masks).  `uniforms_np` is the same generator in numpy on uint64, used by the tests
to pin the torch version (tests/test_synth_stream.py) together with the published first outputs of both generators.
"""
import numpy as np

SEED_BASE = 0x5EED0000
DRAWS_PER_COLUMN = 28

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64_states(seeds):
    """seeds: uint64 array (n,) -> (4, n) uint64: the first four splitmix64 outputs of every seed = xoshiro256** state."""
    x = np.asarray(seeds, dtype=np.uint64).copy()
    out = np.zeros((4, x.shape[0]), dtype=np.uint64)
    with np.errstate(over="ignore"):
        for k in range(4):
            x = x + _GOLD
            z = x.copy()
            z = (z ^ (z >> np.uint64(30))) * _M1
            z = (z ^ (z >> np.uint64(27))) * _M2
            out[k] = z ^ (z >> np.uint64(31))
    return out


def _rotl_np(x, k):
    return (x << np.uint64(k)) | (x >> np.uint64(64 - k))


def xoshiro_next_np(s):
    """s: (4, n) uint64 state, advanced in place; returns the (n,) uint64 outputs."""
    with np.errstate(over="ignore"):
        r = _rotl_np(s[1] * np.uint64(5), 7) * np.uint64(9)
        t = s[1] << np.uint64(17)
        s[2] ^= s[0]
        s[3] ^= s[1]
        s[1] ^= s[2]
        s[0] ^= s[3]
        s[2] ^= t
        s[3] = _rotl_np(s[3], 45)
    return r


def uniforms_np(gids, ndraws):
    """(len(gids), ndraws) float32: the first ndraws uniforms of the streams of the given global template ids."""
    s = splitmix64_states(np.asarray(gids, dtype=np.uint64) + np.uint64(SEED_BASE))
    out = np.zeros((s.shape[1], ndraws), dtype=np.float32)
    for d in range(ndraws):
        out[:, d] = (xoshiro_next_np(s) >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)
    return out


QUERY_SEED = 0x51000000


def query_np(Lq, pb):
    """The query SURVEY.md 8(d) prescribes: stream seeded with 0x51000000, the same draws per column as a template; p = column
    probabilities (no null model: HMM::p of a query), tr in the enum order of src/hhdecl.h:68; rows 0 and Lq as
    AddTransitionPseudocounts leaves them.  Returns (p[(Lq+1), 20], tr[(Lq+1), 7]) float32."""
    from scipy.special import erfinv
    s = splitmix64_states(np.array([QUERY_SEED], dtype=np.uint64))
    u = np.zeros((Lq + 1, DRAWS_PER_COLUMN), dtype=np.float64)
    for i in range(1, Lq + 1):
        for d in range(DRAWS_PER_COLUMN):
            u[i, d] = float(xoshiro_next_np(s)[0] >> np.uint64(40)) * 2.0 ** -24
    g = erfinv(2.0 * u[:, :20] - 1.0 + 2.0 ** -24) ** 2
    g[0] = 1.0
    g = g / g.sum(axis=1, keepdims=True)
    pbd = np.asarray(pb, dtype=np.float64)
    f = 0.7 * g + 0.3 * pbd[None, :]
    f = f / f.sum(axis=1, keepdims=True)
    f[0] = 0.0
    pI, pD = 0.01 + 0.04 * u[:, 20], 0.01 + 0.04 * u[:, 21]
    tr = np.zeros((Lq + 1, 7), dtype=np.float64)   # M2M, M2I, M2D, I2M, I2I, D2M, D2D
    tr[:, 0] = np.log2(1.0 - pI - pD)
    tr[:, 1] = 0.6 * np.log2(pI)
    tr[:, 2] = 0.6 * np.log2(pD)
    tr[:, 3] = tr[:, 5] = np.log2(0.6)
    tr[:, 4] = tr[:, 6] = 0.6 * np.log2(0.4)
    for i in (0, Lq):
        tr[i, 0] = 0.0
        tr[i, 1] = tr[i, 2] = -100000.0
    tr[Lq, 5] = 0.0
    tr[Lq, 6] = -100000.0
    return f.astype(np.float32), tr.astype(np.float32)


class _Xoshiro:
    """xoshiro256** on int64 torch tensors, one stream per element (two's-complement arithmetic wraps like uint64)."""

    def __init__(self, torch, states_u64, device):
        self.t = torch
        st = np.ascontiguousarray(states_u64).view(np.int64)
        self.s = [torch.from_numpy(st[k].copy()).to(device) for k in range(4)]
        n = st.shape[1]
        self.a = torch.empty(n, dtype=torch.int64, device=device)
        self.b = torch.empty(n, dtype=torch.int64, device=device)

    def next_top24(self, cnt, out):
        """advance the first cnt streams; out[:cnt] (int64) = top 24 bits of the outputs"""
        t = self.t
        s0, s1, s2, s3 = (x[:cnt] for x in self.s)
        a, b = self.a[:cnt], self.b[:cnt]
        # result = rotl(s1 * 5, 7) * 9
        t.mul(s1, 5, out=a)
        t.bitwise_right_shift(a, 57, out=b)
        b.bitwise_and_(0x7F)
        a.bitwise_left_shift_(7)
        a.bitwise_or_(b)
        a.mul_(9)
        t.bitwise_right_shift(a, 40, out=out[:cnt])
        out[:cnt].bitwise_and_(0xFFFFFF)
        # state update
        t.bitwise_left_shift(s1, 17, out=a)
        s2.bitwise_xor_(s0)
        s3.bitwise_xor_(s1)
        s1.bitwise_xor_(s2)
        s0.bitwise_xor_(s3)
        s2.bitwise_xor_(a)
        t.bitwise_right_shift(s3, 19, out=b)
        b.bitwise_and_((1 << 45) - 1)
        s3.bitwise_left_shift_(45)
        s3.bitwise_or_(b)


def uniforms_torch(torch, device, gids, ndraws):
    """torch twin of uniforms_np (tests)."""
    gids = np.asarray(gids, dtype=np.uint64)
    g = _Xoshiro(torch, splitmix64_states(gids + np.uint64(SEED_BASE)), device)
    n = gids.shape[0]
    tmp = torch.empty(n, dtype=torch.int64, device=device)
    out = torch.empty((n, ndraws), dtype=torch.float32, device=device)
    for d in range(ndraws):
        g.next_top24(n, tmp)
        out[:, d] = tmp.to(torch.float32) * (2.0 ** -24)
    return out


def gen_stream(torch, device, gids, Ls, pb):
    """Packed record stream of the templates `gids` (global ids, any order) with lengths `Ls`, in that order:
    per template a header (index = position in this set) + L column records; + terminal header + pad.
    Column values and transitions exactly as SURVEY.md 8(d) prescribes for BASELINE's configs 2-4 (normalised Gamma(0.5)
    draws mixed with the background, divided by the null model; M2I / M2D from U[0.01, 0.05], the other transitions
    constants; rows 0 and L as AddTransitionPseudocounts leaves them), drawn from the template's own stream.
    Returns (records tensor [(nrec + 1 + 256), 28] float32, rec_off int64 numpy [n+1], Ls int32 numpy)."""
    gids = np.asarray(gids, dtype=np.int64)
    Ls = np.asarray(Ls, dtype=np.int64)
    n = Ls.shape[0]
    assert gids.shape[0] == n and n >= 1 and Ls.min() >= 1
    rec_off = np.zeros(n + 1, dtype=np.int64)
    rec_off[1:] = np.cumsum(Ls + 1)
    nrec = int(rec_off[-1])
    total = nrec + 1 + 256
    rec = torch.zeros((total, 28), dtype=torch.float32, device=device)
    meta = rec.view(torch.int32)

    # --- the uniforms: streams walked in lock step, longest templates first (the active set is a prefix)
    order = np.argsort(-Ls, kind="stable")
    Ls_o = Ls[order]
    off_o = torch.from_numpy(rec_off[:-1][order]).to(device)
    g = _Xoshiro(torch, splitmix64_states(gids[order].astype(np.uint64) + np.uint64(SEED_BASE)), device)
    Lmax = int(Ls_o[0])
    cnt_of_j = np.searchsorted(-Ls_o, -np.arange(1, Lmax + 1), side="right")   # templates with L >= j
    col = torch.empty((n, DRAWS_PER_COLUMN), dtype=torch.int64, device=device)
    tmp = torch.empty(n, dtype=torch.int64, device=device)
    for j in range(1, Lmax + 1):
        cnt = int(cnt_of_j[j - 1])
        for e in range(DRAWS_PER_COLUMN):
            g.next_top24(cnt, tmp)
            col[:cnt, e] = tmp[:cnt]
        rec[off_o[:cnt] + j] = col[:cnt].to(torch.float32) * (2.0 ** -24)
    del col, tmp, g

    # --- uniforms -> record fields, in slabs ("Configs 2-4"):
    #   column:      f = 0.7 g + 0.3 pb with g = 20 Gamma(0.5) draws, normalised; p = f / pnul, pnul = pb.
    #                Gamma(1/2) = Z^2 / 2 with Z standard normal, and Z = sqrt(2) erfinv(2 u - 1): g = erfinv(x)^2 with
    #                x = 2 u - 1 + 2^-24 (u = k 2^-24: x is the midpoint of the k-th of 2^24 cells of (-1, 1), never +-1)
    #   transitions: per column pI, pD ~ U[0.01, 0.05]: M2I = 0.6 log2 pI, M2D = 0.6 log2 pD, M2M = log2(1 - pI - pD);
    #                I2M = D2M = log2 0.6, I2I = D2D = 0.6 log2 0.4.  Record j carries tr[j-1][M2M, M2D, D2M, D2D, I2M] and
    #                tr[j][I2I, M2I]: the M2M / M2D fields come from the draws of the record before it.
    pbt = torch.tensor(np.asarray(pb, dtype=np.float32), dtype=torch.float32, device=device)
    c_x2m = float(np.log2(np.float32(0.6)))                    # I2M, D2M
    c_x2x = float(np.float32(0.6) * np.log2(np.float32(0.4)))  # I2I, D2D
    pI_all = (0.01 + 0.04 * rec[:nrec, 20]).clone()
    pD_all = (0.01 + 0.04 * rec[:nrec, 21]).clone()
    chunk = 1 << 21
    for a in range(0, nrec, chunk):
        b = min(nrec, a + chunk)
        x = rec[a:b, 0:20].double() * 2.0 - 1.0 + 2.0 ** -24
        gg = torch.special.erfinv(x).square_().float()
        del x
        gg = gg / gg.sum(dim=1, keepdim=True)
        f = 0.7 * gg + 0.3 * pbt
        f = f / f.sum(dim=1, keepdim=True)
        rec[a:b, 0:20] = f / pbt
        # (record a - 1 is the record before record a: a template's header for its column 1, overwritten below)
        a1 = max(a - 1, 0)
        pI_prev, pD_prev = pI_all[a1:b - 1], pD_all[a1:b - 1]
        if a == 0:
            pI_prev = torch.cat([pI_all[:1], pI_prev])
            pD_prev = torch.cat([pD_all[:1], pD_prev])
        rec[a:b, 20] = torch.log2(1.0 - pI_prev - pD_prev)
        rec[a:b, 21] = torch.log2(pD_prev) * 0.6
        rec[a:b, 22] = c_x2m
        rec[a:b, 23] = c_x2x
        rec[a:b, 24] = c_x2m
        rec[a:b, 25] = c_x2x
        rec[a:b, 26] = torch.log2(pI_all[a:b]) * 0.6
        rec[a:b, 27] = 0.0
        del gg, f
    del pI_all, pD_all

    # --- per-record template index and column index; headers; the two boundary columns
    off_t = torch.from_numpy(rec_off).to(device)
    L_t = torch.from_numpy(Ls).to(device)
    pos = torch.arange(nrec, dtype=torch.int64, device=device)
    tid = torch.searchsorted(off_t, pos, right=True) - 1
    j = (pos - off_t[tid]).to(torch.int32)
    Lr = L_t[tid].to(torch.int32)
    is_hdr = j == 0
    # column 1 carries tr[0]: M2M = 0, no M->D out of column 0; column L: no M->I out of L (src/hhhmm.cpp:1755-1785)
    first = j == 1
    rec[:nrec, 20][first] = 0.0
    rec[:nrec, 21][first] = -100000.0
    last = j == Lr
    rec[:nrec, 26][last] = -100000.0
    meta[:nrec, 27] = torch.where(last, j | 0x40000000, j)
    rec[:nrec][is_hdr] = 0.0
    meta[:nrec, 27][is_hdr] = -2 ** 31
    meta[:nrec, 0][is_hdr] = tid[is_hdr].to(torch.int32)
    meta[:nrec, 1][is_hdr] = Lr[is_hdr]
    meta[nrec, 27] = -2 ** 31
    meta[nrec, 0] = -1
    return rec, rec_off, Ls.astype(np.int32)


def unpack_one(body, Lt):
    """header + Lt column records of one template -> (p[(Lt+1),20], tr[(Lt+1),7])"""
    p = np.zeros((Lt + 1, 20), dtype=np.float32)
    tr = np.zeros((Lt + 1, 7), dtype=np.float32)
    p[1:] = body[1:, 0:20]
    tr[:Lt, 0] = body[1:, 20]
    tr[:Lt, 2] = body[1:, 21]
    tr[:Lt, 5] = body[1:, 22]
    tr[:Lt, 6] = body[1:, 23]
    tr[:Lt, 3] = body[1:, 24]
    tr[1:, 4] = body[1:, 25]
    tr[1:, 1] = body[1:, 26]
    return p, tr


def unpack_blocks(rec_host, n, Lt):
    """n equal-length templates (their records, headers included, contiguous in rec_host) -> P (n, Lt+1, 20), T (n, Lt+1, 7)."""
    host = np.asarray(rec_host).reshape(n, Lt + 1, 28)
    P = np.zeros((n, Lt + 1, 20), dtype=np.float32)
    T = np.zeros((n, Lt + 1, 7), dtype=np.float32)
    P[:, 1:] = host[:, 1:, 0:20]
    T[:, :Lt, 0] = host[:, 1:, 20]
    T[:, :Lt, 2] = host[:, 1:, 21]
    T[:, :Lt, 5] = host[:, 1:, 22]
    T[:, :Lt, 6] = host[:, 1:, 23]
    T[:, :Lt, 3] = host[:, 1:, 24]
    T[:, 1:, 4] = host[:, 1:, 25]
    T[:, 1:, 1] = host[:, 1:, 26]
    return P, T


def unpack_templates(rec_host, rec_off, Ls, n, first=0):
    """Packed records (host numpy holding the records of templates first .. first+n-1, starting at rec_off[first])
    -> prepared AoS profiles (p[(L+1),20], tr[(L+1),7]) holding every value the DP reads."""
    tps, ttrs = [], []
    base = int(rec_off[first])
    Lsel = np.asarray(Ls[first:first + n])
    if n > 0 and int(Lsel.min()) == int(Lsel.max()):
        # equal lengths: two blocks, the per-template arrays are views (100 k templates without a Python-level copy each)
        P, T = unpack_blocks(rec_host[: n * (int(Lsel[0]) + 1)], n, int(Lsel[0]))
        return [P[k] for k in range(n)], [T[k] for k in range(n)]
    for k in range(first, first + n):
        Lt = int(Ls[k])
        o = int(rec_off[k]) - base
        p, tr = unpack_one(rec_host[o: o + Lt + 1], Lt)
        tps.append(p)
        ttrs.append(tr)
    return tps, ttrs
