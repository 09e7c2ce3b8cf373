"""Synthetic *prepared* profile HMMs (what Viterbi::Align sees after PrepareTemplateHMM,
/root/reference src/hhfunc.cpp:165-202): per HMM of length L

    p  : (L+1, 20) float32, row 0 unused (zeros)       query: probabilities, template: f / pnull
    tr : (L+1, 7)  float32, log2 transition scores in the reference enum order
         M2M, M2I, M2D, I2M, I2I, D2M, D2D (src/hhdecl.h:68)

The generator is counter based (splitmix64 of (seed, element index)), so any profile can be
regenerated anywhere (CPU checkers here, GPU box there) from its integer seed alone; nothing but
numpy is needed and the streams do not depend on numpy's own RNG implementation.
"""
import numpy as np

M2M, M2I, M2D, I2M, I2I, D2M, D2D = range(7)

# Amino-acid background used as null model (A R N D C Q E G H I L K M F P S T W Y V); any fixed
# positive distribution works for synthetic data -- this one is close to the Gonnet background the
# reference derives in SetSubstitutionMatrix (src/hhmatrices.cpp:53-58).
PB = np.array([0.0787, 0.0512, 0.0448, 0.0536, 0.0135, 0.0403, 0.0610, 0.0688, 0.0229, 0.0590,
               0.0964, 0.0593, 0.0237, 0.0396, 0.0483, 0.0683, 0.0585, 0.0132, 0.0321, 0.0668], dtype=np.float64)
PB = (PB / PB.sum()).astype(np.float32)

_GOLD = np.uint64(0x9E3779B97F4A7C15)


def _splitmix64(x):
    x = (x + _GOLD).astype(np.uint64)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def uniform(seed, n, stream=0):
    """n float32 uniforms in [0,1): u[k] = top 24 bits of splitmix64(mix(seed, stream) + k*GOLD)."""
    with np.errstate(over="ignore"):
        base = _splitmix64(np.array([np.uint64(seed) ^ (np.uint64(stream) * np.uint64(0xD1342543DE82EF95))],
                                    dtype=np.uint64))[0]
        k = np.arange(n, dtype=np.uint64)
        z = _splitmix64(base + k * _GOLD)
    return ((z >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)


def _columns(seed, L, sharp=6.0, stream=1):
    """(L+1, 20) column distributions f: 0.7 * peaky + 0.3 * background (row 0 = zeros)."""
    u = uniform(seed, (L + 1) * 20, stream).reshape(L + 1, 20).astype(np.float64)
    g = np.power(u, sharp) + 1e-9
    g /= g.sum(axis=1, keepdims=True)
    f = 0.7 * g + 0.3 * PB.astype(np.float64)[None, :]
    f /= f.sum(axis=1, keepdims=True)
    f[0, :] = 0.0
    return f


def _transitions(seed, L, stream=2):
    """(L+1, 7) log2 transition scores shaped like AddTransitionPseudocounts leaves them
    (src/hhhmm.cpp:1755-1785): no M->D / M->I from column 0 and L (score -100000), no D->D from L."""
    u = uniform(seed, (L + 1) * 4, stream).reshape(L + 1, 4).astype(np.float64)
    pI = 0.01 + 0.04 * u[:, 0]
    pD = 0.01 + 0.04 * u[:, 1]
    pII = 0.25 + 0.3 * u[:, 2]
    pDD = 0.25 + 0.3 * u[:, 3]
    tr = np.zeros((L + 1, 7), dtype=np.float64)
    tr[:, M2M] = np.log2(1.0 - pI - pD)
    tr[:, M2I] = np.log2(pI) * 0.6
    tr[:, M2D] = np.log2(pD) * 0.6
    tr[:, I2M] = np.log2(1.0 - pII)
    tr[:, I2I] = np.log2(pII) * 0.6
    tr[:, D2M] = np.log2(1.0 - pDD)
    tr[:, D2D] = np.log2(pDD) * 0.6
    for i in (0, L):
        tr[i, M2M] = 0.0
        tr[i, M2I] = -100000.0
        tr[i, M2D] = -100000.0
    tr[L, D2D] = -100000.0
    tr[L, D2M] = 0.0
    return tr.astype(np.float32)


def make_query(seed, L):
    """Query: p = column probabilities (sum to 1), tr as above."""
    f = _columns(seed, L)
    return f.astype(np.float32), _transitions(seed, L)


def make_template(seed, L):
    """Unrelated template: p = f / pnull (null model folded in, src/hhhmm.cpp:2059-2144)."""
    f = _columns(seed, L)
    p = (f / PB.astype(np.float64)[None, :]).astype(np.float32)
    p[0, :] = 0.0
    return p, _transitions(seed, L)


def make_homolog(seed, q_f, L=None, start=None, mut=0.35, indel=0.04):
    """Template derived from query columns q_f ((Lq+1,20) probabilities): a window of the query with
    mutated columns and random insertions/deletions, so that alignments are long, gapped and score
    high (exercises every backtrace state)."""
    Lq = q_f.shape[0] - 1
    u = uniform(seed, 4 * (Lq + 64) + 8, 7)
    if start is None:
        start = 1 + int(u[0] * max(1, Lq // 4))
    cols = []
    i = start
    k = 4
    noise = _columns(seed ^ 0xABCDEF, Lq + 64, stream=9)
    target = L if L is not None else Lq
    while len(cols) < target:
        r = u[k % u.size]
        k += 1
        if i > Lq or r < indel:            # insertion in template: unrelated column
            cols.append(noise[1 + (len(cols) % (Lq + 63))])
            continue
        if r < 2 * indel:                  # deletion: skip query column
            i += 1
            continue
        w = mut * u[(k + 1) % u.size]
        k += 1
        c = (1.0 - w) * q_f[i].astype(np.float64) + w * noise[1 + (len(cols) % (Lq + 63))]
        cols.append(c / c.sum())
        i += 1
    f = np.zeros((target + 1, 20), dtype=np.float64)
    f[1:] = np.array(cols)
    p = (f / PB.astype(np.float64)[None, :]).astype(np.float32)
    p[0, :] = 0.0
    return p, _transitions(seed, target)


def zipf_lengths(seed, n, lo=50, hi=1000, s=1.2):
    """Template lengths for config 5: L = lo - 1 + k, k ~ Zipf(s) truncated to 1..(hi-lo+1)."""
    K = hi - lo + 1
    w = 1.0 / np.power(np.arange(1, K + 1, dtype=np.float64), s)
    cdf = np.cumsum(w / w.sum())
    u = uniform(seed, n, 11).astype(np.float64)
    k = np.searchsorted(cdf, u, side="right") + 1
    return (lo - 1 + np.minimum(k, K)).astype(np.int32)


def make_raw_hmm(seed, L):
    """Raw (unprepared) profile HMM as HMM::Read leaves it (src/hhhmm.cpp:202-694): f[(L+2),20] match-state
    frequencies without pseudocounts (sparse: absent residues carry 2^-99.999 like '*' entries of an .hhm
    file), tr[(L+1),7] raw log2 transitions ('*' = -99.999), neff[(L+1),3] = Neff_M, Neff_I, Neff_D, Neff_HMM."""
    u = uniform(seed, (L + 2) * 20, 21).reshape(L + 2, 20).astype(np.float64)
    g = np.power(u, 8.0)
    g[g < 0.02] = 0.0
    g[np.arange(L + 2), np.argmax(u, axis=1)] += 0.05
    f = g / g.sum(axis=1, keepdims=True)
    f = np.where(f > 0, f, 2.0 ** -99.999).astype(np.float32)
    f[0, :] = 0.0
    f[L + 1, :] = 0.0
    t = uniform(seed, (L + 1) * 8, 22).reshape(L + 1, 8).astype(np.float64)
    pI, pD = 0.002 + 0.06 * t[:, 0], 0.002 + 0.06 * t[:, 1]
    pII, pDD = 0.1 + 0.6 * t[:, 2], 0.1 + 0.6 * t[:, 3]
    tr = np.zeros((L + 1, 7), dtype=np.float64)
    tr[:, M2M] = np.log2(1 - pI - pD)
    tr[:, M2I] = np.log2(pI)
    tr[:, M2D] = np.log2(pD)
    tr[:, I2M] = np.log2(1 - pII)
    tr[:, I2I] = np.log2(pII)
    tr[:, D2M] = np.log2(1 - pDD)
    tr[:, D2D] = np.log2(pDD)
    star = t[:, 4] < 0.3                     # columns without observed inserts/deletes
    tr[star, M2I] = tr[star, M2D] = -99.999
    tr[star, I2I] = tr[star, D2D] = -99.999
    tr[star, M2M] = tr[star, I2M] = tr[star, D2M] = 0.0
    tr[0] = [0.0, -99.999, -99.999, 0.0, -99.999, 0.0, -99.999]
    neff = np.zeros((L + 1, 3), dtype=np.float32)
    neff[:, 0] = 1.0 + 9.0 * t[:, 5]
    neff[:, 1] = np.where(star, 0.0, 3.0 * t[:, 6])
    neff[:, 2] = np.where(star, 0.0, 3.0 * t[:, 7])
    neff_hmm = np.float32(neff[1:, 0].mean())
    return f, tr.astype(np.float32), neff, neff_hmm
