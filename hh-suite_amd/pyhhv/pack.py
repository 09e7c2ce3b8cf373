"""numpy mirror of the device data layout -- used by the tests and the bench to
build / decode packed buffers; the product's own packer is hh-suite_amd/csrc/hhv_pack.cpp and the
two are checked against each other in tests/test_layout.py.

Column record (28 dwords) of column k of an HMM given as p[(L+1),20], tr[(L+1),7]
(enum order M2M,M2I,M2D,I2M,I2I,D2M,D2D, /root/reference src/hhdecl.h:68):
    [0..19]  p[k][a]
    [20..24] tr[k-1][M2M], tr[k-1][M2D], tr[k-1][D2M], tr[k-1][D2D], tr[k-1][I2M]
    [25..26] tr[k][I2I], tr[k][M2I]
    [27]     meta: int32  j | LAST(bit30) for columns,  bit31 set for headers ([0]=template index, [1]=L)
"""
import numpy as np

M2M, M2I, M2D, I2M, I2I, D2M, D2D = range(7)
REC_DW = 28
META_HDR = np.int32(-2 ** 31)
META_LAST = np.int32(0x40000000)
LANES = 64


MAX_R = 5


def strips_for(Lq):
    """(R, P): query rows per lane and number of passes of 64*R rows -- hhv_set_query's rule: minimise
    P * (0.7 + R) (0.7 cell-equivalents of per-step overhead), ties -> fewer passes; R <= 5 keeps the
    kernel at 2 waves per SIMD."""
    best = None
    for r in range(1, MAX_R + 1):
        p = -(-int(Lq) // (LANES * r))
        cost = p * (0.7 + r)
        if best is None or cost < best[0] - 1e-9 or (cost < best[0] + 1e-9 and p < best[2]):
            best = (cost, r, p)
    return best[1], best[2]


def rows_for(Lq):
    return strips_for(Lq)[0]


def pack_columns(p, tr):
    """(L,28) float32 records of columns 1..L (meta left 0)."""
    p = np.asarray(p, dtype=np.float32)
    tr = np.asarray(tr, dtype=np.float32)
    L = p.shape[0] - 1
    rec = np.zeros((L, REC_DW), dtype=np.float32)
    rec[:, 0:20] = p[1:]
    rec[:, 20] = tr[:-1, M2M]
    rec[:, 21] = tr[:-1, M2D]
    rec[:, 22] = tr[:-1, D2M]
    rec[:, 23] = tr[:-1, D2D]
    rec[:, 24] = tr[:-1, I2M]
    rec[:, 25] = tr[1:, I2I]
    rec[:, 26] = tr[1:, M2I]
    return rec


def pack_query(p, tr, R=None, P=None):
    """(P*64*R, 28) float32: row i-1 of the array = query row i; rows > Lq are zero."""
    Lq = p.shape[0] - 1
    if R is None:
        R, P = strips_for(Lq)
    if P is None:
        P = -(-Lq // (LANES * R))
    out = np.zeros((P * LANES * R, REC_DW), dtype=np.float32)
    out[:Lq] = pack_columns(p, tr)
    return out


def pack_stream(tps, ttrs, t_ss=None):
    """Concatenated template stream: per template a header record then L column records, plus one
    terminal header.  t_ss: optional per-template (ss_pred, ss_conf, ss_dssp) arrays -> meta bits 16-21
    (pred_index = ss_pred*11 + ss_conf) and 22-24 (ss_dssp).
    Returns (records[(sum(L+1)+1), 28] float32, rec_off[n+1] int64)."""
    n = len(tps)
    Ls = np.array([t.shape[0] - 1 for t in tps], dtype=np.int64)
    rec_off = np.zeros(n + 1, dtype=np.int64)
    rec_off[1:] = np.cumsum(Ls + 1)
    total = int(rec_off[-1]) + 1
    rec = np.zeros((total, REC_DW), dtype=np.float32)
    meta = rec.view(np.int32)
    for k in range(n):
        o = int(rec_off[k])
        L = int(Ls[k])
        meta[o, 27] = META_HDR
        meta[o, 0] = k
        meta[o, 1] = L
        rec[o + 1:o + 1 + L] = pack_columns(tps[k], ttrs[k])
        meta[o + 1:o + 1 + L, 27] = np.arange(1, L + 1, dtype=np.int32)
        meta[o + L, 27] |= META_LAST
        if t_ss is not None and t_ss[k] is not None:
            pr, cf, ds = (np.asarray(x, dtype=np.int32) for x in t_ss[k])
            meta[o + 1:o + 1 + L, 27] |= (((pr[1:] * 11 + cf[1:]) & 0x3F) << 16) | ((ds[1:] & 7) << 22)
    o = int(rec_off[-1])
    meta[o, 27] = META_HDR
    meta[o, 0] = -1
    meta[o, 1] = 0
    return rec, rec_off


BT_MM_RUNNING, BT_MM_FIRST_EQUAL, BT_MM_FIRST_EQUAL_NEG = 1, 3, 5


def bt_decode(entry, r, R, mm_mode=BT_MM_FIRST_EQUAL):
    """numpy mirror of bt_decode (csrc/viterbi_lane.h): uint64 entries -> the reference's backtrace byte of row r.
    mm_mode: encoding of the MM predecessor (FIRST_EQUAL: e0 = (m > smin), e_k = (c_k == m); FIRST_EQUAL_NEG: the four
    equality bits stored inverted (sign of c_k - m); RUNNING: c_k > running max)."""
    entry = np.asarray(entry, dtype=np.uint64)
    lo = (entry & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    hi = (entry >> np.uint64(32)).astype(np.uint32)
    if r >= 1:
        f7 = (((lo >> np.uint32(2 * (R - 1) + 5 * (r - 1))) & np.uint32(0x1F)) << np.uint32(2)) | ((lo >> np.uint32(2 * (r - 1))) & np.uint32(3))
    else:
        f7 = (hi >> np.uint32(2 * R)) & np.uint32(0x7F)
    c2 = (hi >> np.uint32(2 * (R - 1 - r))) & np.uint32(3)
    b = np.zeros(entry.shape, dtype=np.uint8)
    if mm_mode == BT_MM_FIRST_EQUAL_NEG:
        f7 = f7 ^ np.uint32(0x3C)
        mm_mode = BT_MM_FIRST_EQUAL
    if mm_mode == BT_MM_FIRST_EQUAL:
        b = np.full(entry.shape, 6, dtype=np.uint8)
        for bit, code in ((0x04, 5), (0x08, 4), (0x10, 3), (0x20, 2)):      # the FIRST candidate equal to the maximum wins
            b = np.where(f7 & np.uint32(bit), np.uint8(code), b)
        b = np.where(f7 & np.uint32(0x40), b, np.uint8(0))
    else:
        for bit, code in ((0x40, 2), (0x20, 3), (0x10, 4), (0x08, 5), (0x04, 6)):
            b = np.where(f7 & np.uint32(bit), np.uint8(code), b)
    b |= np.where(f7 & np.uint32(2), 8, 0).astype(np.uint8)
    b |= np.where(f7 & np.uint32(1), 16, 0).astype(np.uint8)
    b |= np.where(c2 & np.uint32(2), 32, 0).astype(np.uint8)
    b |= np.where(c2 & np.uint32(1), 64, 0).astype(np.uint8)
    return b


def bt_to_matrix(bt_entries, rec_off_k, Lq, Lt, R, entry_bytes=8, mm_mode=BT_MM_FIRST_EQUAL):
    """Device backtrace entries (9 compare bits per cell, csrc/viterbi_lane.h bt_push) -> reference layout (Lq+1, Lt+1)
    bytes.  bt_entries: flat uint8 view of the [pass][record][lane][8] buffer."""
    P = -(-Lq // (LANES * R))
    e = np.ascontiguousarray(np.asarray(bt_entries, dtype=np.uint8)).reshape(P, -1, LANES, entry_bytes)
    e = e.view(np.uint64).reshape(P, -1, LANES)                                        # [pass][record][lane]
    out = np.zeros((Lq + 1, Lt + 1), dtype=np.uint8)
    for i in range(1, Lq + 1):
        strip, r = (i - 1) // R, (i - 1) % R
        out[i, 1:] = bt_decode(e[strip // LANES, rec_off_k + 1: rec_off_k + 1 + Lt, strip % LANES], r, R, mm_mode)
    return out


def celloff_bits(bt_entries, rec_off_k, Lq, Lt, R, entry_bytes=8):
    """The cell-off INPUT of the kernel: bit 7 of byte r of the entry (written by matrix_to_bt / hhv_set_celloff)."""
    P = -(-Lq // (LANES * R))
    e = np.asarray(bt_entries, dtype=np.uint8).reshape(P, -1, LANES, entry_bytes)
    out = np.zeros((Lq + 1, Lt + 1), dtype=np.uint8)
    for i in range(1, Lq + 1):
        strip, r = (i - 1) // R, (i - 1) % R
        out[i, 1:] = e[strip // LANES, rec_off_k + 1: rec_off_k + 1 + Lt, strip % LANES, r] >> 7
    return out


def matrix_to_bt(mask, rec_off_k, R, bt_entries, entry_bytes=8, bit=0x80):
    """Scatter a (Lq+1, Lt+1) 0/1 cell-off mask into bit 7 of the device backtrace buffer."""
    Lq, Lt = mask.shape[0] - 1, mask.shape[1] - 1
    P = -(-Lq // (LANES * R))
    e = np.asarray(bt_entries).reshape(P, -1, LANES, entry_bytes)
    for i in range(1, Lq + 1):
        strip, r = (i - 1) // R, (i - 1) % R
        col = e[strip // LANES, rec_off_k + 1: rec_off_k + 1 + Lt, strip % LANES, r]
        col[:] = np.where(mask[i, 1:] != 0, col | bit, col & ~np.uint8(bit))
    return bt_entries
