"""Developer tool (CPU): which LDS accesses of the solve kernels' critical wave conflict, by the banking rules of /opt/skills/guides/MI355X_MICROARCH.md (LDS [CDNA4]):
ds_read_b64 is served in two groups of 32 lanes, bank = (byte address / 4) mod 64 (a double covers two banks); ds_write_b64 in four groups of 16 lanes,
bank = (byte address / 4) mod 32; identical addresses broadcast; an N-way conflict costs N LDS cycles for that group.

For every access pattern of wave 0 in a Newton iteration the script builds the 64 lane addresses (in doubles) exactly as the kernel indexes them and reports the
cycles a conflict-free access would take, the cycles this one takes, and how often it is issued per iteration at N = 12.        python tools/lds_bank_model.py
"""
import numpy as np

lane = np.arange(64)
lg, lc = lane >> 3, lane & 7
qr, qI, qJ, qc = lane >> 4, (lane >> 3) & 1, (lane >> 2) & 1, lane & 3
qR, qC = 4 * qI + qr, 4 * qJ + qc


def cycles(idx, write):
    """LDS-array cycles of one wave-instruction on doubles at indices idx (per lane)."""
    groups = [range(16 * m, 16 * m + 16) for m in range(4)] if write else [range(0, 32), range(32, 64)]
    nb = 32 if write else 64
    tot = 0
    for gr in groups:
        per_bank = {}
        for l in gr:
            for half in (0, 1):
                per_bank.setdefault((2 * int(idx[l]) + half) % nb, set()).add(int(idx[l]))
        tot += max(len(v) for v in per_bank.values())
    return tot, len(groups)


N = 12
pat = [
    # name, indices, write?, issued per Newton iteration (wave 0, four-wave kernel, N = 12)
    ("stage operand [A_k | B_k] (B form)  AB[qr*8 + qC]", qr * 8 + qC, False, N),
    ("stage operand [A_k | B_k] (A form)  AB[qr*8 + 4qI + qc]", qr * 8 + 4 * qI + qc, False, N),
    ("stage operand rows 4, 5 (B form)    AB[(4 + qr|0)*8 + qC]", np.where(qr < 2, 4 + qr, 0) * 8 + qC, False, N),
    ("stage operand [B; I] as A operand   AB[cA*8 + 6 + (qr & 1)]", np.where(4 * qI + qc < 6, 4 * qI + qc, 0) * 8 + 6 + (qr & 1), False, N),
    ("stage Hessian top                   AB[qR*8 + qC] (clamped)", np.where(qR < 6, qR, 0) * 8 + np.where(qC < 6, qC, 0), False, N),
    ("barrier weights kap / th            uniform address (broadcast)", np.zeros(64, int), False, 6 * N),
    ("store Phi_k / Pi_k tile, quad form  T[qR*8 + qC]", qR * 8 + qC, True, 2 * N),
    ("the same tile with row stride 10    T[qR*10 + qC]", qR * 10 + qC, True, 0),
    ("sweep operand read-back, even k     Phi[lg*8 + lc]", lg * 8 + lc, False, N // 2),
    ("sweep operand read-back, odd k      Phi[lc*8 + lg] (transposed)", lc * 8 + lg, False, N // 2),
    ("gamma / phi per stage, sum-over-c   v[k*8 + lg]  (8-fold broadcast)", lg, False, 3 * N // 2),
    ("gamma / phi per stage, sum-over-g   v[k*8 + lc]", lc, False, 3 * N // 2),
    ("sweep result store (branch-free)    writers -> element, others -> dump[lane]", np.where(lc == 0, lg, 64 + lane), True, 3 * N),
    ("terminal block column of M          Mt[(lane)*8 + j]  (one j per instruction)", lane * 8, True, 8),
    ("Gram operands (4x4x4 MFMA)          Mt[k*8 + 4(b>>1) + i + 32 s]", (lane >> 4) * 8 + 4 * (((lane >> 2) & 3) >> 1) + (lane & 3), False, 32),
    ("M c~ partial sums                   Mt[lc*8 + lg + 64 q]", lc * 8 + np.minimum(lg, 6), False, 8),
    ("R^-1 products                       Ri[lc*7 + lg] / Ri[lg*7 + lc]", np.where((lg < 7) & (lc <= lg), lc * 7 + lg, 0), False, 6),
]
print("%-78s %5s %5s %7s %9s" % ("access (wave 0)", "ideal", "real", "factor", "per iter"))
extra = tot = 0
for name, idx, wr, cnt in pat:
    c, g = cycles(np.asarray(idx), wr)
    print("%-78s %5d %5d %6.1fx %9d" % (name, g, c, c / g, cnt))
    extra += (c - g) * cnt; tot += c * cnt
print("conflict cycles per Newton iteration in these accesses: %d of %d LDS-array cycles (%.0f %%)" % (extra, tot, 100.0 * extra / tot))
