"""Zero structure of the pivot columns during the in-place Gauss-Jordan inversion of the collocation block of an interval with several finite
elements (csrc/dompc_factor.h, register-resident elimination): at step kk column kk is nonzero only in the rows [grp0, hi) the kernel visits,
and skipping the last NX columns leaves the inverse.  Random matrices with the block structure of optimizer.py:943-983."""
import numpy as np
rng = np.random.default_rng(0)
for (NX, DEG, NI) in ((4, 2, 2), (3, 3, 3), (10, 3, 2)):
    ER = (DEG + 1) * NX
    NW = NI * ER
    slot_of = lambda i, r: r - 1 if i == 0 else DEG + (i - 1) * (DEG + 1) + r
    next_slot = lambda i: slot_of(i + 1, 0) if i + 1 < NI else NI * (DEG + 1) - 1
    G = np.zeros((NW, NW))
    for i in range(NI):
        for jj in range(DEG + 1):
            for a in range(NX):
                row = i * ER + jj * NX + a
                if jj < DEG:
                    sl = slot_of(i, jj + 1)
                    G[row, sl * NX:(sl + 1) * NX] += rng.standard_normal(NX) * 0.1
                    for r in range(DEG + 1):
                        if not (i == 0 and r == 0):
                            G[row, slot_of(i, r) * NX + a] -= rng.standard_normal() + 3 * (r == jj + 1)
                else:
                    G[row, next_slot(i) * NX + a] += 1
                    for r in range(DEG + 1):
                        if not (i == 0 and r == 0):
                            G[row, slot_of(i, r) * NX + a] -= rng.standard_normal()
    A = G.copy()
    ok = True
    for kk in range(NW - NX):
        pos = kk % ER
        grp0 = kk - pos + (0 if pos < DEG * NX else DEG * NX)
        hi = kk - pos + ER if pos < DEG * NX else min(NW, kk - pos + 2 * ER)
        f = A[:, kk].copy()
        nz = np.nonzero(f)[0]
        ok &= nz.min() >= grp0 and nz.max() < hi
        prow = A[kk, :] / f[kk]
        prow[kk] = 1 / f[kk]
        for r in range(NW):
            if r != kk:
                keep = A[r, :].copy()
                keep[kk] = 0
                A[r, :] = keep - f[r] * prow
        A[kk, :] = prow
    print((NX, DEG, NI), "pivot columns inside [grp0, hi):", bool(ok), " |result - inverse| max", np.abs(A - np.linalg.inv(G)).max())
    assert ok and np.abs(A - np.linalg.inv(G)).max() < 1e-12
