#!/usr/bin/env python3
"""Known-answer vectors for the extension stage's aligner (oracle/mm_align.c mma_ksw_extd2), authored by this build - the reference
holds none.  The expected values do NOT come from the code under test: they are computed here by a plain
O(nm) dual-affine-gap dynamic programme written from the recurrence alone (H, E, F, E2, F2 over the full matrix, global start,
free end), so the difference-encoded, anti-diagonal, banded restatement in the oracle is pinned against an independent statement of
what ksw2's extension alignment computes: the end-to-end score, the best score anywhere, the best score with the query consumed
(mqe) and with the target consumed (mte).  usage: make_align_golden.py > align_kat.json"""
import json
import numpy as np

NEG = -0x40000000


def plain(qs, ts, a, b, amb, q, e, q2, e2):
    def sc(x, y):
        return -amb if (x > 3 or y > 3) else (a if x == y else -b)
    n, m = len(ts), len(qs)
    H = np.full((n + 1, m + 1), NEG, np.int64); E = H.copy(); F = H.copy(); E2 = H.copy(); F2 = H.copy()
    H[0, 0] = 0
    for i in range(1, n + 1):
        H[i, 0] = -min(q + i * e, q2 + i * e2)
    for j in range(1, m + 1):
        H[0, j] = -min(q + j * e, q2 + j * e2)
    for i in range(1, n + 1):
        for j in range(1, m + 1):
            E[i, j] = max(H[i - 1, j] - q, E[i - 1, j]) - e
            E2[i, j] = max(H[i - 1, j] - q2, E2[i - 1, j]) - e2
            F[i, j] = max(H[i, j - 1] - q, F[i, j - 1]) - e
            F2[i, j] = max(H[i, j - 1] - q2, F2[i, j - 1]) - e2
            H[i, j] = max(H[i - 1, j - 1] + sc(ts[i - 1], qs[j - 1]), E[i, j], F[i, j], E2[i, j], F2[i, j])
    H = H[1:, 1:]
    return {"score": int(H[n - 1, m - 1]), "max": max(int(H.max()), 0), "mqe": int(H[:, m - 1].max()), "mte": int(H[n - 1, :].max())}


def main():
    rng = np.random.default_rng(20261003)
    cases = []
    scores = [(2, 8, 1, 12, 2, 24, 1), (2, 4, 1, 4, 2, 24, 1), (1, 4, 1, 6, 2, 26, 1)]      # sr, map-ont, map-hifi
    for it in range(120):
        m, n = int(rng.integers(1, 48)), int(rng.integers(1, 64))
        ts = rng.integers(0, 4, n)
        if it % 3:
            st = int(rng.integers(0, max(1, n - 1)))
            out = []
            for c in ts[st:st + m]:
                r = rng.random()
                if r < 0.06:
                    out.append(int(rng.integers(0, 4)))
                elif r < 0.10:
                    continue
                elif r < 0.14:
                    out += [int(c), int(rng.integers(0, 4))]
                else:
                    out.append(int(c))
            qs = np.array(out[:m] if out else [0])
        else:
            qs = rng.integers(0, 4, m)
        if it % 7 == 0:
            qs[int(rng.integers(0, len(qs)))] = 4
        a, b, amb, q, e, q2, e2 = scores[it % 3]
        cases.append({"query": [int(x) for x in qs], "target": [int(x) for x in ts], "a": a, "b": b, "sc_ambi": amb, "q": q, "e": e, "q2": q2, "e2": e2,
                      "expect": plain(list(qs), list(ts), a, b, amb, q, e, q2, e2)})
    print(json.dumps({"comment": __doc__.split("\n")[0], "cases": cases}))


if __name__ == "__main__":
    main()
