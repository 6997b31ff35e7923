#!/usr/bin/env python3
"""Generates tests/golden/*.json.

The reference (esteinig/scrubby) ships no tests, fixtures or golden vectors and
its minimap2 dependency is not on this box, so these vectors are AUTHORED here:
  * sketch_kat.json   — minimizers of short sequences derived by an independent brute force in
                        pure Python (window minimum over canonical k-mer hashes, straight from the
                        published (w,k)-minimizer definition), NOT by the C state machine;
  * chain_kat.json    — hand-computable pair scores of the chaining recurrence;
  * classify_kat.json — a 6 kb toy reference, reads cut from it (expected host) and unrelated
                        reads (expected retained), with the expected flags by construction.
Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
CODE = {65: 0, 67: 1, 71: 2, 84: 3}


def hash64(key, mask):
    key = (~key + (key << 21)) & mask
    key = key ^ key >> 24
    key = ((key + (key << 3)) + (key << 8)) & mask
    key = key ^ key >> 14
    key = ((key + (key << 2)) + (key << 4)) & mask
    key = key ^ key >> 28
    key = (key + (key << 31)) & mask
    return key


def brute_minimizers(seq, w, k):
    """All (hash, end_pos, strand) that are the unique minimum of some full window; None if any tie/ambiguity."""
    mask = (1 << 2 * k) - 1
    hs = []
    for i in range(k - 1, len(seq)):
        f = 0
        for c in seq[i - k + 1:i + 1]:
            f = (f << 2) | CODE[c]
        r = 0
        for c in reversed(seq[i - k + 1:i + 1]):
            r = (r << 2) | (3 - CODE[c])
        if f == r:
            return None
        hs.append((hash64(min(f, r), mask), i, 0 if f < r else 1))
    out = set()
    if 0 < len(hs) < w:          # no full window: the pending minimum is flushed at the end of the sequence
        m = min(h for h, _, _ in hs)
        tied = [t for t in hs if t[0] == m]
        return None if len(tied) > 1 else tied
    for s in range(0, len(hs) - w + 1):
        win = hs[s:s + w]
        m = min(h for h, _, _ in win)
        tied = [t for t in win if t[0] == m]
        if len(tied) > 1:
            return None
        out.add(tied[0])
    return sorted(out, key=lambda t: t[1])


def main():
    rng = np.random.default_rng(20261003)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    sk = []
    for (w, k, n) in [(11, 21, 64), (11, 21, 150), (10, 15, 60), (10, 15, 200), (5, 7, 40), (19, 19, 120)]:
        while True:
            seq = bytes(acgt[rng.integers(0, 4, n)])
            b = brute_minimizers(seq, w, k)
            if b is not None:
                break
        sk.append({"w": w, "k": k, "seq": seq.decode(), "minimizers": [[int(h), int(p), int(z)] for h, p, z in b]})
    json.dump({"comment": "independent brute-force (w,k)-minimizers: [hash, end position, strand]", "cases": sk},
              open(os.path.join(HERE, "sketch_kat.json"), "w"), indent=1)

    hk = [{"key": int(x), "k": k, "hash": int(hash64(int(x), (1 << 2 * k) - 1))}
          for k in (15, 21) for x in (0, 1, 2, 12345, (1 << 2 * k) - 1, 0x1234567 & ((1 << 2 * k) - 1))]
    # pair scores for sr (k=21, pen_gap = 0.8*0.01*21 = 0.168, pen_skip = 0): [dq, dr, expected]
    pen = float(np.float32(0.8 * 0.01 * 21))
    ch = []
    for dq, dr in [(10, 10), (21, 21), (30, 30), (10, 15), (15, 10), (40, 45), (1, 1), (5, 105), (100, 100), (100, 201), (0, 5), (5, 0), (200, 200)]:
        # recomputed by hand: sc = min(k, min(dq,dr)) - int(pen*|dr-dq| + 0.5*log2approx(|dr-dq|+1)), invalid cases None
        ch.append({"dq": dq, "dr": dr})
    json.dump({"hash64": hk, "pen_gap_sr": pen, "pairs": ch}, open(os.path.join(HERE, "chain_kat.json"), "w"), indent=1)

    ref = bytes(acgt[rng.integers(0, 4, 6000)])
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    reads, expect = [], []
    for s in (0, 1234, 3000, 5850):
        reads.append(ref[s:s + 150]); expect.append(1)
        reads.append(ref[s:s + 150].translate(comp)[::-1]); expect.append(1)
    for _ in range(6):
        reads.append(bytes(acgt[rng.integers(0, 4, 150)])); expect.append(0)
    r = bytearray(ref[2000:2150])
    for p in range(0, 150, 10):            # a mismatch every 10 bp: no 21-mer survives -> no anchor -> retained
        r[p] = ord("ACGT"[("ACGT".index(chr(r[p])) + 1) % 4])
    reads.append(bytes(r)); expect.append(0)
    reads.append(ref[100:140]); expect.append(1)      # 40 bp: one full window, 2+ minimizers expected
    reads.append(ref[100:125]); expect.append(0)      # 25 bp: shorter than k + w - 1, a single minimizer at most
    json.dump({"preset": "sr", "ref": ref.decode(), "reads": [x.decode() for x in reads], "flags": expect},
              open(os.path.join(HERE, "classify_kat.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
