#!/usr/bin/env python3
"""Known-answer vectors for the Kraken2-style path (tests/golden/k2_kat.json).

Everything here is derived in plain Python from the DEFINITIONS in SURVEY.md Appendix B, independently of
oracle/k2_oracle.c and of the HIP kernels (no state machine, no deque: each k-mer is looked at on its own):
  * fmix64: MurmurHash3's 64-bit finaliser;
  * minimizer of the k-mer ending at p: a = last ambiguous position <= p; ambiguous if p - a < l; else the minimum of
    (canonical(l-mer) & spaced_mask) ^ toggle over the complete l-mers ending in [max(p - (k - l), a + l), p], XOR toggle;
  * compact hash table: dict-free linear probing written out longhand;
  * ResolveTree on hand-made trees with hand-computed expectations (stated beside each case).
Run: python tests/golden/make_k2_golden.py   (rewrites k2_kat.json)
"""
import json
import os
import random

M64 = (1 << 64) - 1


def fmix64(k):
    k ^= k >> 33; k = k * 0xff51afd7ed558ccd & M64
    k ^= k >> 33; k = k * 0xc4ceb9fe1a85ec53 & M64
    k ^= k >> 33
    return k


CODE = {"A": 0, "C": 1, "G": 2, "T": 3, "a": 0, "c": 1, "g": 2, "t": 3}


def lmer_value(s):
    v = 0
    for ch in s:
        v = v << 2 | CODE[ch]
    return v


def revcomp(s):
    return "".join("ACGT"[3 - CODE[c]] for c in reversed(s))


def scan(seq, k, l, spaced, toggle):
    out = []
    for p in range(k - 1, len(seq)):
        a = max([i for i in range(p + 1) if seq[i] not in CODE], default=-1)
        if p - a < l:
            out.append(None)
            continue
        best = None
        for e in range(max(p - (k - l), a + l), p + 1):
            s = seq[e - l + 1:e + 1]
            canon = min(lmer_value(s), lmer_value(revcomp(s)))
            if spaced:
                canon &= spaced
            cand = canon ^ toggle
            best = cand if best is None or cand < best else best
        out.append(best ^ toggle)
    return out


def main():
    rng = random.Random(20260115)
    spaced = (0x3ffffffff << 28) | 0x3333333
    toggle = 0xe37e28c4271b5a2d
    kat = {"fmix64": [[str(x), str(fmix64(x))] for x in [0, 1, 2, 0xdeadbeef, (1 << 62) - 1, M64, 0x123456789abcdef0]]}
    seqs = []
    for n in [20, 34, 35, 36, 60, 150, 151]:
        seqs.append("".join(rng.choice("ACGT") for _ in range(n)))
    s = list("".join(rng.choice("ACGT") for _ in range(150))); s[10] = "N"; seqs.append("".join(s))
    s = list("".join(rng.choice("ACGT") for _ in range(150))); s[70] = "N"; s[71] = "n"; s[120] = "R"; seqs.append("".join(s))
    s = list("".join(rng.choice("ACGT") for _ in range(150))); s[149] = "N"; seqs.append("".join(s))
    seqs += ["A" * 80, "ACGT" * 30, ("acgtTTGACA" * 12).lower(), "N" * 60, "ACGTN" * 20]
    kat["scan"] = []
    for k, l, sp in [(35, 31, spaced), (35, 31, 0), (31, 31, spaced), (25, 17, 0)]:
        for q in seqs:
            r = scan(q, k, l, sp, toggle)
            kat["scan"].append({"k": k, "l": l, "spaced": str(sp), "toggle": str(toggle), "seq": q,
                                "min": [None if x is None else str(x) for x in r]})
    # compact hash table: capacity 97, value_bits 9; insertion order as listed; value 0 never stored
    cap, vb = 97, 9
    cells = [0] * cap
    keys = [rng.getrandbits(62) for _ in range(60)]
    vals = [rng.randint(1, (1 << vb) - 1) for _ in keys]
    for key, v in zip(keys, vals):
        hc = fmix64(key)
        comp = hc >> (32 + vb)
        i = hc % cap
        while True:
            if cells[i] & ((1 << vb) - 1) == 0:
                cells[i] = comp << vb | v
                break
            if cells[i] >> vb == comp:
                break                       # first value wins (inserted without a taxonomy)
            i = (i + 1) % cap
    kat["cht"] = {"capacity": cap, "value_bits": vb, "keys": [str(x) for x in keys], "values": vals, "cells": cells,
                  "absent": [str(rng.getrandbits(62)) for _ in range(20)]}
    # ResolveTree.  Tree (internal ids, parent < child):  1 root; 2,3 children of 1; 4,5 of 2; 6 of 3; 7,8 of 4; 9 of 6
    parent = [0, 0, 1, 1, 2, 2, 3, 4, 4, 6]
    kat["tree"] = {"parent": parent,
                   "lca": [[7, 8, 4], [7, 5, 2], [7, 9, 1], [4, 7, 4], [0, 5, 5], [6, 0, 6], [9, 9, 9], [1, 8, 1]],
                   "anc": [[1, 9, 1], [4, 8, 1], [8, 4, 0], [5, 7, 0], [3, 3, 1], [0, 3, 0], [2, 6, 0]],
                   "resolve": [
                       # path scores: 7 -> hits(7)+hits(4)+hits(2)+hits(1) ...
                       {"taxa": [7], "counts": [10], "total": 100, "conf": 0.0, "call": 7, "why": "single taxon"},
                       {"taxa": [7, 8], "counts": [5, 5], "total": 100, "conf": 0.0, "call": 4, "why": "tie between siblings -> their LCA"},
                       {"taxa": [7, 8, 4], "counts": [5, 5, 3], "total": 100, "conf": 0.0, "call": 4, "why": "7 and 8 both score 8 -> LCA 4"},
                       {"taxa": [7, 8, 4], "counts": [6, 5, 3], "total": 100, "conf": 0.0, "call": 7, "why": "7 scores 9 > 8"},
                       {"taxa": [2, 9], "counts": [4, 3], "total": 100, "conf": 0.0, "call": 2, "why": "2 scores 4, 9 scores 3"},
                       {"taxa": [7, 9], "counts": [3, 3], "total": 100, "conf": 0.0, "call": 1, "why": "tie across the root"},
                       {"taxa": [7, 4, 2], "counts": [2, 3, 4], "total": 20, "conf": 0.3, "call": 2,
                        "why": "required 6: clade(7)=2, clade(4)=5, clade(2)=9 -> climbs to 2"},
                       {"taxa": [7, 4, 2], "counts": [2, 3, 4], "total": 20, "conf": 0.5, "call": 0,
                        "why": "required 10: clade(1)=9 < 10 -> unclassified"},
                       {"taxa": [7], "counts": [2], "total": 20, "conf": 0.1, "call": 7, "why": "required 2 met at the call itself"},
                       {"taxa": [], "counts": [], "total": 50, "conf": 0.0, "call": 0, "why": "no hits"},
                   ]}
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "k2_kat.json"), "w") as f:
        json.dump(kat, f, indent=0)
    print("wrote", os.path.join(here, "k2_kat.json"), "scan cases:", len(kat["scan"]))


if __name__ == "__main__":
    main()
