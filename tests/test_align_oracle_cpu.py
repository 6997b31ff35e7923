"""CPU tests of the extension stage's oracle (oracle/mm_align.c): the aligner against committed known-answer
vectors from an independent plain dynamic programme (tests/golden/make_align_golden.py), consistency of what it returns (the CIGAR it
backtracks scores what the matrix says), and the decision stage on reads built to make chains and regions part ways."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from tests import align_cases as AC
from tests import workloads as W

HERE = os.path.dirname(os.path.abspath(__file__))


class Ez(C.Structure):
    _fields_ = [("max", C.c_uint32), ("zdropped", C.c_int), ("max_q", C.c_int), ("max_t", C.c_int), ("mqe", C.c_int), ("mqe_t", C.c_int),
                ("mte", C.c_int), ("mte_q", C.c_int), ("score", C.c_int), ("reach_end", C.c_int), ("n_cigar", C.c_int), ("m_cigar", C.c_int),
                ("cigar", C.POINTER(C.c_uint32))]


def extd2(L, qs, ts, a, b, amb, q, e, q2, e2, w, zdrop, end_bonus, flag):
    mat = np.zeros(25, np.int8)
    L.mma_gen_simple_mat(5, mat.ctypes.data, a, b, amb)
    ez = Ez()
    qa, ta = np.ascontiguousarray(qs, np.uint8), np.ascontiguousarray(ts, np.uint8)
    L.mma_ksw_extd2(len(qa), qa.ctypes.data, len(ta), ta.ctypes.data, 5, mat.ctypes.data, q, e, q2, e2, w, zdrop, end_bonus, flag, C.byref(ez))
    return ez, [(ez.cigar[i] & 0xf, ez.cigar[i] >> 4) for i in range(ez.n_cigar)], mat


def cigar_score(cig, qs, ts, mat, q, e, q2, e2):
    i = j = sc = 0
    for op, ln in cig:
        if op == 0:
            sc += sum(int(mat[ts[i + l] * 5 + qs[j + l]]) for l in range(ln)); i += ln; j += ln
        else:
            sc -= min(q + ln * e, q2 + ln * e2)
            if op == 1:
                j += ln
            else:
                i += ln
    return sc, i, j


def test_ksw_against_the_independent_known_answers(oracle):
    L = oracle.lib()
    kat = json.load(open(os.path.join(HERE, "golden", "align_kat.json")))
    assert len(kat["cases"]) >= 100
    for c in kat["cases"]:
        for flag in (0, 0x40, 0x40 | 0x02 | 0x80, 0x02):          # global; extension; extension, gaps right-aligned, CIGAR reversed; right-aligned
            ez, cig, mat = extd2(L, c["query"], c["target"], c["a"], c["b"], c["sc_ambi"], c["q"], c["e"], c["q2"], c["e2"], 200, -1, 10 if flag & 0x40 else -1, flag)
            x = c["expect"]
            assert (ez.score, ez.max, ez.mqe, ez.mte) == (x["score"], x["max"], x["mqe"], x["mte"]), (c, flag)
            s2, i2, j2 = cigar_score(cig if not (flag & 0x80) else cig[::-1], c["query"], c["target"], mat, c["q"], c["e"], c["q2"], c["e2"])
            if flag & 0x40:
                end = (ez.mqe_t + 1, len(c["query"])) if ez.reach_end else (ez.max_t + 1, ez.max_q + 1)
                want = ez.mqe if ez.reach_end else ez.max
            else:
                end, want = (len(c["target"]), len(c["query"])), ez.score
            if cig:
                assert (i2, j2) == end and s2 == want, (c, flag, cig)


def test_ksw_zdrop_and_band(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(3)
    t = rng.integers(0, 4, 300).astype(np.uint8)
    qy = np.concatenate([t[:60], rng.integers(0, 4, 200).astype(np.uint8)])           # 60 matching bases, then noise: the extension must stop
    ez, cig, _ = extd2(L, qy, t, 2, 8, 1, 12, 2, 24, 1, 151, 100, 10, 0x40)
    assert ez.zdropped == 1 and 100 <= ez.max <= 130 and abs(ez.max_t - 59) <= 4 and abs(ez.max_q - 59) <= 4
    ez2, _, _ = extd2(L, t[:200], t[:200], 2, 8, 1, 12, 2, 24, 1, 3, -1, -1, 0)          # a narrow band still finds the diagonal
    assert ez2.score == 400 and ez2.zdropped == 0
    ez3, _, _ = extd2(L, t[:40], t[:200], 2, 8, 1, 12, 2, 24, 1, 5, -1, -1, 0)           # global alignment outside the band: given up
    assert ez3.zdropped == 1


@pytest.fixture(scope="module")
def cfg1_idx(oracle):
    P, R, ref, seqs, reads, off = W.cfg1(oracle, 2000)
    return ref, oracle.Index.build(seqs, 11, 21)


def test_regions_and_chains_part_ways_on_edge_reads(oracle, cfg1_idx):
    ref, idx = cfg1_idx
    recs, bases, offs = AC.edge_reads(ref, 4000)
    o1, o0 = oracle.preset("sr"), oracle.preset("sr")
    o0.flags = 0
    f1, t1 = idx.classify(o1, bases, offs, threads=8)
    f0, t0 = idx.classify(o0, bases, offs, threads=8)
    for name in ("n_mini", "n_seed", "n_anchor", "n_chain", "best_score"):
        assert np.array_equal(t1[name], t0[name])                                  # the stage does not touch chaining
    assert np.all(f1 <= f0) and np.array_equal(f0 == 1, t0["n_chain"] > 0) and np.array_equal(f1 == 1, t1["n_regs"] > 0)
    dropped = (f0 == 1) & (f1 == 0)
    assert 8 <= int(dropped.sum()) <= 0.05 * len(recs)                            # some chains do not survive, most do
    assert np.all(t1["n_aligned"][t1["n_chain"] > 0] >= 1) and np.all(t1["n_aligned"] <= t1["n_chain"])
    assert np.all(t1["dp_max"][f1 == 1] >= o1.min_dp_max) and np.all(t1["dp_max"][f1 == 0] == 0)
    kinds = np.arange(len(recs)) % 10
    assert int(((t1["n_regs"] > t1["n_aligned"]) & (kinds == 0)).sum()) >= 10      # z-drop splits in the garbage-middle reads
    assert int(dropped[kinds == 2].sum()) >= 4                                     # two cores on different diagonals: mlen < min_chain_score
    clean = (kinds == 7) | (kinds == 9) | (kinds == 4)
    assert int(dropped[clean].sum()) == 0                                          # plain reads with errors / indels always survive


def test_hand_made_decisions(oracle, cfg1_idx):
    """One exact 21-mer and nothing else: a minimizer, but a single anchor is no chain (min_cnt 2).  An exact 36-base copy: when its
    minimizers chain, the region is the whole read - 36 matches, dp_max 72: kept.  Two exact 21-mers on neighbouring diagonals with
    nothing around them to extend into: a chain (score >= 25) but mlen = 21 < min_chain_score: dropped by the stage."""
    ref, idx = cfg1_idx
    o = oracle.preset("sr")
    rng = np.random.default_rng(11)
    noise = lambda n: bytes(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n)])
    got = {"single": 0, "pair": 0, "shifted": 0}
    n_try = 0
    for s in range(50_000, 90_000, 997):
        n_try += 1
        core = bytes(ref[s:s + 36])
        r = idx.map(o, noise(60) + bytes(ref[s:s + 21]) + noise(60))
        got["single"] += r["flag"]
        r = idx.map(o, core)
        if r["n_chain"] == 1:
            got["pair"] += int(r["flag"] == 1 and r["n_regs"] == 1 and r["dp_max"] == 72)
        sh = bytes(ref[s:s + 21]) + noise(9) + bytes(ref[s + 31:s + 52])          # second k-mer one base off the first one's diagonal
        r = idx.map(o, sh)
        if r["n_chain"] == 1 and r["best_score"] >= 25:
            got["shifted"] += int(r["flag"] == 0 and r["n_regs"] == 0)
    assert got["single"] == 0 and got["pair"] >= 1 and got["shifted"] >= 1, (got, n_try)
