"""CPU suite: the oracle against the authored golden vectors and against an independent
re-derivation; the host-side logic; the C ABI's symbol table.  No GPU needed."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

from tests import workloads as W
from tests.golden.make_golden import brute_minimizers, hash64

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_hash64_golden(oracle):
    g = json.load(open(os.path.join(GOLD, "chain_kat.json")))
    for c in g["hash64"]:
        mask = (1 << 2 * c["k"]) - 1
        assert oracle.lib().mmo_hash64(c["key"], mask) == c["hash"] == hash64(c["key"], mask)


def test_hash64_is_invertible_on_2k_bits(oracle):
    k = 6                                      # 4096 keys: the mix must be a permutation
    mask = (1 << 2 * k) - 1
    img = {oracle.lib().mmo_hash64(x, mask) for x in range(mask + 1)}
    assert len(img) == mask + 1


def test_sketch_golden(oracle):
    g = json.load(open(os.path.join(GOLD, "sketch_kat.json")))
    for c in g["cases"]:
        x, y = oracle.sketch(c["seq"].encode(), c["w"], c["k"])
        got = [[int(a) >> 8, (int(b) & 0xffffffff) >> 1, int(b) & 1] for a, b in zip(x, y)]
        assert got == c["minimizers"]
        assert all((int(a) & 0xff) == c["k"] for a in x)


def test_sketch_vs_bruteforce_random(oracle):
    rng = np.random.default_rng(11)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    checked = 0
    for _ in range(200):
        seq = bytes(acgt[rng.integers(0, 4, int(rng.integers(30, 400)))])
        for w, k in ((11, 21), (10, 15), (5, 7), (19, 19)):
            b = brute_minimizers(seq, w, k)
            if b is None:
                continue
            x, y = oracle.sketch(seq, w, k)
            assert [(int(a) >> 8, (int(c) & 0xffffffff) >> 1, int(c) & 1) for a, c in zip(x, y)] == b
            checked += 1
    assert checked > 500


def test_sketch_strand_symmetry(oracle):
    """The minimizer hash multiset of a sequence equals that of its reverse complement."""
    rng = np.random.default_rng(5)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    for _ in range(50):
        s = bytes(acgt[rng.integers(0, 4, 300)])
        if brute_minimizers(s, 11, 21) is None:
            continue
        x1, y1 = oracle.sketch(s, 11, 21)
        x2, y2 = oracle.sketch(s.translate(comp)[::-1], 11, 21)
        assert sorted(x1.tolist()) == sorted(x2.tolist())
        # positions mirror: end position p on one strand <-> start position on the other
        p1 = sorted(((int(v) & 0xffffffff) >> 1) for v in y1)
        p2 = sorted((300 - 1 - (((int(v) & 0xffffffff) >> 1) - 21 + 1)) for v in y2)
        assert p1 == p2


def test_sketch_ambiguous_bases_reset(oracle):
    s = b"ACGTTGCATGCCGATAGCTAGCTAGGATCGATCGATTAGC"
    x0, _ = oracle.sketch(s, 5, 7)
    xs, ys = oracle.sketch(s + b"N" + s, 5, 7)
    # no k-mer spans the N; after the N the state machine restarts exactly like a fresh sequence,
    # while the pending minimum of the first copy may be dropped (its flush is gated on l >= w+k-1)
    pos = [((int(v) & 0xffffffff) >> 1) for v in ys]
    assert all(not (len(s) <= p < len(s) + 7) for p in pos)
    second = [int(a) for a, p in zip(xs, pos) if p > len(s)]
    first = [int(a) for a, p in zip(xs, pos) if p < len(s)]
    assert second == x0.tolist()
    assert first == x0.tolist()[:len(first)] and len(first) >= len(x0) - 2
    assert len(oracle.sketch(b"N" * 100, 11, 21)[0]) == 0
    assert len(oracle.sketch(s[:6], 5, 7)[0]) == 0
    assert oracle.sketch(s.lower(), 5, 7)[0].tolist() == x0.tolist()


def _log2_py(x):
    z = np.array([x], dtype=np.float32).view(np.uint32)[0]
    l2 = np.float32(np.int32((z >> 23) & 255) - 128)
    z = np.uint32((z & ~np.uint32(255 << 23)) + np.uint32(127 << 23))
    f = np.array([z], dtype=np.uint32).view(np.float32)[0]
    return np.float32(l2 + np.float32(np.float32(np.float32(np.float32(-0.34484843) * f) + np.float32(2.02466578)) * f) - np.float32(0.67487759))


def _sc_py(dq, dr, k, pen_gap, max_dist_x, max_dist_y, bw):
    if dq <= 0 or dq > max_dist_x or dr == 0 or dq > max_dist_y:
        return None
    dd = abs(dr - dq)
    if dd > bw:
        return None
    dg = min(dr, dq)
    sc = min(k, dg)
    if dd or dg > k:
        lin = np.float32(np.float32(pen_gap) * np.float32(dd)) + np.float32(0.0)
        lg = _log2_py(np.float32(dd + 1)) if dd >= 1 else np.float32(0)
        sc -= int(np.float32(lin + np.float32(np.float32(0.5) * lg)))
    return sc


def test_pair_score_golden(oracle):
    g = json.load(open(os.path.join(GOLD, "chain_kat.json")))
    pen = np.float32(g["pen_gap_sr"])
    for c in g["pairs"]:
        dq, dr = c["dq"], c["dr"]
        got = oracle.lib().mmo_comput_sc(1000 + dr, (21 << 32) | (500 + dq), 1000, (21 << 32) | 500, 650, 150, 100, pen, 0.0)
        exp = _sc_py(dq, dr, 21, pen, 650, 150, 100)
        assert got == (exp if exp is not None else -2 ** 31), (dq, dr, got, exp)
    # hand-checked anchors of the recurrence: collinear 10 apart -> +10; overlap-free 21+ apart -> span
    assert oracle.lib().mmo_comput_sc(1010, (21 << 32) | 510, 1000, (21 << 32) | 500, 650, 150, 100, pen, 0.0) == 10
    assert oracle.lib().mmo_comput_sc(1021, (21 << 32) | 521, 1000, (21 << 32) | 500, 650, 150, 100, pen, 0.0) == 21
    assert oracle.lib().mmo_comput_sc(1030, (21 << 32) | 530, 1000, (21 << 32) | 500, 650, 150, 100, pen, 0.0) == 21   # dg > span: log term of 1 -> 0


def test_classify_golden(oracle):
    g = json.load(open(os.path.join(GOLD, "classify_kat.json")))
    idx = oracle.Index.build([g["ref"].encode()], 11, 21)
    o = oracle.preset(g["preset"])
    for read, exp in zip(g["reads"], g["flags"]):
        assert idx.map(o, read.encode())["flag"] == exp


def test_presets(oracle):
    sr = oracle.preset("sr")
    assert (sr.k, sr.w, sr.min_cnt, sr.min_chain_score, sr.mid_occ, sr.max_occ, sr.bw, sr.max_gap, sr.max_frag_len) == \
        (21, 11, 2, 25, 1000, 5000, 100, 100, 800)
    ont = oracle.preset("map-ont")
    assert (ont.k, ont.w, ont.min_cnt, ont.min_chain_score, ont.mid_occ, ont.bw, ont.max_gap) == (15, 10, 3, 40, 0, 500, 5000)
    with pytest.raises(ValueError):
        oracle.preset("lr")          # Preset::Lr is rejected by the reference (cleaner.rs:469)


def test_index_positions_sorted_and_complete(oracle):
    P, R, ref, seqs, reads, off = W.cfg1(oracle, 10)
    idx = oracle.Index.build(seqs[:2], 11, 21)
    keys, cnt, pos = idx.dump()
    assert np.all(np.diff(keys.astype(np.int64)) > 0)
    # every contig's sketch is in the index with the right rid
    total = 0
    for rid, s in enumerate(seqs[:2]):
        x, y = oracle.sketch(s, 11, 21, rid=rid)
        total += len(x)
        assert np.all((y >> np.uint64(32)) == rid)
    assert total == int(cnt.sum()) == len(pos)
    o = 0
    for c in cnt:
        assert np.all(np.diff(pos[o:o + c].astype(np.int64)) > 0)
        o += int(c)


def test_cfg1_separates_host_from_nonhost(oracle):
    P, R, ref, seqs, reads, off = W.cfg1(oracle, 4000)
    idx = oracle.Index.build(seqs, 11, 21)
    flags, tr = idx.classify(oracle.preset("sr"), reads, off, threads=4)
    truth = oracle.synth_truth(P, R, 0, len(flags))
    assert np.array_equal(flags, truth)
    # decision margin the generator achieves: host chains score far above 25
    assert tr["best_score"][truth == 1].min() >= 40
    assert tr["n_anchor"][truth == 0].max() <= 1


def test_rechain_and_occurrence_filter_paths(oracle):
    """A tandem array whose k-mers occur > mid_occ times but <= max_occ: first pass filters every
    seed (rep_len > 0, no chain), the sr re-chain pass with max_occ recovers the mapping."""
    rng = np.random.default_rng(99)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    mono = bytes(acgt[rng.integers(0, 4, 171)])
    flank = bytes(acgt[rng.integers(0, 4, 5000)])
    ref = flank + mono * 1500 + flank[::-1]
    idx = oracle.Index.build([ref], 11, 21)
    o = oracle.preset("sr")
    t = idx.map(o, (mono * 2)[30:180])
    assert t["rechained"] == 1 and t["flag"] == 1 and t["n_anchor"] > 1000   # trace reports the last pass
    t2 = idx.map(o, flank[100:250])
    assert t2["rechained"] == 0 and t2["flag"] == 1
    ref3 = flank + mono * 6000 + flank[::-1]           # > max_occ: filtered in both passes -> unmapped
    t3 = oracle.Index.build([ref3], 11, 21).map(o, (mono * 2)[30:180])
    assert t3["rechained"] == 1 and t3["n_anchor"] == 0 and t3["flag"] == 0


def test_edge_reads(oracle):
    P, R, ref, seqs, reads, off = W.cfg1(oracle, 10)
    idx = oracle.Index.build(seqs, 11, 21)
    recs, bases, offs = W.edge_reads(ref)
    flags, tr = idx.classify(oracle.preset("sr"), bases, offs, threads=1)
    assert flags[0] == 2 and tr["n_mini"][0] == 0          # empty read
    assert flags[1] == 0 and tr["n_mini"][1] == 0          # shorter than k
    assert flags[5] == 0                                    # all N
    assert flags[6] == 1                                    # lower case


def test_threads_do_not_change_results(oracle):
    P, R, ref, seqs, reads, off = W.cfg1(oracle, 3000)
    idx = oracle.Index.build(seqs, 11, 21)
    o = oracle.preset("sr")
    f1, t1 = idx.classify(o, reads, off, threads=1)
    f8, t8 = idx.classify(o, reads, off, threads=8)
    assert np.array_equal(f1, f8) and np.array_equal(t1, t8)


# ---- the C ABI -------------------------------------------------------------------------------------
def test_c_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "scrubby_hip.h")).read()
    declared = set(re.findall(r"\b(sh_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"sh_status"}
    from scrubby_amd import lib
    assert os.path.exists(lib.LIB_PATH), "libscrubby_hip.so not built (python -c 'import __graft_entry__ as g; g.build()')"
    L = C.CDLL(lib.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, missing
    assert set(lib.EXPORTS) == declared


def test_c_abi_presets_and_errors_without_gpu():
    from scrubby_amd import lib
    L = lib.load()
    assert L.sh_version() == 104
    o = lib.preset("sr")
    assert (o.k, o.w, o.mid_occ, o.max_occ) == (21, 11, 1000, 5000)
    with pytest.raises(lib.ScrubbyHipError) as e:
        lib.preset("lr")
    assert e.value.status == lib.SH_ERR_PRESET_UNSUPPORTED and "lr" in e.value.message
    with pytest.raises(lib.ScrubbyHipError) as e:
        lib.preset("nonsense")
    assert e.value.status == lib.SH_ERR_PRESET_UNKNOWN


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under scrubby_amd/ may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "scrubby_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")):
                for line in open(os.path.join(dirpath, f)):
                    code = line.strip()
                    if code.startswith("#include"):
                        assert "oracle" not in code, (f, code)
                    if code.startswith(("import ", "from ")):
                        assert "oracle" not in code.split("#")[0], (f, code)
                    assert "libmm_oracle" not in code, (f, code)
