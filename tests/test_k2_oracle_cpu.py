"""The Kraken2-style oracle (oracle/k2_oracle.c) against tests/golden/k2_kat.json.

The vectors come from tests/golden/make_k2_golden.py: a per-k-mer brute force written from the definitions, hand-made
trees with hand-computed calls.  PARITY UNPINNED: neither side is Kraken 2 itself (oracle/k2_oracle.h).
"""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def kat():
    with open(os.path.join(HERE, "golden", "k2_kat.json")) as f:
        return json.load(f)


def test_fmix64(oracle, kat):
    for x, y in kat["fmix64"]:
        assert oracle.lib().k2o_hash(int(x)) == int(y)


def test_default_options(oracle):
    o = oracle.k2_default_opts()
    assert (o.k, o.l, o.value_bits, o.min_hit_groups, o.confidence) == (35, 31, 17, 2, 0.0)
    assert bin(o.spaced_seed_mask)[2:] == "1" * 34 + "0011" * 7         # --minimizer-spaces 7 over l = 31
    assert o.toggle_mask == 0xe37e28c4271b5a2d


def test_scanner_against_brute_force(oracle, kat):
    n_amb = 0
    for c in kat["scan"]:
        o = oracle.k2_default_opts()
        o.k, o.l, o.spaced_seed_mask, o.toggle_mask = c["k"], c["l"], int(c["spaced"]), int(c["toggle"])
        mins, amb = oracle.k2_scan(c["seq"].encode(), o)
        assert len(mins) == len(c["min"]) == max(len(c["seq"]) - c["k"] + 1, 0)
        for i, want in enumerate(c["min"]):
            if want is None:
                assert amb[i] == 1, (c["k"], c["l"], i)
                n_amb += 1
            else:
                assert amb[i] == 0 and int(mins[i]) == int(want), (c["k"], c["l"], i)
    assert n_amb > 100


def test_compact_hash_table(oracle, kat):
    c = kat["cht"]
    t = oracle.K2Table.empty(c["capacity"], np.zeros(2, np.uint32), c["value_bits"])
    for k, v in zip(c["keys"], c["values"]):
        assert t.set(int(k), v, lca=False) == 1
    assert t.cells.tolist() == c["cells"]
    first = {}
    for k, v in zip(c["keys"], c["values"]):
        first.setdefault(int(k), v)
    for k in c["keys"]:
        got = t.get(int(k))
        assert got != 0
        # a key whose truncated hash collides with an earlier one on the same probe path reads that one's value
        assert got == first[int(k)] or got in c["values"]
    hits = sum(t.get(int(k)) != 0 for k in c["absent"])
    assert hits <= 2                                                    # 23-bit truncated keys: false positives are rare


def test_table_full_and_wraparound(oracle):
    t = oracle.K2Table.empty(5, np.zeros(2, np.uint32), 9)
    keys = [3, 9, 27, 81, 243]
    for k in keys:
        assert t.set(k, 7, lca=False) == 1
    assert all(c & 511 for c in t.cells)
    assert t.set(729, 7, lca=False) in (0, 1)                           # full: either collides with a stored truncated key or fails
    assert all(t.get(k) == 7 for k in keys)
    assert t.get(123456789) in (0, 7)                                   # terminates after one lap


def test_lca_and_ancestor(oracle, kat):
    p = np.array(kat["tree"]["parent"], dtype=np.uint32)
    for a, b, want in kat["tree"]["lca"]:
        assert oracle.k2_lca(p, a, b) == want == oracle.k2_lca(p, b, a)
    for a, b, want in kat["tree"]["anc"]:
        assert oracle.k2_is_ancestor(p, a, b) == bool(want)


def test_resolve_tree(oracle, kat):
    p = np.array(kat["tree"]["parent"], dtype=np.uint32)
    for c in kat["tree"]["resolve"]:
        for order in (1, -1):                                           # the call does not depend on the hit map's order
            got = oracle.k2_resolve(c["taxa"][::order], c["counts"][::order], p, c["total"], c["conf"])
            assert got == c["call"], c["why"]


def test_lca_on_insert(oracle, kat):
    p = np.array(kat["tree"]["parent"], dtype=np.uint32)
    t = oracle.K2Table.empty(101, p, 9)
    t.set(12345, 7); t.set(12345, 8)
    assert t.get(12345) == 4
    t.set(12345, 9)
    assert t.get(12345) == 1


def _toy(oracle):
    """Table over the KAT tree with the minimizers of two sequences: `a` -> taxon 7, `b` -> taxon 8, shared ones -> 4."""
    rng = np.random.default_rng(5)
    p = np.array([0, 0, 1, 1, 2, 2, 3, 4, 4, 6], dtype=np.uint32)
    o = oracle.k2_default_opts()
    o.value_bits = 9
    a = bytes(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 400)])
    b = a[:200] + bytes(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 200)])
    t = oracle.K2Table.empty(4099, p, 9)
    for seq, tax in ((a, 7), (b, 8)):
        mins, amb = oracle.k2_scan(seq, o)
        for m in np.unique(mins[amb == 0]):
            t.set(int(m), tax)
    return t, o, a, b


def test_classify_read_and_pair(oracle):
    t, o, a, b = _toy(oracle)
    r, taxa = t.classify_pair(o, a[250:400], want_taxa=True)            # only in a -> 7
    assert r["call"] == 7 and r["total_kmers"] == 116 == len(taxa) and set(taxa.tolist()) == {7}
    r = t.classify_pair(o, b[250:400])
    assert r["call"] == 8
    r, taxa = t.classify_pair(o, a[20:170], want_taxa=True)             # shared prefix: every minimizer has LCA 4
    assert r["call"] == 4 and set(taxa.tolist()) == {4}
    r, taxa = t.classify_pair(o, a[120:270], want_taxa=True)            # spans the fork: 4 then 7 -> 7 (root-to-leaf path)
    assert r["call"] == 7 and {4, 7} <= set(taxa.tolist())
    r, taxa = t.classify_pair(o, a[250:400], b[250:400], want_taxa=True)   # mates disagree: 7 vs 8, equal weight -> LCA 4
    assert r["call"] == 4 and r["total_kmers"] == 232 and len(taxa) == 233 and taxa[116] == oracle.K2_BORDER
    rnd = bytes(np.frombuffer(b"ACGT", np.uint8)[np.random.default_rng(9).integers(0, 4, 150)])
    r = t.classify_pair(o, rnd)
    assert r["call"] == 0 and r["hit_groups"] == 0 and r["n_probes"] > 10
    assert t.classify_pair(o, b"")["total_kmers"] == 0 and t.classify_pair(o, a[:34])["total_kmers"] == 0


def test_ambiguous_bases_and_hit_groups(oracle):
    t, o, a, b = _toy(oracle)
    s = bytearray(a[250:400]); s[75] = ord("N")
    r, taxa = t.classify_pair(o, bytes(s), want_taxa=True)
    assert r["total_kmers"] == 116 and int((taxa == oracle.K2_AMBIG).sum()) == 31      # l k-mers wait for a full l-mer again
    assert r["call"] == 7
    # one hit group only: the call is voided by minimum_hit_groups = 2 (and kept with 1)
    mins, amb = oracle.k2_scan(a[250:400], o)
    first_run = int(np.argmax(mins != mins[0]))                        # k-mers sharing the first minimizer
    short = a[250:250 + 35 + first_run - 1]
    r = t.classify_pair(o, short)
    assert r["hit_groups"] == 1 and r["call"] == 0
    o.min_hit_groups = 1
    assert t.classify_pair(o, short)["call"] == 7


def test_down_sampling_threshold(oracle):
    t, o, a, b = _toy(oracle)
    full = t.classify_pair(o, a[250:400])
    o.min_acceptable_hash = 1 << 63                                    # about half of the minimizers are never looked up
    half = t.classify_pair(o, a[250:400])
    assert 0 < half["n_probes"] < full["n_probes"] and half["hit_groups"] < full["hit_groups"]


def test_batch_equals_single(oracle):
    t, o, a, b = _toy(oracle)
    recs = [a[i:i + 150] for i in range(0, 250, 10)] + [b[i:i + 150] for i in range(0, 250, 10)]
    bases = np.frombuffer(b"".join(recs), dtype=np.uint8)
    offs = np.arange(len(recs) + 1, dtype=np.uint64) * 150
    single = t.classify(o, bases, offs, paired=False, threads=3)
    for i, r in enumerate(recs):
        assert t.classify_pair(o, r) == {n: int(single[i][n]) for n in oracle.K2_RESULT_DTYPE.names}
    paired = t.classify(o, bases, offs, paired=True, threads=2)
    for i in range(len(recs) // 2):
        assert t.classify_pair(o, recs[2 * i], recs[2 * i + 1]) == {n: int(paired[i][n]) for n in oracle.K2_RESULT_DTYPE.names}


# ---- a database written byte by byte by plain Python (tests/golden/make_k2_pydb.py): file formats + classification, independently ----
def load_pydb():
    """parse tests/golden/k2_pydb/{opts,taxo,hash}.k2d with numpy/struct, from the published layouts"""
    import struct
    d = os.path.join(os.path.dirname(__file__), "golden", "k2_pydb")
    raw = open(os.path.join(d, "opts.k2d"), "rb").read()
    assert len(raw) == 64
    k, l, spaced, toggle, dna, min_hash, revcom, dbv, dbt = struct.unpack("<QQQQB7xQiii4x", raw)
    raw = open(os.path.join(d, "taxo.k2d"), "rb").read()
    assert raw[:8] == b"K2TAXDAT"
    n_nodes, n_name, n_rank = struct.unpack("<QQQ", raw[8:32])
    nodes = np.frombuffer(raw, dtype="<u8", count=7 * n_nodes, offset=32).reshape(n_nodes, 7)
    assert len(raw) == 32 + 56 * n_nodes + n_name + n_rank
    raw = open(os.path.join(d, "hash.k2d"), "rb").read()
    cap, size, kb, vb = struct.unpack("<QQQQ", raw[:32])
    cells = np.frombuffer(raw, dtype="<u4", count=cap, offset=32)
    assert len(raw) == 32 + 4 * cap and kb + vb == 32
    return dict(k=k, l=l, spaced=spaced, toggle=toggle, dna=dna, min_hash=min_hash, cells=cells, value_bits=vb, size=size,
                parent=nodes[:, 0].astype(np.uint32), external=nodes[:, 5].astype(np.uint32))


def pydb_units():
    exp = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "k2_pydb_expected.json")))
    return exp


def test_oracle_on_the_python_written_database(oracle):
    db, exp = load_pydb(), pydb_units()
    assert (db["k"], db["l"], db["value_bits"], db["size"]) == (exp["k"], exp["l"], exp["value_bits"], exp["size"]) and db["dna"] == 1
    assert list(db["parent"]) == exp["parents"] and list(db["external"]) == exp["external"]
    assert int((db["cells"] != 0).sum()) == exp["size"]
    t = oracle.K2Table(db["cells"], db["parent"], db["value_bits"])
    assert len(exp["units"]) == 64
    n_called = 0
    for u in exp["units"]:
        o = oracle.k2_default_opts()
        o.k, o.l, o.spaced_seed_mask, o.toggle_mask, o.value_bits = db["k"], db["l"], db["spaced"], db["toggle"], db["value_bits"]
        o.confidence, o.min_hit_groups = u["confidence"], u["min_hit_groups"]
        m = [s.encode() for s in u["mates"]]
        r = t.classify_pair(o, m[0], m[1] if len(m) > 1 else None)
        assert (r["call"], r["total_kmers"], r["hit_groups"]) == (u["call"], u["total_kmers"], u["hit_groups"]), u["what"]
        n_called += r["call"] != 0
    assert n_called > 30
