"""GPU parity of the Kraken2-style path: libscrubby_hip's sh_k2_* (through the C ABI) vs oracle/k2_oracle.c, bit-exact.

The table is built on the GPU, exported, and wrapped by the oracle, so both sides probe the same cells; every per-unit
integer (call, total_kmers, hit_groups) and the probe total are compared.  PARITY UNPINNED (oracle/k2_oracle.h).
"""
import gzip
import os

import numpy as np
import pytest

from tests import workloads as W

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def K():
    from scrubby_amd import lib, k2
    lib.require_gpu()
    return k2


@pytest.fixture(scope="module")
def cfg1(oracle):
    return W.cfg1(oracle, 20000)


@pytest.fixture(scope="module")
def tax():
    return W.k2_taxonomy()


@pytest.fixture(scope="module")
def db(K, cfg1, tax):
    """contig 0 -> Homo sapiens, its first 200 kb also -> Homo heidelbergensis (LCA Homo), contig 1 -> Pan troglodytes,
    contig 2 -> a bacterial species, 64 x 120-bp pieces of contig 3 -> 64 different taxa, + 300 k random filler keys."""
    P, R, ref, seqs, reads, off = cfg1
    parents, externals, names, ranks, ids = tax
    o = K.default_opts()
    d = K.K2Db.create(o, 6_000_011, parents, externals, names, ranks)
    n0 = d.insert_sequence(seqs[0], ids["Homo sapiens"])
    assert n0 > 150_000
    d.insert_sequence(seqs[0][:200_000], ids["Homo heidelbergensis"])
    d.insert_sequence(seqs[1], ids["Pan troglodytes"])
    bact_species = [i for i, r in enumerate(ranks) if r == "species" and i > ids["Bacteria"] and names[i].startswith("species_")]
    d.insert_sequence(seqs[2], bact_species[0])
    for j in range(64):
        d.insert_sequence(seqs[3][1000 + 120 * j: 1000 + 120 * (j + 1) + 34], bact_species[1 + j % (len(bact_species) - 1)])
    d.insert_random(0xC0FFEE, 300_000, ids["Bacteria"], len(parents) - 1)
    return d


@pytest.fixture(scope="module")
def table(oracle, db):
    cells, parent, ext = db.export()
    return oracle.K2Table(cells, parent, db.info()["value_bits"]), ext


def _same(oracle, g, c, ext):
    assert np.array_equal(g["call"], c["call"]), f"{int((g['call'] != c['call']).sum())} calls differ"
    assert np.array_equal(g["total_kmers"], c["total_kmers"])
    assert np.array_equal(g["hit_groups"], c["hit_groups"])
    assert np.array_equal(g["taxid"], ext[c["call"]])


def test_database_info_and_lca_on_insert(K, oracle, db, table, tax, cfg1):
    P, R, ref, seqs, reads, off = cfg1
    parents, externals, names, ranks, ids = tax
    t, ext = table
    i = db.info()
    assert i["capacity"] == 6_000_011 and i["k"] == 35 and i["l"] == 31 and i["value_bits"] == 17 and i["key_bits"] == 15
    assert i["size"] == int((t.cells & ((1 << 17) - 1) != 0).sum())
    o = oracle.k2_default_opts()
    m1, a1 = oracle.k2_scan(seqs[0][100_000:100_150], o)
    m2, a2 = oracle.k2_scan(seqs[0][500_000:500_150], o)
    assert {t.get(int(m)) for m in m1} <= {ids["Homo"], ids["Hominidae"], ids["Homo sapiens"]} and ids["Homo"] in {t.get(int(m)) for m in m1}
    assert ids["Homo sapiens"] in {t.get(int(m)) for m in m2}


def test_single_end_parity(K, oracle, db, table, cfg1):
    P, R, ref, seqs, reads, off = cfg1
    t, ext = table
    g, st = db.classify(reads, off, paired=False)
    c = t.classify(oracle.k2_default_opts(), reads, off, paired=False)
    _same(oracle, g, c, ext)
    assert st["n_probes"] == int(c["n_probes"].sum()) and st["n_kmers"] == int(c["total_kmers"].sum())
    assert st["n_classified"] == int((c["call"] != 0).sum()) > 1000 and st["n_units"] == len(c)


def test_paired_parity(K, oracle, db, table, cfg1):
    P, R, ref, seqs, reads, off = cfg1
    t, ext = table
    g, st = db.classify(reads, off, paired=True)
    c = t.classify(oracle.k2_default_opts(), reads, off, paired=True)
    assert len(g) == 10000
    _same(oracle, g, c, ext)


def test_confidence_and_hit_group_thresholds(K, oracle, db, table, cfg1):
    P, R, ref, seqs, reads, off = cfg1
    t, ext = table
    n = 4000
    for conf, mhg in ((0.3, 2), (0.9, 1), (0.0, 5)):
        go = db.opts(); go.confidence, go.min_hit_groups = conf, mhg
        oo = oracle.k2_default_opts(); oo.confidence, oo.min_hit_groups = conf, mhg
        g, _ = db.classify(reads[: n * 150], off[: n + 1], paired=True, opts=go)
        c = t.classify(oo, reads[: n * 150], off[: n + 1], paired=True)
        _same(oracle, g, c, ext)
    assert int((c["call"] != 0).sum()) > 0


def test_edge_reads(K, oracle, db, table, cfg1):
    P, R, ref, seqs, reads, off = cfg1
    t, ext = table
    recs, bases, offs = W.edge_reads(ref)
    g, st = db.classify(bases, offs, paired=False)
    c = t.classify(oracle.k2_default_opts(), bases, offs, paired=False, threads=1)
    _same(oracle, g, c, ext)
    assert g["total_kmers"][0] == 0 and g["call"][0] == 0            # empty record: unclassified, never an error
    assert g["call"][6] != 0                                          # lower-case host read


def test_misaligned_base_pointer(K, oracle, db, table, cfg1):
    P, R, ref, seqs, reads, off = cfg1
    t, ext = table
    n = 500
    pad = np.concatenate([np.frombuffer(b"GATTACA", dtype=np.uint8), reads[: n * 150]])
    g, _ = db.classify(pad, off[: n + 1] + np.uint64(7), paired=False)
    c = t.classify(oracle.k2_default_opts(), reads[: n * 150], off[: n + 1], paired=False)
    _same(oracle, g, c, ext)


def test_many_taxa_overflow_path(K, oracle, db, table, cfg1):
    """Long reads over the 64-taxon mosaic of contig 3 carry more distinct taxa than the LDS hit list holds."""
    P, R, ref, seqs, reads, off = cfg1
    t, ext = table
    recs = [bytes(seqs[3][900 + 50 * j: 900 + 50 * j + 6000]) for j in range(40)] + [bytes(seqs[0][1000 * j: 1000 * j + 3000]) for j in range(40)]
    bases = np.frombuffer(b"".join(recs), dtype=np.uint8)
    offs = np.zeros(len(recs) + 1, dtype=np.uint64); offs[1:] = np.cumsum([len(r) for r in recs])
    g, st = db.classify(bases, offs, paired=False)
    c = t.classify(oracle.k2_default_opts(), bases, offs, paired=False)
    assert st["n_overflow"] >= 30
    _same(oracle, g, c, ext)
    assert st["n_probes"] == int(c["n_probes"].sum())


def test_other_k_l(K, oracle, cfg1, tax):
    """A window that is not the k = 35 / l = 31 instantiation (run-time window length), no spaced seed."""
    P, R, ref, seqs, reads, off = cfg1
    parents, externals, names, ranks, ids = tax
    for k, l, spaced in ((25, 17, 0), (31, 31, None)):
        go = K.default_opts(); go.k, go.l = k, l
        oo = oracle.k2_default_opts(); oo.k, oo.l = k, l
        if spaced is not None:
            go.spaced_seed_mask = oo.spaced_seed_mask = spaced
        d = K.K2Db.create(go, 3_000_017, parents, externals, names, ranks)
        d.insert_sequence(seqs[0], ids["Homo sapiens"])
        cells, parent, ext = d.export()
        t = oracle.K2Table(cells, parent, 17)
        n = 3000
        g, _ = d.classify(reads[: n * 150], off[: n + 1], paired=True)
        c = t.classify(oo, reads[: n * 150], off[: n + 1], paired=True)
        _same(oracle, g, c, ext)
        assert int((c["call"] != 0).sum()) > 100
        d.close()


def test_down_sampled_database(K, oracle, cfg1, tax):
    P, R, ref, seqs, reads, off = cfg1
    parents, externals, names, ranks, ids = tax
    go = K.default_opts(); go.min_acceptable_hash = 3 << 62
    oo = oracle.k2_default_opts(); oo.min_acceptable_hash = 3 << 62
    d = K.K2Db.create(go, 1_000_003, parents, externals, names, ranks)
    kept = d.insert_sequence(seqs[0], ids["Homo sapiens"])
    full = 2 * (1_000_000 - 34) / 6                                   # ~ one minimizer change per 3 k-mers
    assert 0.1 * full < kept < 0.4 * full                             # a quarter of the hash space survives
    cells, parent, ext = d.export()
    n = 3000
    g, st = d.classify(reads[: n * 150], off[: n + 1], paired=False)
    c = oracle.K2Table(cells, parent, 17).classify(oo, reads[: n * 150], off[: n + 1], paired=False)
    _same(oracle, g, c, ext)
    assert st["n_probes"] == int(c["n_probes"].sum())
    d.close()


def test_save_open_round_trip(K, oracle, db, table, cfg1, tmp_path):
    P, R, ref, seqs, reads, off = cfg1
    db.save(tmp_path)
    assert sorted(os.listdir(tmp_path)) == ["hash.k2d", "opts.k2d", "taxo.k2d"]
    assert os.path.getsize(tmp_path / "hash.k2d") == 32 + 4 * 6_000_011 and os.path.getsize(tmp_path / "opts.k2d") == 64
    with open(tmp_path / "taxo.k2d", "rb") as f:
        assert f.read(8) == b"K2TAXDAT"
    d2 = K.K2Db.open(tmp_path)
    assert d2.info() == db.info()
    a, b = d2.export(), db.export()
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    n = 2000
    g1, _ = db.classify(reads[: n * 150], off[: n + 1], paired=True)
    g2, _ = d2.classify(reads[: n * 150], off[: n + 1], paired=True)
    assert np.array_equal(g1, g2)
    d2.close()


def test_open_errors(K, tmp_path):
    from scrubby_amd.lib import ScrubbyHipError
    with pytest.raises(ScrubbyHipError):
        K.K2Db.open(tmp_path / "nothing-here")


def _parse_report(path):
    rows = []
    for line in open(path):
        f = line.rstrip("\n").split("\t")
        assert len(f) == 6
        rows.append((float(f[0]), int(f[1]), int(f[2]), f[3], int(f[4]), f[5]))
    return rows


def test_report_writer(K, oracle, db, cfg1, tax, tmp_path):
    P, R, ref, seqs, reads, off = cfg1
    parents, externals, names, ranks, ids = tax
    g, _ = db.classify(reads, off, paired=True)
    db.write_report(g, tmp_path / "kraken.report")
    rows = _parse_report(tmp_path / "kraken.report")
    n = len(g)
    direct = np.bincount(g["call"], minlength=len(parents))
    clade = direct.copy()
    for i in range(len(parents) - 1, 1, -1):
        clade[parents[i]] += clade[i]
    assert rows[0][3] == "U" and rows[0][1] == rows[0][2] == int(direct[0]) and rows[0][4] == 0 and rows[0][5] == "unclassified"
    assert rows[1][3] == "R" and rows[1][4] == 1 and rows[1][1] == n - int(direct[0]) and rows[1][5] == "root"
    ext2int = {e: i for i, e in enumerate(externals)}
    seen = set()
    for pct, c_reads, d_reads, code, taxid, name in rows[1:]:
        i = ext2int[taxid]
        seen.add(i)
        assert c_reads == clade[i] > 0 and d_reads == direct[i] and abs(pct - 100.0 * c_reads / n) < 0.006
        assert name.strip() == names[i] and name.startswith("  " * _depth(parents, i)) and not name.startswith("  " * (_depth(parents, i) + 1))
    assert seen == {i for i in range(1, len(parents)) if clade[i] > 0}
    by_name = {r[5].strip(): r for r in rows}
    assert by_name["cellular organisms"][3] == "R1" and by_name["Eukaryota"][3] == "D" and by_name["Opisthokonta"][3] == "D1"
    assert by_name["Metazoa"][3] == "K" and by_name["Chordata"][3] == "P" and by_name["Homo sapiens"][3] == "S" and by_name["Homo"][3] == "G"
    # depth-first: a clade's rows follow it; siblings by clade count, descending
    order = [ext2int[r[4]] for r in rows[1:]]
    pos = {i: k for k, i in enumerate(order)}
    for i in order:
        kids = [j for j in order if parents[j] == i]
        assert all(pos[j] > pos[i] for j in kids)
        assert [clade[j] for j in sorted(kids, key=lambda j: pos[j])] == sorted((clade[j] for j in kids), reverse=True)
    # and the product's own reader of this format (src/classifier.rs:124-252 restated) accepts it
    from scrubby_amd import lib as S
    got = S.classifier_taxids(str(tmp_path / "kraken.report"), taxa=["Chordata"], taxa_direct=["9606"])
    assert "9606" in got and "9605" in got and "2" not in got


def _depth(parents, i):
    d = 0
    while i > 1:
        i = parents[i]; d += 1
    return d


def _write_fastq(path, ids, seqs, mate):
    op = gzip.open if str(path).endswith(".gz") else open
    with op(path, "wt") as f:
        for i, s in zip(ids, seqs):
            f.write(f"@{i}/{mate} extra\n{s.decode()}\n+\n{'I' * len(s)}\n")


def test_kraken_run_end_to_end(K, oracle, db, table, cfg1, tax, tmp_path):
    """`scrubby reads -c kraken2 -I DB -T Chordata -D 9606` on a paired FASTQ: kraken.reads / kraken.report land in the
    workdir, pairs whose call falls under the selected taxa are removed from both files."""
    import json
    P, R, ref, seqs, reads, off = cfg1
    t, ext = table
    dbdir = tmp_path / "db"; dbdir.mkdir()
    db.save(dbdir)
    n_pairs = 1500
    ids = [f"syn.{i}" for i in range(n_pairs)]
    r1 = [bytes(reads[(2 * i) * 150:(2 * i + 1) * 150]) for i in range(n_pairs)]
    r2 = [bytes(reads[(2 * i + 1) * 150:(2 * i + 2) * 150]) for i in range(n_pairs)]
    _write_fastq(tmp_path / "in_1.fastq", ids, r1, 1)
    _write_fastq(tmp_path / "in_2.fastq.gz", ids, r2, 2)
    res = K.kraken_run([tmp_path / "in_1.fastq", tmp_path / "in_2.fastq.gz"], [tmp_path / "out_1.fastq", tmp_path / "out_2.fastq.gz"], dbdir,
                       taxa=["Chordata"], taxa_direct=["9606"], workdir=tmp_path / "work", json=tmp_path / "report.json",
                       command="scrubby reads -c kraken2", read_ids=tmp_path / "ids_suffixed.tsv", classifier_args="--confidence 0.0")
    c = t.classify(oracle.k2_default_opts(), reads[: 2 * n_pairs * 150], off[: 2 * n_pairs + 1], paired=True)
    lines = open(tmp_path / "work" / "kraken.reads").read().splitlines()
    assert len(lines) == n_pairs
    for i, line in enumerate(lines):
        f = line.split("\t")
        assert len(f) == 5 and f[1] == f"syn.{i}" and f[3] == "150|150"            # mate 1's id, "/1" removed
        assert f[0] == ("C" if c["call"][i] else "U") and int(f[2]) == int(ext[c["call"][i]])
    from scrubby_amd import lib as S
    taxids = set(S.classifier_taxids(str(tmp_path / "work" / "kraken.report"), taxa=["Chordata"], taxa_direct=["9606"]))
    hit = {f"syn.{i}" for i in range(n_pairs) if str(int(ext[c["call"][i]])) in taxids}
    assert res["n_depleted_ids"] == len(hit) > 100
    # reference quirk kept: the id set holds mate 1's id without "/1", the filter compares each record's own first
    # token ("syn.N/1", "syn.N/2"), so with /1 /2 suffixed headers nothing matches and nothing is removed
    kept1 = [l[1:].split()[0] for l in open(tmp_path / "out_1.fastq") if l.startswith("@syn.")]
    assert len(kept1) == n_pairs
    rep = json.load(open(tmp_path / "report.json"))
    assert rep["reads_in"] == 2 * n_pairs and rep["settings"]["classifier"] == "kraken2" and rep["settings"]["taxa"] == ["Chordata"]
    assert rep["settings"]["classifier_args"] == "--confidence 0.0"
    # --read-ids lists the input records missing from the outputs (utils.rs:265-279): none here, although the id set is not empty
    assert open(tmp_path / "ids_suffixed.tsv").read().split() == ["id"]
    # plain ids (no /1 /2): the pairs are removed from both files
    with open(tmp_path / "p_1.fastq", "w") as f1, open(tmp_path / "p_2.fastq", "w") as f2:
        for i in range(n_pairs):
            f1.write(f"@syn.{i} 1:N:0\n{r1[i].decode()}\n+\n{'I' * 150}\n")
            f2.write(f"@syn.{i} 2:N:0\n{r2[i].decode()}\n+\n{'I' * 150}\n")
    res = K.kraken_run([tmp_path / "p_1.fastq", tmp_path / "p_2.fastq"], [tmp_path / "q_1.fastq", tmp_path / "q_2.fastq"], dbdir,
                       taxa=["Chordata"], taxa_direct=["9606"], workdir=tmp_path / "work2", json=tmp_path / "report2.json", read_ids=tmp_path / "ids.tsv")
    got = open(tmp_path / "ids.tsv").read().split()
    assert got[0] == "id" and set(got[1:]) == hit and len(got) == 1 + len(hit)
    assert json.load(open(tmp_path / "report2.json"))["settings"]["classifier_args"] is None
    # an id table with niffler's xz extension (utils.rs:28-36): the same ids (taxa_direct as above)
    import lzma
    K.kraken_run([tmp_path / "p_1.fastq", tmp_path / "p_2.fastq"], [tmp_path / "z_1.fastq.bz2", tmp_path / "z_2.fastq.xz"], dbdir,
                 taxa=["Chordata"], taxa_direct=["9606"], workdir=tmp_path / "work4", read_ids=tmp_path / "ids.tsv.xz")
    gotx = lzma.open(tmp_path / "ids.tsv.xz", "rt").read().split()
    assert gotx[0] == "id" and set(gotx[1:]) == hit
    assert open(tmp_path / "z_1.fastq.bz2", "rb").read(3) == b"BZh" and open(tmp_path / "z_2.fastq.xz", "rb").read(6) == b"\xfd7zXZ\x00"
    for name in ("q_1.fastq", "q_2.fastq"):
        kept = {l[1:].split()[0] for l in open(tmp_path / name) if l.startswith("@syn.")}
        assert kept == set(ids) - hit
    rep = json.load(open(tmp_path / "report2.json"))
    assert rep["reads_removed"] == 2 * len(hit) and rep["reads_out"] == 2 * (n_pairs - len(hit))
    # extract mode keeps exactly the selected pairs
    K.kraken_run([tmp_path / "p_1.fastq", tmp_path / "p_2.fastq"], [tmp_path / "e_1.fastq", tmp_path / "e_2.fastq"], dbdir,
                 taxa=["Chordata"], taxa_direct=["9606"], workdir=tmp_path / "work3", extract=True)
    assert {l[1:].split()[0] for l in open(tmp_path / "e_1.fastq") if l.startswith("@syn.")} == hit


def test_kraken_run_single_end_and_cli(K, oracle, db, table, cfg1, tmp_path):
    import subprocess
    P, R, ref, seqs, reads, off = cfg1
    t, ext = table
    dbdir = tmp_path / "db"; dbdir.mkdir()
    db.save(dbdir)
    n = 800
    with open(tmp_path / "in.fastq", "w") as f:
        for i in range(n):
            f.write(f"@r{i}\n{bytes(reads[i * 150:(i + 1) * 150]).decode()}\n+\n{'I' * 150}\n")
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scrubby_amd", "scrubby-hip")
    p = subprocess.run([exe, "reads", "-i", str(tmp_path / "in.fastq"), "-o", str(tmp_path / "out.fastq"), "-c", "kraken2", "-I", str(dbdir),
                        "-T", "Chordata", "-w", str(tmp_path / "w"), "-C", "--confidence 0.1 --minimum-hit-groups 3", "-j", str(tmp_path / "r.json")],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    oo = oracle.k2_default_opts(); oo.confidence, oo.min_hit_groups = 0.1, 3
    c = t.classify(oo, reads[: n * 150], off[: n + 1], paired=False)
    lines = open(tmp_path / "w" / "kraken.reads").read().splitlines()
    assert [l.split("\t")[2] for l in lines] == [str(int(ext[x])) for x in c["call"]]
    assert all(l.split("\t")[3] == "150" for l in lines)
    kept = sum(1 for l in open(tmp_path / "out.fastq") if l.startswith("@r"))
    assert 0 < kept < n
    p = subprocess.run([exe, "reads", "-i", str(tmp_path / "in.fastq"), "-o", str(tmp_path / "o2.fastq"), "-c", "kraken2", "-I", str(dbdir)], capture_output=True, text=True)
    assert p.returncode == 1 and "MissingTaxa" in p.stderr
    p = subprocess.run([exe, "reads", "-i", str(tmp_path / "in.fastq"), "-o", str(tmp_path / "o2.fastq"), "-c", "metabuli", "-I", str(dbdir), "-T", "x"],
                       capture_output=True, text=True)
    assert p.returncode == 2


def test_tiny_tables_wrap_and_ragged_tail(K, oracle, cfg1, tax):
    """Probe chains that run through the ragged last group of cells, wrap to cell 0, or lap a full table."""
    P, R, ref, seqs, reads, off = cfg1
    parents, externals, names, ranks, ids = tax
    rng = np.random.default_rng(17)
    n = 1500
    for cap, n_keys in ((1009, 900), (1024, 1000), (13, 12), (5, 5), (64, 64)):
        d = K.K2Db.create(K.default_opts(), cap, parents, externals, names, ranks)
        keys = rng.integers(0, 1 << 62, n_keys, dtype=np.uint64)
        if cap >= 1000:            # some true minimizers of the reads, so that there are real hits too
            m, a = oracle.k2_scan(bytes(reads[:150]), oracle.k2_default_opts())
            keys[:20] = m[:20]
        d.insert(keys, rng.integers(1, len(parents), n_keys, dtype=np.uint32))
        cells, parent, ext = d.export()
        g, st = d.classify(reads[: n * 150], off[: n + 1], paired=True)
        c = oracle.K2Table(cells, parent, 17).classify(oracle.k2_default_opts(), reads[: n * 150], off[: n + 1], paired=True)
        _same(oracle, g, c, ext)
        assert st["n_probes"] == int(c["n_probes"].sum())
        d.close()


def test_kraken_run_streaming_equals_collect_then_classify(K, oracle, db, cfg1, tmp_path, monkeypatch):
    """sh_kraken_run's streaming form (chunked in-place parser, parallel formatting of kraken.reads, taxid selection from
    the results in memory, parallel filter; csrc/sh_stream.cpp) against the collect-then-classify form
    (SCRUBBY_HIP_LEGACY_HOST=1): every file they write must be the same, with small chunks so that records straddle cuts."""
    import json
    import gzip
    P, R, ref, seqs, reads, off = cfg1
    dbdir = tmp_path / "db"; dbdir.mkdir()
    db.save(dbdir)
    n_pairs = 6000
    with open(tmp_path / "a_1.fastq", "w") as f1, gzip.open(tmp_path / "a_2.fastq.gz", "wt") as f2:
        for i in range(n_pairs):
            s1, s2 = bytes(reads[(2 * i) * 150:(2 * i + 1) * 150]).decode(), bytes(reads[(2 * i + 1) * 150:(2 * i + 2) * 150]).decode()
            f1.write(f"@syn.{i} 1:N:0\n{s1}\n+\n{'I' * 150}\n")
            f2.write(f"@syn.{i} 2:N:0\n{s2[:100 + i % 50]}\n+\n{'I' * (100 + i % 50)}\n")          # ragged mate 2
    monkeypatch.setenv("SCRUBBY_HIP_CHUNK_MB", "1")
    out = {}
    for name, env in (("stream", "0"), ("legacy", "1")):
        monkeypatch.setenv("SCRUBBY_HIP_LEGACY_HOST", env)
        w = tmp_path / f"w_{name}"
        res = K.kraken_run([tmp_path / "a_1.fastq", tmp_path / "a_2.fastq.gz"], [tmp_path / f"{name}_1.fastq", tmp_path / f"{name}_2.fastq.gz"], dbdir,
                           taxa=["Chordata"], taxa_direct=["9606"], workdir=w, json=tmp_path / f"{name}.json", read_ids=tmp_path / f"{name}.tsv")
        rep = json.load(open(tmp_path / f"{name}.json"))
        out[name] = (res["reads_in"], res["reads_out"], res["reads_removed"], res["n_depleted_ids"],
                     open(w / "kraken.reads").read(), open(w / "kraken.report").read(),
                     open(tmp_path / f"{name}_1.fastq").read(), gzip.open(tmp_path / f"{name}_2.fastq.gz", "rt").read(),
                     sorted(open(tmp_path / f"{name}.tsv").read().split()), {k: v for k, v in rep.items() if k not in ("date", "output")})
    assert out["stream"] == out["legacy"]
    assert out["stream"][3] > 500 and out["stream"][2] == 2 * out["stream"][3]
    # single-end, extract mode
    for name, env in (("s_stream", "0"), ("s_legacy", "1")):
        monkeypatch.setenv("SCRUBBY_HIP_LEGACY_HOST", env)
        res = K.kraken_run([tmp_path / "a_1.fastq"], [tmp_path / f"{name}.fastq"], dbdir, taxa_direct=["9606"], workdir=tmp_path / f"w_{name}", extract=True,
                           json=tmp_path / f"{name}.json")      # counts are filled with a report only in the collect-then-classify form
        out[name] = (res["reads_in"], res["reads_out"], res["reads_extracted"], open(tmp_path / f"w_{name}" / "kraken.reads").read(), open(tmp_path / f"{name}.fastq").read())
    assert out["s_stream"] == out["s_legacy"]
    # a mate file with a different record count is an error, as before
    with open(tmp_path / "short_2.fastq", "w") as f:
        f.write("@syn.0 2\nACGT\n+\nIIII\n")
    monkeypatch.setenv("SCRUBBY_HIP_LEGACY_HOST", "0")
    from scrubby_amd import lib as S
    with pytest.raises(S.ScrubbyHipError, match="fewer records than mate 1"):
        K.kraken_run([tmp_path / "a_1.fastq", tmp_path / "short_2.fastq"], [tmp_path / "x1.fastq", tmp_path / "x2.fastq"], dbdir, taxa_direct=["9606"], workdir=tmp_path / "wx")


def test_database_written_by_plain_python(K, oracle):
    """hash.k2d / opts.k2d / taxo.k2d packed byte by byte by tests/golden/make_k2_pydb.py (not by sh_k2_save), opened by sh_k2_open:
    the table, the taxonomy and every call must equal that script's longhand expectations, and the oracle's."""
    from tests.test_k2_oracle_cpu import load_pydb, pydb_units
    raw, exp = load_pydb(), pydb_units()
    d = K.K2Db.open(os.path.join(os.path.dirname(__file__), "golden", "k2_pydb"))
    i = d.info()
    assert (i["capacity"], i["size"], i["k"], i["l"], i["value_bits"], i["key_bits"], i["n_nodes"]) == (exp["capacity"], exp["size"], 35, 31, 6, 26, 13)
    cells, parent, ext = d.export()
    assert np.array_equal(cells, raw["cells"]) and list(parent) == exp["parents"] and list(ext) == exp["external"]
    t = oracle.K2Table(raw["cells"], raw["parent"], raw["value_bits"])
    for conf, mhg in ((0.0, 2), (0.5, 2), (0.0, 1), (1.0, 3)):
        units = [u for u in exp["units"] if (u["confidence"], u["min_hit_groups"]) == (conf, mhg)]
        for paired in (False, True):
            us = [u for u in units if (len(u["mates"]) == 2) == paired]
            seqs = [m.encode() for u in us for m in u["mates"]]
            bases = np.frombuffer(b"".join(seqs), dtype=np.uint8)
            off = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.uint64)
            go = d.opts(); go.confidence, go.min_hit_groups = conf, mhg
            g, _ = d.classify(bases, off, paired=paired, opts=go)
            assert [int(x) for x in g["call"]] == [u["call"] for u in us], (conf, mhg, paired)
            assert [int(x) for x in g["total_kmers"]] == [u["total_kmers"] for u in us]
            assert [int(x) for x in g["hit_groups"]] == [u["hit_groups"] for u in us]
            assert [int(x) for x in g["taxid"]] == [u["taxid"] for u in us]
            oo = oracle.k2_default_opts(); oo.value_bits, oo.confidence, oo.min_hit_groups = 6, conf, mhg
            c = t.classify(oo, bases, off, paired=paired, threads=1)
            assert np.array_equal(g["call"], c["call"]) and np.array_equal(g["hit_groups"], c["hit_groups"])
    d.close()
