"""BASELINE.json configs[0] end to end on the GPU: `scrubby reads` over 10k synthetic 2x150 bp pairs vs the 5 Mb
mini reference, through sh_reads_run (C ABI) and the scrubby-hip CLI.  Expected results come from the CPU oracle's
per-record flags pushed through the reference's ID-set semantics restated in Python (tests/test_host_cpu.py)."""
import gzip
import json
import os
import subprocess

import numpy as np
import pytest

from tests import workloads as W
from tests.test_host_cpu import py_records, py_clean, py_difference

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dataset(oracle, tmp_path_factory):
    d = tmp_path_factory.mktemp("cfg0")
    n_pairs = 10000
    P, R, ref, seqs, reads, off = W.cfg1(oracle, 2 * n_pairs)
    fa = d / "ref.fa.gz"
    with gzip.open(fa, "wt") as f:
        for i, s in enumerate(seqs):
            f.write(f">ctg{i} synthetic\n")
            s = bytes(s).decode()
            for j in range(0, len(s), 60000):
                f.write(s[j:j + 60000] + "\n")
    reads = reads.reshape(-1, 150)
    r1, r2 = d / "R1.fastq", d / "R2.fastq.gz"
    with open(r1, "w") as f1, gzip.open(r2, "wt") as f2:
        for p in range(n_pairs):
            f1.write(f"@syn.{p} 1:N:0:0\n{bytes(reads[2 * p]).decode()}\n+\n{'I' * 150}\n")
            f2.write(f"@syn.{p} 2:N:0:0\n{bytes(reads[2 * p + 1]).decode()}\n+\n{'I' * 150}\n")
    idx = oracle.Index.build(seqs, 11, 21)
    flags, _ = idx.classify(oracle.preset("sr"), reads.reshape(-1), off, threads=8, want_trace=False)
    ids = {f"syn.{p}" for p in range(n_pairs) if flags[2 * p] == 1 or flags[2 * p + 1] == 1}    # HashSet union (cleaner.rs:564-570)
    return d, str(fa), str(r1), str(r2), ids, n_pairs


def check_outputs(r1, r2, o1, o2, ids, extract):
    for i, o in ((r1, o1), (r2, o2)):
        assert py_records(o) == py_clean(py_records(i), ids, extract)


def test_reads_run_deplete_with_report(dataset):
    from scrubby_amd import lib as S
    d, fa, r1, r2, ids, n_pairs = dataset
    o1, o2, js, tsv = str(d / "c1.fastq"), str(d / "c2.fastq.gz"), str(d / "report.json"), str(d / "ids.tsv")
    res = S.reads_run([r1, r2], [o1, o2], fa, json=js, read_ids=tsv, command="scrubby reads -i R1 R2 -o c1 c2 -I ref.fa.gz")
    check_outputs(r1, r2, o1, o2, ids, False)
    exp = py_difference([r1, r2], [o1, o2])
    assert (res["reads_in"], res["reads_out"], res["reads_removed"], res["reads_extracted"]) == (exp[0], exp[1], exp[2], 0)
    assert res["reads_in"] == 2 * n_pairs and res["reads_removed"] == 2 * len(ids) and res["n_depleted_ids"] == len(ids)
    rep = json.load(open(js))
    assert list(rep.keys()) == ["version", "date", "command", "input", "output", "reads_in", "reads_out", "reads_removed",
                                "reads_extracted", "settings"]                       # report.rs:10-22, struct order
    assert list(rep["settings"].keys()) == ["aligner", "classifier", "index", "alignment", "reads", "report", "taxa", "taxa_direct",
                                            "classifier_args", "aligner_args", "preset", "min_len", "min_cov", "min_mapq", "extract"]
    assert rep["settings"]["aligner"] == "minimap2-rs" and rep["settings"]["preset"] == "Sr" and rep["settings"]["extract"] is False
    assert rep["reads_removed"] == 2 * len(ids) and rep["input"] == [r1, r2] and rep["date"].endswith("Z")
    got_ids = open(tsv).read().split("\n")
    assert got_ids[0] == "id" and set(x for x in got_ids[1:] if x) == ids


def test_reads_run_with_bzip2_and_xz_files(dataset, tmp_path):
    """niffler's other containers end to end (utils.rs:28-36, 56-74, 377-383): R1 as bzip2, R2 as xz, outputs .bz2 / .xz, the id table
    as .tsv.xz - same records, counts and ids as the plain / gzip run."""
    import bz2
    import lzma
    from scrubby_amd import lib as S
    d, fa, r1, r2, ids, n_pairs = dataset
    b1, x2 = str(tmp_path / "R1.fastq.bz2"), str(tmp_path / "R2.fastq.xz")
    open(b1, "wb").write(bz2.compress(open(r1, "rb").read()))
    open(x2, "wb").write(lzma.compress(gzip.open(r2, "rb").read()))
    o1, o2, js, tsv = str(tmp_path / "c1.fastq.xz"), str(tmp_path / "c2.fastq.bz2"), str(tmp_path / "report.json"), str(tmp_path / "ids.tsv.xz")
    res = S.reads_run([b1, x2], [o1, o2], fa, json=js, read_ids=tsv, command="scrubby reads")
    check_outputs(r1, r2, o1, o2, ids, False)
    assert open(o1, "rb").read(6) == b"\xfd7zXZ\x00" and open(o2, "rb").read(3) == b"BZh"
    assert res["reads_in"] == 2 * n_pairs and res["reads_removed"] == 2 * len(ids) and res["n_depleted_ids"] == len(ids)
    got_ids = lzma.open(tsv, "rt").read().split("\n")
    assert got_ids[0] == "id" and set(x for x in got_ids[1:] if x) == ids
    assert json.load(open(js))["reads_removed"] == 2 * len(ids)


def test_reads_run_extract_and_single_end(dataset):
    from scrubby_amd import lib as S
    d, fa, r1, r2, ids, n_pairs = dataset
    o1, o2, js = str(d / "e1.fastq.gz"), str(d / "e2.fastq"), str(d / "e.json")
    res = S.reads_run([r1, r2], [o1, o2], fa, preset="sr", extract=True, json=js)
    check_outputs(r1, r2, o1, o2, ids, True)
    assert res["reads_extracted"] == 2 * (n_pairs - len(ids)) and res["reads_removed"] == 0     # `difference` = records not in the output
    # one input file: the reference defaults to map-ont (scrubby.rs:941-942); sr given explicitly keeps the decision per record
    s1 = str(d / "s1.fastq")
    res = S.reads_run([r1], [s1], fa, preset="sr", json=str(d / "s.json"))
    assert json.load(open(d / "s.json"))["reads_in"] == n_pairs
    assert res["reads_out"] == len(py_records(s1))


def test_cli_matches_library(dataset):
    d, fa, r1, r2, ids, n_pairs = dataset
    exe = os.path.join(ROOT, "scrubby_amd", "scrubby-hip")
    o1, o2, js = str(d / "k1.fastq"), str(d / "k2.fastq"), str(d / "k.json")
    cp = subprocess.run([exe, "reads", "-i", r1, r2, "-o", o1, o2, "-I", fa, "-j", js, "-t", "8"], capture_output=True, text=True)
    assert cp.returncode == 0, cp.stderr
    check_outputs(r1, r2, o1, o2, ids, False)
    rep = json.load(open(js))
    assert rep["reads_removed"] == 2 * len(ids) and rep["command"].startswith(exe + " reads -i")
    assert subprocess.run([exe, "reads", "-i", r1, "-o", o1, "-I", fa, "-p", "lr"], capture_output=True).returncode == 1   # Preset::Lr rejected
    assert subprocess.run([exe, "reads", "-i", r1, "-o", o1, "-I", fa, "-a", "bowtie2"], capture_output=True).returncode == 2


def test_index_cache_in_reads_run(dataset):
    from scrubby_amd import lib as S
    d, fa, r1, r2, ids, n_pairs = dataset
    cache = str(d / "ref.shidx")
    S.Index.build_fasta(fa, S.preset("sr")).save(cache)
    o1, o2 = str(d / "x1.fastq"), str(d / "x2.fastq")
    res = S.reads_run([r1, r2], [o1, o2], cache, json=str(d / "x.json"))
    assert res["reads_removed"] == 2 * len(ids)


def test_a_minimap2_index_file_in_reads_run(dataset, oracle):
    """`-I ref.mmi` (cleaner.rs:475-479 hands minimap2 the path as it is): same id set and outputs as with the FASTA (that the file's k
    and w prevail over the preset's is tests/test_index_mmi.py's)."""
    from scrubby_amd import lib as S
    from tests.test_index_mmi import write_mmi
    d, fa, r1, r2, ids, n_pairs = dataset
    P, R, ref, seqs, reads, off = W.cfg1(oracle, 2 * n_pairs)
    mmi = str(d / "ref.mmi")
    write_mmi(mmi, seqs, [f"ctg{i}" for i in range(len(seqs))], k=21, w=11, rng=np.random.default_rng(3))
    o1, o2 = str(d / "m1.fastq"), str(d / "m2.fastq")
    res = S.reads_run([r1, r2], [o1, o2], mmi, preset="sr", json=str(d / "m.json"))
    assert res["reads_removed"] == 2 * len(ids)
    check_outputs(r1, r2, o1, o2, ids, False)


def test_streaming_pipeline_small_chunks_and_second_pass(dataset, monkeypatch):
    """sh_reads_run's streaming host path (csrc/sh_stream.cpp): many chunks per file, with the parsed chunks retained
    in memory and with the files streamed a second time, against the collect-then-map path and the oracle's id set."""
    from scrubby_amd import lib as S
    d, fa, r1, r2, ids, n_pairs = dataset
    monkeypatch.setenv("SCRUBBY_HIP_CHUNK_MB", "1")            # R1 is ~3.3 MB: several chunks, records straddle the cuts
    outs = {}
    # "retained": the device thread classifies whatever chunks have piled up in one call (concatenated in HBM); "per_chunk": one call each
    for name, env in (("retained", {}), ("streamed", {"SCRUBBY_HIP_RETAIN_MB": "0"}), ("per_chunk", {"SCRUBBY_HIP_NO_COALESCE": "1"}),
                      ("legacy", {"SCRUBBY_HIP_LEGACY_HOST": "1"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        o1, o2, js, tsv = (str(d / f"{name}_{x}") for x in ("1.fastq", "2.fastq.gz", "r.json", "ids.tsv.gz"))
        res = S.reads_run([r1, r2], [o1, o2], fa, json=js, read_ids=tsv, threads=6)
        for k in env:
            monkeypatch.delenv(k)
        check_outputs(r1, r2, o1, o2, ids, False)
        got = gzip.open(tsv, "rt").read().split("\n")
        assert got[0] == "id" and set(x for x in got[1:] if x) == ids
        outs[name] = (res["reads_in"], res["reads_out"], res["reads_removed"], res["reads_extracted"], res["n_depleted_ids"],
                      open(o1, "rb").read(), gzip.open(o2, "rb").read())
    assert outs["retained"] == outs["streamed"] == outs["per_chunk"] == outs["legacy"]
    assert outs["retained"][:3] == (2 * n_pairs, 2 * (n_pairs - len(ids)), 2 * len(ids))
    # extract mode with the id table: the ids NOT written are the complement
    o1, o2, tsv = str(d / "sx1.fastq"), str(d / "sx2.fastq"), str(d / "sx.tsv")
    res = S.reads_run([r1, r2], [o1, o2], fa, extract=True, read_ids=tsv, threads=3)
    check_outputs(r1, r2, o1, o2, ids, True)
    got = open(tsv).read().split("\n")
    assert set(x for x in got[1:] if x) == {f"syn.{p}" for p in range(n_pairs)} - ids
    assert res["reads_extracted"] == 2 * (n_pairs - len(ids))


def test_streaming_pipeline_empty_read_aborts(dataset, tmp_path):
    from scrubby_amd import lib as S
    d, fa, r1, r2, ids, n_pairs = dataset
    bad = tmp_path / "empty_read.fastq"
    bad.write_text("@a\nACGTACGTACGTACGTACGTACGTACGTACGT\n+\n" + "I" * 32 + "\n@b\n\n+\n\n")
    with pytest.raises(S.ScrubbyHipError) as e:
        S.reads_run([str(bad)], [str(tmp_path / "o.fastq")], fa, preset="sr")
    assert "Sequence is empty" in str(e.value)                 # minimap2-rs' Err aborts the run (cleaner.rs:552,566)


def test_streaming_pipeline_falls_back_when_a_cut_is_wrong(dataset, tmp_path, monkeypatch):
    """Sequence lines starting with '@' above quality lines starting with '+' fool the backward boundary scan; the chunk
    before the wrong cut fails to parse and pass 1 reruns with the sequential reader: same outputs as the legacy path."""
    from scrubby_amd import lib as S
    d, fa, r1, r2, ids, n_pairs = dataset
    odd = tmp_path / "odd.fastq"
    odd.write_text("".join(f"@r{i} x\n@CGTACGTACGTACGTACGTACGTACGTACG\n+\n+{'I' * 30}\n" for i in range(30000)))
    monkeypatch.setenv("SCRUBBY_HIP_CHUNK_MB", "1")
    o_new, o_old = tmp_path / "new.fastq", tmp_path / "old.fastq"
    a = S.reads_run([str(odd)], [str(o_new)], fa, preset="sr", json=str(tmp_path / "a.json"))
    monkeypatch.setenv("SCRUBBY_HIP_LEGACY_HOST", "1")
    b = S.reads_run([str(odd)], [str(o_old)], fa, preset="sr", json=str(tmp_path / "b.json"))
    assert o_new.read_bytes() == o_old.read_bytes()
    assert (a["reads_in"], a["reads_out"], a["n_depleted_ids"]) == (b["reads_in"], b["reads_out"], b["n_depleted_ids"]) == (30000, 30000, 0)
