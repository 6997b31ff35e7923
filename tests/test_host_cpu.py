"""Host-side mirror of the reference path (C++ in csrc/sh_host.cpp), checked against the in-tree Rust semantics
restated in plain Python here: get_id (utils.rs:91-103), FastqCleaner::clean_reads (cleaner.rs:731-760),
ReadDifference::get_difference (utils.rs:250-285).  No GPU."""
import gzip
import os

import pytest

from scrubby_amd import lib as S


def py_get_id(header):                       # header.split_whitespace()[0]
    t = header.split()
    if not t:
        raise ValueError("NeedletailFastqHeader")
    return t[0]


def py_records(path):
    if not os.path.exists(path):
        return []
    import bz2
    import lzma
    magic = open(path, "rb").read(6)
    op = gzip.open if magic[:2] == b"\x1f\x8b" else bz2.open if magic[:3] == b"BZh" else lzma.open if magic == b"\xfd7zXZ\x00" else open
    txt = op(path, "rt").read()
    if not txt:
        return []
    lines = txt.split("\n")
    recs, i = [], 0
    while i < len(lines):
        if lines[i].startswith("@"):
            recs.append((lines[i][1:], lines[i + 1], lines[i + 3])); i += 4
        elif lines[i].startswith(">"):
            h = lines[i][1:]; i += 1; s = ""
            while i < len(lines) and not lines[i].startswith(">"):
                s += lines[i]; i += 1
            recs.append((h, s, None))
        else:
            i += 1
    return recs


def py_clean(recs, ids, extract):
    return [r for r in recs if (py_get_id(r[0]) in ids) == extract]


def py_difference(inputs, outputs):
    tin = tout = diff = 0
    for a, b in zip(inputs, outputs):
        out_ids = {py_get_id(r[0]) for r in py_records(b)}
        tout += len(py_records(b))
        for r in py_records(a):
            tin += 1
            diff += py_get_id(r[0]) not in out_ids
    return tin, tout, diff


def test_get_id_matches_reference_semantics():
    for h in ["read1 description", "@read1 description", "syn.17 1:N:0:0", "  lead  x", "a\tb c", "single", "r/1", "id\rx"]:
        assert S.get_id(h) == py_get_id(h)
    assert S.get_id("@read1 description") == "@read1"       # the doc example of utils.rs:89 keeps the '@' 
    for h in ["", "   ", "\t"]:
        with pytest.raises(S.ScrubbyHipError):
            S.get_id(h)


FQ = "@r1 1:N:0:0\nACGT\n+\nIIII\n@r2 desc here\nGGCC\n+\nFFFF\n@r3\nTTAA\n+\n####\n@r2 dup\nAAAA\n+\nIIII\n"


def test_filter_deplete_extract_and_duplicates(tmp_path):
    a = tmp_path / "in.fastq"; a.write_text(FQ)
    for extract in (False, True):
        for out_name in ("out.fastq", "out.fastq.gz"):
            o = tmp_path / f"{int(extract)}_{out_name}"
            n_in, n_out = S.filter_fastx(str(a), str(o), ["r2", "zzz"], extract)
            exp = py_clean(py_records(str(a)), {"r2", "zzz"}, extract)
            assert (n_in, n_out) == (4, len(exp))
            assert py_records(str(o)) == exp                  # one mapped occurrence removes ALL records with that id (Q2)
    # the full header (id + description) is written back unchanged
    o = tmp_path / "hdr.fastq"
    S.filter_fastx(str(a), str(o), [], False)
    assert o.read_text() == FQ


def test_filter_gz_input_and_fasta(tmp_path):
    g = tmp_path / "in.fastq.gz"
    with gzip.open(g, "wt") as f:
        f.write(FQ)
    o = tmp_path / "o.fastq"
    assert S.filter_fastx(str(g), str(o), ["r1"], False) == (4, 3)
    fa = tmp_path / "x.fa"; fa.write_text(">c1 first\nACGT\nACGT\n>c2\nGG\n")
    o2 = tmp_path / "o.fa"
    assert S.filter_fastx(str(fa), str(o2), ["c2"], False) == (2, 1)
    assert o2.read_text() == ">c1 first\nACGTACGT\n"


def test_empty_input_creates_no_output(tmp_path):
    a = tmp_path / "empty.fastq"; a.write_text("")
    o = tmp_path / "o.fastq"
    assert S.filter_fastx(str(a), str(o), ["x"], False) == (0, 0)
    assert not o.exists()
    with pytest.raises(S.ScrubbyHipError):
        S.filter_fastx(str(tmp_path / "missing.fastq"), str(o), [], False)


def test_truncated_fastq_is_an_error(tmp_path):
    a = tmp_path / "bad.fastq"; a.write_text("@r1\nACGT\n+\nIII\n")
    with pytest.raises(S.ScrubbyHipError):
        S.filter_fastx(str(a), str(tmp_path / "o.fastq"), [], False)
    with pytest.raises(S.ScrubbyHipError):
        S.filter_fastx(str(a), str(tmp_path / "o.fastq.xz"), [], False)     # (whatever the output's container)


def test_read_difference_counts_records_over_both_files(tmp_path):
    r1 = tmp_path / "r1.fq"; r2 = tmp_path / "r2.fq"
    r1.write_text("@p1 1\nA\n+\nI\n@p2 1\nC\n+\nI\n@p3/1\nG\n+\nI\n")
    r2.write_text("@p1 2\nA\n+\nI\n@p2 2\nC\n+\nI\n@p3/2\nG\n+\nI\n")
    o1 = tmp_path / "o1.fq"; o2 = tmp_path / "o2.fq"
    ids = ["p2", "p3/1"]                                      # suffixed mates are filtered independently (Q1)
    S.filter_fastx(str(r1), str(o1), ids, False); S.filter_fastx(str(r2), str(o2), ids, False)
    got = S.read_difference([str(r1), str(r2)], [str(o1), str(o2)])
    assert got == py_difference([str(r1), str(r2)], [str(o1), str(o2)]) == (6, 3, 3)   # a removed pair counts 2 (Q3)
