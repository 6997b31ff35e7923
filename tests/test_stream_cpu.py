"""The streaming host pipeline's parser and filter (scrubby_amd/csrc/sh_stream.cpp, pass 2 of sh_reads_run) against the
line-by-line filter (sh_host_filter_fastx, the restated FastqCleaner::clean_reads, /root/reference/src/cleaner.rs:731-760)
and the Python restatement in tests/test_host_cpu.py: same bytes out for plain outputs, same decompressed content for
gzip, same counts - for every chunk size (records straddling chunk ends), thread count, and with or without retention."""
import gzip
import os
import random

import pytest

pytestmark = pytest.mark.timeout(300)          # a deadlock in the pipeline fails instead of hanging the suite

from scrubby_amd import lib as S
from tests.test_host_cpu import py_records, py_clean


def content(path):
    with open(path, "rb") as f:
        b = f.read()
    return gzip.decompress(b) if b[:2] == b"\x1f\x8b" else b


def make_fastq(n, rng, crlf=False, plus_hdr=False, blank=False, last_nl=True):
    nl = "\r\n" if crlf else "\n"
    out, ids = [], []
    for i in range(n):
        L = rng.choice([1, 7, 150, 151, 300])
        seq = "".join(rng.choice("ACGTN") for _ in range(L))
        q = "".join(chr(33 + rng.randrange(41)) for _ in range(L))       # '@' (64) and '+' (43) occur as first quality chars
        if i % 5 == 0:
            q = "@" + q[1:]
        hid = f"r{i}"
        hdr = hid + rng.choice(["", " 1:N:0:0", "\tx y", "/1"])
        ids.append(hdr.split()[0])
        out.append(f"@{hdr}{nl}{seq}{nl}+{hdr if plus_hdr and i % 3 == 0 else ''}{nl}{q}{nl}")
        if blank and i % 7 == 0:
            out.append(nl)
    s = "".join(out)
    if not last_nl:
        s = s.rstrip("\r\n")
    return s, ids


def make_fasta(n, rng, width):
    out, ids = [], []
    for i in range(n):
        L = rng.choice([0, 1, 59, 60, 61, 500])
        seq = "".join(rng.choice("ACGT") for _ in range(L))
        ids.append(f"c{i}")
        out.append(f">c{i} desc {i}\n")
        if width:
            out.extend(seq[j:j + width] + "\n" for j in range(0, L, width))
        else:
            out.append(seq + "\n")
    return "".join(out), ids


CASES = {
    "plain": dict(),
    "crlf": dict(crlf=True),
    "plus_hdr": dict(plus_hdr=True),
    "blank_lines": dict(blank=True),
    "no_final_newline": dict(last_nl=False),
}


@pytest.mark.parametrize("case", list(CASES))
@pytest.mark.parametrize("chunk", [64, 1000, 1 << 20])
def test_stream_filter_matches_line_filter_fastq(tmp_path, case, chunk):
    rng = random.Random(list(CASES).index(case) * 131 + chunk)
    text, ids = make_fastq(400, rng, **CASES[case])
    src = tmp_path / "in.fastq"
    src.write_bytes(text.encode())
    drop = [i for i in ids if rng.random() < 0.4] + ["not-there"]
    for extract in (False, True):
        ref = tmp_path / "ref.fastq"
        cnt = S.filter_fastx(str(src), str(ref), drop, extract)
        for threads, retain in ((1, 1), (3, 0), (4, 1), (2, 2), (2, 3)):
            out = tmp_path / f"o_{threads}_{int(retain)}.fastq"
            assert S.filter_fastx_stream(str(src), str(out), drop, extract, chunk_bytes=chunk, threads=threads, retain=retain) == cnt
            assert out.read_bytes() == ref.read_bytes()
        assert py_records(str(ref)) == py_clean(py_records(str(src)), set(drop), extract)


@pytest.mark.parametrize("width", [0, 60, 7])
def test_stream_filter_fasta_multiline(tmp_path, width):
    rng = random.Random(width)
    text, ids = make_fasta(120, rng, width)
    src = tmp_path / "in.fa"
    src.write_bytes(text.encode())
    drop = ids[::3]
    ref = tmp_path / "ref.fa"
    cnt = S.filter_fastx(str(src), str(ref), drop, False)
    for chunk in (64, 333, 1 << 20):
        out = tmp_path / f"o{chunk}.fa"
        for retain in (1, 2, 3):
            assert S.filter_fastx_stream(str(src), str(out), drop, False, chunk_bytes=chunk, threads=2, retain=retain) == cnt
            assert out.read_bytes() == ref.read_bytes()


def test_stream_filter_gzip_in_and_out(tmp_path):
    rng = random.Random(5)
    text, ids = make_fastq(3000, rng)
    src = tmp_path / "in.fastq.gz"
    with gzip.open(src, "wb") as f:
        f.write(text.encode())
    drop = ids[::2]
    ref, out = tmp_path / "ref.fastq.gz", tmp_path / "out.fastq.gz"
    cnt = S.filter_fastx(str(src), str(ref), drop, False)
    assert S.filter_fastx_stream(str(src), str(out), drop, False, chunk_bytes=20000, threads=4, retain=False) == cnt
    assert out.read_bytes()[:2] == b"\x1f\x8b"
    assert content(out) == content(ref)                       # multi-member gzip, same decompressed bytes
    assert len(py_records(str(out))) == cnt[1]                # and the Python reader (gzip module) sees every member
    # nothing kept: still a valid, empty gzip stream
    none = tmp_path / "none.fastq.gz"
    assert S.filter_fastx_stream(str(src), str(none), ids, False, chunk_bytes=20000)[1] == 0
    assert content(none) == b""


def test_stream_filter_errors(tmp_path):
    bad = tmp_path / "bad.fastq"
    bad.write_text("@r1\nACGT\n+\nIIII\n@r2\nACGT\n+\nIII\n")        # quality shorter than sequence
    with pytest.raises(S.ScrubbyHipError):
        S.filter_fastx_stream(str(bad), str(tmp_path / "o.fastq"), [], False)
    trunc = tmp_path / "trunc.fastq"
    trunc.write_text("@r1\nACGT\n+\nIIII\n@r2\nACGT\n")
    with pytest.raises(S.ScrubbyHipError):
        S.filter_fastx_stream(str(trunc), str(tmp_path / "o.fastq"), [], False, chunk_bytes=64)
    junk = tmp_path / "junk.txt"
    junk.write_text("hello\nworld\n")
    with pytest.raises(S.ScrubbyHipError):
        S.filter_fastx_stream(str(junk), str(tmp_path / "o.fastq"), [], False)
    with pytest.raises(S.ScrubbyHipError):
        S.filter_fastx_stream(str(tmp_path / "missing.fastq"), str(tmp_path / "o.fastq"), [], False, retain=False)


def test_stream_filter_long_record_grows_chunk(tmp_path):
    # one record much longer than the chunk target: the reader grows the chunk instead of splitting the record
    seq = "ACGT" * 50000
    src = tmp_path / "long.fastq"
    src.write_text(f"@a\nAC\n+\nII\n@long x\n{seq}\n+\n{'I' * len(seq)}\n@b\nGG\n+\nII\n")
    out = tmp_path / "o.fastq"
    assert S.filter_fastx_stream(str(src), str(out), ["a"], False, chunk_bytes=128, threads=2) == (3, 2)
    assert out.read_text() == f"@long x\n{seq}\n+\n{'I' * len(seq)}\n@b\nGG\n+\nII\n"


def test_boundary_guess_is_verified(tmp_path):
    """A sequence line that starts with '@' two lines above a quality line that starts with '+' looks like a record start to
    the backward scan (find_split).  The chunk before such a cut cannot end on a complete record, so its parse fails
    and the caller falls back to the sequential reader - the guess never changes a result."""
    recs = []
    for i in range(40):
        recs.append(f"@r{i}\n@CGTACGTAC\n+\n+IIIIIIIII\n")      # seq starts with '@', quality with '+'
    src = tmp_path / "odd.fastq"
    src.write_text("".join(recs))
    ref = tmp_path / "ref.fastq"
    cnt = S.filter_fastx(str(src), str(ref), ["r3"], False)
    assert cnt == (40, 39)
    out = tmp_path / "o.fastq"
    assert S.filter_fastx_stream(str(src), str(out), ["r3"], False, chunk_bytes=100, threads=2, retain=1) == cnt     # sequential: exact
    assert out.read_bytes() == ref.read_bytes()
    for mode in (2, 3):          # chunk cuts of one reader; byte ranges of several
        with pytest.raises(S.ScrubbyHipError, match="boundary guess failed"):
            S.filter_fastx_stream(str(src), str(out), ["r3"], False, chunk_bytes=100, threads=2, retain=mode)
    # quality lines that start with '@' (common in real data) do not fool the scan
    ok = tmp_path / "q_at.fastq"
    ok.write_text("".join(f"@r{i}\nACGTACGTAC\n+\n@IIIIIIIII\n" for i in range(40)))
    cnt = S.filter_fastx(str(ok), str(ref), ["r3"], False)
    for mode in (2, 3):
        assert S.filter_fastx_stream(str(ok), str(out), ["r3"], False, chunk_bytes=100, threads=2, retain=mode) == cnt
        assert out.read_bytes() == ref.read_bytes()


def test_truncated_gzip_is_an_error_not_a_short_file(tmp_path):
    """A .gz cut off in the middle of its deflate stream must not pass as a clean end of file (the reference's reader errors):
    the chunked reader (sh_stream.cpp ChunkReader) and the line reader (sh_host.cpp FastxReader) both report it."""
    import gzip
    import random
    rng = random.Random(5)
    recs = "".join(f"@r{i}\n{''.join(rng.choice('ACGT') for _ in range(100))}\n+\n{'I' * 100}\n" for i in range(3000))
    whole = tmp_path / "whole.fastq.gz"
    with gzip.open(whole, "wt") as f:
        f.write(recs)
    raw = whole.read_bytes()
    assert S.filter_fastx_stream(str(whole), str(tmp_path / "o.fastq"), [], False, chunk_bytes=50000, threads=2) == (3000, 3000)
    for cut in (len(raw) // 2, len(raw) - 9):         # mid-stream, and with only the CRC/length trailer missing
        part = tmp_path / f"cut{cut}.fastq.gz"
        part.write_bytes(raw[:cut])
        for retain in (0, 1):
            with pytest.raises(S.ScrubbyHipError, match="read error|truncated|unexpected end"):
                S.filter_fastx_stream(str(part), str(tmp_path / "p.fastq"), [], False, chunk_bytes=50000, threads=2, retain=retain)
        with pytest.raises(S.ScrubbyHipError, match="read error|truncated|unexpected end"):
            S.filter_fastx(str(part), str(tmp_path / "q.fastq"), [], False)


def test_bzip2_and_xz_in_and_out_like_niffler(tmp_path):
    """The reference reads its inputs through needletail / niffler, which sniff gzip, bzip2 and xz by their magic bytes
    (utils.rs:377-383), and picks an output's container by its extension (.gz; .bz / .bz2; .lzma / .xz -> xz: utils.rs:28-36, 56-74).
    Both readers and both writers of this backend (line filter sh_host.cpp, chunked filter sh_stream.cpp) do the same through
    sh_codec.h: every input container x every output container gives the records the plain run gives, the outputs open with
    Python's own bz2 / lzma / gzip modules, and a truncated bzip2 / xz stream is an error, not a short input."""
    import bz2
    import lzma
    rng = random.Random(11)
    text, ids = make_fastq(2500, rng)
    raw = text.encode()
    drop = ids[::3]
    plain = tmp_path / "in.fastq"; plain.write_bytes(raw)
    ref = tmp_path / "ref.fastq"
    cnt = S.filter_fastx(str(plain), str(ref), drop, False)
    want = ref.read_bytes()
    opener = {"fastq": open, "fastq.gz": gzip.open, "fastq.bz2": bz2.open, "fastq.bz": bz2.open, "fastq.xz": lzma.open, "fastq.lzma": lzma.open}
    inputs = {"in.fastq": raw, "in.fastq.gz": gzip.compress(raw), "in.fastq.bz2": bz2.compress(raw), "in.fastq.xz": lzma.compress(raw),
              "two_streams.fastq.bz2": bz2.compress(raw[:len(raw) // 2 + 7]) + bz2.compress(raw[len(raw) // 2 + 7:])}      # bzip2 -c a b > c: concatenated streams
    for name, data in inputs.items():
        src = tmp_path / name
        src.write_bytes(data)
        for ext in opener:
            for which, fn in (("line", lambda a, b: S.filter_fastx(a, b, drop, False)),
                              ("stream", lambda a, b: S.filter_fastx_stream(a, b, drop, False, chunk_bytes=30000, threads=3, retain=False)),
                              ("stream-retain", lambda a, b: S.filter_fastx_stream(a, b, drop, False, chunk_bytes=30000, threads=3, retain=True))):
                out = tmp_path / f"out_{which}.{ext}"
                assert fn(str(src), str(out)) == cnt, (name, ext, which)
                with opener[ext](out, "rb") as f:
                    assert f.read() == want, (name, ext, which)
                out.unlink()
    # the container is what the magic bytes say, whatever the name says
    odd = tmp_path / "named_plain.fastq"
    odd.write_bytes(bz2.compress(raw))
    assert S.filter_fastx_stream(str(odd), str(tmp_path / "o.fastq"), drop, False, chunk_bytes=30000) == cnt
    assert (tmp_path / "o.fastq").read_bytes() == want
    # truncated streams
    for name, data in (("cut.fastq.bz2", bz2.compress(raw)), ("cut.fastq.xz", lzma.compress(raw))):
        part = tmp_path / name
        part.write_bytes(data[:len(data) // 2])
        with pytest.raises(S.ScrubbyHipError, match="read error|truncated|corrupt"):
            S.filter_fastx(str(part), str(tmp_path / "p.fastq"), [], False)
        with pytest.raises(S.ScrubbyHipError, match="read error|truncated|corrupt"):
            S.filter_fastx_stream(str(part), str(tmp_path / "p.fastq"), [], False, chunk_bytes=30000, threads=2)
    # nothing kept: still a valid, empty stream of the container asked for
    for ext in ("fastq.bz2", "fastq.xz"):
        none = tmp_path / ("none." + ext)
        assert S.filter_fastx_stream(str(plain), str(none), ids, False, chunk_bytes=30000)[1] == 0
        with opener[ext](none, "rb") as f:
            assert f.read() == b""
    # the legacy LZMA_alone stream is not in niffler's set either: refused by name
    alone = tmp_path / "a.fastq.lzma"
    alone.write_bytes(lzma.compress(raw, format=lzma.FORMAT_ALONE))
    with pytest.raises(S.ScrubbyHipError, match="lzma-alone"):
        S.filter_fastx_stream(str(alone), str(tmp_path / "o2.fastq"), [], False, chunk_bytes=1000)
