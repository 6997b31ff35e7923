"""`scrubby alignment`: PAF/GAF/TXT -> id set with the reference's filter
(qalen >= min_len OR qcov >= min_cov) AND mapq >= min_mapq  (/root/reference/src/alignment.rs:84-114,242-276).  No GPU."""
import gzip
import json

import pytest

from scrubby_amd import lib as S

PAF = [  # qname qlen qstart qend strand tname tlen tstart tend mlen blen mapq
    "r1\t150\t0\t150\t+\tchr1\t1000\t10\t160\t150\t150\t60",
    "r2\t150\t100\t150\t-\tchr1\t1000\t10\t60\t50\t50\t60",     # qalen 50, qcov 0.333
    "r3\t150\t0\t150\t+\tchr2\t1000\t10\t160\t140\t150\t3",     # low mapq
    "r4\t0\t0\t0\t+\tchr2\t1000\t0\t0\t0\t0\t60",               # qlen 0 -> coverage 0
    "r2\t150\t0\t120\t+\tchr3\t1000\t0\t120\t120\t120\t20",     # second record of r2
]


def py_select(min_len, min_cov, min_mapq):
    out = set()
    for l in PAF:
        f = l.split("\t")
        qlen, qs, qe, mapq = int(f[1]), int(f[2]), int(f[3]), int(f[11])
        qalen = qe - qs
        qcov = 0.0 if qlen == 0 else qalen / qlen
        if (qalen >= min_len or qcov >= min_cov) and mapq >= min_mapq:
            out.add(f[0])
    return out


@pytest.mark.parametrize("min_len,min_cov,min_mapq", [(0, 0.0, 0), (100, 2.0, 0), (1000, 0.5, 0), (100, 2.0, 30), (1000, 0.3, 50), (60, 0.9, 10), (10 ** 6, 1.5, 0)])
def test_paf_filter_matches_reference_rule(tmp_path, min_len, min_cov, min_mapq):
    paf = tmp_path / "aln.paf"; paf.write_text("\n".join(PAF) + "\n")
    fq = tmp_path / "r.fq"; fq.write_text("".join(f"@r{i} x\nACGT\n+\nIIII\n" for i in range(1, 6)))
    out, js = tmp_path / "o.fq", tmp_path / "rep.json"
    res = S.alignment_run([str(fq)], [str(out)], str(paf), min_len=min_len, min_cov=min_cov, min_mapq=min_mapq, json=str(js))
    exp = py_select(min_len, min_cov, min_mapq)
    assert res["n_depleted_ids"] == len(exp)
    kept = [l[1:].split()[0] for l in out.read_text().split("\n") if l.startswith("@")] if out.exists() else []
    assert kept == [f"r{i}" for i in range(1, 6) if f"r{i}" not in exp]
    st = json.load(open(js))["settings"]
    assert st["alignment"] == str(paf) and st["min_len"] == min_len and st["min_cov"] == min_cov and st["min_mapq"] == min_mapq and st["aligner"] is None


def test_formats_and_errors(tmp_path):
    fq = tmp_path / "r.fq"; fq.write_text("".join(f"@r{i}\nACGT\n+\nIIII\n" for i in range(1, 4)))
    txt = tmp_path / "ids.txt"; txt.write_text("r2\nr9\n")
    assert S.alignment_run([str(fq)], [str(tmp_path / "a.fq")], str(txt), extract=True)["n_depleted_ids"] == 2
    assert (tmp_path / "a.fq").read_text() == "@r2\nACGT\n+\nIIII\n"
    gaf = tmp_path / "g.gaf"; gaf.write_text(PAF[0] + "\n")
    assert S.alignment_run([str(fq)], [str(tmp_path / "b.fq")], str(gaf))["n_depleted_ids"] == 1
    pgz = tmp_path / "x.paf.gz"
    with gzip.open(pgz, "wt") as f:
        f.write(PAF[0] + "\n")
    with pytest.raises(S.ScrubbyHipError):                       # Path::extension() is "gz": not recognised without --format
        S.alignment_run([str(fq)], [str(tmp_path / "c.fq")], str(pgz))
    assert S.alignment_run([str(fq)], [str(tmp_path / "c.fq")], str(pgz), fmt="paf")["n_depleted_ids"] == 1
    with pytest.raises(S.ScrubbyHipError):
        S.alignment_run([str(fq)], [str(tmp_path / "d.fq")], str(gaf), fmt="bam")
    empty = tmp_path / "e.paf"; empty.write_text("")
    assert S.alignment_run([str(fq)], [str(tmp_path / "e.fq")], str(empty))["n_depleted_ids"] == 0
    bad = tmp_path / "bad.paf"; bad.write_text("r1\t150\tx\t150\t+\tc\t1\t0\t1\t1\t1\t60\n")
    with pytest.raises(S.ScrubbyHipError):
        S.alignment_run([str(fq)], [str(tmp_path / "f.fq")], str(bad))
