"""`scrubby classifier` rows: the taxid decision rule of /root/reference/src/classifier.rs,
mirrored in C++ (csrc/sh_host.cpp) and checked against a line-by-line Python restatement of the Rust plus
hand-traced vectors for the quirks of SURVEY.md App. C (Q8-Q11).  No GPU."""
import json
import os

import pytest

from scrubby_amd import lib as S

LEVELS = ["None", "Unclassified", "NoRank", "Root", "Domain", "Kingdom", "Phylum", "Class", "Order", "Family", "Genus", "Species", "Unspecified"]


def py_tax_level(rank):                                  # get_tax_level, classifier.rs:345-373
    for prefix, lv in (("U", "Unclassified"), ("no rank", "NoRank"), ("R", "Root")):
        if rank.startswith(prefix):
            return LEVELS.index(lv)
    for letter, word, lv in (("D", "superkingdom", "Domain"), ("K", "kingdom", "Kingdom"), ("P", "phylum", "Phylum"), ("C", "class", "Class"),
                             ("O", "order", "Order"), ("F", "family", "Family"), ("G", "genus", "Genus"), ("S", "species", "Species")):
        if rank.startswith(letter) or rank.startswith(word):
            return LEVELS.index(lv)
    return LEVELS.index("Unspecified")


def py_taxids(lines, taxa, direct):                      # get_taxids_from_report, classifier.rs:124-252
    taxa = [t.strip() for t in taxa]; direct = [t.strip() for t in direct]
    out, level, parent = set(), 0, ""
    for line in lines:
        f = line.split("\t")
        reads_direct = int(f[2])
        rank, tid, name = f[3].strip(), f[4].strip(), f[5].strip()
        lv = py_tax_level(rank)
        if name in direct or tid in direct:
            out.add(tid)
        if lv < LEVELS.index("Domain"):
            continue
        if name in taxa or tid in taxa:
            level, parent = lv, name
            if reads_direct > 0:
                out.add(tid)
        else:
            if level == 0:
                continue
            if lv <= level and len(rank) == 1:
                level = 0
            elif reads_direct > 0:
                out.add(tid)
    return out


REPORT = [  # pct, clade reads, direct reads, rank, taxid, name (Kraken2 style, DFS order)
    " 10.00\t100\t100\tU\t0\tunclassified",
    " 90.00\t900\t2\tR\t1\troot",
    " 89.00\t890\t0\tR1\t131567\t  cellular organisms",
    " 60.00\t600\t1\tD\t2759\t    Eukaryota",
    " 59.00\t590\t0\tD1\t33154\t      Opisthokonta",
    " 58.00\t580\t3\tK\t33208\t        Metazoa",
    " 57.00\t570\t0\tP\t7711\t          Chordata",
    " 56.00\t560\t4\tC\t40674\t            Mammalia",
    " 55.00\t550\t0\tO\t9443\t              Primates",
    " 54.00\t540\t0\tF\t9604\t                Hominidae",
    " 53.00\t530\t5\tG\t9605\t                  Homo",
    " 52.00\t520\t520\tS\t9606\t                    Homo sapiens",
    "  1.00\t10\t10\tS1\t63221\t                      Homo sapiens neanderthalensis",
    "  1.00\t10\t10\tP\t6656\t          Arthropoda",
    " 29.00\t290\t0\tD\t2\t    Bacteria",
    " 28.00\t280\t7\tP\t1224\t      Pseudomonadota",
    " 27.00\t270\t270\tS\t562\t        Escherichia coli",
]


def write(tmp_path, name, lines):
    p = tmp_path / name
    p.write_text("\n".join(lines) + "\n")
    return str(p)


def test_tax_level_order_and_codes():
    assert [LEVELS[py_tax_level(r)] for r in ("U", "R1", "no rank", "D", "superkingdom", "K1", "phylum", "C", "O", "F2", "genus", "S1", "x")] == \
        ["Unclassified", "Root", "NoRank", "Domain", "Domain", "Kingdom", "Phylum", "Class", "Order", "Family", "Genus", "Species", "Unspecified"]


@pytest.mark.parametrize("taxa,direct", [
    (["Chordata"], ["9606"]), (["Chordata"], []), ([], ["9606"]), (["Eukaryota"], []), (["7711"], ["562"]), (["Homo sapiens"], []),
    (["Bacteria", "Chordata"], []), ([" Chordata "], [" Homo "]), (["nothing"], ["nope"]), (["root"], ["root", "unclassified"]), (["Metazoa"], ["131567"]),
])
def test_taxids_match_python_restatement(tmp_path, taxa, direct):
    rp = write(tmp_path, "kraken.report", REPORT)
    assert S.classifier_taxids(rp, taxa, direct) == py_taxids(REPORT, taxa, direct)


def test_hand_traced_quirks(tmp_path):
    rp = write(tmp_path, "kraken.report", REPORT)
    # -T Chordata -D 9606 (BASELINE config 5): the window opens at P Chordata (0 direct reads: not collected itself), collects
    # every deeper row with direct reads, S1 never closes it, and the next single-letter rank <= Phylum (P Arthropoda) closes it
    assert S.classifier_taxids(rp, ["Chordata"], ["9606"]) == {"40674", "9605", "9606", "63221"}
    assert S.classifier_taxids(rp, ["Chordata"], []) == {"40674", "9605", "9606", "63221"}
    # Q8: taxa_direct is taken regardless of rank or read count, even above Domain (root, cellular organisms)
    assert S.classifier_taxids(rp, [], ["131567", "root", "Primates"]) == {"131567", "1", "9443"}
    # Q9: `taxa` never matches rows above Domain
    assert S.classifier_taxids(rp, ["root", "unclassified"], []) == set()
    # Q10: the closing row itself is ignored even if it has direct reads; a later D row re-opens nothing
    assert "6656" not in S.classifier_taxids(rp, ["Chordata"], [])
    assert S.classifier_taxids(rp, ["Eukaryota"], []) == {"2759", "33208", "40674", "9605", "9606", "63221", "6656"}
    # Metabuli's full-word ranks have len != 1, so a window never closes (Q10)
    met = [l.replace("\tP\t", "\tphylum\t").replace("\tD\t", "\tsuperkingdom\t").replace("\tS\t", "\tspecies\t") for l in REPORT]
    mp = write(tmp_path, "metabuli.report", met)
    assert S.classifier_taxids(mp, ["Chordata"], []) == py_taxids(met, ["Chordata"], []) >= {"6656", "1224", "562"}


def test_report_errors(tmp_path):
    with pytest.raises(S.ScrubbyHipError):
        S.classifier_taxids(write(tmp_path, "bad1", ["0.1\tx\t1\tS\t5\tn"]), ["n"], [])          # KrakenReportReadFieldConversion
    with pytest.raises(S.ScrubbyHipError):
        S.classifier_taxids(write(tmp_path, "bad2", ["0.1\t1\t 1\tS\t5\tn"]), ["n"], [])         # strict u64 parse: no spaces
    with pytest.raises(S.ScrubbyHipError):
        S.classifier_taxids(str(tmp_path / "missing"), ["n"], [])
    # doc-comment example line of classifier.rs:447
    assert S.classifier_taxids(write(tmp_path, "doc", ["0.05\t100\t50\tS\t12345\ttaxon_name"]), ["taxon_name"], []) == {"12345"}


def test_classifier_run_kraken_and_metabuli(tmp_path):
    rp = write(tmp_path, "kraken.report", REPORT)
    fq1 = tmp_path / "r1.fq"; fq2 = tmp_path / "r2.fq"
    names = ["h1", "h2", "b1", "u1", "n1"]
    fq1.write_text("".join(f"@{n} 1\nACGT\n+\nIIII\n" for n in names))
    fq2.write_text("".join(f"@{n} 2\nTTTT\n+\nIIII\n" for n in names))
    kr = write(tmp_path, "kraken.reads", ["C\th1\t9606\t4|4\t9606:1", "C\th2\t 9605 \t4|4\tx", "C\tb1\t562\t4|4\tx", "U\tu1\t0\t4|4\tx",
                                         "C\tn1\t63221\t4|4\tx", "C\tread1\t12345\t100\tannotation"])
    o1, o2, js = str(tmp_path / "o1.fq"), str(tmp_path / "o2.fq.gz"), str(tmp_path / "rep.json")
    res = S.classifier_run([str(fq1), str(fq2)], [o1, o2], rp, kr, "kraken2", taxa=["Chordata"], taxa_direct=["9606"], json=js,
                           command="scrubby classifier ...")
    assert res["n_depleted_ids"] == 3 and (res["reads_in"], res["reads_out"], res["reads_removed"]) == (10, 4, 6)   # Q11: taxid compared trimmed
    rep = json.load(open(js))
    assert rep["settings"]["classifier"] == "kraken2" and rep["settings"]["aligner"] is None and rep["settings"]["taxa"] == ["Chordata"]
    assert rep["settings"]["taxa_direct"] == ["9606"] and rep["settings"]["report"] == rp and rep["settings"]["reads"] == kr
    assert open(o1).read() == "@b1 1\nACGT\n+\nIIII\n@u1 1\nACGT\n+\nIIII\n"
    # extraction + Metabuli's 7-column read file (doc example of classifier.rs:495); a missing reads file selects nothing (Q11)
    mr = write(tmp_path, "metabuli.tsv", ["1\th1\t9606\t100\t80.5\tspecies\tannotation", "0\tu1\t0\t100\t0\tno rank\tx"])
    res = S.classifier_run([str(fq1)], [str(tmp_path / "e.fq")], rp, mr, "metabuli", taxa_direct=["9606"], extract=True, json=str(tmp_path / "e.json"))
    assert res["n_depleted_ids"] == 1 and res["reads_extracted"] == 4 and open(tmp_path / "e.fq").read().startswith("@h1 1")
    res = S.classifier_run([str(fq1)], [str(tmp_path / "m.fq")], rp, str(tmp_path / "absent.reads"), "kraken2", taxa=["Chordata"])
    assert res["n_depleted_ids"] == 0 and open(tmp_path / "m.fq").read() == fq1.read_text()
    with pytest.raises(S.ScrubbyHipError):
        S.classifier_run([str(fq1)], [str(tmp_path / "z.fq")], rp, kr, "kraken2")               # neither --taxa nor --taxa-direct
