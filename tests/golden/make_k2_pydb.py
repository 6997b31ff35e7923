#!/usr/bin/env python3
"""A small Kraken 2 database written BYTE BY BYTE in plain Python, plus what its classification must give.

Nothing of this repository's code takes part (not sh_k2_save, not oracle/k2_oracle.c): the three files are packed with
`struct` from the published layouts (SURVEY.md Appendix B "DB files"):
  opts.k2d   struct IndexOptions, 64 B: size_t k, l; uint64 spaced_seed_mask, toggle_mask; bool dna_db (+7 pad);
             uint64 minimum_acceptable_hash_value; int revcom_version, db_version, db_type (+4 pad)
  taxo.k2d   "K2TAXDAT"; uint64 node_count, name_data_len, rank_data_len; node_count x TaxonomyNode (7 x uint64: parent_id,
             first_child, child_count, name_offset, rank_offset, external_id, godparent_id); name pool; rank pool (NUL-terminated
             strings); internal ids are breadth-first, node 0 is the "no taxon" node, node 1 the root
  hash.k2d   uint64 capacity, size, key_bits, value_bits; capacity x uint32 cells = (fmix64(key) >> (32 + value_bits)) << value_bits | taxid,
             placed at fmix64(key) % capacity with linear probing, LCA on a key that is already there
and the expected results come from a longhand classifier (one k-mer at a time, dictionary of hit counts, ResolveTree as
Appendix B states it).  Tests: tests/test_k2_oracle_cpu.py (the CPU oracle reads these files and must agree) and
tests/test_k2_gpu.py (sh_k2_open reads them on the GPU box and must agree).

Run: python tests/golden/make_k2_pydb.py     (rewrites tests/golden/k2_pydb/{opts,taxo,hash}.k2d and k2_pydb_expected.json)
"""
import json
import math
import os
import random
import struct
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_k2_golden import CODE, fmix64, scan      # plain-Python definitions (no state machine), see that file's header

HERE = os.path.dirname(os.path.abspath(__file__))
K, L = 35, 31
SPACED = (0x3ffffffff << 28) | 0x3333333      # 7 alternating positions cleared at the low end (the --minimizer-spaces 7 default)
TOGGLE = 0xe37e28c4271b5a2d
CAPACITY, VALUE_BITS = 4099, 6

# name, rank, external id, parent (index into this list; breadth-first, 0 = the empty node, 1 = root)
TAXA = [
    ("", "", 0, 0),
    ("root", "no rank", 1, 0),
    ("Bacteria", "superkingdom", 2, 1),
    ("Eukaryota", "superkingdom", 2759, 1),
    ("Pseudomonadota", "phylum", 1224, 2),
    ("Chordata", "phylum", 7711, 3),
    ("Escherichia", "genus", 561, 4),
    ("Homo", "genus", 9605, 5),
    ("Pan", "genus", 9596, 5),
    ("Escherichia coli", "species", 562, 6),
    ("Homo sapiens", "species", 9606, 7),
    ("Homo neanderthalensis", "species", 63221, 7),
    ("Pan troglodytes", "species", 9598, 8),
]
PARENT = [t[3] for t in TAXA]


def lca(a, b):
    if a == 0 or b == 0:
        return a or b
    while a != b:
        if a > b:
            a = PARENT[a]
        else:
            b = PARENT[b]
    return a


def is_ancestor(a, b):      # a is an ancestor of b (or b itself)
    if a == 0 or b == 0:
        return False
    while b > a:
        b = PARENT[b]
    return a == b


def table_set(cells, key, taxon):
    h = fmix64(key)
    ck = h >> (32 + VALUE_BITS)
    i = h % CAPACITY
    for _ in range(CAPACITY):
        c = cells[i]
        if c == 0:
            cells[i] = ck << VALUE_BITS | taxon
            return
        if c >> VALUE_BITS == ck:
            cells[i] = ck << VALUE_BITS | lca(c & ((1 << VALUE_BITS) - 1), taxon)
            return
        i = (i + 1) % CAPACITY
    raise RuntimeError("table full")


def table_get(cells, key):
    h = fmix64(key)
    ck = h >> (32 + VALUE_BITS)
    i = h % CAPACITY
    for _ in range(CAPACITY):
        c = cells[i]
        if c == 0:
            return 0
        if c >> VALUE_BITS == ck:
            return c & ((1 << VALUE_BITS) - 1)
        i = (i + 1) % CAPACITY
    return 0


def classify(cells, mates, confidence=0.0, min_hit_groups=2):
    """one unit (a read, or a pair whose mates pool their counts): call, total k-mers, minimizer hit groups"""
    counts, total, groups = {}, 0, 0
    last_min, last_tax = None, 0
    for seq in mates:
        for m in scan(seq, K, L, SPACED, TOGGLE):
            total += 1
            if m is None:                   # k-mer over an ambiguous base: counted, never looked up
                continue
            if m != last_min:
                last_tax = table_get(cells, m)
                last_min = m
                if last_tax:
                    groups += 1
            if last_tax:
                counts[last_tax] = counts.get(last_tax, 0) + 1
    required = math.ceil(confidence * total)
    best, best_score = 0, 0
    for t in sorted(counts):                # Kraken 2 walks an ordered map: ascending taxon id
        score = sum(c for t2, c in counts.items() if is_ancestor(t2, t))
        if score > best_score:
            best, best_score = t, score
        elif score == best_score:
            best = lca(best, t)
    score = counts.get(best, 0)
    while best and score < required:
        score = sum(c for t2, c in counts.items() if is_ancestor(best, t2))
        if score >= required:
            break
        best = PARENT[best]
    if best and groups < min_hit_groups:
        best = 0
    return best, total, groups


def mutate(rng, s, n):
    s = list(s)
    for _ in range(n):
        i = rng.randrange(len(s))
        s[i] = rng.choice([c for c in "ACGT" if c != s[i]])
    return "".join(s)


def revcomp(s):
    return "".join("ACGT"[3 - CODE[c]] if c in CODE else "N" for c in reversed(s))


def main():
    rng = random.Random(20261003)
    genome = {name: "".join(rng.choice("ACGT") for _ in range(700)) for name in ("Escherichia coli", "Homo sapiens", "Pan troglodytes")}
    # Homo neanderthalensis shares its first 300 bases with Homo sapiens (LCA = Homo), Pan shares 150 with Homo sapiens (LCA = Chordata)
    genome["Homo neanderthalensis"] = genome["Homo sapiens"][:300] + "".join(rng.choice("ACGT") for _ in range(300))
    genome["Pan troglodytes"] = genome["Pan troglodytes"][:550] + genome["Homo sapiens"][500:650]
    ids = {t[0]: i for i, t in enumerate(TAXA)}
    cells = [0] * CAPACITY
    for name, g in genome.items():
        for m in scan(g, K, L, SPACED, TOGGLE):
            if m is not None:
                table_set(cells, m, ids[name])
    size = sum(1 for c in cells if c)

    out = os.path.join(HERE, "k2_pydb")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "opts.k2d"), "wb") as f:
        f.write(struct.pack("<QQQQB7xQiii4x", K, L, SPACED, TOGGLE, 1, 0, 1, 0, 0))
    names = b"".join(t[0].encode() + b"\0" for t in TAXA)
    rank_list = []
    for t in TAXA:
        if t[1] not in rank_list:
            rank_list.append(t[1])
    ranks = b"".join(r.encode() + b"\0" for r in rank_list)
    with open(os.path.join(out, "taxo.k2d"), "wb") as f:
        f.write(b"K2TAXDAT" + struct.pack("<QQQ", len(TAXA), len(names), len(ranks)))
        name_off = 0
        for i, t in enumerate(TAXA):
            kids = [j for j in range(2, len(TAXA)) if PARENT[j] == i] if i else []
            if i == 0:
                kids = []
            rank_off = sum(len(r) + 1 for r in rank_list[:rank_list.index(t[1])])
            f.write(struct.pack("<7Q", PARENT[i], kids[0] if kids else 0, len(kids), name_off, rank_off, t[2], 0))
            name_off += len(t[0]) + 1
        f.write(names + ranks)
    with open(os.path.join(out, "hash.k2d"), "wb") as f:
        f.write(struct.pack("<QQQQ", CAPACITY, size, 32 - VALUE_BITS, VALUE_BITS))
        f.write(struct.pack("<%dI" % CAPACITY, *cells))

    # units: (mates, what it is)
    units = []
    hs, hn, pt, ec = genome["Homo sapiens"], genome["Homo neanderthalensis"], genome["Pan troglodytes"], genome["Escherichia coli"]
    units.append(([hs[320:420]], "Homo sapiens, unique part"))
    units.append(([revcomp(hs[320:420])], "the same, reverse complement"))
    units.append(([hs[100:200]], "shared by both Homo species -> Homo"))
    units.append(([hs[250:350]], "straddles the shared / unique border -> Homo sapiens (path score)"))
    units.append(([hs[520:620]], "shared with Pan -> Chordata"))
    units.append(([ec[0:100]], "Escherichia coli"))
    units.append(([mutate(rng, ec[200:330], 2)], "E. coli with two substitutions"))
    units.append((["".join(rng.choice("ACGT") for _ in range(120))], "random: unclassified"))
    units.append(([ec[10:50] + "N" + ec[51:110]], "an N inside: k-mers over it are ambiguous"))
    units.append(([hs[330:364]], "shorter than k: no k-mer"))
    units.append(([hs[330:365]], "exactly k: one k-mer, one hit group -> unclassified under minimum_hit_groups = 2"))
    units.append(([pt[100:200], revcomp(pt[250:350])], "pair, both Pan"))
    units.append(([hs[320:420], revcomp(ec[300:400])], "pair, mates disagree: tie -> LCA = root"))
    units.append(([hs[320:420], "".join(rng.choice("ACGT") for _ in range(100))], "pair, one mate random"))
    units.append((["N" * 60, hn[400:500]], "pair, one mate all N"))
    units.append(([hn[310:450]], "Homo neanderthalensis, unique part"))
    exp = {"k": K, "l": L, "capacity": CAPACITY, "value_bits": VALUE_BITS, "size": size, "spaced": str(SPACED), "toggle": str(TOGGLE),
           "parents": PARENT, "external": [t[2] for t in TAXA], "names": [t[0] for t in TAXA], "ranks": [t[1] for t in TAXA], "units": []}
    for mates, what in units:
        for conf, mhg in ((0.0, 2), (0.5, 2), (0.0, 1), (1.0, 3)):
            call, total, groups = classify(cells, mates, conf, mhg)
            exp["units"].append({"mates": mates, "what": what, "confidence": conf, "min_hit_groups": mhg,
                                 "call": call, "taxid": TAXA[call][2], "total_kmers": total, "hit_groups": groups})
    with open(os.path.join(HERE, "k2_pydb_expected.json"), "w") as f:
        json.dump(exp, f, indent=0)
    calls = [u["call"] for u in exp["units"] if u["confidence"] == 0.0 and u["min_hit_groups"] == 2]
    print("cells used", size, "calls at defaults:", [TAXA[c][0] or "-" for c in calls])


if __name__ == "__main__":
    main()
