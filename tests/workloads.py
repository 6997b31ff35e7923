"""Shared seeded workloads for the CPU and GPU suites (5 x 1 Mb mini reference)."""
import numpy as np

CFG1_CONTIGS = [1_000_000] * 5
CFG1_REF_SEED, CFG1_READ_SEED = 0x5C2B0001, 0x5C2B0002


def cfg1(O, n_records=20000):
    P = O.ref_params(CFG1_REF_SEED, CFG1_CONTIGS)
    R = O.read_params(CFG1_READ_SEED)
    ref = O.synth_ref(P, 0, P.genome_len)
    reads = O.synth_reads(P, R, 0, n_records)
    off = np.arange(n_records + 1, dtype=np.uint64) * R.read_len
    seqs = [ref[P.contig_start[i]:P.contig_start[i + 1]] for i in range(len(CFG1_CONTIGS))]
    return P, R, ref, seqs, reads, off


def edge_reads(ref, seed=7):
    """Ragged / degenerate records: empty, < k, == k, first window, N runs, lower case, low complexity, random."""
    rng = np.random.default_rng(seed)
    recs = [b"", bytes(ref[1000:1010]), bytes(ref[2000:2021]), bytes(ref[3000:3031]), bytes(ref[4000:4032]),
            b"N" * 150, bytes(ref[5000:5150]).lower()]
    r = bytearray(ref[6000:6150]); r[40] = ord("N"); r[41] = ord("n"); r[100] = ord("R"); recs.append(bytes(r))
    recs += [b"A" * 150, b"AC" * 75, b"ACG" * 50, b"AAAAC" * 30, b"T" * 21, b"ACGT" * 100]
    for L in [22, 35, 64, 99, 151, 250, 300, 400, 1000]:
        s = int(rng.integers(0, 900000)); recs.append(bytes(ref[s:s + L]))
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    for L in [150, 250]:
        s = int(rng.integers(0, 900000)); recs.append(bytes(ref[s:s + L]).translate(comp)[::-1])
    for _ in range(40):
        recs.append(bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, int(rng.integers(1, 300)))]))
    bases = np.frombuffer(b"".join(recs), dtype=np.uint8)
    offs = np.zeros(len(recs) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(x) for x in recs])
    return recs, bases, offs


def k2_taxonomy(n_random=300, seed=3):
    """A small NCBI-shaped taxonomy in Kraken 2's layout: breadth-first internal ids, children consecutive.
    Contains the true lineage root -> cellular organisms -> Eukaryota -> Metazoa -> Chordata -> Mammalia -> Primates ->
    Hominidae -> Homo -> Homo sapiens (+ a sibling species and genus), and a random bacterial tree.
    Returns (parents, externals, names, ranks, ids) with ids = {name: internal id}."""
    rng = np.random.default_rng(seed)
    # tree as nested spec: (name, ext, rank, children)
    human = ("Homo", 9605, "genus", [("Homo sapiens", 9606, "species", []), ("Homo heidelbergensis", 1425170, "species", [])])
    pan = ("Pan", 9596, "genus", [("Pan troglodytes", 9598, "species", [])])
    euk = ("Eukaryota", 2759, "superkingdom", [("Opisthokonta", 33154, "clade", [("Metazoa", 33208, "kingdom", [
        ("Chordata", 7711, "phylum", [("Mammalia", 40674, "class", [("Primates", 9443, "order", [
            ("Hominidae", 9604, "family", [human, pan])])])])])])])
    next_ext = [100000]

    def rand_tree(depth, ranks):
        ext = next_ext[0]; next_ext[0] += 1
        kids = []
        if depth < len(ranks) - 1:
            for _ in range(int(rng.integers(1, 4))):
                kids.append(rand_tree(depth + 1, ranks))
        return (f"{ranks[depth]}_{ext}", ext, ranks[depth], kids)

    branks = ["phylum", "class", "order", "family", "genus", "species"]
    bact_kids = []
    while sum(1 for _ in _iter_spec(bact_kids)) < n_random:
        bact_kids.append(rand_tree(0, branks))
    bact = ("Bacteria", 2, "superkingdom", bact_kids)
    root = ("root", 1, "no rank", [("cellular organisms", 131567, "no rank", [euk, bact])])
    parents, externals, names, ranks, ids = [0], [0], [""], [""], {}
    queue = [(root, 0)]
    # breadth-first numbering; a node's children are appended together, so their ids are consecutive
    order = []
    while queue:
        spec, par = queue.pop(0)
        my = len(parents)
        parents.append(par); externals.append(spec[1]); names.append(spec[0]); ranks.append(spec[2]); ids[spec[0]] = my
        order.append((spec, my))
        for ch in spec[3]:
            queue.append((ch, my))
    return parents, externals, names, ranks, ids


def _iter_spec(specs):
    for s in specs:
        yield s
        yield from _iter_spec(s[3])
