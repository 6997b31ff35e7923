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
