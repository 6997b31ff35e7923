"""minimap2 index files (.mmi) as the index argument (`cleaner.rs:475-479` passes any path minimap2 accepts).

The file is written here from the statement of minimap2's `mm_idx_dump` that `csrc/sh_index.hip` restates (magic "MMI\\2"; uint32 w, k, b,
n_seq, flag; per sequence uint8 name length, name, uint32 length; 2^b buckets of int32 n, n 8-byte positions, uint32 size, size 16-byte
pairs; the sequences as 4-bit codes, eight to a uint32) - PARITY UNPINNED: there is no minimap2 on this box to write one.  What is checked:
the importer skips the minimizer tables by their counts, decodes the sequences (N included), lets the file's k and w prevail over the
preset's, refuses HPC and sequence-less indexes by name, and the index it builds equals the one built from the FASTA of the same sequences.
"""
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S():
    from scrubby_amd import lib
    lib.require_gpu()
    return lib


def write_mmi(path, seqs, names, k, w, b=14, flag=0, rng=None, with_seq=True):
    code = np.full(256, 4, dtype=np.uint32)
    for i, c in enumerate(b"ACGT"):
        code[c] = i
        code[ord(chr(c).lower())] = i
    with open(path, "wb") as f:
        f.write(b"MMI\x02")
        f.write(struct.pack("<5I", w, k, b, len(seqs), flag))
        for s, nm in zip(seqs, names):
            nb = nm.encode()[:255]
            f.write(struct.pack("<B", len(nb)) + nb + struct.pack("<I", len(s)))
        for i in range(1 << b):      # minimizer tables the importer must step over: some empty, some with positions and hash entries
            n = int(rng.integers(0, 4)) if rng is not None and i % 97 == 0 else 0
            size = int(rng.integers(0, 3)) if rng is not None and i % 89 == 0 else 0
            f.write(struct.pack("<i", n) + rng.integers(0, 1 << 62, n, dtype=np.uint64).tobytes() if n else struct.pack("<i", 0))
            f.write(struct.pack("<I", size))
            if size:
                f.write(rng.integers(0, 1 << 62, 2 * size, dtype=np.uint64).tobytes())
        if with_seq:
            allb = np.concatenate([np.frombuffer(bytes(s), dtype=np.uint8) for s in seqs])
            c4 = code[allb]
            pad = (-len(c4)) % 8
            c4 = np.concatenate([c4, np.zeros(pad, dtype=np.uint32)]).reshape(-1, 8)
            words = np.zeros(len(c4), dtype=np.uint32)
            for j in range(8):
                words |= c4[:, j] << np.uint32(4 * j)
            f.write(words.astype("<u4").tobytes())


def test_an_mmi_file_gives_the_index_its_sequences_give(S, oracle, tmp_path):
    contigs = [180_000, 90_001, 777]
    Po = oracle.ref_params(0x5C2B0D01, contigs, sat_pct=8, rep_pct=30, n_sat_fam=3, n_rep_fam=12)
    ref = oracle.synth_ref(Po, 0, Po.genome_len)
    seqs = [bytearray(ref[Po.contig_start[i]:Po.contig_start[i + 1]]) for i in range(len(contigs))]
    seqs[0][5000:5040] = b"N" * 40          # ambiguity codes survive the 4-bit form as N
    seqs[2][10:12] = b"nn"
    rng = np.random.default_rng(7)
    fa = tmp_path / "ref.fa"
    with open(fa, "wb") as f:
        for i, s in enumerate(seqs):
            f.write(b">c%d\n" % i + bytes(s).upper() + b"\n")
    mmi = tmp_path / "ref.mmi"
    write_mmi(mmi, seqs, ["c0", "c1", "c2"], k=21, w=11, rng=rng)
    a = S.Index.build_fasta(str(fa), S.preset("sr"))
    b = S.Index.build_fasta(str(mmi), S.preset("sr"))
    ia, ib = a.info(), b.info()
    for key in ("k", "w", "mid_occ", "n_contigs", "n_bases", "n_minimizers", "n_keys", "n_slots", "n_positions"):
        assert ia[key] == ib[key], key
    da = oracle.Index.wrap(*a.export(), 11, 21).dump()      # (keys, counts, positions): slot placement depends on the build's atomics
    db = oracle.Index.wrap(*b.export(), 11, 21).dump()
    assert all(np.array_equal(x, y) for x, y in zip(da, db))
    # reads classified alike (the reference bases for the extension filter come from the file too)
    Ro = oracle.read_params(0x5C2B0D02)
    n = 2000
    bases = oracle.synth_reads(Po, Ro, 0, n)
    offs = np.arange(n + 1, dtype=np.uint64) * 150
    fa_flags, _, _, rc = a.classify(bases, offs, want_trace=False)
    mm_flags, _, _, rc2 = b.classify(bases, offs, want_trace=False)
    assert rc == 0 and rc2 == 0 and np.array_equal(fa_flags, mm_flags) and 0 < int(fa_flags.sum()) < n


def test_the_files_k_and_w_prevail_over_the_presets(S, oracle, tmp_path):
    contigs = [120_000]
    Po = oracle.ref_params(0x5C2B0D03, contigs, sat_pct=0, rep_pct=10, n_sat_fam=1, n_rep_fam=4)
    ref = oracle.synth_ref(Po, 0, Po.genome_len)
    mmi = tmp_path / "k15.mmi"
    write_mmi(mmi, [ref], ["chr"], k=15, w=10, rng=np.random.default_rng(1))
    idx = S.Index.build_fasta(str(mmi), S.preset("sr"))          # sr asks for k = 21, w = 11
    assert (idx.info()["k"], idx.info()["w"]) == (15, 10) and (idx.opts.k, idx.opts.w) == (15, 10)
    ref_idx = S.Index.build([bytes(ref)], S.preset("map-ont"))   # k = 15, w = 10
    assert all(np.array_equal(x, y) for x, y in zip(oracle.Index.wrap(*idx.export(), 10, 15).dump(), oracle.Index.wrap(*ref_idx.export(), 10, 15).dump()))


def test_indexes_this_path_cannot_take_are_refused_by_name(S, oracle, tmp_path):
    seq = [b"ACGT" * 500]
    hpc = tmp_path / "hpc.mmi"
    write_mmi(hpc, seq, ["x"], k=19, w=10, flag=1)
    with pytest.raises(S.ScrubbyHipError, match="homopolymer"):
        S.Index.build_fasta(str(hpc), S.preset("map-ont"))
    noseq = tmp_path / "noseq.mmi"
    write_mmi(noseq, seq, ["x"], k=15, w=10, flag=2, with_seq=False)
    with pytest.raises(S.ScrubbyHipError, match="without sequences"):
        S.Index.build_fasta(str(noseq), S.preset("map-ont"))
    cut = tmp_path / "cut.mmi"
    write_mmi(cut, seq, ["x"], k=15, w=10)
    data = open(cut, "rb").read()
    open(cut, "wb").write(data[:-100])
    with pytest.raises(S.ScrubbyHipError, match="truncated"):
        S.Index.build_fasta(str(cut), S.preset("map-ont"))
