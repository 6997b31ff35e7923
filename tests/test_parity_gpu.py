"""GPU parity: libscrubby_hip (through its C ABI) vs the CPU oracle, bit-exact.

Every comparison is on integers: reference bytes, index content, per-read decision trace
(n_mini, n_seed, n_anchor, rep_len, rechained, n_chain, best_score) and the flag.
"""
import numpy as np
import pytest

from tests import workloads as W

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S():
    from scrubby_amd import lib
    lib.require_gpu()
    return lib


@pytest.fixture(scope="module")
def cfg1(oracle):
    return W.cfg1(oracle, 20000)


@pytest.fixture(scope="module")
def gpu_index(S, cfg1):
    P, R, ref, seqs, reads, off = cfg1
    return S.Index.build([bytes(s) for s in seqs], S.preset("sr"))


@pytest.fixture(scope="module")
def cpu_index(oracle, cfg1):
    P, R, ref, seqs, reads, off = cfg1
    return oracle.Index.build(seqs, 11, 21)


def assert_trace_equal(S, gf, gt, of, ot):
    assert np.array_equal(gf, of), f"{int((gf != of).sum())} flags differ"
    for name in S.TRACE_FIELDS:
        bad = np.where(gt[name] != ot[name])[0]
        assert len(bad) == 0, f"trace.{name}: {len(bad)} differ, first read {bad[0]}: gpu={gt[name][bad[0]]} cpu={ot[name][bad[0]]}"


def test_presets_match_oracle(S, oracle):
    for name in ("sr", "map-ont", "lr:hq", "map-hifi"):
        g, o = S.preset(name), oracle.preset(name)
        for f, _ in S.Opts._fields_:
            assert getattr(g, f) == getattr(o, f), (name, f)


def test_synth_device_matches_cpu(S, oracle, cfg1):
    import torch
    P, R, ref, seqs, reads, off = cfg1
    Pg, Rg = S.ref_params(W.CFG1_REF_SEED, W.CFG1_CONTIGS), S.read_params(W.CFG1_READ_SEED)
    d = torch.empty(P.genome_len + 64, dtype=torch.uint8, device="cuda")
    S.synth_ref_device(Pg, 0, P.genome_len, d)
    assert np.array_equal(d[:P.genome_len].cpu().numpy(), ref)
    n = len(off) - 1
    dr = torch.empty(n * 150 + 64, dtype=torch.uint8, device="cuda")
    do = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    S.synth_reads_device(Pg, Rg, 0, n, dr, do)
    assert np.array_equal(dr[:n * 150].cpu().numpy(), reads)
    assert np.array_equal(do.cpu().numpy().astype(np.uint64), off)


def test_index_content_identical(S, oracle, gpu_index, cpu_index):
    slots, pos = gpu_index.export()
    wrapped = oracle.Index.wrap(slots, pos, 11, 21)
    k1, c1, p1 = wrapped.dump()
    k2, c2, p2 = cpu_index.dump()
    assert np.array_equal(k1, k2) and np.array_equal(c1, c2) and np.array_equal(p1, p2)
    info = gpu_index.info()
    assert info["n_keys"] == len(k2) and info["n_minimizers"] == int(c2.sum())


def test_index_device_build_equals_host_build(S, oracle, cfg1, gpu_index):
    import torch
    P, R, ref, seqs, reads, off = cfg1
    d = torch.from_numpy(np.concatenate([ref, np.zeros(64, np.uint8)])).cuda()
    idx2 = S.Index.build_device(d, [P.contig_start[i] for i in range(len(W.CFG1_CONTIGS) + 1)], S.preset("sr"))
    a = oracle.Index.wrap(*gpu_index.export(), 11, 21).dump()
    b = oracle.Index.wrap(*idx2.export(), 11, 21).dump()
    assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_classify_cfg1_trace_parity(S, oracle, cfg1, gpu_index, cpu_index):
    P, R, ref, seqs, reads, off = cfg1
    gf, gt, st, rc = gpu_index.classify(reads, off, want_trace=True)
    of, ot = cpu_index.classify(oracle.preset("sr"), reads, off, threads=8)
    assert rc == 0
    assert_trace_equal(S, gf, gt, of, ot)
    truth = oracle.synth_truth(P, R, 0, len(gf))
    assert int(gf.sum()) == int(truth.sum()) == st["n_host"]      # reads_removed count
    assert st["n_chain_large"] > 0 and st["n_chain_small"] > 0 and st["n_no_seed"] > 0   # every kernel exercised


def test_classify_flag_only_equals_trace_mode(S, oracle, cfg1, gpu_index):
    P, R, ref, seqs, reads, off = cfg1
    f1, _, _, _ = gpu_index.classify(reads, off, want_trace=False)
    f2, _, _, _ = gpu_index.classify(reads, off, want_trace=True)
    assert np.array_equal(f1, f2)


def test_edge_cases(S, oracle, cfg1, gpu_index, cpu_index):
    P, R, ref, seqs, reads, off = cfg1
    recs, bases, offs = W.edge_reads(ref)
    gf, gt, st, rc = gpu_index.classify(bases, offs, want_trace=True)
    of, ot = cpu_index.classify(oracle.preset("sr"), bases, offs, threads=1)
    assert_trace_equal(S, gf, gt, of, ot)
    assert gf[0] == 2 and rc == S.SH_ERR_EMPTY_READ        # reference: Err("Sequence is empty") aborts the run
    assert gf[6] == 1                                       # lower-case host read maps


def test_misaligned_and_offset_batches(S, oracle, cfg1, gpu_index, cpu_index):
    """offsets[0] != 0 and a base pointer that is not 16-B aligned."""
    P, R, ref, seqs, reads, off = cfg1
    n = 1000
    pad = np.concatenate([np.frombuffer(b"GATTACA", dtype=np.uint8), reads[: n * 150]])
    off2 = off[: n + 1] + np.uint64(7)
    gf, gt, st, rc = gpu_index.classify(pad, off2, want_trace=True)
    of, ot = cpu_index.classify(oracle.preset("sr"), reads[: n * 150], off[: n + 1], threads=2)
    assert_trace_equal(S, gf, gt, of, ot)


def test_device_context_repeatable_and_chunked(S, oracle, cfg1, gpu_index, cpu_index):
    import torch
    P, R, ref, seqs, reads, off = cfg1
    n = len(off) - 1
    d_reads = torch.from_numpy(np.concatenate([reads, np.zeros(64, np.uint8)])).cuda()
    d_off = torch.from_numpy(off.astype(np.int64)).cuda()
    of, ot = cpu_index.classify(oracle.preset("sr"), reads, off, threads=8)
    for chunk in (n, 4096 + 64):                     # one launch, and several launches with a ragged tail
        ctx = S.Context(gpu_index, chunk, n * 150, 150)
        fl = torch.zeros(n, dtype=torch.uint8, device="cuda")
        tr = torch.zeros((n, len(S.TRACE_FIELDS)), dtype=torch.int32, device="cuda")
        for _ in range(2):
            ctx.classify(d_reads[: n * 150], d_off, fl, tr)
        gt = tr.cpu().numpy().view(S.TRACE_DTYPE).reshape(-1)
        assert_trace_equal(S, fl.cpu().numpy(), gt, of, ot)
        ctx.close()


def test_small_arena_defers_and_still_matches(S, oracle, cfg1, gpu_index, cpu_index, monkeypatch):
    """A chain arena too small for the batch forces the deferral loop; results must not change."""
    P, R, ref, seqs, reads, off = cfg1
    monkeypatch.setenv("SCRUBBY_HIP_ARENA_MB", "8")
    gf, gt, st, rc = gpu_index.classify(reads, off, want_trace=True)
    of, ot = cpu_index.classify(oracle.preset("sr"), reads, off, threads=8)
    assert_trace_equal(S, gf, gt, of, ot)


def test_map_ont_preset_short_reads(S, oracle, cfg1):
    """k=15,w=10, mid_occ derived from the index (mm_mapopt_update)."""
    P, R, ref, seqs, reads, off = cfg1
    go = S.preset("map-ont")
    gidx = S.Index.build([bytes(s) for s in seqs], go)
    cidx = oracle.Index.build(seqs, 10, 15)
    oo = cidx.update_opts(oracle.preset("map-ont"))
    assert gidx.info()["mid_occ"] == oo.mid_occ
    n = 4000
    gf, gt, st, rc = gidx.classify(reads[: n * 150], off[: n + 1], want_trace=True)
    of, ot = cidx.classify(oo, reads[: n * 150], off[: n + 1], threads=8)
    assert_trace_equal(S, gf, gt, of, ot)


def test_idempotent_and_order_independent(S, cfg1, gpu_index):
    """Size-independent properties: same batch twice -> same flags; permuting records permutes flags."""
    P, R, ref, seqs, reads, off = cfg1
    n = 5000
    f1, _, _, _ = gpu_index.classify(reads[: n * 150], off[: n + 1])
    f2, _, _, _ = gpu_index.classify(reads[: n * 150], off[: n + 1])
    assert np.array_equal(f1, f2)
    perm = np.random.default_rng(3).permutation(n)
    shuf = reads[: n * 150].reshape(n, 150)[perm].reshape(-1)
    f3, _, _, _ = gpu_index.classify(shuf, off[: n + 1])
    assert np.array_equal(f3, f1[perm])


def test_index_save_load_roundtrip(S, cfg1, gpu_index, tmp_path):
    P, R, ref, seqs, reads, off = cfg1
    path = str(tmp_path / "mini.shidx")
    gpu_index.save(path)
    idx2 = S.Index.load(path, S.preset("sr"))
    n = 2000
    f1, t1, _, _ = gpu_index.classify(reads[: n * 150], off[: n + 1], want_trace=True)
    f2, t2, _, _ = idx2.classify(reads[: n * 150], off[: n + 1], want_trace=True)
    assert np.array_equal(f1, f2) and np.array_equal(t1, t2)
    a, b = gpu_index.export_ref(), idx2.export_ref()          # the cache carries the reference bases the extension stage needs
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_index_cache_is_checked_before_it_is_trusted(S, gpu_index, tmp_path):
    """A corrupt, truncated, stale or foreign cache is an error, never an index (ADVICE: header trusted blindly)."""
    good = tmp_path / "good.shidx"
    gpu_index.save(str(good))
    raw = bytearray(good.read_bytes())

    def load_fails(data, what):
        p = tmp_path / "bad.shidx"
        p.write_bytes(bytes(data))
        with pytest.raises(S.ScrubbyHipError) as ei:
            S.Index.load(str(p), S.preset("sr"))
        assert ei.value.status == 7 and what in ei.value.message, ei.value.message

    load_fails(raw[:len(raw) - 4096], "file size")                       # truncated
    load_fails(raw + b"\0" * 16, "file size")                            # trailing bytes
    flipped = bytearray(raw); flipped[len(raw) // 2] ^= 0x40
    load_fails(flipped, "checksum")                                      # one bit of the payload
    hdr = bytearray(raw); hdr[8 + 4 * 4] ^= 0x01                         # lg_slots no longer matches n_slots
    load_fails(hdr, "header fields")
    old = bytearray(raw); old[:8] = b"SHIDX001"
    load_fails(old, "older scrubby-hip")
    load_fails(b"not an index at all" * 10, "not a scrubby-hip index")
    # an index derived its mid_occ for one preset's occurrence parameters: another preset with the same k, w must not reuse it silently
    ont = S.Index.build([b"ACGT" * 5000 + b"GATTACA" * 3000], S.preset("map-ont"))
    p = tmp_path / "ont.shidx"
    ont.save(str(p))
    o = S.preset("map-ont"); o.min_mid_occ = 50; o.max_mid_occ = 500
    re = S.Index.load(str(p), o)
    with pytest.raises(S.ScrubbyHipError) as ei:
        re.classify(np.frombuffer(b"ACGT" * 100, np.uint8), np.array([0, 400], np.uint64))
    assert "occurrence parameters" in ei.value.message


def test_gzip_reference_is_read_with_zlib_and_truncation_is_an_error(S, oracle, cfg1, gpu_index, tmp_path):
    """ADVICE: the gzip reference used to go through popen("gzip -dc '<path>'") - a quote in the path ran in a shell, and a
    truncated stream gave a silently short reference."""
    import gzip
    P, R, ref, seqs, reads, off = cfg1
    fa = b"".join(b">c%d\n" % i + bytes(s) + b"\n" for i, s in enumerate(seqs))
    odd = tmp_path / "it's a ref; echo pwned > x.fa.gz"
    odd.write_bytes(gzip.compress(fa))
    import os
    os.environ["SCRUBBY_HIP_FASTA_HOST"] = "1"
    try:
        got = oracle.Index.wrap(*S.Index.build_fasta(str(odd), S.preset("sr")).export(), 11, 21).dump()
        want = oracle.Index.wrap(*gpu_index.export(), 11, 21).dump()
        assert all(np.array_equal(a, b) for a, b in zip(got, want))
        cut = tmp_path / "cut.fa.gz"
        cut.write_bytes(gzip.compress(fa)[:-2000])
        with pytest.raises(S.ScrubbyHipError) as ei:
            S.Index.build_fasta(str(cut), S.preset("sr"))
        assert ei.value.status == 7
        long_line = tmp_path / "long.fq"                                # a FASTQ reference whose sequence line is longer than the reader's buffer
        seq = bytes(seqs[0][:200_000])
        long_line.write_bytes(b"@r\n" + seq + b"\n+\n" + b"I" * len(seq) + b"\n")
        i2 = S.Index.build_fasta(str(long_line), S.preset("sr"))
        assert i2.info()["n_contigs"] == 1 and i2.info()["n_bases"] == len(seq)
    finally:
        os.environ.pop("SCRUBBY_HIP_FASTA_HOST")
    assert not (tmp_path / "x.fa.gz").exists() and not os.path.exists("x.fa.gz")


def _ont_like_reads(ref, n, seed, min_len=1200, max_len=30000, sub=0.02, indel=0.015):
    """Long noisy reads: log-normal lengths, 2 % substitutions, 1.5 % insertions, 1.5 % deletions (defaults), either strand."""
    rng = np.random.default_rng(seed)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    acgt = b"ACGT"
    recs = []
    for i in range(n):
        L = int(min(max(rng.lognormal(8.0, 0.7), min_len), max_len))
        if i % 5 == 4:                                     # unrelated read
            recs.append(bytes(np.frombuffer(acgt, dtype=np.uint8)[rng.integers(0, 4, L)]))
            continue
        s = int(rng.integers(0, 1_000_000 - L))
        src = bytes(ref[s:s + L])
        out = bytearray()
        for c in src:
            u = rng.random()
            if u < indel:
                continue
            if u < 2 * indel:
                out.append(acgt[rng.integers(0, 4)])
            out.append(acgt[rng.integers(0, 4)] if u > 1.0 - sub else c)
        b = bytes(out)
        recs.append(b.translate(comp)[::-1] if i % 2 else b)
    bases = np.frombuffer(b"".join(recs), dtype=np.uint8)
    offs = np.zeros(len(recs) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(x) for x in recs])
    return recs, bases, offs


def test_long_reads_map_ont(S, oracle, cfg1):
    """BASELINE config 4 in miniature: noisy multi-kb reads, map-ont preset (k=15, w=10, mid_occ from the index)."""
    P, R, ref, seqs, reads, off = cfg1
    go = S.preset("map-ont")
    gidx = S.Index.build([bytes(s) for s in seqs], go)
    cidx = oracle.Index.build(seqs, 10, 15)
    oo = cidx.update_opts(oracle.preset("map-ont"))
    recs, bases, offs = _ont_like_reads(ref, 60, 42)
    gf, gt, st, rc = gidx.classify(bases, offs, want_trace=True)
    of, ot = cidx.classify(oo, bases, offs, threads=8)
    assert_trace_equal(S, gf, gt, of, ot)
    assert int(gf.sum()) == 48 and int(gf[4::5].sum()) == 0      # every reference-derived read maps, no random one does


def test_long_reads_map_hifi_and_lr_hq(S, oracle, cfg1):
    """Preset::MapHifi / Preset::LrHq (cleaner.rs:458,465): k = w = 19, max_gap 10000, mid_occ clamped to [50, 500]; accurate multi-kb reads."""
    from scrubby_amd.lib import ScrubbyHipError
    P, R, ref, seqs, reads, off = cfg1
    recs, bases, offs = _ont_like_reads(ref, 40, 43, sub=0.005, indel=0.002)
    for name in ("map-hifi", "lr:hq"):
        go = S.preset(name)
        assert (go.k, go.w, go.max_gap, go.min_mid_occ, go.max_mid_occ) == (19, 19, 10000, 50, 500)
        gidx = S.Index.build([bytes(s) for s in seqs], go)
        cidx = oracle.Index.build(seqs, 19, 19)
        oo = cidx.update_opts(oracle.preset(name))
        gf, gt, st, rc = gidx.classify(bases, offs, want_trace=True)
        of, ot = cidx.classify(oo, bases, offs, threads=8)
        assert_trace_equal(S, gf, gt, of, ot)
        assert int(gf.sum()) == 32 and int(gf[4::5].sum()) == 0
    assert S.preset("map-hifi").min_dp_max == 200 and S.preset("lr:hq").min_dp_max != 200
    for name, why in (("map-pb", "homopolymer"), ("splice", "splice"), ("ava-ont", "all-vs-all"), ("asm5", "RMQ")):
        with pytest.raises(ScrubbyHipError, match=why):
            S.preset(name)


def test_long_read_generator_device_matches_cpu(S, oracle):
    """BASELINE config 4 stand-in: the device generator, its CPU twin and bench.py's numpy length ladder agree."""
    import torch
    import bench
    Pg = S.ref_params(W.CFG1_REF_SEED, W.CFG1_CONTIGS)
    Rg = S.read_params(0x5C2B0020, host_pct=50, sub_per_10k=500, n_read_pct=0)
    Po = oracle.ref_params(W.CFG1_REF_SEED, W.CFG1_CONTIGS)
    Ro = oracle.read_params(0x5C2B0020, host_pct=50, sub_per_10k=500, n_read_pct=0)
    n = 300
    cpu, offs = oracle.synth_long_reads(Po, Ro, 7, n)
    assert np.array_equal(bench.long_read_lengths(0x5C2B0020, 7, n), np.diff(offs.astype(np.int64)).astype(np.uint32))
    d_off = torch.from_numpy(offs.astype(np.int64)).cuda()
    d = torch.empty(len(cpu) + 64, dtype=torch.uint8, device="cuda")
    S.synth_long_reads_device(Pg, Rg, 7, n, d_off, len(cpu), d)
    assert np.array_equal(d[:len(cpu)].cpu().numpy(), cpu)
    # with indels switched on (n_read_pct != 0: 2 % substitutions + 1.56 % insertions + 1.56 % deletions, bench.py --workload ont)
    Rgi = S.read_params(0x5C2B0020, host_pct=50, sub_per_10k=200, n_read_pct=1)
    Roi = oracle.read_params(0x5C2B0020, host_pct=50, sub_per_10k=200, n_read_pct=1)
    cpu_i, offs_i = oracle.synth_long_reads(Po, Roi, 7, n)
    assert np.array_equal(offs_i, offs) and not np.array_equal(cpu_i, cpu)
    S.synth_long_reads_device(Pg, Rgi, 7, n, d_off, len(cpu_i), d)
    assert np.array_equal(d[:len(cpu_i)].cpu().numpy(), cpu_i)
    # and they classify identically (map-ont, legacy long-read path)
    seqs = [oracle.synth_ref(Po, Po.contig_start[i], 1_000_000) for i in range(5)]
    gidx = S.Index.build([bytes(s) for s in seqs], S.preset("map-ont"))
    cidx = oracle.Index.build(seqs, 10, 15)
    oo = cidx.update_opts(oracle.preset("map-ont"))
    gf, gt, st, rc = gidx.classify(cpu, offs, want_trace=True)
    of, ot = cidx.classify(oo, cpu, offs, threads=8)
    assert_trace_equal(S, gf, gt, of, ot)


def _short_indel_reads(ref, n, seed, L=150):
    """150-bp reads with 1 % substitutions, 1 % insertions, 1 % deletions, either strand (the gap-penalty float path)."""
    rng = np.random.default_rng(seed)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    acgt = b"ACGT"
    recs = []
    for i in range(n):
        s = int(rng.integers(0, len(ref) - 2 * L))
        out = bytearray()
        p = s
        while len(out) < L:
            u = rng.random()
            if u < 0.01:
                p += 1
                continue
            if u < 0.02:
                out.append(acgt[rng.integers(0, 4)])
                continue
            out.append(acgt[rng.integers(0, 4)] if u > 0.99 else ref[p])
            p += 1
        b = bytes(out[:L])
        recs.append(b.translate(comp)[::-1] if i % 2 else b)
    bases = np.frombuffer(b"".join(recs), dtype=np.uint8)
    offs = np.arange(n + 1, dtype=np.uint64) * L
    return bases, offs


def test_short_reads_with_indels(S, oracle, cfg1, gpu_index, cpu_index):
    """Indels put dd != 0 into nearly every chain: comput_sc's linear + log2 gap penalty decides scores."""
    P, R, ref, seqs, reads, off = cfg1
    bases, offs = _short_indel_reads(ref, 4000, 11)
    gf, gt, st, rc = gpu_index.classify(bases, offs, want_trace=True)
    of, ot = cpu_index.classify(oracle.preset("sr"), bases, offs, threads=8)
    assert_trace_equal(S, gf, gt, of, ot)
    assert int(gf.sum()) > 3500


def test_long_read_front_end_all_paths(S, oracle):
    """1500 stand-in long reads (satellite reads included): the segment-parallel front end takes every read (no
    re-sketch), query-minimizer thinning, multi-tile seed selection, the giant sort and the cluster queue all run;
    traces match the oracle, and the flag-only mode (first-chain shortcut) gives the same flags."""
    Po = oracle.ref_params(W.CFG1_REF_SEED, W.CFG1_CONTIGS)
    Ro = oracle.read_params(0x5C2B0021, host_pct=60, sub_per_10k=200, n_read_pct=1)      # substitutions + indels
    n = 1500
    bases, offs = oracle.synth_long_reads(Po, Ro, 11, n)
    seqs = [oracle.synth_ref(Po, Po.contig_start[i], 1_000_000) for i in range(5)]
    gidx = S.Index.build([bytes(s) for s in seqs], S.preset("map-ont"))
    cidx = oracle.Index.build(seqs, 10, 15)
    oo = cidx.update_opts(oracle.preset("map-ont"))
    gf, gt, st, rc = gidx.classify(bases, offs, want_trace=True)
    of, ot = cidx.classify(oo, bases, offs, threads=8)
    assert_trace_equal(S, gf, gt, of, ot)
    assert st["n_resketch"] == 0 and st["n_clusters"] > 0
    assert int((ot["n_mini"] < ot["n_seed"]).sum()) == 0
    gf2, _, st2, _ = gidx.classify(bases, offs, want_trace=False)
    assert np.array_equal(gf2, of)


def test_long_reads_chunked_context_and_small_arena(S, oracle, monkeypatch):
    """Long reads through a device Context in several launches (offsets of a chunk do not start at 0), then with a chain
    arena far too small for the batch (deferral loop with the cluster queue): flags and traces must not change."""
    import torch
    Po = oracle.ref_params(W.CFG1_REF_SEED, W.CFG1_CONTIGS)
    Ro = oracle.read_params(0x5C2B0022, host_pct=60, sub_per_10k=500, n_read_pct=0)
    n = 900
    bases, offs = oracle.synth_long_reads(Po, Ro, 3, n)
    seqs = [oracle.synth_ref(Po, Po.contig_start[i], 1_000_000) for i in range(5)]
    gidx = S.Index.build([bytes(s) for s in seqs], S.preset("map-ont"))
    cidx = oracle.Index.build(seqs, 10, 15)
    oo = cidx.update_opts(oracle.preset("map-ont"))
    of, ot = cidx.classify(oo, bases, offs, threads=8)
    d_reads = torch.from_numpy(np.concatenate([bases, np.zeros(64, np.uint8)])).cuda()
    d_off = torch.from_numpy(offs.astype(np.int64)).cuda()
    max_len = int(np.diff(offs.astype(np.int64)).max())
    for chunk in (n, 256):
        ctx = S.Context(gidx, chunk, len(bases), max_len)
        fl = torch.zeros(n, dtype=torch.uint8, device="cuda")
        tr = torch.zeros((n, len(S.TRACE_FIELDS)), dtype=torch.int32, device="cuda")
        ctx.classify(d_reads[: len(bases)], d_off, fl, tr)
        gt = tr.cpu().numpy().view(S.TRACE_DTYPE).reshape(-1)
        assert_trace_equal(S, fl.cpu().numpy(), gt, of, ot)
        fl.zero_()
        ctx.classify(d_reads[: len(bases)], d_off, fl, None)            # flag-only mode
        assert np.array_equal(fl.cpu().numpy(), of)
        ctx.close()
    monkeypatch.setenv("SCRUBBY_HIP_ARENA_MB", "96")
    gf, gt, st, rc = gidx.classify(bases, offs, want_trace=True)
    assert_trace_equal(S, gf, gt, of, ot)


def test_pair_test_flag_only_on_repeat_rich_reference(S, oracle, monkeypatch):
    """The flag-only pair test (k_expand / pair_decides) on a reference made of few, large repeat families, so
    that most reads reach the repeat path with dozens to thousands of anchors: flags with the shortcut, without it
    (SCRUBBY_HIP_NO_PAIR) and from the oracle's full chaining + backtrack must be identical.  Reads: substitution-only
    (either strand), indel reads (dd != 0: the test must fall through), reads with an internal duplication (two seeds
    with one key: the distinct-key premise fails and the read takes the full path), and non-host reads."""
    contigs = [600_000, 400_000]
    Po = oracle.ref_params(0x5C2B0A01, contigs, sat_pct=10, rep_pct=70, n_sat_fam=4, n_rep_fam=6)
    ref = oracle.synth_ref(Po, 0, Po.genome_len)
    seqs = [ref[Po.contig_start[i]:Po.contig_start[i + 1]] for i in range(len(contigs))]
    Ro = oracle.read_params(0x5C2B0A02)
    n = 12000
    plain = oracle.synth_reads(Po, Ro, 0, n).reshape(n, 150)
    indel, _ = _short_indel_reads(ref, 3000, 23)
    rng = np.random.default_rng(5)
    dup = []
    for _ in range(1500):                      # 60 bases, then the same 45 again, then on: repeated k-mers inside the read
        s = int(rng.integers(0, len(ref) - 200))
        r = bytes(ref[s:s + 60]) + bytes(ref[s + 15:s + 60]) + bytes(ref[s + 60:s + 105])
        dup.append(np.frombuffer(r, dtype=np.uint8))
    bases = np.concatenate([plain.reshape(-1), indel, np.concatenate(dup)])
    offs = np.arange(len(bases) // 150 + 1, dtype=np.uint64) * 150
    cidx = oracle.Index.build(seqs, 11, 21)
    # (a) the chain-level decision (no SH_F_CIGAR): the pair tests of DESIGN.md section 3, on and off
    g0, o0 = S.preset("sr"), oracle.preset("sr")
    g0.flags = 0; o0.flags = 0
    gidx = S.Index.build([bytes(s) for s in seqs], g0)
    of, ot = cidx.classify(o0, bases, offs, threads=8)
    f_on, _, st_on, rc = gidx.classify(bases, offs, want_trace=False)
    assert rc == 0 and np.array_equal(f_on, of), f"{int((f_on != of).sum())} flags differ with the pair test"
    print({k: st_on[k] for k in ("n_reads", "n_host", "n_no_seed", "n_chain_small", "n_chain_large", "n_anchors", "n_pair_decided")})
    assert st_on["n_chain_large"] > 1500 and st_on["n_pair_decided"] > 2000, st_on      # the shortcut did the deciding
    assert st_on["n_pair_decided"] < st_on["n_chain_large"] + st_on["n_chain_small"]      # and some reads fell through
    monkeypatch.setenv("SCRUBBY_HIP_NO_PAIR", "1")
    f_off, _, st_off, rc = gidx.classify(bases, offs, want_trace=False)
    assert np.array_equal(f_off, of) and st_off["n_pair_decided"] == 0
    monkeypatch.delenv("SCRUBBY_HIP_NO_PAIR")
    gf, gt, st, rc = gidx.classify(bases, offs, want_trace=True)                         # trace mode never takes the shortcut
    assert_trace_equal(S, gf, gt, of, ot)
    assert st["n_pair_decided"] == 0
    # (b) with the extension stage (the preset's default, as `.with_cigar()` sets it): the co-diagonal-singleton shortcut on and off,
    #     and the trace of every region
    gidx1 = S.Index.build([bytes(s) for s in seqs], S.preset("sr"))
    of1, ot1 = cidx.classify(oracle.preset("sr"), bases, offs, threads=8)
    f1, _, st1, rc = gidx1.classify(bases, offs, want_trace=False)
    assert rc == 0 and np.array_equal(f1, of1) and st1["n_pair_decided"] > 0 and st1["n_ext_reads"] > 0
    monkeypatch.setenv("SCRUBBY_HIP_NO_S1", "1")
    f2, _, st2, rc = gidx1.classify(bases, offs, want_trace=False)
    assert np.array_equal(f2, of1) and st2["n_pair_decided"] == 0 and st2["n_ext_shortcut"] > 0
    monkeypatch.delenv("SCRUBBY_HIP_NO_S1")
    gf1, gt1, st3, rc = gidx1.classify(bases, offs, want_trace=True)
    assert_trace_equal(S, gf1, gt1, of1, ot1)
    assert st3["n_ext_dropped"] == int(((ot1["n_chain"] > 0) & (ot1["n_regs"] == 0)).sum())


def test_fasta_taken_apart_on_the_gpu(S, oracle, cfg1, gpu_index, tmp_path, monkeypatch):
    """sh_index_build_fasta: raw file bytes go to HBM and the text is split there (transition-function scan + compaction,
    csrc/sh_index.hip).  Layouts that stress the line machine - 60-column lines, one line per contig, CRLF, blank lines,
    a '>' inside a header, lower case, no final newline, gzip - must give the index that the parsed sequences give, and
    the line-by-line host reader (SCRUBBY_HIP_FASTA_HOST=1) must agree."""
    import gzip
    P, R, ref, seqs, reads, off = cfg1
    want = oracle.Index.wrap(*gpu_index.export(), 11, 21).dump()
    texts = {}
    recs60 = []
    for i, s in enumerate(seqs):
        t = bytes(s).decode()
        recs60.append(f">ctg{i} len={len(t)} >not a header\n" + "\n".join(t[j:j + 60] for j in range(0, len(t), 60)) + "\n")
    texts["cols60"] = "".join(recs60)
    texts["one_line_no_final_newline"] = "".join(f">c{i}\n{bytes(s).decode()}\n" for i, s in enumerate(seqs)).rstrip("\n")
    texts["crlf_blank_lines"] = "\r\n\r\n" + "".join(f">c{i} x\r\n" + "\r\n".join(bytes(s).decode()[j:j + 70000] for j in range(0, len(s), 70000)) + "\r\n\r\n"
                                                  for i, s in enumerate(seqs))
    texts["lower_case"] = texts["cols60"].lower().replace(">ctg", ">CTG")
    for name, text in texts.items():
        fa = tmp_path / f"{name}.fa"
        fa.write_bytes(text.encode())
        for env in ("0", "1"):
            monkeypatch.setenv("SCRUBBY_HIP_FASTA_HOST", env)
            idx = S.Index.build_fasta(str(fa), S.preset("sr"))
            got = oracle.Index.wrap(*idx.export(), 11, 21).dump()
            assert all(np.array_equal(x, y) for x, y in zip(got, want)), (name, env)
            inf = idx.info()
            assert inf["n_contigs"] == len(seqs) and inf["n_bases"] == sum(len(s) for s in seqs), (name, env, inf)
    monkeypatch.setenv("SCRUBBY_HIP_FASTA_HOST", "0")
    gz = tmp_path / "ref.fa.gz"
    with gzip.open(gz, "wb") as f:
        f.write(texts["cols60"].encode())
    got = oracle.Index.wrap(*S.Index.build_fasta(str(gz), S.preset("sr")).export(), 11, 21).dump()
    assert all(np.array_equal(x, y) for x, y in zip(got, want))
    # a file that does not open with a header goes to the line reader, which skips the leading junk as before
    junk = tmp_path / "junk_first.fa"
    junk.write_bytes(("ACGTACGT\n" + texts["cols60"]).encode())
    got = oracle.Index.wrap(*S.Index.build_fasta(str(junk), S.preset("sr")).export(), 11, 21).dump()
    assert all(np.array_equal(x, y) for x, y in zip(got, want))


def test_classify_batch_from_concurrent_host_threads(S, oracle, cfg1, gpu_index, cpu_index):
    """The reference shares one aligner between rayon workers (cleaner.rs:546-552); sh_classify_batch must take concurrent
    calls on one index the same way (each call its own context, taken from / returned to the index's pool)."""
    import threading
    P, R, ref, seqs, reads, off = cfg1
    n = len(off) - 1
    parts = [(i * n // 4, (i + 1) * n // 4) for i in range(4)]
    of, _ = cpu_index.classify(oracle.preset("sr"), reads, off, threads=8, want_trace=False)
    out, errs = {}, []

    def work(i, rounds):
        try:
            a, b = parts[i]
            for _ in range(rounds):
                f, _, st, rc = gpu_index.classify(reads[int(off[a]):int(off[b])], off[a:b + 1] - off[a], want_trace=False)
                assert rc == 0 and st["n_reads"] == b - a
                out[i] = f
        except Exception as e:                     # surfaced in the main thread
            errs.append(e)

    th = [threading.Thread(target=work, args=(i, 3)) for i in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    got = np.concatenate([out[i] for i in range(4)])
    assert np.array_equal(got, of)


def test_pair_test_on_satellite_heavy_reference(S, oracle, monkeypatch):
    """Tandem arrays (171 / 68 / 5 bp monomers) fill 45 % of this reference: seeds with thousands of occurrences (the pair
    test walks one list and binary-searches the other, retries, cost cap), reads whose k-mers repeat inside the read (68-bp and
    5-bp arrays: the distinct-key premise fails) and re-chained reads (max_occ pass).  Flag-only results with and without the
    pair tests must equal the oracle's."""
    contigs = [700_000, 500_000]
    Po = oracle.ref_params(0x5C2B0B01, contigs, sat_pct=45, rep_pct=30, n_sat_fam=3, n_rep_fam=20)
    ref = oracle.synth_ref(Po, 0, Po.genome_len)
    seqs = [ref[Po.contig_start[i]:Po.contig_start[i + 1]] for i in range(len(contigs))]
    Ro = oracle.read_params(0x5C2B0B02)
    n = 6000
    bases = oracle.synth_reads(Po, Ro, 0, n)
    offs = np.arange(n + 1, dtype=np.uint64) * 150
    cidx = oracle.Index.build(seqs, 11, 21)
    g0, o0 = S.preset("sr"), oracle.preset("sr")          # chain-level decision: the pair tests
    g0.flags = 0; o0.flags = 0
    gidx = S.Index.build([bytes(s) for s in seqs], g0)
    of, ot = cidx.classify(o0, bases, offs, threads=8)
    f_on, _, st_on, rc = gidx.classify(bases, offs, want_trace=False)
    assert rc == 0 and np.array_equal(f_on, of), f"{int((f_on != of).sum())} flags differ with the pair tests"
    print({k: st_on[k] for k in ("n_reads", "n_host", "n_chain_small", "n_chain_large", "n_anchors", "n_pair_decided")})
    assert st_on["n_chain_large"] > 500 and 0 < st_on["n_pair_decided"] < st_on["n_chain_large"] + st_on["n_chain_small"]
    monkeypatch.setenv("SCRUBBY_HIP_NO_PAIR", "1")
    f_off, _, st_off, rc = gidx.classify(bases, offs, want_trace=False)
    assert np.array_equal(f_off, of) and st_off["n_pair_decided"] == 0
    monkeypatch.delenv("SCRUBBY_HIP_NO_PAIR")
    gf, gt, st, rc = gidx.classify(bases, offs, want_trace=True)
    assert_trace_equal(S, gf, gt, of, ot)
    # with the extension stage: thousands of chains per read go through mm_set_parent / mm_select_sub (best_n) before alignment
    gidx1 = S.Index.build([bytes(s) for s in seqs], S.preset("sr"))
    of1, ot1 = cidx.classify(oracle.preset("sr"), bases, offs, threads=8)
    assert int(ot1["n_chain"].max()) > 20 and int(ot1["n_aligned"].max()) >= 20
    f1, _, st1, rc = gidx1.classify(bases, offs, want_trace=False)
    assert rc == 0 and np.array_equal(f1, of1)
    gf1, gt1, _, rc = gidx1.classify(bases, offs, want_trace=True)
    assert_trace_equal(S, gf1, gt1, of1, ot1)


def test_parallel_chaining_recurrence_and_read_level_backtrack(S, oracle, monkeypatch):
    """DESIGN.md 3.3 on a reference where it can fail: tandem arrays (171 / 68 / 5 bp monomers) whose windows hold dozens to hundreds of valid
    predecessors, so that anchors break the premise of the parallel recurrence (dirty clusters: sequential DP) next to clusters that keep it.
    Flags of the flag-only call (par_fill_block + backtrack_block_top), of the same call with the read-level backtrack off (cluster path over
    the prefilled DP), with the recurrence off (the sequential DP everywhere), and the full trace must all equal the oracle's."""
    contigs = [700_000, 500_000]
    Po = oracle.ref_params(0x5C2B0C01, contigs, sat_pct=45, rep_pct=30, n_sat_fam=3, n_rep_fam=20)
    ref = oracle.synth_ref(Po, 0, Po.genome_len)
    seqs = [ref[Po.contig_start[i]:Po.contig_start[i + 1]] for i in range(len(contigs))]
    Ro = oracle.read_params(0x5C2B0C02)
    n = 8000
    bases = oracle.synth_reads(Po, Ro, 0, n)
    offs = np.arange(n + 1, dtype=np.uint64) * 150
    cidx = oracle.Index.build(seqs, 11, 21)
    gidx = S.Index.build([bytes(s) for s in seqs], S.preset("sr"))
    of, ot = cidx.classify(oracle.preset("sr"), bases, offs, threads=8)
    f1, _, st1, rc = gidx.classify(bases, offs, want_trace=False)
    assert rc == 0 and np.array_equal(f1, of), f"{int((f1 != of).sum())} flags differ"
    print({k: st1[k] for k in ("n_chain_large", "n_anchors", "n_dp_parallel", "n_dp_dirty", "n_top_settled", "n_ext_reads")})
    assert st1["n_dp_parallel"] > 300 and st1["n_top_settled"] > 100 and st1["n_dp_dirty"] > 0, st1
    monkeypatch.setenv("SCRUBBY_HIP_PFT_GMIN", "1")                          # giant reads: eight lanes per anchor in the tiled recurrence (the bench's largest reads)
    f5, _, st5, rc = gidx.classify(bases, offs, want_trace=False)
    assert rc == 0 and np.array_equal(f5, of) and st5["n_dp_parallel"] == st1["n_dp_parallel"]
    gf5, gt5, _, rc = gidx.classify(bases, offs, want_trace=True)
    assert_trace_equal(S, gf5, gt5, of, ot)
    monkeypatch.delenv("SCRUBBY_HIP_PFT_GMIN")
    monkeypatch.setenv("SCRUBBY_HIP_TOPBT_MAX", "2")                         # reads with more than two candidates at the top score: cluster by cluster
    f4, _, st4, rc = gidx.classify(bases, offs, want_trace=False)
    assert rc == 0 and np.array_equal(f4, of) and 0 < st4["n_top_settled"] < st1["n_top_settled"], st4
    monkeypatch.delenv("SCRUBBY_HIP_TOPBT_MAX")
    monkeypatch.setenv("SCRUBBY_HIP_NO_TOPBT", "1")
    f2, _, st2, rc = gidx.classify(bases, offs, want_trace=False)
    assert rc == 0 and np.array_equal(f2, of) and st2["n_top_settled"] == 0 and st2["n_dp_parallel"] > 300
    monkeypatch.delenv("SCRUBBY_HIP_NO_TOPBT")
    monkeypatch.setenv("SCRUBBY_HIP_NO_PARFILL", "1")
    f3, _, st3, rc = gidx.classify(bases, offs, want_trace=False)
    assert rc == 0 and np.array_equal(f3, of) and st3["n_dp_parallel"] == 0 and st3["n_top_settled"] == 0
    gf0, gt0, _, rc = gidx.classify(bases, offs, want_trace=True)            # the sequential DP's trace ...
    assert_trace_equal(S, gf0, gt0, of, ot)
    monkeypatch.delenv("SCRUBBY_HIP_NO_PARFILL")
    gf, gt, st, rc = gidx.classify(bases, offs, want_trace=True)             # ... and the recurrence's, cluster by cluster
    assert_trace_equal(S, gf, gt, of, ot)
    assert st["n_dp_parallel"] > 300 and st["n_top_settled"] == 0


def test_reads_beyond_the_extension_stages_working_memory_get_memory_of_their_own(S, oracle, monkeypatch):
    """minimap2 has no limit on the chains of a read; the extension stage's per-wave working memory has (16 384; 70 here, through
    SCRUBBY_HIP_EXT_REGCAP).  A read beyond it is redone with working memory allocated for it (sh_stats.n_ext_ondemand) and answered like
    every other read: flags and traces equal the oracle's, nothing is left at a chain-level answer."""
    contigs = [700_000, 500_000]
    Po = oracle.ref_params(0x5C2B0B01, contigs, sat_pct=45, rep_pct=30, n_sat_fam=3, n_rep_fam=20)
    ref = oracle.synth_ref(Po, 0, Po.genome_len)
    seqs = [ref[Po.contig_start[i]:Po.contig_start[i + 1]] for i in range(len(contigs))]
    Ro = oracle.read_params(0x5C2B0B02)
    n = 3000
    bases = oracle.synth_reads(Po, Ro, 0, n)
    offs = np.arange(n + 1, dtype=np.uint64) * 150
    cidx = oracle.Index.build(seqs, 11, 21)
    of, ot = cidx.classify(oracle.preset("sr"), bases, offs, threads=8)
    big = ot["n_chain"] > 70
    assert 0 < int(big.sum()) < n // 2
    monkeypatch.setenv("SCRUBBY_HIP_EXT_REGCAP", "70")
    gidx = S.Index.build([bytes(s) for s in seqs], S.preset("sr"))
    gf, gt, st, rc = gidx.classify(bases, offs, want_trace=True)
    assert rc == 0
    assert st["n_ext_unresolved"] == 0 and st["n_ext_ondemand"] >= int(big.sum())      # (a read may need two rounds: 70 -> 280 -> 1120)
    assert np.array_equal(gf, of)
    for name in ("n_regs", "n_aligned", "dp_max"):
        assert np.array_equal(gt[name], ot[name]), name
    assert int(ot["n_chain"].max()) > 280          # the case does hold reads that need the second round
