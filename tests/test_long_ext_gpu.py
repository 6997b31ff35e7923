"""GPU parity of the extension stage for the long-read presets (scrubby_amd/csrc/sh_long.h vs oracle/mm_align.c align1_lr + oracle/mm_rmq.c;
`.map_ont() / .lrhq() / .map_hifi()` + `.with_cigar()`, /root/reference/src/cleaner.rs:457-458,465,473): the chains after the RMQ long
join, the regions aligned, the regions mm_filter_regs keeps, their largest dp_max and the fingerprint of their coordinates / mlen /
blen / dp_max - bit for bit, in trace mode (every region aligned) and in flag-only mode (what the boundary returns)."""
import numpy as np
import pytest

from tests import long_cases as LC
from tests import workloads as W

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S():
    from scrubby_amd import lib
    lib.require_gpu()
    return lib


@pytest.fixture(scope="module")
def cfg1(oracle):
    return W.cfg1(oracle, 100)


def assert_same(S, gf, gt, of, ot):
    assert np.array_equal(gf, of), f"{int((gf != of).sum())} flags differ, first {np.where(gf != of)[0][:5]}"
    for name in S.TRACE_FIELDS:
        g = gt[name]
        bad = np.where(g != ot[name])[0]
        assert len(bad) == 0, f"trace.{name}: {len(bad)} differ, first read {bad[0]}: gpu={gt[name][bad[0]]} cpu={ot[name][bad[0]]}"


@pytest.mark.parametrize("preset,w,k", [("map-ont", 10, 15), ("lr:hq", 19, 19), ("map-hifi", 19, 19)])
def test_structural_reads_all_presets(S, oracle, cfg1, preset, w, k):
    P, R, ref, seqs, reads, off = cfg1
    recs, bases, offs = LC.long_edge_reads(ref, 260)
    go = S.preset(preset)
    gidx = S.Index.build([bytes(s) for s in seqs], go)
    cidx = oracle.Index.build(seqs, w, k)
    oo = cidx.update_opts(oracle.preset(preset))
    for f, _ in S.Opts._fields_:
        assert getattr(go, f) == getattr(oracle.preset(preset), f), f
    gf, gt, st, rc = gidx.classify(bases, offs, want_trace=True)
    of, ot = cidx.classify(oo, bases, offs, threads=8)
    assert rc == 0
    assert_same(S, gf, gt, of, ot)
    kinds = np.arange(len(recs)) % 13
    if preset == "map-ont":
        assert int((gt["n_regs"][kinds == 4] > gt["n_aligned"][kinds == 4]).sum()) >= 5      # inversions split regions on the device too
        o0 = cidx.update_opts(oracle.preset(preset)); o0.flags = 0
        f0, _ = cidx.classify(o0, bases, offs, threads=8)
        assert int(((f0 == 1) & (gf == 0)).sum()) >= 2                       # the stage flips flags on this set
    # flag-only call: the same flags, whatever it skips
    gf2, _, st2, rc2 = gidx.classify(bases, offs, want_trace=False)
    assert rc2 == 0 and np.array_equal(gf2, of)
    assert st2["n_host"] == int(of.sum())


def test_small_direction_buffer_sends_reads_to_the_large_pass(S, oracle, cfg1, monkeypatch):
    """A first pass with 256 KB of direction bytes per wave: the end extensions of clipped reads and the long joins no longer fit, the
    reads are finished by the large-scratch pass, the answers stay the same."""
    P, R, ref, seqs, reads, off = cfg1
    recs, bases, offs = LC.long_edge_reads(ref, 104, seed=23)
    monkeypatch.setenv("SCRUBBY_HIP_LEXT_P_KB", "256")
    go = S.preset("map-ont")
    gidx = S.Index.build([bytes(s) for s in seqs], go)
    cidx = oracle.Index.build(seqs, 10, 15)
    oo = cidx.update_opts(oracle.preset("map-ont"))
    gf, gt, st, rc = gidx.classify(bases, offs, want_trace=True)
    of, ot = cidx.classify(oo, bases, offs, threads=8)
    assert rc == 0
    assert_same(S, gf, gt, of, ot)


def test_noisy_reads_at_bench_error_rates(S, oracle, cfg1):
    """BASELINE configs[3] in miniature with the decision the reference takes: the synthetic ONT generator's reads (2 % substitutions,
    1.56 % insertions, 1.56 % deletions), map-ont, full trace."""
    Po = oracle.ref_params(0x5C2B0010, [1_000_000] * 5)
    Ro = oracle.read_params(0x5C2B0020, read_len=0, host_pct=50, sub_per_10k=200, n_read_pct=1)
    n = 400
    cpu, offs = oracle.synth_long_reads(Po, Ro, 7, n)
    seqs = [oracle.synth_ref(Po, Po.contig_start[i], 1_000_000) for i in range(5)]
    gidx = S.Index.build([bytes(s) for s in seqs], S.preset("map-ont"))
    cidx = oracle.Index.build(seqs, 10, 15)
    oo = cidx.update_opts(oracle.preset("map-ont"))
    gf, gt, st, rc = gidx.classify(cpu, offs, want_trace=True)
    of, ot = cidx.classify(oo, cpu, offs, threads=8)
    assert rc == 0
    assert_same(S, gf, gt, of, ot)
    gf2, _, _, _ = gidx.classify(cpu, offs, want_trace=False)
    assert np.array_equal(gf2, of)


def test_anchors_by_locus_flag_only(S, oracle, monkeypatch):
    """Flag-only calls chain a long read over the reference windows that can hold regs[0] only (k_lr_locus) and redo the
    reads whose answer could depend on the rest with every anchor.  Repeat-rich reference (the bench generator's satellites and
    interspersed repeats): the flags must equal the oracle's with the selection on and off, reads must have been thinned out, and some
    must have taken the second round."""
    Po = oracle.ref_params(0x5C2B0010, [1_000_000] * 5)
    Ro = oracle.read_params(0x5C2B0020, read_len=0, host_pct=60, sub_per_10k=200, n_read_pct=1)
    n = 3000
    cpu, offs = oracle.synth_long_reads(Po, Ro, 11, n)
    seqs = [oracle.synth_ref(Po, Po.contig_start[i], 1_000_000) for i in range(5)]
    gidx = S.Index.build([bytes(s) for s in seqs], S.preset("map-ont"))
    cidx = oracle.Index.build(seqs, 10, 15)
    oo = cidx.update_opts(oracle.preset("map-ont"))
    of, _ = cidx.classify(oo, cpu, offs, threads=8)
    gf, _, st, rc = gidx.classify(cpu, offs, want_trace=False)
    assert rc == 0 and np.array_equal(gf, of), f"{int((gf != of).sum())} flags differ, first {np.where(gf != of)[0][:5]}"
    assert st["n_locus_reads"] > 0 and st["n_host"] == int(of.sum())
    monkeypatch.setenv("SCRUBBY_HIP_NO_LOCUS", "1")
    gf0, _, st0, rc0 = gidx.classify(cpu, offs, want_trace=False)
    assert rc0 == 0 and np.array_equal(gf0, of) and st0["n_locus_reads"] == 0
    monkeypatch.delenv("SCRUBBY_HIP_NO_LOCUS")
    # only the largest run of windows kept: reads with chains elsewhere must be caught by the checks and redone with every anchor
    monkeypatch.setenv("SCRUBBY_HIP_LOCUS_TOP1", "1")
    gf1, _, st1, rc1 = gidx.classify(cpu, offs, want_trace=False)
    assert rc1 == 0 and np.array_equal(gf1, of), f"{int((gf1 != of).sum())} flags differ, first {np.where(gf1 != of)[0][:5]}"
    assert st1["n_locus_redone"] > 0 and st1["n_host"] == int(of.sum())
    print("locus:", st["n_locus_reads"], "redone:", st["n_locus_redone"], "top-1 only: redone", st1["n_locus_redone"])


def test_tied_priorities_in_the_long_join_take_the_literal_trees(S, oracle):
    """mg_lchain_rmq's range-minimum query now and then meets candidates of EQUAL priority (reads in tandem arrays: lattice points on one
    anti-diagonal reached by mirror-image gaps).  krmq_rmq resolves such a tie by the shape of its tree; the wave scan cannot, so these reads
    are redone on the literal trees (sh_rmq_tree.h).  20 000 reads of the bench's generator (satellite arrays included): full trace and
    flags against the oracle, which restates the trees (oracle/mm_rmq.c) - and ties must actually have been met."""
    Po = oracle.ref_params(0x5C2B0010, [1_000_000] * 5)
    Ro = oracle.read_params(0x5C2B0020, read_len=0, host_pct=100, sub_per_10k=200, n_read_pct=1)
    cpu, offs = oracle.synth_long_reads(Po, Ro, 3, 20000)
    seqs = [oracle.synth_ref(Po, Po.contig_start[i], 1_000_000) for i in range(5)]
    gidx = S.Index.build([bytes(s) for s in seqs], S.preset("map-ont"))
    cidx = oracle.Index.build(seqs, 10, 15)
    oo = cidx.update_opts(oracle.preset("map-ont"))
    gf, gt, st, rc = gidx.classify(cpu, offs, want_trace=True)
    of, ot = cidx.classify(oo, cpu, offs, threads=16)
    assert rc == 0
    assert_same(S, gf, gt, of, ot)
    print("tied:", st["n_rmq_tied"], "exact:", st["n_rmq_exact"], "rechained:", st["n_rmq_rechained"])
    assert st["n_rmq_exact"] >= st["n_rmq_tied"] > 0 and st["n_ext_unresolved"] == 0
    gf2, _, st2, rc2 = gidx.classify(cpu, offs, want_trace=False)
    assert rc2 == 0 and np.array_equal(gf2, of) and st2["n_ext_unresolved"] == 0 and st2["n_rmq_tied"] > 0


def test_lattices_of_anchors_across_perfect_tandem_arrays(S, oracle):
    """Reads across perfect tandem arrays with another copy number than the reference: every pair of copies anchors, the inner window of the
    long join holds several anchors per reference position and outgrows the small LDS ring - the large ring, and beyond it the trees, must
    give the oracle's chains."""
    seqs, bases, offs = LC.tandem_case()
    gidx = S.Index.build([bytes(s) for s in seqs], S.preset("map-ont"))
    cidx = oracle.Index.build([np.frombuffer(s, np.uint8) for s in seqs], 10, 15)
    oo = cidx.update_opts(oracle.preset("map-ont"))
    gf, gt, st, rc = gidx.classify(bases, offs, want_trace=True)
    of, ot = cidx.classify(oo, bases, offs, threads=8)
    assert rc == 0 and st["n_ext_unresolved"] == 0
    assert_same(S, gf, gt, of, ot)
    gf2, _, st2, rc2 = gidx.classify(bases, offs, want_trace=False)
    assert rc2 == 0 and np.array_equal(gf2, of) and st2["n_ext_unresolved"] == 0


def test_ties_after_the_look_back_window_has_emptied(S, oracle):
    """Chimeras of the reads whose long join meets tied priorities (found among 20 000 reads of the bench's generator: sh_ctx_debug_list 3),
    two or three of them end to end, some reverse-complemented: the loci lie megabases apart or on other contigs, so between them the
    join's look-back window - and upstream's tree - is empty, and a tie in a later locus is answered by a tree that only knows that locus.
    lr_rmq_fill keeps the tree only over the stretches that ask it, rebuilt by replay from where the window was last empty; the oracle's
    tree lives through the whole read.  Full trace and flags; and ties must have been met again."""
    import torch
    Po = oracle.ref_params(0x5C2B0010, [1_000_000] * 5)
    Ro = oracle.read_params(0x5C2B0020, read_len=0, host_pct=100, sub_per_10k=200, n_read_pct=1)
    cpu, offs = oracle.synth_long_reads(Po, Ro, 3, 20000)
    seqs = [oracle.synth_ref(Po, Po.contig_start[i], 1_000_000) for i in range(5)]
    gidx = S.Index.build([bytes(s) for s in seqs], S.preset("map-ont"))
    ctx = S.Context(gidx, len(offs) - 1, int(offs[-1]), int(np.diff(offs.astype(np.int64)).max()))
    d_b = torch.from_numpy(np.ascontiguousarray(cpu)).cuda(); d_o = torch.from_numpy(offs.astype(np.int64)).cuda()
    d_f = torch.zeros(len(offs) - 1, dtype=torch.uint8, device="cuda")
    ctx.classify(d_b, d_o, d_f)
    torch.cuda.synchronize()
    tied = np.unique(ctx.debug_list(3).astype(np.int64))
    assert len(tied) >= 2, "the generator's satellite reads no longer meet tied priorities"
    tied = tied[:6]
    rng = np.random.default_rng(3)
    comp = np.zeros(256, np.uint8); comp[:] = np.arange(256); comp[[65, 67, 71, 84]] = [84, 71, 67, 65]
    rd = lambda r: np.asarray(cpu[int(offs[r]):int(offs[r + 1])])
    rcm = lambda x: comp[x][::-1]
    recs = []
    for a in tied:
        for b in tied:
            if a == b:
                continue
            recs.append(np.concatenate([rd(a), rd(b)]))
            recs.append(np.concatenate([rcm(rd(b)), rd(a)]))
    for _ in range(6):
        t3 = rng.choice(tied, min(3, len(tied)), replace=False)
        recs.append(np.concatenate([rd(t) if rng.integers(0, 2) else rcm(rd(t)) for t in t3]))
    bases = np.concatenate(recs).astype(np.uint8)
    co = np.zeros(len(recs) + 1, np.uint64); co[1:] = np.cumsum([len(x) for x in recs])
    cidx = oracle.Index.build(seqs, 10, 15)
    oo = cidx.update_opts(oracle.preset("map-ont"))
    gf, gt, st, rc = gidx.classify(bases, co, want_trace=True)
    of, ot = cidx.classify(oo, bases, co, threads=16)
    assert rc == 0 and st["n_ext_unresolved"] == 0
    assert_same(S, gf, gt, of, ot)
    print("chimeras:", len(recs), "tied:", st["n_rmq_tied"], "exact:", st["n_rmq_exact"], "rechained:", st["n_rmq_rechained"])
    assert st["n_rmq_exact"] >= st["n_rmq_tied"] > 0
    gf2, _, st2, rc2 = gidx.classify(bases, co, want_trace=False)
    assert rc2 == 0 and np.array_equal(gf2, of) and st2["n_ext_unresolved"] == 0 and st2["n_rmq_tied"] > 0


def test_the_long_join_of_one_read_shared_among_waves(S, oracle, monkeypatch, capfd):
    """lr_coop_fill: the x-sorted anchors of a long join fall into stretches between which the look-back window is empty - independent
    problems - and the waves of the giants' launch take runs of them off a queue.  With the first working-memory size made tiny nearly every
    read is a giant, and with the thresholds at 256 / 64 anchors (the defaults: 12 288 / 3 072) every join of a few hundred anchors is cut
    up: the 40 longest of 6 000 reads of the bench's generator and 20 chimeras of pairs of them (loci megabases apart), full trace and flags
    against the oracle, which joins each read in one piece; the stage's debug line must say that joins were shared, in more runs than reads."""
    for k, v in (("SCRUBBY_HIP_LEXT_A", "512"), ("SCRUBBY_HIP_COOP_MIN", "256"), ("SCRUBBY_HIP_COOP_RUN", "64"), ("SCRUBBY_HIP_CTX_CACHE", "0"), ("SCRUBBY_HIP_DBG", "16")):
        monkeypatch.setenv(k, v)
    Po = oracle.ref_params(0x5C2B0010, [1_000_000] * 5)
    Ro = oracle.read_params(0x5C2B0020, read_len=0, host_pct=100, sub_per_10k=200, n_read_pct=1)
    cpu, offs = oracle.synth_long_reads(Po, Ro, 3, 6000)
    seqs = [oracle.synth_ref(Po, Po.contig_start[i], 1_000_000) for i in range(5)]
    gidx = S.Index.build([bytes(s) for s in seqs], S.preset("map-ont"))
    ln = np.diff(offs.astype(np.int64))
    big = np.argsort(-ln)[:40]                                               # the longest reads, and chimeras of pairs of them
    rd = lambda r: np.asarray(cpu[int(offs[r]):int(offs[r + 1])])
    comp = np.zeros(256, np.uint8); comp[:] = np.arange(256); comp[[65, 67, 71, 84]] = [84, 71, 67, 65]
    recs = [rd(r) for r in big]
    for a, b in zip(big[:20], big[20:]):
        recs.append(np.concatenate([rd(a), comp[rd(b)][::-1]]))
    bases = np.concatenate(recs).astype(np.uint8)
    co = np.zeros(len(recs) + 1, np.uint64); co[1:] = np.cumsum([len(x) for x in recs])
    cidx = oracle.Index.build(seqs, 10, 15)
    oo = cidx.update_opts(oracle.preset("map-ont"))
    capfd.readouterr()
    gf, gt, st, rc = gidx.classify(bases, co, want_trace=True)
    err = capfd.readouterr().err
    of, ot = cidx.classify(oo, bases, co, threads=16)
    assert rc == 0 and st["n_ext_unresolved"] == 0
    assert_same(S, gf, gt, of, ot)
    shared = [ln_ for ln_ in err.splitlines() if "long join shared" in ln_]
    print(shared[-1] if shared else "no line")
    n_reads, n_runs = (int(x) for x in __import__("re").search(r"(\d+) reads in (\d+) runs", shared[0]).groups())
    assert n_reads > 10 and n_runs > n_reads, shared
    gf2, _, st2, rc2 = gidx.classify(bases, co, want_trace=False)
    assert rc2 == 0 and np.array_equal(gf2, of)


def test_windows_beyond_the_large_ring_are_counted_or_take_the_one_lane_trees(S, oracle, monkeypatch):
    """A read whose long join holds more anchors within rmq_inner_dist than the 4096-anchor ring (LC.dense_lattice_case: ~10^4 lattice anchors
    over one kilobase) cannot be chained by the wave scan.  By default it keeps its chain-level answer and is COUNTED (n_ext_unresolved, and
    listed by sh_ctx_debug_list(5) for the bench's strata); with SCRUBBY_HIP_RMQ_ONE_LANE=1 the literal one-lane trees over node pools in HBM
    chain it (E3 of the exact passes): full trace and flags equal the oracle's, nothing unresolved."""
    seqs, bases, offs = LC.dense_lattice_case()
    monkeypatch.setenv("SCRUBBY_HIP_CTX_CACHE", "0")
    cidx = oracle.Index.build([np.frombuffer(s, np.uint8) for s in seqs], 10, 15)
    oo = cidx.update_opts(oracle.preset("map-ont"))
    of, ot = cidx.classify(oo, bases, offs, threads=8)
    gidx = S.Index.build([bytes(s) for s in seqs], S.preset("map-ont"))
    gf0, _, st0, rc0 = gidx.classify(bases, offs, want_trace=False)
    assert rc0 == 0 and st0["n_ext_unresolved"] > 0, st0
    assert np.array_equal(gf0, of)      # (the chain-level answer happens to be the oracle's on these reads: they are host reads)
    monkeypatch.setenv("SCRUBBY_HIP_RMQ_ONE_LANE", "1")
    gidx1 = S.Index.build([bytes(s) for s in seqs], S.preset("map-ont"))
    gf, gt, st, rc = gidx1.classify(bases, offs, want_trace=True)
    assert rc == 0 and st["n_ext_unresolved"] == 0 and st["n_rmq_exact"] > 0, st
    assert_same(S, gf, gt, of, ot)
    gf2, _, st2, rc2 = gidx1.classify(bases, offs, want_trace=False)
    assert rc2 == 0 and np.array_equal(gf2, of) and st2["n_ext_unresolved"] == 0
    print("unresolved by default:", st0["n_ext_unresolved"], "one-lane trees:", st["n_rmq_exact"])


def test_reads_beyond_every_prepared_size_get_memory_of_their_own(S, oracle, cfg1, monkeypatch):
    """minimap2 has no capacities.  With both prepared sizes of the stage's working memory made tiny (64 / 128 chain anchors, 64 / 128 KB of
    direction bytes) most reads outgrow them in both kernels; memory is then allocated for them, four times the last size per round, until
    they fit.  Same traces and flags as the oracle, nothing left at a chain-level answer."""
    P, R, ref, seqs, reads, off = cfg1
    recs, bases, offs = LC.long_edge_reads(ref, 104, seed=31)
    for k, v in (("SCRUBBY_HIP_LEXT_A", "64"), ("SCRUBBY_HIP_LEXT_BIG_A", "128"), ("SCRUBBY_HIP_LEXT_P_KB", "64"), ("SCRUBBY_HIP_LEXT_BIG_P_KB", "128"), ("SCRUBBY_HIP_CTX_CACHE", "0")):
        monkeypatch.setenv(k, v)
    gidx = S.Index.build([bytes(s) for s in seqs], S.preset("map-ont"))
    cidx = oracle.Index.build(seqs, 10, 15)
    oo = cidx.update_opts(oracle.preset("map-ont"))
    gf, gt, st, rc = gidx.classify(bases, offs, want_trace=True)
    of, ot = cidx.classify(oo, bases, offs, threads=8)
    assert rc == 0 and st["n_ext_unresolved"] == 0 and st["n_ext_ondemand"] > 20
    assert_same(S, gf, gt, of, ot)
    gf2, _, st2, rc2 = gidx.classify(bases, offs, want_trace=False)
    assert rc2 == 0 and np.array_equal(gf2, of) and st2["n_ext_unresolved"] == 0
    print("on demand:", st["n_ext_ondemand"], st2["n_ext_ondemand"])


@pytest.mark.parametrize("lds", [0, 1, 2])
def test_the_long_joins_tree_on_the_device_answers_like_the_oracles(S, oracle, lds):
    """sh_rmq_tree.h executed by the GPU, one lane, on both storages - nodes in an HBM pool (lds = 0), the whole tree in LDS through
    address-space-3 pointers with 16-bit links (lds = 1), and that tree with the insertions and erasures done by the whole wave (lds = 2:
    rq_insert_w / rq_erase_w, what lr_rmq_fill<NR, true> runs): the random insert / erase / query sequence
    with heavily tied priorities of oracle/mm_rmq.c's mmo_rmq_trace must be answered element for element like the oracle's tree."""
    import ctypes as C
    Lo = oracle.lib()
    Lo.mmo_rmq_trace.restype = C.c_int64
    Lo.mmo_rmq_trace.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L = S.load()
    n_ops = 9000 if lds else 30000      # (the LDS tree of the test kernel holds 4096 nodes; ~0.3 n_ops are alive at the end)
    for seed, key_range, fifo in ((1, 40, 1), (2, 5000, 0), (3, 300, 1), (4, 100000, 1)):
        a = np.full(n_ops, -7, np.int64)
        na = Lo.mmo_rmq_trace(seed, n_ops, key_range, fifo, a.ctypes.data)
        b = np.full(n_ops, -9, np.int64)
        nb = C.c_int64(0)
        S.check(L.sh_dbg_rmq_trace(0, seed, n_ops, key_range, fifo, lds, b.ctypes.data, C.byref(nb)))
        assert nb.value == na, f"device guard / count: {nb.value} vs {na}"
        assert np.array_equal(a[:na], b[:na]), f"seed {seed} lds {lds}: first difference at query {int(np.where(a[:na] != b[:na])[0][0])}"
