"""GPU parity of the extension stage (`.with_cigar()` at /root/reference/src/cleaner.rs:473): regions aligned,
regions surviving mm_filter_regs, their largest dp_max and a fingerprint of their coordinates / mlen / blen / dp_max, bit for bit
against oracle/mm_align.c, in trace mode (every region aligned) and in flag-only mode (shortcuts on)."""
import numpy as np
import pytest

from tests import align_cases as AC
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
    return S.Index.build([bytes(s) for s in cfg1[3]], S.preset("sr"))


@pytest.fixture(scope="module")
def cpu_index(oracle, cfg1):
    return oracle.Index.build(cfg1[3], 11, 21)


def assert_same(S, gf, gt, of, ot):
    assert np.array_equal(gf, of), f"{int((gf != of).sum())} flags differ, first {np.where(gf != of)[0][:5]}"
    for name in S.TRACE_FIELDS:
        bad = np.where(gt[name] != ot[name])[0]
        assert len(bad) == 0, f"trace.{name}: {len(bad)} differ, first read {bad[0]}: gpu={gt[name][bad[0]]} cpu={ot[name][bad[0]]}"


def test_reference_bases_in_hbm_equal_the_oracles(S, gpu_index, cpu_index):
    gp, gs = gpu_index.export_ref()
    cp, cs = cpu_index.ref()
    assert np.array_equal(gs, cs) and np.array_equal(gp[:len(cp)], cp)


def test_presets_carry_the_alignment_scores(S, oracle):
    g, o = S.preset("sr"), oracle.preset("sr")
    assert (g.flags & S.SH_F_CIGAR) and (g.a, g.b, g.q, g.e, g.q2, g.e2, g.zdrop, g.end_bonus, g.min_dp_max, g.best_n) == (2, 8, 12, 2, 24, 1, 100, 10, 40, 20)
    for f, _ in S.Opts._fields_:
        assert getattr(g, f) == getattr(o, f), f


def test_cfg1_trace_with_extension_stage(S, oracle, cfg1, gpu_index, cpu_index):
    P, R, ref, seqs, reads, off = cfg1
    gf, gt, st, rc = gpu_index.classify(reads, off, want_trace=True)
    of, ot = cpu_index.classify(oracle.preset("sr"), reads, off, threads=8)
    assert rc == 0
    assert_same(S, gf, gt, of, ot)
    assert int((ot["n_regs"] > 0).sum()) == int(of.sum()) and st["n_ext_reads"] == int((ot["n_chain"] > 0).sum())


def test_edge_reads_trace_and_flag_only(S, oracle, cfg1, gpu_index, cpu_index):
    ref = cfg1[2]
    recs, bases, offs = AC.edge_reads(ref, 4000)
    o = oracle.preset("sr")
    of, ot = cpu_index.classify(o, bases, offs, threads=8)
    o0 = oracle.preset("sr"); o0.flags = 0
    of0, _ = cpu_index.classify(o0, bases, offs, threads=8)
    assert int((of != of0).sum()) >= 10, "the cases are meant to contain chains that the extension stage drops"
    assert int((ot["n_regs"] > ot["n_aligned"]).sum()) >= 10, "... and z-drop splits"
    gf, gt, st, rc = gpu_index.classify(bases, offs, want_trace=True)
    assert_same(S, gf, gt, of, ot)
    assert st["n_ext_dropped"] == int(((ot["n_chain"] > 0) & (ot["n_regs"] == 0)).sum())
    gf2, _, st2, _ = gpu_index.classify(bases, offs, want_trace=False)        # flag-only: shortcuts + early exit
    assert np.array_equal(gf2, of)
    # without the stage the library answers at chain level, as before
    g0 = S.preset("sr"); g0.flags = 0
    idx0 = S.Index.build([bytes(s) for s in cfg1[3]], g0)
    gf0, _, _, _ = idx0.classify(bases, offs, want_trace=False)
    assert np.array_equal(gf0, of0)


def test_flag_only_shortcut_decides_most_clean_reads(S, oracle, cfg1, gpu_index, cpu_index):
    P, R, ref, seqs, reads, off = cfg1
    gf, _, st, _ = gpu_index.classify(reads, off, want_trace=False)
    of, _ = cpu_index.classify(oracle.preset("sr"), reads, off, threads=8, want_trace=False)
    assert np.array_equal(gf, of)
    assert st["n_pair_decided"] > 0.5 * int(of.sum()) and st["n_ext_reads"] < 0.5 * int(of.sum())
