"""CPU tests of the oracle's long-read branch (oracle/mm_rmq.c, oracle/mm_align.c align1_lr; SURVEY.md App. A.5 last sentence, A.6):
the krmq tree restatement against a brute-force scan, mg_lchain_rmq's scoring pass against a plain restatement of what the two trees
are asked, the local aligner behind the inversion tests against a plain Smith-Waterman, the approximate-maximum mode of ksw_extd2
against the exact one, and the whole stage on reads built to take its branches (tests/long_cases.py)."""
import ctypes as C

import numpy as np
import pytest

from tests import long_cases as LC
from tests import workloads as W
from tests.test_align_oracle_cpu import extd2, cigar_score


def test_krmq_tree_against_brute_force(oracle):
    L = oracle.lib()
    L.mmo_rmq_selftest.argtypes = [C.c_uint64, C.c_int, C.c_int]
    for seed, (n_ops, key_range) in enumerate([(3000, 40), (20000, 1000), (30000, 100000), (4000, 3)]):
        assert L.mmo_rmq_selftest(seed, n_ops, key_range) == 0


def _sc_simple(ai, aj, pen_gap, pen_skip, log2):
    dq = int(np.int32(ai[1] & 0xffffffff)) - int(np.int32(aj[1] & 0xffffffff))
    dr = int(np.int32((ai[0] - aj[0]) & 0xffffffff))
    dd = abs(dr - dq); dg = min(dr, dq); span = (aj[1] >> 32) & 0xff
    sc = min(span, dg)
    exact = dd == 0 and dg <= span
    if dd or dq > span:
        lin = np.float32(np.float32(pen_gap) * np.float32(dd) + np.float32(pen_skip) * np.float32(dg))
        lg = log2(np.float32(dd + 1)) if dd >= 1 else np.float32(0)
        sc -= int(np.float32(lin + np.float32(0.5) * lg))
    return sc, exact, dd


def test_rmq_chaining_pass_against_a_plain_restatement(oracle):
    """What mg_lchain_rmq asks of its trees, stated without trees: the active anchor of smallest priority with y in the window, then the
    inner window in descending (y, i) order.  Random anchor sets with distinct priorities, so no tie rule is involved."""
    L = oracle.lib()
    L.mmo_lchain_rmq_fill.argtypes = [C.c_int] * 5 + [C.c_float, C.c_float, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    log2 = lambda x: np.float32(L.mmo_log2(C.c_float(float(x))))
    rng = np.random.default_rng(5)
    pen_gap, pen_skip, max_dist, inner, bw, max_skip = np.float32(0.12), np.float32(0.0), 5000, 1000, 20000, 25
    for it in range(12):
        n = int(rng.integers(20, 260))
        # a noisy diagonal plus off-diagonal clutter, one strand of one contig; x strictly increasing per group
        x = np.sort(rng.integers(1000, 1000 + 40 * n, n)).astype(np.uint64)
        y = (x.astype(np.int64) - 1000 + rng.integers(-30, 31, n) + np.where(rng.random(n) < 0.2, rng.integers(-3000, 3000, n), 0)).clip(15, None).astype(np.uint64)
        a = np.zeros((n, 2), np.uint64); a[:, 0] = x; a[:, 1] = (np.uint64(15) << np.uint64(32)) | y
        f = np.zeros(n, np.int32); p = np.zeros(n, np.int64); t = np.zeros(n, np.int32)
        L.mmo_lchain_rmq_fill(max_dist, inner, bw, max_skip, 100000, pen_gap, pen_skip, n, a.ctypes.data, f.ctypes.data, p.ctypes.data, t.ctypes.data)
        md = max(max_dist, bw); mdi = min(inner, md)
        A = [(int(a[i, 0]), int(a[i, 1])) for i in range(n)]
        F, Pp, T = [0] * n, [-1] * n, [0] * n
        i0 = st = st_in = 0
        root, inn = set(), set()
        tie = False
        for i in range(n):
            xi, yi = A[i]; qi = int(np.int32(yi & 0xffffffff))
            if i0 < i and A[i0][0] != xi:
                root |= set(range(i0, i)); inn |= set(range(i0, i)); i0 = i
            while st < i and xi > A[st][0] + md: root.discard(st); st += 1
            while st_in < i and xi > A[st_in][0] + mdi: inn.discard(st_in); st_in += 1
            max_f, max_j = 15, -1
            pri = lambda j: -(F[j] + 0.5 * float(pen_gap) * (int(np.int32(A[j][0] & 0xffffffff)) + int(np.int32(A[j][1] & 0xffffffff))))
            cand = [j for j in root if (qi - md < int(np.int32(A[j][1] & 0xffffffff)) < qi) or (int(np.int32(A[j][1] & 0xffffffff)) == qi and j == 0)]
            if cand:
                best = min(cand, key=pri)
                tie |= sum(1 for j in cand if pri(j) == pri(best)) > 1
                sc, exact, width = _sc_simple(A[i], A[best], pen_gap, pen_skip, log2); sc += F[best]
                if width <= bw and sc > max_f: max_f, max_j = sc, best
                if not exact and inn and qi > 0:
                    n_skip = 0
                    order = sorted((j for j in inn if int(np.int32(A[j][1] & 0xffffffff)) <= qi - 1), key=lambda j: (int(np.int32(A[j][1] & 0xffffffff)), j), reverse=True)
                    for j in order:
                        if int(np.int32(A[j][1] & 0xffffffff)) < qi - mdi: break
                        sc, _, width = _sc_simple(A[i], A[j], pen_gap, pen_skip, log2); sc += F[j]
                        if width <= bw:
                            if sc > max_f:
                                max_f, max_j = sc, j
                                if n_skip > 0: n_skip -= 1
                            elif T[j] == i:
                                n_skip += 1
                                if n_skip > max_skip: break
                            if Pp[j] >= 0: T[Pp[j]] = i
            F[i], Pp[i] = max_f, max_j
        if not tie:
            assert F == f.tolist() and Pp == p.tolist(), it


def _plain_local(q, t, a, b, amb, o, e):
    """Smith-Waterman with affine gaps written from the recurrence: best local score."""
    n, m = len(t), len(q)
    H = np.zeros((n + 1, m + 1), np.int64); E = np.zeros_like(H); F = np.zeros_like(H)
    best = 0
    for i in range(1, n + 1):
        for j in range(1, m + 1):
            s = -amb if (t[i - 1] > 3 or q[j - 1] > 3) else (a if t[i - 1] == q[j - 1] else -b)
            E[i, j] = max(E[i - 1, j] - e, H[i - 1, j] - o - e, 0)
            F[i, j] = max(F[i, j - 1] - e, H[i, j - 1] - o - e, 0)
            H[i, j] = max(0, H[i - 1, j - 1] + s, E[i, j], F[i, j])
            best = max(best, int(H[i, j]))
    return best


def test_local_aligner_of_the_inversion_tests(oracle):
    L = oracle.lib()
    L.mma_ksw_ll.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.mma_ksw_ll.restype = C.c_int
    rng = np.random.default_rng(9)
    mat = np.zeros(25, np.int8)
    for it in range(60):
        a, b, o, e = [(2, 4, 4, 2), (1, 4, 6, 2)][it % 2]
        L.mma_gen_simple_mat(5, mat.ctypes.data, a, b, 1)
        n, m = int(rng.integers(1, 70)), int(rng.integers(1, 60))
        t = rng.integers(0, 4, n).astype(np.uint8)
        if it % 3:
            st = int(rng.integers(0, max(1, n - 5)))
            q = t[st:st + m].copy()
            for k in range(len(q)):
                if rng.random() < 0.1: q[k] = rng.integers(0, 4)
            if it % 5 == 0 and len(q) > 6: q = np.delete(q, int(rng.integers(1, len(q) - 1)))
            q = np.concatenate([rng.integers(0, 4, 3).astype(np.uint8), q, rng.integers(0, 4, 3).astype(np.uint8)])
        else:
            q = rng.integers(0, 4, m).astype(np.uint8)
        if it % 7 == 0: q[int(rng.integers(0, len(q)))] = 4
        qe, te = C.c_int(), C.c_int()
        sc = L.mma_ksw_ll(len(q), q.ctypes.data, len(t), t.ctypes.data, mat.ctypes.data, o, e, C.byref(qe), C.byref(te))
        assert sc == _plain_local(list(q), list(t), a, b, 1, o, e), it
        if sc > 0:
            assert 0 <= te.value < len(t) and 0 <= qe.value < (len(q) + 7) // 8 * 8
            # the reported end is an end of a best local alignment: the prefix problem up to it reaches the same score
            assert _plain_local(list(q[:qe.value + 1]), list(t[:te.value + 1]), a, b, 1, o, e) == sc or qe.value >= len(q)


def test_approximate_maximum_mode_scores_the_global_alignment(oracle):
    """KSW_EZ_APPROX_MAX (first pass of the gap filling): the end-to-end score and CIGAR are those of the exact mode."""
    L = oracle.lib()
    rng = np.random.default_rng(2)
    for it in range(40):
        n = int(rng.integers(20, 260)); t = rng.integers(0, 4, n).astype(np.uint8)
        q = [c if rng.random() > 0.06 else int(rng.integers(0, 4)) for c in t]
        for _ in range(int(rng.integers(0, 6))):
            p = int(rng.integers(1, len(q) - 1))
            if rng.random() < 0.5: del q[p]
            else: q.insert(p, int(rng.integers(0, 4)))
        q = np.array(q, np.uint8)
        ez0, cig0, mat = extd2(L, q, t, 2, 4, 1, 4, 2, 24, 1, 30001, 400, -1, 0)
        ez1, cig1, _ = extd2(L, q, t, 2, 4, 1, 4, 2, 24, 1, 30001, 400, -1, 0x08)
        assert ez1.score == ez0.score and cig1 == cig0 and ez1.zdropped == 0
        assert cigar_score(cig1, q, t, mat, 4, 2, 24, 1)[0] == ez1.score


@pytest.fixture(scope="module")
def ont(oracle):
    P, R, ref, seqs, reads, off = W.cfg1(oracle, 100)
    idx = oracle.Index.build(seqs, 10, 15)
    return ref, idx, idx.update_opts(oracle.preset("map-ont"))


def test_long_branch_takes_its_branches(oracle, ont):
    ref, idx, oo = ont
    recs, bases, offs = LC.long_edge_reads(ref, 130, max_len=3500)
    o0 = idx.update_opts(oracle.preset("map-ont")); o0.flags = 0
    f1, t1 = idx.classify(oo, bases, offs, threads=8)
    f0, t0 = idx.classify(o0, bases, offs, threads=8)
    kinds = np.arange(len(recs)) % 13
    assert np.all(f1 <= f0) and np.array_equal(f1 == 1, t1["n_regs"] > 0)
    assert int(f0[kinds == 10].sum()) == 0                                        # unrelated sequence never chains
    assert np.all(f1[np.isin(kinds, (0, 2, 3, 7, 8, 9, 11))] == 1)                # reads of the reference map whatever their structure
    assert np.all(t1["dp_max"][f1 == 1] >= oo.min_dp_max)
    planted = kinds == 12
    assert int(f0[planted].sum()) >= 1 and int(f1[planted].sum()) < int(f0[planted].sum())      # chains that do not survive the alignment
    assert np.all(t1["n_regs"][kinds == 1] >= 2)                                  # chimeras: one region per locus
    inv = kinds == 4
    assert int((t1["n_regs"][inv] > t1["n_aligned"][inv]).sum()) >= 3              # inverted segments: z-drop split (+ the inversion itself)
    assert int(((t1["rechained"] & 2) != 0).sum()) >= 30                           # the RMQ long join ran
    # chimeric / deleted / inserted reads leave more than one chain before the long join and fewer regions after it
    assert np.all(t1["n_chain"][(t1["rechained"] & 2) != 0] >= 1)


def test_short_read_mode_is_untouched_by_the_long_branch(oracle):
    """sr keeps its own mm_align1 branch; the tandem flag reaches its region hash (mm_gen_regs hashes the first anchor's y)."""
    P, R, ref, seqs, reads, off = W.cfg1(oracle, 3000)
    idx = oracle.Index.build(seqs, 11, 21)
    f, t = idx.classify(oracle.preset("sr"), reads, off, threads=8)
    assert np.array_equal(f == 1, t["n_regs"] > 0) and int(((t["rechained"] & 2) != 0).sum()) == 0
