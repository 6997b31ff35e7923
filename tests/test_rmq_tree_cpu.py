"""The product's restatement of minimap2's krmq tree (scrubby_amd/csrc/sh_rmq_tree.h: what answers the tied range-minimum queries of the
long join, /root/reference/src/cleaner.rs:552 with the long-read presets) against the oracle's (oracle/mm_rmq.c), both compiled for the
host: the same random insert / erase / query sequences with HEAVILY tied priorities must return the same elements - which of several equal
minima comes back is decided by the tree's shape and the subtree-minimum pointers its rotations carried over, the very thing the device
code exists to reproduce.  On both storages: the 32-byte node pool (HBM on the device) and the structure-of-arrays form with 16-bit links
(LDS on the device)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host_tree(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("rq") / "librq_host.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, os.path.join(ROOT, "tests", "rmq_tree_host.cpp")])
    L = C.CDLL(so)
    L.rqh_trace.restype = C.c_int64
    L.rqh_trace.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    return L


@pytest.mark.parametrize("fifo", [0, 1])
@pytest.mark.parametrize("key_range", [40, 5000])
def test_device_tree_returns_the_oracles_element_among_equal_minima(oracle, host_tree, fifo, key_range):
    Lo = oracle.lib()
    Lo.mmo_rmq_trace.restype = C.c_int64
    Lo.mmo_rmq_trace.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_void_p]
    n_ops = 40000
    for seed in (1, 2, 3):
        a = np.full(n_ops, -7, np.int64)
        na = Lo.mmo_rmq_trace(seed, n_ops, key_range, fifo, a.ctypes.data)
        assert na > n_ops // 5
        for cache in (0, 1):
            b = np.full(n_ops, -9, np.int64)
            nb = host_tree.rqh_trace(seed, n_ops, key_range, fifo, cache, b.ctypes.data)
            assert nb == na, f"guard {nb}" if nb < 0 else "different number of queries"
            assert np.array_equal(a[:na], b[:nb]), f"seed {seed} lds {cache}: first difference at query {int(np.where(a[:na] != b[:nb])[0][0])}"
        assert int((a[:na] >= 0).sum()) > na // 2


def _lattice(rng, n_per, period, copies_ref, copies_read, flank=60):
    """anchors of a read across a perfect tandem array: every (reference copy, read copy) pair anchors at the array's minimizer offsets"""
    offs = np.sort(rng.choice(period, size=max(2, period // 6), replace=False))
    xs, ys = [], []
    x0, y0 = 1_000_000, 500
    for k in range(flank):                                        # unique flank before the array
        xs.append(x0 - 7 * (flank - k)); ys.append(y0 - 7 * (flank - k))
    for m in range(copies_ref):
        for n in range(copies_read):
            for o in offs:
                xs.append(x0 + m * period + int(o)); ys.append(y0 + n * period + int(o))
    xe, ye = x0 + copies_ref * period, y0 + copies_read * period
    for k in range(flank):
        xs.append(xe + 7 * k); ys.append(ye + 7 * k)
    a = np.array(sorted(zip(xs, ys)), dtype=np.uint64)
    a[:, 1] |= np.uint64(15) << np.uint64(32)
    return np.ascontiguousarray(a)


@pytest.mark.parametrize("cache", [0, 1])
def test_scoring_pass_on_the_device_trees_equals_the_oracles(oracle, host_tree, cache):
    """mg_lchain_rmq's scoring pass over lattices of anchors (perfect tandem arrays: many anchors per reference position and per query
    position, ties everywhere) on the product's trees, against the oracle's mmo_lchain_rmq_fill: f and p equal, no guard of the device
    code tripped - with the inner window's anchors leaving and entering out of order and thousands of elements per tree."""
    Lo = oracle.lib()
    Lo.mmo_lchain_rmq_fill.argtypes = [C.c_int] * 5 + [C.c_float, C.c_float, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    host_tree.rqh_lchain_fill.argtypes = [C.c_int] * 5 + [C.c_float, C.c_float, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    rng = np.random.default_rng(7)
    for period, cr, cq in ((171, 14, 10), (68, 30, 22), (5, 70, 50), (300, 6, 5)):
        a = _lattice(rng, 0, period, cr, cq)
        n = len(a)
        f0 = np.zeros(n, np.int32); p0 = np.zeros(n, np.int64); t0 = np.zeros(n, np.int32)
        Lo.mmo_lchain_rmq_fill(5000, 1000, 20000, 25, 100000, 0.12, 0.0, n, a.ctypes.data, f0.ctypes.data, p0.ctypes.data, t0.ctypes.data)
        f1 = np.zeros(n, np.int32); p1 = np.zeros(n, np.int32); t1 = np.zeros(n, np.int32)
        rc = host_tree.rqh_lchain_fill(5000, 1000, 20000, 25, 100000, 0.12, 0.0, n, a.ctypes.data, f1.ctypes.data, p1.ctypes.data, t1.ctypes.data, cache)
        assert rc == 0, f"period {period}: guard {rc} tripped with {n} anchors"
        assert np.array_equal(f0, f1) and np.array_equal(p0.astype(np.int32), p1), f"period {period}: {int((f0 != f1).sum())} scores differ"
