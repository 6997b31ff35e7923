"""The wave primitives of csrc/sh_wave.h (scans, reductions and broadcasts on DPP / v_readlane) against plain arithmetic."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_wave_primitives(seed):
    from scrubby_amd import lib as S
    L = S.require_gpu()
    rng = np.random.default_rng(seed)
    v = rng.integers(-2**31, 2**31 - 1, 64, dtype=np.int64).astype(np.int32)
    if seed == 2:
        v[:] = rng.integers(-5, 5, 64)          # ties and small sums
    if seed == 3:
        v[10] = np.int32(-2**31); v[50] = np.int32(2**31 - 1)
    w = rng.integers(0, 2**64 - 1, 64, dtype=np.uint64)
    bl = int(rng.integers(0, 64))
    o32 = np.zeros((12, 64), dtype=np.int32)
    o64 = np.zeros((9, 64), dtype=np.uint64)
    S.check(L.sh_dbg_wave_ops(0, v.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p), bl, o32.ctypes.data_as(C.c_void_p), o64.ctypes.data_as(C.c_void_p)))
    vu = v.view(np.uint32)
    want32 = [np.maximum.accumulate(v), np.minimum.accumulate(v), np.cumsum(v.astype(np.int64)).astype(np.int32), np.bitwise_or.accumulate(v),
              np.concatenate([[np.int32(-7)], v[:-1]]), np.full(64, v.max()), np.full(64, v.min()), None,
              np.full(64, np.bitwise_or.reduce(v)), np.full(64, vu.max()).astype(np.uint32).view(np.int32), np.full(64, vu.min()).astype(np.uint32).view(np.int32), np.full(64, v[bl])]
    want32[7] = np.full(64, np.array(v.astype(np.int64).sum() & 0xffffffff, dtype=np.uint64).astype(np.uint32).view(np.int32))
    for i, x in enumerate(want32):
        assert np.array_equal(o32[i], np.asarray(x, dtype=np.int32)), f"int32 primitive {i}"
    ws = w.view(np.int64)
    acc = []
    t = 0
    for x in w.tolist():
        t = (t + x) & (2**64 - 1); acc.append(t)
    want64 = [np.maximum.accumulate(ws).view(np.uint64), np.maximum.accumulate(w), np.minimum.accumulate(w), np.array(acc, dtype=np.uint64),
              np.full(64, ws.max()).view(np.uint64), np.full(64, w.max()), np.full(64, w.min()), np.full(64, w[bl]), np.concatenate([[np.uint64(99)], w[:-1]])]
    for i, x in enumerate(want64):
        assert np.array_equal(o64[i], np.asarray(x, dtype=np.uint64)), f"uint64 primitive {i}"
