import ctypes as C, numpy as np, sys
sys.path.insert(0, "/root/repo")
from scrubby_amd import lib as S
L = S.load()
for lds in (0, 1, 2):
    for n_ops, kr in ((9000, 5000), (9000, 100000)):
        b = np.zeros(n_ops, np.int64); nb = C.c_int64(0)
        S.check(L.sh_dbg_rmq_trace(0, 3, n_ops, kr, 1, lds, b.ctypes.data, C.byref(nb)))
        print("lds", lds, "n_ops", n_ops, "key_range", kr, "queries", nb.value, "ticks", b[-1], "us/op", b[-1] / 100.0 / n_ops)
