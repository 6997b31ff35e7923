"""Worker of tests/test_long_routes_gpu.py: classifies a fixed set of long reads (the bench generator's, satellite reads included) with the
switches of the environment it was started in and prints one line: sha1 of flags and traces, and the counters that say which route ran.
(The switches are read once per process - static in the library - hence a process per route.)"""
import hashlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle  # noqa: E402
from scrubby_amd import lib as S  # noqa: E402


def main():
    S.require_gpu()
    Po = oracle.ref_params(0x5C2B0010, [1_000_000] * 5)
    Ro = oracle.read_params(0x5C2B0020, read_len=0, host_pct=100, sub_per_10k=200, n_read_pct=1)
    cpu, offs = oracle.synth_long_reads(Po, Ro, 3, 20000)
    seqs = [oracle.synth_ref(Po, Po.contig_start[i], 1_000_000) for i in range(5)]
    gidx = S.Index.build([bytes(s) for s in seqs], S.preset("map-ont"))
    gf, gt, st, rc = gidx.classify(cpu, offs, want_trace=True)
    h = hashlib.sha1()
    h.update(np.ascontiguousarray(gf).tobytes())
    for name in S.TRACE_FIELDS:
        h.update(np.ascontiguousarray(gt[name]).tobytes())
    print("ROUTE " + json.dumps({"rc": int(rc), "sha1": h.hexdigest(), "mapped": int(gf.sum()), "tied": st["n_rmq_tied"], "exact": st["n_rmq_exact"], "open": st["n_rmq_open"],
                                  "unresolved": st["n_ext_unresolved"], "rechained": st["n_rmq_rechained"]}))


if __name__ == "__main__":
    main()
