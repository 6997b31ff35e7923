"""debug: long reads, map-ont, extension stage on: GPU trace vs oracle trace"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
from scrubby_amd import lib as S
from tests import workloads as W
from tests.test_parity_gpu import _ont_like_reads
P, R, ref, seqs, reads, off = W.cfg1(O, 100)
go = S.preset("map-ont")
gidx = S.Index.build([bytes(s) for s in seqs], go)
cidx = O.Index.build(seqs, 10, 15)
oo = cidx.update_opts(O.preset("map-ont"))
n = int(os.environ.get("N", "60"))
recs, bases, offs = _ont_like_reads(ref, n, 42)
gf, gt, st, rc = gidx.classify(bases, offs, want_trace=True)
of, ot = cidx.classify(oo, bases, offs, threads=8)
print("rc", rc, "stats", st)
bad = 0
for i in range(n):
    same = all(gt[nm][i] == ot[nm][i] for nm in S.TRACE_FIELDS)
    if not same:
        bad += 1
        if bad <= 12:
            print(i, len(recs[i]), "GPU", [int(gt[nm][i]) for nm in S.TRACE_FIELDS]); print("      CPU", [int(ot[nm][i]) for nm in S.TRACE_FIELDS])
print("reads differing:", bad, "of", n, "flags differing", int((gf != of).sum()))
gf2, _, st2, rc2 = gidx.classify(bases, offs, want_trace=False)
print("flag-only differing from oracle:", int((gf2 != of).sum()), st2)
