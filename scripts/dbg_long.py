import sys; sys.path.insert(0, "/root/repo")
import numpy as np, torch
from oracle import oracle as O
from scrubby_amd import lib as S
from tests import workloads as W
Po = O.ref_params(W.CFG1_REF_SEED, W.CFG1_CONTIGS); Ro = O.read_params(0x5C2B0020, host_pct=50, sub_per_10k=500, n_read_pct=0)
n = 2000
cpu, offs = O.synth_long_reads(Po, Ro, 7, n)
seqs = [O.synth_ref(Po, Po.contig_start[i], 1_000_000) for i in range(5)]
gidx = S.Index.build([bytes(s) for s in seqs], S.preset("map-ont"))
print(gidx.info())
gf, gt, st, rc = gidx.classify(cpu, offs, want_trace=True)
print(st)
cidx = O.Index.build(seqs, 10, 15); oo = cidx.update_opts(O.preset("map-ont"))
of, ot = cidx.classify(oo, cpu, offs, threads=8)
print("flags equal", np.array_equal(gf, of), "trace equal", all(np.array_equal(gt[k], ot[k]) for k in S.TRACE_FIELDS))
import os
bad = np.nonzero(gf != of)[0]
print("flag diffs", len(bad), bad[:10])
lens = np.diff(offs.astype(np.int64))
for k in S.TRACE_FIELDS:
    d = np.nonzero(gt[k] != ot[k])[0]
    print(k, len(d), [(int(i), int(lens[i]), int(gt[k][i]), int(ot[k][i])) for i in d[:8]])
sys.stdout.flush(); os._exit(0)
