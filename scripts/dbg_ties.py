"""Exploration: which read sets make the long join meet tied priorities (sh_stats.n_rmq_tied)?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scrubby_amd import lib as S
from oracle import oracle as O
from tests import long_cases as LC
O.build(); O.lib()
S.require_gpu()

def run(name, seqs, bases, offs):
    gidx = S.Index.build([bytes(s) for s in seqs], S.preset("map-ont"))
    gf, _, st, rc = gidx.classify(bases, offs, want_trace=False)
    print(name, "reads", len(offs) - 1, "host", int(gf.sum()), "rechained", st["n_rmq_rechained"], "tied", st["n_rmq_tied"], "exact", st["n_rmq_exact"], "unres", st["n_ext_unresolved"], flush=True)

for seed in (5, 6):
    run(f"tandem seed {seed}", *LC.tandem_case(seed=seed, n_arrays=40, n_reads=800))
Po = O.ref_params(0x5C2B0010, [1_000_000] * 5)
Ro = O.read_params(0x5C2B0020, read_len=0, host_pct=100, sub_per_10k=200, n_read_pct=1)
cpu, offs = O.synth_long_reads(Po, Ro, 3, 20000)
seqs = [O.synth_ref(Po, Po.contig_start[i], 1_000_000) for i in range(5)]
run("bench generator, 5 Mb", seqs, cpu, offs)
