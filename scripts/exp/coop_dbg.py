import os, sys, numpy as np
for k, v in (("SCRUBBY_HIP_LEXT_A", "512"), ("SCRUBBY_HIP_COOP_MIN", "256"), ("SCRUBBY_HIP_COOP_RUN", os.environ.get("T_COOP_RUN", "64")), ("SCRUBBY_HIP_CTX_CACHE", "0")):
    os.environ[k] = v
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scrubby_amd import lib as S
from oracle import oracle
Po = oracle.ref_params(0x5C2B0010, [1_000_000] * 5)
Ro = oracle.read_params(0x5C2B0020, read_len=0, host_pct=100, sub_per_10k=200, n_read_pct=1)
cpu, offs = oracle.synth_long_reads(Po, Ro, 3, 6000)
seqs = [oracle.synth_ref(Po, Po.contig_start[i], 1_000_000) for i in range(5)]
gidx = S.Index.build([bytes(s) for s in seqs], S.preset("map-ont"))
ln = np.diff(offs.astype(np.int64))
big = np.argsort(-ln)[:40]
rd = lambda r: np.asarray(cpu[int(offs[r]):int(offs[r + 1])])
comp = np.zeros(256, np.uint8); comp[:] = np.arange(256); comp[[65, 67, 71, 84]] = [84, 71, 67, 65]
recs = [rd(r) for r in big]
for a, b in zip(big[:20], big[20:]):
    recs.append(np.concatenate([rd(a), comp[rd(b)][::-1]]))
n_take = int(os.environ.get("T_N", str(len(recs))))
recs = recs[:n_take]
bases = np.concatenate(recs).astype(np.uint8)
co = np.zeros(len(recs) + 1, np.uint64); co[1:] = np.cumsum([len(x) for x in recs])
cidx = oracle.Index.build(seqs, 10, 15)
oo = cidx.update_opts(oracle.preset("map-ont"))
gf, gt, st, rc = gidx.classify(bases, co, want_trace=True)
of, ot = cidx.classify(oo, bases, co, threads=16)
bad = [i for i in range(len(recs)) if any(gt[n][i] != ot[n][i] for n in S.TRACE_FIELDS)]
print("reads", len(recs), "differ", len(bad), bad[:20], "unresolved", st["n_ext_unresolved"], "tied", st["n_rmq_tied"])
