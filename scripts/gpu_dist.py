"""Anchor-count distribution on the bench workload (sample)."""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from scrubby_amd import lib as S
import bench as B
dev = torch.device("cuda:0")
P = S.ref_params(B.REF_SEED, B.CHM13_CONTIGS); R = S.read_params(B.READ_SEED)
G = P.genome_len; opts = S.preset("sr")
d_ref = torch.empty(G + 64, dtype=torch.uint8, device=dev)
S.synth_ref_device(P, 0, G, d_ref)
index = S.Index.build_device(d_ref, [P.contig_start[i] for i in range(len(B.CHM13_CONTIGS) + 1)], opts)
del d_ref
N = 1_000_000
d_reads = torch.empty(N * 150 + 64, dtype=torch.uint8, device=dev); d_off = torch.empty(N + 1, dtype=torch.int64, device=dev)
S.synth_reads_device(P, R, 0, N, d_reads, d_off)
fl = torch.zeros(N, dtype=torch.uint8, device=dev); tr = torch.zeros((N, len(S.TRACE_FIELDS)), dtype=torch.int32, device=dev)
ctx = S.Context(index, N, N * 150, 150)
st = ctx.classify(d_reads[:N*150], d_off, fl, tr)
print(st)
t = tr.cpu().numpy().view(S.TRACE_DTYPE).reshape(-1)
na = t["n_anchor"]; ns = t["n_seed"]
h = ns > 0
print("reads with seeds", int(h.sum()))
print("n_anchor pct", [(p, float(np.percentile(na[h], p))) for p in (10, 25, 50, 75, 90, 95, 99, 99.9, 100)])
edges = [0, 1, 16, 32, 48, 64, 96, 128, 256, 512, 1024, 2048, 4096, 8192, 32768, 1 << 30]
for a, b in zip(edges[:-1], edges[1:]):
    m = (na >= a) & (na < b) & h
    print(f"[{a},{b}) reads={int(m.sum())} anchors={int(na[m].sum())} rechained={int(t['rechained'][m].sum())} nchain_mean={t['n_chain'][m].mean() if m.any() else 0:.1f}")
print("n_seed pct", [(p, float(np.percentile(ns[h], p))) for p in (50, 90, 99, 100)])
print("rep_len>0", int((t["rep_len"] > 0).sum()), "rechained", int(t["rechained"].sum()))
