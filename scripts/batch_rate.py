#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point sh_classify_batch (what a Rust caller binds):
20 M records of the bench workload in pageable host memory -> flags in host memory."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scrubby_amd import lib as S
import bench as B

n_rec = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
P, R = S.ref_params(B.REF_SEED, B.CHM13_CONTIGS), S.read_params(B.READ_SEED)
G, L = P.genome_len, R.read_len
opts = S.preset("sr")
d_ref = torch.empty(G + 64, dtype=torch.uint8, device=dev)
S.synth_ref_device(P, 0, G, d_ref)
index = S.Index.build_device(d_ref, [P.contig_start[i] for i in range(len(B.CHM13_CONTIGS) + 1)], opts, device=0)
del d_ref
d_reads = torch.empty(n_rec * L + 64, dtype=torch.uint8, device=dev)
d_off = torch.empty(n_rec + 1, dtype=torch.int64, device=dev)
S.synth_reads_device(P, R, 0, n_rec, d_reads, d_off)
h_reads = d_reads[:n_rec * L].cpu().numpy()
h_off = np.arange(n_rec + 1, dtype=np.uint64) * L
del d_reads, d_off
torch.cuda.empty_cache()
for it in range(4):
    t0 = time.perf_counter()
    fl, _, st, rc = index.classify(h_reads, h_off, want_trace=False)
    dt = time.perf_counter() - t0
    print(f"run {it}: {dt * 1e3:8.1f} ms  {n_rec / dt / 1e6:7.1f} M reads/s  host flags {int((fl == 1).sum())}  kernel ms_total {st['ms_total']:.1f}", flush=True)
