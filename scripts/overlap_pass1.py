#!/usr/bin/env python3
"""Feasibility: what the headline would gain if the reads that are certain to take mm_map_frag's second chaining pass (no anchor at mid_occ:
satellite reads, 0.13 % of the records, ~60 ms of a 205-ms step) were classified by a second context on a second stream and host thread
BESIDE the rest, instead of serially behind it.  Times: everything in one call; the rest alone; the satellite reads alone; both side by side."""
import os, sys, threading, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from scrubby_amd import lib as S

dev = torch.device("cuda", 0)
S.require_gpu()
P = S.ref_params(B.REF_SEED, B.CHM13_CONTIGS); R = S.read_params(B.READ_SEED)
G = P.genome_len; L = R.read_len; n = int(os.environ.get("OVERLAP_N", "20000000"))
opts = S.preset("sr")
d_ref = torch.empty(G + 64, dtype=torch.uint8, device=dev); S.synth_ref_device(P, 0, G, d_ref)
index = S.Index.build_device(d_ref, [P.contig_start[i] for i in range(len(B.CHM13_CONTIGS) + 1)], opts, device=0); del d_ref
d_reads = torch.empty(n * L + 64, dtype=torch.uint8, device=dev); d_off = torch.empty(n + 1, dtype=torch.int64, device=dev)
S.synth_reads_device(P, R, 0, n, d_reads, d_off)
ctx = S.Context(index, n, n * L, L)
d_flags = torch.zeros(n, dtype=torch.uint8, device=dev)

def run(c, reads, off, flags, reps=5, stream=None):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if stream is not None:
            with torch.cuda.stream(stream):
                c.classify(reads, off, flags, None, want_stats=False)
            stream.synchronize()
        else:
            c.classify(reads, off, flags, None, want_stats=False)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts)

t_all = run(ctx, d_reads[:n * L], d_off, d_flags)
sat = np.unique(ctx.debug_list(0).astype(np.int64))
print("all in one call: %.1f ms; re-chained reads: %d" % (t_all, len(sat)), flush=True)
mask = torch.ones(n, dtype=torch.bool, device=dev); mask[torch.from_numpy(sat).to(dev)] = False
rows = d_reads[:n * L].view(n, L)
rest = rows[mask].contiguous().view(-1); n_rest = int(mask.sum().item())
sub = rows[~mask].contiguous().view(-1); n_sub = n - n_rest
rest = torch.cat([rest, torch.zeros(64, dtype=torch.uint8, device=dev)]); sub = torch.cat([sub, torch.zeros(64, dtype=torch.uint8, device=dev)])
off_rest = torch.arange(n_rest + 1, dtype=torch.int64, device=dev) * L; off_sub = torch.arange(n_sub + 1, dtype=torch.int64, device=dev) * L
f_rest = torch.zeros(n_rest, dtype=torch.uint8, device=dev); f_sub = torch.zeros(n_sub, dtype=torch.uint8, device=dev)
del ctx; torch.cuda.empty_cache()
os.environ["SCRUBBY_HIP_ARENA_MB"] = "49152"
c_rest = S.Context(index, n_rest, n_rest * L, L)
c_sub = S.Context(index, n_sub, n_sub * L, L)
t_rest = run(c_rest, rest[:n_rest * L], off_rest, f_rest)
t_sub = run(c_sub, sub[:n_sub * L], off_sub, f_sub)
print("the rest alone: %.1f ms; the re-chained reads alone: %.1f ms" % (t_rest, t_sub), flush=True)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
best = 1e9
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    th = threading.Thread(target=run, args=(c_sub, sub[:n_sub * L], off_sub, f_sub, 1, s2))
    th.start()
    with torch.cuda.stream(s1):
        c_rest.classify(rest[:n_rest * L], off_rest, f_rest, None, want_stats=False)
    s1.synchronize(); th.join(); torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) * 1e3)
ok = bool((torch.cat([f_rest, f_sub]).sum() == d_flags.sum()).item())
print("side by side (two contexts, two streams, two host threads): %.1f ms; same number of host reads: %s" % (best, ok), flush=True)
