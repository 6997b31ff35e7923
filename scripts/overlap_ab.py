#!/usr/bin/env python3
"""A/B: the 20 M-record batch classified by T host threads, each with its own context and stream over a contiguous
1/T of the records (the reference's rayon workers share one aligner the same way), staggered so that one thread's
sketch/probe kernel (HBM-latency-bound) overlaps another's repeat path (LDS-bound)."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scrubby_amd import lib as S
import bench as B

def main():
    n_rec = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
    dev = torch.device("cuda", 0); torch.cuda.set_device(0)
    P, R = S.ref_params(B.REF_SEED, B.CHM13_CONTIGS), S.read_params(B.READ_SEED)
    G, L = P.genome_len, R.read_len
    opts = S.preset("sr")
    d_ref = torch.empty(G + 64, dtype=torch.uint8, device=dev)
    S.synth_ref_device(P, 0, G, d_ref)
    index = S.Index.build_device(d_ref, [P.contig_start[i] for i in range(len(B.CHM13_CONTIGS) + 1)], opts, device=0)
    del d_ref; torch.cuda.empty_cache()
    d_reads = torch.empty(n_rec * L + 64, dtype=torch.uint8, device=dev)
    d_off = torch.empty(n_rec + 1, dtype=torch.int64, device=dev)
    S.synth_reads_device(P, R, 0, n_rec, d_reads, d_off)
    d_flags = torch.zeros(n_rec, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    ref_flags = None
    for T, parts in ((1, 1), (2, 2), (2, 4), (2, 8), (3, 6), (4, 8)):
        # `parts` sub-batches handed round-robin to T threads
        per = (n_rec // parts + 63) // 64 * 64
        bounds = [(i * per, min(n_rec, (i + 1) * per)) for i in range(parts) if i * per < n_rec]
        ctxs = [S.Context(index, per, per * L, L) for _ in range(T)]
        streams = [torch.cuda.Stream(device=dev) for _ in range(T)]
        offs = [(d_off[a:b + 1] - d_off[a]).contiguous() for a, b in bounds]
        torch.cuda.synchronize()
        def work(t):
            with torch.cuda.stream(streams[t]):
                for i in range(t, len(bounds), T):
                    a, b = bounds[i]
                    ctxs[t].classify(d_reads[a * L:b * L], offs[i], d_flags[a:b], None, want_stats=False)
        def step():
            th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
            for x in th: x.start()
            for x in th: x.join()
            torch.cuda.synchronize()
        step()
        t0 = time.perf_counter()
        K = 3
        for _ in range(K): step()
        dt = (time.perf_counter() - t0) / K
        fl = d_flags.cpu().numpy().copy()
        if ref_flags is None: ref_flags = fl
        print(f"threads {T} parts {parts}: {dt * 1e3:7.2f} ms/step  {n_rec / dt / 1e6:7.1f} M reads/s  flags equal: {bool(np.array_equal(fl, ref_flags))}  removed {int((fl == 1).sum())}", flush=True)
        for c in ctxs: c.close()
        del ctxs; torch.cuda.empty_cache()

main()
