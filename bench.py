#!/usr/bin/env python3
"""bench.py — reads/s depleted on the BASELINE.json workload, one process per GPU.

Workload (BASELINE.json configs[1]): 10 M synthetic 2x150 bp pairs
(20 M records, 50 % host) classified against a CHM13v2-sized (3 117 292 070 bp, 25 contigs)
synthetic reference with the `sr` preset.  No network on the GPU box, so the reference and the
reads are generated on the device from fixed seeds (scrubby_amd/csrc/sh_synth_core.h).

A "step" = one pass of the hot path (sketch+probe -> chain -> extension filter -> flags) over this
rank's batch of records, inputs already resident in HBM.  With N > 1 (BASELINE.json configs[2]) the SAME
20 M records are cut into contiguous pair-aligned shards (scrubby_amd/dist.py shard_range), one per rank,
every rank holds a replica of the index, and the only exchange is the final union of the depleted-record
bitmap (all_gather over RCCL), done inside the timed region: strong scaling, `value` = 20 M records /
max-over-ranks step time.  The weak-scaling figure (20 M records per GPU) is measured right after it and
reported as the secondary object `weak_scaling`.

Launch: `python bench.py --gpus N ...` starts the N ranks itself (child processes, spawned before anything
touches a GPU); under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` it takes
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the launcher instead.  Ranks that have to share one device
(N larger than the visible device count, e.g. a rehearsal on a one-GPU box) do the union over gloo with
CPU copies of the bitmap, since RCCL refuses two ranks on one device.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      the stage that takes longest (`kernel` lists the kernels it sums; --chain-only: k_sketch_probe alone): algorithmic
                bytes / HIP-event time vs 8 TB/s, the other stages in stage_ms_per_step, K1's own figure in streaming_kernel
  cpu_baseline  the CPU oracle (oracle/, "port": restated minimap2 decision path, NOT minimap2-rs)
                on the host cores, on a bounded sample of the same records and the same index
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from scrubby_amd import lib as S  # noqa: E402

# CHM13v2 sequence lengths (chr1..22, X, Y, M of T2T-CHM13v2.0), total 3 117 292 070 bp
CHM13_CONTIGS = [248387328, 242696752, 201105948, 193574945, 182045439, 172126628, 160567428, 146259331,
                 150617247, 134758134, 135127769, 133324548, 113566686, 101161492, 99753195, 96330374,
                 84276897, 80542538, 61707364, 66210255, 45090682, 51324926, 154259566, 62460029, 16569]
REF_SEED, READ_SEED = 0x5C2B0010, 0x5C2B0011
REPEAT_STAGE = "repeat path (k_local_cluster, k_expand, k_sort_top / k_sort_lds classes, k_giant_*, k_cluster_dp, k_finalize)"
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--records", type=int, default=20_000_000, help="records per GPU (2 per pair)")
    ap.add_argument("--chunk", type=int, default=1 << 25, help="records per kernel launch (default: the whole batch in one launch)")
    ap.add_argument("--small", action="store_true", help="5 Mb reference / 200k records (plumbing check)")
    ap.add_argument("--workload", choices=["sr", "sr-div", "ont", "k2", "e2e", "e2e-k2"], default="sr",
                    help="sr = BASELINE configs[1] (headline); ont = configs[3] stand-in: long noisy reads, map-ont preset; "
                         "k2 = configs[4] stand-in: Kraken2-style taxid classification of 2x150 bp pairs against an 8 GB table (not the headline metric)")
    ap.add_argument("--k2-cells", type=int, default=2_000_000_000, help="cells of the compact hash table (4 B each)")
    ap.add_argument("--k2-nodes", type=int, default=50_000, help="taxonomy nodes of the synthetic database")
    ap.add_argument("--ont-preset", default="map-ont", choices=["map-ont", "lr:hq", "map-hifi"], help="preset of --workload ont (the bench line is map-ont; the others are parity checks at scale)")
    ap.add_argument("--ont-chunk", type=int, default=1_000_000, help="long reads per launch (--workload ont): a launch pays the extension stage's longest single reads once, so fewer, larger launches are faster; 1 M reads (6.3 Gbases) is what a 288 GB device holds")
    ap.add_argument("--e2e-threads", type=int, default=0, help="-t of `scrubby reads` for --workload e2e (0: the usable cores - physical, capped by the cgroup quota)")
    ap.add_argument("--e2e-gz", action="store_true", help="--workload e2e: write .fastq.gz outputs")
    ap.add_argument("--e2e-legacy", action="store_true", help="--workload e2e: also time the collect-then-map host path")
    ap.add_argument("--e2e-dir", default=None, help="--workload e2e: scratch directory (default: a temp dir)")
    ap.add_argument("--chain-only", action="store_true",
                    help="decide at chain level (opts.flags without SH_F_CIGAR: round 1's decision, what the reference computes WITHOUT .with_cigar()); "
                         "a comparison line, not the headline")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-weak", action="store_true", help="N > 1: skip the secondary weak-scaling measurement (20 M records per GPU)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--gather-bench", action="store_true", help="also time raw 16-B random gathers over the table")
    ap.add_argument("--no-secondary", action="store_true",
                    help="headline run (--workload sr, one GPU): skip the `secondary` object - configs[3] (long reads, map-ont) and configs[4] (Kraken2-style) "
                         "measured in the same process AFTER the headline's timed region")
    ap.add_argument("--secondary-cpu-seconds", type=float, default=8.0, help="CPU-oracle sample of each secondary line")
    return ap.parse_args()


def spawn_ranks(a):
    """`bench.py --gpus N` outside a launcher: start the N ranks as child processes of this one, which has not touched a GPU
    (no exec from a process with an initialised HIP runtime), hand them the rendezvous through the environment, pass rank 0's
    JSON line through and return the worst exit code."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    n_dev = torch.cuda.device_count()              # does not initialise the runtime
    shared_device = world > max(n_dev, 1)          # rehearsal: more ranks than devices
    local = local % max(n_dev, 1)
    backend = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = "gloo" if shared_device else "nccl"
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    S.require_gpu()
    if a.workload == "e2e":
        return main_e2e(a, rank, world, local, dev)
    if a.workload == "e2e-k2":
        return main_e2e_k2(a, rank, world, local, dev)
    out = main_k2(a, rank, world, local, dev, backend) if a.workload == "k2" else main_reads(a, rank, world, local, dev, backend)
    if out is not None and a.workload == "sr" and world == 1 and not a.chain_only and not a.no_secondary:
        out["secondary"] = secondary_lines(a, rank, local, dev)
    if out is not None:
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def secondary_lines(a, rank, local, dev):
    """configs[3] and configs[4] in the headline's own run, so that the driver's clock witnesses them too: measured after the headline's timed
    region has ended and its buffers are freed (nothing here touches `value` / `ms_per_step` / `roofline` of the headline).  Each entry is the
    line `--workload ont` / `--workload k2` prints, cut to what a reader checks: value, ms_per_step, roofline (with PMC traffic), the
    long-read stage's exactness counters, a small CPU-oracle sample (and, for long reads, the stratified check over every read whose answer
    took a path with a documented limit)."""
    import copy
    import gc
    sec = {"note": "measured in this process after the headline's timed region; not part of value / ms_per_step / roofline above"}
    for name, steps, warm in (("ont", 1, 1), ("k2", 3, 1)):
        gc.collect()
        S.release_cached_context()
        torch.cuda.empty_cache()
        b = copy.copy(a)
        b.workload, b.steps, b.warmup, b.cpu_seconds, b.gather_bench = name, steps, warm, a.secondary_cpu_seconds, False
        b.records = 20_000_000      # the sentinel both workloads read as "BASELINE's own size"
        t0 = time.time()
        try:
            o = main_k2(b, rank, 1, local, dev, None) if name == "k2" else main_reads(b, rank, 1, local, dev, None)
        except Exception as e:      # a secondary line must never take the headline with it
            sec[name] = {"error": f"{type(e).__name__}: {e}"}
            continue
        keep = ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "data", "roofline", "cpu_baseline", "stratified_parity", "result", "database")
        e = {k: o[k] for k in keep if k in o}
        e["workload"] = o["config"]["workload"]
        e["wall_s_incl_setup"] = round(time.time() - t0, 1)
        sec[name] = e
    return sec


def main_reads(a, rank, world, local, dev, backend):
    if world > 1:
        import torch.distributed as dist
    contigs = [1_000_000] * 5 if a.small else CHM13_CONTIGS
    n_rec = 200_000 if a.small else a.records
    ont = a.workload == "ont"
    if ont:
        n_rec = 20_000 if a.small else (a.records if a.records != 20_000_000 else 2_000_000)      # BASELINE configs[3]: 2 M reads
    P = S.ref_params(REF_SEED, contigs)
    # long reads: 2 % substitutions + 1.56 % insertions + 1.56 % deletions (n_read_pct = 1 switches the generator's indels on)
    R = S.read_params(0x5C2B0020, host_pct=50, sub_per_10k=200, n_read_pct=1) if ont else S.read_params(READ_SEED)
    G = P.genome_len
    opts = S.preset(a.ont_preset if ont else "sr")
    if a.chain_only:
        opts.flags &= ~S.SH_F_CIGAR

    # ---- setup (untimed): reference -> index -> reads, all in HBM -------------------------------------
    real_ref = os.environ.get("SCRUBBY_CHM13")
    real_ref = real_ref if real_ref and os.path.exists(real_ref) and not a.small and not ont else None
    ref_source = ("real FASTA from $SCRUBBY_CHM13: " + real_ref) if real_ref else "synthetic, CHM13v2-sized (sh_synth_core.h; no real CHM13 on a box without network)"
    t0 = time.time()
    if real_ref:
        t_ref = 0.0
        index = S.Index.build_fasta(real_ref, opts, device=local)
    else:
        d_ref = torch.empty(G + 64, dtype=torch.uint8, device=dev)
        S.synth_ref_device(P, 0, G, d_ref)
        torch.cuda.synchronize()
        t_ref = time.time() - t0
        t0 = time.time()
        index = S.Index.build_device(d_ref, [P.contig_start[i] for i in range(len(contigs) + 1)], opts, device=local)
        del d_ref
    torch.cuda.synchronize()
    t_idx = time.time() - t0
    info = index.info()
    torch.cuda.empty_cache()
    if real_ref:
        G = info["n_bases"]
        packed_h, starts_h = index.export_ref()
        d_packed = torch.from_numpy(packed_h).to(dev)
        del packed_h

    L = R.read_len
    from scrubby_amd import dist as D
    via_host = backend == "gloo"
    n_total = n_rec                      # records of the whole job (strong scaling: cut over the ranks)
    union = {"bytes": 0}

    def make_batch(r_lo, n):
        """Records [r_lo, r_lo + n) of the global record space, resident in HBM, with a context sized for them."""
        b = {"r_lo": r_lo, "n": n}
        if ont:
            lens = long_read_lengths(R.seed, r_lo, n)
            off_np = np.zeros(n + 1, dtype=np.int64)
            off_np[1:] = np.cumsum(lens.astype(np.int64))
            b["n_bases"] = int(off_np[-1])
            b["d_off"] = torch.from_numpy(off_np).to(dev)
            b["d_reads"] = torch.empty(b["n_bases"] + 64, dtype=torch.uint8, device=dev)
            S.synth_long_reads_device(P, R, r_lo, n, b["d_off"], b["n_bases"], b["d_reads"])
            a.chunk = min(a.chunk, a.ont_chunk)
            b["ctx"] = S.Context(index, max(min(a.chunk, n), 1), b["n_bases"], int(lens.max()) if n else 1)
        else:
            b["n_bases"] = n * L
            b["d_reads"] = torch.empty(b["n_bases"] + 64, dtype=torch.uint8, device=dev)
            b["d_off"] = torch.empty(n + 1, dtype=torch.int64, device=dev)
            if real_ref:
                reads_from_packed(d_packed, G, r_lo, n, L, R.host_pct, R.sub_per_10k, READ_SEED, b["d_reads"], b["d_off"])
            else:
                S.synth_reads_device(P, R, r_lo, n, b["d_reads"], b["d_off"])
            if a.workload == "sr-div":
                diverge_reads(b["d_reads"], n, L, r_lo, dev)
            b["ctx"] = S.Context(index, max(min(a.chunk, n), 1), b["n_bases"], L)
        b["d_flags"] = torch.zeros(max(n, 1), dtype=torch.uint8, device=dev)[:n]
        torch.cuda.synchronize()
        return b

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(b, slice_bytes, steps, warmup):
        """W warm-up steps, then exactly K steps between barrier + synchronize; max over ranks."""
        def step():
            st = b["ctx"].classify(b["d_reads"][:b["n_bases"]], b["d_off"], b["d_flags"], None, want_stats=True)
            if world > 1:   # depleted-record bitmap union: disjoint slices, one all_gather (RCCL) - SURVEY.md §8e, cleaner.rs:564-570
                gathered, _ = D.union_depleted(b["d_flags"], slice_bytes=slice_bytes, via_host=via_host)
                union["bytes"] = gathered.numel()
                b["gathered"] = gathered
            return st
        for _ in range(warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        sts = [step() for _ in range(steps)]
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cpu" if via_host else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, sts

    # ---- headline: the job's n_total records, cut over the ranks (N = 1: the whole batch) -------------------
    if ont and world > 1:      # ragged reads: shards of equal bases, not equal counts
        cum = np.cumsum(long_read_lengths(R.seed, 0, n_total).astype(np.int64))
        cuts = [0] + [int(np.searchsorted(cum, cum[-1] * r // world)) for r in range(1, world)] + [n_total]
        lo, hi = cuts[rank], cuts[rank + 1]
        slice_bytes = (max(cuts[r + 1] - cuts[r] for r in range(world)) + 7) // 8
    else:
        lo, hi = D.shard_range(n_total, rank, world)
        slice_bytes = (D.shard_range(n_total, 0, world)[1] + 7) // 8
    B = make_batch(lo, hi - lo)
    dt, stats = timed(B, slice_bytes, a.steps, a.warmup)
    union_bytes = union["bytes"]
    n_rec = hi - lo                       # this rank's records: what its kernels saw
    n_bases, ctx, d_reads, d_flags, d_off = B["n_bases"], B["ctx"], B["d_reads"], B["d_flags"], B["d_off"]
    n_host = int((d_flags == 1).sum().item())
    ms_step = dt / a.steps * 1e3
    value = n_total * a.steps / dt
    removed_total = n_host
    if world > 1 and not ont:
        removed_total = int(D.gathered_to_flags(B["gathered"], slice_bytes, n_total, world).sum().item())

    # ---- secondary: weak scaling (n_total records PER GPU), N > 1 only ----------------------------------------
    weak = None
    if world > 1 and not a.no_weak:
        del B, ctx, d_reads, d_flags, d_off
        torch.cuda.empty_cache()
        Bw = make_batch(rank * n_total, n_total)
        dtw, _ = timed(Bw, (n_total + 7) // 8, a.steps, max(a.warmup, 1))
        weak = {"value": round(world * n_total * a.steps / dtw, 1), "unit": "reads/s", "ms_per_step": round(dtw / a.steps * 1e3, 3),
                "records_per_gpu": n_total, "scaling": "weak"}
        B = Bw
        n_bases, ctx, d_reads, d_flags, d_off = B["n_bases"], B["ctx"], B["d_reads"], B["d_flags"], B["d_off"]

    # ---- roofline (HIP events on the launch stream, inside the library) ---------------------------------
    # per-stage algorithmic bytes per step: K1 = L + 8 (offset) + 16 per probe + 1 (flag) per read;
    # chain stages = 16 B per seed record + 8 B per anchor.  The primary entry is the stage that takes longest.
    k1_ms = float(np.mean([s["ms_sketch_probe"] for s in stats]))      # per step = sum over its launches
    k2_ms = float(np.mean([s["ms_chain_small"] for s in stats]))
    k3_ms = float(np.mean([s["ms_chain_large"] for s in stats]))
    n_launch = (n_rec + ctx_chunk(a, n_rec) - 1) // ctx_chunk(a, n_rec)
    s0 = stats[-1]
    k1_bytes = s0["n_bases"] + 9 * s0["n_reads"] + 16 * s0["n_minimizers"]
    n_seeded = s0["n_reads"] - s0["n_no_seed"]
    seeds_small = 20 * s0["n_chain_small"]                 # ~20 seed records per read on the LDS path
    k2_bytes = 16 * seeds_small + 8 * seeds_small
    k3_bytes = 16 * 20 * s0["n_chain_large"] + 8 * s0["n_anchors"]
    stages = {
        "k_sketch_probe": (k1_ms, k1_bytes), "k_chain_small": (k2_ms, k2_bytes),
        REPEAT_STAGE: (k3_ms, k3_bytes),
    }
    ext_ms = float(np.mean([s.get("ms_ext", 0.0) for s in stats]))
    if ext_ms > 0:      # SH_F_CIGAR: list building + base-level alignment of the reads no shortcut settles (reads + their reference windows + 40-B chain records)
        stages["extension stage (k_long_chains + k_regs_align_long)" if ont else "extension stage (k_ext_* + k_regs_align)"] = (ext_ms, 40 * s0.get("n_ext_regions", 0) + 2 * (n_bases // max(n_rec, 1)) * s0.get("n_ext_reads", 0))      # mean read length of the batch (a launch's n_bases / n_reads is not it when a step is several launches)
    # the roofline object describes the stage that takes LONGEST.  Round 1 (chain-level decision, --chain-only): k_sketch_probe, one streaming
    # kernel.  With the extension stage (SH_F_CIGAR, the default, what .with_cigar() makes the reference compute) the repeat path leads:
    # a few thousand satellite reads re-chained with max_occ hold most of the anchors, and their sort + sequential DP is latency-bound,
    # far from the HBM roof - the fraction says so.  `streaming_kernel` keeps K1's own figure beside it.
    dom = max(stages, key=lambda k: stages[k][0])
    d_ms, d_bytes = stages[dom]
    achieved = d_bytes / (d_ms * 1e-3) / 1e9
    roofline = {
        "bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
        "launches_per_step": n_launch, "avg_launch_ms": round(d_ms / n_launch, 4),
        "alg_bytes_per_launch": int(d_bytes / n_launch),
        "stage_ms_per_step": {k: round(v[0], 3) for k, v in stages.items()},
        "stage_achieved_GBs": {k: round(v[1] / (v[0] * 1e-3) / 1e9, 1) if v[0] > 0 else None for k, v in stages.items()},
        "path_achieved_GBs": round((k1_bytes + k2_bytes + k3_bytes) / ((k1_ms + k2_ms + k3_ms) * 1e-3) / 1e9, 1),
        "n_seeded_reads": n_seeded,
        "streaming_kernel": {"kernel": "k_sketch_probe", "achieved": round(k1_bytes / (k1_ms * 1e-3) / 1e9, 1) if k1_ms > 0 else None,
                             "frac": round(k1_bytes / (k1_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if k1_ms > 0 else None},
    }
    attach_traffic(roofline, "ont" if ont else "sr", dom, ctx_chunk(a, n_rec), skip=a.small)

    gather = None
    if a.gather_bench and rank == 0:
        gbs, gms = index.gather_bench(1 << 28, 3)
        gather = {"useful_GBs_16B_probes": round(gbs, 1), "ms": round(gms, 3), "probes": 1 << 28}

    # ---- CPU baseline: the oracle on the host cores, same index, bounded sample -------------------------
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu:
        cpu = cpu_baseline(index, info, d_reads, n_rec, L, a.cpu_seconds, d_flags, d_off if ont else None, a.ont_preset if ont else "sr", a.chain_only)
    strat = None
    if rank == 0 and world == 1 and not a.no_cpu and not ont and not a.chain_only and ctx_chunk(a, n_rec) >= n_rec:
        strat = stratified_parity(index, ctx, info, d_reads, d_flags, n_rec, L)
    if rank == 0 and world == 1 and not a.no_cpu and ont and not a.chain_only:
        # every read whose answer took one of the long-read stage's rarer paths (of the whole batch: the lists describe the last CALL), plus a random rest
        strat = stratified_parity_long(index, ctx, info, d_reads, d_off, d_flags, n_rec, a.ont_preset, a.cpu_seconds)
    ext_oracle = None
    if rank == 0 and world == 1 and not a.no_cpu:      # the real tool, when the box has it: `-c -x sr` / `-c -x map-ont`
        ext_oracle = external_oracle(index, d_reads, n_rec, L, d_flags, contigs, P, dev, real_ref, preset=a.ont_preset if ont else "sr", d_off=d_off if ont else None)

    # ---- the host-buffer entry point (sh_classify_batch: what a Rust caller binds), PCIe included; never `value` -----------
    host_path = None
    if rank == 0 and world == 1 and not a.no_cpu and not ont and not a.small:
        h_reads = d_reads[:n_bases].cpu().numpy()
        h_off = np.arange(n_rec + 1, dtype=np.uint64) * L
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            hf, _, _, _ = index.classify(h_reads, h_off, want_trace=False)
            dt_h = time.perf_counter() - t0
            best = dt_h if best is None or dt_h < best else best
        host_path = {"entry": "sh_classify_batch (pageable host buffers in, host flags out, PCIe included)", "reads_per_s": round(n_rec / best, 1),
                     "ms": round(best * 1e3, 1), "flags_equal_device_path": bool(np.array_equal(hf, d_flags.cpu().numpy()))}
        del h_reads

    # what a call costs whatever its size (launches, host read-backs between the passes): 10 k records through the same context
    fixed_ms = None
    if rank == 0 and not ont and n_rec >= 10_000:
        ts = []
        d_flags_small = torch.zeros(10_000, dtype=torch.uint8, device=dev)
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ctx.classify(d_reads[:10_000 * L], d_off[:10_001], d_flags_small, None, want_stats=False)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        fixed_ms = round(min(ts), 3)
    if rank == 0:
        out = {
            "metric": ("reads/s depleted (long reads, map-ont, vs CHM13v2-sized reference) - NOT the headline metric" if ont else
                       "reads/s depleted (2x150bp PE diverged / chimeric, vs CHM13v2-sized reference) - NOT the headline metric" if a.workload == "sr-div" else
                       "reads/s depleted (2x150bp PE vs CHM13v2-sized reference), records classified per second"),
            "value": round(value, 1), "unit": "reads/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u64/i32 (f32 in the chain gap penalty)", "data": "synthetic",
            "union_bytes_gathered": union_bytes, "collective_backend": backend, "weak_scaling": weak,
            "config": {
                "workload": ("configs[3] stand-in: %d long reads (log-normal-like lengths, median 5.4 kb; 2 %% substitutions, 1.56 %% insertions, 1.56 %% deletions), map-ont preset; "
                             "segment-parallel long-read front end + repeat path" % n_rec if ont else
                             "sr-div (NOT the headline): the records of configs[1] diverged in place - a third each with 5 / 10 / 15 %% substitutions, 2 %% chimeras, 5 %% with a 15 %%-diverged 70-base stretch of another record spliced in - vs the same reference, sr preset" if a.workload == "sr-div" else
                             "cfg1-small: 200k records vs 5 Mb" if a.small else
                             ("configs[2]: the 10M synthetic 2x150bp PE (20M records) of configs[1], read-sharded over %d GPUs, vs CHM13v2-sized synthetic reference, sr preset" % world) if world > 1 else
                             "configs[1]: 10M synthetic 2x150bp PE (20M records) vs CHM13v2-sized synthetic reference, sr preset, k-mer/minimizer classifier path"),
                "records_total": n_total, "records_rank0": n_rec, "read_len": (round(n_bases / n_rec, 1) if ont else L), "host_pct": R.host_pct, "reference_bp": int(G),
                "preset": a.ont_preset if ont else "sr", "decision": ("chain level (--chain-only: no extension stage)" if a.chain_only else
                                                                      "mappings.len() > 0 after the long-read branch of the extension filter (with_cigar): RMQ long join, mm_est_err, gap-filling alignment, mm_filter_regs" if ont else
                                                                      "mappings.len() > 0 after the extension filter (with_cigar)"), "k": info["k"], "w": info["w"], "records_per_launch": ctx_chunk(a, n_rec),
                "parallelism": f"read-sharded x{world} (contiguous pair-aligned ranges of the same records), index replicated",
                "ref_seed": hex(REF_SEED), "read_seed": hex(R.seed),
            },
            "result": {"reads_removed": removed_total, "reads_removed_rank0": n_host, "n_no_seed": s0["n_no_seed"], "n_chain_small": s0["n_chain_small"],
                       "n_chain_large": s0["n_chain_large"], "probes": s0["n_minimizers"],
                       "repeat_path_anchors": s0["n_anchors"], "repeat_path_clusters": s0["n_clusters"], "n_resketch": s0["n_resketch"],
                       "pair_decided": s0["n_pair_decided"],
                       # the extension stage `.with_cigar()` enables: reads it had to align, regions, and the flags it flips
                       "ext_shortcut_reads": s0["n_ext_shortcut"], "ext_reads": s0["n_ext_reads"], "ext_regions": s0["n_ext_regions"],
                       "ext_flags_flipped": s0["n_ext_dropped"], "ext_ms_per_step": round(float(np.mean([x["ms_ext"] for x in stats])), 3),
                       # reads whose regs[0] did not survive and that were re-chained with every chain kept (sr), and what that costs per step
                       "ext_fallback_reads": s0.get("n_ext_fallback", 0), "ext_fallback_ms_per_step": round(float(np.mean([x.get("ms_ext_fallback", 0.0) for x in stats])), 3),
                       "ext_fallback_share_of_step": round(float(np.mean([x.get("ms_ext_fallback", 0.0) for x in stats])) / max(ms_step, 1e-9), 4),
                       "rmq_rechained": s0.get("n_rmq_rechained", 0), "rmq_tied": s0.get("n_rmq_tied", 0), "rmq_exact": s0.get("n_rmq_exact", 0), "rmq_open": s0.get("n_rmq_open", 0),
                       "ext_unresolved": s0.get("n_ext_unresolved", 0), "ext_ondemand": s0.get("n_ext_ondemand", 0), "locus_reads": s0.get("n_locus_reads", 0), "locus_redone": s0.get("n_locus_redone", 0),
                       "dp_parallel_reads": s0.get("n_dp_parallel", 0), "dp_dirty_anchors": s0.get("n_dp_dirty", 0), "top_settled_reads": s0.get("n_top_settled", 0)},
            "index": {"n_keys": info["n_keys"], "n_minimizers": info["n_minimizers"], "n_slots": info["n_slots"],
                      "n_positions": info["n_positions"], "hbm_GB": round(info["hbm_bytes"] / 1e9, 2),
                      "build_s": round(t_idx, 2), "ref_synth_s": round(t_ref, 2)},
            "fixed_ms_per_call": fixed_ms,
            "roofline": roofline, "cpu_baseline": cpu, "stratified_parity": strat, "host_buffer_path": host_path,
            # the real tools, if this box has them (it has no network, so normally it does not); a real CHM13v2 FASTA
            # given through $SCRUBBY_CHM13 replaces the synthetic reference of the same size
            "external_oracle": ext_oracle, "reference_source": ref_source,
        }
        if gather:
            out["gather_ceiling"] = gather
        return out
    return None


def k2_taxonomy(n_nodes, seed):
    """Synthetic taxonomy in Kraken 2's layout (breadth-first ids, children consecutive): the true human lineage plus a
    random bacterial tree with real rank names, >= n_nodes nodes."""
    rng = np.random.default_rng(seed)
    lineage = [("root", 1, "no rank"), ("cellular organisms", 131567, "no rank"), ("Eukaryota", 2759, "superkingdom"),
               ("Opisthokonta", 33154, "clade"), ("Metazoa", 33208, "kingdom"), ("Chordata", 7711, "phylum"), ("Mammalia", 40674, "class"),
               ("Primates", 9443, "order"), ("Hominidae", 9604, "family"), ("Homo", 9605, "genus"), ("Homo sapiens", 9606, "species")]
    branks = ["superkingdom", "phylum", "class", "order", "family", "genus", "species", "strain"]
    parents, externals, names, ranks = [0], [0], [""], [""]
    # queue entries: (parent id, lineage index or -1, bacterial depth or -1)
    queue = [(0, 0, -1)]
    next_ext = 1_000_000
    qi = 0
    while qi < len(queue):
        par, li, bd = queue[qi]; qi += 1
        me = len(parents)
        parents.append(par)
        if li >= 0:
            names.append(lineage[li][0]); externals.append(lineage[li][1]); ranks.append(lineage[li][2])
            if li + 1 < len(lineage):
                queue.append((me, li + 1, -1))
            if li == 1:
                queue.append((me, -1, 0))                  # Bacteria beside Eukaryota
        else:
            if bd == 0:
                names.append("Bacteria"); externals.append(2)
            else:
                names.append(f"{branks[bd]} {next_ext}"); externals.append(next_ext); next_ext += 1
            ranks.append(branks[bd])
            if bd + 1 < len(branks) and len(queue) < n_nodes:
                for _ in range(int(rng.integers(3, 9))):
                    queue.append((me, -1, bd + 1))
    ids = {n: i for i, n in enumerate(names)}
    return parents, externals, names, ranks, ids


def main_k2(a, rank, world, local, dev, backend=None):
    """BASELINE configs[4] stand-in: Kraken2-style classification of 2x150 bp pairs against a table of --k2-cells 32-bit
    cells in HBM (default 2e9 = 8 GB, load ~0.7: every minimizer of the CHM13v2-sized synthetic reference under Homo
    sapiens + pseudo-random filler keys over the bacterial taxa), taxonomy of >= 50 000 nodes.
    N > 1 = strong scaling like the headline: the SAME pairs cut into N contiguous shards (dist.shard_range), table replicated, no
    data-path collective; the per-pair calls are all-gathered inside the timed region (what rank 0 needs to write kraken.reads and the
    report for the whole job)."""
    from scrubby_amd import k2 as K
    from scrubby_amd import dist as D
    contigs = [1_000_000] * 5 if a.small else CHM13_CONTIGS
    n_rec = 200_000 if a.small else (a.records if a.records != 20_000_000 else 40_000_000)
    n_rec -= n_rec & 1
    cells = 12_000_017 if a.small else a.k2_cells
    P = S.ref_params(REF_SEED, contigs)
    R = S.read_params(0x5C2B0030)
    G = P.genome_len
    t0 = time.time()
    parents, externals, names, ranks, ids = k2_taxonomy(2_000 if a.small else a.k2_nodes, 0x5C2B0030)
    o = K.default_opts()
    db = K.K2Db.create(o, cells, parents, externals, names, ranks, device=local)
    d_ref = torch.empty(G + 64, dtype=torch.uint8, device=dev)
    S.synth_ref_device(P, 0, G, d_ref)
    n_runs = db.insert_sequence_device(d_ref, G, ids["Homo sapiens"])
    del d_ref
    torch.cuda.empty_cache()
    size_host = db.info()["size"]
    fill = max(int(0.70 * cells) - size_host, 0)
    db.insert_random(0x5C2B0031, fill, ids["Bacteria"], len(parents) - 1)
    torch.cuda.synchronize()
    info = db.info()
    t_db = time.time() - t0

    L = R.read_len
    n_total = n_rec                                  # the job: the same records whatever N
    lo, hi = D.shard_range(n_total, rank, world)
    n_rec = hi - lo                                  # this rank's shard
    n_bases = n_rec * L
    d_reads = torch.empty(n_bases + 64, dtype=torch.uint8, device=dev)
    d_off = torch.empty(n_rec + 1, dtype=torch.int64, device=dev)
    S.synth_reads_device(P, R, lo, n_rec, d_reads, d_off)
    n_units = n_rec // 2
    slice_len = D.shard_range(n_total, 0, world)[1] // 2
    via_host = backend == "gloo"
    gathered = None
    d_out = torch.zeros((n_units, 4), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    copts = None
    if os.environ.get("SCRUBBY_K2_NOPROBE"):      # timing experiment: the scan alone (every minimizer below the down-sampling threshold)
        copts = db.opts(); copts.min_acceptable_hash = (1 << 64) - 1
    st = None

    def step():
        nonlocal gathered
        s_ = db.classify_device(d_reads, d_off, n_rec, True, d_out, copts)
        if world > 1:      # the calls of the whole job on every rank (column 0 of the result = the external taxid)
            gathered, _ = D.gather_calls(d_out[:, 0], slice_len, via_host=via_host)
        return s_
    for _ in range(a.warmup):
        st = step()
    barrier()
    t0 = time.time()
    ms_kernel = 0.0
    for _ in range(a.steps):
        st = step()
        ms_kernel += st["ms_classify"]
    barrier()
    dt = time.time() - t0
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_step = dt * 1e3 / max(a.steps, 1)
    value = n_total / (ms_step * 1e-3)
    res = d_out.cpu().numpy().view(K.RESULT_DTYPE).reshape(-1)
    human_total = int((res["taxid"] == 9606).sum())
    if world > 1:
        human_total = int((D.gathered_to_calls(gathered, slice_len, n_total, world) == 9606).sum().item())
    # algorithmic bytes per pair: L1 + L2 + 8 + 4 * P + 4
    alg = n_bases + 8 * n_units + 4 * st["n_probes"] + 4 * n_units
    avg_ms = ms_kernel / max(a.steps, 1)
    roof = {"bound": "hbm", "kernel": "k_k2_classify", "achieved": round(alg / (avg_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(alg / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None, "launches_per_step": 1, "avg_launch_ms": round(avg_ms, 3),
            "alg_bytes_per_launch": int(alg), "probes_per_launch": int(st["n_probes"]),
            "probe_rate_G_per_s": round(st["n_probes"] / (avg_ms * 1e-3) / 1e9, 2)}
    attach_traffic(roof, "k2", "k_k2_classify", n_rec)
    cpu = None
    if rank == 0 and not a.no_cpu:
        from oracle import oracle as O
        cells_h, parent_h, ext_h = db.export()
        tab = O.K2Table(cells_h, parent_h, info["value_bits"])
        cores = usable_cores()      # physical cores capped by the cgroup CPU quota
        n_s = min(n_rec, 200_000)
        sample = d_reads[: n_s * L].cpu().numpy()
        offs = np.arange(n_s + 1, dtype=np.uint64) * L
        oo = O.k2_default_opts()
        t1 = time.time()
        tab.classify(oo, sample[: 20_000 * L], offs[: 20_001], paired=True, threads=cores)
        rate = 20_000 / max(time.time() - t1, 1e-6)
        n_s = int(min(n_rec, max(20_000, rate * a.cpu_seconds))) & ~1
        sample = d_reads[: n_s * L].cpu().numpy()
        offs = np.arange(n_s + 1, dtype=np.uint64) * L
        t1 = time.time()
        c = tab.classify(oo, sample, offs, paired=True, threads=cores)
        dtc = time.time() - t1
        diff = int((c["call"] != res["call"][: n_s // 2]).sum())
        cpu = {"value": round(n_s / dtc, 1), "unit": "reads/s", "cores": cores, "kind": "port",
               "sample": f"first {n_s} records of the same batch, same table (copied from HBM), {cores} threads, {dtc:.1f} s; "
                         f"restatement baseline - not kraken2; calls differing from the GPU on the sample: {diff}"}
    if rank == 0:
        out = {"metric": "reads/s classified (Kraken2-style taxid path, 2x150bp PE vs 8 GB table) - NOT the headline metric",
               "value": round(value, 1), "unit": "reads/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_step, 3),
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u64 (minimizers) / u32 (cells, taxa)", "data": "synthetic",
               "calls_gathered_bytes": (world * slice_len * 4 if world > 1 else 0), "collective_backend": backend,
               "config": {"workload": "configs[4] stand-in: %d synthetic 2x150bp pairs vs a Kraken2-format table of %d cells (%.1f GB, load %.2f), "
                                      "taxonomy of %d nodes, k=35 l=31, confidence 0, minimum-hit-groups 2" %
                                      (n_total // 2, info["capacity"], info["capacity"] * 4 / 1e9, info["size"] / info["capacity"], info["n_nodes"]),
                          "records_total": n_total, "records_rank0": n_rec, "read_len": L,
                          "parallelism": "pair-sharded x%d (contiguous ranges of the same pairs), table replicated, calls all-gathered" % world,
                          "ref_seed": hex(REF_SEED), "read_seed": hex(0x5C2B0030)},
               "result": {"pairs_classified_rank0": int(st["n_classified"]), "pairs_human_rank0": int((res["taxid"] == 9606).sum()), "pairs_human_total": human_total,
                          "probes": int(st["n_probes"]), "kmers": int(st["n_kmers"]), "overflow_units": int(st["n_overflow"])},
               "database": {"cells": info["capacity"], "occupied": info["size"], "reference_minimizer_runs": int(n_runs), "nodes": info["n_nodes"],
                            "build_s": round(t_db, 2)},
               "roofline": roof, "cpu_baseline": cpu,
               "external_oracle": (external_oracle_k2(db, d_reads, n_rec // 2, L, res["taxid"]) if world == 1 and not a.no_cpu else None)}
    db.close()
    return out if rank == 0 else None


def fastq_file(path, reads, mate, first_ordinal):
    """n x 150 uint8 reads -> FASTQ with headers `@syn.<ordinal, 9 digits> <mate>:N:0:0`, qualities `I`."""
    n, L = reads.shape
    head = b"@syn.000000000 %d:N:0:0\n" % mate
    row = np.frombuffer(head + b"A" * L + b"\n+\n" + b"I" * L + b"\n", dtype=np.uint8)
    arr = np.empty((n, row.size), dtype=np.uint8)
    arr[:] = row
    o = np.arange(first_ordinal, first_ordinal + n, dtype=np.int64)
    for d in range(9):
        arr[:, 5 + 8 - d] = 48 + (o % 10)
        o //= 10
    arr[:, len(head):len(head) + L] = reads
    arr.tofile(path)
    return arr.size


def main_e2e(a, rank, world, local, dev):
    """End-to-end scope of SURVEY.md §8d: FASTQ files in -> filtered FASTQ files + JSON report out (`scrubby reads`, rows
    a3-a8), through sh_reads_run.  Not the headline metric: the timed region includes file IO, parsing and the writer."""
    import shutil
    import tempfile
    assert world == 1, "--workload e2e is a single-process measurement"
    contigs = [1_000_000] * 5 if a.small else CHM13_CONTIGS
    n_rec = 200_000 if a.small else a.records
    n_pairs = n_rec // 2
    P, R = S.ref_params(REF_SEED, contigs), S.read_params(READ_SEED)
    G, L = P.genome_len, R.read_len
    opts = S.preset("sr")
    work = a.e2e_dir or tempfile.mkdtemp(prefix="scrubby_e2e_")
    os.makedirs(work, exist_ok=True)
    t0 = time.time()
    d_ref = torch.empty(G + 64, dtype=torch.uint8, device=dev)
    S.synth_ref_device(P, 0, G, d_ref)
    index = S.Index.build_device(d_ref, [P.contig_start[i] for i in range(len(contigs) + 1)], opts, device=local)
    h_ref = d_ref[:G].cpu().numpy()
    fa = os.path.join(work, "ref.fa")
    with open(fa, "wb") as f:
        for i in range(len(contigs)):
            f.write(b">ctg%d synthetic\n" % i)
            h_ref[P.contig_start[i]:P.contig_start[i + 1]].tofile(f)
            f.write(b"\n")
    del d_ref, h_ref
    d_reads = torch.empty(n_rec * L + 64, dtype=torch.uint8, device=dev)
    d_off = torch.empty(n_rec + 1, dtype=torch.int64, device=dev)
    S.synth_reads_device(P, R, 0, n_rec, d_reads, d_off)
    # expected flags from the device path, in 2 M-record pieces: a whole-batch context would hold (and then free) ~150 GB of HBM,
    # and the driver wipes freed VRAM before it is handed out again - the first sh_reads_run below would pay for that
    piece = min(n_rec, 2_000_000)
    ctx = S.Context(index, piece, piece * L, L)
    d_flags = torch.zeros(n_rec, dtype=torch.uint8, device=dev)
    for r0 in range(0, n_rec, piece):
        r1 = min(n_rec, r0 + piece)
        ctx.classify(d_reads[r0 * L:r1 * L], (d_off[r0:r1 + 1] - d_off[r0]).contiguous(), d_flags[r0:r1], None, want_stats=True)
    torch.cuda.synchronize()
    fl = d_flags.cpu().numpy().reshape(n_pairs, 2)
    expect_pairs = int(((fl[:, 0] == 1) | (fl[:, 1] == 1)).sum())      # HashSet union over both files (cleaner.rs:564-570)
    reads = d_reads[:n_rec * L].cpu().numpy().reshape(n_pairs, 2, L)
    r1, r2 = os.path.join(work, "R1.fastq"), os.path.join(work, "R2.fastq")
    in_bytes = fastq_file(r1, reads[:, 0, :], 1, 0) + fastq_file(r2, reads[:, 1, :], 2, 0)
    del reads, ctx, index, d_reads, d_flags, d_off
    torch.cuda.empty_cache()
    t_setup = time.time() - t0
    ext = ".fastq.gz" if a.e2e_gz else ".fastq"
    o1, o2, js = os.path.join(work, "clean_1" + ext), os.path.join(work, "clean_2" + ext), os.path.join(work, "report.json")
    threads = a.e2e_threads or usable_cores()

    def run():
        for f in (o1, o2, js):          # a fresh run writes new files; truncating the previous run's 1.6 GB outputs costs ~0.4 s of page-cache work
            if os.path.exists(f):
                os.remove(f)
        t = time.perf_counter()
        res = S.reads_run([r1, r2], [o1, o2], fa, json=js, command="scrubby reads -i R1 R2 -o clean_1 clean_2 -I ref.fa", threads=threads, device=local)
        return time.perf_counter() - t, res

    for _ in range(a.warmup):
        run()
    runs = [run() for _ in range(a.steps)]
    dt = sum(r[0] for r in runs)
    res = runs[-1][1]
    rep = json.load(open(js))
    ok = rep["reads_in"] == n_rec and rep["reads_removed"] == 2 * expect_pairs and res["n_depleted_ids"] == expect_pairs
    out_bytes = os.path.getsize(o1) + os.path.getsize(o2)
    mean = lambda k: round(float(np.mean([r[1][k] for r in runs])), 1)
    out = {
        "metric": "reads/s end to end (FASTQ files in -> filtered FASTQ files + JSON report; index built from FASTA excluded) - NOT the headline metric",
        "value": round(n_rec * a.steps / (dt - sum(r[1]["ms_index"] for r in runs) / 1e3), 1), "unit": "reads/s", "n_gpus": 1,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 1), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "configs[1] end to end: `scrubby reads -i R1 R2 -o O1 O2 -I ref.fa -j report.json` on %d synthetic 2x150bp pairs, sr preset" % n_pairs,
                   "records": n_rec, "reference_bp": int(G), "host_threads": threads, "gz_outputs": bool(a.e2e_gz),
                   "input_bytes": int(in_bytes), "output_bytes": int(out_bytes)},
        "stages_ms": {"index (FASTA -> HBM)": mean("ms_index"), "pass 1 (read + parse + classify + id set)": mean("ms_ingest"),
                      "of which device thread busy": mean("ms_classify"), "pass 2 (filter + write) + report": mean("ms_write")},
        "reads_per_s_incl_index": round(n_rec * a.steps / dt, 1),
        "runs_ms": [{"total": round(r[0] * 1e3, 1), "index": round(r[1]["ms_index"], 1), "pass1": round(r[1]["ms_ingest"], 1), "pass2": round(r[1]["ms_write"], 1)} for r in runs],
        "result": {"reads_in": rep["reads_in"], "reads_out": rep["reads_out"], "reads_removed": rep["reads_removed"],
                   "expected_removed (device flags, id union over mates)": 2 * expect_pairs, "identical": bool(ok)},
        "setup_s": round(t_setup, 1),
    }
    if a.e2e_legacy:
        os.environ["SCRUBBY_HIP_LEGACY_HOST"] = "1"
        t, lres = run()
        os.environ.pop("SCRUBBY_HIP_LEGACY_HOST")
        out["legacy_host_path"] = {"reads_per_s_excl_index": round(n_rec / (t - lres["ms_index"] / 1e3), 1), "s": round(t, 2),
                                   "ms_index": round(lres["ms_index"], 1), "ms_ingest": round(lres["ms_ingest"], 1),
                                   "ms_classify": round(lres["ms_classify"], 1), "ms_write": round(lres["ms_write"], 1),
                                   "reads_removed": lres["reads_removed"]}
    print(json.dumps(out), flush=True)
    if not a.e2e_dir:
        shutil.rmtree(work, ignore_errors=True)
    assert ok, "end-to-end result differs from the device flags"


def main_e2e_k2(a, rank, world, local, dev):
    """End-to-end scope of the taxid arm: `scrubby reads -c kraken2 -I DB -T Chordata -D 9606` (rows a9-a11, a4, a7, a8) through
    sh_kraken_run - database directory on disk, FASTQ pairs in, kraken.reads / kraken.report + filtered FASTQ + JSON out.
    Not the headline metric.  The table is 5e8 cells (2 GB) here so that writing and re-reading the database stays short."""
    import shutil
    import tempfile
    from scrubby_amd import k2 as K
    assert world == 1
    contigs = [1_000_000] * 5 if a.small else CHM13_CONTIGS
    n_rec = 200_000 if a.small else a.records
    n_rec -= n_rec & 1
    n_pairs = n_rec // 2
    cells = 12_000_017 if a.small else min(a.k2_cells, 500_000_009)
    P, R = S.ref_params(REF_SEED, contigs), S.read_params(0x5C2B0030)
    G, L = P.genome_len, R.read_len
    work = a.e2e_dir or tempfile.mkdtemp(prefix="scrubby_e2e_k2_")
    os.makedirs(work, exist_ok=True)
    t0 = time.time()
    parents, externals, names, ranks, ids = k2_taxonomy(2_000 if a.small else a.k2_nodes, 0x5C2B0030)
    db = K.K2Db.create(K.default_opts(), cells, parents, externals, names, ranks, device=local)
    d_ref = torch.empty(G + 64, dtype=torch.uint8, device=dev)
    S.synth_ref_device(P, 0, G, d_ref)
    # the table is too small for every minimizer of the 3.1 Gbp reference: the first 400 Mbp under Homo sapiens, filler to load 0.7
    g_ins = G if a.small else min(G, 400_000_000)
    db.insert_sequence_device(d_ref, g_ins, ids["Homo sapiens"])
    del d_ref
    fill = max(int(0.70 * cells) - db.info()["size"], 0)
    db.insert_random(0x5C2B0031, fill, ids["Bacteria"], len(parents) - 1)
    torch.cuda.synchronize()
    dbdir = os.path.join(work, "db")
    os.makedirs(dbdir, exist_ok=True)
    db.save(dbdir)
    d_reads = torch.empty(n_rec * L + 64, dtype=torch.uint8, device=dev)
    d_off = torch.empty(n_rec + 1, dtype=torch.int64, device=dev)
    S.synth_reads_device(P, R, 0, n_rec, d_reads, d_off)
    reads = d_reads[:n_rec * L].cpu().numpy().reshape(n_pairs, 2, L)
    r1, r2 = os.path.join(work, "R1.fastq"), os.path.join(work, "R2.fastq")
    in_bytes = fastq_file(r1, reads[:, 0, :], 1, 0) + fastq_file(r2, reads[:, 1, :], 2, 0)
    del reads, d_reads, d_off, db
    torch.cuda.empty_cache()
    t_setup = time.time() - t0
    o1, o2, js, wd = (os.path.join(work, x) for x in ("clean_1.fastq", "clean_2.fastq", "report.json", "work"))
    threads = a.e2e_threads or usable_cores()

    def run():
        for f in (o1, o2, js):
            if os.path.exists(f):
                os.remove(f)
        t = time.perf_counter()
        res = K.kraken_run([r1, r2], [o1, o2], dbdir, taxa=["Chordata"], taxa_direct=["9606"], workdir=wd, json=js, threads=threads, device=local,
                           command="scrubby reads -c kraken2 -i R1 R2 -o clean_1 clean_2 -I db -T Chordata -D 9606")
        return time.perf_counter() - t, res

    for _ in range(a.warmup):
        run()
    runs = [run() for _ in range(a.steps)]
    dt = sum(r[0] for r in runs)
    rep = json.load(open(js))
    out = {
        "metric": "reads/s end to end, taxid arm (FASTQ pairs + Kraken2-format database directory in -> kraken.reads / kraken.report + filtered FASTQ + JSON) - NOT the headline metric",
        "value": round(n_rec * a.steps / dt, 1), "unit": "reads/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(dt / a.steps * 1e3, 1), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "configs[4] end to end at reduced table size: %d synthetic 2x150bp pairs, table of %d cells (%.1f GB), -T Chordata -D 9606" % (n_pairs, cells, cells * 4 / 1e9),
                   "records": n_rec, "host_threads": threads, "input_bytes": int(in_bytes)},
        "runs_ms": [{"total": round(r[0] * 1e3, 1), "db_open": round(r[1]["ms_index"], 1), "ingest": round(r[1]["ms_ingest"], 1),
                     "classify": round(r[1]["ms_classify"], 1), "files + filter": round(r[1]["ms_write"], 1)} for r in runs],
        "result": {"reads_in": rep["reads_in"], "reads_out": rep["reads_out"], "reads_removed": rep["reads_removed"]},
        "setup_s": round(t_setup, 1),
    }
    if a.e2e_legacy:
        os.environ["SCRUBBY_HIP_LEGACY_HOST"] = "1"
        t, lres = run()
        os.environ.pop("SCRUBBY_HIP_LEGACY_HOST")
        out["legacy_host_path"] = {"reads_per_s": round(n_rec / t, 1), "s": round(t, 2), "reads_removed": lres["reads_removed"],
                                   "identical_counts": bool(lres["reads_removed"] == rep["reads_removed"] and lres["reads_out"] == rep["reads_out"])}
    print(json.dumps(out), flush=True)
    if not a.e2e_dir:
        shutil.rmtree(work, ignore_errors=True)


def reads_from_packed(d_packed, G, r_lo, n, L, host_pct, sub_per_10k, seed, d_out, d_off):
    """$SCRUBBY_CHM13: records [r_lo, r_lo + n) drawn from a REAL reference resident in HBM as 4-bit codes - host_pct % of the pairs
    are fragments of it (mate 1 forward at a hashed position, mate 2 the reverse complement 200 bases downstream, substitutions at
    sub_per_10k per 10^4 bases), the rest random bases; torch ops in pieces of 2^20 records."""
    dev = d_out.device
    acgt = torch.tensor([65, 67, 71, 84, 78], dtype=torch.uint8, device=dev)

    def mix(z):
        z = (z + 0x9E3779B97F4A7C15) & 0x7FFFFFFFFFFFFFFF
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B) & 0x7FFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111E) & 0x7FFFFFFFFFFFFFFF
        return z ^ (z >> 31)
    for c0 in range(0, n, 1 << 20):
        c1 = min(n, c0 + (1 << 20))
        rec = torch.arange(r_lo + c0, r_lo + c1, dtype=torch.int64, device=dev)
        pair, mate = rec >> 1, rec & 1
        h = mix(pair * 0x5851F42D4C957F2D ^ seed)
        host = (h % 100) < host_pct
        pos = (mix(h ^ 0x1234567) % max(G - L - 400, 1)) + mate * 200
        j = torch.arange(L, dtype=torch.int64, device=dev)[None, :]
        g = pos[:, None] + torch.where(mate[:, None] == 1, (L - 1) - j, j)
        code = ((d_packed[g >> 1].to(torch.int64) >> ((g & 1) * 4)) & 15).clamp(max=4)
        code = torch.where((mate[:, None] == 1) & (code < 4), 3 - code, code)
        hj = mix(rec[:, None] * 1000003 + j * 7919 + seed)
        sub = (hj % 10000) < sub_per_10k
        code = torch.where(sub & (code < 4), (code + 1 + (hj >> 20) % 3) % 4, code)
        code = torch.where(host[:, None], code, (hj >> 8) % 4)
        d_out[c0 * L:c1 * L] = acgt[code].reshape(-1)
    d_off.copy_(torch.arange(n + 1, dtype=torch.int64, device=dev) * L)


SR_DIV_SEED = 0x5C2B0040


def diverge_reads(d_reads, n, L, r_lo, dev, slab=2_000_000):
    """--workload sr-div: the headline's records made hard for the extension filter, in place on the device (seeded; same reads every run on
    the same hardware).  A third of the records each get 5 / 10 / 15 % of their bases replaced by a random base (on top of the generator's
    0.5 %), 2 % become chimeras (second half taken from another record), and 5 % get a 70-base stretch of another record, itself 15 %
    diverged, spliced into their middle (microbial reads sharing a diverged repeat family with the host, host reads with a foreign insert):
    chains that die in mm_filter_regs, regs[0] that does not survive, reads that need every chain."""
    reads = d_reads[:n * L].view(n, L)
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    g = torch.Generator(device=dev)
    g.manual_seed(SR_DIV_SEED + r_lo)
    rates = torch.tensor([0.05, 0.10, 0.15], device=dev)
    src = reads.clone()
    for s0 in range(0, n, slab):
        s1 = min(n, s0 + slab)
        m = s1 - s0
        idx = torch.arange(s0, s1, device=dev)
        blk = reads[s0:s1]
        u = torch.rand((m, L), device=dev, generator=g)
        rnd = acgt[torch.randint(0, 4, (m, L), device=dev, generator=g)]
        sub = u < rates[(idx % 3)].unsqueeze(1)
        blk[sub] = rnd[sub]
        sel = torch.rand(m, device=dev, generator=g)
        chim = sel < 0.02
        other = (idx + 2 * 100_003) % n
        blk[chim, L // 2:] = src[other[chim], L // 2:]
        ins = (sel >= 0.02) & (sel < 0.07)
        seg = src[(idx + 2 * 7919 + 1) % n][:, 40:110].clone()
        segsub = torch.rand((m, 70), device=dev, generator=g) < 0.15
        seg[segsub] = acgt[torch.randint(0, 4, (m, 70), device=dev, generator=g)][segsub]
        blk[ins, 40:110] = seg[ins]
    del src
    torch.cuda.synchronize()


def ctx_chunk(a, n_rec):
    return min(a.chunk, n_rec)


def long_read_lengths(seed, r0, n):
    """syn_long_len of csrc/sh_synth_core.h in numpy (uint64 wrap-around arithmetic)."""
    M = np.uint64
    with np.errstate(over="ignore"):
        def mix(z):
            z = z + M(0x9E3779B97F4A7C15)
            z = (z ^ (z >> M(30))) * M(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> M(27))) * M(0x94D049BB133111EB)
            return z ^ (z >> M(31))
        r = np.arange(r0, r0 + n, dtype=np.uint64)
        h = mix(M(seed) ^ M(0x10E6A11) ^ (r * M(0x9E3779B97F4A7C15)))
    k1 = np.array([200, 1674, 2190, 2633, 3056, 3480, 3920, 4389, 4900, 5470, 6124, 6899, 7857, 9118, 10963, 14341, 14341], dtype=np.uint64)
    k2 = np.array([14341, 14672, 15029, 15415, 15838, 16302, 16817, 17395, 18050, 18806, 19697, 20774, 22131, 23941, 26617, 31541, 31541], dtype=np.uint64)
    k3 = np.array([31541, 32023, 32543, 33106, 33721, 34397, 35146, 35985, 36936, 38032, 39321, 40878, 42834, 45437, 49268, 56273, 100000], dtype=np.uint64)
    q1, q2, q3 = (h & M(15)).astype(np.int64), ((h >> M(4)) & M(15)).astype(np.int64), ((h >> M(32)) & M(15)).astype(np.int64)
    lo = np.where(q1 < 15, k1[q1], np.where(q2 < 15, k2[q2], k3[q3]))
    hi = np.where(q1 < 15, k1[np.minimum(q1 + 1, 16)], np.where(q2 < 15, k2[np.minimum(q2 + 1, 16)], k3[q3 + 1]))
    return (lo + ((((h >> M(8)) & M(0xffffff)) * (hi - lo)) >> M(24))).astype(np.uint32)


def physical_cores():
    """Physical cores of this host (unique (package, core id) pairs); the logical count if /proc/cpuinfo does not say."""
    try:
        seen, pkg = set(), "0"
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("physical id"):
                pkg = ln.split(":")[1].strip()
            elif ln.startswith("core id"):
                seen.add((pkg, ln.split(":")[1].strip()))
        if seen:
            return min(len(seen), len(os.sched_getaffinity(0)))
    except Exception:
        pass
    return len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)


def attach_traffic(roofline, workload, stage, records_per_launch, skip=False):
    """roofline.traffic from profiles/traffic[_<workload>].json (scripts/make_traffic.py: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this
    command), only if it was measured on the kernel sources this library was built from and at this launch size."""
    name = "traffic.json" if workload == "sr" else f"traffic_{workload}.json"
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", name)
    if skip or not os.path.exists(path):
        return
    try:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "scripts"))
        from make_traffic import source_hash
        tj = json.load(open(path))
        if tj.get("source_sha1") != source_hash(workload):
            roofline["traffic_source"] = f"profiles/{name} is stale (measured on other kernel sources): refused"
        elif tj.get("records_per_launch") != records_per_launch:
            roofline["traffic_source"] = f"profiles/{name} was measured at {tj.get('records_per_launch')} records per launch, this run uses {records_per_launch}: refused"
        elif stage in tj.get("stages", {}):
            roofline["traffic"] = tj["stages"][stage]["hbm_bytes_per_launch"]
            roofline["traffic_source"] = f"profiles/{name} (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE, raw; same kernel sources: sha1 " + tj["source_sha1"][:12] + ")"
    except Exception as ex:
        roofline["traffic_source"] = f"profiles/{name} unreadable: {ex}"


def cpu_quota():
    """CPUs this process may actually use: the cgroup CPU quota (cpu.max of cgroup v2, cfs_quota of v1) if there is one - a GPU box
    shows all 256 logical CPUs of its host but grants 16 - else None."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return max(1, int(int(q) / int(p)))
    except Exception:
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return max(1, q // p)
    except Exception:
        pass
    return None


def usable_cores():
    """threads worth starting: physical cores, capped by the cgroup quota (more threads than the quota only time-slice)"""
    q = cpu_quota()
    return min(physical_cores(), q) if q else physical_cores()


def cpu_baseline(index, info, d_reads, n_rec, L, seconds, d_flags, d_off=None, preset="sr", chain_only=False):
    """The oracle (restated decision path incl. the extension stage; NOT minimap2-rs) on the host cores: ONE call over a large
    contiguous sample (threads pull 64-read chunks off an atomic counter, so every core works to the end), sized from a short
    calibration call to take about `seconds`.  Also a parity check: flags differing from the GPU's on the sample."""
    from oracle import oracle as O
    cores, logical = usable_cores(), os.cpu_count() or 1
    try:
        avail_gb = int(open("/proc/meminfo").read().split("MemAvailable:")[1].split()[0]) / 1e6
    except Exception:
        avail_gb = 0.0
    need_gb = (info["n_slots"] * 16 + info["n_positions"] * 8 + info["n_bases"] / 2) / 1e9 * 1.1 + 2
    if avail_gb and avail_gb < need_gb:
        return {"value": None, "unit": "reads/s", "cores": cores, "kind": "port",
                "sample": f"skipped: host has {avail_gb:.0f} GB free, index copy needs {need_gb:.0f} GB"}
    slots, pos = index.export()
    oidx = O.Index.wrap(slots, pos, info["w"], info["k"], ref=index.export_ref())      # the reference too: the extension stage aligns against it
    po = O.preset(preset)
    if chain_only:
        po.flags &= ~1      # MMO_F_CIGAR
    oo = oidx.update_opts(po)
    off_all = d_off.cpu().numpy().astype(np.uint64) if d_off is not None else None

    def run(first, count, threads):
        if off_all is None:
            reads = d_reads[first * L:(first + count) * L].cpu().numpy()
            off = np.arange(count + 1, dtype=np.uint64) * L
        else:
            b0, b1 = int(off_all[first]), int(off_all[first + count])
            reads = d_reads[b0:b1].cpu().numpy()
            off = off_all[first:first + count + 1] - np.uint64(b0)
        t0 = time.perf_counter()
        fl, _ = oidx.classify(oo, reads, off, threads=threads, want_trace=False)
        dt = time.perf_counter() - t0
        return dt, int((fl != d_flags[first:first + count].cpu().numpy()).sum())

    n_cal = min(n_rec, 200_000 if off_all is None else (4_000 if chain_only else 600))      # long reads with the extension stage: ~20 reads/s per core
    dt_cal, _ = run(0, n_cal, cores)                                    # calibration (also warms the index pages)
    n_s = int(min(n_rec, max(n_cal, n_cal / max(dt_cal, 1e-6) * seconds)))
    dt, mism = run(0, n_s, cores)
    return {"value": round(n_s / dt, 1), "unit": "reads/s", "cores": cores, "kind": "port",
            "reads_per_s_per_thread": round(n_s / dt / cores, 1), "logical_cpus": logical, "physical_cores": physical_cores(), "cgroup_cpu_quota": cpu_quota(),
            "sample": f"first {n_s} records of the same batch in ONE call, same index and reference (copied from HBM), {cores} threads (one per usable core: physical cores capped by the cgroup CPU quota), {dt:.1f} s; "
                      f"restatement baseline - not minimap2-rs; flags differing from the GPU on the sample: {mism}"}


def stratified_parity(index, ctx, info, d_reads, d_flags, n_rec, L, n_random=300_000, seed=20261004):
    """Flags of the GPU against the oracle where a bug would live: EVERY read of the batch that was re-chained with max_occ, every read whose
    regs[0] had to be aligned base by base, every read that fell back to the complete procedure (sh_ctx_debug_list), plus a random sample
    of the rest.  The batch must have been classified by ONE launch of `ctx` (the lists describe its last chunk)."""
    from oracle import oracle as O
    strata = {}
    names = ("rechained_max_occ", "regs0_aligned", "full_fallback")
    for which, nm in enumerate(names):
        strata[nm] = np.unique(ctx.debug_list(which).astype(np.int64))
        strata[nm] = strata[nm][strata[nm] < n_rec]
    special = np.unique(np.concatenate(list(strata.values()))) if strata else np.zeros(0, np.int64)
    rng = np.random.default_rng(seed)
    rest = rng.choice(n_rec, size=min(n_random, n_rec), replace=False).astype(np.int64)
    rest = np.setdiff1d(rest, special)
    pick = np.concatenate([special, rest])
    if len(pick) == 0:
        return None
    idx = torch.from_numpy(pick).to(d_reads.device)
    rows = d_reads[:n_rec * L].view(n_rec, L)[idx].cpu().numpy().reshape(-1)
    gflags = d_flags[idx].cpu().numpy()
    slots, pos = index.export()
    oidx = O.Index.wrap(slots, pos, info["w"], info["k"], ref=index.export_ref())
    oo = oidx.update_opts(O.preset("sr"))
    t0 = time.perf_counter()
    of, _ = oidx.classify(oo, rows, np.arange(len(pick) + 1, dtype=np.uint64) * L, threads=usable_cores(), want_trace=False)
    dt = time.perf_counter() - t0
    bad = of != gflags
    out = {"reads_checked": int(len(pick)), "flags_differing": int(bad.sum()), "oracle_s": round(dt, 1),
           "strata": {nm: {"reads": int(len(v)), "flags_differing": int(bad[np.isin(pick, v)].sum())} for nm, v in strata.items()},
           "random_rest": {"reads": int(len(rest)), "flags_differing": int(bad[len(special):].sum())}}
    return out


LONG_STRATA = ((3, "rmq_exact_tree"), (4, "rmq_tie_left_open"), (5, "ext_unresolved"), (6, "locus_redone"), (7, "memory_on_demand"),
               (8, "probe_undecided_full_procedure"), (9, "second_working_memory_size"), (10, "one_lane_trees"))


def stratified_parity_long(index, ctx, info, d_reads, d_off, d_flags, n_rec, preset, seconds, seed=20261005, cap_per_stratum=None):
    """The long-read form of stratified_parity: the oracle over EVERY read of the batch (the last call of `ctx`: sh_ctx_debug_list 3..10) whose
    answer came by one of the extension stage's rarer paths - long join redone on the literal tree, a tie left open, left at the chain-level
    answer, redone with every anchor, memory on demand, probe undecided, second working-memory size, one-lane trees - plus a random rest that
    fills `seconds` of CPU time.  A stratum larger than cap_per_stratum is sampled (and says so)."""
    from oracle import oracle as O
    if cap_per_stratum is None:      # the dedicated run (--cpu-seconds >= 15) takes every read of a stratum up to 4000; the headline's secondary line a sample of 150
        cap_per_stratum = 4000 if seconds >= 15 else 150
    rng = np.random.default_rng(seed)
    strata, sampled = {}, {}
    for which, nm in LONG_STRATA:
        v = np.unique(ctx.debug_list(which).astype(np.int64))
        v = v[v < n_rec]
        sampled[nm] = int(len(v))
        if len(v) > cap_per_stratum:
            v = np.sort(rng.choice(v, size=cap_per_stratum, replace=False))
        strata[nm] = v
    special = np.unique(np.concatenate(list(strata.values()))) if strata else np.zeros(0, np.int64)
    off_all = d_off.cpu().numpy().astype(np.int64)
    slots, pos = index.export()
    oidx = O.Index.wrap(slots, pos, info["w"], info["k"], ref=index.export_ref())
    oo = oidx.update_opts(O.preset(preset))
    cores = usable_cores()

    def run(pick):
        lens = off_all[pick + 1] - off_all[pick]
        o = np.zeros(len(pick) + 1, dtype=np.uint64)
        o[1:] = np.cumsum(lens)
        rows = torch.cat([d_reads[off_all[r]:off_all[r + 1]] for r in pick]).cpu().numpy() if len(pick) else np.zeros(0, np.uint8)
        t0 = time.perf_counter()
        of, _ = oidx.classify(oo, rows, o, threads=cores, want_trace=False)
        return of, time.perf_counter() - t0

    out = {"strata": {}, "flags_differing": 0, "reads_checked": 0}
    gflags = d_flags.cpu().numpy()
    t_special = 0.0
    if len(special):
        of, t_special = run(special)
        bad = of != gflags[special]
        out["flags_differing"] += int(bad.sum()); out["reads_checked"] += int(len(special))
        for nm, v in strata.items():
            out["strata"][nm] = {"reads_in_batch": sampled[nm], "reads_checked": int(len(v)), "flags_differing": int(bad[np.isin(special, v)].sum())}
    else:
        for nm in strata:
            out["strata"][nm] = {"reads_in_batch": 0, "reads_checked": 0, "flags_differing": 0}
    # the random rest: sized by what the special reads cost per read (they are the expensive ones: an upper bound), at least 2000 reads
    left = max(seconds - t_special, 3.0)
    per = (t_special / len(special)) if len(special) else 0.01
    n_rest = int(min(n_rec, max(2000, left / max(per, 1e-4) * 4)))
    rest = np.setdiff1d(np.sort(rng.choice(n_rec, size=min(n_rest, n_rec), replace=False)).astype(np.int64), special)
    of, t_rest = run(rest)
    bad = of != gflags[rest]
    out["flags_differing"] += int(bad.sum()); out["reads_checked"] += int(len(rest))
    out["random_rest"] = {"reads": int(len(rest)), "flags_differing": int(bad.sum())}
    out["oracle_s"] = round(t_special + t_rest, 1)
    return out


def external_oracle(index, d_reads, n_rec, L, d_flags, contigs, P, dev, real_ref=None, max_reads=400_000, preset="sr", d_off=None):
    """if a `minimap2` binary exists on this box, map a bounded sample single-end with CIGAR on (`-c -x sr`, or
    `-c -x map-ont` for the long reads: the reference maps every record on its own with `.with_cigar()`, cleaner.rs:457,473,499-501,
    527-529,552) and diff the mapped-read set with the GPU flags - the only route by which parity can become pinned.  Absent: says so.
    d_off (long reads): the records' offsets; the sample is then bounded by bases as well."""
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("minimap2")
    if exe is None:
        return {"minimap2": "absent", "kraken2": "present" if shutil.which("kraken2") else "absent"}
    n = min(n_rec, max_reads)
    work = tempfile.mkdtemp(prefix="scrubby_mm2_")
    try:
        fa = real_ref
        if fa is None:
            G = P.genome_len
            d_ref = torch.empty(G + 64, dtype=torch.uint8, device=dev)
            S.synth_ref_device(P, 0, G, d_ref)
            h_ref = d_ref[:G].cpu().numpy()
            del d_ref
            fa = os.path.join(work, "ref.fa")
            with open(fa, "wb") as f:
                for i in range(len(contigs)):
                    f.write(b">ctg%d\n" % i)
                    h_ref[P.contig_start[i]:P.contig_start[i + 1]].tofile(f)
                    f.write(b"\n")
        fq = os.path.join(work, "reads.fq")
        if d_off is None:
            reads = d_reads[:n * L].cpu().numpy().reshape(n, L)
            fastq_file(fq, reads, 1, 0)
        else:
            off = d_off[:n + 1].cpu().numpy().astype(np.int64)
            n = int(min(n, max(1, np.searchsorted(off, 400_000_000) - 1)))       # at most 0.4 Gbases of long reads
            h = d_reads[:int(off[n])].cpu().numpy()
            with open(fq, "wb") as f:
                for i in range(n):
                    f.write(b"@syn.%09d 1:N:0:0\n" % i)
                    h[off[i]:off[i + 1]].tofile(f)
                    f.write(b"\n+\n" + b"I" * int(off[i + 1] - off[i]) + b"\n")
        t0 = time.perf_counter()
        out = subprocess.run([exe, "-c", "-x", preset, "-t", str(usable_cores()), fa, fq], capture_output=True, check=True).stdout
        dt = time.perf_counter() - t0
        mapped = np.zeros(n, dtype=np.uint8)
        for ln in out.splitlines():
            mapped[int(ln.split(b"\t", 1)[0][4:])] = 1
        gpu = (d_flags[:n].cpu().numpy() == 1).astype(np.uint8)
        return {"minimap2": "present", "version": subprocess.run([exe, "--version"], capture_output=True).stdout.decode().strip(), "preset": preset,
                "sample_reads": n, "mapped_by_minimap2": int(mapped.sum()), "mapped_by_gpu": int(gpu.sum()), "differ": int((mapped != gpu).sum()),
                "reads_per_s_incl_index": round(n / dt, 1), "threads": usable_cores()}
    except Exception as e:      # the real tool is an extra: its failure must not fail the bench
        return {"minimap2": "present", "error": repr(e)[:300]}
    finally:
        shutil.rmtree(work, ignore_errors=True)


def external_oracle_k2(db, d_reads, n_pairs, L, gpu_calls, max_pairs=400_000):
    """The Kraken2 arm's real-tool hook: if a `kraken2` binary exists, write the synthetic database out in Kraken 2's own files (hash.k2d,
    opts.k2d, taxo.k2d: sh_k2_save), classify a bounded sample of the pairs with `kraken2 --paired` (the command of cleaner.rs:300-323) and
    diff the per-pair taxid calls with the GPU's.  Absent: says so."""
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("kraken2")
    if exe is None:
        return {"kraken2": "absent"}
    n = int(min(n_pairs, max_pairs))
    work = tempfile.mkdtemp(prefix="scrubby_k2_")
    try:
        db.save(work)
        reads = d_reads[:2 * n * L].cpu().numpy().reshape(2 * n, L)
        f1, f2 = os.path.join(work, "r1.fq"), os.path.join(work, "r2.fq")
        fastq_file(f1, reads[0::2], 1, 0)
        fastq_file(f2, reads[1::2], 2, 0)
        t0 = time.perf_counter()
        out = subprocess.run([exe, "--threads", str(usable_cores()), "--db", work, "--paired", f1, f2, "--output", "-"], capture_output=True, check=True).stdout
        dt = time.perf_counter() - t0
        calls = np.zeros(n, dtype=np.int64)
        for ln in out.splitlines():
            c = ln.split(b"\t")
            calls[int(c[1][4:13])] = int(c[2])
        g = np.asarray(gpu_calls[:n]).astype(np.int64)
        return {"kraken2": "present", "sample_pairs": n, "classified_by_kraken2": int((calls != 0).sum()), "classified_by_gpu": int((g != 0).sum()),
                "differ": int((calls != g).sum()), "pairs_per_s_incl_db_load": round(n / dt, 1), "threads": usable_cores()}
    except Exception as e:
        return {"kraken2": "present", "error": repr(e)[:300]}
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
