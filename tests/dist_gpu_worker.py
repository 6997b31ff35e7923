"""One rank of the read-sharded path on a GPU box (started by tests/test_dist_gpu.py, one process per rank).

Every rank builds its replica of the index, classifies ITS contiguous pair-aligned shard of the cfg1-mini records
through the C ABI (sh_classify_device), and the depleted-record bitmaps are united with one all_gather
(scrubby_amd/dist.py; /root/reference/src/cleaner.rs:546-559 data-parallel map, :564-570 set union).  The ranks of
this test share device 0 (a one-GPU box), so the collective runs over gloo on CPU copies of the 1-bit-per-record
bitmap; with one device per rank the same code takes RCCL (bench.py --gpus N).

Rank 0 then classifies ALL records alone, runs the CPU oracle on them, and writes what it found as JSON.
usage: dist_gpu_worker.py OUT.json N_RECORDS
"""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_path, n_records = sys.argv[1], int(sys.argv[2])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from scrubby_amd import dist as D
    from scrubby_amd import lib as S
    from tests import workloads as W
    S.require_gpu()
    torch.cuda.set_device(0)
    P, R = S.ref_params(W.CFG1_REF_SEED, W.CFG1_CONTIGS), S.read_params(W.CFG1_READ_SEED)
    G, L = P.genome_len, R.read_len
    opts = S.preset("sr")
    d_ref = torch.empty(G + 64, dtype=torch.uint8, device="cuda")
    S.synth_ref_device(P, 0, G, d_ref)
    index = S.Index.build_device(d_ref, [P.contig_start[i] for i in range(len(W.CFG1_CONTIGS) + 1)], opts)

    def classify(lo, hi):
        n = hi - lo
        d_reads = torch.empty(n * L + 64, dtype=torch.uint8, device="cuda")
        d_off = torch.empty(n + 1, dtype=torch.int64, device="cuda")
        S.synth_reads_device(P, R, lo, n, d_reads, d_off)           # records [lo, hi) of the global record space
        flags = torch.zeros(n, dtype=torch.uint8, device="cuda")
        S.Context(index, n, n * L, L).classify(d_reads[:n * L], d_off, flags, None, want_stats=False)
        torch.cuda.synchronize()
        return flags, d_reads

    lo, hi = D.shard_range(n_records, rank, world)
    mine, _ = classify(lo, hi)
    gathered, sb = D.union_depleted(mine, via_host=True)
    union = D.gathered_to_flags(gathered, sb, n_records, world).numpy()
    counts = D.sum_counters([hi - lo, int((mine == 1).sum().item())], "cpu")
    every = [None] * world
    dist.all_gather_object(every, int(union.sum()))
    if rank == 0:
        alone, d_reads = classify(0, n_records)
        alone = (alone.cpu().numpy() == 1).astype(np.uint8)
        from oracle import oracle as O
        slots, pos = index.export()
        oidx = O.Index.wrap(slots, pos, 11, 21, ref=index.export_ref())
        reads = d_reads[:n_records * L].cpu().numpy()
        of, _ = oidx.classify(O.preset("sr"), reads, np.arange(n_records + 1, dtype=np.uint64) * L, threads=4, want_trace=False)
        res = {"world": world, "shards": [list(D.shard_range(n_records, r, world)) for r in range(world)],
               "union_equals_single_rank": bool(np.array_equal(union, alone)),
               "union_equals_oracle": bool(np.array_equal(union, (of == 1).astype(np.uint8))),
               "same_union_on_every_rank": len(set(every)) == 1,
               "counters": counts, "depleted": int(union.sum()), "records": n_records}
        with open(out_path, "w") as f:
            json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
