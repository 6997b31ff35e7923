"""World-size-2 gloo test of the only exchange step of the sharded path: the depleted-bitmap union."""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, n_records, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from scrubby_amd import dist as D
    rng = np.random.default_rng(1234)
    all_flags = (rng.random(n_records) < 0.5).astype(np.uint8)
    all_flags[::97] = 2                                   # "empty read" flags must not count as depleted
    lo, hi = D.shard_range(n_records, rank, world)
    mine = torch.from_numpy(all_flags[lo:hi].copy())
    gathered, sb = D.union_depleted(mine)
    parts = []
    for r in range(world):
        l2, h2 = D.shard_range(n_records, r, world)
        parts.append(D.unpack_flags(gathered[r * sb:(r + 1) * sb], h2 - l2))
    union = torch.cat(parts).numpy()
    ok = np.array_equal(union, (all_flags == 1).astype(np.uint8))
    tot = D.sum_counters([hi - lo, int((mine == 1).sum())], "cpu")
    ok = ok and tot == [n_records, int((all_flags == 1).sum())]
    ret[rank] = bool(ok)
    dist.destroy_process_group()


def test_bitmap_union_world2():
    world, n = 2, 100_003          # odd count: ragged last pair / padding bits
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_worker, args=(world, 29517, n, ret), nprocs=world, join=True)
        assert dict(ret) == {0: True, 1: True}


def test_shard_ranges_are_pair_aligned_and_cover():
    from scrubby_amd import dist as D
    for n in (0, 1, 2, 7, 20_000_000, 19_999_999):
        for world in (1, 2, 3, 8):
            prev = 0
            for r in range(world):
                lo, hi = D.shard_range(n, r, world)
                assert lo == prev and (lo % 2 == 0 or lo == n)
                prev = hi
            assert prev == n


def test_pack_unpack_roundtrip():
    from scrubby_amd import dist as D
    f = torch.tensor([1, 0, 2, 1, 1, 0, 0, 1, 1, 0, 1], dtype=torch.uint8)
    b = D.pack_flags(f)
    assert b.tolist() == [0b10011001, 0b101]
    assert D.unpack_flags(b, 11).tolist() == [1, 0, 0, 1, 1, 0, 0, 1, 1, 0, 1]


def _calls_worker(rank, world, port, n_records, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from scrubby_amd import dist as D
    rng = np.random.default_rng(99)
    n_pairs = n_records // 2
    all_calls = rng.integers(0, 50_000, n_pairs).astype(np.int32)
    lo, hi = D.shard_range(n_records, rank, world)
    res = np.zeros(((hi - lo) // 2, 4), np.int32)              # the classifier's result rows: column 0 = the call
    res[:, 0] = all_calls[lo // 2:hi // 2]
    res[:, 1:] = 7
    gathered, sl = D.gather_calls(torch.from_numpy(res)[:, 0], D.shard_range(n_records, 0, world)[1] // 2)
    back = D.gathered_to_calls(gathered, sl, n_records, world).numpy()
    ret[rank] = bool(np.array_equal(back, all_calls)) and sl * world >= n_pairs
    dist.destroy_process_group()


def test_taxid_calls_gathered_world2():
    """The exchange of the Kraken2-style arm at N > 1 (bench.py --workload k2 --gpus N): contiguous pair shards, calls all-gathered."""
    world, n = 2, 100_006
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_calls_worker, args=(world, 29519, n, ret), nprocs=world, join=True)
        assert dict(ret) == {0: True, 1: True}
