"""The depleted-bitmap union (scrubby_amd/dist.py) on CUDA tensors over the `nccl` backend (= RCCL on ROCm).  A one-GPU box only
allows a single-rank group, which still runs the device-side packing and the collective call path bench.py uses at N > 1
(the world-size-2 semantics are covered over gloo in tests/test_dist_cpu.py)."""
import os

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]


def test_union_on_device_tensors_single_rank_nccl():
    import torch
    import torch.distributed as dist
    from scrubby_amd import dist as D
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        rng = np.random.default_rng(7)
        n = 1_000_003
        flags = (rng.random(n) < 0.5).astype(np.uint8)
        flags[::101] = 2
        d = torch.from_numpy(flags).cuda()
        assert np.array_equal(D.pack_flags(d).cpu().numpy(), D.pack_flags(torch.from_numpy(flags)).numpy())      # ballot kernel == torch ops
        gathered, sb = D.union_depleted(d, slice_bytes=(n + 7) // 8)
        assert gathered.is_cuda and gathered.numel() == sb == (n + 7) // 8
        back = D.unpack_flags(gathered, n).cpu().numpy()
        assert np.array_equal(back, (flags == 1).astype(np.uint8))
        assert D.sum_counters([n, int((flags == 1).sum())], d.device) == [n, int((flags == 1).sum())]
    finally:
        dist.destroy_process_group()


def test_two_ranks_classify_their_shards_and_unite(tmp_path):
    """BASELINE configs[2] in miniature: 2 processes (device 0 shared, so gloo carries the union), each classifying its
    shard_range of the cfg1-mini records; the gathered bitmap must equal the single-rank flags and the oracle's."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    out = tmp_path / "res.json"
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "tests", "dist_gpu_worker.py"), str(out), "20001"], env=env))
    for p in procs:
        assert p.wait(timeout=280) == 0
    res = json.load(open(out))
    assert res["shards"] == [[0, 10002], [10002, 20001]]
    assert res["union_equals_single_rank"] and res["union_equals_oracle"] and res["same_union_on_every_rank"]
    assert res["counters"] == [20001, res["depleted"]] and 0 < res["depleted"] < 20001


def test_bench_spawns_its_ranks(tmp_path):
    """`bench.py --gpus 2` starts two ranks itself and reports n_gpus 2 with the same records cut in two (strong scaling)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--small", "--steps", "2", "--warmup", "1", "--no-cpu"],
                       env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["config"]["records_total"] == 200_000 and j["config"]["records_rank0"] == 100_000
    assert j["weak_scaling"]["records_per_gpu"] == 200_000 and j["union_bytes_gathered"] == 2 * 12500
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--small", "--steps", "2", "--warmup", "1", "--no-cpu", "--no-secondary"],
                         env=env, capture_output=True, text=True, timeout=280)
    j1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][-1])
    assert j1["n_gpus"] == 1 and j1["result"]["reads_removed"] == j["result"]["reads_removed"]


def test_k2_bench_is_strong_scaled_over_its_ranks(tmp_path):
    """BASELINE configs[4] in miniature: `bench.py --workload k2 --gpus 2` cuts the SAME pairs in two (the ranks share device 0 here, so
    gloo carries the calls), gathers every pair's call inside the timed region and finds the human pairs the one-rank run finds."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    runs = {}
    for n in (1, 2):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "k2", "--gpus", str(n), "--small", "--steps", "2", "--warmup", "1", "--no-cpu"],
                           env=env, capture_output=True, text=True, timeout=280)
        assert r.returncode == 0, r.stderr[-2000:]
        runs[n] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    j1, j2 = runs[1], runs[2]
    assert j2["n_gpus"] == 2 and j2["scaling"] == "strong" and j2["config"]["records_total"] == 200_000 and j2["config"]["records_rank0"] == 100_000
    assert j2["calls_gathered_bytes"] == 2 * 50_000 * 4
    assert j1["result"]["pairs_human_total"] == j2["result"]["pairs_human_total"] > 10_000
