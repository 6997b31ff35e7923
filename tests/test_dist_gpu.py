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
