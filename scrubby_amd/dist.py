"""Multi-GPU plumbing for the read-sharded path.

One process per GPU; the index is replicated, reads are sharded by contiguous record ranges and there is no
data-path collective.  The single exchange is the union of the depleted-record set at the end
(`HashSet` union at /root/reference/src/cleaner.rs:564-570): each rank packs its flags to 1 bit per record
and the disjoint slices are all-gathered (RCCL has no bitwise-OR reduction; disjoint slices avoid needing one).
Works on CUDA tensors over `nccl` (= RCCL on ROCm) and on CPU tensors over `gloo` (tests).
"""
import torch
import torch.distributed as dist


def shard_range(n_records, rank, world):
    """Contiguous, pair-aligned record range [lo, hi) of `rank` (records 2p, 2p+1 are mates)."""
    pairs = (n_records + 1) // 2
    per = (pairs + world - 1) // world
    lo = min(rank * per * 2, n_records)
    hi = min((rank + 1) * per * 2, n_records)
    return lo, hi


def pack_flags(flags):
    """uint8 flags (1 = host) -> little-endian bitmap, 1 bit per record."""
    n = flags.numel()
    if flags.is_cuda and flags.dtype == torch.uint8 and flags.is_contiguous():      # the library's ballot kernel (csrc/sh_api.hip)
        from scrubby_amd import lib as S
        return S.pack_flags_device(flags)
    pad = (-n) % 8
    b = (flags == 1).to(torch.uint8)
    if pad:
        b = torch.cat([b, torch.zeros(pad, dtype=torch.uint8, device=flags.device)])
    w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.int32, device=flags.device)
    return (b.view(-1, 8).to(torch.int32) * w).sum(dim=1).to(torch.uint8)


def unpack_flags(bits, n):
    w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.int32, device=bits.device)
    return ((bits.to(torch.int32).unsqueeze(1) & w) != 0).to(torch.uint8).reshape(-1)[:n]


def union_depleted(flags, slice_bytes=None, group=None, via_host=False):
    """All ranks contribute the bitmap of their own record slice; every rank receives all slices.

    flags: this rank's uint8 flags.  slice_bytes: common slice size in bytes (max over ranks); computed with
    an all_reduce(max) if omitted.  via_host: device flags, but a CPU collective (gloo) - the bitmap is packed on
    the device and its 1 bit per record crosses PCIe (ranks sharing one device, where RCCL cannot be used).
    Returns (gathered uint8 [world * slice_bytes], slice_bytes)."""
    bits = pack_flags(flags)
    if via_host:
        bits = bits.cpu()
    world = dist.get_world_size(group)
    if slice_bytes is None:
        m = torch.tensor([bits.numel()], dtype=torch.int64, device=bits.device)
        dist.all_reduce(m, op=dist.ReduceOp.MAX, group=group)
        slice_bytes = int(m.item())
    if bits.numel() < slice_bytes:
        bits = torch.cat([bits, torch.zeros(slice_bytes - bits.numel(), dtype=torch.uint8, device=bits.device)])
    out = [torch.empty(slice_bytes, dtype=torch.uint8, device=bits.device) for _ in range(world)]
    dist.all_gather(out, bits, group=group)
    return torch.cat(out), slice_bytes


def gathered_to_flags(gathered, slice_bytes, n_records, world):
    """The all-gathered slices back to one uint8 vector over the global record space (1 = depleted)."""
    parts = []
    for r in range(world):
        lo, hi = shard_range(n_records, r, world)
        parts.append(unpack_flags(gathered[r * slice_bytes:(r + 1) * slice_bytes], hi - lo))
    return torch.cat(parts)


def gather_calls(calls, slice_len=None, group=None, via_host=False):
    """The Kraken2-style arm (run_kraken, /root/reference/src/cleaner.rs:288-330): every rank classified a contiguous range of the pairs;
    the per-pair taxid calls (uint32, one per pair - what kraken.reads lists and what the report counts) are all-gathered, so that
    every rank holds the calls of the whole job.  calls: this rank's int32/uint32 vector.  Returns (gathered [world * slice_len], slice_len)."""
    c = calls.contiguous().view(-1)
    if via_host:
        c = c.cpu()
    world = dist.get_world_size(group)
    if slice_len is None:
        m = torch.tensor([c.numel()], dtype=torch.int64, device=c.device)
        dist.all_reduce(m, op=dist.ReduceOp.MAX, group=group)
        slice_len = int(m.item())
    if c.numel() < slice_len:
        c = torch.cat([c, torch.zeros(slice_len - c.numel(), dtype=c.dtype, device=c.device)])
    out = [torch.empty(slice_len, dtype=c.dtype, device=c.device) for _ in range(world)]
    dist.all_gather(out, c, group=group)
    return torch.cat(out), slice_len


def gathered_to_calls(gathered, slice_len, n_records, world):
    """The all-gathered call slices back to one vector over the job's pairs (records 2p, 2p + 1 = pair p)."""
    parts = []
    for r in range(world):
        lo, hi = shard_range(n_records, r, world)
        parts.append(gathered[r * slice_len:r * slice_len + (hi - lo) // 2])
    return torch.cat(parts)


def sum_counters(values, device, group=None):
    """all_reduce(sum) of a few int64 counters ({records, depleted, ...})."""
    t = torch.tensor(list(values), dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return [int(v) for v in t.tolist()]
