"""In-process multi-GPU fan-out at the C ABI (include/scrubby_hip.h: sh_index_replicate / sh_classify_sharded), what replaces the rayon
loop over ONE shared &Aligner of /root/reference/src/cleaner.rs:546-559 for a caller that cannot use torch.distributed.  A one-GPU box
lists device 0 more than once (logical shards: each with its own host thread, context, streams and device buffers); a box with more
devices also gets a real replica per device.  Sharded flags and traces == the single-call ones == the oracle's."""
import numpy as np
import pytest

from tests import workloads as W

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]


@pytest.fixture(scope="module")
def cfg1(oracle):
    from scrubby_amd import lib as S
    S.require_gpu()
    P, R, ref, seqs, reads, off = W.cfg1(oracle, 20001)      # an odd number of records: the last shard ends on the odd one
    idx = S.Index.build([bytes(s) for s in seqs], S.preset("sr"))
    oidx = oracle.Index.build(seqs, 11, 21)
    of, ot = oidx.classify(oracle.preset("sr"), reads, off, threads=8)
    return S, idx, reads, off, of, ot


@pytest.mark.parametrize("n_shards", [1, 2, 3])
def test_logical_shards_on_one_device_equal_single_call_and_oracle(cfg1, n_shards):
    S, idx, reads, off, of, ot = cfg1
    f1, t1, st1, rc1 = idx.classify(reads, off, want_trace=True)
    iset = idx.replicate([0] * n_shards)
    assert iset.n_shards == n_shards
    fs, ts, sts, rcs, first = iset.classify(reads, off, want_trace=True)
    assert rc1 == 0 and rcs == 0
    assert first[0] == 0 and first[-1] == len(off) - 1 and all(int(x) % 2 == 0 for x in first[:-1]) and np.all(np.diff(first.astype(np.int64)) > 0)
    assert np.array_equal(fs, f1) and np.array_equal(fs, of)
    for name in S.TRACE_FIELDS:
        assert np.array_equal(ts[name], t1[name]) and np.array_equal(ts[name], ot[name]), name
    assert sts["n_reads"] == len(off) - 1 and sts["n_host"] == int((of == 1).sum()) == st1["n_host"]
    iset.close()


def test_ragged_reads_are_cut_by_bases_and_an_empty_read_is_reported(cfg1, oracle):
    S, idx, reads, off, of, ot = cfg1
    ref = W.cfg1(oracle, 2)[2]
    recs, bases, offs = W.edge_reads(ref)
    # many short records in front, long ones behind: equal shares of the bases are not equal shares of the records
    big = [bytes(ref[i * 5000:i * 5000 + 2500]) for i in range(40)]
    recs2 = [r for r in recs if len(r) > 0] * 3 + big
    b2 = np.frombuffer(b"".join(recs2), dtype=np.uint8)
    o2 = np.zeros(len(recs2) + 1, dtype=np.uint64); o2[1:] = np.cumsum([len(r) for r in recs2])
    f1, _, _, rc1 = idx.classify(b2, o2)
    iset = idx.replicate([0, 0])
    fs, _, _, rcs, first = iset.classify(b2, o2)
    assert rc1 == 0 and rcs == 0 and np.array_equal(fs, f1)
    half = int(o2[-1]) // 2
    assert abs(int(o2[int(first[1])]) - half) <= 2 * 2500 + 300 and int(first[1]) > len(recs2) // 2      # by bases: the cut lies among the long records
    # with the empty record the run must fail the way the reference's per-read Err does (cleaner.rs:552,566), flags filled all the same
    fe, _, _, rce, _ = iset.classify(bases, offs)
    f0, _, _, rc0 = idx.classify(bases, offs)
    assert rce == S.SH_ERR_EMPTY_READ == rc0 and np.array_equal(fe, f0) and fe[0] == 2
    iset.close()


def test_every_visible_device_gets_a_replica(cfg1):
    S, idx, reads, off, of, ot = cfg1
    n_dev = S.load().sh_device_count()
    iset = idx.replicate(None)
    assert iset.n_shards == n_dev >= 1
    fs, _, sts, rcs, first = iset.classify(reads, off)
    assert rcs == 0 and np.array_equal(fs, of) and len(first) == n_dev + 1
    iset.close()
