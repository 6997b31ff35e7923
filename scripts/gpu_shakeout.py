"""First-contact GPU script: synth parity, index parity, classification parity vs the oracle."""
import sys, time, traceback
sys.path.insert(0, "/root/repo")
import numpy as np
import torch
from oracle import oracle as O
from scrubby_amd import lib as S

def stage(name, fn):
    t = time.time()
    try:
        r = fn()
        print(f"[OK ] {name} ({time.time()-t:.2f}s) {r if r is not None else ''}", flush=True)
        return r
    except Exception:
        print(f"[ERR] {name}", flush=True)
        traceback.print_exc()
        return None

dev = torch.device("cuda:0")
print(torch.cuda.get_device_name(0), flush=True)
CL = [1_000_000] * 5
Pg, Rg = S.ref_params(0x5C2B0001, CL), S.read_params(0x5C2B0002)
Po, Ro = O.ref_params(0x5C2B0001, CL), O.read_params(0x5C2B0002)
G = Pg.genome_len
N = 20000

d_ref = torch.empty(G + 64, dtype=torch.uint8, device=dev)
def t_synth():
    S.synth_ref_device(Pg, 0, G, d_ref); torch.cuda.synchronize()
    cpu = O.synth_ref(Po, 0, G)
    assert np.array_equal(d_ref[:G].cpu().numpy(), cpu), "ref mismatch"
    return "ref bytes equal"
stage("synth ref", t_synth)
ref = O.synth_ref(Po, 0, G)

d_reads = torch.empty(N * 150 + 64, dtype=torch.uint8, device=dev)
d_off = torch.empty(N + 1, dtype=torch.int64, device=dev)
def t_reads():
    S.synth_reads_device(Pg, Rg, 0, N, d_reads, d_off); torch.cuda.synchronize()
    cpu = O.synth_reads(Po, Ro, 0, N)
    assert np.array_equal(d_reads[:N*150].cpu().numpy(), cpu), "reads mismatch"
    assert np.array_equal(d_off.cpu().numpy(), np.arange(N+1)*150)
    return "reads equal"
stage("synth reads", t_reads)
reads = O.synth_reads(Po, Ro, 0, N)
off = np.arange(N + 1, dtype=np.uint64) * 150

o_sr = S.preset("sr")
oo_sr = O.preset("sr")
seqs = [ref[Po.contig_start[i]:Po.contig_start[i+1]] for i in range(5)]
oidx = O.Index.build(seqs, 11, 21)
gidx = stage("index build (device)", lambda: S.Index.build_device(d_ref, [Pg.contig_start[i] for i in range(6)], o_sr))
print(gidx.info(), flush=True)
def t_index():
    slots, pos = gidx.export()
    w = O.Index.wrap(slots, pos, 11, 21)
    k1, c1, p1 = w.dump(); k2, c2, p2 = oidx.dump()
    assert np.array_equal(k1, k2), ("keys", len(k1), len(k2))
    assert np.array_equal(c1, c2), "counts"
    assert np.array_equal(p1, p2), "positions"
    return f"{len(k1)} keys, {len(p1)} positions identical"
stage("index parity", t_index)

def cmp_trace(tag, gf, gt, of, ot):
    bad = np.where(gf != of)[0]
    msg = f"{tag}: flags mismatch {len(bad)}"
    for nm in S.TRACE_FIELDS:
        b = np.where(gt[nm] != ot[nm])[0]
        if len(b):
            msg += f" | {nm}: {len(b)} e.g. r={b[0]} gpu={gt[nm][b[0]]} cpu={ot[nm][b[0]]}"
    return msg

def t_classify():
    gf, gt, st, rc = gidx.classify(reads, off, want_trace=True)
    of, ot = oidx.classify(oo_sr, reads, off, threads=8)
    print(st, flush=True)
    return cmp_trace("cfg1", gf, gt, of, ot) + f" host={int((gf==1).sum())}"
stage("classify parity sr", t_classify)

def t_device_ctx():
    ctx = S.Context(gidx, N, N * 150, 150)
    fl = torch.zeros(N, dtype=torch.uint8, device=dev)
    tr = torch.zeros((N, len(S.TRACE_FIELDS)), dtype=torch.int32, device=dev)
    for it in range(3):
        st = ctx.classify(d_reads[:N*150], d_off, fl, tr)
    of, ot = oidx.classify(oo_sr, reads, off, threads=8)
    gt = tr.cpu().numpy().view(S.TRACE_DTYPE).reshape(-1)
    return cmp_trace("ctx", fl.cpu().numpy(), gt, of, ot) + f" stats={st}"
stage("device ctx", t_device_ctx)

def t_edge():
    rng = np.random.default_rng(7)
    recs = []
    recs.append(b"")                                    # empty
    recs.append(bytes(ref[1000:1010]))                  # shorter than k
    recs.append(bytes(ref[2000:2021]))                  # exactly k
    recs.append(bytes(ref[3000:3031]))                  # k + w - 1
    recs.append(bytes(ref[4000:4032]))
    recs.append(b"N" * 150)
    recs.append(bytes(ref[5000:5150]).lower())
    r = bytearray(ref[6000:6150]); r[40] = ord("N"); r[41] = ord("n"); r[100] = ord("R"); recs.append(bytes(r))
    recs.append(b"A" * 150); recs.append(b"AC" * 75); recs.append(b"ACG" * 50)
    for L in [22, 35, 64, 99, 151, 250, 300, 400]:
        s = int(rng.integers(0, 900000)); recs.append(bytes(ref[s:s+L]))
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    for L in [150, 250]:
        s = int(rng.integers(0, 900000)); recs.append(bytes(ref[s:s+L]).translate(comp)[::-1])
    for _ in range(40):
        recs.append(bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, int(rng.integers(1, 300)))]))
    bases = np.frombuffer(b"".join(recs), dtype=np.uint8)
    offs = np.zeros(len(recs) + 1, dtype=np.uint64); offs[1:] = np.cumsum([len(x) for x in recs])
    gf, gt, st, rc = gidx.classify(bases, offs, want_trace=True)
    of, ot = oidx.classify(oo_sr, bases, offs, threads=1)
    return cmp_trace("edge", gf, gt, of, ot) + f" rc={rc} flags={gf[:12].tolist()}"
stage("edge cases", t_edge)
