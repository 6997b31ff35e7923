"""ctypes loader for the CPU oracle (TEST INFRASTRUCTURE — not product code).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
*** PARITY UNPINNED ***: see oracle/mm_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

SYN_MAX_CONTIGS = 64


class Opts(C.Structure):
    _fields_ = [
        ("k", C.c_int32), ("w", C.c_int32), ("is_sr", C.c_int32), ("mid_occ", C.c_int32),
        ("max_occ", C.c_int32), ("max_max_occ", C.c_int32), ("occ_dist", C.c_int32),
        ("min_mid_occ", C.c_int32), ("max_mid_occ", C.c_int32), ("mid_occ_frac", C.c_float),
        ("q_occ_frac", C.c_float), ("min_cnt", C.c_int32), ("min_chain_score", C.c_int32),
        ("max_gap", C.c_int32), ("max_gap_ref", C.c_int32), ("max_frag_len", C.c_int32),
        ("bw", C.c_int32), ("max_chain_skip", C.c_int32), ("max_chain_iter", C.c_int32),
        ("chain_gap_scale", C.c_float), ("chain_skip_scale", C.c_float),
        # A.6 (mm_align.c): flags bit 0 = MM_F_CIGAR
        ("flags", C.c_int32), ("a", C.c_int32), ("b", C.c_int32), ("q", C.c_int32), ("e", C.c_int32), ("q2", C.c_int32),
        ("e2", C.c_int32), ("sc_ambi", C.c_int32), ("zdrop", C.c_int32), ("zdrop_inv", C.c_int32), ("end_bonus", C.c_int32),
        ("min_dp_max", C.c_int32), ("best_n", C.c_int32), ("bw_long", C.c_int32), ("min_ksw_len", C.c_int32),
        ("pri_ratio", C.c_float), ("mask_level", C.c_float), ("max_clip_ratio", C.c_float),
        ("rmq_inner_dist", C.c_int32), ("rmq_size_cap", C.c_int32), ("rmq_rescue_size", C.c_int32), ("rmq_rescue_ratio", C.c_float),
    ]


F_CIGAR = 1


class Trace(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("n_mini", "n_seed", "n_anchor", "rep_len", "rechained", "n_chain", "best_score", "flag",
                 "n_aligned", "n_regs", "dp_max")] + [("sig", C.c_uint32)]


TRACE_DTYPE = np.dtype([(n, "<u4" if n == "sig" else "<i4") for n, _ in Trace._fields_])


class RefParams(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64), ("genome_len", C.c_uint64), ("n_contigs", C.c_uint32),
        ("sb_shift", C.c_uint32), ("rb_shift", C.c_uint32), ("sat_pct", C.c_uint32),
        ("rep_pct", C.c_uint32), ("n_sat_fam", C.c_uint32), ("n_rep_fam", C.c_uint32),
        ("pad", C.c_uint32), ("contig_start", C.c_uint64 * (SYN_MAX_CONTIGS + 1)),
    ]


class ReadParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("read_len", C.c_uint32), ("host_pct", C.c_uint32),
                ("sub_per_10k", C.c_uint32), ("n_read_pct", C.c_uint32)]


class K2Opts(C.Structure):
    _fields_ = [("k", C.c_int32), ("l", C.c_int32), ("spaced_seed_mask", C.c_uint64), ("toggle_mask", C.c_uint64),
                ("min_acceptable_hash", C.c_uint64), ("value_bits", C.c_int32), ("min_hit_groups", C.c_int32),
                ("confidence", C.c_double)]


K2_RESULT_DTYPE = np.dtype([("call", "<u4"), ("total_kmers", "<u4"), ("hit_groups", "<u4"), ("n_probes", "<u4")])
K2_AMBIG, K2_BORDER = 0xFFFFFFFF, 0xFFFFFFFE


def build():
    """Compile liboracle from oracle/Makefile (gcc)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(_HERE, "libmm_oracle.so")
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    u8p, u64p, i64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint64), C.POINTER(C.c_int64)
    L.mmo_preset.argtypes = [C.c_char_p, C.POINTER(Opts)]
    L.mmo_preset.restype = C.c_int
    L.mmo_hash64.argtypes = [C.c_uint64, C.c_uint64]
    L.mmo_hash64.restype = C.c_uint64
    L.mmo_sketch.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int64]
    L.mmo_sketch.restype = C.c_int64
    L.mmo_index_build.argtypes = [C.c_int, C.POINTER(C.c_void_p), i64p, C.c_int, C.c_int]
    L.mmo_index_build.restype = C.c_void_p
    L.mmo_index_wrap.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int, C.c_int]
    L.mmo_index_wrap.restype = C.c_void_p
    L.mmo_index_free.argtypes = [C.c_void_p]
    L.mmo_index_set_ref.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
    L.mma_ksw_extd2.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int8, C.c_void_p, C.c_int8, C.c_int8, C.c_int8, C.c_int8,
                                C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.mma_gen_simple_mat.argtypes = [C.c_int, C.c_void_p, C.c_int8, C.c_int8, C.c_int8]
    L.mmo_index_n_keys.argtypes = [C.c_void_p]
    L.mmo_index_n_keys.restype = C.c_uint64
    L.mmo_index_n_positions.argtypes = [C.c_void_p]
    L.mmo_index_n_positions.restype = C.c_uint64
    L.mmo_index_dump.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.mmo_index_cal_mid_occ.argtypes = [C.c_void_p, C.c_float]
    L.mmo_index_cal_mid_occ.restype = C.c_int32
    L.mmo_opts_update.argtypes = [C.POINTER(Opts), C.c_void_p]
    L.mmo_map.argtypes = [C.c_void_p, C.POINTER(Opts), C.c_void_p, C.c_int64, C.POINTER(Trace)]
    L.mmo_classify_batch.argtypes = [C.c_void_p, C.POINTER(Opts), C.c_void_p, C.c_void_p, C.c_uint64,
                                     C.c_void_p, C.c_void_p, C.c_int]
    L.mmo_comput_sc.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int32, C.c_int32,
                                C.c_int32, C.c_float, C.c_float]
    L.mmo_comput_sc.restype = C.c_int32
    L.mmo_log2.argtypes = [C.c_float]
    L.mmo_log2.restype = C.c_float
    L.syn_cpu_ref.argtypes = [C.POINTER(RefParams), C.c_uint64, C.c_uint64, C.c_void_p]
    L.syn_cpu_reads.argtypes = [C.POINTER(RefParams), C.POINTER(ReadParams), C.c_uint64, C.c_uint64, C.c_void_p]
    L.syn_cpu_truth.argtypes = [C.POINTER(RefParams), C.POINTER(ReadParams), C.c_uint64, C.c_uint64, C.c_void_p]
    L.syn_cpu_long_lengths.argtypes = [C.POINTER(ReadParams), C.c_uint64, C.c_uint64, C.c_void_p]
    L.syn_cpu_long_reads.argtypes = [C.POINTER(RefParams), C.POINTER(ReadParams), C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]
    L.k2o_default_opts.argtypes = [C.POINTER(K2Opts)]
    L.k2o_hash.argtypes = [C.c_uint64]
    L.k2o_hash.restype = C.c_uint64
    L.k2o_canonical.argtypes = [C.c_uint64, C.c_int]
    L.k2o_canonical.restype = C.c_uint64
    L.k2o_scan.argtypes = [C.c_void_p, C.c_int64, C.POINTER(K2Opts), C.c_void_p, C.c_void_p, C.c_int64]
    L.k2o_scan.restype = C.c_int64
    L.k2o_cht_get.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_uint64]
    L.k2o_cht_get.restype = C.c_uint32
    L.k2o_cht_set.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_uint64, C.c_uint32, C.c_void_p]
    L.k2o_cht_set.restype = C.c_int
    L.k2o_is_ancestor.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
    L.k2o_is_ancestor.restype = C.c_int
    L.k2o_lca.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
    L.k2o_lca.restype = C.c_uint32
    L.k2o_resolve.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_double]
    L.k2o_resolve.restype = C.c_uint32
    L.k2o_classify_pair.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(K2Opts), C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                    C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    L.k2o_classify_batch.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(K2Opts), C.c_void_p, C.c_void_p, C.c_uint64, C.c_int,
                                     C.c_void_p, C.c_int]
    _LIB = L
    return L


def preset(name):
    o = Opts()
    rc = lib().mmo_preset(name.encode(), C.byref(o))
    if rc != 0:
        raise ValueError(f"preset {name!r}: rc={rc}")
    return o


def sketch(seq, w, k, rid=0):
    """Returns (x, y) uint64 arrays: x = hash<<8|span, y = rid<<32|pos<<1|strand."""
    seq = np.frombuffer(bytes(seq), dtype=np.uint8) if not isinstance(seq, np.ndarray) else seq
    cap = 2 * len(seq) + 256
    x = np.zeros(cap, dtype=np.uint64)
    y = np.zeros(cap, dtype=np.uint64)
    n = lib().mmo_sketch(seq.ctypes.data, len(seq), w, k, rid, x.ctypes.data, y.ctypes.data, cap)
    return x[:n].copy(), y[:n].copy()


class Index:
    def __init__(self, handle, keep=None):
        self.h = handle
        self._keep = keep

    @classmethod
    def build(cls, seqs, w, k):
        arrs = [np.frombuffer(bytes(s), dtype=np.uint8) if not isinstance(s, np.ndarray) else np.ascontiguousarray(s)
                for s in seqs]
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        lens = (C.c_int64 * len(arrs))(*[len(a) for a in arrs])
        return cls(lib().mmo_index_build(len(arrs), ptrs, lens, w, k), keep=arrs)

    @classmethod
    def wrap(cls, slots, positions, w, k, ref=None):
        """slots: uint64[2*n_slots] in the product's HBM layout; positions: uint64[]; ref = (packed nt4 uint8[], contig_start
        uint64[n+1]) as the product exports it (Index.export_ref): what the A.6 stage aligns against."""
        slots = np.ascontiguousarray(slots, dtype=np.uint64)
        positions = np.ascontiguousarray(positions, dtype=np.uint64)
        h = lib().mmo_index_wrap(slots.ctypes.data, len(slots) // 2, positions.ctypes.data, len(positions), w, k)
        keep = [slots, positions]
        if ref is not None:
            packed = np.ascontiguousarray(ref[0], dtype=np.uint8)
            starts = np.ascontiguousarray(ref[1], dtype=np.uint64)
            lib().mmo_index_set_ref(h, packed.ctypes.data, starts.ctypes.data, len(starts) - 1)
            keep += [packed, starts]
        return cls(h, keep=keep)

    def ref(self):
        """(packed nt4 codes, contig_start) of the reference this index aligns against (None if it has none)."""
        cs, n = C.c_void_p(), C.c_uint32()
        lib().mmo_index_ref.restype = C.c_void_p
        lib().mmo_index_ref.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32)]
        p = lib().mmo_index_ref(self.h, C.byref(cs), C.byref(n))
        if not p:
            return None
        starts = np.ctypeslib.as_array(C.cast(cs, C.POINTER(C.c_uint64)), shape=(n.value + 1,)).copy()
        packed = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=((int(starts[-1]) + 1) // 2,)).copy()
        return packed, starts

    def dump(self):
        nk, npos = lib().mmo_index_n_keys(self.h), None
        keys = np.zeros(nk, dtype=np.uint64)
        cnt = np.zeros(nk, dtype=np.uint32)
        # total positions = singletons + multi
        pos = np.zeros(nk + lib().mmo_index_n_positions(self.h), dtype=np.uint64)
        lib().mmo_index_dump(self.h, keys.ctypes.data, cnt.ctypes.data, pos.ctypes.data)
        return keys, cnt, pos[: int(cnt.sum())]

    def mid_occ(self, frac):
        return lib().mmo_index_cal_mid_occ(self.h, frac)

    def update_opts(self, o):
        lib().mmo_opts_update(C.byref(o), self.h)
        return o

    def map(self, o, seq):
        seq = np.frombuffer(bytes(seq), dtype=np.uint8) if not isinstance(seq, np.ndarray) else seq
        tr = Trace()
        lib().mmo_map(self.h, C.byref(o), seq.ctypes.data, len(seq), C.byref(tr))
        return {n: getattr(tr, n) for n, _ in Trace._fields_}

    def classify(self, o, bases, offsets, threads=1, want_trace=True):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        flags = np.zeros(n, dtype=np.uint8)
        tr = np.zeros(n, dtype=TRACE_DTYPE) if want_trace else None
        lib().mmo_classify_batch(self.h, C.byref(o), bases.ctypes.data, offsets.ctypes.data, n,
                                 flags.ctypes.data, tr.ctypes.data if want_trace else None, threads)
        return flags, tr

    def __del__(self):
        if getattr(self, "h", None):
            lib().mmo_index_free(self.h)
            self.h = None


def ref_params(seed, contig_lens, sb_shift=17, rb_shift=11, sat_pct=6, rep_pct=45, n_sat_fam=64, n_rep_fam=1000):
    p = RefParams()
    p.seed = seed
    p.n_contigs = len(contig_lens)
    assert p.n_contigs <= SYN_MAX_CONTIGS
    acc = 0
    for i, L in enumerate(contig_lens):
        p.contig_start[i] = acc
        acc += L
    p.contig_start[len(contig_lens)] = acc
    p.genome_len = acc
    p.sb_shift, p.rb_shift, p.sat_pct, p.rep_pct = sb_shift, rb_shift, sat_pct, rep_pct
    p.n_sat_fam, p.n_rep_fam = n_sat_fam, n_rep_fam
    return p


def read_params(seed, read_len=150, host_pct=50, sub_per_10k=50, n_read_pct=1):
    r = ReadParams()
    r.seed, r.read_len, r.host_pct, r.sub_per_10k, r.n_read_pct = seed, read_len, host_pct, sub_per_10k, n_read_pct
    return r


def synth_ref(P, g0, n):
    out = np.zeros(n, dtype=np.uint8)
    lib().syn_cpu_ref(C.byref(P), g0, n, out.ctypes.data)
    return out


def synth_reads(P, R, r0, n):
    out = np.zeros(n * R.read_len, dtype=np.uint8)
    lib().syn_cpu_reads(C.byref(P), C.byref(R), r0, n, out.ctypes.data)
    return out


def synth_truth(P, R, r0, n):
    out = np.zeros(n, dtype=np.uint8)
    lib().syn_cpu_truth(C.byref(P), C.byref(R), r0, n, out.ctypes.data)
    return out


def synth_long_lengths(R, r0, n):
    out = np.zeros(n, dtype=np.uint32)
    lib().syn_cpu_long_lengths(C.byref(R), r0, n, out.ctypes.data)
    return out


def synth_long_reads(P, R, r0, n):
    lens = synth_long_lengths(R, r0, n)
    offs = np.zeros(n + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lens.astype(np.uint64))
    out = np.zeros(int(offs[-1]), dtype=np.uint8)
    lib().syn_cpu_long_reads(C.byref(P), C.byref(R), r0, n, offs.ctypes.data, out.ctypes.data)
    return out, offs


# ---- Kraken2-style classifier (k2_oracle.c; PARITY UNPINNED, see k2_oracle.h) ----------------------------------------
def k2_default_opts():
    o = K2Opts()
    lib().k2o_default_opts(C.byref(o))
    return o


def k2_scan(seq, o):
    seq = np.frombuffer(bytes(seq), dtype=np.uint8) if not isinstance(seq, np.ndarray) else np.ascontiguousarray(seq)
    cap = max(len(seq), 1)
    mins = np.zeros(cap, dtype=np.uint64)
    amb = np.zeros(cap, dtype=np.uint8)
    n = lib().k2o_scan(seq.ctypes.data, len(seq), C.byref(o), mins.ctypes.data, amb.ctypes.data, cap)
    return mins[:n].copy(), amb[:n].copy()


class K2Table:
    """A compact hash table + parent array on the host (built here with k2o_cht_set, or wrapped from the product's export)."""

    def __init__(self, cells, parent, value_bits):
        self.cells = np.ascontiguousarray(cells, dtype=np.uint32)
        self.parent = np.ascontiguousarray(parent, dtype=np.uint32)
        self.value_bits = value_bits

    @classmethod
    def empty(cls, capacity, parent, value_bits):
        return cls(np.zeros(capacity, dtype=np.uint32), parent, value_bits)

    def set(self, key, value, lca=True):
        return lib().k2o_cht_set(self.cells.ctypes.data, len(self.cells), self.value_bits, C.c_uint64(int(key)), int(value),
                                 self.parent.ctypes.data if lca else None)

    def get(self, key):
        return lib().k2o_cht_get(self.cells.ctypes.data, len(self.cells), self.value_bits, C.c_uint64(int(key)))

    def classify_pair(self, o, seq1, seq2=None, want_taxa=False):
        a = np.frombuffer(bytes(seq1), dtype=np.uint8)
        b = np.frombuffer(bytes(seq2), dtype=np.uint8) if seq2 is not None else None
        res = np.zeros(1, dtype=K2_RESULT_DTYPE)
        cap = len(a) + (len(b) if b is not None else 0) + 4
        taxa = np.zeros(cap, dtype=np.uint32)
        nt = C.c_int64()
        lib().k2o_classify_pair(self.cells.ctypes.data, len(self.cells), self.parent.ctypes.data, C.byref(o), a.ctypes.data, len(a),
                                b.ctypes.data if b is not None else None, len(b) if b is not None else 0, res.ctypes.data,
                                taxa.ctypes.data, cap, C.byref(nt))
        r = {n: int(res[0][n]) for n in K2_RESULT_DTYPE.names}
        return (r, taxa[:nt.value].copy()) if want_taxa else r

    def classify(self, o, bases, offsets, paired=False, threads=8):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n_rec = len(offsets) - 1
        n_units = n_rec // 2 if paired else n_rec
        res = np.zeros(max(n_units, 1), dtype=K2_RESULT_DTYPE)
        lib().k2o_classify_batch(self.cells.ctypes.data, len(self.cells), self.parent.ctypes.data, C.byref(o), bases.ctypes.data,
                                 offsets.ctypes.data, n_rec, 1 if paired else 0, res.ctypes.data, threads)
        return res[:n_units]


def k2_lca(parent, a, b):
    parent = np.ascontiguousarray(parent, dtype=np.uint32)
    return lib().k2o_lca(parent.ctypes.data, a, b)


def k2_is_ancestor(parent, a, b):
    parent = np.ascontiguousarray(parent, dtype=np.uint32)
    return bool(lib().k2o_is_ancestor(parent.ctypes.data, a, b))


def k2_resolve(taxa, counts, parent, total_kmers, confidence=0.0):
    taxa = np.ascontiguousarray(taxa, dtype=np.uint32)
    counts = np.ascontiguousarray(counts, dtype=np.uint32)
    parent = np.ascontiguousarray(parent, dtype=np.uint32)
    return lib().k2o_resolve(taxa.ctypes.data, counts.ctypes.data, len(taxa), parent.ctypes.data, total_kmers, confidence)
