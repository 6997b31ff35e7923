"""ctypes binding of the Kraken2-style taxid path of libscrubby_hip.so (include/scrubby_hip.h, `sh_k2_*`, `sh_kraken_run`).

Mirrors what the reference reaches through the external `kraken2` process (Cleaner::run_kraken,
/root/reference/src/cleaner.rs:288-330).  No CPU path: everything here needs the HIP library and a GPU.
"""
import ctypes as C

import numpy as np

from . import lib as S


class K2Opts(C.Structure):
    _fields_ = [("k", C.c_int32), ("l", C.c_int32), ("spaced_seed_mask", C.c_uint64), ("toggle_mask", C.c_uint64),
                ("min_acceptable_hash", C.c_uint64), ("value_bits", C.c_int32), ("min_hit_groups", C.c_int32),
                ("confidence", C.c_double)]


class K2TaxNode(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("parent", "first_child", "child_count", "name_offset", "rank_offset", "external_id", "godparent")]


class K2Info(C.Structure):
    _fields_ = [("capacity", C.c_uint64), ("size", C.c_uint64), ("n_nodes", C.c_uint64), ("hbm_bytes", C.c_uint64),
                ("k", C.c_int32), ("l", C.c_int32), ("value_bits", C.c_int32), ("key_bits", C.c_int32)]


class K2Stats(C.Structure):
    _fields_ = [("n_units", C.c_uint64), ("n_classified", C.c_uint64), ("n_probes", C.c_uint64), ("n_kmers", C.c_uint64),
                ("n_overflow", C.c_uint64), ("ms_classify", C.c_float), ("ms_total", C.c_float)]


class KrakenConfig(C.Structure):
    _fields_ = [("input", C.c_char_p * 2), ("output", C.c_char_p * 2), ("n_files", C.c_uint32), ("extract", C.c_int32),
                ("db", C.c_char_p), ("workdir", C.c_char_p),
                ("taxa", C.POINTER(C.c_char_p)), ("n_taxa", C.c_uint32),
                ("taxa_direct", C.POINTER(C.c_char_p)), ("n_taxa_direct", C.c_uint32),
                ("confidence", C.c_double), ("min_hit_groups", C.c_int32),
                ("json", C.c_char_p), ("read_ids", C.c_char_p), ("command", C.c_char_p),
                ("device", C.c_int32), ("threads", C.c_int32), ("classifier_args", C.c_char_p)]


RESULT_DTYPE = np.dtype([("taxid", "<u4"), ("call", "<u4"), ("total_kmers", "<u4"), ("hit_groups", "<u4")])


def default_opts():
    o = K2Opts()
    S.check(S.load().sh_k2_default_opts(C.byref(o)))
    return o


def make_taxonomy(parents, externals, names, ranks):
    """Node arrays for sh_k2_create from per-node lists (index = internal id; entry 0 is the unused sentinel)."""
    n = len(parents)
    nodes = (K2TaxNode * n)()
    name_pool, rank_pool = bytearray(), bytearray()
    kids = [[] for _ in range(n)]
    for i in range(2, n):
        assert parents[i] < i, "ids must be breadth-first"
        kids[parents[i]].append(i)
    for i in range(n):
        nodes[i].parent = parents[i]
        nodes[i].external_id = externals[i]
        nodes[i].name_offset = len(name_pool); name_pool += names[i].encode() + b"\0"
        nodes[i].rank_offset = len(rank_pool); rank_pool += ranks[i].encode() + b"\0"
        if kids[i]:
            assert kids[i] == list(range(kids[i][0], kids[i][0] + len(kids[i]))), "children must have consecutive ids"
            nodes[i].first_child, nodes[i].child_count = kids[i][0], len(kids[i])
    return nodes, bytes(name_pool), bytes(rank_pool)


class K2Db:
    def __init__(self, handle):
        self.h = handle

    @classmethod
    def open(cls, path, device=0):
        S.require_gpu()
        h = C.c_void_p()
        S.check(S.load().sh_k2_open(str(path).encode(), device, C.byref(h)))
        return cls(h)

    @classmethod
    def create(cls, opts, capacity, parents, externals, names, ranks, device=0):
        S.require_gpu()
        nodes, npool, rpool = make_taxonomy(parents, externals, names, ranks)
        h = C.c_void_p()
        S.check(S.load().sh_k2_create(C.byref(opts), C.c_uint64(capacity), nodes, C.c_uint64(len(parents)), npool, C.c_uint64(len(npool)),
                                      rpool, C.c_uint64(len(rpool)), device, C.byref(h)))
        return cls(h)

    def insert(self, keys, taxa):
        import torch
        dk = torch.from_numpy(np.ascontiguousarray(keys, dtype=np.uint64).view(np.int64)).cuda()
        dt = torch.from_numpy(np.ascontiguousarray(taxa, dtype=np.uint32).view(np.int32)).cuda()
        S.check(S.load().sh_k2_insert_device(self.h, C.c_void_p(dk.data_ptr()), C.c_void_p(dt.data_ptr()), C.c_uint64(len(keys)), None))

    def insert_sequence_device(self, d_bases, n, taxon):
        r = C.c_uint64()
        S.check(S.load().sh_k2_insert_sequence_device(self.h, C.c_void_p(d_bases.data_ptr()), C.c_uint64(n), C.c_uint32(taxon), None, C.byref(r)))
        return r.value

    def insert_sequence(self, seq, taxon):
        import torch
        a = np.frombuffer(bytes(seq), dtype=np.uint8) if not isinstance(seq, np.ndarray) else seq
        d = torch.from_numpy(np.concatenate([a, np.full(64, ord("N"), np.uint8)])).cuda()
        return self.insert_sequence_device(d, len(a), taxon)

    def insert_random(self, seed, n, taxon_lo, taxon_hi):
        S.check(S.load().sh_k2_insert_random(self.h, C.c_uint64(seed), C.c_uint64(n), C.c_uint32(taxon_lo), C.c_uint32(taxon_hi), None))

    def save(self, path):
        S.check(S.load().sh_k2_save(self.h, str(path).encode()))

    def info(self):
        i = K2Info()
        S.check(S.load().sh_k2_info_get(self.h, C.byref(i)))
        return {n: getattr(i, n) for n, _ in K2Info._fields_}

    def opts(self):
        o = K2Opts()
        S.check(S.load().sh_k2_db_opts(self.h, C.byref(o)))
        return o

    def export(self):
        i = self.info()
        cells = np.zeros(i["capacity"], dtype=np.uint32)
        parent = np.zeros(i["n_nodes"], dtype=np.uint32)
        ext = np.zeros(i["n_nodes"], dtype=np.uint32)
        S.check(S.load().sh_k2_export(self.h, C.c_void_p(cells.ctypes.data), C.c_void_p(parent.ctypes.data), C.c_void_p(ext.ctypes.data)))
        return cells, parent, ext

    def classify(self, bases, offsets, paired=False, opts=None):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n_rec = len(offsets) - 1
        n_units = n_rec // 2 if paired else n_rec
        out = np.zeros(max(n_units, 1), dtype=RESULT_DTYPE)
        st = K2Stats()
        S.check(S.load().sh_k2_classify_batch(self.h, C.byref(opts) if opts is not None else None, C.c_void_p(bases.ctypes.data),
                                              C.c_void_p(offsets.ctypes.data), C.c_uint64(n_rec), 1 if paired else 0,
                                              C.c_void_p(out.ctypes.data), C.byref(st)))
        return out[:n_units], {n: getattr(st, n) for n, _ in K2Stats._fields_}

    def classify_device(self, d_bases, d_offsets, n_records, paired, d_out, opts=None):
        st = K2Stats()
        S.check(S.load().sh_k2_classify_device(self.h, C.byref(opts) if opts is not None else None, C.c_void_p(d_bases.data_ptr()),
                                               C.c_void_p(d_offsets.data_ptr()), C.c_uint64(n_records), 1 if paired else 0,
                                               C.c_void_p(d_out.data_ptr()), S._stream_ptr(), C.byref(st)))
        return {n: getattr(st, n) for n, _ in K2Stats._fields_}

    def write_report(self, results, path):
        r = np.ascontiguousarray(results, dtype=RESULT_DTYPE)
        S.check(S.load().sh_k2_write_report(self.h, C.c_void_p(r.ctypes.data), C.c_uint64(len(r)), str(path).encode()))

    def close(self):
        if self.h:
            S.load().sh_k2_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def kraken_run(inputs, outputs, db, taxa=(), taxa_direct=(), workdir=None, confidence=-1.0, min_hit_groups=0, extract=False,
               json=None, read_ids=None, command="", device=0, threads=4, classifier_args=None):
    c = KrakenConfig()
    for i, (a, b) in enumerate(zip(inputs, outputs)):
        c.input[i], c.output[i] = str(a).encode(), str(b).encode()
    c.n_files, c.extract, c.db = len(inputs), int(extract), str(db).encode()
    c.workdir = str(workdir).encode() if workdir else None
    ta = (C.c_char_p * max(len(taxa), 1))(*[t.encode() for t in taxa])
    td = (C.c_char_p * max(len(taxa_direct), 1))(*[t.encode() for t in taxa_direct])
    c.taxa, c.n_taxa, c.taxa_direct, c.n_taxa_direct = ta, len(taxa), td, len(taxa_direct)
    c.confidence, c.min_hit_groups = confidence, min_hit_groups
    c.json = str(json).encode() if json else None
    c.read_ids = str(read_ids).encode() if read_ids else None
    c.command, c.device, c.threads = command.encode(), device, threads
    c.classifier_args = classifier_args.encode() if classifier_args else None
    r = S.ReadsResult()
    S.check(S.load().sh_kraken_run(C.byref(c), C.byref(r)))
    return {n: getattr(r, n) for n, _ in S.ReadsResult._fields_}
