"""ctypes binding of libscrubby_hip.so (include/scrubby_hip.h).

The library is the product; this module only loads it and marshals arguments.  There is no
CPU fallback: if the shared object is missing or no MI355X is visible, calls raise.
torch is used for device memory and streams only (tensor.data_ptr(), current stream).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libscrubby_hip.so")
SYN_MAX_CONTIGS = 64

SH_OK = 0
STATUS_NAMES = {
    0: "SH_OK", 1: "SH_ERR_BAD_ARG", 2: "SH_ERR_PRESET_UNKNOWN", 3: "SH_ERR_PRESET_UNSUPPORTED",
    4: "SH_ERR_NO_DEVICE", 5: "SH_ERR_OOM", 6: "SH_ERR_HIP", 7: "SH_ERR_IO", 8: "SH_ERR_EMPTY_READ",
    9: "SH_ERR_INDEX",
}
SH_ERR_PRESET_UNKNOWN, SH_ERR_PRESET_UNSUPPORTED, SH_ERR_EMPTY_READ = 2, 3, 8


class ScrubbyHipError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")
        self.status = status
        self.message = message


class Opts(C.Structure):
    _fields_ = [
        ("k", C.c_int32), ("w", C.c_int32), ("is_sr", C.c_int32), ("mid_occ", C.c_int32),
        ("max_occ", C.c_int32), ("max_max_occ", C.c_int32), ("occ_dist", C.c_int32),
        ("min_mid_occ", C.c_int32), ("max_mid_occ", C.c_int32), ("mid_occ_frac", C.c_float),
        ("q_occ_frac", C.c_float), ("min_cnt", C.c_int32), ("min_chain_score", C.c_int32),
        ("max_gap", C.c_int32), ("max_gap_ref", C.c_int32), ("max_frag_len", C.c_int32),
        ("bw", C.c_int32), ("max_chain_skip", C.c_int32), ("max_chain_iter", C.c_int32),
        ("chain_gap_scale", C.c_float), ("chain_skip_scale", C.c_float),
        # the extension stage `.with_cigar()` enables (flags bit 0 = SH_F_CIGAR) and its ksw2 scores
        ("flags", C.c_int32), ("a", C.c_int32), ("b", C.c_int32), ("q", C.c_int32), ("e", C.c_int32), ("q2", C.c_int32),
        ("e2", C.c_int32), ("sc_ambi", C.c_int32), ("zdrop", C.c_int32), ("zdrop_inv", C.c_int32), ("end_bonus", C.c_int32),
        ("min_dp_max", C.c_int32), ("best_n", C.c_int32), ("bw_long", C.c_int32), ("min_ksw_len", C.c_int32),
        ("pri_ratio", C.c_float), ("mask_level", C.c_float), ("max_clip_ratio", C.c_float),
        # the RMQ long-join re-chain of the long-read presets
        ("rmq_inner_dist", C.c_int32), ("rmq_size_cap", C.c_int32), ("rmq_rescue_size", C.c_int32), ("rmq_rescue_ratio", C.c_float),
    ]


SH_F_CIGAR = 1
TRACE_FIELDS = ("n_mini", "n_seed", "n_anchor", "rep_len", "rechained", "n_chain", "best_score", "flag",
                "n_aligned", "n_regs", "dp_max", "sig")
TRACE_DTYPE = np.dtype([(n, "<u4" if n == "sig" else "<i4") for n in TRACE_FIELDS])


class IndexInfo(C.Structure):
    _fields_ = [
        ("k", C.c_int32), ("w", C.c_int32), ("mid_occ", C.c_int32), ("n_contigs", C.c_uint32),
        ("n_bases", C.c_uint64), ("n_minimizers", C.c_uint64), ("n_keys", C.c_uint64),
        ("n_slots", C.c_uint64), ("n_positions", C.c_uint64), ("hbm_bytes", C.c_uint64),
        ("build_ms", C.c_double),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("n_reads", C.c_uint64), ("n_host", C.c_uint64), ("n_no_seed", C.c_uint64),
        ("n_chain_small", C.c_uint64), ("n_chain_large", C.c_uint64), ("n_minimizers", C.c_uint64),
        ("n_bases", C.c_uint64), ("ms_sketch_probe", C.c_double), ("ms_chain_small", C.c_double),
        ("ms_chain_large", C.c_double), ("ms_total", C.c_double),
        ("n_anchors", C.c_uint64), ("n_clusters", C.c_uint64), ("n_resketch", C.c_uint64), ("n_pair_decided", C.c_uint64),
        ("n_ext_reads", C.c_uint64), ("n_ext_regions", C.c_uint64), ("n_ext_dropped", C.c_uint64), ("ms_ext", C.c_double),
        ("n_ext_shortcut", C.c_uint64), ("n_ext_fallback", C.c_uint64), ("ms_ext_fallback", C.c_double),
        ("n_ext_unresolved", C.c_uint64), ("n_rmq_rechained", C.c_uint64), ("n_rmq_tied", C.c_uint64),
        ("n_dp_parallel", C.c_uint64), ("n_dp_dirty", C.c_uint64), ("n_top_settled", C.c_uint64),
        ("n_locus_reads", C.c_uint64), ("n_locus_redone", C.c_uint64), ("n_rmq_exact", C.c_uint64), ("n_rmq_open", C.c_uint64), ("n_ext_ondemand", C.c_uint64),
    ]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class ReadsConfig(C.Structure):
    _fields_ = [("input", C.c_char_p * 2), ("output", C.c_char_p * 2), ("n_files", C.c_uint32), ("extract", C.c_int32),
                ("index", C.c_char_p), ("preset", C.c_char_p), ("json", C.c_char_p), ("read_ids", C.c_char_p),
                ("command", C.c_char_p), ("threads", C.c_int32), ("device", C.c_int32)]


class ReadsResult(C.Structure):
    _fields_ = [("reads_in", C.c_uint64), ("reads_out", C.c_uint64), ("reads_removed", C.c_uint64),
                ("reads_extracted", C.c_uint64), ("n_depleted_ids", C.c_uint64), ("ms_index", C.c_double),
                ("ms_ingest", C.c_double), ("ms_classify", C.c_double), ("ms_write", C.c_double),
                ("n_ext_unresolved", C.c_uint64), ("n_rmq_open", C.c_uint64)]


class ClassifierConfig(C.Structure):
    _fields_ = [("input", C.c_char_p * 2), ("output", C.c_char_p * 2), ("n_files", C.c_uint32), ("extract", C.c_int32),
                ("report", C.c_char_p), ("reads", C.c_char_p), ("classifier", C.c_char_p),
                ("taxa", C.POINTER(C.c_char_p)), ("n_taxa", C.c_uint32),
                ("taxa_direct", C.POINTER(C.c_char_p)), ("n_taxa_direct", C.c_uint32),
                ("json", C.c_char_p), ("read_ids", C.c_char_p), ("command", C.c_char_p)]


class AlignmentConfig(C.Structure):
    _fields_ = [("input", C.c_char_p * 2), ("output", C.c_char_p * 2), ("n_files", C.c_uint32), ("extract", C.c_int32),
                ("alignment", C.c_char_p), ("format", C.c_char_p), ("min_len", C.c_uint64), ("min_cov", C.c_double),
                ("min_mapq", C.c_uint32), ("json", C.c_char_p), ("read_ids", C.c_char_p), ("command", C.c_char_p)]


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


EXPORTS = [
    "sh_version", "sh_device_count", "sh_last_error", "sh_preset",
    "sh_index_build", "sh_index_build_device", "sh_index_build_fasta", "sh_index_save", "sh_index_load",
    "sh_index_info_get", "sh_index_export", "sh_index_export_ref", "sh_index_free",
    "sh_ctx_create", "sh_ctx_destroy", "sh_ctx_debug_list", "sh_classify_device", "sh_classify_batch",
    "sh_index_replicate", "sh_index_set_size", "sh_index_set_free", "sh_classify_sharded",
    "sh_synth_ref_device", "sh_synth_reads_device", "sh_synth_long_reads_device", "sh_bench_gather", "sh_dbg_rmq_trace", "sh_dbg_wave_ops", "sh_pack_flags_device",
    "sh_reads_run", "sh_release_cached_ctx", "sh_host_get_id", "sh_host_filter_fastx", "sh_host_filter_fastx_stream", "sh_host_read_difference",
    "sh_classifier_run", "sh_classifier_taxids", "sh_alignment_run",
    "sh_k2_default_opts", "sh_k2_open", "sh_k2_create", "sh_k2_insert_device", "sh_k2_insert_sequence_device",
    "sh_k2_insert_random", "sh_k2_save", "sh_k2_info_get", "sh_k2_db_opts", "sh_k2_export", "sh_k2_free",
    "sh_k2_classify_device", "sh_k2_classify_batch", "sh_k2_write_report", "sh_kraken_run",
]

_LIB = None


def load():
    """Load libscrubby_hip.so; raises if it has not been built (no fallback)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise ScrubbyHipError(4, f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
    # torch ships its own libamdhip64; load it first so that this process has ONE HIP runtime
    # (device memory and streams come from torch, kernels from this library)
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    vp, u64, i32, u32 = C.c_void_p, C.c_uint64, C.c_int32, C.c_uint32
    L.sh_version.restype = i32
    L.sh_device_count.restype = i32
    L.sh_last_error.restype = C.c_char_p
    L.sh_preset.argtypes = [C.c_char_p, C.POINTER(Opts)]
    L.sh_index_build.argtypes = [C.POINTER(vp), C.POINTER(u64), u32, C.POINTER(Opts), i32, C.POINTER(vp)]
    L.sh_index_build_device.argtypes = [vp, C.POINTER(u64), u32, C.POINTER(Opts), i32, vp, C.POINTER(vp)]
    L.sh_index_build_fasta.argtypes = [C.c_char_p, C.POINTER(Opts), i32, C.POINTER(vp)]
    L.sh_index_save.argtypes = [vp, C.c_char_p]
    L.sh_index_load.argtypes = [C.c_char_p, i32, C.POINTER(vp)]
    L.sh_index_info_get.argtypes = [vp, C.POINTER(IndexInfo)]
    L.sh_index_export.argtypes = [vp, vp, vp]
    L.sh_index_export_ref.argtypes = [vp, vp, vp]
    L.sh_index_free.argtypes = [vp]
    L.sh_ctx_create.argtypes = [vp, C.POINTER(Opts), u64, u64, u32, C.POINTER(vp)]
    L.sh_ctx_destroy.argtypes = [vp]
    L.sh_classify_device.argtypes = [vp, vp, vp, u64, u64, vp, vp, vp, C.POINTER(Stats)]
    L.sh_classify_batch.argtypes = [vp, C.POINTER(Opts), vp, vp, u64, vp, vp, C.POINTER(Stats)]
    L.sh_index_replicate.argtypes = [vp, C.POINTER(i32), u32, C.POINTER(vp)]
    L.sh_index_set_size.argtypes = [vp]
    L.sh_index_set_free.argtypes = [vp]
    L.sh_classify_sharded.argtypes = [vp, C.POINTER(Opts), vp, vp, u64, vp, vp, C.POINTER(Stats), vp]
    L.sh_synth_ref_device.argtypes = [C.POINTER(RefParams), u64, u64, vp, vp]
    L.sh_synth_reads_device.argtypes = [C.POINTER(RefParams), C.POINTER(ReadParams), u64, u64, vp, vp, vp]
    L.sh_synth_long_reads_device.argtypes = [C.POINTER(RefParams), C.POINTER(ReadParams), u64, u64, vp, u64, vp, vp]
    L.sh_bench_gather.argtypes = [vp, u64, i32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.sh_dbg_rmq_trace.argtypes = [i32, u64, i32, i32, i32, i32, vp, vp]
    L.sh_dbg_wave_ops.argtypes = [i32, vp, vp, i32, vp, vp]
    L.sh_pack_flags_device.argtypes = [vp, u64, vp, vp]
    L.sh_reads_run.argtypes = [C.POINTER(ReadsConfig), C.POINTER(ReadsResult)]
    L.sh_classifier_run.argtypes = [C.POINTER(ClassifierConfig), C.POINTER(ReadsResult)]
    L.sh_alignment_run.argtypes = [C.POINTER(AlignmentConfig), C.POINTER(ReadsResult)]
    L.sh_classifier_taxids.argtypes = [C.c_char_p, C.POINTER(C.c_char_p), u32, C.POINTER(C.c_char_p), u32, C.c_char_p, C.c_size_t, C.POINTER(u64)]
    L.sh_host_get_id.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
    L.sh_host_filter_fastx.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), u64, i32, C.POINTER(u64), C.POINTER(u64)]
    L.sh_host_filter_fastx_stream.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), u64, i32, u64, i32, i32, C.POINTER(u64), C.POINTER(u64)]
    L.sh_host_read_difference.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), u32, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]
    for name in EXPORTS:
        if name not in ("sh_version", "sh_device_count", "sh_last_error", "sh_index_set_size"):
            getattr(L, name).restype = i32
    L.sh_index_set_size.restype = u32
    _LIB = L
    return L


def check(status):
    if status != SH_OK:
        raise ScrubbyHipError(status, load().sh_last_error().decode(errors="replace"))


def preset(name):
    o = Opts()
    check(load().sh_preset(name.encode(), C.byref(o)))
    return o


def require_gpu():
    L = load()
    if L.sh_device_count() < 1:
        raise ScrubbyHipError(4, "no HIP device visible; libscrubby_hip has no CPU path")
    return L


def _stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Index:
    """Owns an sh_index (table + positions resident in HBM)."""

    def __init__(self, handle, opts):
        self.h = handle
        self.opts = opts

    @classmethod
    def build(cls, seqs, opts, device=0):
        L = require_gpu()
        arrs = [np.frombuffer(bytes(s), dtype=np.uint8) if not isinstance(s, np.ndarray) else np.ascontiguousarray(s, dtype=np.uint8)
                for s in seqs]
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        lens = (C.c_uint64 * len(arrs))(*[len(a) for a in arrs])
        h = C.c_void_p()
        check(L.sh_index_build(ptrs, lens, len(arrs), C.byref(opts), device, C.byref(h)))
        return cls(h, opts)

    @classmethod
    def build_device(cls, d_bases, contig_starts, opts, device=0):
        """d_bases: torch uint8 CUDA tensor of concatenated contigs."""
        L = require_gpu()
        cs = (C.c_uint64 * len(contig_starts))(*[int(x) for x in contig_starts])
        h = C.c_void_p()
        check(L.sh_index_build_device(C.c_void_p(d_bases.data_ptr()), cs, len(contig_starts) - 1, C.byref(opts), device,
                                      _stream_ptr(), C.byref(h)))
        return cls(h, opts)

    @classmethod
    def build_fasta(cls, path, opts, device=0):
        L = require_gpu()
        h = C.c_void_p()
        check(L.sh_index_build_fasta(os.fsencode(path), C.byref(opts), device, C.byref(h)))
        idx = cls(h, opts)
        info = idx.info()      # a minimap2 index file (.mmi) brings its own k and w: they prevail over the preset's
        if (info["k"], info["w"]) != (opts.k, opts.w):
            o2 = Opts()
            C.memmove(C.byref(o2), C.byref(opts), C.sizeof(Opts))
            o2.k, o2.w = info["k"], info["w"]
            idx.opts = o2
        return idx

    @classmethod
    def load(cls, path, opts, device=0):
        L = require_gpu()
        h = C.c_void_p()
        check(L.sh_index_load(os.fsencode(path), device, C.byref(h)))
        return cls(h, opts)

    def save(self, path):
        check(load().sh_index_save(self.h, os.fsencode(path)))

    def info(self):
        i = IndexInfo()
        check(load().sh_index_info_get(self.h, C.byref(i)))
        return {n: getattr(i, n) for n, _ in IndexInfo._fields_}

    def export(self):
        """(slots uint64[2*n_slots], positions uint64[n_positions]) copied to host."""
        inf = self.info()
        slots = np.zeros(2 * inf["n_slots"], dtype=np.uint64)
        pos = np.zeros(max(inf["n_positions"], 1), dtype=np.uint64)
        check(load().sh_index_export(self.h, slots.ctypes.data, pos.ctypes.data))
        return slots, pos[: inf["n_positions"]]

    def export_ref(self):
        """(packed nt4 codes uint8[(n_bases + 1) // 2], contig_start uint64[n_contigs + 1]): the reference as resident in HBM."""
        inf = self.info()
        packed = np.zeros((inf["n_bases"] + 1) // 2 + 1, dtype=np.uint8)
        starts = np.zeros(inf["n_contigs"] + 1, dtype=np.uint64)
        check(load().sh_index_export_ref(self.h, packed.ctypes.data, starts.ctypes.data))
        return packed, starts

    def classify(self, bases, offsets, want_trace=False):
        """Host buffers in, host flags (and trace) out: sh_classify_batch."""
        L = require_gpu()
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        flags = np.zeros(max(n, 1), dtype=np.uint8)
        tr = np.zeros(max(n, 1), dtype=TRACE_DTYPE) if want_trace else None
        st = Stats()
        rc = L.sh_classify_batch(self.h, C.byref(self.opts), bases.ctypes.data, offsets.ctypes.data, n, flags.ctypes.data,
                                 tr.ctypes.data if want_trace else None, C.byref(st))
        if rc not in (SH_OK, SH_ERR_EMPTY_READ):
            check(rc)
        return flags[:n], (tr[:n] if want_trace else None), st.as_dict(), rc

    def replicate(self, devices=None):
        """One replica of the index per shard (sh_index_replicate): devices = list of device ordinals (None: every visible device)."""
        return IndexSet(self, devices)

    def gather_bench(self, n_probes=1 << 28, iters=3):
        gbs, ms = C.c_double(), C.c_double()
        check(load().sh_bench_gather(self.h, n_probes, iters, C.byref(gbs), C.byref(ms)))
        return gbs.value, ms.value

    def close(self):
        if getattr(self, "h", None):
            load().sh_index_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class IndexSet:
    """sh_index_set: replicas of one index, one per shard; classify() is sh_classify_sharded (the in-process multi-GPU fan-out a Rust
    caller binds instead of the rayon loop of cleaner.rs:546-559).  Marshalling only."""

    def __init__(self, index, devices=None):
        L = require_gpu()
        self.index = index      # borrowed for its own device: keep it alive
        h = C.c_void_p()
        if devices is None:
            check(L.sh_index_replicate(index.h, None, 0, C.byref(h)))
        else:
            arr = (C.c_int32 * len(devices))(*devices)
            check(L.sh_index_replicate(index.h, arr, len(devices), C.byref(h)))
        self.h = h
        self.n_shards = int(L.sh_index_set_size(h))

    def classify(self, bases, offsets, want_trace=False):
        L = require_gpu()
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        flags = np.zeros(max(n, 1), dtype=np.uint8)
        tr = np.zeros(max(n, 1), dtype=TRACE_DTYPE) if want_trace else None
        first = np.zeros(self.n_shards + 1, dtype=np.uint64)
        st = Stats()
        rc = L.sh_classify_sharded(self.h, C.byref(self.index.opts), bases.ctypes.data, offsets.ctypes.data, n, flags.ctypes.data,
                                   tr.ctypes.data if want_trace else None, C.byref(st), first.ctypes.data)
        if rc not in (SH_OK, SH_ERR_EMPTY_READ):
            check(rc)
        return flags[:n], (tr[:n] if want_trace else None), st.as_dict(), rc, first

    def close(self):
        if getattr(self, "h", None):
            load().sh_index_set_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """Owns an sh_ctx: stream-ordered scratch for device-resident batches."""

    def __init__(self, index, max_reads, max_bases, max_read_len):
        L = require_gpu()
        self.index = index
        h = C.c_void_p()
        check(L.sh_ctx_create(index.h, C.byref(index.opts), max_reads, max_bases, max_read_len, C.byref(h)))
        self.h = h

    def classify(self, d_bases, d_offsets, d_flags, d_trace=None, want_stats=True):
        """All arguments are torch CUDA tensors: bases uint8, offsets int64 (read as uint64) [n + 1], flags uint8 [n], and optionally
        the trace, int32 [n, 12] = one sh_trace (48 bytes) per read, contiguous - the kernels write all twelve words."""
        n = d_offsets.numel() - 1
        if d_trace is not None:
            assert d_trace.is_contiguous() and d_trace.element_size() == 4 and d_trace.numel() >= n * len(TRACE_FIELDS), \
                f"trace buffer must hold {len(TRACE_FIELDS)} int32 words per read ({n} reads)"
        assert d_flags.numel() >= n and d_offsets.is_contiguous() and d_bases.is_contiguous()
        st = Stats()
        check(load().sh_classify_device(self.h, C.c_void_p(d_bases.data_ptr()), C.c_void_p(d_offsets.data_ptr()), n,
                                        d_bases.numel(), C.c_void_p(d_flags.data_ptr()),
                                        C.c_void_p(d_trace.data_ptr()) if d_trace is not None else None,
                                        _stream_ptr(), C.byref(st) if want_stats else None))
        return st.as_dict() if want_stats else None

    def debug_list(self, which):
        """Ordinals of the reads that took a rare path (sh_ctx_debug_list).  Short reads, within the last chunk classified: 0 re-chained with
        max_occ, 1 regs[0] aligned base by base, 2 the complete procedure over every chain.  Long reads, within the last call: 3 long join on
        the literal tree, 4 tie left open, 5 unresolved, 6 redone with every anchor, 7 memory on demand, 8 probe undecided, 9 second
        working-memory size, 10 one-lane trees."""
        n = C.c_uint64()
        check(load().sh_ctx_debug_list(self.h, which, None, C.c_uint64(0), C.byref(n)))
        out = np.zeros(max(n.value, 1), dtype=np.uint32)
        check(load().sh_ctx_debug_list(self.h, which, C.c_void_p(out.ctypes.data), C.c_uint64(n.value), C.byref(n)))
        return out[:n.value]

    def close(self):
        if getattr(self, "h", None):
            load().sh_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def ref_params(seed, contig_lens, sb_shift=17, rb_shift=11, sat_pct=6, rep_pct=45, n_sat_fam=64, n_rep_fam=1000):
    p = RefParams()
    p.seed = seed
    p.n_contigs = len(contig_lens)
    assert p.n_contigs <= SYN_MAX_CONTIGS
    acc = 0
    for i, ln in enumerate(contig_lens):
        p.contig_start[i] = acc
        acc += ln
    p.contig_start[len(contig_lens)] = acc
    p.genome_len = acc
    p.sb_shift, p.rb_shift, p.sat_pct, p.rep_pct = sb_shift, rb_shift, sat_pct, rep_pct
    p.n_sat_fam, p.n_rep_fam = n_sat_fam, n_rep_fam
    return p


def read_params(seed, read_len=150, host_pct=50, sub_per_10k=50, n_read_pct=1):
    r = ReadParams()
    r.seed, r.read_len, r.host_pct, r.sub_per_10k, r.n_read_pct = seed, read_len, host_pct, sub_per_10k, n_read_pct
    return r


def synth_ref_device(P, g0, n, out):
    """Fill torch uint8 CUDA tensor `out` (>= n bytes) with reference bases [g0, g0+n)."""
    check(require_gpu().sh_synth_ref_device(C.byref(P), g0, n, C.c_void_p(out.data_ptr()), _stream_ptr()))


def synth_reads_device(P, R, r0, n_records, out, offsets=None):
    check(require_gpu().sh_synth_reads_device(C.byref(P), C.byref(R), r0, n_records, C.c_void_p(out.data_ptr()),
                                             C.c_void_p(offsets.data_ptr()) if offsets is not None else None, _stream_ptr()))


# ---- host-side mirror of the reference path (C++ in csrc/sh_host.cpp) ------------------------------------------------
def pack_flags_device(d_flags):
    """uint8 CUDA flags (1 = host) -> little-endian bitmap on the device (wave ballots; one launch on the current stream)."""
    import torch
    n = d_flags.numel()
    bits = torch.empty((n + 7) // 8, dtype=torch.uint8, device=d_flags.device)
    check(load().sh_pack_flags_device(C.c_void_p(d_flags.data_ptr()), n, C.c_void_p(bits.data_ptr()), _stream_ptr()))
    return bits


def get_id(header):
    """utils.rs:91-103: first whitespace token of a FASTX header (bytes or str)."""
    h = header if isinstance(header, bytes) else header.encode()
    out = C.create_string_buffer(len(h) + 2)
    check(load().sh_host_get_id(h, out, len(h) + 2))
    return out.value.decode()


def filter_fastx(inp, out, ids, extract=False):
    """cleaner.rs:731-760 FastqCleaner::clean_reads; returns (records_in, records_out)."""
    arr = (C.c_char_p * max(len(ids), 1))(*[i.encode() for i in ids])
    n_in, n_out = C.c_uint64(), C.c_uint64()
    check(load().sh_host_filter_fastx(os.fsencode(inp), os.fsencode(out), arr, len(ids), int(extract), C.byref(n_in), C.byref(n_out)))
    return n_in.value, n_out.value


def filter_fastx_stream(inp, out, ids, extract=False, chunk_bytes=64 << 20, threads=4, retain=True):
    """The same filter as sh_reads_run's pass 2 runs it (csrc/sh_stream.cpp): chunked, multi-threaded, ordered writer."""
    arr = (C.c_char_p * max(len(ids), 1))(*[i.encode() for i in ids])
    n_in, n_out = C.c_uint64(), C.c_uint64()
    check(load().sh_host_filter_fastx_stream(os.fsencode(inp), os.fsencode(out), arr, len(ids), int(extract), chunk_bytes, threads, int(retain),
                                             C.byref(n_in), C.byref(n_out)))
    return n_in.value, n_out.value


def read_difference(inputs, outputs):
    """utils.rs:250-285 ReadDifference::get_difference; returns (reads_in, reads_out, difference)."""
    n = len(inputs)
    a = (C.c_char_p * n)(*[os.fsencode(p) for p in inputs])
    b = (C.c_char_p * n)(*[os.fsencode(p) for p in outputs])
    r = [C.c_uint64(), C.c_uint64(), C.c_uint64()]
    check(load().sh_host_read_difference(a, b, n, C.byref(r[0]), C.byref(r[1]), C.byref(r[2])))
    return tuple(x.value for x in r)


def release_cached_context():
    """Give back the HBM scratch that reads_run / kraken_run keep between runs of this process (sh_release_cached_ctx)."""
    check(load().sh_release_cached_ctx())


def reads_run(inputs, outputs, index, preset=None, extract=False, json=None, read_ids=None, command="", threads=4, device=0):
    """cleaner.rs:443-575 + clean_reads + ScrubbyReport: the whole `scrubby reads` mm2 path on the GPU."""
    require_gpu()
    c = ReadsConfig()
    for i, (a, b) in enumerate(zip(inputs, outputs)):
        c.input[i] = os.fsencode(a)
        c.output[i] = os.fsencode(b)
    c.n_files, c.extract, c.index = len(inputs), int(extract), os.fsencode(index)
    c.preset = preset.encode() if preset else None
    c.json = os.fsencode(json) if json else None
    c.read_ids = os.fsencode(read_ids) if read_ids else None
    c.command, c.threads, c.device = command.encode(), threads, device
    r = ReadsResult()
    check(load().sh_reads_run(C.byref(c), C.byref(r)))
    return {n: getattr(r, n) for n, _ in ReadsResult._fields_}


def classifier_taxids(report, taxa=(), taxa_direct=()):
    """classifier.rs:124-252 get_taxids_from_report -> set of taxid strings."""
    t = (C.c_char_p * max(len(taxa), 1))(*[x.encode() for x in taxa])
    d = (C.c_char_p * max(len(taxa_direct), 1))(*[x.encode() for x in taxa_direct])
    out = C.create_string_buffer(1 << 20)
    n = C.c_uint64()
    check(load().sh_classifier_taxids(os.fsencode(report), t, len(taxa), d, len(taxa_direct), out, len(out), C.byref(n)))
    return set(x for x in out.value.decode().split("\n") if x != "") if n.value else set()


def classifier_run(inputs, outputs, report, reads, classifier, taxa=(), taxa_direct=(), extract=False, json=None, read_ids=None, command=""):
    """`scrubby classifier`: cleaner.rs:177-194 run_classifier_output."""
    c = ClassifierConfig()
    for i, (a, b) in enumerate(zip(inputs, outputs)):
        c.input[i] = os.fsencode(a)
        c.output[i] = os.fsencode(b)
    c.n_files, c.extract = len(inputs), int(extract)
    c.report, c.reads, c.classifier = os.fsencode(report), os.fsencode(reads), classifier.encode()
    t = (C.c_char_p * max(len(taxa), 1))(*[x.encode() for x in taxa])
    d = (C.c_char_p * max(len(taxa_direct), 1))(*[x.encode() for x in taxa_direct])
    c.taxa, c.n_taxa, c.taxa_direct, c.n_taxa_direct = t, len(taxa), d, len(taxa_direct)
    c.json = os.fsencode(json) if json else None
    c.read_ids = os.fsencode(read_ids) if read_ids else None
    c.command = command.encode()
    r = ReadsResult()
    check(load().sh_classifier_run(C.byref(c), C.byref(r)))
    return {n: getattr(r, n) for n, _ in ReadsResult._fields_}


def alignment_run(inputs, outputs, alignment, fmt=None, min_len=0, min_cov=0.0, min_mapq=0, extract=False, json=None, read_ids=None, command=""):
    """`scrubby alignment`: cleaner.rs:206-219 run_aligner_output with alignment.rs filters."""
    c = AlignmentConfig()
    for i, (a, b) in enumerate(zip(inputs, outputs)):
        c.input[i] = os.fsencode(a)
        c.output[i] = os.fsencode(b)
    c.n_files, c.extract, c.alignment = len(inputs), int(extract), os.fsencode(alignment)
    c.format = fmt.encode() if fmt else None
    c.min_len, c.min_cov, c.min_mapq = min_len, min_cov, min_mapq
    c.json = os.fsencode(json) if json else None
    c.read_ids = os.fsencode(read_ids) if read_ids else None
    c.command = command.encode()
    r = ReadsResult()
    check(load().sh_alignment_run(C.byref(c), C.byref(r)))
    return {n: getattr(r, n) for n, _ in ReadsResult._fields_}


def synth_long_reads_device(P, R, r0, n_records, d_offsets, n_bases, out):
    """Long reads of the config-4 stand-in; d_offsets: int64 CUDA tensor [n_records+1] (lengths from the CPU twin)."""
    check(require_gpu().sh_synth_long_reads_device(C.byref(P), C.byref(R), r0, n_records, C.c_void_p(d_offsets.data_ptr()), n_bases,
                                                  C.c_void_p(out.data_ptr()), _stream_ptr()))
