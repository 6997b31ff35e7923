/*
 * scrubby_hip.h — C ABI of libscrubby_hip.so, the MI355X (gfx950) host-depletion backend.
 *
 * This is the drop-in boundary for the one hot path of esteinig/scrubby that this
 * library replaces: the in-process `mm2` aligner path
 *     Cleaner::run_minimap2_rs        /root/reference/src/cleaner.rs:443-575
 * whose arithmetic the reference reaches through the `minimap2` crate (FFI into
 * lh3/minimap2).  Every entry point below cites the reference interface it stands in
 * for; INTEGRATION.md shows the Rust `extern "C"` block a Scrubby maintainer would add.
 *
 * Conventions
 *   - plain C, no C++/torch types; every function returns an sh_status (0 = OK) and
 *     never throws or aborts across the boundary; sh_last_error() gives the message
 *     (thread-local), mirroring the `&'static str` errors minimap2-rs hands to
 *     ScrubbyError::Minimap2RustAlignerBuilderFailed / Minimap2RustAlignmentFailed
 *     (/root/reference/src/error.rs:147-152).
 *   - the caller owns every input/output buffer; the library owns sh_index / sh_ctx.
 *   - "_device" variants take HIP device pointers (inputs already resident in HBM) and
 *     a hipStream_t passed as void*; the others take host pointers and stage through
 *     HBM internally.
 *   - a read batch is the reference's Vec<(id, Vec<u8>)> (cleaner.rs:547-549) flattened:
 *     `bases` = concatenated ASCII sequence bytes, `offsets[n_reads+1]` = start of each
 *     read in `bases`.  Paired-end input is simply two records per pair: the reference
 *     maps each mate independently as single-end (cleaner.rs:499-501,527-529).
 */
#ifndef SCRUBBY_HIP_H
#define SCRUBBY_HIP_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SH_VERSION 104   /* 0.1.3: sh_index_replicate / sh_classify_sharded, sh_ctx_debug_list 3..10, the long join's tree in LDS (round 5); 0.1.2: sh_stats grew n_locus_* / n_rmq_exact (round 4); 0.1.1: sh_opts grew the rmq_* fields (round 3); sh_trace has 12 words since 0.1.0's second round */

typedef int32_t sh_status;
enum {
    SH_OK = 0,
    SH_ERR_BAD_ARG = 1,        /* null pointer, unsupported k/w, ... */
    SH_ERR_PRESET_UNKNOWN = 2,
    SH_ERR_PRESET_UNSUPPORTED = 3, /* Preset::Lr -> ScrubbyError::Minimap2PresetNotSupported (cleaner.rs:469) */
    SH_ERR_NO_DEVICE = 4,
    SH_ERR_OOM = 5,
    SH_ERR_HIP = 6,
    SH_ERR_IO = 7,
    SH_ERR_EMPTY_READ = 8,     /* minimap2-rs: Err("Sequence is empty") -> Minimap2RustAlignmentFailed; aborts the run (cleaner.rs:552,566) */
    SH_ERR_INDEX = 9           /* index build failed -> Minimap2RustAlignerBuilderFailed (cleaner.rs:480-482) */
};

/* Mapping options = the subset of minimap2's mm_idxopt_t/mm_mapopt_t that decides
 * `mappings.len() > 0`; filled by sh_preset() the way Aligner::builder().sr() / .map_ont()
 * ... fill them (cleaner.rs:455-470).  Layout is shared with the device code. */
typedef struct sh_opts {
    int32_t k, w;
    int32_t is_sr;
    int32_t mid_occ;          /* <= 0: derived from the index (mm_mapopt_update) */
    int32_t max_occ;
    int32_t max_max_occ;
    int32_t occ_dist;
    int32_t min_mid_occ, max_mid_occ;
    float   mid_occ_frac;
    float   q_occ_frac;
    int32_t min_cnt, min_chain_score;
    int32_t max_gap, max_gap_ref, max_frag_len, bw;
    int32_t max_chain_skip, max_chain_iter;
    float   chain_gap_scale, chain_skip_scale;
    /* the base-level extension stage that `.with_cigar()` switches on (cleaner.rs:473): with
     * SH_F_CIGAR a read only counts as mapped when a region of one of its chains survives the banded alignment and
     * minimap2's mm_filter_regs (cnt >= min_cnt, mlen >= min_chain_score, dp_max >= min_dp_max).  sh_preset sets it. */
    int32_t flags;
    int32_t a, b, q, e, q2, e2, sc_ambi;          /* ksw2 scores: match, mismatch, gap open / extend (two affine pieces), N */
    int32_t zdrop, zdrop_inv, end_bonus, min_dp_max;
    int32_t best_n, bw_long, min_ksw_len;
    float   pri_ratio, mask_level, max_clip_ratio;
    /* long-read presets: minimap2 re-chains a read whose first chaining pass left more than one chain with mg_lchain_rmq
     * (mm_map_frag: bw_long > bw), unless the first chain is small on a short read (rmq_rescue_*) */
    int32_t rmq_inner_dist, rmq_size_cap, rmq_rescue_size;
    float   rmq_rescue_ratio;
} sh_opts;
#define SH_F_CIGAR 1

/* Per-read decision trace (optional output; used by the parity tests). */
typedef struct sh_trace {
    int32_t n_mini, n_seed, n_anchor, rep_len, rechained, n_chain, best_score, flag;
    /* extension stage (0 without SH_F_CIGAR or without a chain): regions aligned, regions left (= mappings.len()),
     * their largest dp_max, a fingerprint of their coordinates / mlen / blen / dp_max / cnt */
    int32_t n_aligned, n_regs, dp_max; uint32_t sig;
} sh_trace;

typedef struct sh_index_info {
    int32_t  k, w;
    int32_t  mid_occ;          /* resolved value (sr: 1000; map-ont: from index) */
    uint32_t n_contigs;
    uint64_t n_bases;
    uint64_t n_minimizers;     /* stored (hash,pos) pairs */
    uint64_t n_keys;           /* distinct minimizer hashes */
    uint64_t n_slots;          /* 16-B table slots */
    uint64_t n_positions;      /* entries of the multi-occurrence position array */
    uint64_t hbm_bytes;        /* table + positions resident in HBM */
    double   build_ms;
} sh_index_info;

typedef struct sh_stats {
    uint64_t n_reads;
    uint64_t n_host;           /* flags == 1 */
    uint64_t n_no_seed;        /* decided by the sketch/probe kernel alone */
    uint64_t n_chain_small;    /* reads chained in LDS (lane per read) */
    uint64_t n_chain_large;    /* reads chained in the HBM arena */
    uint64_t n_minimizers;     /* sum of sketch sizes = table probes issued */
    uint64_t n_bases;
    double   ms_sketch_probe;  /* HIP-event time of each stage, summed over chunks */
    double   ms_chain_small;
    double   ms_chain_large;
    double   ms_total;         /* first kernel start -> last kernel end */
    uint64_t n_anchors;        /* anchors the occurrence filter admits on the repeat path (both passes); flag-only calls generate fewer (n_pair_decided) */
    uint64_t n_clusters;       /* independent anchor clusters chained on the repeat path */
    uint64_t n_resketch;       /* reads that took the legacy re-sketch path */
    uint64_t n_pair_decided;   /* flag-only calls: reads (LDS path and repeat path) decided by two co-diagonal seeds, before any anchor exists */
    uint64_t n_ext_reads;      /* SH_F_CIGAR: reads whose chains went through the extension stage (not decided by a shortcut) */
    uint64_t n_ext_regions;    /* regions aligned for them */
    uint64_t n_ext_dropped;    /* reads with a chain but no surviving region: the flags this stage flips */
    double   ms_ext;           /* HIP-event time of the extension stage */
    uint64_t n_ext_shortcut;   /* SH_F_CIGAR flag-only: reads decided inside a chaining kernel - their top chain's max stretch alone passes mm_filter_regs */
    uint64_t n_ext_fallback;   /* SH_F_CIGAR flag-only, sr: reads whose regs[0] did not survive and that were re-chained with every chain kept */
    double   ms_ext_fallback;  /* wall time of that fallback (re-chaining + the complete procedure) */
    uint64_t n_ext_unresolved; /* reads the extension stage left at their chain-level answer, mapped (see the warning): the device had no memory left for them, or - long reads - the long join's inner window (1000 reference bases) outgrew the 4096-anchor LDS ring (chained exactly, at seconds per read, with SCRUBBY_HIP_RMQ_ONE_LANE=1) */
    uint64_t n_rmq_rechained;  /* long-read presets: reads re-chained by the RMQ long join */
    uint64_t n_rmq_tied;       /* ... of which met candidates of equal priority in the join in a way that can change the chains (ties that cannot are recognised and pass) */
    uint64_t n_dp_parallel;    /* repeat-path reads whose mg_lchain_dp ran as the parallel recurrence */
    uint64_t n_dp_dirty;       /* ... their anchors that broke its premise (their clusters were chained by the sequential code) */
    uint64_t n_top_settled;    /* ... reads whose candidates for regs[0] were read off the whole read, no cluster visited */
    uint64_t n_locus_reads;    /* long-read presets, flag-only: reads chained over the reference windows that can hold regs[0] only */
    uint64_t n_locus_redone;   /* ... of which the answer could depend on what was left out: redone with every anchor */
    uint64_t n_rmq_exact;      /* long-read presets: reads whose RMQ join was redone on the literal krmq tree (tied priorities, windows beyond the LDS ring, rmq_size_cap) */
    uint64_t n_rmq_open;       /* ... reads that met such a tie and kept the scan's choice (smallest index): 0 by default since round 5 (every tied read takes the tree); only with SCRUBBY_HIP_RMQ_EXACT_MAX = N >= 0, for reads of more than N chain anchors */
    uint64_t n_ext_ondemand;   /* reads beyond the extension stage's prepared working-memory sizes, redone with memory allocated for them (visits, both kernels) */
} sh_stats;

typedef struct sh_index sh_index;
typedef struct sh_ctx sh_ctx;

/* ---- library ------------------------------------------------------------------------ */
int32_t     sh_version(void);
int32_t     sh_device_count(void);
const char *sh_last_error(void);

/* Aligner::builder().<preset>()  (cleaner.rs:453-470); name = Preset's Display form
 * ("sr", "map-ont", "lr:hq", ...; /root/reference/src/scrubby.rs:137-155). */
sh_status sh_preset(const char *name, sh_opts *out);

/* ---- index:  .with_index_threads(t).with_index(path, None)  (cleaner.rs:472-482) ----- */
/* Build from host sequences (already parsed FASTA records). */
sh_status sh_index_build(const uint8_t *const *seqs, const uint64_t *lens, uint32_t n_seq,
                         const sh_opts *opts, int32_t device, sh_index **out);
/* Build from a reference already resident in HBM: `d_bases` = concatenated ASCII contigs,
 * `contig_starts[n_contigs+1]` (host array) = start of each contig in d_bases. */
sh_status sh_index_build_device(const uint8_t *d_bases, const uint64_t *contig_starts, uint32_t n_contigs,
                                const sh_opts *opts, int32_t device, void *stream, sh_index **out);
/* Build from a FASTA file (plain or gzip); the reference rebuilds the index from FASTA on
 * every run (cleaner.rs:475-479). */
sh_status sh_index_build_fasta(const char *path, const sh_opts *opts, int32_t device, sh_index **out);
/* Binary cache of a built index (the reference passes output=None). */
sh_status sh_index_save(const sh_index *idx, const char *path);
sh_status sh_index_load(const char *path, int32_t device, sh_index **out);
sh_status sh_index_info_get(const sh_index *idx, sh_index_info *out);
/* Copy table and position array to host (tests, CPU baseline).  Either pointer may be NULL. */
sh_status sh_index_export(const sh_index *idx, uint64_t *slots /* 2*n_slots */, uint64_t *positions);
/* The reference bases the extension stage aligns against (minimap2 keeps them in the index too: mi->S), as resident in HBM:
 * 4-bit nt4 codes (A C G T = 0..3, anything else 4), two per byte, low nibble = even position, contigs back to back;
 * packed[(n_bases + 1) / 2], contig_start[n_contigs + 1].  Either pointer may be NULL. */
sh_status sh_index_export_ref(const sh_index *idx, uint8_t *packed, uint64_t *contig_start);
sh_status sh_index_free(sh_index *idx);

/* ---- classification:  aligner.map(&seq,false,false,None,None) -> len()>0  (cleaner.rs:550-558) --- */
/* A context owns the stream-ordered scratch (seed records, work lists, chain arena) for
 * batches of up to max_reads reads / max_bases bases; it plays the role of minimap2's
 * thread-local mm_tbuf_t.  One context per host thread; contexts share the immutable index. */
sh_status sh_ctx_create(const sh_index *idx, const sh_opts *opts, uint64_t max_reads, uint64_t max_bases,
                        uint32_t max_read_len, sh_ctx **out);
sh_status sh_ctx_destroy(sh_ctx *ctx);
/* Measurement aid (bench.py's stratified parity sample): which reads the context classified took the rarer paths.
 * Short-read extension stage, reads of the LAST CHUNK (ordinals within it): which = 0: re-chained with max_occ (mm_map_frag's second
 * pass), 1: regs[0] aligned base by base, 2: the complete procedure over every chain.
 * Long-read extension stage, reads of the LAST CALL (ordinals within it): which = 3: long join redone with the literal krmq tree beside the
 * scan (a tie that matters, or a window beyond the first ring), 4: tie left open (only with SCRUBBY_HIP_RMQ_EXACT_MAX >= 0), 5: left at the
 * chain-level answer (sh_stats.n_ext_unresolved), 6: redone with every anchor after the locus selection, 7: working memory allocated on
 * demand, 8: probe undecided - the complete procedure answered, 9: second working-memory size, 10: long join on the one-lane trees.
 * *n_out = their number (out may be NULL). */
sh_status sh_ctx_debug_list(const sh_ctx *ctx, int32_t which, uint32_t *out, uint64_t cap, uint64_t *n_out);

/* Inputs and outputs in HBM.  d_flags[r] = 1 host (>=1 mapping), 0 retained, 2 empty read.
 * d_trace may be NULL.  Asynchronous on `stream`; stats (nullable) forces a stream sync. */
sh_status sh_classify_device(sh_ctx *ctx, const uint8_t *d_bases, const uint64_t *d_offsets, uint64_t n_reads,
                             uint64_t n_bases, uint8_t *d_flags, sh_trace *d_trace, void *stream, sh_stats *stats);

/* Host buffers in, host flags out.  The batch goes up in pieces of 4 Mi reads from the calling thread while a worker
 * thread classifies the pieces that have arrived (20 M x 150 bp from pageable memory: 0.29 s with the extension filter on, 0.08 s at chain level, PCIe included).  The
 * context, device buffers and streams a call needs stay with the index for the next call (any thread; concurrent calls
 * each get their own set) and are released by sh_index_free.
 * Returns SH_ERR_EMPTY_READ (after filling flags) if any read is empty, as the reference's
 * per-read Err aborts the run. */
sh_status sh_classify_batch(const sh_index *idx, const sh_opts *opts, const uint8_t *bases, const uint64_t *offsets,
                            uint64_t n_reads, uint8_t *out_flags, sh_trace *out_trace, sh_stats *stats);

/* ---- in-process multi-GPU (SURVEY.md 8(b) "Threading", 8(e)) -------------------------------------------------------
 * The reference shares ONE &Aligner among all rayon workers (cleaner.rs:546-559).  A set holds one replica of the index per
 * shard: devices[i] is shard i's device (NULL / 0: every visible device, one shard each; a device may be listed more than
 * once - logical shards); the source index is borrowed for its own device (free the set first), the other devices get
 * copies (device to device where the devices are peers), freed with the set. */
typedef struct sh_index_set sh_index_set;
sh_status sh_index_replicate(const sh_index *idx, const int32_t *devices, uint32_t n_devices, sh_index_set **out);
uint32_t  sh_index_set_size(const sh_index_set *set);
sh_status sh_index_set_free(sh_index_set *set);
/* sh_classify_batch over all replicas at once: the batch is cut into contiguous shards at even record ordinals (mates stay
 * together), balanced by bases (= by count for fixed-length records); each shard is classified on its device by a host thread
 * of its own; the flags land in the caller's array - disjoint ranges, so the union of cleaner.rs:564-570 is a concatenation
 * and needs no collective.  shard_first (nullable, n_shards + 1 entries) receives the first record of each shard.  stats:
 * counters summed, stage times the largest over the devices.  Errors as sh_classify_batch; the message names the shard. */
sh_status sh_classify_sharded(const sh_index_set *set, const sh_opts *opts, const uint8_t *bases, const uint64_t *offsets,
                              uint64_t n_reads, uint8_t *out_flags, sh_trace *out_trace, sh_stats *stats, uint64_t *shard_first);

/* ---- the whole replaced path:  Cleaner::run_minimap2_rs + clean_reads + ScrubbyReport  ------------------------
 * (/root/reference/src/cleaner.rs:443-575, :236-254, :731-760; src/report.rs:24-57; src/utils.rs:250-285)
 * FASTA/FASTQ (plain, gzip, bzip2 or xz: sniffed by magic bytes like needletail / niffler do) in, filtered files out (the container by the
 * output's extension: utils.rs:28-36), optional JSON report and TSV of removed ids.
 * This is what `scrubby reads -i R1 [R2] -o O1 [O2] -I ref.fa [-p sr] [-e] [-j report.json] [-r ids.tsv]` runs. */
typedef struct sh_reads_config {
    const char *input[2];
    const char *output[2];
    uint32_t    n_files;      /* 1 (single-end / long reads) or 2 (paired-end) */
    int32_t     extract;      /* -e: write the mapped reads instead of the unmapped ones */
    const char *index;        /* -I: reference FASTA(.gz), or a .shidx cache written by sh_index_save */
    const char *preset;       /* -p: NULL -> "sr" for two files, "map-ont" for one (scrubby.rs:935-951) */
    const char *json;         /* -j, nullable */
    const char *read_ids;     /* -r, nullable */
    const char *command;      /* argv joined by ' ' (terminal.rs:178), for the report */
    int32_t     threads;      /* -t: host threads of the parse / filter / deflate stages (> 0: as given; <= 0: the CPUs this process may use - hardware threads capped by the cgroup quota and by 64) */
    int32_t     device;
} sh_reads_config;

typedef struct sh_reads_result {
    uint64_t reads_in, reads_out, reads_removed, reads_extracted;   /* as in the JSON report (records over both files) */
    uint64_t n_depleted_ids;                                        /* size of the HashSet<String> of cleaner.rs:564-570 */
    double   ms_index, ms_ingest, ms_classify, ms_write;
    uint64_t n_ext_unresolved;   /* reads the extension stage left at their chain-level answer (mapped); sh_stats of the same name, summed over the run's calls */
    uint64_t n_rmq_open;         /* long reads whose long join met a tie that matters beyond SCRUBBY_HIP_RMQ_EXACT_MAX anchors (sh_stats.n_rmq_open) */
} sh_reads_result;

sh_status sh_reads_run(const sh_reads_config *cfg, sh_reads_result *out);
/* sh_reads_run / sh_kraken_run keep the classification context of the last run (tens of GB of HBM scratch) for the next run of the
 * process, because handing such memory back and asking for it again costs seconds (the driver wipes it).  A long-lived embedding
 * process gets that memory back with this call; SCRUBBY_HIP_CTX_CACHE=0 in the environment switches the caching off.  Returns SH_OK. */
sh_status sh_release_cached_ctx(void);

/* ---- `scrubby classifier`: cleaning from precomputed Kraken2 / Metabuli outputs ------------------------------------
 * Cleaner::run_classifier_output (cleaner.rs:177-194) with the taxid decision rule of src/classifier.rs:
 * get_taxids_from_report (:124-252), get_tax_level (:345-373), get_taxid_reads_kraken / _metabuli (:270-320).
 * String / integer logic only: no GPU involved. */
typedef struct sh_classifier_config {
    const char *input[2];
    const char *output[2];
    uint32_t    n_files;
    int32_t     extract;
    const char *report;             /* -k: Kraken-style report */
    const char *reads;              /* -j: per-read classifications */
    const char *classifier;         /* -c: "kraken2" | "metabuli" */
    const char *const *taxa;        /* -T: names or taxids; sub-levels with direct reads are included */
    uint32_t    n_taxa;
    const char *const *taxa_direct; /* -D: names or taxids taken as they are */
    uint32_t    n_taxa_direct;
    const char *json;               /* --json, nullable */
    const char *read_ids;           /* -r, nullable */
    const char *command;
} sh_classifier_config;

sh_status sh_classifier_run(const sh_classifier_config *cfg, sh_reads_result *out);
/* the taxid set alone: sorted, newline-separated into `out` */
sh_status sh_classifier_taxids(const char *report, const char *const *taxa, uint32_t n_taxa, const char *const *taxa_direct,
                               uint32_t n_direct, char *out, size_t cap, uint64_t *n_out);

/* ---- `scrubby alignment`: cleaning from a precomputed alignment (PAF/GAF, or a one-column TXT of read ids) ---------
 * Cleaner::run_aligner_output (cleaner.rs:206-219), ReadAlignment::from / from_paf / from_txt (alignment.rs:33-114):
 * a read is selected iff (query aligned length >= min_len OR query coverage >= min_cov) AND mapq >= min_mapq.
 * SAM/BAM/CRAM need the reference's optional htslib feature and are rejected here. */
typedef struct sh_alignment_config {
    const char *input[2];
    const char *output[2];
    uint32_t    n_files;
    int32_t     extract;
    const char *alignment;    /* -a */
    const char *format;       /* "paf" | "gaf" | "txt", or NULL: by the file's (last) extension */
    uint64_t    min_len;      /* -l */
    double      min_cov;      /* -c */
    uint32_t    min_mapq;     /* -q */
    const char *json, *read_ids, *command;
} sh_alignment_config;

sh_status sh_alignment_run(const sh_alignment_config *cfg, sh_reads_result *out);

/* host-side pieces on their own (no GPU needed): get_id (utils.rs:91-103), FastqCleaner::clean_reads
 * (cleaner.rs:731-760), ReadDifference::get_difference (utils.rs:250-285) */
sh_status sh_host_get_id(const char *header, char *out, size_t cap);
sh_status sh_host_filter_fastx(const char *in, const char *out, const char *const *ids, uint64_t n_ids, int32_t extract,
                               uint64_t *n_in, uint64_t *n_out);
sh_status sh_host_read_difference(const char *const *inputs, const char *const *outputs, uint32_t n, uint64_t *reads_in,
                                  uint64_t *reads_out, uint64_t *difference);
/* the chunked, multi-threaded form of the same filter that sh_reads_run uses (pass 2 of csrc/sh_stream.cpp): the input is cut
 * into chunks of ~chunk_bytes at record boundaries, `threads` workers filter (and deflate, for .gz outputs), one writer
 * write side by side; retain: 0 = stream the file, 1 = keep the parsed chunks of the sequential reader, 2 = of the boundary-guessing
 * reader, 3 = of several byte-range readers (pass 1's forms).  Same bytes out as
 * sh_host_filter_fastx for plain outputs; .gz outputs are multi-member gzip with the same decompressed content. */
sh_status sh_host_filter_fastx_stream(const char *in, const char *out, const char *const *ids, uint64_t n_ids, int32_t extract,
                                      uint64_t chunk_bytes, int32_t threads, int32_t retain, uint64_t *n_in, uint64_t *n_out);

/* ---- synthetic workload (bench/test utility) ---------------------------- */
/* params structs are syn_ref_params / syn_read_params of csrc/sh_synth_core.h, passed opaquely */
sh_status sh_synth_ref_device(const void *ref_params, uint64_t g0, uint64_t n, uint8_t *d_out, void *stream);
sh_status sh_synth_reads_device(const void *ref_params, const void *read_params, uint64_t r0, uint64_t n_records,
                                uint8_t *d_out, uint64_t *d_offsets /* n_records+1, may be NULL */, void *stream);

/* variable-length long reads (BASELINE config 4 stand-in); d_offsets[n_records+1] given by the caller (lengths: syn_long_len) */
sh_status sh_synth_long_reads_device(const void *ref_params, const void *read_params, uint64_t r0, uint64_t n_records,
                                     const uint64_t *d_offsets, uint64_t n_bases, uint8_t *d_out, void *stream);

/* ---- Kraken2-style taxid classifier (BASELINE configs[4]) ------------------------
 * Replaces the external process of Cleaner::run_kraken (/root/reference/src/cleaner.rs:288-330):
 *     kraken2 --threads T --db DB [--paired] IN.. --output kraken.reads --report kraken.report
 * The database (compact hash table of 32-bit cells + taxonomy) is resident in HBM; reads are scanned for (k, l)
 * minimizers, each distinct consecutive minimizer is probed once, every k-mer counts for its minimizer's taxon and the
 * call is Kraken2's ResolveTree.  File formats of a database directory: hash.k2d, opts.k2d, taxo.k2d. */
typedef struct sh_k2_db sh_k2_db;

typedef struct sh_k2_opts {
    int32_t  k, l;                  /* 35, 31 */
    uint64_t spaced_seed_mask;      /* --minimizer-spaces 7 */
    uint64_t toggle_mask;           /* 0xe37e28c4271b5a2d */
    uint64_t min_acceptable_hash;   /* down-sampled databases ("standard-8"): minimizers hashing below it are not looked up */
    int32_t  value_bits;            /* low bits of a cell = internal taxid (set from hash.k2d on open) */
    int32_t  min_hit_groups;        /* --minimum-hit-groups, default 2 */
    double   confidence;            /* --confidence, default 0 (a double, as Kraken 2 parses it: ceil(c * kmers) decides) */
} sh_k2_opts;

typedef struct sh_k2_taxnode {      /* taxo.k2d node: 7 x u64 */
    uint64_t parent, first_child, child_count, name_offset, rank_offset, external_id, godparent;
} sh_k2_taxnode;

typedef struct sh_k2_result {
    uint32_t taxid;         /* external (NCBI) taxid of the call, 0 = unclassified */
    uint32_t call;          /* internal taxid */
    uint32_t total_kmers;   /* k-mers of the read / pair (ambiguous ones included) */
    uint32_t hit_groups;    /* distinct consecutive minimizers found in the table */
} sh_k2_result;

typedef struct sh_k2_info {
    uint64_t capacity, size, n_nodes, hbm_bytes;
    int32_t  k, l, value_bits, key_bits;
} sh_k2_info;

typedef struct sh_k2_stats {
    uint64_t n_units, n_classified, n_probes, n_kmers, n_overflow;
    float    ms_classify, ms_total;
} sh_k2_stats;

sh_status sh_k2_default_opts(sh_k2_opts *out);
/* open a Kraken2 database directory (hash.k2d, opts.k2d, taxo.k2d) into HBM */
sh_status sh_k2_open(const char *dbdir, int device, sh_k2_db **out);
/* an empty table of `capacity` cells over a caller-supplied taxonomy (synthetic databases, tests).  Node 0 is the
 * unused sentinel, node 1 the root; parents precede children (breadth-first ids). */
sh_status sh_k2_create(const sh_k2_opts *opts, uint64_t capacity, const sh_k2_taxnode *nodes, uint64_t n_nodes,
                       const char *names, uint64_t names_len, const char *ranks, uint64_t ranks_len, int device, sh_k2_db **out);
/* insert (minimizer, internal taxid) pairs that are resident in HBM; a key already present keeps the LCA */
sh_status sh_k2_insert_device(sh_k2_db *db, const uint64_t *d_keys, const uint32_t *d_taxa, uint64_t n, void *stream);
/* insert every minimizer of a device-resident sequence [d_bases, d_bases + n) with one taxid (down-sampled by
 * opts.min_acceptable_hash); *n_inserted (nullable) receives the number of minimizer runs inserted */
sh_status sh_k2_insert_sequence_device(sh_k2_db *db, const uint8_t *d_bases, uint64_t n, uint32_t taxon, void *stream, uint64_t *n_inserted);
/* pseudo-random filler keys (synthetic databases): n keys derived from `seed`, taxa uniform in [taxon_lo, taxon_hi] */
sh_status sh_k2_insert_random(sh_k2_db *db, uint64_t seed, uint64_t n, uint32_t taxon_lo, uint32_t taxon_hi, void *stream);
sh_status sh_k2_save(const sh_k2_db *db, const char *dbdir);
sh_status sh_k2_info_get(const sh_k2_db *db, sh_k2_info *out);
/* the option set stored with the database (k, l, masks, down-sampling threshold) plus the default thresholds */
sh_status sh_k2_db_opts(const sh_k2_db *db, sh_k2_opts *out);
/* host copies for the test oracle: cells[capacity], parent[n_nodes], external[n_nodes] (any pointer may be NULL) */
sh_status sh_k2_export(const sh_k2_db *db, uint32_t *cells, uint32_t *parent, uint32_t *external);
sh_status sh_k2_free(sh_k2_db *db);
/* classify device-resident records.  paired != 0: records 2i and 2i+1 are the mates of pair i (n_records even), one
 * result per pair; else one result per record.  d_bases needs 8 readable bytes past the last base. */
sh_status sh_k2_classify_device(const sh_k2_db *db, const sh_k2_opts *opts, const uint8_t *d_bases, const uint64_t *d_offsets,
                                uint64_t n_records, int32_t paired, sh_k2_result *d_out, void *stream, sh_k2_stats *stats);
/* the same from host memory */
sh_status sh_k2_classify_batch(const sh_k2_db *db, const sh_k2_opts *opts, const uint8_t *bases, const uint64_t *offsets,
                               uint64_t n_records, int32_t paired, sh_k2_result *out, sh_k2_stats *stats);
/* Kraken-style report (pct, clade reads, direct reads, rank code, taxid, indented name) from per-unit calls */
sh_status sh_k2_write_report(const sh_k2_db *db, const sh_k2_result *results, uint64_t n_units, const char *path);

/* `scrubby reads -c kraken2 -I DB -T .. -D ..`: Cleaner::run_kraken (cleaner.rs:288-330) in process: classify on the GPU,
 * write kraken.reads / kraken.report into workdir, then the taxid depletion of parse_classifier_output + clean_reads. */
typedef struct sh_kraken_config {
    const char *input[2];
    const char *output[2];
    uint32_t    n_files;            /* 2 = paired-end (kraken2 --paired) */
    int32_t     extract;
    const char *db;                 /* -I: database directory */
    const char *workdir;            /* -w, nullable: system temp dir */
    const char *const *taxa;        uint32_t n_taxa;
    const char *const *taxa_direct; uint32_t n_taxa_direct;
    double      confidence;         /* from -C "--confidence x"; < 0: default */
    int32_t     min_hit_groups;     /* <= 0: default */
    const char *json, *read_ids, *command;
    int32_t     device, threads;
    const char *classifier_args;    /* -C verbatim, nullable: echoed as settings.classifier_args in the JSON (report.rs:81) */
} sh_kraken_config;
sh_status sh_kraken_run(const sh_kraken_config *cfg, sh_reads_result *out);

/* ---- multi-GPU: the one exchange of the read-sharded path (HashSet union of cleaner.rs:564-570) --------------
 * d_flags[n] (1 = host) -> d_bits[(n + 7) / 8], bit i of byte j = record 8j + i; each rank packs its own slice and the disjoint
 * slices are all-gathered over RCCL (scrubby_amd/dist.py). */
sh_status sh_pack_flags_device(const uint8_t *d_flags, uint64_t n, uint8_t *d_bits, void *stream);

/* ---- micro-benchmarks for the roofline (bench.py) ---------------------------------------- */
/* random 16-B slot gathers over the index table; returns achieved GB/s of useful bytes */
sh_status sh_bench_gather(const sh_index *idx, uint64_t n_probes, int32_t iters, double *out_gbs_useful, double *out_ms);
/* Test aid: the krmq tree of the long join (csrc/sh_rmq_tree.h) on the device - one lane runs a random insert / erase / query sequence with
 * heavily tied priorities (the generator of oracle/mm_rmq.c's mmo_rmq_trace) and returns, per query, the element the tree answered with
 * (its i, or -1).  lds = 0: nodes in a pool in HBM (RqPool); lds = 1: the whole tree in LDS (RqLds, 4096 nodes: *n_out = -100 when the sequence
 * holds more at once), one lane; lds = 2: that tree with the insertions and erasures done by the whole wave (rq_insert_w / rq_erase_w).  out: host array of n_ops int64; *n_out < 0: a guard of the tree code tripped. */
sh_status sh_dbg_rmq_trace(int32_t device, uint64_t seed, int32_t n_ops, int32_t key_range, int32_t fifo, int32_t lds, int64_t *out, int64_t *n_out);
/* test aid: the wave primitives of csrc/sh_wave.h (DPP scans / reductions / broadcasts) on one wave of inputs: 12 x 64 int32 and 9 x 64 uint64
 * results in the order of k_dbg_wave_ops (tests/test_wave_ops_gpu.py compares them with numpy) */
sh_status sh_dbg_wave_ops(int32_t device, const int32_t *in32, const uint64_t *in64, int32_t bcast_lane, int32_t *out32, uint64_t *out64);

#ifdef __cplusplus
}
#endif
#endif
