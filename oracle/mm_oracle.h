/*
 * mm_oracle.h — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE)
 *
 * A plain-C restatement of the per-read "does this read map to the host
 * reference" decision that the reference reaches through
 *     aligner.map(&sequence, false, false, None, None) -> mappings.len() > 0
 * at /root/reference/src/cleaner.rs:550-558 (builder at :453-482).
 *
 * The arithmetic behind that call lives in a third-party dependency that is NOT
 * in /root/reference: crate `minimap2 = "0.1.20"` (Cargo.toml:41, no Cargo.lock)
 * -> minimap2-sys -> lh3/minimap2 (C, ~v2.28).  Nothing of it is on this box, so
 * this file restates minimap2's *published* algorithm (Li 2018, Bioinformatics
 * 34:3094; Li 2021, Bioinformatics 37:4572 for the chaining score) as specified
 * in SURVEY.md Appendix A.
 *
 *                       *** PARITY UNPINNED ***
 * The reference holds no tests, fixtures or golden vectors for this path
 *  and cannot be built or run here, so this oracle is pinned
 * only by self-authored known-answer tests (tests/golden/) and by an independent
 * brute-force re-derivation of the sketch in tests/.  Stages restated:
 *   A.2 minimizer sketch      (mmo_sketch)
 *   A.3 index                 (mmo_index_*)
 *   A.4 seed collection, occurrence filter, sr re-chain with max_occ
 *   A.5 chaining DP + backtrack; decision = "at least one chain kept"
 *   A.6 base-level extension stage and mm_filter_regs (mm_align.c): the short-read branch
 *       (MM_F_SR: preset sr) and the long-read branch (map-ont, lr:hq, map-hifi) with the
 *       RMQ re-chain (mm_rmq.c), mm_est_err and the strand filter that precede it
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (scrubby_amd/) never links or calls it.
 */
#ifndef MM_ORACLE_H
#define MM_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* option set; values per preset follow SURVEY.md App. A.1 */
typedef struct {
    int32_t k, w;
    int32_t is_sr;            /* MM_F_SR: query/ref gap rules of short-read mode */
    int32_t mid_occ;          /* <=0: derive from index with mid_occ_frac */
    int32_t max_occ;          /* sr: 5000; others 0 (no re-chain) */
    int32_t max_max_occ;      /* 4095 */
    int32_t occ_dist;         /* 500 */
    int32_t min_mid_occ, max_mid_occ;
    float   mid_occ_frac;     /* 2e-4 */
    float   q_occ_frac;       /* 0.01 */
    int32_t min_cnt, min_chain_score;
    int32_t max_gap, max_gap_ref, max_frag_len, bw;
    int32_t max_chain_skip, max_chain_iter;
    float   chain_gap_scale, chain_skip_scale;
    /* A.6, the base-level extension stage (mm_align.c); flags bit 0 = MM_F_CIGAR, which `.with_cigar()` sets
     * (/root/reference/src/cleaner.rs:473): with it the decision is "a region survives mm_filter_regs" */
    int32_t flags;
    int32_t a, b, q, e, q2, e2, sc_ambi;
    int32_t zdrop, zdrop_inv, end_bonus, min_dp_max;
    int32_t best_n, bw_long, min_ksw_len;
    float   pri_ratio, mask_level, max_clip_ratio;
    /* the long-join re-chain of mm_map_frag (bw_long > bw, more than one chain): mg_lchain_rmq, mm_rmq.c */
    int32_t rmq_inner_dist, rmq_size_cap, rmq_rescue_size;
    float   rmq_rescue_ratio;
} mmo_opts;
#define MMO_F_CIGAR 1

/* per-read trace of the decision, every field compared bit-exactly with the HIP path */
typedef struct {
    int32_t n_mini;      /* minimizers emitted by the sketch (after q_occ_frac thinning) */
    int32_t n_seed;      /* minimizers present in the index (mm_seed_collect_all) */
    int32_t n_anchor;    /* anchors entering the LAST chaining pass */
    int32_t rep_len;     /* repetitive query length of the last pass */
    int32_t rechained;   /* bit 0: the max_occ second pass ran (sr); bit 1: the RMQ long-join re-chain ran (long-read presets with MM_F_CIGAR) */
    int32_t n_chain;     /* chains kept by backtrack (n_regs0) */
    int32_t best_score;  /* max chain score among kept chains, 0 if none */
    int32_t flag;        /* 1 = host, 0 = retained, 2 = empty read (reference: Err).  host = n_chain > 0 without MM_F_CIGAR, n_regs > 0 with it */
    int32_t n_aligned;   /* A.6: regions entering mm_align_skeleton (after mm_set_parent / mm_select_sub); 0 without MM_F_CIGAR */
    int32_t n_regs;      /* A.6: regions left by mm_filter_regs = mappings.len() */
    int32_t dp_max;      /* A.6: largest dp_max among them */
    uint32_t sig;        /* A.6: fingerprint of the surviving regions (coordinates, mlen, blen, dp_max, cnt) */
} mmo_trace;

typedef struct mmo_index mmo_index;

int  mmo_preset(const char *name, mmo_opts *o);      /* 0 ok, -1 unknown, -2 "lr" unsupported */
uint64_t mmo_hash64(uint64_t key, uint64_t mask);

/* A.2: returns number of minimizers; x[i] = hash<<8|span, y[i] = rid<<32|pos<<1|strand.
 * If cap is exceeded the return value still counts all, but only cap are stored. */
int64_t mmo_sketch(const uint8_t *seq, int64_t len, int w, int k, uint32_t rid,
                   uint64_t *x, uint64_t *y, int64_t cap);

/* A.3 */
mmo_index *mmo_index_build(int n_seq, const uint8_t *const *seqs, const int64_t *lens, int w, int k);
/* wrap an index in the product's HBM layout (16-B slots + position array), copied to host */
mmo_index *mmo_index_wrap(const uint64_t *slots, uint64_t n_slots, const uint64_t *positions,
                          uint64_t n_positions, int w, int k);
/* the reference sequence a wrapped index aligns against (A.6): 4-bit packed nt4 codes (low nibble = even position) of all
 * contigs back to back, contig_start[n_contigs + 1]; borrowed, not copied.  mmo_index_build keeps its own. */
void mmo_index_set_ref(mmo_index *idx, const uint8_t *packed, const uint64_t *contig_start, uint32_t n_contigs);
const uint8_t *mmo_index_ref(const mmo_index *idx, const uint64_t **contig_start, uint32_t *n_contigs);
void mmo_index_free(mmo_index *idx);
const uint64_t *mmo_index_get(const mmo_index *idx, uint64_t minier, int32_t *n);
uint64_t mmo_index_n_keys(const mmo_index *idx);
uint64_t mmo_index_n_positions(const mmo_index *idx);
/* dump in canonical order (keys ascending; positions ascending per key) */
void mmo_index_dump(const mmo_index *idx, uint64_t *keys, uint32_t *counts, uint64_t *positions);
int32_t mmo_index_cal_mid_occ(const mmo_index *idx, float frac);
/* resolve o->mid_occ from the index when <= 0 (mm_mapopt_update) */
void mmo_opts_update(mmo_opts *o, const mmo_index *idx);

/* A.4 + A.5 for one read */
void mmo_map(const mmo_index *idx, const mmo_opts *o, const uint8_t *seq, int64_t len, mmo_trace *tr);

/* the batch form of cleaner.rs:546-559; n_threads workers over independent reads */
void mmo_classify_batch(const mmo_index *idx, const mmo_opts *o, const uint8_t *bases,
                        const uint64_t *offsets, uint64_t n_reads, uint8_t *flags,
                        mmo_trace *traces /* may be NULL */, int n_threads);

/* chaining pair score (exposed for unit tests of the float path) */
int32_t mmo_comput_sc(uint64_t ai_x, uint64_t ai_y, uint64_t aj_x, uint64_t aj_y, int32_t max_dist_x,
                      int32_t max_dist_y, int32_t bw, float chn_pen_gap, float chn_pen_skip);
float mmo_log2(float x);

#ifdef __cplusplus
}
#endif
#endif
