/*
 * k2_oracle.h — CPU ORACLE of the Kraken2-style taxid classifier (TEST INFRASTRUCTURE, NOT PRODUCT CODE)
 *
 * The reference reaches this path by running an external executable:
 *     kraken2 --db <index> --threads <t> --output kraken.reads --report kraken.report [--paired] <inputs>
 * at /root/reference/src/cleaner.rs:288-330 (`Cleaner::run_kraken`), then reads the two text files with
 * /root/reference/src/classifier.rs:124-290 (restated in the product's host code and tested separately).
 * The classification arithmetic is therefore in a third-party program that is NOT in /root/reference and not on
 * this box: `kraken2 >= 2.1.3` (conda.yml:11).  This file restates Kraken2's published algorithm
 * (Wood, Lu & Langmead 2019, Genome Biology 20:257) as specified in SURVEY.md Appendix B:
 *   - minimizer scanner (k = 35, l = 31, spaced-seed mask, toggle mask, ambiguous-base reset)   k2o_scan
 *   - MurmurHash3 fmix64 and the compact hash table probe (32-bit cells, linear probing)        k2o_hash / k2o_cht_get
 *   - per-fragment hit counting with consecutive-minimizer reuse and minimum_hit_groups         k2o_classify_pair
 *   - ResolveTree (root-to-leaf path scores, LCA of ties, confidence climb)                     k2o_resolve
 *   - taxonomy helpers over BFS-ordered internal ids                                            k2o_is_ancestor / k2o_lca
 *
 *                       *** PARITY UNPINNED ***
 * No kraken2 binary, source, database or output file exists in /root/reference or on this image, and the reference
 * holds no golden vectors for the path.  The restatement is pinned only by self-authored
 * known-answer tests (tests/golden/k2_kat.json) derived by hand / by an independent brute-force re-derivation.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#ifndef K2_ORACLE_H
#define K2_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int32_t k, l;
    uint64_t spaced_seed_mask, toggle_mask;
    uint64_t min_acceptable_hash;       /* 0: every minimizer is looked up */
    int32_t value_bits;                 /* low bits of a cell = internal taxid */
    int32_t min_hit_groups;             /* kraken2 --minimum-hit-groups (default 2) */
    double confidence;                  /* kraken2 --confidence (default 0) */
} k2o_opts;

typedef struct {
    uint32_t call;          /* internal taxid, 0 = unclassified */
    uint32_t total_kmers;   /* entries of the k-mer taxa list (both mates, without the pair border) */
    uint32_t hit_groups;    /* distinct consecutive minimizers found in the table */
    uint32_t n_probes;      /* table lookups performed */
} k2o_result;

void k2o_default_opts(k2o_opts *o);
uint64_t k2o_hash(uint64_t key);                                   /* MurmurHash3 fmix64 */
uint64_t k2o_canonical(uint64_t lmer, int l);

/* scanner: one entry per k-mer of seq; out_min[i] = minimizer (valid when out_ambig[i] == 0).  Returns the number of
 * entries (<= cap). */
int64_t k2o_scan(const uint8_t *seq, int64_t n, const k2o_opts *o, uint64_t *out_min, uint8_t *out_ambig, int64_t cap);

/* compact hash table over caller-owned cells */
uint32_t k2o_cht_get(const uint32_t *cells, uint64_t capacity, int value_bits, uint64_t key);
/* sequential insert (tests build small tables with it); returns 0 if the table is full.  An existing key keeps the
 * LCA of the old and the new value when parent != NULL, else the old value. */
int k2o_cht_set(uint32_t *cells, uint64_t capacity, int value_bits, uint64_t key, uint32_t value, const uint32_t *parent);

int k2o_is_ancestor(const uint32_t *parent, uint32_t a, uint32_t b);    /* is a an ancestor of (or equal to) b */
uint32_t k2o_lca(const uint32_t *parent, uint32_t a, uint32_t b);
uint32_t k2o_resolve(const uint32_t *taxa, const uint32_t *counts, int n, const uint32_t *parent, uint32_t total_kmers, double confidence);

/* one read (seq2 == NULL) or one pair.  taxa_out (optional, cap entries): the per-k-mer taxa list with
 * 0xFFFFFFFF = ambiguous, 0xFFFFFFFE = mate border; *n_taxa_out receives its length. */
void k2o_classify_pair(const uint32_t *cells, uint64_t capacity, const uint32_t *parent, const k2o_opts *o,
                       const uint8_t *seq1, int64_t n1, const uint8_t *seq2, int64_t n2, k2o_result *res,
                       uint32_t *taxa_out, int64_t cap, int64_t *n_taxa_out);

/* batch: records offsets[0..n_records]; paired != 0: records 2i, 2i+1 are mates -> n_records / 2 results */
void k2o_classify_batch(const uint32_t *cells, uint64_t capacity, const uint32_t *parent, const k2o_opts *o,
                        const uint8_t *bases, const uint64_t *offsets, uint64_t n_records, int paired,
                        k2o_result *res, int n_threads);

#ifdef __cplusplus
}
#endif
#endif
