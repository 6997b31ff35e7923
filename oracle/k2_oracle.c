/*
 * k2_oracle.c — CPU ORACLE of the Kraken2-style classifier (TEST INFRASTRUCTURE; see k2_oracle.h: PARITY UNPINNED).
 * Call site in the reference: /root/reference/src/cleaner.rs:288-330 (external `kraken2` process).
 * Algorithm restated from SURVEY.md Appendix B (Wood, Lu & Langmead 2019).
 */
#include "k2_oracle.h"
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define K2_AMBIG 0xFFFFFFFFu
#define K2_BORDER 0xFFFFFFFEu

void k2o_default_opts(k2o_opts *o)
{
    memset(o, 0, sizeof(*o));
    o->k = 35; o->l = 31;
    /* --minimizer-spaces 7: 34 one-bits, then 0011 x 7 (two bits per base, lowest positions alternate) */
    o->spaced_seed_mask = (0x3ffffffffULL << 28) | 0x3333333ULL;
    o->toggle_mask = 0xe37e28c4271b5a2dULL;
    o->min_acceptable_hash = 0;
    o->value_bits = 17;
    o->min_hit_groups = 2;
    o->confidence = 0.0;
}

/* MurmurHash3 64-bit finaliser */
uint64_t k2o_hash(uint64_t k)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL;
    k ^= k >> 33;
    return k;
}

static uint64_t revcomp(uint64_t x, int n)
{   /* 2-bit bases, A=0 C=1 G=2 T=3: complement = 3 - c, order reversed */
    uint64_t r = 0;
    for (int i = 0; i < n; ++i) { r = (r << 2) | (3 - (x & 3)); x >>= 2; }
    return r;
}

uint64_t k2o_canonical(uint64_t lmer, int l)
{
    uint64_t rc = revcomp(lmer, l);
    return lmer < rc ? lmer : rc;
}

static int base_code(uint8_t c)
{
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return -1;
    }
}

/* The scanner as a state machine that yields one entry per consumed character once k characters have been read
 * (SURVEY.md App. B "Minimizer"): a deque holds the candidate l-mers of the current window in non-decreasing order. */
typedef struct { uint64_t cand; int64_t pos; } qent;

int64_t k2o_scan(const uint8_t *seq, int64_t n, const k2o_opts *o, uint64_t *out_min, uint8_t *out_ambig, int64_t cap)
{
    const int k = o->k, l = o->l;
    const uint64_t lmask = l < 32 ? ((1ULL << (2 * l)) - 1) : ~0ULL;
    qent *q = (qent *)malloc(sizeof(qent) * (size_t)(k - l + 2));
    int qh = 0, qn = 0;                      /* ring deque: head index, count */
    const int qcap = k - l + 2;
    uint64_t lmer = 0, last_ambig = 0, last_min = ~0ULL;
    int loaded = 0;
    int64_t qpos = 0, pos = 0, n_out = 0;
    while (pos < n) {
        /* one character */
        if (loaded == l) loaded--;
        loaded++;
        lmer <<= 2; last_ambig <<= 2;
        int code = base_code(seq[pos++]);
        if (code < 0) { qn = 0; qh = 0; qpos = 0; lmer = 0; loaded = 0; last_ambig |= 3; }
        else lmer |= (uint64_t)code;
        lmer &= lmask; last_ambig &= lmask;
        if (loaded < l) {
            /* incomplete l-mer: once a full k-mer's worth of characters has been read, the k-mer is reported ambiguous */
            if (pos >= k && n_out < cap) { out_min[n_out] = last_min; out_ambig[n_out] = 1; ++n_out; }
            continue;
        }
        uint64_t canon = k2o_canonical(lmer, l);
        if (o->spaced_seed_mask) canon &= o->spaced_seed_mask;
        const uint64_t cand = canon ^ o->toggle_mask;
        if (k == l) {
            last_min = cand ^ o->toggle_mask;
        } else {
            while (qn > 0 && q[(qh + qn - 1) % qcap].cand > cand) --qn;
            q[(qh + qn) % qcap].cand = cand; q[(qh + qn) % qcap].pos = qpos; ++qn;
            if (q[qh].pos < qpos - k + l) { qh = (qh + 1) % qcap; --qn; }
            ++qpos;
            if (pos < k) continue;           /* not a full k-mer yet */
            last_min = q[qh].cand ^ o->toggle_mask;
        }
        if (pos >= k && n_out < cap) { out_min[n_out] = last_min; out_ambig[n_out] = last_ambig != 0; ++n_out; }
    }
    free(q);
    return n_out;
}

uint32_t k2o_cht_get(const uint32_t *cells, uint64_t capacity, int value_bits, uint64_t key)
{
    const uint64_t hc = k2o_hash(key);
    const uint32_t compacted = (uint32_t)(hc >> (32 + value_bits));
    const uint32_t vmask = (1u << value_bits) - 1;
    uint64_t idx = hc % capacity;
    const uint64_t first = idx;
    for (;;) {
        const uint32_t c = cells[idx];
        if (!(c & vmask)) return 0;                       /* value 0 = empty cell */
        if ((c >> value_bits) == compacted) return c & vmask;
        idx = idx + 1 == capacity ? 0 : idx + 1;          /* linear probing */
        if (idx == first) return 0;
    }
}

int k2o_cht_set(uint32_t *cells, uint64_t capacity, int value_bits, uint64_t key, uint32_t value, const uint32_t *parent)
{
    const uint64_t hc = k2o_hash(key);
    const uint32_t compacted = (uint32_t)(hc >> (32 + value_bits));
    const uint32_t vmask = (1u << value_bits) - 1;
    uint64_t idx = hc % capacity;
    const uint64_t first = idx;
    for (;;) {
        const uint32_t c = cells[idx];
        if (!(c & vmask)) { cells[idx] = compacted << value_bits | value; return 1; }
        if ((c >> value_bits) == compacted) {
            if (parent) cells[idx] = compacted << value_bits | k2o_lca(parent, c & vmask, value);
            return 1;
        }
        idx = idx + 1 == capacity ? 0 : idx + 1;
        if (idx == first) return 0;
    }
}

/* internal ids are assigned breadth-first, so a parent's id is smaller than its children's */
int k2o_is_ancestor(const uint32_t *parent, uint32_t a, uint32_t b)
{
    if (!a || !b) return 0;
    while (b > a) b = parent[b];
    return a == b;
}

uint32_t k2o_lca(const uint32_t *parent, uint32_t a, uint32_t b)
{
    if (!a || !b) return a ? a : b;
    while (a != b) { if (a > b) a = parent[a]; else b = parent[b]; }
    return a;
}

uint32_t k2o_resolve(const uint32_t *taxa, const uint32_t *counts, int n, const uint32_t *parent, uint32_t total_kmers, double confidence)
{
    uint32_t max_taxon = 0, max_score = 0;
    const uint32_t required = (uint32_t)ceil(confidence * (double)total_kmers);
    for (int i = 0; i < n; ++i) {
        uint32_t score = 0;
        for (int j = 0; j < n; ++j) if (k2o_is_ancestor(parent, taxa[j], taxa[i])) score += counts[j];
        if (score > max_score) { max_score = score; max_taxon = taxa[i]; }
        else if (score == max_score) max_taxon = k2o_lca(parent, max_taxon, taxa[i]);
    }
    max_score = 0;
    for (int i = 0; i < n; ++i) if (taxa[i] == max_taxon) max_score = counts[i];
    while (max_taxon && max_score < required) {
        max_score = 0;
        for (int i = 0; i < n; ++i) if (k2o_is_ancestor(parent, max_taxon, taxa[i])) max_score += counts[i];
        if (max_score >= required) return max_taxon;
        max_taxon = parent[max_taxon];
    }
    return max_taxon;
}

void k2o_classify_pair(const uint32_t *cells, uint64_t capacity, const uint32_t *parent, const k2o_opts *o,
                       const uint8_t *seq1, int64_t n1, const uint8_t *seq2, int64_t n2, k2o_result *res,
                       uint32_t *taxa_out, int64_t cap, int64_t *n_taxa_out)
{
    int64_t nt = 0;
    uint32_t total = 0, groups = 0, probes = 0;
    /* hit counts: small open list (reads hit a handful of taxa) */
    int n_hit = 0, hit_cap = 64;
    uint32_t *ht = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)hit_cap), *hc = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)hit_cap);
    const uint8_t *seqs[2] = {seq1, seq2};
    const int64_t lens[2] = {n1, n2};
    const int n_frag = seq2 ? 2 : 1;
    for (int f = 0; f < n_frag; ++f) {
        const int64_t n = lens[f];
        const int64_t mcap = n > 0 ? n : 1;
        uint64_t *mins = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)mcap);
        uint8_t *amb = (uint8_t *)malloc((size_t)mcap);
        const int64_t m = k2o_scan(seqs[f], n, o, mins, amb, mcap);
        uint64_t last_min = ~0ULL; uint32_t last_taxon = 0xFFFFFFFDu;
        for (int64_t i = 0; i < m; ++i) {
            uint32_t taxon;
            if (amb[i]) taxon = K2_AMBIG;
            else {
                if (mins[i] != last_min) {
                    int skip = 0;
                    if (o->min_acceptable_hash && k2o_hash(mins[i]) < o->min_acceptable_hash) skip = 1;
                    taxon = 0;
                    if (!skip) { taxon = k2o_cht_get(cells, capacity, o->value_bits, mins[i]); ++probes; }
                    last_taxon = taxon; last_min = mins[i];
                    if (taxon) ++groups;
                } else taxon = last_taxon;
                if (taxon) {
                    int j = 0;
                    while (j < n_hit && ht[j] != taxon) ++j;
                    if (j == n_hit) {
                        if (n_hit == hit_cap) { hit_cap *= 2; ht = (uint32_t *)realloc(ht, sizeof(uint32_t) * (size_t)hit_cap); hc = (uint32_t *)realloc(hc, sizeof(uint32_t) * (size_t)hit_cap); }
                        ht[n_hit] = taxon; hc[n_hit] = 0; ++n_hit;
                    }
                    ++hc[j];
                }
            }
            ++total;
            if (taxa_out && nt < cap) taxa_out[nt] = taxon;
            ++nt;
        }
        if (n_frag == 2 && f == 0) { if (taxa_out && nt < cap) taxa_out[nt] = K2_BORDER; ++nt; }
        free(mins); free(amb);
    }
    uint32_t call = k2o_resolve(ht, hc, n_hit, parent, total, o->confidence);
    if (call && groups < (uint32_t)o->min_hit_groups) call = 0;
    res->call = call; res->total_kmers = total; res->hit_groups = groups; res->n_probes = probes;
    if (n_taxa_out) *n_taxa_out = nt;
    free(ht); free(hc);
}

typedef struct {
    const uint32_t *cells; uint64_t capacity; const uint32_t *parent; const k2o_opts *o;
    const uint8_t *bases; const uint64_t *offsets; uint64_t n_units; int paired; k2o_result *res;
    int tid, nthr;
} k2o_job;

static void *k2o_worker(void *arg)
{
    k2o_job *j = (k2o_job *)arg;
    const uint64_t chunk = 256;
    for (uint64_t u0 = (uint64_t)j->tid * chunk; u0 < j->n_units; u0 += (uint64_t)j->nthr * chunk) {
        const uint64_t u1 = u0 + chunk < j->n_units ? u0 + chunk : j->n_units;
        for (uint64_t u = u0; u < u1; ++u) {
            if (j->paired) {
                const uint64_t a = j->offsets[2 * u], b = j->offsets[2 * u + 1], c = j->offsets[2 * u + 2];
                k2o_classify_pair(j->cells, j->capacity, j->parent, j->o, j->bases + a, (int64_t)(b - a), j->bases + b, (int64_t)(c - b),
                                  &j->res[u], NULL, 0, NULL);
            } else {
                const uint64_t a = j->offsets[u], b = j->offsets[u + 1];
                k2o_classify_pair(j->cells, j->capacity, j->parent, j->o, j->bases + a, (int64_t)(b - a), NULL, 0, &j->res[u], NULL, 0, NULL);
            }
        }
    }
    return NULL;
}

void k2o_classify_batch(const uint32_t *cells, uint64_t capacity, const uint32_t *parent, const k2o_opts *o,
                        const uint8_t *bases, const uint64_t *offsets, uint64_t n_records, int paired,
                        k2o_result *res, int n_threads)
{
    if (n_threads < 1) n_threads = 1;
    const uint64_t n_units = paired ? n_records / 2 : n_records;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    k2o_job *jobs = (k2o_job *)malloc(sizeof(k2o_job) * (size_t)n_threads);
    for (int t = 0; t < n_threads; ++t) {
        k2o_job jb = {cells, capacity, parent, o, bases, offsets, n_units, paired, res, t, n_threads};
        jobs[t] = jb;
        pthread_create(&th[t], NULL, k2o_worker, &jobs[t]);
    }
    for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
    free(th); free(jobs);
}
