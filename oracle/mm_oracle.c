/*
 * mm_oracle.c — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See mm_oracle.h.
 *
 * *** PARITY UNPINNED *** : restates lh3/minimap2 (~v2.28, reached by the reference via
 * crate minimap2 ^0.1.20, /root/reference/Cargo.toml:41, call sites
 * /root/reference/src/cleaner.rs:453-482 and :552) from its published algorithm and
 * SURVEY.md Appendix A.  Each function cites the appendix section it follows and the
 * reference call site that reaches it.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off: the chaining score has an f32
 * path that must not be fused if it is to match the HIP kernels bit for bit).
 */
#include "mm_oracle.h"
#include "mm_align.h"
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

/* ------------------------------------------------------------------------------------------
 * A.1 presets reached from /root/reference/src/cleaner.rs:455-470
 * ---------------------------------------------------------------------------------------- */
static void opts_default(mmo_opts *o)
{
    memset(o, 0, sizeof(*o));
    o->k = 15; o->w = 10;
    o->is_sr = 0;
    o->mid_occ = 0; o->max_occ = 0;
    o->max_max_occ = 4095; o->occ_dist = 500;
    o->min_mid_occ = 10; o->max_mid_occ = 1000000;
    o->mid_occ_frac = 2e-4f; o->q_occ_frac = 0.01f;
    o->min_cnt = 3; o->min_chain_score = 40;
    o->max_gap = 5000; o->max_gap_ref = -1; o->max_frag_len = 0; o->bw = 500;
    o->max_chain_skip = 25; o->max_chain_iter = 5000;
    o->chain_gap_scale = 0.8f; o->chain_skip_scale = 0.0f;
    /* mm_mapopt_init; .with_cigar() (cleaner.rs:473) sets MM_F_CIGAR on every preset */
    o->flags = MMO_F_CIGAR;
    o->a = 2; o->b = 4; o->q = 4; o->e = 2; o->q2 = 24; o->e2 = 1; o->sc_ambi = 1;
    o->zdrop = 400; o->zdrop_inv = 200; o->end_bonus = -1; o->min_dp_max = o->min_chain_score * o->a;
    o->best_n = 5; o->bw_long = 20000; o->min_ksw_len = 200;
    o->pri_ratio = 0.8f; o->mask_level = 0.5f; o->max_clip_ratio = 1.0f;
    o->rmq_inner_dist = 1000; o->rmq_size_cap = 100000; o->rmq_rescue_size = 1000; o->rmq_rescue_ratio = 0.1f;
}

int mmo_preset(const char *name, mmo_opts *o)
{
    opts_default(o);
    if (strcmp(name, "sr") == 0) {              /* Preset::Sr, cleaner.rs:456 */
        o->k = 21; o->w = 11; o->is_sr = 1;
        o->max_frag_len = 800; o->max_gap = 100; o->bw = 100;
        o->min_cnt = 2; o->min_chain_score = 25;
        o->mid_occ = 1000; o->max_occ = 5000;
        o->a = 2; o->b = 8; o->q = 12; o->e = 2; o->q2 = 24; o->e2 = 1;
        o->zdrop = o->zdrop_inv = 100; o->end_bonus = 10; o->bw_long = 100;
        o->pri_ratio = 0.5f; o->min_dp_max = 40; o->best_n = 20;
        return 0;
    }
    if (strcmp(name, "map-ont") == 0) {         /* Preset::MapOnt, cleaner.rs:457 */
        o->k = 15; o->w = 10;
        return 0;
    }
    if (strcmp(name, "lr:hq") == 0 || strcmp(name, "map-hifi") == 0) {      /* Preset::LrHq, Preset::MapHifi: cleaner.rs:458,465 */
        o->k = 19; o->w = 19; o->max_gap = 10000;
        o->min_mid_occ = 50; o->max_mid_occ = 500;
        if (strcmp(name, "map-hifi") == 0) {    /* its own scores and dp_max floor; the sketch and the chain set-up are lr:hq's */
            o->a = 1; o->b = 4; o->q = 6; o->q2 = 26; o->e = 2; o->e2 = 1; o->min_dp_max = 200;
        }
        return 0;
    }
    if (strcmp(name, "lr") == 0) return -2;     /* Preset::Lr rejected, cleaner.rs:469 */
    return -1;
}

/* ------------------------------------------------------------------------------------------
 * A.2 minimizer sketch
 * ---------------------------------------------------------------------------------------- */
static const uint8_t nt4_init[4][2] = { {'A', 'a'}, {'C', 'c'}, {'G', 'g'}, {'T', 't'} };
static uint8_t nt4_table[256];
static pthread_once_t nt4_once = PTHREAD_ONCE_INIT;
static void nt4_fill(void)
{
    int i;
    memset(nt4_table, 4, 256);
    for (i = 0; i < 4; ++i) nt4_table[nt4_init[i][0]] = nt4_table[nt4_init[i][1]] = (uint8_t)i;
    nt4_table['U'] = nt4_table['u'] = 3;
}

uint64_t mmo_hash64(uint64_t key, uint64_t mask)
{   /* invertible integer mix restricted to 2k bits, App. A.2 */
    key = (~key + (key << 21)) & mask;
    key = key ^ key >> 24;
    key = ((key + (key << 3)) + (key << 8)) & mask;
    key = key ^ key >> 14;
    key = ((key + (key << 2)) + (key << 4)) & mask;
    key = key ^ key >> 28;
    key = (key + (key << 31)) & mask;
    return key;
}

typedef struct { uint64_t x, y; } m128;

#define PUSH(X, Y) do { if (n < cap) { xo[n] = (X); yo[n] = (Y); } ++n; } while (0)

int64_t mmo_sketch(const uint8_t *seq, int64_t len, int w, int k, uint32_t rid,
                   uint64_t *xo, uint64_t *yo, int64_t cap)
{
    uint64_t shift1 = 2 * (k - 1), mask = (1ULL << 2 * k) - 1, kmer[2] = { 0, 0 };
    int64_t i, n = 0;
    int j, l, buf_pos, min_pos, kmer_span = 0;
    m128 buf[256], min = { UINT64_MAX, UINT64_MAX };

    pthread_once(&nt4_once, nt4_fill);
    if (len <= 0 || w <= 0 || w >= 256 || k <= 0 || k > 28) return 0;
    memset(buf, 0xff, (size_t)w * 16);

    for (i = 0, l = buf_pos = min_pos = 0; i < len; ++i) {
        int c = nt4_table[seq[i]];
        m128 info = { UINT64_MAX, UINT64_MAX };
        if (c < 4) {
            int z;
            kmer_span = l + 1 < k ? l + 1 : k;
            kmer[0] = (kmer[0] << 2 | (uint64_t)c) & mask;
            kmer[1] = (kmer[1] >> 2) | (3ULL ^ (uint64_t)c) << shift1;
            if (kmer[0] == kmer[1]) continue;   /* strand-ambiguous k-mer: skipped entirely */
            z = kmer[0] < kmer[1] ? 0 : 1;
            ++l;
            if (l >= k && kmer_span < 256) {
                info.x = mmo_hash64(kmer[z], mask) << 8 | (uint64_t)kmer_span;
                info.y = (uint64_t)rid << 32 | (uint32_t)i << 1 | (uint64_t)z;
            }
        } else l = 0, kmer_span = 0;
        buf[buf_pos] = info;
        if (l == w + k - 1 && min.x != UINT64_MAX) {   /* first full window: identical k-mers */
            for (j = buf_pos + 1; j < w; ++j)
                if (min.x == buf[j].x && buf[j].y != min.y) PUSH(buf[j].x, buf[j].y);
            for (j = 0; j < buf_pos; ++j)
                if (min.x == buf[j].x && buf[j].y != min.y) PUSH(buf[j].x, buf[j].y);
        }
        if (info.x <= min.x) {                          /* new minimum (rightmost on ties) */
            if (l >= w + k && min.x != UINT64_MAX) PUSH(min.x, min.y);
            min = info, min_pos = buf_pos;
        } else if (buf_pos == min_pos) {                /* old minimum left the window */
            if (l >= w + k - 1 && min.x != UINT64_MAX) PUSH(min.x, min.y);
            for (j = buf_pos + 1, min.x = UINT64_MAX; j < w; ++j)
                if (min.x >= buf[j].x) min = buf[j], min_pos = j;
            for (j = 0; j <= buf_pos; ++j)
                if (min.x >= buf[j].x) min = buf[j], min_pos = j;
            if (l >= w + k - 1 && min.x != UINT64_MAX) {
                for (j = buf_pos + 1; j < w; ++j)
                    if (min.x == buf[j].x && min.y != buf[j].y) PUSH(buf[j].x, buf[j].y);
                for (j = 0; j <= buf_pos; ++j)
                    if (min.x == buf[j].x && min.y != buf[j].y) PUSH(buf[j].x, buf[j].y);
            }
        }
        if (++buf_pos == w) buf_pos = 0;
    }
    if (min.x != UINT64_MAX) PUSH(min.x, min.y);
    return n;
}
#undef PUSH

/* ------------------------------------------------------------------------------------------
 * A.3 index.  Logical content = minimap2's: minimizer hash -> ascending list of
 * (rid<<32 | pos<<1 | strand).  Physical layout = the product's HBM layout:
 * open-addressing table of 16-B slots + one position array, so the same oracle code can
 * run on an index copied back from the GPU (bench.py cpu_baseline).
 *   slot.w0 = ~0 (empty) | key | multi<<63
 *   slot.w1 = position word (singleton) | off<<28 | n (multi)
 * ---------------------------------------------------------------------------------------- */
#define SLOT_EMPTY UINT64_MAX
#define SLOT_MULTI (1ULL << 63)
#define SLOT_KEYMASK ((1ULL << 56) - 1)
#define SLOT_NBITS 28

struct mmo_index {
    int w, k;
    uint64_t n_slots, lg_slots, n_keys, n_positions;
    uint64_t *slots;       /* 2 * n_slots */
    uint64_t *positions;
    int owned;
    /* reference sequence for A.6 (mm_idx_getseq): 4-bit packed nt4 codes + contig starts */
    uint8_t *ref; uint64_t *cstart; uint32_t n_contigs; int ref_owned;
};

static inline uint64_t slot_home(uint64_t key, uint64_t lg)
{
    return (key * 0x9E3779B97F4A7C15ULL) >> (64 - lg);
}

static int cmp_m128(const void *a, const void *b)
{
    const m128 *p = (const m128 *)a, *q = (const m128 *)b;
    if (p->x != q->x) return p->x < q->x ? -1 : 1;
    if (p->y != q->y) return p->y < q->y ? -1 : 1;
    return 0;
}

mmo_index *mmo_index_build(int n_seq, const uint8_t *const *seqs, const int64_t *lens, int w, int k)
{
    mmo_index *idx;
    m128 *a;
    uint64_t *xs, *ys;
    int64_t cap = 0, n = 0, i, j, max_len = 0;
    uint64_t n_keys = 0, n_pos = 0, lg, off;
    int s;

    for (s = 0; s < n_seq; ++s) { cap += lens[s] > 0 ? lens[s] : 0; if (lens[s] > max_len) max_len = lens[s]; }
    a = (m128 *)malloc(sizeof(m128) * (size_t)(cap + 1));
    xs = (uint64_t *)malloc(8 * (size_t)(max_len + 1));
    ys = (uint64_t *)malloc(8 * (size_t)(max_len + 1));
    for (s = 0; s < n_seq; ++s) {
        int64_t m;
        if (lens[s] <= 0) continue;
        m = mmo_sketch(seqs[s], lens[s], w, k, (uint32_t)s, xs, ys, max_len + 1);
        for (i = 0; i < m; ++i) a[n].x = xs[i] >> 8, a[n].y = ys[i], ++n;
    }
    free(xs); free(ys);
    qsort(a, (size_t)n, sizeof(m128), cmp_m128);
    for (i = 0; i < n; i = j) {
        for (j = i + 1; j < n && a[j].x == a[i].x; ++j) {}
        ++n_keys;
        if (j - i > 1) n_pos += (uint64_t)(j - i);
    }
    idx = (mmo_index *)calloc(1, sizeof(*idx));
    idx->w = w; idx->k = k; idx->owned = 1;
    {   /* mi->S: the reference as 4-bit nt4 codes */
        uint64_t g = 0, tot = 0;
        for (s = 0; s < n_seq; ++s) tot += lens[s] > 0 ? (uint64_t)lens[s] : 0;
        idx->ref = (uint8_t *)calloc((size_t)(tot / 2 + 2), 1);
        idx->cstart = (uint64_t *)calloc((size_t)n_seq + 1, 8);
        idx->n_contigs = (uint32_t)n_seq; idx->ref_owned = 1;
        for (s = 0; s < n_seq; ++s) {
            idx->cstart[s] = g;
            for (i = 0; i < lens[s]; ++i, ++g) idx->ref[g >> 1] |= (uint8_t)(nt4_table[seqs[s][i]] << ((g & 1) * 4));
        }
        idx->cstart[n_seq] = g;
    }
    for (lg = 4; (1ULL << lg) < 2 * n_keys + 1; ++lg) {}
    idx->lg_slots = lg; idx->n_slots = 1ULL << lg; idx->n_keys = n_keys; idx->n_positions = n_pos;
    idx->slots = (uint64_t *)malloc(16 * (size_t)idx->n_slots);
    memset(idx->slots, 0xff, 16 * (size_t)idx->n_slots);
    idx->positions = (uint64_t *)malloc(8 * (size_t)(n_pos + 1));
    for (i = 0, off = 0; i < n; i = j) {
        uint64_t h, cnt;
        for (j = i + 1; j < n && a[j].x == a[i].x; ++j) {}
        cnt = (uint64_t)(j - i);
        h = slot_home(a[i].x, lg);
        while (idx->slots[2 * h] != SLOT_EMPTY) h = (h + 1) & (idx->n_slots - 1);
        if (cnt == 1) {
            idx->slots[2 * h] = a[i].x;
            idx->slots[2 * h + 1] = a[i].y;
        } else {
            int64_t t;
            idx->slots[2 * h] = a[i].x | SLOT_MULTI;
            idx->slots[2 * h + 1] = off << SLOT_NBITS | cnt;
            for (t = i; t < j; ++t) idx->positions[off++] = a[t].y;   /* already ascending */
        }
    }
    free(a);
    return idx;
}

mmo_index *mmo_index_wrap(const uint64_t *slots, uint64_t n_slots, const uint64_t *positions,
                          uint64_t n_positions, int w, int k)
{
    mmo_index *idx = (mmo_index *)calloc(1, sizeof(*idx));
    uint64_t lg, i;
    for (lg = 0; (1ULL << lg) < n_slots; ++lg) {}
    idx->w = w; idx->k = k; idx->owned = 0;
    idx->n_slots = n_slots; idx->lg_slots = lg; idx->n_positions = n_positions;
    idx->slots = (uint64_t *)slots; idx->positions = (uint64_t *)positions;
    for (i = 0; i < n_slots; ++i) if (slots[2 * i] != SLOT_EMPTY) ++idx->n_keys;
    return idx;
}

void mmo_index_free(mmo_index *idx)
{
    if (!idx) return;
    if (idx->owned) { free(idx->slots); free(idx->positions); }
    if (idx->ref_owned) { free(idx->ref); free(idx->cstart); }
    free(idx);
}

void mmo_index_set_ref(mmo_index *idx, const uint8_t *packed, const uint64_t *contig_start, uint32_t n_contigs)
{
    if (idx->ref_owned) { free(idx->ref); free(idx->cstart); }
    idx->ref = (uint8_t *)packed; idx->cstart = (uint64_t *)contig_start; idx->n_contigs = n_contigs; idx->ref_owned = 0;
}

const uint8_t *mmo_index_ref(const mmo_index *idx, const uint64_t **contig_start, uint32_t *n_contigs)
{
    if (contig_start) *contig_start = idx->cstart;
    if (n_contigs) *n_contigs = idx->n_contigs;
    return idx->ref;
}

uint64_t mmo_index_n_keys(const mmo_index *idx) { return idx->n_keys; }
uint64_t mmo_index_n_positions(const mmo_index *idx) { return idx->n_positions; }

/* mm_idx_get: occurrences of one minimizer hash */
const uint64_t *mmo_index_get(const mmo_index *idx, uint64_t minier, int32_t *n)
{
    uint64_t h = slot_home(minier, idx->lg_slots), m = idx->n_slots - 1;
    *n = 0;
    for (;;) {
        uint64_t w0 = idx->slots[2 * h];
        if (w0 == SLOT_EMPTY) return 0;
        if ((w0 & SLOT_KEYMASK) == minier) {
            if (w0 & SLOT_MULTI) {
                uint64_t w1 = idx->slots[2 * h + 1];
                *n = (int32_t)(w1 & ((1ULL << SLOT_NBITS) - 1));
                return idx->positions + (w1 >> SLOT_NBITS);
            }
            *n = 1;
            return &idx->slots[2 * h + 1];
        }
        h = (h + 1) & m;
    }
}

typedef struct { uint64_t key, slot; } keyslot;
static int cmp_keyslot(const void *a, const void *b)
{
    const keyslot *p = (const keyslot *)a, *q = (const keyslot *)b;
    return p->key < q->key ? -1 : p->key > q->key;
}

void mmo_index_dump(const mmo_index *idx, uint64_t *keys, uint32_t *counts, uint64_t *positions)
{
    keyslot *ks = (keyslot *)malloc(sizeof(keyslot) * (size_t)(idx->n_keys + 1));
    uint64_t i, n = 0, o = 0;
    for (i = 0; i < idx->n_slots; ++i)
        if (idx->slots[2 * i] != SLOT_EMPTY) ks[n].key = idx->slots[2 * i] & SLOT_KEYMASK, ks[n].slot = i, ++n;
    qsort(ks, (size_t)n, sizeof(keyslot), cmp_keyslot);
    for (i = 0; i < n; ++i) {
        int32_t c, t;
        const uint64_t *p = mmo_index_get(idx, ks[i].key, &c);
        keys[i] = ks[i].key; counts[i] = (uint32_t)c;
        for (t = 0; t < c; ++t) positions[o++] = p[t];
    }
    free(ks);
}

static int cmp_u32(const void *a, const void *b)
{
    uint32_t p = *(const uint32_t *)a, q = *(const uint32_t *)b;
    return p < q ? -1 : p > q;
}

/* mm_idx_cal_max_occ: (k-th smallest occurrence count, k = (1-f)*n) + 1, App. A.4 */
int32_t mmo_index_cal_mid_occ(const mmo_index *idx, float f)
{
    uint32_t *a, thres;
    uint64_t i, n = 0;
    if (f <= 0.f || idx->n_keys == 0) return INT32_MAX;
    a = (uint32_t *)malloc(4 * (size_t)idx->n_keys);
    for (i = 0; i < idx->n_slots; ++i) {
        uint64_t w0 = idx->slots[2 * i];
        if (w0 == SLOT_EMPTY) continue;
        a[n++] = (w0 & SLOT_MULTI) ? (uint32_t)(idx->slots[2 * i + 1] & ((1ULL << SLOT_NBITS) - 1)) : 1;
    }
    qsort(a, (size_t)n, 4, cmp_u32);
    thres = a[(uint32_t)((1. - f) * n)] + 1;
    free(a);
    return (int32_t)thres;
}

void mmo_opts_update(mmo_opts *o, const mmo_index *idx)
{   /* mm_mapopt_update */
    if (o->mid_occ <= 0) {
        o->mid_occ = mmo_index_cal_mid_occ(idx, o->mid_occ_frac);
        if (o->mid_occ < o->min_mid_occ) o->mid_occ = o->min_mid_occ;
        if (o->max_mid_occ > o->min_mid_occ && o->mid_occ > o->max_mid_occ) o->mid_occ = o->max_mid_occ;
    }
}

/* ------------------------------------------------------------------------------------------
 * A.4 seed collection
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    uint32_t n, q_pos, q_span;   /* q_pos = pos<<1 | strand */
    uint32_t flt, is_tandem;
    const uint64_t *cr;
} seed_t;

#define SEED_LONG_JOIN (1ULL << 40)
#define SEED_IGNORE    (1ULL << 41)
#define SEED_TANDEM    (1ULL << 42)

typedef struct {          /* per-thread scratch, grown on demand */
    uint64_t *mx, *my; int64_t cap_m;
    seed_t *seeds; int64_t cap_s;
    m128 *a, *a2; int64_t cap_a;
    int32_t *f, *t; int64_t *p; m128 *z; int64_t cap_dp;
    m128 *srt; int64_t cap_srt;
    uint64_t *u; int32_t *v; m128 *b; int64_t cap_u, cap_v;      /* chains: u[i] = score<<32 | cnt, v = anchor indices; b = compact_a's output */
    int32_t n_u; int64_t n_v;
    uint64_t *mini_pos; int64_t cap_mp; int32_t n_mini_pos;      /* mm_collect_matches: q_span<<32 | q_pos of the seeds that were not filtered */
} scratch_t;

static void *grow(void *p, int64_t *cap, int64_t need, size_t sz)
{
    if (need <= *cap) return p;
    *cap = need + (need >> 1) + 16;
    return realloc(p, (size_t)*cap * sz);
}

static void scratch_free(scratch_t *s)
{
    free(s->mx); free(s->my); free(s->seeds); free(s->a); free(s->a2);
    free(s->f); free(s->t); free(s->p); free(s->z); free(s->srt);
    free(s->u); free(s->v); free(s->b); free(s->mini_pos);
}

/* stable merge sort on x (ties keep input order) — radix_sort_128x is stable for the small
 * inputs that dominate (insertion sort below 64 elements); larger inputs: documented choice */
static void sort128x(m128 *a, m128 *tmp, int64_t n)
{
    int64_t width, i;
    m128 *src = a, *dst = tmp;
    if (n < 2) return;
    for (i = 1; i < n; ++i) {          /* already sorted? common for single-strand unique hits */
        if (a[i].x < a[i - 1].x) break;
    }
    if (i == n) return;
    for (width = 1; width < n; width <<= 1) {
        for (i = 0; i < n; i += 2 * width) {
            int64_t l = i, m = i + width < n ? i + width : n, r = i + 2 * width < n ? i + 2 * width : n;
            int64_t p = l, q = m, o = l;
            while (p < m && q < r) dst[o++] = src[q].x < src[p].x ? src[q++] : src[p++];
            while (p < m) dst[o++] = src[p++];
            while (q < r) dst[o++] = src[q++];
        }
        { m128 *t = src; src = dst; dst = t; }
    }
    if (src != a) memcpy(a, src, sizeof(m128) * (size_t)n);
}

/* mm_seed_mz_flt: thin minimizers that repeat within the query, App. A.4 first bullet */
static int64_t seed_mz_flt(scratch_t *s, int64_t n, int32_t q_occ_max, float q_occ_frac)
{
    int64_t i, j, st;
    if (n <= q_occ_max || q_occ_frac <= 0.0f || q_occ_max <= 0) return n;
    s->srt = (m128 *)grow(s->srt, &s->cap_srt, 2 * n, sizeof(m128));
    for (i = 0; i < n; ++i) s->srt[i].x = s->mx[i], s->srt[i].y = (uint64_t)i;
    sort128x(s->srt, s->srt + n, n);
    for (st = 0, i = 1; i <= n; ++i) {
        if (i == n || s->srt[i].x != s->srt[st].x) {
            int32_t cnt = (int32_t)(i - st);
            if (cnt > q_occ_max && cnt > n * q_occ_frac)
                for (j = st; j < i; ++j) s->mx[s->srt[j].y] = 0;
            st = i;
        }
    }
    for (i = j = 0; i < n; ++i)
        if (s->mx[i] != 0) s->mx[j] = s->mx[i], s->my[j] = s->my[i], ++j;
    return j;
}

#define MAX_MAX_HIGH_OCC 128

/* mm_seed_select: in each streak of high-occurrence seeds keep the (pe-ps)/dist lowest ones */
static void seed_select(int32_t n, seed_t *a, int len, int max_occ, int max_max_occ, int dist)
{
    int32_t i, last0, m;
    uint64_t b[MAX_MAX_HIGH_OCC];
    if (n == 0 || n == 1) return;
    for (i = m = 0; i < n; ++i) if (a[i].n > (uint32_t)max_occ) ++m;
    if (m == 0) return;
    for (i = 0, last0 = -1; i <= n; ++i) {
        if (i == n || a[i].n <= (uint32_t)max_occ) {
            if (i - last0 > 1) {
                int32_t ps = last0 < 0 ? 0 : (int32_t)(a[last0].q_pos >> 1);
                int32_t pe = i == n ? len : (int32_t)(a[i].q_pos >> 1);
                int32_t j, k, st = last0 + 1, en = i;
                int32_t max_high_occ = (int32_t)((double)(pe - ps) / dist + .499);
                if (max_high_occ > 0) {
                    if (max_high_occ > MAX_MAX_HIGH_OCC) max_high_occ = MAX_MAX_HIGH_OCC;
                    for (j = st, k = 0; j < en && k < max_high_occ; ++j, ++k)
                        b[k] = (uint64_t)a[j].n << 32 | (uint32_t)j;
                    for (; j < en; ++j) {      /* keep the k smallest; b-max is replaced on strict < of n */
                        int32_t q, mx = 0;
                        for (q = 1; q < k; ++q) if (b[q] > b[mx]) mx = q;
                        if (a[j].n < (uint32_t)(b[mx] >> 32)) b[mx] = (uint64_t)a[j].n << 32 | (uint32_t)j;
                    }
                    for (j = 0; j < k; ++j) a[(uint32_t)b[j]].flt = 1;
                }
                for (j = st; j < en; ++j) a[j].flt ^= 1;
                for (j = st; j < en; ++j) if (a[j].n > (uint32_t)max_max_occ) a[j].flt = 1;
            }
            last0 = i;
        }
    }
}

/* mm_seed_collect_all + mm_collect_matches + collect_seed_hits (anchors), single segment.
 * Returns the number of anchors, sorted by x (strand|rid|rpos). */
static int64_t collect_anchors(const mmo_index *idx, const mmo_opts *o, scratch_t *s, int64_t n_mv,
                               int qlen, int max_occ, int32_t *n_seed_out, int32_t *rep_len_out)
{
    int64_t i, n_a = 0, k_a;
    int32_t n_m0 = 0, rep_st = 0, rep_en = 0, rep_len = 0;
    seed_t *m;
    s->seeds = (seed_t *)grow(s->seeds, &s->cap_s, n_mv, sizeof(seed_t));
    m = s->seeds;
    for (i = 0; i < n_mv; ++i) {
        int32_t t;
        const uint64_t *cr = mmo_index_get(idx, s->mx[i] >> 8, &t);
        if (t == 0) continue;
        m[n_m0].q_pos = (uint32_t)s->my[i]; m[n_m0].q_span = (uint32_t)(s->mx[i] & 0xff);
        m[n_m0].cr = cr; m[n_m0].n = (uint32_t)t; m[n_m0].flt = 0;
        /* mm_seed_collect_all: a seed is "tandem" when a neighbour in the minimizer list has the same hash */
        m[n_m0].is_tandem = (i > 0 && s->mx[i] >> 8 == s->mx[i - 1] >> 8) || (i < n_mv - 1 && s->mx[i] >> 8 == s->mx[i + 1] >> 8);
        ++n_m0;
    }
    *n_seed_out = n_m0;
    if (o->occ_dist > 0 && o->max_max_occ > max_occ) seed_select(n_m0, m, qlen, max_occ, o->max_max_occ, o->occ_dist);
    else for (i = 0; i < n_m0; ++i) if (m[i].n > (uint32_t)max_occ) m[i].flt = 1;
    s->mini_pos = (uint64_t *)grow(s->mini_pos, &s->cap_mp, n_m0 + 1, 8);
    s->n_mini_pos = 0;
    for (i = 0; i < n_m0; ++i) {
        seed_t *q = &m[i];
        if (q->flt) {
            int en = (int)(q->q_pos >> 1) + 1, st = en - (int)q->q_span;
            if (st > rep_en) { rep_len += rep_en - rep_st; rep_st = st; rep_en = en; }
            else rep_en = en;
        } else {
            n_a += q->n;
            s->mini_pos[s->n_mini_pos++] = (uint64_t)q->q_span << 32 | q->q_pos >> 1;
        }
    }
    rep_len += rep_en - rep_st;
    *rep_len_out = rep_len;
    if (n_a > s->cap_a) {
        s->cap_a = n_a + (n_a >> 1) + 16;
        s->a = (m128 *)realloc(s->a, sizeof(m128) * (size_t)s->cap_a);
        s->a2 = (m128 *)realloc(s->a2, sizeof(m128) * (size_t)s->cap_a);
    }
    for (i = 0, k_a = 0; i < n_m0; ++i) {
        seed_t *q = &m[i];
        uint32_t k;
        if (q->flt) continue;
        for (k = 0; k < q->n; ++k) {
            uint64_t r = q->cr[k];
            int32_t rpos = (int32_t)((uint32_t)r >> 1);
            m128 *p = &s->a[k_a++];
            if ((r & 1) == (q->q_pos & 1)) {
                p->x = (r & 0xffffffff00000000ULL) | (uint32_t)rpos;
                p->y = (uint64_t)q->q_span << 32 | q->q_pos >> 1;
            } else {
                p->x = 1ULL << 63 | (r & 0xffffffff00000000ULL) | (uint32_t)rpos;
                p->y = (uint64_t)q->q_span << 32 | (uint32_t)(qlen - ((int32_t)(q->q_pos >> 1) + 1 - (int32_t)q->q_span) - 1);
            }
            if (q->is_tandem) p->y |= SEED_TANDEM;
        }
    }
    sort128x(s->a, s->a2, n_a);
    return n_a;
}

/* ------------------------------------------------------------------------------------------
 * A.5 chaining
 * ---------------------------------------------------------------------------------------- */
float mmo_log2(float x)   /* fast approximate log2, valid for x >= 2 */
{
    union { float f; uint32_t i; } z = { x };
    float log_2 = (float)(((z.i >> 23) & 255) - 128);
    z.i &= ~(255U << 23);
    z.i += 127U << 23;
    log_2 += (-0.34484843f * z.f + 2.02466578f) * z.f - 0.67487759f;
    return log_2;
}

int32_t mmo_comput_sc(uint64_t ai_x, uint64_t ai_y, uint64_t aj_x, uint64_t aj_y, int32_t max_dist_x,
                      int32_t max_dist_y, int32_t bw, float chn_pen_gap, float chn_pen_skip)
{
    int32_t dq = (int32_t)ai_y - (int32_t)aj_y, dr, dd, dg, q_span, sc;
    if (dq <= 0 || dq > max_dist_x) return INT32_MIN;
    dr = (int32_t)(ai_x - aj_x);
    if (dr == 0 || dq > max_dist_y) return INT32_MIN;
    dd = dr > dq ? dr - dq : dq - dr;
    if (dd > bw) return INT32_MIN;
    dg = dr < dq ? dr : dq;
    q_span = (int32_t)(aj_y >> 32 & 0xff);
    sc = q_span < dg ? q_span : dg;
    if (dd || dg > q_span) {
        float lin_pen, log_pen;
        lin_pen = chn_pen_gap * (float)dd + chn_pen_skip * (float)dg;
        log_pen = dd >= 1 ? mmo_log2((float)(dd + 1)) : 0.0f;
        sc -= (int32_t)(lin_pen + .5f * log_pen);
    }
    return sc;
}

static int64_t chain_bk_end(int32_t max_drop, const m128 *z, const int32_t *f, const int64_t *p, int32_t *t, int64_t k)
{
    int64_t i = (int64_t)z[k].y, end_i = -1, max_i = i;
    int32_t max_s = 0;
    if (i < 0 || t[i] != 0) return i;
    do {
        int32_t sc;
        t[i] = 2;
        end_i = i = p[i];
        sc = i < 0 ? (int32_t)z[k].x : (int32_t)z[k].x - f[i];
        if (sc > max_s) max_s = sc, max_i = i;
        else if (max_s - sc > max_drop) break;
    } while (i >= 0 && t[i] == 0);
    for (i = (int64_t)z[k].y; i >= 0 && i != end_i; i = p[i]) t[i] = 0;
    return max_i;
}

/* mg_chain_backtrack over s->f / s->p (n anchors): chains into s->u / s->v; returns their number */
static int32_t chain_backtrack(const mmo_opts *o, scratch_t *s, int64_t n, int32_t max_drop, int32_t *best_score)
{
    int32_t *f = s->f, *t = s->t, n_u = 0, best = 0;
    int64_t *p = s->p, i, n_z, k, n_v = 0;
    m128 *z = s->z;
    *best_score = 0;
    s->n_u = 0; s->n_v = 0;
    /* candidates with f >= min_sc in ascending (f, index) order, visited from the top */
    for (i = 0, n_z = 0; i < n; ++i)
        if (f[i] >= o->min_chain_score) z[n_z].x = (uint64_t)(int64_t)f[i], z[n_z].y = (uint64_t)i, ++n_z;
    if (n_z == 0) return 0;
    sort128x(z, z + n_z, n_z);
    memset(t, 0, 4 * (size_t)n);
    if (n > s->cap_v) { s->cap_v = n + (n >> 1) + 16; s->v = (int32_t *)realloc(s->v, 4 * (size_t)s->cap_v); }
    for (k = n_z - 1; k >= 0; --k) {
        if (t[z[k].y] == 0) {
            int64_t n_v0 = n_v, end_i;
            int32_t sc;
            end_i = chain_bk_end(max_drop, z, f, p, t, k);
            for (i = (int64_t)z[k].y; i != end_i; i = p[i]) s->v[n_v++] = (int32_t)i, t[i] = 1;
            sc = i < 0 ? (int32_t)z[k].x : (int32_t)z[k].x - f[i];
            if (sc >= o->min_chain_score && n_v > n_v0 && n_v - n_v0 >= o->min_cnt) {
                if (n_u >= s->cap_u) { s->cap_u = s->cap_u * 2 + 16; s->u = (uint64_t *)realloc(s->u, 8 * (size_t)s->cap_u); }
                s->u[n_u++] = (uint64_t)sc << 32 | (uint64_t)(n_v - n_v0);
                if (sc > best) best = sc;
            } else n_v = n_v0;
        }
    }
    s->n_u = n_u; s->n_v = n_v;
    *best_score = best;
    return n_u;
}

/* mg_lchain_dp + mg_chain_backtrack: number of chains kept and best kept score */
static int32_t chain_dp(const mmo_opts *o, scratch_t *s, int64_t n, int k_idx, int qlen, int32_t *best_score)
{
    int32_t max_dist_x, max_dist_y, bw = o->bw, max_drop = o->bw;
    int32_t *f, *t;
    int64_t *p, i, j, max_ii, st = 0;
    float chn_pen_gap, chn_pen_skip;
    m128 *a = s->a;

    *best_score = 0;
    if (n == 0) return 0;
    /* gap limits, mm_map_frag */
    max_dist_y = o->is_sr ? (qlen > o->max_gap ? qlen : o->max_gap) : o->max_gap;
    if (o->max_gap_ref > 0) max_dist_x = o->max_gap_ref;
    else if (o->max_frag_len > 0) {
        max_dist_x = o->max_frag_len - qlen;
        if (max_dist_x < o->max_gap) max_dist_x = o->max_gap;
    } else max_dist_x = o->max_gap;
    chn_pen_gap = (float)(o->chain_gap_scale * 0.01 * k_idx);
    chn_pen_skip = (float)(o->chain_skip_scale * 0.01 * k_idx);
    if (max_dist_x < bw) max_dist_x = bw;
    if (max_dist_y < bw) max_dist_y = bw;

    if (n > s->cap_dp) {
        s->cap_dp = n + (n >> 1) + 16;
        s->f = (int32_t *)realloc(s->f, 4 * (size_t)s->cap_dp);
        s->t = (int32_t *)realloc(s->t, 4 * (size_t)s->cap_dp);
        s->p = (int64_t *)realloc(s->p, 8 * (size_t)s->cap_dp);
        s->z = (m128 *)realloc(s->z, 16 * (size_t)s->cap_dp * 2);
    }
    f = s->f; t = s->t; p = s->p;
    memset(t, 0, 4 * (size_t)n);

    for (i = 0, max_ii = -1; i < n; ++i) {
        int64_t max_j = -1, end_j;
        int32_t max_f = (int32_t)(a[i].y >> 32 & 0xff), n_skip = 0;
        while (st < i && (a[i].x >> 32 != a[st].x >> 32 || a[i].x > a[st].x + (uint64_t)max_dist_x)) ++st;
        if (i - st > o->max_chain_iter) st = i - o->max_chain_iter;
        for (j = i - 1; j >= st; --j) {
            int32_t sc = mmo_comput_sc(a[i].x, a[i].y, a[j].x, a[j].y, max_dist_x, max_dist_y, bw, chn_pen_gap, chn_pen_skip);
            if (sc == INT32_MIN) continue;
            sc += f[j];
            if (sc > max_f) {
                max_f = sc, max_j = j;
                if (n_skip > 0) --n_skip;
            } else if (t[j] == (int32_t)i) {
                if (++n_skip > o->max_chain_skip) break;
            }
            if (p[j] >= 0) t[p[j]] = (int32_t)i;
        }
        end_j = j;
        if (max_ii < 0 || a[i].x - a[max_ii].x > (uint64_t)(int64_t)max_dist_x) {
            int32_t max = INT32_MIN;
            max_ii = -1;
            for (j = i - 1; j >= st; --j)
                if (max < f[j]) max = f[j], max_ii = j;
        }
        if (max_ii >= 0 && max_ii < end_j) {
            int32_t tmp = mmo_comput_sc(a[i].x, a[i].y, a[max_ii].x, a[max_ii].y, max_dist_x, max_dist_y, bw, chn_pen_gap, chn_pen_skip);
            if (tmp != INT32_MIN && max_f < tmp + f[max_ii])
                max_f = tmp + f[max_ii], max_j = max_ii;
        }
        f[i] = max_f, p[i] = max_j;
        if (max_ii < 0 || (a[i].x - a[max_ii].x <= (uint64_t)(int64_t)max_dist_x && f[max_ii] < f[i]))
            max_ii = i;
    }

    return chain_backtrack(o, s, n, max_drop, best_score);
}

/* compact_a of mg_lchain_dp: each chain's anchors in ascending order, chains re-ordered by the target position of their
 * first anchor (stable for equal keys); u[] follows.  Result in s->b. */
static void compact_chains(scratch_t *s)
{
    const int32_t n_u = s->n_u;
    int64_t i, j, k;
    m128 *b, *w, *a = s->a;
    uint64_t *u2;
    b = (m128 *)malloc(sizeof(m128) * (size_t)(s->n_v + 1));
    for (i = 0, k = 0; i < n_u; ++i) {
        const int64_t k0 = k; const int32_t ni = (int32_t)s->u[i];
        for (j = 0; j < ni; ++j) b[k++] = a[s->v[k0 + (ni - j - 1)]];
    }
    w = (m128 *)malloc(sizeof(m128) * (size_t)(n_u + 1) * 2);
    for (i = k = 0; i < n_u; ++i) { w[i].x = b[k].x; w[i].y = (uint64_t)k << 32 | (uint64_t)i; k += (int32_t)s->u[i]; }
    sort128x(w, w + n_u, n_u);
    u2 = (uint64_t *)malloc(8 * (size_t)(n_u + 1));
    free(s->b);
    s->b = (m128 *)malloc(sizeof(m128) * (size_t)(s->n_v + 1));
    for (i = k = 0; i < n_u; ++i) {
        const int32_t jj = (int32_t)w[i].y, n = (int32_t)s->u[jj];
        u2[i] = s->u[jj];
        memcpy(&s->b[k], &b[w[i].y >> 32], (size_t)n * sizeof(m128));
        k += n;
    }
    memcpy(s->u, u2, 8 * (size_t)n_u);
    free(b); free(w); free(u2);
}

static void map_one(const mmo_index *idx, const mmo_opts *o, scratch_t *s, const uint8_t *seq, int64_t len, mmo_trace *tr)
{
    int64_t n_mv, n_a;
    int32_t n_seed = 0, rep_len = 0, n_u, best = 0;
    memset(tr, 0, sizeof(*tr));
    if (len <= 0) { tr->flag = 2; return; }      /* minimap2-rs: Err("Sequence is empty") */
    if (2 * len + 256 > s->cap_m) {      /* every k-mer is emitted at most once as minimum and once as tie */
        s->cap_m = 2 * len + 256;
        s->mx = (uint64_t *)realloc(s->mx, 8 * (size_t)s->cap_m);
        s->my = (uint64_t *)realloc(s->my, 8 * (size_t)s->cap_m);
    }
    n_mv = mmo_sketch(seq, len, o->w, o->k, 0, s->mx, s->my, s->cap_m);
    if (o->q_occ_frac > 0.0f) n_mv = seed_mz_flt(s, n_mv, o->mid_occ, o->q_occ_frac);
    tr->n_mini = (int32_t)n_mv;
    n_a = collect_anchors(idx, o, s, n_mv, (int)len, o->mid_occ, &n_seed, &rep_len);
    n_u = chain_dp(o, s, n_a, o->k, (int)len, &best);
    if (n_u == 0 && o->max_occ > o->mid_occ && rep_len > 0) {     /* re-chain, mostly for short reads */
        tr->rechained = 1;
        n_a = collect_anchors(idx, o, s, n_mv, (int)len, o->max_occ, &n_seed, &rep_len);
        n_u = chain_dp(o, s, n_a, o->k, (int)len, &best);
    }
    tr->n_seed = n_seed; tr->n_anchor = (int32_t)n_a; tr->rep_len = rep_len;
    tr->n_chain = n_u; tr->best_score = best; tr->flag = n_u > 0;
    /* A.6: with MM_F_CIGAR (`.with_cigar()`, cleaner.rs:473) a chain only counts once a region of it survives the base-level
     * alignment and mm_filter_regs. */
    if (n_u > 0 && (o->flags & MMO_F_CIGAR) && idx->ref) {
        mma_result res;
        compact_chains(s);
        if (!o->is_sr && o->bw_long > o->bw && n_u > 1) {      /* mm_map_frag: re-chain / long-join for long sequences */
            const int32_t st = (int32_t)s->b[0].y, en = (int32_t)s->b[(int32_t)s->u[0] - 1].y;
            if ((int32_t)len - (en - st) > o->rmq_rescue_size || en - st > (int32_t)len * o->rmq_rescue_ratio) {
                int64_t i;
                const float chn_pen_gap = (float)(o->chain_gap_scale * 0.01 * o->k), chn_pen_skip = (float)(o->chain_skip_scale * 0.01 * o->k);
                for (i = 0, n_a = 0; i < n_u; ++i) n_a += (int32_t)s->u[i];
                memcpy(s->a, s->b, sizeof(m128) * (size_t)n_a);      /* the chains' anchors only; n_a <= the anchors s->a was sized for */
                sort128x(s->a, s->a2, n_a);
                memset(s->t, 0, 4 * (size_t)n_a);
                mmo_lchain_rmq_fill(o->max_gap, o->rmq_inner_dist, o->bw_long, o->max_chain_skip, o->rmq_size_cap, chn_pen_gap, chn_pen_skip,
                                    n_a, (const mma_anchor *)s->a, s->f, s->p, s->t);
                n_u = chain_backtrack(o, s, n_a, o->bw_long, &best);
                tr->rechained |= 2; tr->n_chain = n_u; tr->best_score = best;
                if (n_u > 0) compact_chains(s);
            }
        }
        if (n_u > 0) {
            mma_align_read(o, idx->ref, idx->cstart, idx->n_contigs, seq, (int32_t)len, n_u, s->u, (mma_anchor *)s->b,
                           s->n_mini_pos, s->mini_pos, &res);
            tr->n_aligned = res.n_aligned; tr->n_regs = res.n_regs; tr->dp_max = res.dp_max; tr->sig = res.sig;
            tr->flag = res.n_regs > 0;
        } else tr->flag = 0;
    }
}

void mmo_map(const mmo_index *idx, const mmo_opts *o, const uint8_t *seq, int64_t len, mmo_trace *tr)
{
    scratch_t s;
    memset(&s, 0, sizeof(s));
    map_one(idx, o, &s, seq, len, tr);
    scratch_free(&s);
}

typedef struct {
    const mmo_index *idx; const mmo_opts *o; const uint8_t *bases; const uint64_t *offsets;
    uint64_t n_reads; uint8_t *flags; mmo_trace *traces; uint64_t *next;
} job_t;

static void *worker(void *arg)
{
    job_t *jb = (job_t *)arg;
    scratch_t s;
    const uint64_t CH = 64;       /* small chunks off one atomic counter: every thread stays busy to the end of the batch */
    memset(&s, 0, sizeof(s));
    for (;;) {
        uint64_t b, e, r;
        b = __atomic_fetch_add(jb->next, CH, __ATOMIC_RELAXED);
        if (b >= jb->n_reads) break;
        e = b + CH < jb->n_reads ? b + CH : jb->n_reads;
        for (r = b; r < e; ++r) {
            mmo_trace tr;
            map_one(jb->idx, jb->o, &s, jb->bases + jb->offsets[r], (int64_t)(jb->offsets[r + 1] - jb->offsets[r]), &tr);
            jb->flags[r] = (uint8_t)tr.flag;
            if (jb->traces) jb->traces[r] = tr;
        }
    }
    scratch_free(&s);
    return 0;
}

void mmo_classify_batch(const mmo_index *idx, const mmo_opts *o, const uint8_t *bases,
                        const uint64_t *offsets, uint64_t n_reads, uint8_t *flags,
                        mmo_trace *traces, int n_threads)
{
    pthread_t th[1024];
    uint64_t next = 0;
    job_t jb = { idx, o, bases, offsets, n_reads, flags, traces, &next };
    int i;
    pthread_once(&nt4_once, nt4_fill);
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 1024) n_threads = 1024;
    if (n_threads == 1) { worker(&jb); return; }
    for (i = 0; i < n_threads; ++i) pthread_create(&th[i], 0, worker, &jb);
    for (i = 0; i < n_threads; ++i) pthread_join(th[i], 0);
}
