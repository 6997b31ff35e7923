/*
 * mm_rmq.c — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See mm_oracle.h.
 *
 * *** PARITY UNPINNED ***: restates mg_lchain_rmq of lh3/minimap2 (~v2.28, lchain.c) and the balanced tree it runs on
 * (krmq.h: an AVL tree keyed by (y, i) whose nodes carry a pointer to the minimum-priority node of their subtree).  The
 * reference reaches it through `aligner.map()` (/root/reference/src/cleaner.rs:552) for every long-read preset
 * (`map_ont()/lrhq()/map_hifi()`, :457-458,465): mm_map_frag re-chains a read with bw_long when the first chaining
 * pass left more than one chain (last sentence).
 *
 * Why the tree is restated literally instead of "range minimum over a window": krmq_rmq resolves equal priorities by
 * the shape of the tree and by which subtree-minimum pointers rotations happened to carry over (krmq_rotate1/2 hand the
 * old root's pointer to the new root instead of recomputing it), and the predecessor an anchor gets decides the chains.
 * The statements below follow upstream's insert / erase / rotate / rmq one for one, on an index-based node pool.
 */
#include "mm_oracle.h"
#include "mm_align.h"
#include <stdlib.h>
#include <string.h>

#define RQ_MAX_DEPTH 64
#define NIL (-1)

typedef struct {
    int32_t y; int64_t i; double pri;
    int32_t p[2], s;          /* children, subtree minimum (node indices) */
    signed char balance; uint32_t size;
} rq_node;

typedef struct {
    rq_node *n; int32_t cap, n_used, free_head;      /* free list through p[0] */
    int32_t root;
} rq_tree;

static int32_t rq_alloc(rq_tree *t)
{
    int32_t k;
    if (t->free_head != NIL) { k = t->free_head; t->free_head = t->n[k].p[0]; return k; }
    if (t->n_used == t->cap) { t->cap = t->cap ? t->cap * 2 : 256; t->n = (rq_node *)realloc(t->n, sizeof(rq_node) * (size_t)t->cap); }
    return t->n_used++;
}
static void rq_free(rq_tree *t, int32_t k) { t->n[k].p[0] = t->free_head; t->free_head = k; }

#define N(k) (t->n[k])
static inline int rq_cmp_key(int32_t ay, int64_t ai, const rq_node *b)
{   /* lc_elem_cmp */
    return ay < b->y ? -1 : ay > b->y ? 1 : (ai > b->i) - (ai < b->i);
}
static inline int rq_lt2(const rq_tree *t, int32_t a, int32_t b) { return N(a).pri < N(b).pri; }
static inline uint32_t rq_size_child(const rq_tree *t, int32_t q, int i) { return N(q).p[i] != NIL ? N(N(q).p[i]).size : 0; }

/* krmq_update_min(p, q, r): p's subtree minimum from p itself and the minima of the two given subtrees, in that order */
static inline void rq_update_min(rq_tree *t, int32_t p, int32_t q, int32_t r)
{
    N(p).s = (q == NIL || rq_lt2(t, p, N(q).s)) ? p : N(q).s;
    N(p).s = (r == NIL || rq_lt2(t, N(p).s, N(r).s)) ? N(p).s : N(r).s;
}

/* one rotation: (a,(b,c)q)p => ((a,b)p,c)q */
static int32_t rq_rotate1(rq_tree *t, int32_t p, int dir)
{
    const int opp = 1 - dir;
    const int32_t q = N(p).p[opp], s = N(p).s;
    const uint32_t size_p = N(p).size;
    N(p).size -= N(q).size - rq_size_child(t, q, dir);
    N(q).size = size_p;
    rq_update_min(t, p, N(p).p[dir], N(q).p[dir]);
    N(q).s = s;
    N(p).p[opp] = N(q).p[dir];
    N(q).p[dir] = p;
    return q;
}

/* two consecutive rotations: (a,((b,c)r,d)q)p => ((a,b)p,(c,d)q)r */
static int32_t rq_rotate2(rq_tree *t, int32_t p, int dir)
{
    int b1;
    const int opp = 1 - dir;
    const int32_t q = N(p).p[opp], r = N(q).p[dir], s = N(p).s;
    const uint32_t size_x_dir = rq_size_child(t, r, dir);
    N(r).size = N(p).size;
    N(p).size -= N(q).size - size_x_dir;
    N(q).size -= size_x_dir + 1;
    rq_update_min(t, p, N(p).p[dir], N(r).p[dir]);
    rq_update_min(t, q, N(q).p[opp], N(r).p[opp]);
    N(r).s = s;
    N(p).p[opp] = N(r).p[dir];
    N(r).p[dir] = p;
    N(q).p[dir] = N(r).p[opp];
    N(r).p[opp] = q;
    b1 = dir == 0 ? +1 : -1;
    if (N(r).balance == b1) { N(q).balance = 0; N(p).balance = (signed char)-b1; }
    else if (N(r).balance == 0) N(q).balance = N(p).balance = 0;
    else { N(q).balance = (signed char)b1; N(p).balance = 0; }
    N(r).balance = 0;
    return r;
}

static void rq_insert(rq_tree *t, int32_t x)
{
    unsigned char stack[RQ_MAX_DEPTH];
    int32_t path[RQ_MAX_DEPTH];
    int32_t bp, bq, p, q, r;
    int i, which = 0, top, b1, path_len;
    bp = t->root; bq = NIL;
    for (p = bp, q = bq, top = path_len = 0; p != NIL; q = p, p = N(p).p[which]) {
        const int cmp = rq_cmp_key(N(x).y, N(x).i, &N(p));
        if (cmp == 0) return;     /* (y, i) is unique: never taken */
        if (N(p).balance != 0) { bq = q; bp = p; top = 0; }
        stack[top++] = (unsigned char)(which = (cmp > 0));
        path[path_len++] = p;
    }
    N(x).balance = 0; N(x).size = 1; N(x).p[0] = N(x).p[1] = NIL; N(x).s = x;
    if (q == NIL) t->root = x;
    else N(q).p[which] = x;
    if (bp == NIL) return;
    for (i = 0; i < path_len; ++i) ++N(path[i]).size;
    for (i = path_len - 1; i >= 0; --i) {
        rq_update_min(t, path[i], N(path[i]).p[0], N(path[i]).p[1]);
        if (N(path[i]).s != x) break;
    }
    for (p = bp, top = 0; p != x; p = N(p).p[stack[top]], ++top) {
        if (stack[top] == 0) --N(p).balance;
        else ++N(p).balance;
    }
    if (N(bp).balance > -2 && N(bp).balance < 2) return;
    which = (N(bp).balance < 0);
    b1 = which == 0 ? +1 : -1;
    q = N(bp).p[1 - which];
    if (N(q).balance == b1) {
        r = rq_rotate1(t, bp, which);
        N(q).balance = N(bp).balance = 0;
    } else r = rq_rotate2(t, bp, which);
    if (bq == NIL) t->root = r;
    else N(bq).p[bp != N(bq).p[0]] = r;
}

/* krmq_erase of the node with key (y, i); returns its index or NIL.  path[0] stands for upstream's `fake` node. */
static int32_t rq_erase(rq_tree *t, int32_t ky, int64_t ki)
{
    int32_t p, path[RQ_MAX_DEPTH], fake;
    unsigned char dir[RQ_MAX_DEPTH];
    int i, d = 0, cmp;
    if (t->root == NIL) return NIL;
    fake = rq_alloc(t);
    N(fake) = N(t->root);       /* fake = **root_ */
    N(fake).p[0] = t->root; N(fake).p[1] = NIL;
    for (cmp = -1, p = fake; cmp; cmp = rq_cmp_key(ky, ki, &N(p))) {
        const int which = (cmp > 0);
        dir[d] = (unsigned char)which;
        path[d++] = p;
        p = N(p).p[which];
        if (p == NIL) { rq_free(t, fake); return NIL; }
    }
    for (i = 1; i < d; ++i) --N(path[i]).size;
    if (N(p).p[1] == NIL) {
        N(path[d - 1]).p[dir[d - 1]] = N(p).p[0];
    } else {
        int32_t q = N(p).p[1];
        if (N(q).p[0] == NIL) {
            N(q).p[0] = N(p).p[0];
            N(q).balance = N(p).balance;
            N(path[d - 1]).p[dir[d - 1]] = q;
            path[d] = q; dir[d++] = 1;
            N(q).size = N(p).size - 1;
        } else {
            int32_t r;
            const int e = d++;
            for (;;) {
                dir[d] = 0;
                path[d++] = q;
                r = N(q).p[0];
                if (N(r).p[0] == NIL) break;
                q = r;
            }
            N(r).p[0] = N(p).p[0];
            N(q).p[0] = N(r).p[1];
            N(r).p[1] = N(p).p[1];
            N(r).balance = N(p).balance;
            N(path[e - 1]).p[dir[e - 1]] = r;
            path[e] = r; dir[e] = 1;
            for (i = e + 1; i < d; ++i) --N(path[i]).size;
            N(r).size = N(p).size - 1;
        }
    }
    for (i = d - 1; i >= 0; --i) rq_update_min(t, path[i], N(path[i]).p[0], N(path[i]).p[1]);
    while (--d > 0) {
        const int32_t q = path[d];
        int which, other, b1 = 1, b2 = 2;
        which = dir[d]; other = 1 - which;
        if (which) { b1 = -b1; b2 = -b2; }
        N(q).balance = (signed char)(N(q).balance + b1);
        if (N(q).balance == b1) break;
        else if (N(q).balance == b2) {
            const int32_t r = N(q).p[other];
            if (N(r).balance == -b1) {
                N(path[d - 1]).p[dir[d - 1]] = rq_rotate2(t, q, which);
            } else {
                N(path[d - 1]).p[dir[d - 1]] = rq_rotate1(t, q, which);
                if (N(r).balance == 0) {
                    N(r).balance = (signed char)-b1;
                    N(q).balance = (signed char)b1;
                    break;
                } else N(r).balance = N(q).balance = 0;
            }
        }
    }
    t->root = N(fake).p[0];
    rq_free(t, fake);
    return p;
}

/* krmq_rmq over the CLOSED key interval [(lo_y, lo_i), (hi_y, hi_i)] */
static int32_t rq_rmq(const rq_tree *t, int32_t lo_y, int64_t lo_i, int32_t hi_y, int64_t hi_i)
{
    int32_t p = t->root, path[2][RQ_MAX_DEPTH], min;
    int plen[2] = {0, 0}, pcmp[2][RQ_MAX_DEPTH], i, cmp, lca;
    if (t->root == NIL) return NIL;
    while (p != NIL) {
        cmp = rq_cmp_key(lo_y, lo_i, &N(p));
        path[0][plen[0]] = p; pcmp[0][plen[0]++] = cmp;
        if (cmp < 0) p = N(p).p[0];
        else if (cmp > 0) p = N(p).p[1];
        else break;
    }
    p = t->root;
    while (p != NIL) {
        cmp = rq_cmp_key(hi_y, hi_i, &N(p));
        path[1][plen[1]] = p; pcmp[1][plen[1]++] = cmp;
        if (cmp < 0) p = N(p).p[0];
        else if (cmp > 0) p = N(p).p[1];
        else break;
    }
    for (i = 0; i < plen[0] && i < plen[1]; ++i)
        if (path[0][i] == path[1][i] && pcmp[0][i] <= 0 && pcmp[1][i] >= 0) break;
    if (i == plen[0] || i == plen[1]) return NIL;
    lca = i; min = path[0][lca];
    for (i = lca + 1; i < plen[0]; ++i) {
        if (pcmp[0][i] <= 0) {
            if (rq_lt2(t, path[0][i], min)) min = path[0][i];
            if (N(path[0][i]).p[1] != NIL && rq_lt2(t, N(N(path[0][i]).p[1]).s, min)) min = N(N(path[0][i]).p[1]).s;
        }
    }
    for (i = lca + 1; i < plen[1]; ++i) {
        if (pcmp[1][i] >= 0) {
            if (rq_lt2(t, path[1][i], min)) min = path[1][i];
            if (N(path[1][i]).p[0] != NIL && rq_lt2(t, N(N(path[1][i]).p[0]).s, min)) min = N(N(path[1][i]).p[0]).s;
        }
    }
    return min;
}

/* krmq_interval's lower bound: the largest element <= (y, i), with the root-to-node path kept for krmq_itr_prev */
typedef struct { int32_t stack[RQ_MAX_DEPTH]; int top; } rq_itr;      /* top < 0: exhausted */

static int rq_itr_find_le(const rq_tree *t, int32_t ky, int64_t ki, rq_itr *it)
{   /* positions the iterator at the largest element <= key; 0 if there is none */
    int32_t p = t->root;
    int d = 0, best = -1;
    while (p != NIL) {
        const int cmp = rq_cmp_key(ky, ki, &N(p));
        it->stack[d++] = p;
        if (cmp < 0) p = N(p).p[0];
        else if (cmp > 0) { best = d; p = N(p).p[1]; }
        else { best = d; break; }
    }
    if (best < 0) { it->top = -1; return 0; }
    it->top = best - 1;
    return 1;
}

static int rq_itr_prev(const rq_tree *t, rq_itr *it)
{   /* in-order predecessor (krmq_itr_prev); 0 when there is none */
    int32_t p;
    if (it->top < 0) return 0;
    p = N(it->stack[it->top]).p[0];
    if (p != NIL) {
        for (; p != NIL; p = N(p).p[1]) it->stack[++it->top] = p;
        return 1;
    }
    do { p = it->stack[it->top--]; } while (it->top >= 0 && p == N(it->stack[it->top]).p[0]);
    return it->top >= 0;
}
#undef N

/* comput_sc_simple */
static inline int32_t comput_sc_simple(const mma_anchor *ai, const mma_anchor *aj, float chn_pen_gap, float chn_pen_skip, int32_t *exact, int32_t *width)
{
    int32_t dq = (int32_t)ai->y - (int32_t)aj->y, dr, dd, dg, q_span, sc;
    dr = (int32_t)(ai->x - aj->x);
    *width = dd = dr > dq ? dr - dq : dq - dr;
    dg = dr < dq ? dr : dq;
    q_span = (int32_t)(aj->y >> 32 & 0xff);
    sc = q_span < dg ? q_span : dg;
    if (exact) *exact = (dd == 0 && dg <= q_span);
    if (dd || dq > q_span) {
        float lin_pen, log_pen;
        lin_pen = chn_pen_gap * (float)dd + chn_pen_skip * (float)dg;
        log_pen = dd >= 1 ? mmo_log2((float)(dd + 1)) : 0.0f;
        sc -= (int32_t)(lin_pen + .5f * log_pen);
    }
    return sc;
}

/* mg_lchain_rmq's scoring pass: f[] = best chain score ending at each anchor, p[] = its predecessor (-1: none).
 * a[] sorted by x.  t[] (n int32, zeroed by the caller) is the skip-mark array shared with the backtrack. */
void mmo_lchain_rmq_fill(int max_dist, int max_dist_inner, int bw, int max_chn_skip, int cap_rmq_size,
                         float chn_pen_gap, float chn_pen_skip, int64_t n, const mma_anchor *a, int32_t *f, int64_t *p, int32_t *t)
{
    rq_tree T[2];
    int64_t i, i0, st = 0, st_inner = 0;
    int k;
    for (k = 0; k < 2; ++k) { T[k].n = 0; T[k].cap = T[k].n_used = 0; T[k].free_head = NIL; T[k].root = NIL; }
    if (max_dist < bw) max_dist = bw;
    if (max_dist_inner < 0) max_dist_inner = 0;
    if (max_dist_inner > max_dist) max_dist_inner = max_dist;
#define ROOT_SIZE(tr) ((tr).root != NIL ? (int64_t)(tr).n[(tr).root].size : 0)
    for (i = i0 = 0; i < n; ++i) {
        int64_t max_j = -1;
        const int32_t q_span = (int32_t)(a[i].y >> 32 & 0xff);
        int32_t max_f = q_span, q;
        if (i0 < i && a[i0].x != a[i].x) {      /* add in-range anchors */
            int64_t j;
            for (j = i0; j < i; ++j) {
                const double pri = -(f[j] + 0.5 * chn_pen_gap * ((int32_t)a[j].x + (int32_t)a[j].y));
                for (k = 0; k < (max_dist_inner > 0 ? 2 : 1); ++k) {
                    const int32_t x = rq_alloc(&T[k]);
                    T[k].n[x].y = (int32_t)a[j].y; T[k].n[x].i = j; T[k].n[x].pri = pri;
                    rq_insert(&T[k], x);
                }
            }
            i0 = i;
        }
        while (st < i && (a[i].x >> 32 != a[st].x >> 32 || a[i].x > a[st].x + (uint64_t)max_dist || ROOT_SIZE(T[0]) > cap_rmq_size)) {
            const int32_t e = rq_erase(&T[0], (int32_t)a[st].y, st);
            if (e != NIL) rq_free(&T[0], e);
            ++st;
        }
        if (max_dist_inner > 0) {
            while (st_inner < i && (a[i].x >> 32 != a[st_inner].x >> 32 || a[i].x > a[st_inner].x + (uint64_t)max_dist_inner || ROOT_SIZE(T[1]) > cap_rmq_size)) {
                const int32_t e = rq_erase(&T[1], (int32_t)a[st_inner].y, st_inner);
                if (e != NIL) rq_free(&T[1], e);
                ++st_inner;
            }
        }
        if ((q = rq_rmq(&T[0], (int32_t)a[i].y - max_dist, INT32_MAX, (int32_t)a[i].y, 0)) != NIL) {
            int32_t sc, exact, width, n_skip = 0;
            int64_t j = T[0].n[q].i;
            sc = f[j] + comput_sc_simple(&a[i], &a[j], chn_pen_gap, chn_pen_skip, &exact, &width);
            if (width <= bw && sc > max_f) { max_f = sc; max_j = j; }
            if (!exact && T[1].root != NIL && (int32_t)a[i].y > 0) {
                rq_itr it;
                if (rq_itr_find_le(&T[1], (int32_t)a[i].y - 1, n, &it)) {
                    do {
                        const rq_node *e = &T[1].n[it.stack[it.top]];
                        if (e->y < (int32_t)a[i].y - max_dist_inner) break;
                        j = e->i;
                        sc = f[j] + comput_sc_simple(&a[i], &a[j], chn_pen_gap, chn_pen_skip, 0, &width);
                        if (width <= bw) {
                            if (sc > max_f) {
                                max_f = sc; max_j = j;
                                if (n_skip > 0) --n_skip;
                            } else if (t[j] == (int32_t)i) {
                                if (++n_skip > max_chn_skip) break;
                            }
                            if (p[j] >= 0) t[p[j]] = (int32_t)i;
                        }
                    } while (rq_itr_prev(&T[1], &it));
                }
            }
        }
        f[i] = max_f; p[i] = max_j;
    }
#undef ROOT_SIZE
    free(T[0].n); free(T[1].n);
}

/* ------------------------------------------------------------------------------------------------
 * test hooks (tests/test_long_oracle_cpu.py)
 * ---------------------------------------------------------------------------------------------- */
/* Random inserts / erases / closed-interval queries against a brute-force scan over the live set; priorities are distinct,
 * so the answer does not depend on the tie rules.  Also checks the AVL shape (balance factors, sizes, subtree minima
 * reachable).  Returns the number of disagreements (0 = pass). */
static int rq_check(const rq_tree *t, int32_t p, int *height, uint32_t *size)
{
    int hl = 0, hr = 0, bad = 0;
    uint32_t sl = 0, sr = 0;
    if (p == NIL) { *height = 0; *size = 0; return 0; }
    bad += rq_check(t, t->n[p].p[0], &hl, &sl);
    bad += rq_check(t, t->n[p].p[1], &hr, &sr);
    if (hr - hl != t->n[p].balance || hr - hl > 1 || hr - hl < -1) ++bad;
    if (sl + sr + 1 != t->n[p].size) ++bad;
    {   /* the stored minimum has the smallest priority of the subtree */
        double m = t->n[p].pri;
        if (t->n[p].p[0] != NIL && t->n[t->n[t->n[p].p[0]].s].pri < m) m = t->n[t->n[t->n[p].p[0]].s].pri;
        if (t->n[p].p[1] != NIL && t->n[t->n[t->n[p].p[1]].s].pri < m) m = t->n[t->n[t->n[p].p[1]].s].pri;
        if (t->n[t->n[p].s].pri != m) ++bad;
    }
    *height = 1 + (hl > hr ? hl : hr); *size = sl + sr + 1;
    return bad;
}

int mmo_rmq_selftest(uint64_t seed, int n_ops, int key_range)
{
    rq_tree T; int bad = 0, op, n_live = 0, cap = n_ops + 1;
    int32_t *ly = (int32_t *)malloc(4 * (size_t)cap); int64_t *li = (int64_t *)malloc(8 * (size_t)cap); double *lp = (double *)malloc(8 * (size_t)cap);
    uint64_t s = seed * 0x9E3779B97F4A7C15ULL + 1;
#define RND() (s ^= s << 13, s ^= s >> 7, s ^= s << 17, s)
    T.n = 0; T.cap = T.n_used = 0; T.free_head = NIL; T.root = NIL;
    for (op = 0; op < n_ops; ++op) {
        const unsigned r = (unsigned)(RND() % 10);
        if (r < 5 || n_live == 0) {                         /* insert a new (y, i) with a fresh priority */
            const int32_t x = rq_alloc(&T);
            T.n[x].y = (int32_t)(RND() % (uint64_t)key_range); T.n[x].i = op; T.n[x].pri = (double)(RND() >> 11) + op * 1e-3;
            ly[n_live] = T.n[x].y; li[n_live] = op; lp[n_live] = T.n[x].pri; ++n_live;
            rq_insert(&T, x);
        } else if (r < 7) {                                 /* erase a live element */
            const int k = (int)(RND() % (uint64_t)n_live);
            const int32_t e = rq_erase(&T, ly[k], li[k]);
            if (e == NIL || T.n[e].i != li[k]) ++bad; else rq_free(&T, e);
            ly[k] = ly[n_live - 1]; li[k] = li[n_live - 1]; lp[k] = lp[n_live - 1]; --n_live;
        } else {                                            /* query */
            int32_t a = (int32_t)(RND() % (uint64_t)key_range), b = (int32_t)(RND() % (uint64_t)key_range), q;
            int k, best = -1;
            if (a > b) { const int32_t tt = a; a = b; b = tt; }
            q = rq_rmq(&T, a, INT32_MAX, b, 0);
            for (k = 0; k < n_live; ++k) {
                const int in = (ly[k] > a || (ly[k] == a && li[k] >= INT32_MAX)) && (ly[k] < b || (ly[k] == b && li[k] <= 0));
                if (in && (best < 0 || lp[k] < lp[best])) best = k;
            }
            if ((q == NIL) != (best < 0)) ++bad;
            else if (q != NIL && T.n[q].i != li[best]) ++bad;
        }
        if ((op & 63) == 0) { int h; uint32_t sz; bad += rq_check(&T, T.root, &h, &sz); if ((int)sz != n_live) ++bad; }
    }
#undef RND
    free(T.n); free(ly); free(li); free(lp);
    return bad;
}

/* The answers of a random insert / erase / query sequence with HEAVILY tied priorities (values 0..9): which element krmq_rmq returns
 * among equal minima is decided by the shape of the tree and by the subtree-minimum pointers its rotations carried over.  The device
 * restatement (scrubby_amd/csrc/sh_rmq_tree.h, compiled for the host by tests/test_rmq_tree_cpu.py) must give the same sequence.
 * Erases are FIFO (mg_lchain_rmq's window) when fifo != 0.  out[] receives one int64 per query: the element's i, or -1. */
int64_t mmo_rmq_trace(uint64_t seed, int n_ops, int key_range, int fifo, int64_t *out)
{
    rq_tree T; int op; int64_t n_out = 0, head = 0, n_all = 0;
    int32_t *ly = (int32_t *)malloc(4 * (size_t)(n_ops + 1)); int64_t *li = (int64_t *)malloc(8 * (size_t)(n_ops + 1));
    uint64_t s = seed * 0x9E3779B97F4A7C15ULL + 1;
#define RND() (s ^= s << 13, s ^= s >> 7, s ^= s << 17, s)
    T.n = 0; T.cap = T.n_used = 0; T.free_head = NIL; T.root = NIL;
    for (op = 0; op < n_ops; ++op) {
        const unsigned r = (unsigned)(RND() % 10);
        const int64_t n_live = n_all - head;
        if (r < 5 || n_live == 0) {
            const int32_t x = rq_alloc(&T);
            T.n[x].y = (int32_t)(RND() % (uint64_t)key_range); T.n[x].i = op; T.n[x].pri = (double)(RND() % 10);
            ly[n_all] = T.n[x].y; li[n_all] = op; ++n_all;
            rq_insert(&T, x);
        } else if (r < 7) {
            const int64_t k = fifo ? head : head + (int64_t)(RND() % (uint64_t)n_live);
            const int32_t e = rq_erase(&T, ly[k], li[k]);
            if (e != NIL) rq_free(&T, e);
            ly[k] = ly[head]; li[k] = li[head]; ++head;
        } else {
            int32_t a = (int32_t)(RND() % (uint64_t)key_range), b = (int32_t)(RND() % (uint64_t)key_range), q;
            if (a > b) { const int32_t tt = a; a = b; b = tt; }
            q = rq_rmq(&T, a, INT32_MAX, b, 0);
            out[n_out++] = q == NIL ? -1 : T.n[q].i;
        }
    }
#undef RND
    free(T.n); free(ly); free(li);
    return n_out;
}
