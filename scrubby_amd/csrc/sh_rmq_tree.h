// sh_rmq_tree.h — mg_lchain_rmq's balanced tree, for the reads the wave scan of sh_long.h cannot answer bit for bit.
//
// minimap2 (the crate behind /root/reference/src/cleaner.rs:552, presets :457-458,465) keeps the look-back window of its long join in two
// balanced trees keyed by (y, i) whose nodes carry a pointer to the minimum-priority node of their subtree (krmq.h).  lr_rmq_fill asks the
// same questions of a scan; its answers are the trees' as long as the smallest priority in a query interval is held by ONE candidate.  When
// two candidates tie, krmq_rmq returns the one the shape of the tree and the history of its rotations favour (krmq_rotate1/2 hand the old
// root's subtree-minimum pointer to the new root instead of recomputing it), and the predecessor an anchor gets decides the chains.  Reads
// that meet such a tie keep the scan for everything that has width and ask a literal tree - insert / erase / rotate / rmq as upstream states
// them, on an index-based node pool, maintained by ONE lane - at the ties only (lr_rmq_fill<NR, true>); reads the LDS ring cannot hold, or
// with more anchors than rmq_size_cap (the cap evicts out of order), run both trees on one lane (lr_rmq_fill_tree).
// The pool lives in the wave's HBM scratch; the nodes every walk passes - the top of the tree - are mirrored in a direct-mapped LDS cache
// (write-through), which is what makes a read of 10^5 anchors a matter of tenths of a second instead of seconds.
// The CPU oracle holds the same restatement (oracle/mm_rmq.c, checked there against a brute-force scan); the two share no code.
#pragma once
#include <stdint.h>

#define RQ_MAX_DEPTH 64
#define RQ_NIL (-1)
// The node cache is written for LDS and checked on the host (tests/test_rmq_tree_cpu.py); the device build leaves it out until its generic
// pointers into LDS are sorted out: with RQ_CACHE_ON 0 every access goes to the pool and the cache code folds away.
#ifndef RQ_CACHE_ON
#define RQ_CACHE_ON 0
#endif

struct alignas(16) RqNode {      // 48 bytes, 16-byte aligned: a node moves between pool and cache in 128-bit pieces, which LDS only takes aligned
    int32_t y, i; double pri;
    int32_t p[2], s;           // children, subtree minimum (node indices)
    int32_t balance; uint32_t size;
};
struct RqCache { RqNode *c; int32_t *tag; int32_t mask; };      // LDS: mask + 1 entries (a power of two); c == nullptr: no cache
struct RqTree { RqNode *n; RqCache C; int32_t cap, n_used, free_head, root, bad; };      // bad: an index outside the pool was asked for (the caller gives the read up instead of touching memory it does not own)

__device__ inline void rq_init(RqTree &t, RqNode *pool, int32_t cap, RqCache C = RqCache{nullptr, nullptr, 0})
{
    t.n = pool; t.C = C; t.cap = cap; t.n_used = 0; t.free_head = RQ_NIL; t.root = RQ_NIL; t.bad = 0;
    if (RQ_CACHE_ON && C.c) for (int32_t k = 0; k <= C.mask; ++k) C.tag[k] = RQ_NIL;
}
// the copy of node k to read from (and to write to, together with the pool: RQ_SET); good until the next rq_at
__device__ inline RqNode *rq_at(const RqTree &t, int32_t k)
{
    if ((uint32_t)k >= (uint32_t)t.cap) { if (!t.bad) const_cast<RqTree &>(t).bad = 1; k = 0; }
    if (!RQ_CACHE_ON || !t.C.c) return t.n + k;
    const int32_t sl = k & t.C.mask;
    if (t.C.tag[sl] != k) { t.C.c[sl] = t.n[k]; t.C.tag[sl] = k; }
    return t.C.c + sl;
}
#define RQ_GET(k, fld) (rq_at(t, (k))->fld)
#define RQ_SET(k, fld, v) do { const int32_t k__ = (k); const auto v__ = (v); rq_at(t, k__)->fld = v__; if (RQ_CACHE_ON && t.C.c && (uint32_t)k__ < (uint32_t)t.cap) t.n[k__].fld = v__; } while (0)

__device__ inline int32_t rq_alloc(RqTree &t)
{
    if (t.free_head != RQ_NIL) { const int32_t k = t.free_head; t.free_head = RQ_GET(k, p[0]); return k; }
    if (t.n_used >= t.cap) return RQ_NIL;      // cannot happen: cap = window + 2
    return t.n_used++;
}
__device__ inline void rq_free(RqTree &t, int32_t k) { RQ_SET(k, p[0], t.free_head); t.free_head = k; }
// a fresh node, not yet in the tree
__device__ inline void rq_node_set(RqTree &t, int32_t k, int32_t y, int32_t i, double pri)
{
    RqNode z; z.y = y; z.i = i; z.pri = pri; z.p[0] = z.p[1] = RQ_NIL; z.s = k; z.balance = 0; z.size = 1;
    if ((uint32_t)k >= (uint32_t)t.cap) { if (!t.bad) t.bad = 2; return; }
    t.n[k] = z;
    if (RQ_CACHE_ON && t.C.c) { const int32_t sl = k & t.C.mask; t.C.c[sl] = z; t.C.tag[sl] = k; }
}

__device__ inline int rq_cmp_key(int32_t ay, int32_t ai, int32_t by, int32_t bi)
{   // lc_elem_cmp
    return ay < by ? -1 : ay > by ? 1 : (ai > bi) - (ai < bi);
}
__device__ inline int rq_cmp_node(const RqTree &t, int32_t ay, int32_t ai, int32_t k) { const RqNode *b = rq_at(t, k); return rq_cmp_key(ay, ai, b->y, b->i); }
__device__ inline bool rq_lt2(const RqTree &t, int32_t a, int32_t b) { const double pa = RQ_GET(a, pri); return pa < RQ_GET(b, pri); }
__device__ inline uint32_t rq_size_child(const RqTree &t, int32_t q, int i) { const int32_t c = RQ_GET(q, p[i]); return c != RQ_NIL ? RQ_GET(c, size) : 0u; }

// krmq_update_min(p, q, r): p's subtree minimum from p itself and the minima of the two given subtrees, in that order
__device__ inline void rq_update_min(RqTree &t, int32_t p, int32_t q, int32_t r)
{
    int32_t s = p;
    if (q != RQ_NIL) { const int32_t qs = RQ_GET(q, s); if (!rq_lt2(t, p, qs)) s = qs; }
    if (r != RQ_NIL) { const int32_t rs = RQ_GET(r, s); if (!rq_lt2(t, s, rs)) s = rs; }
    RQ_SET(p, s, s);
}

// one rotation: (a,(b,c)q)p => ((a,b)p,c)q
__device__ inline int32_t rq_rotate1(RqTree &t, int32_t p, int dir)
{
    const int opp = 1 - dir;
    const int32_t q = RQ_GET(p, p[opp]), s = RQ_GET(p, s);
    const uint32_t size_p = RQ_GET(p, size);
    RQ_SET(p, size, size_p - (RQ_GET(q, size) - rq_size_child(t, q, dir)));
    RQ_SET(q, size, size_p);
    rq_update_min(t, p, RQ_GET(p, p[dir]), RQ_GET(q, p[dir]));
    RQ_SET(q, s, s);
    RQ_SET(p, p[opp], RQ_GET(q, p[dir]));
    RQ_SET(q, p[dir], p);
    return q;
}

// two consecutive rotations: (a,((b,c)r,d)q)p => ((a,b)p,(c,d)q)r
__device__ inline int32_t rq_rotate2(RqTree &t, int32_t p, int dir)
{
    const int opp = 1 - dir;
    const int32_t q = RQ_GET(p, p[opp]), r = RQ_GET(q, p[dir]), s = RQ_GET(p, s);
    const uint32_t size_x_dir = rq_size_child(t, r, dir);
    const uint32_t size_p = RQ_GET(p, size), size_q = RQ_GET(q, size);
    RQ_SET(r, size, size_p);
    RQ_SET(p, size, size_p - (size_q - size_x_dir));
    RQ_SET(q, size, size_q - (size_x_dir + 1));
    rq_update_min(t, p, RQ_GET(p, p[dir]), RQ_GET(r, p[dir]));
    rq_update_min(t, q, RQ_GET(q, p[opp]), RQ_GET(r, p[opp]));
    RQ_SET(r, s, s);
    RQ_SET(p, p[opp], RQ_GET(r, p[dir]));
    RQ_SET(r, p[dir], p);
    RQ_SET(q, p[dir], RQ_GET(r, p[opp]));
    RQ_SET(r, p[opp], q);
    const int b1 = dir == 0 ? +1 : -1;
    const int32_t rb = RQ_GET(r, balance);
    if (rb == b1) { RQ_SET(q, balance, 0); RQ_SET(p, balance, -b1); }
    else if (rb == 0) { RQ_SET(q, balance, 0); RQ_SET(p, balance, 0); }
    else { RQ_SET(q, balance, b1); RQ_SET(p, balance, 0); }
    RQ_SET(r, balance, 0);
    return r;
}

// x: a node prepared by rq_node_set
__device__ inline void rq_insert(RqTree &t, int32_t x)
{
    unsigned char stack[RQ_MAX_DEPTH];
    int32_t path[RQ_MAX_DEPTH];
    int32_t bp, bq, p, q, r;
    int i, which = 0, top, path_len;
    const int32_t xy = RQ_GET(x, y), xi = RQ_GET(x, i);
    bp = t.root; bq = RQ_NIL;
    for (p = bp, q = bq, top = path_len = 0; p != RQ_NIL; q = p, p = RQ_GET(p, p[which])) {
        const int cmp = rq_cmp_node(t, xy, xi, p);
        if (cmp == 0) return;     // (y, i) is unique: never taken
        if (RQ_GET(p, balance) != 0) { bq = q; bp = p; top = 0; }
        stack[top++] = (unsigned char)(which = (cmp > 0));
        path[path_len++] = p;
        if (path_len >= RQ_MAX_DEPTH - 1) { if (!t.bad) t.bad = 4; return; }
    }
    if (q == RQ_NIL) t.root = x;
    else RQ_SET(q, p[which], x);
    if (bp == RQ_NIL) return;
    for (i = 0; i < path_len; ++i) RQ_SET(path[i], size, RQ_GET(path[i], size) + 1u);
    for (i = path_len - 1; i >= 0; --i) {
        rq_update_min(t, path[i], RQ_GET(path[i], p[0]), RQ_GET(path[i], p[1]));
        if (RQ_GET(path[i], s) != x) break;
    }
    for (p = bp, top = 0; p != x; p = RQ_GET(p, p[stack[top]]), ++top) {
        if (stack[top] == 0) RQ_SET(p, balance, RQ_GET(p, balance) - 1);
        else RQ_SET(p, balance, RQ_GET(p, balance) + 1);
    }
    const int32_t bb = RQ_GET(bp, balance);
    if (bb > -2 && bb < 2) return;
    which = (bb < 0);
    const int b1 = which == 0 ? +1 : -1;
    q = RQ_GET(bp, p[1 - which]);
    if (RQ_GET(q, balance) == b1) {
        r = rq_rotate1(t, bp, which);
        RQ_SET(q, balance, 0); RQ_SET(bp, balance, 0);
    } else r = rq_rotate2(t, bp, which);
    if (bq == RQ_NIL) t.root = r;
    else { const int wi = bp != RQ_GET(bq, p[0]); RQ_SET(bq, p[wi], r); }      // (the index first: the macro evaluates its field expression twice)
}

// krmq_erase of the node with key (y, i); returns its index or RQ_NIL.  path[0] stands for upstream's `fake` node.
__device__ inline int32_t rq_erase(RqTree &t, int32_t ky, int32_t ki)
{
    int32_t p, path[RQ_MAX_DEPTH], fake;
    unsigned char dir[RQ_MAX_DEPTH];
    int i, d = 0, cmp;
    if (t.root == RQ_NIL) return RQ_NIL;
    fake = rq_alloc(t);
    if (fake == RQ_NIL) { if (!t.bad) t.bad = 3; return RQ_NIL; }
    {   // fake = **root_, with the tree below its left link
        RqNode z = *rq_at(t, t.root);
        z.p[0] = t.root; z.p[1] = RQ_NIL;
        t.n[fake] = z;
        if (RQ_CACHE_ON && t.C.c) { const int32_t sl = fake & t.C.mask; t.C.c[sl] = z; t.C.tag[sl] = fake; }
    }
    for (cmp = -1, p = fake; cmp; cmp = rq_cmp_node(t, ky, ki, p)) {
        const int which = (cmp > 0);
        dir[d] = (unsigned char)which;
        path[d++] = p;
        if (d >= RQ_MAX_DEPTH - 2) { if (!t.bad) t.bad = 5; rq_free(t, fake); return RQ_NIL; }
        p = RQ_GET(p, p[which]);
        if (p == RQ_NIL) { rq_free(t, fake); return RQ_NIL; }
    }
    for (i = 1; i < d; ++i) RQ_SET(path[i], size, RQ_GET(path[i], size) - 1u);
    const int32_t p_l = RQ_GET(p, p[0]), p_r = RQ_GET(p, p[1]), p_bal = RQ_GET(p, balance);
    const uint32_t p_size = RQ_GET(p, size);
    if (p_r == RQ_NIL) {
        RQ_SET(path[d - 1], p[dir[d - 1]], p_l);
    } else {
        int32_t q = p_r;
        if (RQ_GET(q, p[0]) == RQ_NIL) {
            RQ_SET(q, p[0], p_l);
            RQ_SET(q, balance, p_bal);
            RQ_SET(path[d - 1], p[dir[d - 1]], q);
            path[d] = q; dir[d++] = 1;
            RQ_SET(q, size, p_size - 1u);
        } else {
            int32_t r;
            const int e = d++;
            for (;;) {
                dir[d] = 0;
                path[d++] = q;
                if (d >= RQ_MAX_DEPTH - 1) { if (!t.bad) t.bad = 6; rq_free(t, fake); return RQ_NIL; }
                r = RQ_GET(q, p[0]);
                if (RQ_GET(r, p[0]) == RQ_NIL) break;
                q = r;
            }
            RQ_SET(r, p[0], p_l);
            RQ_SET(q, p[0], RQ_GET(r, p[1]));
            RQ_SET(r, p[1], p_r);
            RQ_SET(r, balance, p_bal);
            RQ_SET(path[e - 1], p[dir[e - 1]], r);
            path[e] = r; dir[e] = 1;
            for (i = e + 1; i < d; ++i) RQ_SET(path[i], size, RQ_GET(path[i], size) - 1u);
            RQ_SET(r, size, p_size - 1u);
        }
    }
    for (i = d - 1; i >= 0; --i) rq_update_min(t, path[i], RQ_GET(path[i], p[0]), RQ_GET(path[i], p[1]));
    while (--d > 0) {
        const int32_t q = path[d];
        int which, other, b1 = 1, b2 = 2;
        which = dir[d]; other = 1 - which;
        if (which) { b1 = -b1; b2 = -b2; }
        const int32_t qb = RQ_GET(q, balance) + b1;
        RQ_SET(q, balance, qb);
        if (qb == b1) break;
        else if (qb == b2) {
            const int32_t r = RQ_GET(q, p[other]);
            const int32_t rbal = RQ_GET(r, balance);
            if (rbal == -b1) {
                const int32_t nr = rq_rotate2(t, q, which);
                RQ_SET(path[d - 1], p[dir[d - 1]], nr);
            } else {
                const int32_t nr = rq_rotate1(t, q, which);
                RQ_SET(path[d - 1], p[dir[d - 1]], nr);
                if (rbal == 0) {
                    RQ_SET(r, balance, -b1);
                    RQ_SET(q, balance, b1);
                    break;
                } else { RQ_SET(r, balance, 0); RQ_SET(q, balance, 0); }
            }
        }
    }
    t.root = RQ_GET(fake, p[0]);
    rq_free(t, fake);
    return p;
}

// krmq_rmq over the CLOSED key interval [(lo_y, lo_i), (hi_y, hi_i)]
__device__ inline int32_t rq_rmq(const RqTree &t, int32_t lo_y, int32_t lo_i, int32_t hi_y, int32_t hi_i)
{
    int32_t p = t.root, path[2][RQ_MAX_DEPTH], min;
    int plen[2] = {0, 0}, pcmp[2][RQ_MAX_DEPTH], i, cmp, lca;
    if (t.root == RQ_NIL) return RQ_NIL;
    while (p != RQ_NIL) {
        cmp = rq_cmp_node(t, lo_y, lo_i, p);
        if (plen[0] >= RQ_MAX_DEPTH - 1) { if (!t.bad) const_cast<RqTree &>(t).bad = 7; return RQ_NIL; }
        path[0][plen[0]] = p; pcmp[0][plen[0]++] = cmp;
        if (cmp < 0) p = RQ_GET(p, p[0]);
        else if (cmp > 0) p = RQ_GET(p, p[1]);
        else break;
    }
    p = t.root;
    while (p != RQ_NIL) {
        cmp = rq_cmp_node(t, hi_y, hi_i, p);
        if (plen[1] >= RQ_MAX_DEPTH - 1) { if (!t.bad) const_cast<RqTree &>(t).bad = 8; return RQ_NIL; }
        path[1][plen[1]] = p; pcmp[1][plen[1]++] = cmp;
        if (cmp < 0) p = RQ_GET(p, p[0]);
        else if (cmp > 0) p = RQ_GET(p, p[1]);
        else break;
    }
    for (i = 0; i < plen[0] && i < plen[1]; ++i)
        if (path[0][i] == path[1][i] && pcmp[0][i] <= 0 && pcmp[1][i] >= 0) break;
    if (i == plen[0] || i == plen[1]) return RQ_NIL;
    lca = i; min = path[0][lca];
    for (i = lca + 1; i < plen[0]; ++i) {
        if (pcmp[0][i] <= 0) {
            if (rq_lt2(t, path[0][i], min)) min = path[0][i];
            const int32_t c = RQ_GET(path[0][i], p[1]);
            if (c != RQ_NIL) { const int32_t cs = RQ_GET(c, s); if (rq_lt2(t, cs, min)) min = cs; }
        }
    }
    for (i = lca + 1; i < plen[1]; ++i) {
        if (pcmp[1][i] >= 0) {
            if (rq_lt2(t, path[1][i], min)) min = path[1][i];
            const int32_t c = RQ_GET(path[1][i], p[0]);
            if (c != RQ_NIL) { const int32_t cs = RQ_GET(c, s); if (rq_lt2(t, cs, min)) min = cs; }
        }
    }
    return min;
}

// krmq_interval's lower bound and krmq_itr_prev: the largest element <= (y, i), then its in-order predecessors
struct RqItr { int32_t stack[RQ_MAX_DEPTH]; int top; };      // top < 0: exhausted
__device__ inline bool rq_itr_find_le(const RqTree &t, int32_t ky, int32_t ki, RqItr &it)
{
    int32_t p = t.root;
    int d = 0, best = -1;
    while (p != RQ_NIL) {
        const int cmp = rq_cmp_node(t, ky, ki, p);
        if (d >= RQ_MAX_DEPTH - 1) { if (!t.bad) const_cast<RqTree &>(t).bad = 9; it.top = -1; return false; }
        it.stack[d++] = p;
        if (cmp < 0) p = RQ_GET(p, p[0]);
        else if (cmp > 0) { best = d; p = RQ_GET(p, p[1]); }
        else { best = d; break; }
    }
    if (best < 0) { it.top = -1; return false; }
    it.top = best - 1;
    return true;
}
__device__ inline bool rq_itr_prev(const RqTree &t, RqItr &it)
{
    int32_t p;
    if (it.top < 0) return false;
    p = RQ_GET(it.stack[it.top], p[0]);
    if (p != RQ_NIL) {
        for (; p != RQ_NIL; p = RQ_GET(p, p[1])) { if (it.top >= RQ_MAX_DEPTH - 2) { if (!t.bad) const_cast<RqTree &>(t).bad = 10; it.top = -1; return false; } it.stack[++it.top] = p; }
        return true;
    }
    do { p = it.stack[it.top--]; } while (it.top >= 0 && p == RQ_GET(it.stack[it.top], p[0]));
    return it.top >= 0;
}
