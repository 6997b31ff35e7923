// sh_rmq_tree.h — mg_lchain_rmq on its own data structure, for the reads the wave scan of sh_long.h cannot answer bit for bit.
//
// minimap2 (the crate behind /root/reference/src/cleaner.rs:552, presets :457-458,465) keeps the look-back window of its long join in two
// balanced trees keyed by (y, i) whose nodes carry a pointer to the minimum-priority node of their subtree (krmq.h).  lr_rmq_fill asks the
// same questions of a scan; its answers are the trees' as long as the smallest priority in a query interval is held by ONE candidate.  When
// two candidates tie, krmq_rmq returns the one the shape of the tree and the history of its rotations favour (krmq_rotate1/2 hand the old
// root's subtree-minimum pointer to the new root instead of recomputing it), and the predecessor an anchor gets decides the chains.  Such
// reads (0.3 % of the bench's long reads), reads whose inner window or same-position group outgrows the LDS ring, and reads with more anchors
// than rmq_size_cap (the cap evicts from the tree out of order) are redone here: insert / erase / rotate / rmq / iterate as upstream
// states them, on an index-based node pool in the wave's HBM scratch, by ONE lane - this is the serial path, its cost is irrelevant.
// The CPU oracle holds the same restatement (oracle/mm_rmq.c, checked there against a brute-force scan); the two share no code.
#pragma once
#include <stdint.h>

#define RQ_MAX_DEPTH 64
#define RQ_NIL (-1)

struct RqNode {
    int32_t y, i; double pri;
    int32_t p[2], s;           // children, subtree minimum (node indices)
    int32_t balance; uint32_t size;
};
struct RqTree { RqNode *n; int32_t cap, n_used, free_head, root; };

__device__ inline void rq_init(RqTree &t, RqNode *pool, int32_t cap) { t.n = pool; t.cap = cap; t.n_used = 0; t.free_head = RQ_NIL; t.root = RQ_NIL; }
__device__ inline int32_t rq_alloc(RqTree &t)
{
    if (t.free_head != RQ_NIL) { const int32_t k = t.free_head; t.free_head = t.n[k].p[0]; return k; }
    if (t.n_used >= t.cap) return RQ_NIL;      // cannot happen: cap = window + 2
    return t.n_used++;
}
__device__ inline void rq_free(RqTree &t, int32_t k) { t.n[k].p[0] = t.free_head; t.free_head = k; }

#define RQN(k) (t.n[k])
__device__ inline int rq_cmp_key(int32_t ay, int32_t ai, const RqNode &b)
{   // lc_elem_cmp
    return ay < b.y ? -1 : ay > b.y ? 1 : (ai > b.i) - (ai < b.i);
}
__device__ inline bool rq_lt2(const RqTree &t, int32_t a, int32_t b) { return RQN(a).pri < RQN(b).pri; }
__device__ inline uint32_t rq_size_child(const RqTree &t, int32_t q, int i) { return RQN(q).p[i] != RQ_NIL ? RQN(RQN(q).p[i]).size : 0u; }

// krmq_update_min(p, q, r): p's subtree minimum from p itself and the minima of the two given subtrees, in that order
__device__ inline void rq_update_min(RqTree &t, int32_t p, int32_t q, int32_t r)
{
    RQN(p).s = (q == RQ_NIL || rq_lt2(t, p, RQN(q).s)) ? p : RQN(q).s;
    RQN(p).s = (r == RQ_NIL || rq_lt2(t, RQN(p).s, RQN(r).s)) ? RQN(p).s : RQN(r).s;
}

// one rotation: (a,(b,c)q)p => ((a,b)p,c)q
__device__ inline int32_t rq_rotate1(RqTree &t, int32_t p, int dir)
{
    const int opp = 1 - dir;
    const int32_t q = RQN(p).p[opp], s = RQN(p).s;
    const uint32_t size_p = RQN(p).size;
    RQN(p).size -= RQN(q).size - rq_size_child(t, q, dir);
    RQN(q).size = size_p;
    rq_update_min(t, p, RQN(p).p[dir], RQN(q).p[dir]);
    RQN(q).s = s;
    RQN(p).p[opp] = RQN(q).p[dir];
    RQN(q).p[dir] = p;
    return q;
}

// two consecutive rotations: (a,((b,c)r,d)q)p => ((a,b)p,(c,d)q)r
__device__ inline int32_t rq_rotate2(RqTree &t, int32_t p, int dir)
{
    const int opp = 1 - dir;
    const int32_t q = RQN(p).p[opp], r = RQN(q).p[dir], s = RQN(p).s;
    const uint32_t size_x_dir = rq_size_child(t, r, dir);
    RQN(r).size = RQN(p).size;
    RQN(p).size -= RQN(q).size - size_x_dir;
    RQN(q).size -= size_x_dir + 1;
    rq_update_min(t, p, RQN(p).p[dir], RQN(r).p[dir]);
    rq_update_min(t, q, RQN(q).p[opp], RQN(r).p[opp]);
    RQN(r).s = s;
    RQN(p).p[opp] = RQN(r).p[dir];
    RQN(r).p[dir] = p;
    RQN(q).p[dir] = RQN(r).p[opp];
    RQN(r).p[opp] = q;
    const int b1 = dir == 0 ? +1 : -1;
    if (RQN(r).balance == b1) { RQN(q).balance = 0; RQN(p).balance = -b1; }
    else if (RQN(r).balance == 0) RQN(q).balance = RQN(p).balance = 0;
    else { RQN(q).balance = b1; RQN(p).balance = 0; }
    RQN(r).balance = 0;
    return r;
}

__device__ inline void rq_insert(RqTree &t, int32_t x)
{
    unsigned char stack[RQ_MAX_DEPTH];
    int32_t path[RQ_MAX_DEPTH];
    int32_t bp, bq, p, q, r;
    int i, which = 0, top, path_len;
    bp = t.root; bq = RQ_NIL;
    for (p = bp, q = bq, top = path_len = 0; p != RQ_NIL; q = p, p = RQN(p).p[which]) {
        const int cmp = rq_cmp_key(RQN(x).y, RQN(x).i, RQN(p));
        if (cmp == 0) return;     // (y, i) is unique: never taken
        if (RQN(p).balance != 0) { bq = q; bp = p; top = 0; }
        stack[top++] = (unsigned char)(which = (cmp > 0));
        path[path_len++] = p;
    }
    RQN(x).balance = 0; RQN(x).size = 1; RQN(x).p[0] = RQN(x).p[1] = RQ_NIL; RQN(x).s = x;
    if (q == RQ_NIL) t.root = x;
    else RQN(q).p[which] = x;
    if (bp == RQ_NIL) return;
    for (i = 0; i < path_len; ++i) ++RQN(path[i]).size;
    for (i = path_len - 1; i >= 0; --i) {
        rq_update_min(t, path[i], RQN(path[i]).p[0], RQN(path[i]).p[1]);
        if (RQN(path[i]).s != x) break;
    }
    for (p = bp, top = 0; p != x; p = RQN(p).p[stack[top]], ++top) {
        if (stack[top] == 0) --RQN(p).balance;
        else ++RQN(p).balance;
    }
    if (RQN(bp).balance > -2 && RQN(bp).balance < 2) return;
    which = (RQN(bp).balance < 0);
    const int b1 = which == 0 ? +1 : -1;
    q = RQN(bp).p[1 - which];
    if (RQN(q).balance == b1) {
        r = rq_rotate1(t, bp, which);
        RQN(q).balance = RQN(bp).balance = 0;
    } else r = rq_rotate2(t, bp, which);
    if (bq == RQ_NIL) t.root = r;
    else RQN(bq).p[bp != RQN(bq).p[0]] = r;
}

// krmq_erase of the node with key (y, i); returns its index or RQ_NIL.  path[0] stands for upstream's `fake` node.
__device__ inline int32_t rq_erase(RqTree &t, int32_t ky, int32_t ki)
{
    int32_t p, path[RQ_MAX_DEPTH], fake;
    unsigned char dir[RQ_MAX_DEPTH];
    int i, d = 0, cmp;
    if (t.root == RQ_NIL) return RQ_NIL;
    fake = rq_alloc(t);
    RQN(fake) = RQN(t.root);       // fake = **root_
    RQN(fake).p[0] = t.root; RQN(fake).p[1] = RQ_NIL;
    for (cmp = -1, p = fake; cmp; cmp = rq_cmp_key(ky, ki, RQN(p))) {
        const int which = (cmp > 0);
        dir[d] = (unsigned char)which;
        path[d++] = p;
        p = RQN(p).p[which];
        if (p == RQ_NIL) { rq_free(t, fake); return RQ_NIL; }
    }
    for (i = 1; i < d; ++i) --RQN(path[i]).size;
    if (RQN(p).p[1] == RQ_NIL) {
        RQN(path[d - 1]).p[dir[d - 1]] = RQN(p).p[0];
    } else {
        int32_t q = RQN(p).p[1];
        if (RQN(q).p[0] == RQ_NIL) {
            RQN(q).p[0] = RQN(p).p[0];
            RQN(q).balance = RQN(p).balance;
            RQN(path[d - 1]).p[dir[d - 1]] = q;
            path[d] = q; dir[d++] = 1;
            RQN(q).size = RQN(p).size - 1;
        } else {
            int32_t r;
            const int e = d++;
            for (;;) {
                dir[d] = 0;
                path[d++] = q;
                r = RQN(q).p[0];
                if (RQN(r).p[0] == RQ_NIL) break;
                q = r;
            }
            RQN(r).p[0] = RQN(p).p[0];
            RQN(q).p[0] = RQN(r).p[1];
            RQN(r).p[1] = RQN(p).p[1];
            RQN(r).balance = RQN(p).balance;
            RQN(path[e - 1]).p[dir[e - 1]] = r;
            path[e] = r; dir[e] = 1;
            for (i = e + 1; i < d; ++i) --RQN(path[i]).size;
            RQN(r).size = RQN(p).size - 1;
        }
    }
    for (i = d - 1; i >= 0; --i) rq_update_min(t, path[i], RQN(path[i]).p[0], RQN(path[i]).p[1]);
    while (--d > 0) {
        const int32_t q = path[d];
        int which, other, b1 = 1, b2 = 2;
        which = dir[d]; other = 1 - which;
        if (which) { b1 = -b1; b2 = -b2; }
        RQN(q).balance += b1;
        if (RQN(q).balance == b1) break;
        else if (RQN(q).balance == b2) {
            const int32_t r = RQN(q).p[other];
            if (RQN(r).balance == -b1) {
                RQN(path[d - 1]).p[dir[d - 1]] = rq_rotate2(t, q, which);
            } else {
                RQN(path[d - 1]).p[dir[d - 1]] = rq_rotate1(t, q, which);
                if (RQN(r).balance == 0) {
                    RQN(r).balance = -b1;
                    RQN(q).balance = b1;
                    break;
                } else RQN(r).balance = RQN(q).balance = 0;
            }
        }
    }
    t.root = RQN(fake).p[0];
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
        cmp = rq_cmp_key(lo_y, lo_i, RQN(p));
        path[0][plen[0]] = p; pcmp[0][plen[0]++] = cmp;
        if (cmp < 0) p = RQN(p).p[0];
        else if (cmp > 0) p = RQN(p).p[1];
        else break;
    }
    p = t.root;
    while (p != RQ_NIL) {
        cmp = rq_cmp_key(hi_y, hi_i, RQN(p));
        path[1][plen[1]] = p; pcmp[1][plen[1]++] = cmp;
        if (cmp < 0) p = RQN(p).p[0];
        else if (cmp > 0) p = RQN(p).p[1];
        else break;
    }
    for (i = 0; i < plen[0] && i < plen[1]; ++i)
        if (path[0][i] == path[1][i] && pcmp[0][i] <= 0 && pcmp[1][i] >= 0) break;
    if (i == plen[0] || i == plen[1]) return RQ_NIL;
    lca = i; min = path[0][lca];
    for (i = lca + 1; i < plen[0]; ++i) {
        if (pcmp[0][i] <= 0) {
            if (rq_lt2(t, path[0][i], min)) min = path[0][i];
            if (RQN(path[0][i]).p[1] != RQ_NIL && rq_lt2(t, RQN(RQN(path[0][i]).p[1]).s, min)) min = RQN(RQN(path[0][i]).p[1]).s;
        }
    }
    for (i = lca + 1; i < plen[1]; ++i) {
        if (pcmp[1][i] >= 0) {
            if (rq_lt2(t, path[1][i], min)) min = path[1][i];
            if (RQN(path[1][i]).p[0] != RQ_NIL && rq_lt2(t, RQN(RQN(path[1][i]).p[0]).s, min)) min = RQN(RQN(path[1][i]).p[0]).s;
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
        const int cmp = rq_cmp_key(ky, ki, RQN(p));
        it.stack[d++] = p;
        if (cmp < 0) p = RQN(p).p[0];
        else if (cmp > 0) { best = d; p = RQN(p).p[1]; }
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
    p = RQN(it.stack[it.top]).p[0];
    if (p != RQ_NIL) {
        for (; p != RQ_NIL; p = RQN(p).p[1]) it.stack[++it.top] = p;
        return true;
    }
    do { p = it.stack[it.top--]; } while (it.top >= 0 && p == RQN(it.stack[it.top]).p[0]);
    return it.top >= 0;
}
#undef RQN
