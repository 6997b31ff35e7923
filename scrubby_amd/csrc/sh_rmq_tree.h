// sh_rmq_tree.h — mg_lchain_rmq's balanced tree, for the reads the wave scan of sh_long.h cannot answer bit for bit.
//
// minimap2 (the crate behind /root/reference/src/cleaner.rs:552, presets :457-458,465) keeps the look-back window of its long join in two
// balanced trees keyed by (y, i) whose nodes carry a pointer to the minimum-priority node of their subtree (krmq.h).  lr_rmq_fill asks the
// same questions of a scan; its answers are the trees' as long as the smallest priority in a query interval is held by ONE candidate.  When
// two candidates tie, krmq_rmq returns the one the shape of the tree and the history of its rotations favour (krmq_rotate1/2 hand the old
// root's subtree-minimum pointer to the new root instead of recomputing it), and the predecessor an anchor gets decides the chains.  Reads
// that meet such a tie keep the scan for everything that has width and ask a literal tree - insert / erase / rotate / rmq as upstream states
// them, on an index-based node pool, maintained by ONE lane - at the ties only (lr_rmq_fill<NR, true>); reads the LDS ring cannot hold run
// both trees on one lane (lr_rmq_fill_tree).
//
// Where the nodes live is a storage policy (round 5).  The tree of a read holds the anchors of its look-back window - max_gap reference
// bases: at most ~1 400 anchors on the bench's satellite reads, whatever the read's size (measured) - so the scan's tree
// lives entirely in LDS (RqLds: 32 B a node - a 16-byte piece with the key, the 16-bit links and the balance, which one ds_read_b128 brings to a
// descent step, the priority, and the priority of the subtree minimum kept beside its pointer - every access a ds_read / ds_write through
// address-space-3 pointers: no generic pointer into LDS is ever formed; the walks' path arrays are in LDS too).  RqPool keeps 32-B nodes in the wave's HBM scratch, for windows LDS cannot hold
// and for the one-lane version with both trees.  Upstream's per-node subtree SIZE is not kept: nothing but krmq_size(root) reads it, and
// that is the number of live nodes (n_live); the shape, the subtree-minimum pointers and the answers do not depend on it.
// The CPU oracle holds the same restatement (oracle/mm_rmq.c, checked there against a brute-force scan); the two share no code.
#pragma once
#include <stdint.h>

#define RQ_MAX_DEPTH 64
#define RQ_NIL (-1)
#if defined(__HIP_DEVICE_COMPILE__)
#define RQ_LDS __attribute__((address_space(3)))
#else
#define RQ_LDS
#endif

// ---- storage: a pool of 32-byte nodes in global memory ---------------------------------------------------------------------
struct alignas(16) RqNode {      // 32 bytes: two 128-bit pieces
    int32_t y, i; double pri;
    int32_t p[2], s;           // children, subtree minimum (node indices)
    int32_t balance;
};
struct RqPool {
    RqNode *n; int32_t cap, n_used, free_head;
    // the walks' path arrays (upstream's on-stack arrays): private memory here
    static constexpr int max_depth = RQ_MAX_DEPTH;
    int32_t sp_[2][RQ_MAX_DEPTH]; int8_t sd_[2][RQ_MAX_DEPTH];
    __device__ inline int32_t sp(int w, int i) const { return sp_[w][i]; }
    __device__ inline void set_sp(int w, int i, int32_t v) { sp_[w][i] = v; }
    __device__ inline int sd(int w, int i) const { return sd_[w][i]; }
    __device__ inline void set_sd(int w, int i, int v) { sd_[w][i] = (int8_t)v; }
    __device__ inline void init(RqNode *pool, int32_t c) { n = pool; cap = c; n_used = 0; free_head = RQ_NIL; }
    __device__ inline bool valid(int32_t k) const { return (uint32_t)k < (uint32_t)cap; }
    __device__ inline int32_t alloc()
    {
        if (free_head != RQ_NIL) { const int32_t k = free_head; free_head = n[k].p[0]; return k; }
        if (n_used >= cap) return RQ_NIL;
        return n_used++;
    }
    __device__ inline void release(int32_t k) { n[k].p[0] = free_head; free_head = k; }
    __device__ inline int32_t y(int32_t k) const { return n[k].y; }
    __device__ inline int32_t i(int32_t k) const { return n[k].i; }
    __device__ inline double pri(int32_t k) const { return n[k].pri; }
    __device__ inline int32_t ch(int32_t k, int d) const { return n[k].p[d]; }
    __device__ inline int32_t s(int32_t k) const { return n[k].s; }
    __device__ inline int32_t bal(int32_t k) const { return n[k].balance; }
    __device__ inline void set_ch(int32_t k, int d, int32_t v) { n[k].p[d] = v; }
    __device__ inline void set_s(int32_t k, int32_t v, double) { n[k].s = v; }
    __device__ inline double spri(int32_t k) const { return n[n[k].s].pri; }      // priority of the subtree minimum
    __device__ inline void hot(int32_t k, int32_t &yy, int32_t &ii, int32_t &l, int32_t &r, int32_t &b) const { const RqNode z = n[k]; yy = z.y; ii = z.i; l = z.p[0]; r = z.p[1]; b = z.balance; }
    __device__ inline void set_bal(int32_t k, int32_t v) { n[k].balance = v; }
    __device__ inline void fresh(int32_t k, int32_t yy, int32_t ii, double pp) { RqNode z; z.y = yy; z.i = ii; z.pri = pp; z.p[0] = z.p[1] = RQ_NIL; z.s = k; z.balance = 0; n[k] = z; }
    __device__ inline void fake(int32_t k, int32_t root) { RqNode z = n[root]; z.p[0] = root; z.p[1] = RQ_NIL; n[k] = z; }      // krmq_erase's `fake = **root_` with the tree below its left link
};

// ---- storage: the whole tree in LDS -------------------------------------------------------------------------------------------
// The memory (a __shared__ object of the kernel; CAP <= 65535 nodes, 32 B each) and the handle the tree code works through.
#define RQ_LDS_DEPTH 40      // an AVL tree that deep holds more than 10^8 nodes
struct alignas(16) RqHot { int32_t y, i; uint16_t l, r, s; int8_t bal, pad; };      // what a descent step reads, in one 128-bit piece
template <int CAP>
struct RqLdsMem { RqHot hot[CAP]; double pri[CAP], spri[CAP]; int32_t sp[2][RQ_LDS_DEPTH]; int8_t sd[2][RQ_LDS_DEPTH]; };
struct RqLds {
    RQ_LDS RqHot *h_; RQ_LDS double *pri_, *spri_;
    RQ_LDS int32_t *sp_; RQ_LDS int8_t *sd_;      // the walks' path arrays: in LDS too (in private memory every access is a trip to the L1 / L2: it was most of an operation)
    int32_t cap, n_used, free_head;
    static constexpr int max_depth = RQ_LDS_DEPTH;
    __device__ inline int32_t sp(int w, int i) const { return sp_[w * RQ_LDS_DEPTH + i]; }
    __device__ inline void set_sp(int w, int i, int32_t v) { sp_[w * RQ_LDS_DEPTH + i] = v; }
    __device__ inline int sd(int w, int i) const { return (int)sd_[w * RQ_LDS_DEPTH + i]; }
    __device__ inline void set_sd(int w, int i, int v) { sd_[w * RQ_LDS_DEPTH + i] = (int8_t)v; }
    template <int CAP>
    __device__ inline void init(RqLdsMem<CAP> &m)
    {
        // (an LDS address is the low half of the generic one: the casts below go through the integer, as the compiler asks)
        h_ = (RQ_LDS RqHot *)(uintptr_t)m.hot; pri_ = (RQ_LDS double *)(uintptr_t)m.pri; spri_ = (RQ_LDS double *)(uintptr_t)m.spri;
        sp_ = (RQ_LDS int32_t *)(uintptr_t)&m.sp[0][0]; sd_ = (RQ_LDS int8_t *)(uintptr_t)&m.sd[0][0];
        cap = CAP; n_used = 0; free_head = RQ_NIL;
    }
    __device__ static inline int32_t up(uint16_t v) { return v == 0xffffu ? RQ_NIL : (int32_t)v; }
    __device__ inline bool valid(int32_t k) const { return (uint32_t)k < (uint32_t)cap; }
    __device__ inline int32_t alloc()
    {
        if (free_head != RQ_NIL) { const int32_t k = free_head; free_head = up(h_[k].l); return k; }
        if (n_used >= cap) return RQ_NIL;
        return n_used++;
    }
    __device__ inline void release(int32_t k) { h_[k].l = (uint16_t)free_head; free_head = k; }
    __device__ inline int32_t y(int32_t k) const { return h_[k].y; }
    __device__ inline int32_t i(int32_t k) const { return h_[k].i; }
    __device__ inline double pri(int32_t k) const { return pri_[k]; }
    __device__ inline double spri(int32_t k) const { return spri_[k]; }
    __device__ inline int32_t ch(int32_t k, int d) const { return up(d ? h_[k].r : h_[k].l); }
    __device__ inline int32_t s(int32_t k) const { return up(h_[k].s); }
    __device__ inline int32_t bal(int32_t k) const { return (int32_t)h_[k].bal; }
    __device__ inline void hot(int32_t k, int32_t &yy, int32_t &ii, int32_t &l, int32_t &r, int32_t &b) const
    {
        const RqHot z = h_[k];      // one ds_read_b128
        yy = z.y; ii = z.i; l = up(z.l); r = up(z.r); b = (int32_t)z.bal;
    }
#if defined(__HIP_DEVICE_COMPILE__)
    // the same for a node EVERY lane asks for (the wave forms below): one broadcast ds_read_b128, and the fields made scalar, so that the
    // walk's control flow stays on the scalar unit (values that come out of a VGPR make every branch an exec-mask affair)
    __device__ inline void hot_u(int32_t k, int32_t &yy, int32_t &ii, int32_t &l, int32_t &r, int32_t &b) const
    {
        const uint4 z = *(RQ_LDS const uint4 *)(h_ + k);
        const uint32_t w2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)z.z), w3 = (uint32_t)__builtin_amdgcn_readfirstlane((int)z.w);
        yy = __builtin_amdgcn_readfirstlane((int)z.x); ii = __builtin_amdgcn_readfirstlane((int)z.y);
        l = up((uint16_t)(w2 & 0xffffu)); r = up((uint16_t)(w2 >> 16)); b = (int32_t)(int8_t)(w3 >> 16 & 0xffu);
    }
    __device__ inline int32_t ch_u(int32_t k, int d) const { return __builtin_amdgcn_readfirstlane(ch(k, d)); }
#else      // (the host pass only parses the wave forms)
    __device__ inline void hot_u(int32_t k, int32_t &yy, int32_t &ii, int32_t &l, int32_t &r, int32_t &b) const { hot(k, yy, ii, l, r, b); }
    __device__ inline int32_t ch_u(int32_t k, int d) const { return ch(k, d); }
#endif
    __device__ inline void set_ch(int32_t k, int d, int32_t v) { if (d) h_[k].r = (uint16_t)v; else h_[k].l = (uint16_t)v; }
    __device__ inline void set_s(int32_t k, int32_t v, double pv) { h_[k].s = (uint16_t)v; spri_[k] = pv; }
    __device__ inline void set_bal(int32_t k, int32_t v) { h_[k].bal = (int8_t)v; }
    __device__ inline void fresh(int32_t k, int32_t yy, int32_t ii, double pp) { RqHot z; z.y = yy; z.i = ii; z.l = 0xffffu; z.r = 0xffffu; z.s = (uint16_t)k; z.bal = 0; z.pad = 0; h_[k] = z; pri_[k] = pp; spri_[k] = pp; }
    __device__ inline void fake(int32_t k, int32_t root) { RqHot z = h_[root]; z.l = (uint16_t)root; z.r = 0xffffu; h_[k] = z; pri_[k] = pri_[root]; spri_[k] = spri_[root]; }
};

// ---- the tree ------------------------------------------------------------------------------------------------------------------------
// bad: a walk left the pool or grew deeper than RQ_MAX_DEPTH (the caller gives the read up instead of touching memory it does not own)
template <class ST>
struct RqTreeT { ST st; int32_t root, bad, n_live; };
typedef RqTreeT<RqPool> RqTree;

template <class ST> __device__ inline void rq_reset(RqTreeT<ST> &t) { t.root = RQ_NIL; t.bad = 0; t.n_live = 0; }
__device__ inline void rq_init(RqTree &t, RqNode *pool, int32_t cap) { t.st.init(pool, cap); rq_reset(t); }
// every index a walk follows goes through here: one outside the pool marks the tree bad and reads node 0 instead
template <class ST> __device__ inline int32_t rq_ok(const RqTreeT<ST> &t, int32_t k)
{
    if (!t.st.valid(k)) { if (!t.bad) const_cast<RqTreeT<ST> &>(t).bad = 1; return 0; }
    return k;
}
template <class ST> __device__ inline int32_t rq_alloc(RqTreeT<ST> &t) { return t.st.alloc(); }
template <class ST> __device__ inline void rq_free(RqTreeT<ST> &t, int32_t k) { if (t.st.valid(k)) t.st.release(k); }
// a fresh node, not yet in the tree
template <class ST> __device__ inline void rq_node_set(RqTreeT<ST> &t, int32_t k, int32_t y, int32_t i, double pri)
{
    if (!t.st.valid(k)) { if (!t.bad) t.bad = 2; return; }
    t.st.fresh(k, y, i, pri);
}
template <class ST> __device__ inline int32_t rq_i(const RqTreeT<ST> &t, int32_t k) { return t.st.i(rq_ok(t, k)); }
template <class ST> __device__ inline int32_t rq_y(const RqTreeT<ST> &t, int32_t k) { return t.st.y(rq_ok(t, k)); }
template <class ST> __device__ inline int32_t rq_size(const RqTreeT<ST> &t) { return t.n_live; }      // krmq_size(root)

__device__ inline int rq_cmp_key(int32_t ay, int32_t ai, int32_t by, int32_t bi)
{   // lc_elem_cmp
    return ay < by ? -1 : ay > by ? 1 : (ai > bi) - (ai < bi);
}
#define RQ_Y(k) t.st.y(rq_ok(t, (k)))
#define RQ_I(k) t.st.i(rq_ok(t, (k)))
#define RQ_PRI(k) t.st.pri(rq_ok(t, (k)))
#define RQ_CH(k, d) t.st.ch(rq_ok(t, (k)), (d))
#define RQ_S(k) t.st.s(rq_ok(t, (k)))
#define RQ_BAL(k) t.st.bal(rq_ok(t, (k)))
#define RQ_SET_CH(k, d, v) t.st.set_ch(rq_ok(t, (k)), (d), (v))
#define RQ_SET_S(k, v, pv) t.st.set_s(rq_ok(t, (k)), (v), (pv))
#define RQ_SPRI(k) t.st.spri(rq_ok(t, (k)))
#define RQ_SET_BAL(k, v) t.st.set_bal(rq_ok(t, (k)), (v))
template <class ST> __device__ inline int rq_cmp_node(const RqTreeT<ST> &t, int32_t ay, int32_t ai, int32_t k) { const int32_t kk = rq_ok(t, k); return rq_cmp_key(ay, ai, t.st.y(kk), t.st.i(kk)); }
template <class ST> __device__ inline bool rq_lt2(const RqTreeT<ST> &t, int32_t a, int32_t b) { const double pa = RQ_PRI(a); return pa < RQ_PRI(b); }

// krmq_update_min(p, q, r): p's subtree minimum from p itself and the minima of the two given subtrees, in that order
template <class ST> __device__ inline void rq_update_min(RqTreeT<ST> &t, int32_t p, int32_t q, int32_t r)
{
    // (the subtrees' minima come with their priorities: the loads below do not depend on one another)
    const double pp = RQ_PRI(p);
    const int32_t qs = q != RQ_NIL ? RQ_S(q) : RQ_NIL, rs = r != RQ_NIL ? RQ_S(r) : RQ_NIL;
    const double qp = q != RQ_NIL ? RQ_SPRI(q) : 0.0, rp = r != RQ_NIL ? RQ_SPRI(r) : 0.0;
    int32_t s = p; double sp = pp;
    if (q != RQ_NIL && !(pp < qp)) { s = qs; sp = qp; }
    if (r != RQ_NIL && !(sp < rp)) { s = rs; sp = rp; }
    RQ_SET_S(p, s, sp);
}

// one rotation: (a,(b,c)q)p => ((a,b)p,c)q
template <class ST> __device__ inline int32_t rq_rotate1(RqTreeT<ST> &t, int32_t p, int dir)
{
    const int opp = 1 - dir;
    const int32_t q = RQ_CH(p, opp), s = RQ_S(p);
    const double s_pri = RQ_SPRI(p);
    rq_update_min(t, p, RQ_CH(p, dir), RQ_CH(q, dir));
    RQ_SET_S(q, s, s_pri);
    RQ_SET_CH(p, opp, RQ_CH(q, dir));
    RQ_SET_CH(q, dir, p);
    return q;
}

// two consecutive rotations: (a,((b,c)r,d)q)p => ((a,b)p,(c,d)q)r
template <class ST> __device__ inline int32_t rq_rotate2(RqTreeT<ST> &t, int32_t p, int dir)
{
    const int opp = 1 - dir;
    const int32_t q = RQ_CH(p, opp), r = RQ_CH(q, dir), s = RQ_S(p);
    const double s_pri = RQ_SPRI(p);
    rq_update_min(t, p, RQ_CH(p, dir), RQ_CH(r, dir));
    rq_update_min(t, q, RQ_CH(q, opp), RQ_CH(r, opp));
    RQ_SET_S(r, s, s_pri);
    RQ_SET_CH(p, opp, RQ_CH(r, dir));
    RQ_SET_CH(r, dir, p);
    RQ_SET_CH(q, dir, RQ_CH(r, opp));
    RQ_SET_CH(r, opp, q);
    const int b1 = dir == 0 ? +1 : -1;
    const int32_t rb = RQ_BAL(r);
    if (rb == b1) { RQ_SET_BAL(q, 0); RQ_SET_BAL(p, -b1); }
    else if (rb == 0) { RQ_SET_BAL(q, 0); RQ_SET_BAL(p, 0); }
    else { RQ_SET_BAL(q, b1); RQ_SET_BAL(p, 0); }
    RQ_SET_BAL(r, 0);
    return r;
}

// x: a node prepared by rq_node_set
template <class ST> __device__ inline void rq_insert(RqTreeT<ST> &t, int32_t x)
{
    // stack[] = t.st.sd(0, .), path[] = t.st.sp(0, .)
    int32_t bp, bq, p, q, r;
    int i, which = 0, top, path_len;
    const int32_t xy = RQ_Y(x), xi = RQ_I(x);
    bp = t.root; bq = RQ_NIL;
    for (p = bp, q = bq, top = path_len = 0; p != RQ_NIL;) {
        int32_t ny, ni, nl, nr, nb;
        t.st.hot(rq_ok(t, p), ny, ni, nl, nr, nb);      // key, children and balance of p in one piece
        const int cmp = rq_cmp_key(xy, xi, ny, ni);
        if (cmp == 0) return;     // (y, i) is unique: never taken
        if (nb != 0) { bq = q; bp = p; top = 0; }
        which = (cmp > 0);
        t.st.set_sd(0, top++, which);
        t.st.set_sp(0, path_len++, p);
        if (path_len >= ST::max_depth - 1) { if (!t.bad) t.bad = 4; return; }
        q = p; p = which ? nr : nl;
    }
    ++t.n_live;
    if (q == RQ_NIL) t.root = x;
    else RQ_SET_CH(q, which, x);
    if (bp == RQ_NIL) return;
    for (i = path_len - 1; i >= 0; --i) {
        const int32_t pi = t.st.sp(0, i);
        rq_update_min(t, pi, RQ_CH(pi, 0), RQ_CH(pi, 1));
        if (RQ_S(pi) != x) break;
    }
    for (p = bp, top = 0; p != x; ++top) {
        const int dd = t.st.sd(0, top);
        if (dd == 0) RQ_SET_BAL(p, RQ_BAL(p) - 1);
        else RQ_SET_BAL(p, RQ_BAL(p) + 1);
        p = RQ_CH(p, dd);
    }
    const int32_t bb = RQ_BAL(bp);
    if (bb > -2 && bb < 2) return;
    which = (bb < 0);
    const int b1 = which == 0 ? +1 : -1;
    q = RQ_CH(bp, 1 - which);
    if (RQ_BAL(q) == b1) {
        r = rq_rotate1(t, bp, which);
        RQ_SET_BAL(q, 0); RQ_SET_BAL(bp, 0);
    } else r = rq_rotate2(t, bp, which);
    if (bq == RQ_NIL) t.root = r;
    else { const int wi = bp != RQ_CH(bq, 0); RQ_SET_CH(bq, wi, r); }
}

// krmq_erase's way back up: path / dir in the storage's walk arrays (entries 0 .. d - 1, entry 0 = the fake node)
template <class ST> __device__ inline void rq_erase_rebalance(RqTreeT<ST> &t, int d)
{
#define PATH(i) t.st.sp(0, (i))
#define DIR(i) t.st.sd(0, (i))
    while (--d > 0) {
        const int32_t q = PATH(d);
        int which, other, b1 = 1, b2 = 2;
        which = DIR(d); other = 1 - which;
        if (which) { b1 = -b1; b2 = -b2; }
        const int32_t qb = RQ_BAL(q) + b1;
        RQ_SET_BAL(q, qb);
        if (qb == b1) break;
        else if (qb == b2) {
            const int32_t r = RQ_CH(q, other);
            const int32_t rbal = RQ_BAL(r);
            if (rbal == -b1) {
                const int32_t nr = rq_rotate2(t, q, which);
                RQ_SET_CH(PATH(d - 1), DIR(d - 1), nr);
            } else {
                const int32_t nr = rq_rotate1(t, q, which);
                RQ_SET_CH(PATH(d - 1), DIR(d - 1), nr);
                if (rbal == 0) {
                    RQ_SET_BAL(r, -b1);
                    RQ_SET_BAL(q, b1);
                    break;
                } else { RQ_SET_BAL(r, 0); RQ_SET_BAL(q, 0); }
            }
        }
    }
#undef PATH
#undef DIR
}

// krmq_erase of the node with key (y, i); returns its index or RQ_NIL.  path[0] stands for upstream's `fake` node.
template <class ST> __device__ inline int32_t rq_erase(RqTreeT<ST> &t, int32_t ky, int32_t ki)
{
    // path[] = t.st.sp(0, .), dir[] = t.st.sd(0, .)
#define PATH(i) t.st.sp(0, (i))
#define DIR(i) t.st.sd(0, (i))
    int32_t p, fake;
    int i, d = 0, cmp;
    if (t.root == RQ_NIL) return RQ_NIL;
    fake = rq_alloc(t);
    if (fake == RQ_NIL) { if (!t.bad) t.bad = 3; return RQ_NIL; }
    t.st.fake(fake, rq_ok(t, t.root));
    int32_t p_l = t.root, p_r = RQ_NIL, p_bal = 0;      // (children and balance of the node the walk stands on; fake's left link is the root)
    for (cmp = -1, p = fake; cmp;) {
        const int which = (cmp > 0);
        t.st.set_sd(0, d, which);
        t.st.set_sp(0, d++, p);
        if (d >= ST::max_depth - 2) { if (!t.bad) t.bad = 5; rq_free(t, fake); return RQ_NIL; }
        p = which ? p_r : p_l;
        if (p == RQ_NIL) { rq_free(t, fake); return RQ_NIL; }
        int32_t ny, ni;
        t.st.hot(rq_ok(t, p), ny, ni, p_l, p_r, p_bal);
        cmp = rq_cmp_key(ky, ki, ny, ni);
    }
    --t.n_live;
    if (p_r == RQ_NIL) {
        RQ_SET_CH(PATH(d - 1), DIR(d - 1), p_l);
    } else {
        int32_t q = p_r;
        if (RQ_CH(q, 0) == RQ_NIL) {
            RQ_SET_CH(q, 0, p_l);
            RQ_SET_BAL(q, p_bal);
            RQ_SET_CH(PATH(d - 1), DIR(d - 1), q);
            t.st.set_sp(0, d, q); t.st.set_sd(0, d++, 1);
        } else {
            int32_t r;
            const int e = d++;
            for (;;) {
                t.st.set_sd(0, d, 0);
                t.st.set_sp(0, d++, q);
                if (d >= ST::max_depth - 1) { if (!t.bad) t.bad = 6; rq_free(t, fake); return RQ_NIL; }
                r = RQ_CH(q, 0);
                if (RQ_CH(r, 0) == RQ_NIL) break;
                q = r;
            }
            RQ_SET_CH(r, 0, p_l);
            RQ_SET_CH(q, 0, RQ_CH(r, 1));
            RQ_SET_CH(r, 1, p_r);
            RQ_SET_BAL(r, p_bal);
            RQ_SET_CH(PATH(e - 1), DIR(e - 1), r);
            t.st.set_sp(0, e, r); t.st.set_sd(0, e, 1);
        }
    }
    for (i = d - 1; i >= 0; --i) { const int32_t pi = PATH(i); rq_update_min(t, pi, RQ_CH(pi, 0), RQ_CH(pi, 1)); }
    rq_erase_rebalance(t, d);
    t.root = RQ_CH(fake, 0);
    rq_free(t, fake);
    return p;
#undef PATH
#undef DIR
}

// krmq_rmq over the CLOSED key interval [(lo_y, lo_i), (hi_y, hi_i)]
template <class ST> __device__ inline int32_t rq_rmq(const RqTreeT<ST> &tc, int32_t lo_y, int32_t lo_i, int32_t hi_y, int32_t hi_i)
{
    RqTreeT<ST> &t = const_cast<RqTreeT<ST> &>(tc);      // (the walks' path arrays belong to the storage)
    // path[w][] = t.st.sp(w, .), pcmp[w][] = t.st.sd(w, .)
    int32_t p = t.root, min;
    int plen[2] = {0, 0}, i, cmp, lca;
    if (t.root == RQ_NIL) return RQ_NIL;
    for (int w = 0; w < 2; ++w) {
        const int32_t key_y = w ? hi_y : lo_y, key_i = w ? hi_i : lo_i;
        p = t.root;
        while (p != RQ_NIL) {
            int32_t ny, ni, nl, nr, nb;
            t.st.hot(rq_ok(t, p), ny, ni, nl, nr, nb);
            cmp = rq_cmp_key(key_y, key_i, ny, ni);
            if (plen[w] >= ST::max_depth - 1) { if (!t.bad) t.bad = 7 + w; return RQ_NIL; }
            t.st.set_sp(w, plen[w], p); t.st.set_sd(w, plen[w]++, cmp);
            if (cmp < 0) p = nl;
            else if (cmp > 0) p = nr;
            else break;
        }
    }
    for (i = 0; i < plen[0] && i < plen[1]; ++i)
        if (t.st.sp(0, i) == t.st.sp(1, i) && t.st.sd(0, i) <= 0 && t.st.sd(1, i) >= 0) break;
    if (i == plen[0] || i == plen[1]) return RQ_NIL;
    lca = i; min = t.st.sp(0, lca);
    for (i = lca + 1; i < plen[0]; ++i) {
        if (t.st.sd(0, i) <= 0) {
            const int32_t pi = t.st.sp(0, i);
            if (rq_lt2(t, pi, min)) min = pi;
            const int32_t c = RQ_CH(pi, 1);
            if (c != RQ_NIL) { const int32_t cs = RQ_S(c); if (rq_lt2(t, cs, min)) min = cs; }
        }
    }
    for (i = lca + 1; i < plen[1]; ++i) {
        if (t.st.sd(1, i) >= 0) {
            const int32_t pi = t.st.sp(1, i);
            if (rq_lt2(t, pi, min)) min = pi;
            const int32_t c = RQ_CH(pi, 0);
            if (c != RQ_NIL) { const int32_t cs = RQ_S(c); if (rq_lt2(t, cs, min)) min = cs; }
        }
    }
    return min;
}

// krmq_interval's lower bound and krmq_itr_prev: the largest element <= (y, i), then its in-order predecessors
struct RqItr { int32_t stack[RQ_MAX_DEPTH]; int top; };      // top < 0: exhausted
template <class ST> __device__ inline bool rq_itr_find_le(const RqTreeT<ST> &t, int32_t ky, int32_t ki, RqItr &it)
{
    int32_t p = t.root;
    int d = 0, best = -1;
    while (p != RQ_NIL) {
        const int cmp = rq_cmp_node(t, ky, ki, p);
        if (d >= RQ_MAX_DEPTH - 1) { if (!t.bad) const_cast<RqTreeT<ST> &>(t).bad = 9; it.top = -1; return false; }
        it.stack[d++] = p;
        if (cmp < 0) p = RQ_CH(p, 0);
        else if (cmp > 0) { best = d; p = RQ_CH(p, 1); }
        else { best = d; break; }
    }
    if (best < 0) { it.top = -1; return false; }
    it.top = best - 1;
    return true;
}
template <class ST> __device__ inline bool rq_itr_prev(const RqTreeT<ST> &t, RqItr &it)
{
    int32_t p;
    if (it.top < 0) return false;
    p = RQ_CH(it.stack[it.top], 0);
    if (p != RQ_NIL) {
        for (; p != RQ_NIL; p = RQ_CH(p, 1)) { if (it.top >= RQ_MAX_DEPTH - 2) { if (!t.bad) const_cast<RqTreeT<ST> &>(t).bad = 10; it.top = -1; return false; } it.stack[++it.top] = p; }
        return true;
    }
    do { p = it.stack[it.top--]; } while (it.top >= 0 && p == RQ_CH(it.stack[it.top], 0));
    return it.top >= 0;
}

#if defined(__HIPCC__)
// ---- the same two operations by the whole wave (LDS tree only) ---------------------------------------------------------------------------
// One lane walking the tree spends an operation's time on ~90 dependent LDS round trips (measured: 5 - 6 us an operation), most of them in
// the loops that visit the nodes of the root-to-leaf path one after the other - krmq_update_min over the path, the balance updates.  Here
// all 64 lanes call with the same arguments; the descent is shared (every lane reads the same node: one broadcast ds_read_b128 a level),
// lane d keeps level d of the path in registers, the per-level loads of the path loops leave together (one round trip), the chain that
// carries a subtree minimum from a level to the one above runs on readlane'd registers, and the stores leave together.  What has no width -
// rotations, the walk back up after an erase - is the generic code above on lane 0.  Same tree, bit for bit: the statements are upstream's,
// only their loads are hoisted (tests: sh_dbg_rmq_trace lds = 2 against the oracle's tree).
__device__ inline int rqw_lane() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ inline int32_t rqw_rl(int32_t v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ inline double rqw_rl(double v, int l) { return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l)); }
__device__ inline int32_t rqw_alloc(RqTreeT<RqLds> &t)      // (uniform: every lane keeps the same cursor and free-list head)
{
    if (t.st.free_head != RQ_NIL) { const int32_t k = t.st.free_head; t.st.free_head = __builtin_amdgcn_readfirstlane(RqLds::up(t.st.h_[k].l)); return k; }
    if (t.st.n_used >= t.st.cap) return RQ_NIL;
    return t.st.n_used++;
}
__device__ inline void rqw_free(RqTreeT<RqLds> &t, int32_t k)
{
    if (!t.st.valid(k)) return;
    if (rqw_lane() == 0) t.st.h_[k].l = (uint16_t)t.st.free_head;
    t.st.free_head = k;
}
// the chain of krmq_update_min over path levels [lo, hi): level i's on-path child is level i + 1 (its minimum is what the level below
// just computed; for i == hi - 1 it is (cs0, csp0) when bottom_on, else both children come from memory).  Per lane (level): node, dir,
// own priority, the minima of its left / right child as loaded (ls, lp, rs, rp; xx_ex = the child exists).  stop_at: upstream's insert
// stops behind the first level whose minimum is not `stop_at` (RQ_NIL: never).  Returns the lowest level computed; news / newsp = the
// level's new minimum on its lane.
__device__ inline int rqw_min_chain(int lo, int hi, int32_t pnode, int pdir, double mypri, bool l_ex, int32_t ls, double lp, bool r_ex, int32_t rs, double rp,
                                    bool bottom_on, int32_t cs, double csp, int32_t stop_at, int32_t &news, double &newsp)
{
    const int lane = rqw_lane();
    int dstar = hi;
    for (int d = hi - 1; d >= lo; --d) {
        const int32_t nd = rqw_rl(pnode, d); const double pd = rqw_rl(mypri, d); const int dd = rqw_rl(pdir, d);
        bool lex = rqw_rl((int32_t)l_ex, d) != 0, rex = rqw_rl((int32_t)r_ex, d) != 0;
        int32_t lsd = rqw_rl(ls, d), rsd = rqw_rl(rs, d); double lpd = rqw_rl(lp, d), rpd = rqw_rl(rp, d);
        if (d < hi - 1 || bottom_on) { if (dd == 0) { lex = true; lsd = cs; lpd = csp; } else { rex = true; rsd = cs; rpd = csp; } }
        int32_t sv = nd; double sp = pd;
        if (lex && !(pd < lpd)) { sv = lsd; sp = lpd; }
        if (rex && !(sp < rpd)) { sv = rsd; sp = rpd; }
        if (lane == d) { news = sv; newsp = sp; }
        cs = sv; csp = sp; dstar = d;
        if (stop_at != RQ_NIL && sv != stop_at) break;
    }
    return dstar;
}

// krmq_insert of a new node (y, i, pri); returns its index (RQ_NIL: pool full - the caller gives the read up)
__device__ inline int32_t rq_insert_w(RqTreeT<RqLds> &t, int32_t xy, int32_t xi, double xp)
{
    const int lane = rqw_lane();
    xy = __builtin_amdgcn_readfirstlane(xy); xi = __builtin_amdgcn_readfirstlane(xi); xp = rqw_rl(xp, 0);      // (uniform by contract: scalar from here on)
    const int32_t x = __builtin_amdgcn_readfirstlane(rqw_alloc(t));
    if (x == RQ_NIL) return RQ_NIL;
    if (lane == 0) t.st.fresh(x, xy, xi, xp);
    int32_t pnode = 0, pl = RQ_NIL, pr = RQ_NIL; int pdir = 0, pbal = 0;
    int32_t p = __builtin_amdgcn_readfirstlane(t.root), q = RQ_NIL;
    int depth = 0, which = 0, bp_idx = 0;      // bp = the root until a node with a balance shows up
    while (p != RQ_NIL) {
        int32_t ny, ni, nl, nr, nb;
        t.st.hot_u(rq_ok(t, p), ny, ni, nl, nr, nb);
        const int cmp = rq_cmp_key(xy, xi, ny, ni);
        if (cmp == 0) { rqw_free(t, x); return x; }     // (y, i) is unique: never taken
        if (nb != 0) bp_idx = depth;
        which = (cmp > 0);
        if (lane == depth) { pnode = p; pl = nl; pr = nr; pdir = which; pbal = nb; }
        if (++depth >= RQ_LDS_DEPTH - 1) { if (!t.bad) t.bad = 4; return x; }
        q = p; p = __builtin_amdgcn_readfirstlane(which ? nr : nl);
    }
    ++t.n_live;
    if (q == RQ_NIL) { t.root = x; return x; }      // the tree was empty
    if (lane == 0) t.st.set_ch(q, which, x);
    const int L = depth;
    const bool on = lane < L;
    // krmq_update_min up the path while x is the minimum: the other child's minimum and the node's own priority, all levels at once
    const int32_t off = on ? (pdir ? pl : pr) : RQ_NIL;
    const bool off_ex = off != RQ_NIL;
    const double mypri = on ? t.st.pri(pnode) : 0.0;
    const int32_t os = off_ex ? t.st.s(rq_ok(t, off)) : RQ_NIL;
    const double osp = off_ex ? t.st.spri(rq_ok(t, off)) : 0.0;
    int32_t news = RQ_NIL; double newsp = 0.0;
    const int dstar = rqw_min_chain(0, L, pnode, pdir, mypri, pdir != 0 && off_ex, os, osp, pdir == 0 && off_ex, os, osp, true, x, xp, x, news, newsp);
    if (on && lane >= dstar) t.st.set_s(pnode, news, newsp);
    // balances from bp down to x
    if (on && lane >= bp_idx) { pbal += pdir ? 1 : -1; t.st.set_bal(pnode, pbal); }
    const int bb = rqw_rl(pbal, bp_idx);
    if (bb > -2 && bb < 2) return x;
    const int32_t bp = rqw_rl(pnode, bp_idx), bq = bp_idx > 0 ? rqw_rl(pnode, bp_idx - 1) : RQ_NIL;
    int32_t r = RQ_NIL;
    if (lane == 0) {
        const int wh = (bb < 0);
        const int b1 = wh == 0 ? +1 : -1;
        const int32_t qq = RQ_CH(bp, 1 - wh);
        if (RQ_BAL(qq) == b1) { r = rq_rotate1(t, bp, wh); RQ_SET_BAL(qq, 0); RQ_SET_BAL(bp, 0); }
        else r = rq_rotate2(t, bp, wh);
        if (bq != RQ_NIL) { const int wi = bp != RQ_CH(bq, 0); RQ_SET_CH(bq, wi, r); }
    }
    r = rqw_rl(r, 0); t.bad = rqw_rl(t.bad, 0);
    if (bq == RQ_NIL) t.root = r;
    return x;
}

// krmq_erase of the node with key (y, i); returns its index (freed) or RQ_NIL
__device__ inline int32_t rq_erase_w(RqTreeT<RqLds> &t, int32_t ky, int32_t ki)
{
    const int lane = rqw_lane();
    ky = __builtin_amdgcn_readfirstlane(ky); ki = __builtin_amdgcn_readfirstlane(ki);
    if (t.root == RQ_NIL) return RQ_NIL;
    const int32_t fake = __builtin_amdgcn_readfirstlane(rqw_alloc(t));
    if (fake == RQ_NIL) { if (!t.bad) t.bad = 3; return RQ_NIL; }
    if (lane == 0) t.st.fake(fake, rq_ok(t, t.root));
    int32_t pnode = 0; int pdir = 0;
    int32_t p = fake, p_l = __builtin_amdgcn_readfirstlane(t.root), p_r = RQ_NIL, p_bal = 0;
    int d = 0, cmp = -1;
    while (cmp) {
        const int which = (cmp > 0);
        if (lane == d) { pnode = p; pdir = which; }
        if (++d >= RQ_LDS_DEPTH - 2) { if (!t.bad) t.bad = 5; rqw_free(t, fake); return RQ_NIL; }
        p = __builtin_amdgcn_readfirstlane(which ? p_r : p_l);
        if (p == RQ_NIL) { rqw_free(t, fake); return RQ_NIL; }
        int32_t ny, ni;
        t.st.hot_u(rq_ok(t, p), ny, ni, p_l, p_r, p_bal);
        cmp = rq_cmp_key(ky, ki, ny, ni);
    }
    --t.n_live;
#define WPATH(i) rqw_rl(pnode, (i))
#define WDIR(i) rqw_rl(pdir, (i))
#define WSET(i, nn, dd) do { if (lane == (i)) { pnode = (nn); pdir = (dd); } } while (0)
    const int32_t par = WPATH(d - 1); const int par_dir = WDIR(d - 1);      // the erased node's parent (the fake node stands above the root)
    if (p_r == RQ_NIL) {
        if (lane == 0) RQ_SET_CH(par, par_dir, p_l);
    } else {
        int32_t q = p_r;
        if (t.st.ch_u(rq_ok(t, q), 0) == RQ_NIL) {
            if (lane == 0) { RQ_SET_CH(q, 0, p_l); RQ_SET_BAL(q, p_bal); RQ_SET_CH(par, par_dir, q); }
            WSET(d, q, 1); ++d;
        } else {
            int32_t r;
            const int e = d++;
            for (;;) {
                WSET(d, q, 0); ++d;
                if (d >= RQ_LDS_DEPTH - 1) { if (!t.bad) t.bad = 6; rqw_free(t, fake); return RQ_NIL; }
                r = t.st.ch_u(rq_ok(t, q), 0);
                if (t.st.ch_u(rq_ok(t, r), 0) == RQ_NIL) break;
                q = r;
            }
            const int32_t r_r = t.st.ch_u(rq_ok(t, r), 1);
            if (lane == 0) { RQ_SET_CH(r, 0, p_l); RQ_SET_CH(q, 0, r_r); RQ_SET_CH(r, 1, p_r); RQ_SET_BAL(r, p_bal); RQ_SET_CH(par, par_dir, r); }      // (path[e - 1] is the parent: e = d before the walk)
            WSET(e, r, 1);
        }
    }
    // krmq_update_min over the whole path, bottom up: every level's node and both children's minima at once (the on-path child's minimum is
    // replaced by what the level below computes)
    const int D = d;
    const bool on = lane < D;
    int32_t ny, ni, cl = RQ_NIL, cr = RQ_NIL, cb = 0;
    if (on) t.st.hot(rq_ok(t, pnode), ny, ni, cl, cr, cb);
    const double mypri = on ? t.st.pri(rq_ok(t, pnode)) : 0.0;
    const bool l_ex = on && cl != RQ_NIL, r_ex = on && cr != RQ_NIL;
    const int32_t ls = l_ex ? t.st.s(rq_ok(t, cl)) : RQ_NIL, rs = r_ex ? t.st.s(rq_ok(t, cr)) : RQ_NIL;
    const double lp = l_ex ? t.st.spri(rq_ok(t, cl)) : 0.0, rp = r_ex ? t.st.spri(rq_ok(t, cr)) : 0.0;
    int32_t news = RQ_NIL; double newsp = 0.0;
    rqw_min_chain(0, D, pnode, pdir, mypri, l_ex, ls, lp, r_ex, rs, rp, false, RQ_NIL, 0.0, RQ_NIL, news, newsp);
    if (on) { t.st.set_s(rq_ok(t, pnode), news, newsp); t.st.set_sp(0, lane, pnode); t.st.set_sd(0, lane, pdir); }
    if (lane == 0) rq_erase_rebalance(t, D);
    t.bad = rqw_rl(t.bad, 0);
    t.root = t.st.ch_u(rq_ok(t, fake), 0);
    rqw_free(t, fake);
    return p;
#undef WPATH
#undef WDIR
#undef WSET
}
#endif
