// sh_long.h — the base-level extension stage for the long-read presets (map-ont, lr:hq, map-hifi) on the device.
//
// `.with_cigar()` at /root/reference/src/cleaner.rs:473 applies to every preset of :456-468, and `mappings.len() > 0` (:552-556) is
// then "a region survives mm_filter_regs".  Without MM_F_SR minimap2 takes a longer road from the chains to that count than the
// short-read mode restated in sh_align.h; this header restates it, statement for statement with oracle/mm_align.c and oracle/mm_rmq.c
// (which cite the upstream functions):
//     [mm_map_frag]  more than one chain and bw_long > bw: the chains' anchors are re-chained by mg_lchain_rmq (long join)
//     mm_gen_regs -> mm_set_parent -> mm_select_sub -> mm_est_err -> mm_filter_strand_retained -> mm_squeeze_a
//     per region mm_align1: mm_fix_bad_ends, mm_filter_bad_seeds(_alt), the extension windows from the neighbouring regions' anchors,
//         left extension, ksw_extd2 between anchor mid-points every >= min_ksw_len bases (first pass with the approximate maximum,
//         mm_test_zdrop incl. the inversion test through a local alignment, second pass, mm_split_reg), right extension,
//         mm_update_extra with the logarithmic gap cost; mm_align1_inv behind an inversion split
//     mm_filter_regs
// One WAVE per read.  Data-parallel where there is width (sorting, the RMQ scan over the look-back window, sequence staging, the
// anti-diagonals of ksw_extd2, the rows of the local alignment); the region bookkeeping is scalar work on lane 0 over the wave's HBM
// scratch (phases are separated by lr_sync(): agent-scope fence, so what one lane stored is what the others load).
#pragma once
#include "sh_align.h"
#include "sh_chain.h"
#include "sh_rmq_tree.h"

struct LAnchor { uint64_t x, y; };      // minimap2's mm128_t for an anchor: y = flags | q_span << 32 | qpos
#define LY_LONG_JOIN (1ull << 40)
#define LY_IGNORE    (1ull << 41)
#define LY_TANDEM    (1ull << 42)

struct LReg {                           // mm_reg1_t, the fields the decision reads
    int32_t id, cnt, rid, score, qs, qe, rs, re, parent, as, mlen, blen;
    uint32_t hash;
    int32_t rev, inv, split_inv, strand_retained, split;
    float div;
    int32_t has_p, dp_max, subsc, n_sub;
};
#define LR_PARENT_UNSET (-1)
#define LR_PARENT_TMP_PRI (-2)

struct SKey { uint64_t k, v; uint32_t i, pad; };         // sort element: ascending (k, v); i = payload

struct LongParams {
    int32_t k, min_cnt, min_sc, max_gap, bw, bw_long, min_ksw_len;
    int32_t a, b, q, e, q2, e2, sc_ambi, zdrop, zdrop_inv, end_bonus, min_dp_max, best_n;
    float pri_ratio, mask_level, max_clip_ratio;
    int32_t max_skip, rmq_inner_dist, rmq_size_cap, rmq_rescue_size;
    int32_t rmq_exact_max;          // reads of up to this many chain anchors take the literal tree when the long join meets a tie that matters (-1: all)
    int32_t e2_join_min;            // exact passes: a read of the main grid whose join holds more anchors than this goes to the pass with the 4096-anchor ring (SCRUBBY_HIP_E2_JOIN_MIN; default: never)
    int32_t coop_check;             // (debugging, SCRUBBY_HIP_COOP_CHECK) lr_coop_fill joins the read once more in one piece and reports the first anchor that differs
    int32_t coop_min, coop_run;     // lr_coop_fill: joins of coop_min anchors and more are shared among the launch's waves in runs of coop_run anchors and more
    int32_t rmq_one_lane;           // reads whose windows outgrow the 4096-anchor ring / the LDS tree take the one-lane trees (seconds per read) instead of being counted unresolved
    float rmq_rescue_ratio, pen_gap, pen_skip;
    int32_t mid_occ, max_max_occ, occ_dist;
};

// ---- per-wave working memory (HBM) -------------------------------------------------------------------------------------
struct LongWs {
    LAnchor *a, *b;                 // cap_a: the read's chain anchors (compact_a order) / sort and re-chain buffer
    int32_t *f, *p, *t, *v;         // cap_a: chaining state, backtrack output
    double *pri;                    // cap_a
    SKey *sk, *sk2;                 // cap_a: sort keys
    uint64_t *u; uint32_t *uoff;    // cap_u: chains (score << 32 | cnt) and where their anchors start
    LReg *regs; uint64_t *cov; int32_t *wpri;     // cap_r
    uint64_t *mini_pos;             // cap_m
    uint32_t *tbits;                // cap_q / 32 + 1: query positions of tandem seeds
    int32_t *K;                     // cap_a: positions of long gaps (mm_filter_bad_seeds)
    uint32_t *r_cigar, *ez_cigar;   // cap_c
    uint8_t *qseq, *tseq;           // 2 * cap_q, cap_t
    uint8_t *kmem; int32_t *kH, *koff; uint8_t *kp;      // ksw_extd2: state for up to cap_k x cap_k bases, cap_p direction bytes
    int32_t *lH, *lE, *lHmax;       // local alignment rows (cap_k + 8 each)
    RqNode *rq0, *rq1;              // phase 2 (the exact long join, sh_rmq_tree.h): node pools of the two trees, cap_a + 2 each
    uint32_t cap_a, cap_u, cap_r, cap_m, cap_c, cap_q, cap_t, cap_k; unsigned long long cap_p;
};

struct LongSizes { uint32_t cap_a, cap_u, cap_r, cap_m, cap_q, cap_t, cap_k; unsigned long long cap_p; int32_t phase; };      // phase 0: the chains kernel, 1: the regions / alignment kernel, 2: the chains kernel with the long join on the literal trees

__host__ __device__ inline unsigned long long long_ws_carve(LongWs *W, uint8_t *base, const LongSizes &z)
{
    unsigned long long off = 0;
    auto take = [&](unsigned long long bytes) { uint8_t *q = base ? base + off : nullptr; off += (bytes + 255) & ~255ull; return q; };
    const unsigned long long ca = z.cap_a, cu = z.cap_u, cr = z.cap_r, cq = z.cap_q, ct = z.cap_t, ck = z.cap_k;
    const unsigned long long cc = 2ull * (cq + ct) + 64;
    LongWs w{};
    if (z.phase == 0 || z.phase == 2) {
        if (z.phase == 2) { w.rq0 = (RqNode *)take((ca + 2) * sizeof(RqNode)); w.rq1 = (RqNode *)take((ca + 2) * sizeof(RqNode)); }
        w.a = (LAnchor *)take(ca * 16); w.b = (LAnchor *)take(ca * 16);
        w.f = (int32_t *)take(ca * 4); w.p = (int32_t *)take(ca * 4); w.t = (int32_t *)take(ca * 4); w.v = (int32_t *)take(ca * 4);
        w.pri = (double *)take(ca * 8); w.sk = (SKey *)take(ca * sizeof(SKey)); w.sk2 = (SKey *)take(ca * sizeof(SKey));
        w.u = (uint64_t *)take(cu * 8); w.uoff = (uint32_t *)take((cu + 1) * 4);
        w.tbits = (uint32_t *)take((cq / 32 + 2) * 4);
        w.K = (int32_t *)take(ca * 4);
    } else {
        w.a = (LAnchor *)take(ca * 16);
        w.sk = (SKey *)take(cr * sizeof(SKey)); w.sk2 = (SKey *)take(cr * sizeof(SKey));
        w.regs = (LReg *)take(cr * sizeof(LReg)); w.cov = (uint64_t *)take(cr * 8); w.wpri = (int32_t *)take(cr * 4);
        w.mini_pos = (uint64_t *)take((unsigned long long)z.cap_m * 8);
        w.K = (int32_t *)take(ca * 4);
        w.r_cigar = (uint32_t *)take(cc * 4); w.ez_cigar = (uint32_t *)take(cc * 4);
        w.qseq = take(2 * cq + 32); w.tseq = take(ct + 32);
        w.kmem = take(8 * (ck + 16) + ck + 64); w.kH = (int32_t *)take((ck + 16) * 4); w.koff = (int32_t *)take(8 * (2 * ck + 32)); w.kp = take(z.cap_p);
        w.lH = (int32_t *)take((ck + 16) * 4 * 2); w.lE = (int32_t *)take((ck + 16) * 4); w.lHmax = (int32_t *)take((ck + 16) * 4);
    }
    w.cap_a = z.cap_a; w.cap_u = z.cap_u; w.cap_r = z.cap_r; w.cap_m = z.cap_m; w.cap_c = (uint32_t)cc; w.cap_q = z.cap_q; w.cap_t = z.cap_t; w.cap_k = z.cap_k; w.cap_p = z.cap_p;
    if (W) *W = w;
    return off;
}

// ---- small helpers ---------------------------------------------------------------------------------------------------
// What one lane stored is what the others load: all of it is traffic of ONE wave through one L1 and one L2, so it takes the stores to
// have reached the L2 (s_waitcnt) and the L1 to forget what it holds (buffer_inv) - not the agent-scope release of a seq_cst fence, whose
// L2 write-back made every phase change cost microseconds.
__device__ inline void lr_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\tbuffer_inv sc1" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}
__device__ inline uint64_t lr_b0_64(uint64_t v) { return al_b0_64(v); }
__device__ inline uint64_t lr_readlane64(uint64_t v, int l)
{
    return (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l) | (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l) << 32;
}
__device__ inline bool sk_less(const SKey &a, const SKey &b) { return a.k < b.k || (a.k == b.k && a.v < b.v); }

// Stable-by-construction sort of n distinct (k, v) pairs, ascending, by one wave: tiles of 64 ranked in registers, then merge
// passes in which every lane merges one slice of the output found by a merge-path search.  Result in `a` (tmp is scratch).
__device__ __noinline__ void lr_sort(SKey *a, SKey *tmp, int32_t n)
{
    const int32_t lane = (int32_t)al_lane();
    if (n < 2) return;
    for (int32_t t0 = 0; t0 < n; t0 += 64) {       // 64-tiles: rank = number of smaller elements of the tile
        const int32_t i = t0 + lane, m = n - t0 < 64 ? n - t0 : 64;
        SKey e{~0ull, ~0ull, 0u, 0u};
        if (i < n) e = a[i];
        int32_t rank = 0;
        for (int32_t l = 0; l < m; ++l) {
            SKey o; o.k = lr_readlane64(e.k, l); o.v = lr_readlane64(e.v, l); o.i = 0; o.pad = 0;
            rank += sk_less(o, e);
        }
        if (i < n) tmp[t0 + rank] = e;
    }
    lr_sync();
    SKey *src = tmp, *dst = a;
    for (int32_t width = 64; width < n; width <<= 1) {
        for (int32_t base = 0; base < n; base += 2 * width) {
            const int32_t l0 = base, l1 = base + width < n ? base + width : n, r1 = base + 2 * width < n ? base + 2 * width : n;
            const int32_t la = l1 - l0, lb = r1 - l1, tot = la + lb;
            if (lb == 0) { for (int32_t i = l0 + lane; i < l1; i += 64) dst[i] = src[i]; continue; }
            // slices of the output: lane `lane` produces [o0, o1)
            const int32_t per = (tot + 63) / 64, o0 = lane * per < tot ? lane * per : tot, o1 = o0 + per < tot ? o0 + per : tot;
            if (o0 < o1) {
                // merge path: i elements of A and o0 - i of B precede the slice; A wins ties (stability is irrelevant: pairs are distinct)
                int32_t lo = o0 - lb > 0 ? o0 - lb : 0, hi = o0 < la ? o0 : la;
                while (lo < hi) {
                    const int32_t mid = (lo + hi) >> 1;       // take mid from A?
                    if (sk_less(src[l1 + (o0 - mid - 1)], src[l0 + mid])) hi = mid; else lo = mid + 1;
                }
                int32_t ia = lo, ib = o0 - lo;
                for (int32_t o = o0; o < o1; ++o) {
                    const bool ta = ib >= lb || (ia < la && !sk_less(src[l1 + ib], src[l0 + ia]));
                    dst[l0 + o] = ta ? src[l0 + ia] : src[l1 + ib];
                    if (ta) ++ia; else ++ib;
                }
            }
        }
        lr_sync();
        SKey *x = src; src = dst; dst = x;
    }
    if (src != a) { for (int32_t i = lane; i < n; i += 64) a[i] = src[i]; lr_sync(); }
}

// ---- hashing, coordinates (hit.c) --------------------------------------------------------------------------------------
__device__ inline void lr_reg_set_coor(LReg &r, int32_t qlen, const LAnchor *a)
{   // mm_reg_set_coor + mm_cal_fuzzy_len
    const int32_t k = r.as, q_span = (int32_t)(a[k].y >> 32 & 0xff);
    r.rev = (int32_t)(a[k].x >> 63);
    r.rid = (int32_t)(a[k].x << 1 >> 33);
    r.rs = (int32_t)a[k].x + 1 > q_span ? (int32_t)a[k].x + 1 - q_span : 0;
    r.re = (int32_t)a[k + r.cnt - 1].x + 1;
    if (!r.rev) { r.qs = (int32_t)a[k].y + 1 - q_span; r.qe = (int32_t)a[k + r.cnt - 1].y + 1; }
    else { r.qs = qlen - ((int32_t)a[k + r.cnt - 1].y + 1); r.qe = qlen - ((int32_t)a[k].y + 1 - q_span); }
    r.mlen = r.blen = 0;
    if (r.cnt <= 0) return;
    r.mlen = r.blen = q_span;
    for (int32_t i = r.as + 1; i < r.as + r.cnt; ++i) {
        const int32_t span = (int32_t)(a[i].y >> 32 & 0xff);
        const int32_t tl = (int32_t)a[i].x - (int32_t)a[i - 1].x, ql = (int32_t)a[i].y - (int32_t)a[i - 1].y;
        r.blen += tl > ql ? tl : ql;
        r.mlen += tl > span && ql > span ? span : tl < ql ? tl : ql;
    }
}

// the same with the whole wave on one region (uniform result): the fuzzy lengths are sums over the region's ~1000 anchor pairs
__device__ inline void lr_reg_set_coor_wave(LReg &r, int32_t qlen, const LAnchor *a)
{
    const int32_t k = r.as, q_span = (int32_t)(a[k].y >> 32 & 0xff), lane = (int32_t)al_lane();
    r.rev = (int32_t)(a[k].x >> 63);
    r.rid = (int32_t)(a[k].x << 1 >> 33);
    r.rs = (int32_t)a[k].x + 1 > q_span ? (int32_t)a[k].x + 1 - q_span : 0;
    r.re = (int32_t)a[k + r.cnt - 1].x + 1;
    if (!r.rev) { r.qs = (int32_t)a[k].y + 1 - q_span; r.qe = (int32_t)a[k + r.cnt - 1].y + 1; }
    else { r.qs = qlen - ((int32_t)a[k + r.cnt - 1].y + 1); r.qe = qlen - ((int32_t)a[k].y + 1 - q_span); }
    r.mlen = r.blen = 0;
    if (r.cnt <= 0) return;
    int32_t ml = 0, bl = 0;
    for (int32_t i = r.as + 1 + lane; i < r.as + r.cnt; i += 64) {
        const int32_t span = (int32_t)(a[i].y >> 32 & 0xff);
        const int32_t tl = (int32_t)a[i].x - (int32_t)a[i - 1].x, ql = (int32_t)a[i].y - (int32_t)a[i - 1].y;
        bl += tl > ql ? tl : ql;
        ml += tl > span && ql > span ? span : tl < ql ? tl : ql;
    }
    ml = wave_all_add(ml); bl = wave_all_add(bl);
    r.mlen = q_span + ml; r.blen = q_span + bl;
}

// where a wave's time goes (SCRUBBY_HIP_DBG): 0 gather, 1 rmq sort, 2 rmq fill, 3 backtrack + compact, 4 gen_regs, 5 parent / select / est_err, 6 squeeze,
// 7 region set-up (bad ends / seeds, windows), 8 ksw, 9 z-drop test, 10 update_extra, 11 sequence staging
#define LR_NCLK 12
struct LongClk { unsigned long long t[LR_NCLK]; unsigned long long last; unsigned long long d[8]; unsigned long long w_max, n_q, n_seg, tie_seg_a; };      // d: RMQ statistics (steps, ring blocks evaluated, steps that went behind the ring, old blocks evaluated, sum of the list length, inner chunks, anchors)
__device__ inline void lr_tick(LongClk *c, int ph) { if (c) { const unsigned long long now = wall_clock64(); c->t[ph] += now - c->last; c->last = now; } }

// ---- mg_lchain_rmq on one wave -------------------------------------------------------------------------------------------
// Upstream keeps the look-back window in two balanced trees keyed by (y, i) (oracle/mm_rmq.c).  What the trees are asked:
//   (1) the element of smallest priority -(f + 0.5 * pen_gap * (x + y)) among the active anchors with y in the query interval;
//   (2) the active anchors of the inner window in descending (y, i) order from just below y_i.
// Here (1) is a wave-wide scan over the active range [st, i0) of the x-sorted anchors with an arg-min reduction, and (2) walks a
// y-sorted copy of the inner window kept in LDS.  With distinct priorities the answers are the trees'; when the minimum is shared by
// two candidates the tree's choice depends on its shape, which this scan does not have: such reads are reported (*tie) and the
// caller sends them down the exact serial path.
// NR = the newest anchors kept in LDS: a ring of their (x, y, f, p, t, priority) and the y-sorted list of the inner window.  The kernel runs
// with NR = 512 (19 KB: eight waves per CU); a read whose inner window or same-x group outgrows it is redone with NR = 4096.
#define LRQ_RBLK(NR) ((NR) / 64 - 4)            // completed blocks of 64 anchors that are always inside the ring ...
#define LRQ_SAMEX(NR) ((NR) - 1 - (LRQ_RBLK(NR) + 1) * 64)      // ... with up to this many anchors waiting on one x
// FAT: the list also holds each entry's y and the ring each anchor's priority (36 B a slot).  Without them (24 B a slot: 13 KB at 512 slots,
// twelve waves to a CU instead of eight) an entry's y is read through its index and the priority is recomputed from rf, rx, ry: more waves
// per CU for the many ordinary reads, a longer dependent chain per step for a wave that has a CU's issue slots to itself (the giants).
template <int LRQ_INNER, bool FAT = false>
struct RmqLdsT {
    int32_t ij[LRQ_INNER];      // the inner window's list, ascending (y, j)
    uint32_t rx[LRQ_INNER]; int32_t ry[LRQ_INNER], rf[LRQ_INNER], rp[LRQ_INNER], rt[LRQ_INNER];
    int32_t iy[FAT ? LRQ_INNER : 1]; double rpri[FAT ? LRQ_INNER : 1];
    double pml[64], bml[64];    // pml[b & 63] = smallest priority of the blocks 0 .. b, bml[b & 63] = of block b alone
};

__device__ inline int32_t lr_sc_simple(int32_t dr, int32_t dq, int32_t q_span, float pen_gap, float pen_skip, int32_t &exact, int32_t &width)
{   // comput_sc_simple
    const int32_t dd = dr > dq ? dr - dq : dq - dr, dg = dr < dq ? dr : dq;
    int32_t sc = q_span < dg ? q_span : dg;
    width = dd;
    exact = (dd == 0 && dg <= q_span);
    if (dd || dq > q_span) {
        const float lin_pen = pen_gap * (float)dd + pen_skip * (float)dg;
        const float log_pen = dd >= 1 ? al_mg_log2((float)(dd + 1)) : 0.0f;
        sc -= (int32_t)(lin_pen + .5f * log_pen);
    }
    return sc;
}
// the smallest of a wave's doubles, on every lane: four DPP stages inside the rows of 16 and the four row results through scalar registers (a
// butterfly of __shfl_xor is eighteen ds_bpermute round trips in a row - it was most of a long-join step)
template <int CTRL>
__device__ inline double lr_dpp_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ inline double lr_readlane_f64(double v, int l)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ inline double lr_wave_min_f64(double v)
{
    double t;
    t = lr_dpp_f64<0xB1>(v); v = t < v ? t : v;       // quad_perm [1,0,3,2]
    t = lr_dpp_f64<0x4E>(v); v = t < v ? t : v;       // quad_perm [2,3,0,1]
    t = lr_dpp_f64<0x141>(v); v = t < v ? t : v;      // row_half_mirror
    t = lr_dpp_f64<0x140>(v); v = t < v ? t : v;      // row_mirror
    const double r0 = lr_readlane_f64(v, 0), r1 = lr_readlane_f64(v, 16), r2 = lr_readlane_f64(v, 32), r3 = lr_readlane_f64(v, 48);
    const double m01 = r1 < r0 ? r1 : r0, m23 = r3 < r2 ? r3 : r2;
    return m23 < m01 ? m23 : m01;
}
__device__ inline double lr_cc_f64(const double *p) { return __longlong_as_double((long long)cc_u64(p)); }

// a[] sorted by x (read-only here).  Out: f, p (int32; -1 = none) - visible to the other lanes after the caller's lr_sync().  n_tie: steps
// whose minimum priority was shared (the smallest index was taken; upstream's tree may pick another).  Returns false when the inner window
// outgrows LRQ_INNER (or, TREE, the look-back window outgrows the LDS tree).
// Memory: what a step needs of the newest ~1000 anchors (f, p, the t marks, x, y, the priority) lives in the LDS ring, and the smallest
// priority of everything older is one number (pml): a step only goes to HBM when the answer may lie further back than the ring.
// TREE: lane 0 keeps upstream's main tree beside the scan (insert when anchors enter the window, erase when they leave, sh_rmq_tree.h) and the
// tree answers query (1) whenever the scan finds the smallest priority shared - n_tie then counts those steps, none of which is left open.
template <int LRQ_INNER, bool TREE, bool FAT = false>
__device__ inline bool lr_rmq_fill(const LongParams &P, int32_t max_dist_in, int32_t bw, int32_t n, const LAnchor *a, int32_t *f, int32_t *p,
                                   double *pri, double *bmin /* n / 64 + 1 */, RmqLdsT<LRQ_INNER, FAT> &L, int32_t &n_tie, LongClk *dbg = nullptr,
                                   RqLds TL = RqLds{}, int32_t p_base = 0)
{
    // (p_base: a[] is a stretch of a longer array that starts where the look-back window is empty - lr_coop - and p is stored as an index
    // into that array)
    // TREE: the main tree in LDS (sh_rmq_tree.h, RqLds): it holds the anchors of the look-back window only - max_dist reference bases - and is
    // walked by lane 0; a window that outgrows its nodes ends the call (ok = false), like a window that outgrows the ring
    RqTreeT<RqLds> T0;
    T0.st = TL; T0.st.n_used = 0; T0.st.free_head = RQ_NIL; rq_reset(T0);
    int32_t st_tree = 0;      // anchors [st_tree, i0) are in the tree (while tree_on)
    // The tree is only ever ASKED at a step whose smallest priority is shared, and what it answers depends on its shape - on every insertion
    // and removal since it was last EMPTY, and on nothing before that.  The look-back window empties whenever the reference position jumps by
    // more than max_dist (another contig, strand or locus: some twenty times in a read of 70 k anchors), so the tree is kept only from the
    // start of the stretch that holds such a step: built there by replaying the stretch's insertions and removals (they depend on x and on
    // priorities that are final by then), kept up to date until the window empties again, and dropped.  9 % of the anchors of the bench's
    // exact reads lie in such stretches.
    bool tree_on = false;
    int32_t seg_i = -1, seg_i0 = 0;      // the window was last found empty after step seg_i's trim, with i0 = st = seg_i0
    n_tie = 0;
    // the parameters in registers: P lives in the caller's scratch, and a load from it inside the loop costs more than the step's arithmetic
    const float pen_gap = P.pen_gap, pen_skip = P.pen_skip;
    const int32_t kk = P.k, max_skip = P.max_skip, rmq_inner_dist = P.rmq_inner_dist, rmq_size_cap = P.rmq_size_cap;
    int32_t tie_cnt = 0, dbg_seg0 = 0;
    unsigned long long d_ring = 0, d_oldsteps = 0, d_old = 0, d_nin = 0, d_chunks = 0;
    const int32_t lane = (int32_t)al_lane();
    constexpr int32_t M = LRQ_INNER - 1;
    auto rpri_calc = [&](int32_t j) -> double { return -((double)L.rf[j & M] + 0.5 * (double)pen_gap * (double)((int32_t)L.rx[j & M] + L.ry[j & M])); };      // of a ring anchor whose f is final
    auto rpri_of = [&](int32_t j) -> double { if constexpr (FAT) return L.rpri[j & M]; else return rpri_calc(j); };
    int32_t max_dist = max_dist_in, max_dist_inner = rmq_inner_dist;
    if (max_dist < bw) max_dist = bw;
    if (max_dist_inner < 0) max_dist_inner = 0;
    if (max_dist_inner > max_dist) max_dist_inner = max_dist;
    // (rmq_size_cap: upstream's trim loop also runs while the tree holds more than rmq_size_cap nodes; the tree holds the anchors
    // [st, i0), so that is st = max(st, i0 - rmq_size_cap) - first in, first out like every other exit: handled where st moves.  The inner
    // window's tree has the same rule; its window must fit the ring, so a cap below the ring's size is left to the one-lane version)
    if (rmq_size_cap < LRQ_INNER) return false;
    for (int32_t i = lane; i < LRQ_INNER; i += 64) L.rt[i] = -1;
    __builtin_amdgcn_wave_barrier();
    int32_t blk_done = 0;      // blocks [0, blk_done) of 64 anchors are completely inserted; bmin[b] = their smallest priority
    int32_t i0 = 0, st = 0, st_inner = 0, n_in = 0, seg0 = 0, head = 0;      // the list is circular: entry e lives at (head + e) & M
#define LIJ(e) L.ij[(head + (e)) & M]
#define LIY(e) (FAT ? L.iy[FAT ? (head + (e)) & M : 0] : L.ry[LIJ(e) & M])
#define LIY_SET(e, v) do { if constexpr (FAT) L.iy[(head + (e)) & M] = (v); } while (0)      // inner window: the live entries (j >= st_inner) of L.iy/ij[0 .. n_in), ascending (y, j)
    bool ok = true;
    uint32_t hi_prev = 0;
    LAnchor cur = a[0];
    int32_t xw_base = 0;
    uint64_t xw = a[lane < n ? lane : n - 1].x;
    unsigned long long pt[6] = {0, 0, 0, 0, 0, 0}, pl = dbg ? wall_clock64() : 0ull;
#define LRQ_T(k) do { if (dbg) { const unsigned long long nw = wall_clock64(); pt[k] += nw - pl; pl = nw; } } while (0)
    for (int32_t i = 0; i < n && ok; ++i) {
        const uint64_t xi = cur.x, yi = cur.y;
        if (i + 1 < n) cur = a[i + 1];      // the next anchor travels while this one is worked on
        const int32_t qi = (int32_t)yi, q_span_i = (int32_t)(yi >> 32 & 0xff);
        if (i > 0 && (uint32_t)(xi >> 32) != hi_prev) seg0 = i;      // another strand / contig: nothing before i is in range
        // add the anchors whose x is now strictly smaller
        if (i - i0 > LRQ_SAMEX(LRQ_INNER)) { ok = false; break; }      // more anchors on one reference position than the ring can hold back
        if (i0 < i && (seg0 == i || L.rx[i0 & M] != (uint32_t)xi)) {
            if constexpr (FAT) {
                for (int32_t jb = i0; jb < i; jb += 64) { const int32_t j = jb + lane; if (j < i) L.rpri[j & M] = rpri_calc(j); }
                __builtin_amdgcn_wave_barrier();
            }
            if (max_dist_inner > 0) {
                for (int32_t j = i0; j < i; ++j) {      // insert (y_j, j) into the y-sorted inner window
                    if (n_in >= LRQ_INNER) { ok = false; break; }
                    const int32_t yj = L.ry[j & M];
                    // position: behind every element with y <= yj (j is the largest index so far); searched from the top, where a collinear
                    // anchor belongs
                    int32_t pos = n_in;
                    for (int32_t c = n_in; c > 0; c -= 64) {
                        const int32_t e = c - 1 - lane;      // lane 0 = the topmost entry of the chunk
                        const uint64_t gt = __ballot(e >= 0 && LIY(e) > yj);      // a prefix of the lanes
                        pos -= (int32_t)__popcll(gt);
                        if (__popcll(gt) < 64) break;
                    }
                    for (int32_t c = ((n_in - pos + 63) / 64 - 1) * 64; c >= 0; c -= 64) {      // shift [pos, n_in) up by one, from the top
                        const int32_t e = pos + c + lane;
                        int32_t vj = 0, vy = 0;
                        const bool on = e < n_in;
                        if (on) { vj = LIJ(e); if constexpr (FAT) vy = LIY(e); }
                        __builtin_amdgcn_wave_barrier();
                        if (on) { LIJ(e + 1) = vj; LIY_SET(e + 1, vy); }
                        __builtin_amdgcn_wave_barrier();
                    }
                    if (lane == 0) { LIJ(pos) = j; LIY_SET(pos, yj); }
                    __builtin_amdgcn_wave_barrier();
                    ++n_in;
                }
                if (!ok) break;
            }
            if (TREE && tree_on) {      // (the whole wave: rq_insert_w)
                for (int32_t j = i0; j < i; ++j)
                    if (rq_insert_w(T0, L.ry[j & M], j, rpri_of(j)) == RQ_NIL) { T0.bad = 11; break; }
            }
            if (TREE && T0.bad) { ok = false; break; }      // the tree's nodes or a walk gave out: the caller takes the read to the instance with the larger tree
            i0 = i;
            while ((blk_done + 1) * 64 <= i0) {      // blocks completed by this insertion (their anchors are all in the ring)
                const double m = lr_wave_min_f64(rpri_of(blk_done * 64 + lane));
                const double pm = blk_done > 0 && L.pml[(blk_done - 1) & 63] < m ? L.pml[(blk_done - 1) & 63] : m;
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) { bmin[blk_done] = m; L.pml[blk_done & 63] = pm; L.bml[blk_done & 63] = m; }
                {   // the block leaves for HBM in one piece (a store inside the loop would make every later wait of the step wait for it too)
                    const int32_t j = blk_done * 64 + lane;
                    f[j] = L.rf[j & M]; p[j] = L.rp[j & M] < 0 ? -1 : L.rp[j & M] + p_base; pri[j] = rpri_of(j);
                }
                __builtin_amdgcn_wave_barrier();
                ++blk_done;
            }
        }
        LRQ_T(0);
        // anchors out of range leave
        // (the x of a[xw_base + lane] waits in a register: the window's tail moves by about one anchor a step, and a load per move would be a
        // trip to memory per anchor - the step's longest wait; this way there is one per 64 moves)
        if (st < seg0) st = seg0;
        while (st < i) {
            if (st < xw_base || st >= xw_base + 64) { xw_base = st; xw = a[st + lane < n ? st + lane : n - 1].x; }
            const int32_t j = xw_base + lane;
            const uint64_t stay = __ballot(j >= st && !(j < i && xi > xw + (uint64_t)max_dist));      // a[] is sorted: the anchors that leave are a prefix
            const int32_t first = stay ? (int32_t)__ffsll((unsigned long long)stay) - 1 : 64;
            st = xw_base + first;
            if (first < 64) break;
        }
        if (i0 - st > rmq_size_cap) st = i0 - rmq_size_cap;
        if (TREE && tree_on && st_tree < st) {
            for (int32_t j = st_tree; j < st && j < i0; ++j) { const int32_t e = rq_erase_w(T0, (int32_t)a[j].y, j); if (e != RQ_NIL) rqw_free(T0, e); }
            st_tree = st;
        }
        if (TREE && st >= i0) {      // the window is empty, and so is upstream's tree: what follows does not depend on anything before
            if (tree_on) { tree_on = false; T0.st.n_used = 0; T0.st.free_head = RQ_NIL; rq_reset(T0); }
            seg_i = i; seg_i0 = i0;
        }
        if (TREE && dbg && (unsigned long long)(i0 - st) > dbg->w_max) dbg->w_max = (unsigned long long)(i0 - st);
        if (TREE && dbg && st >= i0 && i0 > dbg_seg0) { ++dbg->n_seg; dbg_seg0 = i0; }
        if (TREE && dbg && tree_on) ++dbg->tie_seg_a;      // steps with the tree kept
        if (max_dist_inner > 0) {
            const int32_t st_old = st_inner;
            if (st_inner < seg0) st_inner = seg0;
            if (i - st_inner >= LRQ_INNER) { ok = false; break; }      // the inner window must fit the ring
            while (st_inner < i && (uint32_t)xi - L.rx[st_inner & M] > (uint32_t)max_dist_inner) ++st_inner;
            if (st_inner > st_old && n_in > 0) {      // drop the entries that left the window
                // the anchors that leave are the oldest; on a collinear chain they are also the lowest in y, i.e. the head of the list
                int32_t d = st_inner - st_old;
                if (d > n_in) d = n_in;
                bool prefix = d <= 64;
                if (prefix) { const uint64_t dm = __ballot(lane < d && LIJ(lane) < st_inner); prefix = (int32_t)__popcll(dm) == d; }
                if (prefix) { head = (head + d) & M; n_in -= d; }
                else {
                    int32_t kept = 0;
                    for (int32_t c = 0; c < n_in; c += 64) {
                        const int32_t e = c + lane;
                        int32_t vj = 0, vy = 0;
                        const bool on = e < n_in;
                        if (on) { vj = LIJ(e); if constexpr (FAT) vy = LIY(e); }
                        const bool keep = on && vj >= st_inner;
                        const uint64_t km = __ballot(keep);
                        __builtin_amdgcn_wave_barrier();
                        if (keep) { const int32_t dd = kept + (int32_t)prefix_popc64(km); LIJ(dd) = vj; LIY_SET(dd, vy); }
                        __builtin_amdgcn_wave_barrier();
                        kept += (int32_t)__popcll(km);
                    }
                    n_in = kept;
                }
            }
        }
        d_nin += (unsigned long long)n_in;
        LRQ_T(1);
        int32_t max_f = q_span_i, max_j = -1;
        // (1) range minimum of the priority over the active anchors [st, i0) with (y_j, j) in [(q_i - max_dist, INT32_MAX), (q_i, 0)].
        // Newest first in blocks of 64: the blocks inside the ring out of LDS; everything older only if its smallest priority (pml, whatever
        // the y) is not above the best found so far - then block by block, skipping those whose own minimum (bmin) is above it.
        double bp = 0.0; int32_t bj = -1, ties = 0;
        {
            auto reduce = [&](bool in, double pj, int32_t j) {      // j = a block's base + lane at every call: among equals the lowest lane has the smallest index
                const uint64_t im = __ballot(in);
                if (im == 0) return;
                const double wm = lr_wave_min_f64(in ? pj : __longlong_as_double(0x7ff0000000000000ll));
                const uint64_t em = __ballot(in && pj == wm);
                const int32_t wj = __builtin_amdgcn_readlane(j, (int)(__ffsll((unsigned long long)em) - 1));
                const int32_t c = (int32_t)__popcll(em);
                if (bj < 0 || wm < bp) { bp = wm; bj = wj; ties = c; }
                else if (wm == bp) { ties += c; bj = wj < bj ? wj : bj; }
            };
            const int32_t ring_lo = blk_done > LRQ_RBLK(LRQ_INNER) ? (blk_done - LRQ_RBLK(LRQ_INNER)) * 64 : 0;      // anchors [ring_lo, i0) are in the ring
            auto ring_eval = [&](int32_t j, bool cand) {
                bool in = false; double pj = 0.0;
                if (cand) {
                    const int32_t yj = L.ry[j & M];
                    in = yj > qi - max_dist && (yj < qi || (yj == qi && j + p_base == 0));      // (upstream's bound is the key (q_i, 0): index 0 of the WHOLE array)
                    if (in) pj = rpri_of(j);
                }
                reduce(in, pj, j);
            };
            {   // the newest, not yet completed block
                const int32_t part = blk_done * 64 > st ? blk_done * 64 : st;
                if (part < i0) { const int32_t j = part + lane; ring_eval(j, j < i0); }
            }
            {   // the completed blocks inside the ring, newest first, those whose own minimum can still matter
                const int32_t rb_lo = (ring_lo >> 6) > (st >> 6) ? (ring_lo >> 6) : (st >> 6);
                const int32_t b = blk_done - 1 - lane;
                bool todo = b >= rb_lo;
                const double v = todo ? L.bml[b & 63] : 0.0;
                for (;;) {
                    const uint64_t m = __ballot(todo && (bj < 0 || v <= bp));
                    if (m == 0) break;
                    const int l = __ffsll((unsigned long long)m) - 1;
                    const int32_t j = (blk_done - 1 - l) * 64 + lane;
                    ring_eval(j, j >= st);
                    ++d_ring;
                    if (lane == l) todo = false;
                }
            }
            const int32_t b_lo = st >> 6, b_hi = ring_lo / 64 - 1;      // completed blocks older than the ring: [b_lo, b_hi]
            if (b_hi >= b_lo && (bj < 0 || L.pml[b_hi & 63] <= bp)) {
                ++d_oldsteps;
                for (int32_t bt = b_hi; bt >= b_lo; bt -= 64) {
                    const int32_t b = bt - lane;
                    bool todo = b >= b_lo;
                    const double v = todo ? lr_cc_f64(bmin + b) : 0.0;
                    for (;;) {
                        uint64_t m = __ballot(todo && (bj < 0 || v <= bp));
                        if (m == 0) break;
                        // up to four such blocks a trip, newest first: their loads leave together (one memory latency instead of eight - a block
                        // evaluated although an earlier one of the trip would have ruled it out only adds candidates of the window to the minimum)
                        int32_t jv[4]; int32_t yv[4]; double pv[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            jv[q] = -1; yv[q] = 0; pv[q] = 0.0;
                            if (m) {
                                const int l = __ffsll((unsigned long long)m) - 1; m &= m - 1;
                                if (lane == l) todo = false;
                                const int32_t j = (bt - l) * 64 + lane;
                                jv[q] = j;
                                if (j >= st) { yv[q] = (int32_t)a[j].y; pv[q] = lr_cc_f64(pri + j); }
                            }
                        }
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            if (jv[q] < 0) continue;      // (uniform: jv is -1 on every lane or on none)
                            const bool in = jv[q] >= st && yv[q] > qi - max_dist && (yv[q] < qi || (yv[q] == qi && jv[q] + p_base == 0));
                            reduce(in, pv[q], jv[q]);
                            ++d_old;
                        }
                    }
                }
            }
        }
        LRQ_T(2);
        bool tie_pending = false; int32_t tie_sc = 0;
        if (TREE && bj >= 0 && ties > 1) {
            if (!tree_on) {
                // upstream's tree as it stands after this step's insertions and removals: the steps seg_i + 1 .. i once more, tree operations only
                int32_t i0r = seg_i0, str = seg_i0;
                uint64_t x_i0r = a[seg_i0 < n ? seg_i0 : n - 1].x;
                st_tree = seg_i0;
                for (int32_t ii = seg_i + 1; ii <= i && !T0.bad; ++ii) {
                    const uint64_t xr = a[ii].x;
                    if (i0r < ii && x_i0r != xr) {
                        for (int32_t j = i0r; j < ii; ++j) {
                            const double pj = j < blk_done * 64 ? lr_cc_f64(pri + j) : rpri_of(j);
                            const int32_t yj = j < blk_done * 64 ? (int32_t)a[j].y : L.ry[j & M];
                            if (rq_insert_w(T0, yj, j, pj) == RQ_NIL) { T0.bad = 11; break; }
                        }
                        i0r = ii; x_i0r = xr;
                    }
                    while (str < ii && xr > a[str].x + (uint64_t)max_dist) ++str;
                    if (i0r - str > rmq_size_cap) str = i0r - rmq_size_cap;
                    if (st_tree < str && !T0.bad) {
                        for (int32_t j = st_tree; j < str && j < i0r; ++j) { const int32_t e = rq_erase_w(T0, (int32_t)a[j].y, j); if (e != RQ_NIL) rqw_free(T0, e); }
                        st_tree = str;
                    }
                }
                if (!T0.bad && (i0r != i0 || str != st)) T0.bad = 12;      // (the replay and the scan must agree on the window)
                if (T0.bad) { ok = false; break; }
                tree_on = true;
                if (dbg) dbg->tie_seg_a += (unsigned long long)(i - seg_i);      // anchors replayed
            }
            int32_t tj = -1;
            if (lane == 0) { const int32_t q = rq_rmq(T0, qi - max_dist, INT32_MAX, qi, -p_base); tj = q != RQ_NIL ? rq_i(T0, q) : -1; }
            tj = al_b0(tj);
            if (tj >= 0) bj = tj;
            ++tie_cnt;
            if (dbg) ++dbg->n_q;
        } else if (bj >= 0 && ties > 1 && tie_cnt == 0) {      // (a read that has met a tie that matters needs no further verdicts)
            // Several candidates share the smallest priority; the tree returns ONE of them, which one depends on its shape.  The state after
            // this step is the same whichever it is when (a) none of them is accepted (band, score) and they agree on `exact` - the step
            // keeps max_f = q_span, max_j = -1 and runs the inner scan or not - or (b) all are accepted with one score, none exact, and the
            // inner scan then finds a strictly better predecessor: max_f and the skip counter evolve alike and max_j is the scan's.
            // Anything else makes the read one for the literal trees (sh_rmq_tree.h).
            const int32_t ring_lo2 = blk_done > LRQ_RBLK(LRQ_INNER) ? (blk_done - LRQ_RBLK(LRQ_INNER)) * 64 : 0;
            uint32_t n_tied = 0, n_acc = 0, n_ex = 0;
            int32_t sc_lo = INT32_MAX, sc_hi = INT32_MIN;
            auto tally_block = [&](int32_t b) {      // the candidates of block b that hold the smallest priority
                const int32_t j = b * 64 + lane;
                bool tied = false; int32_t fb = 0, dr = 0, dq = 0, span_b = kk;
                if (j >= st && j < i0) {
                    const bool near = i - j < LRQ_INNER - 64;
                    const int32_t yj = near ? L.ry[j & M] : (int32_t)a[j].y;
                    if (yj > qi - max_dist && (yj < qi || (yj == qi && j + p_base == 0))) {
                        const double pj = near ? rpri_of(j) : lr_cc_f64(pri + j);
                        if (pj == bp) {
                            tied = true;
                            if (near) { fb = L.rf[j & M]; dr = (int32_t)((uint32_t)xi - L.rx[j & M]); dq = qi - yj; }
                            else { fb = (int32_t)cc_u32(f + j); dr = (int32_t)(xi - a[j].x); dq = qi - yj; span_b = (int32_t)(a[j].y >> 32 & 0xff); }
                        }
                    }
                }
                int32_t ex = 0, wd = 0, sc = 0; bool acc = false;
                if (tied) { sc = fb + lr_sc_simple(dr, dq, span_b, pen_gap, pen_skip, ex, wd); acc = wd <= bw && sc > q_span_i; }
                n_tied += (uint32_t)__popcll(__ballot(tied)); n_acc += (uint32_t)__popcll(__ballot(acc)); n_ex += (uint32_t)__popcll(__ballot(tied && ex));
                if (acc) { sc_lo = sc < sc_lo ? sc : sc_lo; sc_hi = sc > sc_hi ? sc : sc_hi; }
            };
            if (i0 > 0) {
                const int32_t b_top = (i0 - 1) >> 6, b_bot = st >> 6;
                if (b_top >= blk_done) tally_block(b_top);      // the newest, not yet completed block
                // completed blocks, 64 at a time, one lane each: only those whose own minimum is the smallest priority can hold a tied candidate
                for (int32_t bt = (b_top < blk_done ? b_top : blk_done - 1); bt >= b_bot; bt -= 64) {
                    const int32_t b = bt - lane;
                    bool todo = false;
                    if (b >= b_bot) { const double bm = b * 64 >= ring_lo2 ? L.bml[b & 63] : lr_cc_f64(bmin + b); todo = bm <= bp; }
                    uint64_t m = __ballot(todo);
                    while (m) { const int l = __ffsll((unsigned long long)m) - 1; m &= m - 1; tally_block(bt - l); }
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const int32_t a1 = __shfl_xor(sc_lo, o), a2 = __shfl_xor(sc_hi, o); sc_lo = a1 < sc_lo ? a1 : sc_lo; sc_hi = a2 > sc_hi ? a2 : sc_hi; }
            if (n_acc == 0 && (n_ex == 0 || n_ex == n_tied)) { /* (a) */ }
            else if (n_acc == n_tied && sc_lo == sc_hi && n_ex == 0) { tie_pending = true; tie_sc = sc_lo; }      // (b), if the inner scan improves on it
            else ++tie_cnt;
        }
        LRQ_T(3);
        if (bj >= 0) {
            int32_t exact, width, n_skip = 0;
            int32_t fb, dr, dq, span_b;
            if (i - bj < LRQ_INNER - 64) { fb = L.rf[bj & M]; dr = (int32_t)((uint32_t)xi - L.rx[bj & M]); dq = qi - L.ry[bj & M]; span_b = kk; }
            else { fb = (int32_t)cc_u32(f + bj); dr = (int32_t)(xi - a[bj].x); dq = qi - (int32_t)a[bj].y; span_b = (int32_t)(a[bj].y >> 32 & 0xff); }
            int32_t sc = fb + lr_sc_simple(dr, dq, span_b, pen_gap, pen_skip, exact, width);
            if (width <= bw && sc > max_f) { max_f = sc; max_j = bj; }
            if (!exact && n_in > 0 && qi > 0) {
                // (2) the inner window from the largest (y, j) <= (q_i - 1, n) downwards, while y >= q_i - max_dist_inner; entries that have left
                // the window (j < st_inner) are still in the list until it is compacted: they are passed over
                int32_t top = n_in;      // number of elements with y <= q_i - 1, counted down from the top
                for (int32_t c = n_in; c > 0; c -= 64) {
                    const int32_t e = c - 1 - lane;
                    const uint64_t gt = __ballot(e >= 0 && LIY(e) > qi - 1);
                    top -= (int32_t)__popcll(gt);
                    if (__popcll(gt) < 64) break;
                }
                for (int32_t eb = top - 1; eb >= 0; eb -= 64) {
                    const int32_t e = eb - lane;
                    ++d_chunks;
                    const bool inr = e >= 0 && LIY(e) >= qi - max_dist_inner;      // a prefix of the lanes: the list is sorted
                    const uint64_t vm = __ballot(inr);
                    if (vm == 0) break;
                    int32_t j = -1, scj = INT32_MIN, pj = -1;
                    bool has = false;
                    if (inr) {
                        j = LIJ(e);
                        if (j >= st_inner) {
                            int32_t ex2, w2;
                            scj = L.rf[j & M] + lr_sc_simple((int32_t)((uint32_t)xi - L.rx[j & M]), qi - L.ry[j & M], kk, pen_gap, pen_skip, ex2, w2);
                            has = w2 <= bw;
                            pj = L.rp[j & M];
                        }
                    }
                    if (has && pj >= st_inner) L.rt[pj & M] = i;      // a mark only matters on an anchor this scan can still visit
                    __builtin_amdgcn_wave_barrier();
                    const bool is_t = has && L.rt[j & M] == i;
                    const int32_t scv = has ? scj : INT32_MIN;
                    const int32_t incl = wave_scan_max_incl(scv);
                    int32_t excl = wave_shr1(incl, INT32_MIN);
                    if (excl < max_f) excl = max_f;
                    const bool new_max = has && scj > excl;
                    const bool inc_ev = has && !new_max && is_t;
                    const uint64_t inc_m = __ballot(inc_ev), nm_m = __ballot(new_max);
                    const int32_t yl = n_skip + (int32_t)prefix_popc64(inc_m) + (inc_ev ? 1 : 0) - (int32_t)prefix_popc64(nm_m) - (new_max ? 1 : 0);
                    const int32_t mn = wave_scan_min_incl(yl);
                    const int32_t val = yl - (mn < 0 ? mn : 0);
                    const uint64_t brk = __ballot(inc_ev && val > max_skip);
                    const int nv = (int)__popcll(vm);
                    int Lb = brk ? __ffsll((unsigned long long)brk) - 1 : 63;
                    if (Lb > nv - 1) Lb = nv - 1;
                    const int32_t mm = __builtin_amdgcn_readlane(incl, Lb);
                    if (mm > max_f) {
                        max_f = mm;
                        const uint64_t eq = __ballot(lane <= Lb && scv == mm);
                        max_j = __builtin_amdgcn_readlane(j, (int)(__ffsll((unsigned long long)eq) - 1));
                    }
                    n_skip = __builtin_amdgcn_readlane(val, Lb);
                    if ((brk && (__ffsll((unsigned long long)brk) - 1) <= nv - 1) || nv < 64) break;
                }
            }
        }
        LRQ_T(4);
        if (tie_pending && max_f <= tie_sc) ++tie_cnt;      // no better predecessor turned up: max_j is the tree's choice among the tied
        if (lane == 0) { L.rx[i & M] = (uint32_t)xi; L.ry[i & M] = qi; L.rf[i & M] = max_f; L.rp[i & M] = max_j; }
        hi_prev = (uint32_t)(xi >> 32);
        __builtin_amdgcn_wave_barrier();
    }
#undef LIY
#undef LIY_SET
#undef LIJ
    if (TREE && dbg) ++dbg->n_seg;
    if (TREE && T0.bad) ok = false;
    n_tie = tie_cnt;
    if (ok) for (int32_t j = blk_done * 64 + lane; j < n; j += 64) { f[j] = L.rf[j & M]; p[j] = L.rp[j & M] < 0 ? -1 : L.rp[j & M] + p_base; }      // the last, incomplete block(s)
    if (dbg) for (int k2 = 0; k2 < 5; ++k2) dbg->t[4 + k2] += pt[k2];
    if (dbg) { dbg->d[0] += (unsigned long long)n; dbg->d[1] += d_ring; dbg->d[2] += d_oldsteps; dbg->d[3] += d_old; dbg->d[4] += d_nin; dbg->d[5] += d_chunks; dbg->d[6] += 1; if ((unsigned long long)n > dbg->d[7]) dbg->d[7] = (unsigned long long)n; }
    return ok;
}

// ---- the long join of one read by several waves ----------------------------------------------------------------------------------------------
// Where the look-back window is empty - the reference position jumps by more than max_dist, or to another strand / contig - nothing before
// that anchor can be a predecessor and upstream's tree is empty: the stretches between such anchors are independent problems (f, p and the
// tree's answers of one do not depend on another).  A read of 10^5 anchors has some twenty of them, and it is one wave's work for a second
// and more while the launch's other waves have long finished: the wave that owns the read cuts the x-sorted anchors into runs of whole
// stretches (LR_COOP_RUN anchors and more), puts them on a queue of the launch, and every wave of the launch that has nothing else to do
// takes runs off it - its own LDS ring and tree, the owner's arrays in HBM.  The owner works the queue too (whoever's run is at its head), so
// nothing ever waits for another wave to turn up; it goes on to the backtrack when its count of open runs is zero.
#define LR_COOP_RUN 3072       // (defaults of LongParams::coop_run / coop_min; SCRUBBY_HIP_COOP_RUN / _MIN: the tests share joins of a few hundred anchors)
#define LR_COOP_MIN 12288      // reads with fewer anchors in the join stay with their wave
struct LongCoopDesc { const LAnchor *a; int32_t *f, *p; double *pri, *bmin; int32_t remaining, n_tie, fail, pad; };      // one per block of the launch: the read it owns
struct LongCoop {
    uint4 *items;                   // {owner block, first anchor, end, offset into bmin}
    uint32_t *ready;                // per slot: 1 once the item is written (preset to 0)
    uint32_t *q_res, *q_head;       // slots handed out to producers / taken by consumers
    uint32_t *active;               // owners with runs open
    uint32_t *n_reads;              // reads whose join was shared (statistics)
    LongCoopDesc *desc;
    uint32_t cap;
};
// one run off the queue, if its item is there: true = a run was worked on
template <int NR, bool FAT>
__device__ inline bool lr_coop_take(const LongParams &P, const LongCoop &Q, RmqLdsT<NR, FAT> &RL, RqLds TL)
{
    const int32_t lane = (int32_t)al_lane();
    // (every branch on a value all lanes hold - the lanes load the same words, one lane swaps.  A loop with breaks inside `if (lane == 0)` left the
    // code behind it running on that one lane in some of its inlined copies: measured, __ballot(1) == 1 in 3 of 15 calls)
    uint32_t got = ~0u;
    for (int looks = 0; looks < 64; ++looks) {
        const uint32_t cur = (uint32_t)al_b0((int32_t)cc_u32(Q.q_head));
        if (cur >= Q.cap || cur >= (uint32_t)al_b0((int32_t)cc_u32(Q.q_res))) break;
        if (al_b0((int32_t)cc_u32(Q.ready + cur)) == 0) { __builtin_amdgcn_s_sleep(4); continue; }      // handed out, not yet written
        uint32_t old = ~0u;
        if (lane == 0) old = atomicCAS(Q.q_head, cur, cur + 1u);
        if ((uint32_t)al_b0((int32_t)old) == cur) { got = cur; break; }
    }
    if (got == ~0u) return false;
    __threadfence();      // (acquire: the item, the owner's descriptor and anchors)
    const uint64_t i01 = cc_u64(Q.items + got), i23 = cc_u64((const uint64_t *)(Q.items + got) + 1);
    const uint32_t owner = (uint32_t)i01; const int32_t j0 = (int32_t)(i01 >> 32), j1 = (int32_t)(uint32_t)i23; const uint32_t boff = (uint32_t)(i23 >> 32);
    LongCoopDesc *D = Q.desc + owner;
    const LAnchor *a = (const LAnchor *)cc_u64(&D->a);
    int32_t *f = (int32_t *)cc_u64(&D->f), *pp = (int32_t *)cc_u64(&D->p);
    double *pri = (double *)cc_u64(&D->pri), *bmin = (double *)cc_u64(&D->bmin);
    int32_t tie = 0;
    const bool ok = lr_rmq_fill<NR, true, FAT>(P, P.max_gap, P.bw_long, j1 - j0, a + j0, f + j0, pp + j0, pri + j0, bmin + boff, RL, tie, nullptr, TL, j0);
    __threadfence();      // (release: f, p)
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        if (!ok) atomicExch(&D->fail, 1);
        if (tie) atomicAdd(&D->n_tie, tie);
        __threadfence();
        atomicSub(&D->remaining, 1);
    }
    return true;
}
// the owner's side: the join of a[0 .. n) by whoever is free.  tmp: n int32 (the cuts).  false = a run failed (ring / tree outgrown)
template <int NR, bool FAT>
__device__ inline bool lr_coop_fill(const LongParams &P, const LongCoop &Q, uint32_t me, int32_t n, const LAnchor *a, int32_t *f, int32_t *p, double *pri, double *bmin,
                                    int32_t *tmp, RmqLdsT<NR, FAT> &RL, RqLds TL, int32_t &n_tie, int32_t *chk_f = nullptr, int32_t *chk_p = nullptr, double *chk_pri = nullptr, double *chk_bmin = nullptr)
{
    const int32_t lane = (int32_t)al_lane();
    int32_t max_dist = P.max_gap; if (max_dist < P.bw_long) max_dist = P.bw_long;
    // cuts: anchors that start a stretch (the window is empty when the scan reaches them), thinned to runs of LR_COOP_RUN anchors and more
    int32_t n_cut = 0;
    for (int32_t b = 0; b < n; b += 64) {
        const int32_t j = b + lane;
        bool c = false;
        if (j > 0 && j < n) { const uint64_t x1 = a[j].x, x0 = a[j - 1].x; c = (uint32_t)(x1 >> 32) != (uint32_t)(x0 >> 32) || x1 > x0 + (uint64_t)max_dist; }
        const uint64_t m = __ballot(c);
        if (c) tmp[n_cut + (int32_t)prefix_popc64(m)] = j;
        n_cut += (int32_t)__popcll(m);
    }
    lr_sync();
    LongCoopDesc *D = Q.desc + me;
    int32_t n_items = 0;
    uint32_t slot0 = 0;
    if (lane == 0) {
        int32_t k = 0, last = 0;
        for (int32_t c = 0; c < n_cut; ++c) { const int32_t j = (int32_t)cc_u32(tmp + c); if (j - last >= P.coop_run && n - j >= P.coop_run / 2) { tmp[k++] = j; last = j; } }
        n_items = k + 1;
        D->a = a; D->f = f; D->p = p; D->pri = pri; D->bmin = bmin; D->n_tie = 0; D->fail = 0;
        __hip_atomic_store(&D->remaining, n_items, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __threadfence();      // (release: the descriptor and the anchors, before any slot is handed out)
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) { slot0 = atomicAdd(Q.q_res, (uint32_t)n_items); atomicAdd(Q.active, 1u); atomicAdd(Q.n_reads, 1u); }
    n_items = al_b0(n_items); slot0 = (uint32_t)al_b0((int32_t)slot0);
    lr_sync();
    // the items: run i = [cut[i-1], cut[i]) with cut[-1] = 0 and cut[n_items-1] = n.  What finds no slot stays with this wave
    bool ok = true;
    int32_t tie_own = 0, n_local = 0;
    for (int32_t i = 0; i < n_items; ++i) {
        const int32_t j0 = i == 0 ? 0 : (int32_t)cc_u32(tmp + i - 1), j1 = i == n_items - 1 ? n : (int32_t)cc_u32(tmp + i);
        const uint32_t slot = slot0 + (uint32_t)i, boff = (uint32_t)(j0 / 64 + i);
        if (slot < Q.cap) {
            if (lane == 0) {
                uint64_t *it = (uint64_t *)(Q.items + slot);
                it[0] = (uint64_t)me | (uint64_t)(uint32_t)j0 << 32; it[1] = (uint64_t)(uint32_t)j1 | (uint64_t)boff << 32;
                __threadfence();
                __hip_atomic_store(Q.ready + slot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
            int32_t tie = 0;
            if (ok && !lr_rmq_fill<NR, true, FAT>(P, P.max_gap, P.bw_long, j1 - j0, a + j0, f + j0, p + j0, pri + j0, bmin + boff, RL, tie, nullptr, TL, j0)) ok = false;
            tie_own += tie; ++n_local;
        }
    }
    if (n_local && lane == 0) atomicSub(&D->remaining, n_local);
    // work the queue until every run of this read is done
    // (idle looks are bounded - about a minute - like every wait of the stage: a kernel must end whatever happens; a read given up this way is counted unresolved)
    for (unsigned long long idle = 0;;) {
        if (al_b0((int32_t)cc_u32(&D->remaining)) <= 0) break;
        if (lr_coop_take<NR, FAT>(P, Q, RL, TL)) { idle = 0; continue; }
        if (++idle > 30000000ull) { ok = false; break; }
        __builtin_amdgcn_s_sleep(32);
    }
    __threadfence();      // (acquire: what the other waves wrote)
    if (lane == 0) atomicSub(Q.active, 1u);
    n_tie = tie_own + (int32_t)cc_u32(&D->n_tie);
    if (cc_u32(&D->fail)) ok = false;
    lr_sync();
    if (P.coop_check && ok && chk_f) {
        int32_t tie2 = 0;
        const bool ok2 = lr_rmq_fill<NR, true, FAT>(P, P.max_gap, P.bw_long, n, a, chk_f, chk_p, chk_pri, chk_bmin, RL, tie2, nullptr, TL);
        lr_sync();
        int32_t first = n;
        for (int32_t b = 0; b < n; b += 64) { const int32_t j = b + lane; const bool d = j < n && (f[j] != chk_f[j] || p[j] != chk_p[j]); const uint64_t m = __ballot(d); if (m) { first = b + (int32_t)__ffsll((unsigned long long)m) - 1; break; } }
        if (lane == 0 && (first < n || !ok2)) {
            int32_t run0 = 0, k = 0;
            for (int32_t i = 0; i + 1 < n_items; ++i) { const int32_t c = (int32_t)cc_u32(tmp + i); if (c <= first) { run0 = c; k = i + 1; } }
            printf("[coop-check] n %d items %d ok2 %d first diff at %d (run %d starts %d, offset %d): f %d vs %d, p %d vs %d; x %llx y %llx prev x %llx\n", n, n_items, (int)ok2, first, k, run0, first - run0,
                   first < n ? f[first] : 0, first < n ? chk_f[first] : 0, first < n ? p[first] : 0, first < n ? chk_p[first] : 0,
                   first < n ? (unsigned long long)a[first].x : 0ull, first < n ? (unsigned long long)a[first].y : 0ull, first > 0 && first < n ? (unsigned long long)a[first - 1].x : 0ull);
        }
        lr_sync();
    }
    return ok;
}

// ---- mg_lchain_rmq's scoring pass on the literal trees (sh_rmq_tree.h): one lane, statement for statement ---------------------------------
// a[] sorted by x; t[] zeroed by the caller (the skip marks; the backtrack clears them again).  f, p as lr_rmq_fill leaves them.
__device__ __noinline__ bool lr_rmq_fill_tree(const LongParams &P, int32_t max_dist, int32_t bw, int32_t n, const LAnchor *a, int32_t *f, int32_t *p, int32_t *t,
                                              RqNode *pool0, RqNode *pool1, int32_t pool_cap, uint32_t *dbg_code = nullptr)
{
    int32_t okv = 1;
    if (al_lane() == 0) {
        RqTree T0, T1;
        rq_init(T0, pool0, pool_cap); rq_init(T1, pool1, pool_cap);
        const float chn_pen_gap = P.pen_gap, chn_pen_skip = P.pen_skip;
        const int32_t max_chn_skip = P.max_skip, cap_rmq_size = P.rmq_size_cap;
        int32_t max_dist_inner = P.rmq_inner_dist;
        if (max_dist < bw) max_dist = bw;
        if (max_dist_inner < 0) max_dist_inner = 0;
        if (max_dist_inner > max_dist) max_dist_inner = max_dist;
        int32_t i, i0, st = 0, st_inner = 0;
        auto root_size = [](const RqTree &tr) -> int32_t { return rq_size(tr); };
        for (i = i0 = 0; i < n && okv; ++i) {
            int32_t max_j = -1;
            const uint64_t xi = a[i].x; const int32_t yi = (int32_t)a[i].y;
            const int32_t q_span = (int32_t)(a[i].y >> 32 & 0xff);
            int32_t max_f = q_span;
            if (i0 < i && a[i0].x != xi) {      // add in-range anchors
                for (int32_t j = i0; j < i; ++j) {
                    const double pri = -((double)f[j] + 0.5 * (double)chn_pen_gap * (double)((int32_t)a[j].x + (int32_t)a[j].y));
                    for (int k = 0; k < (max_dist_inner > 0 ? 2 : 1); ++k) {
                        RqTree &T = k ? T1 : T0;
                        const int32_t x = rq_alloc(T);
                        if (x == RQ_NIL) { okv = 0; T.bad = 12; break; }
                        rq_node_set(T, x, (int32_t)a[j].y, j, pri);
                        rq_insert(T, x);
                    }
                }
                i0 = i;
            }
            while (st < i && (xi >> 32 != a[st].x >> 32 || xi > a[st].x + (uint64_t)max_dist || root_size(T0) > cap_rmq_size)) {
                const int32_t e = rq_erase(T0, (int32_t)a[st].y, st);
                if (e != RQ_NIL) rq_free(T0, e);
                ++st;
            }
            if (max_dist_inner > 0) {
                while (st_inner < i && (xi >> 32 != a[st_inner].x >> 32 || xi > a[st_inner].x + (uint64_t)max_dist_inner || root_size(T1) > cap_rmq_size)) {
                    const int32_t e = rq_erase(T1, (int32_t)a[st_inner].y, st_inner);
                    if (e != RQ_NIL) rq_free(T1, e);
                    ++st_inner;
                }
            }
            const int32_t q = rq_rmq(T0, yi - max_dist, INT32_MAX, yi, 0);
            if (q != RQ_NIL) {
                int32_t sc, exact, width, n_skip = 0;
                int32_t j = rq_i(T0, q);
                sc = f[j] + lr_sc_simple((int32_t)(xi - a[j].x), yi - (int32_t)a[j].y, (int32_t)(a[j].y >> 32 & 0xff), chn_pen_gap, chn_pen_skip, exact, width);
                if (width <= bw && sc > max_f) { max_f = sc; max_j = j; }
                if (!exact && T1.root != RQ_NIL && yi > 0) {
                    RqItr it;
                    if (rq_itr_find_le(T1, yi - 1, n, it)) {
                        do {
                            const int32_t e = it.stack[it.top];
                            if (rq_y(T1, e) < yi - max_dist_inner) break;
                            j = rq_i(T1, e);
                            int32_t ex2;
                            sc = f[j] + lr_sc_simple((int32_t)(xi - a[j].x), yi - (int32_t)a[j].y, (int32_t)(a[j].y >> 32 & 0xff), chn_pen_gap, chn_pen_skip, ex2, width);
                            if (width <= bw) {
                                if (sc > max_f) {
                                    max_f = sc; max_j = j;
                                    if (n_skip > 0) --n_skip;
                                } else if (t[j] == i) {
                                    if (++n_skip > max_chn_skip) break;
                                }
                                if (p[j] >= 0) t[p[j]] = i;
                            }
                        } while (rq_itr_prev(T1, it));
                    }
                }
            }
            f[i] = max_f; p[i] = max_j;
            if (T0.bad || T1.bad) okv = 0;
        }
        if (!okv && dbg_code) *dbg_code = (uint32_t)(T0.bad ? T0.bad : 20 + T1.bad);
    }
    return al_b0(okv) != 0;
}

// ---- mg_chain_backtrack (lane 0) -----------------------------------------------------------------------------------------
// zc: candidates (f << 32 | index) with f >= min_sc, sorted ascending.  Chains into u (score << 32 | cnt) and v (anchor indices,
// each chain from its end backwards).  Returns n_u; n_v through the reference.
__device__ __noinline__ int32_t lr_backtrack0(const SKey *zc, int32_t n_z, const int32_t *f, const int32_t *p, int32_t *t, int32_t *v, uint64_t *u, uint32_t cap_u,
                                        int32_t min_cnt, int32_t min_sc, int32_t max_drop, int32_t &n_v_out, int32_t &best_out, bool &ovf)
{
    int32_t n_u = 0, n_v = 0, best = 0;
    for (int32_t k = n_z - 1; k >= 0; --k) {
        const int32_t zi = (int32_t)zc[k].v, zf = (int32_t)zc[k].k;
        if (t[zi] != 0) continue;
        // mg_chain_bk_end
        int32_t end_i;
        {
            int32_t i = zi, e = -1, max_i = i, max_s = 0;
            do {
                t[i] = 2;
                e = i = p[i];
                const int32_t s = i < 0 ? zf : zf - f[i];
                if (s > max_s) { max_s = s; max_i = i; }
                else if (max_s - s > max_drop) break;
            } while (i >= 0 && t[i] == 0);
            for (i = zi; i >= 0 && i != e; i = p[i]) t[i] = 0;
            end_i = max_i;
        }
        const int32_t n_v0 = n_v;
        int32_t i;
        for (i = zi; i != end_i; i = p[i]) { v[n_v++] = i; t[i] = 1; }
        const int32_t sc = i < 0 ? zf : zf - f[i];
        if (sc >= min_sc && n_v > n_v0 && n_v - n_v0 >= min_cnt) {
            if ((uint32_t)n_u >= cap_u) { ovf = true; break; }
            u[n_u++] = (uint64_t)(uint32_t)sc << 32 | (uint32_t)(n_v - n_v0);
            if (sc > best) best = sc;
        } else n_v = n_v0;
    }
    n_v_out = n_v; best_out = best;
    return n_u;
}

// mg_chain_backtrack, the whole wave on one control flow: the walks go down the index space a few anchors at a time, so the 64 anchors
// below the current one (f, p, t) sit in registers - one lane each, loaded past the L1 in one go - and a step is a readlane, not a trip
// to memory.  Marks are stored by lane 0 and mirrored in the register block; the block is refetched when a walk leaves it.
struct BtBlock { int32_t lo, f, p, t; };      // lane l holds anchor lo + l
__device__ inline void bt_load(BtBlock &B, int32_t i, int32_t n, const int32_t *f, const int32_t *p, const int32_t *t)
{
    int32_t lo = i - 63; if (lo < 0) lo = 0;
    B.lo = lo;
    const int32_t a = lo + (int32_t)al_lane();
    const bool on = a < n;
    B.f = on ? (int32_t)cc_u32(f + a) : 0; B.p = on ? (int32_t)cc_u32(p + a) : -1; B.t = on ? (int32_t)cc_u32(t + a) : 0;
}
#define BT_IN(B, i) ((i) >= (B).lo && (i) < (B).lo + 64)
#define BT_GET(B, fld, i) __builtin_amdgcn_readlane((B).fld, (i) - (B).lo)
__device__ inline int32_t lr_backtrack_wave(const SKey *zc, int32_t n_z, int32_t n, const int32_t *f, const int32_t *p, int32_t *t, int32_t *v, uint64_t *u, uint32_t cap_u,
                                            int32_t min_cnt, int32_t min_sc, int32_t max_drop, int32_t &n_v_out, int32_t &best_out, bool &ovf)
{
    const int32_t lane = (int32_t)al_lane();
    int32_t n_u = 0, n_v = 0, best = 0;
    BtBlock B; B.lo = -1000; B.f = B.p = B.t = 0;
    auto need = [&](int32_t i) { if (!BT_IN(B, i)) { __builtin_amdgcn_s_waitcnt(0); bt_load(B, i, n, f, p, t); } };
    auto set_t = [&](int32_t i, int32_t val) {      // i is inside the block
        if (lane == 0) t[i] = val;
        if (lane == i - B.lo) B.t = val;
    };
    for (int32_t k = n_z - 1; k >= 0; --k) {
        const int32_t zi = (int32_t)zc[k].v, zf = (int32_t)zc[k].k;
        need(zi);
        if (BT_GET(B, t, zi) != 0) continue;
        // mg_chain_bk_end
        int32_t end_i;
        {
            int32_t i = zi, e = -1, max_i = i, max_s = 0;
            for (;;) {
                need(i);
                set_t(i, 2);
                e = i = BT_GET(B, p, i);
                int32_t s2 = zf;
                if (i >= 0) { need(i); s2 = zf - BT_GET(B, f, i); }
                if (s2 > max_s) { max_s = s2; max_i = i; }
                else if (max_s - s2 > max_drop) break;
                if (!(i >= 0 && BT_GET(B, t, i) == 0)) break;
            }
            for (i = zi; i >= 0 && i != e; ) { need(i); set_t(i, 0); i = BT_GET(B, p, i); }
            end_i = max_i;
        }
        const int32_t n_v0 = n_v;
        int32_t i;
        for (i = zi; i != end_i; ) { need(i); if (lane == 0) v[n_v] = i; ++n_v; set_t(i, 1); i = BT_GET(B, p, i); }
        int32_t sc = zf;
        if (i >= 0) { need(i); sc = zf - BT_GET(B, f, i); }
        if (sc >= min_sc && n_v > n_v0 && n_v - n_v0 >= min_cnt) {
            if ((uint32_t)n_u >= cap_u) { ovf = true; break; }
            if (lane == 0) u[n_u] = (uint64_t)(uint32_t)sc << 32 | (uint32_t)(n_v - n_v0);
            ++n_u;
            if (sc > best) best = sc;
        } else n_v = n_v0;
    }
    n_v_out = n_v; best_out = best;
    return n_u;
}
#undef BT_IN
#undef BT_GET

// ---- region bookkeeping on lane 0 (hit.c) ------------------------------------------------------------------------------------
__device__ inline void lr_set_parent0(float mask_level, int32_t n, LReg *r, uint64_t *cov, int32_t *w)
{
    if (n <= 0) return;
    for (int32_t i = 0; i < n; ++i) r[i].id = i;
    w[0] = 0; r[0].parent = 0;
    int32_t k = 1;
    for (int32_t i = 1; i < n; ++i) {
        LReg *ri = &r[i];
        const int32_t si = ri->qs, ei = ri->qe;
        int32_t n_cov = 0, uncov_len = 0, j;
        for (j = 0; j < k; ++j) {
            const LReg *rp = &r[w[j]];
            int32_t sj = rp->qs, ej = rp->qe;
            if (ej <= si || sj >= ei) continue;
            if (sj < si) sj = si;
            if (ej > ei) ej = ei;
            cov[n_cov++] = (uint64_t)(uint32_t)sj << 32 | (uint32_t)ej;
        }
        if (n_cov > 0) {
            for (int32_t a1 = 1; a1 < n_cov; ++a1) { const uint64_t tt = cov[a1]; int32_t b1 = a1; for (; b1 > 0 && cov[b1 - 1] > tt; --b1) cov[b1] = cov[b1 - 1]; cov[b1] = tt; }
            int32_t x = si;
            for (j = 0; j < n_cov; ++j) {
                if ((int32_t)(cov[j] >> 32) > x) uncov_len += (int32_t)(cov[j] >> 32) - x;
                x = (int32_t)cov[j] > x ? (int32_t)cov[j] : x;
            }
            if (ei > x) uncov_len += ei - x;
            for (j = 0; j < k; ++j) {
                LReg *rp = &r[w[j]];
                const int32_t sj = rp->qs, ej = rp->qe;
                if (ej <= si || sj >= ei) continue;
                const int32_t mn = ej - sj < ei - si ? ej - sj : ei - si, mx = ej - sj > ei - si ? ej - sj : ei - si;
                const int32_t ol = si < sj ? (ei < sj ? 0 : ei < ej ? ei - sj : ej - sj) : (ej < si ? 0 : ej < ei ? ej - si : ei - si);
                if ((float)ol / mn - (float)uncov_len / mx > mask_level) {
                    ri->parent = rp->parent;
                    rp->subsc = rp->subsc > ri->score ? rp->subsc : ri->score;
                    if (ri->cnt >= rp->cnt) ++rp->n_sub;
                    break;
                }
            }
        } else j = k;
        if (j == k) { w[k++] = i; ri->parent = i; ri->n_sub = 0; }
    }
}

__device__ inline void lr_sync_regs0(int32_t n_regs, LReg *regs, int32_t *tmp)
{   // mm_sync_regs; tmp: max id + 1 ints
    if (n_regs <= 0) return;
    int32_t max_id = -1;
    for (int32_t i = 0; i < n_regs; ++i) max_id = max_id > regs[i].id ? max_id : regs[i].id;
    for (int32_t i = 0; i <= max_id; ++i) tmp[i] = -1;
    for (int32_t i = 0; i < n_regs; ++i) if (regs[i].id >= 0) tmp[regs[i].id] = i;
    for (int32_t i = 0; i < n_regs; ++i) {
        LReg *r = &regs[i];
        r->id = i;
        if (r->parent == LR_PARENT_TMP_PRI) r->parent = i;
        else if (r->parent >= 0 && tmp[r->parent] >= 0) r->parent = tmp[r->parent];
        else r->parent = LR_PARENT_UNSET;
    }
}

__device__ inline int32_t lr_select_sub0(float pri_ratio, int32_t min_diff, int32_t best_n, int32_t min_strand_sc, int32_t n, LReg *r, int32_t *tmp)
{   // in place like upstream: r[p] is read after earlier regions moved up
    if (!(pri_ratio > 0.0f) || n <= 0) return n;
    int32_t k = 0, n_2nd = 0;
    for (int32_t i = 0; i < n; ++i) {
        const int32_t p = r[i].parent;
        if (p == i || r[i].inv) r[k++] = r[i];
        else if ((r[i].score >= r[p].score * pri_ratio || r[i].score + min_diff >= r[p].score) && n_2nd < best_n) {
            if (!(r[i].qs == r[p].qs && r[i].qe == r[p].qe && r[i].rid == r[p].rid && r[i].rs == r[p].rs && r[i].re == r[p].re)) { r[k++] = r[i]; ++n_2nd; }
        } else if (n_2nd < best_n && r[i].score > min_strand_sc && r[p].rev != r[i].rev && r[p].rid == r[i].rid && r[i].rs < r[p].re && r[i].re > r[p].rs) {
            r[i].strand_retained = 1;
            r[k++] = r[i]; ++n_2nd;
        }
    }
    if (k != n) lr_sync_regs0(k, r, tmp);
    return k;
}

__device__ inline int32_t lr_get_for_qpos(int32_t qlen, const LAnchor &a)
{
    int32_t x = (int32_t)a.y;
    const int32_t q_span = (int32_t)(a.y >> 32 & 0xff);
    if (a.x >> 63) x = qlen - 1 - (x + 1 - q_span);
    return x;
}

__device__ inline void lr_est_err0(int32_t qlen, int32_t n_regs, LReg *regs, const LAnchor *a, int32_t n, const uint64_t *mini_pos, const uint64_t *cstart)
{
    if (n == 0) return;
    uint64_t sum_k = 0;
    for (int32_t i = 0; i < n; ++i) sum_k += mini_pos[i] >> 32 & 0xff;
    const float avg_k = (float)sum_k / n;
    for (int32_t i = 0; i < n_regs; ++i) {
        LReg *r = &regs[i];
        r->div = -1.0f;
        if (r->cnt == 0) continue;
        int32_t st;
        {   // get_mini_idx
            const int32_t x = lr_get_for_qpos(qlen, r->rev ? a[r->as + r->cnt - 1] : a[r->as]);
            int32_t L = 0, R = n - 1;
            st = -1;
            while (L <= R) {
                const int32_t m = (int32_t)(((uint64_t)L + (uint64_t)R) >> 1);
                const int32_t y = (int32_t)mini_pos[m];
                if (y < x) L = m + 1;
                else if (y > x) R = m - 1;
                else { st = m; break; }
            }
        }
        if (st < 0) continue;
        int32_t en = st, k, j, n_match;
        const int32_t l_ref = (int32_t)(cstart[r->rid + 1] - cstart[r->rid]);
        for (k = 1, j = st + 1, n_match = 1; j < n && k < r->cnt; ++j) {
            const int32_t q = lr_get_for_qpos(qlen, r->rev ? a[r->as + r->cnt - 1 - k] : a[r->as + k]);
            if (q == (int32_t)mini_pos[j]) { ++k; en = j; ++n_match; }
        }
        int32_t n_tot = en - st + 1;
        if (r->qs > avg_k && r->rs > avg_k) ++n_tot;
        if (qlen - r->qs > avg_k && l_ref - r->re > avg_k) ++n_tot;      /* qs, not qe: upstream's mm_est_err reads `qlen - r->qs` here (hit.c, v2.28 as recalled by builder and reviewer alike) */
        r->div = n_match >= n_tot ? 0.0f : (float)(1.0 - pow((double)n_match / n_tot, 1.0 / avg_k));
    }
}

__device__ inline int32_t lr_filter_strand0(int32_t n_regs, LReg *r)
{
    int32_t k = 0;
    for (int32_t i = 0; i < n_regs; ++i) {
        const int32_t p = r[i].parent;
        if (!r[i].strand_retained || r[i].div < r[p].div * 5.0f || r[i].div < 0.01f) {
            if (k < i) r[k++] = r[i];
            else ++k;
        }
    }
    return k;
}

// ---- mm_align1's anchor filters (lane 0) ---------------------------------------------------------------------------------------
__device__ inline void lr_fix_bad_ends0(const LReg &r, const LAnchor *a, int32_t bw, int32_t min_match, int32_t &as_o, int32_t &cnt_o)
{
    int32_t as = r.as, cnt = r.cnt;
    if (r.cnt >= 3) {
        int32_t i, l, m;
        m = l = (int32_t)(a[r.as].y >> 32 & 0xff);
        for (i = r.as + 1; i < r.as + r.cnt - 1; ++i) {
            const int32_t q_span = (int32_t)(a[i].y >> 32 & 0xff);
            if (a[i].y & LY_LONG_JOIN) break;
            const int32_t lr = (int32_t)a[i].x - (int32_t)a[i - 1].x, lq = (int32_t)a[i].y - (int32_t)a[i - 1].y;
            const int32_t mn = lr < lq ? lr : lq, mx = lr > lq ? lr : lq;
            if (mx - mn > l >> 1) as = i;
            l += mn;
            m += mn < q_span ? mn : q_span;
            if (l >= bw << 1 || (m >= min_match && m >= bw) || m >= r.mlen >> 1) break;
        }
        cnt = r.as + r.cnt - as;
        m = l = (int32_t)(a[r.as + r.cnt - 1].y >> 32 & 0xff);
        for (i = r.as + r.cnt - 2; i > as; --i) {
            const int32_t q_span = (int32_t)(a[i + 1].y >> 32 & 0xff);
            if (a[i + 1].y & LY_LONG_JOIN) break;
            const int32_t lr = (int32_t)a[i + 1].x - (int32_t)a[i].x, lq = (int32_t)a[i + 1].y - (int32_t)a[i].y;
            const int32_t mn = lr < lq ? lr : lq, mx = lr > lq ? lr : lq;
            if (mx - mn > l >> 1) cnt = i + 1 - as;
            l += mn;
            m += mn < q_span ? mn : q_span;
            if (l >= bw << 1 || (m >= min_match && m >= bw) || m >= r.mlen >> 1) break;
        }
    }
    as_o = as; cnt_o = cnt;
}

__device__ inline int32_t lr_gap_at(const LAnchor *a, int32_t i)
{
    return (int32_t)((uint32_t)a[i].y - (uint32_t)a[i - 1].y - ((uint32_t)a[i].x - (uint32_t)a[i - 1].x));
}

__device__ inline int32_t lr_collect_long_gaps0(int32_t as1, int32_t cnt1, const LAnchor *a, int32_t min_gap, int32_t *K)
{
    int32_t n = 0;
    for (int32_t i = 1; i < cnt1; ++i) {
        const int32_t gap = lr_gap_at(a, as1 + i);
        if (gap < -min_gap || gap > min_gap) K[n++] = i;
    }
    return n <= 1 ? 0 : n;
}

// collect_long_gaps by the whole wave: a region of a long read has ~1000 anchors, and one lane walking them through HBM scratch (twice per
// region: two thresholds) was most of a probe's time.  K receives the indices in rising order; returns their number (0 when it is 1: upstream).
__device__ inline int32_t lr_collect_long_gaps_wave(int32_t as1, int32_t cnt1, const LAnchor *a, int32_t min_gap, int32_t *K)
{
    const int32_t lane = (int32_t)al_lane();
    int32_t n = 0;
    for (int32_t i0 = 1; i0 < cnt1; i0 += 64) {
        const int32_t i = i0 + lane;
        bool g = false;
        if (i < cnt1) { const int32_t gap = lr_gap_at(a, as1 + i); g = gap < -min_gap || gap > min_gap; }
        const uint64_t m = __ballot(g);
        if (g) K[n + (int32_t)prefix_popc64(m)] = i;
        n += (int32_t)__popcll(m);
    }
    lr_sync();
    return n <= 1 ? 0 : n;
}

// n: what lr_collect_long_gaps_wave returned for (as1, cnt1, min_gap) - or -1: collect here, on this lane
__device__ inline void lr_filter_bad_seeds0(int32_t as1, int32_t cnt1, LAnchor *a, int32_t min_gap, int32_t diff_thres, int32_t max_ext_len, int32_t max_ext_cnt, int32_t *K, int32_t n_pre = -1)
{
    const int32_t n = n_pre >= 0 ? n_pre : lr_collect_long_gaps0(as1, cnt1, a, min_gap, K);
    if (n == 0) return;
    int32_t max = 0, max_st = -1, max_en = -1;
    for (int32_t k = 0;; ++k) {
        int32_t gap, l, n_ins = 0, n_del = 0, qs, rs, max_diff = 0, max_diff_l = -1, i;
        if (k == n || k >= max_en) {
            if (max_en > 0) for (i = K[max_st]; i < K[max_en]; ++i) a[as1 + i].y |= LY_IGNORE;
            max = 0; max_st = max_en = -1;
            if (k == n) break;
        }
        i = K[k];
        gap = lr_gap_at(a, as1 + i);
        if (gap > 0) n_ins += gap; else n_del += -gap;
        qs = (int32_t)a[as1 + i - 1].y;
        rs = (int32_t)a[as1 + i - 1].x;
        for (l = k + 1; l < n && l <= k + max_ext_cnt; ++l) {
            const int32_t j = K[l];
            if ((int32_t)a[as1 + j].y - qs > max_ext_len || (int32_t)a[as1 + j].x - rs > max_ext_len) break;
            gap = lr_gap_at(a, as1 + j);
            if (gap > 0) n_ins += gap; else n_del += -gap;
            const int32_t ad = n_ins - n_del < 0 ? n_del - n_ins : n_ins - n_del;
            const int32_t diff = n_ins + n_del - ad;
            if (max_diff < diff) { max_diff = diff; max_diff_l = l; }
        }
        if (max_diff > diff_thres && max_diff > max) { max = max_diff; max_st = k; max_en = max_diff_l; }
    }
}

__device__ inline void lr_filter_bad_seeds_alt0(int32_t as1, int32_t cnt1, LAnchor *a, int32_t min_gap, int32_t max_ext, int32_t *K, int32_t n_pre = -1)
{
    const int32_t n = n_pre >= 0 ? n_pre : lr_collect_long_gaps0(as1, cnt1, a, min_gap, K);
    if (n == 0) return;
    for (int32_t k = 0; k < n;) {
        const int32_t i = K[k];
        int32_t l;
        int32_t gap1 = lr_gap_at(a, as1 + i);
        int32_t re1 = (int32_t)a[as1 + i].x, qe1 = (int32_t)a[as1 + i].y;
        gap1 = gap1 > 0 ? gap1 : -gap1;
        for (l = k + 1; l < n; ++l) {
            const int32_t j = K[l];
            if ((int32_t)a[as1 + j].y - qe1 > max_ext || (int32_t)a[as1 + j].x - re1 > max_ext) break;
            int32_t gap2 = lr_gap_at(a, as1 + j);
            const int32_t q_span_pre = (int32_t)(a[as1 + j - 1].y >> 32 & 0xff);
            const int32_t rs2 = (int32_t)a[as1 + j - 1].x + q_span_pre, qs2 = (int32_t)a[as1 + j - 1].y + q_span_pre;
            const int32_t m = rs2 - re1 < qs2 - qe1 ? rs2 - re1 : qs2 - qe1;
            gap2 = gap2 > 0 ? gap2 : -gap2;
            if (m > gap1 + gap2) break;
            re1 = (int32_t)a[as1 + j].x; qe1 = (int32_t)a[as1 + j].y;
            gap1 = gap2;
        }
        if (l > k + 1) {
            const int32_t end = K[l - 1];
            for (int32_t j = K[k]; j < end; ++j) a[as1 + j].y |= LY_IGNORE;
            a[as1 + end].y |= LY_LONG_JOIN;
        }
        k = l;
    }
}

// ---- ksw_ll_i16 on one wave: local alignment, rows of the target one after the other, the columns across the lanes ----------------------
// H(i,j) = max(0, H(i-1,j-1) + s, E(i,j), F(i,j)); E(i+1,j) = max(0, E(i,j) - e, H(i,j) - (o + e)); F along a row likewise.  F is the
// only dependency inside a row: F(j) = max(0, max_{j' < j}(H'(j') - (o + e) - (j - 1 - j') * e)) with H' = max(diag, E) - a prefix
// maximum of H'(j') + j' * e.  The query is padded to a multiple of 8 with columns scoring 0, as the striped profile pads it; te / qe
// are the LAST row reaching the maximum and the LAST column (in striped memory order) holding it there (oracle mma_ksw_ll).
__device__ __noinline__ int32_t lr_ksw_ll_wave(int32_t qlen, const uint8_t *query, int32_t tlen, const uint8_t *target, int8_t sc_mch, int8_t sc_mis, int8_t sc_amb,
                                         int32_t gapo, int32_t gape, int32_t &qe, int32_t &te, int32_t *H0, int32_t *E, int32_t *Hmax)
{
    const int32_t lane = (int32_t)al_lane();
    const int32_t slen = (qlen + 7) / 8, qp = slen * 8, gapoe = gapo + gape;
    qe = te = -1;
    if (qlen <= 0) return 0;
    int32_t *H1 = H0 + qp + 8;
    for (int32_t j = lane; j < qp; j += 64) { H0[j] = 0; E[j] = 0; Hmax[j] = 0; }
    lr_sync();
    int32_t gmax = 0;
    for (int32_t i = 0; i < tlen; ++i) {
        const int32_t tb = target[i];
        int32_t imax = 0, fcarry = INT32_MIN / 2;      // max over earlier chunks of (H'(j') - gapoe + j' * gape), "minus infinity" = F of 0
        for (int32_t jb = 0; jb < qp; jb += 64) {
            const int32_t j = jb + lane;
            const bool on = j < qp;
            int32_t hp = 0, fsrc = INT32_MIN / 2;
            if (on) {
                const int32_t hd = j > 0 ? H0[j - 1] : 0;
                int32_t s = 0;
                if (j < qlen) { const int32_t qb = query[j]; s = (tb > 3 || qb > 3) ? sc_amb : (tb == qb ? sc_mch : sc_mis); }
                const int32_t e = E[j];
                hp = hd + s; hp = hp > e ? hp : e;      // H' >= 0 because E >= 0
                fsrc = hp - gapoe + j * gape;
            }
            // exclusive prefix maximum of fsrc over the lanes, seeded with the carry
            int32_t incl = wave_scan_max_incl(fsrc);
            int32_t excl = wave_shr1(incl, INT32_MIN / 2);
            if (excl < fcarry) excl = fcarry;
            fcarry = __builtin_amdgcn_readlane(incl, 63) > fcarry ? __builtin_amdgcn_readlane(incl, 63) : fcarry;
            if (on) {
                int32_t fj = excl - (j - 1) * gape;      // F(j) = max_{j'<j}(H'(j') - gapoe - (j - 1 - j') * gape)
                if (fj < 0) fj = 0;
                const int32_t h = hp > fj ? hp : fj;
                imax = imax > h ? imax : h;
                H1[j] = h;
                int32_t tt = h - gapoe; if (tt < 0) tt = 0;
                int32_t e = E[j] - gape; if (e < 0) e = 0;
                E[j] = e > tt ? e : tt;
            }
        }
        imax = wave_all_max(imax);
        lr_sync();
        if (imax >= gmax) {
            gmax = imax; te = i;
            for (int32_t j = lane; j < qp; j += 64) Hmax[j] = H1[j];
        }
        { int32_t *S = H1; H1 = H0; H0 = S; }
        lr_sync();
    }
    {   // the last hit in striped memory order: memory index m <-> position m / 8 + m % 8 * slen
        int32_t best_m = -1;
        for (int32_t m = lane; m < qp; m += 64) { const int32_t pos = m / 8 + m % 8 * slen; if (Hmax[pos] == gmax) best_m = m; }
        best_m = wave_all_max(best_m);
        if (best_m >= 0) qe = best_m / 8 + best_m % 8 * slen;
    }
    return gmax;
}

// ---- mm_align1 without MM_F_SR (align.c), one region -------------------------------------------------------------------------------
struct LongIn {
    AlignIn in;                      // reference, reads, the chains handed over
    const uint4 *rec; const uint32_t *k1info; const unsigned long long *seed_off; uint32_t seed_cap;      // the reads' seed records
};

struct LongCtx {
    const LongParams *P; const AlignParams *AP; const LongIn *I; LongWs *W; AlignLds *Ls; AlignScratch A; LongClk *clk;
    int32_t qlen; uint32_t read;
    int8_t sc_mch, sc_mis, sc_amb, sc_N;
    int32_t probe_why;               // why lr_probe_region last gave up (statistics)
    bool need_big;                   // an alignment does not fit this wave's direction-byte buffer: the read goes to the large-scratch pass
    uint32_t err;                    // a capacity of the working memory was exceeded (code)
    const LongCoop *coop = nullptr; uint32_t coop_me = 0;      // the launch shares the long join of its largest reads among its waves (lr_coop_fill)
};

// mm_align_pair: false = the caller must stop (need_big / err set)
__device__ __noinline__ bool lr_align_pair(LongCtx &C, int32_t qlen, const uint8_t *qseq, int32_t tlen, const uint8_t *tseq, int32_t w, int32_t end_bonus, int32_t zdrop, int32_t flag, Ez &ez)
{
    const LongParams &P = *C.P;
    if ((long long)tlen * qlen > 100000000ll) { ez_reset(ez); ez.zdropped = 1; return true; }      // max_sw_mat
    if (qlen <= 0 || tlen <= 0) { ez_reset(ez); return true; }
    const int32_t ww = w < 0 ? (tlen > qlen ? tlen : qlen) : w;
    int32_t nc = qlen < tlen ? qlen : tlen;
    nc = (((nc < ww + 1 ? nc : ww + 1) + 15) / 16 + 1) * 16;
    const unsigned long long p_need = (unsigned long long)(qlen + tlen - 1) * (unsigned long long)nc;
    if ((uint32_t)tlen + 16 > C.W->cap_k || (uint32_t)qlen + 16 > C.W->cap_k) { C.err = 11; return false; }
    if (p_need > C.W->cap_p) { C.need_big = true; return false; }
    const int32_t T16 = (tlen + 15) / 16 * 16, Q16 = (qlen + 15) / 16 * 16;
    const bool mem_lds = T16 <= AL_T16 && Q16 <= AL_Q16;
    lr_tick(C.clk, 11);
    if (mem_lds) ksw_extd2_core<false, true>(qlen, qseq, false, tlen, tseq, false, C.sc_mch, C.sc_mis, C.sc_N, P.q, P.e, P.q2, P.e2, w, zdrop, end_bonus, flag, ez, C.W->ez_cigar, C.A, *C.Ls);
    else ksw_extd2_core<true, true>(qlen, qseq, false, tlen, tseq, false, C.sc_mch, C.sc_mis, C.sc_N, P.q, P.e, P.q2, P.e2, w, zdrop, end_bonus, flag, ez, C.W->ez_cigar, C.A, *C.Ls);
    lr_sync();
    lr_tick(C.clk, 8);
    return true;
}

__device__ inline void lr_getseq(const LongCtx &C, int32_t rid, int32_t st, int32_t en, uint8_t *out)
{
    getseq_wave(C.I->in, rid, st, en, out);
    lr_sync();
}

__device__ inline void lr_seq_rev(int32_t len, uint8_t *seq)
{
    lr_sync();
    seq_rev_wave(len, seq, false);
    lr_sync();
}

// mm_test_zdrop, long-read form (uniform): 0, 1 (z-drop), 2 (z-drop over a stretch that aligns to its own reverse complement)
__device__ inline int32_t lr_test_zdrop(LongCtx &C, const uint8_t *qseq, const uint8_t *tseq, int32_t n_cigar, const uint32_t *cigar)
{
    const LongParams &P = *C.P;
    int32_t res[6] = {0, 0, 0, 0, 0, 0};      // max_zdrop, pos[0][0], pos[0][1], pos[1][0], pos[1][1]
    if (al_lane() == 0) {
        int32_t score = 0, max = INT32_MIN, max_i = -1, max_j = -1, i = 0, j = 0, max_zdrop = 0;
        int32_t p00 = -1, p01 = -1, p10 = -1, p11 = -1;
        auto upd = [&](int32_t sc, int32_t ii, int32_t jj) {
            if (sc < max) {
                const int32_t li = ii - max_i, lj = jj - max_j, diff = li > lj ? li - lj : lj - li, z = max - sc - diff * P.e;
                if (z > max_zdrop) { max_zdrop = z; p00 = max_i; p01 = max_j; p10 = ii; p11 = jj; }
            } else { max = sc; max_i = ii; max_j = jj; }
        };
        for (int32_t k = 0; k < n_cigar; ++k) {
            const uint32_t op = cigar[k] & 0xf, len = cigar[k] >> 4;
            if (op == 0) {
                for (uint32_t l = 0; l < len; ++l) {
                    const int32_t ct = tseq[i + (int32_t)l], cq = qseq[j + (int32_t)l];
                    score += (ct > 3 || cq > 3) ? C.sc_amb : (ct == cq ? C.sc_mch : C.sc_mis);
                    upd(score, i + (int32_t)l, j + (int32_t)l);
                }
                i += (int32_t)len; j += (int32_t)len;
            } else if (op == 1 || op == 2) {
                score -= P.q + P.e * (int32_t)len;
                if (op == 1) j += (int32_t)len; else i += (int32_t)len;
                upd(score, i, j);
            }
        }
        res[0] = max_zdrop; res[1] = p00; res[2] = p01; res[3] = p10; res[4] = p11;
    }
    const int32_t max_zdrop = al_b0(res[0]), p00 = al_b0(res[1]), p01 = al_b0(res[2]), p10 = al_b0(res[3]), p11 = al_b0(res[4]);
    const int32_t q_len = p11 - p01, t_len = p10 - p00;
    if (max_zdrop > P.zdrop_inv && q_len < P.max_gap && t_len < P.max_gap) {
        // the reverse complement of the dropped stretch of the query, against the same stretch of the target
        if ((uint32_t)q_len + 16 > C.W->cap_k || (uint32_t)t_len + 16 > C.W->cap_k) { C.err = 12; return 0; }
        uint8_t *q2 = C.W->kmem;      // free between alignments
        for (int32_t i = (int32_t)al_lane(); i < q_len; i += 64) { const int32_t c = qseq[p11 - i - 1]; q2[i] = (uint8_t)(c >= 4 ? 4 : 3 - c); }
        lr_sync();
        int32_t qo, to;
        const int32_t score = lr_ksw_ll_wave(q_len, q2, t_len, tseq + p00, C.sc_mch, C.sc_mis, C.sc_amb, P.q, P.e, qo, to, C.W->lH, C.W->lE, C.W->lHmax);
        lr_sync();
        if (score >= P.min_sc * P.a && score >= P.min_dp_max) return 2;
    }
    return max_zdrop > P.zdrop ? 1 : 0;
}

// One region.  r / r2 live in registers (uniform); the caller stores them.  a = the squeezed anchors, n_a their number.
// flag_only: stop as soon as the region is known to survive mm_filter_regs is NOT done here: this is the complete procedure.
__device__ __noinline__ bool lr_align1(LongCtx &C, LReg &r, LReg &r2, LAnchor *a, int32_t n_a, int32_t pre_as1 = -1, int32_t pre_cnt1 = 0)
{
    const LongParams &P = *C.P;
    LongWs &W = *C.W;
    const uint32_t lane = al_lane();
    const int32_t qlen = C.qlen, hk = P.k >> 1;
    r2.cnt = 0;
    if (r.cnt == 0) return true;
    const int32_t rid = (int32_t)(a[r.as].x << 1 >> 33), rev = (int32_t)(a[r.as].x >> 63);
    const int32_t clen = (int32_t)(C.I->in.cstart[rid + 1] - C.I->in.cstart[rid]);
    const int32_t bw = (int32_t)(P.bw * 1.5 + 1.);
    int32_t bw_long = (int32_t)(P.bw_long * 1.5 + 1.);
    if (bw_long < bw) bw_long = bw;
    int32_t as1, cnt1;
    lr_tick(C.clk, 6);
    if (pre_as1 >= 0) { as1 = pre_as1; cnt1 = pre_cnt1; }      // a probe of this region ran first: mm_fix_bad_ends must not see the flags its seed filters left
    else lr_fix_bad_ends0(r, a, P.bw, P.min_sc * 2, as1, cnt1);
    {
        const int32_t n1 = lr_collect_long_gaps_wave(as1, cnt1, a, 10, W.K);
        if (lane == 0) lr_filter_bad_seeds0(as1, cnt1, a, 10, 40, P.max_gap >> 1, 10, W.K, n1);
        lr_sync();
        const int32_t n2 = lr_collect_long_gaps_wave(as1, cnt1, a, 30, W.K);
        if (lane == 0) lr_filter_bad_seeds_alt0(as1, cnt1, a, 30, P.max_gap >> 1, W.K, n2);
    }
    lr_sync();
    int32_t rs = (int32_t)a[as1].x - hk, qs = (int32_t)a[as1].y - hk;
    int32_t re = (int32_t)a[as1 + cnt1 - 1].x - hk, qe = (int32_t)a[as1 + cnt1 - 1].y - hk;
    int32_t rs0, qs0, re0, qe0, rs1, qs1, re1, qe1, l, i;
    rs0 = (int32_t)a[r.as].x + 1 - (int32_t)(a[r.as].y >> 32 & 0xff);
    qs0 = (int32_t)a[r.as].y + 1 - (int32_t)(a[r.as].y >> 32 & 0xff);
    if (rs0 < 0) rs0 = 0;
    rs1 = qs1 = 0;
    for (i = r.as - 1, l = 0; i >= 0 && a[i].x >> 32 == a[r.as].x >> 32; --i) {
        const int32_t x = (int32_t)a[i].x + 1 - (int32_t)(a[i].y >> 32 & 0xff), y = (int32_t)a[i].y + 1 - (int32_t)(a[i].y >> 32 & 0xff);
        if (x < rs0 && y < qs0) {
            if (++l > P.min_cnt) {
                l = rs0 - x > qs0 - y ? rs0 - x : qs0 - y;
                rs1 = rs0 - l; qs1 = qs0 - l;
                if (rs1 < 0) rs1 = 0;
                break;
            }
        }
    }
    if (qs > 0 && rs > 0) {
        l = qs < P.max_gap ? qs : P.max_gap;
        qs1 = qs1 > qs - l ? qs1 : qs - l;
        qs0 = qs0 < qs1 ? qs0 : qs1;
        l += l * P.a > P.q ? (l * P.a - P.q) / P.e : 0;
        l = l < P.max_gap ? l : P.max_gap;
        l = l < rs ? l : rs;
        rs1 = rs1 > rs - l ? rs1 : rs - l;
        rs0 = rs0 < rs1 ? rs0 : rs1;
        rs0 = rs0 < rs ? rs0 : rs;
    } else { rs0 = rs; qs0 = qs; }
    re0 = (int32_t)a[r.as + r.cnt - 1].x + 1;
    qe0 = (int32_t)a[r.as + r.cnt - 1].y + 1;
    re1 = clen; qe1 = qlen;
    for (i = r.as + r.cnt, l = 0; i < n_a && a[i].x >> 32 == a[r.as].x >> 32; ++i) {
        const int32_t x = (int32_t)a[i].x + 1, y = (int32_t)a[i].y + 1;
        if (x > re0 && y > qe0) {
            if (++l > P.min_cnt) {
                l = x - re0 > y - qe0 ? x - re0 : y - qe0;
                re1 = re0 + l; qe1 = qe0 + l;
                break;
            }
        }
    }
    if (qe < qlen && re < clen) {
        l = qlen - qe < P.max_gap ? qlen - qe : P.max_gap;
        qe1 = qe1 < qe + l ? qe1 : qe + l;
        qe0 = qe0 > qe1 ? qe0 : qe1;
        l += l * P.a > P.q ? (l * P.a - P.q) / P.e : 0;
        l = l < P.max_gap ? l : P.max_gap;
        l = l < clen - re ? l : clen - re;
        re1 = re1 < re + l ? re1 : re + l;
        re0 = re0 > re1 ? re0 : re1;
    } else { re0 = re; qe0 = qe; }
    if ((uint32_t)(re0 - rs0 > 0 ? re0 - rs0 : 0) + 16 > W.cap_t) { C.err = 13; return false; }

    lr_tick(C.clk, 7);
    uint8_t *qrow = W.qseq + (rev ? qlen : 0), *tseq = W.tseq;
    uint32_t *rc = W.r_cigar, *ezc = W.ez_cigar;
    int32_t rn = 0, dropped = 0;
    Ez ez;
    r.has_p = 0;
    if (qs > 0 && rs > 0) {       // left extension
        lr_getseq(C, rid, rs0, rs, tseq);
        lr_seq_rev(qs - qs0, qrow + qs0);
        lr_seq_rev(rs - rs0, tseq);
        if (!lr_align_pair(C, qs - qs0, qrow + qs0, rs - rs0, tseq, bw, P.end_bonus, r.split_inv ? P.zdrop_inv : P.zdrop, EZ_EXTZ_ONLY | EZ_RIGHT | EZ_REV_CIGAR, ez)) {
            lr_seq_rev(qs - qs0, qrow + qs0);      // leave the query as it was for the pass that takes the read over
            return false;
        }
        if (ez.n_cigar > 0) { if (lane == 0) append_cigar0(rc, rn, ez.n_cigar, ezc); r.has_p = 1; }
        rs1 = rs - (ez.reach_end ? ez.mqe_t + 1 : ez.max_t + 1);
        qs1 = qs - (ez.reach_end ? qs - qs0 : ez.max_q + 1);
        lr_seq_rev(qs - qs0, qrow + qs0);
    } else { rs1 = rs; qs1 = qs; }
    re1 = rs; qe1 = qs;

    for (i = 1; i < cnt1; ++i) {       // gap filling
        const uint64_t ay = a[as1 + i].y;
        if ((ay & (LY_IGNORE | LY_TANDEM)) && i != cnt1 - 1) continue;
        re = (int32_t)a[as1 + i].x - hk; qe = (int32_t)ay - hk;
        re1 = re; qe1 = qe;
        if (i == cnt1 - 1 || (ay & LY_LONG_JOIN) || (qe - qs >= P.min_ksw_len && re - rs >= P.min_ksw_len)) {
            int32_t j, bw1 = bw_long, zdrop_code;
            if (ay & LY_LONG_JOIN) bw1 = qe - qs > re - rs ? qe - qs : re - rs;
            if ((uint32_t)(re - rs > 0 ? re - rs : 0) + 16 > W.cap_t) { C.err = 13; return false; }
            lr_getseq(C, rid, rs, re, tseq);
            if (!lr_align_pair(C, qe - qs, qrow + qs, re - rs, tseq, bw1, -1, P.zdrop, EZ_APPROX_MAX, ez)) return false;      // first pass: approximate maximum
            zdrop_code = lr_test_zdrop(C, qrow + qs, tseq, ez.n_cigar, ezc);
            lr_tick(C.clk, 9);
            if (C.err) return false;
            if (zdrop_code != 0) {
                if (!lr_align_pair(C, qe - qs, qrow + qs, re - rs, tseq, bw1, -1, zdrop_code == 2 ? P.zdrop_inv : P.zdrop, 0, ez)) return false;
            }
            if (ez.n_cigar > 0) { if (lane == 0) append_cigar0(rc, rn, ez.n_cigar, ezc); r.has_p = 1; }
            if (ez.zdropped) {
                r.has_p = 1;
                for (j = i - 1; j >= 0; --j) if ((int32_t)a[as1 + j].x <= rs + ez.max_t) break;
                dropped = 1;
                if (j < 0) j = 0;
                re1 = rs + (ez.max_t + 1);
                qe1 = qs + (ez.max_q + 1);
                if (cnt1 - (j + 1) >= P.min_cnt) {      // mm_split_reg(r, r2, as1 + j + 1 - r->as)
                    const int32_t n = as1 + j + 1 - r.as;
                    if (n > 0 && n < r.cnt) {
                        r2 = r;
                        r2.id = -1; r2.has_p = 0; r2.split_inv = 0;
                        r2.cnt = r.cnt - n;
                        r2.score = (int32_t)(r.score * ((float)r2.cnt / r.cnt) + .499);
                        r2.as = r.as + n;
                        if (r.parent == r.id) r2.parent = LR_PARENT_TMP_PRI;
                        lr_reg_set_coor(r2, qlen, a);
                        r.cnt -= r2.cnt;
                        r.score -= r2.score;
                        lr_reg_set_coor(r, qlen, a);
                        r.split |= 1; r2.split |= 2;
                        if (zdrop_code == 2) r2.split_inv = 1;
                    }
                }
                break;
            }
            rs = re; qs = qe;
        }
    }

    if (!dropped && qe < qe0 && re < re0) {   // right extension
        lr_getseq(C, rid, re, re0, tseq);
        if (!lr_align_pair(C, qe0 - qe, qrow + qe, re0 - re, tseq, bw, P.end_bonus, P.zdrop, EZ_EXTZ_ONLY, ez)) return false;
        if (ez.n_cigar > 0) { if (lane == 0) append_cigar0(rc, rn, ez.n_cigar, ezc); r.has_p = 1; }
        re1 = re + (ez.reach_end ? ez.mqe_t + 1 : ez.max_t + 1);
        qe1 = qe + (ez.reach_end ? qe0 - qe : ez.max_q + 1);
    }

    r.rs = rs1; r.re = re1;
    if (rev) { r.qs = qlen - qe1; r.qe = qlen - qs1; } else { r.qs = qs1; r.qe = qe1; }
    lr_tick(C.clk, 11);
    if (r.has_p) {
        if ((uint32_t)(re1 - rs1 > 0 ? re1 - rs1 : 0) + 16 > W.cap_t) { C.err = 13; return false; }
        lr_getseq(C, rid, rs1, re1, tseq);
        int32_t v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (lane == 0) {
            int32_t mlen = 0, blen = 0, dpm = 0;
            update_extra0<LReg>(r, rc, rn, W.qseq + (r.rev ? qlen : 0) + qs1, false, tseq, false, *C.AP, mlen, blen, dpm, true);
            v[0] = mlen; v[1] = blen; v[2] = dpm; v[3] = r.qs; v[4] = r.qe; v[5] = r.rs;
        }
        r.mlen = al_b0(v[0]); r.blen = al_b0(v[1]); r.dp_max = al_b0(v[2]); r.qs = al_b0(v[3]); r.qe = al_b0(v[4]); r.rs = al_b0(v[5]);
        lr_tick(C.clk, 10);
    }
    return true;
}

// mm_align1_inv: between the two halves of a region split by the inversion z-drop, align the reverse complement
__device__ __noinline__ int32_t lr_align1_inv(LongCtx &C, const LReg &r1, const LReg &r2, LReg &ri)
{
    const LongParams &P = *C.P;
    LongWs &W = *C.W;
    const uint32_t lane = al_lane();
    const int32_t qlen = C.qlen;
    if (!(r1.split & 1) || !(r2.split & 2)) return 0;
    if (r1.id != r1.parent && r1.parent != LR_PARENT_TMP_PRI) return 0;
    if (r2.id != r2.parent && r2.parent != LR_PARENT_TMP_PRI) return 0;
    if (r1.rid != r2.rid || r1.rev != r2.rev) return 0;
    const int32_t ql = r1.rev ? r1.qs - r2.qe : r2.qs - r1.qe, tl = r2.rs - r1.re;
    if (ql < P.min_sc || ql > P.max_gap) return 0;
    if (tl < P.min_sc || tl > P.max_gap) return 0;
    if ((uint32_t)tl + 16 > W.cap_t || (uint32_t)tl + 16 > W.cap_k || (uint32_t)ql + 16 > W.cap_k) { C.err = 14; return 0; }
    uint8_t *tseq = W.tseq;
    lr_getseq(C, r1.rid, r1.re, r2.rs, tseq);
    uint8_t *qseq = r1.rev ? W.qseq + r2.qe : W.qseq + qlen + (qlen - r2.qs);
    lr_seq_rev(ql, qseq);
    lr_seq_rev(tl, tseq);
    int32_t q_off, t_off;
    const int32_t score = lr_ksw_ll_wave(ql, qseq, tl, tseq, C.sc_mch, C.sc_mis, C.sc_amb, P.q, P.e, q_off, t_off, W.lH, W.lE, W.lHmax);
    lr_seq_rev(ql, qseq);
    lr_seq_rev(tl, tseq);
    if (score < P.min_dp_max) return 0;
    q_off = ql - (q_off + 1); t_off = tl - (t_off + 1);
    Ez ez;
    if (!lr_align_pair(C, ql - q_off, qseq + q_off, tl - t_off, tseq + t_off, (int32_t)(P.bw * 1.5), -1, P.zdrop, EZ_EXTZ_ONLY, ez)) return -1;
    if (ez.n_cigar == 0) return 0;
    int32_t rn = 0;
    if (lane == 0) append_cigar0(W.r_cigar, rn, ez.n_cigar, W.ez_cigar);
    LReg z{};
    ri = z;
    ri.has_p = 1;
    ri.id = -1; ri.parent = LR_PARENT_UNSET; ri.inv = 1; ri.rev = !r1.rev; ri.rid = r1.rid; ri.div = -1.0f;
    if (ri.rev == 0) { ri.qs = r2.qe + q_off; ri.qe = ri.qs + ez.max_q + 1; }
    else { ri.qe = r2.qs - q_off; ri.qs = ri.qe - (ez.max_q + 1); }
    ri.rs = r1.re + t_off;
    ri.re = ri.rs + ez.max_t + 1;
    int32_t v[6] = {0, 0, 0, 0, 0, 0};
    if (lane == 0) {
        int32_t mlen = 0, blen = 0, dpm = 0;
        rn = ez.n_cigar;      // a single append: the region's CIGAR is the extension's
        update_extra0<LReg>(ri, W.r_cigar, rn, qseq + q_off, false, tseq + t_off, false, *C.AP, mlen, blen, dpm, true);
        v[0] = mlen; v[1] = blen; v[2] = dpm; v[3] = ri.qs; v[4] = ri.qe; v[5] = ri.rs;
    }
    ri.mlen = al_b0(v[0]); ri.blen = al_b0(v[1]); ri.dp_max = al_b0(v[2]); ri.qs = al_b0(v[3]); ri.qe = al_b0(v[4]); ri.rs = al_b0(v[5]);
    return 1;
}


// ---- flag-only calls: a region is known to survive mm_filter_regs long before it is completely aligned ---------------------------------
// The boundary returns `mappings.len() > 0`; one surviving region settles a read.  mm_align1 assembles a region's CIGAR as
// left extension + gap fillings + right extension and mm_update_extra reads mlen and dp_max off it; dp_max is the maximum of a running
// score that is clamped at zero.  Take the gap fillings of the first anchors alone, C = seg_1 + ... + seg_i (no z-drop among them), and
// let M_1 .. M_g be the match runs of mm_fix_cigar(C).  Whatever the rest of the region turns out to be, the final CIGAR F = left + C + rest:
//   * mm_fix_cigar(F) equals mm_fix_cigar(C) on every operation from M_1 to M_{g-1}: left-alignment moves an indel into the match run in
//     front of it only, the I/D merging stops at a non-empty match run, so what lies outside C can lengthen M_1 at its front, shorten or
//     swallow M_g and re-pair the gaps next to it, nothing more;
//   * left-alignment re-pairs equal bases only, so the matching columns of C are matching columns of F: mlen(F) >= mlen(C);
//   * a clamped running score started at zero at M_1 never exceeds the true one (which is >= 0 there and sees the same columns): the largest
//     value it takes up to the end of M_{g-1} is a lower bound of dp_max(F);
//   * a z-drop in a later gap filling cuts the region behind C and leaves it as many anchors as lie `k/2 + 1` before C's last one
//     (mm_align1's split rule), which is checked; max_clip_ratio >= 1 disables the clip test (checked on the host).
// So mlen(C) >= min_chain_score, that bound >= min_dp_max and enough anchors prove the region is kept - typically after one gap filling
// of ~200 bases instead of the whole read.  A region the probe cannot vouch for takes the complete procedure.
__device__ inline int32_t lr_probe_eval0(const LongCtx &C, uint32_t *pc, int32_t n_pc, uint32_t *tmp, const uint8_t *qseq, const uint8_t *tseq)
{   // lane 0.  1: proven
    const LongParams &P = *C.P;
    int32_t mlen = 0;
    {
        int32_t toff = 0, qoff = 0;
        for (int32_t k = 0; k < n_pc; ++k) {
            const uint32_t op = pc[k] & 0xf, len = pc[k] >> 4;
            if (op == 0) {
                for (uint32_t l = 0; l < len; ++l) { const int32_t cq = qseq[qoff + (int32_t)l], ct = tseq[toff + (int32_t)l]; mlen += (ct <= 3 && cq <= 3 && ct == cq); }
                toff += (int32_t)len; qoff += (int32_t)len;
            } else if (op == 1) qoff += (int32_t)len;
            else toff += (int32_t)len;
        }
    }
    if (mlen < P.min_sc) return 0;
    for (int32_t k = 0; k < n_pc; ++k) tmp[k] = pc[k];
    int32_t n = n_pc, qshift = 0, tshift = 0;
    LReg dummy{};
    fix_cigar0(dummy, tmp, n, qseq, false, tseq, false, qshift, tshift);
    qseq += qshift; tseq += tshift;
    int32_t first_m = -1, last_m = -1;
    for (int32_t k = 0; k < n; ++k) if ((tmp[k] & 0xf) == 0 && (tmp[k] >> 4) != 0) { if (first_m < 0) first_m = k; last_m = k; }
    if (first_m < 0 || last_m == first_m) return 0;
    int32_t toff = 0, qoff = 0;
    double sc = 0.0, mx = 0.0;
    for (int32_t k = 0; k < last_m; ++k) {
        const uint32_t op = tmp[k] & 0xf, len = tmp[k] >> 4;
        if (op == 0) {
            if (k >= first_m) {
                for (uint32_t l = 0; l < len; ++l) {
                    const int32_t cq = qseq[qoff + (int32_t)l], ct = tseq[toff + (int32_t)l];
                    sc += (ct > 3 || cq > 3) ? C.sc_amb : (ct == cq ? C.sc_mch : C.sc_mis);
                    if (sc < 0) sc = 0; else mx = mx > sc ? mx : sc;
                }
            }
            toff += (int32_t)len; qoff += (int32_t)len;
        } else {
            if (k > first_m) { sc -= P.q + (double)P.e * al_mg_log2((float)(1.0 + len)); if (sc < 0) sc = 0; }
            if (op == 1) qoff += (int32_t)len; else toff += (int32_t)len;
        }
    }
    return (int32_t)(mx + .499) >= P.min_dp_max ? 1 : 0;
}

// 1: the region survives; 0: unknown (take the complete procedure); -1: stop (C.need_big / C.err)
// follow_splits: a z-drop in one of the fillings does not end the probe - mm_align1 cuts the region there (mm_split_reg) and the anchors behind
// the drop become a region of their own, inserted behind this one and aligned in its turn whatever the other regions do; the probe moves on
// to that region (mm_fix_bad_ends and the seed filters on its anchors, then its fillings).  Only where the caller starts the read over
// after an undecided probe (the flags those filters leave on the anchors are upstream's only in upstream's order).
__device__ __noinline__ int32_t lr_probe_region(LongCtx &C, const LReg &r, LAnchor *a, int32_t &as1, int32_t &cnt1, bool follow_splits = false)
{
    const LongParams &P = *C.P;
    LongWs &W = *C.W;
    const uint32_t lane = al_lane();
    const int32_t hk = P.k >> 1;
    as1 = -1; cnt1 = 0;
    if (r.cnt == 0 || r.inv || !(P.max_clip_ratio >= 1.0f)) { C.probe_why = 1; return 0; }
    const int32_t rid = (int32_t)(a[r.as].x << 1 >> 33), rev = (int32_t)(a[r.as].x >> 63);
    int32_t bw_long = (int32_t)(P.bw_long * 1.5 + 1.);
    { const int32_t bw = (int32_t)(P.bw * 1.5 + 1.); if (bw_long < bw) bw_long = bw; }
    uint8_t *qrow = W.qseq + (rev ? C.qlen : 0), *tseq = W.tseq;
    uint32_t *pc = W.r_cigar, *ezc = W.ez_cigar, *tmp = (uint32_t *)W.K;
    LReg rr = r;      // the region under the probe: r, then what a z-drop splits off it
    for (int depth = 0; depth < 4; ++depth) {
        int32_t as_c, cnt_c;
        lr_fix_bad_ends0(rr, a, P.bw, P.min_sc * 2, as_c, cnt_c);
        if (depth == 0) { as1 = as_c; cnt1 = cnt_c; }
        {
            const int32_t n1 = lr_collect_long_gaps_wave(as_c, cnt_c, a, 10, W.K);
            if (lane == 0) lr_filter_bad_seeds0(as_c, cnt_c, a, 10, 40, P.max_gap >> 1, 10, W.K, n1);
            lr_sync();
            const int32_t n2 = lr_collect_long_gaps_wave(as_c, cnt_c, a, 30, W.K);
            if (lane == 0) lr_filter_bad_seeds_alt0(as_c, cnt_c, a, 30, P.max_gap >> 1, W.K, n2);
        }
        lr_sync();
        const int32_t rs_first = (int32_t)a[as_c].x - hk, qs_first = (int32_t)a[as_c].y - hk;
        int32_t rs = rs_first, qs = qs_first, n_pc = 0, n_seg = 0;
        Ez ez;
        bool split = false;
        for (int32_t i = 1; i < cnt_c && n_seg < 6; ++i) {
            const uint64_t ay = a[as_c + i].y;
            if ((ay & (LY_IGNORE | LY_TANDEM)) && i != cnt_c - 1) continue;
            const int32_t re = (int32_t)a[as_c + i].x - hk, qe = (int32_t)ay - hk;
            if (!(i == cnt_c - 1 || (ay & LY_LONG_JOIN) || (qe - qs >= P.min_ksw_len && re - rs >= P.min_ksw_len))) continue;
            int32_t bw1 = bw_long;
            if (ay & LY_LONG_JOIN) bw1 = qe - qs > re - rs ? qe - qs : re - rs;
            if ((uint32_t)(re - rs_first > 0 ? re - rs_first : 0) + 16 > W.cap_t) { C.probe_why = 2; return 0; }
            // an alignment the first pass's buffers do not hold is not worth probing: the complete procedure deals with it
            {
                const int32_t ql = qe - qs, tl = re - rs;
                if (ql <= 0 || tl <= 0 || (long long)ql * tl > 4000000ll) { C.probe_why = 3; return 0; }
            }
            lr_getseq(C, rid, rs, re, tseq);
            if (!lr_align_pair(C, qe - qs, qrow + qs, re - rs, tseq, bw1, -1, P.zdrop, EZ_APPROX_MAX, ez)) { if (C.need_big) { C.need_big = false; C.probe_why = 4; return 0; } return -1; }
            const int32_t zdrop_code = lr_test_zdrop(C, qrow + qs, tseq, ez.n_cigar, ezc);
            if (C.err) return -1;
            if (zdrop_code != 0) {
                if (!lr_align_pair(C, qe - qs, qrow + qs, re - rs, tseq, bw1, -1, zdrop_code == 2 ? P.zdrop_inv : P.zdrop, 0, ez)) { if (C.need_big) { C.need_big = false; C.probe_why = 4; return 0; } return -1; }
            }
            if (ez.zdropped) {
                if (!follow_splits) { C.probe_why = 5; return 0; }
                // mm_align1: the anchors behind the drop (beyond rs + max_t) leave for a region of their own if enough of them are left
                int32_t j;
                for (j = i - 1; j >= 0; --j) if ((int32_t)a[as_c + j].x <= rs + ez.max_t) break;
                if (j < 0) j = 0;
                const int32_t n = as_c + j + 1 - rr.as;
                if (cnt_c - (j + 1) < P.min_cnt || !(n > 0 && n < rr.cnt)) { C.probe_why = 5; return 0; }
                LReg r2 = rr;
                r2.cnt = rr.cnt - n; r2.as = rr.as + n;
                lr_reg_set_coor_wave(r2, C.qlen, a);
                rr = r2;
                split = true;
                break;
            }
            if (ez.n_cigar == 0) { C.probe_why = 5; return 0; }
            if ((uint32_t)(n_pc + ez.n_cigar + 8) > W.cap_a || (uint32_t)(n_pc + ez.n_cigar + 8) > W.cap_c) { C.probe_why = 6; return 0; }
            if (lane == 0) append_cigar0(pc, n_pc, ez.n_cigar, ezc);
            n_pc = al_b0(n_pc);
            ++n_seg;
            rs = re; qs = qe;
            // anchors the region keeps if a later gap filling drops right behind this one
            int32_t jstar = i;
            while (jstar >= 0 && (int32_t)a[as_c + jstar].x > (int32_t)a[as_c + i].x - hk - 1) --jstar;
            if ((as_c - rr.as) + jstar + 1 < P.min_cnt) continue;
            lr_getseq(C, rid, rs_first, re, tseq);
            int32_t ok = 0;
            if (lane == 0) ok = lr_probe_eval0(C, pc, n_pc, tmp, qrow + qs_first, tseq);
            ok = al_b0(ok);
            lr_tick(C.clk, 10);
            if (ok) return 1;
        }
        if (!split) { C.probe_why = 7; return 0; }
    }
    C.probe_why = 7;
    return 0;
}

// the chain that holds flat anchor index i: largest c with off[c] <= i (off ascending, off[n] = total)
template <class OFF>
__device__ inline int32_t lr_chain_of(OFF off, int32_t n, int32_t i)
{
    int32_t lo = 0, hi = n - 1;
    while (lo < hi) { const int32_t mid = (lo + hi + 1) >> 1; if ((int32_t)off[mid] <= i) lo = mid; else hi = mid - 1; }
    return lo;
}

// ---- the whole stage for one read ---------------------------------------------------------------------------------------------------
struct LongOut { int32_t n_chain, best, rechained, n_aligned, n_regs, dp_max; uint32_t sig; int32_t rmq_tie, probed; int32_t n_join, rmq_asked; };      // n_join: anchors that entered the long join; rmq_asked (EXACT): steps of the join that asked the tree
// The stage runs as two kernels, so that neither carries the other's registers and LDS: the first leaves a read's final chains (after the
// long join, in compact_a's order, MM_SEED_TANDEM set) in an arena; the second turns them into regions and aligns.
struct LongHdr { unsigned long long off; int32_t n_u, n_a, best, rechained; unsigned long long alt; };      // per read: u[n_u] (8 B), uoff[n_u + 1] (4 B), a[n_a] (16 B) at arena + off; alt - 1: a second outcome to prove (lr_chains_wave), {n_a, score, 0, 0} + a[n_a]
struct LongArena { uint8_t *base; unsigned long long cap; unsigned long long *cursor; LongHdr *hdr; };
__host__ __device__ inline unsigned long long long_arena_bytes(int32_t n_u, int32_t n_a)
{
    return (((unsigned long long)n_u * 8 + ((unsigned long long)n_u + 1) * 4 + 15) & ~15ull) + (unsigned long long)n_a * 16;
}


__device__ inline bool lr_region_kept(const LongParams &P, int32_t qlen, const LReg &r)
{   // mm_filter_regs, one region
    int32_t flt = 0;
    if (!r.inv && r.cnt < P.min_cnt) flt = 1;
    if (r.has_p) {
        if (r.mlen < P.min_sc) flt = 1;
        else if (r.dp_max < P.min_dp_max) flt = 1;
        else if (r.qs > qlen * P.max_clip_ratio && qlen - r.qe > qlen * P.max_clip_ratio) flt = 1;
    }
    return !flt;
}

// 0: done; 1: an alignment needs a larger direction-byte buffer; 3: another capacity of the working memory was exceeded (C.err).
// Either way the caller hands the read to the pass with the large working memory.
// ---- first kernel: the read's final chains ----
// 0: done (header written; n_u may be 0); 3: a capacity of the working memory was exceeded (C.err); 4: the arena is full
// drop: the largest cluster bound k_lr_locus left out of this read's anchors (0: the read has all of them).  With anchors left out the read's
// chains are those of the clusters kept - exactly, clusters are independent in both chaining passes - and a chain of a cluster left out would
// score at most k * drop (none at all when drop < min_cnt: such a read is complete for every purpose).  What the answer needs beyond that, and what is checked here (5 = the read must be redone with every anchor, C.err
// says why): whether mm_map_frag re-chains at all (more than one chain in the first pass; for a short read also which chain comes first),
// and that regs[0] - the only region a flag-only call asks about - is among the chains kept: top score > k * drop.
// EXACT: the long join on the literal trees (lr_rmq_fill_tree).  Without it a join that meets two candidates of equal priority, or that the
// LDS ring cannot hold, returns 6: the read is redone by the EXACT instance of the kernel.
template <int NR, bool EXACT, bool FAT = false>
__device__ inline int32_t lr_chains_wave(LongCtx &C, RmqLdsT<NR, FAT> &RL, const LongArena &AR, LongOut &out, uint32_t drop = 0, RqLds TL = RqLds{})
{
    const LongParams &P = *C.P;
    LongWs &W = *C.W;
    const AlignIn &in = C.I->in;
    const int32_t lane = (int32_t)al_lane();
    const uint32_t read = C.read;
    const int32_t qlen = C.qlen;
    out.n_chain = out.best = out.rechained = out.n_aligned = out.n_regs = out.dp_max = 0; out.sig = 0; out.rmq_tie = 0; out.probed = 0; out.n_join = 0; out.rmq_asked = 0;
    if ((uint32_t)qlen > W.cap_q) { C.err = 4; return 3; }

    // ---- the read's chains, in compact_a's order: by the first anchor's x, ties in discovery order (larger (f, index) first)
    int32_t n_u = 0;
    for (uint32_t h = in.head[read]; h != ~0u; h = in.recs[h].next) {
        if ((uint32_t)n_u < W.cap_u && (n_u & 63) == lane) W.K[n_u] = (int32_t)h;
        ++n_u;
    }
    if (n_u == 0) { if (lane == 0) { LongHdr h{0ull, 0, 0, 0, 0}; AR.hdr[read] = h; } return 0; }
    if ((uint32_t)n_u > W.cap_u) { C.err = 2; return 3; }
    lr_sync();
    for (int32_t i = lane; i < n_u; i += 64) {
        const ChainRec rc = in.recs[W.K[i]];
        W.sk[i].k = in.cx[rc.off]; W.sk[i].v = ~((uint64_t)rc.key_f << 32 | rc.key_i); W.sk[i].i = (uint32_t)W.K[i];      // of equal x the earlier discovery first
    }
    lr_sync();
    lr_sort(W.sk, W.sk2, n_u);
    for (int32_t i = lane; i < n_u; i += 64) W.v[i] = (int32_t)W.sk[i].i;
    lr_sync();
    int32_t n_a = 0;
    {
        int32_t tot = 0;
        if (lane == 0) {
            for (int32_t i = 0; i < n_u; ++i) {
                const ChainRec rc = in.recs[W.v[i]];
                W.u[i] = (uint64_t)(uint32_t)rc.score << 32 | rc.cnt;
                W.uoff[i] = (uint32_t)tot;
                tot += (int32_t)rc.cnt;
                if (tot < 0 || (uint32_t)tot > W.cap_a) { tot = -1; break; }
            }
            if (tot >= 0) W.uoff[n_u] = (uint32_t)tot;
        }
        n_a = al_b0(tot);
        if (n_a < 0) { C.err = 5; return 3; }
    }
    lr_sync();
    // ---- which query positions carry a tandem seed (MM_SEED_TANDEM on their anchors)
    {
        const uint32_t info = C.I->k1info[read];
        const uint32_t n_seed = C.I->seed_off ? info >> 16 : (info >> 16 & 0x7fffu);
        const uint4 *rec = C.I->rec + (C.I->seed_off ? (size_t)C.I->seed_off[read] : (size_t)read * C.I->seed_cap);
        const uint32_t n_rec = C.I->seed_off ? n_seed : (n_seed < C.I->seed_cap ? n_seed : C.I->seed_cap);
        for (int32_t i = lane; i < qlen / 32 + 1; i += 64) W.tbits[i] = 0;
        lr_sync();
        for (uint32_t j = (uint32_t)lane; j < n_rec; j += 64) {
            bool td = (rec[j].z & SH_REC_PREV_SAME) != 0;
            if (!td && j + 1 < n_rec) td = (rec[j + 1].z & SH_REC_PREV_SAME) != 0;
            if (td) { const uint32_t qp = rec[j].w >> 1; if (qp < (uint32_t)qlen) atomicOr(&W.tbits[qp >> 5], 1u << (qp & 31)); }
        }
        lr_sync();
    }
    LAnchor *A0 = W.a, *B0 = W.b;
    if (C.clk) C.clk->last = wall_clock64();
    // every lane takes anchors of the flat index space and finds their chain itself: a read with thousands of small chains must not pay a
    // memory round trip per chain
    for (int32_t i = lane; i < n_a; i += 64) {
        const int32_t c = lr_chain_of(W.uoff, n_u, i);
        const ChainRec rc = in.recs[W.v[c]];
        const uint32_t j = (uint32_t)i - W.uoff[c];
        const uint64_t x = in.cx[rc.off + j]; const uint32_t q = in.cq[rc.off + j];
        const uint32_t qp = (x >> 63) ? (uint32_t)(qlen + P.k - 2) - q : q;
        const bool td = qp < (uint32_t)qlen && (W.tbits[qp >> 5] >> (qp & 31) & 1u);
        A0[i].x = x; A0[i].y = (td ? LY_TANDEM : 0ull) | (uint64_t)(uint32_t)P.k << 32 | q;
    }
    lr_sync();
    out.n_chain = n_u;
    {
        int32_t best = 0;
        for (int32_t i = lane; i < n_u; i += 64) { const int32_t s = (int32_t)(W.u[i] >> 32); best = s > best ? s : best; }
        best = wave_all_max(best);
        out.best = best;
    }

    lr_tick(C.clk, 0);
    const bool chainable_out = drop >= (uint32_t)(P.min_cnt > 1 ? P.min_cnt : 1);      // a cluster left out may hold a chain of its own
    // `both`: one chain among the anchors kept, more possible among those left out - whether mm_map_frag runs the long join (n_regs0 > 1) is
    // not known.  Either way regs[0] comes from this cluster (the check on the top score below), so both outcomes are prepared: the chains
    // the join leaves (the read's arena block) and the chain as it stands (`alt`); the regions kernel wants a proof from each.  Most of the
    // time the join returns the one chain with all its anchors and the two coincide.
    bool both = false;
    int32_t s1 = 0;                            // the first pass's top score (`both`)
    unsigned long long alt_off = 0;            // arena offset + 1 of the alternative
    if (chainable_out && P.bw_long > P.bw) {
        // the rescue test on the first chain in x order must not depend on a chain left out: it holds for every span when the read is long
        // enough (qlen - span > rescue_size or span > qlen * rescue_ratio)
        const bool always = (float)(qlen - P.rmq_rescue_size) > (float)qlen * P.rmq_rescue_ratio;
        if (!always) { C.err = 41; return 5; }
        if (n_u < 2) { both = true; s1 = out.best; }
    }
    // ---- mm_map_frag: re-chain / long join
    if (P.bw_long > P.bw && (n_u > 1 || both)) {
        const int32_t st = (int32_t)A0[0].y, en = (int32_t)A0[(int32_t)(uint32_t)W.u[0] - 1].y;
        if (both || qlen - (en - st) > P.rmq_rescue_size || en - st > qlen * P.rmq_rescue_ratio) {
            if (!both) out.rechained |= 2;
            const int32_t n_a_first = n_a;
            out.n_join = n_a;
            for (int32_t i = lane; i < n_a; i += 64) { W.sk[i].k = A0[i].x; W.sk[i].v = (uint64_t)i; }
            lr_sync();
            lr_sort(W.sk, W.sk2, n_a);
            for (int32_t i = lane; i < n_a; i += 64) B0[i] = A0[W.sk[i].v];
            lr_sync();
            int32_t tie = 0;
            lr_tick(C.clk, 1);
            if constexpr (EXACT) {
                // the scan with upstream's main tree beside it (in LDS, asked at the ties).  What this instance's ring or tree cannot hold goes to
                // the instance with the larger ones; beyond those, both trees on one lane over node pools in the wave's scratch (the launch that
                // brings them: W.rq0)
                bool filled;
                if (C.coop && n_a >= P.coop_min) filled = lr_coop_fill<NR, FAT>(P, *C.coop, C.coop_me, n_a, B0, W.f, W.p, W.pri, (double *)W.K, W.t, RL, TL, tie, W.v, (int32_t *)W.sk, (double *)W.sk2, (double *)W.sk + W.cap_a);
                else filled = lr_rmq_fill<NR, true, FAT>(P, P.max_gap, P.bw_long, n_a, B0, W.f, W.p, W.pri, (double *)W.K, RL, tie, C.clk, TL);
                if (!filled) {
                    if (NR < 4096) { C.err = 6; return 3; }
                    if (!P.rmq_one_lane) { C.err = 6; return 7; }      // beyond the large ring / the LDS tree: counted (sh_stats.n_ext_unresolved) unless the one-lane trees are asked for
                    if (!W.rq0) { C.err = 7; return 3; }
                    lr_sync();
                    for (int32_t i = lane; i < n_a; i += 64) W.t[i] = 0;
                    lr_sync();
                    uint32_t code = 0;
                    if (!lr_rmq_fill_tree(P, P.max_gap, P.bw_long, n_a, B0, W.f, W.p, W.t, W.rq0, W.rq1, (int32_t)W.cap_a + 2, &code)) { C.err = 100u + (uint32_t)al_b0((int32_t)code); return 3; }
                }
                out.rmq_asked = tie;
            } else {
                if (!lr_rmq_fill<NR, false, FAT>(P, P.max_gap, P.bw_long, n_a, B0, W.f, W.p, W.pri, (double *)W.K, RL, tie, C.clk)) {
                    if (NR < 4096) { C.err = 6; return 3; }      // beyond this ring: the pass with the large one
                    if (P.rmq_one_lane) { C.err = 51; return 6; }      // beyond that too: the one-lane trees, by way of the exact passes
                    C.err = 6; return 7;      // ... which one lane would walk for seconds on a read this size: given up, counted (sh_stats.n_ext_unresolved)
                }
                if (tie) {      // the scan's choice among equal priorities need not be the tree's
                    if (P.rmq_exact_max < 0 || n_a <= P.rmq_exact_max) { C.err = 50; return 6; }
                    out.rmq_tie = 1;      // left open (the smallest index stands), counted: sh_stats.n_rmq_open
                }
            }
            lr_sync();
            lr_tick(C.clk, 2);
            // mg_chain_backtrack
            int32_t n_z = 0;
            for (int32_t i0 = 0; i0 < n_a; i0 += 64) {
                const int32_t i = i0 + lane;
                const bool c = i < n_a && W.f[i] >= P.min_sc;
                const uint64_t m = __ballot(c);
                if (c) { const int32_t d = n_z + (int32_t)prefix_popc64(m); W.sk[d].k = (uint64_t)(uint32_t)W.f[i]; W.sk[d].v = (uint64_t)i; }
                n_z += (int32_t)__popcll(m);
            }
            lr_sync();
            lr_sort(W.sk, W.sk2, n_z);
            for (int32_t i = lane; i < n_a; i += 64) W.t[i] = 0;
            lr_sync();
            int32_t res[3] = {0, 0, 0};
            {
                int32_t n_v = 0, best = 0; bool ovf = false;
                res[0] = lr_backtrack_wave(W.sk, n_z, n_a, W.f, W.p, W.t, W.v, W.u, W.cap_u, P.min_cnt, P.min_sc, P.bw_long, n_v, best, ovf);
                res[1] = ovf ? -1 : n_v; res[2] = best;
                lr_sync();
                if (!ovf && lane == 0) {      // chain starts in v[] (discovery order)
                    int32_t k0 = 0;
                    for (int32_t i = 0; i < res[0]; ++i) { W.uoff[i] = (uint32_t)k0; k0 += (int32_t)(uint32_t)W.u[i]; }
                    W.uoff[res[0]] = (uint32_t)k0;
                }
            }
            n_u = al_b0(res[0]);
            const int32_t n_v = al_b0(res[1]);
            out.best = al_b0(res[2]); out.n_chain = n_u;
            if (n_v < 0) { C.err = 2; return 3; }
            lr_sync();
            if (both && !(n_u == 1 && n_v == n_a_first)) {
                // the join changes the chain: keep it as it stood (B0 holds its anchors, sorted by x - the order of a chain) for the second proof
                const unsigned long long bytes = 16ull + (unsigned long long)n_a_first * 16ull;
                unsigned long long off = 0;
                if (lane == 0) off = atomicAdd(AR.cursor, bytes);
                off = lr_b0_64(off);
                if (off + bytes > AR.cap) return 4;
                if (lane == 0) { int32_t *hh = (int32_t *)(AR.base + off); hh[0] = n_a_first; hh[1] = s1; hh[2] = hh[3] = 0; }
                LAnchor *AO = (LAnchor *)(AR.base + off + 16);
                for (int32_t i = lane; i < n_a_first; i += 64) AO[i] = B0[i];
                alt_off = off + 1;
            }
            if (n_u == 0) {
                if (chainable_out) { C.err = 43; return 5; }      // the clusters kept hold no chain after the join: the answer lies with the rest
                if (lane == 0) { LongHdr h{0ull, 0, 0, 0, out.rechained | (out.rmq_tie ? 4 : 0)}; AR.hdr[read] = h; }
                return 0;
            }
            // compact_a: every chain ascending, then the chains by the x of their first anchor (ties: discovery order)
            for (int32_t i = lane; i < n_v; i += 64) {
                const int32_t c = lr_chain_of(W.uoff, n_u, i);
                const int32_t k0 = (int32_t)W.uoff[c], ni = (int32_t)(uint32_t)W.u[c], j = i - k0;
                A0[i] = B0[W.v[k0 + (ni - j - 1)]];
            }
            lr_sync();
            for (int32_t c = lane; c < n_u; c += 64) { W.sk[c].k = A0[W.uoff[c]].x; W.sk[c].v = (uint64_t)c; }
            lr_sync();
            lr_sort(W.sk, W.sk2, n_u);
            // chains in their new order: anchors to B0, u to pri (as raw bits), offsets recomputed
            uint64_t *u2 = (uint64_t *)W.pri;
            if (lane == 0) {
                int32_t k0 = 0;
                for (int32_t c = 0; c < n_u; ++c) { const int32_t s = (int32_t)W.sk[c].v; u2[c] = W.u[s]; W.K[c] = k0; k0 += (int32_t)(uint32_t)W.u[s]; }
                W.K[n_u] = k0;
            }
            lr_sync();
            for (int32_t i = lane; i < n_v; i += 64) {      // flat over the new order
                const int32_t c = lr_chain_of(W.K, n_u, i);
                const int32_t s = (int32_t)W.sk[c].v, so = (int32_t)W.uoff[s], d = W.K[c];
                B0[i] = A0[so + (i - d)];
            }
            lr_sync();
            for (int32_t c = lane; c <= n_u; c += 64) { if (c < n_u) W.u[c] = u2[c]; W.uoff[c] = (uint32_t)W.K[c]; }
            n_a = n_v;
            { LAnchor *x = A0; A0 = B0; B0 = x; }
            lr_sync();
        }
    }

    lr_tick(C.clk, 3);
    if (chainable_out && ((long long)out.best <= (long long)P.k * (long long)drop || (both && (long long)s1 <= (long long)P.k * (long long)drop))) { C.err = 42; return 5; }      // a chain left out could be regs[0]
    // ---- hand the chains to the second kernel
    {
        const unsigned long long bytes = long_arena_bytes(n_u, n_a);
        unsigned long long off = 0;
        if (lane == 0) off = atomicAdd(AR.cursor, bytes);
        off = lr_b0_64(off);
        if (off + bytes > AR.cap) return 4;
        uint64_t *U = (uint64_t *)(AR.base + off);
        uint32_t *UO = (uint32_t *)(U + n_u);
        LAnchor *AO = (LAnchor *)(AR.base + off + (((unsigned long long)n_u * 8 + ((unsigned long long)n_u + 1) * 4 + 15) & ~15ull));
        for (int32_t i = lane; i <= n_u; i += 64) { if (i < n_u) U[i] = W.u[i]; UO[i] = W.uoff[i]; }
        for (int32_t i = lane; i < n_a; i += 64) AO[i] = A0[i];
        if (lane == 0) { LongHdr h; h.off = off; h.n_u = n_u; h.n_a = n_a; h.best = out.best; h.rechained = out.rechained | (out.rmq_tie ? 4 : 0) | (chainable_out ? 8 : 0); h.alt = alt_off; AR.hdr[read] = h; }
    }
    return 0;
}

// ---- second kernel: regions, alignment, mm_filter_regs ----
// 0: done; 1: an alignment needs a larger direction-byte buffer; 3: another capacity of the working memory was exceeded (C.err);
// 5: anchors were left out of this read (k_lr_locus) and regs[0] alone does not prove it mapped: it must be redone with every anchor.
__device__ inline int32_t lr_regs_wave(LongCtx &C, const ChainParams &CP, const LongArena &AR, bool flag_only, bool probe, LongOut &out)
{
    const LongParams &P = *C.P;
    LongWs &W = *C.W;
    const AlignIn &in = C.I->in;
    const int32_t lane = (int32_t)al_lane();
    const uint32_t read = C.read;
    const int32_t qlen = C.qlen;
    const LongHdr hd = AR.hdr[read];
    out.n_chain = hd.n_u; out.best = hd.best; out.rechained = hd.rechained & 3; out.rmq_tie = (hd.rechained & 4) != 0;
    out.n_aligned = out.n_regs = out.dp_max = 0; out.sig = 0; out.probed = 0;
    if ((uint32_t)qlen > W.cap_q) { C.err = 4; return 3; }
    const int32_t n_u = hd.n_u;
    if (n_u == 0) return 0;
    if ((uint32_t)n_u > W.cap_r || (uint32_t)n_u > W.cap_a) { C.err = 2; return 3; }
    const uint64_t *Uh = (const uint64_t *)(AR.base + hd.off);
    const uint32_t *UOh = (const uint32_t *)(Uh + n_u);
    // the anchors are worked on in place (mm_squeeze_a moves them, the seed filters flag them): on a copy, so that a read this pass has to
    // give up half-way starts from the same chains in the next one
    if ((uint32_t)hd.n_a > W.cap_a) { C.err = 5; return 3; }
    LAnchor *A0 = W.a;
    {
        const LAnchor *src = (const LAnchor *)(AR.base + hd.off + (((unsigned long long)n_u * 8 + ((unsigned long long)n_u + 1) * 4 + 15) & ~15ull));
        for (int32_t i = lane; i < hd.n_a; i += 64) A0[i] = src[i];
        lr_sync();
    }
    if (C.clk) C.clk->last = wall_clock64();
    const bool partial = (hd.rechained & 8) != 0;      // k_lr_locus left clusters out that may hold chains: only a proof from regs[0] alone counts
    // ---- mm_align_skeleton's query on both strands (staged early: the probe below reads it)
    {
        const uint8_t *seq = in.bases + in.offsets[read];
        for (int32_t i = lane; i < qlen; i += 64) {
            const uint8_t c = (uint8_t)sh_nt4(seq[i]);
            W.qseq[i] = c;
            W.qseq[qlen + (qlen - 1 - i)] = c < 4 ? 3 - c : 4;
        }
    }
    // ---- mm_gen_regs: regions in descending z = (score << 32 | cnt) ^ h; of equal z the later chain first
    uint32_t hash = 0;
    hash ^= al_wang((uint32_t)qlen) + al_wang(11u);
    hash = al_wang(hash);
    bool skip_probe0 = false;
    if (flag_only && probe) {
        // Probe before bookkeeping.  regs[0] is the chain of largest (z, index); it is its own parent whatever mm_set_parent makes of the
        // others, mm_select_sub keeps every primary, mm_filter_strand_retained only drops strand_retained regions, and its gap fillings
        // read its own anchors only - so the proof lr_probe_region gives does not need the ranking of the other ~400 chains, their
        // parents, mm_est_err or mm_squeeze_a.  A probe that cannot conclude starts the read over with the complete procedure.
        unsigned long long bz = 0; int32_t bi = -1;
        for (int32_t i = lane; i < n_u; i += 64) {
            const LAnchor f0 = A0[UOh[i]];
            const uint32_t h = (uint32_t)al_hash64((al_hash64(f0.x) + al_hash64(f0.y)) ^ hash);
            const unsigned long long z = Uh[i] ^ h;
            if (bi < 0 || z > bz || (z == bz && i > bi)) { bz = z; bi = i; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long oz = (unsigned long long)__shfl_xor((long long)bz, o); const int32_t oi = __shfl_xor(bi, o);
            if (oi >= 0 && (bi < 0 || oz > bz || (oz == bz && oi > bi))) { bz = oz; bi = oi; }
        }
        LReg r{};
        r.id = 0; r.parent = 0;
        r.score = (int32_t)(bz >> 32); r.hash = (uint32_t)bz;
        r.cnt = (int32_t)(uint32_t)Uh[bi]; r.as = (int32_t)UOh[bi];
        r.div = -1.0f;
        lr_sync();
        lr_reg_set_coor_wave(r, qlen, A0);
        int32_t as1, cnt1;
        const int32_t pr = lr_probe_region(C, r, A0, as1, cnt1, true);
        if (pr < 0) return C.need_big ? 1 : 3;
        if (pr > 0 && hd.alt != 0) {
            // mm_map_frag may not have run the long join at all (lr_chains_wave, `both`): then regs[0] is the first pass's one chain
            const int32_t *hh = (const int32_t *)(AR.base + (hd.alt - 1));
            const int32_t n_alt = hh[0];
            if ((uint32_t)n_alt > W.cap_a) { C.err = 5; return 3; }
            const LAnchor *src = (const LAnchor *)(AR.base + (hd.alt - 1) + 16);
            for (int32_t i = lane; i < n_alt; i += 64) A0[i] = src[i];
            lr_sync();
            LReg r1{};
            r1.id = 0; r1.parent = 0; r1.score = hh[1]; r1.cnt = n_alt; r1.as = 0; r1.div = -1.0f;
            lr_reg_set_coor_wave(r1, qlen, A0);
            const int32_t pr2 = lr_probe_region(C, r1, A0, as1, cnt1, true);
            if (pr2 < 0) return C.need_big ? 1 : 3;
            if (pr2 == 0) { C.err = 46; return 5; }
        }
        if (pr > 0) { out.n_regs = 1; out.probed = 1; return 0; }
        if (partial) { C.err = 44; return 5; }
        // the seed filters flagged anchors of the region: start from the arena's copy again
        const LAnchor *src = (const LAnchor *)(AR.base + hd.off + (((unsigned long long)n_u * 8 + ((unsigned long long)n_u + 1) * 4 + 15) & ~15ull));
        for (int32_t i = lane; i < hd.n_a; i += 64) A0[i] = src[i];
        lr_sync();
        skip_probe0 = true;
    } else if (partial) { C.err = 45; return 5; }
    for (int32_t i = lane; i < n_u; i += 64) {
        const LAnchor f0 = A0[UOh[i]];
        const uint32_t h = (uint32_t)al_hash64((al_hash64(f0.x) + al_hash64(f0.y)) ^ hash);
        W.sk[i].k = Uh[i] ^ h; W.sk[i].v = (uint64_t)i;
    }
    lr_sync();
    lr_sort(W.sk, W.sk2, n_u);
    for (int32_t j = lane; j < n_u; j += 64) {
        const SKey z = W.sk[n_u - 1 - j];
        LReg r{};
        r.id = j; r.parent = LR_PARENT_UNSET;
        r.score = (int32_t)(z.k >> 32); r.hash = (uint32_t)z.k;
        r.cnt = (int32_t)(uint32_t)Uh[z.v]; r.as = (int32_t)UOh[z.v];
        r.div = -1.0f;
        lr_reg_set_coor(r, qlen, A0);
        W.regs[j] = r;
    }
    lr_sync();
    lr_tick(C.clk, 4);
    // ---- chain_post, mm_est_err, mm_filter_strand_retained (lane 0)
    int32_t n_regs = n_u;
    {
        int32_t res[2] = {0, 0};
        if (lane == 0) {
            lr_set_parent0(P.mask_level, n_regs, W.regs, W.cov, W.wpri);
            n_regs = lr_select_sub0(P.pri_ratio, P.k * 2, P.best_n, (int32_t)(P.max_gap * 0.8), n_regs, W.regs, W.K);
            bool any_sr = false;
            for (int32_t i = 0; i < n_regs; ++i) any_sr |= W.regs[i].strand_retained != 0;
            if (any_sr) {
                // mm_collect_matches' mini_pos: the seeds mm_seed_select lets through, in minimizer order
                const uint32_t info = C.I->k1info[read];
                const uint32_t n_seed = C.I->seed_off ? info >> 16 : (info >> 16 & 0x7fffu);
                uint4 *rec = (uint4 *)C.I->rec + (C.I->seed_off ? (size_t)C.I->seed_off[read] : (size_t)read * C.I->seed_cap);
                SeedView sv; sv.base = rec; sv.stride = 1; sv.n = n_seed;
                int64_t na2 = 0; int32_t rl2 = 0;
                seed_filter(sv, qlen, CP.mid_occ, CP, na2, rl2);
                int32_t m = 0;
                for (uint32_t j = 0; j < n_seed; ++j) {
                    const uint4 s = rec[j];
                    if (s.z >> 31) continue;
                    if ((uint32_t)m >= W.cap_m) { m = -1; break; }
                    W.mini_pos[m++] = (uint64_t)(uint32_t)P.k << 32 | s.w >> 1;
                }
                if (m < 0) res[1] = 7;
                else {
                    lr_est_err0(qlen, n_regs, W.regs, A0, m, W.mini_pos, in.cstart);
                    n_regs = lr_filter_strand0(n_regs, W.regs);
                }
            }
            res[0] = n_regs;
        }
        n_regs = al_b0(res[0]);
        if (al_b0(res[1])) { C.err = 7; return 3; }
    }
    lr_sync();
    lr_tick(C.clk, 5);
    out.n_aligned = n_regs;

    // ---- mm_align_skeleton: mm_squeeze_a, the regions one after the other (the query was staged above)
    for (int32_t i = lane; i < n_regs; i += 64) { W.sk[i].k = (uint64_t)(uint32_t)W.regs[i].as; W.sk[i].v = (uint64_t)i; }
    lr_sync();
    lr_sort(W.sk, W.sk2, n_regs);
    int32_t n_sq = 0;
    for (int32_t c = 0; c < n_regs; ++c) {
        const int32_t ri = (int32_t)W.sk[c].v;
        const int32_t src = W.regs[ri].as, cnt = W.regs[ri].cnt;
        if (src != n_sq) {
            for (int32_t j0 = 0; j0 < cnt; j0 += 64) {
                const int32_t j = j0 + lane;
                LAnchor e{0, 0};
                if (j < cnt) e = A0[src + j];
                lr_sync();
                if (j < cnt) A0[n_sq + j] = e;
                lr_sync();
            }
            if (lane == 0) W.regs[ri].as = n_sq;
        }
        n_sq += cnt;
    }
    lr_sync();

    lr_tick(C.clk, 6);
    uint32_t sig = 2166136261u;
    int32_t n_keep = 0, dp_best = 0;
    auto account = [&](const LReg &r) {      // a region that mm_filter_regs keeps (uniform)
        const int32_t v[8] = { r.rs, r.re, r.qs, r.qe, r.mlen, r.blen, r.has_p ? r.dp_max : -1, r.cnt };
        for (int32_t j = 0; j < 8; ++j) { sig ^= (uint32_t)v[j]; sig *= 16777619u; }
        if (r.has_p && r.dp_max > dp_best) dp_best = r.dp_max;
        ++n_keep;
    };
    auto insert_after = [&](int32_t i, const LReg &x) -> bool {      // mm_insert_reg
        if ((uint32_t)(n_regs + 1) > W.cap_r) { C.err = 8; return false; }
        if (lane == 0) {
            for (int32_t m = n_regs - 1; m > i; --m) W.regs[m + 1] = W.regs[m];
            W.regs[i + 1] = x;
        }
        ++n_regs;
        lr_sync();
        return true;
    };
    for (int32_t i = 0; i < n_regs; ++i) {
        LReg r = W.regs[i], r2;
        r2.cnt = 0;
        int32_t pre_as1 = -1, pre_cnt1 = 0;
        if (flag_only && probe && !(i == 0 && skip_probe0)) {
            const int32_t pr = lr_probe_region(C, r, A0, pre_as1, pre_cnt1);
            if (pr < 0) return C.need_big ? 1 : 3;
            if (pr > 0) { out.n_regs = 1; out.probed = 1; return 0; }
        }
        if (!lr_align1(C, r, r2, A0, n_sq, pre_as1, pre_cnt1)) return C.need_big ? 1 : 3;
        if (lane == 0) W.regs[i] = r;
        lr_sync();
        if (r2.cnt > 0 && !insert_after(i, r2)) return 3;
        if (lr_region_kept(P, qlen, r)) { account(r); if (flag_only) { out.n_regs = 1; return 0; } }
        if (i > 0 && r.split_inv) {
            const LReg r1 = W.regs[i - 1];
            LReg rv;
            const int32_t ok = lr_align1_inv(C, r1, r, rv);
            if (ok < 0 || C.err) return C.need_big ? 1 : 3;
            if (ok > 0) {
                if (!insert_after(i, rv)) return 3;
                ++i;
                if (lr_region_kept(P, qlen, rv)) { account(rv); if (flag_only) { out.n_regs = 1; return 0; } }
            }
        }
    }
    out.n_regs = n_keep; out.dp_max = dp_best; out.sig = n_keep > 0 ? sig : 0u;
    return 0;
}
