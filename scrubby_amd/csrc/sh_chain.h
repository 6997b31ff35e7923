// sh_chain.h — seed filtering, anchor generation, chaining DP and backtrack (device code).
//
// Restates what happens between minimap2's sketch and its `n_regs0 > 0` for one
// single-segment query, i.e. the part of
//     aligner.map(&sequence, false, false, None, None)      /root/reference/src/cleaner.rs:552
// that decides `mappings.len() > 0` (:553) before base-level alignment.  Statement order
// follows the CPU oracle (oracle/mm_oracle.c) one to one so that every intermediate
// (n_anchor, rep_len, n_chain, best_score) is bit-identical; the f32 pair-score path needs
// -ffp-contract=off on both sides.
//
// One lane owns one read.  The code is generic over a Store (where the anchor / DP arrays
// live): SmallStore = LDS, 11 B per anchor, lane-interleaved; LargeStore = per-read slices
// of an HBM arena.
#pragma once
#include "sh_common.h"
#include "sh_wave.h"
#include "sh_align.h"

// Chain hand-over to the extension stage (sh_align.h): every backtrack below takes an emitter that is called once per ACCEPTED
// chain with (last anchor zi, stop index end_i (exclusive, -1 = root), score, anchors, f of zi).  NoEmit compiles to nothing.
struct NoEmit {
    __device__ inline void operator()(int64_t, int64_t, int32_t, int64_t, int32_t) const {}
    __device__ inline bool done(int32_t) const { return false; }      // done(zf): no chain ending in an anchor with f <= zf is wanted any more
};

struct ChainParams {
    int32_t k, is_sr;
    int32_t mid_occ, max_occ, max_max_occ, occ_dist;
    int32_t min_cnt, min_sc;
    int32_t max_gap, max_gap_ref, max_frag_len, bw;
    int32_t max_skip, max_iter;
    float pen_gap, pen_skip;
    float q_occ_frac;
    // Flag-only shortcut (host-computed, INT32_MAX = off): a cluster yields a mapping iff some anchor reaches f >= min_sc.
    // mg_chain_backtrack pops the maximum f first (nothing is marked yet); its walk either reaches the chain's root, where
    // the score is zf >= min_sc over >= ceil(min_sc / k) anchors (a link adds at most k), or stops on a drop > bw, where the
    // kept part scores > bw over > bw / k anchors.  With ceil(min_sc / k) >= min_cnt, bw >= min_sc and bw / k + 1 >= min_cnt
    // (true for every preset) that chain is always accepted, so the DP can stop at the first f >= flag_stop = min_sc and
    // no backtrack is needed; if no anchor gets there, there is no candidate at all.
    int32_t flag_stop;
    // Flag-only pair test (host-computed, pair_dq_max = 0: off; see pair_decides in sh_classify.hip): two anchors of one
    // strand / contig on one diagonal, pair_dq_min <= dq = dr <= pair_dq_max apart, decide the read when all selected seeds
    // have distinct keys (k_expand also takes key multiplicity M with dq <= (max_skip + 1) / M - 1).  A reference position holds at
    // most one minimizer, so at most dq - 1 <= max_skip - 2 anchors (M > 1: (dq + 1) * M - 2 <= max_skip - 1)
    // sort between the two: mg_lchain_dp's look-back from the later one cannot break (n_skip), run out (max_iter) or leave the
    // window before it scores the earlier one, with sc = min(k, dq) and no penalty (dd = 0, pen_skip = 0):
    // f >= k + min(k, dq) >= min_sc = flag_stop, which decides the cluster (above).
    int32_t pair_dq_min, pair_dq_max, pair_min_anchors;
    // SH_F_CIGAR shortcut of k_pair_pass (mode 2; host-computed, 0 = off): the premises that make a read of co-diagonal singleton
    // seeds a decided case - short-read mode, a * k >= min_dp_max, 2k >= min_chain_score, max_clip_ratio >= 1, no skip penalty;
    // ext_unc_max = zdrop / b, the bases outside the k-mers that cannot yet trigger mm_test_zdrop
    int32_t ext_s1, ext_unc_max;
    // SH_F_CIGAR, flag-only: a region passes mm_filter_regs WITHOUT any base-level work when its max stretch (mm_max_stretch: the
    // co-diagonal run of anchors with the largest covered length) has covered = k + sum min(k, d) >= min_chain_score, spans >= k
    // between its first and last anchor and leaves U = sum max(0, d - k) <= ext_unc_max bases outside the exact k-mers - the same
    // argument as k_pair_pass mode 2, applied to one region (chain_lemma below).  ext_lemma = the host-checked premises.
    int32_t ext_lemma;
    int32_t ext_a, ext_b, ext_amb, ext_zdrop;      // scores of the middle check (mm_test_zdrop over the ungapped stretch)
};

// what the middle check reads: the reference (4-bit codes) and the reads (ASCII)
struct BaseCtx { const uint8_t *ref; const uint64_t *cstart; const uint8_t *bases; };

// The stretch whose ungapped alignment has to be checked on the bases, and the check: mm_test_zdrop over query [qs, qe) (on strand
// rev of the read at `seq`) against reference [rs, rs + qe - qs) of contig rid.  true = no z-drop, i.e. mm_align1 keeps the single M
// (sh_align.h / oracle mm_align.c test_zdrop).  The WAVE does it, 64 bases at a time (coalesced byte loads, prefix sums on the lanes):
// max drop = max_j (max_{i <= j} S_i - S_j) over the running score S.  A per-lane loop over 150 dependent byte loads cost ~90 us a read.
struct MidReq { int32_t rid, rev, qs, qe, rs; };

__device__ inline bool middle_no_zdrop_wave(const BaseCtx &B, const uint8_t *seq, int32_t qlen, const MidReq &m, const ChainParams &P)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t g0 = B.cstart[m.rid] + (uint64_t)m.rs;
    int32_t carry_s = 0, carry_mx = INT32_MIN, zd = 0;
    for (int32_t j0 = 0; j0 < m.qe - m.qs; j0 += 64) {
        const int32_t j = j0 + (int32_t)lane;
        int32_t sc = 0;
        const bool on = j < m.qe - m.qs;
        if (on) {
            const int32_t qi = m.qs + j;
            uint32_t cq = sh_nt4(seq[m.rev ? qlen - 1 - qi : qi]);
            if (m.rev && cq < 4) cq = 3 - cq;
            const uint64_t g = g0 + (uint64_t)j;
            const uint32_t ct = (B.ref[g >> 1] >> ((g & 1) * 4)) & 15u;
            sc = (cq > 3 || ct > 3) ? P.ext_amb : (cq == ct ? P.ext_a : P.ext_b);
        }
        int32_t ps = wave_scan_add_incl(sc) + carry_s;    // inclusive prefix sum over the lanes
        int32_t pm = wave_scan_max_incl(on ? ps : INT32_MIN);      // inclusive prefix maximum, seeded with the chunks before
        pm = pm > carry_mx ? pm : carry_mx;
        const int32_t d = wave_all_max(on ? pm - ps : 0);
        zd = d > zd ? d : zd;
        const int32_t last = (m.qe - m.qs - j0) < 64 ? (m.qe - m.qs - j0) - 1 : 63;
        carry_s = wave_bcast(ps, last); carry_mx = wave_bcast(pm, last);
    }
    return zd <= P.ext_zdrop;
}

// Settles the stretches a wave's lanes could not vouch for from their k-mers alone: lane by lane, the whole wave on one request.
// need / req / seq / qlen are per lane; every lane of the wave must call.
__device__ inline bool resolve_mid_wave(bool need, const MidReq &req, const uint8_t *seq, int32_t qlen, const BaseCtx &B, const ChainParams &P)
{
    bool res = false;
    uint64_t todo = __ballot(need);
    const uint32_t lane = threadIdx.x & 63;
    while (todo) {
        const int l = __ffsll((unsigned long long)todo) - 1;
        todo &= todo - 1;
        MidReq m;
        m.rid = wave_bcast(req.rid, l); m.rev = wave_bcast(req.rev, l); m.qs = wave_bcast(req.qs, l); m.qe = wave_bcast(req.qe, l); m.rs = wave_bcast(req.rs, l);
        const uint64_t sp = (uint64_t)(uintptr_t)seq;
        const uint64_t p = wave_bcast_u64(sp, l);
        const bool ok = middle_no_zdrop_wave(B, (const uint8_t *)(uintptr_t)p, wave_bcast(qlen, l), m, P);
        if ((int)lane == l) res = ok;
    }
    return res;
}

// ---- seeds: one 16-B record per query minimizer found in the index ---------------------------
//   .x,.y = w1 of the slot (position word, or off<<28|n)   .z = n | SH_REC_PREV_SAME | flt<<31   .w = qpos<<1|strand
struct SeedView {
    uint4 *base;
    uint32_t stride;   // in records
    uint32_t n;
    __device__ inline uint4 get(uint32_t i) const { return base[(size_t)i * stride]; }
    __device__ inline uint32_t occ(uint32_t i) const { return base[(size_t)i * stride].z & SH_REC_OCC_MASK; }
    __device__ inline uint32_t qposz(uint32_t i) const { return base[(size_t)i * stride].w; }
    __device__ inline void set_flt(uint32_t i, uint32_t occ_, uint32_t flt) { uint32_t &z = base[(size_t)i * stride].z; z = (z & SH_REC_PREV_SAME) | occ_ | flt << 31; }
};

// mm_seed_select + the plain occurrence cut + the rep_len / anchor count loop of
// mm_collect_matches.  Sets the flt bit of every record.
__device__ inline void seed_filter(SeedView sv, int32_t qlen, int32_t max_occ, const ChainParams &P,
                                   int64_t &n_a_out, int32_t &rep_len_out)
{
    const int32_t n = (int32_t)sv.n;
    const bool use_select = P.occ_dist > 0 && P.max_max_occ > max_occ;
    if (!use_select) {
        for (int32_t i = 0; i < n; ++i) { uint32_t o = sv.occ(i); sv.set_flt(i, o, o > (uint32_t)max_occ); }
    } else {
        int32_t m = 0;
        for (int32_t i = 0; i < n; ++i) { uint32_t o = sv.occ(i); sv.set_flt(i, o, 0); m += o > (uint32_t)max_occ; }
        if (n >= 2 && m > 0) {
            int32_t last0 = -1;
            for (int32_t i = 0; i <= n; ++i) {
                bool low = (i == n) || sv.occ(i) <= (uint32_t)max_occ;
                if (!low) continue;
                if (i - last0 > 1) {
                    int32_t ps = last0 < 0 ? 0 : (int32_t)(sv.qposz(last0) >> 1);
                    int32_t pe = i == n ? qlen : (int32_t)(sv.qposz(i) >> 1);
                    int32_t st = last0 + 1, en = i;
                    int32_t mho = (int32_t)((double)(pe - ps) / (double)P.occ_dist + .499);
                    if (mho > 128) mho = 128;
                    for (int32_t j = st; j < en; ++j) {
                        uint32_t oj = sv.occ(j);
                        uint32_t flt = 1;
                        if (mho > 0) {      // keep the mho smallest by (occ, index)
                            int32_t rank = 0;
                            for (int32_t t = st; t < en; ++t) {
                                uint32_t ot = sv.occ(t);
                                rank += (ot < oj) || (ot == oj && t < j);
                            }
                            if (rank < mho) flt = 0;
                        }
                        if (oj > (uint32_t)P.max_max_occ) flt = 1;
                        sv.set_flt(j, oj, flt);
                    }
                }
                last0 = i;
            }
        }
    }
    int64_t n_a = 0;
    int32_t rep_st = 0, rep_en = 0, rep_len = 0;
    for (int32_t i = 0; i < n; ++i) {
        uint4 s = sv.get(i);
        if (s.z >> 31) {
            int32_t en = (int32_t)(s.w >> 1) + 1, st = en - P.k;
            if (st > rep_en) { rep_len += rep_en - rep_st; rep_st = st; rep_en = en; }
            else rep_en = en;
        } else n_a += s.z & SH_REC_OCC_MASK;
    }
    rep_len += rep_en - rep_st;
    n_a_out = n_a; rep_len_out = rep_len;
}

// anchor (x, qpos') of occurrence r of a seed at qposz
__device__ inline void make_anchor(uint64_t r, uint32_t qposz, int32_t qlen, int32_t k, uint64_t &x, uint32_t &q)
{
    uint32_t rpos = (uint32_t)r >> 1;
    if ((r & 1) == (qposz & 1)) {
        x = (r & 0xffffffff00000000ULL) | rpos;
        q = qposz >> 1;
    } else {
        x = 1ULL << 63 | (r & 0xffffffff00000000ULL) | rpos;
        q = (uint32_t)(qlen - ((int32_t)(qposz >> 1) + 1 - k) - 1);
    }
}

// ---- pair score (mg_lchain_dp's comput_sc, single segment, not cDNA) ---------------------------
__device__ inline float sh_mg_log2(float x)
{
    uint32_t zi = __float_as_uint(x);
    float log_2 = (float)((int32_t)((zi >> 23) & 255) - 128);
    zi &= ~(255u << 23);
    zi += 127u << 23;
    float zf = __uint_as_float(zi);
    log_2 += (-0.34484843f * zf + 2.02466578f) * zf - 0.67487759f;
    return log_2;
}

#define SH_SC_NONE INT32_MIN
__device__ inline int32_t comput_sc(uint32_t lo_i, uint32_t q_i, uint32_t lo_j, uint32_t q_j, int32_t max_dist_x,
                                    int32_t max_dist_y, const ChainParams &P)
{
    // branch-free: the lanes of a wave disagree on every early exit of the original
    const int32_t dq = (int32_t)q_i - (int32_t)q_j, dr = (int32_t)(lo_i - lo_j);
    int32_t dd = dr > dq ? dr - dq : dq - dr;
    const bool ok = dq > 0 && dq <= max_dist_x && dr != 0 && dq <= max_dist_y && dd <= P.bw;
    dd = ok ? dd : 0;
    int32_t dg = dr < dq ? dr : dq;
    dg = ok ? dg : 0;
    int32_t sc = P.k < dg ? P.k : dg;
    const float lin_pen = P.pen_gap * (float)dd + P.pen_skip * (float)dg;
    const float log_pen = dd >= 1 ? sh_mg_log2((float)(dd + 1)) : 0.0f;
    const int32_t pen = (int32_t)(lin_pen + .5f * log_pen);
    sc -= (dd != 0 || dg > P.k) ? pen : 0;
    return ok ? sc : SH_SC_NONE;
}

// ---- stores -----------------------------------------------------------------------------------
// LDS, lane-interleaved [i][64]; 11 B per anchor.  `aux` holds x>>32 while sorting, then
// {f:16, p:8, t:8}; group ids (rank of x>>32 among the read's anchors) replace x>>32.
template <int CAP>
struct SmallStore {
    uint32_t *lo; uint32_t *aux; uint16_t *qv; uint8_t *gv;     // already offset by lane
    static constexpr int S = 64;
    __device__ inline void set_raw(int i, uint64_t x, uint32_t q) { lo[i * S] = (uint32_t)x; aux[i * S] = (uint32_t)(x >> 32); qv[i * S] = (uint16_t)q; }
    __device__ inline uint64_t raw_x(int i) const { return (uint64_t)aux[i * S] << 32 | lo[i * S]; }
    __device__ inline void sort_finalize(int n)
    {   // stable insertion sort by x (radix_sort_128x is an insertion sort below 64 elements)
        for (int i = 1; i < n; ++i) {
            uint64_t x = raw_x(i);
            if (x >= raw_x(i - 1)) continue;
            uint32_t q = qv[i * S];
            int j = i;
            for (; j > 0 && x < raw_x(j - 1); --j) { lo[j * S] = lo[(j - 1) * S]; aux[j * S] = aux[(j - 1) * S]; qv[j * S] = qv[(j - 1) * S]; }
            set_raw(j, x, q);
        }
        uint32_t prev = n > 0 ? aux[0] : 0;
        uint8_t g = 0;
        for (int i = 0; i < n; ++i) { uint32_t h = aux[i * S]; g += h != prev; prev = h; gv[i * S] = g; }
    }
    __device__ inline uint32_t grp(int i) const { return gv[i * S]; }
    __device__ inline uint32_t rlo(int i) const { return lo[i * S]; }
    __device__ inline uint32_t qp(int i) const { return qv[i * S]; }
    __device__ inline int32_t F(int i) const { return (int32_t)(aux[i * S] & 0xffffu); }
    __device__ inline int32_t Pm(int i) const { uint32_t v = (aux[i * S] >> 16) & 0xffu; return v == 0xffu ? -1 : (int32_t)v; }
    __device__ inline int32_t T(int i) const { return (int32_t)(aux[i * S] >> 24); }
    __device__ inline void setFP(int i, int32_t f, int32_t p) { aux[i * S] = (aux[i * S] & 0xff000000u) | ((uint32_t)(p & 0xff) << 16) | ((uint32_t)f & 0xffffu); }
    __device__ inline void setT(int i, int32_t t) { aux[i * S] = (aux[i * S] & 0x00ffffffu) | (uint32_t)t << 24; }
    __device__ inline void clearT(int n) { for (int i = 0; i < n; ++i) aux[i * S] &= 0x00ffffffu; }
    __device__ inline void clearAux(int n) { for (int i = 0; i < n; ++i) aux[i * S] = 0; }
};

// HBM arena slices of one read
struct LargeStore {
    uint64_t *x, *x2, *z; uint32_t *q, *q2; int32_t *f, *p, *t;
    static __host__ __device__ inline size_t bytes_for(int64_t n) { return (size_t)(n + 2) * (8 + 8 + 8 + 4 + 4 + 4 + 4 + 4); }
    __device__ inline void carve(uint8_t *m, int64_t n)
    {
        size_t n2 = (size_t)(n + 2);
        x = (uint64_t *)m; x2 = x + n2; z = x2 + n2; q = (uint32_t *)(z + n2); q2 = q + n2;
        f = (int32_t *)(q2 + n2); p = f + n2; t = p + n2;
    }
    __device__ inline void set_raw(int64_t i, uint64_t xv, uint32_t qv_) { x[i] = xv; q[i] = qv_; }
    __device__ inline void sort_finalize(int64_t n)
    {   // stable bottom-up merge sort by x, ping-pong between (x,q) and (x2,q2)
        bool sorted = true;
        for (int64_t i = 1; i < n; ++i) if (x[i] < x[i - 1]) { sorted = false; break; }
        if (sorted) return;
        uint64_t *sx = x, *dx = x2; uint32_t *sq = q, *dq = q2;
        for (int64_t width = 1; width < n; width <<= 1) {
            for (int64_t i = 0; i < n; i += 2 * width) {
                int64_t m = i + width < n ? i + width : n, r = i + 2 * width < n ? i + 2 * width : n;
                int64_t a = i, b = m, o = i;
                while (a < m && b < r) { if (sx[b] < sx[a]) { dx[o] = sx[b]; dq[o] = sq[b]; ++b; } else { dx[o] = sx[a]; dq[o] = sq[a]; ++a; } ++o; }
                while (a < m) { dx[o] = sx[a]; dq[o] = sq[a]; ++a; ++o; }
                while (b < r) { dx[o] = sx[b]; dq[o] = sq[b]; ++b; ++o; }
            }
            uint64_t *tx = sx; sx = dx; dx = tx; uint32_t *tq = sq; sq = dq; dq = tq;
        }
        if (sx != x) { uint64_t *tx = x; x = x2; x2 = tx; uint32_t *tq = q; q = q2; q2 = tq; }
    }
    __device__ inline uint32_t grp(int64_t i) const { return (uint32_t)(x[i] >> 32); }
    __device__ inline uint64_t X(int64_t i) const { return x[i]; }
    __device__ inline uint32_t rlo(int64_t i) const { return (uint32_t)x[i]; }
    __device__ inline uint32_t qp(int64_t i) const { return q[i]; }
    __device__ inline int32_t F(int64_t i) const { return f[i]; }
    __device__ inline int32_t Pm(int64_t i) const { return p[i]; }
    __device__ inline int32_t T(int64_t i) const { return t[i]; }
    __device__ inline void setFP(int64_t i, int32_t fv, int32_t pv) { f[i] = fv; p[i] = pv; }
    __device__ inline void setT(int64_t i, int32_t tv) { t[i] = tv; }
    __device__ inline void clearT(int64_t n) { for (int64_t i = 0; i < n; ++i) t[i] = 0; }
    __device__ inline void clearAux(int64_t n) { clearT(n); }
};

// expand the unfiltered seeds into anchors (seed order, then occurrence order), sort by x
template <class Store>
__device__ inline void gen_anchors(Store &S, SeedView sv, const uint64_t *__restrict__ positions, int32_t qlen, int32_t k)
{
    int64_t na = 0;
    for (uint32_t i = 0; i < sv.n; ++i) {
        uint4 s = sv.get(i);
        if (s.z >> 31) continue;
        uint32_t occ = s.z & SH_REC_OCC_MASK;
        uint64_t w1 = (uint64_t)s.y << 32 | s.x;
        if (occ == 1) {
            uint64_t x; uint32_t q;
            make_anchor(w1, s.w, qlen, k, x, q);
            S.set_raw(na++, x, q);
        } else {
            const uint64_t *cr = positions + (w1 >> SH_SLOT_NBITS);
            for (uint32_t t = 0; t < occ; ++t) {
                uint64_t x; uint32_t q;
                make_anchor(cr[t], s.w, qlen, k, x, q);
                S.set_raw(na++, x, q);
            }
        }
    }
    S.sort_finalize(na);
}

// mg_lchain_dp: fills f/p; t is scratch
template <class Store, class Idx>
__device__ inline bool chain_dp(Store &S, Idx n, int32_t qlen, const ChainParams &P, int32_t stop_at = INT32_MAX)
{
    int32_t max_dist_y = P.is_sr ? (qlen > P.max_gap ? qlen : P.max_gap) : P.max_gap;
    int32_t max_dist_x;
    if (P.max_gap_ref > 0) max_dist_x = P.max_gap_ref;
    else if (P.max_frag_len > 0) { max_dist_x = P.max_frag_len - qlen; if (max_dist_x < P.max_gap) max_dist_x = P.max_gap; }
    else max_dist_x = P.max_gap;
    if (max_dist_x < P.bw) max_dist_x = P.bw;
    if (max_dist_y < P.bw) max_dist_y = P.bw;

    S.clearAux(n);
    Idx st = 0, max_ii = -1;
    for (Idx i = 0; i < n; ++i) {
        Idx max_j = -1, j;
        const uint32_t gi = S.grp(i), li = S.rlo(i), qi = S.qp(i);
        int32_t max_f = P.k, n_skip = 0;
        while (st < i && (gi != S.grp(st) || (uint64_t)li > (uint64_t)S.rlo(st) + (uint64_t)max_dist_x)) ++st;
        if (i - st > (Idx)P.max_iter) st = i - (Idx)P.max_iter;
        for (j = i - 1; j >= st; --j) {
            int32_t sc = comput_sc(li, qi, S.rlo(j), S.qp(j), max_dist_x, max_dist_y, P);
            if (sc == SH_SC_NONE) continue;
            sc += S.F(j);
            if (sc > max_f) {
                max_f = sc; max_j = j;
                if (n_skip > 0) --n_skip;
            } else if (S.T(j) == (int32_t)i) {
                if (++n_skip > P.max_skip) break;
            }
            int32_t pj = S.Pm(j);
            if (pj >= 0) S.setT(pj, (int32_t)i);
        }
        Idx end_j = j;
        bool far = true;
        if (max_ii >= 0) far = (gi != S.grp(max_ii)) || ((uint64_t)(li - S.rlo(max_ii)) > (uint64_t)max_dist_x);
        if (max_ii < 0 || far) {
            int32_t mx = INT32_MIN;
            max_ii = -1;
            for (j = i - 1; j >= st; --j) { int32_t fj = S.F(j); if (mx < fj) { mx = fj; max_ii = j; } }
        }
        if (max_ii >= 0 && max_ii < end_j) {
            int32_t tmp = comput_sc(li, qi, S.rlo(max_ii), S.qp(max_ii), max_dist_x, max_dist_y, P);
            if (tmp != SH_SC_NONE && max_f < tmp + S.F(max_ii)) { max_f = tmp + S.F(max_ii); max_j = max_ii; }
        }
        S.setFP(i, max_f, (int32_t)max_j);
        if (max_f >= stop_at) return true;
        bool near = false;
        if (max_ii >= 0) near = (gi == S.grp(max_ii)) && ((uint64_t)(li - S.rlo(max_ii)) <= (uint64_t)max_dist_x);
        if (max_ii < 0 || (near && S.F(max_ii) < max_f)) max_ii = i;
    }
    return false;
}

// mg_chain_bk_end
template <class Store, class Idx>
__device__ inline Idx chain_bk_end(Store &S, int32_t max_drop, int32_t zf, Idx zi)
{
    Idx i = zi, end_i = -1, max_i = i;
    int32_t max_s = 0;
    if (i < 0 || S.T(i) != 0) return i;
    do {
        S.setT(i, 2);
        end_i = i = (Idx)S.Pm(i);
        int32_t s = i < 0 ? zf : zf - S.F(i);
        if (s > max_s) { max_s = s; max_i = i; }
        else if (max_s - s > max_drop) break;
    } while (i >= 0 && S.T(i) == 0);
    for (i = zi; i >= 0 && i != end_i; i = (Idx)S.Pm(i)) S.setT(i, 0);
    return max_i;
}

// one candidate of mg_chain_backtrack's first loop
template <class Store, class Idx, class EM = NoEmit>
__device__ inline void backtrack_visit(Store &S, const ChainParams &P, int32_t zf, Idx zi, int64_t &n_v, int32_t &n_u, int32_t &best, const EM &em = EM())
{
    if (S.T(zi) != 0) return;
    int64_t n_v0 = n_v;
    Idx end_i = chain_bk_end<Store, Idx>(S, P.bw, zf, zi), i;
    for (i = zi; i != end_i; i = (Idx)S.Pm(i)) { ++n_v; S.setT(i, 1); }
    int32_t sc = i < 0 ? zf : zf - S.F(i);
    if (sc >= P.min_sc && n_v > n_v0 && n_v - n_v0 >= P.min_cnt) { ++n_u; if (sc > best) best = sc; em((int64_t)zi, (int64_t)end_i, sc, n_v - n_v0, zf); }
    else n_v = n_v0;
}

// candidates (f >= min_sc) in descending (f, index) order; small n: O(n^2) selection, no extra memory.
// A candidate already absorbed by an earlier chain (t != 0) is a no-op in mg_chain_backtrack, so the scan
// skips it; with one dominant chain per cluster this ends after two scans.  first_only: stop at the first
// accepted chain (the caller only needs "is there a mapping").
template <class Store>
__device__ inline void backtrack_small(Store &S, int n, const ChainParams &P, int32_t &n_u, int32_t &best, bool first_only = false)
{
    n_u = 0; best = 0;
    S.clearT(n);
    int64_t n_v = 0;
    int64_t bound = INT64_MAX;
    for (;;) {
        int64_t cur = -1;
        for (int i = 0; i < n; ++i) {
            int32_t f = S.F(i);
            if (f < P.min_sc || S.T(i) != 0) continue;
            int64_t key = (int64_t)f << 32 | (uint32_t)i;
            if (key < bound && key > cur) cur = key;
        }
        if (cur < 0) break;
        bound = cur;
        backtrack_visit<Store, int>(S, P, (int32_t)(cur >> 32), (int)(cur & 0xffffffff), n_v, n_u, best);
        if (first_only && n_u > 0) break;
    }
}

// ---- n <= 64: the t[] array of mg_lchain_dp / mg_chain_backtrack lives in one 64-bit register ---------------
// In the DP, t[j] == i only asks "was j the predecessor of an anchor already visited in THIS scan", so a mask
// reset for every i is equivalent.  In the backtrack the transient value 2 never survives a call of
// mg_chain_bk_end (it is set and cleared on the same path), so only t == 1 persists: one mask again.
// Without the LDS store -> load dependency through t[] the j loop's loads are independent and pipeline.
template <class Store>
__device__ inline bool chain_dp_mask(Store &S, int n, int32_t qlen, const ChainParams &P, int32_t stop_at = INT32_MAX)
{
    int32_t max_dist_y = P.is_sr ? (qlen > P.max_gap ? qlen : P.max_gap) : P.max_gap;
    int32_t max_dist_x;
    if (P.max_gap_ref > 0) max_dist_x = P.max_gap_ref;
    else if (P.max_frag_len > 0) { max_dist_x = P.max_frag_len - qlen; if (max_dist_x < P.max_gap) max_dist_x = P.max_gap; }
    else max_dist_x = P.max_gap;
    if (max_dist_x < P.bw) max_dist_x = P.bw;
    if (max_dist_y < P.bw) max_dist_y = P.bw;

    int st = 0, max_ii = -1;
    for (int i = 0; i < n; ++i) {
        int max_j = -1, j;
        const uint32_t gi = S.grp(i), li = S.rlo(i), qi = S.qp(i);
        int32_t max_f = P.k, n_skip = 0;
        uint64_t marked = 0;
        while (st < i && (gi != S.grp(st) || (uint64_t)li > (uint64_t)S.rlo(st) + (uint64_t)max_dist_x)) ++st;
        if (i - st > P.max_iter) st = i - P.max_iter;
#pragma unroll 2
        for (j = i - 1; j >= st; --j) {
            const int32_t fj = S.F(j), pj = S.Pm(j);
            int32_t sc = comput_sc(li, qi, S.rlo(j), S.qp(j), max_dist_x, max_dist_y, P);
            if (sc == SH_SC_NONE) continue;
            sc += fj;
            if (sc > max_f) {
                max_f = sc; max_j = j;
                if (n_skip > 0) --n_skip;
            } else if ((marked >> j) & 1) {
                if (++n_skip > P.max_skip) break;
            }
            if (pj >= 0) marked |= 1ULL << pj;
        }
        const int end_j = j;
        bool far = true;
        if (max_ii >= 0) far = (gi != S.grp(max_ii)) || ((uint64_t)(li - S.rlo(max_ii)) > (uint64_t)max_dist_x);
        if (max_ii < 0 || far) {
            int32_t mx = INT32_MIN;
            max_ii = -1;
            for (j = i - 1; j >= st; --j) { int32_t fj = S.F(j); if (mx < fj) { mx = fj; max_ii = j; } }
        }
        if (max_ii >= 0 && max_ii < end_j) {
            int32_t tmp = comput_sc(li, qi, S.rlo(max_ii), S.qp(max_ii), max_dist_x, max_dist_y, P);
            if (tmp != SH_SC_NONE && max_f < tmp + S.F(max_ii)) { max_f = tmp + S.F(max_ii); max_j = max_ii; }
        }
        S.setFP(i, max_f, max_j);
        if (max_f >= stop_at) return true;
        bool near = false;
        if (max_ii >= 0) near = (gi == S.grp(max_ii)) && ((uint64_t)(li - S.rlo(max_ii)) <= (uint64_t)max_dist_x);
        if (max_ii < 0 || (near && S.F(max_ii) < max_f)) max_ii = i;
    }
    return false;
}

__device__ inline uint32_t chain_max_dist_x(const ChainParams &P, int32_t qlen)
{
    int32_t m;
    if (P.max_gap_ref > 0) m = P.max_gap_ref;
    else if (P.max_frag_len > 0) { m = P.max_frag_len - qlen; if (m < P.max_gap) m = P.max_gap; }
    else m = P.max_gap;
    if (m < P.bw) m = P.bw;
    return (uint32_t)m;
}
__device__ inline int32_t chain_max_dist_y(const ChainParams &P, int32_t qlen)
{
    int32_t m = P.is_sr ? (qlen > P.max_gap ? qlen : P.max_gap) : P.max_gap;
    return m < P.bw ? P.bw : m;
}

// ---- mg_lchain_dp without its sequential state: all anchors of a read at once ---------------------------------------------------------
// mg_lchain_dp carries three pieces of state from anchor to anchor: the t[] marks with the n_skip counter (they END a look-back scan early,
// after max_skip marked non-maxima), the max_iter cut of the window, and max_ii (a shortcut that is only consulted when the scan ended
// early or was cut: `max_ii < end_j`).  n_skip only moves on predecessors j whose comput_sc is valid, so an anchor with at most max_skip
// valid predecessors inside an uncut window scans all of it, never consults max_ii, and
//     f[i] = max(k, max over the valid j in [st, i) of f[j] + sc(i, j)),     p[i] = the largest such j attaining it (strictly above k), else -1
// - a recurrence without hidden state.  A valid link needs dq > 0, so the anchors of one query position are independent of each other and
// depend on smaller query positions only: the read's anchors are ordered by the rank of their query position (counting sort, the
// permutation in the t slots) and each rank is one parallel round over ALL clusters of the read, one thread per anchor.  An anchor that
// breaks the premise (more than max_skip valid predecessors, or a window beyond max_iter) makes its CLUSTER dirty: clusters are
// independent DP problems, a dirty one is chained by the sequential code as before, the others keep what was computed here.  On the bench
// workload 472 of 1.6 M clusters of more than 64 anchors are dirty.
// Predecessors are stored as indices into the read's array (SliceStore::pbase turns them into the slice-relative ones the backtrack uses).
#define PF_MAX_Q 1024
#define PF_MAX_RANK 128
#define PF_DIRTY_CAP 64
#define PF_DIRTY 0x7ffffff1
#define PF_DEAD 0x7ffffff2
struct ParFillLds { unsigned long long qmask[PF_MAX_Q / 64]; uint32_t qpre[PF_MAX_Q / 64 + 1]; uint32_t start[PF_MAX_RANK + 1], cur[PF_MAX_RANK]; int32_t n_dirty; uint32_t dirty[PF_DIRTY_CAP]; };

// x, q: the read's sorted anchors with the cluster starts marked in bit 31 of q; f, pt: DP state (pt = p, t interleaved).  All threads of the
// block call it.  false: not applicable (query positions beyond PF_MAX_Q, more than PF_MAX_RANK of them, too many dirty anchors) - nothing usable
// was written.  true: f and p hold mg_lchain_dp's values for every clean cluster, t = 0 except PF_DIRTY at the first anchor of a dirty one.
template <class PX, class PQ>
__device__ inline bool par_fill_block(PX x, PQ q, int32_t *f, int32_t *pt, uint32_t n, int32_t qlen, const ChainParams &P, uint32_t tid, uint32_t nthr, ParFillLds &L)
{
    if (qlen > PF_MAX_Q || n >= 0x7ffffff0u) return false;
    const int32_t mdy = chain_max_dist_y(P, qlen), mdx = (int32_t)chain_max_dist_x(P, qlen);
    for (uint32_t t = tid; t < PF_MAX_Q / 64; t += nthr) L.qmask[t] = 0;
    if (tid == 0) L.n_dirty = 0;
    __syncthreads();
    for (uint32_t i = tid; i < n; i += nthr) {
        const uint32_t qq = (uint32_t)q[i] & 0x7fffffffu;
        if (!((L.qmask[qq >> 6] >> (qq & 63)) & 1ull)) atomicOr(&L.qmask[qq >> 6], 1ull << (qq & 63));
    }
    __syncthreads();
    if (tid == 0) { uint32_t acc = 0; for (int w = 0; w < PF_MAX_Q / 64; ++w) { L.qpre[w] = acc; acc += (uint32_t)__popcll(L.qmask[w]); } L.qpre[PF_MAX_Q / 64] = acc; }
    __syncthreads();
    const uint32_t R = L.qpre[PF_MAX_Q / 64];
    if (R > PF_MAX_RANK) return false;
    for (uint32_t t = tid; t <= R; t += nthr) L.start[t] = 0;
    __syncthreads();
    auto rank_of = [&](uint32_t qq) { return L.qpre[qq >> 6] + (uint32_t)__popcll(L.qmask[qq >> 6] & ((1ull << (qq & 63)) - 1ull)); };
    for (uint32_t i = tid; i < n; i += nthr) atomicAdd(&L.start[rank_of((uint32_t)q[i] & 0x7fffffffu) + 1], 1u);
    __syncthreads();
    if (tid == 0) { uint32_t acc = 0; for (uint32_t r = 0; r < R; ++r) { acc += L.start[r + 1]; L.start[r + 1] = acc; } }
    __syncthreads();
    for (uint32_t t = tid; t < R; t += nthr) L.cur[t] = L.start[t];
    __syncthreads();
    for (uint32_t i = tid; i < n; i += nthr) { const uint32_t slot = atomicAdd(&L.cur[rank_of((uint32_t)q[i] & 0x7fffffffu)], 1u); pt[2 * (size_t)slot + 1] = (int32_t)i; }
    __syncthreads();
    for (uint32_t r = 0; r < R; ++r) {
        const uint32_t e = L.start[r + 1];
        for (uint32_t idx = L.start[r] + tid; idx < e; idx += nthr) {
            const uint32_t i = (uint32_t)pt[2 * (size_t)idx + 1];
            const uint32_t qraw = (uint32_t)q[i], qi = qraw & 0x7fffffffu, xi = (uint32_t)x[i];
            int32_t max_f = P.k, max_j = -1, nv = 0;
            bool bad = false;
            if (r > 0 && !(qraw >> 31)) {
                // a valid predecessor has 0 < dq <= q_i and |dr - dq| <= bw, so it lies within q_i + bw of x_i: the scan stops there (what lies
                // between that and max_dist_x only matters for the max_iter cut, tested on its own)
                const uint32_t lim = min((uint32_t)mdx, qi + (uint32_t)P.bw);
                if (i > (uint32_t)P.max_iter) { const uint64_t xo = x[i - (uint32_t)P.max_iter - 1u], xf = x[i]; bad = (xo >> 32) == (xf >> 32) && (uint32_t)xf - (uint32_t)xo <= (uint32_t)mdx; }
                for (uint32_t j = i - 1;; --j) {
                    const uint32_t qj = (uint32_t)q[j], xj = (uint32_t)x[j];
                    if (xi - xj > lim) break;
                    const int32_t sc = comput_sc(xi, qi, xj, qj & 0x7fffffffu, mdx, mdy, P);
                    if (sc != SH_SC_NONE) { if (++nv > P.max_skip) break; const int32_t c = sc + f[j]; if (c > max_f) { max_f = c; max_j = (int32_t)j; } }      // dirty: no need to see the rest
                    if (qj >> 31) break;
                }
            }
            f[i] = max_f; pt[2 * (size_t)i] = max_j;
            if (bad || nv > P.max_skip) { const int32_t d = atomicAdd(&L.n_dirty, 1); if (d < PF_DIRTY_CAP) L.dirty[d] = i; }
        }
        __syncthreads();
        if (L.n_dirty > PF_DIRTY_CAP) return false;      // uniform (read after the barrier): a read of dense tandem arrays - every window overflows
    }
    const int32_t nd = L.n_dirty;
    for (uint32_t i = tid; i < n; i += nthr) pt[2 * (size_t)i + 1] = 0;
    __syncthreads();
    for (int32_t d = (int32_t)tid; d < nd; d += (int32_t)nthr) { uint32_t c = L.dirty[d]; while (!((uint32_t)q[c] >> 31)) --c; pt[2 * (size_t)c + 1] = PF_DIRTY; }
    __syncthreads();
    return true;
}

// par_fill_block for reads whose anchors live in HBM (tens of thousands per read): the rounds sweep a read's data once per rank, and with
// hundreds of reads in flight that data does not stay in the L2 between sweeps - measured, the kernel moved ~20 x its input.  The DP only
// looks back a few dozen anchors, so the read is cut into tiles of PFT_T consecutive anchors that are chained one after the other, each
// completely in LDS (its own rank table, all rounds), with the last PFT_H anchors of the tile before it as a halo; f and (p, t = 0) go
// out once, coalesced.  An anchor whose window is not inside tile + halo is dirty (so is one with PFT_H anchors within max_dist_x behind
// it: the max_iter cut is not looked at any closer than that; needs max_iter >= PFT_H).  Same contract as par_fill_block.
__device__ inline int32_t wave_scan_max_incl(int32_t v);      // below, with the other DPP scans
#define PFT_T 4096
#define PFT_H 256
struct PfTile { uint32_t x[PFT_H + PFT_T], q[PFT_H + PFT_T]; int32_t f[PFT_H + PFT_T]; uint16_t p[PFT_T], perm[PFT_T], dcs[PFT_H + PFT_T]; int32_t wtot[16], carry; };

__device__ inline bool par_fill_tiled(const uint64_t *x, const uint32_t *q, int32_t *f, int32_t *pt, uint32_t n, int32_t qlen, const ChainParams &P, uint32_t tid, uint32_t nthr,
                                      ParFillLds &L, PfTile &T, uint32_t g_min = 32768u)
{
    if (qlen > PF_MAX_Q || n >= 0x7ffffff0u || P.max_iter < PFT_H || nthr < PFT_H || nthr > 1024) return false;
    const int32_t mdy = chain_max_dist_y(P, qlen), mdx = (int32_t)chain_max_dist_x(P, qlen);
    // The largest reads of a batch (10^5 anchors of a dense tandem array, hundreds of them inside every window) set the kernel's time however
    // small the batch: for those G = 8 lanes share an anchor, each taking every eighth predecessor (combined as the sequential scan would
    // have decided: largest sum, of equal sums the largest index).  Every lane must then know where its anchor's cluster starts without
    // walking there: dcs = distance to the cluster start, from a block-wide scan of the marks.
    const uint32_t G = n >= g_min ? 8u : 1u, sub = tid & (G - 1u), grp = tid / G, n_grp = nthr / G;
    const uint32_t lane = tid & 63u, wave = tid >> 6, n_wave = nthr >> 6;
    if (tid == 0) L.n_dirty = 0;
    uint32_t h = 0;
    for (uint32_t t0 = 0; t0 < n; t0 += PFT_T) {
        const uint32_t tn = n - t0 < PFT_T ? n - t0 : PFT_T;
        for (uint32_t t = tid; t < PF_MAX_Q / 64; t += nthr) L.qmask[t] = 0;
        if (tid == 0) T.carry = -1;
        __syncthreads();
        for (uint32_t k = tid; k < tn; k += nthr) {
            const uint32_t qr = q[t0 + k], qq = qr & 0x7fffffffu;
            T.x[h + k] = (uint32_t)x[t0 + k]; T.q[h + k] = qr;
            if (!((L.qmask[qq >> 6] >> (qq & 63)) & 1ull)) atomicOr(&L.qmask[qq >> 6], 1ull << (qq & 63));
        }
        __syncthreads();
        if (tid == 0) { uint32_t acc = 0; for (int w = 0; w < PF_MAX_Q / 64; ++w) { L.qpre[w] = acc; acc += (uint32_t)__popcll(L.qmask[w]); } L.qpre[PF_MAX_Q / 64] = acc; }
        if (G > 1) {      // dcs over halo + tile (an anchor whose cluster starts before the halo counts from index 0: the scan then runs out of halo)
            for (uint32_t c0 = 0; c0 < h + tn; c0 += nthr) {
                const uint32_t k = c0 + tid;
                const int32_t v = k < h + tn && (T.q[k] >> 31) ? (int32_t)k : -1;
                int32_t inc = wave_scan_max_incl(v);
                if (lane == 63) T.wtot[wave] = inc;
                __syncthreads();
                int32_t before = T.carry;
                for (uint32_t w = 0; w < wave; ++w) before = T.wtot[w] > before ? T.wtot[w] : before;
                inc = inc > before ? inc : before;
                if (k < h + tn) T.dcs[k] = (uint16_t)(inc < 0 ? k : k - (uint32_t)inc);
                __syncthreads();
                if (tid == nthr - 1) T.carry = inc;
            }
        }
        __syncthreads();
        const uint32_t R = L.qpre[PF_MAX_Q / 64];
        if (R > PF_MAX_RANK) return false;
        for (uint32_t t = tid; t <= R; t += nthr) L.start[t] = 0;
        __syncthreads();
        auto rank_of = [&](uint32_t qq) { return L.qpre[qq >> 6] + (uint32_t)__popcll(L.qmask[qq >> 6] & ((1ull << (qq & 63)) - 1ull)); };
        for (uint32_t k = tid; k < tn; k += nthr) atomicAdd(&L.start[rank_of(T.q[h + k] & 0x7fffffffu) + 1], 1u);
        __syncthreads();
        if (tid == 0) { uint32_t acc = 0; for (uint32_t r = 0; r < R; ++r) { acc += L.start[r + 1]; L.start[r + 1] = acc; } }
        __syncthreads();
        for (uint32_t t = tid; t < R; t += nthr) L.cur[t] = L.start[t];
        __syncthreads();
        for (uint32_t k = tid; k < tn; k += nthr) T.perm[atomicAdd(&L.cur[rank_of(T.q[h + k] & 0x7fffffffu)], 1u)] = (uint16_t)k;
        __syncthreads();
        for (uint32_t r = 0; r < R; ++r) {
            const uint32_t e = L.start[r + 1];
            for (uint32_t base = L.start[r]; base < e; base += n_grp) {
                const uint32_t idx = base + grp;
                const bool on = idx < e;
                const uint32_t kt = on ? T.perm[idx] : 0u, k = h + kt;
                const uint32_t qraw = T.q[k], qi = qraw & 0x7fffffffu, xi = T.x[k];
                int32_t max_f = P.k, max_j = -1, nv = 0;
                bool bad = false;
                if (on && !(qraw >> 31)) {
                    const uint32_t lim = min((uint32_t)mdx, qi + (uint32_t)P.bw);
                    if (sub == 0 && k >= PFT_H && xi - T.x[k - PFT_H] <= (uint32_t)mdx) bad = true;      // a window of PFT_H anchors or more: left to the sequential code
                    if (G == 1) {
                        bool stop = false;
                        for (int32_t j = (int32_t)k - 1; j >= 0; --j) {
                            const uint32_t qj = T.q[j], xj = T.x[j];
                            if (xi - xj > lim) { stop = true; break; }
                            const int32_t sc = comput_sc(xi, qi, xj, qj & 0x7fffffffu, mdx, mdy, P);
                            if (sc != SH_SC_NONE) { if (++nv > P.max_skip) { stop = true; break; } const int32_t c = sc + T.f[j]; if (c > max_f) { max_f = c; max_j = j; } }
                            if (qj >> 31) { stop = true; break; }
                        }
                        if (!stop && t0 > h) bad = true;      // ran out of halo
                    } else {
                        const int32_t jmin = (int32_t)k - (int32_t)T.dcs[k];
                        bool stop = false;
                        for (int32_t j = (int32_t)k - 1 - (int32_t)sub; j >= jmin; j -= (int32_t)G) {
                            const uint32_t xj = T.x[j];
                            if (xi - xj > lim) { stop = true; break; }
                            const int32_t sc = comput_sc(xi, qi, xj, T.q[j] & 0x7fffffffu, mdx, mdy, P);
                            if (sc != SH_SC_NONE) { if (++nv > P.max_skip) { stop = true; break; } const int32_t c = sc + T.f[j]; if (c > max_f) { max_f = c; max_j = j; } }
                        }
                        // the cluster starts before the halo and this lane never met the distance limit
                        if (!stop && jmin == 0 && t0 > h && !(T.q[0] >> 31)) bad = true;
                    }
                }
                if (G == 8u) {      // uniform: G is.  The eight lanes of an anchor combine over DPP (neighbour, other pair, other quad)
#define PF_COMBINE(CTRL) { const int32_t of = dpp_mov<CTRL, 0xf>(max_f, max_f), oj = dpp_mov<CTRL, 0xf>(max_j, max_j); \
                           nv += dpp_mov<CTRL, 0xf>(nv, nv); int32_t bi = (int32_t)bad; bi |= dpp_mov<CTRL, 0xf>(bi, bi); bad = bi != 0; \
                           if (of > max_f || (of == max_f && oj > max_j)) { max_f = of; max_j = oj; } }
                    PF_COMBINE(0xB1) PF_COMBINE(0x4E) PF_COMBINE(0x141)
#undef PF_COMBINE
                }
                if (on && sub == 0) {
                    T.f[k] = max_f; T.p[kt] = (uint16_t)(max_j < 0 ? 0 : (int32_t)k - max_j);
                    if (bad || nv > P.max_skip) { const int32_t d = atomicAdd(&L.n_dirty, 1); if (d < PF_DIRTY_CAP) L.dirty[d] = t0 + kt; }
                }
            }
            __syncthreads();
            if (L.n_dirty > PF_DIRTY_CAP) return false;      // uniform: see par_fill_block
        }
        for (uint32_t k = tid; k < tn; k += nthr) {
            const uint32_t g = t0 + k, d = T.p[k];
            f[g] = T.f[h + k];
            *(int2 *)(pt + 2 * (size_t)g) = make_int2(d ? (int32_t)(g - d) : -1, 0);
        }
        // the halo of the next tile: the last anchors of this one
        const uint32_t tot = h + tn, nh = tot < PFT_H ? tot : PFT_H;
        uint32_t hx = 0, hq = 0; int32_t hf = 0;
        if (tid < nh) { hx = T.x[tot - nh + tid]; hq = T.q[tot - nh + tid]; hf = T.f[tot - nh + tid]; }
        __syncthreads();
        if (tid < nh) { T.x[tid] = hx; T.q[tid] = hq; T.f[tid] = hf; }
        h = nh;
        __syncthreads();
    }
    const int32_t nd = L.n_dirty;
    if (nd > PF_DIRTY_CAP) return false;
    for (int32_t d = (int32_t)tid; d < nd; d += (int32_t)nthr) { uint32_t c = L.dirty[d]; while (!(q[c] >> 31)) --c; pt[2 * (size_t)c + 1] = PF_DIRTY; }
    __syncthreads();
    return true;
}

template <class Store, class EM = NoEmit>
__device__ inline void backtrack_mask(Store &S, int n, const ChainParams &P, int32_t &n_u, int32_t &best, bool first_only, const EM &em = EM())
{
    n_u = 0; best = 0;
    uint64_t done = 0;            // t == 1
    int64_t n_v = 0, bound = INT64_MAX;
    for (;;) {
        int64_t cur = -1;
        for (int i = 0; i < n; ++i) {
            int32_t f = S.F(i);
            if (f < P.min_sc || ((done >> i) & 1)) continue;
            int64_t key = (int64_t)f << 32 | (uint32_t)i;
            if (key < bound && key > cur) cur = key;
        }
        if (cur < 0) break;
        bound = cur;
        const int32_t zf = (int32_t)(cur >> 32);
        const int zi = (int)(cur & 0xffffffff);
        if (em.done(zf)) break;
        // mg_chain_bk_end
        int i = zi, end_i = -1, max_i = zi;
        int32_t max_s = 0;
        do {
            end_i = i = S.Pm(i);
            int32_t sc = i < 0 ? zf : zf - S.F(i);
            if (sc > max_s) { max_s = sc; max_i = i; }
            else if (max_s - sc > P.bw) break;
        } while (i >= 0 && !((done >> i) & 1));
        (void)end_i;
        const int64_t n_v0 = n_v;
        for (i = zi; i != max_i; i = S.Pm(i)) { ++n_v; done |= 1ULL << i; }
        const int32_t sc = i < 0 ? zf : zf - S.F(i);
        if (sc >= P.min_sc && n_v > n_v0 && n_v - n_v0 >= P.min_cnt) { ++n_u; if (sc > best) best = sc; em((int64_t)zi, (int64_t)max_i, sc, n_v - n_v0, zf); }
        else n_v = n_v0;
        if (first_only && n_u > 0) break;
    }
}

// large n: heap sort of (f<<32|index) in caller-provided memory z (n entries), then the same visit order
template <class Store, class Idx, class EM = NoEmit>
__device__ inline void backtrack_heap(Store &S, Idx n, const ChainParams &P, uint64_t *z, int32_t &n_u, int32_t &best, bool first_only = false, const EM &em = EM())
{
    n_u = 0; best = 0;
    Idx nz = 0;
    for (Idx i = 0; i < n; ++i) { int32_t f = S.F(i); if (f >= P.min_sc) z[nz++] = (uint64_t)(uint32_t)f << 32 | (uint32_t)i; }
    if (nz == 0) return;
    S.clearT(n);
    // max-heap; pop order = descending (f, index)
    auto down = [&](Idx i, Idx m) {
        uint64_t tmp = z[i];
        Idx k = i;
        while ((k = (k << 1) + 1) < m) {
            if (k != m - 1 && z[k] < z[k + 1]) ++k;
            if (z[k] < tmp) break;
            z[i] = z[k]; i = k;
        }
        z[i] = tmp;
    };
    for (Idx i = (nz >> 1) - 1; i >= 0; --i) down(i, nz);
    int64_t n_v = 0;
    for (Idx m = nz; m > 0; --m) {
        uint64_t top = z[0];
        if (em.done((int32_t)(top >> 32))) break;
        z[0] = z[m - 1];
        if (m - 1 > 0) down(0, m - 1);
        backtrack_visit<Store, Idx, EM>(S, P, (int32_t)(top >> 32), (Idx)(top & 0xffffffff), n_v, n_u, best, em);
        if (first_only && n_u > 0) break;
    }
}

// A cluster = contiguous slice of a read's sorted anchors on one strand of one contig (so x>>32 is constant).
// x/q are the sorted anchors, f and pt (p,t interleaved) the DP state; LDS or arena pointers alike.
struct SliceStore {
    const uint64_t *x; const uint32_t *q; int32_t *f; int32_t *pt;
    int32_t pbase = 0;      // predecessors as par_fill_block leaves them (indices into the read's array): the slice's first anchor there; 0 = slice-relative
    __device__ inline uint32_t grp(int32_t) const { return 0; }
    __device__ inline uint32_t rlo(int32_t i) const { return (uint32_t)x[i]; }
    __device__ inline uint32_t qp(int32_t i) const { return q[i] & 0x7fffffffu; }     // bit 31 = cluster-start mark
    __device__ inline uint64_t X(int32_t i) const { return x[i]; }
    __device__ inline int32_t F(int32_t i) const { return f[i]; }
    __device__ inline int32_t Pm(int32_t i) const { const int32_t v = pt[2 * i]; return v < 0 ? v : v - pbase; }
    __device__ inline int32_t T(int32_t i) const { return pt[2 * i + 1]; }
    __device__ inline void setFP(int32_t i, int32_t fv, int32_t pv) { f[i] = fv; pt[2 * i] = pv; }
    __device__ inline void setT(int32_t i, int32_t tv) { pt[2 * i + 1] = tv; }
    __device__ inline void clearT(int32_t n) { for (int32_t i = 0; i < n; ++i) pt[2 * i + 1] = 0; }
    __device__ inline void clearAux(int32_t n) { clearT(n); }
};

// emitter over a store that still holds the full x of its anchors.  base = index of the store's anchor 0 in the read's sorted
// anchor array (the discovery key of mg_chain_backtrack is (f, that global index)); on = this lane does the writing.
template <class Store>
struct StoreEmit {
    const ChainSink *sk; Store *S; uint32_t read, base; bool on; int32_t k; uint32_t rhash; int32_t qlen; const TandemQ *tq;
    __device__ inline void operator()(int64_t zi, int64_t end_i, int32_t sc, int64_t cnt, int32_t zf) const
    {
        if (!sk || !on) return;
        Store &St = *S;
        sink_emit(*sk, read, (int32_t)zi, (int32_t)end_i, sc, (uint32_t)cnt, (uint32_t)zf, base + (uint32_t)zi, k, rhash, qlen,
                  [&](int32_t i, uint64_t &x, uint32_t &q) { x = St.X(i); q = St.qp(i); },
                  [&](int32_t i) { return (int32_t)St.Pm(i); }, tq);
    }
    // a chain's score is at most the f of its last anchor: once that falls below the best score handed over, the rest is not wanted
    __device__ inline bool done(int32_t zf) const { return sk && sk->best && zf < sink_best_score(*sk, read); }
};

// ---- SH_F_CIGAR, flag-only: the decision inside a chaining kernel that holds ALL chains of a read ---------------------------------
// regs[0] of mm_gen_regs is the chain with the largest z = (score << 32 | cnt) ^ h (h: hash of the first anchor and the query
// length); it is primary whatever mm_set_parent does to the others, so it is always aligned: if it passes (chain_lemma), the read is
// mapped.  Anything else - a tie in z, a top chain the lemma cannot vouch for - hands all chains to the extension stage.
struct BestChain { unsigned long long z; int32_t zi, end_i, score, cnt; uint32_t base; int32_t tie, n; };

__device__ inline void best_update(BestChain &b, unsigned long long z, int32_t zi, int32_t end_i, int32_t sc, int32_t cnt, uint32_t base)
{
    ++b.n;
    if (b.n == 1 || z > b.z) { b.z = z; b.zi = zi; b.end_i = end_i; b.score = sc; b.cnt = cnt; b.base = base; b.tie = 0; }
    else if (z == b.z) b.tie = 1;
}
// emitter that only remembers the top chain; hi(i) = x >> 32 of anchor i
template <class Store, class HI>
struct BestEmit {
    Store *S; BestChain *b; uint32_t rhash; int32_t k; uint32_t base; HI hi; bool on; TandemQ tq;
    __device__ inline void operator()(int64_t zi, int64_t end_i, int32_t sc, int64_t cnt, int32_t) const
    {
        if (!on) return;
        int32_t first = (int32_t)zi;
        for (int32_t p = (int32_t)S->Pm(first); p != (int32_t)end_i; p = (int32_t)S->Pm(first)) first = p;
        const uint64_t x0 = (uint64_t)hi(first) << 32 | S->rlo(first);
        best_update(*b, chain_z(x0, S->qp(first), k, sc, (uint32_t)cnt, rhash, tandem_yflag(tq, x0, S->qp(first), k)), (int32_t)zi, (int32_t)end_i, sc, (int32_t)cnt, base);
    }
    __device__ inline bool done(int32_t zf) const { return b->n > 0 && zf < b->score; }
};
// mm_max_stretch over the chain zi -> end_i (exclusive), walked backwards, and the test on it.  hi = x >> 32 of the chain's anchors
// (strand | contig).
template <class Store>
__device__ inline int32_t chain_lemma(Store &S, int32_t zi, int32_t end_i, const ChainParams &P, uint32_t hi, MidReq &mid)
{   // 1: the region passes mm_filter_regs; 2: it does if its stretch shows no z-drop on the bases (mid says where: middle_no_zdrop_wave); 0: unknown
    if (!P.ext_lemma) return 0;
    int32_t run_score = P.k, run_unc = 0, run_last = zi, run_first = zi;
    int32_t best_score = -1, best_unc = 0, b_first = zi, b_last = zi;
    for (int32_t i = zi;;) {
        const int32_t p = (int32_t)S.Pm(i);
        if (p == end_i) break;
        const int32_t lr = (int32_t)(S.rlo(i) - S.rlo(p)), lq = (int32_t)S.qp(i) - (int32_t)S.qp(p);
        if (lq == lr) { run_score += lq < P.k ? lq : P.k; run_unc += lq > P.k ? lq - P.k : 0; run_first = p; }
        else {
            if (run_score >= best_score) { best_score = run_score; best_unc = run_unc; b_first = run_first; b_last = run_last; }      // >=: of equal runs the EARLIER one is the stretch
            run_score = P.k; run_unc = 0; run_last = run_first = p;
        }
        i = p;
    }
    if (run_score >= best_score) { best_score = run_score; best_unc = run_unc; b_first = run_first; b_last = run_last; }
    const int32_t qf = (int32_t)S.qp(b_first), ql = (int32_t)S.qp(b_last);
    if (!(best_score >= P.min_sc && ql - qf >= P.k)) return 0;
    if (best_unc <= P.ext_unc_max) return 1;
    mid.rid = (int32_t)(hi & 0x7fffffffu); mid.rev = (int32_t)(hi >> 31); mid.qs = qf + 1 - P.k; mid.qe = ql + 1; mid.rs = (int32_t)S.rlo(b_first) + 1 - P.k;
    return 2;
}

__device__ inline uint32_t prefix_popc64(uint64_t mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// inclusive prefix composition (lane order) of the functions x -> max(x + a, b)
#define SH_COMP_STEP(CTRL, ROWS) { const int32_t pa = dpp_mov<CTRL, ROWS>(0, a), pb = dpp_mov<CTRL, ROWS>(-(1 << 29), b); \
                                   const int32_t nb = pb + a > b ? pb + a : b; a = pa + a; b = nb; }
__device__ inline void wave_scan_compose(int32_t &a, int32_t &b)
{
    SH_COMP_STEP(0x111, 0xf) SH_COMP_STEP(0x112, 0xf) SH_COMP_STEP(0x114, 0xf) SH_COMP_STEP(0x118, 0xf)
    SH_COMP_STEP(0x142, 0xa) SH_COMP_STEP(0x143, 0xc)
}

// ---- wave-cooperative DP: all 64 lanes work on ONE cluster ---------------------------------------------------------
// mg_lchain_dp scans the predecessors j = i-1 .. st of anchor i sequentially, with a running maximum, the
// max_skip counter and the t[] marks.  Here the 64 lanes evaluate 64 predecessors at once and the sequential
// semantics are reproduced exactly with wave scans:
//   * "sc > running max"  = comparison against an exclusive prefix maximum in scan order;
//   * t[j] == i           = "j is the predecessor of an anchor scanned earlier": the marks of a chunk are written
//                           first, then read; a mark written by a lane past the break point can only concern
//                           anchors that are themselves past the break point, so over-marking is harmless;
//   * n_skip (saturating decrement on a new maximum, increment on a marked non-maximum, break above max_skip)
//                         = prefix composition of functions x -> max(x + a, b), which is closed under composition.
// The critical path of a 20-anchor cluster drops from ~190 dependent pair evaluations to 20 wave steps.
__device__ inline void wave_mem_sync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }

__device__ inline bool chain_dp_wave(const SliceStore &S, int n, int32_t qlen, const ChainParams &P, uint32_t lane, int32_t stop_at = INT32_MAX)
{
    int32_t max_dist_y = P.is_sr ? (qlen > P.max_gap ? qlen : P.max_gap) : P.max_gap;
    int32_t max_dist_x;
    if (P.max_gap_ref > 0) max_dist_x = P.max_gap_ref;
    else if (P.max_frag_len > 0) { max_dist_x = P.max_frag_len - qlen; if (max_dist_x < P.max_gap) max_dist_x = P.max_gap; }
    else max_dist_x = P.max_gap;
    if (max_dist_x < P.bw) max_dist_x = P.bw;
    if (max_dist_y < P.bw) max_dist_y = P.bw;
    volatile const uint64_t *x = S.x; volatile const uint32_t *q = S.q; volatile int32_t *f = S.f; volatile int32_t *pt = S.pt;
    const int NEG = -(1 << 28);

    for (int i = (int)lane; i < n; i += 64) pt[2 * i + 1] = 0;
    wave_mem_sync();
    int st = 0, max_ii = -1;
    for (int i = 0; i < n; ++i) {
        const uint32_t li = (uint32_t)x[i], qi = q[i] & 0x7fffffffu;
        while (st < i && (uint64_t)li > (uint64_t)(uint32_t)x[st] + (uint64_t)max_dist_x) ++st;
        if (i - st > P.max_iter) st = i - P.max_iter;
        int32_t max_f = P.k, n_skip = 0;
        int max_j = -1, end_j = st - 1;
        for (int jb = i - 1; jb >= st; jb -= 64) {
            const int j = jb - (int)lane;
            const bool valid = j >= st;
            int32_t sc = SH_SC_NONE, pj = -1;
            if (valid) {
                sc = comput_sc(li, qi, (uint32_t)x[j], q[j] & 0x7fffffffu, max_dist_x, max_dist_y, P);
                if (sc != SH_SC_NONE) { sc += f[j]; pj = pt[2 * j]; }
            }
            const bool has = valid && sc != SH_SC_NONE;
            if (has && pj >= 0) pt[2 * pj + 1] = i;
            wave_mem_sync();
            const bool is_t = has && pt[2 * j + 1] == i;
            // exclusive prefix maximum in scan order (lane 0 = j = jb first), seeded with the running max_f
            const int32_t scv = has ? sc : INT32_MIN;
            const int32_t incl = wave_scan_max_incl(scv);
            int32_t excl = wave_shr1(incl, INT32_MIN);
            if (excl < max_f) excl = max_f;
            const bool new_max = has && sc > excl;
            const bool inc_ev = has && !new_max && is_t;
            // n_skip is a walk reflected at zero (see chain_dp_ring)
            const uint64_t inc_m = __ballot(inc_ev), nm_m = __ballot(new_max);
            const int32_t yl = n_skip + (int32_t)prefix_popc64(inc_m) + (inc_ev ? 1 : 0) - (int32_t)prefix_popc64(nm_m) - (new_max ? 1 : 0);
            const int32_t mn = wave_scan_min_incl(yl);
            const int32_t val = yl - (mn < 0 ? mn : 0);
            const uint64_t brk = __ballot(inc_ev && val > P.max_skip);
            const int L = brk ? __ffsll((unsigned long long)brk) - 1 : 63;
            const int32_t mm = __builtin_amdgcn_readlane(incl, L);
            if (mm > max_f) {
                max_f = mm;
                const uint64_t eq = __ballot((int)lane <= L && scv == mm);
                max_j = jb - (__ffsll((unsigned long long)eq) - 1);
            }
            n_skip = __builtin_amdgcn_readlane(val, L);
            if (brk) { end_j = jb - L; break; }
        }
        // the max_ii shortcut (uniform)
        bool far = true;
        if (max_ii >= 0) far = (uint64_t)(li - (uint32_t)x[max_ii]) > (uint64_t)max_dist_x;
        if (max_ii < 0 || far) {
            int32_t bf = INT32_MIN; int bj = -1;
            for (int jb = i - 1; jb >= st; jb -= 64) {
                const int j = jb - (int)lane;
                int32_t fj = j >= st ? f[j] : INT32_MIN;
                const int32_t cm = wave_all_max(fj);
                if (cm > bf) { bf = cm; bj = jb - (__ffsll((unsigned long long)__ballot(j >= st && fj == cm)) - 1); }
            }
            max_ii = bj;
        }
        if (max_ii >= 0 && max_ii < end_j) {
            int32_t tmp = comput_sc(li, qi, (uint32_t)x[max_ii], q[max_ii] & 0x7fffffffu, max_dist_x, max_dist_y, P);
            if (tmp != SH_SC_NONE && max_f < tmp + f[max_ii]) { max_f = tmp + f[max_ii]; max_j = max_ii; }
        }
        if (lane == 0) { f[i] = max_f; pt[2 * i] = max_j; }
        wave_mem_sync();
        if (max_f >= stop_at) return true;           // wave-uniform
        bool near = false;
        if (max_ii >= 0) near = (uint64_t)(li - (uint32_t)x[max_ii]) <= (uint64_t)max_dist_x;
        if (max_ii < 0 || (near && f[max_ii] < max_f)) max_ii = i;
    }
    return false;
}

// ---- wave-cooperative DP with the recent anchors in LDS -------------------------------------------------------------
// Same exact scheme as chain_dp_wave, built for clusters of thousands of anchors (the true locus of a long read):
// the last RING_WIN anchors' (x, q, f, p, t) live in a per-wave LDS ring, so a step whose scan ends within that
// window - nearly all of them, max_skip ends a scan after a few dozen predecessors - never waits for HBM/L2.  Older
// predecessors are read from the arena arrays (f, p are written through), their t marks live in the arena too: for a
// given i an anchor is either inside the window or not, so its mark for that i has exactly one home.
#define RING_CAP 256
#define RING_WIN 192
#define RING_TBITS 8192         // t marks: one bit per anchor of the look-back window (max_chain_iter <= RING_TMAX_ITER)
#define RING_TMAX_ITER 8000
struct RingMem { uint4 rec[RING_CAP]; uint32_t tb[RING_TBITS / 32]; };       // rec = (x, q, f, p)

__device__ inline int32_t ld_agent(const int32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// `rm` must be a __shared__ object (the accesses below compile to ds_* once inlined; nothing here is volatile: one
// wave's LDS operations execute in program order, and the compiler keeps may-aliasing stores and loads ordered).
// The t[] marks of mg_lchain_dp ("t[j] == i") only live for one i: they are one bit per anchor of the current
// look-back window [st, i), cleared after the step.  f and p are written through to the arena without waiting; a scan
// that leaves the ring window fences once and reads them past the L1 (the next chunk's loads are issued a chunk ahead).
// Requires P.max_iter <= RING_TMAX_ITER.
__device__ inline bool chain_dp_ring(const uint64_t *gx, const uint32_t *gq, int32_t *gf, int32_t *gpt, int n, int32_t qlen,
                                     const ChainParams &P_in, uint32_t lane, RingMem &rm, unsigned long long *dbg_cnt = nullptr, int32_t stop_at = INT32_MAX)
{
    // the parameters by value: through the reference they live in the kernel's private copy of its argument struct, and every use inside the
    // loops below was a scratch load (five per chunk of predecessors in the ISA)
    const ChainParams P = P_in;
    unsigned long long n_ch_in = 0, n_ch_out = 0, n_far = 0; int n_evt = 0, max_win = 0;
    int32_t max_dist_y = P.is_sr ? (qlen > P.max_gap ? qlen : P.max_gap) : P.max_gap;
    int32_t max_dist_x;
    if (P.max_gap_ref > 0) max_dist_x = P.max_gap_ref;
    else if (P.max_frag_len > 0) { max_dist_x = P.max_frag_len - qlen; if (max_dist_x < P.max_gap) max_dist_x = P.max_gap; }
    else max_dist_x = P.max_gap;
    if (max_dist_x < P.bw) max_dist_x = P.bw;
    if (max_dist_y < P.bw) max_dist_y = P.bw;
    constexpr int M = RING_CAP - 1;
    constexpr int TW = RING_TBITS / 32;

    for (int d = (int)lane; d < TW; d += 64) rm.tb[d] = 0;
    int st = 0, sb = 0, max_ii = -1;
    uint32_t sxv = (int)lane < n ? (uint32_t)gx[lane] : 0xffffffffu;      // x of anchor sb + lane (start-of-window search)
    uint32_t mi_x = 0, mi_q = 0; int32_t mi_f = 0;                           // anchor max_ii
    uint32_t nx = 0, nq = 0;                                                  // anchors of the current block of 64
    for (int i = 0; i < n; ++i) {
        if ((i & 63) == 0) {
            const int a = i + (int)lane;
            nx = a < n ? (uint32_t)gx[a] : 0u; nq = a < n ? gq[a] & 0x7fffffffu : 0u;
            // overwrites anchors [i-256, i-193]: outside every window of this block
            rm.rec[a & M].x = nx; rm.rec[a & M].y = nq;
        }
        const uint32_t li = (uint32_t)__builtin_amdgcn_readlane((int)nx, i & 63), qi = (uint32_t)__builtin_amdgcn_readlane((int)nq, i & 63);
        for (;;) {      // while (st < i && x[i] > x[st] + max_dist_x) ++st, 64 candidates at a time
            const int idx = sb + (int)lane;
            const bool ok = idx >= i || (uint64_t)li <= (uint64_t)sxv + (uint64_t)max_dist_x;
            const uint64_t m = __ballot(idx >= st && ok);
            if (m) { st = sb + __ffsll((unsigned long long)m) - 1; break; }
            sb += 64;
            const int a = sb + (int)lane;
            sxv = a < n ? (uint32_t)gx[a] : 0xffffffffu;
        }
        if (i - st > P.max_iter) st = i - P.max_iter;
        int32_t max_f = P.k, n_skip = 0;
        int max_j = -1, end_j = st - 1;
        bool synced = false, marked = false;
        int c = 0;
        uint32_t px = 0, pq = 0; int32_t pf = 0, pp = -1;       // the next chunk beyond the ring window, loaded a chunk ahead
        for (int jb = i - 1; jb >= st; jb -= 64, ++c) {
            const int j = jb - (int)lane;
            const bool valid = j >= st;
            const bool inw = c < RING_WIN / 64;             // uniform: the whole chunk is inside the ring window
            if (inw) ++n_ch_in; else ++n_ch_out;
            uint32_t xj, qj; int32_t fj, pj;
            if (inw) {
                const uint4 rc = rm.rec[j & M];
                xj = rc.x; qj = rc.y; fj = (int32_t)rc.z; pj = (int32_t)rc.w;
            } else { xj = px; qj = pq; fj = pf; pj = pp; }
            if (c + 1 >= RING_WIN / 64 && jb - 64 >= st) {      // issue the loads of the next chunk
                if (!synced) { wave_mem_sync(); synced = true; }
                const int jn = j - 64;
                if (jn >= st) { px = (uint32_t)gx[jn]; pq = gq[jn] & 0x7fffffffu; pf = ld_agent(gf + jn); pp = ld_agent(gpt + 2 * jn); }
            }
            int32_t sc = comput_sc(li, qi, xj, qj, max_dist_x, max_dist_y, P);
            const bool has = valid && sc != SH_SC_NONE;
            sc = has ? sc + fj : INT32_MIN;
            const bool mk = has && pj >= st;               // marks below st are never read in this step
            if (mk) atomicOr(&rm.tb[(pj & (RING_TBITS - 1)) >> 5], 1u << (pj & 31));
            marked |= __ballot(mk) != 0;
            const bool is_t = has && ((rm.tb[(j & (RING_TBITS - 1)) >> 5] >> (j & 31)) & 1u);
            // exclusive prefix maximum in scan order (lane 0 = j = jb first), seeded with the running max_f
            const int32_t incl = wave_scan_max_incl(sc);
            int32_t excl = wave_shr1(incl, INT32_MIN);
            if (excl < max_f) excl = max_f;
            const bool new_max = has && sc > excl;
            const bool inc_ev = has && !new_max && is_t;
            // n_skip is a walk reflected at zero (new maximum: max(x - 1, 0); marked non-maximum: x + 1):
            // x_l = Y_l - min(0, min_{m <= l} Y_m) with the free walk Y_l = n_skip + #inc(<= l) - #new_max(<= l)
            const uint64_t inc_m = __ballot(inc_ev), nm_m = __ballot(new_max);
            const int32_t yl = n_skip + (int32_t)prefix_popc64(inc_m) + (inc_ev ? 1 : 0) - (int32_t)prefix_popc64(nm_m) - (new_max ? 1 : 0);
            const int32_t mn = wave_scan_min_incl(yl);
            const int32_t val = yl - (mn < 0 ? mn : 0);
            const uint64_t brk = __ballot(inc_ev && val > P.max_skip);
            const int L = brk ? __ffsll((unsigned long long)brk) - 1 : 63;
            const int32_t mm = __builtin_amdgcn_readlane(incl, L);      // max over the lanes up to the break point
            if (mm > max_f) {
                max_f = mm;
                const uint64_t eq = __ballot((int)lane <= L && sc == mm);
                max_j = jb - (__ffsll((unsigned long long)eq) - 1);
            }
            n_skip = __builtin_amdgcn_readlane(val, L);
            if (brk) { end_j = jb - L; ++n_evt; break; }
        }
        if (i - st > max_win) max_win = i - st;
        if (marked) {       // clear the marks of this step: they all lie in [st, i)
            const int d1 = (i - 1) >> 5;
            for (int d = (st >> 5) + (int)lane; d <= d1; d += 64) rm.tb[d & (TW - 1)] = 0;
        }
        // the max_ii shortcut (uniform)
        bool far = true;
        if (max_ii >= 0) far = (uint64_t)(li - mi_x) > (uint64_t)max_dist_x;
        if (max_ii < 0 || far) {
            int32_t bf = INT32_MIN; int bj = -1;
            int c2 = 0; ++n_far;
            for (int jb = i - 1; jb >= st; jb -= 64, ++c2) {
                const int j = jb - (int)lane;
                int32_t fj = INT32_MIN;
                if (c2 < RING_WIN / 64) { if (j >= st) fj = (int32_t)rm.rec[j & M].z; }
                else {
                    if (!synced) { wave_mem_sync(); synced = true; }
                    if (j >= st) fj = ld_agent(gf + j);
                }
                const int32_t cm = wave_all_max(fj);
                if (cm > bf) { bf = cm; bj = jb - (__ffsll((unsigned long long)__ballot(j >= st && fj == cm)) - 1); }
            }
            max_ii = bj;
            if (bj >= 0) {
                if (bj >= i - RING_WIN) { mi_x = rm.rec[bj & M].x; mi_q = rm.rec[bj & M].y; }
                else { mi_x = (uint32_t)gx[bj]; mi_q = gq[bj] & 0x7fffffffu; }
                mi_f = bf;
            }
        }
        if (max_ii >= 0 && max_ii < end_j) {
            int32_t tmp = comput_sc(li, qi, mi_x, mi_q, max_dist_x, max_dist_y, P);
            if (tmp != SH_SC_NONE && max_f < tmp + mi_f) { max_f = tmp + mi_f; max_j = max_ii; }
        }
        if (lane == 0) { rm.rec[i & M].z = (uint32_t)max_f; rm.rec[i & M].w = (uint32_t)max_j; gf[i] = max_f; gpt[2 * i] = max_j; }
        if (max_f >= stop_at) { wave_mem_sync(); return true; }      // wave-uniform
        bool near = false;
        if (max_ii >= 0) near = (uint64_t)(li - mi_x) <= (uint64_t)max_dist_x;
        if (max_ii < 0 || (near && mi_f < max_f)) { max_ii = i; mi_x = li; mi_q = qi; mi_f = max_f; }
    }
    wave_mem_sync();
    if (dbg_cnt && lane == 0) {
        atomicAdd(&dbg_cnt[0], n_ch_in); atomicAdd(&dbg_cnt[1], n_ch_out); atomicAdd(&dbg_cnt[2], n_far);
        atomicAdd(&dbg_cnt[3], 1ull); atomicAdd(&dbg_cnt[6], (unsigned long long)n);
        if (n_evt) { atomicAdd(&dbg_cnt[4], 1ull); atomicAdd(&dbg_cnt[5], (unsigned long long)n); }
        atomicMax(&dbg_cnt[7], (unsigned long long)max_win);
        if (max_win > 64) atomicAdd(&dbg_cnt[8], (unsigned long long)n);
        if (max_win > 128) atomicAdd(&dbg_cnt[9], (unsigned long long)n);
    }
    return false;
}

// Flag-only shortcut of mg_chain_backtrack: the first candidate popped is the maximum (f, index) with f >= min_sc.  Its
// chain is accepted as soon as the best score seen on the way back reaches min_sc over >= min_cnt anchors (both only
// grow along the walk).  Returns 1 accepted, 0 no candidate at all, -1 undecided (the caller runs the full procedure).
__device__ inline int first_chain_quick(const int32_t *gf, const int32_t *gpt, int n, const ChainParams &P, uint32_t lane)
{
    volatile const int32_t *vf = gf, *vpt = gpt;
    long long key = -1;
    for (int i = (int)lane; i < n; i += 64) {
        const int32_t f = vf[i];
        if (f >= P.min_sc) { const long long kk = (long long)f << 32 | (uint32_t)i; key = kk > key ? kk : key; }
    }
    key = wave_all_max_i64(key);
    if (key < 0) return 0;
    const int32_t zf = (int32_t)(key >> 32);
    int i = (int)(key & 0xffffffff), steps = 0, cnt = 0;
    int32_t max_s = 0;
    do {
        i = vpt[2 * i]; ++steps;
        const int32_t s = i < 0 ? zf : zf - vf[i];
        if (s > max_s) { max_s = s; cnt = steps; }
        else if (max_s - s > P.bw) break;
        if (max_s >= P.min_sc && cnt >= P.min_cnt) return 1;
    } while (i >= 0);
    return (max_s >= P.min_sc && cnt >= 1 && cnt >= P.min_cnt) ? 1 : -1;
}


// Hand-over backtrack of one cluster when only candidates for regs[0] are wanted (ChainSink::best): mg_chain_backtrack visits the
// candidates in descending (f, index) order, and the emitter's done(zf) ends the visit once f falls below the best score handed over -
// after one or two chains, typically.  A heap over thousands of candidates in HBM costs far more than that: the next candidate is
// found by a wave-wide argmax over the cluster's f instead (n / 64 coalesced loads per chain).  Wave-uniform; stores by the emitter.
template <class EM>
__device__ inline void backtrack_wave_top(SliceStore &S, int32_t n, const ChainParams &P, int32_t &n_u, int32_t &best, const EM &em, uint32_t lane)
{
    n_u = 0; best = 0;
    for (int32_t i = (int32_t)lane; i < n; i += 64) S.setT(i, 0);
    wave_mem_sync();
    volatile const int32_t *vf = S.f, *vpt = S.pt;
    long long bound = (1ll << 62);
    int64_t n_v = 0;
    for (;;) {
        long long key = -1;
        for (int32_t i = (int32_t)lane; i < n; i += 64) {
            const int32_t f = vf[i];
            if (f < P.min_sc || vpt[2 * i + 1] != 0) continue;
            const long long kk = (long long)f << 32 | (uint32_t)i;
            if (kk < bound && kk > key) key = kk;
        }
        key = wave_all_max_i64(key);
        if (key < 0) break;
        bound = key;
        const int32_t zf = (int32_t)(key >> 32), zi = (int32_t)(key & 0xffffffff);
        if (em.done(zf)) break;
        backtrack_visit<SliceStore, int32_t, EM>(S, P, zf, zi, n_v, n_u, best, em);       // uniform: every lane walks, lane 0's emitter writes
        wave_mem_sync();
    }
}

// mg_chain_backtrack over ALL anchors of a read whose DP par_fill_block has done, for hand-overs that only want candidates for regs[0]
// (ChainSink::best): mg_chain_backtrack knows no clusters - it visits the read's candidates in descending (f, index) order - and the
// emitter's done(zf) ends the visit once f falls below the best score handed over.  The maximum (f, index) comes from a block-wide argmax;
// its chain's score S is found by a dry walk (mg_chain_bk_end leaves no marks); ONE pass then lists every candidate with f >= S - the only
// ones that can still matter, the top one included - and wave 0 visits them in order.  That settles a typical read with three sweeps over f
// and a handful of walks, without looking at a single cluster.  Reads with many candidates at the top score (tandem arrays: every copy
// chains alike) are better served cluster by cluster, in parallel: false is returned before anything is marked or emitted, and the caller
// carries on as if this had not run.  S spans the read (p = indices into it); t must be 0 everywhere.  Results in thread 0.
// s_red: 18 long longs of LDS; cand_f / cand_i / cand_n: an LDS list of `cap` entries.
#define TOPBT_MAX 64
template <class EM>
__device__ inline bool backtrack_block_top(SliceStore &S, int32_t n, const ChainParams &P, int32_t &n_u, int32_t &best, const EM &em, const ChainSink &sk, uint32_t read,
                                           uint32_t tid, uint32_t nthr, long long *s_red, uint32_t *cand_f, uint32_t *cand_i, int32_t *cand_n, uint32_t cap, unsigned long long *dbg = nullptr)
{   // cap <= TOPBT_MAX <= 64: the list is one entry per lane of wave 0
    n_u = 0; best = 0;
    const uint32_t lane = tid & 63, wave = tid >> 6, n_wave = nthr >> 6;
    long long key = -1;
    for (int32_t i = (int32_t)tid; i < n; i += (int32_t)nthr) {
        const int32_t f = S.f[i];
        if (f < P.min_sc) continue;
        const long long kk = (long long)f << 32 | (uint32_t)i;
        if (kk > key) key = kk;
    }
    key = wave_all_max_i64(key);
    if (lane == 0) s_red[wave] = key;
    if (tid == 0) *cand_n = 0;
    __syncthreads();
    key = s_red[0];
    for (uint32_t w = 1; w < n_wave; ++w) key = s_red[w] > key ? s_red[w] : key;
    if (key < 0) return true;                 // no candidate at all
    const int32_t zf = (int32_t)(key >> 32), zi = (int32_t)(key & 0xffffffff);
    if (em.done(zf)) return true;
    if (wave == 0) {      // the top chain's score, nothing marked
        const int32_t end_i = chain_bk_end<SliceStore, int32_t>(S, P.bw, zf, zi);
        int32_t cnt = 0;
        for (int32_t i = zi; i != end_i; i = S.Pm(i)) ++cnt;
        const int32_t sc = end_i < 0 ? zf : zf - S.F(end_i);
        if (lane == 0) s_red[16] = sc >= P.min_sc && cnt >= 1 && cnt >= P.min_cnt ? sc : -1;
    }
    __syncthreads();
    const int32_t s1 = (int32_t)s_red[16];
    if (s1 < 0) return false;
    int32_t thr = sink_best_score(sk, read);
    if (thr < s1) thr = s1;
    for (int32_t i = (int32_t)tid; i < n; i += (int32_t)nthr) {
        const int32_t f = S.f[i];
        if (f < thr) continue;
        const int32_t slot = atomicAdd(cand_n, 1);
        if ((uint32_t)slot < cap) { cand_f[slot] = (uint32_t)f; cand_i[slot] = (uint32_t)i; }
    }
    __syncthreads();
    const int32_t nc = *cand_n;
    if (dbg && tid == 0) { atomicAdd(&dbg[8], 1ull); atomicAdd(&dbg[11], (unsigned long long)nc); if ((uint32_t)nc > cap) atomicAdd(&dbg[12], 1ull); }
    if ((uint32_t)nc > cap) return false;
    if (wave == 0) {
        long long bound = 1ll << 62;
        int64_t n_v = 0;
        unsigned long long d_visit = 0;
        for (;;) {
            long long k2 = -1;
            if ((int32_t)lane < nc) { const long long kk = (long long)cand_f[lane] << 32 | cand_i[lane]; if (kk < bound) k2 = kk; }
            k2 = wave_all_max_i64(k2);
            if (k2 < 0) break;
            bound = k2;
            const int32_t f2 = (int32_t)(k2 >> 32), i2 = (int32_t)(k2 & 0xffffffff);
            if (em.done(f2)) break;
            ++d_visit;
            backtrack_visit<SliceStore, int32_t, EM>(S, P, f2, i2, n_v, n_u, best, em);
        }
        if (dbg && lane == 0) atomicAdd(&dbg[10], d_visit);
    }
    __syncthreads();
    return true;
}

// one big cluster, one wave, DP state through the LDS ring
__device__ inline void chain_cluster_ring(const uint64_t *gx, uint32_t *gq, int32_t *gf, int32_t *gpt, int32_t n, int32_t qlen, const ChainParams &P,
                                          int32_t &n_u, int32_t &best, bool first_only, uint32_t lane, RingMem &rm, unsigned long long *dbg_cnt = nullptr,
                                          const ChainSink *sk = nullptr, uint32_t read = 0, uint32_t base = 0, uint64_t *heap = nullptr, bool pre = false)
{   // pre: f and p are there already (par_fill_block; p as indices into the read's array, this cluster starting at `base`)
    n_u = 0; best = 0;
    if (sk) {      // hand-over mode: every chain, anchors intact (the heap lives in `heap`, not over the x slice)
        if (!pre) chain_dp_ring(gx, gq, gf, gpt, n, qlen, P, lane, rm, dbg_cnt);
        SliceStore S{gx, gq, gf, gpt, pre ? (int32_t)base : 0};
        const StoreEmit<SliceStore> em{sk, &S, read, base, lane == 0, P.k, region_hash(qlen), qlen, nullptr};
        if (sk->best) backtrack_wave_top(S, n, P, n_u, best, em, lane);
        else backtrack_heap<SliceStore, int32_t, StoreEmit<SliceStore>>(S, n, P, heap, n_u, best, false, em);
        wave_mem_sync();
        return;
    }
    if (first_only && P.flag_stop != INT32_MAX && !pre) { n_u = chain_dp_ring(gx, gq, gf, gpt, n, qlen, P, lane, rm, dbg_cnt, P.flag_stop) ? 1 : 0; return; }
    if (!pre) chain_dp_ring(gx, gq, gf, gpt, n, qlen, P, lane, rm, dbg_cnt);
    if (first_only && !pre) {
        const int rc = first_chain_quick(gf, gpt, n, P, lane);
        if (rc >= 0) { n_u = rc; return; }
    }
    SliceStore S{gx, gq, gf, gpt, pre ? (int32_t)base : 0};
    backtrack_heap<SliceStore, int32_t>(S, n, P, (uint64_t *)gx, n_u, best, first_only);
    wave_mem_sync();
}

// one cluster, the whole wave: DP in parallel, backtrack executed uniformly by every lane
__device__ inline void chain_cluster_wave(SliceStore &S, int32_t n, int32_t qlen, const ChainParams &P, uint64_t *zbuf, int32_t &n_u, int32_t &best,
                                          bool first_only, uint32_t lane, const ChainSink *sk = nullptr, uint32_t read = 0, uint32_t base = 0, bool pre = false)
{   // pre: f and p are there already (par_fill_block; S.pbase set by the caller)
    if (sk) {      // hand-over mode (zbuf must not alias the anchors)
        const StoreEmit<SliceStore> em{sk, &S, read, base, lane == 0, P.k, region_hash(qlen), qlen, nullptr};
        if (P.ext_s1 && n >= 2 && n <= 64 && !pre) {
            // A cluster whose anchors all lie on ONE diagonal, d <= min(max_dist_x, max_dist_y) apart (a copy of the read up to substitutions -
            // the true locus nearly always): the outcome of mg_lchain_dp + mg_chain_backtrack is known without running them (the argument of
            // k_pair_pass mode 2; no skip penalty: ext_s1) - every anchor links to its predecessor (it is scanned first and no earlier one can
            // beat f[i-1] + min(k, d)), f is the running sum, and the backtrack returns the one chain of all n anchors if it clears min_sc /
            // min_cnt.  f and p are written as the DP would have, then the chain is handed over.
            int32_t mdy = P.is_sr ? (qlen > P.max_gap ? qlen : P.max_gap) : P.max_gap, mdx;
            if (P.max_gap_ref > 0) mdx = P.max_gap_ref;
            else if (P.max_frag_len > 0) { mdx = P.max_frag_len - qlen; if (mdx < P.max_gap) mdx = P.max_gap; }
            else mdx = P.max_gap;
            if (mdx < P.bw) mdx = P.bw;
            if (mdy < P.bw) mdy = P.bw;
            const int32_t dmax = mdx < mdy ? mdx : mdy;
            const bool in = (int32_t)lane < n;
            const uint32_t lo = in ? S.rlo((int32_t)lane) : 0u, qv = in ? S.qp((int32_t)lane) : 0u;
            const uint32_t dg = lo - qv, dg0 = (uint32_t)__builtin_amdgcn_readlane((int)dg, 0);
            const int32_t dq = (int32_t)qv - wave_shr1((int32_t)qv, 0);
            const bool good = !in || (dg == dg0 && (lane == 0 || (dq > 0 && dq <= dmax)));
            if (__ballot(!good) == 0) {
                int32_t v = !in ? 0 : (lane == 0 ? P.k : (dq < P.k ? dq : P.k));
                const int32_t fl = wave_scan_add_incl(v);
                if (in) { S.setFP((int32_t)lane, fl, (int32_t)lane - 1); S.setT((int32_t)lane, 0); }
                wave_mem_sync();
                const int32_t sc = wave_bcast(fl, n - 1);
                n_u = 0; best = 0;
                if (sc >= P.min_sc && n >= P.min_cnt) { n_u = 1; best = sc; em((int64_t)(n - 1), (int64_t)-1, sc, (int64_t)n, sc); }
                wave_mem_sync();
                return;
            }
        }
        if (!pre) chain_dp_wave(S, n, qlen, P, lane);
        if (n <= 64) backtrack_mask(S, n, P, n_u, best, false, em);
        else if (sk->best) backtrack_wave_top(S, n, P, n_u, best, em, lane);
        else { backtrack_heap<SliceStore, int32_t, StoreEmit<SliceStore>>(S, n, P, zbuf, n_u, best, false, em); wave_mem_sync(); }
        return;
    }
    if (first_only && P.flag_stop != INT32_MAX && !pre) { n_u = chain_dp_wave(S, n, qlen, P, lane, P.flag_stop) ? 1 : 0; best = 0; return; }
    if (!pre) chain_dp_wave(S, n, qlen, P, lane);
    if (n <= 64) backtrack_mask(S, n, P, n_u, best, first_only);
    else { backtrack_heap<SliceStore, int32_t>(S, n, P, zbuf, n_u, best, first_only); wave_mem_sync(); }
}

// DP + backtrack of one cluster.  zbuf: n 8-B words for the heap when n > 32 (may alias the x slice: the
// anchors are dead once the DP is done).
__device__ inline void chain_cluster(SliceStore &S, int32_t n, int32_t qlen, const ChainParams &P, uint64_t *zbuf, int32_t &n_u, int32_t &best,
                                     bool first_only, const ChainSink *sk = nullptr, uint32_t read = 0, uint32_t base = 0, bool pre = false)
{   // pre: f and p are there already (par_fill_block; S.pbase set by the caller)
    if (sk) {      // hand-over mode (zbuf must not alias the anchors)
        const StoreEmit<SliceStore> em{sk, &S, read, base, true, P.k, region_hash(qlen), qlen, nullptr};
        if (n <= 64) { if (!pre) chain_dp_mask(S, n, qlen, P); backtrack_mask(S, n, P, n_u, best, false, em); }
        else { if (!pre) chain_dp<SliceStore, int32_t>(S, n, qlen, P); backtrack_heap<SliceStore, int32_t, StoreEmit<SliceStore>>(S, n, P, zbuf, n_u, best, false, em); }
        return;
    }
    if (first_only && P.flag_stop != INT32_MAX && !pre) {
        n_u = (n <= 64 ? chain_dp_mask(S, n, qlen, P, P.flag_stop) : chain_dp<SliceStore, int32_t>(S, n, qlen, P, P.flag_stop)) ? 1 : 0;
        best = 0;
        return;
    }
    if (n <= 64) {
        if (!pre) chain_dp_mask(S, n, qlen, P);
        backtrack_mask(S, n, P, n_u, best, first_only);
    } else {
        if (!pre) chain_dp<SliceStore, int32_t>(S, n, qlen, P);
        backtrack_heap<SliceStore, int32_t>(S, n, P, zbuf, n_u, best, first_only);
    }
}
