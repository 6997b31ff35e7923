/*
 * mm_align.c — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See mm_align.h.
 *
 * *** PARITY UNPINNED ***: restates the part of lh3/minimap2 (~v2.28; reached by the reference through crate
 * minimap2 ^0.1.20, /root/reference/Cargo.toml:41) that `.with_cigar()` at /root/reference/src/cleaner.rs:473 switches
 * on and that turns "chains" into `mappings` (:552-556): SURVEY.md App. A.6.  Stages, in upstream's order:
 *     mm_gen_regs -> mm_set_parent -> mm_select_sub              (hit.c; chain_post in map.c)
 *     mm_align_skeleton: per region mm_align1                     (align.c)
 *         short-read mode: mm_max_stretch, ungapped middle, ksw_extd2 end extensions, mm_test_zdrop + second pass,
 *         mm_split_reg on a z-drop, mm_update_extra (mm_fix_cigar, mlen / blen / dp_max)
 *     mm_filter_regs: cnt < min_cnt | mlen < min_chain_score | dp_max < min_dp_max | both clips too long
 * ksw_extd2_sse is restated as a literal scalar emulation of its 16-lane int8 difference recurrence, buffer layout
 * included, because its band edges read cells that the rounded vector ranges computed outside the band.
 * Long-read presets (no MM_F_SR: map-ont, lr:hq, map-hifi; cleaner.rs:457-458,465) take the other branch of mm_align1 -
 * mm_fix_bad_ends, mm_filter_bad_seeds(_alt), left extension, ksw_extd2 between anchor mid-points every >= min_ksw_len
 * bases (first pass with the approximate maximum, second pass when mm_test_zdrop objects, inversion test through a
 * local alignment = ksw_ll_i16), right extension, mm_update_extra with the logarithmic gap cost - preceded by mm_est_err /
 * mm_filter_strand_retained and followed by mm_align1_inv.  The RMQ re-chain that map.c runs before all this is in
 * mm_rmq.c.  mm_update_dp_max and mm_set_mapq run AFTER mm_filter_regs and cannot change the number of mappings; they
 * are not restated.
 */
#include "mm_align.h"
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include <math.h>

#define KSW_NEG_INF (-0x40000000)
#define CIG_MATCH 0
#define CIG_INS 1
#define CIG_DEL 2

/* ------------------------------------------------------------------------------------------------
 * ksw2.h helpers
 * ---------------------------------------------------------------------------------------------- */
void mma_gen_simple_mat(int m, int8_t *mat, int8_t a, int8_t b, int8_t sc_ambi)
{
    int i, j;
    a = a < 0 ? -a : a;
    b = b > 0 ? -b : b;
    sc_ambi = sc_ambi > 0 ? -sc_ambi : sc_ambi;
    for (i = 0; i < m - 1; ++i) {
        for (j = 0; j < m - 1; ++j) mat[i * m + j] = i == j ? a : b;
        mat[i * m + m - 1] = sc_ambi;
    }
    for (j = 0; j < m; ++j) mat[(m - 1) * m + j] = sc_ambi;
}

static void ez_reset(mma_ez *ez)
{
    ez->max_q = ez->max_t = ez->mqe_t = ez->mte_q = -1;
    ez->max = 0; ez->score = ez->mqe = ez->mte = KSW_NEG_INF;
    ez->n_cigar = 0; ez->zdropped = 0; ez->reach_end = 0;
}

static void push_cigar(int *n_cigar, int *m_cigar, uint32_t **cigar, uint32_t op, int len)
{
    if (*n_cigar == 0 || op != ((*cigar)[*n_cigar - 1] & 0xf)) {
        if (*n_cigar == *m_cigar) {
            *m_cigar = *m_cigar ? *m_cigar << 1 : 4;
            *cigar = (uint32_t *)realloc(*cigar, (size_t)*m_cigar * 4);
        }
        (*cigar)[(*n_cigar)++] = (uint32_t)len << 4 | op;
    } else (*cigar)[*n_cigar - 1] += (uint32_t)len << 4;
}

static int apply_zdrop(mma_ez *ez, int32_t H, int r, int t, int zdrop, int8_t e)
{   /* ksw_apply_zdrop, rotated coordinates */
    if (H > (int32_t)ez->max) {
        ez->max = (uint32_t)H; ez->max_t = t; ez->max_q = r - t;
    } else if (t >= ez->max_t && r - t >= ez->max_q) {
        int tl = t - ez->max_t, ql = (r - t) - ez->max_q, l;
        l = tl > ql ? tl - ql : ql - tl;
        if (zdrop >= 0 && (int32_t)ez->max - H > zdrop + l * e) { ez->zdropped = 1; return 1; }
    }
    return 0;
}

/* ksw_backtrack, rotated matrix */
static void backtrack(int is_rev, const uint8_t *p, const int *off, const int *off_end, int n_col, int i0, int j0,
                      int *m_cigar, int *n_cigar_, uint32_t **cigar_)
{
    int n_cigar = 0, i = i0, j = j0, r, state = 0;
    uint32_t tmp;
    while (i >= 0 && j >= 0) {
        int force_state = -1;
        r = i + j;
        if (i < off[r]) force_state = 2;
        if (off_end && i > off_end[r]) force_state = 1;
        tmp = force_state < 0 ? p[(size_t)r * n_col + i - off[r]] : 0;
        if (state == 0) state = tmp & 7;
        else if (!(tmp >> (state + 2) & 1)) state = 0;
        if (state == 0) state = tmp & 7;
        if (force_state >= 0) state = force_state;
        if (state == 0) { push_cigar(&n_cigar, m_cigar, cigar_, CIG_MATCH, 1); --i; --j; }
        else if (state == 1 || state == 3) { push_cigar(&n_cigar, m_cigar, cigar_, CIG_DEL, 1); --i; }
        else { push_cigar(&n_cigar, m_cigar, cigar_, CIG_INS, 1); --j; }
    }
    if (i >= 0) push_cigar(&n_cigar, m_cigar, cigar_, CIG_DEL, i + 1);
    if (j >= 0) push_cigar(&n_cigar, m_cigar, cigar_, CIG_INS, j + 1);
    if (!is_rev)
        for (i = 0; i < n_cigar >> 1; ++i) { tmp = (*cigar_)[i]; (*cigar_)[i] = (*cigar_)[n_cigar - 1 - i]; (*cigar_)[n_cigar - 1 - i] = tmp; }
    *n_cigar_ = n_cigar;
}

/* ------------------------------------------------------------------------------------------------
 * ksw_extd2_sse: dual affine gap, int8 differences along anti-diagonals r = i + j (t = target index)
 * ---------------------------------------------------------------------------------------------- */
static inline int8_t s8(int v) { return (int8_t)v; }      /* _mm_*_epi8 arithmetic wraps */

void mma_ksw_extd2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                   int8_t q, int8_t e, int8_t q2, int8_t e2, int w, int zdrop, int end_bonus, int flag, mma_ez *ez)
{
    int r, t, qe, qe2, n_col_, n_col, *off = 0, *off_end = 0, tlen_, qlen_, last_st, last_en, max_sc, min_sc, long_thres, long_diff;
    const int with_cigar = !(flag & MMA_EZ_SCORE_ONLY), approx_max = !!(flag & MMA_EZ_APPROX_MAX);
    int32_t *H = 0, H0 = 0, last_H0_t = 0;
    uint8_t *mem, *p = 0;
    int8_t *u, *v, *x, *y, *x2, *y2, *s, sc_mch, sc_mis, sc_N;
    uint8_t *sf, *qr;
    int8_t *ox, *ov, *ox2;       /* the previous anti-diagonal's x, v, x2 of the range being rewritten */
    size_t cap;

    ez_reset(ez);
    if (m <= 1 || qlen <= 0 || tlen <= 0) return;
    if (q2 + e2 < q + e) { t = q; q = q2; q2 = (int8_t)t; t = e; e = e2; e2 = (int8_t)t; }
    qe = q + e; qe2 = q2 + e2;
    if (w < 0) w = tlen > qlen ? tlen : qlen;
    tlen_ = (tlen + 15) / 16;
    n_col_ = qlen < tlen ? qlen : tlen;
    n_col_ = ((n_col_ < w + 1 ? n_col_ : w + 1) + 15) / 16 + 1;
    n_col = n_col_ * 16;
    qlen_ = (qlen + 15) / 16;
    for (t = 1, max_sc = mat[0], min_sc = mat[1]; t < m * m; ++t) {
        max_sc = max_sc > mat[t] ? max_sc : mat[t];
        min_sc = min_sc < mat[t] ? min_sc : mat[t];
    }
    if (-min_sc > 2 * (q + e)) return;

    long_thres = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;
    if (q2 + e2 + long_thres * e2 > q + e + long_thres * e) ++long_thres;
    long_diff = long_thres * (e - e2) - (q2 - q) - e2;

    /* one zero-filled block laid out as upstream: u v x y x2 y2 s | sf (target) | qr (reversed query) | 16 spare bytes.
     * The buffers are per-thread and grow-only: a CPU baseline that spent its time in calloc would flatter the GPU. */
    cap = ((size_t)tlen_ * 8 + qlen_ + 1) * 16;
    {
        static __thread uint8_t *t_mem, *t_p; static __thread int8_t *t_ox; static __thread int32_t *t_H; static __thread int *t_off;
        static __thread size_t c_mem, c_p, c_ox, c_H, c_off;
        const size_t n_p = ((size_t)(qlen + tlen - 1) * n_col_ + 1) * 16, n_off = (size_t)(qlen + tlen - 1) * 2;
        if (cap + 64 > c_mem) { c_mem = (cap + 64) * 2; t_mem = (uint8_t *)realloc(t_mem, c_mem); }
        if ((size_t)tlen_ * 16 * 3 > c_ox) { c_ox = (size_t)tlen_ * 16 * 6; t_ox = (int8_t *)realloc(t_ox, c_ox); }
        if ((size_t)tlen_ * 16 > c_H) { c_H = (size_t)tlen_ * 32; t_H = (int32_t *)realloc(t_H, c_H * 4); }
        if (with_cigar && n_p > c_p) { c_p = n_p * 2; t_p = (uint8_t *)realloc(t_p, c_p); }
        if (with_cigar && n_off > c_off) { c_off = n_off * 2; t_off = (int *)realloc(t_off, c_off * sizeof(int)); }
        mem = t_mem; memset(mem, 0, cap + 64);
        ox = t_ox; H = approx_max ? 0 : t_H; p = with_cigar ? t_p : 0; off = with_cigar ? t_off : 0;
    }
    u = (int8_t *)mem; v = u + tlen_ * 16; x = v + tlen_ * 16; y = x + tlen_ * 16; x2 = y + tlen_ * 16; y2 = x2 + tlen_ * 16;
    s = y2 + tlen_ * 16; sf = (uint8_t *)(s + tlen_ * 16); qr = sf + tlen_ * 16;
    memset(u, -q - e, (size_t)tlen_ * 16); memset(v, -q - e, (size_t)tlen_ * 16);
    memset(x, -q - e, (size_t)tlen_ * 16); memset(y, -q - e, (size_t)tlen_ * 16);
    memset(x2, -q2 - e2, (size_t)tlen_ * 16); memset(y2, -q2 - e2, (size_t)tlen_ * 16);
    ov = ox + tlen_ * 16; ox2 = ov + tlen_ * 16;
    if (!approx_max) for (t = 0; t < tlen_ * 16; ++t) H[t] = KSW_NEG_INF;
    if (with_cigar) off_end = off + qlen + tlen - 1;      /* every direction byte the backtrack reads was written by this call */
    for (t = 0; t < qlen; ++t) qr[t] = query[qlen - 1 - t];
    memcpy(sf, target, (size_t)tlen);
    sc_mch = mat[0]; sc_mis = mat[1]; sc_N = mat[m * m - 1] == 0 ? (int8_t)-e2 : mat[m * m - 1];

    for (r = 0, last_st = last_en = -1; r < qlen + tlen - 1; ++r) {
        int st = 0, en = tlen - 1, st0, en0;
        int8_t x1, x21, v1;
        const uint8_t *qrr = qr + (qlen - 1 - r);
        if (st < r - qlen + 1) st = r - qlen + 1;
        if (en > r) en = r;
        if (st < (r - w + 1) >> 1) st = (r - w + 1) >> 1;
        if (en > (r + w) >> 1) en = (r + w) >> 1;
        if (st > en) { ez->zdropped = 1; break; }
        st0 = st; en0 = en;
        st = st / 16 * 16; en = (en + 16) / 16 * 16 - 1;
        if (st > 0) {
            if (st - 1 >= last_st && st - 1 <= last_en) { x1 = x[st - 1]; x21 = x2[st - 1]; v1 = v[st - 1]; }
            else { x1 = s8(-q - e); x21 = s8(-q2 - e2); v1 = s8(-q - e); }
        } else {
            x1 = s8(-q - e); x21 = s8(-q2 - e2);
            v1 = r == 0 ? s8(-q - e) : r < long_thres ? s8(-e) : r == long_thres ? s8(long_diff) : s8(-e2);
        }
        if (en >= r) {
            y[r] = s8(-q - e); y2[r] = s8(-q2 - e2);
            u[r] = r == 0 ? s8(-q - e) : r < long_thres ? s8(-e) : r == long_thres ? s8(long_diff) : s8(-e2);
        }
        /* scores: 16 at a time from st0 (unaligned), so the last store runs up to 15 bytes past en0 - into sf when s ends there */
        for (t = st0; t <= en0; t += 16) {
            int l;
            int8_t tmp[16];
            for (l = 0; l < 16; ++l) {
                const size_t is = (size_t)(sf - mem) + (size_t)(t + l), iq = (size_t)(qrr - mem) + (size_t)(t + l);
                const uint8_t sq = is < cap ? mem[is] : 0, sqr = iq < cap ? mem[iq] : 0;
                tmp[l] = (sq == (uint8_t)(m - 1) || sqr == (uint8_t)(m - 1)) ? sc_N : (sq == sqr ? sc_mch : sc_mis);
            }
            for (l = 0; l < 16; ++l) { const size_t id = (size_t)((uint8_t *)s - mem) + (size_t)(t + l); if (id < cap) mem[id] = (uint8_t)tmp[l]; }
        }
        /* core loop over the rounded range: every cell reads the previous anti-diagonal only */
        memcpy(ox + st, x + st, (size_t)(en - st + 1)); memcpy(ov + st, v + st, (size_t)(en - st + 1)); memcpy(ox2 + st, x2 + st, (size_t)(en - st + 1));
        if (with_cigar) { off[r] = st; off_end[r] = en; }
        for (t = st; t <= en; ++t) {
            int8_t z = s[t];
            const int8_t xt1 = t == st ? x1 : ox[t - 1], vt1 = t == st ? v1 : ov[t - 1], x2t1 = t == st ? x21 : ox2[t - 1];
            const int8_t ut = u[t];
            int8_t a = s8(xt1 + vt1), b = s8(y[t] + ut), a2 = s8(x2t1 + vt1), b2 = s8(y2[t] + ut), tmp;
            uint8_t d = 0;
            if (!with_cigar || !(flag & MMA_EZ_RIGHT)) {      /* gap left-alignment (also the score-only form) */
                d = a > z ? 1 : 0;  z = z > a ? z : a;
                d = b > z ? 2 : d;  z = z > b ? z : b;
                d = a2 > z ? 3 : d; z = z > a2 ? z : a2;
                d = b2 > z ? 4 : d; z = z > b2 ? z : b2;
                z = z < sc_mch ? z : sc_mch;
                u[t] = s8(z - vt1); v[t] = s8(z - ut);
                tmp = s8(z - q);  a = s8(a - tmp);  b = s8(b - tmp);
                tmp = s8(z - q2); a2 = s8(a2 - tmp); b2 = s8(b2 - tmp);
                x[t] = s8((a > 0 ? a : 0) - qe);    d |= a > 0 ? 0x08 : 0;
                y[t] = s8((b > 0 ? b : 0) - qe);    d |= b > 0 ? 0x10 : 0;
                x2[t] = s8((a2 > 0 ? a2 : 0) - qe2); d |= a2 > 0 ? 0x20 : 0;
                y2[t] = s8((b2 > 0 ? b2 : 0) - qe2); d |= b2 > 0 ? 0x40 : 0;
            } else {                                          /* gap right-alignment */
                d = z > a ? 0 : 1;  z = z > a ? z : a;
                d = z > b ? d : 2;  z = z > b ? z : b;
                d = z > a2 ? d : 3; z = z > a2 ? z : a2;
                d = z > b2 ? d : 4; z = z > b2 ? z : b2;
                z = z < sc_mch ? z : sc_mch;
                u[t] = s8(z - vt1); v[t] = s8(z - ut);
                tmp = s8(z - q);  a = s8(a - tmp);  b = s8(b - tmp);
                tmp = s8(z - q2); a2 = s8(a2 - tmp); b2 = s8(b2 - tmp);
                x[t] = s8((0 > a ? 0 : a) - qe);    d |= 0 > a ? 0 : 0x08;
                y[t] = s8((0 > b ? 0 : b) - qe);    d |= 0 > b ? 0 : 0x10;
                x2[t] = s8((0 > a2 ? 0 : a2) - qe2); d |= 0 > a2 ? 0 : 0x20;
                y2[t] = s8((0 > b2 ? 0 : b2) - qe2); d |= 0 > b2 ? 0 : 0x40;
            }
            if (with_cigar) p[(size_t)r * n_col + (t - st)] = d;
        }
        if (!approx_max) {      /* exact maximum through a 32-bit score array */
            int32_t max_H, max_t;
            if (r > 0) {
                int32_t HH[4], tt[4], en1 = st0 + (en0 - st0) / 4 * 4, i;
                max_H = H[en0] = en0 > 0 ? H[en0 - 1] + u[en0] : H[en0] + v[en0];
                max_t = en0;
                for (i = 0; i < 4; ++i) { HH[i] = max_H; tt[i] = max_t; }
                for (t = st0; t < en1; t += 4)
                    for (i = 0; i < 4; ++i) {
                        H[t + i] += (int32_t)v[t + i];
                        if (H[t + i] > HH[i]) { HH[i] = H[t + i]; tt[i] = t; }
                    }
                for (i = 0; i < 4; ++i)
                    if (max_H < HH[i]) { max_H = HH[i]; max_t = tt[i] + i; }
                for (; t < en0; ++t) {
                    H[t] += (int32_t)v[t];
                    if (H[t] > max_H) { max_H = H[t]; max_t = t; }
                }
            } else { H[0] = v[0] - qe; max_H = H[0]; max_t = 0; }
            if (en0 == tlen - 1 && H[en0] > ez->mte) { ez->mte = H[en0]; ez->mte_q = r - en; }
            if (r - st0 == qlen - 1 && H[st0] > ez->mqe) { ez->mqe = H[st0]; ez->mqe_t = st0; }
            if (apply_zdrop(ez, max_H, r, max_t, zdrop, e2)) break;
            if (r == qlen + tlen - 2 && en0 == tlen - 1) ez->score = H[tlen - 1];
        } else {                /* approximate maximum: follow one path */
            if (r > 0) {
                if (last_H0_t >= st0 && last_H0_t <= en0 && last_H0_t + 1 >= st0 && last_H0_t + 1 <= en0) {
                    const int32_t d0 = v[last_H0_t], d1 = u[last_H0_t + 1];
                    if (d0 > d1) H0 += d0;
                    else { H0 += d1; ++last_H0_t; }
                } else if (last_H0_t >= st0 && last_H0_t <= en0) H0 += v[last_H0_t];
                else { ++last_H0_t; H0 += u[last_H0_t]; }
            } else { H0 = v[0] - qe; last_H0_t = 0; }
            if ((flag & MMA_EZ_APPROX_DROP) && apply_zdrop(ez, H0, r, last_H0_t, zdrop, e2)) break;
            if (r == qlen + tlen - 2 && en0 == tlen - 1) ez->score = H0;
        }
        last_st = st; last_en = en;
    }
    if (with_cigar) {
        const int rev_cigar = !!(flag & MMA_EZ_REV_CIGAR);
        if (!ez->zdropped && !(flag & MMA_EZ_EXTZ_ONLY))
            backtrack(rev_cigar, p, off, off_end, n_col, tlen - 1, qlen - 1, &ez->m_cigar, &ez->n_cigar, &ez->cigar);
        else if (!ez->zdropped && (flag & MMA_EZ_EXTZ_ONLY) && ez->mqe + end_bonus > (int)ez->max) {
            ez->reach_end = 1;
            backtrack(rev_cigar, p, off, off_end, n_col, ez->mqe_t, qlen - 1, &ez->m_cigar, &ez->n_cigar, &ez->cigar);
        } else if (ez->max_t >= 0 && ez->max_q >= 0)
            backtrack(rev_cigar, p, off, off_end, n_col, ez->max_t, ez->max_q, &ez->m_cigar, &ez->n_cigar, &ez->cigar);
    }
}

/* ------------------------------------------------------------------------------------------------
 * regions (mm_reg1_t / mm_extra_t: the fields the decision reads)
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t dp_score, dp_max, n_ambi;
    int n_cigar, m_cigar; uint32_t *cigar;
} extra_t;

typedef struct {
    int32_t id, cnt, rid, score, qs, qe, rs, re, parent, subsc, as, mlen, blen, n_sub;
    uint32_t hash;
    int rev, inv, split_inv, seg_split, strand_retained, split;
    float div;
    extra_t *p;
} reg_t;

#define SEED_LONG_JOIN (1ULL << 40)
#define SEED_IGNORE    (1ULL << 41)
#define SEED_TANDEM    (1ULL << 42)

#define PARENT_UNSET (-1)
#define PARENT_TMP_PRI (-2)

static uint64_t hash64(uint64_t key)
{
    key = (~key + (key << 21));
    key = key ^ key >> 24;
    key = ((key + (key << 3)) + (key << 8));
    key = key ^ key >> 14;
    key = ((key + (key << 2)) + (key << 4));
    key = key ^ key >> 28;
    key = (key + (key << 31));
    return key;
}

static uint32_t wang_hash(uint32_t key)
{
    key += ~(key << 15);
    key ^= (key >> 10);
    key += (key << 3);
    key ^= (key >> 6);
    key += ~(key << 11);
    key ^= (key >> 16);
    return key;
}

static void reg_set_coor(reg_t *r, int32_t qlen, const mma_anchor *a)
{
    const int32_t k = r->as, q_span = (int32_t)(a[k].y >> 32 & 0xff);
    r->rev = (int)(a[k].x >> 63);
    r->rid = (int32_t)(a[k].x << 1 >> 33);
    r->rs = (int32_t)a[k].x + 1 > q_span ? (int32_t)a[k].x + 1 - q_span : 0;
    r->re = (int32_t)a[k + r->cnt - 1].x + 1;
    if (!r->rev) {
        r->qs = (int32_t)a[k].y + 1 - q_span;
        r->qe = (int32_t)a[k + r->cnt - 1].y + 1;
    } else {
        r->qs = qlen - ((int32_t)a[k + r->cnt - 1].y + 1);
        r->qe = qlen - ((int32_t)a[k].y + 1 - q_span);
    }
    {   /* mm_cal_fuzzy_len: what mlen / blen hold until mm_update_extra replaces them (mm_fix_bad_ends reads mlen) */
        int32_t i;
        r->mlen = r->blen = 0;
        if (r->cnt <= 0) return;
        r->mlen = r->blen = (int32_t)(a[r->as].y >> 32 & 0xff);
        for (i = r->as + 1; i < r->as + r->cnt; ++i) {
            const int32_t span = (int32_t)(a[i].y >> 32 & 0xff);
            const int32_t tl = (int32_t)a[i].x - (int32_t)a[i - 1].x, ql = (int32_t)a[i].y - (int32_t)a[i - 1].y;
            r->blen += tl > ql ? tl : ql;
            r->mlen += tl > span && ql > span ? span : tl < ql ? tl : ql;
        }
    }
}

typedef struct { uint64_t x, y; } u128;
static void sort128x_stable(u128 *a, int n)
{   /* radix_sort_128x is an insertion sort below 64 elements (stable); larger inputs: documented choice, as in mm_oracle.c */
    int i, j;
    for (i = 1; i < n; ++i) {
        u128 t = a[i];
        for (j = i; j > 0 && a[j - 1].x > t.x; --j) a[j] = a[j - 1];
        a[j] = t;
    }
}

static reg_t *gen_regs(uint32_t hash, int qlen, int n_u, const uint64_t *u, const mma_anchor *a)
{
    u128 *z, tmp;
    reg_t *r;
    int i, k;
    if (n_u == 0) return 0;
    z = (u128 *)malloc(sizeof(u128) * (size_t)n_u);
    for (i = k = 0; i < n_u; ++i) {
        const uint32_t h = (uint32_t)hash64((hash64(a[k].x) + hash64(a[k].y)) ^ hash);
        z[i].x = u[i] ^ h;
        z[i].y = (uint64_t)k << 32 | (uint32_t)(int32_t)u[i];
        k += (int32_t)u[i];
    }
    sort128x_stable(z, n_u);
    for (i = 0; i < n_u >> 1; ++i) { tmp = z[i]; z[i] = z[n_u - 1 - i]; z[n_u - 1 - i] = tmp; }
    r = (reg_t *)calloc((size_t)n_u, sizeof(reg_t));
    for (i = 0; i < n_u; ++i) {
        reg_t *ri = &r[i];
        ri->id = i;
        ri->parent = PARENT_UNSET;
        ri->score = (int32_t)(z[i].x >> 32);
        ri->hash = (uint32_t)z[i].x;
        ri->cnt = (int32_t)z[i].y;
        ri->as = (int32_t)(z[i].y >> 32);
        ri->div = -1.0f;
        reg_set_coor(ri, qlen, a);
    }
    free(z);
    return r;
}

static int cmp_u64(const void *a, const void *b)
{
    const uint64_t p = *(const uint64_t *)a, q = *(const uint64_t *)b;
    return p < q ? -1 : p > q;
}

static void set_parent(float mask_level, int mask_len, int n, reg_t *r)
{
    int i, j, k, *w;
    uint64_t *cov;
    if (n <= 0) return;
    for (i = 0; i < n; ++i) r[i].id = i;
    cov = (uint64_t *)malloc((size_t)n * sizeof(uint64_t));
    w = (int *)malloc((size_t)n * sizeof(int));
    w[0] = 0; r[0].parent = 0;
    for (i = 1, k = 1; i < n; ++i) {
        reg_t *ri = &r[i];
        const int si = ri->qs, ei = ri->qe;
        int n_cov = 0, uncov_len = 0;
        for (j = 0; j < k; ++j) {
            const reg_t *rp = &r[w[j]];
            int sj = rp->qs, ej = rp->qe;
            if (ej <= si || sj >= ei) continue;
            if (sj < si) sj = si;
            if (ej > ei) ej = ei;
            cov[n_cov++] = (uint64_t)sj << 32 | (uint32_t)ej;
        }
        if (n_cov > 0) {
            int x = si;
            qsort(cov, (size_t)n_cov, sizeof(uint64_t), cmp_u64);
            for (j = 0; j < n_cov; ++j) {
                if ((int)(cov[j] >> 32) > x) uncov_len += (int)(cov[j] >> 32) - x;
                x = (int32_t)cov[j] > x ? (int32_t)cov[j] : x;
            }
            if (ei > x) uncov_len += ei - x;
            for (j = 0; j < k; ++j) {
                reg_t *rp = &r[w[j]];
                const int sj = rp->qs, ej = rp->qe;
                int min, max, ol;
                if (ej <= si || sj >= ei) continue;
                min = ej - sj < ei - si ? ej - sj : ei - si;
                max = ej - sj > ei - si ? ej - sj : ei - si;
                ol = si < sj ? (ei < sj ? 0 : ei < ej ? ei - sj : ej - sj) : (ej < si ? 0 : ej < ei ? ej - si : ei - si);
                if ((float)ol / min - (float)uncov_len / max > mask_level && uncov_len <= mask_len) {
                    ri->parent = rp->parent;
                    rp->subsc = rp->subsc > ri->score ? rp->subsc : ri->score;
                    if (ri->cnt >= rp->cnt) ++rp->n_sub;
                    break;
                }
            }
        } else j = k;
        if (j == k) { w[k++] = i; ri->parent = i; ri->n_sub = 0; }
    }
    free(cov); free(w);
}

static void sync_regs(int n_regs, reg_t *regs)
{   /* mm_sync_regs: id = position; parent follows (a dropped parent leaves PARENT_UNSET) */
    int *tmp, i, max_id = -1, n_tmp;
    if (n_regs <= 0) return;
    for (i = 0; i < n_regs; ++i) max_id = max_id > regs[i].id ? max_id : regs[i].id;
    n_tmp = max_id + 1;
    tmp = (int *)malloc((size_t)(n_tmp + 1) * sizeof(int));
    for (i = 0; i < n_tmp; ++i) tmp[i] = -1;
    for (i = 0; i < n_regs; ++i) if (regs[i].id >= 0) tmp[regs[i].id] = i;
    for (i = 0; i < n_regs; ++i) {
        reg_t *r = &regs[i];
        r->id = i;
        if (r->parent == PARENT_TMP_PRI) r->parent = i;
        else if (r->parent >= 0 && tmp[r->parent] >= 0) r->parent = tmp[r->parent];
        else r->parent = PARENT_UNSET;
    }
    free(tmp);
}

static void select_sub(float pri_ratio, int min_diff, int best_n, int check_strand, int min_strand_sc, int *n_, reg_t *r)
{
    if (pri_ratio > 0.0f && *n_ > 0) {
        int i, k, n = *n_, n_2nd = 0;
        for (i = k = 0; i < n; ++i) {
            const int p = r[i].parent;
            if (p == i || r[i].inv) r[k++] = r[i];
            else if ((r[i].score >= r[p].score * pri_ratio || r[i].score + min_diff >= r[p].score) && n_2nd < best_n) {
                if (!(r[i].qs == r[p].qs && r[i].qe == r[p].qe && r[i].rid == r[p].rid && r[i].rs == r[p].rs && r[i].re == r[p].re)) { r[k++] = r[i]; ++n_2nd; }
            } else if (check_strand && n_2nd < best_n && r[i].score > min_strand_sc && r[p].rev != r[i].rev && r[p].rid == r[i].rid && r[i].rs < r[p].re && r[i].re > r[p].rs) {
                r[i].strand_retained = 1;
                r[k++] = r[i]; ++n_2nd;
            }
        }
        if (k != n) sync_regs(k, r);
        *n_ = k;
    }
}
/* NB: upstream's loop above reads r[p] after r[] has been compacted in place (r[k++] = r[i] with k <= i), so for p > k the
 * parent it sees may already have been overwritten by a later region that moved up.  Primaries are never dropped and move
 * up in order, and a parent index p is a primary's ORIGINAL position: the entry at position p is therefore only intact
 * while no region before it has been dropped.  The in-place behaviour is reproduced literally (same statements). */

/* ------------------------------------------------------------------------------------------------
 * mm_align1, short-read branch
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    const mmo_opts *o; const uint8_t *ref; const uint64_t *cstart; uint32_t n_contigs;
    int32_t qlen; uint8_t *qseq0[2];
    mma_anchor *a; int32_t n_a;      /* n_a: anchors left by mm_squeeze_a (the long-read branch looks at its neighbours') */
} actx_t;

static int32_t contig_len(const actx_t *c, int32_t rid) { return (int32_t)(c->cstart[rid + 1] - c->cstart[rid]); }

static void getseq(const actx_t *c, int32_t rid, int32_t st, int32_t en, uint8_t *out)
{   /* mm_idx_getseq: nt4 codes, 4 = ambiguous */
    const uint64_t g0 = c->cstart[rid];
    int32_t i;
    if (en > contig_len(c, rid)) en = contig_len(c, rid);
    for (i = st; i < en; ++i) { const uint64_t g = g0 + (uint64_t)i; out[i - st] = (c->ref[g >> 1] >> ((g & 1) * 4)) & 15; }
}

static void seq_rev(int32_t len, uint8_t *seq)
{
    int32_t i; uint8_t t;
    for (i = 0; i < len >> 1; ++i) { t = seq[i]; seq[i] = seq[len - 1 - i]; seq[len - 1 - i] = t; }
}

static void append_cigar(reg_t *r, int n_cigar, const uint32_t *cigar)
{
    extra_t *p;
    if (n_cigar == 0) return;
    if (r->p == 0) r->p = (extra_t *)calloc(1, sizeof(extra_t));
    p = r->p;
    if (p->n_cigar + n_cigar > p->m_cigar) {
        p->m_cigar = (p->n_cigar + n_cigar) * 2 + 8;
        p->cigar = (uint32_t *)realloc(p->cigar, (size_t)p->m_cigar * 4);
    }
    if (p->n_cigar > 0 && (p->cigar[p->n_cigar - 1] & 0xf) == (cigar[0] & 0xf)) {
        p->cigar[p->n_cigar - 1] += (cigar[0] >> 4) << 4;
        if (n_cigar > 1) memcpy(p->cigar + p->n_cigar, cigar + 1, (size_t)(n_cigar - 1) * 4);
        p->n_cigar += n_cigar - 1;
    } else {
        memcpy(p->cigar + p->n_cigar, cigar, (size_t)n_cigar * 4);
        p->n_cigar += n_cigar;
    }
}

static void max_stretch(const reg_t *r, const mma_anchor *a, int32_t *as, int32_t *cnt)
{
    int32_t i, score, max_score, len, max_i, max_len;
    *as = r->as; *cnt = r->cnt;
    if (r->cnt < 2) return;
    max_score = -1; max_i = -1; max_len = 0;
    score = (int32_t)(a[r->as].y >> 32 & 0xff); len = 1;
    for (i = r->as; i < r->as + r->cnt - 1; ++i) {
        const int32_t q_span = (int32_t)(a[i + 1].y >> 32 & 0xff);
        const int32_t lr = (int32_t)a[i + 1].x - (int32_t)a[i].x, lq = (int32_t)a[i + 1].y - (int32_t)a[i].y;
        if (lq == lr) { score += lq < q_span ? lq : q_span; ++len; }
        else {
            if (score > max_score) { max_score = score; max_len = len; max_i = i - len + 1; }
            score = q_span; len = 1;
        }
    }
    if (score > max_score) { max_score = score; max_len = len; max_i = i - len + 1; }
    *as = max_i; *cnt = max_len;
}

static void update_max_zdrop(int32_t score, int i, int j, int32_t *max, int *max_i, int *max_j, int e, int *max_zdrop)
{
    if (score < *max) {
        const int li = i - *max_i, lj = j - *max_j;
        const int diff = li > lj ? li - lj : lj - li;
        const int z = *max - score - diff * e;
        if (z > *max_zdrop) *max_zdrop = z;
    } else { *max = score; *max_i = i; *max_j = j; }
}

static int test_zdrop(const mmo_opts *o, const uint8_t *qseq, const uint8_t *tseq, int n_cigar, const uint32_t *cigar, const int8_t *mat)
{   /* mm_test_zdrop; the inversion test is skipped for MM_F_SR */
    int k;
    int32_t score = 0, max = INT32_MIN, max_i = -1, max_j = -1, i = 0, j = 0, max_zdrop = 0;
    for (k = 0; k < n_cigar; ++k) {
        const uint32_t op = cigar[k] & 0xf, len = cigar[k] >> 4;
        uint32_t l;
        if (op == CIG_MATCH) {
            for (l = 0; l < len; ++l) {
                score += mat[tseq[i + l] * 5 + qseq[j + l]];
                update_max_zdrop(score, i + (int)l, j + (int)l, &max, &max_i, &max_j, o->e, &max_zdrop);
            }
            i += (int)len; j += (int)len;
        } else if (op == CIG_INS || op == CIG_DEL) {
            score -= o->q + o->e * (int32_t)len;
            if (op == CIG_INS) j += (int)len; else i += (int)len;
            update_max_zdrop(score, i, j, &max, &max_i, &max_j, o->e, &max_zdrop);
        }
    }
    return max_zdrop > o->zdrop ? 1 : 0;
}

static void align_pair(const mmo_opts *o, int qlen, const uint8_t *qseq, int tlen, const uint8_t *tseq, const int8_t *mat, int w,
                       int end_bonus, int zdrop, int flag, mma_ez *ez)
{   /* mm_align_pair: q != q2 or e != e2 for every preset -> ksw_extd2_sse */
    if ((int64_t)tlen * qlen > 100000000ll) { ez_reset(ez); ez->zdropped = 1; return; }     /* max_sw_mat */
    mma_ksw_extd2(qlen, qseq, tlen, tseq, 5, mat, (int8_t)o->q, (int8_t)o->e, (int8_t)o->q2, (int8_t)o->e2, w, zdrop, end_bonus, flag, ez);
}

static void split_reg(reg_t *r, reg_t *r2, int n, int qlen, const mma_anchor *a)
{
    if (n <= 0 || n >= r->cnt) return;
    *r2 = *r;
    r2->id = -1;
    r2->p = 0;
    r2->split_inv = 0;
    r2->cnt = r->cnt - n;
    r2->score = (int32_t)(r->score * ((float)r2->cnt / r->cnt) + .499);
    r2->as = r->as + n;
    if (r->parent == r->id) r2->parent = PARENT_TMP_PRI;
    reg_set_coor(r2, qlen, a);
    r->cnt -= r2->cnt;
    r->score -= r2->score;
    reg_set_coor(r, qlen, a);
    r->split |= 1; r2->split |= 2;
}

static void fix_cigar(reg_t *r, const uint8_t *qseq, const uint8_t *tseq, int *qshift, int *tshift)
{
    extra_t *p = r->p;
    int32_t toff = 0, qoff = 0, to_shrink = 0;
    int k;
    *qshift = *tshift = 0;
    if (p->n_cigar <= 1) return;
    for (k = 0; k < p->n_cigar; ++k) {       /* indel left alignment */
        const uint32_t op = p->cigar[k] & 0xf, len = p->cigar[k] >> 4;
        if (len == 0) to_shrink = 1;
        if (op == CIG_MATCH) { toff += (int32_t)len; qoff += (int32_t)len; }
        else if (op == CIG_INS || op == CIG_DEL) {
            if (k > 0 && k < p->n_cigar - 1 && (p->cigar[k - 1] & 0xf) == 0 && (p->cigar[k + 1] & 0xf) == 0) {
                int l;
                const int prev_len = (int)(p->cigar[k - 1] >> 4);
                if (op == CIG_INS) { for (l = 0; l < prev_len; ++l) if (qseq[qoff - 1 - l] != qseq[qoff + (int32_t)len - 1 - l]) break; }
                else { for (l = 0; l < prev_len; ++l) if (tseq[toff - 1 - l] != tseq[toff + (int32_t)len - 1 - l]) break; }
                if (l > 0) { p->cigar[k - 1] -= (uint32_t)l << 4; p->cigar[k + 1] += (uint32_t)l << 4; qoff -= l; toff -= l; }
                if (l == prev_len) to_shrink = 1;
            }
            if (op == CIG_DEL) toff += (int32_t)len; else qoff += (int32_t)len;
        }
    }
    for (k = 0; k < p->n_cigar - 2; ++k) {   /* fix CIGAR like 5I6D7I */
        if ((p->cigar[k] & 0xf) > 0 && (p->cigar[k] & 0xf) + (p->cigar[k + 1] & 0xf) == 3) {
            int l;
            uint32_t s[3] = {0, 0, 0};
            for (l = k; l < p->n_cigar; ++l) {
                const uint32_t op = p->cigar[l] & 0xf;
                if (op == CIG_INS || op == CIG_DEL || p->cigar[l] >> 4 == 0) s[op] += p->cigar[l] >> 4;
                else break;
            }
            if (s[1] > 0 && s[2] > 0 && l - k > 2) {
                p->cigar[k] = s[1] << 4 | CIG_INS;
                p->cigar[k + 1] = s[2] << 4 | CIG_DEL;
                for (k += 2; k < l; ++k) p->cigar[k] &= 0xf;
                to_shrink = 1;
            }
            k = l;
        }
    }
    if (to_shrink) {
        int l = 0;
        for (k = 0; k < p->n_cigar; ++k) if (p->cigar[k] >> 4 != 0) p->cigar[l++] = p->cigar[k];
        p->n_cigar = l;
        for (k = l = 0; k < p->n_cigar; ++k)
            if (k == p->n_cigar - 1 || (p->cigar[k] & 0xf) != (p->cigar[k + 1] & 0xf)) p->cigar[l++] = p->cigar[k];
            else p->cigar[k + 1] += p->cigar[k] >> 4 << 4;
        p->n_cigar = l;
    }
    if ((p->cigar[0] & 0xf) == CIG_INS || (p->cigar[0] & 0xf) == CIG_DEL) {      /* leading I or D */
        const int32_t l = (int32_t)(p->cigar[0] >> 4);
        if ((p->cigar[0] & 0xf) == CIG_INS) {
            if (r->rev) r->qe -= l; else r->qs += l;
            *qshift = l;
        } else { r->rs += l; *tshift = l; }
        --p->n_cigar;
        memmove(p->cigar, p->cigar + 1, (size_t)p->n_cigar * 4);
    }
}

static void update_extra(reg_t *r, const uint8_t *qseq, const uint8_t *tseq, const int8_t *mat, int8_t q, int8_t e, int log_gap)
{   /* mm_update_extra; log_gap = !MM_F_SR: a gap of len costs q + e * log2(1 + len) in the running score behind dp_max */
    int k;
    uint32_t l;
    int32_t qshift, tshift, toff = 0, qoff = 0;
    double s = 0.0, max = 0.0;
    extra_t *p = r->p;
    if (p == 0) return;
    fix_cigar(r, qseq, tseq, &qshift, &tshift);
    qseq += qshift; tseq += tshift;
    r->blen = r->mlen = 0;
    for (k = 0; k < p->n_cigar; ++k) {
        const uint32_t op = p->cigar[k] & 0xf, len = p->cigar[k] >> 4;
        if (op == CIG_MATCH) {
            int n_ambi = 0, n_diff = 0;
            for (l = 0; l < len; ++l) {
                const int cq = qseq[qoff + (int32_t)l], ct = tseq[toff + (int32_t)l];
                if (ct > 3 || cq > 3) ++n_ambi;
                else if (ct != cq) ++n_diff;
                s += mat[ct * 5 + cq];
                if (s < 0) s = 0;
                else max = max > s ? max : s;
            }
            r->blen += (int32_t)len - n_ambi; r->mlen += (int32_t)len - (n_ambi + n_diff); p->n_ambi += n_ambi;
            toff += (int32_t)len; qoff += (int32_t)len;
        } else if (op == CIG_INS) {
            int n_ambi = 0;
            for (l = 0; l < len; ++l) if (qseq[qoff + (int32_t)l] > 3) ++n_ambi;
            r->blen += (int32_t)len - n_ambi; p->n_ambi += n_ambi;
            if (log_gap) s -= q + (double)e * mmo_log2((float)(1.0 + len));
            else s -= q + e * (int32_t)len;
            if (s < 0) s = 0;
            qoff += (int32_t)len;
        } else if (op == CIG_DEL) {
            int n_ambi = 0;
            for (l = 0; l < len; ++l) if (tseq[toff + (int32_t)l] > 3) ++n_ambi;
            r->blen += (int32_t)len - n_ambi; p->n_ambi += n_ambi;
            if (log_gap) s -= q + (double)e * mmo_log2((float)(1.0 + len));
            else s -= q + e * (int32_t)len;
            if (s < 0) s = 0;
            toff += (int32_t)len;
        }
    }
    p->dp_max = (int32_t)(max + .499);
}

static void align1_sr(actx_t *c, reg_t *r, reg_t *r2, mma_ez *ez)
{
    const mmo_opts *o = c->o;
    mma_anchor *a = c->a;
    const int32_t qlen = c->qlen;
    const int32_t rid = (int32_t)(a[r->as].x << 1 >> 33), rev = (int32_t)(a[r->as].x >> 63);
    uint8_t *tseq, *qseq;
    int32_t i, l, bw, bw_long, dropped = 0, rs0, re0, qs0, qe0, as1, cnt1;
    int32_t rs, re, qs, qe, rs1, qs1, re1, qe1;
    int8_t mat[25];

    r2->cnt = 0;
    if (r->cnt == 0) return;
    mma_gen_simple_mat(5, mat, (int8_t)o->a, (int8_t)o->b, (int8_t)o->sc_ambi);
    bw = (int)(o->bw * 1.5 + 1.);
    bw_long = (int)(o->bw_long * 1.5 + 1.);
    if (bw_long < bw) bw_long = bw;

    max_stretch(r, a, &as1, &cnt1);
    rs = (int32_t)a[as1].x + 1 - (int32_t)(a[as1].y >> 32 & 0xff);
    qs = (int32_t)a[as1].y + 1 - (int32_t)(a[as1].y >> 32 & 0xff);
    re = (int32_t)a[as1 + cnt1 - 1].x + 1;
    qe = (int32_t)a[as1 + cnt1 - 1].y + 1;

    qs0 = 0; qe0 = qlen;
    l = qs;
    l += l * o->a + o->end_bonus > o->q ? (l * o->a + o->end_bonus - o->q) / o->e : 0;
    rs0 = rs - l > 0 ? rs - l : 0;
    l = qlen - qe;
    l += l * o->a + o->end_bonus > o->q ? (l * o->a + o->end_bonus - o->q) / o->e : 0;
    re0 = re + l < contig_len(c, rid) ? re + l : contig_len(c, rid);
    tseq = (uint8_t *)malloc((size_t)(re0 - rs0) + 16);

    if (qs > 0 && rs > 0) {       /* left extension */
        qseq = &c->qseq0[rev][qs0];
        getseq(c, rid, rs0, rs, tseq);
        seq_rev(qs - qs0, qseq);
        seq_rev(rs - rs0, tseq);
        align_pair(o, qs - qs0, qseq, rs - rs0, tseq, mat, bw, o->end_bonus, r->split_inv ? o->zdrop_inv : o->zdrop,
                   MMA_EZ_EXTZ_ONLY | MMA_EZ_RIGHT | MMA_EZ_REV_CIGAR, ez);
        if (ez->n_cigar > 0) { append_cigar(r, ez->n_cigar, ez->cigar); r->p->dp_score += (int32_t)ez->max; }
        rs1 = rs - (ez->reach_end ? ez->mqe_t + 1 : ez->max_t + 1);
        qs1 = qs - (ez->reach_end ? qs - qs0 : ez->max_q + 1);
        seq_rev(qs - qs0, qseq);
    } else { rs1 = rs; qs1 = qs; }
    re1 = rs; qe1 = qs;

    for (i = cnt1 - 1; i < cnt1; ++i) {       /* gap filling: short-read mode aligns the whole stretch ungapped */
        int j, zdrop_code;
        re = (int32_t)a[as1 + i].x + 1;
        qe = (int32_t)a[as1 + i].y + 1;
        re1 = re; qe1 = qe;
        qseq = &c->qseq0[rev][qs];
        getseq(c, rid, rs, re, tseq);
        ez_reset(ez);
        for (j = 0, ez->score = 0; j < qe - qs; ++j) {
            if (qseq[j] >= 4 || tseq[j] >= 4) ez->score += o->e2;
            else ez->score += qseq[j] == tseq[j] ? o->a : -o->b;
        }
        push_cigar(&ez->n_cigar, &ez->m_cigar, &ez->cigar, CIG_MATCH, qe - qs);
        if ((zdrop_code = test_zdrop(o, qseq, tseq, ez->n_cigar, ez->cigar, mat)) != 0)
            align_pair(o, qe - qs, qseq, re - rs, tseq, mat, bw_long, -1, zdrop_code == 2 ? o->zdrop_inv : o->zdrop, 0, ez);
        if (ez->n_cigar > 0) append_cigar(r, ez->n_cigar, ez->cigar);
        if (ez->zdropped) {
            if (!r->p) r->p = (extra_t *)calloc(1, sizeof(extra_t));
            for (j = i - 1; j >= 0; --j) if ((int32_t)a[as1 + j].x <= rs + ez->max_t) break;
            dropped = 1;
            if (j < 0) j = 0;
            r->p->dp_score += (int32_t)ez->max;
            re1 = rs + (ez->max_t + 1);
            qe1 = qs + (ez->max_q + 1);
            if (cnt1 - (j + 1) >= o->min_cnt) split_reg(r, r2, as1 + j + 1 - r->as, qlen, a);
            break;
        } else r->p->dp_score += ez->score;
        rs = re; qs = qe;
    }

    if (!dropped && qe < qe0 && re < re0) {   /* right extension */
        qseq = &c->qseq0[rev][qe];
        getseq(c, rid, re, re0, tseq);
        align_pair(o, qe0 - qe, qseq, re0 - re, tseq, mat, bw, o->end_bonus, o->zdrop, MMA_EZ_EXTZ_ONLY, ez);
        if (ez->n_cigar > 0) { append_cigar(r, ez->n_cigar, ez->cigar); r->p->dp_score += (int32_t)ez->max; }
        re1 = re + (ez->reach_end ? ez->mqe_t + 1 : ez->max_t + 1);
        qe1 = qe + (ez->reach_end ? qe0 - qe : ez->max_q + 1);
    }

    r->rs = rs1; r->re = re1;
    if (rev) { r->qs = qlen - qe1; r->qe = qlen - qs1; }
    else { r->qs = qs1; r->qe = qe1; }
    if (r->p) {
        free(tseq);
        tseq = (uint8_t *)malloc((size_t)(re1 - rs1) + 16);
        getseq(c, rid, rs1, re1, tseq);
        update_extra(r, &c->qseq0[r->rev][qs1], tseq, mat, (int8_t)o->q, (int8_t)o->e, 0);
    }
    free(tseq);
}


/* ------------------------------------------------------------------------------------------------
 * long-read presets: what runs between chain_post and mm_align_skeleton (map.c), then mm_align1 without MM_F_SR
 * ---------------------------------------------------------------------------------------------- */
static inline int32_t get_for_qpos(int32_t qlen, const mma_anchor *a)
{   /* the position on the forward strand of the query */
    int32_t x = (int32_t)a->y;
    const int32_t q_span = (int32_t)(a->y >> 32 & 0xff);
    if (a->x >> 63) x = qlen - 1 - (x + 1 - q_span);
    return x;
}

static int get_mini_idx(int qlen, const mma_anchor *a, int32_t n, const uint64_t *mini_pos)
{
    int32_t x, L = 0, R = n - 1;
    x = get_for_qpos(qlen, a);
    while (L <= R) {
        const int32_t m = (int32_t)(((uint64_t)L + (uint64_t)R) >> 1);
        const int32_t y = (int32_t)mini_pos[m];
        if (y < x) L = m + 1;
        else if (y > x) R = m - 1;
        else return m;
    }
    return -1;
}

/* mm_est_err: per-region divergence from the fraction of the read's (unfiltered) minimizers that are anchors of the chain */
static void est_err(const actx_t *c, int n_regs, reg_t *regs, int32_t n, const uint64_t *mini_pos)
{
    const mma_anchor *a = c->a;
    const int32_t qlen = c->qlen;
    int i;
    uint64_t sum_k = 0;
    float avg_k;
    if (n == 0) return;
    for (i = 0; i < n; ++i) sum_k += mini_pos[i] >> 32 & 0xff;
    avg_k = (float)sum_k / n;
    for (i = 0; i < n_regs; ++i) {
        reg_t *r = &regs[i];
        int32_t st, en, j, k, n_match, n_tot, l_ref;
        r->div = -1.0f;
        if (r->cnt == 0) continue;
        st = en = get_mini_idx(qlen, r->rev ? &a[r->as + r->cnt - 1] : &a[r->as], n, mini_pos);
        if (st < 0) continue;
        l_ref = contig_len(c, r->rid);
        for (k = 1, j = st + 1, n_match = 1; j < n && k < r->cnt; ++j) {
            const int32_t q = get_for_qpos(qlen, r->rev ? &a[r->as + r->cnt - 1 - k] : &a[r->as + k]);
            if (q == (int32_t)mini_pos[j]) { ++k; en = j; ++n_match; }
        }
        n_tot = en - st + 1;
        if (r->qs > avg_k && r->rs > avg_k) ++n_tot;
        if (qlen - r->qs > avg_k && l_ref - r->re > avg_k) ++n_tot;      /* qs, not qe: upstream's mm_est_err reads `qlen - r->qs` here (hit.c, v2.28 as recalled by builder and reviewer alike) */
        r->div = n_match >= n_tot ? 0.0f : (float)(1.0 - pow((double)n_match / n_tot, 1.0 / avg_k));
    }
}

static int filter_strand_retained(int n_regs, reg_t *r)
{   /* in place like upstream: r[p] is read after earlier regions moved up */
    int i, k;
    for (i = k = 0; i < n_regs; ++i) {
        const int p = r[i].parent;
        if (!r[i].strand_retained || r[i].div < r[p].div * 5.0f || r[i].div < 0.01f) {
            if (k < i) r[k++] = r[i];
            else ++k;
        }
    }
    return k;
}

static int cmp_u64v(const void *a, const void *b)
{
    const uint64_t p = *(const uint64_t *)a, q = *(const uint64_t *)b;
    return p < q ? -1 : p > q;
}

static int squeeze_a(int n_regs, reg_t *regs, mma_anchor *a)
{   /* mm_squeeze_a: keep only the anchors some region refers to, in the order of as */
    int i, as = 0;
    uint64_t *aux = (uint64_t *)malloc(((size_t)n_regs + 1) * 8);
    for (i = 0; i < n_regs; ++i) aux[i] = (uint64_t)(uint32_t)regs[i].as << 32 | (uint32_t)i;
    qsort(aux, (size_t)n_regs, 8, cmp_u64v);
    for (i = 0; i < n_regs; ++i) {
        reg_t *r = &regs[(int32_t)aux[i]];
        if (r->as != as) {
            memmove(&a[as], &a[r->as], (size_t)r->cnt * 16);
            r->as = as;
        }
        as += r->cnt;
    }
    free(aux);
    return as;
}

static void fix_bad_ends(const reg_t *r, const mma_anchor *a, int bw, int min_match, int32_t *as, int32_t *cnt)
{
    int32_t i, l, m;
    *as = r->as; *cnt = r->cnt;
    if (r->cnt < 3) return;
    m = l = (int32_t)(a[r->as].y >> 32 & 0xff);
    for (i = r->as + 1; i < r->as + r->cnt - 1; ++i) {
        int32_t lq, lr, min, max;
        const int32_t q_span = (int32_t)(a[i].y >> 32 & 0xff);
        if (a[i].y & SEED_LONG_JOIN) break;
        lr = (int32_t)a[i].x - (int32_t)a[i - 1].x;
        lq = (int32_t)a[i].y - (int32_t)a[i - 1].y;
        min = lr < lq ? lr : lq;
        max = lr > lq ? lr : lq;
        if (max - min > l >> 1) *as = i;
        l += min;
        m += min < q_span ? min : q_span;
        if (l >= bw << 1 || (m >= min_match && m >= bw) || m >= r->mlen >> 1) break;
    }
    *cnt = r->as + r->cnt - *as;
    m = l = (int32_t)(a[r->as + r->cnt - 1].y >> 32 & 0xff);
    for (i = r->as + r->cnt - 2; i > *as; --i) {
        int32_t lq, lr, min, max;
        const int32_t q_span = (int32_t)(a[i + 1].y >> 32 & 0xff);
        if (a[i + 1].y & SEED_LONG_JOIN) break;
        lr = (int32_t)a[i + 1].x - (int32_t)a[i].x;
        lq = (int32_t)a[i + 1].y - (int32_t)a[i].y;
        min = lr < lq ? lr : lq;
        max = lr > lq ? lr : lq;
        if (max - min > l >> 1) *cnt = i + 1 - *as;
        l += min;
        m += min < q_span ? min : q_span;
        if (l >= bw << 1 || (m >= min_match && m >= bw) || m >= r->mlen >> 1) break;
    }
}

/* the difference in gap length between anchor i and its predecessor (low 32 bits, as upstream's mixed-width arithmetic leaves it) */
static inline int32_t gap_at(const mma_anchor *a, int32_t i)
{
    return (int32_t)((uint32_t)a[i].y - (uint32_t)a[i - 1].y - ((uint32_t)a[i].x - (uint32_t)a[i - 1].x));
}

static int *collect_long_gaps(int as1, int cnt1, const mma_anchor *a, int min_gap, int *n_)
{
    int i, n, *K;
    *n_ = 0;
    for (i = 1, n = 0; i < cnt1; ++i) {
        const int gap = gap_at(a, as1 + i);
        if (gap < -min_gap || gap > min_gap) ++n;
    }
    if (n <= 1) return 0;
    K = (int *)malloc((size_t)n * sizeof(int));
    for (i = 1, n = 0; i < cnt1; ++i) {
        const int gap = gap_at(a, as1 + i);
        if (gap < -min_gap || gap > min_gap) K[n++] = i;
    }
    *n_ = n;
    return K;
}

static void filter_bad_seeds(int as1, int cnt1, mma_anchor *a, int min_gap, int diff_thres, int max_ext_len, int max_ext_cnt)
{
    int max_st, max_en, n, i, k, max, *K;
    K = collect_long_gaps(as1, cnt1, a, min_gap, &n);
    if (K == 0) return;
    max = 0; max_st = max_en = -1;
    for (k = 0;; ++k) {
        int gap, l, n_ins = 0, n_del = 0, qs, rs, max_diff = 0, max_diff_l = -1;
        if (k == n || k >= max_en) {
            if (max_en > 0)
                for (i = K[max_st]; i < K[max_en]; ++i) a[as1 + i].y |= SEED_IGNORE;
            max = 0; max_st = max_en = -1;
            if (k == n) break;
        }
        i = K[k];
        gap = gap_at(a, as1 + i);
        if (gap > 0) n_ins += gap;
        else n_del += -gap;
        qs = (int32_t)a[as1 + i - 1].y;
        rs = (int32_t)a[as1 + i - 1].x;
        for (l = k + 1; l < n && l <= k + max_ext_cnt; ++l) {
            const int j = K[l];
            int diff;
            if ((int32_t)a[as1 + j].y - qs > max_ext_len || (int32_t)a[as1 + j].x - rs > max_ext_len) break;
            gap = gap_at(a, as1 + j);
            if (gap > 0) n_ins += gap;
            else n_del += -gap;
            diff = n_ins + n_del - abs(n_ins - n_del);
            if (max_diff < diff) { max_diff = diff; max_diff_l = l; }
        }
        if (max_diff > diff_thres && max_diff > max) { max = max_diff; max_st = k; max_en = max_diff_l; }
    }
    free(K);
}

static void filter_bad_seeds_alt(int as1, int cnt1, mma_anchor *a, int min_gap, int max_ext)
{
    int n, k, *K;
    K = collect_long_gaps(as1, cnt1, a, min_gap, &n);
    if (K == 0) return;
    for (k = 0; k < n;) {
        const int i = K[k];
        int l;
        int gap1 = gap_at(a, as1 + i);
        int re1 = (int32_t)a[as1 + i].x;
        int qe1 = (int32_t)a[as1 + i].y;
        gap1 = gap1 > 0 ? gap1 : -gap1;
        for (l = k + 1; l < n; ++l) {
            const int j = K[l];
            int gap2, q_span_pre, rs2, qs2, m;
            if ((int32_t)a[as1 + j].y - qe1 > max_ext || (int32_t)a[as1 + j].x - re1 > max_ext) break;
            gap2 = gap_at(a, as1 + j);
            q_span_pre = (int)(a[as1 + j - 1].y >> 32 & 0xff);
            rs2 = (int32_t)a[as1 + j - 1].x + q_span_pre;
            qs2 = (int32_t)a[as1 + j - 1].y + q_span_pre;
            m = rs2 - re1 < qs2 - qe1 ? rs2 - re1 : qs2 - qe1;
            gap2 = gap2 > 0 ? gap2 : -gap2;
            if (m > gap1 + gap2) break;
            re1 = (int32_t)a[as1 + j].x;
            qe1 = (int32_t)a[as1 + j].y;
            gap1 = gap2;
        }
        if (l > k + 1) {
            int j;
            const int end = K[l - 1];
            for (j = K[k]; j < end; ++j) a[as1 + j].y |= SEED_IGNORE;
            a[as1 + end].y |= SEED_LONG_JOIN;
        }
        k = l;
    }
    free(K);
}

/* ksw_ll_i16 (ksw.c; SSE2 striped Smith-Waterman on 16-bit lanes) as the plain recurrence it evaluates: local alignment,
 * gap of length l costs gapo + l * gape, H >= 0.  The query is padded to a multiple of 8 with columns that score 0 against
 * everything, as the striped profile pads it; *te = the LAST target row whose maximum equals the global one, *qe = the LAST
 * column in striped memory order (position = i / 8 + i % 8 * slen for memory index i) holding it in that row. */
int mma_ksw_ll(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat, int gapo, int gape, int *qe, int *te)
{
    const int slen = (qlen + 7) / 8, qp = slen * 8, gapoe = gapo + gape;
    int32_t *H0, *H1, *E, *Hmax, gmax = 0;
    int i, j;
    *qe = *te = -1;
    if (qlen <= 0) return 0;
    H0 = (int32_t *)calloc((size_t)qp * 4, 4);
    H1 = H0 + qp; E = H1 + qp; Hmax = E + qp;
    for (i = 0; i < tlen; ++i) {
        int32_t f = 0, imax = 0, hd = 0;      /* hd = H(i-1, j-1) */
        const int8_t *row = mat + target[i] * 5;
        for (j = 0; j < qp; ++j) {
            int32_t h = hd + (j < qlen ? row[query[j]] : 0), e = E[j], t;
            h = h > e ? h : e;
            h = h > f ? h : f;
            imax = imax > h ? imax : h;
            H1[j] = h;
            t = h - gapoe; if (t < 0) t = 0;
            e -= gape; if (e < 0) e = 0;
            E[j] = e > t ? e : t;
            f -= gape; if (f < 0) f = 0;
            f = f > t ? f : t;
            hd = H0[j];
        }
        if (imax >= gmax) { gmax = imax; *te = i; memcpy(Hmax, H1, (size_t)qp * 4); }
        { int32_t *S = H1; H1 = H0; H0 = S; }
    }
    {   /* the last hit in memory order: memory index m <-> position m / 8 + m % 8 * slen */
        int m;
        for (m = 0; m < qp; ++m) { const int pos = m / 8 + m % 8 * slen; if (Hmax[pos] == gmax) *qe = pos; }
    }
    free(H0 < H1 ? H0 : H1);
    return gmax;
}

static void update_max_zdrop2(int32_t score, int i, int j, int32_t *max, int *max_i, int *max_j, int e, int *max_zdrop, int pos[2][2])
{
    if (score < *max) {
        const int li = i - *max_i, lj = j - *max_j;
        const int diff = li > lj ? li - lj : lj - li;
        const int z = *max - score - diff * e;
        if (z > *max_zdrop) {
            *max_zdrop = z;
            pos[0][0] = *max_i; pos[0][1] = *max_j;
            pos[1][0] = i; pos[1][1] = j;
        }
    } else { *max = score; *max_i = i; *max_j = j; }
}

static int test_zdrop_lr(const mmo_opts *o, const uint8_t *qseq, const uint8_t *tseq, int n_cigar, const uint32_t *cigar, const int8_t *mat)
{   /* mm_test_zdrop, long-read form: 2 = the most dropped stretch aligns to its own reverse complement (a potential inversion) */
    int k;
    int32_t score = 0, max = INT32_MIN, max_i = -1, max_j = -1, i = 0, j = 0, max_zdrop = 0;
    int pos[2][2] = {{-1, -1}, {-1, -1}}, q_len, t_len;
    for (k = 0; k < n_cigar; ++k) {
        const uint32_t op = cigar[k] & 0xf, len = cigar[k] >> 4;
        uint32_t l;
        if (op == CIG_MATCH) {
            for (l = 0; l < len; ++l) {
                score += mat[tseq[i + l] * 5 + qseq[j + l]];
                update_max_zdrop2(score, i + (int)l, j + (int)l, &max, &max_i, &max_j, o->e, &max_zdrop, pos);
            }
            i += (int)len; j += (int)len;
        } else if (op == CIG_INS || op == CIG_DEL) {
            score -= o->q + o->e * (int32_t)len;
            if (op == CIG_INS) j += (int)len; else i += (int)len;
            update_max_zdrop2(score, i, j, &max, &max_i, &max_j, o->e, &max_zdrop, pos);
        }
    }
    q_len = pos[1][1] - pos[0][1]; t_len = pos[1][0] - pos[0][0];
    if (max_zdrop > o->zdrop_inv && q_len < o->max_gap && t_len < o->max_gap) {
        uint8_t *qseq2 = (uint8_t *)malloc((size_t)(q_len > 0 ? q_len : 1));
        int q_off, t_off;
        for (i = 0; i < q_len; ++i) { const int c = qseq[pos[1][1] - i - 1]; qseq2[i] = c >= 4 ? 4 : 3 - c; }
        score = mma_ksw_ll(q_len, qseq2, t_len, tseq + pos[0][0], mat, o->q, o->e, &q_off, &t_off);
        free(qseq2);
        if (score >= o->min_chain_score * o->a && score >= o->min_dp_max) return 2;
    }
    return max_zdrop > o->zdrop ? 1 : 0;
}

static void align1_lr(actx_t *c, reg_t *r, reg_t *r2, mma_ez *ez)
{
    const mmo_opts *o = c->o;
    mma_anchor *a = c->a;
    const int32_t qlen = c->qlen, n_a = c->n_a, hk = o->k >> 1;
    const int32_t rid = (int32_t)(a[r->as].x << 1 >> 33), rev = (int32_t)(a[r->as].x >> 63);
    const int32_t clen = contig_len(c, rid);
    uint8_t *tseq, *qseq;
    int32_t i, l, bw, bw_long, dropped = 0, rs0, re0, qs0, qe0, as1, cnt1;
    int32_t rs, re, qs, qe, rs1, qs1, re1, qe1;
    int8_t mat[25];

    r2->cnt = 0;
    if (r->cnt == 0) return;
    mma_gen_simple_mat(5, mat, (int8_t)o->a, (int8_t)o->b, (int8_t)o->sc_ambi);
    bw = (int)(o->bw * 1.5 + 1.);
    bw_long = (int)(o->bw_long * 1.5 + 1.);
    if (bw_long < bw) bw_long = bw;

    fix_bad_ends(r, a, o->bw, o->min_chain_score * 2, &as1, &cnt1);
    filter_bad_seeds(as1, cnt1, a, 10, 40, o->max_gap >> 1, 10);
    filter_bad_seeds_alt(as1, cnt1, a, 30, o->max_gap >> 1);
    rs = (int32_t)a[as1].x - hk; qs = (int32_t)a[as1].y - hk;                              /* mm_adjust_minier, no HPC */
    re = (int32_t)a[as1 + cnt1 - 1].x - hk; qe = (int32_t)a[as1 + cnt1 - 1].y - hk;

    /* the window the end extensions may use: bounded by neighbouring chains' anchors on the same strand / contig */
    rs0 = (int32_t)a[r->as].x + 1 - (int32_t)(a[r->as].y >> 32 & 0xff);
    qs0 = (int32_t)a[r->as].y + 1 - (int32_t)(a[r->as].y >> 32 & 0xff);
    if (rs0 < 0) rs0 = 0;
    rs1 = qs1 = 0;
    for (i = r->as - 1, l = 0; i >= 0 && a[i].x >> 32 == a[r->as].x >> 32; --i) {
        const int32_t x = (int32_t)a[i].x + 1 - (int32_t)(a[i].y >> 32 & 0xff);
        const int32_t y = (int32_t)a[i].y + 1 - (int32_t)(a[i].y >> 32 & 0xff);
        if (x < rs0 && y < qs0) {
            if (++l > o->min_cnt) {
                l = rs0 - x > qs0 - y ? rs0 - x : qs0 - y;
                rs1 = rs0 - l; qs1 = qs0 - l;
                if (rs1 < 0) rs1 = 0;
                break;
            }
        }
    }
    if (qs > 0 && rs > 0) {
        l = qs < o->max_gap ? qs : o->max_gap;
        qs1 = qs1 > qs - l ? qs1 : qs - l;
        qs0 = qs0 < qs1 ? qs0 : qs1;
        l += l * o->a > o->q ? (l * o->a - o->q) / o->e : 0;
        l = l < o->max_gap ? l : o->max_gap;
        l = l < rs ? l : rs;
        rs1 = rs1 > rs - l ? rs1 : rs - l;
        rs0 = rs0 < rs1 ? rs0 : rs1;
        rs0 = rs0 < rs ? rs0 : rs;
    } else { rs0 = rs; qs0 = qs; }
    re0 = (int32_t)a[r->as + r->cnt - 1].x + 1;
    qe0 = (int32_t)a[r->as + r->cnt - 1].y + 1;
    re1 = clen; qe1 = qlen;
    for (i = r->as + r->cnt, l = 0; i < n_a && a[i].x >> 32 == a[r->as].x >> 32; ++i) {
        const int32_t x = (int32_t)a[i].x + 1;
        const int32_t y = (int32_t)a[i].y + 1;
        if (x > re0 && y > qe0) {
            if (++l > o->min_cnt) {
                l = x - re0 > y - qe0 ? x - re0 : y - qe0;
                re1 = re0 + l; qe1 = qe0 + l;
                break;
            }
        }
    }
    if (qe < qlen && re < clen) {
        l = qlen - qe < o->max_gap ? qlen - qe : o->max_gap;
        qe1 = qe1 < qe + l ? qe1 : qe + l;
        qe0 = qe0 > qe1 ? qe0 : qe1;
        l += l * o->a > o->q ? (l * o->a - o->q) / o->e : 0;
        l = l < o->max_gap ? l : o->max_gap;
        l = l < clen - re ? l : clen - re;
        re1 = re1 < re + l ? re1 : re + l;
        re0 = re0 > re1 ? re0 : re1;
    } else { re0 = re; qe0 = qe; }

    tseq = (uint8_t *)malloc((size_t)(re0 - rs0 > 0 ? re0 - rs0 : 0) + 16);

    if (qs > 0 && rs > 0) {       /* left extension */
        qseq = &c->qseq0[rev][qs0];
        getseq(c, rid, rs0, rs, tseq);
        seq_rev(qs - qs0, qseq);
        seq_rev(rs - rs0, tseq);
        align_pair(o, qs - qs0, qseq, rs - rs0, tseq, mat, bw, o->end_bonus, r->split_inv ? o->zdrop_inv : o->zdrop,
                   MMA_EZ_EXTZ_ONLY | MMA_EZ_RIGHT | MMA_EZ_REV_CIGAR, ez);
        if (ez->n_cigar > 0) { append_cigar(r, ez->n_cigar, ez->cigar); r->p->dp_score += (int32_t)ez->max; }
        rs1 = rs - (ez->reach_end ? ez->mqe_t + 1 : ez->max_t + 1);
        qs1 = qs - (ez->reach_end ? qs - qs0 : ez->max_q + 1);
        seq_rev(qs - qs0, qseq);
    } else { rs1 = rs; qs1 = qs; }
    re1 = rs; qe1 = qs;

    for (i = 1; i < cnt1; ++i) {       /* gap filling */
        if ((a[as1 + i].y & (SEED_IGNORE | SEED_TANDEM)) && i != cnt1 - 1) continue;
        re = (int32_t)a[as1 + i].x - hk; qe = (int32_t)a[as1 + i].y - hk;
        re1 = re; qe1 = qe;
        if (i == cnt1 - 1 || (a[as1 + i].y & SEED_LONG_JOIN) || (qe - qs >= o->min_ksw_len && re - rs >= o->min_ksw_len)) {
            int j, bw1 = bw_long, zdrop_code;
            if (a[as1 + i].y & SEED_LONG_JOIN) bw1 = qe - qs > re - rs ? qe - qs : re - rs;
            qseq = &c->qseq0[rev][qs];
            getseq(c, rid, rs, re, tseq);
            align_pair(o, qe - qs, qseq, re - rs, tseq, mat, bw1, -1, o->zdrop, MMA_EZ_APPROX_MAX, ez);      /* first pass: approximate maximum */
            if ((zdrop_code = test_zdrop_lr(o, qseq, tseq, ez->n_cigar, ez->cigar, mat)) != 0)
                align_pair(o, qe - qs, qseq, re - rs, tseq, mat, bw1, -1, zdrop_code == 2 ? o->zdrop_inv : o->zdrop, 0, ez);
            if (ez->n_cigar > 0) append_cigar(r, ez->n_cigar, ez->cigar);
            if (ez->zdropped) {
                if (!r->p) r->p = (extra_t *)calloc(1, sizeof(extra_t));
                for (j = i - 1; j >= 0; --j) if ((int32_t)a[as1 + j].x <= rs + ez->max_t) break;
                dropped = 1;
                if (j < 0) j = 0;
                r->p->dp_score += (int32_t)ez->max;
                re1 = rs + (ez->max_t + 1);
                qe1 = qs + (ez->max_q + 1);
                if (cnt1 - (j + 1) >= o->min_cnt) {
                    split_reg(r, r2, as1 + j + 1 - r->as, qlen, a);
                    if (zdrop_code == 2) r2->split_inv = 1;
                }
                break;
            } else r->p->dp_score += ez->score;
            rs = re; qs = qe;
        }
    }

    if (!dropped && qe < qe0 && re < re0) {   /* right extension */
        qseq = &c->qseq0[rev][qe];
        getseq(c, rid, re, re0, tseq);
        align_pair(o, qe0 - qe, qseq, re0 - re, tseq, mat, bw, o->end_bonus, o->zdrop, MMA_EZ_EXTZ_ONLY, ez);
        if (ez->n_cigar > 0) { append_cigar(r, ez->n_cigar, ez->cigar); r->p->dp_score += (int32_t)ez->max; }
        re1 = re + (ez->reach_end ? ez->mqe_t + 1 : ez->max_t + 1);
        qe1 = qe + (ez->reach_end ? qe0 - qe : ez->max_q + 1);
    }

    r->rs = rs1; r->re = re1;
    if (rev) { r->qs = qlen - qe1; r->qe = qlen - qs1; }
    else { r->qs = qs1; r->qe = qe1; }
    if (r->p) {
        free(tseq);
        tseq = (uint8_t *)malloc((size_t)(re1 - rs1 > 0 ? re1 - rs1 : 0) + 16);
        getseq(c, rid, rs1, re1, tseq);
        update_extra(r, &c->qseq0[r->rev][qs1], tseq, mat, (int8_t)o->q, (int8_t)o->e, 1);
    }
    free(tseq);
}

/* mm_align1_inv: between the two halves of a region split by the inversion z-drop, align the reverse complement */
static int align1_inv(actx_t *c, const reg_t *r1, const reg_t *r2, reg_t *r_inv, mma_ez *ez)
{
    const mmo_opts *o = c->o;
    const int32_t qlen = c->qlen;
    int tl, ql, score, ret = 0, q_off, t_off;
    uint8_t *tseq, *qseq;
    int8_t mat[25];

    memset(r_inv, 0, sizeof(*r_inv));
    if (!(r1->split & 1) || !(r2->split & 2)) return 0;
    if (r1->id != r1->parent && r1->parent != PARENT_TMP_PRI) return 0;
    if (r2->id != r2->parent && r2->parent != PARENT_TMP_PRI) return 0;
    if (r1->rid != r2->rid || r1->rev != r2->rev) return 0;
    ql = r1->rev ? r1->qs - r2->qe : r2->qs - r1->qe;
    tl = r2->rs - r1->re;
    if (ql < o->min_chain_score || ql > o->max_gap) return 0;
    if (tl < o->min_chain_score || tl > o->max_gap) return 0;

    mma_gen_simple_mat(5, mat, (int8_t)o->a, (int8_t)o->b, (int8_t)o->sc_ambi);
    tseq = (uint8_t *)malloc((size_t)tl + 16);
    getseq(c, r1->rid, r1->re, r2->rs, tseq);
    qseq = r1->rev ? &c->qseq0[0][r2->qe] : &c->qseq0[1][qlen - r2->qs];

    seq_rev(ql, qseq);
    seq_rev(tl, tseq);
    score = mma_ksw_ll(ql, qseq, tl, tseq, mat, o->q, o->e, &q_off, &t_off);
    seq_rev(ql, qseq);
    seq_rev(tl, tseq);
    if (score < o->min_dp_max) goto end_align1_inv;
    q_off = ql - (q_off + 1); t_off = tl - (t_off + 1);
    align_pair(o, ql - q_off, qseq + q_off, tl - t_off, tseq + t_off, mat, (int)(o->bw * 1.5), -1, o->zdrop, MMA_EZ_EXTZ_ONLY, ez);
    if (ez->n_cigar == 0) goto end_align1_inv;
    append_cigar(r_inv, ez->n_cigar, ez->cigar);
    r_inv->p->dp_score = (int32_t)ez->max;
    r_inv->id = -1;
    r_inv->parent = PARENT_UNSET;
    r_inv->inv = 1;
    r_inv->rev = !r1->rev;
    r_inv->rid = r1->rid;
    r_inv->div = -1.0f;
    if (r_inv->rev == 0) {
        r_inv->qs = r2->qe + q_off;
        r_inv->qe = r_inv->qs + ez->max_q + 1;
    } else {
        r_inv->qe = r2->qs - q_off;
        r_inv->qs = r_inv->qe - (ez->max_q + 1);
    }
    r_inv->rs = r1->re + t_off;
    r_inv->re = r_inv->rs + ez->max_t + 1;
    update_extra(r_inv, &qseq[q_off], &tseq[t_off], mat, (int8_t)o->q, (int8_t)o->e, !o->is_sr);
    ret = 1;
end_align1_inv:
    free(tseq);
    return ret;
}

static void filter_regs(const mmo_opts *o, int qlen, int *n_regs, reg_t *regs)
{
    int i, k;
    for (i = k = 0; i < *n_regs; ++i) {
        reg_t *r = &regs[i];
        int flt = 0;
        if (!r->inv && !r->seg_split && r->cnt < o->min_cnt) flt = 1;
        if (r->p) {
            if (r->mlen < o->min_chain_score) flt = 1;
            else if (r->p->dp_max < o->min_dp_max) flt = 1;
            else if (r->qs > qlen * o->max_clip_ratio && qlen - r->qe > qlen * o->max_clip_ratio) flt = 1;
            if (flt) { free(r->p->cigar); free(r->p); r->p = 0; }
        }
        if (flt) continue;
        if (k < i) regs[k++] = regs[i];
        else ++k;
    }
    *n_regs = k;
}

static const uint8_t nt4[256] = {
#define R4 4, 4, 4, 4
#define R16 R4, R4, R4, R4
    R16, R16, R16, R16,
    4, 0, 4, 1, 4, 4, 4, 2, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 3, 3, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,
    4, 0, 4, 1, 4, 4, 4, 2, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 3, 3, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,
    R16, R16, R16, R16, R16, R16, R16, R16
#undef R16
#undef R4
};

void mma_align_read(const mmo_opts *o, const uint8_t *ref_packed, const uint64_t *contig_start, uint32_t n_contigs,
                    const uint8_t *seq, int32_t qlen, int32_t n_u, const uint64_t *u, mma_anchor *a,
                    int32_t n_mini_pos, const uint64_t *mini_pos, mma_result *res)
{
    actx_t c;
    reg_t *regs;
    int n_regs = n_u, i;
    uint32_t hash, sig = 2166136261u;
    mma_ez ez;

    memset(res, 0, sizeof(*res));
    if (n_u <= 0) return;
    /* mm_map_frag: hash of the (absent) query name, the query length and opt->seed = 11 */
    hash = 0;
    hash ^= wang_hash((uint32_t)qlen) + wang_hash(11u);
    hash = wang_hash(hash);
    regs = gen_regs(hash, qlen, n_u, u, a);
    /* chain_post */
    set_parent(o->mask_level, INT_MAX, n_regs, regs);
    select_sub(o->pri_ratio, o->k * 2, o->best_n, 1, (int)(o->max_gap * 0.8), &n_regs, regs);
    c.o = o; c.ref = ref_packed; c.cstart = contig_start; c.n_contigs = n_contigs; c.qlen = qlen; c.a = a; c.n_a = 0;
    if (!o->is_sr) {      /* mm_map_frag: !is_sr && !MM_F_QSTRAND */
        est_err(&c, n_regs, regs, n_mini_pos, mini_pos);
        n_regs = filter_strand_retained(n_regs, regs);
    }
    res->n_aligned = n_regs;

    /* mm_align_skeleton */
    c.qseq0[0] = (uint8_t *)malloc((size_t)qlen * 2 + 16);
    c.qseq0[1] = c.qseq0[0] + qlen;
    for (i = 0; i < qlen; ++i) {
        c.qseq0[0][i] = nt4[seq[i]];
        c.qseq0[1][qlen - 1 - i] = c.qseq0[0][i] < 4 ? 3 - c.qseq0[0][i] : 4;
    }
    memset(&ez, 0, sizeof(ez));
    c.n_a = squeeze_a(n_regs, regs, a);
    for (i = 0; i < n_regs; ++i) {
        reg_t r2;
        if (o->is_sr) align1_sr(&c, &regs[i], &r2, &ez);
        else align1_lr(&c, &regs[i], &r2, &ez);
        if (r2.cnt > 0) {       /* mm_insert_reg */
            regs = (reg_t *)realloc(regs, (size_t)(n_regs + 1) * sizeof(reg_t));
            if (i + 1 != n_regs) memmove(&regs[i + 2], &regs[i + 1], sizeof(reg_t) * (size_t)(n_regs - i - 1));
            regs[i + 1] = r2;
            ++n_regs;
        }
        if (i > 0 && regs[i].split_inv) {
            if (align1_inv(&c, &regs[i - 1], &regs[i], &r2, &ez)) {
                regs = (reg_t *)realloc(regs, (size_t)(n_regs + 1) * sizeof(reg_t));
                if (i + 1 != n_regs) memmove(&regs[i + 2], &regs[i + 1], sizeof(reg_t) * (size_t)(n_regs - i - 1));
                regs[i + 1] = r2;
                ++n_regs;
                ++i;
            }
        }
    }
    free(c.qseq0[0]); free(ez.cigar);
    filter_regs(o, qlen, &n_regs, regs);
    res->n_regs = n_regs;
    for (i = 0; i < n_regs; ++i) {
        const reg_t *r = &regs[i];
        const int32_t v[8] = { r->rs, r->re, r->qs, r->qe, r->mlen, r->blen, r->p ? r->p->dp_max : -1, r->cnt };
        int j;
        for (j = 0; j < 8; ++j) { sig ^= (uint32_t)v[j]; sig *= 16777619u; }
        if (r->p && r->p->dp_max > res->dp_max) res->dp_max = r->p->dp_max;
    }
    res->sig = n_regs > 0 ? sig : 0;
    for (i = 0; i < n_regs; ++i) if (regs[i].p) { free(regs[i].p->cigar); free(regs[i].p); }
    free(regs);
}
