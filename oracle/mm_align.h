/*
 * mm_align.h — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE): App. A.6, the base-level extension stage that
 * `.with_cigar()` (/root/reference/src/cleaner.rs:473) switches on inside `aligner.map()` (:552) and that decides
 * `mappings.len() > 0` (:553) after chaining.   *** PARITY UNPINNED *** (see mm_oracle.h): restated from the published
 * algorithm of lh3/minimap2 ~v2.28 (hit.c, align.c, ksw2_extd2_sse.c, ksw2.h), which is not on this box.
 */
#ifndef MM_ALIGN_H
#define MM_ALIGN_H
#include "mm_oracle.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint64_t x, y; } mma_anchor;     /* x = strand<<63 | rid<<32 | rpos, y = flags (MM_SEED_TANDEM = 1<<42, ...) | q_span<<32 | qpos */

typedef struct {
    uint32_t max; int zdropped;                   /* ksw_extz_t (max is a 31-bit unsigned field upstream) */
    int max_q, max_t, mqe, mqe_t, mte, mte_q, score, reach_end;
    int n_cigar, m_cigar; uint32_t *cigar;
} mma_ez;

#define MMA_EZ_SCORE_ONLY 0x01
#define MMA_EZ_RIGHT      0x02
#define MMA_EZ_GENERIC_SC 0x04
#define MMA_EZ_APPROX_MAX 0x08
#define MMA_EZ_APPROX_DROP 0x10
#define MMA_EZ_EXTZ_ONLY  0x40
#define MMA_EZ_REV_CIGAR  0x80

/* ksw_extd2_sse restated as a scalar emulation of its 16-lane int8 difference recurrence, memory layout included
 * (the rounded [st, en] ranges compute and store cells outside the band; band-limited alignments read them back) */
void mma_ksw_extd2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                   int8_t q, int8_t e, int8_t q2, int8_t e2, int w, int zdrop, int end_bonus, int flag, mma_ez *ez);
void mma_gen_simple_mat(int m, int8_t *mat, int8_t a, int8_t b, int8_t sc_ambi);

/* result of the stage for one read */
typedef struct {
    int32_t n_aligned;       /* regions entering mm_align_skeleton (after mm_set_parent / mm_select_sub) */
    int32_t n_regs;          /* regions left by mm_filter_regs: mappings.len() */
    int32_t dp_max;          /* largest dp_max among them (0 if none) */
    uint32_t sig;            /* fingerprint of (rs, re, qs, qe, mlen, blen, dp_max, cnt) of the survivors, in order */
} mma_result;

/* n_u chains: u[i] = score<<32 | cnt, anchors a[] already compacted in chain order (compact_a).  ref: 4-bit packed nt4
 * codes of all contigs, contig_start[n_contigs + 1]. */
void mma_align_read(const mmo_opts *o, const uint8_t *ref_packed, const uint64_t *contig_start, uint32_t n_contigs,
                    const uint8_t *seq, int32_t qlen, int32_t n_u, const uint64_t *u, mma_anchor *a,
                    int32_t n_mini_pos, const uint64_t *mini_pos /* mm_collect_matches' list, for mm_est_err */, mma_result *res);

/* ksw_ll_i16 (local alignment score, end of the best hit as upstream's striped scan reports it); used by the inversion tests */
int mma_ksw_ll(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat, int gapo, int gape, int *qe, int *te);

/* mg_lchain_rmq's scoring pass (mm_rmq.c): a[] sorted by x; f / p out; t zeroed by the caller */
void mmo_lchain_rmq_fill(int max_dist, int max_dist_inner, int bw, int max_chn_skip, int cap_rmq_size,
                         float chn_pen_gap, float chn_pen_skip, int64_t n, const mma_anchor *a, int32_t *f, int64_t *p, int32_t *t);

#ifdef __cplusplus
}
#endif
#endif
