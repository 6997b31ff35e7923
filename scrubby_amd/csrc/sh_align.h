// sh_align.h — the base-level extension stage on the device.
//
// `.with_cigar()` at /root/reference/src/cleaner.rs:473 makes `aligner.map()` (:552) align every chain it keeps and drop
// the regions that fail minimap2's mm_filter_regs, so `mappings.len() > 0` (:553) is "a region survives", not "a chain
// exists".  This header restates, for the short-read mode (MM_F_SR, preset sr), what happens between the chains and that
// count, statement for statement with oracle/mm_align.c (which cites the upstream functions):
//     mm_gen_regs -> mm_set_parent -> mm_select_sub -> per region mm_align1 (mm_max_stretch, ungapped middle,
//     ksw_extd2 end extensions, mm_test_zdrop + second pass, mm_split_reg) -> mm_update_extra (mm_fix_cigar) -> mm_filter_regs
// One WAVE per read.  The control flow of a region is scalar work (every lane runs it uniformly; lane 0 owns the stores);
// the wave cooperates where there is width: region keys and their sort, staging of the query / reference window, and the
// anti-diagonals of ksw_extd2, where the 64 lanes are the 16-lane SSE vectors of the original four at a time.  The dual
// affine recurrence runs on int8 differences exactly as upstream does (wrap-around included), over the same rounded
// [st, en] ranges, because band-limited alignments read back cells those ranges computed outside the band.
//
// No MFMA: the recurrence is a max-plus wavefront over an anti-diagonal with a dependency on the two previous ones -
// there is no contraction dimension to feed a matrix core.
#pragma once
#include "sh_common.h"
#include "sh_wave.h"

#define KSW_NEG_INF (-0x40000000)
#define EZ_RIGHT 0x02
#define EZ_APPROX_MAX 0x08
#define EZ_APPROX_DROP 0x10
#define EZ_EXTZ_ONLY 0x40
#define EZ_REV_CIGAR 0x80

struct AlignParams {
    int32_t k, min_cnt, min_sc, max_gap, bw, bw_long;
    int32_t a, b, q, e, q2, e2, sc_ambi, zdrop, zdrop_inv, end_bonus, min_dp_max, best_n;
    float pri_ratio, mask_level, max_clip_ratio;
    int32_t lemma, unc_max;      // flag-only: ChainParams::ext_lemma / ext_unc_max
};

// a chain as the chaining kernels hand it over: anchors cx/cq[off .. off + cnt) in ascending order
struct ChainRec { uint32_t next, cnt; int32_t score; uint32_t key_f, key_i, yfl /* y flags of the first anchor >> 32 */; unsigned long long off, z; };
#define SINK_SHARDS 64
// Which seeds of a read are "tandem" (mm_seed_collect_all): their anchors carry MM_SEED_TANDEM in y, which mm_gen_regs hashes into the
// order of equal-score chains and which the long-read gap filling skips.  Looked up by query position in the read's seed records.
#define SH_Y_TANDEM (1ull << 42)
struct TandemQ { const uint4 *rec; uint32_t n, stride; int32_t qlen; uint32_t any; };
__device__ inline uint64_t tandem_yflag(const TandemQ &t, uint64_t x0, uint32_t q0, int32_t k)
{
    if (!t.any) return 0;
    const uint32_t qpos = (x0 >> 63) ? (uint32_t)(t.qlen + k - 2) - q0 : q0;      // back to the forward strand of the query
    for (uint32_t j = 0; j < t.n; ++j) {
        const uint4 s = t.rec[(size_t)j * t.stride];
        if ((s.w >> 1) == qpos) {
            bool td = (s.z & SH_REC_PREV_SAME) != 0;
            if (!td && j + 1 < t.n) td = (t.rec[(size_t)(j + 1) * t.stride].z & SH_REC_PREV_SAME) != 0;
            return td ? SH_Y_TANDEM : 0ull;
        }
    }
    return 0;
}

struct ChainSink {
    // seed records of the batch in K1's layout (read r: trec + r * tseed_cap, count and "has a tandem seed" in tinfo[r]): what sink_emit
    // looks the first anchor's MM_SEED_TANDEM up in; kernels that sketch reads themselves pass their own TandemQ
    const uint4 *trec; const uint32_t *tinfo; uint32_t tseed_cap;
    ChainRec *recs; uint32_t *n_recs; uint32_t cap_recs;                 // 64 shards of cap_recs records (one cursor each: a single address
    uint64_t *cx; uint32_t *cq; unsigned long long *n_anch; unsigned long long cap_anch;   // only sustains ~90 M atomics/s), likewise cap_anch anchors
    uint32_t *head;        // per read of the chunk: newest record, ~0u = none
    uint32_t *overflow;
    // flag-only calls: only a chain that can still be regs[0] of mm_gen_regs - the largest z = (score << 32 | cnt) ^ h of its read - is
    // worth handing over; best[read] is the largest z seen so far (atomicMax), tie[read] is set when two chains share it.
    // nullptr: every chain is kept (trace mode, and the second pass over reads whose top chain did not settle them).
    unsigned long long *best; uint32_t *tie;
};

__device__ inline uint64_t al_hash64(uint64_t key)
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
__device__ inline uint32_t al_wang(uint32_t key)
{
    key += ~(key << 15); key ^= (key >> 10); key += (key << 3); key ^= (key >> 6); key += ~(key << 11); key ^= (key >> 16);
    return key;
}
// mm_map_frag's hash of the (absent) query name, the query length and opt->seed = 11
__device__ inline uint32_t region_hash(int32_t qlen)
{
    uint32_t h = 0;
    h ^= al_wang((uint32_t)qlen) + al_wang(11u);
    return al_wang(h);
}
// mm_gen_regs' sort key of a chain whose first anchor is (x0, q0)
__device__ inline unsigned long long chain_z(uint64_t x0, uint32_t q0, int32_t k, int32_t score, uint32_t cnt, uint32_t rhash, uint64_t yfl)
{
    const uint64_t y0 = yfl | (uint64_t)(uint32_t)k << 32 | q0;
    const uint32_t h = (uint32_t)al_hash64((al_hash64(x0) + al_hash64(y0)) ^ rhash);
    return ((unsigned long long)(uint32_t)score << 32 | cnt) ^ h;
}

// flag-only: the score of the best chain of `read` handed over so far (0: none).  z's high word IS the score (h only touches the low
// word), so a chain, a cluster or the rest of a backtrack whose score cannot reach it can be skipped outright.
__device__ inline int32_t sink_best_score(const ChainSink &sk, uint32_t read)
{
    return sk.best ? (int32_t)(__hip_atomic_load(&sk.best[read], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32) : 0;
}

// append one chain.  pred(i) -> predecessor; xq(i, x, q) reads an anchor.  Called by ONE lane.
template <class XQ, class PRED>
__device__ inline void sink_emit(const ChainSink &sk, uint32_t read, int32_t zi, int32_t end_i, int32_t score, uint32_t cnt,
                                 uint32_t key_f, uint32_t key_i, int32_t k, uint32_t rhash, int32_t qlen, XQ xq, PRED pred, const TandemQ *tq = nullptr)
{
    if (score < sink_best_score(sk, read)) return;
    int32_t first = zi;
    for (int32_t p = pred(first); p != end_i; p = pred(first)) first = p;
    uint64_t x0; uint32_t q0;
    xq(first, x0, q0);
    uint64_t yfl = 0;
    if (tq) yfl = tandem_yflag(*tq, x0, q0, k);
    else if (sk.trec) {
        const uint32_t info = sk.tinfo[read], n = info >> 16 & 0x7fffu;
        const TandemQ t{sk.trec + (size_t)read * sk.tseed_cap, n < sk.tseed_cap ? n : sk.tseed_cap, 1u, qlen, info >> 31};
        yfl = tandem_yflag(t, x0, q0, k);
    }
    const unsigned long long z = chain_z(x0, q0, k, score, cnt, rhash, yfl);
    if (sk.best) {
        const unsigned long long old = atomicMax(&sk.best[read], z);
        if (z < old) return;
        if (z == old) { atomicExch(&sk.tie[read], 1u); return; }
    }
    const uint32_t sh = (blockIdx.x * 7u + (threadIdx.x >> 6) * 3u + (uint32_t)(zi & 7)) & (SINK_SHARDS - 1);
    const uint32_t li = atomicAdd(&sk.n_recs[sh], 1u);
    const unsigned long long lo = atomicAdd(&sk.n_anch[sh], (unsigned long long)cnt);
    if (li >= sk.cap_recs || lo + cnt > sk.cap_anch) { atomicExch(sk.overflow, 1u); return; }
    const uint32_t idx = sh * sk.cap_recs + li;
    const unsigned long long off = (unsigned long long)sh * sk.cap_anch + lo;
    uint32_t j = cnt;
    for (int32_t i = zi; i != end_i && j > 0; i = pred(i)) { --j; uint64_t x; uint32_t q; xq(i, x, q); sk.cx[off + j] = x; sk.cq[off + j] = q; }
    ChainRec rc{0u, cnt, score, key_f, key_i, (uint32_t)(yfl >> 32), off, z};
    rc.next = atomicExch(&sk.head[read], idx);
    sk.recs[idx] = rc;
}

// ------------------------------------------------------------------------------------------------
// memory of one wave
// ------------------------------------------------------------------------------------------------
// Loads of HBM scratch bytes that ANOTHER lane of this wave stored: past the L1 (relaxed agent-scope loads), after an
// al_sync().  LDS needs neither.  `g` = the pointer is HBM scratch.
__device__ inline uint32_t cc_u32(const void *p) { return __hip_atomic_load((const uint32_t *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline uint64_t cc_u64(const void *p) { return __hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline uint8_t cc_u8(const void *p)
{
    const uint32_t w = cc_u32((const void *)((uintptr_t)p & ~(uintptr_t)3));
    return (uint8_t)(w >> (8 * ((uintptr_t)p & 3)));
}
__device__ inline uint8_t ld8(const uint8_t *p, bool g) { return g ? cc_u8(p) : *p; }
__device__ inline int8_t ld8s(const int8_t *p, bool g) { return g ? (int8_t)cc_u8(p) : *p; }
__device__ inline int32_t ld32(const int32_t *p, bool g) { return g ? (int32_t)cc_u32(p) : *p; }
__device__ inline uint64_t ld64(const uint64_t *p, bool g) { return g ? cc_u64(p) : *p; }

__device__ inline void al_sync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier(); }
// the same for data the wave shares through LDS only: outstanding LDS operations are waited for, stores on their way to HBM (direction bytes,
// which nothing reads before the backtrack) are not - the full fence made every anti-diagonal wait for them
#define AL_WIDE_MIN 128     // anti-diagonals wider than this take four cells per lane; narrower ones are bound by the arithmetic, not by LDS issue, and keep one
__device__ inline int32_t al_wave_shr1(int32_t v, int32_t fill) { return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xf, 0xf, false); }   // lane i takes lane i-1's value
__device__ inline void al_sync_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
__device__ inline uint32_t al_lane() { return threadIdx.x & 63; }
__device__ inline int32_t al_b0(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ inline uint64_t al_b0_64(uint64_t v) { return (uint64_t)(uint32_t)al_b0((int32_t)(uint32_t)v) | (uint64_t)(uint32_t)al_b0((int32_t)(uint32_t)(v >> 32)) << 32; }

struct RegLite {            // mm_reg1_t, the fields the decision reads
    int32_t cnt, score, qs, qe, rs, re, parent, rid, rev, pad;
    unsigned long long as;  // anchors at cx/cq[as .. as + cnt)
};

#define AL_Q 320            // reads up to this length keep their codes and reference window in LDS
#define AL_T (2 * AL_Q + 96)
#define AL_T16 448          // ksw buffers in LDS up to this many target bases (rounded to 16) ...
#define AL_Q16 336          // ... and this many query bases
#define AL_P 4096            // ... and this many direction bytes (measured: 24576 -> 72 ms for the bench's 66 k alignments at 3 waves/CU, 4096 -> 33 ms at 8: occupancy beats LDS residency)
#define AL_R 64             // chains of a read kept in LDS
#define AL_PRI 256          // primaries mm_set_parent may find (a 150-bp read has a handful)

// NR chains of a read and NPRI primaries fit the LDS copy (more chains: the region arrays move to the wave's HBM scratch).  The kernel that
// aligns regs[0] alone (k_regs_align_top) holds one chain and the few regions z-drops split off it: 12 KB instead of 19, 12 waves per CU
// instead of 8 - the stage is bound by the latency of one wave's anti-diagonals, so residency is throughput.
template <int NR_, int NPRI_>
struct AlignLdsT {
    static constexpr int NR = NR_, NPRI = NPRI_;
    __attribute__((aligned(16))) uint8_t kmem[8 * AL_T16 + AL_Q16 + 32];
    int32_t kH[AL_T16];
    __attribute__((aligned(16))) uint8_t kp[AL_P];
    uint8_t qseq[2 * AL_Q];
    uint8_t tseq[AL_T];
    RegLite regs[NR_];
    uint64_t kz[NR_], kx[NR_], kk[NR_];
    uint32_t kh[NR_], ord[NR_];
    int32_t pri[NPRI_];
    uint64_t cov[NPRI_];
};
typedef AlignLdsT<AL_R, AL_PRI> AlignLds;
typedef AlignLdsT<16, 16> AlignLdsTop;

struct AlignScratch {       // one per wave, in HBM; sized by the host from max_read_len (align_scratch_bytes)
    uint8_t *qseq, *tseq, *kmem, *kp;
    int32_t *kH, *koff;
    uint32_t *ez_cigar, *r_cigar;
    RegLite *regs;
    uint64_t *kz, *kx, *kk;
    uint32_t *kh, *ord;
    int32_t *pri; uint64_t *cov;      // mm_set_parent's primaries and coverage intervals when the regions live here (more than AL_R chains)
    uint32_t qcap, tcap, reg_cap; unsigned long long pcap;
};

__host__ __device__ inline unsigned long long align_scratch_layout(uint32_t max_read_len, uint32_t reg_cap, uint32_t *qcap_o, uint32_t *tcap_o, unsigned long long *pcap_o)
{
    const unsigned long long L = max_read_len < 32 ? 32 : max_read_len;
    const unsigned long long qcap = (L + 31) & ~15ull, tcap = (2 * L + 96 + 15) & ~15ull;
    const unsigned long long ncol = (((qcap < 512 ? qcap : 512) + 15) / 16 + 1) * 16;      // n_col <= (min(qlen, tlen, w + 1) + 15) / 16 * 16 + 16, w <= 151 + slack
    const unsigned long long pcap = (qcap + tcap) * ncol;
    if (qcap_o) *qcap_o = (uint32_t)qcap;
    if (tcap_o) *tcap_o = (uint32_t)tcap;
    if (pcap_o) *pcap_o = pcap;
    unsigned long long b = 0;
    auto add = [&](unsigned long long bytes) { b += (bytes + 15) & ~15ull; };
    add((unsigned long long)reg_cap * sizeof(RegLite)); add(8ull * reg_cap); add(8ull * reg_cap); add(8ull * reg_cap); add(4ull * reg_cap); add(4ull * reg_cap);
    add(4ull * reg_cap); add(8ull * reg_cap);
    add(4 * tcap); add(8 * (qcap + tcap)); add(4 * (qcap + tcap + 8)); add(4 * (2 * (qcap + tcap) + 16));
    add(2 * qcap); add(tcap); add(8 * tcap + qcap + 32); add(pcap);
    return (b + 255) & ~255ull;
}

__device__ inline void align_scratch_carve(AlignScratch &A, uint8_t *p, uint32_t max_read_len, uint32_t reg_cap)
{
    unsigned long long pcap;
    align_scratch_layout(max_read_len, reg_cap, &A.qcap, &A.tcap, &pcap);
    A.pcap = pcap; A.reg_cap = reg_cap;
    auto take = [&](unsigned long long bytes) { uint8_t *q = p; p += (bytes + 15) & ~15ull; return q; };
    A.regs = (RegLite *)take((unsigned long long)reg_cap * sizeof(RegLite));
    A.kz = (uint64_t *)take(8ull * reg_cap); A.kx = (uint64_t *)take(8ull * reg_cap); A.kk = (uint64_t *)take(8ull * reg_cap);
    A.kh = (uint32_t *)take(4ull * reg_cap); A.ord = (uint32_t *)take(4ull * reg_cap);
    A.pri = (int32_t *)take(4ull * reg_cap); A.cov = (uint64_t *)take(8ull * reg_cap);
    A.kH = (int32_t *)take(4ull * A.tcap);
    A.koff = (int32_t *)take(8ull * (A.qcap + A.tcap));
    A.ez_cigar = (uint32_t *)take(4ull * (A.qcap + A.tcap + 8));
    A.r_cigar = (uint32_t *)take(4ull * (2 * (A.qcap + A.tcap) + 16));
    A.qseq = take(2ull * A.qcap);
    A.tseq = take(A.tcap);
    A.kmem = take(8ull * A.tcap + A.qcap + 32);
    A.kp = take(pcap);
}

struct Ez { int32_t max, zdropped, max_q, max_t, mqe, mqe_t, mte, mte_q, score, reach_end, n_cigar; };

__device__ inline void ez_reset(Ez &ez)
{
    ez.max_q = ez.max_t = ez.mqe_t = ez.mte_q = -1;
    ez.max = 0; ez.score = ez.mqe = ez.mte = KSW_NEG_INF;
    ez.n_cigar = 0; ez.zdropped = 0; ez.reach_end = 0;
}

// ksw_backtrack (rotated).  Every lane runs the walk on the same data (uniform loads); the cigar array is written by lane 0
// alone, which is also its only later reader; the merge test works on a register copy of the last word.
template <class RD>
__device__ inline void ksw_backtrack_dev(bool is_rev, RD rd, const int32_t *off, const int32_t *off_end, int32_t n_col, int32_t i0, int32_t j0,
                                         uint32_t *cigar, int32_t &n_cigar_out)
{
    int32_t n = 0, i = i0, j = j0, state = 0;
    uint32_t last = 0;
    auto push = [&](uint32_t op, int32_t len) {
        if (n == 0 || op != (last & 0xf)) { last = (uint32_t)len << 4 | op; ++n; }
        else last += (uint32_t)len << 4;
        if (al_lane() == 0) cigar[n - 1] = last;
    };
    while (i >= 0 && j >= 0) {
        int32_t force_state = -1;
        const int32_t r = i + j;
        const int32_t o = (int32_t)cc_u32(off + r), oe = (int32_t)cc_u32(off_end + r);
        if (i < o) force_state = 2;
        if (i > oe) force_state = 1;
        const uint32_t tmp = force_state < 0 ? rd((unsigned long long)r * (unsigned long long)n_col + (unsigned long long)(i - o)) : 0u;
        if (state == 0) state = tmp & 7;
        else if (!(tmp >> (state + 2) & 1)) state = 0;
        if (state == 0) state = tmp & 7;
        if (force_state >= 0) state = force_state;
        if (state == 0) { push(0, 1); --i; --j; }
        else if (state == 1 || state == 3) { push(2, 1); --i; }
        else { push(1, 1); --j; }
    }
    if (i >= 0) push(2, i + 1);
    if (j >= 0) push(1, j + 1);
    if (!is_rev && al_lane() == 0)
        for (int32_t a = 0; a < n >> 1; ++a) { const uint32_t t = cigar[a]; cigar[a] = cigar[n - 1 - a]; cigar[n - 1 - a] = t; }
    n_cigar_out = n;
}

// ksw_extd2_sse on one wave (oracle/mm_align.c mma_ksw_extd2 is the scalar statement of the same thing).
// query / target: codes 0..4, readable by every lane (qg / tg: they are HBM scratch).
// G = the buffers are the wave's HBM scratch; false = LDS (every pointer then derives from Ls alone, so the accesses compile to ds_*)
// G: the DP state (u v x y x2 y2 s | sf | qr, H) in the wave's HBM scratch instead of LDS; GP: the direction bytes there.  Three forms: all in
// LDS (short extensions), state in LDS + directions in HBM (written once, read by the backtrack only), all in HBM (beyond AL_T16 / AL_Q16).
template <bool G, bool GP, class LDS>
__device__ inline void ksw_extd2_core(int32_t qlen, const uint8_t *query, bool qg, int32_t tlen, const uint8_t *target, bool tg,
                                      int8_t sc_mch, int8_t sc_mis, int8_t sc_N, int32_t q, int32_t e, int32_t q2, int32_t e2, int32_t w,
                                      int32_t zdrop, int32_t end_bonus, int32_t flag, Ez &ez, uint32_t *cigar, AlignScratch &A, LDS &Ls)
{
    const uint32_t lane = al_lane();
    auto dsync = [&]() { if constexpr (G) al_sync(); else al_sync_lds(); };      // the DP state lives in LDS (!G): no anti-diagonal waits for HBM
    ez_reset(ez);
    if (qlen <= 0 || tlen <= 0) return;
    if (q2 + e2 < q + e) { int32_t t = q; q = q2; q2 = t; t = e; e = e2; e2 = t; }
    const int32_t qe = q + e, qe2 = q2 + e2;
    if (w < 0) w = tlen > qlen ? tlen : qlen;
    const int32_t tlen_ = (tlen + 15) / 16, qlen_ = (qlen + 15) / 16, T16 = tlen_ * 16;
    int32_t n_col_ = qlen < tlen ? qlen : tlen;
    n_col_ = ((n_col_ < w + 1 ? n_col_ : w + 1) + 15) / 16 + 1;
    const int32_t n_col = n_col_ * 16;
    {   // "otherwise, we won't see any mismatches"
        int32_t min_sc = sc_mis < sc_N ? sc_mis : sc_N;
        min_sc = min_sc < sc_mch ? min_sc : sc_mch;
        if (-min_sc > 2 * (q + e)) return;
    }
    int32_t long_thres = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;
    if (q2 + e2 + long_thres * e2 > q + e + long_thres * e) ++long_thres;
    const int32_t long_diff = long_thres * (e - e2) - (q2 - q) - e2;
    const unsigned long long p_need = (unsigned long long)(qlen + tlen - 1) * (unsigned long long)n_col;
    constexpr bool g = G, in_lds = !GP;
    if ((g || GP) && ((uint32_t)T16 > A.tcap || (uint32_t)(qlen_ * 16 + 16) > A.qcap + 32 || p_need > A.pcap || (uint32_t)(qlen + tlen) > A.qcap + A.tcap)) { ez.zdropped = 1; return; }   // outside the context's sizing (never for reads within max_read_len)
    uint8_t *mem = G ? A.kmem : Ls.kmem;
    int32_t *H = G ? A.kH : Ls.kH;
    uint8_t *p = GP ? A.kp : Ls.kp;
    int32_t *off = A.koff, *off_end = A.koff + (qlen + tlen);
    int8_t *u = (int8_t *)mem, *v = u + T16, *x = v + T16, *y = x + T16, *x2 = y + T16, *y2 = x2 + T16, *s = y2 + T16;
    uint8_t *sf = (uint8_t *)(s + T16), *qr = sf + T16;
    const int32_t qr_cap = qlen_ * 16 + 16;
    for (int32_t t = (int32_t)lane; t < T16; t += 64) {
        u[t] = v[t] = x[t] = y[t] = (int8_t)(-q - e);
        x2[t] = y2[t] = (int8_t)(-q2 - e2);
        s[t] = 0;
        sf[t] = t < tlen ? ld8(target + t, tg) : 0;
        H[t] = KSW_NEG_INF;
    }
    for (int32_t t = (int32_t)lane; t < qr_cap; t += 64) qr[t] = t < qlen ? ld8(query + (qlen - 1 - t), qg) : 0;
    dsync();

    int32_t last_st = -1, last_en = -1, r, H0 = 0, last_H0_t = 0;
    for (r = 0; r < qlen + tlen - 1; ++r) {
        int32_t st = 0, en = tlen - 1;
        if (st < r - qlen + 1) st = r - qlen + 1;
        if (en > r) en = r;
        if (st < (r - w + 1) >> 1) st = (r - w + 1) >> 1;
        if (en > (r + w) >> 1) en = (r + w) >> 1;
        if (st > en) { ez.zdropped = 1; break; }
        const int32_t st0 = st, en0 = en;
        st = st / 16 * 16; en = (en + 16) / 16 * 16 - 1;
        int8_t x1, x21, v1;
        if (st > 0) {
            if (st - 1 >= last_st && st - 1 <= last_en) { x1 = ld8s(x + st - 1, g); x21 = ld8s(x2 + st - 1, g); v1 = ld8s(v + st - 1, g); }
            else { x1 = (int8_t)(-q - e); x21 = (int8_t)(-q2 - e2); v1 = (int8_t)(-q - e); }
        } else {
            x1 = (int8_t)(-q - e); x21 = (int8_t)(-q2 - e2);
            v1 = r == 0 ? (int8_t)(-q - e) : r < long_thres ? (int8_t)(-e) : r == long_thres ? (int8_t)long_diff : (int8_t)(-e2);
        }
        dsync();      // the boundary reads above come before this diagonal's stores
        if (en >= r && lane == 0) {
            y[r] = (int8_t)(-q - e); y2[r] = (int8_t)(-q2 - e2);
            u[r] = r == 0 ? (int8_t)(-q - e) : r < long_thres ? (int8_t)(-e) : r == long_thres ? (int8_t)long_diff : (int8_t)(-e2);
        }
        // scores: upstream fills 16 at a time from st0; what would land beyond the array (t >= T16) falls into dead bytes of the
        // target copy (positions below st0, never read again) and is dropped here
        {
            const int32_t cnt = ((en0 - st0) / 16 + 1) * 16;
            for (int32_t o = (int32_t)lane; o < cnt; o += 64) {
                const int32_t t = st0 + o;
                if (t < T16) {
                    const int32_t qi = qlen - 1 - r + t;
                    const uint8_t sq = ld8(sf + t, g), sqr = (qi >= 0 && qi < qr_cap) ? ld8(qr + qi, g) : 0;
                    s[t] = (sq == 4 || sqr == 4) ? sc_N : (sq == sqr ? sc_mch : sc_mis);
                }
            }
        }
        if (lane == 0) { off[r] = st; off_end[r] = en; }
        dsync();
        if (en - st + 1 > AL_WIDE_MIN) {
            // core loop over the rounded range, FOUR cells per lane (256 a step: the state arrays move as dwords - the byte-wide form issued
            // thirteen LDS operations per 64 cells and was bound by them); every cell reads the previous anti-diagonal only.  [st, en] starts on a
            // multiple of 16 and holds a multiple of 16 cells, so a lane's four cells are all inside or all outside.
            uint32_t cwx = (uint32_t)(uint8_t)x1 << 24, cwv = (uint32_t)(uint8_t)v1 << 24, cwx2 = (uint32_t)(uint8_t)x21 << 24;      // byte 3 = the cell before the range
            for (int32_t tb = st; tb <= en; tb += 256) {
                const int32_t t = tb + 4 * (int32_t)lane;
                const bool on = t <= en;
                auto LDW = [&](const int8_t *a) -> uint32_t { if constexpr (G) return cc_u32((const uint32_t *)(a + t)); else return *(const uint32_t *)(a + t); };
                const uint32_t wx = on ? LDW(x) : 0u, wv = on ? LDW(v) : 0u, wx2 = on ? LDW(x2) : 0u;
                uint32_t px = (uint32_t)al_wave_shr1((int32_t)wx, 0), pv = (uint32_t)al_wave_shr1((int32_t)wv, 0), px2 = (uint32_t)al_wave_shr1((int32_t)wx2, 0);
                if (lane == 0) { px = cwx; pv = cwv; px2 = cwx2; }
                cwx = (uint32_t)__builtin_amdgcn_readlane((int)wx, 63); cwv = (uint32_t)__builtin_amdgcn_readlane((int)wv, 63); cwx2 = (uint32_t)__builtin_amdgcn_readlane((int)wx2, 63);
                if (on) {
                    const uint32_t ws = LDW(s), wu = LDW(u), wy = LDW(y), wy2 = LDW(y2);
                    // the cells' left neighbours: bytes (prev.3, own.0, own.1, own.2)
                    const uint32_t nx = wx << 8 | px >> 24, nv = wv << 8 | pv >> 24, nx2 = wx2 << 8 | px2 >> 24;
                    uint32_t ou = 0, ov = 0, ox = 0, oy = 0, ox2 = 0, oy2 = 0, od = 0;
    #pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int sh = 8 * c;
                        const int8_t xt1 = (int8_t)(nx >> sh), vt1 = (int8_t)(nv >> sh), x2t1 = (int8_t)(nx2 >> sh);
                        int8_t z = (int8_t)(ws >> sh);
                        const int8_t ut = (int8_t)(wu >> sh);
                        int8_t a = (int8_t)(xt1 + vt1), b = (int8_t)((int8_t)(wy >> sh) + ut), a2 = (int8_t)(x2t1 + vt1), b2 = (int8_t)((int8_t)(wy2 >> sh) + ut), tmp;
                        uint8_t d;
                        int8_t nxv, nyv, nx2v, ny2v;
                        if (!(flag & EZ_RIGHT)) {
                            d = a > z ? 1 : 0;  z = z > a ? z : a;
                            d = b > z ? 2 : d;  z = z > b ? z : b;
                            d = a2 > z ? 3 : d; z = z > a2 ? z : a2;
                            d = b2 > z ? 4 : d; z = z > b2 ? z : b2;
                            z = z < sc_mch ? z : sc_mch;
                            tmp = (int8_t)(z - q);  a = (int8_t)(a - tmp);  b = (int8_t)(b - tmp);
                            tmp = (int8_t)(z - q2); a2 = (int8_t)(a2 - tmp); b2 = (int8_t)(b2 - tmp);
                            nxv = (int8_t)((a > 0 ? a : 0) - qe);    d |= a > 0 ? 0x08 : 0;
                            nyv = (int8_t)((b > 0 ? b : 0) - qe);    d |= b > 0 ? 0x10 : 0;
                            nx2v = (int8_t)((a2 > 0 ? a2 : 0) - qe2); d |= a2 > 0 ? 0x20 : 0;
                            ny2v = (int8_t)((b2 > 0 ? b2 : 0) - qe2); d |= b2 > 0 ? 0x40 : 0;
                        } else {
                            d = z > a ? 0 : 1;  z = z > a ? z : a;
                            d = z > b ? d : 2;  z = z > b ? z : b;
                            d = z > a2 ? d : 3; z = z > a2 ? z : a2;
                            d = z > b2 ? d : 4; z = z > b2 ? z : b2;
                            z = z < sc_mch ? z : sc_mch;
                            tmp = (int8_t)(z - q);  a = (int8_t)(a - tmp);  b = (int8_t)(b - tmp);
                            tmp = (int8_t)(z - q2); a2 = (int8_t)(a2 - tmp); b2 = (int8_t)(b2 - tmp);
                            nxv = (int8_t)((0 > a ? 0 : a) - qe);    d |= 0 > a ? 0 : 0x08;
                            nyv = (int8_t)((0 > b ? 0 : b) - qe);    d |= 0 > b ? 0 : 0x10;
                            nx2v = (int8_t)((0 > a2 ? 0 : a2) - qe2); d |= 0 > a2 ? 0 : 0x20;
                            ny2v = (int8_t)((0 > b2 ? 0 : b2) - qe2); d |= 0 > b2 ? 0 : 0x40;
                        }
                        ou |= (uint32_t)(uint8_t)(int8_t)(z - vt1) << sh; ov |= (uint32_t)(uint8_t)(int8_t)(z - ut) << sh;
                        ox |= (uint32_t)(uint8_t)nxv << sh; oy |= (uint32_t)(uint8_t)nyv << sh; ox2 |= (uint32_t)(uint8_t)nx2v << sh; oy2 |= (uint32_t)(uint8_t)ny2v << sh;
                        od |= (uint32_t)d << sh;
                    }
                    *(uint32_t *)(u + t) = ou; *(uint32_t *)(v + t) = ov; *(uint32_t *)(x + t) = ox; *(uint32_t *)(y + t) = oy;
                    *(uint32_t *)(x2 + t) = ox2; *(uint32_t *)(y2 + t) = oy2;
                    *(uint32_t *)(p + ((unsigned long long)r * (unsigned long long)n_col + (unsigned long long)(t - st))) = od;
                }
            }
        } else {
            // core loop over the rounded range, 64 cells at a time; every cell reads the previous anti-diagonal only
            int8_t cx1 = x1, cx21 = x21, cv1 = v1;
            for (int32_t tb = st; tb <= en; tb += 64) {
                const int32_t t = tb + (int32_t)lane;
                const bool on = t <= en;
                const int8_t ox = on ? ld8s(x + t, g) : (int8_t)0, ov = on ? ld8s(v + t, g) : (int8_t)0, ox2 = on ? ld8s(x2 + t, g) : (int8_t)0;
                int8_t xt1 = (int8_t)al_wave_shr1((int)ox, 0), vt1 = (int8_t)al_wave_shr1((int)ov, 0), x2t1 = (int8_t)al_wave_shr1((int)ox2, 0);
                if (lane == 0) { xt1 = cx1; vt1 = cv1; x2t1 = cx21; }
                cx1 = (int8_t)__builtin_amdgcn_readlane((int)ox, 63); cv1 = (int8_t)__builtin_amdgcn_readlane((int)ov, 63); cx21 = (int8_t)__builtin_amdgcn_readlane((int)ox2, 63);
                if (on) {
                    int8_t z = ld8s(s + t, g);
                    const int8_t ut = ld8s(u + t, g);
                    int8_t a = (int8_t)(xt1 + vt1), b = (int8_t)(ld8s(y + t, g) + ut), a2 = (int8_t)(x2t1 + vt1), b2 = (int8_t)(ld8s(y2 + t, g) + ut), tmp;
                    uint8_t d;
                    if (!(flag & EZ_RIGHT)) {
                        d = a > z ? 1 : 0;  z = z > a ? z : a;
                        d = b > z ? 2 : d;  z = z > b ? z : b;
                        d = a2 > z ? 3 : d; z = z > a2 ? z : a2;
                        d = b2 > z ? 4 : d; z = z > b2 ? z : b2;
                        z = z < sc_mch ? z : sc_mch;
                        u[t] = (int8_t)(z - vt1); v[t] = (int8_t)(z - ut);
                        tmp = (int8_t)(z - q);  a = (int8_t)(a - tmp);  b = (int8_t)(b - tmp);
                        tmp = (int8_t)(z - q2); a2 = (int8_t)(a2 - tmp); b2 = (int8_t)(b2 - tmp);
                        x[t] = (int8_t)((a > 0 ? a : 0) - qe);    d |= a > 0 ? 0x08 : 0;
                        y[t] = (int8_t)((b > 0 ? b : 0) - qe);    d |= b > 0 ? 0x10 : 0;
                        x2[t] = (int8_t)((a2 > 0 ? a2 : 0) - qe2); d |= a2 > 0 ? 0x20 : 0;
                        y2[t] = (int8_t)((b2 > 0 ? b2 : 0) - qe2); d |= b2 > 0 ? 0x40 : 0;
                    } else {
                        d = z > a ? 0 : 1;  z = z > a ? z : a;
                        d = z > b ? d : 2;  z = z > b ? z : b;
                        d = z > a2 ? d : 3; z = z > a2 ? z : a2;
                        d = z > b2 ? d : 4; z = z > b2 ? z : b2;
                        z = z < sc_mch ? z : sc_mch;
                        u[t] = (int8_t)(z - vt1); v[t] = (int8_t)(z - ut);
                        tmp = (int8_t)(z - q);  a = (int8_t)(a - tmp);  b = (int8_t)(b - tmp);
                        tmp = (int8_t)(z - q2); a2 = (int8_t)(a2 - tmp); b2 = (int8_t)(b2 - tmp);
                        x[t] = (int8_t)((0 > a ? 0 : a) - qe);    d |= 0 > a ? 0 : 0x08;
                        y[t] = (int8_t)((0 > b ? 0 : b) - qe);    d |= 0 > b ? 0 : 0x10;
                        x2[t] = (int8_t)((0 > a2 ? 0 : a2) - qe2); d |= 0 > a2 ? 0 : 0x20;
                        y2[t] = (int8_t)((0 > b2 ? 0 : b2) - qe2); d |= 0 > b2 ? 0 : 0x40;
                    }
                    p[(unsigned long long)r * (unsigned long long)n_col + (unsigned long long)(t - st)] = d;
                }
            }
        }
        dsync();
        if (flag & EZ_APPROX_MAX) {      // ksw2's approximate maximum: follow one path (the first pass of the long-read gap filling)
            if (r > 0) {
                if (last_H0_t >= st0 && last_H0_t <= en0 && last_H0_t + 1 >= st0 && last_H0_t + 1 <= en0) {
                    const int32_t d0 = (int32_t)ld8s(v + last_H0_t, g), d1 = (int32_t)ld8s(u + last_H0_t + 1, g);
                    if (d0 > d1) H0 += d0;
                    else { H0 += d1; ++last_H0_t; }
                } else if (last_H0_t >= st0 && last_H0_t <= en0) H0 += (int32_t)ld8s(v + last_H0_t, g);
                else { ++last_H0_t; H0 += (int32_t)ld8s(u + last_H0_t, g); }
            } else { H0 = (int32_t)ld8s(v, g) - qe; last_H0_t = 0; }
            if (flag & EZ_APPROX_DROP) {
                bool brk = false;
                if (H0 > ez.max) { ez.max = H0; ez.max_t = last_H0_t; ez.max_q = r - last_H0_t; }
                else if (last_H0_t >= ez.max_t && r - last_H0_t >= ez.max_q) {
                    const int32_t tl = last_H0_t - ez.max_t, ql = (r - last_H0_t) - ez.max_q, l = tl > ql ? tl - ql : ql - tl;
                    if (zdrop >= 0 && ez.max - H0 > zdrop + l * e2) { ez.zdropped = 1; brk = true; }
                }
                if (brk) break;
            }
            if (r == qlen + tlen - 2 && en0 == tlen - 1) ez.score = H0;
            last_st = st; last_en = en;
            continue;
        }
        // exact maximum through the 32-bit score array; ties resolve as upstream's 4-lane scan does: the last cell first, then
        // lane class (t - st0) & 3 in rising order (smallest t inside a class), then the scalar tail in rising t
        int32_t max_H, max_t, h_st0, h_en0;
        if (r > 0) {
            const int32_t en1 = st0 + (en0 - st0) / 4 * 4;
            const int32_t h_last = en0 > 0 ? ld32(H + en0 - 1, g) + (int32_t)ld8s(u + en0, g) : ld32(H + en0, g) + (int32_t)ld8s(v + en0, g);
            dsync();
            long long best = ((long long)h_last << 32) | 0x7fffffffll;
            int32_t mine_st0 = KSW_NEG_INF;
            for (int32_t tb = st0; tb < en0; tb += 64) {
                const int32_t t = tb + (int32_t)lane;
                if (t < en0) {
                    const int32_t h = ld32(H + t, g) + (int32_t)ld8s(v + t, g);
                    H[t] = h;
                    if (t == st0) mine_st0 = h;
                    const uint32_t cls = t < en1 ? (uint32_t)((t - st0) & 3) : 4u;
                    const long long key = ((long long)h << 32) | (long long)(((6u - cls) << 26) | (0x3ffffffu - (uint32_t)t));
                    best = key > best ? key : best;
                }
            }
            if (lane == 0) H[en0] = h_last;
            best = wave_all_max_i64(best);
            max_H = (int32_t)(best >> 32);
            const uint32_t lo = (uint32_t)best;
            max_t = lo == 0x7fffffffu ? en0 : (int32_t)(0x3ffffffu - (lo & 0x3ffffffu));
            h_en0 = h_last;
            h_st0 = st0 == en0 ? h_last : al_b0(mine_st0);      // lane 0 of the first chunk owns t = st0
        } else {
            const int32_t h0 = (int32_t)ld8s(v, g) - qe;
            if (lane == 0) H[0] = h0;
            max_H = h0; max_t = 0; h_en0 = h0; h_st0 = h0;
        }
        if (en0 == tlen - 1 && h_en0 > ez.mte) { ez.mte = h_en0; ez.mte_q = r - en; }
        if (r - st0 == qlen - 1 && h_st0 > ez.mqe) { ez.mqe = h_st0; ez.mqe_t = st0; }
        {   // ksw_apply_zdrop
            bool brk = false;
            if (max_H > ez.max) { ez.max = max_H; ez.max_t = max_t; ez.max_q = r - max_t; }
            else if (max_t >= ez.max_t && r - max_t >= ez.max_q) {
                const int32_t tl = max_t - ez.max_t, ql = (r - max_t) - ez.max_q, l = tl > ql ? tl - ql : ql - tl;
                if (zdrop >= 0 && ez.max - max_H > zdrop + l * e2) { ez.zdropped = 1; brk = true; }
            }
            if (brk) break;
        }
        if (r == qlen + tlen - 2 && en0 == tlen - 1) ez.score = h_en0;
        last_st = st; last_en = en;
    }
    al_sync();
    auto rd = [&](unsigned long long i) -> uint32_t { return in_lds ? (uint32_t)p[i] : (uint32_t)cc_u8(p + i); };
    const bool rev_cigar = (flag & EZ_REV_CIGAR) != 0;
    if (!ez.zdropped && !(flag & EZ_EXTZ_ONLY)) ksw_backtrack_dev(rev_cigar, rd, off, off_end, n_col, tlen - 1, qlen - 1, cigar, ez.n_cigar);
    else if (!ez.zdropped && (flag & EZ_EXTZ_ONLY) && ez.mqe + end_bonus > ez.max) {
        ez.reach_end = 1;
        ksw_backtrack_dev(rev_cigar, rd, off, off_end, n_col, ez.mqe_t, qlen - 1, cigar, ez.n_cigar);
    } else if (ez.max_t >= 0 && ez.max_q >= 0) ksw_backtrack_dev(rev_cigar, rd, off, off_end, n_col, ez.max_t, ez.max_q, cigar, ez.n_cigar);
    al_sync();
}

template <class LDS>
__device__ inline void ksw_extd2_wave(int32_t qlen, const uint8_t *query, bool qg, int32_t tlen, const uint8_t *target, bool tg,
                                      int8_t sc_mch, int8_t sc_mis, int8_t sc_N, int32_t q, int32_t e, int32_t q2, int32_t e2, int32_t w,
                                      int32_t zdrop, int32_t end_bonus, int32_t flag, Ez &ez, uint32_t *cigar, AlignScratch &A, LDS &Ls)
{
    const int32_t T16 = (tlen + 15) / 16 * 16, Q16 = (qlen + 15) / 16 * 16;
    int32_t ww = w < 0 ? (tlen > qlen ? tlen : qlen) : w, nc = qlen < tlen ? qlen : tlen;
    nc = (((nc < ww + 1 ? nc : ww + 1) + 15) / 16 + 1) * 16;
    const bool mem_lds = qlen > 0 && tlen > 0 && T16 <= AL_T16 && Q16 <= AL_Q16;
    const bool p_lds = mem_lds && (unsigned long long)(qlen + tlen - 1) * (unsigned long long)nc <= AL_P;
    if (p_lds) ksw_extd2_core<false, false>(qlen, query, qg, tlen, target, tg, sc_mch, sc_mis, sc_N, q, e, q2, e2, w, zdrop, end_bonus, flag, ez, cigar, A, Ls);
    else if (mem_lds) ksw_extd2_core<false, true>(qlen, query, qg, tlen, target, tg, sc_mch, sc_mis, sc_N, q, e, q2, e2, w, zdrop, end_bonus, flag, ez, cigar, A, Ls);
    else ksw_extd2_core<true, true>(qlen, query, qg, tlen, target, tg, sc_mch, sc_mis, sc_N, q, e, q2, e2, w, zdrop, end_bonus, flag, ez, cigar, A, Ls);
}

// ------------------------------------------------------------------------------------------------
// regions: mm_gen_regs, mm_set_parent, mm_select_sub (hit.c), then mm_align_skeleton (align.c)
// ------------------------------------------------------------------------------------------------
struct AlignIn {
    const uint8_t *ref; const uint64_t *cstart; uint32_t n_contigs;       // reference: 4-bit codes
    const uint8_t *bases; const uint64_t *offsets;                         // reads (ASCII)
    const uint64_t *cx; const uint32_t *cq;                                // chain anchors
    const ChainRec *recs; const uint32_t *head;
};

__device__ inline void reg_set_coor(RegLite &r, int32_t qlen, int32_t k, const uint64_t *cx, const uint32_t *cq)
{
    const uint64_t x0 = cx[r.as], x1 = cx[r.as + r.cnt - 1];
    const int32_t y0 = (int32_t)cq[r.as], y1 = (int32_t)cq[r.as + r.cnt - 1];
    r.rev = (int32_t)(x0 >> 63); r.rid = (int32_t)(x0 << 1 >> 33);
    r.rs = (int32_t)x0 + 1 > k ? (int32_t)x0 + 1 - k : 0;
    r.re = (int32_t)x1 + 1;
    if (!r.rev) { r.qs = y0 + 1 - k; r.qe = y1 + 1; }
    else { r.qs = qlen - (y1 + 1); r.qe = qlen - (y0 + 1 - k); }
}

// mm_idx_getseq into the wave's tseq buffer, wave-parallel
__device__ inline void getseq_wave(const AlignIn &in, int32_t rid, int32_t st, int32_t en, uint8_t *out)
{
    const int32_t clen = (int32_t)(in.cstart[rid + 1] - in.cstart[rid]);
    if (en > clen) en = clen;
    const uint64_t g0 = in.cstart[rid];
    for (int32_t i = st + (int32_t)al_lane(); i < en; i += 64) { const uint64_t gg = g0 + (uint64_t)i; out[i - st] = (in.ref[gg >> 1] >> ((gg & 1) * 4)) & 15; }
    al_sync();
}

__device__ inline void seq_rev_wave(int32_t len, uint8_t *seq, bool g)
{
    for (int32_t i0 = 0; i0 < len >> 1; i0 += 64) {
        const int32_t i = i0 + (int32_t)al_lane();
        uint8_t t = 0, t2 = 0;
        if (i < len >> 1) { t = ld8(seq + i, g); t2 = ld8(seq + len - 1 - i, g); }
        al_sync();
        if (i < len >> 1) { seq[i] = t2; seq[len - 1 - i] = t; }
    }
    al_sync();
}

struct AlignOut { int32_t n_aligned, n_regs, dp_max; uint32_t sig; };

// minimap2's fast approximate log2 (mg_log2; valid for x >= 2), bit for bit the oracle's mmo_log2
__device__ inline float al_mg_log2(float x)
{
    union { float f; uint32_t i; } z = { x };
    float log_2 = (float)(int32_t)(((z.i >> 23) & 255) - 128);
    z.i &= ~(255U << 23);
    z.i += 127U << 23;
    log_2 += (-0.34484843f * z.f + 2.02466578f) * z.f - 0.67487759f;
    return log_2;
}

// lane-0 helpers over the region's cigar (plain memory: lane 0 is the only reader and writer) --------------------------
__device__ inline void append_cigar0(uint32_t *rc, int32_t &rn, int32_t n_cigar, const uint32_t *cigar)
{
    if (n_cigar == 0) return;
    if (rn > 0 && (rc[rn - 1] & 0xf) == (cigar[0] & 0xf)) {
        rc[rn - 1] += (cigar[0] >> 4) << 4;
        for (int32_t i = 1; i < n_cigar; ++i) rc[rn + i - 1] = cigar[i];
        rn += n_cigar - 1;
    } else {
        for (int32_t i = 0; i < n_cigar; ++i) rc[rn + i] = cigar[i];
        rn += n_cigar;
    }
}

// mm_fix_cigar + mm_update_extra (log_gap = 0) on lane 0.  qseq / tseq already offset to (qs1, rs1).
// mm_fix_cigar on lane 0: indel left-alignment, I/D run merging, removal of a leading I/D (which moves the region's start)
template <class REG>
__device__ inline void fix_cigar0(REG &r, uint32_t *c, int32_t &n_cigar, const uint8_t *qseq, bool qg, const uint8_t *tseq, bool tg, int32_t &qshift, int32_t &tshift)
{
    qshift = 0; tshift = 0;
    if (n_cigar > 1) {
        int32_t toff = 0, qoff = 0, to_shrink = 0, k;
        for (k = 0; k < n_cigar; ++k) {
            const uint32_t op = c[k] & 0xf, len = c[k] >> 4;
            if (len == 0) to_shrink = 1;
            if (op == 0) { toff += (int32_t)len; qoff += (int32_t)len; }
            else if (op == 1 || op == 2) {
                if (k > 0 && k < n_cigar - 1 && (c[k - 1] & 0xf) == 0 && (c[k + 1] & 0xf) == 0) {
                    int32_t l;
                    const int32_t prev_len = (int32_t)(c[k - 1] >> 4);
                    if (op == 1) { for (l = 0; l < prev_len; ++l) if (ld8(qseq + qoff - 1 - l, qg) != ld8(qseq + qoff + (int32_t)len - 1 - l, qg)) break; }
                    else { for (l = 0; l < prev_len; ++l) if (ld8(tseq + toff - 1 - l, tg) != ld8(tseq + toff + (int32_t)len - 1 - l, tg)) break; }
                    if (l > 0) { c[k - 1] -= (uint32_t)l << 4; c[k + 1] += (uint32_t)l << 4; qoff -= l; toff -= l; }
                    if (l == prev_len) to_shrink = 1;
                }
                if (op == 2) toff += (int32_t)len; else qoff += (int32_t)len;
            }
        }
        for (k = 0; k < n_cigar - 2; ++k) {
            if ((c[k] & 0xf) > 0 && (c[k] & 0xf) + (c[k + 1] & 0xf) == 3) {
                int32_t l;
                uint32_t s[3] = {0, 0, 0};
                for (l = k; l < n_cigar; ++l) {
                    const uint32_t op = c[l] & 0xf;
                    if (op == 1 || op == 2 || c[l] >> 4 == 0) s[op] += c[l] >> 4;
                    else break;
                }
                if (s[1] > 0 && s[2] > 0 && l - k > 2) {
                    c[k] = s[1] << 4 | 1u;
                    c[k + 1] = s[2] << 4 | 2u;
                    for (k += 2; k < l; ++k) c[k] &= 0xf;
                    to_shrink = 1;
                }
                k = l;
            }
        }
        if (to_shrink) {
            int32_t l = 0;
            for (k = 0; k < n_cigar; ++k) if (c[k] >> 4 != 0) c[l++] = c[k];
            n_cigar = l;
            for (k = l = 0; k < n_cigar; ++k)
                if (k == n_cigar - 1 || (c[k] & 0xf) != (c[k + 1] & 0xf)) c[l++] = c[k];
                else c[k + 1] += c[k] >> 4 << 4;
            n_cigar = l;
        }
        if ((c[0] & 0xf) == 1 || (c[0] & 0xf) == 2) {
            const int32_t l = (int32_t)(c[0] >> 4);
            if ((c[0] & 0xf) == 1) { if (r.rev) r.qe -= l; else r.qs += l; qshift = l; }
            else { r.rs += l; tshift = l; }
            --n_cigar;
            for (k = 0; k < n_cigar; ++k) c[k] = c[k + 1];
        }
    }
}

template <class REG>
__device__ inline void update_extra0(REG &r, uint32_t *c, int32_t &n_cigar, const uint8_t *qseq, bool qg, const uint8_t *tseq, bool tg,
                                     const AlignParams &P, int32_t &mlen_o, int32_t &blen_o, int32_t &dp_max_o, bool log_gap = false)
{
    int32_t qshift = 0, tshift = 0;
    fix_cigar0(r, c, n_cigar, qseq, qg, tseq, tg, qshift, tshift);
    qseq += qshift; tseq += tshift;
    int32_t toff = 0, qoff = 0, blen = 0, mlen = 0;
    double s = 0.0, max = 0.0;
    for (int32_t k = 0; k < n_cigar; ++k) {
        const uint32_t op = c[k] & 0xf, len = c[k] >> 4;
        if (op == 0) {
            int32_t n_ambi = 0, n_diff = 0;
            for (uint32_t l = 0; l < len; ++l) {
                const int32_t cq = ld8(qseq + qoff + (int32_t)l, qg), ct = ld8(tseq + toff + (int32_t)l, tg);
                if (ct > 3 || cq > 3) ++n_ambi;
                else if (ct != cq) ++n_diff;
                s += (ct > 3 || cq > 3) ? (P.sc_ambi > 0 ? -P.sc_ambi : P.sc_ambi) : (ct == cq ? (P.a < 0 ? -P.a : P.a) : (P.b > 0 ? -P.b : P.b));
                if (s < 0) s = 0;
                else max = max > s ? max : s;
            }
            blen += (int32_t)len - n_ambi; mlen += (int32_t)len - (n_ambi + n_diff);
            toff += (int32_t)len; qoff += (int32_t)len;
        } else if (op == 1) {
            int32_t n_ambi = 0;
            for (uint32_t l = 0; l < len; ++l) if (ld8(qseq + qoff + (int32_t)l, qg) > 3) ++n_ambi;
            blen += (int32_t)len - n_ambi;
            if (log_gap) s -= P.q + (double)P.e * al_mg_log2((float)(1.0 + len));
            else s -= P.q + P.e * (int32_t)len;
            if (s < 0) s = 0;
            qoff += (int32_t)len;
        } else if (op == 2) {
            int32_t n_ambi = 0;
            for (uint32_t l = 0; l < len; ++l) if (ld8(tseq + toff + (int32_t)l, tg) > 3) ++n_ambi;
            blen += (int32_t)len - n_ambi;
            if (log_gap) s -= P.q + (double)P.e * al_mg_log2((float)(1.0 + len));
            else s -= P.q + P.e * (int32_t)len;
            if (s < 0) s = 0;
            toff += (int32_t)len;
        }
    }
    mlen_o = mlen; blen_o = blen; dp_max_o = (int32_t)(max + .499);
}

// Flag-only first pass (ChainSink::best): the read's list holds the chains that were, when they were found, the largest z of the read;
// the one whose z equals best[read] is regs[0] of mm_gen_regs - primary, hence always aligned.  1: its max stretch alone guarantees
// that it survives mm_filter_regs (same test as sh_chain.h chain_lemma, over the handed-over anchors): the read is mapped.  0: not
// settled here (a tie in z, a stretch the test cannot vouch for): the read goes through the full procedure with all its chains.
struct MidReq;
template <class MID>
__device__ inline int32_t top_chain_settles(const AlignIn &in, const AlignParams &P, uint32_t read, unsigned long long best, uint32_t tie, MID &mid, uint32_t *h_out = nullptr)
{
    // returns 1 settled, 2 settled if the stretch described in `mid` shows no z-drop on the bases (the caller's wave checks:
    // middle_no_zdrop_wave), or minus the reason it is not: -1 tie / off, -2 record missing, -3 stretch too short, -4 z-drop in the stretch
    if (tie || !P.lemma) return -1;
    uint32_t h = in.head[read];
    while (h != ~0u && in.recs[h].z != best) h = in.recs[h].next;
    if (h == ~0u) return -2;
    if (h_out) *h_out = h;
    const ChainRec rc = in.recs[h];
    int32_t run_score = P.k, run_unc = 0, best_score = -1, best_unc = 0;
    uint32_t xp = (uint32_t)in.cx[rc.off], qp = in.cq[rc.off], run_first = 0, run_last = 0, b_first = 0, b_last = 0;
    for (uint32_t j = 1; j < rc.cnt; ++j) {
        const uint32_t xj = (uint32_t)in.cx[rc.off + j], qj = in.cq[rc.off + j];
        const int32_t lr = (int32_t)(xj - xp), lq = (int32_t)(qj - qp);
        if (lq == lr) { run_score += lq < P.k ? lq : P.k; run_unc += lq > P.k ? lq - P.k : 0; run_last = j; }
        else {
            if (run_score > best_score) { best_score = run_score; best_unc = run_unc; b_first = run_first; b_last = run_last; }
            run_score = P.k; run_unc = 0; run_first = run_last = j;
        }
        xp = xj; qp = qj;
    }
    if (run_score > best_score) { best_score = run_score; best_unc = run_unc; b_first = run_first; b_last = run_last; }
    const int32_t qf = (int32_t)in.cq[rc.off + b_first], ql = (int32_t)in.cq[rc.off + b_last];
    if (!(best_score >= P.min_sc && ql - qf >= P.k)) return -3;
    if (best_unc <= P.unc_max) return 1;
    const uint64_t x0 = in.cx[rc.off + b_first];
    mid.rid = (int32_t)(x0 << 1 >> 33); mid.rev = (int32_t)(x0 >> 63); mid.qs = qf + 1 - P.k; mid.qe = ql + 1; mid.rs = (int32_t)x0 + 1 - P.k;
    return 2;
}

// The whole stage for one read, one wave.  flag_only: stop at the first surviving region (the boundary only returns
// `mappings.len() > 0`); else every region is aligned and the counts / fingerprint are those of the oracle's trace.
// top_z != 0: only the chain with that z - regs[0] of mm_gen_regs, which is aligned whatever the other chains are - goes through
// mm_align1; a surviving region settles the read (n_regs > 0), none means the caller must run the full procedure on all chains.
template <class LDS>
__device__ inline bool align_read_wave(const AlignIn &in, const AlignParams &P, uint32_t read, bool flag_only, AlignScratch &A, LDS &Ls,
                                       AlignOut &out, uint32_t *overflow, unsigned long long top_z = 0)
{
    const uint32_t lane = al_lane();
    const int32_t qlen = (int32_t)(in.offsets[read + 1] - in.offsets[read]);
    out.n_aligned = out.n_regs = out.dp_max = 0; out.sig = 0;
    // ---- the read's chains (a linked list, newest first; every lane walks it)
    int32_t n_u = 0;
    uint32_t h_top = ~0u;
    for (uint32_t h = in.head[read]; h != ~0u; h = in.recs[h].next) { ++n_u; if (top_z != 0 && in.recs[h].z == top_z) { h_top = h; break; } }      // (a tandem-array read has hundreds of records: each step is a round trip to HBM)
    if (n_u == 0) return true;
    if (top_z != 0) { if (h_top == ~0u) return true; n_u = 1; }
    if ((uint32_t)n_u > A.reg_cap) { if (lane == 0) atomicExch(overflow, 2u); return false; }
    const bool rg = n_u > LDS::NR;                 // region arrays in HBM scratch
    RegLite *regs = rg ? A.regs : Ls.regs;
    uint64_t *kz = rg ? A.kz : Ls.kz, *kx = rg ? A.kx : Ls.kx, *kk = rg ? A.kk : Ls.kk;
    uint32_t *kh = rg ? A.kh : Ls.kh, *ord = rg ? A.ord : Ls.ord;
    if (top_z != 0) { if (lane == 0) kh[0] = h_top; }
    else {   // list position j -> lane j & 63 parks the record index in kh[j] (read back by the same lane)
        int32_t j = 0;
        for (uint32_t h = in.head[read]; h != ~0u; h = in.recs[h].next, ++j) if ((uint32_t)(j & 63) == lane) kh[j] = h;
    }
    // ---- mm_gen_regs.  compact_a orders the chains by their first anchor's x (ties: discovery order = descending (f, index))
    //      and gives each its index in u[]; the regions are then sorted by z = (score << 32 | cnt) ^ h ascending, stably, and
    //      reversed.  One rank per chain by the composite order: position in r[] = number of chains that precede it.
    uint32_t hash = 0;
    hash ^= al_wang((uint32_t)qlen) + al_wang(11u);
    hash = al_wang(hash);
    for (int32_t i = (int32_t)lane; i < n_u; i += 64) {
        const ChainRec rc = in.recs[kh[i]];
        const uint64_t x0 = in.cx[rc.off], y0 = (uint64_t)rc.yfl << 32 | (uint64_t)(uint32_t)P.k << 32 | in.cq[rc.off];
        const uint32_t h = (uint32_t)al_hash64((al_hash64(x0) + al_hash64(y0)) ^ hash);
        kz[i] = ((uint64_t)(uint32_t)rc.score << 32 | rc.cnt) ^ h;
        kx[i] = x0;
        kk[i] = (uint64_t)rc.key_f << 32 | rc.key_i;
    }
    al_sync();
    for (int32_t i0 = 0; i0 < n_u; i0 += 64) {
        const int32_t i = i0 + (int32_t)lane;
        const bool on = i < n_u;
        const uint64_t zi = on ? kz[i] : 0, xi = on ? kx[i] : 0, ki = on ? kk[i] : 0;       // own stores
        int32_t rank = 0;
        for (int32_t j0 = 0; j0 < n_u; j0 += 64) {
            const int32_t jm = j0 + (int32_t)lane;
            const uint64_t zj_ = jm < n_u ? kz[jm] : 0, xj_ = jm < n_u ? kx[jm] : 0, kj_ = jm < n_u ? kk[jm] : 0;
            const int32_t lim = n_u - j0 < 64 ? n_u - j0 : 64;
            for (int32_t l = 0; l < lim; ++l) {
                const uint64_t zj = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)zj_, l) | (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(zj_ >> 32), l) << 32;
                const uint64_t xj = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)xj_, l) | (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(xj_ >> 32), l) << 32;
                const uint64_t kj = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)kj_, l) | (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(kj_ >> 32), l) << 32;
                // j precedes i in r[]: larger z; then larger u-index, i.e. larger first-anchor x, then the LATER discovery (smaller (f, index))
                rank += (zj > zi) || (zj == zi && (xj > xi || (xj == xi && kj < ki)));
            }
        }
        if (on) ord[rank] = kh[i];
    }
    al_sync();
    for (int32_t i = (int32_t)lane; i < n_u; i += 64) {
        const ChainRec rc = in.recs[rg ? cc_u32(ord + i) : ord[i]];
        RegLite r;
        r.cnt = (int32_t)rc.cnt; r.score = rc.score; r.as = rc.off; r.parent = -1; r.pad = 0;
        reg_set_coor(r, qlen, P.k, in.cx, in.cq);
        regs[i] = r;
    }
    al_sync();

    // ---- mm_set_parent + mm_select_sub: scalar, lane 0 (plain accesses to what it wrote itself; the regions other lanes wrote
    //      are read past the L1 when they live in HBM)
    int32_t n_regs = n_u;
    if (lane == 0) {
        auto R_qs = [&](int32_t i) { return ld32(&regs[i].qs, rg); };
        auto R_qe = [&](int32_t i) { return ld32(&regs[i].qe, rg); };
        bool too_many = false;
        // the primaries' list and the coverage intervals: in LDS with the regions (AL_R chains: never more primaries than that), in the wave's
        // scratch with them otherwise - one instance of the loop each, so that the LDS one keeps its ds_ accesses
        auto set_parent = [&](int32_t *w, uint64_t *cov, const int32_t pri_cap) {
        int32_t k = 1;
        w[0] = 0; regs[0].parent = 0;
        for (int32_t i = 1; i < n_u; ++i) {
            const int32_t si = R_qs(i), ei = R_qe(i);
            int32_t n_cov = 0, uncov_len = 0, j;
            for (j = 0; j < k; ++j) {
                int32_t sj = R_qs(w[j]), ej = R_qe(w[j]);
                if (ej <= si || sj >= ei) continue;
                if (sj < si) sj = si;
                if (ej > ei) ej = ei;
                cov[n_cov++] = (uint64_t)(uint32_t)sj << 32 | (uint32_t)ej;
            }
            if (n_cov > 0) {
                for (int32_t a = 1; a < n_cov; ++a) { const uint64_t t = cov[a]; int32_t b = a; for (; b > 0 && cov[b - 1] > t; --b) cov[b] = cov[b - 1]; cov[b] = t; }
                int32_t x = si;
                for (j = 0; j < n_cov; ++j) {
                    if ((int32_t)(cov[j] >> 32) > x) uncov_len += (int32_t)(cov[j] >> 32) - x;
                    x = (int32_t)cov[j] > x ? (int32_t)cov[j] : x;
                }
                if (ei > x) uncov_len += ei - x;
                for (j = 0; j < k; ++j) {
                    const int32_t sj = R_qs(w[j]), ej = R_qe(w[j]);
                    if (ej <= si || sj >= ei) continue;
                    const int32_t mn = ej - sj < ei - si ? ej - sj : ei - si, mx = ej - sj > ei - si ? ej - sj : ei - si;
                    const int32_t ol = si < sj ? (ei < sj ? 0 : ei < ej ? ei - sj : ej - sj) : (ej < si ? 0 : ej < ei ? ej - si : ei - si);
                    if ((float)ol / mn - (float)uncov_len / mx > P.mask_level) { regs[i].parent = w[j]; break; }     // rp->parent of a primary is its own index
                }
            } else j = k;
            if (j == k) {
                if (k >= pri_cap) { too_many = true; break; }
                w[k++] = i; regs[i].parent = i;
            }
        }
        };
        if (rg) set_parent(A.pri, A.cov, (int32_t)A.reg_cap); else set_parent(Ls.pri, Ls.cov, (int32_t)LDS::NPRI);
        if (too_many) atomicExch(overflow, 3u);
        // mm_select_sub (check_strand = 1, min_strand_sc = max_gap * 0.8), in place like upstream: r[p] is read AFTER earlier
        // regions moved up, so a parent index can alias a later region - kept literally
        if (P.pri_ratio > 0.0f && n_u > 0 && !too_many) {
            const int32_t min_diff = P.k * 2, min_strand_sc = (int32_t)(P.max_gap * 0.8);
            int32_t kk2 = 0, n_2nd = 0;
            auto RD = [&](int32_t i) {      // a region as lane 0 must see it: parent is its own store, the rest may be another lane's
                RegLite r; const RegLite *s = regs + i;
                r.cnt = ld32(&s->cnt, rg); r.score = ld32(&s->score, rg); r.qs = ld32(&s->qs, rg); r.qe = ld32(&s->qe, rg); r.rs = ld32(&s->rs, rg); r.re = ld32(&s->re, rg);
                r.rid = ld32(&s->rid, rg); r.rev = ld32(&s->rev, rg); r.pad = 0; r.as = ld64((const uint64_t *)&s->as, rg); r.parent = ld32(&s->parent, rg);
                return r;
            };
            // once lane 0 has rewritten slot kk2 (kk2 <= i) it reads its own store back: plain loads are right for slots < kk2, the
            // cc loads for untouched ones - both go through RD (an own store is visible to a cc load of the same lane after the wait
            // that atomic loads imply); keep a per-slot origin to be safe
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            for (int32_t i = 0; i < n_u; ++i) {
                const RegLite ri = RD(i);
                const int32_t p = ri.parent;
                if (p == i) { regs[kk2++] = ri; continue; }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                const RegLite rp = RD(p);
                if ((ri.score >= rp.score * P.pri_ratio || ri.score + min_diff >= rp.score) && n_2nd < P.best_n) {
                    if (!(ri.qs == rp.qs && ri.qe == rp.qe && ri.rid == rp.rid && ri.rs == rp.rs && ri.re == rp.re)) { regs[kk2++] = ri; ++n_2nd; }
                } else if (n_2nd < P.best_n && ri.score > min_strand_sc && rp.rev != ri.rev && rp.rid == ri.rid && ri.rs < rp.re && ri.re > rp.rs) {
                    regs[kk2++] = ri; ++n_2nd;
                }
            }
            n_regs = kk2;
        }
    }
    n_regs = al_b0(n_regs);
    al_sync();
    out.n_aligned = n_regs;
    if (ld32((const int32_t *)overflow, true) != 0) return false;

    // ---- mm_align_skeleton ---------------------------------------------------------------------------------------
    const bool sg = qlen > AL_Q;                 // query codes / reference window in HBM scratch
    uint8_t *qseq0 = sg ? A.qseq : Ls.qseq, *tseq = sg ? A.tseq : Ls.tseq;
    if (sg && ((uint32_t)qlen > A.qcap)) { if (lane == 0) atomicExch(overflow, 4u); return false; }
    {
        const uint8_t *seq = in.bases + in.offsets[read];
        for (int32_t i = (int32_t)lane; i < qlen; i += 64) {
            const uint8_t c = (uint8_t)sh_nt4(seq[i]);
            qseq0[i] = c;
            qseq0[qlen + (qlen - 1 - i)] = c < 4 ? 3 - c : 4;
        }
        al_sync();
    }
    const int8_t sc_mch = (int8_t)(P.a < 0 ? -P.a : P.a), sc_mis = (int8_t)(P.b > 0 ? -P.b : P.b), sc_amb = (int8_t)(P.sc_ambi > 0 ? -P.sc_ambi : P.sc_ambi);
    const int8_t sc_N = sc_amb == 0 ? (int8_t)(-P.e2) : sc_amb;      // ksw_extd2: sc_N = mat[24] == 0 ? -e2 : mat[24]; every other user reads mat[24]
    const int32_t bw = (int32_t)(P.bw * 1.5 + 1.);
    int32_t bw_long = (int32_t)(P.bw_long * 1.5 + 1.);
    if (bw_long < bw) bw_long = bw;
    uint32_t sig = 2166136261u;
    int32_t n_keep = 0, dp_best = 0;
    uint32_t *rc = A.r_cigar, *ezc = A.ez_cigar;

    for (int32_t i = 0; i < n_regs; ++i) {
        // region i, broadcast from lane 0 (which may have moved it in select_sub / a split)
        RegLite r;
        {
            const RegLite *s = regs + i;
            r.cnt = al_b0(lane == 0 ? s->cnt : 0); r.score = al_b0(lane == 0 ? s->score : 0);
            r.as = al_b0_64(lane == 0 ? s->as : 0ull);
            r.qs = r.qe = r.rs = r.re = 0; r.parent = 0; r.pad = 0;
        }
        const uint64_t a0x = in.cx[r.as];
        const int32_t rid = (int32_t)(a0x << 1 >> 33), rev = (int32_t)(a0x >> 63);
        r.rid = rid; r.rev = rev;
        if (r.cnt == 0) continue;
        // mm_max_stretch (lane 0)
        int32_t as1 = 0, cnt1 = r.cnt;
        if (lane == 0 && r.cnt >= 2) {
            int32_t score = P.k, len = 1, max_score = -1, max_i = -1, max_len = 0, j;
            for (j = 0; j < r.cnt - 1; ++j) {
                const int32_t lr = (int32_t)in.cx[r.as + j + 1] - (int32_t)in.cx[r.as + j], lq = (int32_t)in.cq[r.as + j + 1] - (int32_t)in.cq[r.as + j];
                if (lq == lr) { score += lq < P.k ? lq : P.k; ++len; }
                else {
                    if (score > max_score) { max_score = score; max_len = len; max_i = j - len + 1; }
                    score = P.k; len = 1;
                }
            }
            if (score > max_score) { max_score = score; max_len = len; max_i = j - len + 1; }
            as1 = max_i; cnt1 = max_len;
        }
        as1 = al_b0(as1); cnt1 = al_b0(cnt1);
        const unsigned long long sa = r.as + (unsigned long long)as1;        // first anchor of the stretch
        int32_t rs = (int32_t)in.cx[sa] + 1 - P.k, qs = (int32_t)in.cq[sa] + 1 - P.k;
        int32_t re = (int32_t)in.cx[sa + cnt1 - 1] + 1, qe = (int32_t)in.cq[sa + cnt1 - 1] + 1;
        const int32_t clen = (int32_t)(in.cstart[rid + 1] - in.cstart[rid]);
        const int32_t qs0 = 0, qe0 = qlen;
        int32_t l = qs;
        l += l * P.a + P.end_bonus > P.q ? (l * P.a + P.end_bonus - P.q) / P.e : 0;
        const int32_t rs0 = rs - l > 0 ? rs - l : 0;
        l = qlen - qe;
        l += l * P.a + P.end_bonus > P.q ? (l * P.a + P.end_bonus - P.q) / P.e : 0;
        const int32_t re0 = re + l < clen ? re + l : clen;
        if ((sg && (uint32_t)(re0 - rs0) > A.tcap) || (!sg && re0 - rs0 > AL_T)) { if (lane == 0) atomicExch(overflow, 5u); return false; }
        uint8_t *qrow = qseq0 + (rev ? qlen : 0);
        int32_t rn = 0, dropped = 0, rs1, qs1, re1, qe1, r2_cnt = 0, split_n = 0;
        Ez ez;
        if (qs > 0 && rs > 0) {       // left extension
            getseq_wave(in, rid, rs0, rs, tseq);
            seq_rev_wave(qs - qs0, qrow + qs0, sg);
            seq_rev_wave(rs - rs0, tseq, sg);
            ksw_extd2_wave(qs - qs0, qrow + qs0, sg, rs - rs0, tseq, sg, sc_mch, sc_mis, sc_N, P.q, P.e, P.q2, P.e2, bw, P.zdrop, P.end_bonus,
                           EZ_EXTZ_ONLY | EZ_RIGHT | EZ_REV_CIGAR, ez, ezc, A, Ls);
            if (lane == 0 && ez.n_cigar > 0) append_cigar0(rc, rn, ez.n_cigar, ezc);
            rs1 = rs - (ez.reach_end ? ez.mqe_t + 1 : ez.max_t + 1);
            qs1 = qs - (ez.reach_end ? qs - qs0 : ez.max_q + 1);
            seq_rev_wave(qs - qs0, qrow + qs0, sg);
        } else { rs1 = rs; qs1 = qs; }
        re1 = rs; qe1 = qs;
        {   // gap filling: the whole stretch, ungapped
            re = (int32_t)in.cx[sa + cnt1 - 1] + 1; qe = (int32_t)in.cq[sa + cnt1 - 1] + 1;
            re1 = re; qe1 = qe;
            getseq_wave(in, rid, rs, re, tseq);
            // mm_test_zdrop on one M of qe - qs (lane 0)
            int32_t zcode = 0;
            if (lane == 0) {
                int32_t score = 0, mx = INT32_MIN, mxi = -1, max_zdrop = 0;
                for (int32_t j = 0; j < qe - qs; ++j) {
                    const int32_t cq = ld8(qrow + qs + j, sg), ct = ld8(tseq + j, sg);
                    score += (ct > 3 || cq > 3) ? sc_amb : (ct == cq ? sc_mch : sc_mis);
                    if (score < mx) { const int32_t z = mx - score; (void)mxi; if (z > max_zdrop) max_zdrop = z; }     // li == lj on an ungapped path: diff = 0
                    else { mx = score; mxi = j; }
                }
                zcode = max_zdrop > P.zdrop ? 1 : 0;
                ezc[0] = (uint32_t)(qe - qs) << 4;
            }
            zcode = al_b0(zcode);
            ez_reset(ez); ez.n_cigar = 1;
            if (zcode != 0)
                ksw_extd2_wave(qe - qs, qrow + qs, sg, re - rs, tseq, sg, sc_mch, sc_mis, sc_N, P.q, P.e, P.q2, P.e2, bw_long, P.zdrop, -1, 0, ez, ezc, A, Ls);
            if (lane == 0 && ez.n_cigar > 0) append_cigar0(rc, rn, ez.n_cigar, ezc);
            if (ez.zdropped) {
                int32_t j = cnt1 - 2;
                if (lane == 0) { for (; j >= 0; --j) if ((int32_t)in.cx[sa + j] <= rs + ez.max_t) break; }
                j = al_b0(j);
                dropped = 1;
                if (j < 0) j = 0;
                re1 = rs + (ez.max_t + 1);
                qe1 = qs + (ez.max_q + 1);
                if (cnt1 - (j + 1) >= P.min_cnt) {     // mm_split_reg(r, r2, as1 + j + 1 - r->as)
                    const int32_t n = as1 + j + 1;
                    if (n > 0 && n < r.cnt) { split_n = n; r2_cnt = r.cnt - n; }
                }
            } else { rs = re; qs = qe; }
        }
        if (!dropped && qe < qe0 && re < re0) {   // right extension
            getseq_wave(in, rid, re, re0, tseq);
            ksw_extd2_wave(qe0 - qe, qrow + qe, sg, re0 - re, tseq, sg, sc_mch, sc_mis, sc_N, P.q, P.e, P.q2, P.e2, bw, P.zdrop, P.end_bonus, EZ_EXTZ_ONLY, ez, ezc, A, Ls);
            if (lane == 0 && ez.n_cigar > 0) append_cigar0(rc, rn, ez.n_cigar, ezc);
            re1 = re + (ez.reach_end ? ez.mqe_t + 1 : ez.max_t + 1);
            qe1 = qe + (ez.reach_end ? qe0 - qe : ez.max_q + 1);
        }
        // a split leaves r with its first split_n anchors and queues r2 = the rest right behind it
        int32_t r_cnt = r.cnt;
        if (r2_cnt > 0) {
            if ((uint32_t)(n_regs + 1) > (rg ? A.reg_cap : (uint32_t)LDS::NR)) { if (lane == 0) atomicExch(overflow, 6u); return false; }
            if (lane == 0) {
                RegLite r2; r2.cnt = r2_cnt; r2.as = r.as + (unsigned long long)split_n; r2.score = (int32_t)(r.score * ((float)r2_cnt / r.cnt) + .499);
                r2.qs = r2.qe = r2.rs = r2.re = 0; r2.parent = -2; r2.rid = rid; r2.rev = rev; r2.pad = 0;
                for (int32_t m = n_regs - 1; m > i; --m) regs[m + 1] = regs[m];
                regs[i + 1] = r2;
            }
            r_cnt = split_n;
            ++n_regs;
            al_sync();
        }
        r.rs = rs1; r.re = re1;
        if (rev) { r.qs = qlen - qe1; r.qe = qlen - qs1; } else { r.qs = qs1; r.qe = qe1; }
        // mm_update_extra + this region's share of mm_filter_regs (lane 0)
        getseq_wave(in, rid, rs1, re1, tseq);
        int32_t keep = 0, mlen = 0, blen = 0, dpm = 0;
        if (lane == 0) {
            update_extra0(r, rc, rn, qrow + qs1, sg, tseq, sg, P, mlen, blen, dpm);
            int32_t flt = 0;
            if (r_cnt < P.min_cnt) flt = 1;
            if (mlen < P.min_sc) flt = 1;
            else if (dpm < P.min_dp_max) flt = 1;
            else if (r.qs > qlen * P.max_clip_ratio && qlen - r.qe > qlen * P.max_clip_ratio) flt = 1;
            keep = !flt;
            if (keep) {
                const int32_t v[8] = { r.rs, r.re, r.qs, r.qe, mlen, blen, dpm, r_cnt };
                for (int32_t j = 0; j < 8; ++j) { sig ^= (uint32_t)v[j]; sig *= 16777619u; }
                if (dpm > dp_best) dp_best = dpm;
                ++n_keep;
            }
        }
        keep = al_b0(keep);
        if (keep && flag_only) { out.n_regs = 1; return true; }
    }
    out.n_regs = al_b0(n_keep); out.dp_max = al_b0(dp_best); out.sig = out.n_regs > 0 ? (uint32_t)al_b0((int32_t)sig) : 0u;
    return true;
}
