// sh_classify.hip — per-read host/non-host classification on gfx950.
//
// Replaces the hot loop of the reference's in-process aligner path
//     .map(|(id, sequence)| aligner.map(&sequence, false, false, None, None) ... mappings.len() > 0)
//     /root/reference/src/cleaner.rs:550-558
// with hand-written HIP kernels over a read batch resident in HBM (DESIGN.md §3 has the table with what bounds each):
//
//   K1 k_sketch_probe   one wave per tile of 64 consecutive reads, one lane per read.
//        A  the tile's bases (one contiguous byte range) are loaded coalesced, 16 B per lane,
//           converted to 2-bit codes + an ambiguity bit and staged in LDS;
//        B  every lane runs the (w,k)-minimizer state machine over its read with the w-entry
//           window in VGPRs (sh_sketch.h: SketchPacked, one 64-bit word per ring entry), queueing
//           minimizers in a per-lane LDS ring;
//        C  every W-step block starts by issuing up to 4 home-slot gathers per lane for what earlier
//           blocks queued and ends by consuming them (one 16-B seed record per hit, per-read row).
//        Reads without a single hit are final here (flag 0): no anchor => no mapping; the others are
//        routed to the LDS path (no seed above mid_occ, <= 32 anchors) or to the repeat path.
//   k_pair_pass         flag-only calls: one lane per read over either list; two singleton seeds on one
//        diagonal decide a read from its seed records alone (ChainParams::pair_dq_*); the undecided
//        reads are compacted for the kernels below.
//   K2 k_chain_small    one lane per read of the LDS path: anchors, chaining DP and backtrack
//        entirely in LDS (11 B per anchor, lane-interleaved), up to CAP anchors.
//   K3 the repeat path, for reads with more anchors (29 % of the host reads of the CHM13-sized
//        workload, carrying 90 % of all anchors).  Sorted anchors split into CLUSTERS at every gap
//        > max_dist_x (or strand / contig change); the chaining DP window, its max_ii shortcut and
//        the backtrack never cross such a gap, so clusters are independent DP problems and the
//        per-read result is (sum of chains, max score) over its clusters:
//          k_expand       one wave per read: lane-per-seed occurrence filter (mm_seed_select by ballots);
//                         flag-only: the general pair test (pair_decides: occurrence lists walked and
//                         binary-searched); else anchors written to the HBM arena, <= 64 anchors sorted in
//                         registers and chained on the spot
//          k_group_probe  flag-only: reads with thousands of anchors, decided from their largest
//                         (strand, contig) group when possible
//          k_sort_lds     one block per read with <= 256 ... 4096 anchors: stable merge sort in LDS, then
//                         every cluster chained straight from LDS; only the per-read result returns to HBM
//          k_giant_*      more anchors: grid-wide tiled merge sort in the arena, then k_giant_chain /
//                         k_cluster_dp (one wave per large cluster, LDS ring DP)
//          k_finalize     per read: flag / trace, or hand the read to the second (max_occ) pass
//   k_long_*        long reads (any length): segment-parallel sketch, thinning, probe; feeds the repeat path.
//   k_chain_large   legacy lane-per-read path over arena slices, kept for the rare reads no front end takes.
//
// Results are bit-identical to oracle/mm_oracle.c (tests/test_parity_gpu.py).
#include "sh_common.h"
#include "sh_wave.h"
#include "sh_sketch.h"
#include "sh_chain.h"
#include "sh_long.h"
#include <algorithm>
#include <chrono>

#define SH_SPLIT ((sh_status)-2)      // internal: classify_chunk asks for smaller chunks
#define K1_LIST_CAP 8           // ring of queued minimizers per lane (power of two; 4 KiB of LDS per wave); 4 are drained per W-step block
#define K2_CAP 32               // anchors per read chained in LDS
#define DP_SMALL_CAP 32         // anchors per cluster chained in LDS
#define SORT_LDS_A 512          // reads with up to this many anchors are sorted and chained in 12 KiB of LDS
#define SORT_LDS_B 2048         // ... 48 KiB
#define SORT_LDS_C 4096         // ... 96 KiB; larger ones: 4096-anchor chunks in LDS, then merge rounds in the arena
#define N_SORT_CLS 6             // LDS classes <= 256, 512, 1024, 2048, 4096 anchors (block LDS sized to the class: occupancy), then giant
#define SORT_CLS_GIANT 5
#define GT 2048u                 // tile of the giant-read merge sort
// k_lr_locus (long-read presets, flag-only): reads of up to LOCUS_N0 / LOCUS_N1 / LOCUS_MAX_N anchors count their anchors per reference
// window in 2^12 / 2^14 / 2^15 sixteen-bit counters of LDS; a run of more than LOCUS_RUN_MAX occupied windows is kept whatever it holds
#define LOCUS_N0 1024u
#define LOCUS_N1 4096u
#define LOCUS_MAX_N 65535u
#define LOCUS_RUN_MAX 16

__device__ inline uint32_t lane_id() { return threadIdx.x & 63; }

// Device counters.  A single address sustains only ~90 M atomics/s on MI355X, so list tails are advanced
// by whole chunks per wave (WaveAlloc) and statistics are sharded over 64 addresses.
struct Counters {
    uint32_t n_small, n_resketch, n_big[2], n_big_defer[2], n_defer, n_small2;      // n_small2: LDS-path reads the pair pass left undecided
    uint32_t n_big_total, pad1;   // repeat-path reads before the pair pass took its share (0: no pair pass)
    uint32_t n_sort[N_SORT_CLS], n_giant_tiles, n_giant_rounds;
    uint32_t expand_ticket, pad_t;         // k_expand: next read of the pass's list
    uint32_t top_ticket[N_SORT_CLS];       // k_sort_top / k_giant_top: likewise
    uint32_t sort_ticket[N_SORT_CLS];      // k_sort_lds: next item of the class (blocks draw reads one by one: their costs differ a hundredfold)
    uint32_t n_long_segs, pad2;
    uint32_t ext_reason[8];       // why the top chain did not settle a read (k_ext_top)
    uint32_t lext_exact_direct, lext_pad_d;      // giants whose first pass ran on the EXACT instance and asked the tree
    uint32_t ext_s3[8];           // SCRUBBY_HIP_DBG & 16: outcome of the local-cluster shortcut (k_expand): 0 tried, 1 no singleton / filtered seed, 2 singletons apart, 3 window grew / too many, 4 K > 64, 5 no margin over U_out, 6 lemma, 7 decided
    uint32_t ext_overflow, ext_n_list, ext_ticket, ext_regions, ext_dropped, ext_n_redo, ext_n_list2, ext_ticket2, ext_n_redo2, ext_ticket3, ext_n_unres, ext_ticket_unres, ext_n_unres_in, ext_pad8;      // extension stage (sh_align.h)
    uint32_t ext_n_recs[SINK_SHARDS]; unsigned long long ext_n_anch[SINK_SHARDS];                 // hand-over cursors, one per shard
    uint32_t lext_n_exact2, lext_ticket_exact2;      // the exact pass's second list (reads of the 4096-anchor ring)
    uint32_t lext_hist[64]; uint32_t lext_ticket_g, lext_pad3; uint32_t lext_n_big, lext_ticket_big, lext_n_big2, lext_ticket_big2, lext_ticket_b, lext_pad2, lext_rechained, lext_rmq_tie, lext_err_read, lext_unresolved, lext_err_code, lext_pad;
    unsigned long long lext_clk[LR_NCLK], lext_d[8], lext_slow, lext_kernel_sum;  // long-read extension stage (sh_long.h): reads for the large-scratch pass, RMQ re-chains, steps with tied priorities
    uint32_t n_cl[4], cl_ticket, cl_ticket3;      // global queue of big clusters (k_cluster_dp), by size class; tickets: classes 0-2 one by one, class 3 eight at a time
    uint32_t n_leg_reason[4];     // why reads left the long-read front end: 0 room/segments, 1 thinning screen, 2 anchors beyond the giant path, 3 unused
    unsigned long long arena_cursor, anchor_cursor;
    unsigned long long cl_tot[4], cl_anchor_tot[4], cl_dbg[10], pf_dbg[16];
    unsigned long long sort_tot[N_SORT_CLS + 1], sort_anchor_tot[N_SORT_CLS + 1];    // SCRUBBY_HIP_DBG & 16: reads / anchors per sort class (4 = chained inside k_expand)     // statistics of k_cluster_dp by size class (whole chunk)
    unsigned long long sh_mini[64], sh_anchors[64];     // sharded sums
    uint32_t sh_host[64], sh_clusters[64], sh_pair[64];      // sh_pair: reads decided by the pair test
    uint32_t sh_pf_reads[64], sh_pf_dirty[64], sh_top[64];   // reads chained by par_fill_block / _tiled, their dirty anchors, reads settled by backtrack_block_top
    uint32_t sh_lemma[64];       // SH_F_CIGAR flag-only: reads decided inside a chaining kernel (top chain + chain_lemma)
    // long-read presets, flag-only: anchors pre-selected by locus (k_lr_locus) - reads listed per table size, reads with anchors dropped,
    // anchors kept, reads that must be redone with every anchor (lr_fb list)
    uint32_t n_locus[3], locus_ticket[3], lr_n_fb, lr_locus_reads; unsigned long long lr_locus_in, lr_locus_kept;
    uint32_t lr_fb_why[8], lr_fb_had, lr_pad, lr_probe_why[8];
    uint32_t lext_n_unres, lext_ticket_unres, lext_n_unres_in, lext_pad4, lext_n_exact, lext_ticket_exact, lext_rmq_open, lext_pad5;      // reads beyond the stage's second working-memory size: redone with memory allocated for them
    unsigned long long lext_slow2, lext_slow3, lext_slow_part[4], lext_sum_part[4], lext_clk_big[LR_NCLK], lext_d_big[8], lext_phase_max[LR_NCLK];
    uint32_t lext_started, lext_t0_done;      // SCRUBBY_HIP_DBG: the slowest read of the chains kernel (time << 32 | read length / read)
    unsigned long long stage_cursor;      // k_expand's raw anchors of the reads k_lr_locus will thin out (their own buffer: 12 B per anchor)
};
#define SHARD() ((blockIdx.x + (blockIdx.x >> 6)) & 63)

// wave-uniform bump allocator over a global counter: one atomic per `chunk` units
struct WaveAlloc {
    uint32_t cur = 0, end = 0;
    // n is wave-uniform; every lane calls; returns the first index of n consecutive units.
    // `lo`/`hi` receive the abandoned range of the previous chunk (to be invalidated by list users).
    __device__ inline uint32_t take(uint32_t *counter, uint32_t n, uint32_t chunk, uint32_t &lo, uint32_t &hi)
    {
        lo = hi = 0;
        if (cur + n > end) {
            lo = cur; hi = end;
            const uint32_t want = n > chunk ? n : chunk;
            uint32_t base = 0;
            if (lane_id() == 0) base = atomicAdd(counter, want);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);      // wave-uniform: stays in scalar registers
            cur = base; end = base + want;
        }
        const uint32_t r = cur;
        cur += n;
        return r;
    }
};

__device__ inline uint32_t prefix_popc(uint64_t mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
// wave-aggregated append; returns this lane's index in the list or ~0u
__device__ inline uint32_t wave_append(uint32_t *counter, bool pred)
{
    uint64_t mask = __ballot(pred);
    if (mask == 0) return ~0u;
    uint32_t base = 0;
    uint32_t leader = __ffsll((unsigned long long)mask) - 1;
    if (lane_id() == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
    base = (uint32_t)wave_bcast((int32_t)base, (int)leader);
    return pred ? base + prefix_popc(mask) : ~0u;
}

__device__ inline void write_trace(sh_trace *tr, uint64_t r, int32_t n_mini, int32_t n_seed, int32_t n_anchor, int32_t rep_len,
                                   int32_t rechained, int32_t n_chain, int32_t best, int32_t flag)
{
    if (!tr) return;
    int4 *p = (int4 *)(tr + r);
    p[0] = make_int4(n_mini, n_seed, n_anchor, rep_len);
    p[1] = make_int4(rechained, n_chain, best, flag);
    p[2] = make_int4(0, 0, 0, 0);      // n_aligned, n_regs, dp_max, sig: the extension stage fills them for reads with a chain
}

// ------------------------------------------------------------------------------------------------
// K1
// ------------------------------------------------------------------------------------------------
struct K1Args {
    const uint8_t *bases; const uint64_t *offsets; uint64_t n_reads, n_bases;
    const uint4 *slots; uint32_t lg_slots; int32_t k;
    uint4 *records; uint32_t seed_cap;
    uint32_t *k1info; uint8_t *flags; sh_trace *trace;
    uint32_t *work_small, *work_resketch, *work_big; Counters *ctr;
    uint32_t lds_words; int32_t mid_occ;
    uint32_t q_occ_max;     // reads with more minimizers than this need mm_seed_mz_flt: legacy path (UINT32_MAX: never)
};

// Slow path of K1: this lane's minimizer queue is full in the middle of a W-step block (tie-heavy,
// low-complexity reads).  Probe the lane's queued entries now, in order, so that seed records stay in
// query order; rare, divergent, deliberately not inlined.
__device__ __noinline__ uint4 k1_lane_flush(const uint64_t *list, uint64_t *prevk, uint32_t lane, uint32_t head, uint32_t tail, const uint4 *slots, uint32_t lg_slots,
                                            uint4 *rec, uint32_t seed_cap, uint4 acc /* n_seed, overflow | tandem << 1, sum_occ, n_high: by value, so the caller's stay in registers */,
                                            uint32_t mid_occ)
{
    const uint64_t slot_mask = (1ULL << lg_slots) - 1;
    for (uint32_t e = head; e != tail; ++e) {
        uint64_t m = list[(e & (K1_LIST_CAP - 1)) * 64 + lane];
        uint64_t key = m >> 18, idx = sh_slot_home(key, lg_slots);
        const uint32_t same = key == prevk[lane] ? SH_REC_PREV_SAME : 0u;
        prevk[lane] = key;
        uint4 sl = slots[idx];
        uint64_t w0 = (uint64_t)sl.y << 32 | sl.x;
        while (w0 != SH_SLOT_EMPTY && (w0 & SH_SLOT_KEYMASK) != key) { idx = (idx + 1) & slot_mask; sl = slots[idx]; w0 = (uint64_t)sl.y << 32 | sl.x; }
        if (w0 != SH_SLOT_EMPTY) {
            uint32_t occ = (w0 & SH_SLOT_MULTI) ? (sl.z & (uint32_t)SH_SLOT_NMASK) : 1u;
            if (acc.x < seed_cap) rec[acc.x] = make_uint4(sl.z, sl.w, occ | same, (uint32_t)m & 0x3ffffu);
            else acc.y |= 1;
            if (same) acc.y |= 2;
            ++acc.x;
            acc.z = acc.z + occ < acc.z ? 0xffffffffu : acc.z + occ;
            acc.w += occ > mid_occ;
        }
    }
    return acc;
}

template <int W>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_sketch_probe(K1Args a)
{
    extern __shared__ __align__(16) uint8_t smem[];
    uint64_t *list = (uint64_t *)smem;
    uint32_t *pk = (uint32_t *)(smem + (size_t)K1_LIST_CAP * 64 * 8);
    uint16_t *nm = (uint16_t *)(pk + a.lds_words + 2);
    // the hash of the minimizer each lane consumed last (mm_sketch order): a seed whose predecessor has the same hash is marked (SH_REC_PREV_SAME)
    uint64_t *prevk = (uint64_t *)(smem + (((size_t)K1_LIST_CAP * 64 * 8 + ((size_t)a.lds_words + 2) * 6 + 7) & ~(size_t)7));

    const uint32_t lane = threadIdx.x;
    const uint64_t tile = blockIdx.x, r0 = tile * 64, r = r0 + lane;
    const bool valid = r < a.n_reads;
    const uint64_t o_beg = a.offsets[valid ? r : a.n_reads], o_end = a.offsets[valid ? r + 1 : a.n_reads];
    const uint32_t len = (uint32_t)(o_end - o_beg);
    const uint64_t t_beg = a.offsets[r0], t_end = a.offsets[r0 + 64 < a.n_reads ? r0 + 64 : a.n_reads];
    const uintptr_t base_addr = (uintptr_t)a.bases;
    const uintptr_t a0 = (base_addr + t_beg) & ~(uintptr_t)15;
    const uint64_t span = (base_addr + t_end) - a0;

    if (span > (uint64_t)a.lds_words * 16) {      // tile does not fit the LDS stage: hand every read to K3
        uint32_t wi = wave_append(&a.ctr->n_resketch, valid);
        if (valid) a.work_resketch[wi] = (uint32_t)r;
        return;
    }

    // ---- A: stage -------------------------------------------------------------------------------
    const uint32_t n_chunks = (uint32_t)((span + 15) >> 4);
    for (uint32_t c = lane; c < n_chunks + 2; c += 64) {
        uint32_t codes = 0, amb = 0;
        if (c < n_chunks) {
            const uintptr_t p = a0 + (uintptr_t)c * 16;
            uint32_t wds[4];
            if (p >= base_addr && p + 16 <= base_addr + a.n_bases) {
                uint4 v = *(const uint4 *)p;
                wds[0] = v.x; wds[1] = v.y; wds[2] = v.z; wds[3] = v.w;
            } else {
                for (int q = 0; q < 4; ++q) {
                    uint32_t wv = 0;
                    for (int b = 0; b < 4; ++b) {
                        uintptr_t pb = p + q * 4 + b;
                        uint32_t ch = (pb >= base_addr && pb < base_addr + a.n_bases) ? *(const uint8_t *)pb : (uint32_t)'N';
                        wv |= ch << (8 * b);
                    }
                    wds[q] = wv;
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    uint32_t ch = (wds[q] >> (8 * b)) & 0xffu;
                    uint32_t idx = (ch & 0xDFu) - 0x41u;
                    uint32_t ok = idx < 32u ? (0x00180045u >> idx) & 1u : 0u;   // A C G T U
                    codes |= (((ch >> 1) ^ (ch >> 2)) & 3u) << (2 * (q * 4 + b));
                    amb |= (ok ^ 1u) << (q * 4 + b);
                }
            }
        } else amb = 0xffffu;
        pk[c] = codes;
        nm[c] = (uint16_t)amb;
    }
    __syncthreads();

    // ---- B + C ----------------------------------------------------------------------------------
    SketchPacked<W> st;      // reads of this kernel are <= 1024 bases and k <= 23: one 64-bit word per ring entry (sh_sketch.h)
    st.init(a.k);
    const uint32_t b0 = (uint32_t)((base_addr + o_beg) - a0);    // tile-relative index of this read's first base
    uint32_t maxlen = len;
    maxlen = wave_all_max_u32(maxlen);

    // The minimizer queue is a ring of K1_LIST_CAP entries per lane in LDS: [qh, qt), of which the first `pend` have their
    // home-slot gathers in flight (sl[]).  Every W-step block starts by issuing up to 4 gathers per lane for what earlier blocks
    // queued and ends by consuming them, so a lane's HBM latency is covered by its own sketch arithmetic, not only by the
    // other waves of the SIMD.  Seed records stay in query order: the ring is FIFO.
    uint32_t qh = 0, qt = 0, pend = 0, n_mini = 0, n_seed = 0, overflow = 0, sum_occ = 0, n_high = 0, tand = 0;
    prevk[lane] = ~0ull;
    const uint64_t slot_mask = (1ULL << a.lg_slots) - 1;
    uint4 *rec = a.records + (size_t)(valid ? r : 0) * a.seed_cap;      // per-read contiguous seed records
    auto emit = [&](uint64_t packed) {
        if (qt - qh >= K1_LIST_CAP) {      // tie-heavy read: the lane drains its whole ring now, in order; gathers in flight are dropped
            const uint4 acc = k1_lane_flush(list, prevk, lane, qh, qt, a.slots, a.lg_slots, rec, a.seed_cap, make_uint4(n_seed, overflow, sum_occ, n_high), (uint32_t)a.mid_occ);
            n_seed = acc.x; overflow = acc.y & 1u; tand |= acc.y >> 1; sum_occ = acc.z; n_high = acc.w;
            qh = qt; pend = 0;
        }
        list[(qt & (K1_LIST_CAP - 1)) * 64 + lane] = sh_packed_entry(packed);      // hash << 18 | pos << 1 | strand
        ++qt; ++n_mini;
    };
    auto consume = [&](const uint4 (&sl)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if ((uint32_t)u < pend) {
                const uint64_t m = list[((qh + u) & (K1_LIST_CAP - 1)) * 64 + lane];
                const uint64_t key = m >> 18;
                const uint32_t same = key == prevk[lane] ? SH_REC_PREV_SAME : 0u;
                prevk[lane] = key;
                uint4 v = sl[u];
                uint64_t w0 = (uint64_t)v.y << 32 | v.x;
                if (w0 != SH_SLOT_EMPTY && (w0 & SH_SLOT_KEYMASK) != key) {
                    uint64_t idx = sh_slot_home(key, a.lg_slots);
                    do { idx = (idx + 1) & slot_mask; v = a.slots[idx]; w0 = (uint64_t)v.y << 32 | v.x; } while (w0 != SH_SLOT_EMPTY && (w0 & SH_SLOT_KEYMASK) != key);
                }
                if (w0 != SH_SLOT_EMPTY) {
                    const uint32_t occ = (w0 & SH_SLOT_MULTI) ? (v.z & (uint32_t)SH_SLOT_NMASK) : 1u;
                    if (n_seed < a.seed_cap) rec[n_seed] = make_uint4(v.z, v.w, occ | same, (uint32_t)m & 0x3ffffu);
                    else overflow = 1;
                    tand |= same;
                    ++n_seed;
                    sum_occ = sum_occ + occ < sum_occ ? 0xffffffffu : sum_occ + occ;
                    n_high += occ > (uint32_t)a.mid_occ;
                }
            }
        }
        qh += pend; pend = 0;
    };
    auto issue = [&](uint4 (&sl)[4]) {
        // straight-line: every lane issues its 4 gathers back to back; a lane with fewer queued entries points the spare ones
        // at slot 0 (one cached line for the whole device), so no exec-mask branch separates the loads
        const uint32_t n = min(qt - qh, 4u);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool on = (uint32_t)u < n;
            const uint64_t m = list[((qh + (on ? (uint32_t)u : 0u)) & (K1_LIST_CAP - 1)) * 64 + lane];
            const uint64_t idx = on ? sh_slot_home(m >> 18, a.lg_slots) : 0ull;
            sl[u] = a.slots[idx];
        }
        pend = n;
    };

    uint32_t codes = 0, amb = 0;
    uint32_t i0 = 0;
    for (;;) {
        // ---- C, first half: gathers for what the previous blocks queued; they fly during this block's sketch (issued and
        // consumed inside one iteration, so no gather result is carried around the loop)
        uint4 sl[4];
        issue(sl);          // unconditional: no merge of old and new gather registers, so no wait lands right behind the loads
        if (i0 < maxlen) {
            // W steps with compile-time ring slots
            auto one = [&](auto Pc) {
                constexpr int P = decltype(Pc)::value;
                const uint32_t i = i0 + P;
                if ((i & 15u) == 0) {       // wave-uniform refill of the next 16 bases
                    uint32_t g = b0 + i;
                    uint32_t wi = min(g >> 4, a.lds_words);
                    uint32_t sh = g & 15u;
                    uint64_t two = (uint64_t)pk[wi + 1] << 32 | pk[wi];
                    codes = (uint32_t)(two >> (2 * sh));
                    uint32_t twon = (uint32_t)nm[wi + 1] << 16 | nm[wi];
                    amb = (twon >> sh) & 0xffffu;
                }
                if (i < len) {
                    uint32_t c = (amb & 1u) ? 4u : (codes & 3u);
                    st.template step<P>(c, i, emit, emit);
                }
                codes >>= 2; amb >>= 1;
            };
            [&]<int... Ps>(std::integer_sequence<int, Ps...>) { (one(std::integral_constant<int, Ps>{}), ...); }
            (std::make_integer_sequence<int, W>{});
            st.block_end();
            i0 += W;
        }
        const bool last = i0 >= maxlen;
        if (last && valid && len > 0) st.finish(emit);
        // ---- C, second half ----
        if (__ballot(pend != 0) != 0) consume(sl);
        if (last) {
            while (__ballot(qt != qh) != 0) { uint4 sl2[4]; issue(sl2); consume(sl2); }
            break;
        }
    }

    // ---- per-read result ------------------------------------------------------------------------
    if (valid) a.k1info[r] = n_mini | (n_seed & 0x7fffu) << 16 | (tand ? 1u << 31 : 0u);      // bit 31: some seed of the read is tandem (K1 reads have < 2^15 seeds)
    // reads whose seeds need no occurrence filtering and give <= K2_CAP anchors are chained by K2 (lane per read);
    // everything else goes to the repeat path (wave per read)
    if (n_mini > a.q_occ_max) overflow = 1;
    const bool to_k3 = valid && overflow != 0;
    const bool done = valid && !overflow && n_seed == 0;
    const bool simple = n_high == 0 && sum_occ <= K2_CAP;
    const bool to_k2 = valid && !overflow && n_seed > 0 && simple;
    const bool to_big = valid && !overflow && n_seed > 0 && !simple;
    if (done) {
        int32_t fl = len == 0 ? 2 : 0;
        a.flags[r] = (uint8_t)fl;
        write_trace(a.trace, r, (int32_t)n_mini, 0, 0, 0, 0, 0, 0, fl);
    }
    uint32_t wi = wave_append(&a.ctr->n_small, to_k2);
    if (to_k2) a.work_small[wi] = (uint32_t)r;
    wi = wave_append(&a.ctr->n_big[0], to_big);
    if (to_big) a.work_big[wi] = (uint32_t)r;
    wi = wave_append(&a.ctr->n_resketch, to_k3);
    if (to_k3) a.work_resketch[wi] = (uint32_t)r;
    // statistics (sharded)
    const uint32_t msum = (uint32_t)wave_all_add((int32_t)n_mini);
    if (lane == 0) atomicAdd(&a.ctr->sh_mini[SHARD()], (unsigned long long)msum);
}

// ---- wave helpers ------------------------------------------------------------------------------------
__device__ inline uint32_t rdlane(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ inline uint32_t wave_sum_u32(uint32_t v)
{
    return (uint32_t)wave_all_add((int32_t)v);
}
__device__ inline uint32_t wave_excl_scan_u32(uint32_t v, uint32_t lane)
{
    return (uint32_t)wave_scan_add_incl((int32_t)v) - v;
}

// ------------------------------------------------------------------------------------------------
// long reads (any length): segment-parallel sketch, then per read thinning screen + probe + compaction
// ------------------------------------------------------------------------------------------------
#define LSEG 256u          // bases per sketch segment (plus w + k warm-up)

struct LongArgs {
    const uint8_t *bases; const uint64_t *offsets; uint64_t n_reads, n_bases;
    const uint4 *slots; uint32_t lg_slots; int32_t k;
    uint32_t *seg_base;            // n_reads + 1: first segment of each read
    uint32_t *seg_cnt;             // minimizers per segment
    unsigned long long *seg_off;   // exclusive scan of seg_cnt (n_segs + 1)
    unsigned long long *scan_tot;  // k_long_scan: one sum per block of 16 384 segments
    uint64_t *mz_hash; uint32_t *mz_y; unsigned long long mz_cap;
    uint4 *lrec;                   // seed records, compacted per read in place of its minimizers
    unsigned long long *seed_off;  // per read
    uint32_t *k1info; uint8_t *flags; sh_trace *trace;
    uint32_t *work_big, *work_resketch; Counters *ctr;
    int32_t mid_occ; uint32_t q_occ_max; float q_occ_frac;
    uint32_t *n_segs_out; uint32_t max_segs;
};

__global__ __launch_bounds__(64) void k_long_segtable(LongArgs a)
{
    const uint32_t lane = threadIdx.x;
    uint32_t run = 0;
    for (uint64_t base = 0; base < a.n_reads; base += 64) {
        const uint64_t r = base + lane;
        uint32_t t = r < a.n_reads ? (uint32_t)((a.offsets[r + 1] - a.offsets[r] + LSEG - 1) / LSEG) : 0;
        // a chunk with more bases than the context was sized for: the reads past the segment table get no segments
        // (k_long_probe sends them to the legacy path)
        if (__ballot((uint64_t)run + wave_excl_scan_u32(t, lane) + t > a.max_segs) != 0) {
            uint32_t acc = run;
            for (uint32_t l = 0; l < 64; ++l) {
                uint32_t tl = rdlane(t, l);
                if ((uint64_t)acc + tl > a.max_segs) tl = 0;
                if (lane == l) t = tl;
                acc += tl;
            }
        }
        const uint32_t ex = wave_excl_scan_u32(t, lane);
        if (r < a.n_reads) a.seg_base[r] = run + ex;
        run += wave_sum_u32(t);
    }
    if (lane == 0) { a.seg_base[a.n_reads] = run; *a.n_segs_out = run; }
}

template <int W, bool EMIT>
__global__ __launch_bounds__(256) void k_long_sketch(LongArgs a)
{
    const uint32_t n_segs = *a.n_segs_out;
    for (uint32_t seg = blockIdx.x * 256 + threadIdx.x; seg < n_segs; seg += gridDim.x * 256) {
        uint32_t lo = 0, hi = (uint32_t)a.n_reads;
        while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (a.seg_base[mid] <= seg) lo = mid; else hi = mid; }
        const uint32_t r = lo;
        const uint64_t o_beg = a.offsets[r];
        const uint32_t len = (uint32_t)(a.offsets[r + 1] - o_beg);
        const uint32_t start = (seg - a.seg_base[r]) * LSEG, end = start + LSEG < len ? start + LSEG : len;
        const uint32_t warm = (uint32_t)(W + a.k), from = start > warm ? start - warm : 0;
        const uint8_t *seq = a.bases + o_beg;
        SketchState<W> st;
        st.init(a.k);
        uint32_t n = 0, cur = 0;
        const unsigned long long o0 = EMIT ? a.seg_off[seg] : 0;
        auto emit = [&](uint64_t x, uint32_t y) {
            if (cur < start) return;
            if (EMIT && o0 + n < a.mz_cap) { a.mz_hash[o0 + n] = x >> 8; a.mz_y[o0 + n] = y; }
            ++n;
        };
        // the lane's bases 16 at a time (one aligned 128-bit load per 16 steps): byte loads made every lane touch its cache line 128 times,
        // and with 64 lines per wave load in flight the lines were evicted from L2 before they were used up (PMC: 42 B fetched per base)
        const uintptr_t b_lo = (uintptr_t)a.bases, b_hi = b_lo + a.n_bases;
        uintptr_t blk = ~(uintptr_t)0;
        uint64_t w_lo = 0, w_hi = 0;
        auto base_at = [&](uint32_t i) -> uint32_t {
            const uintptr_t p = (uintptr_t)(seq + i), pb = p & ~(uintptr_t)15;
            if (pb != blk) {
                blk = pb;
                if (pb >= b_lo && pb + 16 <= b_hi) { const uint4 v = *(const uint4 *)pb; w_lo = (uint64_t)v.y << 32 | v.x; w_hi = (uint64_t)v.w << 32 | v.z; }
                else {
                    w_lo = w_hi = 0;
                    for (int t = 0; t < 16; ++t) {
                        const uintptr_t q = pb + t;
                        const uint64_t ch = (q >= b_lo && q < b_hi) ? *(const uint8_t *)q : (uint64_t)'N';
                        if (t < 8) w_lo |= ch << (8 * t); else w_hi |= ch << (8 * (t - 8));
                    }
                }
            }
            const uint32_t b = (uint32_t)(p & 15);
            return (uint32_t)((b < 8 ? w_lo : w_hi) >> (8 * (b & 7))) & 0xffu;
        };
        for (uint32_t i0 = from; i0 < end; i0 += W) {
            auto one = [&](auto Pc) {
                constexpr int Pk = decltype(Pc)::value;
                const uint32_t i = i0 + Pk;
                if (i < end) { cur = i; st.template step<Pk>(sh_nt4((uint8_t)base_at(i)), i, emit); }
            };
            [&]<int... Ps>(std::integer_sequence<int, Ps...>) { (one(std::integral_constant<int, Ps>{}), ...); }
            (std::make_integer_sequence<int, W>{});
        }
        if (end == len) { cur = end; st.finish(emit); }
        if (!EMIT) a.seg_cnt[seg] = n;
    }
}

// exclusive scan of seg_cnt[0..n] into seg_off[0..n] (n = n_segs, so seg_off[n] = total): one block, sequential chunks
// exclusive scan of seg_cnt[0..n] into seg_off[0..n] in three launches: (0) every block sums its 16 384 counts, (1) one block scans the
// block sums, (2) every block scans its counts from its base.  A single block walking the whole table took 8 ms per launch of 200 k long reads.
#define LSCAN_PER 16u
__global__ __launch_bounds__(1024) void k_long_scan(LongArgs a, int mode)
{
    __shared__ unsigned long long s_wave[16];
    const uint32_t n = *a.n_segs_out, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t n_blocks = n / (1024u * LSCAN_PER) + 1;
    if (mode == 1) {       // one block: exclusive scan of the block sums in place (n_blocks <= a few hundred)
        unsigned long long run = 0;
        for (uint32_t base = 0; base < n_blocks; base += 1024) {
            const uint32_t i = base + tid;
            const unsigned long long v = i < n_blocks ? a.scan_tot[i] : 0ull;
            unsigned long long inc = wave_scan_add_incl_u64(v);
            if (lane == 63) s_wave[wv] = inc;
            __syncthreads();
            unsigned long long pre = run;
            for (uint32_t q = 0; q < wv; ++q) pre += s_wave[q];
            if (i < n_blocks) a.scan_tot[i] = pre + inc - v;
            unsigned long long all = 0;
            for (int q = 0; q < 16; ++q) all += s_wave[q];
            run += all;
            __syncthreads();
        }
        return;
    }
    for (uint32_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        const uint32_t i0 = blk * 1024u * LSCAN_PER + tid * LSCAN_PER;
        uint32_t v[LSCAN_PER], tot = 0;
#pragma unroll
        for (uint32_t t = 0; t < LSCAN_PER; ++t) { v[t] = i0 + t < n ? a.seg_cnt[i0 + t] : 0; tot += v[t]; }
        const uint32_t ex = wave_excl_scan_u32(tot, lane);
        const uint32_t wtot = wave_sum_u32(tot);
        if (lane == 0) s_wave[wv] = wtot;
        __syncthreads();
        if (mode == 0) {
            if (tid == 0) { unsigned long long t = 0; for (int q = 0; q < 16; ++q) t += s_wave[q]; a.scan_tot[blk] = t; }
        } else {
            unsigned long long pre = a.scan_tot[blk] + ex;
            for (uint32_t q = 0; q < wv; ++q) pre += s_wave[q];
#pragma unroll
            for (uint32_t t = 0; t < LSCAN_PER; ++t) { if (i0 + t <= n) a.seg_off[i0 + t] = pre; pre += v[t]; }
        }
        __syncthreads();
    }
}

// one wave per read: mm_seed_mz_flt screen, probes, compaction of the hits, routing
__global__ __launch_bounds__(64) void k_long_probe(LongArgs a)
{
    constexpr uint32_t LT_CAP = 2048;          // distinct hashes of the over-full bins of one read (4096: 56 KB of LDS, two waves per CU)
    __shared__ uint16_t s_cnt[4096];
    __shared__ uint64_t s_tkey[LT_CAP];
    __shared__ uint32_t s_tcnt[LT_CAP], s_tover;
    const uint32_t lane = threadIdx.x;
    for (uint64_t r = blockIdx.x; r < a.n_reads; r += gridDim.x) {
        const uint32_t len = (uint32_t)(a.offsets[r + 1] - a.offsets[r]);
        const unsigned long long base = a.seg_off[a.seg_base[r]], top = a.seg_off[a.seg_base[r + 1]];
        unsigned long long n_mini = top - base;
        bool legacy = top > a.mz_cap || n_mini >= 65536;         // out of room, or counts that do not fit k1info
        legacy |= a.seg_base[r + 1] - a.seg_base[r] != (len + LSEG - 1) / LSEG;
        if (!legacy && n_mini > a.q_occ_max) {
            // mm_seed_mz_flt drops hashes repeated within the query more than mid_occ times and more
            // than q_occ_frac of all minimizers.  Screen: per-bin counts of the hash's low 12 bits bound the true counts
            // from above.  Reads with a bin above mid_occ (satellite reads) count the hashes of those bins exactly in an
            // LDS table and compact the survivors in place.
            for (uint32_t i = lane; i < 4096; i += 64) s_cnt[i] = 0;
            __syncthreads();
            for (unsigned long long i = lane; i < n_mini; i += 64) {
                uint32_t bin = (uint32_t)a.mz_hash[base + i] & 4095u;
                atomicAdd((unsigned int *)&s_cnt[bin & ~1u], (bin & 1u) ? 0x10000u : 1u);      // two 16-bit counters per dword
            }
            __syncthreads();
            uint32_t mx = 0;
            for (uint32_t i = lane; i < 4096; i += 64) mx = s_cnt[i] > mx ? s_cnt[i] : mx;
            mx = wave_all_max_u32(mx);
            if (mx > (uint32_t)a.mid_occ) {
                for (uint32_t i = lane; i < LT_CAP; i += 64) { s_tkey[i] = ~0ull; s_tcnt[i] = 0; }
                if (lane == 0) s_tover = 0;
                __syncthreads();
                for (unsigned long long i = lane; i < n_mini; i += 64) {
                    const uint64_t key = a.mz_hash[base + i];
                    if (s_cnt[(uint32_t)key & 4095u] <= (uint32_t)a.mid_occ) continue;
                    uint32_t slot = (uint32_t)((key * 0x9E3779B97F4A7C15ULL) >> 52) & (LT_CAP - 1);
                    for (uint32_t step = 0;; ++step) {
                        if (step >= LT_CAP / 2) { s_tover = 1; break; }      // table (nearly) full: legacy path
                        unsigned long long prev = atomicCAS((unsigned long long *)&s_tkey[slot], ~0ull, (unsigned long long)key);
                        if (prev == ~0ull || prev == key) { atomicAdd(&s_tcnt[slot], 1u); break; }
                        slot = (slot + 1) & (LT_CAP - 1);
                    }
                }
                __syncthreads();
                if (s_tover) {
                    legacy = true;
                    if (lane == 0) atomicAdd(&a.ctr->n_leg_reason[1], 1u);
                } else {
                    const float lim = (float)n_mini * a.q_occ_frac;
                    unsigned long long kept = 0;
                    for (unsigned long long t0 = 0; t0 < n_mini; t0 += 64) {
                        const bool have = t0 + lane < n_mini;
                        uint64_t key = 0; uint32_t y = 0; bool keep = have;
                        if (have) {
                            key = a.mz_hash[base + t0 + lane]; y = a.mz_y[base + t0 + lane];
                            if (s_cnt[(uint32_t)key & 4095u] > (uint32_t)a.mid_occ) {
                                uint32_t slot = (uint32_t)((key * 0x9E3779B97F4A7C15ULL) >> 52) & (LT_CAP - 1);
                                while (s_tkey[slot] != key) slot = (slot + 1) & (LT_CAP - 1);      // present: inserted above
                                const uint32_t cnt = s_tcnt[slot];
                                keep = !(cnt > (uint32_t)a.mid_occ && (float)cnt > lim);
                            }
                        }
                        const uint64_t km = __ballot(keep);
                        if (keep) { a.mz_hash[base + kept + prefix_popc(km)] = key; a.mz_y[base + kept + prefix_popc(km)] = y; }
                        kept += (unsigned long long)__popcll(km);
                    }
                    n_mini = kept;
                }
            }
            __syncthreads();
        }
        if (legacy || len == 0) {
            if (lane == 0) { uint32_t i = atomicAdd(&a.ctr->n_resketch, 1u); a.work_resketch[i] = (uint32_t)r; a.k1info[r] = 0; a.seed_off[r] = 0; }      // no seed records of this read here
            continue;
        }
        const uint64_t slot_mask = (1ULL << a.lg_slots) - 1;
        uint32_t n_seed = 0;
        // four rounds of 64 home-slot gathers in flight per wave (the 56 KB of LDS above leave two waves per CU: one round at a time left the
        // probes latency-bound at 11.6 G/s); the records are written in minimizer order all the same
        for (unsigned long long t0 = 0; t0 < n_mini; t0 += 256) {
            uint64_t key[4]; uint32_t y[4]; uint4 sl[4]; bool have[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned long long t = t0 + 64ull * u + lane;
                have[u] = t < n_mini;
                key[u] = have[u] ? a.mz_hash[base + t] : 0ull;
                y[u] = have[u] ? a.mz_y[base + t] : 0u;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) sl[u] = a.slots[have[u] ? sh_slot_home(key[u], a.lg_slots) : 0ull];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                uint64_t w0 = SH_SLOT_EMPTY;
                if (have[u]) {
                    uint64_t idx = sh_slot_home(key[u], a.lg_slots);
                    w0 = (uint64_t)sl[u].y << 32 | sl[u].x;
                    while (w0 != SH_SLOT_EMPTY && (w0 & SH_SLOT_KEYMASK) != key[u]) { idx = (idx + 1) & slot_mask; sl[u] = a.slots[idx]; w0 = (uint64_t)sl[u].y << 32 | sl[u].x; }
                }
                const bool hit = have[u] && w0 != SH_SLOT_EMPTY;
                const uint64_t hm = __ballot(hit);
                if (hit) {
                    const uint32_t occ = (w0 & SH_SLOT_MULTI) ? (sl[u].z & (uint32_t)SH_SLOT_NMASK) : 1u;
                    const unsigned long long t = t0 + 64ull * u + lane;
                    const uint32_t same = (t > 0 && a.mz_hash[base + t - 1] == key[u]) ? SH_REC_PREV_SAME : 0u;      // after mm_seed_mz_flt's compaction
                    a.lrec[base + n_seed + prefix_popc(hm)] = make_uint4(sl[u].z, sl[u].w, occ | same, y[u]);
                }
                n_seed += (uint32_t)__popcll(hm);
            }
        }
        if (lane == 0) {
            a.k1info[r] = (uint32_t)n_mini | n_seed << 16;
            a.seed_off[r] = base;
            atomicAdd(&a.ctr->sh_mini[SHARD()], (unsigned long long)n_mini);
            if (n_seed == 0) {
                a.flags[r] = 0;
                write_trace(a.trace, r, (int32_t)n_mini, 0, 0, 0, 0, 0, 0, 0);
            } else {
                uint32_t i = atomicAdd(&a.ctr->n_big[0], 1u);
                a.work_big[i] = (uint32_t)r;
            }
        }
    }
}

// route every read of the batch to K3 (k > 23 or reads too long for the LDS stage)
__global__ void k_route_all(uint64_t n_reads, uint32_t *work_resketch, Counters *ctr)
{
    uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n_reads) work_resketch[r] = (uint32_t)r;
    if (r == 0) ctr->n_resketch = (uint32_t)n_reads;
}


// ------------------------------------------------------------------------------------------------
// K2 / K3
// ------------------------------------------------------------------------------------------------
struct BigMeta { uint32_t r, n_a; int32_t rep_len; uint32_t state; };   // state: 0 expanded, 1 deferred
struct SortItem { uint32_t w, n, qlen, pad; unsigned long long off; };
// The class tables once more, in HBM: a kernel that indexes an array INSIDE its argument struct with a run-time value (the sort class a
// read falls into, the size class of a cluster) gets the whole struct copied to scratch, and every later a.X becomes a scratch load - the
// ISA of k_cluster_dp had five of them per chunk of predecessors.  Run-time indices go through BigBufs::tabs; compile-time ones keep the copy.
struct BigTables { struct SortItem *sort_items[N_SORT_CLS]; struct SortItem *cl_items[4]; uint32_t cl_cap[4]; };
struct BigBufs {        // the repeat path's slice of the arena (all arrays indexed by anchor slot)
    uint64_t *ax, *bx, *az; uint32_t *aq, *bq; int32_t *af;   // anchors (+ sort ping-pong), DP state f and (p,t)
    uint64_t *hz;                                             // SH_F_CIGAR: backtrack heap of clusters chained from LDS (the anchors must survive for the hand-over)
    unsigned long long anchor_cap;
    BigMeta *meta; int32_t *acc_nu, *acc_best;
    SortItem *sort_items[N_SORT_CLS];
    uint32_t *tile_base, *tile_split;       // giant reads: tile table and merge-path splits
    uint32_t *giant_order;                  // giant reads by falling size class (k_giant_scan): k_giant_chain draws the largest first
    SortItem *cl_items[4]; uint32_t cl_cap[4];   // big clusters of giant reads (w, n, qlen, pad = buffer, off = first anchor slot)
    const BigTables *tabs;                       // the three arrays above in HBM, for run-time indices
};

struct K2Args {
    const uint64_t *offsets; const uint8_t *bases; uint64_t n_reads;
    const uint4 *slots; uint32_t lg_slots; int32_t w;
    const uint64_t *positions;
    uint4 *records; uint32_t seed_cap;
    const uint32_t *k1info; uint8_t *flags; sh_trace *trace;
    const uint32_t *work; const uint32_t *work_count;        // input list
    uint32_t *work_big; uint32_t *work_defer;                // outputs
    Counters *ctr;
    uint8_t *arena; unsigned long long arena_bytes;
    ChainParams P;
    uint32_t work_begin; // k_chain_large: first list entry to process
    uint32_t *leftover; uint32_t *leftover_count;   // k_pair_pass: reads it leaves undecided
    ChainSink sink; int32_t emit;                   // SH_F_CIGAR: every kept chain is handed to the extension stage; no flag-only shortcut in the DP
    BaseCtx BC;
    int32_t quiet;      // k_chain_large re-running reads that were counted before: no statistics
};

// hi word (strand | contig) of anchor group g of a read chained in a SmallStore: the store keeps group ranks only, so the
// g-th smallest distinct x >> 32 is recomputed from the seeds (<= K2_CAP anchors)
__device__ inline uint32_t small_group_hi(SeedView sv, const uint64_t *__restrict__ positions, int32_t qlen, int32_t k, uint32_t g)
{
    long long prev = -1;
    for (uint32_t round = 0; round <= g; ++round) {
        long long cur = 1ll << 40;
        for (uint32_t i = 0; i < sv.n; ++i) {
            const uint4 sd = sv.get(i);
            if (sd.z >> 31) continue;
            const uint64_t w1 = (uint64_t)sd.y << 32 | sd.x;
            const uint32_t sd_n = sd.z & SH_REC_OCC_MASK;
            for (uint32_t t = 0; t < sd_n; ++t) {
                const uint64_t rp = sd_n == 1 ? w1 : positions[(w1 >> SH_SLOT_NBITS) + t];
                uint64_t x; uint32_t q;
                make_anchor(rp, sd.w, qlen, k, x, q);
                const long long hi = (long long)(x >> 32);
                if (hi > prev && hi < cur) cur = hi;
            }
        }
        prev = cur;
    }
    return (uint32_t)prev;
}

__device__ inline void finish_read(const K2Args &a, uint32_t r, int32_t n_mini, int32_t n_seed, int64_t n_a, int32_t rep_len,
                                   int32_t rechained, int32_t n_u, int32_t best)
{
    int32_t fl = n_u > 0;
    a.flags[r] = (uint8_t)fl;
    write_trace(a.trace, r, n_mini, n_seed, (int32_t)n_a, rep_len, rechained, n_u, best, fl);
}

template <int CAP>
__global__ __launch_bounds__(64) void k_chain_small(K2Args a)
{
    __shared__ uint32_t s_lo[CAP * 64];
    __shared__ uint32_t s_aux[CAP * 64];
    __shared__ uint16_t s_q[CAP * 64];
    __shared__ uint8_t s_g[CAP * 64];
    const uint32_t lane = threadIdx.x;
    const uint32_t n_work = *a.work_count;
    SmallStore<CAP> S;
    S.lo = s_lo + lane; S.aux = s_aux + lane; S.qv = s_q + lane; S.gv = s_g + lane;
    uint32_t n_host_wave = 0, n_lemma_wave = 0;

    for (uint32_t base = blockIdx.x * 64; base < n_work; base += gridDim.x * 64) {
        const uint32_t wi = base + lane;
        const bool valid = wi < n_work;
        bool host = false, lemma_hit = false, decided = false, need_mid = false, tried = false;
        MidReq mid{0, 0, 0, 0, 0};
        uint32_t r_ = 0; int32_t qlen_ = 0, n_mini_ = 0, n_seed_ = 0, n_u_ = 0, best_ = 0; int64_t n_a_ = 0; SeedView sv_{nullptr, 1, 0};
        if (valid) {
            const uint32_t r = a.work[wi];
            const uint32_t info = a.k1info[r];
            const int32_t n_mini = (int32_t)(info & 0xffffu), n_seed = (int32_t)(info >> 16 & 0x7fffu);
            const int32_t qlen = (int32_t)(a.offsets[r + 1] - a.offsets[r]);
            SeedView sv;
            sv.base = a.records + (size_t)r * a.seed_cap;
            sv.stride = 1; sv.n = (uint32_t)n_seed;
            // K1 only sends reads here whose seeds all pass the occurrence filter (none above mid_occ) and expand
            // to <= CAP anchors: no filtering, rep_len = 0, never re-chained
            int32_t n_u = 0, best = 0;
            gen_anchors(S, sv, a.positions, qlen, a.P.k);
            int64_t n_a = 0;
            for (uint32_t i = 0; i < sv.n; ++i) n_a += sv.occ(i);
            if (a.emit && a.trace == nullptr && a.P.ext_lemma) {      // flag-only: the top chain, vouched for by chain_lemma, decides without a hand-over
                chain_dp_mask(S, (int)n_a, qlen, a.P);
                BestChain bc{};
                auto hi = [&](int32_t i) { return small_group_hi(sv, a.positions, qlen, a.P.k, S.grp(i)); };
                const BestEmit<SmallStore<CAP>, decltype(hi)> be{&S, &bc, region_hash(qlen), a.P.k, 0u, hi, true, TandemQ{sv.base, sv.n, sv.stride, qlen, info >> 31}};
                backtrack_mask(S, (int)n_a, a.P, n_u, best, false, be);
                tried = true;
                if (n_u == 0) decided = true;
                else if (!bc.tie) { const int32_t code = chain_lemma(S, bc.zi, bc.end_i, a.P, hi(bc.zi), mid); decided = code == 1; need_mid = code == 2; }
            }
            r_ = r; qlen_ = qlen; n_a_ = n_a; n_mini_ = n_mini; n_seed_ = n_seed; n_u_ = n_u; best_ = best; sv_ = sv;
        }
        // stretches with too many bases outside their k-mers: mm_test_zdrop on the bases, the wave on one lane's request at a time
        if (a.emit && a.trace == nullptr && a.P.ext_lemma) {
            const bool mid_ok = resolve_mid_wave(need_mid, mid, a.bases + a.offsets[valid ? r_ : 0], qlen_, a.BC, a.P);
            if (need_mid && mid_ok) decided = true;
        }
        if (valid) {
            const uint32_t r = r_;
            const int32_t qlen = qlen_, n_mini = n_mini_, n_seed = n_seed_;
            const int64_t n_a = n_a_;
            int32_t n_u = n_u_, best = best_;
            SeedView sv = sv_;
            lemma_hit = tried && n_u > 0 && decided;
            if (decided) {}
            else if (a.emit) {      // every chain goes to the extension stage, which decides the read
                if (!tried) chain_dp_mask(S, (int)n_a, qlen, a.P);
                auto emf = [&](int64_t zi, int64_t end_i, int32_t sc, int64_t cnt, int32_t zf) {
                    const uint32_t hi = small_group_hi(sv, a.positions, qlen, a.P.k, S.grp((int)zi));
                    sink_emit(a.sink, r, (int32_t)zi, (int32_t)end_i, sc, (uint32_t)cnt, (uint32_t)zf, (uint32_t)zi, a.P.k, region_hash(qlen), qlen,
                              [&](int32_t i, uint64_t &x, uint32_t &q) { x = (uint64_t)hi << 32 | S.rlo(i); q = S.qp(i); },
                              [&](int32_t i) { return S.Pm(i); });
                };
                struct Em { decltype(emf) &f; const ChainSink &sk; uint32_t r;
                            __device__ void operator()(int64_t a1, int64_t a2, int32_t a3, int64_t a4, int32_t a5) const { f(a1, a2, a3, a4, a5); }
                            __device__ bool done(int32_t zf) const { return sk.best && zf < sink_best_score(sk, r); } };
                const Em em{emf, a.sink, r};
                backtrack_mask(S, (int)n_a, a.P, n_u, best, false, em);
            } else if (a.trace == nullptr && a.P.flag_stop != INT32_MAX) n_u = chain_dp_mask(S, (int)n_a, qlen, a.P, a.P.flag_stop) ? 1 : 0;     // flag-only: see ChainParams::flag_stop
            else {
                chain_dp_mask(S, (int)n_a, qlen, a.P);
                backtrack_mask(S, (int)n_a, a.P, n_u, best, a.trace == nullptr);
            }
            finish_read(a, r, n_mini, n_seed, n_a, 0, 0, n_u, best);
            host = n_u > 0;
        }
        n_host_wave += (uint32_t)__popcll(__ballot(host));
        n_lemma_wave += (uint32_t)__popcll(__ballot(lemma_hit));
    }
    if (lane == 0 && n_host_wave) atomicAdd(&a.ctr->sh_host[SHARD()], n_host_wave);
    if (lane == 0 && n_lemma_wave) atomicAdd(&a.ctr->sh_lemma[SHARD()], n_lemma_wave);
}

// Pair pass (flag-only; ChainParams::pair_dq_*), one lane per read over a work list.  Two singleton seeds (their position words
// are in the seed records: no gather) whose anchors lie on one diagonal pair_dq_min..pair_dq_max apart decide a read as mapped,
// because mg_lchain_dp's look-back from the later anchor must then score the earlier one - provided at most max_skip - 1 anchors
// can sort between the two:
//   distinct == 0  reads of the LDS path (no filtered seed): true whenever the read has at most max_skip + 1 anchors in all;
//   distinct == 1  reads of the repeat path: true when all seeds of the read have different keys (a reference position holds
//                  one minimizer, so each of the <= 23 positions between the two anchors contributes at most one anchor).  Keys
//                  are compared through a 32-bit function of the slot payload in a lane-private LDS strip (equal keys always
//                  collide, so a duplicate is never missed); singletons are never filtered by mm_seed_select, and checking all
//                  seeds instead of the selected ones only makes the premise stronger.  Reads with more than 32 seeds are left.
// Undecided reads are compacted into `leftover` (dense input for k_chain_small / k_expand: a straggler would otherwise make
// its whole wave pay for the full path).
__global__ __launch_bounds__(64) void k_pair_pass(K2Args a, int distinct)
{
    __shared__ uint32_t s_key[32 * 64];
    const uint32_t lane = threadIdx.x;
    const uint32_t n_work = *a.work_count;
    uint32_t n_host_wave = 0;
    unsigned long long anchors_wave = 0;
    for (uint32_t base = blockIdx.x * 64; base < n_work; base += gridDim.x * 64) {
        const uint32_t wi = base + lane;
        const bool valid = wi < n_work;
        uint32_t r = 0, info = 0, n_seed = 0, tot = 0;
        bool found = false;
        if (valid && distinct == 2) {
            // SH_F_CIGAR (ChainParams::ext_*): the read is decided here only if the whole outcome of minimap2 is known from its seed
            // records - every seed a singleton (none is filtered, anchors = seeds), all anchors on ONE diagonal of one contig /
            // strand, consecutive anchors d <= min(max_dist_x, max_dist_y) apart.  Then mg_lchain_dp links each anchor to its
            // predecessor (f[i-1] + min(k, d) >= f[j] + min(k, d_ij) for every j, first scanned wins ties), the backtrack returns ONE
            // chain with all anchors, the region is primary and is aligned; its max stretch is the whole chain, the ungapped
            // middle has U = sum max(0, d - k) bases outside the exact k-mers, so mm_test_zdrop sees a drop <= b * U <= zdrop and
            // no second pass runs; mlen >= covered = k + sum min(k, d) >= 2k >= min_chain_score (span >= k); one gap at most
            // can be left-aligned into the middle (mm_fix_cigar, from the right extension), so the first or the last k-mer stays a
            // run of k matches: dp_max >= a * k >= min_dp_max; max_clip_ratio >= 1 disables the clip test.  Host-checked premises
            // in fill_chain_params.  Everything else goes through the full path.
            r = a.work[wi];
            info = a.k1info[r];
            n_seed = info >> 16 & 0x7fffu;
            const uint4 *rec = a.records + (size_t)r * a.seed_cap;
            const int32_t qlen = (int32_t)(a.offsets[r + 1] - a.offsets[r]);
            int32_t mdy = a.P.is_sr ? (qlen > a.P.max_gap ? qlen : a.P.max_gap) : a.P.max_gap, mdx;
            if (a.P.max_gap_ref > 0) mdx = a.P.max_gap_ref;
            else if (a.P.max_frag_len > 0) { mdx = a.P.max_frag_len - qlen; if (mdx < a.P.max_gap) mdx = a.P.max_gap; }
            else mdx = a.P.max_gap;
            if (mdx < a.P.bw) mdx = a.P.bw;
            if (mdy < a.P.bw) mdy = a.P.bw;
            const int32_t dmax = mdx < mdy ? mdx : mdy;
            bool ok = a.P.ext_s1 != 0 && n_seed >= 2 && (int32_t)n_seed >= a.P.min_cnt && n_seed <= a.seed_cap;
            uint64_t w0 = 0; uint32_t q0 = 0, qp = 0; int32_t unc = 0;
            for (uint32_t i = 0; ok && i < n_seed; ++i) {
                const uint4 sd = rec[i];
                const uint64_t w1 = (uint64_t)sd.y << 32 | sd.x;
                if ((sd.z & SH_REC_OCC_MASK) != 1u) { ok = false; break; }
                if (i == 0) { w0 = w1; q0 = sd.w; qp = sd.w >> 1; continue; }
                const bool fw0 = (uint32_t)(w0 & 1u) == (q0 & 1u), fwi = (uint32_t)(w1 & 1u) == (sd.w & 1u);
                const int32_t D = (int32_t)(sd.w >> 1) - (int32_t)(q0 >> 1), d = (int32_t)(sd.w >> 1) - (int32_t)qp;
                const int32_t dr = (int32_t)((uint32_t)w1 >> 1) - (int32_t)((uint32_t)w0 >> 1);
                ok = (w1 >> 32) == (w0 >> 32) && fwi == fw0 && (fw0 ? dr == D : dr == -D) && d > 0 && d <= dmax;
                unc += d > a.P.k ? d - a.P.k : 0;
                qp = sd.w >> 1;
            }
            const int32_t span = (int32_t)qp - (int32_t)(q0 >> 1);
            found = ok && span >= a.P.k && unc <= a.P.ext_unc_max;
            tot = n_seed;
        } else if (valid) {
            r = a.work[wi];
            info = a.k1info[r];
            n_seed = info >> 16 & 0x7fffu;
            const uint4 *rec = a.records + (size_t)r * a.seed_cap;
            uint64_t w_p = 0, w_pp = 0; uint32_t q_p = 0, q_pp = 0; int have = 0;
            auto codiag = [&](uint64_t wf, uint32_t qf, uint64_t wg, uint32_t qg) {
                const int32_t D = (int32_t)(qg >> 1) - (int32_t)(qf >> 1);
                const bool fwf = (uint32_t)(wf & 1u) == (qf & 1u), fwg = (uint32_t)(wg & 1u) == (qg & 1u);
                const int32_t dr = (int32_t)((uint32_t)wg >> 1) - (int32_t)((uint32_t)wf >> 1);
                return D >= a.P.pair_dq_min && D <= a.P.pair_dq_max && (wf >> 32) == (wg >> 32) && fwf == fwg && (fwf ? dr == D : dr == -D);
            };
            const uint32_t lim = distinct ? (n_seed <= 32u ? n_seed : 0u) : n_seed;
            for (uint32_t i = 0; i < lim; ++i) {
                const uint4 sd = rec[i];
                const uint32_t occ = sd.z & SH_REC_OCC_MASK;
                if (occ <= (uint32_t)a.P.mid_occ) tot += occ;
                if (distinct) s_key[i * 64 + lane] = sd.x ^ (sd.y * 0x9E3779B1u);
                if (occ == 1) {
                    const uint64_t w1 = (uint64_t)sd.y << 32 | sd.x;
                    if (have >= 1) found |= codiag(w_p, q_p, w1, sd.w);
                    if (have >= 2) found |= codiag(w_pp, q_pp, w1, sd.w);
                    w_pp = w_p; q_pp = q_p; w_p = w1; q_p = sd.w; ++have;
                }
            }
            if (!distinct) found = found && tot <= (uint32_t)a.P.max_skip + 1u;
        }
        if (distinct == 1 && __ballot(found) != 0) {
            uint32_t mx = found ? n_seed : 0u;
            mx = wave_all_max_u32(mx);
            bool dup = false;
            for (uint32_t i = 1; i < mx; ++i) {
                const uint32_t ki = s_key[i * 64 + lane];
                for (uint32_t j = 0; j < i; ++j) dup |= (i < n_seed) && s_key[j * 64 + lane] == ki;
            }
            found = found && !dup;
        }
        if (found) {
            a.flags[r] = 1;
            write_trace(a.trace, r, (int32_t)(info & 0xffffu), (int32_t)n_seed, (int32_t)tot, 0, 0, 1, a.P.min_sc, 1);
            anchors_wave += distinct == 1 ? tot : 0u;
        }
        n_host_wave += (uint32_t)__popcll(__ballot(found));
        const bool undecided = valid && !found;
        const uint32_t li = wave_append(a.leftover_count, undecided);
        if (undecided) a.leftover[li] = r;
    }
    if (lane == 0 && n_host_wave) { atomicAdd(&a.ctr->sh_host[SHARD()], n_host_wave); atomicAdd(&a.ctr->sh_pair[SHARD()], n_host_wave); }
    anchors_wave = wave_all_add_u64(anchors_wave);
    if (lane == 0 && anchors_wave) atomicAdd(&a.ctr->sh_anchors[SHARD()], anchors_wave);
}

// after the pair pass over the repeat path's list: the survivors (collected in the deferral list) become the list
__global__ void k_pair_swap(Counters *ctr)
{
    if (ctr->n_big_total == 0) ctr->n_big_total = ctr->n_big[0];
    ctr->n_big[0] = ctr->n_big_defer[0];
    ctr->n_big_defer[0] = 0;
}

// stable sort of the first n lanes' (x, q) by x: rank by comparison with every broadcast key, then push
__device__ inline void wave_rank_sort(uint64_t &x, uint32_t &q, uint32_t n, uint32_t lane)
{
    uint32_t xl = (uint32_t)x, xh = (uint32_t)(x >> 32);
    uint32_t rank = 0;
    for (uint32_t b = 0; b < n; ++b) {
        uint64_t xb = (uint64_t)rdlane(xh, b) << 32 | rdlane(xl, b);
        rank += (xb < x) || (xb == x && b < lane);
    }
    if (lane >= n) rank = lane;
    int addr = (int)(rank * 4);
    xl = (uint32_t)__builtin_amdgcn_ds_permute(addr, (int)xl);
    xh = (uint32_t)__builtin_amdgcn_ds_permute(addr, (int)xh);
    q = (uint32_t)__builtin_amdgcn_ds_permute(addr, (int)q);
    x = (uint64_t)xh << 32 | xl;
}

// SH_F_CIGAR, flag-only (ChainParams::ext_*): the cluster(s) around a read's singleton seeds, alone.
// One wave per read of the repeat path's list, one lane per seed.  A read with unique seeds has its true locus there; the other
// occurrences of its repeated seeds - hundreds of loci - need not be expanded, sorted and chained if they cannot matter:
//   1. K = every anchor of the read on the singletons' strand / contig within reach of them: the window [lo - mdx, hi + mdx] around
//      the anchors found so far is searched in each seed's (sorted) occurrence list until nothing new turns up, so K is a union of
//      COMPLETE clusters (no anchor within max_dist_x outside it) and mg_lchain_dp + the backtrack over K give exactly the chains the
//      full anchor set gives there (clusters are independent DP problems);
//   2. a chain made of anchors outside K only uses seeds that have occurrences outside K; its score is at most the query bases
//      those seeds' k-mers cover (U_out: a link adds min(k, dq) at most);
//   3. so if K's top chain scores more than U_out it is regs[0] of mm_gen_regs whatever the rest holds, it is aligned, and chain_lemma
//      decides whether it survives.  Anything else (no singleton, a seed above mid_occ, singletons at several loci, K beyond 64 anchors,
//      no such margin, a stretch the lemma cannot vouch for) stays on the list for the full path.
// Its own kernel (not a step of k_expand): ~60 registers instead of ~150, so three times the waves hide the dependent list probes.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_local_cluster(K2Args a)
{
    __shared__ uint64_t e_x[64];
    __shared__ uint32_t e_q[64];
    __shared__ int32_t e_f[64], e_pt[128];
    const uint32_t lane = threadIdx.x;
    const uint32_t n_work = *a.work_count;
    const ChainParams &P = a.P;
    uint32_t n_hit = 0;
    for (uint32_t w = blockIdx.x; w < n_work; w += gridDim.x) {
        const uint32_t r = a.work[w];
        const uint32_t info = a.k1info[r];
        const uint32_t n_seed = info >> 16 & 0x7fffu;
        const int32_t qlen = (int32_t)(a.offsets[r + 1] - a.offsets[r]);
        bool decided = false;
        if (n_seed <= 64u && n_seed >= 2u) {
            const bool have0 = lane < n_seed;
            const uint4 rec0 = have0 ? a.records[(size_t)r * a.seed_cap + lane] : make_uint4(0, 0, 0, 0);
            const uint32_t occ0 = rec0.z & SH_REC_OCC_MASK;
            const bool single = have0 && occ0 == 1u;
            const uint64_t sm = __ballot(single);
            bool ok = sm != 0 && __ballot(have0 && occ0 > (uint32_t)P.mid_occ) == 0;       // no seed is filtered: every occurrence is an anchor
            uint64_t xs = 0; uint32_t qs_ = 0;
            if (single) make_anchor((uint64_t)rec0.y << 32 | rec0.x, rec0.w, qlen, P.k, xs, qs_);
            const int fl = sm ? __ffsll((unsigned long long)sm) - 1 : 0;
            const uint32_t hiw = (uint32_t)wave_bcast((int32_t)(uint32_t)(xs >> 32), fl);
            // singletons elsewhere (another contig / strand) count as seeds with occurrences outside K
            const bool s_in = single && (uint32_t)(xs >> 32) == hiw;
            uint32_t lo = s_in ? (uint32_t)xs : 0xffffffffu, hi = s_in ? (uint32_t)xs : 0u;
            lo = wave_all_min_u32(lo); hi = wave_all_max_u32(hi);
            const uint32_t mdx = chain_max_dist_x(P, qlen);
            ok = ok && hi - lo <= 4u * mdx;                       // singletons of one locus; far-apart ones on one contig: full path
            const uint32_t rel = hiw >> 31;                      // strand relation of K's anchors
            const uint64_t w1m = (uint64_t)rec0.y << 32 | rec0.x;
            const uint64_t *__restrict__ lst = a.positions + (w1m >> SH_SLOT_NBITS);
            const bool multi = have0 && occ0 > 1u;
            uint32_t c_l = s_in ? 1u : 0u, first_l = 0;
            if (ok) {
                for (int round = 0; round < 4; ++round) {
                    const uint32_t wlo = lo > mdx ? lo - mdx : 0u, whi = hi + mdx < hi ? 0xffffffffu : hi + mdx;
                    uint32_t nlo = lo, nhi = hi;
                    if (multi) {
                        // lower bound of the window in this seed's ascending occurrence list: 8-ary steps (7 independent loads a round instead
                        // of one), then the entries from there 8 at a time
                        const uint64_t key_lo = (uint64_t)(hiw & 0x7fffffffu) << 32 | (uint64_t)wlo << 1;
                        uint32_t b = 0, len = occ0;
                        while (len > 8u) {
                            const uint32_t step = (len + 7u) >> 3;
                            uint64_t pv[7];
#pragma unroll
                            for (uint32_t j = 0; j < 7u; ++j) { const uint32_t ix = b + (j + 1u) * step - 1u; pv[j] = ix < b + len ? lst[ix] : ~0ull; }
                            uint32_t cnt = 0;
#pragma unroll
                            for (uint32_t j = 0; j < 7u; ++j) cnt += pv[j] < key_lo;
                            const uint32_t nb = b + cnt * step;
                            len = min(step, b + len - nb); b = nb;
                        }
                        {
                            uint64_t pv[8];
#pragma unroll
                            for (uint32_t j = 0; j < 8u; ++j) pv[j] = j < len ? lst[b + j] : ~0ull;
                            uint32_t cnt = 0;
#pragma unroll
                            for (uint32_t j = 0; j < 8u; ++j) cnt += pv[j] < key_lo;
                            b += cnt;
                        }
                        first_l = b; c_l = 0;
                        bool more = true;
                        for (uint32_t t0 = b; more && t0 < occ0 && c_l <= 16u; t0 += 8u) {
                            uint64_t pv[8];
#pragma unroll
                            for (uint32_t j = 0; j < 8u; ++j) pv[j] = t0 + j < occ0 ? lst[t0 + j] : ~0ull;
#pragma unroll
                            for (uint32_t j = 0; j < 8u; ++j) {
                                const uint64_t pw = pv[j];
                                if (!more || (uint32_t)(pw >> 32) != (hiw & 0x7fffffffu) || ((uint32_t)pw >> 1) > whi) { more = false; continue; }
                                if ((((uint32_t)pw & 1u) != (rec0.w & 1u)) == (rel != 0)) { ++c_l; const uint32_t xp = (uint32_t)pw >> 1; nlo = min(nlo, xp); nhi = max(nhi, xp); }
                            }
                        }
                    }
                    nlo = wave_all_min_u32(nlo); nhi = wave_all_max_u32(nhi);
                    if (__ballot(c_l > 16u) != 0) { ok = false; break; }
                    if (nlo == lo && nhi == hi) break;
                    if (round == 3) { ok = false; break; }
                    lo = nlo; hi = nhi;
                }
            }
            const uint32_t n_k = ok ? wave_sum_u32(c_l) : 0u;
            if (ok && n_k >= 2 && n_k <= 64) {
                // K's anchors in generation order (seed order, then occurrence order): the order the full expansion gives them
                const uint32_t ex = wave_excl_scan_u32(c_l, lane);
                const uint32_t wlo = lo > mdx ? lo - mdx : 0u, whi = hi + mdx < hi ? 0xffffffffu : hi + mdx;
                if (s_in) { e_x[ex] = xs; e_q[ex] = qs_; }
                else if (multi && c_l) {
                    uint32_t o = 0;
                    for (uint32_t t = first_l; t < occ0 && o < c_l; ++t) {
                        const uint64_t pw = lst[t];
                        if ((((uint32_t)pw & 1u) != (rec0.w & 1u)) != (rel != 0)) continue;
                        if (((uint32_t)pw >> 1) > whi || ((uint32_t)pw >> 1) < wlo) continue;
                        uint64_t x; uint32_t q;
                        make_anchor(pw, rec0.w, qlen, P.k, x, q);
                        e_x[ex + o] = x; e_q[ex + o] = q; ++o;
                    }
                }
                __syncthreads();
                uint64_t x = lane < n_k ? e_x[lane] : ~0ull;
                uint32_t q = lane < n_k ? e_q[lane] : 0u;
                const uint64_t xp = wave_shr1_u64(x, 0ull);
                if (__ballot(lane > 0 && lane < n_k && x < xp) != 0) wave_rank_sort(x, q, n_k, lane);
                __syncthreads();
                if (lane < n_k) { e_x[lane] = x; e_q[lane] = q; }
                // clusters: lane i owns the cluster that starts at anchor i
                const uint64_t xq = wave_shr1_u64(x, 0ull);
                const bool start = lane < n_k && (lane == 0 || (uint32_t)(x >> 32) != (uint32_t)(xq >> 32) || (uint32_t)x - (uint32_t)xq > mdx);
                const uint64_t stm = __ballot(start);
                const uint64_t above = lane >= 63 ? 0ull : stm & ~((2ULL << lane) - 1);
                const uint32_t clen = start ? (above ? (uint32_t)__ffsll((unsigned long long)above) - 1u : n_k) - lane : 0u;
                __syncthreads();
                // U_out: query bases covered by the k-mers of the seeds that have occurrences outside K (lanes are in query order)
                const bool outside = have0 && occ0 > c_l;
                const uint64_t om = __ballot(outside);
                const uint64_t below = om & ((1ULL << lane) - 1);
                const int32_t prev_en = (int32_t)((uint32_t)__shfl((int)rec0.w, below ? 63 - __clzll((unsigned long long)below) : 0) >> 1) + 1;
                const int32_t en = (int32_t)(rec0.w >> 1) + 1, st = en - P.k;
                const int32_t cover = outside ? en - (below && prev_en > st ? prev_en : st) : 0;
                const int32_t u_out = (int32_t)wave_sum_u32((uint32_t)cover);
                int32_t code = 0;
                MidReq mid{0, 0, 0, 0, 0};
                // K is one cluster of co-diagonal anchors, d <= min(max_dist_x, max_dist_y) apart (a read that matches its locus up to
                // substitutions): the situation of k_pair_pass mode 2 - mg_lchain_dp links every anchor to its predecessor, the backtrack returns
                // one chain, its stretch is the chain - with K in place of the read's seeds.  No DP: covered, U and the span come from the lanes.
                bool fast = false;
                {
                    int32_t mdy = P.is_sr ? (qlen > P.max_gap ? qlen : P.max_gap) : P.max_gap;
                    if (mdy < P.bw) mdy = P.bw;
                    const int32_t dmax = (int32_t)mdx < mdy ? (int32_t)mdx : mdy;
                    const bool in = lane < n_k;
                    const uint32_t dg = (uint32_t)x - q, dg0 = (uint32_t)__builtin_amdgcn_readlane((int)dg, 0);
                    const uint32_t qpv = (uint32_t)wave_shr1((int32_t)q, 0);
                    const int32_t dq = (int32_t)q - (int32_t)qpv;
                    const bool good = !in || (dg == dg0 && (lane == 0 || (dq > 0 && dq <= dmax)));
                    fast = P.ext_s1 != 0 && __popcll(stm) == 1 && (int32_t)n_k >= P.min_cnt && __ballot(!good) == 0;
                    if (fast) {
                        const bool link = in && lane > 0;
                        const int32_t covered = P.k + (int32_t)wave_sum_u32(link ? (uint32_t)(dq < P.k ? dq : P.k) : 0u);
                        const int32_t unc = (int32_t)wave_sum_u32(link && dq > P.k ? (uint32_t)(dq - P.k) : 0u);
                        const uint32_t q_first = (uint32_t)__builtin_amdgcn_readlane((int)q, 0), q_last = (uint32_t)wave_bcast((int32_t)q, (int)n_k - 1);
                        const uint32_t lo_first = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, 0), hi_w = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), 0);
                        if (covered > u_out && (int32_t)(q_last - q_first) >= P.k && covered >= P.min_sc) {
                            if (unc <= P.ext_unc_max) code = 1;
                            else {
                                code = 2;
                                mid.rid = (int32_t)(hi_w & 0x7fffffffu); mid.rev = (int32_t)(hi_w >> 31);
                                mid.qs = (int32_t)q_first + 1 - P.k; mid.qe = (int32_t)q_last + 1; mid.rs = (int32_t)lo_first + 1 - P.k;
                            }
                        } else fast = false;      // not regs[0] for sure, or too short: the general way (which will also decline)
                    }
                }
                // else K's clusters one after the other, each chained by the WHOLE wave (64 predecessors at a time, then the backtrack executed
                // uniformly): a single lane walking a 13-anchor cluster through LDS took ~70 us a read, 84 % of this kernel.  Every lane ends with the same bc.
                BestChain bc{};
                uint64_t todo = fast ? 0ull : __ballot(start && clen >= 2u);
                while (todo) {
                    const uint32_t cb = (uint32_t)__ffsll((unsigned long long)todo) - 1u;
                    todo &= todo - 1;
                    const int32_t cl = wave_bcast((int32_t)clen, (int)cb);
                    SliceStore S{(const uint64_t *)&e_x[cb], (const uint32_t *)&e_q[cb], e_f + cb, e_pt + 2 * (size_t)cb};
                    int32_t n_u, best;
                    auto hif = [&](int32_t j) { return (uint32_t)(S.X(j) >> 32); };
                    const BestEmit<SliceStore, decltype(hif)> be{&S, &bc, region_hash(qlen), P.k, cb, hif, true, TandemQ{a.records + (size_t)r * a.seed_cap, n_seed, 1u, qlen, info >> 31}};
                    chain_dp_wave(S, cl, qlen, P, lane);
                    backtrack_mask(S, cl, P, n_u, best, false, be);
                }
                if (!fast && bc.n > 0 && !bc.tie && bc.score > u_out) {      // uniform
                    SliceStore S{(const uint64_t *)&e_x[bc.base], (const uint32_t *)&e_q[bc.base], e_f + bc.base, e_pt + 2 * (size_t)bc.base};
                    code = chain_lemma(S, bc.zi, bc.end_i, P, (uint32_t)(S.X(bc.zi) >> 32), mid);
                }
                const bool mid_ok = resolve_mid_wave(code == 2, mid, a.bases + a.offsets[r], qlen, a.BC, P);
                decided = __ballot(code == 1 || (code == 2 && mid_ok)) != 0;
                __syncthreads();
            }
        }
        if (decided) { if (lane == 0) a.flags[r] = 1; ++n_hit; }
        else if (lane == 0) a.leftover[atomicAdd(a.leftover_count, 1u)] = r;
    }
    if (lane == 0 && n_hit) { atomicAdd(&a.ctr->sh_host[SHARD()], n_hit); atomicAdd(&a.ctr->sh_lemma[SHARD()], n_hit); }
}


// Flag-only hand-over (short-read mode): can the cluster [s, s + len), len <= 64, hold a chain that reaches the score `bs`?  The WAVE
// answers from the anchors' geometry, one lane per anchor, before any lane chains it:
//   * a link needs |d_ref - d_query| <= bw (comput_sc; the max_ii shortcut goes through the same test), so with the anchors' diagonals
//     binned at width bw + 1 a chain never leaves a run of adjacent occupied bins - tandem-array copies of a read sit one monomer
//     (171 > bw) apart in diagonal and fall into different runs;
//   * a chain visits each query position once (dq > 0) and a link adds at most min(k, dq): it scores at most the bases that the
//     k-mers at the DISTINCT query positions of its run cover.
// false = it might (or the diagonals span more than 64 bins): chain it.
__device__ inline bool cluster_cannot_reach(const uint64_t *__restrict__ x, const uint32_t *__restrict__ q, uint32_t s, uint32_t len, int32_t bs,
                                            const ChainParams &P, uint32_t lane)
{
    const bool on = lane < len;
    const uint32_t lo = on ? (uint32_t)x[s + lane] : 0u, qv = on ? q[s + lane] & 0x7fffffffu : 0u;
    const int32_t dg = (int32_t)(lo - qv);
    const int32_t mn = wave_all_min(on ? dg : INT32_MAX);
    const uint32_t bin = on ? (uint32_t)(dg - mn) / (uint32_t)(P.bw + 1) : 0u;
    if (__ballot(on && bin >= 64u) != 0) return false;
    uint32_t ol = on && bin < 32u ? 1u << bin : 0u, oh = on && bin >= 32u ? 1u << (bin - 32u) : 0u;
    ol = wave_all_or(ol); oh = wave_all_or(oh);
    const uint64_t occ = (uint64_t)oh << 32 | ol, starts = occ & ~(occ << 1);
    const uint32_t comp = (uint32_t)__popcll(starts & (bin == 63u ? ~0ull : (2ull << bin) - 1ull));
    uint64_t key = on ? (uint64_t)comp << 32 | qv : ~0ull;
    uint32_t dummy = 0;
    wave_rank_sort(key, dummy, len, lane);
    const uint64_t prev = wave_shr1_u64(key, 0ull);
    const uint32_t kc = (uint32_t)(key >> 32);
    const bool first = lane == 0 || kc != (uint32_t)(prev >> 32);
    int32_t c = !on ? 0 : (first ? P.k : min(P.k, (int32_t)((uint32_t)key - (uint32_t)prev)));
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int32_t t = __shfl_up(c, o);
        const uint32_t tk = (uint32_t)__shfl_up((int)kc, o);
        if ((int)lane >= o && tk == kc) c += t;
    }
    const int32_t mx = wave_all_max(on ? c : 0);
    return mx < bs;
}

// Chains every cluster of a sorted anchor array x[0..n) / q[0..n).  f / pt: DP state arrays of n (2n) int32.
// Returns this thread's (chains, best score, clusters).  `found`: block-shared flag of the flag-only early exit
// (nullptr: chain everything).  All threads of the block call.
//   phase A  cluster starts marked in bit 31 of q (read-only afterwards, so a cluster's owner may recycle its x
//            slice as heap space while other lanes are still measuring their clusters);
//   sweep 0  one lane per cluster start: clusters of <= 6 anchors are chained by their lane (register-mask DP);
//            larger ones are queued in LDS;
//   sweep 1  the queued clusters, one WAVE per cluster (chain_dp_wave).
// The per-read result is a sum / max over clusters, so the order is free; in flag-only mode a read that found a
// chain in a small cluster never touches its big ones (the true-locus cluster, dense tandem arrays).
struct BigList { uint32_t *start, *len; int32_t *count; uint32_t cap; };
// clusters of the giant path that go to k_cluster_dp instead (one wave per cluster, across reads)
struct GlobalQ { SortItem *const *items; const uint32_t *cap; uint32_t *count; uint32_t w, in_b; unsigned long long off; };
__device__ inline int cl_class(uint32_t len) { return len > 4096 ? 0 : (len > 1024 ? 1 : (len > 256 ? 2 : 3)); }

template <bool CONTIG, class PX, class PQ>
__device__ inline void chain_sorted(PX x, PQ q, int32_t *f, int32_t *pt, uint32_t n, uint32_t tid, uint32_t nthr, int32_t qlen,
                                    const ChainParams &P, volatile int32_t *found, BigList bl,
                                    int32_t &n_u_thr, int32_t &best_thr, uint32_t &n_cl_thr, const GlobalQ *gq = nullptr,
                                    uint32_t *nxt = nullptr, const ChainSink *sk = nullptr, uint32_t read = 0, uint64_t *heap = nullptr,
                                    BestChain *bc = nullptr, uint32_t rhash = 0, int phase = -1, TandemQ tq = TandemQ{nullptr, 0u, 1u, 0, 0u},
                                    ParFillLds *pf = nullptr, bool *pre_io = nullptr, unsigned long long *pf_dbg = nullptr, uint32_t *pf_cnt = nullptr,
                                    uint64_t *ax_out = nullptr, uint32_t *aq_out = nullptr, uint32_t gq_min = 0)
{   // ax_out / aq_out (!CONTIG, with gq): the sorted, marked anchors are copied to the arena and clusters of more than gq_min anchors are queued
    // for k_cluster_dp instead of being chained by one wave of this block while the others wait.   pf: the DP of all clusters at once (par_fill_block) before any cluster is visited; *pre_io: whether it applied (out; in for CONTIG phase 1,
    // whose f / p / dirty marks are phase 0's)   // phase (flag-only hand-over, where a cluster that cannot beat the best score found so far is skipped): the BIG clusters first -
    //   CONTIG: 0 = only clusters of more than 64 anchors (queued for k_cluster_dp), 1 = only the others, afterwards; -1 = all at once
    //   else:   the clusters a wave chains (> 6 anchors) before the ones a lane chains, inside this call
    // bc: SH_F_CIGAR flag-only, n <= 64: every lane chains its clusters itself and remembers its top chain (BestChain), nothing is emitted
    // sk: hand-over mode - every chain of every cluster is emitted (found must be nullptr); heap: n words for the backtrack
    // heaps of clusters with more than 64 anchors (nullptr: the x slice itself, which the hand-over must not destroy)
    const uint32_t mdx = chain_max_dist_x(P, qlen);
    const bool keep_single = !(P.k < P.min_sc || P.min_cnt > 1);
    const uint32_t per = CONTIG ? (n + nthr - 1) / nthr : 1;
    const uint32_t i_beg = CONTIG ? (tid * per < n ? tid * per : n) : tid, i_step = CONTIG ? 1 : nthr;
    const uint32_t i_end = CONTIG ? (i_beg + per < n ? i_beg + per : n) : n;
    if (tid == 0) *bl.count = 0;
    for (uint32_t i = tid; i < n && !(CONTIG && phase == 1); i += nthr) {      // phase 1 finds the marks of phase 0 in place
        bool start = i == 0;
        if (!start) { const uint64_t xi = x[i], xp = x[i - 1]; start = (uint32_t)(xi >> 32) != (uint32_t)(xp >> 32) || (uint32_t)xi - (uint32_t)xp > mdx; }
        if (start) q[i] |= 0x80000000u;
    }
    __syncthreads();
    if (!CONTIG && gq && ax_out) { for (uint32_t i = tid; i < n; i += nthr) { ax_out[i] = x[i]; aq_out[i] = q[i]; } }
    bool pre = false;
    if (!pf && pre_io) pre = *pre_io;      // the caller knows (k_giant_top ran par_fill_block)
    if (pf) {
        if (CONTIG && phase == 1) pre = pre_io && *pre_io;
        else {
            pre = par_fill_block(x, q, f, pt, n, qlen, P, tid, nthr, *pf); if (pre_io) *pre_io = pre;
            if (pf_cnt && pre && tid == 0) { atomicAdd(&pf_cnt[SHARD()], 1u); if (pf->n_dirty) atomicAdd(&pf_cnt[64 + SHARD()], (uint32_t)pf->n_dirty); }
            if (pf_dbg && tid == 0) { atomicAdd(&pf_dbg[pre ? 0 : 1], 1ull); atomicAdd(&pf_dbg[2], (unsigned long long)pf->n_dirty); atomicAdd(&pf_dbg[pre ? 3 : 4], (unsigned long long)n); }
        }
    }
    // one cluster [i, i + len): chained by this lane if small, else queued for a whole wave
    auto handle = [&](uint32_t i, uint32_t len, uint32_t known_dirty = 2u) {
        if (len < 2 && !keep_single) return;
        const int32_t tflag = pre && known_dirty == 2u ? pt[2 * (size_t)i + 1] : 0;
        const bool pre_c = pre && (known_dirty == 2u ? tflag != PF_DIRTY : known_dirty == 0u);
        // flag-only hand-over: a cluster of len anchors cannot chain to more than k * len; below the best score already handed over it is moot
        if (CONTIG && phase == 1 && sk && sk->best && P.is_sr && P.ext_lemma && len <= 64u && (pre ? tflag == PF_DEAD : f[i] == INT32_MIN)) return;      // ruled out by its wave (prefilter below)
        if (sk && sk->best) {
            const int32_t bs = sink_best_score(*sk, read);
            if ((int64_t)P.k * (int64_t)len < (int64_t)bs) return;
            if (CONTIG && bs > P.k && len <= 64u && qlen <= 256) {
                // tighter: a chain visits each query position once (dq > 0) and a link adds at most min(k, dq), so it scores at most the
                // bases the k-mers at the cluster's DISTINCT query positions cover.  Tandem-array reads: a cluster of 20 anchors is often
                // the same 2-3 seeds over and over.
                unsigned long long m[4] = {0, 0, 0, 0};
                for (uint32_t t = 0; t < len; ++t) { const uint32_t qq = (uint32_t)q[i + t] & 0xffu; m[qq >> 6] |= 1ull << (qq & 63); }
                int32_t cover = 0, last = -100000;
#pragma unroll
                for (int w = 0; w < 4; ++w) { unsigned long long b = m[w]; while (b) { const int32_t pos = w * 64 + __ffsll(b) - 1; b &= b - 1; cover += pos - last < P.k ? pos - last : P.k; last = pos; } }
                if (cover < bs) return;
            }
        }
        if (bc && bc->n > 0 && (int64_t)P.k * (int64_t)len < (int64_t)bc->score) return;
        if (bc) {      // len <= 64 (the caller's n is)
            SliceStore S{(const uint64_t *)&x[i], (const uint32_t *)&q[i], f + i, pt + 2 * (size_t)i};
            int32_t n_u, best;
            auto hi = [&](int32_t j) { return (uint32_t)(S.X(j) >> 32); };
            const BestEmit<SliceStore, decltype(hi)> be{&S, bc, rhash, P.k, i, hi, true, tq};
            chain_dp_mask(S, (int)len, qlen, P);
            backtrack_mask(S, (int)len, P, n_u, best, false, be);
            ++n_cl_thr;
            if (n_u > 0) { n_u_thr += n_u; if (best > best_thr) best_thr = best; }
            return;
        }
        if (len > (CONTIG ? 64u : 6u)) {      // arena path: only clusters beyond the register-mask DP go wave-wide (flag-only hand-over: beyond 8 -
                                                                       // a lane chaining 64 anchors out of HBM holds its whole block up, and most small ones are ruled out anyway)
            if (gq && (CONTIG || len > gq_min)) {
                const int cc = cl_class(len);
                const uint32_t gs = atomicAdd(&gq->count[cc], 1u);
                if (gs < gq->cap[cc]) { SortItem ci{gq->w, len, (uint32_t)qlen | (pre_c ? 0x80000000u : 0u), gq->in_b | i << 1, gq->off + i}; gq->items[cc][gs] = ci; return; }     // pad = buffer | cluster start << 1; qlen bit 31: DP done
            }
            const int32_t slot = atomicAdd(bl.count, 1);
            if ((uint32_t)slot < bl.cap) { bl.start[slot] = i; bl.len[slot] = len; return; }
        }
        SliceStore S{(const uint64_t *)&x[i], (const uint32_t *)&q[i], f + i, pt + 2 * (size_t)i, pre_c ? (int32_t)i : 0};
        int32_t n_u, best;
        chain_cluster(S, (int32_t)len, qlen, P, heap ? heap + i : (uint64_t *)&x[i], n_u, best, found != nullptr, sk, read, i, pre_c);
        ++n_cl_thr;
        if (n_u > 0) { n_u_thr += n_u; if (best > best_thr) best_thr = best; if (found) *found = 1; }
    };
    if (CONTIG && nxt) {
        // Cluster lengths without walking the clusters: nxt[t] = first cluster start at or after thread t's range (suffix
        // minimum over the threads' first starts), so a start's cluster ends at the next start inside the own range or at
        // nxt[tid + 1].  A tandem-array cluster of 10^5 anchors used to be measured by one thread, one load at a time.
        uint32_t fs = n;
        for (uint32_t i = i_beg; i < i_end; ++i) if (q[i] >> 31) { fs = i; break; }
        nxt[tid] = fs;
        __syncthreads();
        for (uint32_t off = 1; off < nthr; off <<= 1) {
            const uint32_t v = tid + off < nthr ? nxt[tid + off] : n;
            __syncthreads();
            if (v < nxt[tid]) nxt[tid] = v;
            __syncthreads();
        }
        const uint32_t after = tid + 1 < nthr ? nxt[tid + 1] : n;
        // Phase 1 of the flag-only hand-over: the big clusters have set the score to beat.  Before a lane spends ~100 dependent loads chaining
        // a small cluster out of HBM, its wave rules out the clusters whose geometry cannot reach that score (cluster_cannot_reach):
        // verdict in f[start] (every small cluster of the read gets one; the DP of a surviving cluster overwrites it).
        const bool prefilter = phase == 1 && sk && sk->best && P.is_sr && P.ext_lemma;
        bool run_sweeps = true;      // block-uniform
        if (prefilter) {
            const int32_t bs0 = sink_best_score(*sk, read);
            const uint32_t wv = tid >> 6, ln = tid & 63;
            const uint32_t wb = min(n, wv * 64u * per), we = min(n, (wv + 1u) * 64u * per);
            for (uint32_t c0 = wb; c0 < we; c0 += 64) {
                const uint32_t i0 = c0 + ln;
                // marks of this chunk and of the next one (a cluster may end in another wave's span); the starts below `we` are this wave's
                const uint64_t m0 = __ballot(i0 < n && (q[i0] >> 31)), m1 = __ballot(i0 + 64 < n && (q[i0 + 64] >> 31));
                uint64_t todo = m0 & __ballot(i0 < we);
                while (todo) {
                    const uint32_t b = (uint32_t)__ffsll((unsigned long long)todo) - 1u;
                    todo &= todo - 1;
                    const uint64_t above = b == 63u ? 0ull : m0 >> (b + 1u);
                    uint32_t len;
                    if (above) len = (uint32_t)__ffsll((unsigned long long)above);
                    else if (c0 + 64 >= n) len = n - (c0 + b);
                    else if (m1) len = 64u - b + (uint32_t)__ffsll((unsigned long long)m1) - 1u;
                    else len = c0 + 128 >= n ? n - (c0 + b) : 65u;
                    if (len > 64u) continue;
                    const uint32_t st = c0 + b;
                    bool dead = (int64_t)P.k * (int64_t)len < (int64_t)bs0;
                    if (!dead && bs0 > P.k && len >= 2u) dead = cluster_cannot_reach((const uint64_t *)&x[0], (const uint32_t *)&q[0], st, len, bs0, P, ln);
                    if (ln == 0) {
                        // survivors are packed into the block's LDS list, so that every thread gets one instead of the few threads whose ranges
                        // hold them chaining two or three in a row while the rest of their waves idle; a full list leaves them to the sweeps
                        bool listed = false;
                        const bool dirty = pre && pt[2 * (size_t)st + 1] == PF_DIRTY;
                        if (!dead) {
                            const int32_t slot = atomicAdd(bl.count, 1);
                            if ((uint32_t)slot < bl.cap) { bl.start[slot] = st; bl.len[slot] = len | (dirty ? 0x80000000u : 0u); listed = true; }
                        }
                        if (pre) { if (dead || listed) pt[2 * (size_t)st + 1] = PF_DEAD; }      // f[st] is the DP's
                        else f[st] = dead || listed ? INT32_MIN : 0;
                    }
                }
            }
            __syncthreads();
            const uint32_t n_list = (uint32_t)*bl.count < bl.cap ? (uint32_t)*bl.count : bl.cap;
            for (uint32_t b = tid; b < n_list; b += nthr) {
                const uint32_t ci = bl.start[b], cl = bl.len[b] & 0x7fffffffu;
                const bool pre_c = pre && !(bl.len[b] >> 31);
                if ((int64_t)P.k * (int64_t)cl < (int64_t)sink_best_score(*sk, read)) continue;      // the score to beat has risen since
                SliceStore S{(const uint64_t *)&x[ci], (const uint32_t *)&q[ci], f + ci, pt + 2 * (size_t)ci, pre_c ? (int32_t)ci : 0};
                int32_t n_u, best;
                chain_cluster(S, (int32_t)cl, qlen, P, heap ? heap + ci : (uint64_t *)&x[ci], n_u, best, false, sk, read, ci, pre_c);
                ++n_cl_thr;
                if (n_u > 0) { n_u_thr += n_u; if (best > best_thr) best_thr = best; }
            }
            __syncthreads();
            const bool overflow = (uint32_t)*bl.count > bl.cap;
            __syncthreads();
            if (tid == 0) *bl.count = 0;      // the wave-level list of the code below starts empty
            __syncthreads();
            run_sweeps = overflow;
        }
        // sweeps by cluster size: the clusters a lane chains itself in rising order of cost (the DP is quadratic and a wave
        // waits for its slowest lane), then - only if the read is still undecided in flag-only mode - the big ones go to
        // the queue.  Lengths are free now, so one sweep would queue every big cluster before the first chain is found.
        const bool tq = sk && sk->best;
        const uint32_t thr[5] = {0u, 8u, 24u, 64u, 0xffffffffu}; (void)tq;
        for (int sweep = phase == 0 ? 3 : 0; run_sweeps && sweep < (phase == 1 ? 3 : 4); ++sweep) {
            uint32_t i = fs < i_end ? fs : i_end;
            while (i < i_end) {
                if (found && *found) break;            // flag-only: the read is decided
                uint32_t j = i + 1;
                while (j < i_end && !(q[j] >> 31)) ++j;
                const uint32_t len = (j < i_end ? j : after) - i;
                if (len > thr[sweep] && len <= thr[sweep + 1]) handle(i, len);
                i = j;
            }
            __syncthreads();
        }
    } else {
        const bool big_first = sk && sk->best && !bc;      // flag-only hand-over: the clusters most likely to hold regs[0] first
        for (uint32_t i = i_beg; i < i_end; i += i_step) {
            if (found && *found) break;            // flag-only: the read is decided
            if (!(q[i] >> 31)) continue;
            uint32_t j = i + 1;
            while (j < n && !(q[j] >> 31)) ++j;
            if (!big_first || j - i > 6u) handle(i, j - i);
        }
    }
    __syncthreads();
    const uint32_t n_big = (uint32_t)*bl.count < bl.cap ? (uint32_t)*bl.count : bl.cap;
    const uint32_t wave = tid >> 6, n_wave = nthr >> 6, lane = tid & 63;
    for (uint32_t b = wave; b < n_big; b += n_wave) {
        if (found && *found) break;
        const uint32_t i = bl.start[b], len = bl.len[b];
        if (sk && sk->best && !bc && P.is_sr && P.ext_lemma && len <= 64u) {      // flag-only hand-over: the geometry first (cluster_cannot_reach), then the DP
            const int32_t bs = sink_best_score(*sk, read);
            if (bs > P.k && cluster_cannot_reach((const uint64_t *)&x[0], (const uint32_t *)&q[0], i, len, bs, P, lane)) continue;
        }
        const bool pre_c = pre && pt[2 * (size_t)i + 1] != PF_DIRTY;
        SliceStore S{(const uint64_t *)&x[i], (const uint32_t *)&q[i], f + i, pt + 2 * (size_t)i, pre_c ? (int32_t)i : 0};
        int32_t n_u, best;
        chain_cluster_wave(S, (int32_t)len, qlen, P, heap ? heap + i : (uint64_t *)&x[i], n_u, best, found != nullptr, lane, sk, read, i, pre_c);
        if (lane == 0) {
            ++n_cl_thr;
            if (n_u > 0) { n_u_thr += n_u; if (best > best_thr) best_thr = best; if (found) *found = 1; }
        }
    }
    __syncthreads();
    if (!CONTIG && sk && sk->best && !bc) {      // ... then the small ones, most of which the bound k * len now rules out
        for (uint32_t i = i_beg; i < i_end; i += i_step) {
            if (!(q[i] >> 31)) continue;
            uint32_t j = i + 1;
            while (j < n && !(q[j] >> 31)) ++j;
            if (j - i <= 6u) handle(i, j - i);
        }
        __syncthreads();
    }
}

struct K3Args {
    const uint64_t *offsets; const uint64_t *positions;
    uint4 *records; uint32_t seed_cap;
    const uint32_t *k1info; uint8_t *flags; sh_trace *trace;
    const uint32_t *list; const uint32_t *list_count;     // reads of this pass
    uint32_t *defer_list; uint32_t *defer_count;          // reads that found no arena space
    uint32_t *next_list; uint32_t *next_count;            // pass 0: reads that must re-chain with max_occ
    uint32_t *resketch_list;                              // reads this path cannot take (see k_expand)
    const unsigned long long *seed_off;                   // long reads: first seed record of read r (nullptr: r * seed_cap)
    uint32_t *sel_scratch;                                // long reads: >= 2 u32 per seed at [2 * seed_off[r]] for mm_seed_select
    Counters *ctr; BigBufs B; ChainParams P;
    int32_t pass, max_occ, flag_only, dbg;
    ChainSink sink; int32_t emit;      // SH_F_CIGAR: chains are handed to the extension stage (flag_only is 0 then: no early exit)
    BaseCtx BC;
    int32_t t_mode;                    // SH_F_CIGAR and no trace wanted: only regs[0] matters, shortcuts allowed
    int32_t quiet;                     // a second visit of reads that were counted already: no statistics
    uint32_t pft_gmin;                 // par_fill_tiled: reads of this many anchors get eight lanes per anchor (SCRUBBY_HIP_PFT_GMIN, tests)
    int32_t top_max;                   // backtrack_block_top: candidates a read may have at its top score (<= TOPBT_MAX; SCRUBBY_HIP_TOPBT_MAX, tests)
    // long-read presets, flag-only with the extension filter: k_expand lists reads of more than 64 anchors here instead of in the sort
    // classes, k_lr_locus keeps the anchors of the reference windows that can hold regs[0] and passes the read on
    SortItem *locus_items[3]; uint32_t *lr_drop; int32_t locus, locus_shift;
    uint64_t *stage_x; uint32_t *stage_q; unsigned long long stage_cap;      // where k_expand leaves those reads' anchors (generation order)
    int32_t locus_min_qlen;            // reads shorter than this are not thinned out (mm_map_frag's rescue test could depend on what is left out)
    int32_t cl_lds;                    // k_sort_lds queues its big clusters for k_cluster_dp (long-read presets handing over every chain)
};


// Flag-only pair test (ChainParams::pair_dq_*).  One wave, the read's seeds one per lane (my_n = anchors the seed would
// contribute, 0 = filtered).  Picks the pair of selected seeds pair_dq_min..pair_dq_max apart in the query whose shorter
// occurrence list is shortest, walks that list 64 occurrences at a time and looks each one's co-diagonal partner up in the
// other seed's (ascending) list by binary search.  true: the read has a mapping.  false: nothing is known.
#define PAIR_MAX_COST 4096u       // load rounds a pair may cost before the full path is the better deal (a satellite read: ~600 rounds against 26 k anchors to expand, sort and chain)
#define PAIR_ATTEMPTS 3
__device__ inline bool pair_decides(const uint4 rec, uint32_t my_n, uint32_t n_seed, uint32_t lane, const uint64_t *__restrict__ pos, const ChainParams &P, uint32_t *dbg_cost0 = nullptr)
{
    const uint32_t qp = rec.w >> 1;
    const bool sel = my_n > 0;
    // one sweep over the seeds: how many selected seeds share this lane's key (itself included), and the lane's cheapest partner ahead
    // of it in the query.  With key multiplicity M a reference position contributes up to M anchors, so at most (D + 1) * M - 2 sort
    // between two co-diagonal anchors D apart: D is capped at (max_skip + 1) / M - 1 (M = 1: pair_dq_max; tandem arrays whose k-mers
    // recur inside the read: M = 2 -> 12, 3 -> 7, 5 -> 4, more: no pair can be trusted).
    uint32_t mult = 0, p_occ = UINT32_MAX, p_u = 0;
    auto sweep = [&](int32_t dmax, bool count) {
        p_occ = UINT32_MAX; p_u = 0;
        for (uint32_t u = 0; u < n_seed; ++u) {
            const uint32_t ou = rdlane(my_n, u);
            if (ou == 0) continue;
            const uint32_t xu = rdlane(rec.x, u), yu = rdlane(rec.y, u), qu = rdlane(qp, u);
            if (count) mult += sel && xu == rec.x && yu == rec.y;
            const int32_t d = (int32_t)qu - (int32_t)qp;
            if (sel && d >= P.pair_dq_min && d <= dmax && ou < p_occ) { p_occ = ou; p_u = u; }
        }
    };
    sweep(P.pair_dq_max, true);
    uint32_t M = sel ? mult : 0u;
    M = wave_all_max_u32(M);
    if (M > 1) {
        const int32_t dmax = min((int32_t)(((uint32_t)P.max_skip + 1u) / M) - 1, P.pair_dq_max);
        if (dmax < P.pair_dq_min) { if (dbg_cost0 && lane == 0) atomicAdd(dbg_cost0 - 3, 1u); return false; }       // dbg: n_leg_reason[0] = keys repeat too often
        sweep(dmax, false);
    }
    // cost of a pair ~ dependent loads: the shorter list is walked (one load round per 64), each step a binary search of the longer
    uint32_t cost = UINT32_MAX;
    if (p_occ != UINT32_MAX) {
        const uint32_t mn = my_n < p_occ ? my_n : p_occ, mx = my_n < p_occ ? p_occ : my_n;
        cost = (mn > 1 ? (mn + 63) / 64 : 0) + ((mn + 63) / 64) * (mx > 1 ? 32 - __clz(mx) : 0);
    }
  for (int attempt = 0; attempt < PAIR_ATTEMPTS; ++attempt) {      // the cheapest pair first; a pair that finds nothing retires its earlier seed
    unsigned long long key = (unsigned long long)cost << 32 | lane;
    key = wave_all_min_u64(key);
    if ((uint32_t)(key >> 32) > PAIR_MAX_COST) { if (dbg_cost0 && lane == 0 && attempt == 0) atomicAdd(dbg_cost0 - 2, 1u); break; }   // dbg: [1] = no pair in range / too costly
    if (dbg_cost0 && (uint32_t)(key >> 32) == 0 && lane == 0 && attempt == 0) atomicAdd(dbg_cost0, 1u);
    const uint32_t lf = (uint32_t)key & 63u, lg = rdlane(p_u, lf);          // F: earlier in the query, G: later
    const uint32_t nF = rdlane(my_n, lf), nG = rdlane(my_n, lg);
    const uint32_t D = (rdlane(rec.w, lg) >> 1) - (rdlane(rec.w, lf) >> 1);
    const bool iter_f = nF <= nG;
    const uint32_t li = iter_f ? lf : lg, ls = iter_f ? lg : lf;
    const uint32_t nI = iter_f ? nF : nG, nS = iter_f ? nG : nF;
    const uint64_t w1I = (uint64_t)rdlane(rec.y, li) << 32 | rdlane(rec.x, li), w1S = (uint64_t)rdlane(rec.y, ls) << 32 | rdlane(rec.x, ls);
    const uint32_t qsI = rdlane(rec.w, li) & 1u, qsS = rdlane(rec.w, ls) & 1u;
    const int64_t delta = iter_f ? (int64_t)D : -(int64_t)D;                  // qpos_S - qpos_I
    const uint64_t *__restrict__ pi = pos + (w1I >> SH_SLOT_NBITS), *__restrict__ ps = pos + (w1S >> SH_SLOT_NBITS);
    for (uint32_t base = 0; base < nI; base += 64) {
        const uint32_t idx = base + lane;
        const bool live = idx < nI;
        const uint64_t oI = nI == 1 ? w1I : pi[live ? idx : nI - 1];
        const bool fw = (uint32_t)(oI & 1u) == qsI;                            // make_anchor: same strand -> forward anchor
        const int64_t rS = (int64_t)((uint32_t)oI >> 1) + (fw ? delta : -delta);
        const bool ok = live && rS >= 0 && rS < (1ll << 31);
        const uint64_t target = (oI & 0xffffffff00000000ULL) | (uint64_t)rS << 1 | (uint64_t)(fw ? qsS : qsS ^ 1u);
        bool found;
        if (nS == 1) found = ok && target == w1S;
        else {
            uint32_t lo = 0, len = nS;
            while (len > 1) { const uint32_t half = len >> 1; const uint64_t v = ps[lo + half]; lo = v <= target ? lo + half : lo; len -= half; }
            found = ok && ps[lo] == target;
        }
        if (__ballot(found) != 0) return true;
    }
    if (lane == lf) cost = UINT32_MAX;
  }
    if (dbg_cost0 && lane == 0) atomicAdd(dbg_cost0 - 1, 1u);      // dbg: [2] = searched, no co-diagonal partner
    return false;
}

// one wave per read: seeds -> filter -> anchors (arena) -> [<= 64: sort + clusters]
// Seeds are handled 64 at a time, one per lane.  mm_seed_select's streak logic needs a read's seeds in one
// tile; with more seeds it is only needed when some streak could keep a seed (max_high_occ > 0), which for
// occ_dist = 500 means reads longer than 250 bp: those go to the legacy path.
template <bool LONG>      // LONG: per-read seed offsets and mm_seed_select across seed tiles (the long-read front end's records)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_expand(K3Args a)
{
    const uint32_t lane = threadIdx.x;
    const uint32_t n_items = *a.list_count;
    const ChainParams P = a.P;      // local copies (see k_giant_chain)
    const ChainSink sink_l = a.sink;
    const bool plain_cut = !(P.occ_dist > 0 && P.max_max_occ > a.max_occ);
    WaveAlloc al_sort[N_SORT_CLS];
    uint32_t n_clusters = 0, n_pair = 0, n_lemma = 0;
    unsigned long long anchors_wave = 0, a_cur = 0, a_end = 0, s_cur = 0, s_end = 0;      // wave-local slices of the anchor arena / the staging buffer
    __shared__ uint64_t e_x[64];
    __shared__ uint32_t e_q[64];
    __shared__ int32_t e_f[64], e_pt[128], e_bcount;
    __shared__ uint32_t e_bstart[8], e_blen[8];
    // reads by ticket, 4 to a draw (one wave per read: a long read expands 100 to 100 000 anchors; one atomic per read on a single address
    // cost more than the imbalance it removed - a single address sustains ~90 M atomics/s)
    uint32_t w_next = 0, w_stop = 0;
    for (;;) {
        if (w_next >= w_stop) {
            uint32_t t = 0;
            if (lane == 0) t = atomicAdd(&a.ctr->expand_ticket, 4u);
            w_next = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
            w_stop = w_next + 4u < n_items ? w_next + 4u : n_items;
            if (w_next >= n_items) break;
        }
        const uint32_t w = w_next++;
        const uint32_t r = a.list[w];
        const uint32_t info = a.k1info[r];
        const uint32_t n_seed = LONG ? info >> 16 : (info >> 16 & 0x7fffu);      // K1's records: bit 31 = the read has a tandem seed
        const int32_t qlen = (int32_t)(a.offsets[r + 1] - a.offsets[r]);
        const uint32_t n_st = (n_seed + 63) / 64;
        const bool no_keep = plain_cut || (int32_t)((double)qlen / (double)P.occ_dist + .499) <= 0;   // every streak has max_high_occ == 0
        const bool general_mt = LONG && n_st > 1 && !no_keep;     // streak logic across seed tiles (long reads)
        if (n_st > 1 && !no_keep && !general_mt) {
            if (lane == 0) {
                BigMeta m{r, 0, 0, 2u};
                a.B.meta[w] = m;
                uint32_t i = atomicAdd(&a.ctr->n_resketch, 1u);
                a.resketch_list[i] = r;
            }
            continue;
        }
        uint4 *recb = a.records + (LONG ? (size_t)a.seed_off[r] : (size_t)r * a.seed_cap);
        // occurrence filter of seed tile t: my_n = anchors this lane's seed contributes (0 if filtered / absent)
        auto eval = [&](uint32_t t, uint4 &rec, uint32_t &my_n, bool &flt, bool &have) {
            const uint32_t sidx = t * 64 + lane;
            have = sidx < n_seed;
            rec = have ? recb[sidx] : make_uint4(0, 0, 0, 0);
            const uint32_t occ = rec.z & SH_REC_OCC_MASK, qposz = rec.w;
            const bool high = have && occ > (uint32_t)a.max_occ;
            flt = false;
            if (plain_cut) flt = high;
            else if (general_mt) flt = (rec.z >> 31) != 0;      // decided by select_multi_tile below
            else if (n_st > 1) flt = high && n_seed >= 2;      // no_keep holds
            else {
                const uint64_t hm = __ballot(high);
                if (n_seed >= 2 && hm != 0) {
                    const uint64_t low = __ballot(have && !high);
                    const uint64_t below = low & ((1ULL << lane) - 1);
                    const uint64_t above = lane >= 63 ? 0 : low & ~((2ULL << lane) - 1);
                    const int32_t last0 = below ? 63 - __clzll((unsigned long long)below) : -1;
                    const int32_t nxt = above ? __ffsll((unsigned long long)above) - 1 : (int32_t)n_seed;
                    const uint32_t qp_last = (uint32_t)__shfl((int)qposz, last0 < 0 ? 0 : last0);
                    const uint32_t qp_next = (uint32_t)__shfl((int)qposz, nxt >= (int32_t)n_seed ? 0 : nxt);
                    const int32_t ps = last0 < 0 ? 0 : (int32_t)(qp_last >> 1);
                    const int32_t pe = nxt >= (int32_t)n_seed ? qlen : (int32_t)(qp_next >> 1);
                    int32_t mho = (int32_t)((double)(pe - ps) / (double)P.occ_dist + .499);
                    if (mho > 128) mho = 128;
                    int32_t rank = 0;
                    if (__ballot(high && mho > 0) != 0) {
                        for (uint32_t u = 0; u < n_seed; ++u) {
                            uint32_t ot = rdlane(occ, u);
                            bool in = (int32_t)u > last0 && (int32_t)u < nxt;
                            rank += in && (ot < occ || (ot == occ && u < lane));
                        }
                    }
                    if (high) {
                        flt = !(mho > 0 && rank < mho);
                        if (occ > (uint32_t)P.max_max_occ) flt = true;
                    }
                }
            }
            my_n = (have && !flt) ? occ : 0u;
        };
        if (general_mt) {
            // mm_seed_select over more than 64 seeds.  Forward sweep: index of the nearest low-occurrence seed before
            // each seed; backward sweep: the one after it, then the streak's max_high_occ and the seed's rank in it.
            uint32_t *scr = a.sel_scratch + 2 * (size_t)a.seed_off[r];
            int32_t carry = -1;
            uint32_t n_high_total = 0;
            for (uint32_t t = 0; t < n_st; ++t) {
                const uint32_t sidx = t * 64 + lane;
                const bool have = sidx < n_seed;
                const uint32_t occ = have ? recb[sidx].z & SH_REC_OCC_MASK : 0u;
                const bool high = have && occ > (uint32_t)a.max_occ;
                const uint64_t low = __ballot(have && !high);
                const uint64_t below = low & ((1ULL << lane) - 1);
                const int32_t prev = below ? (int32_t)(t * 64) + 63 - __clzll((unsigned long long)below) : carry;
                if (have) scr[2 * sidx] = (uint32_t)prev;
                if (low) carry = (int32_t)(t * 64) + 63 - __clzll((unsigned long long)low);
                n_high_total += (uint32_t)__popcll(__ballot(high));
            }
            int32_t carry_next = (int32_t)n_seed;
            for (int32_t t = (int32_t)n_st - 1; t >= 0; --t) {
                const uint32_t sidx = (uint32_t)t * 64 + lane;
                const bool have = sidx < n_seed;
                const uint4 rec = have ? recb[sidx] : make_uint4(0, 0, 0, 0);
                const uint32_t occ = rec.z & SH_REC_OCC_MASK;
                const bool high = have && occ > (uint32_t)a.max_occ;
                const uint64_t low = __ballot(have && !high);
                const uint64_t above = lane >= 63 ? 0 : low & ~((2ULL << lane) - 1);
                const int32_t nxt = above ? t * 64 + (int32_t)__ffsll((unsigned long long)above) - 1 : carry_next;
                bool flt = false;
                if (high && n_seed >= 2 && n_high_total > 0) {
                    const int32_t last0 = (int32_t)scr[2 * sidx];
                    const int32_t ps = last0 < 0 ? 0 : (int32_t)(recb[last0].w >> 1);
                    const int32_t pe = nxt >= (int32_t)n_seed ? qlen : (int32_t)(recb[nxt].w >> 1);
                    int32_t mho = (int32_t)((double)(pe - ps) / (double)P.occ_dist + .499);
                    if (mho > 128) mho = 128;
                    flt = true;
                    if (mho > 0) {      // keep the mho smallest by (occ, index) of the streak (last0, nxt)
                        int32_t rank = 0;
                        for (int32_t u = last0 + 1; u < nxt && rank < mho; ++u) {
                            const uint32_t ou = recb[u].z & SH_REC_OCC_MASK;
                            rank += (ou < occ) || (ou == occ && u < (int32_t)sidx);
                        }
                        if (rank < mho) flt = false;
                    }
                    if (occ > (uint32_t)P.max_max_occ) flt = true;
                }
                if (have) recb[sidx].z = occ | (uint32_t)flt << 31;
                if (low) carry_next = t * 64 + (int32_t)__ffsll((unsigned long long)low) - 1;
            }
            __syncthreads();
        }
        // ---- pass 1 over the seed tiles: anchor count and rep_len ----
        uint4 rec0 = make_uint4(0, 0, 0, 0); uint32_t my_n0 = 0; bool flt0 = false, have0 = false;
        unsigned long long n_part = 0;
        int32_t contrib = 0, carry_en = 0;
        for (uint32_t t = 0; t < n_st; ++t) {
            uint4 rec; uint32_t my_n; bool flt, have;
            eval(t, rec, my_n, flt, have);
            if (t == 0) { rec0 = rec; my_n0 = my_n; flt0 = flt; have0 = have; }
            n_part += my_n;
            // rep_len: union length of the filtered seeds' query intervals (ascending end positions)
            const uint64_t fm = __ballot(flt);
            const int32_t en = (int32_t)(rec.w >> 1) + 1, st = en - P.k;
            const uint64_t belowf = fm & ((1ULL << lane) - 1);
            const int32_t prev_en_l = (int32_t)((uint32_t)__shfl((int)rec.w, belowf ? 63 - __clzll((unsigned long long)belowf) : 0) >> 1) + 1;
            const int32_t prev_en = belowf ? prev_en_l : carry_en;
            if (flt) contrib += en - (st > prev_en ? st : prev_en);
            if (fm) carry_en = (int32_t)(rdlane(rec.w, 63 - __clzll((unsigned long long)fm)) >> 1) + 1;
        }
        const int32_t rep_len = (int32_t)wave_sum_u32((uint32_t)contrib);
        n_part = wave_all_add_u64(n_part);
        // the count can exceed 31 bits only for absurd inputs; saturate (such a read never gets arena space)
        const uint32_t n_a = n_part > 0x7fffffffull ? 0x7fffffffu : (uint32_t)n_part;

        if (n_a > (GT << 8)) {             // beyond the giant path's 8 merge rounds: legacy path (never for sr: <= 22 x 5000)
            if (lane == 0) {
                atomicAdd(&a.ctr->n_leg_reason[2], 1u);
                BigMeta m{r, 0, 0, 2u};
                a.B.meta[w] = m;
                uint32_t i = atomicAdd(&a.ctr->n_resketch, 1u);
                a.resketch_list[i] = r;
            }
            continue;
        }
        if (!LONG && a.flag_only && !a.emit && P.pair_dq_max > 0 && n_st == 1 && n_a > (uint32_t)P.pair_min_anchors && pair_decides(rec0, my_n0, n_seed, lane, a.positions, P, ((a.dbg & 16) && (!(a.dbg & 128) || n_a > 4096u)) ? &a.ctr->n_leg_reason[3] : nullptr)) {
            // decided without a single anchor: sh_stats.n_anchors still counts what the occurrence filter admitted
            if (lane == 0) { BigMeta m{r, n_a, rep_len, 0u}; a.B.meta[w] = m; a.B.acc_nu[w] = 1; a.B.acc_best[w] = P.min_sc; }
            anchors_wave += n_a; ++n_pair;
            continue;
        }
        const bool in_lds = n_a <= 64;     // short anchor lists never leave the CU
        const bool to_stage = LONG && a.locus && !in_lds && n_a <= LOCUS_MAX_N;      // k_lr_locus takes the arena slots, for what it keeps
        // anchor slots: the wave advances the arena (or staging) cursor by 16 Ki slots at a time
        if (to_stage) {
            if (s_cur + n_a > s_end) {
                const unsigned long long want = n_a > 16384u ? n_a : 16384u;
                unsigned long long base = 0;
                if (lane == 0) base = atomicAdd(&a.ctr->stage_cursor, want);
                base = (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base) | (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32)) << 32;
                s_cur = base; s_end = base + want;
            }
        } else if (!in_lds && a_cur + n_a > a_end) {
            const unsigned long long want = n_a > 16384u ? n_a : 16384u;
            unsigned long long base = 0;
            if (lane == 0) base = atomicAdd(&a.ctr->anchor_cursor, want);
            base = (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base) | (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32)) << 32;
            a_cur = base; a_end = base + want;
        }
        const unsigned long long off = to_stage ? s_cur : a_cur;
        const bool defer = !in_lds && off + n_a > (to_stage ? a.stage_cap : a.B.anchor_cap);
        if (lane == 0) {
            BigMeta m{r, n_a, rep_len, defer ? 1u : 0u};
            a.B.meta[w] = m; a.B.acc_nu[w] = 0; a.B.acc_best[w] = 0;
            if (defer) { uint32_t di = atomicAdd(a.defer_count, 1u); a.defer_list[di] = r; }
        }
        if (defer) { if (to_stage) s_cur = s_end = 0; else a_cur = a_end = 0; continue; }
        if (n_a == 0) continue;
        if (to_stage) s_cur += n_a; else if (!in_lds) a_cur += n_a;
        anchors_wave += n_a;
        uint64_t *const gx = in_lds ? e_x : (to_stage ? a.stage_x + off : a.B.ax + off);
        uint32_t *const gq = in_lds ? e_q : (to_stage ? a.stage_q + off : a.B.aq + off);

        // ---- pass 2: anchors in generation order (seed order, then occurrence order) ----
        uint32_t run = 0;
        for (uint32_t t = 0; t < n_st; ++t) {
            uint4 rec = rec0; uint32_t my_n = my_n0; bool flt = flt0, have = have0;
            if (t > 0) eval(t, rec, my_n, flt, have);
            // one lane per ANCHOR: anchor a of this seed tile belongs to the seed s with start[s] <= a < start[s] + my_n[s];
            // s is found by a 6-step binary search over the lanes' prefix sums, so every position load is independent
            const uint32_t excl = wave_excl_scan_u32(my_n, lane);
            const uint32_t tile_total = wave_sum_u32(my_n);
            const uint32_t incl = excl + my_n;                       // non-decreasing over lanes
            const uint64_t w1 = (uint64_t)rec.y << 32 | rec.x;
            const uint64_t *__restrict__ pos = a.positions;
            for (uint32_t a0 = 0; a0 < tile_total; a0 += 64) {
                const uint32_t ai_raw = a0 + lane;
                const bool live = ai_raw < tile_total;
                const uint32_t ai = live ? ai_raw : tile_total - 1;       // every lane takes part in the shuffles
                uint32_t lo = 0, hi = 63;                                    // first lane with incl > ai
#pragma unroll
                for (int it = 0; it < 6; ++it) {
                    const uint32_t mid = (lo + hi) >> 1;
                    const uint32_t v = (uint32_t)__shfl((int)incl, (int)mid);
                    if (v > ai) hi = mid; else lo = mid + 1;
                }
                const uint32_t sdx = lo;
                const uint32_t s_ex = (uint32_t)__shfl((int)excl, (int)sdx), s_n = (uint32_t)__shfl((int)my_n, (int)sdx);
                const uint32_t s_q = (uint32_t)__shfl((int)rec.w, (int)sdx);
                const uint64_t s_w1 = (uint64_t)(uint32_t)__shfl((int)(uint32_t)(w1 >> 32), (int)sdx) << 32 | (uint32_t)__shfl((int)(uint32_t)w1, (int)sdx);
                if (live) {
                    const uint32_t u = ai - s_ex;
                    const uint64_t rpos = s_n == 1 ? s_w1 : pos[(s_w1 >> SH_SLOT_NBITS) + u];
                    uint64_t x; uint32_t q;
                    make_anchor(rpos, s_q, qlen, P.k, x, q);
                    gx[run + ai] = x; gq[run + ai] = q;
                }
            }
            run += tile_total;
        }
        __syncthreads();
        if (in_lds) {
            uint64_t x = lane < n_a ? e_x[lane] : ~0ull;
            uint32_t q = lane < n_a ? e_q[lane] : 0u;
            uint64_t xp = wave_shr1_u64(x, 0ull);
            if (__ballot(lane > 0 && lane < n_a && x < xp) != 0) {
                wave_rank_sort(x, q, n_a, lane);
                __syncthreads();
                if (lane < n_a) { e_x[lane] = x; e_q[lane] = q; }
                __syncthreads();
            }
            int32_t n_u = 0, best = 0;
            bool lemma_done = false;
            if (a.t_mode && P.ext_lemma) {
                // flag-only: every lane chains its clusters and keeps its top chain; the read's top chain (largest z over the lanes) is
                // regs[0] of mm_gen_regs - if chain_lemma vouches for it the read is mapped and nothing is handed over
                BestChain bc{};
                chain_sorted<false>(e_x, e_q, e_f, e_pt, n_a, lane, 64, qlen, P, nullptr, BigList{e_bstart, e_blen, &e_bcount, 8}, n_u, best, n_clusters,
                                    nullptr, nullptr, nullptr, r, nullptr, &bc, region_hash(qlen), -1, TandemQ{recb, n_seed, 1u, qlen, LONG ? 1u : info >> 31});
                unsigned long long zmax = bc.n ? bc.z : 0ull;
                zmax = wave_all_max_u64(zmax);
                const uint64_t holders = __ballot(bc.n > 0 && bc.z == zmax);
                const uint64_t any = __ballot(bc.n > 0);
                bool ok = false;
                if (any == 0) ok = true;                                       // no chain at all: unmapped
                else if (__popcll(holders) == 1) {
                    int32_t code = 0;
                    MidReq mid{0, 0, 0, 0, 0};
                    if (bc.n > 0 && bc.z == zmax && !bc.tie) {
                        SliceStore S{(const uint64_t *)&e_x[bc.base], (const uint32_t *)&e_q[bc.base], e_f + bc.base, e_pt + 2 * (size_t)bc.base};
                        code = chain_lemma(S, bc.zi, bc.end_i, P, (uint32_t)(S.X(bc.zi) >> 32), mid);
                    }
                    const bool mid_ok = resolve_mid_wave(code == 2, mid, a.BC.bases + a.offsets[r], qlen, a.BC, P);
                    ok = __ballot(code == 1 || (code == 2 && mid_ok)) != 0;
                }
                lemma_done = ok;
                if (ok && any != 0) ++n_lemma;
                if (!ok) { n_u = 0; best = 0; __syncthreads(); }
            }
            if (!lemma_done)
            chain_sorted<false>(e_x, e_q, e_f, e_pt, n_a, lane, 64, qlen, P, nullptr, BigList{e_bstart, e_blen, &e_bcount, 8}, n_u, best, n_clusters,
                                nullptr, nullptr, a.emit ? &sink_l : nullptr, r, nullptr);      // <= 64 anchors: no cluster needs a heap
            n_u = wave_all_add(n_u); best = wave_all_max(best);
            if (lane == 0) { a.B.acc_nu[w] = n_u; a.B.acc_best[w] = best; }
            if ((a.dbg & 16) && lane == 0) { atomicAdd(&a.ctr->sort_tot[N_SORT_CLS], 1ull); atomicAdd(&a.ctr->sort_anchor_tot[N_SORT_CLS], (unsigned long long)n_a); }
            __syncthreads();
        } else if (to_stage) {
            // the anchors are in the staging buffer; k_lr_locus decides which of them can matter, moves those to the arena and lists the read in a sort class
            const int lc = n_a <= LOCUS_N0 ? 0 : (n_a <= LOCUS_N1 ? 1 : 2);
            if (lane == 0) { const uint32_t li = atomicAdd(&a.ctr->n_locus[lc], 1u); SortItem it{w, n_a, (uint32_t)qlen, 0, off}; a.locus_items[lc][li] = it; }
        } else {
            const int cls = n_a <= 256 ? 0 : (n_a <= SORT_LDS_A ? 1 : (n_a <= 1024 ? 2 : (n_a <= SORT_LDS_B ? 3 : (n_a <= SORT_LDS_C ? 4 : SORT_CLS_GIANT))));
            uint32_t lo, hi;
            const uint32_t si = al_sort[cls].take(&a.ctr->n_sort[cls], 1, cls <= 1 ? 32u : (cls <= 3 ? 8u : 1u), lo, hi);
            SortItem *const items = a.B.tabs->sort_items[cls];
            for (uint32_t i = lo + lane; i < hi; i += 64) items[i].n = 0;
            if (lane == 0) { SortItem it{w, n_a, (uint32_t)qlen, 0, off}; items[si] = it; }
            if ((a.dbg & 16) && lane == 0) { atomicAdd(&a.ctr->sort_tot[cls], 1ull); atomicAdd(&a.ctr->sort_anchor_tot[cls], (unsigned long long)n_a); }
        }
    }
    for (int cls = 0; cls < N_SORT_CLS; ++cls)
        for (uint32_t i = al_sort[cls].cur + lane; i < al_sort[cls].end; i += 64) a.B.tabs->sort_items[cls][i].n = 0;
    if (lane == 0 && anchors_wave) atomicAdd(&a.ctr->sh_anchors[SHARD()], anchors_wave);
    n_clusters = wave_sum_u32(n_clusters);
    if (lane == 0 && n_clusters) atomicAdd(&a.ctr->sh_clusters[SHARD()], n_clusters);
    if (lane == 0 && n_pair) atomicAdd(&a.ctr->sh_pair[SHARD()], n_pair);
    if (lane == 0 && n_lemma) atomicAdd(&a.ctr->sh_lemma[SHARD()], n_lemma);
}

// ---- long-read presets, flag-only with the extension filter: which anchors can matter --------------------------------------
// The boundary returns mappings.len() > 0 (cleaner.rs:552-556), and the extension stage proves "mapped" from regs[0] - the chain of largest
// score after mm_map_frag's two chaining passes - alone (lr_probe_region).  Both passes work inside CLUSTERS: anchors of one strand and contig
// whose reference positions lie no further apart than D = max(mg_lchain_dp's max_dist_x, mg_lchain_rmq's max_dist); neither links two anchors
// across a wider gap, their windows, marks and backtracks never reach across it, and a chain gains at most k per anchor.  So a cluster of c
// anchors holds no chain scoring more than k * c, in either pass, and leaving WHOLE clusters out changes nothing about the chains of the
// others.  One block per read counts the anchors per reference window of 2^shift >= D bases (sixteen-bit counters in LDS, addressed by a hash
// of the window: a collision only makes a count too large), takes for every anchor the total of the run of occupied windows around its own -
// an upper bound of its cluster's size, the same for every anchor of a cluster - and keeps the anchors whose bound reaches T: the three largest
// bounds of the read, and everything within a sixteenth of the largest.  What is dropped is remembered per read as the largest bound dropped
// (lr_drop); the chains kernel of the extension stage checks afterwards that the answer cannot depend on it (lr_chains_wave) and sends the
// read through the complete path otherwise.  The survivors keep their generation order, so equal reference positions sort as before.
template <int LG>
__global__ __launch_bounds__(256) void k_lr_locus(K3Args a, int lc)
{
    __shared__ uint32_t s_tab[1u << (LG - 1)];
    __shared__ uint32_t s_it, s_w[4][4], s_wcnt[4], s_off[3];
    __shared__ uint32_t s_keep[(LOCUS_MAX_N + 1) / 32 + 8];      // one bit per anchor: it stays
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t n_items = a.ctr->n_locus[lc];
    const int sh = a.locus_shift;
    const uint32_t min_cnt = (uint32_t)(a.P.min_cnt > 1 ? a.P.min_cnt : 1);
    auto bin_of = [&](uint64_t x) -> uint64_t { return (x >> 32) << 24 | (uint64_t)((uint32_t)x >> sh); };      // window id; id +- 1 = the neighbouring window (never another contig's: shift >= 9)
    auto slot = [&](uint64_t id) -> uint32_t { return (uint32_t)((id * 0x9E3779B97F4A7C15ull) >> (64 - LG)); };
    auto cnt_of = [&](uint64_t id) -> uint32_t { const uint32_t h = slot(id); return (s_tab[h >> 1] >> ((h & 1u) << 4)) & 0xffffu; };
    auto block_max = [&](uint32_t v, int k) -> uint32_t {
        v = wave_all_max_u32(v);
        if (lane == 0) s_w[k][wave] = v;
        __syncthreads();
        return max(max(s_w[k][0], s_w[k][1]), max(s_w[k][2], s_w[k][3]));
    };
    unsigned long long st_in = 0, st_kept = 0; uint32_t st_reads = 0;
    for (;;) {
        if (tid == 0) s_it = atomicAdd(&a.ctr->locus_ticket[lc], 1u);
        __syncthreads();
        const uint32_t it = s_it;
        __syncthreads();
        if (it >= n_items) break;
        const SortItem si = a.locus_items[lc][it];
        const uint32_t n = si.n;
        const uint64_t *gx = a.stage_x + si.off; const uint32_t *gq = a.stage_q + si.off;
        for (uint32_t i = tid; i < (1u << (LG - 1)); i += 256) s_tab[i] = 0;
        __syncthreads();
        for (uint32_t i = tid; i < n; i += 256) { const uint32_t h = slot(bin_of(gx[i])); atomicAdd(&s_tab[h >> 1], 1u << ((h & 1u) << 4)); }      // n <= 65535: no carry
        __syncthreads();
        // the bound of anchor i: the anchors in the run of occupied windows around its own (0x7fffffff: a run of more than LOCUS_RUN_MAX
        // windows - the same verdict from each of them)
        auto bound_of = [&](uint64_t x) -> uint32_t {
            const uint64_t id = bin_of(x);
            uint32_t vr = cnt_of(id);
            int len = 1; bool inf = false;
            for (int d = 1;; ++d) { if (d > LOCUS_RUN_MAX) { inf = true; break; } const uint32_t c = cnt_of(id - (uint64_t)d); if (!c) break; vr += c; ++len; }
            if (!inf) for (int d = 1;; ++d) { if (d > LOCUS_RUN_MAX) { inf = true; break; } const uint32_t c = cnt_of(id + (uint64_t)d); if (!c) break; vr += c; ++len; }
            return (inf || len > LOCUS_RUN_MAX) ? 0x7fffffffu : vr;
        };
        uint32_t t1 = 0, t2 = 0, t3 = 0;      // this thread's three largest distinct bounds
        for (uint32_t i = tid; i < n; i += 256) {
            const uint32_t vr = bound_of(gx[i]);
            if (vr > t1) { t3 = t2; t2 = t1; t1 = vr; } else if (vr < t1 && vr > t2) { t3 = t2; t2 = vr; } else if (vr < t2 && vr > t3) t3 = vr;
        }
        const uint32_t V1 = block_max(t1, 0);
        const uint32_t V2 = block_max(t1 < V1 ? t1 : t2, 1);
        const uint32_t V3 = block_max(t1 < V2 ? t1 : (t2 < V2 ? t2 : t3), 2);
        uint32_t T = V3 < (V1 >> 4) ? V3 : (V1 >> 4);
        if ((int32_t)si.qlen < a.locus_min_qlen) T = 0;      // a short read: whether the long join runs can hinge on a chain left out (lr_chains_wave) - it keeps everything
        if (a.dbg & 1024) T = V1;      // tests (SCRUBBY_HIP_LOCUS_TOP1): the largest run only - reads with a second locus must be caught and redone
        if (T < min_cnt) T = min_cnt;
        // which anchors stay (one bit each), how many, the largest bound left out
        uint32_t kept_thr = 0, dmax = 0;
        for (uint32_t c0 = 0; c0 < n; c0 += 256) {
            const uint32_t i = c0 + tid;
            bool keep = false;
            if (i < n) { const uint32_t vr = bound_of(gx[i]); keep = vr >= T; if (!keep && vr > dmax) dmax = vr; }
            const uint64_t km = __ballot(keep);
            if (lane == 0) { s_keep[(c0 >> 5) + 2 * wave] = (uint32_t)km; s_keep[(c0 >> 5) + 2 * wave + 1] = (uint32_t)(km >> 32); }
            kept_thr += keep;
        }
        kept_thr = wave_sum_u32(kept_thr);
        if (lane == 0) s_wcnt[wave] = kept_thr;
        dmax = block_max(dmax, 3);      // (the barrier inside also publishes s_wcnt and s_keep)
        const uint32_t n_keep = s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
        // arena slots for the survivors; a read that finds none comes back in the next iteration of the pass, like k_expand's
        if (tid == 0) {
            unsigned long long o = 0; uint32_t ok = 1;
            if (n_keep) { o = atomicAdd(&a.ctr->anchor_cursor, (unsigned long long)n_keep); ok = o + n_keep <= a.B.anchor_cap; }
            s_off[0] = (uint32_t)o; s_off[1] = (uint32_t)(o >> 32); s_off[2] = ok;
        }
        __syncthreads();
        const unsigned long long doff = (unsigned long long)s_off[0] | (unsigned long long)s_off[1] << 32;
        const uint32_t r = a.B.meta[si.w].r;
        if (!s_off[2]) {
            if (tid == 0) { a.B.meta[si.w].state = 1u; const uint32_t di = atomicAdd(a.defer_count, 1u); a.defer_list[di] = r; }
            __syncthreads();
            continue;
        }
        uint64_t *dx = a.B.ax + doff; uint32_t *dq = a.B.aq + doff;
        uint32_t base = 0;
        for (uint32_t c0 = 0; c0 < n; c0 += 256) {      // stable: the survivors keep their generation order
            const uint32_t i = c0 + tid;
            const uint64_t km = (uint64_t)s_keep[(c0 >> 5) + 2 * wave] | (uint64_t)s_keep[(c0 >> 5) + 2 * wave + 1] << 32;
            uint32_t woff = 0, tot = 0;
#pragma unroll
            for (uint32_t w2 = 0; w2 < 4; ++w2) { const uint32_t c = (uint32_t)__popc(s_keep[(c0 >> 5) + 2 * w2]) + (uint32_t)__popc(s_keep[(c0 >> 5) + 2 * w2 + 1]); if (w2 < wave) woff += c; tot += c; }
            if (i < n && (km >> lane & 1ull)) { const uint32_t d = base + woff + prefix_popc(km); dx[d] = gx[i]; dq[d] = gq[i]; }
            base += tot;
        }
        if (tid == 0) {
            a.lr_drop[r] = dmax;      // 0: every anchor kept
            if (n_keep > 0) {
                const int cls = n_keep <= 256 ? 0 : (n_keep <= SORT_LDS_A ? 1 : (n_keep <= 1024 ? 2 : (n_keep <= SORT_LDS_B ? 3 : (n_keep <= SORT_LDS_C ? 4 : SORT_CLS_GIANT))));
                const uint32_t oi = atomicAdd(&a.ctr->n_sort[cls], 1u);
                SortItem o{si.w, n_keep, si.qlen, 0, doff};
                a.B.tabs->sort_items[cls][oi] = o;
                if (a.dbg & 16) { atomicAdd(&a.ctr->sort_tot[cls], 1ull); atomicAdd(&a.ctr->sort_anchor_tot[cls], (unsigned long long)n_keep); }
            }
            st_in += n; st_kept += n_keep; st_reads += dmax != 0;
        }
        __syncthreads();
    }
    if (tid == 0 && st_in) { atomicAdd(&a.ctr->lr_locus_in, st_in); atomicAdd(&a.ctr->lr_locus_kept, st_kept); if (st_reads) atomicAdd(&a.ctr->lr_locus_reads, st_reads); }
}

// stable block merge sort: 64-element tiles ranked in registers (one tile per wave at a time), then merge-path
// rounds between (sx,sq) and (dx,dq), C outputs per thread and step.  Returns true if the result ended in the
// second buffer.  Every thread of the block must call.
template <class PX, class PQ>
__device__ inline bool block_merge_sort(PX sx, PQ sq, PX dx, PQ dq, uint32_t n)
{
    const uint32_t tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6, nwave = nthr >> 6;
    for (uint32_t base = wave * 64; base < n; base += nwave * 64) {
        const uint32_t cnt = n - base < 64 ? n - base : 64;
        uint64_t x = lane < cnt ? sx[base + lane] : ~0ull;
        uint32_t q = lane < cnt ? sq[base + lane] : 0u;
        uint64_t xp = wave_shr1_u64(x, 0ull);
        if (__ballot(lane > 0 && lane < cnt && x < xp) != 0) {
            wave_rank_sort(x, q, cnt, lane);
            if (lane < cnt) { sx[base + lane] = x; sq[base + lane] = q; }
        }
    }
    __syncthreads();
    const uint32_t C = n >= 16 * nthr ? 16 : (n >= 8 * nthr ? 8 : 4);     // outputs per thread and step (divides 128)
    const uint32_t n_chunks = (n + C - 1) / C;
    bool flipped = false;
    for (uint32_t width = 64; width < n; width <<= 1) {
        for (uint32_t c = tid; c < n_chunks; c += nthr) {
            const uint32_t o0 = c * C, o1 = o0 + C < n ? o0 + C : n;
            const uint32_t pb = o0 / (2 * width) * (2 * width);
            const uint32_t L0 = pb, L1 = pb + width < n ? pb + width : n, R1 = pb + 2 * width < n ? pb + 2 * width : n;
            const uint32_t lenL = L1 - L0, lenR = R1 - L1, d = o0 - pb;
            uint32_t lo = d > lenR ? d - lenR : 0, hi = d < lenL ? d : lenL;
            while (lo < hi) {
                uint32_t mid = (lo + hi) >> 1;
                if (sx[L0 + mid] <= sx[L1 + (d - 1 - mid)]) lo = mid + 1; else hi = mid;
            }
            uint32_t ia = L0 + lo, ib = L1 + (d - lo);
            uint64_t va = ia < L1 ? sx[ia] : ~0ull, vb = ib < R1 ? sx[ib] : ~0ull;
            for (uint32_t o = o0; o < o1; ++o) {
                const bool takeL = ia < L1 && (ib >= R1 || va <= vb);
                if (takeL) { dx[o] = va; dq[o] = sq[ia]; ++ia; va = ia < L1 ? sx[ia] : ~0ull; }
                else { dx[o] = vb; dq[o] = sq[ib]; ++ib; vb = ib < R1 ? sx[ib] : ~0ull; }
            }
        }
        __syncthreads();
        PX tx = sx; sx = dx; dx = tx; PQ tq = sq; sq = dq; dq = tq;
        flipped = !flipped;
    }
    return flipped;
}

// block reduction of the per-thread chain results of one read; thread 0 stores them
__device__ inline void store_read_result(const K3Args &a, uint32_t w, int32_t n_u, int32_t best, uint32_t n_cl, int32_t *red, bool accumulate = false)
{
    if (threadIdx.x == 0) { red[0] = 0; red[1] = 0; }
    __syncthreads();
    if (n_u > 0) { atomicAdd(&red[0], n_u); atomicMax(&red[1], best); }
    n_cl = wave_sum_u32(n_cl);
    if ((threadIdx.x & 63) == 0 && n_cl) atomicAdd(&a.ctr->sh_clusters[SHARD()], n_cl);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (accumulate) { if (red[0]) { atomicAdd(&a.B.acc_nu[w], red[0]); atomicMax(&a.B.acc_best[w], red[1]); } }
        else { a.B.acc_nu[w] = red[0]; a.B.acc_best[w] = red[1]; }
    }
}

// Flag-only shortcut for reads with thousands of anchors, run BEFORE their full sort.  Anchors on different (strand,
// contig) never share a cluster, so each such group is an independent set of DP problems and a mapping found in ONE group
// decides the read (ChainParams::flag_stop).  One block per read: histogram of x >> 32 in LDS, the largest group that fits
// the LDS sort is compacted in generation order (stable, so ties in x keep the order of the full sort), sorted and
// chained; on success the read is done (acc_nu = 1) and its sort item is emptied, so the sort kernels skip it; otherwise
// nothing has changed and the full path runs.  For a long host read that group holds the true locus (~1 k of ~8 k
// anchors); for a re-chained satellite read any group of its tandem arrays does.
#define GP_CAP 2048
#define GP_SLOTS 256
struct GroupScratch { uint32_t *key, *cnt, *wtot, *sel; int32_t *over; };      // LDS: key/cnt[GP_SLOTS], wtot[16], sel[2]

// The probe proper.  src: the read's n anchors in generation order (HBM or LDS); (ax, aq) / (bx, bq): LDS sort buffers of
// at least `cap` entries, ax != src.  Groups of lo < size <= cap qualify.  Returns (block-uniform) whether a mapping was
// found; src is intact unless bx aliases it.  All threads of the block call.
template <class SX, class SQ>
__device__ inline bool group_probe(SX src_x, SQ src_q, uint32_t n, uint64_t *ax, uint32_t *aq, uint64_t *bx, uint32_t *bq, uint32_t lo, uint32_t cap,
                                   const GroupScratch &G, int32_t *found, BigList bl, const ChainParams &P, int32_t qlen, uint32_t &n_cl)
{
    const uint32_t tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wv = tid >> 6, n_wv = nthr >> 6;
    for (uint32_t i = tid; i < GP_SLOTS; i += nthr) { G.key[i] = 0xffffffffu; G.cnt[i] = 0; }
    if (tid == 0) { *found = 0; *G.over = 0; G.sel[0] = 0xffffffffu; G.sel[1] = 0; }
    __syncthreads();
    // histogram of the groups; a thread walks a contiguous slice and adds whole runs (a seed's occurrences are sorted by contig)
    const uint32_t per = (n + nthr - 1) / nthr, i0 = tid * per < n ? tid * per : n, i1 = i0 + per < n ? i0 + per : n;
    auto add = [&](uint32_t key, uint32_t c) {
        uint32_t slot = (key * 2654435761u) >> 24;
        for (uint32_t step = 0; step < GP_SLOTS; ++step) {
            const uint32_t prev = atomicCAS(&G.key[slot], 0xffffffffu, key);
            if (prev == 0xffffffffu || prev == key) { atomicAdd(&G.cnt[slot], c); return; }
            slot = (slot + 1) & (GP_SLOTS - 1);
        }
        *G.over = 1;
    };
    uint32_t run_key = 0, run_n = 0;
    for (uint32_t i = i0; i < i1; ++i) {
        const uint32_t key = (uint32_t)(src_x[i] >> 32);
        if (run_n && key != run_key) { add(run_key, run_n); run_n = 0; }
        run_key = key; ++run_n;
    }
    if (run_n) add(run_key, run_n);
    __syncthreads();
    // the largest qualifying group (ties: the smaller key), at least min_cnt anchors
    for (uint32_t sl = tid; sl < GP_SLOTS; sl += nthr) {
        const uint32_t c = G.cnt[sl];
        if (G.key[sl] != 0xffffffffu && c > lo && c <= cap && (int32_t)c >= P.min_cnt && c >= 2) atomicMax(&G.sel[1], c);
    }
    __syncthreads();
    for (uint32_t sl = tid; sl < GP_SLOTS; sl += nthr)
        if (G.key[sl] != 0xffffffffu && G.cnt[sl] == G.sel[1] && G.sel[1] != 0) atomicMin(&G.sel[0], G.key[sl]);
    __syncthreads();
    const uint32_t g = G.sel[0], m = G.sel[1];
    if (*G.over || g == 0xffffffffu || m == 0 || m == n) return false;      // m == n: the full sort is the same work
    // stable compaction of the group into (ax, aq)
    uint32_t run = 0;
    for (uint32_t base = 0; base < n; base += nthr) {
        const uint32_t i = base + tid;
        uint64_t x = 0; uint32_t q = 0; bool in = false;
        if (i < n) { x = src_x[i]; in = (uint32_t)(x >> 32) == g; if (in) q = src_q[i]; }
        const uint64_t bm = __ballot(in);
        if (lane == 0) G.wtot[wv] = (uint32_t)__popcll(bm);
        __syncthreads();
        uint32_t off = run, tot = 0;
        for (uint32_t w2 = 0; w2 < n_wv; ++w2) { if (w2 < wv) off += G.wtot[w2]; tot += G.wtot[w2]; }
        if (in) { const uint32_t d = off + prefix_popc(bm); ax[d] = x; aq[d] = q; }
        run += tot;
        __syncthreads();
    }
    const bool fl = block_merge_sort(ax, aq, bx, bq, m);
    uint64_t *rx = fl ? bx : ax; uint32_t *rq = fl ? bq : aq;
    int32_t *f = (int32_t *)(fl ? aq : bq), *pt = (int32_t *)(fl ? ax : bx);
    int32_t n_u = 0, best = 0;
    chain_sorted<false>(rx, rq, f, pt, m, tid, nthr, qlen, P, found, bl, n_u, best, n_cl);
    __syncthreads();
    return *found != 0;
}

__global__ __launch_bounds__(256) void k_group_probe(K3Args a, int cls)
{
    __shared__ uint64_t s_x[2][GP_CAP];
    __shared__ uint32_t s_q[2][GP_CAP];
    __shared__ uint32_t s_key[GP_SLOTS], s_cnt[GP_SLOTS], s_wtot[16], s_sel[2], s_bstart[GP_CAP / 7 + 1], s_blen[GP_CAP / 7 + 1];
    __shared__ int32_t s_found, s_bcount, s_over;
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t n_items = a.ctr->n_sort[cls];
    const GroupScratch G{s_key, s_cnt, s_wtot, s_sel, &s_over};
    for (uint32_t it = blockIdx.x; it < n_items; it += gridDim.x) {
        const SortItem si = a.B.tabs->sort_items[cls][it];
        const uint32_t n = si.n;
        if (n == 0) continue;
        uint32_t n_cl = 0;
        const bool hit = group_probe(a.B.ax + si.off, a.B.aq + si.off, n, s_x[0], s_q[0], s_x[1], s_q[1], 0u, (uint32_t)GP_CAP, G, &s_found,
                                     BigList{s_bstart, s_blen, &s_bcount, GP_CAP / 7 + 1}, a.P, (int32_t)si.qlen, n_cl);
        if (tid == 0 && hit) { a.B.acc_nu[si.w] = 1; a.B.acc_best[si.w] = 0; a.B.tabs->sort_items[cls][it].n = 0; }
        n_cl = wave_sum_u32(n_cl);
        if (lane == 0 && n_cl) atomicAdd(&a.ctr->sh_clusters[SHARD()], n_cl);
        __syncthreads();
    }
}

// one block per read with 64 < anchors <= NMAX: sort in LDS, then chain every cluster straight from LDS
// (the second sort buffer becomes the DP state).  Nothing but the per-read result goes back to HBM.
template <int NMAX, int CLS, int NTHR>
__global__ __launch_bounds__(NTHR) void k_sort_lds(K3Args a)
{
    __shared__ uint64_t s_x[2][NMAX];
    __shared__ uint32_t s_q[2][NMAX];
    __shared__ int32_t s_found, s_red[2], s_bcount, s_gover;
    __shared__ uint32_t s_bstart[NMAX / 7 + 1], s_blen[NMAX / 7 + 1], s_gw[16], s_gsel[2];
    __shared__ union GpPf { struct { uint32_t key[GP_SLOTS], cnt[GP_SLOTS]; } g; ParFillLds pf; } s_u;      // group_probe's table (before the sort, flag-only) / par_fill_block's (after it)
    uint32_t *const s_gkey = s_u.g.key, *const s_gcnt = s_u.g.cnt;
    ParFillLds &s_pf = s_u.pf;
    const bool use_pf = (a.emit || !a.flag_only) && !(a.dbg & 128);
    const uint32_t tid = threadIdx.x;
    const uint32_t n_items = a.ctr->n_sort[CLS];
    __shared__ uint32_t s_it;
    for (;;) {
        // a ticket per read instead of a fixed stride: a read whose anchors are one 500-anchor cluster takes ten times the typical one, and with a
        // few reads per block (small batches: an eighth of the records on each of 8 GPUs) the stride left most blocks waiting for the unlucky ones
        if (tid == 0) s_it = atomicAdd(&a.ctr->sort_ticket[CLS], 1u);
        __syncthreads();
        const uint32_t it = s_it;
        __syncthreads();
        if (it >= n_items) break;
        const SortItem si = a.B.sort_items[CLS][it];
        const uint32_t n = si.n;
        if (n == 0) continue;
        const uint64_t *gx = a.B.ax + si.off; const uint32_t *gq = a.B.aq + si.off;
        for (uint32_t i = tid; i < n; i += NTHR) { s_x[0][i] = gx[i]; s_q[0][i] = gq[i]; }
        if (tid == 0) s_found = 0;
        __syncthreads();
        if (a.flag_only && !a.emit && a.P.flag_stop != INT32_MAX && n >= 96 && !(a.dbg & 64)) {
            // flag-only: the largest (strand, contig) group first (see k_group_probe); its sort ping-pongs over the loaded
            // anchors, so a miss reloads them for the full sort
            uint32_t n_cl0 = 0;
            const bool hit = group_probe(&s_x[0][0], &s_q[0][0], n, &s_x[1][0], &s_q[1][0], &s_x[0][0], &s_q[0][0], 0u, n, GroupScratch{s_gkey, s_gcnt, s_gw, s_gsel, &s_gover},
                                         (int32_t *)&s_found, BigList{s_bstart, s_blen, &s_bcount, NMAX / 7 + 1}, a.P, (int32_t)si.qlen, n_cl0);
            if (hit) { store_read_result(a, si.w, tid == 0 ? 1 : 0, 0, n_cl0, s_red); __syncthreads(); continue; }
            for (uint32_t i = tid; i < n; i += NTHR) { s_x[0][i] = gx[i]; s_q[0][i] = gq[i]; }
            if (tid == 0) s_found = 0;
            __syncthreads();
        }
        const bool fl = block_merge_sort(&s_x[0][0], &s_q[0][0], &s_x[1][0], &s_q[1][0], n);
        uint64_t *rx = fl ? s_x[1] : s_x[0]; uint32_t *rq = fl ? s_q[1] : s_q[0];
        int32_t *f = (int32_t *)(fl ? s_q[0] : s_q[1]), *pt = (int32_t *)(fl ? s_x[0] : s_x[1]);
        int32_t n_u = 0, best = 0; uint32_t n_cl = 0;
        // long-read presets: a read is mostly ONE cluster (its locus), which a single wave of this block would chain while the others wait -
        // clusters of more than 256 anchors go to k_cluster_dp's queue instead (their sorted anchors to the arena's second buffer)
        const bool to_q = CLS >= 1 && a.cl_lds && n > 256u;
        const GlobalQ clq{a.B.tabs->cl_items, a.B.tabs->cl_cap, a.ctr->n_cl, si.w, 1u, si.off};
        if (!(a.dbg & 1))
        chain_sorted<false>(rx, rq, f, pt, n, tid, NTHR, (int32_t)si.qlen, a.P, a.flag_only ? &s_found : nullptr,
                            BigList{s_bstart, s_blen, &s_bcount, NMAX / 7 + 1}, n_u, best, n_cl, to_q ? &clq : nullptr, nullptr,
                            a.emit ? &a.sink : nullptr, a.B.meta[si.w].r, a.emit ? a.B.hz + si.off : nullptr,
                            nullptr, 0u, -1, TandemQ{nullptr, 0u, 1u, 0, 0u}, use_pf ? &s_pf : nullptr, nullptr, (a.dbg & 16) ? a.ctr->pf_dbg : nullptr,
                            (a.quiet || (a.emit && a.sink.best && !(a.dbg & 512))) ? nullptr : a.ctr->sh_pf_reads,      // with ChainSink::best k_sort_top counted the read
                            to_q ? a.B.bx + si.off : nullptr, to_q ? a.B.bq + si.off : nullptr, 256u);
        store_read_result(a, si.w, n_u, best, n_cl, s_red);
        __syncthreads();
    }
}

// ---- giant reads (> SORT_LDS_C anchors; almost all are re-chained satellite reads) ------------------------------
// Bandwidth-oriented merge sort over ALL giant reads of the pass at once:
//   k_giant_scan       tile table: read i owns tiles [tile_base[i], tile_base[i+1]) of GT anchors
//   k_giant_chunksort  one block per tile: load coalesced -> stable sort in LDS -> store coalesced
//   per round (run width GT, 2GT, ...):
//     k_giant_partition  one thread per output tile: merge-path split of its first output (independent binary searches)
//     k_giant_merge      one block per output tile: the two input ranges staged coalesced through LDS, merged in LDS,
//                        GT outputs stored coalesced into the other buffer
//   k_giant_chain      one block per read: clusters chained over arena slices (chain_sorted)
// A read needing r rounds ends in buffer (r & 1); reads that are done sit out the later rounds.

__device__ inline uint32_t giant_rounds(uint32_t n)
{
    uint32_t r = 0;
    for (uint32_t w = GT; w < n; w <<= 1) ++r;
    return r;
}

// The dirty clusters par_fill_block left (PF_DIRTY at their first anchor): the sequential DP, p turned into indices into the read's array,
// marks cleared.  One wave per cluster; every thread of the block calls.  A dirty cluster of more than max_len anchors is left as it is and
// false is returned: one wave stepping through 20 k anchors in HBM would hold the whole block (and, for the largest read of a batch, the
// kernel) for milliseconds - k_cluster_dp chains such a cluster through its LDS ring while other waves do the read's other clusters.
template <class PX, class PQ>
__device__ inline bool fix_dirty_clusters(PX x, PQ q, int32_t *f, int32_t *pt, uint32_t n, int32_t qlen, const ChainParams &P, uint32_t tid, uint32_t nthr, const ParFillLds &L,
                                          uint32_t max_len, int32_t *s_left)
{
    const uint32_t wave = tid >> 6, n_wave = nthr >> 6, lane = tid & 63;
    const int32_t nd = L.n_dirty;
    if (tid == 0) *s_left = 0;
    __syncthreads();
    for (int32_t d = (int32_t)wave; d < nd; d += (int32_t)n_wave) {
        uint32_t c = L.dirty[d];
        while (!((uint32_t)q[c] >> 31)) --c;
        uint32_t e = c + 1;
        while (e < n && !((uint32_t)q[e] >> 31)) ++e;
        if (e - c > max_len) { if (lane == 0) *s_left = 1; continue; }
        int32_t mine = 0;
        if (lane == 0) mine = atomicCAS(&pt[2 * (size_t)c + 1], (int32_t)PF_DIRTY, (int32_t)PF_DEAD) == (int32_t)PF_DIRTY ? 1 : 0;
        mine = __builtin_amdgcn_readfirstlane(mine);
        if (!mine) continue;      // another anchor of the same cluster got there first
        SliceStore S{(const uint64_t *)&x[c], (const uint32_t *)&q[c], f + c, pt + 2 * (size_t)c};
        chain_dp_wave(S, (int)(e - c), qlen, P, lane);
        for (uint32_t i = c + lane; i < e; i += 64) { const int32_t pv = pt[2 * (size_t)i]; if (pv >= 0) pt[2 * (size_t)i] = pv + (int32_t)c; pt[2 * (size_t)i + 1] = 0; }
    }
    __syncthreads();
    return *s_left == 0;
}

// Flag-only hand-over (ChainSink::best), ahead of k_sort_lds: sort in LDS, the DP of all clusters at once (par_fill_block), then
// mg_chain_backtrack's first candidates over the whole read (backtrack_block_top) - no cluster is visited.  A read it settles is emptied in
// its class table, so k_sort_lds skips it; the others (many candidates at the top score; par_fill_block not applicable) are left untouched.
// Its own kernel: inside k_sort_lds the extra code cost 80 VGPRs and half the occupancy.
template <int NMAX, int CLS, int NTHR>
__global__ __launch_bounds__(NTHR) void k_sort_top(K3Args a)
{
    __shared__ uint64_t s_x[2][NMAX];
    __shared__ uint32_t s_q[2][NMAX];
    __shared__ ParFillLds s_pf;
    __shared__ long long s_top[18];
    __shared__ uint32_t s_cf[TOPBT_MAX], s_ci[TOPBT_MAX], s_it;
    __shared__ int32_t s_cn, s_red[2];
    const ChainSink sink_l = a.sink; const ChainParams P_l = a.P;      // local copies: see k_giant_chain
    const uint32_t tid = threadIdx.x;
    const uint32_t n_items = a.ctr->n_sort[CLS];
    unsigned long long *const dbg = (a.dbg & 16) ? a.ctr->pf_dbg : nullptr;
    for (;;) {
        if (tid == 0) s_it = atomicAdd(&a.ctr->top_ticket[CLS], 1u);
        __syncthreads();
        const uint32_t it = s_it;
        __syncthreads();
        if (it >= n_items) break;
        const SortItem si = a.B.sort_items[CLS][it];
        const uint32_t n = si.n;
        if (n == 0) continue;
        const int32_t qlen = (int32_t)si.qlen;
        const uint64_t *gx = a.B.ax + si.off; const uint32_t *gq = a.B.aq + si.off;
        for (uint32_t i = tid; i < n; i += NTHR) { s_x[0][i] = gx[i]; s_q[0][i] = gq[i]; }
        __syncthreads();
        const bool fl = block_merge_sort(&s_x[0][0], &s_q[0][0], &s_x[1][0], &s_q[1][0], n);
        uint64_t *rx = fl ? s_x[1] : s_x[0]; uint32_t *rq = fl ? s_q[1] : s_q[0];
        int32_t *f = (int32_t *)(fl ? s_q[0] : s_q[1]), *pt = (int32_t *)(fl ? s_x[0] : s_x[1]);
        const uint32_t mdx = chain_max_dist_x(P_l, qlen);
        uint32_t starts = 0;
        for (uint32_t i = tid; i < n; i += NTHR) {
            bool start = i == 0;
            if (!start) { const uint64_t xi = rx[i], xp = rx[i - 1]; start = (uint32_t)(xi >> 32) != (uint32_t)(xp >> 32) || (uint32_t)xi - (uint32_t)xp > mdx; }
            if (start) { rq[i] |= 0x80000000u; ++starts; }
        }
        __syncthreads();
        if (!par_fill_block(rx, rq, f, pt, n, qlen, P_l, tid, NTHR, s_pf)) { if (dbg && tid == 0) atomicAdd(&dbg[1], 1ull); continue; }
        if (tid == 0 && !a.quiet) { atomicAdd(&a.ctr->sh_pf_reads[SHARD()], 1u); if (s_pf.n_dirty) atomicAdd(&a.ctr->sh_pf_dirty[SHARD()], (uint32_t)s_pf.n_dirty); }
        fix_dirty_clusters(rx, rq, f, pt, n, qlen, P_l, tid, NTHR, s_pf, 0xffffffffu, &s_cn);
        SliceStore S{(const uint64_t *)rx, (const uint32_t *)rq, f, pt};
        const uint32_t read = a.B.meta[si.w].r;
        const StoreEmit<SliceStore> em{&sink_l, &S, read, 0u, tid == 0, P_l.k, region_hash(qlen), qlen, nullptr};
        int32_t n_u = 0, best = 0;
        const bool settled = backtrack_block_top(S, (int32_t)n, P_l, n_u, best, em, sink_l, read, tid, NTHR, s_top, s_cf, s_ci, &s_cn, (uint32_t)a.top_max, dbg);
        if (!settled) continue;      // uniform; nothing was marked or handed over
        store_read_result(a, si.w, tid == 0 ? n_u : 0, best, starts, s_red);
        if (tid == 0) { a.B.sort_items[CLS][it].n = 0; if (!a.quiet) atomicAdd(&a.ctr->sh_top[SHARD()], 1u); }
        __syncthreads();
    }
}

// The same for the giant reads, after their sort and ahead of k_giant_chain (anchors and DP state in the arena).  Also without
// ChainSink::best (trace mode) it runs par_fill_block, so that k_giant_chain and k_cluster_dp only have the backtracks left: the read's
// table entry then carries the mark (qlen bit 31).
__global__ __launch_bounds__(512) void k_giant_top(K3Args a)
{
    __shared__ ParFillLds s_pf;
    __shared__ PfTile s_tile;
    __shared__ long long s_top[18];
    __shared__ uint32_t s_cf[TOPBT_MAX], s_ci[TOPBT_MAX], s_it;
    __shared__ int32_t s_cn, s_red[2];
    const ChainSink sink_l = a.sink; const ChainParams P_l = a.P;
    const uint32_t tid = threadIdx.x, nthr = blockDim.x;
    const uint32_t n_items = a.ctr->n_sort[SORT_CLS_GIANT];
    unsigned long long *const dbg = (a.dbg & 16) ? a.ctr->pf_dbg : nullptr;
    const bool top = a.emit && sink_l.best != nullptr && !(a.dbg & 512);
    for (;;) {
        if (tid == 0) s_it = atomicAdd(&a.ctr->top_ticket[SORT_CLS_GIANT], 1u);
        __syncthreads();
        const uint32_t it = s_it;
        __syncthreads();
        if (it >= n_items) break;
        const uint32_t slot = a.B.giant_order[it];
        const SortItem si = a.B.sort_items[SORT_CLS_GIANT][slot];
        const uint32_t n = si.n;
        if (n == 0) continue;
        const int32_t qlen = (int32_t)si.qlen;
        const bool in_b = giant_rounds(n) & 1;
        uint64_t *x = (in_b ? a.B.bx : a.B.ax) + si.off; uint32_t *q = (in_b ? a.B.bq : a.B.aq) + si.off;
        int32_t *f = a.B.af + si.off, *pt = (int32_t *)(a.B.az + si.off);
        const uint32_t mdx = chain_max_dist_x(P_l, qlen);
        uint32_t starts = 0;
        for (uint32_t i = tid; i < n; i += nthr) {
            bool start = i == 0;
            if (!start) { const uint64_t xi = x[i], xp = x[i - 1]; start = (uint32_t)(xi >> 32) != (uint32_t)(xp >> 32) || (uint32_t)xi - (uint32_t)xp > mdx; }
            if (start) { q[i] |= 0x80000000u; ++starts; }
        }
        __syncthreads();
        const bool pre = par_fill_tiled(x, q, f, pt, n, qlen, P_l, tid, nthr, s_pf, s_tile, a.pft_gmin);
        if (dbg && tid == 0) { atomicAdd(&dbg[pre ? 0 : 1], 1ull); atomicAdd(&dbg[2], (unsigned long long)s_pf.n_dirty); atomicAdd(&dbg[pre ? 3 : 4], (unsigned long long)n); }
        if (!pre) continue;
        if (tid == 0 && !a.quiet) { atomicAdd(&a.ctr->sh_pf_reads[SHARD()], 1u); if (s_pf.n_dirty) atomicAdd(&a.ctr->sh_pf_dirty[SHARD()], (uint32_t)s_pf.n_dirty); }
        bool settled = false;
        if (top && fix_dirty_clusters(x, q, f, pt, n, qlen, P_l, tid, nthr, s_pf, 512u, &s_cn)) {
            SliceStore S{(const uint64_t *)x, (const uint32_t *)q, f, pt};
            const uint32_t read = a.B.meta[si.w].r;
            const StoreEmit<SliceStore> em{&sink_l, &S, read, 0u, tid == 0, P_l.k, region_hash(qlen), qlen, nullptr};
            int32_t n_u = 0, best = 0;
            settled = backtrack_block_top(S, (int32_t)n, P_l, n_u, best, em, sink_l, read, tid, nthr, s_top, s_cf, s_ci, &s_cn, (uint32_t)a.top_max, dbg);
            if (settled) store_read_result(a, si.w, tid == 0 ? n_u : 0, best, starts, s_red);
        }
        if (tid == 0) {
            if (settled) { a.B.sort_items[SORT_CLS_GIANT][slot].n = 0; if (!a.quiet) atomicAdd(&a.ctr->sh_top[SHARD()], 1u); }      // nothing left for k_giant_chain or k_cluster_dp
            else a.B.sort_items[SORT_CLS_GIANT][slot].qlen = si.qlen | 0x80000000u;         // DP done: backtracks only
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(1024) void k_giant_scan(K3Args a)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63, n_items = a.ctr->n_sort[SORT_CLS_GIANT];
    __shared__ uint32_t s_cnt[33];
    if (tid < 33) s_cnt[tid] = 0;
    __syncthreads();
    if (tid < 64) {      // the tile table: a running sum, one wave
        uint32_t run = 0, max_n = 0;
        for (uint32_t base = 0; base < n_items; base += 64) {
            const uint32_t i = base + lane;
            const uint32_t n_i = i < n_items ? a.B.sort_items[SORT_CLS_GIANT][i].n : 0;
            max_n = n_i > max_n ? n_i : max_n;
            const uint32_t t = (n_i + GT - 1) / GT;
            const uint32_t ex = wave_excl_scan_u32(t, lane);
            if (i < n_items) a.B.tile_base[i] = run + ex;
            run += wave_sum_u32(t);
        }
        max_n = wave_all_max_u32(max_n);
        if (lane == 0) { a.B.tile_base[n_items] = run; a.ctr->n_giant_tiles = run; a.ctr->n_giant_rounds = giant_rounds(max_n); }
    }
    // largest first: a read of 500 k anchors drawn last would run on alone after every other block has finished.  Counting sort by
    // floor(log2 n), falling (the whole block); the order inside a size class does not matter.
    for (uint32_t i = tid; i < n_items; i += 1024) { const uint32_t n_i = a.B.sort_items[SORT_CLS_GIANT][i].n; atomicAdd(&s_cnt[n_i ? 32 - __clz((int)n_i) : 0], 1u); }
    __syncthreads();
    if (tid == 0) { uint32_t acc = 0; for (int b = 32; b >= 0; --b) { const uint32_t c = s_cnt[b]; s_cnt[b] = acc; acc += c; } }
    __syncthreads();
    for (uint32_t i = tid; i < n_items; i += 1024) { const uint32_t n_i = a.B.sort_items[SORT_CLS_GIANT][i].n; a.B.giant_order[atomicAdd(&s_cnt[n_i ? 32 - __clz((int)n_i) : 0], 1u)] = i; }
}

__device__ inline uint32_t giant_item_of(const uint32_t *tile_base, uint32_t n_items, uint32_t t)
{
    uint32_t lo = 0, hi = n_items;           // tile_base[lo] <= t < tile_base[hi]
    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (tile_base[mid] <= t) lo = mid; else hi = mid; }
    return lo;
}

__global__ __launch_bounds__(256) void k_giant_chunksort(K3Args a)
{
    __shared__ uint64_t s_x[2][GT];
    __shared__ uint32_t s_q[2][GT];
    const uint32_t tid = threadIdx.x, n_items = a.ctr->n_sort[SORT_CLS_GIANT], n_tiles = a.ctr->n_giant_tiles;
    for (uint32_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const uint32_t it = giant_item_of(a.B.tile_base, n_items, t);
        const SortItem si = a.B.sort_items[SORT_CLS_GIANT][it];
        const uint32_t c0 = (t - a.B.tile_base[it]) * GT, m = si.n - c0 < GT ? si.n - c0 : GT;
        uint64_t *gx = a.B.ax + si.off + c0; uint32_t *gq = a.B.aq + si.off + c0;
        for (uint32_t i = tid; i < m; i += 256) { s_x[0][i] = gx[i]; s_q[0][i] = gq[i]; }
        __syncthreads();
        const bool fl = block_merge_sort(&s_x[0][0], &s_q[0][0], &s_x[1][0], &s_q[1][0], m);
        const uint64_t *rx = fl ? s_x[1] : s_x[0]; const uint32_t *rq = fl ? s_q[1] : s_q[0];
        for (uint32_t i = tid; i < m; i += 256) { gx[i] = rx[i]; gq[i] = rq[i]; }
        __syncthreads();
    }
}

struct GiantTile { const uint64_t *sx; const uint32_t *sq; uint64_t *dx; uint32_t *dq; uint32_t L0, L1, R1, o0, o1; bool active; };

__device__ inline GiantTile giant_tile(const K3Args &a, uint32_t t, uint32_t n_items, uint32_t round)
{
    GiantTile g;
    const uint32_t it = giant_item_of(a.B.tile_base, n_items, t);
    const SortItem si = a.B.sort_items[SORT_CLS_GIANT][it];
    const uint32_t width = GT << round;
    g.active = width < si.n;                       // this read still has runs to merge in this round
    const bool src_b = round & 1;                  // after `round` rounds the data sits in buffer (round & 1)
    g.sx = (src_b ? a.B.bx : a.B.ax) + si.off; g.sq = (src_b ? a.B.bq : a.B.aq) + si.off;
    g.dx = (src_b ? a.B.ax : a.B.bx) + si.off; g.dq = (src_b ? a.B.aq : a.B.bq) + si.off;
    g.o0 = (t - a.B.tile_base[it]) * GT; g.o1 = g.o0 + GT < si.n ? g.o0 + GT : si.n;
    const uint32_t pb = g.o0 / (2 * width) * (2 * width);
    g.L0 = pb; g.L1 = pb + width < si.n ? pb + width : si.n; g.R1 = pb + 2 * width < si.n ? pb + 2 * width : si.n;
    return g;
}

// split[t] = number of elements the left run contributes before output o0 of tile t
__global__ __launch_bounds__(256) void k_giant_partition(K3Args a, uint32_t round)
{
    if (round >= a.ctr->n_giant_rounds) return;
    const uint32_t n_items = a.ctr->n_sort[SORT_CLS_GIANT], n_tiles = a.ctr->n_giant_tiles;
    for (uint32_t t = blockIdx.x * 256 + threadIdx.x; t < n_tiles; t += gridDim.x * 256) {
        const GiantTile g = giant_tile(a, t, n_items, round);
        if (!g.active) continue;
        const uint32_t lenL = g.L1 - g.L0, lenR = g.R1 - g.L1, d = g.o0 - g.L0;
        uint32_t lo = d > lenR ? d - lenR : 0, hi = d < lenL ? d : lenL;
        while (lo < hi) {
            uint32_t mid = (lo + hi) >> 1;
            if (g.sx[g.L0 + mid] <= g.sx[g.L1 + (d - 1 - mid)]) lo = mid + 1; else hi = mid;
        }
        a.B.tile_split[t] = lo;
    }
}

__global__ __launch_bounds__(256) void k_giant_merge(K3Args a, uint32_t round)
{
    __shared__ uint64_t s_x[GT];          // [0, la): left range, [la, la+lb): right range; la + lb <= GT
    __shared__ uint32_t s_q[GT];
    if (round >= a.ctr->n_giant_rounds) return;
    const uint32_t tid = threadIdx.x, n_items = a.ctr->n_sort[SORT_CLS_GIANT], n_tiles = a.ctr->n_giant_tiles;
    for (uint32_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const GiantTile g = giant_tile(a, t, n_items, round);
        if (!g.active) continue;                               // uniform per block
        const uint32_t d0 = g.o0 - g.L0, d1 = g.o1 - g.L0;
        const uint32_t a0 = a.B.tile_split[t];
        // the next tile of the same pair starts where this one ends; the pair's last tile ends at the run ends
        const bool last_of_pair = g.o1 == g.R1;
        const uint32_t a1 = last_of_pair ? g.L1 - g.L0 : a.B.tile_split[t + 1];
        const uint32_t b0 = d0 - a0, b1 = d1 - a1;
        const uint32_t la = a1 - a0, lb = b1 - b0;
        for (uint32_t i = tid; i < la; i += 256) { s_x[i] = g.sx[g.L0 + a0 + i]; s_q[i] = g.sq[g.L0 + a0 + i]; }
        for (uint32_t i = tid; i < lb; i += 256) { s_x[la + i] = g.sx[g.L1 + b0 + i]; s_q[la + i] = g.sq[g.L1 + b0 + i]; }
        __syncthreads();
        // merge in LDS: 8 outputs per thread, kept in registers until every thread has read its inputs, then through LDS again so that the
        // stores are coalesced (a thread storing its own 8 outputs touched 64 sectors per instruction: a quarter of the kernel's time)
        const uint32_t m = la + lb;
        constexpr uint32_t C = 8;
        uint64_t ox[C]; uint32_t oq[C];
        const uint32_t e0 = tid * C, e1 = e0 + C < m ? e0 + C : m;      // GT = 256 * C outputs at most
        if (e0 < m) {
            uint32_t lo = e0 > lb ? e0 - lb : 0, hi = e0 < la ? e0 : la;
            while (lo < hi) {
                uint32_t mid = (lo + hi) >> 1;
                if (s_x[mid] <= s_x[la + (e0 - 1 - mid)]) lo = mid + 1; else hi = mid;
            }
            uint32_t ia = lo, ib = e0 - lo;
            uint64_t va = ia < la ? s_x[ia] : ~0ull, vb = ib < lb ? s_x[la + ib] : ~0ull;
#pragma unroll
            for (uint32_t u = 0; u < C; ++u) {
                if (e0 + u < e1) {
                    const bool takeL = ia < la && (ib >= lb || va <= vb);
                    if (takeL) { ox[u] = va; oq[u] = s_q[ia]; ++ia; va = ia < la ? s_x[ia] : ~0ull; }
                    else { ox[u] = vb; oq[u] = s_q[la + ib]; ++ib; vb = ib < lb ? s_x[la + ib] : ~0ull; }
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (uint32_t u = 0; u < C; ++u) if (e0 + u < e1) { s_x[e0 + u] = ox[u]; s_q[e0 + u] = oq[u]; }
        __syncthreads();
        for (uint32_t i = tid; i < m; i += 256) { g.dx[g.o0 + i] = s_x[i]; g.dq[g.o0 + i] = s_q[i]; }
        __syncthreads();
    }
}

// one block per giant read: clusters chained over arena slices of the buffer its sort ended in
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_giant_chain(K3Args a, int phase)      // 128 VGPRs: two blocks per CU (at 133 it is one)
{
    __shared__ int32_t s_found, s_red[2], s_bcount;
    __shared__ uint32_t s_bstart[2048], s_blen[2048], s_nxt[1024];
    // local copies: a pointer or reference INTO the argument struct makes the compiler copy all ~640 B of it to scratch at entry and read
    // every a.X from there afterwards
    const ChainSink sink_l = a.sink; const ChainParams P_l = a.P;
    const uint32_t tid = threadIdx.x, nthr = blockDim.x;
    const uint32_t n_items = a.ctr->n_sort[SORT_CLS_GIANT];
    __shared__ uint32_t s_it;
    for (;;) {
        if (tid == 0) s_it = atomicAdd(&a.ctr->sort_ticket[SORT_CLS_GIANT], 1u);      // a ticket per read (reset before every launch): giant reads differ a hundredfold
        __syncthreads();
        const uint32_t it = s_it;
        __syncthreads();
        if (it >= n_items) break;
        SortItem si = a.B.sort_items[SORT_CLS_GIANT][a.B.giant_order[it]];
        const uint32_t n = si.n;
        if (n == 0) continue;                 // decided by k_group_probe
        bool pre = (si.qlen >> 31) != 0;      // k_giant_top's par_fill_block applied: f, p and the dirty marks are there
        si.qlen &= 0x7fffffffu;
        const bool in_b = giant_rounds(n) & 1;
        uint64_t *sx = (in_b ? a.B.bx : a.B.ax) + si.off; uint32_t *sq = (in_b ? a.B.bq : a.B.aq) + si.off;
        if (tid == 0) s_found = 0;
        __syncthreads();
        int32_t n_u = 0, best = 0; uint32_t n_cl = 0;
        const GlobalQ gq{a.B.tabs->cl_items, a.B.tabs->cl_cap, a.ctr->n_cl, si.w, in_b ? 1u : 0u, si.off};
        if (!(a.dbg & 2))
        chain_sorted<true>(sx, sq, a.B.af + si.off, (int32_t *)(a.B.az + si.off), n, tid, nthr, (int32_t)si.qlen, P_l,
                     a.flag_only ? &s_found : nullptr, BigList{s_bstart, s_blen, &s_bcount, 2048}, n_u, best, n_cl, &gq, s_nxt,
                     a.emit ? &sink_l : nullptr, a.B.meta[si.w].r, a.emit ? (in_b ? a.B.ax : a.B.bx) + si.off : nullptr,      // heap: the sort's other buffer
                     nullptr, 0u, phase, TandemQ{nullptr, 0u, 1u, 0, 0u}, nullptr, &pre);
        store_read_result(a, si.w, n_u, best, n_cl, s_red, phase == 1);
        __syncthreads();
    }
}

// The big clusters of all giant reads, largest class first, one wave per cluster (persistent waves drawing tickets).
// Results are merged into the per-read accumulators k_giant_chain stored.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(7, 7))) void k_cluster_dp(K3Args a)
{
    __shared__ RingMem s_ring[4];
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // the four class counts in scalars (an array indexed by the class would live in scratch)
    const uint32_t cnt0 = min(a.ctr->n_cl[0], a.B.cl_cap[0]), cnt1 = min(a.ctr->n_cl[1], a.B.cl_cap[1]);
    const uint32_t cnt2 = min(a.ctr->n_cl[2], a.B.cl_cap[2]), cnt3 = min(a.ctr->n_cl[3], a.B.cl_cap[3]);
    const uint32_t total = cnt0 + cnt1 + cnt2 + cnt3;
    uint32_t n_cl = 0;
    // One address takes ~90 M atomics/s: a ticket per cluster (1.6 M of them per step of the bench workload, most of 65..256 anchors) plus two
    // statistics counters per cluster kept every wave queueing at the L2 - the kernel took 40 ms with the DP itself switched off.  Classes 0-2
    // (few, large) are drawn one by one, class 3 eight at a time; the per-class statistics are debug output only.
    const uint32_t big = cnt0 + cnt1 + cnt2;
    uint32_t t3 = 0, t3_end = 0;
    bool big_done = big == 0;
    for (;;) {
        uint32_t t = 0;
        int c;
        const SortItem *items;
        if (!big_done) {
            if (lane == 0) t = atomicAdd(&a.ctr->cl_ticket, 1u);
            t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
            if (t >= big) { big_done = true; continue; }
            if (t < cnt0) { c = 0; items = a.B.cl_items[0]; }
            else if (t < cnt0 + cnt1) { c = 1; t -= cnt0; items = a.B.cl_items[1]; }
            else { c = 2; t -= cnt0 + cnt1; items = a.B.cl_items[2]; }
        } else {
            if (t3 >= t3_end) {
                if (lane == 0) t3 = atomicAdd(&a.ctr->cl_ticket3, 8u);
                t3 = (uint32_t)__builtin_amdgcn_readfirstlane((int)t3);
                if (t3 >= cnt3) break;
                t3_end = min(t3 + 8u, cnt3);
            }
            t = t3++; c = 3; items = a.B.cl_items[3];
        }
        const SortItem ci = items[t];
        if (a.flag_only && __atomic_load_n(&a.B.acc_nu[ci.w], __ATOMIC_RELAXED) > 0) continue;      // the read is decided
        const bool in_b = (ci.pad & 1u) != 0;
        const uint64_t *gx = (in_b ? a.B.bx : a.B.ax) + ci.off; uint32_t *gq = (in_b ? a.B.bq : a.B.aq) + ci.off;
        uint64_t *heap = a.emit ? (in_b ? a.B.ax : a.B.bx) + ci.off : (uint64_t *)gx;      // hand-over: the anchors must survive the backtrack
        const ChainSink *sk = a.emit ? &a.sink : nullptr;
        const uint32_t rd = a.emit ? a.B.meta[ci.w].r : 0u, cbase = ci.pad >> 1;
        int32_t n_u, best;
        const bool pre = (ci.qlen >> 31) != 0;      // f and p are par_fill_block's
        if ((a.dbg & 16) && lane == 0) { atomicAdd(&a.ctr->pf_dbg[pre ? 5 : 6], 1ull); if (!pre) atomicAdd(&a.ctr->pf_dbg[7], (unsigned long long)ci.n); }
        const int32_t cqlen = (int32_t)(ci.qlen & 0x7fffffffu);
        if (a.P.max_iter <= RING_TMAX_ITER)
            chain_cluster_ring(gx, gq, a.B.af + ci.off, (int32_t *)(a.B.az + ci.off), (int32_t)ci.n, cqlen, a.P, n_u, best,
                               a.flag_only != 0, lane, s_ring[wv], (a.dbg & 16) ? a.ctr->cl_dbg : nullptr, sk, a.emit ? rd : 0u, cbase, heap, pre);
        else {      // look-back windows beyond the LDS mark bitmap: DP state in the arena throughout
            SliceStore S{gx, gq, a.B.af + ci.off, (int32_t *)(a.B.az + ci.off), pre ? (int32_t)cbase : 0};
            chain_cluster_wave(S, (int32_t)ci.n, cqlen, a.P, heap, n_u, best, a.flag_only != 0, lane, sk, rd, cbase, pre);
        }
        ++n_cl;
        if ((a.dbg & 16) && lane == 0) { atomicAdd(&a.ctr->cl_tot[c], 1ull); atomicAdd(&a.ctr->cl_anchor_tot[c], (unsigned long long)ci.n); }
        if (lane == 0 && n_u > 0) { atomicAdd(&a.B.acc_nu[ci.w], n_u); atomicMax(&a.B.acc_best[ci.w], best); }
    }
    if (lane == 0 && n_cl) atomicAdd(&a.ctr->sh_clusters[(blockIdx.x * 4 + wv) & 63], n_cl);
}

__global__ void k_finalize(K3Args a)
{
    const uint32_t n_items = *a.list_count;
    uint32_t n_host_thr = 0;
    for (uint32_t w0 = blockIdx.x * blockDim.x; w0 < n_items; w0 += gridDim.x * blockDim.x) {
        const uint32_t w = w0 + threadIdx.x;
        if (w >= n_items) continue;
        const BigMeta m = a.B.meta[w];
        if (m.state != 0) continue;
        const int32_t n_u = a.B.acc_nu[w], best = a.B.acc_best[w];
        if (a.pass == 0 && n_u == 0 && a.P.max_occ > a.P.mid_occ && m.rep_len > 0) {
            uint32_t i = atomicAdd(a.next_count, 1u);
            a.next_list[i] = m.r;
            continue;
        }
        const uint32_t info = a.k1info[m.r];
        int32_t fl = n_u > 0;
        a.flags[m.r] = (uint8_t)fl;
        write_trace(a.trace, m.r, (int32_t)(info & 0xffffu), (int32_t)(a.seed_off ? info >> 16 : (info >> 16 & 0x7fffu)), (int32_t)m.n_a, m.rep_len, a.pass, n_u, best, fl);
        n_host_thr += (uint32_t)fl;
    }
    if (n_host_thr && a.quiet != 1) atomicAdd(&a.ctr->sh_host[(blockIdx.x + threadIdx.x) & 63], n_host_thr);      // quiet == 2: reads whose earlier visit was taken back (k_lext_forget)
}

__device__ inline uint8_t *arena_alloc(const K2Args &a, size_t bytes)
{
    // compare-and-swap bump: a request that does not fit leaves the cursor alone, so the reads that already hold
    // their sketch buffers can still get their anchor slices (no allocation deadlock when the arena is short)
    bytes = (bytes + 15) & ~(size_t)15;
    unsigned long long cur = a.ctr->arena_cursor;
    for (;;) {
        if (cur + bytes > a.arena_bytes) return nullptr;
        unsigned long long prev = atomicCAS(&a.ctr->arena_cursor, cur, cur + (unsigned long long)bytes);
        if (prev == cur) return a.arena + cur;
        cur = prev;
    }
}

// legacy lane-per-read path (re-sketch of reads K1 could not finish); arrays in the HBM arena
__global__ __launch_bounds__(64) void k_chain_large(K2Args a)
{
    const uint32_t lane = threadIdx.x;
    const uint32_t n_work = *a.work_count;
    for (uint32_t base = a.work_begin + blockIdx.x * 64; base < n_work; base += gridDim.x * 64) {
        const uint32_t wi = base + lane;
        bool host = false;
        if (wi < n_work) {
            const uint32_t r = a.work[wi] & 0x7fffffffu;
            const uint64_t o_beg = a.offsets[r];
            const int32_t qlen = (int32_t)(a.offsets[r + 1] - o_beg);
            int32_t n_mini = 0, n_seed = 0;
            SeedView sv;
            bool defer = false;
            {
                const size_t cap = 2 * (size_t)qlen + 256;
                // records (16 B) + minimizers (hash, y: 12 B) + sorted hashes (8 B) per possible minimizer, + the ring
                uint8_t *m = arena_alloc(a, cap * (16 + 8 + 4 + 8) + (size_t)a.w * 16);
                if (!m) defer = true;
                else {
                    uint4 *recs = (uint4 *)m;
                    uint64_t *mh = (uint64_t *)(m + cap * 16), *hs = mh + cap;
                    uint32_t *my = (uint32_t *)(hs + cap);
                    uint64_t *rbx = (uint64_t *)(my + cap);
                    uint32_t *rby = (uint32_t *)(rbx + a.w);
                    SketchStateDyn st;
                    st.init(rbx, rby, a.w, a.P.k);
                    int32_t nm = 0;
                    auto emit = [&](uint64_t x, uint32_t y) { mh[nm] = x >> 8; my[nm] = y; ++nm; };
                    for (int32_t i = 0; i < qlen; ++i) st.step(sh_nt4(a.bases[o_beg + i]), (uint32_t)i, emit);
                    if (qlen > 0) st.finish(emit);
                    // mm_seed_mz_flt: drop minimizers whose hash repeats within the query more than
                    // q_occ_max (= mid_occ) times and more than q_occ_frac of all minimizers
                    bool thin = a.P.q_occ_frac > 0.0f && a.P.mid_occ > 0 && nm > a.P.mid_occ;
                    if (thin) {
                        for (int32_t i = 0; i < nm; ++i) hs[i] = mh[i];
                        auto down = [&](int32_t i, int32_t mlen) {
                            uint64_t tmp = hs[i]; int32_t kk = i;
                            while ((kk = (kk << 1) + 1) < mlen) { if (kk != mlen - 1 && hs[kk] < hs[kk + 1]) ++kk; if (hs[kk] < tmp) break; hs[i] = hs[kk]; i = kk; }
                            hs[i] = tmp;
                        };
                        for (int32_t i = (nm >> 1) - 1; i >= 0; --i) down(i, nm);
                        for (int32_t e = nm - 1; e > 0; --e) { uint64_t t0 = hs[0]; hs[0] = hs[e]; hs[e] = t0; down(0, e); }
                    }
                    const uint64_t slot_mask = (1ULL << a.lg_slots) - 1;
                    uint64_t prev_key = ~0ull;
                    for (int32_t i = 0; i < nm; ++i) {
                        const uint64_t key = mh[i];
                        if (thin) {     // occurrences of this hash in the query, from the sorted copy
                            int32_t lo = 0, hi = nm;
                            while (lo < hi) { int32_t mid = (lo + hi) >> 1; if (hs[mid] < key) lo = mid + 1; else hi = mid; }
                            int32_t lb = lo; hi = nm;
                            while (lo < hi) { int32_t mid = (lo + hi) >> 1; if (hs[mid] <= key) lo = mid + 1; else hi = mid; }
                            const int32_t cnt = lo - lb;
                            if (cnt > a.P.mid_occ && (float)cnt > (float)nm * a.P.q_occ_frac) continue;
                        }
                        ++n_mini;
                        const uint32_t same = key == prev_key ? SH_REC_PREV_SAME : 0u;
                        prev_key = key;
                        uint64_t idx = sh_slot_home(key, a.lg_slots);
                        uint4 s = a.slots[idx];
                        uint64_t w0 = (uint64_t)s.y << 32 | s.x;
                        while (w0 != SH_SLOT_EMPTY && (w0 & SH_SLOT_KEYMASK) != key) {
                            idx = (idx + 1) & slot_mask; s = a.slots[idx]; w0 = (uint64_t)s.y << 32 | s.x;
                        }
                        if (w0 != SH_SLOT_EMPTY) {
                            uint32_t occ = (w0 & SH_SLOT_MULTI) ? (s.z & (uint32_t)SH_SLOT_NMASK) : 1u;
                            recs[n_seed++] = make_uint4(s.z, s.w, occ | same, my[i]);
                        }
                    }
                    sv.base = recs; sv.stride = 1; sv.n = (uint32_t)n_seed;
                }
            }
            if (!defer && qlen == 0) {
                a.flags[r] = 2;
                write_trace(a.trace, r, 0, 0, 0, 0, 0, 0, 0, 2);
            } else if (!defer) {
                int32_t max_occ = a.P.mid_occ, rechained = 0, n_u = 0, best = 0, rep_len = 0;
                int64_t n_a = 0;
                for (;;) {
                    seed_filter(sv, qlen, max_occ, a.P, n_a, rep_len);
                    LargeStore S;
                    uint8_t *m = arena_alloc(a, LargeStore::bytes_for(n_a));
                    if (!m) { defer = true; break; }
                    S.carve(m, n_a);
                    gen_anchors(S, sv, a.positions, qlen, a.P.k);
                    chain_dp<LargeStore, int64_t>(S, n_a, qlen, a.P);
                    if (a.emit) {
                        const TandemQ tq{sv.base, sv.n, sv.stride, qlen, 1u};
                        const StoreEmit<LargeStore> em{&a.sink, &S, r, 0u, true, a.P.k, region_hash(qlen), qlen, &tq};
                        backtrack_heap<LargeStore, int64_t, StoreEmit<LargeStore>>(S, n_a, a.P, S.z, n_u, best, false, em);
                    } else
                    backtrack_heap<LargeStore, int64_t>(S, n_a, a.P, S.z, n_u, best);
                    if (!rechained && n_u == 0 && a.P.max_occ > a.P.mid_occ && rep_len > 0) { rechained = 1; max_occ = a.P.max_occ; continue; }
                    break;
                }
                if (!defer) {
                    finish_read(a, r, n_mini, n_seed, n_a, rep_len, rechained, n_u, best);
                    host = n_u > 0;
                }
            }
            if (defer) {
                uint32_t li = atomicAdd(&a.ctr->n_defer, 1u);
                a.work_defer[li] = r;
            }
        }
        uint64_t mh = __ballot(host);
        if (lane == 0 && mh && a.quiet != 1) atomicAdd(&a.ctr->sh_host[SHARD()], (uint32_t)__popcll(mh));
    }
}

// ------------------------------------------------------------------------------------------------
// extension stage (SH_F_CIGAR; sh_align.h): reads with at least one chain, one wave each
// ------------------------------------------------------------------------------------------------
struct ExtArgs {
    AlignIn in; AlignParams P;
    uint8_t *scratch; unsigned long long scratch_per_wave; uint32_t max_read_len, reg_cap;
    uint32_t *list; uint32_t *n_list, *ticket; Counters *ctr; uint8_t *flags; sh_trace *trace; int32_t flag_only; uint64_t n_reads;
    const unsigned long long *best; const uint32_t *tie;      // first pass of a flag-only call: decide from the top chain, or queue for the full pass
    uint32_t *redo, *redo2; int32_t top_only;                 // k_regs_align, top_only: align regs[0] alone; reads it does not settle go to redo2
    uint32_t *unres_list, *n_unres;                           // k_regs_align: reads with more chains / regions than this launch's working memory holds (redone with more)
};

__global__ void k_ext_list(ExtArgs a)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool has = r < a.n_reads && a.in.head[r] != ~0u;
    const uint32_t li = wave_append(a.n_list, has);
    if (has) a.list[li] = (uint32_t)r;
}

// flag-only first pass, one LANE per read of the list: regs[0] settles nearly every read (top_chain_settles); the others are queued
__global__ __launch_bounds__(64) void k_ext_top(ExtArgs a)
{
    const uint32_t n_list = *a.n_list;
    uint32_t n_redo_wave = 0;
    for (uint32_t base = blockIdx.x * 64; base < n_list; base += gridDim.x * 64) {
        const uint32_t t = base + threadIdx.x;
        bool redo = false;
        uint32_t r = 0;
        int32_t rc = 0;
        MidReq mid{0, 0, 0, 0, 0};
        uint32_t h_top = ~0u;
        if (t < n_list) {
            r = a.list[t];
            rc = top_chain_settles(a.in, a.P, r, a.best[r], a.tie[r], mid, &h_top);
        }
        {   // rc == 2: the stretch's k-mers leave too many bases uncovered - mm_test_zdrop on the bases, the wave on one lane's request at a time
            ChainParams cp{}; cp.ext_a = a.P.a < 0 ? -a.P.a : a.P.a; cp.ext_b = a.P.b > 0 ? -a.P.b : a.P.b; cp.ext_amb = a.P.sc_ambi > 0 ? -a.P.sc_ambi : a.P.sc_ambi; cp.ext_zdrop = a.P.zdrop;
            const BaseCtx bcx{a.in.ref, a.in.cstart, a.in.bases};
            const int32_t ql = t < n_list ? (int32_t)(a.in.offsets[r + 1] - a.in.offsets[r]) : 0;
            const bool mid_ok = resolve_mid_wave(rc == 2, mid, a.in.bases + a.in.offsets[t < n_list ? r : 0], ql, bcx, cp);
            if (rc == 2) rc = mid_ok ? 1 : -4;
        }
        if (t < n_list) {
            if (rc > 0) a.flags[r] = 1;
            else {
                redo = true; atomicAdd(&a.ctr->ext_reason[-rc & 7], 1u);
                // k_regs_align (top_only) wants this record alone: the list now starts there (a read it does not settle starts over with an empty list)
                if (h_top != ~0u) const_cast<uint32_t *>(a.in.head)[r] = h_top;
            }
        }
        const uint32_t li = wave_append(&a.ctr->ext_n_redo, redo);
        if (redo) a.redo[li] = r;
        n_redo_wave += (uint32_t)__popcll(__ballot(redo));
    }
    (void)n_redo_wave;
}

// reads queued for the full pass start over with an empty chain list
__global__ void k_ext_reset(ExtArgs a)
{
    const uint32_t n = *a.n_list;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) ((uint32_t *)a.in.head)[a.list[i]] = ~0u;
}

__global__ __launch_bounds__(64) void k_regs_align(ExtArgs a)
{
    __shared__ AlignLds Ls;
    __shared__ uint32_t s_ovf;
    const uint32_t lane = threadIdx.x;
    AlignScratch A;
    align_scratch_carve(A, a.scratch + (unsigned long long)blockIdx.x * a.scratch_per_wave, a.max_read_len, a.reg_cap);
    const uint32_t n_list = *a.n_list;
    uint32_t n_regions = 0, n_dropped = 0;
    for (;;) {
        uint32_t t = 0;
        if (lane == 0) { t = atomicAdd(a.ticket, 1u); s_ovf = 0; }
        t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
        if (t >= n_list) break;
        __syncthreads();
        const uint32_t r = a.list[t];
        AlignOut o;
        if (!align_read_wave(a.in, a.P, r, a.flag_only != 0, A, Ls, o, &s_ovf, a.top_only ? a.best[r] : 0ull)) {
            // minimap2 has no such limits.  A read with more chains / primaries / regions than the working memory holds (codes 2, 3, 6) keeps
            // its chain-level answer and is counted (sh_stats.n_ext_unresolved; the host warns); anything else is a sizing error of the call
            if (lane == 0) {
                const uint32_t code = s_ovf;
                if ((code == 2u || code == 3u || code == 6u) && a.unres_list) a.unres_list[atomicAdd(a.n_unres, 1u)] = r;      // once more, with memory sized for it
                else if (code == 2u || code == 3u || code == 6u) {
                    a.flags[r] = 1;
                    if (a.trace) ((int32_t *)(a.trace + r))[7] = 1;
                    atomicAdd(&a.ctr->lext_unresolved, 1u); atomicExch(&a.ctr->lext_err_read, r); atomicExch(&a.ctr->lext_err_code, code);
                } else atomicExch(&a.ctr->ext_overflow, code ? code : 7u);
            }
            __syncthreads();
            continue;
        }
        if (a.top_only) {      // regs[0] alone: a survivor settles the read, otherwise every chain is needed
            if (lane == 0) {
                if (o.n_regs > 0) a.flags[r] = 1;
                else a.redo2[atomicAdd(&a.ctr->ext_n_redo2, 1u)] = r;
            }
            n_regions += (uint32_t)o.n_aligned;
            __syncthreads();
            continue;
        }
        if (lane == 0) {
            a.flags[r] = o.n_regs > 0 ? 1 : 0;
            if (a.trace) {
                int32_t *tr = (int32_t *)(a.trace + r);
                tr[7] = o.n_regs > 0;
                ((int4 *)tr)[2] = make_int4(o.n_aligned, o.n_regs, o.dp_max, (int32_t)o.sig);
            }
        }
        n_regions += (uint32_t)o.n_aligned; n_dropped += o.n_regs == 0;
        __syncthreads();
    }
    if (lane == 0) { if (n_regions) atomicAdd(&a.ctr->ext_regions, n_regions); if (n_dropped) atomicAdd(&a.ctr->ext_dropped, n_dropped); }
}

// regs[0] alone (flag-only second pass): one chain, a handful of regions - a third less LDS per wave than k_regs_align, half as many waves
// again per CU.  A read that outgrows the small arrays is not an error here: it joins the reads regs[0] did not settle (redo2) and
// meets the full-size kernel there.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_regs_align_top(ExtArgs a)
{
    __shared__ AlignLdsTop Ls;
    __shared__ uint32_t s_ovf;
    const uint32_t lane = threadIdx.x;
    AlignScratch A;
    align_scratch_carve(A, a.scratch + (unsigned long long)blockIdx.x * a.scratch_per_wave, a.max_read_len, a.reg_cap);
    const uint32_t n_list = *a.n_list;
    uint32_t n_regions = 0;
    for (;;) {
        uint32_t t = 0;
        if (lane == 0) { t = atomicAdd(a.ticket, 1u); s_ovf = 0; }
        t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
        if (t >= n_list) break;
        __syncthreads();
        const uint32_t r = a.list[t];
        AlignOut o;
        const bool done = align_read_wave(a.in, a.P, r, true, A, Ls, o, &s_ovf, a.best[r]);
        if (lane == 0) {
            if (done && o.n_regs > 0) a.flags[r] = 1;
            else a.redo2[atomicAdd(&a.ctr->ext_n_redo2, 1u)] = r;
        }
        if (done) n_regions += (uint32_t)o.n_aligned;
        __syncthreads();
    }
    if (lane == 0 && n_regions) atomicAdd(&a.ctr->ext_regions, n_regions);
}


// ------------------------------------------------------------------------------------------------
// extension stage, long-read presets (sh_long.h): one wave per read with at least one chain, two kernels
// ------------------------------------------------------------------------------------------------
struct ExtLongArgs {
    LongIn I; LongParams P; AlignParams AP; ChainParams CP;
    uint8_t *scratch; unsigned long long scratch_per_wave; LongSizes sz; LongArena AR;
    uint32_t *list; uint32_t *n_list, *ticket; uint32_t *big_list; uint32_t *n_big; Counters *ctr; uint8_t *flags; sh_trace *trace; int32_t flag_only, clk, probe;
    const uint32_t *hist; int32_t bin_cut, part;      // the size-ordered list's giants (bins >= bin_cut) come first: part 1 = all but them, 2 = only them, 0 = the whole list
    uint32_t *exact_list, *n_exact;                   // reads whose long join must run on the literal trees (lr_chains_wave returns 6)
    uint32_t *exact_list2, *n_exact2;                 // ... those of them that needed the 4096-anchor ring already: straight to the exact pass with that ring
    uint32_t *unres_list, *n_unres;                   // reads that outgrew the large working memory too (with big_list == nullptr): redone with memory sized for them
    const uint32_t *drop; uint32_t *fb_list, *n_fb;   // k_lr_locus: what was left out of a read's anchors; reads that must be redone with every anchor
    uint32_t *started;                                // counts the blocks that have begun (the giants' grid: the main grid is launched once they hold their LDS)
    const uint32_t *follow_done;                      // k_regs_align_long beside the launch that fills its list: set when that launch has ended
    LongCoop coop; int32_t coop_on;                   // k_long_chains, EXACT instance: the launch's waves share the long join of its largest reads (lr_coop_fill)
    uint8_t *kind;                                    // per read of the call: which of the stage's rarer paths it took (LK_*; sh_ctx_debug_list, the bench's stratified oracle check)
};
// bits of ExtLongArgs::kind
enum { LK_EXACT = 1, LK_RMQ_OPEN = 2, LK_UNRESOLVED = 4, LK_LOCUS_REDONE = 8, LK_ONDEMAND = 16, LK_FULL = 32, LK_SECOND_SIZE = 64, LK_ONE_LANE_TREES = 128 };
__device__ inline void lk_mark(const ExtLongArgs &a, uint32_t r, uint8_t bit) { if (a.kind) a.kind[r] |= bit; }      // (one wave owns a read at a time)

// Largest reads first: a read's cost grows with its chain anchors (one with 70 k of them keeps a wave busy for a third of a second), and a
// kernel ends with its slowest wave.  Reads are binned by log2 of their chain-anchor count and listed from the top bin down.
__global__ void k_lext_bins(const ChainRec *recs, const uint32_t *head, const uint32_t *list, const uint32_t *n_list, uint32_t *size_of, uint32_t *hist, const uint64_t *offsets)
{
    const uint32_t n = *n_list;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t r = list[i];
        unsigned long long tot = 0;
        for (uint32_t h = head[r]; h != ~0u; h = recs[h].next) tot += recs[h].cnt;
        // a long read's join looks back over a window of bw_long reference bases: beyond the LDS ring every anchor scans blocks in HBM, and the
        // cost per anchor grows with the read - half its length stands in when that is more than its anchors
        const unsigned long long ql = (offsets[r + 1] - offsets[r]) >> 1;
        if (ql > tot) tot = ql;
        const uint32_t b = tot ? 63u - (uint32_t)__clzll(tot) : 0u;
        size_of[i] = b < 31u ? b : 31u;
        atomicAdd(&hist[size_of[i]], 1u);
    }
}
__global__ void k_lext_scan(uint32_t *hist)      // hist[32] -> start of each bin, largest bin first; hist[32 + b] = running cursor
{
    uint32_t acc = 0;
    for (int b = 31; b >= 0; --b) { const uint32_t c = hist[b]; hist[32 + b] = acc; acc += c; }
}
__global__ void k_lext_scatter(const uint32_t *list, const uint32_t *n_list, const uint32_t *size_of, uint32_t *hist, uint32_t *out)
{
    const uint32_t n = *n_list;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[atomicAdd(&hist[32 + size_of[i]], 1u)] = list[i];
}

// the stage's list: every read with a chain; a read without one among the anchors k_lr_locus kept, but with a cluster left out that could
// hold one (drop >= min_cnt), is redone with every anchor
__global__ void k_lext_list(ExtArgs a, const uint32_t *drop, uint32_t min_cnt, uint32_t *fb_list, uint32_t *n_fb)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool has = r < a.n_reads && a.in.head[r] != ~0u;
    const uint32_t li = wave_append(a.n_list, has);
    if (has) a.list[li] = (uint32_t)r;
    const bool redo = r < a.n_reads && !has && drop != nullptr && drop[r] >= min_cnt;
    const uint32_t fi = wave_append(n_fb, redo);
    if (redo) { fb_list[fi] = (uint32_t)r; atomicAdd(&a.ctr->lr_fb_why[6], 1u); }
}
// second round: the reads of `from` that have a chain now
__global__ void k_lext_list_from(ExtArgs a, const uint32_t *from, const uint32_t *n_from)
{
    const uint32_t n = *n_from;
    for (uint32_t i0 = blockIdx.x * blockDim.x; i0 < n; i0 += gridDim.x * blockDim.x) {
        const uint32_t i = i0 + threadIdx.x;
        const uint32_t r = i < n ? from[i] : 0u;
        const bool has = i < n && a.in.head[r] != ~0u;
        const uint32_t li = wave_append(a.n_list, has);
        if (has) a.list[li] = r;
    }
}
// reads about to be chained again: no chains, nothing left out
__global__ void k_lext_forget(const uint32_t *list, uint32_t n, uint32_t *head, uint32_t *drop, uint32_t *n_had_chain)
{
    uint32_t had = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) { const uint32_t r = list[i]; had += head[r] != ~0u; head[r] = ~0u; drop[r] = 0u; }
    if (had) atomicAdd(n_had_chain, had);      // k_finalize counted them as mapped at chain level; it will count them again
}

// no memory could be had for these reads: they keep their chain-level answer (mapped) and are counted (sh_stats.n_ext_unresolved; the host warns)
__global__ void k_lext_giveup(const uint32_t *list, uint32_t n, uint8_t *flags, sh_trace *trace, LongHdr *hdr, Counters *ctr, uint8_t *kind)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t r = list[i];
        flags[r] = 1;
        if (kind) kind[r] |= 4;      // LK_UNRESOLVED
        if (trace) ((int32_t *)(trace + r))[7] = 1;
        LongHdr h{0ull, -1, 0, 0, 0}; hdr[r] = h;
        atomicAdd(&ctr->lext_unresolved, 1u);
    }
}

// a read that outgrew the pass: to the pass with the large working memory, or - beyond that too - it keeps its chain-level answer and is counted
__device__ inline void lext_defer(const ExtLongArgs &a, uint32_t r, uint32_t code, bool has_hdr)
{
    if (a.big_list) {      // (a launch of the second size may be reading the list while it grows: what this wave wrote for the read first, then the entry)
        lk_mark(a, r, code == 16u + 7u ? (LK_SECOND_SIZE | LK_ONE_LANE_TREES) : LK_SECOND_SIZE);
        __threadfence();
        __atomic_store_n(&a.big_list[atomicAdd(a.n_big, 1u)], r, __ATOMIC_RELAXED);
        return;
    }
    if (a.unres_list) {
        lk_mark(a, r, LK_ONDEMAND);
        a.unres_list[atomicAdd(a.n_unres, 1u)] = r;
        if (!has_hdr) { LongHdr h{0ull, -1, 0, 0, 0}; a.AR.hdr[r] = h; }      // the regions kernel must not take it before the chains kernel has
        atomicExch(&a.ctr->lext_err_read, r); atomicExch(&a.ctr->lext_err_code, code);
        return;
    }
    a.flags[r] = 1;
    lk_mark(a, r, LK_UNRESOLVED);
    if (a.trace) ((int32_t *)(a.trace + r))[7] = 1;
    if (!has_hdr) { LongHdr h{0ull, -1, 0, 0, 0}; a.AR.hdr[r] = h; }
    atomicAdd(&a.ctr->lext_unresolved, 1u); atomicExch(&a.ctr->lext_err_read, r); atomicExch(&a.ctr->lext_err_code, code);
}

// a read k_lr_locus thinned out and whose answer may depend on what was left out: to the list of reads that take the complete path
__device__ inline void lext_redo(const ExtLongArgs &a, uint32_t r, uint32_t why)
{
    a.fb_list[atomicAdd(a.n_fb, 1u)] = r;
    lk_mark(a, r, LK_LOCUS_REDONE);
    LongHdr h{0ull, -1, 0, 0, 0}; a.AR.hdr[r] = h;
    atomicAdd(&a.ctr->lr_fb_why[why >= 40u && why < 48u ? why - 40u : 7u], 1u);
}

__global__ void k_set_word(uint32_t *p, uint32_t v) { if (threadIdx.x == 0 && blockIdx.x == 0) __atomic_store_n(p, v, __ATOMIC_RELAXED); }

// holds a stream until `want` blocks of a kernel on another stream have begun, or ~2 ms have passed (one lane, sleeping between looks)
__global__ void k_wait_started(const uint32_t *cnt, uint32_t want, uint32_t max_looks)
{
    if (threadIdx.x == 0)
        for (uint32_t i = 0; i < max_looks; ++i) {
            if (__atomic_load_n(cnt, __ATOMIC_RELAXED) >= want) break;
            __builtin_amdgcn_s_sleep(64);
        }
}

template <int NR, bool EXACT, bool FAT>
__global__ __launch_bounds__(64) void k_long_chains(ExtLongArgs a)
{
    __shared__ RmqLdsT<NR, FAT> RL;
    // EXACT: the long join's main tree in LDS beside the ring (sh_rmq_tree.h).  It holds the look-back window only (max_gap reference bases:
    // at most ~1 500 anchors on the bench's satellite reads); a window beyond it sends the read to the instance with the larger ring and tree,
    // and from there to the one-lane version over node pools in HBM.
    constexpr int TCAP = EXACT ? (NR >= 4096 ? 1792 : 1664) : 1;      // 32 B a node: 79 KB of LDS with the 1024-anchor ring (two waves to a CU), 156 KB with the 4096-anchor ring
    __shared__ RqLdsMem<TCAP> TM;
    RqLds TL{};
    if (EXACT) TL.init(TM);
    const uint32_t lane = threadIdx.x;
    const LongParams P_l = a.P; const LongIn I_l = a.I; const LongArena AR_l = a.AR; const LongCoop coop_l = a.coop;      // no pointers into the kernel-argument struct
    LongWs W;
    long_ws_carve(&W, a.scratch + (unsigned long long)blockIdx.x * a.scratch_per_wave, a.sz);
    uint32_t n_list = *a.n_list, t_first = 0;
    // the giants' waves share their SIMDs with the main grid's: a dozen reads that each keep one wave busy for half a second and more are
    // the kernel pair's critical path, so they issue first (s_setprio)
    if (a.part == 2) __builtin_amdgcn_s_setprio(3);
    if (a.started && lane == 0) atomicAdd(a.started, 1u);
    if (a.part) {
        uint32_t n_giant = 0;
        for (int b2 = a.bin_cut; b2 < 32; ++b2) n_giant += a.hist[b2];
        if (a.part == 1) t_first = n_giant; else n_list = n_giant;
    }
    uint32_t n_rechain = 0, n_open = 0;
    LongClk clk{};
    for (;;) {
        uint32_t t = 0;
        if (lane == 0) t = atomicAdd(a.ticket, 1u);
        t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t) + t_first;
        if (t >= n_list) break;
        const uint32_t r = a.list[t];
        LongCtx C;
        C.P = &P_l; C.AP = nullptr; C.I = &I_l; C.W = &W; C.Ls = nullptr; C.A = AlignScratch{};
        C.qlen = (int32_t)(I_l.in.offsets[r + 1] - I_l.in.offsets[r]); C.read = r;
        C.sc_mch = C.sc_mis = C.sc_amb = C.sc_N = 0; C.need_big = false; C.err = 0; C.clk = a.clk ? &clk : nullptr;
        C.coop = (EXACT && a.coop_on) ? &coop_l : nullptr; C.coop_me = blockIdx.x;
        LongOut o;
        const unsigned long long t_r0 = a.clk ? wall_clock64() : 0ull;
        const unsigned long long d0_before = clk.d[0];
        if (a.clk) { clk.w_max = 0; clk.n_q = 0; clk.n_seg = 0; clk.tie_seg_a = 0; }
        const int32_t rc = lr_chains_wave<NR, EXACT, FAT>(C, RL, AR_l, o, a.drop ? a.drop[r] : 0u, TL);
        if (a.clk && lane == 0) {
            const unsigned long long dt = wall_clock64() - t_r0;
            // SCRUBBY_HIP_DBG_EXACT: one line per read of the exact passes (what profiles/r05_exact_reads.txt was made with)
            if (EXACT && (a.clk & 2)) printf("[exact] read %u qlen %d chains %d anchors %llu window %llu queries %llu segs %llu tie_seg_anchors %llu ms %.1f rc %d\n", r, C.qlen, o.n_chain, clk.d[0] - d0_before, clk.w_max, clk.n_q, clk.n_seg, clk.tie_seg_a, dt / 1e5, rc);
            atomicMax(&a.ctr->lext_slow, dt << 24 | (unsigned long long)(o.n_chain > 0xffffff ? 0xffffff : o.n_chain)); atomicAdd(&a.ctr->lext_kernel_sum, dt);
            atomicMax(&a.ctr->lext_slow2, dt << 32 | (unsigned long long)(uint32_t)C.qlen);
            atomicMax(&a.ctr->lext_slow3, dt << 32 | (unsigned long long)r);
            atomicMax(&a.ctr->lext_slow_part[a.part & 3], dt << 32 | (unsigned long long)(uint32_t)o.n_chain);
            atomicAdd(&a.ctr->lext_sum_part[a.part & 3], dt);
        }
        if (rc == 4) { if (lane == 0) atomicExch(&a.ctr->ext_overflow, 1u); }      // arena full: the host cuts the chunk in two
        else if (rc == 5) { if (lane == 0) lext_redo(a, r, C.err); }
        else if (rc == 7) {      // beyond the large ring and too large for the one-lane trees: chain-level answer, counted
            if (lane == 0) {
                a.flags[r] = 1;
                lk_mark(a, r, LK_UNRESOLVED);
                if (a.trace) ((int32_t *)(a.trace + r))[7] = 1;
                LongHdr h{0ull, -1, 0, 0, 0}; AR_l.hdr[r] = h;
                atomicAdd(&a.ctr->lext_unresolved, 1u); atomicExch(&a.ctr->lext_err_read, r); atomicExch(&a.ctr->lext_err_code, 16u + C.err);
            }
        }
        else if (rc == 6) {      // tied priorities / beyond the ring: the EXACT instance of this kernel takes the read
            if (lane == 0) {
                // (LongParams::e2_join_min: off by default)
                if ((NR >= 4096 || o.n_join > P_l.e2_join_min) && a.exact_list2) a.exact_list2[atomicAdd(a.n_exact2, 1u)] = r;
                else a.exact_list[atomicAdd(a.n_exact, 1u)] = r;
                lk_mark(a, r, LK_EXACT);
                LongHdr h{0ull, -1, 0, 0, 0}; AR_l.hdr[r] = h;
                if (C.err == 50u) atomicAdd(&a.ctr->lext_rmq_tie, 1u);
            }
        }
        else if (rc != 0) { if (lane == 0) lext_defer(a, r, 16u + C.err, false); }
        else {
            n_rechain += (o.rechained & 2) != 0; n_open += o.rmq_tie != 0; if (o.rmq_tie && lane == 0) lk_mark(a, r, LK_RMQ_OPEN);
            // a first-pass launch of the EXACT instance (the giants): the read's join asked the tree - the same stratum and count as a read of the exact passes
            if (EXACT && a.part && o.rmq_asked && lane == 0) { lk_mark(a, r, LK_EXACT); atomicAdd(&a.ctr->lext_rmq_tie, 1u); atomicAdd(&a.ctr->lext_exact_direct, 1u); }
        }
        __syncthreads();
    }
    if constexpr (EXACT) {
        if (a.coop_on) {      // no read left for this wave: runs of the reads the others still join (lr_coop_fill), while there are any
            for (unsigned long long idle = 0;;) {
                if (lr_coop_take<NR, FAT>(P_l, coop_l, RL, TL)) { idle = 0; continue; }
                if (al_b0((int32_t)cc_u32(coop_l.active)) == 0 || ++idle > 20000000ull) break;      // (bounded: about a minute of idle looks)
                __builtin_amdgcn_s_sleep(64);
            }
        }
    }
    if (lane == 0) {
        if (n_rechain) atomicAdd(&a.ctr->lext_rechained, n_rechain);
        if (n_open) { atomicAdd(&a.ctr->lext_rmq_tie, n_open); atomicAdd(&a.ctr->lext_rmq_open, n_open); }
        if (a.clk) { for (int i = 0; i < LR_NCLK; ++i) atomicAdd(a.part == 2 ? &a.ctr->lext_clk_big[i] : &a.ctr->lext_clk[i], clk.t[i]); unsigned long long *dd = a.part == 2 ? a.ctr->lext_d_big : a.ctr->lext_d; for (int i = 0; i < 7; ++i) atomicAdd(&dd[i], clk.d[i]); atomicMax(&dd[7], clk.d[7]); }
    }
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_regs_align_long(ExtLongArgs a)
{
    __shared__ AlignLds Ls;
    const uint32_t lane = threadIdx.x;
    const LongParams P_l = a.P; const AlignParams AP_l = a.AP; const ChainParams CP_l = a.CP; const LongIn I_l = a.I; const LongArena AR_l = a.AR;
    LongWs W;
    long_ws_carve(&W, a.scratch + (unsigned long long)blockIdx.x * a.scratch_per_wave, a.sz);
    const uint32_t n_list = *a.n_list;
    uint32_t n_regions = 0, n_dropped = 0, n_probed = 0;
    LongClk clk{};
    for (;;) {
        uint32_t t = 0, r = 0;
        if (a.follow_done) {
            // The list is being filled by the launch of the first size, running beside this one: an entry is taken only when there is one
            // (compare-and-swap on the ticket, so none is lost when this block gives up), the block sleeps between looks and leaves when the
            // other launch has ended and the list is empty - or after ~10 s of looks, whatever is left going to the launch that follows.
            // A ticket is only drawn for an entry that is already THERE (the list is preset to ~0 and an entry is written once, right after
            // its index was drawn by the producer): nothing is consumed and then given up on.
            // (every branch on a value all lanes hold: the lanes load the same words, one lane swaps - see lr_coop_take)
            uint32_t got = ~0u, rr = ~0u;
            auto ld = [](const uint32_t *p) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)__atomic_load_n(p, __ATOMIC_RELAXED)); };
            for (uint32_t looks = 0; looks < 2000000u; ++looks) {
                const uint32_t cur = ld(a.ticket), n = ld(a.n_list);
                if (cur < n) {
                    const uint32_t e = ld(&a.list[cur]);
                    if (e != ~0u) {
                        uint32_t old = ~0u;
                        if (lane == 0) old = atomicCAS(a.ticket, cur, cur + 1u);
                        if ((uint32_t)__builtin_amdgcn_readfirstlane((int)old) == cur) { got = cur; rr = e; break; }
                        continue;
                    }
                    __builtin_amdgcn_s_sleep(8);
                    continue;
                }
                if (ld(a.follow_done) && ld(a.n_list) <= ld(a.ticket)) break;
                __builtin_amdgcn_s_sleep(127);
            }
            __threadfence();
            t = (uint32_t)__builtin_amdgcn_readfirstlane((int)got); r = (uint32_t)__builtin_amdgcn_readfirstlane((int)rr);
            if (t == ~0u || r == ~0u) break;
        } else {
            if (lane == 0) t = atomicAdd(a.ticket, 1u);
            t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
            if (t >= n_list) break;
            r = a.list[t];
        }
        if (AR_l.hdr[r].n_u < 0) { __syncthreads(); continue; }      // given up by the chains kernel
        LongCtx C;
        C.P = &P_l; C.AP = &AP_l; C.I = &I_l; C.W = &W; C.Ls = &Ls;
        C.A = AlignScratch{};
        C.A.kmem = W.kmem; C.A.kH = W.kH; C.A.koff = W.koff; C.A.kp = W.kp; C.A.tcap = W.cap_k; C.A.qcap = W.cap_k; C.A.pcap = W.cap_p;
        C.qlen = (int32_t)(I_l.in.offsets[r + 1] - I_l.in.offsets[r]); C.read = r;
        C.sc_mch = (int8_t)(P_l.a < 0 ? -P_l.a : P_l.a); C.sc_mis = (int8_t)(P_l.b > 0 ? -P_l.b : P_l.b);
        C.sc_amb = (int8_t)(P_l.sc_ambi > 0 ? -P_l.sc_ambi : P_l.sc_ambi); C.sc_N = C.sc_amb == 0 ? (int8_t)(-P_l.e2) : C.sc_amb;
        C.need_big = false; C.err = 0; C.clk = a.clk ? &clk : nullptr; C.probe_why = 0;
        LongOut o;
        const unsigned long long t_r0 = a.clk ? wall_clock64() : 0ull;
        LongClk clk0 = clk;
        const int32_t rc = lr_regs_wave(C, CP_l, AR_l, a.flag_only != 0, a.probe != 0, o);
        if (a.clk && lane == 0 && !a.big_list) { const unsigned long long dt = wall_clock64() - t_r0; atomicMax(&a.ctr->lext_slow_part[3], dt << 32 | (unsigned long long)(uint32_t)C.qlen); atomicAdd(&a.ctr->lext_sum_part[3], dt); for (int i = 0; i < LR_NCLK; ++i) atomicMax(&a.ctr->lext_phase_max[i], clk.t[i] - clk0.t[i]); }
        if (rc == 5) {
            if (lane == 0) {      // counted again when the read comes back with all its anchors
                const int32_t rch = AR_l.hdr[r].rechained;
                if (rch & 2) atomicSub(&a.ctr->lext_rechained, 1u);
                if (rch & 4) atomicSub(&a.ctr->lext_rmq_tie, 1u);
                lext_redo(a, r, C.err);
            }
        }
        else if (rc != 0) { if (lane == 0) lext_defer(a, r, rc == 1 ? 32u : 16u + C.err, true); }
        else if (lane == 0) {
            a.flags[r] = o.n_regs > 0 ? 1 : 0;
            if (a.trace) {
                int32_t *tr = (int32_t *)(a.trace + r);
                tr[4] |= o.rechained | (o.rmq_tie ? 4 : 0); tr[5] = o.n_chain; tr[6] = o.best; tr[7] = o.n_regs > 0;
                ((int4 *)tr)[2] = make_int4(o.n_aligned, o.n_regs, o.dp_max, (int32_t)o.sig);
            }
        }
        if (rc == 0 && lane == 0 && a.flag_only && a.probe && !o.probed) lk_mark(a, r, LK_FULL);      // the probe did not decide: the complete procedure did
        if (a.clk && lane == 0 && !o.probed && C.probe_why) atomicAdd(&a.ctr->lr_probe_why[C.probe_why & 7], 1u);
        if (rc == 0) { n_regions += (uint32_t)o.n_aligned; n_dropped += o.n_regs == 0; n_probed += (uint32_t)o.probed; }      // every read of the list had a chain: also one the long join left without
        __syncthreads();
    }
    if (lane == 0) {
        if (n_regions) atomicAdd(&a.ctr->ext_regions, n_regions);
        if (n_dropped) atomicAdd(&a.ctr->ext_dropped, n_dropped);
        if (n_probed) atomicAdd(&a.ctr->sh_lemma[SHARD()], n_probed);
        if (a.clk) for (int i = 0; i < LR_NCLK; ++i) atomicAdd(a.big_list ? &a.ctr->lext_clk[i] : &a.ctr->lext_clk_big[i], clk.t[i]);
    }
}

// ------------------------------------------------------------------------------------------------
// host side: context
// ------------------------------------------------------------------------------------------------
struct sh_ctx {
    const sh_index *idx = nullptr;
    int idx_device = 0;
    sh_opts opts{};
    ChainParams P{};
    uint64_t max_reads = 0, max_bases = 0;
    uint32_t max_read_len = 0, seed_cap = 192, lds_words = 0;
    bool use_k1 = true;
    uint4 *d_records = nullptr;
    uint32_t *d_k1info = nullptr, *d_work_small = nullptr, *d_work_small2 = nullptr, *d_work_resketch = nullptr, *d_work_defer = nullptr, *d_work_defer2 = nullptr;
    uint32_t *d_big[2][2] = {};       // [pass][ping-pong] read lists of the repeat path
    Counters *d_ctr = nullptr;
    Counters *h_ctr = nullptr;        // pinned
    uint8_t *d_arena = nullptr;       // legacy path + BigBufs carved from the same allocation
    uint64_t arena_bytes = 0, legacy_bytes = 0;
    BigBufs B{};
    hipEvent_t ev[5] = {};
    bool use_long = false;           // reads longer than K1 takes: segment-parallel long-read front end
    uint8_t *d_long = nullptr; uint64_t long_bytes = 0, mz_cap = 0, max_segs = 0;
    uint32_t *d_seg_base = nullptr, *d_seg_cnt = nullptr, *d_mz_y = nullptr;
    unsigned long long *d_seg_off = nullptr, *d_seed_off = nullptr, *d_scan_tot = nullptr;
    uint64_t *d_mz_hash = nullptr; uint4 *d_lrec = nullptr;
    // extension stage (SH_F_CIGAR, short-read mode): chain hand-over buffers and the per-wave scratch of k_regs_align
    bool ext = false;
    AlignParams AP{};
    ChainSink sink{};
    uint8_t *d_ext = nullptr; uint64_t ext_bytes = 0;
    uint32_t *d_ext_list = nullptr, *d_ext_redo = nullptr; uint8_t *d_ext_scratch = nullptr; uint32_t *d_ext_unres[2] = {};
    unsigned long long cur_reads = 0;      // records of the chunk being classified
    unsigned long long ext_scratch_per_wave = 0, ext_scratch_per_wave_top = 0; uint32_t ext_waves_top = 0; uint32_t ext_waves = 0, ext_reg_cap = 0;
    hipEvent_t ev_ext[2] = {};
    // which reads of the LAST chunk took the rare paths (sh_ctx_debug_list: the bench's stratified oracle sample): 0 re-chained with max_occ,
    // 1 regs[0] aligned base by base, 2 the full fallback with every chain
    const uint32_t *dbg_ptr[3] = {}; uint32_t dbg_n[3] = {};
    // long-read presets: per read of the LAST CALL, the rarer paths of the extension stage it took (LK_* bits; sh_ctx_debug_list 3 .. 10)
    uint8_t *d_lkind = nullptr; uint64_t lkind_cap = 0, lkind_n = 0, lkind_r0 = 0;
    uint8_t *d_coop = nullptr;      // queue, counters and descriptors of lr_coop_fill (long-read presets, allocated on first use)
    // the same stage for the long-read presets (sh_long.h): per-wave working memory in two sizes
    bool ext_long = false;
    LongParams LP{};
    uint8_t *d_lext[4] = {}; unsigned long long lext_per_wave[4] = {}; uint32_t lext_waves[4] = {}; LongSizes lext_sz[4] = {}; int n_cu = 256;      // [phase * 2 + tier]
    uint32_t *d_lext_big = nullptr, *d_lext_big2 = nullptr, *d_lext_sorted = nullptr, *d_lext_unres[2] = {}, *d_lext_exact_list = nullptr, *d_lext_exact_list2 = nullptr, *d_lext_esorted = nullptr;
    uint8_t *d_lext_exact[2] = {}; unsigned long long lext_exact_per_wave[2] = {}; uint32_t lext_exact_waves[2] = {}; LongSizes lext_exact_sz[2] = {};      // the chains kernel with the long join on the literal trees
    uint8_t *d_larena = nullptr; unsigned long long larena_bytes = 0; LongHdr *d_lhdr = nullptr;
    // flag-only calls: anchors pre-selected by locus (k_lr_locus) - its read lists, what it left out per read, the reads to redo in full
    SortItem *d_locus[3] = {}; uint32_t *d_lr_drop = nullptr, *d_lr_fb = nullptr; int locus_shift = 0;
    uint64_t *d_stage_x = nullptr; uint32_t *d_stage_q = nullptr; uint64_t stage_cap = 0;      // raw anchors of the reads k_lr_locus thins out
    hipStream_t sx[4] = {};          // side streams: K2 and the sort classes run beside the main stream
    int side_pick[2] = {0, 1};       // long reads: the side streams of the giants' launch and of the follower (pick_side_streams)
    hipStream_t side_probed = nullptr; bool side_probed_done = false;
    int par = 1;                     // bit 0: K2 on a side stream (SCRUBBY_HIP_STREAMS=0: on the main stream)
    hipEvent_t evx[8] = {};
};

static void fill_chain_params(const sh_opts &o, int32_t mid_occ, ChainParams &P)
{
    P.k = o.k; P.is_sr = o.is_sr;
    P.mid_occ = mid_occ; P.max_occ = o.max_occ; P.max_max_occ = o.max_max_occ; P.occ_dist = o.occ_dist;
    P.min_cnt = o.min_cnt; P.min_sc = o.min_chain_score;
    P.max_gap = o.max_gap; P.max_gap_ref = o.max_gap_ref; P.max_frag_len = o.max_frag_len; P.bw = o.bw;
    P.max_skip = o.max_chain_skip; P.max_iter = o.max_chain_iter;
    P.pen_gap = (float)(o.chain_gap_scale * 0.01 * o.k);
    P.pen_skip = (float)(o.chain_skip_scale * 0.01 * o.k);
    P.q_occ_frac = o.q_occ_frac;
    const bool early_ok = o.k > 0 && (o.min_chain_score + o.k - 1) / o.k >= o.min_cnt && o.bw >= o.min_chain_score && o.bw / o.k + 1 >= o.min_cnt;
    P.flag_stop = early_ok && !getenv("SCRUBBY_HIP_NO_FLAG_STOP") ? o.min_chain_score : INT32_MAX;
    // pair test (ChainParams::pair_dq_*): needs flag_stop, two anchors enough (min_cnt, 2k >= min_sc), no skip penalty
    const bool pair_ok = P.flag_stop != INT32_MAX && o.chain_skip_scale == 0.0f && o.min_cnt <= 2 && 2 * o.k >= o.min_chain_score && !getenv("SCRUBBY_HIP_NO_PAIR");
    const int32_t dmin = std::max(1, o.min_chain_score - o.k);
    const int32_t dmax = std::min(std::min(std::min(24, o.max_chain_skip - 1), std::min(o.max_chain_iter - 1, o.bw)), o.max_gap);
    P.pair_dq_min = pair_ok && dmin <= dmax ? dmin : 0;
    P.pair_dq_max = pair_ok && dmin <= dmax ? dmax : 0;
    P.pair_min_anchors = getenv("SCRUBBY_HIP_PAIR_MIN") ? atoi(getenv("SCRUBBY_HIP_PAIR_MIN")) : 32;
    // SH_F_CIGAR: the decision is taken by the extension stage from ALL chains of a read, so the DP shortcuts above are off; the one
    // shortcut left is k_pair_pass mode 2 (ChainParams::ext_*)
    P.ext_s1 = 0; P.ext_unc_max = 0; P.ext_lemma = 0; P.ext_a = P.ext_b = P.ext_amb = P.ext_zdrop = 0;
    if ((o.flags & SH_F_CIGAR) && o.is_sr) {
        P.flag_stop = INT32_MAX; P.pair_dq_min = P.pair_dq_max = 0;
        const bool s1 = o.a > 0 && o.b > 0 && o.a * o.k >= o.min_dp_max && 2 * o.k >= o.min_chain_score && o.max_clip_ratio >= 1.0f &&
                        o.chain_skip_scale == 0.0f && o.min_cnt <= 2 && o.zdrop >= 0 && o.bw >= 0 && !getenv("SCRUBBY_HIP_NO_S1");
        P.ext_s1 = s1 ? 1 : 0;
        const bool lem = o.a > 0 && o.b > 0 && o.a * o.k >= o.min_dp_max && 2 * o.k >= o.min_chain_score && o.max_clip_ratio >= 1.0f && o.zdrop >= 0 && !getenv("SCRUBBY_HIP_NO_LEMMA");
        P.ext_lemma = lem ? 1 : 0;
        P.ext_unc_max = (s1 || lem) ? o.zdrop / o.b : 0;
        P.ext_a = o.a; P.ext_b = -o.b; P.ext_amb = o.sc_ambi > 0 ? -o.sc_ambi : o.sc_ambi; P.ext_zdrop = o.zdrop;
    }
    // long-read presets with SH_F_CIGAR (sh_long.h): every chain of a read is handed over, the chain-level shortcuts are off
    if ((o.flags & SH_F_CIGAR) && !o.is_sr) { P.flag_stop = INT32_MAX; P.pair_dq_min = P.pair_dq_max = 0; }
}

static void fill_long_params(const sh_opts &o, int32_t mid_occ, LongParams &L)
{
    L.k = o.k; L.min_cnt = o.min_cnt; L.min_sc = o.min_chain_score; L.max_gap = o.max_gap; L.bw = o.bw; L.bw_long = o.bw_long < o.bw ? o.bw : o.bw_long; L.min_ksw_len = o.min_ksw_len;
    L.a = o.a; L.b = o.b; L.q = o.q; L.e = o.e; L.q2 = o.q2; L.e2 = o.e2; L.sc_ambi = o.sc_ambi;
    L.zdrop = o.zdrop; L.zdrop_inv = o.zdrop_inv; L.end_bonus = o.end_bonus; L.min_dp_max = o.min_dp_max; L.best_n = o.best_n;
    L.pri_ratio = o.pri_ratio; L.mask_level = o.mask_level; L.max_clip_ratio = o.max_clip_ratio;
    L.max_skip = o.max_chain_skip; L.rmq_inner_dist = o.rmq_inner_dist; L.rmq_size_cap = o.rmq_size_cap; L.rmq_rescue_size = o.rmq_rescue_size; L.rmq_rescue_ratio = o.rmq_rescue_ratio;
    L.pen_gap = (float)(o.chain_gap_scale * 0.01 * o.k); L.pen_skip = (float)(o.chain_skip_scale * 0.01 * o.k);
    L.mid_occ = mid_occ; L.max_max_occ = o.max_max_occ; L.occ_dist = o.occ_dist;
    // ties of the long join that matter: the literal tree for reads of up to this many chain anchors (4096 by default; a satellite read of 10^5 anchors would keep
    // one lane chasing pointers for seconds - DESIGN.md 3.2); SCRUBBY_HIP_RMQ_EXACT_MAX=-1 takes every such read to the tree, 0 none
    L.rmq_exact_max = -1;      // every read that meets a tie that matters, or outgrows the rings, takes the literal tree (round 5: the tree lives in LDS)
    if (const char *env = getenv("SCRUBBY_HIP_RMQ_EXACT_MAX")) L.rmq_exact_max = atoi(env);
    // A read whose inner RMQ window (1000 reference bases) holds more than the 4096-anchor ring (5-bp satellite lattices: 23 of the bench's 2 M
    // reads) can only be chained by the literal one-lane trees over node pools in HBM - 10 to 25 s of one wave per read, measured.  Off by
    // default: such reads are counted (sh_stats.n_ext_unresolved) and keep their chain-level answer; SCRUBBY_HIP_RMQ_ONE_LANE=1 chains them.
    L.e2_join_min = INT32_MAX;      // (was 60 000 while the tree was kept for the whole read; with the join shared among waves the 1024-anchor ring's pass is the faster place: 8.35 -> 8.24 s per 2 M reads)
    if (const char *env = getenv("SCRUBBY_HIP_E2_JOIN_MIN")) L.e2_join_min = atoi(env);
    L.coop_min = LR_COOP_MIN; L.coop_run = LR_COOP_RUN; L.coop_check = getenv("SCRUBBY_HIP_COOP_CHECK") ? 1 : 0;
    if (const char *env = getenv("SCRUBBY_HIP_COOP_MIN")) L.coop_min = std::max(1, atoi(env));
    if (const char *env = getenv("SCRUBBY_HIP_COOP_RUN")) L.coop_run = std::max(2, atoi(env));
    L.rmq_one_lane = 0;
    if (const char *env = getenv("SCRUBBY_HIP_RMQ_ONE_LANE")) L.rmq_one_lane = atoi(env) != 0;
}

static void fill_align_params(const sh_opts &o, AlignParams &A)
{
    A.k = o.k; A.min_cnt = o.min_cnt; A.min_sc = o.min_chain_score; A.max_gap = o.max_gap; A.bw = o.bw; A.bw_long = o.bw_long;
    A.a = o.a; A.b = o.b; A.q = o.q; A.e = o.e; A.q2 = o.q2; A.e2 = o.e2; A.sc_ambi = o.sc_ambi;
    A.zdrop = o.zdrop; A.zdrop_inv = o.zdrop_inv; A.end_bonus = o.end_bonus; A.min_dp_max = o.min_dp_max; A.best_n = o.best_n;
    A.pri_ratio = o.pri_ratio; A.mask_level = o.mask_level; A.max_clip_ratio = o.max_clip_ratio;
    A.lemma = 0; A.unc_max = 0;
}

static bool w_supported(int w) { return w == 5 || w == 10 || w == 11 || w == 19; }

// A context built for one index serves another of the same shape (same device, k, w, occurrence thresholds resolved the same way):
// only the index pointer and the occurrence cut-off taken from it change.  The streaming host path keeps its context between runs
// this way (sh_stream.cpp): the reference rebuilds the index on every run, but re-creating ~35 GB of scratch each time means waiting
// for the driver to wipe the memory the previous run has just freed (seconds, not the ~60 ms of a first allocation).
sh_status shi_ctx_rebind(sh_ctx *c, const sh_index *idx)
{
    SH_CHECK(c && idx, SH_ERR_BAD_ARG, "shi_ctx_rebind: null argument");
    SH_CHECK(idx->device == c->idx_device && idx->k == c->opts.k && idx->w == c->opts.w, SH_ERR_BAD_ARG, "shi_ctx_rebind: index of another shape");
    SH_CHECK(!c->ext || (idx->d_ref && idx->d_cstart), SH_ERR_INDEX, "shi_ctx_rebind: this index holds no reference bases");
    SH_CHECK(c->opts.mid_occ > 0 || (idx->o_mid_occ <= 0 && idx->o_min_mid_occ == c->opts.min_mid_occ && idx->o_max_mid_occ == c->opts.max_mid_occ && idx->o_mid_occ_frac == c->opts.mid_occ_frac),
             SH_ERR_BAD_ARG, "shi_ctx_rebind: the index resolved its occurrence threshold under other parameters");
    c->idx = idx;
    const int32_t mid_occ = c->opts.mid_occ > 0 ? c->opts.mid_occ : idx->mid_occ;
    fill_chain_params(c->opts, mid_occ, c->P);
    c->AP.lemma = c->P.ext_lemma; c->AP.unc_max = c->P.ext_unc_max;
    fill_long_params(c->opts, mid_occ, c->LP);
    return SH_OK;
}
uint64_t shi_ctx_max_reads(const sh_ctx *c) { return c->max_reads; }
uint64_t shi_ctx_max_bases(const sh_ctx *c) { return c->max_bases; }
uint32_t shi_ctx_max_len(const sh_ctx *c) { return c->max_read_len; }

extern "C" sh_status sh_ctx_create(const sh_index *idx, const sh_opts *opts, uint64_t max_reads, uint64_t max_bases,
                                   uint32_t max_read_len, sh_ctx **out)
{
    SH_CHECK(idx && opts && out, SH_ERR_BAD_ARG, "sh_ctx_create: null argument");
    SH_CHECK(opts->k == idx->k && opts->w == idx->w, SH_ERR_BAD_ARG, "sh_ctx_create: opts (k=%d,w=%d) do not match index (k=%d,w=%d)", opts->k, opts->w, idx->k, idx->w);
    SH_CHECK(max_reads > 0 && max_reads < (1ULL << 31), SH_ERR_BAD_ARG, "sh_ctx_create: max_reads must be in [1, 2^31)");
    // an index (a cached one in particular) resolved mid_occ with the occurrence parameters of the preset it was built for
    SH_CHECK(opts->mid_occ > 0 || (idx->o_mid_occ <= 0 && idx->o_min_mid_occ == opts->min_mid_occ && idx->o_max_mid_occ == opts->max_mid_occ && idx->o_mid_occ_frac == opts->mid_occ_frac),
             SH_ERR_BAD_ARG, "sh_ctx_create: the index derived mid_occ with other occurrence parameters (min %d, max %d, frac %g) than this preset's (min %d, max %d, frac %g): rebuild it for this preset",
             idx->o_min_mid_occ, idx->o_max_mid_occ, (double)idx->o_mid_occ_frac, opts->min_mid_occ, opts->max_mid_occ, (double)opts->mid_occ_frac);
    SH_HIP(hipSetDevice(idx->device));
    sh_ctx *c = new sh_ctx();
    c->idx = idx; c->idx_device = idx->device; c->opts = *opts; c->max_reads = max_reads; c->max_bases = max_bases; c->max_read_len = max_read_len;
    int32_t mid_occ = opts->mid_occ > 0 ? opts->mid_occ : idx->mid_occ;
    fill_chain_params(*opts, mid_occ, c->P);
    fill_align_params(*opts, c->AP);
    c->AP.lemma = c->P.ext_lemma; c->AP.unc_max = c->P.ext_unc_max;
    fill_long_params(*opts, mid_occ, c->LP);
    c->ext = (opts->flags & SH_F_CIGAR) != 0;
    c->ext_long = c->ext && !opts->is_sr;
    if (c->ext) {
        SH_CHECK(idx->d_ref && idx->d_cstart, SH_ERR_INDEX, "sh_ctx_create: this index holds no reference bases, which the extension stage (SH_F_CIGAR) aligns against: rebuild it from FASTA");
        SH_CHECK(opts->a > 0 && opts->e > 0 && opts->q + opts->e < 100 && opts->q2 + opts->e2 < 100, SH_ERR_BAD_ARG, "sh_ctx_create: alignment scores out of the int8 range of ksw2");
    }
    // K1 stages a tile of 64 reads in LDS; it needs k <= 23 (hash and position share 64 bits),
    // read positions < 2^17 and a supported compile-time window
    c->use_k1 = opts->k <= 23 && max_read_len <= 1024 && w_supported(opts->w);
    uint64_t tile_bytes = (uint64_t)64 * max_read_len + 32;
    c->lds_words = (uint32_t)((tile_bytes + 15) / 16);
    const uint64_t n_tiles = (max_reads + 63) / 64;
    auto fail = [&](hipError_t e, const char *what) {
        size_t mf = 0, mt = 0;
        (void)hipMemGetInfo(&mf, &mt);
        sh_set_error("sh_ctx_create: %s: %s (device memory: %.1f of %.1f GB free)", what, hipGetErrorString(e), mf / 1e9, mt / 1e9);
        sh_ctx_destroy(c);
        return e == hipErrorOutOfMemory ? SH_ERR_OOM : SH_ERR_HIP;
    };
    hipError_t e;
    if ((e = hipMalloc(&c->d_records, (n_tiles * 64) * c->seed_cap * sizeof(uint4))) != hipSuccess) return fail(e, "records");
    if ((e = hipMalloc(&c->d_k1info, max_reads * 4)) != hipSuccess) return fail(e, "k1info");
    if ((e = hipMalloc(&c->d_work_small, max_reads * 4)) != hipSuccess) return fail(e, "work_small");
    if ((e = hipMalloc(&c->d_work_small2, max_reads * 4)) != hipSuccess) return fail(e, "work_small2");
    if ((e = hipMalloc(&c->d_work_resketch, max_reads * 4)) != hipSuccess) return fail(e, "work_resketch");
    if ((e = hipMalloc(&c->d_work_defer, max_reads * 4)) != hipSuccess) return fail(e, "work_defer");
    if ((e = hipMalloc(&c->d_work_defer2, max_reads * 4)) != hipSuccess) return fail(e, "work_defer2");
    for (int p = 0; p < 2; ++p) for (int q = 0; q < 2; ++q)
        if ((e = hipMalloc(&c->d_big[p][q], max_reads * 4)) != hipSuccess) return fail(e, "big lists");
    if ((e = hipMalloc(&c->d_ctr, sizeof(Counters))) != hipSuccess) return fail(e, "counters");
    if ((e = hipHostMalloc(&c->h_ctr, sizeof(Counters))) != hipSuccess) return fail(e, "pinned counters");
    // arena: anchors (+ sort buffer) and DP state of reads with > 64 anchors (36 B per anchor slot) + a slice for
    // the legacy re-sketch path.  Default 4 KiB per read of the batch (~110 anchor slots per read; the CHM13-sized
    // workload averages 72), at least 4 GiB; reads that find no room are deferred and re-run.
    c->arena_bytes = std::max<uint64_t>(4ull << 30, max_reads * 4096ull);      // floor: a small batch still meets reads with 10^5 anchors (4 MB each); a starved arena means deferral rounds of ~1.5 ms
    // without K1 every read takes the legacy path, whose sketch buffers and anchors live in the arena: ~48 B per base
    if (!c->use_k1) c->arena_bytes = std::max<uint64_t>(c->arena_bytes, std::min<uint64_t>(max_reads * (uint64_t)max_read_len * 48ull, (c->ext_long ? 32ull : 64ull) << 30));      // long-read presets with the extension filter: the raw anchors have their own buffer (d_stage_*)
    if (const char *env = getenv("SCRUBBY_HIP_ARENA_MB")) c->arena_bytes = (uint64_t)atoll(env) << 20;
    if ((e = hipMalloc(&c->d_arena, c->arena_bytes)) != hipSuccess) return fail(e, "arena");
    {
        const uint64_t big_min = (64ull << 20) + max_reads * 160;
        const bool long_fe = !c->use_k1 && w_supported(opts->w);      // the long-read front end feeds the repeat path
        c->legacy_bytes = (c->use_k1 || long_fe) ? std::min<uint64_t>(c->arena_bytes / 8, 1ull << 30)
                                                 : (c->arena_bytes > 2 * big_min ? c->arena_bytes - big_min : c->arena_bytes / 2);
        uint8_t *p = c->d_arena + c->legacy_bytes;
        uint64_t left = c->arena_bytes - c->legacy_bytes;
        // k_expand's waves reserve sort-list entries in chunks (<= 32): up to one abandoned chunk per wave and class
        const uint64_t waves = 3 * std::min<uint64_t>(std::max<uint64_t>(n_tiles, 1), 256 * 8);      // k_expand's grid (big_pass)
        const uint64_t sort_cap[N_SORT_CLS] = {max_reads + 32 * waves, max_reads + 32 * waves, max_reads + 8 * waves, max_reads + 8 * waves, max_reads + waves, max_reads + waves};
        uint64_t sort_cap_sum = 0;
        for (int i = 0; i < N_SORT_CLS; ++i) sort_cap_sum += sort_cap[i];
        uint64_t fixed = 4 * 64 * sizeof(SortItem) + 4096 + max_reads * (sizeof(BigMeta) + 8 + 8) + sort_cap_sum * sizeof(SortItem) + sort_cap[N_SORT_CLS - 1] * 4 + 16384;
        uint64_t per_anchor = 8 + 8 + 8 + 4 + 4 + 4 + 2 + (c->ext ? 8 + 3 : 0);   // ax bx az aq bq af (+ tile_split and cluster queue shares) (+ hz)
        uint64_t cap = left > fixed ? (left - fixed) / per_anchor : 0;
        cap &= ~15ull;
        if (cap < 1024) { sh_set_error("sh_ctx_create: arena of %llu MiB is too small", (unsigned long long)(c->arena_bytes >> 20)); sh_ctx_destroy(c); return SH_ERR_OOM; }
        BigBufs &B = c->B;
        B.anchor_cap = cap;
        auto take = [&](uint64_t bytes) { uint8_t *q = p; p += (bytes + 255) & ~255ull; return q; };
        B.ax = (uint64_t *)take(cap * 8); B.bx = (uint64_t *)take(cap * 8); B.az = (uint64_t *)take(cap * 8);
        B.aq = (uint32_t *)take(cap * 4); B.bq = (uint32_t *)take(cap * 4);
        B.af = (int32_t *)take(cap * 4);
        B.hz = c->ext ? (uint64_t *)take(cap * 8) : nullptr;
        B.meta = (BigMeta *)take(max_reads * sizeof(BigMeta));
        B.acc_nu = (int32_t *)take(max_reads * 4); B.acc_best = (int32_t *)take(max_reads * 4);
        for (int i = 0; i < N_SORT_CLS; ++i) B.sort_items[i] = (SortItem *)take(sort_cap[i] * sizeof(SortItem));
        B.tile_base = (uint32_t *)take((max_reads + 1) * 4);
        B.giant_order = (uint32_t *)take(sort_cap[N_SORT_CLS - 1] * 4);
        B.tile_split = (uint32_t *)take((cap / GT + max_reads + 2) * 4);
        {   // a cluster of class c has more than {4096, 1024, 256, 64} anchors
            const uint64_t div[4] = {4096, 1024, 256, c->ext ? 8u : 64u};
            for (int i = 0; i < 4; ++i) { B.cl_cap[i] = (uint32_t)std::min<uint64_t>(cap / div[i] + 64, UINT32_MAX); B.cl_items[i] = (SortItem *)take((uint64_t)B.cl_cap[i] * sizeof(SortItem)); }
        }
        {
            BigTables h{};
            for (int i = 0; i < N_SORT_CLS; ++i) h.sort_items[i] = B.sort_items[i];
            for (int i = 0; i < 4; ++i) { h.cl_items[i] = B.cl_items[i]; h.cl_cap[i] = B.cl_cap[i]; }
            BigTables *d_t = (BigTables *)take(sizeof(BigTables));
            if (hipMemcpy(d_t, &h, sizeof(h), hipMemcpyHostToDevice) != hipSuccess) { sh_set_error("sh_ctx_create: class tables"); sh_ctx_destroy(c); return SH_ERR_HIP; }
            B.tabs = d_t;
        }
        if ((uint64_t)(p - c->d_arena) > c->arena_bytes) {   // alignment slack: shrink
            sh_set_error("sh_ctx_create: internal arena carve overflow"); sh_ctx_destroy(c); return SH_ERR_OOM;
        }
    }
    c->use_long = !c->use_k1 && w_supported(opts->w);
    if (c->use_long) {
        const uint64_t cb = std::min<uint64_t>(max_bases + 64, max_reads * (uint64_t)max_read_len + 64);     // bases of one chunk, at most
        c->max_segs = cb / LSEG + max_reads + 2;
        c->mz_cap = cb * 5 / (2 * ((uint64_t)opts->w + 1)) + 4096;   // minimizer density is ~2/(w+1), + 25 %; reads beyond the cap take the legacy path
        auto al = [](uint64_t b) { return (b + 255) & ~255ull; };
        c->long_bytes = al((max_reads + 1) * 4) + al(c->max_segs * 4) + al((c->max_segs + 1) * 8) + al((c->max_segs / 16384 + 2) * 8) + al(c->mz_cap * 8) + al(c->mz_cap * 4) + al(c->mz_cap * 16) + al(max_reads * 8);
        if ((e = hipMalloc(&c->d_long, c->long_bytes)) != hipSuccess) return fail(e, "long-read buffers");
        uint8_t *p = c->d_long;
        auto take = [&](uint64_t b) { uint8_t *q = p; p += al(b); return q; };
        c->d_seg_base = (uint32_t *)take((max_reads + 1) * 4); c->d_seg_cnt = (uint32_t *)take(c->max_segs * 4);
        c->d_seg_off = (unsigned long long *)take((c->max_segs + 1) * 8); c->d_scan_tot = (unsigned long long *)take((c->max_segs / 16384 + 2) * 8); c->d_mz_hash = (uint64_t *)take(c->mz_cap * 8);
        c->d_mz_y = (uint32_t *)take(c->mz_cap * 4); c->d_lrec = (uint4 *)take(c->mz_cap * 16); c->d_seed_off = (unsigned long long *)take(max_reads * 8);
    }
    if (c->ext) {
        // chain hand-over: 32-B records, 12 B per chain anchor, one list head per read; sized for ~2 chains and ~32 chain anchors per
        // read of the chunk (a chunk that needs more is cut in two and re-run: classify_chunk returns SH_SPLIT)
        uint64_t cap_recs = std::max<uint64_t>(1ull << 20, 2 * max_reads), cap_anch = std::max<uint64_t>(1ull << 24, 32 * max_reads);
        if (c->ext_long) {      // a long read's chains hold most of its minimizers (~2 / (w + 1) per base), secondary chains as many again
            // bases of one chunk: max_bases may describe a whole batch of many chunks; ~8 kb per read is what long-read sets average - a
            // chunk that holds more is cut in two when its chains do not fit (SH_SPLIT)
            const uint64_t cb = std::min<uint64_t>({max_bases + 64, max_reads * (uint64_t)max_read_len + 64, max_reads * 8192ull + (64ull << 20)});
            // flag-only calls hand over the chains of the loci that can hold regs[0] (k_lr_locus: ~100 chain anchors and a dozen chains per read
            // of the bench workload); a call with a trace hands over every chain (thousands of anchors, hundreds of chains per read) and is
            // cut into smaller chunks when they do not fit
            cap_anch = std::max<uint64_t>(cap_anch, cb / 8);
            cap_recs = std::max<uint64_t>(cap_recs, std::min<uint64_t>(cb / 64, 1ull << 28) + 8 * max_reads);
        }
        if (const char *env = getenv("SCRUBBY_HIP_EXT_MB")) { cap_anch = std::max<uint64_t>(1 << 16, ((uint64_t)atoll(env) << 20) / 16); cap_recs = std::max<uint64_t>(1 << 12, cap_anch / 16); }
        cap_recs = std::min<uint64_t>((cap_recs + SINK_SHARDS - 1) / SINK_SHARDS, 0xfffffff0ull / SINK_SHARDS);      // per shard
        cap_anch = (cap_anch + SINK_SHARDS - 1) / SINK_SHARDS;
        auto al = [](uint64_t b) { return (b + 255) & ~255ull; };
        c->ext_bytes = al(SINK_SHARDS * cap_recs * sizeof(ChainRec)) + al(SINK_SHARDS * cap_anch * 8) + al(SINK_SHARDS * cap_anch * 4) + 3 * al(max_reads * 4) + al(max_reads * 8) + al(max_reads * 4) + (c->ext_long ? 0 : 2 * al(max_reads * 4));
        if ((e = hipMalloc(&c->d_ext, c->ext_bytes)) != hipSuccess) return fail(e, "extension-stage buffers");
        uint8_t *p = c->d_ext;
        auto take = [&](uint64_t b) { uint8_t *q = p; p += al(b); return q; };
        c->sink.recs = (ChainRec *)take(SINK_SHARDS * cap_recs * sizeof(ChainRec)); c->sink.cap_recs = (uint32_t)cap_recs;
        c->sink.cx = (uint64_t *)take(SINK_SHARDS * cap_anch * 8); c->sink.cq = (uint32_t *)take(SINK_SHARDS * cap_anch * 4); c->sink.cap_anch = cap_anch;
        c->sink.head = (uint32_t *)take(max_reads * 4);
        c->sink.trec = c->use_k1 ? c->d_records : nullptr; c->sink.tinfo = c->d_k1info; c->sink.tseed_cap = c->seed_cap;
        c->d_ext_list = (uint32_t *)take(max_reads * 4);
        c->d_ext_redo = (uint32_t *)take(max_reads * 4);
        if (!c->ext_long) { c->d_ext_unres[0] = (uint32_t *)take(max_reads * 4); c->d_ext_unres[1] = (uint32_t *)take(max_reads * 4); }
        c->sink.best = (unsigned long long *)take(max_reads * 8);
        c->sink.tie = (uint32_t *)take(max_reads * 4);
        c->sink.n_recs = c->d_ctr->ext_n_recs; c->sink.n_anch = c->d_ctr->ext_n_anch; c->sink.overflow = &c->d_ctr->ext_overflow;
        c->ext_reg_cap = 16384;
        if (const char *env = getenv("SCRUBBY_HIP_EXT_REGCAP")) c->ext_reg_cap = (uint32_t)std::max(65, atoi(env));      // tests: chains of a read the full procedure takes
        if (!c->ext_long) {
            c->ext_scratch_per_wave = align_scratch_layout(max_read_len, c->ext_reg_cap, nullptr, nullptr, nullptr);
            const uint64_t budget = 4ull << 30;
            c->ext_waves = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>({(uint64_t)256 * std::max<uint64_t>(1, (160u << 10) / sizeof(AlignLds)), budget / c->ext_scratch_per_wave, (max_reads + 3) / 4}));
            // k_regs_align_top holds one chain and the few regions z-drops split off it in LDS: its share of the scratch needs no region arrays
            c->ext_scratch_per_wave_top = align_scratch_layout(max_read_len, 64, nullptr, nullptr, nullptr);
            c->ext_waves_top = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>({(uint64_t)256 * std::max<uint64_t>(1, (160u << 10) / (sizeof(AlignLdsTop) + 16)), budget / c->ext_scratch_per_wave_top, (max_reads + 3) / 4}));
            if ((e = hipMalloc(&c->d_ext_scratch, std::max<uint64_t>((uint64_t)c->ext_waves * c->ext_scratch_per_wave, (uint64_t)c->ext_waves_top * c->ext_scratch_per_wave_top))) != hipSuccess) return fail(e, "extension-stage scratch");
        } else {
            // per-wave working memory of the two kernels (k_long_chains, k_regs_align_long), each in two sizes: every wave slot with room for
            // the usual read, and a few waves with room for the largest alignment minimap2 attempts (max_sw_mat = 10^8 cells) / for reads
            // with millions of chain anchors
            const uint64_t L = std::max<uint32_t>(max_read_len, 64);
            // Sizes: every wave slot takes the usual read (16 Ki chain anchors - with the anchors pre-selected by locus a read brings ~1000 -,
            // 8 MB of direction bytes: an end extension of max_gap bases at the band of the long-read presets); a few hundred slots take a
            // read of 256 Ki chain anchors or an alignment of 32 MB; beyond that memory is allocated for the reads that need it (ext_round)
            LongSizes z{};
            z.cap_q = (uint32_t)((L + 31) & ~15ull);
            z.cap_k = (uint32_t)std::min<uint64_t>(32768, ((2 * L + 1024) + 15) & ~15ull);
            z.cap_t = (uint32_t)(2 * L + 65536);
            z.cap_a = 16384;
            z.cap_u = z.cap_r = 4096;
            z.cap_m = (uint32_t)std::min<uint64_t>(65536, L / 4 + 1024);
            // 2 MB of direction bytes in the first size (8 until the second size's launch ran beside the first: a read that needs more costs
            // nothing extra now, and the smaller slots are 1700 waves instead of 1100 within the same budget)
            z.cap_p = std::min<uint64_t>(2ull << 20, (uint64_t)(2 * z.cap_k) * (z.cap_k + 32));
            if (const char *env = getenv("SCRUBBY_HIP_LEXT_P_KB")) z.cap_p = (uint64_t)atoll(env) << 10;
            if (const char *env = getenv("SCRUBBY_HIP_LEXT_A")) { z.cap_a = (uint32_t)std::max(64, atoi(env)); z.cap_u = z.cap_r = std::max(16u, z.cap_a / 4); }      // tests
            LongSizes zb = z;
            zb.cap_p = std::min<uint64_t>(32ull << 20, (uint64_t)(2 * z.cap_k) * (z.cap_k + 32));
            zb.cap_a = 1u << 18; zb.cap_u = zb.cap_r = 1u << 16; zb.cap_m = 65536;
            if (const char *env = getenv("SCRUBBY_HIP_LEXT_BIG_P_KB")) zb.cap_p = (uint64_t)atoll(env) << 10;      // tests: alignments beyond the second size
            if (const char *env = getenv("SCRUBBY_HIP_LEXT_BIG_A")) { zb.cap_a = (uint32_t)std::max(1024, atoi(env)); zb.cap_u = zb.cap_r = std::max(64u, zb.cap_a / 4); }      // tests: reads beyond the second size
            uint64_t budget[4] = {22ull << 30, 8ull << 30, 26ull << 30, 10ull << 30};
            const uint64_t wave_max[4] = {256 * 12, 256, 256 * 8, 256};      // LDS: 13 KB per wave in the chains kernel (twelve to a CU), 19 KB in the regions kernel
            {   // no more than a third of what the device has left (several contexts, ranks sharing a device, smaller GPUs)
                size_t mf = 0, mt = 0;
                if (hipMemGetInfo(&mf, &mt) == hipSuccess && mf > 0) {
                    const double scale = std::min(1.0, (double)mf / 3.0 / (double)(budget[0] + budget[1] + budget[2] + budget[3]));
                    for (auto &b : budget) b = (uint64_t)((double)b * scale);
                }
            }
            for (int ph = 0; ph < 2; ++ph) for (int t = 0; t < 2; ++t) {
                LongSizes q = t ? zb : z;
                q.phase = ph;
                if (ph == 0 && !t && !getenv("SCRUBBY_HIP_LEXT_A")) { q.cap_a = 65536; q.cap_u = 16384; }      // the chains kernel: 64 Ki anchors in every slot (the second size has 256 slots only, and a 40-kb read's join takes a second)
                const int i = ph * 2 + t;
                c->lext_sz[i] = q;
                c->lext_per_wave[i] = long_ws_carve(nullptr, nullptr, q);
                if (i == 0) { int dev = 0, ncu = 0; if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && ncu > 0) c->n_cu = ncu; }
                c->lext_waves[i] = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>({wave_max[i], budget[i] / c->lext_per_wave[i], t ? std::max<uint64_t>(4, max_reads / 16) : max_reads}));
                if (getenv("SCRUBBY_HIP_DBG")) fprintf(stderr, "[dbg] long-read stage working memory %d: %u waves x %.2f MB\n", i, c->lext_waves[i], c->lext_per_wave[i] / 1048576.0);
                if ((e = hipMalloc(&c->d_lext[i], (uint64_t)c->lext_waves[i] * c->lext_per_wave[i])) != hipSuccess) return fail(e, "long-read extension-stage scratch");
            }
            if ((e = hipMalloc(&c->d_lext_big, max_reads * 4)) != hipSuccess) return fail(e, "long-read extension-stage list");
            if ((e = hipMalloc(&c->d_lext_big2, max_reads * 4)) != hipSuccess) return fail(e, "long-read extension-stage list");
            if ((e = hipMalloc(&c->d_lext_sorted, max_reads * 4)) != hipSuccess) return fail(e, "long-read extension-stage list");
            for (auto &q : c->d_lext_unres) if ((e = hipMalloc(&q, max_reads * 4)) != hipSuccess) return fail(e, "long-read extension-stage list");
            if ((e = hipMalloc(&c->d_lext_exact_list, max_reads * 4)) != hipSuccess) return fail(e, "long-read extension-stage list");
            if ((e = hipMalloc(&c->d_lext_exact_list2, max_reads * 4)) != hipSuccess) return fail(e, "long-read extension-stage list");
            if ((e = hipMalloc(&c->d_lext_esorted, max_reads * 4)) != hipSuccess) return fail(e, "long-read extension-stage list");
            for (int t = 1; t < 2; ++t) {   // the exact long join on the one-lane trees (E3): the second size of the chains kernel plus the node pools of the two trees; one wave per CU (LDS)
                LongSizes q = t ? zb : z; q.phase = 2;
                c->lext_exact_sz[t] = q;
                c->lext_exact_per_wave[t] = long_ws_carve(nullptr, nullptr, q);
                const uint64_t bud = (t ? 6ull << 30 : 1ull << 30) * (budget[0] + 1) / ((6ull << 30) + 1);      // scaled like the others
                c->lext_exact_waves[t] = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>({256, bud / c->lext_exact_per_wave[t], std::max<uint64_t>(4, max_reads / 16)}));
                if ((e = hipMalloc(&c->d_lext_exact[t], (uint64_t)c->lext_exact_waves[t] * c->lext_exact_per_wave[t])) != hipSuccess) return fail(e, "long-read extension-stage scratch");
            }
            // the chains between the two kernels: what the hand-over buffers can hold, twice (the long join re-orders, it adds nothing; a read
            // with one chain kept may leave that chain beside the join's outcome: lr_chains_wave, `both`)
            c->larena_bytes = SINK_SHARDS * cap_anch * 32 + SINK_SHARDS * cap_recs * 12 + max_reads * 48 + (1ull << 20);
            if ((e = hipMalloc(&c->d_larena, c->larena_bytes)) != hipSuccess) return fail(e, "long-read extension-stage arena");
            if ((e = hipMalloc(&c->d_lhdr, max_reads * sizeof(LongHdr))) != hipSuccess) return fail(e, "long-read extension-stage headers");
            for (auto &q : c->d_locus) if ((e = hipMalloc(&q, max_reads * sizeof(SortItem))) != hipSuccess) return fail(e, "long-read locus lists");
            if ((e = hipMalloc(&c->d_lr_drop, max_reads * 4)) != hipSuccess) return fail(e, "long-read locus table");
            if ((e = hipMalloc(&c->d_lr_fb, max_reads * 4)) != hipSuccess) return fail(e, "long-read locus list");
            {   // k_expand's raw anchors wait here for k_lr_locus (12 B each; ~1.1 anchors per base on a repeat-rich reference); reads that
                // find no room come back in the pass's next iteration
                const uint64_t cb = std::min<uint64_t>({max_bases + 64, max_reads * (uint64_t)max_read_len + 64, max_reads * 8192ull + (64ull << 20)});
                c->stage_cap = std::min<uint64_t>(cb + cb / 4 + (1ull << 20), (64ull << 30) / 12);
                {   // last of the context's allocations: at most 45 % of what the device still has (a million-read context fits that way - its
                    // launches pay the stage's longest reads once where two half-size ones pay them twice; what finds no room is deferred)
                    size_t fr = 0, tot = 0;
                    if (hipMemGetInfo(&fr, &tot) == hipSuccess) c->stage_cap = std::min<uint64_t>(c->stage_cap, std::max<uint64_t>(1ull << 20, (uint64_t)((double)fr * 0.45) / 12));
                }
                if (const char *env = getenv("SCRUBBY_HIP_STAGE_MB")) c->stage_cap = std::max<uint64_t>(1ull << 16, ((uint64_t)atoll(env) << 20) / 12);
                if ((e = hipMalloc(&c->d_stage_x, c->stage_cap * 8)) != hipSuccess) return fail(e, "long-read anchor staging");
                if ((e = hipMalloc(&c->d_stage_q, c->stage_cap * 4)) != hipSuccess) return fail(e, "long-read anchor staging");
            }
            {   // reference windows of 2^shift bases, at least as wide as the widest gap either chaining pass links across
                const int32_t D = std::max({opts->max_gap, opts->max_gap_ref, opts->bw, opts->bw_long, 1});
                int sh = 9;
                while (sh < 31 && (1ll << sh) < (long long)D) ++sh;
                c->locus_shift = sh;
            }
        }
        for (auto &ev : c->ev_ext) if ((e = hipEventCreate(&ev)) != hipSuccess) return fail(e, "event");
    }
    for (auto &ev : c->ev) if ((e = hipEventCreate(&ev)) != hipSuccess) return fail(e, "event");
    for (auto &ev : c->evx) if ((e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) return fail(e, "event");
    // (the side streams are created when a call first needs one: HIP deals its hardware queues - four by default - out to the streams that
    // exist, and a side stream that shares the caller's queue runs after it, not beside it)
    if (const char *env = getenv("SCRUBBY_HIP_STREAMS")) c->par = atoi(env);
    *out = c;
    return SH_OK;
}

extern "C" sh_status sh_ctx_debug_list(const sh_ctx *c, int32_t which, uint32_t *out, uint64_t cap, uint64_t *n_out)
{
    SH_CHECK(c && n_out && which >= 0 && which < 11, SH_ERR_BAD_ARG, "sh_ctx_debug_list: bad argument");
    if (which >= 3) {      // long reads: the reads of the last call with bit (which - 3) of their kind byte set
        *n_out = 0;
        if (!c->d_lkind || c->lkind_n == 0) return SH_OK;
        std::vector<uint8_t> h(c->lkind_n);
        SH_HIP(hipSetDevice(c->idx_device));
        SH_HIP(hipMemcpy(h.data(), c->d_lkind, c->lkind_n, hipMemcpyDeviceToHost));
        uint64_t n = 0;
        for (uint64_t r = 0; r < c->lkind_n; ++r) if (h[r] >> (which - 3) & 1) { if (out && n < cap) out[n] = (uint32_t)r; ++n; }
        *n_out = n;
        return SH_OK;
    }
    *n_out = c->dbg_ptr[which] ? c->dbg_n[which] : 0;
    if (!out || !c->dbg_ptr[which]) return SH_OK;
    const uint64_t n = std::min<uint64_t>(cap, c->dbg_n[which]);
    SH_HIP(hipSetDevice(c->idx_device));
    if (n) SH_HIP(hipMemcpy(out, c->dbg_ptr[which], n * 4, hipMemcpyDeviceToHost));
    return SH_OK;
}

extern "C" sh_status sh_ctx_destroy(sh_ctx *c)
{
    if (!c) return SH_OK;
    hipFree(c->d_records); hipFree(c->d_k1info); hipFree(c->d_work_small); hipFree(c->d_work_small2); hipFree(c->d_work_resketch);
    hipFree(c->d_work_defer); hipFree(c->d_work_defer2);
    for (auto &pp : c->d_big) for (auto p : pp) hipFree(p);
    hipFree(c->d_ctr); if (c->h_ctr) hipHostFree(c->h_ctr); hipFree(c->d_arena); hipFree(c->d_long);
    hipFree(c->d_ext); hipFree(c->d_ext_scratch); for (auto q : c->d_lext) hipFree(q); hipFree(c->d_lext_big); hipFree(c->d_lext_big2); hipFree(c->d_lext_sorted); for (auto q : c->d_lext_unres) hipFree(q); hipFree(c->d_lext_exact_list); hipFree(c->d_lext_exact_list2); hipFree(c->d_lext_esorted); for (auto q : c->d_lext_exact) hipFree(q); hipFree(c->d_larena); hipFree(c->d_lhdr); for (auto q : c->d_locus) hipFree(q); hipFree(c->d_lr_drop); hipFree(c->d_lr_fb); hipFree(c->d_stage_x); hipFree(c->d_stage_q); hipFree(c->d_lkind); hipFree(c->d_coop);
    for (auto ev : c->ev_ext) if (ev) hipEventDestroy(ev);
    for (auto ev : c->ev) if (ev) hipEventDestroy(ev);
    for (auto ev : c->evx) if (ev) hipEventDestroy(ev);
    for (auto st : c->sx) if (st) hipStreamDestroy(st);
    delete c;
    return SH_OK;
}

template <int W>
static void launch_k1(const K1Args &a, uint32_t n_tiles, size_t lds, hipStream_t s)
{
    hipLaunchKernelGGL(k_sketch_probe<W>, dim3(n_tiles), dim3(64), lds, s, a);
}

template <int W>
static void launch_long(const LongArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(k_long_segtable, dim3(1), dim3(64), 0, s, a);
    hipLaunchKernelGGL((k_long_sketch<W, false>), dim3(256 * 8), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_long_scan, dim3(512), dim3(1024), 0, s, a, 0);
    hipLaunchKernelGGL(k_long_scan, dim3(1), dim3(1024), 0, s, a, 1);
    hipLaunchKernelGGL(k_long_scan, dim3(512), dim3(1024), 0, s, a, 2);
    hipLaunchKernelGGL((k_long_sketch<W, true>), dim3(256 * 8), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_long_probe, dim3(256 * 16), dim3(64), 0, s, a);
}

static sh_status ensure_side_streams(sh_ctx *c, int first, int last)
{
    for (int i = first; i <= last; ++i)
        if (!c->sx[i]) SH_HIP(hipStreamCreateWithFlags(&c->sx[i], hipStreamNonBlocking));
    return SH_OK;
}

// one wave busy for `ticks` of the 100 MHz clock (bounded)
__global__ void k_spin(unsigned long long ticks)
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

// Which side streams run BESIDE the caller's stream?  HIP deals a few hardware queues out to the streams of a process, and two streams on
// one queue take turns.  Measured once per (context, caller's stream): a 200-us kernel on each side stream beside one on the caller's - both
// done after ~200 us (side by side) or ~400 (one after the other).  The giants' launch gets the first stream found to run beside, the
// follower the second; when none does the launches run in turn, which is merely slower.
static sh_status pick_side_streams(sh_ctx *c, hipStream_t s)
{
    if (c->side_probed_done && c->side_probed == s) return SH_OK;
    sh_status es = ensure_side_streams(c, 0, 2);
    if (es != SH_OK) return es;
    // SCRUBBY_HIP_SIDE_PICK=g,f pins the two streams (giants' launch, follower) and skips the probe
    if (const char *pin = getenv("SCRUBBY_HIP_SIDE_PICK")) {
        int g = 0, f = 1;
        if (sscanf(pin, "%d,%d", &g, &f) == 2 && g >= 0 && g < 3 && f >= 0 && f < 3) { c->side_pick[0] = g; c->side_pick[1] = f; c->side_probed = s; c->side_probed_done = true; return SH_OK; }
    }
    struct Events {      // destroyed on every way out
        hipEvent_t t0 = nullptr, t1 = nullptr, e = nullptr;
        ~Events() { if (t0) hipEventDestroy(t0); if (t1) hipEventDestroy(t1); if (e) hipEventDestroy(e); }
    } ev;
    SH_HIP(hipEventCreate(&ev.t0)); SH_HIP(hipEventCreate(&ev.t1)); SH_HIP(hipEventCreateWithFlags(&ev.e, hipEventDisableTiming));
    float ms[3] = {1e9f, 1e9f, 1e9f};
    for (int i = 0; i < 3; ++i) {
        float rep[3] = {0, 0, 0};      // the median of three tries: one 0.2-ms timing is at the mercy of whatever else the device is doing
        for (int k = 0; k < 3; ++k) {
            SH_HIP(hipStreamSynchronize(s));
            SH_HIP(hipEventRecord(ev.t0, s));
            SH_HIP(hipStreamWaitEvent(c->sx[i], ev.t0, 0));
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, 20000ull);
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, c->sx[i], 20000ull);
            SH_HIP(hipEventRecord(ev.e, c->sx[i]));
            SH_HIP(hipStreamWaitEvent(s, ev.e, 0));
            SH_HIP(hipEventRecord(ev.t1, s));
            SH_HIP(hipEventSynchronize(ev.t1));
            SH_HIP(hipEventElapsedTime(&rep[k], ev.t0, ev.t1));
        }
        std::sort(rep, rep + 3);
        ms[i] = rep[1];
    }
    int ord[3] = {0, 1, 2};
    std::sort(ord, ord + 3, [&](int a, int b) { return ms[a] < ms[b]; });
    c->side_pick[0] = ord[0]; c->side_pick[1] = ord[1];
    c->side_probed = s; c->side_probed_done = true;
    if (getenv("SCRUBBY_HIP_DBG")) fprintf(stderr, "[dbg] side streams beside the caller's: %.2f %.2f %.2f ms for two 0.2-ms kernels; giants on %d, follower on %d\n", ms[0], ms[1], ms[2], c->side_pick[0], c->side_pick[1]);
    return SH_OK;
}

// one pass of the repeat path over list[*count]: expand -> sort -> DP -> finalize
static sh_status big_pass(sh_ctx *c, K3Args k, uint32_t grid, hipStream_t s)
{
    // per-pass device counters: arena cursor, sort list, cluster lists
    Counters *ctr = c->d_ctr;
    SH_HIP(hipMemsetAsync(&ctr->anchor_cursor, 0, 8, s));
    SH_HIP(hipMemsetAsync(&ctr->n_sort[0], 0, 4 * N_SORT_CLS, s));
    SH_HIP(hipMemsetAsync(&ctr->sort_ticket[0], 0, 4 * N_SORT_CLS, s));
    SH_HIP(hipMemsetAsync(&ctr->expand_ticket, 0, 4, s));
    SH_HIP(hipMemsetAsync(&ctr->n_cl[0], 0, 4 * 6, s));
    // 6144 waves for 4096 resident (103 VGPRs: 4 per SIMD): measured best; 4096 or 5120 waves, or 5 waves per SIMD at 96 VGPRs, are 0-3 % slower
    if (k.locus) { SH_HIP(hipMemsetAsync(&ctr->n_locus[0], 0, 4 * 6, s)); SH_HIP(hipMemsetAsync(&ctr->stage_cursor, 0, 8, s)); }
    if (k.seed_off) hipLaunchKernelGGL(k_expand<true>, dim3(grid * 3), dim3(64), 0, s, k);
    else hipLaunchKernelGGL(k_expand<false>, dim3(grid * 3), dim3(64), 0, s, k);
    if (k.locus && k.seed_off) {      // long-read presets, flag-only: only the anchors that can hold regs[0] reach the sort classes
        hipLaunchKernelGGL(k_lr_locus<12>, dim3(256 * 8), dim3(256), 0, s, k, 0);
        hipLaunchKernelGGL(k_lr_locus<14>, dim3(256 * 5), dim3(256), 0, s, k, 1);
        hipLaunchKernelGGL(k_lr_locus<15>, dim3(256 * 2), dim3(256), 0, s, k, 2);
    }
    // the sort classes run one after the other.  Side by side (streams sx[0..2]) measured 9 % slower when they carried the whole repeat
    // path (they fight for LDS); that mode predates k_group_probe, which the class-4 and giant kernels must follow: it stays off.
    static const int side_env = getenv("SCRUBBY_HIP_SIDE") ? atoi(getenv("SCRUBBY_HIP_SIDE")) : -1;
    // small batches (an eighth of the bench's records on each of 8 GPUs): every class kernel is mostly tail, side by side they overlap:
    // 39.3 -> 36.6 ms at 2.5 M records; at 20 M it costs 2 % (SCRUBBY_HIP_SIDE=0 / 1 forces either)
    const bool gp = k.flag_only && k.P.flag_stop != INT32_MAX && !(k.dbg & 64);      // k_group_probe runs (chain-level decision): the class-4 and giant kernels must follow it
    const bool side = side_env > 0 || (side_env < 0 && !gp && c->cur_reads <= 3000000ull);
    if (side) { sh_status es = ensure_side_streams(c, 0, 2); if (es != SH_OK) return es; }
    hipStream_t s0 = side ? c->sx[0] : s, s1 = side ? c->sx[1] : s, g = side ? c->sx[2] : s;
    const bool use_pf = (k.emit || !k.flag_only) && !(k.dbg & 128);
    const bool use_top = use_pf && k.emit && k.sink.best != nullptr && !(k.dbg & 512);
    if (use_pf) SH_HIP(hipMemsetAsync(&ctr->top_ticket[0], 0, 4 * N_SORT_CLS, s));
    if (side) {
        SH_HIP(hipEventRecord(c->evx[0], s));
        for (int i = 0; i < 3; ++i) SH_HIP(hipStreamWaitEvent(c->sx[i], c->evx[0], 0));
    }
    if (gp) {      // reads with thousands of anchors: try one (strand, contig) group first
        hipLaunchKernelGGL(k_group_probe, dim3(256 * 3), dim3(256), 0, s, k, 4);
        hipLaunchKernelGGL(k_group_probe, dim3(256 * 3), dim3(256), 0, s, k, (int)SORT_CLS_GIANT);
    }
    // each class: the read-level pass (k_sort_top), then the cluster path for what it left (k_sort_lds), on the class's stream
    if (use_top) hipLaunchKernelGGL((k_sort_top<256, 0, 64>), dim3(grid * 2), dim3(64), 0, s, k);
    hipLaunchKernelGGL((k_sort_lds<256, 0, 64>), dim3(grid * 2), dim3(64), 0, s, k);
    if (use_top) hipLaunchKernelGGL((k_sort_top<SORT_LDS_A, 1, 128>), dim3(grid * 2), dim3(128), 0, s, k);
    hipLaunchKernelGGL((k_sort_lds<SORT_LDS_A, 1, 128>), dim3(grid * 2), dim3(128), 0, s, k);
    if (use_top) hipLaunchKernelGGL((k_sort_top<1024, 2, 256>), dim3(256 * 6), dim3(256), 0, s0, k);
    hipLaunchKernelGGL((k_sort_lds<1024, 2, 256>), dim3(256 * 6), dim3(256), 0, s0, k);
    if (use_top) hipLaunchKernelGGL((k_sort_top<SORT_LDS_B, 3, 256>), dim3(256 * 3), dim3(256), 0, s0, k);
    hipLaunchKernelGGL((k_sort_lds<SORT_LDS_B, 3, 256>), dim3(256 * 3), dim3(256), 0, s0, k);
    if (use_top) hipLaunchKernelGGL((k_sort_top<SORT_LDS_C, 4, 512>), dim3(256), dim3(512), 0, s1, k);
    hipLaunchKernelGGL((k_sort_lds<SORT_LDS_C, 4, 512>), dim3(256), dim3(512), 0, s1, k);
    hipLaunchKernelGGL(k_giant_scan, dim3(1), dim3(1024), 0, g, k);
    hipLaunchKernelGGL(k_giant_chunksort, dim3(256 * 3), dim3(256), 0, g, k);
    for (uint32_t round = 0; round < 8; ++round) {      // run widths GT << round: up to 2^19 anchors per read
        hipLaunchKernelGGL(k_giant_partition, dim3(256), dim3(256), 0, g, k, round);
        hipLaunchKernelGGL(k_giant_merge, dim3(256 * 6), dim3(256), 0, g, k, round);      // 24 KB of LDS per block: six per CU
    }
    // flag-only hand-over (t_mode): the big clusters first (k_cluster_dp), then the small ones against the best score those gave
    const bool two_phase = k.t_mode && k.sink.best != nullptr;
    if (use_pf) hipLaunchKernelGGL(k_giant_top, dim3(1024), dim3(512), 0, g, k);
    hipLaunchKernelGGL(k_giant_chain, dim3(1024), dim3(512), 0, g, k, two_phase ? 0 : -1);
    if (side && k.cl_lds) {      // the LDS classes feed k_cluster_dp's queue too - all of them: the 512 class runs on the main stream
        SH_HIP(hipEventRecord(c->evx[1], c->sx[0])); SH_HIP(hipStreamWaitEvent(g, c->evx[1], 0));
        SH_HIP(hipEventRecord(c->evx[2], c->sx[1])); SH_HIP(hipStreamWaitEvent(g, c->evx[2], 0));
        SH_HIP(hipEventRecord(c->evx[6], s)); SH_HIP(hipStreamWaitEvent(g, c->evx[6], 0));
    }
    if (!(k.dbg & 32)) hipLaunchKernelGGL(k_cluster_dp, dim3(256 * 7), dim3(256), 0, g, k);      // 72 VGPRs: 7 waves per SIMD (4: 74 ms, 6: 60, 8 with spills: 57)
    if (two_phase) {
        SH_HIP(hipMemsetAsync(&ctr->sort_ticket[SORT_CLS_GIANT], 0, 4, g));
        hipLaunchKernelGGL(k_giant_chain, dim3(1024), dim3(512), 0, g, k, 1);
    }
    if (side) for (int i = 0; i < 3; ++i) { SH_HIP(hipEventRecord(c->evx[1 + i], c->sx[i])); SH_HIP(hipStreamWaitEvent(s, c->evx[1 + i], 0)); }
    hipLaunchKernelGGL(k_finalize, dim3(grid), dim3(256), 0, s, k);
    return SH_OK;
}

static sh_status classify_chunk(sh_ctx *c, const uint8_t *d_bases, const uint64_t *d_offsets, uint64_t n_reads, uint64_t n_bases,
                                uint8_t *d_flags, sh_trace *d_trace, hipStream_t s, sh_stats *stats)
{
    const sh_index *idx = c->idx;
    c->cur_reads = n_reads;
    SH_HIP(hipMemsetAsync(c->d_ctr, 0, sizeof(Counters), s));
    if (c->ext) { SH_HIP(hipMemsetAsync(c->sink.head, 0xff, n_reads * 4, s)); SH_HIP(hipMemsetAsync(c->sink.best, 0, n_reads * 8, s)); SH_HIP(hipMemsetAsync(c->sink.tie, 0, n_reads * 4, s)); }
    SH_HIP(hipEventRecord(c->ev[0], s));
    const uint32_t n_tiles = (uint32_t)((n_reads + 63) / 64);
    if (c->use_k1) {
        K1Args a{};
        a.bases = d_bases; a.offsets = d_offsets; a.n_reads = n_reads; a.n_bases = n_bases;
        a.slots = (const uint4 *)idx->d_slots; a.lg_slots = idx->lg_slots; a.k = idx->k;
        a.records = c->d_records; a.seed_cap = c->seed_cap;
        a.k1info = c->d_k1info; a.flags = d_flags; a.trace = d_trace;
        a.work_small = c->d_work_small; a.work_resketch = c->d_work_resketch; a.work_big = c->d_big[0][0]; a.ctr = c->d_ctr;
        a.lds_words = c->lds_words; a.mid_occ = c->P.mid_occ;
        a.q_occ_max = (c->P.q_occ_frac > 0.0f && c->P.mid_occ > 0) ? (uint32_t)c->P.mid_occ : UINT32_MAX;
        size_t lds = (((size_t)K1_LIST_CAP * 64 * 8 + ((size_t)c->lds_words + 2) * 6 + 7) & ~(size_t)7) + 64 * 8;
        switch (idx->w) {
        case 5: launch_k1<5>(a, n_tiles, lds, s); break;
        case 10: launch_k1<10>(a, n_tiles, lds, s); break;
        case 11: launch_k1<11>(a, n_tiles, lds, s); break;
        case 19: launch_k1<19>(a, n_tiles, lds, s); break;
        default: sh_set_error("unsupported w"); return SH_ERR_BAD_ARG;
        }
    } else if (c->use_long) {
        LongArgs a{};
        a.bases = d_bases; a.offsets = d_offsets; a.n_reads = n_reads; a.n_bases = n_bases;
        a.slots = (const uint4 *)idx->d_slots; a.lg_slots = idx->lg_slots; a.k = idx->k;
        a.seg_base = c->d_seg_base; a.seg_cnt = c->d_seg_cnt; a.seg_off = c->d_seg_off; a.scan_tot = c->d_scan_tot;
        a.mz_hash = c->d_mz_hash; a.mz_y = c->d_mz_y; a.mz_cap = c->mz_cap; a.lrec = c->d_lrec; a.seed_off = c->d_seed_off;
        a.k1info = c->d_k1info; a.flags = d_flags; a.trace = d_trace;
        a.work_big = c->d_big[0][0]; a.work_resketch = c->d_work_resketch; a.ctr = c->d_ctr;
        a.mid_occ = c->P.mid_occ;
        a.q_occ_max = (c->P.q_occ_frac > 0.0f && c->P.mid_occ > 0) ? (uint32_t)c->P.mid_occ : UINT32_MAX;
        a.q_occ_frac = c->P.q_occ_frac;
        a.n_segs_out = &c->d_ctr->n_long_segs; a.max_segs = (uint32_t)std::min<uint64_t>(c->max_segs, UINT32_MAX);
        switch (idx->w) {
        case 5: launch_long<5>(a, s); break;
        case 10: launch_long<10>(a, s); break;
        case 11: launch_long<11>(a, s); break;
        case 19: launch_long<19>(a, s); break;
        default: sh_set_error("unsupported w"); return SH_ERR_BAD_ARG;
        }
    } else {
        hipLaunchKernelGGL(k_route_all, dim3((uint32_t)((n_reads + 255) / 256)), dim3(256), 0, s, n_reads, c->d_work_resketch, c->d_ctr);
    }
    SH_HIP(hipEventRecord(c->ev[1], s));

    // flag-only shortcuts over the seed records: the pair test (chain-level decision), or with SH_F_CIGAR mode 2 (co-diagonal singletons)
    const bool pair_pass = d_trace == nullptr && (c->P.pair_dq_max > 0 || c->P.ext_s1) && c->use_k1;
    const int pair_mode_small = c->ext ? 2 : 0, pair_mode_big = c->ext ? 2 : 1;
    K2Args b{};
    b.offsets = d_offsets; b.bases = d_bases; b.n_reads = n_reads;
    b.slots = (const uint4 *)idx->d_slots; b.lg_slots = idx->lg_slots; b.w = idx->w;
    b.positions = idx->d_positions;
    b.records = c->d_records; b.seed_cap = c->seed_cap;
    b.k1info = c->d_k1info; b.flags = d_flags; b.trace = d_trace;
    b.work_big = c->d_big[0][0];
    b.work_defer = c->d_work_defer; b.ctr = c->d_ctr;
    b.arena = c->d_arena; b.arena_bytes = c->legacy_bytes;
    b.P = c->P;
    b.sink = c->sink; b.emit = c->ext ? 1 : 0;
    b.BC = BaseCtx{idx->d_ref, idx->d_cstart, d_bases};
    if (c->ext && (d_trace != nullptr || c->ext_long)) { b.sink.best = nullptr; b.sink.tie = nullptr; }      // trace mode / long-read presets: every chain is handed over
    const uint32_t grid = std::min<uint32_t>(std::max<uint32_t>(n_tiles, 1), 256 * 8);
    // K2 only needs K1's output and nothing waits for it before the end of the call: it runs on a side stream, beside k_local_cluster /
    // k_expand / the sort classes.  Starting it late instead, beside the giant reads' DP kernels of the second pass (SCRUBBY_HIP_K2_LATE=1),
    // was measured with the extension filter on: K2 itself 64 -> 22 ms, but the step 408 -> 428 ms - it does not find room beside those
    // persistent grids and ends up serialised.
    static const int k2_late_env = getenv("SCRUBBY_HIP_K2_LATE") ? atoi(getenv("SCRUBBY_HIP_K2_LATE")) : -1;
    const bool k2_late = c->use_k1 && (c->par & 1) && k2_late_env > 0;
    K2Args kb = b;
    auto launch_k2 = [&]() -> sh_status {
        if (c->par & 1) { sh_status es = ensure_side_streams(c, 3, 3); if (es != SH_OK) return es; }
        hipStream_t sk = (c->par & 1) ? c->sx[3] : s;
        SH_HIP(hipEventRecord(c->evx[4], s));
        SH_HIP(hipStreamWaitEvent(sk, c->evx[4], 0));
        kb.work = c->d_work_small; kb.work_count = &c->d_ctr->n_small; kb.work_begin = 0;
        if (pair_pass) {      // flag-only: pair pass over all reads of the path, then the undecided ones, dense
            kb.leftover = c->d_work_small2; kb.leftover_count = &c->d_ctr->n_small2;
            hipLaunchKernelGGL(k_pair_pass, dim3(grid * 2), dim3(64), 0, sk, kb, pair_mode_small);
            kb.work = c->d_work_small2; kb.work_count = &c->d_ctr->n_small2;
        }
        hipLaunchKernelGGL(k_chain_small<K2_CAP>, dim3(grid), dim3(64), 0, sk, kb);
        SH_HIP(hipEventRecord(c->ev[2], sk));
        SH_HIP(hipEventRecord(c->evx[5], sk));
        return SH_OK;
    };
    if (c->use_k1 && !k2_late) { sh_status st = launch_k2(); if (st != SH_OK) return st; }
    else if (!c->use_k1) SH_HIP(hipEventRecord(c->ev[2], s));

    K3Args k{};
    k.offsets = d_offsets; k.positions = idx->d_positions; k.records = c->use_long ? c->d_lrec : c->d_records; k.seed_cap = c->seed_cap;
    k.seed_off = c->use_long ? c->d_seed_off : nullptr; k.sel_scratch = c->use_long ? (uint32_t *)c->d_mz_hash : nullptr;
    k.k1info = c->d_k1info; k.flags = d_flags; k.trace = d_trace; k.ctr = c->d_ctr; k.B = c->B; k.P = c->P;
    k.flag_only = d_trace == nullptr && !c->ext;      // SH_F_CIGAR: every chain is needed, no early exit
    k.sink = b.sink; k.emit = c->ext ? 1 : 0; k.BC = b.BC; k.t_mode = (c->ext && !c->ext_long && d_trace == nullptr) ? 1 : 0;
    k.dbg = getenv("SCRUBBY_HIP_DBG") ? atoi(getenv("SCRUBBY_HIP_DBG")) : 0;
    if (!getenv("SCRUBBY_HIP_AB_NOCHAIN")) k.dbg &= ~3;      // bits 0 / 1 switch the class kernels' chaining OFF (timing A/Bs: the answers are then wrong) - only with this second switch
    if (getenv("SCRUBBY_HIP_NO_PARFILL")) k.dbg |= 128;      // A/B: every cluster chained by the sequential DP
    k.top_max = TOPBT_MAX; k.pft_gmin = 32768u;
    if (const char *env = getenv("SCRUBBY_HIP_PFT_GMIN")) k.pft_gmin = (uint32_t)std::max(1, atoi(env));
    if (const char *env = getenv("SCRUBBY_HIP_TOPBT_MAX")) k.top_max = std::max(1, std::min(TOPBT_MAX, atoi(env)));
    if (getenv("SCRUBBY_HIP_NO_TOPBT")) k.dbg |= 512;        // A/B: clusters visited one by one even when the read's DP is done
    if (getenv("SCRUBBY_HIP_LOCUS_TOP1")) k.dbg |= 1024;     // tests: k_lr_locus keeps the largest run of windows only
    k.resketch_list = c->d_work_resketch;
    // long-read presets, flag-only: the anchors that cannot hold regs[0] never reach the sort classes (k_lr_locus).  Needs the probe (a read
    // with anchors left out is only ever proven mapped by it), a single chaining pass (max_occ <= mid_occ: whether mm_map_frag chains again
    // must not hinge on chains left out) and windows narrow enough to tell loci apart
    k.locus = c->ext_long && c->use_long && d_trace == nullptr && !getenv("SCRUBBY_HIP_NO_LOCUS") && !getenv("SCRUBBY_HIP_NO_PROBE") &&
              c->opts.max_clip_ratio >= 1.0f && c->P.max_occ <= c->P.mid_occ && c->locus_shift <= 20 && c->opts.min_cnt >= 1;
    {   // the shortest read for which `qlen - span > rmq_rescue_size || span > qlen * rmq_rescue_ratio` holds whatever the span (lr_chains_wave)
        int32_t q = 1;
        while (q < (1 << 30) && !((float)(q - c->LP.rmq_rescue_size) > (float)q * c->LP.rmq_rescue_ratio)) q = q < 64 ? q + 1 : q + q / 64;
        k.locus_min_qlen = c->LP.bw_long > c->LP.bw ? q : 0;
    }
    k.cl_lds = c->ext_long && c->P.max_iter <= RING_TMAX_ITER && !getenv("SCRUBBY_HIP_NO_CL_LDS");
    k.locus_shift = c->locus_shift; k.lr_drop = c->d_lr_drop; k.stage_x = c->d_stage_x; k.stage_q = c->d_stage_q; k.stage_cap = c->stage_cap;
    for (int i = 0; i < 3; ++i) k.locus_items[i] = c->d_locus[i];
    if (c->ext_long) SH_HIP(hipMemsetAsync(c->d_lr_drop, 0, n_reads * 4, s));
    uint32_t resk_done = 0;
    // pass 0 (mid_occ) over the reads K2 routed, pass 1 (max_occ) over the reads pass 0 could not chain;
    // reads that found no arena room come back in the next iteration
    int cur0 = 0, cur1 = 0;
    if (pair_pass) {      // the repeat path's list: reads two singleton seeds decide never reach k_expand; the rest moves to the other list
        K2Args pb = b;
        pb.work = c->d_big[0][0]; pb.work_count = &c->d_ctr->n_big[0];
        pb.leftover = c->d_big[0][1]; pb.leftover_count = &c->d_ctr->n_big_defer[0];
        hipLaunchKernelGGL(k_pair_pass, dim3(grid * 2), dim3(64), 0, s, pb, pair_mode_big);
        hipLaunchKernelGGL(k_pair_swap, dim3(1), dim3(1), 0, s, c->d_ctr);
        cur0 = 1;
    }
    if (k.t_mode && c->P.ext_lemma && c->use_k1 && !c->use_long && !(k.dbg & 256)) {      // reads whose singleton seeds' locus settles them
        K2Args pb = b;
        pb.work = c->d_big[0][cur0]; pb.work_count = &c->d_ctr->n_big[0];
        pb.leftover = c->d_big[0][cur0 ^ 1]; pb.leftover_count = &c->d_ctr->n_big_defer[0];
        hipLaunchKernelGGL(k_local_cluster, dim3(256 * 32), dim3(64), 0, s, pb);      // latency-bound list probes: every wave slot
        hipLaunchKernelGGL(k_pair_swap, dim3(1), dim3(1), 0, s, c->d_ctr);
        cur0 ^= 1;
    }
    bool first = true;
    Counters snap{};
    // the repeat path over the lists d_big[0][cur0] (pass 0, mid_occ) -> d_big[1][cur1] (pass 1, max_occ), until no read is deferred any more
    auto run_repeat_path = [&](K3Args &k, K2Args &b, int &cur0, int &cur1, uint32_t &resk_done) -> sh_status {
        for (int iter = 0;; ++iter) {
            SH_CHECK(iter < 256, SH_ERR_OOM, "chain arena (%llu MiB) too small; set SCRUBBY_HIP_ARENA_MB", (unsigned long long)(c->arena_bytes >> 20));
            k.pass = 0; k.max_occ = c->P.mid_occ;
            k.list = c->d_big[0][cur0]; k.list_count = &c->d_ctr->n_big[0];
            k.defer_list = c->d_big[0][cur0 ^ 1]; k.defer_count = &c->d_ctr->n_big_defer[0];
            k.next_list = c->d_big[1][cur1]; k.next_count = &c->d_ctr->n_big[1];
            sh_status st = big_pass(c, k, grid, s);
            if (st != SH_OK) return st;
            if (first && k2_late) { st = launch_k2(); if (st != SH_OK) return st; }
            k.pass = 1; k.max_occ = c->P.max_occ;
            k.list = c->d_big[1][cur1]; k.list_count = &c->d_ctr->n_big[1];
            k.defer_list = c->d_big[1][cur1 ^ 1]; k.defer_count = &c->d_ctr->n_big_defer[1];
            k.next_list = nullptr; k.next_count = nullptr;
            st = big_pass(c, k, grid, s);
            if (st != SH_OK) return st;
            // the rare reads K1 or k_expand could not take
            b.work = c->d_work_resketch; b.work_count = &c->d_ctr->n_resketch; b.work_begin = resk_done;
            hipLaunchKernelGGL(k_chain_large, dim3(grid), dim3(64), 0, s, b);
            if (first && c->use_k1) SH_HIP(hipStreamWaitEvent(s, c->evx[5], 0));
            SH_HIP(hipMemcpyAsync(c->h_ctr, c->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, s));
            SH_HIP(hipStreamSynchronize(s));
            SH_HIP(hipGetLastError());
            if (first) { snap = *c->h_ctr; first = false; c->dbg_ptr[0] = c->d_big[1][cur1]; c->dbg_n[0] = snap.n_big[1]; c->dbg_ptr[1] = c->dbg_ptr[2] = nullptr; c->dbg_n[1] = c->dbg_n[2] = 0; }
            if (k.dbg & 16) fprintf(stderr, "[dbg] iter %d resketch %u reasons %u %u %u pair tests between two singletons (dbg) %u segs %u big %u/%u defer %u/%u\n", iter, c->h_ctr->n_resketch, c->h_ctr->n_leg_reason[0], c->h_ctr->n_leg_reason[1], c->h_ctr->n_leg_reason[2], c->h_ctr->n_leg_reason[3], c->h_ctr->n_long_segs, c->h_ctr->n_big[0], c->h_ctr->n_big[1], c->h_ctr->n_big_defer[0], c->h_ctr->n_big_defer[1]);
            if (k.dbg & 16) fprintf(stderr, "[dbg] clusters chained by k_cluster_dp by class: %llu %llu %llu %llu, their anchors %llu %llu %llu %llu\n", c->h_ctr->cl_tot[0], c->h_ctr->cl_tot[1], c->h_ctr->cl_tot[2], c->h_ctr->cl_tot[3], c->h_ctr->cl_anchor_tot[0], c->h_ctr->cl_anchor_tot[1], c->h_ctr->cl_anchor_tot[2], c->h_ctr->cl_anchor_tot[3]);
            if (k.dbg & 16) fprintf(stderr, "[dbg] reads (anchors) by sort class: <=64 %llu (%llu), <=256 %llu (%llu), <=512 %llu (%llu), <=1024 %llu (%llu), <=2048 %llu (%llu), <=4096 %llu (%llu), giant %llu (%llu)\n", c->h_ctr->sort_tot[6], c->h_ctr->sort_anchor_tot[6], c->h_ctr->sort_tot[0], c->h_ctr->sort_anchor_tot[0], c->h_ctr->sort_tot[1], c->h_ctr->sort_anchor_tot[1], c->h_ctr->sort_tot[2], c->h_ctr->sort_anchor_tot[2], c->h_ctr->sort_tot[3], c->h_ctr->sort_anchor_tot[3], c->h_ctr->sort_tot[4], c->h_ctr->sort_anchor_tot[4], c->h_ctr->sort_tot[5], c->h_ctr->sort_anchor_tot[5]);
            if (k.dbg & 16) fprintf(stderr, "[dbg] local-cluster shortcut: tried %u, no singleton / filtered %u, singletons apart %u, window %u, K size %u, no margin %u, decided %u\n", c->h_ctr->ext_s3[0], c->h_ctr->ext_s3[1], c->h_ctr->ext_s3[2], c->h_ctr->ext_s3[3], c->h_ctr->ext_s3[4], c->h_ctr->ext_s3[5], c->h_ctr->ext_s3[7]);
            if (k.dbg & 16) fprintf(stderr, "[dbg] ring DP: chunks in window %llu, beyond %llu, far rescans %llu; clusters %llu (anchors %llu), with a max_skip break %llu (anchors %llu), widest window %llu, anchors of clusters with a window > 64: %llu, > 128: %llu\n", c->h_ctr->cl_dbg[0], c->h_ctr->cl_dbg[1], c->h_ctr->cl_dbg[2], c->h_ctr->cl_dbg[3], c->h_ctr->cl_dbg[6], c->h_ctr->cl_dbg[4], c->h_ctr->cl_dbg[5], c->h_ctr->cl_dbg[7], c->h_ctr->cl_dbg[8], c->h_ctr->cl_dbg[9]);
            if (k.dbg & 16) fprintf(stderr, "[dbg] parallel fill: reads done %llu (anchors %llu), not applicable %llu (anchors %llu), dirty anchors %llu; k_cluster_dp clusters prefilled %llu, sequential %llu (anchors %llu); read-level backtracks tried %llu, candidates listed %llu, given up (too many) %llu, chains visited %llu\n", c->h_ctr->pf_dbg[0], c->h_ctr->pf_dbg[3], c->h_ctr->pf_dbg[1], c->h_ctr->pf_dbg[4], c->h_ctr->pf_dbg[2], c->h_ctr->pf_dbg[5], c->h_ctr->pf_dbg[6], c->h_ctr->pf_dbg[7], c->h_ctr->pf_dbg[8], c->h_ctr->pf_dbg[11], c->h_ctr->pf_dbg[12], c->h_ctr->pf_dbg[10]);
            resk_done = c->h_ctr->n_resketch;
            const uint32_t d0 = c->h_ctr->n_big_defer[0], d1 = c->h_ctr->n_big_defer[1];
            if (d0 == 0 && d1 == 0) break;
            // a read deferred when it was alone in the arena can never fit
            SH_CHECK(!(d0 == c->h_ctr->n_big[0] && c->h_ctr->n_big[0] == 1 && d1 == 0) && !(d1 == c->h_ctr->n_big[1] && c->h_ctr->n_big[1] == 1 && d0 == 0),
                     SH_ERR_OOM, "chain arena (%llu MiB) too small for one read; set SCRUBBY_HIP_ARENA_MB", (unsigned long long)(c->arena_bytes >> 20));
            Counters z = *c->h_ctr;
            z.n_big[0] = d0; z.n_big[1] = d1; z.n_big_defer[0] = z.n_big_defer[1] = 0;
            SH_HIP(hipMemcpyAsync(c->d_ctr, &z, sizeof(Counters), hipMemcpyHostToDevice, s));
            cur0 ^= 1; cur1 ^= 1;
        }
        return SH_OK;
    };
    { sh_status st = run_repeat_path(k, b, cur0, cur1, resk_done); if (st != SH_OK) return st; }
    SH_HIP(hipEventRecord(c->ev[3], s));

    // legacy path deferrals: reads of k_chain_large that found no room in its arena slice come back alone
    auto legacy_defers = [&](K2Args &bb) -> sh_status {
        uint32_t n_defer = c->h_ctr->n_defer;
        int rounds = 0;
        while (n_defer > 0) {
            SH_CHECK(++rounds < 64, SH_ERR_OOM, "re-sketch arena too small; set SCRUBBY_HIP_ARENA_MB");
            std::swap(c->d_work_defer, c->d_work_defer2);
            Counters z = *c->h_ctr;
            z.n_defer = 0; z.arena_cursor = 0; z.n_resketch = n_defer;
            SH_HIP(hipMemcpyAsync(c->d_ctr, &z, sizeof(Counters), hipMemcpyHostToDevice, s));
            bb.work = c->d_work_defer2; bb.work_count = &c->d_ctr->n_resketch; bb.work_begin = 0; bb.work_defer = c->d_work_defer;
            hipLaunchKernelGGL(k_chain_large, dim3(grid), dim3(64), 0, s, bb);
            SH_HIP(hipMemcpyAsync(c->h_ctr, c->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, s));
            SH_HIP(hipStreamSynchronize(s));
            uint32_t nd = c->h_ctr->n_defer;
            SH_CHECK(nd < n_defer, SH_ERR_OOM, "re-sketch arena too small; set SCRUBBY_HIP_ARENA_MB");
            n_defer = nd;
        }
        return SH_OK;
    };
    { sh_status st = legacy_defers(b); if (st != SH_OK) return st; }
    uint32_t ext_list = 0, ext_regions = 0, ext_dropped = 0;
    float ms_ext = 0;
    if (c->ext_long) {
        ExtLongArgs x{};
        x.I.in.ref = idx->d_ref; x.I.in.cstart = idx->d_cstart; x.I.in.n_contigs = idx->n_contigs;
        x.I.in.bases = d_bases; x.I.in.offsets = d_offsets; x.I.in.cx = c->sink.cx; x.I.in.cq = c->sink.cq; x.I.in.recs = c->sink.recs; x.I.in.head = c->sink.head;
        x.I.rec = c->use_long ? c->d_lrec : c->d_records; x.I.k1info = c->d_k1info; x.I.seed_off = c->use_long ? c->d_seed_off : nullptr; x.I.seed_cap = c->seed_cap;
        x.P = c->LP; x.AP = c->AP; x.CP = c->P;
        x.AR.base = c->d_larena; x.AR.cap = c->larena_bytes; x.AR.cursor = &c->d_ctr->arena_cursor; x.AR.hdr = c->d_lhdr;
        x.list = c->d_ext_list; x.n_list = &c->d_ctr->ext_n_list; x.ctr = c->d_ctr; x.flags = d_flags; x.trace = d_trace; x.flag_only = d_trace == nullptr;
        x.clk = getenv("SCRUBBY_HIP_DBG") ? (getenv("SCRUBBY_HIP_DBG_EXACT") ? 3 : 1) : 0; x.probe = getenv("SCRUBBY_HIP_NO_PROBE") ? 0 : 1;
        x.drop = k.locus ? c->d_lr_drop : nullptr; x.fb_list = c->d_lr_fb; x.n_fb = &c->d_ctr->lr_n_fb;
        x.exact_list = c->d_lext_exact_list; x.n_exact = &c->d_ctr->lext_n_exact;
        x.exact_list2 = c->d_lext_exact_list2; x.n_exact2 = &c->d_ctr->lext_n_exact2;
        x.kind = c->d_lkind ? c->d_lkind + c->lkind_r0 : nullptr;
        { sh_status ps = pick_side_streams(c, s); if (ps != SH_OK) return ps; }
        SH_HIP(hipEventRecord(c->ev_ext[0], s));
        auto sync_ctr = [&]() -> sh_status {
            SH_HIP(hipMemcpyAsync(c->h_ctr, c->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, s));
            SH_HIP(hipStreamSynchronize(s));
            SH_HIP(hipGetLastError());
            return SH_OK;
        };
        uint32_t n_big_a = 0, n_exact_reads = 0, n_ondemand = 0;
        // Reads that outgrew the second size of a kernel's working memory too (minimap2 has no such limits): memory is allocated for them,
        // four times the size before, until they fit (a chain-anchor count in the millions, an alignment of max_sw_mat cells); only when
        // the device cannot give it do they keep their chain-level answer, counted and named in a warning.  xa: the arguments of the
        // second-size pass; h_ctr holds its counters.
        auto on_demand = [&](int phase, ExtLongArgs xa) -> sh_status {      // phase: 0 chains, 1 regions / alignment, 2 chains with the exact long join
            uint32_t n_un = c->h_ctr->lext_n_unres;
            n_ondemand += n_un;
            LongSizes q = phase == 2 ? c->lext_exact_sz[1] : c->lext_sz[phase * 2 + 1];
            int cur = 0;
            for (int round = 0; n_un > 0; ++round) {
                uint8_t *buf = nullptr;
                unsigned long long per = 0; uint32_t waves = 0;
                if (round < 5) {
                    q.cap_a = (uint32_t)std::min<uint64_t>((uint64_t)q.cap_a * 4, 1u << 28); q.cap_u = (uint32_t)std::min<uint64_t>((uint64_t)q.cap_u * 4, 1u << 26); q.cap_r = q.cap_u;
                    q.cap_m = (uint32_t)std::min<uint64_t>((uint64_t)q.cap_m * 4, 1u << 26); q.cap_k = (uint32_t)std::min<uint64_t>((uint64_t)q.cap_k * 4, 1u << 22);
                    q.cap_t = (uint32_t)std::min<uint64_t>((uint64_t)q.cap_t * 4, 1u << 28); q.cap_p = std::min<uint64_t>(q.cap_p * 4, 4ull << 30);
                    per = long_ws_carve(nullptr, nullptr, q);
                    for (waves = std::min<uint32_t>(n_un, 4); waves > 0 && hipMalloc(&buf, per * waves) != hipSuccess; waves >>= 1) { buf = nullptr; (void)hipGetLastError(); }
                }
                if (!buf) {      // no memory to be had: counted, warned about, chain-level answer
                    hipLaunchKernelGGL(k_lext_giveup, dim3(16), dim3(256), 0, s, (const uint32_t *)c->d_lext_unres[cur], n_un, d_flags, d_trace, c->d_lhdr, c->d_ctr, x.kind);
                    SH_HIP(hipMemsetAsync(&c->d_ctr->lext_n_unres, 0, 4, s));
                    return sync_ctr();
                }
                const uint32_t zero2[4] = {0, 0, n_un, 0};
                SH_HIP(hipMemcpyAsync(&c->d_ctr->lext_n_unres, zero2, 16, hipMemcpyHostToDevice, s));      // lext_n_unres, lext_ticket_unres, lext_n_unres_in
                xa.scratch = buf; xa.scratch_per_wave = per; xa.sz = q;
                xa.list = c->d_lext_unres[cur]; xa.n_list = &c->d_ctr->lext_n_unres_in; xa.ticket = &c->d_ctr->lext_ticket_unres;
                xa.big_list = nullptr; xa.n_big = nullptr; xa.part = 0; xa.unres_list = c->d_lext_unres[cur ^ 1]; xa.n_unres = &c->d_ctr->lext_n_unres;
                if (phase == 0) hipLaunchKernelGGL((k_long_chains<4096, false, true>), dim3(waves), dim3(64), 0, s, xa);
                else if (phase == 2) hipLaunchKernelGGL((k_long_chains<4096, true, false>), dim3(waves), dim3(64), 0, s, xa);
                else hipLaunchKernelGGL(k_regs_align_long, dim3(waves), dim3(64), 0, s, xa);
                sh_status st = sync_ctr();
                hipFree(buf);
                if (st != SH_OK) return st;
                if (c->h_ctr->ext_overflow == 1) return SH_SPLIT;
                n_un = c->h_ctr->lext_n_unres; cur ^= 1;
            }
            return SH_OK;
        };
        // One round of the stage: round 0 over every read with a chain; round 1 - flag-only calls whose anchors k_lr_locus thinned out - over the
        // reads whose answer could depend on what was left out, after the repeat path has chained them again with every anchor
        // lr_coop_fill's queue, counters and descriptors for one launch of an EXACT instance (three regions: the giants, E2a, E1 - launches that
        // may run side by side); zeroed on the launch's stream
        static const bool coop = !getenv("SCRUBBY_HIP_GIANTS_PLAIN") && !getenv("SCRUBBY_HIP_NO_COOP");
        auto coop_setup = [&](ExtLongArgs &xx, int which, hipStream_t st, uint32_t n_blocks) -> sh_status {
            constexpr uint32_t QCAP = 65536, NDESC = 512;
            constexpr size_t off_desc = 256, off_ready = off_desc + NDESC * sizeof(LongCoopDesc), off_items = off_ready + QCAP * 4, region = off_items + QCAP * 16;
            static_assert(sizeof(LongCoopDesc) % 8 == 0 && off_items % 16 == 0 && region % 256 == 0, "coop layout");
            if (!c->d_coop) SH_HIP(hipMalloc(&c->d_coop, 3 * region));
            uint8_t *base = c->d_coop + (size_t)which * region;
            SH_HIP(hipMemsetAsync(base, 0, off_items, st));
            xx.coop.q_res = (uint32_t *)base; xx.coop.q_head = (uint32_t *)(base + 64); xx.coop.active = (uint32_t *)(base + 128); xx.coop.n_reads = (uint32_t *)(base + 192);
            xx.coop.desc = (LongCoopDesc *)(base + off_desc); xx.coop.ready = (uint32_t *)(base + off_ready); xx.coop.items = (uint4 *)(base + off_items);
            xx.coop.cap = QCAP; xx.coop_on = n_blocks <= NDESC;
            return SH_OK;
        };
        auto ext_round = [&](int round) -> sh_status {
            SH_HIP(hipMemsetAsync(&c->d_ctr->arena_cursor, 0, 8, s));
            {
                ExtArgs xl{};
                xl.in = x.I.in; xl.list = c->d_ext_list; xl.n_list = x.n_list; xl.n_reads = n_reads; xl.ctr = c->d_ctr;
                if (round == 0) hipLaunchKernelGGL(k_lext_list, dim3((uint32_t)((n_reads + 255) / 256)), dim3(256), 0, s, xl, x.drop, (uint32_t)std::max(1, c->LP.min_cnt), x.fb_list, x.n_fb);
                else hipLaunchKernelGGL(k_lext_list_from, dim3(256), dim3(256), 0, s, xl, (const uint32_t *)c->d_lr_fb, (const uint32_t *)&c->d_ctr->lr_n_fb);
                // largest reads first (d_ext_redo: bin of each list entry; d_lext_big2: the ordered list, free until the second kernel's first pass ends)
                hipLaunchKernelGGL(k_lext_bins, dim3(256), dim3(256), 0, s, c->sink.recs, c->sink.head, c->d_ext_list, &c->d_ctr->ext_n_list, c->d_ext_redo, c->d_ctr->lext_hist, d_offsets);
                hipLaunchKernelGGL(k_lext_scan, dim3(1), dim3(1), 0, s, c->d_ctr->lext_hist);
                hipLaunchKernelGGL(k_lext_scatter, dim3(256), dim3(256), 0, s, c->d_ext_list, &c->d_ctr->ext_n_list, c->d_ext_redo, c->d_ctr->lext_hist, c->d_lext_sorted);
                x.list = c->d_lext_sorted;
            }
            // first kernel: final chains (tier 0, then the reads that outgrew it with the large working memory)
            {
                ExtLongArgs xa = x;
                xa.scratch = c->d_lext[0]; xa.scratch_per_wave = c->lext_per_wave[0]; xa.sz = c->lext_sz[0];
                xa.ticket = &c->d_ctr->ext_ticket; xa.big_list = c->d_lext_big; xa.n_big = &c->d_ctr->lext_n_big;
                // reads whose chain anchors outgrow the first size go straight to the large working memory, on a side stream beside the rest
                int bin_cut = 0;
                while (bin_cut < 31 && (2ull << bin_cut) <= c->lext_sz[0].cap_a) ++bin_cut;      // bin b holds totals in [2^b, 2^(b+1))
                if (const char *env = getenv("SCRUBBY_HIP_GIANT_BINS_DOWN")) bin_cut = std::max(1, bin_cut - atoi(env));
                xa.hist = c->d_ctr->lext_hist; xa.bin_cut = bin_cut; xa.part = 1;
                ExtLongArgs xg = xa;
                xg.scratch = c->d_lext[1]; xg.scratch_per_wave = c->lext_per_wave[1]; xg.sz = c->lext_sz[1];
                xg.ticket = &c->d_ctr->lext_ticket_g; xg.part = 2;
                // The giants - a dozen reads per half million, 10^5 chain anchors each, half a second and more of one wave apiece - are the pair's
                // critical path.  With the 512-anchor ring their window reaches back into HBM at nearly every step, and those trips take several
                // times longer while the main grid loads the memory system; so they get the 4096-anchor ring (99 KB of LDS a wave, no trip
                // behind it), and because a CU the main grid has filled has no such room left, the main grid is held back until the giants'
                // blocks have begun (k_wait_started: bounded, ~2 ms at most).  The main grid's blocks (13 KB) fit beside them.
                uint32_t g_waves = std::min<uint32_t>(c->lext_waves[1], 64u);
                if (const char *env = getenv("SCRUBBY_HIP_GIANT_WAVES")) g_waves = std::min<uint32_t>(c->lext_waves[1], (uint32_t)std::max(1, atoi(env)));
                xg.started = &c->d_ctr->lext_started;
                SH_HIP(hipMemsetAsync(&c->d_ctr->lext_started, 0, 4, s));
                SH_HIP(hipEventRecord(c->evx[0], s));
                hipStream_t sg = c->sx[c->side_pick[0]];
                SH_HIP(hipStreamWaitEvent(sg, c->evx[0], 0));
                const uint32_t m_waves = c->lext_waves[0];
                // (the giants on the EXACT instance at once: the tree is only kept over the stretches that ask it, lr_rmq_fill - a giant with a tie
                // is not chained twice, and the exact passes lose their longest reads)
                static const bool giants_exact = !getenv("SCRUBBY_HIP_GIANTS_PLAIN");
                if (coop) { sh_status cs = coop_setup(xg, 0, sg, g_waves); if (cs != SH_OK) return cs; }      // the giants' waves share the long join of the launch's largest reads (lr_coop_fill, sh_long.h)
                if (giants_exact) hipLaunchKernelGGL((k_long_chains<4096, true, false>), dim3(g_waves), dim3(64), 0, sg, xg);
                else hipLaunchKernelGGL((k_long_chains<4096, false, false>), dim3(g_waves), dim3(64), 0, sg, xg);
                SH_HIP(hipEventRecord(c->evx[1], sg));
                hipLaunchKernelGGL(k_wait_started, dim3(1), dim3(64), 0, s, (const uint32_t *)&c->d_ctr->lext_started, g_waves, 2000u);
                hipLaunchKernelGGL((k_long_chains<512, false, false>), dim3(m_waves), dim3(64), 0, s, xa);
                SH_HIP(hipStreamWaitEvent(s, c->evx[1], 0));
                sh_status st = sync_ctr(); if (st != SH_OK) return st;
                if (coop && getenv("SCRUBBY_HIP_DBG")) {
                    uint32_t h[64] = {};
                    SH_HIP(hipMemcpy(h, c->d_coop, sizeof(h), hipMemcpyDeviceToHost));
                    fprintf(stderr, "[dbg] long join shared among the giants' waves: %u reads in %u runs (%u taken off the queue)\n", h[48], h[0], h[16]);
                }
                if (c->h_ctr->ext_overflow == 1) return SH_SPLIT;       // hand-over buffers or arena full: the caller cuts the chunk in two
                if (c->h_ctr->lext_n_big > 0) {
                    xa.scratch = c->d_lext[1]; xa.scratch_per_wave = c->lext_per_wave[1]; xa.sz = c->lext_sz[1];
                    xa.list = c->d_lext_big; xa.n_list = &c->d_ctr->lext_n_big; xa.ticket = &c->d_ctr->lext_ticket_big; xa.big_list = nullptr; xa.n_big = nullptr; xa.part = 0;
                    xa.unres_list = c->d_lext_unres[0]; xa.n_unres = &c->d_ctr->lext_n_unres;
                    hipLaunchKernelGGL((k_long_chains<4096, false, true>), dim3(c->lext_waves[1]), dim3(64), 0, s, xa);
                    st = sync_ctr(); if (st != SH_OK) return st;
                    if (c->h_ctr->ext_overflow == 1) return SH_SPLIT;
                    st = on_demand(0, xa); if (st != SH_OK) return st;
                }
            }
            n_big_a += c->h_ctr->lext_n_big;
            if (c->h_ctr->lext_n_exact + c->h_ctr->lext_n_exact2 > 0) {
                // The long join of these reads met two candidates of equal priority in a way that can change the chains: once more with
                // upstream's main tree beside the scan (sh_rmq_tree.h), maintained by the whole wave and asked at the ties.  The tree holds the
                // look-back window only and lives in LDS, so the passes run on the ordinary working memory of the chains kernel:
                //   E1   the 1024-anchor ring + a 1664-node tree (79 KB of LDS a wave, two waves to a CU), first size, largest reads first;
                //   E2a  the reads that needed the 4096-anchor ring in the first pass already, and those whose join holds more than 60 000 anchors: that ring + a 1792-node tree (156 KB, one wave to a
                //        CU), second size - on a side stream BESIDE E1 (launched first, so that its few blocks find their LDS);
                //   E2b  what outgrew E1's ring or working memory after all: the same instance, after E1;
                //   E3   (SCRUBBY_HIP_RMQ_ONE_LANE=1) what outgrew E2's ring or tree: both trees on one lane over node pools in the wave's scratch
                //        (lr_rmq_fill_tree) - the literal mg_lchain_rmq, for windows no LDS holds; then memory on demand for what outgrew the sizes.
                const uint32_t n_e1 = c->h_ctr->lext_n_exact, n_e2a = c->h_ctr->lext_n_exact2;
                n_exact_reads += n_e1 + n_e2a;
                const auto t_ex = std::chrono::steady_clock::now();
                ExtLongArgs xe = x;
                xe.part = 0; xe.exact_list = nullptr; xe.n_exact = nullptr; xe.exact_list2 = nullptr; xe.n_exact2 = nullptr;
                SH_HIP(hipMemsetAsync(&c->d_ctr->lext_n_big, 0, 8, s));       // lext_n_big, lext_ticket_big: E1's list of deferred reads
                SH_HIP(hipMemsetAsync(&c->d_ctr->lext_n_big2, 0, 8, s));      // lext_n_big2, lext_ticket_big2 (the regions kernel's list: free until it starts): E2's
                hipStream_t se = c->sx[c->side_pick[0]];
                if (n_e2a > 0) {
                    ExtLongArgs x2 = xe;
                    x2.scratch = c->d_lext[1]; x2.scratch_per_wave = c->lext_per_wave[1]; x2.sz = c->lext_sz[1];
                    x2.list = c->d_lext_exact_list2; x2.n_list = &c->d_ctr->lext_n_exact2; x2.ticket = &c->d_ctr->lext_ticket_exact2;
                    x2.big_list = c->d_lext_big2; x2.n_big = &c->d_ctr->lext_n_big2; x2.unres_list = nullptr; x2.n_unres = nullptr;
                    x2.started = &c->d_ctr->lext_started;
                    const uint32_t w2 = std::min<uint32_t>({c->lext_waves[1], (uint32_t)c->n_cu / 2u, n_e2a});      // half of the CUs at most: E1's blocks need the others
                    SH_HIP(hipMemsetAsync(&c->d_ctr->lext_started, 0, 4, s));
                    SH_HIP(hipEventRecord(c->evx[0], s));
                    SH_HIP(hipStreamWaitEvent(se, c->evx[0], 0));
                    if (coop) { sh_status cs = coop_setup(x2, 1, se, w2); if (cs != SH_OK) return cs; }
                    hipLaunchKernelGGL((k_long_chains<4096, true, false>), dim3(w2), dim3(64), 0, se, x2);
                    SH_HIP(hipEventRecord(c->evx[1], se));
                    hipLaunchKernelGGL(k_wait_started, dim3(1), dim3(64), 0, s, (const uint32_t *)&c->d_ctr->lext_started, w2, 2000u);
                }
                if (n_e1 > 0) {
                    xe.scratch = c->d_lext[0]; xe.scratch_per_wave = c->lext_per_wave[0]; xe.sz = c->lext_sz[0];
                    // largest first, like every list of the stage: the pass ends with its slowest read (a read of 10^5 chain anchors is more than a second of one wave)
                    SH_HIP(hipMemsetAsync(c->d_ctr->lext_hist, 0, sizeof(c->d_ctr->lext_hist), s));
                    hipLaunchKernelGGL(k_lext_bins, dim3(64), dim3(256), 0, s, c->sink.recs, c->sink.head, c->d_lext_exact_list, &c->d_ctr->lext_n_exact, c->d_ext_redo, c->d_ctr->lext_hist, d_offsets);
                    hipLaunchKernelGGL(k_lext_scan, dim3(1), dim3(1), 0, s, c->d_ctr->lext_hist);
                    hipLaunchKernelGGL(k_lext_scatter, dim3(64), dim3(256), 0, s, c->d_lext_exact_list, &c->d_ctr->lext_n_exact, c->d_ext_redo, c->d_ctr->lext_hist, c->d_lext_esorted);
                    xe.list = c->d_lext_esorted; xe.n_list = &c->d_ctr->lext_n_exact; xe.ticket = &c->d_ctr->lext_ticket_exact;
                    xe.big_list = c->d_lext_big; xe.n_big = &c->d_ctr->lext_n_big; xe.unres_list = nullptr; xe.n_unres = nullptr;
                    const uint32_t w1 = std::min<uint32_t>({c->lext_waves[0], 2u * (uint32_t)c->n_cu, n_e1});
                    ExtLongArgs x1 = xe;
                    if (coop) { sh_status cs = coop_setup(x1, 2, s, w1); if (cs != SH_OK) return cs; }
                    hipLaunchKernelGGL((k_long_chains<1024, true, false>), dim3(w1), dim3(64), 0, s, x1);
                }
                if (n_e2a > 0) SH_HIP(hipStreamWaitEvent(s, c->evx[1], 0));
                sh_status st = sync_ctr(); if (st != SH_OK) return st;
                if (c->h_ctr->ext_overflow == 1) return SH_SPLIT;
                const uint32_t n_e2b = c->h_ctr->lext_n_big;
                if (n_e2b > 0) {
                    xe.scratch = c->d_lext[1]; xe.scratch_per_wave = c->lext_per_wave[1]; xe.sz = c->lext_sz[1];
                    xe.list = c->d_lext_big; xe.n_list = &c->d_ctr->lext_n_big; xe.ticket = &c->d_ctr->lext_ticket_big;
                    xe.big_list = c->d_lext_big2; xe.n_big = &c->d_ctr->lext_n_big2; xe.unres_list = nullptr; xe.n_unres = nullptr; xe.started = nullptr;
                    hipLaunchKernelGGL((k_long_chains<4096, true, false>), dim3(std::min<uint32_t>({c->lext_waves[1], (uint32_t)c->n_cu, n_e2b})), dim3(64), 0, s, xe);
                    st = sync_ctr(); if (st != SH_OK) return st;
                    if (c->h_ctr->ext_overflow == 1) return SH_SPLIT;
                }
                const uint32_t n_e3 = c->h_ctr->lext_n_big2;
                if (n_e3 > 0) {      // (only with SCRUBBY_HIP_RMQ_ONE_LANE, or reads beyond the second working-memory size)
                    xe.scratch = c->d_lext_exact[1]; xe.scratch_per_wave = c->lext_exact_per_wave[1]; xe.sz = c->lext_exact_sz[1];
                    xe.list = c->d_lext_big2; xe.n_list = &c->d_ctr->lext_n_big2; xe.ticket = &c->d_ctr->lext_ticket_big2; xe.big_list = nullptr; xe.n_big = nullptr; xe.started = nullptr;
                    xe.unres_list = c->d_lext_unres[0]; xe.n_unres = &c->d_ctr->lext_n_unres;
                    hipLaunchKernelGGL((k_long_chains<4096, true, false>), dim3(std::min<uint32_t>(c->lext_exact_waves[1], (uint32_t)c->n_cu)), dim3(64), 0, s, xe);
                    st = sync_ctr(); if (st != SH_OK) return st;
                    if (c->h_ctr->ext_overflow == 1) return SH_SPLIT;
                    if (c->h_ctr->lext_n_unres > 0) { st = on_demand(2, xe); if (st != SH_OK) return st; }
                }
                SH_HIP(hipMemsetAsync(&c->d_ctr->lext_n_big2, 0, 8, s));
                if (getenv("SCRUBBY_HIP_DBG")) fprintf(stderr, "[dbg] exact long join: %u reads on the 1024-anchor ring beside %u on the 4096-anchor ring (%u with tied priorities so far), %u more on the large ring afterwards, %u beyond it, %.1f ms\n", n_e1, n_e2a, c->h_ctr->lext_rmq_tie, n_e2b, n_e3,
                                                       std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_ex).count());
            }
            SH_CHECK(c->h_ctr->ext_overflow == 0, SH_ERR_OOM, "long-read extension stage: internal overflow code %u", c->h_ctr->ext_overflow);
            ext_list += c->h_ctr->ext_n_list;
            // second kernel: regions and alignment
            {
                ExtLongArgs xb = x;
                xb.scratch = c->d_lext[2]; xb.scratch_per_wave = c->lext_per_wave[2]; xb.sz = c->lext_sz[2];
                xb.ticket = &c->d_ctr->lext_ticket_b; xb.big_list = c->d_lext_big2; xb.n_big = &c->d_ctr->lext_n_big2;
                // The launch of the second size runs BESIDE the first (side stream), taking reads off the list as the first launch puts them
                // there: the pass used to begin when the first had ended, and it is one alignment of ~4 * 10^8 cells on one wave - a third of a
                // second during which nothing else ran.  Both grids are resident together (two waves per SIMD each: 8 per CU); what the
                // follower leaves (it gives up after a bounded number of looks) the launch after the first one takes, as before.
                static const bool no_follow = getenv("SCRUBBY_HIP_NO_FOLLOW") != nullptr;
                // (both grids must be RESIDENT together - a follower that holds the slot a block of the first launch waits for would wait for that
                // launch to end: the first launch gives up the slots the follower needs; its reads are drawn by ticket, so fewer blocks lose nothing)
                const uint32_t slots = 8u * (uint32_t)c->n_cu;
                const bool follow = !no_follow && c->lext_waves[3] <= slots / 4;
                const uint32_t w_first = follow ? std::min<uint32_t>(c->lext_waves[2], slots - c->lext_waves[3]) : c->lext_waves[2];
                if (follow) {
                    SH_HIP(hipMemsetAsync(c->d_lext_big2, 0xff, (size_t)n_reads * 4, s));
                    SH_HIP(hipMemsetAsync(&c->d_ctr->lext_t0_done, 0, 4, s));
                    SH_HIP(hipEventRecord(c->evx[2], s));
                }
                hipLaunchKernelGGL(k_regs_align_long, dim3(w_first), dim3(64), 0, s, xb);
                if (follow) {
                    // submitted in this order - first launch, its end mark, then the follower: two streams may share a hardware queue, and
                    // a follower submitted ahead of the launch it follows would then sit in front of it until its looks ran out
                    hipLaunchKernelGGL(k_set_word, dim3(1), dim3(64), 0, s, &c->d_ctr->lext_t0_done, 1u);
                    ExtLongArgs xf = xb;
                    xf.scratch = c->d_lext[3]; xf.scratch_per_wave = c->lext_per_wave[3]; xf.sz = c->lext_sz[3];
                    xf.list = c->d_lext_big2; xf.n_list = &c->d_ctr->lext_n_big2; xf.ticket = &c->d_ctr->lext_ticket_big2; xf.big_list = nullptr; xf.n_big = nullptr;
                    xf.unres_list = c->d_lext_unres[0]; xf.n_unres = &c->d_ctr->lext_n_unres; xf.follow_done = &c->d_ctr->lext_t0_done;
                    hipStream_t sf = c->sx[c->side_pick[1]];
                    SH_HIP(hipStreamWaitEvent(sf, c->evx[2], 0));
                    hipLaunchKernelGGL(k_regs_align_long, dim3(c->lext_waves[3]), dim3(64), 0, sf, xf);
                    SH_HIP(hipEventRecord(c->evx[3], sf));
                    SH_HIP(hipStreamWaitEvent(s, c->evx[3], 0));
                }
                sh_status st = sync_ctr(); if (st != SH_OK) return st;
                if (c->h_ctr->lext_n_big2 > 0) {
                    xb.scratch = c->d_lext[3]; xb.scratch_per_wave = c->lext_per_wave[3]; xb.sz = c->lext_sz[3];
                    xb.list = c->d_lext_big2; xb.n_list = &c->d_ctr->lext_n_big2; xb.ticket = &c->d_ctr->lext_ticket_big2; xb.big_list = nullptr; xb.n_big = nullptr;
                    xb.unres_list = c->d_lext_unres[0]; xb.n_unres = &c->d_ctr->lext_n_unres;
                    hipLaunchKernelGGL(k_regs_align_long, dim3(c->lext_waves[3]), dim3(64), 0, s, xb);
                    st = sync_ctr(); if (st != SH_OK) return st;
                    st = on_demand(1, xb); if (st != SH_OK) return st;
                }
                n_big_a += c->h_ctr->lext_n_big2;
            }
            return SH_OK;
        };
        { sh_status st = ext_round(0); if (st != SH_OK) return st; }
        const uint32_t n_fb = c->h_ctr->lr_n_fb;
        if (k.locus && getenv("SCRUBBY_HIP_DBG")) fprintf(stderr, "[dbg] anchors by locus: %u reads thinned out (%llu of %llu anchors kept); %u reads redone with every anchor (one chain and more possible %u, short read %u, top score within reach of what was left out %u, no chain after the join %u, probe undecided %u, no probe %u, no chain among the anchors kept %u)\n",
                                                        c->h_ctr->lr_locus_reads, c->h_ctr->lr_locus_kept, c->h_ctr->lr_locus_in, n_fb, c->h_ctr->lr_fb_why[0], c->h_ctr->lr_fb_why[1], c->h_ctr->lr_fb_why[2], c->h_ctr->lr_fb_why[3], c->h_ctr->lr_fb_why[4], c->h_ctr->lr_fb_why[5], c->h_ctr->lr_fb_why[6]);
        if (getenv("SCRUBBY_HIP_DBG")) fprintf(stderr, "[dbg] probes that gave up (first round): region %u, window %u, filling over 4 M cells %u, direction bytes %u, z-drop / empty %u, CIGAR room %u, no proof within six fillings %u\n",
                                               c->h_ctr->lr_probe_why[1], c->h_ctr->lr_probe_why[2], c->h_ctr->lr_probe_why[3], c->h_ctr->lr_probe_why[4], c->h_ctr->lr_probe_why[5], c->h_ctr->lr_probe_why[6], c->h_ctr->lr_probe_why[7]);
        if (stats && k.locus) { stats->n_locus_reads += c->h_ctr->lr_locus_reads; stats->n_locus_redone += n_fb; }
        if (n_fb > 0) {
            // the reads whose answer could depend on the anchors left out: the repeat path once more with every anchor, then the stage again.
            // Hand-over buffers, arena and lists start over; their first visit is forgotten (chain lists emptied, lr_drop cleared).
            Counters z = *c->h_ctr;
            memset(z.ext_n_recs, 0, sizeof(z.ext_n_recs)); memset(z.ext_n_anch, 0, sizeof(z.ext_n_anch)); memset(z.lext_hist, 0, sizeof(z.lext_hist));
            z.n_defer = 0; z.arena_cursor = 0;
            z.n_big[0] = n_fb; z.n_big[1] = 0; z.n_big_defer[0] = z.n_big_defer[1] = 0;
            z.ext_n_list = 0; z.ext_ticket = 0; z.lext_ticket_g = 0; z.lext_n_big = 0; z.lext_ticket_big = 0; z.lext_n_big2 = 0; z.lext_ticket_big2 = 0; z.lext_ticket_b = 0; z.lext_n_unres = 0; z.lext_ticket_unres = 0; z.lext_n_exact = 0; z.lext_ticket_exact = 0; z.lext_n_exact2 = 0; z.lext_ticket_exact2 = 0;
            SH_HIP(hipMemcpyAsync(c->d_ctr, &z, sizeof(Counters), hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(k_lext_forget, dim3(64), dim3(256), 0, s, (const uint32_t *)c->d_lr_fb, n_fb, c->sink.head, c->d_lr_drop, &c->d_ctr->lr_fb_had);
            SH_HIP(hipMemcpyAsync(c->d_big[0][0], c->d_lr_fb, (size_t)n_fb * 4, hipMemcpyDeviceToDevice, s));
            K3Args kf = k; kf.locus = 0; kf.quiet = 2;      // their chain-level count was taken back (k_lext_forget); the other statistics stay as they are
            K2Args rb = b; rb.quiet = 2;
            int f0 = 0, f1 = 0;
            { sh_status st = run_repeat_path(kf, rb, f0, f1, resk_done); if (st != SH_OK) return st; }
            rb.work_defer = c->d_work_defer;
            { sh_status st = legacy_defers(rb); if (st != SH_OK) return st; }
            x.drop = nullptr;
            { sh_status st = ext_round(1); if (st != SH_OK) return st; }
        }
        if (c->h_ctr->lext_unresolved) {
            static bool warned = false;
            if (!warned) { warned = true; fprintf(stderr, "[scrubby-hip] WARNING: %u read(s) outgrew the extension stage's largest working memory (e.g. read %u of its batch, code %u: 18 chains, 20 read length, 21 chain anchors, 22 RMQ window, 23 seeds, 24 regions, 27-30 alignment window, 32 direction bytes); they keep their chain-level answer (mapped)\n", c->h_ctr->lext_unresolved, c->h_ctr->lext_err_read, c->h_ctr->lext_err_code); }
        }
        if (getenv("SCRUBBY_HIP_DBG")) fprintf(stderr, "[dbg] long-read extension stage: %u reads with chains, %u re-chained (RMQ), %u with tied RMQ priorities, %u needed the large scratch, %u regions aligned, %u reads dropped\n",
                                               ext_list, c->h_ctr->lext_rechained, c->h_ctr->lext_rmq_tie, n_big_a, c->h_ctr->ext_regions, c->h_ctr->ext_dropped);
        if (getenv("SCRUBBY_HIP_DBG")) {
            unsigned long long tot = 0, mr = 0, ma = 0, sr = 0, sa = 0;
            for (int i = 0; i < LR_NCLK; ++i) tot += c->h_ctr->lext_clk[i];
            fprintf(stderr, "[dbg] long-read extension stage, share of wave time: gather %.1f  rmq-sort %.1f  rmq-fill %.1f  backtrack+compact %.1f  gen_regs %.1f  parent/select %.1f  squeeze %.1f  region set-up %.1f  ksw %.1f  z-drop test %.1f  update_extra %.1f  staging %.1f %%  (total %.1f wave-s at 100 MHz)\n",
                    100. * c->h_ctr->lext_clk[0] / (tot + 1), 100. * c->h_ctr->lext_clk[1] / (tot + 1), 100. * c->h_ctr->lext_clk[2] / (tot + 1), 100. * c->h_ctr->lext_clk[3] / (tot + 1), 100. * c->h_ctr->lext_clk[4] / (tot + 1), 100. * c->h_ctr->lext_clk[5] / (tot + 1),
                    100. * c->h_ctr->lext_clk[6] / (tot + 1), 100. * c->h_ctr->lext_clk[7] / (tot + 1), 100. * c->h_ctr->lext_clk[8] / (tot + 1), 100. * c->h_ctr->lext_clk[9] / (tot + 1), 100. * c->h_ctr->lext_clk[10] / (tot + 1), 100. * c->h_ctr->lext_clk[11] / (tot + 1), tot / 1e8);
            fprintf(stderr, "[dbg] RMQ: %llu reads, %llu anchors (largest read %llu), per anchor: %.2f ring blocks, %.3f trips behind the ring with %.2f old blocks, list length %.1f, %.2f inner chunks\n",
                    c->h_ctr->lext_d[6], c->h_ctr->lext_d[0], c->h_ctr->lext_d[7], (double)c->h_ctr->lext_d[1] / (c->h_ctr->lext_d[0] + 1), (double)c->h_ctr->lext_d[2] / (c->h_ctr->lext_d[0] + 1),
                    (double)c->h_ctr->lext_d[3] / (c->h_ctr->lext_d[0] + 1), (double)c->h_ctr->lext_d[4] / (c->h_ctr->lext_d[0] + 1), (double)c->h_ctr->lext_d[5] / (c->h_ctr->lext_d[0] + 1));
            fprintf(stderr, "[dbg] chains kernel: slowest read %.1f ms (%llu chains after the long join; read %llu of the chunk, %llu bases), all reads %.1f wave-ms\n", (c->h_ctr->lext_slow >> 24) / 1e5, c->h_ctr->lext_slow & 0xffffff,
                    c->h_ctr->lext_slow3 & 0xffffffffull, c->h_ctr->lext_slow2 & 0xffffffffull, c->h_ctr->lext_kernel_sum / 1e5);
            { unsigned long long tb = 0; for (int i = 0; i < LR_NCLK; ++i) tb += c->h_ctr->lext_clk_big[i];
              fprintf(stderr, "[dbg] regions kernel, second size: gen_regs %.1f  parent/select %.1f  squeeze %.1f  set-up %.1f  ksw %.1f  z-drop %.1f  update_extra %.1f  staging %.1f %% of %.1f wave-s\n", 100. * c->h_ctr->lext_clk_big[4] / (tb + 1), 100. * c->h_ctr->lext_clk_big[5] / (tb + 1), 100. * c->h_ctr->lext_clk_big[6] / (tb + 1), 100. * c->h_ctr->lext_clk_big[7] / (tb + 1), 100. * c->h_ctr->lext_clk_big[8] / (tb + 1), 100. * c->h_ctr->lext_clk_big[9] / (tb + 1), 100. * c->h_ctr->lext_clk_big[10] / (tb + 1), 100. * c->h_ctr->lext_clk_big[11] / (tb + 1), tb / 1e8); }
            { unsigned long long tg = c->h_ctr->lext_clk_big[0] + c->h_ctr->lext_clk_big[1] + c->h_ctr->lext_clk_big[2] + c->h_ctr->lext_clk_big[3];
              fprintf(stderr, "[dbg] giants rmq-fill sections (wave-s): insert %.2f  trim %.2f  query %.2f  tally %.2f  score+inner %.2f\n", c->h_ctr->lext_clk_big[4] / 1e8, c->h_ctr->lext_clk_big[5] / 1e8, c->h_ctr->lext_clk_big[6] / 1e8, c->h_ctr->lext_clk_big[7] / 1e8, c->h_ctr->lext_clk_big[8] / 1e8);
              fprintf(stderr, "[dbg] all rmq-fill sections (wave-s): insert %.2f  trim %.2f  query %.2f  tally %.2f  score+inner %.2f\n", c->h_ctr->lext_clk[4] / 1e8, c->h_ctr->lext_clk[5] / 1e8, c->h_ctr->lext_clk[6] / 1e8, c->h_ctr->lext_clk[7] / 1e8, c->h_ctr->lext_clk[8] / 1e8);
              fprintf(stderr, "[dbg] chains kernel, the giants: gather %.1f  rmq-sort %.1f  rmq-fill %.1f  backtrack+compact %.1f %% of %.2f wave-s\n", 100. * c->h_ctr->lext_clk_big[0] / (tg + 1), 100. * c->h_ctr->lext_clk_big[1] / (tg + 1), 100. * c->h_ctr->lext_clk_big[2] / (tg + 1), 100. * c->h_ctr->lext_clk_big[3] / (tg + 1), tg / 1e8); }
            { const unsigned long long *dd = c->h_ctr->lext_d_big;
              fprintf(stderr, "[dbg] RMQ, the giants: %llu reads, %llu anchors (largest read %llu), per anchor: %.2f ring blocks, %.3f trips behind the ring with %.2f old blocks, list length %.1f, %.2f inner chunks\n",
                      dd[6], dd[0], dd[7], (double)dd[1] / (dd[0] + 1), (double)dd[2] / (dd[0] + 1), (double)dd[3] / (dd[0] + 1), (double)dd[4] / (dd[0] + 1), (double)dd[5] / (dd[0] + 1)); }
            fprintf(stderr, "[dbg] regions kernel, second size: slowest read %.1f ms (%llu bases), all reads %.1f wave-ms\n", (c->h_ctr->lext_slow_part[3] >> 32) / 1e5, c->h_ctr->lext_slow_part[3] & 0xffffffffull, c->h_ctr->lext_sum_part[3] / 1e5);
            fprintf(stderr, "[dbg] regions kernel, second size, largest per-read time of each step (ms): gen_regs %.1f  parent/select %.1f  squeeze %.1f  set-up %.1f  ksw %.1f  z-drop %.1f  update_extra %.1f  staging %.1f\n", c->h_ctr->lext_phase_max[4] / 1e5, c->h_ctr->lext_phase_max[5] / 1e5, c->h_ctr->lext_phase_max[6] / 1e5, c->h_ctr->lext_phase_max[7] / 1e5, c->h_ctr->lext_phase_max[8] / 1e5, c->h_ctr->lext_phase_max[9] / 1e5, c->h_ctr->lext_phase_max[10] / 1e5, c->h_ctr->lext_phase_max[11] / 1e5);
            for (int pp = 0; pp < 3; ++pp) fprintf(stderr, "[dbg] chains kernel, part %d (0 lists of later passes, 1 all but the giants, 2 the giants): slowest read %.1f ms, all reads %.1f wave-ms\n", pp, (c->h_ctr->lext_slow_part[pp] >> 32) / 1e5, c->h_ctr->lext_sum_part[pp] / 1e5);
            for (int i = 0; i < SINK_SHARDS; ++i) { sr += c->h_ctr->ext_n_recs[i]; sa += c->h_ctr->ext_n_anch[i]; mr = std::max<unsigned long long>(mr, c->h_ctr->ext_n_recs[i]); ma = std::max<unsigned long long>(ma, c->h_ctr->ext_n_anch[i]); }
            fprintf(stderr, "[dbg] chain hand-over: %llu chains, %llu anchors; fullest shard %llu / %u chains, %llu / %llu anchors\n", sr, sa, mr, c->sink.cap_recs, ma, c->sink.cap_anch);
        }
        SH_HIP(hipEventRecord(c->ev_ext[1], s));
        SH_HIP(hipEventSynchronize(c->ev_ext[1]));
        ext_regions = c->h_ctr->ext_regions; ext_dropped = c->h_ctr->ext_dropped;
        if (stats) { stats->n_ext_unresolved += c->h_ctr->lext_unresolved; stats->n_rmq_rechained += c->h_ctr->lext_rechained; stats->n_rmq_tied += c->h_ctr->lext_rmq_tie; stats->n_rmq_exact += n_exact_reads + c->h_ctr->lext_exact_direct; stats->n_ext_ondemand += n_ondemand; stats->n_rmq_open += c->h_ctr->lext_rmq_open; }
        hipEventElapsedTime(&ms_ext, c->ev_ext[0], c->ev_ext[1]);
    } else if (c->ext) {
        ExtArgs x{};
        x.in.ref = idx->d_ref; x.in.cstart = idx->d_cstart; x.in.n_contigs = idx->n_contigs;
        x.in.bases = d_bases; x.in.offsets = d_offsets; x.in.cx = c->sink.cx; x.in.cq = c->sink.cq; x.in.recs = c->sink.recs; x.in.head = c->sink.head;
        x.P = c->AP; x.scratch = c->d_ext_scratch; x.scratch_per_wave = c->ext_scratch_per_wave; x.max_read_len = c->max_read_len; x.reg_cap = c->ext_reg_cap;
        x.list = c->d_ext_list; x.n_list = &c->d_ctr->ext_n_list; x.ticket = &c->d_ctr->ext_ticket;
        x.ctr = c->d_ctr; x.flags = d_flags; x.trace = d_trace; x.flag_only = d_trace == nullptr; x.n_reads = n_reads;
        x.best = c->sink.best; x.tie = c->sink.tie; x.redo = c->d_ext_redo;
        x.unres_list = c->d_ext_unres[0]; x.n_unres = &c->d_ctr->ext_n_unres;
        uint32_t n_sr_ondemand = 0;
        // minimap2 has no limit on a read's chains or regions; k_regs_align's per-wave working memory has (ext_reg_cap).  Reads beyond it are
        // listed and redone with memory allocated for them, four times the last size per round, until none is left; only when the device
        // cannot give it do they keep their chain-level answer (counted: sh_stats.n_ext_unresolved).  h_ctr holds the last launch's counters.
        auto sr_on_demand = [&](ExtArgs xa) -> sh_status {
            uint32_t n_un = c->h_ctr->ext_n_unres, cap = c->ext_reg_cap;
            n_sr_ondemand += n_un;
            int cur = 0;
            while (n_un > 0) {
                uint8_t *buf = nullptr; unsigned long long per = 0; uint32_t waves = 0;
                if (cap < (1u << 24)) {
                    cap = (uint32_t)std::min<uint64_t>((uint64_t)cap * 4, 1u << 24);
                    per = align_scratch_layout(c->max_read_len, cap, nullptr, nullptr, nullptr);
                    for (waves = std::min<uint32_t>(n_un, 8); waves > 0 && hipMalloc(&buf, per * waves) != hipSuccess; waves >>= 1) { buf = nullptr; (void)hipGetLastError(); }
                }
                const uint32_t z4[4] = {0, 0, n_un, 0};      // ext_n_unres, ext_ticket_unres, ext_n_unres_in
                SH_HIP(hipMemcpyAsync(&c->d_ctr->ext_n_unres, z4, 16, hipMemcpyHostToDevice, s));
                xa.list = c->d_ext_unres[cur]; xa.n_list = &c->d_ctr->ext_n_unres_in; xa.ticket = &c->d_ctr->ext_ticket_unres;
                if (buf) { xa.scratch = buf; xa.scratch_per_wave = per; xa.reg_cap = cap; xa.unres_list = c->d_ext_unres[cur ^ 1]; xa.n_unres = &c->d_ctr->ext_n_unres; }
                else {      // no memory to be had: counted, chain-level answer (on the context's own working memory: the last round's is freed)
                    xa.unres_list = nullptr; xa.n_unres = nullptr; waves = std::min<uint32_t>(c->ext_waves, n_un);
                    xa.scratch = c->d_ext_scratch; xa.scratch_per_wave = c->ext_scratch_per_wave; xa.reg_cap = c->ext_reg_cap;
                }
                hipLaunchKernelGGL(k_regs_align, dim3(waves), dim3(64), 0, s, xa);
                SH_HIP(hipMemcpyAsync(c->h_ctr, c->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, s));
                SH_HIP(hipStreamSynchronize(s));
                if (buf) hipFree(buf);
                SH_HIP(hipGetLastError());
                if (c->h_ctr->ext_overflow == 1) return SH_SPLIT;
                SH_CHECK(c->h_ctr->ext_overflow == 0, SH_ERR_OOM, "extension stage: a read exceeds the per-wave working memory (code %u)", c->h_ctr->ext_overflow);
                if (!buf) break;
                n_un = c->h_ctr->ext_n_unres; cur ^= 1;
            }
            return SH_OK;
        };
        SH_HIP(hipEventRecord(c->ev_ext[0], s));
        hipLaunchKernelGGL(k_ext_list, dim3((uint32_t)((n_reads + 255) / 256)), dim3(256), 0, s, x);
        if (d_trace != nullptr) {
            // trace mode: every read that handed over a chain goes through mm_gen_regs .. mm_filter_regs (one wave per read)
            hipLaunchKernelGGL(k_regs_align, dim3(c->ext_waves), dim3(64), 0, s, x);
        } else {
            // flag-only: the lists hold each read's candidates for regs[0]; the top chain settles nearly every read (one lane each)
            hipLaunchKernelGGL(k_ext_top, dim3(grid * 4), dim3(64), 0, s, x);
        }
        SH_HIP(hipMemcpyAsync(c->h_ctr, c->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, s));
        SH_HIP(hipStreamSynchronize(s));
        SH_HIP(hipGetLastError());
        if (c->h_ctr->ext_overflow == 1) return SH_SPLIT;       // hand-over buffers full: the caller cuts the chunk in two
        SH_CHECK(c->h_ctr->ext_overflow == 0, SH_ERR_OOM, "extension stage: a read exceeds the per-wave working memory (code %u: 2 chains, 3 primaries, 4 read length, 5 window, 6 regions)", c->h_ctr->ext_overflow);
        ext_list = c->h_ctr->ext_n_list;
        if (d_trace != nullptr && c->h_ctr->ext_n_unres > 0) { const uint32_t keep = ext_list; sh_status st = sr_on_demand(x); if (st != SH_OK) return st; ext_list = keep; }
        const uint32_t n_redo = c->h_ctr->ext_n_redo;
        if (d_trace == nullptr) { c->dbg_ptr[1] = c->d_ext_redo; c->dbg_n[1] = n_redo; }
        if (getenv("SCRUBBY_HIP_DBG")) fprintf(stderr, "[dbg] extension stage: %u reads handed over chains, %u not settled by their top chain (tie %u, missing %u, short stretch %u, z-drop %u)\n", ext_list, n_redo, c->h_ctr->ext_reason[1], c->h_ctr->ext_reason[2], c->h_ctr->ext_reason[3], c->h_ctr->ext_reason[4]);
        uint32_t n_redo2 = 0;
        if (d_trace == nullptr && n_redo > 0) {
            // second pass: regs[0] of the reads its max stretch could not vouch for goes through mm_align1 (one wave per read)
            ExtArgs x1 = x;
            x1.list = c->d_ext_redo; x1.n_list = &c->d_ctr->ext_n_redo; x1.ticket = &c->d_ctr->ext_ticket2; x1.top_only = 1; x1.redo2 = c->d_ext_list;     // the first list is spent
            x1.scratch_per_wave = c->ext_scratch_per_wave_top; x1.reg_cap = 64;
            hipLaunchKernelGGL(k_regs_align_top, dim3(c->ext_waves_top), dim3(64), 0, s, x1);
            SH_HIP(hipMemcpyAsync(c->h_ctr, c->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, s));
            SH_HIP(hipStreamSynchronize(s));
            SH_HIP(hipGetLastError());
            SH_CHECK(c->h_ctr->ext_overflow == 0, SH_ERR_OOM, "extension stage: a read exceeds the per-wave working memory (code %u)", c->h_ctr->ext_overflow);
            n_redo2 = c->h_ctr->ext_n_redo2;
            c->dbg_ptr[2] = c->d_ext_list; c->dbg_n[2] = n_redo2;
        }
        uint32_t n_fallback = 0; double ms_fallback = 0;
        if (n_redo2 > 0) {
            n_fallback = n_redo2;
            const auto t_fb = std::chrono::steady_clock::now();
            // third pass, reads whose top chain does not survive: all their chains, then the complete procedure.  They take the repeat path
            // again (wave-parallel expansion, sort classes, cluster DP) with nothing filtered; the legacy lane-per-read kernel that used to
            // carry this pass spent 130 ms on 507 reads of the sr-div workload.  The hand-over buffers start over.
            Counters z = *c->h_ctr;
            memset(z.ext_n_recs, 0, sizeof(z.ext_n_recs)); memset(z.ext_n_anch, 0, sizeof(z.ext_n_anch));
            z.n_defer = 0; z.arena_cursor = 0; z.ext_n_list2 = n_redo2; z.ext_ticket3 = 0;
            z.n_big[0] = n_redo2; z.n_big[1] = 0; z.n_big_defer[0] = z.n_big_defer[1] = 0;
            SH_HIP(hipMemcpyAsync(c->d_ctr, &z, sizeof(Counters), hipMemcpyHostToDevice, s));
            ExtArgs x2 = x;
            x2.list = c->d_ext_list; x2.n_list = &c->d_ctr->ext_n_list2; x2.ticket = &c->d_ctr->ext_ticket3;
            hipLaunchKernelGGL(k_ext_reset, dim3(64), dim3(256), 0, s, x2);
            SH_HIP(hipMemcpyAsync(c->d_big[0][0], c->d_ext_list, (size_t)n_redo2 * 4, hipMemcpyDeviceToDevice, s));
            K3Args kf = k;
            kf.sink.best = nullptr; kf.sink.tie = nullptr; kf.t_mode = 0; kf.flag_only = 0; kf.quiet = 1;
            K2Args rb = b;
            rb.sink.best = nullptr; rb.sink.tie = nullptr; rb.quiet = 1;
            int f0 = 0, f1 = c->dbg_ptr[0] == c->d_big[1][0] ? 1 : 0;      // not over the first pass's list of re-chained reads (sh_ctx_debug_list)
            { sh_status st = run_repeat_path(kf, rb, f0, f1, resk_done); if (st != SH_OK) return st; }
            // reads the repeat path handed to the legacy kernel and that found no room there
            rb.work_defer = c->d_work_defer;
            uint32_t left = c->h_ctr->n_defer;
            int rounds2 = 0;
            while (left > 0) {
                SH_CHECK(++rounds2 < 64, SH_ERR_OOM, "re-sketch arena too small for the extension stage's last pass; set SCRUBBY_HIP_ARENA_MB");
                std::swap(c->d_work_defer, c->d_work_defer2);
                Counters z2 = *c->h_ctr;
                z2.n_defer = 0; z2.arena_cursor = 0; z2.n_resketch = left;
                SH_HIP(hipMemcpyAsync(c->d_ctr, &z2, sizeof(Counters), hipMemcpyHostToDevice, s));
                rb.work = c->d_work_defer2; rb.work_count = &c->d_ctr->n_resketch; rb.work_begin = 0; rb.work_defer = c->d_work_defer;
                hipLaunchKernelGGL(k_chain_large, dim3(grid), dim3(64), 0, s, rb);
                SH_HIP(hipMemcpyAsync(c->h_ctr, c->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, s));
                SH_HIP(hipStreamSynchronize(s));
                const uint32_t nd = c->h_ctr->n_defer;
                SH_CHECK(nd < left, SH_ERR_OOM, "re-sketch arena too small for the extension stage's last pass; set SCRUBBY_HIP_ARENA_MB");
                left = nd;
            }
            if (c->h_ctr->ext_overflow == 1) return SH_SPLIT;
            hipLaunchKernelGGL(k_regs_align, dim3(c->ext_waves), dim3(64), 0, s, x2);
            SH_HIP(hipMemcpyAsync(c->h_ctr, c->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, s));
            SH_HIP(hipStreamSynchronize(s));
            SH_HIP(hipGetLastError());
            if (c->h_ctr->ext_overflow == 1) return SH_SPLIT;
            SH_CHECK(c->h_ctr->ext_overflow == 0, SH_ERR_OOM, "extension stage: a read exceeds the per-wave working memory (code %u)", c->h_ctr->ext_overflow);
            if (c->h_ctr->ext_n_unres > 0) { sh_status st = sr_on_demand(x2); if (st != SH_OK) return st; }
            ms_fallback = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_fb).count();
        }
        if (stats) { stats->n_ext_fallback += n_fallback; stats->ms_ext_fallback += ms_fallback; stats->n_ext_unresolved += c->h_ctr->lext_unresolved; stats->n_ext_ondemand += n_sr_ondemand; }
        if (c->h_ctr->lext_unresolved) {
            static bool warned_sr = false;
            if (!warned_sr) { warned_sr = true; fprintf(stderr, "[scrubby-hip] WARNING: %u read(s) outgrew the extension stage's working memory (e.g. read %u of its batch, code %u: 2 chains, 3 primaries, 6 regions); they keep their chain-level answer (mapped)\n", c->h_ctr->lext_unresolved, c->h_ctr->lext_err_read, c->h_ctr->lext_err_code); }
        }
        SH_HIP(hipEventRecord(c->ev_ext[1], s));
        SH_HIP(hipEventSynchronize(c->ev_ext[1]));
        ext_regions = c->h_ctr->ext_regions; ext_dropped = c->h_ctr->ext_dropped;
        if (d_trace == nullptr) ext_list = n_redo;      // reads that needed base-level alignment
        hipEventElapsedTime(&ms_ext, c->ev_ext[0], c->ev_ext[1]);
    }
    SH_HIP(hipEventRecord(c->ev[4], s));
    SH_HIP(hipEventSynchronize(c->ev[4]));
    if (stats) {
        float t01 = 0, t12 = 0, t23 = 0, t04 = 0;
        hipEventElapsedTime(&t01, c->ev[0], c->ev[1]); hipEventElapsedTime(&t12, c->ev[1], c->ev[2]);
        hipEventElapsedTime(&t23, c->ev[1], c->ev[3]); hipEventElapsedTime(&t04, c->ev[0], c->ev[4]);
        uint64_t sum_host = 0, sum_mini = 0, sum_anchors = 0, sum_clusters = 0, sum_pair = 0, sum_lemma = 0, sum_pf = 0, sum_pfd = 0, sum_top = 0;
        for (int i = 0; i < 64; ++i) sum_lemma += c->h_ctr->sh_lemma[i];
        stats->n_ext_shortcut += sum_lemma;
        for (int i = 0; i < 64; ++i) { sum_host += c->h_ctr->sh_host[i]; sum_mini += c->h_ctr->sh_mini[i]; sum_anchors += c->h_ctr->sh_anchors[i]; sum_clusters += c->h_ctr->sh_clusters[i]; sum_pair += c->h_ctr->sh_pair[i]; sum_pf += c->h_ctr->sh_pf_reads[i]; sum_pfd += c->h_ctr->sh_pf_dirty[i]; sum_top += c->h_ctr->sh_top[i]; }
        stats->n_reads += n_reads; stats->n_bases += n_bases;
        stats->n_host += sum_host - ext_dropped - c->h_ctr->lr_fb_had;
        stats->n_ext_reads += ext_list; stats->n_ext_regions += ext_regions; stats->n_ext_dropped += ext_dropped; stats->ms_ext += ms_ext; const uint32_t n_big0 = snap.n_big_total ? snap.n_big_total : snap.n_big[0];
        stats->n_no_seed += n_reads - snap.n_small - n_big0 - snap.n_resketch;
        uint64_t nl = (uint64_t)snap.n_resketch + n_big0;
        stats->n_chain_large += nl; stats->n_chain_small += snap.n_small;
        stats->n_minimizers += sum_mini;
        stats->n_anchors += sum_anchors; stats->n_clusters += sum_clusters; stats->n_resketch += resk_done; stats->n_pair_decided += sum_pair;
        stats->n_dp_parallel += sum_pf; stats->n_dp_dirty += sum_pfd; stats->n_top_settled += sum_top;
        stats->ms_sketch_probe += t01; stats->ms_chain_small += t12; stats->ms_chain_large += t23; stats->ms_total += t04;
    }
    return SH_OK;
}

extern "C" sh_status sh_classify_device(sh_ctx *c, const uint8_t *d_bases, const uint64_t *d_offsets, uint64_t n_reads,
                                        uint64_t n_bases, uint8_t *d_flags, sh_trace *d_trace, void *stream, sh_stats *stats)
{
    SH_CHECK(c && d_offsets && d_flags && (d_bases || n_bases == 0), SH_ERR_BAD_ARG, "sh_classify_device: null argument");
    SH_HIP(hipSetDevice(c->idx->device));
    hipStream_t s = (hipStream_t)stream;
    if (stats) memset(stats, 0, sizeof(*stats));
    if (c->ext_long) {      // the stage's per-read path record of this call
        if (n_reads > c->lkind_cap) {
            if (c->d_lkind) hipFree(c->d_lkind);
            c->d_lkind = nullptr; c->lkind_cap = 0;
            SH_HIP(hipMalloc(&c->d_lkind, n_reads + n_reads / 8 + 64));
            c->lkind_cap = n_reads + n_reads / 8 + 64;
        }
        c->lkind_n = n_reads;
        if (n_reads) SH_HIP(hipMemsetAsync(c->d_lkind, 0, n_reads, s));
    }
    std::vector<std::pair<uint64_t, uint64_t>> todo;      // (first read, count), processed back to front
    for (uint64_t r0 = n_reads; r0 > 0;) { const uint64_t n = (r0 - 1) % c->max_reads + 1; r0 -= n; todo.push_back({r0, n}); }
    while (!todo.empty()) {
        const auto [r0, n] = todo.back();
        todo.pop_back();
        c->lkind_r0 = r0;
        sh_status st = classify_chunk(c, d_bases, d_offsets + r0, n, n_bases, d_flags + r0, d_trace ? d_trace + r0 : nullptr, s, stats);
        if (st == SH_SPLIT) {      // the chains of this chunk did not fit the hand-over buffers of the extension stage
            SH_CHECK(n > 1, SH_ERR_OOM, "extension stage: the chains of a single read exceed the hand-over buffers; raise SCRUBBY_HIP_EXT_MB");
            todo.push_back({r0 + n / 2, n - n / 2}); todo.push_back({r0, n / 2});
            continue;
        }
        if (st != SH_OK) return st;
    }
    return SH_OK;
}
