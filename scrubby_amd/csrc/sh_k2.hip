// sh_k2.hip — Kraken2-style taxid classifier on MI355X (BASELINE configs[4]).
//
// Replaces the external `kraken2` process of Cleaner::run_kraken (/root/reference/src/cleaner.rs:288-330).
// HBM layout: the compact hash table is one array of 32-bit cells (high 32 - value_bits bits = truncated hash,
// low value_bits bits = internal taxid, value 0 = empty), the taxonomy two u32 arrays (parent, external id).
//
// k_k2_classify: one lane per read / pair, 64 units per wave.
//   scan     rolling forward / reverse-complement l-mers, canonical, spaced-seed mask, toggle; the window minimum over
//            the k-l+1 most recent candidates lives in registers (W = 5 for k = 35, l = 31);
//   runs     consecutive k-mers with the same minimizer form one run = one table probe worth `len` k-mer counts;
//            runs queue up per lane in LDS and the whole wave drains its queues together, four probes in flight
//            per lane, so the random 4-B gathers overlap instead of stalling the scan one at a time;
//   resolve  per-lane hit list in LDS (<= HCAP distinct taxa; the rare unit beyond that is redone by the same kernel
//            with its list in HBM), ResolveTree over BFS-ordered parent links.
// The bound is the HBM gather rate: one 32-B sector per probe, ~40 probes per 150-bp read.
#include "sh_common.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <string>
#include <vector>

#define K2_QCAP 8           // pending runs per lane
#define K2_HCAP 8           // distinct taxa per unit kept in LDS (10 KiB of LDS per wave with the queue: 16 waves per CU)
#define K2_BIG_CAP 4096     // ... in HBM for the overflow pass

struct sh_k2_db {
    int device = 0;
    sh_k2_opts opts{};
    uint64_t capacity = 0, size = 0;
    int32_t key_bits = 0, value_bits = 0;
    uint32_t *d_cells = nullptr, *d_parent = nullptr, *d_ext = nullptr;
    unsigned long long *d_ctr = nullptr;         // [0] new cells claimed, [1] table full
    std::vector<sh_k2_taxnode> nodes;
    std::string names, ranks;
    // opts.k2d fields carried through save/open
    int32_t dna_db = 1, revcom_version = 1, db_version = 0, db_type = 0;
};

// ---- device helpers ----------------------------------------------------------------------------------------------
__host__ __device__ static inline uint64_t k2_fmix64(uint64_t k)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL;
    k ^= k >> 33;
    return k;
}

__device__ static inline uint32_t k2_lca(const uint32_t *__restrict__ parent, uint32_t a, uint32_t b)
{
    if (!a || !b) return a ? a : b;
    while (a != b) { if (a > b) a = parent[a]; else b = parent[b]; }
    return a;
}
__device__ static inline bool k2_is_ancestor(const uint32_t *__restrict__ parent, uint32_t a, uint32_t b)
{
    if (!a || !b) return false;
    while (b > a) b = parent[b];
    return a == b;
}

struct K2Table { const uint32_t *cells; uint64_t capacity; int32_t value_bits; double inv_capacity; };

// hc % capacity without the 64-bit division sequence: the double-precision quotient estimate is off by at most one for
// capacity >= 2^20 (q < 2^44, relative error 2^-52), which one conditional add / subtract repairs
__device__ static inline uint64_t k2_mod(uint64_t hc, uint64_t capacity, double inv_capacity)
{
    if (capacity < (1ull << 20)) return hc % capacity;
    const uint64_t q = (uint64_t)((double)hc * inv_capacity);
    int64_t r = (int64_t)(hc - q * capacity);
    if (r < 0) r += (int64_t)capacity;
    else if (r >= (int64_t)capacity) r -= (int64_t)capacity;
    return (uint64_t)r;
}

// rolling scanner state of one lane; the window holds k - l + 1 candidates (<= W; the instantiations are W = 1, 5 and,
// for every other (k, l), 16 with the live part given at run time)
template <int W>
struct K2Scan {
    uint64_t fw, rc, c[W];
    int32_t loaded;
    __device__ inline void reset()
    {
        fw = rc = 0; loaded = 0;
#pragma unroll
        for (int i = 0; i < W; ++i) c[i] = ~0ull;
    }
    // consumes one base code (0..3, > 3 = ambiguous); pos = characters consumed so far in this fragment (after this one).
    // Returns 0 nothing to report (no full k-mer yet), 1 ambiguous k-mer, 2 minimizer in `m`.
    __device__ inline int step(uint32_t code, int32_t pos, int32_t k, int32_t l, uint64_t lmask, uint64_t spaced, uint64_t toggle, uint64_t &m)
    {
        return step_w(code, pos, k, l, lmask, spaced, toggle, m, k - l + 1);
    }
    __device__ inline int step_w(uint32_t code, int32_t pos, int32_t k, int32_t l, uint64_t lmask, uint64_t spaced, uint64_t toggle, uint64_t &m, int32_t wlim)
    {   // branch-free: 64 lanes scan 64 different reads
        const bool amb = code > 3;
        fw = amb ? 0ull : ((fw << 2) | code) & lmask;
        rc = amb ? 0ull : (rc >> 2) | ((uint64_t)(3u - code) << (2 * (l - 1)));
        loaded = amb ? 0 : (loaded < l ? loaded + 1 : l);
        const bool full = loaded == l;
        uint64_t canon = fw < rc ? fw : rc;
        if (spaced) canon &= spaced;
        const uint64_t cand = canon ^ toggle;
#pragma unroll
        for (int i = W - 1; i > 0; --i) c[i] = amb ? ~0ull : (full ? c[i - 1] : c[i]);
        c[0] = amb ? ~0ull : (full ? cand : c[0]);
        uint64_t mn = c[0];
#pragma unroll
        for (int i = 1; i < W; ++i) mn = (i < wlim && c[i] < mn) ? c[i] : mn;
        m = mn ^ toggle;
        return pos >= k ? (full ? 2 : 1) : 0;
    }
};

struct K2Args {
    const uint8_t *bases; const uint64_t *offsets; uint64_t n_units; int32_t paired;
    K2Table T; const uint32_t *parent, *ext; uint32_t n_nodes;
    int32_t k, l; uint64_t spaced, toggle, min_hash; int32_t min_hit_groups; double confidence;
    sh_k2_result *out;
    uint32_t *over_list; unsigned long long *ctr;      // ctr: [0] n_over, [1..] sharded stats
    const uint32_t *unit_list; uint32_t n_list;        // BIG pass: units to redo
    uint32_t *big_tax, *big_cnt;                       // BIG pass: K2_BIG_CAP entries per listed unit
};
// ctr layout
#define K2C_OVER 0
#define K2C_PROBES 8
#define K2C_KMERS 72
#define K2C_CLASSIFIED 136
#define K2C_WORDS 200

// hit list of one lane: entry j at [j * STRIDE]
template <int STRIDE, int CAP>
struct K2Hits {
    uint32_t *tax, *cnt; int32_t n; bool over;
    __device__ inline void add(uint32_t t, uint32_t c)
    {
        int32_t j = 0;
        while (j < n && tax[j * STRIDE] != t) ++j;
        if (j < n) { cnt[j * STRIDE] += c; return; }
        if (n == CAP) { over = true; return; }
        tax[n * STRIDE] = t; cnt[n * STRIDE] = c; ++n;
    }
};

template <int STRIDE, int CAP>
__device__ static inline uint32_t k2_resolve(const K2Hits<STRIDE, CAP> &H, const uint32_t *__restrict__ parent, uint32_t total_kmers, double confidence)
{
    uint32_t max_taxon = 0, max_score = 0;
    const uint32_t required = (uint32_t)ceil(confidence * (double)total_kmers);
    for (int32_t i = 0; i < H.n; ++i) {
        const uint32_t ti = H.tax[i * STRIDE];
        uint32_t score = 0;
        for (int32_t j = 0; j < H.n; ++j) if (k2_is_ancestor(parent, H.tax[j * STRIDE], ti)) score += H.cnt[j * STRIDE];
        if (score > max_score) { max_score = score; max_taxon = ti; }
        else if (score == max_score) max_taxon = k2_lca(parent, max_taxon, ti);
    }
    max_score = 0;
    for (int32_t i = 0; i < H.n; ++i) if (H.tax[i * STRIDE] == max_taxon) max_score = H.cnt[i * STRIDE];
    while (max_taxon && max_score < required) {
        max_score = 0;
        for (int32_t i = 0; i < H.n; ++i) if (k2_is_ancestor(parent, max_taxon, H.tax[i * STRIDE])) max_score += H.cnt[i * STRIDE];
        if (max_score >= required) return max_taxon;
        max_taxon = parent[max_taxon];
    }
    return max_taxon;
}

__device__ static inline uint32_t k2_finish_probe(const K2Table &T, uint64_t idx, uint32_t cell, uint32_t compacted)
{
    const uint32_t vmask = (1u << T.value_bits) - 1;
    const uint64_t first = idx;
    for (;;) {
        if (!(cell & vmask)) return 0;
        if ((cell >> T.value_bits) == compacted) return cell & vmask;
        idx = idx + 1 == T.capacity ? 0 : idx + 1;
        if (idx == first) return 0;
        cell = T.cells[idx];
    }
}

// Linear probing walks consecutive cells, so a probe reads whole 32-B groups of 8 cells (one HBM sector) instead of one
// cell per dependent load: at load 0.7 a miss inspects ~6 cells = 1-2 groups, and the slowest lane of a wave (which every
// other lane waits for) needs 3-4 steps instead of 30.
struct K2Group { uint4 lo, hi; };
__device__ static inline K2Group k2_load_group(const uint32_t *cells, uint64_t g)
{
    const uint4 *p = (const uint4 *)(cells + 8 * g);
    return K2Group{p[0], p[1]};
}
// first terminating cell (empty, or same truncated key) at or after `start`; returns false if the group has none
__device__ static inline bool k2_scan_group(const K2Group &G, uint32_t start, uint32_t vmask, int32_t vb, uint32_t comp, uint32_t &taxon)
{
    const uint32_t c[8] = {G.lo.x, G.lo.y, G.lo.z, G.lo.w, G.hi.x, G.hi.y, G.hi.z, G.hi.w};
    uint32_t term = 0, match = 0;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const bool e = !(c[t] & vmask), mm = (c[t] >> vb) == comp;
        term |= (uint32_t)(e || mm) << t; match |= (uint32_t)(mm && !e) << t;
    }
    term &= 0xffu << start;
    if (!term) return false;
    const int t0 = __ffs((int)term) - 1;
    uint32_t v = 0;
#pragma unroll
    for (int t = 0; t < 8; ++t) v = t == t0 ? c[t] : v;
    taxon = ((match >> t0) & 1u) ? v & vmask : 0u;
    return true;
}
// the rest of a probe whose first group did not decide it
__device__ static inline uint32_t k2_probe_rest(const K2Table &T, uint64_t g, uint32_t comp)
{
    const uint32_t vmask = (1u << T.value_bits) - 1;
    const uint64_t n_full = T.capacity / 8;          // groups [0, n_full) lie entirely inside the table
    for (uint64_t step = 0; step <= n_full + 1; ++step) {
        ++g;
        if (g >= n_full) {                            // the ragged tail group and the wrap to cell 0: cell by cell
            uint64_t idx = g * 8 < T.capacity ? g * 8 : 0;
            for (; idx < T.capacity && idx >= n_full * 8; ++idx) {
                const uint32_t c = T.cells[idx];
                if (!(c & vmask)) return 0;
                if ((c >> T.value_bits) == comp) return c & vmask;
            }
            g = 0;
            if (n_full == 0) continue;
            const K2Group G0 = k2_load_group(T.cells, 0);
            uint32_t taxon;
            if (k2_scan_group(G0, 0, vmask, T.value_bits, comp, taxon)) return taxon;
            continue;
        }
        const K2Group G = k2_load_group(T.cells, g);
        uint32_t taxon;
        if (k2_scan_group(G, 0, vmask, T.value_bits, comp, taxon)) return taxon;
    }
    return 0;
}

template <int W, bool BIG>
__global__ __launch_bounds__(64) void k_k2_classify(K2Args a)
{
    __shared__ uint64_t s_qmin[(K2_QCAP + 1) * 64];       // slot n_pend is written unconditionally, so one spare
    __shared__ uint32_t s_qlen[(K2_QCAP + 1) * 64];
    __shared__ uint32_t s_htax[BIG ? 1 : K2_HCAP * 64], s_hcnt[BIG ? 1 : K2_HCAP * 64];
    const uint32_t lane = threadIdx.x;
    const uint64_t lmask = a.l < 32 ? ((1ULL << (2 * a.l)) - 1) : ~0ULL;
    const int32_t wlim = a.k - a.l + 1;
    const uint64_t n_work = BIG ? a.n_list : a.n_units;
    unsigned long long probes_thr = 0, kmers_thr = 0; uint32_t class_thr = 0;
    for (uint64_t base = (uint64_t)blockIdx.x * 64; base < n_work; base += (uint64_t)gridDim.x * 64) {
        const uint64_t wi = base + lane;
        const bool active = wi < n_work;
        const uint64_t u = active ? (BIG ? (uint64_t)a.unit_list[wi] : wi) : 0;
        K2Hits<BIG ? 1 : 64, BIG ? K2_BIG_CAP : K2_HCAP> H;
        if (BIG) { H.tax = a.big_tax + (active ? wi : 0) * K2_BIG_CAP; H.cnt = a.big_cnt + (active ? wi : 0) * K2_BIG_CAP; }
        else { H.tax = s_htax + lane; H.cnt = s_hcnt + lane; }
        H.n = 0; H.over = false;
        uint32_t total = 0, groups = 0, n_pend = 0, probes_unit = 0;
        const int n_frag = a.paired ? 2 : 1;
        auto drain = [&]() {
            // pass 1: every lane gathers the home group of each pending run, eight 16-B loads in flight; the outcome goes
            // back into the queue slot: the taxon, or (undecided | truncated key | group) for the rare longer chain
            const uint32_t vmask = (1u << a.T.value_bits) - 1;
            const uint64_t n_full = a.T.capacity / 8;
            uint32_t undecided = 0, skipped = 0;         // per-lane bit e: entry e needs the long-chain code / was not looked up
            for (uint32_t e0 = 0; e0 < K2_QCAP; e0 += 4) {
                if (__ballot(e0 < n_pend) == 0) break;
                uint64_t idx[4]; uint32_t comp[4]; bool go[4]; K2Group G[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    go[q] = e0 + q < n_pend;
                    idx[q] = 0; comp[q] = 0;
                    if (go[q]) {
                        const uint64_t hc = k2_fmix64(s_qmin[(e0 + q) * 64 + lane]);
                        if (a.min_hash && hc < a.min_hash) { go[q] = false; skipped |= 1u << (e0 + q); }      // down-sampled database: not looked up
                        else { comp[q] = (uint32_t)(hc >> (32 + a.T.value_bits)); idx[q] = k2_mod(hc, a.T.capacity, a.T.inv_capacity); }
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bool whole = go[q] && (idx[q] >> 3) < n_full;
                    G[q] = whole ? k2_load_group(a.T.cells, idx[q] >> 3) : K2Group{make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (!go[q]) continue;
                    uint32_t taxon = 0;
                    const bool whole = (idx[q] >> 3) < n_full;
                    const bool done = whole && k2_scan_group(G[q], (uint32_t)idx[q] & 7u, vmask, a.T.value_bits, comp[q], taxon);
                    if (done) s_qmin[(e0 + q) * 64 + lane] = taxon;      // else the slot keeps the minimizer for pass 2
                    else undecided |= 1u << (e0 + q);
                }
            }
            // pass 2: one copy of the long-chain code and of the hit-list update
#pragma nounroll
            for (uint32_t e = 0; e < K2_QCAP; ++e) {
                if (__ballot(e < n_pend) == 0) break;
                if (e < n_pend) {
                    const uint64_t v = s_qmin[e * 64 + lane];
                    if (!((skipped >> e) & 1u)) {
                        ++probes_unit;
                        uint32_t taxon = (uint32_t)v;
                        if ((undecided >> e) & 1u) {        // the chain leaves the home group (or starts in the ragged tail group)
                            const uint64_t hc = k2_fmix64(v);
                            const uint32_t comp = (uint32_t)(hc >> (32 + a.T.value_bits));
                            const uint64_t idx = k2_mod(hc, a.T.capacity, a.T.inv_capacity);
                            taxon = (idx >> 3) < n_full ? k2_probe_rest(a.T, idx >> 3, comp) : k2_finish_probe(a.T, idx, a.T.cells[idx], comp);
                        }
                        if (taxon) { ++groups; H.add(taxon, s_qlen[e * 64 + lane]); }
                    }
                }
            }
            n_pend = 0;
        };
        // Both fragments run through ONE character loop (one call site of drain): fragment f covers the chunk range
        // [f * n_chunks_max, (f + 1) * n_chunks_max) of the wave, each lane idling past its own read's end.
        uint64_t o_beg[2] = {0, 0}; int32_t len[2] = {0, 0};
        int32_t max_len = 0;
        for (int f = 0; f < n_frag; ++f) {
            const uint64_t rec = a.paired ? 2 * u + (uint64_t)f : u;
            o_beg[f] = active ? a.offsets[rec] : 0;
            len[f] = active ? (int32_t)(a.offsets[rec + 1] - o_beg[f]) : 0;
            max_len = len[f] > max_len ? len[f] : max_len;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { int32_t t = __shfl_xor(max_len, o); max_len = t > max_len ? t : max_len; }
        const int32_t n_chunks = (max_len + 7) / 8;
        K2Scan<W> S; S.reset();
        uint64_t last_min = ~0ull; uint32_t run = 0;
        int32_t my_len = 0, off8 = 0; uint32_t sh = 0;
        const uint64_t *wp = nullptr;
        uint64_t w_cur = 0, w_next = 0;
        // ONE loop over the characters of both fragments with ONE call site of drain (the probe code is large): step s is
        // character (s & 7) of chunk (s >> 3); the step after the last one flushes the final run and empties the queues.
        const int32_t n_steps = n_chunks * n_frag * 8;
        uint64_t w = 0;
#pragma nounroll
        for (int32_t s = 0;; ++s) {
            const bool end = s == n_steps;
            const int32_t cc = s >> 3, c = cc < n_chunks ? cc : cc - n_chunks;
            if ((s & 7) == 0) {
                if (c == 0 || end) {       // (wave-uniform) fragment boundary: flush the last run, restart the scanner
                    s_qmin[n_pend * 64 + lane] = last_min; s_qlen[n_pend * 64 + lane] = run;
                    n_pend += run != 0;
                    S.reset(); last_min = ~0ull; run = 0;
                    if (!end) {
                        const int f = cc < n_chunks ? 0 : 1;
                        my_len = len[f];
                        const uintptr_t pa = (uintptr_t)(a.bases + o_beg[f]);
                        wp = (const uint64_t *)(pa & ~(uintptr_t)7); off8 = (int32_t)(pa & 7); sh = (uint32_t)off8 * 8;
                        // only the aligned words that overlap the read are ever loaded
                        w_cur = my_len > 0 ? wp[0] : 0;
                        w_next = my_len + off8 > 8 ? wp[1] : 0;
                    }
                }
                if (!end) {
                    w = sh ? (w_cur >> sh) | (w_next << (64 - sh)) : w_cur;
                    w_cur = w_next;
                    w_next = (c + 2) * 8 < my_len + off8 ? wp[c + 2] : 0;      // two words ahead: the load has a whole chunk to land
                }
            }
            if (__ballot(n_pend >= (end ? 1u : (uint32_t)K2_QCAP - 1)) != 0) drain();       // wave-uniform: every lane is here
            if (end) break;
            const int32_t i = c * 8 + (s & 7);
            uint64_t m;
            int ev = S.step_w(sh_nt4((uint32_t)w & 0xffu), i + 1, a.k, a.l, lmask, a.spaced, a.toggle, m, wlim);
            w >>= 8;
            ev = i < my_len ? ev : 0;
            total += ev != 0;
            const bool fresh = ev == 2 && m != last_min;
            s_qmin[n_pend * 64 + lane] = last_min; s_qlen[n_pend * 64 + lane] = run;      // kept only if the run just ended
            n_pend += fresh && run != 0;
            run = fresh ? 1u : run + (ev == 2);
            last_min = fresh ? m : last_min;
        }
        if (active) {
            kmers_thr += total;
            if (BIG || !H.over) probes_thr += probes_unit;       // a unit redone by the overflow pass is counted there
            if (!BIG && H.over) {
                const uint32_t oi = (uint32_t)atomicAdd(&a.ctr[K2C_OVER], 1ull);
                a.over_list[oi] = (uint32_t)u;
            } else {
                uint32_t call = k2_resolve(H, a.parent, total, a.confidence);
                if (call && groups < (uint32_t)a.min_hit_groups) call = 0;
                sh_k2_result r{call ? a.ext[call] : 0u, call, total, groups};
                a.out[u] = r;
                class_thr += call != 0;
            }
        }
    }
    // statistics: one atomic per wave and counter, 64-way sharded
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        probes_thr += (unsigned long long)__shfl_xor((long long)probes_thr, o);
        kmers_thr += (unsigned long long)__shfl_xor((long long)kmers_thr, o);
        class_thr += (uint32_t)__shfl_xor((int)class_thr, o);
    }
    if (lane == 0) {
        const uint32_t sh = blockIdx.x & 63;
        if (probes_thr) atomicAdd(&a.ctr[K2C_PROBES + sh], probes_thr);
        if (kmers_thr && !BIG) atomicAdd(&a.ctr[K2C_KMERS + sh], kmers_thr);
        if (class_thr) atomicAdd(&a.ctr[K2C_CLASSIFIED + sh], (unsigned long long)class_thr);
    }
}

// ---- table construction ----------------------------------------------------------------------------------------------
struct K2Build { uint32_t *cells; uint64_t capacity; int32_t value_bits; const uint32_t *parent; unsigned long long *ctr; };

__device__ static inline void k2_insert(const K2Build &B, uint64_t key, uint32_t value)
{
    const uint64_t hc = k2_fmix64(key);
    const uint32_t compacted = (uint32_t)(hc >> (32 + B.value_bits));
    const uint32_t vmask = (1u << B.value_bits) - 1;
    uint64_t idx = hc % B.capacity;
    const uint64_t first = idx;
    for (;;) {
        uint32_t c = B.cells[idx];
        if (!(c & vmask)) {
            const uint32_t prev = atomicCAS(&B.cells[idx], c, compacted << B.value_bits | value);
            if (prev == c) { atomicAdd(&B.ctr[0], 1ull); return; }
            c = prev;                       // somebody else claimed the cell: look at what is there now
        }
        if ((c >> B.value_bits) == compacted) {
            for (;;) {                      // same (truncated) key: keep the LCA of the two taxa
                const uint32_t nv = k2_lca(B.parent, c & vmask, value);
                if (nv == (c & vmask)) return;
                const uint32_t prev = atomicCAS(&B.cells[idx], c, compacted << B.value_bits | nv);
                if (prev == c) return;
                c = prev;
            }
        }
        idx = idx + 1 == B.capacity ? 0 : idx + 1;
        if (idx == first) { atomicAdd(&B.ctr[1], 1ull); return; }
    }
}

__global__ void k_k2_insert(K2Build B, const uint64_t *keys, const uint32_t *taxa, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) k2_insert(B, keys[i], taxa[i]);
}

__global__ void k_k2_insert_random(K2Build B, uint64_t seed, uint64_t n, uint32_t lo, uint32_t hi)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t h = k2_fmix64(seed + 0x9E3779B97F4A7C15ULL * (i + 1));
        const uint64_t key = k2_fmix64(h ^ 0xD6E8FEB86659FD93ULL) & ((1ULL << 62) - 1);
        k2_insert(B, key, lo + (uint32_t)(h % (uint64_t)(hi - lo + 1)));
    }
}

#define K2_SEG 1024u
// one thread per K2_SEG-base segment of a long sequence: every minimizer whose k-mer ends inside the segment
template <int W>
__global__ void k_k2_insert_seq(K2Build B, const uint8_t *bases, uint64_t n, uint32_t taxon, int32_t k, int32_t l, uint64_t spaced,
                                uint64_t toggle, uint64_t min_hash, unsigned long long *n_runs)
{
    const uint64_t lmask = l < 32 ? ((1ULL << (2 * l)) - 1) : ~0ULL;
    const uint64_t n_seg = (n + K2_SEG - 1) / K2_SEG;
    unsigned long long runs = 0;
    for (uint64_t sg = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; sg < n_seg; sg += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t s0 = sg * K2_SEG, s1 = s0 + K2_SEG < n ? s0 + K2_SEG : n;
        const uint64_t from = s0 >= (uint64_t)(k - 1) ? s0 - (uint64_t)(k - 1) : 0;      // the first k-mer ending in the segment starts here
        K2Scan<W> S; S.reset();
        uint64_t last = ~0ull;
        for (uint64_t i = from; i < s1; ++i) {
            uint64_t m;
            // `pos` only gates "a full k-mer has been read": count from the warm-up start, never below the true position
            const uint64_t consumed = i - from + 1;
            const int ev = S.step(sh_nt4(bases[i]), (int32_t)(consumed > (uint64_t)k ? (uint64_t)k : consumed), k, l, lmask, spaced, toggle, m);
            if (ev != 2 || i < s0) continue;
            if (m == last) continue;
            last = m;
            if (min_hash && k2_fmix64(m) < min_hash) continue;
            k2_insert(B, m, taxon);
            ++runs;
        }
    }
    if (runs) atomicAdd(n_runs, runs);
}

// ---- host side -------------------------------------------------------------------------------------------------------
extern "C" sh_status sh_k2_default_opts(sh_k2_opts *o)
{
    SH_CHECK(o, SH_ERR_BAD_ARG, "sh_k2_default_opts: null argument");
    memset(o, 0, sizeof(*o));
    o->k = 35; o->l = 31;
    o->spaced_seed_mask = (0x3ffffffffULL << 28) | 0x3333333ULL;      // --minimizer-spaces 7
    o->toggle_mask = 0xe37e28c4271b5a2dULL;
    o->value_bits = 17;
    o->min_hit_groups = 2;
    o->confidence = 0.0;
    return SH_OK;
}

static sh_status k2_upload_taxonomy(sh_k2_db *db)
{
    const size_t n = db->nodes.size();
    std::vector<uint32_t> parent(n), ext(n);
    for (size_t i = 0; i < n; ++i) {
        SH_CHECK(db->nodes[i].parent < (i ? i : 1) || i <= 1, SH_ERR_BAD_ARG, "taxonomy: node %zu has parent %llu (ids must be breadth-first)", i,
                 (unsigned long long)db->nodes[i].parent);
        parent[i] = (uint32_t)db->nodes[i].parent; ext[i] = (uint32_t)db->nodes[i].external_id;
    }
    SH_HIP(hipMalloc(&db->d_parent, std::max<size_t>(n, 1) * 4));
    SH_HIP(hipMalloc(&db->d_ext, std::max<size_t>(n, 1) * 4));
    SH_HIP(hipMemcpy(db->d_parent, parent.data(), n * 4, hipMemcpyHostToDevice));
    SH_HIP(hipMemcpy(db->d_ext, ext.data(), n * 4, hipMemcpyHostToDevice));
    SH_HIP(hipMalloc(&db->d_ctr, K2C_WORDS * 8));
    SH_HIP(hipMemset(db->d_ctr, 0, K2C_WORDS * 8));
    return SH_OK;
}

extern "C" sh_status sh_k2_free(sh_k2_db *db)
{
    if (!db) return SH_OK;
    hipFree(db->d_cells); hipFree(db->d_parent); hipFree(db->d_ext); hipFree(db->d_ctr);
    delete db;
    return SH_OK;
}

extern "C" sh_status sh_k2_create(const sh_k2_opts *opts, uint64_t capacity, const sh_k2_taxnode *nodes, uint64_t n_nodes,
                                  const char *names, uint64_t names_len, const char *ranks, uint64_t ranks_len, int device, sh_k2_db **out)
{
    SH_CHECK(opts && nodes && out && capacity > 0 && n_nodes >= 2, SH_ERR_BAD_ARG, "sh_k2_create: bad argument");
    SH_CHECK(opts->l >= 1 && opts->l <= 31 && opts->k >= opts->l && opts->k - opts->l + 1 <= 16, SH_ERR_BAD_ARG, "sh_k2_create: unsupported k=%d l=%d", opts->k, opts->l);
    SH_CHECK(opts->value_bits >= 1 && opts->value_bits <= 31 && n_nodes <= (1ull << opts->value_bits), SH_ERR_BAD_ARG,
             "sh_k2_create: %llu taxonomy nodes do not fit %d value bits", (unsigned long long)n_nodes, opts->value_bits);
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= device) { sh_set_error("no HIP device %d", device); return SH_ERR_NO_DEVICE; }
    SH_HIP(hipSetDevice(device));
    sh_k2_db *db = new sh_k2_db;
    db->device = device; db->opts = *opts; db->capacity = capacity; db->value_bits = opts->value_bits; db->key_bits = 32 - opts->value_bits;
    db->nodes.assign(nodes, nodes + n_nodes);
    if (names) db->names.assign(names, names + names_len);
    if (ranks) db->ranks.assign(ranks, ranks + ranks_len);
    hipError_t e = hipMalloc(&db->d_cells, capacity * 4);
    if (e != hipSuccess) { sh_set_error("sh_k2_create: %llu cells: %s", (unsigned long long)capacity, hipGetErrorString(e)); sh_k2_free(db); return SH_ERR_OOM; }
    e = hipMemset(db->d_cells, 0, capacity * 4);
    if (e != hipSuccess) { sh_set_error("memset: %s", hipGetErrorString(e)); sh_k2_free(db); return SH_ERR_HIP; }
    sh_status st = k2_upload_taxonomy(db);
    if (st != SH_OK) { sh_k2_free(db); return st; }
    *out = db;
    return SH_OK;
}

static sh_status k2_sync_counts(sh_k2_db *db, hipStream_t s)
{
    unsigned long long c[2];
    SH_HIP(hipMemcpyAsync(c, db->d_ctr, 16, hipMemcpyDeviceToHost, s));
    SH_HIP(hipStreamSynchronize(s));
    SH_HIP(hipGetLastError());
    SH_CHECK(c[1] == 0, SH_ERR_OOM, "k2 table of %llu cells is full", (unsigned long long)db->capacity);
    db->size = c[0];
    return SH_OK;
}

static K2Build k2_build_args(sh_k2_db *db) { return K2Build{db->d_cells, db->capacity, db->value_bits, db->d_parent, db->d_ctr}; }

extern "C" sh_status sh_k2_insert_device(sh_k2_db *db, const uint64_t *d_keys, const uint32_t *d_taxa, uint64_t n, void *stream)
{
    SH_CHECK(db && (n == 0 || (d_keys && d_taxa)), SH_ERR_BAD_ARG, "sh_k2_insert_device: null argument");
    SH_HIP(hipSetDevice(db->device));
    hipStream_t s = (hipStream_t)stream;
    if (n) hipLaunchKernelGGL(k_k2_insert, dim3((uint32_t)std::min<uint64_t>((n + 255) / 256, 65536)), dim3(256), 0, s, k2_build_args(db), d_keys, d_taxa, n);
    return k2_sync_counts(db, s);
}

extern "C" sh_status sh_k2_insert_random(sh_k2_db *db, uint64_t seed, uint64_t n, uint32_t lo, uint32_t hi, void *stream)
{
    SH_CHECK(db && lo >= 1 && hi >= lo && hi < db->nodes.size(), SH_ERR_BAD_ARG, "sh_k2_insert_random: taxon range [%u, %u] outside the taxonomy", lo, hi);
    SH_HIP(hipSetDevice(db->device));
    hipStream_t s = (hipStream_t)stream;
    if (n) hipLaunchKernelGGL(k_k2_insert_random, dim3((uint32_t)std::min<uint64_t>((n + 255) / 256, 65536)), dim3(256), 0, s, k2_build_args(db), seed, n, lo, hi);
    return k2_sync_counts(db, s);
}

template <int W>
static void launch_insert_seq(sh_k2_db *db, const uint8_t *d_bases, uint64_t n, uint32_t taxon, hipStream_t s, unsigned long long *d_runs)
{
    const uint64_t n_seg = (n + K2_SEG - 1) / K2_SEG;
    hipLaunchKernelGGL(k_k2_insert_seq<W>, dim3((uint32_t)std::min<uint64_t>((n_seg + 63) / 64, 1 << 20)), dim3(64), 0, s, k2_build_args(db), d_bases, n, taxon,
                       db->opts.k, db->opts.l, db->opts.spaced_seed_mask, db->opts.toggle_mask, db->opts.min_acceptable_hash, d_runs);
}

extern "C" sh_status sh_k2_insert_sequence_device(sh_k2_db *db, const uint8_t *d_bases, uint64_t n, uint32_t taxon, void *stream, uint64_t *n_inserted)
{
    SH_CHECK(db && d_bases && taxon >= 1 && taxon < db->nodes.size(), SH_ERR_BAD_ARG, "sh_k2_insert_sequence_device: bad argument");
    SH_HIP(hipSetDevice(db->device));
    hipStream_t s = (hipStream_t)stream;
    unsigned long long *d_runs = db->d_ctr + 2;
    SH_HIP(hipMemsetAsync(d_runs, 0, 8, s));
    if (n >= (uint64_t)db->opts.k) {
        switch (db->opts.k - db->opts.l + 1) {
        case 1: launch_insert_seq<1>(db, d_bases, n, taxon, s, d_runs); break;
        case 5: launch_insert_seq<5>(db, d_bases, n, taxon, s, d_runs); break;
        default: launch_insert_seq<16>(db, d_bases, n, taxon, s, d_runs); break;
        }
    }
    unsigned long long r = 0;
    SH_HIP(hipMemcpyAsync(&r, d_runs, 8, hipMemcpyDeviceToHost, s));
    sh_status st = k2_sync_counts(db, s);
    if (n_inserted) *n_inserted = r;
    return st;
}

extern "C" sh_status sh_k2_info_get(const sh_k2_db *db, sh_k2_info *o)
{
    SH_CHECK(db && o, SH_ERR_BAD_ARG, "sh_k2_info_get: null argument");
    o->capacity = db->capacity; o->size = db->size; o->n_nodes = db->nodes.size();
    o->hbm_bytes = db->capacity * 4 + db->nodes.size() * 8;
    o->k = db->opts.k; o->l = db->opts.l; o->value_bits = db->value_bits; o->key_bits = db->key_bits;
    return SH_OK;
}

extern "C" sh_status sh_k2_db_opts(const sh_k2_db *db, sh_k2_opts *o)
{
    SH_CHECK(db && o, SH_ERR_BAD_ARG, "sh_k2_db_opts: null argument");
    *o = db->opts;
    return SH_OK;
}

extern "C" sh_status sh_k2_export(const sh_k2_db *db, uint32_t *cells, uint32_t *parent, uint32_t *external)
{
    SH_CHECK(db, SH_ERR_BAD_ARG, "sh_k2_export: null argument");
    SH_HIP(hipSetDevice(db->device));
    if (cells) SH_HIP(hipMemcpy(cells, db->d_cells, db->capacity * 4, hipMemcpyDeviceToHost));
    if (parent) SH_HIP(hipMemcpy(parent, db->d_parent, db->nodes.size() * 4, hipMemcpyDeviceToHost));
    if (external) SH_HIP(hipMemcpy(external, db->d_ext, db->nodes.size() * 4, hipMemcpyDeviceToHost));
    return SH_OK;
}

// ---- database files (SURVEY.md App. B "DB files") --------------------------------------------------------------------
struct K2OptsFile {         // opts.k2d: the index options struct as written by the builder (64 B with padding)
    uint64_t k, l, spaced_seed_mask, toggle_mask;
    uint8_t dna_db; uint8_t pad0[7];
    uint64_t minimum_acceptable_hash_value;
    int32_t revcom_version, db_version, db_type, pad1;
};
static_assert(sizeof(K2OptsFile) == 64, "opts.k2d layout");

extern "C" sh_status sh_k2_save(const sh_k2_db *db, const char *dir)
{
    SH_CHECK(db && dir, SH_ERR_BAD_ARG, "sh_k2_save: null argument");
    SH_HIP(hipSetDevice(db->device));
    const std::string d = dir;
    {
        K2OptsFile of{};
        of.k = (uint64_t)db->opts.k; of.l = (uint64_t)db->opts.l; of.spaced_seed_mask = db->opts.spaced_seed_mask; of.toggle_mask = db->opts.toggle_mask;
        of.dna_db = (uint8_t)db->dna_db; of.minimum_acceptable_hash_value = db->opts.min_acceptable_hash;
        of.revcom_version = db->revcom_version; of.db_version = db->db_version; of.db_type = db->db_type;
        FILE *f = fopen((d + "/opts.k2d").c_str(), "wb");
        SH_CHECK(f, SH_ERR_IO, "cannot write %s/opts.k2d", dir);
        bool ok = fwrite(&of, sizeof(of), 1, f) == 1;
        ok = fclose(f) == 0 && ok;
        SH_CHECK(ok, SH_ERR_IO, "short write to %s/opts.k2d", dir);
    }
    {
        FILE *f = fopen((d + "/taxo.k2d").c_str(), "wb");
        SH_CHECK(f, SH_ERR_IO, "cannot write %s/taxo.k2d", dir);
        const uint64_t hdr[3] = {db->nodes.size(), db->names.size(), db->ranks.size()};
        bool ok = fwrite("K2TAXDAT", 8, 1, f) == 1 && fwrite(hdr, 8, 3, f) == 3;
        ok = ok && fwrite(db->nodes.data(), sizeof(sh_k2_taxnode), db->nodes.size(), f) == db->nodes.size();
        ok = ok && (db->names.empty() || fwrite(db->names.data(), 1, db->names.size(), f) == db->names.size());
        ok = ok && (db->ranks.empty() || fwrite(db->ranks.data(), 1, db->ranks.size(), f) == db->ranks.size());
        ok = fclose(f) == 0 && ok;
        SH_CHECK(ok, SH_ERR_IO, "short write to %s/taxo.k2d", dir);
    }
    {
        FILE *f = fopen((d + "/hash.k2d").c_str(), "wb");
        SH_CHECK(f, SH_ERR_IO, "cannot write %s/hash.k2d", dir);
        const uint64_t hdr[4] = {db->capacity, db->size, (uint64_t)db->key_bits, (uint64_t)db->value_bits};
        bool ok = fwrite(hdr, 8, 4, f) == 4;
        const uint64_t CH = 64ull << 20;            // cells per copy
        std::vector<uint32_t> buf(std::min(CH, db->capacity));
        for (uint64_t o = 0; ok && o < db->capacity; o += CH) {
            const uint64_t m = std::min(CH, db->capacity - o);
            if (hipMemcpy(buf.data(), db->d_cells + o, m * 4, hipMemcpyDeviceToHost) != hipSuccess) { fclose(f); sh_set_error("copy of the table failed"); return SH_ERR_HIP; }
            ok = fwrite(buf.data(), 4, m, f) == m;
        }
        ok = fclose(f) == 0 && ok;
        SH_CHECK(ok, SH_ERR_IO, "short write to %s/hash.k2d", dir);
    }
    return SH_OK;
}

extern "C" sh_status sh_k2_open(const char *dir, int device, sh_k2_db **out)
{
    SH_CHECK(dir && out, SH_ERR_BAD_ARG, "sh_k2_open: null argument");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= device) { sh_set_error("no HIP device %d", device); return SH_ERR_NO_DEVICE; }
    SH_HIP(hipSetDevice(device));
    const std::string d = dir;
    sh_k2_db *db = new sh_k2_db;
    db->device = device;
    auto fail = [&](sh_status st) { sh_k2_free(db); return st; };
    {
        FILE *f = fopen((d + "/opts.k2d").c_str(), "rb");
        if (!f) { sh_set_error("cannot open %s/opts.k2d", dir); return fail(SH_ERR_IO); }
        K2OptsFile of{};
        const size_t got = fread(&of, 1, sizeof(of), f);        // older databases wrote a shorter struct: the rest stays zero
        fclose(f);
        if (got < 32) { sh_set_error("%s/opts.k2d is truncated", dir); return fail(SH_ERR_IO); }
        sh_k2_default_opts(&db->opts);
        db->opts.k = (int32_t)of.k; db->opts.l = (int32_t)of.l; db->opts.spaced_seed_mask = of.spaced_seed_mask; db->opts.toggle_mask = of.toggle_mask;
        db->opts.min_acceptable_hash = got >= 48 ? of.minimum_acceptable_hash_value : 0;
        db->dna_db = got >= 33 ? of.dna_db : 1; db->revcom_version = got >= 52 ? of.revcom_version : 0;
        db->db_version = got >= 56 ? of.db_version : 0; db->db_type = got >= 60 ? of.db_type : 0;
        if (!db->dna_db) { sh_set_error("%s: protein databases are not supported", dir); return fail(SH_ERR_BAD_ARG); }
        if (!(db->opts.l >= 1 && db->opts.l <= 31 && db->opts.k >= db->opts.l && db->opts.k - db->opts.l + 1 <= 16)) {
            sh_set_error("%s: unsupported k=%d l=%d", dir, db->opts.k, db->opts.l); return fail(SH_ERR_BAD_ARG);
        }
    }
    {
        FILE *f = fopen((d + "/taxo.k2d").c_str(), "rb");
        if (!f) { sh_set_error("cannot open %s/taxo.k2d", dir); return fail(SH_ERR_IO); }
        char magic[8]; uint64_t hdr[3];
        bool ok = fread(magic, 8, 1, f) == 1 && memcmp(magic, "K2TAXDAT", 8) == 0 && fread(hdr, 8, 3, f) == 3;
        if (ok) {
            db->nodes.resize(hdr[0]); db->names.resize(hdr[1]); db->ranks.resize(hdr[2]);
            ok = fread(db->nodes.data(), sizeof(sh_k2_taxnode), hdr[0], f) == hdr[0];
            ok = ok && (hdr[1] == 0 || fread(&db->names[0], 1, hdr[1], f) == hdr[1]);
            ok = ok && (hdr[2] == 0 || fread(&db->ranks[0], 1, hdr[2], f) == hdr[2]);
        }
        fclose(f);
        if (!ok || db->nodes.size() < 2) { sh_set_error("%s/taxo.k2d is not a Kraken 2 taxonomy", dir); return fail(SH_ERR_IO); }
    }
    {
        FILE *f = fopen((d + "/hash.k2d").c_str(), "rb");
        if (!f) { sh_set_error("cannot open %s/hash.k2d", dir); return fail(SH_ERR_IO); }
        uint64_t hdr[4];
        if (fread(hdr, 8, 4, f) != 4 || hdr[0] == 0 || hdr[2] + hdr[3] != 32 || hdr[3] < 1 || hdr[3] > 31) { fclose(f); sh_set_error("%s/hash.k2d: bad header", dir); return fail(SH_ERR_IO); }
        db->capacity = hdr[0]; db->size = hdr[1]; db->key_bits = (int32_t)hdr[2]; db->value_bits = (int32_t)hdr[3]; db->opts.value_bits = db->value_bits;
        hipError_t e = hipMalloc(&db->d_cells, db->capacity * 4);
        if (e != hipSuccess) { fclose(f); sh_set_error("table of %llu cells: %s", (unsigned long long)db->capacity, hipGetErrorString(e)); return fail(SH_ERR_OOM); }
        const uint64_t CH = 64ull << 20;
        std::vector<uint32_t> buf(std::min(CH, db->capacity));
        for (uint64_t o = 0; o < db->capacity; o += CH) {
            const uint64_t m = std::min(CH, db->capacity - o);
            if (fread(buf.data(), 4, m, f) != m) { fclose(f); sh_set_error("%s/hash.k2d is truncated", dir); return fail(SH_ERR_IO); }
            if (hipMemcpy(db->d_cells + o, buf.data(), m * 4, hipMemcpyHostToDevice) != hipSuccess) { fclose(f); sh_set_error("copy to HBM failed"); return fail(SH_ERR_HIP); }
        }
        fclose(f);
    }
    if (db->nodes.size() > (1ull << db->value_bits)) { sh_set_error("%s: taxonomy larger than the table's value range", dir); return fail(SH_ERR_BAD_ARG); }
    sh_status st = k2_upload_taxonomy(db);
    if (st != SH_OK) return fail(st);
    *out = db;
    return SH_OK;
}

// ---- classification ---------------------------------------------------------------------------------------------------
template <int W, bool BIG>
static void launch_classify(const K2Args &a, uint64_t n_work, hipStream_t s)
{
    const uint32_t grid = (uint32_t)std::min<uint64_t>(std::max<uint64_t>((n_work + 63) / 64, 1), 256 * 32);
    hipLaunchKernelGGL((k_k2_classify<W, BIG>), dim3(grid), dim3(64), 0, s, a);
}
template <bool BIG>
static sh_status dispatch_classify(const K2Args &a, uint64_t n_work, hipStream_t s)
{
    switch (a.k - a.l + 1) {
    case 1: launch_classify<1, BIG>(a, n_work, s); break;
    case 5: launch_classify<5, BIG>(a, n_work, s); break;
    default: launch_classify<16, BIG>(a, n_work, s); break;
    }
    return SH_OK;
}

extern "C" sh_status sh_k2_classify_device(const sh_k2_db *db, const sh_k2_opts *opts, const uint8_t *d_bases, const uint64_t *d_offsets,
                                           uint64_t n_records, int32_t paired, sh_k2_result *d_out, void *stream, sh_k2_stats *stats)
{
    SH_CHECK(db && d_offsets && d_out && (d_bases || n_records == 0), SH_ERR_BAD_ARG, "sh_k2_classify_device: null argument");
    SH_CHECK(!paired || (n_records & 1) == 0, SH_ERR_BAD_ARG, "paired input needs an even number of records (got %llu)", (unsigned long long)n_records);
    SH_HIP(hipSetDevice(db->device));
    hipStream_t s = (hipStream_t)stream;
    const sh_k2_opts &o = opts ? *opts : db->opts;
    SH_CHECK(o.k == db->opts.k && o.l == db->opts.l, SH_ERR_BAD_ARG, "options disagree with the database (k, l)");
    const uint64_t n_units = paired ? n_records / 2 : n_records;
    if (stats) memset(stats, 0, sizeof(*stats));
    if (n_units == 0) return SH_OK;
    SH_CHECK(n_units <= 0xffffffffull, SH_ERR_BAD_ARG, "at most 2^32 - 1 units per call");
    hipEvent_t e0, e1;
    SH_HIP(hipEventCreate(&e0)); SH_HIP(hipEventCreate(&e1));
    unsigned long long *ctr = nullptr; uint32_t *over = nullptr;
    SH_HIP(hipMalloc(&ctr, K2C_WORDS * 8));
    SH_HIP(hipMalloc(&over, n_units * 4));
    SH_HIP(hipMemsetAsync(ctr, 0, K2C_WORDS * 8, s));
    K2Args a{};
    a.bases = d_bases; a.offsets = d_offsets; a.n_units = n_units; a.paired = paired;
    a.T = K2Table{db->d_cells, db->capacity, db->value_bits, 1.0 / (double)db->capacity}; a.parent = db->d_parent; a.ext = db->d_ext; a.n_nodes = (uint32_t)db->nodes.size();
    a.k = o.k; a.l = o.l; a.spaced = o.spaced_seed_mask; a.toggle = o.toggle_mask; a.min_hash = o.min_acceptable_hash;
    a.min_hit_groups = o.min_hit_groups; a.confidence = o.confidence;
    a.out = d_out; a.over_list = over; a.ctr = ctr;
    SH_HIP(hipEventRecord(e0, s));
    dispatch_classify<false>(a, n_units, s);
    SH_HIP(hipEventRecord(e1, s));
    std::vector<unsigned long long> h(K2C_WORDS);
    SH_HIP(hipMemcpyAsync(h.data(), ctr, K2C_WORDS * 8, hipMemcpyDeviceToHost, s));
    SH_HIP(hipStreamSynchronize(s));
    SH_HIP(hipGetLastError());
    const uint64_t n_over = h[K2C_OVER];
    uint32_t *big_tax = nullptr, *big_cnt = nullptr;
    if (n_over) {       // units with more than K2_HCAP distinct taxa: hit lists in HBM
        SH_HIP(hipMalloc(&big_tax, n_over * K2_BIG_CAP * 4));
        SH_HIP(hipMalloc(&big_cnt, n_over * K2_BIG_CAP * 4));
        a.unit_list = over; a.n_list = (uint32_t)n_over; a.big_tax = big_tax; a.big_cnt = big_cnt;
        dispatch_classify<true>(a, n_over, s);
        SH_HIP(hipMemcpyAsync(h.data(), ctr, K2C_WORDS * 8, hipMemcpyDeviceToHost, s));
        SH_HIP(hipStreamSynchronize(s));
        SH_HIP(hipGetLastError());
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    if (stats) {
        stats->n_units = n_units; stats->n_overflow = n_over; stats->ms_classify = ms; stats->ms_total = ms;
        for (int i = 0; i < 64; ++i) { stats->n_probes += h[K2C_PROBES + i]; stats->n_kmers += h[K2C_KMERS + i]; stats->n_classified += h[K2C_CLASSIFIED + i]; }
    }
    hipFree(ctr); hipFree(over); hipFree(big_tax); hipFree(big_cnt);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return SH_OK;
}

extern "C" sh_status sh_k2_classify_batch(const sh_k2_db *db, const sh_k2_opts *opts, const uint8_t *bases, const uint64_t *offsets,
                                          uint64_t n_records, int32_t paired, sh_k2_result *out, sh_k2_stats *stats)
{
    SH_CHECK(db && offsets && out, SH_ERR_BAD_ARG, "sh_k2_classify_batch: null argument");
    if (stats) memset(stats, 0, sizeof(*stats));
    if (n_records == 0) return SH_OK;
    SH_HIP(hipSetDevice(db->device));
    const uint64_t o0 = offsets[0], n_bases = offsets[n_records] - o0;
    const uint64_t n_units = paired ? n_records / 2 : n_records;
    uint8_t *d_bases = nullptr; uint64_t *d_off = nullptr; sh_k2_result *d_out = nullptr;
    SH_HIP(hipMalloc(&d_bases, n_bases + 64));
    SH_HIP(hipMalloc(&d_off, (n_records + 1) * 8));
    SH_HIP(hipMalloc(&d_out, std::max<uint64_t>(n_units, 1) * sizeof(sh_k2_result)));
    std::vector<uint64_t> rel(n_records + 1);
    for (uint64_t i = 0; i <= n_records; ++i) rel[i] = offsets[i] - o0;
    SH_HIP(hipMemcpy(d_bases, bases + o0, n_bases, hipMemcpyHostToDevice));
    SH_HIP(hipMemset(d_bases + n_bases, 'N', 64));
    SH_HIP(hipMemcpy(d_off, rel.data(), (n_records + 1) * 8, hipMemcpyHostToDevice));
    sh_status st = sh_k2_classify_device(db, opts, d_bases, d_off, n_records, paired, d_out, nullptr, stats);
    if (st == SH_OK && hipMemcpy(out, d_out, n_units * sizeof(sh_k2_result), hipMemcpyDeviceToHost) != hipSuccess) { sh_set_error("copy of the results failed"); st = SH_ERR_HIP; }
    hipFree(d_bases); hipFree(d_off); hipFree(d_out);
    return st;
}

// ---- Kraken-style report (SURVEY.md App. B "Outputs consumed by Scrubby"; parsed by classifier.rs:449-466) -------------
static const char *k2_pool(const std::string &pool, uint64_t off) { return off < pool.size() ? pool.c_str() + off : ""; }

extern "C" sh_status sh_k2_write_report(const sh_k2_db *db, const sh_k2_result *res, uint64_t n_units, const char *path)
{
    SH_CHECK(db && path && (res || n_units == 0), SH_ERR_BAD_ARG, "sh_k2_write_report: null argument");
    const size_t n = db->nodes.size();
    std::vector<uint64_t> direct(n, 0), clade(n, 0);
    uint64_t unclassified = 0;
    for (uint64_t i = 0; i < n_units; ++i) { if (res[i].call && res[i].call < n) ++direct[res[i].call]; else ++unclassified; }
    clade = direct;
    for (size_t i = n - 1; i >= 2; --i) clade[db->nodes[i].parent] += clade[i];      // parents have smaller ids
    FILE *f = fopen(path, "w");
    SH_CHECK(f, SH_ERR_IO, "cannot write %s", path);
    const double total = n_units ? (double)n_units : 1.0;
    if (unclassified) fprintf(f, "%6.2f\t%llu\t%llu\tU\t0\tunclassified\n", 100.0 * (double)unclassified / total, (unsigned long long)unclassified, (unsigned long long)unclassified);
    // depth-first from the root, children by clade count (descending; ties by id), rank codes with a depth suffix
    struct Frame { uint32_t id; std::string code; int code_depth; int depth; };
    std::vector<Frame> stack;
    if (clade[1]) stack.push_back(Frame{1, "R", 0, 0});
    while (!stack.empty()) {
        Frame fr = stack.back(); stack.pop_back();
        const sh_k2_taxnode &nd = db->nodes[fr.id];
        std::string code = fr.code; int cd = fr.code_depth;
        if (fr.id != 1) {
            const std::string rank = k2_pool(db->ranks, nd.rank_offset);
            const char *letter = nullptr;
            if (rank == "superkingdom") letter = "D"; else if (rank == "kingdom") letter = "K"; else if (rank == "phylum") letter = "P";
            else if (rank == "class") letter = "C"; else if (rank == "order") letter = "O"; else if (rank == "family") letter = "F";
            else if (rank == "genus") letter = "G"; else if (rank == "species") letter = "S";
            if (letter) { code = letter; cd = 0; } else ++cd;
        }
        std::string rc = code; if (cd) rc += std::to_string(cd);
        fprintf(f, "%6.2f\t%llu\t%llu\t%s\t%llu\t", 100.0 * (double)clade[fr.id] / total, (unsigned long long)clade[fr.id], (unsigned long long)direct[fr.id], rc.c_str(),
                (unsigned long long)nd.external_id);
        for (int i = 0; i < fr.depth; ++i) fputs("  ", f);
        fprintf(f, "%s\n", k2_pool(db->names, nd.name_offset));
        std::vector<uint32_t> kids;
        for (uint64_t c = 0; c < nd.child_count; ++c) { const uint64_t id = nd.first_child + c; if (id < n && clade[id]) kids.push_back((uint32_t)id); }
        std::sort(kids.begin(), kids.end(), [&](uint32_t x, uint32_t y) { return clade[x] != clade[y] ? clade[x] > clade[y] : x < y; });
        for (size_t i = kids.size(); i-- > 0;) stack.push_back(Frame{kids[i], code, cd, fr.depth + 1});
    }
    const bool ok = fclose(f) == 0;
    SH_CHECK(ok, SH_ERR_IO, "short write to %s", path);
    return SH_OK;
}
