// sh_classify.hip — per-read host/non-host classification on gfx950.
//
// Replaces the hot loop of the reference's in-process aligner path
//     .map(|(id, sequence)| aligner.map(&sequence, false, false, None, None) ... mappings.len() > 0)
//     /root/reference/src/cleaner.rs:550-558
// with three hand-written HIP kernels over a read batch resident in HBM:
//
//   K1 k_sketch_probe   one wave per tile of 64 consecutive reads, one lane per read.
//        A  the tile's bases (one contiguous byte range) are loaded coalesced, 16 B per lane,
//           converted to 2-bit codes + an ambiguity bit and staged in LDS;
//        B  every lane runs the (w,k)-minimizer state machine over its read with the w-entry
//           window in VGPRs (sh_sketch.h), queueing minimizers in an LDS list;
//        C  when a list fills (and at the end) the wave probes the HBM hash index for all queued
//           minimizers, 4 independent 16-B gathers in flight per lane, and writes one 16-B seed
//           record per hit, lane-interleaved so that the 64 lanes of a tile write one 1-KiB row.
//        Reads without a single hit are final here (flag 0): no anchor => no mapping.
//   K2 k_chain_small    one lane per read with >=1 seed: occurrence filter, anchors, chaining DP and
//        backtrack entirely in LDS (11 B per anchor, lane-interleaved), up to CAP anchors.
//   K3 k_chain_large    the same code over per-read slices of an HBM arena for reads with more
//        anchors (repeats) — binned by anchor count so that the lanes of a wave carry similar work —
//        and for the rare reads K1 could not finish (seed/list overflow), which it re-sketches.
//
// Results are bit-identical to oracle/mm_oracle.c (tests/test_parity_gpu.py).
#include "sh_common.h"
#include "sh_sketch.h"
#include "sh_chain.h"
#include <algorithm>

#define K1_LIST_CAP 32          // queued minimizers per lane between two probe phases
#define K1_FLUSH_AT 16
#define K2_CAP 32               // anchors per read chained in LDS
#define N_BUCKETS 6             // (32,128] (128,512] (512,2048] (2048,8192] (8192,32768] >32768

struct Counters {
    uint32_t n_small, n_resketch, n_large[N_BUCKETS], n_defer, n_noseed, n_host, n_done_small, n_done_large, pad;
    unsigned long long arena_cursor, sum_mini;
};

__device__ inline uint32_t lane_id() { return threadIdx.x & 63; }
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
    base = __shfl(base, leader);
    return pred ? base + prefix_popc(mask) : ~0u;
}

__device__ inline void write_trace(sh_trace *tr, uint64_t r, int32_t n_mini, int32_t n_seed, int32_t n_anchor, int32_t rep_len,
                                   int32_t rechained, int32_t n_chain, int32_t best, int32_t flag)
{
    if (!tr) return;
    int4 *p = (int4 *)(tr + r);
    p[0] = make_int4(n_mini, n_seed, n_anchor, rep_len);
    p[1] = make_int4(rechained, n_chain, best, flag);
}

// ------------------------------------------------------------------------------------------------
// K1
// ------------------------------------------------------------------------------------------------
struct K1Args {
    const uint8_t *bases; const uint64_t *offsets; uint64_t n_reads, n_bases;
    const uint4 *slots; uint32_t lg_slots; int32_t k;
    uint4 *records; uint32_t seed_cap;
    uint32_t *k1info; uint8_t *flags; sh_trace *trace;
    uint32_t *work_small, *work_resketch; Counters *ctr;
    uint32_t lds_words;
};

template <int W>
__global__ __launch_bounds__(64) void k_sketch_probe(K1Args a)
{
    extern __shared__ __align__(16) uint8_t smem[];
    uint64_t *list = (uint64_t *)smem;
    uint32_t *pk = (uint32_t *)(smem + (size_t)K1_LIST_CAP * 64 * 8);
    uint16_t *nm = (uint16_t *)(pk + a.lds_words + 2);

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
    SketchState<W> st;
    st.init(a.k);
    const uint32_t b0 = (uint32_t)((base_addr + o_beg) - a0);    // tile-relative index of this read's first base
    uint32_t maxlen = len;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) maxlen = max(maxlen, (uint32_t)__shfl_xor((int)maxlen, o));

    uint32_t cnt = 0, n_mini = 0, n_seed = 0;
    bool overflow = false;
    auto emit = [&](uint64_t x, uint32_t y) {
        if (cnt < K1_LIST_CAP) list[cnt * 64 + lane] = (x >> 8) << 18 | (uint64_t)y;
        else overflow = true;
        ++cnt; ++n_mini;
    };
    const uint64_t slot_mask = (1ULL << a.lg_slots) - 1;
    uint4 *rec = a.records + (size_t)tile * a.seed_cap * 64 + lane;

    uint32_t codes = 0, amb = 0;
    uint32_t i0 = 0;
    for (;;) {
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
                    st.template step<P>(c, i, emit);
                }
                codes >>= 2; amb >>= 1;
            };
            [&]<int... Ps>(std::integer_sequence<int, Ps...>) { (one(std::integral_constant<int, Ps>{}), ...); }
            (std::make_integer_sequence<int, W>{});
            i0 += W;
        }
        const bool last = i0 >= maxlen;
        if (last && valid && len > 0) st.finish(emit);
        if (last || __ballot(cnt >= K1_FLUSH_AT) != 0) {
            // ---- C: probe the index for every queued minimizer --------------------------------
            const uint32_t c_here = cnt < K1_LIST_CAP ? cnt : K1_LIST_CAP;
            for (uint32_t e0 = 0; __ballot(e0 < c_here) != 0; e0 += 4) {
                uint64_t key[4], idx[4]; uint32_t yq[4]; uint4 s[4]; bool act[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    act[u] = e0 + u < c_here;
                    if (act[u]) {
                        uint64_t m = list[(e0 + u) * 64 + lane];
                        key[u] = m >> 18; yq[u] = (uint32_t)m & 0x3ffffu;
                        idx[u] = sh_slot_home(key[u], a.lg_slots);
                        s[u] = a.slots[idx[u]];
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (!act[u]) continue;
                    uint64_t w0 = (uint64_t)s[u].y << 32 | s[u].x;
                    while (w0 != SH_SLOT_EMPTY && (w0 & SH_SLOT_KEYMASK) != key[u]) {
                        idx[u] = (idx[u] + 1) & slot_mask;
                        s[u] = a.slots[idx[u]];
                        w0 = (uint64_t)s[u].y << 32 | s[u].x;
                    }
                    if (w0 != SH_SLOT_EMPTY) {
                        uint32_t occ = (w0 & SH_SLOT_MULTI) ? (s[u].z & (uint32_t)SH_SLOT_NMASK) : 1u;
                        if (n_seed < a.seed_cap) rec[(size_t)n_seed * 64] = make_uint4(s[u].z, s[u].w, occ, yq[u]);
                        else overflow = true;
                        ++n_seed;
                    }
                }
            }
            cnt = 0;
        }
        if (last) break;
    }

    // ---- per-read result ------------------------------------------------------------------------
    if (valid) a.k1info[r] = n_mini | n_seed << 16;
    const bool to_k3 = valid && overflow;
    const bool done = valid && !overflow && n_seed == 0;
    const bool to_k2 = valid && !overflow && n_seed > 0;
    if (done) {
        int32_t fl = len == 0 ? 2 : 0;
        a.flags[r] = (uint8_t)fl;
        write_trace(a.trace, r, (int32_t)n_mini, 0, 0, 0, 0, 0, 0, fl);
    }
    uint32_t wi = wave_append(&a.ctr->n_small, to_k2);
    if (to_k2) a.work_small[wi] = (uint32_t)r;
    wi = wave_append(&a.ctr->n_resketch, to_k3);
    if (to_k3) a.work_resketch[wi] = (uint32_t)r;
    // statistics
    uint64_t mdone = __ballot(done);
    uint32_t msum = n_mini;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) msum += (uint32_t)__shfl_xor((int)msum, o);
    if (lane == 0) {
        if (mdone) atomicAdd(&a.ctr->n_noseed, (uint32_t)__popcll(mdone));
        atomicAdd(&a.ctr->sum_mini, (unsigned long long)msum);
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
struct K2Args {
    const uint64_t *offsets; const uint8_t *bases; uint64_t n_reads;
    const uint4 *slots; uint32_t lg_slots; int32_t w;
    const uint64_t *positions;
    uint4 *records; uint32_t seed_cap;
    const uint32_t *k1info; uint8_t *flags; sh_trace *trace;
    const uint32_t *work; const uint32_t *work_count;        // input list
    uint32_t *work_large[N_BUCKETS]; uint32_t *work_defer;   // outputs
    Counters *ctr;
    uint8_t *arena; unsigned long long arena_bytes;
    ChainParams P;
    uint32_t mode;       // K3: 0 = seeds from tile records, 1 = re-sketch
};

__device__ inline int bucket_of(int64_t n_a)
{
    int lg = 63 - __clzll((unsigned long long)(n_a - 1));   // n_a > 32  =>  lg >= 5
    int b = (lg - 5) >> 1;
    return b < N_BUCKETS - 1 ? b : N_BUCKETS - 1;
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

    for (uint32_t base = blockIdx.x * 64; base < n_work; base += gridDim.x * 64) {
        const uint32_t wi = base + lane;
        const bool valid = wi < n_work;
        bool host = false;
        if (valid) {
            const uint32_t r = a.work[wi];
            const uint32_t info = a.k1info[r];
            const int32_t n_mini = (int32_t)(info & 0xffffu), n_seed = (int32_t)(info >> 16);
            const int32_t qlen = (int32_t)(a.offsets[r + 1] - a.offsets[r]);
            SeedView sv;
            sv.base = a.records + (size_t)(r >> 6) * a.seed_cap * 64 + (r & 63);
            sv.stride = 64; sv.n = (uint32_t)n_seed;
            int32_t max_occ = a.P.mid_occ, rechained = 0, n_u = 0, best = 0, rep_len = 0;
            int64_t n_a = 0;
            bool routed = false;
            for (;;) {
                seed_filter(sv, qlen, max_occ, a.P, n_a, rep_len);
                if (n_a > CAP) { routed = true; break; }
                gen_anchors(S, sv, a.positions, qlen, a.P.k);
                chain_dp<SmallStore<CAP>, int>(S, (int)n_a, qlen, a.P);
                backtrack_small(S, (int)n_a, a.P, n_u, best);
                if (!rechained && n_u == 0 && a.P.max_occ > a.P.mid_occ && rep_len > 0) { rechained = 1; max_occ = a.P.max_occ; continue; }
                break;
            }
            if (routed) {
                int b = bucket_of(n_a);
                uint32_t li = atomicAdd(&a.ctr->n_large[b], 1u);
                a.work_large[b][li] = r;
            } else {
                finish_read(a, r, n_mini, n_seed, n_a, rep_len, rechained, n_u, best);
                host = n_u > 0;
            }
        }
        uint64_t mh = __ballot(host);
        if (lane == 0 && mh) atomicAdd(&a.ctr->n_host, (uint32_t)__popcll(mh));
    }
}

__device__ inline uint8_t *arena_alloc(const K2Args &a, size_t bytes)
{
    bytes = (bytes + 15) & ~(size_t)15;
    unsigned long long off = atomicAdd(&a.ctr->arena_cursor, (unsigned long long)bytes);
    if (off + bytes > a.arena_bytes) return nullptr;
    return a.arena + off;
}

// K3: lane per read, arrays in the HBM arena.  mode 1 first rebuilds the seed records by a
// sequential sketch + probe (runtime w, ring in the arena).
__global__ __launch_bounds__(64) void k_chain_large(K2Args a)
{
    const uint32_t lane = threadIdx.x;
    const uint32_t n_work = *a.work_count;
    for (uint32_t base = blockIdx.x * 64; base < n_work; base += gridDim.x * 64) {
        const uint32_t wi = base + lane;
        bool host = false;
        if (wi < n_work) {
            const uint32_t entry = a.work[wi];
            const uint32_t r = entry & 0x7fffffffu;
            const uint32_t mode = a.mode == 2 ? entry >> 31 : a.mode;     // mode 2: per-entry (deferred reads keep theirs)
            const uint64_t o_beg = a.offsets[r];
            const int32_t qlen = (int32_t)(a.offsets[r + 1] - o_beg);
            int32_t n_mini, n_seed;
            SeedView sv;
            bool defer = false;
            if (mode == 0) {
                const uint32_t info = a.k1info[r];
                n_mini = (int32_t)(info & 0xffffu); n_seed = (int32_t)(info >> 16);
                sv.base = a.records + (size_t)(r >> 6) * a.seed_cap * 64 + (r & 63);
                sv.stride = 64; sv.n = (uint32_t)n_seed;
            } else {
                n_mini = 0; n_seed = 0;
                const size_t cap = 2 * (size_t)qlen + 256;
                uint8_t *m = arena_alloc(a, cap * 16 + (size_t)a.w * 16);
                if (!m) defer = true;
                else {
                    uint4 *recs = (uint4 *)m;
                    uint64_t *rbx = (uint64_t *)(m + cap * 16);
                    uint32_t *rby = (uint32_t *)(rbx + a.w);
                    SketchStateDyn st;
                    st.init(rbx, rby, a.w, a.P.k);
                    const uint64_t slot_mask = (1ULL << a.lg_slots) - 1;
                    auto emit = [&](uint64_t x, uint32_t y) {
                        ++n_mini;
                        uint64_t key = x >> 8, idx = sh_slot_home(key, a.lg_slots);
                        uint4 s = a.slots[idx];
                        uint64_t w0 = (uint64_t)s.y << 32 | s.x;
                        while (w0 != SH_SLOT_EMPTY && (w0 & SH_SLOT_KEYMASK) != key) {
                            idx = (idx + 1) & slot_mask; s = a.slots[idx]; w0 = (uint64_t)s.y << 32 | s.x;
                        }
                        if (w0 != SH_SLOT_EMPTY) {
                            uint32_t occ = (w0 & SH_SLOT_MULTI) ? (s.z & (uint32_t)SH_SLOT_NMASK) : 1u;
                            recs[n_seed++] = make_uint4(s.z, s.w, occ, y);
                        }
                    };
                    for (int32_t i = 0; i < qlen; ++i) st.step(sh_nt4(a.bases[o_beg + i]), (uint32_t)i, emit);
                    if (qlen > 0) st.finish(emit);
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
                    backtrack_large(S, n_a, a.P, n_u, best);
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
                a.work_defer[li] = r | mode << 31;
            }
        }
        uint64_t mh = __ballot(host);
        if (lane == 0 && mh) atomicAdd(&a.ctr->n_host, (uint32_t)__popcll(mh));
    }
}

// ------------------------------------------------------------------------------------------------
// host side: context
// ------------------------------------------------------------------------------------------------
struct sh_ctx {
    const sh_index *idx = nullptr;
    sh_opts opts{};
    ChainParams P{};
    uint64_t max_reads = 0, max_bases = 0;
    uint32_t max_read_len = 0, seed_cap = 32, lds_words = 0;
    bool use_k1 = true;
    uint4 *d_records = nullptr;
    uint32_t *d_k1info = nullptr, *d_work_small = nullptr, *d_work_resketch = nullptr, *d_work_defer = nullptr, *d_work_defer2 = nullptr;
    uint32_t *d_work_large[N_BUCKETS] = {};
    Counters *d_ctr = nullptr;
    Counters *h_ctr = nullptr;     // pinned
    uint8_t *d_arena = nullptr;
    uint64_t arena_bytes = 0;
    hipEvent_t ev[5] = {};
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
}

static bool w_supported(int w) { return w == 5 || w == 10 || w == 11 || w == 19; }

extern "C" sh_status sh_ctx_create(const sh_index *idx, const sh_opts *opts, uint64_t max_reads, uint64_t max_bases,
                                   uint32_t max_read_len, sh_ctx **out)
{
    SH_CHECK(idx && opts && out, SH_ERR_BAD_ARG, "sh_ctx_create: null argument");
    SH_CHECK(opts->k == idx->k && opts->w == idx->w, SH_ERR_BAD_ARG, "sh_ctx_create: opts (k=%d,w=%d) do not match index (k=%d,w=%d)", opts->k, opts->w, idx->k, idx->w);
    SH_CHECK(max_reads > 0 && max_reads < (1ULL << 31), SH_ERR_BAD_ARG, "sh_ctx_create: max_reads must be in [1, 2^31)");
    SH_HIP(hipSetDevice(idx->device));
    sh_ctx *c = new sh_ctx();
    c->idx = idx; c->opts = *opts; c->max_reads = max_reads; c->max_bases = max_bases; c->max_read_len = max_read_len;
    int32_t mid_occ = opts->mid_occ > 0 ? opts->mid_occ : idx->mid_occ;
    fill_chain_params(*opts, mid_occ, c->P);
    // K1 stages a tile of 64 reads in LDS; it needs k <= 23 (hash and position share 64 bits),
    // read positions < 2^17 and a supported compile-time window
    c->use_k1 = opts->k <= 23 && max_read_len <= 1024 && w_supported(opts->w);
    uint64_t tile_bytes = (uint64_t)64 * max_read_len + 32;
    c->lds_words = (uint32_t)((tile_bytes + 15) / 16);
    const uint64_t n_tiles = (max_reads + 63) / 64;
    auto fail = [&](hipError_t e, const char *what) {
        sh_set_error("sh_ctx_create: %s: %s", what, hipGetErrorString(e));
        sh_ctx_destroy(c);
        return e == hipErrorOutOfMemory ? SH_ERR_OOM : SH_ERR_HIP;
    };
    hipError_t e;
    if ((e = hipMalloc(&c->d_records, n_tiles * c->seed_cap * 64 * sizeof(uint4))) != hipSuccess) return fail(e, "records");
    if ((e = hipMalloc(&c->d_k1info, max_reads * 4)) != hipSuccess) return fail(e, "k1info");
    if ((e = hipMalloc(&c->d_work_small, max_reads * 4)) != hipSuccess) return fail(e, "work_small");
    if ((e = hipMalloc(&c->d_work_resketch, max_reads * 4)) != hipSuccess) return fail(e, "work_resketch");
    if ((e = hipMalloc(&c->d_work_defer, max_reads * 4)) != hipSuccess) return fail(e, "work_defer");
    if ((e = hipMalloc(&c->d_work_defer2, max_reads * 4)) != hipSuccess) return fail(e, "work_defer2");
    for (int b = 0; b < N_BUCKETS; ++b)
        if ((e = hipMalloc(&c->d_work_large[b], max_reads * 4)) != hipSuccess) return fail(e, "work_large");
    if ((e = hipMalloc(&c->d_ctr, sizeof(Counters))) != hipSuccess) return fail(e, "counters");
    if ((e = hipHostMalloc(&c->h_ctr, sizeof(Counters))) != hipSuccess) return fail(e, "pinned counters");
    // arena: anchors of repeat reads; 2 KiB per read of the batch, at least 256 MiB (288 GB of HBM to size against)
    c->arena_bytes = std::max<uint64_t>(256ull << 20, max_reads * 2048ull);
    if (const char *env = getenv("SCRUBBY_HIP_ARENA_MB")) c->arena_bytes = (uint64_t)atoll(env) << 20;
    if ((e = hipMalloc(&c->d_arena, c->arena_bytes)) != hipSuccess) return fail(e, "arena");
    for (auto &ev : c->ev) if ((e = hipEventCreate(&ev)) != hipSuccess) return fail(e, "event");
    *out = c;
    return SH_OK;
}

extern "C" sh_status sh_ctx_destroy(sh_ctx *c)
{
    if (!c) return SH_OK;
    hipFree(c->d_records); hipFree(c->d_k1info); hipFree(c->d_work_small); hipFree(c->d_work_resketch);
    hipFree(c->d_work_defer); hipFree(c->d_work_defer2);
    for (auto p : c->d_work_large) hipFree(p);
    hipFree(c->d_ctr); if (c->h_ctr) hipHostFree(c->h_ctr); hipFree(c->d_arena);
    for (auto ev : c->ev) if (ev) hipEventDestroy(ev);
    delete c;
    return SH_OK;
}

template <int W>
static void launch_k1(const K1Args &a, uint32_t n_tiles, size_t lds, hipStream_t s)
{
    hipLaunchKernelGGL(k_sketch_probe<W>, dim3(n_tiles), dim3(64), lds, s, a);
}

static sh_status classify_chunk(sh_ctx *c, const uint8_t *d_bases, const uint64_t *d_offsets, uint64_t n_reads, uint64_t n_bases,
                                uint8_t *d_flags, sh_trace *d_trace, hipStream_t s, sh_stats *stats)
{
    const sh_index *idx = c->idx;
    SH_HIP(hipMemsetAsync(c->d_ctr, 0, sizeof(Counters), s));
    SH_HIP(hipEventRecord(c->ev[0], s));
    const uint32_t n_tiles = (uint32_t)((n_reads + 63) / 64);
    if (c->use_k1) {
        K1Args a{};
        a.bases = d_bases; a.offsets = d_offsets; a.n_reads = n_reads; a.n_bases = n_bases;
        a.slots = (const uint4 *)idx->d_slots; a.lg_slots = idx->lg_slots; a.k = idx->k;
        a.records = c->d_records; a.seed_cap = c->seed_cap;
        a.k1info = c->d_k1info; a.flags = d_flags; a.trace = d_trace;
        a.work_small = c->d_work_small; a.work_resketch = c->d_work_resketch; a.ctr = c->d_ctr;
        a.lds_words = c->lds_words;
        size_t lds = (size_t)K1_LIST_CAP * 64 * 8 + ((size_t)c->lds_words + 2) * 4 + (((size_t)c->lds_words + 2) * 2 + 3) / 4 * 4;
        switch (idx->w) {
        case 5: launch_k1<5>(a, n_tiles, lds, s); break;
        case 10: launch_k1<10>(a, n_tiles, lds, s); break;
        case 11: launch_k1<11>(a, n_tiles, lds, s); break;
        case 19: launch_k1<19>(a, n_tiles, lds, s); break;
        default: sh_set_error("unsupported w"); return SH_ERR_BAD_ARG;
        }
    } else {
        hipLaunchKernelGGL(k_route_all, dim3((uint32_t)((n_reads + 255) / 256)), dim3(256), 0, s, n_reads, c->d_work_resketch, c->d_ctr);
    }
    SH_HIP(hipEventRecord(c->ev[1], s));

    K2Args b{};
    b.offsets = d_offsets; b.bases = d_bases; b.n_reads = n_reads;
    b.slots = (const uint4 *)idx->d_slots; b.lg_slots = idx->lg_slots; b.w = idx->w;
    b.positions = idx->d_positions;
    b.records = c->d_records; b.seed_cap = c->seed_cap;
    b.k1info = c->d_k1info; b.flags = d_flags; b.trace = d_trace;
    for (int i = 0; i < N_BUCKETS; ++i) b.work_large[i] = c->d_work_large[i];
    b.work_defer = c->d_work_defer; b.ctr = c->d_ctr;
    b.arena = c->d_arena; b.arena_bytes = c->arena_bytes;
    b.P = c->P;
    const uint32_t grid = std::min<uint32_t>(n_tiles, 256 * 8);
    if (c->use_k1) {
        b.work = c->d_work_small; b.work_count = &c->d_ctr->n_small; b.mode = 0;
        hipLaunchKernelGGL(k_chain_small<K2_CAP>, dim3(grid), dim3(64), 0, s, b);
    }
    SH_HIP(hipEventRecord(c->ev[2], s));
    for (int i = 0; i < N_BUCKETS; ++i) {
        b.work = c->d_work_large[i]; b.work_count = &c->d_ctr->n_large[i]; b.mode = 0;
        hipLaunchKernelGGL(k_chain_large, dim3(grid), dim3(64), 0, s, b);
    }
    b.work = c->d_work_resketch; b.work_count = &c->d_ctr->n_resketch; b.mode = 1;
    hipLaunchKernelGGL(k_chain_large, dim3(grid), dim3(64), 0, s, b);
    SH_HIP(hipEventRecord(c->ev[3], s));
    SH_HIP(hipMemcpyAsync(c->h_ctr, c->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, s));
    SH_HIP(hipStreamSynchronize(s));
    SH_HIP(hipGetLastError());

    Counters first = *c->h_ctr;
    // reads that did not get arena space: rerun them with a fresh arena until none is left
    uint32_t n_defer = c->h_ctr->n_defer;
    uint32_t n_resk_left = 0;
    int rounds = 0;
    while (n_defer > 0) {
        SH_CHECK(++rounds < 64, SH_ERR_OOM, "chain arena (%llu MiB) too small for a single read; set SCRUBBY_HIP_ARENA_MB", (unsigned long long)(c->arena_bytes >> 20));
        std::swap(c->d_work_defer, c->d_work_defer2);
        Counters z = *c->h_ctr;
        z.n_defer = 0; z.arena_cursor = 0; z.n_resketch = n_defer;
        SH_HIP(hipMemcpyAsync(c->d_ctr, &z, sizeof(Counters), hipMemcpyHostToDevice, s));
        b.work = c->d_work_defer2; b.work_count = &c->d_ctr->n_resketch; b.mode = 2; b.work_defer = c->d_work_defer;
        hipLaunchKernelGGL(k_chain_large, dim3(grid), dim3(64), 0, s, b);
        SH_HIP(hipMemcpyAsync(c->h_ctr, c->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, s));
        SH_HIP(hipStreamSynchronize(s));
        uint32_t nd = c->h_ctr->n_defer;
        SH_CHECK(nd < n_defer, SH_ERR_OOM, "chain arena (%llu MiB) too small; set SCRUBBY_HIP_ARENA_MB", (unsigned long long)(c->arena_bytes >> 20));
        n_defer = nd;
        (void)n_resk_left;
    }
    SH_HIP(hipEventRecord(c->ev[4], s));
    SH_HIP(hipEventSynchronize(c->ev[4]));
    if (stats) {
        float t01 = 0, t12 = 0, t23 = 0, t04 = 0;
        hipEventElapsedTime(&t01, c->ev[0], c->ev[1]); hipEventElapsedTime(&t12, c->ev[1], c->ev[2]);
        hipEventElapsedTime(&t23, c->ev[2], c->ev[3]); hipEventElapsedTime(&t04, c->ev[0], c->ev[4]);
        stats->n_reads += n_reads; stats->n_bases += n_bases;
        stats->n_host += c->h_ctr->n_host; stats->n_no_seed += first.n_noseed;
        uint64_t nl = first.n_resketch;
        for (int i = 0; i < N_BUCKETS; ++i) nl += first.n_large[i];
        stats->n_chain_large += nl; stats->n_chain_small += first.n_small - (nl - first.n_resketch);
        stats->n_minimizers += first.sum_mini;
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
    for (uint64_t r0 = 0; r0 < n_reads; r0 += c->max_reads) {
        uint64_t n = std::min<uint64_t>(c->max_reads, n_reads - r0);
        sh_status st = classify_chunk(c, d_bases, d_offsets + r0, n, n_bases, d_flags + r0, d_trace ? d_trace + r0 : nullptr, s, stats);
        if (st != SH_OK) return st;
    }
    return SH_OK;
}
