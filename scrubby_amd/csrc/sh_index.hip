// sh_index.hip — minimizer index construction on gfx950 and its HBM layout.
//
// Replaces  Aligner::builder().<preset>().with_cigar().with_index_threads(t).with_index(path, None)
//           /root/reference/src/cleaner.rs:472-482   (minimap2's mm_idx_gen behind the crate)
// The logical content is minimap2's: minimizer hash -> ascending list of
// (rid<<32 | pos<<1 | strand).  The physical layout is ours: one open-addressing table of 16-B slots
// (singletons inline, so the common probe is ONE 16-B gather) plus one position array for repeated
// minimizers; both stay resident in HBM (CHM13v2: ~8 GiB table + positions, of 288 GB).
//
// Build pipeline, all on the device:
//   1 k_ref_sketch<count>   one lane per 1024-bp segment (+ w+k warm-up), window in VGPRs
//   2 exclusive scan -> k_ref_sketch<emit> writes (hash, position) pairs in reference order
//   3 rocPRIM radix sort by hash (stable: positions stay ascending per hash)
//   4 run-length encode -> distinct hashes + counts; select -> position array of repeated hashes
//   5 k_table_fill          atomicCAS insertion, linear probing
// rocPRIM supplies the generic sort / scan / RLE / select primitives (the index build is not the
// timed hot path); sketching and the table are hand-written.
#include "sh_common.h"
#include "sh_sketch.h"
#include <rocprim/rocprim.hpp>
#include <chrono>
#include <atomic>
#include <thread>
#include <memory>
#include <algorithm>
#include <fcntl.h>
#include <unistd.h>
#include <sys/stat.h>
#include <zlib.h>

#define SEG_LEN 1024u

struct RefArgs {
    const uint8_t *bases;
    const uint64_t *contig_start;   // device, n_contigs+1
    const uint64_t *seg_first;      // device, n_contigs+1: first segment id of each contig
    uint32_t n_contigs; uint64_t n_segs; int32_t k;
    uint32_t *counts;               // per segment (count pass)
    const uint64_t *seg_off;        // per segment (emit pass)
    uint64_t *keys, *vals;
};

template <int W, bool EMIT>
__global__ __launch_bounds__(256) void k_ref_sketch(RefArgs a)
{
    const uint64_t seg = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (seg >= a.n_segs) return;
    uint32_t lo = 0, hi = a.n_contigs;
    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (a.seg_first[mid] <= seg) lo = mid; else hi = mid; }
    const uint32_t rid = lo;
    const uint64_t cs = a.contig_start[rid], clen = a.contig_start[rid + 1] - cs;
    const uint64_t start = (seg - a.seg_first[rid]) * SEG_LEN;
    const uint64_t end = start + SEG_LEN < clen ? start + SEG_LEN : clen;
    const uint64_t warm = (uint64_t)(W + a.k);
    const uint64_t from = start > warm ? start - warm : 0;
    const uint8_t *seq = a.bases + cs;

    SketchState<W> st;
    st.init(a.k);
    uint64_t n = 0, cur = 0;
    const uint64_t o0 = EMIT ? a.seg_off[seg] : 0;
    auto emit = [&](uint64_t x, uint32_t y) {
        if (cur < start) return;            // pushes made during warm-up belong to the previous segment
        if (EMIT) { a.keys[o0 + n] = x >> 8; a.vals[o0 + n] = (uint64_t)rid << 32 | y; }
        ++n;
    };
    for (uint64_t i0 = from; i0 < end; i0 += W) {
        auto one = [&](auto Pc) {
            constexpr int P = decltype(Pc)::value;
            const uint64_t i = i0 + P;
            if (i < end) { cur = i; st.template step<P>(sh_nt4(seq[i]), (uint32_t)i, emit); }
        };
        [&]<int... Ps>(std::integer_sequence<int, Ps...>) { (one(std::integral_constant<int, Ps>{}), ...); }
        (std::make_integer_sequence<int, W>{});
    }
    if (end == clen) { cur = end; st.finish(emit); }
    if (!EMIT) a.counts[seg] = (uint32_t)n;
}

__global__ void k_multi_flags(const uint64_t *keys, uint64_t n, uint8_t *flags)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t k = keys[i];
    bool m = (i > 0 && keys[i - 1] == k) || (i + 1 < n && keys[i + 1] == k);
    flags[i] = m;
}

__global__ void k_multi_counts(const uint32_t *counts, uint64_t n, uint64_t *mc)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) mc[i] = counts[i] > 1 ? counts[i] : 0;
}

__global__ void k_table_fill(const uint64_t *ukeys, const uint32_t *counts, const uint64_t *start, const uint64_t *moff,
                             const uint64_t *vals, uint64_t n_unique, unsigned long long *slots, uint32_t lg, int *err)
{
    uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n_unique) return;
    const uint64_t key = ukeys[u];
    const uint32_t c = counts[u];
    unsigned long long w0, w1;
    if (c == 1) { w0 = key; w1 = vals[start[u]]; }
    else {
        if (c > SH_SLOT_NMASK) { *err = 1; return; }
        w0 = key | SH_SLOT_MULTI; w1 = (unsigned long long)moff[u] << SH_SLOT_NBITS | c;
    }
    const uint64_t mask = (1ULL << lg) - 1;
    uint64_t h = sh_slot_home(key, lg);
    for (;;) {
        unsigned long long old = atomicCAS(&slots[2 * h], (unsigned long long)SH_SLOT_EMPTY, w0);
        if (old == SH_SLOT_EMPTY) { slots[2 * h + 1] = w1; return; }
        h = (h + 1) & mask;
    }
}

template <int W>
static void launch_ref_sketch(bool emit, const RefArgs &a, hipStream_t s)
{
    dim3 g((uint32_t)((a.n_segs + 255) / 256)), b(256);
    if (emit) hipLaunchKernelGGL((k_ref_sketch<W, true>), g, b, 0, s, a);
    else hipLaunchKernelGGL((k_ref_sketch<W, false>), g, b, 0, s, a);
}

static sh_status dispatch_ref_sketch(int w, bool emit, const RefArgs &a, hipStream_t s)
{
    switch (w) {
    case 5: launch_ref_sketch<5>(emit, a, s); break;
    case 10: launch_ref_sketch<10>(emit, a, s); break;
    case 11: launch_ref_sketch<11>(emit, a, s); break;
    case 19: launch_ref_sketch<19>(emit, a, s); break;
    default: sh_set_error("unsupported minimizer window w=%d (supported: 5, 10, 11, 19)", w); return SH_ERR_BAD_ARG;
    }
    return SH_OK;
}

// mi->S of minimap2: the reference as 4-bit nt4 codes, what mm_idx_getseq hands the base-level alignment (sh_align.h).
// One thread per 16 output bytes = 32 bases.
__global__ void k_ref_pack(const uint8_t *bases, uint64_t n, uint8_t *packed)
{
    const uint64_t o = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    if (o * 2 >= n) return;
    uint32_t w[4] = {0, 0, 0, 0};
    for (uint32_t j = 0; j < 32; ++j) {
        const uint64_t g = o * 2 + j;
        const uint32_t c = g < n ? sh_nt4(bases[g]) : 0u;
        w[j >> 3] |= c << (4 * (j & 7));
    }
    *(uint4 *)(packed + o) = make_uint4(w[0], w[1], w[2], w[3]);
}

struct DevBuf {   // frees on scope exit
    void *p = nullptr;
    ~DevBuf() { if (p) hipFree(p); }
    template <class T> T *as() { return (T *)p; }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
};

sh_status shi_index_build_device(const uint8_t *d_bases, const uint64_t *contig_starts, uint32_t n_contigs,
                                 const sh_opts *opts, int32_t device, hipStream_t s, sh_index **out)
{
    SH_CHECK(d_bases && contig_starts && opts && out && n_contigs > 0, SH_ERR_BAD_ARG, "sh_index_build: null/empty argument");
    SH_CHECK(opts->k > 0 && opts->k <= 28 && (opts->k & 1), SH_ERR_BAD_ARG,
             "k=%d unsupported: k must be odd and <= 28 (every minimap2 preset is; odd k has no strand-ambiguous k-mers)", opts->k);
    SH_CHECK(opts->w > 0 && opts->w < 256, SH_ERR_BAD_ARG, "w=%d out of range", opts->w);
    SH_HIP(hipSetDevice(device));
    auto t0 = std::chrono::steady_clock::now();

    std::vector<uint64_t> seg_first(n_contigs + 1, 0);
    for (uint32_t i = 0; i < n_contigs; ++i) {
        uint64_t len = contig_starts[i + 1] - contig_starts[i];
        SH_CHECK(len < (1ULL << 31) - 1, SH_ERR_BAD_ARG, "contig %u has %llu bases; limit is 2^31-2", i, (unsigned long long)len);
        seg_first[i + 1] = seg_first[i] + (len + SEG_LEN - 1) / SEG_LEN;
    }
    const uint64_t n_segs = seg_first[n_contigs];
    const uint64_t n_bases = contig_starts[n_contigs] - contig_starts[0];

    DevBuf b_cs, b_sf, b_cnt, b_off, b_tmp;
    SH_HIP(b_cs.alloc((n_contigs + 1) * 8)); SH_HIP(b_sf.alloc((n_contigs + 1) * 8));
    SH_HIP(b_cnt.alloc((n_segs + 1) * 4)); SH_HIP(b_off.alloc((n_segs + 1) * 8));
    SH_HIP(hipMemcpyAsync(b_cs.p, contig_starts, (n_contigs + 1) * 8, hipMemcpyHostToDevice, s));
    SH_HIP(hipMemcpyAsync(b_sf.p, seg_first.data(), (n_contigs + 1) * 8, hipMemcpyHostToDevice, s));
    SH_HIP(hipMemsetAsync(b_cnt.p, 0, (n_segs + 1) * 4, s));

    RefArgs ra{};
    ra.bases = d_bases; ra.contig_start = b_cs.as<uint64_t>(); ra.seg_first = b_sf.as<uint64_t>();
    ra.n_contigs = n_contigs; ra.n_segs = n_segs; ra.k = opts->k;
    ra.counts = b_cnt.as<uint32_t>();
    sh_status st = dispatch_ref_sketch(opts->w, false, ra, s);
    if (st != SH_OK) return st;

    // exclusive scan over n_segs+1 entries: last element = total
    size_t tmp_bytes = 0;
    SH_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, b_cnt.as<uint32_t>(), b_off.as<uint64_t>(), (uint64_t)0, n_segs + 1, rocprim::plus<uint64_t>(), s));
    SH_HIP(b_tmp.alloc(tmp_bytes));
    SH_HIP(rocprim::exclusive_scan(b_tmp.p, tmp_bytes, b_cnt.as<uint32_t>(), b_off.as<uint64_t>(), (uint64_t)0, n_segs + 1, rocprim::plus<uint64_t>(), s));
    uint64_t n_mini = 0;
    SH_HIP(hipMemcpyAsync(&n_mini, b_off.as<uint64_t>() + n_segs, 8, hipMemcpyDeviceToHost, s));
    SH_HIP(hipStreamSynchronize(s));

    DevBuf b_k1, b_v1, b_k2, b_v2;
    SH_HIP(b_k1.alloc(n_mini * 8)); SH_HIP(b_v1.alloc(n_mini * 8)); SH_HIP(b_k2.alloc(n_mini * 8)); SH_HIP(b_v2.alloc(n_mini * 8));
    ra.seg_off = b_off.as<uint64_t>(); ra.keys = b_k1.as<uint64_t>(); ra.vals = b_v1.as<uint64_t>();
    st = dispatch_ref_sketch(opts->w, true, ra, s);
    if (st != SH_OK) return st;

    uint64_t *keys = b_k2.as<uint64_t>(), *vals = b_v2.as<uint64_t>();
    if (n_mini > 0) {
        DevBuf b_st;
        size_t sb = 0;
        SH_HIP(rocprim::radix_sort_pairs(nullptr, sb, b_k1.as<uint64_t>(), keys, b_v1.as<uint64_t>(), vals, n_mini, 0, 2 * opts->k, s));
        SH_HIP(b_st.alloc(sb));
        SH_HIP(rocprim::radix_sort_pairs(b_st.p, sb, b_k1.as<uint64_t>(), keys, b_v1.as<uint64_t>(), vals, n_mini, 0, 2 * opts->k, s));
        SH_HIP(hipStreamSynchronize(s));
    }
    // k1/v1 are free for reuse from here: ukeys in k1, start/moff in v1 + extra
    hipFree(b_k1.p); b_k1.p = nullptr; hipFree(b_v1.p); b_v1.p = nullptr;

    DevBuf b_uk, b_uc, b_nr, b_start, b_mc, b_moff, b_flags, b_npos;
    SH_HIP(b_uk.alloc(n_mini * 8)); SH_HIP(b_uc.alloc(n_mini * 4 + 4)); SH_HIP(b_nr.alloc(16)); SH_HIP(b_npos.alloc(16));
    uint64_t n_unique = 0, n_pos = 0;
    sh_index *idx = new sh_index();
    idx->device = device; idx->k = opts->k; idx->w = opts->w; idx->n_contigs = n_contigs; idx->n_bases = n_bases;
    idx->n_minimizers = n_mini;
    for (uint32_t i = 0; i < n_contigs; ++i) idx->contig_len.push_back(contig_starts[i + 1] - contig_starts[i]);
    auto bail = [&](sh_status code) { sh_index_free(idx); return code; };
#define IDX_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { sh_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); return bail(e_ == hipErrorOutOfMemory ? SH_ERR_OOM : SH_ERR_HIP); } } while (0)

    if (n_mini > 0) {
        DevBuf b_t2;
        size_t sb = 0;
        IDX_HIP(rocprim::run_length_encode(nullptr, sb, keys, n_mini, b_uk.as<uint64_t>(), b_uc.as<uint32_t>(), b_nr.as<uint64_t>(), s));
        IDX_HIP(b_t2.alloc(sb));
        IDX_HIP(rocprim::run_length_encode(b_t2.p, sb, keys, n_mini, b_uk.as<uint64_t>(), b_uc.as<uint32_t>(), b_nr.as<uint64_t>(), s));
        IDX_HIP(hipMemcpyAsync(&n_unique, b_nr.p, 8, hipMemcpyDeviceToHost, s));
        IDX_HIP(hipStreamSynchronize(s));
    }
    uint32_t lg = 4;
    while ((1ULL << lg) < 2 * n_unique + 1) ++lg;
    idx->n_keys = n_unique; idx->lg_slots = lg; idx->n_slots = 1ULL << lg;
    IDX_HIP(hipMalloc(&idx->d_slots, idx->n_slots * 16));
    IDX_HIP(hipMemsetAsync(idx->d_slots, 0xff, idx->n_slots * 16, s));

    if (n_unique > 0) {
        const uint32_t gU = (uint32_t)((n_unique + 255) / 256), gM = (uint32_t)((n_mini + 255) / 256);
        IDX_HIP(b_start.alloc(n_unique * 8)); IDX_HIP(b_mc.alloc(n_unique * 8)); IDX_HIP(b_moff.alloc(n_unique * 8));
        IDX_HIP(b_flags.alloc(n_mini));
        hipLaunchKernelGGL(k_multi_counts, dim3(gU), dim3(256), 0, s, b_uc.as<uint32_t>(), n_unique, b_mc.as<uint64_t>());
        hipLaunchKernelGGL(k_multi_flags, dim3(gM), dim3(256), 0, s, keys, n_mini, b_flags.as<uint8_t>());
        {
            DevBuf t;
            size_t sb = 0;
            IDX_HIP(rocprim::exclusive_scan(nullptr, sb, b_uc.as<uint32_t>(), b_start.as<uint64_t>(), (uint64_t)0, n_unique, rocprim::plus<uint64_t>(), s));
            IDX_HIP(t.alloc(sb));
            IDX_HIP(rocprim::exclusive_scan(t.p, sb, b_uc.as<uint32_t>(), b_start.as<uint64_t>(), (uint64_t)0, n_unique, rocprim::plus<uint64_t>(), s));
            IDX_HIP(hipStreamSynchronize(s));
        }
        {
            DevBuf t;
            size_t sb = 0;
            IDX_HIP(rocprim::exclusive_scan(nullptr, sb, b_mc.as<uint64_t>(), b_moff.as<uint64_t>(), (uint64_t)0, n_unique, rocprim::plus<uint64_t>(), s));
            IDX_HIP(t.alloc(sb));
            IDX_HIP(rocprim::exclusive_scan(t.p, sb, b_mc.as<uint64_t>(), b_moff.as<uint64_t>(), (uint64_t)0, n_unique, rocprim::plus<uint64_t>(), s));
            IDX_HIP(hipStreamSynchronize(s));
        }
        // positions of repeated minimizers, in sorted (hash, position) order
        IDX_HIP(hipMalloc(&idx->d_positions, (n_mini + 2) * 8));     // upper bound; shrunk below
        {
            DevBuf t;
            size_t sb = 0;
            IDX_HIP(rocprim::select(nullptr, sb, vals, b_flags.as<uint8_t>(), idx->d_positions, b_npos.as<uint64_t>(), n_mini, s));
            IDX_HIP(t.alloc(sb));
            IDX_HIP(rocprim::select(t.p, sb, vals, b_flags.as<uint8_t>(), idx->d_positions, b_npos.as<uint64_t>(), n_mini, s));
            IDX_HIP(hipMemcpyAsync(&n_pos, b_npos.p, 8, hipMemcpyDeviceToHost, s));
            IDX_HIP(hipStreamSynchronize(s));
        }
        idx->n_positions = n_pos;
        if (n_pos + 2 < n_mini / 2) {      // most minimizers are singletons: give the slack back
            uint64_t *small = nullptr;
            IDX_HIP(hipMalloc(&small, (n_pos + 2) * 8));
            IDX_HIP(hipMemcpyAsync(small, idx->d_positions, n_pos * 8, hipMemcpyDeviceToDevice, s));
            IDX_HIP(hipStreamSynchronize(s));
            hipFree(idx->d_positions);
            idx->d_positions = small;
        }
        DevBuf b_err;
        IDX_HIP(b_err.alloc(4));
        IDX_HIP(hipMemsetAsync(b_err.p, 0, 4, s));
        hipLaunchKernelGGL(k_table_fill, dim3(gU), dim3(256), 0, s, b_uk.as<uint64_t>(), b_uc.as<uint32_t>(), b_start.as<uint64_t>(),
                           b_moff.as<uint64_t>(), vals, n_unique, (unsigned long long *)idx->d_slots, lg, b_err.as<int>());
        int err = 0;
        IDX_HIP(hipMemcpyAsync(&err, b_err.p, 4, hipMemcpyDeviceToHost, s));
        IDX_HIP(hipStreamSynchronize(s));
        if (err) { sh_set_error("a minimizer occurs more than 2^28 times"); return bail(SH_ERR_INDEX); }
    } else {
        IDX_HIP(hipMalloc(&idx->d_positions, 16));
    }

    {   // the reference itself stays in HBM for the extension stage (half a byte per base)
        const uint64_t pbytes = ((n_bases + 31) / 32) * 16 + 16;
        IDX_HIP(hipMalloc(&idx->d_ref, pbytes));
        IDX_HIP(hipMalloc(&idx->d_cstart, (n_contigs + 1) * 8));
        std::vector<uint64_t> rel(n_contigs + 1);
        for (uint32_t i = 0; i <= n_contigs; ++i) rel[i] = contig_starts[i] - contig_starts[0];
        IDX_HIP(hipMemcpyAsync(idx->d_cstart, rel.data(), (n_contigs + 1) * 8, hipMemcpyHostToDevice, s));
        IDX_HIP(hipMemsetAsync(idx->d_ref + pbytes - 16, 0, 16, s));
        if (n_bases) hipLaunchKernelGGL(k_ref_pack, dim3((uint32_t)((n_bases + 32 * 256 - 1) / (32 * 256))), dim3(256), 0, s, d_bases + contig_starts[0], n_bases, idx->d_ref);
        IDX_HIP(hipStreamSynchronize(s));
    }
    // mm_mapopt_update: mid_occ from the occurrence distribution unless the preset fixes it
    idx->o_mid_occ = opts->mid_occ; idx->o_min_mid_occ = opts->min_mid_occ; idx->o_max_mid_occ = opts->max_mid_occ; idx->o_mid_occ_frac = opts->mid_occ_frac;
    idx->mid_occ = opts->mid_occ;
    if (opts->mid_occ <= 0) {
        int32_t mo = INT32_MAX;
        if (opts->mid_occ_frac > 0.f && n_unique > 0) {
            DevBuf b_sorted, t;
            size_t sb = 0;
            IDX_HIP(b_sorted.alloc(n_unique * 4));
            IDX_HIP(rocprim::radix_sort_keys(nullptr, sb, b_uc.as<uint32_t>(), b_sorted.as<uint32_t>(), n_unique, 0, 32, s));
            IDX_HIP(t.alloc(sb));
            IDX_HIP(rocprim::radix_sort_keys(t.p, sb, b_uc.as<uint32_t>(), b_sorted.as<uint32_t>(), n_unique, 0, 32, s));
            uint32_t kth = (uint32_t)((1. - opts->mid_occ_frac) * n_unique), v = 0;
            IDX_HIP(hipMemcpyAsync(&v, b_sorted.as<uint32_t>() + kth, 4, hipMemcpyDeviceToHost, s));
            IDX_HIP(hipStreamSynchronize(s));
            mo = (int32_t)(v + 1);
        }
        if (mo < opts->min_mid_occ) mo = opts->min_mid_occ;
        if (opts->max_mid_occ > opts->min_mid_occ && mo > opts->max_mid_occ) mo = opts->max_mid_occ;
        idx->mid_occ = mo;
    }
    IDX_HIP(hipStreamSynchronize(s));
    IDX_HIP(hipGetLastError());
    idx->build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    *out = idx;
    return SH_OK;
#undef IDX_HIP
}

extern "C" sh_status sh_index_build_device(const uint8_t *d_bases, const uint64_t *contig_starts, uint32_t n_contigs,
                                           const sh_opts *opts, int32_t device, void *stream, sh_index **out)
{
    return shi_index_build_device(d_bases, contig_starts, n_contigs, opts, device, (hipStream_t)stream, out);
}

// ---- FASTA -> bases, on the GPU ----------------------------------------------------------------------------------
// The reference hands the FASTA path to minimap2 (cleaner.rs:475-479), which reads it line by line.  Here the raw file
// bytes go to HBM as they are (parallel preads, each worker uploading its own blocks) and the text is taken apart on the
// device: a 3-state line machine (0 = at a line start, 1 = in a sequence line, 2 = in a header line) is run over all
// bytes at once as an inclusive scan of per-byte transition functions under composition (associative, so rocPRIM's
// device scan applies); a byte is a base iff the state after it is 1 and it is not a line-end '\r'; a header starts
// where '>' meets state 0.  A block-wise compaction then writes the bases and records each header's rank among them.
// 3.1 GB of FASTA: 0.6 s instead of 2.4 s through the line reader (which stays for FASTQ references and odd files).
#define FA_NL 0u        // f(s) packed 2 bits per state: f(0) | f(1) << 2 | f(2) << 4
#define FA_GT 38u       // 0 -> 2, 1 -> 1, 2 -> 2
#define FA_OTHER 37u    // 0 -> 1, 1 -> 1, 2 -> 2
struct FaCompose {      // a first, then b
    __host__ __device__ uint8_t operator()(uint8_t a, uint8_t b) const
    {
        return (uint8_t)(((b >> (2 * (a & 3u))) & 3u) | (((b >> (2 * ((a >> 2) & 3u))) & 3u) << 2) | (((b >> (2 * ((a >> 4) & 3u))) & 3u) << 4));
    }
};
#define FA_TPB 256
#define FA_PER 32
#define FA_BLK (FA_TPB * FA_PER)

__global__ void k_fa_code(const uint8_t *raw, uint64_t n, uint8_t *code)
{
    const uint64_t i0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    if (i0 >= n) return;
    uint8_t b[16], c[16];
    if (i0 + 16 <= n) *(uint4 *)b = *(const uint4 *)(raw + i0);
    else for (int j = 0; j < 16; ++j) b[j] = i0 + j < n ? raw[i0 + j] : (uint8_t)'\n';
#pragma unroll
    for (int j = 0; j < 16; ++j) c[j] = b[j] == '\n' ? FA_NL : (b[j] == '>' ? FA_GT : FA_OTHER);
    if (i0 + 16 <= n) *(uint4 *)(code + i0) = *(uint4 *)c;
    else for (int j = 0; j < 16 && i0 + j < n; ++j) code[i0 + j] = c[j];
}

// first code of a scan piece absorbs the scanned value just before it
__global__ void k_fa_carry(uint8_t *code, const uint8_t *scanned_prev) { code[0] = FaCompose()(scanned_prev[0], code[0]); }

__device__ inline bool fa_keep(const uint8_t *raw, const uint8_t *st, uint64_t i, uint64_t n)
{
    if ((st[i] & 3u) != 1u) return false;           // the machine starts in state 0: the state after byte i is F_i(0)
    if (raw[i] != '\r') return true;
    uint64_t j = i + 1;                               // a run of '\r' that ends the line (or the file) is line ending, not sequence
    while (j < n && raw[j] == '\r') ++j;
    return !(j == n || raw[j] == '\n');
}

__global__ __launch_bounds__(FA_TPB) void k_fa_count(const uint8_t *raw, const uint8_t *st, uint64_t n, uint32_t *blk_cnt)
{
    __shared__ uint32_t red[FA_TPB / 64];
    const uint64_t i0 = (uint64_t)blockIdx.x * FA_BLK + (uint64_t)threadIdx.x * FA_PER;
    uint32_t c = 0;
    for (int j = 0; j < FA_PER; ++j) if (i0 + j < n) c += fa_keep(raw, st, i0 + j, n);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += (uint32_t)__shfl_xor((int)c, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t t = 0; for (int w = 0; w < FA_TPB / 64; ++w) t += red[w]; blk_cnt[blockIdx.x] = t; }
}

__global__ __launch_bounds__(FA_TPB) void k_fa_scatter(const uint8_t *raw, const uint8_t *st, uint64_t n, const uint64_t *blk_off, uint8_t *bases,
                                                        unsigned long long *hdr_pos, unsigned long long *hdr_rank, uint32_t hdr_cap, uint32_t *hdr_n)
{
    __shared__ uint32_t wsum[FA_TPB / 64];
    const uint64_t i0 = (uint64_t)blockIdx.x * FA_BLK + (uint64_t)threadIdx.x * FA_PER;
    uint32_t keep = 0;
    for (int j = 0; j < FA_PER; ++j) if (i0 + j < n && fa_keep(raw, st, i0 + j, n)) keep |= 1u << j;
    const uint32_t c = (uint32_t)__popc(keep);
    uint32_t incl = c;      // wave inclusive scan, then the waves of the block
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)incl, o); if ((threadIdx.x & 63) >= (uint32_t)o) incl += v; }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t wbase = 0;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) wbase += wsum[w];
    uint64_t o = blk_off[blockIdx.x] + wbase + incl - c;
    for (int j = 0; j < FA_PER; ++j) {
        const uint64_t i = i0 + j;
        if (i >= n) break;
        if (raw[i] == '>' && (st[i] & 3u) == 2u && (i == 0 || (st[i - 1] & 3u) == 0u)) {      // '>' at a line start
            const uint32_t h = atomicAdd(hdr_n, 1u);
            if (h < hdr_cap) { hdr_pos[h] = i; hdr_rank[h] = o; }
        }
        if (keep >> j & 1u) bases[o++] = raw[i];
    }
}

// Reads `path` (plain: parallel preads; gzip: one inflating reader) into HBM and strips it there.  *handled = false: not a
// FASTA this reader takes (the caller falls back to the line reader).
static sh_status fasta_to_device(const char *path, int32_t device, DevBuf &d_bases, std::vector<uint64_t> &contig_starts, bool *handled)
{
    *handled = false;
    const bool dbg = getenv("SCRUBBY_HIP_DBG_HOST") && *getenv("SCRUBBY_HIP_DBG_HOST") == '1';
    auto tick = [] { return std::chrono::steady_clock::now(); };
    auto msf = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t_a = tick();
    FILE *f = fopen(path, "rb");
    SH_CHECK(f, SH_ERR_IO, "cannot open %s", path);
    unsigned char mg[2] = {0, 0};
    const size_t got = fread(mg, 1, 2, f);
    fclose(f);
    const bool gz = got == 2 && mg[0] == 0x1f && mg[1] == 0x8b;
    SH_HIP(hipSetDevice(device));
    DevBuf d_raw;
    uint64_t n = 0;
    if (!gz) {
        const int fd = open(path, O_RDONLY);
        SH_CHECK(fd >= 0, SH_ERR_IO, "cannot open %s", path);
        struct stat sb;
        if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) { close(fd); return SH_OK; }
        n = (uint64_t)sb.st_size;
        char first[256];
        const ssize_t fl = pread(fd, first, sizeof first, 0);
        ssize_t q = 0;
        while (q < fl && (first[q] == '\n' || first[q] == '\r')) ++q;
        if (fl <= 0 || q >= fl || first[q] != '>') { close(fd); return SH_OK; }
        if (d_raw.alloc(n + 64) != hipSuccess) { close(fd); sh_set_error("out of device memory for %s", path); return SH_ERR_OOM; }
        const uint64_t BLK = 32ull << 20;
        std::atomic<uint64_t> next{0};
        std::atomic<int> bad{0};
        const unsigned T = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        std::vector<std::thread> thr;
        for (unsigned t = 0; t < T; ++t)
            thr.emplace_back([&]() {
                hipStream_t s = nullptr;
                if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { bad = 1; return; }
                std::vector<char> buf(BLK);
                for (;;) {
                    const uint64_t off = next.fetch_add(BLK);
                    if (off >= n || bad.load()) break;
                    const uint64_t len = std::min(BLK, n - off);
                    uint64_t done = 0;
                    while (done < len) { const ssize_t r = pread(fd, buf.data() + done, len - done, (off_t)(off + done)); if (r <= 0) { bad = 1; break; } done += (uint64_t)r; }
                    if (bad.load()) break;
                    if (hipMemcpyAsync(d_raw.as<uint8_t>() + off, buf.data(), len, hipMemcpyHostToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { bad = 1; break; }
                }
                hipStreamDestroy(s);
            });
        for (auto &t : thr) t.join();
        close(fd);
        SH_CHECK(!bad.load(), SH_ERR_IO, "error while reading %s into device memory", path);
    } else {
        gzFile g = gzopen(path, "rb");
        SH_CHECK(g, SH_ERR_IO, "cannot open %s", path);
        gzbuffer(g, 1 << 20);
        const size_t BLK = 64u << 20;
        std::vector<std::unique_ptr<char[]>> blocks;
        std::vector<size_t> lens;
        for (;;) {
            std::unique_ptr<char[]> b(new char[BLK]);
            const int r = gzread(g, b.get(), (unsigned)BLK);
            if (r < 0) { gzclose(g); sh_set_error("read error in %s", path); return SH_ERR_IO; }
            if (r == 0) break;
            n += (uint64_t)r; lens.push_back((size_t)r); blocks.push_back(std::move(b));
        }
        // a short read is only the end of the file if zlib says so: a truncated or corrupt stream is an error, not a short reference
        int zerr = Z_OK;
        gzerror(g, &zerr);
        const int zc = gzclose(g);
        SH_CHECK((zerr == Z_OK || zerr == Z_STREAM_END) && zc == Z_OK, SH_ERR_IO, "%s: gzip stream is truncated or corrupt", path);
        size_t q = 0;
        while (!blocks.empty() && q < lens[0] && (blocks[0][q] == '\n' || blocks[0][q] == '\r')) ++q;
        if (blocks.empty() || q >= lens[0] || blocks[0][q] != '>') return SH_OK;
        SH_CHECK(d_raw.alloc(n + 64) == hipSuccess, SH_ERR_OOM, "out of device memory for %s", path);
        uint64_t off = 0;
        for (size_t i = 0; i < blocks.size(); ++i) { SH_HIP(hipMemcpy(d_raw.as<uint8_t>() + off, blocks[i].get(), lens[i], hipMemcpyHostToDevice)); off += lens[i]; }
    }
    SH_CHECK(n > 0, SH_ERR_INDEX, "no sequences in %s", path);
    const auto t_b = tick();

    hipStream_t s = nullptr;
    DevBuf d_code, d_st, d_cnt, d_off, d_tmp, d_hp, d_hr, d_hn;
    SH_HIP(d_code.alloc(n + 64)); SH_HIP(d_st.alloc(n + 64));
    hipLaunchKernelGGL(k_fa_code, dim3((uint32_t)((n + 16 * 256 - 1) / (16 * 256))), dim3(256), 0, s, d_raw.as<uint8_t>(), n, d_code.as<uint8_t>());
    // the scan in pieces of 2^30 bytes (32-bit sizes inside the library are then never in question); each piece's first code
    // is composed with the scanned value before it
    const uint64_t PIECE = 1ull << 30;
    size_t tb = 0;
    SH_HIP(rocprim::inclusive_scan(nullptr, tb, d_code.as<uint8_t>(), d_st.as<uint8_t>(), (size_t)std::min(n, PIECE), FaCompose(), s));
    SH_HIP(d_tmp.alloc(tb));
    for (uint64_t p0 = 0; p0 < n; p0 += PIECE) {
        const size_t len = (size_t)std::min(PIECE, n - p0);
        if (p0) hipLaunchKernelGGL(k_fa_carry, dim3(1), dim3(1), 0, s, d_code.as<uint8_t>() + p0, d_st.as<uint8_t>() + p0 - 1);
        size_t tb2 = tb;
        SH_HIP(rocprim::inclusive_scan(d_tmp.p, tb2, d_code.as<uint8_t>() + p0, d_st.as<uint8_t>() + p0, len, FaCompose(), s));
    }
    const uint32_t n_blk = (uint32_t)((n + FA_BLK - 1) / FA_BLK);
    SH_HIP(d_cnt.alloc((size_t)n_blk * 4)); SH_HIP(d_off.alloc((size_t)(n_blk + 1) * 8));
    hipLaunchKernelGGL(k_fa_count, dim3(n_blk), dim3(FA_TPB), 0, s, d_raw.as<uint8_t>(), d_st.as<uint8_t>(), n, d_cnt.as<uint32_t>());
    size_t tb3 = 0;
    DevBuf d_tmp3;
    SH_HIP(rocprim::exclusive_scan(nullptr, tb3, d_cnt.as<uint32_t>(), d_off.as<uint64_t>(), (uint64_t)0, (size_t)n_blk, rocprim::plus<uint64_t>(), s));
    SH_HIP(d_tmp3.alloc(tb3));
    SH_HIP(rocprim::exclusive_scan(d_tmp3.p, tb3, d_cnt.as<uint32_t>(), d_off.as<uint64_t>(), (uint64_t)0, (size_t)n_blk, rocprim::plus<uint64_t>(), s));
    uint64_t last_off = 0; uint32_t last_cnt = 0;
    SH_HIP(hipMemcpy(&last_off, d_off.as<uint64_t>() + (n_blk - 1), 8, hipMemcpyDeviceToHost));
    SH_HIP(hipMemcpy(&last_cnt, d_cnt.as<uint32_t>() + (n_blk - 1), 4, hipMemcpyDeviceToHost));
    const uint64_t n_bases = last_off + last_cnt;
    SH_HIP(d_bases.alloc(n_bases + 64));
    uint32_t hdr_cap = 1u << 16, hdr_n = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {     // second round only for references with more than 65 536 records
        DevBuf hp, hr, hn;
        SH_HIP(hp.alloc((size_t)hdr_cap * 8)); SH_HIP(hr.alloc((size_t)hdr_cap * 8)); SH_HIP(hn.alloc(4));
        SH_HIP(hipMemsetAsync(hn.p, 0, 4, s));
        hipLaunchKernelGGL(k_fa_scatter, dim3(n_blk), dim3(FA_TPB), 0, s, d_raw.as<uint8_t>(), d_st.as<uint8_t>(), n, d_off.as<uint64_t>(), d_bases.as<uint8_t>(),
                           hp.as<unsigned long long>(), hr.as<unsigned long long>(), hdr_cap, hn.as<uint32_t>());
        SH_HIP(hipMemcpy(&hdr_n, hn.p, 4, hipMemcpyDeviceToHost));
        if (hdr_n <= hdr_cap) {
            std::vector<unsigned long long> pos(hdr_n), rank(hdr_n);
            if (hdr_n) { SH_HIP(hipMemcpy(pos.data(), hp.p, (size_t)hdr_n * 8, hipMemcpyDeviceToHost)); SH_HIP(hipMemcpy(rank.data(), hr.p, (size_t)hdr_n * 8, hipMemcpyDeviceToHost)); }
            std::vector<uint32_t> ord(hdr_n);
            for (uint32_t i = 0; i < hdr_n; ++i) ord[i] = i;
            std::sort(ord.begin(), ord.end(), [&](uint32_t x, uint32_t y) { return pos[x] < pos[y]; });
            contig_starts.clear();
            for (uint32_t i = 0; i < hdr_n; ++i) contig_starts.push_back(rank[ord[i]]);
            contig_starts.push_back(n_bases);
            break;
        }
        hdr_cap = hdr_n;
    }
    SH_HIP(hipGetLastError());
    SH_CHECK(hdr_n > 0, SH_ERR_INDEX, "no sequences in %s", path);
    if (dbg) fprintf(stderr, "[scrubby-hip] FASTA on the GPU: %.1f MB read + uploaded in %.0f ms, split into %u records / %llu bases in %.0f ms\n", n / 1e6, msf(t_a, t_b), hdr_n,
                     (unsigned long long)n_bases, msf(t_b, tick()));
    *handled = true;
    return SH_OK;
}

// ---- minimap2 index files (.mmi) ------------------------------------------------------------------------------------------
// `cleaner.rs:475-479` hands minimap2 whatever path the user gave; minimap2 recognises its own index dump by the magic "MMI\2"
// (mm_idx_reader_open / mm_idx_load, index.c of v2.28 - restated from the writer, mm_idx_dump: magic; uint32 w, k, b, n_seq, flag; per
// sequence uint8 name length, name, uint32 length; per bucket (2^b of them) int32 n + n 8-byte positions, uint32 size + size 16-byte
// (key, value) pairs; then, unless MM_I_NO_SEQ, the sequences as 4-bit codes, eight to a uint32, concatenated).  The minimizer tables are
// skipped: the index is rebuilt on the device from the sequences with the file's k and w (0.2 s for a human genome), which prevail over
// the preset's as they do upstream.  Only the first part of a multi-part file is read (as minimap2-rs does).  PARITY UNPINNED: no
// minimap2 is on this box to write such a file; tests/test_index_mmi.py writes one from the statement above.
static sh_status mmi_read(const char *path, std::vector<std::vector<uint8_t>> &seqs, int32_t *k_o, int32_t *w_o)
{
    FILE *f = fopen(path, "rb");
    SH_CHECK(f, SH_ERR_IO, "cannot open %s", path);
    struct Closer { FILE *f; ~Closer() { fclose(f); } } closer{f};
    char mg[4];
    uint32_t x[5];
    SH_CHECK(fread(mg, 1, 4, f) == 4 && memcmp(mg, "MMI\2", 4) == 0, SH_ERR_INDEX, "%s: not a minimap2 index (magic)", path);
    SH_CHECK(fread(x, 4, 5, f) == 5, SH_ERR_IO, "%s: truncated minimap2 index header", path);
    const uint32_t w = x[0], k = x[1], b = x[2], n_seq = x[3], flag = x[4];
    SH_CHECK(k >= 1 && k <= 28 && w >= 1 && w < 256 && b <= 28 && n_seq > 0, SH_ERR_INDEX, "%s: implausible minimap2 index header (k=%u w=%u b=%u n_seq=%u)", path, k, w, b, n_seq);
    SH_CHECK(k & 1u, SH_ERR_INDEX, "%s: minimap2 index built with an even k (%u): this path needs an odd k (every preset's is)", path, k);
    SH_CHECK(!(flag & 1u), SH_ERR_PRESET_UNSUPPORTED, "%s: homopolymer-compressed index (MM_I_HPC) - not implemented on the HIP path", path);
    SH_CHECK(!(flag & 2u), SH_ERR_INDEX, "%s: index written without sequences (MM_I_NO_SEQ): the extension filter needs the reference bases", path);
    std::vector<uint32_t> lens(n_seq);
    uint64_t sum_len = 0;
    for (uint32_t i = 0; i < n_seq; ++i) {
        uint8_t l = 0;
        char name[256];
        SH_CHECK(fread(&l, 1, 1, f) == 1 && (l == 0 || fread(name, 1, l, f) == l) && fread(&lens[i], 4, 1, f) == 1, SH_ERR_IO, "%s: truncated minimap2 index (sequence table)", path);
        sum_len += lens[i];
    }
    for (uint64_t i = 0; i < (1ull << b); ++i) {      // the minimizer tables: skipped by their counts
        int32_t n = 0; uint32_t size = 0;
        SH_CHECK(fread(&n, 4, 1, f) == 1 && n >= 0 && fseeko(f, (off_t)n * 8, SEEK_CUR) == 0 && fread(&size, 4, 1, f) == 1 && fseeko(f, (off_t)size * 16, SEEK_CUR) == 0,
                 SH_ERR_IO, "%s: truncated minimap2 index (bucket %llu)", path, (unsigned long long)i);
    }
    std::vector<uint32_t> S((sum_len + 7) / 8);
    SH_CHECK(fread(S.data(), 4, S.size(), f) == S.size(), SH_ERR_IO, "%s: truncated minimap2 index (sequences)", path);
    seqs.assign(n_seq, {});
    uint64_t o = 0;
    for (uint32_t i = 0; i < n_seq; ++i) {
        seqs[i].resize(lens[i]);
        for (uint32_t j = 0; j < lens[i]; ++j, ++o) { const uint32_t c = S[o >> 3] >> ((o & 7) << 2) & 0xfu; seqs[i][j] = (uint8_t)"ACGTN"[c < 4 ? c : 4]; }
    }
    // what minimap2 says on stderr in the same situations: the file's k / w replace the preset's; only the first part of a multi-part index is used
    if ((int32_t)k != *k_o || (int32_t)w != *w_o) fprintf(stderr, "[scrubby-hip] note: %s was built with k=%u w=%u; they replace the preset's k=%d w=%d\n", path, k, w, *k_o, *w_o);
    { char probe; if (fread(&probe, 1, 1, f) == 1) fprintf(stderr, "[scrubby-hip] note: %s holds more than one index part; only the first is used (as minimap2-rs does)\n", path); }
    *k_o = (int32_t)k; *w_o = (int32_t)w;
    return SH_OK;
}

extern "C" sh_status sh_index_build_fasta(const char *path, const sh_opts *opts, int32_t device, sh_index **out)
{
    SH_CHECK(path && opts && out, SH_ERR_BAD_ARG, "sh_index_build_fasta: null argument");
    {
        FILE *f = fopen(path, "rb");
        SH_CHECK(f, SH_ERR_IO, "cannot open %s", path);
        char mg[4] = {0, 0, 0, 0};
        const size_t got = fread(mg, 1, 4, f);
        fclose(f);
        if (got == 4 && memcmp(mg, "MMI\2", 4) == 0) {
            std::vector<std::vector<uint8_t>> seqs;
            sh_opts o2 = *opts;
            sh_status st = mmi_read(path, seqs, &o2.k, &o2.w);
            if (st != SH_OK) return st;
            std::vector<const uint8_t *> ptr(seqs.size());
            std::vector<uint64_t> lens(seqs.size());
            for (size_t i = 0; i < seqs.size(); ++i) { ptr[i] = seqs[i].data(); lens[i] = seqs[i].size(); }
            return sh_index_build(ptr.data(), lens.data(), (uint32_t)seqs.size(), &o2, device, out);
        }
    }
    if (const char *e = getenv("SCRUBBY_HIP_FASTA_HOST")) if (*e == '1') return shi_index_build_fasta_host(path, opts, device, out);
    DevBuf d_bases;
    std::vector<uint64_t> cs;
    bool handled = false;
    sh_status st = fasta_to_device(path, device, d_bases, cs, &handled);
    if (st != SH_OK) return st;
    if (!handled) return shi_index_build_fasta_host(path, opts, device, out);
    const auto t0 = std::chrono::steady_clock::now();
    st = shi_index_build_device(d_bases.as<uint8_t>(), cs.data(), (uint32_t)(cs.size() - 1), opts, device, nullptr, out);
    if (getenv("SCRUBBY_HIP_DBG_HOST") && *getenv("SCRUBBY_HIP_DBG_HOST") == '1')
        fprintf(stderr, "[scrubby-hip] index build on the device: %.0f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    return st;
}

extern "C" sh_status sh_index_build(const uint8_t *const *seqs, const uint64_t *lens, uint32_t n_seq,
                                    const sh_opts *opts, int32_t device, sh_index **out)
{
    SH_CHECK(seqs && lens && opts && out && n_seq > 0, SH_ERR_BAD_ARG, "sh_index_build: null/empty argument");
    SH_HIP(hipSetDevice(device));
    std::vector<uint64_t> cs(n_seq + 1, 0);
    for (uint32_t i = 0; i < n_seq; ++i) cs[i + 1] = cs[i] + lens[i];
    DevBuf d;
    SH_HIP(d.alloc(cs[n_seq] + 16));
    for (uint32_t i = 0; i < n_seq; ++i)
        if (lens[i]) SH_HIP(hipMemcpy(d.as<uint8_t>() + cs[i], seqs[i], lens[i], hipMemcpyHostToDevice));
    return shi_index_build_device(d.as<uint8_t>(), cs.data(), n_seq, opts, device, nullptr, out);
}

extern "C" sh_status sh_index_info_get(const sh_index *idx, sh_index_info *o)
{
    SH_CHECK(idx && o, SH_ERR_BAD_ARG, "sh_index_info_get: null argument");
    o->k = idx->k; o->w = idx->w; o->mid_occ = idx->mid_occ; o->n_contigs = idx->n_contigs; o->n_bases = idx->n_bases;
    o->n_minimizers = idx->n_minimizers; o->n_keys = idx->n_keys; o->n_slots = idx->n_slots; o->n_positions = idx->n_positions;
    o->hbm_bytes = idx->n_slots * 16 + idx->n_positions * 8; o->build_ms = idx->build_ms;
    return SH_OK;
}

extern "C" sh_status sh_index_export(const sh_index *idx, uint64_t *slots, uint64_t *positions)
{
    SH_CHECK(idx, SH_ERR_BAD_ARG, "sh_index_export: null index");
    SH_HIP(hipSetDevice(idx->device));
    if (slots) SH_HIP(hipMemcpy(slots, idx->d_slots, idx->n_slots * 16, hipMemcpyDeviceToHost));
    if (positions && idx->n_positions) SH_HIP(hipMemcpy(positions, idx->d_positions, idx->n_positions * 8, hipMemcpyDeviceToHost));
    return SH_OK;
}

extern "C" sh_status sh_index_export_ref(const sh_index *idx, uint8_t *packed, uint64_t *contig_start)
{
    SH_CHECK(idx, SH_ERR_BAD_ARG, "sh_index_export_ref: null index");
    SH_CHECK(idx->d_ref && idx->d_cstart, SH_ERR_INDEX, "sh_index_export_ref: this index holds no reference bases (loaded from a cache written without them)");
    SH_HIP(hipSetDevice(idx->device));
    if (packed) SH_HIP(hipMemcpy(packed, idx->d_ref, (idx->n_bases + 1) / 2, hipMemcpyDeviceToHost));
    if (contig_start) SH_HIP(hipMemcpy(contig_start, idx->d_cstart, (idx->n_contigs + 1) * 8, hipMemcpyDeviceToHost));
    return SH_OK;
}

extern "C" sh_status sh_index_free(sh_index *idx)
{
    if (!idx) return SH_OK;
    hipSetDevice(idx->device);
    shi_batch_pool_release(idx);
    if (idx->d_slots) hipFree(idx->d_slots);
    if (idx->d_positions) hipFree(idx->d_positions);
    if (idx->d_ref) hipFree(idx->d_ref);
    if (idx->d_cstart) hipFree(idx->d_cstart);
    delete idx;
    return SH_OK;
}
