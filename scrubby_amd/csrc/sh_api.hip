// sh_api.hip — library-level entry points of the C ABI (include/scrubby_hip.h): errors, presets,
// host-buffer classification, index cache, synthetic workload, gather micro-benchmark.
#include "sh_common.h"
#include "sh_synth_core.h"
#include <cstdarg>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <mutex>
#include <zlib.h>

// ---- errors ----------------------------------------------------------------------------------
static thread_local char g_err[1024] = "";
void sh_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char *sh_last_error(void) { return g_err; }
extern "C" int32_t sh_version(void) { return SH_VERSION; }
extern "C" int32_t sh_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ---- presets: Aligner::builder().sr() / .map_ont() / ...  (/root/reference/src/cleaner.rs:455-470) ------
// values: SURVEY.md App. A.1 (minimap2 mm_set_opt)
static void opts_default(sh_opts *o)
{
    memset(o, 0, sizeof(*o));
    o->k = 15; o->w = 10; o->is_sr = 0; o->mid_occ = 0; o->max_occ = 0;
    o->max_max_occ = 4095; o->occ_dist = 500; o->min_mid_occ = 10; o->max_mid_occ = 1000000;
    o->mid_occ_frac = 2e-4f; o->q_occ_frac = 0.01f;
    o->min_cnt = 3; o->min_chain_score = 40;
    o->max_gap = 5000; o->max_gap_ref = -1; o->max_frag_len = 0; o->bw = 500;
    o->max_chain_skip = 25; o->max_chain_iter = 5000;
    o->chain_gap_scale = 0.8f; o->chain_skip_scale = 0.0f;
    // mm_mapopt_init; `.with_cigar()` (cleaner.rs:473) sets MM_F_CIGAR on whatever preset was chosen
    o->flags = SH_F_CIGAR;
    o->a = 2; o->b = 4; o->q = 4; o->e = 2; o->q2 = 24; o->e2 = 1; o->sc_ambi = 1;
    o->zdrop = 400; o->zdrop_inv = 200; o->end_bonus = -1; o->min_dp_max = o->min_chain_score * o->a;
    o->best_n = 5; o->bw_long = 20000; o->min_ksw_len = 200;
    o->pri_ratio = 0.8f; o->mask_level = 0.5f; o->max_clip_ratio = 1.0f;
    o->rmq_inner_dist = 1000; o->rmq_size_cap = 100000; o->rmq_rescue_size = 1000; o->rmq_rescue_ratio = 0.1f;
}

extern "C" sh_status sh_preset(const char *name, sh_opts *o)
{
    SH_CHECK(name && o, SH_ERR_BAD_ARG, "sh_preset: null argument");
    opts_default(o);
    std::string n(name);
    if (n == "sr") {
        o->k = 21; o->w = 11; o->is_sr = 1; o->max_frag_len = 800; o->max_gap = 100; o->bw = 100;
        o->min_cnt = 2; o->min_chain_score = 25; o->mid_occ = 1000; o->max_occ = 5000;
        o->a = 2; o->b = 8; o->q = 12; o->e = 2; o->q2 = 24; o->e2 = 1;
        o->zdrop = o->zdrop_inv = 100; o->end_bonus = 10; o->bw_long = 100;
        o->pri_ratio = 0.5f; o->min_dp_max = 40; o->best_n = 20;
        return SH_OK;
    }
    if (n == "map-ont") { o->k = 15; o->w = 10; return SH_OK; }
    if (n == "lr:hq" || n == "map-hifi") {      // Preset::LrHq / Preset::MapHifi, cleaner.rs:458,465: one sketch and chaining set-up ...
        o->k = 19; o->w = 19; o->max_gap = 10000; o->min_mid_occ = 50; o->max_mid_occ = 500;
        if (n == "map-hifi") {                  // ... map-hifi adds its own alignment scores and a higher dp_max floor (extension stage)
            o->a = 1; o->b = 4; o->q = 6; o->q2 = 26; o->e = 2; o->e2 = 1; o->min_dp_max = 200;
        }
        return SH_OK;
    }
    if (n == "lr") {   // ScrubbyError::Minimap2PresetNotSupported(Preset::Lr), cleaner.rs:469
        sh_set_error("Minimap2 preset not supported: lr");
        return SH_ERR_PRESET_UNSUPPORTED;
    }
    static const char *later[] = {"asm", "asm5", "asm10", "asm20", "ava-ont", "ava-pb", "map-pb", "splice", "splice:hq"};
    for (const char *l : later)
        if (n == l) {
            sh_set_error("preset %s: %s not implemented on the HIP path yet", name,
                         n == "map-pb" || n == "ava-pb" ? "homopolymer-compressed minimizers (MM_I_HPC: variable k-mer spans in sketch and chain)" :
                         n.rfind("splice", 0) == 0 ? "splice-aware chaining and alignment" :
                         n.rfind("ava", 0) == 0 ? "all-vs-all overlap mode" : "RMQ chaining (MM_F_RMQ)");
            return SH_ERR_PRESET_UNSUPPORTED;
        }
    sh_set_error("unknown preset: %s", name);
    return SH_ERR_PRESET_UNKNOWN;
}

// ---- host-buffer classification ----------------------------------------------------------------
// What a call needs besides the index: a context (minimap2's thread buffer), device copies of the batch, two streams.  Creating
// and freeing ~30 GB of it per call is not free (the driver wipes freed HBM before handing it out again: 0.1 s became 0.6 s),
// so the index keeps what calls leave behind and the next call - from any thread - takes the first set that is large enough.
struct BatchScratch {
    sh_ctx *ctx = nullptr;
    sh_opts opts{};
    std::string env;          // the environment switches a context captures when it is created
    uint64_t ctx_reads = 0, ctx_bases = 0;
    uint32_t ctx_len = 0;
    uint8_t *d_bases = nullptr, *d_flags = nullptr;
    uint64_t *d_off = nullptr;
    sh_trace *d_tr = nullptr;
    size_t cap_bases = 0, cap_reads = 0, cap_tr = 0;
    hipStream_t s = nullptr, us = nullptr;
    void release()
    {
        if (ctx) sh_ctx_destroy(ctx);
        if (d_bases) hipFree(d_bases);
        if (d_flags) hipFree(d_flags);
        if (d_off) hipFree(d_off);
        if (d_tr) hipFree(d_tr);
        if (s) hipStreamDestroy(s);
        if (us) hipStreamDestroy(us);
    }
};

void shi_batch_pool_release(sh_index *idx)
{
    std::lock_guard<std::mutex> lk(idx->pool_mu);
    for (void *p : idx->pool) { ((BatchScratch *)p)->release(); delete (BatchScratch *)p; }
    idx->pool.clear();
}

extern "C" sh_status sh_classify_batch(const sh_index *idx, const sh_opts *opts, const uint8_t *bases, const uint64_t *offsets,
                                       uint64_t n_reads, uint8_t *out_flags, sh_trace *out_trace, sh_stats *stats)
{
    SH_CHECK(idx && opts && offsets && out_flags, SH_ERR_BAD_ARG, "sh_classify_batch: null argument");
    if (stats) memset(stats, 0, sizeof(*stats));
    if (n_reads == 0) return SH_OK;
    SH_HIP(hipSetDevice(idx->device));
    const uint64_t CH = 1ull << 22;       // reads per piece
    const uint64_t n_bases = offsets[n_reads] - offsets[0];
    uint32_t max_len = 0;
    uint64_t piece_bases = 0;
    for (uint64_t r = 0; r < n_reads; ++r) max_len = std::max<uint32_t>(max_len, (uint32_t)std::min<uint64_t>(offsets[r + 1] - offsets[r], UINT32_MAX));
    for (uint64_t r0 = 0; r0 < n_reads; r0 += CH) piece_bases = std::max(piece_bases, offsets[std::min(n_reads, r0 + CH)] - offsets[r0]);
    const uint64_t need_reads = std::min(CH, n_reads);
    std::string env_sig;
    for (const char *v : {"SCRUBBY_HIP_ARENA_MB", "SCRUBBY_HIP_NO_FLAG_STOP", "SCRUBBY_HIP_NO_PAIR", "SCRUBBY_HIP_PAIR_MIN", "SCRUBBY_HIP_NO_S1", "SCRUBBY_HIP_EXT_MB", "SCRUBBY_HIP_EXT_REGCAP", "SCRUBBY_HIP_NO_LEMMA",
                          "SCRUBBY_HIP_LEXT_A", "SCRUBBY_HIP_LEXT_BIG_A", "SCRUBBY_HIP_LEXT_P_KB", "SCRUBBY_HIP_RMQ_EXACT_MAX", "SCRUBBY_HIP_RMQ_ONE_LANE", "SCRUBBY_HIP_COOP_MIN", "SCRUBBY_HIP_COOP_RUN", "SCRUBBY_HIP_E2_JOIN_MIN", "SCRUBBY_HIP_LEXT_BIG_P_KB", "SCRUBBY_HIP_STAGE_MB", "SCRUBBY_HIP_STREAMS"}) { const char *e = getenv(v); env_sig += e ? e : "-"; env_sig += '|'; }
    const bool no_pool = false;

    BatchScratch *B = nullptr;
    if (!no_pool) {
        std::lock_guard<std::mutex> lk(idx->pool_mu);
        for (size_t i = 0; i < idx->pool.size(); ++i) {
            BatchScratch *c = (BatchScratch *)idx->pool[i];
            if (!memcmp(&c->opts, opts, sizeof(sh_opts)) && c->env == env_sig) { B = c; idx->pool.erase(idx->pool.begin() + (long)i); break; }
        }
    }
    if (!B) { B = new BatchScratch; B->opts = *opts; B->env = env_sig; }
    auto give_back = [&](bool keep) {
        if (keep && !no_pool) { std::lock_guard<std::mutex> lk(idx->pool_mu); idx->pool.push_back(B); }
        else { B->release(); delete B; }
    };
#define CB_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { sh_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); give_back(false); return e_ == hipErrorOutOfMemory ? SH_ERR_OOM : SH_ERR_HIP; } } while (0)
    sh_status st = SH_OK;
    if (!B->ctx || need_reads > B->ctx_reads || piece_bases > B->ctx_bases || max_len > B->ctx_len) {
        if (B->ctx) { sh_ctx_destroy(B->ctx); B->ctx = nullptr; }
        B->ctx_reads = std::max(B->ctx_reads, need_reads);
        B->ctx_bases = std::max(B->ctx_bases, piece_bases);
        B->ctx_len = std::max(B->ctx_len, max_len);
        st = sh_ctx_create(idx, opts, B->ctx_reads, B->ctx_bases, B->ctx_len, &B->ctx);
        if (st != SH_OK) { give_back(false); return st; }
    }
    if (!B->s) CB_HIP(hipStreamCreate(&B->s));
    if (!B->us) CB_HIP(hipStreamCreateWithFlags(&B->us, hipStreamNonBlocking));
    if (n_bases + 32 > B->cap_bases) { if (B->d_bases) hipFree(B->d_bases); B->d_bases = nullptr; B->cap_bases = n_bases + 32; CB_HIP(hipMalloc(&B->d_bases, B->cap_bases)); }
    if (n_reads + 1 > B->cap_reads) {
        if (B->d_off) hipFree(B->d_off);
        if (B->d_flags) hipFree(B->d_flags);
        B->d_off = nullptr; B->d_flags = nullptr; B->cap_reads = n_reads + 1;
        CB_HIP(hipMalloc(&B->d_off, B->cap_reads * 8));
        CB_HIP(hipMalloc(&B->d_flags, B->cap_reads));
    }
    if (out_trace && n_reads > B->cap_tr) { if (B->d_tr) hipFree(B->d_tr); B->d_tr = nullptr; B->cap_tr = n_reads; CB_HIP(hipMalloc(&B->d_tr, B->cap_tr * sizeof(sh_trace))); }
    sh_ctx *ctx = B->ctx;
    uint8_t *d_bases = B->d_bases, *d_flags = B->d_flags; uint64_t *d_off = B->d_off; sh_trace *d_tr = out_trace ? B->d_tr : nullptr;
    hipStream_t s = B->s, us = B->us;
    // offsets are rebased so that d_bases[0] is the first base of the batch
    std::vector<uint64_t> off0;
    const uint64_t *offp = offsets;
    if (offsets[0] != 0) {
        off0.resize(n_reads + 1);
        for (uint64_t r = 0; r <= n_reads; ++r) off0[r] = offsets[r] - offsets[0];
        offp = off0.data();
    }
    CB_HIP(hipMemcpyAsync(d_off, offp, (n_reads + 1) * 8, hipMemcpyHostToDevice, s));
    CB_HIP(hipStreamSynchronize(s));
    // bases go up piece by piece (CH reads each) from the calling thread on a stream of their own, while a worker thread classifies
    // the pieces that have arrived: piece i + 1 crosses PCIe while piece i is in the kernels
    const uint64_t n_pieces = (n_reads + CH - 1) / CH;
    std::vector<std::atomic<int>> ready(n_pieces);
    for (auto &r : ready) r.store(0);
    std::atomic<int> up_err{0};
    std::string werr;
    std::thread worker([&]() {
        for (uint64_t p = 0; p < n_pieces && st == SH_OK; ++p) {
            while (!ready[p].load(std::memory_order_acquire)) std::this_thread::yield();
            if (up_err.load()) break;
            const uint64_t r0 = p * CH, r1 = std::min(n_reads, r0 + CH);
            sh_stats ps;
            st = sh_classify_device(ctx, d_bases, d_off + r0, r1 - r0, n_bases, d_flags + r0, d_tr ? d_tr + r0 : nullptr, s, stats ? &ps : nullptr);
            if (st != SH_OK) werr = sh_last_error();
            if (stats && st == SH_OK) {
                stats->n_reads += ps.n_reads; stats->n_host += ps.n_host; stats->n_no_seed += ps.n_no_seed; stats->n_chain_small += ps.n_chain_small;
                stats->n_chain_large += ps.n_chain_large; stats->n_minimizers += ps.n_minimizers; stats->n_bases += ps.n_bases;
                stats->ms_sketch_probe += ps.ms_sketch_probe; stats->ms_chain_small += ps.ms_chain_small; stats->ms_chain_large += ps.ms_chain_large; stats->ms_total += ps.ms_total;
                stats->n_anchors += ps.n_anchors; stats->n_clusters += ps.n_clusters; stats->n_resketch += ps.n_resketch; stats->n_pair_decided += ps.n_pair_decided;
                stats->n_ext_reads += ps.n_ext_reads; stats->n_ext_regions += ps.n_ext_regions; stats->n_ext_dropped += ps.n_ext_dropped; stats->ms_ext += ps.ms_ext; stats->n_ext_shortcut += ps.n_ext_shortcut;
                stats->n_ext_fallback += ps.n_ext_fallback; stats->ms_ext_fallback += ps.ms_ext_fallback; stats->n_ext_unresolved += ps.n_ext_unresolved; stats->n_rmq_rechained += ps.n_rmq_rechained; stats->n_rmq_tied += ps.n_rmq_tied;
                stats->n_dp_parallel += ps.n_dp_parallel; stats->n_dp_dirty += ps.n_dp_dirty; stats->n_top_settled += ps.n_top_settled;
                stats->n_locus_reads += ps.n_locus_reads; stats->n_locus_redone += ps.n_locus_redone; stats->n_rmq_exact += ps.n_rmq_exact; stats->n_ext_ondemand += ps.n_ext_ondemand; stats->n_rmq_open += ps.n_rmq_open;
            }
        }
    });
    for (uint64_t p = 0; p < n_pieces; ++p) {
        const uint64_t r0 = p * CH, r1 = std::min(n_reads, r0 + CH);
        const uint64_t b0 = offp[r0], b1 = offp[r1];
        if (!up_err.load() && b1 > b0 &&
            (hipMemcpyAsync(d_bases + b0, bases + offsets[0] + b0, b1 - b0, hipMemcpyHostToDevice, us) != hipSuccess || hipStreamSynchronize(us) != hipSuccess)) up_err = 1;
        ready[p].store(1, std::memory_order_release);
    }
    worker.join();
    if (up_err.load() && st == SH_OK) { sh_set_error("sh_classify_batch: host-to-device copy failed"); st = SH_ERR_HIP; }
    else if (st != SH_OK) sh_set_error("%s", werr.c_str());
    if (st != SH_OK) { give_back(false); return st; }
    CB_HIP(hipMemcpyAsync(out_flags, d_flags, n_reads, hipMemcpyDeviceToHost, s));
    if (out_trace) CB_HIP(hipMemcpyAsync(out_trace, d_tr, n_reads * sizeof(sh_trace), hipMemcpyDeviceToHost, s));
    CB_HIP(hipStreamSynchronize(s));
    give_back(true);
#undef CB_HIP
    for (uint64_t r = 0; r < n_reads; ++r)
        if (out_flags[r] == 2) {   // minimap2-rs: Err("Sequence is empty") aborts the run (cleaner.rs:552,566)
            sh_set_error("Sequence is empty (read %llu)", (unsigned long long)r);
            return SH_ERR_EMPTY_READ;
        }
    return SH_OK;
}

// ---- in-process multi-GPU: replicas of the index and one call that fans a batch out over them ----------------------------
// The reference shares ONE &Aligner among all rayon workers (cleaner.rs:546-559); a Rust caller of this library cannot use
// torch.distributed, so the fan-out SURVEY.md 8(b) "Threading" / 8(e) asks for lives behind the C ABI: the index is replicated per device
// (<= 20 GB of 288), the batch is cut into contiguous shards - at even record ordinals, so that mates stay together, and balanced by BASES,
// which for fixed-length records is by count - each shard goes through sh_classify_batch on its own host thread (its own context, streams and
// device buffers, kept with that replica), and the union of cleaner.rs:564-570 is the concatenation of the shards' flags in the caller's
// array: one process, no collective.
static uint64_t ref_words(uint64_t n_bases) { return ((n_bases + 31) / 32) * 2; }      // the packed reference is written in 16-B units

struct sh_index_set {
    std::vector<const sh_index *> rep;      // one per shard (a device may appear more than once: logical shards on one device)
    std::vector<sh_index *> owned;          // the copies this set made (freed with it); the source index is borrowed
};

static sh_status index_copy_to(const sh_index *src, int device, sh_index **out)
{
    sh_index *r = new sh_index();
    r->device = device; r->k = src->k; r->w = src->w; r->mid_occ = src->mid_occ; r->n_contigs = src->n_contigs; r->n_bases = src->n_bases;
    r->n_minimizers = src->n_minimizers; r->n_keys = src->n_keys; r->n_slots = src->n_slots; r->n_positions = src->n_positions; r->lg_slots = src->lg_slots;
    r->contig_len = src->contig_len; r->o_mid_occ = src->o_mid_occ; r->o_min_mid_occ = src->o_min_mid_occ; r->o_max_mid_occ = src->o_max_mid_occ;
    r->o_mid_occ_frac = src->o_mid_occ_frac; r->ref_checksum = src->ref_checksum; r->build_ms = src->build_ms;
    const size_t b_slots = (size_t)src->n_slots * 16, b_pos = ((size_t)src->n_positions + 2) * 8, b_ref = src->d_ref ? (size_t)ref_words(src->n_bases) * 8 + 16 : 0,
                 b_cs = src->d_cstart ? ((size_t)src->n_contigs + 1) * 8 : 0;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipMalloc(&r->d_slots, b_slots);
    if (e == hipSuccess) e = hipMalloc(&r->d_positions, b_pos);
    if (e == hipSuccess && b_ref) e = hipMalloc(&r->d_ref, b_ref);
    if (e == hipSuccess && b_cs) e = hipMalloc(&r->d_cstart, b_cs);
    // device to device over xGMI where the two devices are peers; the runtime stages through the host otherwise
    if (e == hipSuccess) e = hipMemcpyPeer(r->d_slots, device, src->d_slots, src->device, b_slots);
    if (e == hipSuccess && src->n_positions) e = hipMemcpyPeer(r->d_positions, device, src->d_positions, src->device, (size_t)src->n_positions * 8);
    if (e == hipSuccess && b_ref) e = hipMemcpyPeer(r->d_ref, device, src->d_ref, src->device, b_ref);
    if (e == hipSuccess && b_cs) e = hipMemcpyPeer(r->d_cstart, device, src->d_cstart, src->device, b_cs);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        sh_set_error("sh_index_replicate: device %d: %s", device, hipGetErrorString(e));
        sh_index_free(r);
        return e == hipErrorOutOfMemory ? SH_ERR_OOM : SH_ERR_HIP;
    }
    *out = r;
    return SH_OK;
}

extern "C" sh_status sh_index_set_free(sh_index_set *set)
{
    if (!set) return SH_OK;
    for (sh_index *r : set->owned) sh_index_free(r);
    delete set;
    return SH_OK;
}

extern "C" sh_status sh_index_replicate(const sh_index *idx, const int32_t *devices, uint32_t n_devices, sh_index_set **out)
{
    SH_CHECK(idx && out, SH_ERR_BAD_ARG, "sh_index_replicate: null argument");
    const int n_vis = sh_device_count();
    SH_CHECK(n_vis > 0, SH_ERR_NO_DEVICE, "sh_index_replicate: no HIP device");
    std::vector<int> devs;
    if (!devices || n_devices == 0) for (int d = 0; d < n_vis; ++d) devs.push_back(d);
    else devs.assign(devices, devices + n_devices);
    for (int d : devs) SH_CHECK(d >= 0 && d < n_vis, SH_ERR_BAD_ARG, "sh_index_replicate: device %d of %d", d, n_vis);
    sh_index_set *S = new sh_index_set;
    for (int d : devs) {
        const sh_index *have = d == idx->device ? idx : nullptr;
        for (sh_index *o : S->owned) if (o->device == d) have = o;
        if (!have) {
            sh_index *r = nullptr;
            const sh_status st = index_copy_to(idx, d, &r);
            if (st != SH_OK) { sh_index_set_free(S); return st; }
            S->owned.push_back(r);
            have = r;
        }
        S->rep.push_back(have);
    }
    hipSetDevice(idx->device);
    *out = S;
    return SH_OK;
}

extern "C" uint32_t sh_index_set_size(const sh_index_set *set) { return set ? (uint32_t)set->rep.size() : 0u; }

// first record of each shard: shard_first[n_shards + 1], even ordinals (mates of a pair are records 2i, 2i + 1), equal shares of the bases
static void shard_cuts(const uint64_t *offsets, uint64_t n_reads, uint32_t n_shards, std::vector<uint64_t> &cut)
{
    cut.assign(n_shards + 1, 0);
    cut[n_shards] = n_reads;
    const uint64_t b0 = offsets[0], tot = offsets[n_reads] - b0;
    for (uint32_t i = 1; i < n_shards; ++i) {
        const uint64_t want = b0 + (uint64_t)((unsigned __int128)tot * i / n_shards);
        uint64_t r = (uint64_t)(std::lower_bound(offsets, offsets + n_reads + 1, want) - offsets);      // first record starting at or behind the share's end
        r &= ~1ull;
        cut[i] = std::min<uint64_t>(std::max<uint64_t>(r, cut[i - 1]), n_reads & ~1ull);
    }
}

extern "C" sh_status sh_classify_sharded(const sh_index_set *set, const sh_opts *opts, const uint8_t *bases, const uint64_t *offsets,
                                         uint64_t n_reads, uint8_t *out_flags, sh_trace *out_trace, sh_stats *stats, uint64_t *shard_first)
{
    SH_CHECK(set && !set->rep.empty() && opts && offsets && out_flags, SH_ERR_BAD_ARG, "sh_classify_sharded: null argument");
    const uint32_t n_sh = (uint32_t)set->rep.size();
    if (stats) memset(stats, 0, sizeof(*stats));
    std::vector<uint64_t> cut;
    shard_cuts(offsets, n_reads, n_sh, cut);
    if (shard_first) for (uint32_t i = 0; i <= n_sh; ++i) shard_first[i] = cut[i];
    if (n_reads == 0) return SH_OK;
    std::vector<sh_status> rc(n_sh, SH_OK);
    std::vector<std::string> msg(n_sh);
    std::vector<sh_stats> ss(n_sh);
    std::vector<std::thread> th;
    for (uint32_t i = 0; i < n_sh; ++i)
        th.emplace_back([&, i]() {
            const uint64_t r0 = cut[i], r1 = cut[i + 1];
            memset(&ss[i], 0, sizeof(sh_stats));
            if (r1 <= r0) return;
            // `bases` stays the caller's pointer: sh_classify_batch reads bases + offsets[r], whatever offsets[r0] is
            rc[i] = sh_classify_batch(set->rep[i], opts, bases, offsets + r0, r1 - r0, out_flags + r0, out_trace ? out_trace + r0 : nullptr, stats ? &ss[i] : nullptr);
            if (rc[i] != SH_OK) msg[i] = sh_last_error();      // (thread-local: carried over to the caller's thread below)
        });
    for (auto &t : th) t.join();
    sh_status st = SH_OK;
    for (uint32_t i = 0; i < n_sh; ++i)
        if (rc[i] != SH_OK && (st == SH_OK || st == SH_ERR_EMPTY_READ)) {      // an empty read is reported only when nothing worse happened
            if (rc[i] == SH_ERR_EMPTY_READ && st == SH_ERR_EMPTY_READ) continue;
            st = rc[i];
            sh_set_error("shard %u (records %llu..%llu, device %d): %s", i, (unsigned long long)cut[i], (unsigned long long)cut[i + 1], set->rep[i]->device, msg[i].c_str());
        }
    if (stats)
        for (uint32_t i = 0; i < n_sh; ++i) {
            // counters add up; the stage times are per-device clocks of launches that ran side by side: the largest is the job's
            const uint64_t *a = (const uint64_t *)&ss[i]; uint64_t *d = (uint64_t *)stats;
            static_assert(sizeof(sh_stats) % 8 == 0, "sh_stats is made of 8-byte fields");
            const size_t dbl[] = {offsetof(sh_stats, ms_sketch_probe) / 8, offsetof(sh_stats, ms_chain_small) / 8, offsetof(sh_stats, ms_chain_large) / 8, offsetof(sh_stats, ms_total) / 8,
                                  offsetof(sh_stats, ms_ext) / 8, offsetof(sh_stats, ms_ext_fallback) / 8};
            for (size_t w = 0; w < sizeof(sh_stats) / 8; ++w) {
                bool is_d = false;
                for (size_t q : dbl) is_d |= q == w;
                if (is_d) { double &x = ((double *)stats)[w]; x = std::max(x, ((const double *)&ss[i])[w]); }
                else d[w] += a[w];
            }
        }
    return st;
}

// ---- index cache -----------------------------------------------------------------
// The reference rebuilds the index on every run (`.with_index(path, None)`, cleaner.rs:475-479: no output file).  The cache is
// one file: header, contig lengths, the 16-B slots, the position array, the 4-bit reference (what the extension stage aligns
// against).  The header carries the sketch parameters, the occurrence parameters mid_occ was derived with, and a checksum per
// section (order-sensitive 64-bit sums computed on the device), the reference's among them: an index is keyed by
// (reference checksum, k, w).  Sections stream between the file and HBM in 64 MiB pieces through one pinned buffer.
struct CacheHeader {
    char magic[8];                 // "SHIDX002"
    int32_t k, w, mid_occ; uint32_t n_contigs, lg_slots, has_ref;
    uint64_t n_bases, n_minimizers, n_keys, n_slots, n_positions;
    int32_t o_mid_occ, o_min_mid_occ, o_max_mid_occ; float o_mid_occ_frac;
    uint64_t cs_slots, cs_positions, cs_ref, cs_contigs;
};

__global__ void k_checksum(const uint64_t *w, uint64_t n, unsigned long long *out)
{
    unsigned long long acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        acc += (w[i] ^ 0x9E3779B97F4A7C15ULL) * (2 * i + 1);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += (unsigned long long)__shfl_xor((long long)acc, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}

static sh_status device_checksum(const void *d, uint64_t n_words, uint64_t *out)
{
    unsigned long long *d_acc = nullptr;
    *out = 0;
    if (n_words == 0) return SH_OK;
    SH_HIP(hipMalloc(&d_acc, 8));
    hipError_t e = hipMemset(d_acc, 0, 8);
    if (e == hipSuccess) { hipLaunchKernelGGL(k_checksum, dim3(1024), dim3(256), 0, 0, (const uint64_t *)d, n_words, d_acc); e = hipMemcpy(out, d_acc, 8, hipMemcpyDeviceToHost); }
    hipFree(d_acc);
    SH_HIP(e);
    return SH_OK;
}


extern "C" sh_status sh_index_save(const sh_index *idx, const char *path)
{
    SH_CHECK(idx && path, SH_ERR_BAD_ARG, "sh_index_save: null argument");
    SH_HIP(hipSetDevice(idx->device));
    CacheHeader h{};
    memcpy(h.magic, "SHIDX002", 8);
    h.k = idx->k; h.w = idx->w; h.mid_occ = idx->mid_occ; h.n_contigs = idx->n_contigs; h.lg_slots = idx->lg_slots; h.has_ref = idx->d_ref ? 1 : 0;
    h.n_bases = idx->n_bases; h.n_minimizers = idx->n_minimizers; h.n_keys = idx->n_keys; h.n_slots = idx->n_slots; h.n_positions = idx->n_positions;
    h.o_mid_occ = idx->o_mid_occ; h.o_min_mid_occ = idx->o_min_mid_occ; h.o_max_mid_occ = idx->o_max_mid_occ; h.o_mid_occ_frac = idx->o_mid_occ_frac;
    sh_status st;
    if ((st = device_checksum(idx->d_slots, idx->n_slots * 2, &h.cs_slots)) != SH_OK) return st;
    if ((st = device_checksum(idx->d_positions, idx->n_positions, &h.cs_positions)) != SH_OK) return st;
    if (idx->d_ref && (st = device_checksum(idx->d_ref, ref_words(idx->n_bases), &h.cs_ref)) != SH_OK) return st;
    for (uint32_t i = 0; i < idx->n_contigs; ++i) h.cs_contigs += (idx->contig_len[i] ^ 0x9E3779B97F4A7C15ULL) * (2 * (uint64_t)i + 1);
    FILE *f = fopen(path, "wb");
    SH_CHECK(f, SH_ERR_IO, "cannot open %s for writing", path);
    bool ok = fwrite(&h, sizeof(h), 1, f) == 1 && fwrite(idx->contig_len.data(), 8, idx->n_contigs, f) == idx->n_contigs;
    const size_t CH = 64u << 20;
    void *pin = nullptr;
    if (hipHostMalloc(&pin, CH) != hipSuccess) { fclose(f); sh_set_error("sh_index_save: no pinned staging buffer"); return SH_ERR_OOM; }
    auto section = [&](const void *d, uint64_t bytes) {
        for (uint64_t o = 0; ok && o < bytes; o += CH) {
            const size_t n = (size_t)std::min<uint64_t>(CH, bytes - o);
            ok = hipMemcpy(pin, (const uint8_t *)d + o, n, hipMemcpyDeviceToHost) == hipSuccess && fwrite(pin, 1, n, f) == n;
        }
    };
    section(idx->d_slots, idx->n_slots * 16);
    section(idx->d_positions, idx->n_positions * 8);
    if (idx->d_ref) { section(idx->d_ref, ref_words(idx->n_bases) * 8); }
    hipHostFree(pin);
    ok = fclose(f) == 0 && ok;
    SH_CHECK(ok, SH_ERR_IO, "short write to %s", path);
    return SH_OK;
}

extern "C" sh_status sh_index_load(const char *path, int32_t device, sh_index **out)
{
    SH_CHECK(path && out, SH_ERR_BAD_ARG, "sh_index_load: null argument");
    FILE *f = fopen(path, "rb");
    SH_CHECK(f, SH_ERR_IO, "cannot open %s", path);
    CacheHeader h{};
    if (fread(&h, sizeof(h), 1, f) != 1 || memcmp(h.magic, "SHIDX002", 8) != 0) {
        const bool old = memcmp(h.magic, "SHIDX001", 8) == 0;
        fclose(f);
        sh_set_error(old ? "%s was written by an older scrubby-hip (no reference bases, no checksums): rebuild it" : "%s is not a scrubby-hip index", path);
        return SH_ERR_IO;
    }
    // nothing of the header is trusted before it has been checked against itself and against the file
    bool sane = h.k > 0 && h.k <= 28 && (h.k & 1) && h.w > 0 && h.w < 256 && h.lg_slots >= 4 && h.lg_slots < 40 && h.n_slots == (1ULL << h.lg_slots) &&
                h.n_keys <= h.n_slots && h.n_positions <= h.n_minimizers && h.n_minimizers <= h.n_bases && h.n_contigs > 0 && h.n_contigs <= h.n_bases + 1 && h.mid_occ > 0 && h.has_ref <= 1;
    long long fsize = -1;
    if (sane && fseeko(f, 0, SEEK_END) == 0) fsize = (long long)ftello(f);
    const unsigned long long want = sizeof(h) + 8ull * h.n_contigs + 16ull * h.n_slots + 8ull * h.n_positions + (h.has_ref ? 8ull * ref_words(h.n_bases) : 0ull);
    sane = sane && fsize >= 0 && (unsigned long long)fsize == want && fseeko(f, (off_t)sizeof(h), SEEK_SET) == 0;
    if (!sane) { fclose(f); sh_set_error("%s: corrupt or truncated index (header fields and file size disagree)", path); return SH_ERR_IO; }
    sh_index *idx = new sh_index();
    idx->device = device; idx->k = h.k; idx->w = h.w; idx->mid_occ = h.mid_occ; idx->n_contigs = h.n_contigs; idx->lg_slots = h.lg_slots;
    idx->n_bases = h.n_bases; idx->n_minimizers = h.n_minimizers; idx->n_keys = h.n_keys; idx->n_slots = h.n_slots; idx->n_positions = h.n_positions;
    idx->o_mid_occ = h.o_mid_occ; idx->o_min_mid_occ = h.o_min_mid_occ; idx->o_max_mid_occ = h.o_max_mid_occ; idx->o_mid_occ_frac = h.o_mid_occ_frac;
    idx->contig_len.resize(h.n_contigs);
    auto fail = [&](sh_status code) { fclose(f); sh_index_free(idx); return code; };
    if (fread(idx->contig_len.data(), 8, h.n_contigs, f) != h.n_contigs) { sh_set_error("short read from %s", path); return fail(SH_ERR_IO); }
    uint64_t tot = 0, cs = 0;
    std::vector<uint64_t> starts(h.n_contigs + 1, 0);
    for (uint32_t i = 0; i < h.n_contigs; ++i) { cs += (idx->contig_len[i] ^ 0x9E3779B97F4A7C15ULL) * (2 * (uint64_t)i + 1); tot += idx->contig_len[i]; starts[i + 1] = tot; if (idx->contig_len[i] >= (1ULL << 31) - 1) tot = ~0ull >> 1; }
    if (cs != h.cs_contigs || tot != h.n_bases) { sh_set_error("%s: contig table does not match its checksum", path); return fail(SH_ERR_IO); }
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipMalloc(&idx->d_slots, h.n_slots * 16);
    if (e == hipSuccess) e = hipMalloc(&idx->d_positions, (h.n_positions + 2) * 8);
    if (e == hipSuccess && h.has_ref) e = hipMalloc(&idx->d_ref, ref_words(h.n_bases) * 8 + 16);
    if (e == hipSuccess && h.has_ref) e = hipMalloc(&idx->d_cstart, (h.n_contigs + 1) * 8);
    if (e == hipSuccess && h.has_ref) e = hipMemcpy(idx->d_cstart, starts.data(), (h.n_contigs + 1) * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess && h.has_ref) e = hipMemset(idx->d_ref + ref_words(h.n_bases) * 8, 0, 16);
    const size_t CH = 64u << 20;
    void *pin = nullptr;
    if (e == hipSuccess) e = hipHostMalloc(&pin, CH);
    if (e != hipSuccess) { sh_set_error("sh_index_load: %s", hipGetErrorString(e)); return fail(e == hipErrorOutOfMemory ? SH_ERR_OOM : SH_ERR_HIP); }
    bool ok = true;
    auto section = [&](void *d, uint64_t bytes) {
        for (uint64_t o = 0; ok && o < bytes; o += CH) {
            const size_t n = (size_t)std::min<uint64_t>(CH, bytes - o);
            ok = fread(pin, 1, n, f) == n && hipMemcpy((uint8_t *)d + o, pin, n, hipMemcpyHostToDevice) == hipSuccess;
        }
    };
    section(idx->d_slots, h.n_slots * 16);
    section(idx->d_positions, h.n_positions * 8);
    if (h.has_ref) section(idx->d_ref, ref_words(h.n_bases) * 8);
    hipHostFree(pin);
    if (!ok) { sh_set_error("short read from %s", path); return fail(SH_ERR_IO); }
    uint64_t c1 = 0, c2 = 0, c3 = 0;
    sh_status st = device_checksum(idx->d_slots, h.n_slots * 2, &c1);
    if (st == SH_OK) st = device_checksum(idx->d_positions, h.n_positions, &c2);
    if (st == SH_OK && h.has_ref) st = device_checksum(idx->d_ref, ref_words(h.n_bases), &c3);
    if (st != SH_OK) return fail(st);
    if (c1 != h.cs_slots || c2 != h.cs_positions || (h.has_ref && c3 != h.cs_ref)) { sh_set_error("%s: payload does not match its checksums (corrupt index)", path); return fail(SH_ERR_IO); }
    idx->ref_checksum = h.cs_ref;
    fclose(f);
    *out = idx;
    return SH_OK;
}

// ---- FASTA -> index -----------------------------------------------------------------------------------
// line-by-line host reader: FASTQ references, files that do not start with a FASTA header, and the A/B baseline of the
// GPU reader in sh_index.hip (SCRUBBY_HIP_FASTA_HOST=1).  Plain or gzip, sniffed by zlib the way needletail sniffs the magic
// bytes; a truncated or corrupt gzip stream is an error, not a short reference.
sh_status shi_index_build_fasta_host(const char *path, const sh_opts *opts, int32_t device, sh_index **out)
{
    SH_CHECK(path && opts && out, SH_ERR_BAD_ARG, "sh_index_build_fasta: null argument");
    gzFile f = gzopen(path, "rb");
    SH_CHECK(f, SH_ERR_IO, "cannot open %s", path);
    gzbuffer(f, 1 << 20);
    std::vector<std::vector<uint8_t>> seqs;
    std::vector<char> buf(1 << 16);
    std::string line;
    bool fastq = false; int fq_state = 0;
    for (;;) {
        // one LINE, however long: gzgets returns pieces of at most the buffer's size
        line.clear();
        bool got = false;
        while (gzgets(f, buf.data(), (int)buf.size())) {
            got = true;
            const size_t n = strlen(buf.data());
            line.append(buf.data(), n);
            if (n && buf[n - 1] == '\n') break;
        }
        if (!got) break;
        size_t n = line.size();
        while (n && (line[n - 1] == '\n' || line[n - 1] == '\r')) --n;
        if (seqs.empty() && n && line[0] == '@') fastq = true;
        if (!fastq) {
            if (n && line[0] == '>') { seqs.emplace_back(); continue; }
            if (seqs.empty()) continue;
            seqs.back().insert(seqs.back().end(), line.data(), line.data() + n);
        } else {
            if (fq_state == 0) { seqs.emplace_back(); fq_state = 1; }
            else if (fq_state == 1) { seqs.back().insert(seqs.back().end(), line.data(), line.data() + n); fq_state = 2; }
            else if (fq_state == 2) fq_state = 3;
            else fq_state = 0;
        }
    }
    int zerr = Z_OK;
    const char *zmsg = gzerror(f, &zerr);
    const bool bad = zerr != Z_OK && zerr != Z_STREAM_END;
    std::string msg = bad && zmsg ? zmsg : "";
    const int crc = gzclose(f);
    SH_CHECK(!bad && crc == Z_OK, SH_ERR_IO, "%s: gzip stream is truncated or corrupt (%s)", path, msg.empty() ? "unexpected end of file" : msg.c_str());
    SH_CHECK(!seqs.empty(), SH_ERR_INDEX, "no sequences in %s", path);
    std::vector<const uint8_t *> ptr; std::vector<uint64_t> len;
    for (auto &sq : seqs) { ptr.push_back(sq.data()); len.push_back(sq.size()); }
    return sh_index_build(ptr.data(), len.data(), (uint32_t)seqs.size(), opts, device, out);
}

// ---- synthetic workload --------------------------------------------------------------------------------
__global__ void k_synth_ref(syn_ref_params P, uint64_t g0, uint64_t n, uint8_t *out)
{
    uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    uint32_t w = 0;
    for (int b = 0; b < 4; ++b) {
        uint8_t ch = i + b < n ? (uint8_t)"ACGT"[syn_ref_base(&P, g0 + i + b)] : 0;
        w |= (uint32_t)ch << (8 * b);
    }
    if (i + 4 <= n && ((uintptr_t)(out + i) & 3) == 0) *(uint32_t *)(out + i) = w;
    else for (int b = 0; b < 4 && i + b < n; ++b) out[i + b] = (uint8_t)(w >> (8 * b));
}

__global__ void k_synth_reads(syn_ref_params P, syn_read_params R, uint64_t r0, uint64_t n_rec, uint8_t *out, uint64_t *offsets)
{
    // one thread per 4 bases of a record
    const uint32_t q4 = (R.read_len + 3) / 4;
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t rec = t / q4;
    uint32_t j = (uint32_t)(t % q4) * 4;
    if (rec >= n_rec) return;
    uint64_t r = r0 + rec;
    syn_pair pl = syn_place_pair(&P, &R, r >> 1);
    for (uint32_t b = 0; b < 4 && j + b < R.read_len; ++b)
        out[rec * R.read_len + j + b] = syn_read_base(&P, &R, &pl, r >> 1, (uint32_t)(r & 1), j + b);
    if (offsets && j == 0) {
        offsets[rec] = rec * R.read_len;
        if (rec + 1 == n_rec) offsets[n_rec] = n_rec * R.read_len;
    }
}

extern "C" sh_status sh_synth_ref_device(const void *ref_params, uint64_t g0, uint64_t n, uint8_t *d_out, void *stream)
{
    SH_CHECK(ref_params && d_out, SH_ERR_BAD_ARG, "sh_synth_ref_device: null argument");
    syn_ref_params P = *(const syn_ref_params *)ref_params;
    if (n == 0) return SH_OK;
    uint64_t nt = (n + 3) / 4;
    hipLaunchKernelGGL(k_synth_ref, dim3((uint32_t)((nt + 255) / 256)), dim3(256), 0, (hipStream_t)stream, P, g0, n, d_out);
    SH_HIP(hipGetLastError());
    return SH_OK;
}

extern "C" sh_status sh_synth_reads_device(const void *ref_params, const void *read_params, uint64_t r0, uint64_t n_records,
                                           uint8_t *d_out, uint64_t *d_offsets, void *stream)
{
    SH_CHECK(ref_params && read_params && d_out, SH_ERR_BAD_ARG, "sh_synth_reads_device: null argument");
    syn_ref_params P = *(const syn_ref_params *)ref_params;
    syn_read_params R = *(const syn_read_params *)read_params;
    if (n_records == 0) return SH_OK;
    uint64_t nt = n_records * ((R.read_len + 3) / 4);
    SH_CHECK((nt + 255) / 256 < (1ull << 31), SH_ERR_BAD_ARG, "too many records for one launch");
    hipLaunchKernelGGL(k_synth_reads, dim3((uint32_t)((nt + 255) / 256)), dim3(256), 0, (hipStream_t)stream, P, R, r0, n_records, d_out, d_offsets);
    SH_HIP(hipGetLastError());
    return SH_OK;
}

// long reads: offsets are computed on the host (lengths from syn_long_len), one thread per 4 bases
__global__ void k_synth_long(syn_ref_params P, syn_read_params R, uint64_t r0, uint64_t n_rec, const uint64_t *offsets, uint8_t *out)
{
    const uint64_t total = offsets[n_rec];
    uint64_t p = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (p >= total) return;
    uint64_t lo = 0, hi = n_rec;          // offsets[lo] <= p < offsets[hi]
    while (hi - lo > 1) { uint64_t mid = (lo + hi) >> 1; if (offsets[mid] <= p) lo = mid; else hi = mid; }
    for (int b = 0; b < 4 && p + b < total; ++b) {
        uint64_t q = p + b;
        while (q >= offsets[lo + 1]) ++lo;
        out[q] = syn_long_read_base(&P, &R, r0 + lo, (uint32_t)(offsets[lo + 1] - offsets[lo]), (uint32_t)(q - offsets[lo]));
    }
}

extern "C" sh_status sh_synth_long_reads_device(const void *ref_params, const void *read_params, uint64_t r0, uint64_t n_records,
                                                const uint64_t *d_offsets, uint64_t n_bases, uint8_t *d_out, void *stream)
{
    SH_CHECK(ref_params && read_params && d_offsets && d_out, SH_ERR_BAD_ARG, "sh_synth_long_reads_device: null argument");
    if (n_records == 0 || n_bases == 0) return SH_OK;
    uint64_t nt = (n_bases + 3) / 4;
    SH_CHECK((nt + 255) / 256 < (1ull << 31), SH_ERR_BAD_ARG, "too many bases for one launch");
    hipLaunchKernelGGL(k_synth_long, dim3((uint32_t)((nt + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *(const syn_ref_params *)ref_params,
                       *(const syn_read_params *)read_params, r0, n_records, d_offsets, d_out);
    SH_HIP(hipGetLastError());
    return SH_OK;
}

// ---- depleted-record bitmap: flags -> 1 bit per record, for the one exchange of the sharded path -------------
// one wave per 64 records: the ballot of (flag == 1) IS the 8 bytes of the bitmap (bit i of byte j = record 8j + i, as
// scrubby_amd/dist.py's pack_flags lays it out)
__global__ __launch_bounds__(256) void k_pack_flags(const uint8_t *flags, uint64_t n, uint8_t *bits)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool host = i < n && flags[i] == 1;
    const unsigned long long m = __ballot(host);
    if ((threadIdx.x & 63) == 0) {
        const uint64_t byte0 = i >> 3, n_bytes = (n + 7) >> 3;
        for (int b = 0; b < 8 && byte0 + b < n_bytes; ++b) bits[byte0 + b] = (uint8_t)(m >> (8 * b));
    }
}

extern "C" sh_status sh_pack_flags_device(const uint8_t *d_flags, uint64_t n, uint8_t *d_bits, void *stream)
{
    SH_CHECK(d_flags && d_bits, SH_ERR_BAD_ARG, "sh_pack_flags_device: null argument");
    if (n == 0) return SH_OK;
    SH_CHECK((n + 255) / 256 < (1ull << 31), SH_ERR_BAD_ARG, "too many records for one launch");
    hipLaunchKernelGGL(k_pack_flags, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_flags, n, d_bits);
    SH_HIP(hipGetLastError());
    return SH_OK;
}

// ---- gather micro-benchmark: the practical ceiling for 16-B random probes into this table ----------------
__global__ void k_gather(const uint4 *slots, uint32_t lg, uint64_t n, uint64_t seed, unsigned long long *sink)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (; i < n; i += stride * 4) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            uint64_t j = i + u * stride;
            uint64_t h = syn_mix(seed ^ j) >> (64 - lg);
            v[u] = j < n ? slots[h] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) atomicAdd(sink, 1ull);
}

extern "C" sh_status sh_bench_gather(const sh_index *idx, uint64_t n_probes, int32_t iters, double *out_gbs_useful, double *out_ms)
{
    SH_CHECK(idx && out_gbs_useful && out_ms && iters > 0, SH_ERR_BAD_ARG, "sh_bench_gather: bad argument");
    SH_HIP(hipSetDevice(idx->device));
    unsigned long long *sink = nullptr;
    SH_HIP(hipMalloc(&sink, 8));
    hipEvent_t e0, e1;
    SH_HIP(hipEventCreate(&e0)); SH_HIP(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_gather, dim3(256 * 16), dim3(256), 0, nullptr, (const uint4 *)idx->d_slots, idx->lg_slots, n_probes, 1ull, sink);
    SH_HIP(hipEventRecord(e0, nullptr));
    for (int it = 0; it < iters; ++it)
        hipLaunchKernelGGL(k_gather, dim3(256 * 16), dim3(256), 0, nullptr, (const uint4 *)idx->d_slots, idx->lg_slots, n_probes, 2ull + it, sink);
    SH_HIP(hipEventRecord(e1, nullptr));
    SH_HIP(hipEventSynchronize(e1));
    float ms = 0;
    SH_HIP(hipEventElapsedTime(&ms, e0, e1));
    *out_ms = ms / iters;
    *out_gbs_useful = 16.0 * (double)n_probes / (*out_ms * 1e-3) / 1e9;
    hipEventDestroy(e0); hipEventDestroy(e1); hipFree(sink);
    return SH_OK;
}


// ---- test aid: the long join's tree on the device (sh_rmq_tree.h), one lane, against the oracle's answers --------------------------------
#include "sh_rmq_tree.h"
template <class TT>
__device__ inline long long dbg_rmq_trace_on(TT &T, uint64_t seed, int32_t n_ops, int32_t key_range, int32_t fifo, int32_t *ly, int32_t *li, long long *out)
{
    long long n_out = 0, head = 0, n_all = 0;
    uint64_t s = seed * 0x9E3779B97F4A7C15ULL + 1;
#define RND() (s ^= s << 13, s ^= s >> 7, s ^= s << 17, s)
    for (int op = 0; op < n_ops; ++op) {
        const unsigned r = (unsigned)(RND() % 10);
        const long long n_live = n_all - head;
        if (r < 5 || n_live == 0) {
            const int32_t x = rq_alloc(T);
            if (x == RQ_NIL) return -100;
            const int32_t y = (int32_t)(RND() % (uint64_t)key_range); const double pri = (double)(RND() % 10);
            rq_node_set(T, x, y, op, pri);
            ly[n_all] = y; li[n_all] = op; ++n_all;
            rq_insert(T, x);
        } else if (r < 7) {
            const long long k = fifo ? head : head + (long long)(RND() % (uint64_t)n_live);
            const int32_t e = rq_erase(T, ly[k], li[k]);
            if (e != RQ_NIL) rq_free(T, e);
            ly[k] = ly[head]; li[k] = li[head]; ++head;
        } else {
            int32_t a = (int32_t)(RND() % (uint64_t)key_range), b = (int32_t)(RND() % (uint64_t)key_range);
            if (a > b) { const int32_t tt = a; a = b; b = tt; }
            const int32_t q = rq_rmq(T, a, INT32_MAX, b, 0);
            out[n_out++] = q == RQ_NIL ? -1 : rq_i(T, q);
        }
    }
#undef RND
    return T.bad ? -(long long)T.bad : n_out;
}
// the same sequence with the insertions and erasures done by the whole wave (rq_insert_w / rq_erase_w); queries on lane 0
__device__ inline long long dbg_rmq_trace_wave(RqTreeT<RqLds> &T, uint64_t seed, int32_t n_ops, int32_t key_range, int32_t fifo, int32_t *ly, int32_t *li, long long *out)
{
    const int lane = rqw_lane();
    long long n_out = 0, head = 0, n_all = 0;
    uint64_t s = seed * 0x9E3779B97F4A7C15ULL + 1;
#define RND() (s ^= s << 13, s ^= s >> 7, s ^= s << 17, s)
    for (int op = 0; op < n_ops; ++op) {
        const unsigned r = (unsigned)(RND() % 10);
        const long long n_live = n_all - head;
        if (r < 5 || n_live == 0) {
            const int32_t y = (int32_t)(RND() % (uint64_t)key_range); const double pri = (double)(RND() % 10);
            if (rq_insert_w(T, y, op, pri) == RQ_NIL) return -100;
            if (lane == 0) { ly[n_all] = y; li[n_all] = op; }
            ++n_all;
        } else if (r < 7) {
            const long long k = fifo ? head : head + (long long)(RND() % (uint64_t)n_live);
            // (lane 0's global stores of ly / li are read back by every lane: stores drained, L1 invalidated)
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\tbuffer_inv sc1" ::: "memory");
            const int32_t ky = __atomic_load_n(ly + k, __ATOMIC_RELAXED), ki = __atomic_load_n(li + k, __ATOMIC_RELAXED);
            const int32_t e = rq_erase_w(T, ky, ki);
            if (e != RQ_NIL) rqw_free(T, e);
            if (lane == 0) { ly[k] = ly[head]; li[k] = li[head]; }
            ++head;
        } else {
            int32_t a = (int32_t)(RND() % (uint64_t)key_range), b = (int32_t)(RND() % (uint64_t)key_range);
            if (a > b) { const int32_t tt = a; a = b; b = tt; }
            int32_t ans = 0;
            if (lane == 0) { const int32_t q = rq_rmq(T, a, INT32_MAX, b, 0); ans = q == RQ_NIL ? -1 : rq_i(T, q); out[n_out] = ans; }
            ++n_out;
        }
    }
#undef RND
    return T.bad ? -(long long)T.bad : n_out;
}
__global__ void k_dbg_rmq_trace(uint64_t seed, int32_t n_ops, int32_t key_range, int32_t fifo, int32_t lds, RqNode *pool, int32_t *ly, int32_t *li, long long *out, long long *n_out_p)
{
    __shared__ RqLdsMem<4096> s_m;
    if (lds == 2) {      // every lane takes part
        const unsigned long long t0 = wall_clock64();
        RqTreeT<RqLds> T;
        T.st.init(s_m); rq_reset(T);
        const long long n = dbg_rmq_trace_wave(T, seed, n_ops, key_range, fifo, ly, li, out);
        if (threadIdx.x == 0) { if (n >= 0 && n < n_ops - 1) out[n_ops - 1] = (long long)(wall_clock64() - t0); *n_out_p = n; }
        return;
    }
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = wall_clock64();
    long long n = 0;
    if (lds) {
        RqTreeT<RqLds> T;
        T.st.init(s_m); rq_reset(T);
        n = dbg_rmq_trace_on(T, seed, n_ops, key_range, fifo, ly, li, out);
    } else {
        RqTree T;
        rq_init(T, pool, n_ops + 4);
        n = dbg_rmq_trace_on(T, seed, n_ops, key_range, fifo, ly, li, out);
    }
    if (n >= 0 && n < n_ops - 1) out[n_ops - 1] = (long long)(wall_clock64() - t0);      // (behind the answers: the sequence's time in 100-MHz ticks, for whoever wants it)
    *n_out_p = n;
}
extern "C" sh_status sh_dbg_rmq_trace(int32_t device, uint64_t seed, int32_t n_ops, int32_t key_range, int32_t fifo, int32_t cache, int64_t *out, int64_t *n_out)
{
    SH_CHECK(out && n_out && n_ops > 0 && key_range > 0 && cache >= 0, SH_ERR_BAD_ARG, "sh_dbg_rmq_trace: bad argument");
    SH_HIP(hipSetDevice(device));
    RqNode *pool = nullptr; int32_t *ly = nullptr, *li = nullptr; long long *d_out = nullptr, *d_n = nullptr;
    SH_HIP(hipMalloc(&pool, sizeof(RqNode) * ((size_t)n_ops + 4))); SH_HIP(hipMalloc(&ly, 4 * ((size_t)n_ops + 1))); SH_HIP(hipMalloc(&li, 4 * ((size_t)n_ops + 1)));
    SH_HIP(hipMalloc(&d_out, 8 * (size_t)n_ops)); SH_HIP(hipMalloc(&d_n, 8));
    hipLaunchKernelGGL(k_dbg_rmq_trace, dim3(1), dim3(64), 0, 0, seed, n_ops, key_range, fifo, cache, pool, ly, li, d_out, d_n);
    SH_HIP(hipDeviceSynchronize());
    long long n = 0;
    SH_HIP(hipMemcpy(&n, d_n, 8, hipMemcpyDeviceToHost));
    if (n > 0) SH_HIP(hipMemcpy(out, d_out, 8 * (size_t)n_ops, hipMemcpyDeviceToHost));
    *n_out = n;
    hipFree(pool); hipFree(ly); hipFree(li); hipFree(d_out); hipFree(d_n);
    return SH_OK;
}

// ---- test aid: the wave primitives of sh_wave.h against plain arithmetic (tests/test_wave_ops_gpu.py) -----------------------------------
#include "sh_wave.h"
__global__ void k_dbg_wave_ops(const int32_t *in32, const unsigned long long *in64, int32_t bl, int32_t *o32, unsigned long long *o64)
{
    const int lane = threadIdx.x;
    const int32_t v = in32[lane];
    const unsigned long long w = in64[lane];
    int32_t r32[12];
    r32[0] = wave_scan_max_incl(v); r32[1] = wave_scan_min_incl(v); r32[2] = wave_scan_add_incl(v); r32[3] = wave_scan_or_incl(v);
    r32[4] = wave_shr1(v, -7); r32[5] = wave_all_max(v); r32[6] = wave_all_min(v); r32[7] = wave_all_add(v);
    r32[8] = (int32_t)wave_all_or((uint32_t)v); r32[9] = (int32_t)wave_all_max_u32((uint32_t)v); r32[10] = (int32_t)wave_all_min_u32((uint32_t)v);
    r32[11] = wave_bcast(v, bl);
    for (int i = 0; i < 12; ++i) o32[i * 64 + lane] = r32[i];
    unsigned long long r64[9];
    r64[0] = (unsigned long long)wave_scan_max_incl_i64((long long)w); r64[1] = wave_scan_max_incl_u64(w); r64[2] = wave_scan_min_incl_u64(w);
    r64[3] = wave_scan_add_incl_u64(w); r64[4] = (unsigned long long)wave_all_max_i64((long long)w); r64[5] = wave_all_max_u64(w);
    r64[6] = wave_all_min_u64(w); r64[7] = wave_bcast_u64(w, bl); r64[8] = wave_shr1_u64(w, 99ull);
    for (int i = 0; i < 9; ++i) o64[i * 64 + lane] = r64[i];
}
extern "C" sh_status sh_dbg_wave_ops(int32_t device, const int32_t *in32, const uint64_t *in64, int32_t bcast_lane, int32_t *out32, uint64_t *out64)
{
    SH_CHECK(in32 && in64 && out32 && out64 && bcast_lane >= 0 && bcast_lane < 64, SH_ERR_BAD_ARG, "sh_dbg_wave_ops: bad argument");
    SH_HIP(hipSetDevice(device));
    int32_t *d_i = nullptr, *d_o = nullptr; unsigned long long *d_i64 = nullptr, *d_o64 = nullptr;
    SH_HIP(hipMalloc(&d_i, 256)); SH_HIP(hipMalloc(&d_i64, 512)); SH_HIP(hipMalloc(&d_o, 12 * 256)); SH_HIP(hipMalloc(&d_o64, 9 * 512));
    SH_HIP(hipMemcpy(d_i, in32, 256, hipMemcpyHostToDevice)); SH_HIP(hipMemcpy(d_i64, in64, 512, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_dbg_wave_ops, dim3(1), dim3(64), 0, 0, d_i, d_i64, bcast_lane, d_o, d_o64);
    SH_HIP(hipDeviceSynchronize());
    SH_HIP(hipMemcpy(out32, d_o, 12 * 256, hipMemcpyDeviceToHost)); SH_HIP(hipMemcpy(out64, d_o64, 9 * 512, hipMemcpyDeviceToHost));
    hipFree(d_i); hipFree(d_i64); hipFree(d_o); hipFree(d_o64);
    return SH_OK;
}
