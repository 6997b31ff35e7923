// sh_common.h — internal declarations shared by the translation units of libscrubby_hip.so.
#pragma once
#include <cstring>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>
#include <mutex>
#include <hip/hip_runtime.h>
#include "../../include/scrubby_hip.h"

// ---- HBM index layout ---------------------------------------------------------
// slot = 16 B {w0, w1}; w0 = ~0 (empty) | key (2k bits <= 56) | multi << 63
//                       w1 = position word rid<<32|pos<<1|strand  (singleton)
//                            off << 28 | n                        (multi: n entries at positions[off])
#define SH_SLOT_EMPTY 0xFFFFFFFFFFFFFFFFULL
#define SH_SLOT_MULTI (1ULL << 63)
#define SH_SLOT_KEYMASK ((1ULL << 56) - 1)
#define SH_SLOT_NBITS 28
#define SH_SLOT_NMASK ((1ULL << SH_SLOT_NBITS) - 1)
// seed record (uint4): x,y = slot payload w1, z = occurrence count | flags, w = qpos << 1 | strand
//   z bit 31: filtered by mm_seed_select (set by the consumers that store their verdict)
//   z bit 30: the minimizer emitted just before this one - in mm_sketch's order, after mm_seed_mz_flt - has the same hash.  A seed is
//             "tandem" (mm_seed_collect_all: MM_SEED_TANDEM on its anchors) when this bit is set on it or on the record behind it.
#define SH_REC_PREV_SAME (1u << 30)
#define SH_REC_OCC_MASK 0x3fffffffu

struct sh_index {
    int32_t device = 0;
    int32_t k = 0, w = 0;
    int32_t mid_occ = 0;          // resolved
    uint32_t n_contigs = 0;
    uint64_t n_bases = 0;
    uint64_t n_minimizers = 0, n_keys = 0, n_slots = 0, n_positions = 0;
    uint32_t lg_slots = 0;
    uint64_t *d_slots = nullptr;      // 2 * n_slots
    uint64_t *d_positions = nullptr;  // n_positions (+1)
    uint8_t *d_ref = nullptr;         // reference bases as 4-bit nt4 codes (mi->S of minimap2), (n_bases + 1) / 2 bytes + 16 of padding
    uint64_t *d_cstart = nullptr;     // n_contigs + 1: first base of each contig in d_ref
    std::vector<uint64_t> contig_len;
    // occurrence parameters mid_occ was resolved with (mm_mapopt_update), and the checksum of the packed reference: the identity a cache is checked by
    int32_t o_mid_occ = 0, o_min_mid_occ = 0, o_max_mid_occ = 0; float o_mid_occ_frac = 0;
    uint64_t ref_checksum = 0;
    double build_ms = 0;
    mutable std::mutex pool_mu;          // scratch that sh_classify_batch calls leave behind for the next one (sh_api.hip)
    mutable std::vector<void *> pool;
};

// ---- errors ----------------------------------------------------------------------------------
void sh_set_error(const char *fmt, ...);
#define SH_HIP(call)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            sh_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
            return e_ == hipErrorOutOfMemory ? SH_ERR_OOM : SH_ERR_HIP;                      \
        }                                                                                    \
    } while (0)
#define SH_CHECK(cond, code, ...)         \
    do {                                  \
        if (!(cond)) {                    \
            sh_set_error(__VA_ARGS__);    \
            return (code);                \
        }                                 \
    } while (0)

// ---- device helpers --------------------------------------------------------------------------
__host__ __device__ static inline uint64_t sh_hash64(uint64_t key, uint64_t mask)
{   // minimap2's invertible integer mix on 2k bits
    key = (~key + (key << 21)) & mask;
    key = key ^ key >> 24;
    key = ((key + (key << 3)) + (key << 8)) & mask;
    key = key ^ key >> 14;
    key = ((key + (key << 2)) + (key << 4)) & mask;
    key = key ^ key >> 28;
    key = (key + (key << 31)) & mask;
    return key;
}

__host__ __device__ static inline uint64_t sh_slot_home(uint64_t key, uint32_t lg)
{
    return (key * 0x9E3779B97F4A7C15ULL) >> (64 - lg);
}

// ASCII -> 0..3 (A,C,G,T/U, either case), 4 otherwise
__host__ __device__ static inline uint32_t sh_nt4(uint32_t c)
{
    uint32_t u = c & 0xDFu;
    bool ok = (u == 'A') | (u == 'C') | (u == 'G') | (u == 'T') | (u == 'U');
    return ok ? ((c >> 1) ^ (c >> 2)) & 3u : 4u;
}

// internal entry points implemented across translation units
void shi_batch_pool_release(sh_index *idx);
sh_status shi_index_build_fasta_host(const char *path, const sh_opts *opts, int32_t device, sh_index **out);
sh_status shi_index_build_device(const uint8_t *d_bases, const uint64_t *contig_starts, uint32_t n_contigs,
                                 const sh_opts *opts, int32_t device, hipStream_t stream, sh_index **out);

// sh_classify.hip: a context re-pointed at another index of the same shape (the streaming host path keeps one between runs)
sh_status shi_ctx_rebind(sh_ctx *c, const sh_index *idx);
uint64_t shi_ctx_max_reads(const sh_ctx *c);
uint64_t shi_ctx_max_bases(const sh_ctx *c);
uint32_t shi_ctx_max_len(const sh_ctx *c);
