// sh_sketch.h — minimizer sketch state machine for gfx950 (device code).
//
// Restates the (w,k)-minimizer selection the reference reaches through
// minimap2::Aligner::map / with_index (/root/reference/src/cleaner.rs:472-482, :552;
// SURVEY.md App. A.2): rolling forward/reverse k-mer, invertible hash on 2k bits, window
// minimum with "rightmost on ties", every identical k-mer of a window emitted once, pending
// minimum flushed at the end.  One lane owns one sequence (or one segment of a sequence).
//
// MI355X mapping: the w-entry ring buffer lives in VGPRs.  The driver unrolls the base loop
// by W so that the ring slot of every step is a compile-time constant (a runtime-indexed
// private array would be demoted to scratch memory: cdna_hip_programming.md §5.4 rule 20).
// That requires buf_pos == step mod W in every lane, which holds because k is odd for every
// minimap2 preset: a k-mer can then never equal its reverse complement, so the reference's
// `continue` on strand-ambiguous k-mers (the only thing that stalls buf_pos) cannot fire.
// Even k is rejected at the API (SH_ERR_BAD_ARG).
#pragma once
#include "sh_common.h"
#include <utility>

#define SH_XMAX 0xFFFFFFFFFFFFFFFFULL
#define SH_YMAX 0xFFFFFFFFu

template <int W>
struct SketchState {
    uint64_t bx[W];   // hash<<8 | span, or SH_XMAX
    uint32_t by[W];   // pos<<1 | strand, or SH_YMAX
    uint64_t minx, kf, kr, mask;
    uint32_t miny, shift1;
    int32_t min_pos, l, k;

    __device__ inline void init(int k_)
    {
#pragma unroll
        for (int j = 0; j < W; ++j) bx[j] = SH_XMAX, by[j] = SH_YMAX;
        minx = SH_XMAX; miny = SH_YMAX; min_pos = 0; l = 0; kf = kr = 0;
        k = k_; mask = (1ULL << 2 * k_) - 1; shift1 = 2 * (k_ - 1);
    }

    // One base.  P = ring slot (== step index mod W), c = 0..3 or 4 (ambiguous),
    // pos = position of this base in its sequence, emit(x, y) receives minimizers in order.
    template <int P, class Emit>
    __device__ inline void step(uint32_t c, uint32_t pos, Emit &&emit)
    {
        uint64_t ix = SH_XMAX;
        uint32_t iy = SH_YMAX;
        if (c < 4) {
            kf = (kf << 2 | (uint64_t)c) & mask;
            kr = (kr >> 2) | ((uint64_t)(3u ^ c) << shift1);
            uint32_t z = kf < kr ? 0u : 1u;
            ++l;
            if (l >= k) {
                ix = sh_hash64(z ? kr : kf, mask) << 8 | (uint64_t)k;
                iy = pos << 1 | z;
            }
        } else {
            l = 0;
        }
        bx[P] = ix; by[P] = iy;
        if (l == W + k - 1 && minx != SH_XMAX) {      // first full window: identical k-mers (rare: count first)
            int eq = 0;
#pragma unroll
            for (int j = 0; j < W; ++j) if (j != P) eq += (minx == bx[j] && by[j] != miny);
            if (eq) {
#pragma unroll
                for (int j = P + 1; j < W; ++j)
                    if (minx == bx[j] && by[j] != miny) emit(bx[j], by[j]);
#pragma unroll
                for (int j = 0; j < P; ++j)
                    if (minx == bx[j] && by[j] != miny) emit(bx[j], by[j]);
            }
        }
        if (ix <= minx) {                              // new minimum, rightmost on ties
            if (l >= W + k && minx != SH_XMAX) emit(minx, miny);
            minx = ix; miny = iy; min_pos = P;
        } else if (min_pos == P) {                     // old minimum left the window
            if (l >= W + k - 1 && minx != SH_XMAX) emit(minx, miny);
            // rightmost minimum of the window, counting how many entries share it: the tie loops below are
            // skipped (wave-wide, almost always) unless some k-mer repeats inside the window
            minx = SH_XMAX;
            int eq = 0;
#pragma unroll
            for (int j = P + 1; j < W; ++j) {
                const bool lt = bx[j] < minx, ge = !(bx[j] > minx);
                eq = lt ? 1 : (ge ? eq + 1 : eq);
                if (ge) { minx = bx[j]; miny = by[j]; min_pos = j; }
            }
#pragma unroll
            for (int j = 0; j <= P; ++j) {
                const bool lt = bx[j] < minx, ge = !(bx[j] > minx);
                eq = lt ? 1 : (ge ? eq + 1 : eq);
                if (ge) { minx = bx[j]; miny = by[j]; min_pos = j; }
            }
            if (eq > 1 && l >= W + k - 1 && minx != SH_XMAX) {
#pragma unroll
                for (int j = P + 1; j < W; ++j)
                    if (minx == bx[j] && miny != by[j]) emit(bx[j], by[j]);
#pragma unroll
                for (int j = 0; j <= P; ++j)
                    if (minx == bx[j] && miny != by[j]) emit(bx[j], by[j]);
            }
        }
    }

    template <class Emit>
    __device__ inline void finish(Emit &&emit)
    {
        if (minx != SH_XMAX) emit(minx, miny);
    }
};

// Packed variant for the read kernel (K1): reads of at most 1024 bases (positions < 2^17) and k <= 23 (hashes < 2^46), so a ring
// entry fits one 64-bit word   hash << 18 | (0x1FFFF - pos) << 1 | strand   (SH_XMAX = no k-mer).  An unsigned minimum over such
// words is the window minimum with "rightmost on ties" built in (equal hashes: the larger position has the smaller word), so
// the rescan after the minimum leaves the window is a plain 11-way minimum: 3 VALU per entry instead of 10, and no second
// array for y.  "The minimum sits in the slot being overwritten" becomes "its position is pos - W".  Ties (the same k-mer
// twice in a window) are looked for on the hashes' low 32 bits first; only if that count exceeds one do the exact tie loops run, in the
// reference's slot order.  Same statement order and emission order as SketchState::step.
// Round 2: the window minimum is the two-block (van Herk / Gil-Werman) scheme - 2 minima a step + W-1 a block instead of W-1 a step -
// and the separate low-word array is gone; k_sketch_probe is pinned at 4 waves/SIMD (128 VGPRs).  Measured: the VALU count drops, the
// kernel time does not (21.4 -> 21.2 ms for 20 M reads): at 4 waves/SIMD it waits on its ~15 k-instruction loop body, not on the ALUs.
//   emit(p) receives the packed word; sh_packed_entry() turns it into the queue entry  hash << 18 | pos << 1 | strand.
__device__ inline uint64_t sh_packed_entry(uint64_t p)
{
    const uint32_t low = (uint32_t)p & 0x3ffffu;
    return (p & ~0x3ffffULL) | (uint64_t)(((0x1ffffu - (low >> 1)) << 1) | (low & 1u));
}

template <int W>
struct SketchPacked {
    uint64_t b[W];
    uint64_t suf[W];      // suf[j] = min(b[j .. W-1]) of the PREVIOUS block of W steps (block_end)
    uint64_t minp, pre, kf, kr, mask;
    uint32_t shift1;
    int32_t l, k;

    __device__ __forceinline__ void init(int k_)
    {
#pragma unroll
        for (int j = 0; j < W; ++j) b[j] = SH_XMAX, suf[j] = SH_XMAX;
        minp = SH_XMAX; pre = SH_XMAX; l = 0; kf = kr = 0;
        k = k_; mask = (1ULL << 2 * k_) - 1; shift1 = 2 * (k_ - 1);
    }

    template <int P, class Emit>
    __device__ __forceinline__ void ties(Emit &&emit, bool with_p)      // emit: the caller's push for tie entries (may differ from the main one)
    {   // every other entry with the minimum's hash, oldest first: slots P+1 .. W-1, then 0 .. P-1, then P itself (only after a rescan).
        // Which slots tie is a bitmask from W compares; the pushes come from ONE site inside a loop that is deliberately not unrolled
        // (the slot's value is fetched by a select chain): ties are rare, and 2 x W inlined push sites per unrolled step, each with its
        // drain call, would make up three quarters of the read kernel's loop body and push it out of the instruction cache.
        const uint64_t mh = minp >> 18;
        uint32_t mask = 0;
#pragma unroll
        for (int j = 0; j < W; ++j) mask |= (uint32_t)((b[j] >> 18) == mh && b[j] != minp && b[j] != SH_XMAX) << j;
        if (!with_p) mask &= ~(1u << P);
#pragma clang loop unroll(disable)
        for (int t = 1; t <= W && mask != 0; ++t) {
            int slot = P + t;
            if (slot >= W) slot -= W;
            if ((mask >> slot) & 1u) {
                mask &= ~(1u << slot);
                uint64_t v = b[0];
#pragma unroll
                for (int j = 1; j < W; ++j) v = slot == j ? b[j] : v;
                emit(v);
            }
        }
    }

    __device__ __forceinline__ int same_low() const
    {
        const uint32_t ml = (uint32_t)(minp >> 18);
        int c = 0;
#pragma unroll
        for (int j = 0; j < W; ++j) c += (uint32_t)(b[j] >> 18) == ml;
        return c;
    }

    // emit: push of the window minimum (one site per step); tie_emit: push used inside the tie loops (2 x W sites per step).  The read
    // kernel passes the same push for both: these cold sites make up most of its 15 k-instruction loop body (without them it is 4 k
    // and 9 % faster, an instruction-cache effect), but handing tie-heavy reads to the re-sketch path instead costs more than that.
    template <int P, class Emit, class TieEmit>
    __device__ __forceinline__ void step(uint32_t c, uint32_t pos, Emit &&emit, TieEmit &&tie_emit)
    {
        uint64_t ip = SH_XMAX;
        if (c < 4) {
            kf = (kf << 2 | (uint64_t)c) & mask;
            kr = (kr >> 2) | ((uint64_t)(3u ^ c) << shift1);
            const uint32_t z = kf < kr ? 0u : 1u;
            ++l;
            if (l >= k) ip = sh_hash64(z ? kr : kf, mask) << 18 | (uint64_t)(0x1ffffu - pos) << 1 | z;
        } else {
            l = 0;
        }
        b[P] = ip;
        const uint64_t old = minp;
        if (l == W + k - 1 && old != SH_XMAX) {        // first full window: identical k-mers
            if (same_low() > 1) ties<P>(tie_emit, false);
        }
        // The reference's two branches, flattened.  "New minimum" (ix <= minx; an equal hash at a later position is the smaller word)
        // and "old minimum left the window" (its position is pos - W) both push the OLD minimum, under l >= W + k and l >= W + k - 1
        // respectively; and in every case the minimum afterwards is the minimum over the ring (a new minimum is below everything in
        // it, an expired one is no longer in it, otherwise the old one still is).  So: one push site, one unconditional window minimum,
        // no divergent rescan.  Ties are looked for only after an expiry, as in the reference.
        // The window minimum itself is the two-block scheme (van Herk / Gil-Werman): the caller steps in blocks of W with P = step mod W,
        // so at slot P the ring holds slots P+1 .. W-1 of the previous block and 0 .. P of this one - min(suffix minimum of the old block,
        // running minimum of the new one): 2 minima a step + W-1 a block (block_end) instead of W-1 a step.  Same value, bit for bit.
        const bool newmin = ip < old;
        const bool expired = !newmin && old != SH_XMAX && (((uint32_t)old >> 1) & 0x1ffffu) == 0x1ffffu - (pos - (uint32_t)W);
        if ((newmin && l >= W + k && old != SH_XMAX) || (expired && l >= W + k - 1)) emit(old);
        pre = (P == 0 || ip < pre) ? ip : pre;
        uint64_t m = pre;
        if (P + 1 < W) m = suf[P + 1 < W ? P + 1 : 0] < m ? suf[P + 1 < W ? P + 1 : 0] : m;
        minp = m;
        if (expired && l >= W + k - 1 && m != SH_XMAX && same_low() > 1) ties<P>(tie_emit, true);
    }

    // after the W steps of a block (slots 0 .. W-1 all rewritten): the suffix minima the next block combines with
    __device__ __forceinline__ void block_end()
    {
        suf[W - 1] = b[W - 1];
#pragma unroll
        for (int j = W - 2; j >= 1; --j) suf[j] = b[j] < suf[j + 1] ? b[j] : suf[j + 1];
    }

    template <class Emit>
    __device__ __forceinline__ void finish(Emit &&emit)
    {
        if (minp != SH_XMAX) emit(minp);
    }
};

// Runtime-w variant with the ring in caller-provided memory (HBM arena); used only by the
// rare re-sketch path of the large-read kernel.  Same statement order as SketchState::step.
struct SketchStateDyn {
    uint64_t *bx; uint32_t *by;       // w entries each
    uint64_t minx, kf, kr, mask;
    uint32_t miny, shift1;
    int32_t min_pos, buf_pos, l, k, w;

    __device__ inline void init(uint64_t *bx_, uint32_t *by_, int w_, int k_)
    {
        bx = bx_; by = by_; w = w_; k = k_;
        for (int j = 0; j < w; ++j) bx[j] = SH_XMAX, by[j] = SH_YMAX;
        minx = SH_XMAX; miny = SH_YMAX; min_pos = 0; buf_pos = 0; l = 0; kf = kr = 0;
        mask = (1ULL << 2 * k_) - 1; shift1 = 2 * (k_ - 1);
    }
    template <class Emit>
    __device__ inline void step(uint32_t c, uint32_t pos, Emit &&emit)
    {
        uint64_t ix = SH_XMAX;
        uint32_t iy = SH_YMAX;
        const int P = buf_pos;
        if (c < 4) {
            kf = (kf << 2 | (uint64_t)c) & mask;
            kr = (kr >> 2) | ((uint64_t)(3u ^ c) << shift1);
            uint32_t z = kf < kr ? 0u : 1u;
            ++l;
            if (l >= k) { ix = sh_hash64(z ? kr : kf, mask) << 8 | (uint64_t)k; iy = pos << 1 | z; }
        } else l = 0;
        bx[P] = ix; by[P] = iy;
        if (l == w + k - 1 && minx != SH_XMAX) {
            for (int j = P + 1; j < w; ++j) if (minx == bx[j] && by[j] != miny) emit(bx[j], by[j]);
            for (int j = 0; j < P; ++j) if (minx == bx[j] && by[j] != miny) emit(bx[j], by[j]);
        }
        if (ix <= minx) {
            if (l >= w + k && minx != SH_XMAX) emit(minx, miny);
            minx = ix; miny = iy; min_pos = P;
        } else if (min_pos == P) {
            if (l >= w + k - 1 && minx != SH_XMAX) emit(minx, miny);
            minx = SH_XMAX;
            for (int j = P + 1; j < w; ++j) if (minx >= bx[j]) { minx = bx[j]; miny = by[j]; min_pos = j; }
            for (int j = 0; j <= P; ++j) if (minx >= bx[j]) { minx = bx[j]; miny = by[j]; min_pos = j; }
            if (l >= w + k - 1 && minx != SH_XMAX) {
                for (int j = P + 1; j < w; ++j) if (minx == bx[j] && miny != by[j]) emit(bx[j], by[j]);
                for (int j = 0; j <= P; ++j) if (minx == bx[j] && miny != by[j]) emit(bx[j], by[j]);
            }
        }
        if (++buf_pos == w) buf_pos = 0;
    }
    template <class Emit>
    __device__ inline void finish(Emit &&emit) { if (minx != SH_XMAX) emit(minx, miny); }
};
