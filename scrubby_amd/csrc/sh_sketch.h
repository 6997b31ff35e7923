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
