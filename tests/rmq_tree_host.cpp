// Host build of the PRODUCT's tree code (scrubby_amd/csrc/sh_rmq_tree.h) for tests/test_rmq_tree_cpu.py: the same random operation
// sequence as the oracle's mmo_rmq_trace, answered by the device restatement - on either storage: lds == 0 the 32-byte node pool (RqPool,
// what lives in HBM on the device), lds != 0 the structure-of-arrays form with 16-bit links (RqLds, what lives in LDS on the device).
#include <cstdint>
#include <climits>
#include <cstddef>
#include <vector>
#include <memory>
#define __device__
#include "../scrubby_amd/csrc/sh_rmq_tree.h"

constexpr int HOST_LDS_CAP = 32768;

template <class T>
static int64_t trace_on(T &Tr, uint64_t seed, int n_ops, int key_range, int fifo, int64_t *out)
{
    std::vector<int32_t> ly((size_t)n_ops + 1), li((size_t)n_ops + 1);
    int64_t n_out = 0, head = 0, n_all = 0;
    uint64_t s = seed * 0x9E3779B97F4A7C15ULL + 1;
#define RND() (s ^= s << 13, s ^= s >> 7, s ^= s << 17, s)
    for (int op = 0; op < n_ops; ++op) {
        const unsigned r = (unsigned)(RND() % 10);
        const int64_t n_live = n_all - head;
        if (r < 5 || n_live == 0) {
            const int32_t x = rq_alloc(Tr);
            if (x == RQ_NIL) return -100;
            const int32_t y = (int32_t)(RND() % (uint64_t)key_range); const double pri = (double)(RND() % 10);
            rq_node_set(Tr, x, y, op, pri);
            ly[n_all] = y; li[n_all] = op; ++n_all;
            rq_insert(Tr, x);
        } else if (r < 7) {
            const int64_t k = fifo ? head : head + (int64_t)(RND() % (uint64_t)n_live);
            const int32_t e = rq_erase(Tr, ly[k], li[k]);
            if (e != RQ_NIL) rq_free(Tr, e);
            ly[k] = ly[head]; li[k] = li[head]; ++head;
        } else {
            int32_t a = (int32_t)(RND() % (uint64_t)key_range), b = (int32_t)(RND() % (uint64_t)key_range);
            if (a > b) { const int32_t tt = a; a = b; b = tt; }
            const int32_t q = rq_rmq(Tr, a, INT32_MAX, b, 0);
            out[n_out++] = q == RQ_NIL ? -1 : rq_i(Tr, q);
        }
        if (rq_size(Tr) != (int32_t)(n_all - head)) return -101;      // krmq_size(root) = the live nodes
    }
#undef RND
    return Tr.bad ? -(int64_t)Tr.bad : n_out;
}

extern "C" int64_t rqh_trace(uint64_t seed, int n_ops, int key_range, int fifo, int lds, int64_t *out)
{
    if (!lds) {
        std::vector<RqNode> pool((size_t)n_ops + 4);
        RqTree T;
        rq_init(T, pool.data(), n_ops + 4);
        return trace_on(T, seed, n_ops, key_range, fifo, out);
    }
    auto mem = std::make_unique<RqLdsMem<HOST_LDS_CAP>>();
    RqTreeT<RqLds> T;
    T.st.init(*mem); rq_reset(T);
    return trace_on(T, seed, n_ops, key_range, fifo, out);
}

// mg_lchain_rmq's scoring pass on the product's trees (the loop of lr_rmq_fill_tree in scrubby_amd/csrc/sh_long.h, restated for the host):
// a[] = (x, y) pairs sorted by x.  Returns 0, or the code of the guard that tripped (tree 0: code, tree 1: 20 + code).
static inline float h_log2(float x)
{
    union { float f; uint32_t i; } z = { x };
    float log_2 = (float)((int32_t)((z.i >> 23) & 255) - 128);
    z.i &= ~(255u << 23); z.i += 127u << 23;
    log_2 += (-0.34484843f * z.f + 2.02466578f) * z.f - 0.67487759f;
    return log_2;
}
static inline int32_t h_sc(int32_t dr, int32_t dq, int32_t q_span, float pen_gap, float pen_skip, int32_t &exact, int32_t &width)
{
    const int32_t dd = dr > dq ? dr - dq : dq - dr, dg = dr < dq ? dr : dq;
    int32_t sc = q_span < dg ? q_span : dg;
    width = dd; exact = (dd == 0 && dg <= q_span);
    if (dd || dq > q_span) { const float lin = pen_gap * (float)dd + pen_skip * (float)dg; const float lg = dd >= 1 ? h_log2((float)(dd + 1)) : 0.0f; sc -= (int32_t)(lin + .5f * lg); }
    return sc;
}
template <class TT>
static int lchain_fill_on(TT &T0, TT &T1, int max_dist, int max_dist_inner, int bw, int max_chn_skip, int cap_rmq_size, float pen_gap, float pen_skip,
                          int32_t n, const uint64_t *a /* x, y interleaved */, int32_t *f, int32_t *p, int32_t *t)
{
    if (max_dist < bw) max_dist = bw;
    if (max_dist_inner < 0) max_dist_inner = 0;
    if (max_dist_inner > max_dist) max_dist_inner = max_dist;
    auto X = [&](int32_t i) { return a[2 * (size_t)i]; };
    auto Y = [&](int32_t i) { return a[2 * (size_t)i + 1]; };
    auto root_size = [](const TT &tr) -> int32_t { return rq_size(tr); };
    int32_t i, i0, st = 0, st_inner = 0;
    for (i = 0; i < n; ++i) t[i] = 0;
    for (i = i0 = 0; i < n; ++i) {
        int32_t max_j = -1;
        const uint64_t xi = X(i); const int32_t yi = (int32_t)Y(i);
        const int32_t q_span = (int32_t)(Y(i) >> 32 & 0xff);
        int32_t max_f = q_span;
        if (i0 < i && X(i0) != xi) {
            for (int32_t j = i0; j < i; ++j) {
                const double pri = -((double)f[j] + 0.5 * (double)pen_gap * (double)((int32_t)X(j) + (int32_t)Y(j)));
                for (int k = 0; k < (max_dist_inner > 0 ? 2 : 1); ++k) {
                    TT &T = k ? T1 : T0;
                    const int32_t x = rq_alloc(T);
                    if (x == RQ_NIL) return 12 + 20 * k;
                    rq_node_set(T, x, (int32_t)Y(j), j, pri);
                    rq_insert(T, x);
                }
            }
            i0 = i;
        }
        while (st < i && (xi >> 32 != X(st) >> 32 || xi > X(st) + (uint64_t)max_dist || root_size(T0) > cap_rmq_size)) { const int32_t e = rq_erase(T0, (int32_t)Y(st), st); if (e != RQ_NIL) rq_free(T0, e); ++st; }
        if (max_dist_inner > 0)
            while (st_inner < i && (xi >> 32 != X(st_inner) >> 32 || xi > X(st_inner) + (uint64_t)max_dist_inner || root_size(T1) > cap_rmq_size)) { const int32_t e = rq_erase(T1, (int32_t)Y(st_inner), st_inner); if (e != RQ_NIL) rq_free(T1, e); ++st_inner; }
        const int32_t q = rq_rmq(T0, yi - max_dist, INT32_MAX, yi, 0);
        if (q != RQ_NIL) {
            int32_t sc, exact, width, n_skip = 0;
            int32_t j = rq_i(T0, q);
            sc = f[j] + h_sc((int32_t)(xi - X(j)), yi - (int32_t)Y(j), (int32_t)(Y(j) >> 32 & 0xff), pen_gap, pen_skip, exact, width);
            if (width <= bw && sc > max_f) { max_f = sc; max_j = j; }
            if (!exact && T1.root != RQ_NIL && yi > 0) {
                RqItr it;
                if (rq_itr_find_le(T1, yi - 1, n, it)) {
                    do {
                        const int32_t e = it.stack[it.top];
                        if (rq_y(T1, e) < yi - max_dist_inner) break;
                        j = rq_i(T1, e);
                        int32_t ex2;
                        sc = f[j] + h_sc((int32_t)(xi - X(j)), yi - (int32_t)Y(j), (int32_t)(Y(j) >> 32 & 0xff), pen_gap, pen_skip, ex2, width);
                        if (width <= bw) {
                            if (sc > max_f) { max_f = sc; max_j = j; if (n_skip > 0) --n_skip; }
                            else if (t[j] == i) { if (++n_skip > max_chn_skip) break; }
                            if (p[j] >= 0) t[p[j]] = i;
                        }
                    } while (rq_itr_prev(T1, it));
                }
            }
        }
        f[i] = max_f; p[i] = max_j;
        if (T0.bad) return T0.bad;
        if (T1.bad) return 20 + T1.bad;
    }
    return 0;
}

extern "C" int rqh_lchain_fill(int max_dist, int max_dist_inner, int bw, int max_chn_skip, int cap_rmq_size, float pen_gap, float pen_skip,
                               int32_t n, const uint64_t *a, int32_t *f, int32_t *p, int32_t *t, int lds)
{
    if (!lds) {
        std::vector<RqNode> pool0((size_t)n + 2), pool1((size_t)n + 2);
        RqTree T0, T1;
        rq_init(T0, pool0.data(), n + 2); rq_init(T1, pool1.data(), n + 2);
        return lchain_fill_on(T0, T1, max_dist, max_dist_inner, bw, max_chn_skip, cap_rmq_size, pen_gap, pen_skip, n, a, f, p, t);
    }
    auto m0 = std::make_unique<RqLdsMem<HOST_LDS_CAP>>(), m1 = std::make_unique<RqLdsMem<HOST_LDS_CAP>>();
    RqTreeT<RqLds> T0, T1;
    T0.st.init(*m0); rq_reset(T0); T1.st.init(*m1); rq_reset(T1);
    return lchain_fill_on(T0, T1, max_dist, max_dist_inner, bw, max_chn_skip, cap_rmq_size, pen_gap, pen_skip, n, a, f, p, t);
}
