/*
 * sh_synth_core.h — deterministic synthetic workload, pure position-wise
 * functions shared by the HIP generator kernels (sh_synth.hip) and their CPU twin
 * (oracle/synth.c).  No network on either box, so CHM13v2 and real read sets are replaced by
 * a seeded stand-in whose every base is a pure function of (seed, position): the GPU fills
 * 3.1 Gbp in HBM in milliseconds and tests can regenerate any slice on the CPU.
 *
 * Reference model ("genome" = concatenation of contigs, global coordinate g):
 *   - 2^sb_shift-bp super-blocks; a super-block is a satellite array with probability
 *     sat_pct %: tandem copies of one of n_sat_fam monomers (length 171 / 68 / 5 by family),
 *     each base independently diverged by 2-5 % (rate fixed per array);
 *   - otherwise 2^rb_shift-bp blocks; a block is a copy of one of n_rep_fam interspersed
 *     repeat families with probability rep_pct %, diverged by 5-20 % (rate fixed per copy);
 *   - otherwise iid uniform ACGT.
 * Read model: 2x150 bp FR pairs, fragment length ~N(350,50) clipped to [150,800], 0.5 %
 * substitutions, 1 % of reads carry one N; host_pct % of pairs come from the reference, the
 * rest from an un-indexed iid "microbial" sequence space.
 */
#ifndef SH_SYNTH_CORE_H
#define SH_SYNTH_CORE_H
#include <stdint.h>

#if defined(__HIPCC__)
#define SYN_FN __host__ __device__ static inline
#else
#define SYN_FN static inline
#endif

#define SYN_MAX_CONTIGS 64

typedef struct {
    uint64_t seed;            /* reference seed */
    uint64_t genome_len;      /* total bases */
    uint32_t n_contigs;
    uint32_t sb_shift;        /* satellite super-block = 1<<sb_shift bp (17) */
    uint32_t rb_shift;        /* repeat block = 1<<rb_shift bp (11) */
    uint32_t sat_pct, rep_pct;
    uint32_t n_sat_fam, n_rep_fam;
    uint32_t pad;
    uint64_t contig_start[SYN_MAX_CONTIGS + 1];   /* global start of each contig; [n] = genome_len */
} syn_ref_params;

typedef struct {
    uint64_t seed;            /* read seed */
    uint32_t read_len;        /* 150 */
    uint32_t host_pct;        /* 50 */
    uint32_t sub_per_10k;     /* 50 = 0.5 % */
    uint32_t n_read_pct;      /* percent of reads carrying one N (1) */
} syn_read_params;

SYN_FN uint64_t syn_mix(uint64_t z)
{   /* splitmix64 finaliser */
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

SYN_FN uint32_t syn_sat_mlen(uint32_t fam)
{
    uint32_t c = fam & 7;
    return c == 6 ? 68u : (c == 7 ? 5u : 171u);
}

/* one reference base, 0..3 */
SYN_FN uint32_t syn_ref_base(const syn_ref_params *P, uint64_t g)
{
    uint64_t sb = g >> P->sb_shift;
    uint64_t u = syn_mix(P->seed ^ (sb * 0xD6E8FEB86659FD93ULL + 1));
    uint64_t hm = syn_mix(P->seed ^ 0xA5A5A5A5ULL ^ (g * 0x9FB21C651E98DF25ULL));   /* per-base mutation draw */
    uint32_t cons, div10k;
    if ((uint32_t)(u % 100) < P->sat_pct) {
        uint32_t fam = (uint32_t)((u >> 8) % P->n_sat_fam);
        uint32_t mlen = syn_sat_mlen(fam);
        uint32_t phase = (uint32_t)((g - (sb << P->sb_shift)) % mlen);
        uint64_t c = syn_mix(P->seed ^ 0x5A7E111EULL ^ ((uint64_t)fam << 20) ^ (phase >> 5));
        cons = (uint32_t)(c >> (2 * (phase & 31))) & 3;
        div10k = 200 + (uint32_t)((u >> 24) % 301);              /* 2.00 .. 5.00 % */
    } else {
        uint64_t rb = g >> P->rb_shift;
        uint64_t v = syn_mix(P->seed ^ 0x0123456789ABCDEFULL ^ (rb * 0xC2B2AE3D27D4EB4FULL));
        if ((uint32_t)(v % 100) < P->rep_pct) {
            uint32_t fam = (uint32_t)((v >> 8) % P->n_rep_fam);
            uint32_t off = (uint32_t)(g & ((1ULL << P->rb_shift) - 1));
            uint64_t c = syn_mix(P->seed ^ 0x7E9EA7ULL ^ ((uint64_t)fam << 24) ^ (off >> 5));
            cons = (uint32_t)(c >> (2 * (off & 31))) & 3;
            div10k = 500 + (uint32_t)((v >> 32) % 1501);          /* 5.00 .. 20.00 % */
        } else {
            uint64_t c = syn_mix(P->seed ^ 0x11D11D11DULL ^ (g >> 5));
            return (uint32_t)(c >> (2 * (g & 31))) & 3;
        }
    }
    if ((uint32_t)(hm % 10000) < div10k) cons = (cons + 1 + (uint32_t)((hm >> 40) % 3)) & 3;
    return cons;
}

SYN_FN uint32_t syn_contig_of(const syn_ref_params *P, uint64_t g)
{
    uint32_t lo = 0, hi = P->n_contigs;      /* contig_start[lo] <= g < contig_start[hi] */
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (P->contig_start[mid] <= g) lo = mid; else hi = mid;
    }
    return lo;
}

/* placement of pair `p`: host?, fragment start (global), fragment length, strand flip */
typedef struct { uint32_t is_host, flip, frag_len, pad; uint64_t start; } syn_pair;

SYN_FN syn_pair syn_place_pair(const syn_ref_params *P, const syn_read_params *R, uint64_t p)
{
    syn_pair o;
    uint64_t h = syn_mix(R->seed ^ (p * 0x9E3779B97F4A7C15ULL));
    uint64_t h2 = syn_mix(h ^ 0x51ED270B7A4F1C3DULL);
    uint64_t h3 = syn_mix(h2 ^ 0x2545F4914F6CDD1DULL);
    uint32_t s4, fl;
    o.is_host = (uint32_t)(h % 100) < R->host_pct;
    o.flip = (uint32_t)(h >> 32) & 1;
    /* ~N(350,50): sum of four uniforms on [0,173) has sd ~100/2=... scaled below */
    s4 = (uint32_t)(h2 & 0xffff) % 174 + (uint32_t)((h2 >> 16) & 0xffff) % 174 +
         (uint32_t)((h2 >> 32) & 0xffff) % 174 + (uint32_t)((h2 >> 48) & 0xffff) % 174;   /* mean 346, sd ~100 */
    fl = 350 + ((int32_t)s4 - 346) / 2;                                                   /* sd ~50 */
    if (fl < R->read_len) fl = R->read_len;
    if (fl > 800) fl = 800;
    o.frag_len = fl; o.pad = 0;
    if (o.is_host) {
        uint64_t g = h3 % (P->genome_len - fl);
        uint32_t c = syn_contig_of(P, g);
        uint64_t cend = P->contig_start[c + 1];
        if (g + fl > cend) g = cend - fl;             /* keep the fragment inside one contig */
        if (g < P->contig_start[c]) g = P->contig_start[c];
        o.start = g;
    } else {
        o.start = h3 % ((1ULL << 40) - fl);           /* position in the iid "microbial" space */
    }
    return o;
}

/* base i (0..read_len-1) of mate `mate` (0/1) of pair p, as ASCII */
SYN_FN uint8_t syn_read_base(const syn_ref_params *P, const syn_read_params *R, const syn_pair *pl,
                             uint64_t p, uint32_t mate, uint32_t i)
{
    uint32_t L = R->read_len, fl = pl->frag_len, b;
    /* mate 0 reads the fragment's left end forward, mate 1 its right end reverse-complemented;
       flip swaps which strand the fragment was taken from */
    uint32_t rev = mate ^ pl->flip;
    uint64_t g = rev ? pl->start + fl - 1 - i : pl->start + i;
    uint64_t hr = syn_mix(R->seed ^ 0xBADC0FFEE0DDF00DULL ^ ((p * 2 + mate) * 1024 + i));
    uint64_t hn = syn_mix(R->seed ^ 0x0DDBA11ULL ^ (p * 2 + mate));
    (void)L;
    if (pl->is_host) b = syn_ref_base(P, g);
    else {
        uint64_t c = syn_mix(R->seed ^ 0x3C0FFEE3ULL ^ (g >> 5));
        b = (uint32_t)(c >> (2 * (g & 31))) & 3;
    }
    if (rev) b ^= 3;
    if ((uint32_t)(hr % 10000) < R->sub_per_10k) b = (b + 1 + (uint32_t)((hr >> 40) % 3)) & 3;
    if ((uint32_t)(hn % 100) < R->n_read_pct && (uint32_t)((hn >> 16) % R->read_len) == i) return 'N';
    return (uint8_t)"ACGT"[b];
}

/* ---- long reads (BASELINE config 4 stand-in): log-normal-like lengths, substitutions and (optionally) indels -----------
 * Length of read r: integer-only (bit-reproducible on both sides, no libm): quantile knots of a log-normal(mu = 8.497, sigma = 0.7)
 * clipped to [200, 100 000] (mean ~6.26 kb, median ~4.9 kb, N50 ~8.0 kb), uniform inside a quantile bin; the top
 * sixteenth is cut into sixteenths again, and so is its top, so the tail out to 100 kb keeps its weight (1 read in 4096 beyond 56 kb). */
SYN_FN uint32_t syn_long_len(uint64_t seed, uint64_t r)
{
    const uint64_t h = syn_mix(seed ^ 0x10E6A11ULL ^ (r * 0x9E3779B97F4A7C15ULL));
    const uint32_t k1[17] = {200, 1674, 2190, 2633, 3056, 3480, 3920, 4389, 4900, 5470, 6124, 6899, 7857, 9118, 10963, 14341, 14341};
    const uint32_t k2[17] = {14341, 14672, 15029, 15415, 15838, 16302, 16817, 17395, 18050, 18806, 19697, 20774, 22131, 23941, 26617, 31541, 31541};
    const uint32_t k3[17] = {31541, 32023, 32543, 33106, 33721, 34397, 35146, 35985, 36936, 38032, 39321, 40878, 42834, 45437, 49268, 56273, 100000};
    const uint32_t q1 = (uint32_t)(h & 15), q2 = (uint32_t)(h >> 4) & 15, q3 = (uint32_t)(h >> 32) & 15;
    uint32_t lo, hi;
    if (q1 < 15) { lo = k1[q1]; hi = k1[q1 + 1]; }
    else if (q2 < 15) { lo = k2[q2]; hi = k2[q2 + 1]; }
    else { lo = k3[q3]; hi = k3[q3 + 1]; }
    return lo + (uint32_t)(((h >> 8) & 0xffffff) * (uint64_t)(hi - lo) >> 24);
}

/* base i of long read r (single-end, either strand), host_pct % from the reference.
 * R->n_read_pct != 0 switches indels on (the field is the short-read generator's; long reads have no N reads): along the walk
 * over the reference every 32-base block of the read carries, each with probability 1/2 (one bit of a hash word per block, 64
 * blocks per word), one inserted base and one skipped reference base at hash-chosen offsets - 1.56 % insertions and 1.56 %
 * deletions per base (1.5 % each).  The reference offset of walk position t is t - insertions before + deletions
 * up to t, in closed form: popcounts of the whole words before t's word, of the bits below t's block, and the block's own events. */
SYN_FN uint8_t syn_long_read_base(const syn_ref_params *P, const syn_read_params *R, uint64_t r, uint32_t len, uint32_t i)
{
    const uint64_t h = syn_mix(R->seed ^ 0x0A7ULL ^ (r * 0xD6E8FEB86659FD93ULL));
    const uint32_t is_host = (uint32_t)(h % 100) < R->host_pct, rev = (uint32_t)(h >> 40) & 1;
    const uint64_t h3 = syn_mix(h ^ 0x2545F4914F6CDD1DULL);
    const uint32_t indel = R->n_read_pct != 0;
    const uint32_t t = rev ? len - 1 - i : i;                 /* walk order along the reference */
    const uint32_t span = indel ? len + (len >> 5) + 64 : len;  /* reference bases the read can cover */
    int64_t ref_off = t;
    uint32_t inserted = 0, ins_base = 0;
    if (indel) {
        const uint32_t blk = t >> 5, grp = blk >> 6, bit = blk & 63, o = t & 31;
        const uint64_t kr = r * 0xA24BAED4963EE407ULL;
        int64_t shift = 0;
        uint32_t g;
        for (g = 0; g < grp; ++g)
            shift += (int64_t)__builtin_popcountll(syn_mix(R->seed ^ 0xDE1E7EULL ^ (kr + g))) - (int64_t)__builtin_popcountll(syn_mix(R->seed ^ 0x1A5E27ULL ^ (kr + g)));
        {
            const uint64_t ins_w = syn_mix(R->seed ^ 0x1A5E27ULL ^ (kr + grp)), del_w = syn_mix(R->seed ^ 0xDE1E7EULL ^ (kr + grp));
            const uint64_t below = bit ? (~0ULL >> (64 - bit)) : 0ULL;
            const uint64_t hb = syn_mix(R->seed ^ 0xB10C0FF5ULL ^ (kr + blk));
            const uint32_t io = (uint32_t)hb & 31, dofs = (uint32_t)(hb >> 8) & 31;
            shift += (int64_t)__builtin_popcountll(del_w & below) - (int64_t)__builtin_popcountll(ins_w & below);
            if ((ins_w >> bit) & 1) { if (io < o) shift -= 1; else if (io == o) { inserted = 1; ins_base = (uint32_t)(hb >> 16) & 3; } }
            if (((del_w >> bit) & 1) && dofs <= o) shift += 1;
        }
        ref_off = (int64_t)t + shift;
    }
    uint64_t start;
    uint32_t b;
    if (is_host) {
        uint64_t g = h3 % (P->genome_len - span);
        uint32_t c = syn_contig_of(P, g);
        uint64_t cend = P->contig_start[c + 1];
        if (g + span > cend) g = cend > span ? cend - span : 0;
        if (g < P->contig_start[c]) g = P->contig_start[c];       /* contig shorter than the read: it runs into the next one */
        start = g;
        b = inserted ? ins_base : syn_ref_base(P, start + (uint64_t)ref_off);
    } else {
        start = h3 % ((1ULL << 40) - span);
        const uint64_t g = start + (uint64_t)ref_off;
        const uint64_t c = syn_mix(R->seed ^ 0x3C0FFEE3ULL ^ (g >> 5));
        b = inserted ? ins_base : (uint32_t)(c >> (2 * (g & 31))) & 3;
    }
    if (rev) b ^= 3;
    const uint64_t hr = syn_mix(R->seed ^ 0xBADC0FFEE0DDF00DULL ^ (r * 131072 + i));
    if ((uint32_t)(hr % 10000) < R->sub_per_10k) b = (b + 1 + (uint32_t)((hr >> 40) % 3)) & 3;
    return (uint8_t)"ACGT"[b];
}

#endif
