/* synth.c — CPU twin of the HIP workload generator (test infrastructure, see synth.h). */
#include "synth.h"

void syn_cpu_ref(const syn_ref_params *P, uint64_t g0, uint64_t n, uint8_t *out)
{
    uint64_t i;
    for (i = 0; i < n; ++i) out[i] = (uint8_t)"ACGT"[syn_ref_base(P, g0 + i)];
}

void syn_cpu_reads(const syn_ref_params *P, const syn_read_params *R, uint64_t r0, uint64_t n, uint8_t *out)
{
    uint64_t r;
    uint32_t i;
    for (r = r0; r < r0 + n; ++r) {
        syn_pair pl = syn_place_pair(P, R, r >> 1);
        for (i = 0; i < R->read_len; ++i)
            out[(r - r0) * R->read_len + i] = syn_read_base(P, R, &pl, r >> 1, (uint32_t)(r & 1), i);
    }
}

void syn_cpu_truth(const syn_ref_params *P, const syn_read_params *R, uint64_t r0, uint64_t n, uint8_t *out)
{
    uint64_t r;
    for (r = r0; r < r0 + n; ++r) out[r - r0] = (uint8_t)syn_place_pair(P, R, r >> 1).is_host;
}

void syn_cpu_long_lengths(const syn_read_params *R, uint64_t r0, uint64_t n, uint32_t *out)
{
    uint64_t r;
    for (r = 0; r < n; ++r) out[r] = syn_long_len(R->seed, r0 + r);
}

void syn_cpu_long_reads(const syn_ref_params *P, const syn_read_params *R, uint64_t r0, uint64_t n, const uint64_t *offsets, uint8_t *out)
{
    uint64_t r;
    uint32_t i;
    for (r = 0; r < n; ++r) {
        uint32_t len = (uint32_t)(offsets[r + 1] - offsets[r]);
        for (i = 0; i < len; ++i) out[offsets[r] + i] = syn_long_read_base(P, R, r0 + r, len, i);
    }
}
