/* synth.h — CPU twin of the HIP workload generator (test infrastructure). */
#ifndef ORACLE_SYNTH_H
#define ORACLE_SYNTH_H
#include "../scrubby_amd/csrc/sh_synth_core.h"
#ifdef __cplusplus
extern "C" {
#endif
/* ASCII reference bases [g0, g0+n) */
void syn_cpu_ref(const syn_ref_params *P, uint64_t g0, uint64_t n, uint8_t *out);
/* records [r0, r0+n): record r = mate (r&1) of pair (r>>1); out is n*read_len ASCII bytes */
void syn_cpu_reads(const syn_ref_params *P, const syn_read_params *R, uint64_t r0, uint64_t n, uint8_t *out);
/* truth label per record: 1 = drawn from the reference */
void syn_cpu_truth(const syn_ref_params *P, const syn_read_params *R, uint64_t r0, uint64_t n, uint8_t *out);
/* long reads: lengths[n] and, given offsets[n+1], the bases */
void syn_cpu_long_lengths(const syn_read_params *R, uint64_t r0, uint64_t n, uint32_t *out);
void syn_cpu_long_reads(const syn_ref_params *P, const syn_read_params *R, uint64_t r0, uint64_t n, const uint64_t *offsets, uint8_t *out);
#ifdef __cplusplus
}
#endif
#endif
