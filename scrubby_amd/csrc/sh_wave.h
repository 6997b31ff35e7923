// Wave-wide primitives on the DPP network and the scalar lanes (gfx950: wave64; row_shr / row_mirror inside rows of 16, row_bcast:15 / :31
// across rows, wave_shr:1, v_readlane).  A __shfl / __shfl_xor / __shfl_up is a ds_bpermute - an LDS round trip each, ~100 cycles of
// latency - and a butterfly of six of them in a dependent chain was the longest stretch of several per-anchor and per-seed loops; the same
// reduction on DPP is six VALU moves.  Every function needs the whole wave active (uniform control flow), as the shuffles did.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int CTRL, int ROWS>
__device__ inline int32_t dpp_mov(int32_t old, int32_t v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, ROWS, 0xf, false); }

#define SH_WAVE_SCAN(NAME, T_ID, OP)                                                                   \
    __device__ inline int32_t NAME(int32_t v)                                                          \
    {                                                                                                  \
        int32_t t;                                                                                     \
        t = dpp_mov<0x111, 0xf>(T_ID, v); v = OP(t, v);                                                \
        t = dpp_mov<0x112, 0xf>(T_ID, v); v = OP(t, v);                                                \
        t = dpp_mov<0x114, 0xf>(T_ID, v); v = OP(t, v);                                                \
        t = dpp_mov<0x118, 0xf>(T_ID, v); v = OP(t, v);                                                \
        t = dpp_mov<0x142, 0xa>(T_ID, v); v = OP(t, v);                                                \
        t = dpp_mov<0x143, 0xc>(T_ID, v); v = OP(t, v);                                                \
        return v;                                                                                      \
    }
#define SH_OP_MAX(a, b) ((a) > (b) ? (a) : (b))
#define SH_OP_MIN(a, b) ((a) < (b) ? (a) : (b))
#define SH_OP_ADD(a, b) ((a) + (b))
#define SH_OP_OR(a, b) ((a) | (b))
SH_WAVE_SCAN(wave_scan_max_incl, INT32_MIN, SH_OP_MAX)      // inclusive prefix maximum in lane order
SH_WAVE_SCAN(wave_scan_min_incl, INT32_MAX, SH_OP_MIN)
SH_WAVE_SCAN(wave_scan_add_incl, 0, SH_OP_ADD)
SH_WAVE_SCAN(wave_scan_or_incl, 0, SH_OP_OR)
// lane l receives lane l-1's value, lane 0 receives `fill`
__device__ inline int32_t wave_shr1(int32_t v, int32_t fill) { return dpp_mov<0x138, 0xf>(fill, v); }

// reductions, the result on every lane (lane 63 of the scan holds it)
__device__ inline int32_t wave_all_max(int32_t v) { return __builtin_amdgcn_readlane(wave_scan_max_incl(v), 63); }
__device__ inline int32_t wave_all_min(int32_t v) { return __builtin_amdgcn_readlane(wave_scan_min_incl(v), 63); }
__device__ inline int32_t wave_all_add(int32_t v) { return __builtin_amdgcn_readlane(wave_scan_add_incl(v), 63); }
__device__ inline uint32_t wave_all_or(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane(wave_scan_or_incl((int32_t)v), 63); }
__device__ inline uint32_t wave_all_max_u32(uint32_t v) { return (uint32_t)wave_all_max((int32_t)(v ^ 0x80000000u)) ^ 0x80000000u; }
__device__ inline uint32_t wave_all_min_u32(uint32_t v) { return (uint32_t)wave_all_min((int32_t)(v ^ 0x80000000u)) ^ 0x80000000u; }

// 64-bit values travel as two dwords
template <int CTRL, int ROWS>
__device__ inline unsigned long long dpp_mov_u64(unsigned long long old, unsigned long long v)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)old, (int)(uint32_t)v, CTRL, ROWS, 0xf, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)(old >> 32), (int)(uint32_t)(v >> 32), CTRL, ROWS, 0xf, false);
    return (unsigned long long)hi << 32 | lo;
}
__device__ inline unsigned long long wave_readlane_u64(unsigned long long v, int l)
{
    return (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l) << 32 | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
}
#define SH_WAVE_SCAN64(NAME, T, T_ID, OP)                                                              \
    __device__ inline T NAME(T v0)                                                                     \
    {                                                                                                  \
        unsigned long long v = (unsigned long long)v0, t;                                              \
        t = dpp_mov_u64<0x111, 0xf>((unsigned long long)(T)(T_ID), v); v = (unsigned long long)OP((T)t, (T)v); \
        t = dpp_mov_u64<0x112, 0xf>((unsigned long long)(T)(T_ID), v); v = (unsigned long long)OP((T)t, (T)v); \
        t = dpp_mov_u64<0x114, 0xf>((unsigned long long)(T)(T_ID), v); v = (unsigned long long)OP((T)t, (T)v); \
        t = dpp_mov_u64<0x118, 0xf>((unsigned long long)(T)(T_ID), v); v = (unsigned long long)OP((T)t, (T)v); \
        t = dpp_mov_u64<0x142, 0xa>((unsigned long long)(T)(T_ID), v); v = (unsigned long long)OP((T)t, (T)v); \
        t = dpp_mov_u64<0x143, 0xc>((unsigned long long)(T)(T_ID), v); v = (unsigned long long)OP((T)t, (T)v); \
        return (T)v;                                                                                   \
    }
SH_WAVE_SCAN64(wave_scan_max_incl_i64, long long, INT64_MIN, SH_OP_MAX)
SH_WAVE_SCAN64(wave_scan_max_incl_u64, unsigned long long, 0ull, SH_OP_MAX)
SH_WAVE_SCAN64(wave_scan_min_incl_u64, unsigned long long, ~0ull, SH_OP_MIN)
SH_WAVE_SCAN64(wave_scan_add_incl_u64, unsigned long long, 0ull, SH_OP_ADD)
__device__ inline long long wave_all_max_i64(long long v) { return (long long)wave_readlane_u64((unsigned long long)wave_scan_max_incl_i64(v), 63); }
__device__ inline unsigned long long wave_all_max_u64(unsigned long long v) { return wave_readlane_u64(wave_scan_max_incl_u64(v), 63); }
__device__ inline unsigned long long wave_all_min_u64(unsigned long long v) { return wave_readlane_u64(wave_scan_min_incl_u64(v), 63); }
__device__ inline unsigned long long wave_all_add_u64(unsigned long long v) { return wave_readlane_u64(wave_scan_add_incl_u64(v), 63); }

// lane l's value on every lane; l must be the same on every lane (it is made scalar here)
__device__ inline int32_t wave_bcast(int32_t v, int l) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(l)); }
__device__ inline unsigned long long wave_bcast_u64(unsigned long long v, int l) { return wave_readlane_u64(v, __builtin_amdgcn_readfirstlane(l)); }
// lane l receives lane l-1's 64-bit value, lane 0 receives `fill`
__device__ inline unsigned long long wave_shr1_u64(unsigned long long v, unsigned long long fill) { return dpp_mov_u64<0x138, 0xf>(fill, v); }
