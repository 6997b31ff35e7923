// Are plain loads coherent across XCDs after __threadfence() when the reader has the OLD contents of the lines in its L2?
// Block A (reader) reads X[0..N) (caches it), raises flag1; block B (writer, another XCD with luck: many block pairs) overwrites X, fences, raises flag2;
// A sees flag2 (agent-scope atomic), fences, reads X with plain loads and counts stale words.  Also the variant with agent-scope atomic loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ inline uint32_t cc_u32(const void *p) { return __hip_atomic_load((const uint32_t *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__global__ void k(uint32_t *X, uint32_t *flags, uint32_t *stale, int N, int rounds, int n_pairs)
{
    const int pair = blockIdx.x % n_pairs, role = blockIdx.x / n_pairs;      // role 0 reader, 1 writer: blocks b and b + n_pairs
    uint32_t *x = X + (size_t)pair * N;
    uint32_t *f1 = flags + pair * 64, *f2 = flags + pair * 64 + 32;
    for (int r = 1; r <= rounds; ++r) {
        if (role == 0) {
            uint32_t s = 0;
            for (int i = threadIdx.x; i < N; i += blockDim.x) s += x[i];      // cache the old contents
            __syncthreads();
            if (threadIdx.x == 0) { __threadfence(); atomicExch(f1, (uint32_t)r); }
            if (threadIdx.x == 0) { int looks = 0; while (cc_u32(f2) != (uint32_t)r && ++looks < 100000000) __builtin_amdgcn_s_sleep(2); }
            __syncthreads();
            __threadfence();
            uint32_t bad_plain = 0, bad_cc = 0;
            for (int i = threadIdx.x; i < N; i += blockDim.x) { bad_plain += x[i] != (uint32_t)(r * 1000003 + i); bad_cc += cc_u32(x + i) != (uint32_t)(r * 1000003 + i); }
            if (bad_plain) atomicAdd(stale + 0, bad_plain);
            if (bad_cc) atomicAdd(stale + 1, bad_cc);
            if (s == 0xffffffffu) atomicAdd(stale + 2, 1u);
            __syncthreads();
        } else {
            if (threadIdx.x == 0) { int looks = 0; while (cc_u32(f1) != (uint32_t)r && ++looks < 100000000) __builtin_amdgcn_s_sleep(2); }
            __syncthreads();
            __threadfence();
            for (int i = threadIdx.x; i < N; i += blockDim.x) x[i] = (uint32_t)(r * 1000003 + i);
            __threadfence();
            __syncthreads();
            if (threadIdx.x == 0) { __threadfence(); atomicExch(f2, (uint32_t)r); }
        }
    }
}
int main()
{
    const int N = 1 << 16, n_pairs = 60, rounds = 20;
    uint32_t *X, *flags, *stale;
    hipMalloc(&X, (size_t)n_pairs * N * 4); hipMalloc(&flags, n_pairs * 64 * 4); hipMalloc(&stale, 16);
    hipMemset(X, 0, (size_t)n_pairs * N * 4); hipMemset(flags, 0, n_pairs * 64 * 4); hipMemset(stale, 0, 16);
    hipLaunchKernelGGL(k, dim3(2 * n_pairs), dim3(64), 0, 0, X, flags, stale, N, rounds, n_pairs);
    hipError_t e = hipDeviceSynchronize();
    uint32_t h[4]; hipMemcpy(h, stale, 16, hipMemcpyDeviceToHost);
    printf("err %d; words checked %lld; stale with plain loads %u, with agent-scope atomic loads %u\n", (int)e, (long long)N * n_pairs * rounds, h[0], h[1]);
    return 0;
}
