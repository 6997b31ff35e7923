// False sharing across XCDs: blocks b and b + n_pairs (other XCD) write the even / odd 16-byte quarters of the same lines with plain stores,
// fence, then block b checks every word (agent-scope loads after an acquire).  Lost updates = the L2 writes back whole lines.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ inline uint32_t cc_u32(const void *p) { return __hip_atomic_load((const uint32_t *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__global__ void k(uint32_t *X, uint32_t *flags, uint32_t *bad, int N, int rounds, int n_pairs, int gran)
{
    const int pair = blockIdx.x % n_pairs, role = blockIdx.x / n_pairs;
    uint32_t *x = X + (size_t)pair * N;
    uint32_t *f = flags + pair * 64;
    for (int r = 1; r <= rounds; ++r) {
        // both read the whole range first (lines valid in both L2s)
        uint32_t s = 0;
        for (int i = threadIdx.x; i < N; i += blockDim.x) s += x[i];
        __syncthreads();
        if (threadIdx.x == 0) { __threadfence(); atomicAdd(f, 1u); int looks = 0; while (cc_u32(f) < (uint32_t)(2 * (2 * r - 1)) && ++looks < 100000000) __builtin_amdgcn_s_sleep(2); }
        __syncthreads();
        for (int i = threadIdx.x; i < N; i += blockDim.x) if (((i / gran) & 1) == role) x[i] = (uint32_t)(r * 1000003 + i);
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) { __threadfence(); atomicAdd(f, 1u); int looks = 0; while (cc_u32(f) < (uint32_t)(2 * (2 * r)) && ++looks < 100000000) __builtin_amdgcn_s_sleep(2); }
        __syncthreads();
        __threadfence();
        if (role == 0) {
            uint32_t b1 = 0, b2 = 0;
            for (int i = threadIdx.x; i < N; i += blockDim.x) { b1 += cc_u32(x + i) != (uint32_t)(r * 1000003 + i); b2 += x[i] != (uint32_t)(r * 1000003 + i); }
            if (b1) atomicAdd(bad, b1);
            if (b2) atomicAdd(bad + 1, b2);
        }
        if (s == 0xffffffffu) atomicAdd(bad + 2, 1u);
        __syncthreads();
        if (threadIdx.x == 0) { __threadfence(); atomicAdd(f + 16, 1u); int looks = 0; while (cc_u32(f + 16) < (uint32_t)(2 * r) && ++looks < 100000000) __builtin_amdgcn_s_sleep(2); }
        __syncthreads();
    }
}
int main()
{
    const int N = 1 << 14, n_pairs = 60, rounds = 20;
    uint32_t *X, *flags, *bad;
    (void)hipMalloc(&X, (size_t)n_pairs * N * 4); (void)hipMalloc(&flags, n_pairs * 64 * 4); (void)hipMalloc(&bad, 16);
    for (int gran : {1, 4, 16, 32}) {
        (void)hipMemset(X, 0, (size_t)n_pairs * N * 4); (void)hipMemset(flags, 0, n_pairs * 64 * 4); (void)hipMemset(bad, 0, 16);
        hipLaunchKernelGGL(k, dim3(2 * n_pairs), dim3(64), 0, 0, X, flags, bad, N, rounds, n_pairs, gran);
        hipError_t e = hipDeviceSynchronize();
        uint32_t h[4]; (void)hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost);
        printf("granularity %d words: err %d; of %lld words: wrong by agent-scope loads %u, by plain loads %u\n", gran, (int)e, (long long)N * n_pairs * rounds, h[0], h[1]);
    }
    return 0;
}
