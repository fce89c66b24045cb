// Microbenchmark: LDS accumulate throughput on MI355X (ds_add_f32 / ds_add_u32 / read-modify-write) for the address
// patterns a brick-local scatter of the voxel gradient would produce.  Cheap address generation (one add + and per op)
// so the LDS pipe, not the VALU, is what is timed.
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics lds_atomics.hip -o lds_atomics
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int BRICK = 4096;   // floats per workgroup (16 KiB)

// PAT 0: lane l -> l + 64 * it            (conflict-free, 64 consecutive words)
// PAT 1: pseudo-random word per lane        (bank conflicts as they fall)
// PAT 2: groups of 4 neighbouring lanes share a word, groups consecutive (the duplicate pattern of dense samples)
// PAT 3: all 64 lanes one word
// OP 0: ds_add_f32   1: ds_add_u32   2: ds_read + v_add + ds_write (non-atomic)   3: ds_add_rtn_f32 (value used)
// OP 4: ds_add_u64 (two neighbouring words per lane)   5: ds_add_u32 with every other lane masked off
template <int OP, int PAT>
__global__ __launch_bounds__(256) void k_lds(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float brick[BRICK];
    for (int i = threadIdx.x; i < BRICK; i += blockDim.x) brick[i] = 0.f;
    __syncthreads();
    const unsigned lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned a = PAT == 0 ? lane : PAT == 1 ? (threadIdx.x * 2654435761u >> 7) : PAT == 2 ? (lane >> 2) : 0u;
    const unsigned stride = PAT == 0 ? 64u : PAT == 1 ? (977u + 2u * threadIdx.x) : PAT == 2 ? 16u : 1u;
    a += w * 1024u;
    float keep = 0.f;
#pragma unroll 8
    for (int it = 0; it < iters; ++it) {
        const unsigned idx = a & (BRICK - 1);
        if (OP == 0) __hip_atomic_fetch_add(&brick[idx], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (OP == 1) __hip_atomic_fetch_add((unsigned*)&brick[idx], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (OP == 2) { volatile float* p = &brick[idx]; *p = *p + 1.0f; }
        else if (OP == 3) keep += __hip_atomic_fetch_add(&brick[idx], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (OP == 4) __hip_atomic_fetch_add((unsigned long long*)__builtin_assume_aligned(&brick[idx & ~1u], 8), 0x100000001ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else { if (lane & 1) __hip_atomic_fetch_add((unsigned*)&brick[idx], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }   // half the lanes
        a += stride;
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = brick[blockIdx.x & (BRICK - 1)] + keep;
}

template <typename K>
double time_kernel(K k, dim3 g, dim3 b, float* out, int iters) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, g, b, 0, 0, out, iters);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, g, b, 0, 0, out, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e-3;
}

int main() {
    float* out; CK(hipMalloc(&out, 1 << 20));
    const int blocks = 256 * 32, threads = 256, iters = 4096;
    const double total = (double)blocks * threads * iters;
    const char* on[] = {"ds_add_f32", "ds_add_u32", "read+add+write", "ds_add_rtn_f32", "ds_add_u64", "ds_add_u32 (32 lanes)"};
    const char* pn[] = {"conflict-free", "random", "4-lane-dup", "all-one-word"};
#define RUN(O, P) { double t = time_kernel(k_lds<O, P>, dim3(blocks), dim3(threads), out, iters); \
    printf("LDS %-15s pattern=%-14s %9.1f Gop/s  = %6.2f lane-ops / clk / CU (2.4 GHz, 256 CUs)\n", on[O], pn[P], total / t * 1e-9, total / t / 2.4e9 / 256); }
    RUN(0, 0) RUN(0, 1) RUN(0, 2) RUN(0, 3)
    RUN(1, 0) RUN(1, 1) RUN(1, 2) RUN(1, 3)
    RUN(2, 0) RUN(2, 1)
    RUN(3, 0) RUN(3, 1)
    RUN(4, 0) RUN(4, 1) RUN(4, 2)
    RUN(5, 0) RUN(5, 1) RUN(5, 2)
    return 0;
}
