// Microbenchmark: fp32 atomic-add throughput on MI355X for the access patterns a voxel-gradient
// scatter can produce.  Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics atomics.hip -o atomics
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}

template <int SCOPE>
__device__ __forceinline__ void add(float* p, float v) {
    if (SCOPE == 0) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (SCOPE == 1) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else *p = v;  // plain store, for reference
}

// PAT 0: every lane a random address          PAT 1: a wave's 64 lanes = 64 consecutive floats at a random base
// PAT 2: 16 lanes share one random address     PAT 3: random inside a 256 KiB window that is private to the block
// PAT 4: random inside a 4 MiB window shared by the 8 blocks that (by b % 8) sit on one XCD... approximated by b/8
// PAT 5: 8x8 tile pattern: lane (x,y) -> cell (x/2, y/2), 2x2x2 corner picked by it&7; base walks along a ray
template <int SCOPE, int PAT>
__global__ void k_atomic(float* buf, unsigned nmask, int iters) {
    unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned lane = threadIdx.x & 63, wave = tid >> 6;
    for (int it = 0; it < iters; ++it) {
        unsigned a;
        if (PAT == 0) a = hash32(tid * 977u + it) & nmask;
        else if (PAT == 1) a = ((hash32(wave * 977u + it) & nmask) & ~63u) + lane;
        else if (PAT == 2) a = hash32((tid >> 4) * 977u + it) & nmask;
        else if (PAT == 3) a = ((blockIdx.x * 65536u) + (hash32(tid * 977u + it) & 65535u)) & nmask;
        else if (PAT == 4) a = (((blockIdx.x >> 3) * 1048576u) + (hash32(tid * 977u + it) & 1048575u)) & nmask;
        else {
            unsigned x = lane & 7, y = lane >> 3, c = it & 7, step = it >> 3;
            unsigned base = (hash32(wave) & nmask & ~0xffffffu);
            unsigned cx = (x >> 1) + (c & 1), cy = (y >> 1) + ((c >> 1) & 1), cz = 2 * step + (c >> 2);
            a = (base + cx * 262144u + cy * 512u + cz) & nmask;
        }
        add<SCOPE>(buf + a, 1.0f);
    }
}

// LDS atomics: every lane adds into a 16 KiB per-block brick; PAT 0 random, PAT 5 the tile pattern
template <int PAT>
__global__ void k_lds(float* out, int iters) {
    __shared__ float brick[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) brick[i] = 0.f;
    __syncthreads();
    unsigned tid = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int it = 0; it < iters; ++it) {
        unsigned a;
        if (PAT == 0) a = hash32(tid * 977u + it) & 4095u;
        else {
            unsigned x = lane & 7, y = lane >> 3, c = it & 7, step = (it >> 3) & 7;
            unsigned cx = (x >> 1) + (c & 1), cy = (y >> 1) + ((c >> 1) & 1), cz = 2 * step + (c >> 2);
            a = (w * 1024u + cx * 160u + cy * 32u + cz) & 4095u;
        }
        __hip_atomic_fetch_add(&brick[a], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = brick[tid & 4095u];
}

template <typename K, typename... Args>
double time_kernel(K k, dim3 g, dim3 b, Args... args) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, g, b, 0, 0, args...);  // warm-up
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, g, b, 0, 0, args...);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e-3;
}

int main() {
    const unsigned N = 1u << 27;  // 128 Mi floats = 512 MiB
    float* buf; CK(hipMalloc(&buf, (size_t)N * 4)); CK(hipMemset(buf, 0, (size_t)N * 4));
    float* out; CK(hipMalloc(&out, 1 << 20));
    const int blocks = 256 * 16, threads = 256, iters = 256;
    const double total = (double)blocks * threads * iters;
    const char* pn[] = {"random", "wave-coalesced", "16-lane-dup", "block-window-256K", "xcd-window-4M", "8x8-tile"};
#define RUN(S, P) { double t = time_kernel(k_atomic<S, P>, dim3(blocks), dim3(threads), buf, N - 1, iters); \
    printf("global scope=%-9s pattern=%-18s %8.2f Gop/s\n", S == 0 ? "agent" : (S == 1 ? "workgroup" : "store"), pn[P], total / t * 1e-9); }
    RUN(0, 0) RUN(1, 0) RUN(2, 0)
    RUN(0, 1) RUN(1, 1) RUN(2, 1)
    RUN(0, 2) RUN(1, 2)
    RUN(0, 3) RUN(1, 3)
    RUN(0, 4) RUN(1, 4)
    RUN(0, 5) RUN(1, 5) RUN(2, 5)
    { double t = time_kernel(k_lds<0>, dim3(blocks), dim3(threads), out, iters); printf("LDS ds_add_f32 pattern=random   %8.2f Gop/s\n", total / t * 1e-9); }
    { double t = time_kernel(k_lds<5>, dim3(blocks), dim3(threads), out, iters); printf("LDS ds_add_f32 pattern=8x8-tile %8.2f Gop/s\n", total / t * 1e-9); }
    return 0;
}
