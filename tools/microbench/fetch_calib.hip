// Calibration of rocprofv3's FETCH_SIZE / TCC_EA0_RDREQ* on gfx950 for 16-byte-per-lane GATHERS (the trilinear forward's
// access: global_load_dwordx4, a handful of 128-byte lines per wavefront instruction), with a known line count.
//
// Every variant issues the same instruction -- 64 lanes x 16 B -- and touches every 128-byte line of its working set exactly
// once per pass, in a pseudo-random order (line = group * ODD mod NLINES, a bijection for a power-of-two NLINES):
//   G = 8  eight consecutive lanes read the eight 16-byte pieces of one line      (8 lines per instruction, lines fully used)
//   G = 4  four lanes read one aligned 64-byte half, the other half is never read (16 lines per instruction, half used)
//   G = 2  two lanes read one aligned 32-byte sector                              (32 lines per instruction)
//   G = 1  one lane reads 16 bytes of its own line                                 (64 lines per instruction)
//   seq    G = 8 with the identity order: the plain coalesced stream (the guide's known case: FETCH_SIZE = bytes / 2)
// Working sets: 2 GiB (far beyond the 256 MiB Infinity Cache: every line comes from HBM) and 64 MiB swept 32 times (beyond the
// 32 MiB of L2, inside the Infinity Cache: do MALL hits count as fetches, and how fast are they?).
// What the memory side MOVED per line follows from the time at the bandwidth limit; what the counters SAY from the PMC passes
// (tools/exp_fetch_calibration.sh): bytes-per-line as counted = FETCH_SIZE / lines.
// Build: hipcc --offload-arch=gfx950 -O3 fetch_calib.hip -o fetch_calib
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int G, bool SEQ>
__global__ __launch_bounds__(256) void k_fetch(const float4* __restrict__ buf, unsigned nlines_mask, int iters, unsigned mult, float* out) {
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned nthreads = gridDim.x * blockDim.x;
    const unsigned sub = tid % G;                 // which 16-byte piece of the group's segment
    float acc = 0.f;
#pragma unroll 4
    for (int it = 0; it < iters; ++it) {
        const unsigned group = (unsigned)it * (nthreads / G) + tid / G;
        const unsigned line = SEQ ? (group & nlines_mask) : ((group * mult) & nlines_mask);
        const float4 v = buf[(size_t)line * 8 + sub];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123.456f) out[0] = acc;
}

template <int G, bool SEQ>
void run(const float4* buf, float* out, size_t ws_bytes, int passes, const char* tag) {
    const unsigned nlines = (unsigned)(ws_bytes / 128);
    const int blocks = 256 * 8, threads = 256;
    const unsigned nthreads = blocks * threads;
    const long long groups = (long long)nlines * passes;               // every line once per pass
    const int iters = (int)(groups / (nthreads / G));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_fetch<G, SEQ>), dim3(blocks), dim3(threads), 0, 0, buf, nlines - 1, iters, 2654435761u | 1u, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_fetch<G, SEQ>), dim3(blocks), dim3(threads), 0, 0, buf, nlines - 1, iters, 2654435761u | 1u, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double lines = (double)iters * (nthreads / G);
    printf("%-22s G=%d%s : %8.3f ms  %10.0f lines  %7.2f Glines/s  = %6.2f TB/s at 128 B/line, %6.2f at 64, %6.2f at 32; used %6.2f TB/s\n", tag, G,
           SEQ ? " seq" : "    ", ms, lines, lines / (ms * 1e-3) / 1e9, lines * 128 / (ms * 1e-3) / 1e12, lines * 64 / (ms * 1e-3) / 1e12,
           lines * 32 / (ms * 1e-3) / 1e12, lines * G * 16 / (ms * 1e-3) / 1e12);
}

int main() {
    const size_t N = (size_t)2 << 30;
    float4* buf; CK(hipMalloc(&buf, N));
    // random (non-zero) contents: DVFS clocks zero-filled inputs higher
    { float* h = (float*)malloc(64 << 20); for (size_t i = 0; i < (64 << 20) / 4; ++i) h[i] = (float)rand() / (float)RAND_MAX;
      for (size_t o = 0; o < N; o += (64 << 20)) CK(hipMemcpy((char*)buf + o, h, 64 << 20, hipMemcpyHostToDevice)); free(h); }
    float* out; CK(hipMalloc(&out, 64));
    const size_t big = (size_t)2 << 30, mall = (size_t)64 << 20;   // (powers of two: the line permutation is a multiply mod 2^k)
    run<8, true>(buf, out, big, 1, "2 GiB once");
    run<8, false>(buf, out, big, 1, "2 GiB once");
    run<4, false>(buf, out, big, 1, "2 GiB once");
    run<2, false>(buf, out, big, 1, "2 GiB once");
    run<1, false>(buf, out, big, 1, "2 GiB once");
    run<8, true>(buf, out, mall, 32, "64 MiB x 32");
    run<8, false>(buf, out, mall, 32, "64 MiB x 32");
    run<4, false>(buf, out, mall, 32, "64 MiB x 32");
    run<1, false>(buf, out, mall, 32, "64 MiB x 32");
    return 0;
}
