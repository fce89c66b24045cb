// Microbenchmark: cost of one 64-lane 8-byte gather (global_load_dwordx2) as a function of how many
// distinct 128-byte cache lines the wave touches, with the data resident in L1/L2.
// Build: hipcc --offload-arch=gfx950 -O3 gather.hip -o gather
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

struct __attribute__((packed, aligned(4))) fpair { float x, y; };

// each wave owns a private 64 KiB window (L2-resident; a 16 KiB window stays in L1).  LINES distinct
// lines per instruction: lane l reads line (l % LINES) at an offset that keeps lanes on distinct words.
template <int LINES>
__global__ void k_gather(const float* __restrict__ buf, int iters, int window_floats, float* out) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const float* base = buf + (size_t)(wave % 4096) * window_floats;
    float acc = 0.f;
    int off = (lane % LINES) * 32 + ((lane / LINES) * 2) % 30;   // floats; 32 floats = one 128-B line
    for (int it = 0; it < iters; ++it) {
        const fpair a = *reinterpret_cast<const fpair*>(base + off);
        const fpair b = *reinterpret_cast<const fpair*>(base + ((off + LINES * 32) & (window_floats - 1)));
        const fpair c = *reinterpret_cast<const fpair*>(base + ((off + 2 * LINES * 32) & (window_floats - 1)));
        const fpair d = *reinterpret_cast<const fpair*>(base + ((off + 3 * LINES * 32) & (window_floats - 1)));
        acc += a.x + a.y + b.x + b.y + c.x + c.y + d.x + d.y;
        off = (off + 4 * LINES * 32 + 2) & (window_floats - 1);
        off = (off & ~31) | ((off & 31) % 30);
    }
    if (acc == 123.456f) out[0] = acc;
}

template <int LINES>
void run(const float* buf, float* out, int window_floats, const char* tag) {
    const int blocks = 256 * 8, threads = 256, iters = 2000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_gather<LINES>, dim3(blocks), dim3(threads), 0, 0, buf, iters, window_floats, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_gather<LINES>, dim3(blocks), dim3(threads), 0, 0, buf, iters, window_floats, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double instr = (double)blocks * (threads / 64) * iters * 4;   // wave-level gather instructions
    const double clk_per_instr_per_cu = ms * 1e-3 * 2.1e9 * 256 / instr; // at ~2.1 GHz, 256 CUs
    printf("%s lines/instr=%2d : %7.3f ms  -> %6.1f clk per gather instruction per CU, %7.1f GB/s of taps\n", tag, LINES, ms,
           clk_per_instr_per_cu, instr * 64 * 8 / (ms * 1e-3) / 1e9);
}

int main() {
    const size_t N = (size_t)4096 * 16384;  // 4096 windows x 64 KiB
    float* buf; CK(hipMalloc(&buf, N * 4)); CK(hipMemset(buf, 0, N * 4));
    float* out; CK(hipMalloc(&out, 64));
    for (int wf : {4096, 16384}) {
        const char* tag = wf == 4096 ? "window 16 KiB (L1)" : "window 64 KiB (L2)";
        run<1>(buf, out, wf, tag); run<2>(buf, out, wf, tag); run<4>(buf, out, wf, tag); run<8>(buf, out, wf, tag);
        run<16>(buf, out, wf, tag); run<32>(buf, out, wf, tag); run<64>(buf, out, wf, tag);
    }
    return 0;
}
