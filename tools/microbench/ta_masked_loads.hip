// What does a 64-lane 4-byte gather cost the CU's texture-address / L1 path when part of its lanes have nothing to load?
// The Siddon slab march (k_siddon_slab) issues three buffer loads per slab of which 1.83 lanes-worth are live on average; lanes
// without a segment hand the buffer resource an out-of-range offset (returns 0, no memory request).  Variants, all on an
// L1-resident working set (every lane its own 128-byte line of an 8 KB window, so that only the address path is measured):
//   full     every lane loads
//   oob      LIVE % of the lanes load, the others pass offset -1 to the buffer resource (what the march does)
//   exec     LIVE % of the lanes load, the others are switched off in EXEC (a branch around the load)
//   lines    every lane loads, but the 64 lanes share LINES distinct 128-byte lines
// Build: hipcc --offload-arch=gfx950 -O3 ta_masked_loads.hip -o ta_masked_loads
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int MODE>   // 0 full / lines, 1 oob, 2 exec, 3 full with the lanes' dwords scattered inside their lines
__global__ __launch_bounds__(256) void k(const float* buf, unsigned bytes, int iters, int live_of_8, int line_shift, float* out) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(buf), (short)0, bytes, 0x00020000);
    const int lane = threadIdx.x & 63;
    const bool live = (lane & 7) < live_of_8;
    int off = ((lane >> line_shift) << 7) + ((lane & 3) << 2) + (blockIdx.x & 1) * 8192;   // one 128-byte line per lane (or per 2^line_shift lanes)
    if (MODE == 3) off = ((lane >> line_shift) << 7) + (((lane * 13) & 31) << 2) + (blockIdx.x & 1) * 8192;   // any of the line's 32 dwords
    if (MODE == 4) off = ((lane >> line_shift) << 7) + (((lane * 13) & 7) << 4) + (blockIdx.x & 1) * 8192;    // any of the line's eight 16-byte pieces
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int o = MODE >= 3 ? ((off & ~127) + ((off + u * 20) & 127)) & 16383 : (off + u * 16) & 16383;
            if (MODE == 1) acc += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, live ? o : -1, 0, 0));
            else if (MODE == 2) { if (live) acc += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, o, 0, 0)); }
            else acc += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, o, 0, 0));
        }
        off ^= 64;
    }
    if (acc == 123.456f) out[0] = acc;
}

template <int MODE>
void run(const float* buf, float* out, int live_of_8, int line_shift, const char* tag) {
    const int blocks = 256 * 8, iters = 2000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, buf, 16384u * 2, 10, live_of_8, line_shift, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, buf, 16384u * 2, iters, live_of_8, line_shift, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double loads_per_cu = (double)blocks * 4 * iters * 8 / 256.0;   // wavefront load instructions per CU
    printf("%-34s %8.3f ms  %6.1f ns per wavefront load and CU  (%5.1f clocks at 2.4 GHz)\n", tag, ms, ms * 1e6 / loads_per_cu, ms * 1e6 / loads_per_cu * 2.4);
}

int main() {
    float* buf; CK(hipMalloc(&buf, 1 << 20)); CK(hipMemset(buf, 0, 1 << 20));
    float* out; CK(hipMalloc(&out, 64));
    run<0>(buf, out, 8, 0, "full, 64 lines");
    run<0>(buf, out, 8, 1, "full, 32 lines");
    run<0>(buf, out, 8, 2, "full, 16 lines");
    run<0>(buf, out, 8, 3, "full, 8 lines");
    run<0>(buf, out, 8, 4, "full, 4 lines");
    run<3>(buf, out, 8, 3, "full, 8 lines, scattered dwords");
    run<3>(buf, out, 8, 2, "full, 16 lines, scattered dwords");
    run<4>(buf, out, 8, 3, "full, 8 lines, scattered 16-B pieces");
    run<1>(buf, out, 5, 3, "oob, 5 of 8 lanes live, 8 lines");
    run<2>(buf, out, 5, 3, "exec, 5 of 8 lanes live, 8 lines");
    run<1>(buf, out, 2, 3, "oob, 2 of 8 lanes live, 8 lines");
    run<2>(buf, out, 2, 3, "exec, 2 of 8 lanes live, 8 lines");
    run<1>(buf, out, 0, 3, "oob, no lane live");
    run<1>(buf, out, 5, 0, "oob, 5 of 8 lanes live, 40 lines");
    run<2>(buf, out, 5, 0, "exec, 5 of 8 lanes live, 40 lines");
    return 0;
}
