// Microbenchmark: how many clocks a wave64 vector-ALU instruction occupies its SIMD on MI355X, by instruction kind and by the
// number of wavefronts resident on the SIMD -- the conversion from SQ_INSTS_VALU to "vector-issue time" that bench.py's
// binding floors use.  Every wavefront times a long run of INDEPENDENT instructions with the shader clock (s_memtime);
// clocks per instruction per SIMD = (slowest wavefront's clocks) / (instructions per wavefront x wavefronts per SIMD).
// Build: hipcc --offload-arch=gfx950 -O3 valu_issue.hip -o valu_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// KIND 0: v_fma_f32   1: v_pk_fma_f32   2: v_mul_lo_u32   3: v_mad_u64_u32   4: v_cvt_rpi_i32_f32   5: v_mad_i32_i24
template <int KIND>
__global__ __launch_bounds__(1024) void k_valu(unsigned long long* clocks, float* sink, int iters) {
    float a[8];
    int n[8];
    float2 p[8];
    for (int i = 0; i < 8; ++i) { a[i] = (float)(threadIdx.x + i) * 1e-3f; n[i] = threadIdx.x + i; p[i] = make_float2(a[i], a[i] + 1.f); }
    const float m = 1.0000001f, c = 1e-7f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            else if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(make_float2(m, m)), "v"(make_float2(c, c)));
            else if (KIND == 2) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(n[i]) : "v"(3));
            else if (KIND == 3) { unsigned long long r; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(r) : "v"(n[i]), "v"(18), "v"((unsigned long long)n[(i + 1) & 7]) : "vcc"); n[i] = (int)r; }
            else if (KIND == 4) asm volatile("v_cvt_rpi_i32_f32 %0, %1" : "=v"(n[i]) : "v"(a[i]));
            else asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(n[i]) : "v"(3), "v"(1));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += a[i] + (float)n[i] + p[i].x + p[i].y;
    if (s == 12345.678f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) clocks[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int KIND>
void run(const char* name, int waves_per_simd, unsigned long long* clocks, float* sink) {
    const int iters = 1 << 16, cus = 256;
    // one workgroup of 4 x waves_per_simd wavefronts per CU (the hardware spreads a workgroup's wavefronts over the 4 SIMDs)
    const int threads = 64 * 4 * waves_per_simd;
    if (threads > 1024) return;
    hipLaunchKernelGGL(k_valu<KIND>, dim3(cus), dim3(threads), 0, 0, clocks, sink, iters);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_valu<KIND>, dim3(cus), dim3(threads), 0, 0, clocks, sink, iters);
    CK(hipGetLastError());
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const int nw = cus * 4 * waves_per_simd;
    unsigned long long* h = (unsigned long long*)malloc(nw * sizeof(unsigned long long));
    CK(hipMemcpy(h, clocks, nw * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    unsigned long long mx = 0; double mean = 0;
    for (int i = 0; i < nw; ++i) { mx = h[i] > mx ? h[i] : mx; mean += (double)h[i] / nw; }
    const double instr = (double)iters * 8;
    printf("%-20s %d wavefront(s) per SIMD: %.2f ticks per instruction per wavefront, %.2f ticks of the SIMD per instruction; wall clock (incl. launch) %.3f ns of a SIMD per instruction\n",
           name, waves_per_simd, mean / instr, mean / (instr * waves_per_simd), ms * 1e6 / (instr * waves_per_simd));
    free(h);
}

int main() {
    unsigned long long* clocks; float* sink;
    CK(hipMalloc(&clocks, 1 << 20)); CK(hipMalloc(&sink, 64));
    printf("(s_memtime ticks; compare with the fma row: if the tick is not the shader clock every row scales alike)\n");
    for (int w = 1; w <= 4; w *= 2) {
        run<0>("v_fma_f32", w, clocks, sink);
        run<1>("v_pk_fma_f32", w, clocks, sink);
        run<2>("v_mul_lo_u32", w, clocks, sink);
        run<3>("v_mad_u64_u32", w, clocks, sink);
        run<4>("v_cvt_rpi_i32_f32", w, clocks, sink);
        run<5>("v_mad_i32_i24", w, clocks, sink);
    }
    // wall-clock cross-check of the fma row: total wave-instructions / (SIMDs x seconds) = instructions per SIMD per second
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 1 << 16;
    hipLaunchKernelGGL(k_valu<0>, dim3(256), dim3(1024), 0, 0, clocks, sink, 16);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_valu<0>, dim3(256), dim3(1024), 0, 0, clocks, sink, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double winstr = 256.0 * 16 * iters * 8;
    printf("wall clock: %.3e v_fma_f32 wavefront-instructions in %.3f ms on 1024 SIMDs = %.3f ns of a SIMD per instruction (1.67 = 4 clocks at 2.4 GHz, 0.83 = 2)\n",
           winstr, ms, ms * 1e6 * 1024 / winstr);
    return 0;
}
