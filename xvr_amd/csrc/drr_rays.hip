// MI355X (gfx950) differentiable-DRR kernels: the per-ray side -- pose gradient from the saved jacobian, ray generation
// from the camera vector and its adjoint, jacobian -> camera in one fixed-order pass.
#include "drr_common.hiph"
#include "j2c_device.hiph"

namespace {

// =============================================================================================
// pose-side backward from the saved jacobian (C == 1): elementwise + wave reduction
// =============================================================================================
template <int RPT>   // rays per thread, as in k_jac_to_cam
__global__ __launch_bounds__(WG) void k_backward_from_jac(const float* __restrict__ jac, const float* __restrict__ gout,
                                                         int n, float* gsrc, float* __restrict__ gtgt,
                                                         float* __restrict__ glen) {
    __shared__ float part[3][WG / 64];
    const int b = blockIdx.y;
    float js[3] = {0.f, 0.f, 0.f};
    float g_[RPT];
    float4 j0_[RPT], j1_[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int r = (blockIdx.x * RPT + k) * WG + threadIdx.x;
        g_[k] = 0.f;
        j0_[k] = j1_[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < n) {
            const size_t ray = (size_t)b * n + r;
            g_[k] = gout[ray];
            const float4* jp = reinterpret_cast<const float4*>(jac + ray * XVR_DRR_JAC_STRIDE);
            j0_[k] = jp[0];
            j1_[k] = jp[1];
        }
    }
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int r = (blockIdx.x * RPT + k) * WG + threadIdx.x;
        if (r < n) {
            const size_t ray = (size_t)b * n + r;
            const float g = g_[k];
            const float4 j0 = j0_[k], j1 = j1_[k];
            js[0] += g * j0.y; js[1] += g * j0.z; js[2] += g * j0.w;
            float* tp = gtgt + ray * 3;
            tp[0] = g * j1.x; tp[1] = g * j1.y; tp[2] = g * j1.z;
            if (glen) glen[ray] = g * j0.x;
        }
    }
    // grad_source is shared by all rays of the pose: wave butterfly (DPP/shfl), then the 4 waves of
    // the block through LDS, then ONE atomic per component per block
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float tot = wave_sum_f(js[i]);
        if ((threadIdx.x & 63) == 0) part[i][threadIdx.x >> 6] = tot;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const float tot = part[threadIdx.x][0] + part[threadIdx.x][1] + part[threadIdx.x][2] + part[threadIdx.x][3];
        if (tot != 0.f) atomic_add_f32(gsrc + 3 * b + threadIdx.x, tot);
    }
}

// =============================================================================================
// Ray generation fused into one pass (rows a2-a4 of SURVEY.md 8a): what xvr does with three torch
// calls and ~40 launches -- drr.detector(pose, None), (target - source).norm(), affinv(source/target)
// (src/xvr/model/trainer.py:283-285) -- is an affine map of the pixel index per pose:
//     target_vox(i, j) = Mv (i, j, 1)^T,   raylen(i, j) = | Mw (i, j, 1)^T - s_w |
// cam[b] = { Mv[3][3], s_v[3], Mw[3][3], s_w[3] } (24 floats, built by the host from the 4x4 pose).
// =============================================================================================
__global__ __launch_bounds__(WG) void k_rays_fwd(const float* __restrict__ cam, int H, int W, float* __restrict__ source,
                                                 float* __restrict__ target, float* __restrict__ raylen) {
    const int b = blockIdx.y, n = H * W;
    const int r = blockIdx.x * WG + threadIdx.x;
    const float* c = cam + 24 * b;
    if (r == 0) { source[3 * b] = c[9]; source[3 * b + 1] = c[10]; source[3 * b + 2] = c[11]; }
    if (r >= n) return;
    const int i = r / W, j = r - i * W;
    const float fi = (float)i, fj = (float)j;
    float* t = target + ((size_t)b * n + r) * 3;
    float l2 = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        t[a] = fmaf(c[3 * a], fi, fmaf(c[3 * a + 1], fj, c[3 * a + 2]));
        const float w = fmaf(c[12 + 3 * a], fi, fmaf(c[12 + 3 * a + 1], fj, c[12 + 3 * a + 2])) - c[21 + a];
        l2 = fmaf(w, w, l2);
    }
    raylen[(size_t)b * n + r] = sqrtf(l2);
}

// backward: 21 sums over a pose's rays (wave butterfly -> LDS across the 4 waves -> one atomic each per block)
template <int RPT>   // rays per thread, as in k_jac_to_cam: 1 for registration-sized launches, 4 for batches
__global__ __launch_bounds__(WG) void k_rays_bwd(const float* __restrict__ cam, int H, int W, const float* __restrict__ g_source,
                                                 const float* __restrict__ g_target, const float* __restrict__ g_raylen,
                                                 float* g_cam) {
    __shared__ float part[21][WG / 64];
    const int b = blockIdx.y, n = H * W;
    const float* c = cam + 24 * b;
    float acc[21];
#pragma unroll
    for (int q = 0; q < 21; ++q) acc[q] = 0.f;
    float gt_[RPT][3], gl_[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {   // every load of the thread first
        const int r = (blockIdx.x * RPT + k) * WG + threadIdx.x;
        gt_[k][0] = gt_[k][1] = gt_[k][2] = gl_[k] = 0.f;
        if (r < n) {
            const float* gt = g_target + ((size_t)b * n + r) * 3;
            gt_[k][0] = gt[0]; gt_[k][1] = gt[1]; gt_[k][2] = gt[2];
            gl_[k] = g_raylen ? g_raylen[(size_t)b * n + r] : 0.f;
        }
    }
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int r = (blockIdx.x * RPT + k) * WG + threadIdx.x;
        if (r < n) {
            const int i = r / W, j = r - i * W;
            const float pix[3] = {(float)i, (float)j, 1.f};
            float w[3], l2 = 0.f;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                w[a] = fmaf(c[12 + 3 * a], pix[0], fmaf(c[12 + 3 * a + 1], pix[1], c[12 + 3 * a + 2])) - c[21 + a];
                l2 = fmaf(w[a], w[a], l2);
            }
            const float s = l2 > 0.f ? gl_[k] / sqrtf(l2) : 0.f;  // g_L * (unit direction) = s * w
#pragma unroll
            for (int a = 0; a < 3; ++a) {
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    acc[3 * a + m] += gt_[k][a] * pix[m];       // d/d Mv[a][m]
                    acc[9 + 3 * a + m] += s * w[a] * pix[m];    // d/d Mw[a][m]
                }
                acc[18 + a] += -s * w[a];                        // d/d s_w[a]
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 21; ++q) {
        const float tot = wave_sum_f(acc[q]);
        if ((threadIdx.x & 63) == 0) part[q][threadIdx.x >> 6] = tot;
    }
    __syncthreads();
    if (threadIdx.x < 21) {
        const int q = threadIdx.x;
        const float tot = part[q][0] + part[q][1] + part[q][2] + part[q][3];
        // layout of g_cam mirrors cam: Mv 0..8, s_v 9..11, Mw 12..20, s_w 21..23
        const int dst = q < 9 ? q : (q < 18 ? q + 3 : q + 3);
        if (tot != 0.f) atomic_add_f32(g_cam + 24 * b + dst, tot);
    }
    if (blockIdx.x == 0 && threadIdx.x < 3 && g_source) atomic_add_f32(g_cam + 24 * b + 9 + threadIdx.x, g_source[3 * b + threadIdx.x]);
}

// ---------------------------------------------------------------------------------------------
// jacobian -> camera: k_backward_from_jac and k_rays_bwd in one pass, for callers that want d/d cam and
// not the per-ray gradients (the registration loop).  grad_target / grad_raylen are never written, and
// the 24 sums are ORDER-DETERMINISTIC: every block stores its partial sums, the block that finishes last
// for a pose (ticket counter) adds them in block order and WRITES grad_cam[b] -- no float atomics, the
// same bits on every run.
// ---------------------------------------------------------------------------------------------
// RPT = rays per thread (ray base + t + k WG of the block's WG * RPT): 1 for registration-sized launches (latency: as many
// blocks as possible); 4 for batches -- a quarter of the 24 wavefront reductions, stores and tickets per ray, all 12 loads of
// a thread in flight at once (C2, 116 poses: 0.32 -> see profiles/r03_bench_final_pose_only.json)
template <int RPT>
__global__ __launch_bounds__(WG) void k_jac_to_cam(const float* __restrict__ jac, const float* __restrict__ gout,
                                                   const float* __restrict__ cam, int H, int W, float* partial,
                                                   unsigned* counter, float* __restrict__ g_cam) {
    const int b = blockIdx.y, n = H * W, nblk = gridDim.x;
    const float* c = cam + 24 * b;
    float acc[24];
#pragma unroll
    for (int q = 0; q < 24; ++q) acc[q] = 0.f;
    float g_[RPT];
    float4 j0_[RPT], j1_[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {   // every load of the thread first
        const int r = (blockIdx.x * RPT + k) * WG + threadIdx.x;
        g_[k] = 0.f;
        j0_[k] = j1_[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < n) {
            const size_t ray = (size_t)b * n + r;
            g_[k] = gout[ray];
            const float4* jp = reinterpret_cast<const float4*>(jac + ray * XVR_DRR_JAC_STRIDE);
            j0_[k] = jp[0];
            j1_[k] = jp[1];
        }
    }
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int r = (blockIdx.x * RPT + k) * WG + threadIdx.x;
        if (r < n) j2c_accumulate(g_[k], j0_[k], j1_[k], c, r, W, acc);
    }
    j2c_reduce(acc, partial, counter, b, nblk, g_cam + 24 * b);
}

// =============================================================================================
// clip_to_volume == 2: the alpha window of a whole call (include/xvr_drr.h: xvr_drr_alpha_window).
// A = the smallest alpha_min and Z = the largest alpha_max over the rays that meet the volume, by 64-bit atomic max on
// (float bits << 32 | ray index) keys -- the value and WHICH ray holds it in one word (ties: the lowest ray index), because
// the backward routes the window's gradient to exactly those two rays (torch's min() / max()).
// =============================================================================================
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)v, o), hi = __shfl_xor((unsigned)(v >> 32), o);
        const unsigned long long w = ((unsigned long long)hi << 32) | lo;
        v = w > v ? w : v;
    }
    return v;
}

__global__ __launch_bounds__(WG) void k_alpha_window_reduce(const float* __restrict__ source, const float* __restrict__ target, int n,
                                                           xvr_drr_spec sp, unsigned long long* keys) {
    const int b = blockIdx.y;
    const int r = blockIdx.x * WG + threadIdx.x;
    unsigned long long kmin_inv = 0ull, kmax = 0ull;   // 0 = no ray
    if (r < n) {
        const float* tp = target + ((size_t)b * n + r) * 3;
        float s[3], d[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) { s[i] = source[3 * b + i]; d[i] = (tp[i] - s[i]) + sp.eps; }
        float amin, amax;
        int ain, aout;
        alpha_range(sp, s, d, amin, amax, ain, aout);
        if (amax > amin) {   // (both in [0, 1]: their bit patterns order like the values)
            const unsigned idx = (unsigned)((size_t)b * n + r);
            kmin_inv = ~(((unsigned long long)__float_as_uint(amin) << 32) | idx);
            kmax = ((unsigned long long)__float_as_uint(amax) << 32) | (0xffffffffu - idx);
        }
    }
    // wavefront -> workgroup (LDS) -> ONE atomic per workgroup on the POSE's pair of slots: a few hundred same-address
    // atomics per slot (all 118 k wavefronts of a batch on two words took 2.7 ms)
    __shared__ unsigned long long part[2][WG / 64];
    kmin_inv = wave_max_u64(kmin_inv);
    kmax = wave_max_u64(kmax);
    if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = kmin_inv; part[1][threadIdx.x >> 6] = kmax; }
    __syncthreads();
    if (threadIdx.x < 2) {
        unsigned long long v = part[threadIdx.x][0];
        for (int w = 1; w < WG / 64; ++w) v = part[threadIdx.x][w] > v ? part[threadIdx.x][w] : v;
        if (v) atomicMax(keys + 2 * b + threadIdx.x, v);
    }
}

__global__ __launch_bounds__(64) void k_alpha_window_final(float* window, int B, float near_, float far_, float inv_denom) {
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(window + WIN_KEYS);
    const unsigned long long* slots = reinterpret_cast<const unsigned long long*>(window + XVR_DRR_ALPHA_WINDOW_FLOATS);
    unsigned long long k0 = 0ull, k1 = 0ull;
    for (int b = threadIdx.x; b < B; b += 64) {
        k0 = slots[2 * b] > k0 ? slots[2 * b] : k0;
        k1 = slots[2 * b + 1] > k1 ? slots[2 * b + 1] : k1;
    }
    k0 = wave_max_u64(k0);
    k1 = wave_max_u64(k1);
    if (threadIdx.x != 0) return;
    keys[0] = k0; keys[1] = k1;
    float A = 0.f, Z = 0.f;
    if (keys[0] && keys[1]) {
        A = __uint_as_float((unsigned)((~keys[0]) >> 32));
        Z = __uint_as_float((unsigned)(keys[1] >> 32));
    }
    const float W = Z - A;
    window[WIN_A] = A; window[WIN_Z] = Z;
    window[WIN_NEAR] = fmaf(near_, W, A); window[WIN_FAR] = fmaf(far_, W, A);
    window[WIN_INV] = inv_denom * W; window[WIN_W] = W;
    window[WIN_GA] = 0.f; window[WIN_GZ] = 0.f;
}

// d loss / d A and d loss / d Z from the saved jacobian: with alpha_k = A + u_k W (W = Z - A) and the image scaled by W,
//   d out / d A |_W = sum_k d out / d alpha_k = sum_i d_i (js_i + jt_i)
//   d out / d W |_A = out / W + sum_k u_k d out / d alpha_k = (out + sum_k (alpha_k - A) d out / d alpha_k) / W   [jacobian row, float 7]
// (js, jt: the per-ray jacobian rows, which already hold scale * a (G - H) and scale * a H), and d/dA|_Z = d/dA|_W - d/dW, d/dZ = d/dW.
__global__ __launch_bounds__(WG) void k_alpha_window_bwd_reduce(const float* __restrict__ jac, const float* __restrict__ gout,
                                                               const float* __restrict__ source, const float* __restrict__ target,
                                                               const float* __restrict__ raylen, int n, float eps, float* window) {
    __shared__ float part[2][WG / 64];
    const int b = blockIdx.y;
    const int r = blockIdx.x * WG + threadIdx.x;
    const float A = window[WIN_A], W = window[WIN_W];
    float ga = 0.f, gz = 0.f;
    if (r < n && W > 0.f) {
        const size_t ray = (size_t)b * n + r;
        const float g = gout[ray];
        const float4* jp = reinterpret_cast<const float4*>(jac + ray * XVR_DRR_JAC_STRIDE);
        const float4 j0 = jp[0], j1 = jp[1];
        const float* tp = target + ray * 3;
        const float js[3] = {j0.y, j0.z, j0.w}, jt[3] = {j1.x, j1.y, j1.z};
        // (sum_k (alpha_k - A) d out / d alpha_k comes ready-made in the jacobian's spare float: rebuilt as sum_i d_i (jt_i -
        //  A (js_i + jt_i)) it is a difference of two large sums, 2 % off in float32)
        float s1 = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float d = (tp[i] - source[3 * b + i]) + eps;
            s1 = fmaf(d, js[i] + jt[i], s1);
        }
        const float dW = fmaf(j0.x, raylen[ray], j1.w) / W;
        ga = g * (s1 - dW);
        gz = g * dW;
    }
    ga = wave_sum_f(ga);
    gz = wave_sum_f(gz);
    if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = ga; part[1][threadIdx.x >> 6] = gz; }
    __syncthreads();
    if (threadIdx.x < 2) {   // (per-pose slots behind the window's header: see k_alpha_window_reduce)
        const float tot = part[threadIdx.x][0] + part[threadIdx.x][1] + part[threadIdx.x][2] + part[threadIdx.x][3];
        if (tot != 0.f) atomic_add_f32(window + XVR_DRR_ALPHA_WINDOW_FLOATS + 4 * b + threadIdx.x, tot);
    }
}

// ... and on to the two extremal rays: A = (plane - s_i) / d_i on the axis the ray enters through, Z likewise on its exit
// axis (no gradient where the value is the clamp 0 / 1)
__global__ void k_alpha_window_bwd_apply(const float* __restrict__ source, const float* __restrict__ target, int B, int n, xvr_drr_spec sp,
                                         float* window, float* gsrc, float* gtgt) {
    const unsigned long long* keys = reinterpret_cast<const unsigned long long*>(window + WIN_KEYS);
    if (!(keys[0] && keys[1])) return;
    const unsigned idx[2] = {(unsigned)(~keys[0]), 0xffffffffu - (unsigned)keys[1]};
    float gr[2] = {0.f, 0.f};
    for (int b = 0; b < B; ++b) {   // the poses' partial sums, in pose order
        gr[0] += window[XVR_DRR_ALPHA_WINDOW_FLOATS + 4 * b];
        gr[1] += window[XVR_DRR_ALPHA_WINDOW_FLOATS + 4 * b + 1];
    }
    window[WIN_GA] = gr[0]; window[WIN_GZ] = gr[1];
    for (int e = 0; e < 2; ++e) {
        const int b = (int)(idx[e] / (unsigned)n);
        const size_t ray = idx[e];
        float s[3], d[3];
        for (int i = 0; i < 3; ++i) { s[i] = source[3 * b + i]; d[i] = (target[ray * 3 + i] - s[i]) + sp.eps; }
        float amin, amax;
        int ain, aout;
        alpha_range(sp, s, d, amin, amax, ain, aout);
        const int ax = e == 0 ? ain : aout;
        const float al = e == 0 ? amin : amax;
        if (ax >= 0 && gr[e] != 0.f) {
            const float dd = ax == 0 ? d[0] : (ax == 1 ? d[1] : d[2]);
            gsrc[3 * b + ax] += gr[e] * (al - 1.f) / dd;
            gtgt[ray * 3 + ax] += gr[e] * (-al) / dd;
        }
    }
}


// ---------------------------------------------------------------------------------------------
// The tail of xvr's render_samples (/root/reference/src/xvr/model/trainer.py:289-304) as one pass over the rendered channels:
//   mask = img > 0;  img = img.sum(dim=1);  keep = mean over the pixels of (C == 1 ? mask : any(mask[1:])) > threshold
// One thread per ray reads its C channel values once (coalesced per channel), writes the sum, the C mask bytes and adds its
// foreground bit to the pose's count (wavefront ballot -> one atomic per wavefront); k_foreground_keep turns the counts into
// `keep` with torch's arithmetic (the mean of 0 / 1 floats is the exact count over n, in float32).  Replaces five torch
// launches that read the [B][C][n] image three times (0.9 ms per render at C2 with 8 channels).
// ---------------------------------------------------------------------------------------------
// V = rays per thread: 4 (float4 loads, one 4-byte store of mask bytes per channel; n % 4 == 0) or 1.  All channel loads of a
// thread are issued before the first is used (channels in groups of 8): the pass is a stream, not a chain of dependent loads.
template <int V>
__global__ __launch_bounds__(WG) void k_foreground(const float* __restrict__ img, int C, int n, float* __restrict__ sum,
                                                   unsigned char* __restrict__ mask, int* __restrict__ count) {
    const int b = blockIdx.y, r = (blockIdx.x * WG + threadIdx.x) * V;
    int fg = 0;
    if (r < n) {
        const float* p = img + (size_t)b * C * n + r;
        unsigned char* m = mask + (size_t)b * C * n + r;
        float s[V];
        bool any[V];
#pragma unroll
        for (int i = 0; i < V; ++i) { s[i] = 0.f; any[i] = false; }
        for (int c0 = 0; c0 < C; c0 += 8) {
            float v[8][V];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int c = c0 + k < C ? c0 + k : C - 1;   // (clamped: always loadable)
                if (V == 4) {
                    const float4 t = *reinterpret_cast<const float4*>(p + (size_t)c * n);
                    v[k][0] = t.x; v[k][1 % V] = t.y; v[k][2 % V] = t.z; v[k][3 % V] = t.w;
                } else {
                    v[k][0] = p[(size_t)c * n];
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int c = c0 + k;
                if (c < C) {
                    unsigned bits = 0;
#pragma unroll
                    for (int i = 0; i < V; ++i) {
                        const bool on = v[k][i] > 0.f;
                        s[i] += v[k][i];
                        bits |= on ? (1u << (8 * i)) : 0u;
                        any[i] = any[i] || (on && (c > 0 || C == 1));
                    }
                    if (V == 4) *reinterpret_cast<unsigned*>(m + (size_t)c * n) = bits;
                    else m[(size_t)c * n] = (unsigned char)bits;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < V; ++i) fg += any[i] ? 1 : 0;
        if (V == 4) *reinterpret_cast<float4*>(sum + (size_t)b * n + r) = make_float4(s[0], s[1 % V], s[2 % V], s[3 % V]);
        else sum[(size_t)b * n + r] = s[0];
    }
    const int tot = (int)wave_sum_u((unsigned)fg);
    if ((threadIdx.x & 63) == 0 && tot) atomicAdd(count + b, tot);
}

// (torch's mean on the device is sum * factor with factor = float(outputs) / elements, ATen ReduceMomentKernel: the same
// float here, so that a pose exactly at the threshold falls on the same side)
__global__ void k_foreground_keep(const int* __restrict__ count, int B, int n, float threshold, unsigned char* __restrict__ keep) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    const float factor = (float)B / (float)((long long)B * n);
    if (b < B) keep[b] = ((float)count[b] * factor > threshold) ? 1 : 0;
}

}  // namespace


extern "C" {

size_t xvr_drr_jac_to_camera_workspace_bytes(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    const size_t nblk = ((size_t)H * W + WG - 1) / WG;
    return align256((size_t)B * sizeof(unsigned)) + (size_t)B * nblk * 24 * sizeof(float);
}

int xvr_drr_jac_to_camera_backward(const float* jac, const float* grad_out, const float* cam, int B, int H, int W,
                                   float* grad_cam, void* workspace, size_t workspace_bytes, void* stream) {
    if (!jac || !grad_out || !cam || !grad_cam || !workspace) return fail(XVR_DRR_E_ARG, "null pointer argument");
    if (B <= 0 || H <= 0 || W <= 0) return fail(XVR_DRR_E_ARG, "B, H, W must be positive");
    if (workspace_bytes < xvr_drr_jac_to_camera_workspace_bytes(B, H, W)) return fail(XVR_DRR_E_ARG, "workspace too small");
    if (reinterpret_cast<uintptr_t>(jac) & 15u) return fail(XVR_DRR_E_ARG, "jac must be 16-byte aligned");
    const bool batch = (size_t)B * H * W >= ((size_t)1 << 20);   // (a registration iteration: one to eight 256^2 ... 512^2 images)
    const unsigned per_block = batch ? 4 * WG : WG;
    const unsigned nblk = (unsigned)(((size_t)H * W + per_block - 1) / per_block);
    char* ws = static_cast<char*>(workspace);
    float* partial = reinterpret_cast<float*>(ws + align256((size_t)B * sizeof(unsigned)));
    if (batch) hipLaunchKernelGGL(k_jac_to_cam<4>, dim3(nblk, (unsigned)B), dim3(WG), 0, (hipStream_t)stream, jac, grad_out, cam, H, W, partial,
                                  reinterpret_cast<unsigned*>(ws), grad_cam);
    else hipLaunchKernelGGL(k_jac_to_cam<1>, dim3(nblk, (unsigned)B), dim3(WG), 0, (hipStream_t)stream, jac, grad_out, cam, H, W, partial,
                            reinterpret_cast<unsigned*>(ws), grad_cam);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(XVR_DRR_E_LAUNCH, hipGetErrorString(e));
    return XVR_DRR_OK;
}

int xvr_drr_rays_forward(const float* cam, int B, int H, int W, float* source, float* target, float* raylen,
                         void* stream) {
    if (!cam || !source || !target || !raylen) return fail(XVR_DRR_E_ARG, "null pointer argument");
    if (B <= 0 || H <= 0 || W <= 0) return fail(XVR_DRR_E_ARG, "B, H, W must be positive");
    dim3 grid((unsigned)(((long long)H * W + WG - 1) / WG), (unsigned)B);
    hipLaunchKernelGGL(k_rays_fwd, grid, dim3(WG), 0, (hipStream_t)stream, cam, H, W, source, target, raylen);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(XVR_DRR_E_LAUNCH, hipGetErrorString(e));
    return XVR_DRR_OK;
}

int xvr_drr_rays_backward(const float* cam, int B, int H, int W, const float* grad_source, const float* grad_target,
                          const float* grad_raylen, float* grad_cam, void* stream) {
    if (!cam || !grad_target || !grad_cam) return fail(XVR_DRR_E_ARG, "null pointer argument");
    if (B <= 0 || H <= 0 || W <= 0) return fail(XVR_DRR_E_ARG, "B, H, W must be positive");
    const bool batch = (size_t)B * H * W >= ((size_t)1 << 20);
    const long long per_block = batch ? 4 * WG : WG;
    dim3 grid((unsigned)(((long long)H * W + per_block - 1) / per_block), (unsigned)B);
    if (batch) hipLaunchKernelGGL(k_rays_bwd<4>, grid, dim3(WG), 0, (hipStream_t)stream, cam, H, W, grad_source, grad_target, grad_raylen, grad_cam);
    else hipLaunchKernelGGL(k_rays_bwd<1>, grid, dim3(WG), 0, (hipStream_t)stream, cam, H, W, grad_source, grad_target, grad_raylen, grad_cam);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(XVR_DRR_E_LAUNCH, hipGetErrorString(e));
    return XVR_DRR_OK;
}

int xvr_drr_backward_from_jac(const float* jac, const float* grad_out, int B, int n, float* grad_source,
                              float* grad_target, float* grad_raylen, void* stream) {
    if (!jac || !grad_out || !grad_source || !grad_target) return fail(XVR_DRR_E_ARG, "null pointer argument");
    if (B <= 0 || n <= 0) return fail(XVR_DRR_E_ARG, "B and n must be positive");
    const bool batch = (size_t)B * n >= ((size_t)1 << 20);
    const int per_block = batch ? 4 * WG : WG;
    dim3 grid((unsigned)((n + per_block - 1) / per_block), (unsigned)B);
    if (batch) hipLaunchKernelGGL(k_backward_from_jac<4>, grid, dim3(WG), 0, (hipStream_t)stream, jac, grad_out, n, grad_source, grad_target, grad_raylen);
    else hipLaunchKernelGGL(k_backward_from_jac<1>, grid, dim3(WG), 0, (hipStream_t)stream, jac, grad_out, n, grad_source, grad_target, grad_raylen);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(XVR_DRR_E_LAUNCH, hipGetErrorString(e));
    return XVR_DRR_OK;
}

size_t xvr_drr_alpha_window_bytes(int B) {   // the header + one 16-byte slot per pose (two keys / two partial sums)
    return B > 0 ? ((size_t)XVR_DRR_ALPHA_WINDOW_FLOATS + 4 * (size_t)B) * sizeof(float) : 0;
}

int xvr_drr_alpha_window(const float* source, const float* target, int B, int n, int D0, int D1, int D2,
                         const xvr_drr_spec* sp, float* window, void* stream) {
    if (!source || !target || !sp || !window) return fail(XVR_DRR_E_ARG, "null pointer argument");
    if (B <= 0 || n <= 0 || D0 < 2 || D1 < 2 || D2 < 2) return fail(XVR_DRR_E_ARG, "B, n must be positive and every volume dimension >= 2");
    if ((long long)B * n >= (1LL << 32) - 1) return fail(XVR_DRR_E_UNSUPPORTED, "the alpha window indexes rays with 32 bits");
    if (reinterpret_cast<uintptr_t>(window) & 15u) return fail(XVR_DRR_E_ARG, "window must be 16-byte aligned");
    hipError_t e = hipMemsetAsync(window, 0, xvr_drr_alpha_window_bytes(B), (hipStream_t)stream);
    if (e != hipSuccess) return fail(XVR_DRR_E_LAUNCH, hipGetErrorString(e));
    hipLaunchKernelGGL(k_alpha_window_reduce, dim3((unsigned)((n + WG - 1) / WG), (unsigned)B), dim3(WG), 0, (hipStream_t)stream,
                       source, target, n, *sp, reinterpret_cast<unsigned long long*>(window + XVR_DRR_ALPHA_WINDOW_FLOATS));
    hipLaunchKernelGGL(k_alpha_window_final, dim3(1), dim3(64), 0, (hipStream_t)stream, window, B, sp->near_, sp->far_, sp->inv_denom);
    e = hipGetLastError();
    if (e != hipSuccess) return fail(XVR_DRR_E_LAUNCH, hipGetErrorString(e));
    return XVR_DRR_OK;
}

int xvr_drr_alpha_window_backward(const float* jac, const float* grad_out, const float* source, const float* target,
                                  const float* raylen, int B, int n, const xvr_drr_spec* sp, float* window,
                                  float* grad_source, float* grad_target, void* stream) {
    if (!jac || !grad_out || !source || !target || !raylen || !sp || !window || !grad_source || !grad_target)
        return fail(XVR_DRR_E_ARG, "null pointer argument");
    if (B <= 0 || n <= 0) return fail(XVR_DRR_E_ARG, "B and n must be positive");
    hipError_t e = hipMemsetAsync(window + XVR_DRR_ALPHA_WINDOW_FLOATS, 0, (size_t)4 * B * sizeof(float), (hipStream_t)stream);   // the poses' slots
    if (e != hipSuccess) return fail(XVR_DRR_E_LAUNCH, hipGetErrorString(e));
    hipLaunchKernelGGL(k_alpha_window_bwd_reduce, dim3((unsigned)((n + WG - 1) / WG), (unsigned)B), dim3(WG), 0, (hipStream_t)stream,
                       jac, grad_out, source, target, raylen, n, sp->eps, window);
    hipLaunchKernelGGL(k_alpha_window_bwd_apply, dim3(1), dim3(1), 0, (hipStream_t)stream, source, target, B, n, *sp, window,
                       grad_source, grad_target);
    e = hipGetLastError();
    if (e != hipSuccess) return fail(XVR_DRR_E_LAUNCH, hipGetErrorString(e));
    return XVR_DRR_OK;
}

int xvr_drr_foreground(const float* img, int B, int C, int n, float threshold, float* sum, unsigned char* mask, int* count,
                       unsigned char* keep, void* stream) {
    if (!img || !sum || !mask || !count || !keep) return fail(XVR_DRR_E_ARG, "null pointer argument");
    if (B <= 0 || n <= 0 || C < 1) return fail(XVR_DRR_E_ARG, "B, C and n must be positive");
    if (B > 65535) return fail(XVR_DRR_E_UNSUPPORTED, "more than 65535 poses");
    hipError_t e = hipMemsetAsync(count, 0, (size_t)B * sizeof(int), (hipStream_t)stream);
    if (e != hipSuccess) return fail(XVR_DRR_E_LAUNCH, hipGetErrorString(e));
    const bool vec = n % 4 == 0 && (reinterpret_cast<uintptr_t>(img) | reinterpret_cast<uintptr_t>(sum)) % 16 == 0 && reinterpret_cast<uintptr_t>(mask) % 4 == 0;
    if (vec) hipLaunchKernelGGL(k_foreground<4>, dim3((unsigned)((n / 4 + WG - 1) / WG), (unsigned)B), dim3(WG), 0, (hipStream_t)stream, img, C, n, sum, mask, count);
    else hipLaunchKernelGGL(k_foreground<1>, dim3((unsigned)((n + WG - 1) / WG), (unsigned)B), dim3(WG), 0, (hipStream_t)stream, img, C, n, sum, mask, count);
    hipLaunchKernelGGL(k_foreground_keep, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, (hipStream_t)stream, count, B, n, threshold, keep);
    e = hipGetLastError();
    if (e != hipSuccess) return fail(XVR_DRR_E_LAUNCH, hipGetErrorString(e));
    return XVR_DRR_OK;
}

}  // extern "C"
