// MI355X (gfx950 / CDNA4) differentiable-DRR kernels + the C ABI of include/xvr_drr.h.
//
// One lane = one ray; one 64-lane wavefront = an 8x8 pixel tile of one pose's detector, so the 64
// rays of a wave form a narrow frustum and, because every ray of a pose samples the SAME alpha_k
// (alphas = linspace(near, far, n_points) is shared), the wave's 64 samples at step k lie on a small
// planar patch: their 8 x 64 taps fall into a few hundred bytes of neighbouring voxel rows and are
// served by the CU's L1 / the XCD's L2 rather than HBM.  Workgroups (4 waves = a 16x16 pixel tile)
// are renumbered so that each XCD works through whole poses (its L2 then holds one frustum at a time).
//
// No MFMA anywhere: this is gather + interpolate, not a contraction.
//
// Contents, in file order:
//   launch geometry: RenderArgs, xcd_remap, map_ray, ray_setup, trilinear taps
//   trilinear forward (+jacobian): tri_march / tri_finish / k_trilinear_fwd
//     sample-split forward for small launches: k_trilinear_fwd_split
//     LDS-staged bricks (opt-in, slower): k_trilinear_fwd_lds
//   trilinear re-march backward / atomic scatter fallback: k_trilinear_bwd
//   voxel gradient as a gather: PoseLattice, k_gather_prep, k_gather_cull
//     k_trilinear_gather_vol
//     k_siddon_gather_vol
//   pose-side backward from the jacobian: k_backward_from_jac
//   Siddon traversal, forward / jacobian / backward / alpha-split: k_siddon
//   ray generation and its adjoint: k_rays_fwd, k_rays_bwd, k_jac_to_cam
//   host side: argument checks, launch helpers, split_factor, workspace layout
//   C ABI entry points (extern "C")
//
// Semantics are those of oracle/diffdrr_restated.py (the restated diffdrr==0.6.0 algorithm; every
// unpinned constant arrives through xvr_drr_spec).  Reference call sites being replaced:
//   /root/reference/src/xvr/model/trainer.py:288   drr.renderer(volume, source, target, img, mask=seg)
//   /root/reference/src/xvr/registrar/base.py:249,252   reg() ... loss.backward()
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "xvr_drr.h"

namespace {

constexpr int WG = 256;  // 4 wavefronts

// voxel-driven gather (see k_trilinear_gather_vol)
constexpr float GATHER_DEV_TOL = 0.02f;     // max lattice deviation, in units of the pixel pitch
constexpr float GATHER_WIN_MARGIN = 0.05f;  // extra half-width (pixels) of the candidate window
constexpr float GATHER_K_SLACK = 1e-3f;

thread_local char g_err[512] = "";

struct RenderArgs {
    const float* __restrict__ volume;
    const float* __restrict__ mask;
    int D0, D1, D2, C;
    const float* __restrict__ source;
    const float* __restrict__ target;
    const float* __restrict__ raylen;
    const float* __restrict__ cam;  // nullable [B][24]: rays are generated from it (k_rays_fwd's arithmetic) instead of loaded
    int B, n;
    xvr_drr_spec sp;
    int grid_w, grid_h, tiles_x, blocks_per_pose;
    int tile_shape;  // -1: choose per workgroup; 0: 8x8 per wave; 1: 16 wide x 4 tall; 2: 4 wide x 16 tall
    float* __restrict__ out;
    float* __restrict__ jac;
    unsigned long long* work;
    const float* __restrict__ gout;
    float* gvol;
    const unsigned* skip_unless_flag_gt;  // nullable: run the scatter only if *flag > GATHER_DEV_TOL
    float* gsrc;
    float* __restrict__ gtgt;
    float* __restrict__ glen;
};

struct __attribute__((packed, aligned(4))) fpair {
    float x, y;
};

__device__ __forceinline__ fpair load_pair(const float* p) { return *reinterpret_cast<const fpair*>(p); }

__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ unsigned wave_sum_u(unsigned v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__device__ __forceinline__ void atomic_add_f32(float* p, float v) {
    // hardware fp32 add at the L2 / memory side (no CAS loop); agent scope so that XCDs agree
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Blocks are dispatched round-robin over the 8 XCDs (block i -> XCD i % 8, observed, speed only).
// Renumber so that XCD x works on one contiguous range of logical blocks, i.e. on whole poses:
// its private 4 MiB L2 then serves the neighbouring tiles of one frustum instead of 8 different ones.
__device__ __forceinline__ unsigned xcd_remap(unsigned i, unsigned nb) {
    unsigned q = nb >> 3, rem = nb & 7u;
    unsigned x = i & 7u, j = i >> 3;
    return x * q + (x < rem ? x : rem) + j;
}

__device__ __forceinline__ bool map_ray(const RenderArgs& A, int& b, int& r, int tid) {
    unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    b = (int)(lb / (unsigned)A.blocks_per_pose);
    int t = (int)(lb - (unsigned)b * (unsigned)A.blocks_per_pose);
    if (A.grid_w > 0) {
        int ty = t / A.tiles_x, tx = t - ty * A.tiles_x;
        const int px0 = tx * 16, py0 = ty * 16;
        // Shape of the 64-pixel patch each wavefront takes out of the workgroup's 16x16 tile.  Voxel rows
        // are contiguous along z (volume axis 2): lanes that differ along the detector axis which runs
        // along z share cache lines, lanes that differ along the other axis cost a line each.  So make the
        // wave long along the z-aligned detector axis (decided per workgroup from two neighbouring rays).
        int shape = A.tile_shape;
        if (shape < 0) {
            shape = 0;
            if (A.grid_w > 1 && A.grid_h > 1) {
                const int pc = px0 + 1 < A.grid_w ? px0 + 1 : px0 - 1;
                const int pr = py0 + 1 < A.grid_h ? py0 + 1 : py0 - 1;
                float zc, zr;   // how far the target moves along z per detector column / row
                if (A.cam) {
                    zc = fabsf(A.cam[24 * b + 7]);
                    zr = fabsf(A.cam[24 * b + 6]);
                } else {
                    const float* T = A.target + (size_t)b * A.n * 3;
                    const float z00 = T[((size_t)py0 * A.grid_w + px0) * 3 + 2];
                    zc = fabsf(T[((size_t)py0 * A.grid_w + pc) * 3 + 2] - z00);
                    zr = fabsf(T[((size_t)pr * A.grid_w + px0) * 3 + 2] - z00);
                }
                shape = zc > 2.f * zr ? 1 : (zr > 2.f * zc ? 2 : 0);
            }
        }
        const int w = tid >> 6, l = tid & 63;
        int dx, dy;
        if (shape == 1) { dx = l & 15; dy = w * 4 + (l >> 4); }
        else if (shape == 2) { dx = w * 4 + (l & 3); dy = l >> 2; }
        else { dx = (w & 1) * 8 + (l & 7); dy = (w >> 1) * 8 + (l >> 3); }
        const int px = px0 + dx, py = py0 + dy;
        r = py * A.grid_w + px;
        return px < A.grid_w && py < A.grid_h;
    }
    r = t * WG + tid;
    return r < A.n;
}

// ---------------------------------------------------------------------------------------------
// per-ray setup shared by all kernels
// ---------------------------------------------------------------------------------------------
struct Ray {
    float s[3], d[3], L;
    float amin, amax;   // slab test against the volume's bounding planes, clamped to [0, 1]
    int ax_in, ax_out;  // axis whose plane gives amin / amax; -1 when clamped to 0 / 1
    bool valid;
};

__device__ __forceinline__ void ray_setup(const RenderArgs& A, int b, int r, bool valid, Ray& R) {
    R.valid = valid;
    float t[3];
    R.L = 0.f;
    if (A.cam) {   // rays from the camera vector, with exactly k_rays_fwd's arithmetic (detector lattice only)
        const float* c = A.cam + 24 * b;
#pragma unroll
        for (int i = 0; i < 3; ++i) t[i] = R.s[i] = c[9 + i];
        if (valid) {
            const int pi = r / A.grid_w, pj = r - pi * A.grid_w;
            const float fi = (float)pi, fj = (float)pj;
            float l2 = 0.f;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                t[a] = fmaf(c[3 * a], fi, fmaf(c[3 * a + 1], fj, c[3 * a + 2]));
                const float w = fmaf(c[12 + 3 * a], fi, fmaf(c[12 + 3 * a + 1], fj, c[12 + 3 * a + 2])) - c[21 + a];
                l2 = fmaf(w, w, l2);
            }
            R.L = sqrtf(l2);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 3; ++i) t[i] = R.s[i] = A.source[3 * b + i];
        if (valid) {
            const float* tp = A.target + ((size_t)b * A.n + r) * 3;
            t[0] = tp[0];
            t[1] = tp[1];
            t[2] = tp[2];
            R.L = A.raylen[(size_t)b * A.n + r];
        }
    }
    float lo = -INFINITY, hi = INFINITY;
    int ain = -1, aout = -1;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        R.d[i] = (t[i] - R.s[i]) + A.sp.eps;
        float a0 = (A.sp.lo[i] - R.s[i]) / R.d[i];
        float a1 = (A.sp.hi[i] - R.s[i]) / R.d[i];
        float mn = fminf(a0, a1), mx = fmaxf(a0, a1);
        if (mn > lo) { lo = mn; ain = i; }
        if (mx < hi) { hi = mx; aout = i; }
    }
    if (!(lo > 0.f)) { lo = 0.f; ain = -1; }
    if (!(hi < 1.f)) { hi = 1.f; aout = -1; }
    R.amin = lo;
    R.amax = hi;
    R.ax_in = ain;
    R.ax_out = aout;
}

// ---------------------------------------------------------------------------------------------
// trilinear taps: grid_sample(mode="bilinear", padding_mode="zeros") on the index point (px,py,pz)
// ---------------------------------------------------------------------------------------------
struct Taps {
    int base[4];        // element offsets of rows (x0,y0) (x0,y1) (x1,y0) (x1,y1), at z = zc
    float wx0, wx1, wy0, wy1;  // interpolation weights (0 where the corner is outside)
    float sx0, sx1, sy0, sy1;  // d/dx, d/dy weights (-1/+1, 0 where the corner is outside)
    float pz0, pz1, qz0, qz1;  // weights / d/dz weights on the loaded (zc, zc+1) pair
};

__device__ __forceinline__ void make_taps(float px, float py, float pz, int D0, int D1, int D2, Taps& T) {
    float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
    int ix = (int)fx, iy = (int)fy, iz = (int)fz;
    float tx = px - fx, ty = py - fy, tz = pz - fz;
    bool x0 = (unsigned)ix < (unsigned)D0, x1 = (unsigned)(ix + 1) < (unsigned)D0;
    bool y0 = (unsigned)iy < (unsigned)D1, y1 = (unsigned)(iy + 1) < (unsigned)D1;
    T.wx0 = x0 ? 1.f - tx : 0.f;
    T.wx1 = x1 ? tx : 0.f;
    T.wy0 = y0 ? 1.f - ty : 0.f;
    T.wy1 = y1 ? ty : 0.f;
    T.sx0 = x0 ? -1.f : 0.f;
    T.sx1 = x1 ? 1.f : 0.f;
    T.sy0 = y0 ? -1.f : 0.f;
    T.sy1 = y1 ? 1.f : 0.f;
    int cx0 = min(max(ix, 0), D0 - 1), cx1 = min(max(ix + 1, 0), D0 - 1);
    int cy0 = min(max(iy, 0), D1 - 1), cy1 = min(max(iy + 1, 0), D1 - 1);
    // the two z taps are adjacent in memory: one 8-byte load of (zc, zc+1), zc clamped so the pair
    // is always inside the row; sh says where the wanted pair (iz, iz+1) sits relative to it.
    int zc = min(max(iz, 0), D2 - 2);
    int sh = iz - zc;
    T.pz0 = sh == 0 ? 1.f - tz : (sh == -1 ? tz : 0.f);
    T.pz1 = sh == 0 ? tz : (sh == 1 ? 1.f - tz : 0.f);
    T.qz0 = sh == 0 ? -1.f : (sh == -1 ? 1.f : 0.f);
    T.qz1 = sh == 0 ? 1.f : (sh == 1 ? -1.f : 0.f);
    T.base[0] = (cx0 * D1 + cy0) * D2 + zc;
    T.base[1] = (cx0 * D1 + cy1) * D2 + zc;
    T.base[2] = (cx1 * D1 + cy0) * D2 + zc;
    T.base[3] = (cx1 * D1 + cy1) * D2 + zc;
}

__device__ __forceinline__ int nearest_label(const float* __restrict__ mask, float px, float py, float pz,
                                             int D0, int D1, int D2, int C) {
    int lx = (int)rintf(px), ly = (int)rintf(py), lz = (int)rintf(pz);
    bool inb = (unsigned)lx < (unsigned)D0 && (unsigned)ly < (unsigned)D1 && (unsigned)lz < (unsigned)D2;
    int lab = inb ? (int)mask[(lx * D1 + ly) * D2 + lz] : 0;
    return min(max(lab, 0), C - 1);
}

// Labels packed into the volume (xvr_drr_pack_labels: the low 4 mantissa bits of every voxel hold its
// label): the nearest voxel of a sample is one of the 8 taps the interpolation has just loaded, so the
// label costs a few selects instead of a fifth gather (the separate lookup makes the masked march 1.5x
// slower than the unmasked one).  Same rule as nearest_label: rintf per axis, 0 outside the volume.
constexpr unsigned LABEL_BITS = 4, LABEL_MASK = (1u << LABEL_BITS) - 1u;
__device__ __forceinline__ int packed_label(const fpair (&P)[4], float px, float py, float pz, int D0, int D1, int D2, int C) {
    const int lx = (int)rintf(px), ly = (int)rintf(py), lz = (int)rintf(pz);
    const bool inb = (unsigned)lx < (unsigned)D0 && (unsigned)ly < (unsigned)D1 && (unsigned)lz < (unsigned)D2;
    const int sx = lx - (int)floorf(px), sy = ly - (int)floorf(py);           // 0: the floor row, 1: the next one
    const int zc = min(max((int)floorf(pz), 0), D2 - 2);                      // first element of the loaded z pair
    const fpair r0 = sy ? P[1] : P[0], r1 = sy ? P[3] : P[2];
    const fpair r = sx ? r1 : r0;
    const unsigned bits = __float_as_uint(lz > zc ? r.y : r.x);
    return inb ? min((int)(bits & LABEL_MASK), C - 1) : 0;
}

// torch.linspace(near, far, N)[k] (symmetric evaluation, as ATen computes it)
__device__ __forceinline__ float linspace_at(int k, int N, float near_, float far_, float step) {
    return (k < N / 2) ? fmaf(step, (float)k, near_) : far_ - step * (float)(N - 1 - k);
}

struct KRange {
    int lo, hi;
};

// Range of sample indices whose 8 taps can touch the volume (exact outside: those samples read only
// zero padding and contribute exactly 0, so skipping them changes nothing).
__device__ __forceinline__ KRange tri_krange(const RenderArgs& A, const Ray& R, bool clip, float step) {
    KRange K = {INT32_MAX, INT32_MIN};  // empty
    if (!R.valid) return K;
    const int N = A.sp.n_points;
    if (clip) {
        if (R.amax > R.amin) { K.lo = 0; K.hi = N - 1; }
        return K;
    }
    float ain = -INFINITY, aout = INFINITY;
    const int S[3] = {A.D0, A.D1, A.D2};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float xlo = (-1.f - A.sp.b[i]) / A.sp.a[i];
        float xhi = ((float)S[i] - A.sp.b[i]) / A.sp.a[i];
        float a0 = (xlo - R.s[i]) / R.d[i], a1 = (xhi - R.s[i]) / R.d[i];
        ain = fmaxf(ain, fminf(a0, a1));
        aout = fminf(aout, fmaxf(a0, a1));
    }
    if (!(aout >= ain)) return K;
    float fN = (float)N;
    float klo = fminf(fmaxf((ain - A.sp.near_) / step - 1.f, 0.f), fN);
    float khi = fminf(fmaxf((aout - A.sp.near_) / step + 1.f, -1.f), fN - 1.f);
    K.lo = (int)ceilf(klo);
    K.hi = (int)floorf(khi);
    return K;
}

// =============================================================================================
// trilinear forward (+ optional per-ray jacobian in the same sweep)
// =============================================================================================
struct TriAcc {  // per-lane sums of one ray (or of one slice of its samples)
    float S, G[3], H[3], E0, E1;
    unsigned cnt;
};

// Samples kbeg..kend (wave-uniform bounds; lanes mask themselves with their own K) of the lane's ray.
// MASK: 0 = one channel; 1 = labels from a separate mask volume; 2 = labels packed into the volume's taps
template <bool JAC, int MASK, bool CLIP>
__device__ __forceinline__ void tri_march(const RenderArgs& A, const Ray& R, const KRange K, const int kbeg, const int kend,
                                          const float step, float* lds, const int tid, TriAcc& acc) {
    const int N = A.sp.n_points;
    const float* __restrict__ vol = A.volume;
    const int D0 = A.D0, D1 = A.D1, D2 = A.D2;
    float S = 0.f;
    float G[3] = {0.f, 0.f, 0.f}, H[3] = {0.f, 0.f, 0.f};
    float E0 = 0.f, E1 = 0.f;
    unsigned cnt = 0;
    const float adx = A.sp.a[0] * R.d[0], ady = A.sp.a[1] * R.d[1], adz = A.sp.a[2] * R.d[2];

    // Two steps per trip: the 8 independent 8-byte gathers of both samples are issued before either
    // is consumed (the march is latency-bound, not bandwidth-bound: L2 at ~20 %, HBM at ~25 %).
    for (int kk = kbeg; kk <= kend; kk += 2) {
        bool act[2];
        float u[2], al[2], pxs[2], pys[2], pzs[2];
        Taps T[2];
        fpair P[2][4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = kk + h;
            act[h] = k >= K.lo && k <= K.hi && k <= kend;
            u[h] = linspace_at(k, N, A.sp.near_, A.sp.far_, step);
            al[h] = CLIP ? fmaf(u[h], R.amax - R.amin, R.amin) : u[h];
            pxs[h] = fmaf(A.sp.a[0], fmaf(al[h], R.d[0], R.s[0]), A.sp.b[0]);
            pys[h] = fmaf(A.sp.a[1], fmaf(al[h], R.d[1], R.s[1]), A.sp.b[1]);
            pzs[h] = fmaf(A.sp.a[2], fmaf(al[h], R.d[2], R.s[2]), A.sp.b[2]);
            make_taps(pxs[h], pys[h], pzs[h], D0, D1, D2, T[h]);  // offsets are clamped: always loadable
        }
        // unconditional (offsets are clamped into the volume): a branch here would split the loads into
        // two exec-masked blocks with a full vmcnt(0) drain between them
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int q = 0; q < 4; ++q) P[h][q] = load_pair(vol + T[h].base[q]);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (!act[h]) continue;
            const Taps& t = T[h];
            const float v0 = fmaf(t.pz1, P[h][0].y, t.pz0 * P[h][0].x), v1 = fmaf(t.pz1, P[h][1].y, t.pz0 * P[h][1].x);
            const float v2 = fmaf(t.pz1, P[h][2].y, t.pz0 * P[h][2].x), v3 = fmaf(t.pz1, P[h][3].y, t.pz0 * P[h][3].x);
            const float r0 = fmaf(t.wy1, v1, t.wy0 * v0), r1 = fmaf(t.wy1, v3, t.wy0 * v2);
            const float v = fmaf(t.wx1, r1, t.wx0 * r0);
            ++cnt;
            if (MASK) {
                const int lab = MASK == 2 ? packed_label(P[h], pxs[h], pys[h], pzs[h], D0, D1, D2, A.C)
                                          : nearest_label(A.mask, pxs[h], pys[h], pzs[h], D0, D1, D2, A.C);
                lds[lab * WG + tid] += v;
                if (JAC) S += v;  // the jacobian saved with a mask is that of the channel SUM
            } else {
                S += v;
            }
            if (JAC) {
                const float d0 = fmaf(t.qz1, P[h][0].y, t.qz0 * P[h][0].x), d1 = fmaf(t.qz1, P[h][1].y, t.qz0 * P[h][1].x);
                const float d2 = fmaf(t.qz1, P[h][2].y, t.qz0 * P[h][2].x), d3 = fmaf(t.qz1, P[h][3].y, t.qz0 * P[h][3].x);
                const float gz = fmaf(t.wx1, fmaf(t.wy1, d3, t.wy0 * d2), t.wx0 * fmaf(t.wy1, d1, t.wy0 * d0));
                const float gx = fmaf(t.sx1, r1, t.sx0 * r0);
                const float gy = fmaf(t.wx1, fmaf(t.sy1, v3, t.sy0 * v2), t.wx0 * fmaf(t.sy1, v1, t.sy0 * v0));
                G[0] += gx; G[1] += gy; G[2] += gz;
                H[0] = fmaf(al[h], gx, H[0]); H[1] = fmaf(al[h], gy, H[1]); H[2] = fmaf(al[h], gz, H[2]);
                if (CLIP) {
                    const float gd = fmaf(gx, adx, fmaf(gy, ady, gz * adz));
                    E0 = fmaf(gd, 1.f - u[h], E0);
                    E1 = fmaf(gd, u[h], E1);
                }
            }
        }
    }
    acc.S = S;
    acc.cnt = cnt;
#pragma unroll
    for (int i = 0; i < 3; ++i) { acc.G[i] = G[i]; acc.H[i] = H[i]; }
    acc.E0 = E0;
    acc.E1 = E1;
}

// Scale the sums and write the pixel (and its jacobian row).
template <bool JAC, int MASK, bool CLIP>
__device__ __forceinline__ void tri_finish(const RenderArgs& A, const Ray& R, const int b, const int r, const float* lds,
                                           const int tid, const TriAcc& acc) {
    const float S = acc.S;
    const float span = fmaxf(R.amax - R.amin, 0.f);
    const float base_scale = R.L * A.sp.inv_denom;
    const float scale = CLIP ? base_scale * span : base_scale;
    if (MASK) {
        for (int c = 0; c < A.C; ++c) A.out[((size_t)b * A.C + c) * A.n + r] = lds[c * WG + tid] * scale;
    } else {
        A.out[(size_t)b * A.n + r] = S * scale;
    }
    if (JAC) {
        float js[3], jt[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            jt[i] = scale * A.sp.a[i] * acc.H[i];
            js[i] = scale * A.sp.a[i] * (acc.G[i] - acc.H[i]);
        }
        if (CLIP) {
            const float dmin = base_scale * (-S + span * acc.E0), dmax = base_scale * (S + span * acc.E1);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                if (R.ax_in == i && span > 0.f) {
                    js[i] += dmin * (R.amin - 1.f) / R.d[i];
                    jt[i] += dmin * (-R.amin) / R.d[i];
                }
                if (R.ax_out == i && span > 0.f) {
                    js[i] += dmax * (R.amax - 1.f) / R.d[i];
                    jt[i] += dmax * (-R.amax) / R.d[i];
                }
            }
        }
        float4* jp = reinterpret_cast<float4*>(A.jac + ((size_t)b * A.n + r) * XVR_DRR_JAC_STRIDE);
        jp[0] = make_float4(S * (CLIP ? A.sp.inv_denom * span : A.sp.inv_denom), js[0], js[1], js[2]);
        jp[1] = make_float4(jt[0], jt[1], jt[2], 0.f);
    }
}

// At most 4 wavefronts per SIMD: the march is bound by the texture-address unit, not by latency hiding, and
// more resident wavefronts only thrash the L1/L2 -- the variant without the jacobian needs 64 VGPRs, ran at
// 8 wavefronts per SIMD and took 8.2 ms where the (heavier) jacobian variant at 6 took 7.0; capped, both take
// ~7.0 ms (measured flat from 3 to 6, worse at 2 and at 8).
template <bool JAC, int MASK, bool CLIP>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(1, 4))) void k_trilinear_fwd(RenderArgs A) {
    extern __shared__ float lds[];  // MASK: per-lane channel accumulators [C][WG]
    int b, r;
    const bool valid = map_ray(A, b, r, threadIdx.x);
    const int tid = threadIdx.x;
    Ray R;
    ray_setup(A, b, r, valid, R);
    const int N = A.sp.n_points;
    const float step = N > 1 ? (A.sp.far_ - A.sp.near_) / (float)(N - 1) : 0.f;
    const KRange K = tri_krange(A, R, CLIP, step);
    const int kbeg = __builtin_amdgcn_readfirstlane(wave_min_i(K.lo));
    const int kend = __builtin_amdgcn_readfirstlane(wave_max_i(K.hi));
    if (MASK) {
        for (int c = 0; c < A.C; ++c) lds[c * WG + tid] = 0.f;
    }
    TriAcc acc;
    tri_march<JAC, MASK, CLIP>(A, R, K, kbeg, kend, step, lds, tid, acc);
    if (valid) tri_finish<JAC, MASK, CLIP>(A, R, b, r, lds, tid, acc);
    if (A.work) {
        unsigned tot = wave_sum_u(acc.cnt);
        if ((tid & 63) == 0 && tot) atomicAdd(A.work, (unsigned long long)tot);
    }
}

// ---------------------------------------------------------------------------------------------
// Sample-split forward for SMALL batches (registration renders one pose: 256^2 rays are 1024
// wavefronts, one per SIMD, and every wavefront walks its ~250 in-volume samples with nothing to hide
// the gather latency behind).  Here the 64 x NS lanes of a workgroup share ONE 8x8 pixel tile:
// wavefront w marches the w-th slice of the samples, the partial sums meet in LDS and wavefront 0
// writes the pixel.  Sums are combined in a fixed order (slice 0, 1, 2, ...): deterministic, but the
// rounding differs from the unsplit kernel's single running sum (same tolerance against the oracle).
// ---------------------------------------------------------------------------------------------
constexpr int SPLIT_MAX = 16;
constexpr int SPLIT_VALS = 9;  // S, G[3], H[3], E0, E1

// Lane -> (pose, ray, slice) in the split kernels.  TILE16 = false: the workgroup is ONE 8x8 tile x NS
// slices; true: the unsplit kernels' 16x16 tile (4 wavefronts, shape-adaptive) x NS slices.
template <bool TILE16>
__device__ __forceinline__ bool map_ray_split(const RenderArgs& A, int& b, int& r, int& l, int& w, int& NS) {
    constexpr int TL = TILE16 ? 256 : 64;
    const int tid = threadIdx.x;
    l = tid & (TL - 1);
    w = tid / TL;
    NS = blockDim.x / TL;
    if (TILE16) return map_ray(A, b, r, l);
    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    b = (int)(lb / (unsigned)A.blocks_per_pose);
    const int t = (int)(lb - (unsigned)b * (unsigned)A.blocks_per_pose);
    if (A.grid_w > 0) {
        const int ty = t / A.tiles_x, tx = t - ty * A.tiles_x;
        const int px = tx * 8 + (l & 7), py = ty * 8 + (l >> 3);
        r = py * A.grid_w + px;
        return px < A.grid_w && py < A.grid_h;
    }
    r = t * 64 + l;
    return r < A.n;
}

// TILE16 = false (NS <= 16) spreads tiny launches over many CUs; TILE16 = true (NS <= 4) keeps the
// wavefronts of one slice marching neighbouring tiles in step, sharing cache lines as in the unsplit kernel.
template <bool JAC, bool CLIP, bool TILE16>
__global__ __launch_bounds__(64 * SPLIT_MAX) void k_trilinear_fwd_split(RenderArgs A) {
    extern __shared__ float lds[];  // [NS - 1][SPLIT_VALS][TILE16 ? 256 : 64]
    constexpr int TL = TILE16 ? 256 : 64;   // rays per workgroup
    const int tid = threadIdx.x;
    int b, r, l, w, NS;
    const bool valid = map_ray_split<TILE16>(A, b, r, l, w, NS);
    Ray R;
    ray_setup(A, b, r, valid, R);
    const int N = A.sp.n_points;
    const float step = N > 1 ? (A.sp.far_ - A.sp.near_) / (float)(N - 1) : 0.f;
    const KRange K = tri_krange(A, R, CLIP, step);
    const int kbeg = __builtin_amdgcn_readfirstlane(wave_min_i(K.lo));
    const int kend = __builtin_amdgcn_readfirstlane(wave_max_i(K.hi));
    TriAcc acc;
    {
        const int len = kend >= kbeg ? kend - kbeg + 1 : 0;
        const int chunk = (((len + NS - 1) / NS) + 1) & ~1;   // even: the march takes two samples per trip
        const int my_beg = kbeg + w * chunk;
        const int my_end = min(kend, my_beg + chunk - 1);
        tri_march<JAC, 0, CLIP>(A, R, K, my_beg, my_end, step, nullptr, tid, acc);
    }
    if (w > 0) {
        float* p = lds + (size_t)(w - 1) * SPLIT_VALS * TL + l;
        p[0] = acc.S;
        if (JAC) {
#pragma unroll
            for (int i = 0; i < 3; ++i) { p[(1 + i) * TL] = acc.G[i]; p[(4 + i) * TL] = acc.H[i]; }
            if (CLIP) { p[7 * TL] = acc.E0; p[8 * TL] = acc.E1; }
        }
    }
    __syncthreads();
    if (w == 0) {
        for (int v = 1; v < NS; ++v) {
            const float* p = lds + (size_t)(v - 1) * SPLIT_VALS * TL + l;
            acc.S += p[0];
            if (JAC) {
#pragma unroll
                for (int i = 0; i < 3; ++i) { acc.G[i] += p[(1 + i) * TL]; acc.H[i] += p[(4 + i) * TL]; }
                if (CLIP) { acc.E0 += p[7 * TL]; acc.E1 += p[8 * TL]; }
            }
        }
        if (valid) tri_finish<JAC, 0, CLIP>(A, R, b, r, nullptr, tid, acc);
    }
    if (A.work) {
        unsigned tot = wave_sum_u(acc.cnt);
        if ((tid & 63) == 0 && tot) atomicAdd(A.work, (unsigned long long)tot);
    }
}


// =============================================================================================
// trilinear forward with LDS-staged voxel bricks
//
// The direct kernel above is limited by how many distinct cache lines the texture-address unit must
// visit per gather instruction (64 lanes x 8 B spread over 10-40 lines).  Here the workgroup (a 16x16
// pixel tile = 256 rays) walks its rays in chunks of KC steps; per chunk it computes a conservative
// bounding brick of every tap its rays will make, loads that brick once with row-contiguous loads
// (16 consecutive lanes per voxel row), zero-fills the part outside the volume (= grid_sample's
// padding, so the taps need no bounds logic), and then takes all 8 taps of every sample from LDS.
// Sample positions are linear in k (p = P0 + k D), so ONE block reduction of min/max(P0), min/max(D)
// gives every chunk's brick with a dozen fmas -- valid for any set of rays; for scattered rays the
// brick simply does not fit and the chunk falls back to direct global loads (wave-uniform branch).
// =============================================================================================
constexpr int LDS_KC = 8;             // steps per chunk
constexpr int LDS_BRICK_CAP = 12160;  // floats: 47.5 KiB brick + 0.5 KiB header -> 3 workgroups per CU
constexpr int LDS_HDR = 128;          // floats reserved in front of the brick (reduction scratch)

template <bool JAC>
__global__ __launch_bounds__(WG) void k_trilinear_fwd_lds(RenderArgs A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const hdr = lds;
    float* const brick = lds + LDS_HDR;
    int b, r;
    const bool valid = map_ray(A, b, r, threadIdx.x);
    const int tid = threadIdx.x;
    Ray R;
    ray_setup(A, b, r, valid, R);
    const int N = A.sp.n_points;
    const float step = N > 1 ? (A.sp.far_ - A.sp.near_) / (float)(N - 1) : 0.f;
    const KRange K = tri_krange(A, R, false, step);
    const bool live = K.lo <= K.hi;
    const float* __restrict__ vol = A.volume;
    const int D0 = A.D0, D1 = A.D1, D2 = A.D2;

    // linear model of this ray's sample positions in index space: p(k) ~ P0 + k * Dl
    float P0[3], Dl[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        P0[i] = fmaf(A.sp.a[i], fmaf(A.sp.near_, R.d[i], R.s[i]), A.sp.b[i]);
        Dl[i] = step * A.sp.a[i] * R.d[i];
    }
    // block reduction: min/max of P0 and Dl over the live rays, min/max of the k ranges
    float red[14];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        red[i] = live ? P0[i] : INFINITY;
        red[3 + i] = live ? -P0[i] : INFINITY;   // max as min of the negation
        red[6 + i] = live ? Dl[i] : INFINITY;
        red[9 + i] = live ? -Dl[i] : INFINITY;
    }
    red[12] = live ? (float)K.lo : INFINITY;
    red[13] = live ? -(float)K.hi : INFINITY;
#pragma unroll
    for (int v = 0; v < 14; ++v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) red[v] = fminf(red[v], __shfl_xor(red[v], o));
    }
    if ((tid & 63) == 0) {
#pragma unroll
        for (int v = 0; v < 14; ++v) hdr[(tid >> 6) * 16 + v] = red[v];
    }
    __syncthreads();
#pragma unroll
    for (int v = 0; v < 14; ++v) red[v] = fminf(fminf(hdr[v], hdr[16 + v]), fminf(hdr[32 + v], hdr[48 + v]));
    __syncthreads();
    const bool any_live = red[12] < INFINITY;
    const int kbeg = any_live ? (int)red[12] : 1, kend = any_live ? (int)(-red[13]) : 0;

    float S = 0.f;
    float G[3] = {0.f, 0.f, 0.f}, H[3] = {0.f, 0.f, 0.f};
    unsigned cnt = 0;

    for (int k0 = kbeg; k0 <= kend; k0 += LDS_KC) {
        const int k1 = min(k0 + LDS_KC - 1, kend);
        // conservative brick of every tap in steps [k0, k1] (uniform across the workgroup)
        int lo3[3], ex3[3];
        bool fits = true;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float mn = fminf(fmaf((float)k0, red[6 + i], red[i]), fmaf((float)k1, red[6 + i], red[i]));
            const float mx = fmaxf(fmaf((float)k0, -red[9 + i], -red[3 + i]), fmaf((float)k1, -red[9 + i], -red[3 + i]));
            const float flo = floorf(mn - 0.02f), fhi = floorf(mx + 0.02f) + 1.f;
            fits = fits && (fhi - flo) < 4096.f && fabsf(flo) < 1e6f;
            lo3[i] = (int)flo;
            ex3[i] = (int)(fhi - flo) + 1;
        }
        const int ex = ex3[0], ey = ex3[1], ez = ex3[2];
        fits = fits && (long long)ex * ey * ez <= LDS_BRICK_CAP && ez <= 64;
        if (fits) {
            // cooperative load: 16 consecutive lanes per voxel row (contiguous along z), 16 rows per pass
            const int sub = tid & 15;
            const int nrows = ex * ey;
            int row = tid >> 4;
            int rx = row / ey, ry = row - rx * ey;
            for (; row < nrows; row += 16) {
                const int gx = lo3[0] + rx, gy = lo3[1] + ry;
                const bool rin = (unsigned)gx < (unsigned)D0 && (unsigned)gy < (unsigned)D1;
                const float* __restrict__ src = vol + ((size_t)(rin ? gx : 0) * D1 + (rin ? gy : 0)) * D2;
                float* dst = brick + row * ez;
                for (int zi = sub; zi < ez; zi += 16) {
                    const int gz = lo3[2] + zi;
                    dst[zi] = (rin && (unsigned)gz < (unsigned)D2) ? src[gz] : 0.f;
                }
                ry += 16;
                while (ry >= ey) { ry -= ey; ++rx; }
            }
            __syncthreads();
            const int sy = ez, sx = ey * ez;
            for (int k = k0; k <= k1; ++k) {
                if (k < K.lo || k > K.hi) continue;
                const float al = linspace_at(k, N, A.sp.near_, A.sp.far_, step);
                const float px = fmaf(A.sp.a[0], fmaf(al, R.d[0], R.s[0]), A.sp.b[0]);
                const float py = fmaf(A.sp.a[1], fmaf(al, R.d[1], R.s[1]), A.sp.b[1]);
                const float pz = fmaf(A.sp.a[2], fmaf(al, R.d[2], R.s[2]), A.sp.b[2]);
                const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
                const float tx = px - fx, ty = py - fy, tz = pz - fz;
                // local coordinates inside the brick (clamped: the brick is conservative by construction)
                const int lx = min(max((int)fx - lo3[0], 0), ex - 2);
                const int ly = min(max((int)fy - lo3[1], 0), ey - 2);
                const int lz = min(max((int)fz - lo3[2], 0), ez - 2);
                const float* t = brick + lx * sx + ly * sy + lz;
                const float c000 = t[0], c001 = t[1], c010 = t[sy], c011 = t[sy + 1];
                const float c100 = t[sx], c101 = t[sx + 1], c110 = t[sx + sy], c111 = t[sx + sy + 1];
                const float v0 = fmaf(tz, c001 - c000, c000), v1 = fmaf(tz, c011 - c010, c010);
                const float v2 = fmaf(tz, c101 - c100, c100), v3 = fmaf(tz, c111 - c110, c110);
                const float r0 = fmaf(ty, v1 - v0, v0), r1 = fmaf(ty, v3 - v2, v2);
                S += fmaf(tx, r1 - r0, r0);
                ++cnt;
                if (JAC) {
                    const float gx = r1 - r0;
                    const float gy = fmaf(tx, (v3 - v2) - (v1 - v0), v1 - v0);
                    const float d0 = c001 - c000, d1 = c011 - c010, d2 = c101 - c100, d3 = c111 - c110;
                    const float e0 = fmaf(ty, d1 - d0, d0), e1 = fmaf(ty, d3 - d2, d2);
                    const float gz = fmaf(tx, e1 - e0, e0);
                    G[0] += gx; G[1] += gy; G[2] += gz;
                    H[0] = fmaf(al, gx, H[0]); H[1] = fmaf(al, gy, H[1]); H[2] = fmaf(al, gz, H[2]);
                }
            }
            __syncthreads();
        } else {
            // brick too large for LDS (scattered rays / extreme obliquity): direct global taps for this chunk
            for (int k = k0; k <= k1; ++k) {
                if (k < K.lo || k > K.hi) continue;
                const float al = linspace_at(k, N, A.sp.near_, A.sp.far_, step);
                const float px = fmaf(A.sp.a[0], fmaf(al, R.d[0], R.s[0]), A.sp.b[0]);
                const float py = fmaf(A.sp.a[1], fmaf(al, R.d[1], R.s[1]), A.sp.b[1]);
                const float pz = fmaf(A.sp.a[2], fmaf(al, R.d[2], R.s[2]), A.sp.b[2]);
                Taps T;
                make_taps(px, py, pz, D0, D1, D2, T);
                const fpair Q0 = load_pair(vol + T.base[0]);
                const fpair Q1 = load_pair(vol + T.base[1]);
                const fpair Q2 = load_pair(vol + T.base[2]);
                const fpair Q3 = load_pair(vol + T.base[3]);
                const float v0 = fmaf(T.pz1, Q0.y, T.pz0 * Q0.x), v1 = fmaf(T.pz1, Q1.y, T.pz0 * Q1.x);
                const float v2 = fmaf(T.pz1, Q2.y, T.pz0 * Q2.x), v3 = fmaf(T.pz1, Q3.y, T.pz0 * Q3.x);
                const float r0 = fmaf(T.wy1, v1, T.wy0 * v0), r1 = fmaf(T.wy1, v3, T.wy0 * v2);
                S += fmaf(T.wx1, r1, T.wx0 * r0);
                ++cnt;
                if (JAC) {
                    const float d0 = fmaf(T.qz1, Q0.y, T.qz0 * Q0.x), d1 = fmaf(T.qz1, Q1.y, T.qz0 * Q1.x);
                    const float d2 = fmaf(T.qz1, Q2.y, T.qz0 * Q2.x), d3 = fmaf(T.qz1, Q3.y, T.qz0 * Q3.x);
                    const float gz = fmaf(T.wx1, fmaf(T.wy1, d3, T.wy0 * d2), T.wx0 * fmaf(T.wy1, d1, T.wy0 * d0));
                    const float gx = fmaf(T.sx1, r1, T.sx0 * r0);
                    const float gy = fmaf(T.wx1, fmaf(T.sy1, v3, T.sy0 * v2), T.wx0 * fmaf(T.sy1, v1, T.sy0 * v0));
                    G[0] += gx; G[1] += gy; G[2] += gz;
                    H[0] = fmaf(al, gx, H[0]); H[1] = fmaf(al, gy, H[1]); H[2] = fmaf(al, gz, H[2]);
                }
            }
        }
    }

    const float scale = R.L * A.sp.inv_denom;
    if (valid) {
        A.out[(size_t)b * A.n + r] = S * scale;
        if (JAC) {
            float4* jp = reinterpret_cast<float4*>(A.jac + ((size_t)b * A.n + r) * XVR_DRR_JAC_STRIDE);
            jp[0] = make_float4(S * A.sp.inv_denom, scale * A.sp.a[0] * (G[0] - H[0]), scale * A.sp.a[1] * (G[1] - H[1]),
                                scale * A.sp.a[2] * (G[2] - H[2]));
            jp[1] = make_float4(scale * A.sp.a[0] * H[0], scale * A.sp.a[1] * H[1], scale * A.sp.a[2] * H[2], 0.f);
        }
    }
    if (A.work) {
        unsigned tot = wave_sum_u(cnt);
        if ((tid & 63) == 0 && tot) atomicAdd(A.work, (unsigned long long)tot);
    }
}

// =============================================================================================
// trilinear backward by re-marching: pose gradient (GPOSE) and/or voxel gradient (GVOL)
// =============================================================================================
template <bool MASK, bool CLIP, bool GPOSE, bool GVOL>
__global__ __launch_bounds__(WG) void k_trilinear_bwd(RenderArgs A) {
    extern __shared__ float lds[];  // MASK: per-lane upstream gradient per channel [C][WG]
    // fallback role: when a gather launch precedes this one, run only if it declined (rays not a lattice)
    if (A.skip_unless_flag_gt && !(*A.skip_unless_flag_gt > __float_as_uint(GATHER_DEV_TOL))) return;
    int b, r;
    const bool valid = map_ray(A, b, r, threadIdx.x);
    const int tid = threadIdx.x;
    Ray R;
    ray_setup(A, b, r, valid, R);
    const int N = A.sp.n_points;
    const float step = N > 1 ? (A.sp.far_ - A.sp.near_) / (float)(N - 1) : 0.f;
    const KRange K = tri_krange(A, R, CLIP, step);
    const int kbeg = __builtin_amdgcn_readfirstlane(wave_min_i(K.lo));
    const int kend = __builtin_amdgcn_readfirstlane(wave_max_i(K.hi));
    const float span = fmaxf(R.amax - R.amin, 0.f);
    const float* __restrict__ vol = A.volume;
    const int D0 = A.D0, D1 = A.D1, D2 = A.D2;
    const float base_scale = R.L * A.sp.inv_denom;
    const float scale = CLIP ? base_scale * span : base_scale;

    float g0 = 0.f;
    if (MASK) {
        for (int c = 0; c < A.C; ++c) lds[c * WG + tid] = valid ? A.gout[((size_t)b * A.C + c) * A.n + r] : 0.f;
    } else if (valid) {
        g0 = A.gout[(size_t)b * A.n + r];
    }
    float SV = 0.f;
    float G[3] = {0.f, 0.f, 0.f}, H[3] = {0.f, 0.f, 0.f};
    float E0 = 0.f, E1 = 0.f;
    const float adx = A.sp.a[0] * R.d[0], ady = A.sp.a[1] * R.d[1], adz = A.sp.a[2] * R.d[2];

    for (int k = kbeg; k <= kend; ++k) {
        if (k < K.lo || k > K.hi) continue;
        const float u = linspace_at(k, N, A.sp.near_, A.sp.far_, step);
        const float al = CLIP ? fmaf(u, R.amax - R.amin, R.amin) : u;
        const float px = fmaf(A.sp.a[0], fmaf(al, R.d[0], R.s[0]), A.sp.b[0]);
        const float py = fmaf(A.sp.a[1], fmaf(al, R.d[1], R.s[1]), A.sp.b[1]);
        const float pz = fmaf(A.sp.a[2], fmaf(al, R.d[2], R.s[2]), A.sp.b[2]);
        Taps T;
        make_taps(px, py, pz, D0, D1, D2, T);
        float gk = g0;
        if (MASK) gk = lds[nearest_label(A.mask, px, py, pz, D0, D1, D2, A.C) * WG + tid];
        if (GVOL) {
            const float c = gk * scale;
            if (c != 0.f) {
                const float w00 = c * T.wx0 * T.wy0, w01 = c * T.wx0 * T.wy1;
                const float w10 = c * T.wx1 * T.wy0, w11 = c * T.wx1 * T.wy1;
                const float w[4] = {w00, w01, w10, w11};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float a0 = w[q] * T.pz0, a1 = w[q] * T.pz1;
                    if (a0 != 0.f) atomic_add_f32(A.gvol + T.base[q], a0);
                    if (a1 != 0.f) atomic_add_f32(A.gvol + T.base[q] + 1, a1);
                }
            }
        }
        if (GPOSE) {
            const fpair P0 = load_pair(vol + T.base[0]);
            const fpair P1 = load_pair(vol + T.base[1]);
            const fpair P2 = load_pair(vol + T.base[2]);
            const fpair P3 = load_pair(vol + T.base[3]);
            const float v0 = fmaf(T.pz1, P0.y, T.pz0 * P0.x), v1 = fmaf(T.pz1, P1.y, T.pz0 * P1.x);
            const float v2 = fmaf(T.pz1, P2.y, T.pz0 * P2.x), v3 = fmaf(T.pz1, P3.y, T.pz0 * P3.x);
            const float r0 = fmaf(T.wy1, v1, T.wy0 * v0), r1 = fmaf(T.wy1, v3, T.wy0 * v2);
            const float v = fmaf(T.wx1, r1, T.wx0 * r0);
            const float d0 = fmaf(T.qz1, P0.y, T.qz0 * P0.x), d1 = fmaf(T.qz1, P1.y, T.qz0 * P1.x);
            const float d2 = fmaf(T.qz1, P2.y, T.qz0 * P2.x), d3 = fmaf(T.qz1, P3.y, T.qz0 * P3.x);
            const float gz = gk * fmaf(T.wx1, fmaf(T.wy1, d3, T.wy0 * d2), T.wx0 * fmaf(T.wy1, d1, T.wy0 * d0));
            const float gx = gk * fmaf(T.sx1, r1, T.sx0 * r0);
            const float gy = gk * fmaf(T.wx1, fmaf(T.sy1, v3, T.sy0 * v2), T.wx0 * fmaf(T.sy1, v1, T.sy0 * v0));
            SV = fmaf(gk, v, SV);
            G[0] += gx; G[1] += gy; G[2] += gz;
            H[0] = fmaf(al, gx, H[0]); H[1] = fmaf(al, gy, H[1]); H[2] = fmaf(al, gz, H[2]);
            if (CLIP) {
                const float gd = fmaf(gx, adx, fmaf(gy, ady, gz * adz));
                E0 = fmaf(gd, 1.f - u, E0);
                E1 = fmaf(gd, u, E1);
            }
        }
    }

    if (GPOSE) {
        float js[3], jt[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            jt[i] = scale * A.sp.a[i] * H[i];
            js[i] = scale * A.sp.a[i] * (G[i] - H[i]);
        }
        if (CLIP) {
            const float dmin = base_scale * (-SV + span * E0), dmax = base_scale * (SV + span * E1);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                if (R.ax_in == i && span > 0.f) {
                    js[i] += dmin * (R.amin - 1.f) / R.d[i];
                    jt[i] += dmin * (-R.amin) / R.d[i];
                }
                if (R.ax_out == i && span > 0.f) {
                    js[i] += dmax * (R.amax - 1.f) / R.d[i];
                    jt[i] += dmax * (-R.amax) / R.d[i];
                }
            }
        }
        if (!valid) js[0] = js[1] = js[2] = 0.f;
        if (valid) {
            float* tp = A.gtgt + ((size_t)b * A.n + r) * 3;
            tp[0] = jt[0]; tp[1] = jt[1]; tp[2] = jt[2];
            if (A.glen) A.glen[(size_t)b * A.n + r] = SV * (CLIP ? A.sp.inv_denom * span : A.sp.inv_denom);
        }
        // grad_source is shared by all rays of the pose: wave butterfly, then one atomic per wave
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float tot = wave_sum_f(js[i]);
            if ((tid & 63) == 0 && tot != 0.f) atomic_add_f32(A.gsrc + 3 * b + i, tot);
        }
    }
}


// =============================================================================================
// Voxel gradient of the trilinear renderer WITHOUT atomics: a voxel-driven exact adjoint.
//
// fp32 atomics are the wrong tool on this chip (measured, profiles/r01_microbench_atomics.txt:
// ~20 G scattered global atomic line-ops/s whatever the scope, ~190 G/s for LDS ds_add_f32
// chip-wide), and the scatter has 8 of them per sample.  Instead one thread OWNS one voxel v and
// gathers every sample that touches it.  Because a pose's rays end on a planar H x W lattice and all
// rays share alpha_k, the samples of step k form a planar patch, so the few (pixel, step) pairs whose
// sample lies inside v's unit box are found by projecting v onto the detector:
//     alpha_v = n.(x_v - s)/h,  pixel (i*, j*) = G.(s + (x_v - s)/alpha_k - T00)
// with a conservative window around (k*, i*, j*).  Each candidate's sample position is recomputed
// with the SAME fmaf sequence as the forward from the SAME target array, so its weight
// prod(1 - |p - v|) is bit-identical to the forward's interpolation weight: this is the exact
// transpose of the forward gather, up to summation order -- and it is deterministic.
// =============================================================================================

struct PoseLattice {  // per pose, 32 floats
    float s[3], dalpha;    // source; half-range of alpha over a voxel's unit box
    float nh[3], nh_norm;  // alpha_v = nh . (x_v - s)
    float gc[3], gc0;      // column  j* = gc0 + (gc . w) / alpha
    float gr[3], gr0;      // row     i* = gr0 + (gr . w) / alpha
    float hwc, hwr, gc_norm, gr_norm;  // window half-widths (pixels) at alpha = 1; |gc|, |gr|
    float st[3], pad0;     // T00 + eps - s
    float ec[3], pad1;     // lattice step per column
    float er[3], pad2;     // lattice step per row
};

struct GatherArgs {
    const float* __restrict__ source;
    const float* __restrict__ target;
    const float* __restrict__ raylen;
    const float* __restrict__ gout;
    int B, n, W, H;
    int D0, D1, D2;
    xvr_drr_spec sp;
    unsigned* flag;          // max lattice deviation (float bits), written by k_gather_prep
    PoseLattice* poses;
    float4* q;               // trilinear: [B][n] = ((target - source) + eps, gout * raylen * inv_denom)
                             // siddon:    [B][n] = (1 / ((target - source) + eps), gout * raylen)
    float2* q2;              // siddon: [B][n] = (alpha_lo, alpha_hi) of the ray, as the forward clamps them
    int siddon;
    int V;                   // voxels per lane and axis in the gather (1 or 2)
    int bd[3];               // voxels per gather workgroup (brick) along x, y, z
    unsigned* cull;          // [bricks][words] bit p set = pose p can touch the brick
    int words;               // ceil(B / 32)
    float* gvol;
};

__device__ __forceinline__ void cross3(const float* a, const float* b, float* o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ float dot3(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// one block per (pose, 256 rays): packs q, measures how far the targets are from an exact lattice,
// and (block 0 of each pose) derives the pose's projection constants.
__global__ __launch_bounds__(WG) void k_gather_prep(GatherArgs G) {
    const int b = blockIdx.y;
    const int r = blockIdx.x * WG + threadIdx.x;
    const float* T = G.target + (size_t)b * G.n * 3;
    float t00[3], ec[3], er[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        t00[i] = T[i];
        ec[i] = (T[(size_t)(G.W - 1) * 3 + i] - t00[i]) / (float)(G.W - 1);
        er[i] = (T[(size_t)(G.H - 1) * G.W * 3 + i] - t00[i]) / (float)(G.H - 1);
    }
    const float pitch = fminf(sqrtf(dot3(ec, ec)), sqrtf(dot3(er, er)));
    float dev = 0.f;
    if (r < G.n) {
        const int i = r / G.W, j = r - i * G.W;
        const float tx = T[(size_t)r * 3], ty = T[(size_t)r * 3 + 1], tz = T[(size_t)r * 3 + 2];
        dev = fmaxf(fabsf(tx - (t00[0] + j * ec[0] + i * er[0])),
                    fmaxf(fabsf(ty - (t00[1] + j * ec[1] + i * er[1])), fabsf(tz - (t00[2] + j * ec[2] + i * er[2]))));
        dev = pitch > 0.f ? dev / pitch : INFINITY;
        if (!(dev == dev)) dev = INFINITY;
        const float c = G.gout[(size_t)b * G.n + r] * G.raylen[(size_t)b * G.n + r] * G.sp.inv_denom;
        // d exactly as the forward forms it, (t - s) + eps, so that the gather's fmaf chain below
        // reproduces the forward's sample positions bit for bit
        const float sx = G.source[3 * b], sy = G.source[3 * b + 1], sz = G.source[3 * b + 2];
        const float ddx = (tx - sx) + G.sp.eps, ddy = (ty - sy) + G.sp.eps, ddz = (tz - sz) + G.sp.eps;
        if (!G.siddon) {
            G.q[(size_t)b * G.n + r] = make_float4(ddx, ddy, ddz, c);
        } else {
            // the ray's own integration interval, computed exactly as ray_setup() does for the forward
            const float dd[3] = {ddx, ddy, ddz}, ss[3] = {sx, sy, sz};
            float lo = -INFINITY, hi = INFINITY;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float a0 = (G.sp.lo[k] - ss[k]) / dd[k], a1 = (G.sp.hi[k] - ss[k]) / dd[k];
                lo = fmaxf(lo, fminf(a0, a1));
                hi = fminf(hi, fmaxf(a0, a1));
            }
            if (!(lo > 0.f)) lo = 0.f;
            if (!(hi < 1.f)) hi = 1.f;
            G.q[(size_t)b * G.n + r] = make_float4(1.f / ddx, 1.f / ddy, 1.f / ddz,
                                                   G.gout[(size_t)b * G.n + r] * G.raylen[(size_t)b * G.n + r]);
            G.q2[(size_t)b * G.n + r] = make_float2(lo, hi);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dev = fmaxf(dev, __shfl_xor(dev, o));
    // only a wave that SEES a violation touches the flag (one word: 10^5 same-address atomics would
    // serialise into more than a millisecond)
    if ((threadIdx.x & 63) == 0 && dev > GATHER_DEV_TOL) atomicMax(G.flag, __float_as_uint(dev));
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        PoseLattice P = {};
        float s[3], nrm[3], st[3], ts[3], tmp[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            s[i] = G.source[3 * b + i];
            ts[i] = (t00[i] + G.sp.eps) - s[i];  // T00e - s
            st[i] = -ts[i];
        }
        cross3(ec, er, nrm);
        const float h = dot3(nrm, ts);  // n . (T00 - s)
        cross3(er, nrm, tmp);
        const float dc = dot3(ec, tmp);
        float gc[3] = {tmp[0] / dc, tmp[1] / dc, tmp[2] / dc};
        cross3(nrm, ec, tmp);
        const float dr = dot3(er, tmp);
        float gr[3] = {tmp[0] / dr, tmp[1] / dr, tmp[2] / dr};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            P.s[i] = s[i];
            P.nh[i] = nrm[i] / h;
            P.gc[i] = gc[i];
            P.gr[i] = gr[i];
            P.st[i] = ts[i];
            P.ec[i] = ec[i];
            P.er[i] = er[i];
            P.dalpha += fabsf(P.nh[i]) / G.sp.a[i];
            P.hwc += fabsf(gc[i]) / G.sp.a[i];
            P.hwr += fabsf(gr[i]) / G.sp.a[i];
        }
        P.nh_norm = sqrtf(dot3(P.nh, P.nh));
        P.gc_norm = sqrtf(dot3(gc, gc));
        P.gr_norm = sqrtf(dot3(gr, gr));
        P.gc0 = dot3(gc, st);
        P.gr0 = dot3(gr, st);
        const float chk = P.dalpha + P.hwc + P.hwr + P.gc0 + P.gr0;
        if (!(chk == chk) || !(fabsf(chk) < 1e30f) || h == 0.f) atomicMax(G.flag, __float_as_uint(INFINITY));
        G.poses[b] = P;
    }
}

// A gather workgroup covers a brick of bd[0] x bd[1] x bd[2] voxels (trilinear: one wavefront per
// compact (4V)^3 brick, each lane a V^3 block; siddon: 256 lanes on 4 x 8 x 8 voxels).
__device__ __forceinline__ void brick_coords(int blk, int D1, int D2, const int* bd, int& bx, int& by, int& bz) {
    const int nz = (D2 + bd[2] - 1) / bd[2], ny = (D1 + bd[1] - 1) / bd[1];
    bz = blk % nz; blk /= nz;
    by = blk % ny; bx = blk / ny;
}

// one thread per (brick, pose): can any sample of the pose fall inside the brick grown by one voxel?
// (bounding sphere against the pose's sample pyramid, conservative).  32 poses per word.
__global__ __launch_bounds__(WG) void k_gather_cull(GatherArgs G, int nbricks) {
    const int brick = blockIdx.x * (WG / 32) + (threadIdx.x >> 5);
    const int lane = threadIdx.x & 31;
    if (brick >= nbricks) return;
    int bx, by, bz;
    brick_coords(brick, G.D1, G.D2, G.bd, bx, by, bz);
    const float c[3] = {bx * G.bd[0] + 0.5f * (G.bd[0] - 1), by * G.bd[1] + 0.5f * (G.bd[1] - 1),
                        bz * G.bd[2] + 0.5f * (G.bd[2] - 1)};
    // half extent to the outermost voxel centre + 1 (interpolation support) + 0.5 (slack), in x units
    const float hx = (0.5f * (G.bd[0] - 1) + 1.5f) / G.sp.a[0], hy = (0.5f * (G.bd[1] - 1) + 1.5f) / G.sp.a[1],
                hz = (0.5f * (G.bd[2] - 1) + 1.5f) / G.sp.a[2];
    const float R = sqrtf(hx * hx + hy * hy + hz * hz);
    for (int wd = 0; wd < G.words; ++wd) {
        const int p = wd * 32 + lane;
        bool hit = false;
        if (p < G.B) {
            const PoseLattice& P = G.poses[p];
            float w[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) w[i] = (c[i] - G.sp.b[i]) / G.sp.a[i] - P.s[i];
            const float av = dot3(P.nh, w), da = R * P.nh_norm;
            const float amin = av - da, amax = av + da;
            if (amax >= G.sp.near_ && amin <= G.sp.far_) {
                if (amin <= 1e-6f) {
                    hit = true;  // the sphere reaches the source plane: no perspective bound, keep
                } else {
                    const float inv = 1.f / av;
                    const float jc = fmaf(dot3(P.gc, w), inv, P.gc0), ic = fmaf(dot3(P.gr, w), inv, P.gr0);
                    // a point of the sphere moves the pixel by at most R |g| / alpha (lateral) plus the
                    // centre's own shift |j - gc0| * da / alpha (depth), with alpha >= amin
                    const float ia = 1.f / amin;
                    const float rj = (R * P.gc_norm + fabsf(jc - P.gc0) * da) * ia + 1.f;
                    const float ri = (R * P.gr_norm + fabsf(ic - P.gr0) * da) * ia + 1.f;
                    hit = jc + rj >= 0.f && jc - rj <= (float)(G.W - 1) && ic + ri >= 0.f && ic - ri <= (float)(G.H - 1);
                }
            }
        }
        const unsigned long long m = __ballot(hit);
        const unsigned bits = (threadIdx.x & 32) ? (unsigned)(m >> 32) : (unsigned)m;
        if (lane == 0) G.cull[(size_t)brick * G.words + wd] = bits;
    }
}

// One lane owns a V x V x V block of voxels (V = 2: per-pose / per-step / per-row setup is paid once
// for 8 voxels and the sample position is computed once per candidate); a workgroup covers a
// (4V) x (8V) x (8V) brick so that its lanes' candidates share pixels.
// max(1 - |d|, 0), the trilinear weight of a voxel at signed distance d, in ONE instruction: 1 - |d| never
// exceeds 1, so the [0, 1] clamp equals the max and folds into the subtraction's clamp bit
// (v_sub_f32 dst, 1.0, |d| clamp) -- the gather's inner loop is VALU-bound and has six of these per candidate.
__device__ __forceinline__ float hat01(float d) { return __builtin_amdgcn_fmed3f(1.f - fabsf(d), 0.f, 1.f); }

// (register budget set for 7 wavefronts per SIMD: 72 VGPRs, no spills, 14.4 ms at C2 against 14.6 at the 6 the
//  compiler chose; 8 spills, 4 takes 17.7 ms)
template <int V, bool NOLOAD = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(7, 7))) void k_trilinear_gather_vol(GatherArgs G) {
    if (*G.flag > __float_as_uint(GATHER_DEV_TOL)) return;  // not a lattice: the scatter kernel runs instead
    constexpr float HS = V == 2 ? 1.5f : 1.0f;  // half-size of the block's interpolation support
    constexpr float CO = V == 2 ? 0.5f : 0.0f;  // block centre relative to its first voxel
    int bx, by, bz;
    brick_coords(blockIdx.x, G.D1, G.D2, G.bd, bx, by, bz);
    const int tid = threadIdx.x;  // one wavefront: 4 x 4 x 4 blocks
    const int vx = (bx * 4 + (tid >> 4)) * V, vy = (by * 4 + ((tid >> 2) & 3)) * V, vz = (bz * 4 + (tid & 3)) * V;
    const bool inb = vx < G.D0 && vy < G.D1 && vz < G.D2;
    const float fv[3] = {(float)vx, (float)vy, (float)vz};
    float xv[3];  // block centre in x coordinates
#pragma unroll
    for (int i = 0; i < 3; ++i) xv[i] = (fv[i] + CO - G.sp.b[i]) / G.sp.a[i];
    const int N = G.sp.n_points;
    const float near_ = G.sp.near_, far_ = G.sp.far_;
    const float step = N > 1 ? (far_ - near_) / (float)(N - 1) : 0.f;
    const float inv_step = step > 0.f ? 1.f / step : 0.f;
    const float a0 = G.sp.a[0], a1 = G.sp.a[1], a2 = G.sp.a[2];
    const float b0 = G.sp.b[0], b1 = G.sp.b[1], b2 = G.sp.b[2];
    // p - v for the first voxel of the block (b - v is exact, and so is folding it into the fmaf for
    // every |p| < 2^23: these are the forward's interpolation weights); the second voxel gets its own
    // constant so that its weight is formed by the same single fmaf
    const float bv0 = b0 - fv[0], bv1 = b1 - fv[1], bv2 = b2 - fv[2];
    const float bw0 = bv0 - 1.f, bw1 = bv1 - 1.f, bw2 = bv2 - 1.f;
    const float jmargin = GATHER_DEV_TOL + 0.01f;
    float acc[V * V * V];
#pragma unroll
    for (int i = 0; i < V * V * V; ++i) acc[i] = 0.f;

    for (int wd = 0; wd < G.words; ++wd) {
        unsigned bits = G.cull[(size_t)blockIdx.x * G.words + wd];  // uniform: scalar load
        while (bits) {
            const int p = wd * 32 + __builtin_ctz(bits);
            bits &= bits - 1;
            const PoseLattice& P = G.poses[p];
            const float s0 = P.s[0], s1 = P.s[1], s2 = P.s[2];
            const float w0 = xv[0] - s0, w1 = xv[1] - s1, w2 = xv[2] - s2;
            const float av = P.nh[0] * w0 + P.nh[1] * w1 + P.nh[2] * w2;
            const float da = HS * P.dalpha;
            int klo, khi;
            if (step > 0.f) {
                const float k0 = (av - da - near_) * inv_step, k1 = (av + da - near_) * inv_step;
                klo = (int)ceilf(fmaxf(k0 - GATHER_K_SLACK, 0.f));
                khi = (int)floorf(fminf(k1 + GATHER_K_SLACK, (float)(N - 1)));
            } else {
                klo = 0;
                khi = (fabsf(av - near_) <= da) ? 0 : -1;
            }
            if (!inb || !(av == av)) khi = -1;
            const float grw = P.gr[0] * w0 + P.gr[1] * w1 + P.gr[2] * w2;
            const float4* __restrict__ q = G.q + (size_t)p * G.n;
            const float Bx = fmaf(a0, s0, bv0), By = fmaf(a1, s1, bv1), Bz = fmaf(a2, s2, bv2);
            for (int k = klo; k <= khi; ++k) {
                const float al = linspace_at(k, N, near_, far_, step);
                if (al > 1e-12f) {
                    const float inv = 1.f / al;
                    const float ic = fmaf(grw, inv, P.gr0);
                    const float hi = fmaf(HS * P.hwr, inv, GATHER_WIN_MARGIN);
                    const int ilo = (int)ceilf(fmaxf(ic - hi, 0.f));
                    const int ihi = (int)floorf(fminf(ic + hi, (float)(G.H - 1)));
                    // lattice model of the sample positions relative to the block centre, in index space:
                    // Q0 + i Ur + j Uc.  Used ONLY to find which pixels to visit; the weights below come
                    // from the real targets.
                    const float ucx = al * a0 * P.ec[0], ucy = al * a1 * P.ec[1], ucz = al * a2 * P.ec[2];
                    const float urx = al * a0 * P.er[0], ury = al * a1 * P.er[1], urz = al * a2 * P.er[2];
                    const float q0x = fmaf(a0, fmaf(al, P.st[0], s0), b0) - (fv[0] + CO);
                    const float q0y = fmaf(a1, fmaf(al, P.st[1], s1), b1) - (fv[1] + CO);
                    const float q0z = fmaf(a2, fmaf(al, P.st[2], s2), b2) - (fv[2] + CO);
                    // reciprocal of the per-column step, clamped: an axis the row does not move along
                    // (|uc| ~ 0) then yields (-huge, +huge) when |q| < HS and an empty interval otherwise
                    const float rx = fabsf(ucx) < 1e-9f ? 1e9f : 1.f / ucx;
                    const float ry = fabsf(ucy) < 1e-9f ? 1e9f : 1.f / ucy;
                    const float rz = fabsf(ucz) < 1e-9f ? 1e9f : 1.f / ucz;
                    const float ax_ = HS * fabsf(rx), ay_ = HS * fabsf(ry), az_ = HS * fabsf(rz);
                    const float Ax = al * a0, Ay = al * a1, Az = al * a2;
                    for (int i = ilo; i <= ihi; ++i) {
                        const float fi = (float)i;
                        const float qx = fmaf(fi, urx, q0x), qy = fmaf(fi, ury, q0y), qz = fmaf(fi, urz, q0z);
                        // exact j-interval on this row where |q + j Uc| < HS on all three axes
                        const float mx = -qx * rx, my = -qy * ry, mz = -qz * rz;
                        const float lo = fmaxf(fmaxf(mx - ax_, my - ay_), mz - az_);
                        const float hiJ = fminf(fminf(mx + ax_, my + ay_), mz + az_);
                        const int jlo = (int)ceilf(fmaxf(lo - jmargin, 0.f));
                        const int jhi = (int)floorf(fminf(hiJ + jmargin, (float)(G.W - 1)));
                        const float4* __restrict__ row = q + (size_t)i * G.W;
                        // two candidates per trip: both 16-byte loads are issued before either is used
                        for (int j = jlo; j <= jhi; j += 2) {
                            const bool two = j < jhi;
                            float4 ta, tb;
                            if (NOLOAD) {  // ablation only (XVR_DRR_GATHER_ABLATE=1): same arithmetic, no memory
                                ta = make_float4(q0x + (float)j, q0y, q0z, 1.f);
                                tb = make_float4(q0x, q0y + (float)j, q0z, 1.f);
                            } else {
                                ta = row[j];
                                tb = row[two ? j + 1 : j];
                            }
                            tb.w = two ? tb.w : 0.f;
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                const float4 t = h ? tb : ta;
                                // signed distance of the sample from the block's first voxel, per axis:
                                // a (s + alpha d) + b - v folded into one fma (within an ulp of the
                                // forward's two-fma chain); the second voxel sits exactly 1 further
                                const float dx = fmaf(Ax, t.x, Bx), dy = fmaf(Ay, t.y, By), dz = fmaf(Az, t.z, Bz);
                                const float ux0 = hat01(dx);
                                const float uy0 = hat01(dy);
                                const float uz0 = hat01(dz) * t.w;
                                if (V == 1) {
                                    acc[0] = fmaf(ux0 * uy0, uz0, acc[0]);
                                } else {
                                    const float ux1 = hat01(dx - 1.f);
                                    const float uy1 = hat01(dy - 1.f);
                                    const float uz1 = hat01(dz - 1.f) * t.w;
                                    // (packed v_pk_mul/fma_f32 on (z, z+1) pairs measured SLOWER, 15.9 vs 14.7 ms:
                                    //  the pair building / broadcast moves cost more than the halved fma count)
                                    const float p00 = ux0 * uy0, p01 = ux0 * uy1, p10 = ux1 * uy0, p11 = ux1 * uy1;
                                    acc[0] = fmaf(p00, uz0, acc[0]);
                                    acc[1 % (V * V * V)] = fmaf(p00, uz1, acc[1 % (V * V * V)]);
                                    acc[2 % (V * V * V)] = fmaf(p01, uz0, acc[2 % (V * V * V)]);
                                    acc[3 % (V * V * V)] = fmaf(p01, uz1, acc[3 % (V * V * V)]);
                                    acc[4 % (V * V * V)] = fmaf(p10, uz0, acc[4 % (V * V * V)]);
                                    acc[5 % (V * V * V)] = fmaf(p10, uz1, acc[5 % (V * V * V)]);
                                    acc[6 % (V * V * V)] = fmaf(p11, uz0, acc[6 % (V * V * V)]);
                                    acc[7 % (V * V * V)] = fmaf(p11, uz1, acc[7 % (V * V * V)]);
                                }
                            }
                        }
                    }
                } else {
                    // alpha_k = 0: every ray's sample sits on the source; all pixels are candidates for the
                    // blocks whose support contains it (a source inside the volume only)
                    const bool hit = fabsf(fmaf(a0, s0, b0) - (fv[0] + CO)) < HS && fabsf(fmaf(a1, s1, b1) - (fv[1] + CO)) < HS &&
                                     fabsf(fmaf(a2, s2, b2) - (fv[2] + CO)) < HS;
                    const int cnt = hit ? G.n : 0;
                    for (int r = 0; r < cnt; ++r) {
                        const float4 t = q[r];
                        const float ix = fmaf(al, t.x, s0), iy = fmaf(al, t.y, s1), iz = fmaf(al, t.z, s2);
#pragma unroll
                        for (int e = 0; e < V * V * V; ++e) {
                            const float ox = (float)(e >> 2 & 1), oy = (float)(e >> 1 & 1), oz = (float)(e & 1);
                            const float ux = hat01(fmaf(a0, ix, bv0 - ox));
                            const float uy = hat01(fmaf(a1, iy, bv1 - oy));
                            const float uz = hat01(fmaf(a2, iz, bv2 - oz));
                            acc[e] = fmaf(ux * uy * uz, t.w, acc[e]);
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < V * V * V; ++e) {
        const int x = vx + (V == 2 ? (e >> 2 & 1) : 0), y = vy + (V == 2 ? (e >> 1 & 1) : 0), z = vz + (V == 2 ? (e & 1) : 0);
        if (x < G.D0 && y < G.D1 && z < G.D2 && acc[e] != 0.f) G.gvol[((size_t)x * G.D1 + y) * G.D2 + z] += acc[e];
    }
}

// Siddon voxel gradient as a gather (exact-geometry index map only: a = 1, b = shift - 1/2, so the
// voxel a segment is credited to is the voxel whose box contains it).  d out / d V[v] for one ray is
// L x (length of the ray inside v's box, clipped to the ray's own [alpha_lo, alpha_hi]); the box's
// entry/exit alphas use the forward's expression ((plane + plane0) - s) * (1 / d), so they are the
// very crossing values the forward traversal produced.
__global__ __launch_bounds__(WG) void k_siddon_gather_vol(GatherArgs G) {
    if (*G.flag > __float_as_uint(GATHER_DEV_TOL)) return;
    int bx, by, bz;
    brick_coords(blockIdx.x, G.D1, G.D2, G.bd, bx, by, bz);
    // 256 lanes on 4 x 8 x 8 voxels (one voxel per lane: the per-pose setup dominates, so a larger
    // workgroup that amortises the cull words and pose constants wins here -- 4^3 bricks measured 10 % slower)
    const int tid = threadIdx.x;
    const int vx = bx * 4 + (tid >> 6), vy = by * 8 + ((tid >> 3) & 7), vz = bz * 8 + (tid & 7);
    const bool inb = vx < G.D0 && vy < G.D1 && vz < G.D2;
    // planes of the voxel's box and its centre, in x coordinates
    const float p0x = (float)vx + G.sp.plane0[0], p0y = (float)vy + G.sp.plane0[1], p0z = (float)vz + G.sp.plane0[2];
    const float p1x = (float)(vx + 1) + G.sp.plane0[0], p1y = (float)(vy + 1) + G.sp.plane0[1],
                p1z = (float)(vz + 1) + G.sp.plane0[2];
    const float cx = p0x + 0.5f, cy = p0y + 0.5f, cz = p0z + 0.5f;
    float acc = 0.f;
    for (int wd = 0; wd < G.words; ++wd) {
        unsigned bits = G.cull[(size_t)blockIdx.x * G.words + wd];
        while (bits) {
            const int p = wd * 32 + __builtin_ctz(bits);
            bits &= bits - 1;
            const PoseLattice& P = G.poses[p];
            const float s0 = P.s[0], s1 = P.s[1], s2 = P.s[2];
            const float w0 = cx - s0, w1 = cy - s1, w2 = cz - s2;
            const float av = P.nh[0] * w0 + P.nh[1] * w1 + P.nh[2] * w2;
            const float da = 0.5f * P.dalpha;
            const float amin = av - da, amax = av + da;
            // pixel = g0 + N / alpha with N in [N0 - dN, N0 + dN], alpha in [amin, amax]
            const float nj = P.gc[0] * w0 + P.gc[1] * w1 + P.gc[2] * w2, dnj = 0.5f * P.hwc;
            const float ni = P.gr[0] * w0 + P.gr[1] * w1 + P.gr[2] * w2, dni = 0.5f * P.hwr;
            int jlo = 0, jhi = -1, ilo = 0, ihi = -1;
            if (inb && amin > 1e-6f && amax >= 0.f && amin <= 1.f) {
                const float i0 = 1.f / amin, i1 = 1.f / amax;
                const float ja = (nj - dnj) * i0, jb = (nj - dnj) * i1, jc = (nj + dnj) * i0, jd = (nj + dnj) * i1;
                const float ia = (ni - dni) * i0, ib = (ni - dni) * i1, ic = (ni + dni) * i0, id = (ni + dni) * i1;
                const float jmn = fminf(fminf(ja, jb), fminf(jc, jd)) + P.gc0 - GATHER_WIN_MARGIN;
                const float jmx = fmaxf(fmaxf(ja, jb), fmaxf(jc, jd)) + P.gc0 + GATHER_WIN_MARGIN;
                const float imn = fminf(fminf(ia, ib), fminf(ic, id)) + P.gr0 - GATHER_WIN_MARGIN;
                const float imx = fmaxf(fmaxf(ia, ib), fmaxf(ic, id)) + P.gr0 + GATHER_WIN_MARGIN;
                jlo = (int)ceilf(fmaxf(jmn, 0.f));
                jhi = (int)floorf(fminf(jmx, (float)(G.W - 1)));
                ilo = (int)ceilf(fmaxf(imn, 0.f));
                ihi = (int)floorf(fminf(imx, (float)(G.H - 1)));
            } else if (inb && amin <= 1e-6f && amax >= 0.f) {
                // the box reaches the source plane: no perspective bound -- visit every ray
                jhi = G.W - 1;
                ihi = G.H - 1;
            }
            const float lx = p0x - s0, ly = p0y - s1, lz = p0z - s2;
            const float hx = p1x - s0, hy = p1y - s1, hz = p1z - s2;
            const float4* __restrict__ q = G.q + (size_t)p * G.n;
            const float2* __restrict__ q2 = G.q2 + (size_t)p * G.n;
            for (int i = ilo; i <= ihi; ++i) {
                for (int j = jlo; j <= jhi; ++j) {
                    const float4 t = q[(size_t)i * G.W + j];
                    const float2 ab = q2[(size_t)i * G.W + j];
                    const float x0 = lx * t.x, x1 = hx * t.x, y0 = ly * t.y, y1 = hy * t.y, z0 = lz * t.z, z1 = hz * t.z;
                    float en = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
                    float ex = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
                    en = fmaxf(en, ab.x);
                    ex = fminf(ex, ab.y);
                    acc = fmaf(fmaxf(ex - en, 0.f), t.w, acc);
                }
            }
        }
    }
    if (inb && acc != 0.f) G.gvol[((size_t)vx * G.D1 + vy) * G.D2 + vz] += acc;
}


// Same gather with a 2 x 2 x 2 voxel block per lane (one wavefront per 8^3 brick, as the trilinear gather):
// the per-pose window and the candidate's loads are paid once for eight voxels, the three planes per axis give
// nine crossing alphas per candidate (the forward's expression, plane by plane), from which every voxel's
// entry / exit are one max3 / min3.  A candidate costs ~57 VALU for 8 voxels instead of 8 x 19.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(6, 6))) void k_siddon_gather_vol2(GatherArgs G) {
    if (*G.flag > __float_as_uint(GATHER_DEV_TOL)) return;
    int bx, by, bz;
    brick_coords(blockIdx.x, G.D1, G.D2, G.bd, bx, by, bz);
    const int tid = threadIdx.x;
    const int vx = (bx * 4 + (tid >> 4)) * 2, vy = (by * 4 + ((tid >> 2) & 3)) * 2, vz = (bz * 4 + (tid & 3)) * 2;
    const bool inb = vx < G.D0 && vy < G.D1 && vz < G.D2;
    // the three planes per axis that bound the block's voxels, and the block centre, in x coordinates
    float px[3], py[3], pz[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        px[k] = (float)(vx + k) + G.sp.plane0[0];
        py[k] = (float)(vy + k) + G.sp.plane0[1];
        pz[k] = (float)(vz + k) + G.sp.plane0[2];
    }
    const float cx = px[1], cy = py[1], cz = pz[1];
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int wd = 0; wd < G.words; ++wd) {
        unsigned bits = G.cull[(size_t)blockIdx.x * G.words + wd];
        while (bits) {
            const int p = wd * 32 + __builtin_ctz(bits);
            bits &= bits - 1;
            const PoseLattice& P = G.poses[p];
            const float s0 = P.s[0], s1 = P.s[1], s2 = P.s[2];
            const float w0 = cx - s0, w1 = cy - s1, w2 = cz - s2;
            const float av = P.nh[0] * w0 + P.nh[1] * w1 + P.nh[2] * w2;
            const float da = P.dalpha;                        // half-range of alpha over the 2-voxel block
            const float amin = av - da, amax = av + da;
            const float nj = P.gc[0] * w0 + P.gc[1] * w1 + P.gc[2] * w2, dnj = P.hwc;
            const float ni = P.gr[0] * w0 + P.gr[1] * w1 + P.gr[2] * w2, dni = P.hwr;
            int jlo = 0, jhi = -1, ilo = 0, ihi = -1;
            if (inb && amin > 1e-6f && amax >= 0.f && amin <= 1.f) {
                const float i0 = 1.f / amin, i1 = 1.f / amax;
                const float ja = (nj - dnj) * i0, jb = (nj - dnj) * i1, jc = (nj + dnj) * i0, jd = (nj + dnj) * i1;
                const float ia = (ni - dni) * i0, ib = (ni - dni) * i1, ic = (ni + dni) * i0, id = (ni + dni) * i1;
                const float jmn = fminf(fminf(ja, jb), fminf(jc, jd)) + P.gc0 - GATHER_WIN_MARGIN;
                const float jmx = fmaxf(fmaxf(ja, jb), fmaxf(jc, jd)) + P.gc0 + GATHER_WIN_MARGIN;
                const float imn = fminf(fminf(ia, ib), fminf(ic, id)) + P.gr0 - GATHER_WIN_MARGIN;
                const float imx = fmaxf(fmaxf(ia, ib), fmaxf(ic, id)) + P.gr0 + GATHER_WIN_MARGIN;
                jlo = (int)ceilf(fmaxf(jmn, 0.f));
                jhi = (int)floorf(fminf(jmx, (float)(G.W - 1)));
                ilo = (int)ceilf(fmaxf(imn, 0.f));
                ihi = (int)floorf(fminf(imx, (float)(G.H - 1)));
            } else if (inb && amin <= 1e-6f && amax >= 0.f) {
                jhi = G.W - 1;   // the block reaches the source plane: no perspective bound -- visit every ray
                ihi = G.H - 1;
            }
            float lx[3], ly[3], lz[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) { lx[k] = px[k] - s0; ly[k] = py[k] - s1; lz[k] = pz[k] - s2; }
            const float4* __restrict__ q = G.q + (size_t)p * G.n;
            const float2* __restrict__ q2 = G.q2 + (size_t)p * G.n;
            for (int i = ilo; i <= ihi; ++i) {
                const float4* __restrict__ row = q + (size_t)i * G.W;
                const float2* __restrict__ row2 = q2 + (size_t)i * G.W;
                // two candidates per trip: the four loads are issued before either candidate is evaluated
                for (int j = jlo; j <= jhi; j += 2) {
                    const int j1 = j < jhi ? j + 1 : j;
                    float4 tt[2] = {row[j], row[j1]};
                    const float2 aa[2] = {row2[j], row2[j1]};
                    if (j1 == j) tt[1].w = 0.f;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const float4 t = tt[h];
                        const float2 ab = aa[h];
                        // crossing alphas of the planes (forward's expression), then per axis the two voxel
                        // intervals; the ray's own [alpha_lo, alpha_hi] is folded into the x intervals once
                        const float x0 = lx[0] * t.x, x1 = lx[1] * t.x, x2 = lx[2] * t.x;
                        const float y0 = ly[0] * t.y, y1 = ly[1] * t.y, y2 = ly[2] * t.y;
                        const float z0 = lz[0] * t.z, z1 = lz[1] * t.z, z2 = lz[2] * t.z;
                        const float xl[2] = {fmaxf(fminf(x0, x1), ab.x), fmaxf(fminf(x1, x2), ab.x)};
                        const float xh[2] = {fminf(fmaxf(x0, x1), ab.y), fminf(fmaxf(x1, x2), ab.y)};
                        const float yl[2] = {fminf(y0, y1), fminf(y1, y2)}, yh[2] = {fmaxf(y0, y1), fmaxf(y1, y2)};
                        const float zl[2] = {fminf(z0, z1), fminf(z1, z2)}, zh[2] = {fmaxf(z0, z1), fmaxf(z1, z2)};
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const int a = e >> 2, b = (e >> 1) & 1, c = e & 1;
                            const float en = fmaxf(fmaxf(xl[a], yl[b]), zl[c]);
                            const float ex = fminf(fminf(xh[a], yh[b]), zh[c]);
                            // (alphas live in [0, 1]: the [0, 1] clamp is the max with 0, folded into the subtract)
                            acc[e] = fmaf(__builtin_amdgcn_fmed3f(ex - en, 0.f, 1.f), t.w, acc[e]);
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int x = vx + (e >> 2), y = vy + ((e >> 1) & 1), z = vz + (e & 1);
        if (x < G.D0 && y < G.D1 && z < G.D2 && acc[e] != 0.f) G.gvol[((size_t)x * G.D1 + y) * G.D2 + z] += acc[e];
    }
}


// =============================================================================================
// pose-side backward from the saved jacobian (C == 1): elementwise + wave reduction
// =============================================================================================
__global__ __launch_bounds__(WG) void k_backward_from_jac(const float* __restrict__ jac, const float* __restrict__ gout,
                                                         int n, float* gsrc, float* __restrict__ gtgt,
                                                         float* __restrict__ glen) {
    __shared__ float part[3][WG / 64];
    const int b = blockIdx.y;
    const int r = blockIdx.x * WG + threadIdx.x;
    float js[3] = {0.f, 0.f, 0.f};
    if (r < n) {
        const size_t ray = (size_t)b * n + r;
        const float g = gout[ray];
        const float4* jp = reinterpret_cast<const float4*>(jac + ray * XVR_DRR_JAC_STRIDE);
        const float4 j0 = jp[0], j1 = jp[1];
        js[0] = g * j0.y; js[1] = g * j0.z; js[2] = g * j0.w;
        float* tp = gtgt + ray * 3;
        tp[0] = g * j1.x; tp[1] = g * j1.y; tp[2] = g * j1.z;
        if (glen) glen[ray] = g * j0.x;
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
// Siddon: exact traversal as an incremental merge of the three per-axis plane-crossing sequences
// (no sort, no materialised alpha list).  MODE 0: forward; 1: forward + jacobian; 2: backward.
// =============================================================================================
// EXACT: the index map is the exact-geometry one (a = 1, b = shift - 1/2), so the voxel a segment
// belongs to is the voxel between the planes just crossed: it is tracked incrementally (+-1 on the
// crossed axis) instead of being re-derived from every segment's midpoint.
// SPLIT (forward, no mask, exact geometry only): 0 = one lane walks the whole ray; 1 / 2 = the ray's
// alpha range is cut into NS equal slices walked by NS wavefronts of the workgroup (8x8 / 16x16 tiles, see
// k_trilinear_fwd_split) and the partial sums meet in LDS.  A voxel segment that straddles a cut is
// credited to the same voxel from both sides (exact geometry: the voxel is the one between the planes),
// so only the rounding of that one product differs from the unsplit walk.
// (at most 5 wavefronts per SIMD: the walk is bound by the texture-address unit; 10.7 ms at C3 against 11.5 ms at
//  the 8 its register count would allow and at 4)
template <int MODE, int MASK, bool GPOSE, bool GVOL, bool EXACT, int SPLIT = 0>
__global__ __launch_bounds__(SPLIT ? 64 * SPLIT_MAX : WG) __attribute__((amdgpu_waves_per_eu(1, 5))) void k_siddon(RenderArgs A) {
    extern __shared__ float lds[];  // MASK: fwd -> channel accumulators, bwd -> upstream gradients; SPLIT: partial sums
    static_assert(!SPLIT || (MODE != 2 && !MASK && EXACT), "split walk: forward, unmasked, exact geometry");
    if (MODE == 2 && A.skip_unless_flag_gt && !(*A.skip_unless_flag_gt > __float_as_uint(GATHER_DEV_TOL))) return;
    int b, r, sl_l = 0, sl_w = 0, sl_n = 1;
    const bool valid = SPLIT ? map_ray_split<SPLIT == 2>(A, b, r, sl_l, sl_w, sl_n) : map_ray(A, b, r, threadIdx.x);
    const int tid = threadIdx.x;
    Ray R;
    ray_setup(A, b, r, valid, R);
    const float* __restrict__ vol = A.volume;
    const int D0 = A.D0, D1 = A.D1, D2 = A.D2;
    constexpr bool BWD = MODE == 2;
    constexpr bool DERIV = MODE == 1 || (BWD && GPOSE);

    float g0 = 1.f;
    if (MASK) {
        for (int c = 0; c < A.C; ++c)
            lds[c * WG + tid] = (BWD && valid) ? A.gout[((size_t)b * A.C + c) * A.n + r] : 0.f;
    } else if (BWD) {
        g0 = valid ? A.gout[(size_t)b * A.n + r] : 0.f;
    }

    float alo = R.amin, ahi = R.amax;
    if (SPLIT) {   // cut k sits at fmaf(k / NS, amax - amin, amin): both neighbours compute it identically
        const float a0 = R.amin, a1 = R.amax, inv = 1.f / (float)sl_n;
        if (sl_w > 0) alo = fmaf((float)sl_w * inv, a1 - a0, a0);
        if (sl_w < sl_n - 1) ahi = fmaf((float)(sl_w + 1) * inv, a1 - a0, a0);
    }
    bool live = valid && (ahi > alo) && (R.amax > R.amin);
    // per axis: reciprocal direction, (plane0 - s) so that alpha(p) = ((float)p + ps) * inv_d -- the same
    // value the sort formulation computes as ((p + plane0) - s) / d up to the reciprocal's rounding --,
    // index of the next plane to cross, step, alpha of that plane.  No range check on p: a plane beyond
    // the volume has alpha >= the axis' exit alpha >= ahi and is never selected before the loop ends.
    float inv_d[3], an3[3];
    int ip[3], stp[3];
    bool on_plane = false;   // the walk starts on a plane (always at the entry face; at a cut: see plane_at_cut)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        inv_d[i] = 1.f / R.d[i];
        const float f = fmaf(alo, R.d[i], R.s[i]) - A.sp.plane0[i];  // position in plane-index units
        if (R.d[i] > 0.f) { stp[i] = 1; ip[i] = (int)floorf(f) + 1; }
        else { stp[i] = -1; ip[i] = (int)ceilf(f) - 1; }
        an3[i] = (((float)ip[i] + A.sp.plane0[i]) - R.s[i]) * inv_d[i];
        if (an3[i] <= alo) {  // a plane at or behind the entry point (fp noise) is skipped
            on_plane = true;
            ip[i] += stp[i];
            an3[i] = (((float)ip[i] + A.sp.plane0[i]) - R.s[i]) * inv_d[i];
        } else if (SPLIT && sl_w > 0) {
            // a slice must start at the FIRST plane beyond its cut by the same alpha arithmetic the previous
            // slice ends with; the position-based guess above can be one plane late when the cut sits
            // within an ulp of a plane
            const float ap = (((float)(ip[i] - stp[i]) + A.sp.plane0[i]) - R.s[i]) * inv_d[i];
            if (ap > alo) { ip[i] -= stp[i]; an3[i] = ap; }
            else if (ap == alo) on_plane = true;
        }
        if (!live) an3[i] = INFINITY;
    }
    // EXACT: the current voxel along an axis is the one just before the next plane in travel direction
    const int vo0 = stp[0] > 0 ? 1 : 0, vo1 = stp[1] > 0 ? 1 : 0, vo2 = stp[2] > 0 ? 1 : 0;

    float acc = 0.f;                 // sum V * dalpha (C==1 fwd) / sum g V dalpha (bwd)
    float As[3] = {0.f, 0.f, 0.f};   // sum dW (alpha-1)/d  per axis
    float At[3] = {0.f, 0.f, 0.f};   // sum dW (-alpha)/d   per axis
    float Wprev = 0.f;
    int ax_prev = (SPLIT && sl_w > 0) ? -1 : R.ax_in;  // axis of the crossing that opened the current segment (-1: none)
    float ac = alo;
    unsigned cnt = 0;
    // the work counter counts voxel segments: a slice that starts inside a voxel continues the previous
    // slice's last segment
    bool first_of_slice = SPLIT && sl_w > 0 && !on_plane;
    const int max_iter = D0 + D1 + D2 + 8;

    // Software pipeline, depth 1: the voxel (and label) of segment i is requested, then segment i-1 --
    // whose load has had a whole traversal step to arrive -- is consumed.  The traversal itself never
    // depends on loaded values, only the accumulation does.
    bool have = false, exited = false;
    float p_v = 0.f, p_seg = 0.f, p_ac = 0.f, p_lab = 0.f;
    int p_ax = -1, p_off = 0;
    bool p_inb = false;
    float W = 0.f;

    auto consume = [&]() {
        const float v = p_v;
        int lab = 0;
        if (MASK) lab = p_inb ? min(max((int)p_lab, 0), A.C - 1) : 0;
        W = v;
        if (BWD) {
            const float gk = MASK ? lds[lab * WG + tid] : g0;
            W = gk * v;
            if (GVOL && p_inb) {
                const float c = gk * R.L * p_seg;
                if (c != 0.f) atomic_add_f32(A.gvol + p_off, c);
            }
            acc = fmaf(W, p_seg, acc);
        } else if (MASK) {
            lds[lab * WG + tid] = fmaf(v, p_seg, lds[lab * WG + tid]);
            if (MODE == 1) acc = fmaf(v, p_seg, acc);  // jacobian of the channel sum
        } else {
            acc = fmaf(v, p_seg, acc);
        }
        if (DERIV) {
            const float dW = Wprev - W;  // d out / d alpha at the crossing that opened this segment
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float m = (p_ax == i) ? dW * inv_d[i] : 0.f;
                As[i] = fmaf(m, p_ac - 1.f, As[i]);
                At[i] = fmaf(m, -p_ac, At[i]);
            }
            Wprev = W;
        }
    };

    for (int it = 0; it < max_iter; ++it) {
        if (!live) break;
        const float an = fminf(fminf(an3[0], an3[1]), fminf(an3[2], ahi));
        int ix, iy, iz;
        if (EXACT) {
            ix = ip[0] - vo0; iy = ip[1] - vo1; iz = ip[2] - vo2;
        } else {
            const float mid = 0.5f * (ac + an);
            ix = (int)rintf(fmaf(A.sp.a[0], fmaf(mid, R.d[0], R.s[0]), A.sp.b[0]));
            iy = (int)rintf(fmaf(A.sp.a[1], fmaf(mid, R.d[1], R.s[1]), A.sp.b[1]));
            iz = (int)rintf(fmaf(A.sp.a[2], fmaf(mid, R.d[2], R.s[2]), A.sp.b[2]));
        }
        const bool inb = (unsigned)ix < (unsigned)D0 && (unsigned)iy < (unsigned)D1 && (unsigned)iz < (unsigned)D2;
        const int off = inb ? (ix * D1 + iy) * D2 + iz : 0;
        const float v_new = vol[off];                       // always loadable (offset 0 when outside)
        // MASK == 2: the label rides in the low mantissa bits of the voxel just loaded (xvr_drr_pack_labels)
        const float lab_new = MASK == 2 ? (float)(__float_as_uint(v_new) & LABEL_MASK) : (MASK ? A.mask[off] : 0.f);
        if (inb && !first_of_slice && (!SPLIT || an > ac)) ++cnt;
        first_of_slice = false;
        if (have) consume();
        p_v = inb ? v_new : 0.f; p_seg = an - ac; p_ac = ac; p_ax = ax_prev; p_off = off; p_inb = inb; p_lab = lab_new;
        have = true;
        // advance every axis whose next plane has been reached (ties advance together), branch-free
        const bool c0 = an3[0] <= an, c1 = an3[1] <= an, c2 = an3[2] <= an;
        // A plane exactly AT a cut (routine: the cuts of opposite-face rays fall on the centre planes) is
        // crossed by the slice that ends there: one more, zero-length, segment carries its jacobian term;
        // the next slice starts behind the plane.
        const bool plane_at_cut = SPLIT && sl_w < sl_n - 1 && (c0 || c1 || c2);
        if (an >= ahi && !plane_at_cut) {
            exited = true;
            live = false;
        } else {
            ip[0] += c0 ? stp[0] : 0; ip[1] += c1 ? stp[1] : 0; ip[2] += c2 ? stp[2] : 0;
#pragma unroll
            for (int i = 0; i < 3; ++i) an3[i] = (((float)ip[i] + A.sp.plane0[i]) - R.s[i]) * inv_d[i];
            ax_prev = c0 ? 0 : (c1 ? 1 : 2);
            ac = an;
        }
    }
    if (have) consume();
    if (DERIV && exited && (!SPLIT || sl_w == sl_n - 1)) {  // exit crossing: beyond it W = 0
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float m = (R.ax_out == i) ? W * inv_d[i] : 0.f;
            As[i] = fmaf(m, ahi - 1.f, As[i]);
            At[i] = fmaf(m, -ahi, At[i]);
        }
    }

    if (SPLIT) {
        constexpr int TL = SPLIT == 2 ? 256 : 64;
        if (sl_w > 0) {
            float* p = lds + (size_t)(sl_w - 1) * 7 * TL + sl_l;
            p[0] = acc;
            if (MODE == 1) {
#pragma unroll
                for (int i = 0; i < 3; ++i) { p[(1 + i) * TL] = As[i]; p[(4 + i) * TL] = At[i]; }
            }
        }
        __syncthreads();
        if (sl_w == 0) {
            for (int v = 1; v < sl_n; ++v) {
                const float* p = lds + (size_t)(v - 1) * 7 * TL + sl_l;
                acc += p[0];
                if (MODE == 1) {
#pragma unroll
                    for (int i = 0; i < 3; ++i) { As[i] += p[(1 + i) * TL]; At[i] += p[(4 + i) * TL]; }
                }
            }
        }
    }
    if (!BWD) {
        if (valid && (!SPLIT || sl_w == 0)) {
            if (MASK) {
                for (int c = 0; c < A.C; ++c) A.out[((size_t)b * A.C + c) * A.n + r] = lds[c * WG + tid] * R.L;
            } else {
                A.out[(size_t)b * A.n + r] = acc * R.L;
            }
            if (MODE == 1) {
                float4* jp = reinterpret_cast<float4*>(A.jac + ((size_t)b * A.n + r) * XVR_DRR_JAC_STRIDE);
                jp[0] = make_float4(acc, R.L * As[0], R.L * As[1], R.L * As[2]);
                jp[1] = make_float4(R.L * At[0], R.L * At[1], R.L * At[2], 0.f);
            }
        }
        if (A.work) {
            unsigned tot = wave_sum_u(cnt);
            if ((tid & 63) == 0 && tot) atomicAdd(A.work, (unsigned long long)tot);
        }
    } else if (GPOSE) {
        float js[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) js[i] = valid ? R.L * As[i] : 0.f;
        if (valid) {
            float* tp = A.gtgt + ((size_t)b * A.n + r) * 3;
            tp[0] = R.L * At[0]; tp[1] = R.L * At[1]; tp[2] = R.L * At[2];
            if (A.glen) A.glen[(size_t)b * A.n + r] = acc;
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float tot = wave_sum_f(js[i]);
            if ((tid & 63) == 0 && tot != 0.f) atomic_add_f32(A.gsrc + 3 * b + i, tot);
        }
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
__global__ __launch_bounds__(WG) void k_rays_bwd(const float* __restrict__ cam, int H, int W, const float* __restrict__ g_source,
                                                 const float* __restrict__ g_target, const float* __restrict__ g_raylen,
                                                 float* g_cam) {
    __shared__ float part[21][WG / 64];
    const int b = blockIdx.y, n = H * W;
    const int r = blockIdx.x * WG + threadIdx.x;
    const float* c = cam + 24 * b;
    float acc[21];
#pragma unroll
    for (int q = 0; q < 21; ++q) acc[q] = 0.f;
    if (r < n) {
        const int i = r / W, j = r - i * W;
        const float pix[3] = {(float)i, (float)j, 1.f};
        const float* gt = g_target + ((size_t)b * n + r) * 3;
        float w[3], l2 = 0.f;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            w[a] = fmaf(c[12 + 3 * a], pix[0], fmaf(c[12 + 3 * a + 1], pix[1], c[12 + 3 * a + 2])) - c[21 + a];
            l2 = fmaf(w[a], w[a], l2);
        }
        const float gl = g_raylen ? g_raylen[(size_t)b * n + r] : 0.f;
        const float s = l2 > 0.f ? gl / sqrtf(l2) : 0.f;  // g_L * (unit direction) = s * w
#pragma unroll
        for (int a = 0; a < 3; ++a) {
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                acc[3 * a + m] = gt[a] * pix[m];           // d/d Mv[a][m]
                acc[9 + 3 * a + m] = s * w[a] * pix[m];    // d/d Mw[a][m]
            }
            acc[18 + a] = -s * w[a];                        // d/d s_w[a]
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
__global__ __launch_bounds__(WG) void k_jac_to_cam(const float* __restrict__ jac, const float* __restrict__ gout,
                                                   const float* __restrict__ cam, int H, int W, float* partial,
                                                   unsigned* counter, float* __restrict__ g_cam) {
    __shared__ float part[24][WG / 64];
    __shared__ float fin[24][WG + 1];
    __shared__ bool last;
    const int b = blockIdx.y, n = H * W, nblk = gridDim.x;
    const int r = blockIdx.x * WG + threadIdx.x;
    const float* c = cam + 24 * b;
    float acc[24];
#pragma unroll
    for (int q = 0; q < 24; ++q) acc[q] = 0.f;
    if (r < n) {
        const size_t ray = (size_t)b * n + r;
        const float g = gout[ray];
        const float4* jp = reinterpret_cast<const float4*>(jac + ray * XVR_DRR_JAC_STRIDE);
        const float4 j0 = jp[0], j1 = jp[1];
        const int i = r / W, j = r - i * W;
        const float pix[3] = {(float)i, (float)j, 1.f};
        const float gt[3] = {g * j1.x, g * j1.y, g * j1.z};
        float w[3], l2 = 0.f;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            w[a] = fmaf(c[12 + 3 * a], pix[0], fmaf(c[12 + 3 * a + 1], pix[1], c[12 + 3 * a + 2])) - c[21 + a];
            l2 = fmaf(w[a], w[a], l2);
        }
        const float gl = g * j0.x;
        const float sc = l2 > 0.f ? gl / sqrtf(l2) : 0.f;  // g_L * (unit direction) = sc * w
        acc[9] = g * j0.y; acc[10] = g * j0.z; acc[11] = g * j0.w;   // d/d s_v = grad_source
#pragma unroll
        for (int a = 0; a < 3; ++a) {
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                acc[3 * a + m] = gt[a] * pix[m];             // d/d Mv[a][m]
                acc[12 + 3 * a + m] = sc * w[a] * pix[m];    // d/d Mw[a][m]
            }
            acc[21 + a] = -sc * w[a];                         // d/d s_w[a]
        }
    }
#pragma unroll
    for (int q = 0; q < 24; ++q) {
        const float tot = wave_sum_f(acc[q]);
        if ((threadIdx.x & 63) == 0) part[q][threadIdx.x >> 6] = tot;
    }
    __syncthreads();
    // Publishing without a device-wide fence (a release fence writes the XCD's whole L2 back: ~70 ns per
    // block, serialised -- measured 4x slower than the kernel itself): the partials are stored and
    // loaded with agent-scope atomics, which go to the level where the 8 XCDs are coherent; the
    // barrier's wait for outstanding stores orders them before this block's ticket, and the block that
    // draws the last ticket therefore reads everybody's partials.
    float* mine = partial + ((size_t)b * nblk + blockIdx.x) * 24;
    if (threadIdx.x < 24)
        __hip_atomic_store(mine + threadIdx.x, (part[threadIdx.x][0] + part[threadIdx.x][1]) + (part[threadIdx.x][2] + part[threadIdx.x][3]),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();   // includes s_waitcnt vmcnt(0): the stores above have completed
    if (threadIdx.x == 0)
        last = __hip_atomic_fetch_add(counter + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(nblk - 1);
    __syncthreads();
    if (!last) return;
    // thread t adds rows t, t + 256, ... (all 24 columns, independent loads), then the 256 per-thread sums
    // are added column by column in thread order
    float t24[24];
#pragma unroll
    for (int q = 0; q < 24; ++q) t24[q] = 0.f;
    const float* P = partial + (size_t)b * nblk * 24;
    for (int k = threadIdx.x; k < nblk; k += WG) {
#pragma unroll
        for (int q = 0; q < 24; ++q)
            t24[q] += __hip_atomic_load(P + (size_t)k * 24 + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int q = 0; q < 24; ++q) fin[q][threadIdx.x] = t24[q];
    __syncthreads();
    if (threadIdx.x < 24) {
        float t = 0.f;
        for (int k = 0; k < WG; ++k) t += fin[threadIdx.x][k];
        g_cam[24 * b + threadIdx.x] = t;
    }
    if (threadIdx.x == 0) counter[b] = 0;   // ready for the next call
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
int fail(int code, const char* msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}

int check_common(const float* volume, int D0, int D1, int D2, int C, const float* source, const float* target,
                 const float* raylen, int B, int n, const xvr_drr_spec* sp, const float* cam = nullptr) {
    if (!volume || !sp || (!cam && (!source || !target || !raylen))) return fail(XVR_DRR_E_ARG, "null pointer argument");
    if (cam && !(sp->ray_grid_w > 1 && n / sp->ray_grid_w > 0)) return fail(XVR_DRR_E_ARG, "camera rays need a detector lattice");
    if (D0 < 2 || D1 < 2 || D2 < 2) return fail(XVR_DRR_E_ARG, "every volume dimension must be >= 2");
    if ((long long)D0 * D1 * D2 >= (1LL << 31)) return fail(XVR_DRR_E_UNSUPPORTED, "volume has >= 2^31 voxels");
    if (B <= 0 || n <= 0) return fail(XVR_DRR_E_ARG, "B and n must be positive");
    if (C < 1 || C > 128) return fail(XVR_DRR_E_ARG, "C must be in [1, 128]");
    for (int i = 0; i < 3; ++i)
        if (!(sp->a[i] > 0.f)) return fail(XVR_DRR_E_ARG, "spec.a must be positive");
    if (sp->ray_grid_w < 0 || (sp->ray_grid_w > 0 && n % sp->ray_grid_w != 0))
        return fail(XVR_DRR_E_ARG, "ray_grid_w must divide n");
    return XVR_DRR_OK;
}

void fill_args(RenderArgs& A, const float* volume, const float* mask, int D0, int D1, int D2, int C,
               const float* source, const float* target, const float* raylen, int B, int n,
               const xvr_drr_spec* sp, const float* cam = nullptr) {
    A = RenderArgs{};
    A.volume = volume; A.mask = mask;
    A.D0 = D0; A.D1 = D1; A.D2 = D2; A.C = C;
    A.source = source; A.target = target; A.raylen = raylen; A.cam = cam;
    A.B = B; A.n = n; A.sp = *sp;
    A.grid_w = sp->ray_grid_w;
    static const int forced_shape = [] {
        const char* e = getenv("XVR_DRR_TILE_SHAPE");  // A/B switch: 0, 1, 2; default: per-workgroup choice
        return (e && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : -1;
    }();
    A.tile_shape = forced_shape;
    if (A.grid_w > 0) {
        A.grid_h = n / A.grid_w;
        A.tiles_x = (A.grid_w + 15) / 16;
        A.blocks_per_pose = A.tiles_x * ((A.grid_h + 15) / 16);
    } else {
        A.grid_h = 0; A.tiles_x = 0;
        A.blocks_per_pose = (n + WG - 1) / WG;
    }
}

template <typename Kern>
int launch(Kern kern, const RenderArgs& A, size_t lds_bytes, void* stream) {
    const long long nblocks = (long long)A.B * A.blocks_per_pose;
    if (nblocks >= (1LL << 31)) return fail(XVR_DRR_E_UNSUPPORTED, "grid too large");
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return fail(XVR_DRR_E_LAUNCH, hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(WG), lds_bytes, (hipStream_t)stream, A);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(XVR_DRR_E_LAUNCH, hipGetErrorString(e));
    return XVR_DRR_OK;
}

// Sample slices per ray for the forward march, measured on MI355X (tools/bench_small_batch.py,
// profiles/r01_small_batch_split.txt).  Splitting pays while the launch is LATENCY-bound: few
// wavefronts, each walking ~250 dependent gathers.  It stops paying once the launch is bound by
// throughput (>= ~2048 wavefronts) or by the first-touch read of a frustum that spans a volume larger
// than the caches.  Returns 1 for the unsplit kernel.
// XVR_DRR_FWD_SPLIT forces a choice: <n> = 8x8 tiles x n slices, 1<nn> (102, 104) = 16x16 tiles x n.
int split_factor(int B, int n, long long voxels, bool siddon, bool* tile16) {
    const char* env = getenv("XVR_DRR_FWD_SPLIT");   // read per call: tests switch it within one process
    const int forced = env ? atoi(env) : 0;
    int ns = 1;
    *tile16 = false;
    if (forced > 0) {
        *tile16 = forced >= 100;
        const int want = forced % 100, cap = *tile16 ? 4 : SPLIT_MAX;
        while (ns * 2 <= want && ns * 2 <= cap) ns *= 2;
        return ns;
    }
    const long long waves = (long long)B * ((n + 63) / 64);
    if (siddon) {   // one dependent load per step: latency-bound for longer than the trilinear march
        if (waves <= 256) return 8;
        if (waves <= 512) return 4;
        if (waves <= 2048) { *tile16 = true; return 2; }
        return 1;
    }
    const bool tiny_vol = voxels * 4 <= (32LL << 20);    // at home in the L2s
    if (waves <= 128) return tiny_vol ? 8 : 4;
    if (waves <= 256) return 4;
    if (waves <= 512) return tiny_vol ? 4 : 2;
    // one pose at 256^2: what decides is the FOOTPRINT of the frustum, which the host does not know.  A
    // detector that looks at 16 % of a 512^3 CT (registration geometry, 0.8-voxel pixel pitch) renders in
    // 57 us split vs 111 us unsplit; one whose frustum spans the whole 0.5 GB volume loses 5 % (99 vs 95 us).
    if (waves <= 1024) { *tile16 = true; return 4; }
    return 1;
}

template <typename Kern>
int launch_split(Kern kern, RenderArgs A, int ns, bool tile16, int vals, void* stream) {
    const int tl = tile16 ? 256 : 64;
    if (!tile16) {
        if (A.grid_w > 0) {   // 8x8 tiles, one per workgroup
            A.tiles_x = (A.grid_w + 7) / 8;
            A.blocks_per_pose = A.tiles_x * ((A.grid_h + 7) / 8);
        } else {
            A.blocks_per_pose = (A.n + 63) / 64;
        }
    }
    const long long nblocks = (long long)A.B * A.blocks_per_pose;
    if (nblocks >= (1LL << 31)) return fail(XVR_DRR_E_UNSUPPORTED, "grid too large");
    const size_t lds_bytes = (size_t)(ns - 1) * vals * tl * sizeof(float);
    hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(tl * ns), lds_bytes, (hipStream_t)stream, A);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(XVR_DRR_E_LAUNCH, hipGetErrorString(e));
    return XVR_DRR_OK;
}

// workspace layout of the gather path:
//   [flag, 256 B][PoseLattice x B, 256-aligned][float4 x B*n][cull words: bricks x ceil(B/32)]
size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
long long n_bricks(int D0, int D1, int D2, const int* bd) {
    return (long long)((D0 + bd[0] - 1) / bd[0]) * ((D1 + bd[1] - 1) / bd[1]) * ((D2 + bd[2] - 1) / bd[2]);
}
long long n_bricks_max(int D0, int D1, int D2) {  // the finest brick any variant uses: 4 x 4 x 4
    const int bd[3] = {4, 4, 4};
    return n_bricks(D0, D1, D2, bd);
}
// voxels per lane and axis in the trilinear gather: 2 unless XVR_DRR_GATHER_BLOCK=1 (A/B switch)
int gather_block() {
    static const int v = [] {
        const char* e = getenv("XVR_DRR_GATHER_BLOCK");
        return (e && e[0] == '1') ? 1 : 2;
    }();
    return v;
}
size_t ws_pose_off() { return 256; }
size_t ws_q_off(int B) { return 256 + align256((size_t)B * sizeof(PoseLattice)); }
size_t ws_q2_off(int B, int n) { return ws_q_off(B) + align256((size_t)B * (size_t)n * sizeof(float4)); }
size_t ws_cull_off(int B, int n) { return ws_q2_off(B, n) + align256((size_t)B * (size_t)n * sizeof(float2)); }
size_t ws_bytes(int B, int n, int D0, int D1, int D2) {
    return ws_cull_off(B, n) + (size_t)n_bricks_max(D0, D1, D2) * (size_t)((B + 31) / 32) * sizeof(unsigned);
}

// Set up the workspace and launch prep -> cull -> gather.  The caller launches the scatter fallback
// (with skip_unless_flag_gt = the returned flag) right behind it.
int launch_gather(bool siddon, const float* source, const float* target, const float* raylen, const float* grad_out,
                  int B, int n, int gw, int D0, int D1, int D2, const xvr_drr_spec* sp, float* grad_volume,
                  void* workspace, void* stream, unsigned** flag_out) {
    char* ws = static_cast<char*>(workspace);
    GatherArgs G = {};
    G.source = source; G.target = target; G.raylen = raylen; G.gout = grad_out;
    G.B = B; G.n = n; G.W = gw; G.H = n / gw; G.D0 = D0; G.D1 = D1; G.D2 = D2; G.sp = *sp;
    G.flag = reinterpret_cast<unsigned*>(ws);
    G.poses = reinterpret_cast<PoseLattice*>(ws + ws_pose_off());
    G.q = reinterpret_cast<float4*>(ws + ws_q_off(B));
    G.q2 = reinterpret_cast<float2*>(ws + ws_q2_off(B, n));
    G.siddon = siddon ? 1 : 0;
    G.V = siddon ? 1 : gather_block();
    // Siddon: 2x2x2 voxels per lane in 8^3 bricks unless XVR_DRR_SIDDON_GATHER_BLOCK=1 (A/B switch: one voxel per
    // lane, 256 lanes on a 4 x 8 x 8 brick)
    static const bool siddon_v1 = [] { const char* e = getenv("XVR_DRR_SIDDON_GATHER_BLOCK"); return e && e[0] == '1'; }();
    if (siddon && siddon_v1) { G.bd[0] = 4; G.bd[1] = 8; G.bd[2] = 8; }
    else if (siddon) { G.bd[0] = G.bd[1] = G.bd[2] = 8; }
    else { G.bd[0] = G.bd[1] = G.bd[2] = 4 * G.V; }
    G.cull = reinterpret_cast<unsigned*>(ws + ws_cull_off(B, n));
    G.words = (B + 31) / 32;
    G.gvol = grad_volume;
    *flag_out = G.flag;
    hipError_t e = hipMemsetAsync(G.flag, 0, 16, (hipStream_t)stream);
    if (e != hipSuccess) return fail(XVR_DRR_E_LAUNCH, hipGetErrorString(e));
    hipLaunchKernelGGL(k_gather_prep, dim3((unsigned)((n + WG - 1) / WG), (unsigned)B), dim3(WG), 0,
                       (hipStream_t)stream, G);
    const long long bricks = n_bricks(D0, D1, D2, G.bd);
    if (bricks >= (1LL << 31)) return fail(XVR_DRR_E_UNSUPPORTED, "grid too large");
    hipLaunchKernelGGL(k_gather_cull, dim3((unsigned)((bricks + WG / 32 - 1) / (WG / 32))), dim3(WG), 0,
                       (hipStream_t)stream, G, (int)bricks);
    if (siddon && siddon_v1) hipLaunchKernelGGL(k_siddon_gather_vol, dim3((unsigned)bricks), dim3(WG), 0, (hipStream_t)stream, G);
    else if (siddon) hipLaunchKernelGGL(k_siddon_gather_vol2, dim3((unsigned)bricks), dim3(64), 0, (hipStream_t)stream, G);
    else if (G.V == 2 && getenv("XVR_DRR_GATHER_ABLATE")) hipLaunchKernelGGL((k_trilinear_gather_vol<2, true>), dim3((unsigned)bricks), dim3(64), 0, (hipStream_t)stream, G);
    else if (G.V == 2) hipLaunchKernelGGL(k_trilinear_gather_vol<2>, dim3((unsigned)bricks), dim3(64), 0, (hipStream_t)stream, G);
    else hipLaunchKernelGGL(k_trilinear_gather_vol<1>, dim3((unsigned)bricks), dim3(64), 0, (hipStream_t)stream, G);
    e = hipGetLastError();
    if (e != hipSuccess) return fail(XVR_DRR_E_LAUNCH, hipGetErrorString(e));
    return XVR_DRR_OK;
}

// a = 1 and b = shift - 1/2: the sampling index is x + shift - 1/2, i.e. the nearest voxel of a segment's
// midpoint is the voxel between the planes that bound the segment
bool siddon_exact_geometry(const xvr_drr_spec* sp) {
    bool ok = true;
    for (int i = 0; i < 3; ++i)
        ok = ok && fabsf(sp->a[i] - 1.f) < 1e-6f && fabsf(sp->b[i] + sp->plane0[i] + 0.5f) < 1e-6f;
    return ok;
}

bool gather_usable(const xvr_drr_spec* sp, int n, void* workspace, size_t workspace_bytes, int B, int D0, int D1,
                   int D2) {
    const int gw = sp->ray_grid_w, gh = gw > 0 ? n / gw : 0;
    return gw > 1 && gh > 1 && workspace && workspace_bytes >= ws_bytes(B, n, D0, D1, D2) &&
           (reinterpret_cast<uintptr_t>(workspace) & 15u) == 0;
}

}  // namespace

extern "C" {

size_t xvr_drr_backward_workspace_bytes(int B, int n, int D0, int D1, int D2) {
    if (B <= 0 || n <= 0 || D0 <= 0 || D1 <= 0 || D2 <= 0) return 0;
    return ws_bytes(B, n, D0, D1, D2);
}

int xvr_drr_abi_version(void) { return XVR_DRR_ABI_VERSION; }
const char* xvr_drr_last_error(void) { return g_err; }
// shared by the other translation units of the library (sim_kernels.hip); not part of the public header
void xvr_drr_set_last_error(const char* msg) { snprintf(g_err, sizeof(g_err), "%s", msg ? msg : ""); }

static int trilinear_forward_impl(const float* volume, const float* mask, int D0, int D1, int D2, int C,
                              const float* source, const float* target, const float* raylen, int B, int n,
                              const xvr_drr_spec* sp, float* out, float* jac, unsigned long long* work,
                              void* stream, const float* cam) {
    int rc = check_common(volume, D0, D1, D2, C, source, target, raylen, B, n, sp, cam);
    if (rc) return rc;
    if (!out) return fail(XVR_DRR_E_ARG, "out is null");
    if (sp->n_points < 1) return fail(XVR_DRR_E_ARG, "n_points must be >= 1");
    if (!(sp->far_ >= sp->near_)) return fail(XVR_DRR_E_ARG, "far must be >= near");
    const bool packed = !mask && C > 1;   // labels in the low mantissa bits of `volume` (xvr_drr_pack_labels)
    if (packed && C > (1 << LABEL_BITS)) return fail(XVR_DRR_E_ARG, "packed labels hold at most 16 channels");
    RenderArgs A;
    fill_args(A, volume, mask, D0, D1, D2, C, source, target, raylen, B, n, sp, cam);
    A.out = out; A.jac = jac; A.work = work;
    const bool clip = sp->clip_to_volume != 0;
    const size_t lds = C > 1 || mask ? (size_t)C * WG * sizeof(float) : 0;
    if (packed && jac) return clip ? launch(k_trilinear_fwd<true, 2, true>, A, lds, stream)
                                   : launch(k_trilinear_fwd<true, 2, false>, A, lds, stream);
    if (packed) return clip ? launch(k_trilinear_fwd<false, 2, true>, A, lds, stream)
                            : launch(k_trilinear_fwd<false, 2, false>, A, lds, stream);
    if (mask && jac) return clip ? launch(k_trilinear_fwd<true, 1, true>, A, lds, stream)
                                 : launch(k_trilinear_fwd<true, 1, false>, A, lds, stream);
    if (mask) return clip ? launch(k_trilinear_fwd<false, 1, true>, A, lds, stream)
                          : launch(k_trilinear_fwd<false, 1, false>, A, lds, stream);
    // LDS-staged bricks are opt-in: measured 2.25x SLOWER than the direct kernel at C2 (19.8 vs 8.8 ms;
    // with ~4 taps per voxel the L1/L2 already capture the reuse)
    static const bool use_lds = [] {
        const char* e = getenv("XVR_DRR_FWD_LDS");
        return e && e[0] == '1';
    }();
    if (use_lds && !clip && A.grid_w > 0) {
        const size_t bytes = (size_t)(LDS_HDR + LDS_BRICK_CAP) * sizeof(float);
        return jac ? launch(k_trilinear_fwd_lds<true>, A, bytes, stream) : launch(k_trilinear_fwd_lds<false>, A, bytes, stream);
    }
    bool tile16 = false;
    const int ns = split_factor(B, n, (long long)D0 * D1 * D2, false, &tile16);
    if (ns > 1) {
#define XVR_SPLIT(J, Cl) (tile16 ? launch_split(k_trilinear_fwd_split<J, Cl, true>, A, ns, true, SPLIT_VALS, stream) \
                                 : launch_split(k_trilinear_fwd_split<J, Cl, false>, A, ns, false, SPLIT_VALS, stream))
        if (jac) return clip ? XVR_SPLIT(true, true) : XVR_SPLIT(true, false);
        return clip ? XVR_SPLIT(false, true) : XVR_SPLIT(false, false);
#undef XVR_SPLIT
    }
    if (jac) return clip ? launch(k_trilinear_fwd<true, 0, true>, A, 0, stream)
                         : launch(k_trilinear_fwd<true, 0, false>, A, 0, stream);
    return clip ? launch(k_trilinear_fwd<false, 0, true>, A, 0, stream)
                : launch(k_trilinear_fwd<false, 0, false>, A, 0, stream);
}

int xvr_drr_trilinear_forward(const float* volume, const float* mask, int D0, int D1, int D2, int C,
                           const float* source, const float* target, const float* raylen, int B, int n,
                           const xvr_drr_spec* sp, float* out, float* jac, unsigned long long* work, void* stream) {
    return trilinear_forward_impl(volume, mask, D0, D1, D2, C, source, target, raylen, B, n, sp, out, jac, work, stream, nullptr);
}

int xvr_drr_trilinear_forward_camera(const float* volume, const float* mask, int D0, int D1, int D2, int C,
                                  const float* cam, int B, int H, int W, const xvr_drr_spec* sp, float* out, float* jac,
                                  unsigned long long* work, void* stream) {
    if (!cam || !sp) return fail(XVR_DRR_E_ARG, "null pointer argument");
    if (H < 1 || W < 2 || (long long)H * W >= (1LL << 31)) return fail(XVR_DRR_E_ARG, "detector must be at least 1 x 2");
    xvr_drr_spec local = *sp;
    local.ray_grid_w = W;
    return trilinear_forward_impl(volume, mask, D0, D1, D2, C, nullptr, nullptr, nullptr, B, H * W, &local, out, jac, work, stream, cam);
}

int xvr_drr_trilinear_backward(const float* volume, const float* mask, int D0, int D1, int D2, int C,
                               const float* source, const float* target, const float* raylen, int B, int n,
                               const xvr_drr_spec* sp, const float* grad_out, float* grad_volume,
                               float* grad_source, float* grad_target, float* grad_raylen, void* workspace,
                               size_t workspace_bytes, void* stream) {
    int rc = check_common(volume, D0, D1, D2, C, source, target, raylen, B, n, sp);
    if (rc) return rc;
    if (!grad_out) return fail(XVR_DRR_E_ARG, "grad_out is null");
    if (sp->n_points < 1) return fail(XVR_DRR_E_ARG, "n_points must be >= 1");
    if (!mask && C != 1) return fail(XVR_DRR_E_ARG, "C must be 1 without a mask");
    if ((grad_source == nullptr) != (grad_target == nullptr))
        return fail(XVR_DRR_E_ARG, "grad_source and grad_target must be requested together");
    if (grad_raylen && !grad_target) return fail(XVR_DRR_E_ARG, "grad_raylen needs grad_source/grad_target");
    const bool gpose = grad_target != nullptr, gvol = grad_volume != nullptr;
    if (!gpose && !gvol) return XVR_DRR_OK;
    RenderArgs A;
    fill_args(A, volume, mask, D0, D1, D2, C, source, target, raylen, B, n, sp);
    A.gout = grad_out; A.gvol = grad_volume; A.gsrc = grad_source; A.gtgt = grad_target; A.glen = grad_raylen;
    const bool clip = sp->clip_to_volume != 0;
    const size_t lds = mask ? (size_t)C * WG * sizeof(float) : 0;

    // Voxel gradient by the atomic-free voxel-driven gather when the rays are a detector lattice
    // (no mask, no per-ray alpha rescaling); the scatter kernel stays as the general fallback and is
    // launched right behind it, reading the lattice flag on the device (no host sync).
    const bool gather = gvol && !mask && !clip && gather_usable(sp, n, workspace, workspace_bytes, B, D0, D1, D2);
    if (gather) {
        unsigned* flag = nullptr;
        rc = launch_gather(false, source, target, raylen, grad_out, B, n, sp->ray_grid_w, D0, D1, D2, sp, grad_volume,
                           workspace, stream, &flag);
        if (rc) return rc;
        if (gpose) {  // the pose part does not depend on how the voxel part is done
            RenderArgs Ap = A;
            Ap.gvol = nullptr;
            rc = launch(k_trilinear_bwd<false, false, true, false>, Ap, 0, stream);
            if (rc) return rc;
        }
        RenderArgs Av = A;
        Av.gsrc = nullptr; Av.gtgt = nullptr; Av.glen = nullptr;
        Av.skip_unless_flag_gt = flag;
        return launch(k_trilinear_bwd<false, false, false, true>, Av, 0, stream);
    }
#define TRI_BWD(M, CL)                                                                         \
    (gpose ? (gvol ? launch(k_trilinear_bwd<M, CL, true, true>, A, lds, stream)                \
                   : launch(k_trilinear_bwd<M, CL, true, false>, A, lds, stream))              \
           : launch(k_trilinear_bwd<M, CL, false, true>, A, lds, stream))
    if (mask) return clip ? TRI_BWD(true, true) : TRI_BWD(true, false);
    return clip ? TRI_BWD(false, true) : TRI_BWD(false, false);
#undef TRI_BWD
}

static int siddon_forward_impl(const float* volume, const float* mask, int D0, int D1, int D2, int C,
                           const float* source, const float* target, const float* raylen, int B, int n,
                           const xvr_drr_spec* sp, float* out, float* jac, unsigned long long* work,
                           void* stream, const float* cam) {
    int rc = check_common(volume, D0, D1, D2, C, source, target, raylen, B, n, sp, cam);
    if (rc) return rc;
    if (!out) return fail(XVR_DRR_E_ARG, "out is null");
    const bool packed = !mask && C > 1;   // labels in the low mantissa bits of `volume` (xvr_drr_pack_labels)
    if (packed && C > (1 << LABEL_BITS)) return fail(XVR_DRR_E_ARG, "packed labels hold at most 16 channels");
    RenderArgs A;
    fill_args(A, volume, mask, D0, D1, D2, C, source, target, raylen, B, n, sp, cam);
    A.out = out; A.jac = jac; A.work = work;
    const size_t lds = C > 1 || mask ? (size_t)C * WG * sizeof(float) : 0;
    const bool ex = siddon_exact_geometry(sp);
    if (packed && jac) return (ex ? launch(k_siddon<1, 2, false, false, true>, A, lds, stream) : launch(k_siddon<1, 2, false, false, false>, A, lds, stream));
    if (packed) return (ex ? launch(k_siddon<0, 2, false, false, true>, A, lds, stream) : launch(k_siddon<0, 2, false, false, false>, A, lds, stream));
    if (mask && jac) return (ex ? launch(k_siddon<1, true, false, false, true>, A, lds, stream) : launch(k_siddon<1, true, false, false, false>, A, lds, stream));
    if (mask) return (ex ? launch(k_siddon<0, true, false, false, true>, A, lds, stream) : launch(k_siddon<0, true, false, false, false>, A, lds, stream));
    bool tile16 = false;
    const int ns = ex ? split_factor(B, n, (long long)D0 * D1 * D2, true, &tile16) : 1;
    if (ns > 1) {
        if (jac) return tile16 ? launch_split(k_siddon<1, false, false, false, true, 2>, A, ns, true, 7, stream)
                               : launch_split(k_siddon<1, false, false, false, true, 1>, A, ns, false, 7, stream);
        return tile16 ? launch_split(k_siddon<0, false, false, false, true, 2>, A, ns, true, 7, stream)
                      : launch_split(k_siddon<0, false, false, false, true, 1>, A, ns, false, 7, stream);
    }
    if (jac) return (ex ? launch(k_siddon<1, false, false, false, true>, A, 0, stream) : launch(k_siddon<1, false, false, false, false>, A, 0, stream));
    return (ex ? launch(k_siddon<0, false, false, false, true>, A, 0, stream) : launch(k_siddon<0, false, false, false, false>, A, 0, stream));
}

int xvr_drr_siddon_forward(const float* volume, const float* mask, int D0, int D1, int D2, int C,
                           const float* source, const float* target, const float* raylen, int B, int n,
                           const xvr_drr_spec* sp, float* out, float* jac, unsigned long long* work, void* stream) {
    return siddon_forward_impl(volume, mask, D0, D1, D2, C, source, target, raylen, B, n, sp, out, jac, work, stream, nullptr);
}

int xvr_drr_siddon_forward_camera(const float* volume, const float* mask, int D0, int D1, int D2, int C,
                                  const float* cam, int B, int H, int W, const xvr_drr_spec* sp, float* out, float* jac,
                                  unsigned long long* work, void* stream) {
    if (!cam || !sp) return fail(XVR_DRR_E_ARG, "null pointer argument");
    if (H < 1 || W < 2 || (long long)H * W >= (1LL << 31)) return fail(XVR_DRR_E_ARG, "detector must be at least 1 x 2");
    xvr_drr_spec local = *sp;
    local.ray_grid_w = W;
    return siddon_forward_impl(volume, mask, D0, D1, D2, C, nullptr, nullptr, nullptr, B, H * W, &local, out, jac, work, stream, cam);
}

int xvr_drr_siddon_backward(const float* volume, const float* mask, int D0, int D1, int D2, int C,
                            const float* source, const float* target, const float* raylen, int B, int n,
                            const xvr_drr_spec* sp, const float* grad_out, float* grad_volume,
                            float* grad_source, float* grad_target, float* grad_raylen, void* workspace,
                            size_t workspace_bytes, void* stream) {

    int rc = check_common(volume, D0, D1, D2, C, source, target, raylen, B, n, sp);
    if (rc) return rc;
    if (!grad_out) return fail(XVR_DRR_E_ARG, "grad_out is null");
    if (!mask && C != 1) return fail(XVR_DRR_E_ARG, "C must be 1 without a mask");
    if ((grad_source == nullptr) != (grad_target == nullptr))
        return fail(XVR_DRR_E_ARG, "grad_source and grad_target must be requested together");
    if (grad_raylen && !grad_target) return fail(XVR_DRR_E_ARG, "grad_raylen needs grad_source/grad_target");
    const bool gpose = grad_target != nullptr, gvol = grad_volume != nullptr;
    if (!gpose && !gvol) return XVR_DRR_OK;
    RenderArgs A;
    fill_args(A, volume, mask, D0, D1, D2, C, source, target, raylen, B, n, sp);
    A.gout = grad_out; A.gvol = grad_volume; A.gsrc = grad_source; A.gtgt = grad_target; A.glen = grad_raylen;
    const size_t lds = mask ? (size_t)C * WG * sizeof(float) : 0;
    // the gather needs the exact-geometry index map (voxel credited = voxel whose box holds the segment)
    const bool exact_geom = siddon_exact_geometry(sp);
    if (gvol && !mask && exact_geom && gather_usable(sp, n, workspace, workspace_bytes, B, D0, D1, D2)) {
        unsigned* flag = nullptr;
        rc = launch_gather(true, source, target, raylen, grad_out, B, n, sp->ray_grid_w, D0, D1, D2, sp, grad_volume,
                           workspace, stream, &flag);
        if (rc) return rc;
        if (gpose) {
            RenderArgs Ap = A;
            Ap.gvol = nullptr;
            rc = launch(k_siddon<2, false, true, false, true>, Ap, 0, stream);
            if (rc) return rc;
        }
        RenderArgs Av = A;
        Av.gsrc = nullptr; Av.gtgt = nullptr; Av.glen = nullptr;
        Av.skip_unless_flag_gt = flag;
        return launch(k_siddon<2, false, false, true, true>, Av, 0, stream);
    }
#define SID_BWD2(M, E)                                                                  \
    (gpose ? (gvol ? launch(k_siddon<2, M, true, true, E>, A, lds, stream)              \
                   : launch(k_siddon<2, M, true, false, E>, A, lds, stream))            \
           : launch(k_siddon<2, M, false, true, E>, A, lds, stream))
#define SID_BWD(M) (exact_geom ? SID_BWD2(M, true) : SID_BWD2(M, false))
    return mask ? SID_BWD(true) : SID_BWD(false);
#undef SID_BWD
#undef SID_BWD2
}

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
    const unsigned nblk = (unsigned)(((size_t)H * W + WG - 1) / WG);
    char* ws = static_cast<char*>(workspace);
    hipLaunchKernelGGL(k_jac_to_cam, dim3(nblk, (unsigned)B), dim3(WG), 0, (hipStream_t)stream, jac, grad_out, cam, H, W,
                       reinterpret_cast<float*>(ws + align256((size_t)B * sizeof(unsigned))), reinterpret_cast<unsigned*>(ws),
                       grad_cam);
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
    dim3 grid((unsigned)(((long long)H * W + WG - 1) / WG), (unsigned)B);
    hipLaunchKernelGGL(k_rays_bwd, grid, dim3(WG), 0, (hipStream_t)stream, cam, H, W, grad_source, grad_target,
                       grad_raylen, grad_cam);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(XVR_DRR_E_LAUNCH, hipGetErrorString(e));
    return XVR_DRR_OK;
}

int xvr_drr_backward_from_jac(const float* jac, const float* grad_out, int B, int n, float* grad_source,
                              float* grad_target, float* grad_raylen, void* stream) {
    if (!jac || !grad_out || !grad_source || !grad_target) return fail(XVR_DRR_E_ARG, "null pointer argument");
    if (B <= 0 || n <= 0) return fail(XVR_DRR_E_ARG, "B and n must be positive");
    dim3 grid((unsigned)((n + WG - 1) / WG), (unsigned)B);
    hipLaunchKernelGGL(k_backward_from_jac, grid, dim3(WG), 0, (hipStream_t)stream, jac, grad_out, n,
                       grad_source, grad_target, grad_raylen);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(XVR_DRR_E_LAUNCH, hipGetErrorString(e));
    return XVR_DRR_OK;
}

}  // extern "C"
