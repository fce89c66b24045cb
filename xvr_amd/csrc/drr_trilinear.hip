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
// The render path's translation units (one per kernel family, linked into libxvr_drr.so):
//   drr_common.hiph   launch geometry (RenderArgs, xcd_remap, map_ray, ray_setup), trilinear taps, label lookups,
//                     split-kernel lane map, host-side checks / launch helpers / split_factor / workspace layout
//   drr_trilinear.hip this file: tri_march / tri_finish / k_trilinear_fwd, k_trilinear_fwd_split (small launches),
//                     k_trilinear_fwd_lds (opt-in, slower), k_trilinear_bwd (re-march / atomic scatter fallback)
//   drr_siddon.hip    k_siddon: forward / jacobian / backward / alpha-split
//   drr_gather.hip    voxel gradient as a gather: k_gather_prep, k_gather_cull, the brick-local splats (drr_splat.hiph), k_trilinear_gather_tab / _px, k_siddon_gather_vol2 / _cells
//   drr_rays.hip      k_backward_from_jac, k_rays_fwd, k_rays_bwd, k_jac_to_cam
//   drr_api.hip       ABI version, error text
//
// Semantics are those of oracle/diffdrr_restated.py (the restated diffdrr==0.6.0 algorithm; every
// unpinned constant arrives through xvr_drr_spec).  Reference call sites being replaced:
//   /root/reference/src/xvr/model/trainer.py:288   drr.renderer(volume, source, target, img, mask=seg)
//   /root/reference/src/xvr/registrar/base.py:249,252   reg() ... loss.backward()
#include "drr_common.hiph"
#include <type_traits>

namespace {

// =============================================================================================
// trilinear forward (+ optional per-ray jacobian in the same sweep)
// =============================================================================================
struct TriAcc {  // per-lane sums of one ray (or of one slice of its samples)
    float S, G[3], H[3], E0, E1;
    unsigned cnt;
};

// Samples kbeg..kend (wave-uniform bounds; lanes mask themselves with their own K) of the lane's ray.
// MASK: 0 = one channel; 1 = labels from a separate mask volume; 2 = labels packed into the volume's taps
// SLAB: only the samples whose index coordinate along `slab.axis` lies in [slab.lo, slab.hi) count (k_trilinear_fwd_slab: the
// slabs partition the samples exactly, because the test is made on the very coordinate the sample is taken at).
struct SlabRange {
    int axis;
    float lo, hi;
};

// WIN (clip_to_volume == 2, with the jacobian): also E1 = sum (alpha_k - A) (a d . grad V) -- d out / d (window width) needs it
// directly; rebuilt from G and H it is a difference of two large sums and loses 2 % in float32
#ifndef XVR_FWD_MASK_REGS   // 0: per-lane LDS accumulators for any C (the product); 1: up to eight label channels summed in registers --
                            // round 6 A/B at C5 (tools/ab_mask_regs.sh): 6.47 against 5.87 ms (forward), 7.21 against 6.63 (+ jacobian):
                            // the eight predicated adds per sample cost more issue slots than the LDS round trip costs latency
#define XVR_FWD_MASK_REGS 0
#endif
template <bool JAC, int MASK, bool CLIP, int YP = 0, bool SLAB = false, bool WIN = false, int SYNC = 0>
__device__ __forceinline__ void tri_march(const RenderArgs& A, const Ray& R, const KRange K, const int kbeg, const int kend,
                                          const float step, const SpecWin Wn, float* lds, const int tid, TriAcc& acc,
                                          const SlabRange slab = SlabRange{0, 0.f, 0.f}) {
    const int N = A.sp.n_points;
    const float* __restrict__ vol = A.volume;
    const int D0 = A.D0, D1 = A.D1, D2 = A.D2;
    float S = 0.f;
    float G[3] = {0.f, 0.f, 0.f}, H[3] = {0.f, 0.f, 0.f};
    float E0 = 0.f, E1 = 0.f;
    unsigned cnt = 0;
#if XVR_FWD_MASK_REGS
    const bool ch_regs = MASK && A.C <= 8;   // (uniform)
    float ch[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#endif
    const float adx = A.sp.a[0] * R.d[0], ady = A.sp.a[1] * R.d[1], adz = A.sp.a[2] * R.d[2];

    // Two steps per trip: the 8 independent 8-byte gathers of both samples are issued before either
    // is consumed (the march is latency-bound, not bandwidth-bound: L2 at ~20 %, HBM at ~25 %).
    for (int kk = kbeg; kk <= kend; kk += 2) {
        // SYNC: the wavefronts of a workgroup march a common step range and meet every SYNC trips (XVR_FWD_SYNC below)
        if (SYNC && (((kk - kbeg) >> 1) % SYNC) == 0) __builtin_amdgcn_s_barrier();
        bool act[2];
        float u[2], al[2], pxs[2], pys[2], pzs[2];
        bool inside = true;   // every active sample of this lane has all eight taps inside the volume
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = kk + h;
            act[h] = k >= K.lo && k <= K.hi && k <= kend;
            u[h] = linspace_at(k, N, Wn.near_, Wn.far_, step);
            al[h] = CLIP ? fmaf(u[h], R.amax - R.amin, R.amin) : u[h];
            pxs[h] = fmaf(A.sp.a[0], fmaf(al[h], R.d[0], R.s[0]), A.sp.b[0]);
            pys[h] = fmaf(A.sp.a[1], fmaf(al[h], R.d[1], R.s[1]), A.sp.b[1]);
            pzs[h] = fmaf(A.sp.a[2], fmaf(al[h], R.d[2], R.s[2]), A.sp.b[2]);
            if (SLAB) {
                const float pa = slab.axis == 0 ? pxs[h] : (slab.axis == 1 ? pys[h] : pzs[h]);
                act[h] = act[h] && pa >= slab.lo && pa < slab.hi;
            }
            const bool in = (unsigned)(int)floorf(pxs[h]) < (unsigned)(D0 - 1) && (unsigned)(int)floorf(pys[h]) < (unsigned)(D1 - 1) &&
                            (unsigned)(int)floorf(pzs[h]) < (unsigned)(D2 - 1);
            inside = inside && (in || !act[h]);
        }
        // INTERIOR (round 3): when every active sample of the wavefront has its eight taps inside the volume -- nine trips in ten --
        // the taps need no bounds flags, clamped offsets or shifted z pairs: the weights are (1 - t, t), the derivative weights
        // -+1 (which fold into the arithmetic), the offsets one multiply-add chain.  Same values, bit for bit, ~35 of the ~116
        // instructions of a sample fewer.  (wavefront-uniform branch: no exec masking around the loads)
        auto trip = [&](auto interior_c) {
        constexpr bool INTERIOR = decltype(interior_c)::value;
        Taps T[2];
        fpair P[2][4];
        int yoff[2][2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (INTERIOR) {
                make_taps_interior<YP>(pxs[h], pys[h], pzs[h], D1, D2, T[h], yoff[h]);
                if (!act[h]) {   // (an idle lane's position can be anywhere: load element 0)
                    yoff[h][0] = yoff[h][1] = 0;
                    T[h].base[0] = T[h].base[1] = T[h].base[2] = T[h].base[3] = 0;
                }
            }
            else if (YP) make_taps_yp<YP>(pxs[h], pys[h], pzs[h], D0, D1, D2, T[h], yoff[h]);
            else make_taps(pxs[h], pys[h], pzs[h], D0, D1, D2, T[h]);  // offsets are clamped: always loadable
        }
        // unconditional (offsets are clamped into the volume): a branch here would split the loads into
        // two exec-masked blocks with a full vmcnt(0) drain between them
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (YP) {
                // y-pair interleaved copy: two 16-byte loads instead of four 8-byte ones; re-filed as the four (z, z + 1)
                // pairs of rows (x0,y0) (x0,y1) (x1,y0) (x1,y1), so that everything below is the same arithmetic
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const fquad Q = load_quad(vol + yoff[h][q]);
                    P[h][2 * q] = fpair{Q.x, Q.z};
                    P[h][2 * q + 1] = fpair{Q.y, Q.w};
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) P[h][q] = load_pair(vol + T[h].base[q]);
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (!act[h]) continue;
            const Taps& t = T[h];
            int lab = 0;
            if (MASK == 2) {
                // the label first, then the label bits are cleared from the taps: a voxel of density exactly 0 (all of
                // the air after transform_hu_to_density) must interpolate to exactly 0, as in the reference, not to a
                // denormal L * 2^-149 that would flip xvr's `img > 0` foreground test (trainer.py:292-302)
                lab = packed_label<INTERIOR>(P[h], pxs[h], pys[h], pzs[h], D0, D1, D2, A.C);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    P[h][q].x = __uint_as_float(__float_as_uint(P[h][q].x) & ~LABEL_MASK);
                    P[h][q].y = __uint_as_float(__float_as_uint(P[h][q].y) & ~LABEL_MASK);
                }
            } else if (MASK == 1) {
                lab = nearest_label(A.mask, pxs[h], pys[h], pzs[h], D0, D1, D2, A.C);
            }
            const float v0 = fmaf(t.pz1, P[h][0].y, t.pz0 * P[h][0].x), v1 = fmaf(t.pz1, P[h][1].y, t.pz0 * P[h][1].x);
            const float v2 = fmaf(t.pz1, P[h][2].y, t.pz0 * P[h][2].x), v3 = fmaf(t.pz1, P[h][3].y, t.pz0 * P[h][3].x);
            const float r0 = fmaf(t.wy1, v1, t.wy0 * v0), r1 = fmaf(t.wy1, v3, t.wy0 * v2);
            const float v = fmaf(t.wx1, r1, t.wx0 * r0);
            ++cnt;
            if (MASK) {
#if XVR_FWD_MASK_REGS
                if (ch_regs) {
                    // (diagnostic build) up to eight channels: the sums stay in registers -- eight predicated adds instead of an LDS
                    // read-modify-write
#pragma unroll
                    for (int c = 0; c < 8; ++c) ch[c] += lab == c ? v : 0.f;
                } else
#endif
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
                if (WIN) E1 = fmaf(fmaf(gx, adx, fmaf(gy, ady, gz * adz)), al[h] - Wn.A, E1);
            }
        }
            };
        if (__builtin_amdgcn_ballot_w64(inside) == ~0ull) trip(std::true_type{});
        else trip(std::false_type{});
    }
#if XVR_FWD_MASK_REGS
    if (MASK && ch_regs) {
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (c < A.C) lds[c * WG + tid] += ch[c];
    }
#endif
    acc.S = S;
    acc.cnt = cnt;
#pragma unroll
    for (int i = 0; i < 3; ++i) { acc.G[i] = G[i]; acc.H[i] = H[i]; }
    acc.E0 = E0;
    acc.E1 = E1;
}

// Scale the sums and write the pixel (and its jacobian row).
template <bool JAC, int MASK, bool CLIP, bool WIN = false>
__device__ __forceinline__ void tri_finish(const RenderArgs& A, const Ray& R, const int b, const int r, const SpecWin Wn, const float* lds,
                                           const int tid, const TriAcc& acc) {
    const float S = acc.S;
    const float span = fmaxf(R.amax - R.amin, 0.f);
    const float base_scale = R.L * Wn.inv_denom;
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
        jp[0] = make_float4(S * (CLIP ? Wn.inv_denom * span : Wn.inv_denom), js[0], js[1], js[2]);
        // (WIN: the spare float carries scale * E1 = the ray's  sum_k (alpha_k - A) d out / d alpha_k  for xvr_drr_alpha_window_backward)
        jp[1] = make_float4(jt[0], jt[1], jt[2], WIN ? scale * acc.E1 : 0.f);
    }
}

// At most 4 wavefronts per SIMD: the march is bound by the texture-address unit, not by latency hiding, and
// more resident wavefronts only thrash the L1/L2 -- the variant without the jacobian needs 64 VGPRs, ran at
// 8 wavefronts per SIMD and took 8.2 ms where the (heavier) jacobian variant at 6 took 7.0; capped, both take
// ~7.0 ms (measured flat from 3 to 6, worse at 2 and at 8).
constexpr double SLAB_TARGET_BYTES = 150e6, SLAB_MIN_VOLUME_BYTES = 192.0 * (1 << 20);   // (the Infinity Cache holds 256 MiB)
// XVR_FWD_SYNC (round 3): the four wavefronts of a workgroup march a COMMON step range and meet at a barrier every SYNC trips
// (a trip = two samples), so that the 8x8 patches of one 16x16 tile touch the cache lines they share while those are in the
// L1.  C2, forward + jacobian: 6.04 ms without, 5.69 at 1, 5.69 at 4, 5.79 at 16 (profiles/r03_forward_sync.txt); two or four
// tiles per workgroup of 512 / 1024 threads in lockstep: 5.83 / 7.90 (one workgroup per CU leaves nothing to overlap).
#ifndef XVR_FWD_SYNC
#define XVR_FWD_SYNC 1
#endif
#ifndef XVR_FWD_WAVES   // (overridable for tuning builds)
#define XVR_FWD_WAVES 4
#endif
template <bool JAC, int MASK, bool CLIP, int YP = 0, bool WIN = false>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(1, XVR_FWD_WAVES))) void k_trilinear_fwd(RenderArgs A) {
    extern __shared__ float lds[];  // MASK: per-lane channel accumulators [C][WG]
    int b, r;
    const bool valid = map_ray(A, b, r, threadIdx.x);
    const int tid = threadIdx.x;
    Ray R;
    ray_setup(A, b, r, valid, R);
    const int N = A.sp.n_points;
    const SpecWin Wn = spec_window(A.sp);
    const float step = N > 1 ? (Wn.far_ - Wn.near_) / (float)(N - 1) : 0.f;
    const KRange K = tri_krange(A, R, CLIP, step, Wn.near_);
    int kbeg = __builtin_amdgcn_readfirstlane(wave_min_i(K.lo));
    int kend = __builtin_amdgcn_readfirstlane(wave_max_i(K.hi));
#if XVR_FWD_SYNC
    __shared__ int s_k[2];
    if (tid == 0) { s_k[0] = 0x7fffffff; s_k[1] = -0x7fffffff; }
    __syncthreads();
    if ((tid & 63) == 0) { atomicMin(&s_k[0], kbeg); atomicMax(&s_k[1], kend); }
    __syncthreads();
    kbeg = __builtin_amdgcn_readfirstlane(s_k[0]);
    kend = __builtin_amdgcn_readfirstlane(s_k[1]);
#endif
    if (MASK) {
        for (int c = 0; c < A.C; ++c) lds[c * WG + tid] = 0.f;
    }
    TriAcc acc;
    tri_march<JAC, MASK, CLIP, YP, false, WIN, XVR_FWD_SYNC>(A, R, K, kbeg, kend, step, Wn, lds, tid, acc);
    if (valid) tri_finish<JAC, MASK, CLIP, WIN>(A, R, b, r, Wn, lds, tid, acc);
    if (A.work) {
        unsigned tot = wave_sum_u(acc.cnt);
        if ((tid & 63) == 0 && tot) atomicAdd(A.work, (unsigned long long)tot);
    }
}

// ---------------------------------------------------------------------------------------------
// Slab-major forward for LARGE batches over a volume that does not fit the 256 MiB Infinity Cache (round 3).
//
// Every pose of a batch fetches its own frustum -- ~15 % of the volume -- and the frusta of a batch are as different as
// its poses: 116 poses pull 116 x 160 MB of the 1 GiB y-pair copy across the fabric, almost all of it from HBM.  The
// same kernel over 116 copies of ONE pose (the whole launch's footprint then sits in the Infinity Cache) takes 3.9 ms
// instead of 6.1 (tools/exp_forward_variants.py).  So the march is cut into `nslabs` launches: launch s takes, of EVERY ray
// of EVERY pose, the samples whose index coordinate along `axis` falls into slab s -- a slice of the volume small enough
// for the Infinity Cache (<= ~150 MB), which all poses then share: each slab comes from HBM once per batch instead of
// once per pose.  The running sums of a ray (S, and with the jacobian G[3], H[3]) travel from launch to launch in the
// buffers the results end up in: `out` (one float per ray) / `jac` (8 floats per ray, raw sums until the last launch
// scales them).  Only wavefronts that have samples in the slab touch them (the first launch initialises, the last
// finalises), and a wavefront's rays stay in ~2-4 of the 8 slabs when the slabs run ALONG the rays rather than across.
// Launches are stream-ordered, so a ray's sums are added in slab order: deterministic, but a different rounding than
// the one-launch march's single running sum (same tolerance against the oracle; the y-pair / natural layouts of THIS
// kernel are bit-identical to each other).
// ---------------------------------------------------------------------------------------------
template <bool JAC, bool YP>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(1, XVR_FWD_WAVES))) void k_trilinear_fwd_slab(RenderArgs A, int slab, int nslabs,
                                                                                                           SlabRange S) {
    int b, r;
    const bool valid = map_ray(A, b, r, threadIdx.x);
    const int tid = threadIdx.x;
    Ray R;
    ray_setup(A, b, r, valid, R);
    const int N = A.sp.n_points;
    const SpecWin Wn = spec_window(A.sp);
    const float step = N > 1 ? (Wn.far_ - Wn.near_) / (float)(N - 1) : 0.f;
    KRange K = tri_krange(A, R, false, step, Wn.near_);
    // the lane's steps inside the slab, from the linear model p(k) = P0 + k Dl of its coordinate along the slab axis, two steps
    // of slack on either side (the march's own test on the computed coordinate decides)
    {
        const int ax = S.axis;
        const float a = ax == 0 ? A.sp.a[0] : (ax == 1 ? A.sp.a[1] : A.sp.a[2]), bb = ax == 0 ? A.sp.b[0] : (ax == 1 ? A.sp.b[1] : A.sp.b[2]);
        const float d = ax == 0 ? R.d[0] : (ax == 1 ? R.d[1] : R.d[2]), s0 = ax == 0 ? R.s[0] : (ax == 1 ? R.s[1] : R.s[2]);
        const float P0 = fmaf(a, fmaf(Wn.near_, d, s0), bb), Dl = step * a * d;
        if (K.lo <= K.hi) {
            if (fabsf(Dl) > 1e-12f) {
                const float inv = 1.f / Dl;
                const float t0 = (S.lo - P0) * inv, t1 = (S.hi - P0) * inv;   // (+-inf at the outer slabs: clamped below)
                const float tlo = fminf(t0, t1), thi = fmaxf(t0, t1);
                const float flo = fmaxf(tlo - 2.f, (float)K.lo), fhi = fminf(thi + 2.f, (float)K.hi);
                if (flo <= fhi) { K.lo = (int)ceilf(flo); K.hi = (int)floorf(fhi); }
                else { K.lo = INT32_MAX; K.hi = INT32_MIN; }
            } else if (!(P0 >= S.lo - 1.f && P0 < S.hi + 1.f)) {
                K.lo = INT32_MAX; K.hi = INT32_MIN;
            }
        }
    }
    const int kbeg = __builtin_amdgcn_readfirstlane(wave_min_i(K.lo));
    const int kend = __builtin_amdgcn_readfirstlane(wave_max_i(K.hi));
    const bool first = slab == 0, last = slab == nslabs - 1;
    const bool active = kbeg <= kend;            // (wave-uniform: a wavefront owns its 64 rays' sums)
    if (!active && !first && !last) return;
    TriAcc acc;
    acc.S = 0.f; acc.cnt = 0; acc.E0 = acc.E1 = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) acc.G[i] = acc.H[i] = 0.f;
    if (active) tri_march<JAC, 0, false, YP ? 1 : 0, true>(A, R, K, kbeg, kend, step, Wn, nullptr, tid, acc, S);
    if (valid) {
        float4* jp = JAC ? reinterpret_cast<float4*>(A.jac + ((size_t)b * A.n + r) * XVR_DRR_JAC_STRIDE) : nullptr;
        float* op = A.out + (size_t)b * A.n + r;
        if (!first) {   // what the earlier slabs left: added in slab order
            if (JAC) {
                const float4 p0 = jp[0], p1 = jp[1];
                acc.S = p0.x + acc.S;
                acc.G[0] = p0.y + acc.G[0]; acc.G[1] = p0.z + acc.G[1]; acc.G[2] = p0.w + acc.G[2];
                acc.H[0] = p1.x + acc.H[0]; acc.H[1] = p1.y + acc.H[1]; acc.H[2] = p1.z + acc.H[2];
            } else {
                acc.S = *op + acc.S;
            }
        }
        if (last) {
            tri_finish<JAC, 0, false>(A, R, b, r, Wn, nullptr, tid, acc);
        } else if (JAC) {
            jp[0] = make_float4(acc.S, acc.G[0], acc.G[1], acc.G[2]);
            jp[1] = make_float4(acc.H[0], acc.H[1], acc.H[2], 0.f);
        } else {
            *op = acc.S;
        }
    }
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
    const SpecWin Wn = spec_window(A.sp);
    const float step = N > 1 ? (Wn.far_ - Wn.near_) / (float)(N - 1) : 0.f;
    const KRange K = tri_krange(A, R, CLIP, step, Wn.near_);
    const int kbeg = __builtin_amdgcn_readfirstlane(wave_min_i(K.lo));
    const int kend = __builtin_amdgcn_readfirstlane(wave_max_i(K.hi));
    TriAcc acc;
    {
        const int len = kend >= kbeg ? kend - kbeg + 1 : 0;
        const int chunk = (((len + NS - 1) / NS) + 1) & ~1;   // even: the march takes two samples per trip
        const int my_beg = kbeg + w * chunk;
        const int my_end = min(kend, my_beg + chunk - 1);
        tri_march<JAC, 0, CLIP>(A, R, K, my_beg, my_end, step, Wn, nullptr, tid, acc);
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
        if (valid) tri_finish<JAC, 0, CLIP>(A, R, b, r, Wn, nullptr, tid, acc);
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
    const KRange K = tri_krange(A, R, false, step, A.sp.near_);
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
    const SpecWin Wn = spec_window(A.sp);
    const float step = N > 1 ? (Wn.far_ - Wn.near_) / (float)(N - 1) : 0.f;
    const KRange K = tri_krange(A, R, CLIP, step, Wn.near_);
    const int kbeg = __builtin_amdgcn_readfirstlane(wave_min_i(K.lo));
    const int kend = __builtin_amdgcn_readfirstlane(wave_max_i(K.hi));
    const float span = fmaxf(R.amax - R.amin, 0.f);
    const float* __restrict__ vol = A.volume;
    const int D0 = A.D0, D1 = A.D1, D2 = A.D2;
    const float base_scale = R.L * Wn.inv_denom;
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
        const float u = linspace_at(k, N, Wn.near_, Wn.far_, step);
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
            if (A.glen) A.glen[(size_t)b * A.n + r] = SV * (CLIP ? Wn.inv_denom * span : Wn.inv_denom);
        }
        // grad_source is shared by all rays of the pose: wave butterfly, then one atomic per wave
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float tot = wave_sum_f(js[i]);
            if ((tid & 63) == 0 && tot != 0.f) atomic_add_f32(A.gsrc + 3 * b + i, tot);
        }
    }
}


}  // namespace

extern "C" {

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
    if (sp->alpha_window && jac && (mask || packed))
        return fail(XVR_DRR_E_UNSUPPORTED, "clip_to_volume == 2: the jacobian is implemented for one channel");
    if (sp->volume_layout != 0 && sp->volume_layout != 1 && sp->volume_layout != 3) return fail(XVR_DRR_E_ARG, "unknown volume_layout");
    const bool ypl = sp->volume_layout == 1 || sp->volume_layout == 3;   // a y-pair copy: rows (1) or 2 x 8 tiles (3)
    if (ypl && mask) return fail(XVR_DRR_E_UNSUPPORTED, "the y-pair layouts take labels packed into the volume, not a mask volume");
    if (sp->volume_layout == 1 && (long long)D0 * (D1 + 1) * D2 * 2 >= (1LL << 31))
        return fail(XVR_DRR_E_UNSUPPORTED, "y-pair copy has >= 2^31 elements");
    if (sp->volume_layout == 3 && ((long long)((D0 + 1) / 2) * (D1 + 1) * ((D2 - 2) / 7 + 1) * 32 >= (1LL << 31) || D2 >= 8192))
        return fail(XVR_DRR_E_UNSUPPORTED, "tiled y-pair copy has >= 2^31 elements");
    RenderArgs A;
    fill_args(A, volume, mask, D0, D1, D2, C, source, target, raylen, B, n, sp, cam);
    A.out = out; A.jac = jac; A.work = work;
    if (sp->clip_to_volume < 0 || sp->clip_to_volume > 2) return fail(XVR_DRR_E_ARG, "clip_to_volume must be 0, 1 or 2");
    if ((sp->clip_to_volume == 2) != (sp->alpha_window != nullptr))
        return fail(XVR_DRR_E_ARG, "clip_to_volume == 2 needs spec.alpha_window (xvr_drr_alpha_window), and only it");
    const bool clip = sp->clip_to_volume == 1;
    const size_t lds = C > 1 || mask ? (size_t)C * WG * sizeof(float) : 0;
#define XVR_YP_LAUNCH(M, Y, L)                                                                                           \
    do {                                                                                                                \
        if (jac) return clip ? launch(k_trilinear_fwd<true, M, true, Y>, A, L, stream) : launch(k_trilinear_fwd<true, M, false, Y>, A, L, stream); \
        return clip ? launch(k_trilinear_fwd<false, M, true, Y>, A, L, stream) : launch(k_trilinear_fwd<false, M, false, Y>, A, L, stream);        \
    } while (0)
    if (packed && sp->volume_layout == 1) XVR_YP_LAUNCH(2, 1, lds);   // the y-pair copy of the label-carrying volume
    if (packed && sp->volume_layout == 3) XVR_YP_LAUNCH(2, 2, lds);
    if (packed && jac) return clip ? launch(k_trilinear_fwd<true, 2, true>, A, lds, stream)
                                   : launch(k_trilinear_fwd<true, 2, false>, A, lds, stream);
    if (packed) return clip ? launch(k_trilinear_fwd<false, 2, true>, A, lds, stream)
                            : launch(k_trilinear_fwd<false, 2, false>, A, lds, stream);
    if (mask && jac) return clip ? launch(k_trilinear_fwd<true, 1, true>, A, lds, stream)
                                 : launch(k_trilinear_fwd<true, 1, false>, A, lds, stream);
    if (mask) return clip ? launch(k_trilinear_fwd<false, 1, true>, A, lds, stream)
                          : launch(k_trilinear_fwd<false, 1, false>, A, lds, stream);
    // LDS-staged bricks are opt-in (option "fwd_lds"; natural layout only): measured 2.25x SLOWER than the direct kernel at C2 (19.8 vs 8.8 ms;
    // with ~4 taps per voxel the L1/L2 already capture the reuse, HISTORY.md section 4.2)
    const bool use_lds = xvr_detail::option(xvr_detail::OPT_FWD_LDS) == 1 && sp->volume_layout == 0 && !sp->alpha_window;
    if (use_lds && !clip && A.grid_w > 0) {
        const size_t bytes = (size_t)(LDS_HDR + LDS_BRICK_CAP) * sizeof(float);
        return jac ? launch(k_trilinear_fwd_lds<true>, A, bytes, stream) : launch(k_trilinear_fwd_lds<false>, A, bytes, stream);
    }
    if (sp->alpha_window && jac) {   // the jacobian of a windowed render also carries d out / d (window width): unsplit kernel
        RenderArgs Aw = A;
        return sp->volume_layout == 1 ? launch(k_trilinear_fwd<true, 0, false, 1, true>, Aw, 0, stream)
                                      : (sp->volume_layout == 3 ? launch(k_trilinear_fwd<true, 0, false, 2, true>, Aw, 0, stream)
                                                                : launch(k_trilinear_fwd<true, 0, false, 0, true>, Aw, 0, stream));
    }
    // Large batches over a volume the Infinity Cache cannot hold: the slab-major march (k_trilinear_fwd_slab).  Option
    // "fwd_slabs": 0 = never, n >= 2 = always n slabs, -1 (default) = as many slabs as keep a slab's bytes under
    // SLAB_TARGET_BYTES, for launches of >= 8192 workgroups over a volume copy of > 192 MiB.  "fwd_slab_axis": 0-2.
    {
        const int want = xvr_detail::option(xvr_detail::OPT_FWD_SLABS);
        const double layout_bytes = (ypl ? 2.0 * D0 * (D1 + 1) * D2 : 1.0 * D0 * D1 * D2) * sizeof(float);
        const long long nblocks = (long long)B * A.blocks_per_pose;
        int nslabs = want;
        if (want < 0) nslabs = (nblocks >= 8192 && layout_bytes > SLAB_MIN_VOLUME_BYTES) ? (int)ceil(layout_bytes / SLAB_TARGET_BYTES) : 0;
        const int axis = xvr_detail::option(xvr_detail::OPT_FWD_SLAB_AXIS);
        const int Daxis = axis == 0 ? D0 : (axis == 1 ? D1 : D2);
        if (nslabs > Daxis / 4) nslabs = Daxis / 4;
        if (nslabs >= 2 && !clip && A.grid_w > 0 && sp->volume_layout != 3) {
            for (int s = 0; s < nslabs; ++s) {
                SlabRange S;
                S.axis = axis;
                // slab boundaries on whole voxels of the index coordinate; the outer slabs are open-ended (samples in the
                // zero-padding margin belong to them)
                S.lo = s == 0 ? -INFINITY : (float)(((long long)Daxis * s) / nslabs);
                S.hi = s == nslabs - 1 ? INFINITY : (float)(((long long)Daxis * (s + 1)) / nslabs);
                const long long nb = (long long)A.B * A.blocks_per_pose;
                if (nb >= (1LL << 31)) return fail(XVR_DRR_E_UNSUPPORTED, "grid too large");
#define XVR_SLAB_LAUNCH(J, Y) hipLaunchKernelGGL((k_trilinear_fwd_slab<J, Y>), dim3((unsigned)nb), dim3(WG), 0, (hipStream_t)stream, A, s, nslabs, S)
                if (sp->volume_layout == 1) { if (jac) XVR_SLAB_LAUNCH(true, true); else XVR_SLAB_LAUNCH(false, true); }
                else { if (jac) XVR_SLAB_LAUNCH(true, false); else XVR_SLAB_LAUNCH(false, false); }
#undef XVR_SLAB_LAUNCH
                hipError_t e = hipGetLastError();
                if (e != hipSuccess) return fail(XVR_DRR_E_LAUNCH, hipGetErrorString(e));
            }
            return XVR_DRR_OK;
        }
    }
    if (sp->volume_layout == 1) XVR_YP_LAUNCH(0, 1, 0);   // `volume` is a y-pair copy: the unsplit kernel, whatever the launch size
    if (sp->volume_layout == 3) XVR_YP_LAUNCH(0, 2, 0);
#undef XVR_YP_LAUNCH
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
    if (sp->volume_layout != 0) return fail(XVR_DRR_E_ARG, "the backward takes the natural volume layout");
    if (!mask && C != 1) return fail(XVR_DRR_E_ARG, "C must be 1 without a mask");
    if ((grad_source == nullptr) != (grad_target == nullptr))
        return fail(XVR_DRR_E_ARG, "grad_source and grad_target must be requested together");
    if (grad_raylen && !grad_target) return fail(XVR_DRR_E_ARG, "grad_raylen needs grad_source/grad_target");
    const bool gpose = grad_target != nullptr, gvol = grad_volume != nullptr;
    if (!gpose && !gvol) return XVR_DRR_OK;
    RenderArgs A;
    fill_args(A, volume, mask, D0, D1, D2, C, source, target, raylen, B, n, sp);
    A.gout = grad_out; A.gvol = grad_volume; A.gsrc = grad_source; A.gtgt = grad_target; A.glen = grad_raylen;
    if ((sp->clip_to_volume == 2) != (sp->alpha_window != nullptr))
        return fail(XVR_DRR_E_ARG, "clip_to_volume == 2 needs spec.alpha_window (xvr_drr_alpha_window), and only it");
    if (sp->clip_to_volume == 2 && (mask || grad_target))
        return fail(XVR_DRR_E_UNSUPPORTED, "the alpha window's pose gradient comes from the jacobian (xvr_drr_backward_from_jac + xvr_drr_alpha_window_backward); one channel");
    const bool clip = sp->clip_to_volume == 1;
    const size_t lds = mask ? (size_t)C * WG * sizeof(float) : 0;

    // Voxel gradient by the atomic-free voxel-driven gather when the rays are a detector lattice: the per-lane flattened
    // table kernel for the plain render, the pixel-major kernel under clip_to_volume and / or a mask; the scatter kernel
    // stays as the general fallback and is launched right behind it, reading the lattice flag on the device (no host sync).
    const bool gather = gvol && gather_usable(sp, n, workspace, workspace_bytes, B, D0, D1, D2);
    if (gather_slab_later() && !gather) return XVR_DRR_OK;   // (option gather_slab: the call for slab 0 did everything)
    if (gather) {
        unsigned* flag = nullptr;
        rc = launch_gather(false, source, target, raylen, grad_out, B, n, sp->ray_grid_w, D0, D1, D2, sp, grad_volume,
                           workspace, stream, &flag, mask, C);
        if (rc || gather_slab_later()) return rc;
        RenderArgs Ap = A, Av = A;
        Ap.gvol = nullptr;
        Av.gsrc = nullptr; Av.gtgt = nullptr; Av.glen = nullptr;
        Av.skip_unless_flag_gt = flag;
#define TRI_BWD_ONE(M, CL, GP, GV, ARGS) launch(k_trilinear_bwd<M, CL, GP, GV>, ARGS, lds, stream)
#define TRI_BWD_PAIR(M, CL)                                                                   \
        do {                                                                                  \
            if (gpose) { rc = TRI_BWD_ONE(M, CL, true, false, Ap); if (rc) return rc; }       \
            return TRI_BWD_ONE(M, CL, false, true, Av);                                       \
        } while (0)
        if (mask) { if (clip) TRI_BWD_PAIR(true, true); else TRI_BWD_PAIR(true, false); }
        if (clip) TRI_BWD_PAIR(false, true); else TRI_BWD_PAIR(false, false);
#undef TRI_BWD_PAIR
#undef TRI_BWD_ONE
    }
#define TRI_BWD(M, CL)                                                                         \
    (gpose ? (gvol ? launch(k_trilinear_bwd<M, CL, true, true>, A, lds, stream)                \
                   : launch(k_trilinear_bwd<M, CL, true, false>, A, lds, stream))              \
           : launch(k_trilinear_bwd<M, CL, false, true>, A, lds, stream))
    if (mask) return clip ? TRI_BWD(true, true) : TRI_BWD(true, false);
    return clip ? TRI_BWD(false, true) : TRI_BWD(false, false);
#undef TRI_BWD
}
}  // extern "C"
