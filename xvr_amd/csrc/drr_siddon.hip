// MI355X (gfx950) differentiable-DRR kernels: Siddon exact traversal -- forward (+jacobian), alpha-split forward,
// re-traversal backward (pose gradient and atomic-scatter voxel gradient).
#include "drr_common.hiph"

namespace {

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
// XVR_SID_PIPE 1: the voxel of segment i is requested, then segment i - 1 -- whose load has had a whole step to arrive -- is
// consumed (round 1).  0: every segment is consumed where it is loaded; the other wavefronts of the SIMD cover the latency
// and the walk saves the hand-over of six values per step.  (tuning builds: -DXVR_SID_PIPE=...)
// (Round 3 also tried the trilinear forward's workgroup lockstep here -- a barrier every 1 / 4 / 16 segments for as many trips as
// the tile's longest ray needs: 12.1 / 12.0 / 11.8 ms against 8.77.  The lanes of a Siddon wavefront are at different depths
// after a few segments anyway, and the idle trips of the shorter rays cost more than the shared lines save.)
#ifndef XVR_SID_PIPE
#define XVR_SID_PIPE 1
#endif
template <int MODE, int MASK, bool GPOSE, bool GVOL, bool EXACT, int SPLIT = 0, bool BRICK = false>
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
    constexpr bool bricked = BRICK;   // (forward, one channel, exact map, unsplit: the only instantiations)
    const int nby = (D1 + 1) >> 1, nbz = (D2 + 7) >> 3;
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
        // (a direction component of exactly 0 -- an axis-aligned ray under eps = 0: a huge finite reciprocal and "forward" put every
        //  plane of that axis at alpha = +huge, never selected; 1 / 0 = inf gave NaN alphas and an empty image, round 5)
        inv_d[i] = R.d[i] == 0.f ? 1e30f : 1.f / R.d[i];
        const float f = fmaf(alo, R.d[i], R.s[i]) - A.sp.plane0[i];  // position in plane-index units
        if (R.d[i] >= 0.f) { stp[i] = 1; ip[i] = (int)floorf(f) + 1; }
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
    // The jacobian: d out / d s_i = sum over the crossings of axis i of dW (alpha - 1) / d_i, d out / d t_i = sum dW (-alpha) / d_i
    // (dW = the jump of the integrand at the crossing).  1 / d_i is the same for every crossing of an axis, so the walk sums
    // U_i = sum dW alpha and M_i = sum dW per axis and scales once at the end (round 3: the walk is vector-issue bound, and this
    // is 12 instructions per segment instead of 17).
    float Uj[3] = {0.f, 0.f, 0.f}, Mj[3] = {0.f, 0.f, 0.f};
    float As[3] = {0.f, 0.f, 0.f};   // (filled from U, M after the walk)
    float At[3] = {0.f, 0.f, 0.f};
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
                const float m = (p_ax == i) ? dW : 0.f;
                Uj[i] = fmaf(m, p_ac, Uj[i]);
                Mj[i] += m;
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
        // volume_layout 2: 2 x 2 x 8 voxel bricks, one per 128-byte line (xvr_drr_pack_bricks) -- the lanes of a wavefront sit
        // at different depths of neighbouring rays, and their voxels fall into half as many lines as in [x][y][z] rows
        const int voff = !bricked ? off
                                  : (inb ? ((((ix >> 1) * nby + (iy >> 1)) * nbz + (iz >> 3)) << 5) + ((ix & 1) << 4) + ((iy & 1) << 3) + (iz & 7) : 0);
        const float v_new = vol[voff];                      // always loadable (offset 0 when outside)
        // MASK == 2: the label rides in the low mantissa bits of the voxel just loaded (xvr_drr_pack_labels)
        const float lab_new = MASK == 2 ? (float)(__float_as_uint(v_new) & LABEL_MASK) : (MASK ? A.mask[off] : 0.f);
        if (inb && !first_of_slice && (!SPLIT || an > ac)) ++cnt;
        first_of_slice = false;
#if XVR_SID_PIPE
        if (have) consume();
#endif
        // (packed labels: the label bits are cleared from the value -- a voxel of density exactly 0 contributes exactly 0)
        p_v = inb ? (MASK == 2 ? __uint_as_float(__float_as_uint(v_new) & ~LABEL_MASK) : v_new) : 0.f; p_seg = an - ac; p_ac = ac; p_ax = ax_prev; p_off = off; p_inb = inb; p_lab = lab_new;
        have = true;
#if !XVR_SID_PIPE
        consume();   // (no software pipeline: see XVR_SID_PIPE)
        have = false;
#endif
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
            const float m = (R.ax_out == i) ? W : 0.f;
            Uj[i] = fmaf(m, ahi, Uj[i]);
            Mj[i] += m;
        }
    }
    if (DERIV) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            As[i] = inv_d[i] * (Uj[i] - Mj[i]);
            At[i] = -inv_d[i] * Uj[i];
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
// Siddon forward (+ jacobian) as a march over the unit slabs of every ray's DOMINANT axis (round 4).
//
// The merge walk above spends ~77 vector instructions per voxel segment: three plane alphas recomputed per segment, a four-way
// minimum, three compares and selects for the axis, bounds tests, a cvt chain per axis -- and it is vector-issue bound (0.93 of
// the launch, profiles/r03_bench_final_siddon.json).  Here a trip of the loop is one unit slab between two consecutive planes of
// the axis m along which the ray moves fastest.  Inside a slab the ray advances by at most one voxel along either of the other
// two axes (|d_u|, |d_v| <= |d_m|), so the slab holds at most three voxel segments, in a FIXED body:
//
//     [ac, lo] in voxel (iu, iv) . [lo, hi] behind the first minor crossing . [hi, a_end] behind both,   lo <= hi the two minor
//     crossing alphas clamped into the slab (med3): a crossing that is not in this slab gives a zero-length segment.
//
// No minimum search, no axis select chain, no per-segment bounds test: the voxel offset moves by constants (+-stride of the
// crossed axis), the walk starts inside the volume and can only cross INTERIOR planes, because the far boundary plane of every
// axis bounds a_hi by the very arithmetic the loop evaluates plane alphas with (alpha(p) = ((float)p + (plane0 - s)) / d is a pure
// function of the plane index: monotone along the ray, the same bits inside and outside the loop).
// (Round 4, measured and not kept: alpha(p) as one fma(p, 1 / d, (plane0 - s) / d) -- three vector instructions fewer per slab of
//  ~70, 5.86 against 5.89 ms, and the image then differs from the merge walk's by 2.8e-5 instead of 1e-6: the march does not
//  wait on the vector ALU alone.)  The three loads of a slab are
// independent and predicated on their segment's length; values of zero-length segments repeat their predecessor's, which makes
// the jumps the jacobian sums (U_i = sum dW alpha, M_i = sum dW per axis, as in the merge walk) vanish where nothing is crossed.
// ~50 vector instructions per slab for 1.8 segments.  The roles (m, u, v) are per-lane data, not template parameters: strides,
// reciprocals and plane counters live in registers either way, so a wavefront whose rays disagree on the dominant axis costs
// nothing extra.  Serves the unsplit one-channel forward with the exact index map on the natural layout (option siddon_slab,
// default 1); masks, non-exact maps, the alpha-split small launches and the re-marching backward keep the merge walk.
// =============================================================================================
#ifndef XVR_SLAB_WAVES
#define XVR_SLAB_WAVES 8
#endif
__device__ __forceinline__ float sel3f(int k, float a, float b, float c) { return k == 0 ? a : (k == 1 ? b : c); }
__device__ __forceinline__ int sel3i(int k, int a, int b, int c) { return k == 0 ? a : (k == 1 ? b : c); }
__device__ __forceinline__ float med3f(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }

// BRICK: the volume is the 2 x 2 x 8-bricked copy (xvr_drr_pack_bricks, one brick per 128-byte line; 4 x 2 x 4 until late in round 4:
// 5.03 against 4.84 ms -- the texture-address unit pays per line a wavefront load touches, and the longer z-run means fewer of them).  On [x][y][z] rows a
// wavefront's 64 voxels of one slab sit in ~6 rows of which it uses a fifth, and the next slab (the usual dominant axis is y) is
// another set of rows: 5.3e8 L1-miss lines per C3 launch (67 GB), 25 GB from the fabric.  A brick holds two slabs of a 4 x 4 patch.
// The brick offset is separable, offset = fx(ix) + fy(iy) + fz(iz), so every axis keeps its own partial offset and replaces it
// when its plane is crossed; the partial offsets come from three small tables in LDS (built by the workgroup at the start) -- a
// ds_read per axis and slab, issued one slab ahead, instead of ~5 vector instructions of shifts and multiplies each.
// NX (round 5): a NON-exact index map (norm_dims_offset = +-1, align_corners: what SURVEY.md Appendix A recalls for upstream) on the same
// march.  The planes a ray crosses are the same -- they are geometry -- only the voxel a segment is credited with is the nearest voxel
// of its MIDPOINT under index = a x + b, as the merge walk's EXACT = false branch looks it up.  The march keeps its alpha bookkeeping
// and drops the incremental offsets: every segment's voxel is one fma per axis on the sum of its end alphas (0.5 a d folded into the
// coefficient), v_cvt_rpi, a clamp into the volume (a midpoint an ulp outside the entry face must not wrap into the next row; the
// host admits only maps that stay inside the volume for points inside it, siddon_map_in_bounds), and the row strides -- or, on
// bricks, the three LDS tables read by index instead of one slab ahead.  ~32 vector instructions per slab on top of the exact
// march's; the merge walk it replaces for these maps spends ~90 per SEGMENT.
template <bool JAC, bool BRICK, bool NX = false>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(1, XVR_SLAB_WAVES))) void k_siddon_slab(RenderArgs A) {
    extern __shared__ unsigned slab_tab[];   // BRICK: byte offsets fx[-1 .. D0], fy[-1 .. D1], fz[-1 .. D2] (indices clamped)
    int b, r;
    const bool valid = map_ray(A, b, r, threadIdx.x);
    Ray R;
    ray_setup(A, b, r, valid, R);
    const float* __restrict__ vol = A.volume;
    const int D[3] = {A.D0, A.D1, A.D2};
    const int nby = (A.D1 + 1) >> 1, nbz = (A.D2 + 7) >> 3;
    const int tbase[3] = {1, A.D0 + 3, A.D0 + A.D1 + 5};                     // entry of index 0 of every axis' table
    if (BRICK) {
        const int total = A.D0 + A.D1 + A.D2 + 6;
        for (int e = threadIdx.x; e < total; e += WG) {
            const int ax = e < A.D0 + 2 ? 0 : (e < A.D0 + A.D1 + 4 ? 1 : 2);
            const int ic = min(max(e - tbase[ax], 0), D[ax] - 1);
            const int f = ax == 0 ? (((ic >> 1) * nby * nbz) << 5) + ((ic & 1) << 4) : (ax == 1 ? (((ic >> 1) * nbz) << 5) + ((ic & 1) << 3) : ((ic >> 3) << 5) + (ic & 7));
            slab_tab[e] = (unsigned)f * 4u;
        }
        __syncthreads();
    }
    const int strd[3] = {A.D1 * A.D2 * 4, A.D2 * 4, 4};   // (natural layout; byte offsets: the buffer loads take them as they are)
    const unsigned vol_bytes = BRICK ? (unsigned)((A.D0 + 1) >> 1) * (unsigned)nby * (unsigned)nbz * 128u
                                     : (unsigned)A.D0 * (unsigned)A.D1 * (unsigned)A.D2 * 4u;
    const bool live = valid && (R.amax > R.amin);
    // The interval starts at the slab test's a_min -- or an ulp later, at the alpha the march's OWN plane arithmetic gives the face the ray
    // enters through, (p + (plane0 - s)) * (1 / d): the ray-driven brick splat (k_siddon_splat) clips a ray against its brick's box with
    // that expression, the volume's faces included, and forward and backward must agree on the ends of the FIRST segment to the bit as
    // on every other (a lookup of its midpoint that lands within an ulp of a rounding threshold: tools/fuzz_soak.py seed 120742,
    // round 5 -- the pair disagreed on one segment of one ray in 3 600 non-exact cases).  a_hi gets the same treatment below.
    float alo = live ? R.amin : 0.f;
    if (live) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (R.d[i] != 0.f) alo = fmaxf(alo, ((R.d[i] >= 0.f ? 0.f : (float)D[i]) + (A.sp.plane0[i] - R.s[i])) * (1.f / R.d[i]));
    }

    // per axis: the voxel the ray enters, the next plane in travel direction and the far boundary plane (as floats: exact small
    // integers), plane0 - s, 1 / d, the signed stride
    float fp[3], fpfar[3], ps[3], inv_d[3], stpf[3];
    int sstr[3], off = 0, i0s[3];
    float ahi = live ? R.amax : 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        // (a direction component of exactly 0 -- an axis-aligned ray under eps = 0, or t - s = -eps -- : 1 / d = +-inf would
        //  make the far-plane bound below -inf for the lower plane (the ray is silently rendered as 0) and a minor plane's
        //  alpha -inf < a_end (a crossing that does not exist).  A huge FINITE reciprocal of the travel direction's sign puts
        //  every plane of that axis beyond a_hi, which is what the merge walk's min / max over +-inf amounts to)
        inv_d[i] = R.d[i] == 0.f ? 1e30f : 1.f / R.d[i];
        ps[i] = A.sp.plane0[i] - R.s[i];
        const float f = fmaf(alo, R.d[i], R.s[i]) - A.sp.plane0[i];   // entry position in plane-index units
        const int i0 = min(max((int)floorf(f), 0), D[i] - 1);
        const bool fwd = R.d[i] >= 0.f;   // (d = 0 counts as forward: its next plane, strictly above the ray, lies at alpha = +huge)
        i0s[i] = i0;
        fp[i] = (float)(i0 + (fwd ? 1 : 0));
        fpfar[i] = fwd ? (float)D[i] : 0.f;
        stpf[i] = fwd ? 1.f : -1.f;
        sstr[i] = fwd ? strd[i] : -strd[i];
        off += i0 * strd[i];
        if (R.d[i] != 0.f) ahi = fminf(ahi, (fpfar[i] + ps[i]) * inv_d[i]);   // the loop's own plane arithmetic: no far plane is ever crossed
    }
    // A ray with the volume BEHIND it (a source inside or beside the volume, the ray pointing away: every far plane at a negative
    // alpha) leaves a_hi < a_lo = 0.  The loop clamps its plane alphas into [a_end, a_c] then -- the wrong way round -- and the two
    // minor planes cut a "segment" of positive length out of it, credited with whatever voxel the clamped index names: such rays
    // rendered as garbage instead of 0 (tools/fuzz_soak.py seed 70034, round 5; the exact map's small launches take the merge walk and
    // never showed it).  An empty interval is [a_lo, a_lo].
    ahi = fmaxf(ahi, alo);
    const float ad0 = fabsf(R.d[0]), ad1 = fabsf(R.d[1]), ad2 = fabsf(R.d[2]);
    const int m = (ad0 >= ad1 && ad0 >= ad2) ? 0 : (ad1 >= ad2 ? 1 : 2);
    const int u = m == 2 ? 0 : m + 1, v = 3 - m - u;
    float fpm = sel3f(m, fp[0], fp[1], fp[2]), fpu = sel3f(u, fp[0], fp[1], fp[2]), fpv = sel3f(v, fp[0], fp[1], fp[2]);
    const float psm = sel3f(m, ps[0], ps[1], ps[2]), psu = sel3f(u, ps[0], ps[1], ps[2]), psv = sel3f(v, ps[0], ps[1], ps[2]);
    const float ivm = sel3f(m, inv_d[0], inv_d[1], inv_d[2]), ivu = sel3f(u, inv_d[0], inv_d[1], inv_d[2]), ivv = sel3f(v, inv_d[0], inv_d[1], inv_d[2]);
    const float stm = sel3f(m, stpf[0], stpf[1], stpf[2]), stu = sel3f(u, stpf[0], stpf[1], stpf[2]), stv = sel3f(v, stpf[0], stpf[1], stpf[2]);
    const int sm = sel3i(m, sstr[0], sstr[1], sstr[2]), su = sel3i(u, sstr[0], sstr[1], sstr[2]), sv = sel3i(v, sstr[0], sstr[1], sstr[2]);
    // BRICK: LDS byte address of the table entry of the NEXT index along every role's axis, its step, and the current partial offsets
    int am_a = 0, au_a = 0, av_a = 0, st4m = 0, st4u = 0, st4v = 0;
    unsigned fm = 0, fu = 0, fv = 0, fm_n = 0, fu_n = 0, fv_n = 0;
    // (LDS addresses, the table's base included: added once here instead of in front of every ds_read)
    typedef __attribute__((address_space(3))) const unsigned lds_u32;
    const int tab0 = BRICK ? (int)(unsigned)(size_t)(lds_u32*)slab_tab : 0;
    auto tabrd = [&](int lds_addr) { return *(lds_u32*)(size_t)(unsigned)lds_addr; };
    if (BRICK && !NX) {
        const int im = sel3i(m, i0s[0], i0s[1], i0s[2]), iu = sel3i(u, i0s[0], i0s[1], i0s[2]), iv = sel3i(v, i0s[0], i0s[1], i0s[2]);
        const int bm = sel3i(m, tbase[0], tbase[1], tbase[2]), bu = sel3i(u, tbase[0], tbase[1], tbase[2]), bv = sel3i(v, tbase[0], tbase[1], tbase[2]);
        st4m = stm > 0.f ? 4 : -4; st4u = stu > 0.f ? 4 : -4; st4v = stv > 0.f ? 4 : -4;
        fm = tabrd(tab0 + (bm + im) * 4); fu = tabrd(tab0 + (bu + iu) * 4); fv = tabrd(tab0 + (bv + iv) * 4);
        am_a = tab0 + (bm + im) * 4 + st4m; au_a = tab0 + (bu + iu) * 4 + st4u; av_a = tab0 + (bv + iv) * 4 + st4v;
        fm_n = tabrd(am_a); fu_n = tabrd(au_a); fv_n = tabrd(av_a);
        off = (int)(fm + fu + fv);
    }

    // NX: index of a segment's voxel along axis k = clamp(floor(S * hA_k + Bc_k + 1/2), 0, D_k - 1), S = the sum of its end alphas
    float hA0 = 0.f, hA1 = 0.f, hA2 = 0.f, Bc0 = 0.f, Bc1 = 0.f, Bc2 = 0.f;
    if (NX) {
        hA0 = 0.5f * A.sp.a[0] * R.d[0]; hA1 = 0.5f * A.sp.a[1] * R.d[1]; hA2 = 0.5f * A.sp.a[2] * R.d[2];
        Bc0 = fmaf(A.sp.a[0], R.s[0], A.sp.b[0]); Bc1 = fmaf(A.sp.a[1], R.s[1], A.sp.b[1]); Bc2 = fmaf(A.sp.a[2], R.s[2], A.sp.b[2]);
    }
    const int Dm0 = A.D0 - 1, Dm1 = A.D1 - 1, Dm2 = A.D2 - 1, SXe = A.D1 * A.D2, SYe = A.D2;   // (element strides: D1 D2 < 2^24, the host checks)
    const int tx0 = tab0 + tbase[0] * 4, tx1 = tab0 + tbase[1] * 4, tx2 = tab0 + tbase[2] * 4;
    auto nx_idx = [&](const float p, const int hi) -> int {   // clamp(floor(p + 1/2), 0, hi): two instructions
        int i, r;
        asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(i) : "v"(p));
        asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(i), "v"(hi));
        return r;
    };
    auto nx_off = [&](const float S) -> int {
        const int ix = nx_idx(fmaf(S, hA0, Bc0), Dm0), iy = nx_idx(fmaf(S, hA1, Bc1), Dm1), iz = nx_idx(fmaf(S, hA2, Bc2), Dm2);
        if (BRICK) return (int)(tabrd(tx0 + ix * 4) + tabrd(tx1 + iy * 4) + tabrd(tx2 + iz * 4));
        // (v_mad_u32_u24, full rate: a 32-bit v_mul_lo_u32 occupies the SIMD four times as long)
        return (int)((unsigned)__umul24((unsigned)ix, (unsigned)SXe) + (unsigned)__umul24((unsigned)iy, (unsigned)SYe) + (unsigned)iz) << 2;
    };
    float ac = alo, acc = 0.f;
    float Um = 0.f, Uu = 0.f, Uv = 0.f, Mm = 0.f, Mu = 0.f, Mv = 0.f;
    // Loads go through a buffer resource over the volume (an offset beyond it returns 0 without a memory request: the bounds net);
    // the three loads of a slab are predicated in EXEC, see the loop.
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(vol), (short)0, vol_bytes, 0x00020000);
#if defined(XVR_SLAB_ABLATE)   // diagnostic build only (tools/ablate_siddon_slab.py): voxel values made up from the offset, no loads -- WRONG image
    auto ld = [&](bool p, int o) { return p ? __int_as_float(0x3f000000 | (o & 0xffff)) : 0.f; };
#else
    auto ld = [&](bool p, int o) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, p ? o : -1, 0, 0)); };
#endif
    // the first voxel's value opens the walk: the entry crossing (on axis ax_in, when the ray enters through a real plane) is
    // added after the loop, the loop's first "dominant-axis crossing" then sees no jump
    const bool walks = live && ahi > alo;
    float vfirst;
    if (NX) {
        // the value of the FIRST segment (the jacobian's entry term, and Wprev of the first trip: no jump at the slab's entry): under
        // the exact map it is the entry voxel's, known up front; here it is looked up at the midpoint of the first trip's first
        // segment of positive length -- the first trip's own expressions, evaluated once more in front of the loop
        const float am = (fpm + psm) * ivm, aend = med3f(am, ac, ahi);
        const float au = med3f((fpu + psu) * ivu, ac, aend), av = med3f((fpv + psv) * ivv, ac, aend);
        const float lo = fminf(au, av), hi = fmaxf(au, av);
        const float Sf = lo > ac ? ac + lo : (hi > lo ? lo + hi : hi + aend);
        vfirst = ld(walks, nx_off(Sf));
    } else {
        vfirst = ld(walks, off);
    }
    float Wprev = vfirst;
    unsigned cnt = 0;

    // (terminates: the dominant plane counter moves every trip, so its alpha passes a_hi after at most D_m trips; a NaN alpha
    //  clamps to the slab's start and the next trip's differs)
    const int tab_last = tab0 + (A.D0 + A.D1 + A.D2 + 5) * 4;
    if (__builtin_amdgcn_ballot_w64(ac < ahi)) do {             // wave-uniform: until every ray of the wavefront has left the volume
        const float am = (fpm + psm) * ivm;
        const float aend = med3f(am, ac, ahi);
        const float aur = (fpu + psu) * ivu, avr = (fpv + psv) * ivv;
        const bool cu = aur < aend, cv = avr < aend;            // the minor planes crossed inside this slab
        const float au = med3f(aur, ac, aend), av = med3f(avr, ac, aend);
        const bool uf = au <= av;                               // u is crossed first
        const float lo = uf ? au : av, hi = uf ? av : au;
        const float l1 = lo - ac, l2 = hi - lo, l3 = aend - hi;
        int off2, off3, off_next = 0;
        if (NX) {
            off = nx_off(ac + lo); off2 = nx_off(lo + hi); off3 = nx_off(hi + aend);
        } else if (BRICK) {
            const unsigned fu2 = cu ? fu_n : fu, fv2 = cv ? fv_n : fv;      // partial offsets behind the minor crossings
            off2 = (int)(fm + (uf ? fu2 + fv : fu + fv2));
            off3 = (int)(fm + fu2 + fv2);
            off_next = (int)(fm_n + fu2 + fv2);
            fu = fu2; fv = fv2; fm = fm_n;
            au_a += cu ? st4u : 0; av_a += cv ? st4v : 0; am_a += st4m;
            // (the minor axes only ever cross interior planes: their next index stays in [-1, D], inside the padded tables; the
            //  dominant axis' address runs on while the wavefront's other rays finish -- those trips' loads are masked -- and is
            //  clamped into the table)
            fu_n = tabrd(au_a); fv_n = tabrd(av_a); fm_n = tabrd(min(max(am_a, tab0), tab_last));
        } else {
            const int du = cu ? su : 0, dv = cv ? sv : 0;
            off2 = off + (uf ? du : dv);
            off3 = off + du + dv;
            off_next = off3 + sm;
        }
        const bool p1 = l1 > 0.f, p2 = l2 > 0.f, p3 = l3 > 0.f;
#if !defined(XVR_SLAB_ABLATE)
        // The three loads of a slab under EXEC masks, back to back, in one asm block (written as branches the compiler chains each
        // load behind its predecessor's select).  A lane without the segment costs the texture-address unit nothing this way.
        // Until late in round 4 such a lane passed the buffer resource an out-of-range offset instead (0 back, no memory
        // request, no branch) -- not free: tools/microbench/ta_masked_loads.hip measures 32 clocks of the TA for a wavefront
        // load of 40 live lanes in 8 lines with 24 out-of-range lanes against 8 with those lanes masked in EXEC, 39 % of the
        // march's lanes are such lanes, and it ran with TA_BUSY at 85 %: 5.87 -> 5.00 ms.  (Letting them re-load the ray's
        // current voxel -- a valid address in a line the wavefront fetches anyway -- was worse: 7.1 ms.)  The registers of
        // masked lanes keep whatever they held: every use below is behind the segment's predicate.  The buffer resource
        // stays as the bounds net.
        float t1, t2, t3;
        {
            const unsigned long long m1 = __builtin_amdgcn_ballot_w64(p1), m2 = __builtin_amdgcn_ballot_w64(p2), m3 = __builtin_amdgcn_ballot_w64(p3);
            unsigned long long sv;
            asm volatile("s_mov_b64 %[sv], exec\n\t"
                         "s_and_b64 exec, %[sv], %[m1]\n\t"
                         "buffer_load_dword %[t1], %[o1], %[rs], 0 offen\n\t"
                         "s_and_b64 exec, %[sv], %[m2]\n\t"
                         "buffer_load_dword %[t2], %[o2], %[rs], 0 offen\n\t"
                         "s_and_b64 exec, %[sv], %[m3]\n\t"
                         "buffer_load_dword %[t3], %[o3], %[rs], 0 offen\n\t"
                         "s_mov_b64 exec, %[sv]"
                         : [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [sv] "=&s"(sv)
                         : [o1] "v"(off), [o2] "v"(off2), [o3] "v"(off3), [rs] "s"(rsrc), [m1] "s"(m1), [m2] "s"(m2), [m3] "s"(m3)
                         : "memory", "scc");
        }
#define XVR_SLAB_WAIT_LOADS() asm volatile("s_waitcnt vmcnt(0)" : "+v"(t1), "+v"(t2), "+v"(t3))
#else
        float t1 = ld(p1, off), t2 = ld(p2, off2), t3 = ld(p3, off3);
#define XVR_SLAB_WAIT_LOADS() ((void)0)
#endif
        if (A.work) cnt += (unsigned)__popcll(__builtin_amdgcn_ballot_w64(p1)) + (unsigned)__popcll(__builtin_amdgcn_ballot_w64(p2)) +
                           (unsigned)__popcll(__builtin_amdgcn_ballot_w64(p3));
        // (what does not need the loaded values, while they are in flight)
        const float ac0 = ac;
        fpu += cu ? stu : 0.f;
        fpv += cv ? stv : 0.f;
        fpm += stm;
        if (!NX) off = off_next;
        ac = aend;
        XVR_SLAB_WAIT_LOADS();
        if (!JAC) {
            acc = fmaf(p1 ? t1 : 0.f, l1, acc);
            acc = fmaf(p2 ? t2 : 0.f, l2, acc);
            acc = fmaf(p3 ? t3 : 0.f, l3, acc);
        }
        if (JAC) {
            const float v1 = p1 ? t1 : Wprev, v2 = p2 ? t2 : v1, v3 = p3 ? t3 : v2;
            acc = fmaf(v1, l1, acc);                            // (a zero-length segment repeats its predecessor's value, times 0)
            acc = fmaf(v2, l2, acc);
            acc = fmaf(v3, l3, acc);
            const float Jm = Wprev - v1, J1 = v1 - v2, J2 = v2 - v3;   // jumps at the slab's entry plane, at lo, at hi
            const float Ju = uf ? J1 : J2, Jv = uf ? J2 : J1;
            Um = fmaf(Jm, ac0, Um); Mm += Jm;
            Uu = fmaf(Ju, au, Uu); Mu += Ju;
            Uv = fmaf(Jv, av, Uv); Mv += Jv;
            Wprev = v3;
        }
    } while (__builtin_amdgcn_ballot_w64(ac < ahi));

    if (valid) {
        A.out[(size_t)b * A.n + r] = acc * R.L;
        if (JAC) {
            float U[3], M[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                U[i] = m == i ? Um : (u == i ? Uu : Uv);
                M[i] = m == i ? Mm : (u == i ? Mu : Mv);
                if (walks) {
                    if (R.ax_in == i) { U[i] = fmaf(-vfirst, alo, U[i]); M[i] -= vfirst; }      // entry: 0 -> first voxel
                    if (R.ax_out == i) { U[i] = fmaf(Wprev, ahi, U[i]); M[i] += Wprev; }         // exit: last voxel -> 0
                }
            }
            float4* jp = reinterpret_cast<float4*>(A.jac + ((size_t)b * A.n + r) * XVR_DRR_JAC_STRIDE);
            jp[0] = make_float4(acc, R.L * inv_d[0] * (U[0] - M[0]), R.L * inv_d[1] * (U[1] - M[1]), R.L * inv_d[2] * (U[2] - M[2]));
            jp[1] = make_float4(-R.L * inv_d[0] * U[0], -R.L * inv_d[1] * U[1], -R.L * inv_d[2] * U[2], 0.f);
        }
    }
    if (A.work && (threadIdx.x & 63) == 0 && cnt) atomicAdd(A.work, (unsigned long long)cnt);
}

}  // namespace

extern "C" {

static int siddon_forward_impl(const float* volume, const float* mask, int D0, int D1, int D2, int C,
                           const float* source, const float* target, const float* raylen, int B, int n,
                           const xvr_drr_spec* sp, float* out, float* jac, unsigned long long* work,
                           void* stream, const float* cam) {
    int rc = check_common(volume, D0, D1, D2, C, source, target, raylen, B, n, sp, cam);
    if (rc) return rc;
    if (!out) return fail(XVR_DRR_E_ARG, "out is null");
    if (sp->volume_layout != 0 && sp->volume_layout != 2) return fail(XVR_DRR_E_UNSUPPORTED, "siddon takes the natural or the bricked volume layout");
    const bool packed = !mask && C > 1;
    // (bricks: one channel, exact index map -- where they pay; masked / non-exact walks measured slower with them)
    if (sp->volume_layout == 2 && (mask || packed || (!siddon_exact_geometry(sp) && !siddon_map_in_bounds(sp, D0, D1, D2))))
        return fail(XVR_DRR_E_UNSUPPORTED, "the bricked layout serves the one-channel forward (exact index map, or one that stays inside the volume)");   // labels in the low mantissa bits of `volume` (xvr_drr_pack_labels)
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
    // the dominant-axis slab march (round 4) -- and, round 5, the same march for NON-exact index maps that keep points of the
    // volume inside it (option siddon_slab: 1 = both, 2 = the exact map only, 0 = the merge walk)
    const int slab_opt = xvr_detail::option(xvr_detail::OPT_SIDDON_SLAB);
    const bool nx = !ex && slab_opt == 1 && siddon_map_in_bounds(sp, D0, D1, D2) && (long long)D1 * D2 < (1LL << 24);
    if ((sp->volume_layout == 0 || sp->volume_layout == 2) && (ex ? slab_opt != 0 : nx) &&
        (long long)D0 * D1 * D2 < (1LL << 29) && (size_t)(D0 + D1 + D2 + 6) * 4 <= 48 * 1024 &&
        // (non-exact maps take the march at EVERY launch size: there is no alpha-split walk for them, the unsplit march beats the
        //  unsplit merge walk, and forward and voxel gradient -- k_siddon_splat -- then break the map's ties the same way)
        (nx || split_factor(B, n, (long long)D0 * D1 * D2, true, &tile16) == 1)) {
        const size_t tab = sp->volume_layout == 2 ? (size_t)(D0 + D1 + D2 + 6) * 4 : 0;
        if (sp->volume_layout == 2) {
            if (nx) return jac ? launch(k_siddon_slab<true, true, true>, A, tab, stream) : launch(k_siddon_slab<false, true, true>, A, tab, stream);
            if (jac) return launch(k_siddon_slab<true, true>, A, tab, stream);
            return launch(k_siddon_slab<false, true>, A, tab, stream);
        }
        if (nx) return jac ? launch(k_siddon_slab<true, false, true>, A, 0, stream) : launch(k_siddon_slab<false, false, true>, A, 0, stream);
        if (jac) return launch(k_siddon_slab<true, false>, A, 0, stream);
        return launch(k_siddon_slab<false, false>, A, 0, stream);
    }
    if (sp->volume_layout == 2 && !ex) return fail(XVR_DRR_E_UNSUPPORTED, "the bricked layout serves non-exact index maps through the slab march only");
    if (sp->volume_layout == 2) {
        if (jac) return launch(k_siddon<1, false, false, false, true, 0, true>, A, 0, stream);
        return launch(k_siddon<0, false, false, false, true, 0, true>, A, 0, stream);
    }
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
    // (a non-exact index map gathers per plane cell into octant sums: needs the larger workspace of
    //  xvr_drr_siddon_backward_workspace_bytes and a map that drifts by less than a voxel)
    int olo[3] = {0, 0, 0};
    const bool drift_ok = !exact_geom && siddon_cell_offsets(sp, D0, D1, D2, olo);
    // option siddon_splat: 1 (default) = the ray-driven brick splat (k_siddon_splat, round 5) for non-exact maps; 2 = for the exact
    // map too (A/B against k_siddon_gather_vol2); 0 = the round-2 per-cell gather (needs the larger workspace)
    const int splat_opt = xvr_detail::option(xvr_detail::OPT_SIDDON_SPLAT);
    // (the maps the forward's slab march serves -- points of the volume look up voxels inside it --, so that forward and voxel gradient
    //  are one pair; a map that leaves the volume keeps the merge walk's family: per-cell gather or scatter)
    const bool nx_in = !exact_geom && siddon_map_in_bounds(sp, D0, D1, D2);
    const bool splat = !mask && siddon_splat_detector_ok(sp, n) && ((nx_in && splat_opt >= 1) || (exact_geom && splat_opt == 2));
    const bool cells = drift_ok && !splat &&
                       workspace_bytes >= align256(ws_bytes(B, n, D0, D1, D2)) + siddon_cells_bytes(D0, D1, D2);
    // (a mask with a per-channel gradient: the one-voxel-per-lane gather that looks the upstream value up by the voxel's own
    //  label -- exact geometry, where a segment's voxel is the voxel whose box holds it)
    if (gvol && ((!mask && (exact_geom || cells || splat)) || (mask && exact_geom)) && gather_usable(sp, n, workspace, workspace_bytes, B, D0, D1, D2)) {
        unsigned* flag = nullptr;
        rc = launch_gather(true, source, target, raylen, grad_out, B, n, sp->ray_grid_w, D0, D1, D2, sp, grad_volume,
                           workspace, stream, &flag, mask, C, exact_geom ? nullptr : olo, splat ? (exact_geom ? 1 : 2) : 0);
        if (rc || gather_slab_later()) return rc;
        RenderArgs Ap = A, Av = A;
        Ap.gvol = nullptr;
        Av.gsrc = nullptr; Av.gtgt = nullptr; Av.glen = nullptr;
        Av.skip_unless_flag_gt = flag;
        if (exact_geom && mask) {
            if (gpose) { rc = launch(k_siddon<2, true, true, false, true>, Ap, lds, stream); if (rc) return rc; }
            return launch(k_siddon<2, true, false, true, true>, Av, lds, stream);
        }
        if (exact_geom) {
            if (gpose) { rc = launch(k_siddon<2, false, true, false, true>, Ap, 0, stream); if (rc) return rc; }
            return launch(k_siddon<2, false, false, true, true>, Av, 0, stream);
        }
        if (gpose) { rc = launch(k_siddon<2, false, true, false, false>, Ap, 0, stream); if (rc) return rc; }
        return launch(k_siddon<2, false, false, true, false>, Av, 0, stream);
    }
    if (gather_slab_later()) return XVR_DRR_OK;   // (option gather_slab: the call for slab 0 did everything)
#define SID_BWD2(M, E)                                                                  \
    (gpose ? (gvol ? launch(k_siddon<2, M, true, true, E>, A, lds, stream)              \
                   : launch(k_siddon<2, M, true, false, E>, A, lds, stream))            \
           : launch(k_siddon<2, M, false, true, E>, A, lds, stream))
#define SID_BWD(M) (exact_geom ? SID_BWD2(M, true) : SID_BWD2(M, false))
    return mask ? SID_BWD(true) : SID_BWD(false);
#undef SID_BWD
#undef SID_BWD2
}
}  // extern "C"
