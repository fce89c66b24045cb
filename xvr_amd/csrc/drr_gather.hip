// MI355X (gfx950) differentiable-DRR kernels: the voxel gradient as an atomic-free, voxel-driven gather
// (trilinear and Siddon), its per-pose preparation and per-brick cull.  HISTORY.md section 4.1.
#include "drr_common.hiph"
#include <type_traits>

#ifdef XVR_GATHER_STATS
// Diagnostic build only (tools/gather_stats.py compiles it into a separate library): loop-trip counters of the
// trilinear gather.  0 (lane, pose) visits . 1 steps k . 2 rows . 3 rows with an empty pixel interval . 4 candidates .
// 5 unused . 6 wavefront-level inner trips . 7 wavefront-level pose iterations
__device__ unsigned long long g_gather_stats[8];
#define XVR_STAT(i, n) (st[i] += (n))
#define XVR_STAT_WAVE(i) do { if ((threadIdx.x & 63) == __builtin_ctzll(__ballot(1))) st[i]++; } while (0)
#else
#define XVR_STAT(i, n) ((void)0)
#define XVR_STAT_WAVE(i) ((void)0)
#endif
#ifdef XVR_S16_TRACE
// Diagnostic build only (tools/splat_trace.py): per workgroup of k_trilinear_splat_b16 -- wall clock at start and end
// (100 MHz constant clock), visits, samples.
__device__ unsigned long long g_s16_trace[12 * 65536];
#endif
namespace {

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

constexpr int CMAX_STRIDE = 32;   // words between two poses' max |c| (GatherArgs.cmax): one 128-byte line each
// above this many samples of one pose per voxel the plain trilinear voxel gradient takes the fp32 table gather instead of the
// fixed-point splat (the benchmark geometry has ~2, registration ~3; the "fine detector" case of tests/test_splat.py ~1500)
constexpr float SPLAT_MAX_SAMPLES_PER_VOXEL = 48.f;

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
    float dev = 0.f, cabs = 0.f;
    if (r < G.n) {
        const int i = r / G.W, j = r - i * G.W;
        const float tx = T[(size_t)r * 3], ty = T[(size_t)r * 3 + 1], tz = T[(size_t)r * 3 + 2];
        dev = fmaxf(fabsf(tx - (t00[0] + j * ec[0] + i * er[0])),
                    fmaxf(fabsf(ty - (t00[1] + j * ec[1] + i * er[1])), fabsf(tz - (t00[2] + j * ec[2] + i * er[2]))));
        dev = pitch > 0.f ? dev / pitch : INFINITY;
        if (!(dev == dev)) dev = INFINITY;
        // (mask -> channels: the upstream gradient depends on the sample's label and is looked up per candidate)
        float c = (G.mask ? 1.f : G.gout[(size_t)b * G.n + r]) * G.raylen[(size_t)b * G.n + r] * spec_window(G.sp).inv_denom;
        // d exactly as the forward forms it, (t - s) + eps, so that the gather's fmaf chain below
        // reproduces the forward's sample positions bit for bit
        const float sx = G.source[3 * b], sy = G.source[3 * b + 1], sz = G.source[3 * b + 2];
        const float ddx = (tx - sx) + G.sp.eps, ddy = (ty - sy) + G.sp.eps, ddz = (tz - sz) + G.sp.eps;
        if (!G.siddon) {
            // a * d: the gather forms a (s + alpha d) + b - v as fma(alpha, a d, a s + b - v) (a = 1 for the default index map)
            float4* row = G.q + (size_t)b * G.qn + (size_t)i * G.qs;
            row[j] = make_float4(G.sp.a[0] * ddx, G.sp.a[1] * ddy, G.sp.a[2] * ddz, c);
            if (G.mask && G.cmax) {   // the splat's bound needs the largest upstream value over the channels
                float gm = 0.f;
                for (int ch = 0; ch < G.C; ++ch) {
                    const float g = G.gout[((size_t)b * G.C + ch) * G.n + r];
                    gm = (g == g) ? fmaxf(gm, fabsf(g)) : INFINITY;
                }
                cabs = c * gm;
                cabs = (cabs == cabs) ? fabsf(cabs) : INFINITY;
            } else {
                cabs = (c == c) ? fabsf(c) : INFINITY;   // (a NaN counts as infinite: the splat poisons what the pose touches)
            }
            if (j == G.W - 1) row[G.W] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (G.clip) {
                // the ray's own [alpha_min, alpha_max], computed exactly as ray_setup() does for the forward; the image is
                // scaled by their span
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
                row[j].w = c * fmaxf(hi - lo, 0.f);
                G.q2[(size_t)b * G.n + r] = make_float2(lo, hi);
                if (G.cmax) {
                    // a ray's samples sit span * L_x / (N - 1) apart (L_x = |a d|, its length in index space): at most m of them
                    // inside one voxel's 2-cube, each worth |c| span at most
                    const float span = fmaxf(hi - lo, 0.f), N1 = (float)(G.sp.n_points > 1 ? G.sp.n_points - 1 : 1);
                    const float lx = sqrtf(G.sp.a[0] * ddx * G.sp.a[0] * ddx + G.sp.a[1] * ddy * G.sp.a[1] * ddy + G.sp.a[2] * ddz * G.sp.a[2] * ddz);
                    const float gap = span * lx * (spec_window(G.sp).far_ - spec_window(G.sp).near_) / N1;
                    const float m = gap > 0.f ? fminf(3.4641016f / gap + 1.f, (float)G.sp.n_points) : (float)G.sp.n_points;
                    cabs = cabs * span * m;
                    if (!(cabs == cabs)) cabs = INFINITY;
                }
            }
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
            // A ray whose range is cut by alpha = 0 or 1 (source or detector inside the volume) needs its (lo, hi) in the gather;
            // for a pose none of whose rays is cut, every voxel's cube lies inside every ray's own range and the gather leaves
            // the q2 loads out (one word per pose, GatherArgs.cmax; one atomic per wavefront that has such a ray and does not see
            // the word set yet)
            const bool cut = (lo < hi) && (!(lo > 0.f) || !(hi < 1.f));
            if (G.cmax && __any(cut) && (threadIdx.x & 63) == __builtin_ctzll(__ballot(cut)) &&
                __hip_atomic_load(G.cmax + (size_t)b * G.cmax_stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u)
                atomicMax(G.cmax + (size_t)b * G.cmax_stride, 1u);
            if (!(lo > 0.f)) lo = 0.f;
            if (!(hi < 1.f)) hi = 1.f;
            // (under a mask the upstream gradient depends on the gathering voxel's label: k_siddon_gather_mask looks it up)
            G.q[(size_t)b * G.n + r] = make_float4(1.f / ddx, 1.f / ddy, 1.f / ddz,
                                                   (G.mask ? 1.f : G.gout[(size_t)b * G.n + r]) * G.raylen[(size_t)b * G.n + r]);
            G.q2[(size_t)b * G.n + r] = make_float2(lo, hi);
            if (G.cells) G.q[(size_t)G.B * G.n + (size_t)b * G.n + r] = make_float4(ddx, ddy, ddz, 0.f);   // d itself, for the midpoints
            if (G.sid_splat) {   // the splat's bound: a chord of length l (index units) is worth |g L| l / |d| <= |g L| l min_k |1 / d_k|
                cabs = fabsf(G.gout[(size_t)b * G.n + r] * G.raylen[(size_t)b * G.n + r]) * fminf(fminf(fabsf(1.f / ddx), fabsf(1.f / ddy)), fabsf(1.f / ddz));
                if (!(cabs == cabs)) cabs = INFINITY;
            }
        }
    }
    const int cmax_word = G.siddon ? 1 : 0;   // (siddon: word 0 of the pose's line is the "a ray is cut" flag)
    if (G.cmax && (!G.siddon || G.sid_splat)) {   // max |c| of the pose: the fixed-point scale of the brick-local splat
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cabs = fmaxf(cabs, __shfl_xor(cabs, o));
        // (256 workgroups per pose meet in its word: the poses' words sit in different cache lines -- packed, the 116
        //  of the benchmark shared four lines and 1.2e5 L2 atomics queued up behind each other for 0.6 ms -- and only a
        //  wavefront that would RAISE the maximum writes.  A stale read only costs a redundant atomic.)
        __shared__ float s_cabs[WG / 64];
        if ((threadIdx.x & 63) == 0) s_cabs[threadIdx.x >> 6] = cabs;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < WG / 64; ++w) cabs = fmaxf(cabs, s_cabs[w]);
            if (cabs > 0.f && __float_as_uint(cabs) > __hip_atomic_load(G.cmax + (size_t)b * G.cmax_stride + cmax_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                atomicMax(G.cmax + (size_t)b * G.cmax_stride + cmax_word, __float_as_uint(cabs));
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
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float Nj = P.nh[j] / G.sp.a[j], Gj = gr[j] / G.sp.a[j];
            // (an axis almost parallel to the sample planes would give a huge breakpoint whose bound is a difference
            //  of huge numbers: skipped -- any subset of the breakpoints, with 0, still bounds from above)
            if (fabsf(Nj) > 1e-3f * P.dalpha) {
                const float lam = Gj / Nj;
                float c = 0.f;
                for (int i = 0; i < 3; ++i) c += fabsf(gr[i] / G.sp.a[i] - lam * (P.nh[i] / G.sp.a[i]));
                P.rl[j] = lam;
                P.rc[j] = c;
            } else {
                P.rl[j] = 0.f;
                P.rc[j] = 1e30f;
            }
        }
        const float HSv = G.V == 2 ? 1.5f : 1.0f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            P.e0[i] = G.sp.a[i] * ts[i];
            P.era[i] = G.sp.a[i] * er[i];
            const float eca = G.sp.a[i] * ec[i];
            P.rec[i] = fabsf(eca) < 1e-9f ? 1e9f : 1.f / eca;
            P.hsr[i] = HSv * fabsf(P.rec[i]);
        }
        {   // the sample lattice's three spacings in index space (the splat's bound on the samples that can touch one voxel)
            float eca[3], cr[3], nha[3];
            for (int i = 0; i < 3; ++i) { eca[i] = G.sp.a[i] * ec[i]; nha[i] = P.nh[i] / G.sp.a[i]; }
            cross3(P.era, eca, cr);
            P.ecl = sqrtf(dot3(eca, eca));
            P.rperp = P.ecl > 0.f ? sqrtf(dot3(cr, cr)) / P.ecl : 0.f;
            P.gn = sqrtf(dot3(nha, nha));
        }
        P.ec_len = sqrtf(dot3(ec, ec));
        P.er_len = sqrtf(dot3(er, er));
        P.nh_norm = sqrtf(dot3(P.nh, P.nh));
        P.gc_norm = sqrtf(dot3(gc, gc));
        P.gr_norm = sqrtf(dot3(gr, gr));
        P.gc0 = dot3(gc, st);
        P.gr0 = dot3(gr, st);
        if (G.spv_limit > 0.f && !G.siddon) {
            // samples of this pose per voxel at the volume's centre: the sample lattice's three spacings in index units are
            // alpha * ecl (pixels of a row), alpha * rperp (rows), step / gn (planes).  Beyond spv_limit the fixed-point floor of the
            // brick-local splat shows (its LSB scales with the bound on a voxel's sum, i.e. with this number): word 3 of the flag
            // line sends the launch to the fp32 table gather instead (round 5; include/xvr_drr.h, "ACCURACY")
            float wc[3];
            const float Dm[3] = {0.5f * (float)(G.D0 - 1), 0.5f * (float)(G.D1 - 1), 0.5f * (float)(G.D2 - 1)};
            for (int i = 0; i < 3; ++i) wc[i] = (Dm[i] - G.sp.b[i]) / G.sp.a[i] - s[i];
            const float ac = fmaxf(dot3(P.nh, wc), 0.05f);
            const int N = G.sp.n_points;
            const float stepa = N > 1 ? (spec_window(G.sp).far_ - spec_window(G.sp).near_) / (float)(N - 1) : 1.f;
            const float cellv = (ac * P.ecl) * (ac * P.rperp) * (stepa / fmaxf(P.gn, 1e-20f));
            if (cellv > 0.f && 1.f / cellv > G.spv_limit) atomicOr(G.flag + 3, 1u);
        }
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
    if (G.only_if_fine >= 0 && (int)(G.flag[3] != 0u) != G.only_if_fine) return;   // (the kernel pair of the other regime runs instead)
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
    // (clip_to_volume == 2: the range of alpha any sample can take is the call's window, known on the device only)
    const float cull_lo = G.sp.alpha_window ? spec_window(G.sp).near_ : G.cull_lo, cull_hi = G.sp.alpha_window ? spec_window(G.sp).far_ : G.cull_hi;
    for (int wd = 0; wd < G.words; ++wd) {
        const int p = wd * 32 + lane;
        bool hit = false;
        if (p < G.B) {
            const PoseLattice& P = G.poses[p];
            float w[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) w[i] = (c[i] - G.sp.b[i]) / G.sp.a[i] - P.s[i];
            // the brick with its interpolation support is a box: alpha range from its L1 extent, and -- when it lies in
            // front of the source -- its pixel footprint is the bounding box of its 8 projected corners (exact for a
            // convex body); one pixel of slack covers the lattice tolerance
            const float en0 = P.nh[0] * hx, en1 = P.nh[1] * hy, en2 = P.nh[2] * hz;
            const float av = dot3(P.nh, w), da = fabsf(en0) + fabsf(en1) + fabsf(en2);
            const float amin = av - da, amax = av + da;
            if (amax >= cull_lo && amin <= cull_hi) {
                if (amin <= 1e-6f) {
                    hit = true;  // the box reaches the source plane: no perspective bound, keep
                } else {
                    const float nj = dot3(P.gc, w), ni = dot3(P.gr, w);
                    const float ec0 = P.gc[0] * hx, ec1 = P.gc[1] * hy, ec2 = P.gc[2] * hz;
                    const float er0 = P.gr[0] * hx, er1 = P.gr[1] * hy, er2 = P.gr[2] * hz;
                    float jmn = INFINITY, jmx = -INFINITY, imn = INFINITY, imx = -INFINITY;
#pragma unroll
                    for (int cc = 0; cc < 8; ++cc) {
                        const float sx = (cc & 4) ? 1.f : -1.f, sy = (cc & 2) ? 1.f : -1.f, sz = (cc & 1) ? 1.f : -1.f;
                        const float inv = 1.f / (av + sx * en0 + sy * en1 + sz * en2);
                        const float jv = (nj + sx * ec0 + sy * ec1 + sz * ec2) * inv, iv = (ni + sx * er0 + sy * er1 + sz * er2) * inv;
                        jmn = fminf(jmn, jv); jmx = fmaxf(jmx, jv);
                        imn = fminf(imn, iv); imx = fmaxf(imx, iv);
                    }
                    hit = jmx + P.gc0 + 1.f >= 0.f && jmn + P.gc0 - 1.f <= (float)(G.W - 1) &&
                          imx + P.gr0 + 1.f >= 0.f && imn + P.gr0 - 1.f <= (float)(G.H - 1);
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
// A lane's 2 x 2 x 2 voxel sums added to the gradient: the eight loads first, then the stores -- ONE memory round trip instead of
// eight guarded read-modify-writes in a row at the end of every workgroup (the lane's voxels are its own: no other lane of the launch
// writes them).
__device__ __forceinline__ void add_block8(float* __restrict__ gvol, const float (&acc)[8], const int vx, const int vy, const int vz,
                                           const int D0, const int D1, const int D2) {
    float old[8];
    size_t at[8];
    bool on[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int x = vx + (e >> 2 & 1), y = vy + (e >> 1 & 1), z = vz + (e & 1);
        on[e] = x < D0 && y < D1 && z < D2 && acc[e] != 0.f;
        at[e] = ((size_t)x * D1 + y) * D2 + z;
        old[e] = on[e] ? gvol[at[e]] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e)
        if (on[e]) gvol[at[e]] = old[e] + acc[e];
}

__device__ __forceinline__ float hat01(float d) { return __builtin_amdgcn_fmed3f(1.f - fabsf(d), 0.f, 1.f); }

// linspace_at() with both halves evaluated and one select (same values, no exec-masked branches in the inner loops)
__device__ __forceinline__ float linspace_sel(int k, int N, float near_, float far_, float step) {
    const float lo = fmaf(step, (float)k, near_), hi = far_ - step * (float)(N - 1 - k);
    return (k < N / 2 || N == 1) ? lo : hi;
}


// ---------------------------------------------------------------------------------------------
// The same gather with the loop nest FLATTENED per lane (round 2; HISTORY.md section 4.1, tools/sim_gather_divergence.py).
//
// In the round-1 kernel (one nest of step -> row -> pixel loops per lane; retired in round 3) the 64 lanes of a wavefront walk step -> row -> pixel loops whose trip counts differ per
// lane: at the benchmark geometry a wavefront's inner trip fills 47 of its 128 candidate slots and a (wavefront, pose)
// visit costs 27 inner trips where the lanes' own work is 12 (10.3 if no lane ever idled).  Here every pose is done in
// two phases:
//   A  the lane enumerates its non-empty lattice rows -- (step k, row i, first pixel, count) -- with the cheap part of
//      the old nest (step set-up from per-pose constants: ~45 instructions instead of ~130; row set-up unchanged) and
//      appends one 32-bit entry per row to ITS column of an LDS table (lane-private: no barrier, no atomics);
//   B  one flat loop: every lane pulls its own next entry whenever its row is exhausted and evaluates two candidates
//      per trip, so a trip is short only for lanes that have run out of rows altogether: 16.3 trips per visit.
// The candidate arithmetic (gather_pair) is the nested kernel's, bit for bit; only the ORDER in which a lane adds its
// candidates is the same too (rows in step-major order), so both kernels produce identical sums.
// Entry (8 bytes) = first element of the run in q | count << 24, alpha_k; a lane that meets a row it cannot enter (table
// full: 5 % of the visits have a lane with more than 13 rows) remembers where and the nested loops of the round-1 kernel
// take over from there once phase B is done.
// Where the time goes (C2): the kernel is VALU-issue bound -- 8.4e9 wave-instructions of which the packed ones cost two
// issues, ~80 % of the issue slots of an 11.8 ms launch (profiles/r02_trilinear_rocprof_summary.md), at 41 of 64 lanes.
// A loads-ablated build (tools/ablate_gather.py) runs in 8.7 ms and one with half the loads in 10.6 ms, but that is NOT
// the memory cost: with constant candidates the compiler hoists their arithmetic out of the trip.  The honest test of the
// memory hypothesis was a variant that takes the sample positions from the lattice constants and loads only the 4-byte
// weights (one 8-byte load per trip instead of two 16-byte ones): 13.2 ms -- slower, for its extra per-row arithmetic.
// Built, measured, NOT adopted (all parity-green): that lattice-position variant; a depth-one software pipeline of the
// loads in two register sets (11.7-12.0 ms: nothing to hide); the wavefront's pixel bounding box copied once per pose
// into an LDS tile by global_load_lds with the candidates read pairwise from LDS (15.3-18.8 ms over four table / tile
// splits at 4 wavefronts per SIMD); rounds instead of the nested slow path (12.2 ms); the two candidates of a trip
// through explicit packed fp32 (v_pk_fma/mul/add_f32 on component-wise pairs, one accumulator set per candidate: 40
// VALU instructions per trip instead of 47, yet 12.1-13.0 ms -- a v_pk_*_f32 costs two plain issues on gfx950, so the
// instruction count is not the time; the compiler's own pairing of the eight accumulations is as far as packing goes).
// ---------------------------------------------------------------------------------------------
#ifndef XVR_TAB_ROWS   // (both overridable for the tuning builds of tools/tune_gather.py)
#define XVR_TAB_ROWS 13
#endif
#ifndef XVR_TAB_WAVES
#define XVR_TAB_WAVES 6
#endif
// 13 x 64 lanes x 8 B = 6.5 KiB of LDS per wavefront: 24 wavefronts per CU (6 per SIMD) fit in 160 KiB; measured flat from
// (7 per SIMD, 11 rows) to (6, 13) and (5, 15), 5-8 % worse at (8, 9) and (4, 19)
constexpr int TAB_ROWS = XVR_TAB_ROWS;
constexpr unsigned TAB_R0_BITS = 24, TAB_N_BITS = 8;   // entry.x = first element of the run in q | count << 24; entry.y = alpha_k
constexpr unsigned TAB_MAX_RAYS = 1u << TAB_R0_BITS;

// two candidates against the lane's 2x2x2 block; signed distance of the sample from the block's first voxel per axis:
// a (s + alpha d) + b - v folded into one fma (within an ulp of the forward's two-fma chain); the second voxel sits
// exactly 1 further.  (packed v_pk_mul/fma_f32 on (z, z+1) pairs measured SLOWER in round 1)
__device__ __forceinline__ void gather_pair(const float4 ta, const float4 tb, const float Ax, const float Ay, const float Az,
                                            const float Bx, const float By, const float Bz, float (&acc)[8]) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float4 t = h ? tb : ta;
        const float dx = fmaf(Ax, t.x, Bx), dy = fmaf(Ay, t.y, By), dz = fmaf(Az, t.z, Bz);
        const float ux0 = hat01(dx), uy0 = hat01(dy), uz0 = hat01(dz) * t.w;
        const float ux1 = hat01(dx - 1.f), uy1 = hat01(dy - 1.f), uz1 = hat01(dz - 1.f) * t.w;
        const float p00 = ux0 * uy0, p01 = ux0 * uy1, p10 = ux1 * uy0, p11 = ux1 * uy1;
        acc[0] = fmaf(p00, uz0, acc[0]);
        acc[1] = fmaf(p00, uz1, acc[1]);
        acc[2] = fmaf(p01, uz0, acc[2]);
        acc[3] = fmaf(p01, uz1, acc[3]);
        acc[4] = fmaf(p10, uz0, acc[4]);
        acc[5] = fmaf(p10, uz1, acc[5]);
        acc[6] = fmaf(p11, uz0, acc[6]);
        acc[7] = fmaf(p11, uz1, acc[7]);
    }
}

#if defined(XVR_GATHER_ABLATE) && XVR_GATHER_ABLATE == 2   // diagnostic build: half the loads (the pair's second = its first)
#define XVR_LOAD_Q(ptr, k) ((ptr)[0])
#elif defined(XVR_GATHER_ABLATE)   // diagnostic build only (tools/ablate_gather.py): same arithmetic, no memory traffic
#define XVR_LOAD_Q(ptr, k) make_float4((float)(k) * 0.25f, 0.5f, 0.75f, 1.f)
#else
#define XVR_LOAD_Q(ptr, k) ((ptr)[k])
#endif

__device__ __forceinline__ void gather_row(const float4* __restrict__ row, const int n, const float Ax, const float Ay, const float Az,
                                           const float Bx, const float By, const float Bz, float (&acc)[8]) {
    for (int j = 0; j < n; j += 2) {
        const bool two = j + 1 < n;
        const float4 ta = XVR_LOAD_Q(row, j);
        float4 tb = XVR_LOAD_Q(row, two ? j + 1 : j);
        tb.w = two ? tb.w : 0.f;
        gather_pair(ta, tb, Ax, Ay, Az, Bx, By, Bz, acc);
    }
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(XVR_TAB_WAVES, XVR_TAB_WAVES))) void k_trilinear_gather_tab(GatherArgs G) {
    if (*G.flag > __float_as_uint(GATHER_DEV_TOL)) return;  // not a lattice: the scatter kernel runs instead
    if (G.only_if_fine >= 0 && (int)(G.flag[3] != 0u) != G.only_if_fine) return;   // (the splat took the launch)
    __shared__ uint2 tab[TAB_ROWS * 64];
    constexpr float HS = 1.5f, CO = 0.5f;
    int bx, by, bz;
    brick_coords(blockIdx.x, G.D1, G.D2, G.bd, bx, by, bz);
    const int tid = threadIdx.x;  // one wavefront: 4 x 4 x 4 blocks of 2 x 2 x 2 voxels
    const int vx = (bx * 4 + (tid >> 4)) * 2, vy = (by * 4 + ((tid >> 2) & 3)) * 2, vz = (bz * 4 + (tid & 3)) * 2;
    const bool inb = vx < G.D0 && vy < G.D1 && vz < G.D2;
    const float fv[3] = {(float)vx, (float)vy, (float)vz};
    float xv[3];  // block centre in x coordinates
#pragma unroll
    for (int i = 0; i < 3; ++i) xv[i] = (fv[i] + CO - G.sp.b[i]) / G.sp.a[i];
    const int N = G.sp.n_points;
    const float near_ = spec_window(G.sp).near_, far_ = spec_window(G.sp).far_;
    const float step = N > 1 ? (far_ - near_) / (float)(N - 1) : 0.f;
    const float inv_step = step > 0.f ? 1.f / step : 0.f;
    const float a0 = G.sp.a[0], a1 = G.sp.a[1], a2 = G.sp.a[2];
    const float b0 = G.sp.b[0], b1 = G.sp.b[1], b2 = G.sp.b[2];
    const float bv0 = b0 - fv[0], bv1 = b1 - fv[1], bv2 = b2 - fv[2];
    const float jmargin = GATHER_DEV_TOL + 0.01f;
    const float Hm1 = (float)(G.H - 1), Wm1 = (float)(G.W - 1);
#ifdef XVR_GATHER_STATS
    unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;

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
            XVR_STAT(0, inb ? 1 : 0);
            XVR_STAT_WAVE(7);
            const float grw = P.gr[0] * w0 + P.gr[1] * w1 + P.gr[2] * w2;
            const float4* __restrict__ q = G.q + (size_t)p * G.qn;
            // a s + b - v for the block's first voxel (the weights' constant) and for its centre (the windows')
            const float Bx = fmaf(a0, s0, bv0), By = fmaf(a1, s1, bv1), Bz = fmaf(a2, s2, bv2);
            const float Cx = Bx - CO, Cy = By - CO, Cz = Bz - CO;
            const float rcx = HS * P.rc[0], rcy = HS * P.rc[1], rcz = HS * P.rc[2], rcw = HS * P.hwr;

            // per-step constants of the window arithmetic (used ONLY to find which pixels to visit; the weights come from
            // the real targets): row range [ilo, ihi] and the lattice model Q0 + i Ur + j Uc of the sample positions
            // relative to the block centre, in index space, with Uc = alpha * (a ec)
            struct StepC { float q0x, q0y, q0z, urx, ury, urz, rx, ry, rz, ax_, ay_, az_; int ilo, ihi; };
            auto step_setup = [&](const float al, StepC& S) {
                const float inv = __builtin_amdgcn_rcpf(al);   // windows only (1 ulp): the weights never see it
                const float ic = fmaf(grw, inv, P.gr0);
                // exact extent, along the detector's row axis, of the block's slice at this alpha (PoseLattice.rl/rc)
                const float dlt = al - av;
                const float up = fminf(fminf(fmaf(P.rl[0], dlt, rcx), fmaf(P.rl[1], dlt, rcy)), fminf(fmaf(P.rl[2], dlt, rcz), rcw));
                const float dn = fminf(fminf(fmaf(-P.rl[0], dlt, rcx), fmaf(-P.rl[1], dlt, rcy)), fminf(fmaf(-P.rl[2], dlt, rcz), rcw));
                S.ilo = (int)ceilf(fmaxf(ic - fmaf(fmaxf(dn, 0.f), inv, GATHER_WIN_MARGIN), 0.f));
                S.ihi = (int)floorf(fminf(ic + fmaf(fmaxf(up, 0.f), inv, GATHER_WIN_MARGIN), Hm1));
                S.q0x = fmaf(al, P.e0[0], Cx); S.q0y = fmaf(al, P.e0[1], Cy); S.q0z = fmaf(al, P.e0[2], Cz);
                S.urx = al * P.era[0]; S.ury = al * P.era[1]; S.urz = al * P.era[2];
                S.rx = inv * P.rec[0]; S.ry = inv * P.rec[1]; S.rz = inv * P.rec[2];      // 1 / Uc
                // HS / |Uc| plus the lattice tolerance: half-width of the pixel interval per axis
                S.ax_ = fmaf(inv, P.hsr[0], jmargin); S.ay_ = fmaf(inv, P.hsr[1], jmargin); S.az_ = fmaf(inv, P.hsr[2], jmargin);
            };
            // exact pixel interval [jlo, jhi] of row i where |q + j Uc| < HS on all three axes (empty: jhi < jlo)
            auto row_setup = [&](const StepC& S, const int i, int& jlo, int& jhi) {
                const float fi = (float)i;
                const float qx = fmaf(fi, S.urx, S.q0x), qy = fmaf(fi, S.ury, S.q0y), qz = fmaf(fi, S.urz, S.q0z);
                const float mx = -qx * S.rx, my = -qy * S.ry, mz = -qz * S.rz;
                const float lo = fmaxf(fmaxf(mx - S.ax_, my - S.ay_), mz - S.az_);
                const float hiJ = fminf(fminf(mx + S.ax_, my + S.ay_), mz + S.az_);
                jlo = (int)ceilf(fmaxf(lo, 0.f));
                jhi = (int)floorf(fminf(hiJ, Wm1));
            };

            // ---- phase A: enumerate the non-empty rows of every step into the lane's column of the table.  Nothing here
            // touches the accumulators.  A lane that meets a row it cannot enter (table full, more than 30 pixels, more
            // than 31 steps, a step at alpha = 0) remembers where and leaves: the slow path below takes over from there.
            int cnt = 0;
            int k_ovf = INT32_MAX, i_ovf = 0;
            for (int k = klo; k <= khi && k_ovf == INT32_MAX; ++k) {
                const float al = linspace_sel(k, N, near_, far_, step);
                XVR_STAT(1, 1);
                if (!(al > 1e-12f)) { k_ovf = k; i_ovf = INT32_MIN; break; }
                StepC S;
                step_setup(al, S);
                for (int i = S.ilo; i <= S.ihi; ++i) {
                    int jlo, jhi;
                    row_setup(S, i, jlo, jhi);
                    int n = jhi - jlo + 1;
                    XVR_STAT(2, 1);
                    XVR_STAT(3, n <= 0 ? 1 : 0);
                    XVR_STAT(4, n > 0 ? n : 0);
                    // an even number of candidates per row, so that phase B takes two per trip without a tail case: the extra
                    // element is the next pixel (outside the exact interval: all eight weights are exactly 0) or the all-zero
                    // element that closes every row of q
                    n += n & 1;
                    // (straight-line: no break / continue -- a lane that has overflown just stops entering rows)
                    const bool want = n > 0 && k_ovf == INT32_MAX;
                    const bool fits = want && cnt < TAB_ROWS && n < (1 << TAB_N_BITS);
                    if (fits) tab[cnt * 64 + tid] = make_uint2((unsigned)(i * G.qs + jlo) | ((unsigned)n << TAB_R0_BITS), __float_as_uint(al));
                    cnt += fits ? 1 : 0;
                    i_ovf = (want && !fits) ? i : i_ovf;
                    k_ovf = (want && !fits) ? k : k_ovf;
                }
            }

            // ---- phase B: one flat loop over the lane's rows, two candidates per trip (rows in the table are even)
            {
                int idx = 0, rem = 0;
                const float4* __restrict__ ptr = q;
                float al = 0.f;
                uint2 e_next = tab[tid];   // (unused when the lane has no rows)
                // pull the lane's next row; false when it has none left.  The entry after it is requested right away, so
                // its LDS latency is hidden behind the row's candidates.
                auto next_row = [&]() -> bool {
                    if (idx >= cnt) return false;
                    const uint2 e = e_next;
                    ++idx;
                    e_next = tab[(idx < TAB_ROWS ? idx : TAB_ROWS - 1) * 64 + tid];
                    rem = (int)(e.x >> TAB_R0_BITS);
                    al = __uint_as_float(e.y);
                    ptr = q + (e.x & (TAB_MAX_RAYS - 1u));
                    return true;
                };
                bool live = next_row();
                while (live) {
                    XVR_STAT_WAVE(6);
                    const float4 ta = XVR_LOAD_Q(ptr, 0), tb = XVR_LOAD_Q(ptr, 1);
                    gather_pair(ta, tb, al, al, al, Bx, By, Bz, acc);
                    ptr += 2;
                    rem -= 2;
                    if (rem <= 0) live = next_row();
                }
            }

            // ---- slow path (rare: ~2 % of the wavefront visits have such a lane): the rows phase A could not enter, by the
            // nested loops of the round-1 kernel
            if (__any(k_ovf != INT32_MAX)) {
                for (int k = k_ovf; k <= khi; ++k) {   // (k_ovf = INT32_MAX: no trip)
                    const float al = linspace_sel(k, N, near_, far_, step);
                    if (al > 1e-12f) {
                        StepC S;
                        step_setup(al, S);
                        const int ifirst = (k == k_ovf && i_ovf != INT32_MIN) ? i_ovf : S.ilo;
                        for (int i = ifirst; i <= S.ihi; ++i) {
                            int jlo, jhi;
                            row_setup(S, i, jlo, jhi);
                            XVR_STAT(5, jhi >= jlo ? 1 : 0);
                            if (jhi >= jlo) gather_row(q + (size_t)i * G.qs + jlo, jhi - jlo + 1, al, al, al, Bx, By, Bz, acc);
                        }
                    } else {
                        // alpha_k = 0: every ray's sample sits on the source; all pixels are candidates for the blocks
                        // whose support contains it (a source inside the volume only)
                        const bool hit = fabsf(Cx) < HS && fabsf(Cy) < HS && fabsf(Cz) < HS;
                        const int nall = hit ? G.qn : 0;   // (the rows' closing elements carry weight 0)
                        for (int r = 0; r < nall; ++r) {
                            const float4 t = q[r];
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const float ox = (float)(e >> 2 & 1), oy = (float)(e >> 1 & 1), oz = (float)(e & 1);
                                const float ux = hat01(fmaf(al, t.x, Bx - ox));
                                const float uy = hat01(fmaf(al, t.y, By - oy));
                                const float uz = hat01(fmaf(al, t.z, Bz - oz));
                                acc[e] = fmaf(ux * uy * uz, t.w, acc[e]);
                            }
                        }
                    }
                }
            }
        }
    }
#ifdef XVR_GATHER_STATS
    for (int i = 0; i < 8; ++i)
        if (st[i]) atomicAdd(&g_gather_stats[i], st[i]);
#endif
    add_block8(G.gvol, acc, vx, vy, vz, G.D0, G.D1, G.D2);
}

#include "drr_splat.hiph"   // k_trilinear_splat_b16, k_trilinear_splat_px: the brick-local fixed-point splats (the default)
#include "drr_siddon_splat.hiph"   // k_siddon_splat: the ray-driven brick splat of the Siddon voxel gradient (non-exact index maps)

// ---------------------------------------------------------------------------------------------
// Pixel-major voxel gather for the renders the lattice-of-planes kernels above cannot take (round 2):
//   CLIP  spec.clip_to_volume: alpha_k = alpha_min(ray) + u_k (alpha_max - alpha_min)(ray) -- the samples of a step no longer
//         lie on one plane, so there is no per-step row table; the image is scaled by the ray's span.
//   MASK  mask -> channels with a gradient that differs between channels: the upstream value of a sample is
//         gout[b][label(sample)][ray], looked up per candidate (a gradient that is the same for all channels -- all xvr
//         ever produces, trainer.py:292-293 -- is the unmasked gradient and takes the table kernel).
// One lane owns a 2x2x2 voxel block; per pose the pixel window is the bounding box of the 8 projected corners of the
// block's interpolation support (as in k_siddon_gather_vol2); per pixel the ray is clipped against that support box
// (slab test) and only the samples inside are evaluated -- with the forward's own alpha and position arithmetic, so the
// weights are the forward's interpolation weights.  Before these kernels both cases fell back to the fp32-atomic scatter
// (443 ms per C2 batch).
// ---------------------------------------------------------------------------------------------
template <bool CLIP, bool MASK>
__global__ __launch_bounds__(64) void k_trilinear_gather_px(GatherArgs G) {
    if (*G.flag > __float_as_uint(GATHER_DEV_TOL)) return;  // not a lattice: the scatter kernel runs instead
    constexpr float HS = 1.5f, CO = 0.5f;
    int bx, by, bz;
    brick_coords(blockIdx.x, G.D1, G.D2, G.bd, bx, by, bz);
    const int tid = threadIdx.x;
    const int vx = (bx * 4 + (tid >> 4)) * 2, vy = (by * 4 + ((tid >> 2) & 3)) * 2, vz = (bz * 4 + (tid & 3)) * 2;
    const bool inb = vx < G.D0 && vy < G.D1 && vz < G.D2;
    const float fv[3] = {(float)vx, (float)vy, (float)vz};
    const float a0 = G.sp.a[0], a1 = G.sp.a[1], a2 = G.sp.a[2];
    float xv[3];  // block centre in x coordinates
#pragma unroll
    for (int i = 0; i < 3; ++i) xv[i] = (fv[i] + CO - G.sp.b[i]) / G.sp.a[i];
    const float ea0 = HS / a0, ea1 = HS / a1, ea2 = HS / a2;   // half-size of the support, in x coordinates
    const int N = G.sp.n_points;
    const float near_ = spec_window(G.sp).near_, far_ = spec_window(G.sp).far_;
    const float step = N > 1 ? (far_ - near_) / (float)(N - 1) : 0.f;
    const float inv_step = step > 0.f ? 1.f / step : 0.f;
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;

    for (int wd = 0; wd < G.words; ++wd) {
        unsigned bits = G.cull[(size_t)blockIdx.x * G.words + wd];
        while (bits) {
            const int p = wd * 32 + __builtin_ctz(bits);
            bits &= bits - 1;
            const PoseLattice& P = G.poses[p];
            const float s0 = P.s[0], s1 = P.s[1], s2 = P.s[2];
            const float w0 = xv[0] - s0, w1 = xv[1] - s1, w2 = xv[2] - s2;
            const float av = P.nh[0] * w0 + P.nh[1] * w1 + P.nh[2] * w2;
            const float en0 = P.nh[0] * ea0, en1 = P.nh[1] * ea1, en2 = P.nh[2] * ea2;
            const float da = fabsf(en0) + fabsf(en1) + fabsf(en2);
            const float amin = av - da, amax = av + da;   // alpha range of the support box on this pose's sample planes
            int jlo = 0, jhi = -1, ilo = 0, ihi = -1;
            if (inb && amin > 1e-6f && amax >= G.cull_lo && amin <= G.cull_hi) {
                // convex and in front of the source: the pixel window is the bounding box of the 8 projected corners
                const float nj = P.gc[0] * w0 + P.gc[1] * w1 + P.gc[2] * w2;
                const float ni = P.gr[0] * w0 + P.gr[1] * w1 + P.gr[2] * w2;
                const float ec0 = P.gc[0] * ea0, ec1 = P.gc[1] * ea1, ec2 = P.gc[2] * ea2;
                const float er0 = P.gr[0] * ea0, er1 = P.gr[1] * ea1, er2 = P.gr[2] * ea2;
                float jmn = INFINITY, jmx = -INFINITY, imn = INFINITY, imx = -INFINITY;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float sx = (c & 4) ? 1.f : -1.f, sy = (c & 2) ? 1.f : -1.f, sz = (c & 1) ? 1.f : -1.f;
                    const float inv = 1.f / (av + sx * en0 + sy * en1 + sz * en2);
                    const float jv = (nj + sx * ec0 + sy * ec1 + sz * ec2) * inv, iv = (ni + sx * er0 + sy * er1 + sz * er2) * inv;
                    jmn = fminf(jmn, jv); jmx = fmaxf(jmx, jv);
                    imn = fminf(imn, iv); imx = fmaxf(imx, iv);
                }
                jlo = (int)ceilf(fmaxf(jmn + P.gc0 - GATHER_WIN_MARGIN, 0.f));
                jhi = (int)floorf(fminf(jmx + P.gc0 + GATHER_WIN_MARGIN, (float)(G.W - 1)));
                ilo = (int)ceilf(fmaxf(imn + P.gr0 - GATHER_WIN_MARGIN, 0.f));
                ihi = (int)floorf(fminf(imx + P.gr0 + GATHER_WIN_MARGIN, (float)(G.H - 1)));
            } else if (inb && amin <= 1e-6f && amax >= G.cull_lo) {
                jhi = G.W - 1;   // the box reaches the source plane: no perspective bound -- visit every ray
                ihi = G.H - 1;
            }
            const float4* __restrict__ q = G.q + (size_t)p * G.qn;
            const float2* __restrict__ q2 = G.q2 + (size_t)p * G.n;
            // a s + b - v for the block's first voxel (the weights' constant) and for its centre (the slab test's)
            const float Bx = fmaf(a0, s0, G.sp.b[0] - fv[0]), By = fmaf(a1, s1, G.sp.b[1] - fv[1]), Bz = fmaf(a2, s2, G.sp.b[2] - fv[2]);
            const float Cx = Bx - CO, Cy = By - CO, Cz = Bz - CO;
            for (int i = ilo; i <= ihi; ++i) {
                for (int j = jlo; j <= jhi; ++j) {
                    const float4 t = q[(size_t)i * G.qs + j];   // a * d, (g *) L / N (* span)
                    float lo = 0.f, span = 1.f;
                    if (CLIP) {
                        const float2 ab = q2[(size_t)i * G.W + j];
                        lo = ab.x;
                        span = ab.y - ab.x;      // exactly the forward's (amax - amin)
                    }
                    // alphas where the ray is inside the support box: |C + alpha t| < HS on all three axes (an axis the ray
                    // does not move along: inside for every alpha or for none)
                    const float ix = fabsf(t.x) < 1e-12f ? copysignf(1e12f, t.x) : 1.f / t.x;
                    const float iy = fabsf(t.y) < 1e-12f ? copysignf(1e12f, t.y) : 1.f / t.y;
                    const float iz = fabsf(t.z) < 1e-12f ? copysignf(1e12f, t.z) : 1.f / t.z;
                    const float x0 = (-HS - Cx) * ix, x1 = (HS - Cx) * ix, y0 = (-HS - Cy) * iy, y1 = (HS - Cy) * iy;
                    const float z0 = (-HS - Cz) * iz, z1 = (HS - Cz) * iz;
                    const float e0 = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
                    const float e1 = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
                    // -> sample indices (u = near + k step; alpha = lo + u * span under CLIP, u otherwise)
                    int klo = 0, khi = -1;
                    if (e1 >= e0 && (!CLIP || span > 0.f)) {
                        const float u0 = CLIP ? (e0 - lo) / span : e0, u1 = CLIP ? (e1 - lo) / span : e1;
                        if (step > 0.f) {
                            klo = (int)ceilf(fmaxf((u0 - near_) * inv_step - 1.f, 0.f));      // one step of slack each side: the
                            khi = (int)floorf(fminf((u1 - near_) * inv_step + 1.f, (float)(N - 1)));   // weights decide, not the window
                        } else {
                            khi = 0;
                        }
                    }
                    for (int k = klo; k <= khi; ++k) {
                        const float u = linspace_at(k, N, near_, far_, step);
                        const float al = CLIP ? fmaf(u, span, lo) : u;   // the forward's alpha, bit for bit
                        const float dx = fmaf(al, t.x, Bx), dy = fmaf(al, t.y, By), dz = fmaf(al, t.z, Bz);
                        float cw = t.w;
                        if (MASK) {
                            // the channel of the sample = the label of its nearest voxel (0 outside the volume), as the forward
                            const int lx = (int)rintf(dx + fv[0]), ly = (int)rintf(dy + fv[1]), lz = (int)rintf(dz + fv[2]);
                            const bool in = (unsigned)lx < (unsigned)G.D0 && (unsigned)ly < (unsigned)G.D1 && (unsigned)lz < (unsigned)G.D2;
                            const int lab = in ? min(max((int)G.mask[((size_t)lx * G.D1 + ly) * G.D2 + lz], 0), G.C - 1) : 0;
                            cw *= G.gout[((size_t)p * G.C + lab) * G.n + (size_t)i * G.W + j];
                        }
                        const float ux0 = hat01(dx), uy0 = hat01(dy), uz0 = hat01(dz) * cw;
                        const float ux1 = hat01(dx - 1.f), uy1 = hat01(dy - 1.f), uz1 = hat01(dz - 1.f) * cw;
                        const float p00 = ux0 * uy0, p01 = ux0 * uy1, p10 = ux1 * uy0, p11 = ux1 * uy1;
                        acc[0] = fmaf(p00, uz0, acc[0]);
                        acc[1] = fmaf(p00, uz1, acc[1]);
                        acc[2] = fmaf(p01, uz0, acc[2]);
                        acc[3] = fmaf(p01, uz1, acc[3]);
                        acc[4] = fmaf(p10, uz0, acc[4]);
                        acc[5] = fmaf(p10, uz1, acc[5]);
                        acc[6] = fmaf(p11, uz0, acc[6]);
                        acc[7] = fmaf(p11, uz1, acc[7]);
                    }
                }
            }
        }
    }
    add_block8(G.gvol, acc, vx, vy, vz, G.D0, G.D1, G.D2);
}

// ---------------------------------------------------------------------------------------------
// Siddon voxel gradient under a mask whose upstream gradient DIFFERS between channels (round 3; exact-geometry index map).
// Siddon credits a segment to ONE voxel -- in exact geometry the voxel whose box holds it -- so the channel of everything a
// voxel receives is the voxel's own label: one lane owns one voxel (256 lanes on a 4 x 8 x 8 brick: the round-1 gather, whose
// 2x2x2-block successor would need eight labels per lane), gathers L x (length of every candidate ray inside its box,
// clipped to the ray's own [alpha_lo, alpha_hi]) and weights it with gout[pose][label][ray].  q.w holds L alone
// (k_gather_prep under a mask).  No atomics, deterministic; before this the case fell back to the fp32-atomic scatter.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WG) void k_siddon_gather_mask(GatherArgs G) {
    if (*G.flag > __float_as_uint(GATHER_DEV_TOL)) return;
    int bx, by, bz;
    brick_coords(blockIdx.x, G.D1, G.D2, G.bd, bx, by, bz);
    // 256 lanes on 4 x 8 x 8 voxels (one voxel per lane: the per-pose setup dominates, so a larger
    // workgroup that amortises the cull words and pose constants wins here -- 4^3 bricks measured 10 % slower)
    const int tid = threadIdx.x;
    const int vx = bx * 4 + (tid >> 6), vy = by * 8 + ((tid >> 3) & 7), vz = bz * 8 + (tid & 7);
    const bool inb = vx < G.D0 && vy < G.D1 && vz < G.D2;
    // the voxel's channel, by the forward's rule (label clamped into [0, C - 1])
    const int lab = inb ? min(max((int)G.mask[((size_t)vx * G.D1 + vy) * G.D2 + vz], 0), G.C - 1) : 0;
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
            const float* __restrict__ go = G.gout + ((size_t)p * G.C + lab) * G.n;
            for (int i = ilo; i <= ihi; ++i) {
                for (int j = jlo; j <= jhi; ++j) {
                    const float4 t = q[(size_t)i * G.W + j];
                    const float2 ab = q2[(size_t)i * G.W + j];
                    const float x0 = lx * t.x, x1 = hx * t.x, y0 = ly * t.y, y1 = hy * t.y, z0 = lz * t.z, z1 = hz * t.z;
                    float en = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
                    float ex = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
                    en = fmaxf(en, ab.x);
                    ex = fminf(ex, ab.y);
                    const float len = fmaxf(ex - en, 0.f);
                    if (len > 0.f) acc = fmaf(len * t.w, go[(size_t)i * G.W + j], acc);
                }
            }
        }
    }
    if (inb && acc != 0.f) G.gvol[((size_t)vx * G.D1 + vy) * G.D2 + vz] += acc;
}


// ---------------------------------------------------------------------------------------------
// Siddon voxel gradient for NON-exact index maps (norm_dims_offset = +-1, align_corners = True: the variants SURVEY.md
// Appendix A recalls for upstream), round 2.  The voxel a segment is credited to is rint(a x_mid + b) of its MIDPOINT, which
// inside plane cell c is c + olo or c + olo + 1 per axis (the map drifts by less than a voxel over the volume:
// siddon_cell_offsets).  One lane owns one CELL and gathers, as k_siddon_gather_vol2 does for its voxels, the length of every
// ray inside it -- but splits it over 8 sums by where the forward's own midpoint arithmetic sends the segment.  The sums go
// to a [cell][8] scratch; k_siddon_cells_to_voxels then adds, for every voxel, the eight (cell, octant) entries that
// name it.  No atomics, deterministic; before this the non-exact maps took the fp32-atomic scatter (151 ms per C3 batch).
// (Round 3 built the 2 x 2 x 2-cells-per-lane version in k_siddon_gather_vol2's frame -- window, loads and the nine crossing
//  alphas shared by eight cells, 64 sums per lane: 34.2 ms against this kernel's 30.3.  Unlike the exact map's three
//  instructions per voxel, a crossed cell costs ~30 (the forward's midpoint rule); a block's window holds 16 candidates that each
//  cross ~3 of its 8 cells, a cell's own window 9 that mostly cross it; and 138 registers leave 3 wavefronts per SIMD.)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WG) void k_siddon_gather_cells(GatherArgs G) {
    if (*G.flag > __float_as_uint(GATHER_DEV_TOL)) return;
    int bx, by, bz;
    brick_coords(blockIdx.x, G.D1, G.D2, G.bd, bx, by, bz);
    const int tid = threadIdx.x;   // 256 lanes on 4 x 8 x 8 cells
    const int vx = bx * 4 + (tid >> 6), vy = by * 8 + ((tid >> 3) & 7), vz = bz * 8 + (tid & 7);
    const bool inb = vx < G.D0 && vy < G.D1 && vz < G.D2;
    const float p0x = (float)vx + G.sp.plane0[0], p0y = (float)vy + G.sp.plane0[1], p0z = (float)vz + G.sp.plane0[2];
    const float p1x = (float)(vx + 1) + G.sp.plane0[0], p1y = (float)(vy + 1) + G.sp.plane0[1],
                p1z = (float)(vz + 1) + G.sp.plane0[2];
    const float cx = p0x + 0.5f, cy = p0y + 0.5f, cz = p0z + 0.5f;
    const float a0 = G.sp.a[0], a1 = G.sp.a[1], a2 = G.sp.a[2], b0 = G.sp.b[0], b1 = G.sp.b[1], b2 = G.sp.b[2];
    // voxel index of the "1" octant along every axis
    const float t1x = (float)(vx + G.olo[0] + 1), t1y = (float)(vy + G.olo[1] + 1), t1z = (float)(vz + G.olo[2] + 1);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
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
            const float nj = P.gc[0] * w0 + P.gc[1] * w1 + P.gc[2] * w2, dnj = 0.5f * P.hwc;
            const float ni = P.gr[0] * w0 + P.gr[1] * w1 + P.gr[2] * w2, dni = 0.5f * P.hwr;
            int jlo = 0, jhi = -1, ilo = 0, ihi = -1;
            if (inb && amin > 1e-6f && amax >= 0.f && amin <= 1.f) {
                const float i0 = 1.f / amin, i1 = 1.f / amax;
                const float ja = (nj - dnj) * i0, jb = (nj - dnj) * i1, jc = (nj + dnj) * i0, jd = (nj + dnj) * i1;
                const float ia = (ni - dni) * i0, ib = (ni - dni) * i1, ic = (ni + dni) * i0, id = (ni + dni) * i1;
                jlo = (int)ceilf(fmaxf(fminf(fminf(ja, jb), fminf(jc, jd)) + P.gc0 - GATHER_WIN_MARGIN, 0.f));
                jhi = (int)floorf(fminf(fmaxf(fmaxf(ja, jb), fmaxf(jc, jd)) + P.gc0 + GATHER_WIN_MARGIN, (float)(G.W - 1)));
                ilo = (int)ceilf(fmaxf(fminf(fminf(ia, ib), fminf(ic, id)) + P.gr0 - GATHER_WIN_MARGIN, 0.f));
                ihi = (int)floorf(fminf(fmaxf(fmaxf(ia, ib), fmaxf(ic, id)) + P.gr0 + GATHER_WIN_MARGIN, (float)(G.H - 1)));
            } else if (inb && amin <= 1e-6f && amax >= 0.f) {
                jhi = G.W - 1;   // the cell reaches the source plane: no perspective bound -- visit every ray
                ihi = G.H - 1;
            }
            const float lx = p0x - s0, ly = p0y - s1, lz = p0z - s2;
            const float hx = p1x - s0, hy = p1y - s1, hz = p1z - s2;
            const float4* __restrict__ q = G.q + (size_t)p * G.n;
            const float4* __restrict__ qd = G.q + (size_t)G.B * G.n + (size_t)p * G.n;
            const float2* __restrict__ q2 = G.q2 + (size_t)p * G.n;
            for (int i = ilo; i <= ihi; ++i) {
                for (int j = jlo; j <= jhi; ++j) {
                    const size_t r = (size_t)i * G.W + j;
                    const float4 t = q[r];
                    const float2 ab = q2[r];
                    const float x0 = lx * t.x, x1 = hx * t.x, y0 = ly * t.y, y1 = hy * t.y, z0 = lz * t.z, z1 = hz * t.z;
                    float en = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
                    float ex = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
                    en = fmaxf(en, ab.x);
                    ex = fminf(ex, ab.y);
                    const float len = fmaxf(ex - en, 0.f) * t.w;
                    if (len != 0.f) {
                        // the forward's midpoint rule (k_siddon, EXACT = false), arithmetic and all
                        const float4 dd = qd[r];
                        const float mid = 0.5f * (en + ex);
                        const bool ox = rintf(fmaf(a0, fmaf(mid, dd.x, s0), b0)) >= t1x;
                        const bool oy = rintf(fmaf(a1, fmaf(mid, dd.y, s1), b1)) >= t1y;
                        const bool oz = rintf(fmaf(a2, fmaf(mid, dd.z, s2), b2)) >= t1z;
                        const float l1 = ox ? len : 0.f, l0 = len - l1;
                        const float l01 = oy ? l0 : 0.f, l00 = l0 - l01, l11 = oy ? l1 : 0.f, l10 = l1 - l11;
                        const float c001 = oz ? l00 : 0.f, c011 = oz ? l01 : 0.f, c101 = oz ? l10 : 0.f, c111 = oz ? l11 : 0.f;
                        acc[0] += l00 - c001; acc[1] += c001;
                        acc[2] += l01 - c011; acc[3] += c011;
                        acc[4] += l10 - c101; acc[5] += c101;
                        acc[6] += l11 - c111; acc[7] += c111;
                    }
                }
            }
        }
    }
    if (inb) {
        float4* out = reinterpret_cast<float4*>(G.cells + (((size_t)vx * G.D1 + vy) * G.D2 + vz) * 8);
        out[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
        out[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
    }
}

// voxel v <- sum over the octants e = (ex, ey, ez) of cell (v - olo - e), entry e
__global__ __launch_bounds__(WG) void k_siddon_cells_to_voxels(GatherArgs G) {
    if (*G.flag > __float_as_uint(GATHER_DEV_TOL)) return;
    const long long v = (long long)blockIdx.x * WG + threadIdx.x;
    const long long nvox = (long long)G.D0 * G.D1 * G.D2;
    if (v >= nvox) return;
    const int z = (int)(v % G.D2), y = (int)((v / G.D2) % G.D1), x = (int)(v / ((long long)G.D1 * G.D2));
    float tot = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int cxi = x - G.olo[0] - (e >> 2), cyi = y - G.olo[1] - ((e >> 1) & 1), czi = z - G.olo[2] - (e & 1);
        if ((unsigned)cxi < (unsigned)G.D0 && (unsigned)cyi < (unsigned)G.D1 && (unsigned)czi < (unsigned)G.D2)
            tot += G.cells[(((size_t)cxi * G.D1 + cyi) * G.D2 + czi) * 8 + e];
    }
    if (tot != 0.f) G.gvol[v] += tot;
}

// Same gather with a 2 x 2 x 2 voxel block per lane (one wavefront per 8^3 brick, as the trilinear gather):
// the per-pose window and the candidate's loads are paid once for eight voxels, the three planes per axis give
// nine crossing alphas per candidate (the forward's expression, plane by plane), from which every voxel's
// entry / exit are one max3 / min3.
// (Round 2 measured the per-lane FLATTENED window loop here too -- lane-private (row, column) cursor, same arithmetic:
//  13.5 ms against 12.2 ms.  Fewer trips, but the lanes of a wavefront drift onto different detector rows, and the four
//  loads of a trip then touch that many more cache lines; the nested loops keep the wavefront on one row at a time.)
// Round 4, second pass.  The kernel is vector-issue bound and a (lane, pose) visit sees only ~5 candidate rays, so the
// visit's set-up weighs as much as its candidates.  Counted in the ISA: 236 vector instructions per visit (80 of them the
// eight IEEE divisions of the projected corners) and 143 per trip of two candidates.  Now:
//   * the pixel window from ONE projection: j(w + sum s_k e_k) - jc = (A - jc B) / (av + B) with A = sum s_k ec_k,
//     B = sum s_k en_k over the corner signs s_k, so |j - jc| <= sum_k |ec_k - jc en_k| / (av - da) -- rigorous, two
//     v_rcp_f32 (1 ulp: 3e-5 pixel, the window has GATHER_WIN_MARGIN) and wider than the corners' bounding box by da / av
//     ~ 1e-3 of itself;
//   * a candidate's eight entries / exits as v_max3_f32 / v_min3_f32 of the per-axis interval ends (the compiler shares
//     max(xl, yl) between two voxels instead: 24 two-operand instructions where 16 three-operand ones do); plain fp32
//     throughout -- a packed pair (v_pk_*_f32) occupies the SIMD for 1.3 x two plain instructions on this chip;
//   * the loop body exists twice: poses none of whose rays is cut at alpha = 0 / 1 neither load nor apply (lo, hi);
//   * ... and twice again: where every ray of every lane's window runs the same way along each axis (the direction is affine in
//     the pixel: its sign at the window's four corners decides, with a margin for the lattice tolerance), the lane's planes are
//     put in the order the rays cross them and the per-axis interval ends ARE the plane alphas -- the twelve min / max per
//     candidate go.  The lane's eight sums are then kept in crossing order too (index e ^ mask, permuted by conditional swaps
//     when a pose's mask differs from the previous one's), so that every sum still adds its candidates in the same order:
//     the result is the general body's, bit for bit.  A wavefront with a lane whose window straddles a sign change (the
//     pixels around the principal ray, for two of the three axes) takes the general body for that pose.
// ~53 vector instructions per candidate (57 with the cut) and ~110 per visit.
__device__ __forceinline__ float max3_raw(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float min3_raw(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float max_raw(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float min_raw(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

//   * FAST: a workgroup is four wavefronts on four bricks IN A ROW ALONG THE VIEWING DIRECTION (the volume axis the first
//     pose's detector normal is closest to: every workgroup derives the same axis).  Their pixel footprints overlap, they
//     run on one CU, and a q line fetched for one brick is in the L1 for the other three: on its own a wavefront finds
//     almost nothing of its ~40 lines per visit in the L1 (1.15e9 L1-miss lines, 148 GB from L2 per C3 launch).
//   * FAST: the candidates come from LDS.  With one 16-byte load per lane and candidate, each lane at its own address, the kernel
//     is bound by the texture-address / L1 path (64 addresses per wavefront load: ~64 clocks of the CU's one TA each, 19 such
//     loads per visit -- the vector-instruction savings above moved the launch by 2 % each until this was gone).  The 64
//     lanes' windows overlap almost entirely: a visit reads ~900 (lane, candidate) pairs out of the ~300 pixels of the brick's
//     footprint.  So the wavefront first copies the footprint -- the pixel window of the whole 8^3 brick, by the same
//     one-projection bound, a few coalesced loads -- into its slice of LDS and the lanes read their candidates with
//     ds_read_b128.  Footprints beyond the slice (SG_FOOT pixels), bricks that reach the source plane and poses with cut rays
//     take the global loads as before.
#ifndef XVR_SG_CUM     // 1: sign-sorted visits accumulate 3-D prefix sums of the eight sums (see the candidate)
#define XVR_SG_CUM 1
#endif
#ifndef XVR_SG_FOOT
#define XVR_SG_FOOT 480
#endif
#ifndef XVR_SG_WAVES   // wavefronts per SIMD the LDS slices leave room for: 160 KB / (4 x 16 B x SG_FOOT) workgroups of four.  Measured at
                       // C3 (ms, launch incl. prep / cull): 640 px x 4 -> 8.20, 512 x 5 (only four fit) 8.24, 416-496 x 5 -> 7.63, 400 x 6 -> 8.38
                       // (spills), 320 x 7 -> 8.88, 256 x 8 -> 12.2, 960-1280 x 2 -> 12.9: occupancy matters more than the last footprints
#define XVR_SG_WAVES 5
#endif
constexpr int SG_FOOT = XVR_SG_FOOT;
template <bool FAST>
__global__ __launch_bounds__(FAST ? 256 : 64) __attribute__((amdgpu_waves_per_eu(FAST ? XVR_SG_WAVES : 6, FAST ? XVR_SG_WAVES : 6))) void k_siddon_gather_vol2(GatherArgs G) {
    if (*G.flag > __float_as_uint(GATHER_DEV_TOL)) return;
    __shared__ float4 s_foot[FAST ? 4 : 1][FAST ? SG_FOOT : 1];
    int bx, by, bz;
    int brick = blockIdx.x;   // the brick's number in the cull array's order (brick_coords)
    if (FAST) {
        // (no array indexed by the axis: it would go through scratch and the brick number -- with it every per-pose load --
        //  would stop being a scalar)
        const int nb0 = (G.D0 + G.bd[0] - 1) / G.bd[0], nb1 = (G.D1 + G.bd[1] - 1) / G.bd[1], nb2 = (G.D2 + G.bd[2] - 1) / G.bd[2];
        const float n0 = fabsf(G.poses[0].nh[0] / G.sp.a[0]), n1 = fabsf(G.poses[0].nh[1] / G.sp.a[1]), n2 = fabsf(G.poses[0].nh[2] / G.sp.a[2]);
        const int ax = __builtin_amdgcn_readfirstlane((n0 >= n1 && n0 >= n2) ? 0 : (n1 >= n2 ? 1 : 2));   // (the same in every workgroup)
        // the other two axes: u, and v the faster one
        const int nba = ax == 0 ? nb0 : (ax == 1 ? nb1 : nb2), nbu = ax == 0 ? nb1 : nb0, nbv = ax == 2 ? nb1 : nb2;
        const int groups = (nba + 3) >> 2;                                         // runs of four bricks along ax
        int g = blockIdx.x;
        const int cv = g % nbv; g /= nbv;
        const int ga = g % groups, cu = g / groups;
        const int ca = ga * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (scalar: everything per brick and pose below is uniform)
        if (ca >= nba || cu >= nbu) return;   // (no barrier below)
        bx = ax == 0 ? ca : cu;
        by = ax == 1 ? ca : (ax == 0 ? cu : cv);
        bz = ax == 2 ? ca : cv;
        brick = (bx * nb1 + by) * nb2 + bz;
    } else {
        brick_coords(blockIdx.x, G.D1, G.D2, G.bd, bx, by, bz);
    }
    const int tid = threadIdx.x & 63;
    float4* const foot = s_foot[FAST ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0];
    const int vx = (bx * 4 + (tid >> 4)) * 2, vy = (by * 4 + ((tid >> 2) & 3)) * 2, vz = (bz * 4 + (tid & 3)) * 2;
    const bool inb = vx < G.D0 && vy < G.D1 && vz < G.D2;
    // the brick's centre (uniform), in x coordinates: the middle of its 4 x 4 x 4 block centres
    const float cbx = (float)(bx * 8 + 4) + G.sp.plane0[0], cby = (float)(by * 8 + 4) + G.sp.plane0[1], cbz = (float)(bz * 8 + 4) + G.sp.plane0[2];
    // the three planes per axis that bound the block's voxels, and the block centre, in x coordinates
    float px[3], py[3], pz[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        px[k] = (float)(vx + k) + G.sp.plane0[0];
        py[k] = (float)(vy + k) + G.sp.plane0[1];
        pz[k] = (float)(vz + k) + G.sp.plane0[2];
    }
    const float cx = px[1], cy = py[1], cz = pz[1];
    float acc[8];   // voxels (a, b, c)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    unsigned cur = 0;   // the sum of voxel e lives in acc[e ^ cur]
    auto permute = [&](const unsigned m) {   // acc[e] <- acc[e ^ m]
        if (__builtin_amdgcn_ballot_w64(m != 0u) == 0ull) return;
#pragma unroll
        for (int bit = 4; bit >= 1; bit >>= 1) {
            const bool sw = (m & (unsigned)bit) != 0u;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (!(e & bit)) {
                    const float lo_ = acc[e], hi_ = acc[e | bit];
                    acc[e] = sw ? hi_ : lo_;
                    acc[e | bit] = sw ? lo_ : hi_;
                }
        }
    };
    bool cum = false;   // (uniform) acc holds the 3-D prefix sums of the eight sums, in the orientation `cur`
    auto to_cum = [&]() {
#pragma unroll
        for (int bit = 1; bit <= 4; bit <<= 1)
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (e & bit) acc[e] += acc[e ^ bit];
        cum = true;
    };
    auto to_chords = [&]() {
#pragma unroll
        for (int bit = 1; bit <= 4; bit <<= 1)
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (e & bit) acc[e] -= acc[e ^ bit];
        cum = false;
    };
#ifdef XVR_GATHER_STATS   // 0 (lane, pose) visits with a window . 2 wavefront rows . 4 candidates . 6 wavefront trips . 7 wavefront visits
    unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    const float ea0 = 1.f / G.sp.a[0], ea1 = 1.f / G.sp.a[1], ea2 = 1.f / G.sp.a[2];   // half a block, in x coordinates
    for (int wd = 0; wd < G.words; ++wd) {
        unsigned bits = G.cull[(size_t)brick * G.words + wd];
        while (bits) {
            const int p = wd * 32 + __builtin_ctz(bits);
            bits &= bits - 1;
            const PoseLattice& P = G.poses[p];
            // (uniform) does any ray of this pose end at alpha = 0 or 1 (source or detector inside the volume)?  If none does, no
            // voxel's chord needs the clamp to [0, 1]
            const bool cut_rays = !G.cmax || G.cmax[(size_t)p * G.cmax_stride] != 0u;
            const float s0 = P.s[0], s1 = P.s[1], s2 = P.s[2];
            const float w0 = cx - s0, w1 = cy - s1, w2 = cz - s2;
            const float av = P.nh[0] * w0 + P.nh[1] * w1 + P.nh[2] * w2;
            const float da = P.dalpha;   // half-range of alpha over the block
            const float amin = av - da, amax = av + da;
            const float nj = P.gc[0] * w0 + P.gc[1] * w1 + P.gc[2] * w2;
            const float ni = P.gr[0] * w0 + P.gr[1] * w1 + P.gr[2] * w2;
            int jlo = 0, jhi = -1, ilo = 0, ihi = -1;
            if (inb && amin > 1e-6f && amax >= 0.f && amin <= 1.f) {
                const float en0 = P.nh[0] * ea0, en1 = P.nh[1] * ea1, en2 = P.nh[2] * ea2;
                const float ec0 = P.gc[0] * ea0, ec1 = P.gc[1] * ea1, ec2 = P.gc[2] * ea2;
                const float er0 = P.gr[0] * ea0, er1 = P.gr[1] * ea1, er2 = P.gr[2] * ea2;
                float jmn, jmx, imn, imx;
                if (FAST) {
                    const float iav = __builtin_amdgcn_rcpf(av), iam = __builtin_amdgcn_rcpf(amin);
                    const float jc = nj * iav, ic = ni * iav;
                    const float hj = (fabsf(fmaf(-jc, en0, ec0)) + fabsf(fmaf(-jc, en1, ec1)) + fabsf(fmaf(-jc, en2, ec2))) * iam;
                    const float hi = (fabsf(fmaf(-ic, en0, er0)) + fabsf(fmaf(-ic, en1, er1)) + fabsf(fmaf(-ic, en2, er2))) * iam;
                    jmn = jc - hj; jmx = jc + hj; imn = ic - hi; imx = ic + hi;
                } else {
                    // the block is convex and in front of the source: its projection is the hull of its 8 projected
                    // corners, whose bounding box is the exact pixel window
                    jmn = INFINITY; jmx = -INFINITY; imn = INFINITY; imx = -INFINITY;
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const float sx = (c & 4) ? 1.f : -1.f, sy = (c & 2) ? 1.f : -1.f, sz = (c & 1) ? 1.f : -1.f;
                        const float inv = 1.f / (av + sx * en0 + sy * en1 + sz * en2);
                        const float jv = (nj + sx * ec0 + sy * ec1 + sz * ec2) * inv, iv = (ni + sx * er0 + sy * er1 + sz * er2) * inv;
                        jmn = fminf(jmn, jv); jmx = fmaxf(jmx, jv);
                        imn = fminf(imn, iv); imx = fmaxf(imx, iv);
                    }
                }
                jmn += P.gc0 - GATHER_WIN_MARGIN; jmx += P.gc0 + GATHER_WIN_MARGIN;
                imn += P.gr0 - GATHER_WIN_MARGIN; imx += P.gr0 + GATHER_WIN_MARGIN;
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
            // the brick's footprint (uniform): the same bound on the whole brick -- four block half-sizes around its centre.  Every
            // block's rays lie inside it; a lane's own window is a bound too and may stick out by its slack: it is clipped (the
            // pixels dropped are rays that miss the block: exact zeros).
            int Ilo = 0, Jlo = 0, fnc = 0;
            bool staged_v = false;
            if (FAST) {
                const float u0 = cbx - s0, u1 = cby - s1, u2 = cbz - s2;
                const float avB = P.nh[0] * u0 + P.nh[1] * u1 + P.nh[2] * u2, aminB = avB - 4.f * P.dalpha;
                if (aminB > 1e-6f) {
                    const float en0 = P.nh[0] * ea0, en1 = P.nh[1] * ea1, en2 = P.nh[2] * ea2;
                    const float ec0 = P.gc[0] * ea0, ec1 = P.gc[1] * ea1, ec2 = P.gc[2] * ea2;
                    const float er0 = P.gr[0] * ea0, er1 = P.gr[1] * ea1, er2 = P.gr[2] * ea2;
                    const float iav = __builtin_amdgcn_rcpf(avB), iam = 4.f * __builtin_amdgcn_rcpf(aminB);
                    const float jc = (P.gc[0] * u0 + P.gc[1] * u1 + P.gc[2] * u2) * iav, ic = (P.gr[0] * u0 + P.gr[1] * u1 + P.gr[2] * u2) * iav;
                    const float hj = (fabsf(fmaf(-jc, en0, ec0)) + fabsf(fmaf(-jc, en1, ec1)) + fabsf(fmaf(-jc, en2, ec2))) * iam;
                    const float hi = (fabsf(fmaf(-ic, en0, er0)) + fabsf(fmaf(-ic, en1, er1)) + fabsf(fmaf(-ic, en2, er2))) * iam;
                    Jlo = __builtin_amdgcn_readfirstlane((int)ceilf(fmaxf(jc - hj + P.gc0 - GATHER_WIN_MARGIN, 0.f)));
                    Ilo = __builtin_amdgcn_readfirstlane((int)ceilf(fmaxf(ic - hi + P.gr0 - GATHER_WIN_MARGIN, 0.f)));
                    const int Jhi = __builtin_amdgcn_readfirstlane((int)floorf(fminf(jc + hj + P.gc0 + GATHER_WIN_MARGIN, (float)(G.W - 1))));
                    const int Ihi = __builtin_amdgcn_readfirstlane((int)floorf(fminf(ic + hi + P.gr0 + GATHER_WIN_MARGIN, (float)(G.H - 1))));
                    const int fnr = Ihi - Ilo + 1;
                    const int fw = Jhi - Jlo + 1;
                    fnc = fw;        // row stride in the slice (an odd stride, rows starting in different banks, measured no faster)
                    const int total = fnr * fw;
                    if (fnr > 0 && fw > 0 && fnr * fnc <= SG_FOOT) {
                        staged_v = true;
                        XVR_STAT(5, tid == 0 ? total : 0);
                        jlo = max(jlo, Jlo); jhi = min(jhi, Jhi); ilo = max(ilo, Ilo); ihi = min(ihi, Ihi);
                        const float inc = __builtin_amdgcn_rcpf((float)fw);   // ((f + 1/2) / fw is at least 1 / (2 fw) from an integer: 1 ulp is harmless)
                        const float4* __restrict__ src = q + (size_t)Ilo * G.W + Jlo;
                        for (int f = tid; f < total; f += 64) {
                            const int r = (int)(((float)f + 0.5f) * inc), c = f - r * fw;
                            foot[r * fnc + c] = src[(size_t)r * G.W + c];
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                    }
                }
            }
            // one candidate ray into the eight sums: crossing alphas of the planes (forward's expression), per axis the two voxel
            // intervals (the ray's own [alpha_lo, alpha_hi] folded into the x intervals once), per voxel entry, exit, chord
            auto candidate = [&](const float4 t, auto cut, auto sorted) {
                constexpr bool SORTED = decltype(sorted)::value;   // the planes are in crossing order: x0 <= x1 <= x2, ...
                const float x0 = lx[0] * t.x, x1 = lx[1] * t.x, x2 = lx[2] * t.x;
                const float y0 = ly[0] * t.y, y1 = ly[1] * t.y, y2 = ly[2] * t.y;
                const float z0 = lz[0] * t.z, z1 = lz[1] * t.z, z2 = lz[2] * t.z;
                float xl[2] = {SORTED ? x0 : min_raw(x0, x1), SORTED ? x1 : min_raw(x1, x2)}, xh[2] = {SORTED ? x1 : max_raw(x0, x1), SORTED ? x2 : max_raw(x1, x2)};
                if (decltype(cut)::value) {
                    // the ray's own range [max(0, entry), min(1, exit)]: a voxel's cube lies inside the volume's, so only the cut
                    // at the source (alpha = 0) and at the detector (alpha = 1) can shorten its chord -- constants, not a load
                    // of the ray's (lo, hi) (round 4; the loads were a third of a cut pose's)
                    xl[0] = max_raw(xl[0], 0.f); xl[1] = max_raw(xl[1], 0.f);
                    xh[0] = min_raw(xh[0], 1.f); xh[1] = min_raw(xh[1], 1.f);
                }
                const float yl[2] = {SORTED ? y0 : min_raw(y0, y1), SORTED ? y1 : min_raw(y1, y2)}, yh[2] = {SORTED ? y1 : max_raw(y0, y1), SORTED ? y2 : max_raw(y1, y2)};
                const float zl[2] = {SORTED ? z0 : min_raw(z0, z1), SORTED ? z1 : min_raw(z1, z2)}, zh[2] = {SORTED ? z1 : max_raw(z0, z1), SORTED ? z2 : max_raw(z1, z2)};
#if XVR_SG_CUM
                if (SORTED) {
                    // planes in crossing order: the ray is inside {x < x_i, y < y_j, z < z_k} from its entry into the block to the first
                    // of the three planes -- nested prefixes of one ray.  Their lengths are the 3-D PREFIX SUMS of the eight chords, so the
                    // lane keeps prefix sums (one entry alpha for all eight, no per-voxel max3) and takes the differences when the
                    // orientation changes: 25 instead of 32 vector instructions per candidate
                    const float en = max3_raw(xl[0], yl[0], zl[0]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int a = e >> 2, b = (e >> 1) & 1, c = e & 1;
                        acc[e] = fmaf(__builtin_amdgcn_fmed3f(min3_raw(xh[a], yh[b], zh[c]) - en, 0.f, 1.f), t.w, acc[e]);
                    }
                    return;
                }
#endif
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int a = e >> 2, b = (e >> 1) & 1, c = e & 1;
                    const float en = max3_raw(xl[a], yl[b], zl[c]), ex = min3_raw(xh[a], yh[b], zh[c]);
                    // (alphas live in [0, 1]: the [0, 1] clamp is the max with 0, folded into the subtract)
                    acc[e] = fmaf(__builtin_amdgcn_fmed3f(ex - en, 0.f, 1.f), t.w, acc[e]);
                }
            };
#if defined(XVR_SG_ABLATE) && XVR_SG_ABLATE == 3   // diagnostic build only: the visits without their candidates -- WRONG sums
            acc[0] += (float)(ihi - ilo) + (float)(jhi - jlo);
            ihi = ilo - 1;
#endif
            XVR_STAT(0, (ihi >= ilo && jhi >= jlo) ? 1 : 0);
            XVR_STAT(4, (ihi >= ilo && jhi >= jlo) ? (ihi - ilo + 1) * (jhi - jlo + 1) : 0);
            XVR_STAT_WAVE(7);
            const bool staged = __builtin_amdgcn_readfirstlane((int)staged_v) != 0;   // (uniform: the brick and the pose decide)
            auto rows = [&](auto cut, auto sorted, auto from_lds) {
                for (int i = ilo; i <= ihi; ++i) {
                    const float4* __restrict__ row = q + (size_t)i * G.W;
                    const float4* frow = foot + ((i - Ilo) * fnc - Jlo);   // (LDS)
                    XVR_STAT_WAVE(2);
                    // two candidates per trip: the loads are issued before either candidate is evaluated
                    for (int j = jlo; j <= jhi; j += 2) {
                        XVR_STAT_WAVE(6);
                        const int j1 = j < jhi ? j + 1 : j;
                        float4 ta, tb;
#if defined(XVR_SG_ABLATE) && XVR_SG_ABLATE == 1   // diagnostic build only (tools/ablate_siddon_gather.py): no candidate loads -- WRONG sums
                        ta = make_float4(__int_as_float(0x3a000000 + j), __int_as_float(0x3a800000 + i), __int_as_float(0x3a400000 + j + i), 1.f);
                        tb = make_float4(__int_as_float(0x3a000000 + j1), __int_as_float(0x3a800000 + i), __int_as_float(0x3a400000 + j1 + i), 1.f);
#else
                        if (decltype(from_lds)::value) { ta = frow[j]; tb = frow[j1]; } else { ta = row[j]; tb = row[j1]; }
#endif
#if defined(XVR_SG_ABLATE) && XVR_SG_ABLATE == 2   // diagnostic build only: the loads without the candidates' arithmetic -- WRONG sums
                        acc[0] += ta.x + tb.x; acc[1] += ta.y + tb.y; acc[2] += ta.z + tb.z; acc[3] += ta.w + tb.w;
                        continue;
#endif
                        if (j1 == j) tb.w = 0.f;
                        candidate(ta, cut, sorted);
                        candidate(tb, cut, sorted);
                    }
                }
            };
            // do all rays of the window run the same way along every axis?  d(i, j) = st + i er + j ec per axis, affine: its extremes
            // over the window are at the corners; 0.05 of a pixel pitch covers the lattice tolerance (GATHER_DEV_TOL = 0.02)
            bool same_way = true;
            unsigned mask = cur;
            if (FAST) {
                // (the pose's constants are read outside the lane-dependent branch: scalar loads)
                const float stv[3] = {P.st[0], P.st[1], P.st[2]}, erv[3] = {P.er[0], P.er[1], P.er[2]}, ecv[3] = {P.ec[0], P.ec[1], P.ec[2]};
                const float marg = 0.05f * fmaxf(P.ec_len, P.er_len);
                const bool has = ihi >= ilo && jhi >= jlo;
                const float fi0 = (float)ilo, fj0 = (float)jlo, di = (float)(ihi - ilo), dj = (float)(jhi - jlo);
                unsigned m = 0u;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float base = fmaf(fj0, ecv[k], fmaf(fi0, erv[k], stv[k])), ei = di * erv[k], ej = dj * ecv[k];
                    const float dlo = base + fminf(ei, 0.f) + fminf(ej, 0.f), dhi = base + fmaxf(ei, 0.f) + fmaxf(ej, 0.f);
                    same_way = same_way && (!has || dlo > marg || dhi < -marg);
                    m |= (dhi < -marg) ? (4u >> k) : 0u;
                }
                mask = has ? m : cur;   // (a lane without a window keeps its order: nothing to permute)
            }
#ifdef XVR_GATHER_STATS   // 1 wavefront visits served from LDS . 3 wavefront visits on sign-sorted planes . 5 pixels staged
            if (staged) { XVR_STAT_WAVE(1); }
            if (FAST && __builtin_amdgcn_ballot_w64(!same_way) == 0ull) { XVR_STAT_WAVE(3); }
#endif
            if (FAST && __builtin_amdgcn_ballot_w64(!same_way) == 0ull) {
#if XVR_SG_CUM
                if (__builtin_amdgcn_ballot_w64(cur != mask) != 0ull) {   // (uniform) some lane turns its block around
                    if (cum) to_chords();
                    permute(cur ^ mask);
                }
                if (!cum) to_cum();
#else
                permute(cur ^ mask);
#endif
                cur = mask;
                if (mask & 4u) { const float t_ = lx[0]; lx[0] = lx[2]; lx[2] = t_; }
                if (mask & 2u) { const float t_ = ly[0]; ly[0] = ly[2]; ly[2] = t_; }
                if (mask & 1u) { const float t_ = lz[0]; lz[0] = lz[2]; lz[2] = t_; }
                if (staged && cut_rays) rows(std::true_type{}, std::true_type{}, std::true_type{});
                else if (staged) rows(std::false_type{}, std::true_type{}, std::true_type{});
                else if (cut_rays) rows(std::true_type{}, std::true_type{}, std::false_type{});
                else rows(std::false_type{}, std::true_type{}, std::false_type{});
            } else {   // (rare: a lane's window straddles a sign change -- one body for it)
                if (cum) to_chords();
                permute(cur);
                cur = 0u;
                if (staged) rows(std::true_type{}, std::false_type{}, std::true_type{});
                else rows(std::true_type{}, std::false_type{}, std::false_type{});
            }
            if (FAST) __builtin_amdgcn_wave_barrier();   // (the slice is overwritten by the next visit)
        }
    }
    if (cum) to_chords();
    permute(cur);
    add_block8(G.gvol, acc, vx, vy, vz, G.D0, G.D1, G.D2);
#ifdef XVR_GATHER_STATS
    for (int i = 0; i < 8; ++i)
        if (st[i]) atomicAdd(&g_gather_stats[i], st[i]);
#endif
}

// (Round 2 also built the RAY-driven counterpart of the trilinear splat for Siddon -- per (16^3 brick, pose) visit the rays
//  of the brick's pixel footprint walk its voxels with the forward's plane arithmetic and add segment * g * L to fixed-point
//  LDS cells, one ds_add_u32 per segment -- and dropped it: parity-green, 17.5-21 ms at C3 against the 12.2 ms of
//  k_siddon_gather_vol2.  A ray crosses only ~20 voxels of a brick, so the per-(ray, brick) set-up (three times a step) and
//  the divergence between rays of 0..45 steps leave 48 % of the lanes live in the walk and 25 % in the set-up: 1.16e10
//  vector instructions against the gather's 7.3e9.  One lesson kept: where a walk is (re)started mid-ray, the start voxel
//  must come from the forward's own plane alphas, alpha(previous plane) <= start < alpha(next plane), not from the
//  position -- for a ray grazing a plane family the two disagree over a visible stretch, an ulp of the plane's position
//  divided by the direction cosine.)

}  // namespace

// Set up the workspace and launch prep -> cull -> gather.  The caller launches the scatter fallback
// (with skip_unless_flag_gt = the returned flag) right behind it.
int xvr_detail::launch_gather(bool siddon, const float* source, const float* target, const float* raylen, const float* grad_out,
                  int B, int n, int gw, int D0, int D1, int D2, const xvr_drr_spec* sp, float* grad_volume,
                  void* workspace, void* stream, unsigned** flag_out, const float* mask, int C, const int* siddon_olo, int siddon_splat) {
    char* ws = static_cast<char*>(workspace);
    GatherArgs G = {};
    const bool sid_splat = siddon && siddon_splat && !mask;
    if (siddon && siddon_olo) {   // non-exact index map: per-cell octant sums in the scratch behind the regular workspace (the splat needs none)
        if (!sid_splat) G.cells = reinterpret_cast<float*>(ws + align256(ws_bytes(B, n, D0, D1, D2)));
        for (int k = 0; k < 3; ++k) G.olo[k] = siddon_olo[k];
    }
    G.sid_splat = sid_splat ? 1 : 0;
    G.mask = mask;
    G.C = C;
    G.clip = (!siddon && sp->clip_to_volume == 1) ? 1 : 0;
    // alphas any sample can take: [near, far] on the shared planes; under clip alpha = amin + u (amax - amin) with
    // 0 <= amin, amin + span <= 1, i.e. within [min(0, near), max(1, far)]
    G.cull_lo = G.clip ? fminf(0.f, sp->near_) : sp->near_;
    G.cull_hi = G.clip ? fmaxf(1.f, sp->far_) : sp->far_;
    if (siddon) { G.cull_lo = 0.f; G.cull_hi = 1.f; }
    G.source = source; G.target = target; G.raylen = raylen; G.gout = grad_out;
    G.B = B; G.n = n; G.W = gw; G.H = n / gw; G.D0 = D0; G.D1 = D1; G.D2 = D2; G.sp = *sp;
    G.flag = reinterpret_cast<unsigned*>(ws);
    G.poses = reinterpret_cast<PoseLattice*>(ws + ws_pose_off());
    G.q = reinterpret_cast<float4*>(ws + ws_q_off(B));
    G.q2 = reinterpret_cast<float2*>(ws + ws_q2_off(B, n));
    G.siddon = siddon ? 1 : 0;
    G.qs = siddon ? gw : gw + 1;
    G.qn = siddon ? n : (n / gw) * (gw + 1);
    G.V = siddon ? 1 : 2;   // voxels per lane and axis
    // trilinear without clip / per-channel masks: the brick-local fixed-point splat on 16^3 bricks (k_trilinear_splat_b16)
    // unless the option "gather_splat" is 0 (A/B switch: the fp32 voxel-driven table gather)
    const int splat_mode = xvr_detail::option(xvr_detail::OPT_GATHER_SPLAT);   // (2: the ray-major splat for every render, A/B)
    const bool use_splat = splat_mode != 0;
    const bool splat = !siddon && (splat_mode == 1 || splat_mode == 3) && sp->clip_to_volume != 1 && !mask;
    if (siddon && (G.cells || G.mask)) { G.bd[0] = 4; G.bd[1] = 8; G.bd[2] = 8; }
    else if (siddon) G.bd[0] = G.bd[1] = G.bd[2] = 8;
    else {
        if (G.clip || G.mask) G.V = 2;   // (the pixel-major kernel is written for 2x2x2 blocks)
        G.bd[0] = G.bd[1] = G.bd[2] = 4 * G.V;
    }
    // clip_to_volume / per-channel mask: the ray-major splat unless XVR_DRR_GATHER_SPLAT=0 (A/B: the voxel-driven pixel-major gather)
    const bool psplat = !siddon && use_splat && !splat && sp->n_points < 65536;   // (its list packs a step and a count into 16 bits each)
    if (splat || psplat) {   // the poses' maxima behind q's used part ([B][2 n] float4, H (W + 1) = n + H used): a line per pose if it fits
        G.cmax = reinterpret_cast<unsigned*>(G.q + (size_t)B * G.qn);
        const int room = 4 * (n - n / gw);
        G.cmax_stride = room < CMAX_STRIDE ? room : CMAX_STRIDE;
    }
    if (siddon && !G.cells && !G.mask) {   // per-pose "a ray is cut at alpha = 0 / 1" words behind q's used half ([B][2 n] float4, [B][n] used)
        G.cmax = reinterpret_cast<unsigned*>(G.q + (size_t)B * n);
        G.cmax_stride = 4 * n < CMAX_STRIDE ? 4 * n : CMAX_STRIDE;
    }
    if (psplat) G.bd[0] = G.bd[1] = G.bd[2] = 16;
    if (splat || sid_splat) G.bd[0] = G.bd[1] = G.bd[2] = 16;
    // gather_splat = 1: the splat, EXCEPT where a pose puts more than SPLAT_MAX_SAMPLES_PER_VOXEL samples on a voxel (decided on the
    // device by k_gather_prep: the fp32 table gather then takes the whole launch); = 3: the splat whatever the sampling density
    const int bd8[3] = {8, 8, 8}, bd16[3] = {16, 16, 16};
    const bool auto_fp32 = splat && splat_mode == 1 && G.V == 2 && (long long)n / gw * (gw + 1) <= (long long)TAB_MAX_RAYS &&
                           n_bricks(D0, D1, D2, bd16) + n_bricks(D0, D1, D2, bd8) <= n_bricks_max(D0, D1, D2);   // (both culls fit the workspace)
    G.only_if_fine = auto_fp32 ? 0 : -1;
    G.spv_limit = auto_fp32 ? SPLAT_MAX_SAMPLES_PER_VOXEL : 0.f;
    if (sid_splat && G.cmax_stride < 2) return fail(XVR_DRR_E_UNSUPPORTED, "siddon splat: detector too small for its per-pose words");
    G.cull = reinterpret_cast<unsigned*>(ws + ws_cull_off(B, n));
    G.words = (B + 31) / 32;
    G.gvol = grad_volume;
    *flag_out = G.flag;
    // option gather_slab = index | count << 8: the voxel gradient in `count` x slabs of whole brick planes, one backward call per slab
    // (index 0 first; same arguments and workspace), so that the caller can hand slab i to a collective while slab i + 1 is computed.
    // The brick splats take their slab's bricks; every other path does the whole volume in call 0 and nothing afterwards.
    const int slab_opt = xvr_detail::option(xvr_detail::OPT_GATHER_SLAB), slab_K = slab_opt >> 8, slab_i = slab_opt & 0xff;
    if (slab_K > 1 && slab_i >= slab_K) return fail(XVR_DRR_E_ARG, "gather_slab: index >= count");
    const bool later_slab = slab_K > 1 && slab_i > 0;
    const bool brick_splat = sid_splat || (!siddon && (psplat || (!G.clip && !G.mask && splat)));
    const int nb0 = (D0 + 15) / 16;
    G.bx0 = 0;
    G.bxn = nb0;
    if (slab_K > 1 && brick_splat) {
        G.bx0 = (int)((long long)slab_i * nb0 / slab_K);
        G.bxn = (int)((long long)(slab_i + 1) * nb0 / slab_K) - G.bx0;
    }
    if (later_slab && !brick_splat) return XVR_DRR_OK;
    const long long bricks = n_bricks(D0, D1, D2, G.bd);
    if (bricks >= (1LL << 31)) return fail(XVR_DRR_E_UNSUPPORTED, "grid too large");
    hipError_t e = hipSuccess;
    if (later_slab) e = hipMemsetAsync(G.flag + 1, 0, sizeof(unsigned), (hipStream_t)stream);   // (the brick queue; lattice flag, sticky words, poses and cull stand from call 0)
    else {
        e = hipMemsetAsync(G.flag, 0, 16, (hipStream_t)stream);
        if (e == hipSuccess && G.cmax) e = hipMemsetAsync(G.cmax, 0, (size_t)B * G.cmax_stride * sizeof(unsigned), (hipStream_t)stream);
    }
    if (e == hipSuccess && sid_splat) e = hipMemsetAsync(G.flag + 16, 0, 8 * sizeof(unsigned), (hipStream_t)stream);   // (k_siddon_splat's per-XCD brick queues)
    if (e != hipSuccess) return fail(XVR_DRR_E_LAUNCH, hipGetErrorString(e));
    if (!later_slab) {
        hipLaunchKernelGGL(k_gather_prep, dim3((unsigned)((n + WG - 1) / WG), (unsigned)B), dim3(WG), 0,
                           (hipStream_t)stream, G);
        hipLaunchKernelGGL(k_gather_cull, dim3((unsigned)((bricks + WG / 32 - 1) / (WG / 32))), dim3(WG), 0,
                           (hipStream_t)stream, G, (int)bricks);
    }
    if (sid_splat) {
        const bool nx = siddon_splat == 2;   // (2: a non-exact index map; 1: the exact one, A/B)
        const void* kern = nx ? (const void*)k_siddon_splat<true> : (const void*)k_siddon_splat<false>;
        int per_cu = 0, dev = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, 0) != hipSuccess || per_cu < 1) per_cu = 2;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        const long long resident = (long long)per_cu * cus;
        const dim3 grid((unsigned)(bricks < resident ? bricks : resident));
        if (nx) hipLaunchKernelGGL(k_siddon_splat<true>, grid, dim3(256), 0, (hipStream_t)stream, G);
        else hipLaunchKernelGGL(k_siddon_splat<false>, grid, dim3(256), 0, (hipStream_t)stream, G);
#ifdef XVR_SS_STATS
        {
            hipStreamSynchronize((hipStream_t)stream);
            unsigned long long st[8];
            hipMemcpyFromSymbol(st, HIP_SYMBOL(g_ss_stats), sizeof(st));
            fprintf(stderr, "ss_stats trips %llu live_lane_trips %llu walks %llu walk_lanes %llu chunks %llu cands %llu live %llu pose_visits %llu\n", st[0], st[1], st[2], st[3], st[4], st[5], st[6], st[7]);
            unsigned long long z[8] = {0};
            hipMemcpyToSymbol(HIP_SYMBOL(g_ss_stats), z, sizeof(z));
        }
#endif
    }
    else if (siddon && G.cells) {
        hipLaunchKernelGGL(k_siddon_gather_cells, dim3((unsigned)bricks), dim3(WG), 0, (hipStream_t)stream, G);
        const long long nvox = (long long)D0 * D1 * D2;
        hipLaunchKernelGGL(k_siddon_cells_to_voxels, dim3((unsigned)((nvox + WG - 1) / WG)), dim3(WG), 0, (hipStream_t)stream, G);
    }
    else if (siddon && G.mask) hipLaunchKernelGGL(k_siddon_gather_mask, dim3((unsigned)bricks), dim3(WG), 0, (hipStream_t)stream, G);
    else if (siddon && xvr_detail::option(xvr_detail::OPT_SIDDON_GATHER_FAST)) {
        // four bricks in a row per workgroup, whichever axis the kernel picks for the rows: enough workgroups for the worst case
        const long long nb0 = (D0 + 7) / 8, nb1 = (D1 + 7) / 8, nb2 = (D2 + 7) / 8;
        const long long g0 = ((nb0 + 3) / 4) * nb1 * nb2, g1 = nb0 * ((nb1 + 3) / 4) * nb2, g2 = nb0 * nb1 * ((nb2 + 3) / 4);
        const long long groups = g0 > g1 ? (g0 > g2 ? g0 : g2) : (g1 > g2 ? g1 : g2);
        hipLaunchKernelGGL(k_siddon_gather_vol2<true>, dim3((unsigned)groups), dim3(256), 0, (hipStream_t)stream, G);
    }
    else if (siddon) hipLaunchKernelGGL(k_siddon_gather_vol2<false>, dim3((unsigned)bricks), dim3(64), 0, (hipStream_t)stream, G);
    else if (psplat) {
        const void* kern = G.clip ? (G.mask ? (const void*)k_trilinear_splat_px<true, true> : (const void*)k_trilinear_splat_px<true, false>)
                                  : (G.mask ? (const void*)k_trilinear_splat_px<false, true> : (const void*)k_trilinear_splat_px<false, false>);
        int per_cu = 0, dev = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, 0) != hipSuccess || per_cu < 1) per_cu = 2;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        const long long resident = (long long)per_cu * cus;
        const dim3 grid((unsigned)(bricks < resident ? bricks : resident));
        if (G.clip && G.mask) hipLaunchKernelGGL((k_trilinear_splat_px<true, true>), grid, dim3(256), 0, (hipStream_t)stream, G);
        else if (G.clip) hipLaunchKernelGGL((k_trilinear_splat_px<true, false>), grid, dim3(256), 0, (hipStream_t)stream, G);
        else if (G.mask) hipLaunchKernelGGL((k_trilinear_splat_px<false, true>), grid, dim3(256), 0, (hipStream_t)stream, G);
        else hipLaunchKernelGGL((k_trilinear_splat_px<false, false>), grid, dim3(256), 0, (hipStream_t)stream, G);
    }
    else if (G.clip && G.mask) hipLaunchKernelGGL((k_trilinear_gather_px<true, true>), dim3((unsigned)bricks), dim3(64), 0, (hipStream_t)stream, G);
    else if (G.clip) hipLaunchKernelGGL((k_trilinear_gather_px<true, false>), dim3((unsigned)bricks), dim3(64), 0, (hipStream_t)stream, G);
    else if (G.mask) hipLaunchKernelGGL((k_trilinear_gather_px<false, true>), dim3((unsigned)bricks), dim3(64), 0, (hipStream_t)stream, G);
    else if (splat) {
        if (auto_fp32 && !later_slab) {
            // the fine-sampling regime's pair behind the splat: cull on the table gather's 8^3 bricks (into the words behind the
            // splat's) and the gather itself, both of which return at once unless k_gather_prep raised word 3 of the flag line
            GatherArgs T = G;
            T.only_if_fine = 1;
            T.cmax = nullptr;
            T.bd[0] = T.bd[1] = T.bd[2] = 4 * T.V;
            T.cull = G.cull + (size_t)bricks * G.words;
            const long long tb = n_bricks(D0, D1, D2, T.bd);
            hipLaunchKernelGGL(k_gather_cull, dim3((unsigned)((tb + WG / 32 - 1) / (WG / 32))), dim3(WG), 0, (hipStream_t)stream, T, (int)tb);
            hipLaunchKernelGGL(k_trilinear_gather_tab, dim3((unsigned)tb), dim3(64), 0, (hipStream_t)stream, T);
        }
        // persistent workgroups: as many as run at once (the occupancy the runtime reports x the CUs), never more than bricks
        static const int resident = [] {
            int per_cu = 0, dev = 0, cus = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_trilinear_splat_b16, 256, 0) != hipSuccess || per_cu < 1) per_cu = 2;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
            return per_cu * cus;
        }();
        hipLaunchKernelGGL(k_trilinear_splat_b16, dim3((unsigned)(bricks < resident ? bricks : resident)), dim3(256), 0, (hipStream_t)stream, G);
    }
    else hipLaunchKernelGGL(k_trilinear_gather_tab, dim3((unsigned)bricks), dim3(64), 0, (hipStream_t)stream, G);   // (qn <= TAB_MAX_RAYS: gather_usable)
    e = hipGetLastError();
    if (e != hipSuccess) return fail(XVR_DRR_E_LAUNCH, hipGetErrorString(e));
    return XVR_DRR_OK;
}

extern "C" {

#ifdef XVR_GATHER_STATS
int xvr_drr_debug_gather_stats(unsigned long long* out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_gather_stats), 64) != hipSuccess) return XVR_DRR_E_LAUNCH;
    if (reset) {
        const unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_gather_stats), zero, 64) != hipSuccess) return XVR_DRR_E_LAUNCH;
    }
    return XVR_DRR_OK;
}
#endif

#ifdef XVR_S16_TRACE
int xvr_drr_debug_s16_occupancy() {   // workgroups of the 16^3 splat the runtime says fit one CU
    int n = -1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_trilinear_splat_b16, 256, 0) != hipSuccess) return -1;
    return n;
}
int xvr_drr_debug_s16_trace(unsigned long long* out, int n_groups) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_s16_trace), (size_t)n_groups * 96) != hipSuccess) return XVR_DRR_E_LAUNCH;
    return XVR_DRR_OK;
}
#endif

size_t xvr_drr_backward_workspace_bytes(int B, int n, int D0, int D1, int D2) {
    if (B <= 0 || n <= 0 || D0 <= 0 || D1 <= 0 || D2 <= 0) return 0;
    return ws_bytes(B, n, D0, D1, D2);
}

size_t xvr_drr_siddon_backward_workspace_bytes(int B, int n, int D0, int D1, int D2, const xvr_drr_spec* sp) {
    if (B <= 0 || n <= 0 || D0 <= 0 || D1 <= 0 || D2 <= 0 || !sp) return 0;
    int olo[3];
    const size_t base = ws_bytes(B, n, D0, D1, D2);
    if (siddon_exact_geometry(sp) || !siddon_cell_offsets(sp, D0, D1, D2, olo)) return base;
    if (xvr_detail::option(xvr_detail::OPT_SIDDON_SPLAT) >= 1 && siddon_map_in_bounds(sp, D0, D1, D2) && siddon_splat_detector_ok(sp, n)) return base;   // (the brick splat needs no per-cell scratch)
    return align256(base) + siddon_cells_bytes(D0, D1, D2);
}

}  // extern "C"
