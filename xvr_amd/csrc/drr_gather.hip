// MI355X (gfx950) differentiable-DRR kernels: the voxel gradient as an atomic-free, voxel-driven gather
// (trilinear and Siddon), its per-pose preparation and per-brick cull.  DESIGN.md section 4.1.
#include "drr_common.hiph"

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
        float c = (G.mask ? 1.f : G.gout[(size_t)b * G.n + r]) * G.raylen[(size_t)b * G.n + r] * G.sp.inv_denom;
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
                    const float gap = span * lx * (G.sp.far_ - G.sp.near_) / N1;
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
            if (!(lo > 0.f)) lo = 0.f;
            if (!(hi < 1.f)) hi = 1.f;
            G.q[(size_t)b * G.n + r] = make_float4(1.f / ddx, 1.f / ddy, 1.f / ddz,
                                                   G.gout[(size_t)b * G.n + r] * G.raylen[(size_t)b * G.n + r]);
            G.q2[(size_t)b * G.n + r] = make_float2(lo, hi);
            if (G.cells) G.q[(size_t)G.B * G.n + (size_t)b * G.n + r] = make_float4(ddx, ddy, ddz, 0.f);   // d itself, for the midpoints
        }
    }
    if (G.cmax) {   // max |c| of the pose: the fixed-point scale of the brick-local splat
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cabs = fmaxf(cabs, __shfl_xor(cabs, o));
        // (a thousand wavefronts per pose meet in its word: the poses' words sit in different cache lines -- packed, the 116
        //  of the benchmark shared four lines and 1.2e5 L2 atomics queued up behind each other for 0.6 ms -- and only a
        //  wavefront that would RAISE the maximum writes.  A stale read only costs a redundant atomic.)
        if ((threadIdx.x & 63) == 0 && cabs > 0.f &&
            __float_as_uint(cabs) > __hip_atomic_load(G.cmax + (size_t)b * G.cmax_stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(G.cmax + (size_t)b * G.cmax_stride, __float_as_uint(cabs));
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
            if (amax >= G.cull_lo && amin <= G.cull_hi) {
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
__device__ __forceinline__ float hat01(float d) { return __builtin_amdgcn_fmed3f(1.f - fabsf(d), 0.f, 1.f); }

// linspace_at() with both halves evaluated and one select (same values, no exec-masked branches in the inner loops)
__device__ __forceinline__ float linspace_sel(int k, int N, float near_, float far_, float step) {
    const float lo = fmaf(step, (float)k, near_), hi = far_ - step * (float)(N - 1 - k);
    return (k < N / 2 || N == 1) ? lo : hi;
}

// (register budget set for 7 wavefronts per SIMD: 72 VGPRs, no spills, 14.4 ms at C2 against 14.6 at the 6 the
//  compiler chose; 8 spills, 4 takes 17.7 ms)
template <int V>
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
#ifdef XVR_GATHER_STATS
    unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
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
            XVR_STAT(0, inb ? 1 : 0);
            XVR_STAT_WAVE(7);
            const float grw = P.gr[0] * w0 + P.gr[1] * w1 + P.gr[2] * w2;
            const float4* __restrict__ q = G.q + (size_t)p * G.qn;
            const float Bx = fmaf(a0, s0, bv0), By = fmaf(a1, s1, bv1), Bz = fmaf(a2, s2, bv2);
            for (int k = klo; k <= khi; ++k) {
                const float al = linspace_at(k, N, near_, far_, step);
                XVR_STAT(1, 1);
                if (al > 1e-12f) {
                    const float inv = 1.f / al;
                    const float ic = fmaf(grw, inv, P.gr0);
                    // exact extent, along the detector's row axis, of the block's slice at this alpha (PoseLattice.rl/rc):
                    // the whole-cube window has 45 % empty rows
                    const float dlt = al - av;
                    const float up = fminf(fminf(fmaf(P.rl[0], dlt, HS * P.rc[0]), fmaf(P.rl[1], dlt, HS * P.rc[1])),
                                           fminf(fmaf(P.rl[2], dlt, HS * P.rc[2]), HS * P.hwr));
                    const float dn = fminf(fminf(fmaf(-P.rl[0], dlt, HS * P.rc[0]), fmaf(-P.rl[1], dlt, HS * P.rc[1])),
                                           fminf(fmaf(-P.rl[2], dlt, HS * P.rc[2]), HS * P.hwr));
                    const int ilo = (int)ceilf(fmaxf(ic - fmaf(fmaxf(dn, 0.f), inv, GATHER_WIN_MARGIN), 0.f));
                    const int ihi = (int)floorf(fminf(ic + fmaf(fmaxf(up, 0.f), inv, GATHER_WIN_MARGIN), (float)(G.H - 1)));
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
                    const float Ax = al, Ay = al, Az = al;   // (q holds a * d)
                    for (int i = ilo; i <= ihi; ++i) {
                        const float fi = (float)i;
                        const float qx = fmaf(fi, urx, q0x), qy = fmaf(fi, ury, q0y), qz = fmaf(fi, urz, q0z);
                        // exact j-interval on this row where |q + j Uc| < HS on all three axes
                        const float mx = -qx * rx, my = -qy * ry, mz = -qz * rz;
                        const float lo = fmaxf(fmaxf(mx - ax_, my - ay_), mz - az_);
                        const float hiJ = fminf(fminf(mx + ax_, my + ay_), mz + az_);
                        const int jlo = (int)ceilf(fmaxf(lo - jmargin, 0.f));
                        const int jhi = (int)floorf(fminf(hiJ + jmargin, (float)(G.W - 1)));
                        const float4* __restrict__ row = q + (size_t)i * G.qs;
                        XVR_STAT(2, 1);
                        XVR_STAT(3, jlo > jhi ? 1 : 0);
                        // two candidates per trip: both 16-byte loads are issued before either is used
                        for (int j = jlo; j <= jhi; j += 2) {
                            const bool two = j < jhi;
                            XVR_STAT(4, two ? 2 : 1);
                            XVR_STAT_WAVE(6);
                            float4 ta = row[j], tb = row[two ? j + 1 : j];
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
                    const int cnt = hit ? G.qn : 0;   // (the rows' closing elements carry weight 0)
                    for (int r = 0; r < cnt; ++r) {
                        const float4 t = q[r];
#pragma unroll
                        for (int e = 0; e < V * V * V; ++e) {
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
#ifdef XVR_GATHER_STATS
    for (int i = 0; i < 8; ++i)
        if (st[i]) atomicAdd(&g_gather_stats[i], st[i]);
#endif
#pragma unroll
    for (int e = 0; e < V * V * V; ++e) {
        const int x = vx + (V == 2 ? (e >> 2 & 1) : 0), y = vy + (V == 2 ? (e >> 1 & 1) : 0), z = vz + (V == 2 ? (e & 1) : 0);
        if (x < G.D0 && y < G.D1 && z < G.D2 && acc[e] != 0.f) G.gvol[((size_t)x * G.D1 + y) * G.D2 + z] += acc[e];
    }
}

// ---------------------------------------------------------------------------------------------
// The same gather with the loop nest FLATTENED per lane (round 2; DESIGN.md section 4.1, tools/sim_gather_divergence.py).
//
// In the nested kernel above the 64 lanes of a wavefront walk step -> row -> pixel loops whose trip counts differ per
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
    const float near_ = G.sp.near_, far_ = G.sp.far_;
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
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int x = vx + (e >> 2 & 1), y = vy + (e >> 1 & 1), z = vz + (e & 1);
        if (x < G.D0 && y < G.D1 && z < G.D2 && acc[e] != 0.f) G.gvol[((size_t)x * G.D1 + y) * G.D2 + z] += acc[e];
    }
}

// ---------------------------------------------------------------------------------------------
// Brick-local splat (round 2): the same voxel gradient, SAMPLE-driven, with the sums of a brick of voxels held in LDS as
// 32-bit fixed point.
//
// Why: the counters say the table gather above is limited twice over -- the vector ALUs are 64 % busy on work of which
// 70 % is multiplications by a zero weight (a candidate of a 2x2x2 block has 2.4 non-zero weights of 8, and 41 of 64
// lanes are live), and the L1 is 85-89 % busy because every lane's 16-byte candidate load is its own cache access (39
// accesses per load instruction; tools/profile_mempipe.sh).  A sample-driven pass has neither problem: every sample is
// evaluated once per brick whose support holds it, with all eight weights useful, and neighbouring lanes read
// neighbouring pixels.  What it needs is an accumulate into shared memory, and tools/microbench/lds_atomics.hip measured
// the one that works: ds_add_f32 retires 0.33 lanes per clock and CU (it is serialised), ds_add_u32 11.5 (random words;
// 4.4 clocks per wavefront instruction when no two lanes share a word), ds_add_u64 half that.  So the sums are integers:
//   * per (brick, pose) visit the workgroup enumerates, with the lattice arithmetic of the table kernel applied to the
//     brick's whole support box, the runs (row, first pixel, count) of samples inside the box -- one row per lane, packed
//     into an LDS list;
//   * the scale is a power of two that keeps (most samples that can touch one voxel) x (the pose's max |c|, from
//     k_gather_prep) below 2^30, so no sum can overflow; a sample's eight products w * c * scale are rounded to nearest
//     (v_cvt_rpi_i32_f32) and added with ds_add_u32 to an array of cells, the brick and one cell around it (the outer
//     cells are never read);
//   * after the pose's samples the threads read their voxels, convert, add them to fp32 registers and clear the cells.
// Integer sums are exact and order-free: the result is deterministic, and differs from the table kernel's by the
// rounding of the products, ~0.3 LSB per add with LSB = bound / 2^30.
// A pose whose upstream gradient holds a non-finite value poisons the voxels of the bricks it visits (NaN).
// ---------------------------------------------------------------------------------------------
#ifndef XVR_SP_ABLATE_ADDS
#define XVR_SP_ABLATE_ADDS 0
#endif
// a value every lane holds alike, moved to a scalar register (gfx950 has no scalar float ALU: uniform float arithmetic is
// done by the vector ALU and would otherwise sit in a vector register for as long as it lives)
__device__ __forceinline__ float uni(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

__device__ __forceinline__ int cvt_nearest(float v) {
    int r;
    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(v));   // floor(v + 0.5)
    return r;
}

// ---------------------------------------------------------------------------------------------
// The same splat on 16^3-voxel bricks, four wavefronts per brick: the per-visit work that does not depend on the number of
// samples (pose constants, one window set-up per step, the flush) is paid once for 8 x the voxels, and the support box
// holds 1.20 x the brick's samples instead of 1.42 x.
//   enumeration  wavefront w takes steps klo + w, klo + w + 4, ...; lane = detector row; the runs are packed into ONE list by
//                an LDS counter (one ds_add_rtn per wavefront and 64 rows);
//   splat        64 groups of four lanes; group g takes runs g, g + 64, ..., lane c of the group the c-th quarter of a run;
//   flush        thread t owns the 16 voxels (t / 16, t % 16, 0..15).
// A visit whose list would overflow, that has more steps than the alpha table holds, or that meets a step at alpha = 0 is
// redone in "safe mode": one step at a time over all four wavefronts, drained after every step.
// ---------------------------------------------------------------------------------------------
#if defined(XVR_S16_ABLATE_LOADS)   // diagnostic build only: no memory traffic for the samples -- WRONG sums
#define XVR_S16_LOAD(p) make_float4(0.25f, 0.5f, 0.125f, (float)((size_t)(p) & 255u))
#else
#define XVR_S16_LOAD(p) (*(p))
#endif
#ifndef XVR_S16_DEPTH
#define XVR_S16_DEPTH 2
#endif
#ifndef XVR_S16_SHARES   // 1: every thread the same number of samples (prefix over the list); 0: quarters of runs, round robin
#define XVR_S16_SHARES 1
#endif
#ifndef XVR_S16_GROUP_STRIDE
#define XVR_S16_GROUP_STRIDE 19
#endif
#ifndef XVR_S16_TAB
#define XVR_S16_TAB 896
#endif
constexpr int S16_DIM = 18, S16_CELLS = S16_DIM * S16_DIM * S16_DIM, S16_TAB = XVR_S16_TAB, S16_STEPS = 128;

#ifndef XVR_S16_CENTRE_OUT
#define XVR_S16_CENTRE_OUT 1
#endif
#ifndef XVR_S16_WAVES   // wavefronts per SIMD the register budget is set for (tools/tune_splat.py)
#define XVR_S16_WAVES 4
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(XVR_S16_WAVES, XVR_S16_WAVES))) void k_trilinear_splat_b16(GatherArgs G) {
    if (*G.flag > __float_as_uint(GATHER_DEV_TOL)) return;  // not a lattice: the scatter kernel runs instead
    __shared__ __attribute__((aligned(16))) int cell[S16_CELLS];
    __shared__ uint2 tab[S16_TAB];          // x = first element of the run in q;  y = count << 8 | step - kbase
    // list length, samples in the list, "redo in safe mode": two sets, alternating between visits, so that the set of visit
    // v is cleared (by thread 0, after v's second barrier) while nobody reads or writes it -- two barriers per visit suffice
    __shared__ __attribute__((aligned(16))) int s_ctl[2][4];   // [0] runs in the list, [1] samples in the list, [2] redo
#if XVR_S16_SHARES
    __shared__ unsigned run_start[S16_TAB]; // samples in the runs before this one (list order)
#endif
    // the pose constants of this visit and of the next one (PoseLattice's floats + max|c|): the next pose's are fetched
    // while this pose's samples are splatted -- a visit's first use of them was 7000 clocks of exposed memory latency
    constexpr int PW = (int)(sizeof(PoseLattice) / sizeof(float));
    __shared__ float s_P[2][PW + 4];
    __shared__ int s_next[2];   // the brick this workgroup takes next (two slots: written for turn t + 1 while t may still be read)
    int par = 0;
    constexpr float HS = 8.5f, CO = 7.5f;   // samples that touch voxels 0..15 sit in [-1, 16): centre 7.5, half 8.5
    constexpr float HSR = HS / 1.5f;        // PoseLattice.hsr is made for the 2x2x2 block's half-size 1.5
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lx = tid >> 4, ly = tid & 15;                     // this thread's column of 16 voxels along z
    const int N = G.sp.n_points;
    const float near_ = G.sp.near_, far_ = G.sp.far_;
    const float step = N > 1 ? (far_ - near_) / (float)(N - 1) : 0.f;
    const float inv_step = step > 0.f ? 1.f / step : 0.f;
    const float a0 = G.sp.a[0], a1 = G.sp.a[1], a2 = G.sp.a[2];
    const float b0 = G.sp.b[0], b1 = G.sp.b[1], b2 = G.sp.b[2];
    const float jmargin = GATHER_DEV_TOL + 0.01f;
    const float Hm1 = (float)(G.H - 1), Wm1 = (float)(G.W - 1);
    const int n0 = (G.D0 + 15) / 16, n1 = (G.D1 + 15) / 16, n2 = (G.D2 + 15) / 16;
    for (int i = tid; i < S16_CELLS / 4; i += 256) reinterpret_cast<int4*>(cell)[i] = make_int4(0, 0, 0, 0);
    if (tid < 8) s_ctl[tid >> 2][tid & 3] = 0;
#ifdef XVR_S16_TRACE
#define XVR_TICK(i) do { const unsigned long long now_ = clock64(); tk[i] += now_ - tl; tl = now_; } while (0)
#else
#define XVR_TICK(i) ((void)0)
#endif

    // Persistent workgroups: the launch holds as many as the chip runs at once and every one takes bricks off a queue (a
    // counter next to the lattice flag) until it is empty.  One workgroup per brick left 42 % of the workgroup slots idle
    // for the whole launch (tools/splat_trace.py) -- the dispatcher does not keep up with 31 KiB / 4-wavefront workgroups
    // of very unequal length.
    for (int turn = 0;; turn ^= 1) {
    if (tid == 0) s_next[turn] = (int)atomicAdd(G.flag + 1, 1u);
    __syncthreads();   // (also: the cells are clear, the previous brick's flush is done)
    const int blk = __builtin_amdgcn_readfirstlane(s_next[turn]);   // (scalar: everything per brick and per pose below is uniform)
    if (blk >= n0 * n1 * n2) break;
    int bx = blk / (n1 * n2), by = (blk / n2) % n1, bz = blk % n2;
#if XVR_S16_CENTRE_OUT
    // bricks from the middle of the volume outwards: the ones most poses cross first, the empty corners last
    bx = (bx & 1) ? (n0 >> 1) - ((bx + 1) >> 1) : (n0 >> 1) + (bx >> 1);
    by = (by & 1) ? (n1 >> 1) - ((by + 1) >> 1) : (n1 >> 1) + (by >> 1);
    bz = (bz & 1) ? (n2 >> 1) - ((bz + 1) >> 1) : (n2 >> 1) + (bz >> 1);
#endif
    const int brick_id = (bx * n1 + by) * n2 + bz;
#ifdef XVR_S16_TRACE
    const unsigned long long t_start = wall_clock64();
    unsigned long long n_visits = 0, n_samples = 0, tk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl = clock64();
#endif
    const int ox = bx * 16, oy = by * 16, oz = bz * 16;         // first voxel of the brick
    const float fv[3] = {(float)ox, (float)oy, (float)oz};
    float xv[3];  // centre of the support box in x coordinates
#pragma unroll
    for (int i = 0; i < 3; ++i) xv[i] = uni((fv[i] + CO - G.sp.b[i]) / G.sp.a[i]);
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    // the brick's poses, one ahead: pc is this visit's pose, pn the next one's
    int wd = 0;
    unsigned bits = G.cull[(size_t)brick_id * G.words];  // uniform: scalar load
    auto next_pose = [&]() -> int {
        while (!bits) {
            if (++wd >= G.words) return -1;
            bits = G.cull[(size_t)brick_id * G.words + wd];
        }
        const int p = wd * 32 + __builtin_ctz(bits);
        bits &= bits - 1;
        return p;
    };
    auto fetch = [&](const int p) -> float {   // thread t <= PW: word t of the pose's constants
        return tid < PW ? reinterpret_cast<const float*>(G.poses + p)[tid] : __uint_as_float(G.cmax[(size_t)p * G.cmax_stride]);
    };
    int pc = next_pose(), visit = 0;
    if (pc >= 0 && tid <= PW) s_P[0][tid] = fetch(pc);
    __syncthreads();
    while (pc >= 0) {
        {
            const int p = pc, pn = next_pose();
            float pre = 0.f;
            if (pn >= 0 && tid <= PW) pre = fetch(pn);    // in flight during the visit; parked in s_P before the visit's last barrier
            const int cur = visit & 1;
            ++visit;
            auto park = [&]() { if (pn >= 0 && tid <= PW) s_P[cur ^ 1][tid] = pre; };
            pc = pn;
            const float* Pf = s_P[cur];
            const PoseLattice& P = *reinterpret_cast<const PoseLattice*>(Pf);
            const float cmax = Pf[PW];
            if (cmax == 0.f) { park(); __syncthreads(); continue; }   // the pose's upstream gradient is all zeros
            const float s0 = P.s[0], s1 = P.s[1], s2 = P.s[2];
            const float w0 = xv[0] - s0, w1 = xv[1] - s1, w2 = xv[2] - s2;
            const float av = uni(P.nh[0] * w0 + P.nh[1] * w1 + P.nh[2] * w2);
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
            if (!(av == av)) khi = -1;
            if (khi < klo) { park(); __syncthreads(); continue; }
            const float grw = uni(P.gr[0] * w0 + P.gr[1] * w1 + P.gr[2] * w2);
            const float4* __restrict__ q = G.q + (size_t)p * G.qn;
            // a s + b - (first cell of the array): array index of a sample = floor of this + alpha (a d); the brick's voxel v
            // is cell v + 1 (a sample in [-1, 16) has its lower tap in cells 0..16)
            const float Bx = uni(fmaf(a0, s0, b0 - (fv[0] - 1.f))), By = uni(fmaf(a1, s1, b1 - (fv[1] - 1.f))), Bz = uni(fmaf(a2, s2, b2 - (fv[2] - 1.f)));
            // ... and relative to the centre of the support box, for the windows
            const float Cx = uni(fmaf(a0, s0, b0 - fv[0]) - CO), Cy = uni(fmaf(a1, s1, b1 - fv[1]) - CO), Cz = uni(fmaf(a2, s2, b2 - fv[2]) - CO);
            const float rcx = HS * P.rc[0], rcy = HS * P.rc[1], rcz = HS * P.rc[2], rcw = HS * P.hwr;
            // A voxel's sum of weights over the pose's samples is at most `tcell`: the samples are a lattice (pixel spacing dc
            // along a row, rows dr apart, planes dn apart, all in voxels at the box's smallest alpha) and the weight w =
            // hat x hat x hat is log-concave, so along a row sum <= integral / dc + max (<= 1), the row integrals are
            // unimodal in the row index (sum <= plane integral / dr + max line integral <= sqrt 3), the plane integrals in the
            // plane index (sum <= 1 / dn + max plane integral <= sqrt 3), and at most mr rows per plane and mp planes touch the
            // voxel's 2-cube:   sum w <= (1 / (dc dr)) (1 / dn + sqrt 3) + mp (sqrt 3 / dc + mr).
            // (The plain count of lattice points in the 2-cube is 4-11 x larger; the fixed-point LSB scales with this bound.)
            float tcell = INFINITY;
            {
                const float amin = av - da;
                if (amin > 1e-6f && step > 0.f) {
                    const float r3 = 1.7320508f;
                    const float idc = __builtin_amdgcn_rcpf(amin * P.ecl), idr = __builtin_amdgcn_rcpf(amin * P.rperp), idn = P.gn * inv_step;
                    const float mp = 2.f * r3 * idn + 1.f, mr = 2.f * r3 * idr + 1.f;
                    tcell = 1.02f * (idc * idr * (idn + r3) + mp * (r3 * idc + mr));
                    if (!(tcell == tcell)) tcell = INFINITY;
                }
                tcell = uni(tcell);
            }

            int kbase = klo;    // step of slot 0 of the list
            // ---- the list's samples into the cells, the cells into the threads' registers.  Called by ALL threads, after the
            // barrier that completes the list; ends with the barrier after which the cells may be added to again.
            auto drain = [&]() {
                int* ctl = s_ctl[par];
                const int cnt = ctl[0];
#ifdef XVR_S16_TRACE
                n_visits += 1; n_samples += (unsigned long long)cnt;
#endif
                if (cnt > 0 && cmax < INFINITY) {
                    // scale = the power of two that puts (bound on a voxel's sum) into [2^29, 2^30): exact to apply and undo.
                    // Where the lattice bound does not exist (the box reaches the source plane) the samples in the list stand
                    // in, counted generously: runs x detector width.  (An exact count would be a same-word add by every lane of
                    // every append: 128 clocks of the LDS pipe each.)
                    const float bound = fminf((float)cnt * (float)G.W, tcell) * cmax;
                    const int ex = (int)(__float_as_uint(bound) >> 23) - 126;   // bound < 2^ex
                    const float cs = uni(__uint_as_float((unsigned)(127 + 30 - ex) << 23)), ics = uni(__uint_as_float((unsigned)(127 - 30 + ex) << 23));
#if XVR_S16_SHARES
                    // every thread the same number of samples (+-1): thread t takes samples [t T / 256, (t + 1) T / 256) of the
                    // list, found by bisection on the runs' offsets; neighbouring threads are a share (~13 pixels, 7 voxels)
                    // apart.  (Static quarters of runs left the slowest thread 18-20 trips for a mean of 13.)
                    const unsigned T = (unsigned)ctl[1];
                    const unsigned lo = (unsigned)(((unsigned long long)tid * T) >> 8), hi = (unsigned)(((unsigned long long)(tid + 1) * T) >> 8);
                    int e = 0;
#pragma unroll
                    for (int stp = 512; stp > 0; stp >>= 1) {
                        const int c = e + stp;
                        if (c < cnt && run_start[c] <= lo) e = c;
                    }
                    int left = (int)(hi - lo), rem = 0;   // samples of the share / of the current run still to come (incl. this one)
                    const float4* __restrict__ ptr = q;
                    float al = 0.f;
                    int first = (int)(lo - run_start[e]);       // (only the first run of a share is entered in its middle)
                    --e;
                    // on to the thread's next sample: 1 = there is one at (ptr, al), -1 = the thread is done
                    auto advance = [&]() -> int {
                        if (left <= 0) return -1;
                        --left;
                        --rem;
                        ++ptr;
                        if (rem > 0) return 1;
                        ++e;
                        const uint2 en = tab[e < cnt ? e : cnt - 1];
                        rem = (int)(en.y >> 8) - first;
                        ptr = q + en.x + first;
                        first = 0;
                        al = linspace_sel(kbase + (int)(en.y & 255u), N, near_, far_, step);   // (recomputed: no second LDS read)
                        return 1;
                    };
#else
                    const int c4 = tid & 3;
                    // group g starts at run (19 g) mod 64: the 16 groups of a wavefront then work on runs at least two detector
                    // rows apart (neighbouring rows' samples fall into the same cells: same-word adds serialise).  Static shares
                    // leave the slowest thread 18-20 trips for a mean of 13; handing quarters out by a counter instead was
                    // measured SLOWER (10.7 against 8.9 ms: a ds_add_rtn and a dependent read on every pull).
                    int idx = (XVR_S16_GROUP_STRIDE * (tid >> 2)) & 63, rem = 0;
                    const float4* __restrict__ ptr = q;
                    float al = 0.f;
                    // on to the thread's next sample: 1 = there is one at (ptr, al), 0 = an idle trip (the run is shorter than
                    // this lane's quarter starts), -1 = the thread is done
                    auto advance = [&]() -> int {
                        --rem;
                        ++ptr;
                        if (rem > 0) return 1;
                        if (idx >= cnt) return -1;
                        const uint2 e = tab[idx];
                        idx += 64;
                        const int n = (int)(e.y >> 8), m = (n + 3) >> 2, first = c4 * m;
                        rem = n - first < m ? n - first : m;
                        ptr = q + e.x + first;
                        al = linspace_sel(kbase + (int)(e.y & 255u), N, near_, far_, step);   // (recomputed: no second LDS read)
                        return rem > 0 ? 1 : 0;
                    };
#endif
                    auto splat = [&](const float4 t, const float alc) {
                        const float px = fmaf(alc, t.x, Bx), py = fmaf(alc, t.y, By), pz = fmaf(alc, t.z, Bz);
                        const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
                        const float rx = px - fx, ry = py - fy, rz = pz - fz;
                        const int ix = (int)fx, iy = (int)fy, iz = (int)fz;
                        // a sample outside [-1, 16) on some axis (the windows are conservative by a fraction of a pixel) touches
                        // no voxel of the brick
                        const bool ok = (unsigned)ix < (unsigned)(S16_DIM - 1) && (unsigned)iy < (unsigned)(S16_DIM - 1) && (unsigned)iz < (unsigned)(S16_DIM - 1);
                        const float cz = ok ? t.w * cs : 0.f;
                        const int base = ok ? (ix * S16_DIM + iy) * S16_DIM + iz : 0;
                        const float z1 = rz * cz, z0 = cz - z1;   // (1 - rz) cz
                        const float x1 = rx, x0 = 1.f - rx, y1 = ry, y0 = 1.f - ry;
                        const float p00 = x0 * y0, p01 = x0 * y1, p10 = x1 * y0, p11 = x1 * y1;
                        int* c = cell + base;
#if XVR_SP_ABLATE_ADDS   // diagnostic build only (tools/tune_splat.py): one add instead of eight -- WRONG sums, the price of seven adds
                        __hip_atomic_fetch_add(c, cvt_nearest(p00 * z0) + cvt_nearest(p00 * z1) + cvt_nearest(p01 * z0) + cvt_nearest(p01 * z1) +
                                               cvt_nearest(p10 * z0) + cvt_nearest(p10 * z1) + cvt_nearest(p11 * z0) + cvt_nearest(p11 * z1),
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        return;
#endif
                        __hip_atomic_fetch_add(c, cvt_nearest(p00 * z0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_add(c + 1, cvt_nearest(p00 * z1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_add(c + S16_DIM, cvt_nearest(p01 * z0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_add(c + S16_DIM + 1, cvt_nearest(p01 * z1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_add(c + S16_DIM * S16_DIM, cvt_nearest(p10 * z0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_add(c + S16_DIM * S16_DIM + 1, cvt_nearest(p10 * z1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_add(c + S16_DIM * S16_DIM + S16_DIM, cvt_nearest(p11 * z0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_add(c + S16_DIM * S16_DIM + S16_DIM + 1, cvt_nearest(p11 * z1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    };
                    // two samples in flight per thread, in two register sets (deeper rings measured no faster).  Unconditional
                    // loads (ptr always points into q, at worst one element past a run) so that no exec-masked join sits
                    // between a load and its use; the only LDS read of the loop is the run's entry, consumed where it is
                    // read -- a pending LDS result anywhere else makes the compiler drain the eight adds on every trip.
                    int sa = advance(), sb;
                    float4 ta = XVR_S16_LOAD(ptr), tb;
                    float ala = al, alb;
                    while (sa >= 0) {
                        sb = advance();
                        tb = XVR_S16_LOAD(ptr);
                        alb = al;
                        if (sa > 0) splat(ta, ala);
                        if (sb < 0) break;
                        sa = advance();
                        ta = XVR_S16_LOAD(ptr);
                        ala = al;
                        if (sb > 0) splat(tb, alb);
                    }
                    XVR_TICK(3);
                    park();
                    __syncthreads();   // every sample is in the cells
                    XVR_TICK(4);
                    if (tid == 0) { ctl[0] = 0; ctl[1] = 0; ctl[2] = 0; }
                    // the brick's voxels are cells 1..16 on every axis; a thread reads its 16 and clears them (the outer cells
                    // are never read: they may hold anything, and wrap around)
                    // (measured and not adopted: deferring this flush past the next visit's pose set-up so that the two latency
                    //  chains overlap in one basic block -- no gain, 7.74 against 7.70 ms; nine 8-byte reads / clears of the column's 18 cells instead of 32 4-byte ones --
                    //  4 550 against 3 660 clocks; rows padded to 20 cells so that four 128-bit operations do -- no difference)
                    int* col = cell + ((lx + 1) * S16_DIM + (ly + 1)) * S16_DIM + 1;
#pragma unroll
                    for (int z = 0; z < 16; ++z) {
                        acc[z] = fmaf((float)col[z], ics, acc[z]);
                        col[z] = 0;
                    }
                    XVR_TICK(5);
                } else {
                    if (cnt > 0) {   // a non-finite upstream gradient poisons the brick
#pragma unroll
                        for (int z = 0; z < 16; ++z) acc[z] = NAN;
                    }
                    park();
                    __syncthreads();
                    if (tid == 0) { ctl[0] = 0; ctl[1] = 0; ctl[2] = 0; }
                }
                par ^= 1;
            };
            // one run per lane of the calling wavefront (n <= 0: none) into the shared list; false = it does not fit
            auto append = [&](const int off, int n, const int slot) -> bool {
                n = n > 0 ? (n < 0xffffff ? n : 0xffffff) : 0;
                const unsigned long long has = __ballot(n > 0);
                const int np = __popcll(has);
                if (np == 0) return true;
#if XVR_S16_SHARES
                // samples of the lanes before this one (wavefront scan by DPP: four shifts within the rows of 16 lanes, two
                // row broadcasts) and of the whole wavefront; runs and samples are reserved by ONE 64-bit add, so that the
                // list's order is the order of the sample offsets
                int x = n;
                x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);   // row_shr:1
                x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);   // row_shr:2
                x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);   // row_shr:4
                x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);   // row_shr:8
                x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
                x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
                const int wsum = __builtin_amdgcn_readlane(x, 63);
                unsigned long long got = 0ull;
                if (lane == 0)
                    got = __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(&s_ctl[par][0]),
                                                 (unsigned long long)(unsigned)np | ((unsigned long long)(unsigned)wsum << 32),
                                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const int base = __builtin_amdgcn_readfirstlane((int)(unsigned)got);
                const int sbase = __builtin_amdgcn_readfirstlane((int)(unsigned)(got >> 32));
                if (base + np > S16_TAB) return false;
                if (n > 0) {
                    const int pos = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(has >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)has, 0u));
                    tab[pos] = make_uint2((unsigned)off, ((unsigned)n << 8) | (unsigned)slot);
                    run_start[pos] = (unsigned)(sbase + x - n);
                }
                return true;
#else
                int base = 0;
                if (lane == 0) base = __hip_atomic_fetch_add(&s_ctl[par][0], np, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                base = __builtin_amdgcn_readfirstlane(base);
                if (base + np > S16_TAB) return false;
                if (n > 0) {
                    const int pos = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(has >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)has, 0u));
                    tab[pos] = make_uint2((unsigned)off, ((unsigned)n << 8) | (unsigned)slot);
                }
                return true;
#endif
            };
            // the rows i0 + lane of step alpha = al (the table kernel's window arithmetic on the brick's support box)
            struct StepC { float q0x, q0y, q0z, urx, ury, urz, rx, ry, rz, hx, hy, hz; int ilo, ihi; };
            auto step_setup = [&](const float al, StepC& S) {
                const float inv = __builtin_amdgcn_rcpf(al);
                const float ic = fmaf(grw, inv, P.gr0);
                const float dlt = al - av;
                const float up = fminf(fminf(fmaf(P.rl[0], dlt, rcx), fmaf(P.rl[1], dlt, rcy)), fminf(fmaf(P.rl[2], dlt, rcz), rcw));
                const float dn = fminf(fminf(fmaf(-P.rl[0], dlt, rcx), fmaf(-P.rl[1], dlt, rcy)), fminf(fmaf(-P.rl[2], dlt, rcz), rcw));
                S.ilo = (int)ceilf(fmaxf(ic - fmaf(fmaxf(dn, 0.f), inv, GATHER_WIN_MARGIN), 0.f));
                S.ihi = (int)floorf(fminf(ic + fmaf(fmaxf(up, 0.f), inv, GATHER_WIN_MARGIN), Hm1));
                S.q0x = fmaf(al, P.e0[0], Cx); S.q0y = fmaf(al, P.e0[1], Cy); S.q0z = fmaf(al, P.e0[2], Cz);
                S.urx = al * P.era[0]; S.ury = al * P.era[1]; S.urz = al * P.era[2];
                S.rx = inv * P.rec[0]; S.ry = inv * P.rec[1]; S.rz = inv * P.rec[2];
                S.hx = fmaf(inv, HSR * P.hsr[0], jmargin); S.hy = fmaf(inv, HSR * P.hsr[1], jmargin); S.hz = fmaf(inv, HSR * P.hsr[2], jmargin);
            };
            auto rows = [&](const StepC& S, const int i, int& off, int& n) {
                const float fi = (float)i;
                const float qx = fmaf(fi, S.urx, S.q0x), qy = fmaf(fi, S.ury, S.q0y), qz = fmaf(fi, S.urz, S.q0z);
                const float mx = -qx * S.rx, my = -qy * S.ry, mz = -qz * S.rz;
                const float lo = fmaxf(fmaxf(mx - S.hx, my - S.hy), mz - S.hz);
                const float hiJ = fminf(fminf(mx + S.hx, my + S.hy), mz + S.hz);
                const int jlo = (int)ceilf(fmaxf(lo, 0.f));
                const int jhi = (int)floorf(fminf(hiJ, Wm1));
                off = i * G.qs + jlo;
                n = i <= S.ihi ? jhi - jlo + 1 : 0;
            };

            // ---- fast mode: every wavefront its own steps, one list for the whole visit
            XVR_TICK(0);
            bool redo = khi - klo >= S16_STEPS;
            if (!redo) {
                for (int k = klo + wave; k <= khi; k += 4) {
                    const float al = linspace_sel(k, N, near_, far_, step);
                    if (!(al > 1e-12f)) { redo = true; break; }
                    StepC S;
                    step_setup(al, S);
                    for (int i0 = S.ilo; i0 <= S.ihi && !redo; i0 += 64) {
                        int off, n;
                        rows(S, i0 + lane, off, n);
                        if (!append(off, n, k - klo)) redo = true;
                    }
                    if (redo) break;
                }
                if (redo && lane == 0) s_ctl[par][2] = 1;
            }
            XVR_TICK(1);
            __syncthreads();   // the list is complete
            XVR_TICK(2);
            redo = redo || s_ctl[par][2] != 0;   // (uniform: the first operand is, where it is set before the barrier)
            if (!redo) {
                drain();
            } else {
                // ---- safe mode (rare): the list is thrown away; one step at a time, all four wavefronts on its rows, drained
                // whenever another 256 runs might not fit and after every step
                __syncthreads();
                if (tid == 0) { s_ctl[par][0] = 0; s_ctl[par][1] = 0; s_ctl[par][2] = 0; }
                __syncthreads();
                for (int k = klo; k <= khi; ++k) {
                    kbase = k;
                    const float al = linspace_sel(k, N, near_, far_, step);
                    if (al > 1e-12f) {
                        StepC S;
                        step_setup(al, S);
                        for (int i0 = S.ilo; i0 <= S.ihi; i0 += 256) {
                            int off, n;
                            rows(S, i0 + tid, off, n);
                            append(off, n, 0);
                            if (i0 + 256 <= S.ihi) {   // (uniform) more rows to come: make room
                                __syncthreads();
                                if (s_ctl[par][0] > S16_TAB - 256) drain(); else __syncthreads();
                            }
                        }
                    } else {
                        // alpha_k = 0: every ray's sample sits on the source; all pixels, if the source is inside the support box
                        const bool hit = fabsf(Cx) < HS && fabsf(Cy) < HS && fabsf(Cz) < HS;
                        if (hit)
                            for (int i0 = 0; i0 < G.H; i0 += 256) {
                                append((i0 + tid) * G.qs, i0 + tid < G.H ? G.W : 0, 0);
                                if (i0 + 256 < G.H) {
                                    __syncthreads();
                                    if (s_ctl[par][0] > S16_TAB - 256) drain(); else __syncthreads();
                                }
                            }
                    }
                    __syncthreads();
                    drain();
                    __syncthreads();   // (the flush's clears before the next step's adds; fast mode has the list barrier there)
                }
            }
        }
    }
#ifdef XVR_S16_TRACE
    if (tid == 0 && blk < 65536) {
        unsigned long long* o = g_s16_trace + 12 * blk;
        o[0] = t_start; o[1] = wall_clock64(); o[2] = n_visits; o[3] = n_samples;
        for (int i = 0; i < 8; ++i) o[4 + i] = tk[i];
    }
#endif
    float* out = G.gvol + ((size_t)(ox + lx) * G.D1 + (oy + ly)) * G.D2 + oz;
    if (ox + lx < G.D0 && oy + ly < G.D1) {
#pragma unroll
        for (int z = 0; z < 16; ++z)
            if (oz + z < G.D2 && acc[z] != 0.f) out[z] += acc[z];
    }
    }   // next brick
}

// ---------------------------------------------------------------------------------------------
// The brick-local splat, RAY-major: for the renders whose samples cannot be enumerated as runs on shared planes.
//   CLIP  spec.clip_to_volume: alpha_k = alpha_min(ray) + u_k (alpha_max - alpha_min)(ray) (k_trilinear_gather_px below);
//   MASK  mask -> channels with a gradient that differs between channels: a sample's upstream value is
//         gout[b][label(sample)][ray].
// Same bricks, cells, fixed point, flush, persistent workgroups and pose prefetch as k_trilinear_splat_b16; per (brick, pose)
// visit the threads take the rays f = thread, thread + 256, ... of the brick's pixel footprint (bounding box of the 8
// projected corners of its support box), clip each against the support box (slab test) and evaluate its samples inside with
// the forward's own alpha arithmetic.  A ray has ~5-25 samples in a brick and many rays of the footprint miss it: a lane
// whose ray is done waits until a quarter of the wavefront is idle, then those lanes set up their next rays together.
// The scale's bound on a voxel's sum: MASK alone -- the samples are the lattice of k_trilinear_splat_b16, same bound, with
// max |c g| over rays and channels; CLIP -- (rays that can cross a voxel's 2-cube) x max over the rays of (samples of the ray
// inside the 2-cube x |c| span), both from k_gather_prep.
// ---------------------------------------------------------------------------------------------
template <bool CLIP, bool MASK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 6))) void k_trilinear_splat_px(GatherArgs G) {
    if (*G.flag > __float_as_uint(GATHER_DEV_TOL)) return;  // not a lattice: the scatter kernel runs instead
    __shared__ __attribute__((aligned(16))) int cell[S16_CELLS];
    constexpr int PW = (int)(sizeof(PoseLattice) / sizeof(float));
    __shared__ float s_P[2][PW + 4];
    __shared__ int s_next[2];
    constexpr float HS = 8.5f, CO = 7.5f;
    const int tid = threadIdx.x;
    const int lx = tid >> 4, ly = tid & 15;
    const int N = G.sp.n_points;
    const float near_ = G.sp.near_, far_ = G.sp.far_;
    const float step = N > 1 ? (far_ - near_) / (float)(N - 1) : 0.f;
    const float inv_step = step > 0.f ? 1.f / step : 0.f;
    const float a0 = G.sp.a[0], a1 = G.sp.a[1], a2 = G.sp.a[2];
    const float b0 = G.sp.b[0], b1 = G.sp.b[1], b2 = G.sp.b[2];
    const float ea0 = HS / a0, ea1 = HS / a1, ea2 = HS / a2;   // half-size of the support box, in x coordinates
    const int n0 = (G.D0 + 15) / 16, n1 = (G.D1 + 15) / 16, n2 = (G.D2 + 15) / 16;
    for (int i = tid; i < S16_CELLS / 4; i += 256) reinterpret_cast<int4*>(cell)[i] = make_int4(0, 0, 0, 0);

    for (int turn = 0;; turn ^= 1) {
    if (tid == 0) s_next[turn] = (int)atomicAdd(G.flag + 1, 1u);
    __syncthreads();   // (also: the cells are clear, the previous brick's flush is done)
    const int blk = __builtin_amdgcn_readfirstlane(s_next[turn]);
    if (blk >= n0 * n1 * n2) break;
    int bx = blk / (n1 * n2), by = (blk / n2) % n1, bz = blk % n2;
    bx = (bx & 1) ? (n0 >> 1) - ((bx + 1) >> 1) : (n0 >> 1) + (bx >> 1);
    by = (by & 1) ? (n1 >> 1) - ((by + 1) >> 1) : (n1 >> 1) + (by >> 1);
    bz = (bz & 1) ? (n2 >> 1) - ((bz + 1) >> 1) : (n2 >> 1) + (bz >> 1);
    const int brick_id = (bx * n1 + by) * n2 + bz;
    const int ox = bx * 16, oy = by * 16, oz = bz * 16;
    const float fv[3] = {(float)ox, (float)oy, (float)oz};
    float xv[3];  // centre of the support box in x coordinates
#pragma unroll
    for (int i = 0; i < 3; ++i) xv[i] = uni((fv[i] + CO - G.sp.b[i]) / G.sp.a[i]);
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    int wd = 0;
    unsigned bits = G.cull[(size_t)brick_id * G.words];
    auto next_pose = [&]() -> int {
        while (!bits) {
            if (++wd >= G.words) return -1;
            bits = G.cull[(size_t)brick_id * G.words + wd];
        }
        const int p = wd * 32 + __builtin_ctz(bits);
        bits &= bits - 1;
        return p;
    };
    auto fetch = [&](const int p) -> float {
        return tid < PW ? reinterpret_cast<const float*>(G.poses + p)[tid] : __uint_as_float(G.cmax[(size_t)p * G.cmax_stride]);
    };
    int pc = next_pose(), visit = 0;
    if (pc >= 0 && tid <= PW) s_P[0][tid] = fetch(pc);
    __syncthreads();
    while (pc >= 0) {
        const int p = pc, pn = next_pose();
        float pre = 0.f;
        if (pn >= 0 && tid <= PW) pre = fetch(pn);
        const int cur = visit & 1;
        ++visit;
        pc = pn;
        const float* Pf = s_P[cur];
        const PoseLattice& P = *reinterpret_cast<const PoseLattice*>(Pf);
        const float cmax = Pf[PW];
        const float s0 = P.s[0], s1 = P.s[1], s2 = P.s[2];
        const float w0 = xv[0] - s0, w1 = xv[1] - s1, w2 = xv[2] - s2;
        const float av = uni(P.nh[0] * w0 + P.nh[1] * w1 + P.nh[2] * w2);
        const float en0 = P.nh[0] * ea0, en1 = P.nh[1] * ea1, en2 = P.nh[2] * ea2;
        const float da = fabsf(en0) + fabsf(en1) + fabsf(en2);
        const float amin = av - da, amax = av + da;   // alpha range of the support box on this pose's rays
        // pixel footprint: bounding box of the box's 8 projected corners; every ray when the box reaches the source plane
        int jlo = 0, jhi = -1, ilo = 0, ihi = -1;
        if (cmax != 0.f && amin > 1e-6f && amax >= G.cull_lo && amin <= G.cull_hi) {
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
        } else if (cmax != 0.f && amin <= 1e-6f && amax >= G.cull_lo) {
            jhi = G.W - 1;
            ihi = G.H - 1;
        }
        const int nc = __builtin_amdgcn_readfirstlane(jhi - jlo + 1), nr = __builtin_amdgcn_readfirstlane(ihi - ilo + 1);
        const int total = (nc > 0 && nr > 0) ? nc * nr : 0;
        if (total > 0 && cmax < INFINITY) {
            // bound on one voxel's sum over this pose (see the header)
            float tsum = (float)total * (CLIP ? 1.f : (float)N);
            if (amin > 1e-6f) {
                const float r3 = 1.7320508f;
                const float idc = __builtin_amdgcn_rcpf(amin * P.ecl), idr = __builtin_amdgcn_rcpf(amin * P.rperp);
                if (CLIP) {
                    tsum = fminf(tsum, 1.02f * (2.f * r3 * idc + 1.f) * (2.f * r3 * idr + 1.f));
                } else if (step > 0.f) {
                    const float idn = P.gn * inv_step;
                    const float mp = 2.f * r3 * idn + 1.f, mr = 2.f * r3 * idr + 1.f;
                    tsum = fminf(tsum, 1.02f * (idc * idr * (idn + r3) + mp * (r3 * idc + mr)));
                }
            }
            if (!(tsum == tsum)) tsum = (float)total * (float)N;
            const float bound = uni(tsum * cmax);
            const int ex = (int)(__float_as_uint(bound) >> 23) - 126;   // bound < 2^ex
            const float cs = uni(__uint_as_float((unsigned)(127 + 30 - ex) << 23)), ics = uni(__uint_as_float((unsigned)(127 - 30 + ex) << 23));
            const float4* __restrict__ q = G.q + (size_t)p * G.qn;
            const float2* __restrict__ q2 = G.q2 + (size_t)p * G.n;
            // a s + b - (first cell of the array) for the cells, relative to the centre of the support box for the slab test
            const float Bx = uni(fmaf(a0, s0, b0 - (fv[0] - 1.f))), By = uni(fmaf(a1, s1, b1 - (fv[1] - 1.f))), Bz = uni(fmaf(a2, s2, b2 - (fv[2] - 1.f)));
            const float Cx = uni(fmaf(a0, s0, b0 - fv[0]) - CO), Cy = uni(fmaf(a1, s1, b1 - fv[1]) - CO), Cz = uni(fmaf(a2, s2, b2 - fv[2]) - CO);
            const float inc = 1.f / (float)nc;

            int f = tid;             // the lane's next ray of the footprint
            bool active = false;
            float tx = 0.f, ty = 0.f, tz = 0.f, tw = 0.f, lo = 0.f, span = 1.f;
            float ddx = 0.f, ddy = 0.f, ddz = 0.f;   // MASK: the forward's own d, for the label
            int k = 0, khi = -1, ray = 0;
            while (true) {
                const unsigned long long idle = __ballot(!active && f < total);
                if (idle != 0ull && (__popcll(idle) >= 16 || __ballot(active) == 0ull)) {
                    if (!active && f < total) {
                        // ---- set up ray f: the alphas where it is inside the support box -> its sample indices
                        const int ri = (int)(((float)f + 0.5f) * inc), rj = f - ri * nc;
                        const int i = ilo + ri, j = jlo + rj;
                        f += 256;
                        const float4 t = q[(size_t)i * G.qs + j];   // a * d, (g *) L / N (* span)
                        ray = i * G.W + j;
                        if (CLIP) {
                            const float2 ab = q2[ray];
                            lo = ab.x;
                            span = ab.y - ab.x;      // exactly the forward's (amax - amin)
                        }
                        tx = t.x; ty = t.y; tz = t.z; tw = t.w * cs;
                        if (MASK) {   // d exactly as the forward forms it, (target - source) + eps
                            const float* T = G.target + ((size_t)p * G.n + ray) * 3;
                            ddx = (T[0] - s0) + G.sp.eps; ddy = (T[1] - s1) + G.sp.eps; ddz = (T[2] - s2) + G.sp.eps;
                        }
                        const float ix = fabsf(tx) < 1e-12f ? copysignf(1e12f, tx) : 1.f / tx;
                        const float iy = fabsf(ty) < 1e-12f ? copysignf(1e12f, ty) : 1.f / ty;
                        const float iz = fabsf(tz) < 1e-12f ? copysignf(1e12f, tz) : 1.f / tz;
                        const float x0 = (-HS - Cx) * ix, x1 = (HS - Cx) * ix, y0 = (-HS - Cy) * iy, y1 = (HS - Cy) * iy;
                        const float z0 = (-HS - Cz) * iz, z1 = (HS - Cz) * iz;
                        const float e0 = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
                        const float e1 = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
                        k = 0;
                        khi = -1;
                        if (e1 >= e0 && (!CLIP || span > 0.f) && tw != 0.f) {
                            const float u0 = CLIP ? (e0 - lo) / span : e0, u1 = CLIP ? (e1 - lo) / span : e1;
                            if (step > 0.f) {   // one step of slack each side: the cell test decides, not the window
                                k = (int)ceilf(fmaxf((u0 - near_) * inv_step - 1.f, 0.f));
                                khi = (int)floorf(fminf((u1 - near_) * inv_step + 1.f, (float)(N - 1)));
                            } else {
                                khi = 0;
                            }
                        }
                        active = k <= khi;
                    }
                }
                if (__ballot(active) == 0ull) {
                    if (__ballot(f < total) == 0ull) break;
                    continue;
                }
                if (active) {
                    // ---- one sample, with the forward's alpha and position arithmetic
                    const float u = linspace_at(k, N, near_, far_, step);
                    const float al = CLIP ? fmaf(u, span, lo) : u;
                    const float px = fmaf(al, tx, Bx), py = fmaf(al, ty, By), pz = fmaf(al, tz, Bz);
                    const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
                    const float rx = px - fx, ry = py - fy, rz = pz - fz;
                    const int ix = (int)fx, iy = (int)fy, iz = (int)fz;
                    const bool ok = (unsigned)ix < (unsigned)(S16_DIM - 1) && (unsigned)iy < (unsigned)(S16_DIM - 1) && (unsigned)iz < (unsigned)(S16_DIM - 1);
                    float cw = tw;
                    if (MASK) {
                        // the channel of the sample = the label of its nearest voxel (0 outside the volume), from the forward's own
                        // position arithmetic, a (s + alpha d) + b as two fmas, bit for bit: a sample within an ulp of the middle
                        // between two voxels must get the label the forward gave it
                        const int mx = (int)rintf(fmaf(a0, fmaf(al, ddx, s0), b0)), my = (int)rintf(fmaf(a1, fmaf(al, ddy, s1), b1)),
                                  mz = (int)rintf(fmaf(a2, fmaf(al, ddz, s2), b2));
                        const bool in = (unsigned)mx < (unsigned)G.D0 && (unsigned)my < (unsigned)G.D1 && (unsigned)mz < (unsigned)G.D2;
                        int lab = 0;
                        if (ok && in) lab = min(max((int)G.mask[((size_t)mx * G.D1 + my) * G.D2 + mz], 0), G.C - 1);
                        cw *= ok ? G.gout[((size_t)p * G.C + lab) * G.n + ray] : 0.f;
                    }
                    const float cz = ok ? cw : 0.f;
                    const int base = ok ? (ix * S16_DIM + iy) * S16_DIM + iz : 0;
                    const float z1 = rz * cz, z0 = cz - z1;
                    const float x1 = rx, x0 = 1.f - rx, y1 = ry, y0 = 1.f - ry;
                    const float p00 = x0 * y0, p01 = x0 * y1, p10 = x1 * y0, p11 = x1 * y1;
                    int* c = cell + base;
                    __hip_atomic_fetch_add(c, cvt_nearest(p00 * z0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(c + 1, cvt_nearest(p00 * z1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(c + S16_DIM, cvt_nearest(p01 * z0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(c + S16_DIM + 1, cvt_nearest(p01 * z1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(c + S16_DIM * S16_DIM, cvt_nearest(p10 * z0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(c + S16_DIM * S16_DIM + 1, cvt_nearest(p10 * z1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(c + S16_DIM * S16_DIM + S16_DIM, cvt_nearest(p11 * z0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(c + S16_DIM * S16_DIM + S16_DIM + 1, cvt_nearest(p11 * z1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    ++k;
                    active = k <= khi;
                }
            }
            if (pn >= 0 && tid <= PW) s_P[cur ^ 1][tid] = pre;
            __syncthreads();   // every sample is in the cells
            int* col = cell + ((lx + 1) * S16_DIM + (ly + 1)) * S16_DIM + 1;
#pragma unroll
            for (int z = 0; z < 16; ++z) {
                acc[z] = fmaf((float)col[z], ics, acc[z]);
                col[z] = 0;
            }
            __syncthreads();   // the cells are clear
        } else {
            if (total > 0) {   // a non-finite upstream gradient poisons the brick
#pragma unroll
                for (int z = 0; z < 16; ++z) acc[z] = NAN;
            }
            if (pn >= 0 && tid <= PW) s_P[cur ^ 1][tid] = pre;
            __syncthreads();
        }
    }
    float* out = G.gvol + ((size_t)(ox + lx) * G.D1 + (oy + ly)) * G.D2 + oz;
    if (ox + lx < G.D0 && oy + ly < G.D1) {
#pragma unroll
        for (int z = 0; z < 16; ++z)
            if (oz + z < G.D2 && acc[z] != 0.f) out[z] += acc[z];
    }
    }   // next brick
}

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
    const float near_ = G.sp.near_, far_ = G.sp.far_;
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
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int x = vx + (e >> 2 & 1), y = vy + (e >> 1 & 1), z = vz + (e & 1);
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


// ---------------------------------------------------------------------------------------------
// Siddon voxel gradient for NON-exact index maps (norm_dims_offset = +-1, align_corners = True: the variants SURVEY.md
// Appendix A recalls for upstream), round 2.  The voxel a segment is credited to is rint(a x_mid + b) of its MIDPOINT, which
// inside plane cell c is c + olo or c + olo + 1 per axis (the map drifts by less than a voxel over the volume:
// siddon_cell_offsets).  One lane owns one CELL and gathers, as k_siddon_gather_vol does for a voxel, the length of every
// ray inside it -- but splits it over 8 sums by where the forward's own midpoint arithmetic sends the segment.  The sums go
// to a [cell][8] scratch; k_siddon_cells_to_voxels then adds, for every voxel, the eight (cell, octant) entries that
// name it.  No atomics, deterministic; before this the non-exact maps took the fp32-atomic scatter (151 ms per C3 batch).
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
// entry / exit are one max3 / min3.  A candidate costs ~57 VALU for 8 voxels instead of 8 x 19.
// (Round 2 measured the per-lane FLATTENED window loop here too -- lane-private (row, column) cursor, same arithmetic:
//  13.5 ms against 12.2 ms.  Fewer trips, but the lanes of a wavefront drift onto different detector rows, and the four
//  loads of a trip then touch that many more cache lines; the nested loops keep the wavefront on one row at a time.)
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
    const float ea0 = 1.f / G.sp.a[0], ea1 = 1.f / G.sp.a[1], ea2 = 1.f / G.sp.a[2];   // half a block, in x coordinates
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
            const float nj = P.gc[0] * w0 + P.gc[1] * w1 + P.gc[2] * w2;
            const float ni = P.gr[0] * w0 + P.gr[1] * w1 + P.gr[2] * w2;
            int jlo = 0, jhi = -1, ilo = 0, ihi = -1;
            if (inb && amin > 1e-6f && amax >= 0.f && amin <= 1.f) {
                // the block is convex and in front of the source: its projection is the hull of its 8 projected
                // corners, whose bounding box is the exact pixel window (the (n +- dn) / alpha box over the alpha range
                // pairs extremes that no single point attains)
                const float en0 = P.nh[0] * ea0, en1 = P.nh[1] * ea1, en2 = P.nh[2] * ea2;
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
                  void* workspace, void* stream, unsigned** flag_out, const float* mask, int C, const int* siddon_olo) {
    char* ws = static_cast<char*>(workspace);
    GatherArgs G = {};
    if (siddon && siddon_olo) {   // non-exact index map: per-cell octant sums in the scratch behind the regular workspace
        G.cells = reinterpret_cast<float*>(ws + align256(ws_bytes(B, n, D0, D1, D2)));
        for (int k = 0; k < 3; ++k) G.olo[k] = siddon_olo[k];
    }
    G.mask = siddon ? nullptr : mask;
    G.C = C;
    G.clip = (!siddon && sp->clip_to_volume) ? 1 : 0;
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
    G.V = siddon ? 1 : gather_block();
    // Siddon: 2x2x2 voxels per lane in 8^3 bricks unless XVR_DRR_SIDDON_GATHER_BLOCK=1 (A/B switch: one voxel per
    // lane, 256 lanes on a 4 x 8 x 8 brick)
    // trilinear: the per-lane flattened (table) kernel unless XVR_DRR_GATHER_TABLE=0 (A/B switch: the round-1 nested kernel;
    // both are exact and give identical sums)
    static const bool use_table = [] { const char* e = getenv("XVR_DRR_GATHER_TABLE"); return !(e && e[0] == '0'); }();
    static const bool siddon_v1 = [] { const char* e = getenv("XVR_DRR_SIDDON_GATHER_BLOCK"); return e && e[0] == '1'; }();
    // trilinear without clip / per-channel masks: the brick-local fixed-point splat on 16^3 bricks (k_trilinear_splat_b16)
    // unless XVR_DRR_GATHER_SPLAT=0 (A/B switch: the voxel-driven table gather)
    // (read at every launch -- a getenv -- so that the tests can compare the two in one process)
    const bool use_splat = [] { const char* e = getenv("XVR_DRR_GATHER_SPLAT"); return !(e && e[0] == '0'); }();
    const bool splat = !siddon && use_splat && !sp->clip_to_volume && !mask;
    if (siddon && (siddon_v1 || G.cells)) { G.bd[0] = 4; G.bd[1] = 8; G.bd[2] = 8; }
    else if (siddon) { G.bd[0] = G.bd[1] = G.bd[2] = 8; }
    else {
        if (G.clip || G.mask) G.V = 2;   // (the pixel-major kernel is written for 2x2x2 blocks)
        G.bd[0] = G.bd[1] = G.bd[2] = 4 * G.V;
    }
    // clip_to_volume / per-channel mask: the ray-major splat unless XVR_DRR_GATHER_SPLAT=0 (A/B: the voxel-driven pixel-major gather)
    const bool psplat = !siddon && use_splat && !splat;
    if (splat || psplat) {   // the poses' maxima behind q's used part ([B][2 n] float4, H (W + 1) = n + H used): a line per pose if it fits
        G.cmax = reinterpret_cast<unsigned*>(G.q + (size_t)B * G.qn);
        const int room = 4 * (n - n / gw);
        G.cmax_stride = room < CMAX_STRIDE ? room : CMAX_STRIDE;
    }
    if (psplat) G.bd[0] = G.bd[1] = G.bd[2] = 16;
    if (splat) G.bd[0] = G.bd[1] = G.bd[2] = 16;
    G.cull = reinterpret_cast<unsigned*>(ws + ws_cull_off(B, n));
    G.words = (B + 31) / 32;
    G.gvol = grad_volume;
    *flag_out = G.flag;
    hipError_t e = hipMemsetAsync(G.flag, 0, 16, (hipStream_t)stream);
    if (e == hipSuccess && G.cmax) e = hipMemsetAsync(G.cmax, 0, (size_t)B * G.cmax_stride * sizeof(unsigned), (hipStream_t)stream);
    if (e != hipSuccess) return fail(XVR_DRR_E_LAUNCH, hipGetErrorString(e));
    hipLaunchKernelGGL(k_gather_prep, dim3((unsigned)((n + WG - 1) / WG), (unsigned)B), dim3(WG), 0,
                       (hipStream_t)stream, G);
    const long long bricks = n_bricks(D0, D1, D2, G.bd);
    if (bricks >= (1LL << 31)) return fail(XVR_DRR_E_UNSUPPORTED, "grid too large");
    hipLaunchKernelGGL(k_gather_cull, dim3((unsigned)((bricks + WG / 32 - 1) / (WG / 32))), dim3(WG), 0,
                       (hipStream_t)stream, G, (int)bricks);
    if (siddon && G.cells) {
        hipLaunchKernelGGL(k_siddon_gather_cells, dim3((unsigned)bricks), dim3(WG), 0, (hipStream_t)stream, G);
        const long long nvox = (long long)D0 * D1 * D2;
        hipLaunchKernelGGL(k_siddon_cells_to_voxels, dim3((unsigned)((nvox + WG - 1) / WG)), dim3(WG), 0, (hipStream_t)stream, G);
    }
    else if (siddon && siddon_v1) hipLaunchKernelGGL(k_siddon_gather_vol, dim3((unsigned)bricks), dim3(WG), 0, (hipStream_t)stream, G);
    else if (siddon) hipLaunchKernelGGL(k_siddon_gather_vol2, dim3((unsigned)bricks), dim3(64), 0, (hipStream_t)stream, G);
    else if (psplat) {
        const void* kern = G.clip ? (G.mask ? (const void*)k_trilinear_splat_px<true, true> : (const void*)k_trilinear_splat_px<true, false>)
                                  : (const void*)k_trilinear_splat_px<false, true>;
        int per_cu = 0, dev = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, 0) != hipSuccess || per_cu < 1) per_cu = 2;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        const long long resident = (long long)per_cu * cus;
        const dim3 grid((unsigned)(bricks < resident ? bricks : resident));
        if (G.clip && G.mask) hipLaunchKernelGGL((k_trilinear_splat_px<true, true>), grid, dim3(256), 0, (hipStream_t)stream, G);
        else if (G.clip) hipLaunchKernelGGL((k_trilinear_splat_px<true, false>), grid, dim3(256), 0, (hipStream_t)stream, G);
        else hipLaunchKernelGGL((k_trilinear_splat_px<false, true>), grid, dim3(256), 0, (hipStream_t)stream, G);
    }
    else if (G.clip && G.mask) hipLaunchKernelGGL((k_trilinear_gather_px<true, true>), dim3((unsigned)bricks), dim3(64), 0, (hipStream_t)stream, G);
    else if (G.clip) hipLaunchKernelGGL((k_trilinear_gather_px<true, false>), dim3((unsigned)bricks), dim3(64), 0, (hipStream_t)stream, G);
    else if (G.mask) hipLaunchKernelGGL((k_trilinear_gather_px<false, true>), dim3((unsigned)bricks), dim3(64), 0, (hipStream_t)stream, G);
    else if (splat) {
        // persistent workgroups: as many as run at once (the occupancy the runtime reports x the CUs), never more than bricks
        static const int resident = [] {
            int per_cu = 0, dev = 0, cus = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_trilinear_splat_b16, 256, 0) != hipSuccess || per_cu < 1) per_cu = 2;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
            return per_cu * cus;
        }();
        hipLaunchKernelGGL(k_trilinear_splat_b16, dim3((unsigned)(bricks < resident ? bricks : resident)), dim3(256), 0, (hipStream_t)stream, G);
    }
    else if (G.V == 2 && use_table && (unsigned)G.qn <= TAB_MAX_RAYS) hipLaunchKernelGGL(k_trilinear_gather_tab, dim3((unsigned)bricks), dim3(64), 0, (hipStream_t)stream, G);
    else if (G.V == 2) hipLaunchKernelGGL(k_trilinear_gather_vol<2>, dim3((unsigned)bricks), dim3(64), 0, (hipStream_t)stream, G);
    else hipLaunchKernelGGL(k_trilinear_gather_vol<1>, dim3((unsigned)bricks), dim3(64), 0, (hipStream_t)stream, G);
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
    return align256(base) + siddon_cells_bytes(D0, D1, D2);
}

}  // extern "C"
