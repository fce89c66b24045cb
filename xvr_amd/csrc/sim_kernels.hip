// Fused image-similarity step of xvr's registration loop for MI355X (gfx950): XrayTransforms
// (Standardize by the global min/max -> Normalize) + multiscale NCC + gradient NCC, value AND exact
// gradient w.r.t. the raw rendered image, in seven small launches instead of ~100 torch kernels.
// C ABI: include/xvr_sim.h.  Reference being replaced:
//   /root/reference/src/xvr/registrar/base.py:115-123 (imagesim), :250-252 (transform, loss, backward)
//   /root/reference/src/xvr/utils/preprocess.py:5-31  (XrayTransforms)
//
// The reference's patch NCC unfolds every p x p patch into a channel (81x / 121x the image); here
// each thread evaluates one patch from an LDS tile with separable box sums (p + p reads, not p^2) of the
// five moments about a tile-wide shift, summed in doubles -- robust on the flat patches where eps decides -- and stores four
// per-patch maps
//   A = 1/s, Bm = mu_f/s, Cm = cov/(v_y s), Dm = cov mu_y/(v_y s),   s = sqrt(v_f v_y),
// from which the gradient is   d ncc / d y_i = (1 / (N_p p^2)) * ( f_i SA - SB - y_i SC + SD ),
// S* = sums of the maps over the patches that contain pixel i (the adjoint box filter).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>

#include "xvr_drr.h"
#include "xvr_pose.h"
#include "xvr_sim.h"
#include "j2c_device.hiph"
#include "pose_device.hiph"

extern "C" void xvr_drr_set_last_error(const char* msg);  // drr_api.hip

namespace {

constexpr int TB = 256;
constexpr int TILE = 16;      // output tile edge of the patch kernels
constexpr int MAXP = 15;      // largest supported patch edge
constexpr int N_ACC = 16;     // doubles per image
constexpr unsigned MM_BLOCKS_MAX = 256;   // blocks of the min / max reduction per group


struct SimHeader {           // first 256 bytes of the workspace
    unsigned enc_min, enc_max;
    int cnt_min, cnt_max;
    double smin, smax;       // sum G_i (1 - x_i), sum G_i x_i  over the whole batch
};

__device__ __forceinline__ unsigned enc(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float dec(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Order-deterministic grid reduction of NV doubles per thread: block `blk` of `nblk` stores its partial
// sums, the block that draws the last ticket adds all partials in a fixed order and WRITES dst[0..NV).
// No floating-point atomics: the similarity and its gradient are the same bits on every run.
// (call at most once per kernel: static shared memory)
template <int NV>
__device__ __forceinline__ void grid_add_det(double (&v)[NV], double* partial, int blk, int nblk, unsigned* counter,
                                             double* dst) {
    constexpr int NG = TB / NV;   // groups of NV threads that add the partials
    __shared__ double part[NV][TB / 64];
    __shared__ double fin[NG][NV];
    __shared__ bool last;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const double t = wave_sum_d(v[i]);
        if ((threadIdx.x & 63) == 0) part[i][threadIdx.x >> 6] = t;
    }
    __syncthreads();
    // Publishing without a device-wide fence (a release fence writes the XCD's whole L2 back, ~70 ns per
    // block, serialised): partials are stored and loaded with agent-scope atomics, which go to the level
    // where the 8 XCDs are coherent; the barrier's wait for outstanding stores orders them before the ticket.
    if (threadIdx.x < NV) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < TB / 64; ++w) t += part[threadIdx.x][w];
        __hip_atomic_store(partial + (size_t)blk * NV + threadIdx.x, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();   // includes s_waitcnt vmcnt(0): the stores above have completed
    if (threadIdx.x == 0)
        last = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(nblk - 1);
    __syncthreads();
    if (!last) return;
    // (every block has drawn its ticket: the counter goes back to 0, so that a second launch over the same state -- a backward
    //  run twice, retain_graph -- draws a full set of tickets again instead of never seeing nblk - 1)
    if (threadIdx.x == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int q = threadIdx.x % NV, grp = threadIdx.x / NV;
    if (grp < NG) {
        double t = 0.0;
#pragma unroll 4
        for (int k = grp; k < nblk; k += NG)
            t += __hip_atomic_load(partial + (size_t)k * NV + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        fin[grp][q] = t;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        double t = 0.0;
        for (int k = 0; k < NG; ++k) t += fin[k][threadIdx.x];
        dst[threadIdx.x] = t;
    }
}

// header + accumulators: everything zero except enc_min = 0xffffffff (one launch instead of two memsets)
__global__ __launch_bounds__(TB) void k_sim_init(unsigned* __restrict__ w, int nwords, int header_words) {
    const int i = blockIdx.x * TB + threadIdx.x;
    constexpr int HW_ = (int)(sizeof(SimHeader) / 4);   // enc_min is the first word of every header
    if (i < nwords) w[i] = (i < header_words && i % HW_ == 0) ? 0xffffffffu : 0u;
}

// global min / max of the batch (Standardize takes them over the whole tensor): few blocks, one atomic
// pair per block (1024 same-address atomics cost 25 us; 64 cost nothing)
// (gridDim.y = 1: over the whole batch tensor, the reference's Standardize; = B: per image, spec.per_image)
__global__ __launch_bounds__(TB) void k_sim_minmax(const float* __restrict__ m, long long n, SimHeader* hd) {
    __shared__ float plo[TB / 64], phi[TB / 64];
    float lo = INFINITY, hi = -INFINITY;
    m += (size_t)blockIdx.y * n;
    hd += blockIdx.y;
    for (long long i = (long long)blockIdx.x * TB + threadIdx.x; i < n; i += (long long)gridDim.x * TB) {
        const float v = m[i];
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o));
        hi = fmaxf(hi, __shfl_xor(hi, o));
    }
    if ((threadIdx.x & 63) == 0) {
        plo[threadIdx.x >> 6] = lo;
        phi[threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < TB / 64; ++w) {
            lo = fminf(lo, plo[w]);
            hi = fmaxf(hi, phi[w]);
        }
        atomicMin(&hd->enc_min, enc(lo));
        atomicMax(&hd->enc_max, enc(hi));
    }
}

// The same min / max WITHOUT the header's initial state (round 6, the registration step): blocks store their pair, the block that
// draws the group's last ticket takes the min / max over the pairs -- exact in any order -- and WRITES the header: enc_min, enc_max,
// and the counts k_sim_prep adds to from zero.  Nothing has to be reset before the call (k_sim_init is not launched: one launch
// fewer in a chain of a few microseconds each); the ticket goes back to zero.  Needs gridDim.x <= MM_BLOCKS_MAX.
__global__ __launch_bounds__(TB) void k_sim_minmax_det(const float* __restrict__ m, long long n, SimHeader* hd, float* partial, unsigned* tickets) {
    __shared__ float plo[TB / 64], phi[TB / 64];
    __shared__ bool last;
    float lo = INFINITY, hi = -INFINITY;
    m += (size_t)blockIdx.y * n;
    hd += blockIdx.y;
    partial += (size_t)blockIdx.y * MM_BLOCKS_MAX * 2;
    for (long long i = (long long)blockIdx.x * TB + threadIdx.x; i < n; i += (long long)gridDim.x * TB) {
        const float v = m[i];
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
    }
    auto block_minmax = [&]() {   // -> thread 0 holds the block's min / max
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo = fminf(lo, __shfl_xor(lo, o));
            hi = fmaxf(hi, __shfl_xor(hi, o));
        }
        if ((threadIdx.x & 63) == 0) {
            plo[threadIdx.x >> 6] = lo;
            phi[threadIdx.x >> 6] = hi;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
#pragma unroll
            for (int w = 1; w < TB / 64; ++w) {
                lo = fminf(lo, plo[w]);
                hi = fmaxf(hi, phi[w]);
            }
        }
    };
    block_minmax();
    if (threadIdx.x == 0) {
        __hip_atomic_store(partial + 2 * blockIdx.x, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(partial + 2 * blockIdx.x + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();   // includes s_waitcnt vmcnt(0): the stores above have completed
    if (threadIdx.x == 0)
        last = __hip_atomic_fetch_add(tickets + blockIdx.y, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    __syncthreads();
    if (!last) return;
    lo = INFINITY; hi = -INFINITY;
    if (threadIdx.x < gridDim.x) {
        lo = __hip_atomic_load(partial + 2 * threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        hi = __hip_atomic_load(partial + 2 * threadIdx.x + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();   // (plo / phi are reused)
    block_minmax();
    if (threadIdx.x == 0) {
        hd->enc_min = enc(lo);
        hd->enc_max = enc(hi);
        hd->cnt_min = 0;
        hd->cnt_max = 0;
        __hip_atomic_store(tickets + blockIdx.y, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// y = ((m - min) / (max - min + std_eps) - mean) / std, the five global moments of (f, y) per image, how
// many pixels attain the min / the max (their gradient is shared evenly), and the 3x3 Sobel pair of y
// with zero padding 1 (torch conv2d = cross-correlation) -- y at the 8 neighbours is recomputed from m
// with the same expression, so the Sobel input is bit-identical to the stored y.
__global__ __launch_bounds__(TB) void k_sim_prep(const float* __restrict__ m, const float* __restrict__ f, int H, int W,
                                                 SimHeader* hd, xvr_sim_spec sp, float* __restrict__ y,
                                                 float* __restrict__ g, double* acc, double* partial, unsigned* tickets) {
    const int b = blockIdx.y, hw = H * W;
    hd += sp.per_image ? b : 0;
    const float mn = dec(hd->enc_min), mx = dec(hd->enc_max);
    const float r = (mx - mn) + sp.std_eps;
    const float* M = m + (size_t)b * hw;
    const bool pre = sp.pre_transformed != 0;   // the moving image is already transformed: y = m
    auto yof = [&](float v) { return pre ? v : (((v - mn) / r) - sp.mean) / sp.std; };
    auto at = [&](int rr, int cc) { return (rr >= 0 && rr < H && cc >= 0 && cc < W) ? yof(M[rr * W + cc]) : 0.f; };
    double s[5] = {0, 0, 0, 0, 0};
    int cmin = 0, cmax = 0;
    for (int i = blockIdx.x * TB + threadIdx.x; i < hw; i += gridDim.x * TB) {
        const size_t k = (size_t)b * hw + i;
        const float v = M[i];
        const float yy = yof(v);
        const float ff = f[k];
        y[k] = yy;
        s[0] += yy; s[1] += (double)yy * yy; s[2] += ff; s[3] += (double)ff * ff; s[4] += (double)ff * yy;
        cmin += v == mn;
        cmax += v == mx;
        if (sp.beta >= 1.f) continue;   // no gradient-NCC term: nobody reads the Sobel pair
        const int rr = i / W, c = i - rr * W;
        const float a00 = at(rr - 1, c - 1), a01 = at(rr - 1, c), a02 = at(rr - 1, c + 1);
        const float a10 = at(rr, c - 1), a12 = at(rr, c + 1);
        const float a20 = at(rr + 1, c - 1), a21 = at(rr + 1, c), a22 = at(rr + 1, c + 1);
        g[((size_t)b * 2 + 0) * hw + i] = (a00 - a02) + 2.f * (a10 - a12) + (a20 - a22);
        g[((size_t)b * 2 + 1) * hw + i] = (a00 + 2.f * a01 + a02) - (a20 + 2.f * a21 + a22);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        cmin += __shfl_xor(cmin, o);
        cmax += __shfl_xor(cmax, o);
    }
    if ((threadIdx.x & 63) == 0) {   // integer atomics: exact in any order
        if (cmin) atomicAdd(&hd->cnt_min, cmin);
        if (cmax) atomicAdd(&hd->cnt_max, cmax);
    }
    grid_add_det<5>(s, partial + (size_t)b * gridDim.x * 5, blockIdx.x, gridDim.x, tickets + b, acc + (size_t)b * N_ACC);
}

// The three patch NCCs of the similarity (local mNCC term on the image, gradient NCC on the two Sobel
// channels) run as ONE launch: blockIdx.z = job * B + image.  Each is only a few hundred workgroups.
struct PatchJob {
    const float* f;      // [B][nch][H][W], channel ch
    const float* y;
    int nch, ch, p;
    float* maps;         // [4][B][Hp][Wp]
    int acc_slot;
    float scale;         // backward: weight of this term
    float* G;            // backward: [B][g_nch][H][W], channel g_ch
    int g_nch, g_ch;
};
struct PatchJobs {
    PatchJob j[3];
};

// one thread = one patch.  fimg / yimg: [B][nch][H][W] with channel `ch` selected.  maps: [4][B][Hp][Wp].
// tpb = tile rows per workgroup (blockIdx.y covers tile rows blockIdx.y * tpb ...): 1 for a registration iteration (as many
// workgroups as possible), 4 for batches -- a quarter of the deterministic grid reductions, which are most of a workgroup's time
__global__ __launch_bounds__(TB) void k_sim_patch(PatchJobs jobs, int B, int H, int W, float eps, double* acc,
                                                  double* partial, unsigned* tickets, int tpb) {
    __shared__ float sf[(TILE + MAXP - 1) * (TILE + MAXP - 1)];
    __shared__ float sy[(TILE + MAXP - 1) * (TILE + MAXP - 1)];
    __shared__ double hs[5][(TILE + MAXP - 1) * TILE];
    const int job = blockIdx.z / B, b = blockIdx.z - job * B;
    const PatchJob J = jobs.j[job];
    const float* __restrict__ fimg = J.f;
    const float* __restrict__ yimg = J.y;
    float* __restrict__ maps = J.maps;
    const int nch = J.nch, ch = J.ch, p = J.p, acc_slot = J.acc_slot;
    const int Hp = H - p + 1, Wp = W - p + 1;
    const int ox0 = blockIdx.x * TILE;
    if ((int)blockIdx.y * tpb * TILE >= Hp || ox0 >= Wp) return;   // the grid is sized for the job with the smallest patch
    const int E = TILE + p - 1;
    double ncc_d = 0.0;
    for (int row = 0; row < tpb; ++row) {
    const int oy0 = ((int)blockIdx.y * tpb + row) * TILE;
    if (oy0 >= Hp) break;
    if (row) __syncthreads();   // the previous tile's shared arrays have been read
    const float* F = fimg + ((size_t)b * nch + ch) * H * W;
    const float* Y = yimg + ((size_t)b * nch + ch) * H * W;
    for (int t = threadIdx.x; t < E * E; t += TB) {
        const int rr = oy0 + t / E, cc = ox0 + t % E;
        const bool in = rr < H && cc < W;
        sf[t] = in ? F[rr * W + cc] : 0.f;
        sy[t] = in ? Y[rr * W + cc] : 0.f;
    }
    __syncthreads();
    // Separable box sums of the five moments about a tile-wide shift (cf, cy) = the tile's centre pixel: p + p reads per
    // output instead of 2 p^2.  The sums are DOUBLES (round 3): a DRR is flat over most of its background, and on a flat patch
    // away from the tile's centre the one-pass variance S2/n - (S1/n)^2 and covariance cancel delta^2 (delta = the patch's
    // value minus the shift, ~3 after Normalize) down to exactly 0 -- in float the residue is delta^2 * 1e-7 ~ 1e-6, not small
    // against eps = 1e-5: whole flat regions contributed +-0.1 per patch instead of 0 (the reference's two-pass unfold
    // formulation gives 0), 2-7 % of the image's patch NCC (tests/test_c4_c5.py found it).  float x float is exact in double
    // and 81-225 such terms lose nothing.
    const int ec = min(E / 2, min(H - 1 - oy0, W - 1 - ox0));
    const float cf = sf[ec * E + ec], cy = sy[ec * E + ec];
    for (int t = threadIdx.x; t < E * TILE; t += TB) {   // horizontal: row r of the tile, output column c
        const int r = t / TILE, c = t - r * TILE;
        double a1 = 0.0, b1 = 0.0, a2 = 0.0, b2 = 0.0, ab = 0.0;
        for (int v = 0; v < p; ++v) {
            const double a = (double)(sf[r * E + c + v] - cf), bb = (double)(sy[r * E + c + v] - cy);
            a1 += a; b1 += bb;
            a2 = fma(a, a, a2); b2 = fma(bb, bb, b2); ab = fma(a, bb, ab);
        }
        hs[0][t] = a1; hs[1][t] = b1; hs[2][t] = a2; hs[3][t] = b2; hs[4][t] = ab;
    }
    __syncthreads();
    const int ty = threadIdx.x / TILE, tx = threadIdx.x % TILE;
    const int oy = oy0 + ty, ox = ox0 + tx;
    if (oy < Hp && ox < Wp) {
        const double inv = 1.0 / (double)(p * p);
        double s1f = 0.0, s1y = 0.0, s2f = 0.0, s2y = 0.0, sfy = 0.0;
        for (int u = 0; u < p; ++u) {                     // vertical
            const int t = (ty + u) * TILE + tx;
            s1f += hs[0][t]; s1y += hs[1][t]; s2f += hs[2][t]; s2y += hs[3][t]; sfy += hs[4][t];
        }
        const double mfs = s1f * inv, mys = s1y * inv;      // means relative to the shift
        const float mf = (float)(mfs + (double)cf), my = (float)(mys + (double)cy);
        const float vf = (float)fmax(fma(-mfs, mfs, s2f * inv), 0.0) + eps, vy = (float)fmax(fma(-mys, mys, s2y * inv), 0.0) + eps;
        const float cv = (float)fma(-mfs, mys, sfy * inv);
        const float s = sqrtf(vf * vy);
        const float ncc = cv / s;
        const size_t np = (size_t)Hp * Wp, o = ((size_t)b * Hp + oy) * Wp + ox, st = (size_t)B * np;
        maps[o] = 1.f / s;
        maps[st + o] = mf / s;
        maps[2 * st + o] = cv / (vy * s);
        maps[3 * st + o] = cv * my / (vy * s);
        ncc_d += ncc;
    }
    }   // next tile row
    double v1[1] = {ncc_d};
    const int ntx = (Wp + TILE - 1) / TILE, nty = ((Hp + TILE - 1) / TILE + tpb - 1) / tpb;   // the workgroups of THIS job
    grid_add_det<1>(v1, partial + (size_t)blockIdx.z * gridDim.x * gridDim.y, blockIdx.y * ntx + blockIdx.x, ntx * nty,
                    tickets + blockIdx.z, acc + (size_t)b * N_ACC + acc_slot);
}

// G[i] (+)= scale * ( f_i SA - SB - y_i SC + SD ),  S* = sums of the maps over the patches containing i
__global__ __launch_bounds__(TB) void k_sim_patch_grad(PatchJobs jobs, int B, int H, int W) {
    __shared__ float sm[4][(TILE + MAXP - 1) * (TILE + MAXP - 1)];
    __shared__ float hs[4][(TILE + MAXP - 1) * TILE];
    const int job = blockIdx.z / B, b = blockIdx.z - job * B;
    const PatchJob J = jobs.j[job];
    const float* __restrict__ fimg = J.f;
    const float* __restrict__ yimg = J.y;
    const float* __restrict__ maps = J.maps;
    float* __restrict__ G = J.G;
    const int nch = J.nch, ch = J.ch, p = J.p, g_nch = J.g_nch, g_ch = J.g_ch;
    const float scale = J.scale;
    const int Hp = H - p + 1, Wp = W - p + 1;
    const int r0 = blockIdx.y * TILE, c0 = blockIdx.x * TILE;
    const int E = TILE + p - 1;
    const size_t np = (size_t)Hp * Wp, st = (size_t)B * np;
    // patches containing pixel (r, c) have origins in [r - p + 1, r] x [c - p + 1, c]: stage that window
    for (int t = threadIdx.x; t < E * E; t += TB) {
        const int oy = r0 - (p - 1) + t / E, ox = c0 - (p - 1) + t % E;
        const bool in = oy >= 0 && oy < Hp && ox >= 0 && ox < Wp;
        const size_t o = ((size_t)b * Hp + oy) * Wp + ox;
#pragma unroll
        for (int k = 0; k < 4; ++k) sm[k][t] = in ? maps[k * st + o] : 0.f;
    }
    __syncthreads();
    // separable box sum of the four maps: horizontal into hs, then vertical (p + p reads instead of p^2)
    for (int t = threadIdx.x; t < E * TILE; t += TB) {
        const int rr = t / TILE, cc = t - rr * TILE;
        float h0 = 0.f, h1 = 0.f, h2 = 0.f, h3 = 0.f;
        for (int v = 0; v < p; ++v) {
            const int q = rr * E + cc + v;
            h0 += sm[0][q]; h1 += sm[1][q]; h2 += sm[2][q]; h3 += sm[3][q];
        }
        hs[0][t] = h0; hs[1][t] = h1; hs[2][t] = h2; hs[3][t] = h3;
    }
    __syncthreads();
    const int ty = threadIdx.x / TILE, tx = threadIdx.x % TILE;
    const int r = r0 + ty, c = c0 + tx;
    if (r >= H || c >= W) return;
    float sa = 0.f, sb = 0.f, sc = 0.f, sd = 0.f;
    for (int u = 0; u < p; ++u) {
        const int t = (ty + u) * TILE + tx;
        sa += hs[0][t]; sb += hs[1][t]; sc += hs[2][t]; sd += hs[3][t];
    }
    const size_t i = ((size_t)b * nch + ch) * H * W + (size_t)r * W + c;
    const float g = scale * (fimg[i] * sa - sb - yimg[i] * sc + sd);
    G[((size_t)b * g_nch + g_ch) * H * W + (size_t)r * W + c] = g;
}

// total gradient w.r.t. y: local term (already in Gy) + global NCC term + Sobel^T of the gradient-NCC
// terms; direct part of d/d moving = a * G; accumulates the sums the min/max terms need.
__global__ __launch_bounds__(TB) void k_sim_final(const float* __restrict__ m, const float* __restrict__ f,
                                                  const float* __restrict__ y, const float* __restrict__ Gy,
                                                  const float* __restrict__ Gg, int H, int W, SimHeader* hd,
                                                  const double* __restrict__ acc, xvr_sim_spec sp,
                                                  float* __restrict__ grad, double* partial, unsigned* tickets) {
    const int b = blockIdx.y, hw = H * W;
    const int i = blockIdx.x * TB + threadIdx.x;
    hd += sp.per_image ? b : 0;
    const float mn = dec(hd->enc_min), mx = dec(hd->enc_max);
    const float r = (mx - mn) + sp.std_eps;
    const float a = sp.pre_transformed ? 1.f : 1.f / (r * sp.std);
    const double* A = acc + (size_t)b * N_ACC;
    const double n = (double)hw;
    const double muy = A[0] / n, muf = A[2] / n;
    const double vy = A[1] / n - muy * muy + sp.ncc_eps, vf = A[3] / n - muf * muf + sp.ncc_eps;
    const double cov = A[4] / n - muf * muy;
    const double sg = sqrt(vf * vy);
    double s2[2] = {0.0, 0.0};
    if (i < hw) {
        const int rr = i / W, cc = i - rr * W;
        const size_t k = (size_t)b * hw + i;
        const float yy = y[k], ff = f[k];
        float G = sp.beta > 0.f ? Gy[k] : 0.f;   // (a term with zero weight is not computed at all)
        G += (float)(0.5 * sp.beta / n * ((ff - muf) / sg - cov * (yy - muy) / (vy * sg)));
        // Sobel^T: dL/dy[r,c] = sum_{u,v} K[u][v] * Gg[r - u + 1, c - v + 1]
        const float* gx = Gg + ((size_t)b * 2 + 0) * hw;
        const float* gy = Gg + ((size_t)b * 2 + 1) * hw;
        auto at = [&](const float* P, int r2, int c2) { return (r2 >= 0 && r2 < H && c2 >= 0 && c2 < W) ? P[r2 * W + c2] : 0.f; };
        if (sp.beta < 1.f) {
            // Kx = [[1,0,-1],[2,0,-2],[1,0,-1]]
            G += at(gx, rr + 1, cc + 1) - at(gx, rr + 1, cc - 1) + 2.f * (at(gx, rr, cc + 1) - at(gx, rr, cc - 1)) +
                 at(gx, rr - 1, cc + 1) - at(gx, rr - 1, cc - 1);
            // Ky = [[1,2,1],[0,0,0],[-1,-2,-1]]
            G += at(gy, rr + 1, cc + 1) + 2.f * at(gy, rr + 1, cc) + at(gy, rr + 1, cc - 1) -
                 (at(gy, rr - 1, cc + 1) + 2.f * at(gy, rr - 1, cc) + at(gy, rr - 1, cc - 1));
        }
        const float x = (m[k] - mn) / r;
        if (grad) grad[k] = a * G;
        s2[0] = (double)G * (1.0 - x);
        s2[1] = (double)G * x;
    }
    // the two sums feed Standardize's min / max gradient (k_sim_minmax_grad): a moving image that arrives transformed has none,
    // and the reduction -- one group of B x blocks partials when the statistics are the batch's -- was most of this kernel's time
    // at the training loss's 116 images (0.36 ms)
    if (sp.pre_transformed) return;
    const bool pi = sp.per_image != 0;   // per image: one reduction group per image; else one over the batch
    grid_add_det<2>(s2, partial + (pi ? (size_t)b * gridDim.x * 2 : 0), pi ? (int)blockIdx.x : (int)(blockIdx.y * gridDim.x + blockIdx.x),
                    pi ? (int)gridDim.x : (int)(gridDim.x * gridDim.y), tickets + (pi ? b : 0), &hd->smin);
}

// Standardize's min and max are functions of the image too: their gradient goes, evenly, to every
// pixel that attains them (torch's full-reduction min/max backward)
__device__ __forceinline__ float sim_loss_of(const double* __restrict__ A, int H, int W, const xvr_sim_spec& sp) {
    const double n = (double)H * W;
    const double muy = A[0] / n, muf = A[2] / n;
    const double vy = A[1] / n - muy * muy + sp.ncc_eps, vf = A[3] / n - muf * muf + sp.ncc_eps;
    const double cov = A[4] / n - muf * muy;
    const double ncc_g = cov / sqrt(vf * vy);
    const double n1 = (double)(H - sp.mncc_patch + 1) * (W - sp.mncc_patch + 1);
    const double n2 = (double)(H - sp.gncc_patch + 1) * (W - sp.gncc_patch + 1);
    const double mncc = 0.5 * ncc_g + 0.5 * A[5] / n1;
    const double gncc = 0.5 * (A[6] + A[7]) / n2;
    return (float)(sp.beta * mncc + (1.0 - sp.beta) * gncc);
}

__global__ __launch_bounds__(TB) void k_sim_minmax_grad(const float* __restrict__ m, long long n, const SimHeader* hd,
                                                        xvr_sim_spec sp, float* __restrict__ grad,
                                                        const double* __restrict__ acc, int B, int H, int W,
                                                        float* __restrict__ loss) {
    if (blockIdx.x == 0 && blockIdx.y == 0)   // the similarity values ride along (saves the launch of k_sim_loss)
        for (int b = threadIdx.x; b < B; b += TB) loss[b] = sim_loss_of(acc + (size_t)b * N_ACC, H, W, sp);
    // gridDim.y = 1: n = the whole batch, one header; = B (spec.per_image): n = one image, header blockIdx.y
    m += (size_t)blockIdx.y * n;
    grad += (size_t)blockIdx.y * n;
    hd += blockIdx.y;
    const float mn = dec(hd->enc_min), mx = dec(hd->enc_max);
    const float r = (mx - mn) + sp.std_eps;
    const double a = 1.0 / ((double)r * sp.std);
    const float gmin = (float)(-a * hd->smin / (double)max(hd->cnt_min, 1));
    const float gmax = (float)(-a * hd->smax / (double)max(hd->cnt_max, 1));
    for (long long i = (long long)blockIdx.x * TB + threadIdx.x; i < n; i += (long long)gridDim.x * TB) {
        const float v = m[i];
        float add = 0.f;
        if (v == mn) add += gmin;
        if (v == mx) add += gmax;
        if (add != 0.f) grad[i] += add;
    }
}

// ---------------------------------------------------------------------------------------------
// Round 6: the tail of a registration iteration in ONE launch (xvr_sim_ncc_registration_step).  What four launches did --
//   k_sim_minmax_grad   the gradient's terms through Standardize's global min / max, and the similarity values
//   k_jac_to_cam        its contraction with the render's per-ray jacobian, 24 sums per pose in a fixed order (drr_rays.hip)
//   k_pose_opt_step     chain rule to (rot, xyz), Adam, ReduceLROnPlateau, stopping rule, history row (pose_kernels.hip)
//   k_pose_camera_fwd   the NEXT iteration's camera vector
// -- with the same expressions in the same order (j2c_device.hiph, pose_device.hiph): a block takes k_jac_to_cam's rays, adds the
// min / max terms to the image gradient in passing (never stored: nobody else reads it in the loop), and the block that draws a
// pose's last ticket finishes that pose.  At 256^2 an iteration is launch- and latency-bound (twelve dependent launches of a few
// microseconds each): three launches fewer and 0.5 MB less traffic.  Euler angles (kind 0) only.
// ---------------------------------------------------------------------------------------------
struct RegTail {
    const float* jac; float* cam; float* partial; unsigned* counter;                       // jacobian -> camera
    float* rot; float* xyz; const float* G; const float* c; xvr_pose_opt_state* state; float* history;   // optimiser step, next camera
    xvr_pose_opt_spec osp;
};

template <int RPT>
__global__ __launch_bounds__(TB) void k_sim_reg_tail(const float* __restrict__ m, const SimHeader* hd, xvr_sim_spec sp,
                                                     const float* __restrict__ grad, const double* __restrict__ acc, int B, int H, int W,
                                                     float* __restrict__ loss, RegTail T) {
    static_assert(TB == J2C_WG, "the tail takes k_jac_to_cam's blocks");
    __shared__ float s_g[24];
    const int b = blockIdx.y, n = H * W, nblk = gridDim.x;
    hd += sp.per_image ? b : 0;
    const float mn = dec(hd->enc_min), mx = dec(hd->enc_max);
    const float r_ = (mx - mn) + sp.std_eps;
    const double a = 1.0 / ((double)r_ * sp.std);
    const float gmin = (float)(-a * hd->smin / (double)max(hd->cnt_min, 1));
    const float gmax = (float)(-a * hd->smax / (double)max(hd->cnt_max, 1));
    const float* c = T.cam + 24 * b;
    float a24[24];
#pragma unroll
    for (int q = 0; q < 24; ++q) a24[q] = 0.f;
    float g_[RPT];
    float4 j0_[RPT], j1_[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {   // every load of the thread first
        const int r = (blockIdx.x * RPT + k) * TB + threadIdx.x;
        g_[k] = 0.f;
        j0_[k] = j1_[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < n) {
            const size_t ray = (size_t)b * n + r;
            const float v = m[ray];
            float g = grad[ray], add = 0.f;
            if (v == mn) add += gmin;      // (k_sim_minmax_grad's `grad[i] += add`, where add != 0)
            if (v == mx) add += gmax;
            if (add != 0.f) g += add;
            g_[k] = g;
            const float4* jp = reinterpret_cast<const float4*>(T.jac + ray * XVR_DRR_JAC_STRIDE);
            j0_[k] = jp[0];
            j1_[k] = jp[1];
        }
    }
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int r = (blockIdx.x * RPT + k) * TB + threadIdx.x;
        if (r < n) j2c_accumulate(g_[k], j0_[k], j1_[k], c, r, W, a24);
    }
    if (!j2c_reduce(a24, T.partial, T.counter, b, nblk, s_g)) return;
    // ---- the pose's last block: every ray of pose b is in s_g.  One wavefront finishes the iteration (k_pose_opt_step's body).
    if (threadIdx.x >= 64) return;
    const float cur = sim_loss_of(acc + (size_t)b * N_ACC, H, W, sp);
    float gm[12];
    wave_gt_g(T.G, s_g, false, gm);
    xvr_pose_opt_state s = T.state[b];
    if (threadIdx.x == 0) loss[b] = cur;
    if (s.done) return;      // (rot / xyz untouched: the camera vector in place is still theirs)
    float p[6], g[6];
    for (int i = 0; i < 3; ++i) { p[i] = T.rot[(size_t)b * 3 + i]; p[3 + i] = T.xyz[b * 3 + i]; }
    Axes ax = {{T.osp.axes[0], T.osp.axes[1], T.osp.axes[2]}};
    pose_chain(ax, p, p + 3, gm, g, g + 3);
    // (every lane carries the scalars and takes the same step on its own copy; lane 0 writes the state, lane r < 24 row r of the camera)
    pose_opt_update(T.osp, 3, p, g, cur, s, threadIdx.x == 0 ? T.history : nullptr, b);
    float R[9], m12[12];
    pose_matrix(ax, p, p + 3, R, m12);
    if (threadIdx.x < 24) T.cam[b * 24 + threadIdx.x] = camera_row(T.G, T.c, m12, threadIdx.x);
    if (threadIdx.x == 0) {
        for (int i = 0; i < 3; ++i) { T.rot[(size_t)b * 3 + i] = p[i]; T.xyz[b * 3 + i] = p[3 + i]; }
        T.state[b] = s;
    }
}

__global__ void k_sim_loss(const double* __restrict__ acc, int B, int H, int W, xvr_sim_spec sp, float* __restrict__ loss) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) loss[b] = sim_loss_of(acc + (size_t)b * N_ACC, H, W, sp);
}

int sim_fail(int code, const char* msg) {
    xvr_drr_set_last_error(msg);  // one error buffer for the whole library (xvr_drr_last_error())
    return code;
}

size_t al(size_t v) { return (v + 255) & ~(size_t)255; }

struct Layout {
    size_t acc, tickets, y, gy, m1, m2, Gy, Gg, part_prep, part_patch, part_final, part_mm, total;
};
constexpr unsigned PREP_BLOCKS_MAX = 256;

Layout layout(int B, int H, int W, int p1, int p2) {
    Layout L;
    const size_t hw = (size_t)H * W;
    size_t o = al((size_t)B * sizeof(SimHeader));   // one header per image (only the first is used unless per_image)
    L.acc = o; o += al((size_t)B * N_ACC * sizeof(double));
    L.tickets = o; o += al((size_t)(6 * B) * sizeof(unsigned));   // prep [B], patch [3B], final [B], min / max [B] (k_sim_minmax_det)
    L.y = o; o += al((size_t)B * hw * 4);   // everything before y is reset by k_sim_init
    L.gy = o; o += al((size_t)B * 2 * hw * 4);
    L.m1 = o; o += al((size_t)4 * B * (size_t)(H - p1 + 1) * (W - p1 + 1) * 4);
    L.m2 = o; o += al((size_t)2 * 4 * B * (size_t)(H - p2 + 1) * (W - p2 + 1) * 4);
    L.Gy = o; o += al((size_t)B * hw * 4);
    L.Gg = o; o += al((size_t)B * 2 * hw * 4);
    const size_t tiles = (size_t)((W - (p1 < p2 ? p1 : p2) + 1 + TILE - 1) / TILE) * ((H - (p1 < p2 ? p1 : p2) + 1 + TILE - 1) / TILE);
    L.part_prep = o; o += al((size_t)B * PREP_BLOCKS_MAX * 5 * sizeof(double));
    L.part_patch = o; o += al((size_t)3 * B * tiles * sizeof(double));
    L.part_final = o; o += al((size_t)B * ((hw + TB - 1) / TB) * 2 * sizeof(double));
    L.part_mm = o; o += al((size_t)B * MM_BLOCKS_MAX * 2 * sizeof(float));
    L.total = o;
    return L;
}

}  // namespace

namespace {

__device__ __forceinline__ int reflect_idx(int i, int n) {   // torch's "reflect" padding (no edge repeat); needs n >= 3 for 2 pixels
    i = i < 0 ? -i : i;
    return i >= n ? 2 * (n - 1) - i : i;
}

// One axis of the 5-tap Gaussian pre-blur of GradientNormalizedCrossCorrelation2d(p, sigma > 0): reflect padding by 2, valid
// correlation.  ADJ = false: out[p] = sum_t k[t] in[reflect(p + t - 2)].  ADJ = true: the exact transpose (what autograd
// gives for the pad + conv pair): out[r] = sum over (p, t) with reflect(p + t - 2) = r of k[t] in[p] -- gathered, one thread
// per element, no atomics.
template <bool ADJ>
__global__ __launch_bounds__(TB) void k_blur5(const float* __restrict__ in, float* __restrict__ out, long long total, int H, int W, int axis,
                                              float k0, float k1, float k2, float k3, float k4) {
    const long long e = (long long)blockIdx.x * TB + threadIdx.x;
    if (e >= total) return;
    const int w = (int)(e % W), h = (int)((e / W) % H);
    const long long base = e - (axis ? w : (long long)h * W);    // element 0 of this row / column
    const int n = axis ? W : H, pos = axis ? w : h;
    const long long stride = axis ? 1 : W;
    const float k[5] = {k0, k1, k2, k3, k4};
    float acc = 0.f;
    if (!ADJ) {
#pragma unroll
        for (int t = 0; t < 5; ++t) acc = fmaf(k[t], in[base + (long long)reflect_idx(pos + t - 2, n) * stride], acc);
    } else {
        // sources p = m - t + 2 for every m that reflects onto pos: m = pos, m = -pos (pos = 1, 2), m = 2 (n - 1) - pos
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            const int d = t - 2;
            int pp = pos - d;                                     // m = pos
            if (pp >= 0 && pp < n) acc = fmaf(k[t], in[base + (long long)pp * stride], acc);
            pp = -pos - d;                                        // m = -pos < 0
            if (pos > 0 && pp >= 0 && pp < n && -pos >= -2) acc = fmaf(k[t], in[base + (long long)pp * stride], acc);
            pp = 2 * (n - 1) - pos - d;                           // m = 2 (n - 1) - pos > n - 1
            if (pos < n - 1 && pp >= 0 && pp < n && n - 1 - pos <= 2) acc = fmaf(k[t], in[base + (long long)pp * stride], acc);
        }
    }
    out[e] = acc;
}

// ---------------------------------------------------------------------------------------------
// Equalize: differentiable (soft-histogram) histogram equalisation of images in [0, 1]
// (/root/reference/src/xvr/utils/preprocess.py:34-66), forward and backward, per image, deterministic (fixed-order sums).
//   w_ib = exp(-(x_i - beta_b)^2 / (2 tau^2)), beta_b = b / (K - 1);   h_b = sum_i w_ib;   H = h / (sum h + eps);
//   cdf = cumsum(H);   cn_b = (cdf_b - cdf_0) / (1 - cdf_0 + eps);   y_i = sum_b w_ib cn_b / (sum_b w_ib + eps)
// The reference materialises the [pixels x bins] weight matrix (64 MB per 256^2 image, several times over under autograd);
// here every weight is recomputed where it is needed.  Weights beyond |x - beta| > EQ_CUT tau underflow to less than 1e-19
// of the row's largest and are skipped (they cannot change an fp32 sum).
// ---------------------------------------------------------------------------------------------
constexpr int EQ_MAX_BINS = 1024;
constexpr float EQ_CUT = 9.5f;   // exp(-9.5^2 / 2) = 2.5e-20

struct EqState {   // per image, in the workspace
    float T, D, dh_scale_pad0, pad1;
};

// h[b] (or, GRAD, dcn[b] = sum_i g_i w_ib / S_i), as EQ_CHUNKS partial sums per (bin, image): one block per (bin, chunk of the
// pixels, image), pixels in a fixed order; the consumer (k_eq_cdf / k_eq_cdf_bwd) adds a bin's partials in chunk order.
// (Round 3 gave a bin ONE block that walked all the pixels: 256 blocks of four wavefronts each, 1024 dependent trips at
// 512^2 -- 163 / 195 us per call, most of an iteration with `equalize`.)
constexpr int EQ_CHUNKS = 32;
template <bool GRAD>
__global__ __launch_bounds__(TB) void k_eq_bins(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ S,
                                                int n, int K, float inv2tau2, float cut, float* __restrict__ part, float gscale) {
    const int b = blockIdx.x, ch = blockIdx.y, img = blockIdx.z;
    const float beta = (float)b / (float)(K - 1);
    const int per = (n + EQ_CHUNKS - 1) / EQ_CHUNKS, lo = ch * per, hi = min(lo + per, n);
    const float* xi = x + (size_t)img * n;
    float acc = 0.f;
    for (int i = lo + (int)threadIdx.x; i < hi; i += TB) {
        const float d = xi[i] - beta;
        const float w = fabsf(d) <= cut ? __expf(-d * d * inv2tau2) : 0.f;
        acc += GRAD ? (g[(size_t)img * n + i] * gscale) * w / S[(size_t)img * n + i] : w;
    }
    __shared__ float red[TB];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = TB / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[((size_t)img * K + b) * EQ_CHUNKS + ch] = red[0];
}

// a bin's value from its partial sums, added in chunk order
__device__ __forceinline__ float eq_bin_sum(const float* __restrict__ part, int img, int K, int b) {
    const float* p = part + ((size_t)img * K + b) * EQ_CHUNKS;
    float s = 0.f;
#pragma unroll 8
    for (int c = 0; c < EQ_CHUNKS; ++c) s += p[c];
    return s;
}

// histogram -> normalised cdf (forward) ; d cn -> d h (backward).  One block of TB threads per image; thread t owns the
// EQ_PER = 4 consecutive bins 4t .. 4t + 3 (K <= 1024) and the block combines the threads' sums with a fixed tree in LDS
// (deterministic; round 3 scanned the bins serially in one thread: 256 dependent global round trips, two thirds of the call).
constexpr int EQ_PER = EQ_MAX_BINS / TB;
static_assert(EQ_PER * TB == EQ_MAX_BINS, "bins per thread");

// inclusive prefix sums of one value per thread over the block (REVERSE: suffix sums), and the block's total, through LDS
template <bool REVERSE>
__device__ __forceinline__ float block_scan(float v, float* sh, float& total) {
    const int t = REVERSE ? TB - 1 - (int)threadIdx.x : (int)threadIdx.x;
    sh[t] = v;
    __syncthreads();
    for (int o = 1; o < TB; o <<= 1) {                 // Hillis-Steele: a fixed order of additions
        const float add = t >= o ? sh[t - o] : 0.f;
        __syncthreads();
        sh[t] += add;
        __syncthreads();
    }
    const float r = sh[t];
    total = sh[TB - 1];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(TB) void k_eq_cdf(const float* __restrict__ part, float* __restrict__ h, int K, float eps, float* __restrict__ cn,
                                               float* __restrict__ cdf, EqState* st) {
    __shared__ float sh[TB];
    const int img = blockIdx.x, b0 = threadIdx.x * EQ_PER;
    float v[EQ_PER], mine = 0.f;
#pragma unroll
    for (int e = 0; e < EQ_PER; ++e) {
        v[e] = b0 + e < K ? eq_bin_sum(part, img, K, b0 + e) : 0.f;
        if (b0 + e < K) h[(size_t)img * K + b0 + e] = v[e];      // the histogram itself: the backward reads it
        mine += v[e];
    }
    float T;
    block_scan<false>(mine, sh, T);
    const float it = 1.f / (T + eps);
    float run = 0.f, loc[EQ_PER];
#pragma unroll
    for (int e = 0; e < EQ_PER; ++e) { run += v[e] * it; loc[e] = run; }      // normalised histogram, prefix sums within the thread
    float all;
    const float incl = block_scan<false>(run, sh, all);
    const float before = incl - run;
    if (threadIdx.x == 0) sh[0] = loc[0];                                      // cdf_0
    __syncthreads();
    const float c0 = sh[0], D = 1.f - c0 + eps;
#pragma unroll
    for (int e = 0; e < EQ_PER; ++e) {
        if (b0 + e < K) {
            const float c = before + loc[e];
            cdf[(size_t)img * K + b0 + e] = c;
            cn[(size_t)img * K + b0 + e] = (c - c0) / D;
        }
    }
    if (threadIdx.x == 0) { st[img].T = T; st[img].D = D; }
}

__global__ __launch_bounds__(TB) void k_eq_cdf_bwd(const float* __restrict__ h, const float* __restrict__ cdf, const float* __restrict__ dcn_part, int K, float eps,
                                                   const EqState* st, float* __restrict__ dh) {
    __shared__ float sh[TB];
    const int img = blockIdx.x, b0 = threadIdx.x * EQ_PER;
    const float T = st[img].T, D = st[img].D;
    const float* hb = h + (size_t)img * K;
    const float* cb = cdf + (size_t)img * K;
    float* out = dh + (size_t)img * K;
    // d cdf_b = d cn_b / D, and through cdf_0's second role: d cdf_0 += sum_b d cn_b (cdf_b - 1 - eps) / D^2
    float g[EQ_PER], hv[EQ_PER], ex = 0.f;
#pragma unroll
    for (int e = 0; e < EQ_PER; ++e) {
        const bool in = b0 + e < K;
        g[e] = in ? eq_bin_sum(dcn_part, img, K, b0 + e) : 0.f;
        hv[e] = in ? hb[b0 + e] : 0.f;
        if (in) ex += g[e] * (cb[b0 + e] - 1.f - eps) / (D * D);
    }
    float extra;
    block_scan<false>(ex, sh, extra);
    // d H_b = sum_{b' >= b} d cdf_b' (suffix sums);  d h_b = d H_b / (T + eps) - (sum_b' d H_b' h_b') / (T + eps)^2
    float suf[EQ_PER], run = 0.f;
#pragma unroll
    for (int e = EQ_PER - 1; e >= 0; --e) { run += g[e] / D + (b0 + e == 0 ? extra : 0.f); suf[e] = run; }
    float all;
    const float incl = block_scan<true>(run, sh, all);
    const float after = incl - run;                  // the bins of the threads behind this one
    float dotp = 0.f;
#pragma unroll
    for (int e = 0; e < EQ_PER; ++e) { suf[e] += after; dotp += suf[e] * hv[e]; }
    float dot;
    block_scan<false>(dotp, sh, dot);
    const float it = 1.f / (T + eps);
#pragma unroll
    for (int e = 0; e < EQ_PER; ++e)
        if (b0 + e < K) out[b0 + e] = suf[e] * it - dot * it * it;
}

// per pixel: forward y_i (and S_i kept for the backward); backward g_x,i
// (forward: yn (nullable) = (y - out_mean) * out_inv_std, the Normalize that follows Equalize in XrayTransforms;
//  backward: g is the gradient w.r.t. that normalised output, i.e. scaled by out_inv_std on the way in)
template <bool BWD>
__global__ __launch_bounds__(TB) void k_eq_pixels(const float* __restrict__ x, int n, int K, float tau, float eps, float cut,
                                                  const float* __restrict__ cn, float* __restrict__ y, float* __restrict__ S,
                                                  const float* __restrict__ g, const float* __restrict__ dh, float* __restrict__ gx,
                                                  float* __restrict__ yn, float out_mean, float out_inv_std) {
    __shared__ float c_s[EQ_MAX_BINS], d_s[EQ_MAX_BINS];
    const int img = blockIdx.y;
    for (int b = threadIdx.x; b < K; b += TB) {
        c_s[b] = cn[(size_t)img * K + b];
        if (BWD) d_s[b] = dh[(size_t)img * K + b];
    }
    __syncthreads();
    const int i = blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    const size_t e = (size_t)img * n + i;
    const float xi = x[e];
    const float inv2 = 1.f / (2.f * tau * tau), invt2 = 1.f / (tau * tau), fk = (float)(K - 1);
    const int blo = max((int)ceilf((xi - cut) * fk), 0), bhi = min((int)floorf((xi + cut) * fk), K - 1);
    if (!BWD) {
        float Ssum = 0.f, N = 0.f;
        for (int b = blo; b <= bhi; ++b) {
            const float d = xi - (float)b / fk;
            const float w = __expf(-d * d * inv2);
            Ssum += w;
            N = fmaf(w, c_s[b], N);
        }
        Ssum += eps;
        const float yi = N / Ssum;
        y[e] = yi;
        S[e] = Ssum;
        if (yn) yn[e] = (yi - out_mean) * out_inv_std;
    } else {
        const float Si = S[e], yi = y[e], gi = g[e] * out_inv_std;
        float acc = 0.f;
        for (int b = blo; b <= bhi; ++b) {
            const float d = xi - (float)b / fk;
            const float w = __expf(-d * d * inv2);
            // d w / d x = -w d / tau^2;  direct term g_i (cn_b - y_i) / S_i, histogram term d h_b
            acc = fmaf(-w * d * invt2, fmaf(gi, (c_s[b] - yi) / Si, d_s[b]), acc);
        }
        gx[e] = acc;
    }
}


// ---------------------------------------------------------------------------------------------
// Dice of two BOOLEAN label maps (the in-tree DiceMetric of /root/reference/src/xvr/model/loss.py:5-40 on the masks
// render_samples returns): per (pose, channel) the counts |a & b|, |a|, |b| in integers, dice = 2 |a & b| / (|a| + |b|) in
// float32 -- the reference's float sums of 0 / 1 are exact, so the value is bit-identical; 0 / 0 = NaN as there.  One block per
// (channel, pose), 16 mask bytes per load.  Replaces two bool -> float casts, a product and three reductions over
// [B][C][n] floats (0.6 ms per training step at C5's size).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TB) void k_dice_bool(const unsigned char* __restrict__ a, const unsigned char* __restrict__ b, int n,
                                                  float* __restrict__ dice) {
    const size_t base = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (size_t)n;
    const unsigned char* pa = a + base;
    const unsigned char* pb = b + base;
    int ci = 0, ca = 0, cb = 0;
    const bool vec = ((reinterpret_cast<uintptr_t>(pa) | reinterpret_cast<uintptr_t>(pb)) & 15u) == 0;
    const int n16 = vec ? n >> 4 : 0;
    for (int i = threadIdx.x; i < n16; i += TB) {
        const uint4 x = reinterpret_cast<const uint4*>(pa)[i], y = reinterpret_cast<const uint4*>(pb)[i];
        const unsigned xs[4] = {x.x, x.y, x.z, x.w}, ys[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {   // (bytes are 0 / 1; any other non-zero byte counts as 1)
            const unsigned u = xs[k] | (xs[k] >> 1), v = ys[k] | (ys[k] >> 1);
            const unsigned u2 = u | (u >> 2), v2 = v | (v >> 2);
            const unsigned ub = (u2 | (u2 >> 4)) & 0x01010101u, vb = (v2 | (v2 >> 4)) & 0x01010101u;
            ci += __popc(ub & vb); ca += __popc(ub); cb += __popc(vb);
        }
    }
    for (int i = (n16 << 4) + threadIdx.x; i < n; i += TB) {
        const int u = pa[i] != 0, v = pb[i] != 0;
        ci += u & v; ca += u; cb += v;
    }
    __shared__ int red[3][TB / 64];
    int vals[3] = {ci, ca, cb};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        int v = vals[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int t[3] = {0, 0, 0};
        for (int k = 0; k < 3; ++k)
            for (int w = 0; w < TB / 64; ++w) t[k] += red[k][w];
        dice[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = (2.0f * (float)t[0]) / ((float)t[1] + (float)t[2]);
    }
}

// ---------------------------------------------------------------------------------------------
// XrayTransforms without Equalize / Resize -- Standardize (min / max over the whole tensor, or per image) then Normalize --
// as a standalone pair (/root/reference/src/xvr/utils/preprocess.py:5-31; the trainer applies it to both renders of every
// step, trainer.py:207,216, and autograd then runs ~35 launches of min / max backward bookkeeping over the batch: 0.7 ms at
// 116 images).  Forward: k_sim_minmax, then y = ((x - lo) / (hi - lo + eps) - mean) / std with torch's own operation order
// (bit-identical) and the counts of the pixels that attain lo / hi.  Backward: the two sums  sum g (1 - xs),  sum g xs
// (xs = the standardised pixel) in a fixed order, then  gx = a g  plus the min's / max's share at the pixels that attain
// them (torch's rule for a full-reduction min / max: spread evenly).  state = one TfState per group (1 or B).
// ---------------------------------------------------------------------------------------------
struct TfState {
    SimHeader h;
    unsigned ticket, pad;
};

__global__ __launch_bounds__(TB) void k_tf_init(TfState* st, int groups) {
    const int i = blockIdx.x * TB + threadIdx.x;
    if (i < groups) {
        st[i].h.enc_min = 0xffffffffu; st[i].h.enc_max = 0u; st[i].h.cnt_min = 0; st[i].h.cnt_max = 0;
        st[i].h.smin = 0.0; st[i].h.smax = 0.0; st[i].ticket = 0u; st[i].pad = 0u;
    }
}

__global__ __launch_bounds__(TB) void k_tf_minmax(const float* __restrict__ m, long long n, TfState* st) {
    __shared__ float plo[TB / 64], phi[TB / 64];
    float lo = INFINITY, hi = -INFINITY;
    m += (size_t)blockIdx.y * n;
    st += blockIdx.y;
    const bool vec = (reinterpret_cast<uintptr_t>(m) & 15u) == 0;   // (an image of a pixel count that is not a multiple of 4)
    for (long long i = ((long long)blockIdx.x * TB + threadIdx.x) * 4; i < n; i += (long long)gridDim.x * TB * 4) {
        if (i + 3 < n && vec) {
            const float4 v = *reinterpret_cast<const float4*>(m + i);
            lo = fminf(fminf(lo, v.x), fminf(fminf(v.y, v.z), v.w));
            hi = fmaxf(fmaxf(hi, v.x), fmaxf(fmaxf(v.y, v.z), v.w));
        } else {
            for (long long j = i; j < n && j < i + 4; ++j) { lo = fminf(lo, m[j]); hi = fmaxf(hi, m[j]); }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o));
        hi = fmaxf(hi, __shfl_xor(hi, o));
    }
    if ((threadIdx.x & 63) == 0) { plo[threadIdx.x >> 6] = lo; phi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < TB / 64; ++w) { lo = fminf(lo, plo[w]); hi = fmaxf(hi, phi[w]); }
        atomicMin(&st->h.enc_min, enc(lo));
        atomicMax(&st->h.enc_max, enc(hi));
    }
}

// BWD = false: y and the counts;  true: gx = a g + the min's / max's share (st->h.smin / smax hold the two sums)
template <bool BWD>
__global__ __launch_bounds__(TB) void k_tf_apply(const float* __restrict__ x, const float* __restrict__ g, long long n, float mean, float std_,
                                                 float eps, TfState* st, float* __restrict__ out) {
    x += (size_t)blockIdx.y * n;
    out += (size_t)blockIdx.y * n;
    if (BWD) g += (size_t)blockIdx.y * n;
    st += blockIdx.y;
    const float mn = dec(st->h.enc_min), mx = dec(st->h.enc_max);
    const float r = (mx - mn) + eps;
    float a = 0.f, gmin = 0.f, gmax = 0.f;
    if (BWD) {
        const double ad = 1.0 / ((double)r * std_);
        a = (float)ad;
        gmin = (float)(-ad * st->h.smin / (double)max(st->h.cnt_min, 1));
        gmax = (float)(-ad * st->h.smax / (double)max(st->h.cnt_max, 1));
    }
    const float inv_std = 1.f / std_;
    int cmin = 0, cmax = 0;
    for (long long i = (long long)blockIdx.x * TB + threadIdx.x; i < n; i += (long long)gridDim.x * TB) {
        const float v = x[i];
        if (BWD) {
            float t = a * g[i];
            if (v == mn) t += gmin;
            if (v == mx) t += gmax;
            out[i] = t;
        } else {
            out[i] = (((v - mn) / r) - mean) * inv_std;   // (torch divides by a Python scalar as a multiplication by its float reciprocal)
            cmin += v == mn;
            cmax += v == mx;
        }
    }
    if (!BWD) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { cmin += __shfl_xor(cmin, o); cmax += __shfl_xor(cmax, o); }
        if ((threadIdx.x & 63) == 0) {   // integer atomics: exact in any order
            if (cmin) atomicAdd(&st->h.cnt_min, cmin);
            if (cmax) atomicAdd(&st->h.cnt_max, cmax);
        }
    }
}

__global__ __launch_bounds__(TB) void k_tf_bwd_reduce(const float* __restrict__ x, const float* __restrict__ g, long long n, float eps,
                                                      TfState* st, double* partial) {
    x += (size_t)blockIdx.y * n;
    g += (size_t)blockIdx.y * n;
    st += blockIdx.y;
    const float mn = dec(st->h.enc_min), mx = dec(st->h.enc_max);
    const float r = (mx - mn) + eps;
    double s2[2] = {0.0, 0.0};
    for (long long i = (long long)blockIdx.x * TB + threadIdx.x; i < n; i += (long long)gridDim.x * TB) {
        const float xs = (x[i] - mn) / r;
        const double gg = (double)g[i];
        s2[0] += gg * (1.0 - (double)xs);
        s2[1] += gg * (double)xs;
    }
    grid_add_det<2>(s2, partial + (size_t)blockIdx.y * gridDim.x * 2, blockIdx.x, gridDim.x, &st->ticket, &st->h.smin);
}

}  // namespace

extern "C" {

size_t xvr_sim_equalize_workspace_bytes(int B, int n_bins) {
    if (B <= 0 || n_bins < 2 || n_bins > EQ_MAX_BINS) return 0;
    return (size_t)B * ((size_t)n_bins * (4 + EQ_CHUNKS) * sizeof(float) + sizeof(EqState)) + 256;
}

// workspace layout: [h][cdf][cn][dh] (B x K floats each) [partial sums of h, then of dcn: B x K x EQ_CHUNKS] [EqState x B]
int xvr_sim_equalize_forward(const float* x, int B, int n, int n_bins, float tau, float eps, float out_mean, float out_std, float* y,
                             float* S, float* y_out, void* workspace, size_t workspace_bytes, void* stream_) {
    if (!x || !y || !S || !workspace) return sim_fail(XVR_DRR_E_ARG, "null pointer argument");
    if (B <= 0 || n <= 0 || n_bins < 2 || n_bins > EQ_MAX_BINS || !(tau > 0.f)) return sim_fail(XVR_DRR_E_ARG, "bad size / bins / tau");
    if (y_out && !(out_std != 0.f)) return sim_fail(XVR_DRR_E_ARG, "out_std must not be zero");
    if (workspace_bytes < xvr_sim_equalize_workspace_bytes(B, n_bins)) return sim_fail(XVR_DRR_E_ARG, "workspace too small");
    hipStream_t stream = (hipStream_t)stream_;
    float* h = static_cast<float*>(workspace);
    float* cdf = h + (size_t)B * n_bins;
    float* cn = cdf + (size_t)B * n_bins;
    float* part = h + (size_t)B * n_bins * 4;
    EqState* st = reinterpret_cast<EqState*>(h + (size_t)B * n_bins * (4 + EQ_CHUNKS));
    const float cut = EQ_CUT * tau;
    hipLaunchKernelGGL(k_eq_bins<false>, dim3(n_bins, EQ_CHUNKS, B), dim3(TB), 0, stream, x, (const float*)nullptr, (const float*)nullptr, n, n_bins,
                       1.f / (2.f * tau * tau), cut, part, 1.f);
    hipLaunchKernelGGL(k_eq_cdf, dim3(B), dim3(TB), 0, stream, (const float*)part, h, n_bins, eps, cn, cdf, st);
    hipLaunchKernelGGL(k_eq_pixels<false>, dim3((n + TB - 1) / TB, B), dim3(TB), 0, stream, x, n, n_bins, tau, eps, cut, (const float*)cn, y, S,
                       (const float*)nullptr, (const float*)nullptr, (float*)nullptr, y_out, out_mean, y_out ? 1.f / out_std : 1.f);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? XVR_DRR_OK : sim_fail(XVR_DRR_E_LAUNCH, hipGetErrorString(e));
}

// needs the workspace exactly as the forward left it (h, cdf, cn, state), and the forward's y and S
int xvr_sim_equalize_backward(const float* x, const float* y, const float* S, const float* grad_out, int B, int n, int n_bins, float tau,
                              float eps, float out_std, float* grad_x, void* workspace, size_t workspace_bytes, void* stream_) {
    if (!x || !y || !S || !grad_out || !grad_x || !workspace) return sim_fail(XVR_DRR_E_ARG, "null pointer argument");
    if (B <= 0 || n <= 0 || n_bins < 2 || n_bins > EQ_MAX_BINS || !(tau > 0.f)) return sim_fail(XVR_DRR_E_ARG, "bad size / bins / tau");
    if (!(out_std != 0.f)) return sim_fail(XVR_DRR_E_ARG, "out_std must not be zero (1 for a gradient w.r.t. y itself)");
    if (workspace_bytes < xvr_sim_equalize_workspace_bytes(B, n_bins)) return sim_fail(XVR_DRR_E_ARG, "workspace too small");
    hipStream_t stream = (hipStream_t)stream_;
    float* h = static_cast<float*>(workspace);
    float* cdf = h + (size_t)B * n_bins;
    float* cn = cdf + (size_t)B * n_bins;
    float* dh = cn + (size_t)B * n_bins;
    float* part = h + (size_t)B * n_bins * 4;      // (the forward's partials are spent: the histogram is in h)
    EqState* st = reinterpret_cast<EqState*>(h + (size_t)B * n_bins * (4 + EQ_CHUNKS));
    const float cut = EQ_CUT * tau, gs = 1.f / out_std;
    hipLaunchKernelGGL(k_eq_bins<true>, dim3(n_bins, EQ_CHUNKS, B), dim3(TB), 0, stream, x, grad_out, S, n, n_bins, 1.f / (2.f * tau * tau), cut, part, gs);
    hipLaunchKernelGGL(k_eq_cdf_bwd, dim3(B), dim3(TB), 0, stream, (const float*)h, (const float*)cdf, (const float*)part, n_bins, eps,
                       (const EqState*)st, dh);
    hipLaunchKernelGGL(k_eq_pixels<true>, dim3((n + TB - 1) / TB, B), dim3(TB), 0, stream, x, n, n_bins, tau, eps, cut, (const float*)cn,
                       const_cast<float*>(y), const_cast<float*>(S), grad_out, (const float*)dh, grad_x, (float*)nullptr, 0.f, gs);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? XVR_DRR_OK : sim_fail(XVR_DRR_E_LAUNCH, hipGetErrorString(e));
}

int xvr_sim_gaussian_blur5(const float* in, float* out, float* scratch, int B, int H, int W, float sigma, int adjoint, void* stream_) {
    if (!in || !out || !scratch) return sim_fail(XVR_DRR_E_ARG, "null pointer argument");
    if (B <= 0 || H < 3 || W < 3) return sim_fail(XVR_DRR_E_ARG, "reflect padding by 2 needs images of at least 3 x 3");
    if (!(sigma > 0.f)) return sim_fail(XVR_DRR_E_ARG, "sigma must be positive");
    float k[5], sum = 0.f;
    for (int t = 0; t < 5; ++t) { k[t] = expf(-0.5f * ((float)(t - 2) / sigma) * ((float)(t - 2) / sigma)); sum += k[t]; }
    for (int t = 0; t < 5; ++t) k[t] /= sum;
    const long long total = (long long)B * H * W;
    const unsigned blocks = (unsigned)((total + TB - 1) / TB);
    hipStream_t stream = (hipStream_t)stream_;
    // forward: along W, then along H (the order of the reference's two conv2d calls); the adjoint runs them in reverse
    if (!adjoint) {
        hipLaunchKernelGGL(k_blur5<false>, dim3(blocks), dim3(TB), 0, stream, in, scratch, total, H, W, 1, k[0], k[1], k[2], k[3], k[4]);
        hipLaunchKernelGGL(k_blur5<false>, dim3(blocks), dim3(TB), 0, stream, (const float*)scratch, out, total, H, W, 0, k[0], k[1], k[2], k[3], k[4]);
    } else {
        hipLaunchKernelGGL(k_blur5<true>, dim3(blocks), dim3(TB), 0, stream, in, scratch, total, H, W, 0, k[0], k[1], k[2], k[3], k[4]);
        hipLaunchKernelGGL(k_blur5<true>, dim3(blocks), dim3(TB), 0, stream, (const float*)scratch, out, total, H, W, 1, k[0], k[1], k[2], k[3], k[4]);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? XVR_DRR_OK : sim_fail(XVR_DRR_E_LAUNCH, hipGetErrorString(e));
}

size_t xvr_sim_workspace_bytes(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    // sized for the smallest legal patches (largest maps)
    return layout(B, H, W, 1, 1).total;
}

static int ncc_launch(const float* fixed, const float* fixed_sobel, const float* moving, int B, int H, int W,
                      const xvr_sim_spec* sp, float* loss, float* grad_moving, void* workspace,
                      size_t workspace_bytes, void* stream_, const RegTail* tail, bool armed = false) {
    if (!fixed || !fixed_sobel || !moving || !sp || !loss || !workspace) return sim_fail(XVR_DRR_E_ARG, "null pointer argument");
    if (B <= 0 || H <= 0 || W <= 0) return sim_fail(XVR_DRR_E_ARG, "B, H, W must be positive");
    const int p1 = sp->mncc_patch, p2 = sp->gncc_patch;
    // (a 1 x 1 patch has no variance: its NCC is 0 / eps, i.e. rounding noise over eps -- left to the caller)
    if (p1 < 2 || p2 < 2 || p1 > MAXP || p2 > MAXP) return sim_fail(XVR_DRR_E_UNSUPPORTED, "patch size must be in [2, 15]");
    if (H < p1 || W < p1 || H < p2 || W < p2) return sim_fail(XVR_DRR_E_ARG, "image smaller than a patch");
    const Layout L = layout(B, H, W, p1, p2);
    if (workspace_bytes < L.total) return sim_fail(XVR_DRR_E_ARG, "workspace too small");
    hipStream_t stream = (hipStream_t)stream_;
    char* ws = static_cast<char*>(workspace);
    SimHeader* hd = reinterpret_cast<SimHeader*>(ws);
    double* acc = reinterpret_cast<double*>(ws + L.acc);
    float* y = reinterpret_cast<float*>(ws + L.y);
    float* gyb = reinterpret_cast<float*>(ws + L.gy);
    float* m1 = reinterpret_cast<float*>(ws + L.m1);
    float* m2 = reinterpret_cast<float*>(ws + L.m2);
    float* Gy = reinterpret_cast<float*>(ws + L.Gy);
    float* Gg = reinterpret_cast<float*>(ws + L.Gg);
    const int hw = H * W;
    const long long n = (long long)B * hw;

    const int nwords = (int)(L.y / 4);   // header + accumulators
    const bool skip_init = armed && !sp->pre_transformed;   // (the caller vouches for zero tickets; the ticketed min / max writes the header)
    if (!skip_init)
        hipLaunchKernelGGL(k_sim_init, dim3((nwords + TB - 1) / TB), dim3(TB), 0, stream, reinterpret_cast<unsigned*>(ws), nwords,
                           (int)(B * sizeof(SimHeader) / 4));
    const bool per_image = sp->per_image != 0;
    const long long n_mm = per_image ? (long long)hw : n;   // elements one min/max group spans
    const unsigned groups = per_image ? (unsigned)B : 1u;
    const long long want = (n_mm + (long long)TB * 16 - 1) / ((long long)TB * 16);
    const unsigned rb = (unsigned)(want < 1 ? 1 : (want > 256 ? 256 : want));
    const bool pre = sp->pre_transformed != 0;   // no Standardize: no min/max, no gradient through them
    unsigned* tickets = reinterpret_cast<unsigned*>(ws + L.tickets);
    if (skip_init) hipLaunchKernelGGL(k_sim_minmax_det, dim3(rb, groups), dim3(TB), 0, stream, moving, n_mm, hd,
                                      reinterpret_cast<float*>(ws + L.part_mm), tickets + 5 * B);
    else if (!pre) hipLaunchKernelGGL(k_sim_minmax, dim3(rb, groups), dim3(TB), 0, stream, moving, n_mm, hd);
    // (a batch fills the chip with a quarter of the blocks per image: four pixels per thread, a quarter of the deterministic
    //  reductions -- 0.23 -> ms below at the training loss's 116 images; one image keeps every block it can get)
    unsigned pb = (unsigned)((hw + TB - 1) / TB < (int)PREP_BLOCKS_MAX ? (hw + TB - 1) / TB : PREP_BLOCKS_MAX);
    if (B >= 16 && pb >= 16) pb = (pb + 3) / 4;
    hipLaunchKernelGGL(k_sim_prep, dim3(pb, B), dim3(TB), 0, stream, moving, fixed, H, W, hd, *sp, y, gyb, acc,
                       reinterpret_cast<double*>(ws + L.part_prep), tickets);
    const size_t np2 = (size_t)B * (H - p2 + 1) * (W - p2 + 1);
    const double n1 = (double)(H - p1 + 1) * (W - p1 + 1), n2 = (double)(H - p2 + 1) * (W - p2 + 1);
    const float sc1 = (float)(0.5 * sp->beta / (n1 * p1 * p1));
    const float sc2 = (float)(0.5 * (1.0 - sp->beta) / (n2 * p2 * p2));
    PatchJobs jobs;
    jobs.j[0] = {fixed, y, 1, 0, p1, m1, 5, sc1, Gy, 1, 0};
    jobs.j[1] = {fixed_sobel, gyb, 2, 0, p2, m2, 6, sc2, Gg, 2, 0};
    jobs.j[2] = {fixed_sobel, gyb, 2, 1, p2, m2 + 4 * np2, 7, sc2, Gg, 2, 1};
    // a term with zero weight is not computed: beta = 1 drops the two gradient-NCC jobs, beta = 0 the local mNCC
    int njobs = 3, pmin = p1 < p2 ? p1 : p2;
    if (sp->beta >= 1.f) { njobs = 1; pmin = p1; }
    else if (sp->beta <= 0.f) { jobs.j[0] = jobs.j[1]; jobs.j[1] = jobs.j[2]; njobs = 2; pmin = p2; }
    auto tiles = [&](int hh, int ww) { return dim3((ww + TILE - 1) / TILE, (hh + TILE - 1) / TILE, njobs * B); };
    const int tpb = B >= 16 ? 4 : 1;
    dim3 pgrid = tiles(H - pmin + 1, W - pmin + 1);
    pgrid.y = (pgrid.y + tpb - 1) / tpb;
    hipLaunchKernelGGL(k_sim_patch, pgrid, dim3(TB), 0, stream, jobs, B, H, W, sp->ncc_eps, acc,
                       reinterpret_cast<double*>(ws + L.part_patch), tickets + B, tpb);
    hipLaunchKernelGGL(k_sim_patch_grad, tiles(H, W), dim3(TB), 0, stream, jobs, B, H, W);
    hipLaunchKernelGGL(k_sim_final, dim3((hw + TB - 1) / TB, B), dim3(TB), 0, stream, moving, fixed, y, Gy, Gg, H, W, hd, acc,
                       *sp, grad_moving, reinterpret_cast<double*>(ws + L.part_final), tickets + 4 * B);
    const long long want2 = (n_mm + (long long)TB * 4 - 1) / ((long long)TB * 4);
    const unsigned gb = (unsigned)(want2 < 1 ? 1 : (want2 > 1024 ? 1024 : want2));
    if (tail) {   // the registration step: the min / max terms, jacobian -> camera, the optimiser step and the next camera in one launch
        const bool batch = (size_t)B * H * W >= ((size_t)1 << 20);   // (k_jac_to_cam's block geometry: the same partial sums)
        const unsigned per_block = batch ? 4 * TB : TB;
        const unsigned nblk = (unsigned)(((size_t)hw + per_block - 1) / per_block);
        if (batch) hipLaunchKernelGGL(k_sim_reg_tail<4>, dim3(nblk, (unsigned)B), dim3(TB), 0, stream, moving, (const SimHeader*)hd, *sp,
                                      (const float*)grad_moving, (const double*)acc, B, H, W, loss, *tail);
        else hipLaunchKernelGGL(k_sim_reg_tail<1>, dim3(nblk, (unsigned)B), dim3(TB), 0, stream, moving, (const SimHeader*)hd, *sp,
                                (const float*)grad_moving, (const double*)acc, B, H, W, loss, *tail);
    }
    else if (grad_moving && !pre) hipLaunchKernelGGL(k_sim_minmax_grad, dim3(gb, groups), dim3(TB), 0, stream, moving, n_mm, hd, *sp, grad_moving, acc, B, H, W, loss);
    else hipLaunchKernelGGL(k_sim_loss, dim3((B + 63) / 64), dim3(64), 0, stream, acc, B, H, W, *sp, loss);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return sim_fail(XVR_DRR_E_LAUNCH, hipGetErrorString(e));
    return XVR_DRR_OK;
}

int xvr_sim_ncc_forward_backward(const float* fixed, const float* fixed_sobel, const float* moving, int B, int H, int W,
                                 const xvr_sim_spec* sp, float* loss, float* grad_moving, void* workspace,
                                 size_t workspace_bytes, void* stream_) {
    return ncc_launch(fixed, fixed_sobel, moving, B, H, W, sp, loss, grad_moving, workspace, workspace_bytes, stream_, nullptr);
}

int xvr_sim_ncc_registration_step(const float* fixed, const float* fixed_sobel, const float* moving, int B, int H, int W,
                                  const xvr_sim_spec* sp, float* loss, float* grad_scratch, void* workspace, size_t workspace_bytes,
                                  const float* jac, float* cam, void* j2c_workspace, size_t j2c_workspace_bytes,
                                  float* rot, float* xyz, const xvr_pose_opt_spec* ospec, const float* G, const float* c,
                                  xvr_pose_opt_state* state, float* history, int workspace_armed, void* stream_) {
    if (!grad_scratch || !jac || !cam || !j2c_workspace || !rot || !xyz || !ospec || !G || !c || !state)
        return sim_fail(XVR_DRR_E_ARG, "null pointer argument");
    if (sp && sp->pre_transformed) return sim_fail(XVR_DRR_E_UNSUPPORTED, "registration step: the similarity of raw renders (pre_transformed = 0)");
    if (B <= 0 || H <= 0 || W <= 0) return sim_fail(XVR_DRR_E_ARG, "B, H, W must be positive");
    if (j2c_workspace_bytes < xvr_drr_jac_to_camera_workspace_bytes(B, H, W)) return sim_fail(XVR_DRR_E_ARG, "jacobian -> camera workspace too small");
    if (reinterpret_cast<uintptr_t>(jac) & 15u) return sim_fail(XVR_DRR_E_ARG, "jac must be 16-byte aligned");
    for (int i = 0; i < 3; ++i)
        if (ospec->axes[i] < 0 || ospec->axes[i] > 2) return sim_fail(XVR_DRR_E_ARG, "axes must be 0, 1 or 2");
    char* jw = static_cast<char*>(j2c_workspace);
    RegTail T;
    T.jac = jac; T.cam = cam;
    T.counter = reinterpret_cast<unsigned*>(jw);
    T.partial = reinterpret_cast<float*>(jw + al((size_t)B * sizeof(unsigned)));   // (xvr_drr_jac_to_camera_backward's own layout)
    T.rot = rot; T.xyz = xyz; T.G = G; T.c = c; T.state = state; T.history = history; T.osp = *ospec;
    return ncc_launch(fixed, fixed_sobel, moving, B, H, W, sp, loss, grad_scratch, workspace, workspace_bytes, stream_, &T, workspace_armed != 0);
}

int xvr_sim_dice_bool(const unsigned char* pred, const unsigned char* truth, int B, int C, int n, float* dice, void* stream_) {
    if (!pred || !truth || !dice) return sim_fail(XVR_DRR_E_ARG, "null pointer argument");
    if (B <= 0 || C <= 0 || n <= 0 || B > 65535) return sim_fail(XVR_DRR_E_ARG, "bad size");
    if (n > (1 << 24)) return sim_fail(XVR_DRR_E_UNSUPPORTED, "more than 2^24 pixels: the reference's float counts stop being exact");
    hipLaunchKernelGGL(k_dice_bool, dim3((unsigned)C, (unsigned)B), dim3(TB), 0, (hipStream_t)stream_, pred, truth, n, dice);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? XVR_DRR_OK : sim_fail(XVR_DRR_E_LAUNCH, hipGetErrorString(e));
}

constexpr unsigned TF_BLOCKS = 256;   // blocks per group of the transform's reductions

size_t xvr_sim_transform_state_bytes(int B) { return B > 0 ? (size_t)B * (sizeof(TfState) + TF_BLOCKS * 2 * sizeof(double)) : 0; }

static int tf_check(const float* x, int B, long long n, float std_, void* state) {
    if (!x || !state) return sim_fail(XVR_DRR_E_ARG, "null pointer argument");
    if (B <= 0 || n <= 0 || !(std_ != 0.f) || B > 65535) return sim_fail(XVR_DRR_E_ARG, "bad size or std");
    if (reinterpret_cast<uintptr_t>(x) & 15u) return sim_fail(XVR_DRR_E_ARG, "images must be 16-byte aligned");
    return XVR_DRR_OK;
}

int xvr_sim_transform_forward(const float* x, int B, long long n, int per_image, float mean, float std_, float eps, float* y, void* state,
                              void* stream_) {
    const int rc = tf_check(x, B, n, std_, state);
    if (rc != XVR_DRR_OK) return rc;
    if (!y) return sim_fail(XVR_DRR_E_ARG, "null pointer argument");
    hipStream_t stream = (hipStream_t)stream_;
    TfState* st = static_cast<TfState*>(state);
    const int groups = per_image ? B : 1;
    const long long gn = per_image ? n : (long long)B * n;
    const unsigned blocks = (unsigned)((gn + TB * 4 - 1) / (TB * 4) < TF_BLOCKS ? (gn + TB * 4 - 1) / (TB * 4) : TF_BLOCKS);
    const unsigned ablocks = per_image ? (blocks < 64 ? blocks : 64) : (unsigned)((gn + TB * 4 - 1) / (TB * 4) < 4096 ? (gn + TB * 4 - 1) / (TB * 4) : 4096);
    hipLaunchKernelGGL(k_tf_init, dim3((groups + TB - 1) / TB), dim3(TB), 0, stream, st, groups);
    hipLaunchKernelGGL(k_tf_minmax, dim3(blocks, groups), dim3(TB), 0, stream, x, gn, st);
    hipLaunchKernelGGL(k_tf_apply<false>, dim3(ablocks, groups), dim3(TB), 0, stream, x, (const float*)nullptr, gn, mean, std_, eps, st, y);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? XVR_DRR_OK : sim_fail(XVR_DRR_E_LAUNCH, hipGetErrorString(e));
}

int xvr_sim_transform_backward(const float* x, const float* grad_y, int B, long long n, int per_image, float mean, float std_, float eps,
                               float* grad_x, void* state, void* stream_) {
    const int rc = tf_check(x, B, n, std_, state);
    if (rc != XVR_DRR_OK) return rc;
    if (!grad_y || !grad_x) return sim_fail(XVR_DRR_E_ARG, "null pointer argument");
    hipStream_t stream = (hipStream_t)stream_;
    TfState* st = static_cast<TfState*>(state);
    const int groups = per_image ? B : 1;
    const long long gn = per_image ? n : (long long)B * n;
    double* partial = reinterpret_cast<double*>(st + B);
    const unsigned rblocks = (unsigned)((gn + TB * 8 - 1) / (TB * 8) < TF_BLOCKS ? (gn + TB * 8 - 1) / (TB * 8) : TF_BLOCKS);
    const unsigned ablocks = per_image ? (rblocks < 64 ? rblocks : 64) : (unsigned)((gn + TB * 4 - 1) / (TB * 4) < 4096 ? (gn + TB * 4 - 1) / (TB * 4) : 4096);
    hipLaunchKernelGGL(k_tf_bwd_reduce, dim3(rblocks, groups), dim3(TB), 0, stream, x, grad_y, gn, eps, st, partial);
    hipLaunchKernelGGL(k_tf_apply<true>, dim3(ablocks, groups), dim3(TB), 0, stream, x, grad_y, gn, mean, std_, eps, st, grad_x);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? XVR_DRR_OK : sim_fail(XVR_DRR_E_LAUNCH, hipGetErrorString(e));
}

}  // extern "C"
