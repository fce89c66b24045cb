// HU -> density for a whole CT in one streaming pass.
//
// xvr re-maps the full volume with a fresh random bone multiplier before the two renders of EVERY
// training step (/root/reference/src/xvr/model/trainer.py:124,196-197, diffdrr.data.transform_hu_to_density):
//     air (<= -800 HU) -> the minimum soft-tissue value; bone (> 350 HU) -> HU * multiplier; soft tissue
//     unchanged; then min-max normalised to [0, 1].
// In torch that is ~8 full passes over 512 MiB (3.5 ms on MI355X).  The normalisation constants follow
// analytically from per-class min / max / count, which do NOT depend on the multiplier: they are
// reduced once per CT (k_hu_stats) and every step is then a single read + write (k_hu_map).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "xvr_drr.h"

extern "C" void xvr_drr_set_last_error(const char* msg);

namespace {

constexpr int TB = 256;
constexpr float HU_AIR = -800.f, HU_BONE = 350.f;

// stats[0..2] = encoded min of air, soft, bone; stats[3..5] = encoded max; stats[6..8] = presence
__device__ __forceinline__ unsigned enc(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float dec(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

__global__ __launch_bounds__(TB) void k_hu_stats(const float* __restrict__ hu, long long n, unsigned* st) {
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    unsigned cnt[3] = {0, 0, 0};
    const long long n4 = n >> 2;
    const long long stride = (long long)gridDim.x * TB;
    for (long long i = (long long)blockIdx.x * TB + threadIdx.x; i < n4; i += stride) {
        const float4 v4 = reinterpret_cast<const float4*>(hu)[i];
        const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = v[k] <= HU_AIR ? 0 : (v[k] > HU_BONE ? 2 : 1);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                mn[q] = c == q ? fminf(mn[q], v[k]) : mn[q];
                mx[q] = c == q ? fmaxf(mx[q], v[k]) : mx[q];
                cnt[q] += c == q;
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const float v = hu[(n4 << 2) + threadIdx.x];
        const int c = v <= HU_AIR ? 0 : (v > HU_BONE ? 2 : 1);
        for (int q = 0; q < 3; ++q)
            if (c == q) { mn[q] = fminf(mn[q], v); mx[q] = fmaxf(mx[q], v); ++cnt[q]; }
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mn[q] = fminf(mn[q], __shfl_xor(mn[q], o));
            mx[q] = fmaxf(mx[q], __shfl_xor(mx[q], o));
            cnt[q] += __shfl_xor(cnt[q], o);
        }
        if ((threadIdx.x & 63) == 0 && cnt[q]) {
            atomicMin(&st[q], enc(mn[q]));
            atomicMax(&st[3 + q], enc(mx[q]));
            atomicAdd(&st[6 + q], cnt[q] > 0 ? 1u : 0u);  // presence only (a count could overflow 32 bits)
        }
    }
}

struct HuMap {
    float soft_min, dmin, inv_range;
};

__device__ __forceinline__ HuMap hu_constants(const unsigned* st, float mult) {
    const bool has_air = st[6] > 0, has_soft = st[7] > 0, has_bone = st[8] > 0;
    const float amin = dec(st[0]), smin = dec(st[1]), bmin = dec(st[2]), smax = dec(st[4]), bmax = dec(st[5]);
    HuMap m;
    // "soft.min() if soft.any() else volume.min()"
    m.soft_min = has_soft ? smin : fminf(has_air ? amin : INFINITY, has_bone ? bmin : INFINITY);
    float lo = INFINITY, hi = -INFINITY;
    if (has_air) { lo = fminf(lo, m.soft_min); hi = fmaxf(hi, m.soft_min); }
    if (has_soft) { lo = fminf(lo, smin); hi = fmaxf(hi, smax); }
    if (has_bone) {
        const float b0 = bmin * mult, b1 = bmax * mult;
        lo = fminf(lo, fminf(b0, b1));
        hi = fmaxf(hi, fmaxf(b0, b1));
    }
    m.dmin = lo;
    m.inv_range = fmaxf(hi - lo, 1.17549435e-38f);  // .clamp_min(tiny)
    return m;
}

__global__ __launch_bounds__(TB) void k_hu_map(const float* __restrict__ hu, long long n, const unsigned* __restrict__ st,
                                               float mult, float* __restrict__ out) {
    const HuMap m = hu_constants(st, mult);
    auto f = [&](float v) {
        const float d = v <= HU_AIR ? m.soft_min : (v > HU_BONE ? v * mult : v);
        return (d - m.dmin) / m.inv_range;
    };
    const long long n4 = n >> 2;
    const long long stride = (long long)gridDim.x * TB;
    for (long long i = (long long)blockIdx.x * TB + threadIdx.x; i < n4; i += stride) {
        const float4 v = reinterpret_cast<const float4*>(hu)[i];
        reinterpret_cast<float4*>(out)[i] = make_float4(f(v.x), f(v.y), f(v.z), f(v.w));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long long i = (n4 << 2) + threadIdx.x;
        out[i] = f(hu[i]);
    }
}

int vfail(int code, const char* msg) {
    xvr_drr_set_last_error(msg);
    return code;
}

}  // namespace

namespace {

// packed[i] = volume[i] with its low 4 mantissa bits replaced by the label min(max((int)mask[i], 0), 15)
__global__ __launch_bounds__(TB) void k_pack_labels(const float* __restrict__ vol, const float* __restrict__ mask, long long n,
                                                    float* __restrict__ packed) {
    const long long n4 = n >> 2;
    for (long long i = (long long)blockIdx.x * TB + threadIdx.x; i < n4; i += (long long)gridDim.x * TB) {
        const float4 v = reinterpret_cast<const float4*>(vol)[i], m = reinterpret_cast<const float4*>(mask)[i];
        auto pk = [](float d, float l) {
            const unsigned lab = (unsigned)min(max((int)l, 0), 15);
            return __uint_as_float((__float_as_uint(d) & ~15u) | lab);
        };
        reinterpret_cast<float4*>(packed)[i] = make_float4(pk(v.x, m.x), pk(v.y, m.y), pk(v.z, m.z), pk(v.w, m.w));
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n & 3)) {
        const long long i = (n4 << 2) + threadIdx.x;
        const unsigned lab = (unsigned)min(max((int)mask[i], 0), 15);
        packed[i] = __uint_as_float((__float_as_uint(vol[i]) & ~15u) | lab);
    }
}

// pairs[x][yp][z] = (V[x][yp - 1][z], V[x][yp][z]), yp = 0 .. D1, zeros outside: one thread per two z (a float4 store).
// LABELS: V = the label-carrying voxel of k_pack_labels (mantissa bits 0..3 := the label) -- both copies in one pass over the
// volume, for the masked renders of a training step, whose density is new every step and rendered twice.
template <bool LABELS, int ZV>   // ZV = z per thread: 2, or 4 where D2 % 4 == 0 and the buffers are 16-byte aligned
__global__ __launch_bounds__(TB) void k_pack_ypairs(const float* __restrict__ vol, const float* __restrict__ mask, int D0, int D1, int D2,
                                                    float* __restrict__ pairs) {
    const long long rows = (long long)D0 * (D1 + 1);
    const int zh = (D2 + ZV - 1) / ZV;
    const long long total = rows * zh;
    auto pk = [](const float d, const float l) {
        const unsigned lab = (unsigned)min(max((int)l, 0), 15);
        return __uint_as_float((__float_as_uint(d) & ~15u) | lab);
    };
    for (long long t = (long long)blockIdx.x * TB + threadIdx.x; t < total; t += (long long)gridDim.x * TB) {
        const long long row = t / zh;
        const int z = (int)(t - row * zh) * ZV;
        const int x = (int)(row / (D1 + 1)), yp = (int)(row - (long long)x * (D1 + 1));
        const long long olo = ((long long)x * D1 + (yp - 1)) * D2, ohi = ((long long)x * D1 + yp) * D2;
        const bool has_lo = yp >= 1, has_hi = yp <= D1 - 1;
        float* dst = pairs + (row * D2 + z) * 2;
        if (ZV == 4) {
            const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
            float4 lo = has_lo ? *reinterpret_cast<const float4*>(vol + olo + z) : zero;
            float4 hi = has_hi ? *reinterpret_cast<const float4*>(vol + ohi + z) : zero;
            if (LABELS) {
                if (has_lo) {
                    const float4 m = *reinterpret_cast<const float4*>(mask + olo + z);
                    lo = make_float4(pk(lo.x, m.x), pk(lo.y, m.y), pk(lo.z, m.z), pk(lo.w, m.w));
                }
                if (has_hi) {
                    const float4 m = *reinterpret_cast<const float4*>(mask + ohi + z);
                    hi = make_float4(pk(hi.x, m.x), pk(hi.y, m.y), pk(hi.z, m.z), pk(hi.w, m.w));
                }
            }
            reinterpret_cast<float4*>(dst)[0] = make_float4(lo.x, hi.x, lo.y, hi.y);
            reinterpret_cast<float4*>(dst)[1] = make_float4(lo.z, hi.z, lo.w, hi.w);
        } else {
            const bool two = z + 1 < D2;
            auto at = [&](const long long o) { return LABELS ? pk(vol[o], mask[o]) : vol[o]; };
            dst[0] = has_lo ? at(olo + z) : 0.f;
            dst[1] = has_hi ? at(ohi + z) : 0.f;
            if (two) {
                dst[2] = has_lo ? at(olo + z + 1) : 0.f;
                dst[3] = has_hi ? at(ohi + z + 1) : 0.f;
            }
        }
    }
}

// tiles[x / 2][yp][z / 7][x % 2][z - 7 (z / 7) in 0..7] = (V[x][yp - 1][z], V[x][yp][z]): the y-pair copy cut into 128-byte lines of
// 2 x-rows x 8 z-entries, the tiles OVERLAPPING by one entry along z (tile b holds z = 7 b .. 7 b + 7), so that the 16 bytes of any
// (z, z + 1) pair lie inside one tile -- a seventh more memory than the row layout.  A planar patch of samples cuts z-runs of 16
// entries after ~2 voxels when the view is oblique; a compact tile of the same 128 bytes is used more densely
// (tools/sim_forward_lines.py).  Until late in round 4 the tiles were 4 x 4 at stride 3 (a third more memory): the forward is
// bound by fabric lines, and the 2 x 8 tiles need fewer of them per voxel -- 5.23 against 5.42 ms at C2.
// HU (round 5): `vol` holds Hounsfield units and the copy holds their DENSITY -- k_hu_map's piecewise map, the same expressions, applied
// on the way: the masked renders of a training step (/root/reference/src/xvr/model/trainer.py:196-204) then never write or read a
// density volume of their own (512 MiB each way at 512^3).
template <bool LABELS, bool HU = false>
__global__ __launch_bounds__(TB) void k_pack_ytiles(const float* __restrict__ vol, const float* __restrict__ mask, int D0, int D1, int D2,
                                                    float* __restrict__ tiles, const unsigned* __restrict__ hu_stats = nullptr, float hu_mult = 1.f) {
    HuMap hm = {0.f, 0.f, 1.f};
    if (HU) hm = hu_constants(hu_stats, hu_mult);
    auto dens = [&](const float v) {
        if (!HU) return v;
        const float d = v <= HU_AIR ? hm.soft_min : (v > HU_BONE ? v * hu_mult : v);
        return (d - hm.dmin) / hm.inv_range;
    };
    // one thread per 16 bytes of the copy (two z-entries of one tile row): consecutive threads write consecutive 16 bytes
    const int nbx = (D0 + 1) >> 1, nbz = (D2 - 2) / 7 + 1;
    const long long total = (long long)nbx * (D1 + 1) * nbz * 8;
    auto pk = [](const float d, const float l) {
        const unsigned lab = (unsigned)min(max((int)l, 0), 15);
        return __uint_as_float((__float_as_uint(d) & ~15u) | lab);
    };
    for (long long t = (long long)blockIdx.x * TB + threadIdx.x; t < total; t += (long long)gridDim.x * TB) {
        const int piece = (int)(t & 7), xr = piece >> 2, quarter = piece & 3;
        const long long tile = t >> 3;
        const int bz = (int)(tile % nbz);
        const long long row = tile / nbz;
        const int yp = (int)(row % (D1 + 1)), x = (int)(row / (D1 + 1)) * 2 + xr;
        float v[4];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int z = 7 * bz + 2 * quarter + r;
            const bool in = x < D0 && z < D2;
            const long long olo = ((long long)x * D1 + (yp - 1)) * D2 + z, ohi = ((long long)x * D1 + yp) * D2 + z;
            v[2 * r] = in && yp >= 1 ? (LABELS ? pk(dens(vol[olo]), mask[olo]) : dens(vol[olo])) : 0.f;
            v[2 * r + 1] = in && yp <= D1 - 1 ? (LABELS ? pk(dens(vol[ohi]), mask[ohi]) : dens(vol[ohi])) : 0.f;
        }
        reinterpret_cast<float4*>(tiles)[t] = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// bricks[x / 2][y / 2][z / 8][x % 2][y % 2][z % 8]: one 128-byte line = a 2 x 2 x 8 block of voxels (zeros beyond the volume);
// one thread per 4 z of one (x, y) = one 16-byte load (where aligned) and one 16-byte store
__global__ __launch_bounds__(TB) void k_pack_bricks(const float* __restrict__ vol, int D0, int D1, int D2, float* __restrict__ bricks) {
    const int nx = (D0 + 1) / 2, ny = (D1 + 1) / 2, nz = (D2 + 7) / 8;
    const long long total = (long long)nx * ny * nz * 8;   // (x % 2, y % 2, half of the 8 z) pieces of 4 z
    for (long long t = (long long)blockIdx.x * TB + threadIdx.x; t < total; t += (long long)gridDim.x * TB) {
        const int row = (int)(t & 7);
        const long long blk = t >> 3;
        const int bz = (int)(blk % nz), by = (int)((blk / nz) % ny), bx = (int)(blk / ((long long)nz * ny));
        const int x = bx * 2 + (row >> 2), y = by * 2 + ((row >> 1) & 1), z = bz * 8 + (row & 1) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (x < D0 && y < D1) {
            const float* src = vol + ((long long)x * D1 + y) * D2 + z;
            v.x = src[0];
            if (z + 1 < D2) v.y = src[1];
            if (z + 2 < D2) v.z = src[2];
            if (z + 3 < D2) v.w = src[3];
        }
        reinterpret_cast<float4*>(bricks)[t] = v;
    }
}

}  // namespace

extern "C" {

size_t xvr_drr_bricks_bytes(int D0, int D1, int D2) {
    if (D0 <= 0 || D1 <= 0 || D2 <= 0) return 0;
    return (size_t)((D0 + 1) / 2) * (size_t)((D1 + 1) / 2) * (size_t)((D2 + 7) / 8) * 32 * sizeof(float);
}

int xvr_drr_pack_bricks(const float* volume, int D0, int D1, int D2, float* bricks, void* stream_) {
    if (!volume || !bricks || D0 < 2 || D1 < 2 || D2 < 2) return vfail(XVR_DRR_E_ARG, "bad argument");
    if (reinterpret_cast<uintptr_t>(bricks) & 15u) return vfail(XVR_DRR_E_ARG, "the brick copy must be 16-byte aligned");
    const long long total = (long long)((D0 + 1) / 2) * ((D1 + 1) / 2) * ((D2 + 7) / 8) * 8;
    if (total * 4 >= (1LL << 31)) return vfail(XVR_DRR_E_UNSUPPORTED, "brick copy has >= 2^31 elements");
    const long long blocks = (total + TB - 1) / TB;
    hipLaunchKernelGGL(k_pack_bricks, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(TB), 0, (hipStream_t)stream_, volume, D0, D1,
                       D2, bricks);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? XVR_DRR_OK : vfail(XVR_DRR_E_LAUNCH, hipGetErrorString(e));
}

size_t xvr_drr_ypairs_bytes(int D0, int D1, int D2) {
    if (D0 <= 0 || D1 <= 0 || D2 <= 0) return 0;
    return (size_t)D0 * (size_t)(D1 + 1) * (size_t)D2 * 2 * sizeof(float);
}

static int pack_ypairs_impl(const float* volume, const float* mask, int D0, int D1, int D2, float* pairs, void* stream_) {
    if (!volume || !pairs || D0 < 2 || D1 < 2 || D2 < 2) return vfail(XVR_DRR_E_ARG, "bad argument");
    if ((long long)D0 * (D1 + 1) * D2 * 2 >= (1LL << 31)) return vfail(XVR_DRR_E_UNSUPPORTED, "y-pair copy has >= 2^31 elements");
    const bool wide = D2 % 4 == 0 && ((reinterpret_cast<uintptr_t>(volume) | reinterpret_cast<uintptr_t>(pairs) | reinterpret_cast<uintptr_t>(mask)) & 15u) == 0;
    const long long total = (long long)D0 * (D1 + 1) * (wide ? D2 / 4 : (D2 + 1) / 2);
    const long long blocks = (total + TB - 1) / TB;
    const dim3 grid((unsigned)(blocks < 16384 ? blocks : 16384));
    hipStream_t stream = (hipStream_t)stream_;
    if (mask && wide) hipLaunchKernelGGL((k_pack_ypairs<true, 4>), grid, dim3(TB), 0, stream, volume, mask, D0, D1, D2, pairs);
    else if (mask) hipLaunchKernelGGL((k_pack_ypairs<true, 2>), grid, dim3(TB), 0, stream, volume, mask, D0, D1, D2, pairs);
    else if (wide) hipLaunchKernelGGL((k_pack_ypairs<false, 4>), grid, dim3(TB), 0, stream, volume, mask, D0, D1, D2, pairs);
    else hipLaunchKernelGGL((k_pack_ypairs<false, 2>), grid, dim3(TB), 0, stream, volume, mask, D0, D1, D2, pairs);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? XVR_DRR_OK : vfail(XVR_DRR_E_LAUNCH, hipGetErrorString(e));
}

int xvr_drr_pack_ypairs(const float* volume, int D0, int D1, int D2, float* pairs, void* stream_) {
    return pack_ypairs_impl(volume, nullptr, D0, D1, D2, pairs, stream_);
}

int xvr_drr_pack_labels_ypairs(const float* volume, const float* mask, int D0, int D1, int D2, float* pairs, void* stream_) {
    if (!mask) return vfail(XVR_DRR_E_ARG, "bad argument");
    return pack_ypairs_impl(volume, mask, D0, D1, D2, pairs, stream_);
}

size_t xvr_drr_ytiles_bytes(int D0, int D1, int D2) {
    if (D0 <= 0 || D1 <= 0 || D2 < 2) return 0;
    return (size_t)((D0 + 1) / 2) * (size_t)(D1 + 1) * (size_t)((D2 - 2) / 7 + 1) * 32 * sizeof(float);
}

static int pack_ytiles_impl(const float* volume, const float* mask, int D0, int D1, int D2, float* tiles, void* stream_,
                            const void* hu_stats = nullptr, float hu_mult = 1.f) {
    if (!volume || !tiles || D0 < 2 || D1 < 2 || D2 < 2) return vfail(XVR_DRR_E_ARG, "bad argument");
    if (reinterpret_cast<uintptr_t>(tiles) & 15u) return vfail(XVR_DRR_E_ARG, "the tiled copy must be 16-byte aligned");
    const long long total = (long long)((D0 + 1) / 2) * (D1 + 1) * ((D2 - 2) / 7 + 1) * 8;   // 16-byte pieces
    if (total * 4 >= (1LL << 31)) return vfail(XVR_DRR_E_UNSUPPORTED, "tiled y-pair copy has >= 2^31 elements");
    if (D2 >= 8192) return vfail(XVR_DRR_E_UNSUPPORTED, "tiled y-pair copy: D2 must be below 8192 (the march's z / 7 is a multiply-shift)");
    const long long blocks = (total + TB - 1) / TB;
    const dim3 grid((unsigned)(blocks < 65536 ? blocks : 65536));
    const unsigned* st = static_cast<const unsigned*>(hu_stats);
    if (st && mask) hipLaunchKernelGGL((k_pack_ytiles<true, true>), grid, dim3(TB), 0, (hipStream_t)stream_, volume, mask, D0, D1, D2, tiles, st, hu_mult);
    else if (st) hipLaunchKernelGGL((k_pack_ytiles<false, true>), grid, dim3(TB), 0, (hipStream_t)stream_, volume, mask, D0, D1, D2, tiles, st, hu_mult);
    else if (mask) hipLaunchKernelGGL((k_pack_ytiles<true, false>), grid, dim3(TB), 0, (hipStream_t)stream_, volume, mask, D0, D1, D2, tiles, st, hu_mult);
    else hipLaunchKernelGGL((k_pack_ytiles<false, false>), grid, dim3(TB), 0, (hipStream_t)stream_, volume, mask, D0, D1, D2, tiles, st, hu_mult);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? XVR_DRR_OK : vfail(XVR_DRR_E_LAUNCH, hipGetErrorString(e));
}

int xvr_drr_pack_ytiles(const float* volume, int D0, int D1, int D2, float* tiles, void* stream_) {
    return pack_ytiles_impl(volume, nullptr, D0, D1, D2, tiles, stream_);
}

int xvr_drr_pack_labels_ytiles(const float* volume, const float* mask, int D0, int D1, int D2, float* tiles, void* stream_) {
    if (!mask) return vfail(XVR_DRR_E_ARG, "bad argument");
    return pack_ytiles_impl(volume, mask, D0, D1, D2, tiles, stream_);
}

int xvr_drr_pack_hu_labels_ytiles(const float* hu, const float* mask, const void* stats, float bone_multiplier, int D0, int D1, int D2,
                                  float* tiles, void* stream_) {
    if (!mask || !stats) return vfail(XVR_DRR_E_ARG, "bad argument");
    return pack_ytiles_impl(hu, mask, D0, D1, D2, tiles, stream_, stats, bone_multiplier);
}

int xvr_drr_pack_labels(const float* volume, const float* mask, long long n, float* packed, void* stream_) {
    if (!volume || !mask || !packed || n <= 0) return vfail(XVR_DRR_E_ARG, "bad argument");
    if ((reinterpret_cast<uintptr_t>(volume) | reinterpret_cast<uintptr_t>(mask) | reinterpret_cast<uintptr_t>(packed)) & 15u)
        return vfail(XVR_DRR_E_ARG, "volumes must be 16-byte aligned");
    const long long blocks = ((n >> 2) + TB - 1) / TB;
    hipLaunchKernelGGL(k_pack_labels, dim3((unsigned)(blocks < 8192 ? (blocks > 0 ? blocks : 1) : 8192)), dim3(TB), 0,
                       (hipStream_t)stream_, volume, mask, n, packed);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? XVR_DRR_OK : vfail(XVR_DRR_E_LAUNCH, hipGetErrorString(e));
}

int xvr_drr_hu_stats(const float* hu, long long n, void* stats, void* stream_) {
    if (!hu || !stats || n <= 0) return vfail(XVR_DRR_E_ARG, "bad argument");
    if (reinterpret_cast<uintptr_t>(hu) & 15u) return vfail(XVR_DRR_E_ARG, "volume must be 16-byte aligned");
    hipStream_t stream = (hipStream_t)stream_;
    // mins start at 0xffffffff (12 bytes), maxes and presence at 0 (device memsets: graph-capture safe)
    hipError_t e = hipMemsetAsync(stats, 0xff, 12, stream);
    if (e == hipSuccess) e = hipMemsetAsync(static_cast<char*>(stats) + 12, 0, 36, stream);
    if (e != hipSuccess) return vfail(XVR_DRR_E_LAUNCH, hipGetErrorString(e));
    const long long blocks = ((n >> 2) + TB - 1) / TB;
    hipLaunchKernelGGL(k_hu_stats, dim3((unsigned)(blocks < 4096 ? (blocks > 0 ? blocks : 1) : 4096)), dim3(TB), 0, stream, hu, n,
                       static_cast<unsigned*>(stats));
    e = hipGetLastError();
    return e == hipSuccess ? XVR_DRR_OK : vfail(XVR_DRR_E_LAUNCH, hipGetErrorString(e));
}

int xvr_drr_hu_to_density(const float* hu, long long n, const void* stats, float bone_multiplier, float* density,
                          void* stream_) {
    if (!hu || !stats || !density || n <= 0) return vfail(XVR_DRR_E_ARG, "bad argument");
    if ((reinterpret_cast<uintptr_t>(hu) | reinterpret_cast<uintptr_t>(density)) & 15u)
        return vfail(XVR_DRR_E_ARG, "volumes must be 16-byte aligned");
    const long long blocks = ((n >> 2) + TB - 1) / TB;
    hipLaunchKernelGGL(k_hu_map, dim3((unsigned)(blocks < 8192 ? (blocks > 0 ? blocks : 1) : 8192)), dim3(TB), 0,
                       (hipStream_t)stream_, hu, n, static_cast<const unsigned*>(stats), bone_multiplier, density);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? XVR_DRR_OK : vfail(XVR_DRR_E_LAUNCH, hipGetErrorString(e));
}

}  // extern "C"
