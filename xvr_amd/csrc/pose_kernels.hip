// Pose arithmetic of the registration loop for gfx950: Euler angles + translation -> camera vector,
// its chain rule, Adam, ReduceLROnPlateau and the stopping rule, all on the device (include/xvr_pose.h).
// One lane per pose: the work is a 24x12 mat-vec and a dozen scalars, the point is to replace ~200
// framework launches and a host sync per iteration by two launches and no sync.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "xvr_drr.h"
#include "xvr_pose.h"
#include "pose_device.hiph"

extern "C" void xvr_drr_set_last_error(const char* msg);

namespace {

int pfail(int code, const char* msg) {
    xvr_drr_set_last_error(msg);
    return code;
}

__global__ void k_pose_camera_fwd(const float* __restrict__ rot, const float* __restrict__ xyz, int B, Axes ax,
                                  const float* __restrict__ G, const float* __restrict__ c, float* __restrict__ cam) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float th[3] = {rot[b * 3], rot[b * 3 + 1], rot[b * 3 + 2]};
    float t[3] = {xyz[b * 3], xyz[b * 3 + 1], xyz[b * 3 + 2]};
    float R[9], m[12];
    pose_matrix(ax, th, t, R, m);
    for (int r = 0; r < 24; ++r) cam[b * 24 + r] = camera_row(G, c, m, r);
}

__global__ __launch_bounds__(64) void k_pose_camera_bwd(const float* __restrict__ rot, const float* __restrict__ xyz, int B,
                                                        Axes ax, const float* __restrict__ G, const float* __restrict__ g_cam,
                                                        float* __restrict__ g_rot, float* __restrict__ g_xyz) {
    const int b = blockIdx.x;   // one wavefront per pose
    float gm[12], gth[3], gt[3];
    wave_gt_g(G, const_cast<float*>(g_cam) + (size_t)b * 24, false, gm);
    float th[3] = {rot[b * 3], rot[b * 3 + 1], rot[b * 3 + 2]};
    float t[3] = {xyz[b * 3], xyz[b * 3 + 1], xyz[b * 3 + 2]};
    pose_chain(ax, th, t, gm, gth, gt);
    if (threadIdx.x < 3) {
        g_rot[b * 3 + threadIdx.x] = gth[threadIdx.x];
        g_xyz[b * 3 + threadIdx.x] = gt[threadIdx.x];
    }
}

__global__ void k_pose_opt_init(xvr_pose_opt_state* st, int B, float lr_rot, float lr_xyz) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    xvr_pose_opt_state s;
    for (int i = 0; i < XVR_POSE_MAX_PARAMS; ++i) s.m[i] = s.v[i] = 0.f;
    s.lr[0] = lr_rot;
    s.lr[1] = lr_xyz;
    s.seen_lr = INFINITY;
    s.step = s.n_bad = s.n_plateaus = s.done = s.iter = 0;
    s.best = -INFINITY;
    st[b] = s;
}

// KIND 0: Euler angles, chain rule in closed form (pose_chain).  Any other parameterisation (xvr_pose_convert_forward's kinds): the
// 12 x (k + 3) Jacobian the forward of THIS iteration stored (k_pose_camera_param) -- g = J^T G^T g_cam; the first k parameters
// are the rotation group (lr[0]), the last three the translation (lr[1]), as the reference's two Adam groups
// (/root/reference/src/xvr/registrar/base.py:221-228).
constexpr int CV_MAXN = XVR_POSE_MAX_PARAMS;
__global__ __launch_bounds__(64) void k_pose_opt_step(float* __restrict__ rot, float* __restrict__ xyz, int B, xvr_pose_opt_spec sp, int kind, int k,
                                const float* __restrict__ G, const float* __restrict__ jac, float* __restrict__ g_cam,
                                const float* __restrict__ loss, xvr_pose_opt_state* __restrict__ state, float* __restrict__ history) {
    const int b = blockIdx.x;   // one wavefront per pose; every lane carries the scalars, lane 0 writes
    float gm[12];
    wave_gt_g(G, g_cam + (size_t)b * 24, true, gm);   // consumed: the next rays-backward accumulates from zero
    xvr_pose_opt_state s = state[b];
    if (s.done) return;
    const int n = k + 3;
    float p[CV_MAXN], g[CV_MAXN];
    for (int i = 0; i < k; ++i) p[i] = rot[(size_t)b * k + i];
    for (int i = 0; i < 3; ++i) p[k + i] = xyz[b * 3 + i];
    if (kind == 0) {
        Axes ax = {{sp.axes[0], sp.axes[1], sp.axes[2]}};
        pose_chain(ax, p, p + 3, gm, g, g + 3);
    } else {
        const float* J = jac + (size_t)b * 12 * CV_MAXN;
        for (int d = 0; d < n; ++d) {
            float acc = 0.f;
            for (int e = 0; e < 12; ++e) acc = fmaf(J[e * CV_MAXN + d], gm[e], acc);
            g[d] = acc;
        }
    }
    if (threadIdx.x != 0) return;   // (after the loads above: every lane read the same, unmodified state)
    pose_opt_update(sp, k, p, g, loss[b], s, history, b);   // Adam, ReduceLROnPlateau, stopping rule, history row (pose_device.hiph)
    for (int i = 0; i < k; ++i) rot[(size_t)b * k + i] = p[i];
    for (int i = 0; i < 3; ++i) xyz[b * 3 + i] = p[k + i];
    state[b] = s;
}

// ---------------------------------------------------------------------------------------------
// Double geodesic on SE(3) for the training loss (/root/reference/src/xvr/model/loss.py:27-28,41-48):
//   R = Ra^T Rb,  angle = atan2(|axis part of R|, (tr R - 1) / 2),  ang = sdd / 2 * angle,
//   trans = |ta - tb|,  d = sqrt(ang^2 + trans^2 + eps)
// and the gradient of d w.r.t. (Rb, tb).  Matrices are row-major 4x4, top three rows used.
// ---------------------------------------------------------------------------------------------
struct Geo {
    float ang, trans, d;
};

__device__ inline Geo geodesic(const float* Ra, const float* ta, const float* Rb, const float* tb, float sdd, float eps,
                               float* gRb, float* gtb) {
    float R[9];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) R[a * 3 + b] = Ra[0 + a] * Rb[0 + b] + Ra[3 + a] * Rb[3 + b] + Ra[6 + a] * Rb[6 + b];
    const float c = 0.5f * ((R[0] + R[4] + R[8]) - 1.f);
    const float w21 = R[7] - R[5], w02 = R[2] - R[6], w10 = R[3] - R[1];
    const float q = w21 * w21 + w02 * w02 + w10 * w10 + 1e-24f;
    const float sn = 0.5f * sqrtf(q);
    const float k = 0.5f * sdd;
    Geo g;
    g.ang = k * atan2f(sn, c);
    const float d0 = ta[0] - tb[0], d1 = ta[1] - tb[1], d2 = ta[2] - tb[2];
    g.trans = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
    g.d = sqrtf(g.ang * g.ang + g.trans * g.trans + eps);
    if (gRb) {
        const float ga = g.ang / g.d, gtau = g.trans / g.d;
        const float den = sn * sn + c * c;
        const float gs = ga * k * c / den, gc = -ga * k * sn / den;
        // d/dR = gc / 2 * I + gs / (4 sn) * (R - R^T)
        const float f = gs / (4.f * sn);
        float GR[9];
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) GR[a * 3 + b] = f * (R[a * 3 + b] - R[b * 3 + a]) + (a == b ? 0.5f * gc : 0.f);
        for (int r = 0; r < 3; ++r)
            for (int b = 0; b < 3; ++b) gRb[r * 3 + b] = Ra[r * 3] * GR[b] + Ra[r * 3 + 1] * GR[3 + b] + Ra[r * 3 + 2] * GR[6 + b];
        const float it = g.trans > 0.f ? gtau / g.trans : 0.f;
        gtb[0] = -it * d0; gtb[1] = -it * d1; gtb[2] = -it * d2;
    }
    return g;
}

__device__ inline void load_rt(const float* M, float* R, float* t) {
    for (int r = 0; r < 3; ++r) {
        R[r * 3] = M[r * 4]; R[r * 3 + 1] = M[r * 4 + 1]; R[r * 3 + 2] = M[r * 4 + 2];
        t[r] = M[r * 4 + 3];
    }
}

// rel = A_j A_i^{-1} for rigid A: R = Rj Ri^T, t = tj - R ti
__device__ inline void relative(const float* Ri, const float* ti, const float* Rj, const float* tj, float* R, float* t) {
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) R[a * 3 + b] = Rj[a * 3] * Ri[b * 3] + Rj[a * 3 + 1] * Ri[b * 3 + 1] + Rj[a * 3 + 2] * Ri[b * 3 + 2];
    for (int a = 0; a < 3; ++a) t[a] = tj[a] - (R[a * 3] * ti[0] + R[a * 3 + 1] * ti[1] + R[a * 3 + 2] * ti[2]);
}

// out [3][N] = (ang, trans, d) of pose pairs (A_n, B_n); gB [N][12] = d d_n / d (R_b | t_b) (nullable)
__global__ void k_geodesic(const float* __restrict__ A, const float* __restrict__ Bm, int N, float sdd, float eps,
                           float* __restrict__ out, float* __restrict__ gB) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float Ra[9], ta[3], Rb[9], tb[3], gR[9], gt[3];
    load_rt(A + (size_t)n * 16, Ra, ta);
    load_rt(Bm + (size_t)n * 16, Rb, tb);
    const Geo g = geodesic(Ra, ta, Rb, tb, sdd, eps, gB ? gR : nullptr, gt);
    out[n] = g.ang; out[N + n] = g.trans; out[2 * N + n] = g.d;
    if (gB) {
        for (int i = 0; i < 9; ++i) gB[(size_t)n * 12 + i] = gR[i];
        for (int i = 0; i < 3; ++i) gB[(size_t)n * 12 + 9 + i] = gt[i];
    }
}

__device__ __forceinline__ int pair_index(int i, int j, int B) { return i * (2 * B - i - 1) / 2 + (j - i - 1); }

// multiview consistency: for every pair i < j the double geodesic between the true and the predicted RELATIVE
// pose A_j A_i^{-1}; mvc in the order of torch.triu_indices(B, B, 1)
__global__ void k_multiview_fwd(const float* __restrict__ T, const float* __restrict__ P, int B, float sdd, float eps,
                                float* __restrict__ mvc) {
    const int i = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= B || j <= i) return;
    float Ri[9], ti[3], Rj[9], tj[3], Rt[9], tt[3], Rp[9], tp[3];
    load_rt(T + (size_t)i * 16, Ri, ti); load_rt(T + (size_t)j * 16, Rj, tj);
    relative(Ri, ti, Rj, tj, Rt, tt);
    load_rt(P + (size_t)i * 16, Ri, ti); load_rt(P + (size_t)j * 16, Rj, tj);
    relative(Ri, ti, Rj, tj, Rp, tp);
    mvc[pair_index(i, j, B)] = geodesic(Rt, tt, Rp, tp, sdd, eps, nullptr, nullptr).d;
}

// gradient w.r.t. the predicted poses: thread m adds up, in a fixed order, the terms of every pair that
// contains pose m (no atomics).  gP [B][12] = (R | t) rows of pose m.
__global__ void k_multiview_bwd(const float* __restrict__ T, const float* __restrict__ P, const float* __restrict__ gout,
                                int B, float sdd, float eps, float* __restrict__ gP) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= B) return;
    float accR[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, acct[3] = {0, 0, 0};
    for (int u = 0; u < B; ++u) {
        if (u == m) continue;
        const int i = m < u ? m : u, j = m < u ? u : m;
        float Ri[9], ti[3], Rj[9], tj[3], Rt[9], tt[3], Rp[9], tp[3], gR[9], gt[3];
        load_rt(T + (size_t)i * 16, Ri, ti); load_rt(T + (size_t)j * 16, Rj, tj);
        relative(Ri, ti, Rj, tj, Rt, tt);
        load_rt(P + (size_t)i * 16, Ri, ti); load_rt(P + (size_t)j * 16, Rj, tj);   // from here on: the predicted pair
        relative(Ri, ti, Rj, tj, Rp, tp);
        geodesic(Rt, tt, Rp, tp, sdd, eps, gR, gt);
        const float go = gout[pair_index(i, j, B)];
        // tp = tj - Rp ti: the translation gradient also reaches Rp
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) gR[a * 3 + b] -= gt[a] * ti[b];
        if (m == j) {          // Rp = Rj Ri^T: d/dRj = G Ri;  d/dtj = g_t
            for (int a = 0; a < 3; ++a) {
                for (int b = 0; b < 3; ++b)
                    accR[a * 3 + b] += go * (gR[a * 3] * Ri[b] + gR[a * 3 + 1] * Ri[3 + b] + gR[a * 3 + 2] * Ri[6 + b]);
                acct[a] += go * gt[a];
            }
        } else {               // d/dRi = G^T Rj;  d/dti = -Rp^T g_t
            for (int a = 0; a < 3; ++a) {
                for (int b = 0; b < 3; ++b)
                    accR[a * 3 + b] += go * (gR[a] * Rj[b] + gR[3 + a] * Rj[3 + b] + gR[6 + a] * Rj[6 + b]);
                acct[a] -= go * (Rp[a] * gt[0] + Rp[3 + a] * gt[1] + Rp[6 + a] * gt[2]);
            }
        }
    }
    for (int a = 0; a < 9; ++a) gP[(size_t)m * 12 + a] = accR[a];
    for (int a = 0; a < 3; ++a) gP[(size_t)m * 12 + 9 + a] = acct[a];
}

bool axes_ok(const int* a) {
    if (!a) return false;
    for (int i = 0; i < 3; ++i)
        if (a[i] < 0 || a[i] > 2) return false;
    return a[1] != a[0] && a[1] != a[2];
}

int launched(const char* what) {
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) return XVR_DRR_OK;
    (void)what;
    return pfail(XVR_DRR_E_LAUNCH, hipGetErrorString(e));
}


// ---------------------------------------------------------------------------------------------
// convert(rotation, translation, parameterization) -> 4x4 pose, and its backward, in ONE launch each
// (/root/reference/src/xvr/model/network.py:49-56 calls diffdrr's `convert` on the regressor's output every training step;
// registration does with non-Euler parameterisations).  The framework evaluates it as 40-110 tiny launches forward +
// backward: 0.8-2.0 ms at 116 poses.  One thread per pose evaluates the map in forward-mode dual numbers -- one tangent per
// input parameter, at most 10 + 3 -- and stores the 12 x (k + 3) Jacobian; the backward is J^T g.
//   kind 0 euler_angles (axes = convention, radians)   1 axis_angle   2 quaternion (real first)
//        3 quaternion_adjugate (10 numbers)   4 rotation_6d   5 se3_log_map
// The formulas are xvr_amd/pose.py's (the PyTorch3D heritage diffdrr shares), branch for branch.
// ---------------------------------------------------------------------------------------------
struct Dual {
    float v;
    float d[CV_MAXN];
};
__device__ inline Dual dconst(float v) { Dual r; r.v = v; for (int i = 0; i < CV_MAXN; ++i) r.d[i] = 0.f; return r; }
__device__ inline Dual dvar(float v, int i) { Dual r = dconst(v); r.d[i] = 1.f; return r; }
__device__ inline Dual operator+(const Dual& a, const Dual& b) { Dual r; r.v = a.v + b.v; for (int i = 0; i < CV_MAXN; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
__device__ inline Dual operator-(const Dual& a, const Dual& b) { Dual r; r.v = a.v - b.v; for (int i = 0; i < CV_MAXN; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
__device__ inline Dual operator-(const Dual& a) { Dual r; r.v = -a.v; for (int i = 0; i < CV_MAXN; ++i) r.d[i] = -a.d[i]; return r; }
__device__ inline Dual operator*(const Dual& a, const Dual& b) { Dual r; r.v = a.v * b.v; for (int i = 0; i < CV_MAXN; ++i) r.d[i] = fmaf(a.v, b.d[i], a.d[i] * b.v); return r; }
__device__ inline Dual operator*(float a, const Dual& b) { Dual r; r.v = a * b.v; for (int i = 0; i < CV_MAXN; ++i) r.d[i] = a * b.d[i]; return r; }
__device__ inline Dual operator+(float a, const Dual& b) { Dual r = b; r.v = a + b.v; return r; }
__device__ inline Dual operator-(float a, const Dual& b) { Dual r = -b; r.v = a - b.v; return r; }
__device__ inline Dual operator/(const Dual& a, const Dual& b) {
    Dual r; const float inv = 1.f / b.v; r.v = a.v * inv;
    for (int i = 0; i < CV_MAXN; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv;
    return r;
}
__device__ inline Dual operator/(float a, const Dual& b) { return dconst(a) / b; }
__device__ inline Dual dsqrt(const Dual& a) { Dual r; r.v = sqrtf(a.v); const float h = 0.5f / r.v; for (int i = 0; i < CV_MAXN; ++i) r.d[i] = a.d[i] * h; return r; }
__device__ inline Dual dsin(const Dual& a) { Dual r; float s, c; sincosf(a.v, &s, &c); r.v = s; for (int i = 0; i < CV_MAXN; ++i) r.d[i] = a.d[i] * c; return r; }
__device__ inline Dual dcos(const Dual& a) { Dual r; float s, c; sincosf(a.v, &s, &c); r.v = c; for (int i = 0; i < CV_MAXN; ++i) r.d[i] = -a.d[i] * s; return r; }

// sin(t)/t, (1 - cos t)/t^2, (t - sin t)/t^3 with the Taylor branches of pose.py's _so3_coeffs
__device__ inline void so3_coeffs(const Dual& th2, Dual& a, Dual& b, Dual& c) {
    if (th2.v < 1e-8f) {
        a = 1.f - (1.f / 6.f) * th2;
        b = 0.5f - (1.f / 24.f) * th2;
        c = (1.f / 6.f) - (1.f / 120.f) * th2;
    } else {
        const Dual th = dsqrt(th2), s = dsin(th), co = dcos(th);
        a = s / th;
        b = (1.f - co) / th2;
        c = (th - s) / (th2 * th);
    }
}
// R (row-major 3x3) of a quaternion that need not be unit (pose.py quaternion_to_matrix)
__device__ inline void quat_to_matrix(const Dual* q, Dual* R) {
    const Dual& r = q[0]; const Dual& i = q[1]; const Dual& j = q[2]; const Dual& k = q[3];
    const Dual two_s = 2.f / (r * r + i * i + j * j + k * k);
    R[0] = 1.f - two_s * (j * j + k * k); R[1] = two_s * (i * j - k * r); R[2] = two_s * (i * k + j * r);
    R[3] = two_s * (i * j + k * r); R[4] = 1.f - two_s * (i * i + k * k); R[5] = two_s * (j * k - i * r);
    R[6] = two_s * (i * k - j * r); R[7] = two_s * (j * k + i * r); R[8] = 1.f - two_s * (i * i + j * j);
}
// R = I + a K + b K^2 (and optionally V = I + b K + c K^2), K = hat(w)
__device__ inline void rodrigues(const Dual* w, const Dual& a, const Dual& b, Dual* R) {
    const Dual x = w[0], y = w[1], z = w[2];
    const Dual xx = x * x, yy = y * y, zz = z * z, xy = x * y, xz = x * z, yz = y * z;
    // K^2 = w w^T - |w|^2 I
    R[0] = 1.f - b * (yy + zz); R[1] = b * xy - a * z;       R[2] = b * xz + a * y;
    R[3] = b * xy + a * z;       R[4] = 1.f - b * (xx + zz); R[5] = b * yz - a * x;
    R[6] = b * xz - a * y;       R[7] = b * yz + a * x;       R[8] = 1.f - b * (xx + yy);
}

// pose p of the batch: R (row-major 3x3) and T = the translation column of the 4x4, as dual numbers over the k + 3 parameters
__device__ inline void convert_dual(const float* __restrict__ rot, const float* __restrict__ xyz, int p, int kind, int k, const Axes ax,
                                    Dual* R, Dual* T) {
    Dual r[10], t[3];
    for (int i = 0; i < k; ++i) r[i] = dvar(rot[(size_t)p * k + i], i);
    for (int i = 0; i < 3; ++i) t[i] = dvar(xyz[(size_t)p * 3 + i], k + i);
    bool rotate_t = true;
    if (kind == 0) {   // intrinsic Euler angles: R = R_c0(a0) R_c1(a1) R_c2(a2)
        Dual M[3][9];
        for (int e = 0; e < 3; ++e) {
            const Dual c = dcos(r[e]), s = dsin(r[e]);
            for (int q = 0; q < 9; ++q) M[e][q] = dconst(0.f);
            const int a = ax.a[e], i = (a + 1) % 3, j = (a + 2) % 3;
            M[e][a * 3 + a] = dconst(1.f);
            M[e][i * 3 + i] = c; M[e][i * 3 + j] = -s;
            M[e][j * 3 + i] = s; M[e][j * 3 + j] = c;
        }
        Dual M01[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) M01[i * 3 + j] = M[0][i * 3] * M[1][j] + M[0][i * 3 + 1] * M[1][3 + j] + M[0][i * 3 + 2] * M[1][6 + j];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) R[i * 3 + j] = M01[i * 3] * M[2][j] + M01[i * 3 + 1] * M[2][3 + j] + M01[i * 3 + 2] * M[2][6 + j];
    } else if (kind == 1) {   // axis-angle: Rodrigues
        Dual a, b, c;
        so3_coeffs(r[0] * r[0] + r[1] * r[1] + r[2] * r[2], a, b, c);
        rodrigues(r, a, b, R);
    } else if (kind == 2) {
        quat_to_matrix(r, R);
    } else if (kind == 3) {   // quaternion adjugate: the column of largest norm of the symmetric 4x4, over that norm
        Dual A[4][4];
        int q = 0;
        for (int i = 0; i < 4; ++i)
            for (int j = i; j < 4; ++j) { A[i][j] = r[q]; A[j][i] = r[q]; ++q; }
        int best = 0;
        float bn = -1.f;
        for (int j = 0; j < 4; ++j) {
            const float nn = sqrtf(A[0][j].v * A[0][j].v + A[1][j].v * A[1][j].v + A[2][j].v * A[2][j].v + A[3][j].v * A[3][j].v);
            if (nn > bn) { bn = nn; best = j; }   // (first maximum, as torch.argmax)
        }
        const Dual nrm = dsqrt(A[0][best] * A[0][best] + A[1][best] * A[1][best] + A[2][best] * A[2][best] + A[3][best] * A[3][best]);
        Dual qq[4];
        for (int i = 0; i < 4; ++i) qq[i] = A[i][best] / nrm;
        quat_to_matrix(qq, R);
    } else if (kind == 4) {   // 6D: Gram-Schmidt of two rows (F.normalize: x / max(|x|, 1e-12))
        const Dual n1 = dsqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
        const Dual d1 = n1.v > 1e-12f ? n1 : dconst(1e-12f);
        Dual b1[3], b2[3];
        for (int i = 0; i < 3; ++i) b1[i] = r[i] / d1;
        const Dual dp = b1[0] * r[3] + b1[1] * r[4] + b1[2] * r[5];
        for (int i = 0; i < 3; ++i) b2[i] = r[3 + i] - dp * b1[i];
        const Dual n2 = dsqrt(b2[0] * b2[0] + b2[1] * b2[1] + b2[2] * b2[2]);
        const Dual d2 = n2.v > 1e-12f ? n2 : dconst(1e-12f);
        for (int i = 0; i < 3; ++i) b2[i] = b2[i] / d2;
        for (int i = 0; i < 3; ++i) { R[i] = b1[i]; R[3 + i] = b2[i]; }
        R[6] = b1[1] * b2[2] - b1[2] * b2[1];
        R[7] = b1[2] * b2[0] - b1[0] * b2[2];
        R[8] = b1[0] * b2[1] - b1[1] * b2[0];
    } else if (kind == 6) {
        // rotation_10d (Peretroukhin et al. 2020; pose.py rotation_10d_to_quaternion): the 10 numbers are the upper triangle of a
        // symmetric 4 x 4 A, the quaternion is the unit eigenvector of its SMALLEST eigenvalue.  Values: cyclic Jacobi on the 4 x 4
        // (converges quadratically; 8 sweeps are far past float32 resolution).  Derivatives by first-order perturbation theory,
        // dq = sum_{j != 0} v_j (v_j^T dA q) / (lambda_0 - lambda_j), with dA / d r_i the symmetric unit matrix of entry i -- the
        // gradient torch.linalg.eigh's backward gives (the sign of q is free: R(q) = R(-q)).
        float A[4][4], V[4][4];
        {
            int q = 0;
            for (int i = 0; i < 4; ++i)
                for (int j = i; j < 4; ++j) { A[i][j] = A[j][i] = r[q].v; ++q; }
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) V[i][j] = i == j ? 1.f : 0.f;
        }
        for (int sweep = 0; sweep < 8; ++sweep) {
            for (int pp = 0; pp < 3; ++pp) {
                for (int qq = pp + 1; qq < 4; ++qq) {
                    const float apq = A[pp][qq];
                    if (fabsf(apq) < 1e-30f) continue;
                    const float tau = (A[qq][qq] - A[pp][pp]) / (2.f * apq);
                    const float tt = (tau >= 0.f ? 1.f : -1.f) / (fabsf(tau) + sqrtf(1.f + tau * tau));
                    const float cc = 1.f / sqrtf(1.f + tt * tt), ss = tt * cc;
                    for (int kx = 0; kx < 4; ++kx) {   // A <- A J (columns pp, qq)
                        const float akp = A[kx][pp], akq = A[kx][qq];
                        A[kx][pp] = cc * akp - ss * akq;
                        A[kx][qq] = ss * akp + cc * akq;
                    }
                    for (int kx = 0; kx < 4; ++kx) {   // A <- J^T A (rows pp, qq)
                        const float apk = A[pp][kx], aqk = A[qq][kx];
                        A[pp][kx] = cc * apk - ss * aqk;
                        A[qq][kx] = ss * apk + cc * aqk;
                    }
                    for (int kx = 0; kx < 4; ++kx) {   // V <- V J
                        const float vkp = V[kx][pp], vkq = V[kx][qq];
                        V[kx][pp] = cc * vkp - ss * vkq;
                        V[kx][qq] = ss * vkp + cc * vkq;
                    }
                }
            }
        }
        int m0 = 0;
        for (int j = 1; j < 4; ++j)
            if (A[j][j] < A[m0][m0]) m0 = j;
        Dual qd[4];
        for (int i = 0; i < 4; ++i) qd[i] = dconst(V[i][m0]);
        {
            int e = 0;
            for (int ri = 0; ri < 4; ++ri) {
                for (int ci = ri; ci < 4; ++ci) {
                    for (int j = 0; j < 4; ++j) {
                        if (j == m0) continue;
                        const float gap = A[m0][m0] - A[j][j];
                        const float num = ri == ci ? V[ri][j] * V[ri][m0] : V[ri][j] * V[ci][m0] + V[ci][j] * V[ri][m0];
                        const float cf = num / gap;
                        for (int i = 0; i < 4; ++i) qd[i].d[e] = fmaf(cf, V[i][j], qd[i].d[e]);
                    }
                    ++e;
                }
            }
        }
        quat_to_matrix(qd, R);
    } else {   // se(3) twist (omega, v): R = exp(omega), t = V(omega) v -- no further rotation of t
        Dual a, b, c, V[9];
        so3_coeffs(r[0] * r[0] + r[1] * r[1] + r[2] * r[2], a, b, c);
        rodrigues(r, a, b, R);
        rodrigues(r, b, c, V);
        for (int i = 0; i < 3; ++i) T[i] = V[i * 3] * t[0] + V[i * 3 + 1] * t[1] + V[i * 3 + 2] * t[2];
        rotate_t = false;
    }
    if (rotate_t)   // C-arm convention: x_world = R (x_cam + t)
        for (int i = 0; i < 3; ++i) T[i] = R[i * 3] * t[0] + R[i * 3 + 1] * t[1] + R[i * 3 + 2] * t[2];
}

// matrix (nullable) [B][16], jac [B][12][CV_MAXN]; with G and c also cam [B][24] = G vec(M[:3,:4]) + c (the registration loop's
// first step for any parameterisation: xvr_pose_camera_forward_param)
__global__ __launch_bounds__(64) void k_pose_convert_fwd(const float* __restrict__ rot, const float* __restrict__ xyz, int B, int kind, int k,
                                                         Axes ax, float* __restrict__ matrix, float* __restrict__ jac,
                                                         const float* __restrict__ G, const float* __restrict__ c, float* __restrict__ cam) {
    const int p = blockIdx.x * 64 + threadIdx.x;
    if (p >= B) return;
    const int n = k + 3;
    Dual R[9], T[3];
    convert_dual(rot, xyz, p, kind, k, ax, R, T);
    float* J = jac + (size_t)p * 12 * CV_MAXN;
    float m12[12];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) {
            m12[i * 4 + j] = R[i * 3 + j].v;
            for (int d = 0; d < n; ++d) J[(i * 4 + j) * CV_MAXN + d] = R[i * 3 + j].d[d];
        }
        m12[i * 4 + 3] = T[i].v;
        for (int d = 0; d < n; ++d) J[(i * 4 + 3) * CV_MAXN + d] = T[i].d[d];
    }
    if (matrix) {
        float* M = matrix + (size_t)p * 16;
        for (int e = 0; e < 12; ++e) M[e] = m12[e];
        M[12] = M[13] = M[14] = 0.f;
        M[15] = 1.f;
    }
    if (cam) {
        for (int r = 0; r < 24; ++r) {
            float sacc = c[r];
            for (int q = 0; q < 12; ++q) sacc = fmaf(G[r * 12 + q], m12[q], sacc);
            cam[(size_t)p * 24 + r] = sacc;
        }
    }
}

// grad_rot [B][k], grad_xyz [B][3] = J^T grad_matrix (rows 0..2 of the 4x4): one thread per (pose, parameter)
__global__ void k_pose_convert_bwd(const float* __restrict__ jac, const float* __restrict__ gmat, int B, int k, float* __restrict__ grot,
                                   float* __restrict__ gxyz) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x, n = k + 3;
    if (idx >= B * n) return;
    const int p = idx / n, d = idx - p * n;
    const float* J = jac + (size_t)p * 12 * CV_MAXN;
    const float* g = gmat + (size_t)p * 16;
    float s = 0.f;
    for (int e = 0; e < 12; ++e) s = fmaf(J[e * CV_MAXN + d], g[e], s);
    if (d < k) grot[(size_t)p * k + d] = s;
    else gxyz[(size_t)p * 3 + (d - k)] = s;
}

}  // namespace

extern "C" int xvr_pose_camera_forward(const float* rot, const float* xyz, int B, const int axes[3], const float* G,
                                       const float* c, float* cam, void* stream) {
    if (!rot || !xyz || !G || !c || !cam || B <= 0) return pfail(XVR_DRR_E_ARG, "bad argument");
    if (!axes_ok(axes)) return pfail(XVR_DRR_E_ARG, "axes must be in {0,1,2} with the middle one distinct from its neighbours");
    Axes ax = {{axes[0], axes[1], axes[2]}};
    hipLaunchKernelGGL(k_pose_camera_fwd, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, rot, xyz, B, ax, G, c, cam);
    return launched("pose_camera_forward");
}

extern "C" int xvr_pose_camera_backward(const float* rot, const float* xyz, int B, const int axes[3], const float* G,
                                        const float* grad_cam, float* grad_rot, float* grad_xyz, void* stream) {
    if (!rot || !xyz || !G || !grad_cam || !grad_rot || !grad_xyz || B <= 0) return pfail(XVR_DRR_E_ARG, "bad argument");
    if (!axes_ok(axes)) return pfail(XVR_DRR_E_ARG, "axes must be in {0,1,2} with the middle one distinct from its neighbours");
    Axes ax = {{axes[0], axes[1], axes[2]}};
    hipLaunchKernelGGL(k_pose_camera_bwd, dim3(B), dim3(64), 0, (hipStream_t)stream, rot, xyz, B, ax, G,
                       grad_cam, grad_rot, grad_xyz);
    return launched("pose_camera_backward");
}

extern "C" size_t xvr_pose_opt_state_bytes(void) { return sizeof(xvr_pose_opt_state); }

extern "C" int xvr_pose_opt_init(xvr_pose_opt_state* state, int B, float lr_rot, float lr_xyz, void* stream) {
    if (!state || B <= 0) return pfail(XVR_DRR_E_ARG, "bad argument");
    hipLaunchKernelGGL(k_pose_opt_init, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, state, B, lr_rot, lr_xyz);
    return launched("pose_opt_init");
}

static const int POSE_K[7] = {3, 3, 4, 10, 6, 3, 10};

extern "C" int xvr_pose_opt_step_param(float* rot, float* xyz, int B, int kind, const xvr_pose_opt_spec* spec, const float* G,
                                       const float* jac, float* grad_cam, const float* loss, xvr_pose_opt_state* state, float* history,
                                       void* stream) {
    if (!rot || !xyz || !spec || !G || !grad_cam || !loss || !state || B <= 0) return pfail(XVR_DRR_E_ARG, "bad argument");
    if (kind < 0 || kind > 6) return pfail(XVR_DRR_E_ARG, "bad parameterisation");
    if (kind == 0 && !axes_ok(spec->axes)) return pfail(XVR_DRR_E_ARG, "axes must be in {0,1,2} with the middle one distinct from its neighbours");
    if (kind != 0 && !jac) return pfail(XVR_DRR_E_ARG, "a non-Euler parameterisation needs the Jacobian xvr_pose_camera_forward_param stored");
    if (spec->max_n_plateaus < 1 || spec->patience < 0 || (history && spec->max_iters < 1))
        return pfail(XVR_DRR_E_ARG, "bad optimiser spec");
    hipLaunchKernelGGL(k_pose_opt_step, dim3(B), dim3(64), 0, (hipStream_t)stream, rot, xyz, B, *spec, kind, POSE_K[kind], G, jac,
                       grad_cam, loss, state, history);
    return launched("pose_opt_step");
}

extern "C" int xvr_pose_opt_step(float* rot, float* xyz, int B, const xvr_pose_opt_spec* spec, const float* G,
                                 float* grad_cam, const float* loss, xvr_pose_opt_state* state, float* history,
                                 void* stream) {
    return xvr_pose_opt_step_param(rot, xyz, B, 0, spec, G, nullptr, grad_cam, loss, state, history, stream);
}

extern "C" int xvr_pose_geodesic(const float* a, const float* b, int N, float sdd, float eps, float* out, float* grad_b,
                                 void* stream) {
    if (!a || !b || !out || N <= 0) return pfail(XVR_DRR_E_ARG, "bad argument");
    hipLaunchKernelGGL(k_geodesic, dim3((N + 127) / 128), dim3(128), 0, (hipStream_t)stream, a, b, N, sdd, eps, out, grad_b);
    return launched("pose_geodesic");
}

extern "C" int xvr_pose_multiview_forward(const float* true_pose, const float* pred_pose, int B, float sdd, float eps,
                                          float* mvc, void* stream) {
    if (!true_pose || !pred_pose || !mvc || B < 2 || B > 46340) return pfail(XVR_DRR_E_ARG, "bad argument");
    hipLaunchKernelGGL(k_multiview_fwd, dim3((B + 127) / 128, B), dim3(128), 0, (hipStream_t)stream, true_pose, pred_pose, B,
                       sdd, eps, mvc);
    return launched("pose_multiview_forward");
}

extern "C" int xvr_pose_multiview_backward(const float* true_pose, const float* pred_pose, const float* grad_mvc, int B,
                                           float sdd, float eps, float* grad_pred, void* stream) {
    if (!true_pose || !pred_pose || !grad_mvc || !grad_pred || B < 2 || B > 46340) return pfail(XVR_DRR_E_ARG, "bad argument");
    hipLaunchKernelGGL(k_multiview_bwd, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, true_pose, pred_pose, grad_mvc, B,
                       sdd, eps, grad_pred);
    return launched("pose_multiview_backward");
}

extern "C" size_t xvr_pose_convert_jacobian_floats(int B) { return B > 0 ? (size_t)B * 12 * CV_MAXN : 0; }

static int convert_axes(int kind, const int axes[3], Axes* ax) {
    *ax = Axes{{0, 1, 2}};
    if (kind == 0) {
        if (!axes) return pfail(XVR_DRR_E_ARG, "euler angles need a convention");
        for (int i = 0; i < 3; ++i) {
            if (axes[i] < 0 || axes[i] > 2) return pfail(XVR_DRR_E_ARG, "axes must be 0, 1 or 2");
            ax->a[i] = axes[i];
        }
    }
    return XVR_DRR_OK;
}

extern "C" int xvr_pose_convert_forward(const float* rot, const float* xyz, int B, int kind, const int axes[3], float* matrix,
                                        float* jac, void* stream) {
    if (!rot || !xyz || !matrix || !jac) return pfail(XVR_DRR_E_ARG, "null pointer argument");
    if (B <= 0 || kind < 0 || kind > 6) return pfail(XVR_DRR_E_ARG, "bad batch size or parameterisation");
    Axes ax;
    if (int rc = convert_axes(kind, axes, &ax)) return rc;
    hipLaunchKernelGGL(k_pose_convert_fwd, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, (hipStream_t)stream, rot, xyz, B, kind, POSE_K[kind], ax,
                       matrix, jac, (const float*)nullptr, (const float*)nullptr, (float*)nullptr);
    return launched("xvr_pose_convert_forward");
}

extern "C" int xvr_pose_camera_forward_param(const float* rot, const float* xyz, int B, int kind, const int axes[3], const float* G,
                                             const float* c, float* cam, float* jac, void* stream) {
    if (!rot || !xyz || !G || !c || !cam || !jac) return pfail(XVR_DRR_E_ARG, "null pointer argument");
    if (B <= 0 || kind < 0 || kind > 6) return pfail(XVR_DRR_E_ARG, "bad batch size or parameterisation");
    Axes ax;
    if (int rc = convert_axes(kind, axes, &ax)) return rc;
    hipLaunchKernelGGL(k_pose_convert_fwd, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, (hipStream_t)stream, rot, xyz, B, kind, POSE_K[kind], ax,
                       (float*)nullptr, jac, G, c, cam);
    return launched("xvr_pose_camera_forward_param");
}

extern "C" int xvr_pose_convert_backward(const float* jac, const float* grad_matrix, int B, int kind, float* grad_rot, float* grad_xyz,
                                         void* stream) {
    if (!jac || !grad_matrix || !grad_rot || !grad_xyz) return pfail(XVR_DRR_E_ARG, "null pointer argument");
    if (B <= 0 || kind < 0 || kind > 6) return pfail(XVR_DRR_E_ARG, "bad batch size or parameterisation");
    const int n = B * (POSE_K[kind] + 3);
    hipLaunchKernelGGL(k_pose_convert_bwd, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, (hipStream_t)stream, jac, grad_matrix, B, POSE_K[kind],
                       grad_rot, grad_xyz);
    return launched("xvr_pose_convert_backward");
}
