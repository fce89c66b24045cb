// Pose arithmetic of the registration loop for gfx950: Euler angles + translation -> camera vector,
// its chain rule, Adam, ReduceLROnPlateau and the stopping rule, all on the device (include/xvr_pose.h).
// One lane per pose: the work is a 24x12 mat-vec and a dozen scalars, the point is to replace ~200
// framework launches and a host sync per iteration by two launches and no sync.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "xvr_drr.h"
#include "xvr_pose.h"

extern "C" void xvr_drr_set_last_error(const char* msg);

namespace {

int pfail(int code, const char* msg) {
    xvr_drr_set_last_error(msg);
    return code;
}

struct Axes {
    int a[3];
};

// R = rotation by `ang` about axis `ax`; dR = its derivative w.r.t. the angle (row-major 3x3).
__device__ inline void axis_rotation(int ax, float ang, float* R, float* dR) {
    float s, c;
    sincosf(ang, &s, &c);
    for (int i = 0; i < 9; ++i) R[i] = dR[i] = 0.f;
    const int i = (ax + 1) % 3, j = (ax + 2) % 3;   // the plane the axis rotates: (i, j) right-handed
    R[ax * 3 + ax] = 1.f;
    R[i * 3 + i] = c;  R[i * 3 + j] = -s;
    R[j * 3 + i] = s;  R[j * 3 + j] = c;
    dR[i * 3 + i] = -s; dR[i * 3 + j] = -c;
    dR[j * 3 + i] = c;  dR[j * 3 + j] = -s;
}

__device__ inline void mat3_mul(const float* A, const float* B, float* C) {
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            C[r * 3 + c] = fmaf(A[r * 3 + 2], B[6 + c], fmaf(A[r * 3 + 1], B[3 + c], A[r * 3] * B[c]));
}

// M[:3,:4] = [R | R t] as vec12 (row-major 3x4) for Euler angles th and translation t.
__device__ inline void pose_matrix(const Axes ax, const float* th, const float* t, float* R, float* m12) {
    float R0[9], R1[9], R2[9], d[9], T[9];
    axis_rotation(ax.a[0], th[0], R0, d);
    axis_rotation(ax.a[1], th[1], R1, d);
    axis_rotation(ax.a[2], th[2], R2, d);
    mat3_mul(R0, R1, T);
    mat3_mul(T, R2, R);
    for (int r = 0; r < 3; ++r) {
        m12[r * 4 + 0] = R[r * 3 + 0];
        m12[r * 4 + 1] = R[r * 3 + 1];
        m12[r * 4 + 2] = R[r * 3 + 2];
        m12[r * 4 + 3] = fmaf(R[r * 3 + 2], t[2], fmaf(R[r * 3 + 1], t[1], R[r * 3] * t[0]));
    }
}

// g_m = G^T g_cam for pose b, computed by one wavefront: lane k < 12 owns column k of G (24 coalesced
// loads), lane r < 24 holds g_cam[r]; every lane ends up with all 12 entries.  `consume` zeroes g_cam.
__device__ inline void wave_gt_g(const float* __restrict__ G, float* g_cam_b, bool consume, float* gm) {
    const int lane = threadIdx.x & 63;
    float mine = 0.f;
    if (lane < 24) {
        mine = g_cam_b[lane];
        if (consume) g_cam_b[lane] = 0.f;
    }
    float col = 0.f;
    for (int r = 0; r < 24; ++r) {
        const float g = __shfl(mine, r);
        if (lane < 12) col = fmaf(G[r * 12 + lane], g, col);
    }
    for (int k = 0; k < 12; ++k) gm[k] = __shfl(col, k);
}

// d loss / d (th, t) from g_m = d loss / d vec(M[:3,:4]), through M = [R | R t].
__device__ inline void pose_chain(const Axes ax, const float* th, const float* t, const float* gm, float* g_th, float* g_t) {
    float R0[9], R1[9], R2[9], d0[9], d1[9], d2[9], T[9], R[9];
    axis_rotation(ax.a[0], th[0], R0, d0);
    axis_rotation(ax.a[1], th[1], R1, d1);
    axis_rotation(ax.a[2], th[2], R2, d2);
    mat3_mul(R0, R1, T);
    mat3_mul(T, R2, R);
    // gR = d/dR of <gm[:, :3], R> + <gm[:, 3], R t>
    float gR[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) gR[r * 3 + c] = fmaf(gm[r * 4 + 3], t[c], gm[r * 4 + c]);
    for (int c = 0; c < 3; ++c)
        g_t[c] = fmaf(R[6 + c], gm[11], fmaf(R[3 + c], gm[7], R[c] * gm[3]));   // R^T gT
    float dR[9], U[9];
    auto dot9 = [&](const float* A, const float* B) {
        float s = 0.f;
        for (int i = 0; i < 9; ++i) s = fmaf(A[i], B[i], s);
        return s;
    };
    mat3_mul(d0, R1, U); mat3_mul(U, R2, dR); g_th[0] = dot9(gR, dR);
    mat3_mul(R0, d1, U); mat3_mul(U, R2, dR); g_th[1] = dot9(gR, dR);
    mat3_mul(T, d2, dR);                      g_th[2] = dot9(gR, dR);
}

__global__ void k_pose_camera_fwd(const float* __restrict__ rot, const float* __restrict__ xyz, int B, Axes ax,
                                  const float* __restrict__ G, const float* __restrict__ c, float* __restrict__ cam) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float th[3] = {rot[b * 3], rot[b * 3 + 1], rot[b * 3 + 2]};
    float t[3] = {xyz[b * 3], xyz[b * 3 + 1], xyz[b * 3 + 2]};
    float R[9], m[12];
    pose_matrix(ax, th, t, R, m);
    for (int r = 0; r < 24; ++r) {
        float s = c[r];
        for (int k = 0; k < 12; ++k) s = fmaf(G[r * 12 + k], m[k], s);
        cam[b * 24 + r] = s;
    }
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
    for (int i = 0; i < 6; ++i) s.m[i] = s.v[i] = 0.f;
    s.lr[0] = lr_rot;
    s.lr[1] = lr_xyz;
    s.seen_lr = INFINITY;
    s.step = s.n_bad = s.n_plateaus = s.done = s.iter = 0;
    s.best = -INFINITY;
    st[b] = s;
}

__global__ __launch_bounds__(64) void k_pose_opt_step(float* __restrict__ rot, float* __restrict__ xyz, int B, xvr_pose_opt_spec sp,
                                const float* __restrict__ G, float* __restrict__ g_cam, const float* __restrict__ loss,
                                xvr_pose_opt_state* __restrict__ state, float* __restrict__ history) {
    const int b = blockIdx.x;   // one wavefront per pose; every lane carries the scalars, lane 0 writes
    float gm[12];
    wave_gt_g(G, g_cam + (size_t)b * 24, true, gm);   // consumed: the next rays-backward accumulates from zero
    xvr_pose_opt_state s = state[b];
    if (s.done) return;
    Axes ax = {{sp.axes[0], sp.axes[1], sp.axes[2]}};
    float p[6] = {rot[b * 3], rot[b * 3 + 1], rot[b * 3 + 2], xyz[b * 3], xyz[b * 3 + 1], xyz[b * 3 + 2]};
    float g[6];
    pose_chain(ax, p, p + 3, gm, g, g + 3);
    if (threadIdx.x != 0) return;   // (after the loads above: every lane read the same, unmodified state)

    // Adam, in the operation order of torch.optim.Adam(capturable=True)
    s.step += 1;
    const float stepf = (float)s.step;
    const float bc1 = 1.f - powf(sp.beta1, stepf);
    const float bc2_sqrt = sqrtf(1.f - powf(sp.beta2, stepf));
    for (int i = 0; i < 6; ++i) {
        const float gi = sp.maximize ? -g[i] : g[i];
        s.m[i] = s.m[i] + (gi - s.m[i]) * (1.f - sp.beta1);                 // lerp_
        s.v[i] = fmaf(gi * gi, 1.f - sp.beta2, s.v[i] * sp.beta2);         // mul_().addcmul_()
        const float step_size_neg = -(s.lr[i / 3] / bc1);
        const float denom = sqrtf(s.v[i]) / (bc2_sqrt * step_size_neg) + sp.eps / step_size_neg;
        p[i] += s.m[i] / denom;                                            // addcdiv_
    }
    for (int i = 0; i < 3; ++i) {
        rot[b * 3 + i] = p[i];
        xyz[b * 3 + i] = p[3 + i];
    }

    // ReduceLROnPlateau(mode="max", threshold_mode="rel", cooldown=0, min_lr=0), in double like the host version
    const double cur = (double)loss[b];
    if (cur > s.best * (sp.threshold + 1.0)) {
        s.best = cur;
        s.n_bad = 0;
    } else {
        s.n_bad += 1;
    }
    if (s.n_bad > sp.patience) {
        for (int k = 0; k < 2; ++k) {
            const double old_lr = (double)s.lr[k];
            const double new_lr = fmax(old_lr * (double)sp.factor, 0.0);
            if (old_lr - new_lr > sp.lr_eps) s.lr[k] = (float)new_lr;
        }
        s.n_bad = 0;
    }
    // stopping rule of the reference loop: count the learning-rate levels (the first one included)
    if (s.lr[0] < s.seen_lr) {
        s.seen_lr = s.lr[0];
        s.n_plateaus += 1;
    }
    if (s.n_plateaus == sp.max_n_plateaus) s.done = 1;
    if (history && s.iter < sp.max_iters) {
        float* h = history + ((size_t)b * sp.max_iters + s.iter) * XVR_POSE_HISTORY_COLS;
        for (int i = 0; i < 6; ++i) h[i] = p[i];
        h[6] = loss[b];
        h[7] = s.lr[0];
        h[8] = s.lr[1];
    }
    s.iter += 1;
    state[b] = s;
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

extern "C" int xvr_pose_opt_step(float* rot, float* xyz, int B, const xvr_pose_opt_spec* spec, const float* G,
                                 float* grad_cam, const float* loss, xvr_pose_opt_state* state, float* history,
                                 void* stream) {
    if (!rot || !xyz || !spec || !G || !grad_cam || !loss || !state || B <= 0) return pfail(XVR_DRR_E_ARG, "bad argument");
    if (!axes_ok(spec->axes)) return pfail(XVR_DRR_E_ARG, "axes must be in {0,1,2} with the middle one distinct from its neighbours");
    if (spec->max_n_plateaus < 1 || spec->patience < 0 || (history && spec->max_iters < 1))
        return pfail(XVR_DRR_E_ARG, "bad optimiser spec");
    hipLaunchKernelGGL(k_pose_opt_step, dim3(B), dim3(64), 0, (hipStream_t)stream, rot, xyz, B, *spec, G,
                       grad_cam, loss, state, history);
    return launched("pose_opt_step");
}
