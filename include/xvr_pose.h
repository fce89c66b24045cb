/*
 * xvr_pose.h -- C ABI of the pose side of xvr's registration loop (part of libxvr_drr.so).
 *
 * Every iteration of `_RegistrarBase.run_test_time_optimization`
 * (/root/reference/src/xvr/registrar/base.py:245-280) wraps the render in three pieces of pose arithmetic
 * that the reference runs as ~200 tiny framework launches:
 *
 *   reg()                 rot, xyz -> convert(...) -> detector / affine_inverse     base.py:249
 *   loss.backward()       ... and the chain back to rot.grad / xyz.grad             base.py:252
 *   optimizer.step()      Adam(maximize=True) on the two parameter groups           base.py:221-228,253
 *   scheduler.step(loss)  ReduceLROnPlateau(factor, patience, threshold, mode=max)  base.py:229-235,262
 *   n_plateaus / break    stop after max_n_plateaus learning-rate drops             base.py:270-278
 *
 * Here they are two launches per iteration:
 *   xvr_pose_camera_forward  Euler angles + translation -> the camera vector cam[24] that
 *                            xvr_drr_rays_forward consumes.  The map pose matrix -> cam is affine
 *                            (calibration, reorientation and the CT's inverse affine are constants of a
 *                            pyramid stage), so it is handed over as G[24][12], c[24]:
 *                            cam = G * vec(M[:3,:4]) + c with M = [R | R t], R = R_a0 R_a1 R_a2.
 *   xvr_pose_opt_step        the chain rule cam -> (rot, xyz), the Adam update, the plateau scheduler and
 *                            the stopping rule, with ALL optimiser state on the device.  Once a pose has
 *                            met the stopping rule the call is a no-op for it, so a host may enqueue (or
 *                            graph-replay) several iterations without a device->host sync in between and
 *                            still obtain exactly the trajectory of a loop that checks after every step.
 *
 * All pointers are device pointers (fp32 unless noted); kernels are enqueued on `stream`; return value 0
 * or a negative XVR_DRR_E_* code (xvr_drr_last_error() has the text).
 */
#ifndef XVR_POSE_H
#define XVR_POSE_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XVR_POSE_MAX_PARAMS 13     /* 10 rotation numbers (quaternion adjugate) + 3 of the translation */

/* Per-pose optimiser state, resident in device memory (one element per pose of the batch). */
typedef struct xvr_pose_opt_state {
    float m[XVR_POSE_MAX_PARAMS], v[XVR_POSE_MAX_PARAMS];   /* Adam first / second moments of (rotation parameters, tx, ty, tz) */
    float lr[2];         /* current learning rates (rotation, translation)                               */
    float seen_lr;       /* smallest rotation lr counted so far (the reference's `current_lr`)           */
    int   step;          /* Adam step count                                                              */
    int   n_bad;         /* ReduceLROnPlateau.num_bad_epochs                                             */
    int   n_plateaus;    /* learning-rate levels seen, the initial one included                          */
    int   done;          /* 1 once n_plateaus == max_n_plateaus: later steps leave this pose untouched  */
    int   iter;          /* iterations taken = rows of `history` written                                 */
    double best;         /* ReduceLROnPlateau.best                                                       */
} xvr_pose_opt_state;

typedef struct xvr_pose_opt_spec {
    int   axes[3];            /* Euler convention, 0 = X, 1 = Y, 2 = Z  ("ZXY" -> {2, 0, 1})             */
    float beta1, beta2, eps;  /* Adam                                    (0.9, 0.999, 1e-8)              */
    int   maximize;           /* 1: ascend (the reference maximises the similarity)                      */
    float factor;             /* ReduceLROnPlateau.factor                (0.1)                           */
    int   patience;           /* ReduceLROnPlateau.patience                                              */
    double threshold;         /* relative improvement threshold, mode = max  (1e-4)                      */
    double lr_eps;            /* minimal lr decrease that is applied      (1e-8)                         */
    int   max_n_plateaus;     /* stopping rule                                                           */
    int   max_iters;          /* rows per pose in `history`                                              */
} xvr_pose_opt_spec;

#define XVR_POSE_HISTORY_COLS 9   /* Euler: r0 r1 r2 tx ty tz (after the update), loss (before it), lr_rot, lr_xyz (after);
                                     a parameterisation of k rotation numbers: k + 6 columns, same order                  */

/* rot [B][3], xyz [B][3], G [24][12], c [24] -> cam [B][24] */
int xvr_pose_camera_forward(const float* rot, const float* xyz, int B, const int axes[3], const float* G,
                            const float* c, float* cam, void* stream);

/* grad_cam [B][24] -> grad_rot [B][3], grad_xyz [B][3] (written) */
int xvr_pose_camera_backward(const float* rot, const float* xyz, int B, const int axes[3], const float* G,
                             const float* grad_cam, float* grad_rot, float* grad_xyz, void* stream);

/* sizeof(xvr_pose_opt_state) as the library was compiled (144): lets a binding verify its own layout */
size_t xvr_pose_opt_state_bytes(void);

/* reset the state of B poses: zero moments, step 0, lr = (lr_rot, lr_xyz), best = -inf */
int xvr_pose_opt_init(xvr_pose_opt_state* state, int B, float lr_rot, float lr_xyz, void* stream);

/*
 * One optimiser iteration for every pose that is not done:
 *   grad_cam [B][24]  d (sum of losses) / d cam; CONSUMED: reset to zero for the next accumulation
 *   loss     [B]      the similarity the gradient belongs to (fed to the plateau scheduler)
 *   rot, xyz          updated in place
 *   history  [B][max_iters][XVR_POSE_HISTORY_COLS]  row state.iter is written (nullable)
 */
int xvr_pose_opt_step(float* rot, float* xyz, int B, const xvr_pose_opt_spec* spec, const float* G,
                      float* grad_cam, const float* loss, xvr_pose_opt_state* state, float* history,
                      void* stream);

/*
 * The same two steps for ANY parameterisation xvr_pose_convert_forward knows (the reference's Registration takes them all,
 * /root/reference/src/xvr/registrar/base.py:168-169,221-235): `kind` as there, rot [B][k].
 *   xvr_pose_camera_forward_param   rot, xyz -> cam [B][24] in one launch, and the 12 x (k + 3) Jacobian of the pose matrix
 *                                   (jac: xvr_pose_convert_jacobian_floats(B) floats) for the step below
 *   xvr_pose_opt_step_param         g = J^T G^T grad_cam, Adam with lr_rot on the k rotation numbers and lr_xyz on the
 *                                   translation, scheduler, stopping rule; history rows of k + 6 columns.  kind = 0 ignores jac
 *                                   and is xvr_pose_opt_step.
 */
int xvr_pose_camera_forward_param(const float* rot, const float* xyz, int B, int kind, const int axes[3], const float* G,
                                  const float* c, float* cam, float* jac, void* stream);
int xvr_pose_opt_step_param(float* rot, float* xyz, int B, int kind, const xvr_pose_opt_spec* spec, const float* G, const float* jac,
                            float* grad_cam, const float* loss, xvr_pose_opt_state* state, float* history, void* stream);

/*
 * Pose terms of the training loss (/root/reference/src/xvr/model/loss.py:27-48), which the reference evaluates
 * with ~200 tiny framework launches per step.  Poses are row-major 4x4 matrices ([N][16]); sdd in mm.
 *   xvr_pose_geodesic            DoubleGeodesicSE3(sdd)(a, b): out [3][N] = (angular, translational, double), all mm;
 *                                grad_b [N][12] (nullable) = d double_n / d (R | t of b_n)
 *   xvr_pose_multiview_forward   mvc [B (B - 1) / 2] = double geodesic between the true and the predicted relative
 *                                pose b_j b_i^{-1} of every pair i < j, in torch.triu_indices(B, B, 1) order
 *   xvr_pose_multiview_backward  grad_pred [B][12] = sum over pairs of grad_mvc[pair] * d mvc[pair] / d pred pose,
 *                                added in a fixed order (no atomics)
 */
int xvr_pose_geodesic(const float* a, const float* b, int N, float sdd, float eps, float* out, float* grad_b, void* stream);
int xvr_pose_multiview_forward(const float* true_pose, const float* pred_pose, int B, float sdd, float eps, float* mvc,
                               void* stream);
int xvr_pose_multiview_backward(const float* true_pose, const float* pred_pose, const float* grad_mvc, int B, float sdd,
                                float eps, float* grad_pred, void* stream);

/*
 * diffdrr's convert(rotation, translation, parameterization=..., convention=...) -> 4x4 pose and its backward, one launch
 * each (/root/reference/src/xvr/model/network.py:49-56 on the regressor's output every training step; the registrar with
 * non-Euler parameterisations, registrar/base.py:168): the framework's 40-110 tiny launches cost 0.8-2.0 ms at 116 poses.
 *   kind   0 euler_angles (axes = the convention, 0 = X .. 2 = Z; radians)   1 axis_angle   2 quaternion (real first)
 *          3 quaternion_adjugate (10 numbers)   4 rotation_6d   5 se3_log_map   6 rotation_10d (the upper triangle of a symmetric
 *          4 x 4 whose smallest eigenvector is the quaternion: Jacobi iteration per pose)   rot [B][3 | 3 | 4 | 10 | 6 | 3 | 10]
 *   matrix [B][16] row-major;  jac: xvr_pose_convert_jacobian_floats(B) floats, written by the forward (the 12 x (k + 3)
 *          Jacobian of every pose, by forward-mode differentiation of the same formulas) and read by the backward
 *   backward: grad_matrix [B][16] (row 3 ignored) -> grad_rot [B][k], grad_xyz [B][3]
 */
size_t xvr_pose_convert_jacobian_floats(int B);
int xvr_pose_convert_forward(const float* rot, const float* xyz, int B, int kind, const int axes[3], float* matrix, float* jac,
                             void* stream);
int xvr_pose_convert_backward(const float* jac, const float* grad_matrix, int B, int kind, float* grad_rot, float* grad_xyz,
                              void* stream);

#ifdef __cplusplus
}
#endif
#endif /* XVR_POSE_H */
