/*
 * xvr_sim.h -- C ABI of the fused image-similarity step of xvr's registration loop (part of
 * libxvr_drr.so).  SURVEY.md section 8f rank 1: the step right after the renderer in every iteration,
 *
 *     pred_img = transform(pred_img)              /root/reference/src/xvr/registrar/base.py:250
 *     loss = imagesim(img, pred_img)              /root/reference/src/xvr/registrar/base.py:251
 *     loss.backward()                             /root/reference/src/xvr/registrar/base.py:252
 *
 * with transform = XrayTransforms (Standardize by the GLOBAL min/max -> Normalize(mean, std);
 * /root/reference/src/xvr/utils/preprocess.py:5-31, the Resize is the identity for a DRR rendered at the
 * stage's resolution) and imagesim = beta * MultiscaleNCC([None, p1], [.5, .5]) + (1 - beta) *
 * GradientNCC(p2, sigma = 0)  (/root/reference/src/xvr/registrar/base.py:115-123).
 *
 * One call computes the similarity of every image of the batch AND its exact gradient w.r.t. the raw
 * (un-transformed) moving image, including the terms through the global min / max of Standardize.
 * A term whose weight is zero is not computed: beta = 1 is the multiscale NCC alone (the training loss,
 * /root/reference/src/xvr/model/loss.py:16,27 -- fixed_sobel is then unused and may alias fixed).
 * All pointers are device pointers; kernels are enqueued on `stream`; returns 0 or a negative
 * XVR_DRR_E_* code (message via xvr_drr_last_error()).
 */
#ifndef XVR_SIM_H
#define XVR_SIM_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct xvr_sim_spec {
    float mean, std;      /* Normalize(mean, std) after Standardize            (0.15, 0.1)          */
    float std_eps;        /* Standardize: (x - min) / (max - min + std_eps)     (1e-6)               */
    float ncc_eps;        /* added to every variance                            (1e-5)               */
    float beta;           /* weight of the multiscale NCC vs the gradient NCC   (0.5)                */
    int   mncc_patch;     /* patch size of the local term of the multiscale NCC (9), in [2, 15]      */
    int   gncc_patch;     /* patch size of the gradient NCC                     (11), in [2, 15]     */
    int   per_image;      /* 0: Standardize by the min/max of the WHOLE batch tensor, as the reference's
                             transform does (identical for B = 1); 1: by each image's own min/max, so that
                             the images of a batch are independent problems (batched multi-start)        */
    int   pre_transformed;/* 1: `moving` is already transformed (y = moving; mean / std / std_eps unused, no
                             min/max terms in the gradient): the training loss, which transforms both images
                             before it calls the similarity (/root/reference/src/xvr/model/trainer.py:215-218)  */
} xvr_sim_spec;

/* bytes of device scratch for a batch of B images of H x W */
size_t xvr_sim_workspace_bytes(int B, int H, int W);

/*
 * fixed        [B][H][W]     the target image, ALREADY transformed (constant over the iterations)
 * fixed_sobel  [B][2][H][W]  its Sobel pair (conv2d with the 3x3 Sobel kernels, zero padding 1)
 * moving       [B][H][W]     the raw rendered DRRs
 * loss         [B]           similarity of every image (written)
 * grad_moving  [B][H][W]     d loss[b] / d moving (written; nullable)
 */
int xvr_sim_ncc_forward_backward(const float* fixed, const float* fixed_sobel, const float* moving,
                                 int B, int H, int W, const xvr_sim_spec* spec,
                                 float* loss, float* grad_moving,
                                 void* workspace, size_t workspace_bytes, void* stream);

/*
 * One registration iteration's tail behind the render, in the similarity's own launches (ABI 10; replaces, for Euler angles,
 * the sequence of /root/reference/src/xvr/registrar/base.py:250-262 -- transform, imagesim, loss.backward(), optimizer.step(),
 * scheduler.step(loss), the stopping rule -- and the first line of the NEXT iteration, reg() -> camera, base.py:249):
 *
 *     xvr_sim_ncc_forward_backward  +  xvr_drr_jac_to_camera_backward  +  xvr_pose_opt_step  +  xvr_pose_camera_forward
 *
 * The similarity's last kernel also contracts the image gradient with the render's per-ray jacobian, and the block that
 * finishes last for a pose takes that pose's optimiser step and writes the camera vector of the updated pose.  Same
 * expressions, same reduction order: rot, xyz, state, history and cam are bit for bit those of the four separate calls
 * (tests/test_pose_opt.py).  Three launches fewer per iteration; the image gradient is not stored.
 *
 *   fixed, fixed_sobel, moving, spec (pre_transformed = 0), loss, workspace   as xvr_sim_ncc_forward_backward
 *   grad_scratch [B][H][W]   scratch for the image gradient before its min / max terms (contents undefined afterwards)
 *   jac  [B][H W][XVR_DRR_JAC_STRIDE]   the forward's per-ray jacobian (xvr_drr_*_forward_camera)
 *   cam  [B][24]   IN: the camera vector the render used;  OUT: that of the updated pose (unchanged for a pose that is done)
 *   j2c_workspace   xvr_drr_jac_to_camera_workspace_bytes(B, H, W) bytes, zero-filled once (xvr_drr_jac_to_camera_backward's)
 *   rot, xyz [B][3], opt_spec, G [24][12], c [24], state [B], history   as xvr_pose_opt_step / xvr_pose_camera_forward
 *   workspace_armed   0: `workspace` may hold anything (its header and tickets are reset first, one more launch); 1: the caller
 *                     vouches that the LAST thing that ran on `workspace` was a complete xvr_sim_ncc_* call (tickets back at
 *                     zero) -- the min / max reduction then writes the header itself and the reset is not launched
 */
struct xvr_pose_opt_spec;
struct xvr_pose_opt_state;
int xvr_sim_ncc_registration_step(const float* fixed, const float* fixed_sobel, const float* moving, int B, int H, int W,
                                  const xvr_sim_spec* spec, float* loss, float* grad_scratch, void* workspace, size_t workspace_bytes,
                                  const float* jac, float* cam, void* j2c_workspace, size_t j2c_workspace_bytes,
                                  float* rot, float* xyz, const struct xvr_pose_opt_spec* opt_spec, const float* G, const float* c,
                                  struct xvr_pose_opt_state* state, float* history, int workspace_armed, void* stream);

/*
 * The Gaussian pre-blur of GradientNormalizedCrossCorrelation2d(patch, sigma > 0)
 * (/root/reference/src/xvr/registrar/base.py:122): 5 taps exp(-x^2 / (2 sigma^2)), x = -2..2, normalised, separable,
 * reflect padding by 2 -- and its exact transpose (adjoint = 1), which is the blur's backward.  The similarity of a
 * configuration with sigma > 0 is then  beta * mNCC(fixed, y) + (1 - beta) * gNCC(blur(fixed), blur(y))  with both NCC
 * terms from xvr_sim_ncc_forward_backward (beta = 1 / beta = 0, pre_transformed) and the second gradient passed back
 * through the adjoint.
 *   in, out, scratch  [B][H][W] each (out may alias in; scratch may not), H, W >= 3
 */
int xvr_sim_gaussian_blur5(const float* in, float* out, float* scratch, int B, int H, int W, float sigma, int adjoint, void* stream);

/*
 * Equalize (/root/reference/src/xvr/utils/preprocess.py:34-66): the differentiable soft-histogram equalisation xvr's
 * XrayTransforms applies between Standardize and Normalize when `equalize` is set -- forward and exact backward, per image,
 * without the reference's [pixels x bins] weight matrix, sums in fixed order.
 *   x, y, S, grad_out, grad_x  [B][n]   (x in [0, 1]: the standardised image; S = per-pixel weight sums the backward reuses)
 *   n_bins in [2, 1024] (256), tau (0.01), eps (1e-10)
 *   y_out (nullable) = (y - out_mean) / out_std: the Normalize that follows Equalize in XrayTransforms, written in the same
 *   pass; the backward takes grad_out = the gradient w.r.t. y_out (out_std = 1: w.r.t. y itself)
 *   workspace  xvr_sim_equalize_workspace_bytes(B, n_bins); the backward needs it as the forward left it
 */
size_t xvr_sim_equalize_workspace_bytes(int B, int n_bins);
int xvr_sim_equalize_forward(const float* x, int B, int n, int n_bins, float tau, float eps, float out_mean, float out_std, float* y,
                             float* S, float* y_out, void* workspace, size_t workspace_bytes, void* stream);
int xvr_sim_equalize_backward(const float* x, const float* y, const float* S, const float* grad_out, int B, int n, int n_bins,
                              float tau, float eps, float out_std, float* grad_x, void* workspace, size_t workspace_bytes, void* stream);

/*
 * XrayTransforms without Equalize and without a Resize -- Standardize then Normalize,
 *   y = ((x - lo) / (hi - lo + eps) - mean) / std,   lo / hi = min / max over the whole [B][n] tensor (per_image = 0, the
 *   reference's transform, /root/reference/src/xvr/utils/preprocess.py:5-31) or over each image (per_image = 1)
 * -- and its backward, which includes the gradient through lo and hi (torch's rule: shared evenly by the pixels that attain
 * them).  The trainer applies it to both renders of every step (trainer.py:207,216).  `state`: xvr_sim_transform_state_bytes(B)
 * bytes, written by the forward and needed by the backward, which writes its two sums into it (the forward's min / max / tie
 * counts stay; the backward may run any number of times over one forward).  y is bit-identical to the torch expression; the
 * backward's two sums are added in a fixed order (doubles).
 */
size_t xvr_sim_transform_state_bytes(int B);
int xvr_sim_transform_forward(const float* x, int B, long long n, int per_image, float mean, float std, float eps, float* y, void* state,
                              void* stream);
int xvr_sim_transform_backward(const float* x, const float* grad_y, int B, long long n, int per_image, float mean, float std, float eps,
                               float* grad_x, void* state, void* stream);

/*
 * The in-tree DiceMetric of /root/reference/src/xvr/model/loss.py:5-40 on two BOOLEAN label maps [B][C][n] (one byte per
 * pixel, as torch.bool; the masks Trainer.render_samples returns): dice[b][c] = 2 |pred & truth| / (|pred| + |truth|) from
 * integer counts -- bit-identical to the reference's float sums of 0 / 1 (n <= 2^24), NaN for 0 / 0 as there.  The caller
 * drops channel 0 (background) and takes 1 - nanmean as DiceLoss does.
 */
int xvr_sim_dice_bool(const unsigned char* pred, const unsigned char* truth, int B, int C, int n, float* dice, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* XVR_SIM_H */
