/*
 * xvr_drr.h -- C ABI of the MI355X (gfx950) differentiable-DRR render library, libxvr_drr.so.
 *
 * This is the drop-in boundary for the one hot path this project accelerates: the renderer call
 * that xvr makes through diffdrr (the reference is pure Python and has no FFI of its own, so each
 * entry point cites the Python interface it replaces):
 *
 *     img = drr.renderer(volume, source, target, img, mask=seg)      -> [B, C, n]
 *         /root/reference/src/xvr/model/trainer.py:288   (training, called directly)
 *         /root/reference/src/xvr/registrar/base.py:249  (registration, via Registration -> DRR.forward)
 *     loss.backward() through that call
 *         /root/reference/src/xvr/model/trainer.py:223, /root/reference/src/xvr/registrar/base.py:252
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 (caller-owned, contiguous, not retained past the call);
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued on it, nothing synchronises;
 *   - coordinates are voxel-index space (the caller already applied the CT's inverse affine,
 *     trainer.py:285); `raylen` is the world-mm length of each ray computed BEFORE that affine
 *     (trainer.py:284);
 *   - layouts: volume/mask [D0][D1][D2] (D2 fastest; coordinate axis i indexes volume axis i),
 *     source [B][3] (one per pose; the reference's [B,1,3]), target [B][n][3], raylen [B][n],
 *     out / grad_out [B][C][n], C = 1 without a mask, else max(label)+1;
 *   - return value: 0 = ok, negative = error (XVR_DRR_E_*); xvr_drr_last_error() gives the text.
 *     Nothing aborts: xvr's trainer swallows exceptions per step and continues (trainer.py:171-175).
 */
#ifndef XVR_DRR_H
#define XVR_DRR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XVR_DRR_ABI_VERSION 10   /* 10: xvr_sim_ncc_registration_step (the tail of a registration iteration in the similarity's launches); 9: xvr_drr_pack_hu_labels_ytiles, options siddon_splat / gather_splat = 3 / siddon_slab = 2, pose kind 6 (rotation_10d), non-exact Siddon index maps on the slab march and the brick splat, guard-banded fixed point; 8: volume_layout 3 + xvr_drr_pack_ytiles / _labels_ytiles (tiled y-pair copy), xvr_pose_camera_forward_param / xvr_pose_opt_step_param (device-resident registration for every parameterisation; xvr_pose_opt_state holds 13 parameters), xvr_sim_equalize_* write / take the normalised output, options tile_geom, siddon_slab, siddon_gather_fast; 7: xvr_drr_foreground, xvr_drr_pack_labels_ypairs, xvr_sim_dice_bool, xvr_sim_transform_* (xvr_sim.h), xvr_pose_convert_* (xvr_pose.h); 6: xvr_drr_spec.alpha_window + xvr_drr_alpha_window (clip_to_volume = 2); 5: xvr_drr_set_option / xvr_drr_get_option (the A/B switches are no longer getenv calls per launch); 4: volume_layout 2 + xvr_drr_pack_bricks (siddon forward); 3: xvr_drr_spec.volume_layout + xvr_drr_pack_ypairs; 2: xvr_sim_spec grew, camera-driven forwards, packed labels, xvr_pose.h */

#define XVR_DRR_OK 0
#define XVR_DRR_E_ARG (-1)     /* bad argument (null pointer, non-positive size, unsupported combo) */
#define XVR_DRR_E_LAUNCH (-2)  /* hipLaunch / hipMemsetAsync failed; see xvr_drr_last_error()       */
#define XVR_DRR_E_UNSUPPORTED (-3)

#define XVR_DRR_JAC_STRIDE 8 /* floats per ray in the jacobian buffer */

/*
 * Numerical convention of a render (every constant that could not be pinned against
 * diffdrr==0.6.0 is explicit; see SURVEY.md Appendix A and DESIGN.md "Semantics").
 */
typedef struct xvr_drr_spec {
    float a[3], b[3];      /* sampling index along axis i = a[i] * x + b[i]  (grid_sample's un-normalisation
                              of u = 2 (x + voxel_shift) / dims - 1)                                       */
    float lo[3], hi[3];    /* bounding planes of the volume in x: -voxel_shift, shape - voxel_shift        */
    float plane0[3];       /* siddon: plane i of axis k sits at x = i + plane0[k]  (= -voxel_shift)         */
    float eps;             /* added to (target - source)                                                   */
    /* trilinear */
    int32_t n_points;      /* samples per ray                                                              */
    float near_, far_;     /* alphas = linspace(near, far, n_points)                                       */
    float inv_denom;       /* out = raylen * sum * inv_denom   (1/n_points or 1/(n_points-1))              */
    int32_t clip_to_volume;/* 0: alphas span source->target; 1: rescaled per ray to [alphamin, alphamax];
                              2: ONE window for the whole call -- alphas = A + linspace(near, far) (Z - A), A / Z the
                              smallest alphamin / largest alphamax over the call's rays that meet the volume, the
                              image scaled by (Z - A); needs `alpha_window` (below)                          */
    /* launch shaping (performance only, never changes results) */
    int32_t ray_grid_w;    /* >0: the n rays form an (n / ray_grid_w) x ray_grid_w row-major detector and
                              lanes are mapped to 8x8 pixel tiles; 0: rays are mapped linearly             */
    int32_t volume_layout; /* forward only: 0 = `volume` is [D0][D1][D2]; 1 (trilinear) = it is the y-pair interleaved
                              copy written by xvr_drr_pack_ypairs; 2 (siddon) = the 2 x 2 x 8 bricks written by
                              xvr_drr_pack_bricks; 3 (trilinear) = the y-pair copy in 2 x 8 tiles written by
                              xvr_drr_pack_ytiles (same results, bit for bit)                               */
    const float* alpha_window; /* clip_to_volume == 2: device buffer of xvr_drr_alpha_window_bytes(B) bytes, 16-byte aligned, written by
                              xvr_drr_alpha_window() on the same stream before the render (the kernels read the call's
                              near / far / scale from it: no host round trip); NULL otherwise                  */
} xvr_drr_spec;

#define XVR_DRR_ALPHA_WINDOW_FLOATS 16   /* header of the window buffer; xvr_drr_alpha_window_bytes(B) is its whole size */

int xvr_drr_abi_version(void);
const char* xvr_drr_last_error(void);

/*
 * A/B switches of the launch logic -- every setting selects among CORRECT kernels (same results up to the documented
 * rounding); they exist for measurements and for tests that compare the alternatives inside one process.  The table is
 * read ONCE from the environment when the library is loaded (XVR_DRR_<NAME>, upper case) and changed afterwards only through
 * xvr_drr_set_option; no entry point calls getenv.  Names and values:
 *   "fwd_lds"       0 | 1      1: the LDS-staged trilinear forward (slower; natural volume layout only)          [0]
 *   "tile_shape"    -1 | 0-2   pixels a wavefront takes out of a 16x16 tile: per workgroup | 8x8 | 16x4 | 4x16   [-1]
 *   "block_order"   -1 | 0-4   logical block -> (pose, tile) map; -1: by batch size                              [-1]
 *   "order_group"   0 | gx + 256 gy   tiles per group of the grouped block orders; 0: full-width strips          [0]
 *   "fwd_split"     0 | n | 100 + n   sample slices per ray of small forwards: measured table | 8x8 tiles x n | 16x16 tiles x n [0]
 *   "gather_splat"  1 | 0 | 2 | 3  trilinear voxel gradient: brick-local fixed-point splat, except launches with more than ~48
 *                              samples of a pose per voxel (counted on the device), which take the fp32 gather | fp32 voxel-driven
 *                              gathers | the ray-major splat (clip_to_volume = 1 and per-channel masks always use it) for every
 *                              render (A/B) | the splat whatever the sampling density                               [1]
 *   "fwd_slabs"     0 | -1 | n  slab-major trilinear forward (one launch per slab of the volume, all poses; measured SLOWER,
 *                              HISTORY.md 4.4): never | by size | n slabs                                         [0]
 *   "fwd_slab_axis" 0-2        volume axis the slabs are cut along                                               [1]
 *   "tile_geom"     1 | 0 | 2  pixels a workgroup takes of a detector that is a multiple of 64: 8 x 32 | 16 x 16 | 4 x 64, long side
 *                              along the detector axis that runs along the volume's contiguous axis (per pose)     [1]
 *   "siddon_slab"   1 | 0 | 2  one-channel Siddon forward: dominant-axis slab march (k_siddon_slab, both volume layouts) for the
 *                              exact index map (unsplit launches) and for non-exact maps that keep the volume's points inside it
 *                              (norm_dims_offset = +1, align_corners; every launch size) | the merge walk (k_siddon) | the march for
 *                              the exact map only                                                                   [1]
 *   "siddon_splat"  1 | 0 | 2  Siddon voxel gradient under a non-exact index map: ray-driven brick-local fixed-point splat
 *                              (k_siddon_splat; shares the march's plane alphas and index arithmetic; one scale per 16^3 brick;
 *                              detectors up to 2^25 pixels, < 32768 rows and columns -- larger ones take the gather) | the
 *                              per-cell fp32 gather (needs the larger workspace) | the splat for the exact map as well (A/B) [1]
 *   "siddon_gather_fast" 1 | 0 Siddon voxel gather: pixel window from one projection of the block centre, four bricks along the
 *                              viewing axis per workgroup, candidates from an LDS copy of the brick's footprint, planes in
 *                              crossing order | the window of the eight projected corners, one brick per workgroup,
 *                              candidates from global memory (A/B: equal to rounding)                           [1]
 *   "gather_slab"   0 | index + 256 * count   the voxel gradient of xvr_drr_*_backward in `count` slabs of whole 16^3-brick planes
 *                              along x, ONE backward call per slab: index 0 first, same arguments and workspace.  Call i adds the
 *                              gradient of the voxels x in [16 * (i * nb / count), 16 * ((i + 1) * nb / count)), nb = ceil(D0 / 16);
 *                              pose gradients, and renders the brick splats do not serve (the whole volume), come with call 0.
 *                              A caller can hand slab i to a collective while slab i + 1 is computed (xvr_amd.distributed
 *                              .SlabAllReduce).  The caller resets the option to 0 afterwards.                      [0]
 * Returns XVR_DRR_E_ARG for an unknown name or a value outside the option's range.
 */
int xvr_drr_set_option(const char* name, int value);
int xvr_drr_get_option(const char* name, int* value);

/*
 * clip_to_volume == 2 (trilinear; a third plausible reading of the alpha rule of diffdrr's Trilinear.forward, reached from
 * /root/reference/src/xvr/model/trainer.py:288 -- SURVEY.md Appendix A marks it "uncertain"): reduce, on the device, the rays
 * of the call to its alpha window and leave { A, Z, near', far', inv_denom', Z - A, ... } in `window`; `spec` is the render's
 * spec (its alpha_window field is ignored here).  The backward of the window itself -- min / max route their gradient to
 * the two extremal rays -- is xvr_drr_alpha_window_backward: call it after xvr_drr_backward_from_jac with the same jacobian
 * and upstream gradient; it ADDS to grad_source / grad_target.  One channel only.
 */
size_t xvr_drr_alpha_window_bytes(int B);
int xvr_drr_alpha_window(const float* source, const float* target, int B, int n, int D0, int D1, int D2,
                         const xvr_drr_spec* spec, float* window, void* stream);
int xvr_drr_alpha_window_backward(const float* jac, const float* grad_out, const float* source, const float* target,
                                  const float* raylen, int B, int n, const xvr_drr_spec* spec, float* window,
                                  float* grad_source, float* grad_target, void* stream);

/*
 * The tail of xvr's render_samples, /root/reference/src/xvr/model/trainer.py:289-304, over a rendered batch img [B][C][n]:
 *   mask [B][C][n] (bytes, 0 / 1) = img > 0;   sum [B][n] = sum over the channels;
 *   keep [B] (bytes) = mean over the pixels of (C == 1 ? mask[0] : any(mask[1:])) > threshold
 * (threshold = the trainer's img_threshold 0.10 for C == 1, mask_threshold 0.05 otherwise).  `count` = B ints of scratch
 * (on return: the foreground pixels per pose).  One pass over the image instead of five torch launches.
 */
int xvr_drr_foreground(const float* img, int B, int C, int n, float threshold, float* sum, unsigned char* mask, int* count,
                       unsigned char* keep, void* stream);

/* Bytes of device scratch the backward entry points can use (see `workspace` below). */
size_t xvr_drr_backward_workspace_bytes(int B, int n, int D0, int D1, int D2);
/* The same for xvr_drr_siddon_backward under `spec`: a NON-exact index map (norm_dims_offset != 0, align_corners) gathers
 * per plane cell into eight sums first and needs 32 bytes per voxel more; with only the smaller size such a spec falls
 * back to the atomic scatter. */
size_t xvr_drr_siddon_backward_workspace_bytes(int B, int n, int D0, int D1, int D2, const xvr_drr_spec* spec);

/*
 * Trilinear ray-marching forward.  Replaces Trilinear.forward(volume, source, target, img, mask=...).
 *   mask     nullable; float labels, same shape as volume.  NULL with C in [2, 16]: the labels are packed
 *            into `volume` (xvr_drr_pack_labels below).
 *   jac      nullable [B][n][8].  Per ray: {out / raylen, d out/d source[3], d out/d target[3], 0},
 *            where with a mask `out` means the SUM over channels.  Filled in the same sweep (no
 *            second gather), consumed by xvr_drr_backward_from_jac -- which is the whole pose-side
 *            backward whenever grad_out is the same for every channel (C == 1, or the channels were
 *            only summed downstream, as xvr's trainer does: src/xvr/model/trainer.py:292-293).
 *   work     nullable device uint64: incremented by the number of samples that touched the volume
 *            (the kernel's own count of algorithmic work, for the roofline).
 */
int xvr_drr_trilinear_forward(const float* volume, const float* mask, int D0, int D1, int D2, int C,
                              const float* source, const float* target, const float* raylen,
                              int B, int n, const xvr_drr_spec* spec,
                              float* out, float* jac, unsigned long long* work, void* stream);

/*
 * Trilinear backward by re-marching (needed for grad_volume, and for the pose gradient when C > 1).
 * Replaces autograd through grid_sample (grid gradient -> pose, input gradient -> voxels).
 *   grad_volume  nullable [D0][D1][D2]; ACCUMULATED into (caller zeroes it).
 *   grad_source  nullable [B][3]; ACCUMULATED into (caller zeroes it) -- the sum over a pose's rays.
 *   grad_target  nullable [B][n][3]; written.
 *   grad_raylen  nullable [B][n]; written.
 *   grad_source/grad_target must be both null or both non-null.
 *   workspace    nullable device scratch of xvr_drr_backward_workspace_bytes(B, n, D0, D1, D2) bytes, 16-B aligned.
 *                With it, and when the rays are a detector lattice (spec.ray_grid_w > 1), grad_volume is computed without
 *                global atomics, as the exact transpose of the forward, bit-reproducibly: by a brick-local splat whose
 *                per-voxel sums are exact 32-bit fixed-point integers in LDS -- over sample runs on the shared planes for
 *                the plain render, ray by ray under clip_to_volume and / or a mask (a mask whose grad_out is the same for
 *                every channel should be passed as mask = NULL, C = 1: the gradient is then the unmasked one).  Every
 *                product w * c is rounded to 2^-31 of a rigorous bound on the voxel's sum over one pose: ~2e-8 of the
 *                pose's largest |grad_out * raylen / n_points| at the benchmark geometry.
 *                ACCURACY (an absolute floor per (pose, brick), so the RELATIVE error grows as a voxel's gradient shrinks;
 *                measured against autograd through the oracle in float64, next to the fp32 gather of gather_splat = 0;
 *                tools/splat_accuracy.py, profiles/r03_splat_accuracy.md, tests/test_splat.py, tests/test_configs.py):
 *                  - benchmark geometry (512^3 -> 256^2, ~13 samples of a pose per voxel, grad_out >= 0): identical to the
 *                    fp32 gather's error in every decade of |g| / max|g| down to 1e-7 -- both are the fp32 sample positions;
 *                  - a few samples per voxel with a SIGNED noise grad_out (the voxel sums cancel, the bound cannot): within
 *                    4 x of the fp32 gather's error down to 1e-4 max|g| (median 2e-5 relative there), 9 x at 1e-5;
 *                  - worst case measured, pixels 15 x finer than voxels (~1500 samples of a pose per voxel; the bound, hence
 *                    the LSB, is 30 x the benchmark's): median 3e-6 relative in the top decade (fp32 gather 8e-7), 3e-4 at
 *                    1e-4 max|g| (fp32 gather 4e-6).  A contribution below bound * 2^-32 rounds to zero.
 *                    Since round 5 that regime does not reach the splat by default: k_gather_prep estimates the samples per
 *                    voxel of every pose and above ~48 the launch is the fp32 table gather's (option gather_splat = 3 forces the splat).
 *                The sums live in three quarters of the int32 range; one found in the guard band at a flush (the bound on a voxel's
 *                sum was optimistic) turns its voxels into NaN and sets word 2 of the workspace -- never a silently wrapped number.
 *                Callers that need fp32 sums throughout set the option gather_splat = 0 (the table gather, 1.6 x slower).
 *                A non-finite grad_out turns the voxels of the 16^3 bricks its pose touches into NaN.  Otherwise -- or when the kernel finds on the
 *                device that the targets are not a lattice -- it falls back to a scatter with fp32 atomics: same result
 *                up to summation order, more than an order of magnitude slower on MI355X.
 */
int xvr_drr_trilinear_backward(const float* volume, const float* mask, int D0, int D1, int D2, int C,
                               const float* source, const float* target, const float* raylen,
                               int B, int n, const xvr_drr_spec* spec, const float* grad_out,
                               float* grad_volume, float* grad_source, float* grad_target,
                               float* grad_raylen, void* workspace, size_t workspace_bytes, void* stream);

/* Siddon exact ray tracing, same contract.  Replaces Siddon.forward(volume, source, target, img, mask=...).
 *   work counts voxel segments traversed. */
int xvr_drr_siddon_forward(const float* volume, const float* mask, int D0, int D1, int D2, int C,
                           const float* source, const float* target, const float* raylen,
                           int B, int n, const xvr_drr_spec* spec,
                           float* out, float* jac, unsigned long long* work, void* stream);

int xvr_drr_siddon_backward(const float* volume, const float* mask, int D0, int D1, int D2, int C,
                            const float* source, const float* target, const float* raylen,
                            int B, int n, const xvr_drr_spec* spec, const float* grad_out,
                            float* grad_volume, float* grad_source, float* grad_target,
                            float* grad_raylen, void* workspace, size_t workspace_bytes, void* stream);

/*
 * Pose-side backward from the jacobian saved by a forward call: an elementwise product with the
 * (channel-uniform) grad_out [B][n] plus a wave-level reduction of grad_source over each pose's rays.
 *   grad_source [B][3] is ACCUMULATED into (caller zeroes it); grad_target [B][n][3] and
 *   grad_raylen [B][n] (nullable) are written.
 */
int xvr_drr_backward_from_jac(const float* jac, const float* grad_out, int B, int n,
                              float* grad_source, float* grad_target, float* grad_raylen,
                              void* stream);

/*
 * Ray generation in one pass.  Replaces the three torch calls xvr makes before every render --
 * `source, target = drr.detector(pose, None)`, `img = (target - source).norm(dim=-1)`,
 * `source, target = affinv(source), affinv(target)` (/root/reference/src/xvr/model/trainer.py:283-285;
 * the same sequence inside DRR.forward, /root/reference/src/xvr/registrar/base.py:249).
 *   cam  [B][24] = { Mv[3][3], s_v[3], Mw[3][3], s_w[3] }: target_vox(i,j) = Mv (i,j,1)^T, source_vox = s_v,
 *        raylen(i,j) = | Mw (i,j,1)^T - s_w |   (i = detector row, j = column; the host folds calibration,
 *        reorientation, pose and the CT's inverse affine into the two 3x3 maps)
 *   out: source [B][3], target [B][H*W][3], raylen [B][H*W]
 * backward: grad_cam [B][24] is ACCUMULATED into (caller zeroes it); grad_source / grad_raylen nullable.
 */
int xvr_drr_rays_forward(const float* cam, int B, int H, int W, float* source, float* target, float* raylen,
                         void* stream);
int xvr_drr_rays_backward(const float* cam, int B, int H, int W, const float* grad_source,
                          const float* grad_target, const float* grad_raylen, float* grad_cam, void* stream);

/*
 * Forward calls that generate their rays from the camera vector (cam [B][24] as in xvr_drr_rays_forward)
 * instead of loading source / target / raylen: same arithmetic as xvr_drr_rays_forward followed by the
 * forward call, bit for bit, without the [B][H*W][4] floats in between.  For callers that differentiate
 * w.r.t. the pose only (the reference's registration loop; /root/reference/src/xvr/registrar/base.py:249):
 * the backward is then xvr_drr_jac_to_camera_backward on the jacobian written here.
 */
int xvr_drr_trilinear_forward_camera(const float* volume, const float* mask, int D0, int D1, int D2, int C,
                                     const float* cam, int B, int H, int W, const xvr_drr_spec* spec, float* out,
                                     float* jac, unsigned long long* work, void* stream);
int xvr_drr_siddon_forward_camera(const float* volume, const float* mask, int D0, int D1, int D2, int C,
                                  const float* cam, int B, int H, int W, const xvr_drr_spec* spec, float* out,
                                  float* jac, unsigned long long* work, void* stream);

/*
 * Labels packed into the volume, for the mask -> channels renders of the training loop
 * (/root/reference/src/xvr/model/trainer.py:288, `renderer(..., mask=seg)`).  The label of a sample is the
 * label of its NEAREST voxel, which is always one of the 8 voxels the interpolation loads anyway; with the
 * label stored in the low 4 mantissa bits of every voxel the separate lookup (a fifth gather per sample,
 * +50 % render time) disappears.
 *   xvr_drr_pack_labels   packed[i] = volume[i] with mantissa bits 0..3 := min(max((int)mask[i], 0), 15)
 *   forward calls         pass `packed` as the volume, mask = NULL and C in [2, 16]
 * The density the render sees differs from `volume` by at most 15 ulp (1.8e-6 relative).  Backward calls
 * take the original volume and mask.
 */
int xvr_drr_pack_labels(const float* volume, const float* mask, long long n, float* packed, void* stream);

/*
 * y-pair interleaved copy of a volume for the trilinear forward (spec.volume_layout = 1):
 *     pairs[x][yp][z] = (V[x][yp - 1][z], V[x][yp][z]),  yp = 0 .. D1, zero outside the volume
 * i.e. [D0][D1 + 1][D2][2] floats = xvr_drr_ypairs_bytes().  One 16-byte load at (x, floor(y) + 1, z) then returns the
 * four taps (y, z), (y + 1, z), (y, z + 1), (y + 1, z + 1): a sample costs two gather instructions instead of four, and
 * the march is bound by the texture-address rate per instruction (HISTORY.md section 4.2).  Twice the memory of the
 * volume; built in one streaming pass; the forward's arithmetic is unchanged (identical output bits).
 */
size_t xvr_drr_ypairs_bytes(int D0, int D1, int D2);
int xvr_drr_pack_ypairs(const float* volume, int D0, int D1, int D2, float* pairs, void* stream);
/* The y-pair copy of the LABEL-CARRYING volume (xvr_drr_pack_labels then xvr_drr_pack_ypairs) in one pass over volume and mask:
 * for masked renders of large launches whose volume is new every call (xvr's training step, trainer.py:185-230). */
int xvr_drr_pack_labels_ypairs(const float* volume, const float* mask, int D0, int D1, int D2, float* pairs, void* stream);
/* The same y-pairs cut into 128-byte lines of 2 x-rows x 8 z-entries, tiles [ceil(D0 / 2)][D1 + 1][(D2 - 2) / 7 + 1][2][8][2] that
 * OVERLAP by one entry along z (tile b holds z = 7 b .. 7 b + 7: the 16 bytes of any (z, z + 1) pair lie in one tile), for
 * volume_layout = 3 (round 4).  The forward is bound by fabric bandwidth, one 128-byte line per L2 miss whatever part is used
 * (profiles/r04_fetch_calibration.txt); an oblique view cuts a z-run of 16 entries after ~2 voxels, a compact tile is used more
 * densely.  8/7 of the row copy's memory; same results, bit for bit.  D2 < 8192. */
size_t xvr_drr_ytiles_bytes(int D0, int D1, int D2);
int xvr_drr_pack_ytiles(const float* volume, int D0, int D1, int D2, float* tiles, void* stream);
int xvr_drr_pack_labels_ytiles(const float* volume, const float* mask, int D0, int D1, int D2, float* tiles, void* stream);
/* ... and with the HU -> density map of xvr_drr_hu_to_density applied on the way (`hu`: Hounsfield units; `stats`: xvr_drr_hu_stats
 * of it): the tiles hold exactly the bits xvr_drr_hu_to_density followed by xvr_drr_pack_labels_ytiles would, without the density
 * volume in between -- the per-step 512 MiB round trip of /root/reference/src/xvr/model/trainer.py:196-197 (round 5). */
int xvr_drr_pack_hu_labels_ytiles(const float* hu, const float* mask, const void* stats, float bone_multiplier, int D0, int D1, int D2,
                                  float* tiles, void* stream);

/*
 * Bricked copy of a volume for the Siddon forward (spec.volume_layout = 2):
 *     bricks[x / 2][y / 2][z / 8][x % 2][y % 2][z % 8],  zeros beyond the volume
 * i.e. one 128-byte cache line per 2 x 2 x 8 block of voxels; ceil(D0/2) ceil(D1/2) ceil(D2/8) 32 floats =
 * xvr_drr_bricks_bytes().  The traversal is bound by the number of distinct lines one wavefront load touches (its 64 rays
 * sit at different depths): the bricks halve it (tools/sim_siddon_lines.py).  Same arithmetic, identical output bits.
 */
size_t xvr_drr_bricks_bytes(int D0, int D1, int D2);
int xvr_drr_pack_bricks(const float* volume, int D0, int D1, int D2, float* bricks, void* stream);

/*
 * Jacobian -> camera in one pass (= xvr_drr_backward_from_jac followed by xvr_drr_rays_backward, without
 * materialising grad_target / grad_raylen), for callers that optimise the pose and never look at per-ray
 * gradients: the registration loop (/root/reference/src/xvr/registrar/base.py:252, loss.backward()).
 *   jac [B][H*W][8] from a forward call, grad_out [B][H*W] (C == 1), cam [B][24]  ->  grad_cam [B][24] WRITTEN
 * The sums are order-deterministic (fixed-order two-level reduction, no float atomics): identical bits on
 * every run.  `workspace`: xvr_drr_jac_to_camera_workspace_bytes() bytes, ZERO-FILLED ONCE by the caller
 * before the first use; every call leaves it ready for the next one.
 */
size_t xvr_drr_jac_to_camera_workspace_bytes(int B, int H, int W);
int xvr_drr_jac_to_camera_backward(const float* jac, const float* grad_out, const float* cam, int B, int H, int W,
                                   float* grad_cam, void* workspace, size_t workspace_bytes, void* stream);

/*
 * HU -> density of a whole CT.  Replaces diffdrr.data.transform_hu_to_density(volume, multiplier), which
 * xvr calls on the full volume before the renders of every training step
 * (/root/reference/src/xvr/model/trainer.py:124,196-197).
 *   xvr_drr_hu_stats      one reduction per CT: per-class (air / soft tissue / bone) min, max, presence
 *                         into `stats` (48 bytes of device memory); independent of the multiplier
 *   xvr_drr_hu_to_density one read + one write per step; the min-max normalisation constants are derived
 *                         on the device from `stats` and the multiplier (no host sync)
 * Volumes must be 16-byte aligned; `n` = number of voxels.
 */
int xvr_drr_hu_stats(const float* hu, long long n, void* stats, void* stream);
int xvr_drr_hu_to_density(const float* hu, long long n, const void* stats, float bone_multiplier,
                          float* density, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* XVR_DRR_H */
