#!/usr/bin/env python
"""CPU model of the control flow of the trilinear voxel gather (k_trilinear_gather_vol) on the benchmark
geometry: how many wave-level loop trips each loop-nest organisation needs, without a GPU.

It replays, in numpy, the integer loop bounds every lane derives (step range, row range per step, pixel
interval per row) for a random sample of 8^3 bricks over the benchmark's 116 poses, and prices a few
candidate organisations of the same work with per-section instruction counts read off the ISA
(`hipcc -S`).  Only geometry: no volume, no weights.  Used to choose the round-2 kernel structure
(HISTORY.md section 4.1); the numbers it prints for the round-1 structure agree with the instrumented
build of tools/gather_stats.py (2.19 steps per visit, 3.4 rows per step, 3.3 candidates per row).
"""
import argparse
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def pose_lattices(size, det, batch, seed=0):
    from bench import deepfluoro_poses
    from xvr_amd.data import read
    from xvr_amd.drr import DRR

    vol = torch.empty(size, size, size)
    subject = read(vol, orientation="AP")
    delx = 1.08821875 * 256 / det
    drr = DRR(subject, 1020.0, det, delx, renderer="trilinear", reverse_x_axis=False)
    pose = deepfluoro_poses(batch, seed=seed)
    s, t = drr.detector(pose, None)
    s = drr.affine_inverse(s).double().numpy()[:, 0]          # [B,3]
    t = drr.affine_inverse(t).double().numpy().reshape(batch, det, det, 3)
    return s, t


def lattice_constants(s, t, a=1.0, eps=1e-8):
    """numpy restatement of k_gather_prep's per-pose constants (a = 1: the default index map)."""
    H, W = t.shape[1:3]
    P = []
    for b in range(len(s)):
        T = t[b]
        t00 = T[0, 0]
        ec = (T[0, W - 1] - t00) / (W - 1)
        er = (T[H - 1, 0] - t00) / (H - 1)
        ts = (t00 + eps) - s[b]
        nrm = np.cross(ec, er)
        h = nrm @ ts
        tmp = np.cross(er, nrm)
        gc = tmp / (ec @ tmp)
        tmp = np.cross(nrm, ec)
        gr = tmp / (er @ tmp)
        nh = nrm / h
        rl, rc = np.zeros(3), np.zeros(3)
        dalpha = np.abs(nh).sum() / a
        for j in range(3):
            if abs(nh[j]) > 1e-3 * dalpha:
                lam = gr[j] / nh[j]
                rl[j], rc[j] = lam, np.abs(gr - lam * nh).sum()
            else:
                rl[j], rc[j] = 0.0, 1e30
        P.append(dict(s=s[b], nh=nh, gc=gc, gr=gr, st=ts, ec=ec, er=er, dalpha=dalpha, hwc=np.abs(gc).sum(), hwr=np.abs(gr).sum(),
                      gc0=gc @ (-ts), gr0=gr @ (-ts), rl=rl, rc=rc))
    return P


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--det", type=int, default=256)
    ap.add_argument("--batch", type=int, default=116)
    ap.add_argument("--bricks", type=int, default=600)
    ap.add_argument("--n-points", type=int, default=500)
    args = ap.parse_args()
    D, H, N = args.size, args.det, args.n_points
    W = H
    s, t = pose_lattices(D, H, args.batch)
    P = lattice_constants(s, t)
    rng = np.random.default_rng(0)
    nb = D // 8
    bricks = rng.integers(0, nb, size=(args.bricks, 3))
    lane = np.arange(64)
    lx, ly, lz = lane >> 4, (lane >> 2) & 3, lane & 3
    # first voxel of every lane's 2x2x2 block: [bricks, 64, 3]
    fv = np.stack([(bricks[:, None, 0] * 4 + lx) * 2, (bricks[:, None, 1] * 4 + ly) * 2, (bricks[:, None, 2] * 4 + lz) * 2], -1).astype(float)
    HS, CO = 1.5, 0.5
    b_ = 0.5 - 0.5     # voxel_shift 0.5 -> b = 0
    xv = fv + CO - b_
    step = 1.0 / (N - 1)
    MARG, JM = 0.03, 0.03

    tot = dict(visits=0, wave_pose=0, steps=0, rows=0, rows_empty=0, cands=0,
               w_steps=0, w_rows=0, w_trips=0, flat_rows=0, flat_trips=0, flat_rows_trips=0, lane_trips=0,
               lane_trip_max=0, rowslot_trips=0)
    hist_steps = np.zeros(8, int)
    hist_rows = np.zeros(16, int)
    hist_visit_rows = np.zeros(64, int)
    tile_sizes = []
    zero_wave = zero_lane = 0
    lane_trips_by_pose, lane_rows_by_pose, keep_by_pose = [], [], []
    for p, L in enumerate(P):
        w = xv - L["s"]
        av = w @ L["nh"]
        da = HS * L["dalpha"]
        klo = np.ceil(np.maximum((av - da) / step - 1e-3, 0)).astype(int)
        khi = np.floor(np.minimum((av + da) / step + 1e-3, N - 1)).astype(int)
        grw = w @ L["gr"]
        # cull at brick level: emulate "any lane has any candidate" (the kernel's cull is conservative; close enough)
        nst = np.maximum(khi - klo + 1, 0)
        # the kernel's cull (k_gather_cull): brick box incl. support against the pose's sample pyramid
        c = bricks * 8 + 3.5
        hx = 0.5 * 7 + 1.5
        wb = c - L["s"]
        avb = wb @ L["nh"]
        en = np.abs(L["nh"]) * hx
        dab = en.sum()
        keep = (avb + dab >= 0.0) & (avb - dab <= 1.0)
        front = avb - dab > 1e-6
        nj, ni = wb @ L["gc"], wb @ L["gr"]
        jv, iv = [], []
        for cc in range(8):
            sg = np.array([1.0 if cc & 4 else -1.0, 1.0 if cc & 2 else -1.0, 1.0 if cc & 1 else -1.0])
            inv_c = 1.0 / (avb + (sg * L["nh"] * hx).sum())
            jv.append((nj + (sg * L["gc"] * hx).sum()) * inv_c)
            iv.append((ni + (sg * L["gr"] * hx).sum()) * inv_c)
        jv, iv = np.array(jv), np.array(iv)
        vis = (jv.max(0) + L["gc0"] + 1 >= 0) & (jv.min(0) + L["gc0"] - 1 <= W - 1) & (iv.max(0) + L["gr0"] + 1 >= 0) & (iv.min(0) + L["gr0"] - 1 <= H - 1)
        keep = keep & (~front | vis)
        nst = np.where(keep[:, None], nst, 0)
        maxst = nst.max(axis=1)
        lane_total_trips = np.zeros(nst.shape, int)
        lane_rows = np.zeros(nst.shape, int)
        lane_nonempty_rows = np.zeros(nst.shape, int)
        per_k = []
        wave_rows_sum = np.zeros(len(bricks), int)
        wave_trips_sum = np.zeros(len(bricks), int)
        wave_steps = np.zeros(len(bricks), int)
        lane_cands = np.zeros(nst.shape, int)
        bb = [np.full(len(bricks), 10**9), np.full(len(bricks), -1), np.full(len(bricks), 10**9), np.full(len(bricks), -1)]
        rowslot = np.zeros(len(bricks), int)
        for it in range(int(maxst.max()) if maxst.size else 0):
            k = klo + it
            act = it < nst
            if not act.any():
                break
            al = k * step
            al = np.where(al > 1e-12, al, 1.0)
            inv = 1.0 / al
            ic = grw * inv + L["gr0"]
            dlt = al - av
            up = np.minimum.reduce([L["rl"][0] * dlt + HS * L["rc"][0], L["rl"][1] * dlt + HS * L["rc"][1], L["rl"][2] * dlt + HS * L["rc"][2],
                                    np.full_like(dlt, HS * L["hwr"])])
            dn = np.minimum.reduce([-L["rl"][0] * dlt + HS * L["rc"][0], -L["rl"][1] * dlt + HS * L["rc"][1], -L["rl"][2] * dlt + HS * L["rc"][2],
                                    np.full_like(dlt, HS * L["hwr"])])
            ilo = np.ceil(np.maximum(ic - (np.maximum(dn, 0) * inv + MARG), 0)).astype(int)
            ihi = np.floor(np.minimum(ic + (np.maximum(up, 0) * inv + MARG), H - 1)).astype(int)
            nrow = np.where(act, np.maximum(ihi - ilo + 1, 0), 0)
            q0 = (L["s"] + al[..., None] * L["st"]) + b_ - (fv + CO)      # [bricks,64,3]
            uc = al[..., None] * L["ec"]
            ur = al[..., None] * L["er"]
            r = np.where(np.abs(uc) < 1e-9, 1e9, 1.0 / np.where(np.abs(uc) < 1e-9, 1.0, uc))
            ahw = HS * np.abs(r)
            maxrow = nrow.max(axis=1)
            step_trips = np.zeros(nst.shape, int)
            w_rows_k = np.zeros(len(bricks), int)
            w_trips_k = np.zeros(len(bricks), int)
            for ri in range(int(maxrow.max()) if maxrow.size else 0):
                i = ilo + ri
                ract = ri < nrow
                qq = q0 + i[..., None] * ur
                m = -qq * r
                lo = (m - ahw).max(axis=-1)
                hi = (m + ahw).min(axis=-1)
                jlo = np.ceil(np.maximum(lo - JM, 0)).astype(int)
                jhi = np.floor(np.minimum(hi + JM, W - 1)).astype(int)
                nc = np.where(ract, np.maximum(jhi - jlo + 1, 0), 0)
                trips = (nc + 1) // 2
                tot["rows"] += int(ract.sum())
                tot["rows_empty"] += int((ract & (nc == 0)).sum())
                tot["cands"] += int(nc.sum())
                wave_has = ract.any(axis=1)
                w_rows_k += wave_has
                w_trips_k += trips.max(axis=1)
                step_trips += trips
                lane_cands += nc
                lane_rows += ract
                lane_nonempty_rows += ract & (nc > 0)
                has = ract & (nc > 0)
                big = 10**9
                bb[0] = np.minimum(bb[0], np.where(has, i, big).min(axis=1))
                bb[1] = np.maximum(bb[1], np.where(has, i, -1).max(axis=1))
                bb[2] = np.minimum(bb[2], np.where(has, jlo, big).min(axis=1))
                bb[3] = np.maximum(bb[3], np.where(has, jhi + 1, -1).max(axis=1))
            rowslot += step_trips.max(axis=1)            # (lane = (block, row) organisation: not priced further)
            wave_act = act.any(axis=1)
            wave_steps += wave_act
            wave_rows_sum += w_rows_k
            wave_trips_sum += w_trips_k
            lane_total_trips += step_trips
            tot["steps"] += int(act.sum())
            hist_rows += np.bincount(np.minimum(nrow[act], 15), minlength=16)
        visited = maxst > 0
        tot["visits"] += int((nst > 0).sum())
        tot["wave_pose"] += int(visited.sum())
        tot["w_steps"] += int(wave_steps.sum())
        tot["w_rows"] += int(wave_rows_sum.sum())
        tot["w_trips"] += int(wave_trips_sum.sum())
        tot["flat_rows"] += int(lane_rows.max(axis=1).sum())            # lane-private (k, row) flattened
        tot["lane_trips"] += int(lane_total_trips.sum())
        tot["lane_trip_max"] += int(lane_total_trips.max(axis=1).sum())   # everything flattened per lane
        hist_steps += np.bincount(np.minimum(nst[nst > 0], 7), minlength=8)
        lane_trips_by_pose.append(lane_total_trips)
        zero_wave += int(((lane_cands.sum(axis=1) == 0) & visited).sum()); zero_lane += int(((lane_cands == 0) & (nst > 0)).sum())
        ok = bb[1] >= 0
        tile_sizes.extend(((bb[1] - bb[0] + 1) * (bb[3] - bb[2] + 1))[ok].tolist())
        lane_rows_by_pose.append(lane_nonempty_rows)
        keep_by_pose.append(visited)
        hist_visit_rows += np.bincount(np.minimum(lane_nonempty_rows[nst > 0], 63), minlength=64)

    wp = tot["wave_pose"]
    print(f"bricks sampled {len(bricks)}, wave-pose iterations {wp} ({wp / len(bricks):.1f} poses per brick)")
    print(f"per (lane,pose) visit: steps {tot['steps'] / tot['visits']:.2f}, rows/step {tot['rows'] / tot['steps']:.2f} "
          f"({100 * tot['rows_empty'] / tot['rows']:.0f} % empty), cands/nonempty row {tot['cands'] / (tot['rows'] - tot['rows_empty']):.2f}, "
          f"cands/visit {tot['cands'] / tot['visits']:.1f}")
    print("steps-per-visit histogram:", hist_steps.tolist())
    print("rows-per-step histogram:", hist_rows.tolist())
    print(f"per wave-pose: steps {tot['w_steps'] / wp:.2f}, row iterations {tot['w_rows'] / wp:.2f}, inner trips {tot['w_trips'] / wp:.2f} "
          f"(ideal, all lanes busy: {tot['cands'] / 2 / 64 / wp:.2f}; lanes' own trips, mean {tot['lane_trips'] / 64 / wp:.2f})")
    print(f"  flattened (k,row) per lane: row iterations {tot['flat_rows'] / wp:.2f}; everything flattened per lane: trips {tot['lane_trip_max'] / wp:.2f}")
    print(f"wave-poses the cull lets through with no candidate at all: {zero_wave / wp:.3f}; (lane, pose) visits with no candidate: {zero_lane / tot['visits']:.3f}")
    ts = np.array(tile_sizes)
    print("pixel bounding box of a wavefront's candidates per pose (elements of 16 B): mean %.0f, median %.0f, 90 %% %.0f, 99 %% %.0f, max %d; fit in 256 / 320 / 384 / 448 / 512: %s" % (
        ts.mean(), np.median(ts), np.percentile(ts, 90), np.percentile(ts, 99), ts.max(), [round(float((ts <= c).mean()), 3) for c in (256, 320, 384, 448, 512)]))
    LT = np.stack(lane_trips_by_pose)      # [poses, bricks, 64]
    KP = np.stack(keep_by_pose)            # [poses, bricks]
    cum = np.cumsum(hist_visit_rows) / hist_visit_rows.sum()
    print("non-empty rows per (lane,pose) visit: mean %.2f; P(<=8) %.3f P(<=12) %.3f P(<=16) %.3f P(<=20) %.3f P(<=24) %.3f P(<=32) %.3f" % (
        (hist_visit_rows * np.arange(64)).sum() / hist_visit_rows.sum(), cum[8], cum[12], cum[16], cum[20], cum[24], cum[32]))
    LR = np.stack(lane_rows_by_pose)
    for T in (8, 10, 12, 13, 16, 24, 32):
        print(f"  table of {T}: wave-poses with an overflowing lane {((LR.max(axis=2) > T) & KP).sum() / KP.sum():.3f}")
    for batch in (1, 2, 4, 8):
        tot_b = 0
        for b in range(len(bricks)):
            idx = np.nonzero(KP[:, b])[0]
            for g in range(0, len(idx), batch):
                tot_b += LT[idx[g:g + batch], b].sum(axis=0).max()
        print(f"  flattened trips, {batch} pose(s) per table fill: {tot_b / wp:.2f} per wave-pose")
    for name, (cp, cs, cr, ct) in {"round-1 ISA (pose 60, step 150, row 35, trip 62)": (60, 150, 35, 62),
                                   "slim step setup (pose 80, step 50, row 30, trip 62)": (80, 50, 30, 62)}.items():
        cur = cp + (cs * tot["w_steps"] + cr * tot["w_rows"] + ct * tot["w_trips"]) / wp
        print(f"{name}: nested {cur:.0f} wave-instr per wave-pose")


if __name__ == "__main__":
    main()
