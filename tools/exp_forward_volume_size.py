import sys, time, torch
sys.path.insert(0, "/root/repo")
from bench import deepfluoro_poses
from xvr_amd import renderers
from xvr_amd.data import make_phantom, read
from xvr_amd.drr import DRR
dev = torch.device("cuda")
B, H = 116, 256
rot, xyz = (t.to(dev) for t in deepfluoro_poses(B, seed=0).convert("euler_angles", "ZXY"))
for size in (512, 256, 128):
    for yp in (True, False):
        renderers.YPAIR_LAYOUT = yp
        vol, _ = make_phantom(size, n_ellipsoids=64, seed=0, device=dev)
        drr = DRR(read(vol, spacing=(512.0 / size,) * 3, orientation="AP"), 1020.0, H, 1.08821875, renderer="trilinear", reverse_x_axis=False).to(dev)
        r = rot.clone().requires_grad_(True)
        for _ in range(2):
            img = drr(r, xyz, parameterization="euler_angles", convention="ZXY", n_points=500)
        renderers.PROFILER = []
        for _ in range(4):
            img = drr(r, xyz, parameterization="euler_angles", convention="ZXY", n_points=500)
        torch.cuda.synchronize()
        ev, renderers.PROFILER = renderers.PROFILER, None
        t = [a.elapsed_time(b) for k, a, b in ev if "forward" in k]
        print(f"volume {size}^3 at {512 / size:.0f} mm ({vol.numel() * 4 / 2**20:.0f} MiB), ypairs {yp}: forward+jac {sum(t) / len(t):.3f} ms", flush=True)
        del drr, vol
