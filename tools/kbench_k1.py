#!/usr/bin/env python3
"""K1 at 512^3 / 1280x720 (bench.py's k1_512 views plus oblique ones) under a list of option settings, with a bit-exact
check of every variant against the row sweep.
usage: python tools/kbench_k1.py [--res 512] [--reps 20] [--variants "default;k1_prefetch=0;k1_brick_y=2,k1_nt=1"]"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamicfusion_body_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=512)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--angles", default="0,60,135")
ap.add_argument("--variants", default="k1_no_bricks=1;default;k1_prefetch=0;k1_brick_y=2;k1_brick_y=2,k1_nt=1;k1_brick_y=2,k1_nt=1,k1_prefetch=0")
ap.add_argument("--no-check", action="store_true")
ap.add_argument("--lib", default=None, help="alternative libdfusion_hip.so (tools/build_variant.sh)")
a = ap.parse_args()
if a.lib:
    _lib.LIB_PATH = os.path.abspath(a.lib)
from dynamicfusion_body_amd import kernels, scene
R = a.res
cam = "C2" if R <= 256 else "C5"
H, W, fx, cx, cy = scene.CAMERAS[cam]
K = scene.intrinsics(fx, cx, cy); Kinv = np.linalg.inv(K)
scale, center, tdist = scene.grid_params(R)
views = []
for ang in [float(x) for x in a.angles.split(",")]:
    lw = scene.view_extrinsic(ang)
    views.append(("%g deg" % ang, lw, torch.from_numpy(scene.render_depth(K, lw, H, W, dtype=np.float32)).cuda()))
lw_full = scene.view_extrinsic(0.0).copy(); lw_full[2, 3] += 1.5
views.append(("all voxels", lw_full, torch.full((H, W), -8.0, dtype=torch.float32, device="cuda")))

def set_variant(v):
    names = []
    if v != "default":
        for kv in v.split(","):
            k, val = kv.split("=")
            _lib.set_option(k, int(val)); names.append(k)
    return names

e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
ref = {}
print("%-44s" % "variant" + "".join("%14s" % n for n, _, _ in views))
for v in a.variants.split(";"):
    names = set_variant(v)
    row = []
    for name, lw, d in views:
        T = torch.full((R, R, R), tdist, dtype=torch.float32, device="cuda"); Wt = torch.zeros_like(T)
        for _ in range(2):
            kernels.integrate_depth(T, Wt, d, K, Kinv, lw, scale, center, tdist)
        if not a.no_check:
            key = name
            if key not in ref:
                ref[key] = (T.clone(), Wt.clone())
            else:
                ok = torch.equal(T, ref[key][0]) and torch.equal(Wt, ref[key][1])
                if not ok:
                    print("MISMATCH variant %s view %s: %d voxels differ" % (v, name, int(((T != ref[key][0]) | (Wt != ref[key][1])).sum())))
        torch.cuda.synchronize()
        e0.record()
        for _ in range(a.reps):
            kernels.integrate_depth(T, Wt, d, K, Kinv, lw, scale, center, tdist)
        e1.record(); torch.cuda.synchronize()
        row.append(e0.elapsed_time(e1) / a.reps * 1e3)
        del T, Wt
    print("%-44s" % v + "".join("%11.1f us" % x for x in row), flush=True)
    for k in names:
        _lib.set_option(k, None)
