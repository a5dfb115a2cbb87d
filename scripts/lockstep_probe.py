#!/usr/bin/env python3
"""Finds the first frame (and the first fp64 stage: PnP-RANSAC pose, pose before the LM, pose after the LM) at which the HIP
front-end and the CPU oracle stop being bit-identical, per camera mode.  Test/debug aid (needs an MI355X); used in round 2 to
bring the closed loop into lockstep for whole runs.  Prints `None` for a stream that never diverged."""
import ctypes as C, os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as O
import flvis_amd
from flvis_amd import synth
def cfgs(text):
    p = os.path.join(tempfile.gettempdir(), "probe.yaml"); open(p, "w").write(text)
    cfg = flvis_amd.load_config(p); ocfg = O.RefConfig(); C.memmove(C.byref(ocfg), C.byref(cfg), C.sizeof(cfg)); return cfg, ocfg
def run(text, rig, streams, nframes, imu=True):
    cfg, ocfg = cfgs(text)
    ctx = flvis_amd.Context(0)
    S = len(streams)
    trajs = [synth.Trajectory(s) for s in streams]
    rnd = synth.Renderer("cuda", rig=rig)
    trk = flvis_amd.Tracker(ctx, cfg, S, seed_base=0xF1715)
    refs = [O.Tracker(ocfg, 0xF1715 + i) for i in range(S)]
    t_prev = -0.05
    first_bad = [None] * S
    maxd = [0.0] * S
    for f in range(nframes):
        t = f / synth.FRAME_HZ
        for i, s in enumerate(streams):
            if not imu: break
            smp = synth.imu_samples(trajs[i], s, t_prev, t)
            trk.imu_feed_flvis(i, smp)
            for r in smp: refs[i].imu(r[0], r[1:4], r[4:7])
        t_prev = t
        i0, i1 = rnd.stereo_frame(trajs, t, f)
        h0, h1 = i0.cpu().numpy(), i1.cpu().numpy()
        outs = trk.image_feed(i0, i1, [t] * S, with_local_map=False)
        for i in range(S):
            w = refs[i].image(t, h0[i], h1[i]); g = outs[i]
            d = np.abs(g["pose7"] - w["pose7"]).max()
            same = g["state"] == w["state"] and g["n_landmarks"] == w["n_landmarks"] and np.array_equal(g["dbg"], w["dbg"])
            if first_bad[i] is None:
                maxd[i] = max(maxd[i], d)
                if not same or d != 0.0:
                    gs = (C.c_double * 21)(); ws = (C.c_double * 21)()
                    ctx._lib.flvis_debug_stage_poses(ctx._h, i, gs); O.lib().ref_tracker_stage_poses(refs[i].h, ws)
                    gs, ws = np.array(gs[:]), np.array(ws[:])
                    first_bad[i] = (f, d, same, "pnp diff", np.abs(gs[:7] - ws[:7]).max(), "lm diff", np.abs(gs[7:14] - ws[7:14]).max(), "pre-lm diff", np.abs(gs[14:] - ws[14:]).max(), g["n_landmarks"])
    ctx.close()
    return first_bad, maxd
for name, text, rig, streams, n, imu in (("euroc", synth.EUROC_LIKE_YAML, synth.euroc_rig(), [9], 50, True),
                                    ("kitti", synth.KITTI_LIKE_YAML, synth.kitti_like_rig(), [5], 26, False)):
    print(name, run(text, rig, streams, n, imu))
