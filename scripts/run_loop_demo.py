#!/usr/bin/env python3
"""All three nodelets' work in one process on one GPU: the tracker (front-end + IMU) on a rendered stereo + IMU sequence that
circles the room and comes back, its keyframes into flvis_loop_closer (ORB, bag of words, landmarks, candidates, PnP check, pose
graph), the corrected keyframe path out.  Prints one JSON line: keyframes, loops, ATE of the keyframe positions against ground
truth before / after loop closing.

usage: run_loop_demo.py [seconds=70] [period=60]"""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import flvis_amd
from flvis_amd import synth
import _geom as G
import _loop_chain as LC
import _voc as V

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 70.0
period = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
ctx = flvis_amd.Context(0)
p = os.path.join(tempfile.gettempdir(), "flvis_loop_demo.yaml")
open(p, "w").write(synth.D435I_STEREO_YAML)
cfg = flvis_amd.load_config(p)
tr = LC.LoopTrajectory(T=period)
rnd = synth.Renderer("cuda")
trk = flvis_amd.Tracker(ctx, cfg, 1)
n_frames = int(secs * synth.FRAME_HZ)
kf_imgs, kf_pose, kf_time = [], [], []
t_prev = -1.0 / synth.FRAME_HZ
states = []
for f in range(n_frames):
    t = f / synth.FRAME_HZ
    trk.imu_feed_flvis(0, synth.imu_samples(tr, 0, t_prev, t))
    t_prev = t
    i0, i1 = rnd.stereo_frame([tr], t, f)
    out = trk.image_feed(i0, i1, [t], with_local_map=False)[0]
    states.append(out["state"])
    if out["new_keyframe"]:
        kf_imgs.append((i0, i1))
        kf_pose.append(out["pose7"].copy())
        kf_time.append(t)
n_kf = len(kf_pose)
# vocabulary from every eighth keyframe's ORB descriptors
train = []
for i in range(0, n_kf, 8):
    k, d, c, _ = ctx.orb_detect_and_compute(kf_imgs[i][0], cap=1024)
    train.append(d[0, :int(c[0])].cpu().numpy())
ctx.bow_set_vocabulary(*V.build_vocabulary(train, k=8, depth=3))
lc = flvis_amd.LoopCloser(ctx, cfg, LC.LC_PARAMS, n_streams=1, max_keyframes=max(n_kf, 1))
events = []
for (i0, i1), T in zip(kf_imgs, kf_pose):
    lc.add_keyframes([0], i0, i1, [T])
    events.append(lc.process()[0])
after = lc.poses(0)


def centres(T):
    return np.array([-(G.pose7_to_Rt(x)[0].T @ G.pose7_to_Rt(x)[1]) for x in T])


def ate(est, gt):
    """RMSE after the best rigid alignment (Horn, no scale)"""
    a, b = est - est.mean(0), gt - gt.mean(0)
    U, _, Vt = np.linalg.svd(a.T @ b)
    D = np.diag([1, 1, np.sign(np.linalg.det(U @ Vt))])
    R = (U @ D @ Vt).T
    return float(np.sqrt((((R @ a.T).T - b) ** 2).sum(1).mean()))


gt = np.array([-(tr.T_c_w(t, rnd.rig)[0].T @ tr.T_c_w(t, rnd.rig)[1]) for t in kf_time])
acc = [e for e in events if e["accepted"]]
print(json.dumps({
    "frames": n_frames, "tracking_frames": int(sum(s == 1 for s in states)), "keyframes": n_kf,
    "candidates": int(sum(e["candidate"] for e in events)), "loops_accepted": len(acc),
    "loops": [(e["kf_prev"], e["kf_curr"], e["n_matches"], e["n_inliers"]) for e in acc][:8],
    "pose_graph_runs": int(sum(e["optimised"] for e in events)),
    "ate_keyframes_m_tracker": ate(centres(np.array(kf_pose)), gt), "ate_keyframes_m_loop_closed": ate(centres(after), gt),
    "end_to_start_gap_m_tracker": float(np.linalg.norm((centres(np.array(kf_pose)) - gt)[-1] - (centres(np.array(kf_pose)) - gt)[0])),
    "end_to_start_gap_m_loop_closed": float(np.linalg.norm((centres(after) - gt)[-1] - (centres(after) - gt)[0])),
    "T_odom_map": [round(float(x), 5) for x in lc.drift(0)]}))
