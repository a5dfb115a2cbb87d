#!/usr/bin/env python3
"""flvis_loop_closer on one GPU: S sequences that each circle the rendered room and come back, one keyframe per sequence and call.
Times add_keyframes (ORB + bag of words + 3-D landmarks + store) and process (similarity row, candidates, verification of all
candidates at once, pose graphs) per batch with HIP events around the calls; prints one JSON line.

usage: loop_closer_bench.py [n_streams=64] [n_keyframes=60]"""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import flvis_amd
from flvis_amd import synth
import _geom as G
import _loop_chain as LC
import _pgo_synth as PS
import _voc as V

S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
PER = 50

ctx = flvis_amd.Context(0)
p = os.path.join(tempfile.gettempdir(), "flvis_loop_closer_bench.yaml")
open(p, "w").write(synth.D435I_STEREO_YAML)
cfg = flvis_amd.load_config(p)
trs = [LC.LoopTrajectory(phase=2 * np.pi * s / S) for s in range(S)]
rnd = synth.Renderer("cuda")
times = LC.keyframe_times(N, PER)
frames = [rnd.stereo_frame(trs, t, i) for i, t in enumerate(times)]
gt = [[G.pose7(*tr.T_c_w(t, rnd.rig)) for t in times] for tr in trs]
odom = [LC.drifted_odometry(gt[s], 100 + s, sigma_t=0.008, sigma_r=0.002) for s in range(S)]
train = []
for i in range(0, N, 6):
    k, d, c, _ = ctx.orb_detect_and_compute(frames[i][0][0:1], cap=1024)
    train.append(d[0, :int(c[0])].cpu().numpy())
ctx.bow_set_vocabulary(*V.build_vocabulary(train, k=8, depth=3))
lc = flvis_amd.LoopCloser(ctx, cfg, LC.LC_PARAMS, n_streams=S, max_keyframes=N)
streams = list(range(S))
t_add, t_proc, n_cand, n_acc, n_opt = [], [], [], [], []
torch.cuda.synchronize()
for i in range(N):
    T = np.array([odom[s][i] for s in range(S)])
    t0 = time.perf_counter()
    lc.add_keyframes(streams, frames[i][0], frames[i][1], T)          # returns after the batch is stored (it synchronises)
    t1 = time.perf_counter()
    ev = lc.process()                                                  # returns with the events on the host
    t2 = time.perf_counter()
    t_add.append((t1 - t0) * 1e3)
    t_proc.append((t2 - t1) * 1e3)
    n_cand.append(sum(e["candidate"] for e in ev))
    n_acc.append(sum(e["accepted"] for e in ev))
    n_opt.append(sum(e["optimised"] for e in ev))
gap0 = np.mean([PS.loop_gap(np.array(odom[s]), np.array(gt[s]), 2, N - 1)[0] for s in range(S)])
gap1 = np.mean([PS.loop_gap(lc.poses(s), np.array(gt[s]), 2, N - 1)[0] for s in range(S)])
quiet = [i for i in range(5, N) if n_cand[i] == 0]
busy = [i for i in range(N) if n_opt[i] > 0]
print(json.dumps({
    "n_streams": S, "n_keyframes": N,
    "add_keyframes_ms_per_batch": float(np.mean([t_add[i] for i in range(5, N)])),
    "process_ms_per_batch_no_candidate": float(np.mean([t_proc[i] for i in quiet])) if quiet else None,
    "process_ms_per_batch_with_pose_graphs": float(np.mean([t_proc[i] for i in busy])) if busy else None,
    "pose_graphs_per_busy_batch": float(np.mean([n_opt[i] for i in busy])) if busy else None,
    "candidates": int(sum(n_cand)), "loops_accepted": int(sum(n_acc)), "pose_graph_runs": int(sum(n_opt)),
    "mean_loop_gap_m_odometry": float(gap0), "mean_loop_gap_m_after": float(gap1),
    "timing": "host wall clock around calls that return synchronised"}))
