#!/usr/bin/env python3
"""Multi-threaded CPU baseline: T independent streams of the synthetic workload through the CPU restatement (oracle front-end
+ local-map BA), one host thread per stream (ctypes releases the GIL).  Test infrastructure, like bench.py's cpu_baseline leg.
usage: cpu_baseline_mt.py [threads=16] [frames=40]"""
import json
import os
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import _oracle as O  # noqa: E402
from flvis_amd import synth  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 16
NF = int(sys.argv[2]) if len(sys.argv) > 2 else 40
p = os.path.join(tempfile.gettempdir(), "flvis_cpu_mt.yaml")
open(p, "w").write(synth.D435I_STEREO_YAML)
cfg = O.load_config(p)
skip = cfg.skip_first_n_imgs
K4 = np.array([cfg.P0[0], cfg.P0[5], cfg.P0[2], cfg.P0[6]])
rnd = synth.Renderer("cpu")
trajs = [synth.Trajectory(s) for s in range(T)]
frames, imus = [], []
t_prev = -0.05
for f in range(skip + NF):
    t = f / synth.FRAME_HZ
    imus.append([synth.imu_samples(trajs[s], s, t_prev, t) for s in range(T)])
    t_prev = t
    if f >= skip:
        i0, i1 = rnd.stereo_frame(trajs, t, f)
        frames.append((i0.numpy(), i1.numpy()))
trk = [O.Tracker(cfg, 0xF1715 + s) for s in range(T)]
lmap = [O.LocalMap(cfg.window_size, K4) for s in range(T)]
blank = np.zeros((cfg.image_height, cfg.image_width), np.uint8)
for f in range(skip):   # untimed: the skipped start-up frames carry no vision work
    for s in range(T):
        for r in imus[f][s]:
            trk[s].imu(r[0], r[1:4], r[4:7])
        trk[s].image(f / synth.FRAME_HZ, blank, blank)
tracked = [0] * T


def worker(s):
    for j in range(NF):
        f = skip + j
        for r in imus[f][s]:
            trk[s].imu(r[0], r[1:4], r[4:7])
        res = trk[s].image(f / synth.FRAME_HZ, frames[j][0][s], frames[j][1][s])
        tracked[s] += res["state"] == 1
        if res["new_keyframe"]:
            kf = trk[s].keyframe()
            lmap[s].push(kf["frame_id"], kf["pose7"], kf["lm_id"], kf["lm_2d"], kf["lm_3d"])


ths = [threading.Thread(target=worker, args=(s,)) for s in range(T)]
t0 = time.perf_counter()
for th in ths:
    th.start()
for th in ths:
    th.join()
dt = time.perf_counter() - t0
print(json.dumps({"threads": T, "frames_per_stream": NF, "frames_per_s": T * NF / dt, "per_thread_fps": NF / dt,
                  "host_cores": os.cpu_count(), "tracked_frames_min": min(tracked)}))
