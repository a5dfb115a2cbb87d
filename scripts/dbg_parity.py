import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import flvis_amd
from flvis_amd import synth
import _oracle as O
import test_gpu_pipeline as T
cfg, ocfg = T._cfgs()
ctx = flvis_amd.Context(0)
S, nframes = 2, int(sys.argv[1]) if len(sys.argv) > 1 else 75
streams = [3, 140]
trajs = [synth.Trajectory(s) for s in streams]
rnd = synth.Renderer("cuda")
seed_base = 0xF1715
trk = flvis_amd.Tracker(ctx, cfg, S, seed_base=seed_base, traj_capacity=nframes)
refs = [O.Tracker(ocfg, seed_base + i) for i in range(S)]
t_prev = -0.05; dp = 0.0
for f in range(nframes):
    t = f / synth.FRAME_HZ
    for i, s in enumerate(streams):
        smp = synth.imu_samples(trajs[i], s, t_prev, t)
        trk.imu_feed_flvis(i, smp)
        for r in smp: refs[i].imu(r[0], r[1:4], r[4:7])
    t_prev = t
    i0, i1 = rnd.stereo_frame(trajs, t, f)
    outs = trk.image_feed(i0, i1, [t] * S, with_local_map=False)
    h0, h1 = i0.cpu().numpy(), i1.cpu().numpy()
    for i in range(S):
        want = refs[i].image(t, h0[i], h1[i]); got = outs[i]
        if want["state"] != 1: continue
        gl, wl = trk.landmarks(i), refs[i].landmarks()
        same_ids = np.array_equal(gl["ids"], wl["ids"]); same_fl = np.array_equal(gl["flags"], wl["flags"])
        dp = np.abs(got["pose7"] - want["pose7"]).max()
        if not (same_ids and same_fl and np.array_equal(got["dbg"], want["dbg"])):
            print("frame", f, "stream", i, "ids", same_ids, "flags", same_fl, "dbg", got["dbg"], want["dbg"], "dpose %.2e" % dp)
            if same_ids and not same_fl:
                k = np.nonzero(gl["flags"] != wl["flags"])[0]
                print("  flag diff at", k, gl["ids"][k], gl["flags"][k], wl["flags"][k], "n", len(gl["ids"]), "p2d", gl["p2d"][k], wl["p2d"][k], "p3w", gl["p3w"][k], wl["p3w"][k])
            if not same_ids:
                sg, sw = set(gl["ids"].tolist()), set(wl["ids"].tolist())
                print("  only gpu", sorted(sg - sw), "only ref", sorted(sw - sg))
            sys.exit(0)
    if f % 10 == 0: print("frame", f, "ok dpose %.2e" % dp)
print("all equal")
