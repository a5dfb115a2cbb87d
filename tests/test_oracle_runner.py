"""oracle/ref_runner.cpp (the native driver behind bench.py's cpu_baseline leg): running streams on several host threads gives
exactly the poses of the ctypes-driven oracle the parity tests use, stream by stream."""
import ctypes as C
import os
import sys
import tempfile

import numpy as np

import _oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_native_runner_matches_the_ctypes_driven_oracle():
    import bench
    from flvis_amd import synth
    p = os.path.join(tempfile.gettempdir(), "flvis_test_runner.yaml")
    open(p, "w").write(synth.D435I_STEREO_YAML)
    cfg = O.load_config(p)
    T, first, n, SPF = 2, cfg.skip_first_n_imgs, 5, 16
    trajs = [synth.Trajectory(s) for s in (3, 4)]
    rnd = synth.Renderer("cpu")
    imu = np.zeros((first + n, T, SPF, 7))
    cnt = np.zeros((first + n, T), np.int32)
    for i, s in enumerate((3, 4)):
        t_prev = -1.0 / synth.FRAME_HZ
        for f in range(first + n):
            smp = synth.imu_samples(trajs[i], s, t_prev, f / synth.FRAME_HZ)
            imu[f, i, :len(smp)] = smp
            cnt[f, i] = len(smp)
            t_prev = f / synth.FRAME_HZ
    frames = []
    for j in range(n):
        i0, i1 = rnd.stereo_frame(trajs, (first + j) / synth.FRAME_HZ, first + j)
        frames.append((i0.numpy(), i1.numpy()))
    # ctypes-driven, one stream after the other
    want_pose, want_state = [], []
    blank = np.zeros((cfg.image_height, cfg.image_width), np.uint8)
    for i in range(T):
        trk = O.Tracker(cfg, 100 + i)
        lm = O.LocalMap(cfg.window_size, np.array([cfg.P0[0], cfg.P0[5], cfg.P0[2], cfg.P0[6]]))
        ps, ss = [], []
        for f in range(first + n):
            for r in imu[f, i, :cnt[f, i]]:
                trk.imu(r[0], r[1:4], r[4:7])
            if f < first:
                trk.image(f / synth.FRAME_HZ, blank, blank)
                continue
            res = trk.image(f / synth.FRAME_HZ, frames[f - first][0][i], frames[f - first][1][i])
            ps.append(res["pose7"])
            ss.append(res["state"])
            if res["new_keyframe"]:
                kf = trk.keyframe()
                lm.push(kf["frame_id"], kf["pose7"], kf["lm_id"], kf["lm_2d"], kf["lm_3d"])
        want_pose.append(ps)
        want_state.append(ss)
    for threads in (1, 2):
        secs, poses, states, fms = bench.cpu_run_streams(O.lib(), cfg, T, threads, first, n, frames, imu, cnt, 100, synth.FRAME_HZ, 1)
        assert secs > 0 and np.all(fms > 0)
        assert np.array_equal(states, np.array(want_state))
        assert np.array_equal(poses, np.array(want_pose))          # same binary, same inputs: bit-identical
    assert states[0][0] == 1                                        # init_frame on the first processed frame
    assert O.lib().ref_hardware_threads() >= 1
    # bad arguments are refused, not dereferenced
    O.lib().ref_run_streams.restype = C.c_double
    assert O.lib().ref_run_streams(None, 1, 1, 0, C.c_double(20.0), None, None, None, None, 16, None, 0, 1, None, None, None) < 0
