"""GPU: flvis_loop_closer (the loop-closing nodelet's control flow around the device kernels, keyframe database resident in HBM)
against the same control flow assembled from the CPU oracle's functions (tests/_loop_chain.py), for two sequences at once that both
return to where they started.  The per-keyframe features the oracle chain works on are the device's (each kernel has its own
parity test in test_gpu_orb.py / test_gpu_loop.py); compared here is everything downstream: similarity rows, candidates, matches,
PnP poses and inliers, accepted loops, when the pose graph is optimised, the optimised poses and the map -> odom correction."""
import os
import tempfile

import numpy as np
import pytest

import _geom as G
import _loop_chain as LC
import _pgo_synth as PS
import _voc as V
from test_oracle_bow import ref_score

pytestmark = pytest.mark.gpu


def test_loop_closer_two_sequences_against_the_oracle_chain():
    import torch
    import flvis_amd
    from flvis_amd import synth
    ctx = flvis_amd.Context(0)
    p = os.path.join(tempfile.gettempdir(), "flvis_loopcloser_gpu.yaml")
    open(p, "w").write(synth.D435I_STEREO_YAML)
    cfg = flvis_amd.load_config(p)
    P0, P1 = np.array(list(cfg.P0)), np.array(list(cfg.P1))
    K4 = np.array([P0[0], P0[5], P0[2], P0[6]])
    trs = [LC.LoopTrajectory(phase=0.0), LC.LoopTrajectory(phase=0.9)]
    rnd = synth.Renderer("cuda")
    n_kf, per = 62, 50
    times = LC.keyframe_times(n_kf, per)
    frames = [rnd.stereo_frame(trs, t, i) for i, t in enumerate(times)]
    gt = [[G.pose7(*tr.T_c_w(t, rnd.rig)) for t in times] for tr in trs]
    odom = [LC.drifted_odometry(gt[s], 10 + s, sigma_t=0.008, sigma_r=0.002) for s in range(2)]
    # vocabulary from the device's descriptors of every sixth keyframe of sequence 0
    train = []
    for i in range(0, n_kf, 6):
        k, d, c, _ = ctx.orb_detect_and_compute(frames[i][0][0:1], cap=1024)
        train.append(d[0, :int(c[0])].cpu().numpy())
    voc = V.build_vocabulary(train, k=8, depth=3)
    ctx.bow_set_vocabulary(*voc)
    with pytest.raises(flvis_amd.FlvisError):                                   # capacity is checked, nothing is overwritten
        flvis_amd.LoopCloser(ctx, cfg, LC.LC_PARAMS, n_streams=2, max_keyframes=0)
    lc = flvis_amd.LoopCloser(ctx, cfg, LC.LC_PARAMS, n_streams=2, max_keyframes=64)
    ref = [LC.RefLoopCloser(K4, stream=s) for s in range(2)]
    n_added = [0, 0]
    log = [[], []]
    for i in range(n_kf):
        i0, i1 = frames[i]
        # sequence 1 misses every ninth call (keyframes of different sequences do not arrive in step)
        streams = [0] if i % 9 == 4 else [0, 1]
        sel = torch.tensor(streams, device="cuda")
        a0, a1 = i0[sel].contiguous(), i1[sel].contiguous()
        T = np.array([odom[s][n_added[s]] if s == 0 else odom[s][i] for s in streams])
        ids = lc.add_keyframes(streams, a0, a1, T)
        assert ids.tolist() == [n_added[s] for s in streams]
        # the same keyframes through the separate entry points, for the oracle chain
        kps, desc, cnt, _ = ctx.orb_detect_and_compute(a0, cap=1024)
        bi, bv, bn = [t.cpu().numpy() for t in ctx.bow_transform(desc, cnt, vcap=1024)]
        lm2, lm3, lmd, lmc = [t.cpu().numpy() for t in ctx.lc_keyframe_landmarks(a0, a1, 0, kps, desc, cnt, P0=P0, P1=P1)]
        for j, s in enumerate(streams):
            ref[s].add(dict(bow=(bi[j, :bn[j]].copy(), bv[j, :bn[j]].copy()), lm2=lm2[j, :lmc[j]].copy(), lm3=lm3[j, :lmc[j]].copy(),
                            lmd=lmd[j, :lmc[j]].copy()), T[j])
            n_added[s] += 1
            if i % 20 == 3:     # what the database holds for this keyframe is what the separate entry points computed
                kf = lc.keyframe(s, n_added[s] - 1)
                f = ref[s].kfs[-1]
                assert np.array_equal(kf["lm2"], f["lm2"]) and np.array_equal(kf["lm3"], f["lm3"]) and np.array_equal(kf["lmd"], f["lmd"])
                assert np.array_equal(kf["bow"][0], f["bow"][0]) and np.array_equal(kf["bow"][1], f["bow"][1])
        ev = lc.process()
        for s in range(2):
            if s not in streams:
                assert ev[s]["kf_curr"] == -1 and not ev[s]["candidate"]
                continue
            want = ref[s].process()
            got = ev[s]
            row = lc.similarity_row(s)
            assert np.array_equal(row, ref[s].rows[-1]), (i, s, np.abs(row - ref[s].rows[-1]).max())
            for key in ("kf_curr", "kf_prev", "candidate", "n_matches", "n_inliers", "accepted", "optimised"):
                assert got[key] == want[key], (i, s, key, got, want)
            if want["pose"] is not None:
                assert np.array_equal(np.array(got["pose"]), want["pose"]), (i, s)
            log[s].append(got)
        for s in streams:
            Tg, Tw = lc.poses(s), np.array(ref[s].T_c_w)
            assert Tg.shape == Tw.shape and np.abs(Tg - Tw).max() < 1e-7, (i, s, np.abs(Tg - Tw).max())
            assert np.abs(lc.drift(s) - ref[s].T_odom_map).max() < 1e-7
    for s in range(2):
        n = n_added[s]
        closing = [e for e in log[s] if e["accepted"] and e["kf_curr"] - e["kf_prev"] >= 40]
        assert len(closing) >= 2 and any(e["optimised"] for e in log[s]), (s, [(e["kf_prev"], e["kf_curr"]) for e in log[s] if e["accepted"]])
        od = np.array(odom[s][:n] if s == 0 else [odom[s][i] for i in range(n_kf) if i % 9 != 4])
        g = np.array(gt[s][:n] if s == 0 else [gt[s][i] for i in range(n_kf) if i % 9 != 4])
        gap0, gap1 = PS.loop_gap(od, g, 2, n - 1), PS.loop_gap(lc.poses(s), g, 2, n - 1)
        assert gap1[0] < 0.5 * gap0[0], (s, gap0, gap1)
    # the same keyframes handed over as HOST images with padded rows (what the nodelet unpacks from KeyFrame.msg): same similarity row
    hostlc = flvis_amd.LoopCloser(ctx, cfg, LC.LC_PARAMS, n_streams=1, max_keyframes=4)
    for k in range(3):
        pad = [np.zeros((480, 704), np.uint8) for _ in range(2)]
        pad[0][:, :640], pad[1][:, :640] = frames[k][0][0].cpu().numpy(), frames[k][1][0].cpu().numpy()
        assert hostlc.add_keyframes_host([0], [pad[0][:, :640]], [pad[1][:, :640]], [odom[0][k]]).tolist() == [k]
        hostlc.process()
    assert np.array_equal(hostlc.similarity_row(0), ref[0].rows[2])
    assert np.abs(hostlc.poses(0) - np.array(odom[0][:3])).max() < 1e-12        # T_odom_map is still the identity
    hostlc.close()
    # a full sequence refuses the next keyframe instead of overwriting
    small = flvis_amd.LoopCloser(ctx, cfg, LC.LC_PARAMS, n_streams=1, max_keyframes=2)
    for k in range(2):
        small.add_keyframes([0], frames[k][0][0:1], frames[k][1][0:1], [odom[0][k]])
    with pytest.raises(flvis_amd.FlvisError) as e:
        small.add_keyframes([0], frames[2][0][0:1], frames[2][1][0:1], [odom[0][2]])
    assert "capacity" in str(e.value)
    with pytest.raises(flvis_amd.FlvisError):
        lc.add_keyframes([0, 0], frames[0][0], frames[0][1], [odom[0][0], odom[0][0]])   # two keyframes for one sequence in one call
    small.close()
    lc.close()
    ctx.close()


def test_loop_closer_on_a_depth_camera_rig():
    """DEPTH_D435 (vo_loopclosing.cpp:325-349): the keyframes' landmarks come from the Z16 image (whole metres, as the reference's integer
    division leaves them); the database holds what flvis_hip_lc_keyframe_landmarks computes for cam_type 2"""
    import flvis_amd
    from flvis_amd import synth
    ctx = flvis_amd.Context(0)
    p = os.path.join(tempfile.gettempdir(), "flvis_loopcloser_depth_gpu.yaml")
    open(p, "w").write(synth.D435I_DEPTH_YAML)
    cfg = flvis_amd.load_config(p)
    assert cfg.cam_type == 2
    K4 = np.array([cfg.P0[0], cfg.P0[5], cfg.P0[2], cfg.P0[6]])
    trs = [LC.LoopTrajectory(phase=0.0), LC.LoopTrajectory(phase=2.0)]
    rnd = synth.Renderer("cuda")
    frames = [rnd.depth_frame(trs, t, i) for i, t in enumerate(LC.keyframe_times(4, 50))]
    train = []
    for i0, _ in frames:                                     # eight training images (a word seen in ALL of them has idf 0 and is stopped)
        k, d, c, _ = ctx.orb_detect_and_compute(i0, cap=1024)
        train += [d[s, :int(c[s])].cpu().numpy() for s in range(2)]
    ctx.bow_set_vocabulary(*V.build_vocabulary(train, k=6, depth=3))
    lc = flvis_amd.LoopCloser(ctx, cfg, LC.LC_PARAMS, n_streams=2, max_keyframes=8)
    ident = np.array([[0, 0, 0, 0, 0, 0, 1.0]] * 2)
    for i0, d16 in frames:
        lc.add_keyframes([0, 1], i0, d16, ident)
        ev = lc.process()
        assert [e["kf_curr"] for e in ev] == [lc.poses(0).shape[0] - 1] * 2 and not any(e["candidate"] for e in ev)
    kps, desc, cnt, _ = ctx.orb_detect_and_compute(frames[3][0], cap=1024)
    lm2, lm3, lmd, lmc = [t.cpu().numpy() for t in ctx.lc_keyframe_landmarks(None, frames[3][1], 2, kps, desc, cnt, K4=K4)]
    for s in range(2):
        kf = lc.keyframe(s, 3)
        assert len(kf["lm2"]) == lmc[s] > 50
        assert np.array_equal(kf["lm2"], lm2[s, :lmc[s]]) and np.array_equal(kf["lm3"], lm3[s, :lmc[s]]) and np.array_equal(kf["lmd"], lmd[s, :lmc[s]])
        assert set(np.unique(kf["lm3"][:, 2])) <= set(float(v) for v in range(1, 11))
        row = lc.similarity_row(s)
        bows = [lc.keyframe(s, j)["bow"] for j in range(4)]
        assert len(bows[3][0]) > 20
        assert np.array_equal(row, [ref_score(bows[3], bows[j]) for j in range(4)]) and abs(row[3] - 1.0) < 1e-12
    with pytest.raises(flvis_amd.FlvisError):
        lc.keyframe(0, 4)
    lc.close()
    ctx.close()


def test_tracker_keyframes_through_the_loop_closer():
    """the three nodelets' work chained in one process (scripts/run_loop_demo.py): the tracker on a rendered stereo + IMU sequence,
    its keyframes into the loop closer; loops are found and verified and the corrected keyframe path is no worse than the tracker's"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "run_loop_demo.py"), "32", "60"], stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=240)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    r = json.loads(out.stdout.decode().strip().splitlines()[-1])
    assert r["tracking_frames"] >= r["frames"] - 60 and r["keyframes"] >= 60, r
    assert r["loops_accepted"] >= 3 and r["pose_graph_runs"] >= 1, r
    assert all(l[3] >= 20 and l[3] >= 0.5 * l[2] for l in r["loops"]), r            # the acceptance rule held for what was accepted
    # On this short tour the tracker has drifted by 4-5 cm when the loops close, and a verified loop pose is SOLVEPNP_P3P's unrefined
    # EPnP on ~100 ORB matches (a few cm of its own): the corrected path must stay in the tracker's range, not necessarily below it
    # (the 70 s tour of profiles/r02_loop_demo.json, with 13 cm of drift, gains 27 %).
    assert r["ate_keyframes_m_loop_closed"] <= 1.3 * r["ate_keyframes_m_tracker"] + 0.01, r
