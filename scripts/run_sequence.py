#!/usr/bin/env python3
"""Runs one EuRoC ASL sequence (stereo + IMU) or KITTI odometry sequence (image_0/ image_1/, no IMU: type_of_vi 4) through the tracker and writes the trajectory the reference's recorder would write
(`stamp x y z qw qx qy qz`, camera pose T_w_c); with ground truth present, prints the Umeyama-aligned ATE.

  run_sequence.py <sequence folder> <config yaml> <out.txt> [--backend hip|cpu] [--frames N] [--local-map] [--imu-out <est.txt>]
                  [--loop-closing --voc <DBoW3 vocabulary file> [--lc-out <keyframes.txt>]]

--imu-out : rigs with an IMU: additionally the IMU-rate trajectory of F2FTracking::imu_feed's outputs (pos_w_i, q_w_i per sample) -- the
/imu_pose topic, which is what the reference's EuRoC launch file records as est.txt (launch/flvis_euroc_mav.launch:83-103) and scores;
with ground truth present its ATE is printed as ate_rmse_m_imu_pose (the body frame itself: no camera-to-body step).

--backend hip : the product (flvis_amd, needs an MI355X)          -- BASELINE.json configs[1..2] on real data
--backend cpu : the CPU restatement under oracle/ (test infrastructure) -- configs[0], "the reference CPU path"
Comparing the two output files with flvis_amd.traj_io.ate_from_files gives the metric's "ATE vs CPU ref".

--loop-closing : the tracker's keyframes additionally go through the loop closing (vo_loopclosing.cpp: the third nodelet of the launch
files) with the vocabulary of --voc (.dbow3 / .txt / .yml[.gz]) and the lcKF* / ratio* / min* block of the yaml; the keyframe path it
maintains (T_w_c of every keyframe, corrected by the pose graph whenever a loop closes) is written to --lc-out in the same format.
hip: flvis_loop_closer; cpu: the same control flow assembled from the oracle's functions (tests/_loop_chain.py)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from flvis_amd import traj_io  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("sequence")
    ap.add_argument("config")
    ap.add_argument("out")
    ap.add_argument("--backend", choices=["hip", "cpu"], default="hip")
    ap.add_argument("--frames", type=int, default=None)
    ap.add_argument("--local-map", action="store_true")
    ap.add_argument("--loop-closing", action="store_true")
    ap.add_argument("--voc", default=None)
    ap.add_argument("--lc-out", default=None)
    ap.add_argument("--imu-out", default=None)
    args = ap.parse_args()
    if args.loop_closing and not args.voc:
        ap.error("--loop-closing needs --voc <vocabulary file>")
    seq = traj_io.open_sequence(args.sequence)
    kitti = isinstance(seq, traj_io.KittiSequence)
    stamps, pos, quat = [], [], []
    imu_rows = []            # (t, q_w_i wxyz, pos_w_i, vel_w_i) per IMU sample
    if args.backend == "cpu":
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import _oracle as O
        cfg = O.load_config(args.config)
        imu_type = {1: 1, 3: 0, 5: 2, 0: 0, 2: 2, 4: 3}[cfg.type_of_vi]
        trk = O.Tracker(cfg, 0xF1715)
        closer, lc_events, kf_stamps = None, [], []
        if args.loop_closing:
            import flvis_amd
            import _loop_chain as LC
            from test_oracle_bow import RefVoc
            v = flvis_amd.read_vocabulary_file(args.voc)              # (host-side file reader of the product library)
            rv = RefVoc((v["child_ptr"], v["child_idx"], v["desc"], v["weight"], v["word_id"]))
            prm = flvis_amd.load_lc_params(args.config)
            P0, P1 = np.array(list(cfg.P0)), np.array(list(cfg.P1))
            K4 = np.array([P0[0], P0[5], P0[2], P0[6]])
            closer = LC.RefLoopCloser(K4, {k: getattr(prm, k) for k, _ in prm._fields_})
        for t, i0, i1, imu in seq.frames(0, args.frames):
            for r in imu:
                a, g = traj_io.sensor_to_flvis_imu(imu_type, r[4:7], r[1:4])
                imu_rows.append(np.concatenate([[r[0]], trk.imu(r[0], a, g)]))
            res = trk.image(t, i0, i1)
            if closer is not None and res["new_keyframe"]:
                k, d = O.orb_detect_and_compute(i0)
                lm2, lm3, lmd = O.lc_keyframe_landmarks(i0, i1, cfg.cam_type, k, d, P0, P1, K4)
                closer.add(dict(bow=rv.transform(d), lm2=lm2, lm3=lm3, lmd=lmd), res["pose7"])
                lc_events.append(closer.process())
                kf_stamps.append(t)
            if res["state"] == 1:
                stamps.append(t)
                p7 = res["pose7"]
                R = traj_io.quat_to_rot(p7[6], p7[3], p7[4], p7[5])           # T_c_w
                pos.append(-R.T @ p7[:3])
                quat.append(traj_io.rot_to_quat(R.T))
        traj_io.write_stamped(args.out, stamps, pos, quat)
        imu_rows = np.array(imu_rows).reshape(-1, 11)
        if args.imu_out and len(imu_rows):
            traj_io.write_stamped(args.imu_out, imu_rows[:, 0], imu_rows[:, 5:8], imu_rows[:, 1:5])
        T_imu_cam = np.array(list(cfg.T_imu_cam0)).reshape(4, 4)
        kf_T_c_w = np.array(closer.T_c_w).reshape(-1, 7) if closer is not None else None
    else:
        import torch
        import flvis_amd
        cfg = flvis_amd.load_config(args.config)
        ctx = flvis_amd.Context(0)
        n = len(seq) if args.frames is None else min(len(seq), args.frames)
        trk = flvis_amd.Tracker(ctx, cfg, 1, traj_capacity=n)
        closer, lc_events, kf_stamps = None, [], []
        if args.loop_closing:
            ctx.bow_load_vocabulary(args.voc)                                  # Vocabulary voc(path), vo_loopclosing.cpp:1097
            closer = flvis_amd.LoopCloser(ctx, cfg, flvis_amd.load_lc_params(args.config), n_streams=1, max_keyframes=max(n, 1))
        for t, i0, i1, imu in seq.frames(0, n):
            for r in imu:
                trk.imu_feed_sensor(0, r[0], r[4:7], r[1:4])                   # the library applies the axis remap
            d0, d1 = torch.from_numpy(i0[None]).cuda(), torch.from_numpy(i1[None]).cuda()
            res = trk.image_feed(d0, d1, [t], want_out=closer is not None, with_local_map=args.local_map)
            if args.imu_out and len(imu):
                imu_rows.append(trk.imu_states(0)[0])
            if closer is not None and res[0]["new_keyframe"]:
                closer.add_keyframes([0], d0, d1, [res[0]["pose7"]])
                lc_events.append(closer.process()[0])
                kf_stamps.append(t)
        trk.write_trajectory(0, 0, n, args.out, 0)
        imu_rows = np.concatenate(imu_rows) if imu_rows else np.zeros((0, 11))
        if args.imu_out and len(imu_rows):
            trk.write_imu_trajectory(imu_rows, args.imu_out)
        kf_T_c_w = closer.poses(0) if closer is not None else None
        T_imu_cam = np.array(list(cfg.T_imu_cam0)).reshape(4, 4)
        stamps, pos, quat = traj_io.read_stamped(args.out)
    out = {"backend": args.backend, "frames": len(seq) if args.frames is None else args.frames, "tracked": len(stamps)}
    if args.loop_closing:
        kf_pos, kf_quat = [], []
        for p7 in kf_T_c_w:
            R = traj_io.quat_to_rot(p7[6], p7[3], p7[4], p7[5])
            kf_pos.append(-R.T @ p7[:3])
            kf_quat.append(traj_io.rot_to_quat(R.T))
        if args.lc_out:
            traj_io.write_stamped(args.lc_out, kf_stamps, kf_pos, kf_quat)
        out["loop_closing"] = {"keyframes": len(kf_stamps), "candidates": int(sum(bool(e["candidate"]) for e in lc_events)),
                               "loops_accepted": int(sum(bool(e["accepted"]) for e in lc_events)),
                               "pose_graph_runs": int(sum(bool(e["optimised"]) for e in lc_events))}
    if seq.groundtruth is not None and len(stamps) >= 3:
        gt_t, gt_p, _ = seq.groundtruth
        if kitti:   # KITTI ground truth is the camera itself
            body_p = np.asarray(pos)
        else:
            body_p, _ = traj_io.camera_to_body(np.asarray(pos), np.asarray(quat), T_imu_cam)
        ia, ib = traj_io.associate(np.asarray(stamps), gt_t, 0.02)
        if len(ia) >= 3:
            out["ate_rmse_m"] = traj_io.ate_rmse(body_p[ia], gt_p[ib])
            out["associated"] = int(len(ia))
        if args.imu_out and len(imu_rows) >= 3:       # /imu_pose: the trajectory the reference records and scores on EuRoC
            ja, jb = traj_io.associate(imu_rows[:, 0], gt_t, 0.02)
            if len(ja) >= 3:
                out["ate_rmse_m_imu_pose"] = traj_io.ate_rmse(imu_rows[ja, 5:8], gt_p[jb])
                out["associated_imu_pose"] = int(len(ja))
        if args.loop_closing and len(kf_stamps) >= 3:
            kp = np.asarray(kf_pos) if kitti else traj_io.camera_to_body(np.asarray(kf_pos), np.asarray(kf_quat), T_imu_cam)[0]
            ka, kb = traj_io.associate(np.asarray(kf_stamps), gt_t, 0.02)
            if len(ka) >= 3:
                out["loop_closing"]["ate_rmse_m_keyframe_path"] = traj_io.ate_rmse(kp[ka], gt_p[kb])
    print(json.dumps(out))


if __name__ == "__main__":
    main()
