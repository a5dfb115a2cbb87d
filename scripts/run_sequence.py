#!/usr/bin/env python3
"""Runs one EuRoC ASL sequence (stereo + IMU) or KITTI odometry sequence (image_0/ image_1/, no IMU: type_of_vi 4) through the tracker and writes the trajectory the reference's recorder would write
(`stamp x y z qw qx qy qz`, camera pose T_w_c); with ground truth present, prints the Umeyama-aligned ATE.

  run_sequence.py <sequence folder> <config yaml> <out.txt> [--backend hip|cpu] [--frames N] [--local-map]

--backend hip : the product (flvis_amd, needs an MI355X)          -- BASELINE.json configs[1..2] on real data
--backend cpu : the CPU restatement under oracle/ (test infrastructure) -- configs[0], "the reference CPU path"
Comparing the two output files with flvis_amd.traj_io.ate_from_files gives the metric's "ATE vs CPU ref"."""
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
    args = ap.parse_args()
    seq = traj_io.open_sequence(args.sequence)
    kitti = isinstance(seq, traj_io.KittiSequence)
    stamps, pos, quat = [], [], []
    if args.backend == "cpu":
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import _oracle as O
        cfg = O.load_config(args.config)
        imu_type = {1: 1, 3: 0, 5: 2, 0: 0, 2: 2, 4: 3}[cfg.type_of_vi]
        trk = O.Tracker(cfg, 0xF1715)
        for t, i0, i1, imu in seq.frames(0, args.frames):
            for r in imu:
                a, g = traj_io.sensor_to_flvis_imu(imu_type, r[4:7], r[1:4])
                trk.imu(r[0], a, g)
            res = trk.image(t, i0, i1)
            if res["state"] == 1:
                stamps.append(t)
                p7 = res["pose7"]
                R = traj_io.quat_to_rot(p7[6], p7[3], p7[4], p7[5])           # T_c_w
                pos.append(-R.T @ p7[:3])
                quat.append(traj_io.rot_to_quat(R.T))
        traj_io.write_stamped(args.out, stamps, pos, quat)
        T_imu_cam = np.array(list(cfg.T_imu_cam0)).reshape(4, 4)
    else:
        import torch
        import flvis_amd
        cfg = flvis_amd.load_config(args.config)
        ctx = flvis_amd.Context(0)
        n = len(seq) if args.frames is None else min(len(seq), args.frames)
        trk = flvis_amd.Tracker(ctx, cfg, 1, traj_capacity=n)
        for t, i0, i1, imu in seq.frames(0, n):
            for r in imu:
                trk.imu_feed_sensor(0, r[0], r[4:7], r[1:4])                   # the library applies the axis remap
            trk.image_feed(torch.from_numpy(i0[None]).cuda(), torch.from_numpy(i1[None]).cuda(), [t],
                           want_out=False, with_local_map=args.local_map)
        trk.write_trajectory(0, 0, n, args.out, 0)
        T_imu_cam = np.array(list(cfg.T_imu_cam0)).reshape(4, 4)
        stamps, pos, quat = traj_io.read_stamped(args.out)
    out = {"backend": args.backend, "frames": len(seq) if args.frames is None else args.frames, "tracked": len(stamps)}
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
    print(json.dumps(out))


if __name__ == "__main__":
    main()
