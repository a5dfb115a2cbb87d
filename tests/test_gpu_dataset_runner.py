"""GPU: scripts/run_sequence.py with the HIP backend on a KITTI odometry folder (type_of_vi 4, no IMU), loop closing switched on,
against the same run with the CPU backend (the oracle): BASELINE.json configs[1..2] "on identical inputs" for the dataset path --
the trajectory files agree, and so do the keyframe paths the loop closing maintains."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from test_dataset_runner import ROOT, make_asl_folder, make_kitti_folder, make_vocabulary_file

pytestmark = pytest.mark.gpu


def test_run_sequence_hip_backend_matches_the_cpu_backend_with_loop_closing():
    from flvis_amd import traj_io
    root, yaml, imgs = make_kitti_folder(9)
    voc = make_vocabulary_file(root, imgs[0][0])
    res, files = {}, {}
    for backend in ("cpu", "hip"):
        out, lc_out = os.path.join(root, "traj_%s.txt" % backend), os.path.join(root, "kf_%s.txt" % backend)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "run_sequence.py"), root, yaml, out, "--backend", backend,
                            "--loop-closing", "--voc", voc, "--lc-out", lc_out], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=400)
        assert r.returncode == 0, (backend, r.stderr.decode()[-2000:])
        res[backend] = json.loads(r.stdout.decode().strip().splitlines()[-1])
        files[backend] = (traj_io.read_stamped(out), traj_io.read_stamped(lc_out))
    assert res["hip"]["tracked"] == res["cpu"]["tracked"] == 9
    assert res["hip"]["loop_closing"]["keyframes"] == res["cpu"]["loop_closing"]["keyframes"] >= 1
    assert abs(res["hip"]["ate_rmse_m"] - res["cpu"]["ate_rmse_m"]) < 1e-6 and res["hip"]["ate_rmse_m"] < 0.01
    for which in (0, 1):                                   # the tracker's trajectory, the loop closing's keyframe path
        (ta, pa, qa), (tb, pb, qb) = files["hip"][which], files["cpu"][which]
        assert np.allclose(ta, tb, atol=1e-9) and np.abs(np.asarray(pa) - np.asarray(pb)).max() < 1e-6
        assert np.abs(np.abs(np.sum(np.asarray(qa) * np.asarray(qb), axis=1)) - 1).max() < 5e-6     # (the files carry six decimals)


def test_run_sequence_imu_pose_trajectory_hip_vs_cpu():
    """EuRoC ASL folder (stereo + IMU): the IMU-rate trajectory of F2FTracking::imu_feed's outputs -- /imu_pose, what the reference's
    EuRoC launch file records as est.txt and scores -- written by both backends: same lines (the files carry six significant digits),
    same ATE against the ground truth."""
    from flvis_amd import traj_io
    root, yaml, frames, imu_sensor, _, _ = make_asl_folder(17)
    res, files = {}, {}
    for backend in ("cpu", "hip"):
        out, imu_out = os.path.join(root, "traj_%s.txt" % backend), os.path.join(root, "imu_%s.txt" % backend)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "run_sequence.py"), root, yaml, out, "--backend", backend,
                            "--imu-out", imu_out], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=400)
        assert r.returncode == 0, (backend, r.stderr.decode()[-2000:])
        res[backend] = json.loads(r.stdout.decode().strip().splitlines()[-1])
        files[backend] = traj_io.read_stamped(imu_out)
    assert res["hip"]["tracked"] == res["cpu"]["tracked"] >= 6
    (ta, pa, qa), (tb, pb, qb) = files["hip"], files["cpu"]
    assert len(ta) == len(tb) == len(imu_sensor) and np.allclose(ta, tb, atol=1e-9)
    assert np.abs(np.asarray(pa) - np.asarray(pb)).max() < 1e-6
    assert np.abs(np.abs(np.sum(np.asarray(qa) * np.asarray(qb), axis=1)) - 1).max() < 5e-6
    assert abs(res["hip"]["ate_rmse_m_imu_pose"] - res["cpu"]["ate_rmse_m_imu_pose"]) < 1e-6
