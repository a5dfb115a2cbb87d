"""Generates tests/golden/*.npz from the CPU oracle on seeded synthetic inputs.

The reference ships no golden vectors for this path, so these fixtures pin the ORACLE itself: both the
oracle (CPU tests) and the HIP path (GPU tests) are compared against them, which catches silent drifts of either side.
Regenerate with:  python tests/golden/make_golden.py   (inputs are stored in the fixture, not regenerated at test time)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _ba_synth as B  # noqa: E402
import _oracle as O  # noqa: E402
import _synth as S  # noqa: E402


def main():
    i0, i1 = S.shifted_pair(96, 128, 41, 2.4, -1.3)
    eq = O.equalize_hist((i0 // 2 + 40).astype(np.uint8))
    pd = O.pyr_down(i0)
    corners = O.gftt(i0, 60, 0.01, 5)
    pts = corners[:40]
    init = pts + np.float32([1.7, -0.9])
    trk, st = O.lk(i0, i1, pts, init, max_level=3)
    fp = [4, 30, 5, 60, 0.01, 5]
    dem = O.dem_detect(i0, fp)
    red = O.dem_redetect(i0, fp, dem[::3].astype(np.float64) + 0.25)
    np.savez_compressed(os.path.join(HERE, "image_stages.npz"), img0=i0, img1=i1, eq_in=(i0 // 2 + 40).astype(np.uint8),
                        eq_out=eq, pyr=pd, gftt=corners, lk_prev=pts, lk_init=init, lk_next=trk, lk_status=st,
                        dem_para=np.array(fp, np.float64), dem_detect=dem, dem_exist=dem[::3].astype(np.float64) + 0.25,
                        dem_redetect=red)
    seq = B.make_sequence(21, n_kf=11, n_lm=140, outlier_frac=0.03)
    lm = O.LocalMap(8, B.K4)
    outs = {}
    for k, kf in enumerate(seq["kfs"]):
        r = lm.push(kf["frame_id"], kf["pose7"], kf["lm_id"], kf["lm_2d"], kf["lm_3d"])
        outs["kf%d_pose" % k] = kf["pose7"]
        outs["kf%d_id" % k] = kf["lm_id"]
        outs["kf%d_2d" % k] = kf["lm_2d"]
        outs["kf%d_3d" % k] = kf["lm_3d"]
        outs["kf%d_frame" % k] = np.int64(kf["frame_id"])
        if r is not None:
            outs["out%d_pose" % k] = r["pose7"]
            outs["out%d_id" % k] = r["lm_id"]
            outs["out%d_3d" % k] = r["lm_3d"]
            outs["out%d_outlier" % k] = r["outlier_id"]
    np.savez_compressed(os.path.join(HERE, "local_map.npz"), n_kf=np.int64(len(seq["kfs"])), K4=B.K4, **outs)
    print("written", os.listdir(HERE))


if __name__ == "__main__":
    main()
