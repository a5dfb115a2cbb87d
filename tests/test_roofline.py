"""The per-kernel roofline table of bench.py (flvis_amd/roofline.py, pure functions): byte counts, the VALU-issue fraction, the flop
formula of the local map and the table's shape, checked on CPU against hand-computed values."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flvis_amd import roofline as rf  # noqa: E402


def test_byte_counts_follow_survey_8d():
    assert rf.pyramid_total_bytes(640, 480) == 307200 + 76800 + 19200 + 4800 == 408000          # one pyramid
    assert rf.pyramid_bytes(640, 480) == (307200 + 76800) + (76800 + 19200) + (19200 + 4800)     # reads + writes of levels 1..3
    assert rf.pyramid_total_bytes(752, 480) == 360960 + 90240 + 22560 + 5640


def test_issue_peak_and_ba_flops():
    assert abs(rf.VALU_ISSUE_PEAK_GINST - 614.4) < 1e-9                                          # 1024 SIMDs x 2.4 GHz / 4
    # SURVEY 8d's worked example: E = 2000, L = 300, P = 9 -> 2000 * 420 + 300 * 22 * 324 + 52488 ~ 3.0 MFLOP per trial
    fl = rf.ba_flops(1, 2000, 300, 9)
    want = 2000 * 420 + (2000 * 2000 / 300) / 2 * 324 + (6 * 9) ** 3 / 3
    assert abs(fl - want) < 1e-6 and 2.9e6 < fl < 3.2e6
    assert rf.ba_flops(20, 20 * 2000, 20 * 300, 20 * 9) == 20 * fl and rf.ba_flops(0, 0, 0, 0) == 0.0


def test_kernel_table_prices_every_kernel_against_its_bound():
    st = {"imu_feed+frame_begin": 0.04, "pyr_down(left)": 0.067, "lk_track(temporal)": 0.4, "ransac_f": 0.06, "ransac_pnp": 0.1,
          "track_post+pose_lm": 0.12, "reproj_filter": 0.03, "gftt:eig_cand": 0.2, "gftt:pick": 0.25, "feature_dem+add_new": 0.05,
          "depth_prepare": 0.07, "lk_track(stereo)": 0.3, "depth_innovate": 0.1, "frame_end": 0.02, "ba_worker(launch)": 2.0}
    pmc = {"kernels": {"k_lk_track": {"valu_insts": 122.88e6, "fetch_kb": 24000.0, "write_kb": 1000.0, "avg_ns": 310000.0},
                       "k_eig_walk": {"valu_insts": 18e6},
                       "k_pyr_walk<1,true>": {"valu_insts": 1e6, "fetch_kb_calibrated": 19200.0, "write_kb_calibrated": 28000.0, "avg_ns": 19000.0},
                       "k_pyr_walk<2,false>": {"valu_insts": 0.5e6, "fetch_kb_calibrated": 4800.0, "write_kb_calibrated": 2000.0, "avg_ns": 11000.0}}}
    ba = {"runs": 10, "trials": 200, "trials_items": 200 * 1000, "trials_landmarks": 200 * 400, "trials_poses": 200 * 7,
          "ms_per_optimisation": 1.5, "worker_ms_per_launch": 2.0}
    rows = rf.kernel_table(st, 64, 640, 480, 5000.0, pmc, ba, 1.5)
    by = {r["kernel"]: r for r in rows}
    lk = by["k_lk_track (temporal)"]
    assert lk["algorithmic_bytes_per_launch"] == 2 * 408000 * 64
    assert abs(lk["valu_ginst_per_s"] - 307.2) < 0.1 and abs(lk["valu_issue_frac"] - 0.5) < 1e-3     # 122.88 M in 0.4 ms
    assert lk["traffic_bytes_per_launch"] == 25000 * 1024 and lk["rocprof_avg_launch_ms"] == 0.31
    e = by["k_eig_walk"]
    assert e["algorithmic_bytes_per_launch"] == 307200 * 64 and abs(e["achieved_GBs"] - 98.3) < 0.1
    assert abs(e["frac_of_measured_copy"] - 98.304 / 5000) < 1e-3 and abs(e["valu_issue_frac"] - (18e6 / 0.2e-3 / 1e9) / 614.4) < 1e-3
    p = by["k_pyr_walk<1,true> + k_pyr_walk<2,false> (left pyramid)"]
    assert p["algorithmic_bytes_per_launch"] == (307200 + rf.pyramid_bytes(640, 480)) * 64
    assert p["traffic_bytes_per_launch"] == 54000 * 1024 and p["traffic_calibrated"] and p["rocprof_launches_ms"] == [0.019, 0.011]
    assert abs(p["achieved_GBs_kernels_only"] - p["algorithmic_bytes_per_launch"] / 30e-6 / 1e9) < 0.1
    b = by["k_ba_worker"]
    per_opt = rf.ba_flops(200, 200000, 80000, 1400) / 10
    assert abs(b["mflop_per_optimisation"] - per_opt / 1e6) < 0.01 and b["lm_trials_per_optimisation"] == 20
    assert abs(b["gflops_per_workgroup"] - per_opt / 1.5e-3 / 1e9) < 0.01
    assert abs(b["frac_of_cu_fp64_peak"] - b["gflops_per_workgroup"] / (78600 / 256)) < 1e-3 and "share_of_step" not in b
    assert all("avg_launch_ms" in r for r in rows) and abs(lk["share_of_step"] - 0.4 / 1.5) < 1e-3
    # without counters and without the local map the table still has its time lines
    rows2 = rf.kernel_table(st, 64, 640, 480, None, None, None, None)
    assert "k_ba_worker" not in {r["kernel"] for r in rows2} and "valu_issue_frac" not in rows2[0]
