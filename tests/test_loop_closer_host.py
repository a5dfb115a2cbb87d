"""CPU: the host side of the loop closer -- the LC_PARAS block of the yaml (pinned on what the reference's yaml-cpp reads from
launch/KITTI/KITTI.yaml) and the oracle-assembled control flow of the nodelet (tests/_loop_chain.py) on a rendered sequence that
returns to its start: a loop is found, verified and the pose graph pulls the drifted end back."""
import os
import tempfile

import numpy as np
import pytest

import _geom as G
import _loop_chain as LC
import _oracle as O
import _pgo_synth as PS
import _voc as V
import flvis_amd
from test_config_yamlcpp import GOLD, REF, read_dump
from test_oracle_bow import RefVoc

FIELDS = ("lcKFStart", "lcKFDist", "lcKFMaxDist", "lcKFLast", "lcNKFClosest", "minPts", "ratioMax", "ratioRansac", "minScore")


def test_lc_params_match_yaml_cpp_on_the_reference_kitti_file():
    d = read_dump(open(os.path.join(GOLD, "yaml_ref_kitti.txt")).read())
    assert all(k in d for k in FIELDS)
    assert {k: float(d[k][0]) for k in FIELDS} == {k: float(v) for k, v in LC.LC_PARAMS.items()}       # what the tests use
    if not os.path.exists(REF["kitti"]):
        pytest.skip("reference not present")
    prm = flvis_amd.load_lc_params(REF["kitti"])
    for k in FIELDS:
        assert float(getattr(prm, k)) == float(d[k][0]), k


def test_lc_params_from_a_written_file_and_refusals(tmp_path):
    p = str(tmp_path / "lc.yaml")
    body = "\n".join("#define %s\n%s: %s   # trailing comment" % (k, k, v) for k, v in LC.LC_PARAMS.items())
    open(p, "w").write("%YAML:1.0\nimage_width: 640\n" + body + "\n")
    prm = flvis_amd.load_lc_params(p)
    assert {k: getattr(prm, k) for k in FIELDS} == LC.LC_PARAMS
    open(p, "w").write(body.replace("minScore: 0.12", "minscore: 0.12"))
    with pytest.raises(flvis_amd.FlvisError) as e:
        flvis_amd.load_lc_params(p)
    assert "minScore" in str(e.value)
    with pytest.raises(flvis_amd.FlvisError):
        flvis_amd.load_lc_params(str(tmp_path / "absent.yaml"))
    # an EuRoC-style file has no loop-closing block (the nodelet is disabled there): refused with the missing key named
    from flvis_amd import synth
    open(p, "w").write(synth.EUROC_LIKE_YAML)
    with pytest.raises(flvis_amd.FlvisError) as e:
        flvis_amd.load_lc_params(p)
    assert "lcKFStart" in str(e.value)


def test_symbols_exported():
    lib = flvis_amd.load_library()
    for name in ("flvis_lc_params_load", "flvis_loop_closer_create", "flvis_loop_closer_destroy", "flvis_loop_closer_add_keyframes",
                 "flvis_loop_closer_process", "flvis_loop_closer_poses", "flvis_loop_closer_drift", "flvis_loop_closer_similarity_row"):
        assert hasattr(lib, name), name


def test_oracle_chain_closes_a_rendered_loop():
    from flvis_amd import synth
    p = os.path.join(tempfile.gettempdir(), "flvis_loopchain_cpu.yaml")
    open(p, "w").write(synth.D435I_STEREO_YAML)
    cfg = O.load_config(p)
    P0, P1 = np.array(list(cfg.P0)), np.array(list(cfg.P1))
    K4 = np.array([P0[0], P0[5], P0[2], P0[6]])
    tr = LC.LoopTrajectory()
    rnd = synth.Renderer("cpu")
    n_kf, per = 56, 50                                      # keyframes 50..55 see what keyframes 0..5 saw
    feats, gt = [], []
    for i, t in enumerate(LC.keyframe_times(n_kf, per)):
        a0, a1 = [x[0].numpy() for x in rnd.stereo_frame([tr], t, i)]
        k, d = O.orb_detect_and_compute(a0)
        lm2, lm3, lmd = O.lc_keyframe_landmarks(a0, a1, 0, k, d, P0, P1)
        R, tt = tr.T_c_w(t, rnd.rig)
        gt.append(G.pose7(R, tt))
        feats.append(dict(desc=d, lm2=lm2, lm3=lm3, lmd=lmd))
    rv = RefVoc(V.build_vocabulary([f["desc"] for f in feats[::6]], k=8, depth=3))
    for f in feats:
        f["bow"] = rv.transform(f["desc"])
    odom = LC.drifted_odometry(gt, 1, sigma_t=0.008, sigma_r=0.002)
    lc = LC.RefLoopCloser(K4)
    events = []
    for f, T in zip(feats, odom):
        lc.add(f, T)
        events.append(lc.process())
    assert not any(e["candidate"] for e in events[:49])                       # the `size < 50` gate
    closing = [e for e in events if e["accepted"] and e["kf_curr"] - e["kf_prev"] >= 45]
    assert len(closing) >= 3 and any(e["optimised"] for e in closing)
    for e in closing:                                                          # the verified pose is the true relative pose
        rel = PS.mul7(gt[e["kf_curr"]], PS.inv7(gt[e["kf_prev"]]))
        assert np.linalg.norm(e["pose"][:3] - rel[:3]) < 0.03 and e["n_inliers"] >= 100    # (EPnP on the inliers, unrefined: what SOLVEPNP_P3P ends with)
    gap0 = PS.loop_gap(np.array(odom), np.array(gt), 2, n_kf - 1)
    gap1 = PS.loop_gap(np.array(lc.T_c_w), np.array(gt), 2, n_kf - 1)
    assert gap0[0] > 0.05 and gap1[0] < 0.3 * gap0[0], (gap0, gap1)
    assert np.linalg.norm(lc.T_odom_map[:3]) > 0.01                           # the map -> odom correction is not the identity any more
