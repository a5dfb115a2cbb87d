"""Pins the yaml loaders (product: flvis_config_load, csrc/config.cpp; oracle: ref_config_load_yaml) on the reference's OWN
yaml dependency: tests/golden/yaml_*.txt hold what yaml-cpp 0.6.2 (compiled from /root/reference/3rdPartLib by
oracle/Makefile, read through the accessors of src/utils/include/yamlRead.h) returns for every key -- see
scripts/make_yaml_fixtures.py.  SURVEY.md §8b: "must accept the reference's yaml files byte-for-byte"."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

import _oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def read_dump(text):
    out = {}
    for line in text.splitlines():
        f = line.split()
        if len(f) >= 3 and f[1] != "s":
            out[f[0]] = np.array([float(x) for x in f[2:2 + int(f[1])]])
    return out


def inv44(m):
    return np.linalg.inv(np.asarray(m).reshape(4, 4))


def check_cfg_against_dump(cfg, d, who):
    """raw (pre-finalize) fields of flvis_cfg against the yaml-cpp values; exact unless a product of 4x4 matrices is involved"""
    assert cfg.type_of_vi == int(d["type_of_vi"][0]), who
    assert cfg.image_width == int(d["image_width"][0]) and cfg.image_height == int(d["image_height"][0]), who
    depth_mode = cfg.type_of_vi in (0, 2)   # vo_tracking.cpp:149-154 reads cam0, depth_factor and T_imu_cam0 only
    kitti = cfg.type_of_vi == 4             # vo_tracking.cpp:265-306 reads the two projection matrices only
    keys = [("cam0_intrinsics", "cam0_intrinsics"), ("cam0_distortion", "cam0_distortion_coeffs")]
    if kitti:
        keys = []
    elif not depth_mode:
        keys += [("cam1_intrinsics", "cam1_intrinsics"), ("cam1_distortion", "cam1_distortion_coeffs")]
    for k, name in keys:
        assert np.array_equal(np.array(list(getattr(cfg, k))), d[name]), (who, k)
    for k, n in (("vifusion_para", 6), ("feature_para", 6), ("dr_para", 3)):
        want = np.array([d["%s%d" % (k, i + 1)][0] for i in range(n)])
        assert np.array_equal(np.array(list(getattr(cfg, k))), want), (who, k)
    assert cfg.window_size == int(d["window_size"][0]), who
    T_i_c0 = np.array(list(cfg.T_imu_cam0)).reshape(4, 4)
    T_c0_c1 = np.array(list(cfg.T_cam0_cam1)).reshape(4, 4)
    if kitti:
        P0, P1 = d["cam0_projection_matrix"].reshape(4, 4), d["cam1_projection_matrix"].reshape(4, 4)
        assert np.array_equal(np.array(list(cfg.P0)), P0[:3].reshape(-1)) and np.array_equal(np.array(list(cfg.P1)), P1[:3].reshape(-1)), who
        K = np.array([P0[0, 0], P0[1, 1], P0[0, 2], P0[1, 2]])       # K0 = K1 = K0_rect = P0(0:3, 0:3), D = 0 (:283-285)
        for k in ("cam0_intrinsics", "cam1_intrinsics"):
            assert np.array_equal(np.array(list(getattr(cfg, k))), K), (who, k)
        for k in ("cam0_distortion", "cam1_distortion"):
            assert not np.any(np.array(list(getattr(cfg, k)))), (who, k)
        want = np.eye(4)
        want[:3, 3] = np.linalg.inv(P0[:3, :3]) @ P1[:3, 3]          # mat_T_c0_c1 = K^-1 * P1 with an identity rotation (:272-276)
        assert np.allclose(T_c0_c1, want, atol=1e-15, rtol=1e-15), (who, T_c0_c1 - want)
        assert np.array_equal(T_i_c0, np.eye(4)), who                # the dummy SE3() of :293
        assert (cfg.cam_type, cfg.skip_first_n_imgs, cfg.need_equal_hist) == (0, 0, 0), who   # STEREO_RECT, init(..., 0, false)
        assert np.array_equal(np.array(list(cfg.R0)), np.eye(3).reshape(-1)) and np.array_equal(np.array(list(cfg.R1)), np.eye(3).reshape(-1))
    elif depth_mode:
        assert cfg.depth_factor == d["depth_factor"][0] and cfg.cam_type == 2, who
        assert np.array_equal(T_i_c0.reshape(-1), d["T_imu_cam0"]), who
    elif cfg.type_of_vi == 1:   # vo_tracking.cpp:228-236: T_i_c0 = T_imu_mavimu * T_mavimu_cam0, T_c0_c1 = T_mavimu_cam0^-1 * T_mavimu_cam1
        a, b, m = d["T_mavimu_cam0"].reshape(4, 4), d["T_mavimu_cam1"].reshape(4, 4), d["T_imu_mavimu"].reshape(4, 4)
        assert np.allclose(T_i_c0, m @ a, atol=1e-12, rtol=0), who
        assert np.allclose(T_c0_c1, inv44(a) @ b, atol=1e-12, rtol=0), who
    else:
        assert np.array_equal(T_i_c0.reshape(-1), d["T_imu_cam0"]), who
        assert np.array_equal(T_c0_c1.reshape(-1), d["T_cam0_cam1"]), who


@pytest.mark.parametrize("name", ["d435i_stereo", "euroc_like", "d435i_depth", "kitti_like"])
def test_loaders_match_yaml_cpp_on_the_synthetic_rig_files(name):
    import flvis_amd
    from flvis_amd import synth
    text = {"d435i_stereo": synth.D435I_STEREO_YAML, "euroc_like": synth.EUROC_LIKE_YAML, "d435i_depth": synth.D435I_DEPTH_YAML,
            "kitti_like": synth.KITTI_LIKE_YAML}[name]
    p = os.path.join(tempfile.gettempdir(), "flvis_yamlcpp_%s.yaml" % name)
    open(p, "w").write(text)
    d = read_dump(open(os.path.join(GOLD, "yaml_synth_%s.txt" % name)).read())
    check_cfg_against_dump(flvis_amd.load_config(p), d, "product")
    check_cfg_against_dump(O.load_config(p), d, "oracle")


REF = {"euroc": "/root/reference/launch/EuRoC_MAV/euroc.yaml",
       "d435i_stereo": "/root/reference/launch/d435i/sn943222072828_stereo.yaml",
       "d435_stereo_px4": "/root/reference/launch/d435_pixhawk/sn943222072828_stereo_px4.yaml",
       "d435i_depth": "/root/reference/launch/d435i/sn943222072828_depth.yaml",
       "d435_depth_px4": "/root/reference/launch/d435_pixhawk/sn841512070537_depth_px4.yaml",
       "kitti": "/root/reference/launch/KITTI/KITTI.yaml"}


@pytest.mark.parametrize("name", sorted(REF))
def test_loaders_match_yaml_cpp_on_the_reference_launch_files(name):
    """the reference's own launch files, read in place (they are not copied into this repo): skipped where /root/reference
    does not exist (the GPU box)."""
    if not os.path.exists(REF[name]):
        pytest.skip("reference not present")
    import flvis_amd
    gold = open(os.path.join(GOLD, "yaml_ref_%s.txt" % name)).read()
    dump = os.path.join(ROOT, "oracle", "_ref", "yaml_dump")
    if os.path.exists(dump):   # the committed fixture is what the reference's yaml-cpp says today
        assert subprocess.check_output([dump, REF[name]]).decode() == gold
    d = read_dump(gold)
    check_cfg_against_dump(flvis_amd.load_config(REF[name]), d, "product")
    check_cfg_against_dump(O.load_config(REF[name]), d, "oracle")
