"""ros/src/*.cpp -- the nodelet wrappers a maintainer drops into the reference's catkin workspace -- cannot be built here (no ROS in the
image).  They are type-checked instead: `g++ -fsyntax-only` against the real include/flvis_hip.h and a declarations-only shape of the
ROS API subset they use (tests/cpp/ros_api_shape/, see its README: not a ROS implementation, not a claim that catkin builds them).
What this catches: every call into the C ABI with the wrong argument count / order / types, and plain C++ errors."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPE = os.path.join(ROOT, "tests", "cpp", "ros_api_shape")
WRAPPERS = ["tracking_nodelet.cpp", "localmap_nodelet.cpp", "loopclosing_nodelet.cpp"]


def _check(path):
    return subprocess.run(["g++", "-std=c++14", "-fsyntax-only", "-Wall", "-Werror=return-type", "-I" + SHAPE, "-I" + os.path.join(ROOT, "include"), path],
                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
@pytest.mark.parametrize("name", WRAPPERS)
def test_wrapper_type_checks_against_the_c_abi(name):
    r = _check(os.path.join(ROOT, "ros", "src", name))
    assert r.returncode == 0, r.stdout


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_the_type_check_is_not_vacuous(tmp_path):
    """The same harness refuses a wrapper that calls the C ABI wrongly (an argument dropped from flvis_ba_push_keyframe, the
    hold_buffers flag dropped from flvis_image_feed_host): the check above means something."""
    for name, good, bad in (("localmap_nodelet.cpp", "4096, &fid, To,", "4096, To,"),
                            ("tracking_nodelet.cpp", "/*with_local_map=*/0, /*hold_buffers=*/0)", "/*with_local_map=*/0)")):
        src = open(os.path.join(ROOT, "ros", "src", name)).read()
        assert good in src
        p = tmp_path / name
        p.write_text(src.replace(good, bad))
        r = _check(str(p))
        assert r.returncode != 0, "the harness accepted a wrong call in " + name


def test_nodelet_xml_names_the_wrapper_classes():
    xml = open(os.path.join(ROOT, "ros", "flvis_hip_nodelets.xml")).read()
    for cls in ("flvis_hip::TrackingNodelet", "flvis_hip::LocalMapNodelet", "flvis_hip::LoopClosingNodelet"):
        assert cls in xml, cls
