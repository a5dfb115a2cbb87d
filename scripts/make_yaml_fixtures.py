#!/usr/bin/env python3
"""Pins the config loaders on the reference's own yaml dependency: runs oracle/_ref/yaml_dump (yaml-cpp 0.6.2 compiled
from /root/reference/3rdPartLib by oracle/Makefile, accessors as in src/utils/include/yamlRead.h) and commits what it
reads as small fixtures under tests/golden/:
  yaml_ref_<name>.txt    <- the reference's launch files of the supported sensor types (type_of_vi 0, 1, 2, 3, 4, 5)
  yaml_synth_<name>.txt  <- this repo's synthetic rig files (flvis_amd/synth.py)
Run in the build container only (needs /root/reference)."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flvis_amd import synth  # noqa: E402

REF_FILES = {
    "euroc": "/root/reference/launch/EuRoC_MAV/euroc.yaml",
    "d435i_stereo": "/root/reference/launch/d435i/sn943222072828_stereo.yaml",
    "d435_stereo_px4": "/root/reference/launch/d435_pixhawk/sn943222072828_stereo_px4.yaml",
    "d435i_depth": "/root/reference/launch/d435i/sn943222072828_depth.yaml",
    "d435_depth_px4": "/root/reference/launch/d435_pixhawk/sn841512070537_depth_px4.yaml",
    "kitti": "/root/reference/launch/KITTI/KITTI.yaml",
}
SYNTH = {"d435i_stereo": synth.D435I_STEREO_YAML, "euroc_like": synth.EUROC_LIKE_YAML, "d435i_depth": synth.D435I_DEPTH_YAML,
         "kitti_like": synth.KITTI_LIKE_YAML}

subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
dump = os.path.join(ROOT, "oracle", "_ref", "yaml_dump")
gold = os.path.join(ROOT, "tests", "golden")
for name, path in REF_FILES.items():
    out = subprocess.check_output([dump, path]).decode()
    open(os.path.join(gold, "yaml_ref_%s.txt" % name), "w").write(out)
    print("yaml_ref_%s.txt: %d keys" % (name, len(out.splitlines())))
for name, text in SYNTH.items():
    p = os.path.join(tempfile.gettempdir(), "flvis_fixture_%s.yaml" % name)
    open(p, "w").write(text)
    out = subprocess.check_output([dump, p]).decode()
    open(os.path.join(gold, "yaml_synth_%s.txt" % name), "w").write(out)
    print("yaml_synth_%s.txt: %d keys" % (name, len(out.splitlines())))
