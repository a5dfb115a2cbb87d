"""bench.py's legs that must never break the GPU line: the EuRoC leg's "dataset missing" record (BASELINE configs[0..1] run only
when an ASL folder is given) and the host-side diagnostics of the host-image leg (PCIe link, NUMA node of a buffer) -- none of them may raise
on a box without the dataset, without a GPU, or without the sysfs entries."""
import argparse
import sys
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def test_euroc_leg_reports_a_missing_dataset():
    for d in (None, "/nonexistent/MH_05_difficult/mav0"):
        out = {}
        bench.leg_euroc({"args": argparse.Namespace(euroc=d), "out": out})
        assert out["euroc"]["status"] == "dataset missing"
        assert ("run_sequence.py" in out["euroc"]["would_run"]) and (d is None or d in out["euroc"]["looked_in"])


def test_host_side_diagnostics_do_not_raise():
    a = np.zeros(1 << 16, np.uint8)
    a[:] = 1                                            # (touched: the pages exist)
    node = bench.numa_node_of(a.ctypes.data)
    assert node is None or (isinstance(node, int) and node >= 0)
    info = bench.pcie_link_info(0)                      # no GPU here: an "error" entry, never an exception
    assert isinstance(info, dict) and ("error" in info or "link" in info)


def test_parse_args_has_the_euroc_option_and_the_fixed_host_image_leg():
    args = bench.parse_args(["--euroc", "/data/MH_05/mav0"])
    assert args.euroc == "/data/MH_05/mav0"
    assert bench.H2D_LEG_FRAMES == 60 and bench.ROUND_TAG == "r06"
