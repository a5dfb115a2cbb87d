#!/usr/bin/env python3
"""bench.py -- frames/sec of the MI355X-native FLVIS hot path (front-end tracking + sliding-window BA).

Workload (BASELINE.json configs[3]): one GPU tracks a batch of 64 independent synthetic 640x480 stereo + 200 Hz IMU
streams; a "step" is one stereo frame for every stream of the batch: full HIP front-end (ingest, 2x3-level pyramids,
temporal LK, F-RANSAC, PnP-RANSAC, pose LM, reprojection filter, GFTT + FeatureDEM redetect, stereo LK, DLT depth
innovation, keyframe decision) plus the batched Schur BA for every stream that emits a keyframe.  Inputs (rendered
frames, IMU samples) are resident in HBM / host memory before the timed region starts.

Frame schedule (flvis_amd/bench_plan.py, tested on CPU): an UNTIMED pre-roll of skip_first_n_imgs + 12 frames brings every
stream into the Tracking state (checked, more frames are fed while some stream is not there yet, the run fails loudly if
that never happens), then --warmup untimed steps, then EXACTLY --steps timed steady-state steps, then 20 untimed steps with
every stage bracketed by HIP events.

--gpus N: one rank per GPU.  Under torchrun (WORLD_SIZE set) the rank joins the job; without it `bench.py --gpus N` spawns
the N ranks itself (torch.distributed.run on 127.0.0.1) and fails loudly when fewer than N GPUs are visible.  Every rank
tracks its own streams (weak: 64 per GPU; --scaling strong: 512 in total), no data-path collective; the only exchange is one
RCCL all-gather of the final poses + all-reduce of counters after the timed region.

--pmc: runs itself twice under rocprofv3 (--pmc FETCH_SIZE, --pmc WRITE_SIZE: separate passes) and writes
profiles/<tag>_lk_pmc.json, the source of roofline.traffic (ignored when the kernel source changed since).

Prints ONE JSON line (rank 0).  `latency_ms.timed_region_ms` says where the wall time of the timed region went (frame chains, idle
time between frames, the local map's tail after the last frame).  FLVIS_BENCH_FRAMES=1 (diagnosis) adds per-frame chain / stage /
host-call times to the line and puts events around every stage in the timed region.
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from flvis_amd import bench_plan as plan  # noqa: E402  (pure python, no GPU)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md: 8 TB/s peak, ~6.3 TB/s achievable)
ROUND_TAG = "r06"
H2D_LEG_FRAMES = 60

# algorithmic HBM bytes per launch of the image-scan kernels for ONE stream, 640x480:
PYR_BYTES = 307200 + 76800 + 19200 + 4800          # one pyramid (levels 0..3)
LK_ALG_BYTES = 2 * PYR_BYTES                       # one pass over two pyramids per k_lk_track launch
METRIC = "frames/sec/node (640×480 stereo+IMU) + ATE vs CPU ref, EuRoC MH_05"  # BASELINE.json's metric, verbatim


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--streams", type=int, default=64, help="independent streams per GPU (weak scaling)")
    ap.add_argument("--input-hold", type=int, default=4, help="multi-lane trackers (FLVIS_LANES > 1): frames an input buffer stays "
                    "untouched after it was handed over (flvis_set_input_hold); every frame of the run has its own buffer here")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--total-streams", type=int, default=512, help="fixed total for --scaling strong")
    ap.add_argument("--cpu-frames", type=int, default=80, help="frames of the bounded single-thread CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-mt-frames", type=int, default=40, help="frames per stream of the one-thread-per-core CPU leg (0 = skip)")
    ap.add_argument("--no-local-map", action="store_true")
    ap.add_argument("--no-epilogue", action="store_true", help="skip the untimed per-stage epilogue")
    ap.add_argument("--no-h2d", action="store_true", help="skip the host-image (PCIe-inclusive) variant")
    ap.add_argument("--euroc", default=os.environ.get("FLVIS_EUROC_DIR"), help="EuRoC ASL sequence folder (e.g. MH_05_difficult/mav0): BASELINE "
                    "configs[0..1] -- the sequence through the HIP path and through the CPU restatement, ATE of each against the ground "
                    "truth and of one against the other; absent: the line says \"dataset missing\" (SURVEY 8d)")
    ap.add_argument("--pmc", action="store_true", help="collect FETCH_SIZE / WRITE_SIZE of k_lk_track under rocprofv3")
    ap.add_argument("--stub", action="store_true",
                    help="CPU test hook: no GPU work, gloo instead of RCCL; exercises rank spawning + result exchange only")
    return ap.parse_args(argv)


def lk_source_hash():
    h = hashlib.sha256()
    for f in ("lk_kernel.hip", "img_kernels.hpp", "dev_common.hpp"):
        h.update(open(os.path.join(ROOT, "flvis_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def maybe_spawn(args, argv):
    """`bench.py --gpus N` outside torchrun: start the N ranks (one per GPU) and replace this process."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    if not args.stub:
        import torch
        n = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if n < args.gpus:
            raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible on this node (no CPU fallback exists)" % (args.gpus, n))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + argv
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def bind_to_gpu_numa_node(local_rank):
    """Pins this process to the CPUs local to its GPU (sysfs `local_cpulist` of the GPU's PCI function), so that the ranks of one node
    do not migrate across sockets or pile up on the same cores.  Returns a description for the bench line; never fails the run."""
    try:
        import torch
        pr = torch.cuda.get_device_properties(local_rank)
        bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        base = "/sys/bus/pci/devices/" + bdf
        node = open(base + "/numa_node").read().strip()
        cpus = set()
        for part in open(base + "/local_cpulist").read().strip().split(","):
            if not part:
                continue
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        cpus &= set(os.sched_getaffinity(0))
        if not cpus:
            return {"pci": bdf, "numa_node": node, "bound": False}
        os.sched_setaffinity(0, cpus)
        return {"pci": bdf, "numa_node": node, "bound": True, "cpus": len(cpus)}
    except Exception as e:  # noqa: BLE001
        return {"bound": False, "why": "%s: %s" % (type(e).__name__, e)}


def pcie_link_info(local_rank):
    """The GPU's PCIe link as sysfs reports it (negotiated and maximal width / speed of the GPU's own function and of the port above it)
    and its NUMA node: what tells a slow link or a remote socket from a regression when with_h2d differs between two boxes."""
    info = {}
    try:
        import torch
        pr = torch.cuda.get_device_properties(local_rank)
        bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        info["pci"] = bdf
        base = os.path.realpath("/sys/bus/pci/devices/" + bdf)

        def rd(d, name):
            try:
                return open(os.path.join(d, name)).read().strip()
            except OSError:
                return None
        hops = []
        d = base
        while len(hops) < 5 and os.path.exists(os.path.join(d, "current_link_speed")):  # the function, then the bridges above it up to the root port
            hops.append({"dev": os.path.basename(d), "speed": rd(d, "current_link_speed"), "width": rd(d, "current_link_width"),
                         "max_speed": rd(d, "max_link_speed"), "max_width": rd(d, "max_link_width")})
            d = os.path.dirname(d)
        info["link"] = hops
        info["gpu_numa_node"] = rd(base, "numa_node")
    except Exception as e:  # noqa: BLE001
        info["error"] = "%s: %s" % (type(e).__name__, e)
    return info


def numa_node_of(addr):
    """NUMA node of the page that holds host address addr (move_pages with a NULL node list only queries), or None."""
    try:
        libc = C.CDLL(None, use_errno=True)
        pages = (C.c_void_p * 1)(C.c_void_p(addr & ~4095))
        status = (C.c_int * 1)(-1)
        rc = libc.syscall(C.c_long(279), C.c_int(0), C.c_ulong(1), pages, C.c_void_p(0), status, C.c_int(0))  # __NR_move_pages (x86-64)
        return int(status[0]) if rc == 0 and status[0] >= 0 else None
    except Exception:  # noqa: BLE001
        return None


# ------------------------------------------------------------------------------------------------ CPU baseline (oracle)
def load_checker_lib():
    """the op-by-op IEEE build of the CPU restatement (what the parity tests compare the HIP path with, bit for bit)"""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return C.CDLL(os.path.join(ROOT, "oracle", "libflvis_ref.so"))


def load_oracle_lib():
    """The CPU restatement's timing build (-O3 -march=native, built on THIS host by `make -C oracle fast`); falls back to the
    op-by-op IEEE checker build the parity tests use.  Returns (ctypes lib, description of the build)."""
    try:
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "fast"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return C.CDLL(os.path.join(ROOT, "oracle", "_fast", "libflvis_ref_fast.so")), "-O3 -march=native (timing build, oracle/_fast)"
    except Exception:
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return C.CDLL(os.path.join(ROOT, "oracle", "libflvis_ref.so")), "-O2 -ffp-contract=off (checker build; the timing build failed)"


def cpu_run_streams(olib, cfg, T, threads, first, n, host_frames, imu, imu_cnt, seed_base, frame_hz, with_local_map):
    """oracle/ref_runner.cpp: T streams on `threads` native host threads (no Python in the timed part).  host_frames: list of n
    (img0 [>=T,H,W], img1 [>=T,H,W]) uint8 arrays.  Returns (seconds, poses [T,n,7], states [T,n], frame_ms [T,n])."""
    f0 = [np.ascontiguousarray(h[0][:T]) for h in host_frames[:n]]
    f1 = [np.ascontiguousarray(h[1][:T]) for h in host_frames[:n]]
    p0 = (C.c_void_p * n)(*[a.ctypes.data for a in f0])
    p1 = (C.c_void_p * n)(*[a.ctypes.data for a in f1])
    im = np.ascontiguousarray(imu[:first + n, :T])
    ic = np.ascontiguousarray(imu_cnt[:first + n, :T]).astype(np.int32)
    seeds = np.array([seed_base + s for s in range(T)], np.uint64)
    poses = np.zeros((T, n, 7))
    states = np.zeros((T, n), np.int32)
    fms = np.zeros((T, n))
    olib.ref_run_streams.restype = C.c_double
    olib.ref_run_streams.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    secs = olib.ref_run_streams(C.byref(cfg), T, n, first, frame_hz, p0, p1, im.ctypes.data, ic.ctypes.data, im.shape[2],
                                seeds.ctypes.data, int(with_local_map), threads, poses.ctypes.data, states.ctypes.data, fms.ctypes.data)
    if secs <= 0:
        raise RuntimeError("ref_run_streams rejected its arguments")
    return secs, poses, states, fms


def centre(p7):
    q = p7[3:7] / np.linalg.norm(p7[3:7])
    x, y, z, w = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    return -R.T @ p7[0:3]


# --------------------------------------------------------------------------------------------------------- stub (CPU)
def run_stub(args, rank, world):
    """No GPU work: the rank-spawning, sharding, barrier / max-over-ranks timing and result exchange of the real run on gloo."""
    import torch
    import torch.distributed as dist
    from flvis_amd import dist as fdist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")
    S = plan.streams_per_gpu(args.scaling, world, args.streams, args.total_streams)
    sched = plan.frame_schedule(args.steps, args.warmup, 50)
    ids = fdist.shard_streams(rank, world, S)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    if world > 1:
        dist.barrier()
    elapsed = fdist.max_over_ranks(time.perf_counter() - t0)
    poses = torch.tensor([[float(s), 0, 0, 0, 0, 0, 1.0] for s in ids], dtype=torch.float64)
    K = sched["timed"][1] - sched["timed"][0]
    all_poses, csum = fdist.exchange_results(poses, [S * K, 0, 0, S])
    assert all_poses.shape[0] == world * S and [int(x) for x in all_poses[:, 0].tolist()] == list(range(world * S))
    if rank == 0:
        print(json.dumps({"metric": METRIC, "value": None, "unit": "frames/s", "stub": True, "n_gpus": world, "steps": K,
                          "warmup": args.warmup, "ms_per_step": round(elapsed / K * 1e3, 4), "higher_is_better": True,
                          "scaling": args.scaling, "vs_baseline": None, "data": "none (stub)",
                          "config": {"workload": "stub", "streams_per_gpu": S, "streams_total": world * S,
                                     "frames_exchanged": csum[0], "streams_tracking_at_end": csum[3]}}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


# --------------------------------------------------------------------------------------------------------------- PMC
def csrc_hash():
    """hash of every kernel source: the per-kernel counters in profiles/ belong to exactly this code"""
    import glob
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "flvis_amd", "csrc", "*"))):
        if os.path.isfile(f):
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def short_kernel_name(full):
    n = full.split("(")[0]
    n = n.split("::")[-1].strip()
    if n.startswith("k_pyr_walk<"):   # (instances kept apart: <levels produced, with the ingest copy>)
        return n.replace(" ", "")
    return n.split("<")[0].strip()   # (the tracker's two LK launches are kernels of their own: k_lk_track_temporal / k_lk_track_stereo)


def run_pmc(args):
    """Four rocprofv3 passes of a short run of this script: a plain kernel trace (durations), then FETCH_SIZE, WRITE_SIZE (they do not
    fit one pass on gfx950) and SQ_INSTS_VALU, each in its own --pmc pass.  Per kernel the averages over the launches of the second half
    of the run (the timed region; the first half is pre-roll and warm-up) -> profiles/<tag>_kernel_pmc.json, and k_lk_track's traffic
    additionally as profiles/<tag>_lk_pmc.json (the file `roofline.traffic` is read from)."""
    import csv
    import glob
    base = [sys.executable, os.path.abspath(__file__), "--steps", str(args.steps), "--warmup", str(args.warmup),
            "--streams", str(args.streams), "--cpu-frames", "0", "--cpu-mt-frames", "0", "--no-epilogue", "--no-h2d"]
    env = dict(os.environ, TMPDIR="/tmp")
    # (counter collection serialises kernel execution: a k_wait_flag that sleeps until a kernel of another stream has stored its word would
    # wait for a kernel that cannot start -- each such join would end by its 4 s limit.  The counter passes therefore run with the joins of
    # rounds 1-5, hipEvents, which the command processor resolves between kernels; the per-kernel instruction and byte counts do not depend
    # on how the streams are joined, and the tracker's kernels are the same)
    env_pmc = dict(env, FLVIS_JOIN="event")
    kern = {}

    def second_half(rows):
        return rows[len(rows) // 2:] if len(rows) >= 2 else rows

    d = tempfile.mkdtemp(prefix="flvis_trace_", dir="/tmp")
    r = subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--"] + base, cwd="/tmp", env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if r.returncode != 0 or not files:
        raise SystemExit("rocprofv3 --kernel-trace failed (rc %d):\n%s" % (r.returncode, r.stdout.decode(errors="replace")[-2000:]))
    per = {}
    for x in csv.DictReader(open(files[0])):
        if "flvis::" in x["Kernel_Name"]:
            per.setdefault(short_kernel_name(x["Kernel_Name"]), []).append(float(x["End_Timestamp"]) - float(x["Start_Timestamp"]))
    for k, v in per.items():
        t = second_half(v)
        kern[k] = {"launches_total": len(v), "launches_averaged": len(t), "avg_ns": sum(t) / len(t)}
    for ctr, key in (("FETCH_SIZE", "fetch_kb"), ("WRITE_SIZE", "write_kb"), ("SQ_INSTS_VALU", "valu_insts")):
        d = tempfile.mkdtemp(prefix="flvis_pmc_%s_" % ctr, dir="/tmp")
        r = subprocess.run(["rocprofv3", "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--"] + base,
                           cwd="/tmp", env=env_pmc, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if r.returncode != 0 or not files:
            raise SystemExit("rocprofv3 --pmc %s failed (rc %d):\n%s" % (ctr, r.returncode, r.stdout.decode(errors="replace")[-2000:]))
        per = {}
        for x in csv.DictReader(open(files[0])):
            if "flvis::" in x["Kernel_Name"] and x["Counter_Name"] == ctr:
                per.setdefault(short_kernel_name(x["Kernel_Name"]), []).append(float(x["Counter_Value"]))
        for k, v in per.items():
            t = second_half(v)
            kern.setdefault(k, {})[key] = sum(t) / len(t)
    # ... and the other instruction classes in one pass of SQ counters (round 6: a SIMD issues ONE instruction of any class per 4 cycles --
    # profiles/r06_chain_ab.md --, so a kernel's issue fraction is priced on all of them, not on the vector ones alone).  A counter the
    # profiler does not know is dropped and the pass repeated without it; a pass that fails leaves the fields out.
    others = [("SQ_INSTS_SALU", "salu_insts"), ("SQ_INSTS_LDS", "lds_insts"), ("SQ_INSTS_SMEM", "smem_insts"), ("SQ_INSTS_VMEM_RD", "vmem_rd_insts"),
              ("SQ_INSTS_VMEM_WR", "vmem_wr_insts")]
    for attempt in range(len(others)):
        d = tempfile.mkdtemp(prefix="flvis_pmc_insts_", dir="/tmp")
        r = subprocess.run(["rocprofv3", "--pmc"] + [c for c, _ in others] + ["--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--"] + base,
                           cwd="/tmp", env=env_pmc, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if r.returncode == 0 and files:
            for ctr, key in others:
                per = {}
                for x in csv.DictReader(open(files[0])):
                    if "flvis::" in x["Kernel_Name"] and x["Counter_Name"] == ctr:
                        per.setdefault(short_kernel_name(x["Kernel_Name"]), []).append(float(x["Counter_Value"]))
                for k, v in per.items():
                    t = second_half(v)
                    kern.setdefault(k, {})[key] = sum(t) / len(t)
            break
        txt = r.stdout.decode(errors="replace")
        bad = [c for c, _ in others if c in txt[-4000:]]
        sys.stderr.write("[pmc] instruction-class pass failed (rc %d)%s\n" % (r.returncode, ": dropping %s" % bad[0] if bad else ""))
        if not bad or len(others) <= 2:
            break
        others = [o for o in others if o[0] != bad[0]]
    cal = read_calibration()
    if cal:   # known-byte probes on this hardware (scripts/pmc_calibrate.py): counter reading x factor = bytes
        for k in kern.values():
            if k.get("fetch_kb") is not None:
                k["fetch_kb_calibrated"] = k["fetch_kb"] * cal["fetch_factor"]
            if k.get("write_kb") is not None:
                k["write_kb_calibrated"] = k["write_kb"] * cal["write_factor"]
    out = {"source_sha": csrc_hash(), "steps": args.steps, "warmup": args.warmup, "streams": args.streams, "kernels": kern,
           "calibration": cal,
           "note": "rocprofv3, four passes of `bench.py --steps %d --warmup %d --no-epilogue --no-h2d`: kernel trace (avg_ns), --pmc "
                   "FETCH_SIZE, --pmc WRITE_SIZE (KB as the counters report them; *_calibrated = x the factors of the known-byte probes in "
                   "`calibration`: on gfx950 FETCH_SIZE reports half of the bytes read at every access width probed, WRITE_SIZE the bytes "
                   "written), --pmc SQ_INSTS_VALU (wave instructions); per kernel the average over the second half of its launches (the "
                   "timed region)" % (args.steps, args.warmup)}
    # the headline roofline object prices "k_lk_track" as one kernel with two launches per step: the mean of its two instances
    inst = [kern[k] for k in ("k_lk_track_temporal", "k_lk_track_stereo") if k in kern]
    if inst and "k_lk_track" not in kern:
        kern["k_lk_track"] = {f: sum(i[f] for i in inst) / len(inst) for f in inst[0] if all(f in i for i in inst)}
    lk = kern.get("k_lk_track", {})
    lkout = {"kernel": "k_lk_track", "lk_source_sha": lk_source_hash(), "steps": args.steps, "streams": args.streams,
             "fetch_size_kb_per_launch": lk.get("fetch_kb"), "write_size_kb_per_launch": lk.get("write_kb"),
             "traffic_bytes_per_launch": (lk.get("fetch_kb_calibrated", lk.get("fetch_kb", 0)) + lk.get("write_kb_calibrated", lk.get("write_kb", 0))) * 1024.0 if lk else None,
             "calibrated": bool(cal), "note": "from %s_kernel_pmc.json" % ROUND_TAG}
    for dst in (os.path.join(ROOT, "profiles"), os.path.join(ROOT, "gpurun_out")):
        os.makedirs(dst, exist_ok=True)
        json.dump(out, open(os.path.join(dst, "%s_kernel_pmc.json" % ROUND_TAG), "w"), indent=1)
        json.dump(lkout, open(os.path.join(dst, "%s_lk_pmc.json" % ROUND_TAG), "w"), indent=1)
    print(json.dumps(out))


def read_calibration():
    """factors of the newest committed counter calibration (profiles/rNN_counter_calibration.json): mean over the probes"""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_counter_calibration.json")), reverse=True):
        try:
            c = json.load(open(path))
            ff = [p["fetch_factor"] for p in c["probes"].values() if p.get("fetch_factor")]
            wf = [p["write_factor"] for p in c["probes"].values() if p.get("write_factor")]
            if ff and wf:
                return {"file": os.path.basename(path), "fetch_factor": round(sum(ff) / len(ff), 4), "write_factor": round(sum(wf) / len(wf), 4),
                        "fetch_factor_range": [round(min(ff), 4), round(max(ff), 4)], "write_factor_range": [round(min(wf), 4), round(max(wf), 4)]}
        except Exception:
            continue
    return None


def read_kernel_pmc(S):
    """per-kernel counters from the newest committed PMC file -- only if it was measured on THIS source tree and batch size"""
    import glob
    want = csrc_hash()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_kernel_pmc.json")), reverse=True):
        try:
            pmc = json.load(open(path))
        except Exception:
            continue
        if pmc.get("source_sha") == want and pmc.get("streams") == S:
            return pmc, os.path.basename(path)
    return None, None


def read_traffic(S):
    """roofline.traffic from the newest committed PMC file -- only if it was measured on THIS kernel source and batch size."""
    import glob
    want = lk_source_hash()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_lk_pmc.json")), reverse=True):
        try:
            pmc = json.load(open(path))
        except Exception:
            continue
        if pmc.get("lk_source_sha") == want and pmc.get("streams") == S and pmc.get("traffic_bytes_per_launch"):
            return int(pmc["traffic_bytes_per_launch"]), os.path.basename(path)
    return None, None


# --------------------------------------------------------------------------------------------------------------- main
def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.pmc:
        return run_pmc(args)
    maybe_spawn(args, argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with --nproc-per-node %d)" % (args.gpus, world, args.gpus))
    if args.stub:
        return run_stub(args, rank, world)

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists)")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit("bench.py: rank %d has no GPU (%d visible)" % (local_rank, torch.cuda.device_count()))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    affinity = bind_to_gpu_numa_node(local_rank)   # this rank's host threads next to its GPU (8 ranks share one host)

    import flvis_amd
    from flvis_amd import dist as fdist
    from flvis_amd import synth

    K, Wm = args.steps, args.warmup
    S = plan.streams_per_gpu(args.scaling, world, args.streams, args.total_streams)
    ypath = os.path.join(tempfile.gettempdir(), "flvis_bench_d435_stereo_%d.yaml" % rank)
    open(ypath, "w").write(synth.D435I_STEREO_YAML)
    cfg = flvis_amd.load_config(ypath)
    skip = cfg.skip_first_n_imgs
    epi = 0 if args.no_epilogue else plan.EPILOGUE
    nmax = plan.max_frames(K, Wm, skip, epilogue=epi) + (0 if args.no_h2d else H2D_LEG_FRAMES + 4)   # (+ the host-image leg's own frames)
    # multi-lane trackers run on their own streams: a private context stream keeps torch's (null) stream out of the frame loop;
    # the rendered inputs are synchronised explicitly below
    own_stream = int(os.environ.get("FLVIS_LANES", "1") or 1) > 1 or os.environ.get("FLVIS_BENCH_OWN_STREAM", "0") != "0"   # (A/B knob)
    ctx = flvis_amd.Context(local_rank, own_stream=own_stream)
    trk = flvis_amd.Tracker(ctx, cfg, S, seed_base=0xF1715 + rank * S, traj_capacity=nmax)
    lib = ctx._lib
    wlm = 0 if args.no_local_map else 1
    if lib.flvis_tracker_lanes(ctx._h) > 1 and args.input_hold > 0:
        # every frame of the run is resident in its own buffer: the lanes need not wait for each other at frame boundaries
        ctx._check(lib.flvis_set_input_hold(ctx._h, args.input_hold), "set_input_hold")

    # ---- synthetic inputs
    stream_ids = fdist.shard_streams(rank, world, S)
    trajs = [synth.Trajectory(s) for s in stream_ids]
    rnd = synth.Renderer(dev)
    SPF = 16
    imu = np.zeros((nmax, S, SPF, 7))
    imu_cnt = np.zeros((nmax, S), np.int32)
    for i, s in enumerate(stream_ids):
        t_prev = -1.0 / synth.FRAME_HZ
        for f in range(nmax):
            t = f / synth.FRAME_HZ
            smp = synth.imu_samples(trajs[i], s, t_prev, t)
            imu[f, i, :len(smp)] = smp
            imu_cnt[f, i] = len(smp)
            t_prev = t
    times = np.array([[f / synth.FRAME_HZ] * S for f in range(nmax)])
    n_mt = 0
    if rank == 0 and args.cpu_mt_frames > 0:
        n_mt = min(S, os.cpu_count() or 1)
    cpu_first, n_cpu = plan.cpu_sample(args.cpu_frames if rank == 0 else 0, plan.frame_schedule(K, Wm, skip, epilogue=epi))
    n_keep = max(n_cpu, args.cpu_mt_frames if n_mt else 0)
    host_frames = {}   # frame -> (img0, img1) host copies of the first n_mt (>= 1) streams, for the CPU legs only

    def render(f):
        fr = rnd.stereo_frame(trajs, f / synth.FRAME_HZ, f)
        if own_stream:
            torch.cuda.synchronize()
        if rank == 0 and cpu_first <= f < cpu_first + n_keep:
            ns = max(1, n_mt) if f < cpu_first + (args.cpu_mt_frames if n_mt else 0) else 1
            host_frames[f] = (fr[0][:ns].cpu().numpy(), fr[1][:ns].cpu().numpy())
        return fr

    out_buf = (flvis_amd.FrameOut * S)()

    def feed(f, fr, want_out=False):
        rc = lib.flvis_imu_feed_all(ctx._h, imu_cnt[f].ctypes.data_as(C.POINTER(C.c_int)),
                                    imu[f].ctypes.data_as(C.POINTER(C.c_double)), SPF)
        if rc:
            ctx._check(rc, "imu_feed_all")
        rc = lib.flvis_image_feed(ctx._h, C.c_void_p(fr[0].data_ptr()), C.c_void_p(fr[1].data_ptr()),
                                  times[f].ctypes.data_as(C.POINTER(C.c_double)),
                                  C.cast(out_buf, C.c_void_p) if want_out else C.c_void_p(0), wlm)
        if rc:
            ctx._check(rc, "image_feed")

    def n_tracking():
        return sum(1 for o in out_buf if o.state == 1)

    def barrier():
        if world > 1:
            dist.barrier()

    # ---- untimed pre-roll: skipped start-up frames, IMU initialisation, init_frame, first tracked frames.  Rendered on the
    # fly (the skipped frames are never looked at: one rendered frame stands in for all of them)
    t_pre = time.time()
    standin = render(skip)
    extra = 0
    sched = plan.frame_schedule(K, Wm, skip, epilogue=epi)
    f = 0
    while True:
        p_end = sched["preroll"][1]
        while f < p_end:
            feed(f, standin if f < skip else (standin if f == skip else render(f)), want_out=(f == p_end - 1))
            f += 1
        ok_local = plan.steady_state(n_tracking(), S, trk.local_map_counts()[1] if wlm else [], bool(wlm))
        ok_all = fdist.exchange_results(torch.zeros((1, 7), dtype=torch.float64, device=dev), [0 if ok_local else 1], dev)[1][0] == 0
        if ok_all:
            break
        if extra >= plan.EXTRA_SETTLE_MAX:
            raise SystemExit("bench.py: after %d frames %d of %d streams are in the Tracking state and %d have optimised their window -- "
                             "refusing to time a non-steady-state region"
                             % (f, n_tracking(), S, int((trk.local_map_counts()[1] >= 1).sum()) if wlm else 0))
        extra += 1
        sched = plan.frame_schedule(K, Wm, skip, epilogue=epi, extra_settle=extra)
    tracking_at_start = n_tracking()
    t_pre = time.time() - t_pre
    # ---- inputs of warm-up, timed region and epilogue: resident in HBM before the clock starts
    frames = {}
    for g in range(sched["warmup"][0], sched["n_frames"]):
        frames[g] = render(g)
    torch.cuda.synchronize()

    nst = lib.flvis_prof_stage_count()
    lib.flvis_prof_stage_name.restype = C.c_char_p
    lib.flvis_prof_enable_stages.argtypes = [C.c_void_p, C.c_int, C.c_uint64]
    names = [lib.flvis_prof_stage_name(i).decode() for i in range(nst)]
    lk_idx = [i for i, n in enumerate(names) if n.startswith("lk_track")]
    chain_idx = names.index("frame(chain)")
    timed_mask = sum(1 << i for i in lk_idx) | (1 << chain_idx)
    if os.environ.get("FLVIS_BENCH_FRAMES"):   # diagnosis: every stage carries events in the timed region (slows it a little)
        timed_mask = (1 << nst) - 1
    if os.environ.get("FLVIS_BENCH_NO_EVENTS"):   # diagnosis: what the LK launches' events cost the timed region (no roofline from such a run)
        timed_mask = 1 << chain_idx

    def read_stages():
        ms = (C.c_double * nst)()
        nrec = C.c_int(0)
        ctx._check(lib.flvis_prof_read(ctx._h, ms, C.byref(nrec)), "prof_read")
        return {names[i]: ms[i] / max(nrec.value, 1) for i in range(nst)}

    def read_steps(stage, n):
        buf = (C.c_double * n)()
        got = lib.flvis_prof_read_steps(ctx._h, stage, buf, n)
        if got < 0:
            ctx._check(got, "prof_read_steps")
        return list(buf[:got])

    for g in range(*sched["warmup"]):
        feed(g, frames[g])
    ctx._check(lib.flvis_hip_synchronize(ctx._h), "synchronize")
    torch.cuda.synchronize()
    kf_at_start, ba_at_start = (int(x.sum()) for x in trk.local_map_counts())   # queues drained: counts at the start of the clock

    # ---- timed region: only k_lk_track (the dominant kernel, 2 launches per step) and the whole-frame chain carry HIP events
    ctx._check(lib.flvis_prof_enable_stages(ctx._h, K, C.c_uint64(timed_mask)), "prof_enable")
    barrier()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def host_times():
        h = (C.c_double * 3)()
        lib.flvis_debug_host_times(ctx._h, h)
        return list(h)

    # the K timed steps are ONE call into the library (flvis_run_steps): no Python between frames
    class FlvisStep(C.Structure):
        _fields_ = [("d_img0", C.c_void_p), ("d_img1", C.c_void_p), ("h_times", C.c_void_p), ("h_imu_counts", C.c_void_p),
                    ("h_imu_samples", C.c_void_p), ("imu_samples_per_stream", C.c_int)]
    steps_arr = (FlvisStep * K)()
    for j, g in enumerate(range(*sched["timed"])):
        steps_arr[j].d_img0, steps_arr[j].d_img1 = frames[g][0].data_ptr(), frames[g][1].data_ptr()
        steps_arr[j].h_times = times[g].ctypes.data
        steps_arr[j].h_imu_counts, steps_arr[j].h_imu_samples = imu_cnt[g].ctypes.data, imu[g].ctypes.data
        steps_arr[j].imu_samples_per_stream = SPF
    call_ms = (C.c_double * K)()
    lib.flvis_run_steps.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    host0 = host_times()
    t0 = time.perf_counter()
    e0.record()
    ctx._check(lib.flvis_run_steps(ctx._h, K, C.cast(steps_arr, C.c_void_p), wlm, C.cast(call_ms, C.c_void_p)), "run_steps")
    host_call_ms = list(call_ms)
    e1.record()
    t_issue = time.perf_counter() - t0   # host loop over the K steps (the calls return before the GPU has run them, but block on the
    host1 = host_times()                 # pinned upload ring once the host is 4 frames ahead); host1 - host0: time inside image_feed
    ctx._check(lib.flvis_hip_synchronize(ctx._h), "synchronize")  # everything enqueued AND every queued keyframe consumed by the local map
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    gpu_ms = e0.elapsed_time(e1)
    elapsed = fdist.max_over_ranks(elapsed, dev)
    kf_in_region, ba_in_region = (int(x.sum()) for x in trk.local_map_counts())  # (the queues were drained inside the clock)
    kf_in_region -= kf_at_start
    ba_in_region -= ba_at_start
    lk_stages = read_stages()
    chain_ms = read_steps(chain_idx, K)
    # ---- untimed epilogue: all stages
    stages = None
    lk_iters = None
    if epi:
        # the last frames of the epilogue count LK iterations instead of carrying stage events (the counting costs one contended atomic
        # per point and level and would distort the stage times)
        n_stat = min(4, epi // 4)
        ep0, ep1 = sched["epilogue"]
        ctx._check(lib.flvis_prof_enable_stages(ctx._h, epi - n_stat, C.c_uint64((1 << 64) - 1)), "prof_enable")
        for g in range(ep0, ep1 - n_stat):
            feed(g, frames[g])
        torch.cuda.synchronize()
        stages = read_stages()
        dbg0 = (C.c_int64 * 64)()
        ctx._check(lib.flvis_debug_counters(ctx._h, dbg0), "debug_counters")
        ctx._check(lib.flvis_debug_lk_stats(ctx._h, 1), "lk_stats")
        for g in range(ep1 - n_stat, ep1):
            feed(g, frames[g])
        torch.cuda.synchronize()
        ctx._check(lib.flvis_debug_lk_stats(ctx._h, 0), "lk_stats")
        dbg = (C.c_int64 * 64)()
        ctx._check(lib.flvis_debug_counters(ctx._h, dbg), "debug_counters")
        lk_iters = {}
        for tag, base in (("temporal", 36), ("stereo", 48)):
            per = {}
            for lv in range(6):
                if dbg[base + 2 * lv + 1]:
                    per["level%d" % lv] = {"points_per_launch": round(dbg[base + 2 * lv + 1] / max(n_stat, 1), 1),
                                           "mean_iterations": round(dbg[base + 2 * lv] / dbg[base + 2 * lv + 1], 2)}
            tot_it = sum(dbg[base + 2 * lv] for lv in range(6))
            lk_iters[tag] = {"per_level": per, "window_evaluations_per_launch": round(tot_it / max(n_stat, 1), 1)}
        lk_iters["template_cache"] = {"templates_taken_per_frame": round((dbg[61] - dbg0[61]) / max(n_stat, 1), 1),
                                      "slow_patch_stagings_per_frame": round((dbg[62] - dbg0[62]) / max(n_stat, 1), 1),
                                      "slow_region_stagings_per_frame": round((dbg[63] - dbg0[63]) / max(n_stat, 1), 1),
                                      "note": "temporal LK: templates read back from the cache the previous frame's stereo LK wrote; "
                                              "stagings that left the pyramids' physical border (both launches)"}
    last = sched["n_frames"] - 1

    # ---- results: tracker health, final poses; the path's only exchange: all-gather poses, all-reduce counters
    cnt = trk.counters()
    rows = np.stack([trk.trajectory(i, last, 1)[0] for i in range(S)])
    tracking = int((rows[:, 8].astype(int) & 15 == 1).sum())
    all_poses, csum = fdist.exchange_results(torch.from_numpy(rows[:, 1:8].copy()).to(dev),
                                             [cnt[0], cnt[1], cnt[2], tracking, kf_in_region, ba_in_region], dev)
    assert all_poses.shape[0] == world * S
    if not plan.region_is_ba_steady(csum[4], csum[5], bool(wlm)):
        raise SystemExit("bench.py: %d keyframes but %d local-map optimisations inside the timed region -- not the steady state of the "
                         "path (one optimisation per keyframe)" % (csum[4], csum[5]))
    if csum[3] == 0:
        raise SystemExit("bench.py: no stream is tracking at the end of the run -- the measured region is not the hot path")
    dbg_end = (C.c_int64 * 64)()
    ctx._check(lib.flvis_debug_counters(ctx._h, dbg_end), "debug_counters")
    if wlm and dbg_end[3]:   # (a stream's keyframe queue was full: the local map did not see every keyframe the tracker made)
        raise SystemExit("bench.py: %d keyframes were dropped between the tracker and the local map -- not the reference's path" % dbg_end[3])

    if rank == 0:
        total_frames = world * S * K
        value = total_frames / elapsed
        # dominant kernel ON THE CRITICAL PATH: k_lk_track (two launches per step: temporal + stereo)
        lk_ms = [lk_stages[names[i]] for i in lk_idx]
        dom_ms = sum(lk_ms) / len(lk_ms)
        dom_bytes = LK_ALG_BYTES * S
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        traffic, traffic_src = read_traffic(S)
        out = {
            "metric": METRIC,
            "metric_note": "synthetic 640x480 stereo+IMU streams (no dataset offline): ATE is reported against the CPU reference "
                           "and the synthetic ground truth, not on EuRoC MH_05",
            "value": round(value, 1), "unit": "frames/s",
            "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": round(elapsed / K * 1e3, 4),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "u8/f32 LK+GFTT, f64 geometry+BA",
            "data": "synthetic",
            "config": {"workload": "%dxMI355X: batch of %d independent 640x480 synthetic stereo+IMU streams per GPU, full HIP "
                                   "front-end + batched Schur BA (BASELINE.json configs[%d])" % (world, S, 3 if world == 1 else 4),
                       "streams_per_gpu": S, "streams_total": world * S, "lanes_per_gpu": lib.flvis_tracker_lanes(ctx._h),
                       "window_size": cfg.window_size, "local_map": bool(wlm),
                       "preroll_frames": sched["preroll"][1], "streams_tracking_at_start": tracking_at_start,
                       "streams_tracking_at_end": int(csum[3]), "keyframes_in_run": int(csum[1]), "ba_runs_in_run": int(csum[2]),
                       "keyframes_in_timed_region": int(csum[4]), "ba_runs_in_timed_region": int(csum[5]),
                       "gpu_ms_per_step_events": round(gpu_ms / K, 4),
                       "host_loop_ms_per_step": round(t_issue / K * 1e3, 4),
                       "host_loop": "flvis_run_steps: one C call for the %d timed steps" % K, "host_affinity": affinity,
                       "host_enqueue_ms_per_step": round((host1[0] - host1[1] - host0[0] + host0[1]) / K, 4),
                       "input_hold_frames": args.input_hold if lib.flvis_tracker_lanes(ctx._h) > 1 else 0},
            "latency_ms": {"gpu_frame_chain_p50": round(plan.percentile(chain_ms, 50), 4),
                           "gpu_frame_chain_p99": round(plan.percentile(chain_ms, 99), 4),
                           "note": "HIP events around the whole main-stream chain of one batch step (all %d streams of the GPU "
                                   "advance together), timed region" % S,
                           # where the wall time of the timed region goes: the frames' own chains, the idle time between them on
                           # the tracking stream, and the time after the last frame in which the local map finishes its keyframes
                           "timed_region_ms": {"wall": round(elapsed * 1e3, 3), "tracking_stream_span": round(gpu_ms, 3),
                                               "sum_of_frame_chains": round(sum(chain_ms), 3),
                                               "local_map_tail_after_last_frame": round(elapsed * 1e3 - gpu_ms, 3)},
                           "frame_chain_ms": [round(v, 3) for v in chain_ms] if os.environ.get("FLVIS_BENCH_FRAMES") else None,
                           "host_call_ms": [round(v, 3) for v in host_call_ms] if os.environ.get("FLVIS_BENCH_FRAMES") else None,
                           "host_in_image_feed_ms": {"total": round(host1[0] - host0[0], 3), "waiting_for_a_staging_slot": round(host1[1] - host0[1], 3)},
                           "frame_stage_ms": {names[i]: [round(v, 3) for v in read_steps(i, K)] for i in range(nst)}
                           if os.environ.get("FLVIS_BENCH_FRAMES") else None} if chain_ms else None,
            "roofline": {"bound": "hbm", "kernel": "k_lk_track", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                         "avg_launch_ms": round(dom_ms, 4), "algorithmic_bytes_per_launch": dom_bytes, "launches_per_step": 2},
        }
        try:
            out["roofline"]["hbm_copy_measured"] = measure_copy_bandwidth(dev)
        except Exception as e:  # noqa: BLE001
            out.setdefault("leg_errors", []).append("copy bandwidth: %s" % e)
        if stages is not None:
            out["stages_ms_per_step"] = {k: round(v, 4) for k, v in stages.items()}
            out["stages_note"] = "per-stage times from %d untimed frames after the timed region (all stages bracketed by events)" % (epi - min(4, epi // 4))
            # ---- per-kernel roofline table: every kernel priced against the resource that bounds it
            try:
                from flvis_amd import roofline as rf
                dbg = (C.c_int64 * 64)()
                ctx._check(lib.flvis_debug_counters(ctx._h, dbg), "debug_counters")
                ba = {"runs": int(dbg[2]), "trials": int(dbg[4]), "trials_items": int(dbg[5]), "trials_landmarks": int(dbg[6]),
                      "trials_poses": int(dbg[7]), "ms_per_optimisation": (dbg[60] * 1e-5 / dbg[2]) if dbg[2] else None,
                      "chunked_runs": int(dbg[28]), "ms_per_chunked_optimisation": (dbg[29] * 1e-5 / dbg[28]) if dbg[28] else None,
                      "worker_ms_per_launch": stages.get("ba_worker(launch)")}
                kpmc, kpmc_src = read_kernel_pmc(S)
                copy_gbs = out["roofline"].get("hbm_copy_measured", {}).get("GB/s")
                out["roofline"]["kernels"] = rf.kernel_table(stages, S, 640, 480, copy_gbs, kpmc, ba if wlm else None, elapsed / K * 1e3)
                if lk_iters:
                    out["roofline"]["lk_iterations"] = lk_iters
                    for r in out["roofline"]["kernels"]:   # instruction budget of LK: wave instructions per iteration (31 x 31 window)
                        for tag in ("temporal", "stereo"):
                            if r["kernel"] == "k_lk_track (%s)" % tag and r.get("valu_insts_per_launch") and lk_iters[tag]["window_evaluations_per_launch"]:
                                per_it = r["valu_insts_per_launch"] / lk_iters[tag]["window_evaluations_per_launch"]
                                r["valu_insts_per_iteration"] = round(per_it, 1)
                                r["valu_insts_per_window_pixel"] = round(per_it * 64 / 961.0, 2)
                # the ceiling the dominant kernel actually runs against (the contract's fields above price it on HBM bytes): VALU issue
                lkrows = [r for r in out["roofline"]["kernels"] if r["kernel"].startswith("k_lk_track") and r.get("valu_issue_frac")]
                if lkrows:
                    out["roofline"]["binding_resource"] = {
                        "resource": "valu_issue", "frac": round(sum(r["valu_issue_frac"] for r in lkrows) / len(lkrows), 4),
                        "peak": rf.VALU_ISSUE_PEAK_GINST, "unit": "G wave-instructions/s",
                        "per_launch": {r["kernel"]: r["valu_issue_frac"] for r in lkrows},
                        "note": "k_lk_track is bound by instruction issue, not by HBM (frac above: algorithmic bytes / 8 TB/s); counters of "
                                "profiles/*_kernel_pmc.json at this source tree"}
                    allrows = [r for r in lkrows if r.get("issue_frac_counted_classes")]
                    if allrows:   # vector + scalar + LDS + memory instructions against the same one-per-4-cycles-and-SIMD ceiling
                        out["roofline"]["binding_resource"]["issue_frac_counted_classes"] = round(
                            sum(r["issue_frac_counted_classes"] for r in allrows) / len(allrows), 4)
                        out["roofline"]["binding_resource"]["issue_note"] = (
                            "a SIMD issues one instruction of ANY class per 4 cycles (profiles/r06_chain_ab.md: the corner response saturates "
                            "at exactly that); issue_frac_counted_classes = (VALU + SALU + LDS + SMEM + VMEM) wave instructions / peak -- waits, "
                            "no-ops and branches take slots too and are not counted")
                out["roofline"]["kernels_note"] = (
                    "avg_launch_ms: HIP events of this run (epilogue frames); counters / rocprof_avg_launch_ms: %s; valu_issue_frac against "
                    "%.1f G wave-instructions/s (1024 SIMDs x 2.4 GHz / 4); k_ba_worker: SURVEY 8d's flop formula on the kernel's own trial / "
                    "observation / landmark counters, time = the kernel's wall clock per optimisation; k_ransac_pnp's event time is ~0.11 ms above its "
                    "rocprof duration since the joins are folded into the kernels: with every stage bracketed by event packets (epilogue only) the "
                    "corner response reaches the chip ahead of it -- the timed region carries no such packets (rocprof_avg_launch_ms is the kernel there)"
                    % (kpmc_src or "no PMC file matches this source tree (run `bench.py --pmc`)", rf.VALU_ISSUE_PEAK_GINST))
            except Exception as e:  # noqa: BLE001
                out.setdefault("leg_errors", []).append("kernel table: %s: %s" % (type(e).__name__, e))
        # ---- legs that must never suppress the GPU line: host-image variant, CPU baselines, ATE
        for leg in (leg_h2d, leg_cpu, leg_euroc):
            try:
                leg(locals())
            except Exception as e:  # noqa: BLE001
                out.setdefault("leg_errors", []).append("%s: %s: %s" % (leg.__name__, type(e).__name__, e))
        print(json.dumps(out))
        sys.stdout.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


def measure_copy_bandwidth(dev, nbytes=1 << 30, reps=10):
    """What this GPU's HBM actually sustains on a plain device-to-device copy (read + write counted), next to the 8 TB/s peak the
    roofline fraction is priced against."""
    import torch
    src = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    dst = torch.empty_like(src)
    src.fill_(3)
    dst.copy_(src)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return {"GB/s": round(2 * nbytes / (ms * 1e-3) / 1e9, 1), "bytes_copied": nbytes, "note": "torch device-to-device copy of 1 GiB, read + write bytes"}


def leg_h2d(L):
    """Same K timed steps with the images handed over as HOST buffers (flvis_image_feed_host: pinned staging + async H2D of
    614,400 B per frame and stream).  Never `value`: reported beside it."""
    args, lib, ctx = L["args"], L["lib"], L["ctx"]
    if args.no_h2d or not hasattr(lib, "flvis_image_feed_host"):
        return
    import torch
    S, K, out, frames, sched = L["S"], L["K"], L["out"], L["frames"], L["sched"]
    imu, imu_cnt, times, SPF, wlm = L["imu"], L["imu_cnt"], L["times"], L["SPF"], L["wlm"]
    # continue the streams' timeline: re-feeding the epilogue's last frames would jump back in time, so fresh frames
    trajs, rnd, synth = L["trajs"], L["rnd"], L["synth"]
    WU = 4                       # untimed calls in front: the first one allocates the device staging buffers and the copy stream, and ONE
                                 # of the first three takes 75-85 ms on this stack (lazy set-up inside the runtime; which one varies from run
                                 # to run: profiles/r04_lk_ab.md) -- inside the clock it halved the leg's rate in one run of four
    n = H2D_LEG_FRAMES + WU      # (60 frames whatever K is: the local map's drain after the last frame -- ~2.5 ms, inside the clock -- weighs the same in
                                 # every line; round 6's first driver-argument line had 40 frames and read 0.92 x where the 60-frame line read 0.995 x)
    f0 = sched["n_frames"]
    if f0 + n > imu.shape[0]:
        n = imu.shape[0] - f0
    if n <= WU:
        return
    host = []
    huge = os.environ.get("FLVIS_H2D_HUGEPAGES", "0") != "0"   # (A/B knob: the caller's page-locked images on 2 MB pages)
    for f in range(f0, f0 + n):
        fr = rnd.stereo_frame(trajs, f / synth.FRAME_HZ, f)
        if huge:
            host.append(tuple(_pinned_on_huge_pages(x.cpu()) for x in fr))
        else:
            host.append((fr[0].cpu().pin_memory(), fr[1].cpu().pin_memory()))
    torch.cuda.synchronize()
    img_t = flvis_image_struct()
    descs = []   # the flvis_image descriptors of every frame (what a caller's capture loop holds anyway), built outside the clock
    for j, f in enumerate(range(f0, f0 + n)):
        a = (img_t * S)()
        b = (img_t * S)()
        for arr, t in ((a, host[j][0]), (b, host[j][1])):
            base, stride = t.data_ptr(), t.stride(0) * t.element_size()
            for s in range(S):
                arr[s].data = base + s * stride
                arr[s].width, arr[s].height, arr[s].pitch, arr[s].channels = 640, 480, 640, 1
                arr[s].t = times[f][s]
        descs.append((a, b))
    t0 = 0.0
    calls = []
    hft0, ht0 = (C.c_double * 4)(), (C.c_double * 3)()
    names, chain_idx = L["names"], L["chain_idx"]
    # ---- what the link gives, measured where the leg runs: (1) the two blocks of one frame uploaded alone (nothing else on the GPU), HIP events
    # around the copies; (2) the uploads of the leg's untimed calls, bracketed by timing events on the copy stream while frames run beside them
    diag = {"bytes_per_call": int(2 * S * 640 * 480)}
    try:
        tgt = [torch.empty_like(host[0][0], device=L["dev"]), torch.empty_like(host[0][1], device=L["dev"])]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for rep in range(6):
            if rep == 1:
                e0.record()
            tgt[0].copy_(host[rep % len(host)][0], non_blocking=True)
            tgt[1].copy_(host[rep % len(host)][1], non_blocking=True)
        e1.record()
        torch.cuda.synchronize()
        diag["upload_alone_GBs"] = round(5 * diag["bytes_per_call"] / (e0.elapsed_time(e1) * 1e-3) / 1e9, 2)
        del tgt
    except Exception as e:  # noqa: BLE001
        diag["upload_alone_error"] = str(e)
    diag["pcie"] = pcie_link_info(L["local_rank"])
    diag["pinned_numa_node"] = numa_node_of(host[0][0].data_ptr())
    diag["host_affinity_cpus"] = len(os.sched_getaffinity(0))
    has_timing = hasattr(lib, "flvis_debug_host_feed_timing")
    keep_timing = os.environ.get("FLVIS_BENCH_H2D_TIMING", "0") != "0"   # (A/B knob: the timing events stay on inside the clock)
    for j, f in enumerate(range(f0, f0 + n)):
        if j == 1 and has_timing:
            lib.flvis_debug_host_feed_timing(ctx._h, 1, None)     # (from the second call: the first one allocates the staging)
        if j == WU:
            if has_timing and not keep_timing:
                lib.flvis_debug_host_feed_timing(ctx._h, 0, None)
            ctx._check(lib.flvis_hip_synchronize(ctx._h), "synchronize")
            # the same HIP events as in the timed region: the LK launches and the whole main-stream chain of every frame of the leg
            ctx._check(lib.flvis_prof_enable_stages(ctx._h, n - WU, C.c_uint64(L["timed_mask"])), "prof_enable")
            lib.flvis_debug_host_feed_times(ctx._h, hft0)
            lib.flvis_debug_host_times(ctx._h, ht0)
            t0 = time.perf_counter()
        rc = lib.flvis_imu_feed_all(ctx._h, imu_cnt[f].ctypes.data_as(C.POINTER(C.c_int)), imu[f].ctypes.data_as(C.POINTER(C.c_double)), SPF)
        if rc:
            ctx._check(rc, "imu_feed_all")
        a, b = descs[j]
        tc = time.perf_counter()
        rc = lib.flvis_image_feed_host(ctx._h, a, b, C.c_void_p(0), wlm, 1)
        calls.append((time.perf_counter() - tc) * 1e3)
        if rc:
            ctx._check(rc, "image_feed_host")
    t_loop = time.perf_counter() - t0
    hft1, ht1 = (C.c_double * 4)(), (C.c_double * 3)()
    lib.flvis_debug_host_feed_times(ctx._h, hft1)
    lib.flvis_debug_host_times(ctx._h, ht1)
    ctx._check(lib.flvis_hip_synchronize(ctx._h), "synchronize")
    dt = time.perf_counter() - t0
    n_all = n
    n -= WU
    out["with_h2d"] = {"value": round(L["world"] * S * n / dt, 1), "unit": "frames/s", "steps": n,
                       "note": "images handed over as pinned host buffers (flvis_image_feed_host, 614,400 B per stereo frame over "
                               "PCIe); %d untimed calls in front; rank 0's rate x n_gpus; never `value`" % WU}
    ncall = max(hft1[3] - hft0[3], 1)
    try:
        leg_stages = L["read_stages"]()
        leg_chain = L["read_steps"](chain_idx, n)
        out["with_h2d"]["gpu_frame_chain_p50_ms"] = round(plan.percentile(leg_chain, 50), 4)
        out["with_h2d"]["lk_ms"] = {k: round(v, 4) for k, v in leg_stages.items() if k.startswith("lk_track")}
        if os.environ.get("FLVIS_BENCH_FRAMES"):   # (every stage carries events then)
            out["with_h2d"]["stages_ms"] = {k: round(v, 4) for k, v in leg_stages.items()}
    except Exception as e:  # noqa: BLE001
        out["with_h2d"]["stage_error"] = str(e)
    if has_timing:
        up = (C.c_double * 3)()
        lib.flvis_debug_host_feed_timing(ctx._h, 0, up)
        if up[2] > 0 and up[0] > 0:
            diag["upload_GBs"] = round(up[1] / (up[0] * 1e-3) / 1e9, 2)
            diag["upload_ms_per_call"] = round(up[0] / up[2], 4)
            diag["uploads_timed"] = int(up[2])
    link = diag.get("upload_GBs") or diag.get("upload_alone_GBs")
    if link:   # what the link alone would allow: S frames per bytes_per_call / link rate
        diag["link_bound_frames_per_s"] = round(L["world"] * S / (diag["bytes_per_call"] / (link * 1e9)), 1)
    out["with_h2d"]["link"] = diag
    out["with_h2d"]["ratio_to_value"] = round(out["with_h2d"]["value"] / out["value"], 3) if out.get("value") else None
    out["with_h2d"]["host_ms_per_call"] = {"loop": round(t_loop * 1e3 / ncall, 3),
                                           "waiting_for_the_previous_uploads": round((hft1[0] - hft0[0]) / ncall, 3),
                                           "issuing_uploads": round((hft1[1] - hft0[1]) / ncall, 3),
                                           "in_image_feed": round((hft1[2] - hft0[2]) / ncall, 3),
                                           "of_it_waiting_for_the_host_lead": round((ht1[1] - ht0[1]) / ncall, 3)}
    # ---- what the leg computed: the poses of ALL its frames and streams against a re-run of the whole sequence through the resident
    # path (flvis_image_feed on a second context, same seeds, same frames; that path is the one the lockstep tests hold against the
    # oracle).  An upload that raced a frame, or a staging slot refilled too early, shows as a differing pose.
    if os.environ.get("FLVIS_BENCH_H2D_NOCHECK", "0") != "0":   # (diagnosis only, e.g. under AMD_LOG_LEVEL: the leg's calls are the log's tail)
        out["with_h2d"]["poses_bit_identical"] = None
        out["with_h2d"]["value_unchecked"] = out["with_h2d"]["value"]
        out["with_h2d"]["value"] = None
        return
    import flvis_amd
    trk = L["trk"]
    rows_h = np.stack([trk.trajectory(i, f0, n_all) for i in range(S)])
    ctx2 = flvis_amd.Context(L["local_rank"])
    try:
        trk2 = flvis_amd.Tracker(ctx2, L["cfg"], S, seed_base=0xF1715 + L["rank"] * S, traj_capacity=f0 + n_all)
        skip, standin = L["skip"], None
        for f in range(f0 + n_all):
            if f >= f0:
                fr = (host[f - f0][0].to(L["dev"]), host[f - f0][1].to(L["dev"]))
            elif f in frames:
                fr = frames[f]
            elif f <= skip:
                standin = standin if standin is not None else rnd.stereo_frame(trajs, skip / synth.FRAME_HZ, skip)
                fr = standin
            else:
                fr = rnd.stereo_frame(trajs, f / synth.FRAME_HZ, f)
            rc = lib.flvis_imu_feed_all(ctx2._h, imu_cnt[f].ctypes.data_as(C.POINTER(C.c_int)), imu[f].ctypes.data_as(C.POINTER(C.c_double)), SPF)
            if rc:
                ctx2._check(rc, "imu_feed_all")
            trk2.image_feed(fr[0], fr[1], times[f], want_out=False, with_local_map=False)
            ctx2.synchronize()
        rows_r = np.stack([trk2.trajectory(i, f0, n_all) for i in range(S)])
        del trk2
    finally:
        ctx2.close()
    same = bool(np.array_equal(rows_h, rows_r))
    tracked = int(((rows_h[:, :, 8].astype(int) & 15) == 1).sum())
    out["with_h2d"]["poses_bit_identical"] = same
    out["with_h2d"]["poses_compared"] = {"streams": S, "frames": n_all, "tracked_poses": tracked,
                                         "against": "a resident re-run of the whole sequence (flvis_image_feed, second context)"}
    if not same or tracked == 0:
        bad = np.argwhere(np.any(rows_h != rows_r, axis=2))
        out["with_h2d"]["rate_of_the_wrong_results"] = out["with_h2d"]["value"]
        out["with_h2d"]["value"] = None      # a rate of wrong results is not a measurement (the line itself survives: leg_errors)
        raise RuntimeError("the host-image leg's poses differ from the resident re-run (first at stream %d, frame %d; %d tracked poses)"
                           % (tuple(int(v) for v in bad[0]) + (tracked,) if len(bad) else (-1, -1, tracked)))
    if os.environ.get("FLVIS_BENCH_FRAMES"):
        out["with_h2d"]["host_call_ms"] = [round(v, 3) for v in calls]
        out["with_h2d"]["loop_ms"] = round(t_loop * 1e3, 3)
        out["with_h2d"]["total_ms"] = round(dt * 1e3, 3)


def leg_euroc(L):
    """BASELINE.json configs[0] / configs[1]: the EuRoC sequence of --euroc / FLVIS_EUROC_DIR (MH_05_difficult is the one the metric names)
    through scripts/run_sequence.py on both backends -- the reference's EuRoC calibration and parameters (launch/EuRoC_MAV/euroc.yaml, the
    values flvis_amd.synth.EUROC_LIKE_YAML carries), front-end + local map -- and the three numbers of the metric: ATE of the HIP path and of
    the CPU restatement against the ground truth, and camera-centre RMSE of one against the other.  No dataset exists offline: the line
    then says so instead of dropping the key."""
    args, out = L["args"], L["out"]
    d = args.euroc
    if not d or not os.path.isdir(d):
        out["euroc"] = {"status": "dataset missing", "looked_in": d or "(no --euroc DIR / FLVIS_EUROC_DIR given)",
                        "would_run": "scripts/run_sequence.py DIR <euroc.yaml values> OUT --local-map, --backend hip and --backend cpu; "
                                     "ATE vs ground truth of each + HIP vs CPU"}
        return
    from flvis_amd import synth, traj_io
    frames = int(os.environ.get("FLVIS_EUROC_FRAMES", "400"))   # (bounded: the CPU restatement runs ~25 frames/s)
    tmp = tempfile.mkdtemp(prefix="flvis_euroc_")
    ypath = os.path.join(tmp, "euroc.yaml")
    open(ypath, "w").write(synth.EUROC_LIKE_YAML)
    res = {"status": "ran", "sequence": d, "frames": frames}
    for backend in ("hip", "cpu"):
        est = os.path.join(tmp, "est_%s.txt" % backend)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "run_sequence.py"), d, ypath, est, "--backend", backend,
                            "--frames", str(frames), "--local-map"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1800)
        if r.returncode != 0:
            res[backend] = {"error": r.stderr.decode(errors="replace")[-400:]}
            continue
        try:
            res[backend] = json.loads(r.stdout.decode().strip().splitlines()[-1])
        except Exception as e:  # noqa: BLE001
            res[backend] = {"error": "unparsable output: %s" % e}
    a, b = os.path.join(tmp, "est_hip.txt"), os.path.join(tmp, "est_cpu.txt")
    if os.path.exists(a) and os.path.exists(b):
        rmse, n = traj_io.ate_from_files(a, b)
        res["hip_vs_cpu_rmse_m"], res["hip_vs_cpu_poses"] = rmse, n
        ha, ca = (res.get("hip") or {}).get("ate_rmse_m"), (res.get("cpu") or {}).get("ate_rmse_m")
        if ha is not None and ca:
            res["ate_relative_difference"] = abs(ha - ca) / ca      # north_star: within 1 % of the CPU reference
    out["euroc"] = res


_HUGE_KEEP = []


def _pinned_on_huge_pages(t):
    """A copy of the host tensor t in anonymous memory on 2 MB pages (madvise(MADV_HUGEPAGE)), page-locked with hipHostRegister: what a
    capture loop that allocates its frame buffers that way hands to flvis_image_feed_host."""
    import mmap
    import torch
    HUGE = 2 << 20
    nbytes = t.numel() * t.element_size()
    size = (nbytes + HUGE - 1) // HUGE * HUGE
    m = mmap.mmap(-1, size + HUGE, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS)
    raw = (C.c_char * (size + HUGE)).from_buffer(m)
    base = (C.addressof(raw) + HUGE - 1) // HUGE * HUGE
    libc = C.CDLL(None, use_errno=True)
    libc.madvise.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    libc.madvise(C.c_void_p(base), size, 14)          # MADV_HUGEPAGE
    C.memset(C.c_void_p(base), 0, size)               # fault the pages in (as huge pages where the kernel has them)
    hip = C.CDLL("libamdhip64.so")
    hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
    rc = hip.hipHostRegister(C.c_void_p(base), size, 0)
    if rc != 0:
        raise RuntimeError("hipHostRegister failed (%d)" % rc)
    arr = np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(base))
    out = torch.from_numpy(arr).view(t.dtype).view(t.shape)
    out.copy_(t)
    _HUGE_KEEP.append((m, raw))
    return out


def flvis_image_struct():
    class FlvisImage(C.Structure):
        _fields_ = [("data", C.c_void_p), ("width", C.c_int), ("height", C.c_int), ("pitch", C.c_int), ("channels", C.c_int),
                    ("t", C.c_double)]
    return FlvisImage


def leg_cpu(L):
    """CPU baselines on this box's host cores (the oracle = a port of the reference path; kind "port"): (i) one stream on one
    thread, (ii) one thread per core, one stream per thread -- both on a bounded sample of the same workload, driven natively
    (oracle/ref_runner.cpp) -- and the ATE of the GPU trajectory against the CPU one."""
    args, out, cfg = L["args"], L["out"], L["cfg"]
    n_cpu, cpu_first, n_mt, host_frames = L["n_cpu"], L["cpu_first"], L["n_mt"], L["host_frames"]
    imu, imu_cnt, synth, trk, trajs, wlm, skip = L["imu"], L["imu_cnt"], L["synth"], L["trk"], L["trajs"], L["wlm"], L["skip"]
    have = 0
    while cpu_first + have in host_frames:
        have += 1
    n_cpu = min(n_cpu, have)
    if n_cpu <= 0:
        return
    olib, build = load_oracle_lib()
    hf = [host_frames[cpu_first + j] for j in range(have)]
    tc, cpu_pos, cpu_state, fms = cpu_run_streams(olib, cfg, 1, 1, cpu_first, n_cpu, hf, imu, imu_cnt, 0xF1715, synth.FRAME_HZ, wlm)
    cpu_pos, cpu_state, lat = cpu_pos[0], cpu_state[0], list(fms[0])
    out["cpu_baseline"] = {"value": round(n_cpu / tc, 2), "unit": "frames/s", "cores": 1, "kind": "port",
                           "sample": "stream 0, %d frames after the %d skipped start-up frames of the same synthetic workload, oracle "
                                     "front-end + local-map BA, one thread, g++ %s (%d host cores present)"
                                     % (n_cpu, skip, build, os.cpu_count()),
                           "latency_ms_p50": round(plan.percentile(lat, 50), 3), "latency_ms_p99": round(plan.percentile(lat, 99), 3)}
    # (ii) one stream per thread, as many streams as cores (bounded by the streams of the batch)
    nf = 0
    while nf < args.cpu_mt_frames and cpu_first + nf in host_frames and host_frames[cpu_first + nf][0].shape[0] >= n_mt:
        nf += 1
    if n_mt > 1 and nf > 0:
        tj, _, st_mt, _ = cpu_run_streams(olib, cfg, n_mt, n_mt, cpu_first, nf, hf, imu, imu_cnt, 0xF1715, synth.FRAME_HZ, wlm)
        out["cpu_baseline"]["multi"] = {"value": round(n_mt * nf / tj, 2), "unit": "frames/s", "cores": n_mt,
                                        "sample": "%d streams x %d frames, one native host thread per stream (the reference runs 1-2 threads "
                                                  "per stream), wall time of the whole job; %d of %d stream-frames in the Tracking state"
                                                  % (n_mt, nf, int((st_mt == 1).sum()), n_mt * nf)}
    # ATE of the GPU trajectory of stream 0 against the CPU reference on the same frames (camera centres, no alignment: both
    # run from the same initial state), and both against the synthetic ground truth.  The CPU reference here is the IEEE
    # op-by-op build (the timing build above contracts multiply-adds, which moves its trajectory by ~1 mm)
    _, cpu_pos, cpu_state, _ = cpu_run_streams(load_checker_lib(), cfg, 1, 1, cpu_first, n_cpu, hf, imu, imu_cnt, 0xF1715, synth.FRAME_HZ, wlm)
    cpu_pos, cpu_state = cpu_pos[0], cpu_state[0]
    grow = trk.trajectory(0, cpu_first, n_cpu)
    sel = [j for j in range(n_cpu) if cpu_state[j] == 1 and (int(grow[j, 8]) & 15) == 1]
    if len(sel) >= 3:
        from flvis_amd import traj_io
        gc = np.array([centre(grow[j, 1:8]) for j in sel])
        cc = np.array([centre(cpu_pos[j]) for j in sel])
        gt = np.array([-(trajs[0].T_c_w((cpu_first + j) / synth.FRAME_HZ)[0]).T @ trajs[0].T_c_w((cpu_first + j) / synth.FRAME_HZ)[1]
                       for j in sel])
        ate_gpu, ate_cpu = traj_io.ate_rmse(gc, gt), traj_io.ate_rmse(cc, gt)
        out["ate"] = {"gpu_vs_cpu_ref_m": float(np.sqrt(np.mean(np.sum((gc - cc) ** 2, 1)))),
                      "gpu_vs_ground_truth_m": ate_gpu, "cpu_ref_vs_ground_truth_m": ate_cpu,
                      "relative_difference": abs(ate_gpu - ate_cpu) / max(ate_cpu, 1e-12), "frames": len(sel),
                      "poses_bit_identical": bool(np.array_equal(grow[sel, 1:8], cpu_pos[sel])),
                      "note": "stream 0, same frames on both sides: camera-centre RMSE HIP vs CPU restatement (no alignment) and "
                              "Umeyama-aligned ATE of each against the synthetic ground truth; EuRoC MH_05 itself is not available offline"}


if __name__ == "__main__":
    main()
