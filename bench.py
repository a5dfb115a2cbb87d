#!/usr/bin/env python3
"""bench.py -- frames/sec of the MI355X-native FLVIS hot path (front-end tracking + sliding-window BA).

Workload (BASELINE.json configs[3]): one GPU tracks a batch of 64 independent synthetic 640x480 stereo + 200 Hz IMU
streams; a "step" is one stereo frame for every stream of the batch: full HIP front-end (copy, 2x3-level pyramids,
temporal LK, F-RANSAC, PnP-RANSAC, pose LM, reprojection filter, GFTT + FeatureDEM redetect, stereo LK, DLT depth
innovation, keyframe decision) plus the batched Schur BA for every stream that emits a keyframe.  Inputs (rendered
frames, IMU samples) are resident in HBM / host memory before the timed region starts.
With --gpus N (torchrun, one rank per GPU) every rank tracks its own 64 streams (weak scaling, no data-path collective);
the only exchange is one RCCL all-gather of the final poses + all-reduce of counters after the timed region.

Prints ONE JSON line (rank 0).
"""
import argparse
import ctypes as C
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md: 8 TB/s peak, ~6.3 TB/s achievable)

# algorithmic HBM bytes per launch of the image-scan kernels for ONE stream, 640x480:
PYR_BYTES = 307200 + 76800 + 19200 + 4800          # one pyramid (levels 0..3)
ALG_BYTES = {
    "lk_track(temporal)": 2 * PYR_BYTES,           # one pass over the previous and the current pyramid
    "lk_track(stereo)": 2 * PYR_BYTES,             # one pass over the img0 and img1 pyramids
    "gftt:eig_cand": 307200,                       # corner response + maximum + 3x3 local maxima: reads img0 once
    "ingest(copy/equalize)": 4 * 307200,           # read + write both images
    "pyr_down x6": 2 * (307200 + 76800 + 19200) + 2 * (76800 + 19200 + 4800),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=70)
    ap.add_argument("--streams", type=int, default=64, help="independent streams per GPU")
    ap.add_argument("--cpu-frames", type=int, default=80, help="frames of the bounded CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-local-map", action="store_true")
    ap.add_argument("--groups", type=int, default=1,
                    help="split the GPU's streams over this many independent tracker contexts (own HIP streams each): the "
                         "per-stream kernels are latency-bound, so sub-batches fed round-robin overlap on the GPU")
    ap.add_argument("--host-threads", type=int, default=-1,
                    help="1: one host thread per tracker context (ctypes releases the GIL, so the contexts' launch sequences "
                         "are issued in parallel, like one thread per rig group in a deployment); 0: round-robin from one "
                         "thread; -1: on when --groups > 1")
    args = ap.parse_args()

    # the pipeline uses 4-6 HIP streams per tracker context; with the default of 4 hardware queues the long local-map kernels
    # share a queue with the front-end chain (must be set before the HIP runtime initialises)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import flvis_amd
    from flvis_amd import synth

    S, K, Wm = args.streams, args.steps, args.warmup
    ypath = os.path.join(tempfile.gettempdir(), "flvis_bench_d435_stereo_%d.yaml" % rank)
    open(ypath, "w").write(synth.D435I_STEREO_YAML)
    cfg = flvis_amd.load_config(ypath)
    skip = cfg.skip_first_n_imgs
    G = args.groups
    assert S % G == 0, "--streams must be a multiple of --groups"
    Sg = S // G
    XTRA = 20  # untimed frames after the timed region: full per-stage event timing (20 event pairs per frame cost ~10%)
    nsteps = Wm + K + XTRA
    ctxs = [flvis_amd.Context(local_rank, own_stream=(G > 1)) for _ in range(G)]
    trks = [flvis_amd.Tracker(ctxs[g], cfg, Sg, seed_base=0xF1715 + rank * S + g * Sg, traj_capacity=nsteps) for g in range(G)]
    ctx, trk = ctxs[0], trks[0]
    lib = ctx._lib

    # ---- synthetic inputs, resident before the timed region
    from flvis_amd.dist import shard_streams
    stream_ids = shard_streams(rank, world, S)
    trajs = [synth.Trajectory(s) for s in stream_ids]
    rnd = synth.Renderer(dev)
    t_render = time.time()
    frames = {}
    for f in range(skip, nsteps):  # the first `skip` frames are dropped by the reference before any processing
        frames[f] = rnd.stereo_frame(trajs, f / synth.FRAME_HZ, f)
    for f in range(0, min(skip, nsteps)):
        frames[f] = frames.get(skip, None) or rnd.stereo_frame(trajs, f / synth.FRAME_HZ, f)
    torch.cuda.synchronize()
    t_render = time.time() - t_render
    SPF = 16
    imu = np.zeros((nsteps, S, SPF, 7))
    imu_cnt = np.zeros((nsteps, S), np.int32)
    for i, s in enumerate(stream_ids):
        t_prev = -1.0 / synth.FRAME_HZ
        for f in range(nsteps):
            t = f / synth.FRAME_HZ
            smp = synth.imu_samples(trajs[i], s, t_prev, t)
            imu[f, i, :len(smp)] = smp
            imu_cnt[f, i] = len(smp)
            t_prev = t
    times = np.array([[f / synth.FRAME_HZ] * S for f in range(nsteps)])
    wlm = 0 if args.no_local_map else 1

    def step_group(f, g):
        i0, i1 = frames[f]
        a, b = g * Sg, (g + 1) * Sg
        rc = lib.flvis_imu_feed_all(ctxs[g]._h, imu_cnt[f, a:b].ctypes.data_as(C.POINTER(C.c_int)),
                                    imu[f, a:b].ctypes.data_as(C.POINTER(C.c_double)), SPF)
        if rc:
            ctxs[g]._check(rc, "imu_feed_all")
        rc = lib.flvis_image_feed(ctxs[g]._h, C.c_void_p(i0[a:b].data_ptr()), C.c_void_p(i1[a:b].data_ptr()),
                                  times[f, a:b].ctypes.data_as(C.POINTER(C.c_double)), C.c_void_p(0), wlm)
        if rc:
            ctxs[g]._check(rc, "image_feed")

    def step(f):
        for g in range(G):
            step_group(f, g)

    threaded = G > 1 and args.host_threads != 0

    def run_frames(f0, f1):
        """feed frames [f0, f1) to every context: from one thread round-robin, or one host thread per context"""
        if not threaded:
            for f in range(f0, f1):
                step(f)
            return
        import threading
        errs = []

        def worker(g):
            try:
                torch.cuda.set_device(local_rank)
                for f in range(f0, f1):
                    step_group(f, g)
            except Exception as e:  # surfaced after join
                errs.append(e)
        ths = [threading.Thread(target=worker, args=(g,)) for g in range(G)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        if errs:
            raise errs[0]

    def barrier():
        if world > 1:
            dist.barrier()

    run_frames(0, Wm)
    torch.cuda.synchronize()
    nst = lib.flvis_prof_stage_count()
    lib.flvis_prof_stage_name.restype = C.c_char_p
    lib.flvis_prof_enable_stages.argtypes = [C.c_void_p, C.c_int, C.c_uint64]
    names = [lib.flvis_prof_stage_name(i).decode() for i in range(nst)]
    lk_mask = sum(1 << i for i, n in enumerate(names) if n.startswith("lk_track"))

    def read_stages(c):
        ms = (C.c_double * nst)()
        nrec = C.c_int(0)
        c._check(lib.flvis_prof_read(c._h, ms, C.byref(nrec)), "prof_read")
        return {names[i]: ms[i] / max(nrec.value, 1) for i in range(nst)}

    # timed region: only the dominant kernel (k_lk_track, 2 launches per frame) is bracketed by HIP events
    for g in range(G):
        ctxs[g]._check(lib.flvis_prof_enable_stages(ctxs[g]._h, K, C.c_uint64(lk_mask)), "prof_enable")
    barrier()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    run_frames(Wm, Wm + K)
    e1.record()
    for g in range(G):  # everything enqueued AND every queued keyframe consumed by the local map
        ctxs[g]._check(lib.flvis_hip_synchronize(ctxs[g]._h), "synchronize")
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    gpu_ms = e0.elapsed_time(e1)
    from flvis_amd import dist as fdist
    elapsed = fdist.max_over_ranks(elapsed, dev)
    lk_stages = read_stages(ctx)
    # untimed epilogue: all stages
    for g in range(G):
        ctxs[g]._check(lib.flvis_prof_enable_stages(ctxs[g]._h, XTRA, C.c_uint64((1 << 64) - 1)), "prof_enable")
    run_frames(Wm + K, Wm + K + XTRA)
    torch.cuda.synchronize()
    stages = read_stages(ctx)

    # ---- results: tracker health, final poses
    cnt = [sum(c) for c in zip(*[t.counters() for t in trks])]
    rows = np.stack([trks[i // Sg].trajectory(i % Sg, Wm + K + XTRA - 1, 1)[0] for i in range(S)])
    tracking = int((rows[:, 8].astype(int) & 15 == 1).sum())
    kfs_total = cnt[1]
    # the path's only exchange: results, after the timed region: all-gather poses, all-reduce counters
    all_poses, csum = fdist.exchange_results(torch.from_numpy(rows[:, 1:8].copy()).to(dev),
                                             [cnt[0], cnt[1], cnt[2], tracking], dev)
    assert all_poses.shape[0] == world * S
    cnt, tracking, kfs_total = csum[:3], csum[3], csum[1]

    out = None
    if rank == 0:
        total_frames = world * S * K
        value = total_frames / elapsed
        # dominant kernel ON THE CRITICAL PATH: k_lk_track (two launches per step: temporal + stereo).  The kernel with the
        # largest total time is k_ba_worker, but it runs beside the front-end on the local-map streams and is latency-bound
        # fp64 with ~64 KB of algorithmic traffic per keyframe
        lk_ms = [lk_stages["lk_track(temporal)"], lk_stages["lk_track(stereo)"]]
        dom = "k_lk_track"
        dom_ms = sum(lk_ms) / 2.0
        dom_bytes = ALG_BYTES["lk_track(temporal)"] * Sg
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        # HBM traffic of the same kernel from the committed PMC passes (profiles/r01_lk_pmc.json, scripts/pmc_to_json.py);
        # PMC collection needs rocprofv3 around the process, so it cannot be measured from inside this script
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_lk_pmc.json")))
            if S == 64 and pmc.get("traffic_bytes_per_launch"):
                traffic = int(pmc["traffic_bytes_per_launch"])
        except Exception:
            traffic = None
        out = {
            "metric": "frames/sec/node (640\u00d7480 stereo+IMU) + ATE vs CPU ref, EuRoC MH_05",  # BASELINE.json's metric, verbatim
            "metric_note": "synthetic 640x480 stereo+IMU streams (no dataset offline): ATE is reported against the CPU reference "
                           "and the synthetic ground truth, not on EuRoC MH_05",
            "value": round(value, 1), "unit": "frames/s",
            "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": round(elapsed / K * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/f32 LK+GFTT, f64 geometry+BA",
            "data": "synthetic",
            "config": {"workload": "1xMI355X: batch of %d independent 640x480 synthetic stereo+IMU streams, full HIP "
                                   "front-end + batched Schur BA (BASELINE.json configs[3])" % S,
                       "streams_per_gpu": S, "contexts_per_gpu": G, "host_threads": (G if threaded else 1), "window_size": cfg.window_size, "local_map": bool(wlm),
                       "streams_tracking_at_end": tracking, "keyframes_in_run": int(kfs_total),
                       "ba_runs_in_run": int(cnt[2]), "gpu_ms_per_step_events": round(gpu_ms / K, 4)},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "avg_launch_ms": round(dom_ms, 4), "algorithmic_bytes_per_launch": dom_bytes,
                         "launches_per_step": 2},
            "stages_ms_per_step": {k: round(v, 4) for k, v in stages.items()},
            "stages_note": "per-stage times from %d untimed frames after the timed region (all stages bracketed by events)" % XTRA,
        }
        # ---- CPU baseline: the oracle (port of the reference path) on a bounded sample of the same workload, 1 core
        if args.cpu_frames > 0:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import _oracle as O
            ocfg = O.RefConfig()
            C.memmove(C.byref(ocfg), C.byref(cfg), C.sizeof(cfg))
            ref = O.Tracker(ocfg, 0xF1715)
            lm = O.LocalMap(cfg.window_size, np.array([cfg.P0[0], cfg.P0[5], cfg.P0[2], cfg.P0[6]]))
            n_cpu = min(args.cpu_frames, nsteps - skip)
            host = [(frames[skip + j][0][0].cpu().numpy(), frames[skip + j][1][0].cpu().numpy()) for j in range(n_cpu)]
            for f in range(skip):  # IMU-only prefix (untimed): the skipped frames carry no vision work
                for r in imu[f, 0, :imu_cnt[f, 0]]:
                    ref.imu(r[0], r[1:4], r[4:7])
                ref.image(f / synth.FRAME_HZ, host[0][0], host[0][1])
            tc = time.perf_counter()
            cpu_pos, cpu_state = [], []
            for j in range(n_cpu):
                f = skip + j
                for r in imu[f, 0, :imu_cnt[f, 0]]:
                    ref.imu(r[0], r[1:4], r[4:7])
                res = ref.image(f / synth.FRAME_HZ, host[j][0], host[j][1])
                cpu_pos.append(np.asarray(res["pose7"], float))
                cpu_state.append(res["state"])
                if res["new_keyframe"] and wlm:
                    kf = ref.keyframe()
                    lm.push(kf["frame_id"], kf["pose7"], kf["lm_id"], kf["lm_2d"], kf["lm_3d"])
            tc = time.perf_counter() - tc
            out["cpu_baseline"] = {"value": round(n_cpu / tc, 2), "unit": "frames/s", "cores": 1, "kind": "port",
                                   "sample": "stream 0, %d frames after the %d skipped start-up frames of the same synthetic "
                                             "workload, oracle front-end + local-map BA, single thread (%d host cores "
                                             "present)" % (n_cpu, skip, os.cpu_count())}
            # ATE of the GPU trajectory of stream 0 against the CPU reference on the same frames (camera centres, no
            # alignment: both run from the same initial state), and both against the synthetic ground truth
            def centre(p7):
                q = p7[3:7] / np.linalg.norm(p7[3:7])
                x, y, z, w = q
                R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                              [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                              [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
                return -R.T @ p7[0:3]
            grow = trks[0].trajectory(0, skip, n_cpu)
            sel = [j for j in range(n_cpu) if cpu_state[j] == 1 and (int(grow[j, 8]) & 15) == 1]
            if len(sel) >= 3:
                from flvis_amd import traj_io
                gc = np.array([centre(grow[j, 1:8]) for j in sel])
                cc = np.array([centre(cpu_pos[j]) for j in sel])
                gt = np.array([-(trajs[0].T_c_w((skip + j) / synth.FRAME_HZ)[0]).T @ trajs[0].T_c_w((skip + j) / synth.FRAME_HZ)[1]
                               for j in sel])
                ate_gpu, ate_cpu = traj_io.ate_rmse(gc, gt), traj_io.ate_rmse(cc, gt)
                out["ate"] = {"gpu_vs_cpu_ref_m": float(np.sqrt(np.mean(np.sum((gc - cc) ** 2, 1)))),
                              "gpu_vs_ground_truth_m": ate_gpu, "cpu_ref_vs_ground_truth_m": ate_cpu,
                              "relative_difference": abs(ate_gpu - ate_cpu) / max(ate_cpu, 1e-12),
                              "frames": len(sel),
                              "note": "stream 0, same frames on both sides: camera-centre RMSE HIP vs CPU restatement (no alignment) "
                                      "and Umeyama-aligned ATE of each against the synthetic ground truth; EuRoC MH_05 "
                                      "itself is not available offline"}
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    for c in ctxs:
        c.close()


if __name__ == "__main__":
    main()
