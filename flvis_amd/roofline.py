"""Per-kernel roofline table of bench.py as pure functions (no torch, no GPU), so that the arithmetic is tested on CPU.

Every kernel that takes more than 5 % of a frame gets one line, priced against the resource that bounds it:

  image-scan kernels (pyramids, corner response)  algorithmic bytes / time against the box's MEASURED copy bandwidth and the 8 TB/s peak
  k_lk_track, k_eig_walk                          VALU issue: wave-instructions per launch (SQ_INSTS_VALU, collected by `bench.py --pmc`)
                                                  / time against 1024 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction = 614.4 G/s
  k_ba_worker                                     fp64 flops by SURVEY 8d's formula from the kernel's own trial / edge / landmark counters
                                                  against the fp64 vector peak (78.6 TFLOP/s chip, 307 GFLOP/s per CU: a window is one workgroup)
  one-workgroup-per-stream chain kernels          latency-bound by construction (64 workgroups of dependent fp64 code): time only

Times are HIP-event stage times measured live by bench.py (epilogue frames); `rocprof_avg_launch_ms` is the average of the same kernel in
the rocprofv3 kernel trace that `bench.py --pmc` stores next to the counters -- the two must agree (the stage of a kernel that runs under
another kernel is longer than its trace time; the table says which)."""
import os

HBM_PEAK_GBS = 8000.0
SIMDS = 1024
CLOCK_GHZ = 2.4
VALU_ISSUE_PEAK_GINST = SIMDS * CLOCK_GHZ / 4.0      # wave64 instructions per second (G), one per 4 cycles and SIMD
FP64_PEAK_TFLOPS = 78.6
CUS = 256


def pyramid_bytes(w, h, levels=3):
    """bytes read + written by building levels 1..levels of one image's pyramid (level 0 is the input)"""
    tot, cw, ch = 0, w, h
    for _ in range(levels):
        nw, nh = (cw + 1) // 2, (ch + 1) // 2
        tot += cw * ch + nw * nh
        cw, ch = nw, nh
    return tot


def pyramid_total_bytes(w, h, levels=3):
    """bytes of levels 0..levels of one pyramid (what one k_lk_track launch may touch per image)"""
    tot, cw, ch = 0, w, h
    for _ in range(levels + 1):
        tot += cw * ch
        cw, ch = (cw + 1) // 2, (ch + 1) // 2
    return tot


def ba_flops(trials, trials_items, trials_landmarks, trials_poses):
    """SURVEY 8d per LM trial: E (120 + 300) + sum_l k_l^2 / 2 * 324 + (6 P)^3 / 3, summed over the optimisations of a run from the
    kernel's counters (sum of trials, of trials x observations, of trials x landmarks, of trials x free poses).  sum_l k_l^2 is taken at
    its lower bound E^2 / L (all landmarks equally often observed)."""
    if trials <= 0:
        return 0.0
    E = trials_items / trials
    L = max(trials_landmarks / trials, 1.0)
    P = trials_poses / trials
    per_trial = E * 420.0 + (E * E / L) / 2.0 * 324.0 + (6.0 * P) ** 3 / 3.0
    return per_trial * trials


def _ms(stages, *names):
    return sum(float(stages.get(n, 0.0)) for n in names)


def kernel_table(stages, S, w, h, copy_gbs, pmc=None, ba=None, ms_per_step=None):
    """stages: stage name -> ms per step (HIP events); pmc: {"kernels": {name: {"valu_insts": .., "fetch_kb": .., "write_kb": ..,
    "avg_ns": ..}}} or None; ba: {"trials", "trials_items", "trials_landmarks", "trials_poses", "runs", "worker_launches",
    "worker_ms_per_launch"} or None.  Returns the list of per-kernel lines."""
    pk = (pmc or {}).get("kernels", {})
    img = w * h
    rows = []

    def counters(name, row, launches=1):
        k = pk.get(name)
        if not k:
            return
        if k.get("valu_insts") is not None:
            row["valu_insts_per_launch"] = int(k["valu_insts"])
            if row.get("avg_launch_ms"):
                rate = k["valu_insts"] / (row["avg_launch_ms"] * 1e-3) / 1e9
                row["valu_ginst_per_s"] = round(rate, 1)
                row["valu_issue_frac"] = round(rate / VALU_ISSUE_PEAK_GINST, 4)
                other = [k.get(f) for f in ("salu_insts", "lds_insts", "smem_insts", "vmem_rd_insts", "vmem_wr_insts")]
                if any(o is not None for o in other):   # (round 6) every counted instruction class against the same ceiling
                    tot = k["valu_insts"] + sum(o for o in other if o is not None)
                    row["insts_per_launch_counted_classes"] = int(tot)
                    row["salu_insts_per_launch"] = int(k["salu_insts"]) if k.get("salu_insts") is not None else None
                    row["issue_frac_counted_classes"] = round(tot / (row["avg_launch_ms"] * 1e-3) / 1e9 / VALU_ISSUE_PEAK_GINST, 4)
        if k.get("fetch_kb_calibrated") is not None and k.get("write_kb_calibrated") is not None:   # counter x probe factor (round 4)
            row["traffic_bytes_per_launch"] = int((k["fetch_kb_calibrated"] + k["write_kb_calibrated"]) * 1024)
            row["traffic_calibrated"] = True
        elif k.get("fetch_kb") is not None and k.get("write_kb") is not None:
            row["traffic_bytes_per_launch"] = int((k["fetch_kb"] + k["write_kb"]) * 1024)
            row["traffic_calibrated"] = False
        if k.get("avg_ns") is not None:
            row["rocprof_avg_launch_ms"] = round(k["avg_ns"] * 1e-6, 4)

    def scan(name, stage_ms, bytes_per_launch, launches, note):
        row = {"kernel": name, "bound": "hbm", "launches_per_step": launches, "avg_launch_ms": round(stage_ms / launches, 4),
               "algorithmic_bytes_per_launch": int(bytes_per_launch), "note": note}
        if stage_ms > 0:
            gbs = bytes_per_launch * launches / (stage_ms * 1e-3) / 1e9
            row["achieved_GBs"] = round(gbs, 1)
            row["frac_of_hbm_peak"] = round(gbs / HBM_PEAK_GBS, 4)
            if copy_gbs:
                row["frac_of_measured_copy"] = round(gbs / copy_gbs, 4)
        return row

    # k_lk_track: two launches per step
    lk_t, lk_s = _ms(stages, "lk_track(temporal)"), _ms(stages, "lk_track(stereo)")
    for tag, ms in (("temporal", lk_t), ("stereo", lk_s)):
        if ms <= 0:
            continue
        b = 2 * pyramid_total_bytes(w, h) * S
        row = {"kernel": "k_lk_track (%s)" % tag, "bound": "valu_issue", "launches_per_step": 1, "avg_launch_ms": round(ms, 4),
               "algorithmic_bytes_per_launch": int(b), "achieved_GBs": round(b / (ms * 1e-3) / 1e9, 1),
               "frac_of_hbm_peak": round(b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
               "note": "one wave per point; both launches share the chip with the local-map workers only"}
        counters("k_lk_track_%s" % tag if ("k_lk_track_%s" % tag) in pk else "k_lk_track", row)   # (instances k_lk_track<1> / <2> since round 4)
        rows.append(row)
    # corner response
    e = _ms(stages, "gftt:eig_cand")
    if e > 0:
        row = scan("k_eig_walk", e, img * S, 1, "corner response, one pass over the left image; runs beside the PnP RANSAC / pose LM chain (detection stream)")
        row["bound"] = "valu_issue"
        counters("k_eig_walk", row)
        rows.append(row)
    p = _ms(stages, "gftt:pick")
    if p > 0:
        row = {"kernel": "k_gftt_pick", "bound": "latency (one workgroup per stream)", "launches_per_step": 1, "avg_launch_ms": round(p, 4)}
        counters("k_gftt_pick", row)
        rows.append(row)
    # left pyramid: the ingest launch (level 0 copy + level 1) and the launch that makes levels 2 and 3, in one stage
    pl = _ms(stages, "pyr_down(left)")
    if pl > 0:
        b = (img + pyramid_bytes(w, h)) * S  # the ingest also writes level 0
        walk = os.environ.get("FLVIS_PYR_TILES", "0") in ("", "0")   # (the A/B knob that selects the LDS-tile kernels)
        name = "k_pyr_walk<1,true> + k_pyr_walk<2,false> (left pyramid)" if walk else "k_pyr_down_ingest + 2 x k_pyr_down (left pyramid)"
        row = scan(name, pl, b, 1, "both launches in one stage (HIP events on the detection stream, beside k_frame_head and the local map's workgroups): "
                   "level 0 copy + levels 1..3; the borders of the levels (not counted) are written as well")
        if walk and pk:
            # the two launches' own durations and counters (rocprofv3 passes): what the kernels do when the stage's waits are left out
            a, c = pk.get("k_pyr_walk<1,true>") or {}, pk.get("k_pyr_walk<2,false>") or {}
            if a.get("avg_ns") and c.get("avg_ns"):
                ms = (a["avg_ns"] + c["avg_ns"]) * 1e-6
                row["rocprof_launches_ms"] = [round(a["avg_ns"] * 1e-6, 4), round(c["avg_ns"] * 1e-6, 4)]
                row["achieved_GBs_kernels_only"] = round(b / (ms * 1e-3) / 1e9, 1)
            for key in ("valu_insts", "fetch_kb", "write_kb", "fetch_kb_calibrated", "write_kb_calibrated"):
                if a.get(key) is not None and c.get(key) is not None:
                    row.setdefault("_sum", {})[key] = a[key] + c[key]
            sm = row.pop("_sum", {})
            if "valu_insts" in sm:
                row["valu_insts_per_launch"] = int(sm["valu_insts"])
            if "fetch_kb_calibrated" in sm and "write_kb_calibrated" in sm:
                row["traffic_bytes_per_launch"] = int((sm["fetch_kb_calibrated"] + sm["write_kb_calibrated"]) * 1024)
                row["traffic_calibrated"] = True
            elif "fetch_kb" in sm and "write_kb" in sm:
                row["traffic_bytes_per_launch"] = int((sm["fetch_kb"] + sm["write_kb"]) * 1024)
                row["traffic_calibrated"] = False
        else:
            counters("k_pyr_down_ingest", row)
        rows.append(row)
    # the one-workgroup-per-stream chain
    chain = [("k_frame_head_prepare" if os.environ.get("FLVIS_HEAD_PREPARE", "1") != "0" else "k_frame_head", ("imu_feed+frame_begin",)), ("k_ransac_f", ("ransac_f",)), ("k_ransac_pnp", ("ransac_pnp",)),
             ("k_pose_lm", ("track_post+pose_lm",)), ("k_reproj_filter", ("reproj_filter",)),
             ("k_feature_dem + k_add_new", ("feature_dem+add_new",)), ("k_depth_seeds", ("depth_prepare",)),
             ("k_depth_innovate", ("depth_innovate",)), ("k_frame_end", ("frame_end",))]
    for name, st in chain:
        ms = _ms(stages, *st)
        if ms <= 0:
            continue
        row = {"kernel": name, "bound": "latency (one workgroup per stream: %d of %d CUs)" % (S, CUS), "launches_per_step": 1,
               "avg_launch_ms": round(ms, 4)}
        counters(name.split(" ")[0], row)
        rows.append(row)
    # local map
    if ba and ba.get("runs"):
        fl = ba_flops(ba["trials"], ba["trials_items"], ba["trials_landmarks"], ba["trials_poses"])
        row = {"kernel": "k_ba_worker", "bound": "fp64 (one workgroup per window)", "optimisations": int(ba["runs"]),
               "lm_trials_per_optimisation": round(ba["trials"] / ba["runs"], 2),
               "observations_per_window": round(ba["trials_items"] / max(ba["trials"], 1), 1),
               "landmarks_per_window": round(ba["trials_landmarks"] / max(ba["trials"], 1), 1),
               "free_poses_per_window": round(ba["trials_poses"] / max(ba["trials"], 1), 2),
               "mflop_per_optimisation": round(fl / ba["runs"] / 1e6, 2)}
        if ba.get("ms_per_optimisation"):
            g = fl / ba["runs"] / (ba["ms_per_optimisation"] * 1e-3) / 1e9
            row["ms_per_optimisation"] = round(ba["ms_per_optimisation"], 4)
            if ba.get("chunked_runs") is not None:   # windows too large for resident records (streamed in chunks), and what they cost
                row["chunked_optimisations"] = ba["chunked_runs"]
                row["ms_per_chunked_optimisation"] = round(ba["ms_per_chunked_optimisation"], 4) if ba.get("ms_per_chunked_optimisation") else None
            row["gflops_per_workgroup"] = round(g, 2)
            row["frac_of_cu_fp64_peak"] = round(g / (FP64_PEAK_TFLOPS * 1e3 / CUS), 4)
        if ba.get("worker_ms_per_launch"):
            row["avg_launch_ms"] = round(ba["worker_ms_per_launch"], 4)
        counters("k_ba_worker", row)
        rows.append(row)
    if ms_per_step:
        for r in rows:
            if r.get("avg_launch_ms") is not None and not r["kernel"].startswith("k_ba_worker"):   # (the local map runs beside the frames)
                r["share_of_step"] = round(r["avg_launch_ms"] * r.get("launches_per_step", 1) / ms_per_step, 3)
    return rows
