#!/usr/bin/env python3
"""Keyframe-rate work of the loop closing on one GPU, stage by stage, for a batch of keyframes (the streams one GPU carries):
ORB -> bag of words (a 10^L-word random vocabulary of DBoW3's shape: k = 10) -> 3-D landmarks (stereo LK + DLT) -> one row of the
similarity matrix against a database of earlier keyframes -> descriptor matching -> PnP-RANSAC.  Rendered 640x480 stereo pairs of
the synthetic D435i rig.  Prints one JSON line (times in ms per batch); run under rocprofv3 --kernel-trace for per-kernel times.

usage: loopkf_bench.py [n_img=64] [iters=10] [L=5] [n_db=2000]"""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import flvis_amd
from flvis_amd import synth

n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 64
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
L = int(sys.argv[3]) if len(sys.argv) > 3 else 5
n_db = int(sys.argv[4]) if len(sys.argv) > 4 else 2000
K = 10
CAP = 1024

ctx = flvis_amd.Context(0)
p = os.path.join(tempfile.gettempdir(), "flvis_loopkf_bench.yaml")
open(p, "w").write(synth.D435I_STEREO_YAML)
cfg = flvis_amd.load_config(p)
P0, P1 = np.array(list(cfg.P0)), np.array(list(cfg.P1))
K4 = np.array([cfg.P0[0], cfg.P0[5], cfg.P0[2], cfg.P0[6]])

# complete k-ary tree, nodes in breadth-first order; random 256-bit node descriptors, random idf weights on the leaves
n_nodes = (K ** (L + 1) - 1) // (K - 1)
n_inner = (K ** L - 1) // (K - 1)
rng = np.random.default_rng(7)
child_ptr = np.zeros(n_nodes + 1, np.int32)
child_ptr[1:n_inner + 1] = K * np.arange(1, n_inner + 1)
child_ptr[n_inner + 1:] = K * n_inner
child_idx = np.arange(1, n_nodes, dtype=np.int32)
desc_v = rng.integers(0, 256, (n_nodes, 32), dtype=np.uint8)
weight = np.zeros(n_nodes)
weight[n_inner:] = rng.uniform(0.5, 8.0, n_nodes - n_inner)
word_id = np.full(n_nodes, -1, np.int32)
word_id[n_inner:] = np.arange(n_nodes - n_inner)
ctx.bow_set_vocabulary(child_ptr, child_idx, desc_v, weight, word_id)

trs = [synth.Trajectory(s) for s in range(n_img)]
rnd = synth.Renderer("cuda")
a0, a1 = rnd.stereo_frame(trs, 1.0, 20)
b0, b1 = rnd.stereo_frame(trs, 1.4, 28)


def timed(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters, out


t_orb, (kps, desc, cnt, ovf) = timed(lambda: ctx.orb_detect_and_compute(a0, cap=CAP))
kb, db, cb, _ = ctx.orb_detect_and_compute(b0, cap=CAP)
t_bow, (ids, vals, nnz) = timed(lambda: ctx.bow_transform(desc, cnt, vcap=CAP))
t_lm, (lm2, lm3, lmd, lmc) = timed(lambda: ctx.lc_keyframe_landmarks(a0, a1, 0, kps, desc, cnt, P0=P0, P1=P1))
m2, m3, md, mc = ctx.lc_keyframe_landmarks(b0, b1, 0, kb, db, cb, P0=P0, P1=P1)
# database of n_db earlier keyframes: the batch's own vectors repeated
rep = (n_db + n_img - 1) // n_img
db_ids, db_vals, db_nnz = [t.repeat((rep,) + (1,) * (t.dim() - 1))[:n_db].contiguous() for t in (ids, vals, nnz)]
t_row, scores = timed(lambda: ctx.bow_score(ids[0], vals[0], nnz[0:1], db_ids, db_vals, db_nnz))
t_match, (pairs, npairs) = timed(lambda: ctx.orb_match(lmd, lmc, md, mc, 0.8))
# gather (3-D of the earlier keyframe, pixel of the later one) per pair of keyframes, then PnP-RANSAC for the whole batch
m = npairs.clamp(max=CAP)
qi, ti = pairs[:, :, 0].long().clamp(min=0), pairs[:, :, 1].long().clamp(min=0)
d3 = torch.gather(lm3, 1, qi[:, :, None].expand(-1, -1, 3)).float().contiguous()
d2 = torch.gather(m2, 1, ti[:, :, None].expand(-1, -1, 2)).contiguous()
seeds = list(range(1, n_img + 1))
t_pnp, (pose, mask, ninl) = timed(lambda: ctx.pnp_ransac(d3, d2, m.int(), K4, seeds))
print(json.dumps({
    "n_img": n_img, "vocabulary_words": int(n_nodes - n_inner), "database_keyframes": n_db,
    "orb_ms": t_orb, "bow_transform_ms": t_bow, "landmarks_ms": t_lm, "similarity_row_ms": t_row, "match_ms": t_match,
    "pnp_ransac_ms": t_pnp, "per_keyframe_total_ms": (t_orb + t_bow + t_lm + t_match + t_pnp) / n_img + t_row,
    "mean_keypoints": float(cnt.float().mean()), "mean_landmarks": float(lmc.float().mean()),
    "mean_bow_entries": float(nnz.float().mean()), "mean_matches": float(npairs.float().mean()),
    "mean_pnp_inliers": float(ninl.float().mean())}))
