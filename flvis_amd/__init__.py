"""flvis_amd -- MI355X-native (gfx950) FLVIS front-end tracking + local-map BA hot path.

Thin ctypes binding over the C ABI in include/flvis_hip.h (libflvis_hip.so, hand-written HIP kernels).  PyTorch is
used only as plumbing (device buffers, streams).  There is NO CPU fallback: without a HIP device every call raises.
"""
import ctypes as C
import os

from . import build as _build

_LIB = None

FLVIS_OK = 0
FLVIS_ERR_NO_DEVICE = -2


class FlvisError(RuntimeError):
    pass


def lib_path():
    return _build.LIB


def load_library(rebuild_if_stale=True):
    """Loads libflvis_hip.so (building it in-tree with hipcc when sources are newer). Raises if unavailable."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # torch bundles its own libamdhip64 (same SONAME as /opt/rocm's).  Import it FIRST so that libflvis_hip.so binds
    # to the HIP runtime torch already loaded; two runtimes in one process cannot both own the device.
    import torch  # noqa: F401
    if rebuild_if_stale and os.path.exists(_build.HIPCC):
        try:
            if _build.stale():
                _build.build()
        except Exception as e:  # stale-but-present library is still usable; a missing one is fatal below
            if not os.path.exists(_build.LIB):
                raise FlvisError("cannot build libflvis_hip.so: %s" % e)
    if not os.path.exists(_build.LIB):
        raise FlvisError("libflvis_hip.so is missing (run python -c 'import __graft_entry__ as g; g.build()')")
    # FLVIS_LIB_PATH: load another build of the same library (A/B runs of a kernel variant inside one benchmark session)
    _LIB = C.CDLL(os.environ.get("FLVIS_LIB_PATH") or _build.LIB)
    _LIB.flvis_version.restype = C.c_char_p
    _LIB.flvis_last_error.restype = C.c_char_p
    _LIB.flvis_last_error.argtypes = [C.c_void_p]
    _LIB.flvis_hip_stream.restype = C.c_void_p
    _LIB.flvis_hip_stream.argtypes = [C.c_void_p]
    return _LIB


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class Context:
    """Owns a flvis_ctx bound to one GPU and one HIP stream: torch's current stream by default, the default stream with
    use_torch_stream=False, or a private non-blocking stream with own_stream=True (several contexts then run concurrently;
    inputs produced on torch's stream must be synchronised by the caller)."""

    def __init__(self, device=0, use_torch_stream=True, own_stream=False):
        import torch
        self._lib = load_library()
        if not torch.cuda.is_available():
            raise FlvisError("flvis_amd needs a HIP device (MI355X); none is visible. No CPU fallback exists.")
        torch.cuda.set_device(device)
        self.device = torch.device("cuda", device)
        stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream) if use_torch_stream else C.c_void_p(0)
        if own_stream:
            stream = C.c_void_p(-1 & (2 ** 64 - 1))  # FLVIS_STREAM_NEW
        h = C.c_void_p(0)
        rc = self._lib.flvis_hip_create(C.c_int(device), stream, C.byref(h))
        if rc != FLVIS_OK:
            raise FlvisError("flvis_hip_create failed: %d" % rc)
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.flvis_hip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != FLVIS_OK:
            raise FlvisError("%s failed (%d): %s" % (what, rc, self._lib.flvis_last_error(self._h).decode()))

    def synchronize(self):
        self._check(self._lib.flvis_hip_synchronize(self._h), "synchronize")

    # ---- kernel-level entry points (torch CUDA tensors in/out) -------------------------------------------------
    def equalize_hist(self, img):
        """img: uint8 [n,h,w] cuda tensor -> equalised copy."""
        import torch
        assert img.dtype == torch.uint8 and img.is_cuda and img.is_contiguous() and img.dim() == 3
        n, h, w = img.shape
        out = torch.empty_like(img)
        self._check(self._lib.flvis_hip_equalize_hist(self._h, _ptr(img), _ptr(out), w, h, n), "equalize_hist")
        return out

    def cvt_bgr_to_gray(self, img):
        """img uint8 [n,h,w,3|4] (BGR / BGRA, interleaved) -> gray [n,h,w]."""
        import torch
        assert img.dtype == torch.uint8 and img.is_cuda and img.is_contiguous() and img.dim() == 4
        n, h, w, c = img.shape
        out = torch.empty((n, h, w), dtype=torch.uint8, device=img.device)
        self._check(self._lib.flvis_hip_cvt_bgr_to_gray(self._h, _ptr(img), c, _ptr(out), w, h, n), "cvt_bgr_to_gray")
        return out

    def pyr_down(self, img):
        import torch
        assert img.dtype == torch.uint8 and img.is_cuda and img.is_contiguous() and img.dim() == 3
        n, h, w = img.shape
        dw, dh = (w + 1) // 2, (h + 1) // 2
        out = torch.empty((n, dh, dw), dtype=torch.uint8, device=img.device)
        self._check(self._lib.flvis_hip_pyr_down(self._h, _ptr(img), w, h, w, _ptr(out), dw, n), "pyr_down")
        return out

    def debug_pyramid(self, img, levels, bx=32, by=24, ingest=True):
        """Test aid (flvis_debug_pyramid): the tracker's pyramid construction.  img uint8 [n,h,w] on the GPU; returns the list of the
        levels 0 .. levels as uint8 [n, h_l + 2 by, w_l + 2 bx] (border included; level 0 is None when ingest is False)."""
        import torch
        assert img.dtype == torch.uint8 and img.is_cuda and img.is_contiguous() and img.dim() == 3
        n, h, w = img.shape
        geo, off = [], 0
        lw, lh = w, h
        for _ in range(levels + 1):
            pitch = ((lw + 15) & ~15) + 2 * bx
            rows = lh + 2 * by
            geo.append((off, pitch, rows, lw, lh))
            off += (pitch * rows * n + 63) & ~63
            lw, lh = (lw + 1) // 2, (lh + 1) // 2
        out = torch.zeros(off + 64, dtype=torch.uint8, device=img.device)
        pad = (-out.data_ptr()) % 64
        buf = out[pad:pad + off]
        self._check(self._lib.flvis_debug_pyramid(self._h, _ptr(img), w, h, n, levels, bx, by, 1 if ingest else 0, _ptr(buf), C.c_size_t(off)), "debug_pyramid")
        res = []
        for l, (o, pitch, rows, lw, lh) in enumerate(geo):
            if l == 0 and not ingest:
                res.append(None)
                continue
            res.append(buf[o:o + pitch * rows * n].view(n, rows, pitch)[:, :, :lw + 2 * bx].clone())
        return res

    def lk_track(self, prev, nxt, prev_pts, next_pts, count, max_level=10, max_iter=30, eps=1e-3, use_initial=True):
        """prev/nxt uint8 [n,h,w]; prev_pts/next_pts float32 [n,nmax,2]; count int32 [n].
        Returns (next_pts_out, status uint8 [n,nmax])."""
        import torch
        prev, nxt = prev.contiguous(), nxt.contiguous()
        n, h, w = prev.shape
        nmax = prev_pts.shape[1]
        assert prev_pts.dtype == torch.float32 and next_pts.dtype == torch.float32 and count.dtype == torch.int32
        out = next_pts.clone().contiguous()
        status = torch.zeros((n, nmax), dtype=torch.uint8, device=prev.device)
        self._check(self._lib.flvis_hip_lk_track(self._h, _ptr(prev), _ptr(nxt), w, h, n, _ptr(prev_pts.contiguous()),
                                                 _ptr(out), _ptr(status), _ptr(count), nmax, max_level, max_iter,
                                                 C.c_double(eps), int(use_initial)), "lk_track")
        return out, status

    def rand_seed(self, seed, n_sets):
        """flvis_hip_rand_seed: the glibc rand() state of n_sets sets after srand(seed) (int32 [n_sets, 35] on the device)."""
        import torch
        st = torch.zeros((n_sets, 35), dtype=torch.int32, device=self.device)
        self._check(self._lib.flvis_hip_rand_seed(self._h, C.c_uint32(seed), _ptr(st), n_sets), "rand_seed")
        return st

    def stereo_depth(self, cfg, img0, img1, pt2d_plane, pt2d_undistort, pt3d_w, has_depth, count, poses7, rng, rand_state):
        """flvis_hip_stereo_depth = CameraFrame::recover3DPts_c_FromStereo (camera_frame.cpp:93-180) for n_sets frames in one call.
        img0 / img1 uint8 [n,h,w]; pt2d_* float32 [n,cap,2]; pt3d_w float32 [n,cap,3]; has_depth uint8 [n,cap]; count int32 [n] (all on
        the device); poses7 host [n,7]; rand_state from rand_seed() (updated in place).  Returns (pt3ds float64 [n,cap,3], mask uint8)."""
        import numpy as np
        import torch
        n, cap = pt2d_plane.shape[0], pt2d_plane.shape[1]
        for a, dt in ((pt2d_plane, torch.float32), (pt2d_undistort, torch.float32), (pt3d_w, torch.float32), (has_depth, torch.uint8),
                      (count, torch.int32), (img0, torch.uint8), (img1, torch.uint8), (rand_state, torch.int32)):
            assert a.is_cuda and a.is_contiguous() and a.dtype == dt
        T = np.ascontiguousarray(poses7, np.float64).reshape(n, 7)
        out = torch.zeros((n, cap, 3), dtype=torch.float64, device=self.device)
        mask = torch.zeros((n, cap), dtype=torch.uint8, device=self.device)
        self._lib.flvis_hip_stereo_depth.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_void_p,
                                                     C.c_void_p, C.c_void_p]
        self._check(self._lib.flvis_hip_stereo_depth(self._h, C.byref(cfg), _ptr(img0), _ptr(img1), n, _ptr(pt2d_plane),
                                                     _ptr(pt2d_undistort), _ptr(pt3d_w), _ptr(has_depth), _ptr(count), cap,
                                                     T.ctypes.data, C.c_float(rng), _ptr(rand_state), _ptr(out), _ptr(mask)),
                    "stereo_depth")
        return out, mask

    def gftt(self, img, max_corners, quality, min_distance):
        import torch
        img = img.contiguous()
        n, h, w = img.shape
        out = torch.zeros((n, max_corners, 2), dtype=torch.float32, device=img.device)
        cnt = torch.zeros((n,), dtype=torch.int32, device=img.device)
        self._check(self._lib.flvis_hip_gftt(self._h, _ptr(img), w, h, n, max_corners, C.c_double(quality),
                                             C.c_double(min_distance), _ptr(out), _ptr(cnt)), "gftt")
        return out, cnt

    def debug_corner_response(self, img, variant, rows=0, key_cap=1 << 17):
        """test aid: (max ordered bits [n], sorted candidate keys per image) of the corner-response pass with the chosen kernel"""
        import numpy as np
        img = img.contiguous()
        n, h, w = img.shape
        mx = np.zeros(n, np.uint32)
        nk = np.zeros(n, np.int32)
        keys = np.zeros((n, key_cap), np.uint64)
        self._check(self._lib.flvis_hip_debug_corner_response(self._h, _ptr(img), w, h, n, int(variant), int(rows), _P(mx, C.c_uint32),
                                                              _P(nk, C.c_int), _P(keys, C.c_uint64), key_cap), "corner_response")
        return mx, [np.sort(keys[i, :nk[i]]) for i in range(n)]

    def debug_sqrt_check(self, first_bits, n):
        """test aid: arguments in [first_bits, first_bits + n) (float bit patterns) on which the corner-response kernel's square
        root differs from the correctly rounded sqrtf"""
        bad = C.c_uint64(0)
        self._check(self._lib.flvis_hip_debug_sqrt_check(self._h, C.c_uint32(first_bits), C.c_uint32(n), C.byref(bad)), "sqrt_check")
        return bad.value

    def feature_dem_detect(self, img, f_para, out_cap=1024):
        import torch
        img = img.contiguous()
        n, h, w = img.shape
        fp = (C.c_double * 6)(*[float(x) for x in f_para])
        out = torch.zeros((n, out_cap, 2), dtype=torch.float32, device=img.device)
        cnt = torch.zeros((n,), dtype=torch.int32, device=img.device)
        self._check(self._lib.flvis_hip_feature_dem_detect(self._h, _ptr(img), w, h, n, fp, _ptr(out), _ptr(cnt),
                                                           out_cap), "feature_dem_detect")
        return out, cnt

    def feature_dem_redetect(self, img, f_para, exist_xy, exist_count, out_cap=1024):
        import torch
        img = img.contiguous()
        n, h, w = img.shape
        fp = (C.c_double * 6)(*[float(x) for x in f_para])
        assert exist_xy.dtype == torch.float64 and exist_count.dtype == torch.int32
        out = torch.zeros((n, out_cap, 2), dtype=torch.float32, device=img.device)
        cnt = torch.zeros((n,), dtype=torch.int32, device=img.device)
        self._check(self._lib.flvis_hip_feature_dem_redetect(self._h, _ptr(img), w, h, n, fp, _ptr(exist_xy.contiguous()),
                                                             _ptr(exist_count), exist_xy.shape[1], _ptr(out), _ptr(cnt),
                                                             out_cap), "feature_dem_redetect")
        return out, cnt


# ---------------------------------------------------------------------------------------------------- pipeline level

    # ---- ORB extraction + Hamming matching ---------------------------------------------------------
    def resize_linear(self, img, dw, dh):
        import torch
        img = img.contiguous()
        n, h, w = img.shape
        out = torch.empty((n, dh, dw), dtype=torch.uint8, device=img.device)
        self._check(self._lib.flvis_hip_resize_linear(self._h, _ptr(img), w, h, _ptr(out), int(dw), int(dh), n), "resize_linear")
        return out

    def fast_score(self, img, threshold):
        import torch
        img = img.contiguous()
        n, h, w = img.shape
        out = torch.empty_like(img)
        self._check(self._lib.flvis_hip_fast_score(self._h, _ptr(img), w, h, n, int(threshold), _ptr(out)), "fast_score")
        return out

    def gaussian_blur7(self, img):
        import torch
        img = img.contiguous()
        n, h, w = img.shape
        out = torch.empty_like(img)
        self._check(self._lib.flvis_hip_gaussian_blur7(self._h, _ptr(img), _ptr(out), w, h, n), "gaussian_blur7")
        return out

    def orb_detect_and_compute(self, img, nfeatures=1000, scale_factor=1.2, nlevels=8, fast_threshold=20, pattern=None,
                               cap=2048):
        """img uint8 [n,h,w] -> (kps float32 [n,cap,6] (x, y, size, angle, response, octave), desc uint8 [n,cap,32],
        count int32 [n], overflow int32 [n])."""
        import numpy as np
        import torch
        img = img.contiguous()
        n, h, w = img.shape
        kps = torch.zeros((n, cap, 6), dtype=torch.float32, device=img.device)
        desc = torch.zeros((n, cap, 32), dtype=torch.uint8, device=img.device)
        cnt = torch.zeros((n,), dtype=torch.int32, device=img.device)
        ovf = torch.zeros((n,), dtype=torch.int32, device=img.device)
        prm = OrbParams(int(nfeatures), float(scale_factor), int(nlevels), int(fast_threshold))
        pat = None
        if pattern is not None:
            pat = np.ascontiguousarray(pattern, np.int8)
            assert pat.size == 1024
        self._check(self._lib.flvis_hip_orb_detect_and_compute(
            self._h, _ptr(img), w, h, n, C.byref(prm), C.c_void_p(pat.ctypes.data if pat is not None else 0), _ptr(kps),
            _ptr(desc), _ptr(cnt), cap, _ptr(ovf)), "orb_detect_and_compute")
        return kps, desc, cnt, ovf

    def hamming_knn2(self, query, nq, train, nt):
        """query uint8 [p,qcap,32], nq int32 [p], train uint8 [p,tcap,32], nt int32 [p] -> (idx, dist) int32 [p,qcap,2]."""
        import torch
        query, train = query.contiguous(), train.contiguous()
        p, qcap, _ = query.shape
        tcap = train.shape[1]
        idx = torch.full((p, qcap, 2), -7, dtype=torch.int32, device=query.device)
        dist = torch.full((p, qcap, 2), -7, dtype=torch.int32, device=query.device)
        self._check(self._lib.flvis_hip_hamming_knn2(self._h, _ptr(query), _ptr(nq), qcap, _ptr(train), _ptr(nt), tcap, p,
                                                     _ptr(idx), _ptr(dist)), "hamming_knn2")
        return idx, dist

    def orb_match(self, a, na, b, nb, ratio_max):
        """mutual-best + ratio test -> (pairs int32 [p,acap,2], npairs int32 [p])."""
        import torch
        a, b = a.contiguous(), b.contiguous()
        p, acap, _ = a.shape
        bcap = b.shape[1]
        pairs = torch.full((p, acap, 2), -1, dtype=torch.int32, device=a.device)
        npairs = torch.zeros((p,), dtype=torch.int32, device=a.device)
        self._check(self._lib.flvis_hip_orb_match(self._h, _ptr(a), _ptr(na), acap, _ptr(b), _ptr(nb), bcap, p,
                                                  C.c_double(ratio_max), _ptr(pairs), _ptr(npairs)), "orb_match")
        return pairs, npairs


    def bow_set_vocabulary(self, child_ptr, child_idx, desc, weight, word_id):
        """flvis_hip_bow_set_vocabulary: the DBoW3 tree as flat arrays (see include/flvis_hip.h)."""
        import numpy as np
        cp = np.ascontiguousarray(child_ptr, np.int32)
        ci = np.ascontiguousarray(child_idx, np.int32)
        ds = np.ascontiguousarray(desc, np.uint8)
        wt = np.ascontiguousarray(weight, np.float64)
        wi = np.ascontiguousarray(word_id, np.int32)
        n = len(cp) - 1
        assert ds.shape == (n, 32) and len(wt) == n and len(wi) == n
        self._check(self._lib.flvis_hip_bow_set_vocabulary(self._h, n, _P(cp, C.c_int), _P(ci, C.c_int), _P(ds, C.c_uint8),
                                                           _P(wt, C.c_double), _P(wi, C.c_int)), "bow_set_vocabulary")

    def bow_load_vocabulary(self, path):
        """flvis_hip_bow_load_vocabulary: `Vocabulary voc(path)` of vo_loopclosing.cpp:1097 (.dbow3 / .txt / .yml / .yml.gz)."""
        self._check(self._lib.flvis_hip_bow_load_vocabulary(self._h, C.c_char_p(os.fsencode(path))), "bow_load_vocabulary")

    def bow_transform(self, desc, count, vcap=2048):
        """desc uint8 [n,dcap,32], count int32 [n] (device) -> (ids int32 [n,vcap], vals float64 [n,vcap], nnz int32 [n])."""
        import torch
        desc = desc.contiguous()
        n, dcap, _ = desc.shape
        ids = torch.full((n, vcap), -1, dtype=torch.int32, device=desc.device)
        vals = torch.zeros((n, vcap), dtype=torch.float64, device=desc.device)
        nnz = torch.zeros((n,), dtype=torch.int32, device=desc.device)
        self._check(self._lib.flvis_hip_bow_transform(self._h, _ptr(desc), _ptr(count), dcap, n, vcap, _ptr(ids), _ptr(vals),
                                                      _ptr(nnz)), "bow_transform")
        return ids, vals, nnz

    def bow_score(self, q_ids, q_vals, q_nnz, db_ids, db_vals, db_nnz):
        """one similarity-matrix row: query (1-D device tensors + nnz [1]) against db [m,vcap] -> scores float64 [m]."""
        import torch
        m, vcap = db_ids.shape
        scores = torch.full((m,), -1.0, dtype=torch.float64, device=db_ids.device)
        self._check(self._lib.flvis_hip_bow_score(self._h, _ptr(q_ids), _ptr(q_vals), _ptr(q_nnz), _ptr(db_ids), _ptr(db_vals),
                                                  _ptr(db_nnz), vcap, m, _ptr(scores)), "bow_score")
        return scores

    def bow_score_jobs(self, jobs, ids, vals, nnz):
        """flvis_hip_bow_score_jobs: jobs [(query vector, first database vector, n database vectors)] over one store ids / vals
        [n_vectors, vcap], nnz [n_vectors] (device) -> scores float64 [n_vectors] (entries outside the jobs' ranges stay -1)."""
        import numpy as np
        import torch
        j = np.ascontiguousarray(jobs, np.int32).reshape(-1, 3)
        nv, vcap = ids.shape
        scores = torch.full((nv,), -1.0, dtype=torch.float64, device=ids.device)
        self._check(self._lib.flvis_hip_bow_score_jobs(self._h, len(j), _P(j, C.c_int), _ptr(ids), _ptr(vals), _ptr(nnz), vcap, _ptr(scores)),
                    "bow_score_jobs")
        return scores

    def lc_keyframe_landmarks(self, img0, img1, cam_type, kps, desc, count, P0=None, P1=None, K4=None, in_place=False):
        """flvis_hip_lc_keyframe_landmarks (vo_loopclosing.cpp:255-372): kps float32 [n,cap,6], desc uint8 [n,cap,32], count int32 [n] as
        orb_detect_and_compute returns them; img0 uint8 [n,h,w]; img1 uint8 (stereo, cam_type 0) or int16/uint16 Z16 (depth, cam_type 2).
        Returns (lm_2d float32 [n,cap,2], lm_3d float64 [n,cap,3], lm_desc uint8 [n,cap,32], lm_count int32 [n])."""
        import numpy as np
        import torch
        kps, desc = kps.contiguous(), desc.contiguous()
        n, cap, _ = kps.shape
        ref = img0 if img0 is not None else img1
        h, w = ref.shape[-2:]
        img0 = img0.contiguous() if img0 is not None else None
        img1 = img1.contiguous() if img1 is not None else None
        dbl = lambda a, m: None if a is None else np.ascontiguousarray(a, np.float64).reshape(m)
        p0, p1, k4 = dbl(P0, 12), dbl(P1, 12), dbl(K4, 4)
        hp = lambda a: _P(a, C.c_double) if a is not None else None
        lm2 = torch.zeros((n, cap, 2), dtype=torch.float32, device=kps.device)
        lm3 = torch.zeros((n, cap, 3), dtype=torch.float64, device=kps.device)
        lmd = desc if in_place else torch.zeros_like(desc)
        cnt = torch.zeros((n,), dtype=torch.int32, device=kps.device)
        self._check(self._lib.flvis_hip_lc_keyframe_landmarks(
            self._h, _ptr(img0) if img0 is not None else C.c_void_p(0), _ptr(img1) if img1 is not None else C.c_void_p(0), w, h, n,
            int(cam_type), hp(p0), hp(p1), hp(k4), _ptr(kps), _ptr(desc), _ptr(count), cap, _ptr(lm2), _ptr(lm3), _ptr(lmd), _ptr(cnt)),
            "lc_keyframe_landmarks")
        return lm2, lm3, lmd, cnt

    def pnp_ransac(self, p3d, p2d, count, K4, seeds, iterations=100, reproj_px=2.0, confidence=0.99):
        """flvis_hip_pnp_ransac: p3d float32 [n,cap,3], p2d float32 [n,cap,2], count int32 [n] (device) -> (pose7 [n,7], mask [n,cap],
        n_inliers [n])."""
        import numpy as np
        import torch
        p3d, p2d = p3d.contiguous(), p2d.contiguous()
        n, cap, _ = p3d.shape
        K = np.ascontiguousarray(K4, np.float64)
        sd = np.ascontiguousarray(seeds, np.uint64)
        assert len(sd) == n and len(K) == 4
        pose = torch.zeros((n, 7), dtype=torch.float64, device=p3d.device)
        mask = torch.zeros((n, cap), dtype=torch.uint8, device=p3d.device)
        ninl = torch.zeros((n,), dtype=torch.int32, device=p3d.device)
        self._check(self._lib.flvis_hip_pnp_ransac(self._h, _ptr(p3d), _ptr(p2d), _ptr(count), cap, n, _P(K, C.c_double), int(iterations),
                                                   C.c_double(reproj_px), C.c_double(confidence), _P(sd, C.c_uint64), _ptr(pose),
                                                   _ptr(mask), _ptr(ninl)), "pnp_ransac")
        return pose, mask, ninl

    def debug_epnp(self, p3d, p2d, count, K4):
        """flvis_hip_debug_epnp: EPnP alone on correspondence sets (p3d float32 [n,cap,3], p2d float32 [n,cap,2], count int32 [n], device)
        -> float64 [n,160] (layout in include/flvis_hip.h)."""
        import numpy as np
        import torch
        p3d, p2d = p3d.contiguous(), p2d.contiguous()
        n, cap, _ = p3d.shape
        K = np.ascontiguousarray(K4, np.float64)
        out = torch.zeros((n, 160), dtype=torch.float64, device=p3d.device)
        self._check(self._lib.flvis_hip_debug_epnp(self._h, _ptr(p3d), _ptr(p2d), _ptr(count), cap, n, _P(K, C.c_double), _ptr(out)), "debug_epnp")
        return out

    def pgo_loop_closure(self, T_c_w_list, present_list, loops_list, loop_poses_list, iterations=100, use_initial_guess=True):
        """flvis_hip_pgo_loop_closure for a batch of pose graphs (lists of per-graph numpy arrays: T_c_w [n,7], present [n], loops
        [m,2], loop poses [m,7]).  Returns (list of optimised T_c_w arrays, drift [g,7], stats [g,5], ran [g])."""
        import numpy as np
        import torch
        g = len(T_c_w_list)
        n_kf = np.array([len(t) for t in T_c_w_list], np.int32)
        n_loops = np.array([len(l) for l in loops_list], np.int32)
        T = torch.from_numpy(np.ascontiguousarray(np.concatenate(T_c_w_list), np.float64).reshape(-1, 7)).cuda()
        pres = np.ascontiguousarray(np.concatenate(present_list), np.uint8)
        loops = np.ascontiguousarray(np.concatenate([np.asarray(l, np.int32).reshape(-1, 2) for l in loops_list] + [np.zeros((1, 2), np.int32)]), np.int32)
        lp = torch.from_numpy(np.ascontiguousarray(np.concatenate([np.asarray(p, np.float64).reshape(-1, 7) for p in loop_poses_list] +
                                                                  [np.zeros((1, 7))]))).cuda()
        drift = torch.zeros((g, 7), dtype=torch.float64, device="cuda")
        stats = torch.zeros((g, 5), dtype=torch.float64, device="cuda")
        ran = np.zeros(g, np.int32)
        self._check(self._lib.flvis_hip_pgo_loop_closure(self._h, g, _P(n_kf, C.c_int), _ptr(T), _P(pres, C.c_uint8), _P(n_loops, C.c_int),
                                                         _P(loops, C.c_int), _ptr(lp), int(iterations), int(bool(use_initial_guess)),
                                                         _ptr(drift), _ptr(stats), _P(ran, C.c_int)), "pgo_loop_closure")
        self._check(self._lib.flvis_hip_synchronize(self._h), "synchronize")
        out = T.cpu().numpy()
        res, o = [], 0
        for k in n_kf:
            res.append(out[o:o + k].copy())
            o += k
        return res, drift.cpu().numpy(), stats.cpu().numpy(), ran


def loop_candidate(row, present, lcKFDist, lcKFMaxDist, lcNKFClosest, minScore):
    """flvis_loop_candidate (host control logic of isLoopCandidate): returns the earlier keyframe's index or None."""
    import numpy as np
    lib = load_library()
    row = np.ascontiguousarray(row, np.float64)
    pres = np.ascontiguousarray(present, np.uint8)
    out = C.c_int64(-1)
    lib.flvis_loop_candidate.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_int, C.c_double,
                                         C.POINTER(C.c_int64)]
    r = lib.flvis_loop_candidate(len(row), _P(row, C.c_double), _P(pres, C.c_uint8), lcKFDist, lcKFMaxDist, lcNKFClosest,
                                 C.c_double(minScore), C.byref(out))
    if r < 0:
        raise FlvisError("flvis_loop_candidate failed: %d" % r)
    return int(out.value) if r == 1 else None


def read_vocabulary_file(path):
    """flvis_voc_file_* (host only): a DBoW3 vocabulary file as the flat arrays `Context.bow_set_vocabulary` takes.

    Returns a dict: child_ptr, child_idx, desc [n,32], weight, word_id (-1 on inner nodes), k, L, scoring, weighting, n_words,
    layout ("binary" | "binary-quicklz" | "text" | "yaml")."""
    import numpy as np
    lib = load_library()
    h = C.c_void_p(0)
    err = C.create_string_buffer(512)
    lib.flvis_voc_file_open.argtypes = [C.c_char_p, C.POINTER(C.c_void_p), C.c_char_p, C.c_int]
    rc = lib.flvis_voc_file_open(os.fsencode(path), C.byref(h), err, 512)
    if rc != FLVIS_OK:
        raise FlvisError("flvis_voc_file_open(%s): %s" % (path, err.value.decode(errors="replace") or rc))
    try:
        info = (C.c_int * 8)()
        lib.flvis_voc_file_info.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        lib.flvis_voc_file_info(h, info)
        n, n_words, k, L, scoring, weighting, n_edges, layout = list(info)
        ptrs = [C.POINTER(C.c_int)(), C.POINTER(C.c_int)(), C.POINTER(C.c_uint8)(), C.POINTER(C.c_double)(), C.POINTER(C.c_int)()]
        lib.flvis_voc_file_arrays.argtypes = [C.c_void_p] + [C.c_void_p] * 5
        lib.flvis_voc_file_arrays(h, *[C.byref(q) for q in ptrs])
        take = lambda q, cnt, dt: np.ctypeslib.as_array(q, shape=(cnt,)).astype(dt, copy=True) if cnt else np.zeros(0, dt)
        return {"child_ptr": take(ptrs[0], n + 1, np.int32), "child_idx": take(ptrs[1], n_edges, np.int32),
                "desc": take(ptrs[2], n * 32, np.uint8).reshape(n, 32), "weight": take(ptrs[3], n, np.float64),
                "word_id": take(ptrs[4], n, np.int32), "k": k, "L": L, "scoring": scoring, "weighting": weighting,
                "n_words": n_words, "layout": ["binary", "binary-quicklz", "text", "yaml"][layout]}
    finally:
        lib.flvis_voc_file_close.argtypes = [C.c_void_p]
        lib.flvis_voc_file_close(h)


class OrbParams(C.Structure):
    """flvis_orb_params of include/flvis_hip.h."""
    _fields_ = [("nfeatures", C.c_int), ("scale_factor", C.c_float), ("nlevels", C.c_int), ("fast_threshold", C.c_int)]


def orb_default_pattern():
    """flvis_orb_default_pattern (host only): the built-in 256-pair sampling pattern as int8 [512,2]."""
    import numpy as np
    p = np.zeros((512, 2), np.int8)
    rc = load_library().flvis_orb_default_pattern(C.c_void_p(p.ctypes.data))
    if rc != FLVIS_OK:
        raise FlvisError("flvis_orb_default_pattern failed")
    return p


class FlvisCfg(C.Structure):
    """flvis_cfg of include/flvis_hip.h."""
    _fields_ = [("type_of_vi", C.c_int), ("image_width", C.c_int), ("image_height", C.c_int),
                ("cam0_intrinsics", C.c_double * 4), ("cam0_distortion", C.c_double * 4),
                ("cam1_intrinsics", C.c_double * 4), ("cam1_distortion", C.c_double * 4),
                ("T_imu_cam0", C.c_double * 16), ("T_cam0_cam1", C.c_double * 16),
                ("vifusion_para", C.c_double * 6), ("feature_para", C.c_double * 6), ("dr_para", C.c_double * 3),
                ("window_size", C.c_int),
                ("cam_type", C.c_int), ("imu_type", C.c_int), ("skip_first_n_imgs", C.c_int),
                ("need_equal_hist", C.c_int),
                ("R0", C.c_double * 9), ("R1", C.c_double * 9), ("P0", C.c_double * 12), ("P1", C.c_double * 12),
                ("depth_factor", C.c_double)]


class FrameOut(C.Structure):
    """flvis_frame_out of include/flvis_hip.h."""
    _fields_ = [("state", C.c_int), ("new_keyframe", C.c_int), ("reset_cmd", C.c_int), ("n_landmarks", C.c_int),
                ("frame_id", C.c_int64), ("T_c_w", C.c_double * 7),
                ("of_inliers", C.c_int), ("f_inliers", C.c_int), ("pnp_inliers", C.c_int), ("pad_", C.c_int),
                ("reprojection_error", C.c_double)]


def load_config(yaml_path):
    """flvis_config_load: accepts the reference's yaml files unchanged.  Host-only (no GPU needed)."""
    lib = load_library()
    cfg = FlvisCfg()
    err = C.create_string_buffer(256)
    rc = lib.flvis_config_load(yaml_path.encode(), C.byref(cfg), err, 256)
    if rc != FLVIS_OK:
        raise FlvisError("flvis_config_load(%s) failed (%d): %s" % (yaml_path, rc, err.value.decode()))
    return cfg


def _P(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class FlvisImage(C.Structure):
    """flvis_image of include/flvis_hip.h: one host image (pitch in bytes)."""
    _fields_ = [("data", C.POINTER(C.c_uint8)), ("width", C.c_int), ("height", C.c_int), ("pitch", C.c_int), ("channels", C.c_int),
                ("t", C.c_double)]


class LcParams(C.Structure):
    """flvis_lc_params of include/flvis_hip.h (LC_PARAS, vo_loopclosing.cpp:86-97)."""
    _fields_ = [("lcKFStart", C.c_int), ("lcKFDist", C.c_int), ("lcKFMaxDist", C.c_int), ("lcKFLast", C.c_int), ("lcNKFClosest", C.c_int),
                ("minPts", C.c_int), ("ratioMax", C.c_double), ("ratioRansac", C.c_double), ("minScore", C.c_double)]


class LcEvent(C.Structure):
    """flvis_lc_event of include/flvis_hip.h."""
    _fields_ = [("kf_prev", C.c_int64), ("kf_curr", C.c_int64), ("candidate", C.c_int), ("n_matches", C.c_int), ("n_inliers", C.c_int),
                ("loop_accepted", C.c_int), ("optimised", C.c_int), ("pgo_iterations", C.c_int), ("loop_pose7", C.c_double * 7),
                ("chi2_before", C.c_double), ("chi2_after", C.c_double)]


def load_lc_params(yaml_path):
    """flvis_lc_params_load: the loop-closing block of the reference's yaml files.  Host-only."""
    prm = LcParams()
    err = C.create_string_buffer(256)
    rc = load_library().flvis_lc_params_load(os.fsencode(yaml_path), C.byref(prm), err, 256)
    if rc != FLVIS_OK:
        raise FlvisError("flvis_lc_params_load(%s) failed (%d): %s" % (yaml_path, rc, err.value.decode()))
    return prm


class LoopCloser:
    """flvis_loop_closer: LoopClosingNodeletClass (vo_loopclosing.cpp) for n_streams sequences; the keyframe database stays on the GPU."""

    def __init__(self, ctx, cfg, prm, n_streams=1, max_keyframes=2000, orb_pattern=None):
        import numpy as np
        self._ctx, self._lib, self.n_streams = ctx, ctx._lib, n_streams
        self.cfg, self.max_keyframes = cfg, int(max_keyframes)
        if isinstance(prm, dict):
            prm = LcParams(**prm)
        pat = None
        if orb_pattern is not None:
            pat = np.ascontiguousarray(orb_pattern, np.int8)
            assert pat.size == 1024
        h = C.c_void_p(0)
        ctx._check(self._lib.flvis_loop_closer_create(ctx._h, C.byref(cfg), C.byref(prm), int(n_streams), int(max_keyframes),
                                                      C.c_void_p(pat.ctypes.data if pat is not None else 0), C.byref(h)), "loop_closer_create")
        self._h = h
        self._lib.flvis_loop_closer_destroy.argtypes = [C.c_void_p]

    def close(self):
        if getattr(self, "_h", None):
            self._lib.flvis_loop_closer_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_keyframes(self, streams, img0, img1, T_c_w_odom):
        """streams: the sequence of each keyframe (distinct); img0 uint8 [n,h,w], img1 uint8 / Z16 [n,h,w] (device); T_c_w_odom [n,7].
        Returns the keyframes' indices in their sequences."""
        import numpy as np
        st = np.ascontiguousarray(streams, np.int32)
        n = len(st)
        T = np.ascontiguousarray(T_c_w_odom, np.float64).reshape(n, 7)
        img0 = img0.contiguous()
        img1 = img1.contiguous() if img1 is not None else None
        hw = (int(self.cfg.image_height), int(self.cfg.image_width))
        assert img0.shape[0] == n and tuple(img0.shape[1:]) == hw, "img0 must be [n, image_height, image_width]"
        if img1 is not None:                                 # a wrong-size tensor would be read out of bounds on the device
            assert img1.shape[0] == n and tuple(img1.shape[1:]) == hw, "img1 must be [n, image_height, image_width]"
            assert img1.element_size() == (2 if self.cfg.cam_type == 2 else 1), "img1: uint8 (stereo) or 16-bit depth (depth rig)"
        ids = np.zeros(n, np.int64)
        self._ctx._check(self._lib.flvis_loop_closer_add_keyframes(self._h, n, _P(st, C.c_int), _ptr(img0), _ptr(img1), _P(T, C.c_double),
                                                                   _P(ids, C.c_int64)), "loop_closer_add_keyframes")
        return ids

    def add_keyframes_host(self, streams, img0, img1, T_c_w_odom):
        """flvis_loop_closer_add_keyframes_host: numpy images [n,h,w(+padding)]; img0 uint8, img1 uint8 or uint16 (depth rig).  A
        2-D-strided view (rows padded) is passed with its pitch."""
        import numpy as np
        st = np.ascontiguousarray(streams, np.int32)
        n = len(st)
        T = np.ascontiguousarray(T_c_w_odom, np.float64).reshape(n, 7)
        ids = np.zeros(n, np.int64)
        a = (FlvisImage * n)()
        b = (FlvisImage * n)()
        keep = []
        for i in range(n):
            for arr, dst in ((img0[i], a), (img1[i], b)):
                assert arr.ndim == 2 and arr.strides[1] == arr.itemsize
                keep.append(arr)
                dst[i] = FlvisImage(C.cast(C.c_void_p(arr.ctypes.data), C.POINTER(C.c_uint8)), arr.shape[1], arr.shape[0], arr.strides[0], 1, 0.0)
        self._ctx._check(self._lib.flvis_loop_closer_add_keyframes_host(self._h, n, _P(st, C.c_int), a, b, _P(T, C.c_double), _P(ids, C.c_int64)),
                         "loop_closer_add_keyframes_host")
        return ids

    def process(self):
        """-> list of n_streams dicts (flvis_lc_event)"""
        ev = (LcEvent * self.n_streams)()
        self._ctx._check(self._lib.flvis_loop_closer_process(self._h, ev), "loop_closer_process")
        return [dict(kf_prev=int(e.kf_prev), kf_curr=int(e.kf_curr), candidate=bool(e.candidate), n_matches=e.n_matches,
                     n_inliers=e.n_inliers, accepted=bool(e.loop_accepted), optimised=bool(e.optimised), pgo_iterations=e.pgo_iterations,
                     pose=[float(x) for x in e.loop_pose7], chi2_before=e.chi2_before, chi2_after=e.chi2_after) for e in ev]

    def poses(self, stream=0, cap=None):
        import numpy as np
        cap = int(cap) if cap else self.max_keyframes      # (never fewer rows than the sequence can hold: no silent truncation)
        n = C.c_int(0)
        buf = np.zeros((cap, 7))
        self._ctx._check(self._lib.flvis_loop_closer_poses(self._h, int(stream), _P(buf, C.c_double), cap, C.byref(n)), "loop_closer_poses")
        if n.value > cap:
            raise FlvisError("loop_closer_poses: %d keyframes, buffer of %d" % (n.value, cap))
        return buf[:n.value].copy()

    def keyframe(self, stream, kf, cap=1024):
        """flvis_loop_closer_keyframe -> dict(lm2 [k,2] f32, lm3 [k,3] f64, lmd [k,32] u8, bow=(ids, vals))"""
        import numpy as np
        lm2, lm3, lmd = np.zeros((cap, 2), np.float32), np.zeros((cap, 3)), np.zeros((cap, 32), np.uint8)
        bi, bv = np.zeros(cap, np.int32), np.zeros(cap)
        nl, nv = C.c_int(0), C.c_int(0)
        self._ctx._check(self._lib.flvis_loop_closer_keyframe(self._h, int(stream), int(kf), cap, _P(lm2, C.c_float), _P(lm3, C.c_double),
                                                              _P(lmd, C.c_uint8), C.byref(nl), _P(bi, C.c_int), _P(bv, C.c_double), C.byref(nv)),
                         "loop_closer_keyframe")
        k, v = min(nl.value, cap), min(nv.value, cap)
        return dict(lm2=lm2[:k].copy(), lm3=lm3[:k].copy(), lmd=lmd[:k].copy(), bow=(bi[:v].copy(), bv[:v].copy()))

    def drift(self, stream=0):
        import numpy as np
        T = np.zeros(7)
        self._ctx._check(self._lib.flvis_loop_closer_drift(self._h, int(stream), _P(T, C.c_double)), "loop_closer_drift")
        return T

    def similarity_row(self, stream=0, cap=None):
        import numpy as np
        cap = int(cap) if cap else self.max_keyframes
        n = C.c_int(0)
        buf = np.zeros(cap)
        self._ctx._check(self._lib.flvis_loop_closer_similarity_row(self._h, int(stream), _P(buf, C.c_double), cap, C.byref(n)),
                         "loop_closer_similarity_row")
        if n.value > cap:
            raise FlvisError("loop_closer_similarity_row: %d entries, buffer of %d" % (n.value, cap))
        return buf[:n.value].copy()


class Tracker:
    """Batched F2FTracking + LocalMap for n_streams independent streams on one GPU (flvis_tracker_create)."""

    def __init__(self, ctx, cfg, n_streams, seed_base=0xF1715, traj_capacity=0):
        import numpy as np
        self.ctx = ctx
        self.lib = ctx._lib
        self.S = n_streams
        self.cfg = cfg
        self.np = np
        self.lib.flvis_tracker_create.argtypes = [C.c_void_p, C.POINTER(FlvisCfg), C.c_int, C.c_uint64, C.c_int]
        ctx._check(self.lib.flvis_tracker_create(ctx._h, C.byref(cfg), n_streams, seed_base, traj_capacity),
                   "tracker_create")
        self._out = (FrameOut * n_streams)()

    def imu_feed_flvis(self, stream, samples7):
        np = self.np
        a = np.ascontiguousarray(samples7, np.float64).reshape(-1, 7)
        if len(a):
            self.ctx._check(self.lib.flvis_imu_feed_flvis_frame(self.ctx._h, stream, len(a), _P(a, C.c_double)),
                            "imu_feed")

    def imu_feed_sensor(self, stream, t, acc, gyro):
        a = (C.c_double * 3)(*[float(x) for x in acc])
        g = (C.c_double * 3)(*[float(x) for x in gyro])
        self.ctx._check(self.lib.flvis_imu_feed(self.ctx._h, stream, C.c_double(t), a, g), "imu_feed")

    def imu_feed_out(self, stream, t, acc, gyro):
        """F2FTracking::imu_feed with its outputs: integrates the sensor-frame sample now; returns (q_w_i wxyz, pos_w_i, vel_w_i)."""
        np = self.np
        a = (C.c_double * 3)(*[float(x) for x in acc])
        g = (C.c_double * 3)(*[float(x) for x in gyro])
        q, p, v = np.zeros(4), np.zeros(3), np.zeros(3)
        self.ctx._check(self.lib.flvis_imu_feed_out(self.ctx._h, stream, C.c_double(t), a, g, _P(q, C.c_double), _P(p, C.c_double),
                                                    _P(v, C.c_double)), "imu_feed_out")
        return q, p, v

    def imu_states(self, stream, cap=512):
        """Rows (t, q_w_i wxyz, pos_w_i, vel_w_i) of the IMU samples integrated since the previous call; also the rows lost."""
        np = self.np
        rows = np.zeros((cap, 11), np.float64)
        n, dropped = C.c_int(0), C.c_int(0)
        self.ctx._check(self.lib.flvis_get_imu_states(self.ctx._h, stream, cap, _P(rows, C.c_double), C.byref(n), C.byref(dropped)),
                        "get_imu_states")
        return rows[:n.value].copy(), dropped.value

    def local_map_counts(self):
        """(keyframes emitted, local-map optimisations run) per stream"""
        np = self.np
        kf, ba = np.zeros(self.S, np.int64), np.zeros(self.S, np.int64)
        self.ctx._check(self.lib.flvis_get_local_map_counts(self.ctx._h, _P(kf, C.c_int64), _P(ba, C.c_int64)), "get_local_map_counts")
        return kf, ba

    def write_imu_trajectory(self, rows11, path, min_dt=0.0, append=False, t_first=None):
        """The recorder on /imu_pose: rows of imu_states() as `stamp x y z qw qx qy qz` lines; returns the lines written.
        t_first: the stamp of the run's first row (flvis_write_imu_trajectory_run) -- pass it with every batch of a run whose first
        batch may cover less than min_dt."""
        np = self.np
        r = np.ascontiguousarray(rows11, np.float64).reshape(-1, 11)
        if t_first is not None:
            self.lib.flvis_write_imu_trajectory_run.argtypes = [C.POINTER(C.c_double), C.c_int, C.c_char_p, C.c_double, C.c_int, C.c_double]
            n = self.lib.flvis_write_imu_trajectory_run(_P(r, C.c_double), len(r), path.encode(), C.c_double(min_dt), int(append),
                                                        C.c_double(t_first))
        else:
            n = self.lib.flvis_write_imu_trajectory(_P(r, C.c_double), len(r), path.encode(), C.c_double(min_dt), int(append))
        if n < 0:
            raise FlvisError("write_imu_trajectory failed (%d)" % n)
        return n

    def image_feed(self, img0, img1, times, want_out=True, with_local_map=True):
        """img0/img1: uint8 cuda tensors [S,H,W]; times: sequence of S floats."""
        np = self.np
        assert img0.is_cuda and img0.is_contiguous() and img1.is_contiguous() and img0.shape[0] == self.S
        t = np.ascontiguousarray(times, np.float64)
        out = C.cast(self._out, C.c_void_p) if want_out else C.c_void_p(0)
        self.ctx._check(self.lib.flvis_image_feed(self.ctx._h, _ptr(img0), _ptr(img1), _P(t, C.c_double), out,
                                                  int(with_local_map)), "image_feed")
        if not want_out:
            return None
        res = []
        for o in self._out:
            res.append(dict(state=o.state, new_keyframe=bool(o.new_keyframe), reset_cmd=bool(o.reset_cmd),
                            n_landmarks=o.n_landmarks, frame_id=o.frame_id, pose7=np.array(o.T_c_w[:]),
                            dbg=np.array([o.of_inliers, o.f_inliers, o.pnp_inliers]),
                            reprojection_error=o.reprojection_error))
        return res

    def run_steps(self, steps, with_local_map=True):
        """flvis_run_steps: a batch of frames whose images are already in HBM, one C call (what bench.py times).  steps: a sequence of
        (img0, img1, times[, imu_counts, imu_samples]) -- uint8 cuda tensors [S,H,W], S floats, and optionally the IMU samples of the
        step for all streams (int32 [S], float64 [S, n, 7]).  The tensors must stay alive until the context is synchronised."""
        np = self.np

        class Step(C.Structure):
            _fields_ = [("d_img0", C.c_void_p), ("d_img1", C.c_void_p), ("h_times", C.c_void_p), ("h_imu_counts", C.c_void_p),
                        ("h_imu_samples", C.c_void_p), ("imu_samples_per_stream", C.c_int)]
        arr = (Step * len(steps))()
        keep = []
        for j, st in enumerate(steps):
            img0, img1, times = st[0], st[1], st[2]
            assert img0.is_cuda and img0.is_contiguous() and img1.is_contiguous() and img0.shape[0] == self.S
            t = np.ascontiguousarray(times, np.float64)
            keep.append(t)
            arr[j].d_img0, arr[j].d_img1, arr[j].h_times = img0.data_ptr(), img1.data_ptr(), t.ctypes.data
            if len(st) > 3 and st[3] is not None:
                cnt = np.ascontiguousarray(st[3], np.int32)
                smp = np.ascontiguousarray(st[4], np.float64)
                assert cnt.shape == (self.S,) and smp.ndim == 3 and smp.shape[0] == self.S and smp.shape[2] == 7
                keep += [cnt, smp]
                arr[j].h_imu_counts, arr[j].h_imu_samples, arr[j].imu_samples_per_stream = cnt.ctypes.data, smp.ctypes.data, smp.shape[1]
        self.lib.flvis_run_steps.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        self.ctx._check(self.lib.flvis_run_steps(self.ctx._h, len(steps), C.cast(arr, C.c_void_p), int(with_local_map), C.c_void_p(0)),
                        "run_steps")

    def _frame_outs(self):
        np = self.np
        return [dict(state=o.state, new_keyframe=bool(o.new_keyframe), reset_cmd=bool(o.reset_cmd),
                     n_landmarks=o.n_landmarks, frame_id=o.frame_id, pose7=np.array(o.T_c_w[:]),
                     dbg=np.array([o.of_inliers, o.f_inliers, o.pnp_inliers]),
                     reprojection_error=o.reprojection_error) for o in self._out]

    def image_feed_host(self, imgs0, imgs1, times, want_out=True, with_local_map=True, hold_buffers=False):
        """flvis_image_feed_host: the frame handed over as HOST images, the call a nodelet makes (vo_tracking.cpp:396-430).
        imgs0 / imgs1: one numpy array per stream, [H, W] (mono8; uint16 for the depth image of a depth rig) or [H, W, 3|4]
        (BGR / BGRA); rows may be padded (a strided view whose pixels are contiguous).  With hold_buffers the arrays must stay
        untouched until the next call on this context has returned."""
        np = self.np
        assert len(imgs0) == self.S and len(imgs1) == self.S
        a = (FlvisImage * self.S)()
        b = (FlvisImage * self.S)()
        for arr, imgs in ((a, imgs0), (b, imgs1)):
            for s, im in enumerate(imgs):
                px = im.itemsize * (im.shape[2] if im.ndim == 3 else 1)
                assert im.strides[1] == px and (im.ndim == 2 or im.strides[2] == im.itemsize), "pixels of a row must be contiguous"
                arr[s].data = C.cast(im.ctypes.data, C.POINTER(C.c_uint8))
                arr[s].width, arr[s].height, arr[s].pitch = im.shape[1], im.shape[0], im.strides[0]
                arr[s].channels = im.shape[2] if im.ndim == 3 else 1
                arr[s].t = float(times[s])
        out = C.cast(self._out, C.c_void_p) if want_out else C.c_void_p(0)
        self.lib.flvis_image_feed_host.argtypes = [C.c_void_p, C.POINTER(FlvisImage), C.POINTER(FlvisImage), C.c_void_p, C.c_int, C.c_int]
        self.ctx._check(self.lib.flvis_image_feed_host(self.ctx._h, a, b, out, int(with_local_map), int(hold_buffers)),
                        "image_feed_host")
        return self._frame_outs() if want_out else None

    def landmarks(self, stream, cap=2048):
        np = self.np
        ids = np.zeros(cap, np.int64)
        p2d = np.zeros((cap, 2))
        p2u = np.zeros((cap, 2))
        p3w = np.zeros((cap, 3))
        fl = np.zeros(cap, np.uint8)
        n = self.lib.flvis_get_landmarks(self.ctx._h, stream, cap, _P(ids, C.c_int64), _P(p2d, C.c_double),
                                         _P(p2u, C.c_double), _P(p3w, C.c_double), _P(fl, C.c_uint8))
        if n < 0:
            self.ctx._check(n, "get_landmarks")
        return dict(ids=ids[:n].copy(), p2d=p2d[:n].copy(), p2u=p2u[:n].copy(), p3w=p3w[:n].copy(), flags=fl[:n].copy())

    def keyframe(self, stream, cap=2048):
        np = self.np
        fid = C.c_int64(0)
        pose = np.zeros(7)
        ids = np.zeros(cap, np.int64)
        p2u = np.zeros((cap, 2))
        p3w = np.zeros((cap, 3))
        n = self.lib.flvis_get_keyframe(self.ctx._h, stream, cap, C.byref(fid), _P(pose, C.c_double), _P(ids, C.c_int64),
                                        _P(p2u, C.c_double), _P(p3w, C.c_double))
        if n < 0:
            self.ctx._check(n, "get_keyframe")
        return dict(frame_id=fid.value, pose7=pose, lm_id=ids[:n].copy(), lm_2d=p2u[:n].copy(), lm_3d=p3w[:n].copy())

    def correction(self, stream, cap=8192):
        np = self.np
        fid = C.c_int64(0)
        pose = np.zeros(7)
        cnt = C.c_int(0)
        ids = np.zeros(cap, np.int64)
        p3 = np.zeros((cap, 3))
        oc = C.c_int(0)
        oid = np.zeros(cap, np.int64)
        r = self.lib.flvis_get_correction(self.ctx._h, stream, cap, C.byref(fid), _P(pose, C.c_double), C.byref(cnt),
                                          _P(ids, C.c_int64), _P(p3, C.c_double), C.byref(oc), _P(oid, C.c_int64))
        if r < 0:
            self.ctx._check(r, "get_correction")
        if r == 0:
            return None
        return dict(frame_id=fid.value, pose7=pose, lm_id=ids[:cnt.value].copy(), lm_3d=p3[:cnt.value].copy(),
                    outlier_id=oid[:oc.value].copy())

    def set_input_hold(self, n_frames):
        """flvis_set_input_hold: multi-lane trackers -- the caller leaves a call's input images untouched during the next n calls."""
        self.ctx._check(self.lib.flvis_set_input_hold(self.ctx._h, int(n_frames)), "set_input_hold")

    def set_imu_factor(self, enable, sigma_gyro=0.002):
        """flvis_set_imu_factor: gyro rotation-preintegration edges between consecutive keyframes in the window BA (off by default)."""
        self.ctx._check(self.lib.flvis_set_imu_factor(self.ctx._h, int(bool(enable)), C.c_double(sigma_gyro)), "set_imu_factor")

    def set_imu_factor_accel(self, sigma_acc):
        """flvis_set_imu_factor_accel: position rows of the IMU factor (accelerometer noise density; <= 0: rotation rows only)"""
        self.ctx._check(self.lib.flvis_set_imu_factor_accel(self.ctx._h, C.c_double(sigma_acc)), "set_imu_factor_accel")

    def get_keyframe_imu_pos(self, stream):
        """flvis_get_keyframe_imu_pos -> (dp, va): displacement preintegrated since the previous keyframe, that keyframe's velocity"""
        np = self.np
        dp, va = np.zeros(3), np.zeros(3)
        r = self.lib.flvis_get_keyframe_imu_pos(self.ctx._h, stream, _P(dp, C.c_double), _P(va, C.c_double))
        if r < 0:
            self.ctx._check(r, "get_keyframe_imu_pos")
        return dp, va

    def get_keyframe_imu(self, stream):
        """flvis_get_keyframe_imu -> (valid, dq (w, x, y, z), dt) of the stream's last keyframe"""
        np = self.np
        dq = np.zeros(4)
        dt = C.c_double(0)
        r = self.lib.flvis_get_keyframe_imu(self.ctx._h, stream, _P(dq, C.c_double), C.byref(dt))
        if r < 0:
            self.ctx._check(r, "get_keyframe_imu")
        return bool(r), dq, dt.value

    def ba_push_keyframe(self, stream, frame_id, pose7, lm_id, lm_2d, lm_3d, cap=8192, imu_dq=None, imu_dt=0.0, imu_dp=None, imu_va=None):
        np = self.np
        p7 = np.ascontiguousarray(pose7, np.float64)
        ids = np.ascontiguousarray(lm_id, np.int64)
        l2 = np.ascontiguousarray(lm_2d, np.float64)
        l3 = np.ascontiguousarray(lm_3d, np.float64)
        fid = C.c_int64(0)
        pose = np.zeros(7)
        cnt = C.c_int(0)
        oid_ = np.zeros(cap, np.int64)
        o3 = np.zeros((cap, 3))
        oc = C.c_int(0)
        ooid = np.zeros(cap, np.int64)
        if imu_dq is not None and imu_dp is not None:
            dq = np.ascontiguousarray(imu_dq, np.float64)
            dp, va = np.ascontiguousarray(imu_dp, np.float64), np.ascontiguousarray(imu_va, np.float64)
            r = self.lib.flvis_ba_push_keyframe_imu_pos(self.ctx._h, stream, C.c_int64(frame_id), _P(p7, C.c_double), _P(dq, C.c_double),
                                                        C.c_double(imu_dt), _P(dp, C.c_double), _P(va, C.c_double), len(ids),
                                                        _P(ids, C.c_int64), _P(l2, C.c_double), _P(l3, C.c_double), cap, C.byref(fid),
                                                        _P(pose, C.c_double), C.byref(cnt), _P(oid_, C.c_int64), _P(o3, C.c_double),
                                                        C.byref(oc), _P(ooid, C.c_int64))
        elif imu_dq is not None:
            dq = np.ascontiguousarray(imu_dq, np.float64)
            r = self.lib.flvis_ba_push_keyframe_imu(self.ctx._h, stream, C.c_int64(frame_id), _P(p7, C.c_double), _P(dq, C.c_double),
                                                    C.c_double(imu_dt), len(ids), _P(ids, C.c_int64), _P(l2, C.c_double),
                                                    _P(l3, C.c_double), cap, C.byref(fid), _P(pose, C.c_double), C.byref(cnt),
                                                    _P(oid_, C.c_int64), _P(o3, C.c_double), C.byref(oc), _P(ooid, C.c_int64))
        else:
            r = self.lib.flvis_ba_push_keyframe(self.ctx._h, stream, C.c_int64(frame_id), _P(p7, C.c_double), len(ids),
                                                _P(ids, C.c_int64), _P(l2, C.c_double), _P(l3, C.c_double), cap,
                                                C.byref(fid), _P(pose, C.c_double), C.byref(cnt), _P(oid_, C.c_int64),
                                                _P(o3, C.c_double), C.byref(oc), _P(ooid, C.c_int64))
        if r < 0:
            self.ctx._check(r, "ba_push_keyframe")
        if r == 0:
            return None
        return dict(frame_id=fid.value, pose7=pose, lm_id=oid_[:cnt.value].copy(), lm_3d=o3[:cnt.value].copy(),
                    outlier_id=ooid[:oc.value].copy())

    def trajectory(self, stream, first, n):
        np = self.np
        rows = np.zeros((n, 9))
        r = self.lib.flvis_get_trajectory(self.ctx._h, stream, first, n, _P(rows, C.c_double))
        if r < 0:
            self.ctx._check(r, "get_trajectory")
        return rows

    def write_trajectory(self, stream, first, n, path, fmt=0, min_dt=0.0):
        """flvis_write_trajectory: fmt 0 = `stamp x y z qw qx qy qz`, 1 = KITTI 12 columns.  Returns lines written."""
        self.lib.flvis_write_trajectory.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_double]
        r = self.lib.flvis_write_trajectory(self.ctx._h, stream, first, n, path.encode(), fmt, C.c_double(min_dt))
        if r < 0:
            self.ctx._check(r, "write_trajectory")
        return r

    def correction_feed(self, stream, frame_id, pose7, lm_id, lm_3d, outlier_id):
        """flvis_correction_feed: F2FTracking::correction_feed (opt-in local-map feedback)."""
        np = self.np
        pose7 = np.ascontiguousarray(pose7, np.float64)
        lm_id = np.ascontiguousarray(lm_id, np.int64)
        lm_3d = np.ascontiguousarray(lm_3d, np.float64).reshape(-1, 3)
        outlier_id = np.ascontiguousarray(outlier_id, np.int64)
        assert len(lm_id) == len(lm_3d)
        self.lib.flvis_correction_feed.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.POINTER(C.c_double), C.c_int,
                                                   C.POINTER(C.c_int64), C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int64)]
        self.ctx._check(self.lib.flvis_correction_feed(self.ctx._h, stream, int(frame_id), _P(pose7, C.c_double), len(lm_id),
                                                       _P(lm_id, C.c_int64), _P(lm_3d, C.c_double), len(outlier_id),
                                                       _P(outlier_id, C.c_int64)), "correction_feed")

    def pose_records(self, stream, cap=1024):
        """flvis_get_pose_records -> rows (frame_id, pose7), oldest first."""
        rows = self.np.zeros((cap, 8))
        self.lib.flvis_get_pose_records.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]
        n = self.lib.flvis_get_pose_records(self.ctx._h, stream, cap, _P(rows, C.c_double))
        if n < 0:
            self.ctx._check(n, "get_pose_records")
        return rows[:min(n, cap)].copy()

    def counters(self):
        """[frames fed, keyframes, local-map optimisations]"""
        c = (C.c_int64 * 3)()
        self.ctx._check(self.lib.flvis_get_counters(self.ctx._h, c), "get_counters")
        return list(c)

    def dropped_keyframes(self):
        """keyframes that met a full keyframe queue (flvis_get_counters_n [3]; 0 under the tracker's back-pressure)"""
        c = (C.c_int64 * 4)()
        self.ctx._check(self.lib.flvis_get_counters_n(self.ctx._h, 4, c), "get_counters_n")
        return int(c[3])
