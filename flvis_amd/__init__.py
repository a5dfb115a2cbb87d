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
    _LIB = C.CDLL(_build.LIB)
    _LIB.flvis_version.restype = C.c_char_p
    _LIB.flvis_last_error.restype = C.c_char_p
    _LIB.flvis_last_error.argtypes = [C.c_void_p]
    _LIB.flvis_hip_stream.restype = C.c_void_p
    _LIB.flvis_hip_stream.argtypes = [C.c_void_p]
    return _LIB


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class Context:
    """Owns a flvis_ctx bound to one GPU and one HIP stream (torch's current stream by default)."""

    def __init__(self, device=0, use_torch_stream=True):
        import torch
        self._lib = load_library()
        if not torch.cuda.is_available():
            raise FlvisError("flvis_amd needs a HIP device (MI355X); none is visible. No CPU fallback exists.")
        torch.cuda.set_device(device)
        self.device = torch.device("cuda", device)
        stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream) if use_torch_stream else C.c_void_p(0)
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

    def pyr_down(self, img):
        import torch
        assert img.dtype == torch.uint8 and img.is_cuda and img.is_contiguous() and img.dim() == 3
        n, h, w = img.shape
        dw, dh = (w + 1) // 2, (h + 1) // 2
        out = torch.empty((n, dh, dw), dtype=torch.uint8, device=img.device)
        self._check(self._lib.flvis_hip_pyr_down(self._h, _ptr(img), w, h, w, _ptr(out), dw, n), "pyr_down")
        return out

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

    def gftt(self, img, max_corners, quality, min_distance):
        import torch
        img = img.contiguous()
        n, h, w = img.shape
        out = torch.zeros((n, max_corners, 2), dtype=torch.float32, device=img.device)
        cnt = torch.zeros((n,), dtype=torch.int32, device=img.device)
        self._check(self._lib.flvis_hip_gftt(self._h, _ptr(img), w, h, n, max_corners, C.c_double(quality),
                                             C.c_double(min_distance), _ptr(out), _ptr(cnt)), "gftt")
        return out, cnt

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
