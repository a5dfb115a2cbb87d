// flvis_amd: extern "C" shim (include/flvis_hip.h) over the HIP kernels.  No CPU fallback anywhere: without a HIP
// device every entry point fails with FLVIS_ERR_NO_DEVICE.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/flvis_hip.h"
#include "ctx.hpp"
#include "pipeline.hpp"

using namespace flvis;

void* flvis_ctx::scratch(const std::string& name, size_t bytes, bool zero_on_alloc) {
  Buf& b = bufs[name];
  if (b.bytes >= bytes && b.p) return b.p;
  if (b.p) {
    hipStreamSynchronize(stream);
    hipFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
  }
  size_t want = (bytes + 255) / 256 * 256;
  if (hipMalloc(&b.p, want) != hipSuccess) {
    b.p = nullptr;
    return nullptr;
  }
  b.bytes = want;
  if (zero_on_alloc) hipMemsetAsync(b.p, 0, want, stream);
  return b.p;
}

#define CHECK_CTX(ctx) \
  if (!(ctx)) return FLVIS_ERR_INVALID_ARG;
#define CHECK_LAUNCH(ctx, what)                                  \
  do {                                                           \
    hipError_t e__ = hipGetLastError();                          \
    if (e__ != hipSuccess) return (ctx)->hip_fail(e__, what);    \
  } while (0)

extern "C" {

const char* flvis_version(void) { return "flvis_hip 0.1 (gfx950)"; }

int flvis_hip_create(int device, void* hip_stream, flvis_ctx** out) {
  if (!out) return FLVIS_ERR_INVALID_ARG;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
    (void)hipGetLastError();
    return FLVIS_ERR_NO_DEVICE;
  }
  if (hipSetDevice(device) != hipSuccess) return FLVIS_ERR_NO_DEVICE;
  flvis_ctx* c = new flvis_ctx();
  c->device = device;
  if (hip_stream == (void*)(intptr_t)-1) {  // FLVIS_STREAM_NEW: private non-blocking stream
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
      delete c;
      return FLVIS_ERR_HIP;
    }
    c->own_stream = true;
  } else {
    c->stream = (hipStream_t)hip_stream;  // NULL == the device's default (null) stream, as in HIP itself
  }
  if (img_kernels_init() != hipSuccess) {
    (void)hipGetLastError();
  }
  *out = c;
  return FLVIS_OK;
}

void flvis_pipeline_destroy_internal(flvis_ctx* ctx);  // pipeline.cpp
void flvis_pipeline_sync_internal(flvis_ctx* ctx);
long long flvis_pipeline_join_timeout_internal(flvis_ctx* ctx);

void flvis_hip_destroy(flvis_ctx* ctx) {
  if (!ctx) return;
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  flvis_pipeline_destroy_internal(ctx);
  for (auto& kv : ctx->bufs)
    if (kv.second.p) hipFree(kv.second.p);
  if (ctx->own_stream) hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char* flvis_last_error(const flvis_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int flvis_hip_synchronize(flvis_ctx* ctx) {
  CHECK_CTX(ctx);
  hipError_t e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) return ctx->hip_fail(e, "hipStreamSynchronize");
  flvis_pipeline_sync_internal(ctx);
  // a stream join (k_wait_flag, a folded wait) that gave up after its ~4 s: the frames behind it ran on whatever was there
  if (const long long seq = flvis_pipeline_join_timeout_internal(ctx)) {
    char msg[128];
    snprintf(msg, sizeof msg, "a stream join or an upload wait timed out (sequence number %lld): results since then are invalid", seq);
    return ctx->fail(FLVIS_ERR_HIP, msg);
  }
  return FLVIS_OK;
}

void* flvis_hip_stream(flvis_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int flvis_hip_equalize_hist(flvis_ctx* ctx, const uint8_t* d_src, uint8_t* d_dst, int w, int h, int n_img) {
  CHECK_CTX(ctx);
  if (!d_src || !d_dst || w <= 0 || h <= 0 || n_img <= 0 || (w & 3)) return ctx->fail(FLVIS_ERR_INVALID_ARG, "equalize_hist: bad args (w % 4 must be 0)");
  unsigned* hist = (unsigned*)ctx->scratch("eq_hist", sizeof(unsigned) * 256 * n_img, true);
  uint8_t* lut = (uint8_t*)ctx->scratch("eq_lut", 256 * (size_t)n_img);
  if (!hist || !lut) return ctx->fail(FLVIS_ERR_HIP, "equalize_hist: scratch allocation failed");
  launch_equalize_hist(ctx->stream, img_plain(d_src), img_plain(d_dst), w, h, w, w, (size_t)w * h, (size_t)w * h, n_img,
                       hist, lut, nullptr);
  CHECK_LAUNCH(ctx, "equalize_hist");
  return FLVIS_OK;
}

int flvis_hip_cvt_bgr_to_gray(flvis_ctx* ctx, const uint8_t* d_src, int channels, uint8_t* d_dst, int w, int h, int n_img) {
  CHECK_CTX(ctx);
  if (!d_src || !d_dst || (channels != 3 && channels != 4) || w <= 0 || h <= 0 || n_img <= 0 || (w & 3) ||
      ((uintptr_t)d_src & 3) || ((uintptr_t)d_dst & 3))
    return ctx->fail(FLVIS_ERR_INVALID_ARG, "cvt_bgr_to_gray: bad args (3 or 4 channels, w % 4 == 0, 4-byte aligned buffers)");
  const size_t npx = (size_t)w * h * n_img;
  if (npx / 4 > 0x7fffffffu) return ctx->fail(FLVIS_ERR_CAPACITY, "cvt_bgr_to_gray: batch too large");
  launch_bgr_to_gray(ctx->stream, d_src, channels, d_dst, npx);
  CHECK_LAUNCH(ctx, "cvt_bgr_to_gray");
  return FLVIS_OK;
}

int flvis_hip_pyr_down(flvis_ctx* ctx, const uint8_t* d_src, int w, int h, int src_pitch, uint8_t* d_dst,
                       int dst_pitch, int n_img) {
  CHECK_CTX(ctx);
  if (!d_src || !d_dst || w < 2 || h < 2 || n_img <= 0 || (src_pitch & 3) || src_pitch < w || dst_pitch < (w + 1) / 2)
    return ctx->fail(FLVIS_ERR_INVALID_ARG, "pyr_down: bad args (src_pitch % 4 must be 0)");
  const int nob[2] = {0, 0};
  // the tracker's own choice of kernel -- when the walking kernel's 16-byte row loads and 8-byte stores are aligned (else the tile kernels)
  if (pyr_walk_ok(w, h, 1, nob, nob, false) && !(dst_pitch & 7) && !((uintptr_t)d_dst & 7) && !(src_pitch & 15) && !((uintptr_t)d_src & 15)) {
    PyrSel q{};
    q.levels = 1;
    q.lvl[1] = img_plain(d_dst);
    q.w[0] = w, q.h[0] = h, q.w[1] = (w + 1) / 2, q.h[1] = (h + 1) / 2;
    q.pitch[1] = dst_pitch, q.stride[1] = (size_t)dst_pitch * ((h + 1) / 2);
    launch_pyr_walk(ctx->stream, img_plain(d_src), w, h, src_pitch, (size_t)src_pitch * h, q, 0, 1, false, n_img, nullptr);
  } else {
    launch_pyr_down(ctx->stream, img_plain(d_src), w, h, src_pitch, (size_t)src_pitch * h, img_plain(d_dst), dst_pitch,
                    (size_t)dst_pitch * ((h + 1) / 2), n_img, nullptr);
  }
  CHECK_LAUNCH(ctx, "pyr_down");
  return FLVIS_OK;
}

static int lk_levels(int w, int h, int win, int max_level) {
  int level = 0;
  for (; level <= max_level; ++level) {
    w = (w + 1) / 2;
    h = (h + 1) / 2;
    if (w <= win || h <= win) return level;
  }
  return max_level;
}

// builds levels 1..L of `img` ([n][h][w]) into scratch `name`, fills `pyr`
static int build_pyramid(flvis_ctx* ctx, const char* name, const uint8_t* d_img, int w, int h, int n_img, int L,
                         PyrSel& pyr) {
  pyr.levels = L;
  pyr.lvl[0] = img_plain(d_img);
  pyr.w[0] = w;
  pyr.h[0] = h;
  pyr.pitch[0] = w;
  pyr.stride[0] = (size_t)w * h;
  if (w & 3) {
    // rows that are not dword aligned (KITTI's 1241 x 376, tightly packed): level 0 is copied into a pitch-aligned buffer first, as
    // the tracker's ingest does
    const int pitch0 = align_up(w, 16);
    const size_t stride0 = (size_t)pitch0 * h;
    uint8_t* l0 = (uint8_t*)ctx->scratch(std::string(name) + "_l0", stride0 * n_img + 256);
    if (!l0) return ctx->fail(FLVIS_ERR_HIP, "pyramid scratch allocation failed");
    launch_copy_image_any(ctx->stream, img_plain(d_img), img_plain(l0), w, h, w, pitch0, (size_t)w * h, stride0, n_img, nullptr);
    pyr.lvl[0] = img_plain(l0);
    pyr.pitch[0] = pitch0;
    pyr.stride[0] = stride0;
  }
  size_t total = 0;
  size_t off[LK_MAX_LEVELS] = {0};
  int lw = w, lh = h;
  for (int l = 1; l <= L; l++) {
    lw = (lw + 1) / 2;
    lh = (lh + 1) / 2;
    pyr.w[l] = lw;
    pyr.h[l] = lh;
    pyr.pitch[l] = align_up(lw, 16);
    pyr.stride[l] = (size_t)pyr.pitch[l] * lh;
    off[l] = total;
    total += pyr.stride[l] * n_img + 256;
    total = (total + 255) / 256 * 256;
  }
  uint8_t* base = (uint8_t*)ctx->scratch(name, total + 256);
  if (!base && L > 0) return ctx->fail(FLVIS_ERR_HIP, "pyramid scratch allocation failed");
  for (int l = 1; l <= L; l++) {
    pyr.lvl[l] = img_plain(base + off[l]);
    launch_pyr_down(ctx->stream, pyr.lvl[l - 1], pyr.w[l - 1], pyr.h[l - 1], pyr.pitch[l - 1], pyr.stride[l - 1],
                    pyr.lvl[l], pyr.pitch[l], pyr.stride[l], n_img, nullptr);
  }
  return FLVIS_OK;
}

int flvis_hip_lk_track(flvis_ctx* ctx, const uint8_t* d_prev, const uint8_t* d_next, int w, int h, int n_img,
                       const float* d_prev_pts, float* d_next_pts, uint8_t* d_status, const int* d_count, int nmax,
                       int max_level, int max_iter, double eps, int use_initial_flow) {
  CHECK_CTX(ctx);
  if (!d_prev || !d_next || !d_prev_pts || !d_next_pts || !d_status || !d_count || w < 32 || h < 32 || n_img <= 0 || nmax <= 0 ||
      max_level < 0)
    return ctx->fail(FLVIS_ERR_INVALID_ARG, "lk_track: bad args");
  int L = lk_levels(w, h, 31, max_level);
  if (L >= LK_MAX_LEVELS) L = LK_MAX_LEVELS - 1;
  PyrSel pp, pn;
  int rc = build_pyramid(ctx, "lk_pyr_prev", d_prev, w, h, n_img, L, pp);
  if (rc) return rc;
  rc = build_pyramid(ctx, "lk_pyr_next", d_next, w, h, n_img, L, pn);
  if (rc) return rc;
  LKParams prm;
  prm.max_iter = max_iter < 0 ? 0 : (max_iter > 100 ? 100 : max_iter);
  double e = eps < 0 ? 0 : (eps > 10 ? 10 : eps);
  prm.eps2 = e * e;
  prm.min_eig = 1e-4f;
  prm.use_initial = use_initial_flow ? 1 : 0;
  launch_lk_track(ctx->stream, pp, pn, d_prev_pts, d_next_pts, d_status, d_count, nmax, n_img, prm, nullptr);
  CHECK_LAUNCH(ctx, "lk_track");
  return FLVIS_OK;
}

// srand(seed) for flvis_hip_stereo_depth's dummy depths: 35 words per set (glibc's r[34] ring + its position)
int flvis_hip_rand_seed(flvis_ctx* ctx, uint32_t seed, int32_t* d_state35, int n_sets) {
  CHECK_CTX(ctx);
  if (!d_state35 || n_sets <= 0) return ctx->fail(FLVIS_ERR_INVALID_ARG, "rand_seed: bad args");
  std::vector<int> h((size_t)35 * n_sets);
  glibc_seed(seed, h.data());
  h[34] = 0;
  for (int s = 1; s < n_sets; s++) memcpy(&h[(size_t)35 * s], h.data(), sizeof(int) * 35);
  hipError_t e = hipMemcpyAsync(d_state35, h.data(), sizeof(int) * h.size(), hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // (h is a local)
  if (e != hipSuccess) return ctx->hip_fail(e, "rand_seed");
  return FLVIS_OK;
}

// CameraFrame::recover3DPts_c_FromStereo (src/processing/camera_frame.cpp:93-180) in one call: seeds -> calcOpticalFlowPyrLK(img0, img1,
// 31 x 31, maxLevel 5, 30 iterations / 0.001, OPTFLOW_USE_INITIAL_FLOW) -> undistortPoints + DLT + validity + dummy depths
int flvis_hip_stereo_depth(flvis_ctx* ctx, const flvis_cfg* cfg, const uint8_t* d_img0, const uint8_t* d_img1, int n_sets,
                           const float* d_pt2d_plane, const float* d_pt2d_undistort, const float* d_pt3d_w, const uint8_t* d_has_depth,
                           const int* d_count, int cap, const double* h_T_c_w7, float range, int32_t* d_rand_state35, double* d_pt3d_c,
                           uint8_t* d_mask_has_3d) {
  CHECK_CTX(ctx);
  if (!cfg || !d_img0 || !d_img1 || !d_pt2d_plane || !d_pt2d_undistort || !d_pt3d_w || !d_has_depth || !d_count || !h_T_c_w7 ||
      !d_rand_state35 || !d_pt3d_c || !d_mask_has_3d || n_sets <= 0 || cap <= 0)
    return ctx->fail(FLVIS_ERR_INVALID_ARG, "stereo_depth: bad args");
  if (cfg->cam_type == CAM_DEPTH) return ctx->fail(FLVIS_ERR_CONFIG, "stereo_depth: the rig has a depth camera (recover3DPts_c_FromDepthImg), no stereo pair");
  const int w = cfg->image_width, h = cfg->image_height;
  if (w < 32 || h < 32) return ctx->fail(FLVIS_ERR_CONFIG, "stereo_depth: image too small");
  // (a configuration that never went through flvis_config_finalize has empty rectified projections: fx = 0 and points at infinity)
  if (cfg->P0[0] == 0.0 || cfg->P0[5] == 0.0 || cfg->P1[0] == 0.0)
    return ctx->fail(FLVIS_ERR_CONFIG, "stereo_depth: the configuration is not finalised (flvis_config_finalize: P0 / P1 are empty)");
  flvis_sd_cam cam;
  memcpy(cam.K1, cfg->cam1_intrinsics, 32);
  memcpy(cam.D1, cfg->cam1_distortion, 32);
  memcpy(cam.R1, cfg->R1, 72);
  memcpy(cam.P0, cfg->P0, 96);
  memcpy(cam.P1, cfg->P1, 96);
  pose7_from_mat44(cfg->T_cam0_cam1, cam.T_c1_c0, true);
  cam.fx = cfg->P0[0], cam.fy = cfg->P0[5], cam.cx = cfg->P0[2], cam.cy = cfg->P0[6];
  float* seeds = (float*)ctx->scratch("sd_seeds", sizeof(float) * 2 * (size_t)n_sets * cap);
  uint8_t* status = (uint8_t*)ctx->scratch("sd_status", (size_t)n_sets * cap);
  double* d_T = (double*)ctx->scratch("sd_pose", sizeof(double) * 7 * (size_t)n_sets);
  if (!seeds || !status || !d_T) return ctx->fail(FLVIS_ERR_HIP, "stereo_depth: scratch allocation failed");
  hipError_t e = hipMemcpyAsync(d_T, h_T_c_w7, sizeof(double) * 7 * (size_t)n_sets, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // (the caller's pose array may be reused when this call returns)
  if (e != hipSuccess) return ctx->hip_fail(e, "stereo_depth pose upload");
  launch_stereo_depth_seeds(ctx->stream, cam, d_pt2d_plane, d_pt3d_w, d_has_depth, d_count, cap, n_sets, d_T, seeds);
  int L = lk_levels(w, h, 31, 5);
  if (L >= LK_MAX_LEVELS) L = LK_MAX_LEVELS - 1;
  PyrSel pp, pn;
  int rc = build_pyramid(ctx, "lk_pyr_prev", d_img0, w, h, n_sets, L, pp);
  if (rc) return rc;
  rc = build_pyramid(ctx, "lk_pyr_next", d_img1, w, h, n_sets, L, pn);
  if (rc) return rc;
  LKParams prm;
  prm.max_iter = 30;
  prm.eps2 = 1e-3 * 1e-3;
  prm.min_eig = 1e-4f;
  prm.use_initial = 1;
  launch_lk_track(ctx->stream, pp, pn, d_pt2d_plane, seeds, status, d_count, cap, n_sets, prm, nullptr);
  launch_stereo_depth_post(ctx->stream, cam, d_pt2d_undistort, seeds, status, d_count, cap, n_sets, range, d_rand_state35, d_pt3d_c,
                           d_mask_has_3d);
  CHECK_LAUNCH(ctx, "stereo_depth");
  return FLVIS_OK;
}

static int gftt_scratch(flvis_ctx* ctx, int w, int h, int n_img, GfttScratch& sc) {
  int cap = 1;
  while (cap < (w / 2 + 1) * (h / 2 + 1)) cap <<= 1;  // strict 3x3 local maxima cannot exceed ~w*h/4
  sc.cap = cap;
  sc.maxenc = (unsigned*)ctx->scratch("gftt_max", sizeof(unsigned) * n_img);
  sc.nkeys = (int*)ctx->scratch("gftt_nkeys", sizeof(int) * n_img);
  sc.keys = (unsigned long long*)ctx->scratch("gftt_keys", sizeof(unsigned long long) * (size_t)cap * n_img);
  if (!sc.maxenc || !sc.nkeys || !sc.keys) return ctx->fail(FLVIS_ERR_HIP, "gftt: scratch allocation failed");
  return FLVIS_OK;
}

int flvis_hip_gftt(flvis_ctx* ctx, const uint8_t* d_img, int w, int h, int n_img, int max_corners, double quality,
                   double min_distance, float* d_out_xy, int* d_out_count) {
  CHECK_CTX(ctx);
  if (!d_img || !d_out_xy || !d_out_count || w < 8 || h < 8 || (w & 3) || n_img <= 0 || max_corners <= 0)
    return ctx->fail(FLVIS_ERR_INVALID_ARG, "gftt: bad args");
  if ((size_t)((w + 31) / 32) * h * 4 > 96 * 1024) return ctx->fail(FLVIS_ERR_CAPACITY, "gftt: image too large for the LDS bitmap");
  if (min_distance > 64.0) return ctx->fail(FLVIS_ERR_CAPACITY, "gftt: minDistance > 64 is not supported");
  GfttScratch sc;
  int rc = gftt_scratch(ctx, w, h, n_img, sc);
  if (rc) return rc;
  launch_gftt(ctx->stream, img_plain(d_img), w, h, w, (size_t)w * h, n_img, sc, nullptr, quality, nullptr, max_corners,
              min_distance, d_out_xy, d_out_count, max_corners, nullptr);
  CHECK_LAUNCH(ctx, "gftt");
  return FLVIS_OK;
}

static DemParams dem_params(int w, int h, const double* f) {
  DemParams p;
  p.regionWidth = (int)std::floor(w / 4.0);
  p.regionHeight = (int)std::floor(h / 4.0);
  p.boundary_dis = (int)std::floor(f[2] / 2.0);
  p.max_region_feature_num = (unsigned)f[0];
  return p;
}

static int dem_common(flvis_ctx* ctx, const uint8_t* d_img, int w, int h, int n_img, const double* f_para, int mode,
                      const double* d_exist_xy, const int* d_exist_count, int exist_cap, float* d_out_xy,
                      int* d_out_count, int out_cap) {
  if (!d_img || !f_para || !d_out_xy || !d_out_count || w < 8 || h < 8 || (w & 3) || n_img <= 0 || out_cap <= 0)
    return ctx->fail(FLVIS_ERR_INVALID_ARG, "feature_dem: bad args");
  int gftt_num = (int)f_para[3];
  int maxc = mode == 1 ? 2 * gftt_num : gftt_num;
  if (maxc <= 0 || maxc > 4096) return ctx->fail(FLVIS_ERR_CAPACITY, "feature_dem: gftt_num out of range (<=2048)");
  GfttScratch sc;
  int rc = gftt_scratch(ctx, w, h, n_img, sc);
  if (rc) return rc;
  float* corners = (float*)ctx->scratch("dem_corners", sizeof(float) * 2 * (size_t)maxc * n_img);
  int* ncorners = (int*)ctx->scratch("dem_ncorners", sizeof(int) * n_img);
  int* modes = (int*)ctx->scratch("dem_mode", sizeof(int) * n_img);
  if (!corners || !ncorners || !modes) return ctx->fail(FLVIS_ERR_HIP, "feature_dem: scratch allocation failed");
  std::vector<int> hm(n_img, mode);
  hipMemcpyAsync(modes, hm.data(), sizeof(int) * n_img, hipMemcpyHostToDevice, ctx->stream);
  hipStreamSynchronize(ctx->stream);  // hm is a stack-lifetime host buffer
  launch_gftt(ctx->stream, img_plain(d_img), w, h, w, (size_t)w * h, n_img, sc, nullptr, f_para[4], nullptr, maxc,
              (double)(int)f_para[5], corners, ncorners, maxc, nullptr);
  float* sorted_xy = (float*)ctx->scratch("dem_sorted", sizeof(float) * 2 * (size_t)maxc * n_img);
  int* region_off = (int*)ctx->scratch("dem_roff", sizeof(int) * 17 * (size_t)n_img);
  if (!sorted_xy || !region_off) return ctx->fail(FLVIS_ERR_HIP, "feature_dem: scratch allocation failed");
  launch_feature_dem_prep(ctx->stream, img_plain(d_img), w, h, w, (size_t)w * h, n_img, dem_params(w, h, f_para), corners, ncorners,
                          maxc, nullptr, sorted_xy, region_off);
  launch_feature_dem(ctx->stream, w, h, n_img, dem_params(w, h, f_para), sorted_xy, region_off, maxc, modes, d_exist_xy,
                     d_exist_count, exist_cap, d_out_xy, d_out_count, out_cap);
  CHECK_LAUNCH(ctx, "feature_dem");
  return FLVIS_OK;
}

int flvis_hip_feature_dem_detect(flvis_ctx* ctx, const uint8_t* d_img, int w, int h, int n_img, const double* f_para,
                                 float* d_out_xy, int* d_out_count, int out_cap) {
  CHECK_CTX(ctx);
  return dem_common(ctx, d_img, w, h, n_img, f_para, 1, nullptr, nullptr, 0, d_out_xy, d_out_count, out_cap);
}

int flvis_hip_feature_dem_redetect(flvis_ctx* ctx, const uint8_t* d_img, int w, int h, int n_img, const double* f_para,
                                   const double* d_exist_xy, const int* d_exist_count, int exist_cap, float* d_out_xy,
                                   int* d_out_count, int out_cap) {
  CHECK_CTX(ctx);
  if (!d_exist_xy || !d_exist_count || exist_cap <= 0) return ctx->fail(FLVIS_ERR_INVALID_ARG, "feature_dem_redetect: bad args");
  return dem_common(ctx, d_img, w, h, n_img, f_para, 2, d_exist_xy, d_exist_count, exist_cap, d_out_xy, d_out_count,
                    out_cap);
}

// Test aid: the corner-response pass of goodFeaturesToTrack alone, with the chosen kernel variant (0 LDS tiles, 1 strip-mined tiles, 2 wave
// walk with `rows` rows per chunk): per image the ordered bits of the maximum response, the number of 3x3 local maxima and their sort keys
// (unsorted; key = ~((ordered(response) << 32) | pixel offset)) in h_keys [n_img][key_cap].
int flvis_hip_debug_corner_response(flvis_ctx* ctx, const uint8_t* d_img, int w, int h, int n_img, int variant, int rows, uint32_t* h_max_bits,
                                    int* h_nkeys, uint64_t* h_keys, int key_cap) {
  CHECK_CTX(ctx);
  if (!d_img || !h_max_bits || !h_nkeys || !h_keys || w < 8 || h < 8 || (w & 3) || n_img <= 0 || variant < 0 || variant > 2 || key_cap <= 0)
    return ctx->fail(FLVIS_ERR_INVALID_ARG, "corner_response: bad args");
  GfttScratch sc;
  int rc = gftt_scratch(ctx, w, h, n_img, sc);
  if (rc) return rc;
  hipMemsetAsync(sc.maxenc, 0, sizeof(unsigned) * n_img, ctx->stream);
  hipMemsetAsync(sc.nkeys, 0, sizeof(int) * n_img, ctx->stream);
  launch_corner_response(ctx->stream, variant, rows, img_plain(d_img), w, h, w, (size_t)w * h, n_img, sc.maxenc, sc.keys, sc.nkeys, sc.cap, nullptr);
  hipError_t e = hipMemcpyAsync(h_max_bits, sc.maxenc, sizeof(unsigned) * n_img, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(h_nkeys, sc.nkeys, sizeof(int) * n_img, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  for (int i = 0; i < n_img && e == hipSuccess; i++) {
    const int n = std::min(std::min(h_nkeys[i], sc.cap), key_cap);
    e = hipMemcpy(h_keys + (size_t)i * key_cap, sc.keys + (size_t)i * sc.cap, sizeof(uint64_t) * n, hipMemcpyDeviceToHost);
  }
  if (e != hipSuccess) return ctx->hip_fail(e, "corner_response");
  return FLVIS_OK;
}

// Test aid for the corner-response kernel's square root (eig_walk.hip: ew_sqrt_pos against the compiler's correctly rounded sqrtf)
// on every float with bit pattern in [first_bits, first_bits + n): the number of arguments on which they differ.
int flvis_hip_debug_sqrt_check(flvis_ctx* ctx, uint32_t first_bits, uint32_t n, uint64_t* h_mismatches) {
  CHECK_CTX(ctx);
  if (!h_mismatches) return ctx->fail(FLVIS_ERR_INVALID_ARG, "sqrt_check: bad args");
  unsigned long long* d = (unsigned long long*)ctx->scratch("sqrt_check", sizeof(unsigned long long));
  if (!d) return ctx->fail(FLVIS_ERR_HIP, "sqrt_check: scratch allocation failed");
  hipMemsetAsync(d, 0, sizeof(unsigned long long), ctx->stream);
  launch_sqrt_check(ctx->stream, first_bits, n, d);
  unsigned long long h = 0;
  hipError_t e = hipMemcpyAsync(&h, d, sizeof(h), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) return ctx->hip_fail(e, "sqrt_check");
  *h_mismatches = h;
  return FLVIS_OK;
}

}  // extern "C"
