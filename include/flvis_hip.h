/* flvis_hip.h -- C ABI of the MI355X-native FLVIS hot path (libflvis_hip.so).
 *
 * Drop-in boundary for the reference's front-end tracking + local-map BA path.  Plain pointers and
 * sizes only; every function returns 0 on success or a negative flvis_status; the message of the last failure on a
 * context is available from flvis_last_error().  Nothing throws across this boundary.  The library owns all device
 * memory it allocates; pointers named d_* are DEVICE pointers supplied by the caller (HBM-resident inputs/outputs),
 * pointers named h_* are host pointers.
 *
 * Kernel-level entry points (what a maintainer binds at the reference's OpenCV call sites):
 *   flvis_hip_equalize_hist      <- cv::equalizeHist            src/frontend/f2f_tracking.cpp:127,143-144
 *   flvis_hip_pyr_down           <- pyramid level of cv::calcOpticalFlowPyrLK (buildOpticalFlowPyramid)
 *   flvis_hip_lk_track           <- cv::calcOpticalFlowPyrLK    src/processing/lkorb_tracking.cpp:64-73,
 *                                                               src/processing/camera_frame.cpp:124-128
 *   flvis_hip_gftt               <- cv::goodFeaturesToTrack     src/processing/feature_dem.cpp:160,221
 *   flvis_hip_feature_dem_detect / _redetect <- FeatureDEM::detect / ::redetect   src/processing/feature_dem.cpp:124-266
 *                                              (include/feature_dem.h:40-50)
 */
#ifndef FLVIS_HIP_H
#define FLVIS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct flvis_ctx flvis_ctx;

typedef enum flvis_status {
  FLVIS_OK = 0,
  FLVIS_ERR_INVALID_ARG = -1,
  FLVIS_ERR_NO_DEVICE = -2,   /* no HIP device / kernels cannot run: the product path never falls back to the CPU */
  FLVIS_ERR_HIP = -3,         /* a HIP runtime call failed (message in flvis_last_error) */
  FLVIS_ERR_CAPACITY = -4,    /* a fixed capacity (points per stream, candidates, window) would be exceeded */
  FLVIS_ERR_CONFIG = -5       /* yaml/config problem */
} flvis_status;

/* Library version string, e.g. "flvis_hip 0.1 (gfx950)". */
const char* flvis_version(void);

/* Creates a context bound to HIP device `device`.  `hip_stream` is an existing hipStream_t (e.g. torch's current
 * stream); NULL means the device's default (null) stream exactly as in HIP; FLVIS_STREAM_NEW asks the library to
 * create a private non-blocking stream.  Fails with FLVIS_ERR_NO_DEVICE when no GPU is visible. */
#define FLVIS_STREAM_NEW ((void*)(intptr_t)-1)
int flvis_hip_create(int device, void* hip_stream, flvis_ctx** out);
void flvis_hip_destroy(flvis_ctx* ctx);
const char* flvis_last_error(const flvis_ctx* ctx);
/* Blocks until all work queued on the context's stream has finished. */
int flvis_hip_synchronize(flvis_ctx* ctx);
/* The hipStream_t the context launches on (for event timing by the caller). */
void* flvis_hip_stream(flvis_ctx* ctx);

/* ---- kernel-level entry points; all operate on a batch of n_img independent images (one per stream) ------------ */

/* cv::equalizeHist on n_img contiguous u8 images of w x h (pitch == w, w % 4 == 0). In place allowed. */
int flvis_hip_equalize_hist(flvis_ctx* ctx, const uint8_t* d_src, uint8_t* d_dst, int w, int h, int n_img);

/* cv::pyrDown (5-tap Gaussian, REFLECT_101, (x+128)>>8): n_img images w x h (pitch src_pitch) ->
 * ((w+1)/2) x ((h+1)/2) images with pitch dst_pitch.  Image i starts at d_src + i*src_pitch*h (resp. dst). */
int flvis_hip_pyr_down(flvis_ctx* ctx, const uint8_t* d_src, int w, int h, int src_pitch, uint8_t* d_dst,
                       int dst_pitch, int n_img);

/* cv::calcOpticalFlowPyrLK(prev, next, prevPts, nextPts, status, err, Size(31,31), max_level,
 *                          TermCriteria(COUNT+EPS, max_iter, eps), use_initial_flow ? OPTFLOW_USE_INITIAL_FLOW : 0)
 * for n_img image pairs at once.  d_prev/d_next: [n_img][h][w] u8.  Points: d_prev_pts/d_next_pts [n_img][nmax][2]
 * float (x,y); d_next_pts is in/out; d_status [n_img][nmax] u8; d_count [n_img] int = valid points per image.
 * Pyramids are built internally (as OpenCV does for raw Mats). */
int flvis_hip_lk_track(flvis_ctx* ctx, const uint8_t* d_prev, const uint8_t* d_next, int w, int h, int n_img,
                       const float* d_prev_pts, float* d_next_pts, uint8_t* d_status, const int* d_count, int nmax,
                       int max_level, int max_iter, double eps, int use_initial_flow);

/* cv::goodFeaturesToTrack(img, corners, max_corners, quality, min_distance) (blockSize 3, min-eigenvalue).
 * d_out_xy [n_img][max_corners][2] float, d_out_count [n_img] int. */
int flvis_hip_gftt(flvis_ctx* ctx, const uint8_t* d_img, int w, int h, int n_img, int max_corners, double quality,
                   double min_distance, float* d_out_xy, int* d_out_count);

/* FeatureDEM::detect(img, newPts) with FeatureDEM(w, h, f_para) (f_para = feature_para1..6 of the yaml).
 * d_out_xy [n_img][out_cap][2] float, d_out_count [n_img]. */
int flvis_hip_feature_dem_detect(flvis_ctx* ctx, const uint8_t* d_img, int w, int h, int n_img, const double* f_para,
                                 float* d_out_xy, int* d_out_count, int out_cap);

/* FeatureDEM::redetect(img, existedPts, newPts, n): d_exist_xy [n_img][exist_cap][2] double, d_exist_count [n_img]. */
int flvis_hip_feature_dem_redetect(flvis_ctx* ctx, const uint8_t* d_img, int w, int h, int n_img, const double* f_para,
                                   const double* d_exist_xy, const int* d_exist_count, int exist_cap, float* d_out_xy,
                                   int* d_out_count, int out_cap);

#ifdef __cplusplus
}
#endif
#endif /* FLVIS_HIP_H */
