// flvis_amd: host-visible declarations for the image / LK kernels (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace flvis {

// Joins folded into a launch (round 6, pipeline.cpp: FLVIS_JOIN_FOLD): the kernel's workgroups wait for up to two device words before their
// first instruction that needs another stream's results (only kernels of one workgroup per stream do: 64 sleeping lanes cannot keep a
// producer off the chip), and the last workgroup to finish stores a sequence number for the streams that wait for this kernel -- instead
// of a k_wait_flag / k_store_flag launch (~5 us each on the frame's chain) in front of / behind it.
struct KJoin {
  const long long* wait[2];
  long long wait_seq[2];
  long long* sig;
  long long sig_seq;
  unsigned* sig_cnt;  // arrival counter of the signalling launch's workgroups (the last one puts it back to 0)
  long long* err;     // host-mapped word that receives the sequence number of a wait that gave up (~4 s), or null
  // a word the kernel waits for BEHIND its work (thread 0 of every workgroup, loads only): the kernel does not end before another
  // stream's result is there, and the launch that follows it on its stream -- an LK launch, which cannot carry a wait itself: thousands of
  // sleeping workgroups would keep a producer off the chip -- needs no k_wait_flag in front.  The next kernel's own acquire orders the data.
  const long long* post;
  long long post_seq;
};


// Selects, per stream, one of two image slots (frame ping-pong: reference F2FTracking swaps last_frame/curr_frame,
// src/frontend/f2f_tracking.cpp:70; the swap is per stream here because streams fail/recover independently).
struct ImgSel {
  const uint8_t* b[2];
  const int* cur;  // device array [S] with the stream's current slot, or nullptr (always slot 0)
  int flip;        // 0 -> current slot, 1 -> the other one (last frame)
  // when set: the base address is read from this device-side table entry instead of b[] -- for the caller's input images,
  // whose address changes from frame to frame while the kernel arguments (a captured graph's nodes) stay the same
  const uint8_t* const* ind;
  __device__ const uint8_t* ptr(int s, size_t stride) const {
    if (ind) return *ind + (size_t)s * stride;
    int k = cur ? (cur[s] ^ flip) : 0;
    return b[k] + (size_t)s * stride;
  }
};
static inline ImgSel img_plain(const uint8_t* p) { return ImgSel{{p, p}, nullptr, 0, nullptr}; }
static inline ImgSel img_indirect(const uint8_t* const* slot) { return ImgSel{{nullptr, nullptr}, nullptr, 0, slot}; }

constexpr int LK_MAX_LEVELS = 6;
struct PyrSel {
  ImgSel lvl[LK_MAX_LEVELS];
  int w[LK_MAX_LEVELS], h[LK_MAX_LEVELS], pitch[LK_MAX_LEVELS];
  size_t stride[LK_MAX_LEVELS];  // bytes between consecutive streams
  int levels;                    // highest level index (0..levels)
  // physical BORDER_REFLECT_101 border around each level (0: none): columns -bx .. w + bx - 1 and rows -by .. h + by - 1 of a
  // level are addressable and hold img(reflect101c(y), reflect101c(x)) once launch_pyr_border has run
  int bx[LK_MAX_LEVELS] = {}, by[LK_MAX_LEVELS] = {};
};
// Border of the tracker's pyramids.  A patch / search region of a point that lies inside the image reaches at most 23 / 32 columns
// and 17 / 21 rows over the edge (lk_kernel.hip); wider excursions (a point tracked out of the image) take the kernel's slow path.
constexpr int LK_BORDER_X = 32, LK_BORDER_Y = 24;

struct LKParams {
  int max_iter;     // clamped to [0,100] like cv::calcOpticalFlowPyrLK
  double eps2;      // epsilon^2
  float min_eig;    // minEigThreshold (1e-4)
  int use_initial;  // OPTFLOW_USE_INITIAL_FLOW
  // optional statistics (nullptr: none): stats[2 l] += Gauss-Newton iterations run at level l, stats[2 l + 1] += points that iterated there
  unsigned long long* stats = nullptr;
  // ... and (nullptr: none) stats_tc[0] += templates taken from the cache, [1] += template patches and [2] += search regions staged by
  // the slow (index-reflecting) path
  unsigned long long* stats_tc = nullptr;
  // Template cache (lk_kernel.hip).  The stereo matcher of frame t computes, for every landmark, the template -- interpolated I, Ix, Iy
  // of the 31 x 31 window on every level + the Hessian sums -- at the landmark's pixel in the left image of frame t; the temporal
  // tracker of frame t + 1 needs exactly that template (previous image = that image, previous point = that pixel).  tc_mode 1: every
  // point p < tc_cap stores its templates in slot p, with the position bits and tc_tag[s] in the slot's header.  tc_mode 2: the caller
  // has compared the headers already (lk_tc_lookup): tc_slot[s * nmax + p] = slot | (mask of the levels stored << 16) if the slot was
  // written for this very position of the image tagged tc_tag[s], else -1 (the point computes its templates).  tc_mode 3 (the stereo
  // launch behind launch_lk_templates_ahead, role 4): tc_slot as in mode 2, or -(slot + 2): the point computes its templates and stores
  // them in that slot.
  // HBM capacity and bandwidth (idle on this path) spent to save the VALU work that bounds the kernel.
  uint32_t* tc = nullptr;             // [S][tc_cap][tc_stride]
  int tc_mode = 0, tc_cap = 0, tc_stride = 0;
  const int* tc_slot = nullptr;       // [S][nmax]
  const long long* tc_tag = nullptr;  // [S] identity of the template image (the stream's frame id)
  int order = 0;                      // dispatch order of a stream's points: 0 first to last, 1 last to first (speed only)
  int dbg_slot = 0;                   // (-DFLVIS_LK_UTIL builds: which of 8 slots the launch's wave times go to; scripts/lk_util.py)
};
// dwords of one template-cache slot for a pyramid with levels 0 .. levels
int lk_tc_slot_dwords(int levels);
// tc_mode 2's per-point code for the point at (px, py) of the image tagged `tag` whose landmark remembers `slot` (device side)
__device__ inline int lk_tc_lookup(const uint32_t* tc, int tc_cap, int tc_stride, int s, int slot, float px, float py, long long tag) {
  if (!tc || slot < 0 || slot >= tc_cap) return -1;
  const uint32_t* h = tc + ((size_t)s * tc_cap + slot) * tc_stride;
  if (h[0] != __float_as_uint(px) || h[1] != __float_as_uint(py) || *reinterpret_cast<const long long*>(h + 2) != tag) return -1;
  return slot | (int)((h[4] & 0x7fffu) << 16);
}

struct DemParams {
  int regionWidth, regionHeight, boundary_dis;
  unsigned max_region_feature_num;
};

struct GfttScratch {
  unsigned* maxenc;          // [S]
  int* nkeys;                // [S]
  unsigned long long* keys;  // [S][cap]
  int cap;                   // power of two
};

hipError_t img_kernels_init();
void launch_equalize_hist(hipStream_t st, ImgSel src, ImgSel dst, int w, int h, int spitch, int dpitch, size_t sstride,
                          size_t dstride, int S, unsigned* hist, uint8_t* lut, const int* active);
void launch_bgr_to_gray(hipStream_t st, const uint8_t* src, int channels, uint8_t* dst, size_t npixels);
void launch_copy_image(hipStream_t st, ImgSel src, ImgSel dst, int w, int h, int spitch, int dpitch, size_t sstride,
                       size_t dstride, int S, const int* active);
// any width / source pitch (rows need not be dword aligned); dpitch % 4 == 0
void launch_copy_image_any(hipStream_t st, ImgSel src, ImgSel dst, int w, int h, int spitch, int dpitch, size_t sstride,
                           size_t dstride, int S, const int* active);
// (bx, by): physical border of the DESTINATION level, written by the same kernel (every produced pixel also goes to the border
// positions that mirror it); 0: none.  Needs pyr_border_fusable(dst w, dst h, bx, by), otherwise pass 0 and use launch_pyr_border.
void launch_pyr_down(hipStream_t st, ImgSel src, int sw, int sh, int spitch, size_t sstride, ImgSel dst, int dpitch,
                     size_t dstride, int S, const int* active, int bx = 0, int by = 0, int sbx = 0, int sby = 0);
// (sbx, sby: physical border of the SOURCE level, complete when the kernel runs: its tiles are then staged without index reflection)
bool pyr_border_fusable(int w, int h, int bx, int by);
// The same levels by walking waves (pyr_walk.hip; needs pyr_walk_ok): ONE launch reads level `first` of a pyramid from `src` (the level
// itself or, with copy0, the caller's image, which is then also stored as pyr.lvl[first] -- the ingest copy) and produces the levels
// first + 1 .. first + nout (nout = 1, 2 or 3) with their physical borders (pyr.bx / by of each stored level; 0: none) complete.
// bx / by of pyr_walk_ok: those of the levels first .. first + nout.
bool pyr_walk_ok(int sw, int sh, int nout, const int* bx, const int* by, bool copy0);
void launch_pyr_walk(hipStream_t st, ImgSel src, int sw, int sh, int spitch, size_t sstride, const PyrSel& pyr, int first, int nout, bool copy0,
                     int S, const int* active);
// fills the border of the levels in level_mask that have one (one launch): for levels whose producer did not write it
void launch_pyr_border(hipStream_t st, const PyrSel& pyr, int S, const int* active, unsigned level_mask = ~0u);
// level 1 of a pyramid fused with the ingest copy: reads the caller's image once, writes level 0 (dst0) and level 1 (dst)
void launch_pyr_down_ingest(hipStream_t st, ImgSel src, int sw, int sh, int spitch, size_t sstride, ImgSel dst0, int d0pitch,
                            size_t d0stride, ImgSel dst, int dpitch, size_t dstride, int S, const int* active, int bx = 0, int by = 0);
void launch_gftt(hipStream_t st, ImgSel src, int w, int h, int pitch, size_t sstride, int S, GfttScratch sc,
                 const double* qual_s, double quality, const int* maxc_s, int max_corners, double min_distance,
                 float* out_xy, int* out_n, int out_cap, const int* active, hipEvent_t* stage_events = nullptr,
                 bool reset_counters = true);
// the corner-response pass of launch_gftt as a wave walk (eig_walk.hip): fills maxenc / keys / nkeys like k_eig_cand
void launch_eig_walk(hipStream_t st, ImgSel src, int w, int h, int pitch, size_t sstride, int S, unsigned* maxenc, unsigned long long* keys,
                     int* nkeys, int cap, const int* active, int rows_per_chunk);
void launch_corner_response(hipStream_t st, int variant, int rows, ImgSel src, int w, int h, int pitch, size_t sstride, int S, unsigned* maxenc,
                            unsigned long long* keys, int* nkeys, int cap, const int* active);
void launch_sqrt_check(hipStream_t st, unsigned first_bits, unsigned n, unsigned long long* mismatches);  // test aid, eig_walk.hip
// FeatureDEM in two launches: what depends on the corners and the image only (regions, Harris scores, per-region order) -> sorted_xy
// [S][corner_cap][2], region_off [S][17]; then the part that needs the existing landmarks (fill, greedy spacing, output)
void launch_feature_dem_prep(hipStream_t st, ImgSel src, int w, int h, int pitch, size_t sstride, int S, DemParams prm,
                             const float* corners, const int* ncorners, int corner_cap, const int* active, float* sorted_xy,
                             int* region_off, const KJoin* kj = nullptr);
// (kj, may be null: joins folded into the launch -- k_feature_dem waits, k_feature_dem_prep signals)
void launch_feature_dem(hipStream_t st, int w, int h, int S, DemParams prm, const float* sorted_xy, const int* region_off,
                        int corner_cap, const int* mode, const double* exist_xy, const int* nexist, int exist_cap, float* out_xy,
                        int* out_n, int out_cap, const KJoin* kj = nullptr);
// pyramidal LK, 31x31 window: one wave per (stream, point)
// max_pts: upper bound of count[] known to the caller (sizes the grid; any value is correct, the kernel strides), <= 0: nmax
// role: 0 stand-alone call, 1 the tracker's temporal launch, 2 its stereo launch (kernel instances k_lk_track<role>: named apart in
// the profiles; <1> reads and <2> writes the template cache), 4 the stereo launch that takes the templates launch_lk_templates_ahead
// made and stores those of the other points (tc_mode 3)
void launch_lk_track(hipStream_t st, const PyrSel& prev, const PyrSel& next, const float* prev_pts, float* next_pts,
                     uint8_t* status, const int* count, int nmax, int S, LKParams prm, const int* active, int max_pts = 0, int role = 0);
// the templates of the points pts[s][0 .. count[s]) of the pyramid `img`, on every level, into the cache slots 0 .. count[s] - 1 of stream s
// with (position bits, tag[s], mask of the levels stored) in the header: what a later LK launch with tc_mode 2 / 3 finds there
void launch_lk_templates_ahead(hipStream_t st, const PyrSel& img, const float* pts, const int* count, int nmax, int S, uint32_t* tc, int tc_cap,
                               int tc_stride, const long long* tag, int max_pts);

// recover3DPts_c_FromStereo on caller arrays (stereo_depth.hip): the rig constants the two kernels need, and their launchers
struct flvis_sd_cam {
  double K1[4], D1[4], R1[9], P0[12], P1[12], T_c1_c0[7];
  double fx, fy, cx, cy;
};
void launch_stereo_depth_seeds(hipStream_t st, const flvis_sd_cam& c, const float* pt2d_plane, const float* pt3d_w, const uint8_t* has_depth,
                               const int* count, int cap, int n_sets, const double* d_T_c_w7, float* seeds);
void launch_stereo_depth_post(hipStream_t st, const flvis_sd_cam& c, const float* pt2d_undistort, const float* matched, const uint8_t* status,
                              const int* count, int cap, int n_sets, float range, int* rand_state35, double* pt3d_c, uint8_t* mask);

}  // namespace flvis
