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
 *   flvis_hip_stereo_depth       <- CameraFrame::recover3DPts_c_FromStereo   src/processing/camera_frame.cpp:93-180
 *   flvis_hip_lk_track           <- cv::calcOpticalFlowPyrLK    src/processing/lkorb_tracking.cpp:64-73,
 *                                                               src/processing/camera_frame.cpp:124-128
 *   flvis_hip_gftt               <- cv::goodFeaturesToTrack     src/processing/feature_dem.cpp:160,221
 *   flvis_hip_feature_dem_detect / _redetect <- FeatureDEM::detect / ::redetect   src/processing/feature_dem.cpp:124-266
 *                                              (include/feature_dem.h:40-50)
 *   flvis_hip_orb_detect_and_compute / _hamming_knn2 / _orb_match  <- cv::ORB, cv::BFMatcher   src/backend/vo_loopclosing.cpp:242-243,601-639
 *   flvis_hip_bow_load_vocabulary / _set_vocabulary / _transform / _score / _score_jobs, flvis_loop_candidate
 *                                <- DBoW3::Vocabulary(file), ::transform, ::score; isLoopCandidate   vo_loopclosing.cpp:1097,249-253,417-437,520-590
 *   flvis_hip_lc_keyframe_landmarks  <- stereo LK + triangulation / depth lookup of the ORB keypoints   vo_loopclosing.cpp:255-372
 *   flvis_hip_pnp_ransac         <- cv::solvePnPRansac of isLoopClosureKF                             vo_loopclosing.cpp:660-686
 *   flvis_hip_pgo_loop_closure   <- loopClosureOnCovGraphG2ONew (g2o EdgeSE3 pose graph)              vo_loopclosing.cpp:742-944
 * Pipeline-level entry points (the nodelets' work for a batch of streams / sequences):
 *   flvis_config_load, flvis_tracker_create, flvis_imu_feed, flvis_image_feed(_host), flvis_get_*   <- TrackingNodeletClass / F2FTracking
 *   flvis_ba_push_keyframe, flvis_get_correction, flvis_correction_feed                               <- LocalMapNodeletClass
 *   flvis_lc_params_load, flvis_loop_closer_*                                                          <- LoopClosingNodeletClass
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
/* cv::cvtColor(img, gray, CV_BGR2GRAY / CV_BGRA2GRAY) as F2FTracking::image_feed applies it to 3- or 4-channel input
 * (src/frontend/f2f_tracking.cpp:74-111; mbRGB is constant 0 there): d_src [n_img][h][w][channels] interleaved,
 * d_dst [n_img][h][w].  w % 4 == 0. */
int flvis_hip_cvt_bgr_to_gray(flvis_ctx* ctx, const uint8_t* d_src, int channels, uint8_t* d_dst, int w, int h, int n_img);

/* cv::pyrDown (5-tap Gaussian, REFLECT_101, (x+128)>>8): n_img images w x h (pitch src_pitch) ->
 * ((w+1)/2) x ((h+1)/2) images with pitch dst_pitch.  Image i starts at d_src + i*src_pitch*h (resp. dst). */
int flvis_hip_pyr_down(flvis_ctx* ctx, const uint8_t* d_src, int w, int h, int src_pitch, uint8_t* d_dst,
                       int dst_pitch, int n_img);

/* cv::calcOpticalFlowPyrLK(prev, next, prevPts, nextPts, status, err, Size(31,31), max_level,
 *                          TermCriteria(COUNT+EPS, max_iter, eps), use_initial_flow ? OPTFLOW_USE_INITIAL_FLOW : 0)
 * for n_img image pairs at once.  d_prev/d_next: [n_img][h][w] u8, tightly packed, any w >= 32 (rows that are not dword aligned, e.g.
 * KITTI's 1241, are first copied into a pitch-aligned level 0).  Points: d_prev_pts/d_next_pts [n_img][nmax][2]
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

/* ---- ORB extraction + Hamming matching: the keyframe-rate front half of the reference's loop closing.
 * Replaces cv::ORB::create(nfeatures, scaleFactor, nlevels, 31, 0, 2, cv::ORB::HARRIS_SCORE, 31, fastThreshold)
 *            ->detectAndCompute(img0, cv::Mat(), keypoints, descriptors)        src/backend/vo_loopclosing.cpp:242-243
 * (edgeThreshold 31, firstLevel 0, WTA_K 2, HARRIS_SCORE, patchSize 31 are the only values the reference uses and are
 * fixed here).  Batched over n_img images of one size.  Keypoints come out level-major, raster order within a level
 * (OpenCV's order is an artefact of std::nth_element); d_kps rows are (x, y, size, angle_deg, response, octave).
 * h_pattern: host table of the 256 test pairs as int8 [512][2] (x, y), e.g. OpenCV's learned bit_pattern_31_ when the
 * descriptors must be compatible with a DBoW vocabulary; NULL selects OpenCV's generator makeRandomPattern(31, ., 512)
 * (flvis_orb_default_pattern).  d_overflow (optional, [n_img]) is set non-zero where a capacity truncated the result. */
typedef struct flvis_orb_params {
  int nfeatures;       /* 1000 in the reference */
  float scale_factor;  /* 1.2f */
  int nlevels;         /* 8 (<= 12) */
  int fast_threshold;  /* 20 */
} flvis_orb_params;
int flvis_hip_orb_detect_and_compute(flvis_ctx* ctx, const uint8_t* d_img, int w, int h, int n_img,
                                     const flvis_orb_params* prm, const int8_t* h_pattern, float* d_kps, uint8_t* d_desc,
                                     int* d_count, int cap, int* d_overflow);
int flvis_orb_default_pattern(int8_t* h_pattern512x2);
/* building blocks at their OpenCV call shapes (device pointers, [n_img] images, tightly packed):
 * cv::resize(src, dst, Size(dw, dh), 0, 0, INTER_LINEAR); the FAST-9/16 score map (0 = not a corner) behind
 * cv::FAST(img, kps, threshold, true); cv::GaussianBlur(src, dst, Size(7,7), 2, 2, BORDER_REFLECT_101). */
int flvis_hip_resize_linear(flvis_ctx* ctx, const uint8_t* d_src, int sw, int sh, uint8_t* d_dst, int dw, int dh, int n_img);
int flvis_hip_fast_score(flvis_ctx* ctx, const uint8_t* d_img, int w, int h, int n_img, int threshold, uint8_t* d_score);
int flvis_hip_gaussian_blur7(flvis_ctx* ctx, const uint8_t* d_src, uint8_t* d_dst, int w, int h, int n_img);
/* cv::BFMatcher(cv::NORM_HAMMING, false).knnMatch(query, train, matches, 2) (vo_loopclosing.cpp:601-613) for n_pairs
 * independent (query, train) sets of 32-byte descriptors: d_query [n_pairs][qcap][32], d_nq [n_pairs], likewise train.
 * d_idx / d_dist: [n_pairs][qcap][2], ascending distance, ties keep the lower train index; -1 / INT_MAX when absent. */
int flvis_hip_hamming_knn2(flvis_ctx* ctx, const uint8_t* d_query, const int* d_nq, int qcap, const uint8_t* d_train,
                           const int* d_nt, int tcap, int n_pairs, int* d_idx, int* d_dist);
/* the mutual-best + ratio test of vo_loopclosing.cpp:603-639 (knnMatch both ways, keep i when the best match of its best
 * match is i and d0/d1 < ratio_max), pairs (index in a, index in b) in ascending a order: d_pairs [n_pairs][acap][2]. */
int flvis_hip_orb_match(flvis_ctx* ctx, const uint8_t* d_a, const int* d_na, int acap, const uint8_t* d_b, const int* d_nb,
                        int bcap, int n_pairs, double ratio_max, int* d_pairs, int* d_npairs);

/* ---- place recognition of the loop closing: DBoW3 bag of words + L1 score + candidate selection ---------------
 * The vocabulary is handed over as flat arrays (the reference loads a DBoW3 file that is not shipped with it,
 * vo_loopclosing.cpp:1097): node 0 is the root, the children of node n are h_child_idx[h_child_ptr[n] .. h_child_ptr[n+1]) in
 * DBoW3's order (3rdPartLib/DBow3/src/Vocabulary.h m_nodes[n].children), a node without children is a word with id h_word_id[n]
 * and weight h_weight[n] (idf); h_desc: 32 bytes per node.  Weighting TF_IDF, scoring L1_NORM (DBoW3's defaults). */
int flvis_hip_bow_set_vocabulary(flvis_ctx* ctx, int n_nodes, const int* h_child_ptr, const int* h_child_idx, const uint8_t* h_desc,
                                 const double* h_weight, const int* h_word_id);
/* `Vocabulary vocTmp(vocFile)` (vo_loopclosing.cpp:1095-1099; Vocabulary::load, 3rdPartLib/DBow3/src/Vocabulary.cpp:1082-1096):
 * reads a DBoW3 vocabulary file -- binary .dbow3 (plain or QuickLZ-compressed, Vocabulary.cpp:1335-1407), ORB-SLAM2 style .txt
 * (:1259-1332) or OpenCV FileStorage yaml / yaml.gz (:1411-1462) -- and makes it the context's vocabulary.  ORB vocabularies
 * (32-byte CV_8U descriptors) with weighting TF_IDF or TF and scoring L1_NORM; anything else fails with FLVIS_ERR_CONFIG. */
int flvis_hip_bow_load_vocabulary(flvis_ctx* ctx, const char* path);
/* the file reader on its own (host only, no device needed): the flat arrays flvis_hip_bow_set_vocabulary takes, owned by the
 * handle until flvis_voc_file_close.  info8 = {n_nodes, n_words, k, L, scoringType, weightingType, n_edges, layout}, layout
 * 0 binary, 1 binary QuickLZ, 2 text, 3 yaml.  word_id is -1 on inner nodes. */
typedef struct flvis_voc_file flvis_voc_file;
int flvis_voc_file_open(const char* path, flvis_voc_file** out, char* err, int errlen);
int flvis_voc_file_info(const flvis_voc_file* voc, int* info8);
int flvis_voc_file_arrays(const flvis_voc_file* voc, const int** child_ptr, const int** child_idx, const uint8_t** desc,
                          const double** weight, const int** word_id);
void flvis_voc_file_close(flvis_voc_file* voc);
/* voc.transform(kf.lm_descriptor, kf_bv) (vo_loopclosing.cpp:249-253; Vocabulary.cpp:628-688) for n_img keyframes:
 * d_desc [n_img][dcap][32] + d_count [n_img] as flvis_hip_orb_detect_and_compute leaves them (dcap <= 2048);
 * out: d_ids / d_vals [n_img][vcap] ascending word ids and L1-normalised values, d_nnz [n_img]; vcap >= min(dcap, words of the
 * vocabulary), so that no vector is ever truncated. */
int flvis_hip_bow_transform(flvis_ctx* ctx, const uint8_t* d_desc, const int* d_count, int dcap, int n_img, int vcap, int* d_ids,
                            double* d_vals, int* d_nnz);
/* STEP 1.5 / 1.6 of the loop-closing keyframe (vo_loopclosing.cpp:255-372) for n_img keyframes: the 3-D position of every ORB
 * keypoint and the lists without the keypoints that have none.  cam_type as flvis_cfg.cam_type: 0 STEREO_RECT -- d_img0 / d_img1
 * [n_img][h][w] mono8, calcOpticalFlowPyrLK(Size(31,31), 5, 30 / 0.001, USE_INITIAL_FLOW) from the keypoints into img1 (:274-278),
 * then Triangulation::trignaulationPtFromStereo with the rectified 3x4 projections h_P0 / h_P1 (row-major; range 100);
 * 2 DEPTH_D435 -- d_img1 [n_img][h][w] Z16, d = Z16 / 1000 as the INTEGER division the reference writes (:331), kept when
 * 0.3 <= d <= 10, back-projected with h_K4 = fx, fy, cx, cy (d_img0 unused); 1 STEREO_UNRECT -- the reference's case is empty,
 * every count is 0.  d_kps [n_img][cap][6] / d_desc [n_img][cap][32] / d_count as flvis_hip_orb_detect_and_compute leaves them
 * (cap <= 2048).  Out, order kept: d_lm_2d [n_img][cap][2] float, d_lm_3d [n_img][cap][3] double (camera frame), d_lm_desc
 * [n_img][cap][32] (may alias d_desc), d_lm_count [n_img]. */
int flvis_hip_lc_keyframe_landmarks(flvis_ctx* ctx, const uint8_t* d_img0, const void* d_img1, int w, int h, int n_img, int cam_type,
                                    const double* h_P0, const double* h_P1, const double* h_K4, const float* d_kps, const uint8_t* d_desc,
                                    const int* d_count, int cap, float* d_lm_2d, double* d_lm_3d, uint8_t* d_lm_desc, int* d_lm_count);
/* one row of the similarity matrix (vo_loopclosing.cpp:417-437): voc.score(query, db[j]) for j < n_db (ScoringObject.cpp:23-68);
 * the query is one vector on the device (d_q_nnz[0] entries), the database [n_db][vcap]; d_db_nnz[j] < 0 marks an absent keyframe
 * (kf_lc_tmp[j] == nullptr: score 0). */
int flvis_hip_bow_score(flvis_ctx* ctx, const int* d_q_ids, const double* d_q_vals, const int* d_q_nnz, const int* d_db_ids,
                        const double* d_db_vals, const int* d_db_nnz, int vcap, int n_db, double* d_scores);
/* several rows in one launch over ONE store of vectors (d_ids / d_vals [n_vectors][vcap], d_nnz [n_vectors]): job i = h_jobs3[3i..3i+2]
 * = (query vector, first database vector, number of database vectors); d_scores[first + j] = score(query, first + j).  The loop
 * closer's rows of all sequences that got a keyframe. */
int flvis_hip_bow_score_jobs(flvis_ctx* ctx, int n_jobs, const int* h_jobs3, const int* d_ids, const double* d_vals, const int* d_nnz, int vcap,
                             double* d_scores);
/* isLoopCandidate (vo_loopclosing.cpp:520-590) on the newest keyframe's row h_row[i] = sim_matrix[i][g_size-1] (host control
 * logic, as in the reference's pgoProcess thread).  Returns 1 and *kf_prev_idx when there is a candidate, 0 when not. */
int flvis_loop_candidate(int g_size, const double* h_row, const uint8_t* h_present, int lcKFDist, int lcKFMaxDist, int lcNKFClosest,
                         double minScore, int64_t* kf_prev_idx);
/* the geometric check of isLoopClosureKF (vo_loopclosing.cpp:660-686): cv::solvePnPRansac(p3d, p2d, K, Mat(), r, t, false,
 * iterations = 100, reprojectionError = 2.0, confidence = 0.99, inliers, SOLVEPNP_P3P) for n_sets independent correspondence sets
 * (one workgroup each): d_p3d [n_sets][cap][3] / d_p2d [n_sets][cap][2] float (what the reference casts to), d_count [n_sets]
 * (cap <= 1024), h_K4 = fx fy cx cy, h_seeds: one 64-bit seed of the sample generator per set.  Out: d_pose7 [n_sets][7]
 * (tx ty tz qx qy qz qw of T_c_w; identity when no model was found), d_inlier_mask [n_sets][cap], d_n_inliers [n_sets] (the caller
 * applies the acceptance rule of :677-686).  The tracker's solver on caller arrays, cv::solvePnPRansac's structure: 4-point subsets from
 * cv::RNG((uint64)-1) in getSubset's order (h_seeds is accepted and ignored: OpenCV seeds every run the same), P3P hypotheses (three
 * points, the fourth picks among the solutions), RANSACUpdateNumIters, and the final solvePnP(SOLVEPNP_EPNP) on the inliers (DESIGN.md
 * section 2 lists what is restated and what cannot be bitwise OpenCV's). */
int flvis_hip_pnp_ransac(flvis_ctx* ctx, const float* d_p3d, const float* d_p2d, const int* d_count, int cap, int n_sets, const double* h_K4,
                         int iterations, double reproj_px, double confidence, const uint64_t* h_seeds, double* d_pose7,
                         uint8_t* d_inlier_mask, int* d_n_inliers);
/* Test hook: cv::solvePnP(..., SOLVEPNP_EPNP) alone (the solver inside flvis_hip_pnp_ransac and the tracker's PnP RANSAC) on n_sets
 * correspondence sets laid out as for flvis_hip_pnp_ransac (d_count[i] >= 4 points each), one wavefront per set.  d_out160 [n_sets][160]:
 * R (9, row-major) t (3) ok (1) | betas of the three approximations (12) | their mean reprojection errors (3) | the four eigenvectors of
 * MtM (48) | L (60) | rho (6) | the eigenvalues of MtM (12) | unused (6). */
int flvis_hip_debug_epnp(flvis_ctx* ctx, const float* d_p3d, const float* d_p2d, const int* d_count, int cap, int n_sets, const double* h_K4,
                         double* d_out160);
/* loopClosureOnCovGraphG2ONew (vo_loopclosing.cpp:742-944) for n_graphs independent sequences in one launch (one workgroup per
 * pose graph): graph g has h_n_kf[g] keyframes with T_c_w (device, 7 doubles each: tx ty tz qx qy qz qw, graphs concatenated, in/out)
 * and presence flags (host, concatenated; 0 = kf_map_lc[i] == nullptr), h_n_loops[g] recorded loops (host ids: earlier, later
 * keyframe; device poses: the verified T_later_earlier the reference keeps in loop_poses).  Vertices kf_prev..kf_curr, EdgeSE3 to
 * the next five keyframes + one per loop, information I, Cauchy kernel, Levenberg with lambda 1e-10, `iterations` (100 in the
 * reference); use_initial_guess = optimizer.computeInitialGuess() (:883).  Out: the optimised keyframes' T_c_w in place,
 * d_drift7 [n_graphs][7] = Tw1_w2 of the last optimised keyframe (:899-910), d_stats5 [n_graphs][5] = iterations run, robust chi2
 * before / after, vertices, edges; h_ran[g] = 1 when graph g was optimised (0: no loop / a loop names an absent keyframe). */
int flvis_hip_pgo_loop_closure(flvis_ctx* ctx, int n_graphs, const int* h_n_kf, double* d_T_c_w7, const uint8_t* h_present,
                               const int* h_n_loops, const int* h_loop_ids, const double* d_loop_pose7, int iterations,
                               int use_initial_guess, double* d_drift7, double* d_stats5, int* h_ran);

/* ---- pipeline-level entry points: F2FTracking + LocalMap for a batch of independent streams ------------------------
 *
 * Mirrors the reference's class surface (include/f2f_tracking.h:51-73, src/backend/vo_localmap.cpp:87-380):
 *   flvis_config_load        <- TrackingNodeletClass::onInit yaml handling      src/frontend/vo_tracking.cpp:103-306
 *                                (accepts the reference's yaml files unchanged; src/utils/include/yamlRead.h)
 *   flvis_tracker_create     <- F2FTracking::init                               src/frontend/f2f_tracking.cpp:5-38
 *   flvis_imu_feed(_out)     <- TrackingNodeletClass::imu_callback + F2FTracking::imu_feed (_out: with its q_w_i / pos_w_i / vel_w_i)
 *   flvis_get_imu_states     <- the same outputs, batched
 *                                                                               src/frontend/vo_tracking.cpp:326-371, f2f_tracking.cpp:46-57
 *   flvis_image_feed         <- F2FTracking::image_feed (+ KeyFrameMsg::pub + LocalMapNodeletClass::frame_callback when
 *                                with_local_map != 0)                           f2f_tracking.cpp:59-400, vo_tracking.cpp:396-430
 *   flvis_get_keyframe       <- flvis/KeyFrame message payload                  msg/KeyFrame.msg, src/utils/keyframe_msg.cpp:30-124
 *   flvis_get_correction     <- flvis/CorrectionInf message payload             msg/CorrectionInf.msg, src/utils/correction_inf_msg.cpp:13-64
 *   flvis_ba_push_keyframe   <- LocalMapNodeletClass::frame_callback alone      src/backend/vo_localmap.cpp:87-380
 */
typedef struct flvis_cfg {
  int type_of_vi;                 /* 1 EuRoC (stereo unrectified + IMU), 3 D435i stereo, 5 D435 stereo + pixhawk,
                                     0 D435i depth, 2 D435 depth + pixhawk (the second image is the Z16 depth image),
                                     4 KITTI (rectified stereo, no IMU: the rig is given by cam{0,1}_projection_matrix, which
                                     flvis_config_load stores in P0 / P1 and flvis_config_finalize turns into K, T_cam0_cam1;
                                     src/frontend/vo_tracking.cpp:146,265-306) */
  int image_width, image_height;
  double cam0_intrinsics[4], cam0_distortion[4], cam1_intrinsics[4], cam1_distortion[4];
  double T_imu_cam0[16];          /* row-major 4x4 (EuRoC: T_imu_mavimu * T_mavimu_cam0) */
  double T_cam0_cam1[16];
  double vifusion_para[6], feature_para[6], dr_para[3];
  int window_size;
  /* derived by flvis_config_finalize (cv::stereoRectify restated, CALIB_ZERO_DISPARITY, alpha 0) */
  int cam_type, imu_type, skip_first_n_imgs, need_equal_hist;
  double R0[9], R1[9], P0[12], P1[12];
  double depth_factor;            /* depth modes: Z16 units per metre (yaml key depth_factor, vo_tracking.cpp:153) */
} flvis_cfg;

typedef struct flvis_frame_out {
  int state;            /* 0 UnInit, 1 Tracking, 2 TrackingFail (enum TRACKINGSTATE) */
  int new_keyframe, reset_cmd, n_landmarks;
  int64_t frame_id;
  double T_c_w[7];      /* tx ty tz qx qy qz qw of curr_frame */
  int of_inliers, f_inliers, pnp_inliers, pad_;   /* the counts lkorb_tracking.cpp:191 prints */
  double reprojection_error;
} flvis_frame_out;

int flvis_config_load(const char* yaml_path, flvis_cfg* cfg, char* err, int errlen);
int flvis_config_finalize(flvis_cfg* cfg);

/* CameraFrame::recover3DPts_c_FromStereo (src/processing/camera_frame.cpp:93-180) in ONE call -- the kernel-level drop-in a
 * configs[1] integrator (the reference's own frame loop, HIP pieces underneath) puts in its place: the stereo matcher's seeds
 * (the pixel itself, or for landmarks with depth the world point projected into camera 1 with T_cam1_cam0 * T_c_w, :108-122),
 * cv::calcOpticalFlowPyrLK(img0, img1, 31 x 31, maxLevel 5, 30 iterations / 0.001, OPTFLOW_USE_INITIAL_FLOW, :124-128),
 * cv::undistortPoints(K1, D1, R1, P1) (:130-131), Triangulation::trignaulationPtFromStereo with the rig's P0 / P1 (valid unless
 * z < 0 or z > range) and, for every landmark whose match or triangulation failed, the rand()-drawn dummy depth 0.3 + rand() / (RAND_MAX
 * / 0.4) through its undistorted pixel (:149-176) -- drawn in landmark order, as the reference's loop calls rand().
 * A "set" is one frame (n_sets frames of independent streams are matched in one launch).  Per set s, landmark i < d_count[s] (arrays
 * [n_sets][cap]...): d_pt2d_plane / d_pt2d_undistort / d_pt3d_w / d_has_depth = what getAll2dPlaneUndistort3d_cvPf and hasDepthInf()
 * hand the reference (cv::Point2f / Point3f: float); h_T_c_w7 (host, [n_sets][7], tx ty tz qx qy qz qw) the frame's pose.
 * d_img0 / d_img1: device [n_sets][h][w] mono8 of cfg's image size.  range: the float the reference passes (dr_para2).
 * d_rand_state35: the glibc generator of each set, [n_sets][35] int32 on the device, in / out -- flvis_hip_rand_seed(seed) initialises it
 * like srand(seed) (seed 1: a process that never called srand, which is what the reference is); consecutive calls continue the sequence.
 * Outputs: d_pt3d_c [n_sets][cap][3] (pt3ds, camera frame) and d_mask_has_3d [n_sets][cap] (maskHas3DInf).
 * Returns FLVIS_ERR_CONFIG for a depth-camera rig (recover3DPts_c_FromDepthImg is the tracker's own, flvis_image_feed). */
int flvis_hip_rand_seed(flvis_ctx* ctx, uint32_t seed, int32_t* d_state35, int n_sets);
int flvis_hip_stereo_depth(flvis_ctx* ctx, const flvis_cfg* cfg, const uint8_t* d_img0, const uint8_t* d_img1, int n_sets,
                           const float* d_pt2d_plane, const float* d_pt2d_undistort, const float* d_pt3d_w, const uint8_t* d_has_depth,
                           const int* d_count, int cap, const double* h_T_c_w7, float range, int32_t* d_rand_state35, double* d_pt3d_c,
                           uint8_t* d_mask_has_3d);

/* Creates the batched tracker (and local map) for n_streams independent streams inside `ctx`.  seed_base + stream is the
 * RANSAC seed of each stream.  traj_capacity > 0 keeps a device-side trajectory of that many frames per stream. */
int flvis_tracker_create(flvis_ctx* ctx, const flvis_cfg* cfg, int n_streams, uint64_t seed_base, int traj_capacity);

/* Number of lanes (sub-batches with their own HIP streams and device state) the tracker splits its streams into.  One by
 * default; the environment variable FLVIS_LANES (1..16) is a tuning knob (more lanes measured slower on MI355X, DESIGN.md
 * section 4).  The partition changes no result (streams are independent). */
int flvis_tracker_lanes(flvis_ctx* ctx);
/* Trackers with more than one lane (FLVIS_LANES): by default the context's stream waits for every lane at the end of a call, so
 * that the caller may overwrite the input images in stream order right away -- which also keeps the lanes in step.  A caller that
 * cycles through n + 1 input buffers (leaves a call's images untouched during the next n calls) says so here (0 <= n <= 7); the
 * lanes may then run up to n frames apart and the image kernels of one lane overlap the geometry chain of another. */
int flvis_set_input_hold(flvis_ctx* ctx, int n_frames);
/* Tuning aid: host milliseconds inside flvis_image_feed since flvis_tracker_create: [0] total, [1] of it blocked on the pinned
 * upload ring, [2] calls. */
int flvis_debug_host_times(flvis_ctx* ctx, double* h_out3);
/* ... and of flvis_image_feed_host since the tracker was created: [0] ms blocked on the previous call's uploads (the hold_buffers contract),
 * [1] ms issuing this call's uploads, [2] ms inside flvis_image_feed, [3] calls. */
int flvis_debug_host_feed_times(flvis_ctx* ctx, double* h_out4);
/* ... and the uploads themselves: enable = 1 brackets every call's uploads with two timing events on the copy stream, 0 stops that, -1
 * leaves the setting; h_out3 (may be NULL) = [0] ms of the bracketed uploads that have finished, [1] their bytes, [2] their number.
 * (What bench.py's with_h2d.upload_GBs is made of: bytes / the copies' own durations, not the host's view.) */
int flvis_debug_host_feed_timing(flvis_ctx* ctx, int enable, double* h_out3);

/* One IMU sample of stream `stream` in the SENSOR frame; remapped per type_of_vi like imu_callback does.  Samples are
 * staged on the host and consumed by the next flvis_image_feed (feed samples with t <= image time before the image). */
int flvis_imu_feed(flvis_ctx* ctx, int stream, double t, const double* acc3, const double* gyro3);
/* Same, n samples already in the FLVIS IMU frame: rows of 7 doubles (t, acc xyz, gyro xyz). */
int flvis_imu_feed_flvis_frame(flvis_ctx* ctx, int stream, int n, const double* samples7);

/* All streams at once: h_counts [n_streams], h_samples [n_streams][samples_per_stream][7] (FLVIS IMU frame). */
int flvis_imu_feed_all(flvis_ctx* ctx, const int* h_counts, const double* h_samples, int samples_per_stream);

/* The outputs of F2FTracking::imu_feed(time, acc, gyro, q_w_i&, pos_w_i&, vel_w_i&) (src/frontend/f2f_tracking.cpp:46-57): the
 * propagated IMU state after every sample -- what imu_callback publishes as /imu_pose, /imu_odom and /imu_path
 * (src/frontend/vo_tracking.cpp:362-369) and what the EuRoC launch file records as the estimated trajectory
 * (launch/flvis_euroc_mav.launch:88,103).  During the attitude initialisation the reference returns the identity attitude and
 * zero position / velocity (vi_motion.cpp:39-40; the sample that sets the first attitude returns it, :60); so do these.
 *   flvis_imu_feed_out     the call-for-call form: the sample is integrated now (one small upload + kernel + read-back) and its
 *                          state is returned: q_w_i (w, x, y, z), pos_w_i, vel_w_i.  For a nodelet that publishes at IMU rate.
 *   flvis_get_imu_states   the batched form for the deferred path (flvis_imu_feed stages, the next flvis_image_feed integrates):
 *                          the rows written since the previous call for that stream, oldest first, at most cap; a row is
 *                          (t, qw, qx, qy, qz, px, py, pz, vx, vy, vz).  Samples still staged are integrated first.  The device keeps
 *                          the last 512 rows per stream; *n_dropped (may be NULL) = older rows lost since the previous call.
 * Both forms run the same device code in the same order as the next frame head would, so the values are identical. */
int flvis_imu_feed_out(flvis_ctx* ctx, int stream, double t, const double* acc3, const double* gyro3, double* q_w_i_wxyz,
                       double* pos_w_i3, double* vel_w_i3);
int flvis_get_imu_states(flvis_ctx* ctx, int stream, int cap, double* h_rows11, int* n_out, int* n_dropped);

/* n_steps frames in one call -- the caller's loop `imu_callback ... ; image_input_callback` (vo_tracking.cpp:326-430) for frames whose
 * images are already in HBM: per step the IMU samples of all streams (h_imu_counts [n_streams], h_imu_samples
 * [n_streams][imu_samples_per_stream][7] in the FLVIS IMU frame; h_imu_counts NULL: none) and the stereo pair (device pointers as for
 * flvis_image_feed, h_times [n_streams]).  Equivalent to calling flvis_imu_feed_all + flvis_image_feed n_steps times; h_call_ms (may be
 * NULL) receives the host milliseconds each step spent enqueuing.  The images of step k must stay untouched as flvis_image_feed says.
 * One scheduling difference, none in the results: between two steps of a call the local-map launch for a step's keyframes is enqueued
 * inside the NEXT step (behind its PnP RANSAC) instead of behind the step itself, so that the ~0.3 ms in which two local-map launches
 * overlap do not fall on the temporal LK (FLVIS_BA_START=0: as flvis_image_feed).  The last step of a call launches at its end. */
typedef struct flvis_step {
  const uint8_t* d_img0;
  const uint8_t* d_img1;
  const double* h_times;
  const int* h_imu_counts;
  const double* h_imu_samples;
  int imu_samples_per_stream;
} flvis_step;
int flvis_run_steps(flvis_ctx* ctx, int n_steps, const flvis_step* steps, int with_local_map, double* h_call_ms);

/* Optional per-stage timing of flvis_image_feed with HIP events on the context's stream (for bench.py's roofline).
 * flvis_prof_enable(max_steps) arms it for the next max_steps frames; flvis_prof_read sums the elapsed ms per stage. */
int flvis_prof_enable(flvis_ctx* ctx, int max_steps);
/* Same, but only the stages whose bit is set in stage_mask record events (an event record costs a few microseconds on
 * the GPU queue; timing all ~20 stages inflates a frame by ~10%). */
int flvis_prof_enable_stages(flvis_ctx* ctx, int max_steps, uint64_t stage_mask);
int flvis_prof_stage_count(void);
const char* flvis_prof_stage_name(int i);
int flvis_prof_read(flvis_ctx* ctx, double* h_ms_per_stage, int* n_steps);
/* Per-frame elapsed ms of one enabled stage (stage "frame(chain)" = the whole main-stream chain of a frame, i.e. the GPU
 * latency of one batch step); returns the number of frames written (<= cap). */
int flvis_prof_read_steps(flvis_ctx* ctx, int stage, double* h_ms, int cap);

/* One stereo frame for every stream of the batch.  d_img0/d_img1: device [n_streams][h][w] mono8; h_times: host
 * [n_streams] seconds.  h_out (host, [n_streams], may be NULL): results; when NULL nothing is copied back and the call
 * does not synchronise.  with_local_map != 0 also runs the sliding-window BA for streams that emit a keyframe.
 * Depth-camera rigs (type_of_vi 0 / 2): d_img1 is the Z16 depth image aligned to cam0, [n_streams][h][w] uint16
 * (F2FTracking::image_feed(time, img0, d_img, ...), src/frontend/vo_tracking.cpp:453, f2f_tracking.cpp:116-119). */
int flvis_image_feed(flvis_ctx* ctx, const uint8_t* d_img0, const uint8_t* d_img1, const double* h_times,
                     flvis_frame_out* h_out, int with_local_map);

/* The same frame step with the images handed over as HOST buffers, the way TrackingNodeletClass::image_input_callback hands
 * two cv::Mat to F2FTracking::image_feed (src/frontend/vo_tracking.cpp:396-430, f2f_tracking.cpp:59-119): one flvis_image
 * per stream and camera.  channels 1 (mono8), 3 (BGR) or 4 (BGRA) -- colour input is converted like cv::cvtColor does at
 * f2f_tracking.cpp:74-111; on depth rigs h_img1 is the 16UC1 depth image (channels 1, 2 bytes per pixel, pitch in bytes).
 * The stamp of a stream's frame is h_img0[stream].t.  Rows may be padded (pitch >= width * bytes per pixel).  The uploads
 * run asynchronously on a copy stream into double-buffered device staging, so the H2D transfer of a frame overlaps the
 * kernels of the previous one; page-locked host memory (hipHostMalloc / hipHostRegister) is what makes them truly async.
 * hold_buffers == 0: returns once the uploads are done (the caller may reuse its buffers immediately);
 * hold_buffers != 0: returns at once, the buffers must stay valid until the next call on this context returns. */
typedef struct flvis_image {
  const uint8_t* data;
  int width, height, pitch, channels;
  double t;
} flvis_image;
int flvis_image_feed_host(flvis_ctx* ctx, const flvis_image* h_img0, const flvis_image* h_img1, flvis_frame_out* h_out,
                          int with_local_map, int hold_buffers);

/* Landmarks of curr_frame of one stream (host arrays of capacity cap): ids, raw pixel, rectified pixel, world point,
 * flags (bit0 has_3d, bit1 is_tracking_inlier).  Returns the landmark count (or <0). */
int flvis_get_landmarks(flvis_ctx* ctx, int stream, int cap, int64_t* h_id, double* h_2d, double* h_2d_undist,
                        double* h_3d_w, uint8_t* h_flags);
/* Last KeyFrame payload of a stream: returns lm_count (or <0); arrays of capacity cap. */
int flvis_get_keyframe(flvis_ctx* ctx, int stream, int cap, int64_t* frame_id, double* T_c_w7, int64_t* h_lm_id,
                       double* h_lm_2d, double* h_lm_3d);
/* The same keyframe as the full flvis/KeyFrame message (msg/KeyFrame.msg:1-11 as KeyFrameMsg::pub fills it,
 * src/utils/keyframe_msg.cpp:30-124): header.stamp, frame_id, command (KFMSG_CMD_NONE = 0; the reference's reset publisher is
 * commented out, vo_tracking.cpp:431 -- the reset request is flvis_frame_out.reset_cmd), the two images the tracker worked on
 * (img0: mono8 AFTER equalizeHist where the rig uses it; img1: mono8, or the 16UC1 depth image on depth rigs), lm_count with the
 * id / undistorted-pixel / world-point arrays, T_c_w; lm_descriptor_data is empty in the reference (:71-81).  The caller provides
 * the arrays (capacity cap) and, optionally, host buffers for the images (width * height bytes, 2 bytes per pixel for a depth
 * img1; NULL = not wanted).  Call it right after the flvis_image_feed that reported new_keyframe: the images are the
 * tracker's working copies of THAT frame.  LIFETIME: img0 is always the tracker's own copy; img1 is read through the pointer that
 * flvis_image_feed was given when the right / depth image is used in place (depth rigs, and stereo rigs without equalizeHist whose
 * rows are dword aligned): a caller of flvis_image_feed that wants img1 here must leave that device buffer untouched until this call
 * has returned (flvis_image_feed_host keeps its own double-buffered staging: nothing to observe there).
 * Returns lm_count (0: the stream's last frame was no keyframe; < 0: error). */
typedef struct flvis_keyframe {
  int64_t frame_id;
  int8_t command;
  double stamp;
  flvis_image img0, img1;       /* data = the host buffers passed in (or NULL), pitch = tightly packed */
  int32_t lm_count;
  const int64_t* lm_id;         /* the caller's arrays */
  const double* lm_2d;          /* [lm_count][2] undistorted pixel */
  const double* lm_3d;          /* [lm_count][3] world */
  double T_c_w[7];              /* tx ty tz qx qy qz qw */
} flvis_keyframe;
int flvis_get_keyframe_msg(flvis_ctx* ctx, int stream, int cap, flvis_keyframe* kf, int64_t* h_lm_id, double* h_lm_2d,
                           double* h_lm_3d, uint8_t* h_img0, uint8_t* h_img1);
/* Last CorrectionInf of a stream: returns 1 if one exists (0 if not yet, <0 error). */
int flvis_get_correction(flvis_ctx* ctx, int stream, int cap, int64_t* frame_id, double* T_c_w7, int* lm_count,
                         int64_t* h_lm_id, double* h_lm_3d, int* lm_outlier_count, int64_t* h_outlier_id);

/* Local-map feedback into the tracker.  F2FTracking::correction_feed (src/frontend/f2f_tracking.cpp:40-44):
 * the reference's v2 unpacks CorrectionInf in correction_feedback_callback (vo_tracking.cpp:373-385) and never calls it, so
 * this is opt-in -- a caller that never feeds a correction gets v2 behaviour.  The correction (e.g. what
 * flvis_get_correction returned) is applied at the stream's next Tracking frame exactly as f2f_tracking.cpp:189-219 does:
 * pose_records and last_frame->T_c_w are re-anchored on the corrected keyframe pose, lm_3d_w of the named landmarks is
 * overwritten, the named outliers lose is_tracking_inlier.  Host buffers; returns after the upload. */
int flvis_correction_feed(flvis_ctx* ctx, int stream, int64_t frame_id, const double* T_c_w7, int lm_count,
                          const int64_t* h_lm_id, const double* h_lm_3d, int lm_outlier_count,
                          const int64_t* h_lm_outlier_id);
/* F2FTracking::pose_records (f2f_tracking.h:59) of a stream, oldest first: rows (frame_id, tx ty tz qx qy qz qw).
 * Returns the number of records (< 1000). */
int flvis_get_pose_records(flvis_ctx* ctx, int stream, int cap, double* h_rows8);
/* Device-side trajectory of one stream: rows of 9 doubles (t, tx ty tz qx qy qz qw, state | new_kf<<4). */
int flvis_get_trajectory(flvis_ctx* ctx, int stream, int first_frame, int n_frames, double* h_rows9);
/* Trajectory recorder (replaces src/independ_modules/vo_repub_rec.cpp:74-124 for offline runs): writes the recorded
 * camera poses T_w_c (inverse of T_c_w) of frames [first_frame, first_frame + n_frames) whose state is TRACKING to a
 * text file.  format 0: `stamp x y z qw qx qy qz` per line (vo_repub_rec.cpp:82-91, the TUM order with qw first);
 * format 1: KITTI, 12 row-major entries of [R | t] per line (:100-111).  min_dt > 0 emulates the recorder's throttle exactly
 * as written (:77-78): its `last_time` is set at the first call and never updated, so poses stamped within min_dt of the
 * first TRACKING pose are dropped and every later one is written (the reference: 0.1 s of wall clock).  Returns the number
 * of lines written or a negative error code. */
int flvis_write_trajectory(flvis_ctx* ctx, int stream, int first_frame, int n_frames, const char* path, int format,
                           double min_dt);
/* The recorder on /imu_pose (launch/flvis_euroc_mav.launch:83-103: vo_repub_rec, sub_type PoseStamped, sub_topic /imu_pose -> est.txt,
 * the trajectory the reference scores on EuRoC): writes rows of flvis_get_imu_states as `stamp x y z qw qx qy qz` (position pos_w_i,
 * attitude q_w_i; vo_repub_rec.cpp:74-91).  min_dt > 0: the throttle as written -- rows within min_dt of the FIRST row of the run are
 * dropped, every later one is written.  The run starts with the call that creates the file (append == 0): the throttle applies to that
 * call only; a batch that is appended (append != 0; flvis_get_imu_states hands out at most 512 rows per fetch) is written in full.
 * Returns the number of lines written, FLVIS_ERR_INVALID_ARG for bad arguments or FLVIS_ERR_CONFIG when the file cannot be opened. */
int flvis_write_imu_trajectory(const double* h_rows11, int n, const char* path, double min_dt, int append);
/* The same with the run's first stamp named by the caller (t_first = h_rows11[0] of the run's first batch; NaN: no throttle): rows
 * within min_dt of t_first are dropped in EVERY batch, also when the batch that created the file was shorter than min_dt
 * (the recorder's last_time is set once and never updated, vo_repub_rec.cpp:77-78). */
int flvis_write_imu_trajectory_run(const double* h_rows11, int n, const char* path, double min_dt, int append, double t_first);
/* Counters: [0] frames fed, [1] keyframes, [2] BA runs. */
int flvis_get_counters(flvis_ctx* ctx, int64_t* h_counters3);
/* The first n (1 .. 4) of: [0] frames fed, [1] keyframes, [2] BA runs, [3] keyframes dropped at a full keyframe queue -- what the
 * reference's /vo_kf subscriber (queue size 10, src/backend/vo_localmap.cpp:452-456) does when the local map is slower than the tracker.
 * The tracker's stream-ordered back-pressure keeps a queue below its capacity, so [3] stays 0 unless a caller pushes keyframes itself
 * (flvis_ba_push_keyframe) faster than it lets the local map run. */
int flvis_get_counters_n(flvis_ctx* ctx, int n, int64_t* h_counters);
/* Per stream (arrays of n_streams, either may be NULL): keyframes the tracker has emitted and optimisations the stream's local map has
 * run (one per keyframe once the window holds window_size keyframes, vo_localmap.cpp:211-214,292-366).  Drains the queues first. */
int flvis_get_local_map_counts(flvis_ctx* ctx, int64_t* h_keyframes, int64_t* h_ba_runs);
/* Test aid: poses (tx ty tz qx qy qz qw) of a stream's last Tracking frame right after PnP-RANSAC and after the pose-only LM
 * (the two fp64 stages of LKORBTracking::tracking / OptimizeInFrame::optimize), h_out21 = 3 x 7 doubles (the third: the pose the LM starts from). */
int flvis_debug_stage_poses(flvis_ctx* ctx, int stream, double* h_out21);
/* Measurement aid (bench.py's LK instruction budget): enable != 0 makes the tracker's two LK launches per frame count their Gauss-Newton
 * iterations and the points that iterated, per pyramid level, into flvis_debug_counters: [36 + 2 l], [37 + 2 l] temporal LK at level l,
 * [48 + 2 l], [49 + 2 l] stereo LK; and, over both launches, [61] templates the temporal LK took from the template cache (written by
 * the previous frame's stereo LK), [62] template patches and [63] search regions staged by the slow (index-reflecting) path, i.e.
 * blocks that leave the pyramids' physical border. */
int flvis_debug_lk_stats(flvis_ctx* ctx, int enable);
/* Test aid: the tracker's pyramid construction on its own (the kernels F2FTracking's frames go through: the walking kernels of
 * pyr_walk.hip where the image geometry allows, else the LDS-tile kernels, then k_pyr_border for what is left).  Builds levels 1 .. levels
 * of n_img images (w x h, tightly packed) with a physical BORDER_REFLECT_101 border of bx columns / by rows around every level, as
 * cv::buildOpticalFlowPyramid keeps it.  d_out: level l = 0 .. levels at consecutive offsets (each level block rounded up to 64 bytes),
 * a block being [n_img][h_l + 2 by][pitch_l] bytes with pitch_l = ((w_l + 15) & ~15) + 2 bx and pixel (0, 0) at row by, column bx.
 * ingest != 0: level 0 of d_out is the copy of the source (with border); 0: level 0 is read in place and its block is left untouched. */
int flvis_debug_pyramid(flvis_ctx* ctx, const uint8_t* d_src, int w, int h, int n_img, int levels, int bx, int by, int ingest, uint8_t* d_out,
                        size_t out_bytes);
/* Test aid: the corner-response pass of flvis_hip_gftt alone (cornerMinEigenVal + the 3x3 local maxima), with the kernel variant chosen
 * (0: LDS tiles, 1: strip-mined tiles, 2: wave walk with `rows` rows per chunk): per image the ordered bits of the maximum response, the
 * number of local maxima and their sort keys ~((ordered(response) << 32) | pixel offset), unsorted, in h_keys [n_img][key_cap]. */
int flvis_hip_debug_corner_response(flvis_ctx* ctx, const uint8_t* d_img, int w, int h, int n_img, int variant, int rows,
                                    uint32_t* h_max_bits, int* h_nkeys, uint64_t* h_keys, int key_cap);
/* Test aid: the corner-response kernel's square root against the correctly rounded sqrtf on every float whose bit pattern lies in
 * [first_bits, first_bits + n); *h_mismatches = arguments on which the two differ (0 over the kernel's domain, see eig_walk.hip). */
int flvis_hip_debug_sqrt_check(flvis_ctx* ctx, uint32_t first_bits, uint32_t n, uint64_t* h_mismatches);
/* Raw device counter block (64 x int64): [0..7] as above, [8..] per-phase cycle counters of the BA kernel, filled only by
 * builds with -DFLVIS_BA_PROF (tuning aid, not part of the reference interface). */
int flvis_debug_counters(flvis_ctx* ctx, int64_t* h_counters64);

/* LocalMapNodeletClass::frame_callback for one stream with a caller-supplied KeyFrame (host arrays).  Returns 1 and fills
 * the outputs when an optimisation ran, 0 while the window is still filling. */
int flvis_ba_push_keyframe(flvis_ctx* ctx, int stream, int64_t frame_id, const double* T_c_w7, int lm_count,
                           const int64_t* h_lm_id, const double* h_lm_2d, const double* h_lm_3d, int cap,
                           int64_t* out_frame_id, double* out_T_c_w7, int* out_lm_count, int64_t* out_lm_id,
                           double* out_lm_3d, int* out_outlier_count, int64_t* out_outlier_id);

/* ---- IMU rotation factor of the window BA (an ADDITION -- the reference's local map, src/backend/vo_localmap.cpp,
 * optimises reprojection edges only, so nothing here has a reference counterpart and everything is off by default).
 *
 * The tracker integrates the bias-corrected gyro samples between keyframes (the samples VIMOTION::viIMUPropagation consumes,
 * src/processing/vi_motion.cpp:78-100) into dq = R_body(previous keyframe)^T R_body(this keyframe); the preintegration travels with
 * the KeyFrame payload.  With the factor enabled, the window BA adds for every pair of consecutive keyframes the edge
 *     r = Log(dq^T R_b(a)^T R_b(b)),   R_b = R_c_w^T R_c_i,   information I3 / (sigma_gyro^2 dt)
 * next to the reprojection edges (pose-pose blocks in the reduced camera system).
 *   flvis_set_imu_factor       enable != 0 switches the edges on for every stream; sigma_gyro = gyro noise density [rad/s/sqrt(Hz)]
 *   flvis_get_keyframe_imu     the preintegration of the stream's last keyframe: dq (w, x, y, z), dt [s]; returns 1 when it links
 *                              the keyframe to its predecessor, 0 otherwise (first keyframe after an initialisation, no IMU)
 *   flvis_ba_push_keyframe_imu flvis_ba_push_keyframe with the preintegration of the pushed keyframe (imu_dt <= 0: none) */
int flvis_set_imu_factor(flvis_ctx* ctx, int enable, double sigma_gyro);
/* The factor's position rows (an addition as the rotation rows are; off by default): between consecutive keyframes a -> b
 *   r_p = R_b(a)^T (p_b(b) - p_b(a) - v_a dt + 1/2 g_w dt^2) - dp,      information I3 / (sigma_acc^2 dt^3 / 3),
 * with dp = sum (dv dt + 1/2 dR f dt^2), dv = sum dR f dt the position / velocity preintegration of the bias-corrected accelerometer
 * samples (VIMOTION's convention: world acceleration = R f - g_w, g_w = (0, 0, -9.81)), and v_a the tracker's filter velocity at keyframe a,
 * a fixed quantity: there is no velocity or bias vertex, the pose blocks stay 6-dimensional.
 *   flvis_set_imu_factor_accel     sigma_acc > 0 [m/s^2/sqrt(Hz)] adds the rows to every IMU edge (flvis_set_imu_factor must be on); <= 0: off
 *   flvis_get_keyframe_imu_pos     dp (body frame of the previous keyframe) and v_a (world frame) of the stream's last keyframe; returns 1
 *   flvis_ba_push_keyframe_imu_pos flvis_ba_push_keyframe_imu with dp / v_a of the pushed keyframe */
int flvis_set_imu_factor_accel(flvis_ctx* ctx, double sigma_acc);
int flvis_get_keyframe_imu_pos(flvis_ctx* ctx, int stream, double* dp3, double* va3);
int flvis_ba_push_keyframe_imu_pos(flvis_ctx* ctx, int stream, int64_t frame_id, const double* T_c_w7, const double* imu_dq_wxyz,
                                   double imu_dt, const double* imu_dp3, const double* imu_va3, int lm_count, const int64_t* h_lm_id,
                                   const double* h_lm_2d, const double* h_lm_3d, int out_cap, int64_t* out_frame_id, double* out_T_c_w7,
                                   int* out_lm_count, int64_t* out_lm_id, double* out_lm_3d, int* out_outlier_count,
                                   int64_t* out_outlier_id);
int flvis_get_keyframe_imu(flvis_ctx* ctx, int stream, double* dq_wxyz, double* dt);
int flvis_ba_push_keyframe_imu(flvis_ctx* ctx, int stream, int64_t frame_id, const double* T_c_w7, const double* imu_dq_wxyz,
                               double imu_dt, int lm_count, const int64_t* h_lm_id, const double* h_lm_2d, const double* h_lm_3d,
                               int cap, int64_t* out_frame_id, double* out_T_c_w7, int* out_lm_count, int64_t* out_lm_id,
                               double* out_lm_3d, int* out_outlier_count, int64_t* out_outlier_id);

/* ---- the loop-closing nodelet's control flow for a batch of independent sequences --------------------------------
 * LoopClosingNodeletClass (src/backend/vo_loopclosing.cpp:118-1119) without ROS: the keyframe database stays on the device.
 *   flvis_lc_params_load             <- onInit's LC_PARAS block                  :955-963
 *   flvis_loop_closer_create         <- onInit (camera from the same yaml, vocabulary already in the context: :1095-1101)
 *   flvis_loop_closer_add_keyframes  <- frame_callback + kfmsgProcess            :178-391
 *   flvis_loop_closer_process        <- one pass of pgoProcess per new keyframe  :393-518 (+ :520-944)
 * The reference's pgoProcess thread polls the newest keyframe (a keyframe may be examined twice or never); here every keyframe is
 * examined exactly once, in order. */
typedef struct flvis_lc_params {
  int lcKFStart, lcKFDist, lcKFMaxDist, lcKFLast, lcNKFClosest, minPts; /* lcKFStart / lcKFLast are read but unused by the reference too */
  double ratioMax, ratioRansac, minScore;
} flvis_lc_params;
typedef struct flvis_lc_event {
  int64_t kf_prev, kf_curr;  /* kf_curr = -1: the sequence had no new keyframe; kf_prev = -1: no candidate */
  int candidate;             /* isLoopCandidate */
  int n_matches, n_inliers;  /* mutual + ratio matches (p3d.size()), inliers of solvePnPRansac */
  int loop_accepted;         /* isLoopClosureKF: the loop went into loop_ids / loop_poses */
  int optimised;             /* loopClosureOnCovGraphG2ONew ran (:492-497) */
  int pgo_iterations;
  double loop_pose7[7];      /* se_ji (tx ty tz qx qy qz qw): the later keyframe's camera from the earlier one's */
  double chi2_before, chi2_after;
} flvis_lc_event;
typedef struct flvis_loop_closer flvis_loop_closer;
int flvis_lc_params_load(const char* yaml_path, flvis_lc_params* prm, char* err, int errlen);
/* cfg: flvis_config_load of the same yaml (image size, cam_type, rectified P0 / P1); the vocabulary must be resident
 * (flvis_hip_bow_load_vocabulary).  max_keyframes slots per sequence are allocated up front (76 KB each).  h_orb_pattern: see
 * flvis_hip_orb_detect_and_compute (NULL = the built-in pattern). */
int flvis_loop_closer_create(flvis_ctx* ctx, const flvis_cfg* cfg, const flvis_lc_params* prm, int n_streams, int max_keyframes,
                             const int8_t* h_orb_pattern, flvis_loop_closer** out);
/* (destroy it before the context it was created on: every other call runs on that context's stream) */
void flvis_loop_closer_destroy(flvis_loop_closer* lc);
/* one keyframe for each of the n sequences h_stream[i] (distinct): d_img0 [n][h][w] mono8, d_img1 [n][h][w] mono8 (stereo) or Z16
 * (depth camera), h_T_c_w_odom7 [n][7] the tracker's pose of the keyframe (KeyFrame.msg T_c_w).  h_kf_id (optional): the keyframe's
 * index in its sequence. */
int flvis_loop_closer_add_keyframes(flvis_loop_closer* lc, int n, const int* h_stream, const uint8_t* d_img0, const void* d_img1,
                                    const double* h_T_c_w_odom7, int64_t* h_kf_id);
/* the same on HOST images (what KeyFrameMsg::unpack hands the nodelet, :206): h_img0[i] mono8, h_img1[i] mono8 or 16UC1, pitch in bytes */
int flvis_loop_closer_add_keyframes_host(flvis_loop_closer* lc, int n, const int* h_stream, const flvis_image* h_img0,
                                         const flvis_image* h_img1, const double* h_T_c_w_odom7, int64_t* h_kf_id);
/* examines the newest keyframe of every sequence that got one since the last call; h_events [n_streams] */
int flvis_loop_closer_process(flvis_loop_closer* lc, flvis_lc_event* h_events);
/* kf_map_lc[i]->T_c_w of one sequence (host, [cap][7]); *n_out = keyframes in the sequence */
int flvis_loop_closer_poses(flvis_loop_closer* lc, int stream, double* h_T_c_w7, int cap, int* n_out);
/* one keyframe of the device-resident database on the host (KeyFrameLC, :100-112): the kept ORB landmarks (pixel, camera-frame position,
 * descriptor; arrays of capacity cap) and the bag-of-words vector; any output pointer may be NULL; the counts are the full counts */
int flvis_loop_closer_keyframe(flvis_loop_closer* lc, int stream, int kf, int cap, float* h_lm_2d, double* h_lm_3d, uint8_t* h_lm_desc,
                               int* lm_count, int* h_bow_ids, double* h_bow_vals, int* bow_count);
/* T_odom_map (:138, the tf map -> odom the nodelet broadcasts is its inverse) */
int flvis_loop_closer_drift(flvis_loop_closer* lc, int stream, double* h_T_odom_map7);
/* the newest row of the sequence's similarity matrix as the last flvis_loop_closer_process computed it */
int flvis_loop_closer_similarity_row(flvis_loop_closer* lc, int stream, double* h_row, int cap, int* n_out);

#ifdef __cplusplus
}
#endif
#endif /* FLVIS_HIP_H */
