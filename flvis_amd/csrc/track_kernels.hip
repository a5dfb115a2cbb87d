// flvis_amd: batched front-end state machine kernels (gfx950).  Device-resident mirror of
//   F2FTracking::image_feed / init_frame        src/frontend/f2f_tracking.cpp:59-453
//   LKORBTracking::tracking (stages around LK)  src/processing/lkorb_tracking.cpp:9-202
//   OptimizeInFrame::optimize                   src/processing/optimize_in_frame.cpp:10-91 (+ g2o LM)
//   CameraFrame::{calReprjInlierOutlier, eraseReprjOutlier, depthInnovation, eraseNoDepthPoint, getKeyFrameInf}
//                                               src/processing/camera_frame.cpp:18-91,93-180,236-330,515-528
// Every kernel handles all S streams of the batch and masks itself with the stream's phase flags, so the host enqueues
// the same kernel sequence every frame (no host decisions, graph-capturable).  Latency-bound stages use one wave (or
// one thread) per stream; the J^T J / J^T r assemblies are wave reductions (DPP/shuffle butterflies).
#include <vector>

#include "dev_common.hpp"
#ifdef FLVIS_RANSAC_PROF
// sub-phase stamps of the 7-point solver (lane 0 of the hypothesis wave), counters[40..]
#define SP_STAMP(i)                                                                                      \
  do {                                                                                                   \
    if (threadIdx.x == 0 && g_sp_prof) {                                                                 \
      long long now_ = (long long)wall_clock64();                                                        \
      atomicAdd((unsigned long long*)&g_sp_prof[i], (unsigned long long)(now_ - g_sp_last));            \
      g_sp_last = now_;                                                                                  \
    }                                                                                                    \
  } while (0)
namespace flvis {
static __device__ long long* g_sp_prof = nullptr;
static __device__ long long g_sp_last = 0;
}
#endif
#include "dev_geom.hpp"
#include "epnp_core.hpp"
#include "track_kernels.hpp"
#include "cv_solvers.hpp"
#include "vi_motion.hpp"

namespace flvis {

FD Landmark* lm_ptr(const Pipe& p, int slot, int s) { return p.lm + ((size_t)slot * p.S + s) * NMAX; }

// dynamic LDS of the kernels that need more than the 64 KB static window (k_pose_lm, the refinement of k_ransac_pnp)
extern __shared__ __attribute__((aligned(16))) unsigned char pl_smem[];

// glibc rand() (TYPE_3): next output from the stream's ring
FD int glibc_rand_next(StreamState& st) {
  int pos = st.rnd_pos;
  int a = st.rnd_r[(pos + 34 - 31) % 34], b = st.rnd_r[(pos + 34 - 3) % 34];
  int n = (int)((unsigned)a + (unsigned)b);
  st.rnd_r[pos] = n;
  st.rnd_pos = (pos + 1) % 34;
  return (int)(((unsigned)n) >> 1);
}

FD void track_fail(StreamState& st) {  // f2f_tracking.cpp:235-247 / 257-269
  st.cont_fail++;
  st.cur ^= 1;  // last_frame.swap(curr_frame): escape this frame
  if (st.cont_fail >= 2) {
    st.state = ST_TRACKFAIL;
    st.cont_fail = 0;
  }
  st.phase = PH_IDLE;
  st.ok = 0;
}

// ------------------------------------------------------------------------------------------------ IMU
// the staged samples of stream s, in order.  Once the filter is initialised the samples are integrated on REGISTER-resident state
// (the previous state, the ring cursor, the keyframe preintegration are loaded once and written back once; only the new ring
// entries are stored) -- per sample the same arithmetic as vi_imu_feed, without a store -> load round trip through HBM per sample
// `in`: the stream's staged samples (7 doubles each) -- p.imu_in's rows, or a copy of them in LDS (k_frame_head: a sample's values then
// cost an LDS read instead of a global round trip in front of every step of the sequential integration)
__device__ inline void imu_feed_dev(const Pipe& p, int s, const double* in) {
  StreamState& st = p.st[s];
  ViRing ring{p.vi + (size_t)s * VI_QUEUE, &st};
  int n = p.n_imu[s];
  if (n > IMU_MAX) n = IMU_MAX;
  // F2FTracking::imu_feed's per-sample outputs go to the stream's output ring (row = sample number % IMU_OUT_CAP)
  double* const out = p.imu_out + (size_t)s * IMU_OUT_CAP * IMU_ROW;
  long long seen = st.imu_seen;
  int i = 0;
  for (; i < n && !st.vi_initialized; i++, seen++)  // start-up: attitude initialisation, sample by sample
    vi_imu_feed(p.cam, st, ring, in[7 * i], V3{in[7 * i + 1], in[7 * i + 2], in[7 * i + 3]},
                V3{in[7 * i + 4], in[7 * i + 5], in[7 * i + 6]}, out + (size_t)(seen % IMU_OUT_CAP) * IMU_ROW);
  if (i < n) {
    const V3 acc_bias = ld3(st.acc_bias), gyro_bias = ld3(st.gyro_bias);
    int head = st.vi_head, count = st.vi_count;
    MotionState prev = ring.back();
    Q4 kdq{st.kf_dq[0], st.kf_dq[1], st.kf_dq[2], st.kf_dq[3]};
    double kdt = st.kf_dt;
    V3 kdp = ld3(st.kf_dp), kdv = ld3(st.kf_dv);
    for (; i < n; i++, seen++) {
      MotionState cur;
      vi_propagate(p.cam, prev, in[7 * i], V3{in[7 * i + 1], in[7 * i + 2], in[7 * i + 3]},
                   V3{in[7 * i + 4], in[7 * i + 5], in[7 * i + 6]}, acc_bias, gyro_bias, cur, kdq, kdt, kdp, kdv);
      imu_row_store(out + (size_t)(seen % IMU_OUT_CAP) * IMU_ROW, cur.t, ms_q(cur), ld3(cur.pos), ld3(cur.vel));
      ring.base[(head + count) % VI_QUEUE] = cur;  // ViRing::push_back on the local cursor
      count++;
      if (count >= VI_QUEUE) {
        head = (head + 1) % VI_QUEUE;
        count--;
      }
      prev = cur;
    }
    st.vi_head = head;
    st.vi_count = count;
    st.kf_dq[0] = kdq.w, st.kf_dq[1] = kdq.x, st.kf_dq[2] = kdq.y, st.kf_dq[3] = kdq.z;
    st.kf_dt = kdt;
    st3(st.kf_dp, kdp);
    st3(st.kf_dv, kdv);
  }
  st.imu_seen = seen;
  p.n_imu[s] = 0;
}
__global__ void k_imu_feed(Pipe p) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= p.S) return;
  imu_feed_dev(p, s, p.imu_in + (size_t)s * IMU_MAX * 7);
}

// pose_records.push_back(...) (+ pop_front once 1000 entries are reached on a tracking frame, f2f_tracking.cpp:334-337)
__device__ inline void pose_record_push(const Pipe& p, int s, StreamState& st, int frame_id, const double* T7, bool pop) {
  if (st.rec_count == POSE_REC) {  // only reachable after ~25 re-initialisations with a full record: drop the oldest
    st.rec_head = (st.rec_head + 1) % POSE_REC;
    st.rec_count--;
  }
  const int k = (st.rec_head + st.rec_count) % POSE_REC;
  p.rec_id[(size_t)s * POSE_REC + k] = frame_id;
  for (int j = 0; j < 7; j++) p.rec_T[((size_t)s * POSE_REC + k) * 7 + j] = T7[j];
  st.rec_count++;
  if (pop && st.rec_count >= 1000) {
    st.rec_head = (st.rec_head + 1) % POSE_REC;
    st.rec_count--;
  }
}

// ------------------------------------------------------------------------------------------------ local-map feedback
// STEP1 of the Tracking case (f2f_tracking.cpp:189-219), run after frame_begin for streams with a pending correction:
// re-anchor pose_records and last_frame->T_c_w on the corrected keyframe pose, overwrite lm_3d_w of the landmarks the
// correction names (forceCorrectLM3DW, camera_frame.cpp:344-360; ids narrowed to int as there) and clear the inlier flag of
// its outliers (forceMarkOutlier, :362-378).  correctLMP3DWByLMP3DCandT (:332-342) iterates by value: a no-op, as there.
constexpr int AC_T = 256;
__global__ __launch_bounds__(AC_T) void k_apply_correction(Pipe p) {
  const int s = blockIdx.x;
  StreamState& st = p.st[s];
  CorrectionDev& c = p.corr_in[s];
  if (!c.valid || st.phase != PH_TRACK) return;  // has_localmap_feedback stays set until a Tracking frame
  __shared__ double s_old_inv[7], s_upd[7];
  __shared__ int s_idx;
  __shared__ long long s_ids[1024];
  const int tid = threadIdx.x;
  const int last = st.cur ^ 1;
  if (tid == 0) {
    const int corr_id = (int)c.frame_id;
    int idx = 0;
    for (int i = st.rec_count - 1; i >= 0; i--)
      if (p.rec_id[(size_t)s * POSE_REC + (st.rec_head + i) % POSE_REC] == corr_id) {
        idx = i;
        break;
      }
    s_idx = idx;
    const SE3d old = load_pose7(p.rec_T + ((size_t)s * POSE_REC + (st.rec_head + idx) % POSE_REC) * 7);
    store_pose7(s_old_inv, se3_inverse(old));
    for (int j = 0; j < 7; j++) s_upd[j] = c.T_c_w[j];
  }
  __syncthreads();
  const SE3d old_inv = load_pose7(s_old_inv), upd = load_pose7(s_upd);
  for (int i = s_idx + tid; i < st.rec_count; i += AC_T) {
    double* T = p.rec_T + ((size_t)s * POSE_REC + (st.rec_head + i) % POSE_REC) * 7;
    store_pose7(T, se3_mul(se3_mul(load_pose7(T), old_inv), upd));
  }
  if (tid == 0) store_pose7(st.T_c_w[last], se3_mul(se3_mul(load_pose7(st.T_c_w[last]), old_inv), upd));
  Landmark* lms = lm_ptr(p, last, s);
  const int n = st.n_lm[last];
  // forceCorrectLM3DW: with unique ids on both sides "first landmark with that id" and "last correction entry wins" both
  // reduce to a plain join; a duplicated correction id resolves to its last entry as in the reference's sequential loop
  const int nc = min(c.lm_count, BA_LMAX);
  for (int c0 = 0; c0 < nc; c0 += 1024) {
    const int m = min(1024, nc - c0);
    __syncthreads();
    for (int i = tid; i < m; i += AC_T) s_ids[i] = (long long)(int)c.lm_id[c0 + i];
    __syncthreads();
    for (int j = tid; j < n; j += AC_T) {
      const long long id = lms[j].id;
      int hit = -1;
      for (int i = 0; i < m; i++)
        if (s_ids[i] == id) hit = i;
      if (hit >= 0)
        for (int k = 0; k < 3; k++) lms[j].p3w[k] = c.lm_3d[c0 + hit][k];
    }
  }
  const int no = min(c.lm_outlier_count, BA_EMAX);
  for (int c0 = 0; c0 < no; c0 += 1024) {
    const int m = min(1024, no - c0);
    __syncthreads();
    for (int i = tid; i < m; i += AC_T) s_ids[i] = (long long)(int)c.lm_outlier_id[c0 + i];
    __syncthreads();
    for (int j = tid; j < n; j += AC_T) {
      const long long id = lms[j].id;
      bool hit = false;
      for (int i = 0; i < m; i++) hit = hit || (s_ids[i] == id);
      if (hit) lms[j].inlier = 0;
    }
  }
  __syncthreads();
  if (tid == 0) c.valid = 0;
}

// ------------------------------------------------------------------------------------------------ frame begin
__device__ inline void frame_begin_dev(const Pipe& p, int s, double time);
__global__ void k_frame_begin(Pipe p, const double* __restrict__ frame_time) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= p.S) return;
  frame_begin_dev(p, s, frame_time[s]);
}
// the head of a frame in one launch: the staged IMU samples (F2FTracking::imu_feed), then the frame set-up
// One wavefront per stream: the integration itself is sequential (a sample's state follows from the previous one: lane 0 runs it), but
// its inputs need not arrive one global round trip at a time -- the wave copies the stream's staged samples into LDS first.
__device__ __forceinline__ void k_frame_head_body(const Pipe& p, const double* __restrict__ frame_time, long long* __restrict__ host_progress,
                                                   long long frame_no) {
  chain_priority();
  const int s = blockIdx.x, tid = threadIdx.x;
  // tell the host that this frame's inputs have been uploaded (this kernel follows the upload in stream order): the pinned
  // staging slot of the frame may be refilled (a store to host-mapped memory; the host polls it, see lane_frame)
  if (s == 0 && tid == 0 && host_progress) __hip_atomic_store(host_progress, frame_no, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  __shared__ double s_in[IMU_MAX * 7];
  int n = p.n_imu[s];
  if (n > IMU_MAX) n = IMU_MAX;
  const double* in = p.imu_in + (size_t)s * IMU_MAX * 7;
  for (int i = tid; i < 7 * n; i += 64) s_in[i] = in[i];
  const double t_frame = frame_time[s];
  __syncthreads();
  if (tid != 0) return;
  imu_feed_dev(p, s, s_in);
  frame_begin_dev(p, s, t_frame);
}
__global__ __launch_bounds__(64) void k_frame_head(Pipe p, const double* __restrict__ frame_time, long long* __restrict__ host_progress,
                                                   long long frame_no) {
  kj_wait(p.kj);
  k_frame_head_body(p, frame_time, host_progress, frame_no);
  kj_signal(p.kj);
  kj_post_wait(p.kj);
}
__device__ inline void frame_begin_dev(const Pipe& p, int s, double time) {
  StreamState& st = p.st[s];
  ViRing ring{p.vi + (size_t)s * VI_QUEUE, &st};
  st.frameCount++;
  st.cur ^= 1;
  const int c = st.cur;
  st.n_lm[c] = 0;
  store_pose7(st.T_c_w[c], se3_identity());
  st.frame_id[c] = st.frameCount;
  st.frame_time[c] = time;
  st.new_kf = 0;
  st.reset_cmd = 0;
  st.phase = PH_IDLE;
  st.ok = 1;
  st.use_guess = 0;
  st.of_cnt = st.f_cnt = st.pnp_cnt = 0;
  st.n_new = 0;
  int det_mode = 0;
  if (st.skip_n > 0) {
    st.skip_n--;
  } else {
    switch (st.state) {
      case ST_UNINIT: {
        M3 R_w_c;
        R_w_c.m[0][0] = 0; R_w_c.m[0][1] = 0; R_w_c.m[0][2] = 1;
        R_w_c.m[1][0] = -1; R_w_c.m[1][1] = 0; R_w_c.m[1][2] = 0;
        R_w_c.m[2][0] = 0; R_w_c.m[2][1] = -1; R_w_c.m[2][2] = 0;
        SE3d T = se3_inverse(se3_from_mat(R_w_c, V3{0, 0, 0}));
        bool go = true;
        if (st.has_imu) {
          if (st.vi_initialized) {
            Q4 q_init;
            vi_vision_trigger(ring, q_init);
            M3 R = q_to_mat(q_init) * q_to_mat(load_pose7(p.cam.T_i_c).q);
            T = se3_inverse(se3_from_mat(R, V3{0, 0, 0}));
          } else {
            go = false;
          }
        }
        store_pose7(st.T_c_w[c], T);
        if (go) {
          st.phase = PH_INIT;
          det_mode = 1;
        }
        break;
      }
      case ST_TRACKING: {
        SE3d g = se3_identity();
        if (st.has_imu) st.use_guess = vi_get_corr_frame_state(p.cam, ring, time, g) ? 1 : 0;
        store_pose7(st.guess, g);
        st.phase = PH_TRACK;
        break;
      }
      case ST_TRACKFAIL: {
        st.tf_cnt++;
        if ((st.tf_cnt % 3) == 0) {
          SE3d T;
          if (vi_get_corr_frame_state(p.cam, ring, time, T)) {
            store_pose7(st.T_c_w[c], T);
            st.phase = PH_INIT;
            det_mode = 1;
          } else {
            st.cur ^= 1;
          }
          st.tf_cnt = 0;
        } else {
          st.cur ^= 1;
          if ((st.tf_cnt % 2) == 0) st.reset_cmd = 1;
        }
        break;
      }
    }
  }
  p.act_img[s] = st.phase != PH_IDLE;
  p.act_track[s] = st.phase == PH_TRACK;
  p.det_mode[s] = det_mode;
  p.det_maxc[s] = det_mode == 1 ? 2 * p.cam.gftt_num : p.cam.gftt_num;
  // goodFeaturesToTrack runs beside the tracking chain: planned here for init frames (detect, 2x corners) and for
  // tracking frames (redetect); the result of a frame whose tracking fails is simply not consumed
  p.gftt_act[s] = (det_mode == 1 || st.phase == PH_TRACK) ? 1 : 0;
  p.gftt_maxc[s] = det_mode == 1 ? 2 * p.cam.gftt_num : p.cam.gftt_num;
  p.n_exist[s] = 0;
  p.img_slot[s] = c;  // image slot written this frame (stays valid even if the frame is escaped later)
  p.lk_count[s] = 0;
}

// ------------------------------------------------------------------------------------------------ temporal LK inputs
__device__ __forceinline__ void track_prepare_dev(const Pipe& p, int s, int i) {
  const StreamState& st = p.st[s];
  if (st.phase != PH_TRACK) return;
  const int last = st.cur ^ 1;
  const int n = st.n_lm[last];
  if (i == 0) {
    p.lk_count[s] = n;
    p.lk_tag[s] = st.frame_id[last];  // the templates come from the last frame's left image
  }
  if (i >= n) return;
  const Landmark& lm = lm_ptr(p, last, s)[i];
  float px = (float)lm.p2d[0], py = (float)lm.p2d[1];
  // template cache: was the landmark's slot written for this very pixel of the last frame's left image? (the LK wave then needs one
  // load instead of a chain of dependent ones)
  p.lk_slot[(size_t)s * NMAX + i] = lk_tc_lookup(p.tc, p.tc_cap, p.tc_stride, s, lm.tslot, px, py, st.frame_id[last]);
  float* pp = p.prev_pts + ((size_t)s * NMAX + i) * 2;
  float* np = p.next_pts + ((size_t)s * NMAX + i) * 2;
  pp[0] = px;
  pp[1] = py;
  if (st.use_guess) {
    SE3d g = load_pose7(st.guess);
    float p3[3] = {(float)lm.p3w[0], (float)lm.p3w[1], (float)lm.p3w[2]};
    if (p.cam.cam_type == CAM_DEPTH) {  // lkorb_tracking.cpp:41-52: pinhole projection of the float-narrowed landmark
      const V3 pc = se3_act(g, V3{(double)p3[0], (double)p3[1], (double)p3[2]});
      np[0] = (float)(p.cam.fx * pc.x / pc.z + p.cam.cx);
      np[1] = (float)(p.cam.fy * pc.y / pc.z + p.cam.cy);
    } else {
      project_point(p3, q_to_mat(g.q), g.t, p.cam.K0, p.cam.D0, np);
    }
  } else {
    np[0] = px;
    np[1] = py;
  }
}

__global__ __launch_bounds__(256) void k_track_prepare(Pipe p) { track_prepare_dev(p, blockIdx.y, blockIdx.x * 256 + threadIdx.x); }
// k_frame_head and k_track_prepare in one launch (one workgroup of 256 threads per stream): the tracker's inputs follow the frame set-up
// after a barrier instead of after a launch -- one dependent launch less at the head of every frame's chain
__device__ __forceinline__ void k_frame_head_prepare_body(const Pipe& p, const double* __restrict__ frame_time, long long* __restrict__ host_progress,
                                                            long long frame_no) {
  chain_priority();
  const int s = blockIdx.x, tid = threadIdx.x;
  if (s == 0 && tid == 0 && host_progress) __hip_atomic_store(host_progress, frame_no, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  __shared__ double s_in[IMU_MAX * 7];
  int n = p.n_imu[s];
  if (n > IMU_MAX) n = IMU_MAX;
  const double* in = p.imu_in + (size_t)s * IMU_MAX * 7;
  for (int i = tid; i < 7 * n; i += 256) s_in[i] = in[i];
  const double t_frame = frame_time[s];
  __syncthreads();
  if (tid == 0) {
    imu_feed_dev(p, s, s_in);
    frame_begin_dev(p, s, t_frame);
  }
  __syncthreads();  // (workgroup-scope release / acquire: the stream's state as thread 0 left it)
  for (int i = tid; i < NMAX; i += 256) track_prepare_dev(p, s, i);
}
__global__ __launch_bounds__(256) void k_frame_head_prepare(Pipe p, const double* __restrict__ frame_time, long long* __restrict__ host_progress,
                                                            long long frame_no) {
  kj_wait(p.kj);
  k_frame_head_prepare_body(p, frame_time, host_progress, frame_no);
  kj_signal(p.kj);
  kj_post_wait(p.kj);
}

// ------------------------------------------------------------------------------------------------ LK survivors
// lkorb_tracking.cpp:93-125: survivors are appended in DESCENDING index order (quirk A1); the parallel from_* arrays
// keep ascending order.  Entered by a whole workgroup of T threads (a multiple of 64): every wave takes chunks of 64 landmarks, the chunks'
// survivor counts meet in LDS.  k_track_collect (one wave per stream) is the stand-alone launch; since round 6 the work is the prologue
// of k_ransac_f (FLVIS_CHAIN_MERGE bit 1: one launch less on the frame's chain, sixteen waves instead of one).
template <int T>
__device__ __forceinline__ void track_collect_dev(const Pipe& p, int s, int* s_cnt /* [NMAX / 64 + 1] LDS */) {
  StreamState& st = p.st[s];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (st.phase != PH_TRACK) {
    if (p.tpl_ahead && tid == 0) p.tpl_count[s] = 0;
    return;
  }
  const int cur = st.cur, last = cur ^ 1;
  const int n = st.n_lm[last];
  const Landmark* from = lm_ptr(p, last, s);
  Landmark* to = lm_ptr(p, cur, s);
  const float* tr = p.next_pts + (size_t)s * NMAX * 2;
  float* const tpl = p.tpl_ahead ? p.tpl_pts + (size_t)s * NMAX * 2 : nullptr;
  const uint8_t* status = p.lk_status + (size_t)s * NMAX;
  const int w = p.cam.w - 1, h = p.cam.h - 1;
  constexpr int NCH = NMAX / 64;
  // survivors per chunk of 64
  for (int c = wv; c < NCH; c += T / 64) {
    const int i = 64 * c + lane;
    const bool pass = i < n && status[i] == 1 && tr[2 * i] > 0 && tr[2 * i + 1] > 0 && tr[2 * i] < (float)w && tr[2 * i + 1] < (float)h;
    const int cnt = __popcll(__ballot(pass));
    if (lane == 0) s_cnt[c] = cnt;
  }
  __syncthreads();
  int total = 0;
#pragma unroll
  for (int c = 0; c < NCH; c++) total += s_cnt[c];
  for (int c = wv; c < NCH; c += T / 64) {
    if (64 * c >= n) break;
    int before = 0;  // survivors with smaller index than this chunk
    for (int k = 0; k < c; k++) before += s_cnt[k];
    const int i = 64 * c + lane;
    const bool pass = i < n && status[i] == 1 && tr[2 * i] > 0 && tr[2 * i + 1] > 0 && tr[2 * i] < (float)w && tr[2 * i + 1] < (float)h;
    const unsigned long long b = __ballot(pass);
    if (pass) {
      int k = before + lane_prefix(b);  // ascending rank
      int j = total - 1 - k;            // position in to.landmarks (descending)
      Landmark lm = from[i];
      float tx = tr[2 * i], ty = tr[2 * i + 1];
      float und[2] = {tx, ty};
      float fu[2];
      if (p.cam.cam_type != CAM_STEREO_UNRECT) {  // STEREO_RECT and DEPTH_D435 (lkorb_tracking.cpp:76-85)
        fu[0] = (float)lm.p2d[0];  // from_p2d_undistort = from_p2d_plane
        fu[1] = (float)lm.p2d[1];
      } else {
        float src[2] = {tx, ty};
        undistort_point(src, p.cam.K0, p.cam.D0, p.cam.R0, p.cam.P0, und);
        fu[0] = (float)lm.p2u[0];
        fu[1] = (float)lm.p2u[1];
      }
      lm.p2d[0] = (double)tx;
      lm.p2d[1] = (double)ty;
      lm.p2u[0] = (double)und[0];
      lm.p2u[1] = (double)und[1];
      if (tpl) {  // the templates of this pixel of the frame's left image go to cache slot j (k_lk_templates_ahead)
        lm.tslot = j < p.tc_cap ? (short)j : (short)-1;
        tpl[2 * j] = tx;
        tpl[2 * j + 1] = ty;
      }
      to[j] = lm;
      float* m1 = p.m1 + ((size_t)s * NMAX + k) * 2;
      float* m2 = p.m2 + ((size_t)s * NMAX + k) * 2;
      m1[0] = fu[0];
      m1[1] = fu[1];
      m2[0] = und[0];
      m2[1] = und[1];
    }
  }
  if (tid == 0) {
    st.n_lm[cur] = total;
    st.n_surv = total;
    st.of_cnt = total;
    if (total < 10) st.ok = 0;
    if (tpl) {
      p.tpl_count[s] = total < 10 ? 0 : total;  // (a frame that fails here never reaches the stereo matcher)
      p.tpl_tag[s] = st.frame_id[cur];
    }
  }
}
__global__ __launch_bounds__(64) void k_track_collect(Pipe p) {
  chain_priority();
  __shared__ int s_cnt[NMAX / 64 + 1];
  track_collect_dev<64>(p, blockIdx.x, s_cnt);
}

// ------------------------------------------------------------------------------------------------ F-matrix RANSAC
// cv::findFundamentalMat(FM_RANSAC, 5.0, 0.99) (lkorb_tracking.cpp:134-135; fundam.cpp + ptsetreg.cpp): from 15 points on the
// RANSAC registrator, with 8 .. 14 points the LMedS registrator, both drawing their 7-point subsets from cv::RNG((uint64)-1) in
// getSubset's order (repeated index redrawn per slot, a subset with a collinear last point redrawn as a whole).  One workgroup of RF_T
// threads per stream.  The subsets of a batch (16 hypotheses first -- the adaptive stop usually ends the search there -- then 64) are
// drawn SERIALLY, as the generator demands, by wave 0 running the draw loop wave-uniformly (the collinearity test of a subset is
// spread over 30 lanes); wave 0 then solves the 7-point systems (one per lane, up to 3 models each -> LDS), ALL waves score the
// models (a wave takes a model, its lanes stride the correspondences, ballot-popcount counts the inliers) and the adaptive stop is
// replayed sequentially over the batch.  Only the mask is used (lkorb_tracking.cpp:133-158).
// The subsets FMEstimatorCallback's RANSAC draws depend on the correspondences only through checkSubset (a collinear last point, rare):
// every run starts from cv::RNG((uint64)-1), so the CANDIDATES of the first batch -- 16 subsets of 7, each drawn as if every earlier
// one had been accepted -- are a function of the point count alone.  They are tabulated at start-up (host, the same generator and
// getSubset loop) with the generator state behind each; the kernel tests the 16 candidates and only falls back to the serial draw
// loop (wave 0 drawing, fifteen waves waiting: 40 us of a 70 us kernel) from the first refused candidate on.
constexpr int F_TAB_N = 1025;  // counts 0 .. 1024 (NMAX)
constexpr int F_TAB_B = 16;
__device__ unsigned short g_f_sub7[F_TAB_N][F_TAB_B][7];
__device__ unsigned long long g_f_rng7[F_TAB_N][F_TAB_B];

#ifdef FLVIS_RANSAC_PROF
#define RPROF(base, i)                                                                                   \
  do {                                                                                                   \
    if (threadIdx.x == 0 && p.counters) {                                                                \
      long long now_ = (long long)wall_clock64();                                                        \
      atomicAdd((unsigned long long*)&p.counters[(base) + (i)], (unsigned long long)(now_ - tlast_));   \
      tlast_ = now_;                                                                                     \
    }                                                                                                    \
  } while (0)
// (a second clock for stamps inside a phase of the outer profile)
#define RPROF2(base, i)                                                                                  \
  do {                                                                                                   \
    if (threadIdx.x == 0 && p.counters) {                                                                \
      long long now_ = (long long)wall_clock64();                                                        \
      atomicAdd((unsigned long long*)&p.counters[(base) + (i)], (unsigned long long)(now_ - tlast2_));  \
      tlast2_ = now_;                                                                                    \
    }                                                                                                    \
  } while (0)
#else
#define RPROF(base, i) do { } while (0)
#define RPROF2(base, i) do { } while (0)
#endif

#ifndef FLVIS_RF_T
#define FLVIS_RF_T 1024
#endif
constexpr int RF_T = FLVIS_RF_T;  // (A/B knob: 256 or 512 threads leave room for the detection stream's waves on the workgroup's CU)
template <bool COLLECT>
__device__ __forceinline__ void k_ransac_f_body(const Pipe& p) {
  chain_priority();
  const int s = blockIdx.x;
  StreamState& st = p.st[s];
  if (COLLECT) {  // k_track_collect's work as this launch's prologue: the survivors, m1 / m2, the stream's counts
    __shared__ int s_cnt[NMAX / 64 + 1];
    track_collect_dev<RF_T>(p, s, s_cnt);
    __syncthreads();  // (workgroup scope: what thread 0 and the copying lanes stored is what everybody reads below)
  }
  if (st.phase != PH_TRACK || !st.ok) return;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
#ifdef FLVIS_RANSAC_PROF
  long long tlast_ = (long long)wall_clock64();
  long long tlast2_ = tlast_;
  if (tid == 0 && p.counters) atomicAdd((unsigned long long*)&p.counters[24 + 7], 1ull);
#endif
  const int n = st.n_surv;
  __shared__ float sm1[NMAX * 2], sm2[NMAX * 2];
  __shared__ double Fm[64 * 3][9];
#ifdef FLVIS_SOLVERS_PRODUCT
  __shared__ double spw[63 * 64];  // 7-point workspaces of the 64 hypothesis lanes (element-major: conflict-free)
#else
  // run7Point's workspaces of the 64 hypotheses: element e of hypothesis h at spw[e * SPW_S + h].  (A/B knob FLVIS_SPW_S: 65 would keep the three lanes of a
  // hypothesis, which work on rows 9 elements apart, out of one LDS bank -- measured: 59.5k / 59.7k against 59.8k frames/s, no gain)
#ifndef FLVIS_SPW_S
#define FLVIS_SPW_S 64
#endif
  constexpr int SPW_S = FLVIS_SPW_S;
  __shared__ double spw[cvs::SP_WORK * SPW_S];
#endif
  __shared__ int hnm[64], mcnt[64 * 3];
  __shared__ int hcnt[64], hmodel[64];
  __shared__ int s_sub[64][8];  // the batch's subsets (7 indices each)
  __shared__ double bestF[9];
  __shared__ int ctl[4];  // niters, maxGood, best_iter, best_model
  const float* gm1 = p.m1 + (size_t)s * NMAX * 2;
  const float* gm2 = p.m2 + (size_t)s * NMAX * 2;
  for (int i = tid; i < 2 * n; i += RF_T) {
    sm1[i] = gm1[i];
    sm2[i] = gm2[i];
  }
  if (tid == 0) {
    ctl[0] = 1000;
    ctl[1] = 0;
    ctl[2] = -1;
    ctl[3] = 0;
  }
  __syncthreads();
  const float thr2 = 25.0f;
  Landmark* to = lm_ptr(p, st.cur, s);
  RPROF(24, 0);
  CvRng rng = cv_rng_init();  // RNG rng((uint64)-1) of RANSACPointSetRegistrator::run / LMeDSPointSetRegistrator::run
  const ModC mc = mod_c_make((uint32_t)(n > 0 ? n : 1));
  // FMEstimatorCallback::checkSubset = !haveCollinearPoints(m1) && !haveCollinearPoints(m2) (fundam.cpp): the LAST point of the subset
  // against the 15 pairs of earlier ones, on the Point2f coordinates; lanes 0 .. 29 take one (pair, image) each
  int* s_chk = s_sub[0];
  // (the subset to be tested is in s_chk)
  auto fm_check_staged = [&]() -> bool {
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    bool bad = false;
    if (lane < 30) {
      const int pr = lane < 15 ? lane : lane - 15;
      int j = 1, k = pr;
      while (k >= j) {
        k -= j;
        j++;
      }
      const float* m = lane < 15 ? sm1 : sm2;
      const int ii = s_chk[6], ij = s_chk[j], ik = s_chk[k];
      const double dx1 = m[2 * ij] - m[2 * ii], dy1 = m[2 * ij + 1] - m[2 * ii + 1];
      const double dx2 = m[2 * ik] - m[2 * ii], dy2 = m[2 * ik + 1] - m[2 * ii + 1];
      bad = fabs(dx2 * dy1 - dy2 * dx1) <= 1.1920928955078125e-07 * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2));
    }
    return __ballot(bad) == 0ull;
  };
  auto fm_check = [&](const int* idx) -> bool {
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
#pragma unroll
      for (int j = 0; j < 7; j++) s_chk[j] = idx[j];
    }
    return fm_check_staged();
  };
  // draws the subsets of hypotheses [base, base + B) that lie below the current iteration limit into s_sub (wave 0, uniform);
  // a failed draw (10000 refused subsets) marks the hypothesis and ends the batch -- the reference loop stops there
  auto draw_batch = [&](int base, int B, int limit, int max_attempts, bool tabulated) {
    s_sub[lane][7] = 0;
    __builtin_amdgcn_wave_barrier();
    int k = 0, first_attempts = 0;
    if (tabulated) {
      // the tabulated candidates of the first batch: all of them are tested; up to the first refused one they ARE the run's subsets
      if (lane < B) {
#pragma unroll
        for (int j = 0; j < 7; j++) s_sub[lane][j] = g_f_sub7[n][lane][j];
      }
      // (round 6: the B x 30 collinearity tests of the table's candidates in ceil(30 B / 64) passes of the wave instead of one candidate
      // after the other -- the serial form was ~20 us of the kernel; the first refused candidate ends the table's part as before)
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      unsigned refused = 0;  // bit c: candidate c has a collinear last point (wave-uniform)
      const int nchk = 30 * (B < 32 ? B : 32);
      for (int c0 = 0; c0 < nchk; c0 += 64) {
        const int c = c0 + lane;
        bool bad = false;
        int cand = 0;
        if (c < nchk) {
          cand = c / 30;
          const int t = c - 30 * cand;
          const int pr = t < 15 ? t : t - 15;
          int j = 1, kk = pr;
          while (kk >= j) {
            kk -= j;
            j++;
          }
          const float* m = t < 15 ? sm1 : sm2;
          const int ii = s_sub[cand][6], ij = s_sub[cand][j], ik = s_sub[cand][kk];
          const double dx1 = m[2 * ij] - m[2 * ii], dy1 = m[2 * ij + 1] - m[2 * ii + 1];
          const double dx2 = m[2 * ik] - m[2 * ii], dy2 = m[2 * ik + 1] - m[2 * ii + 1];
          bad = fabs(dx2 * dy1 - dy2 * dx1) <= 1.1920928955078125e-07 * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2));
        }
        unsigned long long bm = __ballot(bad);
        while (bm) {  // (rare: a refused candidate)
          const int l = __builtin_ctzll(bm);
          bm &= bm - 1;
          refused |= 1u << __shfl(cand, l);
        }
      }
      for (k = 0; k < B && base + k < limit; k++) {
        if (k < 32 ? ((refused >> k) & 1u) : 0u) break;
        if (lane == 0) s_sub[k][7] = 1;
      }
      if (k > 0) rng.state = g_f_rng7[n][k - 1];
      if (k < B && base + k < limit) {  // slot k: its candidate was attempt 0, refused; the generator is behind it
        rng.state = g_f_rng7[n][k];
        first_attempts = 1;
      }
    }
    for (; k < B && base + k < limit; k++) {
      int idx[7];
      s_chk = s_sub[k];
      const bool ok = cv_get_subset<7>(rng, mc, 7, max_attempts, idx, fm_check, first_attempts);
      first_attempts = 0;
      if (lane == 0) s_sub[k][7] = ok ? 1 : 0;
      if (!ok) break;
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  if (n > 7 && n < 15) {
    // ---- LMeDSPointSetRegistrator::run: max(RANSACUpdateNumIters(0.99, 0.45, 7, 1000), 3) subsets, the model with the smallest
    // median error wins, the inliers are the points within sigma = 2.5 * 1.4826 * (1 + 5 / (n - 7)) * sqrt(median) (>= 0.001) of it.
    // n <= 14: a lane scores its own models (errors, insertion sort, median); thread 0 replays the running minimum in order.
    __shared__ double s_med[64 * 3];
    __shared__ double s_lm[2];
    int niters = ransac_update_num_iters(0.99, 0.45, 7, 1000);
    niters = niters > 3 ? niters : 3;
    if (tid == 0) {
      s_lm[0] = 1.7976931348623157e308;  // minMedian
      ctl[2] = -1;
    }
    __syncthreads();
    bool stop = false;
    for (int base = 0; base < niters && !stop; base += 64) {
#ifndef FLVIS_SOLVERS_PRODUCT
      int nm = -1;
      double* const xw = spw + lane;
      if (wv == 0) {
        draw_batch(base, 64, niters, 1000, false);
        const int iter = base + lane;
        double F[3][9];
        if (iter < niters) {
          if (s_sub[lane][7]) {
            double x1[7][2], x2[7][2];
#pragma unroll
            for (int k = 0; k < 7; k++) {
              const int ik = s_sub[lane][k];
              x1[k][0] = sm1[2 * ik];
              x1[k][1] = sm1[2 * ik + 1];
              x2[k][0] = sm2[2 * ik];
              x2[k][1] = sm2[2 * ik + 1];
            }
            nm = cvs::run7point<SPW_S>(x1, x2, xw, F, [](int) {});
          } else {
            nm = -2;
          }
        }
        if (nm > 0) {
          for (int m = 0; m < nm; m++) {
            // `std::sort(errf.ptr<int>(), errf.ptr<int>() + count)`: the float errors ordered through their bit patterns
            int err[14];
#pragma unroll
            for (int i = 0; i < 14; i++)
              err[i] = i < n ? __float_as_int(f_error(F[m], sm1[2 * i], sm1[2 * i + 1], sm2[2 * i], sm2[2 * i + 1])) : 0x7fffffff;
            // (padding sorts last): odd-even transposition network on registers
#pragma unroll
            for (int pass = 0; pass < 14; pass++)
#pragma unroll
              for (int i = pass & 1; i + 1 < 14; i += 2) {
                const int a = err[i], b = err[i + 1];
                err[i] = a < b ? a : b;
                err[i + 1] = a < b ? b : a;
              }
            float e_lo = 0.f, e_hi = 0.f, e_mid = 0.f;
#pragma unroll
            for (int i = 0; i < 14; i++) {
              if (i == n / 2 - 1) e_lo = __int_as_float(err[i]);
              if (i == n / 2) e_hi = e_mid = __int_as_float(err[i]);
            }
            s_med[lane * 3 + m] = (n & 1) ? (double)e_mid : ((double)(float)(e_lo + e_hi)) * 0.5;
            for (int j = 0; j < 9; j++) Fm[lane * 3 + m][j] = F[m][j];
          }
        }
#else
      SevenPointMid sp_mid;
      PolyBracket sp_t;
      double sp_roots[4];
      int sp_nr = 0, sp_mode = -1;
      int nm = -1;
      double* const xw = spw + lane;
      if (wv == 0) {
        draw_batch(base, 64, niters, 1000, false);
        const int iter = base + lane;
        if (iter < niters) {
          if (s_sub[lane][7]) {
            double x1[7][2], x2[7][2];
#pragma unroll
            for (int k = 0; k < 7; k++) {
              const int ik = s_sub[lane][k];
              x1[k][0] = sm1[2 * ik];
              x1[k][1] = sm1[2 * ik + 1];
              x2[k][0] = sm2[2 * ik];
              x2[k][1] = sm2[2 * ik + 1];
            }
            double cc[4];
            nm = 0;
            if (seven_point_a<64>(x1, x2, xw, sp_mid, cc)) sp_mode = poly_cubic_prepare(cc, sp_t, sp_roots, sp_nr);
          } else {
            nm = -2;
          }
        }
        if (sp_mode == 1) {
          // (one lane bisects its own intervals here: the LMedS branch is the rare low-feature case, not worth the wave hand-off)
          bool bis[4] = {false, false, false, false};
          double mid[4] = {0, 0, 0, 0};
          poly_bracket_bisect(sp_t, 0, bis[0], mid[0]);
          poly_bracket_bisect(sp_t, 1, bis[1], mid[1]);
          poly_bracket_bisect(sp_t, 2, bis[2], mid[2]);
          sp_nr = poly_bracket_emit(sp_t, bis, mid, sp_roots);
        }
        if (sp_mode >= 0) {
          double F[3][9];
          nm = seven_point_b(sp_mid, sp_roots, sp_nr, F);
          for (int m = 0; m < nm; m++) {
            // `std::sort(errf.ptr<int>(), errf.ptr<int>() + count)`: the float errors ordered through their bit patterns
            int err[14];
#pragma unroll
            for (int i = 0; i < 14; i++)
              err[i] = i < n ? __float_as_int(f_error(F[m], sm1[2 * i], sm1[2 * i + 1], sm2[2 * i], sm2[2 * i + 1])) : 0x7fffffff;
            // (padding sorts last): odd-even transposition network on registers
#pragma unroll
            for (int pass = 0; pass < 14; pass++)
#pragma unroll
              for (int i = pass & 1; i + 1 < 14; i += 2) {
                const int a = err[i], b = err[i + 1];
                err[i] = a < b ? a : b;
                err[i + 1] = a < b ? b : a;
              }
            float e_lo = 0.f, e_hi = 0.f, e_mid = 0.f;
#pragma unroll
            for (int i = 0; i < 14; i++) {
              if (i == n / 2 - 1) e_lo = __int_as_float(err[i]);
              if (i == n / 2) e_hi = e_mid = __int_as_float(err[i]);
            }
            s_med[lane * 3 + m] = (n & 1) ? (double)e_mid : ((double)(float)(e_lo + e_hi)) * 0.5;
            for (int j = 0; j < 9; j++) Fm[lane * 3 + m][j] = F[m][j];
          }
        }
#endif
        hnm[lane] = nm;
      }
      __syncthreads();
      if (tid == 0) {
        for (int k = 0; k < 64 && base + k < niters; k++) {
          if (hnm[k] == -2) {  // getSubset failed: `if (iter == 0) return false; break;`
            stop = true;
            break;
          }
          for (int m = 0; m < hnm[k]; m++)
            if (s_med[k * 3 + m] < s_lm[0]) {
              s_lm[0] = s_med[k * 3 + m];
              ctl[2] = base + k;
              for (int j = 0; j < 9; j++) bestF[j] = Fm[k * 3 + m][j];
            }
        }
        ctl[3] = stop ? 1 : 0;
      }
      __syncthreads();
      stop = ctl[3] != 0;
    }
    if (ctl[2] >= 0) {
      double sigma = 2.5 * 1.4826 * (1 + 5. / (n - 7)) * sqrt(s_lm[0]);
      sigma = fmax(sigma, 0.001);
      const float tl = (float)(sigma * sigma);
      for (int i = tid; i < n; i += RF_T) {
        bool in = f_error(bestF, sm1[2 * i], sm1[2 * i + 1], sm2[2 * i], sm2[2 * i + 1]) <= tl;
        if (!in) to[i].inlier = 0;  // mask index i applied to to.landmarks[i] (descending order): quirk A1
      }
    } else {
      for (int i = tid; i < n; i += RF_T) to[i].inlier = 0;
    }
  } else if (n > 7) {
    for (int base = 0, B = 16; base < ctl[0]; base += B, B = 64) {
#ifdef FLVIS_RANSAC_PROF
      if (tid == 0 && p.counters) atomicAdd((unsigned long long*)&p.counters[24 + 6], 1ull);
#endif
#ifndef FLVIS_SOLVERS_PRODUCT
      // hypotheses of this batch: cv::run7Point (cv_solvers.hpp).  Four lanes per hypothesis, sixteen hypotheses per wave (the first batch is
      // one wave's work): the one-sided Jacobi SVD of the 7 x 9 system lives in the hypothesis' LDS workspace and its rotations run on the
      // anti-diagonals of two overlapping sweeps, three pairs at a time (cvs::sp_slot: the bits of the cyclic order, in a third of its
      // steps -- a rotation is a chain of three divisions and three square roots, ~0.4 us); lane 0 of the four then completes the basis,
      // solves the cubic and writes up to three matrices.
      if (wv == 0) draw_batch(base, B, ctl[0], 10000, base == 0 && B == F_TAB_B && n < F_TAB_N);
      RPROF2(40, 5);
      __syncthreads();
      RPROF2(40, 6);
      if (16 * wv < B) {
        const int q = lane & 3, hyp = 16 * wv + (lane >> 2);
        const int iter = base + hyp;
        int nm = -1;  // -1: beyond niters, -2: subset impossible (the reference loop stops)
        bool have = false;
        double* const xw = spw + hyp;  // element e of this hypothesis at xw[e * SPW_S]
        if (iter < ctl[0]) {
          if (s_sub[hyp][7]) have = true;
          else nm = -2;
        }
        if (have && q == 0) {
          double x1[7][2], x2[7][2];
#pragma unroll
          for (int k = 0; k < 7; k++) {
            const int ik = s_sub[hyp][k];
            x1[k][0] = sm1[2 * ik];
            x1[k][1] = sm1[2 * ik + 1];
            x2[k][0] = sm2[2 * ik];
            x2[k][1] = sm2[2 * ik + 1];
          }
          cvs::sp_fill<SPW_S>(x1, x2, xw);
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        RPROF2(40, 0);
        bool done = !have, chg_prev = false, chg_cur = false;
        const int sh = lane & ~3;
        for (int T = 1;; T++) {
          const int s_hi = (T - 1) / 7, sigma = T - 7 * s_hi;
          const cvs::SpSlot e = cvs::sp_slot(sigma, q);
          const int sw = s_hi + e.ds;
          bool rot = false;
          if (!done && e.i >= 0 && sw >= 0 && sw < cvs::SVD_MAX_SWEEPS) rot = cvs::sp_pair<SPW_S>(xw, e.i, e.j);
          __builtin_amdgcn_wave_barrier();
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          const unsigned long long bc = __ballot(rot && e.ds == 0), bp = __ballot(rot && e.ds != 0);
          chg_cur = chg_cur || ((bc >> sh) & 0xFull) != 0;
          chg_prev = chg_prev || ((bp >> sh) & 0xFull) != 0;
          if (sigma == 4 && s_hi >= 1 && (!chg_prev || s_hi == cvs::SVD_MAX_SWEEPS)) done = true;  // sweep s_hi - 1: unchanged, or the last
          if (sigma == 7) {
            chg_prev = chg_cur;
            chg_cur = false;
          }
          if (__ballot(!done) == 0ull) break;
        }
        RPROF2(40, 1);
        if (have && q == 0) {
          double F[3][9];
          nm = cvs::sp_finish<SPW_S>(xw, F, [&](int) {});
#pragma unroll
          for (int m = 0; m < 3; m++)
            if (m < nm)
#pragma unroll
              for (int j = 0; j < 9; j++) Fm[hyp * 3 + m][j] = F[m][j];
        }
        if (q == 0) hnm[hyp] = nm;
        RPROF2(40, 2);
      }
#else
      // hypotheses of this batch, one per lane of wave 0.  The bisections of the cubic's (up to three) sign-change intervals are
      // handed to waves 0..2, one interval each (a lane's whole bracketing level inside wave 0 was a third of the generation time:
      // a single wave is VALU-issue bound); the level travels through the per-lane workspace, which is free after the elimination
      SevenPointMid sp_mid;
      PolyBracket sp_t;
      double sp_roots[4];
      int sp_nr = 0, sp_mode = -1;  // -1: no polynomial, 0: roots final, 1: level to be bisected
      int nm = -1;                  // -1: beyond niters, -2: subset impossible (the reference loop stops)
      double* const xw = spw + lane;  // element e of this lane at xw[e * 64]
      if (wv == 0) {
        draw_batch(base, B, ctl[0], 10000, base == 0 && B == F_TAB_B && n < F_TAB_N);
        const int iter = base + lane;
        if (lane < B && iter < ctl[0]) {
#ifdef FLVIS_RANSAC_PROF
          if (tid == 0) {
            g_sp_prof = p.counters ? p.counters + 40 : nullptr;
            g_sp_last = (long long)wall_clock64();
          }
#endif
          if (s_sub[lane][7]) {
            double x1[7][2], x2[7][2];
#pragma unroll
            for (int k = 0; k < 7; k++) {
              const int ik = s_sub[lane][k];
              x1[k][0] = sm1[2 * ik];
              x1[k][1] = sm1[2 * ik + 1];
              x2[k][0] = sm2[2 * ik];
              x2[k][1] = sm2[2 * ik + 1];
            }
            double cc[4];
            nm = 0;
            if (seven_point_a<64>(x1, x2, xw, sp_mid, cc)) sp_mode = poly_cubic_prepare(cc, sp_t, sp_roots, sp_nr);
          } else {
            nm = -2;
          }
        }
        if (sp_mode == 1) {
          xw[0 * 64] = sp_t.a0, xw[1 * 64] = sp_t.a1, xw[2 * 64] = sp_t.a2, xw[3 * 64] = sp_t.a3, xw[4 * 64] = sp_t.a4;
          xw[5 * 64] = sp_t.k0, xw[6 * 64] = sp_t.k1, xw[7 * 64] = sp_t.k2, xw[8 * 64] = sp_t.k3, xw[9 * 64] = sp_t.k4;
        }
        xw[10 * 64] = sp_mode == 1 ? (double)sp_t.nk : 0.0;  // 0: nothing to bisect for this lane
      }
      __syncthreads();
      if (wv < 3 && xw[10 * 64] != 0.0) {
        PolyBracket t;
        t.a0 = xw[0 * 64], t.a1 = xw[1 * 64], t.a2 = xw[2 * 64], t.a3 = xw[3 * 64], t.a4 = xw[4 * 64];
        t.k0 = xw[5 * 64], t.k1 = xw[6 * 64], t.k2 = xw[7 * 64], t.k3 = xw[8 * 64], t.k4 = xw[9 * 64];
        t.nk = (int)xw[10 * 64];
        bool bis;
        double root;
        poly_bracket_bisect(t, wv, bis, root);
        xw[(11 + wv) * 64] = root;
        xw[(15 + wv) * 64] = bis ? 1.0 : 0.0;
      }
      __syncthreads();
      if (wv == 0) {
        if (sp_mode == 1) {
          const bool bis[4] = {xw[15 * 64] != 0.0, xw[16 * 64] != 0.0, xw[17 * 64] != 0.0, false};
          const double mid[4] = {xw[11 * 64], xw[12 * 64], xw[13 * 64], 0.0};
          sp_nr = poly_bracket_emit(sp_t, bis, mid, sp_roots);
        }
        SP_STAMP(3);
        if (sp_mode >= 0) {
          double F[3][9];
          nm = seven_point_b(sp_mid, sp_roots, sp_nr, F);
          SP_STAMP(4);
          for (int m = 0; m < nm; m++)
            for (int j = 0; j < 9; j++) Fm[lane * 3 + m][j] = F[m][j];
        }
        hnm[lane] = nm;
      }
#endif
      __syncthreads();
      RPROF(24, 1);
      // score + replay in sub-batches of 16 hypotheses: the adaptive stop usually ends the search within the first few hypotheses
      // (niters drops to ~8 once a model with 90 % inliers is seen), so the later models of the batch are never looked at
      constexpr int SB = 16;
      for (int sb = 0; sb < B; sb += SB) {
        if (base + sb >= ctl[0]) break;  // (uniform: ctl[0] was written before the last barrier)
        for (int mi = 3 * sb + wv; mi < 3 * (sb + SB); mi += RF_T / 64) {  // score the models
          const int hyp = mi / 3, m = mi - 3 * hyp;
          if (m >= hnm[hyp]) continue;
          double F[9];
#pragma unroll
          for (int j = 0; j < 9; j++) F[j] = Fm[mi][j];
          int good = 0;
          for (int i0 = 0; i0 < n; i0 += 64) {
            const int i = i0 + lane;
            const bool in = i < n && f_error(F, sm1[2 * i], sm1[2 * i + 1], sm2[2 * i], sm2[2 * i + 1]) <= thr2;
            good += __popcll(__ballot(in));
          }
          if (lane == 0) mcnt[mi] = good;
        }
        __syncthreads();
        RPROF(24, 2);
        if (tid < SB) {  // best model of each hypothesis: first maximum in model order
          const int h = sb + tid;
          const int nm = hnm[h];
          int cnt = nm == -2 ? -2 : (nm < 0 ? -1 : 0), model = 0;
          for (int m = 0; m < nm; m++) {
            const int good = mcnt[h * 3 + m];
            if (good > cnt) {
              cnt = good;
              model = m;
            }
          }
          hcnt[h] = cnt;
          hmodel[h] = model;
        }
        __syncthreads();
        if (tid == 0) {
          int niters = ctl[0], maxGood = ctl[1];
          for (int k = sb; k < sb + SB; k++) {
            if (base + k >= niters) break;
            if (hcnt[k] == -2) {
              niters = 0;
              break;
            }
            int good = hcnt[k];
            int lim = maxGood > 6 ? maxGood : 6;
            if (good > lim) {
              maxGood = good;
              ctl[2] = base + k;
              ctl[3] = hmodel[k];
              for (int j = 0; j < 9; j++) bestF[j] = Fm[k * 3 + hmodel[k]][j];
              niters = ransac_update_num_iters(0.99, (double)(n - good) / n, 7, niters);
            }
          }
          ctl[0] = niters;
          ctl[1] = maxGood;
        }
        __syncthreads();
      }
      RPROF(24, 3);
    }
    // apply the winning model's mask with the reference's mirrored index
    if (ctl[2] >= 0) {
      for (int i = tid; i < n; i += RF_T) {
        bool in = f_error(bestF, sm1[2 * i], sm1[2 * i + 1], sm2[2 * i], sm2[2 * i + 1]) <= thr2;
        if (!in) to[i].inlier = 0;  // mask index i applied to to.landmarks[i] (descending order): quirk A1
      }
    } else {
      for (int i = tid; i < n; i += RF_T) to[i].inlier = 0;  // no model: all-zero mask
    }
  }
  __syncthreads();
  if (wv == 0) {
    int fc = 0;
    for (int i = lane; i < n; i += 64) fc += to[i].inlier ? 1 : 0;
    fc = wave_sum_i32(fc);
    if (lane == 0) {
      st.f_cnt = fc;
      if (fc < 10) st.ok = 0;
    }
  }
  RPROF(24, 4);
}
__global__ __launch_bounds__(RF_T) void k_ransac_f(Pipe p) {
  kj_wait(p.kj);
  k_ransac_f_body<false>(p);
  kj_signal(p.kj);
}
__global__ __launch_bounds__(RF_T) void k_collect_ransac_f(Pipe p) {
  kj_wait(p.kj);
  k_ransac_f_body<true>(p);
  kj_signal(p.kj);
}

// ------------------------------------------------------------------------------------------------ PnP RANSAC
// cv::solvePnPRansac(p3d, p2d, K_rect, 0, r, t, false, iterations, reprojErr, confidence, inliers, ITERATIVE|P3P) control flow: the
// RANSAC registrator of ptsetreg.cpp with 5-point (ITERATIVE: the kernel method is EPNP) or 4-point (P3P) subsets drawn from
// cv::RNG((uint64)-1) in getSubset's order, serially by wave 0; PnPRansacCallback has no checkSubset.
//   ITERATIVE  hypotheses by EPnP on the five sample points (epnp_core.hpp), one per wave, eight per batch; final solve on the inliers:
//              Gauss-Newton on the reprojection error from the winning model (the minimum cv's Levenberg-Marquardt converges to)
//   P3P        hypotheses by Grunert P3P on the first three sample points, the fourth picks among the <= 4 solutions (as cv::p3p does),
//              one per lane, 16 then 64 per batch; final solve on the inliers: EPnP (solvePnPRansac hands SOLVEPNP_EPNP to solvePnP)
// The core works on
// correspondences staged in LDS by the caller (the tracker's kernel gathers a frame's landmarks; the standalone kernel of the loop
// closing's geometric check loads caller arrays) and is entered by the WHOLE workgroup; only wave 0 returns with the result.
constexpr int RP_T = 512;
constexpr int PNP_GN_ROW = 29;
// PnPRansacCallback has no checkSubset and every run starts from cv::RNG((uint64)-1): the subsets of a run depend on the number of
// correspondences alone.  The first batch of either branch (8 subsets of 5 / 16 subsets of 4) is therefore tabulated per count at start-up
// (host, the same generator and getSubset loop), together with the generator state behind it, and a kernel whose search ends within the
// first batch -- the usual case -- never runs the serial draw loop (wave 0 drawing while seven waves wait).
constexpr int PNP_TAB_N = 1025;        // counts 0 .. 1024 (NMAX and PNP_MAXN)
constexpr int PNP_TAB_B5 = RP_T / 64;  // first batch of the EPnP branch
constexpr int PNP_TAB_B4 = 16;         // first batch of the P3P branch
__device__ unsigned short g_pnp_sub5[PNP_TAB_N][PNP_TAB_B5][5];
__device__ unsigned short g_pnp_sub4[PNP_TAB_N][PNP_TAB_B4][4];
__device__ unsigned long long g_pnp_rng5[PNP_TAB_N], g_pnp_rng4[PNP_TAB_N];
struct PnpShared {
  float* s2d;            // [n][2]
  float* s3d;            // [n][3]
  unsigned char* smask;  // [n] out: inlier mask of the winning model
  int* hcnt;             // [64]
  double (*hpose)[12];   // [64]
  int* ctl;              // [4]
  int (*sub)[8];         // [64] the batch's subsets (<= 5 indices, [7]: drawn)
  double* bpose;         // [12]
  double* gterms;        // [64 * PNP_GN_ROW]
  double* gn;            // [32]
  epnp::Work* ew;        // [RP_T / 64] one EPnP workspace per wave
  short* inl;            // [n] indices of the inliers, in order (the P3P flag's final EPnP)
};
template <bool PROF>
__device__ __forceinline__ void pnp_ransac_core(const PnpShared sh, const int np, const bool iterative, const SE3d guess, const double fx,
                                                const double fy, const double cx, const double cy, const unsigned long long seed,
                                                const int max_iters, const float t2, const double conf, long long* prof,
                                                long long& tlast_, SE3d& T, int& inliers) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  float* const s2d = sh.s2d;
  float* const s3d = sh.s3d;
  unsigned char* const smask = sh.smask;
  int* const hcnt = sh.hcnt;
  double(*const hpose)[12] = sh.hpose;
  int* const ctl = sh.ctl;
  double* const bpose = sh.bpose;
  double* const gterms = sh.gterms;
  double* const gn = sh.gn;
#define PNP_PROF(i)                                                                             \
  do {                                                                                          \
    if (PROF && threadIdx.x == 0 && prof) {                                                     \
      long long now_ = (long long)wall_clock64();                                               \
      atomicAdd((unsigned long long*)&prof[i], (unsigned long long)(now_ - tlast_));            \
      tlast_ = now_;                                                                            \
    }                                                                                           \
  } while (0)
  if (tid == 0) {
    ctl[0] = max_iters;
    ctl[1] = 0;
    ctl[2] = -1;
  }
  __syncthreads();
  const int modelPoints = iterative ? 5 : 4;
  PNP_PROF(0);
  int(*const s_sub)[8] = sh.sub;
  CvRng rng = cv_rng_init();  // RANSACPointSetRegistrator::run: RNG rng((uint64)-1)
  const ModC mc = mod_c_make((uint32_t)(np > 0 ? np : 1));
  if (np >= modelPoints) {
    // exactly model_points correspondences: solvePnPRansac hands them to solvePnP directly -- one hypothesis on the points themselves
    const int first_limit = np == modelPoints ? 1 : max_iters;
    if (tid == 0) ctl[0] = first_limit;
    __syncthreads();
    const epnp::Camera ecam{fx, fy, cx, cy};
    auto wsync = [] {
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
#ifdef FLVIS_EPNP_WAVES
    constexpr int NW = FLVIS_EPNP_WAVES;
#else
    constexpr int NW = RP_T / 64;
#endif
    for (int base = 0, B = iterative ? NW : 16; base < ctl[0]; base += B, B = iterative ? NW : 64) {
      if (PROF && tid == 0 && prof) atomicAdd((unsigned long long*)&prof[6], 1ull);
      if (iterative) {
        // EPnP on the five sample points: wave w solves hypothesis base + w
        if (wv == 0 && base == 0 && np > modelPoints && np < PNP_TAB_N) {
          if (lane < NW) {  // the tabulated first batch
#pragma unroll
            for (int j = 0; j < 5; j++) s_sub[lane][j] = g_pnp_sub5[np][lane][j];
            s_sub[lane][7] = 1;
          }
          rng.state = g_pnp_rng5[np];
        } else if (wv == 0) {
          if (lane < NW) s_sub[lane][7] = 0;
          __builtin_amdgcn_wave_barrier();
          for (int k = 0; k < B && base + k < ctl[0]; k++) {
            int sidx[5];
            bool ok = true;
            if (np == modelPoints) {
#pragma unroll
              for (int j = 0; j < 5; j++) sidx[j] = j;
            } else {
              ok = cv_get_subset<5>(rng, mc, modelPoints, 10000, sidx, [](const int*) { return true; });
            }
            if (lane == 0) {
#pragma unroll
              for (int j = 0; j < 5; j++) s_sub[k][j] = sidx[j];
              s_sub[k][7] = ok ? 1 : 0;
            }
            if (!ok) break;
          }
        }
        __syncthreads();
        int cnt = -1;  // -1: no model / beyond niters, -2: subset impossible, -3: model in hpose[wv], to be scored
        if (wv < NW && base + wv < ctl[0]) {
          if (s_sub[wv][7]) {
            const int* const sub = s_sub[wv];
            auto pw = [&](int i, double* q) {
              const int k = sub[i];
              q[0] = (double)s3d[3 * k], q[1] = (double)s3d[3 * k + 1], q[2] = (double)s3d[3 * k + 2];
            };
            auto uv = [&](int i, double* z) {  // undistortPoints' float, back to pixels as epnp::init_points does
              const int k = sub[i];
              z[0] = (double)(float)(((double)s2d[2 * k] - cx) / fx) * fx + cx;
              z[1] = (double)(float)(((double)s2d[2 * k + 1] - cy) / fy) * fy + cy;
            };
#ifdef FLVIS_RANSAC_PROF
            // phase times of wave 0's solve into prof[24 ..] (= counters[56 ..]; 40 .. 46 are the 7-point solver's)
            long long et_ = (long long)wall_clock64();
            auto emark = [&](int i) {
              if (PROF && tid == 0 && prof) {
                const long long now_ = (long long)wall_clock64();
                atomicAdd((unsigned long long*)&prof[24 + i], (unsigned long long)(now_ - et_));
                et_ = now_;
              }
            };
            epnp::Work& ew_ = sh.ew[wv];
            epnp::solve_head<64>(ew_, 5, pw, uv, ecam, nullptr, lane, wsync);
            emark(0);
            epnp::jacobi12(ew_, lane, 64, wsync);
            emark(1);
            epnp::phase_pick_vectors(ew_, lane, 64);
            wsync();
            epnp::phase_constraints(ew_, lane, 64);
            wsync();
            emark(2);
            epnp::phase_betas(ew_, lane, 64);
            wsync();
            emark(3);
            epnp::phase_centroids(ew_, 5, pw, nullptr, lane, 64, wsync);
            epnp::phase_abt(ew_, 5, pw, nullptr, lane, 64, wsync);
            emark(4);
            epnp::phase_pose(ew_, 5, pw, uv, ecam, nullptr, lane, 64, wsync);
            emark(5);
            const epnp::Pose P = epnp::result(ew_);
            if (PROF && tid == 0 && prof) atomicAdd((unsigned long long*)&prof[24 + 6], 1ull);
#else
            const epnp::Pose P = epnp::solve<64>(sh.ew[wv], 5, pw, uv, ecam, nullptr, lane, wsync);  // (a sample is one chunk: no scratch)
#endif
            if (P.ok) {
              if (lane == 0) {
#pragma unroll
                for (int j = 0; j < 9; j++) hpose[wv][j] = P.R[j];
                hpose[wv][9] = P.t[0], hpose[wv][10] = P.t[1], hpose[wv][11] = P.t[2];
              }
              cnt = -3;
            }
          } else {
            cnt = -2;
          }
        }
        if (lane == 0 && wv < NW) hcnt[wv] = cnt;
        __syncthreads();
      } else {
      // hypotheses of this batch, one per lane of wave 0: P3P on the first 3 sample points, the rest disambiguate.  The bisections of
      // the quartic's two bracketing levels (its derivative's <= 3 sign-change intervals, then its own <= 4) are handed to waves
      // 0..3, one interval each: inside one lane they were most of the generation time (a single wave is VALU-issue bound).  The
      // level travels through hpose[lane] (written only at the end of the generation), the results through gterms (used later).
      int cnt = -1;  // -1: no model / beyond niters, -2: subset impossible, -3: model in hpose[lane], to be scored
      int idx[5];
      bool have = false;
      P3PMid pm;
      PolyBracket lvl;
      double qa[5], c1[4], roots[4];
      int nc1 = 0, nr = 0, qmode = -1;  // poly_quartic_stage1's result; -1: no polynomial for this lane
      double* const task = hpose[lane];
      double* const res = gterms + lane * PNP_GN_ROW;
      auto put_task = [&](bool on) {
        if (on) {
          task[0] = lvl.a0, task[1] = lvl.a1, task[2] = lvl.a2, task[3] = lvl.a3, task[4] = lvl.a4;
          task[5] = lvl.k0, task[6] = lvl.k1, task[7] = lvl.k2, task[8] = lvl.k3, task[9] = lvl.k4;
        }
        task[10] = on ? (double)lvl.nk : 0.0;
      };
      auto run_task = [&](int n_intervals) {  // waves 0 .. n_intervals-1: interval wv of lane's level
        if (wv < n_intervals && task[10] != 0.0) {
          PolyBracket t;
          t.a0 = task[0], t.a1 = task[1], t.a2 = task[2], t.a3 = task[3], t.a4 = task[4];
          t.k0 = task[5], t.k1 = task[6], t.k2 = task[7], t.k3 = task[8], t.k4 = task[9];
          t.nk = (int)task[10];
          bool bis;
          double root;
          poly_bracket_bisect(t, wv, bis, root);
          res[wv] = root;
          res[4 + wv] = bis ? 1.0 : 0.0;
        }
      };
      if (wv == 0) {
        // the batch's subsets: the first batch from the table, later ones drawn serially (wave-uniform) below the current iteration limit
        s_sub[lane][7] = 0;
        __builtin_amdgcn_wave_barrier();
        if (base == 0 && !iterative && np > modelPoints && np < PNP_TAB_N) {
          if (lane < PNP_TAB_B4) {
#pragma unroll
            for (int j = 0; j < 4; j++) s_sub[lane][j] = g_pnp_sub4[np][lane][j];
            s_sub[lane][4] = -1;
            s_sub[lane][7] = 1;
          }
          rng.state = g_pnp_rng4[np];
        } else
        for (int k = 0; k < B && base + k < ctl[0]; k++) {
          int sidx[5];
          bool ok = true;
          if (np == modelPoints) {
#pragma unroll
            for (int j = 0; j < 5; j++) sidx[j] = j;
          } else {
            ok = cv_get_subset<5>(rng, mc, modelPoints, 10000, sidx, [](const int*) { return true; });
          }
          if (lane == 0) {
#pragma unroll
            for (int j = 0; j < 5; j++) s_sub[k][j] = sidx[j];
            s_sub[k][7] = ok ? 1 : 0;
          }
          if (!ok) break;
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int iter = base + lane;
        if (lane < B && iter < ctl[0]) {
          if (s_sub[lane][7]) {
#pragma unroll
            for (int j = 0; j < 5; j++) idx[j] = s_sub[lane][j];
            have = true;
#ifndef FLVIS_SOLVERS_PRODUCT
            // cv::solvePnP(SOLVEPNP_P3P) on the four sample points (cv_solvers.hpp): undistortPoints to normalised float coordinates,
            // mapped back with x * fx + cx by p3p::extract_points; Gao's solver on the first three, the fourth picks the pose
            const cvs::P3PCamera cam = cvs::p3p_camera(fx, fy, cx, cy);
            double uv[4][2], X[4][3], Rr[9], tr[3];
#pragma unroll
            for (int k = 0; k < 4; k++) {
              uv[k][0] = (double)(float)(((double)s2d[2 * idx[k]] - cx) / fx) * fx + cx;
              uv[k][1] = (double)(float)(((double)s2d[2 * idx[k] + 1] - cy) / fy) * fy + cy;
#pragma unroll
              for (int j = 0; j < 3; j++) X[k][j] = (double)s3d[3 * idx[k] + j];
            }
            if (cvs::p3p_solve4(cam, uv, X, Rr, tr)) {
#pragma unroll
              for (int j = 0; j < 9; j++) hpose[lane][j] = Rr[j];
              hpose[lane][9] = tr[0];
              hpose[lane][10] = tr[1];
              hpose[lane][11] = tr[2];
              cnt = -3;
            }
          } else {
            cnt = -2;
          }
        }
        hcnt[lane] = cnt;
      }
      __syncthreads();
#else
            V3 P[3], f[3];
            for (int k = 0; k < 3; k++) {
              P[k] = V3{(double)s3d[3 * idx[k]], (double)s3d[3 * idx[k] + 1], (double)s3d[3 * idx[k] + 2]};
              V3 d{((double)s2d[2 * idx[k]] - cx) / fx, ((double)s2d[2 * idx[k] + 1] - cy) / fy, 1.0};
              f[k] = (1.0 / norm(d)) * d;
            }
            double q[5];
            qmode = p3p_grunert_a(P, f, pm, q) ? poly_quartic_stage1(q, qa, lvl, c1, nc1, roots, nr) : 0;
          } else {
            cnt = -2;
          }
        }
        put_task(qmode == 1);
      }
      __syncthreads();
      run_task(3);
      __syncthreads();
      if (wv == 0) {
        if (qmode == 1) {
          const bool bis[4] = {res[4] != 0.0, res[5] != 0.0, res[6] != 0.0, false};
          const double mid[4] = {res[0], res[1], res[2], 0.0};
          nc1 = poly_bracket_emit(lvl, bis, mid, c1);
          qmode = 2;
        }
        if (qmode == 2) lvl = poly_quartic_stage2(qa, c1, nc1);
        put_task(qmode == 2);
      }
      __syncthreads();
      run_task(4);
      __syncthreads();
      if (wv == 0) {
        if (qmode == 2) {
          const bool bis[4] = {res[4] != 0.0, res[5] != 0.0, res[6] != 0.0, res[7] != 0.0};
          const double mid[4] = {res[0], res[1], res[2], res[3]};
          nr = poly_bracket_emit(lvl, bis, mid, roots);
        }
        if (have && qmode >= 0) {
          V3 P[3], f[3];
          for (int k = 0; k < 3; k++) {
            P[k] = V3{(double)s3d[3 * idx[k]], (double)s3d[3 * idx[k] + 1], (double)s3d[3 * idx[k] + 2]};
            V3 d{((double)s2d[2 * idx[k]] - cx) / fx, ((double)s2d[2 * idx[k] + 1] - cy) / fy, 1.0};
            f[k] = (1.0 / norm(d)) * d;
          }
          // the first solution with the smallest reprojection error on the remaining sample points wins
          int bk = -1;
          double be = 1.7976931348623157e308;
          M3 R;
          V3 t;
          V3 Pm[2];
          double zm[2][2];
#pragma unroll
          for (int m = 3; m < 5; m++) {
            const int im = m < modelPoints ? idx[m] : idx[3];
            Pm[m - 3] = V3{(double)s3d[3 * im], (double)s3d[3 * im + 1], (double)s3d[3 * im + 2]};
            zm[m - 3][0] = (double)s2d[2 * im];
            zm[m - 3][1] = (double)s2d[2 * im + 1];
          }
          int kk = 0;
          p3p_grunert_b(P, f, pm, roots, nr, [&](const M3& Rk, const V3& tk) {
            double e = 0;
#pragma unroll
            for (int m = 3; m < 5; m++) {
              if (m >= modelPoints) break;
              V3 X = Rk * Pm[m - 3] + tk;
              double z = X.z ? 1. / X.z : 1;
              double du = fx * X.x * z + cx - zm[m - 3][0], dv = fy * X.y * z + cy - zm[m - 3][1];
              e += du * du + dv * dv;
            }
            if (e < be) {
              be = e;
              bk = kk;
              R = Rk;
              t = tk;
            }
            kk++;
          });
          if (bk >= 0) {
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
              for (int c = 0; c < 3; c++) hpose[lane][3 * r + c] = R.m[r][c];
            hpose[lane][9] = t.x;
            hpose[lane][10] = t.y;
            hpose[lane][11] = t.z;
            cnt = -3;
          }
        }
        hcnt[lane] = cnt;
      }
      __syncthreads();
#endif
      }  // (P3P hypotheses)
      PNP_PROF(1);
      // score + replay in sub-batches of 16 hypotheses (the adaptive stop usually ends the search within the first few)
      const int SB = iterative ? NW : 16;
      for (int sb = 0; sb < B; sb += SB) {
        if (base + sb >= ctl[0]) break;  // (uniform: ctl[0] was written before the last barrier)
        for (int hy = sb + wv; hy < sb + SB; hy += RP_T / 64) {  // score the models: lanes stride the correspondences
          if (hcnt[hy] != -3) continue;
          M3 R;
#pragma unroll
          for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) R.m[r][c] = hpose[hy][3 * r + c];
          const V3 t{hpose[hy][9], hpose[hy][10], hpose[hy][11]};
          int good = 0;
          for (int i0 = 0; i0 < np; i0 += 64) {
            const int i = i0 + lane;
            bool in = false;
            if (i < np) {
              V3 Pi{(double)s3d[3 * i], (double)s3d[3 * i + 1], (double)s3d[3 * i + 2]};
              V3 X = R * Pi + t;
              double z = X.z ? 1. / X.z : 1;
              float du = (float)(fx * X.x * z + cx) - s2d[2 * i], dv = (float)(fy * X.y * z + cy) - s2d[2 * i + 1];
              in = du * du + dv * dv <= t2;
            }
            good += __popcll(__ballot(in));
          }
          if (lane == 0) hcnt[hy] = good;
        }
        __syncthreads();
        PNP_PROF(2);
        if (tid == 0) {
          int niters = ctl[0], maxGood = ctl[1];
          for (int k = sb; k < sb + SB; k++) {
            if (base + k >= niters) break;
            if (hcnt[k] == -2) {
              niters = 0;
              break;
            }
            int good = hcnt[k];
            int lim = maxGood > modelPoints - 1 ? maxGood : modelPoints - 1;
            if (good > lim) {
              maxGood = good;
              ctl[2] = base + k;
              for (int j = 0; j < 12; j++) bpose[j] = hpose[k][j];
              niters = ransac_update_num_iters(conf, (double)(np - good) / np, modelPoints, niters);
            }
          }
          ctl[0] = niters;
          ctl[1] = maxGood;
        }
        __syncthreads();
      }
    }
  }
  PNP_PROF(3);
  // the winning model's mask (whole workgroup)
  const bool found = ctl[2] >= 0;
  M3 R;
  V3 t{0, 0, 0};
  if (found) {
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = 0; c < 3; c++) R.m[r][c] = bpose[3 * r + c];
    t = V3{bpose[9], bpose[10], bpose[11]};
    for (int i = tid; i < np; i += RP_T) {
      V3 Pi{(double)s3d[3 * i], (double)s3d[3 * i + 1], (double)s3d[3 * i + 2]};
      V3 X = R * Pi + t;
      double z = X.z ? 1. / X.z : 1;
      float du = (float)(fx * X.x * z + cx) - s2d[2 * i], dv = (float)(fy * X.y * z + cy) - s2d[2 * i + 1];
      smask[i] = (du * du + dv * dv <= t2) ? 1 : 0;
    }
  } else {
    for (int i = tid; i < np; i += RP_T) smask[i] = 0;
  }
  __syncthreads();
  bool have_final = false;
  if (found && !iterative) {
    // SOLVEPNP_P3P: the final solvePnP on the inliers is EPnP.  The O(n) sums of its head by the whole workgroup, the rest by wave 0.
    if (wv == 0) {
      int k0 = 0;
      for (int b0 = 0; b0 < np; b0 += 64) {
        const int i = b0 + lane;
        const bool in = i < np && smask[i];
        const unsigned long long bal = __ballot(in);
        if (in) sh.inl[k0 + lane_prefix(bal)] = (short)i;
        k0 += __popcll(bal);
      }
      if (lane == 0) ctl[3] = k0;
    }
    __syncthreads();
    const int ni = ctl[3];
    const short* const inl = sh.inl;
    auto pw = [&](int i, double* q) {
      const int k = inl[i];
      q[0] = (double)s3d[3 * k], q[1] = (double)s3d[3 * k + 1], q[2] = (double)s3d[3 * k + 2];
    };
    auto uv = [&](int i, double* z) {
      const int k = inl[i];
      z[0] = (double)(float)(((double)s2d[2 * k] - cx) / fx) * fx + cx;
      z[1] = (double)(float)(((double)s2d[2 * k + 1] - cy) / fy) * fy + cy;
    };
    const epnp::Camera ecam{fx, fy, cx, cy};
    // the sums over the inliers (head: control points, MtM; sums: absolute orientation, reprojection errors) by the whole workgroup,
    // the eigen-decomposition .. betas in between by wave 0; the chunk sums live in the workspaces the hypotheses no longer need
    double* const part = reinterpret_cast<double*>(&sh.ew[1]);
    static_assert(sizeof(epnp::Work) * (RP_T / 64 - 1) >= sizeof(double) * epnp::PART_DOUBLES, "chunk-sum scratch");
    auto bsync = [] { __syncthreads(); };
    epnp::solve_head<RP_T>(sh.ew[0], ni, pw, uv, ecam, part, tid, bsync);
    if (wv == 0)
      epnp::solve_mid<64>(sh.ew[0], lane, [] {
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      });
    __syncthreads();
    epnp::solve_sums<RP_T>(sh.ew[0], ni, pw, uv, ecam, part, tid, bsync);
    if (wv == 0) {
      const epnp::Pose P = epnp::result(sh.ew[0]);
      if (P.ok) {
        M3 Rf;
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
          for (int c = 0; c < 3; c++) Rf.m[r][c] = P.R[3 * r + c];
        const SE3d Te = g2o_from_mat(Rf, V3{P.t[0], P.t[1], P.t[2]});
        T = se3_from_mat(q_to_mat(Te.q), Te.t);  // SE3_from_rvec_tvec (common.h:151-158)
        have_final = true;
      }
    }
  }
  if (wv != 0) return;  // final refinement: one wave
  if (!have_final) {
    T = iterative ? guess : se3_identity();
    if (iterative) T = se3_from_mat(q_to_mat(T.q), T.t);
  }
  inliers = 0;
  if (found && !iterative) {
    inliers = ctl[1];
    if (!have_final) {
      const SE3d Te = g2o_from_mat(R, t);
      T = se3_from_mat(q_to_mat(Te.q), Te.t);
    }
  } else if (found) {
    inliers = ctl[1];
    // Gauss-Newton refinement on the inliers (stand-in for OpenCV's final solvePnP).  The 21 + 6 normal-equation sums are
    // SEQUENTIAL sums over the inliers in index order (as the CPU restatement adds them): lane l computes the terms of
    // correspondence 64 c + l into an LDS row, lanes 0..26 then each add one column of the chunk in order.
    constexpr int GN_ROW = PNP_GN_ROW;
    SE3d Tb = g2o_from_mat(R, t);
    for (int it = 0; it < 10; it++) {
      double sum = 0;
      for (int c0 = 0; c0 < np; c0 += 64) {
        const int i = c0 + lane;
        double* row = gterms + lane * GN_ROW;
        if (i < np && smask[i]) {
          double e[2], J[2][6];
          proj_edge(Tb, V3{(double)s3d[3 * i], (double)s3d[3 * i + 1], (double)s3d[3 * i + 2]}, (double)s2d[2 * i],
                    (double)s2d[2 * i + 1], fx, fy, cx, cy, e, J);
          int q = 0;
#pragma unroll
          for (int r = 0; r < 6; r++) {
            row[21 + r] = J[0][r] * e[0] + J[1][r] * e[1];
#pragma unroll
            for (int c = r; c < 6; c++) row[q++] = J[0][r] * J[0][c] + J[1][r] * J[1][c];
          }
        } else {
#pragma unroll
          for (int k = 0; k < 27; k++) row[k] = 0.0;
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane < 27) {
          const double* col = gterms + lane;
          if (lane < 21) {
#pragma unroll 8
            for (int k = 0; k < 64; k++) sum += col[k * GN_ROW];
          } else {
#pragma unroll 8
            for (int k = 0; k < 64; k++) sum -= col[k * GN_ROW];  // b[r] -= J^T e
          }
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
      }
      if (lane < 27) gn[lane] = sum;
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      double H[36], b[6], dx[6];
      int q = 0;
#pragma unroll
      for (int r = 0; r < 6; r++) {
        b[r] = gn[21 + r];
#pragma unroll
        for (int c = r; c < 6; c++) {
          H[6 * r + c] = gn[q];
          H[6 * c + r] = gn[q];
          q++;
        }
      }
      __builtin_amdgcn_wave_barrier();
      asm volatile("" ::: "memory");
      if (!solve_spd6(H, b, dx)) break;
      // CvLevMarq's termination test (relative change of the six parameters below FLT_EPSILON), on the increment
      const double pn = 4.0 * (Tb.q.x * Tb.q.x + Tb.q.y * Tb.q.y + Tb.q.z * Tb.q.z) + (Tb.t.x * Tb.t.x + Tb.t.y * Tb.t.y + Tb.t.z * Tb.t.z);
      Tb = g2o_mul(g2o_exp(dx), Tb);
      double nn = 0;
#pragma unroll
      for (int k = 0; k < 6; k++) nn += dx[k] * dx[k];
      if (nn < 1e-20 || nn < 1.4210854715202004e-14 * pn) break;  // FLT_EPSILON^2
    }
    T = se3_from_mat(q_to_mat(Tb.q), Tb.t);
  }
#undef PNP_PROF
}

__device__ __forceinline__ void k_ransac_pnp_body(const Pipe& p) {
  chain_priority();
  const int s = blockIdx.x;
  StreamState& st = p.st[s];
  if (st.phase != PH_TRACK || !st.ok) return;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  long long tlast_ = 0;
#ifdef FLVIS_RANSAC_PROF
  constexpr bool kProf = true;
  tlast_ = (long long)wall_clock64();
  if (tid == 0 && p.counters) atomicAdd((unsigned long long*)&p.counters[32 + 7], 1ull);
#else
  constexpr bool kProf = false;
#endif
  const int cur = st.cur;
  Landmark* to = lm_ptr(p, cur, s);
  const int nl = st.n_lm[cur];
  __shared__ float s2d[NMAX * 2], s3d[NMAX * 3];
  __shared__ short sidx[NMAX];
  __shared__ unsigned char smask[NMAX];
  __shared__ int hcnt[64];
  __shared__ double hpose[64][12];
  __shared__ int ctl[4];
  __shared__ int ssub[64][8];
  __shared__ double bpose[12];
  __shared__ double gterms[64 * PNP_GN_ROW];
  __shared__ double gn[32];
  __shared__ epnp::Work ework[RP_T / 64];
  __shared__ short sinl[NMAX];
  __shared__ int snp;
  // gather (has3d && inlier) in order (wave 0)
  if (wv == 0) {
    int np0 = 0;
    for (int base = 0; base < nl; base += 64) {
      int i = base + lane;
      bool sel = i < nl && to[i].has3d && to[i].inlier;
      unsigned long long b = __ballot(sel);
      if (sel) {
        int k = np0 + lane_prefix(b);
        s2d[2 * k] = (float)to[i].p2u[0];
        s2d[2 * k + 1] = (float)to[i].p2u[1];
        s3d[3 * k] = (float)to[i].p3w[0];
        s3d[3 * k + 1] = (float)to[i].p3w[1];
        s3d[3 * k + 2] = (float)to[i].p3w[2];
        sidx[k] = (short)i;
      }
      np0 += __popcll(b);
    }
    if (lane == 0) snp = np0;
  }
  __syncthreads();
  const int np = snp;
  SE3d T;
  int inliers = 0;
  // solvePnPRansac(..., 100, 3.0, 0.99, ...) of LKORBTracking::tracking (lkorb_tracking.cpp:170-177)
  pnp_ransac_core<kProf>(PnpShared{s2d, s3d, smask, hcnt, hpose, ctl, ssub, bpose, gterms, gn, ework, sinl}, np, st.use_guess != 0, load_pose7(st.guess),
                         p.cam.fx, p.cam.fy, p.cam.cx, p.cam.cy, 0ull /* no seed: cv::RNG((uint64)-1) per call */,
                         100, 9.0f, 0.99, p.counters ? p.counters + 32 : nullptr, tlast_, T, inliers);
  if (wv != 0) return;
  __syncthreads();
  for (int i = lane; i < np; i += 64)
    if (smask[i] == 0) to[sidx[i]].inlier = 0;  // CameraFrame::updateLMState
  if (lane == 0) {
    store_pose7(st.T_c_w[cur], T);
    store_pose7(st.dbg_T_pnp, T);
    st.pnp_cnt = inliers;
    if (inliers < 10) st.ok = 0;
  }
  RPROF(32, 4);
}
__global__ __launch_bounds__(RP_T) void k_ransac_pnp(Pipe p) {
  kj_wait(p.kj);
  k_ransac_pnp_body(p);
  kj_signal(p.kj);
}

// FLVIS_PNP_TAIL=cv (opt-in): the final solve of cv::solvePnPRansac(..., SOLVEPNP_ITERATIVE) (lkorb_tracking.cpp:172) on its inliers as
// OpenCV runs it -- cv::solvePnP(ITERATIVE, useExtrinsicGuess = false) = cvFindExtrinsicCameraParams2: a DLT start and CvLevMarq -- in
// place of k_ransac_pnp's Gauss-Newton refinement of the winning model.  The function is the checker's (cv_solvers.hpp:
// find_extrinsic_iterative, `make -C oracle TAIL=cv`), compiled for the device and run by one WAVE per stream: every sum in OpenCV's order
// (by one lane each), bit-identical to the checker.  The RANSAC's inliers are the landmarks k_ransac_pnp left with has3d &&
// inlier, in landmark order (= the order of the correspondences).  A planar point set or fewer than six inliers (OpenCV starts from a
// homography there) keeps the Gauss-Newton pose, as the checker does.
struct PnpTailLanes {  // the 64 lanes of the stream's wave; sync: what one lane stored to global memory is there for the others
  int l;
  __device__ int lane() const { return l; }
  __device__ int lanes() const { return 64; }
  __device__ void sync() const {
    __threadfence();
    __builtin_amdgcn_wave_barrier();
  }
};
__global__ __launch_bounds__(64) void k_pnp_tail_cv(Pipe p) {
  const int s = blockIdx.x, lane = threadIdx.x;
  StreamState& st = p.st[s];
  if (st.phase != PH_TRACK || !st.ok || st.use_guess == 0 || st.pnp_cnt <= 0) return;
  const int cur = st.cur;
  const Landmark* to = lm_ptr(p, cur, s);
  const int nl = st.n_lm[cur];
  double* const M = p.pnp_tail_ws + (size_t)s * p.pnp_tail_stride;
  double* const m = M + 3 * NMAX;
  double* const work = m + 2 * NMAX;
  int n = 0;
  for (int base = 0; base < nl; base += 64) {
    const int i = base + lane;
    const bool sel = i < nl && to[i].has3d && to[i].inlier;
    const unsigned long long b = __ballot(sel);
    if (sel) {
      const int k = n + lane_prefix(b);
      // (the correspondences are cv::Point3f / cv::Point2f: camera_frame.cpp:415-427)
      M[3 * k] = (double)(float)to[i].p3w[0], M[3 * k + 1] = (double)(float)to[i].p3w[1], M[3 * k + 2] = (double)(float)to[i].p3w[2];
      m[2 * k] = (double)(float)to[i].p2u[0], m[2 * k + 1] = (double)(float)to[i].p2u[1];
    }
    n += __popcll(b);
  }
  const PnpTailLanes ln{lane};
  ln.sync();
  double rv[3], tv[3];
  // (every lane runs the function: the dense algebra redundantly, the loops over the correspondences dealt out -- cv_solvers.hpp)
  if (!cvs::find_extrinsic_iterative(n, M, m, p.cam.fx, p.cam.fy, p.cam.cx, p.cam.cy, work, rv, tv, nullptr, ln)) return;
  if (lane != 0) return;
  double Rm[9];
  cvs::rodrigues(rv, Rm, nullptr);
  M3 R;
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) R.m[r][c] = Rm[3 * r + c];
  const SE3d T = se3_from_mat(R, V3{tv[0], tv[1], tv[2]});
  store_pose7(st.T_c_w[cur], T);
  store_pose7(st.dbg_T_pnp, T);
}

// The same solver on caller-supplied correspondences: the geometric check of the loop closing, isLoopClosureKF
// (vo_loopclosing.cpp:660-686: solvePnPRansac(p3d, p2d, K, Mat(), r, t, false, 100, 2.0, 0.99, inliers, SOLVEPNP_P3P)); one
// workgroup per correspondence set.
constexpr int PNP_MAXN = 1024;
__global__ __launch_bounds__(RP_T) void k_pnp_ransac_sets(const float* __restrict__ p3d, const float* __restrict__ p2d,
                                                          const int* __restrict__ count, int cap, double fx, double fy, double cx, double cy,
                                                          int iterative, const double* __restrict__ guess7,
                                                          const unsigned long long* __restrict__ seeds, int max_iters, float t2, double conf,
                                                          double* __restrict__ pose7, unsigned char* __restrict__ mask,
                                                          int* __restrict__ n_inliers) {
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  __shared__ float s2d[PNP_MAXN * 2], s3d[PNP_MAXN * 3];
  __shared__ unsigned char smask[PNP_MAXN];
  __shared__ int hcnt[64];
  __shared__ double hpose[64][12];
  __shared__ int ctl[4];
  __shared__ int ssub[64][8];
  __shared__ double bpose[12];
  __shared__ double gterms[64 * PNP_GN_ROW];
  __shared__ double gn[32];
  __shared__ epnp::Work ework[RP_T / 64];
  __shared__ short sinl[PNP_MAXN];
  int np = count[b];
  np = np < 0 ? 0 : (np > cap ? cap : np);
  for (int i = tid; i < np; i += RP_T) {
    s2d[2 * i] = p2d[((size_t)b * cap + i) * 2];
    s2d[2 * i + 1] = p2d[((size_t)b * cap + i) * 2 + 1];
    s3d[3 * i] = p3d[((size_t)b * cap + i) * 3];
    s3d[3 * i + 1] = p3d[((size_t)b * cap + i) * 3 + 1];
    s3d[3 * i + 2] = p3d[((size_t)b * cap + i) * 3 + 2];
  }
  __syncthreads();
  SE3d T;
  int inliers = 0;
  long long tl = 0;
  pnp_ransac_core<false>(PnpShared{s2d, s3d, smask, hcnt, hpose, ctl, ssub, bpose, gterms, gn, ework, sinl}, np, iterative != 0,
                         iterative ? load_pose7(guess7 + 7 * b) : se3_identity(), fx, fy, cx, cy, seeds[b], max_iters, t2, conf, nullptr, tl, T,
                         inliers);
  if (wv != 0) return;
  __syncthreads();
  for (int i = lane; i < cap; i += 64) mask[(size_t)b * cap + i] = i < np ? smask[i] : 0;
  if (lane == 0) {
    store_pose7(pose7 + 7 * b, T);  // (identity / the guess when no model was found, n_inliers 0)
    n_inliers[b] = inliers;
  }
}

// EPnP alone (cv::solvePnP(..., SOLVEPNP_EPNP)) on caller-supplied correspondence sets, one wave per set: the test bench of
// epnp_core.hpp on the device.  out [set][EPNP_DBG_N]: R (9, row-major), t (3), ok, betas (3 x 4), err (3), v (4 x 12), L (6 x 10), rho (6),
// the eigenvalues of MtM as the Jacobi left them (12).
__global__ __launch_bounds__(64) void k_epnp_sets(const float* __restrict__ p3d, const float* __restrict__ p2d, const int* __restrict__ count,
                                                  int cap, double fx, double fy, double cx, double cy, double* __restrict__ out) {
  const int b = blockIdx.x, lane = threadIdx.x;
  __shared__ epnp::Work w;
  __shared__ double part[epnp::PART_DOUBLES];
  int n = count[b];
  n = n < 0 ? 0 : (n > cap ? cap : n);  // (cap <= 1024 is the caller's check: the chunk-sum scratch is sized for it)
  if (n < 4) {                          // fewer than four correspondences: no pose (cv::solvePnP asserts npoints >= 4)
    if (lane < EPNP_DBG_N) out[(size_t)b * EPNP_DBG_N + lane] = 0.0;
    if (lane + 64 < EPNP_DBG_N) out[(size_t)b * EPNP_DBG_N + lane + 64] = 0.0;
    if (lane + 128 < EPNP_DBG_N) out[(size_t)b * EPNP_DBG_N + lane + 128] = 0.0;
    return;
  }
  const float* const P = p3d + (size_t)b * cap * 3;
  const float* const Z = p2d + (size_t)b * cap * 2;
  auto pw = [&](int i, double* q) { q[0] = (double)P[3 * i], q[1] = (double)P[3 * i + 1], q[2] = (double)P[3 * i + 2]; };
  auto uv = [&](int i, double* z) {
    z[0] = (double)(float)(((double)Z[2 * i] - cx) / fx) * fx + cx;
    z[1] = (double)(float)(((double)Z[2 * i + 1] - cy) / fy) * fy + cy;
  };
  const epnp::Pose R = epnp::solve<64>(w, n, pw, uv, epnp::Camera{fx, fy, cx, cy}, part, lane, [] {
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  });
  if (lane != 0) return;
  double* o = out + (size_t)b * EPNP_DBG_N;
  for (int i = 0; i < 9; i++) o[i] = R.R[i];
  for (int i = 0; i < 3; i++) o[9 + i] = R.t[i];
  o[12] = R.ok ? 1.0 : 0.0;
  for (int i = 0; i < 12; i++) o[13 + i] = w.betas[i / 4][i % 4];
  for (int i = 0; i < 3; i++) o[25 + i] = w.err[i];
  for (int i = 0; i < 48; i++) o[28 + i] = w.v[i / 12][i % 12];
  for (int i = 0; i < 60; i++) o[76 + i] = w.L[i];
  for (int i = 0; i < 6; i++) o[136 + i] = w.rho[i];
  for (int i = 0; i < 12; i++) o[142 + i] = w.AV[13 * i];
}
void launch_epnp_sets(hipStream_t st, const float* p3d, const float* p2d, const int* count, int cap, int n_sets, const double* K4, double* out) {
  hipLaunchKernelGGL(k_epnp_sets, dim3(n_sets), dim3(64), 0, st, p3d, p2d, count, cap, K4[0], K4[1], K4[2], K4[3], out);
}

// ------------------------------------------------------------------------------------------------ after tracking
// LKORBTracking::tracking's tail (lkorb_tracking.cpp:179-199): fewer than 10 PnP inliers fail the frame; otherwise the roll / pitch of
// the PnP pose are pulled towards the IMU attitude.  One thread per stream; false: the frame has failed.
__device__ inline bool track_post_dev(const Pipe& p, int s) {
  StreamState& st = p.st[s];
  if (!st.ok) {
    track_fail(st);
    return false;
  }
  st.cont_fail = 0;
  if (st.has_imu) {
    ViRing ring{p.vi + (size_t)s * VI_QUEUE, &st};
    SE3d T = load_pose7(st.T_c_w[st.cur]);
    vi_vision_rp_compensation(p.cam, ring, st.frame_time[st.cur], T);
    store_pose7(st.T_c_w[st.cur], T);
  }
  return true;
}
__global__ void k_track_post(Pipe p) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= p.S) return;
  if (p.st[s].phase != PH_TRACK) return;
  track_post_dev(p, s);
}

// ------------------------------------------------------------------------------------------------ pose-only LM
// OptimizeInFrame::optimize: g2o Levenberg on one free pose, Huber(1), optimize(2), drop chi2 > 3, optimize(2).
//
// One workgroup per stream; every sum is a SEQUENTIAL sum in active-edge order (g2o walks its active edges sorted by edge id
// = landmark id, sparse_optimizer.cpp:493-498, and adds each edge's J^T W J into the Hessian block one after the other): the
// result is bit for bit what the CPU restatement computes, which is what keeps the closed-loop front-end in lockstep.
//   * the edges (has_3d && inlier landmarks) are gathered and sorted by id once (rank by counting, ids are unique);
//   * per pass, thread t computes the terms of edge t of the chunk (256 edges; the 21 + 6 normal-equation entries and the
//     robust chi2) into an LDS row; threads 0..27 then each own ONE of the 28 sums and add their column in edge order:
//     240 dependent fp64 adds per sum, ~1.5 us, while the other threads wait at the barrier;
//   * every lane then solves the same 6x6 system, so the pose needs no broadcast.
constexpr int PL_T = 256;    // 4 waves compute the per-edge terms; wave 0 owns the 28 sequential sums
constexpr int PL_ROW = 29;   // 28 sums per edge, padded: thread t writes row t (stride 29 doubles: 2-way bank conflicts at most)
constexpr int PL_NS = 28;    // 21 (upper H) + 6 (b) + 1 (robust chi2)
constexpr int PL_CH = 32;    // edges per chunk of a sum: 32 consecutive active edges are summed in order, then the chunk sums in order
constexpr int PL_EMAX = 512; // edges of one pose LM (16 regions x 30 landmarks is the largest configured frame: 480)
struct PoseLMShared {
  double pw[3][PL_EMAX];     // edges in id order
  double zu[PL_EMAX], zv[PL_EMAX];
  unsigned char alive[PL_EMAX];
  short src[PL_EMAX];        // landmark index of the k-th gathered edge (frame order)
  double terms[PL_T * PL_ROW]; // one chunk of per-edge terms; holds the ids (long long[PL_EMAX]) while the edges are ranked
  double tot[PL_NS];
  double part[PL_T / PL_CH][PL_NS];  // chunk sums of the current round
  int n;
};
static_assert(sizeof(double) * PL_T * PL_ROW >= sizeof(long long) * PL_EMAX, "the id scratch must fit the terms buffer");

// one pass over the edges at pose T: tot[0..20] upper triangle of H (row major), tot[21..26] b, tot[27] robust chi2
// (want_H == false: only tot[27]).  Sequential sums in edge order (see above).  All threads of the workgroup call it.
__device__ inline void pose_pass(const SE3d& T, PoseLMShared& sh, int n, bool want_H, double fx, double fy, double cx, double cy) {
  const int tid = threadIdx.x;
  double sum = 0;
  for (int c0 = 0; c0 < n; c0 += PL_T) {
    const int e = c0 + tid;
    double* row = sh.terms + tid * PL_ROW;
    const bool on = e < n && sh.alive[e];
    if (on) {
      double er[2], J[2][6];
      proj_edge(T, V3{sh.pw[0][e], sh.pw[1][e], sh.pw[2][e]}, sh.zu[e], sh.zv[e], fx, fy, cx, cy, er, want_H ? J : nullptr);
      const double c2 = er[0] * er[0] + er[1] * er[1];
      row[27] = huber_rho(c2);
      if (want_H) {
        const double w = huber_w(c2);
        const double o0 = -er[0] * w, o1 = -er[1] * w;
        int q = 0;
#pragma unroll
        for (int r = 0; r < 6; r++) {
          row[21 + r] = J[0][r] * o0 + J[1][r] * o1;
#pragma unroll
          for (int cc = r; cc < 6; cc++) row[q++] = (J[0][r] * w) * J[0][cc] + (J[1][r] * w) * J[1][cc];
        }
      }
    } else if (e < n) {
#pragma unroll
      for (int k = 0; k < PL_NS; k++) row[k] = 0.0;  // adding +0.0 is exact: the same as skipping the edge
    }
    __syncthreads();
    // every sum = the sum, in edge order, of the sums of chunks of PL_CH consecutive active edges, each chunk summed in edge order (the
    // oracle's definition: a dependent chain of 32 + 8 additions per round instead of 256; one thread per (chunk, term))
    const int m = (n - c0 < PL_T) ? n - c0 : PL_T;
    const int nch = (m + PL_CH - 1) / PL_CH;
    {
      const int c = tid / PL_NS, col = tid - c * PL_NS;
      if (c < nch && (want_H || col == 27)) {
        const double* colp = sh.terms + col;
        const int k0 = c * PL_CH, k1 = k0 + PL_CH < m ? k0 + PL_CH : m;
        double ps = 0;
        int k = k0;
        for (; k + 8 <= k1; k += 8) {
          double v[8];
#pragma unroll
          for (int u = 0; u < 8; u++) v[u] = colp[(k + u) * PL_ROW];
#pragma unroll
          for (int u = 0; u < 8; u++) ps += v[u];
        }
        for (; k < k1; k++) ps += colp[k * PL_ROW];
        sh.part[c][col] = ps;
      }
    }
    __syncthreads();
    if (tid < PL_NS && (want_H || tid == 27))
      for (int c = 0; c < nch; c++) sum += sh.part[c][tid];
  }
  if (tid < PL_NS) sh.tot[tid] = sum;
  __syncthreads();
}

__device__ inline void pose_lm_optimize(SE3d& T, PoseLMShared& sh, int n, int iterations, double fx, double fy, double cx, double cy) {
  double lambda = -1, ni = 2;
  for (int iteration = 0; iteration < iterations; iteration++) {
    pose_pass(T, sh, n, true, fx, fy, cx, cy);
    double currentChi = sh.tot[27];
    double H[36], b[6];
    int q = 0;
#pragma unroll
    for (int r = 0; r < 6; r++) {
      b[r] = sh.tot[21 + r];
#pragma unroll
      for (int c = r; c < 6; c++) {
        H[6 * r + c] = sh.tot[q];
        H[6 * c + r] = sh.tot[q];
        q++;
      }
    }
    if (iteration == 0) {
      double maxDiag = 0;
#pragma unroll
      for (int j = 0; j < 6; j++) maxDiag = fmax(fabs(H[7 * j]), maxDiag);
      lambda = 1e-5 * maxDiag;
      ni = 2;
    }
    double rho = 0;
    int qmax = 0;
    bool lambda_bad = false;
    do {
      SE3d backup = T;
      double Hl[36], x[6];
#pragma unroll
      for (int k = 0; k < 36; k++) Hl[k] = H[k];
#pragma unroll
      for (int j = 0; j < 6; j++) {
        Hl[7 * j] += lambda;
        x[j] = 0;
      }
      bool ok2 = solve_spd6(Hl, b, x);
      if (ok2) T = g2o_mul(g2o_exp(x), T);
      __syncthreads();  // tot[] was read by every thread above
      pose_pass(T, sh, n, false, fx, fy, cx, cy);
      double tempChi = sh.tot[27];
      if (!ok2) tempChi = 1.7976931348623157e308;
      rho = currentChi - tempChi;
      double scale = 0;
#pragma unroll
      for (int j = 0; j < 6; j++) scale += x[j] * (lambda * x[j] + b[j]);
      scale += 1e-3;
      rho /= scale;
      if (rho > 0 && isfinite(tempChi)) {
        double alpha = 1. - detm::det_powi((2 * rho - 1), 3);
        alpha = fmin(alpha, 2. / 3.);
        double scaleFactor = fmax(1. / 3., alpha);
        lambda *= scaleFactor;
        ni = 2;
        currentChi = tempChi;
      } else {
        lambda *= ni;
        ni *= 2;
        T = backup;
        if (!isfinite(lambda)) {
          lambda_bad = true;
          break;
        }
      }
      qmax++;
    } while (rho < 0 && qmax < 10);
    __syncthreads();
    if (qmax == 10 || rho == 0 || lambda_bad) break;
  }
}

__global__ __launch_bounds__(PL_T) void k_pose_lm(Pipe p) {
  chain_priority();
  const int s = blockIdx.x;
  StreamState& st = p.st[s];
  if (st.phase != PH_TRACK) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int cur = st.cur;
  Landmark* lms = lm_ptr(p, cur, s);
  const int nl = st.n_lm[cur];
  PoseLMShared& sh = *reinterpret_cast<PoseLMShared*>(pl_smem);
  long long* ids = reinterpret_cast<long long*>(sh.terms);
  // the tail of the tracking step (k_track_post's work) by one lane of wave 1, while wave 0 gathers the edges
#ifdef FLVIS_RANSAC_PROF
  long long tlast_ = (long long)wall_clock64();
#endif
  __shared__ int s_go;
  // (track_post_dev may end the frame -- track_fail sets phase = PH_IDLE and flips cur: every wave has read phase / cur / n_lm above
  // before one lane is allowed to change them)
  __syncthreads();
  if (tid == 64) s_go = track_post_dev(p, s) ? 1 : 0;
  // gather the edges (has3d && inlier) in frame order (wave 0)
  if (tid < 64) {
    int n = 0;
    for (int base = 0; base < nl; base += 64) {
      const int i = base + lane;
      const bool sel = i < nl && lms[i].has3d && lms[i].inlier;
      const unsigned long long bm = __ballot(sel);
      if (sel) {
        const int k = n + lane_prefix(bm);
        if (k < PL_EMAX) {
          ids[k] = lms[i].id;
          sh.src[k] = (short)i;
        }
      }
      n += __popcll(bm);
    }
    if (lane == 0) sh.n = n < PL_EMAX ? n : PL_EMAX;
  }
  __syncthreads();
  RPROF(48, 0);  // track_post (one lane) beside the edge gathering (wave 0)
  if (!s_go) return;  // the frame failed before the pose optimisation (track_fail has run)
  const int n = sh.n;
  bool ok = n >= 10;
  if (ok) {
    // active-edge order = ascending edge id (ids are unique): rank by counting, then the edge data goes to its rank
    for (int k = tid; k < n; k += PL_T) {
      const long long id = ids[k];
      int rank = 0;
      for (int j = 0; j < n; j++) rank += ids[j] < id;
      const Landmark& lm = lms[sh.src[k]];
      sh.pw[0][rank] = lm.p3w[0];
      sh.pw[1][rank] = lm.p3w[1];
      sh.pw[2][rank] = lm.p3w[2];
      sh.zu[rank] = lm.p2u[0];
      sh.zv[rank] = lm.p2u[1];
      sh.alive[rank] = 1;
    }
    __syncthreads();
    const double fx = p.cam.fx, fy = p.cam.fy, cx = p.cam.cx, cy = p.cam.cy;
    SE3d T0 = load_pose7(st.T_c_w[cur]);
    if (tid == 0) store_pose7(st.dbg_T_pre, T0);
    SE3d T = g2o_from_mat(q_to_mat(T0.q), T0.t);
    RPROF(48, 1);  // ranks + staging
    pose_lm_optimize(T, sh, n, 2, fx, fy, cx, cy);
    RPROF(48, 2);  // first optimisation (2 iterations)
    __shared__ int s_alive[PL_T / 64];
    int alive = 0;
    for (int base = 0; base < n; base += PL_T) {
      const int e = base + tid;
      bool keep = false;
      if (e < n) {
        double er[2];
        proj_edge(T, V3{sh.pw[0][e], sh.pw[1][e], sh.pw[2][e]}, sh.zu[e], sh.zv[e], fx, fy, cx, cy, er, nullptr);
        keep = !(er[0] * er[0] + er[1] * er[1] > 3.0);
        if (!keep) sh.alive[e] = 0;
      }
      int tot;
      block_rank<PL_T / 64>(keep, s_alive, tot);
      alive += tot;
    }
    __syncthreads();
    RPROF(48, 3);  // chi2 cull
    if (alive < 10) {
      ok = false;
    } else {
      pose_lm_optimize(T, sh, n, 2, fx, fy, cx, cy);
      RPROF(48, 4);  // second optimisation
      if (tid == 0) {
        store_pose7(st.T_c_w[cur], se3_from_mat(q_to_mat(T.q), T.t));
        store_pose7(st.dbg_T_lm, load_pose7(st.T_c_w[cur]));
      }
    }
  }
  if (!ok && tid == 0) track_fail(st);
#ifdef FLVIS_RANSAC_PROF
  if (tid == 0 && p.counters) atomicAdd((unsigned long long*)&p.counters[48 + 7], 1ull);
#endif
}

// ------------------------------------------------------------------------------------------------ reprojection filter
// calReprjInlierOutlier(1.5) + eraseReprjOutlier + viCorrectionFromVision; prepares the redetect inputs.
// One workgroup of NMAX threads per stream, one landmark per thread.
__device__ __forceinline__ void k_reproj_filter_body(const Pipe& p) {
  chain_priority();
  const int s = blockIdx.x;
  StreamState& st = p.st[s];
  if (st.phase != PH_TRACK) return;
  const int i = threadIdx.x;
  const int cur = st.cur;
  Landmark* lms = lm_ptr(p, cur, s);
  const int n = st.n_lm[cur];
  __shared__ double valid[NMAX];
  __shared__ double sh_thr;
  __shared__ int s_cnt[NMAX / 64];
  const SE3d T = load_pose7(st.T_c_w[cur]);
  Landmark lm;
  double d = 0;
  bool v = false;
  if (i < n) {
    lm = lms[i];
    V3 pc = se3_act(T, V3{lm.p3w[0], lm.p3w[1], lm.p3w[2]});
    double u = p.cam.fx * pc.x / pc.z + p.cam.cx, vv = p.cam.fy * pc.y / pc.z + p.cam.cy;
    double ex = lm.p2u[0] - u, ey = lm.p2u[1] - vv;
    d = sqrt(ex * ex + ey * ey);
    v = d < 3.0;
  }
  int nv;
  const int vr = block_rank<NMAX / 64>(v, s_cnt, nv);
  if (v) valid[vr] = d;
  if (i == 0 && nv == 0) sh_thr = 3.0;
  __syncthreads();
  // mean over the valid distances in index order (sequential sum like the reference, thread 0)
  if (i == 0) {
    double sum = 0;
    for (int k = 0; k < nv; k++) sum += valid[k];
    st.reproj_err = sum / (double)nv;
  }
  // median: element of rank nv/2 in ascending order (ties: any equal value)
  if (i < nv) {
    const int target = nv / 2;
    const double vi = valid[i];
    int less = 0, eq = 0;
    for (int j = 0; j < nv; j++) {
      const double vj = valid[j];
      less += vj < vi;
      eq += vj == vi;
    }
    if (less <= target && target < less + eq) sh_thr = 1.5 * vi;
  }
  __syncthreads();
  double sh = sh_thr;
  if (sh >= 3.0) sh = 3.0;
  // flag + erase outliers (order preserving; every landmark was read above)
  const bool keep = i < n && !(d > sh);
  int kept;
  const int k = block_rank<NMAX / 64>(keep, s_cnt, kept);
  if (keep) {
    lm.inlier = 1;
    lms[k] = lm;
    // existing points for FeatureDEM::redetect
    double* ex = p.exist_xy + ((size_t)s * NMAX + k) * 2;
    ex[0] = lm.p2d[0];
    ex[1] = lm.p2d[1];
  }
  if (i == 0) {
    st.n_lm[cur] = kept;
    st.orig_size = kept;
    p.n_exist[s] = kept;
    p.det_mode[s] = 2;
    p.det_maxc[s] = p.cam.gftt_num;
    st.vi_corr_due = st.has_imu ? 1 : 0;  // viCorrectionFromVision follows in k_vi_correction (off the critical path)
  }
}
__global__ __launch_bounds__(NMAX) void k_reproj_filter(Pipe p) {
  kj_wait(p.kj);
  k_reproj_filter_body(p);
  kj_signal(p.kj);
}
// VIMOTION::viCorrectionFromVision of the Tracking branch (f2f_tracking.cpp:256-262, after eraseReprjOutlier): one thread per stream.
// Nothing on the frame's chain before k_frame_end reads the filter state, and the searches through the IMU ring are dependent global
// loads (25 us of one thread): the detection stream runs it beside FeatureDEM / the stereo LK.
__global__ void k_vi_correction(Pipe p) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= p.S) return;
  StreamState& st = p.st[s];
  if (!st.vi_corr_due) return;
  st.vi_corr_due = 0;
  const int cur = st.cur;
  ViRing ring{p.vi + (size_t)s * VI_QUEUE, &st};
  vi_correction_from_vision(p.cam, st, ring, st.frame_time[cur], load_pose7(st.T_c_w[cur]), st.frame_time[cur ^ 1],
                            load_pose7(st.T_c_w[cur ^ 1]));
}

// ------------------------------------------------------------------------------------------------ new landmarks
template <int T>
__device__ __forceinline__ void k_add_new_body(const Pipe& p) {
  chain_priority();
  const int s = blockIdx.x;
  StreamState& st = p.st[s];
  const int mode = p.det_mode[s];
  if (mode == 0) return;
  const int lane = threadIdx.x;
  const int cur = st.cur;
  Landmark* lms = lm_ptr(p, cur, s);
  int n0 = st.n_lm[cur];
  int nn = p.n_new[s];
  if (n0 + nn > NMAX) nn = NMAX - n0;
  const float* xy = p.new_xy + (size_t)s * NEW_MAX * 2;
  const bool as_inlier = (mode == 1) ? true : (st.orig_size < 60);
  for (int k = lane; k < nn; k += T) {
    float src[2] = {xy[2 * k], xy[2 * k + 1]};
    float und[2] = {src[0], src[1]};
    // init_frame undistorts in both stereo modes, redetect only in STEREO_UNRECT, DEPTH_D435 never (f2f_tracking.cpp:294-304,410-437)
    if ((mode == 1 && p.cam.cam_type != CAM_DEPTH) || p.cam.cam_type == CAM_STEREO_UNRECT)
      undistort_point(src, p.cam.K0, p.cam.D0, p.cam.R0, p.cam.P0, und);
    Landmark lm;
    lm.id = st.lm_id_counter + k;
    lm.p3w[0] = lm.p3w[1] = lm.p3w[2] = 0;
    lm.p3c[0] = lm.p3c[1] = lm.p3c[2] = 0;
    lm.p2d[0] = (double)src[0];
    lm.p2d[1] = (double)src[1];
    lm.p2u[0] = (double)und[0];
    lm.p2u[1] = (double)und[1];
    lm.first2d[0] = lm.p2u[0];
    lm.first2d[1] = lm.p2u[1];
    for (int j = 0; j < 7; j++) lm.first_pose[j] = st.T_c_w[cur][j];
    lm.has3d = 0;
    lm.inlier = as_inlier ? 1 : 0;
    lm.tslot = -1;
    for (int j = 0; j < 4; j++) lm.pad[j] = 0;
    lms[n0 + k] = lm;
  }
  __syncthreads();
  if (lane == 0) {
    st.lm_id_counter += p.n_new[s];  // ids are consumed even for points that do not fit (never happens below NMAX)
    st.n_lm[cur] = n0 + nn;
    st.n_new = nn;
  }
}
__global__ __launch_bounds__(64) void k_add_new(Pipe p) {
  kj_wait(p.kj);
  k_add_new_body<64>(p);
  kj_signal(p.kj);
}

// ------------------------------------------------------------------------------------------------ depth: inputs
// Two kernels, because only the first is on the critical path: the stereo matcher needs its seeds; the two-view triangulation of
// recover3DPts_c_FromTriangulation is consumed by k_depth_innovate and runs beside the stereo LK on the detection stream.
__device__ __forceinline__ void depth_seed_dev(const Pipe& p, int s, int i) {
  const StreamState& st = p.st[s];
  const int cur = st.cur;
  const int n = st.n_lm[cur];
  if (i == 0) {
    p.lk_count[s] = n;
    p.lk_tag[s] = st.frame_id[cur];  // the stereo matcher's templates come from this frame's left image
  }
  if (i >= n) return;
  if (p.cam.cam_type == CAM_DEPTH) return;  // the measurement comes from the depth image, no stereo matching
  Landmark& lm = lm_ptr(p, cur, s)[i];
  // stereo LK seeds (camera_frame.cpp:108-122)
  float* p0 = p.prev_pts + ((size_t)s * NMAX + i) * 2;
  float* p1 = p.next_pts + ((size_t)s * NMAX + i) * 2;
  p0[0] = (float)lm.p2d[0];
  p0[1] = (float)lm.p2d[1];
  if (!p.tpl_ahead) {
    lm.tslot = i < p.tc_cap ? (short)i : (short)-1;  // where the stereo LK stores point i's templates (read back by the next frame's temporal LK)
  } else {
    // templates made ahead (k_lk_templates_ahead) are in the slot the landmark has carried since k_track_collect; a landmark without them
    // (new in this frame: behind the n_old older ones) gets a slot behind the survivors' and the stereo launch fills it
    int code = lk_tc_lookup(p.tc, p.tc_cap, p.tc_stride, s, lm.tslot, p0[0], p0[1], st.frame_id[cur]);
    if (code < 0) {
      const int n_old = n - st.n_new, base = p.det_mode[s] == 2 ? st.n_surv : 0;
      const int slot = i >= n_old ? base + (i - n_old) : (int)lm.tslot;
      if (slot >= 0 && slot < p.tc_cap) {
        code = -(slot + 2);
        lm.tslot = (short)slot;
      } else {
        code = -1;
        lm.tslot = -1;
      }
    }
    p.lk_slot[(size_t)s * NMAX + i] = code;
  }
  if (lm.has3d) {
    const SE3d T = load_pose7(st.T_c_w[cur]);
    SE3d T1c = se3_mul(load_pose7(p.cam.T_c1_c0), T);
    float p3[3] = {(float)lm.p3w[0], (float)lm.p3w[1], (float)lm.p3w[2]};
    project_point(p3, q_to_mat(T1c.q), T1c.t, p.cam.K1, p.cam.D1, p1);
  } else {
    p1[0] = p0[0];
    p1[1] = p0[1];
  }
}
__device__ __forceinline__ void k_depth_seeds_body(const Pipe& p) {
  chain_priority();
  const int s = blockIdx.y;
  if (p.det_mode[s] == 0) return;
  depth_seed_dev(p, s, blockIdx.x * 256 + threadIdx.x);
}
__global__ __launch_bounds__(256) void k_depth_seeds(Pipe p) {
  kj_wait(p.kj);
  k_depth_seeds_body(p);
  kj_signal(p.kj);
  kj_post_wait(p.kj);
}
// k_add_new and k_depth_seeds in one launch (round 6, FLVIS_CHAIN_MERGE bit 0): one workgroup of 256 threads per stream, the seeds behind a
// barrier instead of behind a launch
__global__ __launch_bounds__(256) void k_add_new_seeds(Pipe p) {
  kj_wait(p.kj);
  const int s = blockIdx.x;
  if (p.det_mode[s] != 0) {  // (uniform per workgroup)
    k_add_new_body<256>(p);
    __syncthreads();  // (workgroup scope: the landmarks and the count as the threads above left them)
    for (int i = threadIdx.x; i < NMAX; i += 256) depth_seed_dev(p, s, i);
  }
  kj_signal(p.kj);
  kj_post_wait(p.kj);
}
__device__ __forceinline__ void k_depth_triangulate_body(const Pipe& p) {
  chain_priority();
  const int s = blockIdx.y;
  const StreamState& st = p.st[s];
  if (p.det_mode[s] == 0) return;
  const int cur = st.cur;
  const int n = st.n_lm[cur];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const Landmark& lm = lm_ptr(p, cur, s)[i];
  const SE3d T = load_pose7(st.T_c_w[cur]);
  // recover3DPts_c_FromTriangulation (camera_frame.cpp:236-270)
  SE3d T1 = load_pose7(lm.first_pose);
  V3 baseline = T1.t - T.t;
  double* tri = p.tri + ((size_t)s * NMAX + i) * 3;
  unsigned char tm = 0;
  tri[0] = tri[1] = tri[2] = 0;
  if (norm(baseline) >= 0.2) {
    V3 pw = triangulate_two_view(lm.first2d[0], lm.first2d[1], lm.p2u[0], lm.p2u[1], T1, T, p.cam.fx, p.cam.fy, p.cam.cx,
                                 p.cam.cy);
    V3 pc = se3_act(T, pw);
    if (pc.z >= 0.5 && pc.z <= (double)p.cam.range) {
      tri[0] = pc.x;
      tri[1] = pc.y;
      tri[2] = pc.z;
      tm = 1;
    }
  }
  p.tri_mask[(size_t)s * NMAX + i] = tm;
}
__global__ __launch_bounds__(256) void k_depth_triangulate(Pipe p) {
  kj_wait(p.kj);
  k_depth_triangulate_body(p);
  kj_signal(p.kj);
}

// ------------------------------------------------------------------------------------------------ depth innovation
// recover3DPts_c_FromStereo (after LK) + depthInnovation + eraseNoDepthPoint.  One workgroup of NMAX threads per stream,
// one landmark per thread (read once, updated in registers, written to its compacted position).  The rand()-drawn dummy
// depths (quirk A11) are consumed in landmark order: failures are ranked with a workgroup prefix, thread 0 advances the
// stream's glibc generator by the number of failures.
__device__ __forceinline__ void k_depth_innovate_body(const Pipe& p) {
  chain_priority();
  const int s = blockIdx.x;
  StreamState& st = p.st[s];
  if (p.det_mode[s] == 0) return;
  const int i = threadIdx.x;
  const int cur = st.cur;
  Landmark* lms = lm_ptr(p, cur, s);
  const int n = st.n_lm[cur];
  __shared__ float rnd[NMAX];
  __shared__ int s_cnt[NMAX / 64];
  const float* p1 = p.next_pts + (size_t)s * NMAX * 2;
  const uint8_t* status = p.lk_status + (size_t)s * NMAX;
  const double range = (double)p.cam.range;
  const bool valid = i < n;
  Landmark lm;
  bool ok = false;
  V3 meas{0, 0, 0};
  const bool depth_cam = p.cam.cam_type == CAM_DEPTH;
  float ptx = 0.f, pty = 0.f;
  if (valid) {
    lm = lms[i];
    if (depth_cam) {
      // recover3DPts_c_FromDepthImg (camera_frame.cpp:182-234): nearest depth pixel (round half away from zero), metres =
      // Z16 / cam_scale_factor narrowed to float, valid in [0.3, range]
      ptx = (float)round(lm.p2d[0]);
      pty = (float)round(lm.p2d[1]);
      // (landmarks live in the open box (0, W-1) x (0, H-1), lkorb_tracking.cpp:95-102; the clamp only guards the read)
      const int ix = min(max(__float2int_rn(ptx), 0), p.cam.w - 1), iy = min(max(__float2int_rn(pty), 0), p.cam.h - 1);
      const uint16_t d16 = reinterpret_cast<const uint16_t*>(p.in_img1)[(size_t)s * p.cam.w * p.cam.h + (size_t)iy * p.cam.w + ix];
      const float z = (float)((double)d16 / p.cam.depth_scale);
      if ((double)z >= 0.3 && z <= p.cam.range) {
        meas = V3{((double)ptx - p.cam.cx) * (double)z / p.cam.fx, ((double)pty - p.cam.cy) * (double)z / p.cam.fy, (double)z};
        ok = true;
      }
    } else if (status[i] == 1) {
      float src[2] = {p1[2 * i], p1[2 * i + 1]}, u1[2];
      undistort_point(src, p.cam.K1, p.cam.D1, p.cam.R1, p.cam.P1, u1);
      float u0x = (float)lm.p2u[0], u0y = (float)lm.p2u[1];
      V3 pc = triangulate_dlt((double)u0x, (double)u0y, (double)u1[0], (double)u1[1], p.cam.P0, p.cam.P1);
      if (!(pc.z < 0 || pc.z > range)) {
        meas = pc;
        ok = true;
      }
    }
  }
  int nfail;
  const int frank = block_rank<NMAX / 64>(valid && !ok, s_cnt, nfail);
  if (i == 0)
    for (int k = 0; k < nfail; k++)
      rnd[k] = (float)(0.3 + (double)((float)glibc_rand_next(st) / ((float)(2147483647 / (0.4)))));
  __syncthreads();
  if (valid) {
    const SE3d T = load_pose7(st.T_c_w[cur]);
    const SE3d Tinv = se3_inverse(T);
    const float iir = p.cam.iir_ratio;
    const bool tm = p.tri_mask[(size_t)s * NMAX + i] != 0;
    if (!ok) {
      double depth = (double)rnd[frank];
      if (depth_cam) {  // pixel2camera(lm_2d_plane, ..., d_rand): the double plane position (camera_frame.cpp:200-207)
        meas = V3{(lm.p2d[0] - p.cam.cx) * depth / p.cam.fx, (lm.p2d[1] - p.cam.cy) * depth / p.cam.fy, depth};
      } else {
        float u0x = (float)lm.p2u[0], u0y = (float)lm.p2u[1];
        meas = V3{((double)u0x - p.cam.cx) * depth / p.cam.fx, ((double)u0y - p.cam.cy) * depth / p.cam.fy, depth};
      }
    }
    if (!ok && !tm) {
      if (!depth_cam && !lm.has3d && p.cam.enable_dummy) {  // camera_frame.cpp:288-301: stereo types only
        V3 pw = se3_act(Tinv, meas);
        lm.p3c[0] = meas.x; lm.p3c[1] = meas.y; lm.p3c[2] = meas.z;
        lm.p3w[0] = pw.x; lm.p3w[1] = pw.y; lm.p3w[2] = pw.z;
        lm.has3d = 1;
      }
    } else {
      V3 m = meas;
      if (!ok) {
        const double* tri = p.tri + ((size_t)s * NMAX + i) * 3;
        m = V3{tri[0], tri[1], tri[2]};
      }
      if (lm.has3d) {
        V3 lc = se3_act(T, V3{lm.p3w[0], lm.p3w[1], lm.p3w[2]});
        V3 upd = lc * (double)iir + m * (double)(1 - iir);
        V3 pw = se3_act(Tinv, upd);
        lm.p3c[0] = upd.x; lm.p3c[1] = upd.y; lm.p3c[2] = upd.z;
        lm.p3w[0] = pw.x; lm.p3w[1] = pw.y; lm.p3w[2] = pw.z;
      } else {
        V3 pw = se3_act(Tinv, m);
        lm.p3c[0] = m.x; lm.p3c[1] = m.y; lm.p3c[2] = m.z;
        lm.p3w[0] = pw.x; lm.p3w[1] = pw.y; lm.p3w[2] = pw.z;
        lm.has3d = 1;
      }
    }
  }
  // eraseNoDepthPoint (order preserving): every landmark was read above, the barriers inside block_rank separate the
  // reads from the compacted writes
  int kept;
  const bool keep = valid && lm.has3d;
  const int k = block_rank<NMAX / 64>(keep, s_cnt, kept);
  if (keep) lms[k] = lm;
  if (i == 0) st.n_lm[cur] = kept;
}
__global__ __launch_bounds__(NMAX) void k_depth_innovate(Pipe p) {
  kj_wait(p.kj);
  k_depth_innovate_body(p);
  kj_signal(p.kj);
}

// ------------------------------------------------------------------------------------------------ frame end
// init_frame's success test, keyframe decision (f2f_tracking.cpp:329-354,442-452), outputs, KeyFrame payload.
constexpr int FE_T = 256;
__device__ __forceinline__ void k_frame_end_body(const Pipe& p) {
  chain_priority();
  const int s = blockIdx.x;
  StreamState& st = p.st[s];
  const int lane = threadIdx.x;  // (first wave does the scalar bookkeeping)
  __shared__ int s_newkf;
  __shared__ int s_cnt[FE_T / 64];
  const bool s_chain = st.phase == PH_TRACK;  // a keyframe of the Tracking branch continues the keyframe chain
  if (lane == 0) s_newkf = 0;
  __syncthreads();
  if (st.phase == PH_INIT) {
    const int cur = st.cur;
    Landmark* lms = lm_ptr(p, cur, s);
    int valid = 0;
    for (int base = 0; base < st.n_lm[cur]; base += FE_T) {
      const int i = base + lane;
      int tot;
      block_rank<FE_T / 64>(i < st.n_lm[cur] && lms[i].has3d && lms[i].inlier, s_cnt, tot);
      valid += tot;
    }
    if (lane == 0) {
      if (valid > 30) {
        pose_record_push(p, s, st, (int)st.frame_id[cur], st.T_c_w[cur], false);  // f2f_tracking.cpp:443-446
        for (int j = 0; j < 7; j++) st.T_kf[j] = st.T_c_w[cur][j];
        st.new_kf = 1;
        st.state = ST_TRACKING;
      } else if (st.state == ST_TRACKFAIL) {
        st.cur ^= 1;  // "Recover failure": last_frame.swap(curr_frame)
      }
    }
  } else if (st.phase == PH_TRACK && lane == 0) {
    const int cur = st.cur;
    pose_record_push(p, s, st, (int)st.frame_id[cur], st.T_c_w[cur], true);  // STEP7, f2f_tracking.cpp:329-337
    SE3d Tkf = load_pose7(st.T_kf), Tc = load_pose7(st.T_c_w[cur]);
    SE3d Td = se3_mul(Tkf, se3_inverse(Tc));
    V3 r = so3_log(Td.q);
    double t_norm = fabs(Td.t.x) + fabs(Td.t.y) + fabs(Td.t.z);
    double r_norm = fabs(r.x) + fabs(r.y) + fabs(r.z);
    bool kf = false;
    if (st.frameCount < 40 && (st.frameCount % 5) == 0) kf = true;
    if (t_norm >= 0.05 || r_norm >= 0.2) kf = true;
    if (kf) {
      st.new_kf = 1;
      for (int j = 0; j < 7; j++) st.T_kf[j] = st.T_c_w[cur][j];
    }
  }
  __syncthreads();
  const int cur = st.cur;  // curr_frame as the caller sees it after image_feed
  if (lane == 0) {
    p.img_slot_in[s] = cur ^ 1;  // where the next frame's left image goes (frame_begin flips `cur` first thing)
    s_newkf = st.new_kf;
    FrameOut& o = p.out[s];
    o.state = st.state;
    o.new_keyframe = st.new_kf;
    o.reset_cmd = st.reset_cmd;
    o.n_landmarks = st.n_lm[cur];
    o.frame_id = st.frame_id[cur];
    for (int j = 0; j < 7; j++) o.T_c_w[j] = st.T_c_w[cur][j];
    o.of_cnt = st.of_cnt;
    o.f_cnt = st.f_cnt;
    o.pnp_cnt = st.pnp_cnt;
    o.reproj_err = st.reproj_err;
    const int frame_slot = st.feeds++;
    if (p.traj && frame_slot >= 0 && frame_slot < p.traj_cap) {
      double* t = p.traj + ((size_t)s * p.traj_cap + frame_slot) * 9;
      t[0] = st.frame_time[cur];
      for (int j = 0; j < 7; j++) t[1 + j] = st.T_c_w[cur][j];
      t[8] = (double)(st.state | (st.new_kf << 4));
    }
  }
  __syncthreads();
  if (s_newkf) {
    // append the KeyFrame payload to the stream's queue (the local map consumes it on its own HIP streams).  No waiting on
    // another kernel here (HIP gives no forward-progress guarantee between kernels): the host keeps the queue from filling
    // by stream-ordered back-pressure (pipeline.cpp: the tracking stream waits for the local-map launches of KFQ-3 frames
    // ago).  Should the queue be full all the same, the keyframe is dropped and counted -- what the reference's /vo_kf
    // subscriber queue does when the local map is slower than the tracker (src/backend/vo_localmap.cpp:452-456).
    __shared__ unsigned s_tail;
    __shared__ int s_full;
    __shared__ unsigned s_hash;
    if (lane == 0) {
      s_hash = 0u;
      const unsigned tl = p.kfq_tail[s];
      s_full = (tl - __hip_atomic_load(&p.kfq_head[s], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)KFQ) ? 1 : 0;
      if (s_full) atomicAdd((unsigned long long*)&p.counters[3], 1ull);
      s_tail = tl;
    }
    __syncthreads();
    if (s_full) return;
    KeyFrameDev& kf = p.kfq[(size_t)s * KFQ + (s_tail % KFQ)];
    Landmark* lms = lm_ptr(p, cur, s);
    const int n = st.n_lm[cur];
    int cnt = 0;
    unsigned hsum = 0u;
    for (int base = 0; base < n; base += FE_T) {
      int i = base + lane;
      bool sel = i < n && lms[i].has3d && lms[i].inlier;
      int tot;
      const int rk = block_rank<FE_T / 64>(sel, s_cnt, tot);
      if (sel) {
        int k = cnt + rk;
        if (k < KF_MAXLM) {
          kf.lm_id[k] = lms[i].id;
          kf.lm_2d[k][0] = lms[i].p2u[0];
          kf.lm_2d[k][1] = lms[i].p2u[1];
          kf.lm_3d[k][0] = lms[i].p3w[0];
          kf.lm_3d[k][1] = lms[i].p3w[1];
          kf.lm_3d[k][2] = lms[i].p3w[2];
          if (p.kf_check) hsum += kf_entry_hash(k, lms[i].id, lms[i].p2u, lms[i].p3w);
        }
      }
      cnt += tot;
    }
    if (p.kf_check) {  // (test knob: the checksum of what every wave wrote, verified by the consumer -- ba_update_dev)
      if (hsum) atomicAdd(&s_hash, hsum);
      __syncthreads();
    }
    if (lane == 0) {
      kf.frame_id = st.frame_id[cur];
      kf.stamp = st.frame_time[cur];
      kf.lm_count = cnt < KF_MAXLM ? cnt : KF_MAXLM;
      for (int j = 0; j < 7; j++) kf.T_c_w[j] = st.T_c_w[cur][j];
      // the gyro preintegration since the previous keyframe travels with the payload and restarts; a keyframe made by
      // init_frame() starts a new chain (nothing links it to the keyframes before the (re-)initialisation)
      for (int j = 0; j < 4; j++) kf.imu_dq[j] = st.kf_dq[j];
      kf.imu_dt = st.kf_dt;
      kf.imu_valid = (s_chain && st.kf_dt > 0) ? 1 : 0;
      kf.imu_pad = p.kf_check ? (int)s_hash : 0;
      // ... and the position part: the displacement preintegrated over the same interval and the filter velocity recorded when the
      // previous keyframe was made; the velocity at THIS keyframe is kept for the next payload
      for (int j = 0; j < 3; j++) {
        kf.imu_dp[j] = st.kf_dp[j];
        kf.imu_va[j] = st.kf_va[j];
        st.kf_dp[j] = st.kf_dv[j] = 0.0;
      }
      {
        const bool have = st.has_imu && st.vi_count > 0;
        const MotionState& mb = p.vi[(size_t)s * VI_QUEUE + (st.vi_head + (st.vi_count > 0 ? st.vi_count - 1 : 0)) % VI_QUEUE];
        for (int j = 0; j < 3; j++) st.kf_va[j] = have ? mb.vel[j] : 0.0;
      }
      st.kf_dq[0] = 1.0, st.kf_dq[1] = st.kf_dq[2] = st.kf_dq[3] = 0.0;
      st.kf_dt = 0;
      kf.valid = 1;
    }
    // (one release, by the thread that publishes: the barrier orders the other threads' stores before it, and a `buffer_wbl2` per wave
    // -- what a fence executed by every thread costs -- writes the XCD's L2 back once per wave, k_ba_worker)
    __syncthreads();
    if (lane == 0) __hip_atomic_store(&p.kfq_tail[s], s_tail + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__global__ __launch_bounds__(FE_T) void k_frame_end(Pipe p) {
  kj_wait(p.kj);
  k_frame_end_body(p);
  kj_signal(p.kj);
}

// ------------------------------------------------------------------------------------------------ launchers
void launch_imu_feed(hipStream_t st, const Pipe& p) {
  hipLaunchKernelGGL(k_imu_feed, dim3((p.S + 63) / 64), dim3(64), 0, st, p);
}
void launch_frame_begin(hipStream_t st, const Pipe& p, const double* d_time) {
  hipLaunchKernelGGL(k_frame_begin, dim3((p.S + 63) / 64), dim3(64), 0, st, p, d_time);
}
// stream-ordered store of a counter into host-mapped memory (the host polls it where it must know that earlier work of the stream
// has finished: hipEventSynchronize returned only when everything enqueued so far, on any stream, was done -- DESIGN.md section 4)
__global__ void k_store_progress(long long* __restrict__ host_word, long long v) {
  __hip_atomic_store(host_word, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
void launch_store_progress(hipStream_t st, long long* host_word, long long v) {
  hipLaunchKernelGGL(k_store_progress, dim3(1), dim3(1), 0, st, host_word, v);
}
// Stream-ordered wait for an upload (round 6, flvis_image_feed_host): the copy engine writes a block of `n_words` sequence numbers behind
// the images of a call (in-order on its queue); ONE lane of one wave sleeps until the first and the last word of the block have reached
// `seq`.  The copies were issued before this kernel was enqueued and run on the copy engine whatever the compute queues do, so the wait
// ends by itself -- in the steady state it finds the block there and costs a launch.  Not a command-processor wait: an AQL barrier packet
// that polls a signal (what an event recorded behind a copy is turned into) slows the packet processing of the hardware queues next to
// it for as long as it is pending (profiles/r06_h2d.md).  A wait of more than ~4 s stores `seq` into err_word (host-mapped) and gives up.
__global__ void k_wait_flag(const long long* __restrict__ flag, int n_words, long long seq, long long* __restrict__ err_word) {
  const unsigned long long t0 = wall_clock64();
  while (__hip_atomic_load(&flag[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < seq ||
         __hip_atomic_load(&flag[n_words - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
    __builtin_amdgcn_s_sleep(32);
    if (wall_clock64() - t0 > 400000000ull) {  // 100 MHz constant clock
      if (err_word) __hip_atomic_store(err_word, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      break;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}
// stream-ordered store of a sequence number into a device word (the producing side of a flag join, pipeline.cpp)
__global__ void k_store_flag(long long* __restrict__ word, long long v) {
  __hip_atomic_store(word, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
void launch_store_flag(hipStream_t st, long long* word, long long v) { hipLaunchKernelGGL(k_store_flag, dim3(1), dim3(1), 0, st, word, v); }
void launch_wait_flag(hipStream_t st, const long long* flag, int n_words, long long seq, long long* err_word) {
  hipLaunchKernelGGL(k_wait_flag, dim3(1), dim3(1), 0, st, flag, n_words, seq, err_word);
}
void launch_frame_head(hipStream_t st, const Pipe& p, const double* d_time, long long* host_progress, long long frame_no) {
  hipLaunchKernelGGL(k_frame_head, dim3(p.S), dim3(64), 0, st, p, d_time, host_progress, frame_no);
}
void launch_frame_head_prepare(hipStream_t st, const Pipe& p, const double* d_time, long long* host_progress, long long frame_no) {
  hipLaunchKernelGGL(k_frame_head_prepare, dim3(p.S), dim3(256), 0, st, p, d_time, host_progress, frame_no);
}
void launch_apply_correction(hipStream_t st, const Pipe& p) {
  hipLaunchKernelGGL(k_apply_correction, dim3(p.S), dim3(AC_T), 0, st, p);
}
void launch_track_prepare(hipStream_t st, const Pipe& p) {
  hipLaunchKernelGGL(k_track_prepare, dim3(NMAX / 256, p.S), dim3(256), 0, st, p);
}
void launch_track_collect(hipStream_t st, const Pipe& p) { hipLaunchKernelGGL(k_track_collect, dim3(p.S), dim3(64), 0, st, p); }
void launch_ransac_f(hipStream_t st, const Pipe& p, bool with_collect) {
  if (with_collect) hipLaunchKernelGGL(k_collect_ransac_f, dim3(p.S), dim3(RF_T), 0, st, p);
  else hipLaunchKernelGGL(k_ransac_f, dim3(p.S), dim3(RF_T), 0, st, p);
}
void launch_ransac_pnp(hipStream_t st, const Pipe& p) { hipLaunchKernelGGL(k_ransac_pnp, dim3(p.S), dim3(RP_T), 0, st, p); }
void launch_pnp_tail_cv(hipStream_t st, const Pipe& p) { hipLaunchKernelGGL(k_pnp_tail_cv, dim3(p.S), dim3(64), 0, st, p); }
int pnp_ransac_max_points() { return PNP_MAXN; }
static hipError_t pnp_tables_init();
void launch_pnp_ransac_sets(hipStream_t st, const float* p3d, const float* p2d, const int* count, int cap, int n_sets, const double* K4,
                            int iterative, const double* guess7, const unsigned long long* seeds, int max_iters, double reproj_px,
                            double conf, double* pose7, unsigned char* mask, int* n_inliers) {
  (void)pnp_tables_init();  // (the loop closing may use the solver without a tracker)
  hipLaunchKernelGGL(k_pnp_ransac_sets, dim3(n_sets), dim3(RP_T), 0, st, p3d, p2d, count, cap, K4[0], K4[1], K4[2], K4[3], iterative,
                     guess7, seeds, max_iters, (float)(reproj_px * reproj_px), conf, pose7, mask, n_inliers);
}
void launch_track_post(hipStream_t st, const Pipe& p) {
  hipLaunchKernelGGL(k_track_post, dim3((p.S + 63) / 64), dim3(64), 0, st, p);
}
void launch_pose_lm(hipStream_t st, const Pipe& p) { hipLaunchKernelGGL(k_pose_lm, dim3(p.S), dim3(PL_T), sizeof(PoseLMShared), st, p); }
// the tabulated first batches of the PnP RANSAC (see g_pnp_sub5): once per device, before the first launch that reads them
static hipError_t pnp_tables_init() {
  static bool done[64] = {};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev >= 0 && dev < 64 && done[dev]) return hipSuccess;
  std::vector<unsigned short> s5((size_t)PNP_TAB_N * PNP_TAB_B5 * 5, 0), s4((size_t)PNP_TAB_N * PNP_TAB_B4 * 4, 0);
  std::vector<unsigned long long> r5(PNP_TAB_N, 0), r4(PNP_TAB_N, 0);
  auto fill = [](int count, int m, int nsub, unsigned short* out, unsigned long long& state_out) {
    uint64_t st = 0xffffffffffffffffull;  // cv::RNG((uint64)-1)
    for (int sidx = 0; sidx < nsub; sidx++) {
      int idx[5] = {-1, -1, -1, -1, -1};
      for (int i = 0; i < m;) {  // getSubset: rng.uniform(0, count) per slot, redrawn while it repeats an earlier slot
        st = (uint64_t)(uint32_t)st * 4164903690u + (uint32_t)(st >> 32);
        const int v = (int)((uint32_t)st % (uint32_t)count);
        bool dup = false;
        for (int j = 0; j < i; j++) dup = dup || idx[j] == v;
        if (dup) continue;
        idx[i++] = v;
      }
      for (int j = 0; j < m; j++) out[(size_t)sidx * m + j] = (unsigned short)idx[j];
    }
    state_out = st;
  };
  for (int n = 0; n < PNP_TAB_N; n++) {
    if (n > 5) fill(n, 5, PNP_TAB_B5, &s5[(size_t)n * PNP_TAB_B5 * 5], r5[n]);
    if (n > 4) fill(n, 4, PNP_TAB_B4, &s4[(size_t)n * PNP_TAB_B4 * 4], r4[n]);
  }
  {  // the F-matrix RANSAC's candidates (see g_f_sub7): 16 subsets of 7 per count, no checkSubset, the state behind each
    std::vector<unsigned short> s7((size_t)F_TAB_N * F_TAB_B * 7, 0);
    std::vector<unsigned long long> r7((size_t)F_TAB_N * F_TAB_B, 0);
    for (int n = 8; n < F_TAB_N; n++) {
      uint64_t st = 0xffffffffffffffffull;
      for (int k = 0; k < F_TAB_B; k++) {
        int idx[7] = {-1, -1, -1, -1, -1, -1, -1};
        for (int i = 0; i < 7;) {
          st = (uint64_t)(uint32_t)st * 4164903690u + (uint32_t)(st >> 32);
          const int v = (int)((uint32_t)st % (uint32_t)n);
          bool dup = false;
          for (int j = 0; j < i; j++) dup = dup || idx[j] == v;
          if (dup) continue;
          idx[i++] = v;
        }
        for (int j = 0; j < 7; j++) s7[((size_t)n * F_TAB_B + k) * 7 + j] = (unsigned short)idx[j];
        r7[(size_t)n * F_TAB_B + k] = st;
      }
    }
    e = hipMemcpyToSymbol(HIP_SYMBOL(g_f_sub7), s7.data(), s7.size() * sizeof(unsigned short));
    if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_f_rng7), r7.data(), r7.size() * sizeof(unsigned long long));
    if (e != hipSuccess) return e;
  }
  e = hipMemcpyToSymbol(HIP_SYMBOL(g_pnp_sub5), s5.data(), s5.size() * sizeof(unsigned short));
  if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_pnp_sub4), s4.data(), s4.size() * sizeof(unsigned short));
  if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_pnp_rng5), r5.data(), r5.size() * sizeof(unsigned long long));
  if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_pnp_rng4), r4.data(), r4.size() * sizeof(unsigned long long));
  if (e == hipSuccess && dev >= 0 && dev < 64) done[dev] = true;
  return e;
}
hipError_t track_kernels_init() {
  {
    const hipError_t e = pnp_tables_init();
    if (e != hipSuccess) return e;
  }
  return hipFuncSetAttribute((const void*)k_pose_lm, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(PoseLMShared));
}
void launch_reproj_filter(hipStream_t st, const Pipe& p) { hipLaunchKernelGGL(k_reproj_filter, dim3(p.S), dim3(NMAX), 0, st, p); }
void launch_vi_correction(hipStream_t st, const Pipe& p) { hipLaunchKernelGGL(k_vi_correction, dim3((p.S + 63) / 64), dim3(64), 0, st, p); }
void launch_add_new(hipStream_t st, const Pipe& p) { hipLaunchKernelGGL(k_add_new, dim3(p.S), dim3(64), 0, st, p); }
void launch_depth_seeds(hipStream_t st, const Pipe& p) {
  hipLaunchKernelGGL(k_depth_seeds, dim3(NMAX / 256, p.S), dim3(256), 0, st, p);
}
void launch_add_new_seeds(hipStream_t st, const Pipe& p) { hipLaunchKernelGGL(k_add_new_seeds, dim3(p.S), dim3(256), 0, st, p); }
void launch_depth_triangulate(hipStream_t st, const Pipe& p) {
  hipLaunchKernelGGL(k_depth_triangulate, dim3(NMAX / 256, p.S), dim3(256), 0, st, p);
}
void launch_depth_innovate(hipStream_t st, const Pipe& p) { hipLaunchKernelGGL(k_depth_innovate, dim3(p.S), dim3(NMAX), 0, st, p); }
void launch_frame_end(hipStream_t st, const Pipe& p) {
  hipLaunchKernelGGL(k_frame_end, dim3(p.S), dim3(FE_T), 0, st, p);
}

}  // namespace flvis
