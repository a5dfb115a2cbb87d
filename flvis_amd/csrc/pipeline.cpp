// flvis_amd: host orchestration of the batched front-end + local map behind the C ABI (include/flvis_hip.h).
// Host-side mirror of the reference's F2FTracking / TrackingNodeletClass::process / LocalMapNodeletClass call sequence
// (src/frontend/f2f_tracking.cpp:59-400, src/frontend/vo_tracking.cpp:326-371,396-430, src/backend/vo_localmap.cpp:87-380):
// the host only stages the per-frame inputs and enqueues a FIXED kernel sequence per frame; every decision the reference
// takes per frame is taken on the device by the kernels in track_kernels.hip / ba_solve.hip.
//
// Lanes.  The per-frame chain of a stream is a recurrence of ~15 dependent kernels, most of them one workgroup per stream
// (geometry on <= 240 landmarks): with all S streams of a tracker in ONE chain those kernels occupy S of the 256 CUs while
// everything waits for them, and only the image kernels (LK, pyramids, corner response) fill the chip.  A tracker
// therefore splits its streams into LANES (sub-batches): every lane owns its streams' state, its own HIP streams (tracking
// chain + corner detection) and runs the same fixed sequence on its slice of the batch, so the LK of one lane overlaps the
// geometry chain of the others.  Streams never interact, so the partition changes no result.  The
// local-map workers of all lanes share two low-priority HIP streams (the number of hardware queues is limited: streams
// beyond it share a queue and serialise behind each other's long kernels).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <limits>
#include <vector>

#include "../../include/flvis_hip.h"
#include "ctx.hpp"
#include "track_kernels.hpp"

namespace flvis {

constexpr int PROF_STAGES = 20;  // every stage has its own (begin, end) event pair on the stream it runs on
static const char* kStageNames[PROF_STAGES] = {
    "imu_feed+frame_begin", "ingest(equalize)", "pyr_down(left)", "track_prepare", "lk_track(temporal)", "track_collect",
    "ransac_f", "ransac_pnp", "track_post+pose_lm", "reproj_filter", "gftt:eig_cand", "gftt:(merged)", "gftt:pick",
    "feature_dem+add_new", "depth_prepare", "lk_track(stereo)", "depth_innovate", "frame_end",
    "ba_worker(launch)", "frame(chain)"};

// One sub-batch of streams [s0, s0 + S): all device state of those streams and the HIP streams their frames run on.
struct Lane {
  int s0 = 0, S = 0;
  Pipe pipe;
  hipStream_t st = nullptr;  // tracking chain (the context's stream when the tracker has a single lane)
  bool own_st = false;
  // images
  uint8_t* pyr0[2][LK_MAX_LEVELS] = {};
  uint8_t* pyr1[LK_MAX_LEVELS] = {};
  uint32_t* tc = nullptr;  // LK template cache [S][pipe.tc_cap][tc_stride dwords] (LKParams::tc), or nullptr
  int tc_stride = 0;
  GfttScratch gftt;
  float* gftt_xy = nullptr;
  float* dem_sorted = nullptr;  // [S][2 gftt_num][2] FeatureDEM: the corners in region-major, score-sorted order (k_feature_dem_prep)
  int* dem_roff = nullptr;      // [S][17] ... and the offset of every region in that list
  int* gftt_n = nullptr;
  unsigned* eq_hist = nullptr;
  uint8_t* eq_lut = nullptr;
  // per-frame inputs of the lane, uploaded as ONE block: [times S][imu S*IMU_MAX*7][input image bases 2][n_imu S]
  uint8_t* d_inputs = nullptr;
  double* d_time = nullptr;
  const uint8_t** d_tab = nullptr;
  size_t input_bytes = 0;
  std::vector<double> h_imu;  // [S][IMU_MAX][7] staged between two frames
  std::vector<int> h_nimu;
  std::vector<long long> imu_read;  // [S] rows of the stream's IMU-state ring the caller has fetched (flvis_get_imu_states)
  // Host staging: a ring of pinned slots, each guarded by an event recorded after its upload, so that image_feed never
  // waits for the previous frame -- the host runs several frames ahead of the GPU and a frame's launches are already queued
  // when the GPU gets to them.
  static constexpr int PIN_RING = 4;
  void* pinned[PIN_RING] = {};      // page-locked staging of a frame's inputs (host view) ...
  uint8_t* pinned_dev[PIN_RING] = {};  // ... and the same memory as the device sees it (FLVIS_INPUT_ZEROCOPY: k_frame_head reads it in place)
  const uint8_t* h_tab[2] = {nullptr, nullptr};  // the image bases of the last frame fed (host copy of the table)
  size_t in_off_imu = 0, in_off_tab = 0, in_off_n = 0;
  // the slot of frame n may be refilled once frame n's upload is done: k_frame_head (the first kernel after the upload) stores the
  // frame number into this host-mapped word and the host polls it.  (hipEventSynchronize on an event recorded after the upload
  // returned only when EVERYTHING enqueued so far had finished -- measured: the host then slept through four queued frames and the
  // GPU ran dry once per burst, a 0.9 ms hole in every fifth frame.)
  volatile long long* h_progress = nullptr;  // host view
  long long* d_progress = nullptr;           // device view of the same word
  long long frames_uploaded = 0;             // frames this lane has enqueued
  std::vector<void*> allocs;
  std::vector<hipEvent_t> prof_ev;  // optional per-stage HIP-event timing (flvis_prof_enable)
  std::vector<unsigned char> prof18_rec;  // per armed step: the local-map launch's event pair (stage 18) was recorded -- a deferred launch
                                          // (FLVIS_BA_START) belongs to the NEXT step, the first step of a batch has none, the last one two
  // corner detection on its own HIP stream: goodFeaturesToTrack only needs the new image, so it runs beside the temporal
  // tracking chain (LK -> RANSACs -> pose LM) and joins before FeatureDEM consumes the corners
  hipStream_t det_stream = nullptr;
  hipEvent_t ev_img = nullptr, ev_det = nullptr, ev_gftt = nullptr, ev_fe = nullptr, ev_lm = nullptr, ev_tri = nullptr, ev_head = nullptr;
  hipEvent_t ev_endf = nullptr;  // "k_frame_end has finished": the next frame's ingest and an immediate local-map launch wait for it (folded joins)
  // templates ahead of the stereo matcher (FLVIS_TPL_AHEAD, round 6): k_lk_templates_ahead on a low-priority stream of its own, beside the
  // one-workgroup-per-stream geometry kernels of the frame's chain; ev_tpl: "the frame's templates are in the cache" (k_depth_seeds waits)
  hipStream_t tpl_stream = nullptr;
  hipEvent_t ev_tpl = nullptr;
  bool endf_valid = false;       // the last k_frame_end carried that signal
  int idx = 0;  // position in Pipeline::lanes
  // multi-lane trackers: ev_end[n % HOLD_RING] follows the lane's n-th frame (the context's stream waits for the frame whose
  // input buffers the caller may reuse next, see flvis_set_input_hold); ev_stagger follows the temporal LK of the lane's first
  // processed frame (the next lane starts its first frame there, so that the lanes run out of phase)
  static constexpr int HOLD_RING = 8;
  hipEvent_t ev_end[HOLD_RING] = {};
  hipEvent_t ev_stagger = nullptr;
  // back-pressure on the keyframe queues in stream order: ev_ba_done[i % BAQ] follows the lane's i-th local-map launch; the
  // tracking stream waits for the launches of KFQ-3 frames ago before it appends new keyframes (see lane_frame)
  static constexpr int BAQ = 64;
  hipEvent_t ev_ba_done[BAQ] = {};
  // FLVIS_JOIN (round 6; "flag" is the default, "event" the form of rounds 1-5): the joins between the lane's streams as device words instead of events -- the producing stream stores a
  // sequence number behind its last kernel (k_store_flag), the consuming stream waits for it with one sleeping lane (k_wait_flag).  An
  // event record is a system-scope barrier packet and a wait a barrier packet that polls the record's signal for as long as it is pending
  // (and slows the queues beside it, profiles/r06_h2d.md); a word in HBM costs two one-lane launches.
  static constexpr int JOIN_IDS = 9 + BAQ;
  long long* d_join = nullptr;             // [JOIN_IDS] words, 64 bytes apart
  unsigned* d_join_cnt = nullptr;          // [JOIN_IDS] arrival counters of the launches that signal a word themselves (KJoin)
  long long join_seq[JOIN_IDS] = {};
  long long ba_launches = 0;
  bool ba_pending = false;       // FLVIS_BA_START > 0: the local-map launch for the last frame's keyframes has not been enqueued yet
  unsigned ba_tag = 0;           // tag of the lane's last local-map launch (k_ba_worker's stream list is valid for one tag; 0 is never used)
};

struct Pipeline {
  int S = 0;
  flvis_cfg cfg;
  int lane_size = 0;
  std::vector<Lane*> lanes;
  int levels_t = 0, levels_s = 0, levels = 0;
  int lw[LK_MAX_LEVELS], lh[LK_MAX_LEVELS], lpitch[LK_MAX_LEVELS];
  size_t lstride[LK_MAX_LEVELS];
  int lbx = 0, lby = 0;  // physical border of every pyramid level (columns / rows on each side)
  int max_pts = 0;  // bound on the landmarks of a frame (16 regions x max_region_feature_num): sizes the LK grid
  int tpl_start = 1;  // FLVIS_TPL_START: where k_lk_templates_ahead starts (lane_frame)
  int lk_order = 2;     // FLVIS_LK_ORDER (bits: 1 temporal, 2 stereo; default 2): the launch takes a stream's points from the last to the first
  int chain_merge = 3;  // FLVIS_CHAIN_MERGE: launches of the frame's chain folded into their neighbours (lane_frame)
  long long frames_fed = 0;
  std::vector<void*> allocs;  // context-level device allocations (host-feed staging)
  int prof_cap = 0, prof_step = 0;
  unsigned long long prof_mask = ~0ull;  // stages that record events (an event record costs a few us on the GPU queue)
  // The local map runs beside the front-end: k_frame_end appends KeyFrame payloads to per-stream queues, k_ba_worker (one
  // launch per lane and frame on one of the shared local-map streams, round-robin) drains them.  Its output is never fed
  // back into the tracker in the reference (src/frontend/vo_tracking.cpp:373-385).
  static constexpr int NBA = 8;
  static_assert(NBA == BA_PLAN_SLOTS, "one stream list per local-map HIP stream");
  int nba = 2;                   // local-map streams in use: 2 per lane (FLVIS_BA_STREAMS per lane, tuning knob)
  int nba_lane = 2;              // ... of which every lane uses its own nba_lane
  int host_lead = 3;             // frames the host may run ahead of the GPU (FLVIS_HOST_LEAD, 1 .. PIN_RING; 2 until round 6: a host thread that is held up for a millisecond then leaves the GPU dry)
  int host_lead_cap = 4;         // (flvis_image_feed_host lowers it to 1 for its call: its copies and events add to the queued commands)
  int input_hold = 0;            // flvis_set_input_hold: frames the caller keeps its input buffers untouched after handing them over
  bool stagger = true;           // FLVIS_LANE_STAGGER=0: lanes start their first frame together
  double host_ms_total = 0, host_ms_wait = 0;  // host time inside flvis_image_feed / of it blocked on the pinned ring
  bool sync_each_frame = false;  // FLVIS_SYNC_EACH_FRAME=1: image_feed waits for the previous frame (tuning knob)
  int ba_every = 1;              // launch the local-map worker every n-th frame (FLVIS_BA_EVERY)
  bool lk_stats = false;         // flvis_debug_lk_stats: the LK launches count their iterations per level into counters[36 .. 59]
  hipStream_t ba_stream[NBA] = {};
  std::vector<hipStream_t> pad_streams;  // idle streams in front of the lanes' template streams (FLVIS_TPL_QPAD)
  long long ba_rr = 0;           // round-robin counter over the local-map streams
  hipEvent_t ev_in = nullptr;    // the caller's inputs are ready (recorded on the context's stream)
  bool defer_ba = false;         // inside flvis_run_steps, not its last step: the local-map launch of this frame may wait for the next frame (FLVIS_BA_START)
  bool feedback_used = false;    // flvis_correction_feed was called: k_apply_correction runs after every frame_begin
  // flvis_image_feed_host: double-buffered device staging filled by async H2D copies on a copy stream, so that the upload of
  // frame N+1 overlaps the kernels of frame N (allocated by the first call)
  struct HostFeed {
    hipStream_t strm = nullptr;
    static constexpr int SLOTS = 3;   // (mode 2 cycles through three staging slots, mode 1 through two)
    uint8_t* raw[SLOTS][2] = {};   // [slot][camera]: the images as handed over (tightly packed rows of w * bytes-per-pixel)
    uint8_t* gray[SLOTS][2] = {};  // [slot][camera]: cvtColor output for 3/4-channel input
    // mode 2 (round 6, default): nothing but copies on the copy stream.  Behind the images of a call the copy engine writes a block of
    // sequence numbers (FLAG_WORDS x the call's number, from a page-locked ring: large enough not to be turned into a blit kernel);
    // k_wait_flag on the stream that ingests the images waits for it; the host learns that the uploads are done from hipStreamQuery
    // and that a slot is free from the frame-progress word.  No event, no kernel, no AQL barrier packet on the copy stream's queue.
    int mode = 2;
    static constexpr int FLAG_WORDS = 4096, FLAG_RING = 4;
    long long* h_flag[FLAG_RING] = {};
    long long* d_flag = nullptr;
    long long slot_frame[SLOTS] = {0, 0, 0};  // the lane frame number (frames_uploaded) that read the slot last
    long long last_call_frame = -1;           // frames_fed behind this entry's last call (another entry in between: the chain is not continuous)
    size_t raw_bytes[2] = {}, gray_bytes = 0;
    hipEvent_t ev_done[2] = {}, ev_free[2] = {};
    long long n = 0;
    volatile long long* h_up = nullptr;  // host-mapped: number of calls whose uploads have finished (stored by the copy stream)
    long long* d_up = nullptr;
    std::vector<double> times;
    double ms_wait_uploads = 0, ms_issue = 0, ms_feed = 0;  // host time of the calls: blocked on the previous uploads, issuing, in flvis_image_feed
    // flvis_debug_host_feed_timing: the uploads of a call bracketed by two timing events on the copy stream (their own durations, what
    // bench.py's with_h2d.upload_GBs is made of); read back when the slot comes round again
    bool timing = false;
    hipEvent_t ev_t0[2] = {}, ev_t1[2] = {};
    bool timed[2] = {false, false};
    size_t timed_bytes[2] = {0, 0};
    double up_ms = 0, up_bytes = 0, up_calls = 0;
    hipStream_t pad_strm[4] = {};  // FLVIS_H2D_QPAD (A/B knob): streams created in front of the copy stream, so that its hardware queue is another one
  } hf;
  hipEvent_t up_event = nullptr;  // set by flvis_image_feed_host for its flvis_image_feed call: the upload's event, waited for on the stream that ingests the images
  const long long* up_flag = nullptr;  // ... or (mode 2) the sequence block the copy engine writes behind the images, and the number to wait for
  long long up_seq = 0;
  unsigned ev_flags = hipEventDisableTiming;  // flags of every event of the pipeline (FLVIS_EVENT_SCOPE)
  bool flag_joins = false;                    // FLVIS_JOIN=flag
  bool fold_joins = false;                    // FLVIS_JOIN_FOLD: the chain's own kernels wait for / store the words (KJoin)
  bool chain_continuous = false;              // this frame follows the previous one on the main stream with nothing of the caller's in between
  Lane& lane_of(int stream, int& local) {
    const int k = stream / lane_size;
    local = stream - k * lane_size;
    return *lanes[k];
  }
};

}  // namespace flvis

using namespace flvis;

namespace {

template <typename T>
T* dalloc(std::vector<void*>& allocs, size_t n, bool zero = true) {
  void* p = nullptr;
  if (hipMalloc(&p, n * sizeof(T)) != hipSuccess) return nullptr;
  if (zero) hipMemset(p, 0, n * sizeof(T));
  allocs.push_back(p);
  return (T*)p;
}

int lk_levels(int w, int h, int win, int max_level) {
  int level = 0;
  for (; level <= max_level; ++level) {
    w = (w + 1) / 2;
    h = (h + 1) / 2;
    if (w <= win || h <= win) return level;
  }
  return max_level;
}

}  // namespace
namespace flvis {
// SE3 from a row-major 4x4 as Sophus builds it (SO3(Matrix3d): Eigen's matrix -> quaternion, no normalisation) and, with
// `inverse`, Sophus' SE3::inverse() (se3.cpp:76-83): q^-1 = normalised conjugate, t^-1 = q^-1 * (-t) by the quaternion
// rotation formula -- the same operations in the same order as the CPU restatement, so that T_c_i / T_c1_c0 are bit-identical
// on both sides for a general extrinsic rotation (EuRoC), not only for an axis permutation (D435).
void pose7_from_mat44(const double* m, double* out7, bool inverse) {
  const double R[3][3] = {{m[0], m[1], m[2]}, {m[4], m[5], m[6]}, {m[8], m[9], m[10]}};
  double t[3] = {m[3], m[7], m[11]};
  double q[4];  // w x y z
  double tr = R[0][0] + R[1][1] + R[2][2];
  if (tr > 0) {
    double s = std::sqrt(tr + 1.0);
    q[0] = 0.5 * s;
    s = 0.5 / s;
    q[1] = (R[2][1] - R[1][2]) * s;
    q[2] = (R[0][2] - R[2][0]) * s;
    q[3] = (R[1][0] - R[0][1]) * s;
  } else {
    int i = 0;
    if (R[1][1] > R[0][0]) i = 1;
    if (R[2][2] > R[i][i]) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    double s = std::sqrt(R[i][i] - R[j][j] - R[k][k] + 1.0);
    double v[3];
    v[i] = 0.5 * s;
    s = 0.5 / s;
    q[0] = (R[k][j] - R[j][k]) * s;
    v[j] = (R[j][i] + R[i][j]) * s;
    v[k] = (R[k][i] + R[i][k]) * s;
    q[1] = v[0];
    q[2] = v[1];
    q[3] = v[2];
  }
  if (inverse) {
    const double cw = q[0], cx = -q[1], cy = -q[2], cz = -q[3];
    const double n = std::sqrt(cw * cw + cx * cx + cy * cy + cz * cz);
    q[0] = cw / n;
    q[1] = cx / n;
    q[2] = cy / n;
    q[3] = cz / n;
    // quat_rotate(q, -t): uv = cross(qv, v); uv += uv; v + w * uv + cross(qv, uv)
    const double v[3] = {-1.0 * t[0], -1.0 * t[1], -1.0 * t[2]};
    double uv[3] = {q[2] * v[2] - q[3] * v[1], q[3] * v[0] - q[1] * v[2], q[1] * v[1] - q[2] * v[0]};
    for (int k = 0; k < 3; k++) uv[k] = uv[k] + uv[k];
    const double c2[3] = {q[2] * uv[2] - q[3] * uv[1], q[3] * uv[0] - q[1] * uv[2], q[1] * uv[1] - q[2] * uv[0]};
    for (int k = 0; k < 3; k++) t[k] = (v[k] + q[0] * uv[k]) + c2[k];
  }
  out7[0] = t[0];
  out7[1] = t[1];
  out7[2] = t[2];
  out7[3] = q[1];
  out7[4] = q[2];
  out7[5] = q[3];
  out7[6] = q[0];
}

void glibc_seed(unsigned s, int* r34) {
  std::vector<int> v(344);
  v[0] = (int)s;
  for (int i = 1; i < 31; i++) {
    long long w = (16807LL * v[i - 1]) % 2147483647;
    if (w < 0) w += 2147483647;
    v[i] = (int)w;
  }
  for (int i = 31; i < 34; i++) v[i] = v[i - 31];
  for (int i = 34; i < 344; i++) v[i] = (int)((unsigned)v[i - 31] + (unsigned)v[i - 3]);
  for (int i = 0; i < 34; i++) r34[i] = v[344 - 34 + i];
}

}  // namespace flvis
namespace {
}  // namespace


static void lane_destroy(Lane* L) {
  if (!L) return;
  if (L->st) hipStreamSynchronize(L->st);
  if (L->det_stream) {
    hipStreamSynchronize(L->det_stream);
    hipStreamDestroy(L->det_stream);
  }
  if (L->tpl_stream) {
    hipStreamSynchronize(L->tpl_stream);
    hipStreamDestroy(L->tpl_stream);
  }
  if (L->ev_tpl) hipEventDestroy(L->ev_tpl);
  if (L->own_st && L->st) hipStreamDestroy(L->st);
  for (hipEvent_t e : {L->ev_img, L->ev_det, L->ev_gftt, L->ev_fe, L->ev_lm, L->ev_tri, L->ev_head, L->ev_endf, L->ev_stagger, L->ev_end[0], L->ev_end[1], L->ev_end[2], L->ev_end[3],
                       L->ev_end[4], L->ev_end[5], L->ev_end[6], L->ev_end[7]})
    if (e) hipEventDestroy(e);
  for (int k = 0; k < Lane::BAQ; k++)
    if (L->ev_ba_done[k]) hipEventDestroy(L->ev_ba_done[k]);
  for (void* p : L->allocs) hipFree(p);
  for (hipEvent_t e : L->prof_ev) hipEventDestroy(e);
  for (int k = 0; k < Lane::PIN_RING; k++)
    if (L->pinned[k]) hipHostFree(L->pinned[k]);
  if (L->h_progress) hipHostFree((void*)L->h_progress);
  delete L;
}

static void sync_all(flvis_ctx* ctx);
extern "C" void flvis_pipeline_sync_internal(flvis_ctx* ctx) {
  if (ctx && ctx->pipe) sync_all(ctx);
}

// the sequence number a timed-out join stored (k_wait_flag's err_word = word 2 of the lane's host-mapped progress block), or 0; cleared when read
extern "C" long long flvis_pipeline_join_timeout_internal(flvis_ctx* ctx) {
  long long seq = 0;
  if (ctx && ctx->pipe)
    for (Lane* L : ctx->pipe->lanes)
      if (L->h_progress && L->h_progress[2]) {
        seq = L->h_progress[2];
        L->h_progress[2] = 0;
      }
  return seq;
}

extern "C" void flvis_pipeline_destroy_internal(flvis_ctx* ctx) {
  if (!ctx || !ctx->pipe) return;
  Pipeline* pl = ctx->pipe;
  for (int k = 0; k < Pipeline::NBA; k++)
    if (pl->ba_stream[k]) hipStreamSynchronize(pl->ba_stream[k]);
  for (Lane* L : pl->lanes) lane_destroy(L);
  for (int k = 0; k < Pipeline::NBA; k++)
    if (pl->ba_stream[k]) hipStreamDestroy(pl->ba_stream[k]);
  for (hipStream_t ps : pl->pad_streams) {
    hipStreamSynchronize(ps);
    hipStreamDestroy(ps);
  }
  pl->pad_streams.clear();
  if (pl->hf.strm) {
    hipStreamSynchronize(pl->hf.strm);
    hipStreamDestroy(pl->hf.strm);
    for (int k = 0; k < 2; k++) {
      if (pl->hf.ev_done[k]) hipEventDestroy(pl->hf.ev_done[k]);
      if (pl->hf.ev_free[k]) hipEventDestroy(pl->hf.ev_free[k]);
      if (pl->hf.ev_t0[k]) hipEventDestroy(pl->hf.ev_t0[k]);
      if (pl->hf.ev_t1[k]) hipEventDestroy(pl->hf.ev_t1[k]);
    }
    for (hipStream_t ps : pl->hf.pad_strm)
      if (ps) hipStreamSynchronize(ps), hipStreamDestroy(ps);
    if (pl->hf.h_up) hipHostFree((void*)pl->hf.h_up);
    for (long long* fp : pl->hf.h_flag)
      if (fp) hipHostFree(fp);
  }
  if (pl->ev_in) hipEventDestroy(pl->ev_in);
  for (void* p : pl->allocs) hipFree(p);
  delete pl;
  ctx->pipe = nullptr;
}

// device state + streams of one lane; false on any allocation failure (the caller destroys the pipeline)
static bool lane_create(flvis_ctx* ctx, Pipeline* pl, Lane* L, int s0, int S, uint64_t seed_base, int traj_capacity, bool own_stream) {
  const flvis_cfg* cfg = &pl->cfg;
  const int w = cfg->image_width, h = cfg->image_height;
  L->s0 = s0;
  L->S = S;
  Pipe& p = L->pipe;
  memset(&p, 0, sizeof(p));
  p.S = S;
  CamParams& c = p.cam;
  c.cam_type = cfg->cam_type;
  c.w = w;
  c.h = h;
  c.fx = cfg->P0[0];
  c.fy = cfg->P0[5];
  c.cx = cfg->P0[2];
  c.cy = cfg->P0[6];
  memcpy(c.K0, cfg->cam0_intrinsics, 32);
  memcpy(c.D0, cfg->cam0_distortion, 32);
  memcpy(c.K1, cfg->cam1_intrinsics, 32);
  memcpy(c.D1, cfg->cam1_distortion, 32);
  memcpy(c.R0, cfg->R0, 72);
  memcpy(c.R1, cfg->R1, 72);
  memcpy(c.P0, cfg->P0, 96);
  memcpy(c.P1, cfg->P1, 96);
  pose7_from_mat44(cfg->T_cam0_cam1, c.T_c1_c0, true);
  pose7_from_mat44(cfg->T_imu_cam0, c.T_i_c, false);
  pose7_from_mat44(cfg->T_imu_cam0, c.T_c_i, true);
  c.iir_ratio = (float)cfg->dr_para[0];
  c.range = (float)cfg->dr_para[1];
  c.depth_scale = cfg->depth_factor;
  c.enable_dummy = !(cfg->dr_para[2] < 0.5);
  c.need_equal_hist = cfg->need_equal_hist;
  c.skip_first_n = cfg->skip_first_n_imgs;
  for (int i = 0; i < 4; i++) c.vi_para[i] = cfg->vifusion_para[i];
  c.dem.regionWidth = (int)std::floor(w / 4.0);
  c.dem.regionHeight = (int)std::floor(h / 4.0);
  c.dem.boundary_dis = (int)std::floor(cfg->feature_para[2] / 2.0);
  c.dem.max_region_feature_num = (unsigned)cfg->feature_para[0];
  c.gftt_num = (int)cfg->feature_para[3];
  c.gftt_ql = cfg->feature_para[4];
  c.gftt_dis = (int)cfg->feature_para[5];
  c.window = cfg->window_size;
  c.seed = seed_base;

  bool ok = true;
#define DA(field, T, n) ok = ok && ((p.field = dalloc<T>(L->allocs, (n))) != nullptr)
  DA(st, StreamState, S);
  DA(lm, Landmark, (size_t)2 * S * NMAX);
  DA(vi, MotionState, (size_t)S * VI_QUEUE);
  DA(imu_out, double, (size_t)S * IMU_OUT_CAP * 11);
  DA(prev_pts, float, (size_t)S * NMAX * 2);
  DA(next_pts, float, (size_t)S * NMAX * 2);
  DA(lk_status, uint8_t, (size_t)S * NMAX);
  DA(lk_count, int, S);
  DA(lk_slot, int, (size_t)S * NMAX);
  DA(lk_tag, long long, S);
  DA(tpl_pts, float, (size_t)S * NMAX * 2);
  DA(tpl_count, int, S);
  DA(tpl_tag, long long, S);
  p.tpl_ahead = 0;
  DA(m1, float, (size_t)S * NMAX * 2);
  DA(m2, float, (size_t)S * NMAX * 2);
  DA(tri, double, (size_t)S * NMAX * 3);
  DA(tri_mask, uint8_t, (size_t)S * NMAX);
  DA(new_xy, float, (size_t)S * NEW_MAX * 2);
  DA(n_new, int, S);
  DA(exist_xy, double, (size_t)S * NMAX * 2);
  DA(n_exist, int, S);
  DA(act_img, int, S);
  DA(act_track, int, S);
  DA(det_mode, int, S);
  DA(det_maxc, int, S);
  DA(gftt_act, int, S);
  DA(gftt_maxc, int, S);
  DA(img_slot, int, S);
  DA(img_slot_in, int, S);
  DA(out, FrameOut, S);
  DA(kfq, KeyFrameDev, (size_t)S * KFQ);
  DA(kfq_tail, unsigned, S);
  DA(kfq_head, unsigned, S);
  DA(ba_busy, int, S);
  DA(ba_plan, unsigned, (size_t)BA_PLAN_SLOTS * (3 + S));
  DA(win, WindowDev, S);
  DA(kfs_ring, KeyFrameDev, (size_t)S * BA_WMAX);
  DA(corr, CorrectionDev, S);
  DA(rec_id, int, (size_t)S * POSE_REC);
  DA(rec_T, double, (size_t)S * POSE_REC * 7);
  p.corr_in = nullptr;  // allocated by the first flvis_correction_feed
  DA(counters, long long, 64);
  p.ba_scratch_stride = ba_scratch_doubles();
  p.imu_factor = 0;
  p.imu_sigma_g = 0;
  p.imu_sigma_a = 0;
  // Keyframes a local-map workgroup takes from its stream's queue per launch.  A workgroup that kept draining a busy stream's queue
  // (0: the behaviour up to round 4) made its LAUNCH last as long as that stream stayed busy, the launches behind it on the same HIP
  // stream -- with every other stream's keyframes -- waited, and the tracker either met the back-pressure below (5-8 ms frame chains in
  // one run of three at KFQ = 16) or left the local map a backlog at the end of the run (7-18 ms at KFQ = 32), although no queue was
  // near full.  With one keyframe per launch every launch is one optimisation long, the launch that follows every frame takes the
  // stream's next keyframe, and the two local-map streams never fall behind: 54.2-54.3k frames/s in four runs of four, against
  // 36.8-54.1k (profiles/r04_local_map_and_streams_ab.md).
  p.ba_drain = 1;
  if (const char* e = getenv("FLVIS_BA_DRAIN")) p.ba_drain = std::max(0, atoi(e));
  // ... with a bound: an owner that finds KFQ / 2 or more keyframes still waiting after its share stays and goes on (a stream whose
  // optimisations take longer than its keyframes arrive -- every frame a keyframe -- would otherwise grow a backlog that only ends at
  // the full queue, where k_frame_end drops keyframes).  A launch takes at least the keyframes of the frames between two launches.
  if (p.ba_drain > 0) p.ba_drain = std::max(p.ba_drain, pl->ba_every);
  p.ba_backlog = KFQ / 2;
  // FLVIS_BA_REMAP=1 (round 5, A/B knob): workgroup r of a local-map launch serves the r-th stream with a keyframe waiting, not stream r,
  // so that the working workgroups -- a CU each -- are dealt to the XCDs evenly (k_ba_worker).  Measured (one box, two runs each): 57.3k /
  // 57.0k frames/s against 57.4k / 57.3k, LK launches 0.226 / 0.264 ms either way: which XCD loses the CUs is not what the LK pays for
  p.ba_remap = getenv("FLVIS_BA_REMAP") && atoi(getenv("FLVIS_BA_REMAP")) == 1;
  p.ba_mfma = 0;
  if (const char* e = getenv("FLVIS_BA_MFMA")) p.ba_mfma = atoi(e) != 0;
  // FLVIS_BA_BALANCE=1 (opt-in, round 6): the Schur accumulate's lanes per pose pair in proportion to the landmarks the pair shares
  // (ba_balance_pairs) instead of 16 each.  Measured (profiles/r06_ba_phases.md): the benchmark's windows share their landmarks almost
  // evenly (155 per pose, 76-114 per pair of poses), the per-SIMD sums of the accumulate are 262-300 us either way -- the "wait for the
  // slowest wave" of round 5 is the younger wave of each SIMD finishing behind the older one, not an imbalance -- and the partition's own
  // cost (+27 us structure, +19 us combine per optimisation) is not paid back: off by default.
  // FLVIS_KF_CHECK=1 (test knob): k_frame_end leaves a checksum of the keyframe's landmark arrays (written by all of its waves) in the
  // payload, the local-map worker -- another workgroup, usually on another XCD -- recomputes it from what it reads after its acquire:
  // debug counters 30 (payloads checked) and 31 (mismatches).  The hand-over publishes with ONE agent-scope release by one thread behind a
  // workgroup barrier; tests/test_gpu_pipeline.py runs 64 streams under this check.
  p.kf_check = getenv("FLVIS_KF_CHECK") && atoi(getenv("FLVIS_KF_CHECK")) == 1;
  p.ba_balance = 0;
  if (const char* e = getenv("FLVIS_BA_BALANCE")) p.ba_balance = atoi(e) != 0;
  // LDS a local-map workgroup claims (FLVIS_BA_LDS_KB, 64 .. 159): whatever it leaves of the CU's 160 KB lets LK / corner-response
  // waves run on the same CU, whose SIMDs a latency-bound BA workgroup keeps mostly idle
  p.ba_lds_bytes = ba_lds_budget_max();
  if (const char* e = getenv("FLVIS_BA_LDS_KB")) {
    const int kb = atoi(e);
    if (kb >= 64 && kb * 1024 <= ba_lds_budget_max()) p.ba_lds_bytes = kb * 1024;
  }  // A/B knob, see DESIGN.md section 4
  DA(ba_scratch, double, (size_t)S * p.ba_scratch_stride);
  // FLVIS_PNP_TAIL=cv (opt-in fidelity mode, round 6): behind k_ransac_pnp the pose of the ITERATIVE flag is replaced by what
  // cv::solvePnP(ITERATIVE, useExtrinsicGuess = false) leaves on the RANSAC's inliers -- a DLT start and CvLevMarq, the very function the
  // checker's `make -C oracle TAIL=cv` build runs (cv_solvers.hpp), bit-identical to it (tests/test_gpu_pipeline.py) -- instead of the
  // Gauss-Newton refinement of the winning model.  One wave per stream: the small dense algebra (the 12 x 12 and 6 x 6 Jacobi SVDs, OpenCV's
  // loops as they are written) redundantly in every lane, the sums over the correspondences dealt to the lanes; still milliseconds per
  // frame, which is why it is not the default (the two tails agree to 4.4e-9 m on the first tracked frames: the checker's README).
  p.pnp_tail_cv = 0;
  p.pnp_tail_ws = nullptr;
  p.pnp_tail_stride = 0;
  if (const char* e = getenv("FLVIS_PNP_TAIL")) p.pnp_tail_cv = !strcmp(e, "cv");
  if (p.pnp_tail_cv) {
    p.pnp_tail_stride = (size_t)29 * NMAX + 192;  // world points (3 n), pixels (2 n), find_extrinsic_iterative's work (24 n + 192)
    DA(pnp_tail_ws, double, (size_t)S * p.pnp_tail_stride);
  }
  unsigned long long* seeds = dalloc<unsigned long long>(L->allocs, S);
  ok = ok && seeds;
  p.seeds = seeds;
  p.traj_cap = traj_capacity > 0 ? traj_capacity : 0;
  if (p.traj_cap) DA(traj, double, (size_t)S * p.traj_cap * 9);
#undef DA
  // per-frame input block
  const size_t off_imu = sizeof(double) * S, off_tab = off_imu + sizeof(double) * (size_t)S * IMU_MAX * 7,
               off_n = off_tab + 2 * sizeof(void*);
  L->input_bytes = off_n + sizeof(int) * S;
  L->in_off_imu = off_imu, L->in_off_tab = off_tab, L->in_off_n = off_n;
  ok = ok && ((L->d_inputs = dalloc<uint8_t>(L->allocs, L->input_bytes)) != nullptr);
  if (ok) {
    L->d_time = reinterpret_cast<double*>(L->d_inputs);
    p.imu_in = reinterpret_cast<double*>(L->d_inputs + off_imu);
    L->d_tab = reinterpret_cast<const uint8_t**>(L->d_inputs + off_tab);
    p.in_tab = L->d_tab;
    p.n_imu = reinterpret_cast<int*>(L->d_inputs + off_n);
  }
  // image pyramids
  for (int l = 0; l <= pl->levels; l++) {
    for (int k = 0; k < 2; k++) ok = ok && ((L->pyr0[k][l] = dalloc<uint8_t>(L->allocs, pl->lstride[l] * S + 256)) != nullptr);
    ok = ok && ((L->pyr1[l] = dalloc<uint8_t>(L->allocs, pl->lstride[l] * S + 256)) != nullptr);
    if (ok) {  // the level pointers address pixel (0, 0) of stream 0 inside the padded buffers
      const size_t org = (size_t)pl->lby * pl->lpitch[l] + pl->lbx;
      for (int k = 0; k < 2; k++) L->pyr0[k][l] += org;
      L->pyr1[l] += org;
    }
  }
  // LK template cache: the stereo matcher's templates of frame t are the temporal tracker's templates of frame t + 1 (LKParams::tc).
  // Only rigs with a stereo matcher have one; FLVIS_LK_TCACHE=0 turns it off (A/B knob, and the reference point of the cache's test).
  p.tc_cap = 0;
  p.tc = nullptr;
  p.tc_stride = 0;
  L->tc = nullptr;
  {
    const char* e = getenv("FLVIS_LK_TCACHE");
    const bool on = !(e && atoi(e) == 0) && pl->cfg.cam_type != CAM_DEPTH && pl->levels_s == pl->levels_t;
    // FLVIS_TPL_AHEAD=1 (round 6, opt-in A/B knob; default 0 = rounds 4-5): the templates of the landmarks the temporal tracker has followed
    // into the frame are computed by k_lk_templates_ahead beside the frame's geometry kernels, the stereo launch takes them from the cache
    // (lane_frame).  Needs the corner detection behind the F-RANSAC (FLVIS_DET_START >= 2, the default): the signal that starts it starts
    // this too.  Bit-identical results (test_templates_ahead_leave_the_same_results).  Measured (profiles/r06_templates_ahead.md): the
    // stereo launch 0.259 -> 0.215 ms, and the kernels the template kernel runs beside pay it back -- k_ransac_pnp 0.187 -> 0.211 ms,
    // k_gftt_pick 0.121 -> 0.157 ms (FeatureDEM then waits 25 us longer for the corners): 59.0k against 59.9k frames/s; and only with the
    // template stream on a hardware queue of its own that does not share the main or the detection stream's pipe (FLVIS_TPL_QPAD,
    // GPU_MAX_HW_QUEUES=8): 47.7k on the detection stream's pipe, 38.6k on the main stream's.
    const char* ea = getenv("FLVIS_TPL_AHEAD");
    const bool det_late = !(getenv("FLVIS_DET_START") && atoi(getenv("FLVIS_DET_START")) < 2) &&
                          !(getenv("FLVIS_DET_ORDER") && atoi(getenv("FLVIS_DET_ORDER")) == 0);
    const bool ahead = ea && atoi(ea) == 1 && det_late;
    pl->tpl_start = getenv("FLVIS_TPL_START") && atoi(getenv("FLVIS_TPL_START")) == 0 ? 0 : 1;
    if (ok && on) {
      // slots: the stereo launch's points (rounds 4-5) or, with templates ahead, the survivors of the temporal tracker followed by the
      // frame's new landmarks
      const int cap = std::min(NMAX, (pl->max_pts + 63) / 64 * 64 * (ahead ? 2 : 1));
      L->tc_stride = lk_tc_slot_dwords(pl->levels_t);
      const size_t n = (size_t)S * cap * L->tc_stride;
      L->tc = dalloc<uint32_t>(L->allocs, n, false);
      if (L->tc) {
        hipMemset(L->tc, 0xff, n * sizeof(uint32_t));  // no header matches a position or a frame id
        p.tc_cap = cap;
        p.tc = L->tc;
        p.tc_stride = L->tc_stride;
        p.tpl_ahead = ahead ? 1 : 0;
      } else {
        (void)hipGetLastError();  // no memory for it: run without
      }
    }
  }
  // GFTT scratch
  int cap = 1;
  while (cap < (w / 2 + 1) * (h / 2 + 1)) cap <<= 1;
  L->gftt.cap = cap;
  ok = ok && ((L->gftt.maxenc = dalloc<unsigned>(L->allocs, S)) != nullptr);
  ok = ok && ((L->gftt.nkeys = dalloc<int>(L->allocs, S)) != nullptr);
  ok = ok && ((L->gftt.keys = dalloc<unsigned long long>(L->allocs, (size_t)cap * S, false)) != nullptr);
  ok = ok && ((L->gftt_xy = dalloc<float>(L->allocs, (size_t)S * 2 * c.gftt_num * 2)) != nullptr);
  ok = ok && ((L->gftt_n = dalloc<int>(L->allocs, S)) != nullptr);
  ok = ok && ((L->dem_sorted = dalloc<float>(L->allocs, (size_t)S * 2 * c.gftt_num * 2)) != nullptr);
  ok = ok && ((L->dem_roff = dalloc<int>(L->allocs, (size_t)S * 17)) != nullptr);
  ok = ok && ((L->eq_hist = dalloc<unsigned>(L->allocs, (size_t)S * 256)) != nullptr);
  ok = ok && ((L->eq_lut = dalloc<uint8_t>(L->allocs, (size_t)S * 256)) != nullptr);
  ok = ok && ((L->d_join = dalloc<long long>(L->allocs, (size_t)Lane::JOIN_IDS * 8)) != nullptr);
  ok = ok && ((L->d_join_cnt = dalloc<unsigned>(L->allocs, (size_t)Lane::JOIN_IDS)) != nullptr);
  if (!ok) return false;
  // initial per-stream state (F2FTracking::init, VIMOTION ctor, landmark id counter 100, glibc rand seed 1)
  std::vector<StreamState> hs(S);
  memset(hs.data(), 0, sizeof(StreamState) * S);
  std::vector<unsigned long long> hseed(S);
  for (int s = 0; s < S; s++) {
    StreamState& st = hs[s];
    st.state = ST_UNINIT;
    st.cur = 0;
    st.skip_n = cfg->skip_first_n_imgs;
    st.lm_id_counter = 100;
    st.vi_first = 1;
    st.kf_dq[0] = 1.0;
    for (int k = 0; k < 2; k++) st.T_c_w[k][6] = 1.0;
    st.T_kf[6] = 1.0;
    st.guess[6] = 1.0;
    glibc_seed(1, st.rnd_r);
    st.rnd_pos = 0;
    hseed[s] = seed_base + (unsigned long long)(s0 + s);  // the seed of a stream does not depend on the lane partition
  }
  hipMemcpy(p.st, hs.data(), sizeof(StreamState) * S, hipMemcpyHostToDevice);
  {
    std::vector<int> one(S, 1);  // cur = 0: the first image goes to slot 1
    hipMemcpy(p.img_slot_in, one.data(), sizeof(int) * S, hipMemcpyHostToDevice);
  }
  hipMemcpy(seeds, hseed.data(), sizeof(unsigned long long) * S, hipMemcpyHostToDevice);
  L->h_imu.assign((size_t)S * IMU_MAX * 7, 0.0);
  L->h_nimu.assign(S, 0);
  L->imu_read.assign(S, 0);
  for (int k = 0; k < Lane::PIN_RING; k++)
    if (hipHostMalloc(&L->pinned[k], L->input_bytes, hipHostMallocMapped) != hipSuccess) return false;
  for (int k = 0; k < Lane::PIN_RING; k++) {
    void* dv = nullptr;
    if (hipHostGetDevicePointer(&dv, L->pinned[k], 0) != hipSuccess) {
      (void)hipGetLastError();
      dv = nullptr;
    }
    L->pinned_dev[k] = (uint8_t*)dv;
  }
  {
    void* hp = nullptr;
    void* dp = nullptr;
    if (hipHostMalloc(&hp, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess || hipHostGetDevicePointer(&dp, hp, 0) != hipSuccess) return false;
    L->h_progress = (volatile long long*)hp;
    L->d_progress = (long long*)dp;
    for (int k = 0; k < 8; k++) L->h_progress[k] = 0;  // (word 0: frame progress, 1: scratch, 2: a timed-out join's sequence number)
  }
  if (own_stream) {
    if (hipStreamCreateWithFlags(&L->st, hipStreamNonBlocking) != hipSuccess) return false;
    L->own_st = true;
  } else {
    L->st = ctx->stream;
  }
  // (FLVIS_DET_PRIO=1, A/B knob: the detection stream at the lowest queue priority, so that corner-response workgroups that start beside
  // the tracking chain's kernels do not take a compute unit one of those is waiting for)
  bool evok;
  if (getenv("FLVIS_DET_PRIO") && atoi(getenv("FLVIS_DET_PRIO")) != 0) {
    int lo = 0, hi = 0;
    hipDeviceGetStreamPriorityRange(&lo, &hi);
    evok = hipStreamCreateWithPriority(&L->det_stream, hipStreamNonBlocking, lo) == hipSuccess;
  } else {
    evok = hipStreamCreateWithFlags(&L->det_stream, hipStreamNonBlocking) == hipSuccess;
  }
  if (L->pipe.tpl_ahead) evok = evok && hipEventCreateWithFlags(&L->ev_tpl, pl->ev_flags) == hipSuccess;  // (its stream: flvis_tracker_create)
  static_assert(Lane::HOLD_RING == 8, "event list below");  // (ev_endf included)
  for (hipEvent_t* e : {&L->ev_img, &L->ev_det, &L->ev_gftt, &L->ev_fe, &L->ev_lm, &L->ev_tri, &L->ev_head, &L->ev_endf, &L->ev_stagger, &L->ev_end[0], &L->ev_end[1], &L->ev_end[2],
                        &L->ev_end[3], &L->ev_end[4], &L->ev_end[5], &L->ev_end[6], &L->ev_end[7]})
    evok = evok && hipEventCreateWithFlags(e, pl->ev_flags) == hipSuccess;
  for (int k = 0; k < Lane::BAQ && evok; k++)
    evok = hipEventCreateWithFlags(&L->ev_ba_done[k], pl->ev_flags) == hipSuccess;
  return evok;
}

extern "C" {

int flvis_tracker_create(flvis_ctx* ctx, const flvis_cfg* cfg, int n_streams, uint64_t seed_base, int traj_capacity) {
  if (!ctx || !cfg || n_streams <= 0) return FLVIS_ERR_INVALID_ARG;
  if (ctx->pipe) flvis_pipeline_destroy_internal(ctx);
  const int w = cfg->image_width, h = cfg->image_height;
  if (w < 64 || h < 64) return ctx->fail(FLVIS_ERR_CONFIG, "image must be at least 64 x 64");
  if (cfg->need_equal_hist && (w & 15)) return ctx->fail(FLVIS_ERR_CONFIG, "equalizeHist rigs need an image width that is a multiple of 16");
  if (cfg->feature_para[5] > 64.0) return ctx->fail(FLVIS_ERR_CAPACITY, "feature_para6 (GFTT minDistance) > 64 is not supported");
  if (cfg->window_size > BA_WMAX) return ctx->fail(FLVIS_ERR_CAPACITY, "window_size exceeds the LDS-resident solver (16)");
  if (16 * (int)cfg->feature_para[0] > 512) return ctx->fail(FLVIS_ERR_CAPACITY, "feature_para1 (landmarks per region) must be <= 32: the pose LM holds 512 edges");
  if ((int)cfg->feature_para[3] * 2 > 4096) return ctx->fail(FLVIS_ERR_CAPACITY, "feature_para4 (gftt_num) must be <= 2048");
  if ((size_t)((w + 31) / 32) * h * 4 > 96 * 1024) return ctx->fail(FLVIS_ERR_CAPACITY, "image too large for the GFTT LDS bitmap");
  if (cfg->cam_type == CAM_DEPTH && !(cfg->depth_factor > 0)) return ctx->fail(FLVIS_ERR_CONFIG, "depth mode needs depth_factor > 0");
  hipSetDevice(ctx->device);
  Pipeline* pl = new Pipeline();
  ctx->pipe = pl;
  // FLVIS_EVENT_SCOPE=agent (round 6, A/B knob): events created with hipEventReleaseToDevice.  The runtime turns every hipEventRecord into a
  // barrier packet that acquires and releases at SYSTEM scope (AMD_LOG_LEVEL=4: "BarrierValue ... acquire=2, release=2", ~30 per frame) where
  // agent scope would do for events that order kernels of one device.  Measured: the flag changes neither the logged header nor the
  // rate (58.6k / 56.7k frames/s resident / host images with it, 58.5k / 56.5k without): left off.
  {
    const char* e = getenv("FLVIS_EVENT_SCOPE");
    if (e && !strcmp(e, "agent")) pl->ev_flags |= hipEventReleaseToDevice;
    const char* j = getenv("FLVIS_JOIN");
    pl->flag_joins = !(j && !strcmp(j, "event"));  // (default since round 6: 58.6k -> 60.4k frames/s, chain p50 1.047 -> 1.012 ms; "event": rounds 1-5)
    // Under a profiler that collects hardware counters (rocprofv3 --pmc sets ROCPROF_COUNTER_COLLECTION in the application's environment)
    // kernels run ONE AT A TIME: a k_wait_flag that sleeps until a kernel of another stream has stored its word would wait for a kernel
    // that cannot start, and end by its 4 s limit.  Events are resolved by the command processor between kernels: taken there unless
    // FLVIS_JOIN says otherwise.  (The wait for an upload is not affected: the copy engine runs beside a serialised kernel.)
    if (!j) {
      const char* cc = getenv("ROCPROF_COUNTER_COLLECTION");
      if (cc && cc[0] && strcmp(cc, "0") && strcmp(cc, "False") && strcmp(cc, "false")) pl->flag_joins = false;
    }
    const char* f = getenv("FLVIS_JOIN_FOLD");
    pl->fold_joins = pl->flag_joins && !(f && atoi(f) == 0);
  }
  const int S = n_streams;
  pl->S = S;
  pl->cfg = *cfg;
  pl->max_pts = std::min(NMAX, 16 * (int)cfg->feature_para[0]);
  // pyramid geometry
  pl->levels_t = lk_levels(w, h, 31, 10);
  pl->levels_s = lk_levels(w, h, 31, 5);
  pl->levels = std::max(pl->levels_t, pl->levels_s);
  if (pl->levels >= LK_MAX_LEVELS) pl->levels = LK_MAX_LEVELS - 1;
  {
    const char* e = getenv("FLVIS_LK_BORDER");
    const bool on = !(e && atoi(e) == 0);
    pl->lbx = on ? LK_BORDER_X : 0;
    pl->lby = on ? LK_BORDER_Y : 0;
  }
  int lw = w, lh = h;
  for (int l = 0; l <= pl->levels; l++) {
    pl->lw[l] = lw;
    pl->lh[l] = lh;
    // levels are stored with a physical BORDER_REFLECT_101 border (img_kernels.hpp: LK_BORDER_X / LK_BORDER_Y; FLVIS_LK_BORDER=0: none,
    // A/B knob): the level pointers address pixel (0, 0), rows -lby .. lh + lby - 1 and columns -lbx .. pitch - lbx - 1 are storage
    pl->lpitch[l] = align_up(lw, 16) + 2 * pl->lbx;
    pl->lstride[l] = (size_t)pl->lpitch[l] * (lh + 2 * pl->lby) + 64;
    pl->lstride[l] = (pl->lstride[l] + 63) / 64 * 64;
    lw = (lw + 1) / 2;
    lh = (lh + 1) / 2;
  }
  // lanes: one by default.  Measured on MI355X with 64 streams (bench.py, round 2): 1 lane 1.67 ms per step, 2 lanes 2.23 ms,
  // 4 lanes 2.94 ms -- a lane's one-workgroup-per-stream geometry kernels share their CUs with the other lanes' LK waves and
  // slow down by more than the overlap gains.  FLVIS_LANES (1..16) is kept as a tuning knob.
  int n_lanes = 1;
  if (const char* e = getenv("FLVIS_LANES")) {
    int v = atoi(e);
    if (v >= 1 && v <= 16) n_lanes = v;
  }
  n_lanes = std::min(n_lanes, S);
  pl->lane_size = (S + n_lanes - 1) / n_lanes;
  n_lanes = (S + pl->lane_size - 1) / pl->lane_size;
  if (const char* e = getenv("FLVIS_BA_STREAMS")) {
    int v = atoi(e);
    if (v >= 1 && v <= 4) pl->nba_lane = v;
  }
  pl->nba_lane = std::max(1, std::min(pl->nba_lane, Pipeline::NBA / n_lanes));
  pl->nba = std::min(Pipeline::NBA, pl->nba_lane * n_lanes);  // (more than NBA / nba_lane lanes share local-map streams)
  if (const char* e = getenv("FLVIS_LANE_STAGGER")) pl->stagger = atoi(e) != 0;
  if (const char* e = getenv("FLVIS_HOST_LEAD")) pl->host_lead = std::max(1, std::min(atoi(e), (int)Lane::PIN_RING));
  if (const char* e = getenv("FLVIS_SYNC_EACH_FRAME")) pl->sync_each_frame = atoi(e) != 0;
  if (const char* e = getenv("FLVIS_CHAIN_MERGE")) pl->chain_merge = atoi(e) & 3;
  if (const char* e = getenv("FLVIS_LK_ORDER")) pl->lk_order = atoi(e) & 3;
  if (const char* e = getenv("FLVIS_BA_EVERY")) {
    int v = atoi(e);
    if (v >= 1 && v <= KFQ / 4) pl->ba_every = v;  // (the back-pressure in lane_frame needs (D + 2) * ba_every <= KFQ / 2)
  }
  bool ok = true;
  for (int k = 0; k < n_lanes && ok; k++) {
    Lane* L = new Lane();
    L->idx = k;
    pl->lanes.push_back(L);
    const int s0 = k * pl->lane_size;
    ok = lane_create(ctx, pl, L, s0, std::min(pl->lane_size, S - s0), seed_base, traj_capacity, n_lanes > 1);
  }
  // the local map must not displace the tracking chain: its streams get the lowest queue priority
  int prio_least = 0, prio_greatest = 0;
  hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
  if (const char* e = getenv("FLVIS_BA_PRIORITY")) prio_least = atoi(e);  // tuning knob
  for (int k = 0; k < pl->nba && ok; k++)
    ok = hipStreamCreateWithPriority(&pl->ba_stream[k], hipStreamNonBlocking, prio_least) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&pl->ev_in, pl->ev_flags) == hipSuccess;
  // the streams of k_lk_templates_ahead, created LAST and behind FLVIS_TPL_QPAD idle streams: a fifth compute queue shares a pipe of the
  // command processor with one of the first four (queue i sits on pipe i mod 4 in creation order, profiles/r06_h2d.md), and whichever
  // queue that is pays for the neighbour in every dispatch -- it must not be the main stream's
  {
    int lo = 0, hi = 0;
    hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (const char* e = getenv("FLVIS_TPL_PRIO")) lo = atoi(e);  // A/B knob
    const int qpad = getenv("FLVIS_TPL_QPAD") ? std::max(0, std::min(atoi(getenv("FLVIS_TPL_QPAD")), 8)) : 2;
    bool any = false;
    for (Lane* L : pl->lanes) any = any || L->pipe.tpl_ahead;
    for (int k = 0; k < qpad && ok && any; k++) {
      hipStream_t ps = nullptr;
      ok = hipStreamCreateWithPriority(&ps, hipStreamNonBlocking, lo) == hipSuccess;
      if (ok) {
        pl->pad_streams.push_back(ps);
        launch_store_progress(ps, pl->lanes[0]->d_progress + 1, 0);  // (a scratch word: the stream gets its hardware queue)
      }
    }
    for (Lane* L : pl->lanes)
      if (ok && L->pipe.tpl_ahead) ok = hipStreamCreateWithPriority(&L->tpl_stream, hipStreamNonBlocking, lo) == hipSuccess;
  }
  if (!ok) {
    flvis_pipeline_destroy_internal(ctx);
    (void)hipGetLastError();
    return ctx->fail(FLVIS_ERR_HIP, "tracker_create: device / pinned allocation or stream creation failed");
  }
  if (ba_kernels_init() != hipSuccess || track_kernels_init() != hipSuccess) {
    (void)hipGetLastError();
    flvis_pipeline_destroy_internal(ctx);
    return ctx->fail(FLVIS_ERR_HIP, "tracker_create: cannot reserve LDS for the BA kernel");
  }
  hipDeviceSynchronize();
  return FLVIS_OK;
}

// The reference integrates every IMU message as it arrives, without a limit.  Here samples are staged per stream and consumed
// by the next image_feed; when a stream's staging slot (IMU_MAX samples) is full -- an IMU that leads the camera, a dropped
// image -- the staged samples of the lane are integrated at once (blocking upload + k_imu_feed: a rare path) and staging goes on.
static int lane_flush_imu(flvis_ctx* ctx, Lane& L) {
  hipError_t e = hipStreamSynchronize(L.st);
  // (the device block of the copying mode: idle after the synchronisation whichever mode the frames use)
  L.pipe.imu_in = reinterpret_cast<double*>(L.d_inputs + L.in_off_imu);
  L.pipe.n_imu = reinterpret_cast<int*>(L.d_inputs + L.in_off_n);
  if (e == hipSuccess) e = hipMemcpy(L.pipe.imu_in, L.h_imu.data(), sizeof(double) * L.h_imu.size(), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(L.pipe.n_imu, L.h_nimu.data(), sizeof(int) * L.S, hipMemcpyHostToDevice);
  if (e != hipSuccess) return ctx->hip_fail(e, "imu_feed (flush)");
  launch_imu_feed(L.st, L.pipe);
  e = hipStreamSynchronize(L.st);
  if (e != hipSuccess) return ctx->hip_fail(e, "imu_feed (flush)");
  std::fill(L.h_nimu.begin(), L.h_nimu.end(), 0);
  return FLVIS_OK;
}

int flvis_imu_feed_flvis_frame(flvis_ctx* ctx, int stream, int n, const double* samples7) {
  if (!ctx || !ctx->pipe || !samples7) return FLVIS_ERR_INVALID_ARG;
  Pipeline* pl = ctx->pipe;
  if (stream < 0 || stream >= pl->S || n < 0) return ctx->fail(FLVIS_ERR_INVALID_ARG, "imu_feed: bad stream");
  int ls;
  Lane& L = pl->lane_of(stream, ls);
  while (n > 0) {
    int& cnt = L.h_nimu[ls];
    if (cnt == IMU_MAX) {
      hipSetDevice(ctx->device);
      const int rc = lane_flush_imu(ctx, L);
      if (rc != FLVIS_OK) return rc;
    }
    const int m = std::min(n, IMU_MAX - L.h_nimu[ls]);
    memcpy(&L.h_imu[((size_t)ls * IMU_MAX + L.h_nimu[ls]) * 7], samples7, sizeof(double) * 7 * m);
    L.h_nimu[ls] += m;
    samples7 += 7 * (size_t)m;
    n -= m;
  }
  return FLVIS_OK;
}

int flvis_imu_feed(flvis_ctx* ctx, int stream, double t, const double* a, const double* g) {
  if (!ctx || !ctx->pipe || !a || !g) return FLVIS_ERR_INVALID_ARG;
  double s[7];
  s[0] = t;
  if (ctx->pipe->cfg.imu_type == 3) return ctx->fail(FLVIS_ERR_CONFIG, "imu_feed: this rig has no IMU (type_of_vi 4: imu_type NONE)");
  switch (ctx->pipe->cfg.imu_type) {  // src/frontend/vo_tracking.cpp:331-357
    case 0:                            // D435I
      s[1] = -a[2]; s[2] = a[0]; s[3] = a[1];
      s[4] = g[2]; s[5] = -g[0]; s[6] = -g[1];
      break;
    case 1:  // EuRoC_MAV
      s[1] = -a[2]; s[2] = a[1]; s[3] = -a[0];
      s[4] = g[2]; s[5] = -g[1]; s[6] = g[0];
      break;
    default:  // PIXHAWK
      s[1] = -a[0]; s[2] = -a[1]; s[3] = -a[2];
      s[4] = g[0]; s[5] = g[1]; s[6] = g[2];
      break;
  }
  return flvis_imu_feed_flvis_frame(ctx, stream, 1, s);
}

// F2FTracking::imu_feed's outputs (f2f_tracking.cpp:46-57), fetched in batches: the rows written since the previous call.
int flvis_get_imu_states(flvis_ctx* ctx, int stream, int cap, double* h_rows11, int* n_out, int* n_dropped) {
  if (!ctx || !ctx->pipe || !n_out || cap < 0 || (cap > 0 && !h_rows11)) return FLVIS_ERR_INVALID_ARG;
  Pipeline* pl = ctx->pipe;
  if (stream < 0 || stream >= pl->S) return ctx->fail(FLVIS_ERR_INVALID_ARG, "get_imu_states: bad stream");
  hipSetDevice(ctx->device);
  int ls;
  Lane& L = pl->lane_of(stream, ls);
  // samples staged since the last image feed are integrated now (same arithmetic, same order as at the next frame head)
  bool staged = false;
  for (int s = 0; s < L.S; s++) staged = staged || L.h_nimu[s] > 0;
  if (staged) {
    const int rc = lane_flush_imu(ctx, L);
    if (rc != FLVIS_OK) return rc;
  } else {
    hipError_t e = hipStreamSynchronize(L.st);
    if (e != hipSuccess) return ctx->hip_fail(e, "get_imu_states");
  }
  long long seen = 0;
  hipError_t e = hipMemcpy(&seen, reinterpret_cast<const char*>(L.pipe.st + ls) + offsetof(StreamState, imu_seen), sizeof(seen),
                           hipMemcpyDeviceToHost);
  if (e != hipSuccess) return ctx->hip_fail(e, "get_imu_states");
  long long first = L.imu_read[ls];
  int dropped = 0;
  if (seen - first > IMU_OUT_CAP) {  // the ring wrapped since the last fetch: the oldest rows are gone
    dropped = (int)std::min<long long>(seen - first - IMU_OUT_CAP, 0x7fffffff);
    first = seen - IMU_OUT_CAP;
  }
  const long long avail = seen - first;
  const int n = (int)std::min<long long>(avail, cap);
  const double* ring = L.pipe.imu_out + (size_t)ls * IMU_OUT_CAP * 11;
  for (int done = 0; done < n && e == hipSuccess;) {  // at most two pieces (ring wrap)
    const int r0 = (int)((first + done) % IMU_OUT_CAP);
    const int m = std::min(n - done, IMU_OUT_CAP - r0);
    e = hipMemcpy(h_rows11 + (size_t)done * 11, ring + (size_t)r0 * 11, sizeof(double) * 11 * m, hipMemcpyDeviceToHost);
    done += m;
  }
  if (e != hipSuccess) return ctx->hip_fail(e, "get_imu_states");
  L.imu_read[ls] = first + n;  // rows beyond cap stay for the next call
  *n_out = n;
  if (n_dropped) *n_dropped = dropped;
  return FLVIS_OK;
}

// The call-for-call form of F2FTracking::imu_feed(time, acc, gyro, q_w_i&, pos_w_i&, vel_w_i&): the sample is integrated NOW
// (upload + k_imu_feed + read-back, ~0.1 ms) and its state returned, as imu_callback needs it for /imu_pose, /imu_odom and
// /imu_path (vo_tracking.cpp:362-369).
int flvis_imu_feed_out(flvis_ctx* ctx, int stream, double t, const double* acc3, const double* gyro3, double* q_w_i_wxyz,
                       double* pos_w_i, double* vel_w_i) {
  if (!q_w_i_wxyz || !pos_w_i || !vel_w_i) return FLVIS_ERR_INVALID_ARG;
  int rc = flvis_imu_feed(ctx, stream, t, acc3, gyro3);
  if (rc != FLVIS_OK) return rc;
  Pipeline* pl = ctx->pipe;
  hipSetDevice(ctx->device);
  int ls;
  Lane& L = pl->lane_of(stream, ls);
  rc = lane_flush_imu(ctx, L);
  if (rc != FLVIS_OK) return rc;
  long long seen = 0;
  hipError_t e = hipMemcpy(&seen, reinterpret_cast<const char*>(L.pipe.st + ls) + offsetof(StreamState, imu_seen), sizeof(seen),
                           hipMemcpyDeviceToHost);
  double row[11];
  if (e == hipSuccess && seen > 0)
    e = hipMemcpy(row, L.pipe.imu_out + ((size_t)ls * IMU_OUT_CAP + (size_t)((seen - 1) % IMU_OUT_CAP)) * 11, sizeof(row),
                  hipMemcpyDeviceToHost);
  if (e != hipSuccess) return ctx->hip_fail(e, "imu_feed_out");
  if (seen <= 0) return ctx->fail(FLVIS_ERR_HIP, "imu_feed_out: the sample was not integrated");
  memcpy(q_w_i_wxyz, row + 1, 32);
  memcpy(pos_w_i, row + 5, 24);
  memcpy(vel_w_i, row + 8, 24);
  return FLVIS_OK;
}


}  // extern "C"

static void fill_pyr(Pipeline* pl, PyrSel& ps, uint8_t* const* l0, uint8_t* const* l1, const int* cur, int flip, int levels) {
  ps.levels = levels;
  for (int l = 0; l <= levels; l++) {
    ps.lvl[l] = ImgSel{{l0[l], l1 ? l1[l] : l0[l]}, cur, flip, nullptr};
    ps.w[l] = pl->lw[l];
    ps.h[l] = pl->lh[l];
    ps.pitch[l] = pl->lpitch[l];
    ps.stride[l] = pl->lstride[l];
    ps.bx[l] = pl->lbx;
    ps.by[l] = pl->lby;
  }
}

// Levels 1 .. levels of the pyramid `pyr` (level 0 too when `ingest` is set: the copy of the caller's image src0).  Level 0 is read from
// src0 (the caller's image when ingest or in place, else pyr.lvl[0] itself, which another kernel wrote).  The levels' physical borders are
// written by the kernel that produces the level wherever it can; the return value is the mask of the levels whose border is still to
// fill (k_pyr_border).  level0_todo: level 0 has a border that nobody has written yet (equalizeHist / the unaligned copy made it).
// Walking kernels (pyr_walk.hip) where the geometry allows -- 16-pixel lanes: widths that are multiples of 16 up to 1024 --, else the
// LDS-tile kernels level by level.  FLVIS_PYR_PLAN (A/B knob): levels per walking launch, first launch first ("12": default).
static unsigned pyramid_levels(hipStream_t ds, bool bordered, ImgSel src0, int spitch0, size_t sstride0, bool ingest, bool level0_is_buffer,
                               const PyrSel& pyr, int S, const int* active) {
  const int levels = pyr.levels;
  unsigned border_left = bordered ? (1u << (levels + 1)) - 1u : 0u;
  if (!level0_is_buffer && !ingest) border_left &= ~1u;  // (level 0 is the caller's buffer: no border to fill)
  static const char* plan_env = getenv("FLVIS_PYR_PLAN");
  // default: one level, then two ("12": the fastest for 640-pixel rows, profiles/r04_lk_ab.md) -- unless level 1 is not a whole number of
  // 16-pixel lanes while level 0 is (752-pixel rows: 376): then the first launch takes two levels ("21") and only the last one is left
  // to the tile kernels
  const char* plan = plan_env && *plan_env ? plan_env : ((pyr.w[0] & 15) == 0 && (((pyr.w[0] + 1) >> 1) & 15) != 0 ? "21" : "12");
  int step = 0, l = 0;
  while (l < levels) {
    int nout = plan[step] ? plan[step] - '0' : 1;
    if (plan[step]) step++;
    if (nout < 1) nout = 1;
    if (nout > 3) nout = 3;
    if (nout > levels - l) nout = levels - l;
    const bool from0 = l == 0 && (ingest || !level0_is_buffer);
    ImgSel src = from0 ? src0 : pyr.lvl[l];
    const int spitch = from0 ? spitch0 : pyr.pitch[l];
    const size_t sstride = from0 ? sstride0 : pyr.stride[l];
    const bool copy0 = l == 0 && ingest;
    PyrSel q = pyr;
    for (int j = 0; j <= nout; j++)  // a level too small to mirror into its border in one pass keeps it for k_pyr_border
      if (!(bordered && pyr_border_fusable(pyr.w[l + j], pyr.h[l + j], pyr.bx[l + j], pyr.by[l + j]))) q.bx[l + j] = q.by[l + j] = 0;
    if (pyr_walk_ok(pyr.w[l], pyr.h[l], nout, q.bx + l, q.by + l, copy0)) {
      launch_pyr_walk(ds, src, pyr.w[l], pyr.h[l], spitch, sstride, q, l, nout, copy0, S, active);
      for (int j = copy0 ? 0 : 1; j <= nout; j++)
        if (q.bx[l + j]) border_left &= ~(1u << (l + j));
      l += nout;
      continue;
    }
    // one level by the tile kernels
    const int d = l + 1;
    if (copy0) {
      const bool f = q.bx[0] && q.bx[1];
      launch_pyr_down_ingest(ds, src, pyr.w[0], pyr.h[0], spitch, sstride, pyr.lvl[0], pyr.pitch[0], pyr.stride[0], pyr.lvl[1], pyr.pitch[1],
                             pyr.stride[1], S, active, f ? pyr.bx[1] : 0, f ? pyr.by[1] : 0);
      if (f) border_left &= ~3u;
    } else {
      const bool f = q.bx[d] != 0;
      const bool sf = !from0 && bordered && !((border_left >> l) & 1u);  // (the source level's border is complete)
      launch_pyr_down(ds, src, pyr.w[l], pyr.h[l], spitch, sstride, pyr.lvl[d], pyr.pitch[d], pyr.stride[d], S, active, f ? pyr.bx[d] : 0,
                      f ? pyr.by[d] : 0, sf ? pyr.bx[l] : 0, sf ? pyr.by[l] : 0);
      if (f) border_left &= ~(1u << d);
    }
    l = d;
  }
  return border_left;
}

// One local-map launch for the lane (a workgroup per stream takes the stream's next keyframe): on the lane's next local-map HIP stream,
// behind `ev` of the tracking stream (recorded here when `record` is set), its completion event kept for the back-pressure.
static int join_id(const Lane* L, hipEvent_t ev) {
  if (ev == L->ev_img) return 0;
  if (ev == L->ev_det) return 1;
  if (ev == L->ev_gftt) return 2;
  if (ev == L->ev_fe) return 3;
  if (ev == L->ev_lm) return 4;
  if (ev == L->ev_tri) return 5;
  if (ev == L->ev_head) return 6;
  if (ev == L->ev_endf) return 7;
  if (ev == L->ev_tpl) return 8;
  for (int k = 0; k < Lane::BAQ; k++)
    if (ev == L->ev_ba_done[k]) return 9 + k;
  return -1;
}
// "stream s has reached this point" / "stream s goes on when that point has been reached": an event, or (FLVIS_JOIN=flag) a device word
static void join_signal(Pipeline* pl, Lane* L, hipEvent_t ev, hipStream_t s) {
  const int id = pl->flag_joins ? join_id(L, ev) : -1;
  if (id < 0) {
    hipEventRecord(ev, s);
    return;
  }
  launch_store_flag(s, L->d_join + 8 * id, ++L->join_seq[id]);
}
static void join_wait(Pipeline* pl, Lane* L, hipStream_t s, hipEvent_t ev) {
  const int id = pl->flag_joins ? join_id(L, ev) : -1;
  if (id < 0) {
    hipStreamWaitEvent(s, ev, 0);
    return;
  }
  launch_wait_flag(s, L->d_join + 8 * id, 1, L->join_seq[id], L->d_progress + 2);
}

// folded forms (KJoin): the NEXT launch that is handed `kj` stores the word behind its last workgroup / waits for it in front of its first
static bool fold_signal(Pipeline* pl, Lane* L, hipEvent_t ev, KJoin& kj) {
  const int id = pl->fold_joins ? join_id(L, ev) : -1;
  if (id < 0) return false;
  kj.sig = L->d_join + 8 * id;
  kj.sig_seq = ++L->join_seq[id];
  kj.sig_cnt = L->d_join_cnt + id;
  return true;
}
static bool fold_wait(Pipeline* pl, Lane* L, hipEvent_t ev, KJoin& kj, int slot) {
  const int id = pl->fold_joins ? join_id(L, ev) : -1;
  if (id < 0) return false;
  kj.wait[slot] = L->d_join + 8 * id;
  kj.wait_seq[slot] = L->join_seq[id];
  kj.err = L->d_progress + 2;
  return true;
}

// ... or keeps the launch from ENDING before the word is there (KJoin::post): the launch behind it needs no wait of its own
static bool fold_post(Pipeline* pl, Lane* L, hipEvent_t ev, KJoin& kj) {
  const int id = pl->fold_joins ? join_id(L, ev) : -1;
  if (id < 0) return false;
  kj.post = L->d_join + 8 * id;
  kj.post_seq = L->join_seq[id];
  kj.err = L->d_progress + 2;
  return true;
}

static void launch_local_map(Pipeline* pl, Lane* L, hipEvent_t ev, bool record, hipEvent_t* prof_begin_end) {
  const int bi = (L->idx * pl->nba_lane + (int)(L->ba_launches % pl->nba_lane)) % pl->nba;
  hipStream_t bs = pl->ba_stream[bi];
  if (prof_begin_end) {  // (one timed launch per armed step: the first; a second one in the same step keeps its hands off the pair)
    unsigned char& rec = L->prof18_rec[(size_t)pl->prof_step];
    if (rec) prof_begin_end = nullptr;
    else rec = 1;
  }
  if (record) join_signal(pl, L, ev, L->st);
  join_wait(pl, L, bs, ev);
  if (prof_begin_end) hipEventRecord(prof_begin_end[0], bs);
  launch_ba_worker(bs, L->pipe, bi, ++L->ba_tag);
  if (prof_begin_end) hipEventRecord(prof_begin_end[1], bs);
  join_signal(pl, L, L->ev_ba_done[L->ba_launches % Lane::BAQ], bs);
  L->ba_launches++;
  L->ba_pending = false;
}

static void sync_streams(flvis_ctx* ctx) {
  hipStreamSynchronize(ctx->stream);
  Pipeline* pl = ctx->pipe;
  if (!pl) return;
  for (Lane* L : pl->lanes) {
    hipStreamSynchronize(L->st);
    hipStreamSynchronize(L->det_stream);
    if (L->tpl_stream) hipStreamSynchronize(L->tpl_stream);
  }
  for (int k = 0; k < Pipeline::NBA; k++)
    if (pl->ba_stream[k]) hipStreamSynchronize(pl->ba_stream[k]);
}
// waits for everything that was enqueued AND for the local map to have consumed every queued keyframe (a keyframe that
// slipped in while a worker workgroup was releasing its window is picked up by one more worker launch)
static void sync_all(flvis_ctx* ctx) {
  if (ctx->pipe)
    for (Lane* L : ctx->pipe->lanes)
      if (L->ba_pending) launch_local_map(ctx->pipe, L, L->ev_fe, true, nullptr);  // (a launch deferred into the next frame: there is none)
  sync_streams(ctx);
  Pipeline* pl = ctx->pipe;
  if (!pl) return;
  for (Lane* L : pl->lanes) {
    std::vector<unsigned> hd(L->S), tl(L->S);
    for (int guard = 0; guard < 64; guard++) {
      hipMemcpy(hd.data(), L->pipe.kfq_head, sizeof(unsigned) * L->S, hipMemcpyDeviceToHost);
      hipMemcpy(tl.data(), L->pipe.kfq_tail, sizeof(unsigned) * L->S, hipMemcpyDeviceToHost);
      bool pending = false;
      for (int i = 0; i < L->S; i++) pending = pending || hd[i] != tl[i];
      if (!pending) break;
      launch_ba_worker(pl->ba_stream[0], L->pipe, 0, ++L->ba_tag);
      hipStreamSynchronize(pl->ba_stream[0]);
    }
  }
}

// One frame of one lane: stages the lane's host inputs, uploads them as one block and enqueues the fixed kernel sequence
// on the lane's streams.  d_img0 / d_img1 already point at the lane's first stream.
static void lane_frame(flvis_ctx* ctx, Pipeline* pl, Lane* L, const uint8_t* d_img0, const uint8_t* d_img1, const double* h_times,
                       int with_local_map) {
  Pipe& p = L->pipe;
  const int S = L->S;
  hipStream_t st = L->st;
  const int w = pl->cfg.image_width, h = pl->cfg.image_height;
  // ---- stage host inputs (pinned) and upload: [times][imu][input image bases][n_imu]
  const long long frame_no = ++L->frames_uploaded;  // 1-based count of the frames this lane has been fed
  const int pslot = (int)((frame_no - 1) % Lane::PIN_RING);
  uint8_t* pin = (uint8_t*)L->pinned[pslot];
  if (pl->sync_each_frame) hipStreamSynchronize(st);
  // run at most host_lead frames ahead of the GPU (<= PIN_RING: the staging slot of frame N - PIN_RING must be free).  Not further:
  // beyond ~5 queued frames (~300 commands) the HIP runtime itself blocks the enqueuing thread for 7-9 ms at a time and the GPU
  // then runs dry while the queue is refilled (measured)
  // (zero-copy inputs: k_frame_head publishes its progress word when it STARTS, while its other workgroups may still be reading the
  // slot -- one slot of slack, so that the slot the host refills belongs to a frame whose successor has started, i.e. that is over)
  static const bool zc_lead = !(getenv("FLVIS_INPUT_ZEROCOPY") && atoi(getenv("FLVIS_INPUT_ZEROCOPY")) == 0);
  const long long lead = std::min(std::min(pl->host_lead, pl->host_lead_cap), (int)Lane::PIN_RING - (zc_lead ? 1 : 0));
  if (frame_no > lead && *L->h_progress < frame_no - lead) {
    const auto tw = std::chrono::steady_clock::now();
    int polls = 0;
    while (*L->h_progress < frame_no - lead) {
      if ((++polls & 255) == 0 && hipStreamQuery(st) == hipSuccess && *L->h_progress < frame_no - lead) break;  // (idle stream: a failed launch)
      std::this_thread::sleep_for(std::chrono::microseconds(30));
    }
    pl->host_ms_wait += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw).count();
  }
  double* pt = (double*)pin;
  double* pi = pt + S;
  const uint8_t** ptab = (const uint8_t**)(pi + (size_t)S * IMU_MAX * 7);
  int* pn = (int*)(ptab + 2);
  memcpy(pt, h_times, sizeof(double) * S);
  // only the staged samples travel (the block keeps its layout: [S][IMU_MAX][7])
  int max_n = 0;
  for (int s = 0; s < S; s++) max_n = std::max(max_n, L->h_nimu[s]);
  for (int s = 0; s < S; s++)
    memcpy(pi + (size_t)s * IMU_MAX * 7, &L->h_imu[(size_t)s * IMU_MAX * 7], sizeof(double) * 7 * L->h_nimu[s]);
  (void)max_n;
  ptab[0] = d_img0;
  ptab[1] = d_img1;
  memcpy(pn, L->h_nimu.data(), sizeof(int) * S);
  std::fill(L->h_nimu.begin(), L->h_nimu.end(), 0);
  // (round 4, measured: uploading the block on the detection stream into a per-frame device slot -- so that k_frame_head(n + 1) follows
  // k_frame_end(n) without the copy between them -- shortens the gap between two frames by ~19 us and lengthens the chain by ~16 us
  // (1.1914 against 1.1943 ms per step): not kept.  profiles/r04_lk_ab.md)
  // FLVIS_INPUT_ZEROCOPY (default 1): no copy at all -- k_frame_head reads the block where the host staged it (page-locked, mapped into
  // the device's address space: a few KB per stream over PCIe, in parallel over the streams), and the image bases travel as kernel
  // arguments.  A staging slot is rewritten PIN_RING frames later, when the frame that read it is long over (the host-lead wait above).
  static const bool zerocopy_knob = !(getenv("FLVIS_INPUT_ZEROCOPY") && atoi(getenv("FLVIS_INPUT_ZEROCOPY")) == 0);
  const bool zerocopy = zerocopy_knob && L->pinned_dev[pslot] != nullptr;
  L->h_tab[0] = d_img0, L->h_tab[1] = d_img1;
  p.in_img1 = d_img1;
  {
    uint8_t* din = zerocopy ? L->pinned_dev[pslot] : L->d_inputs;
    L->d_time = reinterpret_cast<double*>(din);
    p.imu_in = reinterpret_cast<double*>(din + L->in_off_imu);
    p.n_imu = reinterpret_cast<int*>(din + L->in_off_n);
  }
  if (!zerocopy) hipMemcpyAsync(L->d_inputs, pin, L->input_bytes, hipMemcpyHostToDevice, st);
  // ---- fixed kernel sequence
  const bool prof = pl->prof_cap > 0 && pl->prof_step < pl->prof_cap;
  hipEvent_t* pev = prof ? &L->prof_ev[(size_t)pl->prof_step * (2 * PROF_STAGES)] : nullptr;
#define PB(i, strm) \
  if (prof && ((pl->prof_mask >> (i)) & 1ull)) hipEventRecord(pev[2 * (i)], strm)
#define PE(i, strm) \
  if (prof && ((pl->prof_mask >> (i)) & 1ull)) hipEventRecord(pev[2 * (i) + 1], strm)
  PB(19, st);  // the whole main-stream chain of this frame: per-frame GPU latency (p50/p99 in bench.py)
  // FLVIS_BA_START (round 5): where the local-map launch for the keyframes of frame n is enqueued.  A launch lasts ~1.4 ms, a frame 1.1 ms:
  // for 0.3 ms of every frame TWO launches hold their workgroups' CUs (2 x ~31 of 256), for the rest one.  0: straight behind
  // k_frame_end(n) -- the 0.3 ms are then the head of frame n + 1 and its temporal LK, a kernel that wants every CU.  1 / 2: inside frame
  // n + 1, behind its F-RANSAC / its PnP RANSAC -- the 0.3 ms fall on the one-workgroup-per-stream kernels of the geometry chain.
  // Measured (one box, two runs each): 57.80k / 57.86k frames/s with 0, 58.03k / 58.21k with 1 (the 0.3 ms then fall on the corner response,
  // which is on the critical path of the detection stream), 58.74k / 58.58k with 2 (default; temporal LK 0.212 -> 0.196 ms, frame chain
  // p50 1.065 -> 1.048 ms).  Only between the steps of one flvis_run_steps call -- the next frame is known to follow at once; a
  // per-frame caller (the ROS wrapper) gets its local-map launch when its frame ends, as before, and so does the last step of a batch.
  static const int ba_start = getenv("FLVIS_BA_START") ? atoi(getenv("FLVIS_BA_START")) : 2;
  hipEvent_t* prof18 = (prof && ((pl->prof_mask >> 18) & 1ull)) ? &pev[2 * 18] : nullptr;
  const bool skipped = pl->frames_fed < (long long)pl->cfg.skip_first_n_imgs;
  const bool depth_cam = pl->cfg.cam_type == CAM_DEPTH;  // the second image is the Z16 depth map, read in place
  const bool eq = pl->cfg.need_equal_hist != 0;
  // rows of whole 16-byte lanes at 16-byte aligned bases (the walking kernels' dwordx4 loads); otherwise (KITTI: 1241 x 376 tightly packed
  // rows, or a caller's buffer at an odd offset) both images are copied into pitch-aligned level 0
  const bool aligned = (w & 15) == 0 && !(((uintptr_t)d_img0 | (depth_cam ? (uintptr_t)0 : (uintptr_t)d_img1)) & 15);
  hipStream_t ds = L->det_stream;
  // FLVIS_HEAD_STREAM=1 (round 5, A/B knob): the two halves of the frame's head change streams -- the left pyramid on the MAIN stream,
  // straight behind k_frame_end of the previous frame, k_frame_head on the detection stream; the temporal LK then waits for one event
  // (the head's) instead of following the head and waiting for the pyramid's.  Only with the detection behind the F-RANSAC (modes >= 2).
  // Measured (one box, two runs each): 57.1k / 57.3k frames/s against 57.3k / 57.4k; the temporal LK starts 75 us after the head either
  // way -- the pyramid's two launches take 46 + 22 us beside the local map's workgroups (19 + 12 us alone), and they are what it waits for
  static const bool head_stream_knob = getenv("FLVIS_HEAD_STREAM") && atoi(getenv("FLVIS_HEAD_STREAM")) == 1 &&
                                       !(getenv("FLVIS_DET_START") && atoi(getenv("FLVIS_DET_START")) < 2) &&
                                       !(getenv("FLVIS_DET_ORDER") && atoi(getenv("FLVIS_DET_ORDER")) == 0);
  const bool head_on_det = head_stream_knob && !skipped;
  hipStream_t s_img = head_on_det ? st : ds, s_head = head_on_det ? ds : st;
  ImgSel in0 = img_plain(d_img0), in1 = img_plain(d_img1);  // (kernel arguments: no graph is captured)
  ImgSel l0cur{{L->pyr0[0][0], L->pyr0[1][0]}, p.img_slot, 0, nullptr};
  if (!skipped) {
    // left image -> level 0 of the slot this frame is going to use (+ the pyramid), on the detection stream BESIDE k_frame_head: the
    // slot was fixed when the previous frame ended (img_slot_in), so the image work does not wait for the IMU integration and the
    // state machine (one thread per stream, 45 us).  Every stream's image is ingested, also that of a stream this frame leaves idle.
    // Without equalizeHist the first pyrDown reads the caller's image and writes level 0 and level 1 in one pass; with it the
    // equalised image is level 0.
    ImgSel l0in{{L->pyr0[0][0], L->pyr0[1][0]}, p.img_slot_in, 0, nullptr};
    // what the detection stream needs from the main one here is k_frame_end of the previous frame (the slot the image goes to) and, in
    // the copying mode, the upload of the input table: a signal of its own -- or, folded, the word the last k_frame_end stored itself
    // (only where nothing can have been enqueued on the main stream since that k_frame_end: between the steps of one flvis_run_steps call,
    // and for host images that arrive through the copy stream -- a caller of flvis_image_feed may have produced its images on this stream)
    if (pl->fold_joins && zerocopy && L->endf_valid && pl->chain_continuous) {
      join_wait(pl, L, ds, L->ev_endf);
    } else {
      join_signal(pl, L, L->ev_lm, st);  // (the frame's input table has been uploaded)
      join_wait(pl, L, ds, L->ev_lm);
    }
    // host images (flvis_image_feed_host, FLVIS_H2D_WAIT=1): the upload's event is waited for by the stream that ingests the left image; the
    // main stream only sees the joins it has anyway (left pyramid in front of the temporal LK, right pyramid in front of the stereo LK)
    if (pl->up_event) hipStreamWaitEvent(ds, pl->up_event, 0);
    if (pl->up_flag) launch_wait_flag(ds, pl->up_flag, Pipeline::HostFeed::FLAG_WORDS, pl->up_seq, L->d_progress + 2);
    PB(1, s_img);
    if (eq) launch_equalize_hist(s_img, in0, l0in, w, h, w, pl->lpitch[0], (size_t)w * h, pl->lstride[0], S, L->eq_hist, L->eq_lut, nullptr);
    else if (!aligned) launch_copy_image_any(s_img, in0, l0in, w, h, w, pl->lpitch[0], (size_t)w * h, pl->lstride[0], S, nullptr);
    PE(1, s_img);
    PB(2, s_img);
    // the borders of the levels: written by the pyrDown kernel that produces the level (every pixel also goes to the border positions
    // that mirror it); k_pyr_border only for what is left (level 0 when another kernel makes it, levels smaller than the border)
    PyrSel pyl;
    fill_pyr(pl, pyl, L->pyr0[0], L->pyr0[1], p.img_slot_in, 0, pl->levels);
    unsigned border_left = pyramid_levels(s_img, pl->lbx != 0, in0, w, (size_t)w * h, !eq && aligned, true, pyl, S, nullptr);
    if (pl->levels == 0 && !eq && aligned) launch_copy_image(s_img, in0, l0in, w, h, w, pl->lpitch[0], (size_t)w * h, pl->lstride[0], S, nullptr);
    if (border_left) launch_pyr_border(s_img, pyl, S, nullptr, border_left);
    PE(2, s_img);
    if (!head_on_det) join_signal(pl, L, L->ev_img, ds);
  }
  PB(0, s_head);
  // the staged IMU samples, then frame_begin -- and, unless the local-map feedback has to be applied in between, the temporal tracker's
  // inputs in the same launch (FLVIS_HEAD_PREPARE=0, A/B knob: two launches)
  static const bool head_prepare_knob = !(getenv("FLVIS_HEAD_PREPARE") && atoi(getenv("FLVIS_HEAD_PREPARE")) == 0);
  const bool head_prepare = head_prepare_knob && !pl->feedback_used && !skipped;
  const bool head_signals = !pl->feedback_used && fold_signal(pl, L, L->ev_head, p.kj);  // (the head kernel is the last one in front of the signal)
  // ... and it does not end before the left pyramid is there (the temporal LK follows it, directly or behind k_track_prepare)
  const bool head_posts = !skipped && !head_on_det && fold_post(pl, L, L->ev_img, p.kj);
  if (head_prepare) launch_frame_head_prepare(s_head, p, L->d_time, L->d_progress, frame_no);
  else launch_frame_head(s_head, p, L->d_time, L->d_progress, frame_no);
  p.kj = KJoin{};
  if (pl->feedback_used) launch_apply_correction(s_head, p);  // STEP1 of the Tracking case (local-map feedback, opt-in)
  PE(0, s_head);
  // the detection stream's kernels that read what k_frame_head decides (act_img, gftt_act, gftt_maxc, img_slot) wait for this event:
  // the corner detection and the right pyramid in every FLVIS_DET_START mode (in the default mode they start behind the F-RANSAC anyway)
  if (!head_signals) join_signal(pl, L, L->ev_head, s_head);
  if (head_on_det) join_wait(pl, L, st, L->ev_head);  // join: the head (the left pyramid is on this stream)
  if (skipped) {
    // the reference drops the first skip_first_n_imgs frames before any processing (vo_tracking.cpp image callback): every
    // stream is idle for this frame, so only the IMU filter, the frame counter and the per-frame outputs are advanced
    PB(17, st);
    L->endf_valid = fold_signal(pl, L, L->ev_endf, p.kj);
    launch_frame_end(st, p);
    p.kj = KJoin{};
    PE(17, st);
    PE(19, st);
    if (prof)
      for (int i = 1; i < PROF_STAGES; i++)
        if (i != 17 && i != 19) {
          PB(i, st);
          PE(i, st);
        }
    return;
  }
  // lanes out of phase: a lane's first processed frame starts when the previous lane has finished the temporal LK of its own, so
  // that from then on the image kernels of one lane overlap the one-workgroup-per-stream geometry chain of another
  const bool first_processed = pl->frames_fed == (long long)pl->cfg.skip_first_n_imgs;
  if (first_processed && pl->stagger && L->idx > 0) hipStreamWaitEvent(st, pl->lanes[L->idx - 1]->ev_stagger, 0);
  // (the guesses of the temporal tracker only need the state frame_begin left)
  PB(3, st);
  if (!head_prepare) launch_track_prepare(st, p);
  PE(3, st);
  if (!head_on_det && !head_posts) join_wait(pl, L, st, L->ev_img);  // join: the left pyramid
  // fork: the right pyramid (first used by the stereo matcher) and the corner detection of the new left image (speculative
  // for tracking frames: used only if tracking succeeds) run beside the temporal tracking chain.  The right image is only
  // read within this frame, so without equalizeHist the caller's buffer IS level 0 of the right pyramid (no copy).
  // The right image is only read within this frame: without equalizeHist the caller's buffer IS level 0 of the right pyramid (no copy;
  // that level then has no border, and the few stereo search regions that leave the image at level 0 are staged by the kernel's
  // index-reflecting path).  FLVIS_RIGHT_COPY=1 (A/B knob): a bordered copy instead, written by the first pyrDown.
  static const bool right_copy = getenv("FLVIS_RIGHT_COPY") && atoi(getenv("FLVIS_RIGHT_COPY")) != 0;
  const bool r_in_place = !eq && aligned && !(pl->lbx && right_copy);
  ImgSel r0 = r_in_place ? in1 : img_plain(L->pyr1[0]);
  const int r0pitch = r_in_place ? w : pl->lpitch[0];
  const size_t r0stride = r_in_place ? (size_t)w * h : pl->lstride[0];
  // order on the detection stream: the corner response first (FeatureDEM waits for it at the join; it then overlaps the
  // temporal LK instead of the one-workgroup-per-stream RANSAC kernels it would slow down), the right pyramid after it (only
  // the stereo matcher needs it, much later).  FLVIS_DET_ORDER=0 restores the round-1 order (A/B knob).
  static const bool gftt_first = !(getenv("FLVIS_DET_ORDER") && atoi(getenv("FLVIS_DET_ORDER")) == 0);
  // FLVIS_DET_START (A/B knob) = where the detection stream's work starts: 0 beside the temporal LK (rounds 1-2), 1 when the LK has
  // finished, 2 when the F-RANSAC has finished, 3 (default) like 2 with the right pyramid behind the corner detection.  Measured in one
  // session (64 streams, local map on): 0: 1.643 ms/step (the LK is stretched from 0.40 to 0.45 ms by the corner response and the
  // pyramid kernels), 1: 1.721 (k_ransac_f, 1024 threads per stream, goes from 0.07 to 0.18 ms under k_eig_walk), 2: 1.622, 3: 1.602:
  // k_ransac_pnp / k_pose_lm / k_reproj_filter are latency chains on 64 CUs and leave the rest of the chip to the detection.
  static const int gftt_after_lk = getenv("FLVIS_DET_START") ? atoi(getenv("FLVIS_DET_START")) : 3;
  auto detect_corners = [&] {
    if (!head_on_det) join_wait(pl, L, ds, L->ev_head);
    launch_gftt(ds, l0cur, w, h, pl->lpitch[0], pl->lstride[0], S, L->gftt, nullptr, p.cam.gftt_ql, p.gftt_maxc, p.cam.gftt_num,
                (double)p.cam.gftt_dis, L->gftt_xy, L->gftt_n, 2 * p.cam.gftt_num, p.gftt_act,
                (prof && ((pl->prof_mask >> 10) & 7ull) == 7ull) ? &pev[2 * 10] : nullptr, false);
    // FeatureDEM's image part (regions, Harris scores, per-region order of the corners) follows at once, off the critical path
    KJoin prep_kj{};
    const bool prep_signals = fold_signal(pl, L, L->ev_gftt, prep_kj);
    launch_feature_dem_prep(ds, l0cur, w, h, pl->lpitch[0], pl->lstride[0], S, p.cam.dem, L->gftt_xy, L->gftt_n, 2 * p.cam.gftt_num,
                            p.gftt_act, L->dem_sorted, L->dem_roff, &prep_kj);
    if (!prep_signals) join_signal(pl, L, L->ev_gftt, ds);
  };
  if (gftt_first && !gftt_after_lk) detect_corners();
  auto right_pyramid_on = [&](hipStream_t ds, bool on_main) {  // (ds: the stream it runs on -- the detection stream, or the main one)
    if (!on_main && !head_on_det) join_wait(pl, L, ds, L->ev_head);
    if (!depth_cam) {
      if (eq) launch_equalize_hist(ds, in1, img_plain(L->pyr1[0]), w, h, w, pl->lpitch[0], (size_t)w * h, pl->lstride[0], S, L->eq_hist, L->eq_lut, p.act_img);
      else if (!aligned) launch_copy_image_any(ds, in1, img_plain(L->pyr1[0]), w, h, w, pl->lpitch[0], (size_t)w * h, pl->lstride[0], S, p.act_img);
      const bool ingest = !r_in_place && !eq && aligned;  // level 0 = a copy of the caller's image, written by the first pyrDown
      PyrSel pyr_r;
      fill_pyr(pl, pyr_r, L->pyr1, nullptr, nullptr, 0, pl->levels);
      unsigned border_left = pyramid_levels(ds, pl->lbx != 0, in1, w, (size_t)w * h, ingest, !r_in_place, pyr_r, S, p.act_img);
      if (pl->levels == 0 && ingest)
        launch_copy_image(ds, in1, img_plain(L->pyr1[0]), w, h, w, pl->lpitch[0], (size_t)w * h, pl->lstride[0], S, p.act_img);
      if (border_left) launch_pyr_border(ds, pyr_r, S, p.act_img, border_left);
    }
    if (!gftt_first) {
      launch_gftt(ds, l0cur, w, h, pl->lpitch[0], pl->lstride[0], S, L->gftt, nullptr, p.cam.gftt_ql, p.gftt_maxc, p.cam.gftt_num,
                  (double)p.cam.gftt_dis, L->gftt_xy, L->gftt_n, 2 * p.cam.gftt_num, p.gftt_act,
                  (prof && ((pl->prof_mask >> 10) & 7ull) == 7ull) ? &pev[2 * 10] : nullptr, false);
      launch_feature_dem_prep(ds, l0cur, w, h, pl->lpitch[0], pl->lstride[0], S, p.cam.dem, L->gftt_xy, L->gftt_n, 2 * p.cam.gftt_num,
                              p.gftt_act, L->dem_sorted, L->dem_roff);
    }
    if (!on_main) join_signal(pl, L, L->ev_det, ds);
  };
  auto right_pyramid = [&] { right_pyramid_on(ds, false); };
  // 5 (round 5, A/B knob): the corners as in 3; the right pyramid -- two light launches, 36 us -- on the MAIN stream behind the reprojection
  // filter, where that stream waits ~50 us for the corner detection anyway, and no join in front of the stereo LK.  Measured (two runs
  // each, one box): 56.2k / 56.4k frames/s against 56.6k / 56.7k for 3 -- the stereo LK stage does not get shorter: not adopted
  const bool pyramid_main = gftt_first && gftt_after_lk == 5;
  const bool pyramid_late = gftt_first && gftt_after_lk == 3;  // 3: the right pyramid waits for the F-RANSAC too
  // 4 (round 4): like 3 for the corners, but the right pyramid -- a light, memory-bound pass that only the stereo matcher needs -- runs
  // when the temporal LK has finished, beside the F-RANSAC, instead of behind the corner detection where the stereo LK waited for it
  const bool pyramid_mid = gftt_first && gftt_after_lk == 4;
  if (!pyramid_late && !pyramid_mid && !pyramid_main) right_pyramid();
  // temporal tracking
  PB(4, st);
  {
    PyrSel prev, next;
    fill_pyr(pl, prev, L->pyr0[0], L->pyr0[1], p.img_slot, 1, pl->levels_t);
    fill_pyr(pl, next, L->pyr0[0], L->pyr0[1], p.img_slot, 0, pl->levels_t);
    LKParams prm{30, 1e-3 * 1e-3, 1e-4f, 1};
    if (pl->lk_stats) prm.stats = reinterpret_cast<unsigned long long*>(p.counters) + 36;  // temporal: counters[36 .. 47]
    if (pl->lk_stats) prm.stats_tc = reinterpret_cast<unsigned long long*>(p.counters) + 61;  // both launches: counters[61 .. 63]
    if (L->tc) {  // templates the stereo matcher of the previous frame stored (same image, same pixel): taken instead of computed
      prm.tc = L->tc;
      prm.tc_mode = 2;
      prm.tc_cap = p.tc_cap;
      prm.tc_stride = L->tc_stride;
      prm.tc_slot = p.lk_slot;
      prm.tc_tag = p.lk_tag;
    }
    prm.dbg_slot = (int)(pl->frames_fed & 7);
    prm.order = pl->lk_order & 1;
    launch_lk_track(st, prev, next, p.prev_pts, p.next_pts, p.lk_status, p.lk_count, NMAX, S, prm, p.act_track, pl->max_pts, 1);
  }
  PE(4, st);
  if (gftt_first && gftt_after_lk == 1) {
    join_signal(pl, L, L->ev_lm, st);
    join_wait(pl, L, ds, L->ev_lm);
    detect_corners();
  }
  if (pyramid_mid) {
    join_signal(pl, L, L->ev_lm, st);
    join_wait(pl, L, ds, L->ev_lm);
    right_pyramid();
  }
  // templates ahead of the stereo matcher (lane_create: FLVIS_TPL_AHEAD): the survivors' pixels are known when k_track_collect has run.
  // FLVIS_TPL_START (A/B knob): 1 (default) the kernel starts with the corner detection, behind the F-RANSAC, on that join's word;
  // 0 behind k_track_collect, beside the F-RANSAC, on a word of its own (one more one-lane launch on the chain)
  const int tpl_start = pl->tpl_start;
  auto templates_ahead = [&] {
    hipStream_t ts = L->tpl_stream;
    join_wait(pl, L, ts, tpl_start == 1 ? L->ev_lm : L->ev_tpl);
    PyrSel img;
    fill_pyr(pl, img, L->pyr0[0], L->pyr0[1], p.img_slot, 0, pl->levels_s);
    launch_lk_templates_ahead(ts, img, p.tpl_pts, p.tpl_count, NMAX, S, L->tc, p.tc_cap, L->tc_stride, p.tpl_tag, pl->max_pts);
    join_signal(pl, L, L->ev_tpl, ts);
  };
  // FLVIS_CHAIN_MERGE (round 6; bits, default 3): launches of the frame's chain folded into their neighbours -- every launch on the chain is
  // ~8 us of dispatch, ramp and drain whatever it computes.  Bit 0: k_add_new + k_depth_seeds as one launch (a barrier between them);
  // bit 1: k_track_collect as the prologue of k_ransac_f (and by sixteen waves instead of one).  0: rounds 1-5's launches.
  const int chain_merge = pl->chain_merge;
  const bool merge_collect = (chain_merge & 2) && !(p.tpl_ahead && tpl_start == 0);
  PB(5, st);
  if (!merge_collect) launch_track_collect(st, p);
  PE(5, st);
  if (p.tpl_ahead && tpl_start == 0) {
    join_signal(pl, L, L->ev_tpl, st);
    templates_ahead();
  }
  if (first_processed && pl->lanes.size() > 1) hipEventRecord(L->ev_stagger, st);
  PB(6, st);
  const bool rf_signals = gftt_first && gftt_after_lk >= 2 && fold_signal(pl, L, L->ev_lm, p.kj);
  launch_ransac_f(st, p, merge_collect);
  p.kj = KJoin{};
  PE(6, st);
  if (gftt_first && gftt_after_lk >= 2) {
    if (!rf_signals) join_signal(pl, L, L->ev_lm, st);
    if (p.tpl_ahead && tpl_start == 1) templates_ahead();
    join_wait(pl, L, ds, L->ev_lm);
    detect_corners();
    if (pyramid_late) right_pyramid();
    if (ba_start == 1 && L->ba_pending) launch_local_map(pl, L, L->ev_lm, false, prof18);
  }
  PB(7, st);
  const bool ba_here = ba_start == 2 && L->ba_pending;
  const bool pnp_signals = ba_here && fold_signal(pl, L, L->ev_fe, p.kj);  // (the deferred local-map launch starts behind this kernel)
  launch_ransac_pnp(st, p);
  p.kj = KJoin{};
  if (p.pnp_tail_cv) launch_pnp_tail_cv(st, p);
  PE(7, st);
  if (ba_here) launch_local_map(pl, L, L->ev_fe, !pnp_signals, prof18);
  PB(8, st);
  launch_pose_lm(st, p);  // (with k_track_post's work in its prologue)
  PE(8, st);
  PB(9, st);
  const bool rp_signals = !pyramid_main && fold_signal(pl, L, L->ev_lm, p.kj);
  launch_reproj_filter(st, p);
  p.kj = KJoin{};
  PE(9, st);
  if (pyramid_main) right_pyramid_on(st, true);
  // the IMU filter's correction from this frame's pose: on the detection stream (joined with the triangulation before the depth innovation)
  if (!rp_signals) join_signal(pl, L, L->ev_lm, st);
  join_wait(pl, L, ds, L->ev_lm);
  launch_vi_correction(ds, p);
  // join: FeatureDEM (init: detect, tracking: redetect) consumes the corners; the right pyramid is joined before the stereo LK
  KJoin dem_kj{};
  if (!fold_wait(pl, L, gftt_first ? L->ev_gftt : L->ev_det, dem_kj, 0)) join_wait(pl, L, st, gftt_first ? L->ev_gftt : L->ev_det);
  PB(13, st);
  launch_feature_dem(st, w, h, S, p.cam.dem, L->dem_sorted, L->dem_roff, 2 * p.cam.gftt_num, p.det_mode, p.exist_xy, p.n_exist, NMAX,
                     p.new_xy, p.n_new, NEW_MAX, &dem_kj);
  // k_add_new (one wave per stream) stores the word the two-view triangulation waits for: that kernel reads the landmarks as k_add_new leaves
  // them and nothing k_depth_seeds writes
  const bool an_signals = fold_signal(pl, L, L->ev_lm, p.kj);
  const bool merge_seeds = (chain_merge & 1) != 0;  // (k_add_new_seeds carries both launches' joins)
  if (!merge_seeds) {
    launch_add_new(st, p);
    p.kj = KJoin{};
  }
  PE(13, st);
  // depth innovation: stereo LK img0 -> img1 + DLT + IIR
  PB(14, st);
  // (k_depth_seeds and k_depth_triangulate store no word themselves: 256 workgroups each -- every workgroup's release is a write-back of its
  // XCD's L2, and beside the stereo LK's template stores the kernels doubled their time: 5.9 -> 12.7 us, 87 -> 180 us)
  const bool ds_signals = an_signals;
  // ... and k_depth_seeds does not end before the right pyramid is there (the stereo LK follows it)
  const bool seeds_post = gftt_first && !pyramid_main && fold_post(pl, L, L->ev_det, p.kj);
  // ... and with templates ahead it looks them up: they must be there
  if (p.tpl_ahead && !fold_wait(pl, L, L->ev_tpl, p.kj, 0)) join_wait(pl, L, st, L->ev_tpl);
  if (merge_seeds) launch_add_new_seeds(st, p);
  else launch_depth_seeds(st, p);
  p.kj = KJoin{};
  PE(14, st);
  // the two-view triangulation that k_depth_innovate consumes: on the detection stream (idle by now), under the stereo LK
  if (!ds_signals) join_signal(pl, L, L->ev_lm, st);
  join_wait(pl, L, ds, L->ev_lm);
  const bool tri_signals = false;
  launch_depth_triangulate(ds, p);
  if (!tri_signals) join_signal(pl, L, L->ev_tri, ds);
  if (gftt_first && !pyramid_main && !seeds_post) join_wait(pl, L, st, L->ev_det);
  PB(15, st);
  if (!depth_cam) {
    PyrSel prev, next;
    fill_pyr(pl, prev, L->pyr0[0], L->pyr0[1], p.img_slot, 0, pl->levels_s);
    fill_pyr(pl, next, L->pyr1, nullptr, nullptr, 0, pl->levels_s);
    next.lvl[0] = r0;
    next.pitch[0] = r0pitch;
    next.stride[0] = r0stride;
    if (r_in_place) next.bx[0] = next.by[0] = 0;  // the caller's buffer has no border
    LKParams prm{30, 1e-3 * 1e-3, 1e-4f, 1};
    if (pl->lk_stats) prm.stats = reinterpret_cast<unsigned long long*>(p.counters) + 48;  // stereo: counters[48 .. 59]
    if (pl->lk_stats) prm.stats_tc = reinterpret_cast<unsigned long long*>(p.counters) + 61;
    if (L->tc) {  // ... which stores its templates for the next frame's temporal tracker
      prm.tc = L->tc;
      prm.tc_mode = p.tpl_ahead ? 3 : 1;  // (3: takes those k_lk_templates_ahead made, k_depth_seeds has looked the slots up)
      prm.tc_cap = p.tc_cap;
      prm.tc_stride = L->tc_stride;
      prm.tc_tag = p.lk_tag;
      prm.tc_slot = p.lk_slot;
    }
    prm.dbg_slot = (int)(pl->frames_fed & 7);
    prm.order = (pl->lk_order >> 1) & 1;
    launch_lk_track(st, prev, next, p.prev_pts, p.next_pts, p.lk_status, p.lk_count, NMAX, S, prm, p.det_mode, pl->max_pts,
                    L->tc && p.tpl_ahead ? 4 : 2);
  }
  PE(15, st);
  const bool inn_waits = fold_wait(pl, L, L->ev_tri, p.kj, 0);
  if (!inn_waits) join_wait(pl, L, st, L->ev_tri);
  PB(16, st);
  launch_depth_innovate(st, p);
  p.kj = KJoin{};
  PE(16, st);
  if (with_local_map) {
    // Keyframe-queue back-pressure, expressed in stream order (never by spinning inside a kernel).  Once every local-map launch of
    // this lane up to index j has finished, less than ba_backlog = KFQ / 2 keyframes of the frames up to j * ba_every are left in any
    // queue: the last workgroup that owned the stream's window among those launches stayed until fewer were waiting (k_ba_worker),
    // and every launch behind it found the queue empty.  The frames since then add at most (D + nba_lane) * ba_every keyframes, so
    // waiting for the launches of D frames ago (the last one on each of the lane's local-map streams) keeps a queue at
    // KFQ / 2 - 1 + (D + 2) * ba_every < KFQ when k_frame_end appends: no keyframe is dropped, whatever the optimiser's pace.
    // (ba_every <= KFQ / 4, flvis_tracker_create.)
    const long long D = std::max(0, KFQ / 2 / pl->ba_every - 2 - (ba_start != 0 ? 1 : 0));  // (a deferred launch is one frame late)
    for (int k = 0; k < pl->nba_lane; k++) {
      const long long j = L->ba_launches - 1 - D - k;
      if (j >= 0 && L->ba_launches - j <= Lane::BAQ) {
        if (!(k < 2 && fold_wait(pl, L, L->ev_ba_done[j % Lane::BAQ], p.kj, k))) join_wait(pl, L, st, L->ev_ba_done[j % Lane::BAQ]);
      }
    }
  }
  PB(17, st);
  L->endf_valid = fold_signal(pl, L, L->ev_endf, p.kj);
  launch_frame_end(st, p);
  p.kj = KJoin{};
  PE(17, st);
  PE(19, st);
  if (with_local_map && (pl->frames_fed % pl->ba_every) == 0) {
    if (L->ba_pending) launch_local_map(pl, L, L->ev_fe, true, nullptr);  // (a deferred launch this frame had no place for: skipped frames)
    if (ba_start != 0 && pl->defer_ba) L->ba_pending = true;
    else if (L->endf_valid) launch_local_map(pl, L, L->ev_endf, false, prof18);  // (k_frame_end has stored the word itself)
    else launch_local_map(pl, L, L->ev_fe, true, prof18);
  } else if (!with_local_map) {
    // without a local map nobody consumes the keyframe queue: drop what frame_end appended
    hipMemcpyAsync(p.kfq_head, p.kfq_tail, sizeof(unsigned) * S, hipMemcpyDeviceToDevice, st);
    if (prof) {
      PB(18, st);
      PE(18, st);
      L->prof18_rec[(size_t)pl->prof_step] = 1;
    }
  }
#undef PB
#undef PE
}

extern "C" {

int flvis_image_feed(flvis_ctx* ctx, const uint8_t* d_img0, const uint8_t* d_img1, const double* h_times,
                     flvis_frame_out* h_out, int with_local_map) {
  if (!ctx || !ctx->pipe || !d_img0 || !d_img1 || !h_times) return FLVIS_ERR_INVALID_ARG;
  Pipeline* pl = ctx->pipe;
  hipSetDevice(ctx->device);
  const size_t img_px = (size_t)pl->cfg.image_width * pl->cfg.image_height;
  const size_t img1_bytes = img_px * (pl->cfg.cam_type == CAM_DEPTH ? 2 : 1);
  const bool multi = pl->lanes.size() > 1;
  const auto t_host0 = std::chrono::steady_clock::now();
  // the caller's images were produced on the context's stream; lanes with their own streams wait for them, and the
  // context's stream waits for the lanes afterwards, so that the caller may recycle its buffers in stream order -- the buffers
  // of THIS frame by default, those handed over input_hold frames ago when the caller cycles through more buffers
  // (flvis_set_input_hold): only then can the lanes drift apart by more than a frame
  if (multi) hipEventRecord(pl->ev_in, ctx->stream);
  for (Lane* L : pl->lanes) {
    if (multi) hipStreamWaitEvent(L->st, pl->ev_in, 0);
    lane_frame(ctx, pl, L, d_img0 + (size_t)L->s0 * img_px, d_img1 + (size_t)L->s0 * img1_bytes, h_times + L->s0, with_local_map);
    if (multi) {
      hipEventRecord(L->ev_end[pl->frames_fed % Lane::HOLD_RING], L->st);
      const long long rel = pl->frames_fed - pl->input_hold;
      if (rel >= 0) hipStreamWaitEvent(ctx->stream, L->ev_end[rel % Lane::HOLD_RING], 0);
    }
  }
  pl->host_ms_total += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_host0).count();
  const bool prof = pl->prof_cap > 0 && pl->prof_step < pl->prof_cap;
  if (prof) pl->prof_step++;
  pl->frames_fed++;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ctx->hip_fail(e, "image_feed launch");
  if (h_out) {
    static_assert(sizeof(flvis_frame_out) == sizeof(FrameOut), "FrameOut layout");
    for (Lane* L : pl->lanes) {
      e = hipMemcpyAsync(h_out + L->s0, L->pipe.out, sizeof(FrameOut) * L->S, hipMemcpyDeviceToHost, L->st);
      if (e != hipSuccess) return ctx->hip_fail(e, "image_feed readback");
    }
    for (Lane* L : pl->lanes) {
      e = hipStreamSynchronize(L->st);
      if (e != hipSuccess) return ctx->hip_fail(e, "image_feed readback");
    }
  }
  return FLVIS_OK;
}

// TrackingNodeletClass::image_input_callback (src/frontend/vo_tracking.cpp:396-430) hands F2FTracking::image_feed two HOST
// cv::Mat (mono8, or 3/4 channels converted by cvtColor, f2f_tracking.cpp:74-111; depth rigs: img1 = 16UC1 depth): the same
// hand-over with plain structs.  Uploads run on a copy stream into double-buffered device staging.
int flvis_image_feed_host(flvis_ctx* ctx, const flvis_image* h_img0, const flvis_image* h_img1, flvis_frame_out* h_out,
                          int with_local_map, int hold_buffers) {
  if (!ctx || !ctx->pipe || !h_img0 || !h_img1) return FLVIS_ERR_INVALID_ARG;
  Pipeline* pl = ctx->pipe;
  const int S = pl->S, w = pl->cfg.image_width, h = pl->cfg.image_height;
  const bool depth_cam = pl->cfg.cam_type == CAM_DEPTH;
  const int ch0 = h_img0[0].channels, ch1 = h_img1[0].channels;
  if ((ch0 != 1 && ch0 != 3 && ch0 != 4) || (ch1 != 1 && ch1 != 3 && ch1 != 4) || (depth_cam && ch1 != 1))
    return ctx->fail(FLVIS_ERR_INVALID_ARG, "image_feed_host: channels must be 1, 3 or 4 (depth image: 1)");
  const size_t bpp[2] = {(size_t)ch0, depth_cam ? (size_t)2 : (size_t)ch1};  // bytes per pixel as handed over
  for (int s = 0; s < S; s++) {
    const flvis_image* im[2] = {&h_img0[s], &h_img1[s]};
    for (int c = 0; c < 2; c++)
      if (!im[c]->data || im[c]->width != w || im[c]->height != h || im[c]->channels != (c ? ch1 : ch0) ||
          (size_t)im[c]->pitch < (size_t)w * bpp[c])
        return ctx->fail(FLVIS_ERR_INVALID_ARG, "image_feed_host: image size/channels/pitch do not match the configuration");
  }
  hipSetDevice(ctx->device);
  Pipeline::HostFeed& hf = pl->hf;
  if (!hf.strm) {
    bool ok = true;
    const int qpad = getenv("FLVIS_H2D_QPAD") ? std::max(0, std::min(atoi(getenv("FLVIS_H2D_QPAD")), 4)) : 0;
    for (int k = 0; k < qpad && ok; k++) {
      ok = hipStreamCreateWithFlags(&hf.pad_strm[k], hipStreamNonBlocking) == hipSuccess;
      if (ok) launch_store_progress(hf.pad_strm[k], ctx->pipe->lanes[0]->d_progress + 1, 0);  // (a scratch word: the stream gets its hardware queue)
    }
    ok = ok && hipStreamCreateWithFlags(&hf.strm, hipStreamNonBlocking) == hipSuccess;
    for (int k = 0; k < 2 && ok; k++)
      ok = hipEventCreateWithFlags(&hf.ev_done[k], pl->ev_flags) == hipSuccess &&
           hipEventCreateWithFlags(&hf.ev_free[k], pl->ev_flags) == hipSuccess && hipEventCreate(&hf.ev_t0[k]) == hipSuccess &&
           hipEventCreate(&hf.ev_t1[k]) == hipSuccess;
    if (!ok) return ctx->fail(FLVIS_ERR_HIP, "image_feed_host: cannot create the copy stream");
    void* hp = nullptr;
    void* dp = nullptr;
    if (hipHostMalloc(&hp, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess || hipHostGetDevicePointer(&dp, hp, 0) != hipSuccess)
      return ctx->fail(FLVIS_ERR_HIP, "image_feed_host: cannot map the progress word");
    hf.h_up = (volatile long long*)hp;
    hf.d_up = (long long*)dp;
    *hf.h_up = 0;
    hf.times.assign(S, 0.0);
    if (const char* e = getenv("FLVIS_H2D_MODE")) hf.mode = atoi(e) == 1 ? 1 : 2;
    if (hf.mode == 2) {
      // (fine-grained device memory: the copy engine writes it past the XCDs' L2 caches, the waiting wave must never see a cached line)
      void* dfp = nullptr;
      bool fok = hipExtMallocWithFlags(&dfp, sizeof(long long) * Pipeline::HostFeed::FLAG_WORDS, hipDeviceMallocFinegrained) == hipSuccess;
      if (fok) {
        hipMemset(dfp, 0, sizeof(long long) * Pipeline::HostFeed::FLAG_WORDS);
        pl->allocs.push_back(dfp);
        hf.d_flag = (long long*)dfp;
      }
      for (int k = 0; k < Pipeline::HostFeed::FLAG_RING && fok; k++) {
        void* fp = nullptr;
        fok = hipHostMalloc(&fp, sizeof(long long) * Pipeline::HostFeed::FLAG_WORDS, hipHostMallocDefault) == hipSuccess;
        hf.h_flag[k] = (long long*)fp;
      }
      if (!fok) return ctx->fail(FLVIS_ERR_HIP, "image_feed_host: cannot allocate the sequence blocks");
    }
  }
  const int n_slots = hf.mode == 2 ? Pipeline::HostFeed::SLOTS : 2;
  const size_t npix = (size_t)S * w * h;
  for (int c = 0; c < 2; c++) {
    const size_t need = npix * bpp[c] + 256;
    if (hf.raw_bytes[c] < need) {
      hipStreamSynchronize(hf.strm);
      hipStreamSynchronize(ctx->stream);
      for (Lane* L : pl->lanes) hipStreamSynchronize(L->det_stream);
      for (int k = 0; k < n_slots; k++)
        if (!(hf.raw[k][c] = dalloc<uint8_t>(pl->allocs, need, false))) return ctx->fail(FLVIS_ERR_HIP, "image_feed_host: device allocation failed");
      hf.raw_bytes[c] = need;
    }
  }
  if ((ch0 > 1 || (ch1 > 1 && !depth_cam)) && hf.gray_bytes < npix + 256) {
    for (int k = 0; k < n_slots; k++)
      for (int c = 0; c < 2; c++)
        if (!(hf.gray[k][c] = dalloc<uint8_t>(pl->allocs, npix + 256, false))) return ctx->fail(FLVIS_ERR_HIP, "image_feed_host: device allocation failed");
    hf.gray_bytes = npix + 256;
  }
  const int slot = (int)(hf.n % n_slots), tslot = (int)(hf.n & 1);
  // hold_buffers contract: the previous call's buffers are free when this call returns -- its uploads must be done.
  // Mode 1: polled on a host-mapped counter a kernel of the copy stream stores after the uploads (hipEventSynchronize on its event also
  // waited for the KERNELS of the previous frame, so uploads and frames never overlapped: measured 2.25 ms per step = upload + frame).
  // Mode 2: hipStreamQuery on the copy stream, which holds nothing but the copies (the host reads the last copy's signal; no packet).
  auto wait_uploads = [&](long long calls) {
    int polls = 0;
    if (hf.mode == 2) {
      while (hipStreamQuery(hf.strm) == hipErrorNotReady) {
        if (++polls > 16) std::this_thread::sleep_for(std::chrono::microseconds(20));
      }
      (void)hipGetLastError();
      return;
    }
    while (*hf.h_up < calls) {
      if ((++polls & 255) == 0 && hipStreamQuery(hf.strm) == hipSuccess && *hf.h_up < calls) break;  // (idle copy stream: a failed copy)
      std::this_thread::sleep_for(std::chrono::microseconds(30));
    }
  };
  const auto th0 = std::chrono::steady_clock::now();
  if (hf.n >= 1) wait_uploads(hf.n);
  const auto th1 = std::chrono::steady_clock::now();
  // the frame that used this slot has been consumed.  (NOT implied by the host lead: the uploads are issued before this call waits for the
  // previous frame to start, i.e. while the frame before that may still be reading the slot -- without this wait the leg's poses differ from
  // the resident run's, bench.py's check caught it in round 5)
  // Mode 1: the copy stream waits for an event recorded behind that frame.  Mode 2: the HOST looks at the lanes' progress words -- the slot
  // was read by lane frame slot_frame[slot]; once a later frame of every lane has started (k_frame_head publishes its number when it
  // starts, behind k_frame_end of its predecessor and the joins in front of that), or the lane's stream has run dry, nothing reads it any
  // more.  With three slots the frame in question lies three calls back: the test is true when it is made.
  if (hf.mode == 2) {
    const long long need = hf.slot_frame[slot] + 1;
    if (hf.slot_frame[slot] > 0)
      for (Lane* L : pl->lanes) {
        int polls = 0;
        while (*L->h_progress < need) {
          if ((++polls & 63) == 0 && hipStreamQuery(L->st) == hipSuccess) break;  // (an idle stream: the frame is over, or its launch failed)
          std::this_thread::sleep_for(std::chrono::microseconds(20));
        }
      }
  } else if (hf.n >= 2) {
    hipStreamWaitEvent(hf.strm, hf.ev_free[slot], 0);
  }
  if (hf.timed[tslot]) {  // the uploads of two calls ago: done (the previous call's are, and the copy stream is in order)
    float ms = 0;
    if (hipEventElapsedTime(&ms, hf.ev_t0[tslot], hf.ev_t1[tslot]) == hipSuccess) hf.up_ms += ms, hf.up_bytes += (double)hf.timed_bytes[tslot], hf.up_calls += 1;
    else (void)hipGetLastError();
    hf.timed[tslot] = false;
  }
  if (hf.timing) hipEventRecord(hf.ev_t0[tslot], hf.strm);
  // (Measured in round 4, profiles/r04_h2d_full_timeline_*.txt: beside an SDMA upload the chain's latency-bound kernels run 2-3 x slower,
  // the LK launches do not.  Gating the uploads under LK launches -- left image under the previous frame's stereo LK, right image under
  // the frame's own head / temporal LK -- made the leg slower, 40k -> 31k frames/s: k_frame_head / k_track_prepare are latency-bound too
  // and the later start of the copies costs host lead.  A copy KERNEL of 8 .. 128 workgroups reading the caller's page-locked buffer over
  // PCIe instead of the SDMA engine: 19-34k frames/s against 35.3k (profiles/r04_lk_ab.md).  The uploads start as soon as their
  // staging slot is free, on the SDMA engine.  Round 6: what slowed the chain was not the copy but the barrier packets that waited for
  // it on the copy stream's hardware queue -- mode 2 has none, profiles/r06_h2d.md.)
  hipError_t e = hipSuccess;
  for (int c = 0; c < 2 && e == hipSuccess; c++) {
    const flvis_image* im = c ? h_img1 : h_img0;
    const size_t row = (size_t)w * bpp[c], img_bytes = row * h;
    bool contiguous = true;  // one block [S][h][w*bpp]: a single copy
    for (int s = 0; s < S && contiguous; s++)
      contiguous = (size_t)im[s].pitch == row && im[s].data == im[0].data + (size_t)s * img_bytes;
    // (FLVIS_H2D_CHUNK_MB, A/B knob of round 5, mode 1 only: the block in chunks with a tiny kernel between two of them -- the engine switch
    // leaves the link idle for a few microseconds, a window for the command processor's own traffic.  Measured, lost: 32.6k frames/s at 4 and
    // 8 MB, 22.7k at 2 MB against 28.7-37k in one piece: the uploads take longer and the chain is as long, profiles/r05_h2d.md)
    static const size_t h2d_chunk = getenv("FLVIS_H2D_CHUNK_MB") ? (size_t)atoi(getenv("FLVIS_H2D_CHUNK_MB")) << 20 : 0;
    if (contiguous && h2d_chunk && hf.mode == 1) {
      const size_t total = img_bytes * S;
      for (size_t off = 0; off < total && e == hipSuccess; off += h2d_chunk) {
        e = hipMemcpyAsync(hf.raw[slot][c] + off, im[0].data + off, std::min(h2d_chunk, total - off), hipMemcpyHostToDevice, hf.strm);
        if (off + h2d_chunk < total) launch_store_progress(hf.strm, hf.d_up + 1, (long long)off);  // (a scratch word beside the progress word)
      }
    } else if (contiguous) {
      e = hipMemcpyAsync(hf.raw[slot][c], im[0].data, img_bytes * S, hipMemcpyHostToDevice, hf.strm);
    } else {
      for (int s = 0; s < S && e == hipSuccess; s++)
        e = hipMemcpy2DAsync(hf.raw[slot][c] + (size_t)s * img_bytes, row, im[s].data, (size_t)im[s].pitch, row, h,
                             hipMemcpyHostToDevice, hf.strm);
    }
  }
  if (hf.timing && e == hipSuccess) {
    hipEventRecord(hf.ev_t1[tslot], hf.strm);
    hf.timed[tslot] = true;
    hf.timed_bytes[tslot] = (size_t)w * h * S * (bpp[0] + bpp[1]);
  }
  const long long seq = hf.n + 1;
  if (hf.mode == 2) {
    // the sequence block behind the images (page-locked ring of FLAG_RING blocks: the block of call n is rewritten by call n + FLAG_RING,
    // whose predecessors' uploads the host has seen finish)
    long long* fb = hf.h_flag[hf.n % Pipeline::HostFeed::FLAG_RING];
    for (int k = 0; k < Pipeline::HostFeed::FLAG_WORDS; k++) fb[k] = seq;
    if (e == hipSuccess) e = hipMemcpyAsync(hf.d_flag, fb, sizeof(long long) * Pipeline::HostFeed::FLAG_WORDS, hipMemcpyHostToDevice, hf.strm);
    if (e != hipSuccess) return ctx->hip_fail(e, "image_feed_host upload");
  } else {
    if (e == hipSuccess) e = hipEventRecord(hf.ev_done[slot], hf.strm);
    if (e != hipSuccess) return ctx->hip_fail(e, "image_feed_host upload");
    launch_store_progress(hf.strm, hf.d_up, seq);
  }
  hipStream_t st = ctx->stream;
  // Which stream waits for the uploads.  The detection stream, in front of the left image's ingest (mode 2, or mode 1 with FLVIS_H2D_WAIT=1):
  // every reader of the staged images is ordered behind that stream already -- the left pyramid's join in front of the temporal LK, the
  // right pyramid's in front of the stereo LK, which reads the right image in place.  Only for single-channel stereo input on a single
  // lane past the skipped start-up frames; otherwise the main stream waits, in front of the frame (rounds 1-5: always).
  static const int h2d_wait = getenv("FLVIS_H2D_WAIT") ? atoi(getenv("FLVIS_H2D_WAIT")) : -1;
  const bool wait_on_det = (h2d_wait < 0 ? hf.mode == 2 : h2d_wait == 1) && ch0 == 1 && ch1 == 1 && !depth_cam && pl->lanes.size() == 1 &&
                           pl->frames_fed >= (long long)pl->cfg.skip_first_n_imgs;
  if (hf.mode == 2) {
    if (wait_on_det) pl->up_flag = hf.d_flag, pl->up_seq = seq;
    else launch_wait_flag(st, hf.d_flag, Pipeline::HostFeed::FLAG_WORDS, seq, pl->lanes[0]->d_progress + 2);
  } else {
    if (wait_on_det) pl->up_event = hf.ev_done[slot];
    else hipStreamWaitEvent(st, hf.ev_done[slot], 0);
  }
  const uint8_t* d0 = hf.raw[slot][0];
  const uint8_t* d1 = hf.raw[slot][1];
  if (ch0 > 1) {
    launch_bgr_to_gray(st, hf.raw[slot][0], ch0, hf.gray[slot][0], npix);
    d0 = hf.gray[slot][0];
  }
  if (ch1 > 1 && !depth_cam) {
    launch_bgr_to_gray(st, hf.raw[slot][1], ch1, hf.gray[slot][1], npix);
    d1 = hf.gray[slot][1];
  }
  for (int s = 0; s < S; s++) hf.times[s] = h_img0[s].t;
  // one frame of lead in this mode: with two, the uploads' copies and events push the queued commands over what the HIP runtime
  // accepts without blocking the caller for milliseconds (measured: 16k vs 28k frames/s when that happened mid-run)
  static const int h2d_lead = getenv("FLVIS_H2D_LEAD") ? std::max(1, std::min(atoi(getenv("FLVIS_H2D_LEAD")), 4)) : 1;  // (A/B knob)
  pl->host_lead_cap = h2d_lead;
  const auto th2 = std::chrono::steady_clock::now();
  pl->chain_continuous = wait_on_det && hf.mode == 2 && hf.last_call_frame == pl->frames_fed;  // (the previous frame was this entry's too)
  const int rc = flvis_image_feed(ctx, d0, d1, hf.times.data(), h_out, with_local_map);
  pl->chain_continuous = false;
  hf.last_call_frame = pl->frames_fed;
  pl->up_event = nullptr;
  pl->up_flag = nullptr;
  const auto th3 = std::chrono::steady_clock::now();
  hf.ms_wait_uploads += std::chrono::duration<double, std::milli>(th1 - th0).count();
  hf.ms_issue += std::chrono::duration<double, std::milli>(th2 - th1).count();
  hf.ms_feed += std::chrono::duration<double, std::milli>(th3 - th2).count();
  pl->host_lead_cap = 4;
  if (hf.mode == 2) hf.slot_frame[slot] = pl->lanes[0]->frames_uploaded;
  else hipEventRecord(hf.ev_free[slot], st);
  hf.n++;
  if (rc != FLVIS_OK) return rc;
  if (!hold_buffers) wait_uploads(hf.n);  // the caller may reuse its buffers at once: wait for the uploads (not for the frame)
  return FLVIS_OK;
}


int flvis_imu_feed_all(flvis_ctx* ctx, const int* h_counts, const double* h_samples, int samples_per_stream) {
  if (!ctx || !ctx->pipe || !h_counts || !h_samples || samples_per_stream <= 0) return FLVIS_ERR_INVALID_ARG;
  Pipeline* pl = ctx->pipe;
  for (int s = 0; s < pl->S; s++) {
    int n = h_counts[s];
    if (n < 0 || n > samples_per_stream) return ctx->fail(FLVIS_ERR_INVALID_ARG, "imu_feed_all: bad count");
    int rc = flvis_imu_feed_flvis_frame(ctx, s, n, h_samples + (size_t)s * samples_per_stream * 7);
    if (rc) return rc;
  }
  return FLVIS_OK;
}

// The caller's per-frame loop (IMU samples, then the stereo pair) for n_steps frames in ONE call: a driver that feeds frames already
// resident in HBM -- a replay, bench.py's timed region, one rank of a multi-GPU job -- does not come back to its host language between
// frames (eight Python interpreters on one host contending for cores are then not on the path).
int flvis_run_steps(flvis_ctx* ctx, int n_steps, const flvis_step* steps, int with_local_map, double* h_call_ms) {
  if (!ctx || !ctx->pipe || n_steps < 0 || (n_steps > 0 && !steps)) return FLVIS_ERR_INVALID_ARG;
  for (int k = 0; k < n_steps; k++) {
    const flvis_step& f = steps[k];
    const auto t0 = std::chrono::steady_clock::now();
    if (f.h_imu_counts) {
      const int rc = flvis_imu_feed_all(ctx, f.h_imu_counts, f.h_imu_samples, f.imu_samples_per_stream);
      if (rc != FLVIS_OK) return rc;
    }
    ctx->pipe->defer_ba = k + 1 < n_steps;
    ctx->pipe->chain_continuous = k > 0;
    const int rc = flvis_image_feed(ctx, f.d_img0, f.d_img1, f.h_times, nullptr, with_local_map);
    ctx->pipe->defer_ba = false;
    ctx->pipe->chain_continuous = false;
    if (rc != FLVIS_OK) return rc;
    if (h_call_ms) h_call_ms[k] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }
  return FLVIS_OK;
}

int flvis_prof_enable(flvis_ctx* ctx, int max_steps) { return flvis_prof_enable_stages(ctx, max_steps, ~0ull); }

int flvis_prof_enable_stages(flvis_ctx* ctx, int max_steps, uint64_t stage_mask) {
  if (!ctx || !ctx->pipe || max_steps < 0) return FLVIS_ERR_INVALID_ARG;
  Pipeline* pl = ctx->pipe;
  pl->prof_mask = stage_mask;
  sync_all(ctx);
  pl->prof_cap = max_steps;
  pl->prof_step = 0;
  // release scope of the stage events (FLVIS_PROF_EVENT_SCOPE=agent|system; default: what the pipeline's own events use)
  unsigned prof_scope = pl->ev_flags & hipEventReleaseToDevice;
  if (const char* e = getenv("FLVIS_PROF_EVENT_SCOPE")) prof_scope = !strcmp(e, "agent") ? hipEventReleaseToDevice : 0u;
  for (Lane* L : pl->lanes) {
    for (hipEvent_t e : L->prof_ev) hipEventDestroy(e);
    L->prof_ev.clear();
    L->prof_ev.resize((size_t)max_steps * (2 * PROF_STAGES));
    L->prof18_rec.assign((size_t)max_steps + 1, 0);
    for (auto& e : L->prof_ev)
      if (hipEventCreateWithFlags(&e, prof_scope) != hipSuccess) return ctx->fail(FLVIS_ERR_HIP, "prof_enable: hipEventCreate failed");
  }
  return FLVIS_OK;
}

int flvis_prof_stage_count(void) { return PROF_STAGES; }
const char* flvis_prof_stage_name(int i) { return (i >= 0 && i < PROF_STAGES) ? kStageNames[i] : ""; }
int flvis_tracker_lanes(flvis_ctx* ctx) { return (ctx && ctx->pipe) ? (int)ctx->pipe->lanes.size() : 0; }

static bool prof_stage_timed(const Pipeline* pl, int i) {
  if (!((pl->prof_mask >> i) & 1ull)) return false;
  if (i >= 10 && i <= 12 && ((pl->prof_mask >> 10) & 7ull) != 7ull) return false;  // the GFTT chain is timed as a whole
  return true;
}

// per stage: the elapsed time of one LAUNCH (one lane's kernel(s) of that stage), summed over the armed frames and averaged
// over the lanes -- with several lanes the stages of different lanes overlap, so the sum over stages exceeds the step time
int flvis_prof_read(flvis_ctx* ctx, double* h_ms_per_stage, int* n_steps) {
  if (!ctx || !ctx->pipe || !h_ms_per_stage || !n_steps) return FLVIS_ERR_INVALID_ARG;
  Pipeline* pl = ctx->pipe;
  sync_all(ctx);
  for (int i = 0; i < PROF_STAGES; i++) h_ms_per_stage[i] = 0;
  size_t rec18 = 0;
  for (Lane* L : pl->lanes)
    for (int k = 0; k < pl->prof_step; k++)
      for (int i = 0; i < PROF_STAGES; i++) {
        float ms = 0;
        if (!prof_stage_timed(pl, i)) continue;
        if (i == 18) {
          if (!L->prof18_rec[(size_t)k]) continue;
          rec18++;
        }
        // (a stage that was not enqueued in this frame -- a deferred local-map launch, FLVIS_BA_START -- has no recorded events: it counts 0)
        if (hipEventElapsedTime(&ms, L->prof_ev[(size_t)k * (2 * PROF_STAGES) + 2 * i], L->prof_ev[(size_t)k * (2 * PROF_STAGES) + 2 * i + 1]) != hipSuccess) {
          (void)hipGetLastError();
          ms = 0;
        }
        h_ms_per_stage[i] += ms / (double)pl->lanes.size();
      }
  // (stage 18 as the mean over the steps that timed a launch: the caller divides every stage's sum by n_steps)
  if (rec18 > 0 && rec18 < pl->lanes.size() * (size_t)pl->prof_step) h_ms_per_stage[18] *= (double)(pl->lanes.size() * (size_t)pl->prof_step) / (double)rec18;
  *n_steps = pl->prof_step;
  return FLVIS_OK;
}

// per armed frame: the slowest lane's elapsed time of the stage
int flvis_prof_read_steps(flvis_ctx* ctx, int stage, double* h_ms, int cap) {
  if (!ctx || !ctx->pipe || !h_ms || stage < 0 || stage >= PROF_STAGES || cap < 0) return FLVIS_ERR_INVALID_ARG;
  Pipeline* pl = ctx->pipe;
  if (!prof_stage_timed(pl, stage)) return ctx->fail(FLVIS_ERR_INVALID_ARG, "prof_read_steps: stage was not enabled");
  sync_all(ctx);
  int n = 0;
  for (int k = 0; k < pl->prof_step && n < cap; k++, n++) {
    double worst = 0;
    for (Lane* L : pl->lanes) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, L->prof_ev[(size_t)k * (2 * PROF_STAGES) + 2 * stage], L->prof_ev[(size_t)k * (2 * PROF_STAGES) + 2 * stage + 1]) != hipSuccess) {
        (void)hipGetLastError();
        ms = 0;
      }
      worst = std::max(worst, (double)ms);
    }
    h_ms[n] = worst;
  }
  return n;
}

int flvis_get_landmarks(flvis_ctx* ctx, int stream, int cap, int64_t* h_id, double* h_2d, double* h_2du, double* h_3d,
                        uint8_t* h_flags) {
  if (!ctx || !ctx->pipe) return FLVIS_ERR_INVALID_ARG;
  Pipeline* pl = ctx->pipe;
  if (stream < 0 || stream >= pl->S) return ctx->fail(FLVIS_ERR_INVALID_ARG, "bad stream");
  sync_all(ctx);
  int ls;
  Lane& L = pl->lane_of(stream, ls);
  StreamState st;
  hipMemcpy(&st, L.pipe.st + ls, sizeof(st), hipMemcpyDeviceToHost);
  int n = st.n_lm[st.cur];
  std::vector<Landmark> lm(n);
  if (n) hipMemcpy(lm.data(), L.pipe.lm + ((size_t)st.cur * L.S + ls) * NMAX, sizeof(Landmark) * n, hipMemcpyDeviceToHost);
  for (int i = 0; i < n && i < cap; i++) {
    h_id[i] = lm[i].id;
    h_2d[2 * i] = lm[i].p2d[0];
    h_2d[2 * i + 1] = lm[i].p2d[1];
    h_2du[2 * i] = lm[i].p2u[0];
    h_2du[2 * i + 1] = lm[i].p2u[1];
    for (int j = 0; j < 3; j++) h_3d[3 * i + j] = lm[i].p3w[j];
    h_flags[i] = (uint8_t)((lm[i].has3d ? 1 : 0) | (lm[i].inlier ? 2 : 0));
  }
  return n;
}

int flvis_get_keyframe(flvis_ctx* ctx, int stream, int cap, int64_t* frame_id, double* T7, int64_t* h_id, double* h_2d,
                       double* h_3d) {
  if (!ctx || !ctx->pipe) return FLVIS_ERR_INVALID_ARG;
  Pipeline* pl = ctx->pipe;
  if (stream < 0 || stream >= pl->S) return ctx->fail(FLVIS_ERR_INVALID_ARG, "bad stream");
  sync_all(ctx);
  int ls;
  Lane& L = pl->lane_of(stream, ls);
  std::vector<KeyFrameDev> kfv(1);
  KeyFrameDev& kf = kfv[0];
  FrameOut fo;
  hipMemcpy(&fo, L.pipe.out + ls, sizeof(FrameOut), hipMemcpyDeviceToHost);
  if (!fo.new_keyframe) return 0;  // the last frame of this stream did not become a keyframe
  unsigned tl = 0;
  hipMemcpy(&tl, L.pipe.kfq_tail + ls, sizeof(unsigned), hipMemcpyDeviceToHost);
  if (tl == 0) return 0;
  hipMemcpy(&kf, L.pipe.kfq + (size_t)ls * KFQ + ((tl - 1) % KFQ), sizeof(KeyFrameDev), hipMemcpyDeviceToHost);
  *frame_id = kf.frame_id;
  memcpy(T7, kf.T_c_w, 56);
  for (int i = 0; i < kf.lm_count && i < cap; i++) {
    h_id[i] = kf.lm_id[i];
    h_2d[2 * i] = kf.lm_2d[i][0];
    h_2d[2 * i + 1] = kf.lm_2d[i][1];
    for (int j = 0; j < 3; j++) h_3d[3 * i + j] = kf.lm_3d[i][j];
  }
  return kf.lm_count;
}

int flvis_get_keyframe_msg(flvis_ctx* ctx, int stream, int cap, flvis_keyframe* kf, int64_t* h_id, double* h_2d, double* h_3d,
                           uint8_t* h_img0, uint8_t* h_img1) {
  if (!ctx || !ctx->pipe || !kf || cap < 0 || (cap && (!h_id || !h_2d || !h_3d))) return FLVIS_ERR_INVALID_ARG;
  Pipeline* pl = ctx->pipe;
  if (stream < 0 || stream >= pl->S) return ctx->fail(FLVIS_ERR_INVALID_ARG, "bad stream");
  int64_t fid = 0;
  const int n = flvis_get_keyframe(ctx, stream, cap, &fid, kf->T_c_w, h_id, h_2d, h_3d);
  if (n <= 0) return n;
  int ls;
  Lane& L = pl->lane_of(stream, ls);
  const int w = pl->cfg.image_width, h = pl->cfg.image_height;
  const bool depth_cam = pl->cfg.cam_type == CAM_DEPTH;
  unsigned tl = 0;
  double stamp = 0;
  int slot = 0;
  hipMemcpy(&tl, L.pipe.kfq_tail + ls, sizeof(unsigned), hipMemcpyDeviceToHost);
  hipMemcpy(&stamp, &L.pipe.kfq[(size_t)ls * KFQ + ((tl - 1) % KFQ)].stamp, sizeof(double), hipMemcpyDeviceToHost);
  hipMemcpy(&slot, L.pipe.img_slot + ls, sizeof(int), hipMemcpyDeviceToHost);
  kf->frame_id = fid;
  kf->command = 0;  // KFMSG_CMD_NONE (keyframe_msg.h:10)
  kf->stamp = stamp;
  kf->lm_count = n;
  kf->lm_id = h_id;
  kf->lm_2d = h_2d;
  kf->lm_3d = h_3d;
  kf->img0 = flvis_image{h_img0, w, h, w, 1, stamp};
  kf->img1 = flvis_image{h_img1, w, h, depth_cam ? 2 * w : w, 1, stamp};
  hipError_t e = hipSuccess;
  if (h_img0)  // level 0 of the left pyramid, the slot of the stream's last processed frame (after equalizeHist where used)
    e = hipMemcpy2D(h_img0, w, L.pyr0[slot & 1][0] + (size_t)ls * pl->lstride[0], pl->lpitch[0], w, h, hipMemcpyDeviceToHost);
  if (e == hipSuccess && h_img1) {
    const bool eq = pl->cfg.need_equal_hist != 0, aligned = (w & 15) == 0;
    const uint8_t* tab[2] = {nullptr, nullptr};
    if (depth_cam || (!eq && aligned)) {  // read in place from the caller's buffer of that frame (still the table's entry)
      tab[0] = L.h_tab[0], tab[1] = L.h_tab[1];
      const size_t bpp = depth_cam ? 2 : 1;
      if (e == hipSuccess) e = hipMemcpy(h_img1, tab[1] + (size_t)ls * w * h * bpp, (size_t)w * h * bpp, hipMemcpyDeviceToHost);
    } else {
      e = hipMemcpy2D(h_img1, w, L.pyr1[0] + (size_t)ls * pl->lstride[0], pl->lpitch[0], w, h, hipMemcpyDeviceToHost);
    }
  }
  if (e != hipSuccess) return ctx->hip_fail(e, "get_keyframe_msg");
  return n;
}

static int read_correction(Lane& L, int ls, int cap, int64_t* frame_id, double* T7, int* lm_count, int64_t* h_id,
                           double* h_3d, int* oc, int64_t* h_oid) {
  std::vector<CorrectionDev> cv(1);
  CorrectionDev& c = cv[0];
  hipMemcpy(&c, L.pipe.corr + ls, sizeof(CorrectionDev), hipMemcpyDeviceToHost);
  if (!c.valid) return 0;
  *frame_id = c.frame_id;
  memcpy(T7, c.T_c_w, 56);
  *lm_count = c.lm_count;
  for (int i = 0; i < c.lm_count && i < cap; i++) {
    h_id[i] = c.lm_id[i];
    for (int j = 0; j < 3; j++) h_3d[3 * i + j] = c.lm_3d[i][j];
  }
  *oc = c.lm_outlier_count;
  for (int i = 0; i < c.lm_outlier_count && i < cap; i++) h_oid[i] = c.lm_outlier_id[i];
  return 1;
}

int flvis_get_correction(flvis_ctx* ctx, int stream, int cap, int64_t* frame_id, double* T7, int* lm_count, int64_t* h_id,
                         double* h_3d, int* oc, int64_t* h_oid) {
  if (!ctx || !ctx->pipe) return FLVIS_ERR_INVALID_ARG;
  if (stream < 0 || stream >= ctx->pipe->S) return ctx->fail(FLVIS_ERR_INVALID_ARG, "bad stream");
  sync_all(ctx);
  int ls;
  Lane& L = ctx->pipe->lane_of(stream, ls);
  return read_correction(L, ls, cap, frame_id, T7, lm_count, h_id, h_3d, oc, h_oid);
}

int flvis_get_trajectory(flvis_ctx* ctx, int stream, int first, int n, double* h_rows9) {
  if (!ctx || !ctx->pipe || !h_rows9) return FLVIS_ERR_INVALID_ARG;
  Pipeline* pl = ctx->pipe;
  if (stream < 0 || stream >= pl->S) return ctx->fail(FLVIS_ERR_INVALID_ARG, "get_trajectory: out of range");
  int ls;
  Lane& L = pl->lane_of(stream, ls);
  if (first < 0 || n < 0 || first + n > L.pipe.traj_cap) return ctx->fail(FLVIS_ERR_INVALID_ARG, "get_trajectory: out of range");
  sync_all(ctx);
  hipMemcpy(h_rows9, L.pipe.traj + ((size_t)ls * L.pipe.traj_cap + first) * 9, sizeof(double) * 9 * n, hipMemcpyDeviceToHost);
  return n;
}

int flvis_write_trajectory(flvis_ctx* ctx, int stream, int first, int n, const char* path, int format, double min_dt) {
  if (!ctx || !ctx->pipe || !path || (format != 0 && format != 1)) return FLVIS_ERR_INVALID_ARG;
  std::vector<double> rows((size_t)(n > 0 ? n : 0) * 9);
  int got = flvis_get_trajectory(ctx, stream, first, n, rows.data());
  if (got < 0) return got;
  FILE* f = fopen(path, "w");
  if (!f) return ctx->fail(FLVIS_ERR_INVALID_ARG, "write_trajectory: cannot open the output file");
  int written = 0;
  bool have_first = false;
  double first_t = 0;
  for (int i = 0; i < got; i++) {
    const double* r = &rows[(size_t)i * 9];
    if (((int)r[8] & 15) != 1) continue;  // only frames of a TRACKING stream carry a pose
    const double t = r[0];
    if (min_dt > 0) {
      // vo_repub_rec.cpp:77-78: `last_time` is a function-static initialised at the FIRST call and never updated, so the
      // recorder drops the poses of the first min_dt (0.1 s of wall clock there, stamps here) and writes every pose after
      if (!have_first) {
        have_first = true;
        first_t = t;
      }
      if (!(t - first_t > min_dt)) continue;
    }
    // T_c_w = (t, q) -> T_w_c: R_w_c = R^T, centre = -R^T t
    double qx = r[4], qy = r[5], qz = r[6], qw = r[7];
    const double qn = std::sqrt(qx * qx + qy * qy + qz * qz + qw * qw);
    qx /= qn; qy /= qn; qz /= qn; qw /= qn;
    const double R[3][3] = {{1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw)},
                            {2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw)},
                            {2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)}};
    double c[3];
    for (int k = 0; k < 3; k++) c[k] = -(R[0][k] * r[1] + R[1][k] * r[2] + R[2][k] * r[3]);
    if (format == 0) {
      // operator<< with setprecision(6) (default float notation) == %.6g; the stamp prints like ros::Time (sec.nsec)
      fprintf(f, "%.9f %.6g %.6g %.6g %.6g %.6g %.6g %.6g\n", t, c[0], c[1], c[2], qw, -qx, -qy, -qz);
    } else {
      fprintf(f, "%.6g %.6g %.6g %.6g %.6g %.6g %.6g %.6g %.6g %.6g %.6g %.6g\n", R[0][0], R[1][0], R[2][0], c[0], R[0][1], R[1][1],
              R[2][1], c[1], R[0][2], R[1][2], R[2][2], c[2]);
    }
    written++;
  }
  fclose(f);
  return written;
}


// The recorder on /imu_pose (launch/flvis_euroc_mav.launch:83-103: vo_repub_rec with sub_type PoseStamped, sub_topic /imu_pose,
// output est.txt): pubPose(q_w_i, pos_w_i, stamp) of every IMU sample, `stamp x y z qw qx qy qz` (vo_repub_rec.cpp:74-91), with
// the throttle as written there (see flvis_write_trajectory).  Rows as flvis_get_imu_states returns them.
int flvis_write_imu_trajectory_run(const double* h_rows11, int n, const char* path, double min_dt, int append, double t_first) {
  if (!path || n < 0 || (n > 0 && !h_rows11)) return FLVIS_ERR_INVALID_ARG;
  FILE* f = fopen(path, append ? "a" : "w");
  if (!f) return FLVIS_ERR_CONFIG;
  int written = 0;
  // the recorder's `last_time` is the stamp of the first message of the RUN (vo_repub_rec.cpp:77-78) and is never updated: rows within
  // min_dt of THAT stamp are dropped, whichever batch they arrive in
  const bool throttle = min_dt > 0 && (t_first == t_first);
  for (int i = 0; i < n; i++) {
    const double* r = h_rows11 + (size_t)i * 11;
    if (throttle && !(r[0] - t_first > min_dt)) continue;
    fprintf(f, "%.9f %.6g %.6g %.6g %.6g %.6g %.6g %.6g\n", r[0], r[5], r[6], r[7], r[1], r[2], r[3], r[4]);
    written++;
  }
  fclose(f);
  return written;
}

// (the two-argument form of the run: the batch that creates the file carries the run's first stamp; a batch that is appended without
// naming it is written in full -- callers whose first batch may be shorter than min_dt use flvis_write_imu_trajectory_run)
int flvis_write_imu_trajectory(const double* h_rows11, int n, const char* path, double min_dt, int append) {
  const double nan = std::numeric_limits<double>::quiet_NaN();
  return flvis_write_imu_trajectory_run(h_rows11, n, path, min_dt, append, (!append && n > 0 && h_rows11) ? h_rows11[0] : nan);
}

int flvis_get_counters_n(flvis_ctx* ctx, int n, int64_t* h) {
  if (!ctx || !ctx->pipe || !h || n < 1 || n > 4) return FLVIS_ERR_INVALID_ARG;
  sync_all(ctx);
  int64_t v[4] = {ctx->pipe->frames_fed * ctx->pipe->S, 0, 0, 0};
  for (Lane* L : ctx->pipe->lanes) {
    long long c[8];
    hipError_t e = hipMemcpy(c, L->pipe.counters, sizeof(c), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return ctx->hip_fail(e, "get_counters");
    for (int k = 1; k < 4; k++) v[k] += c[k];
  }
  for (int k = 0; k < n; k++) h[k] = v[k];
  return FLVIS_OK;
}

int flvis_get_counters(flvis_ctx* ctx, int64_t* h3) { return flvis_get_counters_n(ctx, 3, h3); }

// per stream: keyframes the tracker has emitted (KeyFrame messages of /vo_kf) and optimisations its local map has run
int flvis_get_local_map_counts(flvis_ctx* ctx, int64_t* h_keyframes, int64_t* h_ba_runs) {
  if (!ctx || !ctx->pipe) return FLVIS_ERR_INVALID_ARG;
  sync_all(ctx);
  for (Lane* L : ctx->pipe->lanes) {
    if (h_keyframes) {
      std::vector<unsigned> tl(L->S);
      hipError_t e = hipMemcpy(tl.data(), L->pipe.kfq_tail, sizeof(unsigned) * L->S, hipMemcpyDeviceToHost);
      if (e != hipSuccess) return ctx->hip_fail(e, "get_local_map_counts");
      for (int i = 0; i < L->S; i++) h_keyframes[L->s0 + i] = tl[i];
    }
    if (h_ba_runs) {
      std::vector<long long> r(L->S);
      hipError_t e = hipMemcpy2D(r.data(), sizeof(long long), reinterpret_cast<const char*>(L->pipe.win) + offsetof(WindowDev, ba_runs),
                                 sizeof(WindowDev), sizeof(long long), L->S, hipMemcpyDeviceToHost);
      if (e != hipSuccess) return ctx->hip_fail(e, "get_local_map_counts");
      for (int i = 0; i < L->S; i++) h_ba_runs[L->s0 + i] = r[i];
    }
  }
  return FLVIS_OK;
}

// test aid: the tracker's pyramid construction on its own (see include/flvis_hip.h for the layout of d_out)
int flvis_debug_pyramid(flvis_ctx* ctx, const uint8_t* d_src, int w, int h, int n_img, int levels, int bx, int by, int ingest, uint8_t* d_out,
                        size_t out_bytes) {
  if (!ctx) return FLVIS_ERR_INVALID_ARG;
  if (!d_src || !d_out || w < 2 || h < 2 || (w & 3) || n_img <= 0 || levels < 1 || levels >= LK_MAX_LEVELS || bx < 0 || by < 0 || (bx & 15) ||
      ((uintptr_t)d_src & 3) || ((uintptr_t)d_out & 63))
    return ctx->fail(FLVIS_ERR_INVALID_ARG, "debug_pyramid: bad args (w % 4 == 0, 1 <= levels <= 5, bx % 16 == 0, d_out 64-byte aligned)");
  PyrSel q{};
  q.levels = levels;
  size_t off = 0;
  int lw = w, lh = h;
  for (int l = 0; l <= levels; l++) {
    q.w[l] = lw, q.h[l] = lh, q.bx[l] = bx, q.by[l] = by;
    q.pitch[l] = ((lw + 15) & ~15) + 2 * bx;
    q.stride[l] = (size_t)q.pitch[l] * (lh + 2 * by);
    q.lvl[l] = img_plain(d_out + off + (size_t)by * q.pitch[l] + bx);
    off += (q.stride[l] * n_img + 63) & ~(size_t)63;
    lw = (lw + 1) / 2, lh = (lh + 1) / 2;
  }
  if (off > out_bytes) return ctx->fail(FLVIS_ERR_CAPACITY, "debug_pyramid: d_out too small");
  const bool bordered = bx > 0 || by > 0;
  unsigned left = pyramid_levels(ctx->stream, bordered, img_plain(d_src), w, (size_t)w * h, ingest != 0, ingest != 0, q, n_img, nullptr);
  if (left) launch_pyr_border(ctx->stream, q, n_img, nullptr, left);
  hipError_t e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) return ctx->fail(FLVIS_ERR_HIP, hipGetErrorString(e));
  return FLVIS_OK;
}

// measurement aid: from now on the tracker's two LK launches per frame count Gauss-Newton iterations and points per pyramid level
// (flvis_debug_counters [36 + 2 l], [37 + 2 l]: temporal LK, level l; [48 + 2 l], [49 + 2 l]: stereo LK); costs one atomic per point and level
int flvis_debug_lk_stats(flvis_ctx* ctx, int enable) {
  if (!ctx || !ctx->pipe) return FLVIS_ERR_INVALID_ARG;
  ctx->pipe->lk_stats = enable != 0;
  return FLVIS_OK;
}

// sums over the lanes; [3] = keyframes dropped because a stream's queue was full (0 unless the back-pressure was defeated)
int flvis_debug_counters(flvis_ctx* ctx, int64_t* h64) {
  if (!ctx || !ctx->pipe || !h64) return FLVIS_ERR_INVALID_ARG;
  sync_all(ctx);
  for (int i = 0; i < 64; i++) h64[i] = 0;
  for (Lane* L : ctx->pipe->lanes) {
    long long c[64];
    hipMemcpy(c, L->pipe.counters, sizeof(c), hipMemcpyDeviceToHost);
    for (int i = 0; i < 64; i++) h64[i] += c[i];
  }
  return FLVIS_OK;
}

// Multi-lane trackers: how many later flvis_image_feed calls the caller leaves the input buffers of a call untouched (it cycles
// through n + 1 buffers).  0 (default): the context's stream waits for every lane to finish the frame before it goes on.
int flvis_set_input_hold(flvis_ctx* ctx, int n_frames) {
  if (!ctx || !ctx->pipe) return FLVIS_ERR_INVALID_ARG;
  if (n_frames < 0 || n_frames >= Lane::HOLD_RING) return ctx->fail(FLVIS_ERR_INVALID_ARG, "set_input_hold: 0 .. 7 frames");
  ctx->pipe->input_hold = n_frames;
  return FLVIS_OK;
}
// Tuning aid: host milliseconds spent inside flvis_image_feed since the tracker was created, [0] in total, [1] of it blocked on
// the pinned upload ring (the host running PIN_RING frames ahead of the GPU), [2] calls.
int flvis_debug_host_times(flvis_ctx* ctx, double* h_out3) {
  if (!ctx || !ctx->pipe || !h_out3) return FLVIS_ERR_INVALID_ARG;
  h_out3[0] = ctx->pipe->host_ms_total;
  h_out3[1] = ctx->pipe->host_ms_wait;
  h_out3[2] = (double)ctx->pipe->frames_fed;
  return FLVIS_OK;
}
// ... of flvis_image_feed_host: [0] blocked on the previous call's uploads, [1] issuing this call's uploads, [2] inside flvis_image_feed
// (whose share blocked on the host lead is flvis_debug_host_times [1]), [3] calls
int flvis_debug_host_feed_times(flvis_ctx* ctx, double* h_out4) {
  if (!ctx || !ctx->pipe || !h_out4) return FLVIS_ERR_INVALID_ARG;
  const Pipeline::HostFeed& hf = ctx->pipe->hf;
  h_out4[0] = hf.ms_wait_uploads, h_out4[1] = hf.ms_issue, h_out4[2] = hf.ms_feed, h_out4[3] = (double)hf.n;
  return FLVIS_OK;
}

int flvis_debug_host_feed_timing(flvis_ctx* ctx, int enable, double* h_out3) {
  if (!ctx || !ctx->pipe) return FLVIS_ERR_INVALID_ARG;
  Pipeline::HostFeed& hf = ctx->pipe->hf;
  if (enable >= 0) hf.timing = enable != 0;
  if (h_out3) h_out3[0] = hf.up_ms, h_out3[1] = hf.up_bytes, h_out3[2] = hf.up_calls;
  return FLVIS_OK;
}

// IMU rotation factor of the window BA (an addition: the reference's window holds reprojection edges only).  Off by default.
int flvis_set_imu_factor(flvis_ctx* ctx, int enable, double sigma_gyro) {
  if (!ctx || !ctx->pipe) return FLVIS_ERR_INVALID_ARG;
  if (enable && !(sigma_gyro > 0)) return ctx->fail(FLVIS_ERR_INVALID_ARG, "set_imu_factor: sigma_gyro must be positive");
  sync_all(ctx);  // the flag travels in the kernel arguments of the next local-map launch
  for (Lane* L : ctx->pipe->lanes) {
    L->pipe.imu_factor = enable ? 1 : 0;
    L->pipe.imu_sigma_g = sigma_gyro;
  }
  return FLVIS_OK;
}

// Position rows of the factor on top of the rotation rows (flvis_set_imu_factor must be on): sigma_acc = accelerometer noise density
// [m/s^2/sqrt(Hz)], information I3 / (sigma_acc^2 dt^3 / 3); <= 0 switches the position rows off again.
int flvis_set_imu_factor_accel(flvis_ctx* ctx, double sigma_acc) {
  if (!ctx || !ctx->pipe) return FLVIS_ERR_INVALID_ARG;
  sync_all(ctx);
  for (Lane* L : ctx->pipe->lanes) L->pipe.imu_sigma_a = sigma_acc > 0 ? sigma_acc : 0.0;
  return FLVIS_OK;
}

// the position part of the last keyframe's preintegration (flvis_get_keyframe_imu gives the rotation part and says whether it links)
int flvis_get_keyframe_imu_pos(flvis_ctx* ctx, int stream, double* dp3, double* va3) {
  if (!ctx || !ctx->pipe || !dp3 || !va3) return FLVIS_ERR_INVALID_ARG;
  Pipeline* pl = ctx->pipe;
  if (stream < 0 || stream >= pl->S) return ctx->fail(FLVIS_ERR_INVALID_ARG, "bad stream");
  sync_all(ctx);
  int ls;
  Lane& L = pl->lane_of(stream, ls);
  unsigned tl = 0;
  hipMemcpy(&tl, L.pipe.kfq_tail + ls, sizeof(unsigned), hipMemcpyDeviceToHost);
  if (tl == 0) return 0;
  double h[6];
  static_assert(offsetof(KeyFrameDev, imu_va) == offsetof(KeyFrameDev, imu_dp) + 24, "KeyFrameDev imu position block layout");
  const char* src = reinterpret_cast<const char*>(L.pipe.kfq + (size_t)ls * KFQ + ((tl - 1) % KFQ)) + offsetof(KeyFrameDev, imu_dp);
  if (hipMemcpy(h, src, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return ctx->fail(FLVIS_ERR_HIP, "get_keyframe_imu_pos");
  memcpy(dp3, h, 24);
  memcpy(va3, h + 3, 24);
  return 1;
}

int flvis_get_keyframe_imu(flvis_ctx* ctx, int stream, double* dq_wxyz, double* dt) {
  if (!ctx || !ctx->pipe || !dq_wxyz || !dt) return FLVIS_ERR_INVALID_ARG;
  Pipeline* pl = ctx->pipe;
  if (stream < 0 || stream >= pl->S) return ctx->fail(FLVIS_ERR_INVALID_ARG, "bad stream");
  sync_all(ctx);
  int ls;
  Lane& L = pl->lane_of(stream, ls);
  FrameOut fo;
  hipMemcpy(&fo, L.pipe.out + ls, sizeof(FrameOut), hipMemcpyDeviceToHost);
  unsigned tl = 0;
  hipMemcpy(&tl, L.pipe.kfq_tail + ls, sizeof(unsigned), hipMemcpyDeviceToHost);
  if (!fo.new_keyframe || tl == 0) return 0;
  struct {
    double dq[4], dt;
    int valid, pad;
  } h;
  static_assert(offsetof(KeyFrameDev, imu_valid) == offsetof(KeyFrameDev, imu_dq) + 40, "KeyFrameDev imu block layout");
  const char* src = reinterpret_cast<const char*>(L.pipe.kfq + (size_t)ls * KFQ + ((tl - 1) % KFQ)) + offsetof(KeyFrameDev, imu_dq);
  hipMemcpy(&h, src, sizeof(h), hipMemcpyDeviceToHost);
  memcpy(dq_wxyz, h.dq, 32);
  *dt = h.dt;
  return h.valid ? 1 : 0;
}

static int ba_push_impl(flvis_ctx* ctx, int stream, int64_t frame_id, const double* T7, const double* imu_dq, double imu_dt,
                        int lm_count, const int64_t* h_id, const double* h_2d, const double* h_3d, int cap, int64_t* out_frame_id,
                        double* out_T7, int* out_lm_count, int64_t* out_lm_id, double* out_lm_3d, int* out_oc, int64_t* out_oid,
                        const double* imu_dp = nullptr, const double* imu_va = nullptr);

int flvis_ba_push_keyframe(flvis_ctx* ctx, int stream, int64_t frame_id, const double* T7, int lm_count, const int64_t* h_id,
                           const double* h_2d, const double* h_3d, int cap, int64_t* out_frame_id, double* out_T7,
                           int* out_lm_count, int64_t* out_lm_id, double* out_lm_3d, int* out_oc, int64_t* out_oid) {
  return ba_push_impl(ctx, stream, frame_id, T7, nullptr, 0.0, lm_count, h_id, h_2d, h_3d, cap, out_frame_id, out_T7, out_lm_count,
                      out_lm_id, out_lm_3d, out_oc, out_oid);
}
int flvis_ba_push_keyframe_imu(flvis_ctx* ctx, int stream, int64_t frame_id, const double* T7, const double* imu_dq_wxyz,
                               double imu_dt, int lm_count, const int64_t* h_id, const double* h_2d, const double* h_3d, int cap,
                               int64_t* out_frame_id, double* out_T7, int* out_lm_count, int64_t* out_lm_id, double* out_lm_3d,
                               int* out_oc, int64_t* out_oid) {
  return ba_push_impl(ctx, stream, frame_id, T7, imu_dq_wxyz, imu_dt, lm_count, h_id, h_2d, h_3d, cap, out_frame_id, out_T7,
                      out_lm_count, out_lm_id, out_lm_3d, out_oc, out_oid);
}

int flvis_ba_push_keyframe_imu_pos(flvis_ctx* ctx, int stream, int64_t frame_id, const double* T7, const double* imu_dq_wxyz,
                                   double imu_dt, const double* imu_dp3, const double* imu_va3, int lm_count, const int64_t* h_id,
                                   const double* h_2d, const double* h_3d, int cap, int64_t* out_frame_id, double* out_T7,
                                   int* out_lm_count, int64_t* out_lm_id, double* out_lm_3d, int* out_oc, int64_t* out_oid) {
  return ba_push_impl(ctx, stream, frame_id, T7, imu_dq_wxyz, imu_dt, lm_count, h_id, h_2d, h_3d, cap, out_frame_id, out_T7,
                      out_lm_count, out_lm_id, out_lm_3d, out_oc, out_oid, imu_dp3, imu_va3);
}

static int ba_push_impl(flvis_ctx* ctx, int stream, int64_t frame_id, const double* T7, const double* imu_dq, double imu_dt,
                        int lm_count, const int64_t* h_id, const double* h_2d, const double* h_3d, int cap, int64_t* out_frame_id,
                        double* out_T7, int* out_lm_count, int64_t* out_lm_id, double* out_lm_3d, int* out_oc, int64_t* out_oid,
                        const double* imu_dp, const double* imu_va) {
  if (!ctx || !ctx->pipe || !T7 || lm_count < 0 || lm_count > KF_MAXLM) return FLVIS_ERR_INVALID_ARG;
  Pipeline* pl = ctx->pipe;
  if (stream < 0 || stream >= pl->S) return ctx->fail(FLVIS_ERR_INVALID_ARG, "bad stream");
  int ls;
  Lane& L = pl->lane_of(stream, ls);
  std::vector<KeyFrameDev> kfv(1);
  KeyFrameDev& kf = kfv[0];
  memset(&kf, 0, sizeof(kf));
  kf.frame_id = frame_id;
  kf.lm_count = lm_count;
  kf.valid = 1;
  memcpy(kf.T_c_w, T7, 56);
  kf.imu_dq[0] = 1.0;
  if (imu_dq && imu_dt > 0) {
    memcpy(kf.imu_dq, imu_dq, 32);
    kf.imu_dt = imu_dt;
    kf.imu_valid = 1;
    if (imu_dp && imu_va) {
      memcpy(kf.imu_dp, imu_dp, 24);
      memcpy(kf.imu_va, imu_va, 24);
    }
  }
  for (int i = 0; i < lm_count; i++) {
    kf.lm_id[i] = h_id[i];
    kf.lm_2d[i][0] = h_2d[2 * i];
    kf.lm_2d[i][1] = h_2d[2 * i + 1];
    for (int j = 0; j < 3; j++) kf.lm_3d[i][j] = h_3d[3 * i + j];
  }
  sync_all(ctx);
  const int zero = 0;
  unsigned tl = 0;
  hipMemcpy(&tl, L.pipe.kfq_tail + ls, sizeof(unsigned), hipMemcpyDeviceToHost);
  hipMemcpy(L.pipe.kfq + (size_t)ls * KFQ + (tl % KFQ), &kf, sizeof(kf), hipMemcpyHostToDevice);
  tl++;
  hipMemcpy(L.pipe.kfq_tail + ls, &tl, sizeof(unsigned), hipMemcpyHostToDevice);
  hipMemcpy(&L.pipe.corr[ls].valid, &zero, sizeof(int), hipMemcpyHostToDevice);
  launch_ba_worker(pl->ba_stream[0], L.pipe, 0, ++L.ba_tag);
  hipError_t e = hipStreamSynchronize(pl->ba_stream[0]);
  if (e != hipSuccess) return ctx->hip_fail(e, "ba_push_keyframe");
  sync_all(ctx);
  return read_correction(L, ls, cap, out_frame_id, out_T7, out_lm_count, out_lm_id, out_lm_3d, out_oc, out_oid);
}

// F2FTracking::correction_feed (src/frontend/f2f_tracking.cpp:40-44) -- the sink the reference's
// correction_feedback_callback (vo_tracking.cpp:373-385) unpacks for and never calls.  Opt-in: nothing changes for a
// caller that never feeds a correction.  Applied at the stream's next Tracking frame (f2f_tracking.cpp:189-219).
int flvis_correction_feed(flvis_ctx* ctx, int stream, int64_t frame_id, const double* T_c_w7, int lm_count,
                          const int64_t* h_lm_id, const double* h_lm_3d, int lm_outlier_count,
                          const int64_t* h_lm_outlier_id) {
  if (!ctx || !ctx->pipe) return FLVIS_ERR_INVALID_ARG;
  Pipeline* pl = ctx->pipe;
  if (stream < 0 || stream >= pl->S || !T_c_w7 || lm_count < 0 || lm_outlier_count < 0 || (lm_count && (!h_lm_id || !h_lm_3d)) ||
      (lm_outlier_count && !h_lm_outlier_id))
    return ctx->fail(FLVIS_ERR_INVALID_ARG, "correction_feed: bad args");
  if (lm_count > BA_LMAX || lm_outlier_count > BA_EMAX) return ctx->fail(FLVIS_ERR_CAPACITY, "correction_feed: too many entries");
  hipSetDevice(ctx->device);
  if (!pl->feedback_used) {
    for (Lane* L : pl->lanes) {
      hipStreamSynchronize(L->st);
      L->pipe.corr_in = dalloc<CorrectionDev>(L->allocs, L->S);
      if (!L->pipe.corr_in) return ctx->fail(FLVIS_ERR_HIP, "correction_feed: device allocation failed");
    }
  }
  int ls;
  Lane& L = pl->lane_of(stream, ls);
  hipStream_t st = L.st;
  // the previous frame may still be running and reads/clears the slot: order the upload after it on the same stream
  CorrectionDev* d = L.pipe.corr_in + ls;
  struct Head {
    long long frame_id;
    int lm_count, lm_outlier_count, valid, pad;
    double T_c_w[7];
  } hd;
  static_assert(offsetof(CorrectionDev, lm_id) == sizeof(Head), "CorrectionDev header layout");
  hd.frame_id = frame_id;
  hd.lm_count = lm_count;
  hd.lm_outlier_count = lm_outlier_count;
  hd.valid = 1;
  hd.pad = 0;
  for (int j = 0; j < 7; j++) hd.T_c_w[j] = T_c_w7[j];
  hipError_t e = hipSuccess;
  if (lm_count) {
    e = hipMemcpyAsync(d->lm_id, h_lm_id, sizeof(int64_t) * lm_count, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d->lm_3d, h_lm_3d, sizeof(double) * 3 * lm_count, hipMemcpyHostToDevice, st);
  }
  if (e == hipSuccess && lm_outlier_count)
    e = hipMemcpyAsync(d->lm_outlier_id, h_lm_outlier_id, sizeof(int64_t) * lm_outlier_count, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(d, &hd, sizeof(hd), hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);  // the caller's buffers and `hd` may go away after return
  if (e != hipSuccess) return ctx->hip_fail(e, "correction_feed");
  pl->feedback_used = true;
  return FLVIS_OK;
}

// stage poses of a stream's last Tracking frame (test aid): pose7 right after PnP-RANSAC, pose7 after the pose-only LM
int flvis_debug_stage_poses(flvis_ctx* ctx, int stream, double* h_out21) {
  if (!ctx || !ctx->pipe || !h_out21) return FLVIS_ERR_INVALID_ARG;
  Pipeline* pl = ctx->pipe;
  if (stream < 0 || stream >= pl->S) return ctx->fail(FLVIS_ERR_INVALID_ARG, "bad stream");
  sync_all(ctx);
  int ls;
  Lane& L = pl->lane_of(stream, ls);
  StreamState st;
  hipMemcpy(&st, L.pipe.st + ls, sizeof(st), hipMemcpyDeviceToHost);
  memcpy(h_out21, st.dbg_T_pnp, 56);
  memcpy(h_out21 + 7, st.dbg_T_lm, 56);
  memcpy(h_out21 + 14, st.dbg_T_pre, 56);
  return FLVIS_OK;
}

// pose_records of a stream (f2f_tracking.h:59), oldest first: rows (frame_id, tx ty tz qx qy qz qw); returns the count
int flvis_get_pose_records(flvis_ctx* ctx, int stream, int cap, double* h_rows8) {
  if (!ctx || !ctx->pipe) return FLVIS_ERR_INVALID_ARG;
  Pipeline* pl = ctx->pipe;
  if (stream < 0 || stream >= pl->S || cap < 0 || (cap && !h_rows8)) return ctx->fail(FLVIS_ERR_INVALID_ARG, "get_pose_records: bad args");
  hipSetDevice(ctx->device);
  int ls;
  Lane& L = pl->lane_of(stream, ls);
  hipStreamSynchronize(ctx->stream);
  hipStreamSynchronize(L.st);
  StreamState st;
  std::vector<int> ids(POSE_REC);
  std::vector<double> T((size_t)POSE_REC * 7);
  if (hipMemcpy(&st, L.pipe.st + ls, sizeof(st), hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(ids.data(), L.pipe.rec_id + (size_t)ls * POSE_REC, sizeof(int) * POSE_REC, hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(T.data(), L.pipe.rec_T + (size_t)ls * POSE_REC * 7, sizeof(double) * 7 * POSE_REC, hipMemcpyDeviceToHost) != hipSuccess)
    return ctx->fail(FLVIS_ERR_HIP, "get_pose_records: copy failed");
  for (int i = 0; i < st.rec_count && i < cap; i++) {
    const int k = (st.rec_head + i) % POSE_REC;
    h_rows8[8 * i] = (double)ids[k];
    for (int j = 0; j < 7; j++) h_rows8[8 * i + 1 + j] = T[(size_t)k * 7 + j];
  }
  return st.rec_count;
}

}  // extern "C"
