// flvis_amd: the loop-closing nodelet's control flow around the keyframe-rate kernels -- the caller of
// orb_kernels.hip / loop_kernels.hip, for a BATCH of independent sequences on one GPU:
//
//   flvis_loop_closer_add_keyframes  <- kfmsgProcess   src/backend/vo_loopclosing.cpp:191-391   ORB, bag of words, 3-D landmarks,
//                                                      T_c_w = T_c_w_odom * T_odom_map, the keyframe appended to the sequence's map
//   flvis_loop_closer_process        <- pgoProcess     :393-518   similarity row, the `size < 50` gate, isLoopCandidate (:520-590),
//                                                      isLoopClosureKF (:593-735), the loop list, the PGO trigger (:488-497),
//                                                      loopClosureOnCovGraphG2ONew (:742-944) and T_odom_map *= Tw1_w2 (:908)
//
// The keyframe database (bag-of-words vectors, compacted ORB descriptors with their pixels and 3-D positions, T_c_w) lives in HBM
// for the whole run -- 76 KB per keyframe -- and never returns to the host; per keyframe the host sees one similarity row, and per
// candidate three integers and a pose.  Integer / threshold logic stays on the host, as in the reference's pgoProcess thread.
// That thread looks at whatever keyframe is newest whenever it comes round (a keyframe can be looked at twice or never); here
// every keyframe is processed exactly once, in order (deterministic).  tf / path / image publishing is the ROS wrapper's business.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/flvis_hip.h"
#include "ctx.hpp"

namespace {

constexpr int LCC_CAP = 1024;   // keypoints per keyframe (the reference extracts 1000) = correspondences per PnP set
constexpr int LCC_VCAP = 1024;  // bag-of-words entries per keyframe

// ---- pose7 = tx ty tz qx qy qz qw on the host (Sophus::SE3 products of :377, :908) -------------------------------------------
void q_mul(const double* a, const double* b, double* o) {  // Hamilton product, x y z w
  const double x = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  const double y = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
  const double z = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
  const double w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  o[0] = x, o[1] = y, o[2] = z, o[3] = w;
}
void q_rot(const double* q, const double* v, double* o) {  // R(q) v
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double tx = 2 * (y * v[2] - z * v[1]), ty = 2 * (z * v[0] - x * v[2]), tz = 2 * (x * v[1] - y * v[0]);
  o[0] = v[0] + w * tx + (y * tz - z * ty);
  o[1] = v[1] + w * ty + (z * tx - x * tz);
  o[2] = v[2] + w * tz + (x * ty - y * tx);
}
void pose_mul(const double* a, const double* b, double* o) {  // T_a * T_b
  double t[3], q[4];
  q_rot(a + 3, b, t);
  q_mul(a + 3, b + 3, q);
  const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  o[0] = t[0] + a[0], o[1] = t[1] + a[1], o[2] = t[2] + a[2];
  o[3] = q[0] / n, o[4] = q[1] / n, o[5] = q[2] / n, o[6] = q[3] / n;
}

// ---- device helpers ---------------------------------------------------------------------------------------------------------
// copies the batch results of one add call into the keyframe slots of their sequences (slot = stream * maxkf + keyframe)
__global__ __launch_bounds__(256) void k_lcc_store(const int* __restrict__ slot, const int* __restrict__ ids, const double* __restrict__ vals,
                                                   const int* __restrict__ nnz, const float* __restrict__ lm2, const double* __restrict__ lm3,
                                                   const uint8_t* __restrict__ lmd, const int* __restrict__ lmc, int* db_ids, double* db_vals,
                                                   int* db_nnz, float* db_lm2, double* db_lm3, uint8_t* db_lmd, int* db_lmc) {
  const int i = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;  // t: element of the keyframe's row, 0..1023
  const size_t s = (size_t)slot[i];
  if (t >= LCC_CAP) return;
  const int nv = nnz[i], nl = lmc[i];
  if (t < nv) {
    db_ids[s * LCC_VCAP + t] = ids[(size_t)i * LCC_VCAP + t];
    db_vals[s * LCC_VCAP + t] = vals[(size_t)i * LCC_VCAP + t];
  }
  if (t < nl) {
    const size_t a = (size_t)i * LCC_CAP + t, b = s * LCC_CAP + t;
    db_lm2[b * 2] = lm2[a * 2], db_lm2[b * 2 + 1] = lm2[a * 2 + 1];
    db_lm3[b * 3] = lm3[a * 3], db_lm3[b * 3 + 1] = lm3[a * 3 + 1], db_lm3[b * 3 + 2] = lm3[a * 3 + 2];
    const uint4* q = reinterpret_cast<const uint4*>(lmd + a * 32);
    uint4* r = reinterpret_cast<uint4*>(db_lmd + b * 32);
    r[0] = q[0], r[1] = q[1];
  }
  if (t == 0) db_nnz[s] = nv, db_lmc[s] = nl;
}

// descriptors of the two keyframes of every candidate pair into the contiguous arrays flvis_hip_orb_match reads
__global__ __launch_bounds__(256) void k_lcc_fetch(const int* __restrict__ slot_a, const int* __restrict__ slot_b, const uint8_t* __restrict__ db_lmd,
                                                   const int* __restrict__ db_lmc, uint8_t* a, int* na, uint8_t* b, int* nb) {
  const int i = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
  if (t >= LCC_CAP) return;
  const size_t sa = (size_t)slot_a[i], sb = (size_t)slot_b[i];
  const uint4 z = make_uint4(0, 0, 0, 0);
  const int ca = db_lmc[sa], cb = db_lmc[sb];
  const uint4* qa = reinterpret_cast<const uint4*>(db_lmd + (sa * LCC_CAP + t) * 32);
  const uint4* qb = reinterpret_cast<const uint4*>(db_lmd + (sb * LCC_CAP + t) * 32);
  uint4* oa = reinterpret_cast<uint4*>(a + ((size_t)i * LCC_CAP + t) * 32);
  uint4* ob = reinterpret_cast<uint4*>(b + ((size_t)i * LCC_CAP + t) * 32);
  oa[0] = t < ca ? qa[0] : z, oa[1] = t < ca ? qa[1] : z;
  ob[0] = t < cb ? qb[0] : z, ob[1] = t < cb ? qb[1] : z;
  if (t == 0) na[i] = ca, nb[i] = cb;
}

// cv::Point3f(kf0->lm_3d[queryIdx]), cv::Point2f(kf1->lm_2d[trainIdx]) of the selected matches (:643-652)
__global__ __launch_bounds__(256) void k_lcc_correspondences(const int* __restrict__ slot_a, const int* __restrict__ slot_b,
                                                             const int* __restrict__ pairs, const int* __restrict__ npairs,
                                                             const double* __restrict__ db_lm3, const float* __restrict__ db_lm2, float* p3d,
                                                             float* p2d) {
  const int i = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
  if (t >= LCC_CAP) return;
  const size_t o = (size_t)i * LCC_CAP + t;
  float x = 0.f, y = 0.f, z = 0.f, u = 0.f, v = 0.f;
  if (t < npairs[i]) {
    const size_t qa = (size_t)slot_a[i] * LCC_CAP + pairs[o * 2], qb = (size_t)slot_b[i] * LCC_CAP + pairs[o * 2 + 1];
    x = (float)db_lm3[qa * 3], y = (float)db_lm3[qa * 3 + 1], z = (float)db_lm3[qa * 3 + 2];
    u = db_lm2[qb * 2], v = db_lm2[qb * 2 + 1];
  }
  p3d[o * 3] = x, p3d[o * 3 + 1] = y, p3d[o * 3 + 2] = z;
  p2d[o * 2] = u, p2d[o * 2 + 1] = v;
}

struct Seq {
  int n = 0;
  bool fresh = false;
  std::vector<double> T_odom;  // 7 per keyframe
  double T_odom_map[7] = {0, 0, 0, 0, 0, 0, 1};
  std::vector<int> loop_ids;       // (earlier, later) per loop
  std::vector<double> loop_poses;  // 7 per loop
  long long last_pgo = -5000;      // :141
};

}  // namespace

struct flvis_loop_closer {
  flvis_ctx* ctx = nullptr;
  flvis_cfg cfg;
  flvis_lc_params prm;
  flvis_orb_params orb{1000, 1.2f, 8, 20};  // :242
  std::vector<int8_t> pattern;
  int S = 0, maxkf = 0, w = 0, h = 0, device = 0;
  double K4[4];
  // keyframe database, [S * maxkf] slots
  int* db_ids = nullptr;
  double* db_vals = nullptr;
  int* db_nnz = nullptr;
  float* db_lm2 = nullptr;
  double* db_lm3 = nullptr;
  uint8_t* db_lmd = nullptr;
  int* db_lmc = nullptr;
  double* db_T = nullptr;  // [S][maxkf][7] T_c_w
  // per-call staging, [S] items
  float *kps = nullptr, *lm2 = nullptr, *p3d = nullptr, *p2d = nullptr;
  uint8_t *desc = nullptr, *da = nullptr, *db = nullptr, *mask = nullptr;
  int *cnt = nullptr, *ovf = nullptr, *ids = nullptr, *nnz = nullptr, *lmc = nullptr, *slot_a = nullptr, *slot_b = nullptr, *na = nullptr,
      *nb = nullptr, *pairs = nullptr, *npairs = nullptr, *ninl = nullptr;
  double *vals = nullptr, *lm3 = nullptr, *rows = nullptr, *pose = nullptr, *loop_pose = nullptr, *drift = nullptr, *stats = nullptr, *pgo_T = nullptr;
  std::vector<void*> owned;
  std::vector<Seq> seq;
  std::vector<double> h_rows;

  template <class T>
  bool alloc(T*& p, size_t count) {
    void* q = nullptr;
    if (hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess) return false;
    owned.push_back(q);
    p = (T*)q;
    return true;
  }
};

extern "C" {

// the LC_PARAS block of the yaml (vo_loopclosing.cpp:955-963)
int flvis_lc_params_load(const char* yaml_path, flvis_lc_params* prm, char* err, int errlen) {
  auto fail = [&](const std::string& m) {
    if (err && errlen > 0) snprintf(err, errlen, "%s", m.c_str());
    return (int)FLVIS_ERR_CONFIG;
  };
  if (!yaml_path || !prm) return FLVIS_ERR_INVALID_ARG;
  std::ifstream f(yaml_path);
  if (!f) return fail(std::string("cannot open ") + yaml_path);
  struct Key {
    const char* name;
    int* i;
    double* d;
    bool seen;
  } keys[] = {{"lcKFStart", &prm->lcKFStart, nullptr, false},   {"lcKFDist", &prm->lcKFDist, nullptr, false},
              {"lcKFMaxDist", &prm->lcKFMaxDist, nullptr, false}, {"lcKFLast", &prm->lcKFLast, nullptr, false},
              {"lcNKFClosest", &prm->lcNKFClosest, nullptr, false}, {"minPts", &prm->minPts, nullptr, false},
              {"ratioMax", nullptr, &prm->ratioMax, false},     {"ratioRansac", nullptr, &prm->ratioRansac, false},
              {"minScore", nullptr, &prm->minScore, false}};
  std::string line;
  while (std::getline(f, line)) {
    const size_t hash = line.find('#');
    if (hash != std::string::npos) line.resize(hash);
    const size_t colon = line.find(':');
    if (colon == std::string::npos) continue;
    std::string k = line.substr(0, colon);
    k.erase(0, k.find_first_not_of(" \t"));
    k.erase(k.find_last_not_of(" \t") + 1);
    for (Key& key : keys)
      if (k == key.name) {
        std::istringstream is(line.substr(colon + 1));
        double v;
        if (!(is >> v)) return fail(std::string("yaml key ") + key.name + " has no number");
        if (key.i) *key.i = (int)v;
        if (key.d) *key.d = v;
        key.seen = true;
      }
  }
  for (const Key& key : keys)
    if (!key.seen) return fail(std::string("yaml key ") + key.name + " is missing (loop-closing parameters)");
  return FLVIS_OK;
}

int flvis_loop_closer_create(flvis_ctx* ctx, const flvis_cfg* cfg, const flvis_lc_params* prm, int n_streams, int max_keyframes,
                             const int8_t* h_orb_pattern, flvis_loop_closer** out) {
  if (!ctx) return FLVIS_ERR_INVALID_ARG;
  if (!cfg || !prm || !out || n_streams <= 0 || max_keyframes <= 0) return ctx->fail(FLVIS_ERR_INVALID_ARG, "loop_closer_create: bad args");
  *out = nullptr;
  if (ctx->voc_nodes < 2)
    return ctx->fail(FLVIS_ERR_CONFIG, "loop_closer_create: no vocabulary (flvis_hip_bow_load_vocabulary / flvis_hip_bow_set_vocabulary first)");
  if ((long long)n_streams * max_keyframes > (1ll << 31) / LCC_CAP)
    return ctx->fail(FLVIS_ERR_CAPACITY, "loop_closer_create: n_streams * max_keyframes is too large");
  hipSetDevice(ctx->device);
  flvis_loop_closer* lc = new flvis_loop_closer();
  lc->ctx = ctx, lc->device = ctx->device, lc->cfg = *cfg, lc->prm = *prm, lc->S = n_streams, lc->maxkf = max_keyframes;
  lc->w = cfg->image_width, lc->h = cfg->image_height;
  lc->K4[0] = cfg->P0[0], lc->K4[1] = cfg->P0[5], lc->K4[2] = cfg->P0[2], lc->K4[3] = cfg->P0[6];  // dc.K0_rect (:670)
  if (h_orb_pattern) lc->pattern.assign(h_orb_pattern, h_orb_pattern + 1024);
  const size_t slots = (size_t)n_streams * max_keyframes, S = (size_t)n_streams;
  bool ok = lc->alloc(lc->db_ids, slots * LCC_VCAP) && lc->alloc(lc->db_vals, slots * LCC_VCAP) && lc->alloc(lc->db_nnz, slots) &&
            lc->alloc(lc->db_lm2, slots * LCC_CAP * 2) && lc->alloc(lc->db_lm3, slots * LCC_CAP * 3) &&
            lc->alloc(lc->db_lmd, slots * LCC_CAP * 32) && lc->alloc(lc->db_lmc, slots) && lc->alloc(lc->db_T, slots * 7) &&
            lc->alloc(lc->kps, S * LCC_CAP * 6) && lc->alloc(lc->desc, S * LCC_CAP * 32) && lc->alloc(lc->cnt, S) && lc->alloc(lc->ovf, S) &&
            lc->alloc(lc->ids, S * LCC_VCAP) && lc->alloc(lc->vals, S * LCC_VCAP) && lc->alloc(lc->nnz, S) &&
            lc->alloc(lc->lm2, S * LCC_CAP * 2) && lc->alloc(lc->lm3, S * LCC_CAP * 3) && lc->alloc(lc->lmc, S) &&
            lc->alloc(lc->slot_a, S) && lc->alloc(lc->slot_b, S) && lc->alloc(lc->da, S * LCC_CAP * 32) &&
            lc->alloc(lc->db, S * LCC_CAP * 32) && lc->alloc(lc->na, S) && lc->alloc(lc->nb, S) && lc->alloc(lc->pairs, S * LCC_CAP * 2) &&
            lc->alloc(lc->npairs, S) && lc->alloc(lc->p3d, S * LCC_CAP * 3) && lc->alloc(lc->p2d, S * LCC_CAP * 2) &&
            lc->alloc(lc->mask, S * LCC_CAP) && lc->alloc(lc->ninl, S) && lc->alloc(lc->pose, S * 7) && lc->alloc(lc->rows, slots) &&
            lc->alloc(lc->loop_pose, slots * 7) && lc->alloc(lc->drift, S * 7) && lc->alloc(lc->stats, S * 5) && lc->alloc(lc->pgo_T, slots * 7);
  if (!ok) {
    for (void* p : lc->owned) hipFree(p);
    delete lc;
    return ctx->fail(FLVIS_ERR_HIP, "loop_closer_create: device allocation failed (76 KB per keyframe slot)");
  }
  lc->seq.resize(n_streams);
  lc->h_rows.resize(slots);
  *out = lc;
  return FLVIS_OK;
}

void flvis_loop_closer_destroy(flvis_loop_closer* lc) {
  if (!lc) return;
  hipSetDevice(lc->device);  // (the context may already be gone: nothing of it is touched here)
  hipDeviceSynchronize();
  for (void* p : lc->owned) hipFree(p);
  delete lc;
}

int flvis_loop_closer_add_keyframes(flvis_loop_closer* lc, int n, const int* h_stream, const uint8_t* d_img0, const void* d_img1,
                                    const double* h_T_c_w_odom7, int64_t* h_kf_id) {
  if (!lc) return FLVIS_ERR_INVALID_ARG;
  flvis_ctx* ctx = lc->ctx;
  if (n <= 0 || n > lc->S || !h_stream || !d_img0 || !h_T_c_w_odom7) return ctx->fail(FLVIS_ERR_INVALID_ARG, "loop_closer_add_keyframes: bad args");
  std::vector<char> used((size_t)lc->S, 0);
  for (int i = 0; i < n; i++) {
    const int s = h_stream[i];
    if (s < 0 || s >= lc->S || used[s]) return ctx->fail(FLVIS_ERR_INVALID_ARG, "loop_closer_add_keyframes: one keyframe per sequence and call");
    used[s] = 1;
    if (lc->seq[s].n >= lc->maxkf) return ctx->fail(FLVIS_ERR_CAPACITY, "loop_closer_add_keyframes: a sequence's keyframe capacity is used up");
  }
  hipSetDevice(ctx->device);
  hipStream_t st = ctx->stream;
  // STEP 1.3 / 1.4 / 1.5 / 1.6 (:236-372): ORB, bag of words of ALL descriptors, 3-D positions, then the lists without the rest
  int rc = flvis_hip_orb_detect_and_compute(ctx, d_img0, lc->w, lc->h, n, &lc->orb, lc->pattern.empty() ? nullptr : lc->pattern.data(), lc->kps,
                                            lc->desc, lc->cnt, LCC_CAP, lc->ovf);
  if (rc != FLVIS_OK) return rc;
  rc = flvis_hip_bow_transform(ctx, lc->desc, lc->cnt, LCC_CAP, n, LCC_VCAP, lc->ids, lc->vals, lc->nnz);
  if (rc != FLVIS_OK) return rc;
  rc = flvis_hip_lc_keyframe_landmarks(ctx, d_img0, d_img1, lc->w, lc->h, n, lc->cfg.cam_type, lc->cfg.P0, lc->cfg.P1, lc->K4, lc->kps, lc->desc,
                                       lc->cnt, LCC_CAP, lc->lm2, lc->lm3, lc->desc, lc->lmc);
  if (rc != FLVIS_OK) return rc;
  // STEP 2 (:374-383): the keyframe joins its sequence's map with T_c_w = T_c_w_odom * T_odom_map
  std::vector<int> slot((size_t)n);
  std::vector<double> T((size_t)n * 7);
  for (int i = 0; i < n; i++) {
    Seq& q = lc->seq[h_stream[i]];
    slot[i] = h_stream[i] * lc->maxkf + q.n;
    pose_mul(h_T_c_w_odom7 + 7 * i, q.T_odom_map, &T[7 * (size_t)i]);
  }
  hipError_t e = hipMemcpyAsync(lc->slot_a, slot.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice, st);
  for (int i = 0; i < n && e == hipSuccess; i++)
    e = hipMemcpyAsync(lc->db_T + (size_t)slot[i] * 7, &T[7 * (size_t)i], 7 * sizeof(double), hipMemcpyHostToDevice, st);
  if (e != hipSuccess) return ctx->hip_fail(e, "loop_closer_add_keyframes");
  k_lcc_store<<<dim3(LCC_CAP / 256, n), 256, 0, st>>>(lc->slot_a, lc->ids, lc->vals, lc->nnz, lc->lm2, lc->lm3, lc->desc, lc->lmc, lc->db_ids,
                                                       lc->db_vals, lc->db_nnz, lc->db_lm2, lc->db_lm3, lc->db_lmd, lc->db_lmc);
  e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(st);  // slot / T are host temporaries of this call
  if (e != hipSuccess) return ctx->hip_fail(e, "loop_closer_add_keyframes");
  for (int i = 0; i < n; i++) {
    Seq& q = lc->seq[h_stream[i]];
    q.T_odom.insert(q.T_odom.end(), h_T_c_w_odom7 + 7 * i, h_T_c_w_odom7 + 7 * i + 7);
    if (h_kf_id) h_kf_id[i] = q.n;
    q.n++;
    q.fresh = true;
  }
  return FLVIS_OK;
}

// KeyFrameMsg::unpack (:206) hands the nodelet HOST images; this is the same call on host buffers: mono8 img0, mono8 or 16UC1 img1
int flvis_loop_closer_add_keyframes_host(flvis_loop_closer* lc, int n, const int* h_stream, const flvis_image* h_img0, const flvis_image* h_img1,
                                         const double* h_T_c_w_odom7, int64_t* h_kf_id) {
  if (!lc) return FLVIS_ERR_INVALID_ARG;
  flvis_ctx* ctx = lc->ctx;
  if (n <= 0 || n > lc->S || !h_img0 || !h_img1 || !h_stream || !h_T_c_w_odom7)
    return ctx->fail(FLVIS_ERR_INVALID_ARG, "loop_closer_add_keyframes_host: bad args");
  const int bpp1 = lc->cfg.cam_type == 2 ? 2 : 1;
  // every argument is checked BEFORE a copy is queued: the caller may free its images as soon as this call returns with an error
  for (int i = 0; i < n; i++) {
    const flvis_image &a = h_img0[i], &b = h_img1[i];
    if (h_stream[i] < 0 || h_stream[i] >= lc->S) return ctx->fail(FLVIS_ERR_INVALID_ARG, "loop_closer_add_keyframes_host: bad stream index");
    if (!a.data || !b.data || a.width != lc->w || a.height != lc->h || b.width != lc->w || b.height != lc->h || a.channels != 1 || b.channels != 1 ||
        a.pitch < lc->w || b.pitch < lc->w * bpp1)
      return ctx->fail(FLVIS_ERR_INVALID_ARG, "loop_closer_add_keyframes_host: images must be mono8 (img1: 16UC1 on a depth rig) of the configured size");
  }
  const size_t px = (size_t)lc->w * lc->h;
  hipSetDevice(ctx->device);
  uint8_t* d0 = (uint8_t*)ctx->scratch("lc_host_img0", px * (size_t)lc->S);
  uint8_t* d1 = (uint8_t*)ctx->scratch("lc_host_img1", px * 2 * (size_t)lc->S);
  if (!d0 || !d1) return ctx->fail(FLVIS_ERR_HIP, "loop_closer_add_keyframes_host: staging allocation failed");
  hipError_t e = hipSuccess;
  for (int i = 0; i < n && e == hipSuccess; i++) {
    const flvis_image &a = h_img0[i], &b = h_img1[i];
    e = hipMemcpy2DAsync(d0 + px * i, (size_t)lc->w, a.data, (size_t)a.pitch, (size_t)lc->w, (size_t)lc->h, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess)
      e = hipMemcpy2DAsync(d1 + px * bpp1 * i, (size_t)lc->w * bpp1, b.data, (size_t)b.pitch, (size_t)lc->w * bpp1, (size_t)lc->h,
                           hipMemcpyHostToDevice, ctx->stream);
  }
  if (e != hipSuccess) {
    hipStreamSynchronize(ctx->stream);  // copies from the caller's images may still be in flight
    return ctx->hip_fail(e, "loop_closer_add_keyframes_host");
  }
  const int rc = flvis_loop_closer_add_keyframes(lc, n, h_stream, d0, d1, h_T_c_w_odom7, h_kf_id);  // (synchronises when it succeeds)
  if (rc != FLVIS_OK) hipStreamSynchronize(ctx->stream);  // ... and on its error paths the uploads are waited for here
  return rc;
}

int flvis_loop_closer_process(flvis_loop_closer* lc, flvis_lc_event* h_events) {
  if (!lc) return FLVIS_ERR_INVALID_ARG;
  flvis_ctx* ctx = lc->ctx;
  if (!h_events) return ctx->fail(FLVIS_ERR_INVALID_ARG, "loop_closer_process: no event array");
  hipSetDevice(ctx->device);
  hipStream_t st = ctx->stream;
  const flvis_lc_params& p = lc->prm;
  for (int s = 0; s < lc->S; s++) {
    flvis_lc_event& ev = h_events[s];
    memset(&ev, 0, sizeof(ev));
    ev.kf_prev = -1;
    ev.kf_curr = lc->seq[s].fresh ? lc->seq[s].n - 1 : -1;
    ev.loop_pose7[6] = 1.0;
  }
  // STEP 3 (:417-437): the newest keyframe of every sequence that got one against all keyframes of that sequence -- one launch for all
  // sequences, one strided copy of the rows
  std::vector<int> jobs;
  int max_n = 0;
  for (int s = 0; s < lc->S; s++) {
    const Seq& q = lc->seq[s];
    if (!q.fresh) continue;
    const int base = s * lc->maxkf;
    jobs.push_back(base + q.n - 1);
    jobs.push_back(base);
    jobs.push_back(q.n);
    max_n = std::max(max_n, q.n);
  }
  if (jobs.empty()) return FLVIS_OK;
  int rc0 = flvis_hip_bow_score_jobs(ctx, (int)(jobs.size() / 3), jobs.data(), lc->db_ids, lc->db_vals, lc->db_nnz, LCC_VCAP, lc->rows);
  if (rc0 != FLVIS_OK) return rc0;
  hipError_t e = hipMemcpy2DAsync(lc->h_rows.data(), sizeof(double) * (size_t)lc->maxkf, lc->rows, sizeof(double) * (size_t)lc->maxkf,
                                  sizeof(double) * (size_t)max_n, (size_t)lc->S, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) return ctx->hip_fail(e, "loop_closer_process");
  // :453 + isLoopCandidate (:520-590) on the host
  std::vector<int> cand;  // sequences with a candidate
  std::vector<int> sa, sb;
  for (int s = 0; s < lc->S; s++) {
    Seq& q = lc->seq[s];
    if (!q.fresh) continue;
    q.fresh = false;
    if (q.n < 50) continue;
    const std::vector<uint8_t> present((size_t)q.n, 1);
    int64_t prev = -1;
    const int r = flvis_loop_candidate(q.n, &lc->h_rows[(size_t)s * lc->maxkf], present.data(), p.lcKFDist, p.lcKFMaxDist, p.lcNKFClosest, p.minScore,
                                       &prev);
    if (r != 1) continue;
    h_events[s].candidate = 1;
    h_events[s].kf_prev = prev;
    cand.push_back(s);
    sa.push_back(s * lc->maxkf + (int)prev);
    sb.push_back(s * lc->maxkf + q.n - 1);
  }
  const int nc = (int)cand.size();
  if (nc == 0) return FLVIS_OK;
  // isLoopClosureKF (:593-686) for all candidates at once
  e = hipMemcpyAsync(lc->slot_a, sa.data(), sizeof(int) * (size_t)nc, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(lc->slot_b, sb.data(), sizeof(int) * (size_t)nc, hipMemcpyHostToDevice, st);
  if (e != hipSuccess) return ctx->hip_fail(e, "loop_closer_process");
  k_lcc_fetch<<<dim3(LCC_CAP / 256, nc), 256, 0, st>>>(lc->slot_a, lc->slot_b, lc->db_lmd, lc->db_lmc, lc->da, lc->na, lc->db, lc->nb);
  int rc = flvis_hip_orb_match(ctx, lc->da, lc->na, LCC_CAP, lc->db, lc->nb, LCC_CAP, nc, p.ratioMax, lc->pairs, lc->npairs);
  if (rc != FLVIS_OK) return rc;
  k_lcc_correspondences<<<dim3(LCC_CAP / 256, nc), 256, 0, st>>>(lc->slot_a, lc->slot_b, lc->pairs, lc->npairs, lc->db_lm3, lc->db_lm2, lc->p3d,
                                                                  lc->p2d);
  std::vector<uint64_t> seeds((size_t)nc);
  for (int i = 0; i < nc; i++) seeds[i] = ((uint64_t)(cand[i] + 1) << 32) + (uint64_t)lc->seq[cand[i]].n;  // (stream + 1) << 32 | kf_curr + 1
  rc = flvis_hip_pnp_ransac(ctx, lc->p3d, lc->p2d, lc->npairs, LCC_CAP, nc, lc->K4, 100, 2.0, 0.99, seeds.data(), lc->pose, lc->mask, lc->ninl);
  if (rc != FLVIS_OK) return rc;
  std::vector<int> h_np((size_t)nc), h_ni((size_t)nc);
  std::vector<double> h_pose((size_t)nc * 7);
  e = hipMemcpyAsync(h_np.data(), lc->npairs, sizeof(int) * (size_t)nc, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipMemcpyAsync(h_ni.data(), lc->ninl, sizeof(int) * (size_t)nc, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipMemcpyAsync(h_pose.data(), lc->pose, sizeof(double) * 7 * (size_t)nc, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) return ctx->hip_fail(e, "loop_closer_process");
  std::vector<int> pgo;  // sequences whose pose graph is due (:492-497)
  for (int i = 0; i < nc; i++) {
    const int s = cand[i];
    Seq& q = lc->seq[s];
    flvis_lc_event& ev = h_events[s];
    const int m = h_np[i], inl = h_ni[i];
    const double* T = &h_pose[7 * (size_t)i];
    ev.n_matches = m;
    if (m < 5) continue;  // "p3d not enough" (:666)
    ev.n_inliers = inl;
    memcpy(ev.loop_pose7, T, 7 * sizeof(double));
    if (inl * 1.0 / m < p.ratioRansac || inl < p.minPts) continue;  // :677
    const double tn = std::sqrt(T[0] * T[0] + T[1] * T[1] + T[2] * T[2]);
    const double vn = std::sqrt(T[3] * T[3] + T[4] * T[4] + T[5] * T[5]);
    const double angle = 2.0 * std::atan2(vn, std::fabs(T[6]));  // |so3().log()|
    if (!(tn < 3 && angle < 1.5)) continue;                        // :686
    ev.loop_accepted = 1;
    q.loop_ids.push_back((int)ev.kf_prev);
    q.loop_ids.push_back(q.n - 1);
    q.loop_poses.insert(q.loop_poses.end(), T, T + 7);
    const int thre = (int)(((double)q.n / 100) * 2);  // :490
    if ((long long)(q.n - 1) - q.last_pgo > thre) {
      pgo.push_back(s);
      q.last_pgo = q.n - 1;
    }
  }
  // loopClosureOnCovGraphG2ONew (:742-944) for every sequence that asked for it, ONE launch (one workgroup per pose graph): the
  // sequences' pose arrays are gathered into one contiguous batch, optimised, and copied back
  const int ng = (int)pgo.size();
  if (ng == 0) return FLVIS_OK;
  std::vector<int> n_kf((size_t)ng), n_loops((size_t)ng), ids, ran((size_t)ng, 0);
  std::vector<uint8_t> present;
  std::vector<double> lp;
  size_t off = 0;
  e = hipSuccess;
  for (int g = 0; g < ng; g++) {
    const Seq& q = lc->seq[pgo[g]];
    n_kf[g] = q.n;
    n_loops[g] = (int)(q.loop_ids.size() / 2);
    ids.insert(ids.end(), q.loop_ids.begin(), q.loop_ids.end());
    lp.insert(lp.end(), q.loop_poses.begin(), q.loop_poses.end());
    present.insert(present.end(), (size_t)q.n, 1);
    if (e == hipSuccess)
      e = hipMemcpyAsync(lc->pgo_T + off * 7, lc->db_T + (size_t)pgo[g] * lc->maxkf * 7, sizeof(double) * 7 * (size_t)q.n, hipMemcpyDeviceToDevice, st);
    off += (size_t)q.n;
  }
  if (e == hipSuccess) e = hipMemcpyAsync(lc->loop_pose, lp.data(), sizeof(double) * lp.size(), hipMemcpyHostToDevice, st);
  if (e != hipSuccess) return ctx->hip_fail(e, "loop_closer_process");
  rc = flvis_hip_pgo_loop_closure(ctx, ng, n_kf.data(), lc->pgo_T, present.data(), n_loops.data(), ids.data(), lc->loop_pose, 100, 1, lc->drift,
                                  lc->stats, ran.data());
  if (rc != FLVIS_OK) return rc;
  std::vector<double> drift((size_t)ng * 7), stats((size_t)ng * 5);
  off = 0;
  for (int g = 0; g < ng && e == hipSuccess; g++) {
    if (ran[g])
      e = hipMemcpyAsync(lc->db_T + (size_t)pgo[g] * lc->maxkf * 7, lc->pgo_T + off * 7, sizeof(double) * 7 * (size_t)n_kf[g], hipMemcpyDeviceToDevice, st);
    off += (size_t)n_kf[g];
  }
  if (e == hipSuccess) e = hipMemcpyAsync(drift.data(), lc->drift, sizeof(double) * drift.size(), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipMemcpyAsync(stats.data(), lc->stats, sizeof(double) * stats.size(), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) return ctx->hip_fail(e, "loop_closer_process");
  for (int g = 0; g < ng; g++) {
    if (!ran[g]) continue;
    Seq& q = lc->seq[pgo[g]];
    flvis_lc_event& ev = h_events[pgo[g]];
    double m2[7];
    pose_mul(q.T_odom_map, &drift[7 * (size_t)g], m2);  // T_odom_map = T_odom_map * Tw1_w2 (:908)
    memcpy(q.T_odom_map, m2, sizeof(m2));
    ev.optimised = 1;
    ev.pgo_iterations = (int)stats[5 * (size_t)g];
    ev.chi2_before = stats[5 * (size_t)g + 1];
    ev.chi2_after = stats[5 * (size_t)g + 2];
    // (:922-925 re-derives the keyframes BEHIND the last optimised one from their odometry pose; the newest keyframe is the last
    //  optimised one here, so there is none)
  }
  return FLVIS_OK;
}

int flvis_loop_closer_poses(flvis_loop_closer* lc, int stream, double* h_T_c_w7, int cap, int* n_out) {
  if (!lc) return FLVIS_ERR_INVALID_ARG;
  flvis_ctx* ctx = lc->ctx;
  if (stream < 0 || stream >= lc->S || !h_T_c_w7 || !n_out || cap < 0) return ctx->fail(FLVIS_ERR_INVALID_ARG, "loop_closer_poses: bad args");
  const int n = std::min(cap, lc->seq[stream].n);
  *n_out = lc->seq[stream].n;
  if (n == 0) return FLVIS_OK;
  hipSetDevice(ctx->device);
  hipError_t e = hipMemcpyAsync(h_T_c_w7, lc->db_T + (size_t)stream * lc->maxkf * 7, sizeof(double) * 7 * (size_t)n, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) return ctx->hip_fail(e, "loop_closer_poses");
  return FLVIS_OK;
}

// one keyframe of the database back on the host (KeyFrameLC: lm_2d / lm_3d / lm_descriptor / kf_bv, :100-112); any output may be NULL
int flvis_loop_closer_keyframe(flvis_loop_closer* lc, int stream, int kf, int cap, float* h_lm_2d, double* h_lm_3d, uint8_t* h_lm_desc,
                               int* lm_count, int* h_bow_ids, double* h_bow_vals, int* bow_count) {
  if (!lc) return FLVIS_ERR_INVALID_ARG;
  flvis_ctx* ctx = lc->ctx;
  if (stream < 0 || stream >= lc->S || kf < 0 || kf >= lc->seq[stream].n || cap < 0)
    return ctx->fail(FLVIS_ERR_INVALID_ARG, "loop_closer_keyframe: no such keyframe");
  hipSetDevice(ctx->device);
  hipStream_t st = ctx->stream;
  const size_t slot = (size_t)stream * lc->maxkf + kf;
  int cnt[2] = {0, 0};
  hipError_t e = hipMemcpyAsync(&cnt[0], lc->db_lmc + slot, sizeof(int), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipMemcpyAsync(&cnt[1], lc->db_nnz + slot, sizeof(int), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) return ctx->hip_fail(e, "loop_closer_keyframe");
  const size_t nl = (size_t)std::min(cnt[0], cap), nv = (size_t)std::min(cnt[1], cap);
  if (h_lm_2d && nl) e = hipMemcpyAsync(h_lm_2d, lc->db_lm2 + slot * LCC_CAP * 2, sizeof(float) * 2 * nl, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess && h_lm_3d && nl) e = hipMemcpyAsync(h_lm_3d, lc->db_lm3 + slot * LCC_CAP * 3, sizeof(double) * 3 * nl, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess && h_lm_desc && nl) e = hipMemcpyAsync(h_lm_desc, lc->db_lmd + slot * LCC_CAP * 32, 32 * nl, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess && h_bow_ids && nv) e = hipMemcpyAsync(h_bow_ids, lc->db_ids + slot * LCC_VCAP, sizeof(int) * nv, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess && h_bow_vals && nv) e = hipMemcpyAsync(h_bow_vals, lc->db_vals + slot * LCC_VCAP, sizeof(double) * nv, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) return ctx->hip_fail(e, "loop_closer_keyframe");
  if (lm_count) *lm_count = cnt[0];
  if (bow_count) *bow_count = cnt[1];
  return FLVIS_OK;
}

int flvis_loop_closer_drift(flvis_loop_closer* lc, int stream, double* h_T_odom_map7) {
  if (!lc) return FLVIS_ERR_INVALID_ARG;
  if (stream < 0 || stream >= lc->S || !h_T_odom_map7) return lc->ctx->fail(FLVIS_ERR_INVALID_ARG, "loop_closer_drift: bad args");
  memcpy(h_T_odom_map7, lc->seq[stream].T_odom_map, 7 * sizeof(double));
  return FLVIS_OK;
}

int flvis_loop_closer_similarity_row(flvis_loop_closer* lc, int stream, double* h_row, int cap, int* n_out) {
  if (!lc) return FLVIS_ERR_INVALID_ARG;
  if (stream < 0 || stream >= lc->S || !h_row || !n_out || cap < 0) return lc->ctx->fail(FLVIS_ERR_INVALID_ARG, "loop_closer_similarity_row: bad args");
  const int n = lc->seq[stream].n;
  *n_out = n;
  memcpy(h_row, &lc->h_rows[(size_t)stream * lc->maxkf], sizeof(double) * (size_t)std::min(n, cap));
  return FLVIS_OK;
}
}
