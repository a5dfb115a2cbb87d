// flvis_amd: local-map bookkeeping in front of the optimiser (included by ba_solve.hip inside namespace flvis).
//
// LocalMapNodeletClass::frame_callback up to (not including) the optimisation (src/backend/vo_localmap.cpp:114-284) with
// PoseLMBag (src/backend/poselmbag.cpp): keyframe queue, bag of poses / landmarks, the g2o graph's vertices and edges.
// Runs on the whole workgroup of the local-map worker: keyframe landmarks, bag landmarks and edges are handled one per
// thread, every compaction is order preserving (block_rank).  The bag's landmark-id list is staged in LDS.
#pragma once

// ------------------------------------------------------------------------------------------------ bookkeeping
// landmark id -> bag index; the id list of the bag is staged in LDS (sid) by ba_update_dev and kept in sync with appends
// (eight entries per trip, all eight reads in flight before the first comparison: one LDS round trip per eight ids instead of one per id --
// the loop with its early exit waited for every read.  Ids are unique in the bag: any match is the first.  The list lies at the start of
// the worker's dynamic LDS: the up to seven entries read past n are inside the allocation and are not looked at.)
FD int bag_find(const long long* sid, int n, long long id) {
  for (int i = 0; i < n; i += 8) {
    long long v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = sid[i + k];
    int hit = -1;
#pragma unroll
    for (int k = 7; k >= 0; k--)
      if (v[k] == id && i + k < n) hit = i + k;
    if (hit >= 0) return hit;
  }
  return -1;
}

FD void bag_add_pose(WindowDev& w, int W, long long frame_id, const double* pose7) {  // poselmbag.cpp:110-136
  if (w.initialized) {
    w.newest = w.oldest;
    w.pose_frame_id[w.newest] = frame_id;
    for (int j = 0; j < 7; j++) w.bag_pose[w.newest][j] = pose7[j];
    w.oldest++;
    if (w.oldest == W) w.oldest = 0;
  } else {
    w.pose_frame_id[w.wp_init] = frame_id;
    for (int j = 0; j < 7; j++) w.bag_pose[w.wp_init][j] = pose7[j];
    w.wp_init++;
    if (w.wp_init == W) {
      w.initialized = 1;
      w.oldest = 0;
      w.newest = W - 1;
    }
  }
}

constexpr int BU_T = BA_T;  // the bookkeeping runs on the worker's workgroup
constexpr int BU_NW = BU_T / 64;

// removes edges flagged by pred (order preserving, in place); whole workgroup
template <typename Pred>
__device__ inline void edges_remove_if(WindowDev& w, int* s_cnt, Pred pred) {
  const int tid = threadIdx.x;
  const int n = w.n_edge;
  int kept = 0;
  for (int base = 0; base < n; base += BU_T) {
    const int i = base + tid;
    const bool keep = i < n && !pred(i);
    long long id = 0, lm = 0;
    int ps = 0, li = 0;
    double u = 0, v = 0;
    if (keep) {
      id = w.e_id[i];
      lm = w.e_lm[i];
      ps = w.e_pose[i];
      li = w.e_lidx[i];
      u = w.e_uv[i][0];
      v = w.e_uv[i][1];
    }
    int tot;
    const int rk = block_rank<BU_NW>(keep, s_cnt, tot);  // (its barriers separate this pass's reads from its writes)
    if (keep) {
      const int k = kept + rk;
      w.e_id[k] = id;
      w.e_lm[k] = lm;
      w.e_pose[k] = ps;
      w.e_lidx[k] = li;
      w.e_uv[k][0] = u;
      w.e_uv[k][1] = v;
    }
    kept += tot;
  }
  __syncthreads();
  if (tid == 0) w.n_edge = kept;
  __syncthreads();
}

// adds the observations of one keyframe to the bag (init: running mean, sliding: count only) and, if slot >= 0, the
// projection edges to that pose slot.  New landmarks are appended in keyframe order.  One keyframe landmark per thread
// and pass.
__device__ inline void bag_add_keyframe(WindowDev& w, long long* sid, int* s_cnt, const KeyFrameDev& kf, bool sliding, int slot) {
  const int tid = threadIdx.x;
  const int n = kf.lm_count;
  const int e0 = w.n_edge, nl0 = w.n_lm;
  int appended = 0;
  for (int base = 0; base < n; base += BU_T) {
    const int i = base + tid;
    int found = -1;
    bool isnew = false;
    long long id = 0;
    if (i < n) {
      id = kf.lm_id[i];
      found = bag_find(sid, nl0, id);  // (ids are unique inside a keyframe: entries appended below cannot match)
      isnew = found < 0;
    }
    int nnew;
    const int rk = block_rank<BU_NW>(isnew, s_cnt, nnew);
    if (i < n) {
      int li;
      if (isnew) {
        const int k = nl0 + appended + rk;
        li = k;
        if (k < BA_LMAX) {
          w.lm_id[k] = id;
          sid[k] = id;
          w.lm_count[k] = 1;
          for (int j = 0; j < 3; j++) {
            w.lm_p3d[k][j] = kf.lm_3d[i][j];
            w.lm_est[k][j] = kf.lm_3d[i][j];
          }
        }
      } else {
        li = found;
        const int cnt = w.lm_count[found];
        if (!sliding) {  // PoseLMBag::addLMObservation: running mean (poselmbag.cpp:69-91)
          for (int j = 0; j < 3; j++) {
            double pj = (double)cnt * w.lm_p3d[found][j] + kf.lm_3d[i][j];
            w.lm_p3d[found][j] = (1.0 / (double)(cnt + 1)) * pj;
          }
        }
        w.lm_count[found] = cnt + 1;
      }
      if (slot >= 0) {
        const int k = e0 + i;
        if (k < BA_EMAX) {
          w.e_id[k] = w.edge_next_id + i;
          w.e_lm[k] = id;
          w.e_pose[k] = slot;
          w.e_lidx[k] = li < BA_LMAX ? li : 0;
          w.e_uv[k][0] = kf.lm_2d[i][0];
          w.e_uv[k][1] = kf.lm_2d[i][1];
        }
      }
    }
    appended += nnew;
  }
  __syncthreads();
  if (tid == 0) {
    int nn = nl0 + appended;
    if (nn > BA_LMAX) {
      nn = BA_LMAX;
      w.overflow = 1;
    }
    w.n_lm = nn;
    if (slot >= 0) {
      int ne = e0 + n;
      if (ne > BA_EMAX) {
        ne = BA_EMAX;
        w.overflow = 1;
      }
      w.n_edge = ne;
      w.edge_next_id += n;
    }
  }
  __syncthreads();
}

FD void pose_to_g2o(const double* pose7, double* out7) {
  SE3d T = load_pose7(pose7);
  store_pose7(out7, g2o_from_mat(q_to_mat(T.q), T.t));
}

// LocalMapNodeletClass::frame_callback up to (not including) the optimisation for the keyframe `src`.  sid: LDS scratch
// for the bag's landmark ids (BA_LMAX entries), s_cnt: LDS [BU_NW].  Sets w.solve when the optimiser has to run.
__device__ __noinline__ void ba_update_dev(const Pipe& p, int s, const KeyFrameDev& src, long long* sid, int* s_cnt) {
  StreamState& st = p.st[s];
  WindowDev& w = p.win[s];
  const int tid = threadIdx.x;
  if (tid == 0) w.solve = 0;
  __syncthreads();
  const int W = p.cam.window;
  KeyFrameDev* ring = p.kfs_ring + (size_t)s * BA_WMAX;
  for (int i = tid; i < w.n_lm; i += BU_T) sid[i] = w.lm_id[i];
  {  // kfs.push_back(kf)
    KeyFrameDev& dst = ring[(w.kfs_head + w.kfs_size) % W];
    const int n = src.lm_count;
    unsigned hsum = 0u;
    for (int i = tid; i < n; i += BU_T) {
      const long long id = src.lm_id[i];
      const double p2[2] = {src.lm_2d[i][0], src.lm_2d[i][1]}, p3[3] = {src.lm_3d[i][0], src.lm_3d[i][1], src.lm_3d[i][2]};
      dst.lm_id[i] = id;
      dst.lm_2d[i][0] = p2[0];
      dst.lm_2d[i][1] = p2[1];
      dst.lm_3d[i][0] = p3[0];
      dst.lm_3d[i][1] = p3[1];
      dst.lm_3d[i][2] = p3[2];
      if (p.kf_check) hsum += kf_entry_hash(i, id, p2, p3);
    }
    if (p.kf_check) {  // (test knob, FLVIS_KF_CHECK: the payload as this workgroup sees it against the checksum its producer left)
      __syncthreads();
      if (tid == 0) s_cnt[0] = 0;
      __syncthreads();
      if (hsum) atomicAdd(reinterpret_cast<unsigned*>(&s_cnt[0]), hsum);
      __syncthreads();
      if (tid == 0 && p.counters) {
        atomicAdd((unsigned long long*)&p.counters[30], 1ull);
        if ((int)(unsigned)s_cnt[0] != src.imu_pad) atomicAdd((unsigned long long*)&p.counters[31], 1ull);
      }
    }
    __syncthreads();
    if (tid == 0) {
      dst.frame_id = src.frame_id;
      dst.lm_count = n;
      dst.valid = 1;
      for (int j = 0; j < 7; j++) dst.T_c_w[j] = src.T_c_w[j];
      for (int j = 0; j < 4; j++) dst.imu_dq[j] = src.imu_dq[j];
      dst.imu_dt = src.imu_dt;
      dst.imu_valid = src.imu_valid;
      for (int j = 0; j < 3; j++) dst.imu_dp[j] = src.imu_dp[j], dst.imu_va[j] = src.imu_va[j];
      w.kfs_size++;
      if (p.counters) atomicAdd((unsigned long long*)&p.counters[1], 1ull);
    }
  }
  __syncthreads();
  if (w.overflow) return;
  if (st.lm_state == 0) {  // UN_INITIALIZED (vo_localmap.cpp:122-216)
    if (w.kfs_size < W) return;  // returns before pop_front (quirk A22)
    if (tid == 0) {
      w.n_edge = 0;
      w.edge_next_id = 0;
    }
    __syncthreads();
    for (int f = 0; f < W; f++) {
      const KeyFrameDev& kf = ring[(w.kfs_head + f) % W];
      if (tid == 0) bag_add_pose(w, W, kf.frame_id, kf.T_c_w);
      __syncthreads();
      // pose vertex id = ring slot of the frame (getPoseIdByReleventFrameId): slot f during initialisation; edge ids
      // are assigned keyframe by keyframe in the reference (after all vertices exist), same order here
      bag_add_keyframe(w, sid, s_cnt, kf, false, f);
    }
    if (tid < W) {
      w.pose_present[tid] = 1;
      w.pose_fixed[tid] = (tid == w.oldest) ? 1 : 0;
      pose_to_g2o(w.bag_pose[tid], w.pose_est[tid]);
      const KeyFrameDev& kf = ring[(w.kfs_head + tid) % W];  // slot tid holds the tid-th keyframe of the queue
      for (int j = 0; j < 4; j++) w.imu_dq[tid][j] = kf.imu_dq[j];
      w.imu_dt[tid] = kf.imu_dt;
      w.imu_has[tid] = (tid > 0 && kf.imu_valid) ? 1 : 0;
      for (int j = 0; j < 3; j++) w.imu_dp[tid][j] = kf.imu_dp[j], w.imu_va[tid][j] = kf.imu_va[j];
    }
    for (int i = tid; i < w.n_lm; i += BU_T)
      for (int j = 0; j < 3; j++) w.lm_est[i][j] = w.lm_p3d[i][j];  // vertex estimate = running mean (quirk A23)
  } else {  // SLIDING_WINDOW (vo_localmap.cpp:218-284)
    const int old = w.oldest;
    edges_remove_if(w, s_cnt, [&](int i) { return w.e_pose[i] == old; });
    if (tid == 0) w.pose_present[old] = 0;
    {  // for(auto id : kfs.at(0).lm_id) if(bag->removeLMObservation(id)) optimizer.removeVertex(lm)
      const KeyFrameDev& k0 = ring[w.kfs_head % W];
      for (int i = tid; i < k0.lm_count; i += BU_T) {
        int f = bag_find(sid, w.n_lm, k0.lm_id[i]);
        if (f >= 0) w.lm_count[f]--;
      }
      __syncthreads();
      edges_remove_if(w, s_cnt, [&](int i) { return w.lm_count[w.e_lidx[i]] == 0; });  // edges vanish with the vertex
      // erase those landmarks from the bag (order preserving) and remap the edges' bag indices
      const int n = w.n_lm;
      int* remap = reinterpret_cast<int*>(p.ba_scratch + (size_t)s * p.ba_scratch_stride);
      int kept = 0;
      for (int base = 0; base < n; base += BU_T) {
        const int i = base + tid;
        const bool keep = i < n && w.lm_count[i] != 0;
        long long id = 0;
        int cnt = 0;
        double a[3] = {0, 0, 0}, b[3] = {0, 0, 0};
        if (keep) {
          id = w.lm_id[i];
          cnt = w.lm_count[i];
          for (int j = 0; j < 3; j++) {
            a[j] = w.lm_p3d[i][j];
            b[j] = w.lm_est[i][j];
          }
        }
        int tot;
        const int rk = block_rank<BU_NW>(keep, s_cnt, tot);
        if (i < n) remap[i] = keep ? kept + rk : 0;
        if (keep) {
          const int k = kept + rk;
          w.lm_id[k] = id;
          sid[k] = id;
          w.lm_count[k] = cnt;
          for (int j = 0; j < 3; j++) {
            w.lm_p3d[k][j] = a[j];
            w.lm_est[k][j] = b[j];
          }
        }
        kept += tot;
      }
      __syncthreads();
      for (int e = tid; e < w.n_edge; e += BU_T) w.e_lidx[e] = remap[w.e_lidx[e]];
      __syncthreads();
      if (tid == 0) w.n_lm = kept;
      __syncthreads();
    }
    const KeyFrameDev& kn = ring[(w.kfs_head + w.kfs_size - 1) % W];
    if (tid == 0) {
      bag_add_pose(w, W, kn.frame_id, kn.T_c_w);
      w.pose_present[w.newest] = 1;
      w.pose_fixed[w.newest] = 0;
      pose_to_g2o(kn.T_c_w, w.pose_est[w.newest]);
      w.pose_fixed[w.oldest] = 1;
      for (int j = 0; j < 4; j++) w.imu_dq[w.newest][j] = kn.imu_dq[j];
      w.imu_dt[w.newest] = kn.imu_dt;
      w.imu_has[w.newest] = kn.imu_valid ? 1 : 0;
      for (int j = 0; j < 3; j++) w.imu_dp[w.newest][j] = kn.imu_dp[j], w.imu_va[w.newest][j] = kn.imu_va[j];
    }
    __syncthreads();
    bag_add_keyframe(w, sid, s_cnt, kn, true, w.newest);
  }
  __syncthreads();
  if (tid == 0) {
    w.solve = w.overflow ? 0 : 1;
    // kfs.pop_front() happens after the optimisation in the reference; nothing reads kfs in between
    w.kfs_head = (w.kfs_head + 1) % W;
    w.kfs_size--;
  }
}

