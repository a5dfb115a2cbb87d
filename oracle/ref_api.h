// ORACLE (test infrastructure, NOT product code): declarations shared by the oracle translation units.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
#pragma once
#include <cstdint>
#include <vector>

namespace ref {

struct Pt2f {
  float x, y;
};

void equalize_hist(const uint8_t* src, uint8_t* dst, int w, int h);
void pyr_down(const uint8_t* src, int w, int h, uint8_t* dst);
int lk_num_levels(int w, int h, int win, int max_level);
void calc_optical_flow_pyr_lk(const uint8_t* prev, const uint8_t* next, int w, int h, const float* prev_pts,
                              float* next_pts, uint8_t* status, int n, int win, int max_level, int max_iter,
                              double eps, int use_initial_flow, float min_eig_thr);
void min_eigen_map(const uint8_t* img, int w, int h, float* eig);
int good_features_to_track(const uint8_t* img, int w, int h, int max_corners, double quality, double min_distance,
                           float* out_xy);

// src/processing/include/feature_dem.h:40-50
class FeatureDEM {
 public:
  struct Scored {
    Pt2f pt;
    float score;
  };
  FeatureDEM(int image_width, int image_height, const double f_para[6]);
  void detect(const uint8_t* img, std::vector<Pt2f>& newPts) const;
  void redetect(const uint8_t* img, const std::vector<Pt2f>& existedPts, std::vector<Pt2f>& newPts) const;
  float calHarrisR(const uint8_t* img, float ptx, float pty) const;

  int width, height, regionWidth, regionHeight, boundary_dis;
  unsigned max_region_feature_num, min_region_feature_num;
  int gftt_num;
  double gftt_ql;
  int gftt_dis;

 private:
  void fillIntoRegion(const uint8_t* img, const std::vector<Pt2f>& pts, std::vector<Scored> (&region)[16],
                      bool existed) const;
};

}  // namespace ref
