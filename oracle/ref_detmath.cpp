// ORACLE (test infrastructure, NOT product code): C entry points of the deterministic elementary functions the HIP kernels and
// this oracle share (flvis_amd/csrc/det_math.hpp), so that tests can compare them with libm.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
#include "../flvis_amd/csrc/det_math.hpp"

extern "C" {
double ref_det_sin(double x) { return detm::det_sin(x); }
double ref_det_cos(double x) { return detm::det_cos(x); }
double ref_det_atan(double x) { return detm::det_atan(x); }
double ref_det_atan2(double y, double x) { return detm::det_atan2(y, x); }
double ref_det_log(double x) { return detm::det_log(x); }
double ref_det_powi(double x, int n) { return detm::det_powi(x, n); }
double ref_det_acos(double x) { return detm::det_acos(x); }
double ref_det_cbrt(double x) { return detm::det_cbrt(x); }
void ref_det_batch(int which, int n, const double* a, const double* b, double* out) {
  for (int i = 0; i < n; i++) {
    switch (which) {
      case 0: out[i] = detm::det_sin(a[i]); break;
      case 1: out[i] = detm::det_cos(a[i]); break;
      case 2: out[i] = detm::det_atan(a[i]); break;
      case 3: out[i] = detm::det_atan2(a[i], b[i]); break;
      case 4: out[i] = detm::det_log(a[i]); break;
      case 5: out[i] = detm::det_acos(a[i]); break;
      default: out[i] = detm::det_cbrt(a[i]); break;
    }
  }
}
}
