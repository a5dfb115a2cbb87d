// flvis_amd: EPnP (Lepetit, Moreno-Noguer, Fua: "EPnP: An accurate O(n) solution to the PnP problem", IJCV 2009) as cv::solvePnP runs it
// (calib3d/src/epnp.cpp, the authors' code): the RANSAC kernel of cv::solvePnPRansac for SOLVEPNP_ITERATIVE (5-point subsets) and the
// final solve on the inliers for SOLVEPNP_P3P (lkorb_tracking.cpp:170-177).
//
//   control points  C0 = centroid of the object points, C1..C3 = C0 + sqrt(lambda_i / n) e_i from the PCA of the centred points
//   alphas          barycentric coordinates of every object point with respect to the control points
//   M (2n x 12)     rows [a_j fu, 0, a_j (uc - u)], [0, a_j fv, a_j (vc - v)], j = 0..3;  MtM = M^T M (12 x 12)
//   v[0..3]         eigenvectors of MtM for its four smallest eigenvalues (v[0]: the smallest)
//   L (6 x 10), rho the quadratic constraints "distances between control points are preserved"
//   betas           three approximations (N = 4: B11 B12 B13 B14, N = 2: B11 B12 B22, N = 3: B11 B12 B22 B13 B23), each refined by five
//                   Gauss-Newton steps; R, t by absolute orientation (Arun) of the camera-frame points; the N with the smallest mean
//                   reprojection error wins
//
// The 12 x 12 eigen-decomposition is the expensive part.  It is a cyclic Jacobi method in the PARALLEL (round-robin tournament)
// ordering -- six disjoint rotations per step, eleven steps per sweep -- so that one wave can apply a whole step at once; a step reads
// everything it needs before it writes anything, and no entry is written by two work items, so that the items can be computed in any
// order: the CPU checker of the tests runs the very same functions with one "lane" and obtains the same bits.  Sums over the correspondences are sequential in
// index order, one output entry per lane.  OpenCV decomposes MtM with its own SVD; the subspaces, and with them the pose, agree up to
// rounding.  Everything is plain C++ (+, -, *, /, sqrt) compiled with -ffp-contract=off on both sides.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define EPNP_FN __host__ __device__ inline

#else
#define EPNP_FN inline

#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define EPNP_KEEP(x) asm volatile("" : "+v"(x))  // the value is wanted NOW (keeps a load from sinking behind a branch); no code
#else
#define EPNP_KEEP(x) (void)(x)
#endif
#if defined(__clang__)
#define EPNP_UNROLL _Pragma("unroll")
#else
#define EPNP_UNROLL _Pragma("GCC unroll 8")
#endif

namespace flvis {
namespace epnp {

constexpr int SWEEPS_MAX = 12;  // Jacobi sweeps of the 12 x 12 eigen-decomposition at most (it stops by itself after 5-7 on these matrices)

struct Camera {
  double fu, fv, uc, vc;
};

// workspace of one EPnP instance (LDS on the device, stack on the host)
struct Work {
  double cws[4][3];        // control points, world frame
  double ci[9];            // inverse of [C1-C0 C2-C0 C3-C0]
  double AV[288];          // [0, 144): A = MtM, which the Jacobi steps turn into diag(eigenvalues); [144, 288): V, the accumulated
                           // rotations (columns = eigenvectors)
  double cs[6][2];         // rotation (c, s) of the six pairs of the current step
  double tol2;             // (1e-7 x mean diagonal of MtM)^2: a rotation above it keeps the sweeps going
  double skip2;            // (1e-10 x mean diagonal of MtM)^2: an entry below it is not rotated
  int active, rotated;     // this sweep needs a successor; tag of the last step that rotated something
  double v[4][12];         // the four eigenvectors used (v[0]: smallest eigenvalue)
  double L[60], rho[6];
  double betas[3][4];
  double R[3][9], t[3][3], err[3];
  double sgn[3];           // per approximation: -1 when the first camera-frame point came out behind the camera (solve_for_sign)
  double acc[3][16];       // per approximation: centroid sums / ABt of the absolute orientation
  double ccs[3][4][3];     // per approximation: control points in the camera frame
};

struct Pose {
  double R[9], t[3];
  bool ok;
};

// barycentric coordinates of a world point
EPNP_FN void alphas_of(const Work& w, const double* p, double* a) {
  const double d0 = p[0] - w.cws[0][0], d1 = p[1] - w.cws[0][1], d2 = p[2] - w.cws[0][2];
  a[1] = w.ci[0] * d0 + w.ci[1] * d1 + w.ci[2] * d2;
  a[2] = w.ci[3] * d0 + w.ci[4] * d1 + w.ci[5] * d2;
  a[3] = w.ci[6] * d0 + w.ci[7] * d1 + w.ci[8] * d2;
  a[0] = 1.0 - a[1] - a[2] - a[3];
}

// the rotation (c, s) that annihilates a_pq: column / row p' = c p - s q, q' = s p + c q.  With d = a_qq - a_pp, b = 2 a_pq and
// h = sqrt(d^2 + b^2): tan = sgn b / (|d| + h) (the smaller root), c = (|d| + h) / sqrt(2 h (|d| + h)), s = sgn |b| / sqrt(2 h (|d| + h)),
// sgn = the sign of d / b (+ for d = 0) -- two square roots and one division in a row.
EPNP_FN void jacobi_rotation(double app, double aqq, double apq, double& c, double& s) {
  c = 1.0, s = 0.0;
  if (apq != 0.0) {
    const double d = aqq - app, b = 2.0 * apq;
    const double ad = fabs(d), ab = fabs(b);
    const double h = sqrt(d * d + b * b);
    const double g = ad + h;
    const double r = 1.0 / sqrt((2.0 * h) * g);
    const bool neg = d != 0.0 && ((d < 0) != (b < 0));
    c = g * r;
    s = neg ? -(ab * r) : ab * r;
  }
}

// ---- serial helpers.  Every local array is indexed by compile-time constants after unrolling (registers on the device). -----------------
// one Jacobi rotation of a symmetric 3 x 3 matrix on the pair (P, Q), accumulated into v
template <int P, int Q>
EPNP_FN bool sym3_rotate(double (&a)[3][3], double (&v)[3][3]) {
  const double apq = a[P][Q];
  if (apq == 0.0) return false;
  if (apq * apq <= 4.930380657631324e-30 * fabs(a[P][P] * a[Q][Q])) {  // below 10 eps of the pair's scale: done with it
    a[P][Q] = 0.0, a[Q][P] = 0.0;
    return false;
  }
  double c, s;
  jacobi_rotation(a[P][P], a[Q][Q], apq, c, s);
EPNP_UNROLL
  for (int k = 0; k < 3; k++) {  // columns P, Q
    const double akp = a[k][P], akq = a[k][Q];
    a[k][P] = c * akp - s * akq;
    a[k][Q] = s * akp + c * akq;
  }
EPNP_UNROLL
  for (int k = 0; k < 3; k++) {  // rows P, Q
    const double apk = a[P][k], aqk = a[Q][k];
    a[P][k] = c * apk - s * aqk;
    a[Q][k] = s * apk + c * aqk;
  }
EPNP_UNROLL
  for (int k = 0; k < 3; k++) {
    const double vkp = v[k][P], vkq = v[k][Q];
    v[k][P] = c * vkp - s * vkq;
    v[k][Q] = s * vkp + c * vkq;
  }
  a[P][Q] = 0.0, a[Q][P] = 0.0;
  return true;
}
// eigen-decomposition of a symmetric 3 x 3 matrix (cyclic Jacobi until a sweep finds nothing to rotate, 8 sweeps at most): eigenvalues descending in d, eigenvectors as ROWS of ut
EPNP_FN void sym3_eig_desc(const double* S, double* d, double* ut) {
  double a[3][3] = {{S[0], S[1], S[2]}, {S[1], S[4], S[5]}, {S[2], S[5], S[8]}};
  double v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int sweep = 0; sweep < 8; sweep++) {
    const bool r0 = sym3_rotate<0, 1>(a, v), r1 = sym3_rotate<0, 2>(a, v), r2 = sym3_rotate<1, 2>(a, v);
    if (!(r0 || r1 || r2)) break;
  }
  double e0 = a[0][0], e1 = a[1][1], e2 = a[2][2];
  double c0[3] = {v[0][0], v[1][0], v[2][0]}, c1[3] = {v[0][1], v[1][1], v[2][1]}, c2[3] = {v[0][2], v[1][2], v[2][2]};
  auto order = [](double& ea, double* ca, double& eb, double* cb) {  // (ea, ca) before (eb, cb) unless eb is larger
    if (ea < eb) {
      const double t = ea;
      ea = eb, eb = t;
EPNP_UNROLL
      for (int k = 0; k < 3; k++) {
        const double u = ca[k];
        ca[k] = cb[k], cb[k] = u;
      }
    }
  };
  order(e0, c0, e1, c1);
  order(e1, c1, e2, c2);
  order(e0, c0, e1, c1);
  d[0] = e0, d[1] = e1, d[2] = e2;
EPNP_UNROLL
  for (int k = 0; k < 3; k++) ut[k] = c0[k], ut[3 + k] = c1[k], ut[6 + k] = c2[k];
}

// least squares min |A x - b| for a 6 x N system by Householder QR; A (row-major 6 x N) and b are overwritten.  Columns of zeros at the
// end (the shorter beta approximations padded to five unknowns) are passed over and get x = 0.
template <int N>
EPNP_FN void qr_solve6(double (&A)[6 * N], double (&b)[6], double (&x)[N]) {
EPNP_UNROLL
  for (int k = 0; k < N; k++) {
    double nrm = 0;
EPNP_UNROLL
    for (int i = k; i < 6; i++) nrm += A[i * N + k] * A[i * N + k];
    nrm = sqrt(nrm);
    if (nrm != 0.0) {
      const double alpha = A[k * N + k] > 0 ? -nrm : nrm;
      double vk[6];
      vk[k] = A[k * N + k] - alpha;
EPNP_UNROLL
      for (int i = k + 1; i < 6; i++) vk[i] = A[i * N + k];
      double vv = 0;
EPNP_UNROLL
      for (int i = k; i < 6; i++) vv += vk[i] * vk[i];
      if (vv != 0.0) {
        const double tv = 2.0 / vv;
EPNP_UNROLL
        for (int j = k; j < N; j++) {
          double dot = 0;
EPNP_UNROLL
          for (int i = k; i < 6; i++) dot += vk[i] * A[i * N + j];
          const double f = dot * tv;
EPNP_UNROLL
          for (int i = k; i < 6; i++) A[i * N + j] -= f * vk[i];
        }
        double dot = 0;
EPNP_UNROLL
        for (int i = k; i < 6; i++) dot += vk[i] * b[i];
        const double f = dot * tv;
EPNP_UNROLL
        for (int i = k; i < 6; i++) b[i] -= f * vk[i];
      }
    }
  }
EPNP_UNROLL
  for (int k = N - 1; k >= 0; k--) {
    double sum = b[k];
EPNP_UNROLL
    for (int j = k + 1; j < N; j++) sum -= A[k * N + j] * x[j];
    x[k] = A[k * N + k] != 0.0 ? sum / A[k * N + k] : 0.0;
  }
}

// one rotation of the one-sided Jacobi (Hestenes) on the columns (P, Q) of a, accumulated into v
template <int P, int Q>
EPNP_FN bool cols3_rotate(double (&a)[3][3], double (&v)[3][3]) {
  double al = 0, be = 0, ga = 0;
EPNP_UNROLL
  for (int k = 0; k < 3; k++) {
    al += a[k][P] * a[k][P];
    be += a[k][Q] * a[k][Q];
    ga += a[k][P] * a[k][Q];
  }
  if (ga * ga <= 4.930380657631324e-30 * (al * be)) return false;  // the columns are orthogonal to 10 eps (OpenCV's Jacobi SVD test)
  double c, s;
  jacobi_rotation(al, be, ga, c, s);
EPNP_UNROLL
  for (int k = 0; k < 3; k++) {
    const double x = a[k][P], y = a[k][Q];
    a[k][P] = c * x - s * y;
    a[k][Q] = s * x + c * y;
    const double vx = v[k][P], vy = v[k][Q];
    v[k][P] = c * vx - s * vy;
    v[k][Q] = s * vx + c * vy;
  }
  return true;
}
// singular value decomposition of a 3 x 3 matrix by one-sided Jacobi on its columns: M = U diag(s) V^T -> R = U V^T with det(R) made +1
// the way estimate_R_and_t does (row 2 negated)
EPNP_FN void arun_rotation(const double* M, double* R) {
  double a[3][3] = {{M[0], M[1], M[2]}, {M[3], M[4], M[5]}, {M[6], M[7], M[8]}};  // columns get orthogonalised: a = U diag(s)
  double v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int sweep = 0; sweep < 12; sweep++) {
    const bool r0 = cols3_rotate<0, 1>(a, v), r1 = cols3_rotate<0, 2>(a, v), r2 = cols3_rotate<1, 2>(a, v);
    if (!(r0 || r1 || r2)) break;
  }
  // U = normalised columns of a (a column of zero norm -- rank-deficient correspondences -- is completed by the cross product)
  double u[3][3], nrm[3];
EPNP_UNROLL
  for (int j = 0; j < 3; j++) {
    nrm[j] = sqrt(a[0][j] * a[0][j] + a[1][j] * a[1][j] + a[2][j] * a[2][j]);
EPNP_UNROLL
    for (int k = 0; k < 3; k++) u[k][j] = nrm[j] > 0 ? a[k][j] / nrm[j] : 0.0;
  }
  const double tiny = 1e-300 * (nrm[0] + nrm[1] + nrm[2]);
  if (nrm[0] <= nrm[1] && nrm[0] <= nrm[2]) {
    if (!(nrm[0] > tiny)) {
      u[0][0] = u[1][1] * u[2][2] - u[2][1] * u[1][2];
      u[1][0] = u[2][1] * u[0][2] - u[0][1] * u[2][2];
      u[2][0] = u[0][1] * u[1][2] - u[1][1] * u[0][2];
    }
  } else if (nrm[1] <= nrm[2]) {
    if (!(nrm[1] > tiny)) {
      u[0][1] = u[1][2] * u[2][0] - u[2][2] * u[1][0];
      u[1][1] = u[2][2] * u[0][0] - u[0][2] * u[2][0];
      u[2][1] = u[0][2] * u[1][0] - u[1][2] * u[0][0];
    }
  } else {
    if (!(nrm[2] > tiny)) {
      u[0][2] = u[1][0] * u[2][1] - u[2][0] * u[1][1];
      u[1][2] = u[2][0] * u[0][1] - u[0][0] * u[2][1];
      u[2][2] = u[0][0] * u[1][1] - u[1][0] * u[0][1];
    }
  }
EPNP_UNROLL
  for (int i = 0; i < 3; i++)
EPNP_UNROLL
    for (int j = 0; j < 3; j++) R[3 * i + j] = u[i][0] * v[j][0] + u[i][1] * v[j][1] + u[i][2] * v[j][2];
  const double det = R[0] * R[4] * R[8] + R[1] * R[5] * R[6] + R[2] * R[3] * R[7] - R[2] * R[4] * R[6] - R[1] * R[3] * R[8] - R[0] * R[5] * R[7];
  if (det < 0) {
    R[6] = -R[6];
    R[7] = -R[7];
    R[8] = -R[8];
  }
}

// ---- the phases.  `lane` of `nl` lanes takes the elements lane, lane + nl, ...; `sync` separates what one lane writes from what another
// reads (a no-op for one lane).  Points: world points pw(i) and pixel coordinates uv(i) through accessors, so that callers keep their own
// layouts.
//
// Sums over the correspondences have ONE definition whatever the number of lanes: the sum over i < n is the sum, in chunk order, of the
// sequential sums over chunks of consecutive correspondences; a chunk is 16 correspondences long up to n = 256, 32 up to 512, 64 up to
// 1024 (never more than 16 chunks).  For a RANSAC sample (5) that is the plain sequential sum; beyond 16 correspondences (entry, chunk)
// pairs are independent work items and `part` (entries x chunks doubles, <= 144 x 16) holds the chunk sums.
// acc(e, i, s): add correspondence i's term(s) of entry e to s;  out(e, s): store entry e's sum.
constexpr int PART_DOUBLES = 144 * 16;
#ifdef FLVIS_REF_ORDER_G2O  // (the CPU checker's reference-order build, REF_ORDER=g2o: one chunk = the plain sequential sums of
                            //  OpenCV's epnp.cpp loops; never defined for the product)
EPNP_FN int chunk_len(int n) { return n > 0 ? n : 1; }
#else
EPNP_FN int chunk_len(int n) { return n <= 256 ? 16 : (n <= 512 ? 32 : 64); }
#endif
template <class ACC, class OUT, class SYNC>
EPNP_FN void chunked_sums(int n, int entries, double* part, int lane, int nl, SYNC sync, ACC acc, OUT out) {
  const int CHUNK = chunk_len(n), nch = (n + CHUNK - 1) / CHUNK;
  if (nch <= 1) {
    for (int e = lane; e < entries; e += nl) {
      double s = 0;
      for (int i = 0; i < n; i++) acc(e, i, s);
      out(e, s);
    }
    sync();
    return;
  }
  for (int item = lane; item < entries * nch; item += nl) {
    const int e = item / nch, c = item - e * nch;
    const int i1 = (c + 1) * CHUNK < n ? (c + 1) * CHUNK : n;
    double s = 0;
    for (int i = c * CHUNK; i < i1; i++) acc(e, i, s);
    part[item] = s;
  }
  sync();
  for (int e = lane; e < entries; e += nl) {
    double s = 0;
    for (int c = 0; c < nch; c++) s += part[e * nch + c];
    out(e, s);
  }
  sync();
}

// control points and the inverse of the control-point basis (the two sums by everybody, the rest by lane 0)
template <class PW, class SYNC>
EPNP_FN void phase_control_points(Work& w, int n, PW pw, double* part, int lane, int nl, SYNC sync) {
  chunked_sums(n, 3, part, lane, nl, sync,
               [&](int e, int i, double& s) {
                 double p[3];
                 pw(i, p);
                 s += p[e];
               },
               [&](int e, double s) { w.cws[0][e] = s / n; });
  chunked_sums(n, 9, part, lane, nl, sync,
               [&](int e, int i, double& s) {
                 double p[3];
                 pw(i, p);
                 const int r = e / 3, c = e - 3 * r;
                 s += (p[r] - w.cws[0][r]) * (p[c] - w.cws[0][c]);
               },
               [&](int e, double s) { w.acc[0][e] = s; });
  if (lane != 0) return;
  const double c0[3] = {w.cws[0][0], w.cws[0][1], w.cws[0][2]};
  double S[9];
  for (int e = 0; e < 9; e++) S[e] = w.acc[0][e];
  double dc[3], uct[9];
  sym3_eig_desc(S, dc, uct);
  for (int i = 1; i < 4; i++) {
    const double k = sqrt((dc[i - 1] > 0 ? dc[i - 1] : 0.0) / n);
    for (int j = 0; j < 3; j++) w.cws[i][j] = c0[j] + k * uct[3 * (i - 1) + j];
  }
  // cc[3 i + (j - 1)] = cws[j][i] - cws[0][i]; inverse by the adjugate
  double cc[9];
  for (int i = 0; i < 3; i++)
    for (int j = 1; j < 4; j++) cc[3 * i + j - 1] = w.cws[j][i] - w.cws[0][i];
  const double det = cc[0] * (cc[4] * cc[8] - cc[5] * cc[7]) - cc[1] * (cc[3] * cc[8] - cc[5] * cc[6]) + cc[2] * (cc[3] * cc[7] - cc[4] * cc[6]);
  const double id = 1.0 / det;
  w.ci[0] = (cc[4] * cc[8] - cc[5] * cc[7]) * id;
  w.ci[1] = (cc[2] * cc[7] - cc[1] * cc[8]) * id;
  w.ci[2] = (cc[1] * cc[5] - cc[2] * cc[4]) * id;
  w.ci[3] = (cc[5] * cc[6] - cc[3] * cc[8]) * id;
  w.ci[4] = (cc[0] * cc[8] - cc[2] * cc[6]) * id;
  w.ci[5] = (cc[2] * cc[3] - cc[0] * cc[5]) * id;
  w.ci[6] = (cc[3] * cc[7] - cc[4] * cc[6]) * id;
  w.ci[7] = (cc[1] * cc[6] - cc[0] * cc[7]) * id;
  w.ci[8] = (cc[0] * cc[4] - cc[1] * cc[3]) * id;
}

// MtM (all 144 entries, each summed over the 2 n rows of M in row order) into A; V = identity
template <class PW, class UV, class SYNC>
EPNP_FN void phase_mtm(Work& w, int n, PW pw, UV uv, Camera cam, double* part, int lane, int nl, SYNC sync) {
  chunked_sums(n, 144, part, lane, nl, sync,
               [&](int e, int i, double& s) {
                 const int r = e / 12, c = e - 12 * r;
                 const int jr = r / 3, kr = r - 3 * jr, jc = c / 3, kc = c - 3 * jc;
                 double p[3], a[4], z[2];
                 pw(i, p);
                 uv(i, z);
                 alphas_of(w, p, a);
                 // rows M1 = [a fu, 0, a (uc - u)], M2 = [0, a fv, a (vc - v)] per control point
                 const double m1r = kr == 0 ? a[jr] * cam.fu : (kr == 1 ? 0.0 : a[jr] * (cam.uc - z[0]));
                 const double m1c = kc == 0 ? a[jc] * cam.fu : (kc == 1 ? 0.0 : a[jc] * (cam.uc - z[0]));
                 const double m2r = kr == 0 ? 0.0 : (kr == 1 ? a[jr] * cam.fv : a[jr] * (cam.vc - z[1]));
                 const double m2c = kc == 0 ? 0.0 : (kc == 1 ? a[jc] * cam.fv : a[jc] * (cam.vc - z[1]));
                 s += m1r * m1c;
                 s += m2r * m2c;
               },
               [&](int e, double s) {
                 w.AV[e] = s;
                 w.AV[144 + e] = (e / 12 == e % 12) ? 1.0 : 0.0;
               });
}

// ---- the 12 x 12 eigen-decomposition -------------------------------------------------------------------------------------------------
// pair m (0..5) of step k (0..10) of the round-robin tournament: (k, 11) and ((k + m) mod 11, (k - m) mod 11), m = 1..5; p < q
EPNP_FN void jacobi_pair(int k, int m, int& p, int& q) {
  int a = k + m, b = k - m;
  if (a >= 11) a -= 11;
  if (b < 0) b += 11;
  if (m == 0) b = 11;
  p = a < b ? a : b;
  q = a < b ? b : a;
}
// One step is A <- J^T A J, V <- V J for the six disjoint rotations J of the step:
//   (a) six lanes compute the rotations into w.cs (none for an entry below 1e-10 of the matrix scale).  A rotation that still has
//       something to do (|a_pq| above 1e-7 of the matrix scale) keeps the sweeps going; a sweep without one is the last (quadratic
//       convergence: what it leaves is below 1e-10 of the scale, orders below the effect of the pixels' float rounding);
//   (b) the four entries {p, q} x {u, v} of two pairs only mix among themselves, and so do the entries {r, r + 1} x {u, v} of V:
//       57 independent items, each reading and writing only its own entries,
//         0..20   the 2 x 2 blocks of A's upper block triangle (row pair R <= column pair C; the lower triangle is mirrored: A stays
//                 symmetric; in a pair's own block the rotation annihilates a_pq, which is stored as 0)
//         21..56  two rows of V times one column pair (the rows are not rotated: the same arithmetic with the row rotation (1, 0))
//       in ONE instruction stream: the three kinds differ in indices and selected values only.
EPNP_FN void phase_jacobi_angles(Work& w, int k, int tag, int lane, int nl) {
  for (int m = lane; m < 6; m += nl) {
    int p, q;
    jacobi_pair(k, m, p, q);
    double apq = w.AV[12 * p + q], app = w.AV[13 * p], aqq = w.AV[13 * q];  // (all three at once: one LDS round trip)
    EPNP_KEEP(app);
    EPNP_KEEP(aqq);
    double c = 1.0, s = 0.0;
    if (apq * apq > w.skip2) {  // (threshold Jacobi: an entry below 1e-10 of the matrix scale is left alone)
      jacobi_rotation(app, aqq, apq, c, s);
      w.rotated = tag;
      if (apq * apq > w.tol2) w.active = 1;
    }
    w.cs[m][0] = c;
    w.cs[m][1] = s;
  }
}
EPNP_FN void jacobi_item(Work& w, int k, int slot) {
  const bool is_a = slot < 21;
  // A block: pairs R <= C;  V item: rows 2 j2, 2 j2 + 1 and pair C
  const int R = (slot >= 6) + (slot >= 11) + (slot >= 15) + (slot >= 18) + (slot >= 20);
  const int j = slot - 21, j2 = j / 6;
  const int C = is_a ? R + slot - (6 * R - (R * (R - 1)) / 2) : j - 6 * j2;
  int p, q, u, v;
  jacobi_pair(k, is_a ? R : 0, p, q);
  jacobi_pair(k, C, u, v);
  if (!is_a) p = 2 * j2, q = p + 1;
  const bool diag = is_a && R == C;
  const int base = is_a ? 0 : 144;
  const int i_pu = base + 12 * p + u, i_pv = base + 12 * p + v, i_qu = base + 12 * q + u, i_qv = base + 12 * q + v;
  const double cr = is_a ? w.cs[R][0] : 1.0, sr = is_a ? w.cs[R][1] : 0.0, cc = w.cs[C][0], sc = w.cs[C][1];
  const double e_pu = w.AV[i_pu], e_pv = w.AV[i_pv], e_qu = w.AV[i_qu], e_qv = w.AV[i_qv];
  const double t_pu = cr * e_pu - sr * e_qu, t_pv = cr * e_pv - sr * e_qv;
  const double t_qu = sr * e_pu + cr * e_qu, t_qv = sr * e_pv + cr * e_qv;
  const double n_pu = cc * t_pu - sc * t_pv, n_qv = sc * t_qu + cc * t_qv;
  const double n_pv = diag ? 0.0 : sc * t_pu + cc * t_pv, n_qu = diag ? 0.0 : cc * t_qu - sc * t_qv;
  w.AV[i_pu] = n_pu, w.AV[i_pv] = n_pv, w.AV[i_qu] = n_qu, w.AV[i_qv] = n_qv;
  // the mirror image (V items: onto themselves)
  w.AV[is_a ? 12 * u + p : i_pu] = n_pu;
  w.AV[is_a ? 12 * v + p : i_pv] = n_pv;
  w.AV[is_a ? 12 * u + q : i_qu] = n_qu;
  w.AV[is_a ? 12 * v + q : i_qv] = n_qv;
}
// the eigen-decomposition of the symmetric 12 x 12 matrix A (destroyed: its diagonal ends up holding the eigenvalues), eigenvectors as
// the columns of V (identity on entry).  w.tol2 must be set (jacobi12_setup).
template <class SYNC>
EPNP_FN void jacobi12(Work& w, int lane, int nl, SYNC sync) {
  for (int sweep = 0; sweep < SWEEPS_MAX; sweep++) {
    if (lane == 0) w.active = 0;
    sync();
    for (int k = 0; k < 11; k++) {
      const int tag = 11 * sweep + k + 1;
      phase_jacobi_angles(w, k, tag, lane, nl);
      sync();
      if (w.rotated == tag)  // (a step without a rotation changes nothing)
        for (int slot = lane; slot < 57; slot += nl) jacobi_item(w, k, slot);
      sync();
    }
    const int active = w.active;
    sync();
    if (!active) break;
  }
}
// the convergence threshold of a matrix whose diagonal is in w.A
EPNP_FN void jacobi12_setup(Work& w, int lane) {
  if (lane == 0) {
    double tr = 0;
    for (int i = 0; i < 12; i++) tr += w.AV[13 * i];
    tr /= 12.0;
#ifndef EPNP_TOL2
#define EPNP_TOL2 1e-14
#endif
    w.tol2 = EPNP_TOL2 * tr * tr;
    w.skip2 = 1e-20 * tr * tr;
    w.rotated = 0;
  }
}

// after the sweeps: the four eigenvectors of the smallest eigenvalues (rank by counting; ties keep the column order) ...
EPNP_FN void phase_pick_vectors(Work& w, int lane, int nl) {
  for (int c = lane; c < 12; c += nl) {
    const double key = w.AV[13 * c];
    int rank = 0;
    for (int j = 0; j < 12; j++) {
      const double o = w.AV[13 * j];
      rank += (o < key || (o == key && j < c)) ? 1 : 0;
    }
    if (rank < 4)
      for (int r = 0; r < 12; r++) w.v[rank][r] = w.AV[144 + 12 * r + c];
  }
}
// ... and from them L (6 x 10: one entry per lane) and rho
EPNP_FN void phase_constraints(Work& w, int lane, int nl) {
  for (int e = lane; e < 66; e += nl) {
    const int i = e < 60 ? e / 10 : e - 60;  // the pair of control points: (0,1) (0,2) (0,3) (1,2) (1,3) (2,3)
    const int a = i < 3 ? 0 : (i < 5 ? 1 : 2), b = i < 3 ? i + 1 : (i < 5 ? i - 1 : 3);
    if (e >= 60) {
      const double dx = w.cws[a][0] - w.cws[b][0], dy = w.cws[a][1] - w.cws[b][1], dz = w.cws[a][2] - w.cws[b][2];
      w.rho[i] = dx * dx + dy * dy + dz * dz;
      continue;
    }
    const int j = e - 10 * i;
    // column j of L: the product of the eigenvectors (m, n): 0 (0,0) 1 (0,1) 2 (1,1) 3 (0,2) 4 (1,2) 5 (2,2) 6 (0,3) 7 (1,3) 8 (2,3) 9 (3,3)
    const int n = j < 1 ? 0 : (j < 3 ? 1 : (j < 6 ? 2 : 3));
    const int m = j - (n * (n + 1)) / 2;
    const double m0 = w.v[m][3 * a] - w.v[m][3 * b], m1 = w.v[m][3 * a + 1] - w.v[m][3 * b + 1], m2 = w.v[m][3 * a + 2] - w.v[m][3 * b + 2];
    const double n0 = w.v[n][3 * a] - w.v[n][3 * b], n1 = w.v[n][3 * a + 1] - w.v[n][3 * b + 1], n2 = w.v[n][3 * a + 2] - w.v[n][3 * b + 2];
    const double dot = m0 * n0 + m1 * n1 + m2 * n2;
    w.L[e] = m == n ? dot : 2.0 * dot;
  }
}

// beta approximation q (0: N = 4, 1: N = 2, 2: N = 3) + five Gauss-Newton steps + the control points in the camera frame
EPNP_FN void betas_of(Work& w, int q) {
  double* betas = w.betas[q];
  {
    // the approximation's columns of L (N = 4: 0 1 3 6, N = 2: 0 1 2, N = 3: 0 1 2 3 4), padded with zero columns to five unknowns
    double Aq[30], b[6], x[5];
EPNP_UNROLL
    for (int i = 0; i < 6; i++) {
EPNP_UNROLL
      for (int j = 0; j < 5; j++) {
        const int col = q == 0 ? (j < 2 ? j : (j == 2 ? 3 : 6)) : j;
        const bool used = q == 0 ? j < 4 : (q == 1 ? j < 3 : true);
        Aq[i * 5 + j] = used ? w.L[10 * i + col] : 0.0;
      }
      b[i] = w.rho[i];
    }
    qr_solve6<5>(Aq, b, x);
    if (q == 0) {
      if (x[0] < 0) {
        betas[0] = sqrt(-x[0]);
        betas[1] = -x[1] / betas[0];
        betas[2] = -x[2] / betas[0];
        betas[3] = -x[3] / betas[0];
      } else {
        betas[0] = sqrt(x[0]);
        betas[1] = x[1] / betas[0];
        betas[2] = x[2] / betas[0];
        betas[3] = x[3] / betas[0];
      }
    } else {
      if (x[0] < 0) {
        betas[0] = sqrt(-x[0]);
        betas[1] = (x[2] < 0) ? sqrt(-x[2]) : 0.0;
      } else {
        betas[0] = sqrt(x[0]);
        betas[1] = (x[2] > 0) ? sqrt(x[2]) : 0.0;
      }
      if (x[1] < 0) betas[0] = -betas[0];
      betas[2] = q == 2 ? x[3] / betas[0] : 0.0;
      betas[3] = 0.0;
    }
  }
  for (int it = 0; it < 5; it++) {  // gauss_newton
    double A4[24], b[6], x[4];
EPNP_UNROLL
    for (int i = 0; i < 6; i++) {
      const double* rl = w.L + 10 * i;
      A4[4 * i + 0] = 2 * rl[0] * betas[0] + rl[1] * betas[1] + rl[3] * betas[2] + rl[6] * betas[3];
      A4[4 * i + 1] = rl[1] * betas[0] + 2 * rl[2] * betas[1] + rl[4] * betas[2] + rl[7] * betas[3];
      A4[4 * i + 2] = rl[3] * betas[0] + rl[4] * betas[1] + 2 * rl[5] * betas[2] + rl[8] * betas[3];
      A4[4 * i + 3] = rl[6] * betas[0] + rl[7] * betas[1] + rl[8] * betas[2] + 2 * rl[9] * betas[3];
      b[i] = w.rho[i] - (rl[0] * betas[0] * betas[0] + rl[1] * betas[0] * betas[1] + rl[2] * betas[1] * betas[1] + rl[3] * betas[0] * betas[2] +
                         rl[4] * betas[1] * betas[2] + rl[5] * betas[2] * betas[2] + rl[6] * betas[0] * betas[3] + rl[7] * betas[1] * betas[3] +
                         rl[8] * betas[2] * betas[3] + rl[9] * betas[3] * betas[3]);
    }
    qr_solve6<4>(A4, b, x);
EPNP_UNROLL
    for (int i = 0; i < 4; i++) betas[i] += x[i];
  }
  for (int j = 0; j < 4; j++)  // compute_ccs
    for (int k = 0; k < 3; k++) {
      double s = 0;
      for (int i = 0; i < 4; i++) s += betas[i] * w.v[i][3 * j + k];
      w.ccs[q][j][k] = s;
    }
}
EPNP_FN void phase_betas(Work& w, int lane, int nl) {
  for (int q = lane; q < 3; q += nl) betas_of(w, q);
}

// camera-frame position of correspondence i under approximation q (compute_pcs), before solve_for_sign
EPNP_FN void pcs_of(const Work& w, int q, const double* a, double* pc) {
  for (int j = 0; j < 3; j++) pc[j] = a[0] * w.ccs[q][0][j] + a[1] * w.ccs[q][1][j] + a[2] * w.ccs[q][2][j] + a[3] * w.ccs[q][3][j];
}

// sums of the absolute orientation, for the three approximations at once: centroids of the camera points (k < 3) and of the world points
// (k < 6), then the nine entries of ABt.  solve_for_sign: all camera points flip when pcs[0].z < 0.
template <class PW>
EPNP_FN double sign_of(const Work& w, int q, PW pw) {
  double p0[3], a0[4], pc0[3];
  pw(0, p0);
  alphas_of(w, p0, a0);
  pcs_of(w, q, a0, pc0);
  return pc0[2] < 0 ? -1.0 : 1.0;
}
template <class PW, class SYNC>
EPNP_FN void phase_centroids(Work& w, int n, PW pw, double* part, int lane, int nl, SYNC sync) {
  for (int q = lane; q < 3; q += nl) w.sgn[q] = sign_of(w, q, pw);
  sync();
  chunked_sums(n, 18, part, lane, nl, sync,
               [&](int e, int i, double& s) {
                 const int q = e / 6, k = e - 6 * q;
                 double p[3];
                 pw(i, p);
                 if (k < 3) {
                   double a[4], pc[3];
                   alphas_of(w, p, a);
                   pcs_of(w, q, a, pc);
                   s += w.sgn[q] * pc[k];
                 } else {
                   s += p[k - 3];
                 }
               },
               [&](int e, double s) { w.acc[e / 6][e % 6] = s / n; });
}
template <class PW, class SYNC>
EPNP_FN void phase_abt(Work& w, int n, PW pw, double* part, int lane, int nl, SYNC sync) {
  chunked_sums(n, 27, part, lane, nl, sync,
               [&](int e, int i, double& s) {
                 const int q = e / 9, k = e - 9 * q, r = k / 3, c = k - 3 * r;
                 double p[3], a[4], pc[3];
                 pw(i, p);
                 alphas_of(w, p, a);
                 pcs_of(w, q, a, pc);
                 s += (w.sgn[q] * pc[r] - w.acc[q][r]) * (p[c] - w.acc[q][3 + c]);
               },
               [&](int e, double s) { w.acc[e / 9][6 + e % 9] = s; });
}
// R, t of the three approximations (three lanes) and their mean reprojection errors (everybody)
template <class PW, class UV, class SYNC>
EPNP_FN void phase_pose(Work& w, int n, PW pw, UV uv, Camera cam, double* part, int lane, int nl, SYNC sync) {
  for (int q = lane; q < 3; q += nl) {
    double* R = w.R[q];
    arun_rotation(&w.acc[q][6], R);
    for (int i = 0; i < 3; i++)
      w.t[q][i] = w.acc[q][i] - (R[3 * i] * w.acc[q][3] + R[3 * i + 1] * w.acc[q][4] + R[3 * i + 2] * w.acc[q][5]);
  }
  sync();
  chunked_sums(n, 3, part, lane, nl, sync,
               [&](int q, int i, double& s) {
                 const double* R = w.R[q];
                 double p[3], z[2];
                 pw(i, p);
                 uv(i, z);
                 const double Xc = R[0] * p[0] + R[1] * p[1] + R[2] * p[2] + w.t[q][0];
                 const double Yc = R[3] * p[0] + R[4] * p[1] + R[5] * p[2] + w.t[q][1];
                 const double inv_Zc = 1.0 / (R[6] * p[0] + R[7] * p[1] + R[8] * p[2] + w.t[q][2]);
                 const double ue = cam.uc + cam.fu * Xc * inv_Zc, ve = cam.vc + cam.fv * Yc * inv_Zc;
                 s += sqrt((z[0] - ue) * (z[0] - ue) + (z[1] - ve) * (z[1] - ve));
               },
               [&](int q, double s) {
                 const double e = s / n;
                 w.err[q] = e == e ? e : 1e300;  // (NaN: a degenerate approximation never wins)
               });
}
// the winner: N = 1 (q = 0) unless N = 2 is better, then N = 3 against that (epnp::compute_pose)
EPNP_FN Pose result(const Work& w) {
  int q = 0;
  if (w.err[1] < w.err[q]) q = 1;
  if (w.err[2] < w.err[q]) q = 2;
  Pose P;
  for (int i = 0; i < 9; i++) P.R[i] = w.R[q][i];
  for (int i = 0; i < 3; i++) P.t[i] = w.t[q][i];
  P.ok = w.err[q] < 1e300;
  return P;
}

// The solve in three parts so that a caller may give them different lane sets: `head` (control points, MtM: sums over the
// correspondences, as many lanes as there are), `mid` (eigen-decomposition .. betas: 64 lanes at most have work) and `sums` (absolute
// orientation and reprojection errors: sums over the correspondences again).  NL lanes; every part ends synchronised.
template <int NL, class PW, class UV, class SYNC>
EPNP_FN void solve_head(Work& w, int n, PW pw, UV uv, Camera cam, double* part, int lane, SYNC sync) {
  phase_control_points(w, n, pw, part, lane, NL, sync);
  sync();
  phase_mtm(w, n, pw, uv, cam, part, lane, NL, sync);
  jacobi12_setup(w, lane);
  sync();
}
template <int NL, class SYNC>
EPNP_FN void solve_mid(Work& w, int lane, SYNC sync) {
  jacobi12(w, lane, NL, sync);
  phase_pick_vectors(w, lane, NL);
  sync();
  phase_constraints(w, lane, NL);
  sync();
  phase_betas(w, lane, NL);
  sync();
}
template <int NL, class PW, class UV, class SYNC>
EPNP_FN void solve_sums(Work& w, int n, PW pw, UV uv, Camera cam, double* part, int lane, SYNC sync) {
  phase_centroids(w, n, pw, part, lane, NL, sync);
  phase_abt(w, n, pw, part, lane, NL, sync);
  phase_pose(w, n, pw, uv, cam, part, lane, NL, sync);
}
template <int NL, class PW, class UV, class SYNC>
EPNP_FN Pose solve(Work& w, int n, PW pw, UV uv, Camera cam, double* part, int lane, SYNC sync) {
  solve_head<NL>(w, n, pw, uv, cam, part, lane, sync);
  solve_mid<NL>(w, lane, sync);
  solve_sums<NL>(w, n, pw, uv, cam, part, lane, sync);
  return result(w);
}

}  // namespace epnp
}  // namespace flvis
