/* TEST INFRASTRUCTURE ONLY (oracle/) -- the checker and the timed CPU baseline
 * ("port"), never part of the product path.
 *
 * Plain-C, single-threaded restatement of the reference hot path
 *   /root/reference/computer_code/api/helpers.py:203-421
 * on the packed layouts of include/mocap_core.h, carrying blob indices.
 * It is validated against oracle/mocap_oracle.py (which is bit-exact against the
 * reference's own functions run through oracle/ref_harness.py) and against the
 * golden vectors under tests/golden/.  Differences from the Python restatement are
 * rounding-level only: the reference gets P = K[R|t] and B = A^T A from BLAS dgemm
 * and the null vector from LAPACK dgesdd (helpers.py:307,319,320); here they are
 * plain loops and a cyclic Jacobi eigen-solve of the symmetric 4x4 B.
 *
 * PARITY UNPINNED for the three OpenCV calls (see oracle/cv_restate.py).
 *
 * Build: make -C oracle/c   (gcc -O2 -ffp-contract=off: no FMA contraction, so the
 * scalar expressions round like NumPy / baseline-x86-64 OpenCV).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  int C;
  int f32_rounding;
  double* K; /* [C][9]    */
  double* R; /* [C][9]    */
  double* t; /* [C][3]    */
  double* F; /* [C][C][9] */
} mo_cams;

/* ------------------------------------------------------------------ cameras */

/* cv::determinant, n = 4, CV_64F: hal::LU64f partial pivoting then p * prod(diag)
 * (oracle/cv_restate.py cv_determinant). */
static double det4_lu(const double* M) {
  double A[16];
  memcpy(A, M, sizeof A);
  double p = 1.0;
  const double eps = 2.220446049250313e-16 * 100;
  for (int i = 0; i < 4; i++) {
    int k = i;
    for (int j = i + 1; j < 4; j++)
      if (fabs(A[j * 4 + i]) > fabs(A[k * 4 + i])) k = j;
    if (fabs(A[k * 4 + i]) < eps) return 0.0;
    if (k != i) {
      for (int j = i; j < 4; j++) {
        double tmp = A[i * 4 + j];
        A[i * 4 + j] = A[k * 4 + j];
        A[k * 4 + j] = tmp;
      }
      p = -p;
    }
    double d = -1.0 / A[i * 4 + i];
    for (int j = i + 1; j < 4; j++) {
      double alpha = A[j * 4 + i] * d;
      for (int kk = i + 1; kk < 4; kk++) A[j * 4 + kk] = A[j * 4 + kk] + alpha * A[i * 4 + kk];
    }
  }
  double result = p;
  for (int i = 0; i < 4; i++) result = result * A[i * 4 + i];
  return result;
}

/* helpers.py:306-307 / :353-354: P = K @ [R | t] (3x4). */
static void projection(const double* K, const double* R, const double* t, double* P) {
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 4; c++) {
      double rt0 = c < 3 ? R[0 * 3 + c] : t[0];
      double rt1 = c < 3 ? R[1 * 3 + c] : t[1];
      double rt2 = c < 3 ? R[2 * 3 + c] : t[2];
      P[r * 4 + c] = K[r * 3 + 0] * rt0 + K[r * 3 + 1] * rt1 + K[r * 3 + 2] * rt2;
    }
}

/* cv.sfm.fundamentalFromProjections (helpers.py:362). */
static void fundamental(const double* P1, const double* P2, double* F) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double XY[16];
      memcpy(XY + 0, P1 + 4 * ((j + 1) % 3), 4 * sizeof(double));
      memcpy(XY + 4, P1 + 4 * ((j + 2) % 3), 4 * sizeof(double));
      memcpy(XY + 8, P2 + 4 * ((i + 1) % 3), 4 * sizeof(double));
      memcpy(XY + 12, P2 + 4 * ((i + 2) % 3), 4 * sizeof(double));
      F[i * 3 + j] = det4_lu(XY);
    }
}

mo_cams* mo_cams_create(int C, const double* K, const double* R, const double* t, int f32_rounding) {
  mo_cams* cm = (mo_cams*)calloc(1, sizeof *cm);
  cm->C = C;
  cm->f32_rounding = f32_rounding;
  cm->K = (double*)malloc(sizeof(double) * 9 * C);
  cm->R = (double*)malloc(sizeof(double) * 9 * C);
  cm->t = (double*)malloc(sizeof(double) * 3 * C);
  cm->F = (double*)calloc((size_t)9 * C * C, sizeof(double));
  memcpy(cm->K, K, sizeof(double) * 9 * C);
  memcpy(cm->R, R, sizeof(double) * 9 * C);
  memcpy(cm->t, t, sizeof(double) * 3 * C);
  double* P = (double*)malloc(sizeof(double) * 12 * C);
  for (int i = 0; i < C; i++) projection(K + 9 * i, R + 9 * i, t + 3 * i, P + 12 * i);
  for (int a = 0; a < C; a++)
    for (int b = 0; b < C; b++)
      if (a != b) fundamental(P + 12 * a, P + 12 * b, cm->F + 9 * ((size_t)a * C + b));
  free(P);
  return cm;
}

void mo_cams_destroy(mo_cams* cm) {
  if (!cm) return;
  free(cm->K);
  free(cm->R);
  free(cm->t);
  free(cm->F);
  free(cm);
}

void mo_get_fundamental(const mo_cams* cm, double* F) {
  memcpy(F, cm->F, sizeof(double) * 9 * cm->C * cm->C);
}

/* ------------------------------------------------------------------ 4x4 null vector */

/* Cyclic Jacobi on the symmetric 4x4 B; returns in v the eigenvector of the smallest
 * eigenvalue (== the last right-singular vector scipy.linalg.svd returns at
 * helpers.py:320-321, up to sign, which cancels in X = v[0:3]/v[3]). */
static void smallest_eigvec4(const double* Bin, double* v) {
  double a[4][4], V[4][4], d[4], bb[4], z[4];
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      a[i][j] = Bin[i * 4 + j];
      V[i][j] = i == j ? 1.0 : 0.0;
    }
  for (int i = 0; i < 4; i++) {
    bb[i] = d[i] = a[i][i];
    z[i] = 0.0;
  }
  for (int sweep = 0; sweep < 60; sweep++) {
    double sm = 0.0;
    for (int p = 0; p < 3; p++)
      for (int q = p + 1; q < 4; q++) sm += fabs(a[p][q]);
    if (sm == 0.0) break;
    double tresh = sweep < 3 ? 0.2 * sm / 16.0 : 0.0;
    for (int p = 0; p < 3; p++)
      for (int q = p + 1; q < 4; q++) {
        double g = 100.0 * fabs(a[p][q]);
        if (sweep > 3 && fabs(d[p]) + g == fabs(d[p]) && fabs(d[q]) + g == fabs(d[q])) {
          a[p][q] = 0.0;
        } else if (fabs(a[p][q]) > tresh) {
          double h = d[q] - d[p], tt;
          if (fabs(h) + g == fabs(h)) {
            tt = a[p][q] / h;
          } else {
            double theta = 0.5 * h / a[p][q];
            tt = 1.0 / (fabs(theta) + sqrt(1.0 + theta * theta));
            if (theta < 0.0) tt = -tt;
          }
          double c = 1.0 / sqrt(1.0 + tt * tt), s = tt * c, tau = s / (1.0 + c);
          h = tt * a[p][q];
          z[p] -= h;
          z[q] += h;
          d[p] -= h;
          d[q] += h;
          a[p][q] = 0.0;
#define ROT(m, i, j, k, l)                 \
  {                                        \
    double g_ = m[i][j], h_ = m[k][l];     \
    m[i][j] = g_ - s * (h_ + g_ * tau);    \
    m[k][l] = h_ + s * (g_ - h_ * tau);    \
  }
          for (int j = 0; j < p; j++) ROT(a, j, p, j, q);
          for (int j = p + 1; j < q; j++) ROT(a, p, j, j, q);
          for (int j = q + 1; j < 4; j++) ROT(a, p, j, q, j);
          for (int j = 0; j < 4; j++) ROT(V, j, p, j, q);
#undef ROT
        }
      }
    for (int i = 0; i < 4; i++) {
      bb[i] += z[i];
      d[i] = bb[i];
      z[i] = 0.0;
    }
  }
  int m = 0;
  for (int i = 1; i < 4; i++)
    if (fabs(d[i]) < fabs(d[m])) m = i;
  for (int i = 0; i < 4; i++) v[i] = V[i][m];
}

/* ------------------------------------------------------------------ DLT + reprojection */

static inline double round_f32(const mo_cams* cm, double x) {
  return cm->f32_rounding ? (double)(float)x : x;
}

/* helpers.py:293-327 (triangulate_point + DLT).  ox/oy [C], NaN = unseen.
 * Returns the number of views; X valid when >= 2.
 * Quirk kept: intrinsics by compacted index j, pose by camera c (helpers.py:305-307). */
static int dlt_point(const mo_cams* cm, const double* ox, const double* oy, double* X) {
  double B[16] = {0};
  int v = 0;
  for (int c = 0; c < cm->C; c++) {
    if (isnan(ox[c])) continue;
    double P[12];
    projection(cm->K + 9 * v, cm->R + 9 * c, cm->t + 3 * c, P);
    double ra[4], rb[4];
    for (int k = 0; k < 4; k++) {
      ra[k] = oy[c] * P[8 + k] - P[4 + k]; /* helpers.py:315 */
      rb[k] = P[0 + k] - ox[c] * P[8 + k]; /* helpers.py:316 */
    }
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4; j++) B[i * 4 + j] = B[i * 4 + j] + (ra[i] * ra[j] + rb[i] * rb[j]);
    v++;
  }
  if (v <= 1) return v;
  double vec[4];
  smallest_eigvec4(B, vec);
  X[0] = vec[0] / vec[3];
  X[1] = vec[1] / vec[3];
  X[2] = vec[2] / vec[3];
  return v;
}

/* helpers.py:214-241 with cv.projectPoints restated (oracle/cv_restate.py project_points).
 * Summation order of errors.mean(): NumPy pairwise (8 lanes) for float64 arrays (all cameras
 * seen), left-to-right for object arrays (some None). */
static double reproj_error(const mo_cams* cm, const double* ox, const double* oy, const double* Xin,
                           int v) {
  double X[3] = {round_f32(cm, Xin[0]), round_f32(cm, Xin[1]), round_f32(cm, Xin[2])};
  double comps[2 * 256];
  int n = 0, j = 0;
  for (int c = 0; c < cm->C; c++) {
    if (isnan(ox[c])) continue;
    const double* R = cm->R + 9 * c;
    const double* t = cm->t + 3 * c;
    const double* K = cm->K + 9 * j;
    double x = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + t[0];
    double y = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + t[1];
    double z = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + t[2];
    z = z != 0.0 ? 1.0 / z : 1.0;
    x *= z;
    y *= z;
    double pu = round_f32(cm, x * K[0] + K[2]);
    double pv = round_f32(cm, y * K[4] + K[5]);
    double du = ox[c] - pu, dv = oy[c] - pv;
    comps[n++] = du * du;
    comps[n++] = dv * dv;
    j++;
  }
  double s;
  if (v != cm->C || n < 8) {
    s = 0.0;
    for (int i = 0; i < n; i++) s = s + comps[i];
  } else {
    double r[8];
    int i;
    for (i = 0; i < 8; i++) r[i] = comps[i];
    for (i = 8; i < n - (n % 8); i += 8)
      for (int k = 0; k < 8; k++) r[k] = r[k] + comps[i + k];
    s = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) s = s + comps[i];
  }
  return s / n;
}

/* helpers.py:330-336 + :203-211 on explicit correspondences. */
void mo_triangulate(const mo_cams* cm, int64_t N, const double* obs, double* xyz, double* err) {
  int C = cm->C;
  double* ox = (double*)malloc(sizeof(double) * 2 * C);
  double* oy = ox + C;
  for (int64_t n = 0; n < N; n++) {
    for (int c = 0; c < C; c++) {
      ox[c] = obs[(n * C + c) * 2 + 0];
      oy[c] = obs[(n * C + c) * 2 + 1];
      if (isnan(ox[c]) || isnan(oy[c])) ox[c] = oy[c] = NAN;
    }
    double X[3] = {NAN, NAN, NAN};
    int v = dlt_point(cm, ox, oy, X);
    xyz[n * 3 + 0] = v >= 2 ? X[0] : NAN;
    xyz[n * 3 + 1] = v >= 2 ? X[1] : NAN;
    xyz[n * 3 + 2] = v >= 2 ? X[2] : NAN;
    if (err) err[n] = v >= 2 ? reproj_error(cm, ox, oy, X, v) : NAN;
  }
  free(ox);
}

/* ------------------------------------------------------------------ frame path */

/* helpers.py:339-421 for n_frames independent frames (layouts: include/mocap_core.h). */
void mo_match_triangulate(const mo_cams* cm, int64_t n_frames, int M_max, const float* blobs,
                          const int32_t* counts, double gate_px, int K_max, int64_t G_cap,
                          double* xyz, double* err, int16_t* corr, int32_t* n_out, int32_t* status,
                          int32_t* n_cand) {
  const int C = cm->C;
  const int R_cap = C * M_max; /* every blob can become a root at most once */
  int* root_cam = (int*)malloc(sizeof(int) * R_cap);
  int* root_blob = (int*)malloc(sizeof(int) * R_cap);
  /* hits[r][c][k], nh[r][c] */
  int16_t* hits = (int16_t*)malloc(sizeof(int16_t) * (size_t)R_cap * C * M_max);
  int* nh = (int*)malloc(sizeof(int) * (size_t)R_cap * C);
  double* dist = (double*)malloc(sizeof(double) * M_max);
  char* claimed = (char*)malloc(M_max);
  double* ox = (double*)malloc(sizeof(double) * 2 * C);
  double* oy = ox + C;
  int* sel = (int*)malloc(sizeof(int) * C);
  int* best_sel = (int*)malloc(sizeof(int) * C);

  for (int64_t f = 0; f < n_frames; f++) {
    const float* fb = blobs + (size_t)f * C * M_max * 2;
    const int32_t* fc = counts + (size_t)f * C;
    int st = 0;
    long long ncand = 0;
    int nroots = 0;
    memset(nh, 0, sizeof(int) * (size_t)R_cap * C);
    int n0 = fc[0] < M_max ? fc[0] : M_max;
    for (int k = 0; k < n0; k++) { /* helpers.py:349,357 */
      root_cam[nroots] = 0;
      root_blob[nroots] = k;
      nroots++;
    }
    for (int i = 1; i < C; i++) { /* helpers.py:359 */
      int ni = fc[i] < M_max ? fc[i] : M_max;
      const float* pb = fb + (size_t)i * M_max * 2;
      memset(claimed, 0, M_max);
      int nroots_at_start = nroots;
      for (int r = 0; r < nroots_at_start; r++) {
        int rc = root_cam[r], rb = root_blob[r];
        const double* Fm = cm->F + 9 * ((size_t)rc * C + i);
        /* cv.computeCorrespondEpilines on a float32 point (helpers.py:363-364) */
        double x = (double)fb[((size_t)rc * M_max + rb) * 2 + 0];
        double y = (double)fb[((size_t)rc * M_max + rb) * 2 + 1];
        double a = Fm[0] * x + Fm[1] * y + Fm[2];
        double b = Fm[3] * x + Fm[4] * y + Fm[5];
        double c = Fm[6] * x + Fm[7] * y + Fm[8];
        double nu = a * a + b * b;
        nu = nu != 0.0 ? 1.0 / sqrt(nu) : 1.0;
        a = round_f32(cm, a * nu);
        b = round_f32(cm, b * nu);
        c = round_f32(cm, c * nu);
        /* helpers.py:373: |a x + b y + c| / sqrt(a^2 + b^2), strict < gate (helpers.py:375,383) */
        double den = sqrt(a * a + b * b);
        int16_t* h = hits + ((size_t)r * C + i) * M_max;
        int cnt = 0;
        for (int k = 0; k < ni; k++) {
          double px = (double)pb[k * 2 + 0], py = (double)pb[k * 2 + 1];
          double d = fabs(a * px + b * py + c) / den;
          dist[k] = d;
          if (d < gate_px) {
            /* insertion in (distance, index) order: helpers.py:384 with the stable contract */
            int pos = cnt;
            while (pos > 0 && dist[h[pos - 1]] > d) {
              h[pos] = h[pos - 1];
              pos--;
            }
            h[pos] = (int16_t)k;
            cnt++;
          }
        }
        nh[(size_t)r * C + i] = cnt;
        if (cnt > 0) { /* helpers.py:391: removal by value of the closest hit */
          float cx = pb[h[0] * 2 + 0], cy = pb[h[0] * 2 + 1];
          for (int k = 0; k < ni; k++)
            if (pb[k * 2 + 0] == cx && pb[k * 2 + 1] == cy) claimed[k] = 1;
        }
      }
      for (int k = 0; k < ni; k++) /* helpers.py:402-406 */
        if (!claimed[k]) {
          root_cam[nroots] = i;
          root_blob[nroots] = k;
          nroots++;
        }
    }
    if (nroots > K_max) st |= 1;

    int nout = 0;
    for (int r = 0; r < nroots && !(st & 1); r++) { /* helpers.py:410 */
      int rc = root_cam[r];
      long long total = 1;
      int views = 1;
      for (int c = rc + 1; c < C; c++) {
        int cnt = nh[(size_t)r * C + c];
        if (cnt > 0) {
          views++;
          total *= cnt;
          if (total > G_cap) {
            st |= 2;
            break;
          }
        }
      }
      if (st & 2) break;
      if (views <= 1) continue; /* helpers.py:413-414 */
      double best_e = INFINITY, best_X[3] = {0, 0, 0};
      int have = 0;
      for (long long g = 0; g < total; g++) {
        long long rem = g;
        for (int c = 0; c < C; c++) {
          ox[c] = oy[c] = NAN;
          sel[c] = -1;
        }
        sel[rc] = root_blob[r];
        for (int c = rc + 1; c < C; c++) { /* camera rc+1 is the fastest digit (helpers.py:394-400) */
          int cnt = nh[(size_t)r * C + c];
          if (cnt > 0) {
            int dgt = (int)(rem % cnt);
            rem /= cnt;
            sel[c] = hits[((size_t)r * C + c) * M_max + dgt];
          }
        }
        for (int c = 0; c < C; c++)
          if (sel[c] >= 0) {
            ox[c] = (double)fb[((size_t)c * M_max + sel[c]) * 2 + 0];
            oy[c] = (double)fb[((size_t)c * M_max + sel[c]) * 2 + 1];
          }
        double X[3];
        int v = dlt_point(cm, ox, oy, X);
        double e = reproj_error(cm, ox, oy, X, v);
        ncand++;
        if (!have || e < best_e) { /* np.argmin: first minimum (helpers.py:418) */
          have = 1;
          best_e = e;
          memcpy(best_X, X, sizeof X);
          memcpy(best_sel, sel, sizeof(int) * C);
        }
      }
      size_t o = (size_t)f * K_max + nout;
      xyz[o * 3 + 0] = best_X[0];
      xyz[o * 3 + 1] = best_X[1];
      xyz[o * 3 + 2] = best_X[2];
      err[o] = best_e;
      for (int c = 0; c < C; c++) corr[o * C + c] = (int16_t)best_sel[c];
      nout++;
    }
    n_out[f] = st ? 0 : nout;
    status[f] = st;
    if (n_cand) n_cand[f] = (int32_t)(ncand > 2147483647LL ? 2147483647LL : ncand);
  }
  free(root_cam);
  free(root_blob);
  free(hits);
  free(nh);
  free(dist);
  free(claimed);
  free(ox);
  free(sel);
  free(best_sel);
}

/* ------------------------------------------------------------------ bundle adjustment */

/* scipy Rotation.from_rotvec(rv).as_matrix() (helpers.py:258): rotvec -> quaternion
 * (series for angle <= 1e-3) -> matrix. */
static void rotvec_to_matrix(const double* rv, double* R) {
  double angle = sqrt(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
  double scale;
  if (angle <= 1e-3) {
    double a2 = angle * angle;
    scale = 0.5 - a2 / 48 + a2 * a2 / 3840;
  } else {
    scale = sin(angle / 2) / angle;
  }
  double x = scale * rv[0], y = scale * rv[1], z = scale * rv[2], w = cos(angle / 2);
  double x2 = x * x, y2 = y * y, z2 = z * z, w2 = w * w;
  double xy = x * y, zw = z * w, xz = x * z, yw = y * w, yz = y * z, xw = x * w;
  R[0] = x2 - y2 - z2 + w2;
  R[1] = 2 * (xy - zw);
  R[2] = 2 * (xz + yw);
  R[3] = 2 * (xy + zw);
  R[4] = -x2 + y2 - z2 + w2;
  R[5] = 2 * (yz - xw);
  R[6] = 2 * (xz - yw);
  R[7] = 2 * (yz + xw);
  R[8] = -x2 - y2 + z2 + w2;
}

/* helpers.py:264-273 before the float32 cast, for P parameter vectors.
 * params [P][n], n = 1 + 7 (C-1); r [P][N], NaN where < 2 views. */
void mo_ba_residuals(const mo_cams* cm, int P, const double* params, int64_t N, const double* obs,
                     double* r) {
  int C = cm->C, n = 1 + 7 * (C - 1);
  mo_cams tmp = *cm;
  tmp.R = (double*)malloc(sizeof(double) * 9 * C);
  tmp.t = (double*)malloc(sizeof(double) * 3 * C);
  for (int p = 0; p < P; p++) {
    const double* x = params + (size_t)p * n;
    for (int k = 0; k < 9; k++) tmp.R[k] = (k % 4 == 0) ? 1.0 : 0.0; /* helpers.py:250-253 */
    tmp.t[0] = tmp.t[1] = tmp.t[2] = 0.0;
    for (int i = 0; i < C - 1; i++) { /* helpers.py:255-260 */
      rotvec_to_matrix(x + i * 7 + 2, tmp.R + 9 * (i + 1));
      memcpy(tmp.t + 3 * (i + 1), x + i * 7 + 5, 3 * sizeof(double));
    }
    double* xyz = (double*)malloc(sizeof(double) * 3 * N);
    mo_triangulate(&tmp, N, obs, xyz, r + (size_t)p * N);
    free(xyz);
  }
  free(tmp.R);
  free(tmp.t);
}
