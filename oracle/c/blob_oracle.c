/* TEST INFRASTRUCTURE ONLY (oracle/) -- never linked into the product.
 *
 * Plain-C restatement of the reference's blob-extraction stage:
 *   Cameras._camera_read  computer_code/api/helpers.py:68-82   rot90, make_square (helpers.py:507-523),
 *                         cv.undistort, cv.GaussianBlur (9,9), cv.filter2D 5x5, cv.cvtColor RGB2BGR
 *   Cameras._find_dot     computer_code/api/helpers.py:143-163 grey, threshold 255*0.2, cv.findContours
 *                         RETR_TREE / CHAIN_APPROX_SIMPLE, cv.moments, int() centroids
 * following the OpenCV algorithms as restated in oracle/cv_image_restate.py (PARITY UNPINNED at the OpenCV
 * calls: OpenCV is absent from /root/reference and this image).  Contours use the SEQUENTIAL raster scan
 * of Suzuki-Abe with border marks (OpenCV's icvFetchContour / cvFindNextContour), i.e. a different
 * algorithm from the HIP kernel's parallel cycle following: the parity tests cross-check the two.
 * Used by tests/ at sizes the Python restatement is too slow for and as the CPU "port" baseline.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static int reflect101(int i, int n) {
  if (i < 0) i = -i;
  return i >= n ? 2 * (n - 1) - i : i;
}

/* cv::invert 3x3 closed form (core/src/lapack.cpp) */
static void invert3(const double* S, double* t) {
  double d = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) + S[2] * (S[3] * S[7] - S[4] * S[6]);
  d = 1.0 / d;
  t[0] = (S[4] * S[8] - S[5] * S[7]) * d;
  t[1] = (S[2] * S[7] - S[1] * S[8]) * d;
  t[2] = (S[1] * S[5] - S[2] * S[4]) * d;
  t[3] = (S[5] * S[6] - S[3] * S[8]) * d;
  t[4] = (S[0] * S[8] - S[2] * S[6]) * d;
  t[5] = (S[2] * S[3] - S[0] * S[5]) * d;
  t[6] = (S[3] * S[7] - S[4] * S[6]) * d;
  t[7] = (S[1] * S[6] - S[0] * S[7]) * d;
  t[8] = (S[0] * S[4] - S[1] * S[3]) * d;
}

typedef struct {
  int rows, cols, S, ay, rot;
  int16_t *sx, *sy;
  uint8_t *fx, *fy;
} blob_cam;

/* cv::undistort's stripe-wise initUndistortRectifyMap (imgproc/undistort.dispatch.cpp), CV_16SC2 map */
blob_cam* bo_cam_create(int rows, int cols, const double* K, const double* dist, int rotation) {
  const int S = cols > rows ? cols : rows;
  blob_cam* c = (blob_cam*)calloc(1, sizeof *c);
  c->rows = rows;
  c->cols = cols;
  c->S = S;
  c->ay = (S - rows) / 2;
  c->rot = ((rotation % 4) + 4) % 4;
  c->sx = (int16_t*)malloc(sizeof(int16_t) * S * S);
  c->sy = (int16_t*)malloc(sizeof(int16_t) * S * S);
  c->fx = (uint8_t*)malloc((size_t)S * S);
  c->fy = (uint8_t*)malloc((size_t)S * S);
  const double k1 = dist[0], k2 = dist[1], p1 = dist[2], p2 = dist[3], k3 = dist[4];
  const double fxk = K[0], fyk = K[4], u0 = K[2], v0 = K[5];
  int stripe0 = (1 << 12) / (S > 1 ? S : 1);
  if (stripe0 < 1) stripe0 = 1;
  if (stripe0 > S) stripe0 = S;
  for (int y0 = 0; y0 < S; y0 += stripe0) {
    const int n = stripe0 < S - y0 ? stripe0 : S - y0;
    double Ar[9], ir[9];
    memcpy(Ar, K, sizeof Ar);
    Ar[5] = v0 - y0;
    invert3(Ar, ir);
    for (int i = 0; i < n; i++) {
      double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
      for (int j = 0; j < S; j++, _x += ir[0], _y += ir[3], _w += ir[6]) {
        const double w = 1. / _w, x = _x * w, y = _y * w;
        const double x2 = x * x, y2 = y * y;
        const double r2 = x2 + y2, _2xy = 2 * x * y;
        const double kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((0 * r2 + 0) * r2 + 0) * r2);
        const double xd = (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2) + 0 * r2 + 0 * r2 * r2);
        const double yd = (y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy + 0 * r2 + 0 * r2 * r2);
        const double u = fxk * 1.0 * xd + u0, v = fyk * 1.0 * yd + v0;
        const long iu = lrint(u * 32.0), iv = lrint(v * 32.0);
        const size_t o = (size_t)(y0 + i) * S + j;
        c->sx[o] = (int16_t)(iu >> 5);
        c->sy[o] = (int16_t)(iv >> 5);
        c->fx[o] = (uint8_t)(iu & 31);
        c->fy[o] = (uint8_t)(iv & 31);
      }
    }
  }
  return c;
}

void bo_cam_destroy(blob_cam* c) {
  if (!c) return;
  free(c->sx);
  free(c->sy);
  free(c->fx);
  free(c->fy);
  free(c);
}

/* ---- contours: Suzuki-Abe on the zero-padded binary image, labels in an int32 image */
typedef struct {
  int parent, is_hole;
  long long a00, a10, a01;
  int first_child, next_sibling;
} contour_t;

static const int DY8[8] = {0, -1, -1, -1, 0, 1, 1, 1};
static const int DX8[8] = {1, 1, 0, -1, -1, -1, 0, 1};

static void fetch_contour(int32_t* img, int W, int y0, int x0, int nbd, int is_hole, contour_t* c) {
  int s_end = is_hole ? 0 : 4, s = s_end, y1, x1;
  do {
    s = (s - 1) & 7;
    y1 = y0 + DY8[s];
    x1 = x0 + DX8[s];
  } while (img[y1 * W + x1] == 0 && s != s_end);
  c->a00 = c->a10 = c->a01 = 0;
  if (s == s_end) {
    img[y0 * W + x0] = -nbd;
    return; /* single pixel: a one-vertex polygon, all sums 0 */
  }
  int y3 = y0, x3 = x0;
  /* cv::moments sums over consecutive vertices incl. last -> first; CHAIN_APPROX_SIMPLE drops collinear
     vertices, which leaves these integer sums unchanged, so every border pixel is used as a vertex */
  for (;;) {
    s_end = s;
    int y4, x4;
    for (;;) {
      s++;
      y4 = y3 + DY8[s & 7];
      x4 = x3 + DX8[s & 7];
      if (img[y4 * W + x4] != 0) break;
    }
    s &= 7;
    if ((unsigned)(s - 1) < (unsigned)s_end)
      img[y3 * W + x3] = -nbd;
    else if (img[y3 * W + x3] == 1)
      img[y3 * W + x3] = nbd;
    {
      const long long px = x3 - 1, py = y3 - 1, qx = x4 - 1, qy = y4 - 1; /* original pixel coordinates */
      const long long dxy = px * qy - qx * py;
      c->a00 += dxy;
      c->a10 += dxy * (px + qx);
      c->a01 += dxy * (py + qy);
    }
    if (y4 == y0 && x4 == x0 && y3 == y1 && x3 == x1) break;
    y3 = y4;
    x3 = x4;
    s = (s + 4) & 7;
  }
}

/* mask [S][S] (0 / non-zero) -> centroids in cv.findContours(RETR_TREE) order, skipping m00 == 0.
   Returns the number of centroids (all of them, even beyond cap); *n_contours = contours found. */
static int centroids_from_mask(const uint8_t* mask, int S, int cap, float* out, int* n_contours) {
  const int W = S + 2;
  int32_t* img = (int32_t*)calloc((size_t)W * W, sizeof(int32_t));
  for (int y = 0; y < S; y++)
    for (int x = 0; x < S; x++) img[(y + 1) * W + x + 1] = mask[y * S + x] ? 1 : 0;
  int cap_c = 64, n = 0; /* contour label = index + 2 */
  contour_t* cs = (contour_t*)malloc(sizeof(contour_t) * cap_c);
  for (int y = 1; y <= S; y++) {
    int lnbd = 0, prev = 0;
    for (int x = 1; x <= S + 1; x++) {
      int p = img[y * W + x];
      if (p != prev) {
        int start = 0, is_hole = 0;
        if (prev == 0 && p == 1) start = 1;
        else if (p == 0 && prev >= 1 && x <= S + 1) { start = 1; is_hole = 1; }
        if (start) {
          if (is_hole && prev > 1) lnbd = prev;
          if (n == cap_c) cs = (contour_t*)realloc(cs, sizeof(contour_t) * (cap_c *= 2));
          const int nbd = n + 2;
          int parent = -1;
          if (lnbd) {
            const contour_t* b = &cs[lnbd - 2];
            parent = b->is_hole == is_hole ? b->parent : lnbd - 2; /* Suzuki-Abe's parent table */
          }
          cs[n].parent = parent;
          cs[n].is_hole = is_hole;
          fetch_contour(img, W, y, is_hole ? x - 1 : x, nbd, is_hole, &cs[n]);
          n++;
          lnbd = nbd;
          p = img[y * W + x];
        }
        prev = p;
        if (prev != 0 && prev != 1) lnbd = prev < 0 ? -prev : prev;
      }
    }
  }
  /* cvInsertNodeIntoTree: children at the head; output = pre-order */
  int root_first = -1;
  for (int i = 0; i < n; i++) cs[i].first_child = -1;
  for (int i = 0; i < n; i++) {
    int* head = cs[i].parent < 0 ? &root_first : &cs[cs[i].parent].first_child;
    cs[i].next_sibling = *head;
    *head = i;
  }
  int count = 0, node = root_first;
  while (node >= 0) {
    const contour_t* c = &cs[node];
    if (c->a00 != 0) {
      const double s2 = c->a00 > 0 ? 0.5 : -0.5, s6 = c->a00 > 0 ? 0.16666666666666666666666666666667 : -0.16666666666666666666666666666667;
      const double m00 = (double)c->a00 * s2, m10 = (double)c->a10 * s6, m01 = (double)c->a01 * s6;
      if (count < cap) {
        out[2 * count] = (float)(int)(m10 / m00);
        out[2 * count + 1] = (float)(int)(m01 / m00);
      }
      count++;
    }
    if (c->first_child >= 0) node = c->first_child;
    else {
      while (node >= 0 && cs[node].next_sibling < 0) node = cs[node].parent;
      if (node >= 0) node = cs[node].next_sibling;
    }
  }
  *n_contours = n;
  free(cs);
  free(img);
  return count;
}

/* One raw frame [rows][cols][3] RGB -> processed BGR frame [S][S][3] (may be NULL), centroids. */
int bo_find_dots(const blob_cam* c, const uint8_t* raw, int M_max, float* blobs, uint8_t* processed, int* n_contours) {
  const int S = c->S, rows = c->rows, cols = c->cols, ay = c->ay;
  uint8_t* sq = (uint8_t*)calloc((size_t)S * S * 3, 1);
  uint8_t* und = (uint8_t*)malloc((size_t)S * S * 3);
  uint16_t* hp = (uint16_t*)malloc(sizeof(uint16_t) * S * S * 3);
  uint8_t* bl = (uint8_t*)malloc((size_t)S * S * 3);
  uint8_t* fl = (uint8_t*)malloc((size_t)S * S * 3);
  uint8_t* mask = (uint8_t*)malloc((size_t)S * S);
  /* np.rot90(k) for k in {0, 2} + make_square (helpers.py:507-523) */
  for (int r = 0; r < rows; r++)
    for (int x = 0; x < cols; x++) {
      const uint8_t* p = c->rot == 2 ? raw + ((size_t)(rows - 1 - r) * cols + (cols - 1 - x)) * 3 : raw + ((size_t)r * cols + x) * 3;
      memcpy(sq + ((size_t)(r + ay) * S + x) * 3, p, 3);
    }
  for (int i = 0; i < 8; i++)
    for (int x = 0; x < cols * 3; x++) {
      sq[(size_t)(ay - i - 1) * S * 3 + x] = (uint8_t)((sq[(size_t)ay * S * 3 + x] * (7 - i)) >> 3);
      sq[(size_t)(ay + rows + i) * S * 3 + x] = (uint8_t)((sq[(size_t)(ay + rows - 1) * S * 3 + x] * (7 - i)) >> 3);
    }
  /* cv::remap INTER_LINEAR fixed point, BORDER_CONSTANT 0 */
  for (int y = 0; y < S; y++)
    for (int x = 0; x < S; x++) {
      const size_t o = (size_t)y * S + x;
      const int sx = c->sx[o], sy = c->sy[o], fx = c->fx[o], fy = c->fy[o];
      const int w[4] = {(32 - fx) * (32 - fy) * 32, fx * (32 - fy) * 32, (32 - fx) * fy * 32, fx * fy * 32};
      for (int ch = 0; ch < 3; ch++) {
        int acc = 0;
        if (!(sx >= S || sx + 1 < 0 || sy >= S || sy + 1 < 0))
          for (int t = 0; t < 4; t++) {
            const int yy = sy + (t >> 1), xx = sx + (t & 1);
            const int v = (yy >= 0 && yy < S && xx >= 0 && xx < S) ? sq[((size_t)yy * S + xx) * 3 + ch] : 0;
            acc += v * w[t];
          }
        und[o * 3 + ch] = (uint8_t)((acc + (1 << 14)) >> 15);
      }
    }
  /* cv::GaussianBlur (9,9), sigma 0 -> fixed-point kernel, reflect-101 */
  static const int g[9] = {4, 13, 30, 51, 60, 51, 30, 13, 4};
  for (int y = 0; y < S; y++)
    for (int x = 0; x < S; x++)
      for (int ch = 0; ch < 3; ch++) {
        int acc = 0;
        for (int i = 0; i < 9; i++) acc += g[i] * und[((size_t)y * S + reflect101(x + i - 4, S)) * 3 + ch];
        hp[((size_t)y * S + x) * 3 + ch] = (uint16_t)acc;
      }
  for (int y = 0; y < S; y++)
    for (int x = 0; x < S; x++)
      for (int ch = 0; ch < 3; ch++) {
        unsigned acc = 0;
        for (int i = 0; i < 9; i++) acc += (unsigned)g[i] * hp[((size_t)reflect101(y + i - 4, S) * S + x) * 3 + ch];
        bl[((size_t)y * S + x) * 3 + ch] = (uint8_t)((acc + 32768u) >> 16);
      }
  /* cv::filter2D with the sharpening kernel of helpers.py:76-80, reflect-101, saturate */
  static const int kk[5][5] = {{-2, -1, -1, -1, -2}, {-1, 1, 3, 1, -1}, {-1, 3, 4, 3, -1}, {-1, 1, 3, 1, -1}, {-2, -1, -1, -1, -2}};
  for (int y = 0; y < S; y++)
    for (int x = 0; x < S; x++)
      for (int ch = 0; ch < 3; ch++) {
        int acc = 0;
        for (int i = 0; i < 5; i++)
          for (int j = 0; j < 5; j++)
            acc += kk[i][j] * bl[((size_t)reflect101(y + i - 2, S) * S + reflect101(x + j - 2, S)) * 3 + ch];
        fl[((size_t)y * S + x) * 3 + ch] = (uint8_t)(acc < 0 ? 0 : (acc > 255 ? 255 : acc));
      }
  /* RGB2BGR (helpers.py:82); RGB2GRAY on the swapped frame, threshold 255*0.2 (helpers.py:145-146) */
  for (size_t o = 0; o < (size_t)S * S; o++) {
    const int c0 = fl[o * 3 + 2], c1 = fl[o * 3 + 1], c2 = fl[o * 3]; /* BGR frame channels 0, 1, 2 */
    if (processed) {
      processed[o * 3] = (uint8_t)c0;
      processed[o * 3 + 1] = (uint8_t)c1;
      processed[o * 3 + 2] = (uint8_t)c2;
    }
    const int grey = (c0 * 9798 + c1 * 19235 + c2 * 3735 + (1 << 14)) >> 15;
    mask[o] = grey > 51 ? 255 : 0;
  }
  int nc = 0;
  const int n = centroids_from_mask(mask, S, M_max, blobs, &nc);
  if (n_contours) *n_contours = nc;
  free(sq);
  free(und);
  free(hp);
  free(bl);
  free(fl);
  free(mask);
  return n;
}
