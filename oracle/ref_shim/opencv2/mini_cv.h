// mini_cv.h — a LOOK-ALIKE of the part of the OpenCV C++ API that the reference's hot-path sources use (TEST INFRASTRUCTURE).
// OpenCV is not installed in this image and cannot be fetched, so cslam/src/ORBextractor.cpp (and the other reference sources
// compiled by oracle/Makefile.ref) are compiled VERBATIM against these names.  Data structures (Mat with shared, reference-counted
// storage and ROI views, KeyPoint, Point_, Rect_, InputArray/OutputArray) behave like OpenCV's; the image-processing primitives
// (resize, FAST, GaussianBlur, fastAtan2, cvRound) are the restatements of OpenCV 4.2.0 in oracle/cv_prims.h — the [EXT] part that
// stays restated.  Matrix products follow cv::gemm (OpenCV 4.2.0 matmul.simd.hpp, gemmImpl): A*B, A*B+C, -A.t()*B ... are
// lazy expressions evaluated by ONE gemm call; untransposed products with inner dimension 2..4 (equal to the result's width or height) take the
// small-matrix path — FLOAT accumulators summed left to right, then (T)(t*alpha + c*beta) in double —, everything else GEMMSingleMul<T, double>.
// Never used by the product, never shipped.
#pragma once
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "../../cv_prims.h"

#define CV_8U 0
#define CV_8UC1 0
#define CV_32S 4
#define CV_32F 5
#define CV_32FC1 5
#define CV_64F 6
#define CV_64FC1 6
#define CV_PI 3.1415926535897932384626433832795
#define CV_Assert(x) assert(x)

typedef unsigned char uchar;
static inline int cvRound(double v) { return cvprims::cvRoundD(v); }
static inline int cvRound(float v) { return cvprims::cvRoundF(v); }
static inline int cvRound(int v) { return v; }
static inline int cvFloor(double v) { return (int)std::floor(v); }
static inline int cvCeil(double v) { return (int)std::ceil(v); }

namespace cv {

typedef ::uchar uchar;
enum { BORDER_CONSTANT = 0, BORDER_REPLICATE = 1, BORDER_REFLECT = 2, BORDER_WRAP = 3, BORDER_REFLECT_101 = 4, BORDER_REFLECT101 = 4, BORDER_DEFAULT = 4,
       BORDER_ISOLATED = 16 };
enum { INTER_NEAREST = 0, INTER_LINEAR = 1, INTER_CUBIC = 2, INTER_AREA = 3 };
enum { DECOMP_LU = 0, DECOMP_SVD = 1 };
enum { NORM_L2 = 4 };

template <class T> struct Point_ {
  T x, y;
  Point_() : x(0), y(0) {}
  Point_(T x_, T y_) : x(x_), y(y_) {}
  template <class U> Point_(const Point_<U>& o) : x((T)o.x), y((T)o.y) {}   // like cv: saturate_cast; the reference only converts int -> float here
  Point_ operator+(const Point_& o) const { return Point_(x + o.x, y + o.y); }
  Point_ operator-(const Point_& o) const { return Point_(x - o.x, y - o.y); }
  Point_& operator+=(const Point_& o) { x += o.x; y += o.y; return *this; }
  Point_& operator*=(T s) { x *= s; y *= s; return *this; }
  Point_ operator*(T s) const { return Point_(x * s, y * s); }
  bool operator==(const Point_& o) const { return x == o.x && y == o.y; }
};
// Point2i(float, float) as written in ORBextractor.cpp:723-726 truncates like static_cast<int>
template <> template <> inline Point_<int>::Point_(const Point_<float>& o) : x(cvRound(o.x)), y(cvRound(o.y)) {}
typedef Point_<int> Point2i;
typedef Point_<int> Point;
typedef Point_<float> Point2f;
typedef Point_<double> Point2d;
template <class T> struct Point3_ {
  T x, y, z;
  Point3_() : x(0), y(0), z(0) {}
  Point3_(T a, T b, T c) : x(a), y(b), z(c) {}
};
typedef Point3_<float> Point3f;
typedef Point3_<double> Point3d;
template <class T> struct Size_ {
  T width, height;
  Size_() : width(0), height(0) {}
  Size_(T w, T h) : width(w), height(h) {}
  bool operator==(const Size_& o) const { return width == o.width && height == o.height; }
};
typedef Size_<int> Size;
template <class T> struct Rect_ {
  T x, y, width, height;
  Rect_() : x(0), y(0), width(0), height(0) {}
  Rect_(T x_, T y_, T w, T h) : x(x_), y(y_), width(w), height(h) {}
};
typedef Rect_<int> Rect;
struct Range { int start, end; Range(int s, int e) : start(s), end(e) {} };
template <class T, int N> struct Vec { T val[N]; T& operator[](int i) { return val[i]; } const T& operator[](int i) const { return val[i]; } };
typedef Vec<float, 3> Vec3f;
struct Scalar { double val[4]; Scalar(double a = 0, double b = 0, double c = 0, double d = 0) { val[0] = a; val[1] = b; val[2] = c; val[3] = d; } };

struct KeyPoint {
  Point2f pt; float size, angle, response; int octave, class_id;
  KeyPoint() : pt(0, 0), size(0), angle(-1), response(0), octave(0), class_id(-1) {}
  KeyPoint(Point2f p, float s, float a = -1, float r = 0, int o = 0, int c = -1) : pt(p), size(s), angle(a), response(r), octave(o), class_id(c) {}
  KeyPoint(float x, float y, float s, float a = -1, float r = 0, int o = 0, int c = -1) : pt(x, y), size(s), angle(a), response(r), octave(o), class_id(c) {}
};

inline size_t elem_size_of(int type) { return type == CV_8U ? 1 : (type == CV_64F ? 8 : 4); }

template <class T> class Mat_;
// Mat::zeros / ones / eye return an initialiser EXPRESSION, as in OpenCV: assigned to an existing Mat of the same size and type it FILLS
// THAT MAT'S BUFFER IN PLACE (Mat::create is a no-op then) — computeDescriptors (ORBextractor.cpp:1208) relies on this when it zeroes
// the rowRange view it was handed.
struct MatInitExpr { int rows, cols, type; double diag, fill; };
class Mat {
 public:
  int rows = 0, cols = 0;
  uchar* data = nullptr;
  struct Step { size_t v = 0; operator size_t() const { return v; } size_t operator[](int i) const { return i == 0 ? v : 1; } } step;
  int flags_type = CV_8U;
  std::shared_ptr<std::vector<uchar>> buf;   // owner of the pixels (shared by ROI views and shallow copies)

  Mat() {}
  Mat(int r, int c, int type) { create(r, c, type); }
  Mat(Size s, int type) { create(s.height, s.width, type); }
  Mat(int r, int c, int type, const Scalar& s) { create(r, c, type); setTo(s.val[0]); }
  Mat(int r, int c, int type, void* ext, size_t st = 0) : rows(r), cols(c), data((uchar*)ext), flags_type(type) { step.v = st ? st : (size_t)c * elem_size_of(type); }
  Mat(const Mat&) = default;
  Mat& operator=(const Mat&) = default;
  Mat(const Mat& m, const Rect& r) { *this = m(r); }

  void create(int r, int c, int type) {
    if (data && r == rows && c == cols && type == flags_type) return;
    rows = r; cols = c; flags_type = type; step.v = (size_t)c * elem_size_of(type);
    buf = std::make_shared<std::vector<uchar>>((size_t)r * step.v + 16, 0);
    data = buf->data();
  }
  void create(Size s, int type) { create(s.height, s.width, type); }
  void release() { buf.reset(); data = nullptr; rows = cols = 0; }
  bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
  int type() const { return flags_type; }
  int depth() const { return flags_type; }
  int channels() const { return 1; }
  size_t elemSize() const { return elem_size_of(flags_type); }
  size_t elemSize1() const { return elem_size_of(flags_type); }
  size_t total() const { return (size_t)rows * cols; }
  size_t step1(int = 0) const { return step.v / elemSize1(); }
  Size size() const { return Size(cols, rows); }
  bool isContinuous() const { return step.v == (size_t)cols * elemSize() || rows == 1; }
  template <class T> T& at(int r, int c) { return *(T*)(data + (size_t)r * step.v + (size_t)c * sizeof(T)); }
  template <class T> const T& at(int r, int c) const { return *(const T*)(data + (size_t)r * step.v + (size_t)c * sizeof(T)); }
  template <class T> T& at(int i) { return rows == 1 ? at<T>(0, i) : (cols == 1 ? at<T>(i, 0) : at<T>(i / cols, i % cols)); }
  template <class T> const T& at(int i) const { return rows == 1 ? at<T>(0, i) : (cols == 1 ? at<T>(i, 0) : at<T>(i / cols, i % cols)); }
  uchar* ptr(int r = 0) { return data + (size_t)r * step.v; }
  const uchar* ptr(int r = 0) const { return data + (size_t)r * step.v; }
  template <class T> T* ptr(int r = 0) { return (T*)(data + (size_t)r * step.v); }
  template <class T> const T* ptr(int r = 0) const { return (const T*)(data + (size_t)r * step.v); }
  Mat view(int r0, int r1, int c0, int c1) const { Mat m(*this); m.rows = r1 - r0; m.cols = c1 - c0; m.data = data + (size_t)r0 * step.v + (size_t)c0 * elemSize(); return m; }
  Mat rowRange(int a, int b) const { return view(a, b, 0, cols); }
  Mat colRange(int a, int b) const { return view(0, rows, a, b); }
  Mat rowRange(const Range& r) const { return rowRange(r.start, r.end); }
  Mat colRange(const Range& r) const { return colRange(r.start, r.end); }
  Mat row(int i) const { return view(i, i + 1, 0, cols); }
  Mat col(int j) const { return view(0, rows, j, j + 1); }
  Mat operator()(const Rect& r) const { return view(r.y, r.y + r.height, r.x, r.x + r.width); }
  Mat clone() const { Mat m; copyTo(m); return m; }
  void copyTo(Mat& dst) const {
    if (empty()) { dst.release(); return; }
    if (dst.data == data && dst.rows == rows && dst.cols == cols && dst.step.v == step.v) return;
    dst.create(rows, cols, flags_type);
    for (int r = 0; r < rows; r++) std::memmove(dst.ptr(r), ptr(r), (size_t)cols * elemSize());
  }
  Mat reshape(int /*cn*/, int /*rows*/ = 0) const { return *this; }   // single-channel look-alike: an N x 2 CV_32F matrix IS the N two-channel points Frame.cpp:294-296 reshapes to
  void copyTo(Mat&& dst) const { Mat& d = dst; copyTo(d); }   // `A.copyTo(B.rowRange(..).colRange(..))`: OpenCV's OutputArray binds the temporary view; the view shares B's pixels
  double get(int r, int c) const { return flags_type == CV_32F ? (double)at<float>(r, c) : (flags_type == CV_64F ? at<double>(r, c) : (flags_type == CV_32S ? (double)at<int>(r, c) : (double)at<uchar>(r, c))); }
  void set(int r, int c, double v) {
    if (flags_type == CV_32F) at<float>(r, c) = (float)v; else if (flags_type == CV_64F) at<double>(r, c) = v; else if (flags_type == CV_32S) at<int>(r, c) = (int)v; else at<uchar>(r, c) = (uchar)v;
  }
  Mat& setTo(double v) { for (int r = 0; r < rows; r++) for (int c = 0; c < cols; c++) set(r, c, v); return *this; }
  Mat& operator=(const Scalar& s) { return setTo(s.val[0]); }
  void convertTo(Mat& dst, int type, double alpha = 1, double beta = 0) const {
    Mat out(rows, cols, type);
    for (int r = 0; r < rows; r++) for (int c = 0; c < cols; c++) out.set(r, c, get(r, c) * alpha + beta);
    dst = out;
  }
  Mat(const MatInitExpr& e) { *this = e; }
  Mat& operator=(const MatInitExpr& e) {
    create(e.rows, e.cols, e.type);
    for (int r = 0; r < rows; r++) for (int c = 0; c < cols; c++) set(r, c, r == c ? e.diag : e.fill);
    return *this;
  }
  static MatInitExpr zeros(int r, int c, int type) { return MatInitExpr{r, c, type, 0.0, 0.0}; }
  static MatInitExpr zeros(Size s, int type) { return MatInitExpr{s.height, s.width, type, 0.0, 0.0}; }
  static MatInitExpr ones(int r, int c, int type) { return MatInitExpr{r, c, type, 1.0, 1.0}; }
  static MatInitExpr eye(int r, int c, int type) { return MatInitExpr{r, c, type, 1.0, 0.0}; }
  Mat transposed() const { Mat m(cols, rows, flags_type); for (int r = 0; r < rows; r++) for (int c = 0; c < cols; c++) m.set(c, r, get(r, c)); return m; }
  inline struct MatTExpr t() const;
  Mat mul(const Mat& o) const { Mat m(rows, cols, flags_type); for (int r = 0; r < rows; r++) for (int c = 0; c < cols; c++) m.set(r, c, get(r, c) * o.get(r, c)); return m; }
  double dot(const Mat& o) const { double s = 0; for (int r = 0; r < rows; r++) for (int c = 0; c < cols; c++) s += get(r, c) * o.get(r, c); return s; }
  Mat cross(const Mat& o) const {
    Mat m(rows, cols, flags_type);
    const double a0 = at_lin(0), a1 = at_lin(1), a2 = at_lin(2), b0 = o.at_lin(0), b1 = o.at_lin(1), b2 = o.at_lin(2);
    m.set_lin(0, a1 * b2 - a2 * b1); m.set_lin(1, a2 * b0 - a0 * b2); m.set_lin(2, a0 * b1 - a1 * b0);
    return m;
  }
  double at_lin(int i) const { return rows == 1 ? get(0, i) : get(i, 0); }
  void set_lin(int i, double v) { if (rows == 1) set(0, i, v); else set(i, 0, v); }
  Mat inv(int = DECOMP_LU) const {   // Gauss-Jordan with partial pivoting in double
    const int n = rows;
    std::vector<double> a((size_t)n * 2 * n, 0.0);
    for (int r = 0; r < n; r++) { for (int c = 0; c < n; c++) a[(size_t)r * 2 * n + c] = get(r, c); a[(size_t)r * 2 * n + n + r] = 1.0; }
    for (int k = 0; k < n; k++) {
      int p = k; for (int r = k + 1; r < n; r++) if (std::abs(a[(size_t)r * 2 * n + k]) > std::abs(a[(size_t)p * 2 * n + k])) p = r;
      if (p != k) for (int c = 0; c < 2 * n; c++) std::swap(a[(size_t)k * 2 * n + c], a[(size_t)p * 2 * n + c]);
      const double d = a[(size_t)k * 2 * n + k];
      for (int c = 0; c < 2 * n; c++) a[(size_t)k * 2 * n + c] /= d;
      for (int r = 0; r < n; r++) if (r != k) { const double f = a[(size_t)r * 2 * n + k]; if (f != 0) for (int c = 0; c < 2 * n; c++) a[(size_t)r * 2 * n + c] -= f * a[(size_t)k * 2 * n + c]; }
    }
    Mat m(n, n, flags_type);
    for (int r = 0; r < n; r++) for (int c = 0; c < n; c++) m.set(r, c, a[(size_t)r * 2 * n + n + c]);
    return m;
  }
};

// ---- cv::gemm and the MatExprs that end in it (matop.cpp: MatOp_T, MatOp_GEMM; matmul.simd.hpp: gemmImpl) --------------------------------------
inline Mat gemm_eval(const Mat& a, bool ta, const Mat& b, bool tb, double alpha, const Mat* c, double beta) {
  const int arows = ta ? a.cols : a.rows, len = ta ? a.rows : a.cols, bcols = tb ? b.rows : b.cols;
  assert(len == (tb ? b.cols : b.rows));
  Mat m(arows, bcols, a.type());
  auto A = [&](int r, int k) { return ta ? a.get(k, r) : a.get(r, k); };
  auto B = [&](int k, int col) { return tb ? b.get(col, k) : b.get(k, col); };
  const bool small = !ta && !tb && 2 <= len && len <= 4 && (len == bcols || len == arows);      // `flags == 0 && 2 <= len && len <= 4 && (len == d_size.width || ...height)`
  for (int r = 0; r < arows; r++)
    for (int col = 0; col < bcols; col++) {
      const double cv = c ? c->get(r, col) * beta : 0.0;
      if (small && a.type() == CV_32F) {
        float t = (float)A(r, 0) * (float)B(0, col);
        for (int k = 1; k < len; k++) t = t + (float)A(r, k) * (float)B(k, col);                 // `float t0 = a[0]*b[0] + a[1]*b[b_step] + a[2]*b[b_step*2]`
        m.set(r, col, (double)(float)((double)t * alpha + cv));                                  // `d[0] = (float)(t0*alpha + c[0]*beta)` with double alpha, beta
      } else {
        double sacc = 0;
        for (int k = 0; k < len; k++) sacc += A(r, k) * B(k, col);                               // GEMMSingleMul<T, double>
        m.set(r, col, sacc * alpha + cv);
      }
    }
  return m;
}
struct MatGemmExpr {
  Mat a, b; bool ta = false, tb = false; double alpha = 1;
  Mat eval() const { return gemm_eval(a, ta, b, tb, alpha, nullptr, 0); }
  operator Mat() const { return eval(); }
  template <class T> T at(int i) const { return eval().template at<T>(i); }
  template <class T> T at(int r, int c) const { return eval().template at<T>(r, c); }
  Mat clone() const { return eval(); }
  Mat rowRange(int r0, int r1) const { return eval().rowRange(r0, r1); }
  Mat colRange(int c0, int c1) const { return eval().colRange(c0, c1); }
  Mat row(int r) const { return eval().row(r); }
  Mat col(int c) const { return eval().col(c); }
  double dot(const Mat& o) const { return eval().dot(o); }
};
struct MatTExpr {                                     // A.t() * alpha; materialised as transpose + convertTo(alpha) (MatOp_T::assign)
  Mat a; double alpha = 1;
  Mat eval() const {
    Mat m = a.transposed();
    if (alpha != 1) for (int r = 0; r < m.rows; r++) for (int c = 0; c < m.cols; c++) m.set(r, c, m.type() == CV_32F ? (double)((float)m.get(r, c) * (float)alpha) : m.get(r, c) * alpha);
    return m;
  }
  operator Mat() const { return eval(); }
  Mat clone() const { return eval(); }
  Mat rowRange(int r0, int r1) const { return eval().rowRange(r0, r1); }
  Mat colRange(int c0, int c1) const { return eval().colRange(c0, c1); }
  Mat row(int r) const { return eval().row(r); }
  Mat col(int c) const { return eval().col(c); }
  Mat inv(int method = DECOMP_LU) const { return eval().inv(method); }
  template <class T> T at(int r, int c) const { return eval().template at<T>(r, c); }
};
inline MatTExpr Mat::t() const { MatTExpr e; e.a = *this; return e; }
inline MatTExpr operator-(MatTExpr e) { e.alpha = -e.alpha; return e; }
inline MatTExpr operator*(double s, MatTExpr e) { e.alpha *= s; return e; }
inline MatTExpr operator*(MatTExpr e, double s) { e.alpha *= s; return e; }
inline MatGemmExpr operator*(const Mat& a, const Mat& b) { MatGemmExpr g; g.a = a; g.b = b; return g; }
inline MatGemmExpr operator*(const MatTExpr& a, const Mat& b) { MatGemmExpr g; g.a = a.a; g.ta = true; g.alpha = a.alpha; g.b = b; return g; }
inline MatGemmExpr operator*(const Mat& a, const MatTExpr& b) { MatGemmExpr g; g.a = a; g.b = b.a; g.tb = true; g.alpha = b.alpha; return g; }
inline MatGemmExpr operator*(const MatTExpr& a, const MatTExpr& b) { MatGemmExpr g; g.a = a.a; g.ta = true; g.b = b.a; g.tb = true; g.alpha = a.alpha * b.alpha; return g; }
inline MatGemmExpr operator*(const MatGemmExpr& a, const Mat& b) { MatGemmExpr g; g.a = a.eval(); g.b = b; return g; }
inline MatGemmExpr operator*(const Mat& a, const MatGemmExpr& b) { MatGemmExpr g; g.a = a; g.b = b.eval(); return g; }
inline MatGemmExpr operator*(const MatGemmExpr& a, const MatTExpr& b) { MatGemmExpr g; g.a = a.eval(); g.b = b.a; g.tb = true; g.alpha = b.alpha; return g; }
inline MatGemmExpr operator*(MatGemmExpr g, double s) { g.alpha *= s; return g; }
inline MatGemmExpr operator*(double s, MatGemmExpr g) { g.alpha *= s; return g; }
inline MatGemmExpr operator-(MatGemmExpr g) { g.alpha = -g.alpha; return g; }
inline Mat operator+(const MatGemmExpr& g, const Mat& c) { return gemm_eval(g.a, g.ta, g.b, g.tb, g.alpha, &c, 1.0); }          // MatOp_GEMM::add: one gemm with C
inline Mat operator+(const Mat& c, const MatGemmExpr& g) { return gemm_eval(g.a, g.ta, g.b, g.tb, g.alpha, &c, 1.0); }
inline Mat operator-(const MatGemmExpr& g, const Mat& c) { return gemm_eval(g.a, g.ta, g.b, g.tb, g.alpha, &c, -1.0); }
inline Mat operator-(const Mat& c, const MatGemmExpr& g) { return gemm_eval(g.a, g.ta, g.b, g.tb, -g.alpha, &c, 1.0); }
inline Mat operator+(const Mat& a, const Mat& b) { Mat m(a.rows, a.cols, a.type()); for (int r = 0; r < a.rows; r++) for (int c = 0; c < a.cols; c++) m.set(r, c, a.type() == CV_32F ? (double)((float)a.get(r, c) + (float)b.get(r, c)) : a.get(r, c) + b.get(r, c)); return m; }
inline Mat operator-(const Mat& a, const Mat& b) { Mat m(a.rows, a.cols, a.type()); for (int r = 0; r < a.rows; r++) for (int c = 0; c < a.cols; c++) m.set(r, c, a.type() == CV_32F ? (double)((float)a.get(r, c) - (float)b.get(r, c)) : a.get(r, c) - b.get(r, c)); return m; }
inline Mat operator-(const Mat& a) { Mat m(a.rows, a.cols, a.type()); for (int r = 0; r < a.rows; r++) for (int c = 0; c < a.cols; c++) m.set(r, c, -a.get(r, c)); return m; }
// Mat * s and Mat / s are MatExprs that OpenCV evaluates with convertTo(alpha = s resp. 1./s) — or folds into scaleAdd when added to another
// Mat —, and for CV_32F both multiply by the scalar ROUNDED TO FLOAT, in float (cvtScale32f / scaleAdd_32f, baseline build: no FMA)
inline Mat operator*(const Mat& a, double s) {
  Mat m(a.rows, a.cols, a.type());
  for (int r = 0; r < a.rows; r++) for (int c = 0; c < a.cols; c++) m.set(r, c, a.type() == CV_32F ? (double)((float)a.get(r, c) * (float)s) : a.get(r, c) * s);
  return m;
}
inline Mat operator*(double s, const Mat& a) { return a * s; }
inline Mat& operator*=(Mat& a, double s) { a = a * s; return a; }   // (a MatExpr assigned back: new pixels, as cv::Mat::operator*= via MatExpr does)
inline Mat operator/(const Mat& a, double s) { return a * (1. / s); }
inline double norm(const Mat& a) { double s = 0; for (int r = 0; r < a.rows; r++) for (int c = 0; c < a.cols; c++) s += a.get(r, c) * a.get(r, c); return std::sqrt(s); }
inline double norm(const Mat& a, const Mat& b) { return norm(a - b); }
inline std::ostream& operator<<(std::ostream& os, const Mat& m) {
  os << "[";
  for (int r = 0; r < m.rows; r++) { for (int c = 0; c < m.cols; c++) os << (c ? ", " : "") << m.get(r, c); os << (r + 1 < m.rows ? ";\n " : ""); }
  return os << "]";
}

// (cv::Mat_<float>(3, 1) << a, b, c) as used all over the reference
template <class T> struct MatType;
template <> struct MatType<float> { enum { value = CV_32F }; };
template <> struct MatType<double> { enum { value = CV_64F }; };
template <> struct MatType<uchar> { enum { value = CV_8U }; };
template <> struct MatType<int> { enum { value = CV_32S }; };
template <class T> struct MatCommaInit {
  Mat m; int i = 0;
  MatCommaInit(int r, int c) : m(r, c, MatType<T>::value) {}
  MatCommaInit& operator,(T v) { m.at<T>(i / m.cols, i % m.cols) = v; i++; return *this; }
  operator Mat() const { return m; }
};
template <class T> class Mat_ : public Mat {
 public:
  Mat_() {}
  Mat_(int r, int c) : Mat(r, c, MatType<T>::value) {}
  Mat_(const Mat& m) : Mat(m) {}
  T& operator()(int r, int c) { return at<T>(r, c); }
  const T& operator()(int r, int c) const { return at<T>(r, c); }
  MatCommaInit<T> operator<<(T v) { MatCommaInit<T> ci(rows, cols); ci.m = *this; ci.m.template at<T>(0, 0) = v; ci.i = 1; return ci; }
};

class _InputArray {
 protected:
  Mat m_; bool has_ = false;
 public:
  _InputArray() {}
  _InputArray(const Mat& m) : m_(m), has_(true) {}
  Mat getMat(int = -1) const { return m_; }
  bool empty() const { return !has_ || m_.empty(); }
};
typedef const _InputArray& InputArray;
class _OutputArray {
  Mat* pm_ = nullptr;
 public:
  _OutputArray() {}
  _OutputArray(Mat& m) : pm_(&m) {}
  bool needed() const { return pm_ != nullptr; }
  void create(int r, int c, int type) const { if (pm_) pm_->create(r, c, type); }
  void create(Size s, int type) const { if (pm_) pm_->create(s, type); }
  void release() const { if (pm_) pm_->release(); }
  Mat getMat(int = -1) const { return pm_ ? *pm_ : Mat(); }
};
typedef const _OutputArray& OutputArray;
inline _InputArray noArray() { return _InputArray(); }

// ---- image processing: the restated OpenCV 4.2.0 primitives of oracle/cv_prims.h behind the cv:: signatures -----------------------
inline float fastAtan2(float y, float x) { return cvprims::fast_atan2(y, x); }

inline void resize(InputArray src_, OutputArray dst_, Size dsize, double = 0, double = 0, int interpolation = INTER_LINEAR) {
  assert(interpolation == INTER_LINEAR);
  Mat src = src_.getMat();
  assert(src.type() == CV_8U);
  dst_.create(dsize.height, dsize.width, src.type());
  Mat dst = dst_.getMat();
  cvprims::resize_linear_u8(src.data, src.cols, src.rows, (int)src.step.v, dst.data, dst.cols, dst.rows, (int)dst.step.v);
}

inline int borderInterpolate101(int p, int len) { return cvprims::reflect101(p, len); }

// BORDER_REFLECT_101 (+ BORDER_ISOLATED: never look outside the ROI — this implementation never does).  Handles the in-place case of
// ORBextractor::ComputePyramid, where src is the interior ROI of dst.
inline void copyMakeBorder(InputArray src_, OutputArray dst_, int top, int bottom, int left, int right, int borderType, const Scalar& = Scalar()) {
  assert((borderType & ~BORDER_ISOLATED) == BORDER_REFLECT_101);
  Mat src = src_.getMat();
  const size_t es = src.elemSize();
  dst_.create(src.rows + top + bottom, src.cols + left + right, src.type());
  Mat dst = dst_.getMat();
  for (int r = 0; r < src.rows; r++) {   // interior (no-op when src already is dst's interior), then this row's left / right border
    uchar* drow = dst.ptr(r + top);
    if (drow + (size_t)left * es != src.ptr(r)) std::memmove(drow + (size_t)left * es, src.ptr(r), (size_t)src.cols * es);
    for (int c = 0; c < left; c++) std::memcpy(drow + (size_t)c * es, drow + (size_t)(left + borderInterpolate101(c - left, src.cols)) * es, es);
    for (int c = 0; c < right; c++) std::memcpy(drow + (size_t)(left + src.cols + c) * es, drow + (size_t)(left + borderInterpolate101(src.cols + c, src.cols)) * es, es);
  }
  for (int r = 0; r < top; r++) std::memcpy(dst.ptr(r), dst.ptr(top + borderInterpolate101(r - top, src.rows)), (size_t)dst.cols * es);
  for (int r = 0; r < bottom; r++) std::memcpy(dst.ptr(top + src.rows + r), dst.ptr(top + borderInterpolate101(src.rows + r, src.rows)), (size_t)dst.cols * es);
}

inline void GaussianBlur(InputArray src_, OutputArray dst_, Size ksize, double sigmaX, double sigmaY = 0, int borderType = BORDER_DEFAULT) {
  (void)sigmaY;
  assert(ksize.width == 7 && ksize.height == 7 && sigmaX == 2 && (borderType & ~BORDER_ISOLATED) == BORDER_REFLECT_101);
  Mat src = src_.getMat();
  assert(src.type() == CV_8U);
  std::vector<uchar> tmp((size_t)src.rows * src.cols);   // the call in ORBextractor::operator() is in place
  for (int r = 0; r < src.rows; r++) std::memcpy(&tmp[(size_t)r * src.cols], src.ptr(r), (size_t)src.cols);
  dst_.create(src.rows, src.cols, src.type());
  Mat dst = dst_.getMat();
  cvprims::gaussian_blur7(tmp.data(), src.cols, src.rows, src.cols, dst.data, (int)dst.step.v);
}

inline void FAST(InputArray image_, std::vector<KeyPoint>& keypoints, int threshold, bool nonmaxSuppression = true) {
  assert(nonmaxSuppression);
  Mat img = image_.getMat();
  std::vector<cvprims::KP> out;
  cvprims::fast9_16(img.data, img.cols, img.rows, (int)img.step.v, threshold, out);
  keypoints.clear();
  for (const cvprims::KP& k : out) keypoints.push_back(KeyPoint(k.x, k.y, 7.f, -1, k.response));
}

struct KeyPointsFilter {
  static void retainBest(std::vector<KeyPoint>& kps, int n) {   // only the reference's dead ComputeKeyPointsOld calls this
    if (n >= 0 && (int)kps.size() > n) {
      std::stable_sort(kps.begin(), kps.end(), [](const KeyPoint& a, const KeyPoint& b) { return a.response > b.response; });
      kps.resize((size_t)n);
    }
  }
};

// config.h reads conf/config.yaml through cv::FileStorage at static-initialisation time.  A flat "key: value" reader is enough; when
// the file is absent (GPU box: /root/reference does not exist) every value reads as 0 — none of the compiled sources uses them.
class FileNode {
  double v_ = 0; std::string s_;
 public:
  FileNode() {}
  FileNode(double v, const std::string& s) : v_(v), s_(s) {}
  operator double() const { return v_; }
  operator float() const { return (float)v_; }
  operator int() const { return (int)v_; }
  operator std::string() const { return s_; }
  bool empty() const { return s_.empty(); }
  // structured (sequence / map) nodes: DBoW2's YAML vocabulary load/save compiles against these; the pinning harness loads vocabularies through
  // loadFromTextFile only, so reaching them is an error
  FileNode operator[](const char*) const { throw std::runtime_error("mini_cv: structured cv::FileNode access is not provided"); }
  FileNode operator[](const std::string&) const { throw std::runtime_error("mini_cv: structured cv::FileNode access is not provided"); }
  FileNode operator[](int) const { throw std::runtime_error("mini_cv: structured cv::FileNode access is not provided"); }
  size_t size() const { return 0; }
};
class FileStorage {
  std::map<std::string, std::string> kv_;
 public:
  enum { READ = 0, WRITE = 1 };
  FileStorage() {}
  FileStorage(const std::string& path, int) { open(path, READ); }
  bool open(const std::string& path, int) {
    std::ifstream f(path);
    std::string line;
    while (std::getline(f, line)) {
      const size_t c = line.find(':');
      if (c == std::string::npos || line[0] == '%' || line[0] == '#') continue;
      std::string k = line.substr(0, c), v = line.substr(c + 1);
      auto trim = [](std::string& s) { while (!s.empty() && (s.back() == ' ' || s.back() == '\r' || s.back() == '"')) s.pop_back(); size_t i = 0; while (i < s.size() && (s[i] == ' ' || s[i] == '"')) i++; s = s.substr(i); };
      trim(k); trim(v);
      kv_[k] = v;
    }
    return true;
  }
  bool isOpened() const { return true; }
  void release() {}
  template <class T> FileStorage& operator<<(const T&) { throw std::runtime_error("mini_cv: cv::FileStorage writing is not provided"); }
  FileNode operator[](const std::string& k) const {
    auto it = kv_.find(k);
    if (it == kv_.end()) return FileNode();
    return FileNode(std::atof(it->second.c_str()), it->second);
  }
};

}  // namespace cv
