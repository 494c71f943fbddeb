// mini_eigen.h — a LOOK-ALIKE of the part of Eigen 3 that the reference's vendored g2o uses (TEST INFRASTRUCTURE).
//
// Eigen itself is not installed in this image and cannot be fetched, so the reference's g2o
// (/root/reference/cslam/thirdparty/g2o) cannot be compiled against the real thing.  This header provides the same
// names and call signatures with straightforward, eagerly evaluated implementations so that the reference's g2o
// sources compile VERBATIM from where they lie (oracle/Makefile.ref) into oracle/_ref/libg2o_ref.so — the binary the
// oracle restatement (oracle/ba_ref.cpp) is pinned against in tests/test_ref_g2o.py.
//
// What is therefore still a restatement ([EXT] Eigen): the order of additions inside small matrix
// products / dot products (here: k = 0..n-1, sequential), the 3x3 cofactor inverse, Quaternion <-> rotation matrix,
// LLT / LDLT / PartialPivLU, and the sparse SimplicialLDLT (here: minimum-degree ordering + up-looking LDL^T, the
// algorithm Eigen's simplicial Cholesky implements).  Everything ABOVE these primitives — edges, Jacobians, robust
// kernels, quadratic forms, block solver, Schur complement, Levenberg-Marquardt control flow, exponential maps — is the
// reference's own code.
//
// Never used by the product (ccm_slam_amd/), never shipped.
#pragma once
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstddef>
#include <cstring>
#include <iostream>
#include <limits>
#include <memory>
#include <type_traits>
#include <vector>

#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#define EIGEN_WORLD_VERSION 3
#define EIGEN_MAJOR_VERSION 3
#define EIGEN_MINOR_VERSION 7
#define EIGEN_VERSION_AT_LEAST(x, y, z) (EIGEN_WORLD_VERSION > x || (EIGEN_WORLD_VERSION >= x && (EIGEN_MAJOR_VERSION > y || (EIGEN_MAJOR_VERSION >= y && EIGEN_MINOR_VERSION >= z))))

namespace Eigen {

typedef std::ptrdiff_t Index;
const int Dynamic = -1;
enum { ColMajor = 0, RowMajor = 1, AutoAlign = 0, DontAlign = 2 };
enum { Unaligned = 0, Aligned = 16 };
enum { AlignedBit = 0x80 };
enum { Lower = 1, Upper = 2 };
enum ComputationInfo { Success = 0, NumericalIssue = 1, NoConvergence = 2, InvalidInput = 3 };
enum { ComputeEigenvectors = 0x80, EigenvaluesOnly = 0x40 };
enum TransformTraits { Isometry = 0x1, Affine = 0x2, AffineCompact = 0x10 | Affine, Projective = 0x20 };
inline void initParallel() {}

template <class T> using aligned_allocator = std::allocator<T>;

template <class S, int R, int C, int Opt = 0, int MR = R, int MC = C> class Matrix;
template <class S, int R, int C> class View;
template <class Plain, int MapOpt = Unaligned, class Stride = void> class Map;

namespace detail {
constexpr int prod_dim(int a, int b) { return (a == Dynamic || b == Dynamic) ? Dynamic : a * b; }
constexpr int pick(int a, int b) { return a != Dynamic ? a : b; }
}  // namespace detail

template <class D> struct traits;

// ----------------------------------------------------------------------------------------------------------------------
// DenseBase: CRTP interface shared by Matrix and View (Map, blocks, transposes are Views).  Everything that produces a value
// produces a Matrix (eager evaluation): no aliasing hazards, no expression templates.
// ----------------------------------------------------------------------------------------------------------------------
template <class D> class NoAlias;
template <class D> class ArrayWrap;
template <class M> class LLT;
template <class M> class LDLT;
template <class M> class PartialPivLU;

template <class D>
class DenseBase {
 public:
  typedef typename traits<D>::Scalar Scalar;
  typedef typename std::remove_const<Scalar>::type PlainScalar;
  enum { RowsAtCompileTime = traits<D>::Rows, ColsAtCompileTime = traits<D>::Cols,
         SizeAtCompileTime = detail::prod_dim(traits<D>::Rows, traits<D>::Cols), Flags = 0,
         IsVectorAtCompileTime = (traits<D>::Rows == 1 || traits<D>::Cols == 1) };
  typedef Matrix<PlainScalar, traits<D>::Rows, traits<D>::Cols> PlainObject;
  typedef Eigen::Index Index;

  D& derived() { return *static_cast<D*>(this); }
  const D& derived() const { return *static_cast<const D*>(this); }
  Index rows() const { return derived().rows_(); }
  Index cols() const { return derived().cols_(); }
  Index size() const { return rows() * cols(); }

  PlainScalar coeff(Index i, Index j) const { return derived().at(i, j); }
  PlainScalar operator()(Index i, Index j) const { return derived().at(i, j); }
  Scalar& operator()(Index i, Index j) { return derived().ref(i, j); }
  // vector access (row or column vector)
  PlainScalar operator()(Index i) const { return cols() == 1 ? derived().at(i, 0) : derived().at(0, i); }
  Scalar& operator()(Index i) { return cols() == 1 ? derived().ref(i, 0) : derived().ref(0, i); }
  PlainScalar operator[](Index i) const { return (*this)(i); }
  Scalar& operator[](Index i) { return (*this)(i); }
  PlainScalar x() const { return (*this)(0); }
  PlainScalar y() const { return (*this)(1); }
  PlainScalar z() const { return (*this)(2); }
  PlainScalar w() const { return (*this)(3); }
  Scalar& x() { return (*this)(0); }
  Scalar& y() { return (*this)(1); }
  Scalar& z() { return (*this)(2); }
  Scalar& w() { return (*this)(3); }

  PlainObject eval() const { return PlainObject(derived()); }

  // ---- assignment helpers (used by the derived classes' operator=)
  template <class O> void assign_from(const DenseBase<O>& o) {
    assert(rows() == o.rows() && cols() == o.cols());
    for (Index j = 0; j < cols(); j++) for (Index i = 0; i < rows(); i++) derived().ref(i, j) = o.derived().at(i, j);
  }
  template <class O> D& operator+=(const DenseBase<O>& o) {
    assert(rows() == o.rows() && cols() == o.cols());
    for (Index j = 0; j < cols(); j++) for (Index i = 0; i < rows(); i++) derived().ref(i, j) += o.derived().at(i, j);
    return derived();
  }
  template <class O> D& operator-=(const DenseBase<O>& o) {
    assert(rows() == o.rows() && cols() == o.cols());
    for (Index j = 0; j < cols(); j++) for (Index i = 0; i < rows(); i++) derived().ref(i, j) -= o.derived().at(i, j);
    return derived();
  }
  D& operator*=(PlainScalar s) { for (Index j = 0; j < cols(); j++) for (Index i = 0; i < rows(); i++) derived().ref(i, j) *= s; return derived(); }
  D& operator/=(PlainScalar s) { for (Index j = 0; j < cols(); j++) for (Index i = 0; i < rows(); i++) derived().ref(i, j) /= s; return derived(); }

  D& setZero() { fill(PlainScalar(0)); return derived(); }
  D& setOnes() { fill(PlainScalar(1)); return derived(); }
  D& setConstant(PlainScalar v) { fill(v); return derived(); }
  void fill(PlainScalar v) { for (Index j = 0; j < cols(); j++) for (Index i = 0; i < rows(); i++) derived().ref(i, j) = v; }
  D& setIdentity() { for (Index j = 0; j < cols(); j++) for (Index i = 0; i < rows(); i++) derived().ref(i, j) = (i == j) ? PlainScalar(1) : PlainScalar(0); return derived(); }

  NoAlias<D> noalias() { return NoAlias<D>(derived()); }
  // m << a, b, c ...  (row-major fill, as Eigen's CommaInitializer)
  struct CommaInit {
    D& m; Index i;
    CommaInit& operator,(PlainScalar v) { m.ref(i / m.cols_(), i % m.cols_()) = v; i++; return *this; }
  };
  CommaInit operator<<(PlainScalar v) { derived().ref(0, 0) = v; return CommaInit{derived(), 1}; }
  ArrayWrap<D> array() { return ArrayWrap<D>(derived()); }

  // ---- views
  View<Scalar, traits<D>::Cols, traits<D>::Rows> transpose() { return View<Scalar, traits<D>::Cols, traits<D>::Rows>(derived().ptr_(), cols(), rows(), derived().cs_(), derived().rs_()); }
  View<const PlainScalar, traits<D>::Cols, traits<D>::Rows> transpose() const {
    return View<const PlainScalar, traits<D>::Cols, traits<D>::Rows>(derived().ptr_(), cols(), rows(), derived().cs_(), derived().rs_());
  }
  View<Scalar, Dynamic, Dynamic> block(Index r, Index c, Index nr, Index nc) {
    return View<Scalar, Dynamic, Dynamic>(derived().ptr_() + r * derived().rs_() + c * derived().cs_(), nr, nc, derived().rs_(), derived().cs_());
  }
  View<const PlainScalar, Dynamic, Dynamic> block(Index r, Index c, Index nr, Index nc) const {
    return View<const PlainScalar, Dynamic, Dynamic>(derived().ptr_() + r * derived().rs_() + c * derived().cs_(), nr, nc, derived().rs_(), derived().cs_());
  }
  template <int NR, int NC> View<Scalar, NR, NC> block(Index r, Index c) {
    return View<Scalar, NR, NC>(derived().ptr_() + r * derived().rs_() + c * derived().cs_(), NR, NC, derived().rs_(), derived().cs_());
  }
  template <int NR, int NC> View<const PlainScalar, NR, NC> block(Index r, Index c) const {
    return View<const PlainScalar, NR, NC>(derived().ptr_() + r * derived().rs_() + c * derived().cs_(), NR, NC, derived().rs_(), derived().cs_());
  }
  template <int NR, int NC> View<Scalar, NR, NC> topLeftCorner() { return block<NR, NC>(0, 0); }
  template <int NR, int NC> View<const PlainScalar, NR, NC> topLeftCorner() const { return block<NR, NC>(0, 0); }
  View<Scalar, traits<D>::Rows, 1> col(Index c) { return View<Scalar, traits<D>::Rows, 1>(derived().ptr_() + c * derived().cs_(), rows(), 1, derived().rs_(), derived().cs_()); }
  View<const PlainScalar, traits<D>::Rows, 1> col(Index c) const {
    return View<const PlainScalar, traits<D>::Rows, 1>(derived().ptr_() + c * derived().cs_(), rows(), 1, derived().rs_(), derived().cs_());
  }
  View<Scalar, 1, traits<D>::Cols> row(Index r) { return View<Scalar, 1, traits<D>::Cols>(derived().ptr_() + r * derived().rs_(), 1, cols(), derived().rs_(), derived().cs_()); }
  View<const PlainScalar, 1, traits<D>::Cols> row(Index r) const {
    return View<const PlainScalar, 1, traits<D>::Cols>(derived().ptr_() + r * derived().rs_(), 1, cols(), derived().rs_(), derived().cs_());
  }
  // vector segments (column vectors; row vectors go through the same code with swapped strides)
  Index vstride_() const { return cols() == 1 ? derived().rs_() : derived().cs_(); }
  View<Scalar, Dynamic, 1> segment(Index s, Index n) { return View<Scalar, Dynamic, 1>(derived().ptr_() + s * vstride_(), n, 1, vstride_(), 0); }
  View<const PlainScalar, Dynamic, 1> segment(Index s, Index n) const { return View<const PlainScalar, Dynamic, 1>(derived().ptr_() + s * vstride_(), n, 1, vstride_(), 0); }
  template <int N> View<Scalar, N, 1> segment(Index s) { return View<Scalar, N, 1>(derived().ptr_() + s * vstride_(), N, 1, vstride_(), 0); }
  template <int N> View<const PlainScalar, N, 1> segment(Index s) const { return View<const PlainScalar, N, 1>(derived().ptr_() + s * vstride_(), N, 1, vstride_(), 0); }
  template <int N> View<Scalar, N, 1> head() { return segment<N>(0); }
  template <int N> View<const PlainScalar, N, 1> head() const { return segment<N>(0); }
  View<Scalar, Dynamic, 1> head(Index n) { return segment(0, n); }
  View<const PlainScalar, Dynamic, 1> head(Index n) const { return segment(0, n); }
  template <int N> View<Scalar, N, 1> tail() { return segment<N>(size() - N); }
  template <int N> View<const PlainScalar, N, 1> tail() const { return segment<N>(size() - N); }
  View<Scalar, Dynamic, 1> tail(Index n) { return segment(size() - n, n); }
  View<const PlainScalar, Dynamic, 1> tail(Index n) const { return segment(size() - n, n); }
  View<Scalar, Dynamic, 1> diagonal() { return View<Scalar, Dynamic, 1>(derived().ptr_(), std::min(rows(), cols()), 1, derived().rs_() + derived().cs_(), 0); }
  View<const PlainScalar, Dynamic, 1> diagonal() const {
    return View<const PlainScalar, Dynamic, 1>(derived().ptr_(), std::min(rows(), cols()), 1, derived().rs_() + derived().cs_(), 0);
  }

  // ---- reductions (sequential, index order)
  PlainScalar sum() const { PlainScalar s = 0; for (Index j = 0; j < cols(); j++) for (Index i = 0; i < rows(); i++) s += derived().at(i, j); return s; }
  PlainScalar trace() const { PlainScalar s = 0; for (Index i = 0; i < std::min(rows(), cols()); i++) s += derived().at(i, i); return s; }
  PlainScalar squaredNorm() const { PlainScalar s = 0; for (Index j = 0; j < cols(); j++) for (Index i = 0; i < rows(); i++) s += derived().at(i, j) * derived().at(i, j); return s; }
  PlainScalar norm() const { return std::sqrt(squaredNorm()); }
  PlainScalar maxCoeff() const { PlainScalar m = derived().at(0, 0); for (Index j = 0; j < cols(); j++) for (Index i = 0; i < rows(); i++) m = std::max(m, derived().at(i, j)); return m; }
  PlainScalar minCoeff() const { PlainScalar m = derived().at(0, 0); for (Index j = 0; j < cols(); j++) for (Index i = 0; i < rows(); i++) m = std::min(m, derived().at(i, j)); return m; }
  template <class O> PlainScalar dot(const DenseBase<O>& o) const {
    assert(size() == o.size());
    PlainScalar s = 0;
    for (Index i = 0; i < size(); i++) s += (*this)(i) * o(i);
    return s;
  }
  void normalize() { const PlainScalar n = norm(); if (n > PlainScalar(0)) *this /= n; }
  PlainObject normalized() const { PlainObject r(derived()); r.normalize(); return r; }
  template <class O> Matrix<PlainScalar, 3, 1> cross(const DenseBase<O>& o) const {
    Matrix<PlainScalar, 3, 1> r;
    const PlainScalar a0 = (*this)(0), a1 = (*this)(1), a2 = (*this)(2), b0 = o(0), b1 = o(1), b2 = o(2);
    r(0) = a1 * b2 - a2 * b1; r(1) = a2 * b0 - a0 * b2; r(2) = a0 * b1 - a1 * b0;
    return r;
  }
  PlainObject cwiseAbs() const { PlainObject r(derived()); for (Index j = 0; j < cols(); j++) for (Index i = 0; i < rows(); i++) r(i, j) = std::abs(r(i, j)); return r; }
  template <class O> PlainObject cwiseProduct(const DenseBase<O>& o) const {
    PlainObject r(derived()); for (Index j = 0; j < cols(); j++) for (Index i = 0; i < rows(); i++) r(i, j) *= o.derived().at(i, j); return r;
  }
  bool allFinite() const { for (Index j = 0; j < cols(); j++) for (Index i = 0; i < rows(); i++) if (!std::isfinite(derived().at(i, j))) return false; return true; }

  // ---- small dense algebra
  PlainScalar determinant() const;
  PlainObject inverse() const;
  LLT<PlainObject> llt() const;
  LDLT<PlainObject> ldlt() const;
  PartialPivLU<PlainObject> lu() const;
  PartialPivLU<PlainObject> partialPivLu() const;
};

template <class D> using MatrixBase = DenseBase<D>;
template <class D> using EigenBase = DenseBase<D>;

// ----------------------------------------------------------------------------------------------------------------------
// storage of Matrix
// ----------------------------------------------------------------------------------------------------------------------
namespace detail {
template <class S, int R, int C, bool Dyn = (R == Dynamic || C == Dynamic)> struct Storage;
template <class S, int R, int C> struct Storage<S, R, C, false> {
  S m[R * C > 0 ? R * C : 1];
  Storage() { for (int i = 0; i < R * C; i++) m[i] = S(0); }
  S* data() { return m; }
  const S* data() const { return m; }
  Index rows() const { return R; }
  Index cols() const { return C; }
  void resize(Index r, Index c) { (void)r; (void)c; assert(r == R && c == C); }
};
template <class S, int R, int C> struct Storage<S, R, C, true> {
  std::vector<S> m;
  Index r_ = (R == Dynamic ? 0 : R), c_ = (C == Dynamic ? 0 : C);
  S* data() { return m.data(); }
  const S* data() const { return m.data(); }
  Index rows() const { return r_; }
  Index cols() const { return c_; }
  void resize(Index r, Index c) { if (r != r_ || c != c_) { r_ = r; c_ = c; m.assign((size_t)(r * c), S(0)); } }
};
}  // namespace detail

template <class S, int R, int C, int Opt, int MR, int MC> struct traits<Matrix<S, R, C, Opt, MR, MC>> {
  typedef S Scalar;
  enum { Rows = R, Cols = C };
};
template <class S, int R, int C> struct traits<View<S, R, C>> {
  typedef S Scalar;
  enum { Rows = R, Cols = C };
};
template <class P, int O, class St> struct traits<Map<P, O, St>> {
  typedef typename std::conditional<std::is_const<P>::value, const typename P::Scalar, typename P::Scalar>::type Scalar;
  enum { Rows = P::RowsAtCompileTime, Cols = P::ColsAtCompileTime };
};

template <class S, int R, int C, int Opt, int MR, int MC>
class Matrix : public DenseBase<Matrix<S, R, C, Opt, MR, MC>> {
  detail::Storage<S, R, C> st_;

 public:
  typedef DenseBase<Matrix> Base;
  typedef S Scalar;
  typedef S RealScalar;
  typedef Eigen::Index Index;
  typedef Map<Matrix, Unaligned> MapType;
  typedef Map<const Matrix, Unaligned> ConstMapType;
  typedef Map<Matrix, Aligned> AlignedMapType;
  typedef Map<const Matrix, Aligned> ConstAlignedMapType;
  using Base::operator();
  using Base::operator+=;
  using Base::operator-=;

  // DenseBase plumbing (column major)
  Index rows_() const { return st_.rows(); }
  Index cols_() const { return st_.cols(); }
  S at(Index i, Index j) const { return st_.data()[i + j * st_.rows()]; }
  S& ref(Index i, Index j) { return st_.data()[i + j * st_.rows()]; }
  S* ptr_() { return st_.data(); }
  const S* ptr_() const { return st_.data(); }
  Index rs_() const { return 1; }
  Index cs_() const { return st_.rows(); }

  S* data() { return st_.data(); }
  const S* data() const { return st_.data(); }
  Index innerStride() const { return 1; }
  Index outerStride() const { return st_.rows(); }

  Matrix() {}
  Matrix(const Matrix& o) = default;
  Matrix& operator=(const Matrix& o) = default;
  // Matrix(n): dynamic vector of size n  |  fixed 1x1: the coefficient
  template <class T, typename std::enable_if<std::is_arithmetic<T>::value, int>::type = 0>
  explicit Matrix(T n) {
    if (R == Dynamic || C == Dynamic) st_.resize(R == Dynamic ? (Index)n : R, C == Dynamic ? (C == Dynamic && R != Dynamic ? (Index)n : 1) : C);
    else st_.data()[0] = (S)n;
  }
  // Matrix(a, b): fixed 2-vector: coefficients  |  otherwise (rows, cols)
  template <class T0, class T1, typename std::enable_if<std::is_arithmetic<T0>::value && std::is_arithmetic<T1>::value, int>::type = 0>
  Matrix(T0 a, T1 b) {
    if (R != Dynamic && C != Dynamic && R * C == 2) { st_.data()[0] = (S)a; st_.data()[1] = (S)b; }
    else st_.resize((Index)a, (Index)b);
  }
  Matrix(S a, S b, S c) { static_assert(R * C == 3, "3-vector"); st_.data()[0] = a; st_.data()[1] = b; st_.data()[2] = c; }
  Matrix(S a, S b, S c, S d) { static_assert(R * C == 4, "4-vector"); st_.data()[0] = a; st_.data()[1] = b; st_.data()[2] = c; st_.data()[3] = d; }
  explicit Matrix(const S* p) { for (Index i = 0; i < st_.rows() * st_.cols(); i++) st_.data()[i] = p[i]; }
  template <class O> Matrix(const DenseBase<O>& o) { st_.resize(o.rows(), o.cols()); this->assign_from(o); }
  template <class O> Matrix& operator=(const DenseBase<O>& o) {
    if ((const void*)this == (const void*)&o) return *this;
    Matrix tmp; tmp.st_.resize(o.rows(), o.cols()); tmp.assign_from(o);   // via a temporary: o may view this matrix (A = A.transpose() ...)
    st_ = tmp.st_;
    return *this;
  }

  void resize(Index r, Index c) { st_.resize(r, c); }
  void resize(Index n) { if (C == 1 || (R == Dynamic && C != 1 && C != Dynamic)) st_.resize(n, C == Dynamic ? 1 : C); else st_.resize(R == Dynamic ? 1 : R, n); }
  void conservativeResize(Index r, Index c) {
    Matrix old(*this); st_.resize(r, c); this->setZero();
    for (Index j = 0; j < std::min(c, old.cols()); j++) for (Index i = 0; i < std::min(r, old.rows()); i++) ref(i, j) = old.at(i, j);
  }
  void swap(Matrix& o) { std::swap(st_, o.st_); }

  static Matrix Zero() { Matrix m; m.setZero(); return m; }
  static Matrix Zero(Index n) { Matrix m; m.resize(n); m.setZero(); return m; }
  static Matrix Zero(Index r, Index c) { Matrix m; m.resize(r, c); m.setZero(); return m; }
  static Matrix Ones() { Matrix m; m.setOnes(); return m; }
  static Matrix Ones(Index r, Index c) { Matrix m; m.resize(r, c); m.setOnes(); return m; }
  static Matrix Constant(S v) { Matrix m; m.fill(v); return m; }
  static Matrix Constant(Index r, Index c, S v) { Matrix m; m.resize(r, c); m.fill(v); return m; }
  static Matrix Identity() { Matrix m; m.setIdentity(); return m; }
  static Matrix Identity(Index r, Index c) { Matrix m; m.resize(r, c); m.setIdentity(); return m; }
  static Matrix UnitX() { Matrix m; m.setZero(); m(0) = 1; return m; }
  static Matrix UnitY() { Matrix m; m.setZero(); m(1) = 1; return m; }
  static Matrix UnitZ() { Matrix m; m.setZero(); m(2) = 1; return m; }
};

// strided, non-owning window on someone else's coefficients: Map, block, col/row, segment, transpose, diagonal
template <class S, int R, int C>
class View : public DenseBase<View<S, R, C>> {
 protected:
  S* p_; Index r_, c_, rs_v, cs_v;

 public:
  typedef DenseBase<View> Base;
  typedef typename std::remove_const<S>::type PlainScalar;
  using Base::operator();
  using Base::operator+=;
  using Base::operator-=;
  View(S* p, Index r, Index c, Index rs, Index cs) : p_(p), r_(r), c_(c), rs_v(rs), cs_v(cs) {}
  View(const View&) = default;
  // a mutable view converts to a const view
  template <class S2, typename std::enable_if<std::is_same<const S2, S>::value, int>::type = 0>
  View(const View<S2, R, C>& o) : p_(o.ptr_()), r_(o.rows()), c_(o.cols()), rs_v(o.rs_()), cs_v(o.cs_()) {}
  Index rows_() const { return r_; }
  Index cols_() const { return c_; }
  PlainScalar at(Index i, Index j) const { return p_[i * rs_v + j * cs_v]; }
  S& ref(Index i, Index j) { return p_[i * rs_v + j * cs_v]; }
  S* ptr_() const { return p_; }
  Index rs_() const { return rs_v; }
  Index cs_() const { return cs_v; }
  S* data() const { return p_; }
  // writes go THROUGH the view (a temporary protects against overlapping source and destination)
  View& operator=(const View& o) { Matrix<PlainScalar, R, C> t(o); this->assign_from(t); return *this; }
  template <class O> View& operator=(const DenseBase<O>& o) { Matrix<PlainScalar, R, C> t(o); this->assign_from(t); return *this; }
};

template <class P, int MapOpt, class Stride>
class Map : public View<typename traits<Map<P, MapOpt, Stride>>::Scalar, P::RowsAtCompileTime, P::ColsAtCompileTime> {
 public:
  typedef typename traits<Map>::Scalar S;
  typedef View<S, P::RowsAtCompileTime, P::ColsAtCompileTime> V;
  enum { R = P::RowsAtCompileTime, C = P::ColsAtCompileTime };
  using V::operator=;
  using V::operator();
  explicit Map(S* p) : V(p, R, C, 1, R) { static_assert(R != Dynamic && C != Dynamic, "fixed-size Map"); }
  Map(S* p, Index n) : V(p, C == 1 ? n : (R == Dynamic ? n : R), C == 1 ? 1 : (R == Dynamic ? C : n), 1, C == 1 ? n : (R == Dynamic ? n : R)) {}
  Map(S* p, Index r, Index c) : V(p, r, c, 1, r) {}
  Map(const Map&) = default;
  Map& operator=(const Map& o) { V::operator=(static_cast<const V&>(o)); return *this; }
};

template <class D>
class NoAlias {
  D& d_;
 public:
  explicit NoAlias(D& d) : d_(d) {}
  template <class O> D& operator=(const DenseBase<O>& o) { d_ = o; return d_; }
  template <class O> D& operator+=(const DenseBase<O>& o) { d_ += o; return d_; }
  template <class O> D& operator-=(const DenseBase<O>& o) { d_ -= o; return d_; }
};
template <class D>
class ArrayWrap {
  D& d_;
 public:
  explicit ArrayWrap(D& d) : d_(d) {}
  ArrayWrap& operator+=(typename DenseBase<D>::PlainScalar s) { for (Index j = 0; j < d_.cols(); j++) for (Index i = 0; i < d_.rows(); i++) d_.ref(i, j) += s; return *this; }
  ArrayWrap& operator-=(typename DenseBase<D>::PlainScalar s) { for (Index j = 0; j < d_.cols(); j++) for (Index i = 0; i < d_.rows(); i++) d_.ref(i, j) -= s; return *this; }
  ArrayWrap& operator*=(typename DenseBase<D>::PlainScalar s) { for (Index j = 0; j < d_.cols(); j++) for (Index i = 0; i < d_.rows(); i++) d_.ref(i, j) *= s; return *this; }
};

// ----------------------------------------------------------------------------------------------------------------------
// arithmetic (all results are plain matrices)
// ----------------------------------------------------------------------------------------------------------------------
#define ME_PLAIN(A) Matrix<typename DenseBase<A>::PlainScalar, traits<A>::Rows, traits<A>::Cols>
template <class A, class B> Matrix<typename DenseBase<A>::PlainScalar, detail::pick(traits<A>::Rows, traits<B>::Rows), detail::pick(traits<A>::Cols, traits<B>::Cols)>
operator+(const DenseBase<A>& a, const DenseBase<B>& b) {
  Matrix<typename DenseBase<A>::PlainScalar, detail::pick(traits<A>::Rows, traits<B>::Rows), detail::pick(traits<A>::Cols, traits<B>::Cols)> r(a);
  r += b; return r;
}
template <class A, class B> Matrix<typename DenseBase<A>::PlainScalar, detail::pick(traits<A>::Rows, traits<B>::Rows), detail::pick(traits<A>::Cols, traits<B>::Cols)>
operator-(const DenseBase<A>& a, const DenseBase<B>& b) {
  Matrix<typename DenseBase<A>::PlainScalar, detail::pick(traits<A>::Rows, traits<B>::Rows), detail::pick(traits<A>::Cols, traits<B>::Cols)> r(a);
  r -= b; return r;
}
template <class A> ME_PLAIN(A) operator-(const DenseBase<A>& a) { ME_PLAIN(A) r(a); r *= typename DenseBase<A>::PlainScalar(-1); return r; }
template <class A, class T, typename std::enable_if<std::is_arithmetic<T>::value, int>::type = 0>
ME_PLAIN(A) operator*(const DenseBase<A>& a, T s) { ME_PLAIN(A) r(a); r *= (typename DenseBase<A>::PlainScalar)s; return r; }
template <class A, class T, typename std::enable_if<std::is_arithmetic<T>::value, int>::type = 0>
ME_PLAIN(A) operator*(T s, const DenseBase<A>& a) {
  ME_PLAIN(A) r(a);
  for (Index j = 0; j < r.cols(); j++) for (Index i = 0; i < r.rows(); i++) r(i, j) = (typename DenseBase<A>::PlainScalar)s * r(i, j);
  return r;
}
template <class A, class T, typename std::enable_if<std::is_arithmetic<T>::value, int>::type = 0>
ME_PLAIN(A) operator/(const DenseBase<A>& a, T s) { ME_PLAIN(A) r(a); r /= (typename DenseBase<A>::PlainScalar)s; return r; }
template <class A, class B> Matrix<typename DenseBase<A>::PlainScalar, traits<A>::Rows, traits<B>::Cols> operator*(const DenseBase<A>& a, const DenseBase<B>& b) {
  assert(a.cols() == b.rows());
  Matrix<typename DenseBase<A>::PlainScalar, traits<A>::Rows, traits<B>::Cols> r;
  r.resize(a.rows(), b.cols());
  const Index K = a.cols();
  for (Index j = 0; j < b.cols(); j++)
    for (Index i = 0; i < a.rows(); i++) {
      typename DenseBase<A>::PlainScalar s = 0;
      for (Index k = 0; k < K; k++) s += a.derived().at(i, k) * b.derived().at(k, j);
      r(i, j) = s;
    }
  return r;
}
template <class A, class B> bool operator==(const DenseBase<A>& a, const DenseBase<B>& b) {
  if (a.rows() != b.rows() || a.cols() != b.cols()) return false;
  for (Index j = 0; j < a.cols(); j++) for (Index i = 0; i < a.rows(); i++) if (a.derived().at(i, j) != b.derived().at(i, j)) return false;
  return true;
}
template <class A, class B> bool operator!=(const DenseBase<A>& a, const DenseBase<B>& b) { return !(a == b); }
template <class A> std::ostream& operator<<(std::ostream& os, const DenseBase<A>& a) {
  for (Index i = 0; i < a.rows(); i++) {
    for (Index j = 0; j < a.cols(); j++) os << (j ? " " : "") << a.derived().at(i, j);
    if (i + 1 < a.rows()) os << "\n";
  }
  return os;
}

// ----------------------------------------------------------------------------------------------------------------------
// decompositions
// ----------------------------------------------------------------------------------------------------------------------
template <class M>
class LLT {
  M L_; bool ok_ = true;
 public:
  typedef typename M::Scalar S;
  LLT() {}
  template <class O> explicit LLT(const DenseBase<O>& a) { compute(a); }
  template <class O> LLT& compute(const DenseBase<O>& a) {
    const Index n = a.rows();
    L_ = a; ok_ = true;
    for (Index j = 0; j < n; j++) {
      S d = L_(j, j);
      for (Index k = 0; k < j; k++) d -= L_(j, k) * L_(j, k);
      if (!(d > S(0))) { ok_ = false; return *this; }
      d = std::sqrt(d); L_(j, j) = d;
      for (Index i = j + 1; i < n; i++) {
        S s = L_(i, j);
        for (Index k = 0; k < j; k++) s -= L_(i, k) * L_(j, k);
        L_(i, j) = s / d;
      }
    }
    for (Index j = 0; j < n; j++) for (Index i = 0; i < j; i++) L_(i, j) = S(0);
    return *this;
  }
  ComputationInfo info() const { return ok_ ? Success : NumericalIssue; }
  const M& matrixL() const { return L_; }
  template <class B> Matrix<S, traits<B>::Rows, traits<B>::Cols> solve(const DenseBase<B>& b) const {
    Matrix<S, traits<B>::Rows, traits<B>::Cols> x(b);
    const Index n = L_.rows();
    for (Index c = 0; c < x.cols(); c++) {
      for (Index i = 0; i < n; i++) { S s = x(i, c); for (Index k = 0; k < i; k++) s -= L_(i, k) * x(k, c); x(i, c) = s / L_(i, i); }
      for (Index i = n - 1; i >= 0; i--) { S s = x(i, c); for (Index k = i + 1; k < n; k++) s -= L_(k, i) * x(k, c); x(i, c) = s / L_(i, i); }
    }
    return x;
  }
};

// Robust Cholesky with diagonal pivoting, P A P^T = L D L^T — the algorithm of Eigen::LDLT (largest remaining |diagonal| first).
template <class M>
class LDLT {
  M m_; std::vector<Index> tr_; int sign_ = 0; bool init_ = false; ComputationInfo info_ = Success;   // sign_: +1 PSD, -1 NSD, 0 indefinite / zero
 public:
  typedef typename M::Scalar S;
  LDLT() {}
  template <class O> explicit LDLT(const DenseBase<O>& a) { compute(a); }
  template <class O> LDLT& compute(const DenseBase<O>& a) {
    const Index n = a.rows();
    m_ = a; tr_.assign((size_t)n, 0); init_ = true; info_ = Success;
    enum { ZeroSign = 0, PositiveSemiDef = 1, NegativeSemiDef = -1, Indefinite = 2 };
    int sign = ZeroSign;
    bool found_zero_pivot = false;
    for (Index k = 0; k < n; k++) {
      Index piv = k; S big = std::abs(m_(k, k));
      for (Index i = k + 1; i < n; i++) if (std::abs(m_(i, i)) > big) { big = std::abs(m_(i, i)); piv = i; }
      tr_[(size_t)k] = piv;
      if (piv != k) {   // symmetric row/column interchange on the lower triangle
        for (Index j = 0; j < k; j++) std::swap(m_(k, j), m_(piv, j));
        for (Index i = piv + 1; i < n; i++) std::swap(m_(i, k), m_(i, piv));
        std::swap(m_(k, k), m_(piv, piv));
        for (Index i = k + 1; i < piv; i++) std::swap(m_(i, k), m_(piv, i));
      }
      // partition: A10 = m(k, 0..k), A20 = m(k+1.., 0..k), A21 = m(k+1.., k)
      if (k > 0) {
        std::vector<S> temp((size_t)k);
        for (Index j = 0; j < k; j++) temp[(size_t)j] = m_(j, j) * m_(k, j);
        S s = 0;
        for (Index j = 0; j < k; j++) s += m_(k, j) * temp[(size_t)j];
        m_(k, k) -= s;
        for (Index i = k + 1; i < n; i++) {
          S t = 0;
          for (Index j = 0; j < k; j++) t += m_(i, j) * temp[(size_t)j];
          m_(i, k) -= t;
        }
      }
      const S d = m_(k, k);
      const bool pivot_is_valid = std::abs(d) > S(0);
      if (k == 0 && !pivot_is_valid) { sign = ZeroSign; for (Index j = 0; j < n; j++) tr_[(size_t)j] = j; break; }   // the whole matrix is zero
      if (pivot_is_valid) for (Index i = k + 1; i < n; i++) m_(i, k) /= d;
      else for (Index i = k + 1; i < n; i++) m_(i, k) = S(0);
      if (found_zero_pivot && pivot_is_valid) sign = Indefinite;
      else if (!pivot_is_valid) found_zero_pivot = true;
      if (sign == PositiveSemiDef) { if (d < S(0)) sign = Indefinite; }
      else if (sign == NegativeSemiDef) { if (d > S(0)) sign = Indefinite; }
      else if (sign == ZeroSign) { if (d > S(0)) sign = PositiveSemiDef; else if (d < S(0)) sign = NegativeSemiDef; }
    }
    sign_ = sign;
    return *this;
  }
  bool isPositive() const { return sign_ == 1 || sign_ == 0; }
  bool isNegative() const { return sign_ == -1 || sign_ == 0; }
  ComputationInfo info() const { return info_; }
  template <class B> Matrix<S, traits<B>::Rows, traits<B>::Cols> solve(const DenseBase<B>& b) const {
    Matrix<S, traits<B>::Rows, traits<B>::Cols> x(b);
    const Index n = m_.rows();
    for (Index c = 0; c < x.cols(); c++) {
      for (Index k = 0; k < n; k++) std::swap(x(k, c), x(tr_[(size_t)k], c));                                          // P b
      for (Index i = 0; i < n; i++) { S s = x(i, c); for (Index k = 0; k < i; k++) s -= m_(i, k) * x(k, c); x(i, c) = s; }   // L^-1
      // D^-1 with Eigen's tolerance: entries with |d| <= 1 / highest are treated as zero (pseudo-inverse)
      const S tol = S(1) / std::numeric_limits<S>::max();
      for (Index i = 0; i < n; i++) { const S d = m_(i, i); x(i, c) = (std::abs(d) > tol) ? x(i, c) / d : S(0); }
      for (Index i = n - 1; i >= 0; i--) { S s = x(i, c); for (Index k = i + 1; k < n; k++) s -= m_(k, i) * x(k, c); x(i, c) = s; }   // L^-T
      for (Index k = n - 1; k >= 0; k--) std::swap(x(k, c), x(tr_[(size_t)k], c));                                     // P^T
    }
    return x;
  }
};

template <class M>
class PartialPivLU {
  M lu_; std::vector<Index> perm_; int det_sign_ = 1;
 public:
  typedef typename M::Scalar S;
  template <class O> explicit PartialPivLU(const DenseBase<O>& a) {
    lu_ = a;
    const Index n = lu_.rows();
    perm_.resize((size_t)n);
    for (Index k = 0; k < n; k++) {
      Index piv = k; S big = std::abs(lu_(k, k));
      for (Index i = k + 1; i < n; i++) if (std::abs(lu_(i, k)) > big) { big = std::abs(lu_(i, k)); piv = i; }
      perm_[(size_t)k] = piv;
      if (piv != k) { for (Index j = 0; j < n; j++) std::swap(lu_(k, j), lu_(piv, j)); det_sign_ = -det_sign_; }
      if (lu_(k, k) != S(0)) {
        for (Index i = k + 1; i < n; i++) lu_(i, k) /= lu_(k, k);
        for (Index j = k + 1; j < n; j++) for (Index i = k + 1; i < n; i++) lu_(i, j) -= lu_(i, k) * lu_(k, j);
      }
    }
  }
  S determinant() const { S d = S(det_sign_); for (Index i = 0; i < lu_.rows(); i++) d *= lu_(i, i); return d; }
  template <class B> Matrix<S, traits<B>::Rows, traits<B>::Cols> solve(const DenseBase<B>& b) const {
    Matrix<S, traits<B>::Rows, traits<B>::Cols> x(b);
    const Index n = lu_.rows();
    for (Index c = 0; c < x.cols(); c++) {
      for (Index k = 0; k < n; k++) std::swap(x(k, c), x(perm_[(size_t)k], c));
      for (Index i = 0; i < n; i++) { S s = x(i, c); for (Index k = 0; k < i; k++) s -= lu_(i, k) * x(k, c); x(i, c) = s; }
      for (Index i = n - 1; i >= 0; i--) { S s = x(i, c); for (Index k = i + 1; k < n; k++) s -= lu_(i, k) * x(k, c); x(i, c) = s / lu_(i, i); }
    }
    return x;
  }
  M inverse() const { M I; I.resize(lu_.rows(), lu_.cols()); I.setIdentity(); return M(solve(I)); }
};

namespace detail {
template <class T, class M> inline T cof3(const M& m, int i, int j) {   // Eigen's cofactor_3x3<i, j>
  const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
  return m(i1, j1) * m(i2, j2) - m(i1, j2) * m(i2, j1);
}
}  // namespace detail

template <class D> typename DenseBase<D>::PlainScalar DenseBase<D>::determinant() const {
  typedef PlainScalar T;
  const Index n = rows();
  assert(n == cols());
  const D& m = derived();
  if (n == 1) return m.at(0, 0);
  if (n == 2) return m.at(0, 0) * m.at(1, 1) - m.at(1, 0) * m.at(0, 1);
  if (n == 3) {   // Eigen's bruteforce_det3_helper expansion along the first row
    auto h = [&](int a, int b, int c) { return m.at(0, a) * (m.at(1, b) * m.at(2, c) - m.at(1, c) * m.at(2, b)); };
    return h(0, 1, 2) - h(1, 0, 2) + h(2, 0, 1);
  }
  return PartialPivLU<Matrix<T, Dynamic, Dynamic>>(derived()).determinant();
}
template <class D> typename DenseBase<D>::PlainObject DenseBase<D>::inverse() const {
  typedef PlainScalar T;
  const Index n = rows();
  assert(n == cols());
  PlainObject r; r.resize(n, n);
  const D& m = derived();
  if (n == 1) { r(0, 0) = T(1) / m.at(0, 0); return r; }
  if (n == 2) {
    const T invdet = T(1) / determinant();
    r(0, 0) = m.at(1, 1) * invdet; r(1, 0) = -m.at(1, 0) * invdet; r(0, 1) = -m.at(0, 1) * invdet; r(1, 1) = m.at(0, 0) * invdet;
    return r;
  }
  if (n == 3) {   // Eigen's compute_inverse<.., 3>: cofactors of the first column give the determinant, result = adjugate * (1 / det)
    auto mm = [&](int i, int j) { return m.at(i, j); };
    const T c00 = detail::cof3<T>(mm, 0, 0), c10 = detail::cof3<T>(mm, 1, 0), c20 = detail::cof3<T>(mm, 2, 0);
    const T det = (c00 * m.at(0, 0) + c10 * m.at(1, 0)) + c20 * m.at(2, 0);
    const T invdet = T(1) / det;
    r(0, 0) = c00 * invdet; r(0, 1) = c10 * invdet; r(0, 2) = c20 * invdet;
    r(1, 0) = detail::cof3<T>(mm, 0, 1) * invdet; r(1, 1) = detail::cof3<T>(mm, 1, 1) * invdet; r(1, 2) = detail::cof3<T>(mm, 2, 1) * invdet;
    r(2, 0) = detail::cof3<T>(mm, 0, 2) * invdet; r(2, 1) = detail::cof3<T>(mm, 1, 2) * invdet; r(2, 2) = detail::cof3<T>(mm, 2, 2) * invdet;
    return r;
  }
  Matrix<T, Dynamic, Dynamic> I = Matrix<T, Dynamic, Dynamic>::Identity(n, n);
  return PlainObject(PartialPivLU<Matrix<T, Dynamic, Dynamic>>(derived()).solve(I));
}
template <class D> LLT<typename DenseBase<D>::PlainObject> DenseBase<D>::llt() const { return LLT<PlainObject>(derived()); }
template <class D> LDLT<typename DenseBase<D>::PlainObject> DenseBase<D>::ldlt() const { return LDLT<PlainObject>(derived()); }
template <class D> PartialPivLU<typename DenseBase<D>::PlainObject> DenseBase<D>::lu() const { return PartialPivLU<PlainObject>(derived()); }
template <class D> PartialPivLU<typename DenseBase<D>::PlainObject> DenseBase<D>::partialPivLu() const { return PartialPivLU<PlainObject>(derived()); }

// symmetric eigenvalues (cyclic Jacobi) — only OptimizableGraph::verifyInformationMatrices uses it
template <class M>
class SelfAdjointEigenSolver {
  Matrix<typename M::Scalar, Dynamic, 1> ev_;
 public:
  typedef typename M::Scalar S;
  SelfAdjointEigenSolver() {}
  template <class O> SelfAdjointEigenSolver& compute(const DenseBase<O>& a, int = ComputeEigenvectors) {
    Matrix<S, Dynamic, Dynamic> A(a);
    const Index n = A.rows();
    for (int sweep = 0; sweep < 64; sweep++) {
      S off = 0;
      for (Index p = 0; p < n; p++) for (Index q = p + 1; q < n; q++) off += A(p, q) * A(p, q);
      if (off < S(1e-300)) break;
      for (Index p = 0; p < n; p++)
        for (Index q = p + 1; q < n; q++) {
          if (A(p, q) == S(0)) continue;
          const S th = (A(q, q) - A(p, p)) / (2 * A(p, q));
          const S t = (th >= 0 ? S(1) : S(-1)) / (std::abs(th) + std::sqrt(th * th + 1));
          const S c = 1 / std::sqrt(t * t + 1), s = t * c;
          for (Index k = 0; k < n; k++) { const S akp = A(k, p), akq = A(k, q); A(k, p) = c * akp - s * akq; A(k, q) = s * akp + c * akq; }
          for (Index k = 0; k < n; k++) { const S apk = A(p, k), aqk = A(q, k); A(p, k) = c * apk - s * aqk; A(q, k) = s * apk + c * aqk; }
        }
    }
    ev_.resize(n);
    for (Index i = 0; i < n; i++) ev_(i) = A(i, i);
    std::sort(ev_.data(), ev_.data() + n);
    return *this;
  }
  const Matrix<S, Dynamic, 1>& eigenvalues() const { return ev_; }
};

// ---- the usual typedefs
#define ME_TYPEDEFS(T, sfx)                                                                                              \
  typedef Matrix<T, 2, 2> Matrix2##sfx; typedef Matrix<T, 3, 3> Matrix3##sfx; typedef Matrix<T, 4, 4> Matrix4##sfx;       \
  typedef Matrix<T, Dynamic, Dynamic> MatrixX##sfx; typedef Matrix<T, 2, 1> Vector2##sfx; typedef Matrix<T, 3, 1> Vector3##sfx; \
  typedef Matrix<T, 4, 1> Vector4##sfx; typedef Matrix<T, Dynamic, 1> VectorX##sfx; typedef Matrix<T, 1, Dynamic> RowVectorX##sfx; \
  typedef Matrix<T, 1, 3> RowVector3##sfx;
ME_TYPEDEFS(double, d)
ME_TYPEDEFS(float, f)
ME_TYPEDEFS(int, i)
#undef ME_TYPEDEFS

}  // namespace Eigen
