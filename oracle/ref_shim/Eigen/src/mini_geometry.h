// mini_geometry.h — look-alike of Eigen/Geometry for the reference's g2o types (TEST INFRASTRUCTURE, see mini_eigen.h).
// Quaternion <-> rotation-matrix conversions, the quaternion product and the rotation of a vector follow the formulas of
// Eigen 3.3 (Quaternion.h: quaternionbase_assign_impl<Other,3,3>, toRotationMatrix, quat_product, _transformVector).
#pragma once
#include "mini_eigen.h"

namespace Eigen {

template <class S, int Dim, int Mode, int Opt = 0> class Transform;

template <class S, int Opt = 0>
class Quaternion {
  Matrix<S, 4, 1> c_;   // x y z w
 public:
  typedef S Scalar;
  typedef Matrix<S, 3, 1> Vector3;
  typedef Matrix<S, 3, 3> Matrix3;
  Quaternion() {}
  Quaternion(S w, S x, S y, S z) { c_(0) = x; c_(1) = y; c_(2) = z; c_(3) = w; }
  explicit Quaternion(const S* d) { for (int i = 0; i < 4; i++) c_(i) = d[i]; }
  Quaternion(const Quaternion&) = default;
  Quaternion& operator=(const Quaternion&) = default;
  template <class O> explicit Quaternion(const DenseBase<O>& m) { *this = m; }
  // from a rotation matrix (the branchy trace method of Eigen) or from a 4-vector of coefficients x y z w
  template <class O> Quaternion& operator=(const DenseBase<O>& mat) {
    if (mat.rows() == 4 && mat.cols() == 1) { for (int i = 0; i < 4; i++) c_(i) = mat(i); return *this; }
    S t = mat.trace();
    if (t > S(0)) {
      t = std::sqrt(t + S(1.0));
      w() = S(0.5) * t;
      t = S(0.5) / t;
      x() = (mat(2, 1) - mat(1, 2)) * t;
      y() = (mat(0, 2) - mat(2, 0)) * t;
      z() = (mat(1, 0) - mat(0, 1)) * t;
    } else {
      Index i = 0;
      if (mat(1, 1) > mat(0, 0)) i = 1;
      if (mat(2, 2) > mat(i, i)) i = 2;
      const Index j = (i + 1) % 3, k = (j + 1) % 3;
      t = std::sqrt(mat(i, i) - mat(j, j) - mat(k, k) + S(1.0));
      c_(i) = S(0.5) * t;
      t = S(0.5) / t;
      w() = (mat(k, j) - mat(j, k)) * t;
      c_(j) = (mat(j, i) + mat(i, j)) * t;
      c_(k) = (mat(k, i) + mat(i, k)) * t;
    }
    return *this;
  }
  S x() const { return c_(0); } S y() const { return c_(1); } S z() const { return c_(2); } S w() const { return c_(3); }
  S& x() { return c_(0); } S& y() { return c_(1); } S& z() { return c_(2); } S& w() { return c_(3); }
  const Matrix<S, 4, 1>& coeffs() const { return c_; }
  Matrix<S, 4, 1>& coeffs() { return c_; }
  View<const S, 3, 1> vec() const { return View<const S, 3, 1>(c_.data(), 3, 1, 1, 0); }
  View<S, 3, 1> vec() { return View<S, 3, 1>(c_.data(), 3, 1, 1, 0); }
  Quaternion& setIdentity() { c_(0) = c_(1) = c_(2) = 0; c_(3) = 1; return *this; }
  static Quaternion Identity() { return Quaternion(1, 0, 0, 0); }
  S squaredNorm() const { return c_.squaredNorm(); }
  S norm() const { return c_.norm(); }
  void normalize() { c_.normalize(); }
  Quaternion normalized() const { Quaternion q(*this); q.normalize(); return q; }
  Quaternion conjugate() const { return Quaternion(w(), -x(), -y(), -z()); }
  Quaternion inverse() const {
    const S n2 = squaredNorm();
    if (n2 > S(0)) { Quaternion q = conjugate(); q.c_ /= n2; return q; }
    Quaternion q; q.c_.setZero(); return q;
  }
  S dot(const Quaternion& o) const { return c_.dot(o.c_); }
  Quaternion operator*(const Quaternion& b) const {
    const Quaternion& a = *this;
    return Quaternion(a.w() * b.w() - a.x() * b.x() - a.y() * b.y() - a.z() * b.z(),
                      a.w() * b.x() + a.x() * b.w() + a.y() * b.z() - a.z() * b.y(),
                      a.w() * b.y() + a.y() * b.w() + a.z() * b.x() - a.x() * b.z(),
                      a.w() * b.z() + a.z() * b.w() + a.x() * b.y() - a.y() * b.x());
  }
  Quaternion& operator*=(const Quaternion& b) { *this = *this * b; return *this; }
  // rotation of a vector: v + w * uv + q.vec x uv with uv = 2 q.vec x v  (QuaternionBase::_transformVector)
  template <class O> Vector3 operator*(const DenseBase<O>& v) const {
    Vector3 qv(x(), y(), z());
    Vector3 uv = qv.cross(v);
    uv += uv;
    return Vector3(v) + w() * uv + qv.cross(uv);
  }
  Matrix3 toRotationMatrix() const {
    Matrix3 res;
    const S tx = S(2) * x(), ty = S(2) * y(), tz = S(2) * z();
    const S twx = tx * w(), twy = ty * w(), twz = tz * w();
    const S txx = tx * x(), txy = ty * x(), txz = tz * x();
    const S tyy = ty * y(), tyz = tz * y(), tzz = tz * z();
    res(0, 0) = S(1) - (tyy + tzz); res(0, 1) = txy - twz; res(0, 2) = txz + twy;
    res(1, 0) = txy + twz; res(1, 1) = S(1) - (txx + tzz); res(1, 2) = tyz - twx;
    res(2, 0) = txz - twy; res(2, 1) = tyz + twx; res(2, 2) = S(1) - (txx + tyy);
    return res;
  }
  Matrix3 matrix() const { return toRotationMatrix(); }
  explicit operator Transform<S, 3, Isometry>() const;
};
typedef Quaternion<double> Quaterniond;
typedef Quaternion<float> Quaternionf;

template <class S>
class AngleAxis {
  Matrix<S, 3, 1> axis_; S angle_ = 0;
 public:
  AngleAxis() {}
  template <class O> AngleAxis(S angle, const DenseBase<O>& axis) : axis_(axis), angle_(angle) {}
  S angle() const { return angle_; }
  const Matrix<S, 3, 1>& axis() const { return axis_; }
  Matrix<S, 3, 3> toRotationMatrix() const {
    Matrix<S, 3, 3> res;
    const S s = std::sin(angle_), c = std::cos(angle_);
    Matrix<S, 3, 1> sin_axis = s * axis_, cos1_axis = (S(1) - c) * axis_;
    S tmp;
    tmp = cos1_axis.x() * axis_.y(); res(0, 1) = tmp - sin_axis.z(); res(1, 0) = tmp + sin_axis.z();
    tmp = cos1_axis.x() * axis_.z(); res(0, 2) = tmp + sin_axis.y(); res(2, 0) = tmp - sin_axis.y();
    tmp = cos1_axis.y() * axis_.z(); res(1, 2) = tmp - sin_axis.x(); res(2, 1) = tmp + sin_axis.x();
    res(0, 0) = cos1_axis.x() * axis_.x() + c; res(1, 1) = cos1_axis.y() * axis_.y() + c; res(2, 2) = cos1_axis.z() * axis_.z() + c;
    return res;
  }
};
typedef AngleAxis<double> AngleAxisd;

// just enough of Transform for SE3Quat::operator Isometry3d and the typedefs in g2o's eigen_types.h
template <class S, int Dim, int Mode, int Opt>
class Transform {
  Matrix<S, Dim + 1, Dim + 1> m_;
 public:
  Transform() { m_.setIdentity(); }
  template <class O> explicit Transform(const DenseBase<O>& m) { m_ = m; }
  static Transform Identity() { return Transform(); }
  const Matrix<S, Dim + 1, Dim + 1>& matrix() const { return m_; }
  Matrix<S, Dim + 1, Dim + 1>& matrix() { return m_; }
  View<S, Dim, Dim> linear() { return m_.template block<Dim, Dim>(0, 0); }
  View<const S, Dim, Dim> linear() const { return m_.template block<Dim, Dim>(0, 0); }
  View<S, Dim, Dim> rotation() { return linear(); }
  View<const S, Dim, Dim> rotation() const { return linear(); }
  View<S, Dim, 1> translation() { return View<S, Dim, 1>(m_.data() + Dim * (Dim + 1), Dim, 1, 1, 0); }
  View<const S, Dim, 1> translation() const { return View<const S, Dim, 1>(m_.data() + Dim * (Dim + 1), Dim, 1, 1, 0); }
  Transform operator*(const Transform& o) const { return Transform(m_ * o.m_); }
  template <class O> Matrix<S, Dim, 1> operator*(const DenseBase<O>& v) const { return Matrix<S, Dim, 1>(linear() * v + translation()); }
  Transform inverse() const { return Transform(m_.inverse()); }
};
template <class S, int Opt> Quaternion<S, Opt>::operator Transform<S, 3, Isometry>() const {
  Transform<S, 3, Isometry> t;
  t.linear() = toRotationMatrix();
  return t;
}
typedef Transform<double, 3, Isometry> Isometry3d;
typedef Transform<double, 2, Isometry> Isometry2d;
typedef Transform<double, 3, Affine> Affine3d;
typedef Transform<double, 2, Affine> Affine2d;

}  // namespace Eigen
