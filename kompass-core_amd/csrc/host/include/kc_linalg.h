// Minimal dense containers with the handful of Eigen type names the
// kompass_cpp class surface uses in its signatures (Eigen is not
// available in the build image nor on the GPU box).  Only storage + element
// access: every numeric kernel of the hot path runs on the device.  The names live in
// namespace Eigen so the reference signatures read the same; a translation unit
// that also includes the real Eigen must not include this header.
#pragma once

#include <array>
#include <cassert>
#include <cmath>
#include <cstddef>
#include <initializer_list>
#include <vector>

namespace Eigen {

using Index = std::ptrdiff_t;

template <typename T, int N>
class FixedVec {
 public:
  FixedVec() { v_.fill(T(0)); }
  FixedVec(std::initializer_list<T> l) {
    v_.fill(T(0));
    int i = 0;
    for (T x : l)
      if (i < N) v_[i++] = x;
  }
  template <typename A, typename B>
  FixedVec(A a, B b) : FixedVec() {
    static_assert(N >= 2, "size");
    v_[0] = static_cast<T>(a);
    v_[1] = static_cast<T>(b);
  }
  template <typename A, typename B, typename C>
  FixedVec(A a, B b, C c) : FixedVec() {
    static_assert(N >= 3, "size");
    v_[0] = static_cast<T>(a);
    v_[1] = static_cast<T>(b);
    v_[2] = static_cast<T>(c);
  }
  template <typename A, typename B, typename C, typename D>
  FixedVec(A a, B b, C c, D d) : FixedVec() {
    static_assert(N >= 4, "size");
    v_[0] = static_cast<T>(a);
    v_[1] = static_cast<T>(b);
    v_[2] = static_cast<T>(c);
    v_[3] = static_cast<T>(d);
  }
  T &operator()(Index i) { return v_[static_cast<size_t>(i)]; }
  const T &operator()(Index i) const { return v_[static_cast<size_t>(i)]; }
  T &operator[](Index i) { return v_[static_cast<size_t>(i)]; }
  const T &operator[](Index i) const { return v_[static_cast<size_t>(i)]; }
  T &x() { return v_[0]; }
  T &y() { return v_[1]; }
  T &z() { return v_[2]; }
  const T &x() const { return v_[0]; }
  const T &y() const { return v_[1]; }
  const T &z() const { return v_[2]; }
  static constexpr Index size() { return N; }
  const T *data() const { return v_.data(); }
  T *data() { return v_.data(); }

 private:
  std::array<T, N> v_;
};

using Vector2f = FixedVec<float, 2>;
using Vector3f = FixedVec<float, 3>;
using Vector4f = FixedVec<float, 4>;
using Vector2i = FixedVec<int, 2>;
using Vector4d = FixedVec<double, 4>;

// coefficient order (x, y, z, w) like Eigen; ctor order (w, x, y, z) like Eigen
class Quaternionf {
 public:
  Quaternionf() : c_{0.f, 0.f, 0.f, 1.f} {}
  Quaternionf(float w, float x, float y, float z) : c_{x, y, z, w} {}
  explicit Quaternionf(const Vector4f &xyzw)
      : c_{xyzw(0), xyzw(1), xyzw(2), xyzw(3)} {}
  float x() const { return c_[0]; }
  float y() const { return c_[1]; }
  float z() const { return c_[2]; }
  float w() const { return c_[3]; }
  const std::array<float, 4> &coeffs() const { return c_; }

 private:
  std::array<float, 4> c_;
};

template <typename T>
class DynVec {
 public:
  DynVec() = default;
  explicit DynVec(Index n) : v_(static_cast<size_t>(n)) {}
  DynVec(const T *p, Index n) : v_(p, p + n) {}
  void resize(Index n) { v_.resize(static_cast<size_t>(n)); }
  Index size() const { return static_cast<Index>(v_.size()); }
  T &operator()(Index i) { return v_[static_cast<size_t>(i)]; }
  const T &operator()(Index i) const { return v_[static_cast<size_t>(i)]; }
  T &operator[](Index i) { return v_[static_cast<size_t>(i)]; }
  const T &operator[](Index i) const { return v_[static_cast<size_t>(i)]; }
  T *data() { return v_.data(); }
  const T *data() const { return v_.data(); }
  void setZero() { std::fill(v_.begin(), v_.end(), T(0)); }

 private:
  std::vector<T> v_;
};
using VectorXf = DynVec<float>;

// row-major (RowMajor = true) or column-major dense matrix
template <typename T, bool RowMajor>
class DynMat {
 public:
  DynMat() = default;
  DynMat(Index r, Index c) : r_(r), c_(c), v_(static_cast<size_t>(r * c)) {}
  void resize(Index r, Index c) {
    r_ = r;
    c_ = c;
    v_.resize(static_cast<size_t>(r * c));
  }
  Index rows() const { return r_; }
  Index cols() const { return c_; }
  Index size() const { return r_ * c_; }
  T &operator()(Index i, Index j) { return v_[idx(i, j)]; }
  const T &operator()(Index i, Index j) const { return v_[idx(i, j)]; }
  T *data() { return v_.data(); }
  const T *data() const { return v_.data(); }
  void fill(T x) { std::fill(v_.begin(), v_.end(), x); }
  // contiguous row access (row-major only)
  T *rowPtr(Index i) {
    static_assert(RowMajor, "row-major only");
    return v_.data() + static_cast<size_t>(i * c_);
  }
  const T *rowPtr(Index i) const {
    static_assert(RowMajor, "row-major only");
    return v_.data() + static_cast<size_t>(i * c_);
  }

 private:
  size_t idx(Index i, Index j) const {
    return static_cast<size_t>(RowMajor ? i * c_ + j : i + j * r_);
  }
  Index r_ = 0, c_ = 0;
  std::vector<T> v_;
};
using MatrixXi = DynMat<int, false>;   // column-major like Eigen's default
using MatrixXf = DynMat<float, false>;

}  // namespace Eigen

namespace Kompass {
namespace Control {
using MatrixXfR = Eigen::DynMat<float, true>;
}
}  // namespace Kompass
