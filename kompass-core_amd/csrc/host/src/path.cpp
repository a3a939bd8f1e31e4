// Path preparation on the host (reference: src/datatypes/path.cpp).  Float /
// double mixing follows the reference statement by statement so that the
// tracked segment handed to the device is bit-identical.
#include "datatypes/path.h"

#include <atomic>
#include <algorithm>
#include <stdexcept>
#include <string>

namespace Path {

namespace {
// Eigen's fixed-size 3-term reduction order
inline float add3(float a, float b, float c) { return a + (b + c); }

// piecewise-linear interpolant over strictly increasing knots (the LINEAR mode
// the follower uses, follower.h:192)
class LinearInterpolant {
 public:
  LinearInterpolant(const std::vector<double> &x, const std::vector<double> &y)
      : x_(x), y_(y), slope_(x.size()) {
    const size_t n = x.size();
    for (size_t i = 0; i + 1 < n; ++i)
      slope_[i] = (y_[i + 1] - y_[i]) / (x_[i + 1] - x_[i]);
    slope_[n - 1] = slope_[n - 2];
  }
  double operator()(double s) const {
    // last knot <= s (first knot when s is in front of the table)
    auto it = std::upper_bound(x_.begin(), x_.end(), s);
    const size_t k = static_cast<size_t>(std::max<int>(int(it - x_.begin()) - 1, 0));
    const double h = s - x_[k];
    const size_t last = x_.size() - 1;
    if (s > x_[last]) return (0.0 * h + slope_[last]) * h + y_[last];
    return ((0.0 * h + 0.0) * h + slope_[k]) * h + y_[k];
  }

 private:
  std::vector<double> x_, y_, slope_;
};
}  // namespace

Path::View::View(const Path &p, size_t start, size_t length)
    : X(p.X_.data() + start), Y(p.Y_.data() + start), Z(p.Z_.data() + start),
      Curvature(p.K_.data() + start), AccumulatedLengths(p.acc_.data() + start),
      start_idx_(start), size_(length),
      acc_available_(p.acc_.size() > start ? p.acc_.size() - start : 0) {}

float Path::View::totalSegmentLength() const {
  float len = 0.0f;
  for (size_t i = 0; i + 1 < size_; ++i)
    len += Path::distance(getIndex(i), getIndex(i + 1));
  return len;
}

Path::Path(const std::vector<Point> &points) {
  if (points.size() < 2)
    throw std::invalid_argument(
        "At least two points are required to create a path.");
  size_ = points.size();
  resize(size_);
  for (size_t i = 0; i < size_; ++i) {
    X_[i] = points[i].x();
    Y_[i] = points[i].y();
    Z_[i] = points[i].z();
    K_[i] = 0.0f;
  }
  touch();
}

Path::Path(const Eigen::VectorXf &x, const Eigen::VectorXf &y,
           const Eigen::VectorXf &z) {
  if (x.size() != y.size() || x.size() != z.size())
    throw std::invalid_argument("X, Y and Z vectors must have the same size.");
  if (x.size() < 2)
    throw std::invalid_argument(
        "At least two points are required to create a path.");
  size_ = static_cast<size_t>(x.size());
  resize(size_);
  for (size_t i = 0; i < size_; ++i) {
    X_[i] = x[(Eigen::Index)i];
    Y_[i] = y[(Eigen::Index)i];
    Z_[i] = z[(Eigen::Index)i];
  }
  touch();
}

void Path::touch() {
  static std::atomic<unsigned long long> next{1};
  serial_ = next.fetch_add(1, std::memory_order_relaxed);
}

void Path::resize(size_t n) {
  X_.resize(n);
  Y_.resize(n);
  Z_.resize(n);
  K_.resize(n);
  interpolated_ = false;
  touch();
}

float Path::distanceSquared(const Point &a, const Point &b) {
  const float dx = a.x() - b.x(), dy = a.y() - b.y(), dz = a.z() - b.z();
  return add3(dx * dx, dy * dy, dz * dz);
}
float Path::distance(const Point &a, const Point &b) {
  return std::sqrt(distanceSquared(a, b));
}
float Path::distanceSquared(const State &s, const Point &p) {
  return distanceSquared(Point(s.x, s.y, 0.0), p);
}

bool Path::endReached(State st, double minDist) {
  const Point e = getEnd();
  const double d = std::sqrt(std::pow(e.x() - st.x, 2) + std::pow(e.y() - st.y, 2));
  return d <= minDist;
}

void Path::checkSegment(size_t s) const {
  if (s >= segments_.size())
    throw std::out_of_range(
        "Invalid segment index. Maximum number of segments is " +
        std::to_string(segments_.size() - 1) +
        ", but requested segment index is " + std::to_string(s));
}
size_t Path::getSegmentStartIndex(size_t s) const {
  checkSegment(s);
  return segments_[s];
}
size_t Path::getSegmentEndIndex(size_t s) const {
  checkSegment(s);
  return s + 1 < segments_.size() ? segments_[s + 1] - 1 : size_ - 1;
}
size_t Path::getSegmentSize(size_t s) const {
  return getSegmentEndIndex(s) - getSegmentStartIndex(s) + 1;
}
Point Path::getSegmentStart(size_t s) const {
  return getIndex(getSegmentStartIndex(s));
}
Point Path::getSegmentEnd(size_t s) const {
  return getIndex(getSegmentEndIndex(s));
}
Path::View Path::getSegment(size_t s) const {
  return getPart(getSegmentStartIndex(s), getSegmentEndIndex(s));
}
Path::View Path::getPart(size_t start, size_t end) const {
  if (start >= size_ || end >= size_ || start > end)
    throw std::out_of_range(
        "Invalid range for path part. Maximum path size is " +
        std::to_string(size_) + ", but requested part start= " +
        std::to_string(start) + ", and requested end= " + std::to_string(end));
  return View(*this, start, end - start + 1);
}

void Path::pushPoint(const Point &p) {
  X_.resize(size_ + 1);
  Y_.resize(size_ + 1);
  Z_.resize(size_ + 1);
  K_.resize(size_ + 1);
  interpolated_ = false;
  X_[size_] = p.x();
  Y_[size_] = p.y();
  Z_[size_] = p.z();
  ++size_;
  touch();
}

static float heading(const Point &a, const Point &b) {
  const float dx = b.x() - a.x(), dy = b.y() - a.y();
  return std::atan2(dy, dx);  // float overload, as in the reference
}
float Path::getEndOrientation() const {
  return heading(getIndex(size_ - 2), getIndex(size_ - 1));
}
float Path::getStartOrientation() const { return heading(getIndex(0), getIndex(1)); }
float Path::getOrientation(size_t i) const {
  if (i + 1 < size_) return heading(getIndex(i), getIndex(i + 1));
  return heading(getIndex(size_ - 2), getIndex(size_ - 1));
}

float Path::totalPathLength() const {
  if (size_ < 2) return 0.0f;
  if (interpolated_) return total_length_;
  float total = 0.0f;
  for (size_t i = 1; i < size_; ++i) total += distance(getIndex(i - 1), getIndex(i));
  return total;
}

void Path::interpolate(double max_dist, InterpolationType type) {
  if (size_ < 2)
    throw std::invalid_argument(
        "At least two points are required to perform interpolation.");
  if (type != InterpolationType::LINEAR)
    throw std::invalid_argument(
        "only PathInterpolationType.LINEAR is available in this build (the "
        "reference's cubic modes come from a vendored GPL spline header)");
  // chord-length parametrisation: float running length, double knots
  std::vector<double> s(size_), xs(size_), ys(size_);
  s[0] = 0.0;
  xs[0] = X_[0];
  ys[0] = Y_[0];
  total_length_ = 0.0f;
  for (size_t i = 1; i < size_; ++i) {
    const double seg = std::hypot(X_[i] - X_[i - 1], Y_[i] - Y_[i - 1]);  // float overload
    total_length_ = static_cast<float>(static_cast<double>(total_length_) + seg);
    s[i] = total_length_;
    xs[i] = X_[i];
    ys[i] = Y_[i];
  }
  const LinearInterpolant fx(s, xs), fy(s, ys);
  const size_t n_new =
      static_cast<size_t>(static_cast<double>(total_length_) / max_dist) + 1;
  X_.assign(n_new, 0.0f);
  Y_.assign(n_new, 0.0f);
  Z_.assign(n_new, 0.0f);
  K_.assign(n_new, 0.0f);
  acc_.resize(n_new);  // keeps earlier entries, zero-fills new ones
  size_t k = 0;
  const double total = total_length_;
  for (double t = 0.0; t <= total && k < n_new; t += max_dist) {
    acc_[k] = static_cast<float>(t);
    X_[k] = static_cast<float>(fx(t));
    Y_[k] = static_cast<float>(fy(t));
    ++k;
  }
  if (k < n_new && k > 0) {  // closing point; its prefix entry stays untouched
    X_[k] = static_cast<float>(fx(total));
    Y_[k] = static_cast<float>(fy(total));
    ++k;
  }
  interpolated_ = true;
  touch();
  size_ = k;
  // discrete curvature from successive chords
  if (size_ >= 2) {
    float dx0 = X_[1] - X_[0], dy0 = Y_[1] - Y_[0];
    for (size_t i = 1; i + 1 < size_; ++i) {
      const float dx = X_[i + 1] - X_[i], dy = Y_[i + 1] - Y_[i];
      const float ddx = dx - dx0, ddy = dy - dy0;
      const float v = dx * dx + dy * dy;
      const float den = v * std::sqrt(v);
      K_[i] = den > 1e-6f ? (dx0 * ddy - ddx * dy0) / den : 0.0f;
      dx0 = dx;
      dy0 = dy;
    }
  }
}

void Path::segment(double seg_len, size_t max_pts) {
  if (size_ < 2) return;
  segments_.assign(1, 0);
  if (!interpolated_) {  // per-edge lengths (reference quirk Q6)
    acc_.assign(size_ - 1, 0.0f);
    for (size_t i = 0; i + 1 < size_; ++i)
      acc_[i] = distance(getIndex(i), getIndex(i + 1));
  }
  size_t first = 0;
  float first_len = acc_[0];
  for (size_t i = 1; i < size_; ++i) {
    const float at = i < acc_.size() ? acc_[i] : 0.0f;
    const bool too_long = seg_len > 0.0 && static_cast<double>(at - first_len) >= seg_len;
    const bool too_many = max_pts > 0 && (i - first + 1) > max_pts;
    if (too_long || too_many) {
      segments_.push_back(i);
      first = i;
      first_len = at;
    }
  }
  touch();
}

}  // namespace Path
