// CostEvaluator + DWA host side (reference: src/utils/cost_evaluator.cpp,
// src/controllers/dwa.cpp).  The five built-in costs and the argmin are device
// work; this file only moves arguments across the C ABI and keeps the
// reference's host glue (adaptive horizon, tracked segment).
#include "utils/cost_evaluator.h"

#include <cstdlib>
#include <cmath>
#include <cstring>

#include "controllers/dwa.h"

namespace Kompass {
namespace Control {

namespace {
kc_weights toKc(const CostEvaluator::TrajectoryCostsWeights &w) {
  kc_weights k;
  k.reference_path_distance_weight = w.getParameter<double>("reference_path_distance_weight");
  k.goal_distance_weight = w.getParameter<double>("goal_distance_weight");
  k.obstacles_distance_weight = w.getParameter<double>("obstacles_distance_weight");
  k.smoothness_weight = w.getParameter<double>("smoothness_weight");
  k.jerk_weight = w.getParameter<double>("jerk_weight");
  return k;
}
hip::DwaHandle makeEvaluatorContext(const Eigen::Vector3f &spos, const Eigen::Quaternionf &srot,
                                    const ControlLimitsParams &lim, size_t maxN, size_t P,
                                    size_t maxSeg) {
  kc_dwa_params p;
  std::memset(&p, 0, sizeof(p));
  p.shape = KC_CYLINDER;  // unused by the evaluator-only context
  p.dims[0] = 0.1f;
  p.dims[1] = 0.1f;
  p.ndims = 2;
  for (int i = 0; i < 3; ++i) p.sensor_pos[i] = spos(i);
  p.sensor_rot_xyzw[0] = srot.x();
  p.sensor_rot_xyzw[1] = srot.y();
  p.sensor_rot_xyzw[2] = srot.z();
  p.sensor_rot_xyzw[3] = srot.w();
  p.octree_res = 0.1;
  p.time_step = 0.1;
  p.max_samples = std::max<size_t>(maxN, 1);
  p.max_points = std::max<size_t>(P, 2);
  p.max_segment = std::max<size_t>(maxSeg, 16);
  p.max_obstacles = 1024;
  p.acc_limits[0] = static_cast<float>(lim.velXParams.maxAcceleration);
  p.acc_limits[1] = static_cast<float>(lim.velYParams.maxAcceleration);
  p.acc_limits[2] = static_cast<float>(lim.omegaParams.maxAcceleration);
  p.device = 0;
  return hip::makeDwa(p);
}
}  // namespace

CostEvaluator::CostEvaluator(TrajectoryCostsWeights &w, ControlLimitsParams lim, size_t maxN,
                             size_t P, size_t maxSeg)
    : CostEvaluator(w, Eigen::Vector3f(0.f, 0.f, 0.f), Eigen::Quaternionf(), lim, maxN, P, maxSeg) {}

CostEvaluator::CostEvaluator(TrajectoryCostsWeights &w, const Eigen::Vector3f &spos,
                             const Eigen::Quaternionf &srot, ControlLimitsParams lim, size_t maxN,
                             size_t P, size_t maxSeg) {
  accLimits_ = {static_cast<float>(lim.velXParams.maxAcceleration),
                static_cast<float>(lim.velYParams.maxAcceleration),
                static_cast<float>(lim.omegaParams.maxAcceleration)};
  if (const char *e = std::getenv("KOMPASS_RESIDENT_PATH")) residentPath_ = e[0] == '1';
  ctx_ = makeEvaluatorContext(spos, srot, lim, maxN, P, maxSeg);
  updateCostWeights(w);
}

CostEvaluator::CostEvaluator(TrajectoryCostsWeights &w, hip::DwaHandle ctx) : ctx_(std::move(ctx)) {
  accLimits_ = {0.f, 0.f, 0.f};
  if (const char *e = std::getenv("KOMPASS_RESIDENT_PATH")) residentPath_ = e[0] == '1';
  updateCostWeights(w);
}

CostEvaluator::~CostEvaluator() { customTrajCostsPtrs_.clear(); }

void CostEvaluator::updateCostWeights(TrajectoryCostsWeights &w) {
  costWeights = std::make_unique<TrajectoryCostsWeights>(w);
  const kc_weights k = toKc(w);
  hip::check(kc_dwa_set_weights(ctx_.get(), &k));
}

void CostEvaluator::setPointScan(const LaserScan &scan, const Path::State &s, const float range,
                                 const float multiple) {
  if (sensorDataResident) return;  // the sampler uploaded this cycle's data
  const kc_state st{s.x, s.y, s.yaw, s.speed};
  // maxObstaclesDist = range / multiple; the ABI divides by the default 3
  const float eff = range / multiple * 3.0f;
  hip::check(kc_dwa_set_scan(ctx_.get(), &st, scan.ranges.data(), scan.angles.data(),
                             std::min(scan.ranges.size(), scan.angles.size()), eff));
}
void CostEvaluator::setPointScan(const std::vector<Path::Point> &cloud, const Path::State &s,
                                 const float range, const float multiple) {
  if (sensorDataResident) return;
  const kc_state st{s.x, s.y, s.yaw, s.speed};
  // Path::Point is three packed floats: the list goes to the device as it lies
  static_assert(sizeof(Path::Point) == 3 * sizeof(float), "Path::Point must be packed (x, y, z)");
  const float eff = range / multiple * 3.0f;
  hip::check(kc_dwa_set_points(ctx_.get(), &st, cloud.empty() ? nullptr : cloud.data()->data(), cloud.size(), eff));
}

void CostEvaluator::uploadSegment(const Path::Path *ref, const Path::Path::View &seg) {
  const size_t S = seg.getSize();
  // a view into `ref` (what the controllers pass): the path stays resident on
  // the device, only the window moves (kc_dwa_set_path once per path content)
  // Opt-in (useResidentPath / KOMPASS_RESIDENT_PATH=1): it takes 10 us of segment
  // handling off the host per cycle, but the table kernel then sits in the stream
  // in front of the cycle (+7 us), while the host-built tables are ready before
  // the kernel needs them.
  const bool resident = residentPath_;
  const size_t start = seg.getStartIndex();
  if (resident && S > 0 && start + S <= ref->getSize() && seg.getXPointer() == ref->xData() + start &&
      seg.getYPointer() == ref->yData() + start && seg.getZPointer() == ref->zData() + start) {
    if (residentSerial_ != ref->serial()) {
      const size_t N = ref->getSize();
      std::vector<float> acc(N);
      for (size_t j = 0; j < N; ++j) acc[j] = ref->getDistanceAtIndex(j);
      hip::check(kc_dwa_set_path(ctx_.get(), ref->xData(), ref->yData(), ref->zData(), acc.data(), N,
                                 ref->totalPathLength()));
      residentSerial_ = ref->serial();
    }
    hip::check(kc_dwa_set_tracked_window(ctx_.get(), start, S));
    return;
  }
  std::vector<float> acc(S);
  for (size_t j = 0; j < S; ++j)  // Path::getDistanceAtIndex(closest_abs_idx)
    acc[j] = ref->getDistanceAtIndex(seg.getStartIndex() + j);
  hip::check(kc_dwa_set_tracked_segment(ctx_.get(), seg.getXPointer(), seg.getYPointer(),
                                        seg.getZPointer(), acc.data(), S, ref->totalPathLength()));
}

// custom costs are host callbacks: fetch the per-sample device totals, add the
// callbacks in registration order with the reference's float += double*float
// rounding, and take the (first) minimum on the host -- the structure of the
// reference's own GPU build (cost_evaluator_gpu.cpp:344-383)
TrajSearchResult CostEvaluator::finishWithCustomCosts(const Path::Path *ref, size_t P, int64_t *raw_out) {
  size_t rows = 0;
  hip::check(kc_dwa_get_samples(ctx_.get(), nullptr, nullptr, nullptr, nullptr, 0, &rows));
  TrajSearchResult out;
  if (rows == 0) return out;
  std::vector<float> px(rows * P), py(rows * P), costs(rows);
  std::vector<int32_t> raw(rows);
  hip::check(kc_dwa_get_samples(ctx_.get(), px.data(), py.data(), raw.data(), costs.data(), rows, &rows));
  float best = DEFAULT_MIN_DIST;
  size_t arg = 0;
  bool found = false;
  Trajectory2D traj(P);
  for (size_t r = 0; r < rows; ++r) {
    for (size_t i = 0; i < P; ++i) traj.path.add(i, px[r * P + i], py[r * P + i], 0.0f);
    float total = costs[r];
    for (const auto &c : customTrajCostsPtrs_)
      total = static_cast<float>(static_cast<double>(total) +
                                 c->weight * static_cast<double>(c->evaluator_(traj, *ref)));
    if (total < best) {
      best = total;
      arg = r;
      found = true;
    }
  }
  if (!found) return out;
  out.isTrajFound = true;
  out.trajCost = best;
  if (raw_out) *raw_out = raw[arg];
  out.trajectory = Trajectory2D(P);
  for (size_t i = 0; i < P; ++i) out.trajectory.path.add(i, px[arg * P + i], py[arg * P + i], 0.0f);
  double vx = 0.0, vy = 0.0, om = 0.0;
  hip::check(kc_dwa_get_sample_velocity(ctx_.get(), raw[arg], &vx, &vy, &om));
  for (size_t i = 0; i + 1 < P; ++i) out.trajectory.velocities.add(i, Velocity2D(vx, vy, om));
  return out;
}

TrajSearchResult
CostEvaluator::getMinTrajectoryCost(const std::unique_ptr<TrajectorySamples2D> &trajs,
                                    const Path::Path *ref, const Path::Path::View &seg) {
  const size_t N = trajs->size(), P = trajs->numPointsPerTrajectory_;
  TrajSearchResult out;
  out.trajectory = Trajectory2D(P);
  if (N == 0) return out;
  uploadSegment(ref, seg);
  kc_result r;
  std::vector<float> costs(N);
  hip::check(kc_cost_evaluate(ctx_.get(), trajs->paths.x.data(), trajs->paths.y.data(),
                              trajs->velocities.vx.data(), trajs->velocities.vy.data(),
                              trajs->velocities.omega.data(), N, P, costs.data(), &r));
  if (!customTrajCostsPtrs_.empty()) {
    float best = DEFAULT_MIN_DIST;
    Eigen::Index arg = -1;
    for (size_t n = 0; n < N; ++n) {
      const Trajectory2D t = trajs->getIndex((Eigen::Index)n);
      float total = costs[n];
      for (const auto &c : customTrajCostsPtrs_)
        total = static_cast<float>(static_cast<double>(total) +
                                   c->weight * static_cast<double>(c->evaluator_(t, *ref)));
      if (total < best) {
        best = total;
        arg = (Eigen::Index)n;
      }
    }
    if (arg >= 0) {
      out.isTrajFound = true;
      out.trajCost = best;
      out.trajectory = trajs->getIndex(arg);
    }
    return out;
  }
  if (r.found) {
    out.isTrajFound = true;
    out.trajCost = r.cost;
    out.trajectory = trajs->getIndex((Eigen::Index)r.index);
  }
  return out;
}

TrajSearchResult CostEvaluator::cycleOnDevice(const Path::Path *ref, const Path::Path::View &seg, size_t P,
                                              const Path::State &pose, double time_step,
                                              const std::function<Velocity2D(size_t)> &sampleVelocity,
                                              size_t n_generated, kc_comm *comm) {
  uploadSegment(ref, seg);
  const kc_state st{pose.x, pose.y, pose.yaw, pose.speed};
  // (sharded: the context already holds this rank's share of the lattice -- DWA::enableSharding set the
  // rule, kc_dwa_sample_window applied it)
  (void)n_generated;
  kc_result r;
  TrajSearchResult mine_best;   // sharded + custom costs: this rank's own best, with its path
  int64_t mine_raw = -1;
  int rc;
  if (comm && !customTrajCostsPtrs_.empty()) {
    // SURVEY 8e row 2 / cost_evaluator.cpp:96-100: every rank scores its share on the device, adds the callbacks
    // to the totals of its own admissible rows on the host (registration order, the reference's rounding) and
    // hands its best into the same single exchange a plain sharded cycle runs.  Whatever fails here, the rank
    // still takes part in the exchange (status != 0): the failure is every rank's, of the same cycle.
    int status = kc_dwa_cycle(ctx_.get(), &st, P, &r);
    std::string why = status != KC_OK ? kc_last_error() : "";
    if (status == KC_OK) {
      try {
        mine_best = finishWithCustomCosts(ref, P, &mine_raw);
      } catch (const std::exception &e) {
        status = KC_ERR_STATE;
        why = e.what();
      }
    }
    rc = kc_dwa_exchange_best(ctx_.get(), comm, status, mine_best.isTrajFound ? 1 : 0, mine_best.trajCost, mine_raw, &r);
    if (status != KC_OK) throw std::runtime_error("sharded cycle with custom costs: " + why);
  } else {
    rc = comm ? kc_dwa_cycle_sharded(ctx_.get(), comm, &st, P, &r) : kc_dwa_cycle(ctx_.get(), &st, P, &r);
  }
  hip::check(rc);
  sensorDataResident = false;
  if (!customTrajCostsPtrs_.empty() && !comm) return finishWithCustomCosts(ref, P);
  TrajSearchResult out;
  out.trajectory = Trajectory2D(P);
  if (!r.found) return out;
  out.isTrajFound = true;
  out.trajCost = r.cost;
  int mine = 1;
  if (comm) hip::check(kc_dwa_owns_sample(ctx_.get(), r.raw_index, &mine));
  if (mine && comm && !customTrajCostsPtrs_.empty() && mine_raw == r.raw_index) return mine_best;
  if (mine)
    hip::check(kc_dwa_get_best(ctx_.get(), out.trajectory.path.x.data(), out.trajectory.path.y.data(),
                               out.trajectory.velocities.vx.data(), out.trajectory.velocities.vy.data(),
                               out.trajectory.velocities.omega.data()));
  if (!mine) {
    // the winner lives on another rank: its velocity is known here (the lattice is replicated),
    // its path is Path::State::update (path.h:24-30) from that velocity -- the device's arithmetic
    const Velocity2D v = sampleVelocity(static_cast<size_t>(r.raw_index));
    const double dt = static_cast<double>(static_cast<float>(time_step));
    double x = pose.x, y = pose.y, yaw = pose.yaw;
    out.trajectory.path.add(0, static_cast<float>(x), static_cast<float>(y), 0.0f);
    for (size_t i = 0; i + 1 < P; ++i) {
      const double c = std::cos(yaw), sn = std::sin(yaw);
      const double ix = (v.vx() * c - v.vy() * sn) * dt, iy = (v.vx() * sn + v.vy() * c) * dt;
      x += ix;
      y += iy;
      yaw += v.omega() * dt;
      out.trajectory.path.add(i + 1, static_cast<float>(x), static_cast<float>(y), 0.0f);
      out.trajectory.velocities.add(i, v);
    }
  }
  out.trajectory.path.z.setZero();
  return out;
}

TrajSearchResult CostEvaluator::getMinTrajectoryCostOnDevice(const Path::Path *ref,
                                                             const Path::Path::View &seg, size_t P) {
  uploadSegment(ref, seg);
  hip::check(kc_dwa_evaluate(ctx_.get()));
  kc_result r;
  hip::check(kc_dwa_fetch_result(ctx_.get(), &r));
  sensorDataResident = false;
  if (!customTrajCostsPtrs_.empty()) return finishWithCustomCosts(ref, P);
  TrajSearchResult out;
  out.trajectory = Trajectory2D(P);
  if (!r.found) return out;
  out.isTrajFound = true;
  out.trajCost = r.cost;
  hip::check(kc_dwa_get_best(ctx_.get(), out.trajectory.path.x.data(), out.trajectory.path.y.data(),
                             out.trajectory.velocities.vx.data(),
                             out.trajectory.velocities.vy.data(),
                             out.trajectory.velocities.omega.data()));
  out.trajectory.path.z.setZero();
  return out;
}

// ===========================================================================
// DWA
// ===========================================================================
DWA::DWA(ControlLimitsParams lim, ControlType type, double dt, double predictionHorizon,
         double controlHorizon, int maxLin, int maxAng, const CollisionChecker::ShapeType shape,
         const std::vector<float> dims, const Eigen::Vector3f &spos, const Eigen::Vector4f &srot,
         const double octreeRes, CostEvaluator::TrajectoryCostsWeights w, const int host_threads_)
    : Follower() {
  configure(lim, type, dt, predictionHorizon, controlHorizon, maxLin, maxAng, shape, dims, spos,
            srot, octreeRes, w, host_threads_);
  max_forward_distance_ =
      (type == ControlType::OMNI ? std::max(lim.velXParams.maxVel, lim.velYParams.maxVel)
                                 : lim.velXParams.maxVel) *
      predictionHorizon;
  initJitCompile();
}

DWA::DWA(TrajectorySampler::TrajectorySamplerParameters config, ControlLimitsParams lim,
         ControlType type, const CollisionChecker::ShapeType shape, const std::vector<float> dims,
         const Eigen::Vector3f &spos, const Eigen::Vector4f &srot,
         CostEvaluator::TrajectoryCostsWeights w, const int host_threads_)
    : Follower() {
  configure(config, lim, type, shape, dims, spos, srot, w, host_threads_);
  const double horizon = config.getParameter<double>("control_horizon");
  max_forward_distance_ =
      (type == ControlType::OMNI ? std::max(lim.velXParams.maxVel, lim.velYParams.maxVel)
                                 : lim.velXParams.maxVel) *
      horizon;
  initJitCompile();
}

// the HIP code objects are built ahead of time; the reference needs this call
// to force its SYCL JIT (dwa.cpp:75-91).  Kept as a cheap device round trip so
// the first planner tick does not pay first-use initialisation.
void DWA::initJitCompile() {
  std::vector<double> x{0.0}, y{0.0}, yaw{0.0};
  uint8_t hit = 0;
  hip::check(kc_dwa_check_poses(trajSampler->context().get(), x.data(), y.data(), yaw.data(), 1, &hit));
}

void DWA::configure(ControlLimitsParams lim, ControlType type, double dt, double predictionHorizon,
                    double controlHorizon, int maxLin, int maxAng,
                    const CollisionChecker::ShapeType shape, const std::vector<float> dims,
                    const Eigen::Vector3f &spos, const Eigen::Vector4f &srot, const double octreeRes,
                    CostEvaluator::TrajectoryCostsWeights w, const int host_threads_) {
  trajSampler = std::make_unique<TrajectorySampler>(lim, type, dt, predictionHorizon, controlHorizon,
                                                    maxLin, maxAng, shape, dims, spos,
                                                    Eigen::Quaternionf(srot), octreeRes, host_threads_);
  // sampler and evaluator share one device context: the rolled-out samples
  // never leave HBM between the two stages
  trajCostEvaluator = std::make_unique<CostEvaluator>(w, trajSampler->context());
  this->host_threads_ = host_threads_;
}

void DWA::configure(TrajectorySampler::TrajectorySamplerParameters config, ControlLimitsParams lim,
                    ControlType type, const CollisionChecker::ShapeType shape,
                    const std::vector<float> dims, const Eigen::Vector3f &spos,
                    const Eigen::Vector4f &srot, CostEvaluator::TrajectoryCostsWeights w,
                    const int host_threads_) {
  trajSampler = std::make_unique<TrajectorySampler>(config, lim, type, shape, dims, spos,
                                                    Eigen::Quaternionf(srot), host_threads_);
  trajCostEvaluator = std::make_unique<CostEvaluator>(w, trajSampler->context());
  this->host_threads_ = host_threads_;
}

void DWA::resetOctreeResolution(const double r) { trajSampler->resetOctreeResolution(r); }
void DWA::setSensorMaxRange(const float r) { maxLocalRange_ = r; }
void DWA::addCustomCost(double weight, CostEvaluator::CustomCostFunction f) {
  trajCostEvaluator->addCustomCost(weight, std::move(f));
}
void DWA::setCurrentState(const Path::State &s) {
  pose_ = s;
  trajSampler->updateState(s);
}

void DWA::adaptPredictionHorizonToCurvature() {
  const double base = trajSampler->getBasePredictionHorizon();
  const double v_max = limits_.velXParams.maxVel;  // see Controller (Q1)
  if (!on_.path || v_max < 1e-3 || knob_.point_spacing <= 0.0) {
    trajSampler->setPredictionHorizon(base);
    max_forward_distance_ = base * v_max;
    return;
  }
  const size_t last = on_.path->getSize() - 1;
  const size_t first = std::min(on_.nearest->index, last);
  const size_t peek = static_cast<size_t>(std::ceil(base * v_max / knob_.point_spacing));
  const size_t end = std::min(first + peek, last);
  float kappa = 0.0f;
  for (size_t i = first; i <= end; ++i)
    kappa = std::max(kappa, std::abs(static_cast<float>(on_.path->getCurvature(i))));
  double horizon = base;
  if (kappa > knob_.horizon_tolerance) {
    const double cap = std::sqrt(8.0 * knob_.horizon_tolerance / kappa) / v_max;
    horizon = std::min(base, cap);
    LOG_DEBUG("Using Adaptive Horizon: ", horizon);
  }
  trajSampler->setPredictionHorizon(horizon);
  max_forward_distance_ = horizon * v_max;
}

Path::Path::View DWA::findTrackedPathSegment() {
  const size_t last = on_.path->getSize() - 1;
  const size_t first = std::min(on_.nearest->index, last);
  size_t look = on_.longest_segment;
  if (knob_.point_spacing > 0.0)
    look = std::max(on_.longest_segment,
                    static_cast<size_t>(std::ceil(max_forward_distance_ /
                                                  knob_.point_spacing)) + 1);
  return on_.path->getPart(first, std::min(first + look, last));
}

std::tuple<MatrixXfR, MatrixXfR> DWA::getDebuggingSamples() const {
  if (!debuggingSamples_) throw std::invalid_argument("No debugging samples are available");
  const size_t n = debuggingSamples_->paths.size(), P = debuggingSamples_->numPointsPerTrajectory_;
  MatrixXfR x((Eigen::Index)n, (Eigen::Index)P), y((Eigen::Index)n, (Eigen::Index)P);
  if (n) {
    std::memcpy(x.data(), debuggingSamples_->paths.x.data(), n * P * sizeof(float));
    std::memcpy(y.data(), debuggingSamples_->paths.y.data(), n * P * sizeof(float));
  }
  return std::make_tuple(x, y);
}
Control::TrajectorySamples2D DWA::getDebuggingSamplesPure() const {
  if (!debuggingSamples_) throw std::invalid_argument("No debugging samples are available");
  return *debuggingSamples_;
}

}  // namespace Control
}  // namespace Kompass
