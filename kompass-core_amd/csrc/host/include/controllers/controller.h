// Controller base of the kompass_cpp surface (reference: controllers/
// controller.{h,cpp}).  Host-side state holder.
#pragma once

#include <algorithm>
#include <string>

#include "datatypes/control.h"
#include "datatypes/parameter.h"
#include "datatypes/path.h"

namespace Kompass {
namespace Control {

std::string controlTypeToString(ControlType ctrlType);

class Controller {
 public:
  struct Result {
    enum class Status { GOAL_REACHED, LOOSING_GOAL, COMMAND_FOUND, NO_COMMAND_POSSIBLE };
    Status status;
    Control::Velocity2D velocity_command;
  };

  class ControllerParameters : public Parameters {
   public:
    ControllerParameters() : Parameters() {
      addParameter("enable_reverse_driving", Parameter(true));
      addParameter("enable_check_blocked", Parameter(false));
      addParameter("max_blocked_duration", Parameter(1.0, 0.1, 360.0));
      addParameter("reverse_slowdown_factor", Parameter(0.5, 0.01, 0.99));
    }
  };

  Controller();
  virtual ~Controller();

  void setLinearControlLimits(const Control::LinearVelocityControlParams &vx,
                              const Control::LinearVelocityControlParams &vy);
  void setAngularControlLimits(const Control::AngularVelocityControlParams &p);
  void setControlType(const Control::ControlType &controlType);
  void setCurrentVelocity(const Control::Velocity2D &vel);
  void setCurrentState(const Path::State &position);
  void setCurrentState(double pose_x, double pose_y, double pose_yaw, double speed);
  Control::ControlType getControlType() const;
  Control::Velocity2D getControl() const;
  double restrictVelocityTolimits(double currentVelocity, double targetVelocity,
                                  double accelerationLimit, double decelerationLimit,
                                  double maxVel, double timeStep) const;

 protected:
  // NOTE (reference quirk Q1): DWA never fills the first two, so they keep the value-initialised type (ACKERMANN) and
  // the default limits (1.0 / 1.0) -- getLinearVelocityCmdX and friends clamp with them.
  Control::ControlType drive_;
  Control::ControlLimitsParams limits_;
  Path::State pose_;               // where the robot is (setCurrentState)
  Control::Velocity2D velocity_;   // how it moves (setCurrentVelocity)
  Control::Velocity2D last_command_;
  int host_threads_;
  ControllerParameters config;
};

}  // namespace Control
}  // namespace Kompass
