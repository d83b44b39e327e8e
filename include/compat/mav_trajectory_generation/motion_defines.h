// Compat veneer: derivative-order constants (reference: motion_defines.h:28-40).
#ifndef MAV_TRAJECTORY_GENERATION_MOTION_DEFINES_H_
#define MAV_TRAJECTORY_GENERATION_MOTION_DEFINES_H_
#include <string>
namespace mav_trajectory_generation {
namespace derivative_order {
static constexpr int POSITION = 0, VELOCITY = 1, ACCELERATION = 2, JERK = 3, SNAP = 4;
static constexpr int ORIENTATION = 0, ANGULAR_VELOCITY = 1, ANGULAR_ACCELERATION = 2;
static constexpr int INVALID = -1;
}  // namespace derivative_order

// names of the position derivatives 0 .. 4 and back (reference: motion_defines.h:42-46, src/motion_defines.cpp)
inline std::string positionDerivativeToString(int derivative) {
  switch (derivative) {
    case 0: return "position";
    case 1: return "velocity";
    case 2: return "acceleration";
    case 3: return "jerk";
    case 4: return "snap";
    default: return "invalid";
  }
}
inline int positionDerivativeToInt(const std::string& name) {
  for (int d = 0; d <= 4; ++d)
    if (positionDerivativeToString(d) == name) return d;
  return derivative_order::INVALID;
}
inline std::string orientationDerivativeToString(int derivative) {
  switch (derivative) {
    case 0: return "orientation";
    case 1: return "angular_velocity";
    case 2: return "angular_acceleration";
    default: return "invalid";
  }
}
inline int orientationDerivativeToInt(const std::string& name) {
  for (int d = 0; d <= 2; ++d)
    if (orientationDerivativeToString(d) == name) return d;
  return derivative_order::INVALID;
}
}  // namespace mav_trajectory_generation
#endif
