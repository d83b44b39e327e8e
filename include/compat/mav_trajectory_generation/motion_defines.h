// Compat veneer: derivative-order constants (reference: motion_defines.h:28-40).
#ifndef MAV_TRAJECTORY_GENERATION_MOTION_DEFINES_H_
#define MAV_TRAJECTORY_GENERATION_MOTION_DEFINES_H_
namespace mav_trajectory_generation {
namespace derivative_order {
static constexpr int POSITION = 0, VELOCITY = 1, ACCELERATION = 2, JERK = 3, SNAP = 4;
static constexpr int ORIENTATION = 0, ANGULAR_VELOCITY = 1, ANGULAR_ACCELERATION = 2;
static constexpr int INVALID = -1;
}  // namespace derivative_order
}  // namespace mav_trajectory_generation
#endif
