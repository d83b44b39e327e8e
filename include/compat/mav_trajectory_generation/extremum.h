// Compat veneer: extremum record (reference: extremum.h:29-45).
#ifndef MAV_TRAJECTORY_GENERATION_EXTREMUM_H_
#define MAV_TRAJECTORY_GENERATION_EXTREMUM_H_
#include <ostream>
namespace mav_trajectory_generation {
struct Extremum {
  Extremum() : time(0.0), value(0.0), segment_idx(0) {}
  Extremum(double _time, double _value, int _segment_idx) : time(_time), value(_value), segment_idx(_segment_idx) {}
  bool operator<(const Extremum& rhs) const { return value < rhs.value; }
  bool operator>(const Extremum& rhs) const { return value > rhs.value; }
  double time;      // relative to the segment start
  double value;
  int segment_idx;
};
inline std::ostream& operator<<(std::ostream& s, const Extremum& e) {
  return s << "time: " << e.time << ", value: " << e.value << ", segment idx: " << e.segment_idx << std::endl;
}
}  // namespace mav_trajectory_generation
#endif
