// Shared plumbing of the compat veneer: Eigen (real one if installed, else the stand-in) and glog-style CHECKs.
#ifndef MAV_TRAJECTORY_GENERATION_MTG_COMPAT_BASE_H_
#define MAV_TRAJECTORY_GENERATION_MTG_COMPAT_BASE_H_

#if defined(__has_include)
#if __has_include(<Eigen/Core>) && !defined(MTG_FORCE_MINI_EIGEN)
#include <Eigen/Core>
#define MTG_HAVE_REAL_EIGEN 1
#endif
#endif
#ifndef MTG_HAVE_REAL_EIGEN
#include "../mtg_mini_eigen/Eigen/Core"
#endif

#if defined(__has_include)
#if __has_include(<glog/logging.h>)
#include <glog/logging.h>
#define MTG_HAVE_GLOG 1
#endif
#endif
#ifndef MTG_HAVE_GLOG
#include <cstdlib>
#include <iostream>
#include <sstream>
namespace mtg_compat {
// Reference convention: programmer errors abort with a message, like glog's CHECK.
class CheckFailure {
 public:
  CheckFailure(const char* file, int line, const char* what) { s_ << file << ":" << line << " Check failed: " << what << " "; }
  [[noreturn]] ~CheckFailure() { std::cerr << s_.str() << std::endl; std::abort(); }
  template <class T> CheckFailure& operator<<(const T& t) { s_ << t; return *this; }
 private:
  std::ostringstream s_;
};
struct Voidify { void operator&(const CheckFailure&) {} };
}  // namespace mtg_compat
#define CHECK(cond) (cond) ? (void)0 : ::mtg_compat::Voidify() & ::mtg_compat::CheckFailure(__FILE__, __LINE__, #cond)
#define CHECK_EQ(a, b) CHECK((a) == (b))
#define CHECK_GE(a, b) CHECK((a) >= (b))
#define CHECK_GT(a, b) CHECK((a) > (b))
#define CHECK_LT(a, b) CHECK((a) < (b))
#define CHECK_NOTNULL(p) (((p) == nullptr) ? (::mtg_compat::CheckFailure(__FILE__, __LINE__, #p " != nullptr"), (p)) : (p))
namespace mtg_compat {
// LOG(WARNING) / LOG(ERROR): one line on stderr, like glog's default sink
class LogLine {
 public:
  LogLine(const char* sev, const char* file, int line) { s_ << sev << " " << file << ":" << line << "] "; }
  ~LogLine() { std::cerr << s_.str() << std::endl; }
  template <class T> LogLine& operator<<(const T& t) { s_ << t; return *this; }
 private:
  std::ostringstream s_;
};
}  // namespace mtg_compat
#define LOG(severity) ::mtg_compat::LogLine(#severity, __FILE__, __LINE__)
#endif

#endif
