// Stand-in for the glog macros used by the reference's hot-path translation units (glog is an un-vendored dependency
// of the reference, absent from this image).  CHECK* failures print and abort(), as glog does; LOG(WARNING/ERROR/INFO),
// DLOG and VLOG are swallowed unless MTG_REF_VERBOSE is set.  Test infrastructure only (oracle/_ref).
#pragma once
#include <cstdlib>
#include <iostream>
#include <sstream>

namespace mtg_glog_shim {
enum Severity { INFO = 0, WARNING = 1, ERROR = 2, FATAL = 3 };

class Message {
 public:
  Message(const char* file, int line, int severity) : severity_(severity) { s_ << file << ":" << line << "] "; }
  ~Message() {
    if (severity_ >= FATAL) {
      std::cerr << "F " << s_.str() << std::endl;
      std::abort();
    }
    static const bool verbose = std::getenv("MTG_REF_VERBOSE") != nullptr;
    if (verbose) std::cerr << "IWEF"[severity_ & 3] << " " << s_.str() << std::endl;
  }
  std::ostream& stream() { return s_; }

 private:
  std::ostringstream s_;
  int severity_;
};

struct Voidify {
  void operator&(std::ostream&) {}
};

template <class T>
T* check_not_null(const char* file, int line, const char* what, T* p) {
  if (p == nullptr) Message(file, line, FATAL).stream() << "'" << what << "' Must be non NULL";
  return p;
}
}  // namespace mtg_glog_shim

#define MTG_GLOG_MSG(sev) ::mtg_glog_shim::Message(__FILE__, __LINE__, ::mtg_glog_shim::sev).stream()
#define LOG(sev) MTG_GLOG_MSG(sev)
#define DLOG(sev) MTG_GLOG_MSG(sev)
#define VLOG(n) MTG_GLOG_MSG(INFO)
#define LOG_IF(sev, cond) !(cond) ? (void)0 : ::mtg_glog_shim::Voidify() & MTG_GLOG_MSG(sev)
#define CHECK(cond) (cond) ? (void)0 : ::mtg_glog_shim::Voidify() & MTG_GLOG_MSG(FATAL) << "Check failed: " #cond " "
#define MTG_GLOG_CHECK_OP(a, b, op) CHECK((a)op(b))
#define CHECK_EQ(a, b) MTG_GLOG_CHECK_OP(a, b, ==)
#define CHECK_NE(a, b) MTG_GLOG_CHECK_OP(a, b, !=)
#define CHECK_LT(a, b) MTG_GLOG_CHECK_OP(a, b, <)
#define CHECK_LE(a, b) MTG_GLOG_CHECK_OP(a, b, <=)
#define CHECK_GT(a, b) MTG_GLOG_CHECK_OP(a, b, >)
#define CHECK_GE(a, b) MTG_GLOG_CHECK_OP(a, b, >=)
#define CHECK_NOTNULL(p) ::mtg_glog_shim::check_not_null(__FILE__, __LINE__, #p, (p))
