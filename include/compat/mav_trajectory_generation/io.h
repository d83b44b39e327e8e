// Compat veneer: the reference's on-disk trajectory format (src/io.cpp:27-31, :126-218) without yaml-cpp:
//
//   segments:
//     - N: 10
//       D: 3
//       time: 3970847833  # [ns]
//       coefficients:
//         - [c0, c1, ..., c9]      (one flow sequence per dimension, increasing powers)
//         - [...]
//
// segmentsToFile / trajectoryToFile write exactly this block-style document; segmentsFromFile / trajectoryFromFile
// read it back (and any file the reference's yaml-cpp emitter produces for this schema: key order free, flow
// sequences may wrap over lines, comments allowed).  Segment times travel as truncated integer nanoseconds
// (Segment::getTimeNSec, segment.h:58-60).  The YAML::Node conversion functions (io.h:31-40) need yaml-cpp and are
// not provided; sampledTrajectoryStatesToFile depends on mav_msgs and is out of scope.
#ifndef MAV_TRAJECTORY_GENERATION_IO_H_
#define MAV_TRAJECTORY_GENERATION_IO_H_
#include <cctype>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "segment.h"
#include "trajectory.h"

namespace mav_trajectory_generation {

inline bool segmentsToFile(const std::string& filename, const Segment::Vector& segments) {
  std::ostringstream out;
  out << "segments:\n";
  char buf[40];
  for (const Segment& segment : segments) {
    out << "  - N: " << segment.N() << "\n";
    out << "    D: " << segment.D() << "\n";
    out << "    time: " << segment.getTimeNSec() << "  # [ns]\n";
    out << "    coefficients:\n";
    for (int i = 0; i < segment.D(); ++i) {
      out << "      - [";
      const Eigen::VectorXd c = segment[i].getCoefficients();
      for (int j = 0; j < segment.N(); ++j) {
        std::snprintf(buf, sizeof(buf), "%.17g", c[j]);   // round-trips every double
        out << (j ? ", " : "") << buf;
      }
      out << "]\n";
    }
  }
  std::ofstream fout(filename);
  if (!fout) return false;
  fout << out.str();
  fout.close();
  return !fout.fail();
}

inline bool trajectoryToFile(const std::string& filename, const Trajectory& trajectory) {
  Segment::Vector segments;
  trajectory.getSegments(&segments);
  return segmentsToFile(filename, segments);
}

namespace mtg_compat_detail {
// Token stream over the YAML subset above: comments stripped, punctuation "-[]," split off, "key:" kept whole.
inline std::vector<std::string> yamlTokens(const std::string& text) {
  std::vector<std::string> tokens;
  std::string cur;
  auto flush = [&]() { if (!cur.empty()) { tokens.push_back(cur); cur.clear(); } };
  for (size_t i = 0; i < text.size(); ++i) {
    const char ch = text[i];
    if (ch == '#') { flush(); while (i < text.size() && text[i] != '\n') ++i; continue; }
    if (std::isspace((unsigned char)ch)) { flush(); continue; }
    if (ch == '[' || ch == ']' || ch == ',') { flush(); tokens.push_back(std::string(1, ch)); continue; }
    // a '-' starts a block-sequence entry only when followed by whitespace; otherwise it is a sign
    if (ch == '-' && cur.empty() && i + 1 < text.size() && std::isspace((unsigned char)text[i + 1])) {
      tokens.push_back("-");
      continue;
    }
    cur.push_back(ch);
  }
  flush();
  return tokens;
}
inline bool parseDouble(const std::string& s, double* v) {
  char* end = nullptr;
  *v = std::strtod(s.c_str(), &end);
  if (end != s.c_str() && *end == '\0') return true;
  // YAML spellings yaml-cpp emits for non-finite values
  if (s == ".inf" || s == ".Inf" || s == "+.inf") { *v = HUGE_VAL; return true; }
  if (s == "-.inf" || s == "-.Inf") { *v = -HUGE_VAL; return true; }
  if (s == ".nan" || s == ".NaN") { *v = std::nan(""); return true; }
  return false;
}
}  // namespace mtg_compat_detail

inline bool segmentsFromFile(const std::string& filename, Segment::Vector* segments) {
  CHECK_NOTNULL(segments);
  std::ifstream in(filename);
  if (!in.good()) return false;
  segments->clear();
  std::stringstream ss;
  ss << in.rdbuf();
  const std::vector<std::string> tok = mtg_compat_detail::yamlTokens(ss.str());
  size_t i = 0;
  while (i < tok.size() && tok[i] != "segments:") ++i;
  if (i == tok.size()) return false;   // "No segments element."
  ++i;
  while (i < tok.size()) {
    if (tok[i] != "-") return false;
    ++i;
    long n = -1, d = -1;
    bool have_time = false, have_coeffs = false;
    uint64_t t_ns = 0;
    std::vector<std::vector<double>> coeffs;
    // one map: keys until the next top-level "-" (sequence entries inside "coefficients:" are consumed below)
    while (i < tok.size() && tok[i] != "-") {
      const std::string key = tok[i++];
      if (key == "N:" || key == "D:" || key == "time:") {
        if (i >= tok.size()) return false;
        char* end = nullptr;
        const unsigned long long v = std::strtoull(tok[i].c_str(), &end, 10);
        if (end == tok[i].c_str() || *end != '\0') return false;
        ++i;
        if (key == "N:") n = (long)v; else if (key == "D:") d = (long)v; else { t_ns = v; have_time = true; }
      } else if (key == "coefficients:") {
        have_coeffs = true;
        while (i + 1 < tok.size() && tok[i] == "-" && tok[i + 1] == "[") {
          i += 2;
          std::vector<double> row;
          while (i < tok.size() && tok[i] != "]") {
            if (tok[i] == ",") { ++i; continue; }
            double v;
            if (!mtg_compat_detail::parseDouble(tok[i], &v)) return false;
            row.push_back(v);
            ++i;
          }
          if (i == tok.size()) return false;
          ++i;   // "]"
          coeffs.push_back(row);
        }
      } else {
        return false;   // unknown key
      }
    }
    if (n < 0 || d < 0 || !have_time || !have_coeffs) return false;   // "Wrong format, missing elements."
    if ((long)coeffs.size() != d) return false;                       // "Coefficients and dimensions do not coincide."
    Segment segment((int)n, (int)d);
    segment.setTimeNSec(t_ns);
    for (long j = 0; j < d; ++j) {
      if ((long)coeffs[j].size() != n) return false;                  // "Number of coefficients does no coincide."
      Eigen::VectorXd v((int)n);
      for (long k = 0; k < n; ++k) v[(int)k] = coeffs[j][k];
      segment[j] = Polynomial((int)n, v);
    }
    segments->push_back(segment);
  }
  return true;
}

inline bool trajectoryFromFile(const std::string& filename, Trajectory* trajectory) {
  CHECK_NOTNULL(trajectory);
  Segment::Vector segments;
  if (!segmentsFromFile(filename, &segments)) return false;
  if (segments.empty()) return false;
  trajectory->setSegments(segments);
  return true;
}

}  // namespace mav_trajectory_generation
#endif
