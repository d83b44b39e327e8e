// Minimal stand-in for the part of googletest the reference's test files use (gtest is not in this image; no network):
// value-parameterised fixtures (TEST_P / TestWithParam / INSTANTIATE_TEST_CASE_P + ::testing::Values), plain TEST, the
// EXPECT_* / ASSERT_* comparison macros with `<< message` streaming, a --gtest_filter with '*' wildcards, ':' alternatives and
// a '-' negative part, RUN_ALL_TESTS().  Test infrastructure only: it lets /root/reference's own test sources compile
// UNMODIFIED against the replacement (tests/ref_tests/README.md).
#ifndef MTG_MINI_GTEST_H_
#define MTG_MINI_GTEST_H_
#include <cmath>
#include <cstdio>
#include <functional>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

namespace testing {

class Test {
 public:
  virtual ~Test() {}
  virtual void SetUp() {}
  virtual void TearDown() {}
  virtual void TestBody() = 0;
};

template <class T>
class WithParamInterface {
 public:
  typedef T ParamType;
  const T& GetParam() const { return *param_; }
  static const T*& slot() { static const T* p = nullptr; return p; }
  WithParamInterface() : param_(slot()) {}
 private:
  const T* param_;
};
template <class T>
class TestWithParam : public Test, public WithParamInterface<T> {};

struct Registry {
  struct Case { std::string name; std::function<void()> run; };
  std::vector<Case> cases;
  int failures_in_current = 0;
  bool fatal = false;
  static Registry& get() { static Registry r; return r; }
};

// parameterised: the patterns (suite, test, factory) and the instantiations (prefix, suite, values) meet at RUN_ALL_TESTS
template <class Fixture>
struct ParamSuite {
  typedef typename Fixture::ParamType P;
  struct Pattern { std::string test; std::function<Test*()> make; };
  struct Inst { std::string prefix; std::vector<P> values; };
  std::vector<Pattern> patterns;
  std::vector<Inst> insts;
  std::string suite;
  bool expanded = false;
  static ParamSuite& get() { static ParamSuite s; return s; }
  void expand() {
    if (expanded) return;
    expanded = true;
    for (auto& in : insts)
      for (auto& pt : patterns)
        for (size_t i = 0; i < in.values.size(); ++i) {
          const P* val = &in.values[i];
          auto make = pt.make;
          Registry::get().cases.push_back({in.prefix + "/" + suite + "." + pt.test + "/" + std::to_string(i), [val, make]() {
            WithParamInterface<P>::slot() = val;
            std::unique_ptr<Test> t(make());
            t->SetUp();
            if (!Registry::get().fatal) t->TestBody();
            t->TearDown();
          }});
        }
  }
};
struct Expanders {
  std::vector<std::function<void()>> fns;
  static Expanders& get() { static Expanders e; return e; }
};
template <class Fixture>
int RegisterPattern(const char* suite, const char* test, std::function<Test*()> make) {
  auto& s = ParamSuite<Fixture>::get();
  if (s.patterns.empty() && s.insts.empty()) Expanders::get().fns.push_back([]() { ParamSuite<Fixture>::get().expand(); });
  s.suite = suite;
  s.patterns.push_back({test, make});
  return 0;
}
template <class Fixture>
int RegisterInst(const char* prefix, const char* suite, std::vector<typename Fixture::ParamType> values) {
  auto& s = ParamSuite<Fixture>::get();
  if (s.patterns.empty() && s.insts.empty()) Expanders::get().fns.push_back([]() { ParamSuite<Fixture>::get().expand(); });
  s.suite = suite;
  s.insts.push_back({prefix, std::move(values)});
  return 0;
}
template <class T, class... Ts>
std::vector<T> Values(T first, Ts... rest) { return std::vector<T>{first, rest...}; }

inline int RegisterPlain(const char* suite, const char* test, std::function<Test*()> make) {
  Registry::get().cases.push_back({std::string(suite) + "." + test, [make]() {
    std::unique_ptr<Test> t(make());
    t->SetUp();
    if (!Registry::get().fatal) t->TestBody();
    t->TearDown();
  }});
  return 0;
}

// a failed expectation: prints file:line, the expression text and whatever was streamed into it
class Failure {
 public:
  Failure(const char* file, int line, const std::string& what, bool fatal) : fatal_(fatal) {
    s_ << file << ":" << line << ": Failure\n" << what << "\n";
  }
  ~Failure() {
    std::cout << s_.str() << std::endl;
    ++Registry::get().failures_in_current;
    if (fatal_) Registry::get().fatal = true;
  }
  template <class T> Failure& operator<<(const T& t) { s_ << t; return *this; }
  Failure& operator<<(std::ostream& (*f)(std::ostream&)) { s_ << f; return *this; }
 private:
  std::ostringstream s_;
  bool fatal_;
};
struct Voidify { void operator&(const Failure&) {} };

// result of a predicate that carries its own explanation (EIGEN_MATRIX_NEAR)
class AssertionResult {
 public:
  explicit AssertionResult(bool ok) : ok_(ok) {}
  operator bool() const { return ok_; }
  template <class T> AssertionResult& operator<<(const T& t) { std::ostringstream s; s << t; msg_ += s.str(); return *this; }
  const std::string& message() const { return msg_; }
 private:
  bool ok_;
  std::string msg_;
};
inline AssertionResult AssertionSuccess() { return AssertionResult(true); }
inline AssertionResult AssertionFailure() { return AssertionResult(false); }
inline std::string explain(bool) { return ""; }
inline std::string explain(const AssertionResult& r) { return r.message(); }

template <class A, class B>
std::string cmp_text(const char* ea, const char* eb, const char* op, const A& a, const B& b) {
  std::ostringstream s;
  s << "Expected: (" << ea << ") " << op << " (" << eb << "), actual: " << a << " vs " << b;
  return s.str();
}

inline bool wildcard(const char* p, const char* s) {
  if (!*p) return !*s;
  if (*p == '*') return wildcard(p + 1, s) || (*s && wildcard(p, s + 1));
  return *s && (*p == '?' || *p == *s) && wildcard(p + 1, s + 1);
}
inline bool any_match(const std::string& pats, const std::string& name) {
  size_t b = 0;
  while (b <= pats.size()) {
    size_t e = pats.find(':', b);
    if (e == std::string::npos) e = pats.size();
    if (e > b && wildcard(pats.substr(b, e - b).c_str(), name.c_str())) return true;
    b = e + 1;
  }
  return false;
}
struct Flags { std::string filter = "*"; static Flags& get() { static Flags f; return f; } };
inline void InitGoogleTest(int* argc, char** argv) {
  for (int i = 1; i < *argc; ++i) {
    const std::string a = argv[i];
    if (a.rfind("--gtest_filter=", 0) == 0) Flags::get().filter = a.substr(15);
  }
}
inline int RunAllTests() {
  for (auto& f : Expanders::get().fns) f();
  std::string pos = Flags::get().filter, neg;
  const size_t dash = pos.find('-');
  if (dash != std::string::npos) { neg = pos.substr(dash + 1); pos = pos.substr(0, dash); }
  if (pos.empty()) pos = "*";
  int ran = 0, failed = 0, skipped = 0;
  std::vector<std::string> failed_names;
  for (auto& c : Registry::get().cases) {
    if (!any_match(pos, c.name) || (!neg.empty() && any_match(neg, c.name))) { ++skipped; continue; }
    std::cout << "[ RUN      ] " << c.name << std::endl;
    Registry::get().failures_in_current = 0;
    Registry::get().fatal = false;
    c.run();
    ++ran;
    if (Registry::get().failures_in_current) {
      ++failed;
      failed_names.push_back(c.name);
      std::cout << "[  FAILED  ] " << c.name << std::endl;
    } else {
      std::cout << "[       OK ] " << c.name << std::endl;
    }
  }
  std::cout << "[==========] " << ran << " tests ran, " << skipped << " filtered out." << std::endl;
  std::cout << "[  PASSED  ] " << ran - failed << " tests." << std::endl;
  for (auto& n : failed_names) std::cout << "[  FAILED  ] " << n << std::endl;
  return failed ? 1 : 0;
}
}  // namespace testing

#define RUN_ALL_TESTS() ::testing::RunAllTests()
#define MTG_GT_CLASS(suite, test) suite##_##test##_Test

#define TEST_P(suite, test)                                                                                        \
  class MTG_GT_CLASS(suite, test) : public suite {                                                                 \
   public:                                                                                                         \
    void TestBody() override;                                                                                      \
  };                                                                                                               \
  static int mtg_gt_reg_##suite##_##test =                                                                         \
      ::testing::RegisterPattern<suite>(#suite, #test, []() -> ::testing::Test* { return new MTG_GT_CLASS(suite, test); }); \
  void MTG_GT_CLASS(suite, test)::TestBody()

#define INSTANTIATE_TEST_CASE_P(prefix, suite, values) \
  static int mtg_gt_inst_##prefix##_##suite = ::testing::RegisterInst<suite>(#prefix, #suite, values)
#define INSTANTIATE_TEST_SUITE_P(prefix, suite, values) INSTANTIATE_TEST_CASE_P(prefix, suite, values)

#define TEST(suite, test)                                                                                          \
  class MTG_GT_CLASS(suite, test) : public ::testing::Test {                                                       \
   public:                                                                                                         \
    void TestBody() override;                                                                                      \
  };                                                                                                               \
  static int mtg_gt_reg_##suite##_##test =                                                                         \
      ::testing::RegisterPlain(#suite, #test, []() -> ::testing::Test* { return new MTG_GT_CLASS(suite, test); }); \
  void MTG_GT_CLASS(suite, test)::TestBody()
#define TEST_F(suite, test)                                                                                        \
  class MTG_GT_CLASS(suite, test) : public suite {                                                                 \
   public:                                                                                                         \
    void TestBody() override;                                                                                      \
  };                                                                                                               \
  static int mtg_gt_reg_##suite##_##test =                                                                         \
      ::testing::RegisterPlain(#suite, #test, []() -> ::testing::Test* { return new MTG_GT_CLASS(suite, test); }); \
  void MTG_GT_CLASS(suite, test)::TestBody()

// `cond ? (void)0 : Voidify() & Failure(...) << user message`  -- the streamed message binds to the Failure object
#define MTG_GT_CHECK(ok, text, fatal) \
  (ok) ? (void)0 : ::testing::Voidify() & ::testing::Failure(__FILE__, __LINE__, text, fatal)
#define MTG_GT_BOOL(expr, want, fatal)                                                                              \
  if (const auto& mtg_gt_r = (expr); static_cast<bool>(mtg_gt_r) == want) {                                        \
  } else                                                                                                           \
    ::testing::Voidify() & ::testing::Failure(__FILE__, __LINE__, std::string("Value of: " #expr "\n  Actual: ") + (want ? "false" : "true") + \
                                                                      "\nExpected: " + (want ? "true" : "false") + "\n" + ::testing::explain(mtg_gt_r), fatal)
#define MTG_GT_CMP(a, b, op, fatal)                                                                                 \
  if (const auto& mtg_gt_a = (a); true)                                                                            \
    if (const auto& mtg_gt_b = (b); mtg_gt_a op mtg_gt_b) {                                                        \
    } else                                                                                                         \
      ::testing::Voidify() & ::testing::Failure(__FILE__, __LINE__, ::testing::cmp_text(#a, #b, #op, mtg_gt_a, mtg_gt_b), fatal)
#define MTG_GT_NEAR(a, b, tol, fatal)                                                                               \
  if (const double mtg_gt_d = std::fabs(double(a) - double(b)); mtg_gt_d <= double(tol)) {                         \
  } else                                                                                                           \
    ::testing::Voidify() & ::testing::Failure(__FILE__, __LINE__, std::string("The difference between " #a " and " #b " is ") + std::to_string(mtg_gt_d) + \
                                                                      ", which exceeds " #tol, fatal)

#define EXPECT_TRUE(e) MTG_GT_BOOL(e, true, false)
#define EXPECT_FALSE(e) MTG_GT_BOOL(e, false, false)
#define ASSERT_TRUE(e) MTG_GT_BOOL(e, true, true)
#define ASSERT_FALSE(e) MTG_GT_BOOL(e, false, true)
#define EXPECT_EQ(a, b) MTG_GT_CMP(a, b, ==, false)
#define EXPECT_NE(a, b) MTG_GT_CMP(a, b, !=, false)
#define EXPECT_LT(a, b) MTG_GT_CMP(a, b, <, false)
#define EXPECT_LE(a, b) MTG_GT_CMP(a, b, <=, false)
#define EXPECT_GT(a, b) MTG_GT_CMP(a, b, >, false)
#define EXPECT_GE(a, b) MTG_GT_CMP(a, b, >=, false)
#define ASSERT_EQ(a, b) MTG_GT_CMP(a, b, ==, true)
#define ASSERT_NE(a, b) MTG_GT_CMP(a, b, !=, true)
#define ASSERT_LT(a, b) MTG_GT_CMP(a, b, <, true)
#define ASSERT_LE(a, b) MTG_GT_CMP(a, b, <=, true)
#define ASSERT_GT(a, b) MTG_GT_CMP(a, b, >, true)
#define ASSERT_GE(a, b) MTG_GT_CMP(a, b, >=, true)
#define EXPECT_NEAR(a, b, tol) MTG_GT_NEAR(a, b, tol, false)
#define ASSERT_NEAR(a, b, tol) MTG_GT_NEAR(a, b, tol, true)
#define EXPECT_DOUBLE_EQ(a, b) MTG_GT_NEAR(a, b, 4 * 2.220446049250313e-16 * std::fmax(std::fabs(double(a)), std::fabs(double(b))), false)
#endif
