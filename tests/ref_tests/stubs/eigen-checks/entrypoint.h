// Stand-in for eigen_checks/entrypoint.h: UNITTEST_ENTRYPOINT = a main() that parses the flags and runs every test.
#ifndef MTG_EIGEN_CHECKS_ENTRYPOINT_H_
#define MTG_EIGEN_CHECKS_ENTRYPOINT_H_
#include <gtest/gtest.h>
#define UNITTEST_ENTRYPOINT                      \
  int main(int argc, char** argv) {              \
    ::testing::InitGoogleTest(&argc, argv);      \
    return RUN_ALL_TESTS();                      \
  }
#endif
