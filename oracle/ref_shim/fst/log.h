// Test-infrastructure shim: stands in for the reference's utils/log.h (glog)
// so that runtime/core/frontend/fbank.h + fft.cc compile stand-alone from
// where they lie under /root/reference (oracle/Makefile).  Not product code.
#ifndef ORACLE_REF_SHIM_UTILS_LOG_H_
#define ORACLE_REF_SHIM_UTILS_LOG_H_
#include <cstdlib>
#include <iostream>
struct OracleNullLog {
  template <class T> OracleNullLog& operator<<(const T&) { return *this; }
};
#define ORACLE_CHECK_IMPL(x)                                   \
  if (!(x)) {                                                  \
    std::cerr << "CHECK failed: " #x << std::endl;             \
    abort();                                                   \
  } else                                                       \
    OracleNullLog()
#define CHECK(x) ORACLE_CHECK_IMPL(x)
#define CHECK_GE(a, b) ORACLE_CHECK_IMPL((a) >= (b))
#define CHECK_GT(a, b) ORACLE_CHECK_IMPL((a) > (b))
#define CHECK_EQ(a, b) ORACLE_CHECK_IMPL((a) == (b))
#define CHECK_NE(a, b) ORACLE_CHECK_IMPL((a) != (b))
#define CHECK_LE(a, b) ORACLE_CHECK_IMPL((a) <= (b))
#define CHECK_LT(a, b) ORACLE_CHECK_IMPL((a) < (b))
#define LOG(x) OracleNullLog()
#define VLOG(x) OracleNullLog()
#endif
