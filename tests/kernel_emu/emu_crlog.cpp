// TEST INFRASTRUCTURE: host build of csrc/cr_log.hpp (tests/test_cr_log.py)
#include "cr_log.hpp"
extern "C" void cr_log_many(const double* x, double* out, long n) { for (long i = 0; i < n; i++) out[i] = dartk::log_cr(x[i]); }
