/* test helper: the HOST C library's powf / cbrtf on arrays (tests/test_oracle_golden.py compares
 * oracle/glibc_flt32.h with them where the host runs glibc 2.35) */
#include <math.h>
#include <stdint.h>
void libm_powf_vec(const float *x, float y, int64_t n, float *out)
{
    for (int64_t i = 0; i < n; ++i) out[i] = powf(x[i], y);
}
void libm_cbrtf_vec(const float *x, int64_t n, float *out)
{
    for (int64_t i = 0; i < n; ++i) out[i] = cbrtf(x[i]);
}
