/* oracle/iir_ref.c — C restatement of the reference's direct-form-II-transposed IIR loop
 * (friture/signal/lfilter.py:131-139).  TEST INFRASTRUCTURE: a faster stand-in for the pure
 * Python loop in oracle/dsp.py:lfilter_df2t with the same IEEE double operations in the same
 * order (built with -ffp-contract=off so no multiply-add is fused).  Never linked into the
 * product library.
 */
#include <stddef.h>

void oracle_lfilter_df2t(const double* b, const double* a, int nb, const double* x, long nx, double* z, double* y) {
    if (nb <= 1) {
        for (long k = 0; k < nx; ++k) y[k] = x[k] * b[0];
        return;
    }
    for (long k = 0; k < nx; ++k) {
        const double xk = x[k];
        const double yk = z[0] + b[0] * xk;
        y[k] = yk;
        for (int n = 0; n < nb - 2; ++n) z[n] = z[n + 1] + xk * b[n + 1] - yk * a[n + 1];
        z[nb - 2] = xk * b[nb - 1] - yk * a[nb - 1];
    }
}
