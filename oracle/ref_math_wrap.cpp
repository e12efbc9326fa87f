// oracle/ref_math_wrap.cpp -- C entry point over the REFERENCE's own include/math/math.hpp (KahanSum), compiled from
// the header where it lies under /root/reference (it includes nothing).  TEST INFRASTRUCTURE: pins the oracle's and the
// product's Kahan accumulation (the forward-axis sum of fit_motion, src/fit_motion.cc:166-167,233) to the real code.
#include <math/math.hpp>

extern "C" int ref_kahan_sum(const double* values, int n, int dim, double* sum)
{
    for (int k = 0; k < dim; k++) {                        // KahanSum<Eigen::Vector3d> works component by component
        pilotguru::KahanSum<double> acc(0.0, 0.0);
        for (int i = 0; i < n; i++) acc.add(values[(long)i * dim + k]);
        sum[k] = acc.sum();
    }
    return 0;
}
