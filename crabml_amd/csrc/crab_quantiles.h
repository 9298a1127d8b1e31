// Quantile functions behind the median-quantile discretisation of the site-class models (crab_quantiles.cpp).
#pragma once
namespace crab {
// x with I_x(a, b) = p  (scipy.stats.beta.ppf(p, a, b))
double beta_ppf(double p, double a, double b);
// x with P(a, x / scale) = p  (scipy.stats.gamma.ppf(p, a, scale=scale))
double gamma_ppf(double p, double a, double scale);
// Median quantiles of a two-component mixture, the class omegas of crabML's M6 and M9 (models/codon.py:982-1020, 1183-1222):
// out[j] solves cdf(x) = (2j+1)/(2K) on [1e-4, 99] by Brent's method exactly as scipy.optimize.brentq runs it with its default
// tolerances, with the reference's fallback 10 p where the CDF does not change sign over the bracket.
//   kind 2 (M6): cdf = p0 Gamma(shape a1, rate b1) + (1 - p0) Gamma(shape a2, rate a2);   params = {p0, a1, b1, a2}
//   kind 3 (M9): cdf = p0 Beta(pb, qb) + (1 - p0) Gamma(shape a, rate b);                 params = {p0, pb, qb, a, b}
// Returns false if Brent's iteration does not converge in 100 steps (scipy raises RuntimeError there).
bool mixture_median_quantiles(int kind, const double *params, int K, double *out);
}  // namespace crab
