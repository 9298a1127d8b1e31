// Quantile functions behind the median-quantile discretisation of the site-class models (crab_quantiles.cpp).
#pragma once
namespace crab {
// x with I_x(a, b) = p  (scipy.stats.beta.ppf(p, a, b))
double beta_ppf(double p, double a, double b);
// x with P(a, x / scale) = p  (scipy.stats.gamma.ppf(p, a, scale=scale))
double gamma_ppf(double p, double a, double scale);
}  // namespace crab
