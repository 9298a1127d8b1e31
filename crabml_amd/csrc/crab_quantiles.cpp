// Median-quantile discretisation of the beta and gamma distributions: the class omegas of crabML's M7 / M8 / M8a
// (models/codon.py:581-605, 681-707, 782-808: scipy.stats.beta.ppf((2j+1)/(2K), p, q)) and M5 (:  gamma.ppf with
// scale 1/beta).  Host arithmetic, a few microseconds per model; the reference spends ~1 ms per call in scipy here,
// which is half of a whole evaluation once the likelihood runs on the GPU.
//
// Method: regularised incomplete beta / gamma by series and continued fractions (modified Lentz), inverted by a
// bracketed Newton iteration on u = ln x, so that quantiles of the steep distributions the optimiser visits
// (shape parameters down to 0.005: quantiles down to 1e-300) keep full relative accuracy.
#include "crab_quantiles.h"

#include <cmath>
#include <limits>

namespace crab {
namespace {

constexpr double EPS = 1.1102230246251565e-16;
constexpr double TINY = 1e-300;

// continued fraction of I_x(a, b) (convergent for x < (a+1)/(a+b+2))
double beta_cf(double a, double b, double x)
{
    const double qab = a + b, qap = a + 1.0, qam = a - 1.0;
    double c = 1.0, d = 1.0 - qab * x / qap;
    if (std::fabs(d) < TINY) d = TINY;
    d = 1.0 / d;
    double h = d;
    for (int m = 1; m <= 100000; ++m) {
        const double m2 = 2.0 * m;
        double aa = m * (b - m) * x / ((qam + m2) * (a + m2));
        d = 1.0 + aa * d;
        if (std::fabs(d) < TINY) d = TINY;
        c = 1.0 + aa / c;
        if (std::fabs(c) < TINY) c = TINY;
        d = 1.0 / d;
        h *= d * c;
        aa = -(a + m) * (qab + m) * x / ((a + m2) * (qap + m2));
        d = 1.0 + aa * d;
        if (std::fabs(d) < TINY) d = TINY;
        c = 1.0 + aa / c;
        if (std::fabs(c) < TINY) c = TINY;
        d = 1.0 / d;
        const double del = d * c;
        h *= del;
        if (std::fabs(del - 1.0) < EPS) break;
    }
    return h;
}

// I_x(a,b) and its complement, each computed directly where it is the small one; lx = ln x, lxc = ln(1-x)
void beta_cdf(double a, double b, double x, double xc, double lx, double lxc, double lbeta, double *F, double *Fc)
{
    if (x <= 0.0) { *F = 0.0; *Fc = 1.0; return; }
    if (xc <= 0.0) { *F = 1.0; *Fc = 0.0; return; }
    const double front = std::exp(a * lx + b * lxc - lbeta);
    if (x < (a + 1.0) / (a + b + 2.0)) {
        *F = front * beta_cf(a, b, x) / a;
        *Fc = 1.0 - *F;
    } else {
        *Fc = front * beta_cf(b, a, xc) / b;
        *F = 1.0 - *Fc;
    }
}

// regularised lower incomplete gamma P(a, x) and its complement
void gamma_cdf(double a, double x, double lx, double lgam, double *F, double *Fc)
{
    if (x <= 0.0) { *F = 0.0; *Fc = 1.0; return; }
    const double front = std::exp(a * lx - x - lgam);
    if (x < a + 1.0) {   // series
        double ap = a, del = 1.0 / a, sum = del;
        for (int n = 0; n < 100000; ++n) {
            ap += 1.0;
            del *= x / ap;
            sum += del;
            if (std::fabs(del) < std::fabs(sum) * EPS) break;
        }
        *F = sum * front;
        *Fc = 1.0 - *F;
    } else {             // continued fraction for Q(a, x)
        double b = x + 1.0 - a, c = 1.0 / TINY, d = 1.0 / b, h = d;
        for (int i = 1; i <= 100000; ++i) {
            const double an = -i * (i - a);
            b += 2.0;
            d = an * d + b;
            if (std::fabs(d) < TINY) d = TINY;
            c = b + an / c;
            if (std::fabs(c) < TINY) c = TINY;
            d = 1.0 / d;
            const double del = d * c;
            h *= del;
            if (std::fabs(del - 1.0) < EPS) break;
        }
        *Fc = front * h;
        *F = 1.0 - *Fc;
    }
}

// Solve F(e^u) = p for u in [ulo, uhi] by Newton on u with a bisection safeguard.  eval(u, &g, &dg) returns
// g = F - p (computed from whichever tail is small) and dg = dF/du = pdf(x) x.
template <typename Eval>
double solve_log(double ulo, double uhi, double u0, Eval &&eval)
{
    double u = u0;
    if (!(u > ulo && u < uhi)) u = 0.5 * (ulo + uhi);
    for (int it = 0; it < 400; ++it) {
        double g, dg;
        eval(u, &g, &dg);
        if (g == 0.0) return u;
        if (g < 0.0) ulo = u; else uhi = u;
        double un = (dg > 0.0 && std::isfinite(dg)) ? u - g / dg : std::numeric_limits<double>::quiet_NaN();
        if (!(un > ulo && un < uhi)) un = 0.5 * (ulo + uhi);
        const double tol = 2.0 * EPS * std::fmax(1.0, std::fabs(un));
        if (std::fabs(un - u) <= tol || uhi - ulo <= tol) return un;
        u = un;
    }
    return u;
}

}  // namespace

double beta_ppf(double p, double a, double b)
{
    if (!(a > 0.0) || !(b > 0.0) || !(p >= 0.0) || !(p <= 1.0)) return std::numeric_limits<double>::quiet_NaN();
    if (p == 0.0) return 0.0;
    if (p == 1.0) return 1.0;
    const double lbeta = std::lgamma(a) + std::lgamma(b) - std::lgamma(a + b);
    const double pc = 1.0 - p;
    auto eval = [&](double u, double *g, double *dg) {
        const double x = std::exp(u), xc = -std::expm1(u);
        const double lxc = xc > 0.0 ? std::log(xc) : -std::numeric_limits<double>::infinity();
        double F, Fc;
        beta_cdf(a, b, x, xc, u, lxc, lbeta, &F, &Fc);
        *g = (F < 0.5) ? F - p : pc - Fc;
        *dg = std::exp(a * u + (b - 1.0) * lxc - lbeta);   // pdf(x) * x
    };
    // lower-tail start: I_x ~ x^a / (a B(a,b))
    const double u0 = std::fmin(-1e-3, (std::log(p) + std::log(a) + lbeta) / a);
    const double u = solve_log(-745.0, 0.0, u0, eval);
    return std::exp(u);
}

double gamma_ppf(double p, double a, double scale)
{
    if (!(a > 0.0) || !(scale > 0.0) || !(p >= 0.0) || !(p <= 1.0)) return std::numeric_limits<double>::quiet_NaN();
    if (p == 0.0) return 0.0;
    if (p == 1.0) return std::numeric_limits<double>::infinity();
    const double lgam = std::lgamma(a), pc = 1.0 - p;
    auto eval = [&](double u, double *g, double *dg) {
        const double x = std::exp(u);
        double F, Fc;
        gamma_cdf(a, x, u, lgam, &F, &Fc);
        *g = (F < 0.5) ? F - p : pc - Fc;
        *dg = std::exp(a * u - x - lgam);
    };
    const double u0 = (std::log(p) + std::lgamma(a + 1.0)) / a;   // P(a,x) ~ x^a / Gamma(a+1)
    const double u = solve_log(-745.0, 709.0, u0, eval);
    return std::exp(u) * scale;
}

}  // namespace crab
