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


// ---- M6 / M9: quantiles of a mixture CDF ------------------------------------------------------------------------------------
namespace {

// scipy.stats.beta.cdf(x, a, b) / scipy.stats.gamma.cdf(x, a, scale = 1 / rate) on the bracket the reference searches
double beta_cdf_value(double x, double a, double b, double lbeta)
{
    if (x <= 0.0) return 0.0;
    if (x >= 1.0) return 1.0;
    double F, Fc;
    beta_cdf(a, b, x, 1.0 - x, std::log(x), std::log1p(-x), lbeta, &F, &Fc);
    return F;
}
double gamma_cdf_value(double x, double a, double rate, double lgam)
{
    const double scale = 1.0 / rate, y = x / scale;   // (scipy's loc-scale wrapper: (x - loc) / scale with the reference's scale = 1.0 / rate)
    if (y <= 0.0) return 0.0;
    double F, Fc;
    gamma_cdf(a, y, std::log(y), lgam, &F, &Fc);
    return F;
}

// Brent's method in the form of scipy.optimize.brentq (scipy 1.15, scipy/optimize/Zeros/brentq.c; a dependency of the
// reference, not part of it: restated from the published algorithm, same variables, same order of the tests, so that the
// iterates -- and with them the last digits the reference's omegas carry at xtol = 2e-12 -- are the reference's).
// status: 0 converged, 1 no sign change over [xa, xb], 2 not converged in `iter` steps.
template <typename F>
double brentq_like_scipy(F &&f, double xa, double xb, double xtol, double rtol, int iter, int *status)
{
    double xpre = xa, xcur = xb, xblk = 0.0, fpre, fcur, fblk = 0.0, spre = 0.0, scur = 0.0, sbis, delta, stry, dpre, dblk;
    *status = 0;
    fpre = f(xpre);
    fcur = f(xcur);
    if (fpre == 0.0) return xpre;
    if (fcur == 0.0) return xcur;
    if (std::signbit(fpre) == std::signbit(fcur)) {
        *status = 1;
        return 0.0;
    }
    for (int i = 0; i < iter; ++i) {
        if (fpre != 0.0 && fcur != 0.0 && std::signbit(fpre) != std::signbit(fcur)) {
            xblk = xpre;
            fblk = fpre;
            spre = scur = xcur - xpre;
        }
        if (std::fabs(fblk) < std::fabs(fcur)) {
            xpre = xcur;
            xcur = xblk;
            xblk = xpre;
            fpre = fcur;
            fcur = fblk;
            fblk = fpre;
        }
        delta = (xtol + rtol * std::fabs(xcur)) / 2.0;
        sbis = (xblk - xcur) / 2.0;
        if (fcur == 0.0 || std::fabs(sbis) < delta) return xcur;
        if (std::fabs(spre) > delta && std::fabs(fcur) < std::fabs(fpre)) {
            if (xpre == xblk) {   // interpolate
                stry = -fcur * (xcur - xpre) / (fcur - fpre);
            } else {              // extrapolate
                dpre = (fpre - fcur) / (xpre - xcur);
                dblk = (fblk - fcur) / (xblk - xcur);
                stry = -fcur * (fblk * dblk - fpre * dpre) / (dblk * dpre * (fblk - fpre));
            }
            if (2.0 * std::fabs(stry) < std::fmin(std::fabs(spre), 3.0 * std::fabs(sbis) - delta)) {   // good short step
                spre = scur;
                scur = stry;
            } else {              // bisect
                spre = sbis;
                scur = sbis;
            }
        } else {                  // bisect
            spre = sbis;
            scur = sbis;
        }
        xpre = xcur;
        fpre = fcur;
        if (std::fabs(scur) > delta)
            xcur += scur;
        else
            xcur += (sbis > 0.0 ? delta : -delta);
        fcur = f(xcur);
    }
    *status = 2;
    return xcur;
}

}  // namespace

bool mixture_median_quantiles(int kind, const double *params, int K, double *out)
{
    // the reference's constructor clamps (models/codon.py:1168-1172 M6, :966-971 M9)
    const double p0 = std::fmin(std::fmax(params[0], 1e-6), 1.0 - 1e-6);
    double a1 = 0, b1 = 0, a2 = 0, b2 = 0, lg1 = 0, lg2 = 0, lbeta = 0;
    if (kind == 2) {
        a1 = std::fmax(params[1], 0.005), b1 = std::fmax(params[2], 0.005), a2 = std::fmax(params[3], 0.005), b2 = a2;
        lg1 = std::lgamma(a1), lg2 = std::lgamma(a2);
    } else {
        a1 = std::fmax(params[1], 0.005), b1 = std::fmax(params[2], 0.005);   // beta(p, q)
        a2 = std::fmax(params[3], 0.005), b2 = std::fmax(params[4], 0.005);   // gamma(shape, rate)
        lbeta = std::lgamma(a1) + std::lgamma(b1) - std::lgamma(a1 + b1), lg2 = std::lgamma(a2);
    }
    for (int j = 0; j < K; ++j) {
        const double p = (j * 2.0 + 1.0) / (2.0 * K);
        // cdf(x) - p with the reference's grouping: p0 * F1 + (1.0 - p0) * F2, then minus p
        auto g = [&](double x) {
            const double F1 = kind == 2 ? gamma_cdf_value(x, a1, b1, lg1) : beta_cdf_value(x, a1, b1, lbeta);
            const double F2 = gamma_cdf_value(x, a2, b2, lg2);
            return (p0 * F1 + (1.0 - p0) * F2) - p;
        };
        int status = 0;
        const double x = brentq_like_scipy(g, 0.0001, 99.0, 2e-12, 8.881784197001252e-16, 100, &status);
        if (status == 2) return false;
        out[j] = status == 1 ? p * 10 : x;
    }
    return true;
}

}  // namespace crab
