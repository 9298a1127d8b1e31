// Host-side spectral decomposition (see crab_eigen.h) and the worker pool that runs one
// decomposition per site class / per branch in parallel.
#include "crab_eigen.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace crab {

// ---------------------------------------------------------------------------------------------
// Householder reduction to tridiagonal form followed by QL with implicit shifts.
// a is n x n row-major; on exit column k of a is the eigenvector of d[k].
// ---------------------------------------------------------------------------------------------
static void tridiagonalise(int n, double *a, double *d, double *e)
{
    // Householder reduction A = Q T Q^T on full symmetric storage, every inner loop over a contiguous row
    // (the textbook lower-triangle form walks columns; at n = 61 that, not the QL iteration, dominated the
    // decomposition).  On exit a = Q, d = diag(T), e[i] = T[i][i-1] (e[0] = 0).
    thread_local std::vector<double> V;
    V.assign((size_t)n * n, 0.0);
    std::vector<double> beta(n, 0.0), p(n), w(n), t(n);
    for (int k = 0; k + 2 < n; ++k) {
        double *rowk = a + (size_t)k * n;
        d[k] = rowk[k];
        const int m = n - k - 1;
        const double *x = rowk + k + 1;
        // the reflector is invariant under scaling of x: work on x / max|x| so that columns of any magnitude
        // (a site class with omega ~ 1e-100 has them) neither underflow nor overflow in the squares
        double scale = 0.0;
        for (int i = 0; i < m; ++i) scale = std::max(scale, std::fabs(x[i]));
        if (scale == 0.0) {
            e[k + 1] = 0.0;
            continue;
        }
        double *v = &V[(size_t)k * n];
        double sigma = 0.0;
        for (int i = 0; i < m; ++i) {
            v[i] = x[i] / scale;
            sigma += v[i] * v[i];
        }
        const double x0 = v[0], norm = std::sqrt(sigma);
        const double alpha = x0 >= 0.0 ? -norm : norm;
        v[0] -= alpha;
        const double vv = sigma - x0 * x0 + v[0] * v[0];
        const double bk = 2.0 / vv;
        beta[k] = bk;
        double *sub = a + (size_t)(k + 1) * n + (k + 1);
        // p = beta A v as a sum of rows (A is symmetric): a row-by-row dot product is a scalar chain of dependent
        // multiply-adds -- a third of the reduction's arithmetic at a quarter of a flop per cycle; this form vectorises
        for (int j = 0; j < m; ++j) p[j] = 0.0;
        int i4 = 0;
        for (; i4 + 4 <= m; i4 += 4) {
            const double *r0 = sub + (size_t)i4 * n, *r1 = r0 + n, *r2 = r1 + n, *r3 = r2 + n;
            const double v0 = v[i4], v1 = v[i4 + 1], v2 = v[i4 + 2], v3 = v[i4 + 3];
            for (int j = 0; j < m; ++j) p[j] += (v0 * r0[j] + v1 * r1[j]) + (v2 * r2[j] + v3 * r3[j]);
        }
        for (; i4 < m; ++i4) {
            const double *row = sub + (size_t)i4 * n;
            const double vi = v[i4];
            for (int j = 0; j < m; ++j) p[j] += vi * row[j];
        }
        for (int j = 0; j < m; ++j) p[j] *= bk;
        double pv = 0.0;
        for (int i = 0; i < m; ++i) pv += p[i] * v[i];
        const double c = 0.5 * bk * pv;
        for (int i = 0; i < m; ++i) w[i] = p[i] - c * v[i];
        for (int i = 0; i < m; ++i) {
            double *row = sub + (size_t)i * n;
            const double vi = v[i], wi = w[i];
            for (int j = 0; j < m; ++j) row[j] -= vi * w[j] + wi * v[j];
        }
        e[k + 1] = alpha * scale;
    }
    e[0] = 0.0;
    if (n >= 2) {
        d[n - 2] = a[(size_t)(n - 2) * n + (n - 2)];
        e[n - 1] = a[(size_t)(n - 1) * n + (n - 2)];
    }
    d[n - 1] = a[(size_t)(n - 1) * n + (n - 1)];
    // Q = H_0 H_1 ... H_{n-3}, accumulated backwards so that step k only touches the trailing block
    for (size_t i = 0; i < (size_t)n * n; ++i) a[i] = 0.0;
    for (int i = 0; i < n; ++i) a[(size_t)i * n + i] = 1.0;
    for (int k = n - 3; k >= 0; --k) {
        if (beta[k] == 0.0) continue;
        const int m = n - k - 1;
        const double *v = &V[(size_t)k * n];
        double *sub = a + (size_t)(k + 1) * n + (k + 1);
        for (int j = 0; j < m; ++j) t[j] = 0.0;
        for (int i = 0; i < m; ++i) {
            const double *row = sub + (size_t)i * n;
            const double vi = v[i];
            for (int j = 0; j < m; ++j) t[j] += vi * row[j];
        }
        for (int i = 0; i < m; ++i) {
            double *row = sub + (size_t)i * n;
            const double bv = beta[k] * v[i];
            for (int j = 0; j < m; ++j) row[j] -= bv * t[j];
        }
    }
}

// zt holds the eigenvector matrix transposed (row i = vector i), so that a plane rotation of vectors i and i+1 --
// the O(n^3) part of the whole decomposition -- runs over two contiguous rows.
static bool ql_implicit(int n, double *d, double *e, double *zt)
{
    for (int i = 1; i < n; ++i) e[i - 1] = e[i];
    e[n - 1] = 0.0;
    const double eps = 2.220446049250313e-16;
    for (int l = 0; l < n; ++l) {
        int iter = 0, m;
        do {
            for (m = l; m < n - 1; ++m) {
                const double dd = std::fabs(d[m]) + std::fabs(d[m + 1]);
                if (std::fabs(e[m]) <= eps * dd) break;
            }
            if (m != l) {
                if (iter++ == 120) return false;
                double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
                double r = std::hypot(g, 1.0);
                g = d[m] - d[l] + e[l] / (g + std::copysign(r, g));
                double s = 1.0, c = 1.0, p = 0.0;
                int i;
                for (i = m - 1; i >= l; --i) {
                    double f = s * e[i];
                    const double b = c * e[i];
                    r = std::sqrt(f * f + g * g);  // std::hypot costs more than the rotation it guards: use it only off the safe range
                    if (!(r > 1e-140 && r < 1e140)) r = std::hypot(f, g);
                    e[i + 1] = r;
                    if (r == 0.0) {
                        d[i + 1] -= p;
                        e[m] = 0.0;
                        break;
                    }
                    s = f / r;
                    c = g / r;
                    g = d[i + 1] - p;
                    r = (d[i] - g) * s + 2.0 * c * b;
                    d[i + 1] = g + (p = s * r);
                    g = c * r - b;
                    double *__restrict z0 = zt + (size_t)i * n, *__restrict z1 = zt + (size_t)(i + 1) * n;
                    for (int k = 0; k < n; ++k) {
                        const double f1 = z1[k], f0 = z0[k];
                        z1[k] = s * f0 + c * f1;
                        z0[k] = c * f0 - s * f1;
                    }
                }
                if (r == 0.0 && i >= l) continue;
                d[l] -= p;
                e[l] = g;
                e[m] = 0.0;
            }
        } while (m != l);
    }
    return true;
}

// One Newton-Schulz step V <- V (3 I - V^T V) / 2 on the accumulated eigenvectors (columns of row-major v).
// Householder + QL leaves ||V^T V - I|| at a few n eps (2e-15 at n = 61).  The engine takes U^-1 = V^T Pi^1/2
// analytically instead of inverting U like the reference (matrix.rs:66), so that loss of orthogonality goes straight
// into P(t) = U e^{lam t} U^-1 as an absolute error of the same size -- which is what decides the likelihood of a
// site that leans on entries of P of 1e-8 ... 1e-12.  After the step the orthogonality error is at the rounding level
// (5e-16) and P(0) = I to rounding, as with a numerical inverse.  Measured against 60-digit arithmetic
// (tests/golden/make_parity_tail.py): the largest error of a P entry over four test cases drops from 2.3e-15 to 1.1e-15
// (the reference's LAPACK route: 1.7e-15), and the worst site of the lysin fixture moves from 1.7e-8 to 1e-9 relative
// (LAPACK route: 1.0e-9).
// Both products are dot products of contiguous rows, taken 4 x 4 at a time on zero-padded copies (vector extensions, so
// the host compiler emits AVX2 FMAs without reassociation flags): ~30 us at n = 61 against ~0.3 ms for the decomposition.
typedef double v4d_host __attribute__((vector_size(32), aligned(8)));
// C[k][l] = row k of A . row l of B for k, l < np (np a multiple of 4; row stride np; zero padding beyond n);
// upper_only: only blocks with l >= k are computed (C symmetric, the caller mirrors)
static void rows_dot_rows(int np, const double *A, const double *B, double *C, bool upper_only)
{
    for (int k0 = 0; k0 < np; k0 += 4)
        for (int l0 = upper_only ? k0 : 0; l0 < np; l0 += 2) {   // 4 x 2 blocks: 8 accumulators + 6 operands fit 16 registers
            v4d_host acc[4][2];
            for (int x = 0; x < 4; ++x)
                for (int y = 0; y < 2; ++y) acc[x][y] = (v4d_host){0.0, 0.0, 0.0, 0.0};
            for (int i = 0; i < np; i += 4) {
                v4d_host a[4], b[2];
                for (int x = 0; x < 4; ++x) a[x] = *(const v4d_host *)(A + (size_t)(k0 + x) * np + i);
                for (int y = 0; y < 2; ++y) b[y] = *(const v4d_host *)(B + (size_t)(l0 + y) * np + i);
                for (int x = 0; x < 4; ++x)
                    for (int y = 0; y < 2; ++y) acc[x][y] += a[x] * b[y];
            }
            for (int x = 0; x < 4; ++x)
                for (int y = 0; y < 2; ++y) C[(size_t)(k0 + x) * np + l0 + y] = (acc[x][y][0] + acc[x][y][2]) + (acc[x][y][1] + acc[x][y][3]);
        }
}
// zt: eigenvectors in rows (the layout the QL iteration works on); on exit v (n x n row-major) holds them in columns.
static void reorthogonalise(int n, const double *zt, double *v)
{
    const int np = (n + 3) & ~3;
    thread_local std::vector<double> Z, V, E, W;   // per thread: four 32 KB allocations per decomposition were 15 % of this step
    Z.assign((size_t)np * np, 0.0), V.assign((size_t)np * np, 0.0), E.resize((size_t)np * np), W.resize((size_t)np * np);
    for (int k = 0; k < n; ++k)
        for (int i = 0; i < n; ++i) {
            const double x = zt[(size_t)k * n + i];
            Z[(size_t)k * np + i] = x;     // row k = eigenvector k
            V[(size_t)i * np + k] = x;     // column k = eigenvector k
        }
    rows_dot_rows(np, Z.data(), Z.data(), E.data(), true);           // G = V^T V
    for (int k = 0; k < np; ++k)                                     // E = (I - G) / 2, symmetric; zero on the padding
        for (int l = k; l < np; ++l) {
            const double g = (k < n && l < n) ? 0.5 * ((k == l ? 1.0 : 0.0) - E[(size_t)k * np + l]) : 0.0;
            E[(size_t)k * np + l] = g;
            E[(size_t)l * np + k] = g;
        }
    rows_dot_rows(np, V.data(), E.data(), W.data(), false);          // (V E)[i][l] = row i of V . row l of E
    for (int i = 0; i < n; ++i)
        for (int l = 0; l < n; ++l) v[(size_t)i * n + l] = V[(size_t)i * np + l] + W[(size_t)i * np + l];
}

bool symmetric_eigen(int n, double *a, double *d)
{
    thread_local std::vector<double> e, zt;
    e.assign(n, 0.0);
    for (int i = 0; i < n; ++i)  // symmetrise from the lower triangle
        for (int j = i + 1; j < n; ++j) a[(size_t)i * n + j] = a[(size_t)j * n + i];
    tridiagonalise(n, a, d, e.data());
    auto transpose = [&] {
        for (int i = 0; i < n; ++i)
            for (int j = i + 1; j < n; ++j) std::swap(a[(size_t)i * n + j], a[(size_t)j * n + i]);
    };
    transpose();
    const bool ok = ql_implicit(n, d, e.data(), a);
    if (ok) {
        zt.assign(a, a + (size_t)n * n);
        reorthogonalise(n, zt.data(), a);
    } else {
        transpose();
    }
    return ok;
}

bool spectral_decompose_plain(int n, const double *Q, const double *pi, double *U, double *lam, double *Uinv)
{
    std::vector<double> S((size_t)n * n), sq(n), isq(n);
    for (int i = 0; i < n; ++i) {
        if (!(pi[i] > 0.0)) return false;
        sq[i] = std::sqrt(pi[i]);
        isq[i] = 1.0 / sq[i];
    }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) S[(size_t)i * n + j] = sq[i] * Q[(size_t)i * n + j] * isq[j];
    if (!symmetric_eigen(n, S.data(), lam)) return false;
    for (int k = 0; k < n; ++k)
        if (!std::isfinite(lam[k])) return false;
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < n; ++k) {
            const double v = S[(size_t)i * n + k];
            U[(size_t)i * n + k] = isq[i] * v;
            Uinv[(size_t)k * n + i] = v * sq[i];
        }
    return true;
}

bool spectral_decompose(int n, const double *Q, const double *pi, double *out)
{
    // (the same arithmetic as spectral_decompose_plain, written straight into the packed layout from per-thread scratch: the
    // allocations, the zero fill and the strided repacking were a tenth of a decomposition)
    if (n < 1 || n > NS) return false;
    thread_local std::vector<double> S, lam, sq, isq;
    S.resize((size_t)n * n), lam.resize(n), sq.resize(n), isq.resize(n);
    for (int i = 0; i < n; ++i) {
        if (!(pi[i] > 0.0)) return false;
        sq[i] = std::sqrt(pi[i]);
        isq[i] = 1.0 / sq[i];
    }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) S[(size_t)i * n + j] = sq[i] * Q[(size_t)i * n + j] * isq[j];
    if (!symmetric_eigen(n, S.data(), lam.data())) return false;
    for (int k = 0; k < n; ++k)
        if (!std::isfinite(lam[k])) return false;
    for (int i = 0; i < n; ++i) {   // S[i][k]: component i of eigenvector k
        const double *row = S.data() + (size_t)i * n;
        const double a = isq[i], b = sq[i];
        for (int k = 0; k < n; ++k) {
            out[EIG_UT + k * NS + i] = a * row[k];
            out[EIG_UINV + k * NS + i] = row[k] * b;
        }
    }
    for (int k = 0; k < NS; ++k) {   // zero padding: rows n .. NS-1 whole, columns n .. NS-1 of the others
        const int from = k < n ? n : 0;
        std::memset(out + EIG_UT + k * NS + from, 0, sizeof(double) * (NS - from));
        std::memset(out + EIG_UINV + k * NS + from, 0, sizeof(double) * (NS - from));
    }
    std::memcpy(out + EIG_LAM, lam.data(), sizeof(double) * n);
    std::memset(out + EIG_LAM + n, 0, sizeof(double) * (EIG_STRIDE - EIG_LAM - n));
    return true;
}

// ---------------------------------------------------------------------------------------------
// worker pool
// ---------------------------------------------------------------------------------------------
namespace {
// Several engines (one host thread each, e.g. crabml_amd.dist.map_genes) may call parallel_for at once: every call is a
// job in a shared list; pool threads take indices from the oldest job that still has some, and the caller works on its
// own job too, so a job never waits for a thread that is busy elsewhere.
class Pool {
    struct Job {   // lives on its caller's stack; every field is guarded by m_
        const std::function<void(int)> *fn;
        int n, chunk = 1, next = 0, done = 0;
    };

  public:
    Pool()
    {
        int want = 0;
        if (const char *s = std::getenv("CRAB_HOST_THREADS")) want = std::atoi(s);
        if (want <= 0) {
            unsigned hc = std::thread::hardware_concurrency();
            want = (int)std::min<unsigned>(hc ? hc : 1u, 16u);
        }
        n_threads_ = std::max(1, want);
        for (int t = 1; t < n_threads_; ++t) workers_.emplace_back([this] { loop(); });
    }
    ~Pool()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &w : workers_) w.join();
    }
    int size() const { return n_threads_; }

    void run(int n, const std::function<void(int)> &fn)
    {
        if (n <= 0) return;
        if (n == 1 || n_threads_ == 1) {
            for (int i = 0; i < n; ++i) fn(i);
            return;
        }
        Job job;
        job.fn = &fn;
        job.n = n;
        job.chunk = std::max(1, n / (8 * n_threads_));   // 11 decompositions: one each; 700 hashes: blocks of 5
        std::unique_lock<std::mutex> lk(m_);
        jobs_.push_back(&job);
        lk.unlock();
        cv_.notify_all();   // with the mutex released: the woken threads must not queue up behind their waker
        lk.lock();
        drain(&job, lk);
        done_cv_.wait(lk, [&] { return job.done == job.n; });
    }

  private:
    // Run indices of `job` until none is left to hand out.  lk is held on entry and on return.  A job is alive as long
    // as one of its indices has not finished, so it is only touched under the lock, and not again after the increment
    // that completes it.
    void drain(Job *job, std::unique_lock<std::mutex> &lk)
    {
        while (job->next < job->n) {
            const int i0 = job->next;
            const int cnt = std::min(job->chunk, job->n - i0);   // short iterations are handed out in blocks
            job->next += cnt;
            if (job->next == job->n) jobs_.erase(std::find(jobs_.begin(), jobs_.end(), job));   // nothing left to hand out
            const std::function<void(int)> *fn = job->fn;
            lk.unlock();
            for (int i = i0; i < i0 + cnt; ++i) (*fn)(i);
            lk.lock();
            job->done += cnt;
            if (job->done == job->n) {
                done_cv_.notify_all();
                return;
            }
        }
    }
    void loop()
    {
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            cv_.wait(lk, [&] { return stop_ || !jobs_.empty(); });
            if (stop_) return;
            drain(jobs_.front(), lk);
        }
    }
    std::vector<std::thread> workers_;
    std::vector<Job *> jobs_;
    std::mutex m_;
    std::condition_variable cv_, done_cv_;
    int n_threads_ = 1;
    bool stop_ = false;
};

Pool &pool()
{
    static Pool p;
    return p;
}
}  // namespace

void parallel_for(int n, const std::function<void(int)> &fn) { pool().run(n, fn); }
int host_threads() { return pool().size(); }

}  // namespace crab
