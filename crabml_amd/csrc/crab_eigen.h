// Host-side spectral decomposition of reversible rate matrices, and a small worker pool.
// Product code (part of libcrabml_amd.so); independent of oracle/.
#pragma once
#include <cstddef>
#include <functional>

namespace crab {

constexpr int NS = 64;                 // padded state count on the device
constexpr int EIG_UT = 0;              // offsets (in doubles) inside one packed eigensystem
constexpr int EIG_UINV = NS * NS;
constexpr int EIG_LAM = 2 * NS * NS;
constexpr int EIG_STRIDE = 2 * NS * NS + NS;

// Replaces CachedQMatrix::new (rust/src/matrix.rs:31-75).
//   S = Pi^1/2 Q Pi^-1/2 (only the lower triangle is used, as eigh(UPLO::Lower) does), S = V diag(lam) V^T,
//   U = Pi^-1/2 V and, instead of a numerical inverse, U^-1 = V^T Pi^1/2 (exact for orthogonal V).
// Output is packed for the device, zero-padded to 64:
//   out[EIG_UT   + k*64 + i] = U[i][k]
//   out[EIG_UINV + k*64 + j] = U^-1[k][j]
//   out[EIG_LAM  + k]        = lam[k]
// Returns false if the QL iteration fails to converge or pi has a non-positive entry.
bool spectral_decompose(int n, const double *Q, const double *pi, double *out);

// Plain (unpacked) variant used by crab_matrix_exponential's host checks and tests: U, lam, Uinv are n x n / n.
bool spectral_decompose_plain(int n, const double *Q, const double *pi, double *U, double *lam, double *Uinv);

// Symmetric eigen-solver: Householder tridiagonalisation + implicit-shift QL.
// a: n x n row-major symmetric (lower triangle read), overwritten with eigenvectors in columns; d: eigenvalues.
bool symmetric_eigen(int n, double *a, double *d);

// Runs fn(i) for i in [0, n) on a persistent pool (CRAB_HOST_THREADS, default min(16, cores)).
void parallel_for(int n, const std::function<void(int)> &fn);
int host_threads();

}  // namespace crab
