/*
 * wright_fused.c -- C restatement of momlevel's global steric inner loop.  TEST INFRASTRUCTURE ONLY
 * (see oracle/__init__.py): a second, independent checker beside oracle/momlevel_numpy.py and the
 * "what a fused, multithreaded CPU implementation achieves" line of bench.py.  The product never
 * links or loads it.
 *
 * Follows src/momlevel/eos/wright.py:6-20 (constants), :44-48 (density, same operator order;
 * compile with -ffp-contract=off so gcc emits no FMA) and src/momlevel/derived.py:435-438
 * (masso = sum(rho * volcello) over z,y,x, NaN terms skipped as xarray's .sum() does).
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC)
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

static const double A0 = 7.057924e-4, A1 = 3.480336e-7, A2 = -1.112733e-7;
static const double B0 = 5.790749e8, B1 = 3.516535e6, B2 = -4.002714e4, B3 = 2.084372e2,
                    B4 = 5.944068e5, B5 = -9.643486e3;
static const double C0 = 1.704853e5, C1 = 7.904722e2, C2 = -7.984422, C3 = 5.140652e-2,
                    C4 = -2.302158e2, C5 = -3.079464;

static inline double wright_density(double T, double S, double p) {
  const double al0 = (A0 + A1 * T) + A2 * S;
  const double p0 = (B0 + B4 * S) + T * ((B1 + T * (B2 + B3 * T)) + B5 * S);
  const double lam = (C0 + C4 * S) + T * ((C1 + T * (C2 + C3 * T)) + C5 * S);
  const double pp0 = p + p0;
  const double I_denom = 1.0 / (lam + al0 * pp0);
  return pp0 * I_denom;
}

/* rho[i] for a (nz, plane) slab with a z-profile pressure: eos/wright.py:23-50 via calc_rho */
void oracle_density_slab(const double *T, const double *S, const double *pz, int64_t nz,
                         int64_t plane, double *rho) {
#pragma omp parallel for schedule(static)
  for (int64_t z = 0; z < nz; ++z) {
    const double p = pz[z];
    for (int64_t i = 0; i < plane; ++i)
      rho[z * plane + i] = wright_density(T[z * plane + i], S[z * plane + i], p);
  }
}

/* masso of one (nz, plane) slab, fused (no rho temporary): one partial per z level, summed in z
 * order, so the result does not depend on the thread count */
double oracle_masso_slab(const double *T, const double *S, const double *vol, const double *pz,
                         int64_t nz, int64_t plane, double *zpartials) {
#pragma omp parallel for schedule(dynamic, 1)
  for (int64_t z = 0; z < nz; ++z) {
    const double p = pz[z];
    double acc = 0.0;
    for (int64_t i = 0; i < plane; ++i) {
      const double term = wright_density(T[z * plane + i], S[z * plane + i], p) * vol[z * plane + i];
      if (term == term) acc += term; /* skipna */
    }
    zpartials[z] = acc;
  }
  double total = 0.0;
  for (int64_t z = 0; z < nz; ++z) total += zpartials[z];
  return total;
}

void oracle_set_threads(int n) {
#ifdef _OPENMP
  extern void omp_set_num_threads(int);
  if (n > 0) omp_set_num_threads(n);
#endif
}

int oracle_num_threads(void) {
  int n = 1;
#ifdef _OPENMP
#pragma omp parallel
  {
#pragma omp master
    n = __builtin_omp_get_num_threads();
  }
#endif
  return n;
}
