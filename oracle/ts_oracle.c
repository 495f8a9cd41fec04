/*
 * ts_oracle.c -- CPU restatement of TeraStructure's SNPSamplingE hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see ts_oracle.h): the checker for the HIP path
 * and the "port" CPU baseline of bench.py.  Never linked into the product.
 *
 * Parity status: pinned statistically against the reference's fixtures;
 * bit-level trajectories of the real binary are PARITY UNPINNED (GSL absent,
 * see header).  Arithmetic is fp64 in the reference's operation order
 * (compile with -ffp-contract=off).
 *
 * Layout note: the reference stores y[n][l] (one heap row per individual,
 * src/snp.cc:142).  Here genotypes are SNP-major bytes y[l][n]; bit 7 of a
 * byte marks a held-out (validation map) entry, low 2 bits are 0/1/2/3.
 * That is storage, not semantics.
 */
#include "ts_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ===================== GSL restatements ================================== */

#define MT_N 624
#define MT_M 397

/* gsl rng/mt.c mt_set: s==0 -> 4357; mt[i] = 1812433253*(mt[i-1]^(mt[i-1]>>30))+i */
void orc_rng_seed(orc_rng *r, unsigned long seed) {
  if (seed == 0) seed = 4357;
  r->mt[0] = (uint32_t)(seed & 0xffffffffUL);
  for (int i = 1; i < MT_N; i++)
    r->mt[i] = (uint32_t)(1812433253UL * (r->mt[i - 1] ^ (r->mt[i - 1] >> 30)) + (unsigned long)i);
  r->mti = MT_N;
}

uint32_t orc_rng_get(orc_rng *r) {
  static const uint32_t UPPER = 0x80000000U, LOWER = 0x7fffffffU;
  uint32_t *mt = r->mt;
  if (r->mti >= MT_N) {
    int kk;
    for (kk = 0; kk < MT_N - MT_M; kk++) {
      uint32_t y = (mt[kk] & UPPER) | (mt[kk + 1] & LOWER);
      mt[kk] = mt[kk + MT_M] ^ (y >> 1) ^ ((y & 1) ? 0x9908b0dfU : 0);
    }
    for (; kk < MT_N - 1; kk++) {
      uint32_t y = (mt[kk] & UPPER) | (mt[kk + 1] & LOWER);
      mt[kk] = mt[kk + (MT_M - MT_N)] ^ (y >> 1) ^ ((y & 1) ? 0x9908b0dfU : 0);
    }
    {
      uint32_t y = (mt[MT_N - 1] & UPPER) | (mt[0] & LOWER);
      mt[MT_N - 1] = mt[MT_M - 1] ^ (y >> 1) ^ ((y & 1) ? 0x9908b0dfU : 0);
    }
    r->mti = 0;
  }
  uint32_t k = mt[r->mti++];
  k ^= (k >> 11);
  k ^= (k << 7) & 0x9d2c5680U;
  k ^= (k << 15) & 0xefc60000U;
  k ^= (k >> 18);
  return k;
}

/* gsl rng/rng.c gsl_rng_uniform_int (min 0, max 0xffffffff for mt19937) */
uint32_t orc_rng_uniform_int(orc_rng *r, uint32_t n) {
  uint32_t scale = 0xffffffffU / n;
  uint32_t k;
  do {
    k = orc_rng_get(r) / scale;
  } while (k >= n);
  return k;
}

double orc_rng_uniform(orc_rng *r) { return orc_rng_get(r) / 4294967296.0; }

double orc_rng_uniform_pos(orc_rng *r) {
  double x;
  do {
    x = orc_rng_uniform(r);
  } while (x == 0);
  return x;
}

/* gsl randist/gauss.c gsl_ran_gaussian (polar Box-Muller), sigma = 1 */
static double orc_ran_ugaussian(orc_rng *r) {
  double x, y, r2;
  do {
    x = -1 + 2 * orc_rng_uniform_pos(r);
    y = -1 + 2 * orc_rng_uniform_pos(r);
    r2 = x * x + y * y;
  } while (r2 > 1.0 || r2 == 0);
  return y * sqrt(-2.0 * log(r2) / r2);
}

/* gsl randist/gamma.c gsl_ran_gamma (Marsaglia & Tsang 2000) */
double orc_ran_gamma(orc_rng *r, double a, double b) {
  if (a < 1) {
    double u = orc_rng_uniform_pos(r);
    return orc_ran_gamma(r, 1.0 + a, b) * pow(u, 1.0 / a);
  }
  {
    double x, v, u;
    double d = a - 1.0 / 3.0;
    double c = (1.0 / 3.0) / sqrt(d);
    while (1) {
      do {
        x = orc_ran_ugaussian(r);
        v = 1.0 + c * x;
      } while (v <= 0);
      v = v * v * v;
      u = orc_rng_uniform_pos(r);
      if (u < 1 - 0.0331 * x * x * x * x) break;
      if (log(u) < 0.5 * x * x + d * (1 - v + log(v))) break;
    }
    return b * d * v;
  }
}

/* psi(x), x > 0: psi(x) = psi(x+m) - sum 1/(x+i); asymptotic series at x >= 10
 * with Bernoulli terms B2n/(2n x^2n), n = 1..7 (next term < 5e-17 at x = 10). */
double orc_digamma(double x) {
  double r = 0.0;
  while (x < 10.0) {
    r -= 1.0 / x;
    x += 1.0;
  }
  double f = 1.0 / (x * x);
  double t = f * (-1.0 / 12.0 +
                  f * (1.0 / 120.0 +
                       f * (-1.0 / 252.0 +
                            f * (1.0 / 240.0 +
                                 f * (-1.0 / 132.0 + f * (691.0 / 32760.0 + f * (-1.0 / 12.0)))))));
  return r + log(x) - 0.5 / x + t;
}

/* ===================== engine state ====================================== */

struct orc_state {
  orc_config c;
  uint8_t *y; /* [l][n]; low 2 bits genotype (3 = missing), bit 7 = held out */
  double *gamma, *elogtheta, *etheta; /* [n][k] */
  double *lambda, *elogbeta;          /* [l][k][2] */
  double *ebeta;                      /* [l][k] */
  double *phimom, *phidad;            /* [n][k]  (PhiRunnerE::_phimom/_phidad, hh:87-88) */
  double *rho_indiv;
  uint32_t *c_indiv;
  uint32_t *held_locs; /* sorted ascending */
  uint32_t n_held_locs, cap_held_locs;
  /* worker state of the single PhiRunnerE (hh:78-84) */
  int w_first;
  uint32_t w_loc;
  int w_prev_hol;
  /* driver state */
  uint32_t iter;
  int hol_mode;
};

void orc_default_config(orc_config *c, uint32_t n, uint32_t l, uint32_t k) {
  c->n = n;
  c->l = l;
  c->k = k;
  c->alpha = (double)1.0 / k;
  c->eta0 = 1.0;
  c->eta1 = 1.0;
  c->nodetau0 = 1 + 1;
  c->nodekappa = 0.5;
  c->meanchangethresh = 0.001;
  c->online_iterations = 10;
  c->gamma_scale = l;
  c->nthreads = 1;
}

static void *xcalloc(size_t n, size_t sz) {
  void *p = calloc(n ? n : 1, sz);
  if (!p) {
    fprintf(stderr, "ts_oracle: out of memory (%zu x %zu)\n", n, sz);
    abort();
  }
  return p;
}

orc_state *orc_create(const orc_config *c) {
  orc_state *s = (orc_state *)xcalloc(1, sizeof(*s));
  s->c = *c;
  if (s->c.nthreads < 1) s->c.nthreads = 1;
  size_t n = c->n, l = c->l, k = c->k;
  s->y = (uint8_t *)xcalloc(n * l, 1);
  s->gamma = (double *)xcalloc(n * k, 8);
  s->elogtheta = (double *)xcalloc(n * k, 8);
  s->etheta = (double *)xcalloc(n * k, 8);
  s->lambda = (double *)xcalloc(l * k * 2, 8);
  s->elogbeta = (double *)xcalloc(l * k * 2, 8);
  s->ebeta = (double *)xcalloc(l * k, 8);
  s->phimom = (double *)xcalloc(n * k, 8);
  s->phidad = (double *)xcalloc(n * k, 8);
  s->rho_indiv = (double *)xcalloc(n, 8);
  s->c_indiv = (uint32_t *)xcalloc(n, 4);
  s->w_first = 1;
  orc_init_lambda(s);
  return s;
}

void orc_destroy(orc_state *s) {
  if (!s) return;
  free(s->y);
  free(s->gamma);
  free(s->elogtheta);
  free(s->etheta);
  free(s->lambda);
  free(s->elogbeta);
  free(s->ebeta);
  free(s->phimom);
  free(s->phidad);
  free(s->rho_indiv);
  free(s->c_indiv);
  free(s->held_locs);
  free(s);
}

/* src/snp.cc:195-228: 2 bits per individual, LSB first;
 * code%4: 1 -> missing(3), 3 -> 2, 2 -> 1, 0 -> 0 */
uint64_t orc_load_bed_payload(orc_state *s, const uint8_t *payload, uint64_t bytes_per_snp,
                              uint32_t first_loc, uint32_t n_locs) {
  static const uint8_t dec[4] = {0, 3, 1, 2};
  uint64_t missing = 0;
  size_t n = s->c.n;
  for (uint32_t j = 0; j < n_locs; ++j) {
    const uint8_t *col = payload + (uint64_t)j * bytes_per_snp;
    uint8_t *out = s->y + (size_t)(first_loc + j) * n;
    for (size_t i = 0; i < n; ++i) {
      uint8_t code = (col[i >> 2] >> (2 * (i & 3))) & 3;
      out[i] = dec[code];
      missing += (dec[code] == 3);
    }
  }
  return missing;
}

int orc_read_bed_file(orc_state *s, const char *bed_path) {
  FILE *f = fopen(bed_path, "rb");
  if (!f) return -1;
  unsigned char magic[3];
  if (fread(magic, 1, 3, f) != 3 || magic[0] != 108 || magic[1] != 27 || magic[2] != 1) {
    fclose(f);
    return -1;
  }
  uint64_t nb = s->c.n / 4 + (s->c.n % 4 != 0); /* src/snp.cc:146-148 */
  uint8_t *buf = (uint8_t *)xcalloc(nb, 1);
  uint32_t loc = 0;
  while (loc < s->c.l && fread(buf, 1, nb, f) == nb) {
    orc_load_bed_payload(s, buf, nb, loc, 1);
    loc++;
  }
  free(buf);
  fclose(f);
  return loc == s->c.l ? 0 : -1;
}

uint8_t orc_y(const orc_state *s, uint32_t n, uint32_t loc) {
  return s->y[(size_t)loc * s->c.n + n] & 3;
}

static void held_loc_insert(orc_state *s, uint32_t loc) {
  uint32_t lo = 0, hi = s->n_held_locs;
  while (lo < hi) {
    uint32_t mid = (lo + hi) / 2;
    if (s->held_locs[mid] < loc)
      lo = mid + 1;
    else
      hi = mid;
  }
  if (lo < s->n_held_locs && s->held_locs[lo] == loc) return;
  if (s->n_held_locs == s->cap_held_locs) {
    s->cap_held_locs = s->cap_held_locs ? 2 * s->cap_held_locs : 64;
    s->held_locs = (uint32_t *)realloc(s->held_locs, s->cap_held_locs * sizeof(uint32_t));
  }
  memmove(s->held_locs + lo + 1, s->held_locs + lo, (s->n_held_locs - lo) * sizeof(uint32_t));
  s->held_locs[lo] = loc;
  s->n_held_locs++;
}

void orc_set_heldout(orc_state *s, uint32_t loc, const uint32_t *indivs, uint32_t count) {
  for (uint32_t i = 0; i < count; ++i) s->y[(size_t)loc * s->c.n + indivs[i]] |= 0x80;
  if (count) held_loc_insert(s, loc);
}

/* kv_ok: not in test map, not in validation map, not missing (hh:389-408).
 * The test map is never populated on the live path (-use-test-set is broken,
 * SURVEY section 5), so one flag bit covers both maps. */
static inline int kv_ok_byte(uint8_t b) { return b < 3; }

int orc_kv_ok(const orc_state *s, uint32_t indiv, uint32_t loc) {
  return kv_ok_byte(s->y[(size_t)loc * s->c.n + indiv]);
}

/* PopLib::set_dir_exp (src/lib.hh:19-35) + theta (estimate_all_theta cc:595-609) for one row */
static void row_theta(orc_state *s, uint32_t n) {
  uint32_t K = s->c.k;
  const double *g = s->gamma + (size_t)n * K;
  double sum = .0;
  for (uint32_t k = 0; k < K; ++k) sum += g[k];
  double psi_sum = orc_digamma(sum);
  for (uint32_t k = 0; k < K; ++k) {
    s->etheta[(size_t)n * K + k] = g[k] / sum;
    s->elogtheta[(size_t)n * K + k] = orc_digamma(g[k]) - psi_sum;
  }
}

void orc_init_gamma(orc_state *s, orc_rng *r) {
  uint32_t K = s->c.k;
  for (uint32_t i = 0; i < s->c.n; ++i)
    for (uint32_t j = 0; j < K; ++j) {
      double v = (K < 100) ? 1.0 : (double)100.0 / K;
      s->gamma[(size_t)i * K + j] = orc_ran_gamma(r, 100 * v, 0.01);
    }
  for (uint32_t i = 0; i < s->c.n; ++i) row_theta(s, i);
}

void orc_set_gamma(orc_state *s, const double *gamma) {
  memcpy(s->gamma, gamma, (size_t)s->c.n * s->c.k * sizeof(double));
  for (uint32_t i = 0; i < s->c.n; ++i) row_theta(s, i);
}

/* estimate_beta(loc) src/snpsamplinge.cc:279-296 */
void orc_estimate_beta(orc_state *s, uint32_t loc) {
  uint32_t K = s->c.k;
  const double *ld = s->lambda + (size_t)loc * K * 2;
  double *eb = s->elogbeta + (size_t)loc * K * 2;
  for (uint32_t k = 0; k < K; ++k) {
    double sum = .0;
    for (uint32_t t = 0; t < 2; ++t) sum += ld[2 * k + t];
    s->ebeta[(size_t)loc * K + k] = ld[2 * k] / sum;
    double psi_sum = orc_digamma(sum);
    eb[2 * k + 0] = orc_digamma(ld[2 * k + 0]) - psi_sum;
    eb[2 * k + 1] = orc_digamma(ld[2 * k + 1]) - psi_sum;
  }
}

void orc_init_lambda(orc_state *s) {
  uint32_t K = s->c.k;
  for (uint32_t l = 0; l < s->c.l; ++l) {
    for (uint32_t k = 0; k < K; ++k) {
      s->lambda[((size_t)l * K + k) * 2 + 0] = s->c.eta0;
      s->lambda[((size_t)l * K + k) * 2 + 1] = s->c.eta1;
    }
    /* PopLib::set_dir_exp(D3) src/lib.hh:37-57 (does not touch Ebeta; the
     * initial Ebeta is filled by estimate_beta in snp_likelihood(first)) */
    double *eb = s->elogbeta + (size_t)l * K * 2;
    for (uint32_t k = 0; k < K; ++k) {
      double sum = s->c.eta0 + s->c.eta1;
      double psi_sum = orc_digamma(sum);
      eb[2 * k + 0] = orc_digamma(s->c.eta0) - psi_sum;
      eb[2 * k + 1] = orc_digamma(s->c.eta1) - psi_sum;
    }
  }
}

void orc_set_lambda(orc_state *s, uint32_t loc, const double *lam) {
  memcpy(s->lambda + (size_t)loc * s->c.k * 2, lam, (size_t)s->c.k * 2 * sizeof(double));
  orc_estimate_beta(s, loc);
}

uint32_t orc_set_validation_sample(orc_state *s, orc_rng *r) {
  uint32_t n = s->c.n, l = s->c.l;
  uint32_t per_loc_h = n < 2000 ? (n / 10) : (n / 100);
  double validation_ratio = 0.005; /* src/env.hh:211 */
  uint32_t nlocs = (uint32_t)(l * validation_ratio);
  uint8_t *seen = (uint8_t *)xcalloc(l, 1);
  uint32_t nseen = 0;
  do {
    uint32_t loc = orc_rng_uniform_int(r, l);
    if (seen[loc]) continue;
    seen[loc] = 1;
    nseen++;
    uint32_t c = 0;
    while (c < per_loc_h) {
      uint32_t indiv = orc_rng_uniform_int(r, n);
      if (orc_kv_ok(s, indiv, loc)) {
        orc_set_heldout(s, loc, &indiv, 1);
        c++;
      }
    }
  } while (nseen < nlocs);
  free(seen);
  return nseen;
}

/* ===================== the hot path ====================================== */

/* D1Array<double>::logsum + lognormalize, src/matrix.hh:271-293 */
static inline void lognormalize(double *v, uint32_t K) {
  double r = v[0];
  if (K > 1) {
    for (uint32_t i = 1; i < K; ++i)
      if (v[i] < r)
        r = r + log(1 + exp(v[i] - r));
      else
        r = v[i] + log(1 + exp(r - v[i]));
  }
  for (uint32_t i = 0; i < K; ++i) v[i] = exp(v[i] - r);
}

/* PhiRunnerE::process (hh:416-431) restricted to [begin,end):
 * update_phimom/update_phidad (hh:276-300) then update_lambda_t (cc:742-759,
 * k-outer, n-inner, sequential sums). */
void orc_pass_partial(orc_state *s, uint32_t loc, uint32_t begin, uint32_t end, double *ldt) {
  uint32_t K = s->c.k;
  const uint8_t *col = s->y + (size_t)loc * s->c.n;
  const double *elogbeta = s->elogbeta + (size_t)loc * K * 2;
  for (uint32_t n = begin; n < end; ++n) {
    if (!kv_ok_byte(col[n])) continue;
    double *pm = s->phimom + (size_t)n * K;
    double *pd = s->phidad + (size_t)n * K;
    const double *et = s->elogtheta + (size_t)n * K;
    for (uint32_t k = 0; k < K; ++k) pm[k] = et[k] + elogbeta[2 * k + 0];
    lognormalize(pm, K);
    for (uint32_t k = 0; k < K; ++k) pd[k] = et[k] + elogbeta[2 * k + 1];
    lognormalize(pd, K);
  }
  for (uint32_t k = 0; k < K; ++k) {
    double a0 = 0.0, a1 = 0.0;
    for (uint32_t n = begin; n < end; ++n) {
      if (!kv_ok_byte(col[n])) continue;
      uint8_t y = col[n];
      a0 += s->phimom[(size_t)n * K + k] * y;
      a1 += s->phidad[(size_t)n * K + k] * (2 - y);
    }
    ldt[2 * k + 0] = a0;
    ldt[2 * k + 1] = a1;
  }
}

/* split_all_indivs (cc:298-318): chunk t of T */
static void chunk_bounds(uint32_t n, int T, int t, uint32_t *b, uint32_t *e) {
  uint32_t chunk = (uint32_t)(((double)n) / T);
  *b = (uint32_t)t * chunk;
  *e = (t == T - 1) ? n : (uint32_t)(t + 1) * chunk;
  if (*b > n) *b = n;
  if (*e > n) *e = n;
}

/* one inner pass over all chunks; main-thread sum in chunk order (cc:337-352) */
static void pass_all(orc_state *s, uint32_t loc, double *ldt) {
  uint32_t K = s->c.k;
  int T = s->c.nthreads;
  if (T == 1) {
    orc_pass_partial(s, loc, 0, s->c.n, ldt);
    return;
  }
  double *part = (double *)xcalloc((size_t)T * 2 * K, 8);
#pragma omp parallel for schedule(static, 1) num_threads(T)
  for (int t = 0; t < T; ++t) {
    uint32_t b, e;
    chunk_bounds(s->c.n, T, t, &b, &e);
    orc_pass_partial(s, loc, b, e, part + (size_t)t * 2 * K);
  }
  for (uint32_t j = 0; j < 2 * K; ++j) ldt[j] = 0.0;
  for (int t = 0; t < T; ++t)
    for (uint32_t j = 0; j < 2 * K; ++j) ldt[j] += part[(size_t)t * 2 * K + j];
  free(part);
}

/* update_lambda (cc:267-277) + estimate_beta (cc:279-296) + sub/abs_mean
 * (matrix.hh:873-893); returns _v.abs_mean() */
double orc_epilogue(orc_state *s, uint32_t loc, const double *ldt) {
  uint32_t K = s->c.k;
  double *ld = s->lambda + (size_t)loc * K * 2;
  double old[2 * 4096];
  double *lo = (2 * K <= 2 * 4096) ? old : (double *)xcalloc(2 * K, 8);
  memcpy(lo, ld, 2 * K * sizeof(double));
  for (uint32_t k = 0; k < K; ++k) {
    ld[2 * k + 0] = s->c.eta0 + ldt[2 * k + 0];
    ld[2 * k + 1] = s->c.eta1 + ldt[2 * k + 1];
  }
  orc_estimate_beta(s, loc);
  double sum = .0;
  for (uint32_t k = 0; k < K; ++k)
    for (uint32_t t = 0; t < 2; ++t) sum += fabs(ld[2 * k + t] - lo[2 * k + t]);
  if (lo != old) free(lo);
  return sum / (K * 2);
}

/* update_gamma (cc:695-719) + update_rho_indiv (cc:688-693) + estimate_theta
 * (cc:721-740) over [begin,end) for location loc with the phi currently stored */
static void gamma_step_range(orc_state *s, uint32_t loc, uint32_t begin, uint32_t end) {
  uint32_t K = s->c.k;
  const uint8_t *col = s->y + (size_t)loc * s->c.n;
  double gamma_scale = s->c.gamma_scale;
  for (uint32_t n = begin; n < end; ++n) {
    if (!kv_ok_byte(col[n])) continue;
    s->rho_indiv[n] = pow(s->c.nodetau0 + s->c_indiv[n], -1 * s->c.nodekappa);
    s->c_indiv[n]++;
    uint8_t y = col[n];
    double *gd = s->gamma + (size_t)n * K;
    const double *pm = s->phimom + (size_t)n * K;
    const double *pd = s->phidad + (size_t)n * K;
    for (uint32_t k = 0; k < K; ++k)
      gd[k] += s->rho_indiv[n] * (s->c.alpha + (gamma_scale * (y * pm[k] + (2 - y) * pd[k])) - gd[k]);
  }
  for (uint32_t n = begin; n < end; ++n) row_theta(s, n);
}

void orc_gamma_step(orc_state *s, uint32_t loc) {
  int T = s->c.nthreads;
  if (T == 1) {
    gamma_step_range(s, loc, 0, s->c.n);
    return;
  }
#pragma omp parallel for schedule(static, 1) num_threads(T)
  for (int t = 0; t < T; ++t) {
    uint32_t b, e;
    chunk_bounds(s->c.n, T, t, &b, &e);
    gamma_step_range(s, loc, b, e);
  }
}

/* optimize_lambda(loc) (cc:320-366) as seen by one worker (do_work cc:649-686):
 * a new SNP first applies the pending gamma/theta step of the previous SNP
 * unless that SNP ran in hol mode, using the phi of its last pass. */
uint32_t orc_snp_update(orc_state *s, uint32_t loc, int hol_mode) {
  uint32_t K = s->c.k;
  if (!s->w_first && !s->w_prev_hol) orc_gamma_step(s, s->w_loc);
  /* reset() hh:266-274 */
  s->w_loc = loc;
  s->w_prev_hol = hol_mode;
  s->w_first = 0;

  double ldt[2 * 4096];
  double *lt = (2 * K <= 2 * 4096) ? ldt : (double *)xcalloc(2 * K, 8);
  uint32_t x = 0;
  do {
    pass_all(s, loc, lt);
    double v = orc_epilogue(s, loc, lt);
    x++;
    if (v < s->c.meanchangethresh) break;
  } while (x < s->c.online_iterations);
  if (lt != ldt) free(lt);
  return x;
}

/* snp_likelihood, second half (hh:336-360) */
double orc_heldout_loglik(const orc_state *s, uint32_t loc, uint32_t *count) {
  uint32_t K = s->c.k;
  const uint8_t *col = s->y + (size_t)loc * s->c.n;
  static const double fact[3] = {1.0, 1.0, 2.0}; /* gsl_sf_fact(0..2) */
  double lsum = .0;
  uint32_t c = 0;
  for (uint32_t n = 0; n < s->c.n; ++n) {
    if (!(col[n] & 0x80)) continue;
    uint8_t x = col[n] & 3;
    double q = .0;
    double v = fact[2] / (fact[x] * fact[2 - x]);
    for (uint32_t k = 0; k < K; ++k) q += s->ebeta[(size_t)loc * K + k] * s->etheta[(size_t)n * K + k];
    double sum = v * pow(q, x) * pow(1 - q, 2 - x);
    if (sum < 1e-30) sum = 1e-30;
    lsum += log(sum);
    c++;
  }
  if (count) *count = c;
  return lsum;
}

const double *orc_gamma(const orc_state *s) { return s->gamma; }
const double *orc_elogtheta(const orc_state *s) { return s->elogtheta; }
const double *orc_etheta(const orc_state *s) { return s->etheta; }
const double *orc_lambda(const orc_state *s) { return s->lambda; }
const double *orc_elogbeta(const orc_state *s) { return s->elogbeta; }
const double *orc_ebeta(const orc_state *s) { return s->ebeta; }
const uint32_t *orc_c_indiv(const orc_state *s) { return s->c_indiv; }
uint32_t orc_n_heldout_locs(const orc_state *s) { return s->n_held_locs; }

uint32_t orc_heldout_locs(const orc_state *s, uint32_t *out, uint32_t cap) {
  uint32_t m = s->n_held_locs < cap ? s->n_held_locs : cap;
  memcpy(out, s->held_locs, m * sizeof(uint32_t));
  return s->n_held_locs;
}

uint32_t orc_heldout_indivs(const orc_state *s, uint32_t loc, uint32_t *out, uint32_t cap) {
  const uint8_t *col = s->y + (size_t)loc * s->c.n;
  uint32_t c = 0;
  for (uint32_t n = 0; n < s->c.n; ++n)
    if (col[n] & 0x80) {
      if (c < cap) out[c] = n;
      c++;
    }
  return c;
}

/* ===================== whole-program driver ============================== */

typedef struct {
  double prev_h, max_h;
  uint32_t nh;
} stop_state;

/* compute_likelihood(first, validation=true) src/snpsamplinge.cc:461-544.
 * Returns 1 when the stop rule fires. */
static int compute_likelihood(orc_state *s, orc_run_params *p, stop_state *st, int first) {
  s->hol_mode = 1;
  uint32_t k = 0;
  double sum = .0;
  for (uint32_t i = 0; i < s->n_held_locs; ++i) {
    uint32_t loc = s->held_locs[i];
    if (first)
      orc_estimate_beta(s, loc);
    else {
      orc_snp_update(s, loc, s->hol_mode);
      s->iter++;
    }
    uint32_t c = 0;
    sum += orc_heldout_loglik(s, loc, &c);
    k += c;
  }
  double a = sum / k;
  if (p->lines && p->n_lines < p->lines_cap) {
    p->lines[p->n_lines].iter = s->iter;
    p->lines[p->n_lines].mean_ll = a;
    p->lines[p->n_lines].count = k;
  }
  p->n_lines++;

  int stop = 0;
  if (s->iter > 2000) {
    if (a > st->prev_h && st->prev_h != 0 && fabs((a - st->prev_h) / st->prev_h) < p->stop_threshold)
      stop = 1;
    else if (a < st->prev_h)
      st->nh++;
    else if (a > st->prev_h)
      st->nh = 0;
    if (a > st->max_h) st->max_h = a;
    if (st->nh > 3) stop = 1;
  }
  st->prev_h = a;
  s->hol_mode = 0;
  return stop; /* use_validation_stop is always true, src/env.hh:238 */
}

int orc_run(orc_state *s, orc_run_params *p) {
  orc_rng r;
  orc_rng_seed(&r, 0); /* gsl_rng_alloc seeds with gsl_rng_default_seed = 0 */
  if (p->seed) orc_rng_seed(&r, p->seed);
  stop_state st = {-2147483647, -2147483647, 0};
  p->n_lines = 0;
  p->stopped = 0;
  s->iter = 0;
  s->hol_mode = 0;
  s->w_first = 1;

  orc_set_validation_sample(s, &r);
  orc_init_gamma(s, &r);
  orc_init_lambda(s);
  compute_likelihood(s, p, &st, 1);

  while (1) {
    uint32_t loc = orc_rng_uniform_int(&r, s->c.l);
    orc_snp_update(s, loc, s->hol_mode);
    s->iter++;
    if (s->iter % p->reportfreq == 0) {
      if (compute_likelihood(s, p, &st, 0)) {
        p->stopped = 1;
        break;
      }
    }
    if (p->max_iter && s->iter >= p->max_iter) break;
  }
  p->final_iter = s->iter;
  return 0;
}

void orc_compute_all_lambda(orc_state *s) {
  s->c.online_iterations = 100;
  s->hol_mode = 0;
  for (uint32_t loc = 0; loc < s->c.l; ++loc) {
    orc_snp_update(s, loc, 0);
    s->iter++;
  }
  /* estimate_all_beta cc:611-625 is a no-op here: estimate_beta(loc) already
   * ran after the last update_lambda of every loc. */
}
