/*
 * ts_oracle.h -- CPU restatement of TeraStructure's SNPSamplingE hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under terastructure_amd/ or host/ may
 * include, link or dlopen this.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and there only as the checker / the
 * reported CPU baseline.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - pinned statistically against the reference's own fixtures
 *     (data/test.bed + data/oracle_theta.txt, data/oracle_beta.txt,
 *     data/output_theta.txt), tests/test_oracle_golden.py;
 *   - digamma pinned against scipy.special.digamma, mt19937 against the
 *     published MT19937 known answers (10000th output of seed 5489 and the
 *     GSL seeding recurrence);
 *   - the reference cannot be built here (GSL absent; a stand-in library is
 *     not allowed), so bit-level trajectories of the real binary are
 *     PARITY UNPINNED.  Exactly two GSL behaviours stay unpinned:
 *       1. the stream of gsl_ran_gamma (init_gamma, src/snpsamplinge.cc:233).
 *          Real GSL runs Marsaglia-Tsang on normals from
 *          gsl_ran_gaussian_ziggurat (128-entry tables, extra uniform draws
 *          on the rejection path); orc_ran_gamma runs the same
 *          Marsaglia-Tsang recurrence on polar Box-Muller normals.  Same
 *          distribution, different numbers: the INITIAL gamma differs from a
 *          real run's (the C ABI takes gamma from the host, so the device
 *          path is unaffected), and with it every later digit.
 *       2. the last ulp of gsl_sf_psi (src/lib.hh:29-70,
 *          src/snpsamplinge.cc:292-294, :734-737).  Real GSL evaluates
 *          Chebyshev fits (psi_cs / apsi_cs); orc_digamma uses the
 *          recurrence to x >= 10 plus the asymptotic series -- within 2e-15
 *          relative of scipy.special.digamma (tested), not bit-identical.
 *     Everything else the path uses from GSL is pinned by known answers:
 *     mt19937 seeding (0 -> 4357) / tempering, gsl_rng_uniform_int's
 *     rejection rule, gsl_sf_fact (exact for the arguments used).
 *     tests/test_oracle_golden.py::test_config1_regression_on_own_sampler
 *     is a regression check on this sampler, not a pin to the reference.
 *
 * Every function cites the reference file:line it restates
 * (paths relative to the upstream repository root).
 */
#ifndef TS_ORACLE_H
#define TS_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- GSL restatements (GNU Scientific Library, version unpinned by the
 *      reference: configure.ac:17-19 only checks -lgsl -lgslcblas) -------- */

typedef struct {
  uint32_t mt[624];
  int mti;
} orc_rng;

/* gsl_rng_set on gsl_rng_mt19937 (seed 0 -> 4357). src/snpsamplinge.cc:59-63 */
void orc_rng_seed(orc_rng *r, unsigned long seed);
uint32_t orc_rng_get(orc_rng *r);
/* gsl_rng_uniform_int: scale = 0xffffffff / n; do k = get()/scale while k >= n */
uint32_t orc_rng_uniform_int(orc_rng *r, uint32_t n);
double orc_rng_uniform(orc_rng *r);     /* get() / 2^32 */
double orc_rng_uniform_pos(orc_rng *r); /* rejects 0 */
/* gsl_ran_gamma (Marsaglia-Tsang).  The unit normal comes from the polar
 * Box-Muller method (gsl_ran_gaussian), NOT GSL's ziggurat: documented
 * deviation, statistically equivalent. src/snpsamplinge.cc:233 */
double orc_ran_gamma(orc_rng *r, double a, double b);
/* gsl_sf_psi restated: recurrence to x >= 10 + asymptotic series. x > 0. */
double orc_digamma(double x);

/* ---- engine ------------------------------------------------------------ */

typedef struct {
  uint32_t n, l, k;
  double alpha;            /* 1/k            src/env.hh:209 */
  double eta0, eta1;       /* 1, 1           src/env.hh:221-222 */
  double nodetau0;         /* env 1 (+1)=2   src/snpsamplinge.cc:16 */
  double nodekappa;        /* 0.5            src/env.hh:230 */
  double meanchangethresh; /* 1e-3           src/env.hh:208 */
  uint32_t online_iterations; /* 10          src/env.hh:232 */
  double gamma_scale;      /* = l            src/snpsamplinge.cc:702 */
  int nthreads;            /* work partition for the timed baseline; 1 = parity semantics */
} orc_config;

typedef struct orc_state orc_state;

void orc_default_config(orc_config *c, uint32_t n, uint32_t l, uint32_t k);
orc_state *orc_create(const orc_config *c);
void orc_destroy(orc_state *s);

/* PLINK SNP-major payload (after the 3 magic bytes), bytes_per_snp = ceil(n/4).
 * decode rule src/snp.cc:195-228.  Returns number of missing genotypes. */
uint64_t orc_load_bed_payload(orc_state *s, const uint8_t *payload, uint64_t bytes_per_snp,
                              uint32_t first_loc, uint32_t n_locs);
/* read <prefix>.bed; checks magic 6c 1b 01 (src/snp.cc:162-183); -1 on error */
int orc_read_bed_file(orc_state *s, const char *bed_path);
uint8_t orc_y(const orc_state *s, uint32_t n, uint32_t loc); /* 0,1,2,3(missing) */
/* mark (indiv, loc) held out (validation map entry) src/snpsamplinge.cc:213-217 */
void orc_set_heldout(orc_state *s, uint32_t loc, const uint32_t *indivs, uint32_t count);
int orc_kv_ok(const orc_state *s, uint32_t indiv, uint32_t loc); /* src/snpsamplinge.hh:389-408 */

/* init_gamma src/snpsamplinge.cc:226-237 (draws n*k gammas from r) */
void orc_init_gamma(orc_state *s, orc_rng *r);
/* set gamma (row-major n*k) and recompute Elogtheta/Etheta (estimate_all_theta :595-609) */
void orc_set_gamma(orc_state *s, const double *gamma);
/* init_lambda src/snpsamplinge.cc:239-250 */
void orc_init_lambda(orc_state *s);
void orc_set_lambda(orc_state *s, uint32_t loc, const double *lam /* [k][2] */);
/* set_validation_sample src/snpsamplinge.cc:196-224; returns number of locs */
uint32_t orc_set_validation_sample(orc_state *s, orc_rng *r);

/* one optimize_lambda(loc) with -nthreads 1 deferred-gamma semantics
 * (src/snpsamplinge.cc:320-366 + do_work :649-686); returns inner passes */
uint32_t orc_snp_update(orc_state *s, uint32_t loc, int hol_mode);

/* single building blocks, exposed for known-answer tests ------------------ */
/* one process() pass over all individuals: phi + lambda_t (hh:416-431, cc:742-759);
 * writes lambdat[k][2]; individuals [begin,end) only (shard partials). */
void orc_pass_partial(orc_state *s, uint32_t loc, uint32_t begin, uint32_t end, double *lambdat);
/* update_lambda + estimate_beta + convergence value (cc:267-296, :356-364) */
double orc_epilogue(orc_state *s, uint32_t loc, const double *lambdat);
/* update_gamma + estimate_theta for loc using current phi (cc:695-740) */
void orc_gamma_step(orc_state *s, uint32_t loc);

/* snp_likelihood second half (hh:336-360): sum of log-lik over held-out indivs of loc */
double orc_heldout_loglik(const orc_state *s, uint32_t loc, uint32_t *count);
/* estimate_beta(loc) only (first=true path hh:328-329) */
void orc_estimate_beta(orc_state *s, uint32_t loc);

/* accessors (row-major copies) */
const double *orc_gamma(const orc_state *s);
const double *orc_elogtheta(const orc_state *s);
const double *orc_etheta(const orc_state *s);
const double *orc_lambda(const orc_state *s);   /* [l][k][2] */
const double *orc_elogbeta(const orc_state *s); /* [l][k][2] */
const double *orc_ebeta(const orc_state *s);    /* [l][k]    */
const uint32_t *orc_c_indiv(const orc_state *s);
uint32_t orc_n_heldout_locs(const orc_state *s);
/* held-out locations ascending; indivs of loc ascending (compute_likelihood order :478-499) */
uint32_t orc_heldout_locs(const orc_state *s, uint32_t *locs_out, uint32_t cap);
uint32_t orc_heldout_indivs(const orc_state *s, uint32_t loc, uint32_t *out, uint32_t cap);

/* ---- whole-program driver (ctor + infer, src/snpsamplinge.cc:6-120, 417-459) */
typedef struct {
  uint32_t iter;
  double mean_ll;
  uint32_t count;
} orc_val_line;

typedef struct {
  unsigned long seed;     /* 0 => unseeded default stream (4357) */
  uint32_t reportfreq;
  double stop_threshold;  /* 1e-5 */
  uint32_t max_iter;      /* safety cap (0 = none); reference has none */
  orc_val_line *lines;    /* out: validation.txt content */
  uint32_t lines_cap;
  uint32_t n_lines;       /* out */
  uint32_t final_iter;    /* out */
  int stopped;            /* out: 1 if validation stop rule fired */
} orc_run_params;

/* Runs init_heldout_sets, init_gamma, init_lambda, initial likelihood, then
 * infer() until the validation stop rule (or max_iter).  State s must have
 * genotypes loaded. */
int orc_run(orc_state *s, orc_run_params *p);

/* -compute-beta mode on current gamma (src/snpsamplinge.cc:74-95, 368-381):
 * online_iterations=100, sweep loc=0..l-1 with the keep-updating-gamma quirk. */
void orc_compute_all_lambda(orc_state *s);

#ifdef __cplusplus
}
#endif
#endif
