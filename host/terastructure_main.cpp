// terastructure (MI355X host) -- command-line front end over libtsamd.
//
// Keeps the reference's command-line surface and on-disk outputs so that
// data/run.sh works with only the binary path changed:
//   flags                src/main.cc:84-187   (dead flags accepted and ignored)
//   run directory, param.txt, network.dat symlink   src/env.hh:251-311
//   validation sample    src/snpsamplinge.cc:196-224
//   init_gamma           src/snpsamplinge.cc:226-237
//   infer loop, reports  src/snpsamplinge.cc:417-459
//   held-out likelihood, stop rule   src/snpsamplinge.cc:461-544
//   gamma.txt / theta.txt / beta.txt src/snpsamplinge.cc:546-587, :761-798
//   -compute-beta        src/snpsamplinge.cc:74-95, :368-413, :800-862
// The SVI inner loop itself (optimize_lambda + deferred gamma step) runs on the GPU
// behind include/tsamd.h.  There is no CPU path.  `-bfile <prefix>` is accepted as an
// alias of `-file <prefix>.bed`.
#include <errno.h>
#include <math.h>
#include <signal.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <fcntl.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <sstream>
#include <string>
#include <vector>

#include "fast_format.h"
#include "model_writer.h"
#include "tsamd.h"

namespace {

// ---- GSL-compatible random numbers (gsl_rng_mt19937 / gsl_rng_uniform_int /
// gsl_ran_gamma; the unit normal uses the polar Box-Muller method) -----
struct Mt19937 {
  uint32_t mt[624];
  int mti;
  explicit Mt19937(unsigned long seed = 0) { set(seed); }
  void set(unsigned long s) {
    if (s == 0) s = 4357;  // gsl mt_set
    mt[0] = (uint32_t)(s & 0xffffffffUL);
    for (int i = 1; i < 624; ++i) mt[i] = (uint32_t)(1812433253UL * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (unsigned long)i);
    mti = 624;
  }
  uint32_t get() {
    if (mti >= 624) {
      for (int kk = 0; kk < 624; ++kk) {
        const uint32_t y = (mt[kk] & 0x80000000U) | (mt[(kk + 1) % 624] & 0x7fffffffU);
        mt[kk] = mt[(kk + 397) % 624] ^ (y >> 1) ^ ((y & 1U) ? 0x9908b0dfU : 0U);
      }
      mti = 0;
    }
    uint32_t k = mt[mti++];
    k ^= (k >> 11);
    k ^= (k << 7) & 0x9d2c5680U;
    k ^= (k << 15) & 0xefc60000U;
    k ^= (k >> 18);
    return k;
  }
  uint32_t uniform_int(uint32_t n) {
    const uint32_t scale = 0xffffffffU / n;
    uint32_t k;
    do k = get() / scale;
    while (k >= n);
    return k;
  }
  double uniform_pos() {
    double x;
    do x = get() / 4294967296.0;
    while (x == 0);
    return x;
  }
  double gaussian() {
    double x, y, r2;
    do {
      x = -1 + 2 * uniform_pos();
      y = -1 + 2 * uniform_pos();
      r2 = x * x + y * y;
    } while (r2 > 1.0 || r2 == 0);
    return y * sqrt(-2.0 * log(r2) / r2);
  }
  double gamma(double a, double b) {  // Marsaglia-Tsang
    if (a < 1) {
      const double u = uniform_pos();
      return gamma(1.0 + a, b) * pow(u, 1.0 / a);
    }
    const double d = a - 1.0 / 3.0, c = (1.0 / 3.0) / sqrt(d);
    double x, v, u;
    while (true) {
      do {
        x = gaussian();
        v = 1.0 + c * x;
      } while (v <= 0);
      v = v * v * v;
      u = uniform_pos();
      if (u < 1 - 0.0331 * x * x * x * x) break;
      if (log(u) < 0.5 * x * x + d * (1 - v + log(v))) break;
    }
    return b * d * v;
  }
};

volatile sig_atomic_t g_terminate = 0;
void term_handler(int) {
  printf("Got termination signal. Saving model state and quitting.\n");
  fflush(stdout);
  g_terminate = 1;
}

struct Options {
  std::string datfname = "network.dat", label, eta_type = "default", idfile, locations_file;
  uint32_t n = 0, k = 0, l = 0, rfreq = 10000, nthreads = 6;
  bool rfreq_set = false, force = false, file_suffix = false, save_beta = false, adagrad = false;
  bool use_test_set = false, compute_beta = false, logl = false, loadcmp = false;
  double seed = 0, stop_threshold = 1e-5;
  int device = 0;
  std::vector<int> devices;  // extension: -devices a,b,...: shard the individuals over these GPUs
  uint32_t max_iter = 0;  // extension: stop after this many iterations (0 = reference behaviour)
  unsigned ingest_threads = 0;  // extension: reader threads of the .bed ingest (0 = up to 8)
  bool ingest_only = false;     // extension: read the genotypes into HBM, write param.txt, stop (ingest measurement)
};

struct Timing {
  double ingest = 0, validation_sample = 0, init_gamma = 0, training = 0, report = 0, save_blocking = 0, save_writer = 0, save_wait = 0;
  uint32_t reports = 0, saves = 0;
};

struct Run {
  Options o;
  Timing tm;
  ModelWriter writer;
  std::string prefix;  // run directory
  FILE *plog = nullptr, *logf = nullptr, *vf = nullptr;
  tsamd_ctx *ctx = nullptr;            // shard 0 (the only one on a single GPU)
  std::vector<tsamd_ctx *> ctxs;       // all shards, rank order
  time_t start_time = time(nullptr);
  uint32_t iter = 0;
  uint32_t saved_iter = 0xffffffffu;  // iteration of the last save_model
  std::map<uint32_t, std::vector<uint32_t>> validation;  // loc -> ascending individuals
  // stop rule state (src/snpsamplinge.cc:26-31)
  double prev_h = -2147483647, max_h = -2147483647;
  uint32_t nh = 0;
  std::string bed_path;
  std::vector<std::string> labels;      // -idfile: individual labels, echoed into gammasave.txt
  std::vector<uint8_t> text_payload;   // .012 input: the columns re-packed as PLINK codes, kept for read_column
  uint64_t bytes_per_snp = 0;
  bool columns_on_device = false;       // individual-major input: a column exists only in HBM (read_column downloads it)
  // a resident launch that could not get its workgroups resident lowers the context to one launch per pass; the mode is
  // raised again after `raise_in` further schedules, with a back-off that doubles while the tenant keeps coming back
  uint32_t recoveries_seen = 0, raise_in = 0, raise_backoff = 1;
  bool lowered = false;

  std::string file_str(const std::string &f) const { return prefix + f; }
  uint32_t duration() const { return (uint32_t)(time(nullptr) - start_time); }
  void lerr(const char *fmt, ...) {
    if (!logf) return;
    va_list ap;
    va_start(ap, fmt);
    vfprintf(logf, fmt, ap);
    va_end(ap);
    fputc('\n', logf);
    fflush(logf);
  }
  void plog_u(const char *k, unsigned long v) { fprintf(plog, "%s: %lu\n", k, v), fflush(plog); }
  void plog_d(const char *k, double v) { fprintf(plog, "%s: %.9f\n", k, v), fflush(plog); }
  void plog_b(const char *k, bool v) { fprintf(plog, "%s: %s\n", k, v ? "True" : "False"), fflush(plog); }
};

[[noreturn]] void die(Run &r, const char *what) {
  fprintf(stderr, "error: %s: %s\n", what, tsamd_last_error(r.ctx));
  r.lerr("error: %s: %s", what, tsamd_last_error(r.ctx));
  r.writer.stop();  // (a snapshot already handed over is written out completely: no half-written gamma.txt)
  exit(-1);
}
#define TS(r, call)               \
  do {                            \
    if ((call) != 0) die(r, #call); \
  } while (0)

// ---- the shards (one context per GPU; a single one unless -devices is given) ----------
// Every operation goes to all shards from this thread, like the reference's main thread
// driving its workers: enqueue on every context, then wait for each.
void shard_span(const Run &r, size_t i, uint32_t &begin, uint32_t &count) {
  tsamd_shard_range(r.o.n, (uint32_t)i, (uint32_t)r.ctxs.size(), &begin, &count);
}

void run_all(Run &r, const uint32_t *locs, uint32_t n, int hol_mode) {
  if (r.lowered && r.ctxs.size() == 1 && --r.raise_in == 0u) {
    // a transient tenant must not cost the rest of a multi-hour run its launch mode: try the resident kernels again
    if (tsamd_set_launch_mode(r.ctx, TSAMD_LAUNCH_PER_SCHEDULE) == 0 || tsamd_set_launch_mode(r.ctx, TSAMD_LAUNCH_PER_SNP) == 0) {
      r.lowered = false;
      r.lerr("launch mode raised again after a replayed schedule (next retry distance %u schedules)", r.raise_backoff);
    } else {
      r.raise_in = r.raise_backoff;
    }
  }
  if (tsamd_run_schedule_all(r.ctxs.data(), (uint32_t)r.ctxs.size(), locs, n, hol_mode) != 0) {
    for (tsamd_ctx *c : r.ctxs)
      if (*tsamd_last_error(c)) r.ctx = c;
    die(r, "tsamd_run_schedule_all");
  }
  for (tsamd_ctx *c : r.ctxs)
    if (tsamd_synchronize(c) != 0) {
      r.ctx = c;
      die(r, "tsamd_synchronize");
    }
  // a resident launch that found compute units of its GPU taken was replayed one launch per pass (include/tsamd.h,
  // tsamd_set_launch_mode): the run goes on, the log says so once per event
  uint32_t now = 0;
  if (tsamd_recoveries(r.ctx, &now) == 0 && now != r.recoveries_seen) {
    r.recoveries_seen = now;
    fprintf(stderr, "%s\n", tsamd_last_error(r.ctx));
    r.lerr("%s", tsamd_last_error(r.ctx));
    r.lowered = true;
    r.raise_in = r.raise_backoff;
    r.raise_backoff = std::min<uint32_t>(r.raise_backoff * 2u, 64u);
  }
}

void set_gamma_all(Run &r, const std::vector<double> &g) {
  for (size_t i = 0; i < r.ctxs.size(); ++i) {
    uint32_t b, c;
    shard_span(r, i, b, c);
    TS(r, tsamd_set_gamma(r.ctxs[i], g.data() + (size_t)b * r.o.k));
  }
}

void destroy_all(Run &r) {
  for (tsamd_ctx *c : r.ctxs) tsamd_destroy(c);
  r.ctxs.clear();
  r.ctx = nullptr;
}

void usage() {
  fprintf(stdout,
          "Population inference software for SNP data (MI355X build).\n"
          "terastructure [OPTIONS]\n"
          "\t-help\t\tusage\n"
          "\t-file <name>\t PLINK .bed file (SNP-major; .bim/.fam are found next to it), or a .012 text file\n"
          "\t\t\t (one line per location, one character 0/1/2/- per individual)\n"
          "\t-bfile <prefix>\t same as -file <prefix>.bed\n"
          "\t-n <N>\t\t number of individuals\n"
          "\t-l <L>\t\t number of locations\n"
          "\t-k <K>\t\t number of populations (this build: at most 128; the reference takes any K)\n"
          "\t-label\t\t descriptive tag for the output directory\n"
          "\t-force\t\t overwrite existing output directory\n"
          "\t-rfreq <val>\t checks for convergence and logs output every <val> iterations\n"
          "\t-seed <val>\t random seed\n"
          "\t-compute-beta\t compute allele frequencies given ./gamma.txt\n"
          "\t-device <id>\t HIP device ordinal (default 0)\n"
          "\t-devices <a,b,..>\t shard the individuals over these HIP devices (one shard each)\n"
          "\t-ingest-threads <T>\t reader threads of the .bed ingest (default: up to 8)\n"
          "\t-ingest-only\t read the genotypes into HBM, report the rate, write param.txt and stop\n");
  fflush(stdout);
}

int count_lines(const std::string &path) {
  FILE *f = fopen(path.c_str(), "r");
  if (!f) return -1;
  int n = 0;
  char buf[20480];
  while (fgets(buf, sizeof buf, f) != nullptr) n++;
  fclose(f);
  return n;
}

// Env::Env directory naming + param.txt (src/env.hh:251-311)
void setup_run_dir(Run &r) {
  const Options &o = r.o;
  std::ostringstream sa;
  sa << "n" << o.n << "-" << "k" << o.k << "-" << "l" << o.l;
  if (o.label != "")
    sa << "-" << o.label;
  else if (o.datfname.length() > 3) {
    std::string q = o.datfname.substr(0, 2);
    if (q == "..") q = "xx";
    sa << "-" << q;
  }
  if (o.seed != 0) sa << "-" << "seed" << o.seed;
  r.prefix = sa.str();
  fprintf(stdout, "+ Creating directory %s\n", r.prefix.c_str());
  struct stat st;
  if (stat(r.prefix.c_str(), &st) != 0) {
    if (errno != ENOENT || mkdir(r.prefix.c_str(), S_IRWXU | S_IRWXG | S_IROTH | S_IXOTH) != 0) {
      fprintf(stderr, "Warning: could not create dir %s\n", r.prefix.c_str());
      exit(-1);
    }
  } else if (!o.force) {
    fprintf(stderr, "Error: dir %s already exists\n", r.prefix.c_str());
    exit(-1);
  }
  r.logf = fopen(r.file_str("/infer.log").c_str(), "w");
  r.plog = fopen(r.file_str("/param.txt").c_str(), "w");
  if (!r.plog || !r.logf) {
    printf("cannot open param file:%s\n", strerror(errno));
    exit(-1);
  }
  const uint32_t blocks = o.n > 10000 ? 100 : 10;
  r.plog_u("n", o.n);
  r.plog_u("k", o.k);
  r.plog_u("t", 2);
  r.plog_u("l", o.l);
  r.plog_u("nthreads", o.nthreads);
  r.plog_d("tau0", 1);
  r.plog_d("nodetau0", 1);
  r.plog_d("kappa", 0.5);
  r.plog_d("nodekappa", 0.5);
  r.plog_d("alpha", 1.0 / o.k);
  r.plog_d("heldout_indiv_ratio", 0.001);
  r.plog_d("validation_ratio", 0.005);
  r.plog_u("online_iterations", 10);
  r.plog_d("GSL seed", o.seed);
  r.plog_b("file suffix", o.file_suffix);
  r.plog_b("save beta", o.save_beta);
  r.plog_b("adagrad", o.adagrad);
  r.plog_u("indiv sample size", o.n / blocks);
  r.plog_u("blocks", blocks);
  r.plog_b("compute_beta", o.compute_beta);
  r.plog_d("stop_threshold", o.stop_threshold);
  const std::string nd = r.file_str("/network.dat");
  unlink(nd.c_str());
  if (symlink(o.datfname.c_str(), nd.c_str()) < 0) fprintf(stderr, "warning: cannot symlink %s\n", nd.c_str());
  fprintf(stderr, "+ done initializing env\n");
}

// SNP::read_bed (src/snp.cc:95-253): line counts must match -l / -n, magic 6c 1b 01,
// then the SNP-major payload goes to HBM as is.
void read_bed(Run &r) {
  const Options &o = r.o;
  const std::string prefix = o.datfname.substr(0, o.datfname.length() - 4);
  const int l = count_lines(prefix + ".bim");
  if (l < 0) {
    fprintf(stderr, "cannot open file %s.bim:%s\n", prefix.c_str(), strerror(errno));
    exit(-1);
  }
  printf("+ bim file tells us %d SNPs\n", l);
  if ((uint32_t)l != o.l) {
    fprintf(stderr, "-l input doesn't match SNPs in bim file\n");
    exit(-1);
  }
  const int n = count_lines(prefix + ".fam");
  if (n < 0) {
    fprintf(stderr, "cannot open file %s.fam:%s\n", prefix.c_str(), strerror(errno));
    exit(-1);
  }
  printf("+ fam file tells us %d individuals\n", n);
  if ((uint32_t)n != o.n) {
    fprintf(stderr, "-n input doesn't match individuals in fam file\n");
    exit(-1);
  }
  r.bed_path = prefix + ".bed";
  r.bytes_per_snp = ((uint64_t)o.n + 3) / 4;
  FILE *f = fopen(r.bed_path.c_str(), "rb");
  if (!f) {
    fprintf(stderr, "cannot open file %s:%s\n", r.bed_path.c_str(), strerror(errno));
    exit(-1);
  }
  unsigned char magic[3];
  if (fread(magic, 1, 3, f) != 3 || magic[0] != 108 || magic[1] != 27) {
    fprintf(stderr, "%s magic number incorrect\n", r.bed_path.c_str());
    exit(-1);
  }
  if (magic[2] == 0) {
    // PLINK individual-major layout: refused by the reference (src/snp.cc:176-178); here the rows are
    // transposed on the device into the SNP-major columns the engine keeps (tsamd_upload_bed_indiv_major)
    printf("+ individual-major .bed: transposing on the device\n");
    const uint64_t bpi = ((uint64_t)o.l + 3) / 4;
    const uint32_t per = (uint32_t)std::max<uint64_t>(16, std::min<uint64_t>(o.n, ((uint64_t)256 << 20) / bpi / 16 * 16));
    std::vector<uint8_t> rows((size_t)per * bpi);
    for (uint32_t i0 = 0; i0 < o.n; i0 += per) {
      const uint32_t ni = std::min(per, o.n - i0);
      if (fread(rows.data(), bpi, ni, f) != ni) {
        fprintf(stderr, "%s is truncated: fewer than %u individuals\n", r.bed_path.c_str(), o.n);
        exit(-1);
      }
      for (tsamd_ctx *c : r.ctxs) TS(r, tsamd_upload_bed_indiv_major(c, rows.data(), bpi, i0, ni));
      printf("\r%d individuals read", i0 + ni);
      fflush(stdout);
    }
    fclose(f);
    r.columns_on_device = true;
    uint64_t cnt[4] = {0, 0, 0, 0};
    for (tsamd_ctx *c : r.ctxs) {
      uint64_t part[4];
      TS(r, tsamd_genotype_counts(c, 0, o.l, part));
      for (int q = 0; q < 4; ++q) cnt[q] += part[q];
    }
    r.plog_u("missing snps", cnt[1]);
    r.plog_u("0s snps", cnt[3]);
    r.plog_u("1s snps", cnt[2]);
    r.plog_u("2s snps", cnt[0]);
    return;
  } else if (magic[2] != 1) {
    fprintf(stderr, "mode problem in %s\n", r.bed_path.c_str());
    exit(-1);
  }
  fclose(f);
  // Streaming ingest: reader threads pread() a batch of columns into one of two pinned buffers
  // while the previous batch's strided DMA into HBM is in flight (tsamd_upload_bed_async, every
  // shard from the same buffer).  The genotype tallies of param.txt are taken on the device
  // afterwards (tsamd_genotype_counts), not one 2-bit code at a time on the host.
  const int fd = open(r.bed_path.c_str(), O_RDONLY);
  if (fd < 0) {
    fprintf(stderr, "cannot open file %s:%s\n", r.bed_path.c_str(), strerror(errno));
    exit(-1);
  }
  struct stat st;
  if (fstat(fd, &st) != 0 || (uint64_t)st.st_size < 3 + (uint64_t)o.l * r.bytes_per_snp) {
    fprintf(stderr, "%s is truncated: %llu of %llu locations\n", r.bed_path.c_str(),
            (unsigned long long)((st.st_size > 3 ? (uint64_t)st.st_size - 3 : 0) / r.bytes_per_snp), (unsigned long long)o.l);
    exit(-1);
  }
  const size_t batch = std::max<size_t>(1, std::min<size_t>((size_t)(256u << 20) / r.bytes_per_snp, o.l));
  uint8_t *buf[2] = {nullptr, nullptr};
  for (auto &b : buf) {
    void *ptr = nullptr;
    if (tsamd_host_alloc(&ptr, batch * r.bytes_per_snp) != 0) {
      fprintf(stderr, "error: %s\n", tsamd_last_error(nullptr));
      exit(-1);
    }
    b = (uint8_t *)ptr;
  }
  const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
  const unsigned nread = std::max(1u, std::min({o.ingest_threads ? o.ingest_threads : 8u, hw, (unsigned)batch}));
  auto read_batch = [&](uint8_t *dst, uint32_t loc0, size_t cols) {  // columns [loc0, loc0 + cols) -> dst
    std::vector<std::thread> th;
    std::vector<int> bad(nread, 0);
    for (unsigned t = 0; t < nread; ++t)
      th.emplace_back([&, t]() {
        const size_t c0 = cols * t / nread, c1 = cols * (t + 1) / nread;
        size_t off = c0 * r.bytes_per_snp;
        const size_t end = c1 * r.bytes_per_snp;
        while (off < end) {
          const ssize_t got = pread(fd, dst + off, std::min<size_t>(end - off, (size_t)64 << 20),
                                    3 + (off_t)loc0 * (off_t)r.bytes_per_snp + (off_t)off);
          if (got <= 0) {
            bad[t] = 1;
            return;
          }
          off += (size_t)got;
        }
      });
    for (auto &x : th) x.join();
    for (int b : bad)
      if (b) {
        fprintf(stderr, "cannot read %s\n", r.bed_path.c_str());
        exit(-1);
      }
  };
  const auto t_ing = std::chrono::steady_clock::now();
  uint32_t loc = 0;
  int cur = 0;
  size_t cols = std::min<size_t>(batch, o.l);
  read_batch(buf[cur], 0, cols);
  while (loc < o.l) {
    for (tsamd_ctx *c : r.ctxs) TS(r, tsamd_upload_bed_async(c, buf[cur], r.bytes_per_snp, loc, (uint32_t)cols));
    const uint32_t next = loc + (uint32_t)cols;
    const size_t ncols = std::min<size_t>(batch, o.l - next);
    if (ncols) read_batch(buf[cur ^ 1], next, ncols);        // overlaps the DMA of buf[cur]
    for (tsamd_ctx *c : r.ctxs) TS(r, tsamd_synchronize(c));  // buf[cur] is free again
    loc = next;
    cols = ncols;
    cur ^= 1;
    printf("\r%d locations read", loc);
    fflush(stdout);
  }
  const double ing_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_ing).count();
  close(fd);
  for (auto &b : buf) tsamd_host_free(b);
  printf("\n+ ingest: %.2f GB in %.2f s (%.1f GB/s, %u reader threads)\n", (double)o.l * r.bytes_per_snp / 1e9, ing_s,
         (double)o.l * r.bytes_per_snp / 1e9 / ing_s, nread);
  // tallies over every shard (tail bits of a column's last byte and padding excluded by the kernel)
  uint64_t cnt[4] = {0, 0, 0, 0};
  for (tsamd_ctx *c : r.ctxs) {
    uint64_t part[4];
    TS(r, tsamd_genotype_counts(c, 0, o.l, part));
    for (int q = 0; q < 4; ++q) cnt[q] += part[q];
  }
  r.plog_u("missing snps", cnt[1]);
  r.plog_u("0s snps", cnt[3]);  // labels swapped like the reference (src/snp.cc:207-216, :245-247)
  r.plog_u("1s snps", cnt[2]);
  r.plog_u("2s snps", cnt[0]);
}

// SNP::read, text branch (src/snp.cc:16-92): one line per location, one character per
// individual: '0' '1' '2' = copies of the counted allele, '-' = missing.  Each line is re-packed
// into the PLINK 2-bit codes the engine stores (0 -> 00, 1 -> 10, 2 -> 11, missing -> 01); the
// format is for small data, so the packed columns stay in host memory for the held-out
// sampling.  (The reference reads into a 20 KB line buffer; any line length works here.)
void read_012(Run &r) {
  const Options &o = r.o;
  printf("+ .012 detected");
  fprintf(stdout, "+ reading (%d,%d) snps from %s\n", o.n, o.l, o.datfname.c_str());
  fflush(stdout);
  FILE *f = fopen(o.datfname.c_str(), "r");
  if (!f) {
    r.lerr("cannot open file %s:%s", o.datfname.c_str(), strerror(errno));
    fprintf(stderr, "cannot open file %s:%s\n", o.datfname.c_str(), strerror(errno));
    exit(-1);
  }
  r.bytes_per_snp = ((uint64_t)o.n + 3) / 4;
  r.text_payload.assign((size_t)o.l * r.bytes_per_snp, 0);
  uint64_t missing = 0, a[3] = {0, 0, 0};
  static const uint8_t code_of_y[3] = {0u, 2u, 3u};
  char *line = nullptr;
  size_t cap = 0;
  uint32_t loc = 0;
  while (loc < o.l) {
    ssize_t len = getline(&line, &cap, f);
    if (len < 0) break;
    while (len > 0 && (line[len - 1] == '\n' || line[len - 1] == '\r' || line[len - 1] == ' ')) --len;
    if (len == 0) continue;
    if ((size_t)len < o.n) {
      printf("Error: unexpected lines in file\n");
      exit(-1);
    }
    uint8_t *col = r.text_payload.data() + (size_t)loc * r.bytes_per_snp;
    for (uint32_t i = 0; i < o.n; ++i) {
      uint8_t code;
      if (line[i] == '-') {
        missing++;
        code = 1u;
      } else if (line[i] >= '0' && line[i] <= '2') {
        a[line[i] - '0']++;
        code = code_of_y[line[i] - '0'];
      } else {
        fprintf(stderr, "%s: unexpected character '%c' at location %u\n", o.datfname.c_str(), line[i], loc);
        exit(-1);
      }
      col[i >> 2] |= (uint8_t)(code << (2 * (i & 3)));
    }
    loc++;
    if (loc % 10000 == 0) {
      printf("\r%d locations read", loc);
      fflush(stdout);
    }
  }
  free(line);
  fclose(f);
  if (loc != o.l) {
    fprintf(stderr, "%s is truncated: %u of %u locations\n", o.datfname.c_str(), loc, o.l);
    exit(-1);
  }
  const uint32_t batch = (uint32_t)std::max<size_t>(1, (size_t)(64u << 20) / r.bytes_per_snp);
  for (uint32_t l0 = 0; l0 < o.l; l0 += batch)
    for (tsamd_ctx *c : r.ctxs)
      TS(r, tsamd_upload_bed(c, r.text_payload.data() + (size_t)l0 * r.bytes_per_snp, r.bytes_per_snp, l0,
                             std::min(batch, o.l - l0)));
  r.plog_u("missing snps", missing);
  r.plog_u("0s snps", a[0]);
  r.plog_u("1s snps", a[1]);
  r.plog_u("2s snps", a[2]);
}

std::vector<uint8_t> read_column(Run &r, uint32_t loc) {
  if (r.columns_on_device) {  // shard after shard (shards start on multiples of 4 individuals: whole bytes)
    std::vector<uint8_t> col(r.bytes_per_snp);
    for (size_t i = 0; i < r.ctxs.size(); ++i) {
      uint32_t b, c;
      shard_span(r, i, b, c);
      TS(r, tsamd_download_bed(r.ctxs[i], loc, col.data() + b / 4, ((uint64_t)c + 3) / 4));
    }
    return col;
  }
  if (!r.text_payload.empty())
    return std::vector<uint8_t>(r.text_payload.begin() + (size_t)loc * r.bytes_per_snp,
                                r.text_payload.begin() + (size_t)(loc + 1) * r.bytes_per_snp);
  std::vector<uint8_t> col(r.bytes_per_snp);
  FILE *f = fopen(r.bed_path.c_str(), "rb");
  if (!f || fseeko(f, 3 + (off_t)loc * (off_t)r.bytes_per_snp, SEEK_SET) != 0 ||
      fread(col.data(), 1, col.size(), f) != col.size()) {
    fprintf(stderr, "cannot read location %u of %s\n", loc, r.bed_path.c_str());
    exit(-1);
  }
  fclose(f);
  return col;
}

// set_validation_sample (src/snpsamplinge.cc:196-224): kv_ok = not held out yet, not missing
void set_validation_sample(Run &r, Mt19937 &rng) {
  const uint32_t n = r.o.n, l = r.o.l;
  const uint32_t per_loc_h = n < 2000 ? (n / 10) : (n / 100);
  const double validation_ratio = 0.005;
  const uint32_t nlocs = (uint32_t)(l * validation_ratio);
  std::map<uint32_t, bool> lm;
  do {
    const uint32_t loc = rng.uniform_int(l);
    if (lm.find(loc) != lm.end()) continue;
    lm[loc] = true;
    const std::vector<uint8_t> col = read_column(r, loc);
    std::vector<bool> held(n, false);
    std::vector<uint32_t> &v = r.validation[loc];
    uint32_t c = 0;
    while (c < per_loc_h) {
      const uint32_t indiv = rng.uniform_int(n);
      const uint32_t code = (col[indiv >> 2] >> (2 * (indiv & 3))) & 3;
      if (!held[indiv] && code != 1) {
        held[indiv] = true;
        v.push_back(indiv);
        c++;
      }
    }
    std::sort(v.begin(), v.end());
    if (v.empty())
      r.validation.erase(loc);
    else
      for (tsamd_ctx *c : r.ctxs) TS(r, tsamd_set_heldout(c, loc, v.data(), (uint32_t)v.size()));
  } while (lm.size() < nlocs);
  r.plog_u("validation snps per location", per_loc_h);
  r.plog_u("validation locations", nlocs);
  r.plog_u("total validation snps", (unsigned long)per_loc_h * nlocs);
  size_t total = 0;
  for (auto &kv : r.validation) total += kv.second.size();
  r.plog_u("(VAL1) total validation snps (check)", total);
  r.plog_d("test ratio", 0.005);
  r.plog_d("validation ratio", 0.005);
}

std::string add_iter_suffix(const Run &r, const char *c) {
  std::ostringstream sa;
  if (r.o.file_suffix)
    sa << c << "_" << r.iter << ".txt";
  else
    sa << c << ".txt";
  return r.file_str(sa.str());
}

// save_gamma (src/snpsamplinge.cc:546-576)
// The main thread snapshots gamma / theta; the writer thread formats and writes them (ModelWriter above) while the next
// schedules run.  The file names are fixed here (-file-suffix: the iteration of THIS save).
void save_model(Run &r) {
  Stopwatch sw;
  const size_t n = r.o.n, k = r.o.k;
  std::unique_ptr<SaveJob> job(new SaveJob);
  job->n = n;
  job->k = k;
  job->g.resize(n * k);
  job->t.resize(n * k);
  job->gamma_path = add_iter_suffix(r, "/gamma");
  job->theta_path = add_iter_suffix(r, "/theta");
  for (size_t i = 0; i < r.ctxs.size(); ++i) {
    uint32_t b, c;
    shard_span(r, i, b, c);
    TS(r, tsamd_get_gamma(r.ctxs[i], job->g.data() + (size_t)b * k));
    TS(r, tsamd_get_theta(r.ctxs[i], job->t.data() + (size_t)b * k));
  }
  double waited = 0;
  if (!r.writer.submit(std::move(job), &waited)) {
    r.lerr("%s\n", r.writer.error().c_str());
    exit(-1);
  }
  r.tm.save_wait += waited;
  r.tm.save_blocking += sw.lap();
  r.tm.saves++;
  r.saved_iter = r.iter;
}

// every gamma.txt / theta.txt handed to the writer is complete and closed (before the process ends, on every path)
void finish_saves(Run &r) {
  double waited = 0;
  const bool ok = r.writer.drain(&waited);
  r.tm.save_wait += waited;
  r.tm.save_blocking += waited;
  r.tm.save_writer = r.writer.busy_seconds();
  if (!ok) {
    r.lerr("%s\n", r.writer.error().c_str());
    fprintf(stderr, "%s\n", r.writer.error().c_str());
    r.writer.stop();
    exit(-1);
  }
}

void write_timing(Run &r) {
  FILE *f = fopen(r.file_str("/timing.txt").c_str(), "w");
  if (!f) return;
  const Timing &t = r.tm;
  fprintf(f, "# wall-clock seconds of this run (host/terastructure_main.cpp; not a reference file)\n");
  fprintf(f, "ingest: %.3f\nvalidation sample: %.3f\ninit gamma: %.3f\ntraining: %.3f\nvalidation reports: %.3f (%u)\n", t.ingest, t.validation_sample,
          t.init_gamma, t.training, t.report, t.reports);
  fprintf(f, "save_model, main thread blocked: %.3f (%u saves; of which waiting for the writer: %.3f)\n", t.save_blocking, t.saves, t.save_wait);
  fprintf(f, "save_model, writer thread busy (overlapped with training): %.3f\n", t.save_writer);
  fprintf(f, "total: %u\n", r.duration());
  fclose(f);
}

// compute_likelihood(first, validation = true) (src/snpsamplinge.cc:461-544);
// returns true when the stop rule fires
bool compute_likelihood(Run &r, bool first) {
  uint32_t k = 0;
  double s = .0;
  // The whole block at once (tsamd_heldout_eval): theta is frozen while it runs, so the
  // reference's "per location: optimize_lambda in hol mode, _iter++, held-out sum" is one
  // hol-mode schedule over the validation locations (ascending, the map's order) followed by
  // one evaluation kernel per shard; the sums are then added in the reference's order.
  std::vector<uint32_t> locs;
  locs.reserve(r.validation.size());
  for (auto &kv : r.validation) locs.push_back(kv.first);
  const uint32_t nl = (uint32_t)locs.size();
  if (!first && nl) {  // snp_likelihood: optimize_lambda(loc) in hol mode, then _iter++
    run_all(r, locs.data(), nl, 1);
    r.iter += nl;
  }
  std::vector<std::vector<double>> sums(r.ctxs.size(), std::vector<double>(nl));
  std::vector<std::vector<uint32_t>> cnts(r.ctxs.size(), std::vector<uint32_t>(nl));
  for (size_t i = 0; i < r.ctxs.size(); ++i) {
    r.ctx = r.ctxs[i];
    TS(r, tsamd_heldout_eval(r.ctxs[i], locs.data(), nl, 0, sums[i].data(), cnts[i].data(), nullptr, nullptr));
  }
  r.ctx = r.ctxs[0];
  for (uint32_t j = 0; j < nl; ++j)
    for (size_t i = 0; i < r.ctxs.size(); ++i) {  // ascending individuals: shard after shard
      s += sums[i][j];
      k += cnts[i][j];
    }
  printf("\rdone:%.2f%%", 100.0);
  fprintf(r.vf, "%d\t%d\t%.9f\t%d\t%f\n", r.iter, r.duration(), (s / k), k, exp(s / k));
  fflush(r.vf);
  const double a = s / k;
  bool stop = false;
  if (r.iter > 2000) {
    if (a > r.prev_h && r.prev_h != 0 && fabs((a - r.prev_h) / r.prev_h) < r.o.stop_threshold)
      stop = true;
    else if (a < r.prev_h)
      r.nh++;
    else if (a > r.prev_h)
      r.nh = 0;
    if (a > r.max_h) r.max_h = a;
    if (r.nh > 3) stop = true;
  }
  r.prev_h = a;
  return stop;
}

void save_beta(Run &r, const std::vector<uint32_t> *locs) {
  const size_t k = r.o.k;
  FILE *f = fopen(add_iter_suffix(r, "/beta").c_str(), "w");
  if (!f) {
    r.lerr("cannot open beta or lambda file:%s\n", strerror(errno));
    exit(-1);
  }
  // "%d\t" loc, K x "%.8f\t", "\n" (src/snpsamplinge.cc:761-798) -- through fmt_fixed8, a block of rows per fwrite
  const uint32_t chunk = 1u << 16;
  std::vector<double> eb((size_t)chunk * k);
  std::vector<char> buf;
  auto put_rows = [&](const uint32_t *ids, uint32_t first, uint32_t nl) {
    buf.resize((size_t)nl * (k * (tsfmt::kMaxLen + 1) + 16));
    char *p = buf.data();
    for (uint32_t j = 0; j < nl; ++j) {
      p = tsfmt::fmt_uint(p, ids ? ids[j] : first + j);
      *p++ = '\t';
      for (size_t q = 0; q < k; ++q) {
        p = tsfmt::fmt_fixed8(p, eb[(size_t)j * k + q]);
        *p++ = '\t';
      }
      *p++ = '\n';
    }
    if (fwrite(buf.data(), 1, (size_t)(p - buf.data()), f) != (size_t)(p - buf.data())) {
      r.lerr("error writing beta file:%s\n", strerror(errno));
      exit(-1);
    }
  };
  if (!locs) {
    for (uint32_t l0 = 0; l0 < r.o.l; l0 += chunk) {
      const uint32_t nl = std::min(chunk, r.o.l - l0);
      TS(r, tsamd_get_ebeta(r.ctx, l0, nl, eb.data()));
      put_rows(nullptr, l0, nl);
    }
  } else {
    for (uint32_t loc : *locs) {
      TS(r, tsamd_get_ebeta(r.ctx, loc, 1, eb.data()));
      put_rows(&loc, 0, 1);
    }
  }
  fclose(f);
}

// SNP::read_idfile (src/snp.cc:255-276): one whitespace-separated label per individual, in order
void read_idfile(Run &r) {
  FILE *f = fopen(r.o.idfile.c_str(), "r");
  if (!f) {
    r.lerr("cannot open file %s:%s", r.o.idfile.c_str(), strerror(errno));
    fprintf(stderr, "error reading %s; quitting\n", r.o.idfile.c_str());
    exit(-1);
  }
  char tok[128];
  while (fscanf(f, "%127s", tok) == 1) r.labels.push_back(tok);
  fclose(f);
}

// load_gamma (src/snpsamplinge.cc:800-862): ./gamma.txt of the CURRENT directory
void load_gamma(Run &r) {
  const size_t n = r.o.n, k = r.o.k;
  FILE *gf = fopen("gamma.txt", "r");
  if (!gf) {
    r.lerr("cannot open gamma file:%s\n", strerror(errno));
    fprintf(stderr, "cannot open gamma file:%s\n", strerror(errno));
    exit(-1);
  }
  std::vector<double> g(n * k, 1.0);
  const int sz = 128 * (int)k;
  std::vector<char> line(sz);
  size_t row = 0;
  while (row < n && fgets(line.data(), sz, gf) != nullptr) {
    char *p = line.data();
    for (size_t j = 0; j < k; ++j) {
      char *q = nullptr;
      const double d = strtod(p, &q);
      if (p == q) {
        fprintf(stderr, "error parsing gamma file\n");
        exit(-1);
      }
      g[row * k + j] = d;
      p = q;
    }
    row++;
  }
  fclose(gf);
  if (row != n) {
    fprintf(stderr, "gamma.txt has %zu rows, expected %zu\n", row, n);
    exit(-1);
  }
  set_gamma_all(r, g);
  FILE *f = fopen(r.file_str("/gammasave.txt").c_str(), "w");
  if (f) {  // (id, label, K x "%.8f\t", argmax: src/snpsamplinge.cc:846-860 -- the values through fmt_fixed8, the bytes of "%.8f")
    std::vector<char> line(k * (tsfmt::kMaxLen + 1) + 1);
    for (size_t i = 0; i < n; ++i) {
      fprintf(f, "%zu\t%s\t", i, (i < r.labels.size() && !r.labels[i].empty()) ? r.labels[i].c_str() : "unknown");
      double mx = .0;
      size_t mk = 0;
      char *p = line.data();
      for (size_t j = 0; j < k; ++j) {
        p = tsfmt::fmt_fixed8(p, g[i * k + j]);
        *p++ = '\t';
        if (g[i * k + j] > mx) mx = g[i * k + j], mk = j;
      }
      fwrite(line.data(), 1, (size_t)(p - line.data()), f);
      fprintf(f, "%zu\n", mk);
    }
    fclose(f);
  }
}

void run_batch(Run &r, const std::vector<uint32_t> &locs) {
  run_all(r, locs.data(), (uint32_t)locs.size(), 0);
}

}  // namespace

int main(int argc, char **argv) {
  signal(SIGTERM, term_handler);
  Run r;
  Options &o = r.o;
  if (argc == 1) {
    usage();
    exit(-1);
  }
  for (int i = 1; i < argc; ++i) {
    auto need = [&](const char *flag) -> const char * {
      if (i + 1 > argc - 1) {
        fprintf(stderr, "+ insufficient arguments for %s!\n", flag);
        exit(-1);
      }
      return argv[++i];
    };
    const char *a = argv[i];
    if (!strcmp(a, "-help")) {
      usage();
      exit(0);
    } else if (!strcmp(a, "-force")) {
      fprintf(stdout, "+ overwrite option set\n");
      o.force = true;
    } else if (!strcmp(a, "-online") || !strcmp(a, "-E") || !strcmp(a, "-stochastic")) {
      fprintf(stdout, "+ stochastic option set\n");
    } else if (!strcmp(a, "-file") || !strcmp(a, "-bed")) {
      o.datfname = need(a);
      fprintf(stdout, "+ using file %s\n", o.datfname.c_str());
    } else if (!strcmp(a, "-bfile")) {
      o.datfname = std::string(need(a)) + ".bed";
      fprintf(stdout, "+ using file %s\n", o.datfname.c_str());
    } else if (!strcmp(a, "-batch")) {
      fprintf(stdout, "batch option currently not available");
      exit(0);
    } else if (!strcmp(a, "-n")) {
      o.n = atoi(need(a));
      fprintf(stdout, "+ n = %d\n", o.n);
    } else if (!strcmp(a, "-k")) {
      o.k = atoi(need(a));
      fprintf(stdout, "+ K = %d\n", o.k);
    } else if (!strcmp(a, "-l")) {
      o.l = atoi(need(a));
      fprintf(stdout, "+ L = %d\n", o.l);
    } else if (!strcmp(a, "-label")) {
      o.label = need(a);
    } else if (!strcmp(a, "-eta-type")) {
      o.eta_type = need(a);
    } else if (!strcmp(a, "-rfreq")) {
      o.rfreq = atoi(need(a));
      o.rfreq_set = true;
      fprintf(stdout, "+ rfreq = %d\n", o.rfreq);
    } else if (!strcmp(a, "-logl")) {
      o.logl = true;
    } else if (!strcmp(a, "-idfile")) {
      o.idfile = need(a);  // labels are only echoed into gammasave.txt by the reference
      fprintf(stdout, "+ idfile = %s\n", o.idfile.c_str());
    } else if (!strcmp(a, "-loadcmp")) {
      o.loadcmp = true;
    } else if (!strcmp(a, "-seed")) {
      o.seed = atof(need(a));
      fprintf(stdout, "+ random seed set to %.5f\n", o.seed);
    } else if (!strcmp(a, "-file-suffix")) {
      o.file_suffix = true;
    } else if (!strcmp(a, "-save-beta")) {
      o.save_beta = true;
    } else if (!strcmp(a, "-adagrad")) {
      o.adagrad = true;
    } else if (!strcmp(a, "-nthreads")) {
      o.nthreads = atoi(need(a));  // accepted; the GPU replaces the worker threads
    } else if (!strcmp(a, "-use-test-set")) {
      o.use_test_set = true;  // broken in the reference (writes to an unopened FILE*); ignored
    } else if (!strcmp(a, "-locations-file")) {
      o.locations_file = need(a);
    } else if (!strcmp(a, "-compute-beta")) {
      o.compute_beta = true;
    } else if (!strcmp(a, "-stop-threshold")) {
      o.stop_threshold = atof(need(a));
    } else if (!strcmp(a, "-device")) {
      o.device = atoi(need(a));
    } else if (!strcmp(a, "-devices")) {
      o.devices.clear();
      for (const char *q = need(a); *q;) {
        char *e = nullptr;
        o.devices.push_back((int)strtol(q, &e, 10));
        if (e == q) {
          fprintf(stdout, "error: -devices expects a comma separated list of device ordinals\n");
          exit(-1);
        }
        q = (*e == ',') ? e + 1 : e;
      }
    } else if (!strcmp(a, "-max-iter")) {
      o.max_iter = atoi(need(a));
    } else if (!strcmp(a, "-ingest-threads")) {
      o.ingest_threads = (unsigned)atoi(need(a));
    } else if (!strcmp(a, "-ingest-only")) {
      o.ingest_only = true;
    } else {
      fprintf(stdout, "error: unknown option %s\n", a);
      exit(-1);
    }
  }
  if (!o.rfreq_set) o.rfreq = 100000;
  if (o.n == 0 || o.l == 0 || o.k == 0) {
    fprintf(stderr, "error: -n, -l and -k are required\n");
    exit(-1);
  }
  if (o.loadcmp) {
    fprintf(stdout, "+ loadcmp option set: nothing to do\n");
    return 0;
  }

  // the GPU context comes first so that a missing device fails before any output exists
  tsamd_config cfg;
  tsamd_default_config(&cfg, o.n, o.l, o.k);
  if (o.compute_beta) cfg.max_inner = 100;  // tightly optimize given the thetas (:75)
  if (o.devices.empty()) o.devices.push_back(o.device);
  cfg.world = (uint32_t)o.devices.size();
  for (size_t i = 0; i < o.devices.size(); ++i) {
    cfg.device = o.devices[i];
    cfg.rank = (uint32_t)i;
    tsamd_ctx *c = nullptr;
    if (tsamd_create(&cfg, &c) != 0) {
      fprintf(stderr, "error: tsamd_create: %s\n", tsamd_last_error(nullptr));
      return -1;
    }
    r.ctxs.push_back(c);
  }
  r.ctx = r.ctxs[0];
  if (r.ctxs.size() > 1 && tsamd_p2p_connect_local(r.ctxs.data(), (uint32_t)r.ctxs.size()) != 0) {
    fprintf(stderr, "error: tsamd_p2p_connect_local: %s\n", tsamd_last_error(r.ctx));
    return -1;
  }

  setup_run_dir(r);
  Stopwatch phase;
  {  // SNP::read (src/snp.cc:9-21): the extension decides
    const std::string ext = o.datfname.size() >= 4 ? o.datfname.substr(o.datfname.size() - 4) : "";
    if (ext == ".bed") {
      printf("+ bed format detected\n");
      read_bed(r);
    } else if (ext == ".012") {
      read_012(r);
    } else {
      r.lerr("unrecognized file extension");
      fprintf(stderr, "unrecognized file extension\n");
      exit(-1);
    }
  }
  r.tm.ingest = phase.lap();
  if (o.ingest_only) {
    write_timing(r);
    destroy_all(r);
    return 0;
  }
  if (o.idfile != "") read_idfile(r);  // (src/main.cc:208)
  printf("+ initialization begin\n");
  fflush(stdout);
  r.plog_u("individuals n", o.n);
  r.plog_u("locations l", o.l);
  r.plog_u("populations k", o.k);

  Mt19937 rng(0);  // gsl_rng_alloc: default seed 0 -> 4357
  if (o.seed) rng.set((unsigned long)o.seed);
  unlink(r.file_str("/likelihood-analysis.txt").c_str());
  r.vf = fopen(r.file_str("/validation.txt").c_str(), "w");
  if (!r.vf) {
    printf("cannot open heldout file:%s\n", strerror(errno));
    exit(-1);
  }

  if (o.compute_beta) {
    set_validation_sample(r, rng);
    r.lerr("done starting threads");
    load_gamma(r);
    r.lerr("done estimating all theta");
    std::vector<uint32_t> locs;
    if (o.locations_file == "") {
      locs.resize(o.l);
      for (uint32_t i = 0; i < o.l; ++i) locs[i] = i;
    } else {
      FILE *f = fopen(o.locations_file.c_str(), "r");
      if (!f) {
        fprintf(stderr, "cannot open %s\n", o.locations_file.c_str());
        exit(-1);
      }
      char line[16384];
      while (fgets(line, sizeof line, f) != nullptr) {
        unsigned v;
        if (sscanf(line, "%u", &v) == 1 && v < o.l) locs.push_back(v);
      }
      fclose(f);
    }
    // compute_all_lambda (:368-381): the same hot path per location, gamma steps included
    const size_t chunk = 4096;
    for (size_t i0 = 0; i0 < locs.size(); i0 += chunk) {
      std::vector<uint32_t> part(locs.begin() + i0, locs.begin() + std::min(locs.size(), i0 + chunk));
      run_batch(r, part);
      r.iter += (uint32_t)part.size();
      printf("\rloc = %d took %d secs", r.iter, r.duration());
      fflush(stdout);
    }
    save_beta(r, o.locations_file == "" ? nullptr : &locs);
    destroy_all(r);
    return 0;
  }

  phase.lap();
  set_validation_sample(r, rng);
  r.tm.validation_sample = phase.lap();
  {  // init_gamma (:226-237): n-major, k inner, Gamma(100 v, 0.01)
    std::vector<double> g((size_t)o.n * o.k);
    for (size_t i = 0; i < g.size(); ++i) {
      const double v = (o.k < 100) ? 1.0 : (double)100.0 / o.k;
      g[i] = rng.gamma(100 * v, 0.01);
    }
    set_gamma_all(r, g);
  }
  r.tm.init_gamma = phase.lap();
  printf("+ computing initial heldout likelihood\n");
  compute_likelihood(r, true);
  r.tm.report += phase.lap();
  r.tm.reports++;
  save_model(r);
  printf("\n+ computing initial training likelihood\n+ done..\n+ initialization end\n");
  fflush(stdout);

  // infer() (:417-459).  Locations are drawn one per iteration from the same stream; they
  // are data-independent, so a whole report period is enqueued at once.
  while (true) {
    uint32_t c = o.rfreq - (r.iter % o.rfreq);
    c = std::min<uint32_t>(c, 4096);  // bounds the latency of a SIGTERM
    if (o.max_iter && r.iter + c > o.max_iter) c = o.max_iter > r.iter ? o.max_iter - r.iter : 0;
    std::vector<uint32_t> locs(c);
    for (uint32_t i = 0; i < c; ++i) locs[i] = rng.uniform_int(o.l);
    phase.lap();
    if (c) run_batch(r, locs);
    r.tm.training += phase.lap();
    r.iter += c;
    printf("\riteration = %d took %d secs", r.iter, r.duration());
    fflush(stdout);
    if (c && r.iter % o.rfreq == 0) {
      printf("iteration = %d took %d secs\n", r.iter, r.duration());
      r.lerr("iteration = %d took %d secs\n", r.iter, r.duration());
      r.lerr("computing heldout likelihood @ %d secs", r.duration());
      phase.lap();
      const bool stop = compute_likelihood(r, false);
      r.tm.report += phase.lap();
      r.tm.reports++;
      if (stop) {
        save_model(r);
        break;
      }
      r.lerr("saving theta @ %d secs", r.duration());
      save_model(r);
      r.lerr("done @ %d secs", r.duration());
    }
    if (g_terminate || (o.max_iter && r.iter >= o.max_iter)) {
      // (a report that ended at this very iteration has just saved this state: the same bytes would be written again)
      if (r.saved_iter != r.iter) save_model(r);
      break;
    }
  }
  printf("\n");
  finish_saves(r);  // gamma.txt / theta.txt complete and closed before the process ends
  write_timing(r);
  destroy_all(r);
  return 0;
}
