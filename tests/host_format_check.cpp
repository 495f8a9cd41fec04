// Checker for host/fast_format.h and host/model_writer.h (compiled and run by tests/test_host_format_cpu.py):
//   1. tsfmt::fmt_fixed8 against snprintf("%.8f") byte for byte: edge cases, every dyadic tie j / 2^s and its neighbours,
//      the neighbours of the decimal ties k * 1e-8 + 0.5e-8, and n random doubles of eight kinds (any bit pattern included);
//   2. ModelWriter (the asynchronous gamma.txt / theta.txt writer) against the fprintf("%.8f\t") loop it replaces
//      (save_gamma, src/snpsamplinge.cc:546-576), file against file;
//   3. the time of both at the given size.
// usage: host_format_check <random values> <rows> <k> <dir>
#include <cmath>
#include <cstdlib>
#include <random>
#include <vector>

#include "model_writer.h"

static unsigned long long bad = 0, total = 0;
static void check(double v) {
  char a[tsfmt::kMaxLen + 8], b[tsfmt::kMaxLen + 8];
  char *e = tsfmt::fmt_fixed8(a, v);
  *e = 0;
  snprintf(b, sizeof b, "%.8f", v);
  total++;
  if (strcmp(a, b) != 0) {
    if (bad < 10) printf("MISMATCH %a: fast '%s' glibc '%s'\n", v, a, b);
    bad++;
  }
}

static std::vector<char> slurp(const std::string &path) {
  std::vector<char> out;
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) return out;
  char buf[1 << 16];
  size_t n;
  while ((n = fread(buf, 1, sizeof buf, f)) > 0) out.insert(out.end(), buf, buf + n);
  fclose(f);
  return out;
}

int main(int argc, char **argv) {
  const unsigned long long n = argc > 1 ? strtoull(argv[1], 0, 10) : 10000000ull;
  const size_t rows = argc > 2 ? strtoull(argv[2], 0, 10) : 250000;
  const size_t k = argc > 3 ? strtoull(argv[3], 0, 10) : 8;
  const std::string dir = argc > 4 ? argv[4] : "/tmp";
  std::mt19937_64 g(12345);
  std::uniform_real_distribution<double> u01(0.0, 1.0);
  const double edges[] = {0.0, 1.0, 0.5, 1.0 / 512, 3.0 / 512, 5.0 / 1024, 0.000000005, 0.000000015, 0.000000025, 1e-9, 4.9e-9, 5e-9, 5.1e-9, 1e-300, 4.9e-324,
                          0.999999995, 0.9999999949999, 0.99999999500001, 9.999999995, 999999.999999995, 1e6, 1e6 + 0.5, 1234567.891011121, 2e6 + 1.0 / 3,
                          9007199254740992.0, 9007199254740993.0, 4611686018427387904.0, 9223372036854775807.0, 9223372036854775808.0, 1e19, 1e22, 1e300,
                          1.7976931348623157e308, INFINITY, NAN, 2.5e-8, 3.5e-8, 0.125, 0.375, 1e15 + 0.3, 123456789012.12345678, 0.1, 0.2, 0.3};
  for (double v : edges) check(v), check(-v);
  for (int s = 1; s <= 20; ++s)
    for (int j = 1; j < 4000; ++j) {
      const double v = (double)j / (double)(1u << s);
      check(v);
      check(std::nextafter(v, 0.0));
      check(std::nextafter(v, 10.0));
    }
  for (int i = 0; i < 1000000; ++i) {
    const double base = (double)(g() % 300000000ull) * 1e-8 + 0.5e-8;
    check(base);
    check(std::nextafter(base, 0.0));
    check(std::nextafter(base, 10.0));
  }
  for (unsigned long long i = 0; i < n; ++i) {
    double v;
    switch (i & 7) {
      case 0: v = u01(g); break;                                     // theta
      case 1: v = u01(g) * 2.0e6; break;                             // gamma at L = 1M (>= 10^6 included)
      case 2: v = std::ldexp(u01(g), (int)(g() % 120) - 90); break;  // a wide range of exponents
      case 3: v = (double)(g() % 100000000ull) * 1e-8; break;        // few-digit decimals
      case 4: v = -u01(g) * 1e3; break;
      case 5: {                                                      // any bit pattern (inf / nan / huge / denormal included)
        uint64_t b = g();
        memcpy(&v, &b, 8);
      } break;
      case 6: v = std::round(u01(g) * 1e9) / 1e9; break;             // round-half cases at the ninth digit
      default: v = 1.0 / (double)(1 + g() % 100000); break;
    }
    check(v);
  }
  printf("formatter: %llu values, %llu mismatches\n", total, bad);

  // the writer against the loop it replaces
  SaveJob proto;
  proto.n = rows;
  proto.k = k;
  proto.g.resize(rows * k);
  proto.t.resize(rows * k);
  for (size_t i = 0; i < rows; ++i) {
    double sum = 0;
    for (size_t j = 0; j < k; ++j) sum += (proto.g[i * k + j] = (i % 97 == 0 && j == 0) ? 1.0 / 512 : u01(g) * (i % 3 ? 1.0 : 2.0e6));
    for (size_t j = 0; j < k; ++j) proto.t[i * k + j] = proto.g[i * k + j] / sum;
  }
  Stopwatch sw;
  {
    FILE *f = fopen((dir + "/gamma_ref.txt").c_str(), "w"), *h = fopen((dir + "/theta_ref.txt").c_str(), "w");
    if (!f || !h) return 2;
    for (size_t i = 0; i < rows; ++i) {
      for (size_t j = 0; j < k; ++j) {
        fprintf(f, "%.8f\t", proto.g[i * k + j]);
        fprintf(h, "%.8f\t", proto.t[i * k + j]);
      }
      fprintf(f, "\n");
      fprintf(h, "\n");
    }
    fclose(f);
    fclose(h);
  }
  const double t_loop = sw.lap();
  ModelWriter w;
  double waited = 0, t_submit = 0;
  for (int rep = 0; rep < 2; ++rep) {  // two jobs: the second overwrites the first's files, in order
    std::unique_ptr<SaveJob> job(new SaveJob(proto));
    job->gamma_path = dir + "/gamma_new.txt";
    job->theta_path = dir + "/theta_new.txt";
    if (rep == 0) job->g[0] = -1.0;  // (must not survive)
    Stopwatch s1;
    if (!w.submit(std::move(job), &waited)) return 3;
    if (rep == 0) t_submit = s1.lap();
  }
  if (!w.drain(&waited)) return 4;
  const double t_writer = w.busy_seconds() / 2;
  w.stop();
  const bool same = slurp(dir + "/gamma_ref.txt") == slurp(dir + "/gamma_new.txt") && slurp(dir + "/theta_ref.txt") == slurp(dir + "/theta_new.txt") &&
                    !slurp(dir + "/gamma_ref.txt").empty();
  printf("writer: %zu x %zu, files %s; fprintf loop %.3f s, writer thread %.3f s per save, main thread blocked %.4f s to hand a snapshot over\n", rows, k,
         same ? "identical" : "DIFFER", t_loop, t_writer, t_submit);
  // an unwritable path is reported, not ignored
  {
    ModelWriter w2;
    std::unique_ptr<SaveJob> job(new SaveJob(proto));
    job->gamma_path = dir + "/no/such/dir/gamma.txt";
    job->theta_path = dir + "/no/such/dir/theta.txt";
    w2.submit(std::move(job), nullptr);
    if (w2.drain() || w2.error().find("cannot open") == std::string::npos) {
      printf("writer: an unwritable path went unreported\n");
      return 5;
    }
  }
  return (bad || !same) ? 1 : 0;
}
