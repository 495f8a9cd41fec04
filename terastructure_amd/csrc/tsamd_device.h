// Device-side building blocks of the SNP-minibatch SVI engine (gfx950 only).
//
// Restates, for the GPU, the arithmetic of
//   PhiRunnerE::update_phimom/update_phidad   src/snpsamplinge.hh:276-300
//   D1Array<double>::logsum/lognormalize      src/matrix.hh:271-293
//   PhiRunnerE::update_lambda_t               src/snpsamplinge.cc:742-759
//   PhiRunnerE::update_gamma/estimate_theta   src/snpsamplinge.cc:695-740
//   SNPSamplingE::update_lambda/estimate_beta src/snpsamplinge.cc:267-296
// in the linear domain: with w[n,k] = exp(Elogtheta[n,k] - max_k) and
// b[k,t] = exp(Elogbeta[loc,k,t]),  phi_t[n,k] = w[n,k] b[k,t] / sum_j w[n,j] b[j,t]
// which is the reference's softmax without a per-pass exp/log.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tsamd.h"

namespace tsamd {

constexpr int kBlock = 256;       // threads per workgroup of the generic kernels
constexpr int kMaxGrid = 2048;    // upper bound on pass-kernel workgroups

// Device-resident state machine.  No workgroup ever reads a word that another workgroup
// of the SAME launch writes: every kernel of the stream-ordered sequence carries a parity
// bit (a launch-time constant), reads State/rows of slot parity^1 and writes slot parity.
// So there are no tickets, atomics or fences in the hot path; every workgroup recomputes
// the tiny K x 2 epilogue of the previous pass redundantly from the same partial rows, in
// the same order, and therefore reaches the same decision.
struct State {
  uint32_t idx;    // schedule index of the SNP described here; 0xffffffff: none started in this schedule
  uint32_t valid;  // an SNP has been started (its gamma step is pending unless hol)
  uint32_t loc, hol;
  uint32_t iters;  // passes run for it; the partial rows of pass `iters` are pending unless done
  uint32_t done;   // SNP complete: lam[loc] / eb[loc] in the global arrays are final
  uint32_t nrows;  // partial rows written by pass `iters`
  uint32_t pad;
  unsigned long long epoch;  // launches of the state-machine sequence so far (tags peer exchanges)
  double lam[2 * TSAMD_MAX_K];  // !done: lambda[loc] before the pending pass' epilogue
  double eb[2 * TSAMD_MAX_K];   // exp(Elogbeta[loc]) used by pass `iters` (the last executed pass)
};

// What the first pass of the NEXT SNP needs besides the state, captured one SNP ahead by workgroup
// 0 of a first pass (slot = schedule index & 1): the schedule entry and lambda / exp(Elogbeta) of its
// location.  Valid for schedule index for_idx only; unusable when that location equals the
// previous SNP's (its values were still being updated when this was captured).
struct NextSnp {
  uint32_t for_idx, ent, pad[2];
  double lam[2 * TSAMD_MAX_K];
  double eb[2 * TSAMD_MAX_K];
};

struct Ctl {
  const uint32_t *sched; // the current schedule (entries: loc | hol_mode << 31), set by ts_begin: not a kernel
                         // argument of the pass kernels, so captured graphs survive a reallocation
  uint32_t sched_len;   // entries in the schedule; kernels past the end only carry state forward
  uint32_t last_iters;  // inner passes of the most recently completed SNP (for tsamd_snp_update)
  uint32_t xseq;        // in-launch exchanges done so far (ts_resident / ts_schedule: the tag of the next one is xseq + 1)
  uint32_t pad_;
  unsigned long long total_passes;
  unsigned long long pass_hist[TSAMD_PASS_HIST_BINS];  // completed SNPs by inner passes run (last bin: that many or more)
  State st[2];
  double lt[2][2 * TSAMD_MAX_K];      // sharded: this shard's summed partial rows (all-reduce input)
  double lt_sum[2][2 * TSAMD_MAX_K];  // all-reduced; read as the single "row" of the previous pass
  NextSnp nxt[2];
};

// Peer-to-peer exchange buffer of one rank (fine-grained, IPC-shared).  Every workgroup of
// a pass kernel writes its partial row for slot `parity` into
// rows[parity][(source rank * nblk + workgroup) * J ..] of EVERY rank's buffer over xGMI and
// then publishes seq[parity][source rank * nblk + workgroup] = epoch with a system-scope
// release; the next launch's prologue waits for world * nblk flags and adds the rows in
// that fixed order.  Pass kernels then run with at most 512 / world workgroups (every
// workgroup of the next launch polls all flags and re-adds all rows), never more than
// kXchgBlocks.
constexpr int kMaxRanks = 16;
constexpr int kXchgBlocks = 256;
struct Xchg {
  double rows[2][kMaxRanks * kXchgBlocks * 2 * TSAMD_MAX_K];
  unsigned long long seq[2][kMaxRanks * kXchgBlocks];
  // prog[r] = epoch of the launch rank r has STARTED (stored by its workgroup 0 at the start of
  // every launch of the sequence, into every rank's buffer): rank r has then completed every
  // earlier launch, reads of the exchange slots included.  A launch that stores rows without
  // having waited for its peers' rows of the previous launch (a first pass that follows an
  // early-converged SNP, or starts a schedule) waits for prog >= its own epoch before it
  // overwrites a slot: two slots are then always enough.
  unsigned long long prog[kMaxRanks];
  unsigned long long error;  // a bounded wait gave up
  // ts_schedule on several GPUs: level 2 of its in-launch exchange (tsamd_resident_kernels.h).  The leader of group g of
  // rank r stores its group sum as granules into row r * 8 + g of (slot, region) of EVERY rank's buffer (8-byte
  // system-scope stores over xGMI); every workgroup then polls its own rank's copy -- local memory -- for world * 8 rows.
  // Laid out by ResLay<K>::rank_sums: [2 slots][2 regions][kMaxRanks * 8 rows][granules per row of the context's K <= 128]
  unsigned long long res_sums[2 * 2 * kMaxRanks * 8 * 128];
  // ... and the same for ts_holblock's wide rows (a batch of validation locations per exchange: up to 512 granules per row,
  // one region): [2 slots][kMaxRanks * 8 rows][512 granules] (WideLay::rank_sums)
  unsigned long long res_wide[2 * kMaxRanks * 8 * 512];
};

struct ResXchg;  // in-launch exchange buffer of the resident kernels (tsamd_resident_kernels.h)

struct DevParams {
  uint8_t *bed;        // [l][colstride] 2-bit PLINK codes, shard-local, padding = missing
  uint64_t colstride;  // bytes per column (multiple of 128)
  double *w;           // [K][npad]  exp(Elogtheta - rowmax)
  double *gam;         // [K][npad]
  uint32_t *cnt;       // [npad]     c_n
  double *lam;         // [l][K][2]
  double *eb;          // [l][K][2]  exp(Elogbeta)
  Ctl *ctl;
  double *partials;    // [2][kMaxGrid][2K] partial rows, slot = launch parity
  uint32_t npad;       // padded individuals (multiple of 512)
  uint32_t npairs;     // npad / 2
  uint32_t chunk;      // items (pairs of individuals) per workgroup of the plain pass kernel
  uint32_t chunk_first; // items (individuals, or pairs with TSAMD_FIRST_VEC=2) per workgroup of the first pass
  uint32_t K;
  uint32_t max_inner;
  uint32_t sweep_alternate; // plain passes alternate their sweep direction (L2 reuse); TSAMD_SWEEP=0 disables
  uint32_t rows_from_lt; // sharded over RCCL: the previous pass' rows are the one all-reduced row ctl->lt_sum[parity^1]
  uint32_t xchg_world;   // > 0: sharded over the peer-to-peer exchange: rows = xchg->rows[parity^1][0..world)
  uint32_t xchg_rank;
  Xchg *xchg;            // this rank's buffer
  Xchg *peers[kMaxRanks]; // every rank's buffer as mapped into this process (peers[xchg_rank] == xchg)
  unsigned long long *host_error;  // pinned host words.  [0]: a bounded in-kernel wait that gives up also writes its tag here,
                                   // so tsamd_synchronize sees it without a device-to-host copy; [1]: the inner passes of
                                   // the most recently completed SNP (Ctl::last_iters), so tsamd_snp_update needs no copy either
  ResXchg *res;              // resident kernels: their exchange buffer (NULL: the context never qualified for them)
  uint32_t probe_ticks;      // resident kernels: bound of the launch's first exchange in 10 ns ticks (are all workgroups resident?)
  uint32_t xchg_gather_leaders; // ts_schedule on several GPUs: only the group leaders gather the ranks' group sums and hand the
                                // total to their members (three levels) instead of every workgroup polling world x 8 rows (two)
  uint32_t xchg_test_delay;  // test hook (TSAMD_TEST_XCHG_DELAY_US): stall between flag wait and row reads, 10 ns ticks
  uint32_t xchg_test_noguard; // test hook (TSAMD_TEST_XCHG_NOGUARD): skip the slot-reuse guard (to show the test sees the hazard)
  double alpha, eta0, eta1, nodetau0, nodekappa, gamma_scale, thresh;
};

// ---------------------------------------------------------------------------
// psi(x), x > 0.  Branch-free: shift by 10 with one rational accumulation
// (sum_{i<10} 1/(x+i) = P'(x)/P(x)), then the asymptotic series at x+10 >= 10
// (terms B2n/(2n z^2n), n = 1..7; the next term is < 5e-17 there).
// Restates what the reference gets from gsl_sf_psi (src/lib.hh:29-33,
// src/snpsamplinge.cc:292-294, :734-737).
__device__ __forceinline__ double digamma(double x) {
  const bool big = x >= 1.0e8;       // P(x) would not overflow until ~1e30, but nothing is gained
  const double xs = big ? 1.0 : x;
  double num = 0.0, den = 1.0;
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const double xi = xs + (double)i;
    num = fma(num, xi, den);
    den *= xi;
  }
  const double z = big ? x : x + 10.0;
  const double rz = 1.0 / z;
  const double f = rz * rz;
  double t = -1.0 / 12.0;
  t = fma(f, t, 691.0 / 32760.0);
  t = fma(f, t, -1.0 / 132.0);
  t = fma(f, t, 1.0 / 240.0);
  t = fma(f, t, -1.0 / 252.0);
  t = fma(f, t, 1.0 / 120.0);
  t = fma(f, t, -1.0 / 12.0);
  const double tail = log(z) - 0.5 * rz + f * t;
  return big ? tail : tail - num / den;
}

// 1/x for positive normal x: hardware estimate r0 (relative error e, |e| < 2^-24.4 measured: tools/ubench/rcp_accuracy.hip)
// times 1 + e + e^2 -- one third-order step, three FMAs: the error left is e^3 < 2^-73 plus the final rounding.  Without
// the scaling / fix-up sequence of an IEEE division.  Every vector instruction costs a resident kernel's lone wave the
// same 4.3 ... 4.7 cycles (tools/ubench/op_cost.hip), so the gamma step and the sweeps are priced in instructions.
__device__ __forceinline__ double fast_rcp(double x) {
  const double r = __builtin_amdgcn_rcp(x);
  const double e = fma(-x, r, 1.0);
  return fma(r, fma(e, e, e), r);
}

// 1/sqrt(x) for positive normal x: hardware estimate y0 (|e| = |1 - x y0^2| < 2^-23.2) times 1 + e/2 + 3 e^2/8, again one
// third-order step (ts_schedule, whose gamma step is bound by its arithmetic).
__device__ __forceinline__ double fast_rsqrt(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  const double e = fma(-(x * y), y, 1.0);
  return fma(y, e * fma(0.375, e, 0.5), y);
}

// exp(psi(x)) split as z * exp(a): z = x + 10, a = u(z) - r(x) with
// u = -1/(2z) - sum B2n/(2n z^2n) (the asymptotic series of psi(z) - log z, |u| <= 0.051) and
// r = sum_{i<10} 1/(x+i).  Pairing the terms i and 9-i gives r = (2x+9) * Q'(q)/Q(q) with
// q = x(x+9) and Q(q) = prod_{i<5} (q + i(9-i)) = prod_{i<10} (x+i)
//      = q (q^4 + 60 q^3 + 1308 q^2 + 12176 q + 40320),   Q'(q) = 5 q^4 + 240 q^3 + 3924 q^2 + 24352 q + 40320:
// two Horner chains in q with exact integer coefficients, all terms positive (nothing cancels), and one reciprocal
// serves both r and 1/z.
// Lets the gamma step form w[k] = z_k * exp(a_k - a_max) -- exp(Elogtheta) up to a
// per-individual factor -- with one exp and no log per population.
__device__ __forceinline__ void exp_digamma_split(double x, double &z, double &a) {
  const double q = x * (x + 9.0);
  double den = q + 60.0;
  den = fma(den, q, 1308.0);
  den = fma(den, q, 12176.0);
  den = fma(den, q, 40320.0);
  den *= q;
  double num = fma(5.0, q, 240.0);
  num = fma(num, q, 3924.0);
  num = fma(num, q, 24352.0);
  num = fma(num, q, 40320.0);
  z = x + 10.0;
  const double inv = fast_rcp(den * z);
  const double r = (fma(2.0, x, 9.0) * num) * (z * inv);
  const double rz = den * inv;
  const double f = rz * rz;
  // u + 1/(2z) = f P(f), f = 1/z^2 <= 0.01: degree-4 minimax fit (tools/fit/psi_tail_minimax.py; absolute error 1.0e-17, below
  // the 4e-17 the seven-term asymptotic series -1/12 + f/120 - f^2/252 + ... leaves at z = 10)
  double t = -0x1.ca8ce68269ac5p-8;
  t = fma(f, t, 0x1.10a92b671fc5cp-8);
  t = fma(f, t, -0x1.040fbe5140bffp-8);
  t = fma(f, t, 0x1.111110ed07d8fp-7);
  t = fma(f, t, -0x1.5555555554867p-4);
  a = fma(rz, fma(rz, t, -0.5), -r);  // -1/(2z) + f P(f) - r
}

// exp(d) for d <= 0 (the a_k - a_max above; also fine for moderate d > 0 -- nothing here depends on the sign, only
// overflow is not handled): n = round(d / ln 2) taken from the low bits of d / ln 2 + 1.5 * 2^52 (no rounding and no
// conversion instruction; |d| < 1.4e9, i.e. gamma > 1e-9), two-constant Cody-Waite reduction by ln 2, degree-11 minimax
// polynomial on |r| <= ln(2)/2, v_ldexp for the scaling -- which also flushes the far tail to 0 --
// and none of the overflow / NaN selects of the library exp.
__device__ __forceinline__ double exp_nonpos(double d) {
  constexpr double kShift = 6755399441055744.0;  // 1.5 * 2^52: the sum's low word is n in two's complement
  const double t = fma(d, 1.4426950408889634074, kShift);
  const double n = t - kShift;
  double r = fma(n, -6.93147180369123816490e-01, d);
  r = fma(n, -1.90821492927058770002e-10, r);
  // degree-11 minimax fit of exp on |r| <= 0.3475 (tools/fit/exp_minimax.py: relative error 1.6e-17 with the coefficients
  // rounded to double; the degree-13 Taylor polynomial it replaces left 6e-18)
  double p = 0x1.ad64c1d19cd83p-26;
  p = fma(p, r, 0x1.28b42b3d7df6ep-22);
  p = fma(p, r, 0x1.71df47fc1ca59p-19);
  p = fma(p, r, 0x1.a01991a1cec39p-16);
  p = fma(p, r, 0x1.a01a010e063f0p-13);
  p = fma(p, r, 0x1.6c16c187f1c93p-10);
  p = fma(p, r, 0x1.11111111318cbp-7);
  p = fma(p, r, 0x1.555555554f156p-5);
  p = fma(p, r, 0x1.555555555549dp-3);
  p = fma(p, r, 0x1.0000000000010p-1);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return __builtin_amdgcn_ldexp(p, (int)(uint32_t)__double_as_longlong(t));  // (2^-huge flushes to 0)
}

// Cross-lane moves of the wave fold, all on the vector ALU (no trip through the LDS pipe):
//  * pair_add<32|16>(a, b): v_permlane32_swap / v_permlane16_swap (gfx950) exchange the upper
//    half-wave (odd 16-lane rows) of a with the lower half-wave (even rows) of b, after which
//    a' + b' is, in every lane, "the operand this lane keeps + the same operand of its partner
//    lane (lane ^ 32 / lane ^ 16)": lower lanes keep a, upper lanes keep b.
//  * partner<8|4|2|1>(v): DPP row_mirror / row_half_mirror / quad_perm: the value of the lane's
//    partner at that level (lane ^ 15, ^ 7, ^ 2, ^ 1 -- any pairing across the level's bit works
//    for a halving butterfly).
template <int OFF>
__device__ __forceinline__ double pair_add(double a, double b) {
  static_assert(OFF == 32 || OFF == 16, "half-wave / row swaps only");
  const unsigned long long ua = __double_as_longlong(a), ub = __double_as_longlong(b);
  unsigned a_lo, a_hi, b_lo, b_hi;
  if constexpr (OFF == 32) {
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)ua, (unsigned)ub, false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(ua >> 32), (unsigned)(ub >> 32), false, false);
    a_lo = lo[0], b_lo = lo[1], a_hi = hi[0], b_hi = hi[1];
  } else {
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)ua, (unsigned)ub, false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(ua >> 32), (unsigned)(ub >> 32), false, false);
    a_lo = lo[0], b_lo = lo[1], a_hi = hi[0], b_hi = hi[1];
  }
  return __longlong_as_double(((unsigned long long)a_hi << 32) | a_lo) +
         __longlong_as_double(((unsigned long long)b_hi << 32) | b_lo);
}

template <int OFF>
__device__ __forceinline__ double partner(double v) {
  static_assert(OFF == 8 || OFF == 4 || OFF == 2 || OFF == 1, "DPP levels only");
  constexpr int ctrl = OFF == 8 ? 0x140 /* row_mirror */ : OFF == 4 ? 0x141 /* row_half_mirror */
                       : OFF == 2 ? 0x4E /* quad_perm [2,3,0,1] */ : 0xB1 /* quad_perm [1,0,3,2] */;
  const unsigned long long u = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)u, ctrl, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(u >> 32), ctrl, 0xf, 0xf, false);
  return __longlong_as_double(((unsigned long long)(uint32_t)hi << 32) | (uint32_t)lo);
}

// a in the lanes of the banks (groups of four lanes of a 16-lane row) not in BANKS, b in the others
template <int BANKS>
__device__ __forceinline__ double bank_pick(double a, double b) {
  const unsigned long long ua = __double_as_longlong(a), ub = __double_as_longlong(b);
  const int lo = __builtin_amdgcn_update_dpp((int)(uint32_t)ua, (int)(uint32_t)ub, 0xE4 /* quad_perm [0,1,2,3] */, 0xf, BANKS, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(uint32_t)(ua >> 32), (int)(uint32_t)(ub >> 32), 0xE4, 0xf, BANKS, false);
  return __longlong_as_double(((unsigned long long)(uint32_t)hi << 32) | (uint32_t)lo);
}

// Sum NV per-lane values across the 64 lanes of a wave with a halving butterfly: at
// each step a lane keeps half of its values and trades the other half with its partner,
// so the traffic is NV + NV/2 + ... instead of 6 * NV cross-lane moves.  NV is padded to a power
// of two P <= 64.  On return v[0] of lane l holds the wave total of value
// slot(l).  Fixed order; lanes sharing a slot add in different association orders, so the
// caller takes the value of the lowest such lane.
template <int NV>
struct WaveFold {
  static constexpr int P = NV <= 1 ? 1 : NV <= 2 ? 2 : NV <= 4 ? 4 : NV <= 8 ? 8 : NV <= 16 ? 16 : NV <= 32 ? 32 : 64;
  // value index owned by lane l after fold(): bit (5 - s) of the lane selects the half at step s
  __device__ static __forceinline__ int slot(uint32_t lane) {
    int idx = 0, h = P;
#pragma unroll
    for (int s = 0; s < 6; ++s) {
      if (h == 1) break;
      h >>= 1;
      if ((lane >> (5 - s)) & 1u) idx += h;
    }
    return idx;
  }
  template <int S, int H>
  __device__ static __forceinline__ void step(double (&v)[P], uint32_t lane) {
    if constexpr (S < 6) {
      constexpr int off = 32 >> S;
      if constexpr (H > 1) {
        constexpr int h = H / 2;
        if constexpr (off >= 16) {
#pragma unroll
          for (int i = 0; i < h; ++i) v[i] = pair_add<off>(v[i], v[i + h]);
        } else if constexpr (off >= 4) {
          // the upper partners (lanes 8..15 / 4..7, 12..15 of a row: whole banks of four) keep the upper half: both picks
          // are identity DPP moves under a bank mask -- a v_cndmask on vcc costs a lone wave 17 cycles, a DPP move 4.7
          // (tools/ubench/op_cost.hip)
          constexpr int up_banks = off == 8 ? 0xC : 0xA;
#pragma unroll
          for (int i = 0; i < h; ++i) {
            const double keep = bank_pick<up_banks>(v[i], v[i + h]);
            const double send = bank_pick<up_banks>(v[i + h], v[i]);
            v[i] = keep + partner<off>(send);
          }
        } else {
          const bool up = (lane & (uint32_t)off) != 0u;  // upper partner keeps the upper half
#pragma unroll
          for (int i = 0; i < h; ++i) {
            const double keep = up ? v[i + h] : v[i];
            const double send = up ? v[i] : v[i + h];
            v[i] = keep + partner<off>(send);
          }
        }
        step<S + 1, h>(v, lane);
      } else {
        if constexpr (off >= 16)
          v[0] = pair_add<off>(v[0], v[0]);
        else
          v[0] += partner<off>(v[0]);
        step<S + 1, 1>(v, lane);
      }
    }
  }
  __device__ static __forceinline__ double fold(double (&v)[P], uint32_t lane) {
    step<0, P>(v, lane);
    return v[0];
  }
};

// PLINK 2-bit code -> (weight of the "mom" copy = y, weight of the "dad" copy = 2 - y);
// 01 (missing, or held out) -> (0, 0).  src/snp.cc:203-216, src/snpsamplinge.cc:755-756.
__device__ __forceinline__ void code_weights(uint32_t c, double &mom, double &dad, bool &ok) {
  // (all integer arithmetic, no select: a v_cndmask on vcc costs a resident kernel's lone wave four times an integer
  // instruction, tools/ubench/op_cost.hip -- and this runs once per individual and pass)
  const uint32_t hi = c >> 1, lo = c & 1u;
  const uint32_t miss = lo & (hi ^ 1u);  // 01
  ok = miss == 0u;
  const uint32_t y = hi * (1u + lo);     // 00 -> 0, 10 -> 1, 11 -> 2 (01 -> 0)
  mom = (double)y;
  dad = (double)(2u - y - 2u * miss);    // 2 - y, and 0 for a missing genotype (y = 0 there)
}

// wave-uniform double -> scalar registers (frees VGPRs; v_fma_f64 takes one SGPR pair)
__device__ __forceinline__ double uniform_f64(double v) {
  const unsigned long long u = __double_as_longlong(v);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
  return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

// the double held by lane `lane` (a compile-time constant) of this wave -> scalar registers
__device__ __forceinline__ double lane_f64(double v, int lane) {
  const unsigned long long u = __double_as_longlong(v);
  const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)u, lane);
  const uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(u >> 32), lane);
  return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

}  // namespace tsamd
