// Host-side helper of the contraction planner (no device code): the exact dynamic programme over the subsets of a
// subtree's frontier that tcmi/tn.py::reconfigure_path runs thousands of times per path search (3^k split tests per
// call, k <= 10).  Same arithmetic, same evaluation order and the same tie-breaking as the Python loop it replaces
// (compiled with -ffp-contract=off: a fused multiply-add would change the last bit of a cost and with it a tie), so
// every rank of a distributed run still arrives at the same tree.
// Replaces: the subtree reconfiguration cotengra applies to its trees (tensorcircuit/cons.py:1168-1190
// `optimizer_reconf`, experimental.py `slicing_reconf_opts`).
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

#include "../../include/tcmi.h"

namespace {
inline double lsize(const uint64_t* m, int W, const double* lw) {
  if (!lw) {
    int c = 0;
    for (int w = 0; w < W; ++w) c += __builtin_popcountll(m[w]);
    return (double)c;
  }
  double t = 0.0;
  for (int w = 0; w < W; ++w) {
    uint64_t x = m[w];
    while (x) {
      const int b = __builtin_ctzll(x);
      t += lw[64 * w + b];
      x &= x - 1;
    }
  }
  return t;
}
}  // namespace

extern "C" int tcmi_subtree_dp(int k, int W, const unsigned long long* masks, const double* lw, double cap, double alpha,
                               int* split, double* best_full) {
  if (k < 2 || k > 16 || W < 1 || W > 64 || !masks || !split || !best_full) return TCMI_ERR_ARG;
  const double inf = std::numeric_limits<double>::infinity();
  const int full = (1 << k) - 1;
  std::vector<uint64_t> sidx((size_t)(full + 1) * W, 0);
  std::vector<double> ssz(full + 1, 0.0), best(full + 1, inf);
  std::vector<uint64_t> tmp(W);
  for (int S = 0; S <= full; ++S) split[S] = 0;
  for (int i = 0; i < k; ++i) {
    for (int w = 0; w < W; ++w) sidx[(size_t)(1 << i) * W + w] = masks[(size_t)i * W + w];
    ssz[1 << i] = alpha != 0.0 ? std::pow(2.0, lsize(&sidx[(size_t)(1 << i) * W], W, lw)) : 0.0;
    best[1 << i] = 0.0;
  }
  for (int level = 2; level <= k; ++level) {
    for (int S = 1; S <= full; ++S) {
      if (__builtin_popcount(S) != level) continue;
      const int low = S & -S;
      uint64_t* sS = &sidx[(size_t)S * W];
      for (int w = 0; w < W; ++w) sS[w] = sidx[(size_t)low * W + w] ^ sidx[(size_t)(S ^ low) * W + w];
      const double ls = lsize(sS, W, lw);
      if (alpha != 0.0) ssz[S] = std::pow(2.0, ls);
      if (S != full && ls > cap) continue;
      double bS = inf;
      int sSplit = 0;
      for (int A = (S - 1) & S; A; A = (A - 1) & S) {
        const int B = S ^ A;
        if (A <= B) continue;
        const double ca = best[A], cb = best[B];
        if (!(ca < inf) || !(cb < inf)) continue;
        const uint64_t *a = &sidx[(size_t)A * W], *b = &sidx[(size_t)B * W];
        bool shared = false;
        for (int w = 0; w < W; ++w) {
          shared = shared || (a[w] & b[w]);
          tmp[w] = a[w] | b[w];
        }
        if (!shared) continue;
        double c = ca + cb + std::pow(2.0, lsize(tmp.data(), W, lw));
        if (alpha != 0.0) c += alpha * (ssz[A] + ssz[B] + ssz[S]);
        if (c < bS) {
          bS = c;
          sSplit = A;
        }
      }
      best[S] = bS;
      split[S] = sSplit;
    }
  }
  *best_full = best[full];
  return TCMI_OK;
}
