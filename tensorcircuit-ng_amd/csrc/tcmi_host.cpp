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

// ---- random-greedy pairwise path of a circuit network (tcmi/tn.py::greedy_path, opt_einsum's RandomGreedy; the reference
// reaches it through cotengra's "greedy" method, tensorcircuit/cons.py:1168-1190) ----------------------------------------
// Two thirds of a path search were this loop in Python (257 calls of 60 ms for the 32-qubit RQC).  Same algorithm, same
// arithmetic and tie-breaking: candidates (cost, a, b) ordered lexicographically, cost = 2^|out| - alpha (2^|a| + 2^|b|)
// with exact integer sizes converted once; stale heap entries re-costed on pop; with temperature > 0 the next pair is
// drawn from the nbranch best candidates with Boltzmann weights by inverse CDF of ONE pre-drawn uniform number per
// step (`uniforms`, drawn by the caller's numpy generator: the choice is reproducible on every rank).
// Networks: every index has dimension 2 and at most two ends (one end: an output index).  inputs as bit masks of W
// 64-bit words per tensor.  ssa[2 * step] receives the (a, b) tensor ids in SSA numbering; returns the step count.
#include <algorithm>
#include <queue>

namespace {
struct Cand {
  double cost;
  int a, b;
  bool operator>(const Cand& o) const {
    if (cost != o.cost) return cost > o.cost;
    if (a != o.a) return a > o.a;
    return b > o.b;
  }
};
}  // namespace

extern "C" int tcmi_greedy_path(int ntensors, int W, const unsigned long long* masks, const unsigned long long* outmask,
                                double alpha, double temperature, int nbranch, const double* uniforms, int nuniforms,
                                int* ssa) {
  if (ntensors < 1 || W < 1 || W > 256 || !masks || !outmask || !ssa || nbranch < 1 || nbranch > 64) return -1;
  const int cap = 2 * ntensors;
  std::vector<uint64_t> live((size_t)cap * W, 0);
  std::vector<char> alive(cap, 0);
  std::vector<int> len(cap, 0);
  for (int i = 0; i < ntensors; ++i) {
    int c = 0;
    for (int w = 0; w < W; ++w) {
      live[(size_t)i * W + w] = masks[(size_t)i * W + w];
      c += __builtin_popcountll(masks[(size_t)i * W + w]);
    }
    len[i] = c;
    alive[i] = 1;
  }
  // owners of every index (at most two live tensors)
  const int nbits = 64 * W;
  std::vector<int> own((size_t)nbits * 2, -1);
  auto add_owner = [&](int e, int t) {
    if (own[2 * e] < 0) own[2 * e] = t;
    else own[2 * e + 1] = t;
  };
  auto del_owner = [&](int e, int t) {
    if (own[2 * e] == t) own[2 * e] = own[2 * e + 1], own[2 * e + 1] = -1;
    else if (own[2 * e + 1] == t) own[2 * e + 1] = -1;
  };
  for (int i = 0; i < ntensors; ++i)
    for (int w = 0; w < W; ++w) {
      uint64_t x = live[(size_t)i * W + w];
      while (x) {
        add_owner(64 * w + __builtin_ctzll(x), i);
        x &= x - 1;
      }
    }
  auto pow2 = [](int l) -> double { return std::ldexp(1.0, l); };
  auto size_sum = [&](int la, int lb) -> double {
    // Python: alpha * (2^la + 2^lb) with an exact integer sum rounded once to double
    if (la < 63 && lb < 63) return (double)((1ull << la) + (1ull << lb));
    const int hi = la > lb ? la : lb, lo = la > lb ? lb : la;
    if (hi - lo > 53) return std::ldexp(1.0, hi);
    return std::ldexp(1.0, hi) + std::ldexp(1.0, lo);
  };
  std::vector<uint64_t> tmp(W);
  auto merged_len = [&](int a, int b) -> int {
    int c = 0;
    const uint64_t *sa = &live[(size_t)a * W], *sb = &live[(size_t)b * W];
    for (int w = 0; w < W; ++w) c += __builtin_popcountll((sa[w] ^ sb[w]) | (sa[w] & sb[w] & outmask[w]));
    return c;
  };
  auto cost_of = [&](int a, int b) -> double { return pow2(merged_len(a, b)) - alpha * size_sum(len[a], len[b]); };
  std::priority_queue<Cand, std::vector<Cand>, std::greater<Cand>> heap;
  std::vector<int> seen_stamp(cap, -1);
  int stamp = 0;
  auto push = [&](int i) {
    ++stamp;
    for (int w = 0; w < W; ++w) {
      uint64_t x = live[(size_t)i * W + w];
      while (x) {
        const int e = 64 * w + __builtin_ctzll(x);
        x &= x - 1;
        for (int s = 0; s < 2; ++s) {
          const int j = own[2 * e + s];
          if (j < 0 || j == i || !alive[j] || seen_stamp[j] == stamp) continue;
          seen_stamp[j] = stamp;
          const int a = i < j ? i : j, b = i < j ? j : i;
          heap.push(Cand{cost_of(a, b), a, b});
        }
      }
    }
  };
  for (int i = 0; i < ntensors; ++i) push(i);
  auto pop_valid = [&](Cand& out) -> bool {
    while (!heap.empty()) {
      Cand c = heap.top();
      heap.pop();
      if (!alive[c.a] || !alive[c.b]) continue;
      const double real = cost_of(c.a, c.b);
      if (real != c.cost) {
        c.cost = real;
        heap.push(c);
        continue;
      }
      out = c;
      return true;
    }
    return false;
  };
  int nxt = ntensors, nstep = 0, ustep = 0;
  std::vector<Cand> cands;
  std::vector<double> wts;
  while (!heap.empty()) {
    Cand first;
    if (!pop_valid(first)) break;
    if (temperature > 0 && uniforms) {
      cands.clear();
      cands.push_back(first);
      while ((int)cands.size() < nbranch) {
        Cand c;
        if (!pop_valid(c)) break;
        bool dup = false;
        for (const Cand& o : cands) dup = dup || (o.a == c.a && o.b == c.b);
        if (dup) continue;
        cands.push_back(c);
      }
      const double c0 = cands[0].cost;
      const double scale = temperature * std::max(1.0, std::fabs(c0));
      wts.assign(cands.size(), 0.0);
      double tot = 0.0;
      for (size_t q = 0; q < cands.size(); ++q) {
        wts[q] = std::exp(-(cands[q].cost - c0) / scale);
        tot += wts[q];
      }
      // numpy's Generator.choice(n, p = w / sum w): cdf = cumsum(p), cdf /= cdf[-1], first index with u < cdf
      const double u = ustep < nuniforms ? uniforms[ustep] : 0.5;
      ++ustep;
      double acc = 0.0;
      for (size_t q = 0; q < cands.size(); ++q) {
        acc += wts[q] / tot;
        wts[q] = acc;
      }
      size_t k = cands.size() - 1;
      for (size_t q = 0; q < cands.size(); ++q)
        if (u < wts[q] / acc) {
          k = q;
          break;
        }
      for (size_t q = 0; q < cands.size(); ++q)
        if (q != k) heap.push(cands[q]);
      first = cands[k];
    }
    const int a = first.a, b = first.b;
    if (nxt >= cap) return -2;
    uint64_t* m = &live[(size_t)nxt * W];
    const uint64_t *sa = &live[(size_t)a * W], *sb = &live[(size_t)b * W];
    int c = 0;
    for (int w = 0; w < W; ++w) {
      m[w] = (sa[w] ^ sb[w]) | (sa[w] & sb[w] & outmask[w]);
      c += __builtin_popcountll(m[w]);
    }
    for (int t : {a, b}) {
      for (int w = 0; w < W; ++w) {
        uint64_t x = live[(size_t)t * W + w];
        while (x) {
          del_owner(64 * w + __builtin_ctzll(x), t);
          x &= x - 1;
        }
      }
      alive[t] = 0;
    }
    len[nxt] = c;
    alive[nxt] = 1;
    for (int w = 0; w < W; ++w) {
      uint64_t x = m[w];
      while (x) {
        add_owner(64 * w + __builtin_ctzll(x), nxt);
        x &= x - 1;
      }
    }
    ssa[2 * nstep] = a;
    ssa[2 * nstep + 1] = b;
    ++nstep;
    push(nxt);
    ++nxt;
  }
  // leftovers (disconnected components): outer products, smallest first (ties: lowest id, as Python's stable sort)
  std::vector<int> rest;
  for (int i = 0; i < nxt; ++i)
    if (alive[i]) rest.push_back(i);
  auto by_size = [&](int x, int y) { return len[x] != len[y] ? len[x] < len[y] : false; };
  std::stable_sort(rest.begin(), rest.end(), by_size);
  while (rest.size() > 1) {
    const int a = rest[0], b = rest[1];
    if (nxt >= cap) return -2;
    int c = 0;
    for (int w = 0; w < W; ++w) {
      live[(size_t)nxt * W + w] = live[(size_t)a * W + w] | live[(size_t)b * W + w];
      c += __builtin_popcountll(live[(size_t)nxt * W + w]);
    }
    len[nxt] = c;
    ssa[2 * nstep] = a;
    ssa[2 * nstep + 1] = b;
    ++nstep;
    std::vector<int> nr;
    nr.push_back(nxt);
    for (size_t q = 2; q < rest.size(); ++q) nr.push_back(rest[q]);
    std::stable_sort(nr.begin(), nr.end(), by_size);
    rest.swap(nr);
    ++nxt;
  }
  return nstep;
}

// ---- the whole subtree-reconfiguration loop (tcmi/tn.py::reconfigure_path) ------------------------------------------------
// After the dynamic programme and the greedy loop had moved here, the bookkeeping around them -- frontier expansion, step
// costs on python integers, the sorted work list of every pass -- was 2.2 of the 3.4 s a seed of the 32-qubit search takes
// (VERDICT r05 weak 12).  Same algorithm as the Python function, statement by statement: same frontier (the most
// expensive internal node first, ties to the earliest in the list), same old / new cost comparison, same node numbering
// of a rebuilt subtree (left subtree, right subtree, then the node), same work-list order (cost descending, id ascending)
// and evaluation budget -- so a search gives the same tree whichever implementation runs.
// Replaces: cotengra's subtree reconfiguration behind tensorcircuit/cons.py:1168-1190 and experimental.py
// `slicing_reconf_opts`.
namespace {
struct Reconf {
  int W;
  const double* lw;
  double alpha, cap;
  std::vector<uint64_t> idx;       // [node][W]
  std::vector<int> ka, kb;         // kids, -1 = leaf / deleted
  int nxt;
  std::vector<uint64_t> t1, t2;
  const uint64_t* m(int v) const { return &idx[(size_t)v * W]; }
  uint64_t* m(int v) { return &idx[(size_t)v * W]; }
  double step_cost(int a, int b) {
    for (int w = 0; w < W; ++w) t1[w] = m(a)[w] | m(b)[w];
    double c = std::pow(2.0, lsize(t1.data(), W, lw));
    if (alpha != 0.0) {
      for (int w = 0; w < W; ++w) t2[w] = m(a)[w] ^ m(b)[w];
      c += alpha * (std::pow(2.0, lsize(m(a), W, lw)) + std::pow(2.0, lsize(m(b), W, lw)) + std::pow(2.0, lsize(t2.data(), W, lw)));
    }
    return c;
  }
  void ensure(int v) {
    if ((size_t)(v + 1) * W > idx.size()) {
      idx.resize((size_t)(v + 1024) * W, 0);
      ka.resize(v + 1024, -1);
      kb.resize(v + 1024, -1);
    }
  }
};
}  // namespace

extern "C" int tcmi_reconfigure_path(int ntensors, int W, const unsigned long long* masks, const int* ssa_pairs, const double* lw,
                                     double cap, double alpha, int subtree_size, int max_passes, int max_evals,
                                     int* nodes_out, int* kids_out, int max_nodes_out, int* nnodes_out) {
  if (ntensors < 3 || W < 1 || W > 64 || !masks || !ssa_pairs || subtree_size < 3 || subtree_size > 16 || !nodes_out ||
      !kids_out || !nnodes_out)
    return TCMI_ERR_ARG;
  Reconf R;
  R.W = W;
  R.lw = lw;
  R.alpha = alpha;
  R.cap = cap;
  R.t1.resize(W);
  R.t2.resize(W);
  const int n = ntensors;
  R.idx.assign((size_t)(2 * n + 1024) * W, 0);
  R.ka.assign(2 * n + 1024, -1);
  R.kb.assign(2 * n + 1024, -1);
  for (int i = 0; i < n; ++i)
    for (int w = 0; w < W; ++w) R.m(i)[w] = masks[(size_t)i * W + w];
  for (int s = 0; s < n - 1; ++s) {
    const int v = n + s, a = ssa_pairs[2 * s], b = ssa_pairs[2 * s + 1];
    if (a < 0 || b < 0 || a >= v || b >= v) return TCMI_ERR_ARG;
    for (int w = 0; w < W; ++w) R.m(v)[w] = R.m(a)[w] ^ R.m(b)[w];
    R.ka[v] = a;
    R.kb[v] = b;
  }
  R.nxt = 2 * n - 1;
  std::vector<int> split((size_t)1 << subtree_size);
  std::vector<uint64_t> fmasks((size_t)subtree_size * W);
  std::vector<int> front, inner, todo;
  int evals = 0;
  for (int pass = 0; pass < max_passes; ++pass) {
    bool changed = false;
    todo.clear();
    std::vector<std::pair<double, int>> keyed;
    for (int v = 0; v < R.nxt; ++v)
      if (R.ka[v] >= 0) keyed.push_back({-R.step_cost(R.ka[v], R.kb[v]), v});
    std::sort(keyed.begin(), keyed.end());
    bool stop = false;
    for (auto& kv : keyed) {
      const int x = kv.second;
      if (R.ka[x] < 0) continue;
      // ---- optimise(x)
      bool improved = false;
      front.assign(1, x);
      inner.clear();
      while ((int)front.size() < subtree_size) {
        int pick = -1;
        double pc = 0.0;
        for (size_t i = 0; i < front.size(); ++i) {
          const int f = front[i];
          if (R.ka[f] < 0) continue;
          const double c = R.step_cost(R.ka[f], R.kb[f]);
          if (pick < 0 || c > pc) {      // python max(): the first of equal maxima
            pick = (int)i;
            pc = c;
          }
        }
        if (pick < 0) break;
        const int f = front[pick];
        front.erase(front.begin() + pick);
        inner.push_back(f);
        front.push_back(R.ka[f]);
        front.push_back(R.kb[f]);
      }
      const int k = (int)front.size();
      if (k >= 3) {
        double old = 0.0;
        for (int v : inner) old += R.step_cost(R.ka[v], R.kb[v]);
        for (int i = 0; i < k; ++i)
          for (int w = 0; w < W; ++w) fmasks[(size_t)i * W + w] = R.m(front[i])[w];
        double bf = 0.0;
        if (tcmi_subtree_dp(k, W, reinterpret_cast<const unsigned long long*>(fmasks.data()), lw, cap, alpha, split.data(), &bf) !=
            TCMI_OK)
          return TCMI_ERR_ARG;
        if (bf < old * (1.0 - 1e-9)) {
          for (int v : inner) {
            R.ka[v] = -1;
            R.kb[v] = -1;
          }
          // rebuild: left subtree, right subtree, then the node (the root keeps its id)
          struct Frame { int S; bool top; int stage; int l, r; };
          std::vector<Frame> st;
          std::vector<int> ret;
          st.push_back({(1 << k) - 1, true, 0, -1, -1});
          while (!st.empty()) {
            Frame& fr = st.back();
            if ((fr.S & (fr.S - 1)) == 0) {
              ret.push_back(front[__builtin_ctz(fr.S)]);
              st.pop_back();
              continue;
            }
            const int A = split[fr.S];
            if (fr.stage == 0) {
              fr.stage = 1;
              st.push_back({A, false, 0, -1, -1});
              continue;
            }
            if (fr.stage == 1) {
              fr.l = ret.back();
              ret.pop_back();
              fr.stage = 2;
              st.push_back({fr.S ^ A, false, 0, -1, -1});
              continue;
            }
            fr.r = ret.back();
            ret.pop_back();
            int v;
            if (fr.top) {
              v = x;
            } else {
              v = R.nxt++;
              R.ensure(v);
              uint64_t* mv = R.m(v);
              for (int w = 0; w < W; ++w) mv[w] = 0;
              for (int T = fr.S; T; T &= T - 1) {
                const uint64_t* fm = &fmasks[(size_t)__builtin_ctz(T) * W];
                for (int w = 0; w < W; ++w) mv[w] ^= fm[w];
              }
            }
            R.ka[v] = fr.l;
            R.kb[v] = fr.r;
            const int done = v;
            st.pop_back();
            ret.push_back(done);
          }
          improved = true;
        }
      }
      if (improved) changed = true;
      ++evals;
      if (evals >= max_evals) {
        stop = true;
        break;
      }
    }
    if (!changed || stop) break;
  }
  int cnt = 0;
  for (int v = 0; v < R.nxt; ++v) {
    if (R.ka[v] < 0) continue;
    if (cnt >= max_nodes_out) return TCMI_ERR_ARG;
    nodes_out[cnt] = v;
    kids_out[2 * cnt] = R.ka[v];
    kids_out[2 * cnt + 1] = R.kb[v];
    ++cnt;
  }
  *nnodes_out = cnt;
  return TCMI_OK;
}

// ---- greedy slicing of a fixed tree (tcmi/tn.py::ContractionTree._slice_fixed, networks of dimension-2 indices) -----------
// 129 candidate trees per seed are sliced this way (every random-greedy trial): half of a path search.  Same arithmetic
// and order as the Python loop: per round the score of an index = sum of the sizes of the oversize intermediates that hold
// it (accumulated in step order), candidates by (-score, label), the candidate that leaves the smallest (total oversize,
// flops) wins (first of equals in candidate order).
// Replaces: cotengra's slicing of a contraction tree (reference experimental.py:936-953 `slicing_opts`).
extern "C" int tcmi_slice_fixed(int nsteps, int W, const unsigned long long* um, const unsigned long long* km,
                                const unsigned long long* outm, const long long* labels, int target_bits, long long max_slices,
                                int max_candidates, int* sliced_bits_out, int max_sliced_out, int* nsliced_out,
                                double* flops_out) {
  if (nsteps < 1 || W < 1 || W > 64 || !um || !km || !outm || !labels || !sliced_bits_out || !nsliced_out || !flops_out ||
      max_candidates < 1)
    return TCMI_ERR_ARG;
  const int nb = 64 * W;
  std::vector<uint64_t> live(W, ~0ull), lv(W);
  std::vector<double> score(nb);
  std::vector<char> has(nb);
  std::vector<int> cands;
  long long nsl = 1;
  int ns = 0;
  auto pc_and = [&](const unsigned long long* a, const uint64_t* b) {
    int c = 0;
    for (int w = 0; w < W; ++w) c += __builtin_popcountll(a[w] & b[w]);
    return c;
  };
  for (;;) {
    int mx = 0;
    for (int s = 0; s < nsteps; ++s) {
      const int c = pc_and(km + (size_t)s * W, live.data());
      if (c > mx) mx = c;
    }
    if (mx <= target_bits) {
      double flops = 0.0;
      for (int s = 0; s < nsteps; ++s) flops += std::pow(2.0, (double)pc_and(um + (size_t)s * W, live.data()));
      *flops_out = flops;
      *nsliced_out = ns;
      return TCMI_OK;
    }
    std::fill(score.begin(), score.end(), 0.0);
    std::fill(has.begin(), has.end(), 0);
    bool any = false;
    for (int s = 0; s < nsteps; ++s) {
      const unsigned long long* k = km + (size_t)s * W;
      const int lk = pc_and(k, live.data());
      if (lk <= target_bits) continue;
      const double wgt = std::pow(2.0, (double)lk);
      for (int w = 0; w < W; ++w) {
        uint64_t kk = k[w] & live[w] & ~outm[w];
        while (kk) {
          const int b = 64 * w + __builtin_ctzll(kk);
          score[b] += wgt;
          has[b] = 1;
          any = true;
          kk &= kk - 1;
        }
      }
    }
    if (!any) {
      *nsliced_out = -1;      // cannot be sliced to the target (python: None)
      return TCMI_OK;
    }
    cands.clear();
    for (int b = 0; b < nb; ++b)
      if (has[b]) cands.push_back(b);
    std::sort(cands.begin(), cands.end(), [&](int a, int b) {
      if (score[a] != score[b]) return score[a] > score[b];
      return labels[a] < labels[b];
    });
    if ((int)cands.size() > max_candidates) cands.resize(max_candidates);
    int best = -1;
    double bo = 0.0, bfl = 0.0;
    for (int b : cands) {
      for (int w = 0; w < W; ++w) lv[w] = live[w];
      lv[b >> 6] &= ~(1ull << (b & 63));
      double over = 0.0, flops = 0.0;
      for (int s = 0; s < nsteps; ++s) {
        const int lk = pc_and(km + (size_t)s * W, lv.data());
        flops += std::pow(2.0, (double)pc_and(um + (size_t)s * W, lv.data()));
        if (lk > target_bits) over += std::pow(2.0, (double)lk);
      }
      if (best < 0 || over < bo || (over == bo && flops < bfl)) {
        best = b;
        bo = over;
        bfl = flops;
      }
    }
    nsl *= 2;
    if (nsl > max_slices || ns >= max_sliced_out) {
      *nsliced_out = -1;
      return TCMI_OK;
    }
    sliced_bits_out[ns++] = best;
    live[best >> 6] &= ~(1ull << (best & 63));
  }
}
