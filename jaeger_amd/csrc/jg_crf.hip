// Linear-chain CRF (Viterbi) decoding of per-window class logits, one chain per contig (host only).
// Replaces the per-contig Python loop over postprocess/helpers.py:398-449 (viterbi_decode) that
// postprocess/collect.py:343-346,365-372 runs for `jaeger predict --crf`: emissions are the f64
// log-softmax of the logits (helpers.py:180-186 logsumexp: max-shifted), a switch from class a to b
// between adjacent windows costs costs[a][b], ties resolve to the lowest index (numpy argmax).
#include <math.h>
#include <stdint.h>

#include <vector>

#include "jg_common.h"

extern "C" int jg_viterbi_decode(const float *logits, int64_t n_windows, int32_t n_classes,
                                 const int64_t *first, int64_t n_chains, const double *costs,
                                 int32_t *path) {
  JG_REQUIRE(n_windows >= 0 && n_classes >= 1 && n_chains >= 0, JG_ERR_INVALID,
             "viterbi: n_windows=%lld n_classes=%d n_chains=%lld", (long long)n_windows, n_classes, (long long)n_chains);
  JG_REQUIRE(n_windows == 0 || (logits != nullptr && path != nullptr), JG_ERR_INVALID, "viterbi: null buffer");
  JG_REQUIRE(n_chains == 0 || (first != nullptr && costs != nullptr), JG_ERR_INVALID, "viterbi: null chain table / costs");
  const int C = n_classes;
  for (int64_t c = 0; c < n_chains; ++c)
    JG_REQUIRE(first[c] >= 0 && first[c] <= first[c + 1] && first[c + 1] <= n_windows, JG_ERR_INVALID,
               "viterbi: chain %lld spans [%lld, %lld) of %lld windows", (long long)c, (long long)first[c],
               (long long)first[c + 1], (long long)n_windows);
  std::vector<double> em, delta(C), next(C);
  std::vector<int32_t> back;
  for (int64_t c = 0; c < n_chains; ++c) {
    const int64_t w0 = first[c], T = first[c + 1] - w0;
    if (T == 0) continue;
    em.resize((size_t)T * C);
    for (int64_t t = 0; t < T; ++t) {                 // emissions = z - logsumexp(z)
      const float *z = logits + (size_t)(w0 + t) * C;
      double zmax = (double)z[0];
      for (int k = 1; k < C; ++k) zmax = fmax(zmax, (double)z[k]);
      double s = 0.0;
      for (int k = 0; k < C; ++k) s += exp((double)z[k] - zmax);
      const double lse = zmax + log(s);
      for (int k = 0; k < C; ++k) em[(size_t)t * C + k] = (double)z[k] - lse;
    }
    auto argmax = [&](const double *v) {
      int best = 0;
      for (int k = 1; k < C; ++k)
        if (v[k] > v[best]) best = k;
      return best;
    };
    if (T == 1 || C == 1) {
      for (int64_t t = 0; t < T; ++t) path[w0 + t] = argmax(&em[(size_t)t * C]);
      continue;
    }
    back.resize((size_t)T * C);
    for (int k = 0; k < C; ++k) delta[k] = em[k];
    for (int64_t t = 1; t < T; ++t) {
      for (int cur = 0; cur < C; ++cur) {             // best predecessor of `cur`: max_prev delta[prev] - cost[prev][cur]
        int bp = 0;
        double best = delta[0] - costs[cur];
        for (int prev = 1; prev < C; ++prev) {
          const double v = delta[prev] - costs[(size_t)prev * C + cur];
          if (v > best) { best = v; bp = prev; }
        }
        back[(size_t)t * C + cur] = bp;
        next[cur] = em[(size_t)t * C + cur] + best;
      }
      delta.swap(next);
    }
    int k = argmax(delta.data());
    path[w0 + T - 1] = k;
    for (int64_t t = T - 2; t >= 0; --t) {
      k = back[(size_t)(t + 1) * C + k];
      path[w0 + t] = k;
    }
  }
  return JG_OK;
}
