// N-gram draft proposal for speculative decoding: continue the text the way it continued the last
// time its current suffix appeared (prompt lookup).  Called once per eligible request per step from
// the scheduler, so it is native: the search is O(history x n-gram length) in the worst case.
#include <cstdint>

extern "C" {

// tokens[0..n): prompt + output so far.  For L = max_n down to min_n, find the most recent earlier
// occurrence of the last L tokens that has at least one token after it, and copy up to k following
// tokens to out.  Returns the number of draft tokens (0 = no match).
__attribute__((visibility("default")))
int vdamd_ngram_propose(const int32_t* tokens, long n, int max_n, int min_n, int k, int32_t* out) {
  if (k <= 0 || min_n < 1) return 0;
  for (int L = max_n; L >= min_n; --L) {
    if (n < L + 1) continue;
    const int32_t* suffix = tokens + (n - L);
    for (long i = n - L - 1; i >= 0; --i) {
      bool same = tokens[i + L - 1] == suffix[L - 1];       // last token first: the cheap reject
      for (int j = 0; same && j < L - 1; ++j) same = tokens[i + j] == suffix[j];
      if (!same) continue;
      const long from = i + L;
      int m = 0;
      for (; m < k && from + m < n; ++m) out[m] = tokens[from + m];
      if (m > 0) return m;
    }
  }
  return 0;
}

}  // extern "C"
