// ORACLE -- TEST INFRASTRUCTURE ONLY (see common.hpp header).
// Reference repeat gate, restated literally: pairwise Hamming distance over all k-mer
// pairs of the window (base/repeat.cpp:348-375), mismatch budget 2 for the k-cascade
// gate (cbdg/graph.h:127-131) and 0 for the max-k window skip
// (core/variant_builder.cpp:116-117).
#include "oracle.hpp"

#include <unordered_set>

namespace orc {

// base/repeat.cpp:219-332 (scalar meaning of the SIMD kernels): count differing bytes.
usize HammingDist(std::string_view a, std::string_view b) {
  usize d = 0;
  for (usize i = 0; i < a.size(); ++i) d += (a[i] != b[i]);
  return d;
}

// base/repeat.cpp:55-204: true iff the two equal-length strings differ in <= max_mm places.
static bool IsWithinHammingDist(const char* a, const char* b, usize len, usize max_mm) {
  usize d = 0;
  for (usize i = 0; i < len; ++i) {
    d += (a[i] != b[i]);
    if (d > max_mm) return false;
  }
  return true;
}

// base/repeat.cpp:348-371 over base/sliding.h:17-32 k-mer views of `seq`.
bool HasRepeat(std::string_view seq, usize k, usize max_mm) {
  if (seq.size() < k || k == 0) return false;  // SlidingView -> empty span
  usize const n = seq.size() - k + 1;
  if (max_mm == 0) {
    std::unordered_set<std::string_view> seen;
    seen.reserve(n);
    for (usize i = 0; i < n; ++i) {
      if (!seen.insert(seq.substr(i, k)).second) return true;
    }
    return false;
  }
  if (n < 2) return false;
  for (usize i = 0; i < n; ++i) {
    for (usize j = i + 1; j < n; ++j) {
      if (IsWithinHammingDist(seq.data() + i, seq.data() + j, k, max_mm)) return true;
    }
  }
  return false;
}

}  // namespace orc
