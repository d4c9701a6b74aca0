// ORACLE (test infrastructure only -- never linked into or called by the product path).
// CPU restatement of Lancet2's sequence-complexity annotation:
//   base/longdust_scorer.h      LongdustQScorer (k-mer count concentration, GC-corrected null model)
//   base/sequence_complexity.*  SequenceComplexityScorer (11 features) + MergeMax
//   core/variant_annotator.cpp  AnnotateSequenceComplexity / AnnotateGraphComplexity
// Pinned by tests/test_oracle_kat.py against the known answers of the reference's own tests
// (tests/base/sequence_complexity_test.cpp, tests/base/longdust_scorer_test.cpp: the cases that need no
// genome file) and against an independent closed form for homopolymers.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

#include "oracle.hpp"

namespace orc {

// ---- LongdustQScorer (base/longdust_scorer.h:217-462) -------------------------------------
namespace {
// longdust_scorer.h:350-389: expected sum log(c!) of one Poisson(lambda) count
f64 ComputeFSingle(f64 lambda) {
  if (lambda < 1e-10) return 0.0;
  if (lambda >= 30.0) {
    f64 const inv = 1.0 / lambda;
    f64 const pi = 3.141592653589793238462643383279502884, e = 2.718281828459045235360287471352662498;
    f64 const stirling = (0.5 * std::log(2.0 * pi * e * lambda)) -
                         (inv / 12.0 * (1.0 + (0.5 * inv) + (19.0 / 30.0 * inv * inv)));
    return stirling + (lambda * (std::log(lambda) - 1.0));
  }
  f64 accum = 0.0, sum_n = 0.0, scaled = lambda;
  for (int count = 2; count <= 10000; ++count) {
    sum_n += std::log(static_cast<f64>(count));
    scaled *= lambda / count;
    f64 const z = scaled * sum_n;
    if (z < accum * 1e-9) break;
    accum += z;
  }
  return accum * std::exp(-lambda);
}

u8 DnaCode(char c) {  // longdust_scorer.h:190-208
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 4;
  }
}
}  // namespace

LongdustQ::LongdustQ(int kmer_len, int max_len, f64 gc_frac)
    : gc(std::clamp(gc_frac, 0.0, 1.0)), k(kmer_len), mask((1U << (2 * kmer_len)) - 1), num_kmers(1U << (2 * kmer_len)) {
  F.assign(static_cast<usize>(max_len) + 1, 0.0);  // longdust_scorer.h:446-451
  for (int ell = 1; ell <= max_len; ++ell) F[ell] = ComputeF(ell);
}

f64 LongdustQ::ComputeF(int ell) const {  // longdust_scorer.h:399-440
  if (std::abs(gc - 0.5) < 1e-6) {
    f64 const lambda = static_cast<f64>(ell) / num_kmers;
    return static_cast<f64>(num_kmers) * ComputeFSingle(lambda);
  }
  f64 const safe_gc = std::clamp(gc, 1e-6, 1.0 - 1e-6);
  f64 const p_gc = safe_gc / 2.0, p_at = (1.0 - safe_gc) / 2.0;
  f64 const two_pow_k = static_cast<f64>(1ULL << k);
  f64 total = 0.0;
  for (int g = 0; g <= k; ++g) {
    f64 comb = 1.0;
    for (int j = 1; j <= g; ++j) comb *= static_cast<f64>(k - j + 1) / static_cast<f64>(j);
    f64 const n_kmers = comb * two_pow_k;
    f64 const prob = std::pow(p_gc, g) * std::pow(p_at, k - g);
    f64 const lambda = static_cast<f64>(ell) * prob;
    total += n_kmers * ComputeFSingle(lambda);
  }
  return total;
}

f64 LongdustQ::ScoreOneStrand(std::string_view seq) const {  // longdust_scorer.h:266-330
  i64 const n_positions = static_cast<i64>(seq.size()) - k + 1;
  if (n_positions <= 0) return 0.0;
  std::vector<u16> counts(num_kmers, 0);
  u32 kmer = 0;
  int run = 0;
  usize valid = 0;
  for (char const ch : seq) {
    u8 const b = DnaCode(ch);
    if (b < 4) {
      kmer = ((kmer << 2) | b) & mask;
      if (++run >= k) {
        counts[kmer]++;
        valid++;
      }
    } else {
      run = 0;
    }
  }
  if (valid == 0) return 0.0;
  f64 sum = 0.0;
  for (u32 i = 0; i < num_kmers; ++i)
    if (counts[i] >= 2) sum += std::lgamma(static_cast<f64>(counts[i] + 1));
  int const ell = static_cast<int>(valid);
  f64 const f = (static_cast<usize>(ell) < F.size()) ? F[ell] : ComputeF(ell);
  f64 const q = sum - f;
  return std::max(0.0, q / static_cast<f64>(valid));
}

f64 LongdustQ::Score(std::string_view seq) const {  // longdust_scorer.h:246-250
  f64 const fwd = ScoreOneStrand(seq);
  f64 const rev = ScoreOneStrand(RevComp(seq));
  return std::max(fwd, rev);
}

// ---- SequenceComplexityScorer (base/sequence_complexity.cpp) ------------------------------
std::string_view ExtractFlank(std::string_view hap, usize pos, usize len, i64 flank) {  // :31-41
  i64 const hl = static_cast<i64>(hap.size());
  i64 const start = std::max<i64>(0, static_cast<i64>(pos) - flank);
  i64 const end = std::min(hl, static_cast<i64>(pos + len) + flank);
  if (start >= end) return {};
  return hap.substr(static_cast<usize>(start), static_cast<usize>(end - start));
}

i32 MaxHomopolymerRun(std::string_view s) {  // :47-62
  if (s.empty()) return 0;
  i32 best = 1, cur = 1;
  for (usize i = 1; i < s.size(); ++i) {
    if (s[i] == s[i - 1]) {
      cur++;
      best = std::max(best, cur);
    } else {
      cur = 1;
    }
  }
  return best;
}

f32 LocalShannonEntropy(std::string_view s) {  // :75-120
  if (s.empty()) return 0.0F;
  usize cnt[4] = {0, 0, 0, 0};
  for (char const c : s) {
    u8 const b = DnaCode(c);
    if (b < 4) cnt[b]++;
  }
  f32 const total = static_cast<f32>(cnt[0] + cnt[1] + cnt[2] + cnt[3]);
  if (total <= 0.0F) return 0.0F;
  f32 h = 0.0F;
  for (usize const c : cnt) {
    if (c == 0) continue;
    f32 const freq = static_cast<f32>(c) / total;
    h -= freq * std::log2(freq);
  }
  return h;
}

namespace {
bool IsPrimitiveMotif(std::string_view m) {  // :130-148
  i32 const len = static_cast<i32>(m.size());
  for (i32 p = 1; p < len; ++p) {
    if (len % p != 0) continue;
    bool all = true;
    for (i32 i = p; i < len; ++i)
      if (m[i] != m[i % p]) {
        all = false;
        break;
      }
    if (all) return false;
  }
  return true;
}
}  // namespace

std::vector<TandemRepeat> FindExactRepeats(std::string_view s, i32 max_period, f32 min_copies) {  // :188-236
  std::vector<TandemRepeat> out;
  i32 const n = static_cast<i32>(s.size());
  for (i32 p = 1; p <= max_period && p <= n; ++p) {
    for (i32 start = 0; start <= n - p; ++start) {
      auto const motif = s.substr(start, p);
      if (p > 1 && !IsPrimitiveMotif(motif)) continue;
      i32 match = p;
      while (start + match + p <= n) {
        bool same = true;
        for (i32 j = 0; j < p; ++j)
          if (s[start + match + j] != motif[j]) {
            same = false;
            break;
          }
        if (!same) break;
        match += p;
      }
      i32 partial = 0;
      while (start + match + partial < n && partial < p && s[start + match + partial] == motif[partial]) partial++;
      f32 const copies = static_cast<f32>(match + partial) / static_cast<f32>(p);
      if (copies >= min_copies) {
        out.push_back({p, copies, start, match + partial, 0, true});
        start += match - 1;
      }
    }
  }
  return out;
}

std::vector<TandemRepeat> FindApproxRepeats(std::string_view s, i32 max_period, f32 min_copies, i32 max_edits) {  // :246-291
  std::vector<TandemRepeat> out;
  i32 const n = static_cast<i32>(s.size());
  for (i32 p = 1; p <= max_period && p <= n; ++p) {
    for (i32 start = 0; start <= n - p; ++start) {
      auto const motif = s.substr(start, p);
      if (p > 1 && !IsPrimitiveMotif(motif)) continue;
      i32 span = p, errors = 0;
      while (start + span + p <= n) {
        i32 ue = 0;
        for (i32 j = 0; j < p; ++j)
          if (s[start + span + j] != motif[j]) ++ue;
        if (ue > max_edits) break;
        errors += ue;
        span += p;
      }
      f32 const copies = static_cast<f32>(span) / static_cast<f32>(p);
      f32 const purity = span > 0 ? 1.0F - (static_cast<f32>(errors) / static_cast<f32>(span)) : 0.0F;
      if (copies >= min_copies && purity >= 0.75F) {
        out.push_back({p, copies, start, span, errors, false});
        start += span - 1;
      }
    }
  }
  return out;
}

TrFeatures FlattenTRFeatures(const std::vector<TandemRepeat>& rs, i32 vpos, i32 vlen) {  // :300-339
  TrFeatures f;
  if (rs.empty()) return f;
  i32 nearest = std::numeric_limits<i32>::max();
  i32 const vend = vpos + vlen;
  for (auto const& t : rs) {
    i32 const tend = t.start + t.span;
    i32 dist = 0;
    if (vpos >= t.start && vpos < tend) dist = 0;
    else if (vpos < t.start) dist = t.start - vend;
    else dist = vpos - tend;
    dist = std::max(0, dist);
    if (dist < nearest) {
      nearest = dist;
      f.dist = dist;
      f.period = t.period;
      f.purity = t.Purity();
    }
    if (dist <= 1 && vlen > 0 && vlen <= t.period) f.stutter = 1;
  }
  return f;
}

SeqCxScorer::SeqCxScorer(f64 gc_frac) : flank(4, 1024, gc_frac), hap(7, 4096, gc_frac) {}  // :23-25

SeqCx SeqCxScorer::Score(const HapRegion& ref, const HapRegion& alt) const {  // :373-380
  SeqCx c;
  // ScoreContext (:388-401)
  auto const ctx = ExtractFlank(ref.hap, ref.pos, ref.len, 20);
  c.ctx_hrun = MaxHomopolymerRun(ctx);
  c.ctx_entropy = LocalShannonEntropy(ctx);
  c.ctx_flank_lq = std::log1p(std::max(0.0, flank.Score(ExtractFlank(ref.hap, ref.pos, ref.len, 50))));
  c.ctx_hap_lq = std::log1p(std::max(0.0, hap.Score(ref.hap)));
  // ScoreDeltas (:413-430)
  c.delta_hrun = MaxHomopolymerRun(ExtractFlank(alt.hap, alt.pos, alt.len, 5)) -
                 MaxHomopolymerRun(ExtractFlank(ref.hap, ref.pos, ref.len, 5));
  c.delta_entropy = LocalShannonEntropy(ExtractFlank(alt.hap, alt.pos, alt.len, 10)) -
                    LocalShannonEntropy(ExtractFlank(ref.hap, ref.pos, ref.len, 10));
  f64 const alt_lq = std::log1p(std::max(0.0, flank.Score(ExtractFlank(alt.hap, alt.pos, alt.len, 50))));
  c.delta_flank_lq = alt_lq - c.ctx_flank_lq;
  // ScoreTrMotif (:442-461) with AccumulateTRFeatures (:345-367) on a fresh feature set
  auto const win = ExtractFlank(alt.hap, alt.pos, alt.len, 50);
  i64 const start = std::max<i64>(0, static_cast<i64>(alt.pos) - 50);
  i32 const vpos = static_cast<i32>(static_cast<i64>(alt.pos) - start);
  auto all = FindExactRepeats(win, 6, 2.5F);
  auto const approx = FindApproxRepeats(win, 6, 3.0F, 1);
  all.insert(all.end(), approx.begin(), approx.end());
  TrFeatures const nf = FlattenTRFeatures(all, vpos, static_cast<i32>(alt.len));
  TrFeatures tr;
  if (nf.dist >= 0 && (tr.dist < 0 || nf.dist < tr.dist)) {
    tr.dist = nf.dist;
    tr.period = nf.period;
    tr.purity = nf.purity;
  }
  tr.stutter = std::max(tr.stutter, nf.stutter);
  if (tr.dist < 0) {
    c.tr_affinity = 0.0F;
    c.tr_purity = 0.0F;
    c.tr_period = 0;
  } else {
    c.tr_affinity = 1.0F / (1.0F + static_cast<f32>(tr.dist));
    c.tr_purity = tr.purity;
    c.tr_period = tr.period;
  }
  c.stutter = tr.stutter;
  return c;
}

void SeqCx::MergeMax(const SeqCx& o) {  // :489-507
  ctx_hrun = std::max(ctx_hrun, o.ctx_hrun);
  ctx_entropy = std::max(ctx_entropy, o.ctx_entropy);
  ctx_flank_lq = std::max(ctx_flank_lq, o.ctx_flank_lq);
  ctx_hap_lq = std::max(ctx_hap_lq, o.ctx_hap_lq);
  delta_hrun = std::max(delta_hrun, o.delta_hrun);
  delta_entropy = std::max(delta_entropy, o.delta_entropy);
  delta_flank_lq = std::max(delta_flank_lq, o.delta_flank_lq);
  tr_affinity = std::max(tr_affinity, o.tr_affinity);
  tr_purity = std::max(tr_purity, o.tr_purity);
  tr_period = std::max(tr_period, o.tr_period);
  stutter = std::max(stutter, o.stutter);
}

// core/variant_annotator.cpp:43-85.  `alts[a]` = (allele length, [(hap index, start on that hap)]).
SeqCx AnnotateVariant(const SeqCxScorer& sc, const std::vector<std::string_view>& haps, usize ref_pos, usize ref_len,
                      const std::vector<AltSites>& alts) {
  SeqCx merged;  // var.mSeqCx starts all-zero and only ever MergeMax'es (negative deltas clamp at 0)
  bool any = false;
  HapRegion const ref{haps[0], ref_pos, ref_len};
  for (auto const& al : alts) {
    usize const alt_len = std::max(ref_len, al.len);
    for (auto const& hs : al.hap_starts) {
      if (hs.first >= haps.size() || hs.first == 0) continue;
      merged.MergeMax(sc.Score(ref, {haps[hs.first], hs.second, alt_len}));
      any = true;
    }
  }
  if (!any) merged = sc.Score(ref, ref);
  return merged;
}

}  // namespace orc
