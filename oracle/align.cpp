// ORACLE -- TEST INFRASTRUCTURE ONLY (see common.hpp header).
// Read <-> haplotype aligner and the CIGAR-based allele-scoring epilogue.
//
// PARITY UNPINNED against minimap2 itself: the reference calls minimap2 2.30 (mm_map with
// k=11,w=5 seeds, a=1 b=4 q=12 e=3 single-affine, bw=10000, zdrop off, end_bonus=10000,
// best_n=1: caller/genotyper.cpp:89-191, :376-411); its source is not under /root/reference and
// no reference test pins its output.  What follows is the engine's CANONICAL restatement of that
// contract (DESIGN.md "read<->haplotype aligner"); its optimal SCORES are pinned to an
// independent brute force (tests/brute_force.py, tests/test_aligner_pins.py):
//   1. nt4 encoding; substitution +1 / -4, any ambiguous base -1 (minimap2 sc_ambi);
//      a gap of length L costs 12 + 3L.
//   2. Seeds: every exact 11-mer shared by read and haplotype lies on a diagonal d = hap_pos - read_pos and counts as
//      one vote for it.  minimap2 drops chains scoring < 40 (min_chain_score, not overridden by the reference), and
//      40 chained bases need at least 4 exact 11-mers: a seeded diagonal d is an ANCHOR iff the diagonals within reach
//      of it, [d - K, d + K] (K below), hold >= 4 votes together.  Stray single matches (a random 11-mer recurs in
//      3 % of the read x haplotype pairs of a 1 kb window) anchor nothing.  No anchor => no hit.
//   3. Search region.  The reference sets bw = 10000 and disables z-drop, i.e. the band never
//      limits an alignment that its seeds support.  With cost(P) = m - score(P) every row of the
//      read costs >= 0 (match 0, mismatch 5, ambiguous 2, clipped overhang row 1, inserted row 4)
//      and a gap that shifts the diagonal by s costs >= 12 + 3s.  A reported hit has
//      score >= min_score, i.e. cost <= m - min_score, so an alignment through a seed on diagonal d
//      never leaves [d - K, d + K] with K = max(0, floor((m - min_score - 12) / 3)).  The canonical
//      search region is therefore R = [vmin - K, vmax + K] over the extreme ANCHOR diagonals: no
//      free "band" parameter, nothing a seed-anchored hit could reach is excluded.
//   4. Overlap alignment inside R: the read is aligned end to end unless it overhangs a haplotype
//      end, in which case the overhang is soft-clipped (start cells (0,j) and (i,0); end cells
//      (m,j) and (i,n)).
//   5. Best end cell: max score, ties -> larger i, then smaller j.  Traceback: diagonal first, then
//      deletion (E), then insertion (F); inside a gap prefer opening over extending (left-aligned
//      gaps, ksw2's default).
//   6. A hit is reported iff score >= 80 (minimap2's default min_dp_max, which the reference
//      does not override).
// The scoring epilogue below IS first-party reference code and is restated literally.
#include "oracle.hpp"

#include <limits>

namespace orc {

// K of rule 3
i32 AlignReach(i32 m, i32 min_score) {
  i32 const c = m - min_score - 12;
  return c > 0 ? c / 3 : 0;
}

// rule 2: extreme ANCHOR diagonals; false when there is none
bool SeedDiagonals(const std::vector<u8>& q, const std::vector<u8>& t, i32 SK, i32 K, i32* vmin, i32* vmax) {
  i32 const m = static_cast<i32>(q.size()), n = static_cast<i32>(t.size());
  if (m < SK || n < SK) return false;
  auto code_at = [&](const std::vector<u8>& s, i32 p, u64* code) {
    u64 c = 0;
    for (i32 x = 0; x < SK; ++x) {
      if (s[p + x] > 3) return false;
      c = (c << 2) | s[p + x];
    }
    *code = c;
    return true;
  };
  std::unordered_map<u64, std::vector<i32>> index;
  for (i32 j = 0; j + SK <= n; ++j) {
    u64 c;
    if (code_at(t, j, &c)) index[c].push_back(j);
  }
  std::vector<i32> votes(static_cast<usize>(m + n + 1), 0);  // diagonal d -> votes[d + m]
  for (i32 i = 0; i + SK <= m; ++i) {
    u64 c;
    if (!code_at(q, i, &c)) continue;
    auto it = index.find(c);
    if (it == index.end()) continue;
    for (i32 j : it->second) votes[static_cast<usize>(j - i + m)]++;
  }
  constexpr i32 kMinChainVotes = 4;
  bool any = false;
  i32 const nd = m + n + 1;
  for (i32 x = 0; x < nd; ++x) {
    if (votes[static_cast<usize>(x)] == 0) continue;
    i32 sum = 0;
    for (i32 y = std::max(0, x - K); y <= std::min(nd - 1, x + K); ++y) sum += votes[static_cast<usize>(y)];
    if (sum < kMinChainVotes) continue;
    i32 const d = x - m;
    if (!any) *vmin = d;
    *vmax = d;
    any = true;
  }
  return any;
}

AlnResult AlignReadToHap(std::string_view read, std::string_view hap, const AlignParams& ap) {
  AlnResult res;
  i32 const m = static_cast<i32>(read.size()), n = static_cast<i32>(hap.size());
  std::vector<u8> q(m), t(n);
  for (i32 i = 0; i < m; ++i) q[i] = EncodeBase(read[i]);
  for (i32 j = 0; j < n; ++j) t[j] = EncodeBase(hap[j]);
  i32 vmin = 0, vmax = 0;
  i32 const K = AlignReach(m, ap.min_score);
  if (!SeedDiagonals(q, t, ap.seed_k, K, &vmin, &vmax)) return res;
  // region R: diagonals lo .. hi; cells outside the matrix are skipped anyway, so clamp to it
  i32 const lo = std::max(vmin - K, -m), hi = std::min(vmax + K, n);

  // --- overlap DP inside R ---
  i32 const GO = 12, GE = 3;
  i32 const NEG = -(1 << 28);
  i32 const Wd = hi - lo + 1;  // positions tt = j - i - lo
  auto IDX = [&](i32 i, i32 tt) { return static_cast<usize>(i) * Wd + tt; };
  std::vector<i32> H(static_cast<usize>(m + 1) * Wd, NEG), E(H.size(), NEG), F(H.size(), NEG);
  auto in_band = [&](i32 i, i32 j) {
    i32 const tt = j - i - lo;
    return tt >= 0 && tt < Wd && j >= 0 && j <= n;
  };
  auto getH = [&](i32 i, i32 j) { return in_band(i, j) ? H[IDX(i, j - i - lo)] : NEG; };
  auto getE = [&](i32 i, i32 j) { return in_band(i, j) ? E[IDX(i, j - i - lo)] : NEG; };
  auto getF = [&](i32 i, i32 j) { return in_band(i, j) ? F[IDX(i, j - i - lo)] : NEG; };
  auto sub = [&](i32 i, i32 j) -> i32 {  // 1-based cell (i,j) pairs q[i-1], t[j-1]
    u8 const a = q[i - 1], b = t[j - 1];
    if (a > 3 || b > 3) return -1;
    return a == b ? 1 : -4;
  };
  for (i32 i = 0; i <= m; ++i) {
    for (i32 tt = 0; tt < Wd; ++tt) {
      i32 const j = i + lo + tt;
      if (j < 0 || j > n) continue;
      i32 h, e = NEG, f = NEG;
      if (i == 0 || j == 0) {
        h = 0;  // free start: read begins inside the haplotype, or overhangs its left end
      } else {
        e = std::max(getH(i, j - 1) - (GO + GE), getE(i, j - 1) - GE);
        f = std::max(getH(i - 1, j) - (GO + GE), getF(i - 1, j) - GE);
        h = std::max(getH(i - 1, j - 1) + sub(i, j), std::max(e, f));
      }
      H[IDX(i, tt)] = h;
      E[IDX(i, tt)] = e;
      F[IDX(i, tt)] = f;
    }
  }
  // --- best end cell: (m, j) any j, or (i, n) any i ---
  i32 best = NEG, bi = -1, bj = -1;
  auto consider = [&](i32 i, i32 j) {
    if (!in_band(i, j)) return;
    i32 const h = getH(i, j);
    if (h > best || (h == best && (i > bi || (i == bi && j < bj)))) {
      best = h;
      bi = i;
      bj = j;
    }
  };
  for (i32 j = 0; j <= n; ++j) consider(m, j);
  for (i32 i = 0; i < m; ++i) consider(i, n);
  if (bi < 0 || best < ap.min_score) return res;

  // --- traceback ---
  std::vector<char> ops;  // reversed
  i32 i = bi, j = bj;
  int state = 0;  // 0 = H, 1 = E (deletion run), 2 = F (insertion run)
  while (true) {
    if (state == 0) {
      if (i == 0 || j == 0) break;
      i32 const h = getH(i, j);
      if (h == getH(i - 1, j - 1) + sub(i, j)) {
        ops.push_back('M');
        --i;
        --j;
      } else if (h == getE(i, j)) {
        state = 1;
      } else {
        state = 2;
      }
    } else if (state == 1) {
      i32 const e = getE(i, j);
      ops.push_back('D');
      bool const open = e == getH(i, j - 1) - (GO + GE);
      --j;
      if (open) state = 0;
    } else {
      i32 const f = getF(i, j);
      ops.push_back('I');
      bool const open = f == getH(i - 1, j) - (GO + GE);
      --i;
      if (open) state = 0;
    }
  }
  res.hit = true;
  res.score = best;
  res.qs = i;
  res.rs = j;
  res.qe = bi;
  res.re = bj;
  // genotyper.cpp:45-69 BuildCigar: S(qs) + core + S(qlen - qe)
  if (res.qs > 0) res.cigar.push_back({'S', static_cast<u32>(res.qs)});
  for (auto it = ops.rbegin(); it != ops.rend(); ++it) {
    if (!res.cigar.empty() && res.cigar.back().op == *it && *it != 'S') res.cigar.back().len++;
    else res.cigar.push_back({*it, 1});
  }
  if (res.qe < m) res.cigar.push_back({'S', static_cast<u32>(m - res.qe)});
  return res;
}

// caller/genotype_likelihood.cpp:20-79 (constants), :93-111 (LogDirichletMultinomial), :113-128 (NormalizeToPLs),
// :205-248 (ComputeGenotypePLs).  FORMAT PL; the germline site QUAL is PL[0/0] (variant_call.cpp:289-303).
std::vector<u32> ComputeGenotypePLs(const std::vector<int>& allele_counts) {
  constexpr f64 kBackgroundError = 0.005, kOverdispersion = 0.01, kAlphaFloor = 1e-6;
  int const K = static_cast<int>(allele_counts.size());
  if (K == 0) return {};
  f64 const precision = (1.0 - kOverdispersion) / kOverdispersion;
  std::vector<f64> lls;
  for (int b = 0; b < K; ++b) {
    for (int a = 0; a <= b; ++a) {
      std::vector<f64> mu(static_cast<usize>(K), kBackgroundError / K);
      f64 const main_mass = 1.0 - kBackgroundError;
      if (a == b) {
        mu[static_cast<usize>(a)] += main_mass;
      } else {
        mu[static_cast<usize>(a)] += main_mass / 2.0;
        mu[static_cast<usize>(b)] += main_mass / 2.0;
      }
      f64 log_prob = 0.0, alpha_sum = 0.0, count_alpha_sum = 0.0;
      for (int k = 0; k < K; ++k) {
        f64 const alpha = std::max(kAlphaFloor, mu[static_cast<usize>(k)] * precision);
        f64 const cnt = static_cast<f64>(allele_counts[static_cast<usize>(k)]);
        log_prob += std::lgamma(cnt + alpha) - std::lgamma(alpha);
        alpha_sum += alpha;
        count_alpha_sum += cnt + alpha;
      }
      log_prob += std::lgamma(alpha_sum) - std::lgamma(count_alpha_sum);
      lls.push_back(log_prob);
    }
  }
  f64 const best = *std::max_element(lls.begin(), lls.end());
  f64 const cap = static_cast<f64>(std::numeric_limits<u32>::max()) / 2.0;
  f64 const ln_ten = 2.302585092994045684017991454684364208;  // std::numbers::ln10
  std::vector<u32> pls(lls.size());
  for (usize i = 0; i < lls.size(); ++i) {
    f64 const raw = -10.0 * (lls[i] - best) / ln_ten;
    pls[i] = static_cast<u32>(std::round(std::min(raw, cap)));
  }
  return pls;
}

// caller/genotype_likelihood.cpp:250-272: second-smallest PL, capped at 99
u32 ComputeGenotypeQuality(const std::vector<u32>& pls) {
  if (pls.size() < 2) return 0;
  u32 min1 = std::numeric_limits<u32>::max(), min2 = min1;
  for (u32 v : pls) {
    if (v < min1) {
      min2 = min1;
      min1 = v;
    } else if (v < min2) {
      min2 = v;
    }
  }
  return std::min<u32>(min2 - min1, 99u);
}

// hts/cigar_utils.h:48-111
u32 ComputeEditDistance(const std::vector<CigarUnit>& cigar, const std::vector<u8>& q, const u8* t,
                        usize tlen) {
  u32 ed = 0;
  usize qp = 0, tp = 0;
  for (auto const& u : cigar) {
    switch (u.op) {
      case 'M':
        for (u32 i = 0; i < u.len; ++i, ++qp, ++tp)
          if (qp < q.size() && tp < tlen && q[qp] != t[tp]) ++ed;
        break;
      case '=': qp += u.len; tp += u.len; break;
      case 'X': ed += u.len; qp += u.len; tp += u.len; break;
      case 'I': ed += u.len; qp += u.len; break;
      case 'D': ed += u.len; tp += u.len; break;
      case 'S': qp += u.len; break;
      case 'N': tp += u.len; break;
      default: break;
    }
  }
  return ed;
}

// hts/cigar_utils.h:113-139
usize CigarRefPosToQueryPos(const std::vector<CigarUnit>& cigar, usize ref_pos) {
  usize qp = 0, tp = 0;
  for (auto const& u : cigar) {
    switch (u.op) {
      case 'M': case '=': case 'X':
        for (u32 i = 0; i < u.len; ++i, ++qp, ++tp)
          if (tp == ref_pos) return qp;
        break;
      case 'I': qp += u.len; break;
      case 'D': case 'N':
        for (u32 i = 0; i < u.len; ++i, ++tp)
          if (tp == ref_pos) return qp;
        break;
      case 'S': qp += u.len; break;
      default: break;
    }
  }
  return qp;
}

namespace {

constexpr i8 kMatrix[25] = {1, -4, -4, -4, 0, -4, 1, -4, -4, 0, -4, -4, 1, -4, 0,
                            -4, -4, -4, 1, 0, 0, 0, 0, 0, 0};  // scoring_constants.h:35-41

struct LocalScore { f64 pbq = 0, raw = 0, identity = 0; u8 bq = 0; };

bool ConsumesRef(char op) { return op == 'M' || op == 'D' || op == 'N' || op == '=' || op == 'X'; }

// caller/local_scorer.cpp:166-279
LocalScore ComputeLocalScore(const std::vector<CigarUnit>& cigar, const std::vector<u8>& qry,
                             const u8* target, usize tlen, const u8* quals, usize nquals,
                             i32 aln_start, i32 var_start, i32 var_len) {
  LocalScore out;
  if (cigar.empty() || var_len == 0) return out;
  i32 const var_end = var_start + var_len;
  f64 pbq = 0.0, raw = 0.0;
  usize matches = 0, aligned = 0;
  u8 min_bq = 255;
  auto in_region = [&](i32 tp) {
    i32 const a = aln_start + tp;
    return a >= var_start && a < var_end;
  };
  auto track_bq = [&](usize qp) {
    if (qp < nquals) min_bq = std::min(min_bq, quals[qp]);
  };
  i32 tpos = 0;
  usize qpos = 0;
  for (auto const& u : cigar) {
    if (aln_start + tpos >= var_end && ConsumesRef(u.op)) break;
    switch (u.op) {
      case 'M': case '=': case 'X':
        for (u32 i = 0; i < u.len; ++i, ++tpos, ++qpos) {
          if (!in_region(tpos)) continue;
          ++aligned;
          if (!(qpos >= qry.size() || static_cast<usize>(tpos) >= tlen)) {
            i8 const r = kMatrix[target[tpos] * 5 + qry[qpos]];
            raw += static_cast<f64>(r);
            f64 const w = qpos < nquals ? 1.0 - PhredToErrorProb(quals[qpos]) : 1.0;
            pbq += static_cast<f64>(r) * w;
            matches += (qry[qpos] == target[tpos]);
          }
          track_bq(qpos);
        }
        break;
      case 'I': {
        bool const inr = in_region(tpos);
        for (u32 i = 0; i < u.len; ++i, ++qpos) {
          if (!inr) continue;
          ++aligned;
          track_bq(qpos);
          pbq += 3.0;  // + SCORING_GAP_EXTEND (local_scorer.cpp:228)
        }
        break;
      }
      case 'D':
        for (u32 i = 0; i < u.len; ++i, ++tpos)
          if (in_region(tpos)) {
            ++aligned;
            pbq += 3.0;
          }
        if (qpos > 0 && qpos - 1 < nquals) min_bq = std::min(min_bq, quals[qpos - 1]);
        if (qpos < nquals) min_bq = std::min(min_bq, quals[qpos]);
        break;
      case 'S': qpos += u.len; break;
      case 'N': tpos += static_cast<i32>(u.len); break;
      default: break;
    }
  }
  out.pbq = pbq;
  out.raw = raw;
  out.identity = aligned > 0 ? static_cast<f64>(matches) / static_cast<f64>(aligned) : 0.0;
  out.bq = min_bq == 255 ? 0 : min_bq;
  return out;
}

// caller/local_scorer.cpp:290-305
f64 SoftClipPenalty(const std::vector<CigarUnit>& c) {
  if (c.empty()) return 0.0;
  i32 const s5 = c.front().op == 'S' ? static_cast<i32>(c.front().len) : 0;
  i32 const s3 = (c.size() > 1 && c.back().op == 'S') ? static_cast<i32>(c.back().len) : 0;
  return static_cast<f64>(s5 + s3) * 4;
}

}  // namespace

// caller/genotyper.cpp:269-362 + caller/combined_scorer.cpp:24-108
std::vector<Assignment> AssignReadToAlleles(const Read& rd, const std::vector<std::string>& haps,
                                            const std::vector<RawVariant>& vars, const AlignParams& ap,
                                            std::vector<AlnResult>* alns_out) {
  std::vector<Assignment> out(vars.size());
  std::vector<AlnResult> alns;
  for (usize h = 0; h < haps.size(); ++h) {  // AlignToAllHaplotypes (genotyper.cpp:376-411)
    AlnResult a = AlignReadToHap(rd.seq, haps[h], ap);
    if (!a.hit) continue;
    a.hap = static_cast<u32>(h);
    alns.push_back(std::move(a));
  }
  if (alns_out) *alns_out = alns;
  if (alns.empty()) return out;
  std::vector<u8> qenc(rd.seq.size());
  for (usize i = 0; i < rd.seq.size(); ++i) qenc[i] = EncodeBase(rd.seq[i]);
  std::vector<std::vector<u8>> henc(haps.size());
  for (usize h = 0; h < haps.size(); ++h) {
    henc[h].resize(haps[h].size());
    for (usize i = 0; i < haps[h].size(); ++i) henc[h][i] = EncodeBase(haps[h][i]);
  }
  usize const rlen = rd.seq.size();
  // combined_scorer.cpp:24-38: NM against the REF haplotype; read length when no REF alignment
  u32 ref_nm = static_cast<u32>(rlen);
  for (auto const& a : alns) {
    if (a.hap != 0 || a.rs >= a.re) continue;
    ref_nm = ComputeEditDistance(a.cigar, qenc, henc[0].data() + a.rs, static_cast<usize>(a.re - a.rs));
    break;
  }
  for (auto const& a : alns) {
    for (usize v = 0; v < vars.size(); ++v) {
      auto const& var = vars[v];
      // ExtractHapBounds (genotyper.cpp:329-352)
      i32 vstart = 0, vlen = 0;
      u32 allele = 0;
      bool have = false;
      if (a.hap == 0) {
        vstart = static_cast<i32>(var.ref_start0);
        vlen = static_cast<i32>(var.ref.size());
        allele = 0;
        have = true;
      } else {
        for (usize ai = 0; ai < var.alts.size() && !have; ++ai)
          for (auto const& hs : var.alts[ai].hap_starts)
            if (hs.first == a.hap) {
              vstart = static_cast<i32>(hs.second);
              vlen = static_cast<i32>(var.alts[ai].seq.size());
              allele = static_cast<u32>(ai + 1);
              have = true;
              break;
            }
      }
      if (!have) continue;
      if (!((vstart + vlen) > a.rs && vstart < a.re)) continue;  // OverlapsAlignment :360-362
      // ScoreReadAtVariant (combined_scorer.cpp:60-108)
      usize const alen = static_cast<usize>(a.re - a.rs);
      const u8* target = henc[a.hap].data() + a.rs;
      LocalScore const ls = ComputeLocalScore(a.cigar, qenc, target, alen, rd.qual, rlen, a.rs, vstart, vlen);
      f64 const global_adj = static_cast<f64>(a.score) - SoftClipPenalty(a.cigar);
      Assignment s;
      s.valid = true;
      s.allele = allele;
      s.global_score = static_cast<i32>(global_adj - ls.raw);
      s.local_score = ls.pbq;
      s.local_identity = ls.identity;
      s.base_qual = ls.bq;
      s.hap_id = a.hap;
      s.own_nm = ComputeEditDistance(a.cigar, qenc, target, alen);
      usize vsa = 0;
      if (vstart > a.rs) vsa = static_cast<usize>(vstart - a.rs);
      usize const qp = CigarRefPosToQueryPos(a.cigar, vsa);
      f64 const rel = rlen > 0 ? static_cast<f64>(qp) / static_cast<f64>(rlen) : 0.5;
      s.folded_pos = std::min(rel, 1.0 - rel);
      s.ref_nm = ref_nm;
      if (out[v].valid && s.Combined() <= out[v].Combined()) continue;  // first wins ties
      out[v] = s;
    }
  }
  return out;
}

}  // namespace orc
