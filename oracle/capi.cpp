// ORACLE -- TEST INFRASTRUCTURE ONLY (see common.hpp header).
// Flat C entry points (host pointers only) with the SAME structs and output layout as the
// product's include/microasm.h, so that parity tests compare buffer for buffer.  Prefix orc_.
#include "../include/microasm.h"
#include "oracle.hpp"

#include <map>
#include <set>

using namespace orc;

namespace {

Params ToParams(const ma_params_t* p) {
  Params q;
  q.min_k = p->min_k;
  q.max_k = p->max_k;
  q.k_step = p->k_step;
  q.min_node_cov = p->min_node_cov;
  q.min_anchor_cov = p->min_anchor_cov;
  q.num_samples = p->num_samples;
  q.min_anchor_len = p->min_anchor_len;
  q.max_mismatch = p->max_mismatch;
  q.bfs_limit = p->bfs_limit;
  return q;
}

std::vector<Read> WindowReads(const ma_batch_t* b, int w) {
  std::vector<Read> reads;
  for (u32 r = b->read_win_off[w]; r < b->read_win_off[w + 1]; ++r) {
    Read rd;
    u64 const o = b->read_off[r], e = b->read_off[r + 1];
    rd.seq = std::string_view(reinterpret_cast<const char*>(b->read_bases) + o, e - o);
    rd.qual = b->read_quals + o;
    rd.qname_id = b->read_qname_id[r];
    rd.sample = b->read_sample[r];
    u8 const f = b->read_flags[r];
    rd.role = (f & MA_RF_CASE) ? 1 : 0;
    rd.pass = (f & MA_RF_PASS) != 0;
    rd.rev = (f & MA_RF_REV) != 0;
    reads.push_back(rd);
  }
  return reads;
}

std::string_view WindowRef(const ma_batch_t* b, int w) {
  return std::string_view(reinterpret_cast<const char*>(b->ref_bases) + b->ref_off[w],
                          b->ref_off[w + 1] - b->ref_off[w]);
}

// largest k with HasRepeat(k, mm) (monotone in k: a k-window with <= mm mismatches contains a
// (k-1)-window with <= mm); literal HasRepeat evaluations only.
u32 MaxRepeatLen(std::string_view ref, usize mm) {
  usize lo = 0, hi = ref.size() > 0 ? ref.size() - 1 : 0;  // HasRepeat(0) treated as trivially true
  while (lo < hi) {
    usize const mid = (lo + hi + 1) / 2;
    if (HasRepeat(ref, mid, mm)) lo = mid; else hi = mid - 1;
  }
  return static_cast<u32>(lo);
}

}  // namespace

extern "C" {

void orc_default_params(ma_params_t* p) {
  std::memset(p, 0, sizeof(*p));
  p->min_k = 13; p->max_k = 127; p->k_step = 6;
  p->min_node_cov = 2; p->min_anchor_cov = 5; p->num_samples = 2;
  p->min_anchor_len = 150; p->max_mismatch = 2; p->bfs_limit = 1 << 20;
  p->aln_tier = 0; p->min_aln_score = 80;
  p->max_comps = 4; p->max_haps = 16; p->max_hap_len = 2048; p->max_runs = 256;
  p->max_vars = 64; p->max_alts = 4; p->max_allele_bytes = 4096; p->max_cigar = 16;
  p->case_ctrl_mode = 1;
}

// FORMAT PL / GQ of one sample from its allele depths (caller/genotype_likelihood.cpp:205-272); returns the PL count
int orc_genotype_pls(const int32_t* counts, int k, uint32_t* pls, uint32_t* gq) {
  std::vector<int> c(counts, counts + k);
  auto const out = ComputeGenotypePLs(c);
  for (usize i = 0; i < out.size(); ++i) pls[i] = out[i];
  *gq = ComputeGenotypeQuality(out);
  return static_cast<int>(out.size());
}

// control-flow event counters of the assembler since the last call (see graph.cpp); reset on read
void orc_debug_counters(unsigned long long* out) {
  for (int i = 0; i < 4; ++i) {
    out[i] = g_debug_counters[i];
    g_debug_counters[i] = 0;
  }
}

// ---- small known-answer hooks (tests/test_oracle_kat.py) ----
uint64_t orc_hamming(const char* a, const char* b, uint64_t n) {
  return HammingDist(std::string_view(a, n), std::string_view(b, n));
}
int orc_has_repeat(const char* seq, uint64_t n, uint64_t k, uint64_t mm) {
  return HasRepeat(std::string_view(seq, n), k, mm) ? 1 : 0;
}
uint64_t orc_hash64(const char* s, uint64_t n) { return HashStr64(std::string_view(s, n)); }
void orc_revcomp(const char* s, uint64_t n, char* out) {
  std::string r = RevComp(std::string_view(s, n));
  std::memcpy(out, r.data(), n);
}
double orc_phred(int q) { return PhredToErrorProb(static_cast<u8>(q)); }
uint32_t orc_median_u32(const uint32_t* v, uint64_t n) { return Median(std::vector<u32>(v, v + n)); }
void orc_online_stats(const double* v, uint64_t n, double* mean, double* var, double* sd) {
  OnlineStats s;
  for (uint64_t i = 0; i < n; ++i) s.Add(v[i]);
  *mean = s.Mean(); *var = s.Variance(); *sd = s.StdDev();
}
// CIGAR given as parallel arrays (op chars, lengths)
uint32_t orc_edit_distance(const char* ops, const uint32_t* lens, int n, const uint8_t* q, int qn,
                           const uint8_t* t, int tn) {
  std::vector<CigarUnit> c;
  for (int i = 0; i < n; ++i) c.push_back({ops[i], lens[i]});
  return ComputeEditDistance(c, std::vector<u8>(q, q + qn), t, static_cast<usize>(tn));
}
uint64_t orc_refpos_to_qpos(const char* ops, const uint32_t* lens, int n, uint64_t ref_pos) {
  std::vector<CigarUnit> c;
  for (int i = 0; i < n; ++i) c.push_back({ops[i], lens[i]});
  return CigarRefPosToQueryPos(c, ref_pos);
}

// POA + variant extraction on NUL-separated sequences with unit weights and an arbitrary engine
// (tests/caller/variant_set_test.cpp uses Create(kNW, 3, -5, -3)).  Output: text, one line per
// variant "pos1\tref\talt1,alt2\thaps(alt1);haps(alt2)\n".
int orc_poa_variants(const char* seqs_nul, int nseq, int m, int n, int g, int e, int q, int c,
                     uint64_t anchor_pos1, char* out, int out_cap) {
  std::vector<std::string> seqs;
  const char* p = seqs_nul;
  for (int i = 0; i < nseq; ++i) {
    seqs.emplace_back(p);
    p += seqs.back().size() + 1;
  }
  PoaScoring sc{m, n, g, e, q, c};
  PoaGraph gr;
  std::vector<std::string_view> views(seqs.begin(), seqs.end());
  std::vector<std::vector<u32>> ws;
  for (auto const& s : seqs) ws.emplace_back(s.size(), 1u);
  UpdateSpoaState(gr, sc, views, ws);
  auto const vars = ExtractVariants(gr, anchor_pos1);
  std::string txt;
  for (auto const& v : vars) {
    txt += std::to_string(v.pos1) + "\t" + v.ref + "\t";
    for (usize a = 0; a < v.alts.size(); ++a) txt += (a ? "," : "") + v.alts[a].seq;
    txt += "\t";
    for (usize a = 0; a < v.alts.size(); ++a) {
      if (a) txt += ";";
      for (usize h = 0; h < v.alts[a].hap_starts.size(); ++h)
        txt += (h ? "," : "") + std::to_string(v.alts[a].hap_starts[h].first) + ":" +
               std::to_string(v.alts[a].hap_starts[h].second);
    }
    txt += "\t" + std::to_string(v.ref_start0) + "\n";
  }
  if (static_cast<int>(txt.size()) + 1 > out_cap) return -1;
  std::memcpy(out, txt.c_str(), txt.size() + 1);
  return static_cast<int>(vars.size());
}

// Pin hook for the restated SPOA engine (tests/test_aligner_pins.py): build the POA graph from the first nseq-1
// NUL-separated sequences (unit weights), align the LAST one against it and dump, as int32 words into `out`:
//   V, then per node v (by id): letter, n_preds, preds...;  then V ids in topological (rank) order;
//   then A = number of alignment pairs, then A x (node id | -1, seq pos | -1); last word: the DP's optimal score.
// Returns the number of words written, or -1 when `cap` is too small.
int orc_poa_align_dump(const char* seqs_nul, int nseq, int m, int n, int g, int e, int q, int c, int32_t* out, int cap) {
  std::vector<std::string> seqs;
  const char* p = seqs_nul;
  for (int i = 0; i < nseq; ++i) {
    seqs.emplace_back(p);
    p += seqs.back().size() + 1;
  }
  PoaScoring sc{m, n, g, e, q, c};
  PoaGraph gr;
  std::vector<std::string_view> views(seqs.begin(), seqs.end() - 1);
  std::vector<std::vector<u32>> ws;
  for (usize i = 0; i + 1 < seqs.size(); ++i) ws.emplace_back(seqs[i].size(), 1u);
  UpdateSpoaState(gr, sc, views, ws);
  i32 dp_score = 0;
  PoaAlignment const aln = PoaAlign(sc, seqs.back(), gr, &dp_score);
  std::vector<int32_t> w;
  w.push_back(static_cast<int32_t>(gr.nodes.size()));
  for (auto const& nd : gr.nodes) {
    w.push_back(static_cast<int32_t>(gr.decoder[nd.code]));
    w.push_back(static_cast<int32_t>(nd.in_edges.size()));
    for (u32 ei : nd.in_edges) w.push_back(static_cast<int32_t>(gr.edges[ei].tail));
  }
  for (u32 id : gr.rank_to_node) w.push_back(static_cast<int32_t>(id));
  w.push_back(static_cast<int32_t>(aln.size()));
  for (auto const& pr : aln) {
    w.push_back(pr.first);
    w.push_back(pr.second);
  }
  w.push_back(dp_score);
  if (static_cast<int>(w.size()) > cap) return -1;
  std::memcpy(out, w.data(), w.size() * sizeof(int32_t));
  return static_cast<int>(w.size());
}

// single read<->haplotype alignment: rec[6] = hit, score, rs, re, qs, qe; cigar as text
int orc_align_pair(const char* read, int m, const char* hap, int n, int min_score,
                   int32_t* rec, char* cigar_txt, int cap) {
  AlignParams ap;
  ap.min_score = min_score;
  AlnResult a = AlignReadToHap(std::string_view(read, m), std::string_view(hap, n), ap);
  rec[0] = a.hit; rec[1] = a.score; rec[2] = a.rs; rec[3] = a.re; rec[4] = a.qs; rec[5] = a.qe;
  std::string t;
  for (auto const& u : a.cigar) t += std::to_string(u.len) + u.op;
  if (static_cast<int>(t.size()) + 1 > cap) return -1;
  std::memcpy(cigar_txt, t.c_str(), t.size() + 1);
  return 0;
}

// ---- batched stage entry points, same layout as the product ----
int orc_repeat_gate_batch(const ma_params_t* prm, const ma_batch_t* b, const ma_gate_out_t* out) {
  for (int w = 0; w < b->n_windows; ++w) {
    auto const ref = WindowRef(b, w);
    out->max_approx[w] = MaxRepeatLen(ref, static_cast<usize>(prm->max_mismatch));
    out->max_exact[w] = MaxRepeatLen(ref, 0);
  }
  return 0;
}

int orc_assemble_batch(const ma_params_t* prm, const ma_batch_t* b, const ma_asm_out_t* o) {
  Params const P = ToParams(prm);
  int const MC = prm->max_comps, MH = prm->max_haps, ML = prm->max_hap_len, MR = prm->max_runs;
  for (int w = 0; w < b->n_windows; ++w) {
    auto const reads = WindowReads(b, w);
    AssemblyResult const res = BuildComponentResults(WindowRef(b, w), reads, P);
    u32 status = 0;
    o->win_k[w] = res.used_k;
    u32 nalt = 0;
    for (auto const& c : res.comps) nalt += static_cast<u32>(c.haps.size()) - 1;
    if (nalt == 0) status |= MA_W_NO_HAPLOTYPE;
    if (res.hit_bfs_limit) status |= MA_W_BFS_LIMIT;
    u32 ncomp = 0, slot = 0;
    for (auto const& c : res.comps) {
      if (c.hit_bfs_limit) status |= MA_W_BFS_LIMIT;
      if (static_cast<int>(ncomp) >= MC || static_cast<int>(slot + c.haps.size()) > MH) {
        status |= MA_W_HAP_OVERFLOW;
        break;
      }
      usize const ci = static_cast<usize>(w) * MC + ncomp;
      o->comp_anchor[ci] = c.anchor_start;
      o->comp_hap0[ci] = slot;
      o->comp_nhaps[ci] = static_cast<u32>(c.haps.size());
      o->comp_cx[ci * 3 + 0] = static_cast<u32>(c.metrics.cyclomatic);
      o->comp_cx[ci * 3 + 1] = static_cast<u32>(c.metrics.branch_points);
      o->comp_cx[ci * 3 + 2] = static_cast<u32>(c.metrics.max_dir_degree);
      o->comp_cxf[ci * 4 + 0] = c.metrics.unitig_ratio;
      o->comp_cxf[ci * 4 + 1] = c.metrics.coverage_cv;
      o->comp_cxf[ci * 4 + 2] = c.metrics.tip_to_path;
      o->comp_cxf[ci * 4 + 3] = c.MaxAltPathCv();
      for (auto const& h : c.haps) {
        usize const hi = static_cast<usize>(w) * MH + slot;
        u32 len = static_cast<u32>(h.seq.size()), nr = static_cast<u32>(h.node_weights.size());
        if (static_cast<int>(len) > ML || static_cast<int>(nr) > MR) {
          status |= MA_W_LEN_OVERFLOW;
          len = std::min<u32>(len, ML);
          nr = std::min<u32>(nr, MR);
        }
        o->hap_len[hi] = len;
        o->hap_nruns[hi] = nr;
        double const st[6] = {h.mean_cov, h.median_cov, h.sd_cov, h.cv_cov, h.qcv_cov, h.total_cov};
        std::memcpy(o->hap_stats + hi * 6, st, sizeof(st));
        std::memcpy(o->hap_bases + hi * ML, h.seq.data(), len);
        for (u32 r = 0; r < nr; ++r) {
          o->hap_runs[(hi * MR + r) * 2 + 0] = h.node_weights[r].first;
          o->hap_runs[(hi * MR + r) * 2 + 1] = h.node_weights[r].second;
        }
        slot++;
      }
      ncomp++;
    }
    o->win_ncomp[w] = ncomp;
    o->win_status[w] = status;
  }
  return 0;
}

namespace {
struct CompView {
  std::vector<std::string> haps;
  std::vector<std::vector<u32>> weights;
  u32 anchor = 0, hap0 = 0;
};
std::vector<CompView> LoadComps(const ma_params_t* prm, const ma_asm_out_t* a, int w) {
  std::vector<CompView> cs;
  int const MC = prm->max_comps, MH = prm->max_haps, ML = prm->max_hap_len, MR = prm->max_runs;
  for (u32 c = 0; c < a->win_ncomp[w]; ++c) {
    usize const ci = static_cast<usize>(w) * MC + c;
    CompView cv;
    cv.anchor = a->comp_anchor[ci];
    cv.hap0 = a->comp_hap0[ci];
    for (u32 h = 0; h < a->comp_nhaps[ci]; ++h) {
      usize const hi = static_cast<usize>(w) * MH + cv.hap0 + h;
      cv.haps.emplace_back(reinterpret_cast<const char*>(a->hap_bases) + hi * ML, a->hap_len[hi]);
      std::vector<u32> wt;
      for (u32 r = 0; r < a->hap_nruns[hi]; ++r)
        wt.insert(wt.end(), a->hap_runs[(hi * MR + r) * 2 + 1], a->hap_runs[(hi * MR + r) * 2 + 0]);
      wt.resize(a->hap_len[hi], wt.empty() ? 0 : wt.back());
      cv.weights.push_back(std::move(wt));
    }
    cs.push_back(std::move(cv));
  }
  return cs;
}
}  // namespace

int orc_msa_batch(const ma_params_t* prm, const ma_batch_t* b, const ma_asm_out_t* a, const ma_var_out_t* o) {
  int const MH = prm->max_haps, MV = prm->max_vars, MA = prm->max_alts, MP = prm->max_allele_bytes;
  PoaScoring const sc;  // production convex parameters (msa_builder.h:72-77)
  for (int w = 0; w < b->n_windows; ++w) {
    u32 nv = 0, pool = 0;
    bool overflow = false;
    if (!(a->win_status[w] & MA_W_NO_HAPLOTYPE)) {
      auto const comps = LoadComps(prm, a, w);
      for (usize c = 0; c < comps.size() && !overflow; ++c) {
        PoaGraph g;
        std::vector<std::string_view> views(comps[c].haps.begin(), comps[c].haps.end());
        UpdateSpoaState(g, sc, views, comps[c].weights);
        // pos is reported window-relative: ref_anchor_pos1 = StartPos1 + anchor (variant_builder.cpp:146)
        auto const vars = ExtractVariants(g, comps[c].anchor);
        for (auto const& v : vars) {
          u32 need = static_cast<u32>(v.ref.size());
          for (auto const& al : v.alts) need += static_cast<u32>(al.seq.size());
          if (static_cast<int>(nv) >= MV || static_cast<int>(v.alts.size()) > MA ||
              static_cast<int>(pool + need) > MP) {
            overflow = true;
            break;
          }
          usize const vi = static_cast<usize>(w) * MV + nv;
          u8* pl = o->allele_pool + static_cast<usize>(w) * MP;
          o->var_comp[vi] = static_cast<u32>(c);
          o->var_pos[vi] = static_cast<u32>(v.pos1);
          o->var_ref_start[vi] = static_cast<u32>(v.ref_start0);
          o->var_ref_off[vi] = pool;
          o->var_ref_len[vi] = static_cast<u32>(v.ref.size());
          std::memcpy(pl + pool, v.ref.data(), v.ref.size());
          pool += static_cast<u32>(v.ref.size());
          o->var_nalts[vi] = static_cast<u32>(v.alts.size());
          for (int h = 0; h < MH; ++h) {
            o->var_hap_allele[vi * MH + h] = 0;
            o->var_hap_start[vi * MH + h] = 0;
          }
          for (usize ai = 0; ai < v.alts.size(); ++ai) {
            auto const& al = v.alts[ai];
            o->alt_off[vi * MA + ai] = pool;
            o->alt_len[vi * MA + ai] = static_cast<u32>(al.seq.size());
            std::memcpy(pl + pool, al.seq.data(), al.seq.size());
            pool += static_cast<u32>(al.seq.size());
            o->alt_type[vi * MA + ai] = al.type;
            o->alt_length[vi * MA + ai] = static_cast<i32>(al.length);
            for (auto const& hs : al.hap_starts) {
              o->var_hap_allele[vi * MH + hs.first] = static_cast<u8>(ai + 1);
              o->var_hap_start[vi * MH + hs.first] = hs.second;
            }
          }
          // haplotype 0 and REF-carrying haplotypes: allele 0; the REF start is var_ref_start
          o->var_hap_start[vi * MH + 0] = static_cast<u32>(v.ref_start0);
          nv++;
        }
      }
    }
    o->win_nvars[w] = nv;
    if (overflow) a->win_status[w] |= MA_W_VAR_OVERFLOW;
  }
  return 0;
}

int orc_genotype_batch(const ma_params_t* prm, const ma_batch_t* b, const ma_asm_out_t* a,
                       const ma_var_out_t* vo, const ma_geno_out_t* o) {
  int const MH = prm->max_haps, MV = prm->max_vars, MA = prm->max_alts, MP = prm->max_allele_bytes;
  int const S = prm->num_samples, NA = MA + 1, MCG = prm->max_cigar;
  AlignParams ap;
  ap.min_score = prm->min_aln_score;
  for (int w = 0; w < b->n_windows; ++w) {
    u32 const nv = vo->win_nvars[w];
    for (u32 v = 0; v < static_cast<u32>(MV); ++v) {
      usize const vi = static_cast<usize>(w) * MV + v;
      o->var_qual[vi] = 0.0;
      for (int x = 0; x < S * NA * 2; ++x) o->allele_counts[vi * S * NA * 2 + x] = 0;
      if (o->var_pl)
        for (int x = 0; x < S * (NA * (NA + 1) / 2); ++x) o->var_pl[vi * S * (NA * (NA + 1) / 2) + x] = 0;
      if (o->var_gq)
        for (int x = 0; x < S; ++x) o->var_gq[vi * S + x] = 0;
    }
    auto const reads = WindowReads(b, w);
    u32 const r0 = b->read_win_off[w];
    if (o->aln_rec)
      for (usize r = 0; r < reads.size(); ++r)
        for (int h = 0; h < MH; ++h) {
          for (int x = 0; x < 6; ++x) o->aln_rec[((r0 + r) * MH + h) * 6 + x] = 0;
          if (o->aln_cigar) o->aln_cigar[((r0 + r) * MH + h) * (1 + MCG)] = 0;
        }
    if (o->asg_allele)
      for (usize r = 0; r < reads.size(); ++r)
        for (int v = 0; v < MV; ++v) {
          o->asg_allele[(r0 + r) * MV + v] = 255;
          if (o->asg_score) o->asg_score[(r0 + r) * MV + v] = 0.0;
        }
    if (nv == 0) continue;
    auto const comps = LoadComps(prm, a, w);
    const u8* pl = vo->allele_pool + static_cast<usize>(w) * MP;
    std::vector<u8> sample_case(S, 0);
    for (auto const& rd : reads)
      if (rd.role == 1 && rd.sample < S) sample_case[rd.sample] = 1;
    for (usize c = 0; c < comps.size(); ++c) {
      // rebuild this component's RawVariant list from the flat layout
      std::vector<RawVariant> vars;
      std::vector<u32> vslot;
      for (u32 v = 0; v < nv; ++v) {
        usize const vi = static_cast<usize>(w) * MV + v;
        if (vo->var_comp[vi] != c) continue;
        RawVariant rv;
        rv.pos1 = vo->var_pos[vi];
        rv.ref_start0 = vo->var_ref_start[vi];
        rv.ref.assign(reinterpret_cast<const char*>(pl) + vo->var_ref_off[vi], vo->var_ref_len[vi]);
        for (u32 ai = 0; ai < vo->var_nalts[vi]; ++ai) {
          AltAllele al;
          al.seq.assign(reinterpret_cast<const char*>(pl) + vo->alt_off[vi * MA + ai], vo->alt_len[vi * MA + ai]);
          for (usize h = 1; h < comps[c].haps.size(); ++h)
            if (vo->var_hap_allele[vi * MH + h] == ai + 1)
              al.hap_starts.push_back({static_cast<u32>(h), vo->var_hap_start[vi * MH + h]});
          rv.alts.push_back(std::move(al));
        }
        vars.push_back(std::move(rv));
        vslot.push_back(v);
      }
      if (vars.empty()) continue;  // variant_builder.cpp:248
      // dedup sets keyed (variant, sample, allele) -> qname ids (variant_support.cpp:24-30)
      std::map<std::tuple<u32, u32, u32>, std::set<u32>> seen;
      for (usize r = 0; r < reads.size(); ++r) {
        auto const& rd = reads[r];
        std::vector<AlnResult> alns;
        auto const asg = AssignReadToAlleles(rd, comps[c].haps, vars, ap, &alns);
        if (o->aln_rec)
          for (auto const& al : alns) {
            usize const base = ((r0 + r) * MH + comps[c].hap0 + al.hap);
            int32_t* rec = o->aln_rec + base * 6;
            rec[0] = 1; rec[1] = al.score; rec[2] = al.rs; rec[3] = al.re; rec[4] = al.qs; rec[5] = al.qe;
            if (o->aln_cigar) {
              u32* cg = o->aln_cigar + base * (1 + MCG);
              cg[0] = static_cast<u32>(al.cigar.size());
              for (usize x = 0; x < al.cigar.size() && static_cast<int>(x) < MCG; ++x) {
                u32 const op = al.cigar[x].op == 'M' ? 0 : al.cigar[x].op == 'I' ? 1 : al.cigar[x].op == 'D' ? 2 : 4;
                cg[1 + x] = (al.cigar[x].len << 4) | op;
              }
            }
          }
        for (usize x = 0; x < vars.size(); ++x) {
          if (!asg[x].valid) continue;
          u32 const v = vslot[x];
          usize const vi = static_cast<usize>(w) * MV + v;
          if (o->asg_allele) {
            o->asg_allele[(r0 + r) * MV + v] = static_cast<u8>(asg[x].allele);
            if (o->asg_score) o->asg_score[(r0 + r) * MV + v] = asg[x].Combined();
          }
          if (rd.sample >= S) continue;
          auto& ss = seen[{v, rd.sample, asg[x].allele}];
          if (!ss.insert(rd.qname_id).second) continue;
          o->allele_counts[((vi * S + rd.sample) * NA + asg[x].allele) * 2 + (rd.rev ? 1 : 0)] += 1;
        }
      }
      // FORMAT PL / GQ of every sample with evidence (variant_call.cpp:141-163, variant_support.cpp:408-426) and, outside
      // case/control mode, QUAL = max over those samples of PL[0/0] (variant_call.cpp:289-303)
      int const G = NA * (NA + 1) / 2;
      for (u32 v : vslot) {
        usize const vi = static_cast<usize>(w) * MV + v;
        int const K = static_cast<int>(vo->var_nalts[vi]) + 1;
        f64 qual = 0.0;
        for (int s = 0; s < S; ++s) {
          std::vector<int> counts(static_cast<usize>(K), 0);
          u64 tot = 0;
          for (int al = 0; al < NA; ++al) {
            u32 const c2 = o->allele_counts[((vi * S + s) * NA + al) * 2] + o->allele_counts[((vi * S + s) * NA + al) * 2 + 1];
            if (al < K) counts[static_cast<usize>(al)] = static_cast<int>(c2);
            tot += c2;
          }
          if (o->var_pl)
            for (int x = 0; x < G; ++x) o->var_pl[(vi * S + s) * G + x] = 0;
          if (o->var_gq) o->var_gq[vi * S + s] = 0;
          if (tot == 0) continue;  // evidence.Find(sample) == nullptr: missing support
          auto const pls = ComputeGenotypePLs(counts);
          if (o->var_pl)
            for (usize x = 0; x < pls.size(); ++x) o->var_pl[(vi * S + s) * G + x] = pls[x];
          if (o->var_gq) o->var_gq[vi * S + s] = ComputeGenotypeQuality(pls);
          qual = std::max(qual, static_cast<f64>(pls.empty() ? 0u : pls[0]));
        }
        if (!prm->case_ctrl_mode) o->var_qual[vi] = qual;
      }
      // QUAL = max over samples with evidence of SOLOR (variant_call.cpp:289-345)
      if (prm->case_ctrl_mode)
        for (u32 v : vslot) {
          usize const vi = static_cast<usize>(w) * MV + v;
          auto cov = [&](int s, bool alt) {
            u64 t = 0;
            for (int al = alt ? 1 : 0; al < (alt ? NA : 1); ++al)
              t += o->allele_counts[((vi * S + s) * NA + al) * 2] + o->allele_counts[((vi * S + s) * NA + al) * 2 + 1];
            return t;
          };
          f64 sum_alt = 0, sum_ref = 0, cnt = 0;
          for (int s = 0; s < S; ++s) {
            if (sample_case[s] || cov(s, false) + cov(s, true) == 0) continue;
            sum_alt += static_cast<f64>(cov(s, true));
            sum_ref += static_cast<f64>(cov(s, false));
            cnt += 1.0;
          }
          f64 const cc = std::max(cnt, 1.0);
          f64 const ctrl_alt = sum_alt / cc + 1.0, ctrl_ref = sum_ref / cc + 1.0;
          f64 qual = 0.0;
          for (int s = 0; s < S; ++s) {
            if (!sample_case[s] || cov(s, false) + cov(s, true) == 0) continue;
            f64 const case_alt = static_cast<f64>(cov(s, true)) + 1.0;
            f64 const case_ref = static_cast<f64>(cov(s, false)) + 1.0;
            qual = std::max(qual, std::log((case_alt * ctrl_ref) / (case_ref * ctrl_alt)));
          }
          o->var_qual[vi] = qual;
        }
    }
  }
  return 0;
}

// ---- unit taps for the known-answer tests + the batched annotator ------------------
int orc_max_hrun(const char* s, uint64_t n) { return MaxHomopolymerRun(std::string_view(s, n)); }
float orc_entropy(const char* s, uint64_t n) { return LocalShannonEntropy(std::string_view(s, n)); }
// out: per repeat 5 x i32 (period, start, span, errors, exact) ; copies[] f32.  Returns the number found.
int orc_find_repeats(const char* s, uint64_t n, int approx, int32_t* out, float* copies, int cap) {
  auto const rs = approx ? FindApproxRepeats(std::string_view(s, n)) : FindExactRepeats(std::string_view(s, n));
  for (usize i = 0; i < rs.size() && static_cast<int>(i) < cap; ++i) {
    out[i * 5 + 0] = rs[i].period;
    out[i * 5 + 1] = rs[i].start;
    out[i * 5 + 2] = rs[i].span;
    out[i * 5 + 3] = rs[i].errors;
    out[i * 5 + 4] = rs[i].exact ? 1 : 0;
    copies[i] = rs[i].copies;
  }
  return static_cast<int>(rs.size());
}
double orc_longdust(const char* s, uint64_t n, int k, int max_len, double gc, int one_strand) {
  LongdustQ const q(k, max_len, gc);
  return one_strand ? q.ScoreOneStrand(std::string_view(s, n)) : q.Score(std::string_view(s, n));
}
int orc_longdust_ftable(int k, int max_len, double gc, double* out) {
  LongdustQ const q(k, max_len, gc);
  for (usize i = 0; i < q.F.size(); ++i) out[i] = q.F[i];
  return static_cast<int>(q.F.size());
}
// SequenceComplexityScorer::Score on one (ref, alt) pair; out_i[4], out_f[4], out_d[3] as ma_cx_out_t
void orc_seqcx_score(const char* ref, uint64_t rn, uint64_t rpos, uint64_t rlen, const char* alt, uint64_t an,
                     uint64_t apos, uint64_t alen, double gc, int32_t* out_i, float* out_f, double* out_d) {
  SeqCxScorer const sc(gc);
  SeqCx const c = sc.Score({std::string_view(ref, rn), rpos, rlen}, {std::string_view(alt, an), apos, alen});
  out_i[0] = c.ctx_hrun; out_i[1] = c.delta_hrun; out_i[2] = c.tr_period; out_i[3] = c.stutter;
  out_f[0] = c.ctx_entropy; out_f[1] = c.delta_entropy; out_f[2] = c.tr_affinity; out_f[3] = c.tr_purity;
  out_d[0] = c.ctx_flank_lq; out_d[1] = c.ctx_hap_lq; out_d[2] = c.delta_flank_lq;
}

int orc_annotate_batch(const ma_params_t* prm, const ma_batch_t* b, const ma_asm_out_t* a, const ma_var_out_t* v,
                       double gc_frac, const ma_cx_out_t* o) {
  int const MC = prm->max_comps, MH = prm->max_haps, MV = prm->max_vars, MA = prm->max_alts;
  SeqCxScorer const sc(gc_frac);
  for (int w = 0; w < b->n_windows; ++w) {
    if (v->win_nvars[w] == 0) continue;
    auto const comps = LoadComps(prm, a, w);
    for (u32 i = 0; i < v->win_nvars[w]; ++i) {
      usize const vi = static_cast<usize>(w) * MV + i;
      u32 const c = v->var_comp[vi];
      auto const& cv = comps[c];
      std::vector<std::string_view> haps(cv.haps.begin(), cv.haps.end());
      std::vector<AltSites> alts(v->var_nalts[vi]);
      for (u32 ai = 0; ai < v->var_nalts[vi]; ++ai) {
        alts[ai].len = v->alt_len[vi * MA + ai];
        // var_hap_* are indexed by haplotype-of-the-component (AltAllele::mLocalHapStart0Idxs)
        for (u32 h = 1; h < haps.size(); ++h)
          if (v->var_hap_allele[vi * MH + h] == ai + 1) alts[ai].hap_starts.emplace_back(h, v->var_hap_start[vi * MH + h]);
      }
      SeqCx const s = AnnotateVariant(sc, haps, v->var_ref_start[vi], v->var_ref_len[vi], alts);
      o->seq_cx_i[vi * 4 + 0] = s.ctx_hrun; o->seq_cx_i[vi * 4 + 1] = s.delta_hrun;
      o->seq_cx_i[vi * 4 + 2] = s.tr_period; o->seq_cx_i[vi * 4 + 3] = s.stutter;
      o->seq_cx_f[vi * 4 + 0] = s.ctx_entropy; o->seq_cx_f[vi * 4 + 1] = s.delta_entropy;
      o->seq_cx_f[vi * 4 + 2] = s.tr_affinity; o->seq_cx_f[vi * 4 + 3] = s.tr_purity;
      o->seq_cx_d[vi * 3 + 0] = s.ctx_flank_lq; o->seq_cx_d[vi * 3 + 1] = s.ctx_hap_lq;
      o->seq_cx_d[vi * 3 + 2] = s.delta_flank_lq;
      // cbdg/graph_complexity.h:160-166 + variant_annotator.cpp:87-99
      usize const ci = static_cast<usize>(w) * MC + c;
      f64 const cc = static_cast<f64>(a->comp_cx[ci * 3 + 0]), bp = static_cast<f64>(a->comp_cx[ci * 3 + 1]);
      f64 const raw = (cc * bp * a->comp_cxf[ci * 4 + 1]) / (a->comp_cxf[ci * 4 + 0] + 1e-6);
      o->graph_cx[vi * 3 + 0] = std::log10(1.0 + raw);
      o->graph_cx[vi * 3 + 1] = a->comp_cxf[ci * 4 + 2];
      o->graph_cx[vi * 3 + 2] = static_cast<f64>(a->comp_cx[ci * 3 + 2]);
    }
  }
  return 0;
}

}  // extern "C"
