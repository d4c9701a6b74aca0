// ORACLE -- TEST INFRASTRUCTURE ONLY (see common.hpp header).
// Haplotype <-> reference partial-order alignment.  The reference calls SPOA 4.1.5
// (caller/msa_builder.cpp:29-42; engine configured at caller/msa_builder.h:72-89), whose
// source is NOT under /root/reference (cmake/dependencies.cmake:173-176).  This file
// restates SPOA's published algorithm (scalar "SISD" engine: sequence-to-DAG global
// alignment with linear / affine / convex gaps; Graph::AddAlignment; DFS topological sort
// that keeps aligned nodes adjacent).  Parity is PINNED only by the seven known-answer
// cases of tests/caller/variant_set_test.cpp:35-247 (linear gaps 3/-5/-3) through
// ExtractVariants below; the production convex parameters are unpinned.
#include "oracle.hpp"

#include <limits>
#include <map>
#include <stack>

namespace orc {

void PoaGraph::Clear() {
  nodes.clear();
  edges.clear();
  seq_first.clear();
  rank_to_node.clear();
  for (auto& c : coder) c = -1;
  std::memset(decoder, 0, sizeof(decoder));
  num_codes = 0;
}

i32 PoaGraph::Successor(u32 node, u32 label) const {
  for (u32 ei : nodes[node].out_edges) {
    auto const& lab = edges[ei].labels;
    if (std::find(lab.begin(), lab.end(), label) != lab.end()) return static_cast<i32>(edges[ei].head);
  }
  return -1;
}

namespace {

u32 AddNode(PoaGraph& g, u8 code) {
  PoaNode n;
  n.id = static_cast<u32>(g.nodes.size());
  n.code = code;
  g.nodes.push_back(std::move(n));
  return g.nodes.back().id;
}

void AddEdge(PoaGraph& g, u32 tail, u32 head, i64 weight) {
  u32 const label = static_cast<u32>(g.seq_first.size());
  for (u32 ei : g.nodes[tail].out_edges) {
    if (g.edges[ei].head == head) {
      g.edges[ei].labels.push_back(label);
      g.edges[ei].weight += weight;
      return;
    }
  }
  PoaEdge e;
  e.tail = tail;
  e.head = head;
  e.weight = weight;
  e.labels.push_back(label);
  u32 const ei = static_cast<u32>(g.edges.size());
  g.edges.push_back(std::move(e));
  g.nodes[tail].out_edges.push_back(ei);
  g.nodes[head].in_edges.push_back(ei);
}

// spoa::Graph::AddSequence: chain of fresh nodes for seq[begin,end); returns first id or -1.
i32 AddSequence(PoaGraph& g, std::string_view seq, const std::vector<u32>& w, u32 begin, u32 end) {
  if (begin == end) return -1;
  i32 prev = -1;
  for (u32 i = begin; i < end; ++i) {
    u32 const cur = AddNode(g, static_cast<u8>(g.coder[static_cast<u8>(seq[i])]));
    if (prev >= 0) AddEdge(g, static_cast<u32>(prev), cur, static_cast<i64>(w[i - 1]) + w[i]);
    prev = static_cast<i32>(cur);
  }
  return static_cast<i32>(g.nodes.size() - (end - begin));
}

// spoa::Graph::TopologicalSort: iterative DFS over in-edges; a node is emitted together
// with all nodes aligned to it so they occupy consecutive ranks.
void TopologicalSort(PoaGraph& g) {
  g.rank_to_node.clear();
  usize const n = g.nodes.size();
  std::vector<u8> marks(n, 0);
  std::vector<u8> ignored(n, 0);
  std::vector<u32> stack;
  for (u32 s = 0; s < n; ++s) {
    if (marks[s] != 0) continue;
    stack.push_back(s);
    while (!stack.empty()) {
      u32 const cur = stack.back();
      bool valid = true;
      if (marks[cur] != 2) {
        for (u32 ei : g.nodes[cur].in_edges) {
          u32 const t = g.edges[ei].tail;
          if (marks[t] != 2) {
            stack.push_back(t);
            valid = false;
          }
        }
        if (!ignored[cur]) {
          for (u32 a : g.nodes[cur].aligned) {
            if (marks[a] != 2) {
              stack.push_back(a);
              ignored[a] = 1;
              valid = false;
            }
          }
        }
        if (valid) {
          marks[cur] = 2;
          if (!ignored[cur]) {
            g.rank_to_node.push_back(cur);
            for (u32 a : g.nodes[cur].aligned) g.rank_to_node.push_back(a);
          }
        } else {
          marks[cur] = 1;
        }
      }
      if (valid) stack.pop_back();
    }
  }
}

constexpr i32 kNegInf = std::numeric_limits<i32>::min() + 1024;

}  // namespace

// spoa::Graph::AddAlignment(alignment, sequence, len, weights)
void PoaAddAlignment(PoaGraph& g, const PoaAlignment& aln, std::string_view seq,
                     const std::vector<u32>& weights) {
  u32 const len = static_cast<u32>(seq.size());
  if (len == 0) return;
  for (u32 i = 0; i < len; ++i) {
    u8 const c = static_cast<u8>(seq[i]);
    if (g.coder[c] == -1) {
      g.coder[c] = static_cast<i32>(g.num_codes);
      g.decoder[g.num_codes++] = c;
    }
  }
  if (aln.empty()) {
    i32 const first = AddSequence(g, seq, weights, 0, len);
    g.seq_first.push_back(first);
    TopologicalSort(g);
    return;
  }
  std::vector<u32> valid;
  for (auto const& p : aln)
    if (p.second != -1) valid.push_back(static_cast<u32>(p.second));
  i32 begin = AddSequence(g, seq, weights, 0, valid.front());
  i32 prev = begin >= 0 ? static_cast<i32>(g.nodes.size() - 1) : -1;
  i32 const last = AddSequence(g, seq, weights, valid.back() + 1, len);
  for (auto const& p : aln) {
    if (p.second == -1) continue;
    u8 const code = static_cast<u8>(g.coder[static_cast<u8>(seq[p.second])]);
    i32 curr = -1;
    if (p.first == -1) {
      curr = static_cast<i32>(AddNode(g, code));
    } else {
      u32 const jt = static_cast<u32>(p.first);
      if (g.nodes[jt].code == code) {
        curr = static_cast<i32>(jt);
      } else {
        for (u32 kt : g.nodes[jt].aligned)
          if (g.nodes[kt].code == code) {
            curr = static_cast<i32>(kt);
            break;
          }
        if (curr < 0) {
          curr = static_cast<i32>(AddNode(g, code));
          std::vector<u32> const al = g.nodes[jt].aligned;
          for (u32 kt : al) {
            g.nodes[kt].aligned.push_back(static_cast<u32>(curr));
            g.nodes[curr].aligned.push_back(kt);
          }
          g.nodes[jt].aligned.push_back(static_cast<u32>(curr));
          g.nodes[curr].aligned.push_back(jt);
        }
      }
    }
    if (begin < 0) begin = curr;
    if (prev >= 0)
      AddEdge(g, static_cast<u32>(prev), static_cast<u32>(curr),
              static_cast<i64>(weights[p.second - 1]) + weights[p.second]);
    prev = curr;
  }
  if (last >= 0)
    AddEdge(g, static_cast<u32>(prev), static_cast<u32>(last),
            static_cast<i64>(weights[valid.back()]) + weights[valid.back() + 1]);
  g.seq_first.push_back(begin);
  TopologicalSort(g);
}

// spoa::SisdAlignmentEngine::Align, AlignmentType::kNW.  Subtype selection follows
// AlignmentEngine::Create: g >= e -> linear; g <= q || e >= c -> affine; else convex.
PoaAlignment PoaAlign(const PoaScoring& sc, std::string_view seq, const PoaGraph& g, i32* score_out) {
  u32 const L = static_cast<u32>(seq.size());
  if (g.nodes.empty() || L == 0) return {};
  enum { LINEAR, AFFINE, CONVEX } sub = sc.g >= sc.e ? LINEAR : ((sc.g <= sc.q || sc.e >= sc.c) ? AFFINE : CONVEX);
  i32 const m_ = sc.m, n_ = sc.n, g_ = sc.g;
  i32 const e_ = sub == LINEAR ? sc.g : sc.e;
  i32 const q_ = sub == CONVEX ? sc.q : g_, c_ = sub == CONVEX ? sc.c : e_;

  u64 const W = L + 1;
  u32 const V = static_cast<u32>(g.rank_to_node.size());
  u64 const HH = V + 1;
  std::vector<u32> rank_of(g.nodes.size(), 0);
  for (u32 r = 0; r < V; ++r) rank_of[g.rank_to_node[r]] = r;
  // sequence profile by code (compares decoded char with sequence char)
  std::vector<i32> prof(static_cast<usize>(g.num_codes) * W, 0);
  for (u32 c = 0; c < g.num_codes; ++c) {
    prof[c * W] = 0;
    for (u32 j = 0; j < L; ++j) prof[c * W + j + 1] = (g.decoder[c] == static_cast<u8>(seq[j])) ? m_ : n_;
  }
  std::vector<i32> H(HH * W), F, E, O, Q;
  if (sub != LINEAR) {
    F.assign(HH * W, 0);
    E.assign(HH * W, 0);
  }
  if (sub == CONVEX) {
    O.assign(HH * W, 0);
    Q.assign(HH * W, 0);
  }
  auto pred_rows = [&](u32 node, std::vector<u32>& out) {
    out.clear();
    for (u32 ei : g.nodes[node].in_edges) out.push_back(rank_of[g.edges[ei].tail] + 1);
  };
  std::vector<u32> preds;

  // --- Initialize (kNW) ---
  if (sub == CONVEX) {
    O[0] = 0;
    Q[0] = 0;
    for (u64 j = 1; j < W; ++j) {
      O[j] = kNegInf;
      Q[j] = q_ + static_cast<i32>(j - 1) * c_;
    }
    for (u64 i = 1; i < HH; ++i) {
      pred_rows(g.rank_to_node[i - 1], preds);
      i32 pen = preds.empty() ? q_ - c_ : kNegInf;
      for (u32 p : preds) pen = std::max(pen, O[p * W]);
      O[i * W] = pen + c_;
      Q[i * W] = kNegInf;
    }
  }
  if (sub != LINEAR) {
    F[0] = 0;
    E[0] = 0;
    for (u64 j = 1; j < W; ++j) {
      F[j] = kNegInf;
      E[j] = g_ + static_cast<i32>(j - 1) * e_;
    }
    for (u64 i = 1; i < HH; ++i) {
      pred_rows(g.rank_to_node[i - 1], preds);
      i32 pen = preds.empty() ? g_ - e_ : kNegInf;
      for (u32 p : preds) pen = std::max(pen, F[p * W]);
      F[i * W] = pen + e_;
      E[i * W] = kNegInf;
    }
  }
  H[0] = 0;
  if (sub == CONVEX) {
    for (u64 j = 1; j < W; ++j) H[j] = std::max(Q[j], E[j]);
    for (u64 i = 1; i < HH; ++i) H[i * W] = std::max(O[i * W], F[i * W]);
  } else if (sub == AFFINE) {
    for (u64 j = 1; j < W; ++j) H[j] = E[j];
    for (u64 i = 1; i < HH; ++i) H[i * W] = F[i * W];
  } else {
    for (u64 j = 1; j < W; ++j) H[j] = static_cast<i32>(j) * g_;
    for (u64 i = 1; i < HH; ++i) {
      pred_rows(g.rank_to_node[i - 1], preds);
      i32 pen = preds.empty() ? 0 : kNegInf;
      for (u32 p : preds) pen = std::max(pen, H[p * W]);
      H[i * W] = pen + g_;
    }
  }

  // --- fill ---
  i32 max_score = kNegInf;
  u32 max_i = 0, max_j = 0;
  for (u32 r = 0; r < V; ++r) {
    u32 const node = g.rank_to_node[r];
    const i32* cp = &prof[static_cast<usize>(g.nodes[node].code) * W];
    u64 const i = r + 1;
    pred_rows(node, preds);
    u64 const p0 = preds.empty() ? 0 : preds[0];
    i32* Hr = &H[i * W];
    if (sub == LINEAR) {
      const i32* Hp = &H[p0 * W];
      for (u64 j = 1; j < W; ++j) Hr[j] = std::max(Hp[j - 1] + cp[j], Hp[j] + g_);
      for (usize p = 1; p < preds.size(); ++p) {
        Hp = &H[preds[p] * W];
        for (u64 j = 1; j < W; ++j) Hr[j] = std::max(Hp[j - 1] + cp[j], std::max(Hr[j], Hp[j] + g_));
      }
      for (u64 j = 1; j < W; ++j) Hr[j] = std::max(Hr[j - 1] + g_, Hr[j]);
    } else {
      i32* Fr = &F[i * W];
      i32* Or = sub == CONVEX ? &O[i * W] : nullptr;
      {
        const i32* Hp = &H[p0 * W];
        const i32* Fp = &F[p0 * W];
        const i32* Op = sub == CONVEX ? &O[p0 * W] : nullptr;
        for (u64 j = 1; j < W; ++j) {
          Fr[j] = std::max(Hp[j] + g_, Fp[j] + e_);
          if (Or) Or[j] = std::max(Hp[j] + q_, Op[j] + c_);
          Hr[j] = Hp[j - 1] + cp[j];
        }
      }
      for (usize p = 1; p < preds.size(); ++p) {
        const i32* Hp = &H[preds[p] * W];
        const i32* Fp = &F[preds[p] * W];
        const i32* Op = sub == CONVEX ? &O[preds[p] * W] : nullptr;
        for (u64 j = 1; j < W; ++j) {
          Fr[j] = std::max(Fr[j], std::max(Hp[j] + g_, Fp[j] + e_));
          if (Or) Or[j] = std::max(Or[j], std::max(Hp[j] + q_, Op[j] + c_));
          Hr[j] = std::max(Hr[j], Hp[j - 1] + cp[j]);
        }
      }
      i32* Er = &E[i * W];
      i32* Qr = sub == CONVEX ? &Q[i * W] : nullptr;
      for (u64 j = 1; j < W; ++j) {
        Er[j] = std::max(Hr[j - 1] + g_, Er[j - 1] + e_);
        if (Qr) {
          Qr[j] = std::max(Hr[j - 1] + q_, Qr[j - 1] + c_);
          Hr[j] = std::max(Hr[j], std::max(std::max(Fr[j], Er[j]), std::max(Or[j], Qr[j])));
        } else {
          Hr[j] = std::max(Hr[j], std::max(Fr[j], Er[j]));
        }
      }
    }
    if (g.nodes[node].out_edges.empty()) {  // kNW: only (sink node, last column)
      if (max_score < Hr[W - 1]) {
        max_score = Hr[W - 1];
        max_i = static_cast<u32>(i);
        max_j = static_cast<u32>(W - 1);
      }
    }
  }
  if (score_out) *score_out = max_score;
  if (max_i == 0 && max_j == 0) return {};

  // --- backtrack ---
  PoaAlignment aln;
  u32 i = max_i, j = max_j, prev_i = 0, prev_j = 0;
  while (!(i == 0 && j == 0)) {
    i32 const Hij = H[i * W + j];
    bool found = false, ext_left = false, ext_up = false;
    if (i != 0 && j != 0) {
      u32 const node = g.rank_to_node[i - 1];
      i32 const mc = prof[static_cast<usize>(g.nodes[node].code) * W + j];
      pred_rows(node, preds);
      u64 const p0 = preds.empty() ? 0 : preds[0];
      if (Hij == H[p0 * W + (j - 1)] + mc) {
        prev_i = static_cast<u32>(p0);
        prev_j = j - 1;
        found = true;
      } else {
        for (usize p = 1; p < preds.size(); ++p)
          if (Hij == H[preds[p] * W + (j - 1)] + mc) {
            prev_i = preds[p];
            prev_j = j - 1;
            found = true;
            break;
          }
      }
    }
    if (!found && i != 0) {
      u32 const node = g.rank_to_node[i - 1];
      pred_rows(node, preds);
      usize const np = std::max<usize>(preds.size(), 1);
      for (usize p = 0; p < np; ++p) {
        u64 const pi = preds.empty() ? 0 : preds[p];
        bool ok;
        if (sub == LINEAR) {
          ok = Hij == H[pi * W + j] + g_;
        } else if (sub == AFFINE) {
          ok = (ext_up = (Hij == F[pi * W + j] + e_)) || Hij == H[pi * W + j] + g_;
        } else {
          ok = (ext_up |= (Hij == F[pi * W + j] + e_)) || Hij == H[pi * W + j] + g_ ||
               (ext_up |= (Hij == O[pi * W + j] + c_)) || Hij == H[pi * W + j] + q_;
        }
        if (ok) {
          prev_i = static_cast<u32>(pi);
          prev_j = j;
          found = true;
          break;
        }
      }
    }
    if (!found && j != 0) {
      bool ok;
      if (sub == LINEAR) {
        ok = Hij == H[i * W + j - 1] + g_;
      } else if (sub == AFFINE) {
        ok = (ext_left = (Hij == E[i * W + j - 1] + e_)) || Hij == H[i * W + j - 1] + g_;
      } else {
        ok = (ext_left |= (Hij == E[i * W + j - 1] + e_)) || Hij == H[i * W + j - 1] + g_ ||
             (ext_left |= (Hij == Q[i * W + j - 1] + c_)) || Hij == H[i * W + j - 1] + q_;
      }
      if (ok) {
        prev_i = i;
        prev_j = j - 1;
        found = true;
      }
    }
    aln.emplace_back(i == prev_i ? -1 : static_cast<i32>(g.rank_to_node[i - 1]),
                     j == prev_j ? -1 : static_cast<i32>(j - 1));
    i = prev_i;
    j = prev_j;
    if (ext_left) {
      while (true) {
        aln.emplace_back(-1, static_cast<i32>(j - 1));
        --j;
        bool const e_stop = E[i * W + j] + e_ != E[i * W + j + 1];
        bool const q_stop = sub == CONVEX ? (Q[i * W + j] + c_ != Q[i * W + j + 1]) : true;
        if (e_stop && q_stop) break;
      }
    } else if (ext_up) {
      while (true) {
        bool stop = false;
        prev_i = 0;
        pred_rows(g.rank_to_node[i - 1], preds);
        if (sub == AFFINE) {
          for (u32 pi : preds) {
            if ((stop = (F[i * W + j] == H[pi * W + j] + g_)) || F[i * W + j] == F[pi * W + j] + e_) {
              prev_i = pi;
              break;
            }
          }
        } else {
          stop = true;
          for (u32 pi : preds) {
            if (F[i * W + j] == F[pi * W + j] + e_ || O[i * W + j] == O[pi * W + j] + c_) {
              prev_i = pi;
              stop = false;
              break;
            }
          }
          if (stop) {
            for (u32 pi : preds) {
              if (F[i * W + j] == H[pi * W + j] + g_ || O[i * W + j] == H[pi * W + j] + q_) {
                prev_i = pi;
                break;
              }
            }
          }
        }
        aln.emplace_back(static_cast<i32>(g.rank_to_node[i - 1]), -1);
        i = prev_i;
        if (stop || i == 0) break;
      }
    }
  }
  std::reverse(aln.begin(), aln.end());
  return aln;
}

// caller/msa_builder.cpp:29-42
void UpdateSpoaState(PoaGraph& g, const PoaScoring& sc, const std::vector<std::string_view>& seqs,
                     const std::vector<std::vector<u32>>& weights) {
  g.Clear();
  for (usize i = 0; i < seqs.size(); ++i) {
    auto const aln = PoaAlign(sc, seqs[i], g);
    PoaAddAlignment(g, aln, seqs[i], weights[i]);
  }
}

// ---------------------------------------------------------------------------------------
// POA DAG -> multiallelic variants: caller/variant_extractor.cpp:24-233,
// caller/variant_bubble.cpp:16-116, caller/raw_variant.cpp:44-77.
// ---------------------------------------------------------------------------------------
namespace {

AlleleType ClassifyVariant(std::string_view r, std::string_view a) {  // raw_variant.cpp:44-77
  usize s = 0;
  while (s < r.size() && s < a.size() && r[s] == a[s]) s++;
  if (s == r.size() && s == a.size()) return T_REF;
  usize e = 0;
  while (e < (r.size() - s) && e < (a.size() - s) && r[r.size() - 1 - e] == a[a.size() - 1 - e]) e++;
  usize const rc = r.size() - s - e, ac = a.size() - s - e;
  if (rc == 0 && ac > 0) return T_INS;
  if (rc > 0 && ac == 0) return T_DEL;
  if (rc == 0 || ac == 0) return T_REF;
  if (rc != ac) return T_CPX;
  return rc == 1 ? T_SNV : T_MNP;
}

i64 CalculateVariantLength(std::string_view r, std::string_view a, AlleleType t) {  // variant_bubble.cpp:16-47
  if (t == T_SNV) return 1;
  i64 const rl = static_cast<i64>(r.size()), al = static_cast<i64>(a.size());
  if (t == T_INS || t == T_DEL || t == T_CPX) return al - rl;
  i64 s = 0;
  while (s < rl && s < al && r[s] == a[s]) s++;
  i64 e = 0;
  while (e < (rl - s) && e < (al - s) && r[rl - 1 - e] == a[al - 1 - e]) e++;
  return al - s - e;
}

}  // namespace

std::vector<RawVariant> ExtractVariants(const PoaGraph& g, u64 ref_anchor_pos1) {
  std::vector<RawVariant> out;
  usize const ns = g.seq_first.size();
  if (ns < 2) return out;  // variant_set.cpp:13
  std::vector<u32> node_to_rank(g.nodes.size(), 0xFFFFFFFFu);
  for (u32 r = 0; r < g.rank_to_node.size(); ++r) node_to_rank[g.rank_to_node[r]] = r;
  std::vector<i32> active(g.seq_first.begin(), g.seq_first.end());
  std::vector<usize> hap_pos(ns, 0);
  u64 ref_pos = ref_anchor_pos1;
  i32 prev_match = -1;

  auto converged = [&]() {
    for (usize i = 1; i < ns; ++i)
      if (active[i] != active[0]) return false;
    return true;
  };
  std::map<std::pair<u64, std::string>, usize> dedupe;  // btree_set<RawVariant> uniqueness proxy

  while (true) {
    if (converged()) {
      if (active[0] < 0) break;
      // AdvanceConvergedPaths (variant_extractor.cpp:84-94)
      prev_match = active[0];
      for (usize i = 0; i < ns; ++i)
        if (active[i] >= 0) {
          active[i] = g.Successor(static_cast<u32>(active[i]), static_cast<u32>(i));
          hap_pos[i]++;
        }
      ref_pos++;
      continue;
    }
    // EatTopologicalBubble (variant_extractor.cpp:97-117)
    std::vector<std::string> raw(ns);
    std::vector<usize> starts(ns, 0);
    bool const has_prev = prev_match >= 0;
    usize const aoff = has_prev ? 1 : 0;
    u64 const bubble_start = ref_pos - aoff;
    if (has_prev) {
      char const c = static_cast<char>(g.decoder[g.nodes[prev_match].code]);
      for (auto& a : raw) a += c;
    }
    for (usize i = 0; i < ns; ++i) starts[i] = hap_pos[i] - aoff;
    // SinkPointers (variant_extractor.cpp:151-181)
    while (!converged()) {
      u32 min_rank = 0xFFFFFFFFu;
      for (i32 p : active)
        if (p >= 0) min_rank = std::min(min_rank, node_to_rank[p]);
      if (min_rank == 0xFFFFFFFFu) break;
      for (usize i = 0; i < ns; ++i) {
        if (active[i] >= 0 && node_to_rank[active[i]] == min_rank) {
          raw[i] += static_cast<char>(g.decoder[g.nodes[active[i]].code]);
          active[i] = g.Successor(static_cast<u32>(active[i]), static_cast<u32>(i));
          hap_pos[i]++;
          if (i == 0) ref_pos++;
        }
      }
    }
    // CreateNormalizedBubble (variant_extractor.cpp:184-199) + NormalizeVcfParsimony
    std::string ref_allele = raw[0];
    std::map<std::string, std::vector<usize>> alts;
    for (usize i = 1; i < ns; ++i)
      if (raw[i] != ref_allele) alts[raw[i]].push_back(i);
    u64 start_pos = bubble_start;
    if (!alts.empty() && !ref_allele.empty()) {  // variant_bubble.cpp:89-116
      auto trim = [&](bool right) {
        while (ref_allele.size() > 1) {
          bool ok = true;
          for (auto const& kv : alts) {
            auto const& a = kv.first;
            ok &= a.size() > 1 && (right ? a.back() == ref_allele.back() : a.front() == ref_allele.front());
          }
          if (!ok) break;
          if (right) ref_allele.pop_back(); else ref_allele.erase(0, 1);
          std::map<std::string, std::vector<usize>> re;
          for (auto& kv : alts) {
            std::string na = kv.first;
            if (right) na.pop_back(); else na.erase(0, 1);
            re.emplace(std::move(na), std::move(kv.second));
          }
          alts = std::move(re);
        }
      };
      trim(true);
      usize const init_len = ref_allele.size();
      trim(false);
      start_pos += (init_len - ref_allele.size());
    }
    if (alts.empty()) continue;
    // AssembleMultiallelicVariant (variant_extractor.cpp:202-231); NB hap starts are the
    // un-trimmed bubble starts (variant_extractor.cpp:111).
    RawVariant v;
    v.pos1 = start_pos;
    v.ref_start0 = starts[0];
    v.ref = ref_allele;
    for (auto const& kv : alts) {  // std::map iterates in sequence order == the final sort
      AltAllele a;
      a.seq = kv.first;
      a.type = ClassifyVariant(v.ref, a.seq);
      a.length = CalculateVariantLength(v.ref, a.seq, a.type);
      for (usize h : kv.second) a.hap_starts.push_back({static_cast<u32>(h), static_cast<u32>(starts[h])});
      v.alts.push_back(std::move(a));
    }
    out.push_back(std::move(v));
  }
  // absl::btree_set<RawVariant> ordering (raw_variant.h:108-118): pos, ref, alts
  std::stable_sort(out.begin(), out.end(), [](const RawVariant& a, const RawVariant& b) {
    if (a.pos1 != b.pos1) return a.pos1 < b.pos1;
    if (a.ref != b.ref) return a.ref < b.ref;
    std::vector<std::string> sa, sb;
    for (auto const& x : a.alts) sa.push_back(x.seq);
    for (auto const& x : b.alts) sb.push_back(x.seq);
    return sa < sb;
  });
  (void)dedupe;
  return out;
}

}  // namespace orc
