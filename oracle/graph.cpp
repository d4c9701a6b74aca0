// ORACLE -- TEST INFRASTRUCTURE ONLY (see common.hpp header).
// Restatement of Lancet2's colored bidirected de Bruijn graph assembly
// (cbdg/graph.cpp and friends).  CANONICAL ORDER: wherever the reference
// iterates its absl::flat_hash_map node table, this restatement iterates nodes in
// FIRST-INSERTION order (reference k-mers in reference order, then reads in the given
// order); every sort is stable.
#include "oracle.hpp"

#include <cstdio>
#include <cstdlib>
#include <deque>
#include <map>
#include <optional>
#include <unordered_set>

namespace orc {

// test instrumentation (tests/: which control-flow branches a crafted window reached): [0] cycle found, [1] complexity
// gate fired, [2] traversal limit hit -- counted over every k attempt since the last reset
unsigned long long g_debug_counters[4] = {0, 0, 0, 0};


namespace {

enum EdgeKind : u8 { PP = 0, PM = 1, MP = 2, MM = 3 };  // cbdg/kmer.h:12
constexpr u8 PLUS = 1, MINUS = 0;                        // cbdg/kmer.h:25

inline EdgeKind MakeFwdEdgeKind(u8 s, u8 d) {  // kmer.h:66-75
  return s == PLUS ? (d == PLUS ? PP : PM) : (d == PLUS ? MP : MM);
}
inline u8 SrcSignOf(u8 kind) { return (kind == PP || kind == PM) ? PLUS : MINUS; }  // kmer.h:77-92
inline u8 DstSignOf(u8 kind) { return (kind == PP || kind == MP) ? PLUS : MINUS; }
inline u8 RevEdgeKind(u8 kind) { return kind == PP ? static_cast<u8>(MM) : (kind == MM ? static_cast<u8>(PP) : kind); }  // kmer.h:94-105
inline u8 RevSign(u8 s) { return s == PLUS ? MINUS : PLUS; }

struct Edge {  // cbdg/edge.h:12-58 (node ids replaced by first-insertion indices)
  u32 src, dst;
  u8 kind;
  bool operator==(const Edge& o) const { return src == o.src && dst == o.dst && kind == o.kind; }
  bool IsSelfLoop() const { return src == dst; }
  Edge Mirror() const { return Edge{dst, src, RevEdgeKind(kind)}; }
  u8 SrcSign() const { return SrcSignOf(kind); }
  u8 DstSign() const { return DstSignOf(kind); }
};

struct Node {  // cbdg/node.h:40-170
  std::vector<Edge> edges;
  std::string seq;  // Kmer::mDfltSeq
  u64 id = 0;
  u8 sign = PLUS;  // Kmer::mDfltSign
  u32 comp = 0;
  std::vector<u32> counts;
  u32 role_counts[2] = {0, 0};
  u8 label = 0;
  bool alive = true;

  void EmplaceEdge(const Edge& e) {  // node.h:59-64
    if (std::find(edges.begin(), edges.end(), e) == edges.end()) edges.push_back(e);
  }
  void EraseEdge(const Edge& e) {  // node.h:66-71
    auto it = std::find(edges.begin(), edges.end(), e);
    if (it != edges.end()) edges.erase(it);
  }
  u8 SignFor(bool dflt) const { return dflt ? sign : RevSign(sign); }  // kmer.cpp:141-143
  std::string SequenceFor(bool dflt) const { return dflt ? seq : RevComp(seq); }  // kmer.cpp:145-147
  u32 Total() const {  // node.cpp:30-32
    u32 t = 0;
    for (u32 c : counts) t += c;
    return t;
  }
  bool IsAllSingletons() const {  // node.cpp:38-42
    bool any = false, all = true;
    for (u32 c : counts) {
      any |= c > 0;
      all &= c <= 1;
    }
    return any && all;
  }
  u32 Confidence(usize num_samples) const {  // node.cpp:59-79
    if (IsAllSingletons()) return 1;
    u32 const total = Total();
    if (total == 0) return 0;
    i64 confirming = 0;
    for (u32 c : counts) confirming += (c > 0);
    f64 const denom = static_cast<f64>(std::max<usize>(num_samples, 1));
    f64 const concordance = static_cast<f64>(confirming) / denom;
    u32 const bonus = (label & L_REFERENCE) ? 1u : 0u;
    return static_cast<u32>(static_cast<f64>(total) * concordance) + bonus;
  }
  bool HasSelfLoop() const {  // node.cpp:114-116
    for (auto const& e : edges)
      if (e.IsSelfLoop()) return true;
    return false;
  }
  std::vector<Edge> EdgesInDirection(bool dflt) const {  // node.cpp:118-127
    std::vector<Edge> r;
    u8 const want = SignFor(dflt);
    for (auto const& e : edges)
      if (e.SrcSign() == want) r.push_back(e);
    return r;
  }
};

// cbdg/kmer.cpp:17-28
bool IsCanonicallyPlus(std::string_view s) {
  usize const len = s.size(), half = (len + 1) / 2;
  for (usize i = 0; i < half; ++i) {
    char const f = s[i], r = Complement(s[len - 1 - i]);
    if (f < r) return true;
    if (f > r) return false;
  }
  return true;
}

// cbdg/kmer.cpp:48-109 MergeCords
void MergeCords(std::string& k1, std::string_view k2, u8 kind, usize k) {
  auto non_ovl_suffix = [&](std::string_view d) { return d.substr(k - 1, d.size() - k + 1); };
  auto non_ovl_prefix = [&](std::string_view d) { return d.substr(0, d.size() - k + 1); };
  switch (kind) {
    case PP: k1.append(non_ovl_suffix(k2)); break;
    case PM: {
      std::string rc = RevComp(k2);
      k1.append(non_ovl_suffix(rc));
      break;
    }
    case MP: {
      std::string rc = RevComp(k2);
      k1.insert(0, non_ovl_prefix(rc));
      break;
    }
    default: k1.insert(0, non_ovl_prefix(k2)); break;
  }
}

struct TraversalIndex {  // cbdg/traversal_index.h:81-140
  struct OutEdge { u32 dst_state, ordinal; };
  struct Range { u32 start = 0, count = 0; };
  std::vector<Range> ranges;
  std::vector<OutEdge> adj;
  std::vector<Edge> orig_edges;
  std::vector<u32> nodes;  // flat idx -> node idx
  u32 src_state = 0, snk_node = 0;
  static u32 MakeState(u32 n, u8 sign) { return n * 2 + (sign == PLUS ? 0 : 1); }
  bool IsSinkState(u32 s) const { return s / 2 == snk_node; }
};

class Graph {
 public:
  Graph(std::string_view ref, const std::vector<Read>& reads, const Params& prm)
      : mRef(ref), mReads(reads), mPrm(prm) {}

  AssemblyResult Run();

 private:
  std::string_view mRef;
  const std::vector<Read>& mReads;
  Params mPrm;
  usize mK = 0;
  std::vector<Node> mNodes;                 // first-insertion order == canonical order
  std::unordered_map<u64, u32> mIdToIdx;
  std::vector<u32> mRefNodeIdx;             // mRefNodeIds
  i64 mSource = -1, mSink = -1;             // mSourceAndSinkIds ({0,0} == none)

  std::vector<u32> AddNodes(std::string_view seq, u8 label);
  void BuildGraph();
  void RemoveNode(u32 idx);
  void RemoveLowCovNodes(u32 comp);
  struct CompInfo { u32 id, n; };
  std::vector<CompInfo> MarkConnectedComponents();
  struct Anchor { u32 node = 0; usize off = 0; bool found = false; };
  Anchor FindSource(u32 comp) const;
  Anchor FindSink(u32 comp) const;
  void PruneComponent(u32 comp);
  void CompressGraph(u32 comp);
  void CompressNode(u32 nid, bool dflt, std::vector<u8>& absorbed);
  std::optional<Edge> FindCompressibleEdge(const Node& src, u32 src_idx, bool dflt) const;
  bool IsPotentialBuddyEdge(const Node& src, const Edge& conn) const;
  void RemoveTips(u32 comp);
  void MergeNode(Node& dst, const Node& other, u8 kind);
  TraversalIndex BuildTraversalIndex(u32 comp) const;
  static bool HasCycle(const TraversalIndex& idx);
  GraphComplexity ComputeComplexity(u32 comp) const;
  std::vector<Haplotype> BuildHaplotypes(u32 comp, const TraversalIndex& idx,
                                         std::string_view ref_anchor, bool* hit_limit) const;
  Haplotype BuildRefHaplotype(u32 comp, std::string_view ref_anchor) const;
};

// cbdg/graph.cpp:311-341
std::vector<u32> Graph::AddNodes(std::string_view seq, u8 label) {
  std::vector<u32> result;
  if (seq.size() < mK + 1) return result;  // SlidingView(seq, k+1) empty
  usize const n_kp1 = seq.size() - (mK + 1) + 1;
  auto get_or_add = [&](std::string_view mer) -> u32 {
    bool const plus = IsCanonicallyPlus(mer);  // kmer.cpp:115-128
    std::string canon = plus ? std::string(mer) : RevComp(mer);
    u64 const id = HashStr64(canon);
    auto it = mIdToIdx.find(id);
    if (it != mIdToIdx.end()) return it->second;  // try_emplace keeps the first inserter
    Node nd;
    nd.seq = std::move(canon);
    nd.id = id;
    nd.sign = plus ? PLUS : MINUS;
    nd.label = label;
    nd.counts.assign(mPrm.num_samples, 0);
    u32 const idx = static_cast<u32>(mNodes.size());
    mNodes.push_back(std::move(nd));
    mIdToIdx.emplace(id, idx);
    return idx;
  };
  for (usize m = 0; m < n_kp1; ++m) {
    u32 const first = get_or_add(seq.substr(m, mK));
    u32 const second = get_or_add(seq.substr(m + 1, mK));
    if (m == 0) result.push_back(first);
    // NB the edge kind comes from the STORED signs of the two nodes (first inserter's
    // orientation), not from this occurrence: graph.cpp:333-336.
    u8 const fwd = MakeFwdEdgeKind(mNodes[first].sign, mNodes[second].sign);
    mNodes[first].EmplaceEdge(Edge{first, second, fwd});
    mNodes[second].EmplaceEdge(Edge{second, first, RevEdgeKind(fwd)});
    result.push_back(second);
  }
  return result;
}

// cbdg/graph.cpp:262-309
void Graph::BuildGraph() {
  mRefNodeIdx = AddNodes(mRef, L_REFERENCE);
  struct MateMer {
    u32 qname;
    u32 node;
    u8 tag;
    bool operator==(const MateMer& o) const { return qname == o.qname && node == o.node && tag == o.tag; }
  };
  struct MMHash {
    usize operator()(const MateMer& m) const {
      return Fmix64((static_cast<u64>(m.qname) << 32) ^ (static_cast<u64>(m.node) << 2) ^ m.tag);
    }
  };
  std::unordered_set<MateMer, MMHash> mate_mers;
  std::vector<f64> prefix;
  for (auto const& rd : mReads) {
    if (!rd.pass) continue;
    // graph.cpp:280-285: sequential f64 prefix sums of Phred error probabilities
    prefix.assign(rd.seq.size() + 1, 0.0);
    f64 acc = 0.0;
    for (usize i = 0; i < rd.seq.size(); ++i) {
      f64 const p = PhredToErrorProb(rd.qual[i]);
      acc = (i == 0) ? p : acc + p;  // std::partial_sum
      prefix[i + 1] = acc;
    }
    usize offset = 0;
    auto const added = AddNodes(rd.seq, rd.Tag());
    for (u32 nidx : added) {
      f64 const raw = prefix[offset + mK] - prefix[offset];
      i64 const expected_error = static_cast<i64>(std::floor(raw));
      offset++;
      MateMer mm{rd.qname_id, nidx, static_cast<u8>(rd.Tag())};
      if (expected_error > 0 || mate_mers.count(mm)) continue;
      Node& nd = mNodes[nidx];  // node.cpp:18-24
      if (rd.sample >= nd.counts.size()) nd.counts.resize(rd.sample + 1, 0);
      nd.counts[rd.sample] += 1;
      nd.role_counts[rd.role == 1 ? 1 : 0] += 1;
      mate_mers.insert(mm);
    }
  }
}

// cbdg/graph.cpp:347-361
void Graph::RemoveNode(u32 idx) {
  Node& nd = mNodes[idx];
  if (!nd.alive) return;
  for (auto const& conn : nd.edges) {
    if (conn.IsSelfLoop()) continue;
    Node& nb = mNodes[conn.dst];
    if (nb.alive) nb.EraseEdge(conn.Mirror());
  }
  nd.alive = false;
  nd.edges.clear();
}

// cbdg/graph.cpp:363-390
void Graph::RemoveLowCovNodes(u32 comp) {
  std::vector<u32> rm;
  for (u32 i = 0; i < mNodes.size(); ++i) {
    Node const& nd = mNodes[i];
    if (!nd.alive || nd.comp != comp) continue;
    if (static_cast<i64>(i) == mSource || static_cast<i64>(i) == mSink) continue;
    if (nd.IsAllSingletons() || nd.Total() < mPrm.min_node_cov) rm.push_back(i);
  }
  for (u32 i : rm) RemoveNode(i);
}

// cbdg/graph.cpp:392-463 (component ids in discovery order over canonical node order; the
// size sort is made stable: ties keep the lower component id first)
std::vector<Graph::CompInfo> Graph::MarkConnectedComponents() {
  std::vector<CompInfo> info;
  u32 current = 0;
  for (u32 i = 0; i < mNodes.size(); ++i) {
    if (!mNodes[i].alive || mNodes[i].comp != 0) continue;
    current++;
    info.push_back({current, 0});
    std::deque<u32> q;
    q.push_back(i);
    while (!q.empty()) {
      u32 const cur = q.front();
      q.pop_front();
      if (mNodes[cur].comp != 0) continue;
      mNodes[cur].comp = current;
      info[current - 1].n += 1;
      for (auto const& e : mNodes[cur].edges) q.push_back(e.dst);
    }
  }
  std::stable_sort(info.begin(), info.end(), [](const CompInfo& a, const CompInfo& b) { return a.n > b.n; });
  return info;
}

// cbdg/graph.cpp:469-488
Graph::Anchor Graph::FindSource(u32 comp) const {
  Anchor a;
  for (usize r = 0; r < mRefNodeIdx.size(); ++r) {
    Node const& nd = mNodes[mRefNodeIdx[r]];
    if (!nd.alive) continue;
    if (nd.comp != comp || nd.Total() < mPrm.min_anchor_cov) continue;
    a.node = mRefNodeIdx[r];
    a.off = r;
    a.found = true;
    break;
  }
  return a;
}

// cbdg/graph.cpp:490-509
Graph::Anchor Graph::FindSink(u32 comp) const {
  Anchor a;
  for (i64 r = static_cast<i64>(mRefNodeIdx.size()) - 1; r >= 0; --r) {
    Node const& nd = mNodes[mRefNodeIdx[r]];
    if (!nd.alive) continue;
    if (nd.comp != comp || nd.Total() < mPrm.min_anchor_cov) continue;
    a.node = mRefNodeIdx[r];
    a.off = static_cast<usize>(r);
    a.found = true;
    break;
  }
  return a;
}

// cbdg/node.cpp:81-112 (+ kmer.cpp:130-139)
void Graph::MergeNode(Node& dst, const Node& other, u8 kind) {
  MergeCords(dst.seq, other.seq, kind, mK);
  dst.label |= other.label;
  usize const max_size = std::max(dst.counts.size(), other.counts.size());
  dst.counts.resize(max_size, 0);
  u64 const this_len = dst.seq.size();  // NB: length AFTER the merge (node.cpp:91)
  u64 const other_len = other.seq.size();
  u64 const total_len = this_len + other_len;
  for (usize i = 0; i < max_size; ++i) {
    u64 const a = dst.counts[i];
    u64 const b = i < other.counts.size() ? other.counts[i] : 0;
    dst.counts[i] = static_cast<u32>((a * this_len + b * other_len) / total_len);
  }
  for (int r = 0; r < 2; ++r) {
    u64 const a = dst.role_counts[r], b = other.role_counts[r];
    dst.role_counts[r] = static_cast<u32>((a * this_len + b * other_len) / total_len);
  }
}

// cbdg/graph.cpp:758-799
bool Graph::IsPotentialBuddyEdge(const Node& src, const Edge& conn) const {
  Node const& nb = mNodes[conn.dst];
  if (src.edges.size() == 1 && nb.edges.size() == 1) {
    if (src.edges[0].dst == conn.dst && nb.edges[0].dst == conn.src) return false;
  }
  if (nb.edges.size() > 2 || nb.edges.empty() || nb.HasSelfLoop()) return false;
  Edge const expected = conn.Mirror();
  bool const dir_dflt = expected.SrcSign() == nb.SignFor(true);
  auto const in_dir = nb.EdgesInDirection(dir_dflt);
  if (in_dir.size() != 1 || !(in_dir[0] == expected)) return false;
  auto const opp = nb.EdgesInDirection(!dir_dflt);
  if (opp.size() != 1 || opp[0].dst == conn.src) return false;
  return mNodes[opp[0].dst].edges.size() <= 2;
}

// cbdg/graph.cpp:688-717
std::optional<Edge> Graph::FindCompressibleEdge(const Node& src, u32 src_idx, bool dflt) const {
  if (src.edges.size() > 2 || src.edges.empty() || src.HasSelfLoop()) return std::nullopt;
  if (static_cast<i64>(src_idx) == mSource || static_cast<i64>(src_idx) == mSink) return std::nullopt;
  auto const mergeable = src.EdgesInDirection(dflt);
  if (mergeable.size() != 1) return std::nullopt;
  Edge const cand = mergeable[0];
  if (static_cast<i64>(cand.dst) == mSource || static_cast<i64>(cand.dst) == mSink) return std::nullopt;
  if (!IsPotentialBuddyEdge(src, cand)) return std::nullopt;
  auto const opp = src.EdgesInDirection(!dflt);
  if (opp.empty()) return cand;
  if (opp.size() > 1) return std::nullopt;
  if (!IsPotentialBuddyEdge(src, opp[0])) return std::nullopt;
  return cand;
}

// cbdg/graph.cpp:600-645
void Graph::CompressNode(u32 nid, bool dflt, std::vector<u8>& absorbed) {
  auto ce = FindCompressibleEdge(mNodes[nid], nid, dflt);
  while (ce.has_value()) {
    Edge const src2obdy = *ce;
    u32 const ob = src2obdy.dst;
    MergeNode(mNodes[nid], mNodes[ob], src2obdy.kind);
    mNodes[nid].EraseEdge(src2obdy);
    u8 const rev_src_sign = RevSign(src2obdy.SrcSign());
    std::vector<Edge> const ob_edges = mNodes[ob].edges;  // buddy's list is not mutated below
    for (auto const& ob2nb : ob_edges) {
      if (ob2nb == src2obdy.Mirror()) continue;
      u8 const ne_src_sign = src2obdy.DstSign() != ob2nb.SrcSign() ? rev_src_sign : src2obdy.SrcSign();
      Edge const src2nb{nid, ob2nb.dst, MakeFwdEdgeKind(ne_src_sign, ob2nb.DstSign())};
      mNodes[nid].EmplaceEdge(src2nb);
      mNodes[ob2nb.dst].EmplaceEdge(src2nb.Mirror());
      mNodes[ob2nb.dst].EraseEdge(ob2nb.Mirror());
    }
    absorbed[ob] = 1;
    ce = FindCompressibleEdge(mNodes[nid], nid, dflt);
  }
}

// cbdg/graph.cpp:558-576
void Graph::CompressGraph(u32 comp) {
  std::vector<u8> absorbed(mNodes.size(), 0);
  for (u32 i = 0; i < mNodes.size(); ++i) {
    if (!mNodes[i].alive || mNodes[i].comp != comp) continue;
    if (absorbed[i]) continue;
    CompressNode(i, true, absorbed);
    CompressNode(i, false, absorbed);
  }
  for (u32 i = 0; i < mNodes.size(); ++i)
    if (absorbed[i]) RemoveNode(i);
}

// cbdg/graph.cpp:801-840
void Graph::RemoveTips(u32 comp) {
  usize current_tips = 1;
  while (current_tips > 0) {
    std::vector<u32> rm;
    for (u32 i = 0; i < mNodes.size(); ++i) {
      Node const& nd = mNodes[i];
      if (!nd.alive) continue;
      bool const is_anchor = static_cast<i64>(i) == mSource || static_cast<i64>(i) == mSink;
      if (nd.comp != comp || is_anchor || nd.edges.size() > 1) continue;
      usize const uniq_len = nd.seq.size() - mK + 1;
      if (uniq_len >= mK) continue;
      rm.push_back(i);
    }
    if (!rm.empty()) {
      for (u32 i : rm) RemoveNode(i);
      CompressGraph(comp);
    }
    current_tips = rm.size();
  }
}

// cbdg/graph.cpp:515-540
void Graph::PruneComponent(u32 comp) {
  CompressGraph(comp);
  RemoveLowCovNodes(comp);
  CompressGraph(comp);
  RemoveTips(comp);
}

// cbdg/traversal_index.cpp:34-119
TraversalIndex Graph::BuildTraversalIndex(u32 comp) const {
  TraversalIndex t;
  std::unordered_map<u32, u32> to_flat;
  for (u32 i = 0; i < mNodes.size(); ++i) {
    if (!mNodes[i].alive || mNodes[i].comp != comp) continue;
    to_flat.emplace(i, static_cast<u32>(t.nodes.size()));
    t.nodes.push_back(i);
  }
  u32 const nn = static_cast<u32>(t.nodes.size());
  t.ranges.assign(nn * 2, {});
  for (u32 f = 0; f < nn; ++f)
    for (auto const& e : mNodes[t.nodes[f]].edges) {
      if (!to_flat.count(e.dst)) continue;
      t.ranges[TraversalIndex::MakeState(f, e.SrcSign())].count++;
    }
  u32 off = 0;
  for (auto& r : t.ranges) {
    r.start = off;
    off += r.count;
    r.count = 0;
  }
  t.adj.resize(off);
  for (u32 f = 0; f < nn; ++f)
    for (auto const& e : mNodes[t.nodes[f]].edges) {
      auto it = to_flat.find(e.dst);
      if (it == to_flat.end()) continue;
      u32 const ss = TraversalIndex::MakeState(f, e.SrcSign());
      u32 const ds = TraversalIndex::MakeState(it->second, e.DstSign());
      // every directed edge lives in exactly one node's (de-duplicated) list, so the
      // reference's edge_to_ordinal.emplace always inserts: ordinals are a running counter
      // over (canonical node order, edge-list order)  (traversal_index.cpp:92-103)
      u32 const ord = static_cast<u32>(t.orig_edges.size());
      t.orig_edges.push_back(e);
      auto& r = t.ranges[ss];
      t.adj[r.start + r.count] = {ds, ord};
      r.count++;
    }
  u32 const sf = to_flat.at(static_cast<u32>(mSource));
  t.src_state = TraversalIndex::MakeState(sf, mNodes[mSource].sign);
  t.snk_node = to_flat.at(static_cast<u32>(mSink));
  return t;
}

// cbdg/cycle_finder.cpp:55-100
bool Graph::HasCycle(const TraversalIndex& idx) {
  std::vector<u8> color(idx.ranges.size(), 0);
  struct Frame { u32 state, pos; };
  std::vector<Frame> st;
  color[idx.src_state] = 1;
  st.push_back({idx.src_state, 0});
  while (!st.empty()) {
    Frame& fr = st.back();
    auto const& r = idx.ranges[fr.state];
    if (fr.pos >= r.count) {
      color[fr.state] = 2;
      st.pop_back();
      continue;
    }
    auto const& out = idx.adj[r.start + fr.pos];
    fr.pos++;
    if (color[out.dst_state] == 1) return true;
    if (color[out.dst_state] != 0) continue;
    color[out.dst_state] = 1;
    st.push_back({out.dst_state, 0});
  }
  return false;
}

// cbdg/graph_complexity.cpp:16-93
GraphComplexity Graph::ComputeComplexity(u32 comp) const {
  GraphComplexity c;
  usize nn = 0, ne = 0, unitigs = 0;
  OnlineStats cov, tip, uni;
  for (auto const& nd : mNodes) {
    if (!nd.alive || nd.comp != comp) continue;
    nn++;
    usize d = 0, o = 0;
    for (auto const& e : nd.edges) (e.SrcSign() == nd.sign ? d : o)++;
    ne += d + o;
    c.max_dir_degree = std::max<u64>(c.max_dir_degree, std::max(d, o));
    if (d >= 2 || o >= 2) c.branch_points++;
    if (d == 1 && o == 1) unitigs++;
    f64 const cv = static_cast<f64>(nd.Total());
    cov.Add(cv);
    if (d == 0 || o == 0) tip.Add(cv);
    else if (d == 1 && o == 1) uni.Add(cv);
  }
  ne /= 2;
  c.cyclomatic = ne >= nn ? ne - nn + 1 : 0;
  c.unitig_ratio = nn > 0 ? static_cast<f64>(unitigs) / static_cast<f64>(nn) : 0.0;
  if (!cov.Empty() && cov.Mean() > 0.0) c.coverage_cv = cov.StdDev() / cov.Mean();
  if (!tip.Empty() && !uni.Empty() && uni.Mean() > 0.0) c.tip_to_path = tip.Mean() / uni.Mean();
  return c;
}

// cbdg/graph.cpp:902-924
Haplotype Graph::BuildRefHaplotype(u32 comp, std::string_view ref_anchor) const {
  std::vector<u32> confs;
  for (auto const& nd : mNodes) {
    if (!nd.alive || nd.comp != comp) continue;
    if (!(nd.label & L_REFERENCE)) continue;
    confs.push_back(nd.Confidence(mPrm.num_samples));
  }
  u32 const w = confs.empty() ? 1u : Median(confs);
  Haplotype h;
  h.seq = std::string(ref_anchor);
  h.node_weights.push_back({w, static_cast<u32>(ref_anchor.size())});
  h.Finalize();  // no node coverages -> all stats stay 0 (path.cpp:40)
  return h;
}

// cbdg/max_flow.cpp:162-280 + graph.cpp:846-891
std::vector<Haplotype> Graph::BuildHaplotypes(u32 comp, const TraversalIndex& idx,
                                              std::string_view ref_anchor, bool* hit_limit) const {
  std::vector<Haplotype> haps;
  std::vector<u8> traversed(idx.orig_edges.size(), 0);
  struct WalkNode { u32 ordinal, dst_state, parent, score; };
  constexpr u32 NO_PARENT = 0xFFFFFFFFu;
  *hit_limit = false;

  auto conf_of_state = [&](u32 st) { return mNodes[idx.nodes[st / 2]].Confidence(mPrm.num_samples); };

  while (true) {  // one NextPath() per iteration
    std::vector<WalkNode> arena;
    usize head = 0;  // FIFO frontier == arena indices in creation order
    auto enqueue = [&](u32 state, u32 parent, u32 pscore) {  // max_flow.cpp:235-280
      auto const& r = idx.ranges[state];
      if (r.count == 0) return;
      std::vector<TraversalIndex::OutEdge> outs(idx.adj.begin() + r.start, idx.adj.begin() + r.start + r.count);
      std::stable_sort(outs.begin(), outs.end(), [&](auto const& a, auto const& b) {
        return conf_of_state(a.dst_state) > conf_of_state(b.dst_state);
      });
      for (auto const& o : outs)
        if (!traversed[o.ordinal]) arena.push_back({o.ordinal, o.dst_state, parent, pscore + 1});
      for (auto const& o : outs)
        if (traversed[o.ordinal]) arena.push_back({o.ordinal, o.dst_state, parent, pscore});
    };
    enqueue(idx.src_state, NO_PARENT, 0);
    u32 nvisits = 0;
    i64 best = -1;
    while (head < arena.size()) {
      nvisits++;
      if (nvisits > mPrm.bfs_limit) {
        *hit_limit = true;
        g_debug_counters[2]++;
        break;
      }
      u32 const ai = static_cast<u32>(head++);
      WalkNode const wn = arena[ai];
      if (idx.IsSinkState(wn.dst_state)) {
        if (wn.score == 0) continue;
        best = ai;
        break;
      }
      enqueue(wn.dst_state, ai, wn.score);
    }
    if (best < 0) break;
    // ReconstructWalk (max_flow.cpp:42-54) + mark traversed
    std::vector<Edge> walk;
    for (u32 i = static_cast<u32>(best); i != NO_PARENT; i = arena[i].parent) {
      walk.push_back(idx.orig_edges[arena[i].ordinal]);
      traversed[arena[i].ordinal] = 1;
    }
    std::reverse(walk.begin(), walk.end());
    // BuildSequence (max_flow.cpp:64-113)
    Haplotype h;
    bool first = true;
    bool dflt = walk[0].SrcSign() == PLUS;
    for (auto const& conn : walk) {
      if (first) {
        Node const& s = mNodes[conn.src];
        std::string sq = s.SequenceFor(dflt);
        h.node_covs.push_back(s.Total());
        h.node_weights.push_back({s.Confidence(mPrm.num_samples), static_cast<u32>(sq.size())});
        h.seq += sq;
        first = false;
      }
      Node const& d = mNodes[conn.dst];
      dflt = conn.DstSign() == PLUS;
      std::string const dsq = d.SequenceFor(dflt);
      u32 const ulen = static_cast<u32>(dsq.size() - mK + 1);
      h.seq += dsq.substr(mK - 1, ulen);
      h.node_covs.push_back(d.Total());
      h.node_weights.push_back({d.Confidence(mPrm.num_samples), ulen});
    }
    h.Finalize();
    haps.push_back(std::move(h));
  }
  if (haps.empty()) return haps;
  std::stable_sort(haps.begin(), haps.end(),
                   [](const Haplotype& a, const Haplotype& b) { return a.MinWeight() > b.MinWeight(); });
  std::vector<Haplotype> kept;
  std::unordered_set<std::string> seen;
  for (auto& h : haps) {
    bool const inserted = seen.insert(h.seq).second;
    if (!inserted || h.seq == ref_anchor) continue;
    kept.push_back(std::move(h));
  }
  kept.insert(kept.begin(), BuildRefHaplotype(comp, ref_anchor));
  return kept;
}

// cbdg/graph.cpp:78-256
AssemblyResult Graph::Run() {
  AssemblyResult out;
  i64 k = static_cast<i64>(mPrm.min_k) - static_cast<i64>(mPrm.k_step);
  while (out.comps.empty() && (k + mPrm.k_step) <= static_cast<i64>(mPrm.max_k)) {
    k += mPrm.k_step;
    mK = static_cast<usize>(k);
    out.used_k = static_cast<u32>(mK);
    mSource = mSink = -1;
    if (HasRepeat(mRef, mK, mPrm.max_mismatch)) continue;  // graph.cpp:120
    mNodes.clear();
    mIdToIdx.clear();
    BuildGraph();
    static const bool dbg = getenv("ORC_DEBUG") != nullptr;
    if (dbg) fprintf(stderr, "[orc] k=%zu raw nodes=%zu refnodes=%zu\n", mK, mNodes.size(), mRefNodeIdx.size());
    RemoveLowCovNodes(0);
    auto const comps = MarkConnectedComponents();
    if (dbg) {
      usize alive = 0;
      for (auto const& nd : mNodes) alive += nd.alive;
      fprintf(stderr, "[orc] after lowcov1 alive=%zu comps=%zu (largest %u)\n", alive, comps.size(), comps.empty() ? 0 : comps[0].n);
    }
    bool retry = false, attempt_limit = false;
    for (auto const& ci : comps) {
      auto const src = FindSource(ci.id), snk = FindSink(ci.id);
      if (dbg) fprintf(stderr, "[orc] comp %u n=%u src=%d@%zu snk=%d@%zu\n", ci.id, ci.n, src.found, src.off, snk.found, snk.off);
      if (!src.found || !snk.found || src.node == snk.node) continue;
      usize const anchor_len = snk.off - src.off + mK;  // graph.h:182-185
      if (anchor_len < mPrm.min_anchor_len) continue;
      mSource = src.node;
      mSink = snk.node;
      std::string_view const ref_anchor = mRef.substr(src.off, anchor_len);
      PruneComponent(ci.id);
      auto const tidx = BuildTraversalIndex(ci.id);
      if (dbg) fprintf(stderr, "[orc] pruned: nodes=%zu edges=%zu\n", tidx.nodes.size(), tidx.orig_edges.size());
      if (HasCycle(tidx)) {
        g_debug_counters[0]++;
        if (dbg) fprintf(stderr, "[orc] cycle\n");
        retry = true;
        break;
      }
      auto const cx = ComputeComplexity(ci.id);
      if (cx.IsComplex()) {
        g_debug_counters[1]++;
        retry = true;
        break;
      }
      bool hit = false;
      auto haps = BuildHaplotypes(ci.id, tidx, ref_anchor, &hit);
      attempt_limit = attempt_limit || hit;
      if (dbg) fprintf(stderr, "[orc] haps=%zu cc=%llu bp=%llu\n", haps.size(), (unsigned long long)cx.cyclomatic, (unsigned long long)cx.branch_points);
      if (haps.empty()) continue;
      ComponentResult cr;
      cr.haps = std::move(haps);
      cr.metrics = cx;
      cr.anchor_start = static_cast<u32>(src.off);
      cr.hit_bfs_limit = hit;
      out.comps.push_back(std::move(cr));
    }
    if (retry) out.comps.clear();
    // MA_W_BFS_LIMIT: some component of the k attempt that produced the window's result stopped at the traversal cap --
    // whether or not that component itself still yielded walks (it is skipped when it did not, graph.cpp:225-226)
    out.hit_bfs_limit = !out.comps.empty() && attempt_limit;
  }
  return out;
}

}  // namespace

u32 Haplotype::MinWeight() const {
  if (node_weights.empty()) return 0;
  u32 m = node_weights[0].first;
  for (auto const& nw : node_weights) m = std::min(m, nw.first);
  return m;
}

std::vector<u32> Haplotype::PerBaseWeights() const {
  std::vector<u32> w;
  w.reserve(seq.size());
  for (auto const& nw : node_weights) w.insert(w.end(), nw.second, nw.first);
  return w;
}

void Haplotype::Finalize() {
  if (node_covs.empty()) return;
  OnlineStats st;
  for (u32 c : node_covs) st.Add(static_cast<f64>(c));
  mean_cov = st.Mean();
  sd_cov = st.StdDev();
  total_cov = mean_cov * static_cast<f64>(st.n);
  if (mean_cov > 0.0) cv_cov = sd_cov / mean_cov;
  median_cov = static_cast<f64>(Median(node_covs));
  if (node_covs.size() >= 4) {
    std::vector<u32> s = node_covs;
    std::sort(s.begin(), s.end());
    f64 const q1 = static_cast<f64>(s[s.size() / 4]);
    f64 const q3 = static_cast<f64>(s[(s.size() * 3) / 4]);
    if (q3 + q1 > 0.0) qcv_cov = (q3 - q1) / (q3 + q1);
  }
}

f64 ComponentResult::MaxAltPathCv() const {
  f64 mx = -1.0;
  bool has = false;
  for (usize h = 1; h < haps.size(); ++h) {
    mx = has ? std::max(mx, haps[h].cv_cov) : haps[h].cv_cov;
    has = true;
  }
  return has ? mx : -1.0;
}

AssemblyResult BuildComponentResults(std::string_view ref, const std::vector<Read>& reads,
                                     const Params& prm) {
  Graph g(ref, reads, prm);
  return g.Run();
}

}  // namespace orc

// ---- known-answer hooks for tests/test_oracle_kat.py (tests/cbdg/kmer_test.cpp, node.cpp:53-58) ----
extern "C" {

// tests/cbdg/kmer_test.cpp:212-248: fold the sliding k-mers of `seq` into one Kmer with
// Kmer::Merge, forward (reverse=0) or from the last k-mer backwards (reverse=1).
int orc_kmer_merge_chain(const char* seq, uint64_t n, uint64_t k, int reverse, char* out, uint64_t cap) {
  using namespace orc;
  std::string_view s(seq, n);
  if (n < k) return -1;
  struct Mer { std::string dflt; u8 sign; };
  std::vector<Mer> mers;
  for (usize i = 0; i + k <= n; ++i) {
    auto const m = s.substr(i, k);
    bool const plus = IsCanonicallyPlus(m);
    mers.push_back({plus ? std::string(m) : RevComp(m), static_cast<u8>(plus ? PLUS : MINUS)});
  }
  std::string merged;
  u8 msign = PLUS;  // default-constructed Kmer (kmer.h:63)
  bool empty = true;
  auto merge = [&](const Mer& other, u8 kind) {  // Kmer::Merge (kmer.cpp:130-139)
    if (empty) {
      merged = other.dflt;
      msign = other.sign;
      empty = false;
      return;
    }
    MergeCords(merged, other.dflt, kind, k);
  };
  if (!reverse) {
    for (auto const& m : mers) merge(m, MakeFwdEdgeKind(msign, m.sign));
  } else {
    for (auto it = mers.rbegin(); it != mers.rend(); ++it) merge(*it, RevEdgeKind(MakeFwdEdgeKind(it->sign, msign)));
  }
  if (merged.size() + 1 > cap) return -2;
  std::memcpy(out, merged.c_str(), merged.size() + 1);
  return static_cast<int>(merged.size());
}

// Node::Confidence (node.cpp:59-79)
uint32_t orc_confidence(const uint32_t* counts, int n, int num_samples, int is_ref) {
  orc::Node nd;
  nd.counts.assign(counts, counts + n);
  nd.label = is_ref ? orc::L_REFERENCE : orc::L_CTRL;
  return nd.Confidence(static_cast<orc::usize>(num_samples));
}

}  // extern "C"
