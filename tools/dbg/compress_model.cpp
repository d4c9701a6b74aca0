// DEVELOPER TOOL (not product, not shipped): CPU model of k_clean_chains' first unitig compaction, checked against the
// oracle's sequential CompressGraph on whole batches.  It exists to validate the THEORY the kernel is built on before any
// HIP is written: on the raw graph (every node one k-mer) the reference's order-dependent CompressGraph decomposes into
// independent SEGMENTS -- maximal paths of "fully plain" nodes between boundary nodes -- and inside a segment into turns
// taken in node-index order, each of which absorbs ALL units on one side (and, if the boundary it reaches is a buddy, all
// on the other).  See DESIGN.md "first compaction as an interval process".
//
//   g++ -O2 -std=c++17 -shared -fPIC -I include -I oracle tools/dbg/compress_model.cpp oracle/repeat.cpp -o /tmp/libcmodel.so
//   python tools/dbg/compress_model.py
#define private public
#include "../../oracle/graph.cpp"
#undef private
#include "../../include/microasm.h"
#include <array>

namespace orc {
namespace {

struct ModelStats {
  u64 windows = 0, comps = 0, mismatched = 0, punts = 0, nodes = 0, alive_after = 0, segments = 0, turns = 0, max_turns = 0,
      max_alive = 0, max_deg = 0, nested = 0, max_seg = 0, max_nodes = 0, noop_turns = 0, deg_gt4 = 0, str_bad = 0;
};

struct SideView {
  int cnt[2] = {0, 0};
  Edge e[2];  // the edge on side s when cnt[s] == 1
};

inline u32 src_minus(const Edge& e) { return SrcSignOf(e.kind) == PLUS ? 0u : 1u; }
inline u32 dst_minus(const Edge& e) { return DstSignOf(e.kind) == PLUS ? 0u : 1u; }

struct Unit {  // a maximal merged run of a segment (or a single raw node)
  u32 owner;
  u32 lo, hi;  // positions covered
};

// returns false when the component has a shape the interval model does not cover (the kernel would punt)
bool ModelCompress(std::vector<Node>& nodes, u32 comp, i64 source, i64 sink, usize K, Graph& g, ModelStats& st) {
  u32 const n = static_cast<u32>(nodes.size());
  std::vector<SideView> sv(n);
  std::vector<u8> selfloop(n, 0), plain(n, 0), inI(n, 0);
  auto in_comp = [&](u32 i) { return nodes[i].alive && nodes[i].comp == comp; };
  for (u32 i = 0; i < n; ++i) {
    if (!in_comp(i)) continue;
    st.max_deg = std::max<u64>(st.max_deg, nodes[i].edges.size());
    if (nodes[i].edges.size() > 4) st.deg_gt4++;
    for (auto const& e : nodes[i].edges) {
      if (e.dst == i) selfloop[i] = 1;
      u32 const s = src_minus(e);
      if (sv[i].cnt[s] == 0) sv[i].e[s] = e;
      sv[i].cnt[s]++;
      // mirror consistency of the raw graph (AddNodes always emplaces both; RemoveNode erases both)
      bool found = false;
      for (auto const& m : nodes[e.dst].edges) found |= (m == e.Mirror());
      if (!found) return false;
    }
  }
  for (u32 i = 0; i < n; ++i) {
    if (!in_comp(i)) continue;
    plain[i] = nodes[i].edges.size() == 2 && !selfloop[i] && sv[i].cnt[0] == 1 && sv[i].cnt[1] == 1;
  }
  for (u32 i = 0; i < n; ++i) {
    if (!in_comp(i) || !plain[i]) continue;
    if (static_cast<i64>(i) == source || static_cast<i64>(i) == sink) continue;
    u32 const a = sv[i].e[0].dst, b = sv[i].e[1].dst;
    if (a == b) return false;  // two-ring
    inI[i] = nodes[a].edges.size() <= 2 && nodes[b].edges.size() <= 2;
  }
  // ---- segments ----
  struct Seg {
    std::vector<u32> c;       // node at position p
    std::vector<u8> sideL;    // side bit of c[p] facing L (the other faces R)
    u32 L, R;
    bool bL, bR, tipL, tipR;
  };
  std::vector<Seg> segs;
  std::vector<u8> seen(n, 0);
  auto buddy_boundary = [&](u32 B, const Edge& into_B) {  // is_potential_buddy(unit, unit->B) for a boundary node B
    if (!plain[B]) return false;
    u32 const back = dst_minus(into_B) ^ 1u;  // side of B's edge back to the unit
    Edge const far = sv[B].e[back ^ 1u];
    return nodes[far.dst].edges.size() <= 2;
  };
  for (u32 i = 0; i < n; ++i) {
    if (!in_comp(i) || !inI[i] || seen[i]) continue;
    // is i an end of its segment?  (a neighbour outside I)
    bool const out0 = !inI[sv[i].e[0].dst], out1 = !inI[sv[i].e[1].dst];
    if (!out0 && !out1) continue;
    // walk from this end to the other; keep the orientation in which position 0 is the end with the smaller index
    auto walk = [&](u32 start, u32 side_out, std::vector<u32>& c, std::vector<u8>& sl) {
      u32 cur = start, so = side_out;
      while (true) {
        c.push_back(cur);
        sl.push_back(static_cast<u8>(so));
        Edge const on = sv[cur].e[so ^ 1u];
        if (!inI[on.dst]) break;
        u32 const j = dst_minus(on);  // arrives with bit j: leaves by side j, came in by side !j
        cur = on.dst;
        so = j ^ 1u;
        if (c.size() > n) return false;
      }
      return true;
    };
    Seg s;
    u32 const so = out0 ? 0u : 1u;
    if (!walk(i, so, s.c, s.sideL)) return false;
    u32 const other = s.c.back();
    if (other < i) {  // the other end has the smaller index: it will (or did) define the segment
      if (seen[other]) continue;
      Seg t;
      u32 const so2 = s.sideL.back() ^ 1u;  // its outward side is the one facing away from the segment
      if (!walk(other, so2, t.c, t.sideL)) return false;
      s = t;
    }
    for (u32 x : s.c) {
      if (seen[x]) return false;
      seen[x] = 1;
    }
    u32 const m = static_cast<u32>(s.c.size());
    Edge const toL = sv[s.c[0]].e[s.sideL[0]], toR = sv[s.c[m - 1]].e[s.sideL[m - 1] ^ 1u];
    s.L = toL.dst;
    s.R = toR.dst;
    if (s.L == s.R) return false;
    s.bL = buddy_boundary(s.L, toL);
    s.bR = buddy_boundary(s.R, toR);
    auto tip = [&](u32 B) { return nodes[B].edges.size() == 1 && static_cast<i64>(B) != source && static_cast<i64>(B) != sink; };
    s.tipL = tip(s.L);
    s.tipR = tip(s.R);
    segs.push_back(std::move(s));
  }
  for (u32 i = 0; i < n; ++i)
    if (in_comp(i) && inI[i] && !seen[i]) return false;  // a ring of fully plain nodes

  // ---- the interval process, per segment; events are replayed on the node table afterwards ----
  struct Absorb { u32 walker, unit_owner; u8 kind; };  // Node::Merge(walker, unit, kind) in this order
  std::vector<std::vector<Absorb>> seg_events(segs.size());
  std::vector<u8> absorbed(n, 0);
  // what the kernel keeps per node: who absorbed it, where it sits in its segment, what its block covers
  constexpr u32 NONE = 0xFFFFFFFFu;
  std::vector<u32> abs_of(n, NONE), nseg(n, NONE);
  std::vector<i64> npos(n, 0), blo(n, 0), bhi(n, 0);
  std::vector<u8> side_l(n, 0), owns(n, 0);
  // final edges: (node, index of the raw edge it replaces in the node's raw list, new value, time of the last rewrite)
  struct FinalEdge { u32 node; Edge old_raw, now; i64 key; };
  std::vector<FinalEdge> final_edges;
  auto key_of = [](u32 walker, u32 pass, u32 step) { return (static_cast<i64>(walker) << 24) | (static_cast<i64>(pass) << 20) | step; };

  for (usize si = 0; si < segs.size(); ++si) {
    Seg const& s = segs[si];
    i64 const m = static_cast<i64>(s.c.size());
    st.segments++;
    st.max_seg = std::max<u64>(st.max_seg, m);
    auto sideR = [&](i64 p) { return static_cast<u32>(s.sideL[p] ^ 1u); };
    auto sideLf = [&](i64 p) { return static_cast<u32>(s.sideL[p]); };
    // state: left block [0..a-1] owned by the raw node at position LBp (a > 0), singles [a..b], right block [b+1..m-1] (RBp)
    i64 a = 0, b = m - 1, LBp = -1, RBp = -1;
    i64 keyL = -1, keyR = -1;        // time of the last rewrite of the edge L -> segment / R -> segment
    std::vector<i64> keyOwnL(m, -1), keyOwnR(m, -1);  // per position (owner): last rewrite of its own L-facing / R-facing edge
    i64 tip_owner = -1;              // 1: L swallowed everything, 2: R did
    i64 tip_key = -1;
    i64 clock = -1;
    bool tipL_done = !s.tipL, tipR_done = !s.tipR;
    u64 turns = 0;
    for (i64 p = 0; p < m; ++p) {
      npos[s.c[p]] = blo[s.c[p]] = bhi[s.c[p]] = p;
      nseg[s.c[p]] = static_cast<u32>(si);
      side_l[s.c[p]] = s.sideL[p];
    }
    while (tip_owner < 0) {
      i64 best = -1, best_pos = -1;
      int who = 0;  // 0 single, 1 tip L, 2 tip R
      for (i64 p = a; p <= b; ++p) {
        i64 const id = s.c[p];
        if (id > clock && (best < 0 || id < best)) { best = id; best_pos = p; who = 0; }
      }
      if (!tipL_done && static_cast<i64>(s.L) > clock && (best < 0 || static_cast<i64>(s.L) < best)) { best = s.L; who = 1; }
      if (!tipR_done && static_cast<i64>(s.R) > clock && (best < 0 || static_cast<i64>(s.R) < best)) { best = s.R; who = 2; }
      if (best < 0) break;
      clock = best;
      if (who != 0) {
        // a tip end swallows every unit of the segment, nearest first (it has no other side to check)
        (who == 1 ? tipL_done : tipR_done) = true;
        u32 const T = who == 1 ? s.L : s.R;
        u32 const sideT = src_minus(nodes[T].edges[0]);
        u32 step = 0;
        auto absorb = [&](i64 owner_pos, u32 arrival_bit) {
          seg_events[si].push_back({T, s.c[owner_pos], static_cast<u8>((sideT << 1) | arrival_bit)});
          absorbed[s.c[owner_pos]] = 1;
          abs_of[s.c[owner_pos]] = T;
          ++step;
        };
        if (who == 1) {
          if (a > 0) absorb(LBp, sideLf(LBp) ^ 1u);
          for (i64 p = a; p <= b; ++p) absorb(p, sideLf(p) ^ 1u);
          if (b < m - 1) absorb(RBp, sideLf(RBp) ^ 1u);
        } else {
          if (b < m - 1) absorb(RBp, sideR(RBp) ^ 1u);
          for (i64 p = b; p >= a; --p) absorb(p, sideR(p) ^ 1u);
          if (a > 0) absorb(LBp, sideR(LBp) ^ 1u);
        }
        tip_owner = who;
        tip_key = key_of(T, 0, step);
        owns[T] = 1;
        nseg[T] = static_cast<u32>(si);
        npos[T] = who == 1 ? -1 : m;
        blo[T] = who == 1 ? -1 : 0;
        bhi[T] = who == 1 ? m - 1 : m;
        side_l[T] = static_cast<u8>(who == 1 ? (sideT ^ 1u) : sideT);
        st.turns++;
        turns++;
        break;
      }
      i64 const j = best_pos;
      u32 const x = s.c[j];
      bool const f_right = (nodes[x].sign == PLUS ? 0u : 1u) == sideR(j);  // CompressNode(x, dflt = true) first
      bool const has_l = j > 0, has_r = j < m - 1;  // units on either side (j in [a..b]: everything else is a unit)
      bool walked_f = false, walked_any = false;
      for (u32 pass = 0; pass < 2; ++pass) {
        bool const right = pass == 0 ? f_right : !f_right;
        bool const cand_units = right ? has_r : has_l;
        if (!cand_units) continue;
        bool opp_ok;
        if (pass == 0) {
          opp_ok = (right ? has_l : has_r) || (right ? s.bL : s.bR);
        } else {
          // the other side now ends at its boundary if the first walk happened or there was nothing there to begin with
          bool const other_had_units = right ? has_l : has_r;
          if (other_had_units && !walked_f) continue;  // (cannot be reached: then this side has no units)
          opp_ok = right ? s.bL : s.bR;
        }
        if (!opp_ok) continue;
        u32 step = 0;
        if (right) {
          for (i64 p = j + 1; p <= b; ++p) {
            seg_events[si].push_back({x, s.c[p], static_cast<u8>((sideR(j) << 1) | (sideLf(p) ^ 1u))});
            absorbed[s.c[p]] = 1;
            abs_of[s.c[p]] = x;
            ++step;
          }
          if (b < m - 1) {
            seg_events[si].push_back({x, s.c[RBp], static_cast<u8>((sideR(j) << 1) | (sideLf(RBp) ^ 1u))});
            absorbed[s.c[RBp]] = 1;
            abs_of[s.c[RBp]] = x;
            ++step;
            st.nested++;
          }
          keyOwnR[j] = key_of(x, pass, step);
          owns[x] = 1;
          bhi[x] = m - 1;
          keyR = key_of(x, pass, step);
          b = j - 1;
          RBp = j;
        } else {
          for (i64 p = j - 1; p >= a; --p) {
            seg_events[si].push_back({x, s.c[p], static_cast<u8>((sideLf(j) << 1) | (sideR(p) ^ 1u))});
            absorbed[s.c[p]] = 1;
            abs_of[s.c[p]] = x;
            ++step;
          }
          if (a > 0) {
            seg_events[si].push_back({x, s.c[LBp], static_cast<u8>((sideLf(j) << 1) | (sideR(LBp) ^ 1u))});
            absorbed[s.c[LBp]] = 1;
            abs_of[s.c[LBp]] = x;
            ++step;
            st.nested++;
          }
          keyOwnL[j] = key_of(x, pass, step);
          owns[x] = 1;
          blo[x] = 0;
          keyL = key_of(x, pass, step);
          a = j + 1;
          LBp = j;
        }
        if (pass == 0) walked_f = true;
        walked_any = true;
      }
      if (walked_any) {
        if (LBp == j && RBp == j) {  // x holds the whole segment
          a = m;
          b = m - 1;
          RBp = -1;
        }
        st.turns++;
        turns++;
      } else {
        st.noop_turns++;
      }
    }
    st.max_turns = std::max(st.max_turns, turns);
    // ---- final adjacency -> edges of every surviving unit and of the two boundary nodes ----
    Edge const rawL = sv[s.c[0]].e[s.sideL[0]], rawR = sv[s.c[m - 1]].e[s.sideL[m - 1] ^ 1u];  // extremities -> boundaries
    u32 const sigL = dst_minus(rawL) ^ 1u, sigR = dst_minus(rawR) ^ 1u;                          // boundary sides facing the segment
    auto kind_of = [](u32 sm, u32 dm) { return static_cast<u8>(MakeFwdEdgeKind(sm ? MINUS : PLUS, dm ? MINUS : PLUS)); };
    if (tip_owner > 0) {
      bool const from_left = tip_owner == 1;
      u32 const T = from_left ? s.L : s.R, far = from_left ? s.R : s.L;
      u32 const sideT = src_minus(nodes[T].edges[0]);
      Edge const t2f{T, far, kind_of(sideT, (from_left ? sigR : sigL) ^ 1u)};
      final_edges.push_back({T, nodes[T].edges[0], t2f, tip_key});
      final_edges.push_back({far, from_left ? rawR.Mirror() : rawL.Mirror(), t2f.Mirror(), tip_key});
      continue;
    }
    struct U { i64 opos; i64 lo, hi; };
    std::vector<U> units;
    if (a > 0) units.push_back({LBp, 0, a - 1});
    for (i64 p = a; p <= b; ++p) units.push_back({p, p, p});
    if (b < m - 1) units.push_back({RBp, b + 1, m - 1});
    for (usize u = 0; u < units.size(); ++u) {
      U const& un = units[u];
      u32 const owner = s.c[un.opos];
      // towards R
      {
        bool const last = u + 1 == units.size();
        u32 const dstn = last ? s.R : s.c[units[u + 1].opos];
        u32 const dm = last ? (sigR ^ 1u) : (sideLf(units[u + 1].opos) ^ 1u);
        Edge const e{owner, dstn, kind_of(sideR(un.opos), dm)};
        final_edges.push_back({owner, sv[owner].e[sideR(un.opos)], e, keyOwnR[un.opos]});
        if (last) final_edges.push_back({s.R, rawR.Mirror(), e.Mirror(), keyR});
      }
      {
        bool const first = u == 0;
        u32 const dstn = first ? s.L : s.c[units[u - 1].opos];
        u32 const dm = first ? (sigL ^ 1u) : (sideR(units[u - 1].opos) ^ 1u);
        Edge const e{owner, dstn, kind_of(sideLf(un.opos), dm)};
        final_edges.push_back({owner, sv[owner].e[sideLf(un.opos)], e, keyOwnL[un.opos]});
        if (first) final_edges.push_back({s.L, rawL.Mirror(), e.Mirror(), keyL});
      }
    }
  }
  // ---- strings the kernel's way: every leaf byte (an owner's k-mer byte, an absorbed single's one base) finds its place
  // in its top-level block through one linear map per nesting level; bytes trimmed at a level vanish ----
  std::vector<std::string> kstr(n);
  i64 const Ki = static_cast<i64>(K);
  for (u32 i = 0; i < n; ++i)
    if (in_comp(i) && owns[i] && abs_of[i] == NONE) kstr[i].assign(static_cast<usize>(Ki - 1 + (bhi[i] - blo[i] + 1)), '?');
  auto sideRb = [&](u32 v) { return static_cast<u32>(side_l[v] ^ 1u); };
  auto prepend_count = [&](u32 y) { return sideRb(y) == 1u ? bhi[y] - npos[y] : npos[y] - blo[y]; };
  auto place = [&](u32 y, i64 idx, char base, bool comp) {  // byte `base` sits at index idx of block y's own string
    while (abs_of[y] != NONE) {
      u32 const z = abs_of[y];
      i64 const sz = bhi[y] - blo[y] + 1, LY = Ki - 1 + sz;
      bool const right = blo[y] > npos[z];
      u32 const sz_bit = right ? sideRb(z) : side_l[z];
      u32 const j = (right ? side_l[y] : sideRb(y)) ^ 1u;
      bool const rc = sz_bit != j, append = sz_bit == 0u;
      i64 const d1 = right ? blo[y] - npos[z] : npos[z] - bhi[y], d2 = d1 + sz - 1, PZ = prepend_count(z);
      i64 v = rc ? LY - 1 - idx : idx;
      if (append) {
        if (v < Ki - 1) return;
        idx = PZ + Ki + d1 - 1 + (v - (Ki - 1));
      } else {
        if (v > sz - 1) return;
        idx = PZ - d2 + v;
      }
      comp ^= rc;
      y = z;
    }
    kstr[y][static_cast<usize>(idx)] = comp ? Complement(base) : base;
  };
  for (u32 q = 0; q < n; ++q) {
    if (!in_comp(q)) continue;
    if (owns[q]) {
      i64 const P = prepend_count(q);
      for (i64 i = 0; i < Ki; ++i) place(q, P + i, nodes[q].seq[static_cast<usize>(i)], false);
    } else if (abs_of[q] != NONE) {
      u32 const y = abs_of[q];
      bool const right = npos[q] > npos[y];
      u32 const sy = right ? sideRb(y) : side_l[y];
      u32 const j = (right ? side_l[q] : sideRb(q)) ^ 1u;
      bool const rc = sy != j, append = sy == 0u;
      i64 const d = right ? npos[q] - npos[y] : npos[y] - npos[q], P = prepend_count(y);
      std::string const& kc = nodes[q].seq;
      char const base = append ? (rc ? kc[0] : kc[static_cast<usize>(Ki - 1)]) : (rc ? kc[static_cast<usize>(Ki - 1)] : kc[0]);
      place(y, append ? P + Ki + d - 1 : P - d, base, rc);
    }
  }
  // ---- replay the merges (counts, labels, strings) with the oracle's own Node::Merge ----
  for (auto const& evs : seg_events)
    for (auto const& ev : evs) g.MergeNode(nodes[ev.walker], nodes[ev.unit_owner], ev.kind);
  for (u32 i = 0; i < n; ++i)
    if (in_comp(i) && owns[i] && abs_of[i] == NONE && kstr[i] != nodes[i].seq) {
      st.str_bad++;
      if (getenv("CMODEL_V")) fprintf(stderr, "string of %u:\n want %s\n got  %s\n", i, nodes[i].seq.c_str(), kstr[i].c_str());
    }
  // ---- edges: content from the final adjacency; a rewritten edge moves to the END of its node's list at the time of its
  // last rewrite, an edge that never was keeps its place ----
  std::vector<std::vector<std::pair<i64, Edge>>> lists(n);
  for (u32 i = 0; i < n; ++i)
    if (in_comp(i) && !absorbed[i])
      for (auto const& e : nodes[i].edges) lists[i].push_back({-1, e});
  for (auto const& fe : final_edges) {
    if (absorbed[fe.node]) continue;
    bool hit = false;
    for (auto& pr : lists[fe.node])
      if (pr.second == fe.old_raw && !hit) {
        pr.second = fe.now;
        pr.first = fe.key;
        hit = true;
      }
    if (!hit) return false;
  }
  for (u32 i = 0; i < n; ++i) {
    if (!in_comp(i) || absorbed[i]) continue;
    std::stable_sort(lists[i].begin(), lists[i].end(), [](auto const& p, auto const& q) { return p.first < q.first; });
    nodes[i].edges.clear();
    for (auto const& pr : lists[i]) nodes[i].edges.push_back(pr.second);
  }
  for (u32 i = 0; i < n; ++i)
    if (in_comp(i) && absorbed[i]) {
      nodes[i].alive = false;
      nodes[i].edges.clear();
    }
  return true;
}

bool SameNodes(const std::vector<Node>& a, const std::vector<Node>& b, u32 comp, std::string* why) {
  for (usize i = 0; i < a.size(); ++i) {
    if (a[i].comp != comp) continue;
    if (a[i].alive != b[i].alive) { *why = "alive " + std::to_string(i); return false; }
    if (!a[i].alive) continue;
    if (a[i].seq != b[i].seq) { *why = "seq " + std::to_string(i); return false; }
    if (a[i].counts != b[i].counts) { *why = "counts " + std::to_string(i); return false; }
    if (a[i].role_counts[0] != b[i].role_counts[0] || a[i].role_counts[1] != b[i].role_counts[1]) { *why = "roles " + std::to_string(i); return false; }
    if (a[i].label != b[i].label) { *why = "label " + std::to_string(i); return false; }
    if (!(a[i].edges == b[i].edges)) {
      *why = "edges " + std::to_string(i) + " want";
      for (auto const& e : a[i].edges) *why += " " + std::to_string(e.dst) + ":" + std::to_string(e.kind);
      *why += " got";
      for (auto const& e : b[i].edges) *why += " " + std::to_string(e.dst) + ":" + std::to_string(e.kind);
      return false;
    }
  }
  return true;
}

}  // namespace
}  // namespace orc

extern "C" int model_check(const ma_params_t* prm, const ma_batch_t* b, int k, unsigned long long* stats_out, int verbose) {
  using namespace orc;
  Params P;
  P.min_k = prm->min_k; P.max_k = prm->max_k; P.k_step = prm->k_step; P.min_node_cov = prm->min_node_cov;
  P.min_anchor_cov = prm->min_anchor_cov; P.num_samples = prm->num_samples; P.min_anchor_len = prm->min_anchor_len;
  P.max_mismatch = prm->max_mismatch; P.bfs_limit = prm->bfs_limit;
  ModelStats st;
  for (int w = 0; w < b->n_windows; ++w) {
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
    std::string_view const ref(reinterpret_cast<const char*>(b->ref_bases) + b->ref_off[w], b->ref_off[w + 1] - b->ref_off[w]);
    Graph g(ref, reads, P);
    g.mK = static_cast<usize>(k);
    if (HasRepeat(ref, g.mK, P.max_mismatch)) continue;
    g.BuildGraph();
    g.RemoveLowCovNodes(0);
    auto const comps = g.MarkConnectedComponents();
    st.windows++;
    for (auto const& ci : comps) {
      auto const src = g.FindSource(ci.id), snk = g.FindSink(ci.id);
      if (!src.found || !snk.found || src.node == snk.node) continue;
      if (snk.off - src.off + g.mK < P.min_anchor_len) continue;
      g.mSource = src.node;
      g.mSink = snk.node;
      st.comps++;
      st.nodes += ci.n;
      st.max_nodes = std::max<u64>(st.max_nodes, ci.n);
      std::vector<Node> const raw = g.mNodes;
      g.CompressGraph(ci.id);
      std::vector<Node> const want = g.mNodes;
      g.mNodes = raw;
      std::vector<Node> got = raw;
      bool const ok = ModelCompress(got, ci.id, g.mSource, g.mSink, g.mK, g, st);
      if (!ok) {
        st.punts++;
        if (verbose) fprintf(stderr, "window %d comp %u: punt\n", w, ci.id);
        continue;
      }
      u64 alive = 0;
      for (auto const& nd : want) alive += nd.alive && nd.comp == ci.id;
      st.alive_after += alive;
      st.max_alive = std::max(st.max_alive, alive);
      std::string why;
      if (!SameNodes(want, got, ci.id, &why)) {
        st.mismatched++;
        if (verbose) fprintf(stderr, "window %d comp %u (n=%u, src %lld snk %lld): MISMATCH %s\n", w, ci.id, ci.n, (long long)g.mSource, (long long)g.mSink, why.c_str());
      }
    }
  }
  unsigned long long const v[] = {st.windows, st.comps, st.mismatched, st.punts, st.nodes, st.alive_after, st.segments, st.turns,
                                  st.max_turns, st.max_alive, st.max_deg, st.nested, st.max_seg, st.max_nodes, st.noop_turns, st.deg_gt4 + (st.str_bad << 32)};
  for (int i = 0; i < 16; ++i) stats_out[i] = v[i];
  return 0;
}
